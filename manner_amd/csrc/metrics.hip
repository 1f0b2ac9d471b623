// Global binary AUC over all (score, label) pairs of an evaluation run: what
// torchmetrics AUROC(task="binary") computes for the reference (constructed at manner/models/cr_module.py:81,
// fed with the ragged preds/targets at :267-273).  Integer work end to end: a device radix sort of the
// order-preserving key of every score, then an exact Mann-Whitney count with ties at one half — the area under
// the ROC curve drawn through the distinct thresholds.  The sort/scan/compaction primitives are rocPRIM's
// (through hipCUB); the key, tie-group and reduction kernels are below.
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace manner {
namespace {

struct AucWs {            // carved from the caller's workspace
  uint32_t* key_in;
  uint32_t* key_out;
  uint32_t* lab_in;       // 1 = negative (what the prefix scan counts)
  uint32_t* lab_out;
  uint32_t* cneg;         // exclusive prefix count of negatives in sorted order, [n]
  uint32_t* head_pos;     // compacted start index of every tie group, [n]
  uint32_t* scalars;      // [0] outside-[0,1] flag, [1] number of tie groups, [2] total negatives
  unsigned long long* acc;   // [0] 2U
  void* cub;
  size_t cub_bytes;
};

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

struct HeadFlag {          // flag iterator functor: 1 where a new key value starts
  const uint32_t* key;
  __host__ __device__ uint32_t operator()(uint32_t i) const { return i == 0 || key[i] != key[i - 1]; }
};

size_t cub_bytes_for(int64_t n) {
  size_t a = 0, b = 0, c = 0;
  uint32_t* p = nullptr;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, p, p, p, p, (int)n);
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, p, p, (int)n);
  hipcub::CountingInputIterator<uint32_t> idx(0);
  hipcub::TransformInputIterator<uint32_t, HeadFlag, hipcub::CountingInputIterator<uint32_t>> flags(idx, HeadFlag{p});
  (void)hipcub::DeviceSelect::Flagged(nullptr, c, idx, flags, p, p, (int)n);
  return align256(a > b ? (a > c ? a : c) : (b > c ? b : c));
}

size_t carve(int64_t n, char* base, AucWs* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  };
  const size_t vec = (size_t)n * sizeof(uint32_t);
  char* k0 = take(vec); char* k1 = take(vec); char* l0 = take(vec); char* l1 = take(vec);
  char* cn = take(vec); char* hp = take(vec); char* sc = take(64); char* ac = take(64);
  const size_t cb = cub_bytes_for(n);
  char* cu = take(cb);
  if (w) {
    w->key_in = (uint32_t*)k0; w->key_out = (uint32_t*)k1; w->lab_in = (uint32_t*)l0; w->lab_out = (uint32_t*)l1;
    w->cneg = (uint32_t*)cn; w->head_pos = (uint32_t*)hp; w->scalars = (uint32_t*)sc;
    w->acc = (unsigned long long*)ac; w->cub = cu; w->cub_bytes = cb;
  }
  return off;
}

// any score outside [0, 1] (or NaN) -> the whole vector goes through the logistic function first, as
// torchmetrics' binary format step does.
__global__ __launch_bounds__(256) void auc_range_kernel(const float* __restrict__ s, int64_t n, uint32_t* scalars) {
  bool out = false;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
    const float v = s[i];
    out |= !(v >= 0.f && v <= 1.f);
  }
  if (__any(out) && (threadIdx.x & 63) == 0) atomicOr(scalars, 1u);
}

// ascending order-preserving key of a float: flip all bits of negatives, the sign bit of the rest
__global__ __launch_bounds__(256) void auc_key_kernel(const float* __restrict__ s, const float* __restrict__ lab, int64_t n,
                                                     int sigmoid_rule, const uint32_t* scalars,
                                                     uint32_t* __restrict__ key, uint32_t* __restrict__ neg) {
  const bool squash = sigmoid_rule && scalars[0];
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += 256ll * gridDim.x) {
    float v = s[i];
    if (squash) v = 1.0f / (1.0f + expf(-v));
    v += 0.0f;                                   // -0 and +0 are one threshold
    const uint32_t u = __float_as_uint(v);
    key[i] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    neg[i] = lab[i] > 0.5f ? 0u : 1u;
  }
}

__global__ void auc_total_kernel(const uint32_t* cneg, const uint32_t* lab_sorted, int64_t n, uint32_t* scalars) {
  scalars[2] = cneg[n - 1] + lab_sorted[n - 1];
}

// one thread per tie group [s, e): its positives each beat the cneg[s] negatives below and tie with the
// cneg[e] - cneg[s] negatives inside, so the group adds P_g * (cneg[s] + cneg[e]) to 2U.
__global__ __launch_bounds__(256) void auc_groups_kernel(const uint32_t* __restrict__ head_pos, const uint32_t* __restrict__ cneg,
                                                        const uint32_t* scalars, int64_t n, unsigned long long* acc) {
  const uint32_t groups = scalars[1], neg_total = scalars[2];
  unsigned long long part = 0;
  for (int64_t g = blockIdx.x * 256ll + threadIdx.x; g < groups; g += 256ll * gridDim.x) {
    const uint32_t s = head_pos[g];
    const uint32_t e = g + 1 < groups ? head_pos[g + 1] : (uint32_t)n;
    const uint32_t cs = cneg[s], ce = e < n ? cneg[e] : neg_total;
    const unsigned long long pos = (unsigned long long)(e - s) - (ce - cs);
    part += pos * ((unsigned long long)cs + ce);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((threadIdx.x & 63) == 0 && part) atomicAdd(acc, part);
}

__global__ void auc_final_kernel(const unsigned long long* acc, const uint32_t* scalars, int64_t n, double* auc,
                                 int64_t* counts) {
  const unsigned long long neg = scalars[2], pos = (unsigned long long)n - neg, u2 = acc[0];
  if (auc) auc[0] = (pos == 0 || neg == 0) ? 0.0 : (double)u2 / (2.0 * (double)pos * (double)neg);
  if (counts) { counts[0] = (int64_t)u2; counts[1] = (int64_t)pos; counts[2] = (int64_t)neg; }
}

// ---- evaluation loss of CRModule.model_step (cr_module.py:140-171), one wave per impression, ragged scores.
// mode 0: SupConLoss on the score matrix (losses.py:12-40): -mean_{pos} (s_j/T - logsumexp_{real j} s_j/T)
// mode 1: nn.CrossEntropyLoss(scores[B,Cmax], y_true[B,Cmax]) with probability targets: the zero-padded entries of the
//         dense row take part in the softmax (c_max - c_i scores of 0.0), as they do in the reference.
__global__ __launch_bounds__(256) void eval_loss_kernel(const float* __restrict__ scores, const float* __restrict__ labels,
                                                       const int64_t* __restrict__ off, int64_t B, int mode, float inv_t,
                                                       int64_t c_max, float tiny, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= B) return;
  const int64_t c0 = off[i], c1 = off[i + 1];
  const int64_t pad = mode == 1 ? c_max - (c1 - c0) : 0;
  float mx = pad > 0 ? 0.f : -INFINITY;
  for (int64_t j = c0 + lane; j < c1; j += 64) mx = fmaxf(mx, scores[j] * inv_t);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float se = 0.f, sp = 0.f, np = 0.f;
  for (int64_t j = c0 + lane; j < c1; j += 64) {
    const float v = scores[j] * inv_t - mx;          // contraction is off in this kernel: the row maximum gives exactly 0
    se += expf(v);
    const float y = labels[j];
    if (mode == 1) { sp = fmaf(y, v, sp); np += y; }
    else if (y > 0.5f) { sp += v; np += 1.f; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { se += __shfl_xor(se, o, 64); sp += __shfl_xor(sp, o, 64); np += __shfl_xor(np, o, 64); }
  if (lane == 0) {
    if (pad > 0) se += (float)pad * expf(-mx);
    const float lse = logf(se);
    // sum_pos (v - lse) = sp - np * lse;  SupCon divides by (n_pos + tiny): 0 without a positive
    out[i] = mode == 1 ? -(sp - np * lse) : -(sp - np * lse) / (np + tiny);
  }
}

}  // namespace
}  // namespace manner

using namespace manner;

extern "C" {

size_t manner_hip_auc_workspace_bytes(int64_t n) {
  if (n <= 0 || n > 0x7fffffffll) return 0;
  return carve(n, nullptr, nullptr);
}

int manner_hip_auc(const float* scores, const float* labels, int64_t n, int32_t sigmoid_rule, void* workspace,
                   size_t workspace_bytes, double* auc, int64_t* counts, manner_hip_stream_t stream_) {
  if (n <= 0 || n > 0x7fffffffll || !scores || !labels || !workspace || (!auc && !counts))
    return fail(MANNER_HIP_E_INVALID, "auc: bad argument");
  AucWs w;
  if (carve(n, (char*)workspace, &w) > workspace_bytes)
    return fail(MANNER_HIP_E_WORKSPACE, "auc: workspace %zu B < %zu B", workspace_bytes, carve(n, nullptr, nullptr));
  hipStream_t stream = (hipStream_t)stream_;
  MANNER_HIP_TRY(hipMemsetAsync(w.scalars, 0, 64, stream));
  MANNER_HIP_TRY(hipMemsetAsync(w.acc, 0, 64, stream));
  const unsigned grid = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  if (sigmoid_rule) hipLaunchKernelGGL(auc_range_kernel, dim3(grid), dim3(256), 0, stream, scores, n, w.scalars);
  hipLaunchKernelGGL(auc_key_kernel, dim3(grid), dim3(256), 0, stream, scores, labels, n, (int)sigmoid_rule, w.scalars,
                     w.key_in, w.lab_in);
  MANNER_LAUNCH_CHECK();
  size_t cb = w.cub_bytes;
  MANNER_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(w.cub, cb, w.key_in, w.key_out, w.lab_in, w.lab_out, (int)n, 0, 32, stream));
  cb = w.cub_bytes;
  MANNER_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w.cub, cb, w.lab_out, w.cneg, (int)n, stream));
  hipLaunchKernelGGL(auc_total_kernel, dim3(1), dim3(1), 0, stream, w.cneg, w.lab_out, n, w.scalars);
  hipcub::CountingInputIterator<uint32_t> idx(0);
  hipcub::TransformInputIterator<uint32_t, HeadFlag, hipcub::CountingInputIterator<uint32_t>> flags(idx, HeadFlag{w.key_out});
  cb = w.cub_bytes;
  MANNER_HIP_TRY(hipcub::DeviceSelect::Flagged(w.cub, cb, idx, flags, w.head_pos, w.scalars + 1, (int)n, stream));
  hipLaunchKernelGGL(auc_groups_kernel, dim3(grid), dim3(256), 0, stream, w.head_pos, w.cneg, w.scalars, n, w.acc);
  hipLaunchKernelGGL(auc_final_kernel, dim3(1), dim3(1), 0, stream, w.acc, w.scalars, n, auc, counts);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_eval_loss(const float* scores, const float* labels, const int64_t* cand_off, int64_t B, int32_t mode,
                         float temperature, int64_t c_max, float* losses, manner_hip_stream_t stream) {
  if (B == 0) return MANNER_HIP_OK;
  if (!scores || !labels || !cand_off || !losses || B < 0 || mode < 0 || mode > 1 || !(temperature > 0.f) || (mode == 1 && c_max < 1))
    return fail(MANNER_HIP_E_INVALID, "eval_loss: bad argument");
  hipLaunchKernelGGL(eval_loss_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, scores, labels, cand_off, B,
                     (int)mode, 1.0f / temperature, c_max, 1.17549435e-38f, losses);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

}  // extern "C"
