// Global binary AUC over all (score, label) pairs of an evaluation run: what
// torchmetrics AUROC(task="binary") computes for the reference (constructed at manner/models/cr_module.py:81,
// fed with the ragged preds/targets at :267-273).  Integer work end to end, every kernel below hand-written
// (no hipCUB / rocPRIM):
//   1. order-preserving u32 key of every score; the keys are split by label into a negatives and a positives
//      array (wave-aggregated appends — their order is irrelevant, see 2 and 3);
//   2. the NEGATIVES are sorted: LSD radix sort, 8-bit digits, 4 passes of {per-block digit histogram, one
//      exclusive scan of the [digit][block] counts, stable scatter} — the stable rank of a key inside its block is
//      (keys of the same digit in earlier waves) + (earlier rounds of its wave) + (lower lanes of its round), the
//      last from 8 ballots;
//   3. every POSITIVE key binary-searches the sorted negatives: it beats lb = #(neg < key) of them and ties with
//      ub - lb, so it adds lb + ub to 2U — the exact Mann-Whitney count with ties at one half, i.e. the area under
//      the ROC curve drawn through the distinct thresholds.  u64 integer atomics: order-free, deterministic.
#include "common.h"

namespace manner {
namespace {

constexpr int RS_ITEMS = 16;                       // keys per lane
constexpr int RS_TILE = 256 * RS_ITEMS;
constexpr int SCAN_SEG = 256 * 16;                 // entries per workgroup of the two-level histogram scan            // keys per workgroup (4 waves x 64 lanes x 8 rounds)

struct AucWs {            // carved from the caller's workspace
  uint32_t* neg_a;        // negatives' keys (ping)
  uint32_t* neg_b;        // (pong)
  uint32_t* pos;          // positives' keys
  uint32_t* hist;         // [256][blocks] digit counts of one pass
  uint32_t* seg;          // per-segment totals of the two-level scan
  uint32_t* scalars;      // [0] outside-[0,1] flag, [1] #negatives, [2] #positives
  unsigned long long* acc;   // [0] 2U
};

size_t align256(size_t v) { return (v + 255) / 256 * 256; }
int64_t rs_blocks(int64_t n) { return (n + RS_TILE - 1) / RS_TILE; }

size_t carve(int64_t n, char* base, AucWs* w) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  };
  const size_t vec = (size_t)n * sizeof(uint32_t);
  char* a = take(vec); char* b = take(vec); char* p = take(vec);
  char* h = take((size_t)256 * rs_blocks(n) * sizeof(uint32_t));
  char* sg = take((size_t)(256 * rs_blocks(n) / SCAN_SEG + 1) * sizeof(uint32_t));
  char* sc = take(64); char* ac = take(64);
  if (w) {
    w->neg_a = (uint32_t*)a; w->neg_b = (uint32_t*)b; w->pos = (uint32_t*)p; w->hist = (uint32_t*)h; w->seg = (uint32_t*)sg;
    w->scalars = (uint32_t*)sc; w->acc = (unsigned long long*)ac;
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
  // one atomic per workgroup at most, and none once the flag is up (a racy read is fine: the flag only ever rises) —
  // with every score outside [0, 1], the common case for raw dot products, a per-wave atomicOr was 8 k serialised atomics
  __shared__ int any_out;
  if (threadIdx.x == 0) any_out = 0;
  __syncthreads();
  if (__any(out) && (threadIdx.x & 63) == 0) any_out = 1;
  __syncthreads();
  if (threadIdx.x == 0 && any_out && *reinterpret_cast<volatile uint32_t*>(scalars) == 0u) atomicOr(scalars, 1u);
}

// ascending order-preserving key of a float (flip all bits of negatives, the sign bit of the rest), appended to the
// negatives or the positives.  A workgroup owns RS_TILE consecutive pairs: it counts its two classes (ballots + a
// 64-entry scan in LDS), reserves its two output ranges with ONE atomic each — a per-wave atomic on the two global
// counters serialised 80 k returning atomics and cost 0.9 ms — and writes; the order inside the arrays is irrelevant.
__global__ __launch_bounds__(256) void auc_split_kernel(const float* __restrict__ s, const float* __restrict__ lab, int64_t n,
                                                       int sigmoid_rule, uint32_t* scalars, uint32_t* __restrict__ neg,
                                                       uint32_t* __restrict__ pos) {
  __shared__ uint32_t cnt[2][RS_ITEMS * 4];       // [class][round * 4 + wave] -> exclusive offsets
  __shared__ uint32_t base[2];
  const bool squash = sigmoid_rule && scalars[0];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
  const int64_t i0 = (int64_t)blockIdx.x * RS_TILE;
  uint32_t key[RS_ITEMS];
  unsigned long long mneg[RS_ITEMS], mpos[RS_ITEMS];
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    const int64_t i = i0 + r * 256 + threadIdx.x;
    const bool valid = i < n;
    bool is_neg = false;
    key[r] = 0;
    if (valid) {
      float v = s[i];
      if (squash) v = 1.0f / (1.0f + expf(-v));
      v += 0.0f;                                   // -0 and +0 are one threshold
      const uint32_t u = __float_as_uint(v);
      key[r] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      is_neg = !(lab[i] > 0.5f);
    }
    mneg[r] = __ballot(valid && is_neg);
    mpos[r] = __ballot(valid && !is_neg);
    if (lane == 0) { cnt[0][r * 4 + wave] = (uint32_t)__popcll(mneg[r]); cnt[1][r * 4 + wave] = (uint32_t)__popcll(mpos[r]); }
  }
  __syncthreads();
  if (threadIdx.x < 2) {                           // one thread per class: exclusive scan of the 64 counts, reserve the range
    uint32_t run = 0;
    for (int j = 0; j < RS_ITEMS * 4; ++j) { const uint32_t c = cnt[threadIdx.x][j]; cnt[threadIdx.x][j] = run; run += c; }
    base[threadIdx.x] = run ? atomicAdd(scalars + 1 + threadIdx.x, run) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    const unsigned long long bit = 1ull << lane;
    if (mneg[r] & bit) neg[base[0] + cnt[0][r * 4 + wave] + (uint32_t)__popcll(mneg[r] & below)] = key[r];
    else if (mpos[r] & bit) pos[base[1] + cnt[1][r * 4 + wave] + (uint32_t)__popcll(mpos[r] & below)] = key[r];
  }
}

// lanes of the wave whose (valid) key has the same 8-bit digit as this lane's: 8 ballots.  Score keys share their high
// bytes (same sign / exponent), so per-lane atomics on a digit counter would serialise 64 deep; one lane per distinct
// digit adds the whole group's count instead.
__device__ __forceinline__ unsigned long long same_digit_lanes(uint32_t d, bool valid) {
  unsigned long long same = __ballot(valid);
#pragma unroll
  for (int bit = 0; bit < 8; ++bit) {
    const unsigned long long m = __ballot((d >> bit) & 1u);
    same &= ((d >> bit) & 1u) ? m : ~m;
  }
  return same;
}

// ---- LSD radix sort of keys[0 .. *count): one pass = hist -> scan -> scatter on digit (key >> shift) & 255.
// Grids are sized for the upper bound n; workgroups past *count have nothing to do (their counts are 0).
__global__ __launch_bounds__(256) void rs_hist_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ count,
                                                     int shift, uint32_t* __restrict__ hist, int blocks) {
  __shared__ uint32_t cnt[256];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t n = *count, base = (int64_t)blockIdx.x * RS_TILE;
  const int lane = threadIdx.x & 63;
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    const bool valid = i < n;
    const uint32_t d = valid ? (keys[i] >> shift) & 255u : 0u;
    const unsigned long long same = same_digit_lanes(d, valid);
    if (valid && (same & below) == 0) atomicAdd(&cnt[d], (uint32_t)__popcll(same));
  }
  __syncthreads();
  hist[(size_t)threadIdx.x * blocks + blockIdx.x] = cnt[threadIdx.x];
}

// exclusive scan of hist[0 .. len) in place (digit-major, so the result is each (digit, block)'s first output slot),
// two levels: per-segment totals (SCAN_SEG entries per workgroup) -> one workgroup scans the totals -> every workgroup
// rescans its segment from its base.  A thread owns 16 consecutive entries.

__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* sm /*[256]*/, uint32_t* total) {
  const int t = threadIdx.x;
  sm[t] = v;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t a = t >= o ? sm[t - o] : 0;
    __syncthreads();
    sm[t] += a;
    __syncthreads();
  }
  if (total) *total = sm[255];
  return sm[t] - v;
}

__global__ __launch_bounds__(256) void rs_scan_totals_kernel(const uint32_t* __restrict__ hist, int64_t len,
                                                            uint32_t* __restrict__ seg) {
  const int64_t base = (int64_t)blockIdx.x * SCAN_SEG + threadIdx.x * 16;
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) if (base + i < len) s += hist[base + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  __shared__ uint32_t part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) seg[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(256) void rs_scan_segs_kernel(uint32_t* __restrict__ seg, int n_seg) {
  __shared__ uint32_t sm[256];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int s0 = 0; s0 < n_seg; s0 += 256) {
    const int i = s0 + threadIdx.x;
    const uint32_t v = i < n_seg ? seg[i] : 0;
    uint32_t tot;
    const uint32_t ex = block_exclusive_scan_256(v, sm, &tot);
    const uint32_t c = carry;
    if (i < n_seg) seg[i] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void rs_scan_apply_kernel(uint32_t* __restrict__ hist, int64_t len,
                                                           const uint32_t* __restrict__ seg) {
  __shared__ uint32_t sm[256];
  const int64_t base = (int64_t)blockIdx.x * SCAN_SEG + threadIdx.x * 16;
  uint32_t v[16], s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = base + i < len ? hist[base + i] : 0; s += v[i]; }
  uint32_t run = seg[blockIdx.x] + block_exclusive_scan_256(s, sm, nullptr);
#pragma unroll
  for (int i = 0; i < 16; ++i) { if (base + i < len) hist[base + i] = run; run += v[i]; }
}

// stable scatter.  Wave w of a workgroup owns the contiguous keys [base + w*1024, +1024) in 16 rounds of 64; the order
// (wave, round, lane) is the input order, so ranking in that order keeps the pass stable.
__global__ __launch_bounds__(256) void rs_scatter_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                        const uint32_t* __restrict__ count, int shift,
                                                        const uint32_t* __restrict__ hist, int blocks) {
  __shared__ uint32_t run[4][256];                 // per wave: next output slot of each digit
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n = *count, base = (int64_t)blockIdx.x * RS_TILE + wave * (64 * RS_ITEMS);
  if ((int64_t)blockIdx.x * RS_TILE >= n) return;                   // workgroup-uniform
  for (int d = lane; d < 256; d += 64) run[wave][d] = 0;
  __syncthreads();
  const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
  uint32_t key[RS_ITEMS];
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool valid = i < n;
    key[r] = valid ? in[i] : 0u;
    const uint32_t d = (key[r] >> shift) & 255u;
    const unsigned long long same = same_digit_lanes(d, valid);
    if (valid && (same & below) == 0) run[wave][d] += (uint32_t)__popcll(same);   // this wave's digit counts (wave-private row)
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {                                                 // thread d: first slot of digit d for each wave, in wave order
    const int d = threadIdx.x;
    uint32_t o = hist[(size_t)d * blocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; ++w) { const uint32_t c = run[w][d]; run[w][d] = o; o += c; }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ITEMS; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool valid = i < n;
    const uint32_t d = (key[r] >> shift) & 255u;
    const unsigned long long same = same_digit_lanes(d, valid);      // lanes of this round holding the same digit
    if (valid) {
      const uint32_t slot = run[wave][d] + (uint32_t)__popcll(same & below);
      out[slot] = key[r];
    }
    __builtin_amdgcn_wave_barrier();                // every lane has read run[] before the leaders advance it
    if (valid && (same & below) == 0) run[wave][d] += (uint32_t)__popcll(same);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// every positive key against the sorted negatives: 2U += #(neg < key) + #(neg <= key)
__global__ __launch_bounds__(256) void auc_count_kernel(const uint32_t* __restrict__ pos, const uint32_t* __restrict__ neg,
                                                       const uint32_t* __restrict__ scalars, unsigned long long* acc) {
  const uint32_t n_neg = scalars[1], n_pos = scalars[2];
  unsigned long long part = 0;
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n_pos; i += 256ll * gridDim.x) {
    const uint32_t k = pos[i];
    uint32_t lo = 0, hi = n_neg;                    // lower bound: first index with neg >= k
    while (lo < hi) { const uint32_t mid = lo + ((hi - lo) >> 1); if (neg[mid] < k) lo = mid + 1; else hi = mid; }
    uint32_t lo2 = lo, hi2 = n_neg;                 // upper bound: first index with neg > k
    while (lo2 < hi2) { const uint32_t mid = lo2 + ((hi2 - lo2) >> 1); if (neg[mid] <= k) lo2 = mid + 1; else hi2 = mid; }
    part += (unsigned long long)lo + lo2;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((threadIdx.x & 63) == 0 && part) atomicAdd(acc, part);
}

__global__ void auc_final_kernel(const unsigned long long* acc, const uint32_t* scalars, double* auc, int64_t* counts) {
  const unsigned long long neg = scalars[1], pos = scalars[2], u2 = acc[0];
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
  hipLaunchKernelGGL(auc_split_kernel, dim3((unsigned)rs_blocks(n)), dim3(256), 0, stream, scores, labels, n, (int)sigmoid_rule,
                     w.scalars, w.neg_a, w.pos);
  MANNER_LAUNCH_CHECK();
  const int blocks = (int)rs_blocks(n);
  const int64_t hist_len = (int64_t)256 * blocks;
  const int n_seg = (int)((hist_len + SCAN_SEG - 1) / SCAN_SEG);
  uint32_t *src = w.neg_a, *dst = w.neg_b;
  for (int shift = 0; shift < 32; shift += 8) {
    hipLaunchKernelGGL(rs_hist_kernel, dim3(blocks), dim3(256), 0, stream, src, w.scalars + 1, shift, w.hist, blocks);
    hipLaunchKernelGGL(rs_scan_totals_kernel, dim3(n_seg), dim3(256), 0, stream, w.hist, hist_len, w.seg);
    hipLaunchKernelGGL(rs_scan_segs_kernel, dim3(1), dim3(256), 0, stream, w.seg, n_seg);
    hipLaunchKernelGGL(rs_scan_apply_kernel, dim3(n_seg), dim3(256), 0, stream, w.hist, hist_len, w.seg);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, w.scalars + 1, shift, w.hist, blocks);
    MANNER_LAUNCH_CHECK();
    uint32_t* t = src; src = dst; dst = t;
  }
  hipLaunchKernelGGL(auc_count_kernel, dim3(grid), dim3(256), 0, stream, w.pos, src, w.scalars, w.acc);
  hipLaunchKernelGGL(auc_final_kernel, dim3(1), dim3(1), 0, stream, w.acc, w.scalars, auc, counts);
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
