// K8: the entity branch of MannerNewsEncoder (reference manner/models/components/news_encoder.py:40-72,
// 98-113, 119-124) — small f32 VALU kernels, batch-faithful to the reference's quirk Q1:
// nn.MultiheadAttention is built with batch_first=False but fed [N, E, D], so attention runs ACROSS THE
// N NEWS OF THE CALL at each entity slot e (sequence axis = news, batch axis = entity slot), with no
// key_padding_mask.  A news embedding therefore depends on the other news of the batch, exactly as
// in the reference; the table architecture is not offered for use_entities=True.
#include <math.h>

#include "common.h"

namespace manner {
namespace {

constexpr int LIN_ROWS = 8;      // rows per workgroup of the generic linear kernel

// y[r, o] = b[o] + sum_k x[r, k] W[o, k]   (nn.Linear; x rows may be gathered through `gather`)
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ gather,
                                                          int64_t n_src_rows, const float* __restrict__ W,
                                                          const float* __restrict__ b, int64_t R, int K, int O,
                                                          float* __restrict__ y, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) float xs[];     // [LIN_ROWS][K]
  const int64_t r0 = (int64_t)blockIdx.x * LIN_ROWS;
  const int nr = (int)min((int64_t)LIN_ROWS, R - r0);
  for (int i = threadIdx.x; i < nr * K; i += 256) {
    const int rr = i / K, k = i - rr * K;
    int64_t src = r0 + rr;
    if (gather) {                                    // nn.Embedding raises IndexError on such an id: flag it, read row 0
      src = gather[src];
      if (src < 0 || src >= n_src_rows) { if (status && k == 0) atomicOr(status, MANNER_HIP_STATUS_INDEX); src = 0; }
    }
    xs[rr * K + k] = x[src * K + k];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < O; o += 256) {
    const float* w = W + (size_t)o * K;
    float acc[LIN_ROWS];
#pragma unroll
    for (int rr = 0; rr < LIN_ROWS; ++rr) acc[rr] = 0.f;
    for (int k = 0; k < K; ++k) {
      const float wv = w[k];
#pragma unroll
      for (int rr = 0; rr < LIN_ROWS; ++rr) acc[rr] = fmaf(xs[rr * K + k], wv, acc[rr]);
    }
    const float bv = b ? b[o] : 0.f;
#pragma unroll
    for (int rr = 0; rr < LIN_ROWS; ++rr)
      if (rr < nr) y[(r0 + rr) * O + o] = acc[rr] + bv;
  }
}

// attention over the NEWS axis for one (entity slot, head): qkv [N, E, 3D] = [q | k | v], out [N, E, D].
// One thread per query news, keys streamed through LDS tiles of KT keys, online softmax.
template <int DH, int KT>
__global__ __launch_bounds__(256) void entity_attn_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                          int64_t N, int E, int D, int heads) {
  __shared__ float ks[KT * DH], vs[KT * DH];
  const int eh = blockIdx.x, e = eh / heads, hh = eh - e * heads;
  const int64_t n = (int64_t)blockIdx.y * 256 + threadIdx.x;
  const float scale = 1.0f / sqrtf((float)DH);
  const size_t ld = (size_t)E * 3 * D;
  const float* base = qkv + (size_t)e * 3 * D + hh * DH;
  float q[DH], o[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) { q[d] = n < N ? base[n * ld + d] * scale : 0.f; o[d] = 0.f; }   // q is scaled first, as torch does
  float mx = -INFINITY, sum = 0.f;
  for (int64_t t0 = 0; t0 < N; t0 += KT) {
    __syncthreads();
    for (int i = threadIdx.x; i < KT * DH; i += 256) {
      const int j = i / DH, d = i - j * DH;
      const int64_t kn = t0 + j;
      ks[i] = kn < N ? base[kn * ld + D + d] : 0.f;
      vs[i] = kn < N ? base[kn * ld + 2 * D + d] : 0.f;
    }
    __syncthreads();
    const int cnt = (int)min((int64_t)KT, N - t0);
    for (int j = 0; j < cnt; ++j) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) s = fmaf(q[d], ks[j * DH + d], s);
      if (s > mx) {
        const float f = expf(mx - s);
        sum *= f;
#pragma unroll
        for (int d = 0; d < DH; ++d) o[d] *= f;
        mx = s;
      }
      const float p = expf(s - mx);
      sum += p;
#pragma unroll
      for (int d = 0; d < DH; ++d) o[d] = fmaf(p, vs[j * DH + d], o[d]);
    }
  }
  if (n < N) {
    const float inv = 1.0f / sum;
    float* dst = out + (size_t)n * E * D + (size_t)e * D + hh * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) dst[d] = o[d] * inv;
  }
}

// nn.MultiheadAttention(batch_first=False) core on [N, E, 3D] projections: attention along axis 0
int launch_axis0_attention(const float* qkv, float* att, int64_t N, int64_t E, int D, int heads, hipStream_t s) {
  const int dh = D / heads;
  dim3 g((unsigned)(E * heads), (unsigned)((N + 255) / 256)), b(256);
  switch (dh) {
    case 4: hipLaunchKernelGGL((entity_attn_kernel<4, 256>), g, b, 0, s, qkv, att, N, (int)E, D, heads); break;
    case 8: hipLaunchKernelGGL((entity_attn_kernel<8, 256>), g, b, 0, s, qkv, att, N, (int)E, D, heads); break;
    case 10: hipLaunchKernelGGL((entity_attn_kernel<10, 256>), g, b, 0, s, qkv, att, N, (int)E, D, heads); break;
    case 16: hipLaunchKernelGGL((entity_attn_kernel<16, 256>), g, b, 0, s, qkv, att, N, (int)E, D, heads); break;
    case 32: hipLaunchKernelGGL((entity_attn_kernel<32, 64>), g, b, 0, s, qkv, att, N, (int)E, D, heads); break;
    case 48: hipLaunchKernelGGL((entity_attn_kernel<48, 64>), g, b, 0, s, qkv, att, N, (int)E, D, heads); break;
    case 64: hipLaunchKernelGGL((entity_attn_kernel<64, 64>), g, b, 0, s, qkv, att, N, (int)E, D, heads); break;
    default: return fail(MANNER_HIP_E_INVALID, "axis-0 attention: head_dim %d unsupported (4, 8, 10, 16, 32, 48, 64)", dh);
  }
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int launch_linear(const float* x, const int64_t* gather, int64_t n_src, const float* W, const float* b, int64_t R, int K,
                  int O, float* y, hipStream_t s, int32_t* status = nullptr) {
  if (R == 0) return MANNER_HIP_OK;
  if (K <= 0 || O <= 0 || K > 4096) return fail(MANNER_HIP_E_INVALID, "linear: K=%d O=%d unsupported (K <= 4096)", K, O);
  hipLaunchKernelGGL(linear_rows_kernel, dim3((unsigned)((R + LIN_ROWS - 1) / LIN_ROWS)), dim3(256), LIN_ROWS * K * sizeof(float), s,
                     x, gather, n_src, W, b, R, K, O, y, status);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

}  // namespace
}  // namespace manner

using namespace manner;

extern "C" {

int manner_hip_linear(const float* x, const float* weight, const float* bias, int64_t R, int32_t K, int32_t O, float* y,
                      manner_hip_stream_t stream) {
  if (R < 0 || (R > 0 && (!x || !weight || !y))) return fail(MANNER_HIP_E_INVALID, "linear: null pointer");
  return launch_linear(x, nullptr, 0, weight, bias, R, K, O, y, (hipStream_t)stream);
}

size_t manner_hip_entity_workspace_bytes(int64_t N, int64_t E, int32_t D) {
  if (N <= 0 || E <= 0 || D <= 0) return 0;
  return (size_t)(N * E) * (size_t)(6 * D + 1) * sizeof(float) + 1024;
}

int manner_hip_entity_encode(const int64_t* entity_ids, int64_t N, int64_t E, const float* table, int64_t n_entities,
                             int32_t D, int32_t heads, const float* in_proj_w, const float* in_proj_b,
                             const float* out_proj_w, const float* out_proj_b, const float* pool_w, const float* pool_b,
                             const float* pool_q, int32_t Q, float* out, void* workspace, size_t workspace_bytes,
                             int32_t* status, manner_hip_stream_t stream) {
  if (N == 0) return MANNER_HIP_OK;
  if (!entity_ids || !table || !in_proj_w || !in_proj_b || !out_proj_w || !out_proj_b || !pool_w || !pool_b || !pool_q || !out || !workspace)
    return fail(MANNER_HIP_E_INVALID, "entity_encode: null pointer");
  if (N < 0 || E <= 0 || D <= 0 || heads <= 0 || D % heads || D % 4 || n_entities <= 0)
    return fail(MANNER_HIP_E_INVALID, "entity_encode: N=%lld E=%lld D=%d heads=%d unsupported", (long long)N, (long long)E, D, heads);
  if (workspace_bytes < manner_hip_entity_workspace_bytes(N, E, D)) return fail(MANNER_HIP_E_WORKSPACE, "entity_encode: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int64_t R = N * E;
  float* qkv = static_cast<float*>(workspace);          // [R, 3D]
  float* att = qkv + (size_t)R * 3 * D;                 // [R, D]
  float* proj = att + (size_t)R * D;                    // [R, D]
  float* scratch = proj + (size_t)R * D;                // [R] logits of the additive pooler (+ D spare)
  int rc;
  // embedding lookup fused into the in-projection (rows gathered from the table)
  if ((rc = launch_linear(table, entity_ids, n_entities, in_proj_w, in_proj_b, R, D, 3 * D, qkv, s, status))) return rc;
  if ((rc = launch_axis0_attention(qkv, att, N, E, D, heads, s))) return rc;
  if ((rc = launch_linear(att, nullptr, 0, out_proj_w, out_proj_b, R, D, D, proj, s))) return rc;
  return manner_hip_additive_pool(proj, pool_w, pool_b, pool_q, N, E, D, Q, out, scratch, stream);
}

size_t manner_hip_mha_axis0_workspace_bytes(int64_t L0, int64_t B1, int32_t E) {
  if (L0 <= 0 || B1 <= 0 || E <= 0) return 0;
  return (size_t)round_up(L0 * B1, 128) * (size_t)(5 * E) * sizeof(float) + 1024;
}

int manner_hip_mha_axis0(const float* x, int64_t L0, int64_t B1, int32_t E, int32_t heads, const float* in_proj_w,
                         const float* in_proj_b, const float* out_proj_w, const float* out_proj_b, float* out, void* workspace,
                         size_t workspace_bytes, manner_hip_stream_t stream) {
  if (L0 == 0 || B1 == 0) return MANNER_HIP_OK;
  if (!x || !in_proj_w || !in_proj_b || !out_proj_w || !out_proj_b || !out || !workspace)
    return fail(MANNER_HIP_E_INVALID, "mha_axis0: null pointer");
  if (L0 < 0 || B1 < 0 || E <= 0 || heads <= 0 || E % heads)
    return fail(MANNER_HIP_E_INVALID, "mha_axis0: L0=%lld B1=%lld E=%d heads=%d", (long long)L0, (long long)B1, E, heads);
  if (workspace_bytes < manner_hip_mha_axis0_workspace_bytes(L0, B1, E)) return fail(MANNER_HIP_E_WORKSPACE, "mha_axis0: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int64_t R = L0 * B1, Rp = round_up(R, 128);
  float* qkv = static_cast<float*>(workspace);          // [Rp, 3E]
  float* att = qkv + (size_t)Rp * 3 * E;                // [Rp, E]
  int rc;
  if (E % 128 == 0 && Rp <= 0x7fffffff / (3 * (int64_t)E)) {
    // PLM-sized embeddings: both projections on the f32 MFMA GEMM (its tiles read whole 128-row panels: padded copy of x)
    float* xpad = att + (size_t)Rp * E;                 // [Rp, E]
    int32_t* mtot = reinterpret_cast<int32_t*>(xpad + (size_t)Rp * E);
    MANNER_HIP_TRY(hipMemcpyAsync(xpad, x, (size_t)R * E * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (Rp > R) MANNER_HIP_TRY(hipMemsetAsync(xpad + (size_t)R * E, 0, (size_t)(Rp - R) * E * sizeof(float), s));
    if ((rc = set_device_int(mtot, (int32_t)R, s))) return rc;
    if ((rc = gemm_tn(DT_F32, DT_F32, EPI_BIAS, xpad, in_proj_w, in_proj_b, nullptr, qkv, Rp, 3 * E, E, mtot, s))) return rc;
    if ((rc = launch_axis0_attention(qkv, att, L0, B1, E, heads, s))) return rc;
    if (Rp > R) MANNER_HIP_TRY(hipMemsetAsync(att + (size_t)R * E, 0, (size_t)(Rp - R) * E * sizeof(float), s));
    // the output GEMM stores rows < R only; `out` holds exactly R rows
    return gemm_tn(DT_F32, DT_F32, EPI_BIAS, att, out_proj_w, out_proj_b, nullptr, out, Rp, E, E, mtot, s);
  }
  if ((rc = launch_linear(x, nullptr, 0, in_proj_w, in_proj_b, R, E, 3 * E, qkv, s))) return rc;
  if ((rc = launch_axis0_attention(qkv, att, L0, B1, E, heads, s))) return rc;
  return launch_linear(att, nullptr, 0, out_proj_w, out_proj_b, R, E, E, out, s);
}

}  // extern "C"
