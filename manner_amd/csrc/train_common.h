// Shared pieces of the training path (train.hip, train_attn.hip): the counter-based dropout generator.  Not part of the ABI.
#pragma once
#include "common.h"

namespace manner {

// ---- dropout bits
// splitmix64 over (seed, site, index): 32 uniform bits; an element is KEPT when bits >= thr, thr = p * 2^32.
__host__ __device__ __forceinline__ uint32_t drop_bits(uint64_t seed, uint32_t site, uint64_t idx) {
  uint64_t z = (seed ^ ((uint64_t)site * 0xD6E8FEB86659FD93ull)) + (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}
struct Drop {
  uint64_t seed;
  uint32_t site, thr;
  float scale;          // 1 / (1 - p)
  __host__ __device__ __forceinline__ float apply(float v, uint64_t idx) const {
    return (thr == 0 || drop_bits(seed, site, idx) >= thr) ? v * scale : 0.f;
  }
};
static inline Drop make_drop(uint64_t seed, uint32_t site, float p) {
  Drop d;
  d.seed = seed;
  d.site = site;
  const double t = (double)p * 4294967296.0;
  d.thr = p <= 0.f ? 0u : (t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t);
  d.scale = p <= 0.f ? 1.f : 1.f / (1.f - p);
  return d;
}
// dropout sites of a layer (site = 8 * (layer + 1) + k; site 0..7 belong to the embeddings / the [CLS] output)
enum { SITE_EMB = 0, SITE_CLS = 1, SITE_ATTN = 0, SITE_PROJ = 1, SITE_FFN = 2 };
__host__ __device__ inline uint32_t layer_site(int layer, int k) { return 8u * (uint32_t)(layer + 1) + (uint32_t)k; }


// gemm.hip: Y [m, N] f32 = dropout(X W^T + bias) + R in one GEMM (16-bit operands, 256-tileable shapes, K >= 128)
int gemm_tn_drop_res(DType in, const void* X, const void* W, const float* bias, const float* residual, float* Y, int64_t m_bound, int N,
                     int K, const int* m_total, const Drop& drop, const int32_t* rowmap, hipStream_t stream);

// gemm.hip: the GeLU of the training FFN inside its neighbouring GEMMs (16-bit operands; round 4).
//   gemm_gelu_fusable : the shapes the two entry points below take (256-tileable, K >= 128, not a small problem)
//   gemm_tn_gelu_dual : Y [m, N] f32 = X W^T + bias (the saved pre-activation) and G16 [m, N] = gelu(Y) in the type `in`
//   gemm_tn_gelu_grad : Y16 [m, N] (type `in`) = round16(X W^T) * gelu'(pre [m, N] f32)   — d inter from d y2 . W2 in one launch
// GeLU and its derivative use the |error| <= 1.5e-7 erf of the inference engine's 16-bit epilogues (below the 16-bit rounding
// step of the outputs); the separate kernels (fp32 mode, small problems, MANNER_HIP_TRAIN_GELU_FUSED=0) keep erff.
bool gemm_gelu_fusable(DType in, int64_t m_bound, int N, int K);
int gemm_tn_gelu_dual(DType in, const void* X, const void* W, const float* bias, float* Y, void* G16, int64_t m_bound, int N, int K,
                      const int* m_total, hipStream_t stream);
int gemm_tn_gelu_grad(DType in, const void* X, const void* W, const float* zero_bias, const float* pre, void* Y16, int64_t m_bound, int N,
                      int K, const int* m_total, hipStream_t stream);
// Round 5 (16-bit saved activations): the same pair with the pre-activation SAVED IN THE 16-BIT TYPE — what the reference's 16-mixed
// autocast keeps for the backward (the f16 output of `intermediate.dense`; HF applies gelu to that tensor):
//   gemm_tn_gelu_dual16 : Pre16 [m, N] = round16(X W^T + bias) and G16 [m, N] = gelu(Pre16), both of the type `in`
//   gemm_tn_gelu_grad16 : Y16 [m, N] = round16(X W^T) * gelu'(pre16 [m, N])
int gemm_tn_gelu_dual16(DType in, const void* X, const void* W, const float* bias, void* Pre16, void* G16, int64_t m_bound, int N, int K,
                        const int* m_total, hipStream_t stream);
int gemm_tn_gelu_grad16(DType in, const void* X, const void* W, const float* zero_bias, const void* pre16, void* Y16, int64_t m_bound, int N,
                        int K, const int* m_total, hipStream_t stream);

// Training attention on the matrix pipe (train_attn.hip; 16-bit modes): S <= 128 keys per news, head_dim 64, one wave per
// (news, head).  qkv16 [m, 3H] = [Q | K | V] of the 16-bit type `dt`.
//   forward : ctx [m, H] f32 (may be NULL: round 5, 16-bit saved activations) and / or its 16-bit form ctx16 (may be NULL) =
//             dropout(softmax(q k^T / 8)) v;  ml[m, heads] = {row max of the RAW scores q.k, sum of exp((s - max) / 8)} so that the
//             backward rebuilds P without a reduction pass.
//   backward: dsum[m, heads] = dctx . ctx per head first — dctx / ctx f32 or (`*_is16`) of the type `dt`; an f32 dctx is also copied
//             to dctx16 (scratch [m_bound, H] of `dt`), a 16-bit one is used where it lies — then d qkv [m, 3H] as f32 rows (dqkv,
//             may be NULL) and / or 16-bit rows (dqkv16, may be NULL).  Dropout bits: element ((row * heads + head) * 256 + key).
int attn_train_mfma_forward(DType dt, const void* qkv16, float* ctx, void* ctx16, float2* ml, const int32_t* cu, int64_t n_news,
                            int heads, int H, int max_len, Drop drop, hipStream_t stream);
int attn_train_mfma_backward(DType dt, const void* qkv16, const void* dctx, bool dctx_is16, const void* ctx, bool ctx_is16, const float2* ml,
                             float* dqkv, void* dqkv16, void* dctx16, float* dsum, const int32_t* cu, int64_t n_news, int heads, int H,
                             int max_len, Drop drop, int64_t m_bound, const int* m_total, hipStream_t stream);

// wgrad.hip: out [slices][N, K] f32 = per-slice sums over token rows of dY[m, :]^T X[m, :] (row-major 16-bit operands read
// transposed from LDS; slices == 1 writes dW itself); N, K % 256 == 0, rows_per_slice % 32 == 0, zero_page >= 16 zero bytes
int wgrad_tr(DType dt, const void* dY, const void* X, float* out, int N, int K, int slices, int64_t rows_per_slice,
             const int* m_total, const void* zero_page, hipStream_t stream);

}  // namespace manner
