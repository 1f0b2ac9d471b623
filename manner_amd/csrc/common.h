// Shared declarations of the gfx950 MANNeR hot-path library (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "manner_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

namespace manner {

int fail(int code, const char* fmt, ...);   // records the thread-local error text, returns code

#define MANNER_HIP_TRY(expr)                                                            \
  do {                                                                                  \
    hipError_t e__ = (expr);                                                            \
    if (e__ != hipSuccess)                                                              \
      return ::manner::fail(MANNER_HIP_E_RUNTIME, "%s: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

#define MANNER_LAUNCH_CHECK()                                                           \
  do {                                                                                  \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess)                                                              \
      return ::manner::fail(MANNER_HIP_E_RUNTIME, "kernel launch at %s:%d: %s", __FILE__, __LINE__, \
                            hipGetErrorString(e__));                                    \
  } while (0)

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Per-DEVICE one-time state (a process may drive several GPUs through different handles): the current device id,
// clamped to the cache size.
constexpr int MAX_DEVICES = 64;
static inline int current_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev < MAX_DEVICES ? dev : MAX_DEVICES - 1;
}
int device_cus();   // compute units of the current device (gemm.hip)

// ------------------------------------------------------------------ GEMM (gemm.hip)
enum Epilogue { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RES = 2, EPI_NORM = 3, EPI_NORM_GELU = 4, EPI_NRES = 5,
                EPI_BIAS_RES_F32 = 6 /* bf16 operands, f32 residual and output: the bf16x3 parity mode */,
                EPI_BIAS_GELU_SPLIT3 = 7 /* x3 modes' FFN1: exact-erf gelu(acc + bias) written as the NEXT GEMM's split operand
                                            [hi | hi | lo], row stride 3 N of the 16-bit type (the f32 intermediate never exists) */,
                EPI_BIAS_GELU_DUAL = 8 /* training forward FFN1 (gemm_tn_gelu_dual): f32 pre-activation acc + bias AND its gelu in
                                          the 16-bit operand type, one launch */,
                EPI_GELU_GRAD = 9 /* training backward (gemm_tn_gelu_grad): 16-bit out = round16(acc) * gelu'(saved f32 pre-activation) */,
                EPI_BIAS_GELU_DUAL16 = 10 /* round 5, 16-bit saved activations: pre16 = round16(acc + bias) AND gelu(pre16), both in the
                                             16-bit operand type (what the reference's 16-mixed autocast saves: the f16 Linear output) */,
                EPI_GELU_GRAD16 = 11 /* ... and the backward: 16-bit out = round16(acc) * gelu'(pre16) */ };
enum DType { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };
static inline bool is_16bit(DType d) { return d != DT_F32; }

// The two 16-bit element types of the MFMA encoder path (bf16: 8 mantissa bits, f32's exponent range; f16: 11 mantissa
// bits, what the reference's own `precision: 16-mixed` autocast computes in): same size, same MFMA rate, same kernels.
template <typename T> struct E16;
template <> struct E16<bf16_t> {
  typedef bf16x8 v8; typedef bf16x4 v4;
  static constexpr DType dtype = DT_BF16;
  static __device__ __forceinline__ f32x4 mfma16(const v8& a, const v8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mfma32(const v8& a, const v8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct E16<f16_t> {
  typedef f16x8 v8; typedef f16x4 v4;
  static constexpr DType dtype = DT_F16;
  static __device__ __forceinline__ f32x4 mfma16(const v8& a, const v8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mfma32(const v8& a, const v8& b, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// Y[m, n] = epi( sum_k X[m, k] * W[n, k] + bias[n] )  for m < *m_total (device scalar).
// X [m_bound, K], W [N, K] row-major of dtype `in`; Y [m_bound, N] of dtype `out`;
// residual [m_bound, N] of dtype `in` (EPI_BIAS_RES only).  m_bound is a multiple of 128 and all
// buffers hold that many rows; N % 128 == 0; K*sizeof(in) % 128 == 0.
int gemm_tn(DType in, DType out, Epilogue epi, const void* X, const void* W, const float* bias,
            const void* residual, void* Y, int64_t m_bound, int N, int K, const int* m_total,
            hipStream_t stream);

// `batch` independent problems Y_b [rows, N] f32 = X_b [rows, K] . W_b [N, K]^T + bias (16-bit operands `in`; rows, N
// multiples of 256, K % 64 == 0, K >= 128) in ONE launch; problem b lives at element offsets b * {xs, ws, ys}.
int gemm_tn_batched16(DType in, const void* X, const void* W, const float* bias, float* Y, int batch, int64_t xs, int64_t ws,
                      int64_t ys, int rows, int N, int K, const int* m_total, hipStream_t stream);

// Deferred-LayerNorm GEMMs (16-bit elements `dt` in/out, 256x256 tiles: N % 256 == 0, m_bound % 256 == 0, K >= 128).  The residual
// stream holds pre-LayerNorm sums `raw` plus per-row {mean, rstd} (`mr`, float2 [m_bound]):
//   EPI_NORM / EPI_NORM_GELU : Y = [gelu]( rstd * (X W'^T - mean * vec) + bias ),  X = raw, W' = gamma-folded weight,
//                              vec = c1 (column sums of W'), bias = c2 (see fold_layernorm)
//   EPI_NRES (in place on Y) : Y = X W^T + bias + ((Y - mean) * rstd) * vec,  vec = gamma, bias = b + beta;
//                              also writes part[N/64][m_bound] = per-wave {sum, sum of squares} of the new rows.
// m_exact >= 0: the host knows *m_total (a launch whose rows fill whole rounds then skips the tail launch of the round-aware split)
// fin (EPI_NRES, optional): the reduction of `part` to the NEXT {mean, rstd} inside the launch — the workgroup whose tile completes a
// row panel last (an agent-scope arrival counter per panel, self-resetting; `arrive` >= m_bound / 192 + 1 zeroed ints owned by the
// caller's stream) reduces that panel's partial sums exactly as dln_finalize does (dln_row_stats: the same bits).  `done` tells the
// caller whether this launch took it (persistent kernels) or dln_finalize still has to run (128x128 kernel, round-aware split, A/B forms).
struct DlnFinalize { void* mr_out; int32_t* arrive; float eps; bool done; };
int gemm_tn_dln(DType dt, Epilogue epi, const void* X, const void* W, const float* bias, const float* vec, const void* mr,
                void* part, void* Y, int64_t m_bound, int N, int K, const int* m_total, hipStream_t stream, int64_t m_exact = -1,
                DlnFinalize* fin = nullptr);

// K11 logits[r] = sum_j tanh(x[r] . W[j] + b[j]) q[j] on the f32 MFMA (D % 32 == 0, Q <= 256); x [R, D], W [Q, D]
int pool_logits_mfma(const float* x, const float* W, const float* bias, const float* query, int64_t R, int D, int Q,
                     float* logits, hipStream_t stream);

// K11 in one pass over x (pool.hip): split (x3) logits on the f16 matrix pipe from register-resident scaled-half hi/lo rows, softmax, weighted sum
bool pool_fused_supported(int64_t B, int64_t S, int D, int Q);
size_t pool_fused_workspace_bytes(int D, int Q);
int pool_fused(const float* x, const float* W, const float* bias, const float* query, int64_t B, int64_t S, int D, int Q, float* out,
               void* workspace, size_t workspace_bytes, hipStream_t stream);

// ------------------------------------------------------------------ row ops (rowops.hip)
// m_bound: rows the chunk's buffers hold; expect_tokens >= 0: what the caller's host_lengths promised.  A mask that
// yields more tokens than m_bound is truncated there (cu[n] clamped to m_bound - 1 for a news START, cu[n_news] to m_bound:
// every row index cu[n] + t, t < len, and every [CLS] row cu[n] stays below m_bound) and, like any disagreement with
// expect_tokens, raises MANNER_HIP_STATUS_LENGTHS.
int lengths_and_offsets(const int64_t* mask, int64_t n_news, int64_t padded_len, int32_t* lens,
                        int32_t* cu /*[n_news+1]*/, int32_t* m_total /*[2]: tokens, news*/, int64_t m_bound,
                        int64_t expect_tokens, int32_t* status, hipStream_t stream);
int embed_layernorm(DType out, const int64_t* ids, int64_t n_news, int64_t padded_len, const int32_t* cu,
                    const float* word, const float* pos, const float* type0, const float* gamma,
                    const float* beta, int H, float eps, int pos_offset, int vocab, int max_pos,
                    void* x, int32_t* status, hipStream_t stream);
int layernorm_rows(DType out, const float* pre, const float* gamma, const float* beta, int H, float eps,
                   void* x, int64_t m_bound, const int* m_total, hipStream_t stream);
// ---- deferred LayerNorm helpers (bf16 path)
// raw[cu[n]+t] = bf16(word + type + pos) and mr = {mean, rstd} of the rounded row (K1 without the normalisation)
int embed_raw(DType dt, const int64_t* ids, int64_t n_news, int64_t padded_len, const int32_t* cu, const float* word,
              const float* pos, const float* type0, int H, float eps, int pos_offset, int vocab, int max_pos,
              void* raw, void* mr, int32_t* status, hipStream_t stream);
// {mean, rstd} of a row from its (<= 16, zero-padded) partial {sum, sum of squares} in ascending group order: ONE definition for
// dln_finalize_kernel and for the in-launch finalize of the producing GEMM (gemm.hip nres_fan_in), with the one contraction the
// compiler had chosen for the kernel written out (var = fma(sum2, 1/H, -(mean * mean))), so that both give the same bits
__device__ __forceinline__ float2 dln_row_stats(const float2 (&v)[16], float inv_h, float eps) {
  float s1 = v[0].x, s2 = v[0].y;
#pragma unroll
  for (int g = 1; g < 16; ++g) { s1 += v[g].x; s2 += v[g].y; }
  const float mean = s1 * inv_h;
  const float var = fmaxf(fmaf(s2, inv_h, -(mean * mean)), 0.f);
  return float2{mean, 1.0f / sqrtf(var + eps)};
}
// mr[m] = {mean, rstd} from the `groups` partial sums part[g][m] of row m (fixed summation order)
int dln_finalize(const void* part, int groups, int H, float eps, void* mr, int64_t m_bound, const int* m_total,
                 hipStream_t stream);
// dst[n] = bf16(LN(raw[cu[n]])) — the [CLS] rows of the last layer, normalised on the way
int gather_cls_ln(DType dt, const void* raw, const void* mr, const int32_t* cu, int64_t n_news, int H, const float* gamma,
                  const float* beta, void* dst, hipStream_t stream);
// pack time: wf[n,k] = bf16(gamma[k] w[n,k]); c1[n] = sum_k wf[n,k]; c2[n] = bias[n] + sum_k beta[k] w[n,k]
int fold_layernorm(DType dt, const float* w, const float* bias, const float* gamma, const float* beta, int N, int K, void* wf,
                   float* c1, float* c2, hipStream_t stream);
// out[n, t, :] = x[cu[n]+t] (normalised with mr/gamma/beta when mr != NULL) for t < len(n), zeros for padded positions
int scatter_hidden(DType in, DType out_dt, const void* x, const void* mr, const int32_t* cu, int64_t n_news, int64_t lp, int H,
                   const float* gamma, const float* beta, void* out, hipStream_t stream);
// bf16x3 / f16x3: out [rows, 3K] of the 16-bit type `dt` = [hi | hi | lo] of x [rows, K] f32 (activations) or [hi | lo | hi] (weights); rows >= *m_total skipped
int split3_rows(DType dt, bool weight, const float* x, void* out, int K, int64_t rows, const int* m_total, hipStream_t stream);
// x3 modes: LayerNorm that also emits the split operand of the GEMM that consumes its output: x f32 [m, H] and a3 [m, 3H] = [hi | hi | lo]
int layernorm_rows_split(DType split_dt, const float* pre, const float* gamma, const float* beta, int H, float eps, float* x, void* a3,
                         int64_t m_bound, const int* m_total, hipStream_t stream);
int add_vectors(const float* a, const float* b, float* out, int n, hipStream_t stream);
int set_device_int(int32_t* p, int32_t value, hipStream_t stream);
int gather_cls(DType in, const void* x, const int32_t* cu, int64_t n_news, int H, float* out, hipStream_t stream);
int gather_cls_rows(DType dt, const void* x, const int32_t* cu, int64_t n_news, int H, void* dst, hipStream_t stream);
int convert_f32_to_16(DType dt, const float* src, void* dst, int64_t n, hipStream_t stream);

// ------------------------------------------------------------------ attention (attention.hip)
// ctx[tok, head*64 + d] = softmax_k(q.k/8) v over the tokens of the same news; qkv [m, 3H] = [Q|K|V].
// split_out != NULL (dt == DT_F32, MFMA kernel only): the rows are written as the split operand [hi | hi | lo] of the 16-bit type
// split_dt, row stride 3H, instead of as f32 ctx rows (x3 modes).
int attention_varlen(DType dt, const void* qkv, void* ctx, const int32_t* cu, int64_t n_news, int heads,
                     int H, int max_len, hipStream_t stream, void* split_out = nullptr, DType split_dt = DT_F32);

// last layer: only the [CLS] query of every news attends.  qcls [n_news, H]; kv [m, 2H] = [K|V];
// ctx_cls [n_news, H].
int attention_cls(DType dt, const void* qcls, const void* kv, void* ctx_cls, const int32_t* cu, int64_t n_news,
                  int heads, int H, hipStream_t stream);

}  // namespace manner
