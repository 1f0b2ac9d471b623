// Shared declarations of the gfx950 MANNeR hot-path library (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "manner_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
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

// ------------------------------------------------------------------ GEMM (gemm.hip)
enum Epilogue { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RES = 2, EPI_BIAS_RES_LN = 3 };
enum DType { DT_F32 = 0, DT_BF16 = 1 };

// Y[m, n] = epi( sum_k X[m, k] * W[n, k] + bias[n] )  for m < *m_total (device scalar).
// X [m_bound, K], W [N, K] row-major of dtype `in`; Y [m_bound, N] of dtype `out`;
// residual [m_bound, N] of dtype `in` (EPI_BIAS_RES only).  m_bound is a multiple of 128 and all
// buffers hold that many rows; N % 128 == 0; K*sizeof(in) % 128 == 0.
int gemm_tn(DType in, DType out, Epilogue epi, const void* X, const void* W, const float* bias,
            const void* residual, void* Y, int64_t m_bound, int N, int K, const int* m_total,
            hipStream_t stream);

// bf16 only: Y = LayerNorm(X W^T + bias + Y) * gamma + beta, in place on the residual stream Y [m_bound, N]
// (K4 / K6 in one launch).  The N/256 column-tile workgroups of a 256-row panel exchange row sums
// through `sync`: m_bound/256 arrival counters (zeroed here per call), then per row and column tile
// one {sum, sum of squares} f32 slot (deterministic: summed in tile order, no float atomics).  N % 256 == 0.  `status` |= 4 if a bounded wait expires.
size_t gemm_ln_sync_bytes(int64_t m_bound, int N);
int gemm_tn_ln(const void* X, const void* W, const float* bias, void* Y, const float* gamma, const float* beta,
               float eps, void* sync, int64_t m_bound, int N, int K, const int* m_total, int32_t* status,
               hipStream_t stream);

// ------------------------------------------------------------------ row ops (rowops.hip)
int lengths_and_offsets(const int64_t* mask, int64_t n_news, int64_t padded_len, int32_t* lens,
                        int32_t* cu /*[n_news+1]*/, int32_t* m_total /*[2]: tokens, news*/, int32_t* status, hipStream_t stream);
int embed_layernorm(DType out, const int64_t* ids, int64_t n_news, int64_t padded_len, const int32_t* cu,
                    const float* word, const float* pos, const float* type0, const float* gamma,
                    const float* beta, int H, float eps, int pos_offset, int vocab, int max_pos,
                    void* x, int32_t* status, hipStream_t stream);
int layernorm_rows(DType out, const float* pre, const float* gamma, const float* beta, int H, float eps,
                   void* x, int64_t m_bound, const int* m_total, hipStream_t stream);
int gather_cls(DType in, const void* x, const int32_t* cu, int64_t n_news, int H, float* out, hipStream_t stream);
int gather_cls_rows(DType dt, const void* x, const int32_t* cu, int64_t n_news, int H, void* dst, hipStream_t stream);
int convert_f32_to_bf16(const float* src, bf16_t* dst, int64_t n, hipStream_t stream);

// ------------------------------------------------------------------ attention (attention.hip)
// ctx[tok, head*64 + d] = softmax_k(q.k/8) v over the tokens of the same news; qkv [m, 3H] = [Q|K|V].
int attention_varlen(DType dt, const void* qkv, void* ctx, const int32_t* cu, int64_t n_news, int heads,
                     int H, int max_len, hipStream_t stream);

// last layer: only the [CLS] query of every news attends.  qcls [n_news, H]; kv [m, 2H] = [K|V];
// ctx_cls [n_news, H].
int attention_cls(DType dt, const void* qcls, const void* kv, void* ctx_cls, const int32_t* cu, int64_t n_news,
                  int heads, int H, hipStream_t stream);

}  // namespace manner
