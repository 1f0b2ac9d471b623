// HBM-bound row kernels of the encoder: sequence lengths + packing offsets, embedding gather +
// LayerNorm (K1), LayerNorm (tail of K4/K6), CLS gather (K7), f32 -> bf16 weight conversion.
// One 64-lane wave per row, 16-byte accesses, two-pass mean/variance in registers, shuffle
// reductions (no LDS).
#include "common.h"

namespace manner {
namespace {

constexpr int MAX_H = 1024;          // 4 x float4 per lane
constexpr int VEC_PER_LANE = MAX_H / 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- lengths: one wave per news; validates that mask is a 0/1 prefix with 1 <= len <= MAX_LEN
__global__ __launch_bounds__(256) void lengths_kernel(const int64_t* __restrict__ mask, int64_t n_news,
                                                      int64_t lp, int32_t* __restrict__ lens,
                                                      int32_t* __restrict__ status) {
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= n_news) return;
  const int64_t* row = mask + n * lp;
  int cnt = 0, bad = 0, last_one = -1;
  for (int64_t j = lane; j < lp; j += 64) {
    const int64_t v = row[j];
    if (v != 0) { cnt++; last_one = (int)j; if (v != 1) bad = 1; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    bad |= __shfl_xor(bad, o, 64);
    last_one = max(last_one, __shfl_xor(last_one, o, 64));
  }
  if (last_one + 1 != cnt) bad = 1;                      // a hole: not a prefix mask
  if (cnt < 1 || cnt > MANNER_HIP_MAX_LEN) bad = 1;
  if (lane == 0) {
    lens[n] = bad ? 1 : cnt;                             // keep downstream indexing in bounds
    if (bad) atomicOr(status, 1);
  }
}

// ---- exclusive scan of lens -> cu[0..n], m_total; single workgroup (n is at most a chunk of news)
__global__ __launch_bounds__(1024) void scan_kernel(const int32_t* __restrict__ lens, int64_t n,
                                                    int32_t* __restrict__ cu, int32_t* __restrict__ m_total,
                                                    int m_bound, int expect_tokens, int32_t* __restrict__ status) {
  __shared__ int part[1024];
  const int t = threadIdx.x;
  const int64_t per = (n + 1023) / 1024;
  const int64_t lo = t * per, hi = min(n, lo + per);
  int s = 0;
  for (int64_t i = lo; i < hi; ++i) s += lens[i];
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    int v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  // a news never STARTS at row m_bound: when the mask holds more tokens than the buffers (a wrong host_lengths / token_bound —
  // STATUS_LENGTHS below), the news past the bound collapse onto the last row, so the [CLS] gathers / scatters that index
  // row cu[n] stay inside the [m_bound, *] buffers (they used to touch the first row of the NEXT buffer)
  for (int64_t i = lo; i < hi; ++i) { cu[i] = min(run, m_bound - 1); run += lens[i]; }
  if (t == 1023) {
    const int total = part[1023];
    cu[n] = min(total, m_bound);
    m_total[0] = min(total, m_bound);
    m_total[1] = (int32_t)n;
    if (total > m_bound || (expect_tokens >= 0 && total != expect_tokens)) atomicOr(status, MANNER_HIP_STATUS_LENGTHS);
  }
}

template <typename TOut>
__device__ __forceinline__ void store_vec(TOut* p, const f32x4& v);
template <>
__device__ __forceinline__ void store_vec<float>(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
template <>
__device__ __forceinline__ void store_vec<bf16_t>(bf16_t* p, const f32x4& v) {
  *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
}
template <>
__device__ __forceinline__ void store_vec<f16_t>(f16_t* p, const f32x4& v) {
  *reinterpret_cast<f16x4*>(p) = f16x4{(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3]};
}

// normalise the H values a wave holds in v[] (lane owns columns (i*64+lane)*4 .. +3) and store
template <typename TOut>
__device__ __forceinline__ void wave_layernorm_store(f32x4 v[VEC_PER_LANE], int H, float eps,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, TOut* __restrict__ dst,
                                                     int lane) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VEC_PER_LANE; ++i)
    if ((i * 64 + lane) * 4 < H) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VEC_PER_LANE; ++i)
    if ((i * 64 + lane) * 4 < H) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
#pragma unroll
  for (int i = 0; i < VEC_PER_LANE; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
      const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      store_vec<TOut>(dst + c, o);
      v[i] = o;                                          // the normalised row stays in registers for callers that go on with it
    }
  }
}
// the x3 modes' split operand of a row held as in wave_layernorm_store: a3[0 .. 3H) = [hi | hi | lo], hi = TE(x), lo = TE(x - hi)
template <typename TE>
__device__ __forceinline__ void wave_split3_store(const f32x4 v[VEC_PER_LANE], int H, TE* __restrict__ a3, int lane) {
#pragma unroll
  for (int i = 0; i < VEC_PER_LANE; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) {
      typename E16<TE>::v4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { hi[e] = (TE)v[i][e]; lo[e] = (TE)(v[i][e] - (float)hi[e]); }
      *reinterpret_cast<typename E16<TE>::v4*>(a3 + c) = hi;
      *reinterpret_cast<typename E16<TE>::v4*>(a3 + H + c) = hi;
      *reinterpret_cast<typename E16<TE>::v4*>(a3 + 2 * H + c) = lo;
    }
  }
}

// ---- K1: x[cu[n]+t] = LN(word[ids[n,t]] + type[0] + pos[t + pos_offset])
// (transformers/models/bert/modeling_bert.py:68-108; RoBERTa positions start at pad_id+1,
// transformers/models/roberta/modeling_roberta.py:142-155, for right-padded inputs.)
template <typename TOut>
__global__ __launch_bounds__(256) void embed_ln_kernel(
    const int64_t* __restrict__ ids, int64_t lp, const int32_t* __restrict__ cu,
    const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ type0,
    const float* __restrict__ gamma, const float* __restrict__ beta, int H, float eps, int pos_offset,
    int vocab, int max_pos, TOut* __restrict__ x, int32_t* __restrict__ status) {
  const int64_t n = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int start = cu[n], len = cu[n + 1] - start;
  for (int t = wave; t < len; t += 4) {
    int64_t id = ids[n * lp + t];
    int p = t + pos_offset;
    if (id < 0 || id >= vocab || p >= max_pos) {       // would be an IndexError in the reference
      if (lane == 0) atomicOr(status, 2);
      id = 0; p = 0;
    }
    const float* wr = word + (size_t)id * H;
    const float* pr = pos + (size_t)p * H;
    f32x4 v[VEC_PER_LANE];
#pragma unroll
    for (int i = 0; i < VEC_PER_LANE; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(wr + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(type0 + c);
        const f32x4 d = *reinterpret_cast<const f32x4*>(pr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = (a[e] + b[e]) + d[e];   // HF order: (word + type) + pos
      }
    }
    wave_layernorm_store<TOut>(v, H, eps, gamma, beta, x + (size_t)(start + t) * H, lane);
  }
}

// ---- LayerNorm over rows of the f32 pre-activation buffer
template <typename TOut>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ pre,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int H, float eps,
                                                        TOut* __restrict__ x, const int* __restrict__ m_total) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= *m_total) return;
  const float* src = pre + (size_t)row * H;
  f32x4 v[VEC_PER_LANE];
#pragma unroll
  for (int i = 0; i < VEC_PER_LANE; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) v[i] = *reinterpret_cast<const f32x4*>(src + c);
  }
  wave_layernorm_store<TOut>(v, H, eps, gamma, beta, x + (size_t)row * H, lane);
}

// x3 modes: LayerNorm + the split operand of the GEMM that consumes it, in one pass over the row
template <typename TE>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* __restrict__ pre, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int H, float eps, float* __restrict__ x,
                                                              TE* __restrict__ a3, const int* __restrict__ m_total) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= *m_total) return;
  const float* src = pre + (size_t)row * H;
  f32x4 v[VEC_PER_LANE];
#pragma unroll
  for (int i = 0; i < VEC_PER_LANE; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < H) v[i] = *reinterpret_cast<const f32x4*>(src + c);
  }
  wave_layernorm_store<float>(v, H, eps, gamma, beta, x + (size_t)row * H, lane);
  wave_split3_store<TE>(v, H, a3 + (size_t)row * 3 * H, lane);
}

// ---- deferred LayerNorm (16-bit path): K1 without the normalisation.  The row is rounded to the element type first and
// the statistics are those of the rounded values — exactly what the consuming GEMM reads.
template <typename TE>
__global__ __launch_bounds__(256) void embed_raw_kernel(
    const int64_t* __restrict__ ids, int64_t lp, const int32_t* __restrict__ cu, const float* __restrict__ word,
    const float* __restrict__ pos, const float* __restrict__ type0, int H, float eps, int pos_offset, int vocab,
    int max_pos, TE* __restrict__ raw, float2* __restrict__ mr, int32_t* __restrict__ status) {
  // grid (news, 4): token t of a news goes to wave t % 16 of its four workgroups, so a wave walks at most 8 tokens
  // (each a chain of dependent loads: id -> row address -> row) instead of 32
  const int64_t n = blockIdx.x;
  const int wave = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int start = cu[n], len = cu[n + 1] - start;
  for (int t = wave; t < len; t += 16) {
    int64_t id = ids[n * lp + t];
    int p = t + pos_offset;
    if (id < 0 || id >= vocab || p >= max_pos) {
      if (lane == 0) atomicOr(status, 2);
      id = 0; p = 0;
    }
    const float* wr = word + (size_t)id * H;
    const float* pr = pos + (size_t)p * H;
    f32x4 v[VEC_PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC_PER_LANE; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c < H) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(wr + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(type0 + c);
        const f32x4 d = *reinterpret_cast<const f32x4*>(pr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = (float)(TE)((a[e] + b[e]) + d[e]);
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        store_vec<TE>(raw + (size_t)(start + t) * H + c, v[i]);
      }
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VEC_PER_LANE; ++i)
      if ((i * 64 + lane) * 4 < H) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
      }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
    if (lane == 0) mr[start + t] = float2{mean, rstd};
  }
}

__global__ __launch_bounds__(256) void dln_finalize_kernel(const float2* __restrict__ part, int groups, int64_t stride, float inv_h,
                                                           float eps, float2* __restrict__ mr, const int* __restrict__ m_total) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= *m_total) return;
  // all loads of a row in flight at once (groups = H / 64 <= 16 on this path), summed in ascending group order: with
  // one load per loop trip the kernel was 12 serialised L2 latencies long
  float2 v[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) v[g] = g < groups ? part[(size_t)g * stride + m] : float2{0.f, 0.f};
  mr[m] = dln_row_stats(v, inv_h, eps);
}

template <typename TE>
__global__ __launch_bounds__(256) void gather_cls_ln_kernel(const TE* __restrict__ raw, const float2* __restrict__ mr,
                                                            const int32_t* __restrict__ cu, int H, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, TE* __restrict__ dst) {
  const int64_t n = blockIdx.x;
  const int64_t row = cu[n];
  const float2 ms = mr[row];
  const TE* src = raw + (size_t)row * H;
  for (int c = threadIdx.x; c < H; c += 256) dst[n * H + c] = (TE)(((float)src[c] - ms.x) * ms.y * gamma[c] + beta[c]);
}

__global__ __launch_bounds__(256) void add_vec_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}

// one wave per output row n of the folded weight
template <typename TE>
__global__ __launch_bounds__(256) void fold_ln_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, int N, int K,
                                                      TE* __restrict__ wf, float* __restrict__ c1, float* __restrict__ c2) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float x = w[(size_t)n * K + k];
    const TE f = (TE)(gamma[k] * x);
    wf[(size_t)n * K + k] = f;
    s1 += (float)f;
    s2 = fmaf(beta[k], x, s2);
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { c1[n] = s1; c2[n] = bias[n] + s2; }
}

// ---- hidden states back in the reference's padded layout: out[n, t, :] = (LN of) x[cu[n] + t] for t < len, 0 beyond
template <typename TIn, typename TOut, bool NORM>
__global__ __launch_bounds__(256) void scatter_hidden_kernel(const TIn* __restrict__ x, const float2* __restrict__ mr,
                                                             const int32_t* __restrict__ cu, int64_t lp, int H,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             TOut* __restrict__ out) {
  const int64_t n = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int start = cu[n], len = cu[n + 1] - start;
  for (int t = wave; t < lp; t += 4) {
    TOut* dst = out + ((size_t)n * lp + t) * H;
    if (t >= len) {
      for (int c = lane * 4; c < H; c += 256) store_vec<TOut>(dst + c, f32x4{0.f, 0.f, 0.f, 0.f});
      continue;
    }
    const TIn* src = x + (size_t)(start + t) * H;
    float2 ms = {0.f, 1.f};
    if (NORM) ms = mr[start + t];
    for (int c = lane * 4; c < H; c += 256) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (float)src[c + e];
      if (NORM) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] - ms.x) * ms.y * g[e] + b[e];
      }
      store_vec<TOut>(dst + c, v);
    }
  }
}

// ---- bf16x3 operand splits.  Activations: out[m] = [hi(x[m]) | hi(x[m]) | lo(x[m])], weights: [hi | lo | hi] with
// hi = bf16(v), lo = bf16(v - hi): the depth-3K dot product of the two is hi.hi + hi.lo + lo.hi.
template <typename TE, bool WEIGHT>
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, TE* __restrict__ out, int K,
                                                     int64_t rows, const int* __restrict__ m_total) {
  const int64_t row = blockIdx.x;
  if (row >= rows || (m_total && row >= *m_total)) return;
  const float* src = x + (size_t)row * K;
  TE* dst = out + (size_t)row * 3 * K;
  for (int c = threadIdx.x * 4; c < K; c += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
    typename E16<TE>::v4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { hi[e] = (TE)v[e]; lo[e] = (TE)(v[e] - (float)hi[e]); }
    *reinterpret_cast<typename E16<TE>::v4*>(dst + c) = hi;
    *reinterpret_cast<typename E16<TE>::v4*>(dst + K + c) = WEIGHT ? lo : hi;
    *reinterpret_cast<typename E16<TE>::v4*>(dst + 2 * K + c) = WEIGHT ? hi : lo;
  }
}

// ---- K7: out[n] = x[cu[n]] (the [CLS] row), as f32
template <typename TIn>
__global__ __launch_bounds__(256) void gather_cls_kernel(const TIn* __restrict__ x, const int32_t* __restrict__ cu,
                                                         int64_t n_news, int H, float* __restrict__ out) {
  const int64_t n = blockIdx.x;
  const TIn* src = x + (size_t)cu[n] * H;
  for (int c = threadIdx.x; c < H; c += 256) out[n * H + c] = (float)src[c];
}

// compact copy of the [CLS] rows, same dtype: dst[n] = x[cu[n]]
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ x, const int32_t* __restrict__ cu,
                                                          int H, T* __restrict__ dst) {
  const int64_t n = blockIdx.x;
  const f32x4* src = reinterpret_cast<const f32x4*>(x + (size_t)cu[n] * H);
  f32x4* d = reinterpret_cast<f32x4*>(dst + (size_t)n * H);
  for (int c = threadIdx.x; c < H * (int)sizeof(T) / 16; c += 256) d[c] = src[c];
}

template <typename TE>
__global__ __launch_bounds__(256) void cvt_16_kernel(const float* __restrict__ src, TE* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = (TE)src[i];
}

}  // namespace

int lengths_and_offsets(const int64_t* mask, int64_t n_news, int64_t padded_len, int32_t* lens, int32_t* cu,
                        int32_t* m_total, int64_t m_bound, int64_t expect_tokens, int32_t* status, hipStream_t stream) {
  hipLaunchKernelGGL(lengths_kernel, dim3((unsigned)((n_news + 3) / 4)), dim3(256), 0, stream, mask, n_news,
                     padded_len, lens, status);
  MANNER_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, lens, n_news, cu, m_total,
                     (int)min(m_bound, (int64_t)0x7fffffff), (int)expect_tokens, status);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int embed_layernorm(DType out, const int64_t* ids, int64_t n_news, int64_t padded_len, const int32_t* cu,
                    const float* word, const float* pos, const float* type0, const float* gamma,
                    const float* beta, int H, float eps, int pos_offset, int vocab, int max_pos, void* x,
                    int32_t* status, hipStream_t stream) {
  if (H % 4 || H > MAX_H) return fail(MANNER_HIP_E_INVALID, "hidden size %d unsupported (<= %d, %%4)", H, MAX_H);
  dim3 g((unsigned)n_news), b(256);
  if (out == DT_BF16)
    hipLaunchKernelGGL(embed_ln_kernel<bf16_t>, g, b, 0, stream, ids, padded_len, cu, word, pos, type0, gamma, beta,
                       H, eps, pos_offset, vocab, max_pos, static_cast<bf16_t*>(x), status);
  else if (out == DT_F16)
    hipLaunchKernelGGL(embed_ln_kernel<f16_t>, g, b, 0, stream, ids, padded_len, cu, word, pos, type0, gamma, beta,
                       H, eps, pos_offset, vocab, max_pos, static_cast<f16_t*>(x), status);
  else
    hipLaunchKernelGGL(embed_ln_kernel<float>, g, b, 0, stream, ids, padded_len, cu, word, pos, type0, gamma, beta,
                       H, eps, pos_offset, vocab, max_pos, static_cast<float*>(x), status);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int layernorm_rows(DType out, const float* pre, const float* gamma, const float* beta, int H, float eps, void* x,
                   int64_t m_bound, const int* m_total, hipStream_t stream) {
  if (H % 4 || H > MAX_H) return fail(MANNER_HIP_E_INVALID, "hidden size %d unsupported", H);
  dim3 g((unsigned)((m_bound + 3) / 4)), b(256);
  if (out == DT_BF16)
    hipLaunchKernelGGL(layernorm_kernel<bf16_t>, g, b, 0, stream, pre, gamma, beta, H, eps, static_cast<bf16_t*>(x), m_total);
  else if (out == DT_F16)
    hipLaunchKernelGGL(layernorm_kernel<f16_t>, g, b, 0, stream, pre, gamma, beta, H, eps, static_cast<f16_t*>(x), m_total);
  else
    hipLaunchKernelGGL(layernorm_kernel<float>, g, b, 0, stream, pre, gamma, beta, H, eps, static_cast<float*>(x), m_total);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int layernorm_rows_split(DType split_dt, const float* pre, const float* gamma, const float* beta, int H, float eps, float* x, void* a3,
                         int64_t m_bound, const int* m_total, hipStream_t stream) {
  if (H % 4 || H > MAX_H || !is_16bit(split_dt)) return fail(MANNER_HIP_E_INVALID, "layernorm_rows_split: hidden size %d / dtype unsupported", H);
  dim3 g((unsigned)((m_bound + 3) / 4)), b(256);
  if (split_dt == DT_F16) hipLaunchKernelGGL(layernorm_split_kernel<f16_t>, g, b, 0, stream, pre, gamma, beta, H, eps, x, static_cast<f16_t*>(a3), m_total);
  else hipLaunchKernelGGL(layernorm_split_kernel<bf16_t>, g, b, 0, stream, pre, gamma, beta, H, eps, x, static_cast<bf16_t*>(a3), m_total);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int embed_raw(DType dt, const int64_t* ids, int64_t n_news, int64_t padded_len, const int32_t* cu, const float* word,
              const float* pos, const float* type0, int H, float eps, int pos_offset, int vocab, int max_pos,
              void* raw, void* mr, int32_t* status, hipStream_t stream) {
  if (H % 4 || H > MAX_H) return fail(MANNER_HIP_E_INVALID, "hidden size %d unsupported (<= %d, %%4)", H, MAX_H);
  if (dt == DT_F16)
    hipLaunchKernelGGL(embed_raw_kernel<f16_t>, dim3((unsigned)n_news, 4), dim3(256), 0, stream, ids, padded_len, cu, word, pos, type0, H,
                       eps, pos_offset, vocab, max_pos, static_cast<f16_t*>(raw), static_cast<float2*>(mr), status);
  else
    hipLaunchKernelGGL(embed_raw_kernel<bf16_t>, dim3((unsigned)n_news, 4), dim3(256), 0, stream, ids, padded_len, cu, word, pos, type0, H,
                       eps, pos_offset, vocab, max_pos, static_cast<bf16_t*>(raw), static_cast<float2*>(mr), status);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int dln_finalize(const void* part, int groups, int H, float eps, void* mr, int64_t m_bound, const int* m_total,
                 hipStream_t stream) {
  hipLaunchKernelGGL(dln_finalize_kernel, dim3((unsigned)((m_bound + 255) / 256)), dim3(256), 0, stream,
                     static_cast<const float2*>(part), groups, m_bound, 1.0f / (float)H, eps, static_cast<float2*>(mr), m_total);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int gather_cls_ln(DType dt, const void* raw, const void* mr, const int32_t* cu, int64_t n_news, int H, const float* gamma,
                  const float* beta, void* dst, hipStream_t stream) {
  if (dt == DT_F16)
    hipLaunchKernelGGL(gather_cls_ln_kernel<f16_t>, dim3((unsigned)n_news), dim3(256), 0, stream, static_cast<const f16_t*>(raw),
                       static_cast<const float2*>(mr), cu, H, gamma, beta, static_cast<f16_t*>(dst));
  else
    hipLaunchKernelGGL(gather_cls_ln_kernel<bf16_t>, dim3((unsigned)n_news), dim3(256), 0, stream, static_cast<const bf16_t*>(raw),
                       static_cast<const float2*>(mr), cu, H, gamma, beta, static_cast<bf16_t*>(dst));
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int scatter_hidden(DType in, DType out_dt, const void* x, const void* mr, const int32_t* cu, int64_t n_news, int64_t lp, int H,
                   const float* gamma, const float* beta, void* out, hipStream_t stream) {
  if (H % 4) return fail(MANNER_HIP_E_INVALID, "hidden size %d unsupported", H);
  dim3 g((unsigned)n_news), b(256);
  const float2* m = static_cast<const float2*>(mr);
#define SCATTER(TI, TO, NORM) hipLaunchKernelGGL((scatter_hidden_kernel<TI, TO, NORM>), g, b, 0, stream, static_cast<const TI*>(x), m, cu, lp, H, gamma, beta, static_cast<TO*>(out))
  if (in == DT_BF16) {
    if (mr) { if (out_dt == DT_F32) SCATTER(bf16_t, float, true); else SCATTER(bf16_t, bf16_t, true); }
    else { if (out_dt == DT_F32) SCATTER(bf16_t, float, false); else SCATTER(bf16_t, bf16_t, false); }
  } else if (in == DT_F16) {
    if (mr) { if (out_dt == DT_F32) SCATTER(f16_t, float, true); else SCATTER(f16_t, bf16_t, true); }
    else { if (out_dt == DT_F32) SCATTER(f16_t, float, false); else SCATTER(f16_t, bf16_t, false); }
  } else {
    if (out_dt == DT_F32) SCATTER(float, float, false); else SCATTER(float, bf16_t, false);
  }
#undef SCATTER
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int split3_rows(DType dt, bool weight, const float* x, void* out, int K, int64_t rows, const int* m_total, hipStream_t stream) {
  if (K % 4) return fail(MANNER_HIP_E_INVALID, "split3: K=%d", K);
  const dim3 g((unsigned)rows), b(256);
  if (dt == DT_F16) {
    if (weight) hipLaunchKernelGGL((split3_kernel<f16_t, true>), g, b, 0, stream, x, static_cast<f16_t*>(out), K, rows, m_total);
    else hipLaunchKernelGGL((split3_kernel<f16_t, false>), g, b, 0, stream, x, static_cast<f16_t*>(out), K, rows, m_total);
  } else {
    if (weight) hipLaunchKernelGGL((split3_kernel<bf16_t, true>), g, b, 0, stream, x, static_cast<bf16_t*>(out), K, rows, m_total);
    else hipLaunchKernelGGL((split3_kernel<bf16_t, false>), g, b, 0, stream, x, static_cast<bf16_t*>(out), K, rows, m_total);
  }
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

__global__ void set_int_kernel(int32_t* p, int32_t v) { *p = v; }

int set_device_int(int32_t* p, int32_t value, hipStream_t stream) {
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, stream, p, value);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int add_vectors(const float* a, const float* b, float* out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(add_vec_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, b, out, n);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int fold_layernorm(DType dt, const float* w, const float* bias, const float* gamma, const float* beta, int N, int K, void* wf,
                   float* c1, float* c2, hipStream_t stream) {
  if (dt == DT_F16)
    hipLaunchKernelGGL(fold_ln_kernel<f16_t>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, w, bias, gamma, beta, N, K,
                       static_cast<f16_t*>(wf), c1, c2);
  else
    hipLaunchKernelGGL(fold_ln_kernel<bf16_t>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, stream, w, bias, gamma, beta, N, K,
                       static_cast<bf16_t*>(wf), c1, c2);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int gather_cls(DType in, const void* x, const int32_t* cu, int64_t n_news, int H, float* out, hipStream_t stream) {
  dim3 g((unsigned)n_news), b(256);
  if (in == DT_BF16)
    hipLaunchKernelGGL(gather_cls_kernel<bf16_t>, g, b, 0, stream, static_cast<const bf16_t*>(x), cu, n_news, H, out);
  else if (in == DT_F16)
    hipLaunchKernelGGL(gather_cls_kernel<f16_t>, g, b, 0, stream, static_cast<const f16_t*>(x), cu, n_news, H, out);
  else
    hipLaunchKernelGGL(gather_cls_kernel<float>, g, b, 0, stream, static_cast<const float*>(x), cu, n_news, H, out);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int gather_cls_rows(DType dt, const void* x, const int32_t* cu, int64_t n_news, int H, void* dst, hipStream_t stream) {
  dim3 g((unsigned)n_news), b(256);
  if (is_16bit(dt))      // a 2-byte row copy: one instantiation serves bf16 and f16
    hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, g, b, 0, stream, static_cast<const bf16_t*>(x), cu, H, static_cast<bf16_t*>(dst));
  else
    hipLaunchKernelGGL(gather_rows_kernel<float>, g, b, 0, stream, static_cast<const float*>(x), cu, H, static_cast<float*>(dst));
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int convert_f32_to_16(DType dt, const float* src, void* dst, int64_t n, hipStream_t stream) {
  const int64_t blocks = (n + 255) / 256;
  const dim3 g((unsigned)(blocks < 4096 ? blocks : 4096));
  if (dt == DT_F16) hipLaunchKernelGGL(cvt_16_kernel<f16_t>, g, dim3(256), 0, stream, src, static_cast<f16_t*>(dst), n);
  else hipLaunchKernelGGL(cvt_16_kernel<bf16_t>, g, dim3(256), 0, stream, src, static_cast<bf16_t*>(dst), n);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

}  // namespace manner
