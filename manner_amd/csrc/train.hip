// the TRAINING path of MannerTextEncoder — forward in train() mode (HF BertModel/RobertaModel with its three
// dropouts per layer, reference manner/models/components/news_encoder.py:24-37; the model_step that drives it is
// manner/models/cr_module.py:140-171, a_module.py:102-108) and the backward pass down to the parameter gradients.
//
// Layout: the same packed-token layout as the inference engine (row m = cu[n] + t for the real tokens of news n; padded
// positions never influence a [CLS] output or a gradient, so they are not computed).  Activations and gradients are
// f32; the twelve GEMMs of a layer (4 forward, 4 data-gradient, 4 weight-gradient) go through gemm.hip's MFMA kernels
// — on f32 operands in the "fp32" mode (what the gradient-parity tests use) or on f16 / bf16 copies of the operands with
// f32 accumulation and f32 outputs ("16-mixed", the reference's trainer precision, configs/trainer/default.yaml:12).
// Everything else is an HBM-bound elementwise / row-reduction kernel written here.  Dropout masks come from a
// counter-based generator keyed by (seed, site, element index): the backward pass regenerates them instead of storing
// them, and manner_hip_dropout_mask hands the very same bits to the tests' oracle.
//
// Parameter gradients are produced for the tensors whose entry in the `grads` table is non-NULL; activation gradients
// are propagated exactly as far down as some requested gradient needs them (to the embeddings when an embedding
// tensor is trainable — the reference freezes "layer.N." parameters only, news_encoder.py:24-27, so by default the
// backward pass runs through the frozen layers into the embedding tables — or to `start_layer` when the caller
// supplies cached hidden states of a frozen prefix).
#include <math.h>
#include <stdlib.h>
#include <mutex>
#include <unordered_map>

#include "train_common.h"

namespace manner {
namespace {

// optional 16-bit copy of an activation for the next GEMM (saves that GEMM's separate f32 -> 16-bit pass)
struct Out16 {
  void* p;
  int dt;            // 0: none, DT_BF16, DT_F16
};
__device__ __forceinline__ void put16(const Out16& o, size_t idx, float v) {
  if (o.dt == DT_F16) static_cast<f16_t*>(o.p)[idx] = (f16_t)v;
  else if (o.dt == DT_BF16) static_cast<bf16_t*>(o.p)[idx] = (bf16_t)v;
}

// eight consecutive elements at once (idx a multiple of 8): one 16-byte store instead of eight 2-byte ones
__device__ __forceinline__ void put16x8(const Out16& o, size_t idx, const float* v) {
  if (o.dt == DT_F16) {
    f16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (f16_t)v[e];
    *reinterpret_cast<f16x8*>(static_cast<f16_t*>(o.p) + idx) = r;
  } else if (o.dt == DT_BF16) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x8*>(static_cast<bf16_t*>(o.p) + idx) = r;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------ embeddings
// e[cu[n]+t] = word[id] + pos[t + pos_offset] + type[0]            (modeling_bert.py:68-108 without LN / dropout)
// klen != NULL ("full rows": cu[n] = n * lp, every position is a row, klen[n] = real tokens): padded positions embed
// their pad token at position t (BERT) or at position pad_id (RoBERTa: cumsum(mask) * mask + pad_id, pad_pos >= 0).
__global__ __launch_bounds__(256) void embed_sum_kernel(const int64_t* __restrict__ ids, int64_t n_news, int lp,
                                                        const int32_t* __restrict__ cu, const float* __restrict__ word,
                                                        const float* __restrict__ pos, const float* __restrict__ type0, int H,
                                                        int pos_offset, int vocab, int max_pos, float* __restrict__ e,
                                                        int32_t* __restrict__ status, const int32_t* __restrict__ klen, int pad_pos) {
  const int64_t idx = blockIdx.x;
  const int64_t n = idx / lp;
  const int t = (int)(idx - n * lp);
  const int len = cu[n + 1] - cu[n];
  if (t >= len) return;
  int64_t id = ids[n * lp + t];
  int p = t + pos_offset;
  if (klen && pad_pos >= 0 && t >= klen[n]) p = pad_pos;
  if (id < 0 || id >= vocab) { if (threadIdx.x == 0 && status) atomicOr(status, MANNER_HIP_STATUS_TOKEN); id = 0; }
  if (p >= max_pos) { if (threadIdx.x == 0 && status) atomicOr(status, MANNER_HIP_STATUS_TOKEN); p = max_pos - 1; }
  float* dst = e + (size_t)(cu[n] + t) * H;
  const float* w = word + (size_t)id * H;
  const float* ps = pos + (size_t)p * H;
  for (int c = threadIdx.x; c < H; c += 256) dst[c] = w[c] + ps[c] + type0[c];
}

// d word[id] += de[m], d pos[p] += de[m]   (f32 atomics: rows repeat across news; the token-type row is a column sum)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, int64_t n_news, int lp,
                                                        const int32_t* __restrict__ cu, const float* __restrict__ de, int H,
                                                        int pos_offset, int vocab, int max_pos, float* __restrict__ dword,
                                                        float* __restrict__ dpos, const int32_t* __restrict__ klen, int pad_pos,
                                                        int word_pad, int pos_pad) {
  const int64_t idx = blockIdx.x;
  const int64_t n = idx / lp;
  const int t = (int)(idx - n * lp);
  const int len = cu[n + 1] - cu[n];
  if (t >= len) return;
  int64_t id = ids[n * lp + t];
  int p = t + pos_offset;
  if (klen && pad_pos >= 0 && t >= klen[n]) p = pad_pos;      // "full rows": padded positions as embed_sum_kernel places them
  if (id < 0 || id >= vocab) id = 0;
  if (p >= max_pos) p = max_pos - 1;
  const float* src = de + (size_t)(cu[n] + t) * H;
  // HF builds word_embeddings (and RoBERTa's position_embeddings) with padding_idx = pad_token_id: nn.Embedding never gives that
  // row a gradient (modeling_bert.py:71, modeling_roberta.py:75-80) — the pad token embedded at the padded positions of the
  // "full rows" path contributes to the forward only
  const bool wg = dword && id != word_pad, pg = dpos && p != pos_pad;
  for (int c = threadIdx.x; c < H; c += 256) {
    const float g = src[c];
    if (wg) atomicAdd(dword + (size_t)id * H + c, g);
    if (pg) atomicAdd(dpos + (size_t)p * H + c, g);
  }
}

// x[cu[n]+t] = hidden[n, t]  (cached frozen-prefix activations, [N, Lp, H] f32 -> packed rows) and its inverse for the
// gradient of the prefix
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ padded, float* __restrict__ packed,
                                                        int64_t n_news, int lp, const int32_t* __restrict__ cu, int H,
                                                        int to_padded) {
  const int64_t idx = blockIdx.x;
  const int64_t n = idx / lp;
  const int t = (int)(idx - n * lp);
  const int len = cu[n + 1] - cu[n];
  if (to_padded) {
    float* dst = const_cast<float*>(padded) + (size_t)idx * H;
    const float* src = packed + (size_t)(cu[n] + t) * H;
    for (int c = threadIdx.x; c < H; c += 256) dst[c] = t < len ? src[c] : 0.f;
  } else if (t < len) {
    const float* src = padded + (size_t)idx * H;
    float* dst = packed + (size_t)(cu[n] + t) * H;
    for (int c = threadIdx.x; c < H; c += 256) dst[c] = src[c];
  }
}

// ------------------------------------------------------------------------------------------------ LayerNorm
// one wave per row; H <= 64 * LN_MAX
constexpr int LN_MAX = 16;

// y = dropout( LN(x) ), stats = {mean, rstd}
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int H, float eps, float* __restrict__ y,
                                                     float2* __restrict__ stats, const int* __restrict__ m_total, Drop drop, Out16 o16) {
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= m_total[0]) return;
  const float* row = x + (size_t)m * H;
  float v[LN_MAX];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < H ? row[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX; ++i) {
    const int c = lane + 64 * i;
    const float d = c < H ? v[i] - mean : 0.f;
    q = fmaf(d, d, q);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);      // biased variance, as F.layer_norm
  if (lane == 0) stats[m] = float2{mean, rstd};
  float* out = y + (size_t)m * H;
#pragma unroll
  for (int i = 0; i < LN_MAX; ++i) {
    const int c = lane + 64 * i;
    if (c < H) {
      const float r = drop.apply(fmaf((v[i] - mean) * rstd, gamma[c], beta[c]), (uint64_t)m * H + c);
      out[c] = r;
      put16(o16, (size_t)m * H + c, r);
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  per-block partial sums of dgamma = dy * xhat and
// dbeta = dy over the rows this block visits (deterministic second stage: reduce_partials_kernel).
// `dy` may alias `dx`.
constexpr int LN_BWD_BLOCKS = 1024;
constexpr int LN_V4 = LN_MAX / 4;                     // float4 pieces per lane: column 4 (lane + 64 i) .. + 3
// d16 (16-bit modes): also the 16-bit copy of dropout(dx) — the gradient of the GEMM output that sat under this LayerNorm's
// residual add (r = dropout(y) + x: d y = dropout(d r)); the dropout bits are those of element rowmap[m] * H + c (rowmap:
// compact [CLS] rows of the last layer) — saves the separate dropout pass over d r
template <int NV4>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* dy, const float* __restrict__ x,
                                                     const float2* __restrict__ stats, const float* __restrict__ gamma, int H,
                                                     float* dx, float* __restrict__ pgamma, float* __restrict__ pbeta,
                                                     const int* __restrict__ m_total, Drop drop, Out16 d16,
                                                     const int32_t* __restrict__ rowmap, int pstride) {
  __shared__ __attribute__((aligned(16))) float red[4][64 * LN_MAX + 4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t M = m_total[0];
  const int nv = H / 4;                               // float4 pieces per row (H % 4 == 0: check_cfg asks for H % 128 == 0)
  f32x4 ag[NV4], ab[NV4], gm[NV4];
#pragma unroll
  for (int i = 0; i < NV4; ++i) {
    ag[i] = ab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int c = lane + 64 * i;
    gm[i] = c < nv ? reinterpret_cast<const f32x4*>(gamma)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // the wave's NEXT row is requested before this row's two wave reductions and stores (round 4: the loads of a row used to start
  // only after the previous row had been written — 3.9 TB/s on the 15 k-row launches)
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t m = (int64_t)blockIdx.x * 4 + wave;
  f32x4 nd[NV4], nx[NV4];
  float2 nst = float2{0.f, 0.f};
  auto request = [&](int64_t r) {
    nst = stats[r];
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)r * H);
    const f32x4* gr = reinterpret_cast<const f32x4*>(dy + (size_t)r * H);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      const int c = lane + 64 * i;
      if (c < nv) { nd[i] = gr[c]; nx[i] = xr[c]; }
    }
  };
  if (m < M) request(m);
  for (; m < M; m += stride) {
    const float2 st = nst;
    f32x4 dcur[NV4], xcur[NV4];
#pragma unroll
    for (int i = 0; i < NV4; ++i) { dcur[i] = nd[i]; xcur[i] = nx[i]; }
    if (m + stride < M) request(m + stride);
    f32x4 xh[NV4], g[NV4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      const int c = lane + 64 * i;
      const bool ok = c < nv;
      const f32x4 d = ok ? dcur[i] : f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4 xv = ok ? xcur[i] : f32x4{st.x, st.x, st.x, st.x};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[i][e] = (xv[e] - st.x) * st.y;
        g[i][e] = d[e] * gm[i][e];
        ag[i][e] = fmaf(d[e], xh[i][e], ag[i][e]);
        ab[i][e] += d[e];
        s1 += g[i][e];
        s2 = fmaf(g[i][e], xh[i][e], s2);
      }
    }
    const float c1 = wave_sum(s1) / (float)H, c2 = wave_sum(s2) / (float)H;
    f32x4* out = reinterpret_cast<f32x4*>(dx + (size_t)m * H);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
      const int c = lane + 64 * i;
      if (c < nv) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = st.y * (g[i][e] - c1 - xh[i][e] * c2);
        out[c] = o;
        if (d16.dt) {
          const uint64_t di = (uint64_t)(rowmap ? rowmap[m] : m) * H + 4 * c;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = drop.apply(o[e], di + e);
          if (d16.dt == DT_F16) reinterpret_cast<f16x4*>(static_cast<f16_t*>(d16.p) + (size_t)m * H)[c] = f16x4{(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3]};
          else reinterpret_cast<bf16x4*>(static_cast<bf16_t*>(d16.p) + (size_t)m * H)[c] = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
      }
    }
  }
  if (!pgamma) return;
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV4; ++i) *reinterpret_cast<f32x4*>(&red[wave][4 * (lane + 64 * i)]) = pass ? ab[i] : ag[i];
    __syncthreads();
    float* dst = (pass ? pbeta : pgamma) + (size_t)blockIdx.x * pstride;      // row `block` of the partials: [d gamma | d beta] side by side
    for (int c = threadIdx.x; c < H; c += 256) dst[c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
  }
}

// Destination of a column sum: one vector, or (seg > 0) three vectors of seg columns each — the Q | K | V gradients of the packed
// projection land in the three parameter gradients directly (a NULL part is not wanted and skipped)
struct SumDst {
  float* p[3];
  int seg;
  __device__ __forceinline__ void put(int c, float v) const {
    if (seg <= 0) { p[0][c] = v; return; }
    const int k = c / seg;
    if (p[k]) p[k][c - k * seg] = v;
  }
};

// out[c] = sum_b partial[b][c]   (fixed order)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int blocks, int width, SumDst out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= width) return;
  float s[4] = {0.f, 0.f, 0.f, 0.f};                  // four loads in flight per thread; the order of the sum stays fixed
  int b = 0;
  for (; b + 4 <= blocks; b += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += partial[(size_t)(b + u) * width + c];
  }
  for (; b < blocks; ++b) s[0] += partial[(size_t)b * width + c];
  out.put(c, (s[0] + s[1]) + (s[2] + s[3]));
}

// The same sum for NARROW outputs (bias / LayerNorm gradients: width <= a few thousand, hundreds of partial rows): the kernel above
// would run 3 - 12 workgroups, each thread walking all partial rows 4 at a time — ~40 us of dependent round trips per call, 100
// calls per training step.  Here a workgroup owns 64 columns and splits the partial rows over 16 groups (8 loads in flight each);
// the 16 group sums meet in LDS and are added in a fixed order.
__global__ __launch_bounds__(1024) void reduce_partials_narrow_kernel(const float* __restrict__ partial, int blocks, int width, SumDst out) {
  __shared__ float red[16][64];
  const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + col;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < width) {
    int b = g;
    for (; b + 7 * 16 < blocks; b += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += partial[(size_t)(b + 16 * u) * width + c];
    }
    for (; b < blocks; b += 16) s[0] += partial[(size_t)b * width + c];
  }
  red[g][col] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (g == 0 && c < width) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][col];
    out.put(c, t);
  }
}
void reduce_partials(hipStream_t s, const float* partial, int blocks, int64_t width, const SumDst& out) {
  if (width <= 16384 && blocks >= 32)
    hipLaunchKernelGGL(reduce_partials_narrow_kernel, dim3((unsigned)((width + 63) / 64)), dim3(1024), 0, s, partial, blocks, (int)width, out);
  else
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)((width + 255) / 256)), dim3(256), 0, s, partial, blocks, (int)width, out);
}
void reduce_partials(hipStream_t s, const float* partial, int blocks, int64_t width, float* out) {
  reduce_partials(s, partial, blocks, width, SumDst{{out, nullptr, nullptr}, 0});
}

// partial[b][c] = sum over the rows of block b of y[m][c]   (bias gradients, token-type embedding gradient)
// grid (ceil(width / 256), COLSUM_BLOCKS): block (x, b) sums 256 columns over the rows b, b + COLSUM_BLOCKS, ...
constexpr int COLSUM_BLOCKS = 256;   // (64 left every thread a chain of ~240 dependent-latency row reads at 15 k tokens)
template <typename TS>
__global__ __launch_bounds__(256) void colsum_kernel(const TS* __restrict__ y, int width, float* __restrict__ partial,
                                                     const int* __restrict__ m_total) {
  const int64_t M = m_total[0];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= width) return;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  int64_t m = blockIdx.y;
  for (; m + 3 * COLSUM_BLOCKS < M; m += 4 * COLSUM_BLOCKS) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += (float)y[(size_t)(m + u * COLSUM_BLOCKS) * width + c];
  }
  for (; m < M; m += COLSUM_BLOCKS) s[0] += (float)y[(size_t)m * width + c];
  partial[(size_t)blockIdx.y * width + c] = (s[0] + s[1]) + (s[2] + s[3]);
}

// ------------------------------------------------------------------------------------------------ elementwise
// out = dropout(a) [+ res]; rows >= *m_total untouched
// rowmap != NULL: row r of these (compact) tensors stands for row rowmap[r] of the full tensor — the dropout bits are those
// of the full tensor's element, so the [CLS]-only last layer draws exactly the masks the all-rows computation would
__global__ __launch_bounds__(256) void dropout_add_kernel(const float* a, const float* res, float* out, int width,
                                                          const int* __restrict__ m_total, Drop drop, Out16 o16,
                                                          const int32_t* __restrict__ rowmap) {
  const int64_t total = (int64_t)m_total[0] * width;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    uint64_t di = (uint64_t)i;
    if (rowmap) { const int64_t r = i / width; di = (uint64_t)rowmap[r] * width + (uint64_t)(i - r * width); }
    float v = drop.apply(a[i], di);
    if (res) v += res[i];
    out[i] = v;
    put16(o16, (size_t)i, v);
  }
}

__global__ __launch_bounds__(256) void add_rows_kernel(const float* a, const float* b, float* out, int width,
                                                       const int* __restrict__ m_total) {
  const int64_t total = (int64_t)m_total[0] * width;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) out[i] = a[i] + b[i];
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * expf(-0.5f * x * x);
}
// mode 0: out = gelu(a); mode 1: out = b * gelu'(a)
__global__ __launch_bounds__(256) void gelu_kernel(const float* a, const float* b, float* out, int width,
                                                   const int* __restrict__ m_total, int mode, Out16 o16) {
  const int64_t total = (int64_t)m_total[0] * width;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const float v = mode ? b[i] * gelu_grad(a[i]) : gelu_exact(a[i]);
    out[i] = v;
    put16(o16, (size_t)i, v);
  }
}

// 16-bit modes: the two I-wide tensors live in the 16-bit type only (what the GEMMs that consume them read anyway):
//   forward  g16 = gelu(inter)                         (inter stays f32: the backward's gelu' is exact)
//   backward dz16 <- dz16 * gelu'(inter)               (dz16 = d g as the data-gradient GEMM wrote it, in place)
template <typename TE>
__global__ __launch_bounds__(256) void gelu16_kernel(const float* __restrict__ a, TE* __restrict__ z, int width, const int* __restrict__ m_total,
                                                     int mode) {
  typedef TE v4 __attribute__((ext_vector_type(4)));
  const int64_t total4 = (int64_t)m_total[0] * width / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const f32x4 x = reinterpret_cast<const f32x4*>(a)[i];
    v4 o;
    if (mode) {
      const v4 d = reinterpret_cast<const v4*>(z)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (TE)((float)d[e] * gelu_grad(x[e]));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (TE)gelu_exact(x[e]);
    }
    reinterpret_cast<v4*>(z)[i] = o;
  }
}

// out[n] = dropout(x[cu[n]])  ([CLS] rows; news_encoder.py:34-35) — and the scatter of its gradient
__global__ __launch_bounds__(256) void cls_kernel(const float* __restrict__ x, const int32_t* __restrict__ cu, int H,
                                                  float* __restrict__ out, Drop drop) {
  const int64_t n = blockIdx.x;
  const float* src = x + (size_t)cu[n] * H;
  for (int c = threadIdx.x; c < H; c += 256) out[(size_t)n * H + c] = drop.apply(src[c], (uint64_t)n * H + c);
}
__global__ __launch_bounds__(256) void cls_bwd_kernel(const float* __restrict__ g, const int32_t* __restrict__ cu, int H,
                                                      float* __restrict__ dx, Drop drop) {
  const int64_t n = blockIdx.x;
  float* dst = dx + (size_t)cu[n] * H;
  for (int c = threadIdx.x; c < H; c += 256) dst[c] = drop.apply(g[(size_t)n * H + c], (uint64_t)n * H + c);
}

// dst[n] = src[cu[n]] ([CLS] rows -> compact) and dst[cu[n]] += src[n] (compact -> [CLS] rows)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ cu, int H,
                                                          float* __restrict__ dst) {
  const int64_t n = blockIdx.x;
  for (int c = threadIdx.x; c < H; c += 256) dst[(size_t)n * H + c] = src[(size_t)cu[n] * H + c];
}
// the same gather from a tensor that exists in the 16-bit type only (round 5: ctx with 16-bit saved activations); exact widening
template <typename TE>
__global__ __launch_bounds__(256) void gather_rows16_kernel(const TE* __restrict__ src, const int32_t* __restrict__ cu, int H,
                                                            float* __restrict__ dst) {
  const int64_t n = blockIdx.x;
  for (int c = threadIdx.x; c < H; c += 256) dst[(size_t)n * H + c] = (float)src[(size_t)cu[n] * H + c];
}
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ cu, int H,
                                                               float* __restrict__ dst) {
  const int64_t n = blockIdx.x;
  for (int c = threadIdx.x; c < H; c += 256) dst[(size_t)cu[n] * H + c] += src[(size_t)n * H + c];
}

// out[c][r] = T(in[r][c]) for r < rows_valid, 0 for rows_valid <= r < rows_out  (operands of the gradient GEMMs)
// `ks` < rows_out: slice-major output for the split weight-gradient GEMMs — slice s = rows [s * ks, (s + 1) * ks) is its own
// contiguous [cols, ks] matrix at offset s * cols * ks.
template <typename TS, typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const TS* __restrict__ in, int64_t cols, T* __restrict__ out,
                                                        int64_t rows_out, const int* __restrict__ rows_valid_dev,
                                                        int64_t rows_valid_host, int64_t ks) {
  __shared__ float tile[32][33];
  const int64_t rv = rows_valid_dev ? (int64_t)rows_valid_dev[0] : rows_valid_host;
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < rv && c < cols) ? (float)in[(size_t)r * cols + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < cols && r < rows_out) {
      const int64_t sl = r / ks, rr = r - sl * ks;
      out[((size_t)sl * cols + c) * ks + rr] = (T)tile[tx][ty + 8 * k];
    }
  }
}

// ------------------------------------------------------------------------------------------------ attention (train)
// qkv [m, 3H] = [Q | K | V].  A workgroup of AT threads serves one news and HPB heads.  Two adjacent lanes own one row
// (query row in the forward / backward-q kernels, key row in backward-kv) and split its 64 dims: 32 each, so the row
// state (q, o / q, dctx, dq / k, v, dk, dv) stays in 128 arch VGPRs — the first version kept a whole row per lane and
// spent more instructions moving accumulators through AGPRs than on FMAs — and dot products are completed with one
// lane exchange.  RPH = AT / (2 HPB) rows per head: titles (<= 32 tokens) fill a 64-lane workgroup with one head, <= 16
// tokens with two.  The other side's rows go through LDS in chunks of KC = 32 / HPB per head; a row is stored as two
// 36-word halves (16-byte aligned for ds_read_b128; the two half-lanes and the HPB head groups read different banks).
// P = softmax(q k^T / 8), Pd = dropout(P) (modeling_bert.py:128-140: dropout on the probabilities), ctx = Pd v.  The
// forward keeps {row max, row sum} so the backward rebuilds P without a reduction pass.
constexpr int AD = 64, AH = 32, RS = 72;              // head dim, dims per lane, LDS words per staged row (2 x (32 + 4))
template <int HPB> struct AttnGeom {
  static constexpr int KC = 32 / HPB;                 // rows per chunk and head
  static constexpr int HS = KC * RS + 8;              // words between two heads' chunks
};

template <int AT, int HPB>
__device__ __forceinline__ void stage_rows(float* dst, const float* __restrict__ src, size_t ld, int col0, int row0, int S, float scale) {
  // dst[hs][r][half][d'] = src[(row0 + r) * ld + col0 + hs * AD + 32 * half + d'] * scale for r < KC (0 beyond S)
  constexpr int KC = AttnGeom<HPB>::KC, HS = AttnGeom<HPB>::HS;
  for (int e = threadIdx.x; e < HPB * KC * AD; e += AT) {
    const int hs = e / (KC * AD), rem = e - hs * (KC * AD), r = rem / AD, d = rem - r * AD;
    dst[hs * HS + r * RS + (d >> 5) * (AH + 4) + (d & 31)] = row0 + r < S ? src[(size_t)(row0 + r) * ld + col0 + hs * AD + d] * scale : 0.f;
  }
}
__device__ __forceinline__ float pair_sum(float v) { return v + __shfl_xor(v, 1, 64); }

// klen != NULL: the rows of news n are queries, only its first klen[n] rows are keys (HF's additive key mask on the
// padded positions, which still produce outputs of their own: manner_hip_encode_full).
template <int AT, int HPB>
__global__ __launch_bounds__(AT) void attn_train_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                            float2* __restrict__ ml, const int32_t* __restrict__ cu, int heads,
                                                            int H, Drop drop, const int32_t* __restrict__ klen, Out16 o16) {
  constexpr int KC = AttnGeom<HPB>::KC, HS = AttnGeom<HPB>::HS, RPH = AT / (2 * HPB);
  __shared__ __attribute__((aligned(16))) float ks[HPB * HS], vs[HPB * HS];
  const int hs = threadIdx.x / (2 * RPH), i = (threadIdx.x - hs * 2 * RPH) >> 1, half = threadIdx.x & 1;
  const int h0 = blockIdx.x * HPB, h = h0 + hs;
  const int64_t n = blockIdx.y;
  const int base = cu[n], S = cu[n + 1] - base;
  const bool active = i < S;
  const int SK = klen ? min(klen[n], S) : S;
  const size_t ld = (size_t)3 * H;
  const float* rows = qkv + (size_t)base * ld;
  float q[AH], o[AH];
#pragma unroll
  for (int d = 0; d < AH; ++d) { q[d] = active ? rows[(size_t)i * ld + h * AD + AH * half + d] * 0.125f : 0.f; o[d] = 0.f; }
  float mx = -INFINITY, l = 0.f;
  const float* kh = ks + hs * HS + half * (AH + 4);
  const float* vh = vs + hs * HS + half * (AH + 4);
  for (int j0 = 0; j0 < SK; j0 += KC) {
    __syncthreads();
    stage_rows<AT, HPB>(ks, rows, ld, H + h0 * AD, j0, SK, 1.f);
    stage_rows<AT, HPB>(vs, rows, ld, 2 * H + h0 * AD, j0, SK, 1.f);
    __syncthreads();
    const int cnt = min(KC, SK - j0);
    for (int j = 0; j < cnt; ++j) {                          // (inactive lanes run along: the lane exchange needs both halves)
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < AH; ++d) s = fmaf(q[d], kh[j * RS + d], s);
      s = pair_sum(s);
      if (s > mx) {
        const float f = expf(mx - s);
        l *= f;
#pragma unroll
        for (int d = 0; d < AH; ++d) o[d] *= f;
        mx = s;
      }
      const float p = expf(s - mx);
      l += p;
      const float pd = drop.apply(p, ((uint64_t)(base + i) * heads + h) * 256 + (uint64_t)(j0 + j));
#pragma unroll
      for (int d = 0; d < AH; ++d) o[d] = fmaf(pd, vh[j * RS + d], o[d]);
    }
  }
  if (!active) return;
  const float inv = 1.f / l;
  const size_t c0 = (size_t)(base + i) * H + h * AD + AH * half;
#pragma unroll
  for (int d = 0; d < AH; ++d) { o[d] *= inv; ctx[c0 + d] = o[d]; }
#pragma unroll
  for (int d = 0; d < AH; d += 8) put16x8(o16, c0 + d, o + d);
  if (ml && half == 0) ml[(size_t)(base + i) * heads + h] = float2{mx, l};
}

// query-row owner: D_i = sum_j dP_ij P_ij = dctx_i . ctx_i (ctx = Pd v carries the same dropout), then
// dq_i = sum_j P_ij (dP_ij - D_i) k_j / 8
template <int AT, int HPB>
__global__ __launch_bounds__(AT) void attn_train_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx,
                                                              const float* __restrict__ ctx, const float2* __restrict__ ml,
                                                              float* __restrict__ dqkv, float* __restrict__ dsum,
                                                              const int32_t* __restrict__ cu, int heads, int H, Drop drop, Out16 o16,
                                                              const int32_t* __restrict__ klen) {
  constexpr int KC = AttnGeom<HPB>::KC, HS = AttnGeom<HPB>::HS, RPH = AT / (2 * HPB);
  __shared__ __attribute__((aligned(16))) float ks[HPB * HS], vs[HPB * HS];
  const int hs = threadIdx.x / (2 * RPH), i = (threadIdx.x - hs * 2 * RPH) >> 1, half = threadIdx.x & 1;
  const int h0 = blockIdx.x * HPB, h = h0 + hs;
  const int64_t n = blockIdx.y;
  const int base = cu[n], SQ_ = cu[n + 1] - base;
  const int S = klen ? min(klen[n], SQ_) : SQ_;             // keys; every row of the news is a query
  const bool active = i < SQ_;
  const size_t ld = (size_t)3 * H;
  const float* rows = qkv + (size_t)base * ld;
  float q[AH], go[AH], dq[AH];
  float D = 0.f;
#pragma unroll
  for (int d = 0; d < AH; ++d) {
    q[d] = active ? rows[(size_t)i * ld + h * AD + AH * half + d] * 0.125f : 0.f;
    go[d] = active ? dctx[(size_t)(base + i) * H + h * AD + AH * half + d] : 0.f;
    dq[d] = 0.f;
    D = fmaf(go[d], active ? ctx[(size_t)(base + i) * H + h * AD + AH * half + d] : 0.f, D);
  }
  D = pair_sum(D);
  const float2 st = active ? ml[(size_t)(base + i) * heads + h] : float2{0.f, 1.f};
  const float inv = 1.f / st.y;
  const float* kh = ks + hs * HS + half * (AH + 4);
  const float* vh = vs + hs * HS + half * (AH + 4);
  for (int j0 = 0; j0 < S; j0 += KC) {
    __syncthreads();
    stage_rows<AT, HPB>(ks, rows, ld, H + h0 * AD, j0, S, 1.f);
    stage_rows<AT, HPB>(vs, rows, ld, 2 * H + h0 * AD, j0, S, 1.f);
    __syncthreads();
    const int cnt = min(KC, S - j0);
    for (int j = 0; j < cnt; ++j) {
      float s = 0.f, gv = 0.f;
#pragma unroll
      for (int d = 0; d < AH; ++d) { s = fmaf(q[d], kh[j * RS + d], s); gv = fmaf(go[d], vh[j * RS + d], gv); }
      s = pair_sum(s);
      gv = pair_sum(gv);
      const float p = expf(s - st.x) * inv;
      const float dp = drop.apply(gv, ((uint64_t)(base + i) * heads + h) * 256 + (uint64_t)(j0 + j));
      const float ds = p * (dp - D) * 0.125f;
#pragma unroll
      for (int d = 0; d < AH; ++d) dq[d] = fmaf(ds, kh[j * RS + d], dq[d]);
    }
  }
  if (!active) return;
  const size_t c0 = (size_t)(base + i) * ld + h * AD + AH * half;
#pragma unroll
  for (int d = 0; d < AH; ++d) dqkv[c0 + d] = dq[d];
#pragma unroll
  for (int d = 0; d < AH; d += 8) put16x8(o16, c0 + d, dq + d);
  if (half == 0) dsum[(size_t)(base + i) * heads + h] = D;
}

// key-row owner: dk_j = sum_i dS_ij q_i / 8, dv_j = sum_i Pd_ij dctx_i
template <int AT, int HPB>
__global__ __launch_bounds__(AT) void attn_train_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dctx,
                                                               const float2* __restrict__ ml, const float* __restrict__ dsum,
                                                               float* __restrict__ dqkv, const int32_t* __restrict__ cu,
                                                               int heads, int H, Drop drop, Out16 o16, const int32_t* __restrict__ klen) {
  constexpr int KC = AttnGeom<HPB>::KC, HS = AttnGeom<HPB>::HS, RPH = AT / (2 * HPB);
  __shared__ __attribute__((aligned(16))) float qs[HPB * HS], gs[HPB * HS];
  __shared__ float sm[HPB][KC], sl[HPB][KC], sd[HPB][KC];
  const int hs = threadIdx.x / (2 * RPH), j = (threadIdx.x - hs * 2 * RPH) >> 1, half = threadIdx.x & 1;
  const int h0 = blockIdx.x * HPB, h = h0 + hs;
  const int64_t n = blockIdx.y;
  const int base = cu[n], S = cu[n + 1] - base;             // queries: every row; a row past klen is no key: its d k = d v = 0
  const bool active = j < S;
  const bool is_key = !klen || j < klen[n];
  const size_t ld = (size_t)3 * H;
  const float* rows = qkv + (size_t)base * ld;
  float k[AH], v[AH], dk[AH], dv[AH];
#pragma unroll
  for (int d = 0; d < AH; ++d) {
    k[d] = active ? rows[(size_t)j * ld + H + h * AD + AH * half + d] : 0.f;
    v[d] = active ? rows[(size_t)j * ld + 2 * H + h * AD + AH * half + d] : 0.f;
    dk[d] = dv[d] = 0.f;
  }
  const float* qh = qs + hs * HS + half * (AH + 4);
  const float* gh = gs + hs * HS + half * (AH + 4);
  for (int i0 = 0; i0 < S; i0 += KC) {
    __syncthreads();
    stage_rows<AT, HPB>(qs, rows, ld, h0 * AD, i0, S, 0.125f);
    stage_rows<AT, HPB>(gs, dctx + (size_t)base * H, (size_t)H, h0 * AD, i0, S, 1.f);
    if (threadIdx.x < HPB * KC) {
      const int sh = threadIdx.x / KC, r = threadIdx.x - sh * KC;
      const bool ok = i0 + r < S;
      const float2 st = ok ? ml[(size_t)(base + i0 + r) * heads + h0 + sh] : float2{0.f, 1.f};
      sm[sh][r] = st.x;
      sl[sh][r] = 1.f / st.y;
      sd[sh][r] = ok ? dsum[(size_t)(base + i0 + r) * heads + h0 + sh] : 0.f;
    }
    __syncthreads();
    const int cnt = min(KC, S - i0);
    for (int i = 0; i < cnt; ++i) {
      float s = 0.f, gv = 0.f;
#pragma unroll
      for (int d = 0; d < AH; ++d) { s = fmaf(qh[i * RS + d], k[d], s); gv = fmaf(gh[i * RS + d], v[d], gv); }
      s = pair_sum(s);
      gv = pair_sum(gv);
      const float p = expf(s - sm[hs][i]) * sl[hs][i];
      const uint64_t idx = ((uint64_t)(base + i0 + i) * heads + h) * 256 + (uint64_t)j;
      const float dp = drop.apply(gv, idx);
      const float pd = drop.apply(p, idx);
      const float ds = p * (dp - sd[hs][i]);         // qs already carries the 1/8
#pragma unroll
      for (int d = 0; d < AH; ++d) { dk[d] = fmaf(ds, qh[i * RS + d], dk[d]); dv[d] = fmaf(pd, gh[i * RS + d], dv[d]); }
    }
  }
  if (!active) return;
  const size_t ck = (size_t)(base + j) * ld + H + h * AD + AH * half, cv = ck + H;
  if (!is_key) {
#pragma unroll
    for (int d = 0; d < AH; ++d) dk[d] = dv[d] = 0.f;
  }
#pragma unroll
  for (int d = 0; d < AH; ++d) { dqkv[ck + d] = dk[d]; dqkv[cv + d] = dv[d]; }
#pragma unroll
  for (int d = 0; d < AH; d += 8) { put16x8(o16, ck + d, dk + d); put16x8(o16, cv + d, dv + d); }
}

// launch geometry from the padded length (<= MANNER_HIP_MAX_LEN = 128): two lanes per row; rows per head = 16 / 32 / 64 / 128
#define MANNER_ATTN_DISPATCH(LP, HEADS, CALL)                   \
  do {                                                          \
    if ((LP) <= 16 && (HEADS) % 2 == 0) { CALL(64, 2); }        \
    else if ((LP) <= 32) { CALL(64, 1); }                       \
    else if ((LP) <= 64) { CALL(128, 1); }                      \
    else { CALL(256, 1); }                                      \
  } while (0)

// ------------------------------------------------------------------------------------------------ scorer + loss (train)
// CRModule.forward with late_fusion=True in training (cr_module.py:105-131): every occurrence is encoded, so the
// history / candidate vectors arrive in ragged order.  user_i = mean(hist rows of i); s_j = <user_i, cand_j>.
__global__ __launch_bounds__(256) void lf_train_fwd_kernel(const float* __restrict__ hist, const int64_t* __restrict__ hoff,
                                                           const float* __restrict__ cand, const int64_t* __restrict__ coff, int D,
                                                           float* __restrict__ user, float* __restrict__ scores) {
  extern __shared__ float us[];
  const int64_t b = blockIdx.x;
  const int64_t h0 = hoff[b], h1 = hoff[b + 1], c0 = coff[b], c1 = coff[b + 1];
  const float inv = 1.f / (float)(h1 - h0);                 // an empty history divides by zero, as torch.div does
  for (int c = threadIdx.x; c < D; c += 256) {
    float s = 0.f;
    for (int64_t r = h0; r < h1; ++r) s += hist[(size_t)r * D + c];
    us[c] = s * inv;
    user[(size_t)b * D + c] = us[c];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t j = c0 + wave; j < c1; j += 4) {
    float a = 0.f;
    for (int c = lane; c < D; c += 64) a = fmaf(us[c], cand[(size_t)j * D + c], a);
    a = wave_sum(a);
    if (lane == 0) scores[j] = a;
  }
}
// d cand_j = g_j user_i;  d user_i = sum_j g_j cand_j;  d hist_r = d user_i / h_i
__global__ __launch_bounds__(256) void lf_train_bwd_kernel(const float* __restrict__ g, const float* __restrict__ user,
                                                           const int64_t* __restrict__ hoff, const float* __restrict__ cand,
                                                           const int64_t* __restrict__ coff, int D, float* __restrict__ dhist,
                                                           float* __restrict__ dcand) {
  const int64_t b = blockIdx.x;
  const int64_t h0 = hoff[b], h1 = hoff[b + 1], c0 = coff[b], c1 = coff[b + 1];
  const float inv = 1.f / (float)(h1 - h0);
  for (int c = threadIdx.x; c < D; c += 256) {
    const float u = user[(size_t)b * D + c];
    float du = 0.f;
    for (int64_t j = c0; j < c1; ++j) {
      const float gj = g[j];
      du = fmaf(gj, cand[(size_t)j * D + c], du);
      dcand[(size_t)j * D + c] = gj * u;
    }
    du *= inv;
    for (int64_t r = h0; r < h1; ++r) dhist[(size_t)r * D + c] = du;
  }
}

// DotProduct backward (click_predictors.py:9-12: torch.bmm(user [B,1,D], cand [B,D,C])), strided cand / d cand
__global__ __launch_bounds__(256) void dot_bwd_kernel(const float* __restrict__ g, const float* __restrict__ user,
                                                      const float* __restrict__ cand, int64_t C, int D, int64_t sb, int64_t sd,
                                                      int64_t sc, float* __restrict__ duser, float* __restrict__ dcand) {
  const int64_t b = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += 256) {
    const float u = user[(size_t)b * D + d];
    float du = 0.f;
    for (int64_t c = 0; c < C; ++c) {
      const float gc = g[b * C + c];
      du = fmaf(gc, cand[b * sb + d * sd + c * sc], du);
      dcand[(b * D + d) * C + c] = gc * u;          // contiguous [B, D, C], the layout of the reference's permuted view
    }
    duser[(size_t)b * D + d] = du;
  }
}

// per-impression loss (as eval_loss_kernel in metrics.hip) and its UNSCALED gradient wrt the ragged scores:
// mode 0 SupCon (losses.py:12-40): l_i = -(sum_pos (v_j - lse)) / (n_pos + tiny), v = s / T (the detached row maximum
//   drops out); d l_i / d s_j = -(y_j - n_pos softmax_j) / (T (n_pos + tiny)) over the real candidates;
// mode 1 CrossEntropyLoss with probability targets over the dense zero-padded row: d l_i / d s_j = softmax_j sum(y) - y_j.
__global__ __launch_bounds__(256) void train_loss_kernel(const float* __restrict__ scores, const float* __restrict__ labels,
                                                         const int64_t* __restrict__ off, int64_t B, int mode, float inv_t,
                                                         int64_t c_max, float tiny, float* __restrict__ losses,
                                                         float* __restrict__ grad) {
#pragma clang fp contract(off)
  const int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= B) return;
  const int64_t c0 = off[i], c1 = off[i + 1];
  const int64_t pad = mode == 1 ? c_max - (c1 - c0) : 0;
  float mx = pad > 0 ? 0.f : -INFINITY;
  for (int64_t j = c0 + lane; j < c1; j += 64) mx = fmaxf(mx, scores[j] * inv_t);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float se = 0.f, sp = 0.f, np = 0.f;
  for (int64_t j = c0 + lane; j < c1; j += 64) {
    const float v = scores[j] * inv_t - mx;
    se += expf(v);
    const float y = labels[j];
    if (mode == 1) { sp = fmaf(y, v, sp); np += y; }
    else if (y > 0.5f) { sp += v; np += 1.f; }
  }
  for (int o = 32; o > 0; o >>= 1) { se += __shfl_xor(se, o, 64); sp += __shfl_xor(sp, o, 64); np += __shfl_xor(np, o, 64); }
  if (pad > 0) se += (float)pad * expf(-mx);
  const float lse = logf(se);
  const float loss = mode == 1 ? -(sp - np * lse) : -(sp - np * lse) / (np + tiny);
  if (lane == 0) losses[i] = loss;
  for (int64_t j = c0 + lane; j < c1; j += 64) {
    const float sm = expf(scores[j] * inv_t - mx - lse);
    const float y = labels[j];
    if (mode == 1) grad[j] = sm * np - y;
    else grad[j] = -((y > 0.5f ? 1.f : 0.f) - np * sm) * inv_t / (np + tiny);
  }
}
// batch value and the factor that turns the unscaled gradient into d L / d s: mode 0 mean over the NON-ZERO losses
// (pytorch_metric_learning's AvgNonZeroReducer, the default of SupConLoss; zero-loss impressions get no gradient),
// mode 1 mean over the batch.  One workgroup, fixed order.
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ losses, int64_t B, int mode,
                                                          float* __restrict__ out /*[2]: loss, scale*/) {
  __shared__ float ss[256], sn[256];
  float s = 0.f, n = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += 256) {
    const float l = losses[i];
    if (mode == 1 || l > 0.f) { s += l; n += 1.f; }
  }
  ss[threadIdx.x] = s; sn[threadIdx.x] = n;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { ss[threadIdx.x] += ss[threadIdx.x + o]; sn[threadIdx.x] += sn[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sn[0] > 0.f ? ss[0] / sn[0] : 0.f;
    out[1] = sn[0] > 0.f ? 1.f / sn[0] : 0.f;
  }
}
__global__ __launch_bounds__(256) void loss_scale_kernel(float* __restrict__ grad, const float* __restrict__ losses,
                                                         const int64_t* __restrict__ off, int64_t B, int mode,
                                                         const float* __restrict__ red) {
  const int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= B) return;
  const float f = (mode == 1 || losses[i] > 0.f) ? red[1] : 0.f;
  for (int64_t j = off[i] + lane; j < off[i + 1]; j += 64) grad[j] *= f;
}

// ---------------------------------------------------------------- A-Module loss (a_module.py:73-75,102-108)
// pytorch_metric_learning SupConLoss(temperature, distance=DotProductSimilarity(normalize_embeddings=False)) on the news
// embeddings E [N, D] and their aspect labels: mat = E E^T / T, positives of anchor i = same label (i itself excluded),
// negatives = other labels; per anchor l_i = -(sum_pos (m_ij - lse_{j != i} m_ij)) / (n_pos + tiny).
__global__ __launch_bounds__(256) void a_sim_kernel(const float* __restrict__ emb, int64_t N, int D, float inv_t, float* __restrict__ mat) {
  const int64_t i = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int64_t j = wave; j < N; j += 4) {
    float a = 0.f;
    for (int c = lane; c < D; c += 64) a = fmaf(emb[i * D + c], emb[j * D + c], a);
    a = wave_sum(a);
    if (lane == 0) mat[i * N + j] = a * inv_t;
  }
}
// one wave per anchor: loss and the unscaled d l_i / d mat_ij (written over mat; the diagonal gets 0); flags: any positive / any negative
__global__ __launch_bounds__(256) void a_loss_kernel(float* __restrict__ mat, const int64_t* __restrict__ labels, int64_t N, float tiny,
                                                     float* __restrict__ losses, int32_t* __restrict__ flags) {
#pragma clang fp contract(off)
  const int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= N) return;
  float* row = mat + i * N;
  const int64_t li = labels[i];
  // mat - mat.max(dim=1) (the row maximum INCLUDES the diagonal |E_i|^2 / T, which dominates every other entry for real
  // [CLS] vectors), then lmu.logsumexp over the kept entries j != i = torch.logsumexp: its own maximum m2 over the kept set
  // comes out first, so the sum never underflows to 0 however far the diagonal sits above the rest.
  float mx = -INFINITY;
  for (int64_t j = lane; j < N; j += 64) mx = fmaxf(mx, row[j]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float m2 = -INFINITY;
  for (int64_t j = lane; j < N; j += 64)
    if (j != i) m2 = fmaxf(m2, row[j] - mx);
  for (int o = 32; o > 0; o >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, o, 64));
  float se = 0.f, sp = 0.f, np = 0.f, nn = 0.f;
  for (int64_t j = lane; j < N; j += 64) {
    if (j == i) continue;
    const float v = row[j] - mx;
    se += expf(v - m2);
    if (labels[j] == li) { sp += v; np += 1.f; } else nn += 1.f;
  }
  se = wave_sum(se); sp = wave_sum(sp); np = wave_sum(np); nn = wave_sum(nn);
  const float lse = N > 1 ? m2 + logf(se) : 0.f;
  if (lane == 0) {
    losses[i] = -(sp - np * lse) / (np + tiny);
    if (np > 0.f) atomicOr(flags, 1);
    if (nn > 0.f) atomicOr(flags, 2);
  }
  for (int64_t j = lane; j < N; j += 64) {
    if (j == i) { row[j] = 0.f; continue; }
    const float sm = expf((row[j] - mx) - lse);
    row[j] = -((labels[j] == li ? 1.f : 0.f) - np * sm) / (np + tiny);
  }
}
// {loss, scale} as loss_reduce_kernel (mean over the losses > 0), zero when the batch has no positive pair or no negative pair
__global__ __launch_bounds__(256) void a_reduce_kernel(float* __restrict__ losses, int64_t N, const int32_t* __restrict__ flags,
                                                       float* __restrict__ out) {
  __shared__ float ss[256], sn[256];
  const bool live = flags[0] == 3;
  float s = 0.f, n = 0.f;
  for (int64_t i = threadIdx.x; i < N; i += 256) {
    if (!live) losses[i] = 0.f;
    const float l = losses[i];
    if (l > 0.f) { s += l; n += 1.f; }
  }
  ss[threadIdx.x] = s; sn[threadIdx.x] = n;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { ss[threadIdx.x] += ss[threadIdx.x + o]; sn[threadIdx.x] += sn[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sn[0] > 0.f ? ss[0] / sn[0] : 0.f;
    out[1] = sn[0] > 0.f ? 1.f / sn[0] : 0.f;
  }
}
// d L / d E_i = (scale / T) sum_j (g_ij [l_i > 0] + g_ji [l_j > 0]) E_j
__global__ __launch_bounds__(256) void a_grad_kernel(const float* __restrict__ g, const float* __restrict__ losses, const float* __restrict__ red,
                                                     const float* __restrict__ emb, int64_t N, int D, float inv_t, float* __restrict__ demb) {
  const int64_t i = blockIdx.x;
  const float f = red[1] * inv_t;
  const float li = losses[i] > 0.f ? 1.f : 0.f;
  for (int c = threadIdx.x; c < D; c += 256) {
    float a = 0.f;
    for (int64_t j = 0; j < N; ++j) {
      const float w = g[i * N + j] * li + g[j * N + i] * (losses[j] > 0.f ? 1.f : 0.f);
      a = fmaf(w, emb[j * D + c], a);
    }
    demb[i * D + c] = a * f;
  }
}

__global__ void mask_kernel(uint8_t* out, int64_t n, Drop drop) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = (drop.thr == 0 || drop_bits(drop.seed, drop.site, (uint64_t)i) >= drop.thr) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------ host side
struct Bump {
  char* base;
  size_t off = 0;
  explicit Bump(void* p) : base(static_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t n) {
    off = (size_t)round_up((int64_t)off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

struct LayerSaved {
  float *x_in, *qkv, *ctx, *r1, *h1, *inter, *g, *r2;
  float2 *st1, *st2, *ml;
};
struct Saved {
  int32_t *lens, *cu, *m_total;
  float* esum;      // start_layer == 0 only
  float2* st0;
  LayerSaved l[64];
};
// lean (round 5, Ctx::lean): Q | K | V, ctx, the FFN pre-activation and its gelu exist in the 16-bit type only — their slots of every
// layer but the last (whose compact [CLS] tail keeps f32 tensors) are half the size: 31 instead of 49 KB per token and layer (bert-base)
void plan_saved(Bump& b, Saved& s, const manner_hip_encoder_config& c, int64_t N, int64_t Mb, int start, bool lean = false) {
  const size_t H = c.hidden, I = c.intermediate;
  s.lens = b.take<int32_t>(N);
  s.cu = b.take<int32_t>(N + 1);
  s.m_total = b.take<int32_t>(4);
  s.esum = start == 0 ? b.take<float>(Mb * H) : nullptr;
  s.st0 = start == 0 ? b.take<float2>(Mb) : nullptr;
  for (int l = start; l < c.layers; ++l) {
    LayerSaved& L = s.l[l];
    const size_t half = (lean && l + 1 < c.layers) ? 2 : 1;           // 16-bit tensors in f32-typed slots (H, I even)
    L.x_in = b.take<float>(Mb * H);
    L.qkv = b.take<float>(Mb * 3 * H / half);
    L.ctx = b.take<float>(Mb * H / half);
    L.r1 = b.take<float>(Mb * H);
    L.h1 = b.take<float>(Mb * H);
    L.inter = b.take<float>(Mb * I / half);
    L.g = b.take<float>(Mb * I / half);
    L.r2 = b.take<float>(Mb * H);
    L.st1 = b.take<float2>(Mb);
    L.st2 = b.take<float2>(Mb);
    L.ml = b.take<float2>(Mb * c.heads);
  }
}

constexpr int WGRAD_MAX_SLICES = 64;
struct Work {
  void *a16, *b16;                 // operand copies (16-bit modes) / transposed operands (all modes): each max(I,3H) * max(Mb, H) elements of 4 bytes
  float *wcat, *bcat, *zero, *tmp; // [3H, H] concatenated Q|K|V weight, its bias, a zero bias, one [Mb, max(I,3H)] temporary
  float *dx, *dr, *dbig, *dqkv, *dsum, *dw, *part;
  float* dwp;                      // partial weight gradients of the split GEMMs: [slices][Nout, K]
  void *h16a, *h16b, *big16;       // 16-bit copies written by the producers: layer input / ctx or h1 / g or a data gradient
  int32_t* dims;                   // device ints holding row counts of the weight-gradient GEMMs
};
void plan_work(Bump& b, Work& w, const manner_hip_encoder_config& c, int64_t Mb) {
  const size_t H = c.hidden, I = c.intermediate;
  const size_t wide = I > 3 * H ? I : 3 * H;
  const size_t rows = ((size_t)Mb > wide ? (size_t)Mb : wide) + 64 * WGRAD_MAX_SLICES;     // + the split GEMMs' slice padding
  w.a16 = b.take<float>(wide * rows);
  w.b16 = b.take<float>(wide * rows);
  w.wcat = b.take<float>(3 * H * H);
  w.bcat = b.take<float>(3 * H);
  w.zero = b.take<float>(wide);
  w.tmp = b.take<float>(Mb * wide);
  w.dx = b.take<float>(Mb * H);
  w.dr = b.take<float>(Mb * H);
  w.dbig = b.take<float>(Mb * wide);
  w.dqkv = b.take<float>(Mb * 3 * H);
  w.dsum = b.take<float>(Mb * c.heads);
  w.dw = b.take<float>(wide * (I > H ? I : H));
  w.part = b.take<float>(2 * (size_t)LN_BWD_BLOCKS * wide);
  w.h16a = b.take<uint16_t>(Mb * H);
  w.h16b = b.take<uint16_t>(Mb * H);
  w.big16 = b.take<uint16_t>(Mb * wide);
  w.dwp = b.take<float>((size_t)1024 * 65536 + wide * (I > H ? I : H));    // slices * Nout * K <= (#CUs <= 1024) * 256 * 256 + Nout * K
  w.dims = b.take<int32_t>(16);
}

int check_cfg(const manner_hip_encoder_config* c, int64_t N, int64_t Lp, int64_t Mb, int32_t precision, int start) {
  if (!c) return fail(MANNER_HIP_E_INVALID, "train: null config");
  if (c->hidden % 128 || c->intermediate % 128 || c->hidden / c->heads != AD || c->hidden % c->heads || c->hidden > 64 * LN_MAX ||
      c->layers <= 0 || c->layers > 64)
    return fail(MANNER_HIP_E_INVALID, "train: unsupported architecture H=%d I=%d heads=%d layers=%d", c->hidden, c->intermediate,
                c->heads, c->layers);
  if (N <= 0 || Lp <= 0 || Lp > MANNER_HIP_MAX_LEN)
    return fail(MANNER_HIP_E_INVALID, "train: n_news=%lld padded_len=%lld (padded_len <= %d, as the inference engine)", (long long)N, (long long)Lp, MANNER_HIP_MAX_LEN);
  if (Mb <= 0 || Mb % 256 || Mb > 0x7fffff00ll / (3 * (int64_t)c->hidden)) return fail(MANNER_HIP_E_INVALID, "train: m_bound=%lld must be a positive multiple of 256 within int range", (long long)Mb);
  if (Mb < round_up(N, 256)) return fail(MANNER_HIP_E_INVALID, "train: m_bound=%lld holds fewer rows than there are news (%lld): every news has a token", (long long)Mb, (long long)N);
  if (precision != MANNER_HIP_PREC_F32 && precision != MANNER_HIP_PREC_BF16 && precision != MANNER_HIP_PREC_F16)
    return fail(MANNER_HIP_E_INVALID, "train: precision %d (fp32, bf16, f16)", precision);
  if (start < 0 || start >= c->layers) return fail(MANNER_HIP_E_INVALID, "train: start_layer %d of %d layers", start, c->layers);
  return MANNER_HIP_OK;
}

struct Ctx {
  const manner_hip_encoder_config* c;
  const float* const* w;
  int64_t N, Lp, Mb;
  int prec;
  hipStream_t s;
  Saved sv;
  Work wk;
  const float* emb(int i) const { return w[i]; }
  const float* lw(int l, int i) const { return w[MANNER_HIP_W_EMB_COUNT + l * MANNER_HIP_WL_COUNT + i]; }
  DType dt() const { return prec == MANNER_HIP_PREC_F32 ? DT_F32 : prec == MANNER_HIP_PREC_F16 ? DT_F16 : DT_BF16; }
  Out16 o16(void* p) const { return (p && dt() != DT_F32) ? Out16{p, (int)dt()} : Out16{nullptr, 0}; }
  // 16-bit modes: Q | K | V are kept in the 16-bit type (the f32 slot of the saved buffer, half used) and the attention runs on the
  // matrix pipe (train_attn.hip).  MANNER_HIP_TRAIN_ATTN_VALU=1 keeps the f32 VALU kernels for A/B; the fp32 mode always uses them.
  // ONE decision per forward, made in setup() (`choose_attn_path`: environment read per call, "full rows" excluded: they need the key
  // length mask) and RECORDED against the saved buffer; the backward reads the record instead of re-deriving it, because the layout
  // of L.qkv (16-bit or f32) follows from it.
  bool attn_mfma = false;
  bool mfma_attn() const { return attn_mfma; }
  // Round 5 — 16-bit saved activations (`lean`; MANNER_HIP_TRAIN_SAVE16=0 switches it off for A/B): with the matrix-pipe attention
  // and the fused GeLU epilogues available for the token-row shapes, the tensors that only GEMMs / the attention consume exist in the
  // 16-bit type ONLY — what the reference's `precision: 16-mixed` autocast saves (the f16 outputs of its Linear layers and matmuls):
  //   ctx    : the attention writes the 16-bit rows alone (its f32 rows had no reader but a conversion)
  //   inter  : FFN1 saves round16(h1 W1^T + b1) and gelu of THAT value (EPI_BIAS_GELU_DUAL16); gelu' in the backward reads it
  //   d ctx  : leaves its data-gradient GEMM in the 16-bit type, straight into the attention backward's operand buffer
  //   d qkv  : the attention backward writes the 16-bit rows alone; the Q|K|V bias gradient is their column sum
  // The f32 residual stream (r1, h1, r2, x: LayerNorm inputs / outputs, f32 under autocast too) is untouched.  Decided once per
  // forward (setup of train_forward_impl), recorded with the attention path against the saved buffer, read back by the backward.
  // The compact [CLS] tail of the last layer keeps the f32 pre-activation (its n_news rows are a small problem: separate GeLU kernels).
  bool lean = false;
  // Optional cache of 16-bit weight copies across calls (manner_hip_train_weight_cache; round 4): slot 2 i = the mode's 16-bit copy of
  // weight i of the table in its own layout, slot 2 i + 1 = its transpose (what the data-gradient GEMMs read); at a layer's Q weight:
  // the packed [3H, H] Q | K | V copy and its transpose, at its Q bias (slot 2 i): the f32 [3H] bias concatenation.  A slot the caller
  // left NULL is not cached; valid[s] == 0: filled on first use and marked.  FROZEN weights only — the caller clears `valid` when a
  // weight's contents change.
  void* const* wc_slots = nullptr;
  int32_t* wc_valid = nullptr;
  int wc_n = 0, n_w = 0;
  int index_of(const float* W) const {
    for (int i = 0; i < n_w; ++i) if (w[i] == W) return i;
    return -1;
  }
  void* slot(int s) const { return (wc_slots && s >= 0 && s < wc_n) ? wc_slots[s] : nullptr; }
  unsigned ew_grid(int64_t width) const { const int64_t b = (Mb * width + 255) / 256; return (unsigned)(b < 8192 ? b : 8192); }
};

int transpose_to(Ctx& t, const void* in, DType in_dt, int64_t rows_in, int64_t cols, void* out, int64_t rows_out, const int* rv_dev,
                 int64_t rv_host, int64_t ks = 0);

// Weight-gradient work on a SECOND stream (round 4).  The parameter gradients of a layer (4 weight-gradient GEMMs with their slice
// reductions, 4 bias column sums, the operand conversions) are needed by nobody before the optimiser step, while the activation-
// gradient chain runs launches that leave CUs idle (the N = H data-gradient GEMMs fill 183 of 256, the attention backward is
// latency-bound).  Per device: one non-blocking stream, one "inputs are ready" event and one "done" event per group of a layer
// (FFN2, FFN1, out-projection, Q|K|V); the main stream waits for a group's "done" right before it overwrites that group's input.
// Host threads: the stream and its events are ONE set per device, and a backward's record / wait pairs on them must not
// interleave with another backward's (B's record of `ready` between A's record and A's wait would order A's side work behind B's main
// stream instead of its own).  `mu` guards the lazy creation and is held by a backward for as long as it enqueues with the side stream
// — two host threads training on one device serialise their (1 - 2 ms) enqueue phases; their kernels still share the device in stream
// order.
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t ready = nullptr, done[4] = {nullptr, nullptr, nullptr, nullptr};
  bool ok = false, tried = false;
  std::mutex mu;
};
SideStream& side_stream() {
  static SideStream per_device[MAX_DEVICES];
  SideStream& sd = per_device[current_device_slot()];
  std::lock_guard<std::mutex> guard(sd.mu);
  if (!sd.tried) {
    sd.tried = true;
    // LOWEST priority: the side stream should take the CUs the chain leaves idle, not compete for them (measured, default step of
    // bench.py's training leg: lowest 14.10 ms, default priority 14.38, highest 14.54; one stream 14.79)
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    bool ok = hipStreamCreateWithPriority(&sd.s, hipStreamNonBlocking, least) == hipSuccess &&
              hipEventCreateWithFlags(&sd.ready, hipEventDisableTiming) == hipSuccess;
    for (int g = 0; ok && g < 4; ++g) ok = hipEventCreateWithFlags(&sd.done[g], hipEventDisableTiming) == hipSuccess;
    sd.ok = ok;
  }
  return sd;
}

// The mode's 16-bit copy of weight matrix W [rows, cols] f32 (tr: of its transpose [cols, rows]) — from the caller's cache slot when
// one was registered for it (filled here on first use), else converted into `scratch`.  wi: W's index in the weight table (-1: look it up).
int weight16(Ctx& t, const float* W, int rows, int cols, bool tr, void* scratch, const void** out, int wi = -1) {
  if (wi < 0) wi = t.index_of(W);
  const int si = wi < 0 ? -1 : 2 * wi + (tr ? 1 : 0);
  void* dst = t.slot(si);
  if (dst && t.wc_valid[si]) { *out = dst; return MANNER_HIP_OK; }
  int rc;
  void* to = dst ? dst : scratch;
  if (tr) rc = transpose_to(t, W, DT_F32, rows, cols, to, rows, nullptr, rows, 0);
  else rc = convert_f32_to_16(t.dt(), W, to, (int64_t)rows * cols, t.s);
  if (rc) return rc;
  if (dst) t.wc_valid[si] = 1;
  *out = to;
  return MANNER_HIP_OK;
}

// Y [Mb, Nout] = X [Mb, K] . W [Nout, K]^T + bias      (f32 in / out; operands rounded to the 16-bit type in the mixed modes)
// X16: the producer's 16-bit copy of X when there is one (else X is converted here)
int linear_fwd(Ctx& t, const float* X, const float* W, const float* bias, float* Y, int Nout, int K, const void* X16 = nullptr) {
  int rc;
  if (t.dt() == DT_F32) return gemm_tn(DT_F32, DT_F32, EPI_BIAS, X, W, bias, nullptr, Y, t.Mb, Nout, K, t.sv.m_total, t.s);
  if (!X16) {
    if ((rc = convert_f32_to_16(t.dt(), X, t.wk.a16, t.Mb * K, t.s))) return rc;
    X16 = t.wk.a16;
  }
  const void* w16;
  if ((rc = weight16(t, W, Nout, K, false, t.wk.b16, &w16))) return rc;
  return gemm_tn(t.dt(), DT_F32, EPI_BIAS, X16, w16, bias, nullptr, Y, t.Mb, Nout, K, t.sv.m_total, t.s);
}

// A/B switch (MANNER_HIP_TRAIN_GELU_FUSED=0): the GeLU of the FFN as separate elementwise kernels in the 16-bit modes too
static bool gelu_fused_enabled() {
  const char* e = getenv("MANNER_HIP_TRAIN_GELU_FUSED");
  return !e || atoi(e) != 0;
}

int gelu16(Ctx& t, const float* inter, void* z, int width, int mode);

// FFN1 of the 16-bit modes: inter [Mb, I] f32 = h1 W1^T + b1 (saved for the backward's gelu') and g [Mb, I] = gelu(inter) in the
// 16-bit type — one GEMM with both outputs where the shape allows, else GEMM + gelu16
// pre16 (round 5, Ctx::lean): the saved pre-activation is round16(h1 W1^T + b1) in the 16-bit type (the f32 slot, half used)
int ffn1_fwd16(Ctx& t, const void* h16, const float* W, const float* bias, float* inter, void* g, int I, int H, bool pre16 = false) {
  int rc;
  if (pre16) {                                              // lean was chosen only where this shape is fusable
    const void* w16;
    if ((rc = weight16(t, W, I, H, false, t.wk.b16, &w16))) return rc;
    return gemm_tn_gelu_dual16(t.dt(), h16, w16, bias, inter, g, t.Mb, I, H, t.sv.m_total, t.s);
  }
  if (gelu_fused_enabled() && gemm_gelu_fusable(t.dt(), t.Mb, I, H)) {
    const void* w16;
    if ((rc = weight16(t, W, I, H, false, t.wk.b16, &w16))) return rc;
    return gemm_tn_gelu_dual(t.dt(), h16, w16, bias, inter, g, t.Mb, I, H, t.sv.m_total, t.s);
  }
  if ((rc = linear_fwd(t, nullptr, W, bias, inter, I, H, h16))) return rc;
  return gelu16(t, inter, g, I, 0);
}

// R = dropout(X W^T + bias) + residual (attention-output / FFN-output projection of the forward pass, modeling_bert.py:289-293,
// 347-351): 16-bit modes on 256-tileable shapes run it as ONE GEMM (dropout and residual add in the epilogue); otherwise GEMM ->
// wk.tmp, then the elementwise pass.  rowmap: compact [CLS] rows draw the bits of the token rows they stand for.
int linear_fwd_drop_res(Ctx& t, const float* X, const void* X16, const float* W, const float* bias, const float* residual, float* R, int Nout,
                        int K, const Drop& drop, const int32_t* rowmap);

template <typename TS, typename T>
void launch_transpose(const void* in, int64_t cols, void* out, int64_t rows_out, const int* rv_dev, int64_t rv_host, int64_t ks,
                      hipStream_t s) {
  dim3 g((unsigned)((cols + 31) / 32), (unsigned)((rows_out + 31) / 32));
  hipLaunchKernelGGL((transpose_kernel<TS, T>), g, dim3(256), 0, s, static_cast<const TS*>(in), cols, static_cast<T*>(out), rows_out, rv_dev,
                     rv_host, ks);
}
// `in` is [rows_in, cols] of type in_dt (f32, or — 16-bit modes — already the mode's 16-bit type); ks = 0: one slice
int transpose_to(Ctx& t, const void* in, DType in_dt, int64_t rows_in, int64_t cols, void* out, int64_t rows_out, const int* rv_dev,
                 int64_t rv_host, int64_t ks) {
  (void)rows_in;
  if (ks <= 0) ks = rows_out;
  if (in_dt != DT_F32 && in_dt != t.dt()) return fail(MANNER_HIP_E_INVALID, "train: transpose source type %d in mode %d", (int)in_dt, (int)t.dt());
  if (t.dt() == DT_F32) launch_transpose<float, float>(in, cols, out, rows_out, rv_dev, rv_host, ks, t.s);
  else if (t.dt() == DT_F16) {
    if (in_dt == DT_F32) launch_transpose<float, f16_t>(in, cols, out, rows_out, rv_dev, rv_host, ks, t.s);
    else launch_transpose<f16_t, f16_t>(in, cols, out, rows_out, rv_dev, rv_host, ks, t.s);
  } else {
    if (in_dt == DT_F32) launch_transpose<float, bf16_t>(in, cols, out, rows_out, rv_dev, rv_host, ks, t.s);
    else launch_transpose<bf16_t, bf16_t>(in, cols, out, rows_out, rv_dev, rv_host, ks, t.s);
  }
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

// dX [Mb, K] = dY [Mb, Nout] . W [Nout, K]  (+ residual [Mb, K] f32 when given: fused into the GEMM's epilogue where the shape
// allows — *fused tells the caller whether it was).  out_dt: f32, or (16-bit modes) the mode's 16-bit type for the I-wide d g.
int linear_dgrad(Ctx& t, const float* dY, const float* W, void* dX, int Nout, int K, const void* dY16 = nullptr, DType out_dt = DT_F32,
                 const float* residual = nullptr, bool* fused = nullptr, int wi = -1) {
  int rc;
  if (fused) *fused = false;
  const void* wt = t.wk.b16;                                                                        // W^T [K, Nout]
  if (t.dt() == DT_F32) { if ((rc = transpose_to(t, W, DT_F32, Nout, K, t.wk.b16, Nout, nullptr, Nout))) return rc; }
  else if ((rc = weight16(t, W, Nout, K, true, t.wk.b16, &wt, wi))) return rc;
  const void* x = dY;
  if (t.dt() != DT_F32) {
    if (!dY16) {
      if ((rc = convert_f32_to_16(t.dt(), dY, t.wk.a16, t.Mb * Nout, t.s))) return rc;
      dY16 = t.wk.a16;
    }
    x = dY16;
    if (residual && out_dt == DT_F32 && t.Mb % 256 == 0 && K % 256 == 0 && Nout >= 128) {
      if (fused) *fused = true;
      return gemm_tn(t.dt(), DT_F32, EPI_BIAS_RES_F32, x, wt, t.wk.zero, residual, dX, t.Mb, K, Nout, t.sv.m_total, t.s);
    }
  }
  return gemm_tn(t.dt(), out_dt, EPI_BIAS, x, wt, t.wk.zero, nullptr, dX, t.Mb, K, Nout, t.sv.m_total, t.s);
}

// dW [Nout, K] = dY [Mb, Nout]^T . X [Mb, K]   (the reduction runs over the token rows; rows >= *m_total contribute zeros)
// dy_dt / x_dt: the operands' storage type (f32, or the mode's 16-bit type when the tensor exists in 16 bits only)
// split (optional; *split_done reports whether it was honoured): the reduction of the slice partials writes three row blocks of dW
// [3 seg / K rows each] straight into three destinations instead of dW
int linear_wgrad(Ctx& t, const void* dY, DType dy_dt, const void* X, DType x_dt, float* dW, int Nout, int K, int dim_slot,
                 const SumDst* split = nullptr, bool* split_done = nullptr) {
  int rc;
  if (split_done) *split_done = false;
  if (t.dt() != DT_F32 && Nout % 256 == 0 && K % 256 == 0) {
    // Few output tiles, one long reduction: split the token axis into `slices` independent GEMMs of one launch
    // (gridDim.y) so that every CU has a tile, then add the partial gradients in a fixed order.
    const int tiles = (Nout / 256) * (K / 256);
    int slices = (device_cus() + tiles - 1) / tiles;
    const int64_t max_slices = t.Mb / 128;
    if (slices > max_slices) slices = (int)max_slices;
    if (slices > WGRAD_MAX_SLICES) slices = WGRAD_MAX_SLICES;
    if (slices < 1) slices = 1;
    const char* tr_env = getenv("MANNER_HIP_WGRAD_TR");                 // A/B switch, read per call (the tests flip it)
    if (!tr_env || atoi(tr_env) != 0) {
      // No transposed copies: the row-major 16-bit operands go to LDS as they are and the MFMA fragments are read transposed
      // (wgrad.hip).  An f32 source is converted (no transposition) first.
      const int64_t ks = round_up((t.Mb + slices - 1) / slices, 64);
      const void* dy16 = dY;
      const void* x16 = X;
      if (dy_dt == DT_F32) { if ((rc = convert_f32_to_16(t.dt(), static_cast<const float*>(dY), t.wk.a16, t.Mb * Nout, t.s))) return rc; dy16 = t.wk.a16; }
      if (x_dt == DT_F32) { if ((rc = convert_f32_to_16(t.dt(), static_cast<const float*>(X), t.wk.b16, t.Mb * K, t.s))) return rc; x16 = t.wk.b16; }
      if ((rc = wgrad_tr(t.dt(), dy16, x16, slices >= 2 ? t.wk.dwp : dW, Nout, K, slices, ks, t.sv.m_total, t.wk.zero, t.s))) return rc;
      if (slices >= 2) {
        if (split && (int64_t)Nout * K < 0x7fffffff) {
          reduce_partials(t.s, t.wk.dwp, slices, (int64_t)Nout * K, *split);
          if (split_done) *split_done = true;
        } else {
          reduce_partials(t.s, t.wk.dwp, slices, (int64_t)Nout * K, dW);
        }
        MANNER_LAUNCH_CHECK();
      }
      return MANNER_HIP_OK;
    }
    if (slices >= 2) {
      const int64_t ks = round_up((t.Mb + slices - 1) / slices, 64), Mp = ks * slices;
      if ((rc = transpose_to(t, dY, dy_dt, t.Mb, Nout, t.wk.a16, Mp, t.sv.m_total, 0, ks))) return rc;     // [slices][Nout, ks]
      if ((rc = transpose_to(t, X, x_dt, t.Mb, K, t.wk.b16, Mp, t.sv.m_total, 0, ks))) return rc;          // [slices][K, ks]
      if ((rc = set_device_int(t.wk.dims + dim_slot, Nout, t.s))) return rc;
      if ((rc = gemm_tn_batched16(t.dt(), t.wk.a16, t.wk.b16, t.wk.zero, t.wk.dwp, slices, (int64_t)Nout * ks, (int64_t)K * ks,
                                  (int64_t)Nout * K, Nout, K, (int)ks, t.wk.dims + dim_slot, t.s)))
        return rc;
      const int64_t width = (int64_t)Nout * K;
      reduce_partials(t.s, t.wk.dwp, slices, width, dW);
      MANNER_LAUNCH_CHECK();
      return MANNER_HIP_OK;
    }
  }
  if ((rc = transpose_to(t, dY, dy_dt, t.Mb, Nout, t.wk.a16, t.Mb, t.sv.m_total, 0))) return rc;  // dY^T [Nout, Mb]
  if ((rc = transpose_to(t, X, x_dt, t.Mb, K, t.wk.b16, t.Mb, t.sv.m_total, 0))) return rc;      // X^T  [K, Mb]
  if ((rc = set_device_int(t.wk.dims + dim_slot, Nout, t.s))) return rc;
  return gemm_tn(t.dt(), DT_F32, EPI_BIAS, t.wk.a16, t.wk.b16, t.wk.zero, nullptr, dW, Nout, K, (int)t.Mb, t.wk.dims + dim_slot, t.s);
}

int bias_grad(Ctx& t, const void* dY, DType dy_dt, int width, float* db, const SumDst* split = nullptr) {
  const dim3 g((unsigned)((width + 255) / 256), COLSUM_BLOCKS), b(256);
  if (dy_dt == DT_F32) hipLaunchKernelGGL(colsum_kernel<float>, g, b, 0, t.s, static_cast<const float*>(dY), width, t.wk.part, t.sv.m_total);
  else if (dy_dt == DT_F16) hipLaunchKernelGGL(colsum_kernel<f16_t>, g, b, 0, t.s, static_cast<const f16_t*>(dY), width, t.wk.part, t.sv.m_total);
  else hipLaunchKernelGGL(colsum_kernel<bf16_t>, g, b, 0, t.s, static_cast<const bf16_t*>(dY), width, t.wk.part, t.sv.m_total);
  MANNER_LAUNCH_CHECK();
  if (split) reduce_partials(t.s, t.wk.part, COLSUM_BLOCKS, width, *split);
  else reduce_partials(t.s, t.wk.part, COLSUM_BLOCKS, width, db);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

// gelu between the two FFN GEMMs: fp32 mode f32 -> f32 (out / grad as before); 16-bit modes: the 16-bit-only wide tensor `z`
int gelu16(Ctx& t, const float* inter, void* z, int width, int mode) {
  const unsigned grid = t.ew_grid(width / 4 > 0 ? width / 4 : 1);
  if (t.dt() == DT_F16) hipLaunchKernelGGL(gelu16_kernel<f16_t>, dim3(grid), dim3(256), 0, t.s, inter, static_cast<f16_t*>(z), width, t.sv.m_total, mode);
  else hipLaunchKernelGGL(gelu16_kernel<bf16_t>, dim3(grid), dim3(256), 0, t.s, inter, static_cast<bf16_t*>(z), width, t.sv.m_total, mode);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int ln_forward(Ctx& t, const float* x, const float* g, const float* b, float* y, float2* st, Drop drop, void* y16 = nullptr) {
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)(t.Mb / 4)), dim3(256), 0, t.s, x, g, b, t.c->hidden, t.c->ln_eps, y, st, t.sv.m_total, drop,
                     t.o16(y16));
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}
int dropout_add(Ctx& t, const float* a, const float* res, float* out, int width, Drop drop, void* out16 = nullptr,
                const int32_t* rowmap = nullptr);
int linear_fwd_drop_res(Ctx& t, const float* X, const void* X16, const float* W, const float* bias, const float* residual, float* R, int Nout,
                        int K, const Drop& drop, const int32_t* rowmap) {
  int rc;
  if (t.dt() != DT_F32 && X16 && t.Mb % 256 == 0 && Nout % 256 == 0 && K >= 128 && (K * 2) % 128 == 0) {
    const void* w16;
    if ((rc = weight16(t, W, Nout, K, false, t.wk.b16, &w16))) return rc;
    return gemm_tn_drop_res(t.dt(), X16, w16, bias, residual, R, t.Mb, Nout, K, t.sv.m_total, drop, rowmap, t.s);
  }
  if ((rc = linear_fwd(t, X, W, bias, t.wk.tmp, Nout, K, X16))) return rc;
  return dropout_add(t, t.wk.tmp, residual, R, Nout, drop, nullptr, rowmap);
}

int ln_backward(Ctx& t, const float* dy, const float* x, const float2* st, const float* gamma, float* dx, float* dgamma, float* dbeta,
                Drop drop = Drop{0, 0, 0, 1.f}, void* d16 = nullptr, const int32_t* rowmap = nullptr) {
  const int H = t.c->hidden;
  // partial rows [block][d gamma (H) | d beta (H)]: ONE fixed-order reduction writes both parameter gradients (round 5; two launches
  // of 18 us each before — same sums in the same order)
  float* pg = dgamma ? t.wk.part : nullptr;
  float* pb = dgamma ? t.wk.part + H : nullptr;
  // H <= 768 (bert-base): three float4 pieces per lane instead of four — 36 registers fewer, one more wave per SIMD
  if (H <= 768) hipLaunchKernelGGL(ln_bwd_kernel<3>, dim3(LN_BWD_BLOCKS), dim3(256), 0, t.s, dy, x, st, gamma, H, dx, pg, pb, t.sv.m_total, drop, t.o16(d16), rowmap, 2 * H);
  else hipLaunchKernelGGL(ln_bwd_kernel<LN_V4>, dim3(LN_BWD_BLOCKS), dim3(256), 0, t.s, dy, x, st, gamma, H, dx, pg, pb, t.sv.m_total, drop, t.o16(d16), rowmap, 2 * H);
  MANNER_LAUNCH_CHECK();
  if (dgamma) {
    reduce_partials(t.s, pg, LN_BWD_BLOCKS, 2 * H, SumDst{{dgamma, dbeta, nullptr}, H});
    MANNER_LAUNCH_CHECK();
  }
  return MANNER_HIP_OK;
}
int dropout_add(Ctx& t, const float* a, const float* res, float* out, int width, Drop drop, void* out16, const int32_t* rowmap) {
  hipLaunchKernelGGL(dropout_add_kernel, dim3(t.ew_grid(width)), dim3(256), 0, t.s, a, res, out, width, t.sv.m_total, drop, t.o16(out16), rowmap);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}
int add_rows(Ctx& t, const float* a, const float* b, float* out, int width) {
  hipLaunchKernelGGL(add_rows_kernel, dim3(t.ew_grid(width)), dim3(256), 0, t.s, a, b, out, width, t.sv.m_total);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

// [Wq; Wk; Wv] -> wcat [3H, H] f32 (+ its 16-bit copy when w16 != NULL), [bq; bk; bv] -> bcat: ONE launch (it was six
// 5 us copies + a conversion pass per layer and call)
template <typename TE>
__global__ __launch_bounds__(256) void pack3_kernel(const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
                                                    const float* __restrict__ bq, const float* __restrict__ bk, const float* __restrict__ bv,
                                                    int H, float* __restrict__ wcat, float* __restrict__ bcat, TE* __restrict__ w16) {
  const int64_t per = (int64_t)H * H / 4;                    // float4 pieces per matrix
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 3 * per; i += (int64_t)gridDim.x * 256) {
    const int k = (int)(i / per);
    const int64_t j = i - k * per;
    const f32x4 v = reinterpret_cast<const f32x4*>(k == 0 ? wq : k == 1 ? wk : wv)[j];
    reinterpret_cast<f32x4*>(wcat)[i] = v;
    if (w16) {
      typedef TE v4 __attribute__((ext_vector_type(4)));
      reinterpret_cast<v4*>(w16)[i] = v4{(TE)v[0], (TE)v[1], (TE)v[2], (TE)v[3]};
    }
  }
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < 3 * H; c += 256) bcat[c] = c < H ? bq[c] : c < 2 * H ? bk[c - H] : bv[c - 2 * H];
}
int pack_qkv_weights(Ctx& t, int l, void* w16 = nullptr) {
  const int H = t.c->hidden;
  const int64_t pieces = 3 * (int64_t)H * H / 4;
  const dim3 g((unsigned)((pieces + 255) / 256 < 2048 ? (pieces + 255) / 256 : 2048)), b(256);
  const float *wq = t.lw(l, MANNER_HIP_WL_Q_W), *wk = t.lw(l, MANNER_HIP_WL_K_W), *wv = t.lw(l, MANNER_HIP_WL_V_W);
  const float *bq = t.lw(l, MANNER_HIP_WL_Q_B), *bk = t.lw(l, MANNER_HIP_WL_K_B), *bv = t.lw(l, MANNER_HIP_WL_V_B);
  if (w16 && t.dt() == DT_F16)
    hipLaunchKernelGGL(pack3_kernel<f16_t>, g, b, 0, t.s, wq, wk, wv, bq, bk, bv, H, t.wk.wcat, t.wk.bcat, static_cast<f16_t*>(w16));
  else
    hipLaunchKernelGGL(pack3_kernel<bf16_t>, g, b, 0, t.s, wq, wk, wv, bq, bk, bv, H, t.wk.wcat, t.wk.bcat,
                       (w16 && t.dt() == DT_BF16) ? static_cast<bf16_t*>(w16) : static_cast<bf16_t*>(nullptr));
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

// attention path of a training forward: 1 = matrix pipe with 16-bit Q | K | V (train_attn.hip), 0 = f32 VALU kernels.  The forward
// records its choice per saved buffer; the backward of the same buffer looks it up (falls back to the rule for a buffer it has
// never seen, e.g. one produced by another process image).
static std::mutex g_attn_mu;
static std::unordered_map<const void*, int> g_attn_path;
static bool choose_attn_path(const manner_hip_encoder_config* c, int32_t precision, bool full) {
  const char* v = getenv("MANNER_HIP_TRAIN_ATTN_VALU");                    // read per call: tests compare both paths in one process
  const bool valu = v && *v && *v != '0';
  return precision != MANNER_HIP_PREC_F32 && !valu && !full && c->hidden == c->heads * 64;   // the MFMA kernels are written for head_dim 64
}
static bool choose_lean(const manner_hip_encoder_config* c, int32_t precision, int64_t Mb, bool mfma) {
  const char* v = getenv("MANNER_HIP_TRAIN_SAVE16");                       // read per call: the tests compare both layouts in one process
  const bool off = v && *v == '0';
  const DType dt = precision == MANNER_HIP_PREC_F16 ? DT_F16 : DT_BF16;
  return mfma && !off && gelu_fused_enabled() && gemm_gelu_fusable(dt, Mb, c->intermediate, c->hidden) && gemm_gelu_fusable(dt, Mb, c->hidden, c->hidden);
}
static void record_attn_path(const void* saved, bool mfma, bool lean) {
  std::lock_guard<std::mutex> g(g_attn_mu);
  // one entry per distinct saved-buffer ADDRESS (an allocator recycles them, so a handful in practice); bounded all the same: past
  // the cap the table starts over, and a backward that no longer finds its record falls back to the rule its forward applied
  if (g_attn_path.size() >= 4096 && g_attn_path.find(saved) == g_attn_path.end()) g_attn_path.clear();
  g_attn_path[saved] = (mfma ? 1 : 0) | (lean ? 2 : 0);
}
static thread_local int32_t g_layout_last = -1, g_layout_next = -1;      // ABI v8: the layout word carried by the caller
// the layout the forward of `saved` chose: the word the caller handed over (consumed), else this process image's record; -1 = neither
static int recorded_layout(const void* saved) {
  const int32_t w = g_layout_next;
  g_layout_next = -1;
  if (w >= 0) return w & 3;
  std::lock_guard<std::mutex> g(g_attn_mu);
  auto it = g_attn_path.find(saved);
  return it == g_attn_path.end() ? -1 : it->second;
}

struct WCacheArg { void* const* slots = nullptr; int32_t* valid = nullptr; int n = 0; };
static thread_local WCacheArg g_next_wcache;
// The registration belongs to the NEXT train_* call of the thread and to that call only: every entry point takes it (and clears the
// thread-local) before anything can fail, so a call that returns early never leaves pointers into the caller's dropped arrays behind
// for a later call that registers nothing.
static WCacheArg take_wcache() {
  const WCacheArg a = g_next_wcache;
  g_next_wcache = WCacheArg{};
  return a;
}

int setup(Ctx& t, const manner_hip_encoder_config* cfg, const float* const* weights, int32_t n_weights, int64_t N, int64_t Lp, int64_t Mb,
          int32_t precision, int start, void* saved, size_t saved_bytes, void* ws, size_t ws_bytes, hipStream_t s, const WCacheArg& wc,
          bool lean_layout) {
  int rc;
  if ((rc = check_cfg(cfg, N, Lp, Mb, precision, start))) return rc;
  if (!weights || n_weights != MANNER_HIP_W_EMB_COUNT + cfg->layers * MANNER_HIP_WL_COUNT)
    return fail(MANNER_HIP_E_INVALID, "train: weight table has %d entries, the architecture needs %d", n_weights,
                MANNER_HIP_W_EMB_COUNT + cfg->layers * MANNER_HIP_WL_COUNT);
  for (int i = 0; i < n_weights; ++i)
    if (!weights[i]) return fail(MANNER_HIP_E_INVALID, "train: weight %d is NULL", i);
  if (!saved || !ws) return fail(MANNER_HIP_E_INVALID, "train: null buffer");
  t.c = cfg; t.w = weights; t.N = N; t.Lp = Lp; t.Mb = Mb; t.prec = precision; t.s = s;
  t.n_w = n_weights;
  if (wc.slots && wc.n == 2 * n_weights && precision != MANNER_HIP_PREC_F32) {    // registered for THIS call (same thread)
    t.wc_slots = wc.slots;
    t.wc_valid = wc.valid;
    t.wc_n = wc.n;
  }
  Bump bs(saved), bw(ws);
  plan_saved(bs, t.sv, *cfg, N, Mb, start, lean_layout);
  plan_work(bw, t.wk, *cfg, Mb);
  if (bs.off > saved_bytes) return fail(MANNER_HIP_E_WORKSPACE, "train: saved buffer %zu < %zu bytes", saved_bytes, bs.off);
  if (bw.off > ws_bytes) return fail(MANNER_HIP_E_WORKSPACE, "train: workspace %zu < %zu bytes", ws_bytes, bw.off);
  const size_t wide = (size_t)(cfg->intermediate > 3 * cfg->hidden ? cfg->intermediate : 3 * cfg->hidden);
  MANNER_HIP_TRY(hipMemsetAsync(t.wk.zero, 0, wide * sizeof(float), s));
  return MANNER_HIP_OK;
}


// One BertLayer in train() arithmetic (modeling_bert.py:175-203, 289-293, 334-351); L receives what the backward needs.
// cls_only (the training path's LAST layer): after the attention only the [CLS] row of every news goes on — the rows
// r1 / h1 / inter / g / r2 / st1 / st2 of L and x_out are then COMPACT (row n = news n, n_news rows), as in the
// inference engine's [CLS] tail; x_out [n_news, H].
int layer_forward(Ctx& t, int l, LayerSaved& L, const float* x_in, float* x_out, const int32_t* cu, const int32_t* klen, float p_hidden,
                  float p_attn, uint64_t seed, bool x_in_has_16, bool cls_only = false) {
  // 16-bit copies ride along with the f32 activations (mixed modes): the layer input in wk.h16a (written by whoever produced
  // x_in when x_in_has_16), ctx then h1 in wk.h16b, gelu's output in wk.big16; this layer's output copy goes to wk.h16a again
  int rc;
  const manner_hip_encoder_config* cfg = t.c;
  const int H = cfg->hidden, I = cfg->intermediate;
  hipStream_t s = t.s;
  const bool mixed = t.dt() != DT_F32;
  const bool mfma = t.mfma_attn();
  // the packed Q | K | V weight (16-bit) and bias: from the caller's cache for a frozen layer (no launch at all once filled)
  const int wiq = MANNER_HIP_W_EMB_COUNT + l * MANNER_HIP_WL_COUNT + MANNER_HIP_WL_Q_W, wibq = MANNER_HIP_W_EMB_COUNT + l * MANNER_HIP_WL_COUNT + MANNER_HIP_WL_Q_B;
  const void* qkv_w16 = t.wk.b16;
  const float* qkv_b = t.wk.bcat;
  void* cw = mfma ? t.slot(2 * wiq) : nullptr;
  float* cb = mfma ? static_cast<float*>(t.slot(2 * wibq)) : nullptr;
  if (cw && cb && t.wc_valid[2 * wiq] && t.wc_valid[2 * wibq]) {
    qkv_w16 = cw;
    qkv_b = cb;
  } else {
    if ((rc = pack_qkv_weights(t, l, mfma ? (cw && cb ? cw : t.wk.b16) : nullptr))) return rc;
    if (cw && cb) {
      MANNER_HIP_TRY(hipMemcpyAsync(cb, t.wk.bcat, (size_t)3 * H * sizeof(float), hipMemcpyDeviceToDevice, s));
      t.wc_valid[2 * wiq] = t.wc_valid[2 * wibq] = 1;
      qkv_w16 = cw;
      qkv_b = cb;
    }
  }
  const Drop da_m = make_drop(seed, layer_site(l, SITE_ATTN), p_attn);
  if (mfma) {
    // Q | K | V straight in the 16-bit type (one rounding, at the GEMM's output — what the MFMA attention reads), then the
    // matrix-pipe attention: ctx f32 + its 16-bit copy + {row max, row sum}
    const void* x16 = t.wk.h16a;
    if (!x_in_has_16) {
      if ((rc = convert_f32_to_16(t.dt(), x_in, t.wk.a16, t.Mb * H, s))) return rc;
      x16 = t.wk.a16;
    }
    if ((rc = gemm_tn(t.dt(), t.dt(), EPI_BIAS, x16, qkv_w16, qkv_b, nullptr, L.qkv, t.Mb, 3 * H, H, t.sv.m_total, s))) return rc;
    // lean: the 16-bit rows alone, into the saved slot (half used) — the out-projection GEMM reads them there
    if (t.lean) { if ((rc = attn_train_mfma_forward(t.dt(), L.qkv, nullptr, L.ctx, L.ml, cu, t.N, cfg->heads, H, (int)t.Lp, da_m, s))) return rc; }
    else if ((rc = attn_train_mfma_forward(t.dt(), L.qkv, L.ctx, t.wk.h16b, L.ml, cu, t.N, cfg->heads, H, (int)t.Lp, da_m, s))) return rc;
  } else {
    if ((rc = linear_fwd(t, x_in, t.wk.wcat, t.wk.bcat, L.qkv, 3 * H, H, mixed && x_in_has_16 ? t.wk.h16a : nullptr))) return rc;
    const Drop da = da_m;
#define MANNER_ATTN_FWD(AT_, HPB_)                                                                                             \
  hipLaunchKernelGGL((attn_train_fwd_kernel<AT_, HPB_>), dim3((unsigned)(cfg->heads / HPB_), (unsigned)t.N), dim3(AT_), 0, s, L.qkv, \
                     L.ctx, L.ml, cu, cfg->heads, H, da, klen, t.o16(t.wk.h16b))
    MANNER_ATTN_DISPATCH(t.Lp, cfg->heads, MANNER_ATTN_FWD);
#undef MANNER_ATTN_FWD
    MANNER_LAUNCH_CHECK();
  }
  if (cls_only) {
    Ctx c = t;                                               // the same helpers over n_news compact rows
    c.Mb = round_up(t.N, 256);
    c.sv.m_total = t.sv.m_total + 1;                         // {tokens, news}: the second entry bounds the compact rows
    float* ctx_c = t.wk.dr;                                  // [n_news, H] gathers (the backward gathers ctx again)
    float* x_c = t.wk.dbig;
    if (!t.lean) hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)t.N), dim3(256), 0, s, L.ctx, cu, H, ctx_c);
    else if (t.dt() == DT_F16) hipLaunchKernelGGL(gather_rows16_kernel<f16_t>, dim3((unsigned)t.N), dim3(256), 0, s, reinterpret_cast<const f16_t*>(L.ctx), cu, H, ctx_c);
    else hipLaunchKernelGGL(gather_rows16_kernel<bf16_t>, dim3((unsigned)t.N), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(L.ctx), cu, H, ctx_c);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)t.N), dim3(256), 0, s, x_in, cu, H, x_c);
    MANNER_LAUNCH_CHECK();
    if ((rc = linear_fwd_drop_res(c, ctx_c, nullptr, t.lw(l, MANNER_HIP_WL_AO_W), t.lw(l, MANNER_HIP_WL_AO_B), x_c, L.r1, H, H,
                                  make_drop(seed, layer_site(l, SITE_PROJ), p_hidden), cu))) return rc;
    if ((rc = ln_forward(c, L.r1, t.lw(l, MANNER_HIP_WL_ALN_G), t.lw(l, MANNER_HIP_WL_ALN_B), L.h1, L.st1, make_drop(0, 0, 0.f), t.wk.h16b))) return rc;
    if (mixed) {                                           // g exists in the 16-bit type only (the saved slot, half used)
      if ((rc = ffn1_fwd16(c, t.wk.h16b, t.lw(l, MANNER_HIP_WL_FF1_W), t.lw(l, MANNER_HIP_WL_FF1_B), L.inter, L.g, I, H))) return rc;
    } else {
      if ((rc = linear_fwd(c, L.h1, t.lw(l, MANNER_HIP_WL_FF1_W), t.lw(l, MANNER_HIP_WL_FF1_B), L.inter, I, H, nullptr))) return rc;
      hipLaunchKernelGGL(gelu_kernel, dim3(c.ew_grid(I)), dim3(256), 0, s, L.inter, nullptr, L.g, I, c.sv.m_total, 0, Out16{nullptr, 0});
      MANNER_LAUNCH_CHECK();
    }
    if ((rc = linear_fwd_drop_res(c, L.g, mixed ? (const void*)L.g : nullptr, t.lw(l, MANNER_HIP_WL_FF2_W), t.lw(l, MANNER_HIP_WL_FF2_B), L.h1, L.r2, H, I,
                                  make_drop(seed, layer_site(l, SITE_FFN), p_hidden), cu))) return rc;
    return ln_forward(c, L.r2, t.lw(l, MANNER_HIP_WL_OLN_G), t.lw(l, MANNER_HIP_WL_OLN_B), x_out, L.st2, make_drop(0, 0, 0.f));
  }
  if ((rc = linear_fwd_drop_res(t, t.lean ? nullptr : L.ctx, t.lean ? (const void*)L.ctx : (mixed ? (const void*)t.wk.h16b : nullptr), t.lw(l, MANNER_HIP_WL_AO_W),
                                t.lw(l, MANNER_HIP_WL_AO_B), x_in, L.r1, H, H, make_drop(seed, layer_site(l, SITE_PROJ), p_hidden), nullptr))) return rc;
  if ((rc = ln_forward(t, L.r1, t.lw(l, MANNER_HIP_WL_ALN_G), t.lw(l, MANNER_HIP_WL_ALN_B), L.h1, L.st1, make_drop(0, 0, 0.f), t.wk.h16b))) return rc;
  if (mixed) {
    if ((rc = ffn1_fwd16(t, t.wk.h16b, t.lw(l, MANNER_HIP_WL_FF1_W), t.lw(l, MANNER_HIP_WL_FF1_B), L.inter, L.g, I, H, t.lean))) return rc;
  } else {
    if ((rc = linear_fwd(t, L.h1, t.lw(l, MANNER_HIP_WL_FF1_W), t.lw(l, MANNER_HIP_WL_FF1_B), L.inter, I, H, nullptr))) return rc;
    hipLaunchKernelGGL(gelu_kernel, dim3(t.ew_grid(I)), dim3(256), 0, s, L.inter, nullptr, L.g, I, t.sv.m_total, 0, Out16{nullptr, 0});
    MANNER_LAUNCH_CHECK();
  }
  if ((rc = linear_fwd_drop_res(t, L.g, mixed ? (const void*)L.g : nullptr, t.lw(l, MANNER_HIP_WL_FF2_W), t.lw(l, MANNER_HIP_WL_FF2_B), L.h1, L.r2, H, I,
                                make_drop(seed, layer_site(l, SITE_FFN), p_hidden), nullptr))) return rc;
  return ln_forward(t, L.r2, t.lw(l, MANNER_HIP_WL_OLN_G), t.lw(l, MANNER_HIP_WL_OLN_B), x_out, L.st2, make_drop(0, 0, 0.f), t.wk.h16a);
}

// cu[n] = n * lp (every position is a row), m_total = {n * lp, n}
__global__ void full_offsets_kernel(int32_t* __restrict__ cu, int32_t* __restrict__ m_total, int64_t n_news, int lp) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= n_news; i += (int64_t)gridDim.x * 256) cu[i] = (int32_t)(i * lp);
  if (blockIdx.x == 0 && threadIdx.x == 0) { m_total[0] = (int32_t)(n_news * lp); m_total[1] = (int32_t)n_news; }
}

}  // namespace
}  // namespace manner

using namespace manner;

extern "C" {

int manner_hip_train_weight_cache(void* const* slots, int32_t* valid, int32_t n_slots) {
  if ((slots == nullptr) != (valid == nullptr) || n_slots < 0) return fail(MANNER_HIP_E_INVALID, "train_weight_cache: slots and valid come together");
  g_next_wcache = WCacheArg{slots, valid, slots ? n_slots : 0};
  return MANNER_HIP_OK;
}

int32_t manner_hip_train_layout_last(void) { return g_layout_last; }
int manner_hip_train_layout_next(int32_t word) {
  if (word < -1 || word > 3) return fail(MANNER_HIP_E_INVALID, "train_layout_next: %d is not a layout word", word);
  g_layout_next = word;
  return MANNER_HIP_OK;
}

size_t manner_hip_train_saved_bytes(const manner_hip_encoder_config* cfg, int64_t n_news, int64_t m_bound, int32_t start_layer) {
  if (!cfg || n_news <= 0 || m_bound <= 0 || start_layer < 0 || start_layer >= cfg->layers || cfg->layers > 64) return 0;
  Bump b(nullptr);
  Saved s;
  plan_saved(b, s, *cfg, n_news, m_bound, start_layer);
  return b.off + 256;
}

size_t manner_hip_train_saved_bytes_for(const manner_hip_encoder_config* cfg, int64_t n_news, int64_t m_bound, int32_t start_layer,
                                        int32_t precision) {
  if (!cfg || n_news <= 0 || m_bound <= 0 || start_layer < 0 || start_layer >= cfg->layers || cfg->layers > 64) return 0;
  const bool lean = choose_lean(cfg, precision, m_bound, choose_attn_path(cfg, precision, false));
  Bump b(nullptr);
  Saved s;
  plan_saved(b, s, *cfg, n_news, m_bound, start_layer, lean);
  return b.off + 256;
}

size_t manner_hip_train_workspace_bytes(const manner_hip_encoder_config* cfg, int64_t m_bound) {
  if (!cfg || m_bound <= 0) return 0;
  Bump b(nullptr);
  Work w;
  plan_work(b, w, *cfg, m_bound);
  return b.off + 256;
}

int manner_hip_dropout_mask(uint64_t seed, uint32_t site, float p, int64_t n, uint8_t* keep, manner_hip_stream_t stream) {
  if (n < 0 || (n > 0 && !keep)) return fail(MANNER_HIP_E_INVALID, "dropout_mask: null pointer");
  if (!(p >= 0.f && p < 1.f)) return fail(MANNER_HIP_E_INVALID, "dropout_mask: p=%f outside [0, 1)", p);
  if (n == 0) return MANNER_HIP_OK;
  const int64_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(mask_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, keep, n,
                     make_drop(seed, site, p));
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

// HF last_hidden_state INCLUDING the padded positions ([n_news, padded_len, H] f32), for the consumers that mix them
// into real tokens (PLMTextEncoder, news_encoder.py:132-171: un-masked batch_first=False attention + un-masked pooler).
struct FullPlan {
  Work wk;
  int32_t *lens, *cu_real, *cu_full, *m_total, *m_real;
  float *esum, *xa, *xb;
  float2* st0;
  LayerSaved L;
};
void plan_full(Bump& b, FullPlan& f, const manner_hip_encoder_config& c, int64_t N, int64_t Mb) {
  const size_t H = c.hidden, I = c.intermediate;
  plan_work(b, f.wk, c, Mb);
  f.lens = b.take<int32_t>(N);
  f.cu_real = b.take<int32_t>(N + 1);
  f.cu_full = b.take<int32_t>(N + 1);
  f.m_total = b.take<int32_t>(4);
  f.m_real = b.take<int32_t>(4);
  f.esum = b.take<float>(Mb * H);
  f.st0 = b.take<float2>(Mb);
  f.xa = b.take<float>(Mb * H);
  f.xb = b.take<float>(Mb * H);
  f.L.x_in = nullptr;
  f.L.qkv = b.take<float>(Mb * 3 * H);
  f.L.ctx = b.take<float>(Mb * H);
  f.L.r1 = b.take<float>(Mb * H);
  f.L.h1 = b.take<float>(Mb * H);
  f.L.inter = b.take<float>(Mb * I);
  f.L.g = b.take<float>(Mb * I);
  f.L.r2 = b.take<float>(Mb * H);
  f.L.st1 = b.take<float2>(Mb);
  f.L.st2 = b.take<float2>(Mb);
  f.L.ml = nullptr;
}

size_t manner_hip_encode_full_workspace_bytes(const manner_hip_encoder_config* cfg, int64_t n_news, int64_t padded_len) {
  if (!cfg || n_news <= 0 || padded_len <= 0) return 0;
  Bump b(nullptr);
  FullPlan f;
  plan_full(b, f, *cfg, n_news, round_up(n_news * padded_len, 256));
  return b.off + 256;
}

int manner_hip_encode_full(const manner_hip_encoder_config* cfg, const float* const* weights, int32_t n_weights, const int64_t* ids,
                           const int64_t* mask, int64_t n_news, int64_t padded_len, int32_t precision, float* hidden,
                           void* workspace, size_t workspace_bytes, int32_t* status, manner_hip_stream_t stream) {
  int rc;
  hipStream_t s = (hipStream_t)stream;
  const int64_t Mb = round_up(n_news * padded_len, 256);
  if ((rc = check_cfg(cfg, n_news, padded_len, Mb, precision, 0))) return rc;
  if (!weights || n_weights != MANNER_HIP_W_EMB_COUNT + cfg->layers * MANNER_HIP_WL_COUNT) return fail(MANNER_HIP_E_INVALID, "encode_full: weight table size");
  for (int i = 0; i < n_weights; ++i)
    if (!weights[i]) return fail(MANNER_HIP_E_INVALID, "encode_full: weight %d is NULL", i);
  if (!ids || !mask || !hidden || !workspace) return fail(MANNER_HIP_E_INVALID, "encode_full: null pointer");
  Bump b(workspace);
  FullPlan f;
  plan_full(b, f, *cfg, n_news, Mb);
  if (b.off > workspace_bytes) return fail(MANNER_HIP_E_WORKSPACE, "encode_full: workspace %zu < %zu bytes", workspace_bytes, b.off);
  Ctx t;
  t.c = cfg; t.w = weights; t.N = n_news; t.Lp = padded_len; t.Mb = Mb; t.prec = precision; t.s = s;
  t.wk = f.wk;
  t.sv.lens = f.lens; t.sv.cu = f.cu_full; t.sv.m_total = f.m_total;
  const int H = cfg->hidden;
  const size_t wide = (size_t)(cfg->intermediate > 3 * H ? cfg->intermediate : 3 * H);
  MANNER_HIP_TRY(hipMemsetAsync(t.wk.zero, 0, wide * sizeof(float), s));
  // real lengths (validated like every other entry point) = key counts; rows = every position
  if ((rc = lengths_and_offsets(mask, n_news, padded_len, f.lens, f.cu_real, f.m_real, Mb, -1, status, s))) return rc;
  hipLaunchKernelGGL(full_offsets_kernel, dim3((unsigned)((n_news + 256) / 256)), dim3(256), 0, s, f.cu_full, f.m_total, n_news, (int)padded_len);
  MANNER_LAUNCH_CHECK();
  const bool roberta = cfg->arch == MANNER_HIP_ARCH_ROBERTA;
  hipLaunchKernelGGL(embed_sum_kernel, dim3((unsigned)(n_news * padded_len)), dim3(256), 0, s, ids, n_news, (int)padded_len, f.cu_full,
                     t.emb(MANNER_HIP_W_WORD_EMB), t.emb(MANNER_HIP_W_POS_EMB), t.emb(MANNER_HIP_W_TYPE_EMB), H, roberta ? cfg->pad_id + 1 : 0,
                     cfg->vocab, cfg->max_pos, f.esum, status, f.lens, roberta ? cfg->pad_id : -1);
  MANNER_LAUNCH_CHECK();
  if ((rc = ln_forward(t, f.esum, t.emb(MANNER_HIP_W_EMB_LN_G), t.emb(MANNER_HIP_W_EMB_LN_B), f.xa, f.st0, make_drop(0, 0, 0.f), t.wk.h16a))) return rc;
  float *x_in = f.xa, *x_out = f.xb;
  for (int l = 0; l < cfg->layers; ++l) {
    float* dst = l + 1 == cfg->layers ? hidden : x_out;          // LayerNorm writes rows < n_news * padded_len: exactly `hidden`
    if ((rc = layer_forward(t, l, f.L, x_in, dst, f.cu_full, f.lens, 0.f, 0.f, 0, true))) return rc;
    float* sw = x_in; x_in = x_out; x_out = sw;
  }
  return MANNER_HIP_OK;
}

// `full` ("full rows", the PLMTextEncoder path): every position of the padded [n_news, padded_len] batch is a row (cu[n] = n * lp),
// the real tokens are the KEYS of a news (klen = lens), no layer is pruned to the [CLS] rows and the output is HF's
// last_hidden_state [n_news * padded_len, H] including the padded positions (which the reference's un-masked consumers mix in).
static int train_forward_impl(const manner_hip_encoder_config* cfg, const float* const* weights, int32_t n_weights,
                              const int64_t* ids, const int64_t* mask, int64_t n_news, int64_t padded_len, int64_t m_bound,
                              int32_t precision, int32_t start_layer, const float* prefix_hidden, float p_hidden, float p_attn,
                              float p_out, uint64_t seed, float* cls_out, void* saved, size_t saved_bytes, void* workspace,
                              size_t workspace_bytes, int32_t* status, manner_hip_stream_t stream, bool full) {
  const WCacheArg wc = take_wcache();                  // first thing: consumed whatever this call goes on to do
  Ctx t;
  int rc;
  hipStream_t s = (hipStream_t)stream;
  // the attention path and the saved-tensor layout are decided BEFORE the buffers are planned (the layout sizes the slots)
  const bool mfma_rule = cfg && choose_attn_path(cfg, precision, full);
  const bool lean_rule = cfg && m_bound > 0 && choose_lean(cfg, precision, m_bound, mfma_rule);
  if ((rc = setup(t, cfg, weights, n_weights, n_news, padded_len, m_bound, precision, start_layer, saved, saved_bytes, workspace,
                  workspace_bytes, s, wc, lean_rule)))
    return rc;
  if (!ids || !mask || !cls_out) return fail(MANNER_HIP_E_INVALID, "train_forward: null pointer");
  t.attn_mfma = mfma_rule;
  t.lean = lean_rule;
  record_attn_path(saved, t.attn_mfma, t.lean);
  g_layout_last = (t.attn_mfma ? 1 : 0) | (t.lean ? 2 : 0);
  if ((start_layer > 0) != (prefix_hidden != nullptr))
    return fail(MANNER_HIP_E_INVALID, "train_forward: prefix_hidden goes with start_layer > 0");
  for (float p : {p_hidden, p_attn, p_out})
    if (!(p >= 0.f && p < 1.f)) return fail(MANNER_HIP_E_INVALID, "train_forward: dropout probability %f outside [0, 1)", p);
  const int H = cfg->hidden, N = (int)n_news;
  (void)N;
  Saved& sv = t.sv;
  if (full) {
    if (start_layer != 0 || m_bound != round_up(n_news * padded_len, 256)) return fail(MANNER_HIP_E_INVALID, "train_full_forward: m_bound must be round_up(n_news * padded_len, 256)");
    // real lengths (validated as everywhere) = key counts; the packed offsets they imply are scratch (wk.dsum / wk.dims)
    if ((rc = lengths_and_offsets(mask, n_news, padded_len, sv.lens, reinterpret_cast<int32_t*>(t.wk.dsum), t.wk.dims + 8, m_bound, -1, status, s))) return rc;
    hipLaunchKernelGGL(full_offsets_kernel, dim3((unsigned)((n_news + 256) / 256)), dim3(256), 0, s, sv.cu, sv.m_total, n_news, (int)padded_len);
    MANNER_LAUNCH_CHECK();
  } else if ((rc = lengths_and_offsets(mask, n_news, padded_len, sv.lens, sv.cu, sv.m_total, m_bound, -1, status, s))) {
    return rc;
  }
  float* x0 = sv.l[start_layer].x_in;
  const unsigned tok_blocks = (unsigned)(n_news * padded_len);
  if (start_layer == 0) {
    const bool roberta = cfg->arch == MANNER_HIP_ARCH_ROBERTA;
    const int pos_offset = roberta ? cfg->pad_id + 1 : 0;
    hipLaunchKernelGGL(embed_sum_kernel, dim3(tok_blocks), dim3(256), 0, s, ids, n_news, (int)padded_len, sv.cu,
                       t.emb(MANNER_HIP_W_WORD_EMB), t.emb(MANNER_HIP_W_POS_EMB), t.emb(MANNER_HIP_W_TYPE_EMB), H, pos_offset,
                       cfg->vocab, cfg->max_pos, sv.esum, status, full ? sv.lens : nullptr, (full && roberta) ? cfg->pad_id : -1);
    MANNER_LAUNCH_CHECK();
    if ((rc = ln_forward(t, sv.esum, t.emb(MANNER_HIP_W_EMB_LN_G), t.emb(MANNER_HIP_W_EMB_LN_B), x0, sv.st0,
                         make_drop(seed, SITE_EMB, p_hidden), t.wk.h16a)))
      return rc;
  } else {
    hipLaunchKernelGGL(pack_rows_kernel, dim3(tok_blocks), dim3(256), 0, s, prefix_hidden, x0, n_news, (int)padded_len, sv.cu, H, 0);
    MANNER_LAUNCH_CHECK();
  }
  if (full) {
    for (int l = 0; l < cfg->layers; ++l) {
      float* x_next = l + 1 < cfg->layers ? sv.l[l + 1].x_in : cls_out;    // the last LayerNorm writes rows < n_news * padded_len: exactly the output
      if ((rc = layer_forward(t, l, sv.l[l], sv.l[l].x_in, x_next, sv.cu, sv.lens, p_hidden, p_attn, seed, true, false))) return rc;
    }
    return MANNER_HIP_OK;
  }
  for (int l = start_layer; l < cfg->layers; ++l) {
    float* x_next = l + 1 < cfg->layers ? sv.l[l + 1].x_in : t.wk.dx;      // the last layer's output is only needed for its [CLS] rows
    if ((rc = layer_forward(t, l, sv.l[l], sv.l[l].x_in, x_next, sv.cu, nullptr, p_hidden, p_attn, seed, l > start_layer || start_layer == 0,
                            l + 1 == cfg->layers)))
      return rc;
  }
  // wk.dx holds the last layer's [CLS] rows, compact: out[n] = dropout(x_L[cu[n]]) (news_encoder.py:34-35), mask index n * H + c
  Ctx c = t;
  c.Mb = round_up(n_news, 256);
  c.sv.m_total = sv.m_total + 1;
  return dropout_add(c, t.wk.dx, nullptr, cls_out, H, make_drop(seed, SITE_CLS, p_out));
}

static int train_backward_impl(const manner_hip_encoder_config* cfg, const float* const* weights, int32_t n_weights,
                               const int64_t* ids, int64_t n_news, int64_t padded_len, int64_t m_bound, int32_t precision,
                               int32_t start_layer, float p_hidden, float p_attn, float p_out, uint64_t seed, const float* grad_cls,
                               void* saved, size_t saved_bytes, float* const* grads, float* grad_prefix, void* workspace,
                               size_t workspace_bytes, manner_hip_stream_t stream, bool full) {
  const WCacheArg wc = take_wcache();                  // first thing: consumed whatever this call goes on to do
  Ctx t;
  int rc;
  hipStream_t s = (hipStream_t)stream;
  // what the forward of this buffer chose (it sized the slots by it): carried by the caller (manner_hip_train_layout_next) or recorded
  // per address by that forward — never re-derived from the environment, whose switches may have changed since
  const int layout = recorded_layout(saved);
  if (layout < 0 && cfg && m_bound > 0)
    return fail(MANNER_HIP_E_INVALID, "train_backward: no layout for this saved buffer (no manner_hip_train_layout_next word and no forward "
                                      "of this process recorded it)");
  const bool mfma_rec = layout > 0 && (layout & 1), lean_rec = layout > 0 && (layout & 2);
  if ((rc = setup(t, cfg, weights, n_weights, n_news, padded_len, m_bound, precision, start_layer, saved, saved_bytes, workspace,
                  workspace_bytes, s, wc, lean_rec)))
    return rc;
  if (!ids || !grad_cls || !grads) return fail(MANNER_HIP_E_INVALID, "train_backward: null pointer");
  t.attn_mfma = mfma_rec;
  t.lean = lean_rec;
  if (t.attn_mfma && (precision == MANNER_HIP_PREC_F32 || full)) return fail(MANNER_HIP_E_INVALID, "train_backward: the forward of this saved buffer ran another precision / row layout");
  if (grad_prefix && start_layer == 0) return fail(MANNER_HIP_E_INVALID, "train_backward: grad_prefix goes with start_layer > 0");
  const int H = cfg->hidden, I = cfg->intermediate;
  Saved& sv = t.sv;
  Work& wk = t.wk;
  auto gemb = [&](int i) { return grads[i]; };
  auto gl = [&](int l, int i) { return grads[MANNER_HIP_W_EMB_COUNT + l * MANNER_HIP_WL_COUNT + i]; };
  // how far down the activation gradient has to travel
  bool emb_grads = false;
  for (int i = 0; i < MANNER_HIP_W_EMB_COUNT; ++i) emb_grads |= gemb(i) != nullptr;
  if (emb_grads && start_layer > 0) return fail(MANNER_HIP_E_INVALID, "train_backward: embedding gradients need start_layer == 0");
  int lowest = cfg->layers;
  for (int l = start_layer; l < cfg->layers; ++l) {
    bool any = false;
    for (int i = 0; i < MANNER_HIP_WL_COUNT; ++i) any |= gl(l, i) != nullptr;
    if (any) { lowest = l; break; }
  }
  for (int l = 0; l < start_layer; ++l)
    for (int i = 0; i < MANNER_HIP_WL_COUNT; ++i)
      if (gl(l, i)) return fail(MANNER_HIP_E_INVALID, "train_backward: gradient requested for layer %d below start_layer %d", l, start_layer);
  const int stop = (emb_grads || grad_prefix) ? start_layer : lowest;     // layers >= stop are walked
  bool ln_pairs = (gemb(MANNER_HIP_W_EMB_LN_G) != nullptr) == (gemb(MANNER_HIP_W_EMB_LN_B) != nullptr);
  for (int l = start_layer; l < cfg->layers; ++l)
    ln_pairs = ln_pairs && (gl(l, MANNER_HIP_WL_ALN_G) != nullptr) == (gl(l, MANNER_HIP_WL_ALN_B) != nullptr) &&
               (gl(l, MANNER_HIP_WL_OLN_G) != nullptr) == (gl(l, MANNER_HIP_WL_OLN_B) != nullptr);
  if (!ln_pairs) return fail(MANNER_HIP_E_INVALID, "train_backward: a LayerNorm's weight and bias gradients come together");
  // One layer, output side first.  `compact` (the last layer): its AO / FFN half ran on the [CLS] rows only (layer_forward), so
  // that half of the backward runs on n_news compact rows (context `c`), is expanded to the token rows in front of the attention
  // backward, and the residual gradient is scattered back into d x_in at the end.  On entry wk.dx holds d x_out (compact or full).
  Ctx cc = t;
  cc.Mb = round_up(n_news, 256);
  cc.sv.m_total = sv.m_total + 1;
  // the side context: same buffers, its own stream and its own scratch — operand conversions go to wk.a16 (nothing on the main
  // stream touches it in a 16-bit backward), column-sum partials behind the LayerNorm backward's share of wk.part
  const char* ws_env = getenv("MANNER_HIP_TRAIN_WGRAD_STREAM");          // A/B switch, read per call
  const char* tr_env0 = getenv("MANNER_HIP_WGRAD_TR");
  SideStream* sd = nullptr;
  std::unique_lock<std::mutex> side_lock;                // held while this call enqueues with the device's side stream
  // the scratch the side stream carves out of buffers the main stream of a 16-bit backward does not use (plan_work): its column-sum
  // partials behind the LayerNorm backward's share of wk.part, the second pair of 16-bit gradient buffers (4 x [Mb, H]) in wk.tmp
  const size_t wide_ = (size_t)(I > 3 * H ? I : 3 * H);
  const bool part_fits = (size_t)LN_BWD_BLOCKS * H * 2 + (size_t)COLSUM_BLOCKS * wide_ <= (size_t)2 * LN_BWD_BLOCKS * wide_;
  const bool tmp_fits = (size_t)4 * t.Mb * H * 2 <= (size_t)t.Mb * wide_ * sizeof(float);
  if (t.dt() != DT_F32 && !full && (!ws_env || atoi(ws_env) != 0) && (!tr_env0 || atoi(tr_env0) != 0) && cfg->hidden % 256 == 0 &&
      cfg->intermediate % 256 == 0 && part_fits && tmp_fits) {
    SideStream& cand = side_stream();
    if (cand.ok) {
      sd = &cand;
      side_lock = std::unique_lock<std::mutex>(cand.mu);
    }
  }
  Ctx cs = t;
  bool pending[4] = {false, false, false, false};
  // on EVERY way out of this function the main stream is ordered behind whatever the side stream still has to do: the caller's
  // stream-ordered allocator and the optimiser see the parameter gradients (and may reuse the buffers) only after that
  struct SideJoin {
    SideStream*& sd; bool (&pending)[4]; hipStream_t main;
    ~SideJoin() {
      if (!sd) return;
      for (int g = 0; g < 4; ++g)
        if (pending[g]) { (void)hipStreamWaitEvent(main, sd->done[g], 0); pending[g] = false; }
    }
  } side_join{sd, pending, s};
  if (sd) {
    cs.s = sd->s;
    cs.wk.b16 = t.wk.a16;
    cs.wk.a16 = nullptr;
    cs.wk.part = t.wk.part + (size_t)2 * LN_BWD_BLOCKS * H;
  }
  // side_begin: the side stream continues behind everything enqueued on the main stream so far; side_end(g): group g is complete on
  // the side stream; main_wait(g): the main stream does not overwrite group g's input before the group has read it
  auto side_begin = [&]() -> int {
    MANNER_HIP_TRY(hipEventRecord(sd->ready, s));
    MANNER_HIP_TRY(hipStreamWaitEvent(sd->s, sd->ready, 0));
    return MANNER_HIP_OK;
  };
  auto side_end = [&](int g) -> int {
    MANNER_HIP_TRY(hipEventRecord(sd->done[g], sd->s));
    pending[g] = true;
    return MANNER_HIP_OK;
  };
  auto main_wait = [&](int g) -> int {
    if (sd && pending[g]) {
      MANNER_HIP_TRY(hipStreamWaitEvent(s, sd->done[g], 0));
      pending[g] = false;
    }
    return MANNER_HIP_OK;
  };
  auto layer_backward = [&](int l, bool compact) -> int {
    Ctx& c = compact ? cc : t;
    const bool aside = sd != nullptr && !compact;              // this layer's parameter gradients go to the side stream
    Ctx& w = aside ? cs : c;
    const int32_t* rowmap = compact ? sv.cu : nullptr;
    LayerSaved& L = sv.l[l];
    // With the side stream every 16-bit gradient tensor a weight-gradient group reads has a buffer of its own (the second pair
    // is carved from wk.tmp, which a 16-bit backward does not use), so that a group has a whole layer of main-stream work before
    // its input is written again — by the SAME producer of the next layer, which is where the main stream waits for it:
    //   d y2 -> h_dy2 (LN2 backward)   d proj -> h_dproj (LN1 backward)   d inter -> b_dinter (FFN2 data gradient)   d qkv -> b_dqkv (attention)
    void* const h_dy2 = wk.h16b;
    void* const h_dproj = sd ? static_cast<void*>(wk.tmp) : wk.h16b;
    void* const b_dinter = wk.big16;
    void* const b_dqkv = sd ? static_cast<void*>(reinterpret_cast<uint16_t*>(wk.tmp) + (size_t)m_bound * H) : wk.big16;
    const bool below = l > stop || emb_grads || grad_prefix;       // is d x_in needed?
    const bool mixed = t.dt() != DT_F32;
    // 16-bit modes: every GEMM-output gradient that is consumed only by GEMMs / column sums exists in the 16-bit type only —
    // d y2 and d proj (H wide, wk.h16b: written by the LayerNorm backward together with d r), d g and d inter (I wide, wk.big16:
    // written by the data-gradient GEMM, then gelu' in place) — and d h1 / d x_in come out of their data-gradient GEMM with the
    // residual gradient already added (EPI_BIAS_RES_F32).  The fp32 mode keeps every tensor in f32.
    const DType g16 = mixed ? t.dt() : DT_F32;
    const float* dy2 = wk.tmp;                               // fp32 mode: d y2 / d proj in f32
    // LN2: dx -> d r2 (wk.dr); r2 = dropout(y2) + h1: d y2 = dropout(d r2), d h1 starts as d r2
    if (mixed) {
      if ((rc = main_wait(0))) return rc;                      // h_dy2 still holds d y2 of the layer above for its FFN2 group
      if ((rc = ln_backward(c, wk.dx, L.r2, L.st2, t.lw(l, MANNER_HIP_WL_OLN_G), wk.dr, gl(l, MANNER_HIP_WL_OLN_G), gl(l, MANNER_HIP_WL_OLN_B),
                            make_drop(seed, layer_site(l, SITE_FFN), p_hidden), h_dy2, rowmap))) return rc;
    } else {
      if ((rc = ln_backward(c, wk.dx, L.r2, L.st2, t.lw(l, MANNER_HIP_WL_OLN_G), wk.dr, gl(l, MANNER_HIP_WL_OLN_G), gl(l, MANNER_HIP_WL_OLN_B)))) return rc;
      if ((rc = dropout_add(c, wk.dr, nullptr, wk.tmp, H, make_drop(seed, layer_site(l, SITE_FFN), p_hidden), nullptr, rowmap))) return rc;
    }
    const void* dy2_any = mixed ? (const void*)h_dy2 : (const void*)dy2;
    const bool ff2_grads = gl(l, MANNER_HIP_WL_FF2_B) || gl(l, MANNER_HIP_WL_FF2_W);
    if (aside && ff2_grads && (rc = side_begin())) return rc;
    if (gl(l, MANNER_HIP_WL_FF2_B) && (rc = bias_grad(w, dy2_any, g16, H, gl(l, MANNER_HIP_WL_FF2_B)))) return rc;
    if (gl(l, MANNER_HIP_WL_FF2_W) && (rc = linear_wgrad(w, dy2_any, g16, L.g, g16, gl(l, MANNER_HIP_WL_FF2_W), H, I, 0))) return rc;
    if (aside && ff2_grads && (rc = side_end(0))) return rc;
    const void* dinter_any;
    if ((rc = main_wait(1))) return rc;                        // b_dinter still holds d inter of the layer above for its FFN1 group
    if (mixed) {
      if (t.lean && !compact) {                                                               // ... with the 16-bit saved pre-activation
        const void* wt;
        if ((rc = weight16(c, t.lw(l, MANNER_HIP_WL_FF2_W), H, I, true, wk.b16, &wt))) return rc;
        if ((rc = gemm_tn_gelu_grad16(t.dt(), h_dy2, wt, wk.zero, L.inter, b_dinter, c.Mb, I, H, c.sv.m_total, s))) return rc;
      } else if (gelu_fused_enabled() && gemm_gelu_fusable(t.dt(), c.Mb, I, H)) {                // d inter = (d y2 . W2) * gelu'(inter): one launch
        const void* wt;
        if ((rc = weight16(c, t.lw(l, MANNER_HIP_WL_FF2_W), H, I, true, wk.b16, &wt))) return rc;
        if ((rc = gemm_tn_gelu_grad(t.dt(), h_dy2, wt, wk.zero, L.inter, b_dinter, c.Mb, I, H, c.sv.m_total, s))) return rc;
      } else {
        if ((rc = linear_dgrad(c, nullptr, t.lw(l, MANNER_HIP_WL_FF2_W), b_dinter, H, I, h_dy2, t.dt()))) return rc;          // d g (16-bit)
        if ((rc = gelu16(c, L.inter, b_dinter, I, 1))) return rc;                                                         // d inter, in place
      }
      dinter_any = b_dinter;
    } else {
      if ((rc = linear_dgrad(c, dy2, t.lw(l, MANNER_HIP_WL_FF2_W), wk.dbig, H, I))) return rc;                           // d g
      hipLaunchKernelGGL(gelu_kernel, dim3(c.ew_grid(I)), dim3(256), 0, s, L.inter, wk.dbig, wk.dbig, I, c.sv.m_total, 1, Out16{nullptr, 0});   // d inter
      MANNER_LAUNCH_CHECK();
      dinter_any = wk.dbig;
    }
    const bool ff1_grads = gl(l, MANNER_HIP_WL_FF1_B) || gl(l, MANNER_HIP_WL_FF1_W);
    if (aside && ff1_grads && (rc = side_begin())) return rc;
    if (gl(l, MANNER_HIP_WL_FF1_B) && (rc = bias_grad(w, dinter_any, g16, I, gl(l, MANNER_HIP_WL_FF1_B)))) return rc;
    if (gl(l, MANNER_HIP_WL_FF1_W) && (rc = linear_wgrad(w, dinter_any, g16, L.h1, DT_F32, gl(l, MANNER_HIP_WL_FF1_W), I, H, 1))) return rc;
    if (aside && ff1_grads && (rc = side_end(1))) return rc;
    {
      bool fused = false;                                    // d h1 = d inter . W1 + d r2
      if ((rc = linear_dgrad(c, mixed ? nullptr : wk.dbig, t.lw(l, MANNER_HIP_WL_FF1_W), mixed ? wk.dx : wk.tmp, I, H, mixed ? b_dinter : nullptr,
                             DT_F32, mixed ? wk.dr : nullptr, &fused)))
        return rc;
      if (!fused && (rc = add_rows(c, mixed ? wk.dx : wk.tmp, wk.dr, wk.dx, H))) return rc;
    }
    // LN1: d h1 -> d r1 (wk.dr); r1 = dropout(proj) + x_in
    if (mixed) {
      if ((rc = main_wait(2))) return rc;                      // h_dproj still holds d proj of the layer above for its out-projection group
      if ((rc = ln_backward(c, wk.dx, L.r1, L.st1, t.lw(l, MANNER_HIP_WL_ALN_G), wk.dr, gl(l, MANNER_HIP_WL_ALN_G), gl(l, MANNER_HIP_WL_ALN_B),
                            make_drop(seed, layer_site(l, SITE_PROJ), p_hidden), h_dproj, rowmap))) return rc;
    } else {
      if ((rc = ln_backward(c, wk.dx, L.r1, L.st1, t.lw(l, MANNER_HIP_WL_ALN_G), wk.dr, gl(l, MANNER_HIP_WL_ALN_G), gl(l, MANNER_HIP_WL_ALN_B)))) return rc;
      if ((rc = dropout_add(c, wk.dr, nullptr, wk.tmp, H, make_drop(seed, layer_site(l, SITE_PROJ), p_hidden), nullptr, rowmap))) return rc;    // d proj
    }
    const void* dproj_any = mixed ? (const void*)h_dproj : (const void*)wk.tmp;
    const bool ao_grads = gl(l, MANNER_HIP_WL_AO_B) || gl(l, MANNER_HIP_WL_AO_W);
    if (aside && ao_grads && (rc = side_begin())) return rc;
    if (gl(l, MANNER_HIP_WL_AO_B) && (rc = bias_grad(w, dproj_any, g16, H, gl(l, MANNER_HIP_WL_AO_B)))) return rc;
    if (gl(l, MANNER_HIP_WL_AO_W)) {
      const void* ctx_rows = L.ctx;
      DType ctx_dt = t.lean ? g16 : DT_F32;              // lean: ctx exists in the 16-bit type only — the operand as it lies, no conversion
      if (compact) {                                     // the [CLS] rows of ctx, gathered as in the forward
        if (!t.lean) hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)n_news), dim3(256), 0, s, L.ctx, sv.cu, H, wk.dbig);
        else if (t.dt() == DT_F16) hipLaunchKernelGGL(gather_rows16_kernel<f16_t>, dim3((unsigned)n_news), dim3(256), 0, s, reinterpret_cast<const f16_t*>(L.ctx), sv.cu, H, wk.dbig);
        else hipLaunchKernelGGL(gather_rows16_kernel<bf16_t>, dim3((unsigned)n_news), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(L.ctx), sv.cu, H, wk.dbig);
        MANNER_LAUNCH_CHECK();
        ctx_rows = wk.dbig;
        ctx_dt = DT_F32;
      }
      if ((rc = linear_wgrad(w, dproj_any, g16, ctx_rows, ctx_dt, gl(l, MANNER_HIP_WL_AO_W), H, H, 2))) return rc;
    }
    if (aside && ao_grads && (rc = side_end(2))) return rc;
    const bool qkv_w = gl(l, MANNER_HIP_WL_Q_W) || gl(l, MANNER_HIP_WL_K_W) || gl(l, MANNER_HIP_WL_V_W) || gl(l, MANNER_HIP_WL_Q_B) ||
                       gl(l, MANNER_HIP_WL_K_B) || gl(l, MANNER_HIP_WL_V_B);
    if (!below && !qkv_w) return MANNER_HIP_OK;
    if (compact) {
      // d ctx of the [CLS] rows -> token rows (every other row of d ctx is zero)
      if ((rc = linear_dgrad(c, mixed ? nullptr : wk.tmp, t.lw(l, MANNER_HIP_WL_AO_W), wk.dqkv, H, H, mixed ? h_dproj : nullptr))) return rc;
      MANNER_HIP_TRY(hipMemsetAsync(wk.dx, 0, (size_t)m_bound * H * sizeof(float), s));
      hipLaunchKernelGGL(cls_bwd_kernel, dim3((unsigned)n_news), dim3(256), 0, s, wk.dqkv, sv.cu, H, wk.dx, make_drop(0, 0, 0.f));
      MANNER_LAUNCH_CHECK();
    } else if (t.lean) {                                 // d ctx in the 16-bit type, straight into the attention backward's operand buffer
      if ((rc = linear_dgrad(t, nullptr, t.lw(l, MANNER_HIP_WL_AO_W), wk.h16a, H, H, h_dproj, t.dt()))) return rc;
    } else if ((rc = linear_dgrad(t, mixed ? nullptr : wk.tmp, t.lw(l, MANNER_HIP_WL_AO_W), wk.dx, H, H, mixed ? h_dproj : nullptr))) {   // d ctx
      return rc;
    }
    const Drop da = make_drop(seed, layer_site(l, SITE_ATTN), p_attn);
    if ((rc = main_wait(3))) return rc;                        // b_dqkv / wk.dqkv still hold d qkv of the layer above for its Q|K|V group
    if (t.mfma_attn()) {
      // matrix-pipe backward: D = dctx . ctx and the 16-bit copy of dctx (wk.h16a is free in the backward), then d q and d k / d v
      // lean: ctx (and, but for the compact last layer, d ctx) in the 16-bit type; d qkv as 16-bit rows only
      if (t.lean) {
        if ((rc = attn_train_mfma_backward(t.dt(), L.qkv, compact ? (const void*)wk.dx : (const void*)wk.h16a, !compact, L.ctx, true, L.ml, nullptr, b_dqkv,
                                           wk.h16a, wk.dsum, sv.cu, n_news, cfg->heads, H, (int)padded_len, da, m_bound, sv.m_total, s)))
          return rc;
      } else if ((rc = attn_train_mfma_backward(t.dt(), L.qkv, wk.dx, false, L.ctx, false, L.ml, wk.dqkv, b_dqkv, wk.h16a, wk.dsum, sv.cu, n_news, cfg->heads, H,
                                                (int)padded_len, da, m_bound, sv.m_total, s)))
        return rc;
    } else {
#define MANNER_ATTN_BWD(AT_, HPB_)                                                                                              \
  do {                                                                                                                            \
    const dim3 ag((unsigned)(cfg->heads / HPB_), (unsigned)n_news);                                                               \
    hipLaunchKernelGGL((attn_train_bwd_q_kernel<AT_, HPB_>), ag, dim3(AT_), 0, s, L.qkv, wk.dx, L.ctx, L.ml, wk.dqkv, wk.dsum, sv.cu, cfg->heads, H, da, t.o16(b_dqkv), full ? sv.lens : nullptr);  \
    hipLaunchKernelGGL((attn_train_bwd_kv_kernel<AT_, HPB_>), ag, dim3(AT_), 0, s, L.qkv, wk.dx, L.ml, wk.dsum, wk.dqkv, sv.cu, cfg->heads, H, da, t.o16(b_dqkv), full ? sv.lens : nullptr); \
  } while (0)
    MANNER_ATTN_DISPATCH(padded_len, cfg->heads, MANNER_ATTN_BWD);
#undef MANNER_ATTN_BWD
    MANNER_LAUNCH_CHECK();
    }
    if (qkv_w) {
      // the packed projection's gradients go to the three parameters' gradients in place: the fixed-order sums of the column /
      // slice partials write three destinations (round 4: 6 device copies per layer before)
      const SumDst db3{{gl(l, MANNER_HIP_WL_Q_B), gl(l, MANNER_HIP_WL_Q_B + 2), gl(l, MANNER_HIP_WL_Q_B + 4)}, H};
      const SumDst dw3{{gl(l, MANNER_HIP_WL_Q_W), gl(l, MANNER_HIP_WL_Q_W + 2), gl(l, MANNER_HIP_WL_Q_W + 4)}, H * H};
      Ctx& wq = sd ? cs : t;                                  // (the compact last layer's attention half runs on the token rows too)
      if (sd && (rc = side_begin())) return rc;
      if ((db3.p[0] || db3.p[1] || db3.p[2]) &&
          (rc = t.lean ? bias_grad(wq, b_dqkv, g16, 3 * H, nullptr, &db3) : bias_grad(wq, wk.dqkv, DT_F32, 3 * H, nullptr, &db3))) return rc;
      bool in_place = false;
      if ((rc = linear_wgrad(wq, mixed ? (const void*)b_dqkv : (const void*)wk.dqkv, g16, L.x_in, DT_F32, wk.dw, 3 * H, H, 3, &dw3, &in_place))) return rc;
      if (!in_place)
        for (int k = 0; k < 3; ++k)
          if (gl(l, MANNER_HIP_WL_Q_W + 2 * k))
            MANNER_HIP_TRY(hipMemcpyAsync(gl(l, MANNER_HIP_WL_Q_W + 2 * k), wk.dw + (size_t)k * H * H, (size_t)H * H * sizeof(float), hipMemcpyDeviceToDevice, wq.s));
      if (sd && (rc = side_end(3))) return rc;
    }
    if (!below) return MANNER_HIP_OK;
    // W^T of the packed Q | K | V weight: from the caller's cache for a frozen layer (then neither the pack nor the transpose runs)
    const int wiq = MANNER_HIP_W_EMB_COUNT + l * MANNER_HIP_WL_COUNT + MANNER_HIP_WL_Q_W;
    const bool qkv_t_cached = mixed && t.slot(2 * wiq + 1) && t.wc_valid[2 * wiq + 1];
    if (!qkv_t_cached && (rc = pack_qkv_weights(t, l))) return rc;
    if (compact) {                                       // d x_in = d qkv . W on every row, + d r1 on the [CLS] rows
      if ((rc = linear_dgrad(t, wk.dqkv, wk.wcat, wk.dx, 3 * H, H, mixed ? b_dqkv : nullptr, DT_F32, nullptr, nullptr, wiq))) return rc;
      hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)n_news), dim3(256), 0, s, wk.dr, sv.cu, H, wk.dx);
      MANNER_LAUNCH_CHECK();
      return MANNER_HIP_OK;
    }
    bool fused = false;                                  // d x_in = d qkv . W + d r1
    if ((rc = linear_dgrad(t, wk.dqkv, wk.wcat, mixed ? wk.dx : wk.tmp, 3 * H, H, mixed ? b_dqkv : nullptr, DT_F32, mixed ? wk.dr : nullptr, &fused, wiq)))
      return rc;
    return fused ? MANNER_HIP_OK : add_rows(t, mixed ? wk.dx : wk.tmp, wk.dr, wk.dx, H);
  };
  if (full) {
    // d last_hidden_state arrives for every row; no layer was pruned
    MANNER_HIP_TRY(hipMemcpyAsync(wk.dx, grad_cls, (size_t)n_news * padded_len * H * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (int l = cfg->layers - 1; l >= stop; --l)
      if ((rc = layer_backward(l, false))) return rc;
  } else if (cfg->layers - 1 >= stop) {
    // d x_L of the [CLS] rows, compact: the backward of out[n] = dropout(x_L[cu[n]])
    if ((rc = dropout_add(cc, grad_cls, nullptr, wk.dx, H, make_drop(seed, SITE_CLS, p_out)))) return rc;
    if ((rc = layer_backward(cfg->layers - 1, true))) return rc;
  }
  for (int l = cfg->layers - 2; l >= stop && !full; --l)
    if ((rc = layer_backward(l, false))) return rc;
  const unsigned tok_blocks = (unsigned)(n_news * padded_len);
  if (grad_prefix) {
    hipLaunchKernelGGL(pack_rows_kernel, dim3(tok_blocks), dim3(256), 0, s, grad_prefix, wk.dx, n_news, (int)padded_len, sv.cu, H, 1);
    MANNER_LAUNCH_CHECK();
  }
  if (emb_grads) {
    // x0 = dropout(LN(esum))
    if ((rc = dropout_add(t, wk.dx, nullptr, wk.dx, H, make_drop(seed, SITE_EMB, p_hidden)))) return rc;
    if ((rc = ln_backward(t, wk.dx, sv.esum, sv.st0, t.emb(MANNER_HIP_W_EMB_LN_G), wk.dr, gemb(MANNER_HIP_W_EMB_LN_G), gemb(MANNER_HIP_W_EMB_LN_B)))) return rc;
    const int pos_offset = cfg->arch == MANNER_HIP_ARCH_ROBERTA ? cfg->pad_id + 1 : 0;
    if (gemb(MANNER_HIP_W_WORD_EMB)) MANNER_HIP_TRY(hipMemsetAsync(gemb(MANNER_HIP_W_WORD_EMB), 0, (size_t)cfg->vocab * H * sizeof(float), s));
    if (gemb(MANNER_HIP_W_POS_EMB)) MANNER_HIP_TRY(hipMemsetAsync(gemb(MANNER_HIP_W_POS_EMB), 0, (size_t)cfg->max_pos * H * sizeof(float), s));
    if (gemb(MANNER_HIP_W_WORD_EMB) || gemb(MANNER_HIP_W_POS_EMB)) {
      hipLaunchKernelGGL(embed_bwd_kernel, dim3(tok_blocks), dim3(256), 0, s, ids, n_news, (int)padded_len, sv.cu, wk.dr, H, pos_offset,
                         cfg->vocab, cfg->max_pos, gemb(MANNER_HIP_W_WORD_EMB), gemb(MANNER_HIP_W_POS_EMB), full ? sv.lens : nullptr,
                         (full && cfg->arch == MANNER_HIP_ARCH_ROBERTA) ? cfg->pad_id : -1, cfg->pad_id,
                         cfg->arch == MANNER_HIP_ARCH_ROBERTA ? cfg->pad_id : -1);
      MANNER_LAUNCH_CHECK();
    }
    if (gemb(MANNER_HIP_W_TYPE_EMB)) {
      MANNER_HIP_TRY(hipMemsetAsync(gemb(MANNER_HIP_W_TYPE_EMB), 0, (size_t)cfg->type_vocab * H * sizeof(float), s));
      if ((rc = bias_grad(t, wk.dr, DT_F32, H, gemb(MANNER_HIP_W_TYPE_EMB)))) return rc;      // every token is of type 0
    }
  }
  return MANNER_HIP_OK;
}

int manner_hip_train_forward(const manner_hip_encoder_config* cfg, const float* const* weights, int32_t n_weights,
                             const int64_t* ids, const int64_t* mask, int64_t n_news, int64_t padded_len, int64_t m_bound,
                             int32_t precision, int32_t start_layer, const float* prefix_hidden, float p_hidden, float p_attn,
                             float p_out, uint64_t seed, float* cls_out, void* saved, size_t saved_bytes, void* workspace,
                             size_t workspace_bytes, int32_t* status, manner_hip_stream_t stream) {
  return train_forward_impl(cfg, weights, n_weights, ids, mask, n_news, padded_len, m_bound, precision, start_layer, prefix_hidden, p_hidden,
                            p_attn, p_out, seed, cls_out, saved, saved_bytes, workspace, workspace_bytes, status, stream, false);
}

int manner_hip_train_backward(const manner_hip_encoder_config* cfg, const float* const* weights, int32_t n_weights,
                              const int64_t* ids, int64_t n_news, int64_t padded_len, int64_t m_bound, int32_t precision,
                              int32_t start_layer, float p_hidden, float p_attn, float p_out, uint64_t seed, const float* grad_cls,
                              void* saved, size_t saved_bytes, float* const* grads, float* grad_prefix, void* workspace,
                              size_t workspace_bytes, manner_hip_stream_t stream) {
  return train_backward_impl(cfg, weights, n_weights, ids, n_news, padded_len, m_bound, precision, start_layer, p_hidden, p_attn, p_out, seed,
                             grad_cls, saved, saved_bytes, grads, grad_prefix, workspace, workspace_bytes, stream, false);
}

int manner_hip_train_full_forward(const manner_hip_encoder_config* cfg, const float* const* weights, int32_t n_weights,
                                  const int64_t* ids, const int64_t* mask, int64_t n_news, int64_t padded_len, int32_t precision,
                                  float p_hidden, float p_attn, uint64_t seed, float* hidden, void* saved, size_t saved_bytes,
                                  void* workspace, size_t workspace_bytes, int32_t* status, manner_hip_stream_t stream) {
  if (n_news <= 0 || padded_len <= 0) return fail(MANNER_HIP_E_INVALID, "train_full_forward: empty batch");
  return train_forward_impl(cfg, weights, n_weights, ids, mask, n_news, padded_len, round_up(n_news * padded_len, 256), precision, 0, nullptr,
                            p_hidden, p_attn, 0.f, seed, hidden, saved, saved_bytes, workspace, workspace_bytes, status, stream, true);
}

int manner_hip_train_full_backward(const manner_hip_encoder_config* cfg, const float* const* weights, int32_t n_weights,
                                   const int64_t* ids, int64_t n_news, int64_t padded_len, int32_t precision, float p_hidden,
                                   float p_attn, uint64_t seed, const float* grad_hidden, void* saved, size_t saved_bytes,
                                   float* const* grads, void* workspace, size_t workspace_bytes, manner_hip_stream_t stream) {
  if (n_news <= 0 || padded_len <= 0) return fail(MANNER_HIP_E_INVALID, "train_full_backward: empty batch");
  return train_backward_impl(cfg, weights, n_weights, ids, n_news, padded_len, round_up(n_news * padded_len, 256), precision, 0, p_hidden,
                             p_attn, 0.f, seed, grad_hidden, saved, saved_bytes, grads, nullptr, workspace, workspace_bytes, stream, true);
}

int manner_hip_late_fusion_train_forward(const float* hist, const int64_t* hist_off, const float* cand, const int64_t* cand_off,
                                         int64_t B, int32_t D, float* user, float* scores, manner_hip_stream_t stream) {
  if (B == 0) return MANNER_HIP_OK;
  if (!hist || !hist_off || !cand || !cand_off || !user || !scores || B < 0 || D <= 0 || D > 8192)
    return fail(MANNER_HIP_E_INVALID, "late_fusion_train_forward: bad argument");
  hipLaunchKernelGGL(lf_train_fwd_kernel, dim3((unsigned)B), dim3(256), D * sizeof(float), (hipStream_t)stream, hist, hist_off, cand,
                     cand_off, D, user, scores);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_late_fusion_train_backward(const float* grad_scores, const float* user, const int64_t* hist_off, const float* cand,
                                          const int64_t* cand_off, int64_t B, int32_t D, float* grad_hist, float* grad_cand,
                                          manner_hip_stream_t stream) {
  if (B == 0) return MANNER_HIP_OK;
  if (!grad_scores || !user || !hist_off || !cand || !cand_off || !grad_hist || !grad_cand || B < 0 || D <= 0)
    return fail(MANNER_HIP_E_INVALID, "late_fusion_train_backward: bad argument");
  hipLaunchKernelGGL(lf_train_bwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, grad_scores, user, hist_off, cand,
                     cand_off, D, grad_hist, grad_cand);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_dot_backward(const float* grad_out, const float* user, const float* cand, int64_t B, int64_t C, int32_t D,
                            int64_t cand_stride_b, int64_t cand_stride_d, int64_t cand_stride_c, float* grad_user,
                            float* grad_cand, manner_hip_stream_t stream) {
  if (B == 0 || C == 0) return MANNER_HIP_OK;
  if (!grad_out || !user || !cand || !grad_user || !grad_cand || B < 0 || C < 0 || D <= 0)
    return fail(MANNER_HIP_E_INVALID, "dot_backward: bad argument");
  hipLaunchKernelGGL(dot_bwd_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, grad_out, user, cand, C, D, cand_stride_b,
                     cand_stride_d, cand_stride_c, grad_user, grad_cand);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_train_loss(const float* scores, const float* labels, const int64_t* cand_off, int64_t B, int32_t mode,
                          float temperature, int64_t c_max, float* losses, float* loss_and_scale, float* grad_scores,
                          manner_hip_stream_t stream) {
  if (!scores || !labels || !cand_off || !losses || !loss_and_scale || !grad_scores || B <= 0 || mode < 0 || mode > 1 ||
      !(temperature > 0.f) || (mode == 1 && c_max < 1))
    return fail(MANNER_HIP_E_INVALID, "train_loss: bad argument");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(train_loss_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, scores, labels, cand_off, B, (int)mode,
                     1.0f / temperature, c_max, 1.17549435e-38f, losses, grad_scores);
  MANNER_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, losses, B, (int)mode, loss_and_scale);
  MANNER_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_scale_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, grad_scores, losses, cand_off, B, (int)mode,
                     loss_and_scale);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

size_t manner_hip_supcon_embeddings_workspace_bytes(int64_t N) { return N > 0 ? (size_t)N * N * sizeof(float) + 256 : 0; }

int manner_hip_supcon_embeddings(const float* emb, const int64_t* labels, int64_t N, int32_t D, float temperature, float* losses,
                                 float* loss_and_scale, float* grad_emb, void* workspace, size_t workspace_bytes, manner_hip_stream_t stream) {
  if (!emb || !labels || !losses || !loss_and_scale || !grad_emb || !workspace || N <= 0 || N > 46000 || D <= 0 || !(temperature > 0.f))
    return fail(MANNER_HIP_E_INVALID, "supcon_embeddings: bad argument");
  if (workspace_bytes < manner_hip_supcon_embeddings_workspace_bytes(N)) return fail(MANNER_HIP_E_WORKSPACE, "supcon_embeddings: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* mat = static_cast<float*>(workspace);
  int32_t* flags = reinterpret_cast<int32_t*>(mat + (size_t)N * N);
  MANNER_HIP_TRY(hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), s));
  hipLaunchKernelGGL(a_sim_kernel, dim3((unsigned)N), dim3(256), 0, s, emb, N, D, 1.0f / temperature, mat);
  MANNER_LAUNCH_CHECK();
  hipLaunchKernelGGL(a_loss_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, mat, labels, N, 1.17549435e-38f, losses, flags);
  MANNER_LAUNCH_CHECK();
  hipLaunchKernelGGL(a_reduce_kernel, dim3(1), dim3(256), 0, s, losses, N, flags, loss_and_scale);
  MANNER_LAUNCH_CHECK();
  hipLaunchKernelGGL(a_grad_kernel, dim3((unsigned)N), dim3(256), 0, s, mat, losses, loss_and_scale, emb, N, D, 1.0f / temperature, grad_emb);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

}  // extern "C"
