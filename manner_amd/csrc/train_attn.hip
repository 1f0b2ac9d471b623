// Training attention on the matrix pipe (HF BertSelfAttention in train() mode, transformers/models/bert/
// modeling_bert.py:111-140: softmax(q k^T / 8) -> dropout on the PROBABILITIES -> . v, and its backward), for the "16-mixed"
// modes of the training path: 16-bit Q | K | V operands (f16 or bf16), f32 accumulation, f32 softmax statistics, f32
// outputs (+ 16-bit copies for the GEMMs that consume them).  The fp32 mode keeps the VALU kernels of train.hip.
//
// One 64-lane wave per (news, head), no workgroup barriers — the structure of attention.hip's inference kernel:
//   * a 32-row tile of a 16-bit matrix is fetched as 16-byte pieces of whole 128-byte row segments, passed through a
//     wave-private 4 KiB LDS slab (XOR-swizzled chunks) and read back as MFMA operand fragments: lane (rr, h) holds
//     row rr, features 16 ks + 8 h .. + 7;
//   * a row-major LDS image of a matrix (filled by LDS-DMA) is read TRANSPOSED with ds_read_b64_tr_b16 as the A
//     operand of a product whose B operand is an accumulator tile converted in place (k-order permuted alike on both
//     sides): out^T[d][col] += sum_r image[r][d] * acc[r][col].
// forward      S^T = K Q^T (lane = query): online softmax over 32-key tiles, dropout applied to the un-normalised
//              probabilities that feed O^T = V^T Pd^T while the row sum keeps the undropped ones; {row max, row sum} saved.
// backward-q   (lane = query)  S^T = K Q^T, dP^T = V dO^T, dS^T = P^T o (M dP^T - D) / 8, dQ^T += K^T dS^T
// backward-kv  (lane = key)    S = Q K^T, dP = dO V^T, dV^T += dO^T (M P), dK^T += Q^T dS         — no atomics anywhere
// with D_i = dctx_i . ctx_i (= sum_j dP_ij P_ij; ctx carries the same dropout) from a small pre-pass that also writes the
// 16-bit copy of dctx.  Dropout bits are regenerated from (seed, site, (row * heads + head) * 256 + key) in all three.
#include <math.h>

#include "train_common.h"

namespace manner {
namespace {

constexpr float kC = 0.125f * 1.44269504088896340736f;   // head_dim^-0.5 * log2(e)

// 32 rows x 64 features (128 B per row) starting at row0, rows clamped to L - 1 (replicas: finite, masked or unused):
// 4 x 16 bytes per lane, source chunk XOR-swizzled so that the lane-linear slab image is conflict-free to read
template <typename TE>
__device__ __forceinline__ void load_tile(const TE* __restrict__ base, size_t ld, int row0, int L, f32x4 (&t)[4]) {
  const int lane = threadIdx.x & 63, r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = min(row0 + 8 * i + r8, L - 1);
    t[i] = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + 8 * (c8 ^ (((8 * i + r8) >> 1) & 7)));
  }
}
template <typename TE>
__device__ __forceinline__ void tile_to_frags(char* slab, const f32x4 (&t)[4], typename E16<TE>::v8 (&f)[4]) {
  typedef typename E16<TE>::v8 e16x8;
  const int lane = threadIdx.x & 63, r8 = lane >> 3, c8 = lane & 7, rr = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(slab + (8 * i + r8) * 128 + (c8 << 4)) = t[i];
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const e16x8*>(slab + rr * 128 + (((2 * ks + h) ^ ((rr >> 1) & 7)) << 4));
  __builtin_amdgcn_wave_barrier();
}
template <typename TE>
__device__ __forceinline__ void fetch_frags(const TE* __restrict__ base, size_t ld, int row0, int L, char* slab, typename E16<TE>::v8 (&f)[4]) {
  f32x4 t[4];
  load_tile<TE>(base, ld, row0, L, t);
  tile_to_frags<TE>(slab, t, f);
}
// row-major image of rows 0 .. 32 nkt - 1 (clamped to L - 1) by LDS-DMA: 8 rows x 128 B per instruction, no registers
template <typename TE, int NKT>
__device__ __forceinline__ void dma_image(const TE* __restrict__ base, size_t ld, int L, char* img) {
  const int lane = threadIdx.x & 63, r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = min(32 * kt + 8 * i + r8, L - 1);
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(base + (size_t)row * ld + 8 * c8), LDS_PTR(img + (32 * kt + 8 * i) * 128), 16, 0, 0);
    }
}
__device__ __forceinline__ int tr_lane_base() {
  const int lane = threadIdx.x & 63, gi = lane & 15, h = lane >> 5;
  return ((gi >> 2) * 64 + 16 * ((lane >> 4) & 1) + 4 * (gi & 3)) * 2 + (4 * h) * 128;
}
// A operand = image^T: features 32 dt .. + 31 x the 16 k-slots of step s2 of tile `tile` (rows 32 tile + 16 s2 .. + 15, permuted
// exactly as acc_to_b permutes them)
template <typename TE>
__device__ __forceinline__ typename E16<TE>::v8 tr_frag(const char* img, int tr_base, int tile, int s2, int dt) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const char* a0 = img + tr_base + (32 * tile + 16 * s2) * 128 + (32 * dt) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 8 * 128));
  const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(typename E16<TE>::v8, both);
}
// acc element e of lane (rr, h) = [row (e & 3) + 8 (e >> 2) + 4 h][column rr]
__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// rows row0 .. row0 + 31 (< L) of an accumulator pair o[dt][4 g + e] = value at [row rr][feature 32 dt + 8 g + 4 h + e], scaled:
// f32 rows to dst32 (+ col0, row stride ld32) and, when dst16 != NULL, 16-bit rows to dst16 (row stride ld16) — through the
// slab so that every global store is a 16-byte piece of a whole 128-byte row segment
template <typename TE>
__device__ __forceinline__ void store_rows(char* slab, const f32x16 (&o)[2], float scale, int row0, int L, float* __restrict__ dst32,
                                           size_t ld32, TE* __restrict__ dst16, size_t ld16) {
  typedef typename E16<TE>::v4 e16x4;
  const int lane = threadIdx.x & 63, r8 = lane >> 3, c8 = lane & 7, rr = lane & 31, h = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    if (!dst32) break;                                       // round 5: 16-bit-only outputs (the f32 rows have no reader)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(slab + rr * 128 + (((2 * g + h) ^ ((rr >> 1) & 7)) << 4)) =
          f32x4{o[dt][4 * g] * scale, o[dt][4 * g + 1] * scale, o[dt][4 * g + 2] * scale, o[dt][4 * g + 3] * scale};
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + r8;
      const f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * 128 + (c8 << 4));
      if (row0 + row < L) *reinterpret_cast<f32x4*>(dst32 + (size_t)(row0 + row) * ld32 + 32 * dt + 4 * (c8 ^ ((row >> 1) & 7))) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (!dst16) return;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<e16x4*>(slab + rr * 128 + (((4 * dt + g) ^ ((rr >> 1) & 7)) << 4) + 8 * h) =
          e16x4{(TE)(o[dt][4 * g] * scale), (TE)(o[dt][4 * g + 1] * scale), (TE)(o[dt][4 * g + 2] * scale), (TE)(o[dt][4 * g + 3] * scale)};
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + r8;
    const f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * 128 + (c8 << 4));
    if (row0 + row < L) *reinterpret_cast<f32x4*>(dst16 + (size_t)(row0 + row) * ld16 + 8 * (c8 ^ ((row >> 1) & 7))) = v;
  }
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.f;
  return z;
}

// ------------------------------------------------------------------------------------------------ forward
template <typename TE, int NKT>
__device__ __forceinline__ void fwd_wave(const TE* __restrict__ qkv, float* __restrict__ ctx, TE* __restrict__ ctx16,
                                         float2* __restrict__ ml, int tok0, int L, int H, int heads, int head, char* vl, const Drop& drop) {
  typedef typename E16<TE>::v8 e16x8;
  const int lane = threadIdx.x & 63, rr = lane & 31, h = lane >> 5;
  char* ol = vl + NKT * 4096;
  const size_t ld = 3 * (size_t)H;
  const TE* Qb = qkv + (size_t)tok0 * ld + head * 64;
  const TE* Kb = Qb + H;
  const TE* Vb = Qb + 2 * H;
  dma_image<TE, NKT>(Vb, ld, L, vl);
  e16x8 kf[NKT][4];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) fetch_frags<TE>(Kb, ld, 32 * kt, L, ol, kf[kt]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the V image (LDS-DMA is not tracked by the compiler)
  const int tr_base = tr_lane_base();
#pragma unroll
  for (int qb = 0; qb < NKT; ++qb) {
    if (32 * qb >= L) break;
    e16x8 qfb[4];
    fetch_frags<TE>(Qb, ld, 32 * qb, L, ol, qfb);
    const int q = 32 * qb + rr;
    const uint64_t didx0 = ((uint64_t)(tok0 + q) * heads + head) * 256;
    float m = -INFINITY, l = 0.f;
    f32x16 o[2] = {zero16(), zero16()};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x16 st = zero16();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st = E16<TE>::mfma32(kf[kt][ks], qfb[ks], st);
      if (kt == NKT - 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = 32 * kt + acc_row(e, h) < L ? st[e] : -INFINITY;
      }
      float tmx = st[0];
#pragma unroll
      for (int e = 1; e < 16; ++e) tmx = fmaxf(tmx, st[e]);
      tmx = fmaxf(tmx, __shfl_xor(tmx, 32, 64));
      const float mn = fmaxf(m, tmx);                  // finite: tile 0 always holds key 0 < L
      const float nmc = -mn * kC;
      float rs = 0.f;
      e16x8 pf[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[e], kC, nmc));   // exp((s - max) / 8); 0 for masked keys
        rs += p;
        pf[e >> 3][e & 7] = (TE)drop.apply(p, didx0 + (uint64_t)(32 * kt + acc_row(e, h)));
      }
      rs += __shfl_xor(rs, 32, 64);
      if (kt == 0) {
        l = rs;
      } else {
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * kC);
        l = l * alpha + rs;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
      }
      m = mn;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) o[dt] = E16<TE>::mfma32(tr_frag<TE>(vl, tr_base, kt, s2, dt), pf[s2], o[dt]);
    }
    store_rows<TE>(ol, o, 1.0f / l, 32 * qb, L, ctx ? ctx + (size_t)tok0 * H + head * 64 : nullptr, (size_t)H,
                   ctx16 ? ctx16 + (size_t)tok0 * H + head * 64 : nullptr, (size_t)H);
    if (h == 0 && q < L) ml[(size_t)(tok0 + q) * heads + head] = float2{m, l};
  }
}

template <typename TE, int NKTMAX>
__global__ __launch_bounds__(256) void attn_train_mfma_fwd_kernel(const TE* __restrict__ qkv, float* __restrict__ ctx, TE* __restrict__ ctx16,
                                                                  float2* __restrict__ ml, const int32_t* __restrict__ cu, int64_t n_pairs,
                                                                  int heads, int H, int lds_per_wave, Drop drop) {
  extern __shared__ __attribute__((aligned(16))) char vlds[];
  const int wave = threadIdx.x >> 6;
  const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
  if (pair >= n_pairs) return;
  const int n = (int)(pair / heads), head = (int)(pair - (int64_t)n * heads);
  const int tok0 = __builtin_amdgcn_readfirstlane(cu[n]);
  const int L = __builtin_amdgcn_readfirstlane(cu[n + 1]) - tok0;
  if (L <= 0) return;
  char* vl = vlds + wave * lds_per_wave;
  // NKTMAX = key tiles of the batch's padded length: a title-length batch compiles to the one-tile body only (few registers,
  // 8 KiB of LDS per wave -> many waves per CU); the host guarantees L <= 32 NKTMAX
  if (NKTMAX == 1 || L <= 32) fwd_wave<TE, 1>(qkv, ctx, ctx16, ml, tok0, L, H, heads, head, vl, drop);
  else if (NKTMAX == 2 || L <= 64) { if constexpr (NKTMAX >= 2) fwd_wave<TE, 2>(qkv, ctx, ctx16, ml, tok0, L, H, heads, head, vl, drop); }
  else if (NKTMAX == 3 || L <= 96) { if constexpr (NKTMAX >= 3) fwd_wave<TE, 3>(qkv, ctx, ctx16, ml, tok0, L, H, heads, head, vl, drop); }
  else { if constexpr (NKTMAX >= 4) fwd_wave<TE, 4>(qkv, ctx, ctx16, ml, tok0, L, H, heads, head, vl, drop); }
}

// ------------------------------------------------------------------------------------------------ backward pre-pass
// dsum[m, head] = sum_d dctx[m, 64 head + d] ctx[m, 64 head + d];  dctx16 = 16-bit copy of dctx.  One wave per row.
// TD / TC = storage type of dctx / ctx (float or TE).  Round 5: with 16-bit saved activations ctx exists in the 16-bit type only and
// d ctx leaves its data-gradient GEMM in the 16-bit type (TD == TE: nothing to copy, dctx16 is not written).
template <typename T> __device__ __forceinline__ f32x4 load4f(const T* p);
template <> __device__ __forceinline__ f32x4 load4f<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 load4f<f16_t>(const f16_t* p) {
  const f16x4 v = *reinterpret_cast<const f16x4*>(p);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <> __device__ __forceinline__ f32x4 load4f<bf16_t>(const bf16_t* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename TE, typename TD, typename TC>
__global__ __launch_bounds__(256) void attn_train_prep_kernel(const TD* __restrict__ dctx, const TC* __restrict__ ctx,
                                                              TE* __restrict__ dctx16, float* __restrict__ dsum, int H, int heads,
                                                              const int* __restrict__ m_total) {
  typedef typename E16<TE>::v4 e16x4;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (m >= m_total[0]) return;
  const TD* a = dctx + (size_t)m * H;
  const TC* b = ctx + (size_t)m * H;
  e16x4* o = reinterpret_cast<e16x4*>(dctx16 + (size_t)m * H);
  const int nf = H / 4;                                      // a head = 16 consecutive groups of 4
  for (int f0 = 0; f0 < nf; f0 += 64) {
    const int f = f0 + lane;
    float s = 0.f;
    if (f < nf) {
      const f32x4 x = load4f<TD>(a + 4 * f), y = load4f<TC>(b + 4 * f);
      s = x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
      if constexpr (sizeof(TD) == 4) o[f] = e16x4{(TE)x[0], (TE)x[1], (TE)x[2], (TE)x[3]};
    }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) s += __shfl_xor(s, d, 64);
    if (f < nf && (lane & 15) == 0) dsum[(size_t)m * heads + (f >> 4)] = s;
  }
}

// ------------------------------------------------------------------------------------------------ backward: d q
template <typename TE, int NKT>
__device__ __forceinline__ void bwd_q_wave(const TE* __restrict__ qkv, const TE* __restrict__ dctx16, const float2* __restrict__ ml,
                                           const float* __restrict__ dsum, float* __restrict__ dqkv, TE* __restrict__ dqkv16, int tok0,
                                           int L, int H, int heads, int head, char* vl, const Drop& drop) {
  typedef typename E16<TE>::v8 e16x8;
  const int lane = threadIdx.x & 63, rr = lane & 31, h = lane >> 5;
  char* ol = vl + NKT * 4096;
  const size_t ld = 3 * (size_t)H;
  const TE* Qb = qkv + (size_t)tok0 * ld + head * 64;
  const TE* Kb = Qb + H;
  const TE* Vb = Qb + 2 * H;
  const TE* Gb = dctx16 + (size_t)tok0 * H + head * 64;
  dma_image<TE, NKT>(Kb, ld, L, vl);
  e16x8 kf[NKT][4], vf[NKT][4];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    fetch_frags<TE>(Kb, ld, 32 * kt, L, ol, kf[kt]);
    fetch_frags<TE>(Vb, ld, 32 * kt, L, ol, vf[kt]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the K image
  const int tr_base = tr_lane_base();
#pragma unroll
  for (int qb = 0; qb < NKT; ++qb) {
    if (32 * qb >= L) break;
    e16x8 qfb[4], gfb[4];
    fetch_frags<TE>(Qb, ld, 32 * qb, L, ol, qfb);
    fetch_frags<TE>(Gb, (size_t)H, 32 * qb, L, ol, gfb);
    const int q = min(32 * qb + rr, L - 1);
    const float2 st2 = ml[(size_t)(tok0 + q) * heads + head];
    const float nmc = -st2.x * kC, invl = 1.0f / st2.y;
    const float D = dsum[(size_t)(tok0 + q) * heads + head];
    const uint64_t didx0 = ((uint64_t)(tok0 + 32 * qb + rr) * heads + head) * 256;
    f32x16 dq[2] = {zero16(), zero16()};
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x16 st = zero16(), dp = zero16();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        st = E16<TE>::mfma32(kf[kt][ks], qfb[ks], st);
        dp = E16<TE>::mfma32(vf[kt][ks], gfb[ks], dp);
      }
      e16x8 dsf[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = 32 * kt + acc_row(e, h);
        const float p = key < L ? __builtin_amdgcn_exp2f(fmaf(st[e], kC, nmc)) * invl : 0.f;
        const float g = drop.apply(dp[e], didx0 + (uint64_t)key);
        dsf[e >> 3][e & 7] = (TE)(p * (g - D) * 0.125f);
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) dq[dt] = E16<TE>::mfma32(tr_frag<TE>(vl, tr_base, kt, s2, dt), dsf[s2], dq[dt]);
    }
    store_rows<TE>(ol, dq, 1.0f, 32 * qb, L, dqkv ? dqkv + (size_t)tok0 * ld + head * 64 : nullptr, ld,
                   dqkv16 ? dqkv16 + (size_t)tok0 * ld + head * 64 : nullptr, ld);
  }
}

template <typename TE, int NKTMAX>
__global__ __launch_bounds__(256) void attn_train_mfma_bwd_q_kernel(const TE* __restrict__ qkv, const TE* __restrict__ dctx16,
                                                                    const float2* __restrict__ ml, const float* __restrict__ dsum,
                                                                    float* __restrict__ dqkv, TE* __restrict__ dqkv16,
                                                                    const int32_t* __restrict__ cu, int64_t n_pairs, int heads, int H,
                                                                    int lds_per_wave, Drop drop) {
  extern __shared__ __attribute__((aligned(16))) char vlds[];
  const int wave = threadIdx.x >> 6;
  const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
  if (pair >= n_pairs) return;
  const int n = (int)(pair / heads), head = (int)(pair - (int64_t)n * heads);
  const int tok0 = __builtin_amdgcn_readfirstlane(cu[n]);
  const int L = __builtin_amdgcn_readfirstlane(cu[n + 1]) - tok0;
  if (L <= 0) return;
  char* vl = vlds + wave * lds_per_wave;
  // NKTMAX = key tiles of the batch's padded length: a title-length batch compiles to the one-tile body only (few registers,
  // 8 KiB of LDS per wave -> many waves per CU); the host guarantees L <= 32 NKTMAX
  if (NKTMAX == 1 || L <= 32) bwd_q_wave<TE, 1>(qkv, dctx16, ml, dsum, dqkv, dqkv16, tok0, L, H, heads, head, vl, drop);
  else if (NKTMAX == 2 || L <= 64) { if constexpr (NKTMAX >= 2) bwd_q_wave<TE, 2>(qkv, dctx16, ml, dsum, dqkv, dqkv16, tok0, L, H, heads, head, vl, drop); }
  else if (NKTMAX == 3 || L <= 96) { if constexpr (NKTMAX >= 3) bwd_q_wave<TE, 3>(qkv, dctx16, ml, dsum, dqkv, dqkv16, tok0, L, H, heads, head, vl, drop); }
  else { if constexpr (NKTMAX >= 4) bwd_q_wave<TE, 4>(qkv, dctx16, ml, dsum, dqkv, dqkv16, tok0, L, H, heads, head, vl, drop); }
}

// ------------------------------------------------------------------------------------------------ backward: d k, d v
template <typename TE, int NKT>
__device__ __forceinline__ void bwd_kv_wave(const TE* __restrict__ qkv, const TE* __restrict__ dctx16, const float2* __restrict__ ml,
                                            const float* __restrict__ dsum, float* __restrict__ dqkv, TE* __restrict__ dqkv16, int tok0,
                                            int L, int H, int heads, int head, char* vl, const Drop& drop) {
  typedef typename E16<TE>::v8 e16x8;
  const int lane = threadIdx.x & 63, rr = lane & 31, h = lane >> 5;
  char* ql = vl;                                     // Q image, dO image, slab, row statistics
  char* gl = vl + NKT * 4096;
  char* ol = vl + 2 * NKT * 4096;
  float* sm = reinterpret_cast<float*>(ol + 4096);   // [128] -max * C, [128] 1 / sum, [128] D
  float* sl = sm + 128;
  float* sd = sl + 128;
  const size_t ld = 3 * (size_t)H;
  const TE* Qb = qkv + (size_t)tok0 * ld + head * 64;
  const TE* Kb = Qb + H;
  const TE* Vb = Qb + 2 * H;
  const TE* Gb = dctx16 + (size_t)tok0 * H + head * 64;
  dma_image<TE, NKT>(Qb, ld, L, ql);
  dma_image<TE, NKT>(Gb, (size_t)H, L, gl);
  for (int r = lane; r < 32 * NKT; r += 64) {
    const int row = min(r, L - 1);
    const float2 st2 = ml[(size_t)(tok0 + row) * heads + head];
    sm[r] = -st2.x * kC;
    sl[r] = 1.0f / st2.y;
    sd[r] = dsum[(size_t)(tok0 + row) * heads + head];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  const int tr_base = tr_lane_base();
#pragma unroll
  for (int kb = 0; kb < NKT; ++kb) {
    if (32 * kb >= L) break;
    e16x8 kfb[4], vfb[4];
    fetch_frags<TE>(Kb, ld, 32 * kb, L, ol, kfb);
    fetch_frags<TE>(Vb, ld, 32 * kb, L, ol, vfb);
    const int key = 32 * kb + rr;
    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
#pragma unroll
    for (int qt = 0; qt < NKT; ++qt) {
      if (32 * qt >= L) break;
      e16x8 qf[4], gf[4];
      fetch_frags<TE>(Qb, ld, 32 * qt, L, ol, qf);
      fetch_frags<TE>(Gb, (size_t)H, 32 * qt, L, ol, gf);
      f32x16 s = zero16(), dp = zero16();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s = E16<TE>::mfma32(qf[ks], kfb[ks], s);          // [query][key]: lane = key rr, 16 queries per lane
        dp = E16<TE>::mfma32(gf[ks], vfb[ks], dp);
      }
      e16x8 pdf[2], dsf[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qi = 32 * qt + acc_row(e, h);
        const float p = qi < L ? __builtin_amdgcn_exp2f(fmaf(s[e], kC, sm[qi])) * sl[qi] : 0.f;
        const bool keep = drop.thr == 0 || drop_bits(drop.seed, drop.site, ((uint64_t)(tok0 + qi) * heads + head) * 256 + (uint64_t)key) >= drop.thr;
        const float pd = keep ? p * drop.scale : 0.f;
        const float g = keep ? dp[e] * drop.scale : 0.f;
        pdf[e >> 3][e & 7] = (TE)pd;
        dsf[e >> 3][e & 7] = (TE)(p * (g - sd[qi]) * 0.125f);
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          dv[dt] = E16<TE>::mfma32(tr_frag<TE>(gl, tr_base, qt, s2, dt), pdf[s2], dv[dt]);
          dk[dt] = E16<TE>::mfma32(tr_frag<TE>(ql, tr_base, qt, s2, dt), dsf[s2], dk[dt]);
        }
    }
    store_rows<TE>(ol, dk, 1.0f, 32 * kb, L, dqkv ? dqkv + (size_t)tok0 * ld + H + head * 64 : nullptr, ld,
                   dqkv16 ? dqkv16 + (size_t)tok0 * ld + H + head * 64 : nullptr, ld);
    store_rows<TE>(ol, dv, 1.0f, 32 * kb, L, dqkv ? dqkv + (size_t)tok0 * ld + 2 * H + head * 64 : nullptr, ld,
                   dqkv16 ? dqkv16 + (size_t)tok0 * ld + 2 * H + head * 64 : nullptr, ld);
  }
}

template <typename TE, int NKTMAX>
__global__ __launch_bounds__(256) void attn_train_mfma_bwd_kv_kernel(const TE* __restrict__ qkv, const TE* __restrict__ dctx16,
                                                                     const float2* __restrict__ ml, const float* __restrict__ dsum,
                                                                     float* __restrict__ dqkv, TE* __restrict__ dqkv16,
                                                                     const int32_t* __restrict__ cu, int64_t n_pairs, int heads, int H,
                                                                     int lds_per_wave, Drop drop) {
  extern __shared__ __attribute__((aligned(16))) char vlds[];
  const int wave = threadIdx.x >> 6;
  const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
  if (pair >= n_pairs) return;
  const int n = (int)(pair / heads), head = (int)(pair - (int64_t)n * heads);
  const int tok0 = __builtin_amdgcn_readfirstlane(cu[n]);
  const int L = __builtin_amdgcn_readfirstlane(cu[n + 1]) - tok0;
  if (L <= 0) return;
  char* vl = vlds + wave * lds_per_wave;
  // NKTMAX = key tiles of the batch's padded length: a title-length batch compiles to the one-tile body only (few registers,
  // 8 KiB of LDS per wave -> many waves per CU); the host guarantees L <= 32 NKTMAX
  if (NKTMAX == 1 || L <= 32) bwd_kv_wave<TE, 1>(qkv, dctx16, ml, dsum, dqkv, dqkv16, tok0, L, H, heads, head, vl, drop);
  else if (NKTMAX == 2 || L <= 64) { if constexpr (NKTMAX >= 2) bwd_kv_wave<TE, 2>(qkv, dctx16, ml, dsum, dqkv, dqkv16, tok0, L, H, heads, head, vl, drop); }
  else if (NKTMAX == 3 || L <= 96) { if constexpr (NKTMAX >= 3) bwd_kv_wave<TE, 3>(qkv, dctx16, ml, dsum, dqkv, dqkv16, tok0, L, H, heads, head, vl, drop); }
  else { if constexpr (NKTMAX >= 4) bwd_kv_wave<TE, 4>(qkv, dctx16, ml, dsum, dqkv, dqkv16, tok0, L, H, heads, head, vl, drop); }
}

constexpr int MAX_LDS_FWD = 4 * (4 * 4096 + 4096);                 // 4 waves x (V image of 4 tiles + slab)
constexpr int MAX_LDS_KV = 4 * (2 * 4 * 4096 + 4096 + 2048);       // 4 waves x (Q + dO images + slab + row statistics)

template <typename TE, int NKTMAX>
int raise_lds() {
  MANNER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_train_mfma_fwd_kernel<TE, NKTMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_FWD));
  MANNER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_train_mfma_bwd_q_kernel<TE, NKTMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_FWD));
  MANNER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_train_mfma_bwd_kv_kernel<TE, NKTMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_KV));
  return MANNER_HIP_OK;
}
int ensure_lds() {
  static bool raised[MAX_DEVICES] = {};
  bool& r = raised[current_device_slot()];
  if (!r) {
    int rc;
    if ((rc = raise_lds<f16_t, 1>()) || (rc = raise_lds<f16_t, 2>()) || (rc = raise_lds<f16_t, 3>()) || (rc = raise_lds<f16_t, 4>()) ||
        (rc = raise_lds<bf16_t, 1>()) || (rc = raise_lds<bf16_t, 2>()) || (rc = raise_lds<bf16_t, 3>()) || (rc = raise_lds<bf16_t, 4>()))
      return rc;
    r = true;
  }
  return MANNER_HIP_OK;
}

int check_shape(DType dt, int heads, int H, int max_len) {
  if (!is_16bit(dt)) return fail(MANNER_HIP_E_INVALID, "attn_train_mfma: 16-bit element types only");
  if (H != heads * 64 || H % 8) return fail(MANNER_HIP_E_INVALID, "attn_train_mfma: head_dim must be 64 (H=%d heads=%d)", H, heads);
  if (max_len < 1 || max_len > MANNER_HIP_MAX_LEN) return fail(MANNER_HIP_E_INVALID, "attn_train_mfma: padded length %d outside [1, %d]", max_len, MANNER_HIP_MAX_LEN);
  return MANNER_HIP_OK;
}

}  // namespace

int attn_train_mfma_forward(DType dt, const void* qkv16, float* ctx, void* ctx16, float2* ml, const int32_t* cu, int64_t n_news,
                            int heads, int H, int max_len, Drop drop, hipStream_t stream) {
  int rc;
  if ((rc = check_shape(dt, heads, H, max_len)) || (rc = ensure_lds())) return rc;
  const int64_t pairs = n_news * heads;
  const int nkt = (max_len + 31) / 32;
  const int lds_per_wave = nkt * 4096 + 4096;
  const dim3 g((unsigned)((pairs + 3) / 4)), b(256);
#define MANNER_ATTN_FWD_LAUNCH(TE_, N_)                                                                                                   \
  hipLaunchKernelGGL((attn_train_mfma_fwd_kernel<TE_, N_>), g, b, 4 * lds_per_wave, stream, static_cast<const TE_*>(qkv16), ctx,         \
                     static_cast<TE_*>(ctx16), ml, cu, pairs, heads, H, lds_per_wave, drop)
#define MANNER_ATTN_BY_NKT(M_, TE_) \
  do { if (nkt == 1) { M_(TE_, 1); } else if (nkt == 2) { M_(TE_, 2); } else if (nkt == 3) { M_(TE_, 3); } else { M_(TE_, 4); } } while (0)
  if (dt == DT_F16) MANNER_ATTN_BY_NKT(MANNER_ATTN_FWD_LAUNCH, f16_t);
  else MANNER_ATTN_BY_NKT(MANNER_ATTN_FWD_LAUNCH, bf16_t);
#undef MANNER_ATTN_FWD_LAUNCH
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int attn_train_mfma_backward(DType dt, const void* qkv16, const void* dctx, bool dctx_is16, const void* ctx, bool ctx_is16, const float2* ml,
                             float* dqkv, void* dqkv16, void* dctx16, float* dsum, const int32_t* cu, int64_t n_news, int heads, int H,
                             int max_len, Drop drop, int64_t m_bound, const int* m_total, hipStream_t stream) {
  int rc;
  if ((rc = check_shape(dt, heads, H, max_len)) || (rc = ensure_lds())) return rc;
  if (!dqkv && !dqkv16) return fail(MANNER_HIP_E_INVALID, "attn_train_mfma_backward: no output");
  if (dctx_is16) dctx16 = const_cast<void*>(dctx);          // d ctx arrived in the 16-bit type: it IS the operand of the two kernels
  const int64_t pairs = n_news * heads;
  const int nkt = (max_len + 31) / 32;
  const int lds_q = nkt * 4096 + 4096, lds_kv = 2 * nkt * 4096 + 4096 + 2048;
  const dim3 g((unsigned)((pairs + 3) / 4)), b(256), gp((unsigned)((m_bound + 3) / 4));
#define MANNER_ATTN_BWD_LAUNCH(TE_, N_)                                                                                                      \
  do {                                                                                                                                       \
    if (dctx_is16 && ctx_is16)                                                                                                                \
      hipLaunchKernelGGL((attn_train_prep_kernel<TE_, TE_, TE_>), gp, b, 0, stream, static_cast<const TE_*>(dctx), static_cast<const TE_*>(ctx), \
                         static_cast<TE_*>(dctx16), dsum, H, heads, m_total);                                                                 \
    else if (dctx_is16)                                                                                                                       \
      hipLaunchKernelGGL((attn_train_prep_kernel<TE_, TE_, float>), gp, b, 0, stream, static_cast<const TE_*>(dctx), static_cast<const float*>(ctx), \
                         static_cast<TE_*>(dctx16), dsum, H, heads, m_total);                                                                 \
    else if (ctx_is16)                                                                                                                        \
      hipLaunchKernelGGL((attn_train_prep_kernel<TE_, float, TE_>), gp, b, 0, stream, static_cast<const float*>(dctx), static_cast<const TE_*>(ctx), \
                         static_cast<TE_*>(dctx16), dsum, H, heads, m_total);                                                                 \
    else                                                                                                                                      \
      hipLaunchKernelGGL((attn_train_prep_kernel<TE_, float, float>), gp, b, 0, stream, static_cast<const float*>(dctx), static_cast<const float*>(ctx), \
                         static_cast<TE_*>(dctx16), dsum, H, heads, m_total);                                                                 \
    hipLaunchKernelGGL((attn_train_mfma_bwd_q_kernel<TE_, N_>), g, b, 4 * lds_q, stream, static_cast<const TE_*>(qkv16),                      \
                       static_cast<const TE_*>(dctx16), ml, dsum, dqkv, static_cast<TE_*>(dqkv16), cu, pairs, heads, H, lds_q, drop);        \
    hipLaunchKernelGGL((attn_train_mfma_bwd_kv_kernel<TE_, N_>), g, b, 4 * lds_kv, stream, static_cast<const TE_*>(qkv16),                    \
                       static_cast<const TE_*>(dctx16), ml, dsum, dqkv, static_cast<TE_*>(dqkv16), cu, pairs, heads, H, lds_kv, drop);       \
  } while (0)
  if (dt == DT_F16) MANNER_ATTN_BY_NKT(MANNER_ATTN_BWD_LAUNCH, f16_t);
  else MANNER_ATTN_BY_NKT(MANNER_ATTN_BWD_LAUNCH, bf16_t);
#undef MANNER_ATTN_BWD_LAUNCH
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

}  // namespace manner
