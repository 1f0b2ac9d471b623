// HBM-bound tail of the hot path: additive-attention pooler (K11), dot-product scorer (K12), fused
// gather-mean-dot scorer over the news-embedding table (K9+K10+K12), per-impression z-score +
// weighted fusion (K13+K14) and stable ranking / nDCG@k (K15).  All of it is f32, coalesced
// 16-byte row reads with 64-lane shuffle reductions; no matrix cores (≈0.5 FLOP per byte).
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace manner {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// dot of a table row with a vector held in LDS; D % 4 == 0
__device__ __forceinline__ float row_dot(const float* __restrict__ row, const float* vec, int D, int lane) {
  float a = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(row + c);
    const f32x4 u = *reinterpret_cast<const f32x4*>(vec + c);
    a += (x[0] * u[0] + x[1] * u[1]) + (x[2] * u[2] + x[3] * u[3]);
  }
  return wave_sum(a);
}

// ---------------------------------------------------------------- K9 + K10 + K12 fused
// one workgroup per impression: user = mean(table[hist]) (cr_module.py:116-123), then one wave per
// candidate row (click_predictors.py:12).  `user` lives in LDS; table rows are read exactly once.
// An index outside [0, n_rows) is an IndexError in the reference; here it raises MANNER_HIP_STATUS_INDEX in the
// caller's status word (the row is then read as row 0 so the kernel stays in bounds).
__device__ __forceinline__ int64_t checked_row(int64_t r, int64_t n_rows, int32_t* status, int lane) {
  if (r >= 0 && r < n_rows) return r;
  if (status && lane == 0) atomicOr(status, MANNER_HIP_STATUS_INDEX);
  return 0;
}

// `user_in` != NULL: the user vectors are given ([B, D], e.g. the additive pooler's output — early fusion,
// cr_module.py:125) and the history lists are not read.
__global__ __launch_bounds__(256) void score_late_fusion_kernel(
    const float* __restrict__ table, int64_t n_rows, int D, const int32_t* __restrict__ hist_idx,
    const int64_t* __restrict__ hist_off, const float* __restrict__ user_in, const int32_t* __restrict__ cand_idx,
    const int64_t* __restrict__ cand_off, float* __restrict__ out, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][D] wave partials + [D] user
  const int64_t b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* user = sm + 4 * D;
  if (user_in) {
    for (int c = threadIdx.x * 4; c < D; c += 1024)
      *reinterpret_cast<f32x4*>(user + c) = *reinterpret_cast<const f32x4*>(user_in + b * D + c);
  } else {
    const int64_t h0 = hist_off[b], h1 = hist_off[b + 1];
    for (int cb = 0; cb < D; cb += 256) {
      const int c = cb + lane * 4;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (c < D)
        for (int64_t j = h0 + wave; j < h1; j += 4) {
          const int64_t r = checked_row(hist_idx[j], n_rows, status, lane);
          acc += *reinterpret_cast<const f32x4*>(table + r * D + c);
        }
      if (c < D) *reinterpret_cast<f32x4*>(sm + wave * D + c) = acc;
    }
    __syncthreads();
    const float hn = (float)(h1 - h0);
    for (int c = threadIdx.x; c < D; c += 256)
      user[c] = ((sm[c] + sm[D + c]) + (sm[2 * D + c] + sm[3 * D + c])) / hn;   // torch.div(sum, hist_size)
  }
  __syncthreads();
  const int64_t c0 = cand_off[b], c1 = cand_off[b + 1];
  // two candidate rows in flight per wave (the gathers are latency-bound).  ONE code path for every candidate — an odd last
  // candidate is paired with itself — so that two occurrences of the same news get bit-identical scores (exact ties must
  // stay ties for the stable ranking)
  for (int64_t j = c0 + wave; j < c1; j += 8) {
#pragma clang fp contract(off)                        // the two accumulations must round alike (no FMA picked for one of them only)
    const bool two = j + 4 < c1;
    const int64_t r0 = checked_row(cand_idx[j], n_rows, status, lane), r1 = two ? checked_row(cand_idx[j + 4], n_rows, status, lane) : r0;
    const float* p0 = table + r0 * D;
    const float* p1 = table + r1 * D;
    float a0 = 0.f, a1 = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(p0 + c), x1 = *reinterpret_cast<const f32x4*>(p1 + c);
      const f32x4 u = *reinterpret_cast<const f32x4*>(user + c);
      a0 += (x0[0] * u[0] + x0[1] * u[1]) + (x0[2] * u[2] + x0[3] * u[3]);
      a1 += (x1[0] * u[0] + x1[1] * u[1]) + (x1[2] * u[2] + x1[3] * u[3]);
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    if (lane == 0) { out[j] = a0; if (two) out[j + 4] = a1; }
  }
}

// D == NCB * 256 (768 and 1024: every PLM the reference ships): a lane holds its NCB 16-byte pieces of a WHOLE row, and a wave
// keeps RIF rows in flight — in the history sum as well, which in the kernel above issues one dependent 16-byte load per lane at
// a time (22 history rows = 16 serial round trips to L2 / HBM per wave: the scorer was bound by that latency chain, not by bytes).
// Row -> wave assignment, the order of every addition and the dot product's association are those of the kernel above: the two
// produce the same bits (zero rows added for the slots past the end: x + 0 is exact and the sums start at +0).
// The body is shared by the stand-alone scorer kernel (emit = a store into the ragged score vector) and by the fused Phase-C kernel
// below (emit = a store into the impression's LDS plane): one code path, so the two give the same bits.  `sm`: 5 D floats of LDS.
template <int NCB, int RIF, typename Emit>
__device__ __forceinline__ void score_rows_body(
    const float* __restrict__ table, int64_t n_rows, const int32_t* __restrict__ hist_idx, const int64_t* __restrict__ hist_off,
    const float* __restrict__ user_in, const int32_t* __restrict__ cand_idx, const int64_t* __restrict__ cand_off,
    const int64_t b, float* sm, int32_t* __restrict__ status, Emit emit) {
  constexpr int D = NCB * 256;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // wave in an SGPR: the index lists are read by scalar loads
  float* user = sm + 4 * D;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // RIF rows of one list, all in flight: the RIF indices first (wave-uniform, scalar loads: they do not queue behind the rows'
  // vector loads), then every 16-byte piece of every row, and only then the first use
  auto fetch = [&](const int32_t* __restrict__ idx, int64_t j, int64_t end, f32x4 (&x)[RIF][NCB]) {
    int64_t r[RIF];
#pragma unroll
    for (int q = 0; q < RIF; ++q) r[q] = idx[j + 4 * q < end ? j + 4 * q : j];
#pragma unroll
    for (int q = 0; q < RIF; ++q) {
      if (j + 4 * q < end) {                            // wave-uniform
        const float* p = table + checked_row(r[q], n_rows, status, lane) * D + lane * 4;
#pragma unroll
        for (int k = 0; k < NCB; ++k) x[q][k] = *reinterpret_cast<const f32x4*>(p + k * 256);
      } else {
#pragma unroll
        for (int k = 0; k < NCB; ++k) x[q][k] = zero;
      }
    }
  };
  const int64_t c0 = cand_off[b], c1 = cand_off[b + 1];
  f32x4 x[RIF][NCB];                                  // rows in flight: history batches, then candidate batches
  if (user_in) {
    if (c0 + wave < c1) fetch(cand_idx, c0 + wave, c1, x);
    for (int c = threadIdx.x * 4; c < D; c += 1024)
      *reinterpret_cast<f32x4*>(user + c) = *reinterpret_cast<const f32x4*>(user_in + b * D + c);
  } else {
    const int64_t h0 = hist_off[b], h1 = hist_off[b + 1];
    f32x4 acc[NCB];
#pragma unroll
    for (int k = 0; k < NCB; ++k) acc[k] = zero;
    for (int64_t j = h0 + wave; j < h1; j += 4 * RIF) {
      fetch(hist_idx, j, h1, x);
#pragma unroll
      for (int q = 0; q < RIF; ++q)
#pragma unroll
        for (int k = 0; k < NCB; ++k) acc[k] += x[q][k];
    }
    // the first candidate batch does not depend on the user vector: its rows travel while the history sums meet in LDS
    if (c0 + wave < c1) fetch(cand_idx, c0 + wave, c1, x);
#pragma unroll
    for (int k = 0; k < NCB; ++k) *reinterpret_cast<f32x4*>(sm + wave * D + k * 256 + lane * 4) = acc[k];
    __syncthreads();
    const float hn = (float)(h1 - h0);
    for (int c = threadIdx.x; c < D; c += 256)
      user[c] = ((sm[c] + sm[D + c]) + (sm[2 * D + c] + sm[3 * D + c])) / hn;   // torch.div(sum, hist_size)
  }
  __syncthreads();
  f32x4 u[NCB];
#pragma unroll
  for (int k = 0; k < NCB; ++k) u[k] = *reinterpret_cast<const f32x4*>(user + k * 256 + lane * 4);
  for (int64_t j = c0 + wave; j < c1; j += 4 * RIF) {
#pragma clang fp contract(off)                        // every candidate's accumulation must round alike (exact ties stay ties)
    if (j != c0 + wave) fetch(cand_idx, j, c1, x);
    float a[RIF];
#pragma unroll
    for (int q = 0; q < RIF; ++q) {
      a[q] = 0.f;
#pragma unroll
      for (int k = 0; k < NCB; ++k)
        a[q] += (x[q][k][0] * u[k][0] + x[q][k][1] * u[k][1]) + (x[q][k][2] * u[k][2] + x[q][k][3] * u[k][3]);
      a[q] = wave_sum(a[q]);
    }
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < RIF; ++q)
        if (j + 4 * q < c1) emit(j + 4 * q, a[q]);
    }
  }
}

template <int NCB, int RIF>
__global__ __launch_bounds__(256) void score_late_fusion_rows_kernel(
    const float* __restrict__ table, int64_t n_rows, const int32_t* __restrict__ hist_idx, const int64_t* __restrict__ hist_off,
    const float* __restrict__ user_in, const int32_t* __restrict__ cand_idx, const int64_t* __restrict__ cand_off,
    float* __restrict__ out, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][D] wave partials + [D] user
  score_rows_body<NCB, RIF>(table, n_rows, hist_idx, hist_off, user_in, cand_idx, cand_off, (int64_t)blockIdx.x, sm, status,
                            [&](int64_t j, float v) { out[j] = v; });
}

// The same scorer over an IEEE-half copy of the table (manner_hip_score_late_fusion_f16): rows are 2 D bytes, so the MIND-large
// table (161 013 x 768) is 247 MB and stays resident in the 256 MiB Infinity Cache while the impressions stream their
// ~50 row gathers each — with the f32 table (495 MB) the Zipf tail of the gathers re-fetches rows from HBM 15.7 times over.
// Accumulation and the user vector stay f32; only the stored rows are rounded (2^-12 relative, below the error the 16-bit
// encoder modes put into those rows in the first place).  D % 8 == 0.
// `mu` != NULL: the table holds the CENTRED rows T' = T - mu (mu = column mean of T, f32 [D]).  Embedding tables of one
// encoder are nearly collinear (|score| in the hundreds, neighbouring candidates 0.01 - 0.1 apart), so the deviations from the
// mean row are an order of magnitude smaller than the entries and half precision rounds THEM: with w = mu + mean(T'[hist]),
// <mu + u', mu + c'> = <w, mu> + <w, c'> exactly, i.e. one extra dot product per impression.
__global__ __launch_bounds__(256) void score_late_fusion_f16_kernel(
    const f16_t* __restrict__ table, int64_t n_rows, int D, const int32_t* __restrict__ hist_idx,
    const int64_t* __restrict__ hist_off, const int32_t* __restrict__ cand_idx, const int64_t* __restrict__ cand_off,
    float* __restrict__ out, int32_t* __restrict__ status, const float* __restrict__ mu) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][D] wave partials + [D] user
  __shared__ float cpart[4];
  const int64_t b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* user = sm + 4 * D;
  const int64_t h0 = hist_off[b], h1 = hist_off[b + 1];
  for (int cb = 0; cb < D; cb += 512) {
    const int c = cb + lane * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < D) {
      for (int64_t j = h0 + wave; j < h1; j += 4) {
        const int64_t r = checked_row(hist_idx[j], n_rows, status, lane);
        const f16x8 x = *reinterpret_cast<const f16x8*>(table + r * D + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)x[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sm[wave * D + c + e] = acc[e];
    }
  }
  __syncthreads();
  const float hn = (float)(h1 - h0);
  float cp = 0.f;
  for (int c = threadIdx.x; c < D; c += 256) {
    float u = ((sm[c] + sm[D + c]) + (sm[2 * D + c] + sm[3 * D + c])) / hn;   // torch.div(sum, hist_size)
    if (mu) { const float m = mu[c]; u += m; cp = fmaf(u, m, cp); }
    user[c] = u;
  }
  cp = wave_sum(cp);
  if (lane == 0) cpart[wave] = cp;
  __syncthreads();
  const float konst = (cpart[0] + cpart[1]) + (cpart[2] + cpart[3]);          // <w, mu> (0 without centring)
  const int64_t c0 = cand_off[b], c1 = cand_off[b + 1];
  auto dot8 = [&](const f16x8& x, const f32x4& u0, const f32x4& u1) {
#pragma clang fp contract(off)
    return (((float)x[0] * u0[0] + (float)x[1] * u0[1]) + ((float)x[2] * u0[2] + (float)x[3] * u0[3])) +
           (((float)x[4] * u1[0] + (float)x[5] * u1[1]) + ((float)x[6] * u1[2] + (float)x[7] * u1[3]));
  };
  for (int64_t j = c0 + wave; j < c1; j += 8) {      // two candidate rows in flight per wave, one code path (see the f32 kernel)
#pragma clang fp contract(off)
    const bool two = j + 4 < c1;
    const int64_t r0 = checked_row(cand_idx[j], n_rows, status, lane), r1 = two ? checked_row(cand_idx[j + 4], n_rows, status, lane) : r0;
    const f16_t* p0 = table + r0 * D;
    const f16_t* p1 = table + r1 * D;
    float a0 = 0.f, a1 = 0.f;
    for (int c = lane * 8; c < D; c += 512) {
      const f16x8 x0 = *reinterpret_cast<const f16x8*>(p0 + c), x1 = *reinterpret_cast<const f16x8*>(p1 + c);
      const f32x4 u0 = *reinterpret_cast<const f32x4*>(user + c), u1 = *reinterpret_cast<const f32x4*>(user + c + 4);
      a0 += dot8(x0, u0, u1);
      a1 += dot8(x1, u0, u1);
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1);
    if (lane == 0) { out[j] = a0 + konst; if (two) out[j + 4] = a1 + konst; }
  }
}

// D == NCB * 256 over the half table: a row is NCB x 512 bytes = NCB 16-byte pieces for each of 32 lanes, so a wave works on TWO
// rows at a time (lanes 0-31 / 32-63) with every lane loading 16 bytes per instruction, RIF such pairs in flight — as in the f32
// kernel above the indices come first (scalar loads), then all pieces of all rows, then the first use.  One code path for every
// candidate (the two halves run the same instructions on their own lanes): occurrences of one news get identical bits.
template <int NCB, int RIF>
__global__ __launch_bounds__(256) void score_late_fusion_f16_rows_kernel(
    const f16_t* __restrict__ table, int64_t n_rows, const int32_t* __restrict__ hist_idx, const int64_t* __restrict__ hist_off,
    const int32_t* __restrict__ cand_idx, const int64_t* __restrict__ cand_off, float* __restrict__ out, int32_t* __restrict__ status,
    const float* __restrict__ mu) {
  constexpr int D = NCB * 256;
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [4][D] wave partials + [D] user
  __shared__ float cpart[4];
  const int64_t b = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
  float* user = sm + 4 * D;
  // rows j0 + 8 q + 2 wave + half (q < RIF) of one list
  auto fetch = [&](const int32_t* __restrict__ idx, int64_t j0, int64_t end, f16x8 (&x)[RIF][NCB], bool (&ok)[RIF]) {
    int64_t r[RIF];
#pragma unroll
    for (int q = 0; q < RIF; ++q) {
      const int64_t ja = j0 + 8 * q + 2 * wave, jb = ja + 1;                   // wave-uniform: scalar loads
      const int64_t ra = idx[ja < end ? ja : end - 1], rb = idx[jb < end ? jb : end - 1];
      r[q] = half ? rb : ra;
      ok[q] = (half ? jb : ja) < end;
    }
#pragma unroll
    for (int q = 0; q < RIF; ++q) {
      if (ok[q]) {
        const f16_t* p = table + checked_row(r[q], n_rows, status, hl) * D + hl * 8;
#pragma unroll
        for (int k = 0; k < NCB; ++k) x[q][k] = *reinterpret_cast<const f16x8*>(p + k * 256);
      } else {
#pragma unroll
        for (int k = 0; k < NCB; ++k)
#pragma unroll
          for (int e = 0; e < 8; ++e) x[q][k][e] = (f16_t)0.f;
      }
    }
  };
  const int64_t h0 = hist_off[b], h1 = hist_off[b + 1];
  const int64_t c0 = cand_off[b], c1 = cand_off[b + 1];
  f16x8 x[RIF][NCB];                                  // rows in flight: history batches, then candidate batches
  bool ok[RIF];
  {
    float acc[NCB][8];
#pragma unroll
    for (int k = 0; k < NCB; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
    for (int64_t j0 = h0; j0 < h1; j0 += 8 * RIF) {
      fetch(hist_idx, j0, h1, x, ok);
#pragma unroll
      for (int q = 0; q < RIF; ++q)
#pragma unroll
        for (int k = 0; k < NCB; ++k)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[k][e] += (float)x[q][k][e];           // zero rows past the end: x + 0 is exact
    }
    // the first candidate batch does not depend on the user vector: its rows travel while the history sums meet in LDS
    if (c0 < c1) fetch(cand_idx, c0, c1, x, ok);
#pragma unroll
    for (int k = 0; k < NCB; ++k) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[k][e] += __shfl_xor(acc[k][e], 32, 64);   // the wave's two rows-in-progress
      if (half == 0) {
        float* d = sm + wave * D + k * 256 + hl * 8;
        *reinterpret_cast<f32x4*>(d) = f32x4{acc[k][0], acc[k][1], acc[k][2], acc[k][3]};
        *reinterpret_cast<f32x4*>(d + 4) = f32x4{acc[k][4], acc[k][5], acc[k][6], acc[k][7]};
      }
    }
  }
  __syncthreads();
  const float hn = (float)(h1 - h0);
  float cp = 0.f;
  for (int c = threadIdx.x; c < D; c += 256) {
    float u = ((sm[c] + sm[D + c]) + (sm[2 * D + c] + sm[3 * D + c])) / hn;   // torch.div(sum, hist_size)
    if (mu) { const float m = mu[c]; u += m; cp = fmaf(u, m, cp); }
    user[c] = u;
  }
  cp = wave_sum(cp);
  if (lane == 0) cpart[wave] = cp;
  __syncthreads();
  const float konst = (cpart[0] + cpart[1]) + (cpart[2] + cpart[3]);          // <w, mu> (0 without centring)
  f32x4 u0[NCB], u1[NCB];
#pragma unroll
  for (int k = 0; k < NCB; ++k) {
    u0[k] = *reinterpret_cast<const f32x4*>(user + k * 256 + hl * 8);
    u1[k] = *reinterpret_cast<const f32x4*>(user + k * 256 + hl * 8 + 4);
  }
  for (int64_t j0 = c0; j0 < c1; j0 += 8 * RIF) {
#pragma clang fp contract(off)
    if (j0 != c0) fetch(cand_idx, j0, c1, x, ok);
#pragma unroll
    for (int q = 0; q < RIF; ++q) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < NCB; ++k)
        a += (((float)x[q][k][0] * u0[k][0] + (float)x[q][k][1] * u0[k][1]) + ((float)x[q][k][2] * u0[k][2] + (float)x[q][k][3] * u0[k][3])) +
             (((float)x[q][k][4] * u1[k][0] + (float)x[q][k][5] * u1[k][1]) + ((float)x[q][k][6] * u1[k][2] + (float)x[q][k][7] * u1[k][3]));
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);              // over the 32 lanes of the row
      if (hl == 0 && ok[q]) out[j0 + 8 * q + 2 * wave + half] = a + konst;
    }
  }
}

// column means of a table [n_rows, D] (two stages, fixed order) and the centred half copy T' = half(T - mu)
constexpr int MEAN_BLOCKS = 256;
__global__ __launch_bounds__(256) void col_partial_kernel(const float* __restrict__ t, int64_t n_rows, int D, float* __restrict__ part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  float s0 = 0.f, s1 = 0.f;
  int64_t r = blockIdx.y;
  for (; r + MEAN_BLOCKS < n_rows; r += 2 * MEAN_BLOCKS) { s0 += t[r * D + c]; s1 += t[(r + MEAN_BLOCKS) * D + c]; }
  if (r < n_rows) s0 += t[r * D + c];
  part[(size_t)blockIdx.y * D + c] = s0 + s1;
}
__global__ __launch_bounds__(256) void col_mean_kernel(const float* __restrict__ part, int64_t n_rows, int D, float* __restrict__ mu) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  double s = 0.0;
  for (int b = 0; b < MEAN_BLOCKS; ++b) s += (double)part[(size_t)b * D + c];
  mu[c] = (float)(s / (double)n_rows);
}
__global__ __launch_bounds__(256) void centre_f16_kernel(const float* __restrict__ t, const float* __restrict__ mu, int64_t n, int D,
                                                         f16_t* __restrict__ out) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(t + i);
    const f32x4 m = *reinterpret_cast<const f32x4*>(mu + (int)(i % D));
    *reinterpret_cast<f16x4*>(out + i) = f16x4{(f16_t)(x[0] - m[0]), (f16_t)(x[1] - m[1]), (f16_t)(x[2] - m[2]), (f16_t)(x[3] - m[3])};
  }
}

// ---------------------------------------------------------------- K9 to_dense_batch
// ragged rows x[off[b] + j] -> dense[b, j, :] for j < min(count_b, width); the other slots get fill[b] (or 0) and
// mask 0.  One wave per output row for D >= 4 (16-byte pieces), one thread per slot for scalars (D == 1).
__global__ __launch_bounds__(256) void to_dense_rows_kernel(const float* __restrict__ x, const int64_t* __restrict__ off,
                                                            int64_t B, int64_t width, int D, const float* __restrict__ fill,
                                                            float* __restrict__ dense, uint8_t* __restrict__ mask) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= B * width) return;
  const int64_t b = row / width, j = row - b * width;
  const int64_t o0 = off[b], cnt = off[b + 1] - o0;
  const bool real = j < cnt;
  float* dst = dense + row * D;
  if (real) {
    const float* src = x + (o0 + j) * D;
    for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<f32x4*>(dst + c) = *reinterpret_cast<const f32x4*>(src + c);
  } else {
    const float f = fill ? fill[b] : 0.f;
    const f32x4 fv = {f, f, f, f};
    for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<f32x4*>(dst + c) = fv;
  }
  if (mask && lane == 0) mask[row] = real ? 1 : 0;
}
__global__ __launch_bounds__(256) void to_dense_scalar_kernel(const float* __restrict__ x, const int64_t* __restrict__ off,
                                                              int64_t B, int64_t width, int D, const float* __restrict__ fill,
                                                              float* __restrict__ dense, uint8_t* __restrict__ mask) {
  const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (slot >= B * width) return;
  const int64_t b = slot / width, j = slot - b * width;
  const int64_t o0 = off[b], cnt = off[b + 1] - o0;
  const bool real = j < cnt;
  const float f = fill ? fill[b] : 0.f;
  for (int c = 0; c < D; ++c) dense[slot * D + c] = real ? x[(o0 + j) * D + c] : f;
  if (mask) mask[slot] = real ? 1 : 0;
}

// ---------------------------------------------------------------- K12 drop-in
// rows contiguous along D (the permuted view the reference passes): one wave per candidate
__global__ __launch_bounds__(256) void dot_rows_kernel(const float* __restrict__ user, const float* __restrict__ cand,
                                                       int64_t B, int64_t C, int D, int64_t sb, int64_t sc,
                                                       float* __restrict__ out, int vec_ok) {
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (p >= B * C) return;
  const int64_t b = p / C, c = p - b * C;
  const float* u = user + b * D;
  const float* x = cand + b * sb + c * sc;
  float a = 0.f;
  if (vec_ok) {
    for (int d = lane * 4; d < D; d += 256) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + d);
      const f32x4 uv = *reinterpret_cast<const f32x4*>(u + d);
      a += (xv[0] * uv[0] + xv[1] * uv[1]) + (xv[2] * uv[2] + xv[3] * uv[3]);
    }
  } else {
    for (int d = lane; d < D; d += 64) a += x[d] * u[d];
  }
  a = wave_sum(a);
  if (lane == 0) out[p] = a;
}
// general strides (e.g. a materialised [B,D,C] tensor): one thread per candidate, coalesced over c
__global__ __launch_bounds__(256) void dot_strided_kernel(const float* __restrict__ user, const float* __restrict__ cand,
                                                          int64_t B, int64_t C, int D, int64_t sb, int64_t sd, int64_t sc,
                                                          float* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= B * C) return;
  const int64_t b = p / C, c = p - b * C;
  const float* u = user + b * D;
  const float* x = cand + b * sb + c * sc;
  float a = 0.f;
  for (int d = 0; d < D; ++d) a = fmaf(x[(int64_t)d * sd], u[d], a);
  out[p] = a;
}

// ---------------------------------------------------------------- K11 additive attention
// AdditiveAttention.forward (attention.py:21-27), NO padding mask, in two passes:
//   logits[b,s] = sum_j tanh(<x[b,s], W[j]> + b[j]) q[j]   — on the f32 matrix cores (pool_logits_mfma, gemm.hip) when
//                 D % 32 == 0 and Q <= 256, else the VALU kernel below (one workgroup per (b,s) row; the entity
//                 branch's D = 100);
//   out[b]      = sum_s softmax_s(logits[b,:]) x[b,s]      — pool_apply_kernel: x is read once, 16 bytes per lane.
__global__ __launch_bounds__(256) void pool_logits_valu_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                               const float* __restrict__ bias, const float* __restrict__ query,
                                                               int D, int Q, float* __restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [D] row + [4] wave partials
  const int64_t r = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* part = sm + D;
  for (int c = threadIdx.x * 4; c < D; c += 1024)
    *reinterpret_cast<f32x4*>(sm + c) = *reinterpret_cast<const f32x4*>(x + r * D + c);
  __syncthreads();
  float acc = 0.f;
  for (int j = wave; j < Q; j += 4) {
    const float d = row_dot(W + (size_t)j * D, sm, D, lane);
    acc += tanhf(d + bias[j]) * query[j];
  }
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) logits[r] = (part[0] + part[1]) + (part[2] + part[3]);
}

// grid (B, ceil(D / 1024)): a thread owns 4 consecutive columns; the softmax over s (dim=1) is recomputed per block
__global__ __launch_bounds__(256) void pool_apply_kernel(const float* __restrict__ x, const float* __restrict__ logits,
                                                         int64_t S, int D, float* __restrict__ out) {
  __shared__ float wts[256];
  __shared__ float red[8];
  const int64_t b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* lg = logits + b * S;
  float mx = -INFINITY;
  for (int64_t s = threadIdx.x; s < S; s += 256) mx = fmaxf(mx, lg[s]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float den = 0.f;
  for (int64_t s = threadIdx.x; s < S; s += 256) den += expf(lg[s] - mx);
  den = wave_sum(den);
  if (lane == 0) red[4 + wave] = den;
  __syncthreads();
  den = (red[4] + red[5]) + (red[6] + red[7]);
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  const float* xb = x + b * S * D;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int64_t s0 = 0; s0 < S; s0 += 256) {
    const int64_t ns = min((int64_t)256, S - s0);
    __syncthreads();
    if (threadIdx.x < ns) wts[threadIdx.x] = expf(lg[s0 + threadIdx.x] - mx) / den;
    __syncthreads();
    if (c < D)
      for (int64_t s = 0; s < ns; ++s) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (s0 + s) * D + c);
        const float w = wts[s];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = fmaf(w, v[e], a[e]);
      }
  }
  if (c < D) *reinterpret_cast<f32x4*>(out + b * D + c) = a;
}

// ---------------------------------------------------------------- K13 + K14
struct FuseWeights { float w[8]; };

__global__ __launch_bounds__(256) void zscore_fuse_kernel(const float* __restrict__ scores, int64_t plane_stride, int K,
                                                          FuseWeights fw, const int64_t* __restrict__ off, int64_t B,
                                                          float* __restrict__ out, float* __restrict__ pad_out) {
  // no FMA contraction: the reference does `scores += w * z` as separate torch ops (ensemble_module.py:102,107), and near-tied
  // candidates make the ranking sensitive to the last bit; both paths below then also give identical bits
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= B) return;
  const int64_t c0 = off[i], c1 = off[i + 1];
  const float cn = (float)(c1 - c0);
  float pad = 0.f;                                     // what the reference's dense matrix holds in a padded slot
  // Up to 256 candidates (every MIND impression but a handful): the scores of ALL planes are requested at once and held in
  // registers, the fused value is accumulated there and written once — one memory round trip per impression instead of
  // three dependent passes per plane plus a read-modify-write of `out` (the kernel is latency-bound: ~124 B per impression
  // and plane).  The arithmetic and its order are those of the loop below, so both paths give the same bits.
  constexpr int ZC = 4, ZK = 9;
  if (c1 - c0 <= 64 * ZC) {
    float v[ZK][ZC];
#pragma unroll
    for (int k = 0; k < ZK; ++k) {
      if (k >= K || (k > 0 && fw.w[k - 1] == 0.0f)) continue;
      const float* s = scores + (int64_t)k * plane_stride;
#pragma unroll
      for (int t = 0; t < ZC; ++t) { const int64_t j = c0 + lane + 64 * t; v[k][t] = j < c1 ? s[j] : 0.f; }
    }
    float acc[ZC] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < ZK; ++k) {
      if (k >= K) continue;
      const float wk = k == 0 ? 1.0f : fw.w[k - 1];
      if (k > 0 && wk == 0.0f) continue;
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < ZC; ++t) if (c0 + lane + 64 * t < c1) a += v[k][t];
      const float mean = wave_sum(a) / cn;
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < ZC; ++t) if (c0 + lane + 64 * t < c1) { const float d = v[k][t] - mean; q += d * d; }
      const float sd = sqrtf(wave_sum(q) / (cn - 1.0f));
#pragma unroll
      for (int t = 0; t < ZC; ++t) {
        const float z = (v[k][t] - mean) / sd;
        acc[t] = k == 0 ? z : acc[t] + wk * z;
      }
      const float zp = (0.0f - mean) / sd;
      pad = k == 0 ? zp : pad + wk * zp;
    }
#pragma unroll
    for (int t = 0; t < ZC; ++t) { const int64_t j = c0 + lane + 64 * t; if (j < c1) out[j] = acc[t]; }
    if (pad_out && lane == 0) pad_out[i] = pad;
    return;
  }
  for (int k = 0; k < K; ++k) {
    const float wk = k == 0 ? 1.0f : fw.w[k - 1];
    if (k > 0 && wk == 0.0f) continue;                 // ensemble_module.py:100,105: module skipped
    const float* s = scores + (int64_t)k * plane_stride;
    float a = 0.f;
    for (int64_t j = c0 + lane; j < c1; j += 64) a += s[j];
    const float mean = wave_sum(a) / cn;               // sum(scores, dim=1) / cand_size  (:147)
    float q = 0.f;
    for (int64_t j = c0 + lane; j < c1; j += 64) { const float d = s[j] - mean; q += d * d; }
    const float sd = sqrtf(wave_sum(q) / (cn - 1.0f)); // torch.std: unbiased; c == 1 -> NaN  (:143)
    for (int64_t j = c0 + lane; j < c1; j += 64) {
      const float z = (s[j] - mean) / sd;
      out[j] = k == 0 ? z : out[j] + wk * z;           // scores += w * z  (:102,:107)
    }
    // the z-score runs over the whole zero-padded row (:145-149), so a padded slot becomes (0 - mean) / std
    const float zp = (0.0f - mean) / sd;
    pad = k == 0 ? zp : pad + wk * zp;
  }
  if (pad_out && lane == 0) pad_out[i] = pad;
}

// ---------------------------------------------------------------- K15 ranking + nDCG@k
__device__ __forceinline__ bool ranks_before(float a, int ia, float b, int ib) {
  // descending, stable; NaN sorts first like torch.argsort(descending=True)
  const bool na = a != a, nb = b != b;
  if (na || nb) return na && (!nb || ia < ib);
  return a > b || (a == b && ia < ib);
}

__global__ __launch_bounds__(256) void rank_ndcg_kernel(const float* __restrict__ scores, const float* __restrict__ labels,
                                                        const int64_t* __restrict__ off, int64_t B, int k,
                                                        int32_t* __restrict__ topk, float* __restrict__ ndcg,
                                                        float* __restrict__ mrr) {
  constexpr int CAP = 512;
  __shared__ float ssm[4][CAP];
  __shared__ float lsm[4][CAP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + wave;
  if (i >= B) return;
  const int64_t c0 = off[i];
  const int c = (int)(off[i + 1] - c0);
  const float* s = scores + c0;
  const float* l = labels ? labels + c0 : nullptr;
  if (c <= CAP) {                                      // wave-private staging, no workgroup barrier
    for (int j = lane; j < c; j += 64) { ssm[wave][j] = s[j]; if (l) lsm[wave][j] = l[j]; }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    s = ssm[wave];
    if (l) l = lsm[wave];
  }
  if (topk)
    for (int r = lane; r < k; r += 64) if (r >= c) topk[i * k + r] = -1;
  float dcg = 0.f, idcg = 0.f, pos = 0.f;
  int best = 0x7fffffff;                               // rank of the best-ranked positive (RetrievalMRR)
  for (int a = lane; a < c; a += 64) {
    const float sa = s[a];
    int rank = 0;
    for (int j = 0; j < c; ++j) rank += ranks_before(s[j], j, sa, a) ? 1 : 0;
    if (rank < k && topk) topk[i * k + rank] = a;
    if (l) {
      const float la = l[a];
      if (rank < k) dcg += la / log2f((float)rank + 2.0f);
      if (la > 0.f) best = min(best, rank);
      int lrank = 0;
      for (int j = 0; j < c; ++j) lrank += ranks_before(l[j], j, la, a) ? 1 : 0;
      if (lrank < k) idcg += la / log2f((float)lrank + 2.0f);
      pos += la;
    }
  }
  if (ndcg && l) {
    dcg = wave_sum(dcg); idcg = wave_sum(idcg); pos = wave_sum(pos);
    if (lane == 0) ndcg[i] = pos == 0.f ? 0.f : dcg / idcg;   // empty_target_action="neg" -> 0
  }
  if (mrr && l) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, __shfl_xor(best, o, 64));
    if (lane == 0) mrr[i] = best == 0x7fffffff ? 0.f : 1.0f / (float)(best + 1);
  }
}

// ---------------------------------------------------------------- Phase C in ONE launch (round 4)
// EnsembleModule.forward over K module tables (reference manner/models/ensemble_module.py:95-151) + the ranking consumer
// (cr_module.py:267-273): per impression, for every active module gather-mean-dot over that module's table (score_rows_body: the very
// code of the stand-alone scorer), the per-impression z-score and the weighted fusion (the arithmetic of zscore_fuse_kernel, in its
// order), then the stable ranking, top-k, nDCG@k and MRR (the arithmetic of rank_ndcg_kernel, in its order) — ONE workgroup per
// impression, the K score planes live in LDS and never reach HBM (the three-kernel path writes and re-reads K + 1 planes and pays K + 2
// launches).  Outputs are BIT-IDENTICAL to manner_hip_score_late_fusion x K -> manner_hip_zscore_fuse -> manner_hip_rank_ndcg.
// Impressions with more than PC_CAP candidates (MIND caps an impression at 300) keep their planes in the caller's scratch and are
// fused and ranked by wave 0 alone with the stand-alone kernels' loops.
constexpr int PC_CAP = 320, PC_MAXK = 9;          // MIND caps an impression at 300 candidates; LDS per workgroup decides how many gathers a CU keeps in flight
struct TablePtrs { const float* t[PC_MAXK]; };

template <int NCB, int RIF>
__global__ __launch_bounds__(256) void score_fuse_rank_kernel(
    TablePtrs tabs, int K, int fuse, FuseWeights fw, int64_t n_rows, const int32_t* __restrict__ hist_idx, const int64_t* __restrict__ hist_off,
    const int32_t* __restrict__ cand_idx, const int64_t* __restrict__ cand_off, const float* __restrict__ labels, int topn,
    float* __restrict__ out, float* __restrict__ pad_out, int32_t* __restrict__ topk, float* __restrict__ ndcg, float* __restrict__ mrr,
    float* __restrict__ scratch, int64_t total, int32_t* __restrict__ status) {
  constexpr int D = NCB * 256;
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [5 D] of the scorer body | [max(K, 2)][PC_CAP] planes | fused | labels
  __shared__ int bestw[4];
  float* planes = sm + 5 * D;
  float* fsm = planes + (K > 2 ? K : 2) * PC_CAP;
  float* lsm = fsm + PC_CAP;
  float* dterm = planes;                                // the planes are dead once the fusion is done (a barrier lies between)
  float* iterm = planes + PC_CAP;
  const int64_t b = blockIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t c0 = cand_off[b], c1 = cand_off[b + 1];
  const int c = (int)(c1 - c0);
  const bool in_lds = c <= PC_CAP;
  for (int k = 0; k < K; ++k) {
    if (k > 0 && fw.w[k - 1] == 0.0f) continue;        // ensemble_module.py:100,105: module skipped
    __syncthreads();                                    // the previous module's user vector has been read by every wave
    float* dst = in_lds ? planes + k * PC_CAP : scratch + (int64_t)k * total + c0;
    score_rows_body<NCB, RIF>(tabs.t[k], n_rows, hist_idx, hist_off, nullptr, cand_idx, cand_off, b, sm, status,
                              [&](int64_t j, float v) { dst[j - c0] = v; });
  }
  __syncthreads();
  if (!in_lds) {
    // ---- rare: more than PC_CAP candidates.  Wave 0 runs the stand-alone kernels' loops over the scratch planes.
    if (wave != 0) return;
    __threadfence_block();
    float pad = 0.f;
    const float cn = (float)c;
    float* o = out + c0;
    if (!fuse) {
      for (int j = lane; j < c; j += 64) o[j] = scratch[c0 + j];
    } else {
#pragma clang fp contract(off)
      for (int k = 0; k < K; ++k) {
        const float wk = k == 0 ? 1.0f : fw.w[k - 1];
        if (k > 0 && wk == 0.0f) continue;
        const float* s = scratch + (int64_t)k * total + c0;
        float a = 0.f;
        for (int j = lane; j < c; j += 64) a += s[j];
        const float mean = wave_sum(a) / cn;
        float q = 0.f;
        for (int j = lane; j < c; j += 64) { const float d = s[j] - mean; q += d * d; }
        const float sd = sqrtf(wave_sum(q) / (cn - 1.0f));
        for (int j = lane; j < c; j += 64) {
          const float z = (s[j] - mean) / sd;
          o[j] = k == 0 ? z : o[j] + wk * z;
        }
        const float zp = (0.0f - mean) / sd;
        pad = k == 0 ? zp : pad + wk * zp;
      }
    }
    if (pad_out && lane == 0) pad_out[b] = pad;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    const float* s = o;
    const float* l = labels ? labels + c0 : nullptr;
    if (topk)
      for (int r = lane; r < topn; r += 64) if (r >= c) topk[b * topn + r] = -1;
    float dcg = 0.f, idcg = 0.f, pos = 0.f;
    int best = 0x7fffffff;
    for (int a = lane; a < c; a += 64) {
      const float sa = s[a];
      int rank = 0;
      for (int j = 0; j < c; ++j) rank += ranks_before(s[j], j, sa, a) ? 1 : 0;
      if (rank < topn && topk) topk[b * topn + rank] = a;
      if (l) {
        const float la = l[a];
        if (rank < topn) dcg += la / log2f((float)rank + 2.0f);
        if (la > 0.f) best = min(best, rank);
        int lrank = 0;
        for (int j = 0; j < c; ++j) lrank += ranks_before(l[j], j, la, a) ? 1 : 0;
        if (lrank < topn) idcg += la / log2f((float)lrank + 2.0f);
        pos += la;
      }
    }
    if (ndcg && l) {
      dcg = wave_sum(dcg); idcg = wave_sum(idcg); pos = wave_sum(pos);
      if (lane == 0) ndcg[b] = pos == 0.f ? 0.f : dcg / idcg;
    }
    if (mrr && l) {
#pragma unroll
      for (int o2 = 32; o2 > 0; o2 >>= 1) best = min(best, __shfl_xor(best, o2, 64));
      if (lane == 0) mrr[b] = best == 0x7fffffff ? 0.f : 1.0f / (float)(best + 1);
    }
    return;
  }
  // ---- z-score + fusion: wave 0, lane-strided sums and xor-butterfly reductions exactly as zscore_fuse_kernel (the reference does
  // `scores += w * z` as separate torch ops: no FMA contraction)
  if (wave == 0) {
#pragma clang fp contract(off)
    float pad = 0.f;
    const float cn = (float)c;
    if (!fuse) {
      for (int j = lane; j < c; j += 64) fsm[j] = planes[j];
    } else {
      for (int k = 0; k < K; ++k) {
        const float wk = k == 0 ? 1.0f : fw.w[k - 1];
        if (k > 0 && wk == 0.0f) continue;
        const float* s = planes + k * PC_CAP;
        float a = 0.f;
        for (int j = lane; j < c; j += 64) a += s[j];
        const float mean = wave_sum(a) / cn;               // sum(scores, dim=1) / cand_size  (:147)
        float q = 0.f;
        for (int j = lane; j < c; j += 64) { const float d = s[j] - mean; q += d * d; }
        const float sd = sqrtf(wave_sum(q) / (cn - 1.0f)); // torch.std: unbiased; c == 1 -> NaN  (:143)
        for (int j = lane; j < c; j += 64) {
          const float z = (s[j] - mean) / sd;
          fsm[j] = k == 0 ? z : fsm[j] + wk * z;           // scores += w * z  (:102,:107)
        }
        const float zp = (0.0f - mean) / sd;               // what the reference's dense matrix holds in a padded slot
        pad = k == 0 ? zp : pad + wk * zp;
      }
    }
    if (pad_out && lane == 0) pad_out[b] = pad;
  }
  if (labels)
    for (int j = threadIdx.x; j < c; j += 256) lsm[j] = labels[c0 + j];
  __syncthreads();
  for (int j = threadIdx.x; j < c; j += 256) out[c0 + j] = fsm[j];
  // ---- ranking: every thread ranks its candidates (integer counts: order-free); the per-candidate DCG terms go to LDS and wave 0
  // adds them per lane in rank_ndcg_kernel's order (candidate lane, lane + 64, ...) before the same butterfly reductions
  if (topk)
    for (int r = threadIdx.x; r < topn; r += 256) if (r >= c) topk[b * topn + r] = -1;
  int best = 0x7fffffff;
  for (int a = threadIdx.x; a < c; a += 256) {
    const float sa = fsm[a];
    int rank = 0;
    for (int j = 0; j < c; ++j) rank += ranks_before(fsm[j], j, sa, a) ? 1 : 0;
    if (rank < topn && topk) topk[b * topn + rank] = a;
    if (labels) {
      const float la = lsm[a];
      dterm[a] = rank < topn ? la / log2f((float)rank + 2.0f) : -1.0f;       // -1: "no term" (a term is never negative: labels >= 0)
      if (la > 0.f) best = min(best, rank);
      int lrank = 0;
      for (int j = 0; j < c; ++j) lrank += ranks_before(lsm[j], j, la, a) ? 1 : 0;
      iterm[a] = lrank < topn ? la / log2f((float)lrank + 2.0f) : -1.0f;
    }
  }
  if (!labels) return;
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) best = min(best, __shfl_xor(best, o2, 64));
  if (lane == 0) bestw[wave] = best;
  __syncthreads();
  if (wave != 0) return;
  float dcg = 0.f, idcg = 0.f, pos = 0.f;
  for (int a = lane; a < c; a += 64) {
    if (dterm[a] >= 0.f) dcg += dterm[a];
    if (iterm[a] >= 0.f) idcg += iterm[a];
    pos += lsm[a];
  }
  if (ndcg) {
    dcg = wave_sum(dcg); idcg = wave_sum(idcg); pos = wave_sum(pos);
    if (lane == 0) ndcg[b] = pos == 0.f ? 0.f : dcg / idcg;   // empty_target_action="neg" -> 0
  }
  if (mrr && lane == 0) {
    const int bb = min(min(bestw[0], bestw[1]), min(bestw[2], bestw[3]));
    mrr[b] = bb == 0x7fffffff ? 0.f : 1.0f / (float)(bb + 1);
  }
}

// ---------------------------------------------------------------- aspect Diversity / Personalization @k
// (reference manner/metrics/functional.py:8-28, 31-62, 65-70 over the per-impression grouping of
// manner/metrics/base.py:92-129).  One wave per impression, lane c owns aspect class c (<= 64 classes):
//   diversity       = H(class distribution of the top-k candidates) / ln(num_classes)
//                     (the reference divides the counts by num_classes, not k, before Categorical —
//                      which renormalises, so only the distribution matters)
//   personalization = sum_c min(top-k count_c, history count_c) / sum_c max(...)   (generalised Jaccard)
// Both are 0 for an impression whose candidate aspect labels sum to 0 (the metric base classes treat
// that as "no positive target", empty_target_action="neg" — reproduced as is).
__global__ __launch_bounds__(256) void aspect_metrics_kernel(const int32_t* __restrict__ topk, const int32_t* __restrict__ cand_aspect,
                                                             const int32_t* __restrict__ hist_aspect,
                                                             const int64_t* __restrict__ cand_off, const int64_t* __restrict__ hist_off,
                                                             int64_t B, int k, int num_classes, float* __restrict__ div,
                                                             float* __restrict__ pers) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= B) return;
  const int64_t c0 = cand_off[i], c1 = cand_off[i + 1];
  int any = 0;
  for (int64_t j = c0 + lane; j < c1; j += 64) any |= cand_aspect[j] != 0;
  any = __any(any);
  int top_cnt = 0, n_top = 0;
  for (int r = 0; r < k; ++r) {
    const int idx = topk[i * k + r];                    // wave-uniform
    if (idx < 0) break;
    ++n_top;
    top_cnt += cand_aspect[c0 + idx] == lane ? 1 : 0;
  }
  if (div) {
    float h = 0.f;
    if (top_cnt > 0 && lane < num_classes) { const float p = (float)top_cnt / (float)n_top; h = -p * logf(p); }
    h = wave_sum(h);
    if (lane == 0) div[i] = any ? h / logf((float)num_classes) : 0.f;
  }
  if (pers) {
    int hist_cnt = 0;
    for (int64_t j = hist_off[i]; j < hist_off[i + 1]; ++j) hist_cnt += hist_aspect[j] == lane ? 1 : 0;
    float mn = lane < num_classes ? (float)min(top_cnt, hist_cnt) : 0.f;
    float mxv = lane < num_classes ? (float)max(top_cnt, hist_cnt) : 0.f;
    mn = wave_sum(mn); mxv = wave_sum(mxv);
    if (lane == 0) pers[i] = any ? mn / mxv : 0.f;
  }
}

// the fused scorer's launch: whole-row kernels for the widths of the shipped PLMs, the column-block kernel otherwise
void launch_score(const float* table, int64_t n_rows, int D, const int32_t* hist_idx, const int64_t* hist_off, const float* user,
                  const int32_t* cand_idx, const int64_t* cand_off, int64_t B, float* out, int32_t* status, hipStream_t s) {
  const size_t lds = 5 * (size_t)D * sizeof(float);
  // A/B switch, read per launch (tests flip it): 1 = the column-block kernel for every width.  Two rows in flight per wave: 4 and
  // 6 measured the same or slower (the CU's vector-memory path is saturated at ~74 GB/s per CU either way) at twice the registers.
  const char* g = getenv("MANNER_HIP_SCORER_GENERIC");
  const bool generic = g && atoi(g) != 0;
  if (D == 768 && !generic)
    hipLaunchKernelGGL((score_late_fusion_rows_kernel<3, 2>), dim3((unsigned)B), dim3(256), lds, s, table, n_rows, hist_idx, hist_off,
                       user, cand_idx, cand_off, out, status);
  else if (D == 1024 && !generic)
    hipLaunchKernelGGL((score_late_fusion_rows_kernel<4, 2>), dim3((unsigned)B), dim3(256), lds, s, table, n_rows, hist_idx, hist_off,
                       user, cand_idx, cand_off, out, status);
  else
    hipLaunchKernelGGL(score_late_fusion_kernel, dim3((unsigned)B), dim3(256), lds, s, table, n_rows, D, hist_idx, hist_off, user,
                       cand_idx, cand_off, out, status);
}

}  // namespace
}  // namespace manner

using namespace manner;

extern "C" {

int manner_hip_score_late_fusion(const float* table, int64_t n_rows, int32_t D, const int32_t* hist_idx,
                                 const int64_t* hist_off, const int32_t* cand_idx, const int64_t* cand_off, int64_t B,
                                 float* out, int32_t* status, manner_hip_stream_t stream) {
  if (B == 0) return MANNER_HIP_OK;
  if (!table || !hist_idx || !hist_off || !cand_idx || !cand_off || !out) return fail(MANNER_HIP_E_INVALID, "score_late_fusion: null pointer");
  // 5 D floats of dynamic LDS must stay within the 64 KiB a kernel gets without raising its limit
  if (D <= 0 || D % 4 || D > 3072 || n_rows <= 0) return fail(MANNER_HIP_E_INVALID, "score_late_fusion: D=%d must be a multiple of 4, <= 3072", D);
  launch_score(table, n_rows, D, hist_idx, hist_off, nullptr, cand_idx, cand_off, B, out, status, (hipStream_t)stream);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_score_late_fusion_f16(const void* table16, const float* mean, int64_t n_rows, int32_t D, const int32_t* hist_idx,
                                     const int64_t* hist_off, const int32_t* cand_idx, const int64_t* cand_off, int64_t B,
                                     float* out, int32_t* status, manner_hip_stream_t stream) {
  if (B == 0) return MANNER_HIP_OK;
  if (!table16 || !hist_idx || !hist_off || !cand_idx || !cand_off || !out) return fail(MANNER_HIP_E_INVALID, "score_late_fusion_f16: null pointer");
  if (D <= 0 || D % 8 || D > 3072 || n_rows <= 0 || (uintptr_t)table16 % 16) return fail(MANNER_HIP_E_INVALID, "score_late_fusion_f16: D=%d must be a multiple of 8, <= 3072, rows 16-byte aligned", D);
  const char* g = getenv("MANNER_HIP_SCORER_GENERIC");                       // A/B switch, as in launch_score
  const bool generic = g && atoi(g) != 0;
  const f16_t* t16 = static_cast<const f16_t*>(table16);
  if (D == 768 && !generic)
    hipLaunchKernelGGL((score_late_fusion_f16_rows_kernel<3, 2>), dim3((unsigned)B), dim3(256), 5 * D * sizeof(float), (hipStream_t)stream,
                       t16, n_rows, hist_idx, hist_off, cand_idx, cand_off, out, status, mean);
  else if (D == 1024 && !generic)
    hipLaunchKernelGGL((score_late_fusion_f16_rows_kernel<4, 2>), dim3((unsigned)B), dim3(256), 5 * D * sizeof(float), (hipStream_t)stream,
                       t16, n_rows, hist_idx, hist_off, cand_idx, cand_off, out, status, mean);
  else
    hipLaunchKernelGGL(score_late_fusion_f16_kernel, dim3((unsigned)B), dim3(256), 5 * D * sizeof(float), (hipStream_t)stream,
                       t16, n_rows, D, hist_idx, hist_off, cand_idx, cand_off, out, status, mean);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

size_t manner_hip_table_to_f16_workspace_bytes(int32_t D) { return D > 0 ? (size_t)MEAN_BLOCKS * D * sizeof(float) + 256 : 0; }

int manner_hip_table_to_f16(const float* table, int64_t n_rows, int32_t D, float* mean, void* table16, void* workspace,
                            size_t workspace_bytes, manner_hip_stream_t stream) {
  if (n_rows == 0) return MANNER_HIP_OK;
  if (!table || !table16 || n_rows < 0 || D <= 0 || D % 4) return fail(MANNER_HIP_E_INVALID, "table_to_f16: bad argument (D %% 4 == 0)");
  hipStream_t s = (hipStream_t)stream;
  if (!mean) return convert_f32_to_16(DT_F16, table, table16, n_rows * D, s);
  if (!workspace || workspace_bytes < manner_hip_table_to_f16_workspace_bytes(D)) return fail(MANNER_HIP_E_WORKSPACE, "table_to_f16: workspace too small");
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(col_partial_kernel, dim3((unsigned)((D + 255) / 256), MEAN_BLOCKS), dim3(256), 0, s, table, n_rows, D, part);
  hipLaunchKernelGGL(col_mean_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, s, part, n_rows, D, mean);
  const int64_t blocks = (n_rows * D / 4 + 255) / 256;
  hipLaunchKernelGGL(centre_f16_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, table, mean, n_rows * D, D,
                     static_cast<f16_t*>(table16));
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_score_user(const float* table, int64_t n_rows, int32_t D, const float* user, const int32_t* cand_idx,
                          const int64_t* cand_off, int64_t B, float* out, int32_t* status, manner_hip_stream_t stream) {
  if (B == 0) return MANNER_HIP_OK;
  if (!table || !user || !cand_idx || !cand_off || !out) return fail(MANNER_HIP_E_INVALID, "score_user: null pointer");
  if (D <= 0 || D % 4 || D > 3072 || n_rows <= 0) return fail(MANNER_HIP_E_INVALID, "score_user: D=%d must be a multiple of 4, <= 3072", D);
  launch_score(table, n_rows, D, nullptr, nullptr, user, cand_idx, cand_off, B, out, status, (hipStream_t)stream);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_to_dense(const float* x, const int64_t* off, int64_t B, int64_t width, int32_t D, const float* fill,
                        float* dense, uint8_t* mask, manner_hip_stream_t stream) {
  if (B * width == 0) return MANNER_HIP_OK;
  if (!x || !off || !dense || B < 0 || width < 0 || D < 1) return fail(MANNER_HIP_E_INVALID, "to_dense: bad argument");
  if (B * width > 0x7fffffffll * 4) return fail(MANNER_HIP_E_INVALID, "to_dense: %lld x %lld slots exceed the grid", (long long)B, (long long)width);
  if (D % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)dense % 16 == 0)
    hipLaunchKernelGGL(to_dense_rows_kernel, dim3((unsigned)((B * width + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, off, B,
                       width, D, fill, dense, mask);
  else
    hipLaunchKernelGGL(to_dense_scalar_kernel, dim3((unsigned)((B * width + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, off,
                       B, width, D, fill, dense, mask);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_dot(const float* user, const float* cand, int64_t B, int64_t C, int32_t D, int64_t sb, int64_t sd,
                   int64_t sc, float* out, manner_hip_stream_t stream) {
  if (B * C == 0) return MANNER_HIP_OK;
  if (!user || !cand || !out || D <= 0) return fail(MANNER_HIP_E_INVALID, "dot: bad argument");
  if (sd == 1) {
    const int vec_ok = (D % 4 == 0) && (sb % 4 == 0) && (sc % 4 == 0) && ((uintptr_t)cand % 16 == 0) && ((uintptr_t)user % 16 == 0);
    hipLaunchKernelGGL(dot_rows_kernel, dim3((unsigned)((B * C + 3) / 4)), dim3(256), 0, (hipStream_t)stream, user, cand, B, C,
                       D, sb, sc, out, vec_ok);
  } else {
    hipLaunchKernelGGL(dot_strided_kernel, dim3((unsigned)((B * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, user, cand,
                       B, C, D, sb, sd, sc, out);
  }
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_additive_pool(const float* x, const float* lin_w, const float* lin_b, const float* query, int64_t B,
                             int64_t S, int32_t D, int32_t Q, float* out, float* scratch, manner_hip_stream_t stream) {
  if (B == 0) return MANNER_HIP_OK;
  if (!x || !lin_w || !lin_b || !query || !out || !scratch) return fail(MANNER_HIP_E_INVALID, "additive_pool: null pointer");
  if (S <= 0 || D <= 0 || D % 4 || D > 8192 || Q <= 0) return fail(MANNER_HIP_E_INVALID, "additive_pool: S=%lld D=%d Q=%d unsupported", (long long)S, D, Q);
  if (B > 0x7fffffffll || B * S > 0x7fffffffll) return fail(MANNER_HIP_E_INVALID, "additive_pool: B*S exceeds the grid");
  static const bool force_valu = getenv("MANNER_HIP_POOL_VALU") != nullptr;      // A/B switch for development
  if (!force_valu && D % 32 == 0 && Q <= 256 && (uintptr_t)x % 16 == 0 && (uintptr_t)lin_w % 16 == 0) {
    int rc = pool_logits_mfma(x, lin_w, lin_b, query, B * S, D, Q, scratch, (hipStream_t)stream);
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(pool_logits_valu_kernel, dim3((unsigned)(B * S)), dim3(256), (D + 4) * sizeof(float), (hipStream_t)stream,
                       x, lin_w, lin_b, query, D, Q, scratch);
    MANNER_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(pool_apply_kernel, dim3((unsigned)B, (unsigned)((D + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, x,
                     scratch, S, D, out);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

size_t manner_hip_additive_pool_workspace_bytes(int64_t B, int64_t S, int32_t D, int32_t Q) {
  const size_t logits = (size_t)(B > 0 ? B : 0) * (size_t)(S > 0 ? S : 1) * sizeof(float) + 256;
  const size_t packed = (D > 0 && D % 32 == 0 && Q > 0 && Q <= 320) ? pool_fused_workspace_bytes(D, Q) : 0;
  return logits > packed ? logits : packed;
}

int manner_hip_additive_pool_fused(const float* x, const float* lin_w, const float* lin_b, const float* query, int64_t B, int64_t S,
                                   int32_t D, int32_t Q, float* out, void* workspace, size_t workspace_bytes, int32_t strict,
                                   manner_hip_stream_t stream) {
  if (B == 0) return MANNER_HIP_OK;
  if (!x || !lin_w || !lin_b || !query || !out || !workspace) return fail(MANNER_HIP_E_INVALID, "additive_pool_fused: null pointer");
  if (workspace_bytes < manner_hip_additive_pool_workspace_bytes(B, S, D, Q)) return fail(MANNER_HIP_E_WORKSPACE, "additive_pool_fused: workspace too small");
  const char* env = getenv("MANNER_HIP_POOL_STRICT");                      // read per call (A/B in one process)
  const bool want_strict = strict != 0 || (env && *env && *env != '0');
  if (!want_strict && S > 0 && D > 0 && Q > 0 && pool_fused_supported(B, S, D, Q) && (uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0 &&
      (uintptr_t)workspace % 256 == 0)
    return pool_fused(x, lin_w, lin_b, query, B, S, D, Q, out, workspace, workspace_bytes, (hipStream_t)stream);
  return manner_hip_additive_pool(x, lin_w, lin_b, query, B, S, D, Q, out, static_cast<float*>(workspace), stream);
}

int manner_hip_zscore_fuse(const float* scores, int64_t plane_stride, int32_t K, const float* weights, const int64_t* cand_off,
                           int64_t B, float* out, float* pad_value, manner_hip_stream_t stream) {
  if (B == 0) return MANNER_HIP_OK;
  if (!scores || !cand_off || !out || K < 1 || K > 9 || (K > 1 && !weights)) return fail(MANNER_HIP_E_INVALID, "zscore_fuse: bad argument (1 <= K <= 9)");
  FuseWeights fw;
  for (int k = 0; k < 8; ++k) fw.w[k] = k < K - 1 ? weights[k] : 0.f;
  hipLaunchKernelGGL(zscore_fuse_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, scores, plane_stride, K,
                     fw, cand_off, B, out, pad_value);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_aspect_metrics(const int32_t* topk_idx, const int32_t* cand_aspect, const int32_t* hist_aspect,
                              const int64_t* cand_off, const int64_t* hist_off, int64_t B, int32_t k, int32_t num_classes,
                              float* diversity, float* personalization, manner_hip_stream_t stream) {
  if (B == 0) return MANNER_HIP_OK;
  if (!topk_idx || !cand_aspect || !cand_off || k < 1 || num_classes < 2 || num_classes > 64 || (personalization && (!hist_aspect || !hist_off)))
    return fail(MANNER_HIP_E_INVALID, "aspect_metrics: bad argument (2 <= num_classes <= 64)");
  hipLaunchKernelGGL(aspect_metrics_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, topk_idx, cand_aspect,
                     hist_aspect, cand_off, hist_off, B, k, num_classes, diversity, personalization);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

size_t manner_hip_score_fuse_rank_workspace_bytes(int32_t K, int64_t total_candidates) {
  return (size_t)(K > 0 ? K : 0) * (size_t)(total_candidates > 0 ? total_candidates : 0) * sizeof(float) + 256;
}

int manner_hip_score_fuse_rank(const float* const* tables, int32_t K, const float* weights, int64_t n_rows, int32_t D, const int32_t* hist_idx,
                               const int64_t* hist_off, const int32_t* cand_idx, const int64_t* cand_off, int64_t B, int64_t total_candidates,
                               const float* labels, int32_t k, float* scores, float* pad_value, int32_t* topk_idx, float* ndcg, float* mrr,
                               void* workspace, size_t workspace_bytes, int32_t* status, manner_hip_stream_t stream) {
  if (B == 0) return MANNER_HIP_OK;
  if (!tables || K < 1 || K > PC_MAXK || (K > 1 && !weights) || !hist_idx || !hist_off || !cand_idx || !cand_off || !scores || k < 1 ||
      ((ndcg || mrr) && !labels) || !workspace)
    return fail(MANNER_HIP_E_INVALID, "score_fuse_rank: bad argument (1 <= K <= %d)", PC_MAXK);
  if (D != 768 && D != 1024) return fail(MANNER_HIP_E_INVALID, "score_fuse_rank: D=%d (768 or 1024; other widths: the separate kernels)", D);
  if (n_rows <= 0 || total_candidates < 0) return fail(MANNER_HIP_E_INVALID, "score_fuse_rank: bad sizes");
  if (workspace_bytes < manner_hip_score_fuse_rank_workspace_bytes(K, total_candidates)) return fail(MANNER_HIP_E_WORKSPACE, "score_fuse_rank: workspace too small");
  TablePtrs tp{};
  FuseWeights fw{};
  for (int i = 0; i < K; ++i) {
    if (!tables[i]) return fail(MANNER_HIP_E_INVALID, "score_fuse_rank: table %d is NULL", i);
    tp.t[i] = tables[i];
    if (i > 0) fw.w[i - 1] = weights[i - 1];
  }
  const int fuse = K > 1 ? 1 : 0;                        // one table: the CR-Module's raw late-fusion scores (cr_module.py:105-131)
  const size_t lds = (size_t)(5 * D + ((K > 2 ? K : 2) + 2) * PC_CAP) * sizeof(float);
  float* scratch = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
  if (D == 768)
    hipLaunchKernelGGL((score_fuse_rank_kernel<3, 2>), dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, tp, K, fuse, fw, n_rows, hist_idx,
                       hist_off, cand_idx, cand_off, labels, k, scores, pad_value, topk_idx, ndcg, mrr, scratch, total_candidates, status);
  else
    hipLaunchKernelGGL((score_fuse_rank_kernel<4, 2>), dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, tp, K, fuse, fw, n_rows, hist_idx,
                       hist_off, cand_idx, cand_off, labels, k, scores, pad_value, topk_idx, ndcg, mrr, scratch, total_candidates, status);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int manner_hip_rank_ndcg(const float* scores, const float* labels, const int64_t* cand_off, int64_t B, int32_t k,
                         int32_t* topk_idx, float* ndcg, float* mrr, manner_hip_stream_t stream) {
  if (B == 0) return MANNER_HIP_OK;
  if (!scores || !cand_off || k < 1 || ((ndcg || mrr) && !labels)) return fail(MANNER_HIP_E_INVALID, "rank_ndcg: bad argument");
  hipLaunchKernelGGL(rank_ndcg_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, scores, labels, cand_off,
                     B, k, topk_idx, ndcg, mrr);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

}  // extern "C"
