// K11 additive-attention pooler, ONE pass over x (round 4) — AdditiveAttention.forward, reference
// manner/models/components/attention.py:21-27 (and through it NAMLUserEncoder.forward, user_encoder.py:17-21):
//
//   logit[b,s] = sum_j tanh(<x[b,s], W[j]> + bias[j]) q[j]      (no padding mask: quirk Q2)
//   out[b]     = sum_s softmax_s(logit[b,:]) x[b,s]
//
// The two-pass path (gemm.hip pool_logits_kernel on the f32 matrix pipe + scoring.hip pool_apply_kernel) reads x twice and is bound by
// the f32 MFMA rate (155 TF): 1.05 ms at B = 4096, S = 50, D = 768, Q = 200 — 0.15 of the HBM roofline this operator belongs to.
// Here a workgroup keeps its rows of x ON THE CU for both uses:
//
//   * NW = 4 waves (S <= 64: two workgroups per CU, so one's loads and epilogue run under the other's matrix work) or 8 waves (S <= 128),
//     256 registers each.  A wave owns one strip of 16 rows of x; a lane (r = lane & 15, g = lane >> 4) holds, for every 32-column step
//     t, the 8 values x[row r][32t + 4g + 0..3] and x[row r][32t + 16 + 4g + 0..3] — the B-operand layout of v_mfma_f32_16x16x32_f16
//     (which 8 of a step's 32 k indices a lane group takes is free as long as the A operand agrees) chosen so that each wave load
//     instruction covers whole 64-byte pieces of 16 rows; all 48 loads of a lane are in flight at once.
//   * SPLIT OPERANDS WITH POWER-OF-TWO SCALES.  Every row of x is scaled by 2^kx (kx from the row's largest magnitude, so that the
//     scaled row peaks in [2^13, 2^14)) and kept as an IEEE-half pair hi = f16(v), lo = f16(v - hi): 22 significant bits, 4 bytes per
//     value, 192 VGPRs for 768 columns; W is scaled by ONE 2^kw (from max |W|) and split the same way.  The scales are exact and
//     undone in f32 (z = acc 2^-(kx+kw), weighted sum from (hi + lo) 2^-kx), so neither of DESIGN r3's objections to an f16 split
//     holds: nothing can overflow (|scaled| < 2^14 << 65504) and the low parts of every element within 2^-10 of its row's / the
//     matrix' maximum are normal numbers; smaller elements lose relative, not absolute, precision.  (The plain bf16 split the same
//     kernel started with — 16 significant bits, what VERDICT r3 proposed — measured 1.3e-4 on a peaked softmax; this one 1.1e-5,
//     the strict f32 path 1.3e-5 on the same input: both sit at the oracle's own rounding noise.)
//   * the products run as x3 on the f16 matrix pipe: W.hi x.hi + W.hi x.lo + W.lo x.hi, f32 accumulation (the dropped lo.lo term is
//     2^-22 relative).  W is laid out in fragment order once per call by pool_pack_w_kernel (a few hundred KB, L2 resident) and streamed
//     through an LDS ring by LDS-DMA, shared by the waves: 16 x TP units per pass, ceil(Q / (16 TP)) passes over the resident x
//     (hipcc keeps MFMA A / B operands in arch VGPRs, at most 256: 192 operand registers + two fragment buffers leave TP = 5
//     accumulator tiles; a 4-wave variant with two strips per wave — 384 operand registers — spilled 430 of them).  A ring stage
//     holds SPS = 2 k-steps, ONE s_barrier per stage; the fragments are read by INLINE-ASM ds_read_b128 one slot ahead of their use
//     with counted lgkmcnt waits (for a C++ LDS load hipcc cannot tell the stage being read from the stages the DMA is filling and
//     waits vmcnt(0): 1 600 cycles per step instead of 1 100); the next stage's DMA pieces are issued after the stage's matrix work.
//   * after each pass: tanh (hardware exp + rcp) . q summed over the pass's units in the accumulator layout; after the last:
//     softmax over the rows of a batch element (its strips sit in several waves: logits meet in LDS), then the weighted sum
//     from the RESIDENT registers on the matrix pipe again: the fragment as A operand against a 0 / 1 selection matrix gives
//     2^kx x[row][16 columns] exactly in the accumulator layout, where a lane holds four rows of one column — four FMAs with the
//     rows' softmax weights (a DPP reduction in the operand layout took 8 VALU per value) — and the lane groups' / strips' partial
//     sums meet in LDS in a fixed order (deterministic).
//
// A batch element occupies SP = 1, 2, 4 or 8 strips (16 SP >= S rows; rows past S and batch elements past B are zero rows whose logit
// is -inf), a workgroup NW / SP batch elements.  x is read from HBM exactly once (PMC: 1.03x the algorithmic bytes), nothing but
// out[B, D] is written.  Shapes other than D = 768 with S <= 128 and Q <= 320, or unaligned pointers, take the two-pass path, which
// also stays as the strict-f32 mode (MANNER_HIP_POOL_STRICT=1).
//
// Measured (B = 4096, S = 50, D = 768, Q = 200): 0.41-0.43 ms against 1.0 ms for the two-pass path; by unit count 0.18 ms + 0.10 ms per
// full pass of 5 unit tiles.  In-kernel s_memtime sums (the MANNER_POOL_DIAG build, tools/pool_diag.sh; 4-wave workgroups): of a
// wave's ~100 k cycles 19 % wait for its 48 KiB of x (every CU asks for 384 KiB at once), 4 % scale + split, 9 % logits / softmax /
// weighted sum, and 68 % are the three passes — per ring stage (2 k-steps, 480 cycles of the wave's own MFMAs): 730 cycles fragment
// reads + MFMAs (the SIMD partner's MFMAs included), **1 020 cycles ISSUING the stage's five 1 KiB LDS-DMA pieces** (180-200 cycles of
// wave time per global_load_lds, the same whether the pieces go out right after the barrier or after the matrix work), 130 in the
// barrier, 120 waiting for the pieces.  The W stream is what bounds the kernel — through the issue cost of LDS-DMA, not through L2
// bandwidth (8-wave workgroups halve the pieces per wave and the L2 -> LDS bytes, and wait that much longer in the barrier: same
// time).  Staging W through registers instead (global_load + ds_write: ~25 cycles of issue per piece) needs 10-20 VGPRs per wave that
// do not exist next to 192 operand registers under hipcc's 256-arch-VGPR operand limit; fewer W bytes per MFMA needs more rows per
// workgroup — the same limit.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace manner {
namespace {

constexpr int PF_TP = 5;           // unit tiles (of 16) per pass
constexpr int PF_MAX_PASS = 4;     // Q <= 16 TP MAX_PASS = 320

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 16 lanes of a DPP row (lanes 16g .. 16g + 15); every lane of the row ends with the sum
__device__ __forceinline__ float rowsum16(float v) {
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);    // row_half_mirror
  v += dpp_mov<0x140>(v);    // row_mirror
  return v;
}
__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// k with 2^k * m in [2^13, 2^14) for a finite m > 0 (k clamped to [-114, 126]); 0 for m = 0, inf or nan
__device__ __forceinline__ int scale_exponent(float m) {
  const int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 0xffu);
  if (e == 0 || e == 255) return 0;
  const int k = 140 - e;
  return k > 126 ? 126 : k;
}
__device__ __forceinline__ float pow2i(int k) { return __builtin_bit_cast(float, (unsigned)(127 + k) << 23); }   // -126 <= k <= 127

// max |W| over the whole matrix (the bits of a non-negative float order like unsigned integers: one atomicMax per workgroup into a
// zeroed word); pool_pack_w_kernel turns it into the exponent kw of W's power-of-two scale
__global__ __launch_bounds__(256) void pool_w_max_kernel(const float* __restrict__ W, int64_t n, uint32_t* __restrict__ wmax_bits) {
  __shared__ float red[4];
  float m = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(W + i);                       // n % 4 == 0 (D % 32 == 0), 16-byte aligned rows
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));    // fmaxf drops NaNs: a NaN weight gives NaN products anyway
  }
  m = wave_max64(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(wmax_bits, __builtin_bit_cast(uint32_t, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// W [Q, D] f32 -> fragment-ordered f16 hi / lo of 2^kw W:  Wp[step = pass * KS + t][slot < TP][part: 0 hi, 1 lo][lane 64][8]  (16 bytes per
// lane and fragment: one LDS-DMA piece per (step, slot, part), read back with one ds_read_b128 per lane), unit = 16 (pass TP + slot) + (lane & 15),
// columns 32 t + 4 g + e (e < 4) and 32 t + 16 + 4 g + e - 4 (e >= 4), g = lane >> 4; units >= Q are zero.  bq[unit] = {bias, q} (zero-padded).
__global__ __launch_bounds__(256) void pool_pack_w_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                          const float* __restrict__ query, int Q, int D, int KS, int n_pass,
                                                          const uint32_t* __restrict__ wmax_bits, int32_t* __restrict__ kw_out,
                                                          f16x8* __restrict__ Wp, float2* __restrict__ bq) {
  const int kw = scale_exponent(__builtin_bit_cast(float, *wmax_bits));
  if (blockIdx.x == 0 && threadIdx.x == 0) *kw_out = kw;
  const float sw = pow2i(kw);
  const int64_t n_frag = (int64_t)n_pass * KS * PF_TP * 64;        // (step, slot, lane) triples
  for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < n_frag; f += (int64_t)gridDim.x * 256) {
    const int lane = (int)(f & 63);
    const int64_t ss = f >> 6;
    const int slot = (int)(ss % PF_TP);
    const int64_t step = ss / PF_TP;
    const int t = (int)(step % KS), pass = (int)(step / KS);
    const int unit = 16 * (pass * PF_TP + slot) + (lane & 15), g = lane >> 4;
    f16x8 hi, lo;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 32 * t + (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
      v[e] = unit < Q ? W[(int64_t)unit * D + c] * sw : 0.f;
      hi[e] = (f16_t)v[e];
    }
    asm volatile("" : "+v"(hi));                       // the remainder is taken against THESE bits (gemm.hip's FFN1-epilogue lesson)
#pragma unroll
    for (int e = 0; e < 8; ++e) lo[e] = (f16_t)(v[e] - (float)hi[e]);
    Wp[(ss * 2 + 0) * 64 + lane] = hi;
    Wp[(ss * 2 + 1) * 64 + lane] = lo;
  }
  const int n_units = n_pass * PF_TP * 16;
  for (int u = blockIdx.x * 256 + threadIdx.x; u < n_units; u += gridDim.x * 256)
    bq[u] = u < Q ? make_float2(bias[u], query[u]) : make_float2(0.f, 0.f);
}

#ifndef POOL_EARLY_ALL
#define POOL_EARLY_ALL 0
#endif
#ifdef MANNER_POOL_DIAG   // diagnostic build only (tools/pool_diag.sh): in-kernel s_memtime sums per phase of the pass loop
#define PD_DECL unsigned long long pd_t[6] = {0, 0, 0, 0, 0, 0}, pd_a = 0, pd_b = 0
#define PD_A() pd_a = __builtin_amdgcn_s_memtime()
#define PD_B(i) do { pd_b = __builtin_amdgcn_s_memtime(); pd_t[i] += pd_b - pd_a; pd_a = pd_b; } while (0)
#else
#define PD_DECL
#define PD_A()
#define PD_B(i)
#endif

template <int OFF>
__device__ __forceinline__ void lds_read128(f16x8& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}

// PROD (lab, MANNER_HIP_POOL_PRODUCER=1 — round 5's producer / consumer experiment, TIMING ONLY: the last wave's strip is not computed):
// the last wave of the workgroup issues EVERY LDS-DMA piece of the ring and runs no matrix work; the other waves never issue.
template <int KS, int NW, int R, int SPS, bool PROD = false>
__global__ __launch_bounds__(64 * NW, 2) void pool_fused_kernel(
    const float* __restrict__ x, const f16x8* __restrict__ Wp, const float2* __restrict__ bq, const int32_t* __restrict__ kw_p, int n_pass,
    int n_tiles, int64_t B, int S, int SP, float* __restrict__ out, unsigned long long* __restrict__ diag) {
  constexpr int TP = PF_TP, D = 32 * KS, NT = 64 * NW;
  PD_DECL;
  constexpr int STEP_B = TP * 2 * 1024;                // bytes of one k-step of packed W: TP slots x {hi, lo} x 1 KiB
  constexpr int STAGE = SPS * STEP_B;                  // a ring stage = SPS k-steps: ONE barrier per stage
  constexpr int PCS = SPS * 2 * TP;                    // LDS-DMA pieces (1 KiB) per stage
  constexpr int P_LO = PCS / NW, N_HI = PCS % NW;      // per stage: waves < N_HI issue P_LO + 1 pieces, the others P_LO
  constexpr int KST = KS / SPS;                        // stages per pass
  static_assert(KS % SPS == 0 && KST % R == 0, "the ring stage of a step must not depend on the pass");
  static_assert(4 * NW * D * 4 <= R * STAGE, "the partial sums (4 lane groups per wave) reuse the ring");
  static_assert((R - 1) * (P_LO + 1) < 60, "vmcnt is a 6-bit counter");
  static_assert(!PROD || (R - 2) * PCS < 60, "the producer keeps R - 2 whole stages in flight");
  __shared__ __attribute__((aligned(1024))) char ring[R * STAGE];
  __shared__ float2 bq_s[PF_MAX_PASS * TP * 16];
  __shared__ float lg[NW * 16];

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, r = lane & 15, g = lane >> 4;
  const int BPW = NW / SP;                              // batch elements per workgroup
  const int64_t b0 = (int64_t)blockIdx.x * BPW;
  const int total = n_pass * KST;                       // ring steps (stages)
  const uint32_t ring_lds = (uint32_t)(size_t)LDS_PTR(ring);      // the ring's byte address inside LDS

  const bool is_prod = PROD && wave == NW - 1;
  auto issue = [&](int gs) {                            // this wave's LDS-DMA pieces of ring step gs
    const char* src = reinterpret_cast<const char*>(Wp) + (size_t)gs * STAGE + lane * 16;
    char* dst = ring + (gs % R) * STAGE;
    if (PROD) {
      if (!is_prod) return;
#pragma unroll 1
      for (int pc = 0; pc < PCS; ++pc)
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + pc * 1024), LDS_PTR(dst + pc * 1024), 16, 0, 0);
      return;
    }
    for (int pc = wave; pc < PCS; pc += NW)
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + pc * 1024), LDS_PTR(dst + pc * 1024), 16, 0, 0);
  };
#pragma unroll 1
  for (int gs = 0; gs < R - 1; ++gs)
    if (gs < total) issue(gs);
  for (int u = tid; u < n_pass * TP * 16; u += NT) bq_s[u] = bq[u];
  const float inv_w = pow2i(-*kw_p);

  // ---- x: the wave's strip, straight into the operand layout: every load of the lane in flight at once (48 KiB per wave), then the
  // row's scale, then the split in place, step by step (the scheduling barriers keep the landing and the packed registers from
  // being live together: 192 + 192 would spill).
  const int64_t b = b0 + wave / SP;
  const int s_row = 16 * (wave % SP) + r;
  const bool valid = b < B && s_row < S;
  const float* rowp = x + (valid ? (b * S + s_row) * (int64_t)D : 0) + 4 * g;
  f32x4 land[KS][2];
#pragma unroll
  for (int t = 0; t < KS; ++t) {
    land[t][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    land[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (valid) {
      land[t][0] = *reinterpret_cast<const f32x4*>(rowp + 32 * t);
      land[t][1] = *reinterpret_cast<const f32x4*>(rowp + 32 * t + 16);
    }
  }
  float amax = 0.f;
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(land[t][0][e]), fabsf(land[t][1][e])));
  amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 32, 64));         // the row's largest magnitude (its four lane groups hold its four column quarters)
  const int kx = scale_exponent(amax);
  const float sx = pow2i(kx), inv_x = pow2i(-kx);
  f16x8 xh[KS], xl[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t) {
    __builtin_amdgcn_sched_barrier(0);
    float v[8];
    f16x8 h8, l8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = (e < 4 ? land[t][0][e] : land[t][1][e - 4]) * sx;
      h8[e] = (f16_t)v[e];
    }
    asm volatile("" : "+v"(h8));                        // the remainder is taken against THESE bits
#pragma unroll
    for (int e = 0; e < 8; ++e) l8[e] = (f16_t)(v[e] - (float)h8[e]);
    xh[t] = h8;
    xl[t] = l8;
  }
  __builtin_amdgcn_sched_barrier(0);

  float p = 0.f;
  // One pass = NSL unit tiles over the resident x.  NSL is a compile-time count (a per-slot branch inside the unrolled steps makes hipcc
  // spill 1600 registers): full passes run NSL = TP, a short last pass the smallest of {4, 3, 2} that holds its tiles (the packed W has
  // zero tiles there; the ring still streams whole stages).
  auto run_pass = [&](auto nsl_c, int pass) __attribute__((always_inline)) {
    constexpr int NSL = decltype(nsl_c)::value;
    f32x4 acc[NSL];
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) acc[sl] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KST; ++u) {
      const int gs = pass * KST + u;
      PD_A();
      // this wave's pieces of stage gs have landed (the pieces of the R - 2 younger stages may fly)
      if (PROD) {
        // consumers hold no DMA of their own
      } else if (gs + R - 1 < total) {
        if (wave < N_HI) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * (P_LO + 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * P_LO) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      PD_B(0);
      __builtin_amdgcn_s_barrier();                     // everybody's pieces of gs landed; everybody finished reading stage gs - 1
      PD_B(1);
      // The stage of ring step gs - 1 (nobody reads it any more) takes step gs + R - 1.  8-wave workgroups: the two waves of a SIMD (w and
      // w + 4) issue their LDS-DMA pieces at OPPOSITE ends of the stage — waves 4-7 here, waves 0-3 after the matrix work — so that one
      // partner's issue time lies under the other's MFMAs instead of both stalling the pipe at the same moment.
      const bool early = PROD || POOL_EARLY_ALL || (NW == 8 && wave >= 4);
      if (early && gs + R - 1 < total) issue(gs + R - 1);
      PD_B(2);
#pragma unroll
      for (int hs = 0; hs < SPS; ++hs) {
      const int t = u * SPS + hs;
      // Fragments by INLINE-ASM ds_read_b128, one slot ahead of their use, with counted lgkmcnt waits that carry the registers: for a C++
      // LDS load hipcc cannot tell the stage being read from the stages the LDS-DMA is still filling and puts s_waitcnt vmcnt(0) in
      // front of it — every step then waited for the pieces issued a moment ago (1 600 cycles per step instead of ~500).
      const uint32_t sa = ring_lds + (uint32_t)((u % R) * STAGE + hs * STEP_B) + (uint32_t)lane * 16u;
      // two fragment buffers used alternately (compile-time indices): copying "next" into "current" cost 8 v_mov per slot — 2 900 VALU
      // instructions per wave, as much issue time as the MFMAs themselves
      f16x8 fh[2], fl[2];
      lds_read128<0>(fh[0], sa);
      lds_read128<1024>(fl[0], sa);
#pragma unroll
      for (int sl = 0; sl < NSL; ++sl) {
        const int cur = sl & 1, nxt = cur ^ 1;
        if (sl + 1 < NSL) {
          lds_read128<0>(fh[nxt], sa + (uint32_t)((2 * sl + 2) * 1024));
          lds_read128<1024>(fl[nxt], sa + (uint32_t)((2 * sl + 2) * 1024));
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fh[cur]), "+v"(fl[cur]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fh[cur]), "+v"(fl[cur]));
        }
        acc[sl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cur], xh[t], acc[sl], 0, 0, 0);
        acc[sl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fh[cur], xl[t], acc[sl], 0, 0, 0);
        acc[sl] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fl[cur], xh[t], acc[sl], 0, 0, 0);
      }
      }
      PD_B(3);
      // (4-wave workgroups and waves 0-3: AFTER the stage's matrix work, under the SIMD partner's)
      if (!early && gs + R - 1 < total) issue(gs + R - 1);
      PD_B(4);
    }
    // acc[sl][i] 2^-(kx + kw) = z[unit 16 (pass TP + sl) + 4 g + i][row r of the strip]: tanh(z + bias) q, summed over this lane's units
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float2 bqv = bq_s[(pass * TP + sl) * 16 + 4 * g + i];
        // tanh(z) = 1 - 2 / (exp(2z) + 1) on the hardware exp / rcp (|z| clamped: tanh(+-15) is +-1 in f32); absolute error ~1e-7
        const float z = fminf(fmaxf((acc[sl][i] * inv_x) * inv_w + bqv.x, -15.f), 15.f);
        const float th = 1.0f - 2.0f * __frcp_rn(__expf(2.0f * z) + 1.0f);
        p = fmaf(th, bqv.y, p);
      }
    }
  };
  if (is_prod) {
    // the producer's whole pass phase: per ring step wait for its pieces, meet the consumers at their barrier, refill the freed stage
#pragma unroll 1
    for (int gs = 0; gs < total; ++gs) {
      if (gs + R - 1 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * PCS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (gs + R - 1 < total) issue(gs + R - 1);
    }
  } else
#pragma unroll 1
  for (int pass = 0; pass < n_pass; ++pass) {
    const int nsl = n_tiles - pass * TP;                // unit tiles left
    if (nsl > 4) run_pass(std::integral_constant<int, TP>{}, pass);
    else if (nsl > 3) run_pass(std::integral_constant<int, 4>{}, pass);
    else if (nsl > 2) run_pass(std::integral_constant<int, 3>{}, pass);
    else run_pass(std::integral_constant<int, 2>{}, pass);
    }

  // ---- logits of the rows: the four lane groups hold the four unit quarters of a tile
  p += __shfl_xor(p, 16, 64);
  p += __shfl_xor(p, 32, 64);
  if (!valid) p = -INFINITY;
  if (g == 0) lg[wave * 16 + r] = p;
  __syncthreads();                                      // logits visible; every wave is done with the ring

  // ---- softmax over the rows of this wave's batch element (16 SP rows, possibly spread over several waves)
  const int bl = wave / SP;                             // batch element of the workgroup this wave belongs to
  const int rows = 16 * SP;                             // <= 128
  const float l0 = lane < rows ? lg[bl * rows + lane] : -INFINITY;
  const float l1 = lane + 64 < rows ? lg[bl * rows + 64 + lane] : -INFINITY;
  const float mx = wave_max64(fmaxf(l0, l1));
  float wgt = 0.f;                                      // stays 0 for a batch element past B (all logits -inf): nothing is written for it
  if (mx != -INFINITY) {
    const float den = wave_sum64(expf(l0 - mx) + expf(l1 - mx));
    wgt = (expf(p - mx) / den) * inv_x;                 // the row's scale folded into its weight
  }

  // ---- weighted sum from the resident registers.  The rows of the strip sit on the lane index (lane & 15) of the fragments, so
  // summing over rows is a cross-lane reduction; the matrix pipe does the data movement: with the fragment as the A operand and a 0 / 1
  // selection matrix as B, D = x.hi Sel + x.lo Sel is 2^kx x[row][16 columns] EXACTLY (products with 1.0, f32 accumulation) in the
  // accumulator layout, where a lane holds FOUR ROWS (4 g + i) of ONE column (lane & 15): four FMAs with the rows' weights and the
  // lane group's partial sum of 16 columns is done (a DPP reduction of the operand layout took 8 VALU per value: 1 500 per wave).
  // The four lane groups' and the strips' partial sums meet in LDS in a fixed order: part[4 wave + g][D] lives in the ring.
  __shared__ float wrow_s[NW * 16];
  if (g == 0) wrow_s[wave * 16 + r] = wgt;
  f16x8 sel[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 8; ++j) sel[h][j] = (f16_t)(((j >> 2) == h && 4 * g + (j & 3) == r) ? 1.0f : 0.0f);
  // (same-wave LDS write -> read: the compiler's lgkmcnt wait orders them; all 64 lanes of the wave take part)
  __builtin_amdgcn_wave_barrier();
  const f32x4 wr = *reinterpret_cast<const f32x4*>(wrow_s + wave * 16 + 4 * g);
  float* part = reinterpret_cast<float*>(ring) + (size_t)(4 * wave + g) * D + r;
#pragma unroll
  for (int t = 0; t < KS; ++t) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
      d = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[t], sel[h], d, 0, 0, 0);
      d = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[t], sel[h], d, 0, 0, 0);
      part[32 * t + 16 * h] = fmaf(wr[3], d[3], fmaf(wr[2], d[2], fmaf(wr[1], d[1], wr[0] * d[0])));
    }
  }
  __syncthreads();
  // ---- the strips of a batch element and their four lane groups, added in a fixed order; 16-byte stores
  const float* parts = reinterpret_cast<const float*>(ring);
  for (int idx = tid; idx < BPW * (D / 4); idx += NT) {
    const int b_l = idx / (D / 4), c4 = (idx % (D / 4)) * 4;
    if (b0 + b_l >= B) continue;
    f32x4 a = *reinterpret_cast<const f32x4*>(parts + (size_t)(b_l * SP * 4) * D + c4);
    for (int k = 1; k < 4 * SP; ++k) a += *reinterpret_cast<const f32x4*>(parts + (size_t)(b_l * SP * 4 + k) * D + c4);
    *reinterpret_cast<f32x4*>(out + (b0 + b_l) * D + c4) = a;
  }
#ifdef MANNER_POOL_DIAG
  if (diag && lane == 0 && blockIdx.x % 41 == 0 && blockIdx.x / 41 < 64)
    for (int i = 0; i < 6; ++i) diag[((blockIdx.x / 41) * NW + wave) * 6 + i] = pd_t[i];
#endif
}

}  // namespace

// strips (of 16 rows) per batch element for S rows: 1, 2, 4 or 8; 0 = S does not fit one workgroup
static int pool_fused_sp(int64_t S) {
  for (int sp = 1; sp <= 8; sp *= 2)
    if (16 * (int64_t)sp >= S) return sp;
  return 0;
}

bool pool_fused_supported(int64_t B, int64_t S, int D, int Q) {
  return D == 768 && S >= 1 && pool_fused_sp(S) && Q >= 1 && Q <= PF_MAX_PASS * PF_TP * 16 && B >= 1;
}

size_t pool_fused_workspace_bytes(int D, int Q) {
  const int ks = D / 32, n_pass = (Q + PF_TP * 16 - 1) / (PF_TP * 16);
  return (size_t)n_pass * ks * PF_TP * 2 * 1024 + (size_t)n_pass * PF_TP * 16 * sizeof(float2) + 512;
}

int pool_fused(const float* x, const float* W, const float* bias, const float* query, int64_t B, int64_t S, int D, int Q, float* out,
               void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!pool_fused_supported(B, S, D, Q)) return fail(MANNER_HIP_E_INVALID, "pool_fused: B=%lld S=%lld D=%d Q=%d unsupported", (long long)B, (long long)S, D, Q);
  if ((uintptr_t)x % 16 || (uintptr_t)out % 16 || (uintptr_t)workspace % 256) return fail(MANNER_HIP_E_INVALID, "pool_fused: unaligned pointer");
  if (workspace_bytes < pool_fused_workspace_bytes(D, Q)) return fail(MANNER_HIP_E_WORKSPACE, "pool_fused: workspace %zu < %zu bytes", workspace_bytes, pool_fused_workspace_bytes(D, Q));
  const int ks = D / 32, n_tiles = (Q + 15) / 16, n_pass = (n_tiles + PF_TP - 1) / PF_TP, sp = pool_fused_sp(S);
  char* ws = static_cast<char*>(workspace);
  f16x8* Wp = reinterpret_cast<f16x8*>(ws);
  float2* bq = reinterpret_cast<float2*>(ws + (size_t)n_pass * ks * PF_TP * 2 * 1024);
  int32_t* kw = reinterpret_cast<int32_t*>(ws + (size_t)n_pass * ks * PF_TP * 2 * 1024 + (size_t)n_pass * PF_TP * 16 * sizeof(float2));
  uint32_t* wmax = reinterpret_cast<uint32_t*>(kw + 1);
  MANNER_HIP_TRY(hipMemsetAsync(wmax, 0, sizeof(uint32_t), stream));
  hipLaunchKernelGGL(pool_w_max_kernel, dim3(64), dim3(256), 0, stream, W, (int64_t)Q * D, wmax);
  MANNER_LAUNCH_CHECK();
  hipLaunchKernelGGL(pool_pack_w_kernel, dim3(128), dim3(256), 0, stream, W, bias, query, Q, D, ks, n_pass, wmax, kw, Wp, bq);
  MANNER_LAUNCH_CHECK();
  // 8-wave workgroups (one per CU) with ring stages of FOUR k-steps (one barrier and one counted wait per 4 x 15 matrix instructions
  // per wave; 3 stages = 120 KiB): 0.418 ms at B = 4096, S = 50 against 0.454 for the 4-wave form (two workgroups per CU, stages of
  // two k-steps) and 0.436 for 8 waves with stages of two.  MANNER_HIP_POOL_NW=4 selects the 4-wave form (S <= 64 only).
  unsigned long long* diag = nullptr;
#ifdef MANNER_POOL_DIAG
  if (const char* de = getenv("MANNER_HIP_POOL_DIAG")) diag = reinterpret_cast<unsigned long long*>(strtoull(de, nullptr, 0));
#endif
  const char* nw_env = getenv("MANNER_HIP_POOL_NW");
#ifdef MANNER_POOL_LAB    // lab build only (round 5's producer / consumer TIMING experiment: the last wave's strip is not computed, so the
                          // shipped library must not be able to reach it through an environment variable)
  const char* prod_env = getenv("MANNER_HIP_POOL_PRODUCER");
  if (prod_env && atoi(prod_env) == 1) {
    const int bpw = 8 / sp;
    hipLaunchKernelGGL((pool_fused_kernel<24, 8, 3, 4, true>), dim3((unsigned)((B + bpw - 1) / bpw)), dim3(512), 0, stream, x, Wp, bq, kw, n_pass, n_tiles, B,
                       (int)S, sp, out, diag);
    MANNER_LAUNCH_CHECK();
    return MANNER_HIP_OK;
  }
#endif
  if (sp <= 4 && nw_env && atoi(nw_env) == 4) {
    const int bpw = 4 / sp;
    hipLaunchKernelGGL((pool_fused_kernel<24, 4, 3, 2>), dim3((unsigned)((B + bpw - 1) / bpw)), dim3(256), 0, stream, x, Wp, bq, kw, n_pass, n_tiles, B,
                       (int)S, sp, out, diag);
  } else {
    const int bpw = 8 / sp;
    hipLaunchKernelGGL((pool_fused_kernel<24, 8, 3, 4>), dim3((unsigned)((B + bpw - 1) / bpw)), dim3(512), 0, stream, x, Wp, bq, kw, n_pass, n_tiles, B,
                       (int)S, sp, out, diag);
  }
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

}  // namespace manner
