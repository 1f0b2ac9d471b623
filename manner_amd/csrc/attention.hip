// K3: bidirectional multi-head self-attention over packed variable-length news
// (HF BertSelfAttention / eager_attention_forward, transformers/models/bert/modeling_bert.py:111-136,
// 188-203; additive padding mask of :704-708).  Tokens are packed, so the padding mask becomes
// "keys of the same news only"; head_dim is 64 and a news has at most 128 tokens (<= 4 key tiles).
//
// bf16 path — one 64-lane wave per (news, head), no workgroup barriers:
//   memory    every global access is a 16-byte piece of a whole 128-byte row segment (8 lanes per row): K tiles and
//             Q blocks go registers -> wave-private 4 KiB LDS slab (chunk-swizzled) -> MFMA fragments, V goes by
//             LDS-DMA straight into its row-major image, O^T leaves through the same slab.  Stores straight from the
//             accumulator layout (16 bytes per row per instruction) cost 20 us of a 97 us launch; with the slab the
//             kernel moves its 403 MB at 5.5 TB/s.
//   S^T = K Q^T   v_mfma_f32_32x32x16_bf16 with K rows as the A operand: a lane then owns ONE query
//                 (column) and 16 keys per 32-key tile, so max / sum are in-lane reductions plus
//                 one exchange with lane^32; softmax runs ONLINE over the key tiles so that only one
//                 S^T tile is live; the next query block's rows are requested while the current one computes;
//   O^T = V^T P^T the S^T accumulators, converted pairwise to bf16, ARE the B operand of the second
//                 product (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's
//                 operand"), k-order permuted: element j of lane half h is key 16s+8(j>>2)+4h+(j&3);
//                 V sits row-major in a wave-private LDS image and is read key-permuted and
//                 transposed with ds_read_b64_tr_b16.  O^T keeps the query on the lane, so the
//                 1/sum scaling is lane-local.
// f32 path — exact-f32 kernel for the parity modes on the f32 matrix pipe (attn_wave_f32); the older VALU kernel, one
// workgroup per (news, head), stays selectable for A/B.
#include <math.h>
#include <stdlib.h>

#include "common.h"

namespace manner {
namespace {

template <typename TE, int NKT>
__device__ __forceinline__ void attn_wave_bf16(const TE* __restrict__ qkv, TE* __restrict__ ctx,
                                               int tok0, int L, int H, int head, char* vl) {
  typedef typename E16<TE>::v8 e16x8;           // TE: bf16_t or f16_t — same layout, the MFMA of the type
  typedef typename E16<TE>::v4 e16x4;
  const int lane = threadIdx.x & 63, rr = lane & 31, h = lane >> 5;
  char* ol = vl + NKT * 32 * 128;                 // output slab behind the V image
  const size_t ld = 3 * (size_t)H;
  const TE* Qb = qkv + (size_t)tok0 * ld + head * 64;
  const TE* Kb = Qb + H;
  const TE* Vb = Qb + 2 * H;

  // Every global access is a 16-byte piece of a whole 128-byte row segment (lane = (row lane>>3, chunk lane&7));
  // the MFMA-layout fragments are then read back from LDS.  Loads straight in the fragment layout touch each row in
  // four instructions, 32 bytes at a time, and were what bounded the kernel.
  const int r8 = lane >> 3, c8 = lane & 7;
  // all loads of the pair in flight first: K tiles and the first Q block to registers, V by LDS-DMA straight into its
  // row-major image (lane-linear destination = 8 rows x 128 B per instruction; no registers)
  f32x4 kt_[NKT][4], qt_[4];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = min(32 * kt + 8 * i + r8, L - 1);      // rows >= L replicate row L-1 (finite; masked / weighted by 0)
      kt_[kt][i] = *reinterpret_cast<const f32x4*>(Kb + (size_t)row * ld + 8 * (c8 ^ (((8 * i + r8) >> 1) & 7)));
    }
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = min(32 * kt + 8 * i + r8, L - 1);
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(Vb + (size_t)row * ld + 8 * c8), LDS_PTR(vl + (32 * kt + 8 * i) * 128), 16, 0, 0);
    }
  auto load_q = [&](int qb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = min(32 * qb + 8 * i + r8, L - 1);
      qt_[i] = *reinterpret_cast<const f32x4*>(Qb + (size_t)row * ld + 8 * (c8 ^ (((8 * i + r8) >> 1) & 7)));
    }
  };
  load_q(0);
  // K: one 32-key tile at a time through the 4 KiB slab, chunk-swizzled -> fragments: lane (rr, h) holds
  // K[key = 32kt + rr][d = 16ks + 8h .. +7], i.e. chunk 2ks + h of row rr
  e16x8 kf[NKT][4];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(ol + (8 * i + r8) * 128 + (c8 << 4)) = kt_[kt][i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      kf[kt][ks] = *reinterpret_cast<const e16x8*>(ol + rr * 128 + (((2 * ks + h) ^ ((rr >> 1) & 7)) << 4));
    __builtin_amdgcn_wave_barrier();
  }

  // per-lane address of the transposed V reads: 16-lane group g, lane i = 4q + p of the group
  // supplies row q, columns 4p..4p+3 of the group's 4x16 block
  const int gi = lane & 15;
  const int tr_base = ((gi >> 2) * 64 + 16 * ((lane >> 4) & 1) + 4 * (gi & 3)) * 2 + (4 * h) * 128;

#pragma unroll
  for (int qb = 0; qb < NKT; ++qb) {
    if (32 * qb >= L) break;
    // Q block: rows -> output slab (free between blocks), chunk-swizzled -> fragments; the next block's rows are
    // requested before this block's arithmetic starts
    e16x8 qfb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(ol + (8 * i + r8) * 128 + (c8 << 4)) = qt_[i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qfb[ks] = *reinterpret_cast<const e16x8*>(ol + rr * 128 + (((2 * ks + h) ^ ((rr >> 1) & 7)) << 4));
    __builtin_amdgcn_wave_barrier();
    if (qb == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the V image (LDS-DMA is not tracked by the compiler)
    if (qb + 1 < NKT && 32 * (qb + 1) < L) load_q(qb + 1);
    // online softmax over the 32-key tiles (one S^T tile = 16 registers live at a time): running max m of
    // the RAW scores, running sum l, O^T rescaled by exp(m_old - m_new) only when some row's max moved.
    // The kernel is issue-bound on this VALU work, so it is kept minimal: scale (1/8) and log2(e) are
    // folded into one FMA in front of v_exp_f32, and only the last key tile is masked.
    constexpr float C = 0.125f * 1.44269504088896340736f;   // head_dim^-0.5 * log2(e)
    float m = -INFINITY, l = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x16 st;
#pragma unroll
      for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) st = E16<TE>::mfma32(kf[kt][ks], qfb[ks], st);
      if (kt == NKT - 1) {                             // L > 32 (NKT - 1): earlier tiles hold real keys only
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = 32 * kt + (e & 3) + 8 * (e >> 2) + 4 * h;
          st[e] = key < L ? st[e] : -INFINITY;
        }
      }
      float tmx = fmaxf(fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])), fmaxf(fmaxf(st[4], st[5]), fmaxf(st[6], st[7])));
      tmx = fmaxf(tmx, fmaxf(fmaxf(fmaxf(st[8], st[9]), fmaxf(st[10], st[11])), fmaxf(fmaxf(st[12], st[13]), fmaxf(st[14], st[15]))));
      tmx = fmaxf(tmx, __shfl_xor(tmx, 32, 64));
      const float mn = fmaxf(m, tmx);                  // finite: tile 0 always holds key 0 < L
      const float nmc = -mn * C;
      float rs = 0.f;
      e16x8 pf[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(fmaf(st[e], C, nmc));   // exp((s - max) / 8); 0 for masked keys
        pf[e >> 3][e & 7] = (TE)p;
        rs += p;
      }
      rs += __shfl_xor(rs, 32, 64);
      if (kt == 0) {
        l = rs;
      } else {
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * C);
        l = l * alpha + rs;
        if (__any(mn > m)) {                           // wave-uniform: some query's running max moved
#pragma unroll
          for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
        }
      }
      m = mn;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const char* a0 = vl + tr_base + (32 * kt + 16 * s2) * 128 + (32 * dt) * 2;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 8 * 128));
          typedef short s16x8 __attribute__((ext_vector_type(8)));
          const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          const e16x8 vf = __builtin_bit_cast(e16x8, both);
          o[dt] = E16<TE>::mfma32(vf, pf[s2], o[dt]);
        }
      }
    }
    // O^T -> ctx rows through a wave-private 4 KiB slab (32 queries x 128 B, 16-byte chunks XOR-swizzled with (row>>1)&7: two 128-byte rows share a 64-bank span): the lane
    // that owns query rr writes its 8-byte runs, then every store instruction covers 8 whole 128-byte row segments
    // (straight from the accumulator layout each row would be touched by 8 instructions, 16 bytes at a time)
    {
      const float inv = 1.0f / l;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = 4 * dt + g;
          *reinterpret_cast<e16x4*>(ol + rr * 128 + ((c ^ ((rr >> 1) & 7)) << 4) + 8 * h) =
              e16x4{(TE)(o[dt][4 * g] * inv), (TE)(o[dt][4 * g + 1] * inv),
                     (TE)(o[dt][4 * g + 2] * inv), (TE)(o[dt][4 * g + 3] * inv)};
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + r8;
        const int q = 32 * qb + row;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ol + row * 128 + (c8 << 4));
        if (q < L) *reinterpret_cast<f32x4*>(ctx + (size_t)(tok0 + q) * H + head * 64 + 8 * (c8 ^ ((row >> 1) & 7))) = v;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <typename TE>
__global__ __launch_bounds__(256, 2) void attn_bf16_kernel(const TE* __restrict__ qkv, TE* __restrict__ ctx,
                                                           const int32_t* __restrict__ cu, int64_t n_pairs,
                                                           int heads, int H, int lds_per_wave) {
  extern __shared__ __attribute__((aligned(16))) char vlds[];
  const int wave = threadIdx.x >> 6;
  const int64_t pair = (int64_t)blockIdx.x * 4 + wave;
  if (pair >= n_pairs) return;
  const int n = (int)(pair / heads), head = (int)(pair - (int64_t)n * heads);
  const int tok0 = __builtin_amdgcn_readfirstlane(cu[n]);
  const int L = __builtin_amdgcn_readfirstlane(cu[n + 1]) - tok0;
  char* vl = vlds + wave * lds_per_wave;
  if (L <= 32) attn_wave_bf16<TE, 1>(qkv, ctx, tok0, L, H, head, vl);
  else if (L <= 64) attn_wave_bf16<TE, 2>(qkv, ctx, tok0, L, H, head, vl);
  else if (L <= 96) attn_wave_bf16<TE, 3>(qkv, ctx, tok0, L, H, head, vl);
  else attn_wave_bf16<TE, 4>(qkv, ctx, tok0, L, H, head, vl);
}

// ---- exact-f32 attention on the f32 matrix pipe: the structure of attn_wave_bf16 with v_mfma_f32_32x32x2_f32 (exact f32
// products, f32 accumulation): S^T = K Q^T with lane (key rr, half h) holding K[key][32h .. 32h+31] — the MFMA's two
// k-slots pair feature s with feature 32+s for Q and K alike, which is just another summation order — softmax online
// over 32-key tiles with expf, and the S^T accumulators as the B operand of O^T = V^T P^T, V^T read row-wise from a
// row-major f32 LDS image (two key rows per ds_read_b32).  All global traffic in whole 256-byte row segments.
// a3 != NULL (x3 modes): the output rows are written as the out-projection GEMM's split operand [hi | hi | lo] (row stride 3H of
// the 16-bit type a3_dt) instead of as f32 ctx rows — the f32 tensor and the separate split pass over it never exist.
template <typename TE>
__device__ __forceinline__ void store_split4(TE* __restrict__ dst, int H, const f32x4& v) {
  typename E16<TE>::v4 hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) { hi[e] = (TE)v[e]; lo[e] = (TE)(v[e] - (float)hi[e]); }
  *reinterpret_cast<typename E16<TE>::v4*>(dst) = hi;
  *reinterpret_cast<typename E16<TE>::v4*>(dst + H) = hi;
  *reinterpret_cast<typename E16<TE>::v4*>(dst + 2 * H) = lo;
}
template <int NKT>
__device__ __forceinline__ void attn_wave_f32(const float* __restrict__ qkv, float* __restrict__ ctx, int tok0, int L, int H,
                                              int head, char* vl, void* __restrict__ a3 = nullptr, int a3_dt = 0) {
  const int lane = threadIdx.x & 63, rr = lane & 31, h = lane >> 5;
  char* ol = vl + NKT * 32 * 256;                 // 8 KiB slab (32 rows x 256 B) behind the V image
  const size_t ld = 3 * (size_t)H;
  const float* Qb = qkv + (size_t)tok0 * ld + head * 64;
  const float* Kb = Qb + H;
  const float* Vb = Qb + 2 * H;
  const int r4 = lane >> 4, c16 = lane & 15;      // coalesced piece: row r4 of a 4-row group, 16-byte chunk c16 of 16
  // V by LDS-DMA straight into its row-major image: 4 rows x 256 B per instruction, lane-linear
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = min(32 * kt + 4 * i + r4, L - 1);
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(Vb + (size_t)row * ld + 4 * c16), LDS_PTR(vl + (32 * kt + 4 * i) * 256), 16, 0, 0);
    }
  // a 32-row tile of K or Q: coalesced rows -> slab (16-byte chunks XOR-swizzled with row & 15) -> lane (rr, h) takes
  // the 32 features 32h .. 32h+31 of row rr
  auto tile_to_frag = [&](const float* base, int row0, float (&frag)[32]) {
    f32x4 t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = min(row0 + 4 * i + r4, L - 1);
      t[i] = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + 4 * (c16 ^ ((4 * i + r4) & 15)));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(ol + (4 * i + r4) * 256 + (c16 << 4)) = t[i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(ol + rr * 256 + (((8 * h + j) ^ (rr & 15)) << 4));
      frag[4 * j] = v[0]; frag[4 * j + 1] = v[1]; frag[4 * j + 2] = v[2]; frag[4 * j + 3] = v[3];
    }
    __builtin_amdgcn_wave_barrier();
  };
  float kf[NKT][32];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) tile_to_frag(Kb, 32 * kt, kf[kt]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the V image has landed (LDS-DMA is not tracked by the compiler)
#pragma unroll
  for (int qb = 0; qb < NKT; ++qb) {
    if (32 * qb >= L) break;
    float qf[32];
    tile_to_frag(Qb, 32 * qb, qf);
    float m = -INFINITY, l = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x16 st;
#pragma unroll
      for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[kt][s2], qf[s2], st, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = 32 * kt + (e & 3) + 8 * (e >> 2) + 4 * h;
        st[e] = key < L ? st[e] * 0.125f : -INFINITY;            // scores / sqrt(64), as the reference scales them
      }
      float tmx = st[0];
#pragma unroll
      for (int e = 1; e < 16; ++e) tmx = fmaxf(tmx, st[e]);
      tmx = fmaxf(tmx, __shfl_xor(tmx, 32, 64));
      const float mn = fmaxf(m, tmx);
      float rs = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { st[e] = expf(st[e] - mn); rs += st[e]; }
      rs += __shfl_xor(rs, 32, 64);
      const float alpha = kt == 0 ? 0.f : expf(m - mn);
      l = l * alpha + rs;
      if (kt > 0) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
      }
      m = mn;
      // O^T += V^T P^T: MFMA step t pairs key 8(t>>2) + (t&3) (k-slot 0) with key + 4 (k-slot 1) — the keys the two
      // lane halves hold in accumulator element t
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const char* vrow = vl + (32 * kt + 8 * (t >> 2) + (t & 3) + 4 * h) * 256 + rr * 4;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const float vv = *reinterpret_cast<const float*>(vrow + 128 * dt);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv, st[t], o[dt], 0, 0, 0);
        }
      }
    }
    // O^T -> ctx rows through the slab: lane (query rr, half h) owns features 32dt + 8g + 4h .. +3
    {
      const float inv = 1.0f / l;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = (32 * dt + 8 * g + 4 * h) >> 2;             // 16-byte chunk 0..15 of the 256-byte row
          *reinterpret_cast<f32x4*>(ol + rr * 256 + ((c ^ (rr & 15)) << 4)) =
              f32x4{o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + r4;
        const int q = 32 * qb + row;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ol + row * 256 + (c16 << 4));
        if (q < L) {
          const size_t col = head * 64 + 4 * (c16 ^ (row & 15));
          if (!a3) *reinterpret_cast<f32x4*>(ctx + (size_t)(tok0 + q) * H + col) = v;
          else if (a3_dt == DT_F16) store_split4<f16_t>(static_cast<f16_t*>(a3) + (size_t)(tok0 + q) * 3 * H + col, H, v);
          else store_split4<bf16_t>(static_cast<bf16_t*>(a3) + (size_t)(tok0 + q) * 3 * H + col, H, v);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---- f16x3 attention (round 5): the structure of attn_wave_f32 — f32 Q | K | V rows in, f32 softmax, the out-projection's split
// operand out — with BOTH products as split (x3) f16 products on the 16-bit matrix pipe instead of v_mfma_f32_32x32x2_f32 (4x slower per
// FLOP and 32 instructions per 32 x 32 score tile): every f32 value v is used as hi = f16(v), lo = f16(v - hi) and a product a.b as
// a.hi b.hi + a.hi b.lo + a.lo b.hi (f32 accumulation; the dropped lo.lo term is 2^-22 relative — the arithmetic of the mode's GEMMs).
//   S^T = K Q^T : lane (row rr, half h) holds the 32 features 32h .. 32h+31 of its K / Q row (tile_to_frag of the f32 kernel); MFMA step
//                 s takes elements 8s .. 8s+7 of both halves — the k-set {8s..8s+7} u {32+8s..32+8s+7}, the same for K and Q, which is
//                 just another summation order; 12 v_mfma_f32_32x32x16_f16 per score tile.
//   O^T = V^T P^T: V is split when it is loaded (f32 rows -> registers -> two row-major 16-bit LDS images, hi and lo: the same bytes as
//                 the f32 kernel's one f32 image) and read key-permuted and transposed with ds_read_b64_tr_b16 exactly as in
//                 attn_wave_bf16; the softmax numerators, split in the accumulator layout, are the B operand; 12 MFMAs per key tile.
// |q|, |k|, |v| of an encoder are O(1..50): far inside f16's range, low parts of values below 1e-4 keep 3e-8 absolute (subnormals).
template <int NKT>
__device__ __forceinline__ void attn_wave_x3(const float* __restrict__ qkv, int tok0, int L, int H, int head, char* vl,
                                             f16_t* __restrict__ a3) {
  const int lane = threadIdx.x & 63, rr = lane & 31, h = lane >> 5;
  char* vhi = vl;                                  // [32 NKT keys][64] f16, row-major 128-byte rows
  char* vlo = vl + NKT * 32 * 128;
  char* ol = vl + NKT * 32 * 256;                  // 8 KiB slab (32 rows x 256 B) behind the two V images
  const size_t ld = 3 * (size_t)H;
  const float* Qb = qkv + (size_t)tok0 * ld + head * 64;
  const float* Kb = Qb + H;
  const float* Vb = Qb + 2 * H;
  const int r4 = lane >> 4, c16 = lane & 15;      // coalesced piece: row r4 of a 4-row group, 16-byte chunk c16 of 16
  // the K tiles and the first Q block are requested before anything waits for a load (the kernel is latency-bound: 4 waves per CU);
  // 8 x 16 bytes per lane and tile.  (All V tiles in flight as well: 150 spilled registers in the four-tile body.)
  f32x4 kt_[NKT <= 3 ? NKT : 1][8], qt_[8];
  auto tile_load = [&](const float* base, int row0, bool swz, f32x4 (&t)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = min(row0 + 4 * i + r4, L - 1);
      t[i] = *reinterpret_cast<const f32x4*>(base + (size_t)row * ld + 4 * (swz ? (c16 ^ ((4 * i + r4) & 15)) : c16));
    }
  };
  constexpr bool PRELOAD = NKT <= 3;               // four K tiles in flight next to everything else do not fit the register file
#pragma unroll
  for (int kt = 0; kt < (PRELOAD ? NKT : 1); ++kt) tile_load(Kb, 32 * kt, true, kt_[kt]);
  tile_load(Qb, 0, true, qt_);
  // a loaded 32-row tile of K or Q: rows -> slab (16-byte chunks XOR-swizzled with row & 15) -> lane (rr, h) takes the 32 features
  // 32h .. 32h+31 of row rr and splits them: fh[s] / fl[s] = elements 8s .. 8s+7
  auto tile_to_split = [&](const f32x4 (&t)[8], f16x8 (&fh)[4], f16x8 (&fl)[4]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(ol + (4 * i + r4) * 256 + (c16 << 4)) = t[i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ol + rr * 256 + (((8 * h + 2 * s) ^ (rr & 15)) << 4));
      const f32x4 b = *reinterpret_cast<const f32x4*>(ol + rr * 256 + (((8 * h + 2 * s + 1) ^ (rr & 15)) << 4));
      f16x8 hi;
#pragma unroll
      for (int e = 0; e < 4; ++e) { hi[e] = (f16_t)a[e]; hi[4 + e] = (f16_t)b[e]; }
      asm volatile("" : "+v"(hi));
      f16x8 lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo[e] = (f16_t)(a[e] - (float)hi[e]); lo[4 + e] = (f16_t)(b[e] - (float)hi[4 + e]); }
      fh[s] = hi;
      fl[s] = lo;
    }
    __builtin_amdgcn_wave_barrier();
  };
  // K tile kt is split while the V rows of tile kt (requested just before) are on their way; V: f32 rows -> hi / lo halves -> the two
  // 16-bit images (8 bytes per lane and image; a 16-lane group writes one whole 128-byte row)
  f16x8 kh[NKT][4], kl[NKT][4];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    f32x4 vt_[8];
    tile_load(Vb, 32 * kt, false, vt_);
    tile_to_split(kt_[PRELOAD ? kt : 0], kh[kt], kl[kt]);
    if (!PRELOAD && kt + 1 < NKT) tile_load(Kb, 32 * (kt + 1), true, kt_[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f16x4 hi4, lo4;
#pragma unroll
      for (int e = 0; e < 4; ++e) hi4[e] = (f16_t)vt_[i][e];
      asm volatile("" : "+v"(hi4));                 // the remainder is taken against THESE bits
#pragma unroll
      for (int e = 0; e < 4; ++e) lo4[e] = (f16_t)(vt_[i][e] - (float)hi4[e]);
      const int off = (32 * kt + 4 * i + r4) * 128 + c16 * 8;
      *reinterpret_cast<f16x4*>(vhi + off) = hi4;
      *reinterpret_cast<f16x4*>(vlo + off) = lo4;
    }
  }
  // per-lane address of the transposed V reads (attn_wave_bf16): 16-lane group g, lane i = 4q + p of the group supplies row q,
  // columns 4p .. 4p+3 of the group's 4 x 16 block
  const int gi = lane & 15;
  const int tr_base = ((gi >> 2) * 64 + 16 * ((lane >> 4) & 1) + 4 * (gi & 3)) * 2 + (4 * h) * 128;
  auto tr_frag = [&](const char* img, int kt, int s2, int dt) -> f16x8 {
    const char* a0 = img + tr_base + (32 * kt + 16 * s2) * 128 + (32 * dt) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 8 * 128));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8, both);
  };
#pragma unroll
  for (int qb = 0; qb < NKT; ++qb) {
    if (32 * qb >= L) break;
    f16x8 qh[4], ql[4];
    tile_to_split(qt_, qh, ql);
    if (qb + 1 < NKT && 32 * (qb + 1) < L) tile_load(Qb, 32 * (qb + 1), true, qt_);     // the next block's rows, under this block's arithmetic
    // scale (1/8) and log2(e) folded into one FMA in front of v_exp_f32 (1 ulp: far inside the split products' 2^-22), as in the 16-bit
    // kernel — expf's software expansion was a third of this kernel's VALU work.  The three split terms accumulate in SEPARATE chains
    // (a wave has its SIMD to itself here: a dependent MFMA chain of 12 exposes 12 full latencies) and meet in one VALU add.
    constexpr float C = 0.125f * 1.44269504088896340736f;
    float m = -INFINITY, l = 0.f;
    f32x16 o[2], oc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) { o[dt][e] = 0.f; oc[dt][e] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      f32x16 st, s1, s2;
#pragma unroll
      for (int e = 0; e < 16; ++e) { st[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[kt][s], qh[s], st, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[kt][s], ql[s], s1, 0, 0, 0);
        s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[kt][s], qh[s], s2, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = 32 * kt + (e & 3) + 8 * (e >> 2) + 4 * h;
        st[e] = key < L ? st[e] + (s1[e] + s2[e]) : -INFINITY;       // RAW scores q.k
      }
      float tmx = st[0];
#pragma unroll
      for (int e = 1; e < 16; ++e) tmx = fmaxf(tmx, st[e]);
      tmx = fmaxf(tmx, __shfl_xor(tmx, 32, 64));
      const float mn = fmaxf(m, tmx);
      const float nmc = -mn * C;
      float rs = 0.f;
      f16x8 ph[2], pl[2];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(st[e], C, nmc));   // exp((s - max) / 8); 0 for masked keys
        rs += pv;
        const f16_t hv = (f16_t)pv;
        ph[e >> 3][e & 7] = hv;
        pl[e >> 3][e & 7] = (f16_t)(pv - (float)hv);
      }
      rs += __shfl_xor(rs, 32, 64);
      const float alpha = kt == 0 ? 0.f : __builtin_amdgcn_exp2f((m - mn) * C);
      l = l * alpha + rs;
      if (kt > 0) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) { o[dt][e] *= alpha; oc[dt][e] *= alpha; }
      }
      m = mn;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int s2i = 0; s2i < 2; ++s2i) {
          const f16x8 vh = tr_frag(vhi, kt, s2i, dt), vlw = tr_frag(vlo, kt, s2i, dt);
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s2i], o[dt], 0, 0, 0);
          oc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s2i], oc[dt], 0, 0, 0);
          oc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vlw, ph[s2i], oc[dt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] += oc[dt][e];
    // O^T -> the split operand rows through the slab: lane (query rr, half h) owns features 32dt + 8g + 4h .. +3
    {
      const float inv = 1.0f / l;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = (32 * dt + 8 * g + 4 * h) >> 2;             // 16-byte chunk 0..15 of the 256-byte row
          *reinterpret_cast<f32x4*>(ol + rr * 256 + ((c ^ (rr & 15)) << 4)) =
              f32x4{o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv};
        }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + r4;
        const int q = 32 * qb + row;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ol + row * 256 + (c16 << 4));
        if (q < L) store_split4<f16_t>(a3 + (size_t)(tok0 + q) * 3 * H + head * 64 + 4 * (c16 ^ (row & 15)), H, v);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// (Single-wave workgroups under a 256-register budget would let a CU keep 5 instead of 4 pairs in flight at 96 tokens — LDS: 8 KiB per
// key tile for the two V images + the 8 KiB slab — but the four-tile body then spills 238 registers: tried, not kept.)
// NKTMAX = key tiles of the call's padded length (the host guarantees L <= 32 NKTMAX): a batch of at most 96 tokens per news — the
// reference's tokenizer_max_length — does not carry the four-tile body, whose K tiles alone are 256 registers
template <int NKTMAX>
__global__ __launch_bounds__(128, 1) void attn_x3_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ cu, int64_t n_pairs,
                                                         int heads, int H, int lds_per_wave, f16_t* __restrict__ a3) {
  extern __shared__ __attribute__((aligned(16))) char vlds[];
  const int wave = threadIdx.x >> 6;
  const int64_t pair = (int64_t)blockIdx.x * 2 + wave;
  if (pair >= n_pairs) return;
  const int n = (int)(pair / heads), head = (int)(pair - (int64_t)n * heads);
  const int tok0 = __builtin_amdgcn_readfirstlane(cu[n]);
  const int L = __builtin_amdgcn_readfirstlane(cu[n + 1]) - tok0;
  if (L <= 0) return;
  char* vl = vlds + wave * lds_per_wave;
  if (NKTMAX == 1 || L <= 32) attn_wave_x3<1>(qkv, tok0, L, H, head, vl, a3);
  else if (NKTMAX == 2 || L <= 64) { if constexpr (NKTMAX >= 2) attn_wave_x3<2>(qkv, tok0, L, H, head, vl, a3); }
  else if (NKTMAX == 3 || L <= 96) { if constexpr (NKTMAX >= 3) attn_wave_x3<3>(qkv, tok0, L, H, head, vl, a3); }
  else { if constexpr (NKTMAX >= 4) attn_wave_x3<4>(qkv, tok0, L, H, head, vl, a3); }
}

__global__ __launch_bounds__(128, 1) void attn_f32_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                               const int32_t* __restrict__ cu, int64_t n_pairs, int heads,
                                                               int H, int lds_per_wave, void* __restrict__ a3, int a3_dt) {
  extern __shared__ __attribute__((aligned(16))) char vlds[];
  const int wave = threadIdx.x >> 6;
  const int64_t pair = (int64_t)blockIdx.x * 2 + wave;
  if (pair >= n_pairs) return;
  const int n = (int)(pair / heads), head = (int)(pair - (int64_t)n * heads);
  const int tok0 = __builtin_amdgcn_readfirstlane(cu[n]);
  const int L = __builtin_amdgcn_readfirstlane(cu[n + 1]) - tok0;
  char* vl = vlds + wave * lds_per_wave;
  if (L <= 32) attn_wave_f32<1>(qkv, ctx, tok0, L, H, head, vl, a3, a3_dt);
  else if (L <= 64) attn_wave_f32<2>(qkv, ctx, tok0, L, H, head, vl, a3, a3_dt);
  else if (L <= 96) attn_wave_f32<3>(qkv, ctx, tok0, L, H, head, vl, a3, a3_dt);
  else attn_wave_f32<4>(qkv, ctx, tok0, L, H, head, vl, a3, a3_dt);
}

// exact-f32 attention, VALU form (MANNER_HIP_ATTN_F32_VALU=1): one workgroup (128 threads, one per query) per (news, head)
__global__ __launch_bounds__(128) void attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ ctx,
                                                       const int32_t* __restrict__ cu, int heads, int H) {
  __shared__ __attribute__((aligned(16))) float Ks[MANNER_HIP_MAX_LEN * 64];
  __shared__ __attribute__((aligned(16))) float Vs[MANNER_HIP_MAX_LEN * 64];
  const int n = blockIdx.x / heads, head = blockIdx.x - n * heads;
  const int tok0 = cu[n], L = cu[n + 1] - tok0;
  const size_t ld = 3 * (size_t)H;
  const float* Qb = qkv + (size_t)tok0 * ld + head * 64;
  for (int i = threadIdx.x; i < L * 16; i += 128) {
    const int row = i >> 4, c = (i & 15) * 4;
    *reinterpret_cast<f32x4*>(Ks + row * 64 + c) = *reinterpret_cast<const f32x4*>(Qb + (size_t)row * ld + H + c);
    *reinterpret_cast<f32x4*>(Vs + row * 64 + c) = *reinterpret_cast<const f32x4*>(Qb + (size_t)row * ld + 2 * H + c);
  }
  __syncthreads();
  const int q = threadIdx.x;
  if (q >= L) return;
  float qv[64];
#pragma unroll
  for (int c = 0; c < 64; c += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(Qb + (size_t)q * ld + c);
    qv[c] = t[0]; qv[c + 1] = t[1]; qv[c + 2] = t[2]; qv[c + 3] = t[3];
  }
  auto score = [&](int key) {
    const float* kr = Ks + key * 64;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < 64; ++c) a = fmaf(qv[c], kr[c], a);
    return a * 0.125f;
  };
  float mx = -INFINITY;
  for (int key = 0; key < L; ++key) mx = fmaxf(mx, score(key));
  float sum = 0.f, o[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) o[c] = 0.f;
  for (int key = 0; key < L; ++key) {
    const float p = expf(score(key) - mx);
    sum += p;
    const float* vr = Vs + key * 64;
#pragma unroll
    for (int c = 0; c < 64; ++c) o[c] = fmaf(p, vr[c], o[c]);
  }
  const float inv = 1.0f / sum;
  float* dst = ctx + (size_t)(tok0 + q) * H + head * 64;
#pragma unroll
  for (int c = 0; c < 64; c += 4)
    *reinterpret_cast<f32x4*>(dst + c) = f32x4{o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv};
}

// ---- last layer: one query ([CLS]) per news.  One wave per (news, head); lane (g = lane>>3,
// c = lane&7) walks key rows g, g+8, ... and owns the 8 head dims 8c..8c+7 of each, so every load
// instruction fetches 8 whole 128-byte K (or V) rows.  f32 arithmetic in both precisions.
template <typename T> struct Row8;
template <> struct Row8<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float v[8]) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float v[8]) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x8*>(p) = t;
  }
};
template <> struct Row8<f16_t> {
  static __device__ __forceinline__ void load(const f16_t* p, float v[8]) {
    const f16x8 t = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
  }
  static __device__ __forceinline__ void store(f16_t* p, const float v[8]) {
    f16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = (f16_t)v[e];
    *reinterpret_cast<f16x8*>(p) = t;
  }
};
template <> struct Row8<float> {
  static __device__ __forceinline__ void load(const float* p, float v[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
  }
  static __device__ __forceinline__ void store(float* p, const float v[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
};

template <typename T>
__global__ __launch_bounds__(256) void attn_cls_kernel(const T* __restrict__ qcls, const T* __restrict__ kv,
                                                       T* __restrict__ ctx, const int32_t* __restrict__ cu,
                                                       int64_t n_pairs, int heads, int H) {
  const int64_t pair = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= n_pairs) return;
  const int lane = threadIdx.x & 63, g = lane >> 3, c = lane & 7;
  const int n = (int)(pair / heads), head = (int)(pair - (int64_t)n * heads);
  const int tok0 = cu[n], L = cu[n + 1] - tok0;
  const size_t ld = 2 * (size_t)H;
  const T* Kb = kv + (size_t)tok0 * ld + head * 64 + 8 * c;
  const T* Vb = Kb + H;
  float q[8];
  Row8<T>::load(qcls + (size_t)n * H + head * 64 + 8 * c, q);
  constexpr int MAXR = MANNER_HIP_MAX_LEN / 8;       // 16 key rows per lane group
  float sc[MAXR];
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    const int key = 8 * r + g;
    float s = -INFINITY;
    if (8 * r < L) {                                  // wave-uniform
      float kr[8];
      Row8<T>::load(Kb + (size_t)min(key, L - 1) * ld, kr);
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) d = fmaf(q[e], kr[e], d);
      d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
      s = key < L ? d * 0.125f : -INFINITY;
    }
    sc[r] = s;
    mx = fmaxf(mx, s);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 8, 64)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f, o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < MAXR; ++r) {
    if (8 * r < L) {
      const int key = 8 * r + g;
      const float p = expf(sc[r] - mx);               // 0 for masked keys
      sum += p;
      float vr[8];
      Row8<T>::load(Vb + (size_t)min(key, L - 1) * ld, vr);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaf(p, vr[e], o[e]);
    }
  }
  // every lane of a row group carries the same p: sum over groups only (xor 8, 16, 32)
  sum += __shfl_xor(sum, 8, 64); sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    o[e] += __shfl_xor(o[e], 8, 64); o[e] += __shfl_xor(o[e], 16, 64); o[e] += __shfl_xor(o[e], 32, 64);
  }
  if (g == 0) {
    const float inv = 1.0f / sum;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= inv;
    Row8<T>::store(ctx + (size_t)n * H + head * 64 + 8 * c, o);
  }
}

}  // namespace

int attention_cls(DType dt, const void* qcls, const void* kv, void* ctx_cls, const int32_t* cu, int64_t n_news, int heads,
                  int H, hipStream_t stream) {
  if (H != heads * 64) return fail(MANNER_HIP_E_INVALID, "head_dim must be 64 (H=%d heads=%d)", H, heads);
  const int64_t pairs = n_news * heads;
  dim3 g((unsigned)((pairs + 3) / 4)), b(256);
  if (dt == DT_BF16)
    hipLaunchKernelGGL(attn_cls_kernel<bf16_t>, g, b, 0, stream, static_cast<const bf16_t*>(qcls), static_cast<const bf16_t*>(kv),
                       static_cast<bf16_t*>(ctx_cls), cu, pairs, heads, H);
  else if (dt == DT_F16)
    hipLaunchKernelGGL(attn_cls_kernel<f16_t>, g, b, 0, stream, static_cast<const f16_t*>(qcls), static_cast<const f16_t*>(kv),
                       static_cast<f16_t*>(ctx_cls), cu, pairs, heads, H);
  else
    hipLaunchKernelGGL(attn_cls_kernel<float>, g, b, 0, stream, static_cast<const float*>(qcls), static_cast<const float*>(kv),
                       static_cast<float*>(ctx_cls), cu, pairs, heads, H);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int attention_varlen(DType dt, const void* qkv, void* ctx, const int32_t* cu, int64_t n_news, int heads, int H,
                     int max_len, hipStream_t stream, void* split_out, DType split_dt) {
  if (H != heads * 64) return fail(MANNER_HIP_E_INVALID, "head_dim must be 64 (H=%d heads=%d)", H, heads);
  if (max_len < 1 || max_len > MANNER_HIP_MAX_LEN)
    return fail(MANNER_HIP_E_INVALID, "padded length %d exceeds the %d-token attention tile", max_len, MANNER_HIP_MAX_LEN);
  const int64_t pairs = n_news * heads;
  if (is_16bit(dt)) {
    const int nkt = (max_len + 31) / 32;
    const int lds_per_wave = nkt * 32 * 128 + 4096;   // V image + the 32-query output slab
    static bool lds_raised_dev[MAX_DEVICES] = {};     // 4 waves x 20 KiB exceeds the 64 KiB default of dynamic LDS
    bool& lds_raised = lds_raised_dev[current_device_slot()];   // the attribute is per device
    if (!lds_raised) {
      MANNER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bf16_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (4 * 32 * 128 + 4096)));
      MANNER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bf16_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (4 * 32 * 128 + 4096)));
      lds_raised = true;
    }
    if (dt == DT_F16)
      hipLaunchKernelGGL(attn_bf16_kernel<f16_t>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 4 * lds_per_wave, stream,
                         static_cast<const f16_t*>(qkv), static_cast<f16_t*>(ctx), cu, pairs, heads, H, lds_per_wave);
    else
      hipLaunchKernelGGL(attn_bf16_kernel<bf16_t>, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 4 * lds_per_wave, stream,
                         static_cast<const bf16_t*>(qkv), static_cast<bf16_t*>(ctx), cu, pairs, heads, H, lds_per_wave);
  } else {
    static const bool valu = getenv("MANNER_HIP_ATTN_F32_VALU") != nullptr;       // A/B switch: the older VALU kernel
    if (valu) {
      hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)pairs), dim3(128), 0, stream, static_cast<const float*>(qkv),
                         static_cast<float*>(ctx), cu, heads, H);
    } else {
      const int nkt = (max_len + 31) / 32;
      const int lds_per_wave = nkt * 32 * 256 + 8192;   // f32 V image (or its two 16-bit halves) + the 32-row slab
      static bool lds_raised_dev[MAX_DEVICES] = {};
      bool& lds_raised = lds_raised_dev[current_device_slot()];
      if (!lds_raised) {
        MANNER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_f32_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (4 * 32 * 256 + 8192)));
        MANNER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_x3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (4 * 32 * 256 + 8192)));
        MANNER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_x3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (4 * 32 * 256 + 8192)));
        MANNER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_x3_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (4 * 32 * 256 + 8192)));
        MANNER_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_x3_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (4 * 32 * 256 + 8192)));
        lds_raised = true;
      }
      // f16x3 mode (the rows go out as the f16 split operand): both products as split f16 products on the 16-bit matrix pipe
      // (round 5; MANNER_HIP_ATTN_X3=0 keeps the f32-MFMA kernel for A/B — read per launch, the tests flip it)
      const char* x3_env = getenv("MANNER_HIP_ATTN_X3");
      if (split_out && split_dt == DT_F16 && !(x3_env && atoi(x3_env) == 0)) {
#define MANNER_ATTN_X3(N_)                                                                                                   \
  hipLaunchKernelGGL(attn_x3_kernel<N_>, dim3((unsigned)((pairs + 1) / 2)), dim3(128), 2 * lds_per_wave, stream,                  \
                     static_cast<const float*>(qkv), cu, pairs, heads, H, lds_per_wave, static_cast<f16_t*>(split_out))
        if (nkt == 1) MANNER_ATTN_X3(1);
        else if (nkt == 2) MANNER_ATTN_X3(2);
        else if (nkt == 3) MANNER_ATTN_X3(3);
        else MANNER_ATTN_X3(4);
#undef MANNER_ATTN_X3
        MANNER_LAUNCH_CHECK();
        return MANNER_HIP_OK;
      }
      hipLaunchKernelGGL(attn_f32_mfma_kernel, dim3((unsigned)((pairs + 1) / 2)), dim3(128), 2 * lds_per_wave, stream,
                         static_cast<const float*>(qkv), static_cast<float*>(ctx), cu, pairs, heads, H, lds_per_wave, split_out,
                         (int)split_dt);
    }
  }
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

}  // namespace manner
