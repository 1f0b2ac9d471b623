// Y = epi(X W^T + b) on the gfx950 matrix cores — the PLM's Linear layers (K2, K4, K5, K6 of
// SURVEY.md §2.2; HF BertSelfAttention/BertSelfOutput/BertIntermediate/BertOutput dense layers,
// transformers/models/bert/modeling_bert.py:175-177, 289-293, 334-337, 347-351).
//
// Both operands are K-contiguous (activations [M,K], nn.Linear weight [N,K]), so one staging and
// fragment-read routine serves both.  The kernel computes the TRANSPOSED tile D[n][m] =
// sum_k W[n][k] X[m][k] (weights as the MFMA A operand): in the 32x32 accumulator layout a lane
// then owns one token m and runs of 4 consecutive features n, so the epilogue (bias, erf-GeLU,
// residual) is lane-local and stores are 8/16-byte vectors along the output row.
//
// Tile 128(m) x 128(n), 4 waves as 2(n) x 2(m), each 64x64 = 2x2 MFMA tiles of 32x32.
// K advances 128 BYTES per stage for either dtype (64 bf16 or 32 f32): the LDS image is
// [128 rows][128 B] per operand, filled by global_load_lds_dwordx4 (1 KiB = 8 rows per
// wave-instruction, lane-linear destination) and read back as 16-byte chunks.  Chunk c of row r
// is stored at chunk slot c ^ ((r >> 1) & 7): with 128-byte rows that makes every 16-lane group of
// a ds_read_b128 (rows r..r+31 at one chunk index) hit 16 distinct 16-byte slots of the 256-byte
// bank row.  The swizzle is applied on the per-lane global SOURCE address and again on the read
// (cdna_hip_programming.md rule 21).
//   bf16: one v_mfma_f32_32x32x16_bf16 per 16-byte chunk pair (lane half h holds k = 8h..8h+7).
//   f32 : four v_mfma_f32_32x32x2_f32 per chunk (element e of lane half h is k = 4h+e of the
//         chunk pair) — exact f32 FMA chains, the 1e-4 parity mode.
#include <stdlib.h>

#include <type_traits>

#include "train_common.h"
#include "gemm_w4_asm.inc"
#include "gemm_w8_asm.inc"
#ifdef MANNER_P4_LAB
#include "gemm_p4_asm.inc"
#endif

namespace manner {
namespace {

constexpr int BM = 128, BN = 128, ROW_BYTES = 128, TILE_BYTES = 128 * ROW_BYTES;  // 16 KiB
constexpr int STAGE_BYTES = 2 * TILE_BYTES;                                        // W + X

__device__ __forceinline__ float gelu_erf(float x) {
  // nn.functional.gelu default (transformers/activations.py ACT2FN["gelu"]): x/2 * (1 + erf(x/sqrt2))
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): used where the result is rounded to bf16
// (relative step 2^-9), i.e. the bf16 encoder's FFN; the f32 parity mode keeps erff.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.0f - p * t * __expf(-z * z);     // erf(|x|/sqrt2)
  return 0.5f * x * (1.0f + copysignf(e, x));
}
// d gelu / dx = (1 + erf(x/sqrt2)) / 2 + x * phi(x) with the same erf (its exp(-x^2/2) is phi's): the training backward's
// 16-bit epilogue (EPI_GELU_GRAD)
__device__ __forceinline__ float gelu_grad_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float ex = __expf(-z * z);
  const float e = 1.0f - p * t * ex;
  return fmaf(x * 0.39894228040143268f, ex, 0.5f * (1.0f + copysignf(e, x)));
}
// GeLU for the 256x256 bf16 kernel's epilogue, transcendental-free so that the compiler packs it two lanes-elements
// per instruction (v_pk_fma_f32):  gelu(x) = x/2 + |x| * P(min(|x|, 4.25)),  P ~ erf(t/sqrt2)/2 a degree-8 minimax
// fit (LP on 3001 nodes, P(4.25) = 1/2 exactly so x > 4.25 gives x and x < -4.25 gives 0).  Max |error| 7.4e-5
// evaluated in f32 — below the bf16 rounding step (2^-9 relative) of any output above 0.02; the f32 parity
// mode keeps erff.
__device__ __forceinline__ float gelu_poly(float x) {
  const float a = fabsf(x);
  const float t = fminf(a, 4.25f);
  float p = -2.944768248e-05f;
  p = fmaf(p, t, 4.707596855e-04f);
  p = fmaf(p, t, -2.366253315e-03f);
  p = fmaf(p, t, -2.697367422e-04f);
  p = fmaf(p, t, 4.205343857e-02f);
  p = fmaf(p, t, -1.310204100e-01f);
  p = fmaf(p, t, 4.837078524e-02f);
  p = fmaf(p, t, 3.822678287e-01f);
  p = fmaf(p, t, 1.922160747e-03f);
  return fmaf(a, p, 0.5f * x);
}
template <typename TIn> __device__ __forceinline__ float gelu_for(float x);
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }
template <> __device__ __forceinline__ float gelu_for<bf16_t>(float x) { return gelu_erf_fast(x); }
template <> __device__ __forceinline__ float gelu_for<f16_t>(float x) { return gelu_erf_fast(x); }

template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<f16_t> { typedef f16x8 type; };
template <> struct Frag<float> { typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ void mma_chunk(const typename Frag<T>::type& a, const typename Frag<T>::type& b,
                                          f32x16& acc);
template <>
__device__ __forceinline__ void mma_chunk<bf16_t>(const bf16x8& a, const bf16x8& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<f16_t>(const f16x8& a, const f16x8& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<float>(const f32x4& a, const f32x4& b, f32x16& acc) {
#pragma unroll
  for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
}

template <typename TOut>
__device__ __forceinline__ void store4(TOut* p, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}
template <>
__device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
}
template <>
__device__ __forceinline__ void store4<f16_t>(f16_t* p, float a, float b, float c, float d) {
  *reinterpret_cast<f16x4*>(p) = f16x4{(f16_t)a, (f16_t)b, (f16_t)c, (f16_t)d};
}
template <typename T>
__device__ __forceinline__ void load4(const T* p, float v[4]);
template <>
__device__ __forceinline__ void load4<f16_t>(const f16_t* p, float v[4]) {
  f16x4 t = *reinterpret_cast<const f16x4*>(p);
  v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}
template <>
__device__ __forceinline__ void load4<float>(const float* p, float v[4]) {
  f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <>
__device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float v[4]) {
  bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
  v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}

template <typename TIn, typename TOut, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(
    const TIn* __restrict__ X, const TIn* __restrict__ W, const float* __restrict__ bias,
    const TIn* __restrict__ R, TOut* __restrict__ Y, int N, int K, const int* __restrict__ m_total,
    int n_tiles) {
  constexpr int EPC = 16 / sizeof(TIn);          // elements per 16-byte chunk
  constexpr int BK = ROW_BYTES / sizeof(TIn);    // elements of K per stage
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE_BYTES];

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
  // contiguous run of tiles (bijective form for any grid size).  Tiles that share an activation
  // row-panel are consecutive, so the panel is fetched into one XCD's L2 once.
  const int nwg = gridDim.x, b = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  const int mt = t / n_tiles, nt = t - mt * n_tiles;
  const int M = *m_total;
  if (mt * BM >= M) return;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int rr = lane & 31, h = lane >> 5;
  const int wn = wave >> 1, wm = wave & 1;

  // staging: wave w fills pieces 4w..4w+3 (8 rows each) of both operand tiles
  const int srow = lane >> 3, sdst = lane & 7;
  const TIn* wsrc[4];
  const TIn* xsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (4 * wave + i) + srow;
    const int csrc = sdst ^ ((row >> 1) & 7);
    wsrc[i] = W + (size_t)(nt * BN + row) * K + csrc * EPC;
    xsrc[i] = X + (size_t)(mt * BM + row) * K + csrc * EPC;
  }
  auto stage = [&](int buf, int k0) {
    char* base = lds + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wsrc[i] + k0), LDS_PTR(base + (4 * wave + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(xsrc[i] + k0), LDS_PTR(base + TILE_BYTES + (4 * wave + i) * 1024),
                                       16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int swz = (rr >> 1) & 7;   // rows 32i + rr (+64 wn): (row >> 1) & 7 == (rr >> 1) & 7
  const int woff = (wn * 64 + rr) * ROW_BYTES;
  const int xoff = TILE_BYTES + (wm * 64 + rr) * ROW_BYTES;

  const int nk = K / BK;
  stage(0, 0);
  __syncthreads();   // emits vmcnt(0) for the pending LDS-DMA
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
    const char* base = lds + cur * STAGE_BYTES;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      const int coff = ((2 * kc + h) ^ swz) << 4;
      typename Frag<TIn>::type a[2], bb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const typename Frag<TIn>::type*>(base + woff + i * 32 * ROW_BYTES + coff);
        bb[i] = *reinterpret_cast<const typename Frag<TIn>::type*>(base + xoff + i * 32 * ROW_BYTES + coff);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma_chunk<TIn>(a[i], bb[j], acc[i][j]);
    }
    __syncthreads();
  }

  // epilogue: acc[i][j][reg] = D[n][m], n = 32i + (reg&3) + 8(reg>>2) + 4h, m = 32j + rr
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = mt * BM + wm * 64 + 32 * j + rr;
    if (m >= M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n0 = nt * BN + wn * 64 + 32 * i + 8 * g + 4 * h;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n0);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bv[e];
        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_for<TIn>(v[e]);
        }
        if (EPI == EPI_BIAS_RES) {
          float r[4];
          load4<TIn>(R + (size_t)m * N + n0, r);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += r[e];
        }
        store4<TOut>(Y + (size_t)m * N + n0, v[0], v[1], v[2], v[3]);
      }
    }
  }
}

template <typename TIn, typename TOut>
int launch(Epilogue epi, const void* X, const void* W, const float* bias, const void* R, void* Y,
           int64_t m_bound, int N, int K, const int* m_total, hipStream_t stream) {
  const int n_tiles = N / BN;
  const int64_t grid = (m_bound / BM) * n_tiles;
  if (grid <= 0 || grid > 0x7fffffff) return fail(MANNER_HIP_E_INVALID, "gemm grid %lld out of range", (long long)grid);
  dim3 g((unsigned)grid), b(256);
  const TIn* x = static_cast<const TIn*>(X);
  const TIn* w = static_cast<const TIn*>(W);
  const TIn* r = static_cast<const TIn*>(R);
  TOut* y = static_cast<TOut*>(Y);
  switch (epi) {
    case EPI_BIAS:
      hipLaunchKernelGGL((gemm_tn_kernel<TIn, TOut, EPI_BIAS>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles);
      break;
    case EPI_BIAS_GELU:
      hipLaunchKernelGGL((gemm_tn_kernel<TIn, TOut, EPI_BIAS_GELU>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles);
      break;
    case EPI_BIAS_RES:
      hipLaunchKernelGGL((gemm_tn_kernel<TIn, TOut, EPI_BIAS_RES>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles);
      break;
    default:
      return fail(MANNER_HIP_E_INVALID, "gemm epilogue %d has its own entry point", (int)epi);
  }
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}


// ---------------------------------------------------------------------------------------------
// Small problems (the training path at the reference's batch size: 8 impressions = 1.5 - 8 k tokens per encoder call): the
// persistent 256x256 kernel below then has 20 - 90 tiles for 256 CUs and a launch lasts one whole tile — 12 - 48 K-steps plus
// a 128 KB epilogue, 35 - 90 us — whatever the problem size.  This is the 128x128 kernel above with a DEEPER pipeline instead
// (4 LDS stages = 128 KiB, one workgroup per CU, K-step kt+3 requested while kt is multiplied: the ~2 us a first-touch DMA takes
// is spread over three K-steps) and 4x the tiles; 16-bit operands, the training path's two epilogues:
//   EPI_BIAS          Y = X W^T + bias                          (f32 or 16-bit out)
//   EPI_BIAS_RES_F32  Y = dropout(X W^T + bias) + R             (f32 residual / out, the counter-based bits of train_common.h)
// Round 4: the kernel runs on v_mfma_f32_16x16x32 like the persistent 256x256 kernel — per output element the SAME sequence of
// matrix instructions over K (k ascending, 32 per instruction, zero-initialised accumulator), the same epilogue expressions in the
// same order — so a problem gives the same BITS whichever of the two kernels takes it (tests hold them to torch.equal).  That is
// what lets the inference engine send its small calls here too (EPI_NORM / EPI_NORM_GELU / EPI_NRES of the deferred-LayerNorm
// schedule): a handful of unseen news behind the embedding cache, an A-Module batch, the reference's batch of 8 — a
// 256x256x3072 tile is 45 us on one CU however few tiles the launch has; 128x128 tiles are 4x as many and a quarter each.
struct SmallAux {
  const float* vec;       // EPI_NORM*: c1 [N];  EPI_NRES: gamma [N]
  const float2* mr;       // {mean, rstd} per row (EPI_NORM*: of X;  EPI_NRES: of the residual rows in Y)
  float2* part;           // EPI_NRES: [N/64][part_stride] partial {sum, sum of squares} of the rows written
  int64_t part_stride;
  int tail_cus;           // > 0: TAIL mode — this launch takes the row panels the persistent kernel leaves (split_panels)
};

// Round-aware split of a launch (round 4; opt-in, MANNER_HIP_GEMM_TAIL_SPLIT=1 — a measured near-zero, see gemm_tn_dln).  A persistent launch of `tiles` 256x256 tiles on `cus` workgroups lasts ceil(tiles / cus)
// whole tile times, so 282 tiles (94 row panels x N = 768: the history call of a batch of 8 impressions) take two rounds with 26
// tiles in the second.  Because the 128x128 kernel gives the same bits, the launch can be cut at a row panel: the persistent kernel
// takes the panels that fill r whole rounds, the 128x128 kernel the rest — if that rest is at most ONE round of small tiles (else
// the persistent kernel keeps everything).  Both kernels evaluate this function on the device token count, so they always agree.
__host__ __device__ inline int split_panels(int m_tiles, int n_tiles256, int cus) {
  const int tiles = m_tiles * n_tiles256;
  const int r = tiles / cus;
  if (cus <= 0 || r < 1 || tiles == r * cus) return m_tiles;
  const int p0 = (r * cus) / n_tiles256;
  if ((m_tiles - p0) * 4 * n_tiles256 > cus) return m_tiles;
  return p0;
}

template <typename TE, typename TOut, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_tn_small_kernel(
    const TE* __restrict__ X, const TE* __restrict__ W, const float* __restrict__ bias,
    const float* __restrict__ R, TOut* Y, int N, int K, const int* __restrict__ m_total,
    int n_tiles, Drop drop, const int32_t* __restrict__ rowmap, SmallAux aux) {
  typedef typename E16<TE>::v8 e16x8;
  typedef typename E16<TE>::v4 e16x4;
  constexpr int STAGES = 4;
  constexpr int EPC = 8, BK = 64;
  __shared__ __attribute__((aligned(1024))) char lds[STAGES * STAGE_BYTES];

  const int nwg = gridDim.x, b = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  int mt = t / n_tiles;
  const int nt = t - mt * n_tiles;
  const int M = *m_total;
  if (aux.tail_cus > 0) {                                // the panels behind the persistent kernel's whole rounds
    const int m256 = (M + 255) / 256;
    const int p0 = split_panels(m256, n_tiles / 2, aux.tail_cus);
    if (p0 >= m256) return;
    mt += 2 * p0;
  }
  if (mt * BM >= M) return;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l15 = lane & 15, lq = lane >> 4;
  const int wn = wave >> 1, wm = wave & 1;

  const int srow = lane >> 3, sdst = lane & 7;
  const TE* wsrc[4];
  const TE* xsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * (4 * wave + i) + srow;
    const int csrc = sdst ^ ((row >> 1) & 7);            // global chunk c of row r lands in slot c ^ ((r >> 1) & 7)
    wsrc[i] = W + (size_t)(nt * BN + row) * K + csrc * EPC;
    xsrc[i] = X + (size_t)(mt * BM + row) * K + csrc * EPC;
  }
  auto stage = [&](int buf, int k0) {                  // 8 LDS-DMA instructions per wave
    char* base = lds + buf * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(wsrc[i] + k0), LDS_PTR(base + (4 * wave + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(xsrc[i] + k0), LDS_PTR(base + TILE_BYTES + (4 * wave + i) * 1024), 16, 0, 0);
    }
  };

  // wave tile 64 (n) x 64 (m) = 4 x 4 tiles of 16 x 16: acc[a][bb][e] = D[n = 16a + 4lq + e][m = 16bb + l15]
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) acc[a][bb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int swz = (l15 >> 1) & 7;
  const int woff = (wn * 64 + l15) * ROW_BYTES;
  const int xoff = TILE_BYTES + (wm * 64 + l15) * ROW_BYTES;

  const int nk = K / BK;
#pragma unroll
  for (int s0 = 0; s0 < STAGES - 1; ++s0)
    if (s0 < nk) stage(s0, s0 * BK);
  for (int kt = 0; kt < nk; ++kt) {
    // requested so far: K-steps < min(nk, kt + STAGES - 1); K-step kt must have landed, the younger ones may fly on
    const int ahead = min(nk, kt + STAGES - 1) - (kt + 1);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // everyone's pieces of kt landed; everyone left the stage of kt - 1
    if (kt + STAGES - 1 < nk) stage((kt + STAGES - 1) % STAGES, (kt + STAGES - 1) * BK);
    const char* base = lds + (kt % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {                   // the two 32-element slices of the K-step, in order
      const int coff = ((4 * s2 + lq) ^ swz) << 4;
      e16x8 wf[4], xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wf[i] = *reinterpret_cast<const e16x8*>(base + woff + i * 16 * ROW_BYTES + coff);
        xf[i] = *reinterpret_cast<const e16x8*>(base + xoff + i * 16 * ROW_BYTES + coff);
      }
#pragma unroll
      for (int bb = 0; bb < 4; ++bb)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][bb] = E16<TE>::mfma16(wf[a], xf[bb], acc[a][bb]);
    }
  }

  // ---- epilogue, straight from the MFMA layout (8- or 16-byte stores per lane: a small problem's stores are not its bottleneck).
  // Every expression below is the persistent kernel's, operand for operand.
  const int nbase = nt * BN + wn * 64;
  f32x4 bv[4], cv[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    bv[a] = *reinterpret_cast<const f32x4*>(bias + nbase + 16 * a + 4 * lq);
    if constexpr (EPI == EPI_NORM || EPI == EPI_NORM_GELU || EPI == EPI_NRES) cv[a] = *reinterpret_cast<const f32x4*>(aux.vec + nbase + 16 * a + 4 * lq);
  }
#pragma unroll
  for (int bb = 0; bb < 4; ++bb) {
    const int m = mt * BM + wm * 64 + 16 * bb + l15;
    const int mc = min(m, M - 1);
    float2 ms = float2{0.f, 0.f};
    if constexpr (EPI == EPI_NORM || EPI == EPI_NORM_GELU || EPI == EPI_NRES) ms = aux.mr[mc];
    if constexpr (EPI == EPI_NRES) {
      // raw' = acc + (bias + beta) + ((raw - mean) * rstd) * gamma in place, + the row's {sum, sum of squares} over the wave's 64 columns
      const TE* rrow = reinterpret_cast<const TE*>(Y) + (size_t)mc * N + nbase;
      const float rs = ms.y, nm = -ms.x * ms.y;
      float p1 = 0.f, p2 = 0.f;
      e16x4 res[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) res[a] = *reinterpret_cast<const e16x4*>(rrow + 16 * a + 4 * lq);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float r = fmaf((float)res[a][e], rs, nm);
          const float x = fmaf(r, cv[a][e], acc[a][bb][e] + bv[a][e]);
          acc[a][bb][e] = x;
          p1 += x;
          p2 = fmaf(x, x, p2);
        }
      p1 += __shfl_xor(p1, 16, 64); p2 += __shfl_xor(p2, 16, 64);
      p1 += __shfl_xor(p1, 32, 64); p2 += __shfl_xor(p2, 32, 64);
      if (lq == 0 && m < M) aux.part[(size_t)(2 * nt + wn) * aux.part_stride + m] = float2{p1, p2};
      if (m < M) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
          store4<TOut>(Y + (size_t)m * N + nbase + 16 * a + 4 * lq, acc[a][bb][0], acc[a][bb][1], acc[a][bb][2], acc[a][bb][3]);
      }
      continue;
    }
    if (m >= M) continue;
    const uint64_t drow = EPI == EPI_BIAS_RES_F32 ? (uint64_t)(rowmap ? rowmap[m] : m) * N : 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int n0 = nbase + 16 * a + 4 * lq;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if constexpr (EPI == EPI_NORM || EPI == EPI_NORM_GELU) v[e] = fmaf(ms.y, fmaf(-ms.x, cv[a][e], acc[a][bb][e]), bv[a][e]);
        else v[e] = acc[a][bb][e] + bv[a][e];
      }
      if constexpr (EPI == EPI_NORM_GELU || EPI == EPI_BIAS_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = sizeof(TOut) == 4 ? gelu_erf(v[e]) : gelu_poly(v[e]);
      }
      if constexpr (EPI == EPI_BIAS_RES) {               // the [CLS] tail's projections: residual of the operand type, f32 out
        const e16x4 rv = *reinterpret_cast<const e16x4*>(reinterpret_cast<const TE*>(R) + (size_t)m * N + n0);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
      }
      if constexpr (EPI == EPI_BIAS_RES_F32) {
        const f32x4 rv = *reinterpret_cast<const f32x4*>(R + (size_t)m * N + n0);
        if (drop.thr != 0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = drop_bits(drop.seed, drop.site, drow + n0 + e) >= drop.thr ? v[e] * drop.scale : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rv[e];
      }
      store4<TOut>(Y + (size_t)m * N + n0, v[0], v[1], v[2], v[3]);
    }
  }
}

// few 256x256 tiles -> the deep-pipeline 128x128 kernel (A/B: MANNER_HIP_GEMM_SMALL_TILES=0 turns it off, =n moves the threshold)
bool small_problem(int64_t m_bound, int N) {
  const char* e = getenv("MANNER_HIP_GEMM_SMALL_TILES");          // read per launch: the tests flip it
  const int thr = e ? atoi(e) : 128;
  return (m_bound / 256) * (N / 256) <= thr;
}
template <typename TIn, typename TOut, int EPI>
int launch_small(const void* X, const void* W, const float* bias, const float* R, void* Y, int64_t m_bound, int N, int K,
                 const int* m_total, const Drop& drop, const int32_t* rowmap, hipStream_t stream, const SmallAux& aux = SmallAux{}) {
  const int n_tiles = N / BN;
  int64_t grid = (m_bound / BM) * n_tiles;
  if (aux.tail_cus > 0 && grid > aux.tail_cus) grid = aux.tail_cus;          // a tail is at most one round of small tiles
  if (grid <= 0 || grid > 0x7fffffff) return fail(MANNER_HIP_E_INVALID, "gemm grid %lld out of range", (long long)grid);
  hipLaunchKernelGGL((gemm_tn_small_kernel<TIn, TOut, EPI>), dim3((unsigned)grid), dim3(256), 0, stream, static_cast<const TIn*>(X),
                     static_cast<const TIn*>(W), bias, R, static_cast<TOut*>(Y), N, K, m_total, n_tiles, drop, rowmap, aux);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Main kernel: 256(m) x 256(n) tile, 8 waves as 2(m) x 4(n), wave tile 128(m) x 64(n) = 4 x 2 MFMA
// tiles (128 accumulator VGPRs), BK = 128 bytes, two 64 KiB LDS stages.  Per K-step a wave issues
// 8 LDS-DMA pieces, 24 ds_read_b128 and 32 MFMAs (the 128x128 kernel above: 8 / 16 / 16), and the
// K-chunk loop is software-pipelined in registers: the fragments of chunk kc+1 and two DMA pieces
// of the NEXT K-step are issued before the 8 MFMAs of chunk kc, so the matrix pipe of a SIMD (two
// waves) is not left idle while its waves issue LDS reads and DMA.
//   iteration kt:  s_waitcnt vmcnt(0)   my pieces of tile kt (issued during iteration kt-1) landed
//                  s_barrier            everyone's pieces landed; everyone finished reading buf^1
//                  for kc: read frags(kc+1); issue 2 pieces of tile kt+1 -> buf^1; 8 MFMAs(kc)
constexpr int G_BM = 256, G_BN = 256;
constexpr int G_OP_BYTES = 256 * ROW_BYTES, G_STAGE_BYTES = 2 * G_OP_BYTES;   // 32 KiB per operand

// ABL (lab only, tools/gemm_lab.hip): 0 production; 1 no epilogue stores; 2 no DMA inside the loop;
// 3 no fragment reads inside the loop.  Timing-only builds: outputs are wrong for ABL != 0.
template <typename TIn, typename TOut, int EPI, int ABL = 0>
__global__ __launch_bounds__(512, 1) void gemm_tn_big_kernel(
    const TIn* __restrict__ X, const TIn* __restrict__ W, const float* __restrict__ bias,
    const TIn* __restrict__ R, TOut* __restrict__ Y, int N, int K, const int* __restrict__ m_total,
    int n_tiles) {
  constexpr int EPC = 16 / sizeof(TIn);
  constexpr int BK = ROW_BYTES / sizeof(TIn);
  typedef typename Frag<TIn>::type frag_t;
  __shared__ __attribute__((aligned(1024))) char lds[2 * G_STAGE_BYTES];   // 128 KiB

  const int nwg = gridDim.x, b = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (b >> 3);
  const int mt = t / n_tiles, nt = t - mt * n_tiles;
  const int M = *m_total;
  if (mt * G_BM >= M) return;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int rr = lane & 31, h = lane >> 5;
  const int wn = wave & 3, wm = wave >> 2;

  // staging: 64 pieces (1 KiB = 8 rows) per stage; waves 0-3 bring the weight tile, 4-7 the
  // activation tile, 8 consecutive pieces (64 rows) each
  const bool is_w = wave < 4;
  const int prow0 = 64 * (wave & 3);                                  // first row of this wave's pieces
  const TIn* gbase = is_w ? W + (size_t)(nt * G_BN + prow0) * K : X + (size_t)(mt * G_BM + prow0) * K;
  const int ldst0 = (is_w ? 0 : G_OP_BYTES) + prow0 * ROW_BYTES;
  // piece i covers rows prow0 + 8i + (lane>>3); (row>>1)&7 = (4i + (lane>>4)) & 7
  const int lrow = lane >> 3;
  int voff[2];                                                         // even / odd pieces
#pragma unroll
  for (int par = 0; par < 2; ++par)
    voff[par] = lrow * K + (((lane & 7) ^ ((4 * par + (lane >> 4)) & 7)) * EPC);
  auto issue2 = [&](int buf, int k0, int pair) {                       // pieces 2*pair, 2*pair+1
    char* base = lds + buf * G_STAGE_BYTES + ldst0;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int i = 2 * pair + par;
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gbase + (size_t)(8 * i) * K + k0 + voff[par]),
                                       LDS_PTR(base + i * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int swz = (rr >> 1) & 7;
  const int woff = (wn * 64 + rr) * ROW_BYTES;
  const int xoff = G_OP_BYTES + (wm * 128 + rr) * ROW_BYTES;
  auto read_frags = [&](const char* base, int kc, frag_t (&wf)[2], frag_t (&xf)[4]) {
    const int coff = ((2 * kc + h) ^ swz) << 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const frag_t*>(base + woff + i * 32 * ROW_BYTES + coff);
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const frag_t*>(base + xoff + j * 32 * ROW_BYTES + coff);
  };
  auto mma8 = [&](const frag_t (&wf)[2], const frag_t (&xf)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma_chunk<TIn>(wf[i], xf[j], acc[i][j]);
  };

  // one K-step on buffer `cur`; NEXT: also issue the DMA of the following K-step into cur^1
  auto kstep = [&](int cur, int k1, auto next_tag) {
    constexpr bool NEXT = decltype(next_tag)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* base = lds + cur * G_STAGE_BYTES;
    frag_t wa[2], xa[4], wb[2], xb[4];
    if (ABL != 3 || k1 == BK) read_frags(base, 0, wa, xa);
    if (ABL != 3 || k1 == BK) read_frags(base, 1, wb, xb);
    if (NEXT && ABL != 2) issue2(cur ^ 1, k1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma8(wa, xa);
    __builtin_amdgcn_sched_barrier(0);
    if (ABL != 3) read_frags(base, 2, wa, xa);
    if (NEXT && ABL != 2) issue2(cur ^ 1, k1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma8(wb, xb);
    __builtin_amdgcn_sched_barrier(0);
    if (ABL != 3) read_frags(base, 3, wb, xb);
    if (NEXT && ABL != 2) issue2(cur ^ 1, k1, 2);
    __builtin_amdgcn_sched_barrier(0);
    mma8(wa, xa);
    __builtin_amdgcn_sched_barrier(0);
    if (NEXT && ABL != 2) issue2(cur ^ 1, k1, 3);
    __builtin_amdgcn_sched_barrier(0);
    mma8(wb, xb);
  };

  const int nk = K / BK;
#pragma unroll
  for (int pair = 0; pair < 4; ++pair) issue2(0, 0, pair);
  for (int kt = 0; kt + 1 < nk; ++kt) kstep(kt & 1, (kt + 1) * BK, std::true_type{});
  kstep((nk - 1) & 1, 0, std::false_type{});

  // ---- epilogue.  acc[i][j][reg] = D[n][m], n = 32i + (reg&3) + 8(reg>>2) + 4h, m = 32j + rr.
  // Direct stores from this layout are 8-byte pieces at a row stride (store-issue bound: they cost
  // 38 % of the K=768 GEMMs).  Instead each wave passes its 32(m) x 64(n) slabs through a private
  // LDS slab (XOR-swizzled 16-byte chunks) and writes whole 128/256-byte output rows with 16-byte
  // accesses; the residual is added on the row-contiguous side, so its loads are coalesced too.
  __builtin_amdgcn_s_barrier();                      // all waves finished reading the K-loop stages
  constexpr int OUT_ROW = 64 * sizeof(TOut);         // 128 B (bf16) / 256 B (f32) per token row
  constexpr int CHUNKS = OUT_ROW / 16;               // 8 / 16 chunks of 16 bytes
  constexpr int OPC = 16 / sizeof(TOut);             // outputs per chunk: 8 / 4
  char* slab = lds + wave * (32 * OUT_ROW);
  const int nbase = nt * G_BN + wn * 64;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = 32 * i + 8 * g + 4 * h;        // local n of the lane's 4 values
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + nbase + nl);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bv[e];
        if (EPI == EPI_BIAS_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_for<TIn>(v[e]);
        }
        const int c = nl / OPC;                       // logical chunk; bf16: + 8-byte half h
        const int off = rr * OUT_ROW + ((c ^ (rr & (CHUNKS - 1))) << 4) + (sizeof(TOut) == 2 ? 8 * h : 0);
        store4<TOut>(reinterpret_cast<TOut*>(slab + off), v[0], v[1], v[2], v[3]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int ROWS_PER_INST = 64 / CHUNKS;        // 8 / 4 token rows per wave-instruction
    constexpr int NQ = 32 / ROWS_PER_INST;
    const int row0 = lane / CHUNKS, sl = lane % CHUNKS;
    float res[EPI == EPI_BIAS_RES ? NQ : 1][4];
    if (EPI == EPI_BIAS_RES && ABL != 1) {            // all residual loads of the slab in flight at once
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int row = q * ROWS_PER_INST + row0;
        const int m = min(mt * G_BM + wm * 128 + 32 * j + row, M - 1);
        load4<TIn>(R + (size_t)m * N + nbase + (sl ^ (row & (CHUNKS - 1))) * OPC, res[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int row = q * ROWS_PER_INST + row0;
      const int c = sl ^ (row & (CHUNKS - 1));
      const int m = mt * G_BM + wm * 128 + 32 * j + row;
      f32x4 raw = *reinterpret_cast<const f32x4*>(slab + row * OUT_ROW + (sl << 4));
      if (ABL == 1) { asm volatile("" ::"v"(raw)); continue; }
      if (EPI == EPI_BIAS_RES) {                      // TOut == float here: 4 outputs per chunk
#pragma unroll
        for (int e = 0; e < 4; ++e) raw[e] += res[q][e];
      }
      if (m < M)
        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(Y) + ((size_t)m * N + nbase + c * OPC) * sizeof(TOut)) = raw;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// K11 logits on the f32 matrix cores: logits[r] = sum_{j<Q} tanh(<x[r,:], W[j,:]> + b[j]) q[j]   (AdditiveAttention,
// reference manner/models/components/attention.py:21-25).  The x W^T product is the gemm_tn_big_kernel<float> main
// loop (exact f32 FMA chains on v_mfma_f32_32x32x2_f32, 256 rows x 256 columns per workgroup, BK = 32 floats); Q <= 256
// so ONE column tile holds every attention unit and the tanh / dot-with-query reduction happens in the epilogue —
// the [R, Q] intermediate never exists.  Operands are the caller's tensors as they are: rows past R (or past Q for W)
// are CLAMPED to the last valid row on the staging address (no padded copies, nothing read out of bounds) and their
// contributions dropped (rows >= R are not written, units >= Q carry q = 0).
__global__ __launch_bounds__(512, 1) void pool_logits_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                             const float* __restrict__ bias, const float* __restrict__ query,
                                                             int64_t R, int K, int Q, float* __restrict__ logits) {
  constexpr int EPC = 4, BK = 32;
  typedef f32x4 frag_t;
  __shared__ __attribute__((aligned(1024))) char lds[2 * G_STAGE_BYTES];   // 128 KiB
  const int64_t mt = blockIdx.x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int rr = lane & 31, h = lane >> 5;
  const int wn = wave & 3, wm = wave >> 2;

  const bool is_w = wave < 4;
  const int prow0 = 64 * (wave & 3);
  const float* gsrc = is_w ? W : X;
  const int64_t row_base = is_w ? 0 : mt * G_BM, row_last = (is_w ? (int64_t)Q : R) - 1;
  const int ldst0 = (is_w ? 0 : G_OP_BYTES) + prow0 * ROW_BYTES;
  const int lrow = lane >> 3;
  uint32_t poff[8];                              // element offset of this lane's 16 bytes in piece i (R * K < 2^31: launcher)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t row = min(row_base + prow0 + 8 * i + lrow, row_last);
    poff[i] = (uint32_t)(row * K) + (((lane & 7) ^ ((4 * (i & 1) + (lane >> 4)) & 7)) * EPC);
  }
  auto issue2 = [&](int buf, int k0, int pair) {
    char* base = lds + buf * G_STAGE_BYTES + ldst0;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int i = 2 * pair + par;
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gsrc + poff[i] + k0), LDS_PTR(base + i * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int swz = (rr >> 1) & 7;
  const int woff = (wn * 64 + rr) * ROW_BYTES;
  const int xoff = G_OP_BYTES + (wm * 128 + rr) * ROW_BYTES;
  auto read_frags = [&](const char* base, int kc, frag_t (&wf)[2], frag_t (&xf)[4]) {
    const int coff = ((2 * kc + h) ^ swz) << 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const frag_t*>(base + woff + i * 32 * ROW_BYTES + coff);
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const frag_t*>(base + xoff + j * 32 * ROW_BYTES + coff);
  };
  auto mma8 = [&](const frag_t (&wf)[2], const frag_t (&xf)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) mma_chunk<float>(wf[i], xf[j], acc[i][j]);
  };
  auto kstep = [&](int cur, int k1, auto next_tag) {
    constexpr bool NEXT = decltype(next_tag)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* base = lds + cur * G_STAGE_BYTES;
    frag_t wa[2], xa[4], wb[2], xb[4];
    read_frags(base, 0, wa, xa);
    read_frags(base, 1, wb, xb);
    if (NEXT) issue2(cur ^ 1, k1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma8(wa, xa);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(base, 2, wa, xa);
    if (NEXT) issue2(cur ^ 1, k1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma8(wb, xb);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(base, 3, wb, xb);
    if (NEXT) issue2(cur ^ 1, k1, 2);
    __builtin_amdgcn_sched_barrier(0);
    mma8(wa, xa);
    __builtin_amdgcn_sched_barrier(0);
    if (NEXT) issue2(cur ^ 1, k1, 3);
    __builtin_amdgcn_sched_barrier(0);
    mma8(wb, xb);
  };
  const int nk = K / BK;
#pragma unroll
  for (int pair = 0; pair < 4; ++pair) issue2(0, 0, pair);
  for (int kt = 0; kt + 1 < nk; ++kt) kstep(kt & 1, (kt + 1) * BK, std::true_type{});
  kstep((nk - 1) & 1, 0, std::false_type{});

  // ---- epilogue.  acc[i][j][reg] = D[n][m], n = 64 wn + 32i + (reg&3) + 8(reg>>2) + 4h, m = 128 wm + 32j + rr:
  // a lane sums tanh(.) q over its 32 units, the two lane halves combine by one shuffle, the four unit-quarter waves
  // through LDS in a fixed order (deterministic).
  __builtin_amdgcn_s_barrier();                      // all waves finished reading the K-loop stages
  float* red = reinterpret_cast<float*>(lds);        // [4][256]
  float p[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n0 = 64 * wn + 32 * i + 8 * g + 4 * h;
      float bv[4], qv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = n0 + e < Q;
        bv[e] = ok ? bias[n0 + e] : 0.f;
        qv[e] = ok ? query[n0 + e] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // tanh(z) = 1 - 2 / (exp(2z) + 1) on the hardware exp / rcp (|z| clamped: tanh(+-15) is +-1 in f32);
          // absolute error ~1e-7, far inside the 1e-4 bar of the pooled vectors
          const float z = fminf(fmaxf(acc[i][j][4 * g + e] + bv[e], -15.f), 15.f);
          const float t = 1.0f - 2.0f * __frcp_rn(__expf(2.0f * z) + 1.0f);
          p[j] = fmaf(t, qv[e], p[j]);
        }
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    p[j] += __shfl_xor(p[j], 32, 64);
    if (h == 0) red[wn * 256 + wm * 128 + 32 * j + rr] = p[j];
  }
  __syncthreads();
  if (tid < 256) {
    const int64_t r = mt * G_BM + tid;
    if (r < R) logits[r] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 production kernel: the 256x256 / 8-wave / 2-stage structure of gemm_tn_big_kernel on
// v_mfma_f32_16x16x32_bf16, PERSISTENT over tiles.
//  * MFMA shape: on MI355X with random operands a register-only MFMA loop sustains 2.04 PFLOP/s with
//    16x16x32 against 1.68-1.77 with 32x32x16 (the chip holds a higher clock on it —
//    cdna_hip_programming.md rule 28).  Wave tile 128(m) x 64(n) = 8 x 4 MFMA tiles,
//    acc[a][b] = D[n = 16a + 4(l>>4) + reg][m = 16b + (l&15)]: a lane still owns one token and runs
//    of 4 consecutive features.
//  * K-step (64 bf16) = two k32 steps x two m-halves = 4 chunks of 16 MFMAs; weight fragments of a
//    k32 step (4) and activation fragments of a chunk (4) are double-buffered in registers and read
//    one chunk ahead.  The step is rotated so its only barrier sits between chunks 2 and 3:
//      entry : fragments of chunk 0 in (w0, xa); the weight half of DMA(k+1) already issued into cur^1
//      c0    : read chunk 1; waves 4-7 issue the activation half of DMA(k+1);   16 MFMAs
//      c1,c2 : read chunks 2, 3;                                      16 MFMAs each
//      sync  : lgkmcnt(0) (every read of `cur` completed), vmcnt(0) (own DMA(k+1) landed), s_barrier
//      c3    : read chunk 0 of step k+1 from cur^1; waves 0-3 issue the weight half of DMA(k+2) into `cur`;
//              16 MFMAs   (DMA issue staggered between the two waves of a SIMD, see kstep)
//  * Persistent: gridDim.x = #CUs workgroups walk the tiles (round r, XCD-remapped slot).  In the
//    last two K-steps of a tile the stage that has just been freed receives K-step 0 of the workgroup's
//    NEXT tile, so the ~3 us DMA latency that used to open every tile is covered by the epilogue.  The epilogue slabs
//    live in the stage the last K-step just finished with; the next tile's first barrier separates
//    them from its DMA(1), and
//    the next tile starts with a COUNTED vmcnt that leaves exactly the epilogue's stores in flight.
// Deferred LayerNorm (bf16 path): the residual stream holds the PRE-LayerNorm sums `raw` (bf16) plus, per row,
// {mean, rstd}; LayerNorm itself never runs as a kernel.
//   consumer GEMM (EPI_NORM / EPI_NORM_GELU), A = raw, W' = gamma o W folded at pack time:
//       LN(raw) W^T + b  =  rstd * (raw W'^T - mean * c1) + c2,   c1[n] = sum_k W'[n,k],  c2[n] = b[n] + sum_k beta[k] W[n,k]
//   producer GEMM (EPI_NRES), in place on the residual stream:  raw' = acc + bias + LN(raw), the residual rebuilt in
//       f32 from raw and its row statistics; each wave also emits the {sum, sum of squares} of its 64 output
//       columns per row (`part`, group-major), reduced to the next {mean, rstd} by dln_finalize (rowops.hip) — fixed order,
//       no atomics, so results do not depend on scheduling.
struct DlnAux {
  const float* vec;       // EPI_NORM*: c1 [N];  EPI_NRES: gamma [N] of the LayerNorm that produced the residual
  const float2* mr;       // {mean, rstd} per row of the deferred operand (EPI_NORM*: of X;  EPI_NRES: of R)
  float2* part;           // EPI_NRES: [N/64][m_bound] partial {sum, sum of squares} of the rows written (group-major:
                          // a wave's 16 rows per store instruction are one contiguous 128-byte line)
  int64_t part_stride;    // m_bound
  int col_group;          // column tiles per pass over the rows (0 = all): see the tile order in the kernel
  int plain_stores;       // A/B (MANNER_HIP_NT_STORES=0): the streaming Q|K|V / FFN outputs with default-policy instead of
                          // non-temporal stores, so that a small chunk's intermediates may stay in the 256 MiB Infinity Cache
  // EPI_BIAS only: gridDim.y independent problems of the same shape (problem y reads X + y * batch_x, W + y * batch_w
  // and writes Y + y * batch_y, elements) — the split-K slices of the training path's weight-gradient GEMMs
  int64_t batch_x, batch_w, batch_y;
  // EPI_BIAS_RES_F32 only (training forward): Y = dropout(X W^T + bias) + R with the training path's counter-based bits
  // (element index drop_rowmap[m] * N + n, or m * N + n); drop_thr == 0: no dropout
  uint64_t drop_seed;
  uint32_t drop_site, drop_thr;
  float drop_scale;
  const int32_t* drop_rowmap;
  // A/B (MANNER_HIP_GEMM_STAGGER=n): row panel p starts (p % 4) * n sleeps of ~2 us late, so that the workgroups of a launch do
  // not all reach their epilogue — a burst of 128 KB reads + 128 KB writes per CU — at the same moment
  int stagger;
  // EPI_BIAS_GELU_DUAL: second output [m, N] of the operand type (gelu of the f32 rows written to Y);
  // EPI_GELU_GRAD: saved f32 pre-activation [m, N] whose gelu' multiplies the output
  // EPI_BIAS_GELU_DUAL16: aux16 = gelu of the 16-bit rows written to Y;  EPI_GELU_GRAD16: pre16 = saved 16-bit pre-activation [m, N]
  void* aux16;
  const float* aux32;
  const void* pre16;
  // > 0 (and col_group == 0): this launch covers only the row panels of split_panels(); a tail launch of the 128x128 kernel follows
  int split_cus;
  // Row-panel height (round 5): 0 = the kernel picks 256 or 192 rows per tile from the DEVICE token count (panel_rows()), 1 = always
  // 256, 2 = always 192 (A/B: MANNER_HIP_GEMM_PANEL=256|192).  x_rows = rows the X buffer holds (a multiple of 8): a 192-row tile's
  // last panel may reach past it, so its 8-row DMA pieces are clamped to the buffer (rows >= *m_total are computed and never stored).
  int panel_mode;
  int64_t x_rows;
  // Round 6: 1 = every XCD walks a CONTIGUOUS range of the tile order (tile_walk) instead of every 8th slot of each round
  int xcd_ranges;
  // EPI_NRES, in-launch finalize (nres_fan_in): != nullptr -> the workgroup that completes a row panel last reduces its `part` to
  // fin_mr[m] = {mean, rstd}; fin_arrive: one self-resetting arrival counter per row panel; fin_groups = N / 64
  float2* fin_mr;
  int32_t* fin_arrive;
  float fin_inv_h, fin_eps;
  int fin_groups;
};

// The tiles one workgroup of a persistent launch walks: {first, stride, end}.  Workgroups are dealt round-robin over the 8 XCDs
// (blockIdx & 7), each with its own 4 MiB L2.  Interleaved (ranges == 0, rounds 1-5): XCD x takes slots [32 x, 32 x + 32) of EVERY
// round of 256 tiles — the 32 tiles it runs together share their row panels, but from round to round it moves on across the whole
// tile order and meets every weight tile again after 7/8 of the launch's operands have passed through its L2: FFN1 fetched 4.7 MB of
// weights per XCD per round (8 x 12 x 4.7 = 453 MB of a 600 MB fetch for 105 MB of operands, profiles/r5_final/pmc_traffic.json).
// Ranges (== 1): XCD x owns tiles [T x / 8, T (x + 1) / 8) of the order and its 32 workgroups walk them in rounds; with the order
// column-group-major (col_group) the XCD then works on ONE group of weight tiles for the whole launch (or two, around a group boundary)
// and they stay in its L2.  The number of rounds is the same (ceil(ceil(T / 8) / 32) = ceil(T / 256)); results do not depend on it.
struct TileWalk { int first, stride, end; };
__device__ __forceinline__ TileWalk tile_walk(int G, int b, int valid_tiles, int ranges) {
  const int q8 = G >> 3, r8 = G & 7, xcd = b & 7;
  const int first_slot = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  if (!ranges) return TileWalk{first_slot + (b >> 3), G, valid_tiles};
  const int n_x = q8 + (xcd < r8 ? 1 : 0);             // workgroups on this XCD (>= 1: this one)
  const int lo = (int)((int64_t)valid_tiles * first_slot / G), hi = (int)((int64_t)valid_tiles * (first_slot + n_x) / G);
  return TileWalk{lo + (b >> 3), n_x, hi};
}

// Wave quantisation of a persistent launch (round 5).  tiles = ceil(M / rows) * n_tiles on `cus` workgroups last ceil(tiles / cus)
// whole tile times, so 61 row panels x N = 768 (183 tiles of 256 rows) are ONE round with 73 of 256 CUs idle — while the same rows
// as 82 panels of 192 rows are 246 tiles: one round of three-quarter tiles.  A row's result does not depend on the panel height (per
// output element the same sequence of matrix instructions over K), so the choice is free of numerical consequences: both bodies give
// the same BITS (tests/test_gpu_parity.py holds them to torch.equal).  A 192-row tile costs 0.80 - 0.85 of a 256-row tile (measured,
// tools/panel_probe.py: out-projection 38.1 -> 30.3 us, FFN2 95 -> 77 us, Q|K|V 76.4 -> 65.2 us at 15.6 k tokens; the weight half of its
// DMA and LDS traffic does not shrink); 192 wins when its rounds x 0.83 undercut the 256-row rounds by at least 3 % — in practice when
// both heights need the same number of rounds.
__host__ __device__ inline int panel_rows(int M, int n_tiles, int cus, int mode) {
  if (mode == 1) return 256;
  if (mode == 2) return 192;
  const int t256 = ((M + 255) / 256) * n_tiles, t192 = ((M + 191) / 192) * n_tiles;
  const int r256 = (t256 + cus - 1) / cus, r192 = (t192 + cus - 1) / cus;
  return r192 * 83 < r256 * 97 ? 192 : 256;
}

// The epilogue of one wave's 128(m) x 64(n) accumulator block acc[a][bb] (a = 16-column block 0..3, bb = 16-row block) of the tile
// (mt, nt): shared by the 8-wave kernel (ew = its wave: column group ew & 3, row half ew >> 2) and by the 4-wave kernel (round 6), whose
// waves own 128 x 128 and call it once per 64-column half with the virtual wave index 4 wm + 2 wn + half — the same expressions in the
// same order, hence the same bits.  slab / rslab1: two wave-private 4 KiB LDS slabs (rslab1: EPI_NRES only; ONE_SLAB: it has one).
template <typename TE, typename TOut, int EPI, int ABL, int MBT, bool ONE_SLAB = false>
__device__ __forceinline__ void x16_epilogue(f32x4 (&acc)[4][MBT], char* slab, char* rslab1, int lane, int ew, int mt, int nt, int M, int N,
                                             const float* __restrict__ bias, const TE* __restrict__ R, TOut* __restrict__ Y, const DlnAux& dln) {
  typedef typename E16<TE>::v8 e16x8;
  typedef typename E16<TE>::v4 e16x4;
  constexpr int TM = 32 * MBT, WROWS = TM / 2;
  constexpr int OUT_ROW = 64 * sizeof(TOut);
  constexpr int CHUNKS = OUT_ROW / 16;
  constexpr int OPC = 16 / sizeof(TOut);
  constexpr int ROWS_PER_INST = 64 / CHUNKS;
  if constexpr (EPI == EPI_NRES) {
    // ---- residual epilogue of the deferred-LayerNorm path (K4/K6 tails, modeling_bert.py:289-293, 347-351), in
    // the MFMA register layout (lane = token 16b + l15, features 16a + 4lq .. +3), in place on R == Y:
    //   v = acc + (bias + beta) + ((raw - mean) * rstd) * gamma        (f32; `bias` already holds bias + beta)
    //   per row {sum, sum of squares} over the wave's 64 columns -> part[m][4 nt + wn]
    //   v -> bf16 slab -> whole 128-byte row segments.
    int el = lane;
    asm volatile("" : "+v"(el));
    const int l15 = el & 15, lq = el >> 4, wn = ew & 3, wm = ew >> 2;
    const int nbase = nt * G_BN + wn * 64;
    const int mrow0 = mt * TM + wm * WROWS;
    if (ABL == 1) {                                  // lab: main loop only
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < MBT; ++bb) asm volatile("" ::"v"(acc[a][bb]));
    } else {
      f32x4 bv[4], g4[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        bv[a] = *reinterpret_cast<const f32x4*>(bias + nbase + 16 * a + 4 * lq);
        g4[a] = *reinterpret_cast<const f32x4*>(dln.vec + nbase + 16 * a + 4 * lq);
      }
      // The residual is fetched as whole 128-byte row segments (16 B per lane, the same coalesced shape as the
      // output stores) and turned into the MFMA layout through two wave-private 4 KiB LDS slabs: 8-byte loads
      // straight in the MFMA layout touch 16 rows per instruction and cost ~10 us per tile in the address path.
      char* rslab0 = slab;                                          // 32 rows each, XOR-swizzled 16-byte chunks (rslab1: the caller's second slab)
      const int rrow0 = el >> 3, rsl = el & 7;
#pragma unroll
      for (int hb = 0; hb < (MBT + 3) / 4; ++hb) {     // halves of 4 token blocks (64 rows = two slabs; a 192-row tile's
        const int nsb = MBT / 2 - 2 * hb >= 2 ? 2 : 1; // second half has one: 32 rows); keeping more loads in flight (all 16,
        const int nb4 = 2 * nsb;                       // or half 1 under half 0's arithmetic) measured 15 % slower per tile
        f32x4 rawres[2][4];
        float2 ms[4];
#pragma unroll
        for (int sb = 0; sb < nsb; ++sb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = 8 * q + rrow0;
            const int m = min(mrow0 + 64 * hb + 32 * sb + row, M - 1);
            rawres[sb][q] = ABL == 3 ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(R + (size_t)m * N + nbase + 8 * (rsl ^ (row & 7)));
          }
#pragma unroll
        for (int b4 = 0; b4 < nb4; ++b4) ms[b4] = dln.mr[min(mrow0 + 16 * (4 * hb + b4) + l15, M - 1)];
        e16x4 res[4][4];
        if constexpr (!ONE_SLAB) {                     // two slabs: both 32-row sub-blocks go through LDS at once
#pragma unroll
          for (int sb = 0; sb < nsb; ++sb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int row = 8 * q + rrow0;
              *reinterpret_cast<f32x4*>((sb ? rslab1 : rslab0) + row * 128 + (rsl << 4)) = rawres[sb][q];
            }
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int b4 = 0; b4 < nb4; ++b4) {
            const int row = 16 * (b4 & 1) + l15;
            const char* rs_ = (b4 >> 1) ? rslab1 : rslab0;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              const int c = (16 * a + 4 * lq) / 8;
              res[b4][a] = *reinterpret_cast<const e16x4*>(rs_ + row * 128 + ((c ^ (row & 7)) << 4) + 8 * (lq & 1));
            }
          }
          __builtin_amdgcn_wave_barrier();
        } else {                                       // one slab (the 8-wave register-staged kernel has 4 KiB per wave to spare): the
#pragma unroll                                         // sub-blocks take turns — a wave's LDS operations execute in program order
          for (int sb = 0; sb < nsb; ++sb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int row = 8 * q + rrow0;
              *reinterpret_cast<f32x4*>(rslab0 + row * 128 + (rsl << 4)) = rawres[sb][q];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
              const int row = 16 * b2 + l15;
#pragma unroll
              for (int a = 0; a < 4; ++a) {
                const int c = (16 * a + 4 * lq) / 8;
                res[2 * sb + b2][a] = *reinterpret_cast<const e16x4*>(rslab0 + row * 128 + ((c ^ (row & 7)) << 4) + 8 * (lq & 1));
              }
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
#pragma unroll
        for (int b4 = 0; b4 < nb4; ++b4) {
          const int bb = 4 * hb + b4;
          const float rs = ms[b4].y, nm = -ms[b4].x * ms[b4].y;
          float p1 = 0.f, p2 = 0.f;
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float r = fmaf((float)res[b4][a][e], rs, nm);
              const float x = fmaf(r, g4[a][e], acc[a][bb][e] + bv[a][e]);
              acc[a][bb][e] = x;
              p1 += x;
              p2 = fmaf(x, x, p2);
            }
          p1 += __shfl_xor(p1, 16, 64); p2 += __shfl_xor(p2, 16, 64);
          p1 += __shfl_xor(p1, 32, 64); p2 += __shfl_xor(p2, 32, 64);
          const int m = mrow0 + 16 * bb + l15;
          if (ABL == 4) asm volatile("" ::"v"(p1), "v"(p2));
          else if (ABL != 2 && lq == 0 && m < M)       // write-through (sc1): another XCD's workgroup may reduce them in this launch (nres_fan_in)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(&dln.part[(size_t)(4 * nt + wn) * dln.part_stride + m]),
                               __builtin_bit_cast(unsigned long long, float2{p1, p2}), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    if (ABL != 1) {
      const int row0 = el >> 3, sl = el & 7;
#pragma unroll
      for (int j = 0; j < MBT / 2; ++j) {
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
          const int row = 16 * b2 + l15;
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const int c = (16 * a + 4 * lq) / 8;
            store4<TE>(reinterpret_cast<TE*>(slab + row * 128 + ((c ^ (row & 7)) << 4) + 8 * (lq & 1)),
                           acc[a][2 * j + b2][0], acc[a][2 * j + b2][1], acc[a][2 * j + b2][2], acc[a][2 * j + b2][3]);
          }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = 8 * q + row0;
          const int c = sl ^ (row & 7);
          const int m = mrow0 + 32 * j + row;
          const f32x4 raw = *reinterpret_cast<const f32x4*>(slab + row * 128 + (sl << 4));
          if (m < M) *reinterpret_cast<f32x4*>(reinterpret_cast<TE*>(Y) + (size_t)m * N + nbase + 8 * c) = raw;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    return;
  }
  // ---- epilogue: 4 KiB slabs through a wave-private XOR-swizzled LDS slab, whole-row 16-byte stores.
  // The epilogue's lane-constant addressing is recomputed per tile from an opaque copy of the lane
  // id: hoisted out of the tile loop it would stay live across the K-loop and spill (the K-loop
  // already uses ~240 of the 256 registers).
  int el = lane;
  asm volatile("" : "+v"(el));
  const int l15 = el & 15, lq = el >> 4, wn = ew & 3, wm = ew >> 2;
  // slabs: 4 KiB per wave (the caller's `slab`: in the activation stage the last K-step has finished with); the
  // f32 output path therefore moves 16-token half slabs
  const int nbase = nt * G_BN + wn * 64;
  const int row0 = el / CHUNKS, sl = el % CHUNKS;
  f32x4 bv[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) bv[a] = *reinterpret_cast<const f32x4*>(bias + nbase + 16 * a + 4 * lq);
  constexpr bool NORM = EPI == EPI_NORM || EPI == EPI_NORM_GELU;   // deferred LayerNorm of the A operand
  constexpr bool SPLIT = EPI == EPI_BIAS_GELU_SPLIT3;              // 16-bit [hi | hi | lo] output rows of 3 N elements
  const size_t ldy = SPLIT ? (size_t)3 * N : (size_t)N;
  f32x4 cv[NORM ? 4 : 1];
  float2 ms[NORM ? MBT : 1];
  const int mrow0 = mt * TM + wm * WROWS;            // first row of this wave's tile
  if constexpr (NORM) {
#pragma unroll
    for (int a = 0; a < 4; ++a) cv[a] = *reinterpret_cast<const f32x4*>(dln.vec + nbase + 16 * a + 4 * lq);
#pragma unroll
    for (int bb = 0; bb < MBT; ++bb) ms[bb] = dln.mr[min(mrow0 + 16 * bb + l15, M - 1)];
  }
  constexpr int SLAB_ROWS = 4096 / OUT_ROW;          // 32 tokens (bf16) / 16 tokens (f32) per slab
  constexpr int MB = SLAB_ROWS / 16;                 // MFMA token blocks per slab
  constexpr int SQ = SLAB_ROWS / ROWS_PER_INST;      // row-contiguous 16-byte instructions per slab (4)
#pragma unroll
  for (int j = 0; j < WROWS / SLAB_ROWS; ++j) {
#pragma unroll
    for (int b2 = 0; b2 < MB; ++b2) {
      const int row = 16 * b2 + l15;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int nl = 16 * a + 4 * lq;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (NORM) v[e] = fmaf(ms[MB * j + b2].y, fmaf(-ms[MB * j + b2].x, cv[a][e], acc[a][MB * j + b2][e]), bv[a][e]);
          else v[e] = acc[a][MB * j + b2][e] + bv[a][e];
        }
        if (EPI == EPI_BIAS_GELU || EPI == EPI_NORM_GELU || SPLIT) {
#pragma unroll
          // f32 out: erff; split (x3 modes' FFN1): the |error| <= 1.5e-7 erf by exp / rcp (2 transcendentals + 6 FMAs against erff's
          // ~30-instruction expansion: this epilogue was 177 us of an 881 us launch) — two orders below the mode's measured 1e-5
          for (int e = 0; e < 4; ++e) v[e] = SPLIT ? gelu_erf_fast(v[e]) : sizeof(TOut) == 4 ? gelu_erf(v[e]) : gelu_poly(v[e]);
        }
        const int c = nl / OPC;
        const int off = row * OUT_ROW + ((c ^ (row & (CHUNKS - 1))) << 4) + (sizeof(TOut) == 2 ? 8 * (lq & 1) : 0);
        if constexpr (SPLIT) {
          // hi is converted ONCE and the remainder is taken against that very value (a second f32 -> 16-bit conversion of v may
          // be a different instruction — packed vs scalar — and round a tie the other way: hi + lo would then miss v by an ulp
          // of hi); the remainders go through the slab in a second pass
          typedef TOut o4 __attribute__((ext_vector_type(4)));
          o4 hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) hv[e] = (TOut)v[e];
          asm volatile("" : "+v"(hv));                 // opaque: the store and the remainder below use THESE bits
          *reinterpret_cast<o4*>(slab + off) = hv;
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[a][MB * j + b2][e] = v[e] - (float)hv[e];
        } else {
          store4<TOut>(reinterpret_cast<TOut*>(slab + off), v[0], v[1], v[2], v[3]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if constexpr (SPLIT) {
      // pass 1: the hi rows go out twice (columns n and N + n); pass 2: the lo rows (columns 2 N + n)
#pragma unroll
      for (int q = 0; q < SQ; ++q) {
        const int row = q * ROWS_PER_INST + row0;
        const int c = sl ^ (row & (CHUNKS - 1));
        const int m = mrow0 + SLAB_ROWS * j + row;
        const f32x4 raw = *reinterpret_cast<const f32x4*>(slab + row * OUT_ROW + (sl << 4));
        if (m < M) {
          TOut* dst = Y + (size_t)m * ldy + nbase + c * OPC;
          __builtin_nontemporal_store(raw, reinterpret_cast<f32x4*>(dst));
          __builtin_nontemporal_store(raw, reinterpret_cast<f32x4*>(dst + N));
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int b2 = 0; b2 < MB; ++b2) {
        const int row = 16 * b2 + l15;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int c = (16 * a + 4 * lq) / OPC;
          const int off = row * OUT_ROW + ((c ^ (row & (CHUNKS - 1))) << 4) + 8 * (lq & 1);
          store4<TOut>(reinterpret_cast<TOut*>(slab + off), acc[a][MB * j + b2][0], acc[a][MB * j + b2][1], acc[a][MB * j + b2][2],
                       acc[a][MB * j + b2][3]);
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int q = 0; q < SQ; ++q) {
        const int row = q * ROWS_PER_INST + row0;
        const int c = sl ^ (row & (CHUNKS - 1));
        const int m = mrow0 + SLAB_ROWS * j + row;
        const f32x4 raw = *reinterpret_cast<const f32x4*>(slab + row * OUT_ROW + (sl << 4));
        if (m < M) __builtin_nontemporal_store(raw, reinterpret_cast<f32x4*>(Y + (size_t)m * ldy + 2 * (size_t)N + nbase + c * OPC));
      }
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    e16x4 res[EPI == EPI_BIAS_RES ? SQ : 1];
    f32x4 resf[EPI == EPI_BIAS_RES_F32 ? SQ : 1];
    if ((EPI == EPI_BIAS_RES || EPI == EPI_BIAS_RES_F32) && ABL != 1) {   // TOut == float: the slab's residual loads in one batch
#pragma unroll
      for (int q = 0; q < SQ; ++q) {
        const int row = q * ROWS_PER_INST + row0;
        const int m = min(mrow0 + SLAB_ROWS * j + row, M - 1);
        const size_t idx = (size_t)m * N + nbase + (sl ^ (row & (CHUNKS - 1))) * OPC;
        if (EPI == EPI_BIAS_RES) res[q] = *reinterpret_cast<const e16x4*>(R + idx);
        else resf[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(R) + idx);
      }
    }
    f32x4 pre16v[EPI == EPI_GELU_GRAD16 ? SQ : 1];
    if constexpr (EPI == EPI_GELU_GRAD16) {                              // TOut == TE: 8 outputs per chunk, their 8 saved 16-bit pre-activations
#pragma unroll
      for (int q = 0; q < SQ; ++q) {
        const int row = q * ROWS_PER_INST + row0;
        const int m = min(mrow0 + SLAB_ROWS * j + row, M - 1);
        pre16v[q] = *reinterpret_cast<const f32x4*>(static_cast<const TE*>(dln.pre16) + (size_t)m * N + nbase + (sl ^ (row & (CHUNKS - 1))) * OPC);
      }
    }
    f32x4 pre[EPI == EPI_GELU_GRAD ? SQ : 1][2];
    if constexpr (EPI == EPI_GELU_GRAD) {                                // TOut == TE: 8 outputs per chunk, their 8 saved f32 pre-activations
#pragma unroll
      for (int q = 0; q < SQ; ++q) {
        const int row = q * ROWS_PER_INST + row0;
        const int m = min(mrow0 + SLAB_ROWS * j + row, M - 1);
        const f32x4* ip = reinterpret_cast<const f32x4*>(dln.aux32 + (size_t)m * N + nbase + (sl ^ (row & (CHUNKS - 1))) * OPC);
        pre[q][0] = ip[0];
        pre[q][1] = ip[1];
      }
    }
#pragma unroll
    for (int q = 0; q < SQ; ++q) {
      const int row = q * ROWS_PER_INST + row0;
      const int c = sl ^ (row & (CHUNKS - 1));
      const int m = mrow0 + SLAB_ROWS * j + row;
      f32x4 raw = *reinterpret_cast<const f32x4*>(slab + row * OUT_ROW + (sl << 4));
      if (ABL == 1) { asm volatile("" ::"v"(raw)); continue; }
      if constexpr (EPI == EPI_GELU_GRAD) {
        e16x8 d = __builtin_bit_cast(e16x8, raw);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = (TE)((float)d[e] * gelu_grad_fast(pre[q][e >> 2][e & 3]));
        raw = __builtin_bit_cast(f32x4, d);
      }
      if constexpr (EPI == EPI_GELU_GRAD16) {
        e16x8 d = __builtin_bit_cast(e16x8, raw);
        const e16x8 pv = __builtin_bit_cast(e16x8, pre16v[q]);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = (TE)((float)d[e] * gelu_grad_fast((float)pv[e]));
        raw = __builtin_bit_cast(f32x4, d);
      }
      if (EPI == EPI_BIAS_RES) {
#pragma unroll
        for (int e = 0; e < 4; ++e) raw[e] += (float)res[q][e];
      }
      if (EPI == EPI_BIAS_RES_F32) {
        if (dln.drop_thr != 0) {                                   // dropout on the GEMM output, then the residual (r = dropout(y) + x)
          const uint64_t di = (uint64_t)(dln.drop_rowmap ? dln.drop_rowmap[min(m, M - 1)] : m) * N + (nbase + c * OPC);
#pragma unroll
          for (int e = 0; e < 4; ++e) raw[e] = drop_bits(dln.drop_seed, dln.drop_site, di + e) >= dln.drop_thr ? raw[e] * dln.drop_scale : 0.f;
        }
        raw += resf[q];
      }
      if (m < M) {
        f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(Y) + ((size_t)m * N + nbase + c * OPC) * sizeof(TOut));
        // the big streaming outputs (Q|K|V 302 MB, FFN intermediate 403 MB per launch) are written non-temporally so
        // that they do not push the weight tiles, re-read by every row panel, out of the XCD's 4 MB L2
        if (NORM && ABL != 5 && !dln.plain_stores) __builtin_nontemporal_store(raw, dst);
        else *dst = raw;
        if constexpr (EPI == EPI_BIAS_GELU_DUAL) {                       // TOut == float: 4 pre-activations -> 4 gelu values, 8-byte store
          e16x4 gv;
#pragma unroll
          for (int e = 0; e < 4; ++e) gv[e] = (TE)gelu_erf_fast(raw[e]);
          *reinterpret_cast<e16x4*>(static_cast<TE*>(dln.aux16) + (size_t)m * N + nbase + c * OPC) = gv;
        }
        if constexpr (EPI == EPI_BIAS_GELU_DUAL16) {                     // TOut == TE: 8 rounded pre-activations -> their 8 gelu values
          const e16x8 pv = __builtin_bit_cast(e16x8, raw);
          e16x8 gv;
#pragma unroll
          for (int e = 0; e < 8; ++e) gv[e] = (TE)gelu_erf_fast((float)pv[e]);
          *reinterpret_cast<e16x8*>(static_cast<TE*>(dln.aux16) + (size_t)m * N + nbase + c * OPC) = gv;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}


// In-launch finalize of the deferred-LayerNorm row statistics (round 6; EPI_NRES with dln.fin_mr set).  Every tile of a row panel
// publishes per-64-column partial {sum, sum of squares} of its rows (`part`, written through: sc1); when a workgroup has finished a
// tile it drains its stores (every wave: s_waitcnt vmcnt(0)), meets at a barrier, and ONE lane adds 1 to the panel's arrival counter
// at agent scope.  The workgroup whose add returns n_tiles - 1 came last: every other tile's partials were drained before their
// workgroup's add, so its first wave reads them back (sc1 loads: past this CU's L1, which no other CU's store refreshes), reduces each
// row in the fixed group order of dln_finalize_kernel (dln_row_stats: the same bits), writes {mean, rstd} for the NEXT kernel and
// puts the counter back to 0 for the next launch.  Which workgroup reduces depends on timing; what it computes does not.  This is
// the fan-in form of cdna_hip_programming.md Guideline 16 (sc1 payload, drained stores, a workgroup barrier between every wave's
// drain and the one agent-scope add, sc1 loads by the last adder); nothing depends on workgroup -> XCD placement.
// Price and gain (docs/rounds/r6.md 1k): a launch is 6 us longer — the add's return from the memory side (~2 us) holds the first wave
// and, at the next tile's first barrier, the workgroup; signalling a tile's arrival later (where the next tile's K-loop has ended and
// its stores are complete by the in-order rule), issuing the add before the epilogue and reading it after, or leaving the tile's
// output stores in flight behind a counted wait all measured the same — more than the 5 us finalize kernel; but with two streams the
// tiny finalize launches sat in the queue behind the other stream's persistent GEMM and held their own stream's next GEMM back: the
// timed step is 0.5 - 1.3 % shorter without them.
template <int TM>
__device__ __forceinline__ void nres_fan_in(const DlnAux& dln, int mt, int M, int n_tiles, int wave, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's partial sums (and everything before them) have left the CU
  __syncthreads();                                     // ... every wave's
  if (wave != 0) return;
  int old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add(dln.fin_arrive + mt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  old = __builtin_amdgcn_readfirstlane(old);
  if (old != n_tiles - 1) return;
  // TM / 64 rows per lane, every partial of every row requested before the first is used: one memory round trip
  constexpr int RPL = TM / 64;
  float2 v[RPL][16];
#pragma unroll
  for (int q = 0; q < RPL; ++q) {
    const int m = min(mt * TM + 64 * q + lane, M - 1);
#pragma unroll
    for (int g = 0; g < 16; ++g)
      v[q][g] = g < dln.fin_groups ? __builtin_bit_cast(float2, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(&dln.part[(size_t)g * dln.part_stride + m]),
                                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                                   : float2{0.f, 0.f};
  }
#pragma unroll
  for (int q = 0; q < RPL; ++q) {
    const int m = mt * TM + 64 * q + lane;
    if (m < M) dln.fin_mr[m] = dln_row_stats(v[q], dln.fin_inv_h, dln.fin_eps);
  }
  if (lane == 0) __hip_atomic_store(dln.fin_arrive + mt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// MBT = 16-row MFMA token blocks per wave: 8 (256-row tile, wave tile 128 x 64) or 6 (192-row tile, wave tile 96 x 64).  The weight
// half of a tile (256 columns: 4 waves x 64) is the same for both; the activation stage keeps its 32 KiB stride and holds 24 KiB.
template <typename TE, typename TOut, int EPI, int ABL, int MBT>
__device__ __forceinline__ void gemm_tn_x16_body(
    char* lds, const TE* __restrict__ X, const TE* __restrict__ W, const float* __restrict__ bias,
    const TE* __restrict__ R, TOut* __restrict__ Y, int N, int K, int M, int n_tiles, const DlnAux& dln) {
  typedef TE TIn;                               // bf16_t or f16_t (E16<TE>: vector types and the MFMA of the type)
  typedef typename E16<TE>::v8 e16x8;
  constexpr int EPC = 8, BK = 64;
  constexpr int TM = 32 * MBT;                  // rows of a tile
  constexpr int HB = MBT / 2;                   // token blocks per m-half of a K-step chunk (and DMA piece pairs of an activation wave)
  constexpr int WROWS = TM / 2;                 // rows of a wave's tile
  static_assert(MBT == 8 || MBT == 6, "row panels of 256 or 192 rows");
  constexpr int XB = 2 * G_OP_BYTES;                                        // base of the activation ring

  const int G = gridDim.x, b = blockIdx.x;
  const int m_tiles = (MBT == 8 && dln.split_cus > 0) ? split_panels((M + TM - 1) / TM, n_tiles, dln.split_cus) : (M + TM - 1) / TM;
  const int valid_tiles = m_tiles * n_tiles;
  // Tile order: column tiles are walked in groups of `gsz`; within a group the order is row-panel-major.  One pass
  // over the rows then touches only gsz weight tiles (gsz * 256 * K * 2 bytes), which stay resident in the XCD's 4 MB
  // L2 while activations stream — at the price of reading the activations once per group instead of once.
  const int gsz = dln.col_group > 0 && dln.col_group < n_tiles ? dln.col_group : n_tiles;
  const int per_group = m_tiles * gsz;
  auto decode = [&](int tile, int& mt_, int& nt_) {
    const int g = tile / per_group;
    const int r = tile - g * per_group;
    const int width = min(gsz, n_tiles - g * gsz);     // the last group may be narrower
    mt_ = r / width;
    nt_ = g * gsz + (r - mt_ * width);
  };
  const TileWalk walk = tile_walk(G, b, valid_tiles, dln.xcd_ranges);
  int t = walk.first;
  if (t >= walk.end) return;
  if (dln.stagger > 0) {
    const int phase = (walk.first / n_tiles) & 3;
    for (int i = 0; i < phase * dln.stagger; ++i) __builtin_amdgcn_s_sleep(64);
  }

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int kl15 = lane & 15, klq = lane >> 4;        // K-loop copies (the epilogue recomputes its own)
  const int kwn = wave & 3, kwm = wave >> 2;

  // DMA: waves 0-3 bring the weight tile (8 pieces = 64 rows each), 4-7 the activation tile (MBT pieces = TM / 4 rows each)
  const bool is_w = wave < 4;
  const int prow0 = is_w ? 64 * (wave & 3) : (TM / 4) * (wave & 3);
  const int ldst0 = (is_w ? 0 : XB) + prow0 * ROW_BYTES;
  const int lrow = lane >> 3;
  int voff[2];
#pragma unroll
  for (int par = 0; par < 2; ++par)
    voff[par] = lrow * K + (((lane & 7) ^ ((4 * par + (lane >> 4)) & 7)) * EPC);
  // a tile operand as this wave sees it: first row of its pieces and — 192-row tiles only — the largest row offset at which an
  // 8-row piece still lies inside the X buffer (the last panel may reach past x_rows; W tiles and 256-row panels never do)
  struct Src { const TIn* p; int lim; };
  auto tile_src = [&](int tile) -> Src {
    int mt_, nt_;
    decode(tile, mt_, nt_);
    if (is_w) return Src{W + (size_t)(nt_ * G_BN + prow0) * K, 64};
    if constexpr (MBT == 8) {
      return Src{X + (size_t)(mt_ * TM + prow0) * K, 64};
    } else {
      const int last = (int)dln.x_rows - 8;
      const int r0 = min(mt_ * TM + prow0, last);
      return Src{X + (size_t)r0 * K, min(last - r0, 64)};
    }
  };
  auto issue2 = [&](int buf, const Src& gsrc, int k0, int pair) {
    char* base = lds + buf * G_OP_BYTES + ldst0;          // buf: weight stage 0..1 / activation stage 0..2
#pragma unroll
    for (int par = 0; par < 2; ++par) {
      const int i = 2 * pair + par;
      const int ro = MBT == 8 ? 8 * i : min(8 * i, gsrc.lim);
      if (ABL == 6 && !is_w)
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gsrc.p + (size_t)ro * K + k0 + voff[par]), LDS_PTR(base + i * 1024), 16, 0, 2);
      else
        __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gsrc.p + (size_t)ro * K + k0 + voff[par]),
                                         LDS_PTR(base + i * 1024), 16, 0, 0);
    }
  };

  const int swz = (kl15 >> 1) & 7;
  const int woff = (kwn * 64 + kl15) * ROW_BYTES;
  const int xoff = (kwm * WROWS + kl15) * ROW_BYTES;
  auto read_w = [&](const char* base, int s2, e16x8 (&wf)[4]) {
    const int coff = ((4 * s2 + klq) ^ swz) << 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) wf[a] = *reinterpret_cast<const e16x8*>(base + woff + a * 16 * ROW_BYTES + coff);
  };
  auto read_x = [&](const char* base, int s2, int half, e16x8 (&xf)[HB]) {
    const int coff = ((4 * s2 + klq) ^ swz) << 4;
#pragma unroll
    for (int bb = 0; bb < HB; ++bb)
      xf[bb] = *reinterpret_cast<const e16x8*>(base + xoff + (HB * half + bb) * 16 * ROW_BYTES + coff);
  };

  f32x4 acc[4][MBT];
  auto mma16 = [&](const e16x8 (&wf)[4], const e16x8 (&xf)[HB], int half) {
#pragma unroll
    for (int bb = 0; bb < HB; ++bb)
#pragma unroll
      for (int a = 0; a < 4; ++a)
        acc[a][HB * half + bb] = E16<TE>::mfma16(wf[a], xf[bb], acc[a][HB * half + bb]);
  };

  e16x8 w0[4], w1[4], xa[HB], xb[HB];
  // DMA issue is STAGGERED between the two waves of a SIMD (waves w and w+4): an LDS-DMA piece costs the
  // issuing wave ~60-100 cycles, 8 pieces per K-step; if both partners issued in the same chunk neither
  // could feed the matrix pipe meanwhile.  Waves 0-3 (weight tile) issue all 8 pieces of step k+2 in
  // chunk 3 (right after the barrier that frees stage `cur`), waves 4-7 (activation tile) all 8 pieces
  // of step k+1 in chunk 0; each half has >= 2.5 chunks to land before the vmcnt(0) of the sync.
  // MODE 0: steady state; 1: second-to-last step (chunk 3 prefetches K-step 0 of the NEXT tile instead of
  // step k+2); 2: last step (chunk 0 prefetches the activation half of the next tile's step 0).
  auto issue_all = [&](int buf, const Src& src, int k0) {
#pragma unroll
    for (int pair = 0; pair < 4; ++pair)
      if (MBT == 8 || pair < HB || is_w) issue2(buf, src, k0, pair);
  };
  // xs = activation stage of this K-step (ring of 3, continuous across tiles); its successor holds the
  // next step (already requested), the one after receives, in chunk 0, the step TWO ahead:
  // MODE 0: step k+2 of this tile; MODE 1 (k = nk-2): step 0 of the next tile; MODE 2 (k = nk-1): its step 1.
  auto kstep = [&](int cur, int xs, const Src& gsrc, int k1, int k2, const Src& gnext, bool has_next, auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    const char* wbase = lds + cur * G_OP_BYTES;
    const char* wnxt = lds + (cur ^ 1) * G_OP_BYTES;
    const int xs1 = xs == 2 ? 0 : xs + 1, xs2 = xs1 == 2 ? 0 : xs1 + 1;
    const char* xbase = lds + XB + xs * G_OP_BYTES;
    const char* xnxt = lds + XB + xs1 * G_OP_BYTES;
    if (ABL != 2 && !is_w) {                                   // waves 4-7: activations two steps ahead
      if (MODE == 0) issue_all(xs2, gsrc, k2);
      else if (has_next) issue_all(xs2, gnext, MODE == 1 ? 0 : BK);
    }
    // fragments are read one chunk ahead, each batch issued right AFTER the MFMA cluster of the current
    // chunk: the lgkmcnt(0) the compiler puts in front of a cluster then waits only for reads that are a
    // whole cluster old, never for reads issued a moment ago
    __builtin_amdgcn_sched_barrier(0);
    mma16(w0, xa, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_x(xbase, 0, 1, xb);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mma16(w0, xb, 1);
    __builtin_amdgcn_sched_barrier(0);
    read_w(wbase, 1, w1);
    read_x(xbase, 1, 0, xa);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mma16(w1, xa, 0);
    __builtin_amdgcn_sched_barrier(0);
    read_x(xbase, 1, 1, xb);
    __builtin_amdgcn_sched_barrier(0);
    // every read of this step's stages has completed; the NEXT step's tiles have landed: waves 0-3 wait
    // for all their DMA, waves 4-7 leave the 8 newest pieces (two steps ahead) in flight
    if (MODE <= 1) {
      // (the last tile's second-to-last step has no younger DMA behind the pieces the last step reads: wait for all)
      if (is_w || (MODE == 1 && !has_next)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else if constexpr (MBT == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");   // an activation wave's pieces per K-step
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (ABL != 2 && is_w) {                                    // waves 0-3: weights one step ahead of the sync
      if (MODE == 0) issue_all(cur, gsrc, k2);
      else if (MODE == 1 && has_next) issue_all(cur, gnext, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    mma16(w1, xb, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (MODE <= 1) { read_w(wnxt, 0, w0); read_x(xnxt, 0, 0, xa); }
    __builtin_amdgcn_sched_barrier(0);
  };
  typedef std::integral_constant<int, 0> Mode0;
  typedef std::integral_constant<int, 1> Mode1;
  typedef std::integral_constant<int, 2> Mode2;

  const int nk = K / BK;

  Src gsrc = tile_src(t);
  issue_all(0, gsrc, 0);             // K-step 0: weight stage 0 / activation stage 0
  if (!is_w) issue_all(1, gsrc, BK); // activations of K-step 1 (nk >= 2)
  int buf = 0;                       // weight stage that holds K-step 0 of the current tile
  int xs = 0;                        // activation stage of the current K-step (ring of 3 across tiles)
  bool first = true;
  while (true) {
    int mt, nt;
    decode(t, mt, nt);
    const int tn = t + walk.stride;
    const bool has_next = tn < walk.end;
    const Src gnext = tile_src(has_next ? tn : t);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int bb = 0; bb < MBT; ++bb) acc[a][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // tile prologue: K-step 0 has landed (issued one epilogue ago; the previous tile's stores may fly on)
    if (first) {
      if (is_w) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr (MBT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // step 1's activations may still fly
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (ABL == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (sizeof(TOut) == 2) {                       // every 16-bit epilogue ends with its 2 * MBT row stores
      if constexpr (MBT == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // residual variant: the last half slab's 4 stores
    }
    first = false;
#ifdef MANNER_W8_STAMPS   // lab build only: the same shader-clock split as gemm_tn_w8_kernel's
    const uint64_t xst0 = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();    // also: every wave has left the previous epilogue's LDS slabs
    read_w(lds + buf * G_OP_BYTES, 0, w0);
    read_x(lds + XB + xs * G_OP_BYTES, 0, 0, xa);
    if (ABL != 2 && is_w) issue_all(buf ^ 1, gsrc, BK);      // weight half of step 1 (nk >= 2, see launch_x16)
    int kt = 0;
    for (; kt + 2 < nk; ++kt) {
      kstep((buf + kt) & 1, xs, gsrc, (kt + 1) * BK, (kt + 2) * BK, gnext, has_next, Mode0{});
      xs = xs == 2 ? 0 : xs + 1;
    }
    kstep((buf + kt) & 1, xs, gsrc, (kt + 1) * BK, 0, gnext, has_next, Mode1{});
    xs = xs == 2 ? 0 : xs + 1;
    ++kt;
    const int last = (buf + kt) & 1;
    kstep(last, xs, gsrc, 0, 0, gnext, has_next, Mode2{});
    const int xfree = xs;              // activation stage of the last step: free for the epilogue slabs
    xs = xs == 2 ? 0 : xs + 1;
#ifdef MANNER_W8_STAMPS
    const uint64_t xst1 = __builtin_amdgcn_s_memtime();
#endif

    {
      int ew = wave;
      asm volatile("" : "+s"(ew));
      x16_epilogue<TE, TOut, EPI, ABL, MBT>(acc, lds + XB + xfree * G_OP_BYTES + ew * 4096, lds + last * G_OP_BYTES + ew * 4096, lane, ew, mt, nt, M, N,
                                             bias, R, Y, dln);
    }
    if constexpr (EPI == EPI_NRES && ABL == 0)
      if (dln.fin_mr) nres_fan_in<TM>(dln, mt, M, n_tiles, wave, lane);
#ifdef MANNER_W8_STAMPS
    if (dln.aux32) {
      const uint64_t xst2 = __builtin_amdgcn_s_memtime();
      uint64_t* dst = reinterpret_cast<uint64_t*>(const_cast<float*>(dln.aux32)) + ((size_t)blockIdx.x * 8 + wave) * 3;
      if (lane == 0) { dst[0] += xst1 - xst0; dst[1] += xst2 - xst1; dst[2] += 1; }
    }
#endif
    if (!has_next) break;
    t = tn;
    gsrc = gnext;
    buf = last ^ 1;
  }
}

// PANELS = true: the workgroup-uniform choice between the 256-row and the 192-row body (panel_rows() of the device token count);
// false (lab builds, ABL != 0): the 256-row body only.
template <typename TE, typename TOut, int EPI, int ABL = 0, bool PANELS = (ABL == 0)>
__global__ __launch_bounds__(512, 1) void gemm_tn_x16_kernel(
    const TE* __restrict__ X, const TE* __restrict__ W, const float* __restrict__ bias,
    const TE* __restrict__ R, TOut* __restrict__ Y, int N, int K, const int* __restrict__ m_total,
    int n_tiles, DlnAux dln) {
  // 2 weight stages [0, 64 KiB) + 3 activation stages [64, 160 KiB): the activation tile's first touch
  // comes from HBM (~2 us), so it is prefetched TWO K-steps ahead; weights are L2-resident (one ahead)
  __shared__ __attribute__((aligned(1024))) char lds[5 * G_OP_BYTES];       // 160 KiB
  if constexpr (EPI == EPI_BIAS) {
    X += (size_t)blockIdx.y * dln.batch_x;
    W += (size_t)blockIdx.y * dln.batch_w;
    Y += (size_t)blockIdx.y * dln.batch_y;
  }
  const int M = *m_total;
  if constexpr (PANELS) {
    if (panel_rows(M, n_tiles, (int)gridDim.x, dln.panel_mode) == 192) {
      gemm_tn_x16_body<TE, TOut, EPI, ABL, 6>(lds, X, W, bias, R, Y, N, K, M, n_tiles, dln);
      return;
    }
  }
  gemm_tn_x16_body<TE, TOut, EPI, ABL, 8>(lds, X, W, bias, R, Y, N, K, M, n_tiles, dln);
}

// ---------------------------------------------------------------------------------------------
// Round 6 — the production 16-bit GEMM with a HAND-SCHEDULED K-loop (gemm_tn_w8_kernel).  Geometry, tile walk, LDS image, the matrix
// instruction and its order over K per output element and the epilogues (x16_epilogue, called unchanged) are those of
// gemm_tn_x16_kernel, so the outputs are its bits (tests/test_gpu_parity.py holds the two to torch.equal).  What changes is the
// pipeline: operands are staged through registers (global_load_dwordx4 -> ds_write_b128) instead of LDS-DMA — the lab had priced
// the DMA's issue cost at a fifth of the main loop —, requested two K-steps ahead ACROSS tiles, with every s_waitcnt counted, one
// barrier per K-step and the fragment reads placed one per matrix-instruction gap; the K-loop of a tile is one inline-asm block
// with explicit registers (tools/gen_gemm_w.py -> gemm_w8_asm.inc), because the compiler's schedule of the same loop is what it
// replaces.  Same-box lab A/B of the main loops (tools/gemm4w_lab.hip, profiles/r6_final/lab_w4_w8_time.txt): Q|K|V 195 -> 152 us,
// out-projection 67 -> 52.5, FFN1 243 -> 201, FFN2 240 -> 216.  (The 4-wave / 128 x 128-wave-tile form below has the same main
// loop speed and loses it in the epilogue: one wave per SIMD issues vector instructions at half the rate of two.)
// State across tiles: the "+v" operands (piece offsets, LDS addresses) and LDS — stage P = K-step 0 of the tile, stage Q = K-step 1
// (staged by the previous tile's last steps / by the prologue here); 2 stages x [W 32 KiB | X 32 KiB] at 0 and 64 KiB (the stage is
// bit 16 of an LDS address), one 4 KiB epilogue slab per wave at 128 KiB.  256-row panels only (the host launches this kernel where
// panel_rows() picks 256; 192-row launches keep the LDS-DMA kernel).
template <typename TE, typename TOut, int EPI, int ABL = 0>
__global__ __launch_bounds__(512, 1) void gemm_tn_w8_kernel(
    const TE* __restrict__ X, const TE* __restrict__ W, const float* __restrict__ bias,
    const TE* __restrict__ R, TOut* __restrict__ Y, int N, int K, const int* __restrict__ m_total,
    int n_tiles, DlnAux dln) {
  typedef typename E16<TE>::v8 e16x8;
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  __shared__ __attribute__((aligned(1024))) char lds[5 * G_OP_BYTES];       // 160 KiB
  constexpr int BK = 64;
  if constexpr (EPI == EPI_BIAS) {
    X += (size_t)blockIdx.y * dln.batch_x;
    W += (size_t)blockIdx.y * dln.batch_w;
    Y += (size_t)blockIdx.y * dln.batch_y;
  }
  const int M = *m_total;
  const int G = gridDim.x, blk = blockIdx.x;
  const int m_tiles = (M + G_BM - 1) / G_BM;
  const int valid_tiles = m_tiles * n_tiles;
  const int gsz = dln.col_group > 0 && dln.col_group < n_tiles ? dln.col_group : n_tiles;      // the tile order of gemm_tn_x16_kernel
  const int per_group = m_tiles * gsz;
  auto decode = [&](int tile, int& mt_, int& nt_) {
    const int g = tile / per_group;
    const int r = tile - g * per_group;
    const int width = min(gsz, n_tiles - g * gsz);
    mt_ = r / width;
    nt_ = g * gsz + (r - mt_ * width);
  };
  const TileWalk walk = tile_walk(G, blk, valid_tiles, dln.xcd_ranges);
  int t = walk.first;
  if (t >= walk.end) return;

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wn = wave & 3, wm = wave >> 2;             // the wave's 128 x 64 block: columns 64 wn .., rows 128 wm ..
  const int op = wave >> 2, qf = wave & 3;             // staging role: waves 0..3 the weight tile's 64-row quarters, 4..7 the activation tile's
  const int l15 = lane & 15, lq = lane >> 4, lrow = lane >> 3, lc = lane & 7;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(lds);
  if (lds0 & 0x1ffffu) __builtin_trap();               // the stage bit is bit 16 of an absolute LDS address
  // fragment read addresses (the production LDS image: 128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7)
  const uint32_t swz = (l15 >> 1) & 7;
  uint32_t rw0 = lds0 + (64 * wn + l15) * ROW_BYTES + (((0 + lq) ^ swz) << 4);
  uint32_t rw1 = lds0 + (64 * wn + l15) * ROW_BYTES + (((4 + lq) ^ swz) << 4);
  uint32_t rx0 = lds0 + G_OP_BYTES + (128 * wm + l15) * ROW_BYTES + (((0 + lq) ^ swz) << 4);
  uint32_t rx1 = lds0 + G_OP_BYTES + (128 * wm + l15) * ROW_BYTES + (((4 + lq) ^ swz) << 4);
  // staging: piece p (8 rows x 128 B) of the wave's operand quarter goes to row 64 qf + 8 p + lrow, chunk lc ^ ((row >> 1) & 7)
  const uint32_t wrow = lds0 + op * G_OP_BYTES + (64 * qf + lrow) * ROW_BYTES;
  uint32_t wa0 = 0x10000u + wrow + ((lc ^ (((lrow >> 1)) & 7)) << 4);          // even pieces; -> stage Q (1)
  uint32_t wa1 = 0x10000u + wrow + ((lc ^ ((4 + (lrow >> 1)) & 7)) << 4);      // odd pieces
  // the same image by LDS-DMA (linear 1 KiB per piece): the lane that fills slot lc of its row fetches chunk lc ^ ((row >> 1) & 7)
  const uint32_t d0 = (uint32_t)(((lc ^ ((lrow >> 1) & 7)) - lc) * 16), d1 = (uint32_t)(((lc ^ ((4 + (lrow >> 1)) & 7)) - lc) * 16);
  const uint32_t rowb = (uint32_t)K * 2u;
  const uint32_t lane_off = (uint32_t)lrow * rowb + (uint32_t)lc * 16u;
  uint32_t g = lane_off;                               // the lane's byte offset of piece 0; piece p adds the wave-uniform p * 8 * rowb
  const uint32_t c1 = 8 * rowb, c2 = 16 * rowb, c3 = 24 * rowb, c4 = 32 * rowb, c5 = 40 * rowb, c6 = 48 * rowb, c7 = 56 * rowb;
  // this wave's operand quarter of a tile, as a wave-uniform byte pointer
  auto tile_base = [&](int tile) -> const char* {
    int mt_, nt_;
    decode(tile, mt_, nt_);
    const TE* p_ = op == 0 ? W + (size_t)(nt_ * G_BN + 64 * qf) * K : X + (size_t)(mt_ * G_BM + 64 * qf) * K;
    return reinterpret_cast<const char*>(p_);
  };
  // Where the next tile's K-step 1 waits out the epilogue: in 32 registers `sa` (requested by the tile's last step like any other
  // step's operands, written to LDS by the next tile's first step — nothing has to land before the epilogue may start) where the
  // epilogue leaves 32 registers (the deferred-LayerNorm consumers), else by LDS-DMA behind the tile's last barrier.
  constexpr bool AGPR = EPI == EPI_NORM || EPI == EPI_NORM_GELU;
  e16x8 sa[8];
  // ---- prologue (once per workgroup): K-step 0 -> stage 0, K-step 1 -> stage 1 (-> sa where the operands ride in registers, as every later tile finds them)
  {
    const char* cb = tile_base(t);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      e16x8 v[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) v[p] = *reinterpret_cast<const e16x8*>(cb + g + (uint32_t)p * 8u * rowb + kt * (BK * 2));
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        if (AGPR && kt == 1) sa[p] = v[p];
        else *reinterpret_cast<e16x8*>(lds + (((p & 1) ? wa1 : wa0) - lds0 - (kt == 0 ? 0x10000u : 0u)) + p * 1024) = v[p];
      }
    }
    g += 256;                                          // the load stream stands at K-step 2
    __syncthreads();
  }
  const int nk = K / BK;
#ifdef MANNER_W8_SETPRIO  // lab: static priority for the later-dispatched half (MI355X_MICROARCH.md, "Two waves per SIMD", item 4)
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef MANNER_W8_STAMPS
  uint64_t stamp_k = 0, stamp_e = 0, stamp_n = 0;
#endif
  char* slab = lds + 4 * G_OP_BYTES + wave * 4096;     // one wave-private 4 KiB epilogue slab in the 32 KiB behind the stages
  while (true) {
    int mt, nt;
    decode(t, mt, nt);
    const int tn = t + walk.stride;
    const bool has_next = tn < walk.end;
    // (readfirstlane: the asm block takes them as SGPR pairs — "s" — and the compiler must not doubt that they are wave-uniform)
    auto uniform = [](const char* p_) -> const char* {
      const uint64_t v = reinterpret_cast<uint64_t>(p_);
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
      return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    const char* cbase = uniform(tile_base(t));
    const char* nbase = uniform(tile_base(has_next ? tn : t));  // (no next tile: the last two steps re-read this one — legal, unused)
    int cnt = nk - 3;
    int m0_saved;                                      // (the LDS-DMA form saves and restores M0 around its pieces)
    (void)m0_saved;
    f32x16 o[8];
#ifdef MANNER_W8_STAMPS   // lab build only (tools/gemm4w_lab.hip): shader-clock stamps around the K-loop and the epilogue of every tile
    const uint64_t st0 = __builtin_amdgcn_s_memtime();
#endif
#define MANNER_W8_STATE [g] "+v"(g), [wa0] "+v"(wa0), [wa1] "+v"(wa1), [rw0] "+v"(rw0), [rw1] "+v"(rw1), [rx0] "+v"(rx0), [rx1] "+v"(rx1), [cnt] "+s"(cnt)
#define MANNER_W8_CONSTS                                                                                                                   \
  [base] "s"(cbase), [nbase] "s"(nbase), [rowb] "s"(rowb), [c1] "s"(c1), [c2] "s"(c2), [c3] "s"(c3), [c4] "s"(c4), [c5] "s"(c5),          \
      [c6] "s"(c6), [c7] "s"(c7)
#define MANNER_W8_RUN_TILE(MFMA_STR)                                                                                                       \
  asm volatile(MANNER_W8_TILE_ASM(MFMA_STR)                                                                                                \
               : MANNER_W8_ACC_OUTPUTS(o), MANNER_W8_STATE, [m0s] "=&s"(m0_saved)                                                          \
               : MANNER_W8_CONSTS, [d0] "v"(d0), [d1] "v"(d1)                                                                              \
               : MANNER_W8_CLOBBERS)
#define MANNER_W8_RUN_TILE_REGS(MFMA_STR)                                                                                                  \
  asm volatile(MANNER_W8_TILE_ASM_REGS(MFMA_STR)                                                                                           \
               : MANNER_W8_ACC_OUTPUTS(o), MANNER_W8_REGS_OPERANDS(sa), MANNER_W8_STATE                                                    \
               : MANNER_W8_CONSTS                                                                                                          \
               : MANNER_W8_CLOBBERS)
    if constexpr (AGPR) {
      if constexpr (E16<TE>::dtype == DT_BF16) {
        MANNER_W8_RUN_TILE_REGS("v_mfma_f32_16x16x32_bf16");
      } else {
        MANNER_W8_RUN_TILE_REGS("v_mfma_f32_16x16x32_f16");
      }
    } else {
      if constexpr (E16<TE>::dtype == DT_BF16) {
        MANNER_W8_RUN_TILE("v_mfma_f32_16x16x32_bf16");
      } else {
        MANNER_W8_RUN_TILE("v_mfma_f32_16x16x32_f16");
      }
    }
#undef MANNER_W8_RUN_TILE
#undef MANNER_W8_RUN_TILE_REGS
#undef MANNER_W8_STATE
#undef MANNER_W8_CONSTS
#ifdef MANNER_W8_STAMPS
    const uint64_t st1 = __builtin_amdgcn_s_memtime();
#endif
    {
      f32x4 acc[4][8];                                 // acc[a][b] = v[4 (8 a + b) ..] = o[2 a + (b >> 2)][4 (b & 3) ..]
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const f32x16& s16 = o[2 * a + (b >> 2)];
          const int k4 = 4 * (b & 3);
          acc[a][b] = f32x4{s16[k4], s16[k4 + 1], s16[k4 + 2], s16[k4 + 3]};
        }
      int ew = wave;
      asm volatile("" : "+s"(ew));
      x16_epilogue<TE, TOut, EPI, ABL, 8, true>(acc, slab, slab, lane, ew, mt, nt, M, N, bias, R, Y, dln);
    }
    if constexpr (EPI == EPI_NRES && ABL == 0)
      if (dln.fin_mr) nres_fan_in<G_BM>(dln, mt, M, n_tiles, wave, lane);
#ifdef MANNER_W8_STAMPS
    {
      const uint64_t st2 = __builtin_amdgcn_s_memtime();
      stamp_k += st1 - st0; stamp_e += st2 - st1; stamp_n += 1;
    }
#endif
    if (!has_next) break;
    t = tn;
  }
#ifdef MANNER_W8_STAMPS
  if (lane == 0 && dln.aux32) {                        // {cycles in K-loops, cycles in epilogues, tiles} per wave
    uint64_t* dst = reinterpret_cast<uint64_t*>(const_cast<float*>(dln.aux32)) + ((size_t)blk * 8 + wave) * 3;
    dst[0] = stamp_k; dst[1] = stamp_e; dst[2] = stamp_n;
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// Round 6 — the 4-wave form of the 16-bit production GEMM (the one work-removing design left).  Same 256 x 256
// tile, same persistent tile walk, same LDS image, the SAME matrix instruction over K in the same order per output element and the
// epilogues of the 8-wave kernel called unchanged (x16_epilogue, once per 64-column half of a wave's 128 x 128 block) — so the
// outputs are the bits of gemm_tn_x16_kernel (tests/test_gpu_parity.py holds the two to torch.equal), and what changes is the work
// per FLOP: four waves per CU (one per SIMD, 128 x 128 wave tiles, 256 accumulator registers in AGPRs) read 16 operand fragments
// per 64 MFMAs instead of 12 per 32 (-33 % LDS fragment traffic), synchronise four waves instead of eight per K-step, and stage
// operands through registers (global_load_dwordx4 -> ds_write_b128: with one wave per SIMD every LDS-DMA piece's 60-100 issue cycles
// would come out of the matrix pipe's issue time).  Lab A/B of the main loops on one box (tools/gemm4w_lab.hip,
// profiles/r6_final/lab4w_time.txt): Q|K|V 197 -> 160 us, out-projection 66 -> 54, FFN1 250 -> 212, FFN2 243 -> 228.
// The K-loop of a tile is ONE inline-asm block with explicit registers (tools/gen_gemm_w4.py -> gemm_w4_asm.inc: hipcc cannot
// allocate this kernel); its state across tiles lives in the operands below and in LDS:
//   stage P = K-step 0 of the tile, stage Q = K-step 1 (both staged by the previous tile's last steps / by the prologue here);
//   LDS: 2 stages x [W 32 KiB | X 32 KiB] at 0 and 64 KiB (the stage is bit 16 of an LDS address), epilogue slabs at 128 KiB.
// Loads run two K-steps ahead of their use ACROSS tiles (the last two steps of a tile fetch the next tile's first two), so a tile
// never opens with an exposed HBM round trip.  256-row panels only: the host launches this kernel where panel_rows() picks 256.
template <typename TE, typename TOut, int EPI>
__global__ __launch_bounds__(256, 1) void gemm_tn_w4_kernel(
    const TE* __restrict__ X, const TE* __restrict__ W, const float* __restrict__ bias,
    const TE* __restrict__ R, TOut* __restrict__ Y, int N, int K, const int* __restrict__ m_total,
    int n_tiles, DlnAux dln) {
  typedef typename E16<TE>::v8 e16x8;
  __shared__ __attribute__((aligned(1024))) char lds[5 * G_OP_BYTES];       // 160 KiB
  constexpr int BK = 64;
  const int M = *m_total;
  const int G = gridDim.x, blk = blockIdx.x;
  const int m_tiles = (M + G_BM - 1) / G_BM;
  const int valid_tiles = m_tiles * n_tiles;
  const int gsz = dln.col_group > 0 && dln.col_group < n_tiles ? dln.col_group : n_tiles;      // the 8-wave kernel's tile order
  const int per_group = m_tiles * gsz;
  auto decode = [&](int tile, int& mt_, int& nt_) {
    const int g = tile / per_group;
    const int r = tile - g * per_group;
    const int width = min(gsz, n_tiles - g * gsz);
    mt_ = r / width;
    nt_ = g * gsz + (r - mt_ * width);
  };
  const TileWalk walk = tile_walk(G, blk, valid_tiles, dln.xcd_ranges);
  int t = walk.first;
  if (t >= walk.end) return;

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wn = wave & 1, wm = wave >> 1;             // the wave's 128 x 128 block: columns 128 wn .., rows 128 wm ..
  const int op = wave >> 1, hf = wave & 1;             // staging role: waves 0 / 1 the weight tile's row halves, 2 / 3 the activation tile's
  const int l15 = lane & 15, lq = lane >> 4, lrow = lane >> 3, lc = lane & 7;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(lds);
  if (lds0 & 0x1ffffu) __builtin_trap();               // the stage bit is bit 16 of an absolute LDS address
  // fragment read addresses (the production LDS image: 128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7)
  const uint32_t swz = (l15 >> 1) & 7;
  uint32_t rw0 = lds0 + (128 * wn + l15) * ROW_BYTES + (((0 + lq) ^ swz) << 4);
  uint32_t rw1 = lds0 + (128 * wn + l15) * ROW_BYTES + (((4 + lq) ^ swz) << 4);
  uint32_t rx0 = lds0 + G_OP_BYTES + (128 * wm + l15) * ROW_BYTES + (((0 + lq) ^ swz) << 4);
  uint32_t rx1 = lds0 + G_OP_BYTES + (128 * wm + l15) * ROW_BYTES + (((4 + lq) ^ swz) << 4);
  // staging: piece p (8 rows x 128 B) of the wave's operand half goes to row 128 hf + 8 p + lrow, chunk lc ^ ((row >> 1) & 7)
  const uint32_t wrow = lds0 + op * G_OP_BYTES + (128 * hf + lrow) * ROW_BYTES;
  uint32_t wa0 = 0x10000u + wrow + ((lc ^ (((lrow >> 1)) & 7)) << 4);          // even pieces; -> stage Q (1)
  uint32_t wa1 = 0x10000u + wrow + ((lc ^ ((4 + (lrow >> 1)) & 7)) << 4);      // odd pieces
  // the same image by LDS-DMA (linear 1 KiB per piece): the lane that fills slot lc of its row fetches chunk lc ^ ((row >> 1) & 7)
  const uint32_t d0 = (uint32_t)(((lc ^ ((lrow >> 1) & 7)) - lc) * 16), d1 = (uint32_t)(((lc ^ ((4 + (lrow >> 1)) & 7)) - lc) * 16);
  const uint32_t rowb = (uint32_t)K * 2u;
  const uint32_t lane_off = (uint32_t)lrow * rowb + (uint32_t)lc * 16u;
  uint32_t g = lane_off;                               // the lane's byte offset of piece 0; piece p adds the wave-uniform p * 8 * rowb
  const uint32_t c1 = 8 * rowb, c2 = 16 * rowb, c3 = 24 * rowb, c4 = 32 * rowb, c5 = 40 * rowb, c6 = 48 * rowb, c7 = 56 * rowb, c8 = 64 * rowb,
                 c9 = 72 * rowb, c10 = 80 * rowb, c11 = 88 * rowb, c12 = 96 * rowb, c13 = 104 * rowb, c14 = 112 * rowb, c15 = 120 * rowb;
  // this wave's operand half of a tile, as a wave-uniform byte pointer
  auto tile_base = [&](int tile) -> const char* {
    int mt_, nt_;
    decode(tile, mt_, nt_);
    const TE* p_ = op == 0 ? W + (size_t)(nt_ * G_BN + 128 * hf) * K : X + (size_t)(mt_ * G_BM + 128 * hf) * K;
    return reinterpret_cast<const char*>(p_);
  };
  // ---- prologue (once per workgroup): K-step 0 -> stage 0, K-step 1 -> stage 1
  {
    const char* cb = tile_base(t);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      e16x8 v[16];
#pragma unroll
      for (int p = 0; p < 16; ++p) v[p] = *reinterpret_cast<const e16x8*>(cb + g + (uint32_t)p * 8u * rowb + kt * (BK * 2));
#pragma unroll
      for (int p = 0; p < 16; ++p)
        *reinterpret_cast<e16x8*>(lds + (((p & 1) ? wa1 : wa0) - lds0 - (kt == 0 ? 0x10000u : 0u)) + p * 1024) = v[p];
    }
    g += 256;                                          // the load stream stands at K-step 2
    __syncthreads();
  }
  const int nk = K / BK;
  char* slab = lds + 4 * G_OP_BYTES + wave * 8192;     // two wave-private 4 KiB epilogue slabs in the 32 KiB behind the stages
  while (true) {
    int mt, nt;
    decode(t, mt, nt);
    const int tn = t + walk.stride;
    const bool has_next = tn < walk.end;
    // (readfirstlane: the asm block takes them as SGPR pairs — "s" — and the compiler must not doubt that they are wave-uniform)
    auto uniform = [](const char* p_) -> const char* {
      const uint64_t v = reinterpret_cast<uint64_t>(p_);
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
      return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };
    const char* cbase = uniform(tile_base(t));
    const char* nbase = uniform(tile_base(has_next ? tn : t));  // (no next tile: the last two steps re-read this one — legal, unused)
    int cnt = nk - 3;
    int m0_saved;
#define MANNER_W4_RUN_TILE(MFMA_STR)                                                                                                       \
  asm volatile(MANNER_W4_TILE_ASM(MFMA_STR)                                                                                                \
               : [g] "+v"(g), [wa0] "+v"(wa0), [wa1] "+v"(wa1), [rw0] "+v"(rw0), [rw1] "+v"(rw1), [rx0] "+v"(rx0), [rx1] "+v"(rx1),        \
                 [cnt] "+s"(cnt), [m0s] "=&s"(m0_saved)                                                                                    \
               : [base] "s"(cbase), [nbase] "s"(nbase), [rowb] "s"(rowb), [d0] "v"(d0), [d1] "v"(d1), [c1] "s"(c1), [c2] "s"(c2),         \
                 [c3] "s"(c3), [c4] "s"(c4), [c5] "s"(c5), [c6] "s"(c6), [c7] "s"(c7), [c8] "s"(c8), [c9] "s"(c9), [c10] "s"(c10),         \
                 [c11] "s"(c11), [c12] "s"(c12), [c13] "s"(c13), [c14] "s"(c14), [c15] "s"(c15)                                            \
               : MANNER_W4_CLOBBERS)
    if constexpr (E16<TE>::dtype == DT_BF16) {
      MANNER_W4_RUN_TILE("v_mfma_f32_16x16x32_bf16");
    } else {
      MANNER_W4_RUN_TILE("v_mfma_f32_16x16x32_f16");
    }
#undef MANNER_W4_RUN_TILE
    // the accumulators sit in a[0:255]; the 8-wave kernel's epilogue runs on each 64-column half with the virtual wave index
    // 4 wm + (2 wn + half): column group (2 wn + half), row half wm — the same expressions, the same bits
    {
      f32x4 acc[4][8];
      MANNER_W4_READ_HALF0(acc)
      int ew = 4 * wm + 2 * wn;
      asm volatile("" : "+s"(ew));
      x16_epilogue<TE, TOut, EPI, 0, 8>(acc, slab, slab + 4096, lane, ew, mt, nt, M, N, bias, R, Y, dln);
    }
    {
      f32x4 acc[4][8];
      MANNER_W4_READ_HALF1(acc)
      int ew = 4 * wm + 2 * wn + 1;
      asm volatile("" : "+s"(ew));
      x16_epilogue<TE, TOut, EPI, 0, 8>(acc, slab, slab + 4096, lane, ew, mt, nt, M, N, bias, R, Y, dln);
    }
    if (!has_next) break;
    t = tn;
  }
}

#ifdef MANNER_P4_LAB
// ---------------------------------------------------------------------------------------------
// Round 6 — the PAIRED 4-wave form (tools/gen_gemm_p4.py -> gemm_p4_asm.inc).  LAB ONLY (MANNER_P4_LAB: tools/gemm_p4_lab.hip; not in
// libmanner_hip.so): bit-identical to the 8-wave kernel and 13 - 23 % SLOWER (profiles/r6_final/lab_p4_time.txt) — one 4-wave workgroup
// alone on a CU already runs at 0.95 of the pair's rate (lab_p4_single_vs_pair.txt): the second workgroup finds almost nothing to
// fill, and starting it late changes nothing (lab_p4_dephase.txt).  The idea was: a CU holds TWO independent workgroups of four waves
// (one per SIMD each: still two waves per SIMD, so the epilogues keep their VALU issue rate), each walking its own 256 x 128 tiles:
// one workgroup's barriers, LDS write phase and epilogue are the other's matrix time.  Wave tile 128 x 64 and accumulators v[0:127] as
// in the 8-wave kernel — the same fragments and the same matrix instruction per output element in the same order over K, and the same
// x16_epilogue (virtual wave index 4 wm + 2 (column half of the 256-column tile) + wn): the same BITS.  LDS per workgroup: ONE stage
// [W 128 rows | X 256 rows] x 128 B = 48 KiB + a 4 KiB epilogue slab per wave = 64 KiB.  Register staging, 12 pieces per wave and K-step
// (weight rows 32 w.., activation rows 64 w..), requested one K-step before they are written; nothing is in flight across the epilogue.
// dln.stagger (here: dephase) > 0: the workgroups in the second slot of their CU start that many ~4 us sleeps late, so that the two
// workgroups of a CU do not reach their epilogues together.
template <typename TE, typename TOut, int EPI, int ABL = 0>
__global__ __launch_bounds__(256, 2) void gemm_tn_p4_kernel(
    const TE* __restrict__ X, const TE* __restrict__ W, const float* __restrict__ bias,
    const TE* __restrict__ R, TOut* __restrict__ Y, int N, int K, const int* __restrict__ m_total,
    int n_tiles, DlnAux dln) {
  typedef typename E16<TE>::v8 e16x8;
  constexpr int P4_W = 0, P4_X = 16384, P4_SLAB = 49152;
  __shared__ __attribute__((aligned(1024))) char lds[65536];
  constexpr int BK = 64;
  const int M = *m_total;
  const int G = gridDim.x, blk = blockIdx.x;
  const int m_tiles = (M + G_BM - 1) / G_BM;
  const int n_half = 2 * n_tiles;                       // 128-column tiles
  const int valid_tiles = m_tiles * n_half;
  const int gsz = dln.col_group > 0 && dln.col_group < n_tiles ? 2 * dln.col_group : n_half;      // the 8-wave kernel's tile order, in halves
  const int per_group = m_tiles * gsz;
  auto decode = [&](int tile, int& mt_, int& nh_) {
    const int g_ = tile / per_group;
    const int r = tile - g_ * per_group;
    const int width = min(gsz, n_half - g_ * gsz);
    mt_ = r / width;
    nh_ = g_ * gsz + (r - mt_ * width);
  };
  const TileWalk walk = tile_walk(G, blk, valid_tiles, dln.xcd_ranges);
  int t = walk.first;
  if (t >= walk.end) return;

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wn = wave & 1, wm = wave >> 1;
  const int l15 = lane & 15, lq = lane >> 4, lrow = lane >> 3, lc = lane & 7;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(lds);
  const uint32_t swz = (l15 >> 1) & 7;
  const uint32_t rw0 = lds0 + P4_W + (64 * wn + l15) * ROW_BYTES + (((0 + lq) ^ swz) << 4);
  const uint32_t rw1 = lds0 + P4_W + (64 * wn + l15) * ROW_BYTES + (((4 + lq) ^ swz) << 4);
  const uint32_t rx0 = lds0 + P4_X + (128 * wm + l15) * ROW_BYTES + (((0 + lq) ^ swz) << 4);
  const uint32_t rx1 = lds0 + P4_X + (128 * wm + l15) * ROW_BYTES + (((4 + lq) ^ swz) << 4);
  // staging: piece p (8 rows x 128 B) of the wave's share goes to row r = share + 8 p + lrow, chunk lc ^ ((r >> 1) & 7)
  const uint32_t sw0 = (lc ^ ((lrow >> 1) & 7)) << 4, sw1 = (lc ^ ((4 + (lrow >> 1)) & 7)) << 4;
  const uint32_t ww0 = lds0 + P4_W + (32 * wave + lrow) * ROW_BYTES + sw0, ww1 = lds0 + P4_W + (32 * wave + lrow) * ROW_BYTES + sw1;
  const uint32_t xw0 = lds0 + P4_X + (64 * wave + lrow) * ROW_BYTES + sw0, xw1 = lds0 + P4_X + (64 * wave + lrow) * ROW_BYTES + sw1;
  const uint32_t rowb = (uint32_t)K * 2u;
  const uint32_t lane_off = (uint32_t)lrow * rowb + (uint32_t)lc * 16u;
  uint32_t g = lane_off;
  const uint32_t c1 = 8 * rowb, c2 = 16 * rowb, c3 = 24 * rowb, c4 = 32 * rowb, c5 = 40 * rowb, c6 = 48 * rowb, c7 = 56 * rowb;
  auto uniform = [](const char* p_) -> const char* {
    const uint64_t v = reinterpret_cast<uint64_t>(p_);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
  };
  auto w_share = [&](int tile) -> const char* {
    int mt_, nh_;
    decode(tile, mt_, nh_);
    return reinterpret_cast<const char*>(W + (size_t)(nh_ * 128 + 32 * wave) * K);
  };
  auto x_share = [&](int tile) -> const char* {
    int mt_, nh_;
    decode(tile, mt_, nh_);
    return reinterpret_cast<const char*>(X + (size_t)(mt_ * G_BM + 64 * wave) * K);
  };
  if ((dln.stagger & 0xff) > 0) {                      // dephase: the second workgroup of a CU starts late (lab: which bit tells them apart)
    const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, all 32 bits
    const int mode = dln.stagger >> 8;
    const bool second = mode == 0 ? (hw >> 16) & 1 : mode == 1 ? hw & 1 : mode == 2 ? (blk >> 3) & 1 : (blk >> 3) >= (G >> 4);
    if (second)
      for (int i = 0; i < (dln.stagger & 0xff); ++i) __builtin_amdgcn_s_sleep(127);
  }
  // ---- prologue (once per workgroup): K-step 0 of the first tile -> the stage
  {
    const char* wb = w_share(t);
    const char* xb = x_share(t);
    e16x8 v[12];
#pragma unroll
    for (int p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const e16x8*>(wb + g + (uint32_t)p * 8u * rowb);
#pragma unroll
    for (int p = 0; p < 8; ++p) v[4 + p] = *reinterpret_cast<const e16x8*>(xb + g + (uint32_t)p * 8u * rowb);
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<e16x8*>(lds + (((p & 1) ? ww1 : ww0) - lds0) + p * 1024) = v[p];
#pragma unroll
    for (int p = 0; p < 8; ++p) *reinterpret_cast<e16x8*>(lds + (((p & 1) ? xw1 : xw0) - lds0) + p * 1024) = v[4 + p];
    g += 128;                                          // the load stream stands at K-step 1
    __syncthreads();
  }
  const int nk = K / BK;
  char* slab = lds + P4_SLAB + wave * 4096;
  while (true) {
    int mt, nh;
    decode(t, mt, nh);
    const int tn = t + walk.stride;
    const bool has_next = tn < walk.end;
    const char* cbw = uniform(w_share(t));
    const char* cbx = uniform(x_share(t));
    const char* nbw = uniform(w_share(has_next ? tn : t));      // (no next tile: the last steps re-read this one — legal, unused)
    const char* nbx = uniform(x_share(has_next ? tn : t));
    int cnt = nk - 3;
    f32x16 o[8];
#define MANNER_P4_RUN_TILE(MFMA_STR)                                                                                                       \
  asm volatile(MANNER_P4_TILE_ASM(MFMA_STR)                                                                                                \
               : MANNER_P4_ACC_OUTPUTS(o), [g] "+v"(g), [cnt] "+s"(cnt)                                                                    \
               : [ww0] "v"(ww0), [ww1] "v"(ww1), [xw0] "v"(xw0), [xw1] "v"(xw1), [rw0] "v"(rw0), [rw1] "v"(rw1), [rx0] "v"(rx0),            \
                 [rx1] "v"(rx1), [basew] "s"(cbw), [basex] "s"(cbx), [nbasew] "s"(nbw), [nbasex] "s"(nbx), [rowb] "s"(rowb), [c1] "s"(c1), \
                 [c2] "s"(c2), [c3] "s"(c3), [c4] "s"(c4), [c5] "s"(c5), [c6] "s"(c6), [c7] "s"(c7)                                        \
               : MANNER_P4_CLOBBERS)
    if constexpr (E16<TE>::dtype == DT_BF16) {
      MANNER_P4_RUN_TILE("v_mfma_f32_16x16x32_bf16");
    } else {
      MANNER_P4_RUN_TILE("v_mfma_f32_16x16x32_f16");
    }
#undef MANNER_P4_RUN_TILE
    {
      f32x4 acc[4][8];                                 // acc[a][b] = v[4 (8 a + b) ..] = o[2 a + (b >> 2)][4 (b & 3) ..]
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const f32x16& s16 = o[2 * a + (b >> 2)];
          const int k4 = 4 * (b & 3);
          acc[a][b] = f32x4{s16[k4], s16[k4 + 1], s16[k4 + 2], s16[k4 + 3]};
        }
      int ew = 4 * wm + 2 * (nh & 1) + wn;             // the 8-wave kernel's wave that owns this 128 x 64 block of the 256 x 256 tile
      asm volatile("" : "+s"(ew));
      x16_epilogue<TE, TOut, EPI, ABL, 8, true>(acc, slab, slab, lane, ew, mt, nh >> 1, M, N, bias, R, Y, dln);
    }
    if (!has_next) break;
    t = tn;
  }
}
#endif  // MANNER_P4_LAB

// Persistent grid of an x16 launch and the panel fields of its DlnAux: one workgroup per CU (160 KiB LDS each), fewer only when even
// the finer (192-row) tiling of the host's row bound has fewer tiles.  MANNER_HIP_GEMM_PANEL=256|192 pins the panel height (A/B; read
// per launch: the tests flip it).
static int x16_grid(int64_t m_bound, int n_tiles, DlnAux& aux, dim3& g) {
  const int64_t tiles = ((m_bound + 191) / 192) * n_tiles;
  if (tiles <= 0 || tiles > 0x7fffffff) return fail(MANNER_HIP_E_INVALID, "gemm grid %lld out of range", (long long)tiles);
  const int n_cus = device_cus();
  g = dim3((unsigned)(tiles < n_cus ? tiles : n_cus));
  const char* e = getenv("MANNER_HIP_GEMM_PANEL");
  aux.panel_mode = e ? (atoi(e) == 256 ? 1 : atoi(e) == 192 ? 2 : 0) : 0;
  aux.x_rows = m_bound;
  return MANNER_HIP_OK;
}

// Which kernel takes a 16-bit persistent launch (round 6): 8 = gemm_tn_w8_kernel (hand-scheduled K-loop, the default), 4 =
// gemm_tn_w4_kernel (A/B, deferred-LayerNorm launches only), 0 = gemm_tn_x16_kernel.  The asm kernels run 256-row panels: they take a
// launch where panel_rows() picks 256 for the best estimate of the token count the host has.  MANNER_HIP_GEMM_ASM is read per launch
// (the equality tests flip it).
static int pick_asm(int64_t m_est, int n_tiles, int K, const DlnAux& aux) {
  const char* e = getenv("MANNER_HIP_GEMM_ASM");
  const int req = e ? atoi(e) : 8;
  if (req != 8 && req != 4) return 0;
  if (aux.stagger != 0 || aux.split_cus > 0 || K % 64 || K < 192 || aux.panel_mode == 2) return 0;
  return panel_rows((int)m_est, n_tiles, device_cus(), aux.panel_mode) == 256 ? req : 0;
}

// Column groups (round 6, with the per-XCD tile ranges of tile_walk): all column tiles in one pass read the activations once but, when
// their weights (N K 2 bytes) exceed what an XCD's 4 MiB L2 keeps beside the streaming operands (~2.5 MiB), every XCD re-fetches
// them every round: 8 (rounds - 1) N K 2 bytes.  g passes over the rows in groups of ceil(n_tiles / g) column tiles keep a group
// resident and read the activations g times: (g - 1) M K 2 bytes more.  Take the cheaper by this count: FFN1 (12 tiles, 4.7 MB) and
// Q|K|V (9 tiles, 3.5 MB) walk two groups (+100 MB of activations for -430 / -230 MB of weights at 65 536 tokens), FFN2 (4.7 MB in
// 3 tiles of K = 3072, activations 400 MB) and the out-projection (1.2 MB) one.  Returns the group width (0 = all tiles).
static int pick_col_group(int64_t m_est, int N, int K, int cus) {
  const int n_tiles = N / G_BN;
  const double w_bytes = 2.0 * N * K, x_bytes = 2.0 * (double)m_est * K, l2_keep = 2.5 * 1048576.0;
  const int64_t rounds = ((m_est + G_BM - 1) / G_BM * n_tiles + cus - 1) / cus;
  if (w_bytes <= l2_keep) return 0;
  const int groups = (int)((w_bytes + l2_keep - 1) / l2_keep);
  return (groups - 1) * x_bytes < 8.0 * (double)(rounds - 1) * w_bytes ? (n_tiles + groups - 1) / groups : 0;
}

// The tile order of a persistent launch from the two switches (read per launch: the equality tests flip them)
static void set_tile_order(DlnAux& aux, int64_t m_est, int N, int K) {
  const char* cg_env = getenv("MANNER_HIP_COL_GROUP");
  const char* xr_env = getenv("MANNER_HIP_XCD_RANGES");
  aux.xcd_ranges = xr_env ? (atoi(xr_env) != 0) : 1;
  aux.col_group = cg_env ? atoi(cg_env) : aux.xcd_ranges ? pick_col_group(m_est, N, K, device_cus()) : 0;
}

template <typename TE, typename TOut>
int launch_x16(Epilogue epi, const void* X, const void* W, const float* bias, const void* R, void* Y,
               int64_t m_bound, int N, int K, const int* m_total, hipStream_t stream) {
  const int n_tiles = N / G_BN;
  DlnAux aux0{};
  set_tile_order(aux0, m_bound, N, K);
  // measured (tools/tile_order_probe.py, f16x3, 65 k tokens): every x3 GEMM is indifferent to the order (+-0.3 %) except FFN1, whose
  // 1.2 GB of split output rows and 302 MB operand run 3 % slower in ranges (877 -> 904 us); it keeps the interleaved order
  if (epi == EPI_BIAS_GELU_SPLIT3 && !getenv("MANNER_HIP_XCD_RANGES")) aux0.xcd_ranges = aux0.col_group = 0;
  dim3 g, b(512);
  if (int rc = x16_grid(m_bound, n_tiles, aux0, g)) return rc;
  const TE* x = static_cast<const TE*>(X);
  const TE* w = static_cast<const TE*>(W);
  const TE* r = static_cast<const TE*>(R);
  TOut* y = static_cast<TOut*>(Y);
  if (pick_asm(m_bound, n_tiles, K, aux0) == 8) {      // the same bits from the hand-scheduled K-loop
    const int64_t tiles = (m_bound / G_BM) * n_tiles;
    const int64_t cus = device_cus();
    g = dim3((unsigned)(tiles < cus ? tiles : cus));
    switch (epi) {
      case EPI_BIAS:
        hipLaunchKernelGGL((gemm_tn_w8_kernel<TE, TOut, EPI_BIAS>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles, aux0);
        break;
      case EPI_BIAS_GELU:
        hipLaunchKernelGGL((gemm_tn_w8_kernel<TE, TOut, EPI_BIAS_GELU>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles, aux0);
        break;
      case EPI_BIAS_RES:
        hipLaunchKernelGGL((gemm_tn_w8_kernel<TE, TOut, EPI_BIAS_RES>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles, aux0);
        break;
      case EPI_BIAS_GELU_SPLIT3:
        if constexpr (sizeof(TOut) == 2) {
          hipLaunchKernelGGL((gemm_tn_w8_kernel<TE, TOut, EPI_BIAS_GELU_SPLIT3>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles, aux0);
          break;
        }
        return fail(MANNER_HIP_E_INVALID, "EPI_BIAS_GELU_SPLIT3 writes the 16-bit split operand");
      case EPI_BIAS_RES_F32:
        if constexpr (sizeof(TOut) == 4) {
          hipLaunchKernelGGL((gemm_tn_w8_kernel<TE, TOut, EPI_BIAS_RES_F32>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles, aux0);
          break;
        }
        return fail(MANNER_HIP_E_INVALID, "EPI_BIAS_RES_F32 writes f32");
      default:
        return fail(MANNER_HIP_E_INVALID, "gemm epilogue %d has its own entry point", (int)epi);
    }
    MANNER_LAUNCH_CHECK();
    return MANNER_HIP_OK;
  }
  switch (epi) {
    case EPI_BIAS:
      hipLaunchKernelGGL((gemm_tn_x16_kernel<TE, TOut, EPI_BIAS>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles, aux0);
      break;
    case EPI_BIAS_GELU:
      hipLaunchKernelGGL((gemm_tn_x16_kernel<TE, TOut, EPI_BIAS_GELU>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles, aux0);
      break;
    case EPI_BIAS_RES:
      hipLaunchKernelGGL((gemm_tn_x16_kernel<TE, TOut, EPI_BIAS_RES>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles, aux0);
      break;
    case EPI_BIAS_GELU_SPLIT3:
      if constexpr (sizeof(TOut) == 2) {
        hipLaunchKernelGGL((gemm_tn_x16_kernel<TE, TOut, EPI_BIAS_GELU_SPLIT3>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles, aux0);
        break;
      }
      return fail(MANNER_HIP_E_INVALID, "EPI_BIAS_GELU_SPLIT3 writes the 16-bit split operand");
    case EPI_BIAS_RES_F32:
      if constexpr (sizeof(TOut) == 4) {
        hipLaunchKernelGGL((gemm_tn_x16_kernel<TE, TOut, EPI_BIAS_RES_F32>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles, aux0);
        break;
      }
      return fail(MANNER_HIP_E_INVALID, "EPI_BIAS_RES_F32 writes f32");
    default:
      return fail(MANNER_HIP_E_INVALID, "gemm epilogue %d has its own entry point", (int)epi);
  }
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

template <typename TIn, typename TOut>
int launch_big(Epilogue epi, const void* X, const void* W, const float* bias, const void* R, void* Y,
               int64_t m_bound, int N, int K, const int* m_total, hipStream_t stream) {
  const int n_tiles = N / G_BN;
  const int64_t grid = (m_bound / G_BM) * n_tiles;
  if (grid <= 0 || grid > 0x7fffffff) return fail(MANNER_HIP_E_INVALID, "gemm grid %lld out of range", (long long)grid);
  dim3 g((unsigned)grid), b(512);
  const TIn* x = static_cast<const TIn*>(X);
  const TIn* w = static_cast<const TIn*>(W);
  const TIn* r = static_cast<const TIn*>(R);
  TOut* y = static_cast<TOut*>(Y);
  switch (epi) {
    case EPI_BIAS:
      hipLaunchKernelGGL((gemm_tn_big_kernel<TIn, TOut, EPI_BIAS>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles);
      break;
    case EPI_BIAS_GELU:
      hipLaunchKernelGGL((gemm_tn_big_kernel<TIn, TOut, EPI_BIAS_GELU>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles);
      break;
    case EPI_BIAS_RES:
      hipLaunchKernelGGL((gemm_tn_big_kernel<TIn, TOut, EPI_BIAS_RES>), g, b, 0, stream, x, w, bias, r, y, N, K, m_total, n_tiles);
      break;
    default:
      return fail(MANNER_HIP_E_INVALID, "gemm epilogue %d has its own entry point", (int)epi);
  }
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

}  // namespace

int device_cus() {
  static int cus[MAX_DEVICES] = {};
  int& n_cus = cus[current_device_slot()];
  if (!n_cus) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 256;
    n_cus = v > 0 ? v : 256;
    // A/B switch: persistent GEMM grids of fewer workgroups than CUs, so that two streams' GEMMs run side by side on
    // disjoint CUs (measured: -3 % — kept for re-measurement only)
    if (const char* ev = getenv("MANNER_HIP_GEMM_CUS")) { const int c = atoi(ev); if (c > 0 && c < n_cus) n_cus = c; }
  }
  return n_cus;
}

int pool_logits_mfma(const float* x, const float* W, const float* bias, const float* query, int64_t R, int D, int Q,
                     float* logits, hipStream_t stream) {
  if (R <= 0) return MANNER_HIP_OK;
  if (D % 32 || D < 32 || Q < 1 || Q > G_BN || (uintptr_t)x % 16 || (uintptr_t)W % 16)
    return fail(MANNER_HIP_E_INVALID, "pool_logits_mfma: D=%d Q=%d unsupported (D %% 32 == 0, Q <= 256, 16-byte aligned rows)", D, Q);
  // 32-bit element offsets inside the kernel: rows are processed in slices of fewer than 2^31 / D
  const int64_t slice = ((int64_t)0x7fffffff / D) / G_BM * G_BM;
  for (int64_t r0 = 0; r0 < R; r0 += slice) {
    const int64_t rows = R - r0 < slice ? R - r0 : slice;
    hipLaunchKernelGGL(pool_logits_kernel, dim3((unsigned)((rows + G_BM - 1) / G_BM)), dim3(512), 0, stream, x + r0 * D, W, bias,
                       query, rows, D, Q, logits + r0);
    MANNER_LAUNCH_CHECK();
  }
  return MANNER_HIP_OK;
}

template <typename TE>
static int launch_dln(Epilogue epi, const void* X, const void* W, const float* bias, void* Y, int N, int K, const int* m_total,
                      int n_tiles, dim3 g, const DlnAux& aux, hipStream_t stream, int asm_mode) {
  const dim3 b(512);
  const TE* x = static_cast<const TE*>(X);
  const TE* w = static_cast<const TE*>(W);
  TE* y = static_cast<TE*>(Y);
  if (asm_mode == 8) {                                 // round 6: the hand-scheduled 8-wave kernel (same bits; 256-row panels)
    switch (epi) {
      case EPI_NORM:
        hipLaunchKernelGGL((gemm_tn_w8_kernel<TE, TE, EPI_NORM>), g, b, 0, stream, x, w, bias, nullptr, y, N, K, m_total, n_tiles, aux);
        break;
      case EPI_NORM_GELU:
        hipLaunchKernelGGL((gemm_tn_w8_kernel<TE, TE, EPI_NORM_GELU>), g, b, 0, stream, x, w, bias, nullptr, y, N, K, m_total, n_tiles, aux);
        break;
      case EPI_NRES:
        hipLaunchKernelGGL((gemm_tn_w8_kernel<TE, TE, EPI_NRES>), g, b, 0, stream, x, w, bias, y, y, N, K, m_total, n_tiles, aux);
        break;
      default:
        return fail(MANNER_HIP_E_INVALID, "gemm_dln: epilogue %d", (int)epi);
    }
    MANNER_LAUNCH_CHECK();
    return MANNER_HIP_OK;
  }
  if (asm_mode == 4) {                                 // the 4-wave form (A/B only: same bits, slower behind the epilogues)
    const dim3 b4(256);
    switch (epi) {
      case EPI_NORM:
        hipLaunchKernelGGL((gemm_tn_w4_kernel<TE, TE, EPI_NORM>), g, b4, 0, stream, x, w, bias, nullptr, y, N, K, m_total, n_tiles, aux);
        break;
      case EPI_NORM_GELU:
        hipLaunchKernelGGL((gemm_tn_w4_kernel<TE, TE, EPI_NORM_GELU>), g, b4, 0, stream, x, w, bias, nullptr, y, N, K, m_total, n_tiles, aux);
        break;
      case EPI_NRES:
        hipLaunchKernelGGL((gemm_tn_w4_kernel<TE, TE, EPI_NRES>), g, b4, 0, stream, x, w, bias, y, y, N, K, m_total, n_tiles, aux);
        break;
      default:
        return fail(MANNER_HIP_E_INVALID, "gemm_dln: epilogue %d", (int)epi);
    }
    MANNER_LAUNCH_CHECK();
    return MANNER_HIP_OK;
  }
  switch (epi) {
    case EPI_NORM:
      hipLaunchKernelGGL((gemm_tn_x16_kernel<TE, TE, EPI_NORM>), g, b, 0, stream, x, w, bias, nullptr, y, N, K, m_total, n_tiles, aux);
      break;
    case EPI_NORM_GELU:
      hipLaunchKernelGGL((gemm_tn_x16_kernel<TE, TE, EPI_NORM_GELU>), g, b, 0, stream, x, w, bias, nullptr, y, N, K, m_total, n_tiles, aux);
      break;
    case EPI_NRES:   // in place: the residual is the output buffer
      hipLaunchKernelGGL((gemm_tn_x16_kernel<TE, TE, EPI_NRES>), g, b, 0, stream, x, w, bias, y, y, N, K, m_total, n_tiles, aux);
      break;
    default:
      return fail(MANNER_HIP_E_INVALID, "gemm_dln: epilogue %d", (int)epi);
  }
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int gemm_tn_dln(DType dt, Epilogue epi, const void* X, const void* W, const float* bias, const float* vec, const void* mr,
                void* part, void* Y, int64_t m_bound, int N, int K, const int* m_total, hipStream_t stream, int64_t m_exact, DlnFinalize* fin) {
  if (fin) fin->done = false;
  if (N % G_BN || (K * 2) % ROW_BYTES || K < 128 || m_bound % G_BM)
    return fail(MANNER_HIP_E_INVALID, "gemm_dln shape m_bound=%lld N=%d K=%d not tileable", (long long)m_bound, N, K);
  if (!vec || !mr || (epi == EPI_NRES && !part)) return fail(MANNER_HIP_E_INVALID, "gemm_dln: missing operand");
  if (!is_16bit(dt)) return fail(MANNER_HIP_E_INVALID, "gemm_dln: 16-bit element types only");
  if (small_problem(m_bound, N) && K % 64 == 0) {          // few tiles: 128x128 tiles of the same arithmetic (bit-identical results)
    const Drop none{0, 0, 0, 1.f};
    const SmallAux sa{vec, static_cast<const float2*>(mr), static_cast<float2*>(part), m_bound, 0};
#define MANNER_SMALL_DLN(TE_)                                                                                                                    \
  switch (epi) {                                                                                                                                 \
    case EPI_NORM: return launch_small<TE_, TE_, EPI_NORM>(X, W, bias, nullptr, Y, m_bound, N, K, m_total, none, nullptr, stream, sa);            \
    case EPI_NORM_GELU: return launch_small<TE_, TE_, EPI_NORM_GELU>(X, W, bias, nullptr, Y, m_bound, N, K, m_total, none, nullptr, stream, sa);  \
    case EPI_NRES: return launch_small<TE_, TE_, EPI_NRES>(X, W, bias, nullptr, Y, m_bound, N, K, m_total, none, nullptr, stream, sa);            \
    default: return fail(MANNER_HIP_E_INVALID, "gemm_dln: epilogue %d", (int)epi);                                                                \
  }
    if (dt == DT_F16) { MANNER_SMALL_DLN(f16_t) }
    MANNER_SMALL_DLN(bf16_t)
#undef MANNER_SMALL_DLN
  }
  const int n_tiles = N / G_BN;
  const int64_t tiles = (m_bound / G_BM) * n_tiles;
  const int64_t cus = device_cus();
  static const bool plain_stores = getenv("MANNER_HIP_NT_STORES") && atoi(getenv("MANNER_HIP_NT_STORES")) == 0;   // A/B switch
  DlnAux aux{vec, static_cast<const float2*>(mr), static_cast<float2*>(part), m_bound, 0, plain_stores ? 1 : 0};
  set_tile_order(aux, m_exact >= 0 ? m_exact : m_bound, N, K);
  const int col_group = aux.col_group;
  dim3 g;
  if (int rc0 = x16_grid(m_bound, n_tiles, aux, g)) return rc0;
  static const int stagger = getenv("MANNER_HIP_GEMM_STAGGER") ? atoi(getenv("MANNER_HIP_GEMM_STAGGER")) : 0;   // A/B switch
  aux.stagger = stagger;
  // Round-aware split (see split_panels): only where a tail can exist by the host's bound (more tiles than workgroups), for the narrow
  // outputs whose rounds are long and few (N <= 1024: out-projection / FFN2, 3 - 4 column tiles), with the default tile order.
  const char* ts_env = getenv("MANNER_HIP_GEMM_TAIL_SPLIT");                 // A/B switch, read per launch (the tests flip it)
  // OFF by default — measured (drop-in eval, f16, bert-base): B = 64 batches 62.4 -> 61.7 ms, B = 8 batches 8.76 -> 8.99 ms: at the
  // reference's batch size most calls have FEWER tiles than CUs (one under-filled round, which no cut can fix) and the calls that do
  // overflow rarely land in the one-round window, so the extra launch (3 - 8 us, 22 per call) costs more than the cut saves.
  bool split = ts_env && atoi(ts_env) != 0 && col_group == 0 && n_tiles <= 4 && tiles > cus && K % 64 == 0 && stagger == 0;
  if (split && m_exact >= 0) {                       // the caller knows the token count: no tail launch that would find nothing to do
    const int mp = (int)((m_exact + G_BM - 1) / G_BM);
    split = split_panels(mp, n_tiles, (int)cus) < mp;
  }
  if (split) { aux.split_cus = (int)cus; aux.panel_mode = 1; aux.xcd_ranges = 0; }     // the round-aware split cuts at 256-row panels
  if (stagger > 0) aux.xcd_ranges = 0;
  // Round 6: the hand-scheduled kernels (gemm_tn_w8_kernel: the same bits from a 17-22 % shorter main loop) where the panel height is
  // 256 rows by what the host knows of the token count (m_exact, else its bound) — 192-row launches and the A/B forms stay on the
  // compiler-scheduled LDS-DMA kernel.  MANNER_HIP_GEMM_ASM = 8 (default) | 4 (the 4-wave form) | 0 (gemm_tn_x16_kernel everywhere);
  // read per launch: the equality test flips it.
  const int asm_mode = split ? 0 : pick_asm(m_exact >= 0 ? m_exact : m_bound, n_tiles, K, aux);
  if (asm_mode) g = dim3((unsigned)(tiles < cus ? tiles : cus));
  // the row statistics finished inside the launch (nres_fan_in) by the persistent kernels that carry the fan-in: w8 and x16
  // (MANNER_HIP_DLN_FANIN=0: the separate dln_finalize launch, A/B; read per launch)
  if (fin && epi == EPI_NRES && !split && asm_mode != 4 && fin->mr_out && fin->arrive && N / 64 <= 16) {
    const char* fe = getenv("MANNER_HIP_DLN_FANIN");
    if (!fe || atoi(fe) != 0) {
      aux.fin_mr = static_cast<float2*>(fin->mr_out);
      aux.fin_arrive = fin->arrive;
      aux.fin_inv_h = 1.0f / (float)N;
      aux.fin_eps = fin->eps;
      aux.fin_groups = N / 64;
      fin->done = true;
    }
  }
  int rc;
  if (dt == DT_F16) rc = launch_dln<f16_t>(epi, X, W, bias, Y, N, K, m_total, n_tiles, g, aux, stream, asm_mode);
  else rc = launch_dln<bf16_t>(epi, X, W, bias, Y, N, K, m_total, n_tiles, g, aux, stream, asm_mode);
  if (rc || !split) return rc;
  const Drop none{0, 0, 0, 1.f};
  const SmallAux sa{vec, static_cast<const float2*>(mr), static_cast<float2*>(part), m_bound, (int)cus};
#define MANNER_TAIL_DLN(TE_)                                                                                                                     \
  switch (epi) {                                                                                                                                 \
    case EPI_NORM: return launch_small<TE_, TE_, EPI_NORM>(X, W, bias, nullptr, Y, m_bound, N, K, m_total, none, nullptr, stream, sa);            \
    case EPI_NORM_GELU: return launch_small<TE_, TE_, EPI_NORM_GELU>(X, W, bias, nullptr, Y, m_bound, N, K, m_total, none, nullptr, stream, sa);  \
    case EPI_NRES: return launch_small<TE_, TE_, EPI_NRES>(X, W, bias, nullptr, Y, m_bound, N, K, m_total, none, nullptr, stream, sa);            \
    default: return fail(MANNER_HIP_E_INVALID, "gemm_dln: epilogue %d", (int)epi);                                                                \
  }
  if (dt == DT_F16) { MANNER_TAIL_DLN(f16_t) }
  MANNER_TAIL_DLN(bf16_t)
#undef MANNER_TAIL_DLN
}

// `batch` independent Y_b [rows, N] (f32) = X_b [rows, K] . W_b [N, K]^T + bias on 16-bit operands, problem b at element
// offsets b * {xs, ws, ys}; rows, N multiples of 256, K a multiple of 64 and >= 128.  One launch, gridDim.y = batch.
int gemm_tn_batched16(DType in, const void* X, const void* W, const float* bias, float* Y, int batch, int64_t xs, int64_t ws,
                      int64_t ys, int rows, int N, int K, const int* m_total, hipStream_t stream) {
  if (!is_16bit(in) || rows % G_BM || N % G_BN || K % 64 || K < 128 || batch < 1 || batch > 65535)
    return fail(MANNER_HIP_E_INVALID, "gemm_tn_batched16: rows=%d N=%d K=%d batch=%d", rows, N, K, batch);
  const int n_tiles = N / G_BN;
  const int tiles = (rows / G_BM) * n_tiles;
  const int n_cus = device_cus();
  dim3 g((unsigned)(tiles < n_cus ? tiles : n_cus), (unsigned)batch), b(512);
  DlnAux aux{};
  aux.batch_x = xs; aux.batch_w = ws; aux.batch_y = ys;
  aux.panel_mode = 1;            // gridDim.y problems share the CUs: rounds of ONE problem say nothing; 256-row panels
  aux.x_rows = rows;
  if (in == DT_F16)
    hipLaunchKernelGGL((gemm_tn_x16_kernel<f16_t, float, EPI_BIAS>), g, b, 0, stream, static_cast<const f16_t*>(X),
                       static_cast<const f16_t*>(W), bias, static_cast<const f16_t*>(nullptr), Y, N, K, m_total, n_tiles, aux);
  else
    hipLaunchKernelGGL((gemm_tn_x16_kernel<bf16_t, float, EPI_BIAS>), g, b, 0, stream, static_cast<const bf16_t*>(X),
                       static_cast<const bf16_t*>(W), bias, static_cast<const bf16_t*>(nullptr), Y, N, K, m_total, n_tiles, aux);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

// Y [m, N] f32 = dropout(X W^T + bias) + R   (16-bit operands, f32 residual / output; 256-tileable shapes, K >= 128): the
// training forward's attention-output and FFN-output projections with their dropout and residual add in the epilogue
int gemm_tn_drop_res(DType in, const void* X, const void* W, const float* bias, const float* residual, float* Y, int64_t m_bound, int N,
                     int K, const int* m_total, const Drop& drop, const int32_t* rowmap, hipStream_t stream) {
  if (!is_16bit(in) || m_bound % G_BM || N % G_BN || K < 128 || (K * 2) % ROW_BYTES || !residual)
    return fail(MANNER_HIP_E_INVALID, "gemm_tn_drop_res: m_bound=%lld N=%d K=%d not tileable", (long long)m_bound, N, K);
  if (small_problem(m_bound, N) && K % 64 == 0) {
    if (in == DT_F16) return launch_small<f16_t, float, EPI_BIAS_RES_F32>(X, W, bias, residual, Y, m_bound, N, K, m_total, drop, rowmap, stream);
    return launch_small<bf16_t, float, EPI_BIAS_RES_F32>(X, W, bias, residual, Y, m_bound, N, K, m_total, drop, rowmap, stream);
  }
  const int n_tiles = N / G_BN;
  DlnAux aux{};
  dim3 g, b(512);
  if (int rc = x16_grid(m_bound, n_tiles, aux, g)) return rc;
  set_tile_order(aux, m_bound, N, K);
  aux.drop_seed = drop.seed; aux.drop_site = drop.site; aux.drop_thr = drop.thr; aux.drop_scale = drop.scale; aux.drop_rowmap = rowmap;
  if (in == DT_F16)
    hipLaunchKernelGGL((gemm_tn_x16_kernel<f16_t, float, EPI_BIAS_RES_F32>), g, b, 0, stream, static_cast<const f16_t*>(X), static_cast<const f16_t*>(W),
                       bias, reinterpret_cast<const f16_t*>(residual), Y, N, K, m_total, n_tiles, aux);
  else
    hipLaunchKernelGGL((gemm_tn_x16_kernel<bf16_t, float, EPI_BIAS_RES_F32>), g, b, 0, stream, static_cast<const bf16_t*>(X), static_cast<const bf16_t*>(W),
                       bias, reinterpret_cast<const bf16_t*>(residual), Y, N, K, m_total, n_tiles, aux);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

bool gemm_gelu_fusable(DType in, int64_t m_bound, int N, int K) {
  return is_16bit(in) && m_bound > 0 && m_bound % G_BM == 0 && N % G_BN == 0 && K >= 128 && (K * 2) % ROW_BYTES == 0 && !small_problem(m_bound, N);
}

template <typename TE, typename TOut, int EPI>
static int launch_gelu_fused(const void* X, const void* W, const float* bias, void* Y, int64_t m_bound, int N, int K, const int* m_total,
                             DlnAux aux, hipStream_t stream) {
  const int n_tiles = N / G_BN;
  dim3 g, b(512);
  if (int rc = x16_grid(m_bound, n_tiles, aux, g)) return rc;
  set_tile_order(aux, m_bound, N, K);
  hipLaunchKernelGGL((gemm_tn_x16_kernel<TE, TOut, EPI>), g, b, 0, stream, static_cast<const TE*>(X), static_cast<const TE*>(W), bias,
                     static_cast<const TE*>(nullptr), static_cast<TOut*>(Y), N, K, m_total, n_tiles, aux);
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

int gemm_tn_gelu_dual(DType in, const void* X, const void* W, const float* bias, float* Y, void* G16, int64_t m_bound, int N, int K,
                      const int* m_total, hipStream_t stream) {
  if (!gemm_gelu_fusable(in, m_bound, N, K) || !G16 || !Y)
    return fail(MANNER_HIP_E_INVALID, "gemm_tn_gelu_dual: m_bound=%lld N=%d K=%d", (long long)m_bound, N, K);
  DlnAux aux{};
  aux.aux16 = G16;
  if (in == DT_F16) return launch_gelu_fused<f16_t, float, EPI_BIAS_GELU_DUAL>(X, W, bias, Y, m_bound, N, K, m_total, aux, stream);
  return launch_gelu_fused<bf16_t, float, EPI_BIAS_GELU_DUAL>(X, W, bias, Y, m_bound, N, K, m_total, aux, stream);
}

int gemm_tn_gelu_grad(DType in, const void* X, const void* W, const float* zero_bias, const float* pre, void* Y16, int64_t m_bound, int N,
                      int K, const int* m_total, hipStream_t stream) {
  if (!gemm_gelu_fusable(in, m_bound, N, K) || !pre || !Y16)
    return fail(MANNER_HIP_E_INVALID, "gemm_tn_gelu_grad: m_bound=%lld N=%d K=%d", (long long)m_bound, N, K);
  DlnAux aux{};
  aux.aux32 = pre;
  if (in == DT_F16) return launch_gelu_fused<f16_t, f16_t, EPI_GELU_GRAD>(X, W, zero_bias, Y16, m_bound, N, K, m_total, aux, stream);
  return launch_gelu_fused<bf16_t, bf16_t, EPI_GELU_GRAD>(X, W, zero_bias, Y16, m_bound, N, K, m_total, aux, stream);
}

int gemm_tn_gelu_dual16(DType in, const void* X, const void* W, const float* bias, void* Pre16, void* G16, int64_t m_bound, int N, int K,
                        const int* m_total, hipStream_t stream) {
  if (!gemm_gelu_fusable(in, m_bound, N, K) || !G16 || !Pre16)
    return fail(MANNER_HIP_E_INVALID, "gemm_tn_gelu_dual16: m_bound=%lld N=%d K=%d", (long long)m_bound, N, K);
  DlnAux aux{};
  aux.aux16 = G16;
  if (in == DT_F16) return launch_gelu_fused<f16_t, f16_t, EPI_BIAS_GELU_DUAL16>(X, W, bias, Pre16, m_bound, N, K, m_total, aux, stream);
  return launch_gelu_fused<bf16_t, bf16_t, EPI_BIAS_GELU_DUAL16>(X, W, bias, Pre16, m_bound, N, K, m_total, aux, stream);
}

int gemm_tn_gelu_grad16(DType in, const void* X, const void* W, const float* zero_bias, const void* pre16, void* Y16, int64_t m_bound, int N,
                        int K, const int* m_total, hipStream_t stream) {
  if (!gemm_gelu_fusable(in, m_bound, N, K) || !pre16 || !Y16)
    return fail(MANNER_HIP_E_INVALID, "gemm_tn_gelu_grad16: m_bound=%lld N=%d K=%d", (long long)m_bound, N, K);
  DlnAux aux{};
  aux.pre16 = pre16;
  if (in == DT_F16) return launch_gelu_fused<f16_t, f16_t, EPI_GELU_GRAD16>(X, W, zero_bias, Y16, m_bound, N, K, m_total, aux, stream);
  return launch_gelu_fused<bf16_t, bf16_t, EPI_GELU_GRAD16>(X, W, zero_bias, Y16, m_bound, N, K, m_total, aux, stream);
}

int gemm_tn(DType in, DType out, Epilogue epi, const void* X, const void* W, const float* bias,
            const void* residual, void* Y, int64_t m_bound, int N, int K, const int* m_total,
            hipStream_t stream) {
  const int esz = in == DT_F32 ? 4 : 2;
  if (N % BN || (K * esz) % ROW_BYTES || m_bound % BM)
    return fail(MANNER_HIP_E_INVALID, "gemm shape m_bound=%lld N=%d K=%d not tileable", (long long)m_bound, N, K);
  if ((epi == EPI_BIAS_RES || epi == EPI_BIAS_RES_F32) && !residual) return fail(MANNER_HIP_E_INVALID, "gemm residual missing");
  if (epi == EPI_BIAS_RES_F32 && !(is_16bit(in) && out == DT_F32 && m_bound % G_BM == 0 && N % G_BN == 0 && K >= 128))
    return fail(MANNER_HIP_E_INVALID, "EPI_BIAS_RES_F32 needs 16-bit operands, f32 output and 256-tileable shapes");
  if (epi == EPI_BIAS_GELU_SPLIT3 && !(is_16bit(in) && out == in && m_bound % G_BM == 0 && N % G_BN == 0 && K >= 128))
    return fail(MANNER_HIP_E_INVALID, "EPI_BIAS_GELU_SPLIT3 needs 16-bit operands and output and 256-tileable shapes");
  if (out != DT_F32 && out != in) return fail(MANNER_HIP_E_INVALID, "gemm dtype combination unsupported");
  static const bool use_v1 = getenv("MANNER_HIP_GEMM_V1") != nullptr;   // A/B switch for development
  if (!use_v1 && is_16bit(in) && m_bound % G_BM == 0 && N % G_BN == 0 && K % 64 == 0 && small_problem(m_bound, N) &&
      (epi == EPI_BIAS || epi == EPI_BIAS_RES_F32 || (epi == EPI_BIAS_RES && out == DT_F32) || (epi == EPI_BIAS_GELU && out == in))) {
    const Drop none{0, 0, 0, 1.f};
    const float* rf = static_cast<const float*>(residual);
    if (epi == EPI_BIAS_RES) {                              // the inference engine's [CLS] tail (same bits as the persistent kernel)
      if (in == DT_F16) return launch_small<f16_t, float, EPI_BIAS_RES>(X, W, bias, rf, Y, m_bound, N, K, m_total, none, nullptr, stream);
      return launch_small<bf16_t, float, EPI_BIAS_RES>(X, W, bias, rf, Y, m_bound, N, K, m_total, none, nullptr, stream);
    }
    if (epi == EPI_BIAS_GELU) {
      if (in == DT_F16) return launch_small<f16_t, f16_t, EPI_BIAS_GELU>(X, W, bias, nullptr, Y, m_bound, N, K, m_total, none, nullptr, stream);
      return launch_small<bf16_t, bf16_t, EPI_BIAS_GELU>(X, W, bias, nullptr, Y, m_bound, N, K, m_total, none, nullptr, stream);
    }
    if (epi == EPI_BIAS_RES_F32) {
      if (in == DT_F16) return launch_small<f16_t, float, EPI_BIAS_RES_F32>(X, W, bias, rf, Y, m_bound, N, K, m_total, none, nullptr, stream);
      return launch_small<bf16_t, float, EPI_BIAS_RES_F32>(X, W, bias, rf, Y, m_bound, N, K, m_total, none, nullptr, stream);
    }
    if (in == DT_F16 && out == DT_F16) return launch_small<f16_t, f16_t, EPI_BIAS>(X, W, bias, nullptr, Y, m_bound, N, K, m_total, none, nullptr, stream);
    if (in == DT_F16) return launch_small<f16_t, float, EPI_BIAS>(X, W, bias, nullptr, Y, m_bound, N, K, m_total, none, nullptr, stream);
    if (out == DT_BF16) return launch_small<bf16_t, bf16_t, EPI_BIAS>(X, W, bias, nullptr, Y, m_bound, N, K, m_total, none, nullptr, stream);
    return launch_small<bf16_t, float, EPI_BIAS>(X, W, bias, nullptr, Y, m_bound, N, K, m_total, none, nullptr, stream);
  }
  if (!use_v1 && m_bound % G_BM == 0 && N % G_BN == 0) {
    static const bool use_x32 = getenv("MANNER_HIP_GEMM_X32") != nullptr;   // A/B: bf16 on the 32x32x16 shape
    if (in == DT_F16 && K >= 128) {
      if (out == DT_F16) return launch_x16<f16_t, f16_t>(epi, X, W, bias, residual, Y, m_bound, N, K, m_total, stream);
      return launch_x16<f16_t, float>(epi, X, W, bias, residual, Y, m_bound, N, K, m_total, stream);
    }
    if (in == DT_BF16 && !use_x32 && K >= 128) {          // the staggered DMA schedule needs >= 2 K-steps
      if (out == DT_BF16) return launch_x16<bf16_t, bf16_t>(epi, X, W, bias, residual, Y, m_bound, N, K, m_total, stream);
      return launch_x16<bf16_t, float>(epi, X, W, bias, residual, Y, m_bound, N, K, m_total, stream);
    }
    if (in == DT_BF16 && out == DT_BF16) return launch_big<bf16_t, bf16_t>(epi, X, W, bias, residual, Y, m_bound, N, K, m_total, stream);
    if (in == DT_BF16 && out == DT_F32) return launch_big<bf16_t, float>(epi, X, W, bias, residual, Y, m_bound, N, K, m_total, stream);
    if (in == DT_F32 && out == DT_F32) return launch_big<float, float>(epi, X, W, bias, residual, Y, m_bound, N, K, m_total, stream);
  }
  if (in == DT_BF16 && out == DT_BF16) return launch<bf16_t, bf16_t>(epi, X, W, bias, residual, Y, m_bound, N, K, m_total, stream);
  if (in == DT_BF16 && out == DT_F32) return launch<bf16_t, float>(epi, X, W, bias, residual, Y, m_bound, N, K, m_total, stream);
  if (in == DT_F16 && out == DT_F16) return launch<f16_t, f16_t>(epi, X, W, bias, residual, Y, m_bound, N, K, m_total, stream);
  if (in == DT_F16 && out == DT_F32) return launch<f16_t, float>(epi, X, W, bias, residual, Y, m_bound, N, K, m_total, stream);
  if (in == DT_F32 && out == DT_F32) return launch<float, float>(epi, X, W, bias, residual, Y, m_bound, N, K, m_total, stream);
  return fail(MANNER_HIP_E_INVALID, "gemm dtype combination unsupported");
}

}  // namespace manner
