// Weight-gradient GEMM of the training path without transposed operand copies (torch.nn.Linear's backward
// dW = dY^T X behind reference cr_module.py:140-171 / news_encoder.py:24-37):
//
//     dW [N, K] (f32) = sum over the token rows m of  dY [m, N]^T . X [m, K]          (dY, X: 16-bit, ROW-major)
//
// The reduction runs over the SLOW index of both operands, so the K-contiguous kernels of gemm.hip need dY^T and X^T — two
// transposing copy kernels per weight gradient, 11 % of a training step.  Here the row-major tiles go to LDS as they lie in
// memory (LDS-DMA, 32 token rows x 256 columns per operand and stage) and the MFMA operands are read TRANSPOSED with
// ds_read_b64_tr_b16 (cdna_hip_programming.md T10): a 16-lane group fetches a block of 4 token rows x 16 columns and every lane
// receives one column's 4 values — 8 token rows per lane after two reads, exactly the k-run of a v_mfma_f32_32x32x16 operand.
// Both operands are read by the same routine, so the (permuted) order of the 16 token rows inside an MFMA k-step is the same on
// both sides and the sum is over the same pairs.
//   * tile 256 (n) x 256 (k) of dW per workgroup, 8 waves as 2 (n) x 4 (k), wave tile 128 x 64 = 4 x 2 accumulators of 32 x 32;
//   * the token axis is cut into gridDim.y slices (few output tiles: 9 .. 36; every CU gets one) whose partial results are added in
//     a fixed order afterwards (reduce_partials in train.hip) — slices = 1 writes dW itself;
//   * LDS image of an operand's 32 x 128-column half: 256-byte rows, 16-byte chunk ch of row r at slot ch ^ (((r & 3) << 2) |
//     ((r >> 2) & 3)) — the dual-use image (b) of T10: the transposed reads of the 32x32x16 operand are conflict-free; the swizzle
//     is applied on the per-lane global SOURCE address of the DMA (lane-linear destination);
//   * 4 LDS stages (128 KiB): token rows 32 (s + 3) .. are requested while stage s is multiplied, counted vmcnt;
//   * rows >= *m_total (never written by the producers) are replaced by a zero page at the source address.
#include "train_common.h"

namespace manner {
namespace {

constexpr int WG_STAGES = 4, WG_ROWS = 32, WG_HALF_BYTES = WG_ROWS * 256, WG_STAGE_BYTES = 4 * WG_HALF_BYTES;   // A: 2 halves, B: 2 halves

typedef int i32x2 __attribute__((ext_vector_type(2)));

template <int OFF>
__device__ __forceinline__ void tr_read(i32x2& d, uint32_t addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// one MFMA k-step (16 token rows at byte offset OFF of the stage at `sb`): 12 operand fragments = 24 transposed reads, 8 MFMAs
template <int OFF, typename TE>
__device__ __forceinline__ void kstep_impl(uint32_t sb, const uint32_t (&a_adr)[4][2], const uint32_t (&b_adr)[2][2], f32x16 (&acc)[4][2]) {
  typedef typename E16<TE>::v8 e16x8;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x2 al[4], ah[4], bl[2], bh[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) { tr_read<OFF>(al[i], sb + a_adr[i][0]); tr_read<OFF>(ah[i], sb + a_adr[i][1]); }
#pragma unroll
  for (int j = 0; j < 2; ++j) { tr_read<OFF>(bl[j], sb + b_adr[j][0]); tr_read<OFF>(bh[j], sb + b_adr[j][1]); }
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]), "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]),
                 "+v"(bl[0]), "+v"(bl[1]), "+v"(bh[0]), "+v"(bh[1]));
  e16x8 a[4], b[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(e16x8, i32x4{al[i][0], al[i][1], ah[i][0], ah[i][1]});
#pragma unroll
  for (int j = 0; j < 2; ++j) b[j] = __builtin_bit_cast(e16x8, i32x4{bl[j][0], bl[j][1], bh[j][0], bh[j][1]});
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = E16<TE>::mfma32(a[i], b[j], acc[i][j]);
}

template <typename TE>
__global__ __launch_bounds__(512, 1) void wgrad_tr_kernel(const TE* __restrict__ dY, const TE* __restrict__ X, float* __restrict__ out,
                                                          int N, int K, int k_tiles, int64_t rows_per_slice,
                                                          const int* __restrict__ m_total, const TE* __restrict__ zero_page) {
  __shared__ __attribute__((aligned(1024))) char lds[WG_STAGES * WG_STAGE_BYTES];       // 128 KiB
  const int tile = blockIdx.x, nt = tile / k_tiles, kt = tile - nt * k_tiles;
  const int n0 = nt * 256, k0 = kt * 256;
  const int64_t M = *m_total;
  const int64_t m_begin = (int64_t)blockIdx.y * rows_per_slice;
  int64_t m_end = m_begin + rows_per_slice;
  if (m_end > M) m_end = M;
  const int steps = m_end > m_begin ? (int)((m_end - m_begin + WG_ROWS - 1) / WG_ROWS) : 0;

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int wn = wave & 1, wk = wave >> 1;

  // ---- DMA: 32 pieces of 1 KiB per stage (operand, half, 4 rows), 4 per wave
  const TE* src[4];
  int64_t ld[4];
  int drow[4], ldst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = 4 * wave + i, opnd = t >> 4, half = (t >> 3) & 1, piece = t & 7;
    const int row = 4 * piece + (lane >> 4), slot = lane & 15;
    const int ch = slot ^ (((row & 3) << 2) | ((row >> 2) & 3));
    drow[i] = row;
    ld[i] = opnd ? K : N;
    src[i] = (opnd ? X + k0 : dY + n0) + half * 128 + ch * 8;
    ldst[i] = (2 * opnd + half) * WG_HALF_BYTES + piece * 1024;
  }
  auto stage = [&](int buf, int s) {
    char* base = lds + buf * WG_STAGE_BYTES;
    const int64_t m0 = m_begin + (int64_t)s * WG_ROWS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t m = m0 + drow[i];
      const TE* g = m < m_end ? src[i] + m * ld[i] : zero_page;
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(base + ldst[i]), 16, 0, 0);
    }
  };

  // ---- transposed fragment reads: lane 4q + p of a 16-lane group supplies row q, columns 4p .. 4p+3 of its block
  const int gi = lane & 15, q = gi >> 2, p = gi & 3, nb = (lane >> 4) & 1, hh = lane >> 5;
  auto tr_off = [&](int chunk0, int hi) {             // block: rows 4 hh + 8 hi + (0..3), chunks chunk0, chunk0 + 1 (16 columns)
    const int r = 4 * hh + 8 * hi + q;
    const int ch = (chunk0 + (p >> 1)) ^ (((r & 3) << 2) | ((r >> 2) & 3));
    return 256 * r + 16 * ch + 8 * (p & 1);
  };
  int a_off[4][2], b_off[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) a_off[i][hi] = wn * WG_HALF_BYTES + tr_off(4 * i + 2 * nb, hi);
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) b_off[j][hi] = (2 + (wk >> 1)) * WG_HALF_BYTES + tr_off(8 * (wk & 1) + 4 * j + 2 * nb, hi);
  // The reads are inline asm: for the builtin the compiler cannot tell the LDS-DMA writes of the stage just requested from the
  // stage being read and puts s_waitcnt vmcnt(0) in front of every read — i.e. it would wait for the data it has just asked for.
  // The asm reads are invisible to its lgkmcnt bookkeeping, so all 24 of a k-step are issued and ONE explicit s_waitcnt lgkmcnt(0)
  // carries their 24 destination registers as operands (nothing that uses them can be scheduled in front of it).
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(lds);
  uint32_t a_adr[4][2], b_adr[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) a_adr[i][hi] = lds0 + (uint32_t)a_off[i][hi];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) b_adr[j][hi] = lds0 + (uint32_t)b_off[j][hi];
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
  for (int s0 = 0; s0 < WG_STAGES - 1; ++s0)
    if (s0 < steps) stage(s0, s0);
  for (int s = 0; s < steps; ++s) {
    const int ahead = min(steps, s + WG_STAGES - 1) - (s + 1);          // requested and younger than s: may fly on
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                       // everyone's pieces of s landed; everyone left stage s - 1
    if (s + WG_STAGES - 1 < steps) stage((s + WG_STAGES - 1) % WG_STAGES, s + WG_STAGES - 1);
    const uint32_t sb = (uint32_t)((s % WG_STAGES) * WG_STAGE_BYTES);
    kstep_impl<0, TE>(sb, a_adr, b_adr, acc);                                    // 16 token rows per MFMA k-step
    kstep_impl<16 * 256, TE>(sb, a_adr, b_adr, acc);
  }

  // acc[i][j][e] = dW[n0 + 128 wn + 32 i + (e & 3) + 8 (e >> 2) + 4 h][k0 + 64 wk + 32 j + rr]: 32 lanes write 128 contiguous bytes
  const int rr = lane & 31, h = lane >> 5;
  float* o = out + (size_t)blockIdx.y * N * K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + 128 * wn + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h, k = k0 + 64 * wk + 32 * j + rr;
        o[(size_t)n * K + k] = acc[i][j][e];
      }
}

}  // namespace

// out [slices][N, K] f32 (slices == 1: dW itself) = per-slice sums of dY[m, :]^T X[m, :] over token rows
// m in [slice * rows_per_slice, (slice + 1) * rows_per_slice) below *m_total.  dY [*, N], X [*, K] row-major 16-bit `dt`;
// N, K multiples of 256; rows_per_slice a multiple of 32; zero_page: >= 16 zero bytes on the device.
int wgrad_tr(DType dt, const void* dY, const void* X, float* out, int N, int K, int slices, int64_t rows_per_slice,
             const int* m_total, const void* zero_page, hipStream_t stream) {
  if (!is_16bit(dt) || N % 256 || K % 256 || N <= 0 || K <= 0 || slices < 1 || slices > 65535 || rows_per_slice % WG_ROWS || rows_per_slice <= 0 ||
      !dY || !X || !out || !m_total || !zero_page)
    return fail(MANNER_HIP_E_INVALID, "wgrad_tr: N=%d K=%d slices=%d rows_per_slice=%lld", N, K, slices, (long long)rows_per_slice);
  const int k_tiles = K / 256;
  const dim3 g((unsigned)((N / 256) * k_tiles), (unsigned)slices), b(512);
  if (dt == DT_F16)
    hipLaunchKernelGGL(wgrad_tr_kernel<f16_t>, g, b, 0, stream, static_cast<const f16_t*>(dY), static_cast<const f16_t*>(X), out, N, K, k_tiles,
                       rows_per_slice, m_total, static_cast<const f16_t*>(zero_page));
  else
    hipLaunchKernelGGL(wgrad_tr_kernel<bf16_t>, g, b, 0, stream, static_cast<const bf16_t*>(dY), static_cast<const bf16_t*>(X), out, N, K, k_tiles,
                       rows_per_slice, m_total, static_cast<const bf16_t*>(zero_page));
  MANNER_LAUNCH_CHECK();
  return MANNER_HIP_OK;
}

}  // namespace manner
