// Development lab (round 4): does a 256 x 384 output tile beat 256 x 256 on the N = 768 GEMMs of the encoder
// (attention-output projection K = 768, FFN2 K = 3072)?  Not shipped, not part of libmanner_hip.so.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imanner_amd/csrc tools/tile_lab.hip -o gpurun_out/tile_lab && gpurun_out/tile_lab
//
// ONE main loop, templated on the tile width BN (256 or 384), so that the two shapes are compared at equal pipeline structure:
// 512 threads = 8 waves as 2 (m) x 4 (n), wave tile 128 tokens x BN / 4 units on v_mfma_f32_16x16x32_bf16 (BN = 384: 6 x 8 tiles =
// 192 accumulator registers per lane, the limit VERDICT names), K-step = 64 elements = 128-byte rows, operands staged by LDS-DMA
// into XOR-swizzled row images (the production kernel's image: chunk c of row r at slot c ^ ((r >> 1) & 7), conflict-free
// ds_read_b128), double-buffered: 2 x BN x 128 B of weights + 2 x 256 x 128 B of activations = 160 KiB for BN = 384 — all of the
// CU's LDS, which is why the production kernel's third activation stage and its cross-tile prefetch cannot come along.  Fragments
// are read by inline-asm ds_read_b128 with counted waits (hipcc puts vmcnt(0) in front of a C++ LDS load while LDS-DMA is in
// flight).  No epilogue (the accumulators are folded into one value per lane): this is the MAIN LOOP's rate, to be set against the
// production kernel's main-loop-only rate (tools/gemm_lab.hip, abl1) — and it bounds what the tile shape can buy:
// the epilogue work per output element does not depend on the tile shape.
#include <stdio.h>
#include <stdlib.h>
#include <random>
#include <vector>

#include "../manner_amd/csrc/common.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OFF>
__device__ __forceinline__ void lds_read128(bf16x8& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}

template <int BN>
__global__ __launch_bounds__(512, 2) void lab_tile_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ W, float* __restrict__ Y, int N,
                                                          int K) {
  constexpr int BM = 256, BK = 64, ROWB = 128;
  constexpr int WT = BN / 64;                               // 16-unit tiles per wave: 4 or 6
  constexpr int WST = BN * ROWB, XST = BM * ROWB;           // stage bytes
  constexpr int PW = BN / 8, PX = BM / 8, PCS = PW + PX;    // LDS-DMA pieces (8 rows = 1 KiB) per K-step
  static_assert(PCS % 8 == 0, "pieces split evenly over the 8 waves");
  __shared__ __attribute__((aligned(1024))) char lds[2 * WST + 2 * XST];
  const int n_tiles = N / BN;
  const int mt = blockIdx.x / n_tiles, nt = blockIdx.x % n_tiles;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int l15 = lane & 15, lq = lane >> 4, wn = wave & 3, wm = wave >> 2;
  const uint32_t lds0 = (uint32_t)(size_t)LDS_PTR(lds);
  const bf16_t* wsrc = W + (size_t)nt * BN * K;
  const bf16_t* xsrc = X + (size_t)mt * BM * K;
  const int lrow = lane >> 3;
  auto issue = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < PCS / 8; ++i) {
      const int pc = wave + 8 * i;                          // wave-uniform
      const bool is_w = pc < PW;
      const int prow = is_w ? 8 * pc : 8 * (pc - PW);
      const bf16_t* g = (is_w ? wsrc : xsrc) + (size_t)(prow + lrow) * K + k0 + (((lane & 7) ^ (((prow + lrow) >> 1) & 7)) * 8);
      char* dst = lds + (is_w ? buf * WST : 2 * WST + buf * XST) + prow * ROWB;
      __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(dst), 16, 0, 0);
    }
  };
  f32x4 acc[WT][8];
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int swz = (l15 >> 1) & 7;
  const uint32_t woff = (uint32_t)((wn * (BN / 4) + l15) * ROWB), xoff = (uint32_t)((wm * 128 + l15) * ROWB);
  const int nk = K / BK;
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) issue(cur ^ 1, (kt + 1) * BK);
    const uint32_t wb = lds0 + cur * WST + woff, xb = lds0 + 2 * WST + cur * XST + xoff;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const uint32_t coff = (uint32_t)(((4 * s2 + lq) ^ swz) << 4);
      bf16x8 wf[WT], xf, xn;
#pragma unroll
      for (int a = 0; a < WT; ++a) lds_read128<0>(wf[a], wb + a * 16 * ROWB + coff);
      lds_read128<0>(xf, xb + coff);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (b + 1 < 8) {
          lds_read128<0>(xn, xb + (b + 1) * 16 * ROWB + coff);
          if (b == 0) {
            if constexpr (WT == 6) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4]), "+v"(wf[5]), "+v"(xf));
            else asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(xf));
          } else {
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(xf));
          }
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf));
        }
#pragma unroll
        for (int a = 0; a < WT; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], xf, acc[a][b], 0, 0, 0);
        xf = xn;
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < WT; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) s += (acc[a][b][0] + acc[a][b][1]) + (acc[a][b][2] + acc[a][b][3]);
  Y[(size_t)blockIdx.x * 512 + tid] = s;
}

template <typename F>
static double time_ms(F f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536;
  const int iters = argc > 2 ? atoi(argv[2]) : 30;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  const size_t xe = (size_t)M * 3072, we = (size_t)768 * 3072;
  std::vector<bf16_t> hx(xe), hw(we);
  for (auto& v : hx) v = (bf16_t)nd(rng);
  for (auto& v : hw) v = (bf16_t)(0.03f * nd(rng));
  bf16_t *X, *W; float* Y;
  CK(hipMalloc(&X, xe * 2)); CK(hipMalloc(&W, we * 2)); CK(hipMalloc(&Y, (size_t)(M / 256) * 3 * 512 * 4));
  CK(hipMemcpy(X, hx.data(), xe * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hw.data(), we * 2, hipMemcpyHostToDevice));
  // correctness of both shapes against each other: the per-tile sums of a 768-wide row panel must agree (768 = 3 x 256 = 2 x 384)
  for (int K : {768, 3072}) {
    const int N = 768;
    std::vector<float> y256((size_t)(M / 256) * 3 * 512), y384((size_t)(M / 256) * 2 * 512);
    hipLaunchKernelGGL(lab_tile_kernel<256>, dim3((M / 256) * 3), dim3(512), 0, 0, X, W, Y, N, K);
    CK(hipMemcpy(y256.data(), Y, y256.size() * 4, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(lab_tile_kernel<384>, dim3((M / 256) * 2), dim3(512), 0, 0, X, W, Y, N, K);
    CK(hipMemcpy(y384.data(), Y, y384.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int mt = 0; mt < M / 256; ++mt) {
      double a = 0, b = 0;
      for (int i = 0; i < 3 * 512; ++i) a += y256[(size_t)mt * 3 * 512 + i];
      for (int i = 0; i < 2 * 512; ++i) b += y384[(size_t)mt * 2 * 512 + i];
      worst = fabs(a - b) > worst ? fabs(a - b) : worst;
      scale = fabs(a) > scale ? fabs(a) : scale;
    }
    printf("check K=%d: row-panel sums of the two tilings differ by at most %.3e (largest panel sum %.3e)\n", K, worst, scale);
    const double fl = 2.0 * M * N * K;
    const double ms256 = time_ms([&] { hipLaunchKernelGGL(lab_tile_kernel<256>, dim3((M / 256) * 3), dim3(512), 0, 0, X, W, Y, N, K); }, iters);
    const double ms384 = time_ms([&] { hipLaunchKernelGGL(lab_tile_kernel<384>, dim3((M / 256) * 2), dim3(512), 0, 0, X, W, Y, N, K); }, iters);
    printf("main loop only, M=%d N=%d K=%d:  256x256 tiles (%d = %.1f per CU) %.1f us %.0f TF   |   256x384 tiles (%d = %.1f per CU) %.1f us %.0f TF   |   ratio %.3f\n",
           M, N, K, (M / 256) * 3, (M / 256) * 3 / 256.0, ms256 * 1e3, fl / ms256 / 1e9, (M / 256) * 2, (M / 256) * 2 / 256.0, ms384 * 1e3,
           fl / ms384 / 1e9, ms256 / ms384);
  }
  return 0;
}
