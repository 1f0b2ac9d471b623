// Development lab: times the encoder GEMM kernels standalone on random data (not shipped, not
// part of libmanner_hip.so).  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imanner_amd/csrc
//   tools/gemm_lab.hip -o gpurun_out/gemm_lab && gpurun_out/gemm_lab
#include <stdio.h>
#include <vector>
#include <random>
#include "../manner_amd/csrc/gemm.hip"
namespace manner { int fail(int code, const char* fmt, ...) { fprintf(stderr, "fail %d: %s\n", code, fmt); return code; } }
using namespace manner;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

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

// pure-MFMA ceiling: 8 waves per CU, register operands, 8 independent accumulators per wave
__global__ __launch_bounds__(512, 1) void mfma_peak_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const bf16x8*>(src + (size_t)(threadIdx.x * 8 + i) * 8);
    b[i] = *reinterpret_cast<const bf16x8*>(src + (size_t)(threadIdx.x * 8 + 4 + i) * 8);
  }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(t + kc) & 3], b[(t >> 1) & 3], acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  dst[blockIdx.x * 512 + threadIdx.x] = s + lane;
}

__global__ __launch_bounds__(512, 1) void mfma_peak16_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = *reinterpret_cast<const bf16x8*>(src + (size_t)(threadIdx.x * 8 + i) * 8);
    b[i] = *reinterpret_cast<const bf16x8*>(src + (size_t)(threadIdx.x * 8 + 4 + i) * 8);
  }
  f32x4 acc[32];
  for (int i = 0; i < 32; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int t = 0; t < 32; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(t + kc) & 3], b[(t >> 3) & 3], acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 32; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
  dst[blockIdx.x * 512 + threadIdx.x] = s + lane;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536;
  struct Shape { const char* name; int N, K; Epilogue epi; bool f32out; };
  Shape shapes[] = {{"qkv", 2304, 768, EPI_BIAS, false}, {"out", 768, 768, EPI_BIAS_RES, true},
                    {"ffn1", 3072, 768, EPI_BIAS_GELU, false}, {"ffn2", 768, 3072, EPI_BIAS_RES, true},
                    {"ffn1h", 1536, 768, EPI_BIAS_GELU, false}, {"qkvh", 1280, 768, EPI_BIAS, false}, {"qkv1", 256, 768, EPI_BIAS, false}};
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  const size_t maxe = (size_t)M * 3072;
  std::vector<bf16_t> h(maxe);
  for (size_t i = 0; i < maxe; ++i) h[i] = (bf16_t)nd(rng);
  bf16_t *X, *W, *R; float* bias; void* Y; int* mtot;
  CK(hipMalloc(&X, maxe * 2)); CK(hipMalloc(&W, (size_t)3072 * 3072 * 2)); CK(hipMalloc(&R, maxe * 2));
  CK(hipMalloc(&Y, maxe * 4)); CK(hipMalloc(&bias, 3072 * 4)); CK(hipMalloc(&mtot, 4));
  CK(hipMemcpy(X, h.data(), maxe * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(R, h.data(), maxe * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), (size_t)3072 * 3072 * 2, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, 3072 * 4));
  CK(hipMemcpy(mtot, &M, 4, hipMemcpyHostToDevice));
  {
    const int iters = 2000;
    for (int blocks : {256, 512}) {
      double ms = time_ms([&] { hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(512), 0, 0, X, (float*)Y, iters); }, 10);
      const double fl = (double)blocks * 8 * iters * 32 * (2.0 * 32 * 32 * 16);
      printf("mfma_peak blocks=%d: %.1f us  %.0f TF  (=> %.2f GHz if issue-bound)\n", blocks, ms * 1e3, fl / ms / 1e9,
             fl / ms / 1e9 / 2500.0 * 2.4);
    }
  }
  {
    const int iters = 2000;
    for (int blocks : {256, 512}) {
      double ms = time_ms([&] { hipLaunchKernelGGL(mfma_peak16_kernel, dim3(blocks), dim3(512), 0, 0, X, (float*)Y, iters); }, 10);
      const double fl = (double)blocks * 8 * iters * 64 * (2.0 * 16 * 16 * 32);
      printf("mfma_peak 16x16x32 blocks=%d: %.1f us  %.0f TF\n", blocks, ms * 1e3, fl / ms / 1e9);
    }
  }
  for (auto& s : shapes) {
    const int n_tiles = s.N / G_BN;
    dim3 g((M / G_BM) * n_tiles), b(512);
    const double fl = 2.0 * M * s.N * s.K;
    auto run = [&](int abl) {
#define LAUNCH(TO, EPI, ABL) hipLaunchKernelGGL((gemm_tn_big_kernel<bf16_t, TO, EPI, ABL>), g, b, 0, 0, X, W, bias, R, (TO*)Y, s.N, s.K, mtot, n_tiles)
#define BY_ABL(TO, EPI) switch (abl) { case 0: LAUNCH(TO, EPI, 0); break; case 1: LAUNCH(TO, EPI, 1); break; case 2: LAUNCH(TO, EPI, 2); break; case 3: LAUNCH(TO, EPI, 3); break; }
      if (s.epi == EPI_BIAS) { BY_ABL(bf16_t, EPI_BIAS) }
      else if (s.epi == EPI_BIAS_GELU) { BY_ABL(bf16_t, EPI_BIAS_GELU) }
      else { BY_ABL(float, EPI_BIAS_RES) }
    };
    printf("%-5s M=%d N=%d K=%d:", s.name, M, s.N, s.K);
    for (int abl = 0; abl < 4; ++abl) {
      double ms = time_ms([&] { run(abl); }, 20);
      printf("  abl%d %.1f us %.0f TF", abl, ms * 1e3, fl / ms / 1e9);
    }
    {
      auto runx = [&](int abl) {
#define LAUNCHX(TO, EPI, ABL) hipLaunchKernelGGL((gemm_tn_x16_kernel<TO, EPI, ABL>), dim3(g.x < 256 ? g.x : 256), b, 0, 0, X, W, bias, R, (TO*)Y, s.N, s.K, mtot, n_tiles, LnFuse{})
#define BY_ABLX(TO, EPI) switch (abl) { case 0: LAUNCHX(TO, EPI, 0); break; case 1: LAUNCHX(TO, EPI, 1); break; case 2: LAUNCHX(TO, EPI, 2); break; }
        if (s.epi == EPI_BIAS) { BY_ABLX(bf16_t, EPI_BIAS) }
        else if (s.epi == EPI_BIAS_GELU) { BY_ABLX(bf16_t, EPI_BIAS_GELU) }
        else { BY_ABLX(float, EPI_BIAS_RES) }
      };
      for (int abl = 0; abl < 3; ++abl) {
        double ms = time_ms([&] { runx(abl); }, 20);
        printf("  | x16_%d %.1f us %.0f TF", abl, ms * 1e3, fl / ms / 1e9);
      }
      // x16 vs big: same products, different summation order inside a K-step -> compare numerically
      const size_t n_out = (size_t)M * s.N;
      const size_t bytes = n_out * (s.f32out ? 4 : 2);
      std::vector<char> y1(bytes), y2(bytes);
      CK(hipMemset(Y, 0, bytes)); run(0); CK(hipDeviceSynchronize());
      CK(hipMemcpy(y1.data(), Y, bytes, hipMemcpyDeviceToHost));
      CK(hipMemset(Y, 0, bytes)); runx(0); CK(hipDeviceSynchronize());
      CK(hipMemcpy(y2.data(), Y, bytes, hipMemcpyDeviceToHost));
      double maxd = 0, maxv = 0;
      for (size_t i = 0; i < n_out; i += 7) {
        float a_ = s.f32out ? ((float*)y1.data())[i] : (float)((bf16_t*)y1.data())[i];
        float b_ = s.f32out ? ((float*)y2.data())[i] : (float)((bf16_t*)y2.data())[i];
        maxd = fmax(maxd, fabs((double)a_ - b_)); maxv = fmax(maxv, fabs((double)a_));
      }
      printf("  | x16 vs big: max|d| %.3g of %.3g", maxd, maxv);
    }
    // v1 128x128
    {
      dim3 g1((M / 128) * (s.N / 128)), b1(256);
      double ms = time_ms([&] {
        if (s.epi == EPI_BIAS) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, bf16_t, EPI_BIAS>), g1, b1, 0, 0, X, W, bias, R, (bf16_t*)Y, s.N, s.K, mtot, s.N / 128);
        else if (s.epi == EPI_BIAS_GELU) hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, bf16_t, EPI_BIAS_GELU>), g1, b1, 0, 0, X, W, bias, R, (bf16_t*)Y, s.N, s.K, mtot, s.N / 128);
        else hipLaunchKernelGGL((gemm_tn_kernel<bf16_t, float, EPI_BIAS_RES>), g1, b1, 0, 0, X, W, bias, R, (float*)Y, s.N, s.K, mtot, s.N / 128);
      }, 20);
      printf("  | v1 %.1f us %.0f TF", ms * 1e3, fl / ms / 1e9);
    }
    printf("\n");
  }
  CK(hipDeviceSynchronize());
  return 0;
}
