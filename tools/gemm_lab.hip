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

// register-only MFMA rate against the ORDER in which operands change between consecutive instructions (the rate is power-
// limited on random data, so operand toggling is a cost): PAT 0 = A changes every instruction, B every 8th; 1 = B changes
// every instruction, A every 8th; 2 = both change every instruction; 3 = A every 4th, B every 16th (32 accumulators)
template <int PAT>
__global__ __launch_bounds__(512, 1) void mfma_order_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int iters) {
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
      for (int t = 0; t < 32; ++t) {
        const int ia = PAT == 0 ? (t & 3) : PAT == 1 ? ((t >> 3) & 3) : PAT == 2 ? (t & 3) : ((t >> 2) & 3);
        const int ib = PAT == 0 ? ((t >> 3) & 3) : PAT == 1 ? (t & 3) : PAT == 2 ? ((t + 1) & 3) : ((t >> 4) & 3);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ia], b[(ib + kc) & 3], acc[t], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 32; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
  dst[blockIdx.x * blockDim.x + threadIdx.x] = s + lane;
}

// register-only MFMA at reduced DUTY: after every 16 MFMAs (16 x 16 cycles of pipe time) the wave idles NOPS x 16 cycles
// (both waves of a SIMD do, in the same place): is the data-dependent slowdown still there when the pipe is far from full?
template <int NOPS>
__global__ __launch_bounds__(512, 1) void mfma_duty_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int iters) {
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
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int u = 16 * (g & 1) + t;
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(t + g) & 3], b[(t >> 2) & 3], acc[u], 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < NOPS; ++n) asm volatile("s_nop 15");
    }
  }
  float s = 0.f;
  for (int i = 0; i < 32; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
  dst[blockIdx.x * blockDim.x + threadIdx.x] = s + lane;
}

// shader-clock probe: one wave counts s_memtime ticks against the constant 100 MHz s_memrealtime while other kernels load the
// chip (it has to be resident BEFORE the persistent GEMM grid takes every CU's registers)
__global__ void clock_probe_kernel(unsigned long long* out, long long wall_ticks) {
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long w0 = wall_clock64();
  while ((long long)(wall_clock64() - w0) < wall_ticks) __builtin_amdgcn_s_sleep(64);
  out[0] = __builtin_readcyclecounter() - c0;
  out[1] = wall_clock64() - w0;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536;
  const char* data = argc > 2 ? argv[2] : "randn";      // randn | zero | small (values in {-1,0,1}): data-dependent power
  struct Shape { const char* name; int N, K; Epilogue epi; bool f32out; };
  Shape shapes[] = {{"qkv", 2304, 768, EPI_BIAS, false}, {"out", 768, 768, EPI_BIAS_RES, true},
                    {"ffn1", 3072, 768, EPI_BIAS_GELU, false}, {"ffn2", 768, 3072, EPI_BIAS_RES, true},
                    {"ffn1h", 1536, 768, EPI_BIAS_GELU, false}, {"qkvh", 1280, 768, EPI_BIAS, false}, {"qkv1", 256, 768, EPI_BIAS, false}};
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  const size_t maxe = (size_t)M * 3072;
  std::vector<bf16_t> h(maxe);
  for (size_t i = 0; i < maxe; ++i)
    h[i] = data[0] == 'z' ? (bf16_t)0.f : data[0] == 's' ? (bf16_t)(float)((int)(rng() % 3) - 1) : (bf16_t)nd(rng);
  printf("operand data: %s\n", data);
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
  {
    const int iters = 2000;
    // two waves per SIMD: 16 MFMAs of each = 512 cycles of pipe time per group; NOPS x 16 idle cycles per wave and group
    const int nops[5] = {0, 8, 16, 32, 64};
    for (int v = 0; v < 5; ++v) {
      double ms = time_ms([&] {
        if (v == 0) hipLaunchKernelGGL(mfma_duty_kernel<0>, dim3(256), dim3(512), 0, 0, X, (float*)Y, iters);
        else if (v == 1) hipLaunchKernelGGL(mfma_duty_kernel<8>, dim3(256), dim3(512), 0, 0, X, (float*)Y, iters);
        else if (v == 2) hipLaunchKernelGGL(mfma_duty_kernel<16>, dim3(256), dim3(512), 0, 0, X, (float*)Y, iters);
        else if (v == 3) hipLaunchKernelGGL(mfma_duty_kernel<32>, dim3(256), dim3(512), 0, 0, X, (float*)Y, iters);
        else hipLaunchKernelGGL(mfma_duty_kernel<64>, dim3(256), dim3(512), 0, 0, X, (float*)Y, iters);
      }, 10);
      const double fl = 256.0 * 8 * iters * 64 * (2.0 * 16 * 16 * 32);
      // per group and SIMD: max(2 x 256 MFMA cycles, 256 + NOPS x 16) cycles if a wave's idle time hides under its partner's MFMAs
      printf("mfma_duty nops %2d x16 cycles per 16 MFMAs: %.1f us  %.0f TF\n", nops[v], ms * 1e3, fl / ms / 1e9);
    }
  }
  for (int threads : {512, 256}) {
    const int iters = 2000;
    for (int pat = 0; pat < 4; ++pat) {
      double ms = time_ms([&] {
        if (pat == 0) hipLaunchKernelGGL(mfma_order_kernel<0>, dim3(256), dim3(threads), 0, 0, X, (float*)Y, iters);
        else if (pat == 1) hipLaunchKernelGGL(mfma_order_kernel<1>, dim3(256), dim3(threads), 0, 0, X, (float*)Y, iters);
        else if (pat == 2) hipLaunchKernelGGL(mfma_order_kernel<2>, dim3(256), dim3(threads), 0, 0, X, (float*)Y, iters);
        else hipLaunchKernelGGL(mfma_order_kernel<3>, dim3(256), dim3(threads), 0, 0, X, (float*)Y, iters);
      }, 10);
      const double fl = 256.0 * (threads / 64) * iters * 64 * (2.0 * 16 * 16 * 32);
      printf("mfma_order pat %d, %d waves/CU: %.1f us  %.0f TF\n", pat, threads / 64, ms * 1e3, fl / ms / 1e9);
    }
  }
  {
    unsigned long long* probe; CK(hipMalloc(&probe, 16));
    hipStream_t sp, sw; CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sw, hipStreamNonBlocking));
    int wall_khz = 0; CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    float2* mr0; CK(hipMalloc(&mr0, (size_t)M * 8)); CK(hipMemset(mr0, 0, (size_t)M * 8));
    for (int load = 0; load < 4; ++load) {       // 0 idle, 1 register-only MFMA, 2 FFN1-shaped GEMM, 3 FFN2-shaped GEMM
      const long long ticks = (long long)wall_khz * 20;              // 20 ms
      hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, sp, probe, ticks);
      for (int rep = 0; rep < (load == 1 ? 12 : 90) && load; ++rep) {
        if (load == 1) hipLaunchKernelGGL(mfma_peak16_kernel, dim3(256), dim3(512), 0, sw, X, (float*)Y, 2000);
        else {
          const int N = load == 2 ? 3072 : 768, K = load == 2 ? 768 : 3072;
          hipLaunchKernelGGL((gemm_tn_x16_kernel<bf16_t, bf16_t, EPI_BIAS, 0>), dim3(256), dim3(512), 0, sw, X, W, bias, R, (bf16_t*)Y, N, K, mtot,
                             N / G_BN, DlnAux{});
        }
      }
      CK(hipDeviceSynchronize());
      unsigned long long hp[2]; CK(hipMemcpy(hp, probe, 16, hipMemcpyDeviceToHost));
      printf("clock probe under load %d (%s): s_memtime %llu ticks over %.2f ms of s_memrealtime (%d kHz) => %.1f MHz\n", load,
             load == 0 ? "idle" : load == 1 ? "mfma 16x16x32 registers only" : load == 2 ? "gemm N=3072 K=768" : "gemm N=768 K=3072",
             hp[0], hp[1] * 1.0 / wall_khz, wall_khz, hp[0] * 1e-3 / (hp[1] * 1.0 / wall_khz));
    }
  }
  // the production 256x256 kernel, every epilogue it runs in the bf16 encoder; abl1 = main loop only
  float2* mr; float2* part; float* vec;
  CK(hipMalloc(&mr, (size_t)M * 8)); CK(hipMalloc(&part, (size_t)M * 16 * 8)); CK(hipMalloc(&vec, 3072 * 4));
  {
    std::vector<float2> hm(M, float2{0.01f, 1.0f});
    CK(hipMemcpy(mr, hm.data(), (size_t)M * 8, hipMemcpyHostToDevice));
    std::vector<float> hv(3072, 1.0f);
    CK(hipMemcpy(vec, hv.data(), 3072 * 4, hipMemcpyHostToDevice));
  }
  struct Case { const char* name; int N, K; Epilogue epi; };
  Case cases[] = {{"qkv  NORM", 2304, 768, EPI_NORM}, {"qkv  BIAS", 2304, 768, EPI_BIAS}, {"out  NRES", 768, 768, EPI_NRES},
                  {"out  BIAS", 768, 768, EPI_BIAS}, {"ffn1 NORM_GELU", 3072, 768, EPI_NORM_GELU}, {"ffn1 BIAS", 3072, 768, EPI_BIAS},
                  {"ffn2 NRES", 768, 3072, EPI_NRES}, {"ffn2 BIAS", 768, 3072, EPI_BIAS}};
  for (auto& s : cases) {
    const int n_tiles = s.N / G_BN;
    const int tiles = (M / G_BM) * n_tiles;
    dim3 g(tiles < 256 ? tiles : 256), b(512);
    const double fl = 2.0 * M * s.N * s.K;
    int cg = 0;
    DlnAux aux{vec, mr, part, M, cg};
    bf16_t* y = (bf16_t*)Y;
    auto run = [&](int abl) {
#define LX(EPI, ABL) hipLaunchKernelGGL((gemm_tn_x16_kernel<bf16_t, bf16_t, EPI, ABL>), g, b, 0, 0, X, W, bias, EPI == EPI_NRES ? y : R, y, s.N, s.K, mtot, n_tiles, aux)
#define BYA(EPI) if (abl == 0) LX(EPI, 0); else if (abl == 1) LX(EPI, 1); else if (abl == 2) LX(EPI, 2); else if (abl == 3) LX(EPI, 3); else if (abl == 4) LX(EPI, 4); else if (abl == 5) LX(EPI, 5); else LX(EPI, 6);
      switch (s.epi) {
        case EPI_NORM: BYA(EPI_NORM) break;
        case EPI_NORM_GELU: BYA(EPI_NORM_GELU) break;
        case EPI_NRES: BYA(EPI_NRES) break;
        default: BYA(EPI_BIAS) break;
      }
    };
    printf("%-15s M=%d N=%d K=%d tiles/CU %.1f:", s.name, M, s.N, s.K, tiles / 256.0);
    for (int c : {0, 2, 3, 4, 6}) {
      cg = c;
      aux.col_group = c;
      double ms = time_ms([&] { run(0); }, 20);
      printf("  cg%d %.1f us %.0f TF", c, ms * 1e3, fl / ms / 1e9);
    }
    cg = 0; aux.col_group = 0;
    for (int abl : {1, 3}) {
      if (abl == 3 && s.epi != EPI_NRES) continue;
      double ms = time_ms([&] { run(abl); }, 20);
      printf("  abl%d %.1f us %.0f TF (%.1f us/tile)", abl, ms * 1e3, fl / ms / 1e9, ms * 1e3 / (tiles / 256.0));
    }
    printf("\n");
  }
  return 0;
}
