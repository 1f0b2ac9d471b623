// Development lab (round 6): the PAIRED 4-wave GEMM (gemm_tn_p4_kernel: two independent 4-wave workgroups per CU, 256 x 128 tiles, one
// LDS stage each — tools/gen_gemm_p4.py) against the production 8-wave kernel (gemm_tn_w8_kernel) on the four layer shapes, same box,
// same process, alternating: K-loops only (ABL = 1) and with each deferred-LayerNorm epilogue; `check` compares the outputs bit for bit.
// Not part of libmanner_hip.so.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Imanner_amd/csrc tools/gemm_p4_lab.hip -o tools/bin/gemm_p4_lab
//   tools/bin/gemm_p4_lab [M=65536] [check|time] [f16|bf16]         P4_DEPHASE=n: sleeps of the second workgroup of a CU (default K / 256)
#include <stdio.h>
#include <string.h>
#include <vector>
#include <random>
#define MANNER_P4_LAB 1
#include "../manner_amd/csrc/gemm.hip"
namespace manner { int fail(int code, const char* fmt, ...) { fprintf(stderr, "fail %d: %s\n", code, fmt); return code; } }
using namespace manner;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F>
static double time_us(F f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return 1e3 * ms / iters;
}

struct Shape { const char* name; int N, K; };

template <typename TE>
static int run(int M, bool check) {
  const Shape all_shapes[] = {{"qkv", 2304, 768}, {"out", 768, 768}, {"ffn1", 3072, 768}, {"ffn2", 768, 3072}};
  std::vector<Shape> shapes;
  for (const auto& s : all_shapes) if (!getenv("P4_SHAPE") || !strcmp(getenv("P4_SHAPE"), s.name)) shapes.push_back(s);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  const size_t xe = (size_t)M * 3072, we = (size_t)3072 * 3072;
  std::vector<TE> hx(xe), hw(we);
  for (auto& v : hx) v = (TE)nd(rng);
  for (auto& v : hw) v = (TE)(0.05f * nd(rng));
  TE *X, *W, *Ya, *Yb, *R0; float *bias, *vec; float2 *mr, *parta, *partb; int* mtot;
  CK(hipMalloc(&X, xe * 2)); CK(hipMalloc(&W, we * 2)); CK(hipMalloc(&bias, 3072 * 4)); CK(hipMalloc(&vec, 3072 * 4)); CK(hipMalloc(&mtot, 4));
  CK(hipMalloc(&Ya, (size_t)M * 3072 * 2)); CK(hipMalloc(&Yb, (size_t)M * 3072 * 2)); CK(hipMalloc(&R0, (size_t)M * 768 * 2));
  CK(hipMalloc(&mr, (size_t)M * 8)); CK(hipMalloc(&parta, (size_t)M * 8 * 48)); CK(hipMalloc(&partb, (size_t)M * 8 * 48));
  CK(hipMemcpy(X, hx.data(), xe * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, hw.data(), we * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(R0, hx.data(), (size_t)M * 768 * 2, hipMemcpyHostToDevice));
  {
    std::vector<float> hb(3072), hv(3072);
    for (auto& v : hb) v = 0.1f * nd(rng);
    for (auto& v : hv) v = 1.f + 0.1f * nd(rng);
    CK(hipMemcpy(bias, hb.data(), 3072 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vec, hv.data(), 3072 * 4, hipMemcpyHostToDevice));
    std::vector<float2> hm(M);
    for (auto& v : hm) v = float2{0.05f * nd(rng), 1.f + 0.05f * nd(rng)};
    CK(hipMemcpy(mr, hm.data(), (size_t)M * 8, hipMemcpyHostToDevice));
  }
  CK(hipMemcpy(mtot, &M, 4, hipMemcpyHostToDevice));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int dephase_env = getenv("P4_DEPHASE") ? atoi(getenv("P4_DEPHASE")) : -1;
  const int ranges = getenv("P4_RANGES") ? atoi(getenv("P4_RANGES")) : 1;
  int bad = 0;
  {
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm_tn_p4_kernel<TE, TE, EPI_NORM_GELU>, 256, 0));
    printf("occupancy of gemm_tn_p4_kernel: %d workgroups per CU\n", occ);
  }
  const int iters = getenv("LAB_ITERS") ? atoi(getenv("LAB_ITERS")) : 20, reps = check ? 1 : (getenv("LAB_REPS") ? atoi(getenv("LAB_REPS")) : 3);
  for (int rep = 0; rep < reps; ++rep)
    for (const auto& s : shapes) {
      const int n_tiles = s.N / 256, tiles = (M / 256) * n_tiles;
      const double fl = 2.0 * M * s.N * s.K;
      const int wgs = getenv("P4_WGS_PER_CU") ? atoi(getenv("P4_WGS_PER_CU")) : 2;      // 1: one workgroup per CU (what pairing buys)
      const dim3 g8(tiles < cus ? tiles : cus), g4(2 * tiles < wgs * cus ? 2 * tiles : wgs * cus);
      DlnAux a8{};
      a8.vec = vec; a8.mr = mr; a8.part_stride = M; a8.panel_mode = 1; a8.x_rows = M; a8.xcd_ranges = ranges;
      a8.col_group = ranges ? pick_col_group(M, s.N, s.K, cus) : 0;
      DlnAux a4 = a8;
      a4.stagger = (dephase_env >= 0 ? dephase_env : s.K / 256) | ((getenv("P4_DEPHASE_MODE") ? atoi(getenv("P4_DEPHASE_MODE")) : 0) << 8);
      a8.part = parta; a4.part = partb;
      const bool nres_ok = s.N == 768;                 // the in-place residual epilogue: N = 768 shapes
      auto w8 = [&](int e, TE* y) {
        if (e == 0) hipLaunchKernelGGL((gemm_tn_w8_kernel<TE, TE, EPI_BIAS, 1>), g8, dim3(512), 0, 0, X, W, bias, (const TE*)nullptr, y, s.N, s.K, mtot, n_tiles, a8);
        else if (e == 1) hipLaunchKernelGGL((gemm_tn_w8_kernel<TE, TE, EPI_NORM>), g8, dim3(512), 0, 0, X, W, bias, (const TE*)nullptr, y, s.N, s.K, mtot, n_tiles, a8);
        else if (e == 2) hipLaunchKernelGGL((gemm_tn_w8_kernel<TE, TE, EPI_NORM_GELU>), g8, dim3(512), 0, 0, X, W, bias, (const TE*)nullptr, y, s.N, s.K, mtot, n_tiles, a8);
        else hipLaunchKernelGGL((gemm_tn_w8_kernel<TE, TE, EPI_NRES>), g8, dim3(512), 0, 0, X, W, bias, (const TE*)y, y, s.N, s.K, mtot, n_tiles, a8);
      };
      auto p4 = [&](int e, TE* y) {
        if (e == 0) hipLaunchKernelGGL((gemm_tn_p4_kernel<TE, TE, EPI_BIAS, 1>), g4, dim3(256), 0, 0, X, W, bias, (const TE*)nullptr, y, s.N, s.K, mtot, n_tiles, a4);
        else if (e == 1) hipLaunchKernelGGL((gemm_tn_p4_kernel<TE, TE, EPI_NORM>), g4, dim3(256), 0, 0, X, W, bias, (const TE*)nullptr, y, s.N, s.K, mtot, n_tiles, a4);
        else if (e == 2) hipLaunchKernelGGL((gemm_tn_p4_kernel<TE, TE, EPI_NORM_GELU>), g4, dim3(256), 0, 0, X, W, bias, (const TE*)nullptr, y, s.N, s.K, mtot, n_tiles, a4);
        else hipLaunchKernelGGL((gemm_tn_p4_kernel<TE, TE, EPI_NRES>), g4, dim3(256), 0, 0, X, W, bias, (const TE*)y, y, s.N, s.K, mtot, n_tiles, a4);
      };
      static const char* en[] = {"K-loops", "NORM", "NORM_GELU", "NRES"};
      if (check) {
        for (int e = 1; e < 4; ++e) {
          if (e == 3 && !nres_ok) continue;
          const size_t bytes = (size_t)M * s.N * 2;
          if (e == 3) { CK(hipMemcpy(Ya, R0, bytes, hipMemcpyDeviceToDevice)); CK(hipMemcpy(Yb, R0, bytes, hipMemcpyDeviceToDevice)); }
          else { CK(hipMemset(Ya, 0xff, bytes)); CK(hipMemset(Yb, 0xee, bytes)); }
          CK(hipMemset(parta, 0, (size_t)M * 8 * 48)); CK(hipMemset(partb, 0, (size_t)M * 8 * 48));
          w8(e, Ya); p4(e, Yb);
          CK(hipDeviceSynchronize());
          std::vector<uint16_t> ha((size_t)M * s.N), hb((size_t)M * s.N);
          CK(hipMemcpy(ha.data(), Ya, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), Yb, bytes, hipMemcpyDeviceToHost));
          size_t diff = 0, first = 0;
          for (size_t i = 0; i < ha.size(); ++i) if (ha[i] != hb[i]) { if (!diff) first = i; ++diff; }
          size_t pdiff = 0;
          if (e == 3) {
            std::vector<float> pa((size_t)M * 2 * 12), pb((size_t)M * 2 * 12);
            CK(hipMemcpy(pa.data(), parta, pa.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(pb.data(), partb, pb.size() * 4, hipMemcpyDeviceToHost));
            pdiff = memcmp(pa.data(), pb.data(), pa.size() * 4) != 0;
          }
          printf("check %-4s %-9s: %zu of %zu outputs differ%s%s  %s\n", s.name, en[e], diff, ha.size(), diff ? " (first at " : "",
                 diff ? (std::to_string(first / s.N) + ", " + std::to_string(first % s.N) + ")").c_str() : "", !diff && !pdiff ? "ok" : (pdiff ? "WRONG (row statistics)" : "WRONG"));
          if (diff || pdiff) bad = 1;
        }
        continue;
      }
      for (int e = 0; e < 4; ++e) {
        if (e == 3 && !nres_ok) continue;
        const double t8 = time_us([&] { w8(e, Ya); }, iters);
        const double t4 = time_us([&] { p4(e, Yb); }, iters);
        printf("rep %d %-4s N=%4d K=%4d %-9s  8-wave %7.1f us %6.0f TF | paired 4-wave %7.1f us %6.0f TF (%+5.1f %%)\n", rep, s.name, s.N, s.K, en[e], t8,
               fl / t8 / 1e6, t4, fl / t4 / 1e6, 100 * (t4 / t8 - 1));
      }
    }
  return bad;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65536;
  const bool check = argc > 2 && !strcmp(argv[2], "check");
  const bool f16 = argc > 3 && !strcmp(argv[3], "f16");
  printf("M = %d, %s, N(0,1) activations; 8-wave = gemm_tn_w8_kernel, paired 4-wave = gemm_tn_p4_kernel (two workgroups per CU)\n", M, f16 ? "f16" : "bf16");
  return f16 ? run<f16_t>(M, check) : run<bf16_t>(M, check);
}
