// bf16p_probe — A/B timing of the bf16 wide-tile convolution's two forms (conv_bf16t_kernel: two workgroups per CU, every
// wave stages + multiplies; conv_bf16p_kernel: persistent, 4 producer + 4 consumer waves) on the shapes of the 128^3 / 64^3
// levels of the bf16 storage mode, plus probe copies of the persistent kernel with parts switched off and its wall-clock
// timeline (development probe, not part of the library).  Random data; timing only, nothing is checked.
// Build: bash tools/build_bf16p_probe.sh      Usage: bf16p_probe [iters=20] [nshapes]
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../holo_diffusion_amd/csrc/holo_kernels.h"
namespace holo {
void set_error(const char* fmt, ...) {
  va_list a;
  va_start(a, fmt);
  vprintf(fmt, a);
  va_end(a);
  printf("\n");
}
int conv_bf16p_launch_tl(const ConvParams& p, void* stream);
int conv_bf16p_launch_p1(const ConvParams& p, void* stream);
int conv_bf16p_launch_p2(const ConvParams& p, void* stream);
int conv_bf16p_launch_p3(const ConvParams& p, void* stream);
int conv_bf16p_launch_p4(const ConvParams& p, void* stream);
int conv_bf16p_launch_p7(const ConvParams& p, void* stream);
int conv_bf16p_launch_p8(const ConvParams& p, void* stream);
int conv_bf16p_launch_p16(const ConvParams& p, void* stream);
int conv_bf16p_launch_p128(const ConvParams& p, void* stream);
int conv_bf16p_launch_p256(const ConvParams& p, void* stream);
int conv_bf16p_launch_p512(const ConvParams& p, void* stream);
int conv_bf16p_launch_prio(const ConvParams& p, void* stream);
}  // namespace holo
using namespace holo;
#define CK(x)                                                               \
  do {                                                                      \
    hipError_t e = (x);                                                     \
    if (e != hipSuccess) {                                                  \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);       \
      exit(1);                                                              \
    }                                                                       \
  } while (0)

static uint16_t bf16_of(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static uint16_t* dev_random_bf16(size_t n, float scale) {
  std::vector<uint16_t> h(n);
  for (auto& x : h) x = bf16_of((rand() % 2001 - 1000) * 1e-3f * scale);
  uint16_t* d;
  CK(hipMalloc(&d, n * 2));
  CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}
static float* dev_random(size_t n, float scale) {
  std::vector<float> h(n);
  for (auto& x : h) x = (rand() % 2001 - 1000) * 1e-3f * scale;
  float* d;
  CK(hipMalloc(&d, n * 4));
  CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}
struct Shape {
  int R, C0, C1, Cout, act, epi, skip, ups;
  const char* what;
};

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const int nshapes = argc > 2 ? atoi(argv[2]) : 100;
  const Shape shapes[] = {
      {128, 64, 0, 64, 1, 0, 0, 0, "128^3 64->64 GN+SiLU (ResBlock conv1)"},
      {128, 64, 0, 64, 1, 1, 0, 0, "128^3 64->64 GN+SiLU + residual/bias/stats (ResBlock conv2)"},
      {128, 32, 0, 64, 0, 1, 0, 0, "128^3 32->64 plain (input conv)"},
      {128, 64, 64, 64, 1, 0, 0, 0, "128^3 (64+64)->64 concat, GN+SiLU (up-path conv1)"},
      {128, 64, 0, 64, 1, 1, 1, 0, "128^3 64->64 + fused 1x1x1 skip of (64+64) (up-path conv2)"},
      {128, 64, 0, 32, 1, 0, 0, 0, "128^3 64->32 GN+SiLU (output conv)"},
      {64, 64, 0, 64, 1, 1, 0, 0, "64^3 64->64 GN+SiLU + epilogue"},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int shape_no = 0;
  for (const Shape& s : shapes) {
    if (shape_no++ >= nshapes) break;
    const int R = s.R, Cin = s.C0 + s.C1, Cout = s.Cout;
    const int64_t V = (int64_t)R * R * R;
    const int CinP = (Cin + 31) / 32 * 32, CoutP = Cout >= 64 ? (Cout + 63) / 64 * 64 : 32;
    ConvParams p{};
    p.src0 = (const float*)dev_random_bf16(V * s.C0, 1.f);
    p.C0 = s.C0;
    if (s.C1) {
      p.src1 = (const float*)dev_random_bf16(V * s.C1, 1.f);
      p.C1 = s.C1;
    }
    p.N = 1;
    p.ID = p.IH = p.IW = p.OD = p.OH = p.OW = R;
    p.stride = 1, p.pad = 1, p.ksz = 3;
    p.Cout = Cout, p.CoutP = CoutP, p.CinP = CinP;
    p.bf16 = 1, p.in_bf16 = 1, p.res_bf16 = 1, p.out_bf16 = 1;
    p.w = dev_random(64, 0.1f);  // (not read on this path)
    p.w_bf = dev_random_bf16(64, 0.1f);
    p.w_bft = dev_random_bf16((size_t)27 * CinP * CoutP, 0.05f);
    CK(hipMalloc((void**)&p.out, V * Cout * 2));
    if (s.act) {
      std::vector<float> hc((size_t)Cin * 2);
      for (int c = 0; c < Cin; ++c) hc[2 * c] = 1.0f + 0.01f * (c % 7), hc[2 * c + 1] = 0.01f * (c % 5);
      float* coef;
      CK(hipMalloc(&coef, hc.size() * 4));
      CK(hipMemcpy(coef, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
      p.coef = coef;
      p.act = 1;
    }
    if (s.epi) {
      p.residual = (const float*)dev_random_bf16(V * Cout, 1.f);
      p.bias = dev_random(Cout, 1.f);
      double* st;
      CK(hipMalloc(&st, (size_t)(V / 512) * Cout * 2 * 8));
      p.stats = st;
    }
    if (s.skip) {
      p.skip_src0 = (const float*)dev_random_bf16(V * 64, 1.f);
      p.skip_src1 = (const float*)dev_random_bf16(V * 64, 1.f);
      p.skip_C0 = p.skip_C1 = 64;
      p.skip_CinP = 128;
      p.skip_w = dev_random(64, 0.1f);
      p.skip_w_bf = dev_random_bf16(64, 0.1f);
      p.skip_w_bft = dev_random_bf16((size_t)128 * CoutP, 0.05f);
      p.skip_bias = dev_random(Cout, 1.f);
    }
    printf("%s\n", s.what);
    auto time_fn = [&](const char* what, int (*fn)(const ConvParams&, void*), const ConvParams& q) {
      for (int i = 0; i < 60; ++i)
        if (fn(q, nullptr)) exit(1);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) fn(q, nullptr);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      const double fl = conv_flops(q);
      printf("   %-52s %8.1f us  %7.1f TF/s = %.3f of the bf16 pipe (grid %d)\n", what, ms * 1e3, fl / ms * 1e-9,
             fl / ms * 1e-9 / 2516.0, q.grid_x);
      return ms;
    };
    setenv("HOLO_CONV_BF16P", "0", 1);
    ConvParams qt = p;
    conv_plan(qt, 256);
    if (!qt.bf16t || qt.bf16p || qt.nsplit != 1) {
      printf("   planner: bf16t %d bf16p %d nsplit %d - skipped\n", qt.bf16t, qt.bf16p, qt.nsplit);
      continue;
    }
    time_fn("conv_bf16t_kernel (2 workgroups per CU)", conv_launch, qt);
    setenv("HOLO_CONV_BF16P", "1", 1);
    ConvParams qp = p;
    conv_plan(qp, 256);
    if (!qp.bf16p) {
      printf("   planner did not choose the persistent form\n");
      continue;
    }
    time_fn("conv_bf16p_kernel (persistent, producers + consumers)", conv_launch, qp);
    {
      unsigned long long* dbg;
      CK(hipMalloc(&dbg, (size_t)qp.grid_x * 64));
      CK(hipMemset(dbg, 0, (size_t)qp.grid_x * 64));
      ConvParams q = qp;
      q.dbg = dbg;
      for (int i = 0; i < 20; ++i) conv_bf16p_launch_tl(q, nullptr);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> d((size_t)q.grid_x * 8);
      CK(hipMemcpy(d.data(), dbg, d.size() * 8, hipMemcpyDeviceToHost));
      double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < q.grid_x; ++i)
        for (int j = 0; j < 8; ++j) sum[j] += (double)d[(size_t)i * 8 + j];
      const double items = std::max(sum[4], 1.0), steps = std::max(sum[3], 1.0);
      printf("       timeline (10 ns ticks -> us), consumer wave 0 per item: barrier wait %.2f, tap loops %.2f, epilogue %.2f "
             "(%.1f steps per item; per step: wait %.2f, taps %.2f) | producer wave per step: work %.2f (of it weights request + halo commit %.2f), barrier wait %.2f\n",
             sum[0] / items * 0.01, sum[1] / items * 0.01, sum[2] / items * 0.01, steps / items, sum[0] / steps * 0.01,
             sum[1] / steps * 0.01, sum[5] / steps * 0.01, sum[7] / steps * 0.01, sum[6] / steps * 0.01);
      CK(hipFree(dbg));
    }
    struct { const char* what; int (*fn)(const ConvParams&, void*); } probes[] = {
        {"probe: consumers read no weights", conv_bf16p_launch_p1}, {"probe: consumers read no A operands", conv_bf16p_launch_p2},
        {"probe: neither (MFMAs + barriers + producers)", conv_bf16p_launch_p3},
        {"probe: producers stage nothing", conv_bf16p_launch_p4},
        {"probe: MFMAs + barriers only", conv_bf16p_launch_p7}, {"probe: everything but the MFMAs", conv_bf16p_launch_p8},
        {"probe: producers without the activation", conv_bf16p_launch_p16},
        {"probe: producers load the halo, write nothing", conv_bf16p_launch_p128},
        {"probe: producers write + compute, load nothing", conv_bf16p_launch_p256},
        {"probe: weights not staged (flags and polls stay)", conv_bf16p_launch_p512},
        {"variant: producers and consumers in turns (P_PHASED = 1)", conv_bf16p_launch_prio}};
    for (auto& pr : probes) time_fn(pr.what, pr.fn, qp);
  }
  return 0;
}
