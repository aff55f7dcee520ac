// conv_ab — A/B timing of the stride-1 3x3x3 convolution kernels on one shape list (development probe, not part of the
// library): conv_wino2_kernel (two workgroups per CU, (z,y) Winograd) against conv_wino3_kernel (one persistent 512-register
// wave per SIMD, F(2x2x2,3x3x3)), random data, planner-chosen split-K.
// Build: bash tools/build_conv_ab.sh
// Usage: conv_ab [iters=20] [nshapes]   CONV_AB_COLD=1: also time every launch behind a 768 MB write (input and weights
//        evicted from the L2s and the memory-side cache: what a layer sees in the middle of a forward pass)
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
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
}  // namespace holo
namespace holo {  // probe copies of kernels_conv3.hip (see its W3_PROBE)
int conv_wino3_launch_p1(const ConvParams& p, void* stream);
int conv_wino3_launch_p2(const ConvParams& p, void* stream);
int conv_wino3_launch_p4(const ConvParams& p, void* stream);
int conv_wino3_launch_p7(const ConvParams& p, void* stream);
int conv_wino3_launch_p8(const ConvParams& p, void* stream);
int conv_wino3_launch_p16(const ConvParams& p, void* stream);
int conv_wino3_launch_p32(const ConvParams& p, void* stream);
int conv_wino3_launch_p64(const ConvParams& p, void* stream);
int conv_wino3_launch_p128(const ConvParams& p, void* stream);
int conv_wino3_launch_tl(const ConvParams& p, void* stream);
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

__global__ void flush_kernel(float4* buf, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = make_float4(v, v, v, v);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const int nshapes = argc > 2 ? atoi(argv[2]) : 100;
  const bool cold = getenv("CONV_AB_COLD") != nullptr;
  float4* flushbuf = nullptr;
  const size_t flush_n = (size_t)768 << 20 >> 4;
  if (cold) CK(hipMalloc(&flushbuf, flush_n * 16));
  const Shape shapes[] = {
      {64, 64, 0, 64, 1, 0, 0, 0, "64^3 64->64 GN+SiLU (ResBlock conv1)"},
      {64, 64, 0, 64, 1, 1, 0, 0, "64^3 64->64 GN+SiLU + residual/bias/stats (ResBlock conv2)"},
      {64, 64, 64, 64, 1, 0, 0, 0, "64^3 (64+64)->64 concat, GN+SiLU (up-path conv1)"},
      {64, 64, 0, 64, 1, 1, 1, 0, "64^3 64->64 + fused 1x1x1 skip of (64+64) (up-path conv2)"},
      {64, 32, 0, 64, 0, 0, 0, 0, "64^3 32->64 plain (input conv)"},
      {64, 64, 0, 64, 0, 0, 0, 1, "64^3 64->64 upsample-on-load (Upsample conv)"},
      {32, 64, 0, 64, 1, 1, 0, 0, "32^3 64->64 GN+SiLU + epilogue"},
      {32, 128, 64, 64, 1, 0, 0, 0, "32^3 (128+64)->64 concat"},
      {16, 128, 0, 128, 1, 1, 0, 0, "16^3 128->128"},
      {16, 256, 128, 128, 1, 0, 0, 0, "16^3 (256+128)->128 concat"},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int shape_no = 0;
  for (const Shape& s : shapes) {
    if (shape_no++ >= nshapes) break;
    const int R = s.R, Cin = s.C0 + s.C1, Cout = s.Cout;
    const int64_t V = (int64_t)R * R * R;
    const int SR = s.ups ? R / 2 : R;
    const int64_t SV = (int64_t)SR * SR * SR;
    const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
    ConvParams p{};
    p.src0 = dev_random(SV * s.C0, 1.f);
    p.C0 = s.C0;
    if (s.C1) {
      p.src1 = dev_random(SV * s.C1, 1.f);
      p.C1 = s.C1;
    }
    p.N = 1;
    p.ID = p.IH = p.IW = p.OD = p.OH = p.OW = R;
    p.ups = s.ups;
    p.stride = 1, p.pad = 1, p.ksz = 3;
    p.Cout = Cout, p.CoutP = CoutP, p.CinP = CinP;
    p.w = dev_random((size_t)27 * CinP * CoutP, 0.1f);
    p.w_wino = p.w_wino2 = dev_random((size_t)48 * CinP * CoutP, 0.1f);
    p.w_wino3 = dev_random((size_t)conv_wino3_weight_floats(CoutP, CinP, 27), 0.1f);
    CK(hipMalloc((void**)&p.out, V * Cout * 4));
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
      p.residual = dev_random(V * Cout, 1.f);
      p.bias = dev_random(Cout, 1.f);
      double* st;
      CK(hipMalloc(&st, (size_t)std::max<int64_t>(V / 64, 256) * Cout * 2 * 8));
      p.stats = st;
    }
    if (s.skip) {
      p.skip_src0 = dev_random(V * 64, 1.f);
      p.skip_src1 = dev_random(V * 64, 1.f);
      p.skip_C0 = p.skip_C1 = 64;
      p.skip_CinP = 128;
      p.skip_w = dev_random((size_t)128 * CoutP, 0.1f);
      p.skip_w_wino = p.skip_w_wino2 = dev_random((size_t)4 * 128 * CoutP, 0.1f);
      p.skip_w_wino3 = dev_random((size_t)conv_wino3_weight_floats(CoutP, 128, 1), 0.1f);
      p.skip_bias = dev_random(Cout, 1.f);
    }
    printf("%s\n", s.what);
    for (int form = 2; form <= 3; ++form) {
      setenv("HOLO_CONV_WINO3", form == 3 ? "1" : "0", 1);
      setenv("HOLO_CONV_WINO3_MIN_ITEMS", "1", 1);
      ConvParams q = p;
      const size_t sb = conv_plan(q, 256);
      if (sb) CK(hipMalloc((void**)&q.partial, sb));
      if (q.wino != form) {
        printf("   form %d: planner chose wino=%d\n", form, q.wino);
        continue;
      }
      // ~50 ms of the same launches first: the clocks of a short burst are 5 - 10 % below the sustained ones (a variant timed
      // later in the process used to look that much faster than the first one)
      for (int i = 0; i < 200; ++i)
        if (conv_launch(q, nullptr)) exit(1);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) conv_launch(q, nullptr);
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      const double fl = conv_flops(q), fx = conv_exec_flops(q);
      printf("   wino%d: %8.1f us  nsplit %d grid %d | algorithmic %6.1f TF/s, issued %6.1f TF/s = %.3f of the fp32 pipe\n", form,
             ms * 1e3, q.nsplit, q.grid_x, fl / ms * 1e-9, fx / ms * 1e-9, fx / ms * 1e-9 / 157.3);
      if (cold) {
        float tot = 0.f;
        for (int i = 0; i < iters; ++i) {
          flush_kernel<<<2048, 256>>>(flushbuf, flush_n, (float)i);
          CK(hipEventRecord(e0, nullptr));
          conv_launch(q, nullptr);
          CK(hipEventRecord(e1, nullptr));
          CK(hipEventSynchronize(e1));
          float t;
          CK(hipEventElapsedTime(&t, e0, e1));
          tot += t;
        }
        printf("          cold (behind a 768 MB write): %8.1f us\n", tot / iters * 1e3);
      }
      if (form == 3) {  // per-workgroup timeline of one launch
        unsigned long long* dbg;
        CK(hipMalloc(&dbg, (size_t)q.grid_x * 64));
        CK(hipMemset(dbg, 0, (size_t)q.grid_x * 64));
        q.dbg = dbg;
        conv_launch(q, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> d((size_t)q.grid_x * 8);
        CK(hipMemcpy(d.data(), dbg, d.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        double pro = 0, tot = 0, items = 0;
        for (int i = 0; i < q.grid_x; ++i) {
          t0 = std::min(t0, d[i * 8]);
          t1 = std::max(t1, d[i * 8 + 3]);
          pro += d[i * 8 + 1] - d[i * 8];
          tot += d[i * 8 + 3] - d[i * 8];
          items += d[i * 8 + 7];
        }
        double bar = 0, epi = 0, work = 0;
        for (int i = 0; i < q.grid_x; ++i) bar += d[i * 8 + 4], epi += d[i * 8 + 5], work += d[i * 8 + 6];
        printf("          timeline: span %.1f us; per workgroup: prologue %.2f us, busy %.1f us, %.1f items -> %.2f us per item "
               "(stage work %.2f + barrier wait %.2f + skip/epilogue %.2f)\n",
               (t1 - t0) * 0.01, pro / q.grid_x * 0.01, tot / q.grid_x * 0.01, items / q.grid_x, (tot - pro) / items * 0.01,
               work / items * 0.01, bar / items * 0.01, epi / items * 0.01);
        {  // shader-clock sums of wave 0 of every workgroup (the W3_TIMELINE copy of the kernel)
          CK(hipFree(dbg));
          CK(hipMalloc(&dbg, (size_t)q.grid_x * 128));
          CK(hipMemset(dbg, 0, (size_t)q.grid_x * 128));
          q.dbg = dbg;
          if (cold) flush_kernel<<<2048, 256>>>(flushbuf, flush_n, 3.f);
          conv_wino3_launch_tl(q, nullptr);
          CK(hipDeviceSynchronize());
          std::vector<unsigned long long> t((size_t)q.grid_x * 16);
          CK(hipMemcpy(t.data(), dbg, t.size() * 8, hipMemcpyDeviceToHost));
          double sum[7] = {0, 0, 0, 0, 0, 0, 0};
          for (int i = 0; i < q.grid_x; ++i)
            for (int j = 0; j < 7; ++j) sum[j] += t[((size_t)q.grid_x + i) * 8 + j];
          const double ns = sum[6];
          printf("          shader cycles per stage (wave 0): 32 x 16 MFMAs + requests %.0f | A-operand clumps (20) %.0f | x-transform clumps (8) "
                 "%.0f | commit clumps (4) %.0f | setup clump %.0f | barrier %.0f\n",
                 sum[0] / ns, sum[1] / ns, sum[2] / ns, sum[3] / ns, sum[4] / ns, sum[5] / ns);
        }
        CK(hipFree(dbg));
        q.dbg = nullptr;
        // probe copies of the kernel: what the stage loop costs without one of its parts
        struct { const char* what; int (*fn)(const ConvParams&, void*); } probes[] = {
            {"no halo requests/commits in the loop", conv_wino3_launch_p1}, {"no weight requests", conv_wino3_launch_p2},
            {"no patch reads / input transforms", conv_wino3_launch_p4}, {"MFMAs only (none of the three)", conv_wino3_launch_p7},
            {"everything but the MFMAs", conv_wino3_launch_p8},
            {"weight requests nobody waits for", conv_wino3_launch_p16},
            {"halo requests but no commits", conv_wino3_launch_p32},
            {"commits without the activation", conv_wino3_launch_p64},
            {"work list in eight XCD lanes", conv_wino3_launch_p128}};
        for (auto& pr : probes) {
          for (int i = 0; i < 100; ++i) pr.fn(q, nullptr);
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0, nullptr));
          for (int i = 0; i < iters; ++i) pr.fn(q, nullptr);
          CK(hipEventRecord(e1, nullptr));
          CK(hipEventSynchronize(e1));
          float pms;
          CK(hipEventElapsedTime(&pms, e0, e1));
          printf("          probe %-40s %8.1f us\n", pr.what, pms / iters * 1e3);
        }
      }
      if (q.partial) CK(hipFree(q.partial));
    }
  }
  return 0;
}
