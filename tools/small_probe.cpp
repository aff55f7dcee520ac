// small_probe — what a weight-streaming launch of the deepest UNet level costs (development probe, not part of the library):
// the 4^3 512 -> 512 3x3x3 convolution (M = 64 rows, 27.6 MB of weights; conv_small_kernel + splitk_reduce_kernel) with the
// weights COLD (NW weight sets cycled: 16 x 27.6 MB pass the memory-side cache) or hot (one set), next to a kernel that does
// nothing but read the same bytes with the same workgroup count.
// Build: hipcc -O3 --offload-arch=gfx950 tools/small_probe.cpp holo_diffusion_amd/csrc/kernels_conv.o \
//              holo_diffusion_amd/csrc/kernels_conv3.o holo_diffusion_amd/csrc/kernels_misc.o -o tools/small_probe
// Usage: small_probe [Cin=512] [Cout=512] [R=4] [iters=64]
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
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
using namespace holo;
#define CK(x)                                                         \
  do {                                                                \
    hipError_t e = (x);                                               \
    if (e != hipSuccess) {                                            \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                        \
    }                                                                 \
  } while (0)

// every workgroup reads its contiguous share of `bytes` (16-byte loads, `depth` of them in flight per thread), one write
template <int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ src, size_t n16, float* __restrict__ out) {
  const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n16 ? b0 + per : n16;
  float s = 0.f;
  for (size_t i = b0 + threadIdx.x; i < b1; i += 256 * DEPTH) {
    float4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v[d] = i + (size_t)d * 256 < b1 ? src[i + (size_t)d * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) s += v[d].x + v[d].y + v[d].z + v[d].w;
  }
  if (s == 123.456f) out[blockIdx.x] = s;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int Cin = argc > 1 ? atoi(argv[1]) : 512, Cout = argc > 2 ? atoi(argv[2]) : 512, R = argc > 3 ? atoi(argv[3]) : 4;
  const int iters = argc > 4 ? atoi(argv[4]) : 64;
  const int64_t V = (int64_t)R * R * R;
  const int CinP = (Cin + 31) / 32 * 32, CoutP = (Cout + 63) / 64 * 64;
  const size_t wfloats = (size_t)27 * CinP * CoutP;
  const int NW = 16;
  std::vector<float*> w(NW);
  {
    std::vector<float> hw(wfloats);
    for (auto& x : hw) x = (rand() % 2001 - 1000) * 1e-4f;
    for (int i = 0; i < NW; ++i) {
      CK(hipMalloc(&w[i], wfloats * 4));
      CK(hipMemcpy(w[i], hw.data(), wfloats * 4, hipMemcpyHostToDevice));
    }
  }
  float *src, *out, *coef, *res, *bias;
  double* stats;
  CK(hipMalloc(&src, V * Cin * 4));
  CK(hipMalloc(&out, V * Cout * 4));
  CK(hipMalloc(&res, V * Cout * 4));
  CK(hipMemset(res, 0, V * Cout * 4));
  CK(hipMalloc(&bias, Cout * 4));
  CK(hipMemset(bias, 0, Cout * 4));
  CK(hipMalloc(&stats, (size_t)64 * Cout * 16));
  {
    std::vector<float> h(V * Cin);
    for (auto& x : h) x = (rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> hc((size_t)Cin * 2);
    for (int c = 0; c < Cin; ++c) hc[2 * c] = 1.f + 0.01f * (c % 7), hc[2 * c + 1] = 0.01f * (c % 5);
    CK(hipMalloc(&coef, hc.size() * 4));
    CK(hipMemcpy(coef, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  }
  ConvParams p{};
  p.src0 = src, p.C0 = Cin, p.N = 1, p.ID = p.IH = p.IW = p.OD = p.OH = p.OW = R, p.stride = 1, p.pad = 1, p.ksz = 3;
  p.Cout = Cout, p.w = w[0], p.CoutP = CoutP, p.CinP = CinP, p.out = out, p.coef = coef, p.act = 1, p.bias = bias;
  const size_t sb = conv_plan(p, 256);
  if (sb) CK(hipMalloc((void**)&p.partial, sb));
  printf("%d^3 %d -> %d: mode %d, split-K %d x %d chunks, partials %.1f MB, weights %.1f MB\n", R, Cin, Cout, p.mode, p.nsplit,
         p.chunks_per_split, sb / 1e6, wfloats * 4 / 1e6);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int cold = 1; cold >= 0; --cold) {
    for (int i = 0; i < 8; ++i) {
      p.w = w[cold ? i % NW : 0];
      if (conv_launch(p, nullptr)) exit(1);
    }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) {
      p.w = w[cold ? i % NW : 0];
      conv_launch(p, nullptr);
    }
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("conv + reduce, weights %s: %.2f us per launch pair\n", cold ? "cold (16 sets cycled)" : "hot (one set)", ms * 1e3 / iters);
  }
  // pure reads of the same bytes: 432 / 512 / 1024 workgroups
  float* dummy;
  CK(hipMalloc(&dummy, 4096 * 4));
  const size_t n16 = wfloats / 4;
  for (int wgs : {256, 432, 512, 1024, 2048}) {
    for (int depth : {4, 16}) {
      for (int cold = 1; cold >= 0; --cold) {
        auto launch = [&](int i) {
          const float4* s4 = reinterpret_cast<const float4*>(w[cold ? i % NW : 0]);
          if (depth == 4)
            hipLaunchKernelGGL(stream_kernel<4>, dim3(wgs), dim3(256), 0, nullptr, s4, n16, dummy);
          else
            hipLaunchKernelGGL(stream_kernel<16>, dim3(wgs), dim3(256), 0, nullptr, s4, n16, dummy);
        };
        for (int i = 0; i < 8; ++i) launch(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) launch(i);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        printf("stream %4d wgs, %2d loads in flight per thread, %s: %.2f us = %.2f TB/s\n", wgs, depth, cold ? "cold" : "hot ", us,
               wfloats * 4 / us / 1e6);
      }
    }
  }
  return 0;
}
