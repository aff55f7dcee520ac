// coexec_probe — do MFMA and ordinary VALU instructions of DIFFERENT waves on one SIMD overlap on gfx950?
// One 512-thread workgroup per CU (waves w and w+4 share a SIMD).  mode 1: waves 0-3 run dependent MFMA chains,
// mode 2: waves 4-7 run dependent FMA chains, mode 3: both.  T(3) ~ max(T1,T2) => co-execution, ~ T1+T2 => serialised.
// Build: hipcc -O2 --offload-arch=gfx950 tools/coexec_probe.cpp -o tools/coexec_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void probe(int mode, int iters, float* out) {
  __shared__ float pad[24 * 1024];  // 96 KB: one workgroup per CU
  const int wave = threadIdx.x >> 6;
  pad[threadIdx.x] = 0.f;
  float r = 0.f;
  if (wave < 4 && (mode & 1)) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float a = threadIdx.x * 1e-3f, b = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) r += acc[i];
  } else if (wave >= 4 && (mode & 2)) {
    float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 64; ++k) {  // 256 FMAs per iteration, 4 independent chains
        x0 = fmaf(x0, 1.0001f, 0.5f);
        x1 = fmaf(x1, 1.0001f, 0.5f);
        x2 = fmaf(x2, 1.0001f, 0.5f);
        x3 = fmaf(x3, 1.0001f, 0.5f);
      }
    }
    r = x0 + x1 + x2 + x3;
  }
  if (r == 12345.678f) out[threadIdx.x] = r + pad[threadIdx.x];
}
int main() {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode = 1; mode <= 3; ++mode) {
    probe<<<256, 512>>>(mode, 100, out);
    hipDeviceSynchronize();
    hipEventRecord(e0); probe<<<256, 512>>>(mode, iters, out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d (%s): %.3f ms  [MFMA: %d x 16 x 64 cyc = %.1f Mcyc; VALU: %d x 256 x 4 cyc = %.1f Mcyc]\n", mode,
           mode == 1 ? "MFMA waves only" : mode == 2 ? "VALU waves only" : "both", ms, iters, iters * 16 * 64 / 1e6, iters,
           iters * 256 * 4 / 1e6);
  }
  return 0;
}
