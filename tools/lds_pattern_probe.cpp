// lds_pattern_probe — cycles per ds_read_b128 for the lane -> address patterns of the kernels' MFMA-operand reads (development
// probe): is a pattern bank-conflict free on this chip, whatever the counters say?
//   0 contiguous (16 bytes per lane)            1 conv_bf16t_kernel's A fragment (32-byte voxels, halves swapped on odd rows)
//   2 the same without the swap                 3 flash_attn_bf16v2_kernel's K fragment (144-byte rows)
//   4 conv_s2_bf16_kernel's A fragment          5 the same stride as 1 with the swap on bit 2 of x instead of the row parity
// Build: hipcc -O2 --offload-arch=gfx950 tools/lds_pattern_probe.cpp -o tools/lds_pattern_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(256) void probe(int iters, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 1024];  // 64 KB
  for (int i = threadIdx.x; i < 16 * 1024; i += 256) lds[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, kg = lane >> 5, ys = li >> 3, x = li & 7;
  int off;  // bytes
  if (PAT == 0) off = lane * 16;
  if (PAT == 1) off = ((2 * wave * 100 + ys * 10 + x) * 32) + ((kg ^ (ys & 1)) * 16);
  if (PAT == 2) off = ((2 * wave * 100 + ys * 10 + x) * 32) + kg * 16;
  if (PAT == 3) off = li * 144 + kg * 16;
  if (PAT == 4) off = (((2 * (wave >> 1) * 17 + 2 * ys) * 18 + x) * 32) + ((kg ^ (ys & 1)) * 16);
  if (PAT == 5) off = ((2 * wave * 100 + ys * 10 + x) * 32) + ((kg ^ ((x >> 2) & 1)) * 16);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned base = (unsigned)(size_t)lds + off;
  for (int it = 0; it < iters; ++it) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(base), "n"(j * 32));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  if (acc[0] == 12345.f) out[threadIdx.x] = acc[1];
}

template <int PAT>
float run(int threads, float* out) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  probe<PAT><<<256, threads>>>(100, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  probe<PAT><<<256, threads>>>(iters, out);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3f * 2.4e9f / (iters * 8.f);  // cycles per ds_read_b128 of one wave (all waves of the CU reading)
}

int main() {
  float* out;
  (void)hipMalloc(&out, 4096);
  const char* names[] = {"contiguous", "conv_bf16t A (halves swapped on odd rows)", "conv_bf16t A without the swap", "attention K (144-byte rows)",
                         "conv_s2 A", "32-byte voxels, swap on bit 2 of x"};
  for (int threads = 64; threads <= 256; threads *= 4) {
    printf("%d wave(s) per CU reading: cycles per ds_read_b128 per wave (nominal 2.4 GHz)\n", threads / 64);
    const float r[] = {run<0>(threads, out), run<1>(threads, out), run<2>(threads, out), run<3>(threads, out), run<4>(threads, out), run<5>(threads, out)};
    for (int i = 0; i < 6; ++i) printf("  %-48s %6.1f\n", names[i], r[i]);
  }
  return 0;
}
