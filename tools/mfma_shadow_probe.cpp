// mfma_shadow_probe — with ONE wave per SIMD, what can a wave issue in the shadow of its own exact-fp32 MFMA?
// (development probe behind the design of conv_wino3_kernel, not part of the library)
// Every variant runs [v_mfma_f32_16x16x4_f32 ; K fillers] x 16 per loop iteration on 16 independent accumulators and
// reports shader cycles per MFMA (wall time x 2.4 GHz nominal; the MFMA alone is 32).  Fillers: independent v_add_f32,
// v_pk_add_f32, s_add_u32, ds_read_b128 (waited once per iteration), global_load_dwordx4 of an L2-resident line (waited
// once per iteration), s_nop 0.  A second table runs the same with TWO waves per SIMD (512 threads).
// Build: hipcc -O2 --offload-arch=gfx950 tools/mfma_shadow_probe.cpp -o tools/mfma_shadow_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifdef PROBE_BF16
// -DPROBE_BF16: the same tables for v_mfma_f32_32x32x16_bf16 (8 passes = 32 cycles alone; 8 independent 16-register accumulators)
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifdef PROBE_ACC_VGPR  // accumulators in ARCHITECTURAL registers (what the compiler does when a kernel's accumulators are read by vector code)
#define MFMA(c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a4), "v"(b4))
#else
#define MFMA(c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a4), "v"(b4))
#endif
#define NACC 8
typedef f32x16 acc_t;
#else
#define MFMA(c) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b))
#define NACC 16
typedef f32x4 acc_t;
#endif

template <int TYPE, int K>
__device__ __forceinline__ void fill(float (&v)[8], f32x2 (&pk)[4], unsigned& s, f32x4 (&ld)[4], const float* lds, const float* g,
                                     int slot) {
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int r = (slot * K + k) & 7;
    if (TYPE == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[r]) : "v"(1.0f));
    if (TYPE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[r & 3]) : "v"(pk[(r + 1) & 3]));
    if (TYPE == 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) : : "scc");
    if (TYPE == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[r & 3]) : "v"((unsigned)(size_t)lds));
    if (TYPE == 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[r & 3]) : "v"(g));
    if (TYPE == 5) asm volatile("s_nop 0");
    if (TYPE == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
    if (TYPE == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[(r + 1) & 7]));
    if (TYPE == 8) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(1.0f));
    if (TYPE == 9) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[(r + 1) & 7]));
    if (TYPE == 11) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(v[r]) : "v"(v[(r + 1) & 7]), "v"(0x3f803f80u));
    if (TYPE == 10) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[r]) : "a"(pk[r & 3][0]));  // (an AGPR no MFMA writes)
  }
}

template <int TYPE, int K>
__global__ __launch_bounds__(512) void probe(int iters, float* out, const float* g) {
  __shared__ float lds[24 * 1024];  // 96 KB: one workgroup per CU
  lds[threadIdx.x] = 1.f;
  __syncthreads();
  acc_t acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < (int)(sizeof(acc_t) / 4); ++j) acc[i][j] = 0.f;
  const float a = threadIdx.x * 1e-3f, b = 1e-3f;
  const f32x4 a4 = f32x4{a, a, a, a}, b4 = f32x4{b, b, b, b};
  (void)a4, (void)b4;
  float v[8];
  f32x2 pk[4];
  f32x4 ld[4];
  for (int i = 0; i < 8; ++i) v[i] = i;
  for (int i = 0; i < 4; ++i) pk[i] = f32x2{(float)i, 1.f}, ld[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned s = 0;
  const float* lp = lds + (threadIdx.x & 63) * 4;
  const float* gp = g + (threadIdx.x & 63) * 4;
  for (int it = 0; it < iters; ++it) {
#define STEP(i) \
  MFMA(acc[(i) % NACC]); \
  fill<TYPE, K>(v, pk, s, ld, lp, gp, i);
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14)
    STEP(15)
    if (TYPE == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (TYPE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float r = 0.f;
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) r += v[i];
  for (int i = 0; i < 4; ++i) r += pk[i][0] + pk[i][1] + ld[i][0] + ld[i][1] + ld[i][2] + ld[i][3];
  if (r == 12345.678f) out[threadIdx.x] = r + s;
}

template <int TYPE, int K>
float run(int threads, float* out, const float* g) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<TYPE, K><<<256, threads>>>(50, out, g);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<TYPE, K><<<256, threads>>>(iters, out, g);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const int waves_per_simd = threads / 256;
  return ms * 1e-3f * 2.4e9f / (iters * 16.f * waves_per_simd);  // cycles per MFMA issued on a SIMD
}

template <int TYPE>
void row(const char* name, int threads, float* out, const float* g) {
  printf("  %-22s K=0 %5.1f", name, run<TYPE, 0>(threads, out, g));
  printf(" | 1 %5.1f", run<TYPE, 1>(threads, out, g));
  printf(" | 2 %5.1f", run<TYPE, 2>(threads, out, g));
  printf(" | 3 %5.1f", run<TYPE, 3>(threads, out, g));
  printf(" | 4 %5.1f", run<TYPE, 4>(threads, out, g));
  printf(" | 6 %5.1f", run<TYPE, 6>(threads, out, g));
  printf(" | 8 %5.1f\n", run<TYPE, 8>(threads, out, g));
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float *out, *g;
  hipMalloc(&out, 4096);
  hipMalloc(&g, 4096);
  hipMemset(g, 0, 4096);
  for (int threads = 256; threads <= 512; threads += 256) {
    printf("%d wave(s) per SIMD: cycles per MFMA (nominal 2.4 GHz) with K fillers behind every MFMA\n", threads / 256);
    row<2>("s_add_u32", threads, out, g);
    row<5>("s_nop 0", threads, out, g);
    row<3>("ds_read_b128", threads, out, g);
    row<4>("global_load_dwordx4", threads, out, g);
    row<0>("v_add_f32", threads, out, g);
    row<1>("v_pk_add_f32", threads, out, g);
    row<6>("v_exp_f32", threads, out, g);
    row<8>("v_fma_f32", threads, out, g);
    row<9>("v_max_f32", threads, out, g);
    row<7>("v_cvt_pk_bf16_f32", threads, out, g);
    row<10>("v_accvgpr_read_b32", threads, out, g);
    row<11>("v_dot2c_f32_bf16", threads, out, g);
  }
  return 0;
}
