// holo_common.h — shared declarations for the gfx950 kernels and their host launchers.
//
// The product is built with hipcc --offload-arch=gfx950 only.  HOLO_EMU is a TEST-ONLY build of
// the same kernel sources against tests/emu/emu_runtime.h (host threads standing in for lanes) so
// that index arithmetic can be checked in the GPU-less development container; it is never linked
// into libholo_mi355x.so and is not a fallback.
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef HOLO_EMU
#include "emu_runtime.h"
struct f32x2 {
  float x, y;
};
static inline f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return f32x2{std::fma(a.x, b.x, c.x), std::fma(a.y, b.y, c.y)}; }
static inline f32x2 pk_mul(f32x2 a, f32x2 b) { return f32x2{a.x * b.x, a.y * b.y}; }
static inline f32x2 pk_add(f32x2 a, f32x2 b) { return f32x2{a.x + b.x, a.y + b.y}; }
static inline f32x2 pk_sub(f32x2 a, f32x2 b) { return f32x2{a.x - b.x, a.y - b.y}; }
static inline uint32_t pack_bf16x2(float lo, float hi) { return emu_bf16_bits(lo) | (emu_bf16_bits(hi) << 16); }
static inline f32x4 mfma_bf16_16x16x32(float4 a, float4 b, f32x4 c) { return emu_mfma_f32_16x16x32_bf16(a, b, c); }
static inline f32x16 mfma_bf16_32x32x16(float4 a, float4 b, f32x16 c) { return emu_mfma_f32_32x32x16_bf16(a, b, c); }
#define HOLO_LAUNDER(x) asm volatile("" : "+r"(x))
static inline float holo_rcp(float x) { return 1.0f / x; }
static inline float holo_rcp_exact(float x) { return 1.0f / x; }
static inline float holo_exp2(float x) { return std::exp2(x); }
static inline float holo_max_xor32(float x) { return std::fmax(x, __shfl_xor(x, 32)); }
static inline float holo_add_xor32(float x) { return x + __shfl_xor(x, 32); }
#define HOLO_WAVE_SYNC() emu_wave().bar.wait()
template <typename T>
static inline T holo_ld_sys(const T* p) { return *p; }
#define HOLO_PROBE_CLOCK() 0ull
#define HOLO_PROBE_HWID(hw, xcc) ((hw) = 0u, (xcc) = 0u)
#define HOLO_PHASE_DELAY(ticks) ((void)(ticks))
#define HOLO_UNIFORM(x) (x)
#define HOLO_PIN_ACC(x) ((void)0)
#define HOLO_MFMA16_ACC(acc, a, b) ((acc) = emu_mfma_f32_16x16x4f32((a), (b), (acc)))
#define HOLO_MFMA16_ACC_FIRST(acc, a, b) ((acc) = emu_mfma_f32_16x16x4f32((a), (b), (acc)))
#define HOLO_MFMA_DRAIN() ((void)0)
#define HOLO_SINK8(a, b, c, d, e, f, g, h) ((void)0)
#define HOLO_PIN_V2(x) ((void)0)
#define HOLO_ATOMIC_ADD_F32(ptr, v) atomicAdd((ptr), (v))
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two fused multiply-adds in one v_pk_fma_f32 (each element rounds exactly like fmaf)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { return a * b; }
// v_pk_add_f32 (a subtraction is the same instruction with a negated operand).  Plain vector arithmetic, NOT inline
// asm: the hazard recogniser does not see inside an asm statement, and a hand-placed VALU write next to in-flight
// MFMAs produced wrong results on gfx950.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return a - b; }
// two floats -> two bf16 (round to nearest even, v_cvt_pk_bf16_f32), `lo` in the low half
typedef __bf16 holo_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 holo_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const holo_bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
// v_mfma_f32_16x16x32_bf16 on raw 16-byte operands (8 bf16 per lane: A row / B column lane&15, k-group lane>>4)
__device__ __forceinline__ f32x4 mfma_bf16_16x16x32(float4 a, float4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(holo_bf16x8, a), __builtin_bit_cast(holo_bf16x8, b), c,
                                                 0, 0, 0);
}
// v_mfma_f32_32x32x16_bf16 on raw 16-byte operands (8 bf16 per lane: A row / B column lane&31, k-group lane>>5)
__device__ __forceinline__ f32x16 mfma_bf16_32x32x16(float4 a, float4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(holo_bf16x8, a), __builtin_bit_cast(holo_bf16x8, b), c,
                                                 0, 0, 0);
}
// v_rcp_f32 (1 ulp) / the correctly rounded reciprocal
__device__ __forceinline__ float holo_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// v_exp_f32 (2^x, no denormal fix-up: the callers' results are rounded to bf16 or summed in fp32)
__device__ __forceinline__ float holo_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// x combined with the value of lane ^ 32: v_permlane32_swap (vector pipe) instead of a trip through the LDS crossbar
__device__ __forceinline__ float holo_max_xor32(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float holo_add_xor32(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float holo_rcp_exact(float x) { return 1.0f / x; }
#define HOLO_LAUNCH(kernel, grid, block, stream, ...) \
  hipLaunchKernelGGL(kernel, grid, block, 0, (hipStream_t)(stream), __VA_ARGS__)
// Passes a per-lane value through an empty asm: the optimiser can no longer prove it loop-invariant, so index
// arithmetic derived from it is recomputed where it is used instead of being hoisted and kept live in VGPRs.
#define HOLO_LAUNDER(x) asm volatile("" : "+v"(x))
// Orders the LDS traffic of ONE wave (its lanes exchange data through a region no other wave touches): the LDS unit
// serves a wave's requests in issue order, so only the compiler has to be kept from moving accesses across this point.
#define HOLO_WAVE_SYNC()                                   \
  do {                                                     \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                       \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
// System-scope load for SMALL caller-provided tensors (timesteps, ray lists, random streams, cotangents).  Introduced in
// round 3 against a suspected stale L2 line behind fresh pageable host->device copies; round 4 traced those runs to a
// workspace race in the planner instead (tests/test_gpu_unet.py::test_repeated_forwards_are_bit_identical) and
// could not provoke a stale read with ANY kind of load (tools/h2d_stale_probe.cpp).  Kept: a handful of loads per call.
template <typename T>
__device__ __forceinline__ T holo_ld_sys(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// A wave-uniform value the compiler cannot prove uniform (e.g. threadIdx.x >> 6): v_readfirstlane moves it to an SGPR,
// so that addresses built from it use scalar bases.
#define HOLO_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// Forces a value (an MFMA accumulator) to sit in AGPRs at this point of the program.
#define HOLO_PIN_ACC(x) asm volatile("" : "+a"(x))
// v_mfma_f32_16x16x4_f32 with the accumulator TIED to an AGPR tuple (dst = src C).  For kernels whose accumulators fill
// the whole accumulation half of the register file (256 registers, one wave per SIMD): through the builtin the register
// allocator renames accumulators between MFMAs and, with no spare AGPR, parks a dozen sets in arch VGPRs (copies at every
// loop edge, scratch spills); tied operands leave it nothing to shuffle.  hipcc pads no hazards inside an asm statement
// (cdna_hip_programming.md 5.7 item 2): _FIRST opens with `s_nop 1` for A / B operands a VALU instruction has just
// written (use it for the first MFMA after the vector arithmetic that produced the operands); the accumulate chain itself
// needs no states; HOLO_MFMA_DRAIN() before anything but an MFMA reads the accumulators (8-pass MFMA: 12 states).
#define HOLO_MFMA16_ACC(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define HOLO_MFMA16_ACC_FIRST(acc, a, b) \
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#define HOLO_MFMA_DRAIN() asm volatile("s_nop 15" ::: "memory")
// pins the computation of a register pair at this point of the program (an empty asm volatile that "modifies" it: asm
// volatile statements keep their order, so the value is formed before the next asm MFMA)
#define HOLO_PIN_V2(x) asm volatile("" : "+v"(x))
// fp32 atomic add that does not return the old value: the hardware instruction (global_atomic_add_f32), not the
// compare-and-swap loop atomicAdd(float*) compiles to without -munsafe-fp-atomics (device memory from hipMalloc only)
#define HOLO_ATOMIC_ADD_F32(ptr, v) unsafeAtomicAdd((ptr), (v))
// keeps eight values (and the loads behind them) alive without using them (development probes)
#define HOLO_SINK8(a, b, c, d, e, f, g, h) asm volatile("" ::"v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h))
#define HOLO_PROBE_CLOCK() wall_clock64()
// Delays the waves that landed in an odd wave slot of their SIMD (= the second resident workgroup of the CU).
#define HOLO_PHASE_DELAY(ticks)                                                      \
  do {                                                                               \
    unsigned hw_;                                                                    \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                \
    if ((ticks) > 0 && (hw_ & 1u)) {                                                 \
      const unsigned long long t_ = wall_clock64();                                  \
      while (wall_clock64() - t_ < (unsigned long long)(ticks)) __builtin_amdgcn_s_sleep(16); \
    }                                                                                \
  } while (0)
#define HOLO_PROBE_HWID(hw, xcc)                                          \
  do {                                                                    \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));      \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));    \
  } while (0)
#endif

#define HOLO_WAVE 64

// ---- exact three-term bf16 split of fp32 values (x = hi + mid + lo, the two subtractions are exact in fp32)
__device__ __forceinline__ float bf_lo_f32(uint32_t pk) { return __uint_as_float(pk << 16); }
__device__ __forceinline__ float bf_hi_f32(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }
// two floats -> packed (hi, mid, lo) bf16 pairs
__device__ __forceinline__ void split3_pair(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = pack_bf16x2(x0, x1);
  const float r0 = x0 - bf_lo_f32(h), r1 = x1 - bf_hi_f32(h);
  m = pack_bf16x2(r0, r1);
  l = pack_bf16x2(r0 - bf_lo_f32(m), r1 - bf_hi_f32(m));
}

// ---- order-independent scatter-add (the deterministic mode of the scatter kernels, holo_ctx_set_deterministic): the values
// are added as 64-bit fixed-point integers (integer addition commutes, so the sum does not depend on the order the
// atomics land in), with the binary point placed from the largest magnitude that will be added: |v| < 2^(e+1) is scaled to
// |q| < 2^HOLO_FIX_BITS, which leaves 2^(63 - HOLO_FIX_BITS) = 8 M worst-case addends per element and quantises every
// addend to 2^-HOLO_FIX_BITS of that largest magnitude (fp32 rounds a sum of that size to 2^-24).
#define HOLO_FIX_BITS 40
// maxbits = bit pattern of max |v| (non-negative floats order like their bit patterns: atomicMax on uint32_t)
__device__ __forceinline__ int holo_fix_shift(uint32_t maxbits) { return HOLO_FIX_BITS - 1 - ((int)((maxbits >> 23) & 0xffu) - 127); }
__device__ __forceinline__ void holo_fix_add(long long* p, float v, int shift) {
  const long long q = (long long)rint(ldexp((double)v, shift));
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)q);
}
__device__ __forceinline__ float holo_fix_value(long long q, int shift) { return (float)ldexp((double)q, -shift); }

namespace holo {

// thread-local error string shared by all translation units
void set_error(const char* fmt, ...);
const char* get_error();

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace holo
