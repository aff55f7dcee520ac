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
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define HOLO_LAUNCH(kernel, grid, block, stream, ...) \
  hipLaunchKernelGGL(kernel, grid, block, 0, (hipStream_t)(stream), __VA_ARGS__)
#endif

#define HOLO_WAVE 64

namespace holo {

// thread-local error string shared by all translation units
void set_error(const char* fmt, ...);
const char* get_error();

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace holo
