// emu_runtime.cpp — storage for the test-only emulation runtime (see emu_runtime.h).
#include "emu_runtime.h"
namespace emu {
thread_local BlockCtx* t_block = nullptr;
thread_local int t_tid = 0;
std::mutex g_atomic_mutex;
}  // namespace emu
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
