// emu_runtime.h — TEST-ONLY host emulation of the handful of HIP constructs the kernels use.
//
// The development container has no GPU.  To catch index / layout / bounds bugs before spending GPU
// minutes, tests/emu builds the SAME kernel sources with -DHOLO_EMU: every GPU thread becomes a host
// thread, a workgroup runs to completion before the next starts, __syncthreads() is a barrier, and the
// wave-collective instructions (MFMA, shuffles) exchange operands through a per-wave buffer using the
// documented gfx950 lane mappings.  This is not a product path: nothing in holo_diffusion_amd/ links it.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 {
  float x, y, z, w;
};
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct uint2 {
  uint32_t x, y;
};
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
struct float2 {
  float x, y;
};
struct double2 {
  double x, y;
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef float f32x16 __attribute__((vector_size(64)));
typedef float f32x4 __attribute__((vector_size(16)));

#include <algorithm>
using std::max;
using std::min;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__

namespace emu {

struct SpinBarrier {
  std::atomic<int> expected{0};
  std::atomic<int> waiting{0};
  std::atomic<uint64_t> gen{0};
  std::mutex m;
  void init(int n) {
    expected = n;
    waiting = 0;
    gen = 0;
  }
  void wait() {
    uint64_t g;
    {
      std::lock_guard<std::mutex> lk(m);
      g = gen.load();
      if (waiting.load() + 1 == expected.load()) {
        waiting = 0;
        gen.store(g + 1);
        return;
      }
      waiting++;
    }
    while (gen.load() == g) std::this_thread::yield();
  }
  void drop() {
    std::lock_guard<std::mutex> lk(m);
    expected--;
    if (expected.load() > 0 && waiting.load() == expected.load()) {
      waiting = 0;
      gen++;
    }
  }
};

struct WaveCtx {
  SpinBarrier bar;
  float a[64], b[64];
  float a4[64][4], b4[64][4];  // 16-byte operands of the bf16 MFMA
};

struct BlockCtx {
  SpinBarrier bar;
  std::vector<WaveCtx> waves;
};

extern thread_local BlockCtx* t_block;
extern thread_local int t_tid;  // linear thread id in block
extern std::mutex g_atomic_mutex;

}  // namespace emu

extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

static inline void emu_serial_abort(const char* what) {
  fprintf(stderr, "[emu] %s inside a kernel launched in serial mode (emu_serial_kernel list is wrong)\n", what);
  abort();
}
static inline void __syncthreads() {
  if (!emu::t_block) emu_serial_abort("__syncthreads");
  emu::t_block->bar.wait();
}

static inline emu::WaveCtx& emu_wave() {
  if (!emu::t_block) emu_serial_abort("a wave-collective operation");
  return emu::t_block->waves[emu::t_tid >> 6];
}

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D col = l&31, row = (r&3)+8*(r>>2)+4*(l>>5)
static inline f32x16 emu_mfma_f32_32x32x2f32(float a, float b, f32x16 c) {
  emu::WaveCtx& w = emu_wave();
  const int lane = emu::t_tid & 63;
  w.a[lane] = a;
  w.b[lane] = b;
  w.bar.wait();
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int col = lane & 31;
    float acc = c[r];
    for (int k = 0; k < 2; ++k) acc = std::fma(w.a[row + 32 * k], w.b[col + 32 * k], acc);
    c[r] = acc;
  }
  w.bar.wait();
  return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_f32_32x32x2f32(a, b, c)

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D col = l&15, row = 4*(l>>4)+r
static inline f32x4 emu_mfma_f32_16x16x4f32(float a, float b, f32x4 c) {
  emu::WaveCtx& w = emu_wave();
  const int lane = emu::t_tid & 63;
  w.a[lane] = a;
  w.b[lane] = b;
  w.bar.wait();
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (lane >> 4) + r;
    const int col = lane & 15;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = std::fma(w.a[row + 16 * k], w.b[col + 16 * k], acc);
    c[r] = acc;
  }
  w.bar.wait();
  return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) emu_mfma_f32_16x16x4f32(a, b, c)

static inline uint32_t __float_as_uint(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
static inline float __uint_as_float(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
// bf16 helpers (round-to-nearest-even, like v_cvt_pk_bf16_f32)
static inline uint32_t emu_bf16_bits(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
static inline float emu_bf16_to_f32(uint32_t h) {
  uint32_t u = h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
// v_mfma_f32_16x16x32_bf16: lane l holds 8 bf16 of A row l&15 / B column l&15 for k-group l>>4;
// D col = l&15, row = 4*(l>>4)+r.  Operands are passed as raw 16-byte registers.
static inline f32x4 emu_mfma_f32_16x16x32_bf16(float4 a, float4 b, f32x4 c) {
  emu::WaveCtx& w = emu_wave();
  const int lane = emu::t_tid & 63;
  std::memcpy(w.a4[lane], &a, 16);
  std::memcpy(w.b4[lane], &b, 16);
  w.bar.wait();
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * (lane >> 4) + r;
    const int col = lane & 15;
    float acc = c[r];
    for (int g = 0; g < 4; ++g) {
      uint32_t aw[4], bw[4];
      std::memcpy(aw, w.a4[row + 16 * g], 16);
      std::memcpy(bw, w.b4[col + 16 * g], 16);
      for (int e = 0; e < 8; ++e) {
        const float av = emu_bf16_to_f32((aw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
        const float bv = emu_bf16_to_f32((bw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
        acc = std::fma(av, bv, acc);
      }
    }
    c[r] = acc;
  }
  w.bar.wait();
  return c;
}

// v_mfma_f32_32x32x16_bf16: lane l holds 8 bf16 of A row l&31 / B column l&31 for k-group l>>5;
// D col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
static inline f32x16 emu_mfma_f32_32x32x16_bf16(float4 a, float4 b, f32x16 c) {
  emu::WaveCtx& w = emu_wave();
  const int lane = emu::t_tid & 63;
  std::memcpy(w.a4[lane], &a, 16);
  std::memcpy(w.b4[lane], &b, 16);
  w.bar.wait();
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int col = lane & 31;
    float acc = c[r];
    for (int g = 0; g < 2; ++g) {
      uint32_t aw[4], bw[4];
      std::memcpy(aw, w.a4[row + 32 * g], 16);
      std::memcpy(bw, w.b4[col + 32 * g], 16);
      for (int e = 0; e < 8; ++e) {
        const float av = emu_bf16_to_f32((aw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
        const float bv = emu_bf16_to_f32((bw[e >> 1] >> (16 * (e & 1))) & 0xffffu);
        acc = std::fma(av, bv, acc);
      }
    }
    c[r] = acc;
  }
  w.bar.wait();
  return c;
}

static inline float __shfl_xor(float v, int mask) {
  emu::WaveCtx& w = emu_wave();
  const int lane = emu::t_tid & 63;
  w.a[lane] = v;
  w.bar.wait();
  float r = w.a[(lane ^ mask) & 63];
  w.bar.wait();
  return r;
}
static inline float __shfl(float v, int src) {
  emu::WaveCtx& w = emu_wave();
  const int lane = emu::t_tid & 63;
  w.a[lane] = v;
  w.bar.wait();
  float r = w.a[src & 63];
  w.bar.wait();
  return r;
}

// wave vote: true if the predicate holds on any lane (every lane of the wave must call it)
static inline int __any(int pred) {
  emu::WaveCtx& w = emu_wave();
  const int lane = emu::t_tid & 63;
  w.a[lane] = pred ? 1.f : 0.f;
  w.bar.wait();
  int r = 0;
  const int n = w.bar.expected.load();
  for (int l = 0; l < n && l < 64; ++l) r |= (w.a[l] != 0.f);
  w.bar.wait();
  return r;
}

static inline double atomicAdd(double* p, double v) {
  std::lock_guard<std::mutex> lk(emu::g_atomic_mutex);
  double o = *p;
  *p = o + v;
  return o;
}
static inline int atomicAdd(int* p, int v) {
  std::lock_guard<std::mutex> lk(emu::g_atomic_mutex);
  int o = *p;
  *p = o + v;
  return o;
}
static inline float atomicAdd(float* p, float v) {
  std::lock_guard<std::mutex> lk(emu::g_atomic_mutex);
  float o = *p;
  *p = o + v;
  return o;
}
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
  std::lock_guard<std::mutex> lk(emu::g_atomic_mutex);
  unsigned long long o = *p;
  *p = o + v;
  return o;
}
static inline unsigned int atomicMax(unsigned int* p, unsigned int v) {
  std::lock_guard<std::mutex> lk(emu::g_atomic_mutex);
  unsigned int o = *p;
  if (v > o) *p = v;
  return o;
}

static inline float emu_expf(float x) { return std::exp(x); }
#define __expf(x) emu_expf(x)
static inline float __fmul_rn(float a, float b) {
  volatile float r = a * b;
  return r;
}
static inline float __fadd_rn(float a, float b) {
  volatile float r = a + b;
  return r;
}
static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }

static inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
static inline void emu_wave_barrier() { emu_wave().bar.wait(); }
#define __builtin_amdgcn_wave_barrier() emu_wave_barrier()

// ---- minimal HIP runtime API shims (host memory stands in for device memory)
typedef int hipError_t;
typedef void* hipStream_t;
typedef struct EmuEvent { double t; }* hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
struct hipDeviceProp_t {
  int multiProcessorCount;
};
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  const char* e = getenv("HOLO_EMU_CUS");
  p->multiProcessorCount = e ? atoi(e) : 4;  // small "chip" so that split-K paths are exercised on tiny problems
  return hipSuccess;
}
static inline hipError_t hipMalloc(void** p, size_t n) {
  *p = malloc(n);
  return *p ? hipSuccess : 1;
}
static inline hipError_t hipFree(void* p) {
  free(p);
  return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  memcpy(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
  memset(d, v, n);
  return hipSuccess;
}
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new EmuEvent{0};
  return hipSuccess;
}
static inline hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) {
  *ms = 1.f;
  return hipSuccess;
}

namespace emu {
// Persistent worker pool: a block's lanes are host threads that live across launches (creating 256 threads per block
// dominated the run time of the barrier-free repack kernels).
struct Pool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_start, cv_done;
  std::function<void(int)> job;
  uint64_t gen = 0;
  int nactive = 0, remaining = 0;
  void worker(int i) {
    uint64_t seen = 0;
    for (;;) {
      std::function<void(int)> f;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_start.wait(lk, [&] { return gen != seen; });
        seen = gen;
        if (i >= nactive) continue;
        f = job;
      }
      f(i);
      {
        std::lock_guard<std::mutex> lk(m);
        if (--remaining == 0) cv_done.notify_all();
      }
    }
  }
  void run(int n, std::function<void(int)> f) {
    while ((int)th.size() < n) {
      const int i = (int)th.size();
      {
        std::lock_guard<std::mutex> lk(m);  // a new worker must not mistake the CURRENT generation for new work
      }
      th.emplace_back([this, i] { worker(i); });
      th.back().detach();
    }
    std::unique_lock<std::mutex> lk(m);
    job = std::move(f);
    nactive = n;
    remaining = n;
    ++gen;
    cv_start.notify_all();
    cv_done.wait(lk, [&] { return remaining == 0; });
  }
};
inline Pool& pool() {
  static Pool* p = new Pool();  // leaked on purpose: detached workers outlive static destruction
  return *p;
}

// Kernels known to be free of barriers and wave-collective operations (weight repacks, element-wise copies): their
// lanes run one after the other on the calling thread.  A wrong entry aborts loudly (emu_serial_abort).
static inline bool serial_kernel(const char* name) {
  static const char* const pre[] = {"repack_", "copy_sys_kernel", "tanh_kernel", "clip_kernel", "ddpm_step_kernel"};
  for (const char* p : pre)
    if (strncmp(name, p, strlen(p)) == 0) return true;
  return false;
}

template <class K, class... Args>
void launch(const char* name, K kernel, dim3 grid, dim3 block, Args... args) {
  const int nthreads = (int)(block.x * block.y * block.z);
  const int nwaves = (nthreads + 63) / 64;
  if (serial_kernel(name)) {
    t_block = nullptr;
    blockDim = block;
    gridDim = grid;
    for (unsigned bz = 0; bz < grid.z; ++bz)
      for (unsigned by = 0; by < grid.y; ++by)
        for (unsigned bx = 0; bx < grid.x; ++bx) {
          blockIdx = dim3(bx, by, bz);
          for (int t = 0; t < nthreads; ++t) {
            t_tid = t;
            threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            kernel(args...);
          }
        }
    return;
  }
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        BlockCtx ctx;
        ctx.bar.init(nthreads);
        ctx.waves = std::vector<WaveCtx>(nwaves);
        for (int w = 0; w < nwaves; ++w) {
          int n = nthreads - w * 64;
          ctx.waves[w].bar.init(n > 64 ? 64 : n);
        }
        pool().run(nthreads, [&](int t) {
          t_block = &ctx;
          t_tid = t;
          threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          blockIdx = dim3(bx, by, bz);
          blockDim = block;
          gridDim = grid;
          kernel(args...);
          ctx.waves[t >> 6].bar.drop();
          ctx.bar.drop();
        });
      }
}
}  // namespace emu

static inline void emu_trace_launch(const char* name, dim3 g, dim3 b) {
  static const bool on = getenv("HOLO_EMU_TRACE") != nullptr;
  if (on) fprintf(stderr, "[emu] %s grid (%u,%u,%u) block %u\n", name, g.x, g.y, g.z, b.x * b.y * b.z);
}
#define HOLO_LAUNCH(kernel, grid, block, stream, ...) \
  (emu_trace_launch(#kernel, grid, block), emu::launch(#kernel, kernel, grid, block, __VA_ARGS__))
