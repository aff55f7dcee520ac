// h2d_stale_probe — which kind of read can return STALE bytes of a small tensor that was just copied from pageable host
// memory on the same stream?  (The round-3 finding behind holo_ld_sys, holo_common.h: 8 of 610 denoiser calls read an old
// timestep right after `tensor.to(device)`; this probe isolates the mechanism.)
//
// Loop: write a fresh 16-byte value into a pageable host buffer, hipMemcpyAsync it to the SAME device address as the last
// iteration, launch a kernel (same stream, no host synchronisation in between) in which every workgroup reads the value
// three ways - a wave-uniform read the compiler turns into s_load_dwordx2 (scalar cache), a per-lane global_load (vector
// L1 / L2), a system-scope atomic load (sc0 sc1: past the L2) - and stores what it saw.  The previous iteration's kernel has
// left the line in whatever caches its reads went through.  Counted per kind: iterations in which ANY workgroup saw a value
// other than the fresh one.  Variants: copy from pageable / pinned memory; a second "toucher" kernel that re-reads the
// address with scalar loads only (keeps the scalar caches warm) before the next copy.
// Build: hipcc -O2 --offload-arch=gfx950 tools/h2d_stale_probe.cpp -o tools/h2d_stale_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x)                                                         \
  do {                                                                \
    hipError_t e = (x);                                               \
    if (e != hipSuccess) {                                            \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
      exit(1);                                                        \
    }                                                                 \
  } while (0)

__global__ void reader(const long long* t, long long* out) {
  const long long a = t[0];                    // wave-uniform address: s_load_dwordx2
  const long long b = t[threadIdx.x & 1];      // per-lane address: global_load_dwordx2
  const long long c = __hip_atomic_load(t + (threadIdx.x & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (threadIdx.x < 2) {
    out[(blockIdx.x * 2 + threadIdx.x) * 3 + 0] = a;
    out[(blockIdx.x * 2 + threadIdx.x) * 3 + 1] = b;
    out[(blockIdx.x * 2 + threadIdx.x) * 3 + 2] = c;
  }
}
__global__ void toucher(const long long* t, long long* sink) {
  if (t[0] == 0x7fffffffffffffffll) sink[0] = 1;  // scalar read only
}
// the address's previous owner: a kernel writes other bytes there (block 0), ...
__global__ void polluter(long long* t, long long v) {
  if (blockIdx.x == 0 && threadIdx.x < 2) t[threadIdx.x] = v;
}
// ... and keeps the queue busy for a while, so that the next host->device copy is enqueued behind running work
__global__ void spinner(long long* sink, int ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks < 0) sink[0] = 2;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int nblk = 512;
  long long *dev, *out, *sink;
  CK(hipMalloc(&dev, 4096));
  CK(hipMalloc(&out, nblk * 2 * 3 * 8));
  CK(hipMalloc(&sink, 8));
  CK(hipMemset(dev, 0, 4096));
  long long* pinned;
  CK(hipHostMalloc(&pinned, 4096));
  long long* pageable = (long long*)malloc(4096);
  std::vector<long long> h(nblk * 2 * 3);
  for (int variant = 0; variant < 8; ++variant) {
    const bool use_pinned = variant & 1, touch = variant & 2, busy = variant & 4;
    long long stale[3] = {0, 0, 0}, first_bad[3] = {-1, -1, -1};
    for (int it = 1; it <= (busy ? iters / 10 : iters); ++it) {
      long long* src = use_pinned ? pinned : pageable;
      src[0] = 1000003ll * it + variant;
      src[1] = src[0];
      if (busy) {  // previous owner's bytes in every L2 (written, then read from all XCDs), and ~100 us of queued work
        polluter<<<nblk, 64>>>(dev, -7 - it);
        reader<<<nblk, 64>>>(dev, out);
        spinner<<<256, 64>>>(sink, 10000);
      }
      CK(hipMemcpyAsync(dev, src, 16, hipMemcpyHostToDevice, nullptr));
      reader<<<nblk, 64>>>(dev, out);
      if (touch) toucher<<<nblk, 64>>>(dev, sink);
      CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
      for (int k = 0; k < 3; ++k) {
        bool bad = false;
        for (int i = 0; i < nblk * 2; ++i) bad |= h[i * 3 + k] != src[0];
        if (bad) {
          stale[k]++;
          if (first_bad[k] < 0) first_bad[k] = it;
        }
      }
    }
    printf("%s%s source%s: %d copies of 16 bytes, iterations with a stale read: scalar load %lld (first at %lld) | vector load %lld "
           "(first at %lld) | system-scope load %lld (first at %lld)\n",
           busy ? "busy queue, " : "idle queue, ", use_pinned ? "pinned  " : "pageable", touch ? " + scalar re-reads" : "", busy ? iters / 10 : iters, stale[0], first_bad[0], stale[1], first_bad[1],
           stale[2], first_bad[2]);
  }
  return 0;
}
