// conv_timeline — per-workgroup timeline of one LDS-halo conv3d launch (development probe, not part of the library).
// Build: hipcc -O2 --offload-arch=gfx950 tools/conv_timeline.cpp holo_diffusion_amd/csrc/kernels_conv.o \
//              holo_diffusion_amd/csrc/kernels_misc.o -o tools/conv_timeline
// Usage: conv_timeline [R=64] [Cin=64] [Cout=64] [kernel: 0 direct, 2 = (z,y) Winograd, 3 = bf16 wide-tile (bf16 storage)]
//                      [tile_depth=0 (planner)] [stagger_us=0] [act=0: 1 = GroupNorm affine + SiLU while staging]
//                      [epi=0: 1 = residual + bias + GroupNorm statistics in the epilogue]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>
#include "../holo_diffusion_amd/csrc/holo_kernels.h"
#include <stdarg.h>
namespace holo { void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vprintf(fmt, a); va_end(a); printf("\n"); } }
using namespace holo;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 64, Cin = argc > 2 ? atoi(argv[2]) : 64, Cout = argc > 3 ? atoi(argv[3]) : 64;
  const int gx = argc > 4 ? atoi(argv[4]) : 0;
  const int tzo = argc > 5 ? atoi(argv[5]) : 0;
  const int stag = argc > 6 ? atoi(argv[6]) : 0;  // microseconds
  const int64_t V = (int64_t)R * R * R;
  float *src, *w, *out; unsigned long long* dbg;
  const int CinP = (Cin + 31) / 32 * 32, CoutP = Cout >= 64 ? (Cout + 63) / 64 * 64 : 32;
  CK(hipMalloc(&src, V * Cin * 4)); CK(hipMalloc(&out, V * Cout * 4)); CK(hipMalloc(&w, (size_t)27 * CinP * CoutP * 4));
  std::vector<float> h(V * Cin); for (auto& x : h) x = (rand() % 2001 - 1000) * 1e-3f;
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> hw((size_t)27 * CinP * CoutP); for (auto& x : hw) x = (rand() % 2001 - 1000) * 1e-4f;
  CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  ConvParams p{}; p.src0 = src; p.C0 = Cin; p.N = 1; p.ID = p.IH = p.IW = p.OD = p.OH = p.OW = R; p.stride = 1; p.pad = 1; p.ksz = 3;
  p.Cout = Cout; p.w = w; p.CoutP = CoutP; p.CinP = CinP; p.out = out;
  float* w2 = nullptr;
  if (gx == 2) {  // (z,y) Winograd form: 48 pseudo-taps (random values: timing only)
    CK(hipMalloc(&w2, (size_t)48 * CinP * CoutP * 4));
    std::vector<float> hw2((size_t)48 * CinP * CoutP); for (auto& x : hw2) x = (rand() % 2001 - 1000) * 1e-4f;
    CK(hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice));
    p.w_wino = w2; p.w_wino2 = w2;
  }
  const int act = argc > 7 ? atoi(argv[7]) : 0;
  if (act) {
    float* coef; CK(hipMalloc(&coef, (size_t)Cin * 2 * 4));
    std::vector<float> hc((size_t)Cin * 2); for (int c = 0; c < Cin; ++c) { hc[2 * c] = 1.0f + 0.01f * (c % 7); hc[2 * c + 1] = 0.01f * (c % 5); }
    CK(hipMemcpy(coef, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    p.coef = coef; p.act = 1;
  }
  if (gx == 3) {  // bf16 storage + wide-tile kernel: bf16 activations and packed bf16 weights (random values: timing only)
    std::vector<uint16_t> hb((size_t)V * Cin); for (auto& x : hb) x = (uint16_t)(0x3c00 + rand() % 0x300) | (rand() & 1 ? 0x8000 : 0);
    CK(hipMemcpy(src, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
    uint16_t* wb; CK(hipMalloc(&wb, (size_t)27 * CinP * CoutP * 2));
    std::vector<uint16_t> hwb((size_t)27 * CinP * CoutP); for (auto& x : hwb) x = (uint16_t)(0x3800 + rand() % 0x300) | (rand() & 1 ? 0x8000 : 0);
    CK(hipMemcpy(wb, hwb.data(), hwb.size() * 2, hipMemcpyHostToDevice));
    p.bf16 = 1; p.w_bf = wb; p.w_bft = wb; p.in_bf16 = p.res_bf16 = p.out_bf16 = 1;
  }
  if (argc > 8 && atoi(argv[8])) {  // the epilogue of a resblock's second conv: residual, bias, statistics of the output
    float *res, *bias; double* stats;
    CK(hipMalloc(&res, V * Cout * 4)); CK(hipMemset(res, 0, V * Cout * 4));
    CK(hipMalloc(&bias, Cout * 4)); CK(hipMemset(bias, 0, Cout * 4));
    CK(hipMalloc(&stats, (size_t)(V / 64) * Cout * 2 * 8));
    p.residual = res; p.bias = bias; p.stats = stats;
  }
  conv_plan(p, 256);
  if (tzo > 0) { p.tz = tzo; p.grid_x = (int)(V / (64 * p.tz)); }
  (void)gx;  // (the persistent multi-tile form was removed from the kernel after these measurements; one tile per workgroup)
  p.stagger_ticks = stag * 100;
  const int ntile = (int)(V / (64 * p.tz));
  const int ny = (Cout + 63) / 64;
  CK(hipMalloc(&dbg, (size_t)ntile * ny * 64)); CK(hipMemset(dbg, 0, (size_t)ntile * ny * 64));
  printf("mode %d tz %d nsplit %d grid_x %d tiles %d\n", p.mode, p.tz, p.nsplit, p.grid_x, ntile);
  for (int i = 0; i < 3; ++i) conv_launch(p, nullptr);
  CK(hipDeviceSynchronize());
  p.dbg = dbg;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, nullptr)); conv_launch(p, nullptr); CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> d((size_t)ntile * ny * 8);
  CK(hipMemcpy(d.data(), dbg, d.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, t1 = 0; double pro = 0, loop = 0, epi = 0;
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> per_cu;
  for (int i = 0; i < ntile * ny; ++i) {
    auto* e = &d[(size_t)i * 8];
    t0 = std::min(t0, e[0]); t1 = std::max(t1, e[3]);
    pro += e[1] - e[0]; loop += e[2] - e[1]; epi += e[3] - e[2];
    unsigned long long key = (e[5] & 0xf) << 32 | (e[4] & 0xff00);  // xcc, se/sh/cu
    per_cu[key].push_back({e[0], e[3]});
  }
  const double n = ntile * ny, tick = 0.01;  // us per 100 MHz tick
  printf("event ms %.4f  span(first start..last end) %.2f us\n", ms, (t1 - t0) * tick);
  printf("per tile: prologue %.2f us  loops %.2f us  epilogue %.2f us  total %.2f us\n", pro / n * tick, loop / n * tick, epi / n * tick, (pro + loop + epi) / n * tick);
  { double stg = 0; for (int i = 0; i < ntile * ny; ++i) stg += d[(size_t)i * 8 + 6]; printf("per tile: staging (issue -> barrier, all chunks) %.2f us [wino %d]\n", stg / n * tick, p.wino); }
  // occupancy of CU slots over time
  double busy = 0, gaps = 0; int ncu = 0; double firsts = 0, lasts = 0;
  for (auto& kv : per_cu) {
    auto& v = kv.second; std::sort(v.begin(), v.end());
    ++ncu; firsts += (v.front().first - t0) * tick; 
    unsigned long long mx = 0; for (auto& a : v) { busy += (a.second - a.first) * tick; mx = std::max(mx, a.second); }
    lasts += (t1 - mx) * tick;
  }
  printf("CUs seen %d; tiles per CU %.1f; mean first-start delay %.2f us; mean idle tail %.2f us; mean busy tile-time per CU %.1f us (= %.2f tiles in flight over the span)\n",
         ncu, n / ncu, firsts / ncu, lasts / ncu, busy / ncu, busy / ncu / ((t1 - t0) * tick));
  {  // per-XCD statistics: mean tile duration, mean loop duration, last end
    double dur[16] = {0}, lp[16] = {0}; int cnt[16] = {0}; unsigned long long last[16] = {0};
    for (int i = 0; i < ntile * ny; ++i) {
      auto* e = &d[(size_t)i * 8]; const int x = (int)(e[5] & 0xf);
      dur[x] += (e[3] - e[0]) * tick; lp[x] += (e[2] - e[1]) * tick; cnt[x]++; last[x] = std::max(last[x], e[3]);
    }
    for (int x = 0; x < 16; ++x) if (cnt[x]) printf("  xcc %d: tiles %d mean tile %.2f us loops %.2f us last end %.1f us\n", x, cnt[x], dur[x] / cnt[x], lp[x] / cnt[x], (last[x] - t0) * tick);
    // per-CU spread of the time the CU finished
    std::vector<double> fin; for (auto& kv : per_cu) { unsigned long long mx = 0; for (auto& a : kv.second) mx = std::max(mx, a.second); fin.push_back((mx - t0) * tick); }
    std::sort(fin.begin(), fin.end());
    printf("  CU finish time: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f us\n", fin.front(), fin[fin.size() / 10], fin[fin.size() / 2], fin[fin.size() * 9 / 10], fin.back());
  }
  {  // start-time percentiles of the FIRST tile of every workgroup (dispatch ramp) and per-slot order on one CU
    std::vector<double> st; const int nwg = std::min(p.grid_x, ntile) * ny;
    for (int i = 0; i < nwg; ++i) st.push_back((d[(size_t)i * 8] - t0) * tick);
    std::vector<double> so = st; std::sort(so.begin(), so.end());
    printf("  first-tile start: p25 %.2f p50 %.2f p60 %.2f p75 %.2f p90 %.2f p99 %.2f max %.2f us\n", so[nwg / 4], so[nwg / 2], so[nwg * 6 / 10], so[nwg * 3 / 4], so[nwg * 9 / 10], so[nwg * 99 / 100], so.back());
    printf("  start of wg 0,8,16,...: "); for (int i = 0; i < nwg && i < 8 * 70; i += 8) printf("%.1f ", st[i]); printf("\n");
  }
  // histogram of start times in 10 buckets
  int hist[10] = {0}; for (int i = 0; i < ntile * ny; ++i) { int b = (int)((d[(size_t)i * 8] - t0) * 10 / (t1 - t0 + 1)); hist[b]++; }
  printf("tile starts per tenth of the span:"); for (int b = 0; b < 10; ++b) printf(" %d", hist[b]); printf("\n");
  return 0;
}
