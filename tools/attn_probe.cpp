// attn_probe — timing + spot check of the bf16 attention (attn_pack_kernel + flash_attn_bf16v2_kernel + attn_combine_kernel)
// on the shapes of the 128^3 net (T = 32 768 / 4 096 tokens; development probe, not part of the library).
// Checks a few queries per case against a float64 softmax over the bf16-rounded operands, on four input distributions:
//   0 plain (scores of a few units), 1 large scores (|s| up to ~60 in the exp2 domain), 2 adversarial: the scores of a query
//   climb by ~3 per 64-key block along the sequence (every block moves the running maximum; exercises the re-referencing)
//   3 a jump of several hundred between two blocks (exercises the exact fallback of the lazy-reference loop)
// Build: bash tools/build_attn_probe.sh      Usage: attn_probe [iters=20] [ncases]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

static float frand() { return (rand() % 20001 - 10000) * 1e-4f; }
static float bf16r(float f) {  // round to nearest even bf16, back to float (what attn_pack_kernel stores)
  uint32_t u;
  memcpy(&u, &f, 4);
  u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const int ncases = argc > 2 ? atoi(argv[2]) : 100;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int num_cus = prop.multiProcessorCount;
  struct Case {
    int T, C, H, dist;
    const char* what;
  };
  const Case cases[] = {
      {32768, 128, 2, 0, "T 32768, C 128, 2 heads (32^3 level of the 128^3 net), plain"},
      {32768, 128, 2, 1, "T 32768, large scores"},
      {32768, 128, 2, 2, "T 32768, scores climbing along the sequence"},
      {32768, 128, 2, 3, "T 32768, one jump of several hundred"},
      {4096, 256, 4, 0, "T 4096, C 256, 4 heads (16^3 level), plain"},
      {4096, 256, 4, 2, "T 4096, climbing"},
      {4096, 256, 2, 0, "T 4096, C 256, 2 heads (128 head channels), plain"},
      {1024, 64, 2, 0, "T 1024, C 64, 2 heads (32 head channels), plain"},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int bad_cases = 0;
  int case_no = 0;
  for (const Case& c : cases) {
    if (case_no++ >= ncases) break;
    const int T = c.T, C = c.C, H = c.H, CH = C / H;
    std::vector<float> qkv((size_t)T * 3 * C);
    for (auto& x : qkv) x = frand();
    // per head: [q(CH) k(CH) v(CH)]
    for (int t = 0; t < T; ++t)
      for (int h = 0; h < H; ++h) {
        float* q = &qkv[((size_t)t * H + h) * 3 * CH];
        float* k = q + CH;
        if (c.dist == 1)
          for (int i = 0; i < CH; ++i) q[i] *= 6.f, k[i] *= 6.f;
        if (c.dist == 2) {  // k = (t / 64) * 0.35 * u, q = 6 * u + noise, u = ones / sqrt(CH): score ~ +2 per block (x log2 e)
          for (int i = 0; i < CH; ++i) k[i] = 0.2f * k[i] + (t / 64) * 0.35f / sqrtf((float)CH), q[i] = 0.2f * q[i] + 6.f;
        }
        if (c.dist == 3) {
          const float lvl = (t % (T / 2)) < 3 * T / 8 ? 0.f : 60.f;  // (inside each half: the key range may be split in two)
          for (int i = 0; i < CH; ++i) k[i] = 0.2f * k[i] + lvl / sqrtf((float)CH), q[i] = 0.2f * q[i] + 6.f;
        }
      }
    float *d_qkv, *d_out;
    CK(hipMalloc(&d_qkv, qkv.size() * 4));
    CK(hipMemcpy(d_qkv, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, (size_t)T * C * 4));
    AttnParams p{};
    p.qkv = d_qkv, p.out = d_out, p.N = 1, p.T = T, p.C = C, p.H = H;
    p.scale2 = 1.f / sqrtf((float)CH);
    void* work;
    CK(hipMalloc(&work, flash_attn_bf16v2_workspace_bytes(p, num_cus)));
    if (flash_attn_bf16v2_launch(p, work, 0, num_cus, nullptr) != 0) return 1;
    CK(hipDeviceSynchronize());
    for (int i = 0; i < 3; ++i) flash_attn_bf16v2_launch(p, work, 0, num_cus, nullptr);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) flash_attn_bf16v2_launch(p, work, 0, num_cus, nullptr);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 4.0 * T * (double)T * C / (us * 1e-6) / 1e12;
    std::vector<float> out((size_t)T * C);
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    // spot check: float64 softmax over the operands as the kernel sees them (bf16 Q * scale * log2 e, K, V)
    const float qscale = p.scale2 * 1.4426950408889634f;
    double worst = 0, scale = 0;
    int nan = 0;
    const int qs[] = {0, 1, 31, 32, 63, 64, 255, 256, T / 2 - 1, T / 2, T - 257, T - 1};
    for (int q : qs)
      for (int h = 0; h < H; ++h) {
        const float* qv = &qkv[((size_t)q * H + h) * 3 * CH];
        std::vector<double> s(T);
        double mx = -1e300;
        for (int t = 0; t < T; ++t) {
          const float* kv = &qkv[((size_t)t * H + h) * 3 * CH + CH];
          double a = 0;
          for (int i = 0; i < CH; ++i) a += (double)bf16r(qv[i] * qscale) * bf16r(kv[i]);
          s[t] = a * 0.6931471805599453;  // (the kernel's scores are in the exp2 domain: q carries scale * log2 e)
          mx = fmax(mx, s[t]);
        }
        double l = 0;
        std::vector<double> o(CH, 0.0);
        for (int t = 0; t < T; ++t) {
          const double w = exp(s[t] - mx);
          l += w;
          const float* vv = &qkv[((size_t)t * H + h) * 3 * CH + 2 * CH];
          for (int i = 0; i < CH; ++i) o[i] += w * bf16r(vv[i]);
        }
        for (int i = 0; i < CH; ++i) {
          const double ref = o[i] / l, got = out[(size_t)q * C + h * CH + i];
          if (!(got == got)) ++nan;
          worst = fmax(worst, fabs(ref - got));
          scale = fmax(scale, fabs(ref));
        }
      }
    const bool ok = nan == 0 && worst <= 2e-2 * fmax(scale, 0.05);
    if (!ok) ++bad_cases;
    printf("%-62s %8.1f us  %7.1f TF/s = %.3f of the bf16 pipe | max|d| %.2e (max|ref| %.2e, %d NaN) %s\n", c.what, us, tf,
           tf / 2516.0, worst, scale, nan, ok ? "ok" : "MISMATCH");
    CK(hipFree(d_qkv));
    CK(hipFree(d_out));
    CK(hipFree(work));
  }
  printf(bad_cases ? "FAILED: %d case(s)\n" : "all cases ok\n", bad_cases);
  return bad_cases ? 1 : 0;
}
