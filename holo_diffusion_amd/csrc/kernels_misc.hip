// kernels_misc.hip — HBM-bound kernels around the MFMA ops of the denoiser.
//
//   layout     NCDHW (plugin boundary) <-> channels-last (kernel layout); optional tanh
//              (holo_diffusion_model.py:425)
//   gn_stats   per-(sample, channel) sum / sum-of-squares of GroupNorm32's input (nn.py:23-25), in double
//   gn_finalize GroupNorm(32, C, eps=1e-5) folded with affine and FiLM (unet.py:248-252) into (a,b)/channel
//   time_embed timestep_embedding + time_embed MLP (nn.py:109-127, unet.py:645-650)
//   rows_linear all ResBlock emb_layers Linear(SiLU(emb)) at once (unet.py:199-205,245)
//   ddpm_step  clamp + posterior mean + noise (gaussian_diffusion.py:314-343,237-240,499-506)
#include <stdlib.h>

#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {
namespace {

// ---------------------------------------------------------------------------------------------
// [N][C][V] -> [N][V][C] through a 32x33 LDS tile.  grid = (V/32, C/32, N), block = (32, 8)
// ---------------------------------------------------------------------------------------------
// out_bf16 / in_bf16: the channels-last side is a bf16 tensor (bf16 storage mode of the denoiser)
__global__ __launch_bounds__(256) void ncdhw_to_ndhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             int C, int64_t V, int tanh_flag, int out_bf16) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int64_t v0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  in += (int64_t)n * C * V;
  const int64_t obase = (int64_t)n * C * V;
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j;
    const int64_t v = v0 + tx;
    float x = 0.f;
    if (c < C && v < V) x = in[(int64_t)c * V + v];
    if (tanh_flag) x = tanhf(x);
    tile[j][tx] = x;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int64_t v = v0 + j;
    const int c = c0 + tx;
    if (c < C && v < V) {
      if (out_bf16)
        reinterpret_cast<uint16_t*>(out)[obase + v * C + c] = (uint16_t)(pack_bf16x2(tile[tx][j], 0.f) & 0xffffu);
      else
        out[obase + v * C + c] = tile[tx][j];
    }
  }
}

// fp32 -> bf16 of a tensor that already is channels-last (the bf16 storage mode's input on the channels-last entry,
// holo_unet_forward_cl): 8 elements per thread, 2 x 16 bytes in, 16 bytes out
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float4* __restrict__ in, float4* __restrict__ out, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = in[2 * i], b = in[2 * i + 1];
    out[i] = make_float4(__uint_as_float(pack_bf16x2(a.x, a.y)), __uint_as_float(pack_bf16x2(a.z, a.w)),
                         __uint_as_float(pack_bf16x2(b.x, b.y)), __uint_as_float(pack_bf16x2(b.z, b.w)));
  }
}

__global__ __launch_bounds__(256) void ndhwc_to_ncdhw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             int C, int64_t V, int in_bf16) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z;
  const int64_t v0 = (int64_t)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t ibase = (int64_t)n * C * V;
  out += (int64_t)n * C * V;
  for (int j = ty; j < 32; j += 8) {
    const int64_t v = v0 + j;
    const int c = c0 + tx;
    float x = 0.f;
    if (c < C && v < V)
      x = in_bf16 ? __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(in)[ibase + v * C + c] << 16)
                  : in[ibase + v * C + c];
    tile[j][tx] = x;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j;
    const int64_t v = v0 + tx;
    if (c < C && v < V) out[(int64_t)c * V + v] = tile[tx][j];
  }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics, two deterministic stages (no atomics, fixed summation order):
//   stage 1 (gn_stats):    partial[n][b][c] = (sum, sum of squares) over voxel slab b, in double.
//                          x: [N][V][C] channels-last, C % 4 == 0, C/4 <= 256.  block = 256 threads:
//                          cq = C/4 threads across channels, rows = 256/cq voxels per pass; grid = (B, N).
//   stage 2 (gn_finalize): one block per (group, sample) reduces the partials of its channels over all
//                          slabs, then folds GroupNorm(eps) + affine (+ FiLM) into per-channel (a, b).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, double* __restrict__ partial,
                                                       int C, int64_t V, int vox_per_block, int x_bf16) {
  __shared__ double red[256 * 8];
  const int n = blockIdx.y;
  const int cq = C >> 2;
  const int rows = 256 / cq;
  const int tid = threadIdx.x;
  const int c4 = tid % cq;
  const int vr = tid / cq;
  const int64_t vbeg = (int64_t)blockIdx.x * vox_per_block;
  int64_t vend = vbeg + vox_per_block;
  if (vend > V) vend = V;
  const int64_t xbase = (int64_t)n * V * C;
  double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
  if (vr < rows) {
    float fs[4] = {0, 0, 0, 0}, fq[4] = {0, 0, 0, 0};
    int cnt = 0;
#pragma unroll 4
    for (int64_t v = vbeg + vr; v < vend; v += rows) {
      float4 t;
      if (x_bf16) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(x) + xbase + v * C + c4 * 4);
        t = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                        __uint_as_float(u.y & 0xffff0000u));
      } else {
        t = *reinterpret_cast<const float4*>(x + xbase + v * C + c4 * 4);
      }
      fs[0] += t.x;
      fs[1] += t.y;
      fs[2] += t.z;
      fs[3] += t.w;
      fq[0] += t.x * t.x;
      fq[1] += t.y * t.y;
      fq[2] += t.z * t.z;
      fq[3] += t.w * t.w;
      if (++cnt == 32) {  // fp32 partial sums stay short; long sums are carried in double
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ds[e] += fs[e];
          dq[e] += fq[e];
          fs[e] = 0.f;
          fq[e] = 0.f;
        }
        cnt = 0;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ds[e] += fs[e];
      dq[e] += fq[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[tid * 8 + e] = ds[e];
    red[tid * 8 + 4 + e] = dq[e];
  }
  __syncthreads();
  if (tid < cq) {
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < rows; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += red[(r * cq + tid) * 8 + e];
    double* dst = partial + (((int64_t)n * gridDim.x + blockIdx.x) * C + tid * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dst[e * 2 + 0] = s[e];
      dst[e * 2 + 1] = s[4 + e];
    }
  }
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  unsigned long long b;
  memcpy(&b, &v, 8);
  const float lo = __shfl_xor(__uint_as_float((uint32_t)(b & 0xffffffffull)), mask);
  const float hi = __shfl_xor(__uint_as_float((uint32_t)(b >> 32)), mask);
  b = (unsigned long long)__float_as_uint(lo) | ((unsigned long long)__float_as_uint(hi) << 32);
  double r;
  memcpy(&r, &b, 8);
  return r;
}

// coef[n][c] = (a, b) with GN(x)*(1+scale)+shift = a*x + b.  grid = (groups, N), block = 256
__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ part0, int C0, int B0,
                                                          const double* __restrict__ part1, int C1, int B1, int64_t V,
                                                          int groups, float eps, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ film, int film_stride,
                                                          int film_cout, float* __restrict__ coef,
                                                          float* __restrict__ moments) {
  __shared__ double rs[4], rq[4];
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int Cin = C0 + C1;
  const int cpg = Cin / groups;
  // the channel's affine / FiLM rows are requested FIRST: they come back under the reduction instead of after it (this
  // kernel is nothing but dependent round trips: ~66 launches per forward)
  float pg = 0.f, pb = 0.f, psc = 0.f, psh = 0.f;
  if (tid < cpg) {
    const int c = g * cpg + tid;
    pg = gamma[c];
    pb = beta[c];
    if (film) {
      psc = film[(int64_t)n * film_stride + c];
      psh = film[(int64_t)n * film_stride + film_cout + c];
    }
  }
  // Every (slab, channel-of-the-group) pair is one 16-byte (sum, sumsq) record; consecutive channels of a slab are
  // contiguous, so the threads walk the pairs with the channel index fastest (coalesced runs of cpg records) and keep
  // eight independent loads in flight.  The summation order depends on nothing but the launch geometry: deterministic.
  double s = 0.0, sq = 0.0;
  {
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    // the group may straddle the seam of the virtual concat: channels [c_lo, c_mid) come from part0, [c_mid, c_hi) from part1
    const int c_mid = c_lo < C0 ? (c_hi < C0 ? c_hi : C0) : c_lo;
    for (int half = 0; half < 2; ++half) {
      const int ca = half == 0 ? c_lo : c_mid, cb = half == 0 ? c_mid : c_hi;
      const int nch = cb - ca;
      if (nch <= 0) continue;
      const double* p = half == 0 && ca < C0 ? part0 : part1;
      const int Cs = p == part0 ? C0 : C1, B = p == part0 ? B0 : B1;
      const int cs0 = p == part0 ? ca : ca - C0;
      const int64_t total = (int64_t)B * nch;
      const double* base = p + ((int64_t)n * B * Cs + cs0) * 2;
      // record j of the run -> (slab j / nch, channel j % nch): 32-bit, and a shift / mask when the run is a power of two wide
      // (it is, except where a group straddles the seam of a concat): the 64-bit division this was costs ~150 instructions,
      // twice per record, in a kernel that is nothing but latency
      const bool pow2 = (nch & (nch - 1)) == 0;
      const int sh = 31 - __builtin_clz((unsigned)nch);
      auto rec = [&](int64_t j64) -> int64_t {
        const unsigned j = (unsigned)j64;  // (total < 2^31: gn_finalize_launch)
        const unsigned q = pow2 ? (j >> sh) : j / (unsigned)nch;
        const unsigned r = pow2 ? (j & (unsigned)(nch - 1)) : j - q * (unsigned)nch;
        return ((int64_t)q * Cs + r) * 2;
      };
      int64_t i = tid;
      // (sixteen loads in flight: the 2 048 x 2 records of a 64^3 tensor of the north-star net - 16 per thread - are ONE round
      //  trip instead of two; a thread adds its records in the same order whatever the batch width: bit-identical sums)
      for (; i + 15 * 256 < total; i += 16 * 256) {
        double2 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int64_t j = i + u * 256;
          v[u] = *reinterpret_cast<const double2*>(base + rec(j));
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          s += v[u].x;
          sq += v[u].y;
        }
      }
      for (; i + 7 * 256 < total; i += 8 * 256) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t j = i + u * 256;
          v[u] = *reinterpret_cast<const double2*>(base + rec(j));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          s += v[u].x;
          sq += v[u].y;
        }
      }
      for (; i < total; i += 256) {
        const double2 v = *reinterpret_cast<const double2*>(base + rec(i));
        s += v.x;
        sq += v.y;
      }
    }
  }
  // fixed-order butterfly inside each wave (no barrier), then the four wave totals in wave order: one barrier in all
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    s += shfl_xor_f64(s, o);
    sq += shfl_xor_f64(sq, o);
  }
  if ((tid & 63) == 0) {
    rs[tid >> 6] = s;
    rq[tid >> 6] = sq;
  }
  __syncthreads();
  if (tid < cpg) {
    const int c = g * cpg + tid;
    const double cnt = (double)cpg * (double)V;
    const double mean = (((rs[0] + rs[1]) + rs[2]) + rs[3]) / cnt;
    double var = (((rq[0] + rq[1]) + rq[2]) + rq[3]) / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    double a = rstd * (double)pg;
    double b = (double)pb - mean * a;
    if (film) {
      const double sc = 1.0 + (double)psc;
      const double sh = (double)psh;
      a *= sc;
      b = b * sc + sh;
    }
    coef[((int64_t)n * Cin + c) * 2 + 0] = (float)a;
    coef[((int64_t)n * Cin + c) * 2 + 1] = (float)b;
    if (moments) {  // training forward: the backward needs the group's mean and 1/std
      moments[((int64_t)n * Cin + c) * 2 + 0] = (float)mean;
      moments[((int64_t)n * Cin + c) * 2 + 1] = (float)rstd;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// time embedding: one block per sample.  dynamic-free: mc <= 256, ted <= 1024
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void time_embed_kernel(const int64_t* __restrict__ t, int mc, int ted,
                                                         const float* __restrict__ w1, const float* __restrict__ b1,
                                                         const float* __restrict__ w2, const float* __restrict__ b2,
                                                         float* __restrict__ emb, float* __restrict__ emb_silu,
                                                         int load_kind) {
  __shared__ float te[256];
  __shared__ float h1[1024];
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  // the caller's tensor, possibly just copied from the host: system-scope load (holo_ld_sys).  load_kind != 0 is a
  // DEVELOPMENT knob (HOLO_DEBUG_TIMESTEP_LOAD, scripts/h2d_stress_kinds.py) that reads it the two ways a compiler would:
  // 1 = wave-uniform address (s_load_dwordx2, through the scalar cache), 2 = per-lane address (global_load_dwordx2)
  float tv;
  if (load_kind == 1) {
    tv = (float)t[n];
  } else if (load_kind == 2) {
    const int64_t* tp = t + n;
    HOLO_LAUNDER(tp);  // the address in vector registers
    tv = (float)*tp;
  } else {
    tv = (float)holo_ld_sys(t + n);
  }
  const int half = mc / 2;
  for (int i = tid; i < mc; i += 256) {
    float v = 0.f;
    if (i < 2 * half) {
      const int k = i < half ? i : i - half;
      const float freq = expf((-9.210340371976184f * (float)k) / (float)half);
      const float arg = tv * freq;
      v = i < half ? cosf(arg) : sinf(arg);
    }
    te[i] = v;
  }
  __syncthreads();
  for (int j = tid; j < ted; j += 256) {
    float acc = 0.f;
    const float* w = w1 + (int64_t)j * mc;
    for (int k = 0; k < mc; ++k) acc = fmaf(w[k], te[k], acc);
    acc += b1[j];
    h1[j] = acc / (1.0f + expf(-acc));
  }
  __syncthreads();
  for (int j = tid; j < ted; j += 256) {
    float acc = 0.f;
    const float* w = w2 + (int64_t)j * ted;
    for (int k = 0; k < ted; ++k) acc = fmaf(w[k], h1[k], acc);
    acc += b2[j];
    emb[(int64_t)n * ted + j] = acc;
    emb_silu[(int64_t)n * ted + j] = acc / (1.0f + expf(-acc));
  }
}

// out[n][r] = bias[r] + W[r][:] . in[n][:]   one wave per (r, n)
__global__ __launch_bounds__(256) void rows_linear_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int rows, int K) {
  const int n = blockIdx.y;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const float* wr = w + (int64_t)r * K;
  const float* x = in + (int64_t)n * K;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc = fmaf(wr[k], x[k], acc);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) out[(int64_t)n * rows + r] = acc + bias[r];
}

// ---------------------------------------------------------------------------------------------
// DDPM ancestral update, float4 per thread.  grid = (blocks over per/4, batch)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ddpm_step_kernel(const float* __restrict__ tables, int T,
                                                        const int64_t* __restrict__ timesteps, int64_t per,
                                                        const float* __restrict__ x_t,
                                                        const float* __restrict__ model_out,
                                                        const float* __restrict__ noise, int clip,
                                                        float* __restrict__ sample, float* __restrict__ pred) {
  const int b = blockIdx.y;
  int64_t tt = holo_ld_sys(timesteps + b);
  if (tt < 0) tt = 0;
  if (tt >= T) tt = T - 1;
  const float c1 = holo_ld_sys(tables + tt * 4 + 0);  // (a 16 KB table uploaded from the host: see holo_ld_sys)
  const float c2 = holo_ld_sys(tables + tt * 4 + 1);
  const float lv = holo_ld_sys(tables + tt * 4 + 2);
  const float sig = tt != 0 ? expf(0.5f * lv) : 0.f;
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= per) return;
  const int64_t o = (int64_t)b * per + i;
  float4 x = *reinterpret_cast<const float4*>(x_t + o);
  float4 m = *reinterpret_cast<const float4*>(model_out + o);
  float4 e = *reinterpret_cast<const float4*>(noise + o);
  if (clip) {
    m.x = fminf(fmaxf(m.x, -1.f), 1.f);
    m.y = fminf(fmaxf(m.y, -1.f), 1.f);
    m.z = fminf(fmaxf(m.z, -1.f), 1.f);
    m.w = fminf(fmaxf(m.w, -1.f), 1.f);
  }
  float4 s;
  // reference order: (c1*x0 + c2*x) + nonzero*sigma*noise, each product rounded (no fma contraction)
  s.x = __fadd_rn(__fadd_rn(__fmul_rn(c1, m.x), __fmul_rn(c2, x.x)), __fmul_rn(sig, e.x));
  s.y = __fadd_rn(__fadd_rn(__fmul_rn(c1, m.y), __fmul_rn(c2, x.y)), __fmul_rn(sig, e.y));
  s.z = __fadd_rn(__fadd_rn(__fmul_rn(c1, m.z), __fmul_rn(c2, x.z)), __fmul_rn(sig, e.z));
  s.w = __fadd_rn(__fadd_rn(__fmul_rn(c1, m.w), __fmul_rn(c2, x.w)), __fmul_rn(sig, e.w));
  *reinterpret_cast<float4*>(sample + o) = s;
  *reinterpret_cast<float4*>(pred + o) = m;
}

// ---------------------------------------------------------------------------------------------
// The same update with the noise drawn IN the kernel (perf mode; gaussian_diffusion.py:498 `th.randn_like(x)`, :604): a
// thread's four elements take the four outputs of ONE Philox4x32-10 block - counter (element quad low, high, sample,
// stream offset), key (seed low, seed high) - turned into standard normals by two Box-Muller pairs.  A draw depends on
// nothing but (seed, offset, sample, element): bit-reproducible whatever the launch geometry; no generator state, no
// separate randn launch, and the 4 B / element of noise never cross HBM (optionally written out for tests).
// "Element" is the LOGICAL element: the quads are numbered in channels-last order (voxel, channel / 4).  NCDHW = 0: the
// tensors are in that order in memory (the sampler's channels-last chain) - quad = four consecutive floats.  NCDHW = 1:
// the tensors are (C, voxels) planes - a thread takes the same quad's four channels of one voxel from four planes
// (consecutive threads = consecutive voxels: coalesced), so both layouts of a chain draw the same noise.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}
// two uniforms -> two standard normals; u1 in (0, 1): 24 bits + half a step, so the logarithm is finite (|z| <= 5.9)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& z0, float& z1) {
  const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  z0 = r * cs;
  z1 = r * sn;
}

template <bool NCDHW>
__global__ __launch_bounds__(256) void ddpm_step_philox_kernel(const float* __restrict__ tables, int T,
                                                               const int64_t* __restrict__ timesteps, int64_t per,
                                                               const float* __restrict__ x_t,
                                                               const float* __restrict__ model_out, uint32_t seed_lo,
                                                               uint32_t seed_hi, uint32_t offset, int clip,
                                                               float* __restrict__ sample, float* __restrict__ pred,
                                                               float* __restrict__ noise_out, int channels) {
  const int b = blockIdx.y;
  int64_t tt = holo_ld_sys(timesteps + b);
  if (tt < 0) tt = 0;
  if (tt >= T) tt = T - 1;
  const float c1 = holo_ld_sys(tables + tt * 4 + 0);
  const float c2 = holo_ld_sys(tables + tt * 4 + 1);
  const float lv = holo_ld_sys(tables + tt * 4 + 2);
  const float sig = tt != 0 ? expf(0.5f * lv) : 0.f;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid * 4 >= per) return;
  int64_t qd = tid;                       // the canonical (channels-last) quad this thread draws for
  int64_t o = (int64_t)b * per + tid * 4;  // ... and where its first element lives
  int64_t es = 1;                          // distance between the quad's elements in memory
  if (NCDHW) {
    const int64_t V = per / channels;
    const int64_t cq = tid / V, v = tid - cq * V;
    qd = v * (channels >> 2) + cq;
    o = (int64_t)b * per + cq * 4 * V + v;
    es = V;
  }
  float4 x, m;
  if (NCDHW) {
    x = make_float4(x_t[o], x_t[o + es], x_t[o + 2 * es], x_t[o + 3 * es]);
    m = make_float4(model_out[o], model_out[o + es], model_out[o + 2 * es], model_out[o + 3 * es]);
  } else {
    x = *reinterpret_cast<const float4*>(x_t + o);
    m = *reinterpret_cast<const float4*>(model_out + o);
  }
  uint32_t rnd[4];
  philox4x32_10((uint32_t)qd, (uint32_t)((uint64_t)qd >> 32), (uint32_t)b, offset, seed_lo, seed_hi, rnd);
  float4 e;
  box_muller(rnd[0], rnd[1], e.x, e.y);
  box_muller(rnd[2], rnd[3], e.z, e.w);
  if (clip) {
    m.x = fminf(fmaxf(m.x, -1.f), 1.f);
    m.y = fminf(fmaxf(m.y, -1.f), 1.f);
    m.z = fminf(fmaxf(m.z, -1.f), 1.f);
    m.w = fminf(fmaxf(m.w, -1.f), 1.f);
  }
  float4 s;
  s.x = __fadd_rn(__fadd_rn(__fmul_rn(c1, m.x), __fmul_rn(c2, x.x)), __fmul_rn(sig, e.x));
  s.y = __fadd_rn(__fadd_rn(__fmul_rn(c1, m.y), __fmul_rn(c2, x.y)), __fmul_rn(sig, e.y));
  s.z = __fadd_rn(__fadd_rn(__fmul_rn(c1, m.z), __fmul_rn(c2, x.z)), __fmul_rn(sig, e.z));
  s.w = __fadd_rn(__fadd_rn(__fmul_rn(c1, m.w), __fmul_rn(c2, x.w)), __fmul_rn(sig, e.w));
  if (NCDHW) {
    sample[o] = s.x, sample[o + es] = s.y, sample[o + 2 * es] = s.z, sample[o + 3 * es] = s.w;
    if (pred) pred[o] = m.x, pred[o + es] = m.y, pred[o + 2 * es] = m.z, pred[o + 3 * es] = m.w;
    if (noise_out) noise_out[o] = e.x, noise_out[o + es] = e.y, noise_out[o + 2 * es] = e.z, noise_out[o + 3 * es] = e.w;
  } else {
    *reinterpret_cast<float4*>(sample + o) = s;
    if (pred) *reinterpret_cast<float4*>(pred + o) = m;
    if (noise_out) *reinterpret_cast<float4*>(noise_out + o) = e;
  }
}

// copy of a SMALL caller-provided tensor (biases, GroupNorm affine parameters, ...) with system-scope loads (holo_ld_sys)
__global__ __launch_bounds__(256) void copy_sys_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = holo_ld_sys(src + i);
}
__global__ __launch_bounds__(256) void tanh_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = tanhf(x[i]);
}
__global__ __launch_bounds__(256) void clip_kernel(const float* __restrict__ x, float* __restrict__ y, float lo,
                                                   float hi, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = fminf(fmaxf(x[i], lo), hi);
}

// OIDHW [Cout][Cin][taps] -> zero padded packed layout [tap][CinP/32][CoutP/16][half][kq][lj][4]
// (see wpack_block in kernels_conv.hip): element (tap, co, ci) with k = ci % 32 goes to
//   block(tap, ci/32, co/16) + (k>>2 & 1)*256 + ((k>>3)*16 + co%16)*4 + (k & 3)
__global__ __launch_bounds__(256) void repack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                                 int Cout, int Cin, int taps, int CoutP, int CinP) {
  const int64_t total = (int64_t)CoutP * CinP * taps;
  const int ncc = CinP >> 5, nsl = CoutP >> 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3);
    const int lj = (int)((i >> 2) & 15);
    const int kq = (int)((i >> 6) & 3);
    const int half = (int)((i >> 8) & 1);
    int64_t blk = i >> 9;
    const int slice = (int)(blk % nsl);
    blk /= nsl;
    const int cc = (int)(blk % ncc);
    const int tap = (int)(blk / ncc);
    const int co = slice * 16 + lj;
    const int ci = cc * 32 + kq * 8 + half * 4 + e;
    out[i] = (ci < Cin && co < Cout) ? holo_ld_sys(w + ((int64_t)co * Cin + ci) * taps + tap) : 0.f;  // (caller's tensor: holo_ld_sys)
  }
}

// Winograd-in-depth weights (conv_wino_kernel): pseudo-tap pt = xi*9 + ky*3 + kx of a 3x3x3 kernel is
// U_xi[ky][kx] = sum_kz G[xi][kz] w[kz][ky][kx] with G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1] (formed in double, rounded
// once); a 1x1x1 skip weight (centre tap) has the two non-zero pseudo-taps xi = 1, 2: +w/2, -w/2.  Packed layout as above.
__global__ __launch_bounds__(256) void repack_conv_weight_wino_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                                      int Cout, int Cin, int src_taps, int CoutP, int CinP,
                                                                      int dims) {
  const int taps = dims == 2 ? (src_taps == 27 ? 48 : 4) : (src_taps == 27 ? 36 : 2);
  const int64_t total = (int64_t)CoutP * CinP * taps;
  const int ncc = CinP >> 5, nsl = CoutP >> 4;
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3);
    const int lj = (int)((i >> 2) & 15);
    const int kq = (int)((i >> 6) & 3);
    const int half = (int)((i >> 8) & 1);
    int64_t blk = i >> 9;
    const int slice = (int)(blk % nsl);
    blk /= nsl;
    const int cc = (int)(blk % ncc);
    const int pt = (int)(blk / ncc);
    const int co = slice * 16 + lj;
    const int ci = cc * 32 + kq * 8 + half * 4 + e;
    float v = 0.f;
    if (ci < Cin && co < Cout) {
      const float* src = w + ((int64_t)co * Cin + ci) * src_taps;
      if (dims == 2) {
        if (src_taps == 27) {  // pt = (xi_z*4 + xi_y)*3 + kx
          const int kx = pt % 3, xy = (pt / 3) & 3, xz = pt / 12;
          double u = 0.0;
          for (int kz = 0; kz < 3; ++kz)
            for (int ky = 0; ky < 3; ++ky) u += G[xz][kz] * G[xy][ky] * (double)holo_ld_sys(src + kz * 9 + ky * 3 + kx);
          v = (float)u;
        } else {  // pt = (xi_z-1)*2 + (xi_y-1): G[xi][1] = +.5 (xi = 1), -.5 (xi = 2)
          v = ((pt >> 1) == (pt & 1) ? 0.25f : -0.25f) * holo_ld_sys(src);
        }
      } else if (src_taps == 27) {
        const int xi = pt / 9, kyx = pt - xi * 9;
        const double g0 = holo_ld_sys(src + kyx), g1 = holo_ld_sys(src + 9 + kyx), g2 = holo_ld_sys(src + 18 + kyx);
        const double u = xi == 0 ? g0 : xi == 1 ? 0.5 * (g0 + g1 + g2) : xi == 2 ? 0.5 * (g0 - g1 + g2) : g2;
        v = (float)u;
      } else {
        v = pt == 0 ? 0.5f * holo_ld_sys(src) : -0.5f * holo_ld_sys(src);
      }
    }
    out[i] = v;
  }
}

// OIDHW [Cout][Cin][taps] -> FOUR bf16 planes: (hi, mid, lo with w = hi + mid + lo exactly: hi = rne(w),
// mid = rne(w - hi), lo = rne(w - hi - mid)), each packed [tap][CinP/32][CoutP/16][lane = 16*kq + lj][8]: lane's 8
// values are channels 8*kq .. 8*kq+7 of the chunk for output channel 16*slice + lj (B operand of
// v_mfma_f32_16x16x32_bf16).  Plane 0 alone is the plain bf16 rounding used by the bf16 mode.
__global__ __launch_bounds__(256) void repack_conv_weight_bf16_kernel(const float* __restrict__ w,
                                                                      uint16_t* __restrict__ out, int Cout, int Cin,
                                                                      int taps, int CoutP, int CinP) {
  const int64_t total = (int64_t)CoutP * CinP * taps / 2;  // pairs per plane
  const int ncc = CinP >> 5, nsl = CoutP >> 4;
  uint32_t* o32 = reinterpret_cast<uint32_t*>(out);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int e2 = (int)(i & 3);  // pair index inside the lane's 8 values
    const int lane = (int)((i >> 2) & 63);
    int64_t blk = i >> 8;
    const int slice = (int)(blk % nsl);
    blk /= nsl;
    const int cc = (int)(blk % ncc);
    const int tap = (int)(blk / ncc);
    const int co = slice * 16 + (lane & 15);
    const int ci = cc * 32 + (lane >> 4) * 8 + e2 * 2;
    const float v0 = (ci < Cin && co < Cout) ? holo_ld_sys(w + ((int64_t)co * Cin + ci) * taps + tap) : 0.f;
    const float v1 = (ci + 1 < Cin && co < Cout) ? holo_ld_sys(w + ((int64_t)co * Cin + ci + 1) * taps + tap) : 0.f;
    const uint32_t h = pack_bf16x2(v0, v1);
    const float r0 = v0 - __uint_as_float(h << 16), r1 = v1 - __uint_as_float(h & 0xffff0000u);
    const uint32_t m = pack_bf16x2(r0, r1);
    const uint32_t l = pack_bf16x2(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xffff0000u));
    o32[i] = h;
    o32[total + i] = m;
    o32[2 * total + i] = l;
    // plane 3: the bf16 rounding again, packed for v_mfma_f32_32x32x16_bf16 (conv_bf16t_kernel):
    // [tap][CinP/16][CoutP/32][lane][8], lane's 8 values = channels 8*(lane>>5) .. +7 of the chunk, output channel lane&31
    {
      const int nc16 = CinP >> 4, ns32 = CoutP >> 5;
      int64_t b2 = i >> 8;
      const int sl2 = (int)(b2 % ns32);
      b2 /= ns32;
      const int cc2 = (int)(b2 % nc16);
      const int tap2 = (int)(b2 / nc16);
      const int co2 = sl2 * 32 + (lane & 31);
      const int ci2 = cc2 * 16 + (lane >> 5) * 8 + e2 * 2;
      const float u0 = (ci2 < Cin && co2 < Cout) ? w[((int64_t)co2 * Cin + ci2) * taps + tap2] : 0.f;
      const float u1 = (ci2 + 1 < Cin && co2 < Cout) ? w[((int64_t)co2 * Cin + ci2 + 1) * taps + tap2] : 0.f;
      o32[3 * total + i] = pack_bf16x2(u0, u1);
    }
  }
}

}  // namespace

int ncdhw_to_ndhwc_launch(const float* in, float* out, int N, int C, int64_t V, int tanh_flag, void* stream, int out_bf16) {
  dim3 grid((unsigned)cdiv(V, 32), (unsigned)cdiv(C, 32), (unsigned)N);
  HOLO_LAUNCH(ncdhw_to_ndhwc_kernel, grid, dim3(256), stream, in, out, C, V, tanh_flag, out_bf16);
  return 0;
}
int ndhwc_to_ncdhw_launch(const float* in, float* out, int N, int C, int64_t V, void* stream, int in_bf16) {
  dim3 grid((unsigned)cdiv(V, 32), (unsigned)cdiv(C, 32), (unsigned)N);
  HOLO_LAUNCH(ndhwc_to_ncdhw_kernel, grid, dim3(256), stream, in, out, C, V, in_bf16);
  return 0;
}

// slab decomposition shared by the planner (buffer sizes) and the launcher
void gn_stats_geometry(int C, int64_t V, int* n_blocks, int* vox_per_block) {
  const int cq = C >> 2;
  const int rows = 256 / cq;
  int64_t B = cdiv(V, rows);
  if (B > 256) B = 256;
  int64_t vpb = cdiv(cdiv(V, B), rows) * rows;
  *n_blocks = (int)cdiv(V, vpb);
  *vox_per_block = (int)vpb;
}

int gn_stats_launch(const float* x, double* partial, int N, int C, int64_t V, void* stream, int x_bf16) {
  if ((C & 3) || (C >> 2) > 256) {
    set_error("gn_stats: unsupported C=%d", C);
    return -1;
  }
  int B, vpb;
  gn_stats_geometry(C, V, &B, &vpb);
  dim3 grid((unsigned)B, (unsigned)N);
  HOLO_LAUNCH(gn_stats_kernel, grid, dim3(256), stream, x, partial, C, V, vpb, x_bf16);
  return 0;
}

int gn_finalize_launch(const double* part0, int C0, int B0, const double* part1, int C1, int B1, int N, int64_t V,
                       int groups, float eps, const float* gamma, const float* beta, const float* film,
                       int film_stride, int film_cout, float* coef, void* stream, float* moments) {
  const int Cin = C0 + C1;
  if (Cin % groups || Cin / groups > 256) {
    set_error("gn_finalize: C=%d not compatible with %d groups", Cin, groups);
    return -1;
  }
  dim3 grid((unsigned)groups, (unsigned)N);
  HOLO_LAUNCH(gn_finalize_kernel, grid, dim3(256), stream, part0, C0, B0, part1, C1, B1, V, groups, eps, gamma, beta,
              film, film_stride, film_cout, coef, moments);
  return 0;
}

int time_embed_launch(const int64_t* t, int N, int mc, int ted, const float* w1, const float* b1, const float* w2,
                      const float* b2, float* emb, float* emb_silu, void* stream) {
  if (mc > 256 || ted > 1024) {
    set_error("time_embed: model_channels=%d too large", mc);
    return -1;
  }
  const char* lk = getenv("HOLO_DEBUG_TIMESTEP_LOAD");  // development knob, see the kernel
  HOLO_LAUNCH(time_embed_kernel, dim3((unsigned)N), dim3(256), stream, t, mc, ted, w1, b1, w2, b2, emb, emb_silu,
              lk ? atoi(lk) : 0);
  return 0;
}

int rows_linear_launch(const float* in, const float* w, const float* bias, float* out, int N, int rows, int K,
                       void* stream) {
  dim3 grid((unsigned)cdiv(rows, 4), (unsigned)N);
  HOLO_LAUNCH(rows_linear_kernel, grid, dim3(256), stream, in, w, bias, out, rows, K);
  return 0;
}

int ddpm_step_launch(const float* tables, int T, const int64_t* timesteps, int batch, int64_t per, const float* x_t,
                     const float* model_out, const float* noise, int clip, float* sample, float* pred_xstart,
                     void* stream) {
  if (per & 3) {
    set_error("ddpm_step: elems_per_sample must be a multiple of 4");
    return -1;
  }
  dim3 grid((unsigned)cdiv(per / 4, 256), (unsigned)batch);
  HOLO_LAUNCH(ddpm_step_kernel, grid, dim3(256), stream, tables, T, timesteps, per, x_t, model_out, noise, clip, sample,
              pred_xstart);
  return 0;
}

int ddpm_step_philox_launch(const float* tables, int T, const int64_t* timesteps, int batch, int64_t per, const float* x_t,
                            const float* model_out, uint64_t seed, uint64_t offset, int clip, float* sample,
                            float* pred_xstart, float* noise_out, int ncdhw_channels, void* stream) {
  if (per & 3) {
    set_error("ddpm_step: elems_per_sample must be a multiple of 4");
    return -1;
  }
  if (ncdhw_channels < 0 || (ncdhw_channels & 3) || (ncdhw_channels > 0 && per % ncdhw_channels)) {
    set_error("ddpm_step_philox: ncdhw_channels must be 0 (channels-last tensors) or a multiple of 4 that divides elems_per_sample");
    return -1;
  }
  dim3 grid((unsigned)cdiv(per / 4, 256), (unsigned)batch);
  // (the high half of the offset goes into the key: counters stay distinct for any 64-bit stream offset)
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32);
  if (ncdhw_channels)
    HOLO_LAUNCH(ddpm_step_philox_kernel<true>, grid, dim3(256), stream, tables, T, timesteps, per, x_t, model_out, k0, k1,
                (uint32_t)offset, clip, sample, pred_xstart, noise_out, ncdhw_channels);
  else
    HOLO_LAUNCH(ddpm_step_philox_kernel<false>, grid, dim3(256), stream, tables, T, timesteps, per, x_t, model_out, k0, k1,
                (uint32_t)offset, clip, sample, pred_xstart, noise_out, 0);
  return 0;
}

int f32_to_bf16_launch(const float* in, float* out_bf16, int64_t n, void* stream) {
  if (n & 7) {
    set_error("f32_to_bf16: the element count must be a multiple of 8");
    return -1;
  }
  int64_t blocks = cdiv(n >> 3, 256);
  if (blocks > 16384) blocks = 16384;
  HOLO_LAUNCH(f32_to_bf16_kernel, dim3((unsigned)blocks), dim3(256), stream, reinterpret_cast<const float4*>(in),
              reinterpret_cast<float4*>(out_bf16), n >> 3);
  return 0;
}

int copy_sys_launch(const float* src, float* dst, int64_t n, void* stream) {
  int64_t blocks = cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  HOLO_LAUNCH(copy_sys_kernel, dim3((unsigned)blocks), dim3(256), stream, src, dst, n);
  return 0;
}
int tanh_launch(const float* x, float* y, int64_t n, void* stream) {
  int64_t blocks = cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  HOLO_LAUNCH(tanh_kernel, dim3((unsigned)blocks), dim3(256), stream, x, y, n);
  return 0;
}
int clip_launch(const float* x, float* y, float lo, float hi, int64_t n, void* stream) {
  int64_t blocks = cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  HOLO_LAUNCH(clip_kernel, dim3((unsigned)blocks), dim3(256), stream, x, y, lo, hi, n);
  return 0;
}
int repack_conv_weight_bf16_launch(const float* w, uint16_t* out, int Cout, int Cin, int taps, int CoutP, int CinP,
                                   void* stream) {
  int64_t total = (int64_t)CoutP * CinP * taps / 2;
  int64_t blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  HOLO_LAUNCH(repack_conv_weight_bf16_kernel, dim3((unsigned)blocks), dim3(256), stream, w, out, Cout, Cin, taps, CoutP,
              CinP);
  return 0;
}
int repack_conv_weight_wino_launch(const float* w, float* out, int Cout, int Cin, int src_taps, int CoutP, int CinP,
                                   void* stream, int dims) {
  if ((src_taps != 27 && src_taps != 1) || (dims != 1 && dims != 2)) {
    set_error("repack_conv_weight_wino: 27 or 1 source taps, 1 or 2 transformed dimensions");
    return -1;
  }
  const int taps = dims == 2 ? (src_taps == 27 ? 48 : 4) : (src_taps == 27 ? 36 : 2);
  int64_t total = (int64_t)CoutP * CinP * taps;
  int64_t blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  HOLO_LAUNCH(repack_conv_weight_wino_kernel, dim3((unsigned)blocks), dim3(256), stream, w, out, Cout, Cin, src_taps,
              CoutP, CinP, dims);
  return 0;
}
int repack_conv_weight_launch(const float* w, float* out, int Cout, int Cin, int taps, int CoutP, int CinP,
                              void* stream) {
  int64_t total = (int64_t)CoutP * CinP * taps;
  int64_t blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  HOLO_LAUNCH(repack_conv_weight_kernel, dim3((unsigned)blocks), dim3(256), stream, w, out, Cout, Cin, taps, CoutP,
              CinP);
  return 0;
}

}  // namespace holo
