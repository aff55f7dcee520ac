// kernels_bwd.hip — backward kernels of the denoiser (second half): gradients of UNetModel.forward
// (holo_diffusion/guided_diffusion/unet.py:800-837) with respect to its input and every parameter, what
// `output.mean().backward()` computes in the reference's own backward test (holo_diffusion/tests/test_diffusion_utils.py:47-66).
//
// All tensors are the forward's channels-last fp32 activations [n][z][y][x][c].  The forward never stores the ACTIVATED
// operand of a convolution (GroupNorm . FiLM . SiLU is applied while the tile is staged); the backward recomputes it the
// same way from the raw tensor and the (a, b) coefficients of the forward pass.
//
//   conv_wgrad_kernel        dW[co][ci][tap] = sum_m gy[m][co] act(x)[m + tap][ci] on v_mfma_f32_32x32x2_f32 with the VOXELS
//                            as the K dimension: A = gy^T (lane: output channel li, voxel of the k pair), B = act(x) (lane:
//                            input channel li) - both operands are 128-byte coalesced rows of the channels-last tensors and
//                            go global -> register; one wave owns one (32 co, 32 ci, tap) tile over a slab of voxels;
//                            slabs are summed by wgrad_reduce_kernel (deterministic, no atomics)
//   dgrad of stride-1 convs  is the forward conv kernel itself on flipped / transposed weights (flip_transpose_weight_kernel)
//   conv_dgrad_s2_kernel     transposed 3x3x3 stride-2 convolution (Downsample.op)
//   sumpool2_kernel          backward of the nearest x2 upsampling that Upsample applies on load
//   gn_bwd_*                 GroupNorm32 + FiLM + SiLU backward: per-(n, channel) sums S1 = sum gu, S2 = sum gu xhat (two
//                            deterministic stages, double), group terms + parameter / FiLM gradients, then the elementwise
//                            gx = rstd (gu g' - m1 - xhat m2), accumulated into the gradient of the raw tensor
//   attn_ds_kernel, transpose_kernel   softmax backward rows and the T x T transposition around the batched fp32 MFMA GEMMs
//   colsum_*                 bias gradients;  time_embed_bwd_kernel, film_bwd_* the (tiny) embedding path
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {
namespace {

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + __expf(-v)); }
__device__ __forceinline__ float silu_b(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ float dsilu(float u) {
  const float s = sigm(u);
  return s * (1.0f + u * (1.0f - s));
}

// out[ci][co][T-1-t] = in[co][ci][t]  (OIDHW -> the weight of the transposed convolution, again OIDHW)
__global__ __launch_bounds__(256) void flip_transpose_weight_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                   int Co, int Ci, int T) {
  const int64_t total = (int64_t)Co * Ci * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int ci = (int)((i / T) % Ci);
    const int co = (int)(i / ((int64_t)T * Ci));
    out[((int64_t)ci * Co + co) * T + (T - 1 - t)] = holo_ld_sys(in + i);  // (caller's tensor: holo_ld_sys)
  }
}

// ---- weight gradient -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
  __shared__ float s_red[4][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int Cin = p.C0 + p.C1;
  const int nct = (p.Cout + 31) / 32, nit = (Cin + 31) / 32;
  int b = blockIdx.x;
  const int it = b % nit;
  b /= nit;
  const int ct = b % nct;
  const int tap = b / nct;
  int kd = 0, kh = 0, kw = 0;
  if (p.ksz == 3) {
    kd = tap / 9;
    kh = (tap - kd * 9) / 3;
    kw = tap - kd * 9 - kh * 3;
  }
  const int co = ct * 32 + li, ci = it * 32 + li;
  const bool co_ok = co < p.Cout, ci_ok = ci < Cin;
  const float* src = p.src0;
  int Cs = p.C0, cs = ci_ok ? ci : 0;
  if (cs >= p.C0) {
    src = p.src1;
    Cs = p.C1;
    cs -= p.C0;
  }
  const int SD = p.ups ? (p.ID >> 1) : p.ID, SH = p.ups ? (p.IH >> 1) : p.IH, SW = p.ups ? (p.IW >> 1) : p.IW;
  // rows (n, od, oh) of the output; a slab (blockIdx.y) is a contiguous range of rows, a wave takes every 4th row of it
  const int64_t nrows = (int64_t)p.N * p.OD * p.OH;
  const int64_t r_lo = nrows * blockIdx.y / gridDim.y, r_hi = nrows * (blockIdx.y + 1) / gridDim.y;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int64_t row = r_lo + wave; row < r_hi; row += 4) {
    const int oh = (int)(row % p.OH);
    const int od = (int)((row / p.OH) % p.OD);
    const int n = (int)(row / ((int64_t)p.OH * p.OD));
    int z = od * p.stride - p.pad + kd, y = oh * p.stride - p.pad + kh;
    const bool zy_ok = z >= 0 && z < p.ID && y >= 0 && y < p.IH;
    if (!zy_ok) continue;  // wave-uniform: the whole input row is padding
    if (p.ups) {
      z >>= 1;
      y >>= 1;
    }
    float ca = 1.f, cb = 0.f;
    if (p.coef) {
      ca = p.coef[((int64_t)n * Cin + (ci_ok ? ci : 0)) * 2 + 0];
      cb = p.coef[((int64_t)n * Cin + (ci_ok ? ci : 0)) * 2 + 1];
    }
    const float* gyrow = p.gy + (((int64_t)n * p.OD + od) * p.OH + oh) * (int64_t)p.OW * p.Cout + (co_ok ? co : 0);
    const float* xrow = src + (((int64_t)n * SD + z) * SH + y) * (int64_t)SW * Cs + cs;
    // k pairs: output voxels ow = 2 j + lh
    for (int j0 = 0; j0 < p.OW; j0 += 16) {
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int ow = j0 + 2 * u + lh;
        int x = ow * p.stride - p.pad + kw;
        const bool ok = ow < p.OW && x >= 0 && x < p.IW;
        if (p.ups) x >>= 1;
        x = min(max(x, 0), SW - 1);
        const float g = gyrow[(int64_t)min(ow, p.OW - 1) * p.Cout];
        float v = xrow[(int64_t)x * Cs];
        if (p.coef) {
          v = fmaf(v, ca, cb);
          if (p.act) v = silu_b(v);
        }
        av[u] = (ok && co_ok) ? g : 0.f;   // zero padding AFTER the activation, zero rows beyond the tensors
        bv[u] = (ok && ci_ok) ? v : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
  }
  // the four waves' tiles are summed through LDS; D layout: column = ci tile index li, rows = co (r&3) + 8 (r>>2) + 4 lh
#pragma unroll
  for (int r = 0; r < 16; ++r) s_red[wave][((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[r];
  __syncthreads();
  for (int i = tid; i < 32 * 32; i += 256) {
    const int row = i >> 5, col = i & 31;
    const float v = s_red[0][row * 33 + col] + s_red[1][row * 33 + col] + s_red[2][row * 33 + col] + s_red[3][row * 33 + col];
    const int oc = ct * 32 + row, ic = it * 32 + col;
    if (oc < p.Cout && ic < Cin)
      p.partial[(((int64_t)blockIdx.y * p.Cout + oc) * Cin + ic) * p.ntaps + tap] = v;
  }
}

// ---- weight gradient, row-staged form (3x3x3, stride 1, pad 1, OW <= 64) -----------------------------------------------
// The tile kernel above re-reads gy and x once per (tap, channel-tile pair): 27 x more fabric traffic than the tensors.
// Here a workgroup owns one (32 co, 32 ci) pair for ALL 27 taps over a slab of output rows (n, od, oh): per output row the
// gy row [OW][32] and the nine ACTIVATED input rows (kd, kh) [OW + 2][32] sit in LDS (84 KB at OW = 64; a ring of three y
// rows per kd, so a step along oh brings three new rows, requested under the previous row's MFMAs); wave w accumulates taps
// w, w + 4, ... (7 x 16 accumulator registers): per voxel pair one A operand (gy) and seven B operands (x at the tap's
// shift) from LDS, both bank-conflict free (a voxel is one 128-byte row, the pair's two voxels cover the 64 banks).
// Three shapes: rows up to 64 voxels - 8 waves (4 / 3 taps each), 84 KB: one workgroup per CU, two waves per SIMD;
// up to 32 and up to 16 - 4 waves (7 / 6 taps), 43 / 23 KB: two workgroups per CU (168 registers would spill).
constexpr int WR_MAXW = 64;
template <int MAXW>
struct WgradRowLds {
  float gy[MAXW * 32];
  float x[9][(MAXW + 2) * 32];
};
template <int MAXW, int NW>
__global__ __launch_bounds__((64 * NW), 2) void conv_wgrad_rows_kernel(WgradParams p) {
  constexpr int NT = 64 * NW;              // threads
  constexpr int TPW = (27 + NW - 1) / NW;  // taps per wave
  __shared__ __attribute__((aligned(16))) WgradRowLds<MAXW> S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int Cin = p.C0 + p.C1;
  const int nit = (Cin + 31) / 32;
  const int it = blockIdx.x % nit, ct = blockIdx.x / nit;
  const int OW = p.OW, XW = OW + 2;
  const int SD = p.ups ? (p.ID >> 1) : p.ID, SH = p.ups ? (p.IH >> 1) : p.IH, SW = p.ups ? (p.IW >> 1) : p.IW;
  // the ci tile lies inside one source of the virtual concat (C0 % 32 == 0)
  const float* src = p.src0;
  int Cs = p.C0, cs0 = it * 32;
  if (cs0 >= p.C0) {
    src = p.src1;
    Cs = p.C1;
    cs0 -= p.C0;
  }
  const int64_t nrows = (int64_t)p.N * p.OD * p.OH;
  const int64_t r_lo = nrows * blockIdx.y / gridDim.y, r_hi = nrows * (blockIdx.y + 1) / gridDim.y;
  // staging items: (voxel, 4-channel group g) as float4; NT % 8 == 0, so a thread's items all have g = tid & 7 and ONE set
  // of (a, b) coefficients per sample serves them
  constexpr int GYI = (MAXW * 8 + NT - 1) / NT;
  constexpr int XI = ((MAXW + 2) * 8 * 3 + NT - 1) / NT;  // three rows per step
  const int g = tid & 7;
  const bool g_in = it * 32 + g * 4 < Cin, g_out = ct * 32 + g * 4 < p.Cout;
  const int gc_in = min(cs0 + g * 4, Cs - 4), gc_out = min(ct * 32 + g * 4, p.Cout - 4);
  float4 rg[GYI], rx[XI];
  float4 c01 = make_float4(1.f, 0.f, 1.f, 0.f), c23 = c01;
  int coef_n = -1;
  auto gy_issue = [&](int n, int od, int oh) {
    const float* row = p.gy + (((int64_t)n * p.OD + od) * p.OH + oh) * (int64_t)OW * p.Cout + gc_out;
#pragma unroll
    for (int i = 0; i < GYI; ++i) {
      const int v = min((tid + NT * i) >> 3, OW - 1);
      rg[i] = *reinterpret_cast<const float4*>(row + (int64_t)v * p.Cout);
    }
  };
  auto gy_commit = [&]() {
#pragma unroll
    for (int i = 0; i < GYI; ++i) {
      const int v = (tid + NT * i) >> 3;
      if (v < OW) *reinterpret_cast<float4*>(S.gy + v * 32 + g * 4) = g_out ? rg[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // x rows (kd = 0..2) at input row y of plane od (virtual coordinates z = od + kd - 1): item = (kd, voxel)
  auto x_issue = [&](int n, int od, int y) {
    int yy = min(max(y, 0), p.IH - 1);
    if (p.ups) yy >>= 1;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int id = min((tid + NT * i) >> 3, 3 * XW - 1);
      const int kd = id / XW, v = id - kd * XW;
      int z = min(max(od + kd - 1, 0), p.ID - 1), xx = min(max(v - 1, 0), p.IW - 1);
      if (p.ups) z >>= 1, xx >>= 1;
      rx[i] = *reinterpret_cast<const float4*>(src + ((((int64_t)n * SD + z) * SH + yy) * SW + xx) * Cs + gc_in);
    }
  };
  auto x_commit = [&](int n, int od, int y) {
    const int ys = ((y % 3) + 3) % 3;
    if (p.coef && n != coef_n) {
      const float* cf = p.coef + ((int64_t)n * Cin + min(it * 32 + g * 4, Cin - 4)) * 2;
      c01 = *reinterpret_cast<const float4*>(cf);
      c23 = *reinterpret_cast<const float4*>(cf + 4);
      coef_n = n;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int id = (tid + NT * i) >> 3;
      if (id >= 3 * XW) continue;
      const int kd = id / XW, v = id - kd * XW;
      const int z = od + kd - 1, xx = v - 1;
      const bool inside = z >= 0 && z < p.ID && y >= 0 && y < p.IH && xx >= 0 && xx < p.IW && g_in;
      float4 t = rx[i];
      if (p.coef) {
        t.x = fmaf(t.x, c01.x, c01.y), t.y = fmaf(t.y, c01.z, c01.w), t.z = fmaf(t.z, c23.x, c23.y), t.w = fmaf(t.w, c23.z, c23.w);
        if (p.act) t.x = silu_b(t.x), t.y = silu_b(t.y), t.z = silu_b(t.z), t.w = silu_b(t.w);
      }
      if (!inside) t = make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding AFTER the activation
      *reinterpret_cast<float4*>(S.x[kd * 3 + ys] + v * 32 + g * 4) = t;
    }
  };
  f32x16 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  int toff[TPW], tkd3[TPW], tkh[TPW];  // per tap of this wave: 32 kw, 3 kd, kh
  int ntap = 0;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int tap = min(wave + NW * j, 26);
    const int kd = tap / 9, kh = (tap - kd * 9) / 3, kw = tap - kd * 9 - kh * 3;
    toff[j] = kw * 32;
    tkd3[j] = kd * 3;
    tkh[j] = kh;
    if (wave + NW * j < 27) ntap = j + 1;
  }

  bool have_plane = false;
  for (int64_t row = r_lo; row < r_hi; ++row) {
    const int oh = (int)(row % p.OH);
    const int od = (int)((row / p.OH) % p.OD);
    const int n = (int)(row / ((int64_t)p.OH * p.OD));
    if (!have_plane || oh == 0) {  // first row of the slab or of a plane: all nine rows
      __syncthreads();
      for (int y = oh - 1; y <= oh + 1; ++y) {
        x_issue(n, od, y);
        x_commit(n, od, y);
      }
      gy_issue(n, od, oh);
      gy_commit();
      __syncthreads();
      have_plane = true;
    }
    const bool nxt = row + 1 < r_hi && oh + 1 < p.OH;  // the next row continues this plane: request its new data now
    if (nxt) {
      x_issue(n, od, oh + 2);
      gy_issue(n, od, oh + 1);
    }
    // ---- MFMAs of this row: voxel pairs (2u, 2u + 1); A = gy[voxel][co li], B = x[voxel + kw][ci li]
    const float* xb[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) xb[j] = S.x[tkd3[j] + ((oh + tkh[j] - 1) % 3 + 3) % 3] + toff[j] + lh * 32 + li;
    const float* ga = S.gy + lh * 32 + li;
    // (two straight-line bodies - TPW or TPW - 1 taps - chosen once per row: a guard per MFMA inside the loop would keep
    // the compiler from hoisting the next pair's LDS reads above the current pair's MFMAs)
    auto pairs = [&](auto nt) {
      constexpr int NTAP = decltype(nt)::value;
      // software pipelined by hand: the operands of pair u + 1 are requested before the MFMAs of pair u are issued
      const int np = OW / 2;
      float av = ga[0], bv[NTAP];
#pragma unroll
      for (int j = 0; j < NTAP; ++j) bv[j] = xb[j][0];
#pragma unroll 2
      for (int u = 0; u < np; ++u) {
        const int un = u + 1 < np ? u + 1 : u;
        const float av_n = ga[un * 64];
        float bv_n[NTAP];
#pragma unroll
        for (int j = 0; j < NTAP; ++j) bv_n[j] = xb[j][un * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NTAP; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        av = av_n;
#pragma unroll
        for (int j = 0; j < NTAP; ++j) bv[j] = bv_n[j];
      }
    };
    if (ntap == TPW)
      pairs(std::integral_constant<int, TPW>());
    else
      pairs(std::integral_constant<int, TPW - 1>());
    if (nxt) {
      __syncthreads();  // every wave is done with this row's operands
      x_commit(n, od, oh + 2);
      gy_commit();
      __syncthreads();
    }
  }
  // D layout: column = ci li, rows = co (r&3) + 8 (r>>2) + 4 lh
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    if (j >= ntap) continue;
    const int tap = wave + NW * j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int oc = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, ic = it * 32 + li;
      // tap-major partials [slab][tap][co][ci]: a register's 32 lanes write one 128-byte row (wgrad_reduce_t_kernel
      // transposes to OIDHW while it sums the slabs)
      if (oc < p.Cout && ic < Cin) p.partial[(((int64_t)blockIdx.y * p.ntaps + tap) * p.Cout + oc) * Cin + ic] = acc[j][r];
    }
  }
}

// dW[co][ci][tap] (+)= sum_s partial[s][tap][co][ci]   (threads walk the partials' layout: coalesced reads)
__global__ __launch_bounds__(256) void wgrad_reduce_t_kernel(const float* __restrict__ partial, float* __restrict__ dw, int Cout,
                                                            int Cin, int ntaps, int splits, int accumulate) {
  const int64_t per_tap = (int64_t)Cout * Cin, n = per_tap * ntaps;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i / per_tap);
    const int64_t cc = i - (int64_t)tap * per_tap;  // co * Cin + ci
    double s = 0.0;
    int k = 0;
    for (; k + 7 < splits; k += 8) {  // eight loads in flight, added in slab order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(k + u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; k < splits; ++k) s += (double)partial[(int64_t)k * n + i];
    float* d = dw + cc * ntaps + tap;
    *d = (accumulate ? *d : 0.f) + (float)s;
  }
}

// The same for LARGE weights (>= 1 024 tiles of 64 (co, ci) pairs): a workgroup owns 64 consecutive pairs - it reads their 27
// tap rows of every slab (256-byte runs; a thread's seven taps are seven independent loads), sums the slabs in slab order, turns
// the 64 x 27 tile round in LDS and writes the pairs' taps as ONE contiguous run instead of 4-byte pieces 108 bytes apart
// (measured: 134 against 220 us for the largest launches; for small weights the one-element-per-thread form above keeps more
// loads in flight and wins: 45 against 64 us on average over a training step's 60 launches)
__global__ __launch_bounds__(256) void wgrad_reduce_tile_kernel(const float* __restrict__ partial, float* __restrict__ dw, int Cout,
                                                               int Cin, int ntaps, int splits, int accumulate) {
  __shared__ float s_tile[64 * 28];
  const int64_t per_tap = (int64_t)Cout * Cin, n = per_tap * ntaps;
  const int c = threadIdx.x & 63, tq = threadIdx.x >> 6;
  const int64_t ntiles = (per_tap + 63) / 64;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t cc0 = tile * 64;
    const int ncc = per_tap - cc0 < 64 ? (int)(per_tap - cc0) : 64;
    {
      double s[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      const float* src = partial + (int64_t)tq * per_tap + cc0 + (c < ncc ? c : 0);
      for (int k = 0; k < splits; ++k) {
        float v[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) v[j] = tq + 4 * j < ntaps ? src[(int64_t)k * n + (int64_t)(4 * j) * per_tap] : 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) s[j] += (double)v[j];
      }
#pragma unroll
      for (int j = 0; j < 7; ++j)
        if (tq + 4 * j < ntaps) s_tile[c * 28 + tq + 4 * j] = (float)s[j];
    }
    __syncthreads();
    const int cnt = ncc * ntaps;
    float* d = dw + cc0 * ntaps;
    for (int j = threadIdx.x; j < cnt; j += 256) {
      const int cc = j / ntaps, tap = j - cc * ntaps;
      d[j] = (accumulate ? d[j] : 0.f) + s_tile[cc * 28 + tap];
    }
    __syncthreads();
  }
}

// dW[i] (+)= sum_s partial[s][i]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int64_t n,
                                                          int splits, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    int k = 0;
    for (; k + 7 < splits; k += 8) {  // eight loads in flight, added in slab order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(k + u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; k < splits; ++k) s += (double)partial[(int64_t)k * n + i];
    dw[i] = (accumulate ? dw[i] : 0.f) + (float)s;
  }
}

// ---- column sums (bias gradients): part[b][c] = sum over the block's rows; then reduced
// threads = (channel, sub-row): with C < 256 the 256 / C sub-rows of a thread block stride the rows together and are summed
// through LDS in a fixed order
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ g, double* __restrict__ part, int64_t M, int C) {
  __shared__ double red[256];
  const int64_t rows_per = (M + gridDim.x - 1) / gridDim.x;
  const int64_t m0 = (int64_t)blockIdx.x * rows_per, m1 = m0 + rows_per < M ? m0 + rows_per : M;
  const int nsub = C >= 256 ? 1 : 256 / C;
  const int sub = threadIdx.x / C, c0 = threadIdx.x - sub * C;
  for (int cb = 0; cb < C; cb += 256) {
    const int c = cb + c0;
    double s = 0.0;
    if (sub < nsub && c < C) {
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
      int64_t m = m0 + sub;
      for (; m + 3 * nsub < m1; m += 4 * nsub) {  // 4 independent loads in flight
#pragma unroll
        for (int u = 0; u < 4; ++u) s4[u] = g[(m + u * nsub) * C + c];
        s += ((double)s4[0] + (double)s4[1]) + ((double)s4[2] + (double)s4[3]);
      }
      for (; m < m1; m += nsub) s += (double)g[m * C + c];
    }
    if (nsub > 1) {
      red[threadIdx.x] = s;
      __syncthreads();
      if (sub == 0 && c < C)
        for (int k = 1; k < nsub; ++k) s += red[k * C + c0];
      __syncthreads();
    }
    if (sub == 0 && c < C) part[(int64_t)blockIdx.x * C + c] = s;
  }
}
// one block per 32 channels: 8 lanes per channel stride the partials, summed through LDS in a fixed order
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ part, float* __restrict__ out, int nb, int C,
                                                          int accumulate) {
  __shared__ double red[256];
  const int cl = threadIdx.x & 31, sub = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0;
  if (c < C) {  // eight loads in flight, added in slab order (this kernel is nothing but their latency: 104 launches per step)
    int b = sub;
    for (; b + 56 < nb; b += 64) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(b + 8 * u) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nb; b += 8) s += part[(int64_t)b * C + c];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (sub == 0 && c < C) {
    for (int k = 1; k < 8; ++k) s += red[k * 32 + cl];
    out[c] = (accumulate ? out[c] : 0.f) + (float)s;
  }
}

// ---- GroupNorm + FiLM + SiLU backward --------------------------------------------------------------------------------
// u = a x + b (the forward's coefficients), act: SiLU.  gu = ga * silu'(u) (or ga).  xhat = (x - mean) rstd.
// stage 1: part[blk][n][c] = (sum gu, sum gu xhat) over the block's voxels
__global__ __launch_bounds__(256) void gn_bwd_part_kernel(GnBwdParams p) {
  __shared__ double red[2][256];
  const int Cin = p.C0 + p.C1;
  const int n = blockIdx.y;
  const int64_t v_per = (p.V + gridDim.x - 1) / gridDim.x;
  const int64_t v0 = (int64_t)blockIdx.x * v_per, v1 = v0 + v_per < p.V ? v0 + v_per : p.V;
  // threads = (channel, sub-voxel): with Cin < 256 the 256 / Cin sub-voxels stride the block's voxels together
  const int nsub = Cin >= 256 ? 1 : 256 / Cin;
  const int sub = threadIdx.x / Cin, c0 = threadIdx.x - sub * Cin;
  for (int cb = 0; cb < Cin; cb += 256) {
    const int c = cb + c0;
    const bool live = sub < nsub && c < Cin;
    double s1 = 0.0, s2 = 0.0;
    if (live) {
      const float* src = c < p.C0 ? p.x0 + c : p.x1 + (c - p.C0);
      const int Cs = c < p.C0 ? p.C0 : p.C1;
      const float a = p.coef[((int64_t)n * Cin + c) * 2], b = p.coef[((int64_t)n * Cin + c) * 2 + 1];
      const float mean = p.mom[((int64_t)n * Cin + c) * 2], rstd = p.mom[((int64_t)n * Cin + c) * 2 + 1];
      int64_t v = v0 + sub;
      for (; v + 3 * nsub < v1; v += 4 * nsub) {  // 8 independent loads in flight
        float xs[4], gs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          xs[u] = src[((int64_t)n * p.V + v + u * nsub) * Cs];
          gs[u] = p.ga[((int64_t)n * p.V + v + u * nsub) * Cin + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (p.act) gs[u] *= dsilu(fmaf(xs[u], a, b));
          s1 += (double)gs[u];
          s2 += (double)(gs[u] * ((xs[u] - mean) * rstd));
        }
      }
      for (; v < v1; v += nsub) {
        const float x = src[((int64_t)n * p.V + v) * Cs];
        float g = p.ga[((int64_t)n * p.V + v) * Cin + c];
        if (p.act) g *= dsilu(fmaf(x, a, b));
        s1 += (double)g;
        s2 += (double)(g * ((x - mean) * rstd));
      }
    }
    if (nsub > 1) {
      red[0][threadIdx.x] = s1;
      red[1][threadIdx.x] = s2;
      __syncthreads();
      if (sub == 0 && c < Cin)
        for (int k = 1; k < nsub; ++k) {
          s1 += red[0][k * Cin + c0];
          s2 += red[1][k * Cin + c0];
        }
      __syncthreads();
    }
    if (sub == 0 && c < Cin) {
      double* d = p.part + (((int64_t)blockIdx.x * p.N + n) * Cin + c) * 2;
      d[0] = s1;
      d[1] = s2;
    }
  }
}
// stage 2 (one block per sample): channel sums, group terms, parameter and FiLM gradients.
//   sums[n][c] = (S1, S2);  grp[n][c] = (m1, m2) of the channel's group;  dgamma/dbeta accumulate over n (atomic-free: the
//   block of sample n handles ... one block handles ALL samples sequentially so the accumulation order is fixed)
__global__ __launch_bounds__(256) void gn_bwd_group_kernel(GnBwdParams p, int nblk) {
  __shared__ double sh[2 * 2048 + 64];  // [Cin][2] sums, then per group (A, B)
  __shared__ double R[512];
  const int Cin = p.C0 + p.C1;
  const int cpg = Cin / 32;
  double* S = sh;
  double* G = sh + 2 * Cin;
  for (int n = 0; n < p.N; ++n) {
    // threads = (channel, sub-block) as in stage 1; the sub-blocks' sums meet in LDS in a fixed order
    const int nsub = Cin >= 256 ? 1 : 256 / Cin;
    const int sub = threadIdx.x / Cin, c0 = threadIdx.x - sub * Cin;
    for (int cb = 0; cb < Cin; cb += 256) {
      const int c = cb + c0;
      double s1 = 0.0, s2 = 0.0;
      if (sub < nsub && c < Cin) {
        // ONE workgroup walks up to 512 slabs: eight 16-byte loads in flight, added in slab order (one load per iteration
        // made the 64^3 launches 130 us of pure latency)
        int b = sub;
        for (; b + 7 * nsub < nblk; b += 8 * nsub) {
          double2 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            v[u] = *reinterpret_cast<const double2*>(p.part + (((int64_t)(b + u * nsub) * p.N + n) * Cin + c) * 2);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            s1 += v[u].x;
            s2 += v[u].y;
          }
        }
        for (; b < nblk; b += nsub) {
          const double* d = p.part + (((int64_t)b * p.N + n) * Cin + c) * 2;
          s1 += d[0];
          s2 += d[1];
        }
      }
      if (nsub > 1) {
        R[2 * threadIdx.x] = s1;
        R[2 * threadIdx.x + 1] = s2;
        __syncthreads();
        if (sub == 0 && c < Cin)
          for (int k = 1; k < nsub; ++k) {
            s1 += R[2 * (k * Cin + c0)];
            s2 += R[2 * (k * Cin + c0) + 1];
          }
      }
      if (sub == 0 && c < Cin) {
        S[2 * c] = s1;
        S[2 * c + 1] = s2;
      }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int g = threadIdx.x;
      double A = 0.0, B = 0.0;
      for (int k = 0; k < cpg; ++k) {
        const int c = g * cpg + k;
        double gm = (double)p.gamma[c];
        if (p.film) gm *= 1.0 + (double)p.film[(int64_t)n * p.film_stride + c];
        A += gm * S[2 * c];
        B += gm * S[2 * c + 1];
      }
      const double cnt = (double)cpg * (double)p.V;
      G[2 * g] = A / cnt;
      G[2 * g + 1] = B / cnt;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Cin; c += blockDim.x) {
      const int g = c / cpg;
      p.grp[((int64_t)n * Cin + c) * 2] = (float)G[2 * g];
      p.grp[((int64_t)n * Cin + c) * 2 + 1] = (float)G[2 * g + 1];
      const double sc = p.film ? 1.0 + (double)p.film[(int64_t)n * p.film_stride + c] : 1.0;
      const float dg = (float)(sc * S[2 * c + 1]), db = (float)(sc * S[2 * c]);
      p.dgamma[c] = (n == 0 && !p.acc_params ? 0.f : p.dgamma[c]) + dg;
      p.dbeta[c] = (n == 0 && !p.acc_params ? 0.f : p.dbeta[c]) + db;
      if (p.dfilm) {  // d scale = sum gu (xhat gamma + beta), d shift = sum gu
        p.dfilm[(int64_t)n * p.film_stride + c] = (float)((double)p.gamma[c] * S[2 * c + 1] + (double)p.beta[c] * S[2 * c]);
        p.dfilm[(int64_t)n * p.film_stride + p.film_cout + c] = (float)S[2 * c];
      }
    }
    __syncthreads();
  }
}
// stage 3: gx[m][c] (+)= rstd (gu g' - m1 - xhat m2), written per source tensor
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnBwdParams p) {
  const int Cin = p.C0 + p.C1;
  const int64_t total = (int64_t)p.N * p.V * Cin;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cin);
    const int64_t m = i / Cin;
    const int n = (int)(m / p.V);
    const bool first = c < p.C0;
    const float* src = first ? p.x0 : p.x1;
    const int Cs = first ? p.C0 : p.C1, cs = first ? c : c - p.C0;
    const float x = src[m * Cs + cs];
    const float a = p.coef[((int64_t)n * Cin + c) * 2], b = p.coef[((int64_t)n * Cin + c) * 2 + 1];
    const float mean = p.mom[((int64_t)n * Cin + c) * 2], rstd = p.mom[((int64_t)n * Cin + c) * 2 + 1];
    float g = p.ga[i];
    if (p.act) g *= dsilu(fmaf(x, a, b));
    float gm = p.gamma[c];
    if (p.film) gm *= 1.0f + p.film[(int64_t)n * p.film_stride + c];
    const float xh = (x - mean) * rstd;
    const float gx = rstd * (g * gm - p.grp[((int64_t)n * Cin + c) * 2] - xh * p.grp[((int64_t)n * Cin + c) * 2 + 1]);
    float* dst = first ? p.gx0 : p.gx1;
    const int accf = first ? p.acc0 : p.acc1;
    dst[m * Cs + cs] = (accf ? dst[m * Cs + cs] : 0.f) + gx;
  }
}

// dst (+)= src
__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n, int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = (accumulate ? dst[i] : 0.f) + src[i];
}
// splits a gradient w.r.t. the virtual concat [x0 | x1] (rows of C0 + C1) into the two tensors' gradients
__global__ __launch_bounds__(256) void split_cat_kernel(const float* __restrict__ g, float* __restrict__ g0, float* __restrict__ g1,
                                                       int64_t M, int C0, int C1, int acc0, int acc1) {
  const int C = C0 + C1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * C; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t m = i / C;
    if (c < C0)
      g0[m * C0 + c] = (acc0 ? g0[m * C0 + c] : 0.f) + g[i];
    else
      g1[m * C1 + c - C0] = (acc1 ? g1[m * C1 + c - C0] : 0.f) + g[i];
  }
}

// backward of the nearest x2 upsampling: g_in[n][z][y][x][c] (+)= sum of the 8 children of g_up (R_up = 2 R)
__global__ __launch_bounds__(256) void sumpool2_kernel(const float* __restrict__ gup, float* __restrict__ gin, int N, int R, int C,
                                                      int accumulate) {
  const int64_t total = (int64_t)N * R * R * R * C;
  const int R2 = 2 * R;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t v = i / C;
    const int x = (int)(v % R);
    v /= R;
    const int y = (int)(v % R);
    v /= R;
    const int z = (int)(v % R);
    const int n = (int)(v / R);
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const int zz = 2 * z + (d >> 2), yy = 2 * y + ((d >> 1) & 1), xx = 2 * x + (d & 1);
      s += gup[((((int64_t)n * R2 + zz) * R2 + yy) * R2 + xx) * C + c];
    }
    gin[i] = (accumulate ? gin[i] : 0.f) + s;
  }
}

// transposed 3x3x3 stride-2 pad-1 convolution: gx[n][z][y][x][ci] (+)= sum_{tap, co} gy[n][o][co] W[co][ci][tap], o = (in + 1 - k) / 2
// where that is an integer inside the output.  wt is the weight as [tap][co][ci].  One thread per (input voxel, ci).
__global__ __launch_bounds__(256) void conv_dgrad_s2_kernel(const float* __restrict__ gy, const float* __restrict__ wt,
                                                           float* __restrict__ gx, int N, int RI, int RO, int Ci, int Co,
                                                           int accumulate) {
  const int64_t total = (int64_t)N * RI * RI * RI * Ci;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ci);
    int64_t v = i / Ci;
    const int x = (int)(v % RI);
    v /= RI;
    const int y = (int)(v % RI);
    v /= RI;
    const int z = (int)(v % RI);
    const int n = (int)(v / RI);
    float s = 0.f;
    for (int kd = 0; kd < 3; ++kd) {
      const int tz = z + 1 - kd;
      if (tz < 0 || (tz & 1) || (tz >> 1) >= RO) continue;
      for (int kh = 0; kh < 3; ++kh) {
        const int ty = y + 1 - kh;
        if (ty < 0 || (ty & 1) || (ty >> 1) >= RO) continue;
        for (int kw = 0; kw < 3; ++kw) {
          const int tx = x + 1 - kw;
          if (tx < 0 || (tx & 1) || (tx >> 1) >= RO) continue;
          const float* g = gy + ((((int64_t)n * RO + (tz >> 1)) * RO + (ty >> 1)) * RO + (tx >> 1)) * Co;
          const float* w = wt + ((int64_t)((kd * 3 + kh) * 3 + kw) * Co) * Ci + ci;
          for (int co = 0; co < Co; ++co) s = fmaf(g[co], w[(int64_t)co * Ci], s);
        }
      }
    }
    gx[i] = (accumulate ? gx[i] : 0.f) + s;
  }
}
// zero insertion in front of a transposed stride-2 convolution: out[n][z][y][x][c] (R = 2 Rs cubed) = gy[n][z/2][y/2][x/2][c]
// where z, y and x are all even, 0 elsewhere.  The transposed 3x3x3 stride-2 pad-1 convolution of gy is then the STRIDE-1
// transposed convolution of `out` - eight times the multiply-adds, on the Winograd kernels instead of scalar FMAs
// (Downsample.op at 64^3 <- 32^3, 64 channels: 1.35 ms with conv_dgrad_s2_kernel, 0.24 ms this way).  16-byte items.
__global__ __launch_bounds__(256) void zero_insert2_kernel(const float4* __restrict__ gy, float4* __restrict__ out, int N, int Rs,
                                                          int cq) {
  const int R = 2 * Rs;
  const int64_t total = (int64_t)N * R * R * R * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq);
    int64_t v = i / cq;
    const int x = (int)(v % R);
    v /= R;
    const int y = (int)(v % R);
    v /= R;
    const int z = (int)(v % R);
    const int n = (int)(v / R);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (((x | y | z) & 1) == 0) o = gy[((((int64_t)n * Rs + (z >> 1)) * Rs + (y >> 1)) * Rs + (x >> 1)) * cq + c];
    out[i] = o;
  }
}
// OIDHW [co][ci][t] -> [t][co][ci]
__global__ __launch_bounds__(256) void weight_tco_ci_kernel(const float* __restrict__ in, float* __restrict__ out, int Co, int Ci, int T) {
  const int64_t total = (int64_t)Co * Ci * T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int ci = (int)((i / T) % Ci);
    const int co = (int)(i / ((int64_t)T * Ci));
    out[((int64_t)t * Co + co) * Ci + ci] = holo_ld_sys(in + i);
  }
}

// ---- attention ---------------------------------------------------------------------------------------------------------
// rows of the softmax backward, in place on dP: dS = P (dP - sum_j dP P)
__global__ __launch_bounds__(256) void attn_ds_kernel(const float* __restrict__ P, float* __restrict__ dP, int cols) {
  __shared__ float red[256];
  const int64_t row = blockIdx.x;
  const float* pr = P + row * cols;
  float* dr = dP + row * cols;
  float s = 0.f;
  for (int j = threadIdx.x; j < cols; j += 256) s = fmaf(pr[j], dr[j], s);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float dot = red[0];
  for (int j = threadIdx.x; j < cols; j += 256) dr[j] = pr[j] * (dr[j] - dot);
}
// batched square transpose out[b][j][i] = in[b][i][j] (T x T), 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int T) {
  __shared__ float tile[32][33];
  const int64_t base = (int64_t)blockIdx.z * T * T;
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8)
    if (by + r < T && bx + tx < T) tile[r][tx] = in[base + (int64_t)(by + r) * T + bx + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (bx + r < T && by + tx < T) out[base + (int64_t)(bx + r) * T + by + tx] = tile[tx][r];
}

// ---- embedding path ----------------------------------------------------------------------------------------------------
// all emb_layers at once (the forward concatenates them): dfilm [N][rows] is the gradient of Linear(SiLU(emb)):
//   d emb_w[r][k] = sum_n dfilm[n][r] embs[n][k];  d emb_b[r] = sum_n dfilm[n][r];  g_embs[n][k] = sum_r dfilm[n][r] W[r][k]
__global__ __launch_bounds__(256) void film_bwd_w_kernel(const float* __restrict__ dfilm, const float* __restrict__ embs,
                                                        float* __restrict__ dw, float* __restrict__ db, int N, int rows, int K) {
  const int r = blockIdx.x;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s = fmaf(dfilm[(int64_t)n * rows + r], embs[(int64_t)n * K + k], s);
    dw[(int64_t)r * K + k] = s;
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dfilm[(int64_t)n * rows + r];
    db[r] = s;
  }
}
// one block per (4 columns k, sample): 64 row lanes per column stride the rows, summed through LDS in a fixed order
__global__ __launch_bounds__(256) void film_bwd_x_kernel(const float* __restrict__ dfilm, const float* __restrict__ w,
                                                        float* __restrict__ gembs, int rows, int K) {
  __shared__ double red[256];
  const int n = blockIdx.y;
  const int kl = threadIdx.x & 3, sub = threadIdx.x >> 2;
  const int k = blockIdx.x * 4 + kl;
  double s = 0.0;
  if (k < K) {
    int r = sub;
    for (; r + 3 * 64 < rows; r += 4 * 64) {
      float d4[4], w4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        d4[u] = dfilm[(int64_t)n * rows + r + u * 64];
        w4[u] = w[(int64_t)(r + u * 64) * K + k];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) s += (double)d4[u] * (double)w4[u];
    }
    for (; r < rows; r += 64) s += (double)dfilm[(int64_t)n * rows + r] * (double)w[(int64_t)r * K + k];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (sub == 0 && k < K) {
    for (int j = 1; j < 64; ++j) s += red[j * 4 + kl];
    gembs[(int64_t)n * K + k] = (float)s;
  }
}
// time_embed (unet.py:645-650) backward, one block: recomputes the forward of every sample, accumulates over the samples
__global__ __launch_bounds__(256) void time_embed_bwd_kernel(const int64_t* __restrict__ t, int N, int mc, int ted,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, const float* __restrict__ b2,
                                                            const float* __restrict__ gembs, float* __restrict__ dw1,
                                                            float* __restrict__ db1, float* __restrict__ dw2,
                                                            float* __restrict__ db2) {
  __shared__ float te[256], pre1[1024], h1[1024], gemb[1024], gh[1024];
  const int tid = threadIdx.x;
  for (int n = 0; n < N; ++n) {
    const float tv = (float)holo_ld_sys(t + n);
    const int half = mc / 2;
    for (int i = tid; i < mc; i += 256) {
      float v = 0.f;
      if (i < 2 * half) {
        const int k = i < half ? i : i - half;
        const float freq = expf((-9.210340371976184f * (float)k) / (float)half);
        v = i < half ? cosf(tv * freq) : sinf(tv * freq);
      }
      te[i] = v;
    }
    __syncthreads();
    for (int j = tid; j < ted; j += 256) {
      float acc = 0.f;
      for (int k = 0; k < mc; ++k) acc = fmaf(w1[(int64_t)j * mc + k], te[k], acc);
      acc += b1[j];
      pre1[j] = acc;
      h1[j] = acc / (1.0f + expf(-acc));
    }
    __syncthreads();
    for (int j = tid; j < ted; j += 256) {
      float acc = 0.f;
      for (int k = 0; k < ted; ++k) acc = fmaf(w2[(int64_t)j * ted + k], h1[k], acc);
      acc += b2[j];  // emb; embs = silu(emb)
      gemb[j] = gembs[(int64_t)n * ted + j] * dsilu(acc);
    }
    __syncthreads();
    for (int j = tid; j < ted; j += 256) {
      db2[j] = (n ? db2[j] : 0.f) + gemb[j];
      for (int k = 0; k < ted; ++k) dw2[(int64_t)j * ted + k] = (n ? dw2[(int64_t)j * ted + k] : 0.f) + gemb[j] * h1[k];
    }
    for (int k = tid; k < ted; k += 256) {
      float s = 0.f;
      for (int j = 0; j < ted; ++j) s = fmaf(gemb[j], w2[(int64_t)j * ted + k], s);
      gh[k] = s * dsilu(pre1[k]);
    }
    __syncthreads();
    for (int j = tid; j < ted; j += 256) {
      db1[j] = (n ? db1[j] : 0.f) + gh[j];
      for (int k = 0; k < mc; ++k) dw1[(int64_t)j * mc + k] = (n ? dw1[(int64_t)j * mc + k] : 0.f) + gh[j] * te[k];
    }
    __syncthreads();
  }
}

// NCDHW <-> channels-last for the gradient at the boundary
__global__ __launch_bounds__(256) void scale_fill_kernel(float* __restrict__ dst, float v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = v;
}

static inline unsigned blocks_for(int64_t n, int cap = 16384) {
  int64_t b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

int flip_transpose_weight_launch(const float* in, float* out, int Co, int Ci, int T, void* stream) {
  HOLO_LAUNCH(flip_transpose_weight_kernel, dim3(blocks_for((int64_t)Co * Ci * T)), dim3(256), stream, in, out, Co, Ci, T);
  return 0;
}
int weight_tco_ci_launch(const float* in, float* out, int Co, int Ci, int T, void* stream) {
  HOLO_LAUNCH(weight_tco_ci_kernel, dim3(blocks_for((int64_t)Co * Ci * T)), dim3(256), stream, in, out, Co, Ci, T);
  return 0;
}

// the row-staged kernel serves the stride-1 3x3x3 convolutions of rows up to 64 voxels (HOLO_WGRAD_TILES=1: tile kernel)
static bool wgrad_rows_ok(const WgradParams& p) {
#ifndef HOLO_EMU
  static const char* e = getenv("HOLO_WGRAD_TILES");
  if (e && atoi(e) > 0) return false;
#endif
  const int Cin = p.C0 + p.C1;
  return p.ksz == 3 && p.stride == 1 && p.pad == 1 && p.OW <= WR_MAXW && (p.OW & 1) == 0 && p.ID == p.OD && p.IH == p.OH &&
         p.IW == p.OW && (Cin & 3) == 0 && (p.Cout & 3) == 0 && (p.C0 & 3) == 0 && (p.C1 & 3) == 0 && (!p.src1 || (p.C0 & 31) == 0);
}
int wgrad_splits(const WgradParams& p, int num_cus) {
  const int Cin = p.C0 + p.C1;
  const int64_t nrows = (int64_t)p.N * p.OD * p.OH;
  const int ncu = num_cus > 0 ? num_cus : 256;
  if (wgrad_rows_ok(p)) {  // workgroups per CU by LDS / registers (1 or 2); slabs of at least 8 rows keep the row ring useful
    const int64_t pairs = (int64_t)((p.Cout + 31) / 32) * ((Cin + 31) / 32);
    int64_t s = ((p.OW > 32 ? 1 : 2) * (int64_t)ncu + pairs - 1) / pairs;
    if (s > nrows / 8) s = nrows / 8;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return (int)s;
  }
  const int64_t tiles = (int64_t)((p.Cout + 31) / 32) * ((Cin + 31) / 32) * p.ntaps;
  int64_t s = (8LL * ncu + tiles - 1) / tiles;  // ~8 workgroups per CU in flight over the launch
  if (s > nrows / 4) s = nrows / 4;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}
size_t wgrad_partial_bytes(const WgradParams& p, int num_cus) {
  return (size_t)wgrad_splits(p, num_cus) * p.Cout * (p.C0 + p.C1) * p.ntaps * sizeof(float);
}
// dw: OIDHW [Cout][C0 + C1][ntaps]
int conv_wgrad_launch(const WgradParams& p, float* dw, int accumulate, int num_cus, void* stream) {
  const int Cin = p.C0 + p.C1;
  if (p.src1 && (p.C0 & 31)) {
    set_error("conv_wgrad: the first source of a virtual concat must have a multiple of 32 channels");
    return -1;
  }
  const int splits = wgrad_splits(p, num_cus);
  if (wgrad_rows_ok(p)) {
    dim3 grid((unsigned)(((p.Cout + 31) / 32) * ((Cin + 31) / 32)), (unsigned)splits);
    if (p.OW > 32) {
      HOLO_LAUNCH((conv_wgrad_rows_kernel<64, 8>), grid, dim3(512), stream, p);
    } else if (p.OW > 16) {
      HOLO_LAUNCH((conv_wgrad_rows_kernel<32, 4>), grid, dim3(256), stream, p);
    } else {
      HOLO_LAUNCH((conv_wgrad_rows_kernel<16, 4>), grid, dim3(256), stream, p);
    }
  } else {
    dim3 grid((unsigned)(((p.Cout + 31) / 32) * ((Cin + 31) / 32) * p.ntaps), (unsigned)splits);
    HOLO_LAUNCH(conv_wgrad_kernel, grid, dim3(256), stream, p);
  }
  const int64_t n = (int64_t)p.Cout * Cin * p.ntaps;
  if (wgrad_rows_ok(p)) {
    const int64_t tiles = ((int64_t)p.Cout * Cin + 63) / 64;
    const char* rt = getenv("HOLO_WGRAD_REDUCE_TILE_MIN");  // development / test knob: tiles from which the tile form runs
    const int64_t tmin = rt ? atoll(rt) : 1024;
    if (tiles >= tmin && p.ntaps <= 28) {
      HOLO_LAUNCH(wgrad_reduce_tile_kernel, dim3((unsigned)(tiles < 8192 ? tiles : 8192)), dim3(256), stream, p.partial, dw, p.Cout,
                  Cin, p.ntaps, splits, accumulate);
    } else {
      HOLO_LAUNCH(wgrad_reduce_t_kernel, dim3(blocks_for(n)), dim3(256), stream, p.partial, dw, p.Cout, Cin, p.ntaps, splits, accumulate);
    }
  } else {
    HOLO_LAUNCH(wgrad_reduce_kernel, dim3(blocks_for(n)), dim3(256), stream, p.partial, dw, n, splits, accumulate);
  }
  return 0;
}

int partial_reduce_launch(const float* partial, float* dw, int64_t n, int splits, int accumulate, void* stream) {
  HOLO_LAUNCH(wgrad_reduce_kernel, dim3(blocks_for(n)), dim3(256), stream, partial, dw, n, splits, accumulate);
  return 0;
}

size_t colsum_scratch_bytes(int C) { return (size_t)256 * C * sizeof(double); }
int colsum_launch(const float* g, int64_t M, int C, double* scratch, float* out, int accumulate, void* stream) {
  int nb = (int)(M < 256 ? M : 256);
  if (nb < 1) nb = 1;
  HOLO_LAUNCH(colsum_part_kernel, dim3((unsigned)nb), dim3(256), stream, g, scratch, M, C);
  HOLO_LAUNCH(colsum_final_kernel, dim3((unsigned)((C + 31) / 32)), dim3(256), stream, scratch, out, nb, C, accumulate);
  return 0;
}

int gn_bwd_blocks(int64_t V) {
  int64_t b = V / 64;
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}
size_t gn_bwd_scratch_bytes(const GnBwdParams& p) {
  return (size_t)gn_bwd_blocks(p.V) * p.N * (p.C0 + p.C1) * 2 * sizeof(double);
}
int gn_bwd_launch(const GnBwdParams& p, void* stream) {
  const int Cin = p.C0 + p.C1;
  if (Cin % 32 || Cin > 2048) {
    set_error("gn_bwd: channel count %d is not a multiple of the 32 groups (or > 2048)", Cin);
    return -1;
  }
  const int nblk = gn_bwd_blocks(p.V);
  HOLO_LAUNCH(gn_bwd_part_kernel, dim3((unsigned)nblk, (unsigned)p.N), dim3(256), stream, p);
  HOLO_LAUNCH(gn_bwd_group_kernel, dim3(1), dim3(256), stream, p, nblk);
  HOLO_LAUNCH(gn_bwd_apply_kernel, dim3(blocks_for((int64_t)p.N * p.V * Cin)), dim3(256), stream, p);
  return 0;
}

int add_launch(float* dst, const float* src, int64_t n, int accumulate, void* stream) {
  HOLO_LAUNCH(add_kernel, dim3(blocks_for(n)), dim3(256), stream, dst, src, n, accumulate);
  return 0;
}
int split_cat_launch(const float* g, float* g0, float* g1, int64_t M, int C0, int C1, int acc0, int acc1, void* stream) {
  HOLO_LAUNCH(split_cat_kernel, dim3(blocks_for(M * (C0 + C1))), dim3(256), stream, g, g0, g1, M, C0, C1, acc0, acc1);
  return 0;
}
int sumpool2_launch(const float* gup, float* gin, int N, int R, int C, int accumulate, void* stream) {
  HOLO_LAUNCH(sumpool2_kernel, dim3(blocks_for((int64_t)N * R * R * R * C)), dim3(256), stream, gup, gin, N, R, C, accumulate);
  return 0;
}
int conv_dgrad_s2_launch(const float* gy, const float* wt, float* gx, int N, int RI, int RO, int Ci, int Co, int accumulate,
                         void* stream) {
  HOLO_LAUNCH(conv_dgrad_s2_kernel, dim3(blocks_for((int64_t)N * RI * RI * RI * Ci, 65536)), dim3(256), stream, gy, wt, gx, N, RI,
              RO, Ci, Co, accumulate);
  return 0;
}
int zero_insert2_launch(const float* gy, float* out, int N, int Rs, int C, void* stream) {
  if (C & 3) {
    set_error("zero_insert2: C=%d is not a multiple of 4", C);
    return -1;
  }
  HOLO_LAUNCH(zero_insert2_kernel, dim3(blocks_for((int64_t)N * 8 * Rs * Rs * Rs * (C >> 2), 65536)), dim3(256), stream,
              reinterpret_cast<const float4*>(gy), reinterpret_cast<float4*>(out), N, Rs, C >> 2);
  return 0;
}
int attn_ds_launch(const float* P, float* dP, int64_t rows, int cols, void* stream) {
  HOLO_LAUNCH(attn_ds_kernel, dim3((unsigned)rows), dim3(256), stream, P, dP, cols);
  return 0;
}
int transpose_launch(const float* in, float* out, int batch, int T, void* stream) {
  HOLO_LAUNCH(transpose_kernel, dim3((unsigned)((T + 31) / 32), (unsigned)((T + 31) / 32), (unsigned)batch), dim3(256), stream, in,
              out, T);
  return 0;
}
int film_bwd_launch(const float* dfilm, const float* embs, const float* w, float* dw, float* db, float* gembs, int N, int rows,
                    int K, void* stream) {
  HOLO_LAUNCH(film_bwd_w_kernel, dim3((unsigned)rows), dim3(256), stream, dfilm, embs, dw, db, N, rows, K);
  HOLO_LAUNCH(film_bwd_x_kernel, dim3((unsigned)((K + 3) / 4), (unsigned)N), dim3(256), stream, dfilm, w, gembs, rows, K);
  return 0;
}
int time_embed_bwd_launch(const int64_t* t, int N, int mc, int ted, const float* w1, const float* b1, const float* w2,
                          const float* b2, const float* gembs, float* dw1, float* db1, float* dw2, float* db2, void* stream) {
  if (mc > 256 || ted > 1024) {
    set_error("time_embed_bwd: model_channels=%d too large", mc);
    return -1;
  }
  HOLO_LAUNCH(time_embed_bwd_kernel, dim3(1), dim3(256), stream, t, N, mc, ted, w1, b1, w2, b2, gembs, dw1, db1, dw2, db2);
  return 0;
}
int fill_launch(float* dst, float v, int64_t n, void* stream) {
  HOLO_LAUNCH(scale_fill_kernel, dim3(blocks_for(n)), dim3(256), stream, dst, v, n);
  return 0;
}

}  // namespace holo
