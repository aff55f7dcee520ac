// kernels_viewpool_bwd.hip — backward of the view-pooling entry (kernels_viewpool.hip: view_pool_kernel) for a gradient on
// its output: what autograd leaves in the reference behind `tanh(pooled_feature_mapper(view_pooler(...)))`
// (holo_diffusion/holo_diffusion_model.py:358-373 with the released AngleWeightedReductionFeatureAggregator,
// configs/apple.yaml:183-196) when the encoder side is trained: gradients of the per-view feature maps (handed on to the
// image feature extractor, which is outside this library), of pooled_feature_mapper.weight and .bias.
//
// Per voxel p (views v, channels c; w_v = the angular weights, geometry only: no gradient):
//   D = max(sum_v w_v, 1e-2), S0 = sum_v w_v
//   mu_c = sum_v w_v x_vc / D,  var_c = sum_v w_v (x_vc - mu_c)^2 / D,  std_c = sqrt(max(var_c, 1e-4))
//   z = M [mu | std] + b,  out = tanh(z)
// Backward of a gradient g on out:
//   dz = g (1 - out^2);   dM += dz agg^T;   db += dz;   dagg = M^T dz = [dmu | dstd]
//   dvar_c = dstd_c / (2 std_c) where var_c > 1e-4, else 0            (clamp passes no gradient below its bound)
//   dmu'_c = dmu_c - 2 dvar_c mu_c (D - S0) / D                       (the mean inside the variance; 0 when S0 >= 1e-2)
//   dx_vc  = (w_v / D) (dmu'_c + 2 dvar_c (x_vc - mu_c))
//   d feature map: dx_vc scattered through the four bilinear tap weights (atomic adds, as grid_sample's backward)
// One PERSISTENT kernel, the forward's thread layout (16 voxels x 16 lanes, a lane owns channel quads q, q + 16, ...):
// pass 1 over the views rebuilds agg (nothing of the forward is kept), the mapper's Linear and its transpose run from the
// 16 x A tile in LDS, pass 2 gathers the samples again and scatters.  dM is accumulated in REGISTERS across all of a
// workgroup's voxel groups (thread t owns rows a = t, t + 256 of M^T: 2 x F values) and written once as a per-workgroup
// partial; viewpool_partial_reduce_kernel sums the partials in a fixed order (deterministic; the feature-map gradients are
// atomics and are not).
// (Round 5, first attempt at the 1.1 G atomics: bricks of 64 voxels, the four 16-channel maps accumulated per view in LDS
// tiles over the brick's bounding box, one global atomic per touched (pixel, channel) - 7x fewer global atomics, 16.3 ms
// against 15.5 ms: the LDS atomics met the same same-address contention.  Removed.  Second form: view_pool_bwd2_kernel
// below - occupancy first, then a segmented sum along rows of voxels in front of the atomics, then the
// atomics issued as whole pixels: 3.9 ms.  The kernel in this
// first half stays for calls beyond the second form's limits.)
#include <stdint.h>

#include <stdio.h>
#include <stdlib.h>

#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {
namespace {

constexpr int VB_F = 32;  // output features held per thread for dM (pooled_feature_mapper is feature_size = 32 wide)

struct Tap {
  int o00, o01, o10, o11;  // element offsets of the four taps inside one view's (H, W, Cp) map, channel 0
  float w00, w01, w10, w11;
  int key;                 // the bilinear cell (floor of the sample position, both axes), > 0: equal keys = the same four taps
};

// ndc_grid_sample's tap geometry (kernels_viewpool.hip: view_pool_kernel)
__device__ __forceinline__ Tap tap_of(const ViewPoolParams::Feat& f, float ndcx, float ndcy) {
  float gx = -ndcx, gy = -ndcy;
  if (f.W >= f.H) gx /= (float)f.W / (float)f.H; else gy /= (float)f.H / (float)f.W;
  const float ix = ((gx + 1.f) * (float)f.W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)f.H - 1.f) * 0.5f;
  const float fx0 = floorf(ix), fy0 = floorf(iy);
  const float tx = ix - fx0, ty = iy - fy0;
  const float fW = (float)(f.W - 1), fH = (float)(f.H - 1);
  const float wx0 = (fx0 >= 0.f && fx0 <= fW) ? 1.f - tx : 0.f, wx1 = (fx0 >= -1.f && fx0 <= fW - 1.f) ? tx : 0.f;
  const float wy0 = (fy0 >= 0.f && fy0 <= fH) ? 1.f - ty : 0.f, wy1 = (fy0 >= -1.f && fy0 <= fH - 1.f) ? ty : 0.f;
  const int x0 = (int)fminf(fmaxf(fx0, 0.f), fW), x1 = (int)fminf(fmaxf(fx0 + 1.f, 0.f), fW);
  const int y0 = (int)fminf(fmaxf(fy0, 0.f), fH), y1 = (int)fminf(fmaxf(fy0 + 1.f, 0.f), fH);
  Tap t;
  t.o00 = (y0 * f.W + x0) * f.Cp, t.o01 = (y0 * f.W + x1) * f.Cp, t.o10 = (y1 * f.W + x0) * f.Cp, t.o11 = (y1 * f.W + x1) * f.Cp;
  t.w00 = wx0 * wy0, t.w01 = wx1 * wy0, t.w10 = wx0 * wy1, t.w11 = wx1 * wy1;
  // (positions far outside the map share the border keys: all their weights are zero)
  t.key = (((int)fminf(fmaxf(fy0, -2.f), fH + 1.f) + 3) << 16) | ((int)fminf(fmaxf(fx0, -2.f), fW + 1.f) + 3);
  return t;
}

__device__ __forceinline__ void sample4(const float* base, const Tap& t, float (&s)[4]) {
  const float4 t00 = *reinterpret_cast<const float4*>(base + t.o00), t01 = *reinterpret_cast<const float4*>(base + t.o01);
  const float4 t10 = *reinterpret_cast<const float4*>(base + t.o10), t11 = *reinterpret_cast<const float4*>(base + t.o11);
  s[0] = t00.x * t.w00 + t01.x * t.w01 + t10.x * t.w10 + t11.x * t.w11;
  s[1] = t00.y * t.w00 + t01.y * t.w01 + t10.y * t.w10 + t11.y * t.w11;
  s[2] = t00.z * t.w00 + t01.z * t.w01 + t10.z * t.w10 + t11.z * t.w11;
  s[3] = t00.w * t.w00 + t01.w * t.w01 + t10.w * t.w10 + t11.w * t.w11;
}

// Where a scatter kernel's sums go.  MODE 0 (default): hardware fp32 atomics on the gradient map.  The deterministic mode
// (holo_ctx_set_deterministic) runs a launch twice: MODE 1 only measures - the bits of the largest |addend| of every map, LDS
// word first, one global atomicMax per workgroup and map at the end - and MODE 2 adds the same values as 64-bit fixed-point
// integers whose binary point comes from that maximum (holo_common.h: the sums no longer depend on the order of the atomics).
struct ScatterDst {
  float* f32;
  long long* fix;
  int shift;
  uint32_t* lmax;
};
template <int MODE, typename P>
__device__ __forceinline__ ScatterDst scatter_dst(const P& b, int k, uint32_t* s_max) {
  ScatterDst d;
  d.f32 = b.gfeat[k];
  d.fix = MODE == 2 ? b.gfix[k] : nullptr;
  d.shift = 0;
  if (MODE == 2) {
    const uint32_t mb = b.fix_max[k];
    d.shift = holo_fix_shift(mb);
    if (mb == 0u || mb >= 0x7f800000u) d.fix = nullptr;  // nothing to add / not finite (fix_flush_kernel writes NaN)
  }
  d.lmax = s_max + k;
  return d;
}
template <int MODE>
__device__ __forceinline__ void scatter_add(const ScatterDst& d, int64_t off, float val, uint32_t& vmax) {
  if (MODE == 0) {
    HOLO_ATOMIC_ADD_F32(d.f32 + off, val);
  } else if (MODE == 1) {
    const uint32_t bits = __float_as_uint(val) & 0x7fffffffu;
    vmax = bits > vmax ? bits : vmax;
  } else if (d.fix) {
    holo_fix_add(d.fix + off, val, d.shift);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void view_pool_bwd_kernel(ViewPoolBwdParams b) {
  const ViewPoolParams& p = b.fwd;
  __shared__ uint32_t s_max[ViewPoolParams::MAX_FEATS];
  if (MODE == 1 && threadIdx.x < ViewPoolParams::MAX_FEATS) s_max[threadIdx.x] = 0u;
  __shared__ float s_agg[16 * ViewPoolParams::MAX_AGG];   // [voxel][aggregated feature]
  __shared__ float s_dagg[16 * ViewPoolParams::MAX_AGG];  // [voxel][d loss / d aggregated feature]
  __shared__ float s_dz[16 * VB_F];
  const int tid = threadIdx.x;
  const int vl = tid >> 4, ql = tid & 15;
  const int R = p.R;
  const int64_t nvox = (int64_t)R * R * R;
  const int64_t ngroups = (nvox + 15) / 16;
  const float step = 2.0f / (float)(R - 1);
  auto lin = [&](int i) { return (i < R / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(R - 1 - i)) * p.half_extent; };
  const float std_floor = sqrtf(1e-4f);

  float dM[2][VB_F];  // rows a = tid, tid + 256 of d M^T
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int o = 0; o < VB_F; ++o) dM[h][o] = 0.f;
  float db = 0.f;  // thread tid < F: d bias[tid]

  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t v = grp * 16 + vl;
    const bool vok = v < nvox;
    const int64_t vc = vok ? v : nvox - 1;
    const int x = (int)(vc % R), y = (int)((vc / R) % R), z = (int)(vc / ((int64_t)R * R));
    const float px = lin(x), py = lin(y), pz = lin(z);
    float ndcx[ViewPoolParams::MAX_VIEWS], ndcy[ViewPoolParams::MAX_VIEWS], wv[ViewPoolParams::MAX_VIEWS];
    float d0x = 0.f, d0y = 0.f, d0z = 0.f, S0 = 0.f;
#pragma unroll 1
    for (int vi = 0; vi < p.n_views; ++vi) {
      const ViewPoolParams::Cam& c = p.cams[vi];
      const float cx = px * c.Rm[0] + py * c.Rm[3] + pz * c.Rm[6] + c.T[0];
      const float cy = px * c.Rm[1] + py * c.Rm[4] + pz * c.Rm[7] + c.T[1];
      float cz = px * c.Rm[2] + py * c.Rm[5] + pz * c.Rm[8] + c.T[2];
      if (fabsf(cz) < p.proj_eps) cz = cz < 0.f ? -p.proj_eps : p.proj_eps;
      ndcx[vi] = c.focal[0] * cx / cz + c.pp[0];
      ndcy[vi] = c.focal[1] * cy / cz + c.pp[1];
      float dx = px - c.centre[0], dy = py - c.centre[1], dz = pz - c.centre[2];
      const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
      dx /= nrm;
      dy /= nrm;
      dz /= nrm;
      if (vi == 0) d0x = dx, d0y = dy, d0z = dz;
      float a = 0.5f * ((dx * d0x + dy * d0y + dz * d0z) + 1.0f);
      if (p.gamma != 1.0f) a = powf(a, p.gamma);
      wv[vi] = fmaxf(a, p.min_weight);
      S0 += wv[vi];
    }
    const float D = fmaxf(S0, 1e-2f);

    // ---- pass 1: the forward's aggregation, [AVG | STD] per key into the LDS tile
    for (int q = ql; q < p.n_quads; q += 16) {
      int k = 0;
      while (k + 1 < p.n_feats && q >= p.feat[k + 1].quad0) ++k;
      const ViewPoolParams::Feat& f = p.feat[k];
      const int cq = q - f.quad0;
      float S1[4] = {0.f, 0.f, 0.f, 0.f}, S2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int vi = 0; vi < p.n_views; ++vi) {
        const Tap t = tap_of(f, ndcx[vi], ndcy[vi]);
        float s[4];
        sample4(f.data + ((int64_t)vi * f.H * f.W) * f.Cp + cq * 4, t, s);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          S1[e] = fmaf(wv[vi], s[e], S1[e]);
          S2[e] = fmaf(wv[vi] * s[e], s[e], S2[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = cq * 4 + e;
        if (c < f.C) {
          const float mu = S1[e] / D;
          const float var = (S2[e] - 2.f * mu * S1[e] + mu * mu * S0) / D;
          s_agg[vl * ViewPoolParams::MAX_AGG + f.out0 + c] = mu;
          s_agg[vl * ViewPoolParams::MAX_AGG + f.out0 + f.C + c] = sqrtf(fmaxf(var, 1e-4f));
        }
      }
    }
    __syncthreads();
    // ---- mapper forward, tanh, dz = g (1 - out^2); voxels beyond the grid contribute nothing
    for (int o = ql; o < p.F; o += 16) {
      float acc = p.bias ? p.bias[o] : 0.f;
      for (int a = 0; a < p.A; ++a) acc = fmaf(s_agg[vl * ViewPoolParams::MAX_AGG + a], p.wt[(int64_t)a * p.F + o], acc);
      const float out = tanhf(acc);
      const float g = vok ? b.gout[(int64_t)o * nvox + v] : 0.f;
      s_dz[vl * VB_F + o] = g * (1.f - out * out);
    }
    __syncthreads();
    // ---- dagg = M^T dz (the lane's aggregated features a = ql, ql + 16, ...)
    for (int a = ql; a < p.A; a += 16) {
      float acc = 0.f;
      for (int o = 0; o < p.F; ++o) acc = fmaf(p.wt[(int64_t)a * p.F + o], s_dz[vl * VB_F + o], acc);
      s_dagg[vl * ViewPoolParams::MAX_AGG + a] = acc;
    }
    // ---- d M^T rows tid, tid + 256 and d bias over the 16 voxels of the group
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int a = tid + 256 * h;
      if (a < p.A) {
#pragma unroll 4
        for (int pv = 0; pv < 16; ++pv) {
          const float ag = s_agg[pv * ViewPoolParams::MAX_AGG + a];
#pragma unroll
          for (int o = 0; o < VB_F; ++o) dM[h][o] = fmaf(ag, o < p.F ? s_dz[pv * VB_F + o] : 0.f, dM[h][o]);
        }
      }
    }
    if (tid < p.F) {
#pragma unroll 4
      for (int pv = 0; pv < 16; ++pv) db += s_dz[pv * VB_F + tid];
    }
    __syncthreads();
    // ---- pass 2: the samples again, dx per view, scattered through the bilinear weights
    if (b.want_feats) {
      for (int q = ql; q < p.n_quads; q += 16) {
        int k = 0;
        while (k + 1 < p.n_feats && q >= p.feat[k + 1].quad0) ++k;
        const ViewPoolParams::Feat& f = p.feat[k];
        if (!b.gfeat[k]) continue;
        const ScatterDst dst = scatter_dst<MODE>(b, k, s_max);
        uint32_t vmax = 0u;
        const int cq = q - f.quad0;
        float mu[4], dmu[4], dvar2[4];  // dvar2 = 2 dvar
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = cq * 4 + e;
          const bool cok = c < f.C;
          const int ia = vl * ViewPoolParams::MAX_AGG + f.out0 + (cok ? c : 0);
          mu[e] = s_agg[ia];
          const float sd = s_agg[ia + f.C];
          const float dsd = s_dagg[ia + f.C];
          dvar2[e] = (cok && sd > std_floor) ? dsd / sd : 0.f;  // 2 dvar = dstd / std
          dmu[e] = cok ? s_dagg[ia] - dvar2[e] * mu[e] * (D - S0) / D : 0.f;
        }
#pragma unroll 1
        for (int vi = 0; vi < p.n_views; ++vi) {
          const Tap t = tap_of(f, ndcx[vi], ndcy[vi]);
          const int64_t vbase = ((int64_t)vi * f.H * f.W) * f.Cp + cq * 4;
          float s[4];
          sample4(f.data + vbase, t, s);
          const float wD = vok ? wv[vi] / D : 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dx = wD * (dmu[e] + dvar2[e] * (s[e] - mu[e]));
            if (dx != 0.f) {
              const int64_t g = vbase + e;
              if (t.w00 != 0.f) scatter_add<MODE>(dst, g + t.o00, t.w00 * dx, vmax);
              if (t.w01 != 0.f) scatter_add<MODE>(dst, g + t.o01, t.w01 * dx, vmax);
              if (t.w10 != 0.f) scatter_add<MODE>(dst, g + t.o10, t.w10 * dx, vmax);
              if (t.w11 != 0.f) scatter_add<MODE>(dst, g + t.o11, t.w11 * dx, vmax);
            }
          }
        }
        if (MODE == 1 && vmax) atomicMax(dst.lmax, vmax);
      }
    }
    __syncthreads();  // the LDS tiles are rewritten by the next group
  }
  if (MODE == 1) {  // the measuring pass leaves the maxima and nothing else
    if (tid < p.n_feats && s_max[tid]) atomicMax(b.fix_max + tid, s_max[tid]);
    return;
  }
  // ---- per-workgroup partials: [wg][A * F (+ F)] in the (A, F) order of the transposed weight
  float* part = b.partial + (int64_t)blockIdx.x * ((int64_t)p.A * p.F + p.F);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int a = tid + 256 * h;
    if (a < p.A)
      for (int o = 0; o < p.F; ++o) part[(int64_t)a * p.F + o] = dM[h][o];
  }
  if (tid < p.F) part[(int64_t)p.A * p.F + tid] = db;
}


// ---------------------------------------------------------------------------------------------------------------------
// The same backward, second form (round 5; the default where it applies: A + 1 <= VB2_AMAX aggregated features, F <= 32).
// Measured on MI355X at 64^3 x 16 views with development probes (HOLO_VIEWPOOL_BWD_PROBE): everything but pass 2 1.2 ms,
// pass 2 without its atomics 2.5 ms, with them 13.9 ms - and the atomics of ONE 16-channel map alone cost 0.9 / 1.4 / 2.1 /
// 3.1 ms at 64^2 / 32^2 / 16^2 / 8^2 for the same count: same-address contention (neighbouring voxels land in the same
// bilinear cell of a coarse map) is what they cost.  So:
//   * pass 2 runs with the layout turned round - a 16-lane row = the group's 16 voxels (consecutive in x) for one channel
//     quad - and sums the 4 taps x 4 channels of voxels that share a cell along the row (DPP row shifts, segmented by the
//     cell key) before anything is issued: 1.14 G -> ~0.4 G atomics, none of them contended inside a wave: 13.9 -> 7.5 ms;
//   * the runs' sums are issued TRANSPOSED through a row-private LDS tile: one run per instruction, lane j = (tap j / 4,
//     channel j % 4), the four rows of a wave (the four channel quads of the same pixels) completing 64-byte pixels.  Issued
//     from the run's first lane they were 16 one-word instructions that visited the same two or three cache lines again
//     and again (the first tap's four instructions alone cost 0.5 ms, all sixteen 5 ms): 7.5 -> 3.9 ms;
//   * shape: the 16 lanes of a voxel project it into ONE view each (NDC and angular weight of the (voxel, view) pairs in
//     LDS, no dynamically indexed private arrays); d M^T += agg^T dz (K = the group's 16 voxels) on the matrix cores
//     (v_mfma_f32_16x16x4_f32, <= 6 accumulator tiles per wave, a constant-1 column behind the aggregated features makes
//     row A of the product the bias gradient) instead of 2 x F values per thread in registers for the workgroup's lifetime;
//     LDS tiles at the stride the call needs (161 floats): 26 KB and 128 registers, four workgroups (16 waves) per CU where
//     the first form runs one wave per SIMD (on its own this changed little: 14.9 -> 13.9 ms).
// What is left: 2.5 ms of gathers and arithmetic (1.2 ms of it the forward's work again), 1.4 ms of atomics.
// Same arithmetic per voxel, same partial layout ([A][F] | [F]) and reduce as above.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int VB2_AMAX = 160;       // columns of the aggregated tile (A features + the constant 1, zero beyond): 10 MFMA row tiles
constexpr int VB2_AS = VB2_AMAX + 1;  // LDS row stride (odd: the 16 voxels of a column read land in 16 banks)
constexpr int VB2_ZS = VB_F + 1;


// value of lane + D inside the 16-lane row (0 beyond the row's end): DPP row_shl on the device, a shuffle in the host emulation
#ifndef HOLO_EMU
template <int D>
__device__ __forceinline__ float row_next_f(float v, int) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x100 + D, 0xf, 0xf, false));
}
template <int D>
__device__ __forceinline__ int row_next_i(int v, int) {
  return __builtin_amdgcn_update_dpp(0, v, 0x100 + D, 0xf, 0xf, false);
}
__device__ __forceinline__ int row_prev_i(int v, int) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); }
template <int D>
__device__ __forceinline__ int row_back_i(int v, int) {  // value of lane - D inside the row, 0 before the row's first lane
  return __builtin_amdgcn_update_dpp(0, v, 0x110 + D, 0xf, 0xf, false);
}
#else
template <int D>
__device__ __forceinline__ int row_back_i(int v, int lane) {
  const float o = __shfl(__uint_as_float((uint32_t)v), (lane & 15) >= D ? lane - D : lane);
  return (lane & 15) >= D ? (int)__float_as_uint(o) : 0;
}
template <int D>
__device__ __forceinline__ float row_next_f(float v, int lane) {
  const float o = __shfl(v, (lane & 15) + D < 16 ? lane + D : lane);
  return (lane & 15) + D < 16 ? o : 0.f;
}
template <int D>
__device__ __forceinline__ int row_next_i(int v, int lane) {
  const float o = __shfl(__uint_as_float((uint32_t)v), (lane & 15) + D < 16 ? lane + D : lane);
  return (lane & 15) + D < 16 ? (int)__float_as_uint(o) : 0;
}
__device__ __forceinline__ int row_prev_i(int v, int lane) {
  const float o = __shfl(__uint_as_float((uint32_t)v), (lane & 15) >= 1 ? lane - 1 : lane);
  return (lane & 15) >= 1 ? (int)__float_as_uint(o) : 0;
}
#endif
// one step of the segmented sum along a row: c[i] += c[i + D] where lane + D carries the same run number (run numbers rise
// along the row, so equal numbers D apart mean one run in between: after D = 1, 2, 4, 8 a run's first lane holds its total)
template <int D>
__device__ __forceinline__ void seg_step(float (&c)[16], int key, int lane) {
  const float m = row_next_i<D>(key, lane) == key ? 1.f : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = fmaf(row_next_f<D>(c[i], lane), m, c[i]);
}

constexpr int VB2_STAGE = 8;  // runs of a row staged per round of the scatter
// Scatter of one 16-lane row: lane = one voxel's 4 taps x 4 channels c[tap * 4 + channel] for the bilinear cell t of ONE
// feature-map view (element offset vb, channel quad included), act = the lane has something to add.  Every lane of the
// wave calls it; st = the row's private LDS tile (VB2_STAGE * 20 floats).
//   1. voxels of the row that share a cell with their neighbours are summed along the row (maximal contiguous runs of equal
//      keys: the voxels of a row are consecutive in x), the run's first lane holds the total;
//   2. the totals go out TRANSPOSED: issued from a run's first lane they would be 16 instructions of one word per run, each
//      visiting the same two or three cache lines again (measured in view_pool_bwd2_kernel: the first tap's four
//      instructions alone cost 0.5 ms, all sixteen 5 ms).  Staged through the tile, the row's 16 lanes issue ONE run per
//      instruction - lane j = (tap j / 4, channel j % 4) - and rows of a wave that carry neighbouring channel quads of the
//      same pixels complete 64-byte pixels: as many instructions as the row has runs, every cache line visited once per run.
template <int MODE>
__device__ __forceinline__ void row_scatter(float (&c)[16], const Tap& t, int vb, const ScatterDst& dst, bool act, int lane, float* st) {
  uint32_t vmax = 0u;
  const int pv = lane & 15;
  const int key = t.key;
  // runs = maximal CONTIGUOUS stretches of equal keys, numbered along the row (rid).  The sums are segmented by the run number,
  // not by the key: on a grid narrower than a row (R = 8: a row is two lines of voxels) the second line comes back to the first
  // line's cells, and equal keys across the gap must stay two runs - each is issued by its own first lane.
  const bool rhead = row_prev_i(key, lane) != key;  // first lane of a run (lane 0 of a row: 0 is no key)
  int rid = rhead ? 1 : 0;
  rid += row_back_i<1>(rid, lane);
  rid += row_back_i<2>(rid, lane);
  rid += row_back_i<4>(rid, lane);
  rid += row_back_i<8>(rid, lane);
  if (__any(act && !rhead)) {  // some voxels of a row share a cell with their neighbour: sum along the runs
    seg_step<1>(c, rid, lane);
    seg_step<2>(c, rid, lane);
    seg_step<4>(c, rid, lane);
    seg_step<8>(c, rid, lane);
  }
  const bool head = act && rhead;
  int inc = head ? 1 : 0;
  inc += row_back_i<1>(inc, lane);
  inc += row_back_i<2>(inc, lane);
  inc += row_back_i<4>(inc, lane);
  inc += row_back_i<8>(inc, lane);
  const int hidx = inc - (head ? 1 : 0);  // runs in front of this lane's
  const int nh = (int)__float_as_uint(__shfl(__uint_as_float((uint32_t)inc), lane | 15));  // runs of the row
  for (int r0 = 0; __any(r0 < nh); r0 += VB2_STAGE) {
    if (head && hidx >= r0 && hidx < r0 + VB2_STAGE) {
      float* d = st + (hidx - r0) * 20;
      *reinterpret_cast<float4*>(d) = make_float4(c[0], c[1], c[2], c[3]);
      *reinterpret_cast<float4*>(d + 4) = make_float4(c[4], c[5], c[6], c[7]);
      *reinterpret_cast<float4*>(d + 8) = make_float4(c[8], c[9], c[10], c[11]);
      *reinterpret_cast<float4*>(d + 12) = make_float4(c[12], c[13], c[14], c[15]);
      *reinterpret_cast<float4*>(d + 16) =
          make_float4(__uint_as_float((uint32_t)(vb + t.o00)), __uint_as_float((uint32_t)(vb + t.o01)),
                      __uint_as_float((uint32_t)(vb + t.o10)), __uint_as_float((uint32_t)(vb + t.o11)));
    }
    HOLO_WAVE_SYNC();
    const int cnt = nh - r0 < VB2_STAGE ? nh - r0 : VB2_STAGE;
    for (int i = 0; __any(i < cnt); ++i) {
      if (i < cnt) {
        const float val = st[i * 20 + pv];
        const int off = (int)__float_as_uint(st[i * 20 + 16 + (pv >> 2)]) + (pv & 3);
        if (val != 0.f) scatter_add<MODE>(dst, off, val, vmax);
      }
    }
    HOLO_WAVE_SYNC();  // the tile is rewritten by the next round / the next call
  }
  if (MODE == 1 && vmax) atomicMax(dst.lmax, vmax);
}

// WPS: waves per SIMD the register allocation aims at (4, the default: 128 registers + 172 bytes of scratch; 3: 166 registers)
template <int WPS, int MODE>
__global__ __launch_bounds__(256, WPS) void view_pool_bwd2_kernel(ViewPoolBwdParams b) {
  const ViewPoolParams& p = b.fwd;
  __shared__ uint32_t s_max[ViewPoolParams::MAX_FEATS];
  if (MODE == 1 && threadIdx.x < ViewPoolParams::MAX_FEATS) s_max[threadIdx.x] = 0u;
  __shared__ float s_agg[16 * VB2_AS];   // [voxel][aggregated feature | 1 | 0 ...]
  __shared__ float s_dagg[16 * VB2_AS];  // [voxel][d loss / d aggregated feature]
  __shared__ float s_dz[16 * VB2_ZS];    // [voxel][d loss / d mapper output] (zero beyond F)
  __shared__ float s_ndcx[16 * 16], s_ndcy[16 * 16], s_w[16 * 16];  // [voxel][view]
  __shared__ __attribute__((aligned(16))) float s_stage[16 * VB2_STAGE * 20];  // pass 2: [row][run][16 sums | 4 tap offsets]
  const int tid = threadIdx.x;
  const int vl = tid >> 4, ql = tid & 15;
  const int lane = tid & 63, wave = tid >> 6;
  const int R = p.R;
  const int64_t nvox = (int64_t)R * R * R;
  const int64_t ngroups = (nvox + 15) / 16;
  const float step = 2.0f / (float)(R - 1);
  auto lin = [&](int i) { return (i < R / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(R - 1 - i)) * p.half_extent; };
  const float std_floor = sqrtf(1e-4f);
  const int A = p.A, F = p.F;

  // the constant parts of the tiles: 1 behind the aggregated features (bias gradient), zeros beyond; dz columns beyond F
  for (int i = tid; i < 16 * VB2_AS; i += 256) {
    const int a = i % VB2_AS;
    s_agg[i] = a == A ? 1.f : 0.f;
  }
  for (int i = tid; i < 16 * VB2_ZS; i += 256) s_dz[i] = 0.f;

  // d M^T tiles of this wave: row tiles mt = wave, wave + 4, wave + 8 (< 10) x both column tiles
  f32x4 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int n_mt = (A + 1 + 15) >> 4;  // row tiles that hold anything

  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t v = grp * 16 + vl;
    const bool vok = v < nvox;
    const int64_t vc = vok ? v : nvox - 1;
    const int x = (int)(vc % R), y = (int)((vc / R) % R), z = (int)(vc / ((int64_t)R * R));
    const float px = lin(x), py = lin(y), pz = lin(z);
    // ---- lane ql: view ql of the voxel (direction of view 0 by every lane itself)
    {
      const ViewPoolParams::Cam& c0 = p.cams[0];
      float d0x = px - c0.centre[0], d0y = py - c0.centre[1], d0z = pz - c0.centre[2];
      const float n0 = fmaxf(sqrtf(d0x * d0x + d0y * d0y + d0z * d0z), 1e-12f);
      d0x /= n0;
      d0y /= n0;
      d0z /= n0;
      const int vi = ql < p.n_views ? ql : 0;
      const ViewPoolParams::Cam& c = p.cams[vi];
      const float cx = px * c.Rm[0] + py * c.Rm[3] + pz * c.Rm[6] + c.T[0];
      const float cy = px * c.Rm[1] + py * c.Rm[4] + pz * c.Rm[7] + c.T[1];
      float cz = px * c.Rm[2] + py * c.Rm[5] + pz * c.Rm[8] + c.T[2];
      if (fabsf(cz) < p.proj_eps) cz = cz < 0.f ? -p.proj_eps : p.proj_eps;
      float dx = px - c.centre[0], dy = py - c.centre[1], dz = pz - c.centre[2];
      const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
      dx /= nrm;
      dy /= nrm;
      dz /= nrm;
      float a = 0.5f * ((dx * d0x + dy * d0y + dz * d0z) + 1.0f);
      if (p.gamma != 1.0f) a = powf(a, p.gamma);
      s_ndcx[vl * 16 + ql] = c.focal[0] * cx / cz + c.pp[0];
      s_ndcy[vl * 16 + ql] = c.focal[1] * cy / cz + c.pp[1];
      s_w[vl * 16 + ql] = ql < p.n_views ? fmaxf(a, p.min_weight) : 0.f;
    }
    __syncthreads();
    float S0 = 0.f;
    for (int vi = 0; vi < p.n_views; ++vi) S0 += s_w[vl * 16 + vi];  // (view order, as the forward sums it)
    const float D = fmaxf(S0, 1e-2f);

    // ---- pass 1: the forward's aggregation, [AVG | STD] per key into the LDS tile
    for (int q = ql; q < p.n_quads; q += 16) {
      int k = 0;
      while (k + 1 < p.n_feats && q >= p.feat[k + 1].quad0) ++k;
      const ViewPoolParams::Feat& f = p.feat[k];
      const int cq = q - f.quad0;
      float S1[4] = {0.f, 0.f, 0.f, 0.f}, S2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
      for (int vi = 0; vi < p.n_views; ++vi) {
        const Tap t = tap_of(f, s_ndcx[vl * 16 + vi], s_ndcy[vl * 16 + vi]);
        const float w = s_w[vl * 16 + vi];
        float s[4];
        sample4(f.data + ((int64_t)vi * f.H * f.W) * f.Cp + cq * 4, t, s);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          S1[e] = fmaf(w, s[e], S1[e]);
          S2[e] = fmaf(w * s[e], s[e], S2[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = cq * 4 + e;
        if (c < f.C) {
          const float mu = S1[e] / D;
          const float var = (S2[e] - 2.f * mu * S1[e] + mu * mu * S0) / D;
          s_agg[vl * VB2_AS + f.out0 + c] = mu;
          s_agg[vl * VB2_AS + f.out0 + f.C + c] = sqrtf(fmaxf(var, 1e-4f));
        }
      }
    }
    __syncthreads();
    // ---- mapper forward, tanh, dz = g (1 - out^2); voxels beyond the grid contribute nothing
    for (int o = ql; o < F; o += 16) {
      float z0 = p.bias ? p.bias[o] : 0.f, z1 = 0.f;
      int a = 0;
      for (; a + 1 < A; a += 2) {  // two chains
        z0 = fmaf(s_agg[vl * VB2_AS + a], p.wt[(int64_t)a * F + o], z0);
        z1 = fmaf(s_agg[vl * VB2_AS + a + 1], p.wt[(int64_t)(a + 1) * F + o], z1);
      }
      if (a < A) z0 = fmaf(s_agg[vl * VB2_AS + a], p.wt[(int64_t)a * F + o], z0);
      const float out = tanhf(z0 + z1);
      const float g = vok ? b.gout[(int64_t)o * nvox + v] : 0.f;
      s_dz[vl * VB2_ZS + o] = g * (1.f - out * out);
    }
    __syncthreads();
    // ---- dagg = M^T dz (the lane's aggregated features a = ql, ql + 16, ...)
    if (b.want_feats) {
      for (int a = ql; a < A; a += 16) {
        float d0 = 0.f, d1 = 0.f;
        int o = 0;
        for (; o + 1 < F; o += 2) {
          d0 = fmaf(p.wt[(int64_t)a * F + o], s_dz[vl * VB2_ZS + o], d0);
          d1 = fmaf(p.wt[(int64_t)a * F + o + 1], s_dz[vl * VB2_ZS + o + 1], d1);
        }
        if (o < F) d0 = fmaf(p.wt[(int64_t)a * F + o], s_dz[vl * VB2_ZS + o], d0);
        s_dagg[vl * VB2_AS + a] = d0 + d1;
      }
    }
    // ---- d M^T (+ d bias as row A) += agg^T dz over the group's 16 voxels: D[m = feature][n = output], k = voxel
    {
      const int mrow = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int vox = ks * 4 + kq;
        const float b0 = s_dz[vox * VB2_ZS + mrow], b1 = s_dz[vox * VB2_ZS + 16 + mrow];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int mt = wave + 4 * i;
          if (mt < n_mt) {  // (wave-uniform)
            const float av = s_agg[vox * VB2_AS + mt * 16 + mrow];
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1, acc[i][1], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
    // ---- pass 2: the samples again, dx per view, scattered through the bilinear weights.  Thread layout turned round: a
    //      16-lane ROW = the group's 16 voxels (consecutive in x) for ONE channel quad, so voxels that land in the same
    //      bilinear cell of a map - most of a row on the coarse maps - sit in neighbouring lanes: their 4 taps x 4 channels
    //      are summed along the row (segmented by cell) and only a run's first lane issues atomics.  (Measured on MI355X
    //      with every voxel issuing its own: the 16-channel maps at 64^2 / 32^2 / 16^2 / 8^2 cost 0.9 / 1.4 / 2.1 / 3.1 ms
    //      of atomics for the same count - same-address contention, not the count, is what the atomics cost.)
    if (b.want_feats) {
      const int prow = tid >> 4, pv = tid & 15;
      const int64_t v2 = grp * 16 + pv;
      const bool vok2 = v2 < nvox;
      float S02 = 0.f;
      for (int vi = 0; vi < p.n_views; ++vi) S02 += s_w[pv * 16 + vi];
      const float D2 = fmaxf(S02, 1e-2f);
      for (int q0 = 0; q0 < p.n_quads; q0 += 16) {  // (uniform trip count: the votes below need every lane)
        const int qq = q0 + prow;
        const int q = qq < p.n_quads ? qq : p.n_quads - 1;
        int k = 0;
        while (k + 1 < p.n_feats && q >= p.feat[k + 1].quad0) ++k;
        const ViewPoolParams::Feat& f = p.feat[k];
        const ScatterDst dst = scatter_dst<MODE>(b, k, s_max);
#ifdef HOLO_DEV_PROBES  // (timing probes of a development build: want_feats 2 = pass 2 without its atomics, 10 + k = map k alone)
        const bool act = qq < p.n_quads && dst.f32 != nullptr && !(b.want_feats == 2 || (b.want_feats >= 10 && b.want_feats - 10 != k));
#else
        const bool act = qq < p.n_quads && dst.f32 != nullptr;
#endif
        const int cq = q - f.quad0;
        float mu[4], dmu[4], dvar2[4];  // dvar2 = 2 dvar
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = cq * 4 + e;
          const bool cok = c < f.C;
          const int ia = pv * VB2_AS + f.out0 + (cok ? c : 0);
          mu[e] = s_agg[ia];
          const float sd = s_agg[ia + f.C];
          const float dsd = s_dagg[ia + f.C];
          dvar2[e] = (cok && sd > std_floor) ? dsd / sd : 0.f;  // 2 dvar = dstd / std
          dmu[e] = cok ? s_dagg[ia] - dvar2[e] * mu[e] * (D2 - S02) / D2 : 0.f;
        }
#pragma unroll 1
        for (int vi = 0; vi < p.n_views; ++vi) {
          const Tap t = tap_of(f, s_ndcx[pv * 16 + vi], s_ndcy[pv * 16 + vi]);
          const int64_t vbase = ((int64_t)vi * f.H * f.W) * f.Cp + cq * 4;
          float s[4];
          sample4(f.data + vbase, t, s);
          const float wD = vok2 ? s_w[pv * 16 + vi] / D2 : 0.f;
          float c[16];  // [tap][channel]
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float dx = wD * (dmu[e] + dvar2[e] * (s[e] - mu[e]));
            c[e] = t.w00 * dx;
            c[4 + e] = t.w01 * dx;
            c[8 + e] = t.w10 * dx;
            c[12 + e] = t.w11 * dx;
          }
          row_scatter<MODE>(c, t, (int)vbase, dst, act, lane, s_stage + prow * (VB2_STAGE * 20));  // (a view's maps stay far below 2^31 elements)
        }
      }
    }
    __syncthreads();  // the LDS tiles are rewritten by the next group
  }
  if (MODE == 1) {  // the measuring pass leaves the maxima and nothing else
    if (tid < p.n_feats && s_max[tid]) atomicMax(b.fix_max + tid, s_max[tid]);
    return;
  }
  // ---- per-workgroup partials: rows 0 .. A of the product = [A][F] d M^T followed by the F values of d bias
  float* part = b.partial + (int64_t)blockIdx.x * ((int64_t)A * F + F);
  const int col = lane & 15;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int mt = wave + 4 * i;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = mt * 16 + 4 * (lane >> 4) + r, o = j * 16 + col;
        if (mt < n_mt && a <= A && o < F) part[(int64_t)a * F + o] = acc[i][j][r];
      }
  }
}

// dW (F, A) and db (F) from the per-workgroup partials: 64 elements per workgroup, its four waves sum a quarter of the
// partials each in workgroup order, the four sums are added in wave order (fixed order: deterministic)
__global__ __launch_bounds__(256) void viewpool_partial_reduce_kernel(const float* __restrict__ partial, int n_wgs, int A, int F,
                                                                      float* __restrict__ dW, float* __restrict__ dbias) {
  __shared__ float s_part[4 * 64];
  const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e;
  const int per = A * F + F;
  const int chunk = (n_wgs + 3) / 4;
  const int w0 = sl * chunk, w1 = min(w0 + chunk, n_wgs);
  float s = 0.f;
  if (i < per)
    for (int w = w0; w < w1; ++w) s += partial[(int64_t)w * per + i];
  s_part[sl * 64 + e] = s;
  __syncthreads();
  if (sl != 0 || i >= per) return;
  s = ((s_part[e] + s_part[64 + e]) + s_part[128 + e]) + s_part[192 + e];
  if (i < A * F) {
    const int a = i / F, o = i - a * F;
    if (dW) dW[(int64_t)o * A + a] = s;
  } else if (dbias) {
    dbias[i - A * F] = s;
  }
}

// (n, H, W, Cp) channels-last padded -> (n, C, H, W)
__global__ __launch_bounds__(256) void nhwc_pad_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                               int Cp, int64_t HW, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t px = i % HW;
    const int c = (int)((i / HW) % C);
    const int64_t n = i / (HW * C);
    out[i] = in[(n * HW + px) * Cp + c];
  }
}


// =====================================================================================================================
// MLPMeanFeatureAggregator (custom_modules.py:162-293), backward.  The forward kernel (mlp_mean_pool_kernel) works on the
// FOLDED parameters A = W1 Ws, Am = W1 Wm, b' = W1 (bs + bm) + b1, G = M Wl, g0 = M bl + mapper bias, l = Wl[0], l0 = bl[0]
// (viewpool_exec.cpp); so does the backward, and the host un-folds the gradients in float64.  Per voxel p, views v (weights 1):
//   x_v = [bilinear samples | harmonic(ray direction)],  mean = sum_v x_v / max(V, 1e-2)
//   pre_v = A x_v + Am mean + b',  h_v = LeakyReLU_0.2(pre_v),  logit_v = l.h_v + l0,  a = softmax_v(logit)
//   out = tanh(sum_v a_v G h_v + g0)
// Backward of g on out:  dz = g (1 - out^2);  du_v = a_v dz;  da_v = dz . (G h_v);  dlogit_v = a_v (da_v - sum_u a_u da_u)
//   dh_v = G^T du_v + dlogit_v l;  dpre_v = dh_v lrelu'(pre_v);  dc = sum_v dpre_v
//   dx_v = A^T dpre_v + Am^T dc / max(V, 1e-2)   -> feature columns scattered through the bilinear taps
//   dG = sum du_v h_v^T, dl = sum dlogit_v h_v, dl0 = sum dlogit_v, dg0 = sum dz, dA = sum dpre_v x_v^T, dAm = sum dc mean^T,
//   db' = sum dc
// A training-side path built from plain pieces rather than one fused kernel: every per-(voxel, view) row vector lives in the
// workspace (rows = view * P + voxel; ~3 GB at 64^3 with four views - 1 % of the HBM), the matrix products run on
// gemm_kernel (fp32 MFMA; the reductions over all rows as split-K batches with the transposed operand written explicitly by
// the producing kernel), the kernels below are the element-wise and per-voxel steps between them.
// =====================================================================================================================
struct VoxelProj {
  float ndcx, ndcy, d[3];
};
__device__ __forceinline__ VoxelProj project_voxel(const ViewPoolParams& vp, int vi, int64_t p) {
  const int R = vp.R;
  const float step = 2.0f / (float)(R - 1);
  auto lin = [&](int i) { return (i < R / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(R - 1 - i)) * vp.half_extent; };
  const int x = (int)(p % R), y = (int)((p / R) % R), z = (int)(p / ((int64_t)R * R));
  const float px = lin(x), py = lin(y), pz = lin(z);
  const ViewPoolParams::Cam& c = vp.cams[vi];
  const float cx = px * c.Rm[0] + py * c.Rm[3] + pz * c.Rm[6] + c.T[0];
  const float cy = px * c.Rm[1] + py * c.Rm[4] + pz * c.Rm[7] + c.T[1];
  float cz = px * c.Rm[2] + py * c.Rm[5] + pz * c.Rm[8] + c.T[2];
  if (fabsf(cz) < vp.proj_eps) cz = cz < 0.f ? -vp.proj_eps : vp.proj_eps;
  VoxelProj o;
  o.ndcx = c.focal[0] * cx / cz + c.pp[0];
  o.ndcy = c.focal[1] * cy / cz + c.pp[1];
  float dx = px - c.centre[0], dy = py - c.centre[1], dz = pz - c.centre[2];
  const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
  o.d[0] = dx / nrm, o.d[1] = dy / nrm, o.d[2] = dz / nrm;
  return o;
}

// X[row][dp]: the padded input rows of the forward kernel (feature maps at their quads, the embedding at emb0, zeros else)
__global__ __launch_bounds__(256) void mm_gather_kernel(MlpMeanBwdParams b) {
  const MlpMeanParams& m = b.fwd;
  const ViewPoolParams& vp = m.vp;
  const int dq = m.dp >> 2;
  const int64_t P = b.Pc;  // voxels of this chunk (rows = view * Pc + local voxel)
  const int64_t total = (int64_t)vp.n_views * P * dq;
  const int nh = m.n_harmonic;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % dq);
    const int64_t row = i / dq;
    const int vi = (int)(row / P);
    const int64_t p = row - (int64_t)vi * P;
    const VoxelProj pr = project_voxel(vp, vi, b.p0 + p);
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    bool done = false;
    for (int k = 0; k < vp.n_feats && !done; ++k) {
      const ViewPoolParams::Feat& f = vp.feat[k];
      if (q >= f.quad0 && q < f.quad0 + f.Cp / 4) {
        const Tap t = tap_of(f, pr.ndcx, pr.ndcy);
        float s[4];
        sample4(f.data + ((int64_t)vi * f.H * f.W) * f.Cp + (q - f.quad0) * 4, t, s);
        o = make_float4(s[0], s[1], s[2], s[3]);
        done = true;
      }
    }
    if (!done && q * 4 >= m.emb0) {
      float e[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = q * 4 + c - m.emb0;  // [sin(2^f d_a) | cos(2^f d_a) | d], index a * n + f
        float val = 0.f;
        if (j < 3 * nh) {
          val = sinf(pr.d[j / nh] * (float)(1 << (j % nh)));
        } else if (j < 6 * nh) {
          const int jj = j - 3 * nh;
          val = cosf(pr.d[jj / nh] * (float)(1 << (jj % nh)));
        } else if (j < 6 * nh + 3) {
          val = pr.d[j - 6 * nh];
        }
        e[c] = val;
      }
      o = make_float4(e[0], e[1], e[2], e[3]);
    }
    *reinterpret_cast<float4*>(b.X + row * m.dp + q * 4) = o;
  }
}

// MEAN[p][dp] = sum_v X[v, p] / max(V, 1e-2)
__global__ __launch_bounds__(256) void mm_mean_kernel(MlpMeanBwdParams b) {
  const MlpMeanParams& m = b.fwd;
  const int dq = m.dp >> 2, V = m.vp.n_views;
  const int64_t P = b.Pc;
  const float inv = 1.f / fmaxf((float)V, 1e-2f);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P * dq; i += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % dq);
    const int64_t p = i / dq;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int v = 0; v < V; ++v) {
      const float4 t = *reinterpret_cast<const float4*>(b.X + ((int64_t)v * P + p) * m.dp + q * 4);
      s.x += t.x, s.y += t.y, s.z += t.z, s.w += t.w;
    }
    *reinterpret_cast<float4*>(b.MEAN + p * m.dp + q * 4) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}

// PRE[row] += CM[p] + b';  H = LeakyReLU_0.2(PRE)
__global__ __launch_bounds__(256) void mm_hidden_kernel(MlpMeanBwdParams b) {
  const MlpMeanParams& m = b.fwd;
  const int64_t P = b.Pc;
  const int64_t total = (int64_t)m.vp.n_views * P * 32;  // 128 / 4 quads per row
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(i & 31);
    const int64_t row = i >> 5, p = row % P;
    float4 v = *reinterpret_cast<const float4*>(b.PRE + row * 128 + q * 4);
    const float4 c = *reinterpret_cast<const float4*>(b.CM + p * 128 + q * 4);
    const float4 cb = *reinterpret_cast<const float4*>(m.cb + q * 4);
    v.x += c.x + cb.x, v.y += c.y + cb.y, v.z += c.z + cb.z, v.w += c.w + cb.w;
    *reinterpret_cast<float4*>(b.PRE + row * 128 + q * 4) = v;
    *reinterpret_cast<float4*>(b.H + row * 128 + q * 4) =
        make_float4(v.x > 0.f ? v.x : 0.2f * v.x, v.y > 0.f ? v.y : 0.2f * v.y, v.z > 0.f ? v.z : 0.2f * v.z, v.w > 0.f ? v.w : 0.2f * v.w);
  }
}

// one wave per voxel: logits, softmax over the views, out, dz, du_v (columns 0..F-1 of DUL), dlogit_v (column F)
__global__ __launch_bounds__(256) void mm_head_kernel(MlpMeanBwdParams b) {
  const MlpMeanParams& m = b.fwd;
  const ViewPoolParams& vp = m.vp;
  const int lane = threadIdx.x & 63, F = vp.F, V = vp.n_views, FW = b.FW;
  const int64_t P = b.Pc;  // voxels of this chunk (rows = view * Pc + local voxel)
  const float l_lo = m.l[lane], l_hi = m.l[lane + 64];
  for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < P; p += (int64_t)gridDim.x * 4) {
    float logit[ViewPoolParams::MAX_VIEWS], a[ViewPoolParams::MAX_VIEWS], u[ViewPoolParams::MAX_VIEWS];
    float mx = -3.0e38f;
    for (int v = 0; v < V; ++v) {
      const float* h = b.H + ((int64_t)v * P + p) * 128;
      float part = h[lane] * l_lo + h[lane + 64] * l_hi;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
      logit[v] = part + m.l0;
      mx = fmaxf(mx, logit[v]);
      u[v] = lane < F ? b.U[((int64_t)v * P + p) * FW + lane] : 0.f;
    }
    float den = 0.f;
    for (int v = 0; v < V; ++v) {
      a[v] = expf(logit[v] - mx);
      den += a[v];
    }
    float z = lane < F ? m.g0[lane] : 0.f;
    for (int v = 0; v < V; ++v) {
      a[v] /= den;
      z = fmaf(a[v], u[v], z);
    }
    const float out = tanhf(z);
    const float dz = lane < F ? b.gout[(int64_t)lane * b.Pall + b.p0 + p] * (1.f - out * out) : 0.f;
    float s = 0.f, da[ViewPoolParams::MAX_VIEWS];
    for (int v = 0; v < V; ++v) {
      float part = dz * u[v];
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
      da[v] = part;
      s = fmaf(a[v], part, s);
    }
    for (int v = 0; v < V; ++v) {
      const int64_t row = (int64_t)v * P + p;
      const float val = lane < F ? a[v] * dz : (lane == F ? a[v] * (da[v] - s) : 0.f);
      if (lane < FW) {
        b.DUL[row * FW + lane] = val;
        b.DULT[(int64_t)lane * b.NRp + row] = val;
      }
    }
  }
}

// DPRE = (DH + dlogit l) lrelu'(PRE), in place of PRE, and its transpose DPRET (the A operand of the split-K product with X):
// a workgroup owns 64 rows x 128 hidden units, reads and writes them row-major (coalesced) and turns the tile through LDS so
// that the transposed copy goes out as 64 consecutive rows per hidden unit (a plain per-element transposed store was 4 ms)
__global__ __launch_bounds__(256) void mm_dpre_kernel(MlpMeanBwdParams b) {
  __shared__ float tile[64 * 129];
  const MlpMeanParams& m = b.fwd;
  const int64_t P = b.Pc, NR = P * m.vp.n_views;
  const int tid = threadIdx.x;
  for (int64_t r0 = (int64_t)blockIdx.x * 64; r0 < NR; r0 += (int64_t)gridDim.x * 64) {
    for (int e = tid; e < 64 * 128; e += 256) {
      const int r = e >> 7, k = e & 127;
      const int64_t row = r0 + r;
      float d = 0.f;
      if (row < NR) {
        const int64_t i = row * 128 + k;
        const float dh = b.H[i] + b.DUL[row * b.FW + m.vp.F] * m.l[k];
        d = b.PRE[i] > 0.f ? dh : 0.2f * dh;
        b.PRE[i] = d;
      }
      tile[r * 129 + k] = d;
    }
    __syncthreads();
    for (int e = tid; e < 64 * 128; e += 256) {
      const int r = e & 63, k = e >> 6;
      if (r0 + r < NR) b.DPRET[(int64_t)k * b.NRp + r0 + r] = tile[r * 129 + k];
    }
    __syncthreads();
  }
}

// DC[p] = sum_v DPRE[v, p], and its transpose (the same 64 x 128 tile turn)
__global__ __launch_bounds__(256) void mm_dc_kernel(MlpMeanBwdParams b) {
  __shared__ float tile[64 * 129];
  const MlpMeanParams& m = b.fwd;
  const int V = m.vp.n_views;
  const int64_t P = b.Pc;
  const int tid = threadIdx.x;
  for (int64_t p0 = (int64_t)blockIdx.x * 64; p0 < P; p0 += (int64_t)gridDim.x * 64) {
    for (int e = tid; e < 64 * 128; e += 256) {
      const int r = e >> 7, k = e & 127;
      const int64_t p = p0 + r;
      float s = 0.f;
      if (p < P) {
        for (int v = 0; v < V; ++v) s += b.PRE[((int64_t)v * P + p) * 128 + k];
        b.DC[p * 128 + k] = s;
      }
      tile[r * 129 + k] = s;
    }
    __syncthreads();
    for (int e = tid; e < 64 * 128; e += 256) {
      const int r = e & 63, k = e >> 6;
      if (p0 + r < P) b.DCT[(int64_t)k * b.Pp + p0 + r] = tile[r * 129 + k];
    }
    __syncthreads();
  }
}

// d x_v = DX[row] + DCA[p] / max(V, 1e-2): the feature columns through the bilinear taps into the channels-last gradient maps.
// Thread layout and scatter of view_pool_bwd2_kernel's pass 2 (row_scatter): a 16-lane row = 16 consecutive voxels of one
// view for one channel quad (round 5: every thread issuing its own 16 atomics took 5.6 ms of the 14 ms backward at 64^3 x 4
// views).  Row items = (view, group of 16 voxels, quad), quad fastest: the four rows of a wave read 64 contiguous bytes of a
// DX row and complete 64-byte pixels in the scatter.
template <int MODE>
__global__ __launch_bounds__(256) void mm_scatter_kernel(MlpMeanBwdParams b) {
  __shared__ uint32_t s_max[ViewPoolParams::MAX_FEATS];
  if (MODE == 1 && threadIdx.x < ViewPoolParams::MAX_FEATS) s_max[threadIdx.x] = 0u;
  if (MODE == 1) __syncthreads();
  __shared__ __attribute__((aligned(16))) float s_stage[16 * VB2_STAGE * 20];
  const MlpMeanParams& m = b.fwd;
  const ViewPoolParams& vp = m.vp;
  const int tid = threadIdx.x, lane = tid & 63, prow = tid >> 4, pv = tid & 15;
  const int fq = m.emb0 >> 2;  // feature quads come first in the padded order
  const int64_t P = b.Pc;      // voxels of this chunk (rows = view * Pc + local voxel)
  const int64_t ngrp = (P + 15) / 16;
  const int64_t nitems = (int64_t)vp.n_views * ngrp * fq;
  const float inv = 1.f / fmaxf((float)vp.n_views, 1e-2f);
  float* st = s_stage + prow * (VB2_STAGE * 20);
  for (int64_t base = (int64_t)blockIdx.x * 16; base < nitems; base += (int64_t)gridDim.x * 16) {  // (uniform trip count)
    const int64_t item = base + prow;
    const int64_t itc = item < nitems ? item : nitems - 1;
    const int q = (int)(itc % fq);
    const int64_t tg = itc / fq;
    const int64_t g = tg % ngrp;
    const int vi = (int)(tg / ngrp);
    const int64_t pl = g * 16 + pv;
    const int64_t p = pl < P ? pl : P - 1;
    const int64_t row = (int64_t)vi * P + p;
    int k = 0;
    while (k + 1 < vp.n_feats && q >= vp.feat[k + 1].quad0) ++k;
    const ViewPoolParams::Feat& f = vp.feat[k];
    const ScatterDst dst = scatter_dst<MODE>(b, k, s_max);
    const bool act = item < nitems && pl < P && dst.f32 != nullptr;
    const float4 dx = *reinterpret_cast<const float4*>(b.DX + row * m.dp + q * 4);
    const float4 dc = *reinterpret_cast<const float4*>(b.DCA + p * m.dp + q * 4);
    const int cq = q - f.quad0;
    float g4[4] = {dx.x + dc.x * inv, dx.y + dc.y * inv, dx.z + dc.z * inv, dx.w + dc.w * inv};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (!act || cq * 4 + e >= f.C) g4[e] = 0.f;
    const VoxelProj pr = project_voxel(vp, vi, b.p0 + p);
    const Tap t = tap_of(f, pr.ndcx, pr.ndcy);
    float c[16];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      c[e] = t.w00 * g4[e];
      c[4 + e] = t.w01 * g4[e];
      c[8 + e] = t.w10 * g4[e];
      c[12 + e] = t.w11 * g4[e];
    }
    const int vb = (int)(((int64_t)vi * f.H * f.W) * f.Cp + cq * 4);
    row_scatter<MODE>(c, t, vb, dst, act, lane, st);
  }
  if (MODE == 1) {
    __syncthreads();
    if (tid < vp.n_feats && s_max[tid]) atomicMax(b.fix_max + tid, s_max[tid]);
  }
}

// deterministic mode: the sums leave the fixed-point image (zero again afterwards) and are ADDED to the float map
__global__ __launch_bounds__(256) void fix_flush_kernel(long long* __restrict__ fix, const uint32_t* __restrict__ maxbits,
                                                        float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t mb = *maxbits;
  if (mb == 0u) return;
  if (mb >= 0x7f800000u) {
    out[i] = __uint_as_float(0x7fc00000u);
    return;
  }
  const long long q = fix[i];
  if (q != 0) {
    out[i] += holo_fix_value(q, holo_fix_shift(mb));
    fix[i] = 0;
  }
}

// column sums of a (rows, cols <= 256) matrix: block b sums its row range -> partial[b][cols].  256 / cols row lanes per block
// (thread = (row lane, column): consecutive threads read consecutive columns), combined through LDS in lane order.
__global__ __launch_bounds__(256) void mm_colsum_kernel(const float* __restrict__ src, int64_t rows, int cols, int ld,
                                                        float* __restrict__ partial) {
  __shared__ float red[256];
  const int lanes = 256 / cols, tid = threadIdx.x;
  const int c = tid % cols, rl = tid / cols;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  float s = 0.f;
  if (rl < lanes)
    for (int64_t r = r0 + rl; r < r1; r += lanes) s += src[r * ld + c];
  red[tid] = s;
  __syncthreads();
  if (tid < cols) {
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[l * cols + tid];
    partial[(int64_t)blockIdx.x * cols + tid] = t;
  }
}

// out[i] (+)= sum_s partial[s][i], s in order
__global__ __launch_bounds__(256) void mm_sum_partials_kernel(const float* __restrict__ partial, int S, int64_t n,
                                                              float* __restrict__ out, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? out[i] : 0.f;  // (the voxel chunks of holo_mlp_mean_backward add up in chunk order: deterministic)
  for (int k = 0; k < S; ++k) s += partial[(int64_t)k * n + i];
  out[i] = s;
}

}  // namespace

int view_pool_bwd_launch(const ViewPoolBwdParams& b, int n_wgs, void* stream) {
  if (b.fwd.F > VB_F || b.fwd.A > 512) {
    set_error("view_pool_backward: feature_size <= %d and <= 512 aggregated features (got %d, %d)", VB_F, b.fwd.F, b.fwd.A);
    return -1;
  }
  // (HOLO_VIEWPOOL_BWD_V1=1: the register-accumulating form on every call - the knob tests/test_viewpool.py flips per call)
  const char* ev = getenv("HOLO_VIEWPOOL_BWD_V1");
  const bool v1 = ev && ev[0] == '1';
  const bool fixed = b.want_feats && b.fix_max != nullptr;  // deterministic mode: measure the addends, then add in fixed point
  if (!v1 && b.fwd.A + 1 <= VB2_AMAX) {
    ViewPoolBwdParams q = b;
#ifdef HOLO_DEV_PROBES  // timing probes of a development build only (-DHOLO_DEV_PROBES): they DROP gradients
    const char* ep = getenv("HOLO_VIEWPOOL_BWD_PROBE");  // 0 no pass 2, 2 pass 2 without its atomics, 10 + k the atomics of map k alone
    if (ep && q.want_feats) {
      q.want_feats = atoi(ep);
      fprintf(stderr, "[holo] HOLO_VIEWPOOL_BWD_PROBE=%d: feature-map gradients are INCOMPLETE (timing probe)\n", q.want_feats);
    }
    const char* eo = getenv("HOLO_VIEWPOOL_BWD_OCC");  // 3 = the 166-register build (measured 6.3 vs 3.9 ms)
    if (eo && eo[0] == '3' && !fixed) {
      HOLO_LAUNCH((view_pool_bwd2_kernel<3, 0>), dim3((unsigned)n_wgs), dim3(256), stream, q);
    } else
#endif
    if (fixed) {
      HOLO_LAUNCH((view_pool_bwd2_kernel<4, 1>), dim3((unsigned)n_wgs), dim3(256), stream, q);
      HOLO_LAUNCH((view_pool_bwd2_kernel<4, 2>), dim3((unsigned)n_wgs), dim3(256), stream, q);
    } else {
      HOLO_LAUNCH((view_pool_bwd2_kernel<4, 0>), dim3((unsigned)n_wgs), dim3(256), stream, q);
    }
  } else if (fixed) {
    HOLO_LAUNCH(view_pool_bwd_kernel<1>, dim3((unsigned)n_wgs), dim3(256), stream, b);
    HOLO_LAUNCH(view_pool_bwd_kernel<2>, dim3((unsigned)n_wgs), dim3(256), stream, b);
  } else {
    HOLO_LAUNCH(view_pool_bwd_kernel<0>, dim3((unsigned)n_wgs), dim3(256), stream, b);
  }
  const int per = b.fwd.A * b.fwd.F + b.fwd.F;
  HOLO_LAUNCH(viewpool_partial_reduce_kernel, dim3((unsigned)((per + 63) / 64)), dim3(256), stream, (const float*)b.partial, n_wgs,
              b.fwd.A, b.fwd.F, b.dW, b.dbias);
  return 0;
}

int nhwc_pad_to_nchw_launch(const float* in, float* out, int n, int C, int Cp, int64_t HW, void* stream) {
  const int64_t total = (int64_t)n * C * HW;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  HOLO_LAUNCH(nhwc_pad_to_nchw_kernel, dim3((unsigned)blocks), dim3(256), stream, in, out, C, Cp, HW, total);
  return 0;
}

int fix_flush_launch(long long* fix, const uint32_t* maxbits, float* out, int64_t n, void* stream) {
  HOLO_LAUNCH(fix_flush_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), stream, fix, maxbits, out, n);
  return 0;
}

static unsigned mm_blocks(int64_t total) {
  int64_t bl = (total + 255) / 256;
  return (unsigned)(bl < 1 ? 1 : (bl > 65535 ? 65535 : bl));
}
int mm_bwd_step_launch(const MlpMeanBwdParams& b, int step, void* stream) {
  const int64_t P = b.Pc, NR = P * b.fwd.vp.n_views;
  switch (step) {
    case 0:
      HOLO_LAUNCH(mm_gather_kernel, dim3(mm_blocks(NR * (b.fwd.dp / 4))), dim3(256), stream, b);
      HOLO_LAUNCH(mm_mean_kernel, dim3(mm_blocks(P * (b.fwd.dp / 4))), dim3(256), stream, b);
      return 0;
    case 1:
      HOLO_LAUNCH(mm_hidden_kernel, dim3(mm_blocks(NR * 32)), dim3(256), stream, b);
      return 0;
    case 2:
      HOLO_LAUNCH(mm_head_kernel, dim3(mm_blocks(P * 64)), dim3(256), stream, b);
      return 0;
    case 3:
      HOLO_LAUNCH(mm_dpre_kernel, dim3(mm_blocks(NR * 4)), dim3(256), stream, b);  // one workgroup per 64 rows
      HOLO_LAUNCH(mm_dc_kernel, dim3(mm_blocks(P * 4)), dim3(256), stream, b);
      return 0;
    case 4:
      if (b.fix_max) {  // deterministic mode: measure the addends, then add in fixed point
        HOLO_LAUNCH(mm_scatter_kernel<1>, dim3(mm_blocks(NR * (b.fwd.emb0 / 4))), dim3(256), stream, b);
        HOLO_LAUNCH(mm_scatter_kernel<2>, dim3(mm_blocks(NR * (b.fwd.emb0 / 4))), dim3(256), stream, b);
      } else {
        HOLO_LAUNCH(mm_scatter_kernel<0>, dim3(mm_blocks(NR * (b.fwd.emb0 / 4))), dim3(256), stream, b);
      }
      return 0;
  }
  return -1;
}
int mm_colsum_launch(const float* src, int64_t rows, int cols, int ld, float* partial, int n_blocks, float* out, void* stream,
                     int accumulate) {
  if (cols > 256) {
    set_error("mm_colsum: at most 256 columns");
    return -1;
  }
  HOLO_LAUNCH(mm_colsum_kernel, dim3((unsigned)n_blocks), dim3(256), stream, src, rows, cols, ld, partial);
  HOLO_LAUNCH(mm_sum_partials_kernel, dim3(1), dim3(256), stream, (const float*)partial, n_blocks, (int64_t)cols, out, accumulate);
  return 0;
}
int mm_sum_partials_launch(const float* partial, int S, int64_t n, float* out, void* stream, int accumulate) {
  HOLO_LAUNCH(mm_sum_partials_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), stream, partial, S, n, out, accumulate);
  return 0;
}

}  // namespace holo
