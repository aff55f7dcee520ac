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
#include <stdint.h>

#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {
namespace {

constexpr int VB_F = 32;  // output features held per thread for dM (pooled_feature_mapper is feature_size = 32 wide)

struct Tap {
  int o00, o01, o10, o11;  // element offsets of the four taps inside one view's (H, W, Cp) map, channel 0
  float w00, w01, w10, w11;
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

__global__ __launch_bounds__(256) void view_pool_bwd_kernel(ViewPoolBwdParams b) {
  const ViewPoolParams& p = b.fwd;
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
        float* gmap = b.gfeat[k];
        if (!gmap) continue;
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
              float* g = gmap + vbase + e;
              if (t.w00 != 0.f) HOLO_ATOMIC_ADD_F32(g + t.o00, t.w00 * dx);
              if (t.w01 != 0.f) HOLO_ATOMIC_ADD_F32(g + t.o01, t.w01 * dx);
              if (t.w10 != 0.f) HOLO_ATOMIC_ADD_F32(g + t.o10, t.w10 * dx);
              if (t.w11 != 0.f) HOLO_ATOMIC_ADD_F32(g + t.o11, t.w11 * dx);
            }
          }
        }
      }
    }
    __syncthreads();  // the LDS tiles are rewritten by the next group
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

// dW (F, A) and db (F) from the per-workgroup partials, summed in workgroup order
__global__ __launch_bounds__(256) void viewpool_partial_reduce_kernel(const float* __restrict__ partial, int n_wgs, int A, int F,
                                                                      float* __restrict__ dW, float* __restrict__ dbias) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int per = A * F + F;
  if (i >= per) return;
  float s = 0.f;
  for (int w = 0; w < n_wgs; ++w) s += partial[(int64_t)w * per + i];
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

}  // namespace

int view_pool_bwd_launch(const ViewPoolBwdParams& b, int n_wgs, void* stream) {
  if (b.fwd.F > VB_F || b.fwd.A > 512) {
    set_error("view_pool_backward: feature_size <= %d and <= 512 aggregated features (got %d, %d)", VB_F, b.fwd.F, b.fwd.A);
    return -1;
  }
  HOLO_LAUNCH(view_pool_bwd_kernel, dim3((unsigned)n_wgs), dim3(256), stream, b);
  const int per = b.fwd.A * b.fwd.F + b.fwd.F;
  HOLO_LAUNCH(viewpool_partial_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), stream, (const float*)b.partial, n_wgs,
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

}  // namespace holo
