// kernels_viewpool.hip — source-view feature maps -> voxel feature grid (the encoder-side entry of the model).
//
// Replaces, fused in one kernel (holo_diffusion/holo_diffusion_model.py:340-373 and the PyTorch3D pieces behind it):
//   VolumeLocator.get_coord_grid                                   voxel centres
//   ViewSampler / project_points_and_sample                        camera.transform_points (NDC) + ndc_grid_sample
//                                                                  (bilinear, zeros padding, align_corners=False); masks = 1
//   _get_point_to_source_camera_ray_dirs (custom_modules.py:279-334) + AngleWeightedReductionFeatureAggregator
//                                                                  w_v = clamp((0.5 (d_v.d_0 + 1))^gamma, min); [AVG | STD]
//                                                                  per feature key (configs/apple.yaml:183-196)
//   pooled_feature_mapper (LazyLinear -> feature_size, :113,368)   + tanh (:373), written straight into the NCDHW grid
//
// A gather-bound kernel: per voxel n_views x 4 taps x sum(C) floats are read from the (small, L2/Infinity-Cache resident)
// channels-last feature maps - a tap is one contiguous 16-byte read per thread.  Workgroup = 16 voxels x 16 lanes; a lane
// owns channel quads q, q + 16, ... and keeps the weighted moments S1 = sum w f, S2 = sum w f^2 of its four channels over
// the views (S0 = sum w is per voxel); AVG = S1 / D and the weighted variance (S2 - 2 AVG S1 + AVG^2 S0) / D with
// D = max(S0, 1e-2) follow in ONE pass over the views (algebraically the reference's wmean((x - AVG)^2)).  The 2 sum(C)
// aggregated features of the 16 voxels then meet in LDS for the mapper's dot products.
#include <math.h>

#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {
namespace {

// (n, C, H, W) -> (n, H, W, Cp) channels-last, channels zero-padded to a multiple of 4
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                               int Cp, int64_t HW, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    const int64_t px = (i / Cp) % HW, n = i / ((int64_t)Cp * HW);
    out[i] = c < C ? in[(n * C + c) * HW + px] : 0.f;
  }
}

// mapper weight (F, A) -> (A, F) so that consecutive threads (outputs) read consecutive floats
__global__ __launch_bounds__(256) void transpose_small_kernel(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                              int cols) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * cols) out[(i % cols) * rows + i / cols] = in[i];
}

__global__ __launch_bounds__(256) void view_pool_kernel(ViewPoolParams p) {
  __shared__ float s_agg[16 * ViewPoolParams::MAX_AGG];  // [voxel][aggregated feature]
  const int tid = threadIdx.x;
  const int vl = tid >> 4, ql = tid & 15;
  const int R = p.R;
  const int64_t nvox = (int64_t)R * R * R;
  const int64_t v = (int64_t)blockIdx.x * 16 + vl;
  const int64_t vc = v < nvox ? v : nvox - 1;
  const int x = (int)(vc % R), y = (int)((vc / R) % R), z = (int)(vc / ((int64_t)R * R));
  // voxel centre: linspace(-1, 1, R) * half extent (torch.linspace: symmetric evaluation around the midpoint)
  const float step = 2.0f / (float)(R - 1);
  auto lin = [&](int i) { return (i < R / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(R - 1 - i)) * p.half_extent; };
  const float px = lin(x), py = lin(y), pz = lin(z);

  // per view: NDC projection and the angular weight against view 0
  float ndcx[ViewPoolParams::MAX_VIEWS], ndcy[ViewPoolParams::MAX_VIEWS], wv[ViewPoolParams::MAX_VIEWS];
  float d0x = 0.f, d0y = 0.f, d0z = 0.f, S0 = 0.f;
#pragma unroll 1
  for (int vi = 0; vi < p.n_views; ++vi) {
    const ViewPoolParams::Cam& c = p.cams[vi];
    const float cx = px * c.Rm[0] + py * c.Rm[3] + pz * c.Rm[6] + c.T[0];  // X_cam = X R + T (row vectors)
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
    if (vi == 0) {
      d0x = dx;
      d0y = dy;
      d0z = dz;
    }
    float a = 0.5f * ((dx * d0x + dy * d0y + dz * d0z) + 1.0f);
    if (p.gamma != 1.0f) a = powf(a, p.gamma);
    wv[vi] = fmaxf(a, p.min_weight);
    S0 += wv[vi];
  }
  const float D = fmaxf(S0, 1e-2f);

  // channel quads of all feature maps: the lane walks q = ql, ql + 16, ...
  for (int q = ql; q < p.n_quads; q += 16) {
    int k = 0;
    while (k + 1 < p.n_feats && q >= p.feat[k + 1].quad0) ++k;
    const ViewPoolParams::Feat& f = p.feat[k];
    const int cq = q - f.quad0;  // quad inside the map
    float S1[4] = {0.f, 0.f, 0.f, 0.f}, S2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int vi = 0; vi < p.n_views; ++vi) {
      // ndc_grid_sample: grid = -ndc with the longer side divided by the aspect ratio; align_corners = False
      float gx = -ndcx[vi], gy = -ndcy[vi];
      if (f.W >= f.H) gx /= (float)f.W / (float)f.H; else gy /= (float)f.H / (float)f.W;
      const float ix = ((gx + 1.f) * (float)f.W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)f.H - 1.f) * 0.5f;
      const float fx0 = floorf(ix), fy0 = floorf(iy);
      const float tx = ix - fx0, ty = iy - fy0;
      // zeros padding as tap weights; the loads themselves go to clamped (always valid) addresses
      const float fW = (float)(f.W - 1), fH = (float)(f.H - 1);
      const float wx0 = (fx0 >= 0.f && fx0 <= fW) ? 1.f - tx : 0.f, wx1 = (fx0 >= -1.f && fx0 <= fW - 1.f) ? tx : 0.f;
      const float wy0 = (fy0 >= 0.f && fy0 <= fH) ? 1.f - ty : 0.f, wy1 = (fy0 >= -1.f && fy0 <= fH - 1.f) ? ty : 0.f;
      const int x0 = (int)fminf(fmaxf(fx0, 0.f), fW), x1 = (int)fminf(fmaxf(fx0 + 1.f, 0.f), fW);
      const int y0 = (int)fminf(fmaxf(fy0, 0.f), fH), y1 = (int)fminf(fmaxf(fy0 + 1.f, 0.f), fH);
      const float* base = f.data + ((int64_t)vi * f.H * f.W) * f.Cp + cq * 4;
      const float4 t00 = *reinterpret_cast<const float4*>(base + ((int64_t)y0 * f.W + x0) * f.Cp);
      const float4 t01 = *reinterpret_cast<const float4*>(base + ((int64_t)y0 * f.W + x1) * f.Cp);
      const float4 t10 = *reinterpret_cast<const float4*>(base + ((int64_t)y1 * f.W + x0) * f.Cp);
      const float4 t11 = *reinterpret_cast<const float4*>(base + ((int64_t)y1 * f.W + x1) * f.Cp);
      const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
      const float s[4] = {t00.x * w00 + t01.x * w01 + t10.x * w10 + t11.x * w11,
                          t00.y * w00 + t01.y * w01 + t10.y * w10 + t11.y * w11,
                          t00.z * w00 + t01.z * w01 + t10.z * w10 + t11.z * w11,
                          t00.w * w00 + t01.w * w01 + t10.w * w10 + t11.w * w11};
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
        s_agg[vl * ViewPoolParams::MAX_AGG + f.out0 + c] = mu;                              // [AVG_k | STD_k] per key
        s_agg[vl * ViewPoolParams::MAX_AGG + f.out0 + f.C + c] = sqrtf(fmaxf(var, 1e-4f));
      }
    }
  }
  __syncthreads();
  // pooled_feature_mapper + tanh: thread (voxel, lane) computes outputs lane, lane + 16, ...
  if (v < nvox) {
    for (int o = ql; o < p.F; o += 16) {
      float acc = p.bias ? p.bias[o] : 0.f;
      for (int a = 0; a < p.A; ++a) acc = fmaf(s_agg[vl * ViewPoolParams::MAX_AGG + a], p.wt[(int64_t)a * p.F + o], acc);
      p.out[(int64_t)o * nvox + v] = tanhf(acc);
    }
  }
}

}  // namespace

int nchw_to_nhwc_pad_launch(const float* in, float* out, int n, int C, int Cp, int64_t HW, void* stream) {
  const int64_t total = (int64_t)n * HW * Cp;
  int64_t blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  HOLO_LAUNCH(nchw_to_nhwc_pad_kernel, dim3((unsigned)blocks), dim3(256), stream, in, out, C, Cp, HW, total);
  return 0;
}

int transpose_small_launch(const float* in, float* out, int rows, int cols, void* stream) {
  HOLO_LAUNCH(transpose_small_kernel, dim3((unsigned)cdiv((int64_t)rows * cols, 256)), dim3(256), stream, in, out, rows, cols);
  return 0;
}

int view_pool_launch(const ViewPoolParams& p, void* stream) {
  if (p.n_views < 1 || p.n_views > ViewPoolParams::MAX_VIEWS || p.n_feats < 1 || p.n_feats > ViewPoolParams::MAX_FEATS ||
      p.A > ViewPoolParams::MAX_AGG || p.R < 2) {
    set_error("view_pool: 1..%d views, 1..%d feature maps, at most %d aggregated features", ViewPoolParams::MAX_VIEWS,
              ViewPoolParams::MAX_FEATS, ViewPoolParams::MAX_AGG);
    return -1;
  }
  const int64_t nvox = (int64_t)p.R * p.R * p.R;
  HOLO_LAUNCH(view_pool_kernel, dim3((unsigned)cdiv(nvox, 16)), dim3(256), stream, p);
  return 0;
}

}  // namespace holo
