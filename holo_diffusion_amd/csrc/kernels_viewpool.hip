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


// ---------------------------------------------------------------------------------------------------------------------
// mlp_mean_pool_kernel - the REFERENCE's own learnt aggregator, MLPMeanFeatureAggregator (custom_modules.py:162-293;
// configs/hydrant.yaml:184, old_base_config.yaml:205), fused with the sampling in front of it and the mapper + tanh
// behind it (holo_diffusion_model.py:358-373).  Per voxel p and source view v:
//     x_v    = [bilinear samples of every feature map | harmonic(normalize(p - C_v))] * w_v          (w_v = 1: unmasked
//              sampling, exclude_target_view forced off, holo_diffusion_model.py:114-116)
//     mean   = sum_v x_v w_v / max(sum_v w_v, 1e-2)
//     h_v    = LeakyReLU_0.2(W1 (Ws x_v + bs + Wm mean + bm) + b1)     (MLPWithInputSkips(n_layers=1): its only layer is
//              the LAST one, which is where the construction quirk puts the hidden activation, custom_modules.py:108-112)
//     o_v    = Wl h_v + bl ;   agg = sum_v o_v softmax_v(o_v[0]) ;   out = tanh(M agg + bm)
// Everything between x_v and h_v is affine, and so is everything between h_v and `out` up to the softmax weights, so
// viewpool_exec.cpp folds (float64): A = W1 Ws, Am = W1 Wm, b' = W1 (bs + bm) + b1, G = M Wl, g0 = M bl + bm,
// l = Wl[0], l0 = bl[0]:
//     h_v = LeakyReLU(A x_v + (Am mean + b')) ;  logit_v = l.h_v + l0 ;  out = tanh(sum_v softmax(logit)_v G h_v + g0)
// GEMM-shaped work on the fp32 matrix cores, renderer-style: a wave owns 32 voxels (MFMA columns); lanes l and l+32
// share voxel l&31 and split the K dimension (the two k indices of v_mfma_f32_32x32x2_f32).
//   pass 1 over the views: gather x_v into the wave's LDS tile, accumulate the mean in registers;
//   c = Am mean + b' on the matrix cores (A operand straight from L2: once per 32 voxels);
//   pass 2 over the views: gather again (the maps are cache resident; keeping V tiles on chip is not an option),
//     acc = c, acc += A x_v (A rows from LDS), LeakyReLU, logit (lane-local dot + one cross-half add), G h_v with the
//     ACCUMULATOR REGISTERS AS B OPERANDS: register r of hidden tile t holds rows t*32 + rho(r) + 4*half, exactly the two k
//     indices of one MFMA step when the A operand is G[:, that row]; online softmax over the views.
// LDS: A 128 x (Dp+4) | G 32 x 132 | l | x tiles of the 4 waves.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MM_NH = 128;  // MLPMeanFeatureAggregator.n_hidden

struct ViewProj {
  float ndcx, ndcy, dx, dy, dz;
};
__device__ __forceinline__ ViewProj project_view(const ViewPoolParams::Cam& c, float px, float py, float pz, float eps) {
  ViewProj o;
  const float cx = px * c.Rm[0] + py * c.Rm[3] + pz * c.Rm[6] + c.T[0];  // X_cam = X R + T (row vectors)
  const float cy = px * c.Rm[1] + py * c.Rm[4] + pz * c.Rm[7] + c.T[1];
  float cz = px * c.Rm[2] + py * c.Rm[5] + pz * c.Rm[8] + c.T[2];
  if (fabsf(cz) < eps) cz = cz < 0.f ? -eps : eps;
  o.ndcx = c.focal[0] * cx / cz + c.pp[0];
  o.ndcy = c.focal[1] * cy / cz + c.pp[1];
  float dx = px - c.centre[0], dy = py - c.centre[1], dz = pz - c.centre[2];
  const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
  o.dx = dx / nrm;
  o.dy = dy / nrm;
  o.dz = dz / nrm;
  return o;
}
// one channel quad of a bilinear sample (ndc_grid_sample: grid = -ndc, longer side divided by the aspect ratio,
// align_corners = False, zeros padding as tap weights)
__device__ __forceinline__ float4 bilinear_quad(const ViewPoolParams::Feat& f, int vi, int cq, float ndcx, float ndcy) {
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
  const float* base = f.data + ((int64_t)vi * f.H * f.W) * f.Cp + cq * 4;
  const float4 t00 = *reinterpret_cast<const float4*>(base + ((int64_t)y0 * f.W + x0) * f.Cp);
  const float4 t01 = *reinterpret_cast<const float4*>(base + ((int64_t)y0 * f.W + x1) * f.Cp);
  const float4 t10 = *reinterpret_cast<const float4*>(base + ((int64_t)y1 * f.W + x0) * f.Cp);
  const float4 t11 = *reinterpret_cast<const float4*>(base + ((int64_t)y1 * f.W + x1) * f.Cp);
  const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
  return make_float4(t00.x * w00 + t01.x * w01 + t10.x * w10 + t11.x * w11, t00.y * w00 + t01.y * w01 + t10.y * w10 + t11.y * w11,
                     t00.z * w00 + t01.z * w01 + t10.z * w10 + t11.z * w11, t00.w * w00 + t01.w * w01 + t10.w * w10 + t11.w * w11);
}

template <int DH>  // DH = Dp / 2: the K slice of one lane half
struct MlpMeanLds {
  static constexpr int DP = 2 * DH;
  static constexpr int LDA = DP + 4;
  static constexpr int LDG = MM_NH + 4;
  float a[MM_NH * LDA];   // folded first layer, rows = hidden units
  float g[32 * LDG];      // folded output map G (rows >= F are zero)
  float l[MM_NH];         // logit row
  float cb[MM_NH];        // b'
  float x[4][32 * LDA];   // per wave: x_v of its 32 voxels
};

template <int DH>
__global__ __launch_bounds__(256, 1) void mlp_mean_pool_kernel(MlpMeanParams p) {
  constexpr int DP = 2 * DH, LDA = DP + 4, LDG = MM_NH + 4;
  __shared__ __attribute__((aligned(16))) MlpMeanLds<DH> S;  // 87 KB (Dp = 32) ... 153 KB (Dp = 128): one workgroup per CU
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const ViewPoolParams& vp = p.vp;
  // ---- stage the folded weights (p.a is [128][DP] in the kernel's padded channel order)
  for (int i = tid; i < MM_NH * (DP / 4); i += 256) {
    const int row = i / (DP / 4), c4 = i - row * (DP / 4);
    *reinterpret_cast<float4*>(S.a + row * LDA + c4 * 4) = *reinterpret_cast<const float4*>(p.a + row * DP + c4 * 4);
  }
  for (int i = tid; i < 32 * (MM_NH / 4); i += 256) {
    const int row = i / (MM_NH / 4), c4 = i - row * (MM_NH / 4);
    *reinterpret_cast<float4*>(S.g + row * LDG + c4 * 4) =
        row < vp.F ? *reinterpret_cast<const float4*>(p.g + row * MM_NH + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int i = tid; i < MM_NH; i += 256) {
    S.l[i] = p.l[i];
    S.cb[i] = p.cb[i];
  }
  __syncthreads();  // from here on the waves are independent workers

  const int R = vp.R;
  const int64_t nvox = (int64_t)R * R * R;
  const int64_t ntiles = (nvox + 31) / 32;
  float* xt = S.x[wave];
  const float step = 2.0f / (float)(R - 1);
  auto lin = [&](int i) { return (i < R / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(R - 1 - i)) * vp.half_extent; };
  const int emb0 = p.emb0;  // first channel of the ray-direction embedding in the padded order

  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t v = tile * 32 + li;
    const int64_t vc = v < nvox ? v : nvox - 1;
    const int x = (int)(vc % R), y = (int)((vc / R) % R), z = (int)(vc / ((int64_t)R * R));
    const float px = lin(x), py = lin(y), pz = lin(z);

    // gathers x_view of the wave's 32 voxels into the LDS tile (both lane halves work: the halves take alternate quads)
    auto gather = [&](int vi) {
      const ViewProj pr = project_view(vp.cams[vi], px, py, pz, vp.proj_eps);
      for (int k = 0; k < vp.n_feats; ++k) {
        const ViewPoolParams::Feat& f = vp.feat[k];
        for (int cq = lh; cq < f.Cp / 4; cq += 2)
          *reinterpret_cast<float4*>(xt + li * LDA + (f.quad0 + cq) * 4) = bilinear_quad(f, vi, cq, pr.ndcx, pr.ndcy);
      }
      if (lh == 0) {  // harmonic embedding of the unit direction: [sin(2^f d_a) | cos(2^f d_a) | d], index a * n + f
        const float d[3] = {pr.dx, pr.dy, pr.dz};
        const int nh = p.n_harmonic;
        float* e = xt + li * LDA + emb0;
        for (int a = 0; a < 3; ++a) {
          float fr = 1.f;
          for (int f = 0; f < nh; ++f) {
            const float arg = d[a] * fr;
            e[a * nh + f] = sinf(arg);
            e[3 * nh + a * nh + f] = cosf(arg);
            fr *= 2.f;
          }
          e[6 * nh + a] = d[a];
        }
        for (int j = emb0 + 6 * nh + 3; j < DP; ++j) xt[li * LDA + j] = 0.f;  // padding columns
      }
    };
    // the lane's K slice of its voxel's x row
    auto load_slice = [&](float (&fv)[DH]) {
      const float4* xp = reinterpret_cast<const float4*>(xt + li * LDA + lh * DH);
#pragma unroll
      for (int q = 0; q < DH / 4; ++q) {
        const float4 t = xp[q];
        fv[4 * q + 0] = t.x;
        fv[4 * q + 1] = t.y;
        fv[4 * q + 2] = t.z;
        fv[4 * q + 3] = t.w;
      }
    };

    // ---- pass 1: mean over the views (weights 1: sum_v x_v / max(V, 1e-2))
    float mean[DH];
#pragma unroll
    for (int k = 0; k < DH; ++k) mean[k] = 0.f;
#pragma unroll 1
    for (int vi = 0; vi < vp.n_views; ++vi) {
      gather(vi);
      HOLO_WAVE_SYNC();
      float fv[DH];
      load_slice(fv);
#pragma unroll
      for (int k = 0; k < DH; ++k) mean[k] += fv[k];
      HOLO_WAVE_SYNC();
    }
    {
      const float den = fmaxf((float)vp.n_views, 1e-2f);
#pragma unroll
      for (int k = 0; k < DH; ++k) mean[k] = mean[k] / den;
    }
    // ---- c = Am mean + b' (A operand = Am rows straight from global memory / L2)
    f32x16 cacc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) cacc[t][r] = S.cb[t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
      const float4* ap = reinterpret_cast<const float4*>(p.am + (int64_t)(t * 32 + li) * DP + lh * DH);
#pragma unroll
      for (int q = 0; q < DH / 4; ++q) {
        const float4 a4 = ap[q];
        cacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, mean[4 * q + 0], cacc[t], 0, 0, 0);
        cacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, mean[4 * q + 1], cacc[t], 0, 0, 0);
        cacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, mean[4 * q + 2], cacc[t], 0, 0, 0);
        cacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, mean[4 * q + 3], cacc[t], 0, 0, 0);
      }
    }
    // ---- pass 2: per view hidden layer, logit, output map; online softmax over the views
    float m_run = -3.0e38f, s_run = 0.f;
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll 1
    for (int vi = 0; vi < vp.n_views; ++vi) {
      gather(vi);
      HOLO_WAVE_SYNC();
      float fv[DH];
      load_slice(fv);
      HOLO_WAVE_SYNC();
      float lg = 0.f;
      f32x16 gacc;
#pragma unroll
      for (int r = 0; r < 16; ++r) gacc[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x16 acc = cacc[t];
        const float4* ap = reinterpret_cast<const float4*>(S.a + (t * 32 + li) * LDA + lh * DH);
#pragma unroll
        for (int q = 0; q < DH / 4; ++q) {
          const float4 a4 = ap[q];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, fv[4 * q + 0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, fv[4 * q + 1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, fv[4 * q + 2], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, fv[4 * q + 3], acc, 0, 0, 0);
        }
        // LeakyReLU; the lane's rows of this tile: t*32 + 8u + 4 lh + e for register 4u + e
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row0 = t * 32 + 8 * u + 4 * lh;
          const float4 l4 = *reinterpret_cast<const float4*>(S.l + row0);
          const float4 g4 = *reinterpret_cast<const float4*>(S.g + li * LDG + row0);  // A operand rows of G h: G[j = li][row]
          float h0 = acc[4 * u + 0], h1 = acc[4 * u + 1], h2 = acc[4 * u + 2], h3 = acc[4 * u + 3];
          h0 = fmaxf(h0, 0.2f * h0);
          h1 = fmaxf(h1, 0.2f * h1);
          h2 = fmaxf(h2, 0.2f * h2);
          h3 = fmaxf(h3, 0.2f * h3);
          lg = fmaf(l4.x, h0, fmaf(l4.y, h1, fmaf(l4.z, h2, fmaf(l4.w, h3, lg))));
          // register 4u+e holds hidden row (row0 + e) in THIS lane half and (row0 + e) -+ 4 in the other: as a B operand it
          // is the k pair {8u + e, 8u + e + 4} (+ t*32) of one MFMA step whose A operand is G[:, own row]
          gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.x, h0, gacc, 0, 0, 0);
          gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.y, h1, gacc, 0, 0, 0);
          gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.z, h2, gacc, 0, 0, 0);
          gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(g4.w, h3, gacc, 0, 0, 0);
        }
      }
      lg += __shfl_xor(lg, 32);
      lg += p.l0;
      const float m_new = fmaxf(m_run, lg);
      const float alpha = __expf(m_run - m_new), pw = __expf(lg - m_new);
      s_run = s_run * alpha + pw;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] = oacc[r] * alpha + pw * gacc[r];
    }
    // ---- out[j][voxel] = tanh(oacc / s + g0[j]); D rows j = (r&3) + 8 (r>>2) + 4 lh, column = voxel li
    if (v < nvox) {
      const float inv = 1.f / s_run;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (j < vp.F) vp.out[(int64_t)j * nvox + v] = tanhf(oacc[r] * inv + p.g0[j]);
      }
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

template <int DH>
static int mlp_mean_launch_t(const MlpMeanParams& p, void* stream, int n_wgs) {
  HOLO_LAUNCH(mlp_mean_pool_kernel<DH>, dim3((unsigned)n_wgs), dim3(256), stream, p);
  return 0;
}

int mlp_mean_pool_launch(const MlpMeanParams& p, int num_cus, void* stream) {
  const ViewPoolParams& vp = p.vp;
  if (vp.n_views < 1 || vp.n_views > ViewPoolParams::MAX_VIEWS || vp.n_feats < 1 || vp.n_feats > ViewPoolParams::MAX_FEATS ||
      vp.R < 2 || vp.F < 1 || vp.F > 32 || p.dp < 8 || p.dp > 128 || (p.dp & 7)) {
    set_error("mlp_mean_pool: 1..%d views, 1..%d feature maps, feature_size <= 32, padded input width 8..128 (got %d)",
              ViewPoolParams::MAX_VIEWS, ViewPoolParams::MAX_FEATS, p.dp);
    return -1;
  }
  const int64_t ntiles = cdiv((int64_t)vp.R * vp.R * vp.R, 32);
  int n_wgs = (int)cdiv(ntiles, 4);
  const int cap = num_cus > 0 ? num_cus : 256;
  if (n_wgs > cap) n_wgs = cap;  // persistent: one 4-wave workgroup per CU (the LDS image is > 100 KB)
  switch (p.dp / 2) {
    case 16: return mlp_mean_launch_t<16>(p, stream, n_wgs);
    case 24: return mlp_mean_launch_t<24>(p, stream, n_wgs);
    case 32: return mlp_mean_launch_t<32>(p, stream, n_wgs);
    case 48: return mlp_mean_launch_t<48>(p, stream, n_wgs);
    case 64: return mlp_mean_launch_t<64>(p, stream, n_wgs);
    default:
      set_error("mlp_mean_pool: padded input width %d is not one of 32, 48, 64, 96, 128", p.dp);
      return -1;
  }
}

}  // namespace holo
