// viewpool_exec.cpp — host side of the view-pooling entry (holo_view_pool): argument checks, layout conversion of the
// caller's NCHW feature maps into the workspace, camera centres, launch.
//
// Reference interface replaced (relative to /root/reference/holo_diffusion):
//   HoloDiffusionModel.forward, image_rgb branch      holo_diffusion_model.py:327-374
//   (ViewPooler = ViewSampler + AngleWeightedReductionFeatureAggregator of PyTorch3D 0.7.4, configs/apple.yaml:183-196;
//    _get_point_to_source_camera_ray_dirs custom_modules.py:279-334; pooled_feature_mapper :113,368; tanh :373)
#include <math.h>
#include <string.h>

#include "../../include/holo_abi.h"
#include "holo_common.h"
#include "holo_kernels.h"

using namespace holo;

static size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" {

size_t holo_view_pool_workspace_bytes(const HoloViewPoolCfg* cfg, const HoloViewFeature* feats, int n_feats, int n_views) {
  if (!cfg || !feats || n_feats < 1 || n_views < 1) return 0;
  size_t b = 0;
  int sumC = 0;
  for (int k = 0; k < n_feats; ++k) {
    const size_t Cp = (size_t)((feats[k].channels + 3) / 4 * 4);
    b += align256((size_t)n_views * feats[k].height * feats[k].width * Cp * sizeof(float));
    sumC += feats[k].channels;
  }
  b += align256((size_t)2 * sumC * cfg->feature_size * sizeof(float));  // transposed mapper weight
  return b + 256;
}

int holo_view_pool(HoloCtx* ctx, const HoloViewPoolCfg* cfg, const HoloViewFeature* feats, int n_feats,
                   const HoloCamera* cameras, int n_views, const float* mapper_weight, const float* mapper_bias,
                   float* voxel_features, void* workspace, size_t workspace_bytes, void* stream) {
  if (!ctx || !cfg || !feats || !cameras || !mapper_weight || !voxel_features || !workspace) {
    set_error("holo_view_pool: null argument");
    return HOLO_E_INVALID;
  }
  if (n_views < 1 || n_views > ViewPoolParams::MAX_VIEWS || n_feats < 1 || n_feats > ViewPoolParams::MAX_FEATS ||
      cfg->resol < 2 || cfg->feature_size < 1) {
    set_error("holo_view_pool: 1..%d source views, 1..%d feature maps", ViewPoolParams::MAX_VIEWS, ViewPoolParams::MAX_FEATS);
    return HOLO_E_UNSUPPORTED;
  }
  if (workspace_bytes < holo_view_pool_workspace_bytes(cfg, feats, n_feats, n_views)) {
    set_error("holo_view_pool: workspace too small");
    return HOLO_E_WORKSPACE;
  }
  ViewPoolParams p;
  memset(&p, 0, sizeof p);
  char* ws = (char*)workspace;
  int quad = 0, outc = 0;
  for (int k = 0; k < n_feats; ++k) {
    const HoloViewFeature& f = feats[k];
    if (!f.feats || f.channels < 1 || f.height < 1 || f.width < 1) {
      set_error("holo_view_pool: feature map %d is empty", k);
      return HOLO_E_INVALID;
    }
    ViewPoolParams::Feat& o = p.feat[k];
    o.C = f.channels;
    o.Cp = (f.channels + 3) / 4 * 4;
    o.H = f.height;
    o.W = f.width;
    o.quad0 = quad;
    o.out0 = outc;
    o.data = (const float*)ws;
    if (nchw_to_nhwc_pad_launch(f.feats, (float*)ws, n_views, o.C, o.Cp, (int64_t)o.H * o.W, stream)) return HOLO_E_INVALID;
    ws += align256((size_t)n_views * o.H * o.W * o.Cp * sizeof(float));
    quad += o.Cp / 4;
    outc += 2 * o.C;
  }
  if (outc > ViewPoolParams::MAX_AGG) {
    set_error("holo_view_pool: %d aggregated features (at most %d)", outc, ViewPoolParams::MAX_AGG);
    return HOLO_E_UNSUPPORTED;
  }
  p.n_feats = n_feats;
  p.n_quads = quad;
  p.A = outc;
  p.F = cfg->feature_size;
  if (transpose_small_launch(mapper_weight, (float*)ws, p.F, p.A, stream)) return HOLO_E_INVALID;  // (F, A) -> (A, F)
  p.wt = (const float*)ws;
  p.bias = mapper_bias;
  p.n_views = n_views;
  for (int v = 0; v < n_views; ++v) {
    const HoloCamera& c = cameras[v];
    ViewPoolParams::Cam& o = p.cams[v];
    for (int k = 0; k < 9; ++k) o.Rm[k] = c.R[k];
    for (int k = 0; k < 3; ++k) o.T[k] = c.T[k];
    for (int k = 0; k < 2; ++k) {
      o.focal[k] = c.focal[k];
      o.pp[k] = c.principal_point[k];
    }
    for (int j = 0; j < 3; ++j)  // camera centre C = -T R^T (custom_modules.py:288-296)
      o.centre[j] = -(c.T[0] * c.R[j * 3 + 0] + c.T[1] * c.R[j * 3 + 1] + c.T[2] * c.R[j * 3 + 2]);
  }
  p.R = cfg->resol;
  p.half_extent = 0.5f * (float)(cfg->resol - 1) * (cfg->volume_extent / (float)cfg->resol);
  p.gamma = cfg->weight_by_ray_angle_gamma;
  p.min_weight = cfg->min_ray_angle_weight;
  p.proj_eps = cfg->projection_eps;
  p.out = voxel_features;
  return view_pool_launch(p, stream) ? HOLO_E_INVALID : 0;
}

}  // extern "C"
