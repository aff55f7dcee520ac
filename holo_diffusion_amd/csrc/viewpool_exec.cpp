// viewpool_exec.cpp — host side of the view-pooling entry (holo_view_pool): argument checks, layout conversion of the
// caller's NCHW feature maps into the workspace, camera centres, launch.
//
// Reference interface replaced (relative to /root/reference/holo_diffusion):
//   HoloDiffusionModel.forward, image_rgb branch      holo_diffusion_model.py:327-374
//   (ViewPooler = ViewSampler + AngleWeightedReductionFeatureAggregator of PyTorch3D 0.7.4, configs/apple.yaml:183-196;
//    _get_point_to_source_camera_ray_dirs custom_modules.py:279-334; pooled_feature_mapper :113,368; tanh :373)
#include <math.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

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

}  // extern "C"

// everything of holo_view_pool up to the launch: argument checks, channels-last copies of the feature maps and the transposed
// mapper weight at the start of the workspace (ws is advanced behind them), cameras
static int setup_view_pool(HoloCtx* ctx, const HoloViewPoolCfg* cfg, const HoloViewFeature* feats, int n_feats,
                           const HoloCamera* cameras, int n_views, const float* mapper_weight, const float* mapper_bias,
                           void* workspace, size_t workspace_bytes, size_t workspace_need, void* stream, ViewPoolParams& p,
                           char*& ws) {
  if (!ctx || !cfg || !feats || !cameras || !mapper_weight || !workspace) {
    set_error("holo_view_pool: null argument");
    return HOLO_E_INVALID;
  }
  if (n_views < 1 || n_views > ViewPoolParams::MAX_VIEWS || n_feats < 1 || n_feats > ViewPoolParams::MAX_FEATS ||
      cfg->resol < 2 || cfg->feature_size < 1) {
    set_error("holo_view_pool: 1..%d source views, 1..%d feature maps", ViewPoolParams::MAX_VIEWS, ViewPoolParams::MAX_FEATS);
    return HOLO_E_UNSUPPORTED;
  }
  if (workspace_bytes < workspace_need) {
    set_error("holo_view_pool: workspace too small");
    return HOLO_E_WORKSPACE;
  }
  memset(&p, 0, sizeof p);
  ws = (char*)workspace;
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
  ws += align256((size_t)p.A * p.F * sizeof(float));
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
  return 0;
}

extern "C" {

int holo_view_pool(HoloCtx* ctx, const HoloViewPoolCfg* cfg, const HoloViewFeature* feats, int n_feats,
                   const HoloCamera* cameras, int n_views, const float* mapper_weight, const float* mapper_bias,
                   float* voxel_features, void* workspace, size_t workspace_bytes, void* stream) {
  if (!voxel_features) {
    set_error("holo_view_pool: null argument");
    return HOLO_E_INVALID;
  }
  ViewPoolParams p;
  char* ws;
  const int rc = setup_view_pool(ctx, cfg, feats, n_feats, cameras, n_views, mapper_weight, mapper_bias, workspace,
                                 workspace_bytes, cfg && feats ? holo_view_pool_workspace_bytes(cfg, feats, n_feats, n_views) : 0,
                                 stream, p, ws);
  if (rc) return rc;
  p.out = voxel_features;
  return view_pool_launch(p, stream) ? HOLO_E_INVALID : 0;
}

// ---- backward (kernels_viewpool_bwd.hip).  Workspace: the forward's part, then the zeroed channels-last gradient maps,
//      then the per-workgroup partials of d weight / d bias.
static int view_pool_bwd_wgs(HoloCtx* ctx, int resol) {
  const int64_t groups = ((int64_t)resol * resol * resol + 15) / 16;
  int64_t n = 4 * (int64_t)ctx->num_cus;  // (view_pool_bwd2_kernel: four resident workgroups per CU)
  return (int)(groups < n ? groups : n);
}

size_t holo_view_pool_backward_workspace_bytes(HoloCtx* ctx, const HoloViewPoolCfg* cfg, const HoloViewFeature* feats,
                                               int n_feats, int n_views) {
  if (!ctx || !cfg || !feats || n_feats < 1 || n_views < 1) return 0;
  size_t b = holo_view_pool_workspace_bytes(cfg, feats, n_feats, n_views);
  int sumC = 0;
  for (int k = 0; k < n_feats; ++k) {
    const size_t Cp = (size_t)((feats[k].channels + 3) / 4 * 4);
    b += align256((size_t)n_views * feats[k].height * feats[k].width * Cp * sizeof(float));
    sumC += feats[k].channels;
  }
  b += align256((size_t)view_pool_bwd_wgs(ctx, cfg->resol) * ((size_t)2 * sumC * cfg->feature_size + cfg->feature_size) * sizeof(float));
  if (ctx->deterministic) {  // 64-bit fixed-point images of the gradient maps + one word per map (holo_ctx_set_deterministic)
    for (int k = 0; k < n_feats; ++k) {
      const size_t Cp = (size_t)((feats[k].channels + 3) / 4 * 4);
      b += align256((size_t)n_views * feats[k].height * feats[k].width * Cp * sizeof(long long));
    }
    b += 256;
  }
  return b + 256;
}

int holo_view_pool_backward(HoloCtx* ctx, const HoloViewPoolCfg* cfg, const HoloViewFeature* feats, int n_feats,
                            const HoloCamera* cameras, int n_views, const float* mapper_weight, const float* mapper_bias,
                            const float* grad_voxel_features, float* const* grad_feats, float* grad_mapper_weight,
                            float* grad_mapper_bias, void* workspace, size_t workspace_bytes, void* stream) {
  if (!grad_voxel_features) {
    set_error("holo_view_pool_backward: null argument");
    return HOLO_E_INVALID;
  }
  ViewPoolBwdParams b;
  memset(&b, 0, sizeof b);
  char* ws;
  int rc = setup_view_pool(ctx, cfg, feats, n_feats, cameras, n_views, mapper_weight, mapper_bias, workspace, workspace_bytes,
                           ctx && cfg && feats ? holo_view_pool_backward_workspace_bytes(ctx, cfg, feats, n_feats, n_views) : 0, stream,
                           b.fwd, ws);
  if (rc) return rc;
  b.gout = grad_voxel_features;
  for (int k = 0; k < n_feats; ++k) {
    const ViewPoolParams::Feat& f = b.fwd.feat[k];
    const size_t bytes = (size_t)n_views * f.H * f.W * f.Cp * sizeof(float);
    if (grad_feats && grad_feats[k]) {
      b.gfeat[k] = (float*)ws;
      b.want_feats = 1;
      if (hipMemsetAsync(ws, 0, bytes, (hipStream_t)stream) != hipSuccess) {
        set_error("holo_view_pool_backward: hipMemsetAsync failed");
        return HOLO_E_HIP;
      }
    }
    ws += align256(bytes);
  }
  const int n_wgs = view_pool_bwd_wgs(ctx, cfg->resol);
  b.partial = (float*)ws;
  ws += align256((size_t)n_wgs * ((size_t)b.fwd.A * b.fwd.F + b.fwd.F) * sizeof(float));
  b.dW = grad_mapper_weight;
  b.dbias = grad_mapper_bias;
  if (ctx->deterministic && b.want_feats) {
    char* w0 = ws;
    for (int k = 0; k < n_feats; ++k) {
      const ViewPoolParams::Feat& f = b.fwd.feat[k];
      if (b.gfeat[k]) b.gfix[k] = (long long*)ws;
      ws += align256((size_t)n_views * f.H * f.W * f.Cp * sizeof(long long));
    }
    b.fix_max = (uint32_t*)ws;
    ws += 256;
    if (hipMemsetAsync(w0, 0, (size_t)(ws - w0), (hipStream_t)stream) != hipSuccess) {
      set_error("holo_view_pool_backward: hipMemsetAsync failed");
      return HOLO_E_HIP;
    }
  }
  if (view_pool_bwd_launch(b, n_wgs, stream)) return HOLO_E_UNSUPPORTED;
  for (int k = 0; k < n_feats; ++k)
    if (b.gfeat[k]) {
      const ViewPoolParams::Feat& f = b.fwd.feat[k];
      if (b.gfix[k] && fix_flush_launch(b.gfix[k], b.fix_max + k, b.gfeat[k], (int64_t)n_views * f.H * f.W * f.Cp, stream))
        return HOLO_E_INVALID;
      if (nhwc_pad_to_nchw_launch(b.gfeat[k], grad_feats[k], n_views, f.C, f.Cp, (int64_t)f.H * f.W, stream)) return HOLO_E_INVALID;
    }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// MLPMeanFeatureAggregator path (custom_modules.py:162-334; configs/hydrant.yaml:184): parameter binding, the float64
// fold of the affine stretches (see kernels_viewpool.hip) and the launch.
// ---------------------------------------------------------------------------------------------------------------------
struct HoloMlpMeanPooler {
  HoloCtx* ctx;
  HoloMlpMeanCfg cfg;
  std::map<std::string, std::vector<float>> host;
  std::map<std::string, std::vector<int64_t>> expected;
  int D = 0, E = 0, dp = 0, emb0 = 0;
  int quad0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float* dev = nullptr;  // a | am | cb | g | g0 | l
  float* stage = nullptr;  // staging buffer of upload_via_stage
  size_t stage_floats = 0;
  float l0 = 0.f;
  bool committed = false;
  std::map<std::string, std::vector<float>> grads;  // holo_mlp_mean_backward: by reference parameter name
};

static int mlp_mean_fill(HoloMlpMeanPooler* h, const HoloViewFeature* feats, int n_feats, const HoloCamera* cameras, int n_views,
                         void* stream, MlpMeanParams& p, char*& ws);

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return HOLO_E_HIP;                                                               \
    }                                                                                  \
  } while (0)

int holo_mlp_mean_create(HoloCtx* ctx, const HoloMlpMeanCfg* cfg, HoloMlpMeanPooler** out) {
  if (!ctx || !cfg || !out) {
    set_error("holo_mlp_mean_create: null argument");
    return HOLO_E_INVALID;
  }
  if (cfg->n_hidden != 128 || cfg->n_layers != 1 || cfg->feature_size < 1 || cfg->feature_size > 32 || cfg->dim_out < 1 ||
      cfg->n_harmonic_functions_ray < 0 || cfg->n_harmonic_functions_ray > 8 || cfg->n_feats < 1 ||
      cfg->n_feats > ViewPoolParams::MAX_FEATS || cfg->resol < 2) {
    set_error("holo_mlp_mean_create: unsupported configuration (n_hidden 128, n_layers 1, feature_size <= 32, "
              "1..%d feature maps, <= 8 harmonic functions)", ViewPoolParams::MAX_FEATS);
    return HOLO_E_UNSUPPORTED;
  }
  HoloMlpMeanPooler* h = new HoloMlpMeanPooler;
  h->ctx = ctx;
  h->cfg = *cfg;
  int quad = 0;
  for (int k = 0; k < cfg->n_feats; ++k) {
    if (cfg->channels[k] < 1) {
      delete h;
      set_error("holo_mlp_mean_create: feature map %d has no channels", k);
      return HOLO_E_INVALID;
    }
    h->quad0[k] = quad;
    quad += (cfg->channels[k] + 3) / 4;
    h->D += cfg->channels[k];
  }
  h->E = 3 * (2 * cfg->n_harmonic_functions_ray + 1);
  h->D += h->E;
  h->emb0 = quad * 4;
  const int dq = h->emb0 + (h->E + 3) / 4 * 4;
  static const int widths[5] = {32, 48, 64, 96, 128};
  for (int w : widths)
    if (!h->dp && dq <= w) h->dp = w;
  if (!h->dp) {
    delete h;
    set_error("holo_mlp_mean_create: %d input channels (padded %d) exceed the kernel's 128", h->D, dq);
    return HOLO_E_UNSUPPORTED;
  }
  const int64_t nh = cfg->n_hidden, D = h->D, dout = cfg->dim_out, F = cfg->feature_size;
  h->expected["_first_sampled.weight"] = {nh, D};
  h->expected["_first_sampled.bias"] = {nh};
  h->expected["_first_mean.weight"] = {nh, D};
  h->expected["_first_mean.bias"] = {nh};
  h->expected["_mlp.mlp.0.0.weight"] = {nh, nh};
  h->expected["_mlp.mlp.0.0.bias"] = {nh};
  h->expected["_last.weight"] = {dout, nh};
  h->expected["_last.bias"] = {dout};
  h->expected["pooled_feature_mapper.weight"] = {F, dout};
  h->expected["pooled_feature_mapper.bias"] = {F};
  const size_t n = (size_t)(2 * nh * h->dp + nh + F * nh + F + nh + 64);
  if (hipMalloc((void**)&h->dev, n * sizeof(float)) != hipSuccess) {
    delete h;
    set_error("holo_mlp_mean_create: hipMalloc failed");
    return HOLO_E_HIP;
  }
  *out = h;
  return 0;
}

int holo_mlp_mean_destroy(HoloMlpMeanPooler* h) {
  if (!h) return 0;
  if (h->dev) (void)hipFree(h->dev);
  if (h->stage) (void)hipFree(h->stage);
  delete h;
  return 0;
}

int holo_mlp_mean_set_param(HoloMlpMeanPooler* h, const char* name, const void* dev_ptr, int ndim, const int64_t* shape,
                            void* stream) {
  if (!h || !name || !dev_ptr || !shape) {
    set_error("holo_mlp_mean_set_param: null argument");
    return HOLO_E_INVALID;
  }
  auto it = h->expected.find(name);
  if (it == h->expected.end()) {
    set_error("holo_mlp_mean_set_param: unknown parameter '%s'", name);
    return HOLO_E_INVALID;
  }
  bool ok = ndim == (int)it->second.size();
  int64_t numel = 1;
  for (int i = 0; ok && i < ndim; ++i) {
    ok = shape[i] == it->second[i];
    numel *= shape[i];
  }
  if (!ok) {
    set_error("holo_mlp_mean_set_param: shape mismatch for '%s'", name);
    return HOLO_E_INVALID;
  }
  std::vector<float>& v = h->host[name];
  v.resize((size_t)numel);
  HIP_TRY(hipMemcpyAsync(v.data(), dev_ptr, (size_t)numel * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  h->committed = false;
  return 0;
}

int holo_mlp_mean_commit(HoloMlpMeanPooler* h, void* stream) {
  if (!h) {
    set_error("holo_mlp_mean_commit: null");
    return HOLO_E_INVALID;
  }
  for (auto& kv : h->expected)
    if (!h->host.count(kv.first)) {
      set_error("holo_mlp_mean_commit: parameter '%s' has not been set", kv.first.c_str());
      return HOLO_E_STATE;
    }
  const int nh = h->cfg.n_hidden, D = h->D, dp = h->dp, dout = h->cfg.dim_out, F = h->cfg.feature_size;
  auto W = [&](const char* n) -> const std::vector<float>& { return h->host[n]; };
  const auto &Ws = W("_first_sampled.weight"), &bs = W("_first_sampled.bias"), &Wm = W("_first_mean.weight"),
             &bm = W("_first_mean.bias"), &W1 = W("_mlp.mlp.0.0.weight"), &b1 = W("_mlp.mlp.0.0.bias"),
             &Wl = W("_last.weight"), &bl = W("_last.bias"), &M = W("pooled_feature_mapper.weight"),
             &mb = W("pooled_feature_mapper.bias");
  // column of reference channel c (torch.cat order: maps in dict order, then the embedding) in the padded order
  std::vector<int> col(D);
  {
    int c = 0;
    for (int k = 0; k < h->cfg.n_feats; ++k)
      for (int j = 0; j < h->cfg.channels[k]; ++j) col[c++] = h->quad0[k] * 4 + j;
    for (int j = 0; j < h->E; ++j) col[c++] = h->emb0 + j;
  }
  std::vector<float> pk((size_t)(2 * nh * dp + nh + F * nh + F + nh), 0.f);
  float* a = pk.data();
  float* am = a + (size_t)nh * dp;
  float* cb = am + (size_t)nh * dp;
  float* g = cb + nh;
  float* g0 = g + (size_t)F * nh;
  float* l = g0 + F;
  for (int i = 0; i < nh; ++i) {
    std::vector<double> ra(D, 0.0), rm(D, 0.0);
    double c = b1[i];
    for (int k = 0; k < nh; ++k) {
      const double w = W1[(size_t)i * nh + k];
      c += w * ((double)bs[k] + (double)bm[k]);
      for (int j = 0; j < D; ++j) {
        ra[j] += w * Ws[(size_t)k * D + j];
        rm[j] += w * Wm[(size_t)k * D + j];
      }
    }
    for (int j = 0; j < D; ++j) {
      a[(size_t)i * dp + col[j]] = (float)ra[j];
      am[(size_t)i * dp + col[j]] = (float)rm[j];
    }
    cb[i] = (float)c;
    l[i] = Wl[i];  // row 0 of _last
  }
  for (int f = 0; f < F; ++f) {
    double c = mb[f];
    std::vector<double> rg(nh, 0.0);
    for (int o = 0; o < dout; ++o) {
      const double w = M[(size_t)f * dout + o];
      c += w * bl[o];
      for (int k = 0; k < nh; ++k) rg[k] += w * Wl[(size_t)o * nh + k];
    }
    for (int k = 0; k < nh; ++k) g[(size_t)f * nh + k] = (float)rg[k];
    g0[f] = (float)c;
  }
  h->l0 = bl[0];
  if (upload_via_stage(&h->stage, &h->stage_floats, h->dev, pk.data(), pk.size(), stream)) {
    set_error("holo_mlp_mean_commit: upload of the folded weights failed");
    return HOLO_E_HIP;
  }
  h->committed = true;
  return 0;
}

size_t holo_mlp_mean_workspace_bytes(const HoloMlpMeanPooler* h, const HoloViewFeature* feats, int n_feats, int n_views) {
  if (!h || !feats || n_feats < 1 || n_views < 1) return 0;
  size_t b = 0;
  for (int k = 0; k < n_feats; ++k) {
    const size_t Cp = (size_t)((feats[k].channels + 3) / 4 * 4);
    b += align256((size_t)n_views * feats[k].height * feats[k].width * Cp * sizeof(float));
  }
  return b + 256;
}

int holo_mlp_mean_pool(HoloMlpMeanPooler* h, const HoloViewFeature* feats, int n_feats, const HoloCamera* cameras,
                       int n_views, float* voxel_features, void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !feats || !cameras || !voxel_features || !workspace) {
    set_error("holo_mlp_mean_pool: null argument");
    return HOLO_E_INVALID;
  }
  if (!h->committed) {
    set_error("holo_mlp_mean_pool: call holo_mlp_mean_commit after setting the parameters");
    return HOLO_E_STATE;
  }
  if (n_feats != h->cfg.n_feats || n_views < 1 || n_views > ViewPoolParams::MAX_VIEWS) {
    set_error("holo_mlp_mean_pool: %d feature maps (created for %d), 1..%d source views", n_feats, h->cfg.n_feats,
              ViewPoolParams::MAX_VIEWS);
    return HOLO_E_INVALID;
  }
  if (workspace_bytes < holo_mlp_mean_workspace_bytes(h, feats, n_feats, n_views)) {
    set_error("holo_mlp_mean_pool: workspace too small");
    return HOLO_E_WORKSPACE;
  }
  MlpMeanParams p;
  char* ws = (char*)workspace;
  const int rc = mlp_mean_fill(h, feats, n_feats, cameras, n_views, stream, p, ws);
  if (rc) return rc;
  p.vp.out = voxel_features;
  return mlp_mean_pool_launch(p, h->ctx->num_cus, stream) ? HOLO_E_INVALID : 0;
}

}  // extern "C"

// the forward's launch parameters: channels-last copies of the maps at ws (advanced), cameras, folded weights
static int mlp_mean_fill(HoloMlpMeanPooler* h, const HoloViewFeature* feats, int n_feats, const HoloCamera* cameras, int n_views,
                         void* stream, MlpMeanParams& p, char*& ws) {
  memset(&p, 0, sizeof p);
  for (int k = 0; k < n_feats; ++k) {
    const HoloViewFeature& f = feats[k];
    if (!f.feats || f.channels != h->cfg.channels[k] || f.height < 1 || f.width < 1) {
      set_error("holo_mlp_mean_pool: feature map %d must have %d channels", k, h->cfg.channels[k]);
      return HOLO_E_INVALID;
    }
    ViewPoolParams::Feat& o = p.vp.feat[k];
    o.C = f.channels;
    o.Cp = (f.channels + 3) / 4 * 4;
    o.H = f.height;
    o.W = f.width;
    o.quad0 = h->quad0[k];
    o.data = (const float*)ws;
    if (nchw_to_nhwc_pad_launch(f.feats, (float*)ws, n_views, o.C, o.Cp, (int64_t)o.H * o.W, stream)) return HOLO_E_INVALID;
    ws += align256((size_t)n_views * o.H * o.W * o.Cp * sizeof(float));
  }
  p.vp.n_feats = n_feats;
  p.vp.n_views = n_views;
  for (int v = 0; v < n_views; ++v) {
    const HoloCamera& c = cameras[v];
    ViewPoolParams::Cam& o = p.vp.cams[v];
    for (int k = 0; k < 9; ++k) o.Rm[k] = c.R[k];
    for (int k = 0; k < 3; ++k) o.T[k] = c.T[k];
    for (int k = 0; k < 2; ++k) {
      o.focal[k] = c.focal[k];
      o.pp[k] = c.principal_point[k];
    }
    for (int j = 0; j < 3; ++j)  // camera centre C = -T R^T (custom_modules.py:304-312)
      o.centre[j] = -(c.T[0] * c.R[j * 3 + 0] + c.T[1] * c.R[j * 3 + 1] + c.T[2] * c.R[j * 3 + 2]);
  }
  p.vp.R = h->cfg.resol;
  p.vp.half_extent = 0.5f * (float)(h->cfg.resol - 1) * (h->cfg.volume_extent / (float)h->cfg.resol);
  p.vp.proj_eps = h->cfg.projection_eps;
  p.vp.F = h->cfg.feature_size;
  const int nh = h->cfg.n_hidden, F = h->cfg.feature_size;
  p.a = h->dev;
  p.am = p.a + (size_t)nh * h->dp;
  p.cb = p.am + (size_t)nh * h->dp;
  p.g = p.cb + nh;
  p.g0 = p.g + (size_t)F * nh;
  p.l = p.g0 + F;
  p.l0 = h->l0;
  p.dp = h->dp;
  p.emb0 = h->emb0;
  p.n_harmonic = h->cfg.n_harmonic_functions_ray;
  return 0;
}

// ---- backward (kernels_viewpool_bwd.hip: the MLPMean section).  Workspace layout, shared by the size query and the run.
struct MmBwdLayout {
  int64_t P, NR, NRp, Pp;  // per CHUNK of voxels
  int64_t Pall;
  int nchunks;
  int S, S2, FW, dp;
  size_t maps, gmaps, X, MEAN, CM, PRE, H, U, DUL, DULT, DPRET, DC, DCT, DX, DCA, part, fold, gfix, fixmax, total;
  bool fixed;  // deterministic mode (holo_ctx_set_deterministic): fixed-point images of the gradient maps
};
static MmBwdLayout mm_bwd_layout(const HoloMlpMeanPooler* h, const HoloViewFeature* feats, int n_feats, int n_views) {
  MmBwdLayout L;
  memset(&L, 0, sizeof L);
  const int R = h->cfg.resol, F = h->cfg.feature_size;
  // the row buffers hold one CHUNK of voxels (all chunks the same size: the padding of the split-K operands stays valid):
  // the largest power-of-two fraction of the grid whose rows fit a 4 GiB budget (~2.5 KB per (voxel, view): 64^3 x 4 views
  // runs in one pass - 3.5 GB, 14.0 ms; chunked it measured 16.8 ms in 4 passes, 17.2 ms in 8 -, 64^3 x 16 views in four
  // passes of 65 536 voxels instead of 14 GB).  HOLO_MLP_MEAN_BWD_CHUNK=<voxels>: development / test knob
  L.Pall = (int64_t)R * R * R;
  L.nchunks = 1;
  {
    const int64_t row_bytes = (int64_t)(2 * h->dp + 3 * 128 + 3 * ((F + 1 + 3) / 4 * 4)) * 4 * n_views;
    int64_t target = ((int64_t)4 << 30) / (row_bytes > 0 ? row_bytes : 1);
#ifndef HOLO_EMU
    const char* e = getenv("HOLO_MLP_MEAN_BWD_CHUNK");
    if (e && atoll(e) > 0) target = atoll(e);
#else
    target = 2048;  // (the emulation's small grids: two chunks at 16^3)
#endif
    while (L.Pall / L.nchunks > target && (L.Pall % (2 * L.nchunks)) == 0) L.nchunks *= 2;
  }
  L.P = L.Pall / L.nchunks;
  L.NR = L.P * n_views;
  L.dp = h->dp;
  L.FW = (F + 1 + 3) / 4 * 4;
  auto splits = [](int64_t rows, int64_t& padded) {
    int64_t S = rows / 1024;
    S = S < 1 ? 1 : (S > 128 ? 128 : S);
    const int64_t kc = ((rows + S - 1) / S + 31) / 32 * 32;
    padded = S * kc;
    return (int)S;
  };
  L.S = splits(L.NR, L.NRp);
  L.S2 = splits(L.P, L.Pp);
  size_t off = 0;
  auto take = [&](size_t floats) {
    const size_t o = off;
    off += align256(floats * sizeof(float));
    return o;
  };
  size_t mapf = 0;
  for (int k = 0; k < n_feats; ++k)
    mapf += align256((size_t)n_views * feats[k].height * feats[k].width * ((feats[k].channels + 3) / 4 * 4) * sizeof(float)) / sizeof(float);
  L.maps = take(mapf);
  L.gmaps = take(mapf);
  L.X = take((size_t)L.NRp * L.dp);
  L.MEAN = take((size_t)L.Pp * L.dp);
  L.CM = take((size_t)L.P * 128);
  L.PRE = take((size_t)L.NR * 128);
  L.H = take((size_t)L.NRp * 128);
  L.U = take((size_t)L.NR * L.FW);
  L.DUL = take((size_t)L.NR * L.FW);
  L.DULT = take((size_t)L.FW * L.NRp);
  L.DPRET = take((size_t)128 * L.NRp);
  L.DC = take((size_t)L.P * 128);
  L.DCT = take((size_t)128 * L.Pp);
  L.DX = take((size_t)L.NR * L.dp);
  L.DCA = take((size_t)L.P * L.dp);
  const size_t pa = (size_t)L.S * 128 * L.dp, pam = (size_t)L.S2 * 128 * L.dp, pg = (size_t)L.S * L.FW * 128;
  const size_t pcol = (size_t)256 * 256;
  L.part = take(pa > pam ? (pa > pg ? (pa > pcol ? pa : pcol) : (pg > pcol ? pg : pcol)) : (pam > pg ? (pam > pcol ? pam : pcol) : (pg > pcol ? pg : pcol)));
  L.fold = take((size_t)2 * 128 * L.dp + 128 + (size_t)L.FW * 128 + L.FW);  // dA | dAm | dcb | dGext | dg0 dl0
  L.fixed = h->ctx && h->ctx->deterministic;
  L.gfix = L.fixed ? take(2 * mapf) : off;
  L.fixmax = L.fixed ? take(64) : off;
  L.total = off + 256;
  return L;
}

static int mm_gemm(const float* A, int lda, const float* B, int ldb, int b_kmajor, float* C, int ldc, int M, int N, int K, int nb,
                   int64_t sa, int64_t sb, int64_t sc, void* stream) {
  GemmParams g;
  memset(&g, 0, sizeof g);
  g.A = A, g.B = B, g.C = C;
  g.M = M, g.Nn = N, g.K = K;
  g.lda = lda, g.ldb = ldb, g.ldc = ldc;
  g.nb0 = nb, g.nb1 = 1;
  g.sa0 = sa, g.sb0 = sb, g.sc0 = sc;
  g.b_kmajor = b_kmajor;
  g.alpha = 1.f;
  return gemm_launch(g, stream);
}

extern "C" {

size_t holo_mlp_mean_backward_workspace_bytes(const HoloMlpMeanPooler* h, const HoloViewFeature* feats, int n_feats, int n_views) {
  if (!h || !feats || n_feats < 1 || n_views < 1) return 0;
  return mm_bwd_layout(h, feats, n_feats, n_views).total;
}

int holo_mlp_mean_backward(HoloMlpMeanPooler* h, const HoloViewFeature* feats, int n_feats, const HoloCamera* cameras, int n_views,
                           const float* grad_voxel_features, float* const* grad_feats, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (!h || !feats || !cameras || !grad_voxel_features || !workspace) {
    set_error("holo_mlp_mean_backward: null argument");
    return HOLO_E_INVALID;
  }
  if (!h->committed) {
    set_error("holo_mlp_mean_backward: call holo_mlp_mean_commit after setting the parameters");
    return HOLO_E_STATE;
  }
  if (n_feats != h->cfg.n_feats || n_views < 1 || n_views > ViewPoolParams::MAX_VIEWS || (h->cfg.feature_size & 3)) {
    set_error("holo_mlp_mean_backward: %d feature maps (created for %d), 1..%d source views, feature_size a multiple of 4", n_feats,
              h->cfg.n_feats, ViewPoolParams::MAX_VIEWS);
    return HOLO_E_INVALID;
  }
  const MmBwdLayout L = mm_bwd_layout(h, feats, n_feats, n_views);
  if (workspace_bytes < L.total) {
    set_error("holo_mlp_mean_backward: workspace too small");
    return HOLO_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace;
  // zero everything behind the forward's maps: the gradient maps, the padding rows of the split-K operands
  HIP_TRY(hipMemsetAsync(base + L.gmaps, 0, L.total - 256 - L.gmaps, st));
  MlpMeanBwdParams b;
  memset(&b, 0, sizeof b);
  char* ws = base + L.maps;
  int rc = mlp_mean_fill(h, feats, n_feats, cameras, n_views, stream, b.fwd, ws);
  if (rc) return rc;
  const int F = h->cfg.feature_size, dp = L.dp, FW = L.FW;
  b.gout = grad_voxel_features;
  b.FW = FW;
  b.NRp = L.NRp;
  b.Pp = L.Pp;
  auto fp = [&](size_t off) { return (float*)(base + off); };
  b.X = fp(L.X), b.MEAN = fp(L.MEAN), b.CM = fp(L.CM), b.PRE = fp(L.PRE), b.H = fp(L.H), b.U = fp(L.U), b.DUL = fp(L.DUL);
  b.DULT = fp(L.DULT), b.DPRET = fp(L.DPRET), b.DC = fp(L.DC), b.DCT = fp(L.DCT), b.DX = fp(L.DX), b.DCA = fp(L.DCA);
  {
    char* g = base + L.gmaps;
    for (int k = 0; k < n_feats; ++k) {
      const ViewPoolParams::Feat& f = b.fwd.vp.feat[k];
      if (grad_feats && grad_feats[k]) b.gfeat[k] = (float*)g;
      g += align256((size_t)n_views * f.H * f.W * f.Cp * sizeof(float));
    }
    if (L.fixed) {
      char* gx = base + L.gfix;
      for (int k = 0; k < n_feats; ++k) {
        const ViewPoolParams::Feat& f = b.fwd.vp.feat[k];
        if (b.gfeat[k]) b.gfix[k] = (long long*)gx;
        gx += 2 * align256((size_t)n_views * f.H * f.W * f.Cp * sizeof(float));
      }
      b.fix_max = (uint32_t*)(base + L.fixmax);
    }
  }
  float* part = fp(L.part);
  float* fold = fp(L.fold);
  float *dA = fold, *dAm = dA + (size_t)128 * dp, *dcb = dAm + (size_t)128 * dp, *dG = dcb + 128, *dgl = dG + (size_t)FW * 128;
  const int NR = (int)L.NR, P = (int)L.P;
  const int Kc = (int)(L.NRp / L.S), Kc2 = (int)(L.Pp / L.S2);
#define MM_TRY(x)                      \
  do {                                 \
    if (x) return HOLO_E_INVALID;      \
  } while (0)
  bool want = false;
  for (int k = 0; k < n_feats; ++k) want |= b.gfeat[k] != nullptr;
  b.Pc = L.P;
  b.Pall = L.Pall;
  for (int ck = 0; ck < L.nchunks; ++ck) {
    b.p0 = (int64_t)ck * L.P;
    const int acc = ck > 0 ? 1 : 0;  // the parameter gradients of the chunks add up in chunk order
    // forward recomputation
    MM_TRY(mm_bwd_step_launch(b, 0, stream));                                                                    // X, MEAN
    MM_TRY(mm_gemm(b.MEAN, dp, b.fwd.am, dp, 0, b.CM, 128, P, 128, dp, 1, 0, 0, 0, stream));                     // CM = MEAN Am^T
    MM_TRY(mm_gemm(b.X, dp, b.fwd.a, dp, 0, b.PRE, 128, NR, 128, dp, 1, 0, 0, 0, stream));                       // PRE = X A^T
    MM_TRY(mm_bwd_step_launch(b, 1, stream));                                                                    // + CM + b', H
    MM_TRY(mm_gemm(b.H, 128, b.fwd.g, 128, 0, b.U, FW, NR, F, 128, 1, 0, 0, 0, stream));                         // U = H G^T
    MM_TRY(mm_bwd_step_launch(b, 2, stream));                                                                    // DUL, DULT
    // dG (rows 0..F-1) and dl (row F) = DULT H, split over the rows
    MM_TRY(mm_gemm(b.DULT, (int)L.NRp, b.H, 128, 1, part, 128, FW, 128, Kc, L.S, Kc, (int64_t)Kc * 128, (int64_t)FW * 128, stream));
    MM_TRY(mm_sum_partials_launch(part, L.S, (int64_t)FW * 128, dG, stream, acc));
    MM_TRY(mm_colsum_launch(b.DUL, L.NR, FW, FW, part, 1024, dgl, stream, acc));                                 // dg0 | dl0
    MM_TRY(mm_gemm(b.DUL, FW, b.fwd.g, 128, 1, b.H, 128, NR, 128, F, 1, 0, 0, 0, stream));                       // DH = DU G  (into H)
    MM_TRY(mm_bwd_step_launch(b, 3, stream));                                                                    // DPRE (in PRE), DPRET, DC, DCT
    MM_TRY(mm_gemm(b.DPRET, (int)L.NRp, b.X, dp, 1, part, dp, 128, dp, Kc, L.S, Kc, (int64_t)Kc * dp, (int64_t)128 * dp, stream));
    MM_TRY(mm_sum_partials_launch(part, L.S, (int64_t)128 * dp, dA, stream, acc));
    MM_TRY(mm_gemm(b.DCT, (int)L.Pp, b.MEAN, dp, 1, part, dp, 128, dp, Kc2, L.S2, Kc2, (int64_t)Kc2 * dp, (int64_t)128 * dp, stream));
    MM_TRY(mm_sum_partials_launch(part, L.S2, (int64_t)128 * dp, dAm, stream, acc));
    MM_TRY(mm_colsum_launch(b.DC, L.P, 128, 128, part, 256, dcb, stream, acc));
    if (want) {
      MM_TRY(mm_gemm(b.PRE, 128, b.fwd.a, dp, 1, b.DX, dp, NR, dp, 128, 1, 0, 0, 0, stream));                    // DX = DPRE A
      MM_TRY(mm_gemm(b.DC, 128, b.fwd.am, dp, 1, b.DCA, dp, P, dp, 128, 1, 0, 0, 0, stream));                    // DCA = DC Am
      if (L.fixed && ck > 0) HIP_TRY(hipMemsetAsync(b.fix_max, 0, 64, st));  // every chunk has its own binary point
      MM_TRY(mm_bwd_step_launch(b, 4, stream));
      if (L.fixed)  // the chunk's sums are added to the gradient maps in chunk order
        for (int k = 0; k < n_feats; ++k)
          if (b.gfix[k]) {
            const ViewPoolParams::Feat& f = b.fwd.vp.feat[k];
            MM_TRY(fix_flush_launch(b.gfix[k], b.fix_max + k, b.gfeat[k], (int64_t)n_views * f.H * f.W * f.Cp, stream));
          }
    }
  }
  if (want) {
    for (int k = 0; k < n_feats; ++k)
      if (b.gfeat[k]) {
        const ViewPoolParams::Feat& f = b.fwd.vp.feat[k];
        MM_TRY(nhwc_pad_to_nchw_launch(b.gfeat[k], grad_feats[k], n_views, f.C, f.Cp, (int64_t)f.H * f.W, stream));
      }
  }
#undef MM_TRY
  // ---- the folded gradients come to the host and are un-folded in float64 (the chain rule of holo_mlp_mean_commit's fold)
  const size_t nfold = (size_t)2 * 128 * dp + 128 + (size_t)FW * 128 + FW;
  std::vector<float> hf(nfold);
  HIP_TRY(hipMemcpyAsync(hf.data(), fold, nfold * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const int nh = 128, D = h->D, dout = h->cfg.dim_out;
  const float *hA = hf.data(), *hAm = hA + (size_t)128 * dp, *hcb = hAm + (size_t)128 * dp, *hG = hcb + 128, *hgl = hG + (size_t)FW * 128;
  auto W = [&](const char* n) -> const std::vector<float>& { return h->host[n]; };
  const auto &Ws = W("_first_sampled.weight"), &bs = W("_first_sampled.bias"), &Wm = W("_first_mean.weight"),
             &bm = W("_first_mean.bias"), &W1 = W("_mlp.mlp.0.0.weight"), &Wl = W("_last.weight"), &bl = W("_last.bias"),
             &M = W("pooled_feature_mapper.weight");
  std::vector<int> col(D);
  {
    int c = 0;
    for (int k = 0; k < h->cfg.n_feats; ++k)
      for (int j = 0; j < h->cfg.channels[k]; ++j) col[c++] = h->quad0[k] * 4 + j;
    for (int j = 0; j < h->E; ++j) col[c++] = h->emb0 + j;
  }
  std::vector<double> gW1((size_t)nh * nh, 0.0), gWs((size_t)nh * D, 0.0), gWm((size_t)nh * D, 0.0), gbsm(nh, 0.0), gb1(nh, 0.0);
  for (int i = 0; i < nh; ++i) {  // A = W1 Ws, Am = W1 Wm, b' = W1 (bs + bm) + b1
    gb1[i] = hcb[i];
    for (int k = 0; k < nh; ++k) {
      double acc = (double)hcb[i] * ((double)bs[k] + (double)bm[k]);
      const double w = W1[(size_t)i * nh + k];
      for (int j = 0; j < D; ++j) {
        const double da = hA[(size_t)i * dp + col[j]], dam = hAm[(size_t)i * dp + col[j]];
        acc += da * Ws[(size_t)k * D + j] + dam * Wm[(size_t)k * D + j];
        gWs[(size_t)k * D + j] += w * da;
        gWm[(size_t)k * D + j] += w * dam;
      }
      gW1[(size_t)i * nh + k] = acc;
      gbsm[k] += w * hcb[i];
    }
  }
  std::vector<double> gM((size_t)F * dout, 0.0), gWl((size_t)dout * nh, 0.0), gbl(dout, 0.0), gmb(F, 0.0);
  for (int f = 0; f < F; ++f) {  // G = M Wl, g0 = M bl + mapper bias
    gmb[f] = hgl[f];
    for (int o = 0; o < dout; ++o) {
      double acc = (double)hgl[f] * bl[o];
      const double w = M[(size_t)f * dout + o];
      for (int k = 0; k < nh; ++k) {
        const double dg = hG[(size_t)f * 128 + k];
        acc += dg * Wl[(size_t)o * nh + k];
        gWl[(size_t)o * nh + k] += w * dg;
      }
      gM[(size_t)f * dout + o] = acc;
      gbl[o] += w * hgl[f];
    }
  }
  for (int k = 0; k < nh; ++k) gWl[k] += hG[(size_t)F * 128 + k];  // l = Wl[0]
  gbl[0] += hgl[F];                                                  // l0 = bl[0]
  auto put = [&](const char* name, const std::vector<double>& v) {
    std::vector<float>& o = h->grads[name];
    o.resize(v.size());
    for (size_t i = 0; i < v.size(); ++i) o[i] = (float)v[i];
  };
  put("_first_sampled.weight", gWs);
  put("_first_sampled.bias", gbsm);
  put("_first_mean.weight", gWm);
  put("_first_mean.bias", gbsm);
  put("_mlp.mlp.0.0.weight", gW1);
  put("_mlp.mlp.0.0.bias", gb1);
  put("_last.weight", gWl);
  put("_last.bias", gbl);
  put("pooled_feature_mapper.weight", gM);
  put("pooled_feature_mapper.bias", gmb);
  return 0;
}

int holo_mlp_mean_get_grad(HoloMlpMeanPooler* h, const char* name, float* out_dev, int64_t numel, void* stream) {
  if (!h || !name || !out_dev) {
    set_error("holo_mlp_mean_get_grad: null argument");
    return HOLO_E_INVALID;
  }
  auto it = h->grads.find(name);
  if (it == h->grads.end()) {
    set_error("holo_mlp_mean_get_grad: no gradient for '%s' (run holo_mlp_mean_backward first)", name);
    return HOLO_E_STATE;
  }
  if ((int64_t)it->second.size() != numel) {
    set_error("holo_mlp_mean_get_grad: '%s' has %lld elements, not %lld", name, (long long)it->second.size(), (long long)numel);
    return HOLO_E_INVALID;
  }
  if (upload_via_stage(&h->stage, &h->stage_floats, out_dev, it->second.data(), it->second.size(), stream)) {
    set_error("holo_mlp_mean_get_grad: upload failed");
    return HOLO_E_HIP;
  }
  return 0;
}

}  // extern "C"
