// render_exec.cpp — host side of the fused renderer: RenderMLP parameter binding, the float64 fold of
// the activation-free density net, per-camera depth bounds and the frame launch.
//
// Reference interfaces replaced (relative to /root/reference/holo_diffusion):
//   RenderMLP / MLPWithInputSkips parameters   holo_voxel_grid_implicit_function.py:73-92,
//                                              custom_modules.py:94-113 (state_dict names `mlp.<i>.0.*`)
//   AdaptiveRaySampler depth bounds            configs/apple.yaml:135-146 (PyTorch3D get_min_max_depth_bounds)
//   HoloDiffusionModel.forward render section  holo_diffusion_model.py:431-457,515-523
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/holo_abi.h"
#include "holo_common.h"
#include "holo_kernels.h"

using namespace holo;

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return HOLO_E_HIP;                                                               \
    }                                                                                  \
  } while (0)

struct HoloRenderer {
  HoloCtx* ctx;
  HoloRenderCfg cfg;
  std::map<std::string, std::vector<float>> host;        // raw parameters (host copies)
  std::map<std::string, std::vector<int64_t>> expected;  // expected shapes
  float* packed = nullptr;  // device: w_feat | b_feat | w_dens | w_rad | w_dir | u_rad | w_fnet [Fd][Hd] | b_fnet [Fd]
  size_t fnet_off = 0;      // offset (floats) of w_fnet inside `packed`
  float k_rad[3] = {0, 0, 0};
  float b_dens = 0.f;
  float b_rad[3] = {0, 0, 0};
  bool committed = false;
  int split3 = 0;  // holo_renderer_set_compute_dtype
  // backward of the training-mode renderer: the float64 stages of the density-net fold (kept by holo_renderer_commit),
  // device copies of the folded weights in the GEMM layouts, the parameter gradients of the last backward (host)
  std::vector<double> fA1, fc1, fA2, fc2, fWe, fbe;
  float* bwd_pack = nullptr;  // WeP [Hp][C] | WeT [C][Hp] | be [Hp]
  bool bwd_pack_valid = false;
  std::map<std::string, std::vector<float>> grads;
  // ... and their device image: ONE upload per backward; holo_renderer_get_grad copies out of it on the device (a
  // per-parameter host->device copy of a 3-float bias would be the tiny-copy pattern holo_ld_sys exists for)
  float* grad_dev = nullptr;
  size_t grad_dev_floats = 0;
  float* stage = nullptr;  // staging buffer of upload_via_stage (commit / backward re-packs)
  size_t stage_floats = 0;
  std::map<std::string, size_t> grad_off;
};

static int dir_emb(const HoloRenderCfg& c) { return 3 * (2 * c.dir_emb_dims + 1); }

extern "C" {

int holo_renderer_create(HoloCtx* ctx, const HoloRenderCfg* cfg, HoloRenderer** out) {
  if (!ctx || !cfg || !out) {
    set_error("holo_renderer_create: null argument");
    return HOLO_E_INVALID;
  }
  if (cfg->dnet_hidden_dim != 256 || cfg->dir_emb_dims != 4 ||
      !(cfg->feature_size == 16 || cfg->feature_size == 32 || cfg->feature_size == 64 || cfg->feature_size == 128) ||
      cfg->n_pts_coarse < 3 || cfg->n_pts_coarse > 64 || cfg->n_pts_fine < 2 || cfg->resol < 2 || cfg->feature_dim < 0 ||
      (cfg->feature_dim & 3)) {
    set_error("holo_renderer_create: unsupported configuration (hidden 256, dir_emb 4, feature_size 16/32/64/128, "
              "3<=n_pts_coarse<=64, n_pts_fine>=2, feature_dim a multiple of 4)");
    return HOLO_E_UNSUPPORTED;
  }
  HoloRenderer* r = new HoloRenderer;
  r->ctx = ctx;
  r->cfg = *cfg;
  const int64_t C = cfg->feature_size, Hd = cfg->dnet_hidden_dim, De = dir_emb(*cfg);
  r->expected["_density_net.mlp.0.0.weight"] = {Hd, C};
  r->expected["_density_net.mlp.0.0.bias"] = {Hd};
  r->expected["_density_net.mlp.1.0.weight"] = {Hd, Hd};
  r->expected["_density_net.mlp.1.0.bias"] = {Hd};
  r->expected["_density_net.mlp.2.0.weight"] = {Hd, Hd + C};
  r->expected["_density_net.mlp.2.0.bias"] = {Hd};
  r->expected["_density_net.mlp.3.0.weight"] = {Hd + 1, Hd};
  r->expected["_density_net.mlp.3.0.bias"] = {Hd + 1};
  r->expected["_radiance_net.mlp.0.0.weight"] = {3, Hd + De};
  r->expected["_radiance_net.mlp.0.0.bias"] = {3};
  const int64_t Fd = cfg->feature_dim;
  if (Fd > 0) {  // the view-point independent feature head (holo_voxel_grid_implicit_function.py:94-105,125-129)
    r->expected["_feature_net.mlp.0.0.weight"] = {Fd, Hd};
    r->expected["_feature_net.mlp.0.0.bias"] = {Fd};
  }
  r->fnet_off = (size_t)(Hd * C + Hd + C + 3 * Hd + 3 * De + 3 * C + 64);
  const size_t n = r->fnet_off + (size_t)(Fd * Hd + Fd);
  if (hipMalloc((void**)&r->packed, n * sizeof(float)) != hipSuccess) {
    set_error("holo_renderer_create: hipMalloc failed");
    delete r;
    return HOLO_E_HIP;
  }
  *out = r;
  return 0;
}

int holo_renderer_destroy(HoloRenderer* r) {
  if (!r) return 0;
  if (r->packed) (void)hipFree(r->packed);
  if (r->bwd_pack) (void)hipFree(r->bwd_pack);
  if (r->grad_dev) (void)hipFree(r->grad_dev);
  if (r->stage) (void)hipFree(r->stage);
  delete r;
  return 0;
}

int holo_renderer_set_param(HoloRenderer* r, const char* name, const void* dev_ptr, int dtype, int ndim,
                            const int64_t* shape, void* stream) {
  if (!r || !name || !dev_ptr) {
    set_error("holo_renderer_set_param: null argument");
    return HOLO_E_INVALID;
  }
  if (dtype != HOLO_DTYPE_F32) {
    set_error("holo_renderer_set_param: only fp32 parameters are supported");
    return HOLO_E_UNSUPPORTED;
  }
  auto it = r->expected.find(name);
  if (it == r->expected.end()) {
    set_error("holo_renderer_set_param: unknown parameter '%s'", name);
    return HOLO_E_INVALID;
  }
  bool ok = ndim == (int)it->second.size();
  int64_t numel = 1;
  for (int i = 0; ok && i < ndim; ++i) {
    ok = shape[i] == it->second[i];
    numel *= shape[i];
  }
  if (!ok) {
    set_error("holo_renderer_set_param: shape mismatch for '%s'", name);
    return HOLO_E_INVALID;
  }
  std::vector<float>& h = r->host[name];
  h.resize((size_t)numel);
  HIP_TRY(hipMemcpyAsync(h.data(), dev_ptr, (size_t)numel * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  r->committed = false;
  return 0;
}

// Fold  y0=W0 f+b0; y1=W1 y0+b1; y2=W2 [y1;f]+b2; o=W3 y2+b3  (no activation in between:
// custom_modules.py:108-112) into  o = We f + be, in float64.
int holo_renderer_commit(HoloRenderer* r, void* stream) {
  if (!r) {
    set_error("holo_renderer_commit: null");
    return HOLO_E_INVALID;
  }
  for (auto& kv : r->expected)
    if (!r->host.count(kv.first)) {
      set_error("holo_renderer_commit: parameter '%s' has not been set", kv.first.c_str());
      return HOLO_E_STATE;
    }
  const int C = r->cfg.feature_size, Hd = r->cfg.dnet_hidden_dim, De = dir_emb(r->cfg);
  auto W = [&](const char* n) -> const std::vector<float>& { return r->host[n]; };
  const auto &W0 = W("_density_net.mlp.0.0.weight"), &b0 = W("_density_net.mlp.0.0.bias");
  const auto &W1 = W("_density_net.mlp.1.0.weight"), &b1 = W("_density_net.mlp.1.0.bias");
  const auto &W2 = W("_density_net.mlp.2.0.weight"), &b2 = W("_density_net.mlp.2.0.bias");
  const auto &W3 = W("_density_net.mlp.3.0.weight"), &b3 = W("_density_net.mlp.3.0.bias");
  const auto &Wr = W("_radiance_net.mlp.0.0.weight"), &br = W("_radiance_net.mlp.0.0.bias");
  // A1 = W1 W0 (Hd x C), c1 = W1 b0 + b1
  std::vector<double> A1((size_t)Hd * C, 0.0), c1(Hd, 0.0);
  for (int i = 0; i < Hd; ++i) {
    double cb = b1[i];
    for (int k = 0; k < Hd; ++k) {
      const double w = W1[(size_t)i * Hd + k];
      cb += w * b0[k];
      for (int j = 0; j < C; ++j) A1[(size_t)i * C + j] += w * W0[(size_t)k * C + j];
    }
    c1[i] = cb;
  }
  // A2 = W2a A1 + W2b (Hd x C), c2 = W2a c1 + b2 ; W2 = [W2a (Hd x Hd) | W2b (Hd x C)]
  std::vector<double> A2((size_t)Hd * C, 0.0), c2(Hd, 0.0);
  const int L2 = Hd + C;
  for (int i = 0; i < Hd; ++i) {
    double cb = b2[i];
    for (int j = 0; j < C; ++j) A2[(size_t)i * C + j] = W2[(size_t)i * L2 + Hd + j];
    for (int k = 0; k < Hd; ++k) {
      const double w = W2[(size_t)i * L2 + k];
      cb += w * c1[k];
      for (int j = 0; j < C; ++j) A2[(size_t)i * C + j] += w * A1[(size_t)k * C + j];
    }
    c2[i] = cb;
  }
  // We = W3 A2 ((Hd+1) x C), be = W3 c2 + b3
  std::vector<double> We((size_t)(Hd + 1) * C, 0.0), be(Hd + 1, 0.0);
  for (int i = 0; i < Hd + 1; ++i) {
    double cb = b3[i];
    for (int k = 0; k < Hd; ++k) {
      const double w = W3[(size_t)i * Hd + k];
      cb += w * c2[k];
      for (int j = 0; j < C; ++j) We[(size_t)i * C + j] += w * A2[(size_t)k * C + j];
    }
    be[i] = cb;
  }
  // pack: w_feat [Hd][C] | b_feat [Hd] | w_dens [C] | w_rad [3][Hd] | w_dir [3][De] | u_rad [3][C]
  std::vector<float> pk((size_t)Hd * C + Hd + C + 3 * Hd + 3 * De + 3 * C);
  size_t o = 0;
  for (int i = 0; i < Hd * C; ++i) pk[o++] = (float)We[i];
  for (int i = 0; i < Hd; ++i) pk[o++] = (float)be[i];
  for (int j = 0; j < C; ++j) pk[o++] = (float)We[(size_t)Hd * C + j];
  for (int c = 0; c < 3; ++c)
    for (int i = 0; i < Hd; ++i) pk[o++] = Wr[(size_t)c * (Hd + De) + i];
  for (int c = 0; c < 3; ++c)
    for (int j = 0; j < De; ++j) pk[o++] = Wr[(size_t)c * (Hd + De) + Hd + j];
  // LeakyReLU_0.2(h) = 0.6 h + 0.4 |h|: fold the linear part of sum_rows w_rad[c][row] * leaky(h[row]) into
  // u_rad[c] = 0.6 * W_eff[:Hd]^T w_rad[c] (a C-vector) and k_rad[c] = 0.6 * w_rad[c] . b_eff[:Hd]
  for (int c = 0; c < 3; ++c) {
    double kc = 0.0;
    std::vector<double> uc(C, 0.0);
    for (int i = 0; i < Hd; ++i) {
      const double w = Wr[(size_t)c * (Hd + De) + i];
      kc += w * be[i];
      for (int j = 0; j < C; ++j) uc[j] += w * We[(size_t)i * C + j];
    }
    for (int j = 0; j < C; ++j) pk[o++] = (float)(0.6 * uc[j]);
    r->k_rad[c] = (float)(0.6 * kc);
  }
  r->b_dens = (float)be[Hd];
  for (int c = 0; c < 3; ++c) r->b_rad[c] = br[c];
  r->fA1 = A1, r->fc1 = c1, r->fA2 = A2, r->fc2 = c2, r->fWe = We, r->fbe = be;
  r->bwd_pack_valid = false;
  r->grads.clear();
  if (r->cfg.feature_dim > 0) {  // the feature head's weights and bias follow the folded pack at fnet_off
    const auto &Wf = W("_feature_net.mlp.0.0.weight"), &bf = W("_feature_net.mlp.0.0.bias");
    pk.resize(r->fnet_off + Wf.size() + bf.size(), 0.f);
    memcpy(pk.data() + r->fnet_off, Wf.data(), Wf.size() * sizeof(float));
    memcpy(pk.data() + r->fnet_off + Wf.size(), bf.data(), bf.size() * sizeof(float));
  }
  if (upload_via_stage(&r->stage, &r->stage_floats, r->packed, pk.data(), pk.size(), stream)) {
    set_error("holo_renderer_commit: upload of the packed RenderMLP failed");
    return HOLO_E_HIP;
  }
  r->committed = true;
  return 0;
}

static void fill_mlp(const HoloRenderer* r, MlpParams& m) {
  const int C = r->cfg.feature_size, Hd = r->cfg.dnet_hidden_dim, De = dir_emb(r->cfg);
  m.w_feat = r->packed;
  m.b_feat = m.w_feat + (size_t)Hd * C;
  m.w_dens = m.b_feat + Hd;
  m.w_rad = m.w_dens + C;
  m.w_dir = m.w_rad + 3 * Hd;
  m.u_rad = m.w_dir + 3 * De;
  m.b_dens = r->b_dens;
  for (int k = 0; k < 3; ++k) {
    m.b_rad[k] = r->b_rad[k];
    m.k_rad[k] = r->k_rad[k];
  }
  m.Hd = Hd;
}

static size_t grid_cl_bytes(const HoloRenderer* r) {
  const size_t R = r->cfg.resol;
  return ((R * R * R * (size_t)r->cfg.feature_size * sizeof(float)) + 255) & ~(size_t)255;
}
// The persistent render kernel runs ONE workgroup per CU; every wave of it is a worker with its own scratch slot
// (64 coarse samples x 32 rays x float4 = 32 KB; the same again for the normals).  The scratch therefore depends on the
// chip and the configuration only - not on the number of cameras or the image size.
static int render_workgroups(const HoloRenderer* r) {
#ifndef HOLO_EMU
  static const char* e = getenv("HOLO_RENDER_WGS");  // development knob
  if (e && atoi(e) > 0) return atoi(e);
#endif
  return r->ctx->num_cus > 0 ? r->ctx->num_cus : 256;
}
static int split3_active(const HoloRenderer* r) { return (r->split3 && r->cfg.feature_size == 32) ? 1 : 0; }
static size_t render_slots(const HoloRenderer* r, int with_normals) {
  return (size_t)render_workgroups(r) *
         (size_t)render_waves_per_wg(r->cfg.feature_size, r->cfg.n_pts_fine, with_normals, split3_active(r), 0);
}
// scratch of the ray-per-column kernel (rendered normals / split arithmetic): one 32 KB slot per resident wave; the
// (ray, depth)-tiled kernel keeps every per-ray value in LDS and needs none
static size_t val_ws_bytes(const HoloRenderer* r, int with_normals) {
  if (render_rays_per_tile(r->cfg.feature_size, r->cfg.n_pts_fine, with_normals, split3_active(r), 0) == 4) return 256;
  return render_slots(r, with_normals) * 64 * 32 * 4 * sizeof(float);
}

static void fill_cam(const HoloRenderCfg& c, const HoloCamera& cam, RenderKernelParams::Cam& pc) {
  for (int k = 0; k < 9; ++k) pc.Rm[k] = cam.R[k];
  for (int k = 0; k < 3; ++k) pc.T[k] = cam.T[k];
  for (int k = 0; k < 2; ++k) {
    pc.focal[k] = cam.focal[k];
    pc.pp[k] = cam.principal_point[k];
  }
  // AdaptiveRaySampler: near/far from the camera centre C = -T R^T (fp32, as torch computes it)
  float d2 = 0.f;
  for (int j = 0; j < 3; ++j) {
    float cj = -(cam.T[0] * cam.R[j * 3 + 0] + cam.T[1] * cam.R[j * 3 + 1] + cam.T[2] * cam.R[j * 3 + 2]);
    const float d = cj - c.scene_center[j];
    d2 += d * d;
  }
  if (d2 < 0.001f) d2 = 0.001f;
  float dist = sqrtf(d2);
  if (dist < c.scene_extent + 1e-3f) dist = c.scene_extent + 1e-3f;
  pc.zmin = dist - c.scene_extent;
  pc.zmax = dist + c.scene_extent;
}

int holo_renderer_set_compute_dtype(HoloRenderer* r, int dtype) {
  if (!r || (dtype != HOLO_DTYPE_F32 && dtype != HOLO_DTYPE_F32_BF16X3)) {
    set_error("holo_renderer_set_compute_dtype: HOLO_DTYPE_F32 or HOLO_DTYPE_F32_BF16X3");
    return HOLO_E_INVALID;
  }
  r->split3 = dtype == HOLO_DTYPE_F32_BF16X3 ? 1 : 0;
  return 0;
}

// the per-voxel scalars w_dens . F[v] of the rendered normals on the (ray, depth)-tiled kernel (behind everything else)
static size_t dens_field_bytes(const HoloRenderer* r) {
  const size_t R = (size_t)r->cfg.resol;
  return ((R * R * R * sizeof(float)) + 255) & ~(size_t)255;
}

size_t holo_render_workspace_bytes(const HoloRenderer* r, int n_cameras, int with_normals) {
  (void)n_cameras;  // the scratch is per resident wave: any number of cameras renders out of the same buffer
  if (!r) return 0;
  return grid_cl_bytes(r) + val_ws_bytes(r, with_normals) * (with_normals ? 2 : 1) + 256 + (with_normals ? dens_field_bytes(r) : 0);
}

int holo_render(HoloRenderer* r, const float* grid, const HoloCamera* cameras, int n_cameras, float* images,
                float* depths, float* masks, float* images_coarse, float* depths_coarse, float* masks_coarse,
                float* normals, float* normals_coarse, void* workspace, size_t workspace_bytes, void* stream) {
  if (!r || !grid || !cameras || n_cameras < 1 || !images || !depths || !masks || !workspace) {
    set_error("holo_render: null/invalid argument");
    return HOLO_E_INVALID;
  }
  if (!r->committed) {
    set_error("holo_render: call holo_renderer_commit after setting the RenderMLP parameters");
    return HOLO_E_STATE;
  }
  if (r->cfg.feature_dim != 0) {
    set_error("holo_render: rendered view-point independent features are not on this path (HoloDiffusionModel builds its "
              "implicit function with feature_dim = 0, holo_diffusion_model.py:156)");
    return HOLO_E_UNSUPPORTED;
  }
  const bool want_nrm = normals != nullptr || normals_coarse != nullptr;
  if (workspace_bytes < holo_render_workspace_bytes(r, n_cameras, want_nrm ? 1 : 0)) {
    set_error("holo_render: workspace too small");
    return HOLO_E_WORKSPACE;
  }
  const HoloRenderCfg& c = r->cfg;
  const int R = c.resol, C = c.feature_size;
  float* grid_cl = (float*)workspace;
  int rc = ncdhw_to_ndhwc_launch(grid, grid_cl, 1, C, (int64_t)R * R * R, 0, stream);
  if (rc) return HOLO_E_INVALID;
  const int H = c.image_height, Wd = c.image_width;
  const int64_t npix = (int64_t)H * Wd;
  float* dens_field = nullptr;
  if (want_nrm && render_rays_per_tile(C, c.n_pts_fine, 1, split3_active(r), 0) == 4) {
    MlpParams mp;
    fill_mlp(r, mp);
    dens_field = (float*)((char*)workspace + grid_cl_bytes(r) + val_ws_bytes(r, 1) * 2 + 256);
    if (density_field_launch(grid_cl, mp.w_dens, C, (int64_t)R * R * R, dens_field, stream)) return HOLO_E_INVALID;
  }
  // frames per launch: the launch parameters hold MAX_CAMS cameras; more cameras are split EVENLY over the launches
  // (40 frames = 20 + 20, not 32 + 8: every launch then ends on an almost full round of wave tiles)
  const int n_launches = (n_cameras + RenderKernelParams::MAX_CAMS - 1) / RenderKernelParams::MAX_CAMS;
  const int G = (n_cameras + n_launches - 1) / n_launches;
  const int n_wgs_max = render_workgroups(r);
  const int sp3 = split3_active(r);
  const int waves_per_wg = render_waves_per_wg(C, c.n_pts_fine, want_nrm ? 1 : 0, sp3, 0);
  const int rays_per_tile = render_rays_per_tile(C, c.n_pts_fine, want_nrm ? 1 : 0, sp3, 0);
#ifndef HOLO_EMU
  static const bool timeline = getenv("HOLO_RENDER_TIMELINE") != nullptr;  // development probe (synchronises!)
  static const char* xcd_env = getenv("HOLO_RENDER_XCD");
  const int xcd = xcd_env ? atoi(xcd_env) : 8;
#else
  const bool timeline = false;
  const int xcd = 2;
#endif
  for (int c0 = 0; c0 < n_cameras; c0 += G) {
    const int ng = n_cameras - c0 < G ? n_cameras - c0 : G;
    RenderKernelParams p;
    memset(&p, 0, sizeof p);
    p.grid_cl = grid_cl;
    p.R = R;
    p.C = C;
    const float voxel_size = c.volume_extent / (float)R;
    p.half_extent = 0.5f * (float)(R - 1) * voxel_size;
    fill_mlp(r, p.mlp);
    p.n_cams = ng;
    for (int g = 0; g < ng; ++g) fill_cam(c, cameras[c0 + g], p.cams[g]);
    p.H = H;
    p.W = Wd;
    if (Wd >= H) {
      p.range_x = (float)Wd / (float)H;
      p.range_y = 1.f;
    } else {
      p.range_x = 1.f;
      p.range_y = (float)H / (float)Wd;
    }
    p.n_coarse = c.n_pts_coarse;
    p.n_fine = c.n_pts_fine;
    for (int k = 0; k < 3; ++k) p.bg[k] = c.bg_color[k];
    p.background_opacity = c.background_opacity;
    p.pdf_eps = c.sample_pdf_eps;
    p.split3 = sp3;
    p.val_ws = (float*)((char*)workspace + grid_cl_bytes(r));
    p.nrm_ws = want_nrm ? (float*)((char*)workspace + grid_cl_bytes(r) + val_ws_bytes(r, 1)) : nullptr;
    p.dens_field = dens_field;
    p.rgb = images + (size_t)c0 * 3 * npix;  // the kernel adds the per-frame offsets
    p.depth = depths + (size_t)c0 * npix;
    p.mask = masks + (size_t)c0 * npix;
    if (images_coarse && depths_coarse && masks_coarse) {
      p.rgb_c = images_coarse + (size_t)c0 * 3 * npix;
      p.depth_c = depths_coarse + (size_t)c0 * npix;
      p.mask_c = masks_coarse + (size_t)c0 * npix;
    }
    p.nrm = normals ? normals + (size_t)c0 * 3 * npix : nullptr;
    p.nrm_c = (normals_coarse && p.rgb_c) ? normals_coarse + (size_t)c0 * 3 * npix : nullptr;
    p.n_tiles = (int64_t)ng * ((npix + rays_per_tile - 1) / rays_per_tile);
    // no more workgroups than there is work for (a tiny launch must not stage the MLP on idle CUs)
    int n_wgs = (int)((p.n_tiles + waves_per_wg - 1) / waves_per_wg);
    if (n_wgs > n_wgs_max) n_wgs = n_wgs_max;
    p.xcd = (xcd > 1 && n_wgs == n_wgs_max && n_wgs % xcd == 0) ? xcd : 1;
#ifndef HOLO_EMU
    static const bool static_tiles = getenv("HOLO_RENDER_STATIC_TILES") != nullptr;  // development knob
#else
    const bool static_tiles = false;
#endif
    if (rays_per_tile == 4 && !static_tiles) {  // dynamic tile hand-out: the counters live in the (otherwise unused) scratch area
      p.tile_ctr = (int*)p.val_ws;
      HIP_TRY(hipMemsetAsync(p.tile_ctr, 0, 8 * sizeof(int), (hipStream_t)stream));
      // the end of every XCD range in single-ray items: about two rounds of them on the range's resident waves
      // (HOLO_RENDER_TAIL=<quads per range>: development knob, 0 = off)
      const int ranges = p.xcd > 1 ? p.xcd : 1;
      p.tail_quads = (n_wgs * waves_per_wg / ranges) / 2;
#ifndef HOLO_EMU
      static const char* tq = getenv("HOLO_RENDER_TAIL");
      if (tq) p.tail_quads = atoi(tq);
#else
      p.tail_quads = 3;  // (the emulation's tiny frames: a few tail tiles in every test)
#endif
    }
    const int nslots = n_wgs * waves_per_wg;
#ifndef HOLO_EMU
    if (timeline) {
      HIP_TRY(hipMalloc((void**)&p.dbg, (size_t)nslots * 64));
      HIP_TRY(hipMemsetAsync(p.dbg, 0, (size_t)nslots * 64, (hipStream_t)stream));
    }
#endif
    rc = render_launch(p, stream, n_wgs);
    if (rc) return HOLO_E_INVALID;
#ifndef HOLO_EMU
    if (timeline) {
      std::vector<unsigned long long> d((size_t)nslots * 8);
      HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
      HIP_TRY(hipMemcpy(d.data(), p.dbg, d.size() * 8, hipMemcpyDeviceToHost));
      (void)hipFree(p.dbg);
      double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tiles = 0;
      for (int w = 0; w < nslots; ++w) {
        for (int k = 0; k < 8; ++k) ph[k] += (double)d[w * 8 + k];
        tiles += (double)d[w * 8 + 4];
      }
      if (tiles < 1) tiles = 1;
      if (rays_per_tile == 4)
        fprintf(stderr, "[render timeline] frames %d slots %d 4-ray tiles %.0f | per tile (us): setup %.2f  coarse eval %.2f  "
                "coarse composite+cdf+inverse-cdf %.2f  new-sample eval %.2f  merged composite %.2f\n", ng, nslots, tiles,
                ph[0] / tiles * 0.01, ph[1] / tiles * 0.01, ph[2] / tiles * 0.01, ph[3] / tiles * 0.01, ph[5] / tiles * 0.01);
      else
        fprintf(stderr, "[render timeline] frames %d slots %d tiles %.0f | per tile: setup %.1f  coarse %.1f  cdf+inverse-cdf %.1f  "
                "fine+composite %.1f us\n", ng, nslots, tiles, ph[0] / tiles * 0.01, ph[1] / tiles * 0.01,
                (ph[2] - ph[1]) / tiles * 0.01, (ph[3] - ph[2]) / tiles * 0.01);
    }
#endif
  }
  return 0;
}

// points per pass of the feature head: the hidden features of a pass (Hd floats per point) live in the workspace
static const int64_t IMPLICIT_CHUNK = 65536;

size_t holo_implicit_workspace_bytes(const HoloRenderer* r, int64_t n_points, int64_t pts_per_dir, int with_features) {
  if (!r || n_points < 0 || pts_per_dir < 1) return 0;
  const int64_t n_dirs = (n_points + pts_per_dir - 1) / pts_per_dir;
  size_t b = grid_cl_bytes(r) + (((size_t)n_dirs * 3 * sizeof(float) + 255) & ~(size_t)255);
  if (with_features && r->cfg.feature_dim > 0) {
    const int64_t chunk = n_points < IMPLICIT_CHUNK ? n_points : IMPLICIT_CHUNK;
    b += (size_t)chunk * r->cfg.dnet_hidden_dim * sizeof(float);
  }
  return b + 256;
}

int holo_implicit_eval_features(HoloRenderer* r, const float* grid, const float* pts, const float* dirs, int64_t n_points,
                                int64_t pts_per_dir, float* densities, float* colours, float* vp_features, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (!r || !grid || !pts || !dirs || !densities || !colours || !workspace || n_points < 0 || pts_per_dir < 1) {
    set_error("holo_implicit_eval: null/invalid argument");
    return HOLO_E_INVALID;
  }
  if (!r->committed) {
    set_error("holo_implicit_eval: call holo_renderer_commit after setting the RenderMLP parameters");
    return HOLO_E_STATE;
  }
  if (vp_features && r->cfg.feature_dim <= 0) {
    set_error("holo_implicit_eval_features: the renderer was created with feature_dim = 0");
    return HOLO_E_INVALID;
  }
  const size_t grid_bytes = grid_cl_bytes(r);
  const int64_t n_dirs = (n_points + pts_per_dir - 1) / pts_per_dir;
  if (workspace_bytes < holo_implicit_workspace_bytes(r, n_points, pts_per_dir, vp_features ? 1 : 0) - 256) {
    set_error("holo_implicit_eval: workspace too small (holo_implicit_workspace_bytes)");
    return HOLO_E_WORKSPACE;
  }
  const HoloRenderCfg& c = r->cfg;
  float* grid_cl = (float*)workspace;
  if (ncdhw_to_ndhwc_launch(grid, grid_cl, 1, c.feature_size, (int64_t)c.resol * c.resol * c.resol, 0, stream))
    return HOLO_E_INVALID;
  ImplicitEvalParams p;
  memset(&p, 0, sizeof p);
  p.grid_cl = grid_cl;
  p.R = c.resol;
  p.C = c.feature_size;
  p.half_extent = 0.5f * (float)(c.resol - 1) * (c.volume_extent / (float)c.resol);
  fill_mlp(r, p.mlp);
  p.pts = pts;
  p.dirs = dirs;
  p.rdir = (float*)((char*)workspace + grid_bytes);
  p.n_points = n_points;
  p.pts_per_dir = pts_per_dir;
  p.densities = densities;
  p.colours = colours;
  if (!vp_features) return implicit_eval_launch(p, stream) ? HOLO_E_INVALID : 0;
  // with the feature head: passes of IMPLICIT_CHUNK points (a multiple of pts_per_dir is not needed: the direction of a
  // point is looked up by its GLOBAL index, so a pass may start anywhere): hidden features -> GEMM with the head's
  // weight -> bias + LeakyReLU (the activation the construction quirk attaches to a last layer, custom_modules.py:108-112)
  const int Hd = c.dnet_hidden_dim, Fd = c.feature_dim;
  float* hidden = (float*)((char*)workspace + grid_bytes + (((size_t)n_dirs * 3 * sizeof(float) + 255) & ~(size_t)255));
  const float* w_fnet = r->packed + r->fnet_off;
  const float* b_fnet = w_fnet + (size_t)Fd * Hd;
  if (implicit_dirs_launch(p, stream)) return HOLO_E_INVALID;
  for (int64_t s0 = 0; s0 < n_points; s0 += IMPLICIT_CHUNK) {
    const int64_t m = n_points - s0 < IMPLICIT_CHUNK ? n_points - s0 : IMPLICIT_CHUNK;
    ImplicitEvalParams q = p;
    q.point0 = s0;
    q.n_points = s0 + m;
    q.hidden = hidden;
    if (implicit_points_launch(q, stream)) return HOLO_E_INVALID;
    GemmParams g;
    memset(&g, 0, sizeof g);
    g.A = hidden;
    g.B = w_fnet;
    g.C = vp_features + s0 * Fd;
    g.M = (int)m;
    g.Nn = Fd;
    g.K = Hd;
    g.lda = Hd;
    g.ldb = Hd;
    g.ldc = Fd;
    g.nb0 = g.nb1 = 1;
    g.alpha = 1.f;
    if (gemm_launch(g, stream)) return HOLO_E_INVALID;
    if (bias_leaky_launch(vp_features + s0 * Fd, b_fnet, m, Fd, stream)) return HOLO_E_INVALID;
  }
  return 0;
}

int holo_implicit_eval(HoloRenderer* r, const float* grid, const float* pts, const float* dirs, int64_t n_points,
                       int64_t pts_per_dir, float* densities, float* colours, void* workspace, size_t workspace_bytes,
                       void* stream) {
  return holo_implicit_eval_features(r, grid, pts, dirs, n_points, pts_per_dir, densities, colours, nullptr, workspace,
                                     workspace_bytes, stream);
}

// Training-mode rendering: an explicit list of rays per camera, optional injected random streams.
}  // extern "C"

// grid_cl: the channels-last grid, already converted.  z_merged / new_flags: optional outputs for the backward pass.
static int render_rays_impl(HoloRenderer* r, const float* grid_cl, const HoloCamera* cameras, int n_cameras, int n_rays,
                            const float* xys, const float* u_coarse, const float* u_fine, const float* noise_coarse,
                            const float* noise_fine, float density_noise_std, float* images, float* depths, float* masks,
                            float* images_coarse, float* depths_coarse, float* masks_coarse, float* z_merged,
                            unsigned char* new_flags, void* stream) {
  const HoloRenderCfg& c = r->cfg;
  const int R = c.resol, C = c.feature_size;
  const int G = RenderKernelParams::MAX_CAMS;
  const int waves_per_wg = render_waves_per_wg(C, c.n_pts_fine, 0, 0, 1);
  const int n_wgs_max = render_workgroups(r);
  const int64_t nm = (int64_t)c.n_pts_coarse + c.n_pts_fine;
  for (int c0 = 0; c0 < n_cameras; c0 += G) {
    const int ng = n_cameras - c0 < G ? n_cameras - c0 : G;
    RenderKernelParams p;
    memset(&p, 0, sizeof p);
    p.grid_cl = grid_cl;
    p.R = R;
    p.C = C;
    p.half_extent = 0.5f * (float)(R - 1) * (c.volume_extent / (float)R);
    fill_mlp(r, p.mlp);
    p.n_cams = ng;
    for (int g = 0; g < ng; ++g) fill_cam(c, cameras[c0 + g], p.cams[g]);
    p.H = c.image_height;
    p.W = c.image_width;
    p.n_coarse = c.n_pts_coarse;
    p.n_fine = c.n_pts_fine;
    for (int k = 0; k < 3; ++k) p.bg[k] = c.bg_color[k];
    p.background_opacity = c.background_opacity;
    p.pdf_eps = c.sample_pdf_eps;
    const int64_t o = (int64_t)c0 * n_rays;
    p.train.n_rays = n_rays;
    p.train.xys = xys + o * 2;
    p.train.u_coarse = u_coarse ? u_coarse + o * c.n_pts_coarse : nullptr;
    p.train.u_fine = u_fine ? u_fine + o * c.n_pts_fine : nullptr;
    p.train.noise_coarse = (noise_coarse && density_noise_std > 0.f) ? noise_coarse + o * c.n_pts_coarse : nullptr;
    p.train.noise_fine = (noise_fine && density_noise_std > 0.f) ? noise_fine + o * nm : nullptr;
    p.train.noise_std = density_noise_std;
    p.train.z_merged = z_merged ? z_merged + o * nm : nullptr;
    p.train.new_flags = new_flags ? new_flags + o * nm : nullptr;
    p.rgb = images + o * 3;
    p.depth = depths + o;
    p.mask = masks + o;
    if (images_coarse && depths_coarse && masks_coarse) {
      p.rgb_c = images_coarse + o * 3;
      p.depth_c = depths_coarse + o;
      p.mask_c = masks_coarse + o;
    }
    p.n_tiles = (int64_t)ng * ((n_rays + 3) / 4);
    int n_wgs = (int)((p.n_tiles + waves_per_wg - 1) / waves_per_wg);
    if (n_wgs > n_wgs_max) n_wgs = n_wgs_max;
    p.xcd = 1;
    if (render_launch(p, stream, n_wgs)) return HOLO_E_INVALID;
  }
  return 0;
}

extern "C" {

int holo_render_rays(HoloRenderer* r, const float* grid, const HoloCamera* cameras, int n_cameras, int n_rays,
                     const float* xys, const float* u_coarse, const float* u_fine, const float* noise_coarse,
                     const float* noise_fine, float density_noise_std, float* images, float* depths, float* masks,
                     float* images_coarse, float* depths_coarse, float* masks_coarse, void* workspace, size_t workspace_bytes,
                     void* stream) {
  if (!r || !grid || !cameras || n_cameras < 1 || n_rays < 1 || !xys || !images || !depths || !masks || !workspace) {
    set_error("holo_render_rays: null/invalid argument");
    return HOLO_E_INVALID;
  }
  if (!r->committed) {
    set_error("holo_render_rays: call holo_renderer_commit after setting the RenderMLP parameters");
    return HOLO_E_STATE;
  }
  if (r->cfg.feature_dim != 0 || r->cfg.feature_size > 64) {
    set_error("holo_render_rays: colours only, feature_size 16/32/64");
    return HOLO_E_UNSUPPORTED;
  }
  if (workspace_bytes < grid_cl_bytes(r) + 256) {
    set_error("holo_render_rays: workspace too small (holo_render_workspace_bytes)");
    return HOLO_E_WORKSPACE;
  }
  const HoloRenderCfg& c = r->cfg;
  const int R = c.resol, C = c.feature_size;
  float* grid_cl = (float*)workspace;
  if (ncdhw_to_ndhwc_launch(grid, grid_cl, 1, C, (int64_t)R * R * R, 0, stream)) return HOLO_E_INVALID;
  return render_rays_impl(r, grid_cl, cameras, n_cameras, n_rays, xys, u_coarse, u_fine, noise_coarse, noise_fine,
                          density_noise_std, images, depths, masks, images_coarse, depths_coarse, masks_coarse, nullptr, nullptr,
                          stream);
}

int holo_implicit_normals(HoloRenderer* r, const float* grid, const float* pts, int64_t n_points, float* normals,
                          void* workspace, size_t workspace_bytes, void* stream) {
  if (!r || !grid || !pts || !normals || !workspace || n_points < 0) {
    set_error("holo_implicit_normals: null/invalid argument");
    return HOLO_E_INVALID;
  }
  if (!r->committed) {
    set_error("holo_implicit_normals: call holo_renderer_commit after setting the RenderMLP parameters");
    return HOLO_E_STATE;
  }
  const size_t grid_bytes = grid_cl_bytes(r);
  if (workspace_bytes < grid_bytes) {
    set_error("holo_implicit_normals: workspace too small (need holo_render_workspace_bytes)");
    return HOLO_E_WORKSPACE;
  }
  const HoloRenderCfg& c = r->cfg;
  float* grid_cl = (float*)workspace;
  if (ncdhw_to_ndhwc_launch(grid, grid_cl, 1, c.feature_size, (int64_t)c.resol * c.resol * c.resol, 0, stream))
    return HOLO_E_INVALID;
  ImplicitEvalParams p;
  memset(&p, 0, sizeof p);
  p.grid_cl = grid_cl;
  p.R = c.resol;
  p.C = c.feature_size;
  p.half_extent = 0.5f * (float)(c.resol - 1) * (c.volume_extent / (float)c.resol);
  fill_mlp(r, p.mlp);
  p.pts = pts;
  p.n_points = n_points;
  return implicit_normals_launch(p, normals, stream) ? HOLO_E_INVALID : 0;
}

}  // extern "C"

// ---- backward of the training-mode renderer (kernels_render_bwd.hip) ------------------------------------
namespace {
struct RbwdLayout {
  int64_t NR, nm, cap, rays_per_chunk;
  int Hd, Hp, C, S;
  size_t o_ggrid, o_fwd, o_zm, o_flags, o_rays, o_grray, o_F, o_YT, o_AT, o_GFT, o_val, o_drad, o_gval, o_GR, o_tmp, o_part,
      o_dWe, o_dbe, o_partr, o_dWrh, o_dir, o_gmax, o_gfix, total;
  bool fixed;  // the deterministic mode of the grid scatter (holo_ctx_set_deterministic)
};
size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }
RbwdLayout rbwd_layout(const HoloRenderer* r, int n_cameras, int n_rays) {
  RbwdLayout L;
  const HoloRenderCfg& c = r->cfg;
  L.NR = (int64_t)n_cameras * n_rays;
  L.nm = (int64_t)c.n_pts_coarse + c.n_pts_fine;
  L.Hd = c.dnet_hidden_dim;
  L.Hp = (L.Hd + 1 + 3) & ~3;
  L.C = c.feature_size;
  L.S = 16;  // splits of the products over the points
  L.rays_per_chunk = 65536 / L.nm;
  if (L.rays_per_chunk < 1) L.rays_per_chunk = 1;
  if (L.rays_per_chunk > L.NR) L.rays_per_chunk = L.NR;
  L.cap = (L.rays_per_chunk * L.nm + 63) & ~(int64_t)63;  // a multiple of 4 * S
  size_t o = grid_cl_bytes(r);
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += al256(bytes);
    return at;
  };
  L.o_ggrid = take(grid_cl_bytes(r));
  L.o_fwd = take((size_t)L.NR * 10 * sizeof(float));
  L.o_zm = take((size_t)L.NR * L.nm * sizeof(float));
  L.o_flags = take((size_t)L.NR * L.nm);
  L.o_rays = take((size_t)L.NR * RBWD_REC * sizeof(float));
  L.o_grray = take((size_t)L.NR * 4 * sizeof(float));
  L.o_F = take((size_t)L.cap * L.C * sizeof(float));
  L.o_YT = take((size_t)L.Hp * L.cap * sizeof(float));
  L.o_AT = take((size_t)L.Hp * L.cap * sizeof(float));
  L.o_GFT = take((size_t)L.C * L.cap * sizeof(float));
  L.o_val = take((size_t)L.cap * 16);
  L.o_drad = take((size_t)L.cap * 16);
  L.o_gval = take((size_t)L.cap * 16);
  L.o_GR = take((size_t)L.cap * 16);
  L.o_tmp = take((size_t)L.cap * 32);
  L.o_part = take((size_t)L.S * L.Hp * L.C * sizeof(float));
  L.o_dWe = take((size_t)L.Hp * L.C * sizeof(float));
  L.o_dbe = take((size_t)L.Hp * sizeof(float));
  L.o_partr = take((size_t)L.S * L.Hd * 4 * sizeof(float));
  L.o_dWrh = take((size_t)L.Hd * 4 * sizeof(float));
  L.o_dir = take(128 * sizeof(float));
  L.o_gmax = take(256);
  L.fixed = r->ctx && r->ctx->deterministic;
  L.o_gfix = L.fixed ? take(2 * grid_cl_bytes(r)) : o;  // 64-bit fixed-point image of the grid gradient
  L.total = o + 256;
  return L;
}
}  // namespace

extern "C" {

size_t holo_render_rays_backward_workspace_bytes(const HoloRenderer* r, int n_cameras, int n_rays) {
  if (!r || n_cameras < 1 || n_rays < 1) return 0;
  return rbwd_layout(r, n_cameras, n_rays).total;
}

int holo_render_rays_backward(HoloRenderer* r, const float* grid, const HoloCamera* cameras, int n_cameras, int n_rays,
                              const float* xys, const float* u_coarse, const float* u_fine, const float* noise_coarse,
                              const float* noise_fine, float density_noise_std, const float* grad_images,
                              const float* grad_depths, const float* grad_masks, const float* grad_images_coarse,
                              const float* grad_depths_coarse, const float* grad_masks_coarse, float* grad_grid,
                              float* merged_depths, unsigned char* merged_is_new, void* workspace, size_t workspace_bytes,
                              void* stream) {
  if (!r || !grid || !cameras || n_cameras < 1 || n_rays < 1 || !xys || !grad_grid || !workspace) {
    set_error("holo_render_rays_backward: null/invalid argument");
    return HOLO_E_INVALID;
  }
  if (!r->committed) {
    set_error("holo_render_rays_backward: call holo_renderer_commit after setting the RenderMLP parameters");
    return HOLO_E_STATE;
  }
  if (r->cfg.feature_dim != 0 || r->cfg.feature_size > 64) {
    set_error("holo_render_rays_backward: colours only, feature_size 16/32/64");
    return HOLO_E_UNSUPPORTED;
  }
  const RbwdLayout L = rbwd_layout(r, n_cameras, n_rays);
  if (workspace_bytes < L.total) {
    set_error("holo_render_rays_backward: workspace too small (holo_render_rays_backward_workspace_bytes)");
    return HOLO_E_WORKSPACE;
  }
  const HoloRenderCfg& c = r->cfg;
  const int R = c.resol, C = c.feature_size, Hd = L.Hd, Hp = L.Hp, De = dir_emb(c);
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  float* grid_cl = (float*)ws;
  float* ggrid_cl = (float*)(ws + L.o_ggrid);
  // folded weights in the GEMM layouts (once per commit)
  if (!r->bwd_pack_valid) {
    if (!r->bwd_pack) HIP_TRY(hipMalloc((void**)&r->bwd_pack, ((size_t)2 * Hp * C + Hp) * sizeof(float)));
    std::vector<float> pk((size_t)2 * Hp * C + Hp, 0.f);
    for (int i = 0; i <= Hd; ++i)
      for (int j = 0; j < C; ++j) {
        pk[(size_t)i * C + j] = (float)r->fWe[(size_t)i * C + j];
        pk[(size_t)Hp * C + (size_t)j * Hp + i] = (float)r->fWe[(size_t)i * C + j];
      }
    for (int i = 0; i <= Hd; ++i) pk[(size_t)2 * Hp * C + i] = (float)r->fbe[i];
    if (upload_via_stage(&r->stage, &r->stage_floats, r->bwd_pack, pk.data(), pk.size(), stream)) {
      set_error("holo_render_rays_backward: upload of the folded weights failed");
      return HOLO_E_HIP;
    }
    r->bwd_pack_valid = true;
  }
  const float* WeP = r->bwd_pack;
  const float* WeT = WeP + (size_t)Hp * C;
  const float* be = WeT + (size_t)Hp * C;
  MlpParams mlp;
  fill_mlp(r, mlp);

  if (ncdhw_to_ndhwc_launch(grid, grid_cl, 1, C, (int64_t)R * R * R, 0, stream)) return HOLO_E_INVALID;
  HIP_TRY(hipMemsetAsync(ggrid_cl, 0, grid_cl_bytes(r), st));
  if (L.fixed) HIP_TRY(hipMemsetAsync(ws + L.o_gfix, 0, 2 * grid_cl_bytes(r), st));
  HIP_TRY(hipMemsetAsync(ws + L.o_AT, 0, (size_t)Hp * L.cap * sizeof(float), st));
  // 1. the forward pass once more: the merged depth list of every ray
  float* fwd = (float*)(ws + L.o_fwd);
  float* z_merged = (float*)(ws + L.o_zm);
  unsigned char* flags = (unsigned char*)(ws + L.o_flags);
  int rc = render_rays_impl(r, grid_cl, cameras, n_cameras, n_rays, xys, u_coarse, u_fine, noise_coarse, noise_fine,
                            density_noise_std, fwd, fwd + L.NR * 3, fwd + L.NR * 4, fwd + L.NR * 5, fwd + L.NR * 8, fwd + L.NR * 9,
                            z_merged, flags, stream);
  if (rc) return rc;
  if (merged_depths)
    HIP_TRY(hipMemcpyAsync(merged_depths, z_merged, (size_t)L.NR * L.nm * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (merged_is_new) HIP_TRY(hipMemcpyAsync(merged_is_new, flags, (size_t)L.NR * L.nm, hipMemcpyDeviceToDevice, st));
  // 2. per-ray records
  float* rays = (float*)(ws + L.o_rays);
  for (int c0 = 0; c0 < n_cameras; c0 += RenderKernelParams::MAX_CAMS) {
    const int ng = n_cameras - c0 < RenderKernelParams::MAX_CAMS ? n_cameras - c0 : RenderKernelParams::MAX_CAMS;
    RenderBwdRays q;
    memset(&q, 0, sizeof q);
    for (int g = 0; g < ng; ++g) fill_cam(c, cameras[c0 + g], q.cams[g]);
    q.n_cams = ng;
    q.n_rays = n_rays;
    q.ray0 = (int64_t)c0 * n_rays;
    q.xys = xys;
    q.rays = rays;
    q.w_dir = mlp.w_dir;
    for (int k = 0; k < 3; ++k) q.b_rad[k] = r->b_rad[k];
    if (rbwd_rays_launch(q, stream)) return HOLO_E_INVALID;
  }
  // 3. chunks of whole rays
  RenderBwdChunk p;
  memset(&p, 0, sizeof p);
  p.grid_cl = grid_cl;
  p.ggrid_cl = ggrid_cl;
  p.gfix = L.fixed ? (long long*)(ws + L.o_gfix) : nullptr;
  p.gfix_max = (uint32_t*)(ws + L.o_gmax);
  p.R = R;
  p.C = C;
  p.half_extent = 0.5f * (float)(R - 1) * (c.volume_extent / (float)R);
  p.Hd = Hd;
  p.Hp = Hp;
  p.nm = (int)L.nm;
  p.n_coarse = c.n_pts_coarse;
  p.rays_per_cam = n_rays;
  p.ld = L.cap;
  p.n_pad = L.cap;
  p.rays = rays;
  p.z_merged = z_merged;
  p.new_flags = flags;
  p.F = (float*)(ws + L.o_F);
  p.YT = (float*)(ws + L.o_YT);
  p.AT = (float*)(ws + L.o_AT);
  p.GFT = (float*)(ws + L.o_GFT);
  p.val = (float4*)(ws + L.o_val);
  p.drad = (float4*)(ws + L.o_drad);
  p.gval = (float4*)(ws + L.o_gval);
  p.GR = (float4*)(ws + L.o_GR);
  p.tmp = (double*)(ws + L.o_tmp);
  p.gr_ray = (float*)(ws + L.o_grray);
  p.be = be;
  p.w_rad = mlp.w_rad;
  const bool noisy = density_noise_std > 0.f;
  p.noise_fine = noisy ? noise_fine : nullptr;
  p.noise_coarse = noisy ? noise_coarse : nullptr;
  p.noise_std = density_noise_std;
  p.g_rgb = grad_images, p.g_depth = grad_depths, p.g_mask = grad_masks;
  p.g_rgb_c = grad_images_coarse, p.g_depth_c = grad_depths_coarse, p.g_mask_c = grad_masks_coarse;
  for (int k = 0; k < 3; ++k) p.bg[k] = c.bg_color[k];
  p.background_opacity = c.background_opacity;
  float* part = (float*)(ws + L.o_part);
  float* dWe = (float*)(ws + L.o_dWe);
  float* dbe = (float*)(ws + L.o_dbe);
  float* partr = (float*)(ws + L.o_partr);
  float* dWrh = (float*)(ws + L.o_dWrh);
  float* ddir = (float*)(ws + L.o_dir);
  const int Ks = (int)(L.cap / L.S);
  int first = 1;
  for (int64_t r0 = 0; r0 < L.NR; r0 += L.rays_per_chunk) {
    const int64_t nr = L.NR - r0 < L.rays_per_chunk ? L.NR - r0 : L.rays_per_chunk;
    p.ray0 = r0;
    p.n_rays_chunk = (int)nr;
    p.n = nr * L.nm;
    if (rbwd_gather_launch(p, stream)) return HOLO_E_INVALID;
    GemmParams g;
    memset(&g, 0, sizeof g);
    g.alpha = 1.f;
    g.nb0 = g.nb1 = 1;
    // YT [Hp][cap] = WeP [Hp][C] . F[cap][C]^T
    g.A = WeP, g.lda = C, g.B = p.F, g.ldb = C, g.b_kmajor = 0, g.C = p.YT, g.ldc = (int)L.cap;
    g.M = Hp, g.Nn = (int)L.cap, g.K = C;
    if (gemm_launch(g, stream)) return HOLO_E_INVALID;
    if (rbwd_point_fwd_launch(p, stream)) return HOLO_E_INVALID;
    if (rbwd_composite_launch(p, stream)) return HOLO_E_INVALID;
    if (rbwd_point_bwd_launch(p, stream)) return HOLO_E_INVALID;
    // GFT [C][cap] = WeT [C][Hp] . YT [Hp][cap]
    g.A = WeT, g.lda = Hp, g.B = p.YT, g.ldb = (int)L.cap, g.b_kmajor = 1, g.C = p.GFT, g.ldc = (int)L.cap;
    g.M = C, g.Nn = (int)L.cap, g.K = Hp;
    if (gemm_launch(g, stream)) return HOLO_E_INVALID;
    // dWe partials [S][Hp][C] = YT[:, split] . F[split, :]
    g.A = p.YT, g.lda = (int)L.cap, g.sa0 = Ks, g.B = p.F, g.ldb = C, g.sb0 = (int64_t)Ks * C, g.b_kmajor = 1;
    g.C = part, g.ldc = C, g.sc0 = (int64_t)Hp * C, g.nb0 = L.S, g.M = Hp, g.Nn = C, g.K = Ks;
    if (gemm_launch(g, stream)) return HOLO_E_INVALID;
    if (partial_reduce_launch(part, dWe, (int64_t)Hp * C, L.S, first ? 0 : 1, stream)) return HOLO_E_INVALID;
    if (rbwd_rowsum_launch(p.YT, L.cap, L.cap, Hp, dbe, first ? 0 : 1, stream)) return HOLO_E_INVALID;
    // dWr_h partials [S][Hd][4] = AT[:, split] . GR[split, :]
    g.A = p.AT, g.lda = (int)L.cap, g.sa0 = Ks, g.B = (const float*)p.GR, g.ldb = 4, g.sb0 = (int64_t)Ks * 4, g.b_kmajor = 1;
    g.C = partr, g.ldc = 4, g.sc0 = (int64_t)Hd * 4, g.nb0 = L.S, g.M = Hd, g.Nn = 4, g.K = Ks;
    if (gemm_launch(g, stream)) return HOLO_E_INVALID;
    if (partial_reduce_launch(partr, dWrh, (int64_t)Hd * 4, L.S, first ? 0 : 1, stream)) return HOLO_E_INVALID;
    if (L.fixed) {  // max |GFT| of the chunk -> fixed-point sums -> added to the gradient in chunk order
      HIP_TRY(hipMemsetAsync(p.gfix_max, 0, sizeof(uint32_t), st));
      if (rbwd_absmax_launch(p, stream)) return HOLO_E_INVALID;
    }
    if (rbwd_scatter_launch(p, stream)) return HOLO_E_INVALID;
    if (L.fixed && rbwd_fix_flush_launch(p, stream)) return HOLO_E_INVALID;
    first = 0;
  }
  if (rbwd_dir_grad_launch(p.gr_ray, rays, L.NR, ddir, stream)) return HOLO_E_INVALID;
  if (ndhwc_to_ncdhw_launch(ggrid_cl, grad_grid, 1, C, (int64_t)R * R * R, stream)) return HOLO_E_INVALID;
  // 4. unfold the gradients of the folded density net to its four Linear layers (float64, host)
  std::vector<float> hWe((size_t)Hp * C), hbe(Hp), hWrh((size_t)Hd * 4), hdir(84);
  HIP_TRY(hipMemcpyAsync(hWe.data(), dWe, hWe.size() * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(hbe.data(), dbe, hbe.size() * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(hWrh.data(), dWrh, hWrh.size() * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(hdir.data(), ddir, hdir.size() * sizeof(float), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const int H1 = Hd + 1, L2 = Hd + C;
  auto W = [&](const char* n) -> const std::vector<float>& { return r->host[n]; };
  const auto &W0 = W("_density_net.mlp.0.0.weight"), &b0 = W("_density_net.mlp.0.0.bias");
  const auto& W1 = W("_density_net.mlp.1.0.weight");
  const auto& W2 = W("_density_net.mlp.2.0.weight");
  const auto& W3 = W("_density_net.mlp.3.0.weight");
  std::vector<double> G((size_t)H1 * C), gv(H1);
  for (int i = 0; i < H1; ++i) {
    gv[i] = hbe[i];
    for (int j = 0; j < C; ++j) G[(size_t)i * C + j] = hWe[(size_t)i * C + j];
  }
  auto out = [&](const char* n, size_t sz) -> std::vector<float>& {
    std::vector<float>& v = r->grads[n];
    v.assign(sz, 0.f);
    return v;
  };
  {  // layer 3: y3 = A2 f + c2
    auto& dW3 = out("_density_net.mlp.3.0.weight", (size_t)H1 * Hd);
    auto& db3 = out("_density_net.mlp.3.0.bias", H1);
    for (int i = 0; i < H1; ++i) {
      db3[i] = (float)gv[i];
      for (int k = 0; k < Hd; ++k) {
        double s = gv[i] * r->fc2[k];
        for (int j = 0; j < C; ++j) s += G[(size_t)i * C + j] * r->fA2[(size_t)k * C + j];
        dW3[(size_t)i * Hd + k] = (float)s;
      }
    }
  }
  std::vector<double> G3((size_t)Hd * C, 0.0), g3(Hd, 0.0);  // W3^T G, W3^T g
  for (int i = 0; i < H1; ++i)
    for (int k = 0; k < Hd; ++k) {
      const double w = W3[(size_t)i * Hd + k];
      g3[k] += w * gv[i];
      for (int j = 0; j < C; ++j) G3[(size_t)k * C + j] += w * G[(size_t)i * C + j];
    }
  {  // layer 2: input [y2 = A1 f + c1 ; f]
    auto& dW2 = out("_density_net.mlp.2.0.weight", (size_t)Hd * L2);
    auto& db2 = out("_density_net.mlp.2.0.bias", Hd);
    for (int i = 0; i < Hd; ++i) {
      db2[i] = (float)g3[i];
      for (int k = 0; k < Hd; ++k) {
        double s = g3[i] * r->fc1[k];
        for (int j = 0; j < C; ++j) s += G3[(size_t)i * C + j] * r->fA1[(size_t)k * C + j];
        dW2[(size_t)i * L2 + k] = (float)s;
      }
      for (int j = 0; j < C; ++j) dW2[(size_t)i * L2 + Hd + j] = (float)G3[(size_t)i * C + j];
    }
  }
  std::vector<double> G2((size_t)Hd * C, 0.0), g2(Hd, 0.0);  // W2a^T G3, W2a^T g3
  for (int i = 0; i < Hd; ++i)
    for (int k = 0; k < Hd; ++k) {
      const double w = W2[(size_t)i * L2 + k];
      g2[k] += w * g3[i];
      for (int j = 0; j < C; ++j) G2[(size_t)k * C + j] += w * G3[(size_t)i * C + j];
    }
  {  // layer 1: input y1 = W0 f + b0
    auto& dW1 = out("_density_net.mlp.1.0.weight", (size_t)Hd * Hd);
    auto& db1 = out("_density_net.mlp.1.0.bias", Hd);
    for (int i = 0; i < Hd; ++i) {
      db1[i] = (float)g2[i];
      for (int k = 0; k < Hd; ++k) {
        double s = g2[i] * b0[k];
        for (int j = 0; j < C; ++j) s += G2[(size_t)i * C + j] * W0[(size_t)k * C + j];
        dW1[(size_t)i * Hd + k] = (float)s;
      }
    }
  }
  {  // layer 0
    auto& dW0 = out("_density_net.mlp.0.0.weight", (size_t)Hd * C);
    auto& db0 = out("_density_net.mlp.0.0.bias", Hd);
    std::vector<double> G1((size_t)Hd * C, 0.0), g1(Hd, 0.0);
    for (int i = 0; i < Hd; ++i)
      for (int k = 0; k < Hd; ++k) {
        const double w = W1[(size_t)i * Hd + k];
        g1[k] += w * g2[i];
        for (int j = 0; j < C; ++j) G1[(size_t)k * C + j] += w * G2[(size_t)i * C + j];
      }
    for (int k = 0; k < Hd; ++k) {
      db0[k] = (float)g1[k];
      for (int j = 0; j < C; ++j) dW0[(size_t)k * C + j] = (float)G1[(size_t)k * C + j];
    }
  }
  {  // radiance layer: [hidden | direction embedding]
    auto& dWr = out("_radiance_net.mlp.0.0.weight", (size_t)3 * (Hd + De));
    auto& dbr = out("_radiance_net.mlp.0.0.bias", 3);
    for (int j = 0; j < 3; ++j) {
      for (int h = 0; h < Hd; ++h) dWr[(size_t)j * (Hd + De) + h] = hWrh[(size_t)h * 4 + j];
      for (int e = 0; e < De; ++e) dWr[(size_t)j * (Hd + De) + Hd + e] = hdir[j * 28 + e];
      dbr[j] = hdir[j * 28 + 27];
    }
  }
  {  // device image of all parameter gradients
    size_t total = 0;
    r->grad_off.clear();
    for (auto& kv : r->grads) {
      r->grad_off[kv.first] = total;
      total += (kv.second.size() + 63) & ~(size_t)63;
    }
    if (total > r->grad_dev_floats) {
      if (r->grad_dev) (void)hipFree(r->grad_dev);
      r->grad_dev = nullptr;
      HIP_TRY(hipMalloc((void**)&r->grad_dev, total * sizeof(float)));
      r->grad_dev_floats = total;
    }
    std::vector<float> pack(total, 0.f);
    for (auto& kv : r->grads) memcpy(pack.data() + r->grad_off[kv.first], kv.second.data(), kv.second.size() * sizeof(float));
    HIP_TRY(hipMemcpyAsync(r->grad_dev, pack.data(), total * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  return 0;
}

int holo_renderer_get_grad(HoloRenderer* r, const char* name, float* out_dev, int64_t numel, void* stream) {
  if (!r || !name || !out_dev) {
    set_error("holo_renderer_get_grad: null argument");
    return HOLO_E_INVALID;
  }
  auto it = r->grads.find(name);
  if (it == r->grads.end()) {
    set_error("holo_renderer_get_grad: no gradient for '%s' (run holo_render_rays_backward first)", name);
    return HOLO_E_STATE;
  }
  if ((int64_t)it->second.size() != numel) {
    set_error("holo_renderer_get_grad: '%s' has %lld elements, not %lld", name, (long long)it->second.size(), (long long)numel);
    return HOLO_E_INVALID;
  }
  if (!r->grad_dev || !r->grad_off.count(name)) {
    set_error("holo_renderer_get_grad: no device image of the gradients (run holo_render_rays_backward first)");
    return HOLO_E_STATE;
  }
  return copy_sys_launch(r->grad_dev + r->grad_off[name], out_dev, numel, stream) ? HOLO_E_INVALID : 0;
}

}  // extern "C"
