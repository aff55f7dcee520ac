/*
 * holo_abi.h — C ABI of libholo_mi355x.so, the MI355X-native denoise-and-render hot path
 * of HoloDiffusion.
 *
 * The reference has no native code and no FFI: the hot path sits behind PyTorch3D-Implicitron
 * registry plugins.  The Python plugin classes in holo_diffusion_amd/ keep that
 * surface and bind the entry points below through ctypes; each entry point names the reference
 * interface it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - every function returns 0 on success, a negative HOLO_E_* code on failure; the message is
 *     available from holo_last_error() (thread-local).  No exceptions cross the boundary.
 *   - all tensor pointers are DEVICE pointers owned by the caller (torch) unless a parameter says
 *     "host"; they must stay valid until the stream work that uses them has completed.
 *   - no hidden device synchronisation and no allocation inside *_forward / *_step / holo_render:
 *     the caller supplies the workspace (size from *_workspace_bytes).  *_create / *_set_param /
 *     *_commit may allocate and synchronise (setup time).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - fp32 everywhere (the reference computes in fp32: unet.py:639 `self.dtype = th.float32`).
 */
#ifndef HOLO_ABI_H
#define HOLO_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HOLO_ABI_VERSION 6

enum {
  HOLO_OK = 0,
  HOLO_E_INVALID = -1,   /* bad argument / unknown parameter name / shape mismatch */
  HOLO_E_HIP = -2,       /* a HIP runtime call failed */
  HOLO_E_WORKSPACE = -3, /* workspace too small */
  HOLO_E_STATE = -4,     /* parameters missing / not committed */
  HOLO_E_UNSUPPORTED = -5
};

enum { HOLO_DTYPE_F32 = 0, HOLO_DTYPE_BF16 = 1, HOLO_DTYPE_F32_BF16X3 = 2 };

typedef struct HoloCtx HoloCtx;
typedef struct HoloUnet HoloUnet;
typedef struct HoloRenderer HoloRenderer;

int holo_abi_version(void);
const char* holo_last_error(void);

/* One context per process per GPU (one process per GPU). */
int holo_ctx_create(int device_id, HoloCtx** out);
int holo_ctx_destroy(HoloCtx* ctx);

/* Deterministic mode of the backward entries that scatter-add (ABI 6).  The reference's autograd scatters with atomic adds
 * (grid_sample's backward on CUDA is order-dependent in the same way); by default so do holo_render_rays_backward (grid
 * gradient), holo_view_pool_backward and holo_mlp_mean_backward (feature-map gradients): hardware fp32 atomics, sums that
 * differ in their last bits from run to run.  With on != 0 every such scatter of handles created from `ctx` adds 64-bit
 * FIXED-POINT integers instead (binary point from the largest magnitude of the launch, 2^-40 of it per addend; integer
 * addition commutes), so two calls on the same inputs return bit-identical gradients.  Costs one extra pass over the
 * scattered values and an 8-byte image of the destination in the workspace (re-query *_workspace_bytes after switching).
 * Everything else in the library is deterministic in both modes (fixed-order partial sums). */
int holo_ctx_set_deterministic(HoloCtx* ctx, int on);
int holo_ctx_get_deterministic(const HoloCtx* ctx);

/* ------------------------------------------------------------------------------------------
 * Denoiser.  Replaces SimpleUnet3D / UNetModel:
 *   holo_diffusion/utils/diffusion_utils.py:41-86   (plugin + ctor-arg mapping)
 *   holo_diffusion/guided_diffusion/unet.py:566-837 (UNetModel ctor + forward)
 * The config fields are SimpleUnet3D's dataclass fields (diffusion_utils.py:43-53).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t image_size;
  int32_t in_channels;
  int32_t out_channels;
  int32_t model_channels;
  int32_t num_res_blocks;
  int32_t n_channel_mult;
  int32_t channel_mult[8];
  int32_t n_attention_resolutions;
  int32_t attention_resolutions[8];
  int32_t num_heads;
  int32_t homogeneous_resample; /* only 1 is supported (diffusion_utils.py:53 default) */
} HoloUnetCfg;

int holo_unet_create(HoloCtx* ctx, const HoloUnetCfg* cfg, HoloUnet** out);
int holo_unet_destroy(HoloUnet* net);

/* Parameter enumeration: names are the reference state_dict keys below `net_3d._net.`
 * (e.g. "input_blocks.1.0.in_layers.2.weight"), shapes the reference shapes. */
int holo_unet_num_params(const HoloUnet* net);
int holo_unet_param_info(const HoloUnet* net, int index, char* name, int name_cap, int64_t shape[8], int* ndim);

/* Bind one parameter.  The library keeps a repacked private copy (conv weights in an MFMA-fragment-packed
 * [tap][Cin/32][Cout/16][...] layout, concatenated embedding linears); call again after the caller's tensor changes. */
int holo_unet_set_param(HoloUnet* net, const char* name, const void* dev_ptr, int dtype, int ndim,
                        const int64_t* shape, void* stream);

/* Arithmetic and storage mode of the denoiser:
 *   HOLO_DTYPE_F32  (default) exact-fp32 MFMA, the reference's arithmetic (unet.py:639)
 *   HOLO_DTYPE_BF16 bf16 mode: every activation inside the library is STORED as bf16 in the workspace (half the HBM
 *                   footprint and traffic), convolutions and the attention of long sequences multiply on the bf16
 *                   matrix cores with fp32 accumulation, GroupNorm statistics are fp32/double sums of the fp32
 *                   values before rounding, x / y at the boundary and the attention's qkv stay fp32.  Opt-in for
 *                   the bf16 configurations (BASELINE configs[4]); tolerance rtol 2e-2 of the output scale.
 *   HOLO_DTYPE_F32_BF16X3  fp32-accurate arithmetic on the bf16 matrix cores: every fp32 operand is split exactly
 *                   into three bf16 terms and each product is assembled from the six leading cross terms (dropped
 *                   terms <= 2^-23 relative, the size of one fp32 rounding), fp32 accumulation.  Opt-in; meets the
 *                   fp32 tolerances of the parity tests.
 * May be called at any time; takes effect at the next forward. */
int holo_unet_set_compute_dtype(HoloUnet* net, int dtype);

size_t holo_unet_workspace_bytes(HoloUnet* net, int batch);

/* y = UNetModel.forward(x, timesteps)  (unet.py:800-837).
 *   x, y        : (batch, C, R, R, R) fp32, NCDHW contiguous (the plugin boundary layout)
 *   timesteps   : (batch,) int64 on the device (as the reference passes them, gaussian_diffusion.py:630) */
int holo_unet_forward(HoloUnet* net, int batch, const float* x, const int64_t* timesteps, float* y,
                      void* workspace, size_t workspace_bytes, void* stream);

/* The same forward on CHANNELS-LAST tensors (ABI 5): x_cl / y_cl are (N, R, R, R, C) fp32 - the layout the library works in -
 * so the two layout passes of holo_unet_forward (NCDHW -> channels-last of x, back of y) do not run: the first convolution
 * reads x_cl and the last one writes y_cl directly.  For sampling chains that stay on the device (holo_ddpm_step* is
 * elementwise, hence layout-agnostic): the chain converts once at its start and once at its end.  All modes (ABI 6: in the
 * bf16 storage mode an element-wise fp32 -> bf16 cast of x_cl replaces the transposing layout pass; y_cl is written by the
 * last convolution as in the other modes). */
int holo_unet_forward_cl(HoloUnet* net, int batch, const float* x_cl, const int64_t* timesteps, float* y_cl, void* workspace,
                         size_t workspace_bytes, void* stream);

/* Debug/parity hook: copy an intermediate block output (NCDHW) of the LAST forward into `dst`.
 * tag = "input_blocks.<i>", "middle_block", "output_blocks.<i>".  Returns element count in *numel. */
int holo_unet_fetch_block(HoloUnet* net, const char* tag, float* dst, int64_t dst_capacity, int64_t* numel,
                          void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the denoiser.  Replaces autograd through UNetModel.forward
 * (holo_diffusion/guided_diffusion/unet.py:800-837): what `output.mean().backward()` does in the reference's own backward
 * test (holo_diffusion/tests/test_diffusion_utils.py:47-66) and what the training step needs from net_3d
 * (holo_diffusion_model.py:386-418).  fp32 mode.
 *   holo_unet_set_dgrad_weight   per convolution weight (same names / OIDHW tensors as holo_unet_set_param): prepares the
 *                                weight of the transposed convolution (packed, and for the wide levels its Winograd
 *                                copies; device buffers are allocated on the first call per weight); call again
 *                                whenever the parameter changes
 *   holo_unet_backward           runs the forward (every intermediate kept in the workspace) and the backward for
 *                                grad_out = dL/dy (N, Cout, R, R, R); y (optional) receives the forward output, grad_x
 *                                (optional) dL/dx.  Parameter gradients stay in the workspace ...
 *   holo_unet_get_grad           ... and are copied out by name, in the reference's parameter layouts, from the
 *                                workspace of the last backward call.
 * No allocation / synchronisation inside holo_unet_backward; workspace size from holo_unet_backward_workspace_bytes. */
int holo_unet_set_dgrad_weight(HoloUnet* net, const char* name, const void* dev_ptr, void* stream);
size_t holo_unet_backward_workspace_bytes(HoloUnet* net, int batch);
int holo_unet_backward(HoloUnet* net, int batch, const float* x, const int64_t* timesteps, const float* grad_out, float* y,
                       float* grad_x, void* workspace, size_t workspace_bytes, void* stream);

/* The two halves of holo_unet_backward as separate entries (ABI 4), for a cotangent that depends on the output (the clamp of
 * pred_xstart in the training branch, holo_diffusion_model.py:400-418): holo_unet_forward_train runs the taped forward and
 * writes y (may be NULL); holo_unet_backward_taped then runs the backward for grad_out on the SAME workspace and batch and
 * consumes the tape (HOLO_E_STATE without a preceding forward_train; `timesteps` of the forward must still be alive).
 * Gradients are fetched with holo_unet_get_grad as after holo_unet_backward. */
int holo_unet_forward_train(HoloUnet* net, int batch, const float* x, const int64_t* timesteps, float* y, void* workspace,
                            size_t workspace_bytes, void* stream);
int holo_unet_backward_taped(HoloUnet* net, int batch, const float* grad_out, float* grad_x, void* workspace,
                             size_t workspace_bytes, void* stream);
int holo_unet_get_grad(HoloUnet* net, const char* name, float* dst, int64_t numel, const void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * DDPM ancestral step.  Replaces the elementwise tail of GaussianDiffusion.p_sample:
 *   gaussian_diffusion.py:314-343 (clamp, START_X), :237-240 (posterior mean), :499-506 (noise add)
 *   tables   : (T, 4) fp32 on the device: {posterior_mean_coef1, posterior_mean_coef2,
 *              posterior_log_variance_clipped, 0}, cast from the float64 schedule (:1056)
 *   timesteps: (batch,) int64 on the device
 *   sample = c1*clamp(model_out) + c2*x_t + [t!=0]*exp(0.5*logvar)*noise ;  pred_xstart = clamp(model_out)
 * ------------------------------------------------------------------------------------------ */
int holo_ddpm_step(HoloCtx* ctx, const float* tables, int num_timesteps, const int64_t* timesteps, int batch,
                   int64_t elems_per_sample, const float* x_t, const float* model_out, const float* noise,
                   int clip_denoised, float* sample, float* pred_xstart, void* stream);

/* The same step with the noise drawn inside the kernel (ABI 5; perf mode of gaussian_diffusion.py:498 `th.randn_like(x)`
 * / :604 - SURVEY 8d "on-device Philox"): element quad q of sample b takes the four outputs of Philox4x32-10 with
 * counter (q low, q high, b, stream_offset low) and key (seed low, seed high ^ stream_offset high), as two Box-Muller
 * pairs (24-bit uniforms, |z| <= 5.9).  Statistically equivalent to, NOT bit-equal with, torch's generator: parity
 * tests keep holo_ddpm_step with injected noise.  pred_xstart and noise_out may be null (not written).
 * The quads are numbered over the LOGICAL elements in channels-last order, q = voxel * (C / 4) + channel / 4 (ABI 6):
 *   ncdhw_channels = 0  the tensors are channels-last (N, R, R, R, C) - or any flat order the caller treats as canonical:
 *                       quad q is four consecutive floats (the sampler's channels-last chain, holo_unet_forward_cl)
 *   ncdhw_channels = C  the tensors are (N, C, R, R, R) with C a multiple of 4: the same quad's four values are taken from
 *                       four channel planes, so a chain draws the SAME noise for element (n, c, z, y, x) in either layout. */
int holo_ddpm_step_philox(HoloCtx* ctx, const float* tables, int num_timesteps, const int64_t* timesteps, int batch,
                          int64_t elems_per_sample, const float* x_t, const float* model_out, uint64_t seed,
                          uint64_t stream_offset, int clip_denoised, float* sample, float* pred_xstart, float* noise_out,
                          int ncdhw_channels, void* stream);

/* Elementwise helpers on the path: torch.tanh (holo_diffusion_model.py:425) and
 * torch.clip(x,-1,1) (holo_diffusion_model.py:186). */
int holo_tanh(HoloCtx* ctx, const float* x, float* y, int64_t n, void* stream);
int holo_clip(HoloCtx* ctx, const float* x, float* y, float lo, float hi, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Renderer.  Replaces, fused in one kernel per frame:
 *   AdaptiveRaySampler call            holo_diffusion_model.py:442-448 (+ configs/apple.yaml:135-146)
 *   GenericModel._render chunk loop    holo_diffusion_model.py:451-457
 *   HoloMultiPassEmissionAbsorptionRenderer._run_raymarcher   holo_multipass_ea.py:79-125
 *   HoloVoxelGridImplicitFunction.forward                     holo_voxel_grid_implicit_function.py:182-269
 *   RenderMLP.forward / MLPWithInputSkips                     holo_voxel_grid_implicit_function.py:107-129,
 *                                                             custom_modules.py:133-160
 *   EmissionAbsorptionRaymarcher / RayPointRefiner (PyTorch3D 0.7.4; configs/apple.yaml:147-165)
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t resol;            /* HoloDiffusionModel.resol */
  int32_t feature_size;     /* HoloDiffusionModel.feature_size (= RenderMLP.input_dims) */
  float volume_extent;      /* 8.0 */
  float scene_extent;       /* raysampler_AdaptiveRaySampler_args.scene_extent = 4.0 */
  float scene_center[3];
  int32_t n_pts_coarse;     /* n_pts_per_ray_evaluation = 64 */
  int32_t n_pts_fine;       /* n_pts_per_ray_fine_evaluation = 64 (16 in unet_with_no_diffusion.yaml) */
  int32_t image_height;
  int32_t image_width;
  float bg_color[3];
  float background_opacity; /* 1e10 */
  int32_t dnet_hidden_dim;  /* 256 */
  int32_t dir_emb_dims;     /* 4 */
  float sample_pdf_eps;     /* 1e-5 */
  int32_t feature_dim;      /* RenderMLP.output_vp_independent_feature_dims (HoloVoxelGridImplicitFunction.feature_dim):
                               0 inside HoloDiffusionModel (holo_diffusion_model.py:156), 64 by default; > 0 adds the
                               parameters "_feature_net.mlp.0.0.{weight,bias}" and is served by
                               holo_implicit_eval_features only */
} HoloRenderCfg;

/* One camera in PyTorch3D PerspectiveCameras/NDC convention (X_cam = X_world R + T), host memory. */
typedef struct {
  float R[9]; /* row-major 3x3 */
  float T[3];
  float focal[2];
  float principal_point[2];
} HoloCamera;

int holo_renderer_create(HoloCtx* ctx, const HoloRenderCfg* cfg, HoloRenderer** out);
int holo_renderer_destroy(HoloRenderer* r);

/* RenderMLP parameters by reference name below `..._fn.render_mlp.`:
 * "_density_net.mlp.{0..3}.0.{weight,bias}", "_radiance_net.mlp.0.0.{weight,bias}". */
int holo_renderer_set_param(HoloRenderer* r, const char* name, const void* dev_ptr, int dtype, int ndim,
                            const int64_t* shape, void* stream);
/* Folds the activation-free density layers (custom_modules.py:108-112 quirk) into one affine map in
 * float64 and uploads the packed weights.  Must be called after all set_param calls. */
int holo_renderer_commit(HoloRenderer* r, void* stream);

/* Arithmetic of the RenderMLP hidden layer inside holo_render: HOLO_DTYPE_F32 (default, exact-fp32 MFMA) or
 * HOLO_DTYPE_F32_BF16X3 (weights and interpolated features split exactly into three bf16 terms, six bf16 MFMAs per
 * product, fp32 accumulation; feature_size 32 only, other sizes keep the exact path).  Opt-in. */
int holo_renderer_set_compute_dtype(HoloRenderer* r, int dtype);

/* Scratch of holo_render.  The renderer is a PERSISTENT kernel (one workgroup per CU, every wave a worker with a
 * private 32 KB slot for the coarse-pass values of its 32 rays - twice that when normals are rendered), so the size
 * depends on the device and the configuration only: `n_cameras` is accepted for interface stability and ignored. */
size_t holo_render_workspace_bytes(const HoloRenderer* r, int n_cameras, int with_normals);

/* Render n_cameras full-grid frames of one voxel grid (up to 32 cameras per kernel launch; all wave tiles of all
 * frames of a launch are walked by the resident waves, so a launch has one tail whatever the frame count).
 *   grid        : (1, C, R, R, R) fp32 NCDHW (what HoloDiffusionModel.forward binds as
 *                 voxel_grid_features, holo_diffusion_model.py:431-438)
 *   cameras     : host array of n_cameras HoloCamera (depth bounds are computed per camera)
 *   images      : (n_cameras, 3, H, W); depths, masks : (n_cameras, 1, H, W)   [fine pass]
 *   *_coarse    : optional (may be NULL) outputs of the coarse pass (RendererOutput.prev_stage)
 *   normals, normals_coarse : optional (may be NULL) rendered normals sum_i w_i n_i, (n_cameras, 3, H, W)
 *                 (holo_multipass_ea.py:105-109 with render_normals=True; n_i = RenderMLP.get_normals,
 *                 holo_voxel_grid_implicit_function.py:131-145, evaluated analytically); normals_coarse needs the
 *                 other *_coarse outputs */
int holo_render(HoloRenderer* r, const float* grid, const HoloCamera* cameras, int n_cameras, float* images,
                float* depths, float* masks, float* images_coarse, float* depths_coarse, float* masks_coarse,
                float* normals, float* normals_coarse, void* workspace, size_t workspace_bytes, void* stream);

/* Training-mode rendering: the same two-pass renderer on an EXPLICIT list of rays per camera with the
 * random streams of the reference's training branch injected by the caller:
 *   xys           (n_cameras, n_rays, 2) NDC coordinates of the rays (mask-sampled rays of the AdaptiveRaySampler,
 *                 configs/apple.yaml:135-146; PyTorch3D convention +x left, +y up)
 *   u_coarse      (n_cameras, n_rays, n_pts_coarse) uniforms in [0,1) or NULL: stratified depths (PyTorch3D
 *                 _jiggle_within_stratas; stratified_point_sampling_training)
 *   u_fine        (n_cameras, n_rays, n_pts_fine) uniforms or NULL: stratified importance samples (sample_pdf det=False)
 *   noise_coarse  (n_cameras, n_rays, n_pts_coarse), noise_fine (n_cameras, n_rays, n_pts_coarse + n_pts_fine, in DEPTH
 *                 ORDER) standard normals or NULL: raw density += density_noise_std * noise before the ReLU
 *                 (holo_multipass_ea.py:77,87-91; the fine pass draws a new value for each of its sorted points)
 *   images (n_cameras, 3, n_rays), depths / masks (n_cameras, n_rays); *_coarse optional (all three or none).
 * n_pts_coarse / n_pts_fine come from the renderer's configuration (the training values of the YAML). */
int holo_render_rays(HoloRenderer* r, const float* grid, const HoloCamera* cameras, int n_cameras, int n_rays,
                     const float* xys, const float* u_coarse, const float* u_fine, const float* noise_coarse,
                     const float* noise_fine, float density_noise_std, float* images, float* depths, float* masks,
                     float* images_coarse, float* depths_coarse, float* masks_coarse, void* workspace, size_t workspace_bytes,
                     void* stream);

/* Backward of holo_render_rays: what autograd computes for the reference's rendering losses
 * (holo_diffusion_model.py:458-489 on the outputs of holo_multipass_ea.py:79-125) - the gradients of BOTH passes' rgb /
 * depth / mask with respect to the voxel grid (grid_sample's scatter-add, holo_voxel_grid_implicit_function.py:221-243)
 * and the RenderMLP parameters (:73-129).  Same ray / random-stream arguments as the forward call; the importance
 * sampling carries no gradient (PyTorch3D RayPointRefiner samples under torch.no_grad()).
 *   grad_images (n_cameras, 3, n_rays), grad_depths / grad_masks (n_cameras, n_rays), *_coarse the same for the coarse
 *   pass; any of the six may be NULL (= zero)
 *   grad_grid   : (1, C, R, R, R) fp32, overwritten
 *   merged_depths (n_cameras, n_rays, n_pts_coarse + n_pts_fine) / merged_is_new (same shape, bytes): optional outputs
 *                 (may be NULL) - the fine pass's depth list of every ray in depth order and a flag per importance sample
 *                 (the sample placement of the forward kernel, which the backward pass holds fixed)
 * The parameter gradients of the call are fetched with holo_renderer_get_grad (names of holo_renderer_set_param; device
 * destination of `numel` floats).  The call synchronises the stream (the gradients of the folded density net are unfolded
 * to its four Linear layers on the host in float64, like holo_renderer_commit folds them). */
size_t holo_render_rays_backward_workspace_bytes(const HoloRenderer* r, int n_cameras, int n_rays);
int holo_render_rays_backward(HoloRenderer* r, const float* grid, const HoloCamera* cameras, int n_cameras, int n_rays,
                              const float* xys, const float* u_coarse, const float* u_fine, const float* noise_coarse,
                              const float* noise_fine, float density_noise_std, const float* grad_images,
                              const float* grad_depths, const float* grad_masks, const float* grad_images_coarse,
                              const float* grad_depths_coarse, const float* grad_masks_coarse, float* grad_grid,
                              float* merged_depths, unsigned char* merged_is_new, void* workspace, size_t workspace_bytes,
                              void* stream);
int holo_renderer_get_grad(HoloRenderer* r, const char* name, float* out_dev, int64_t numel, void* stream);

/* Stand-alone implicit function.  Replaces HoloVoxelGridImplicitFunction.forward
 * (holo_voxel_grid_implicit_function.py:182-269): trilinear fetch of `grid` at the world points, RenderMLP.
 *   pts        : (n_points, 3) world coordinates; dirs : (n_points / pts_per_dir, 3) ray directions
 *                (normalised inside like F.normalize, :239); point i uses dirs[i / pts_per_dir]
 *   densities  : (n_points,) raw densities (no softplus, :113-114); colours : (n_points, 3) after sigmoid */
int holo_implicit_eval(HoloRenderer* r, const float* grid, const float* pts, const float* dirs, int64_t n_points,
                       int64_t pts_per_dir, float* densities, float* colours, void* workspace, size_t workspace_bytes,
                       void* stream);
/* The same with RenderMLP's third output (holo_voxel_grid_implicit_function.py:125-129,265-269: the reference returns
 * features = cat(colour, view-point independent features) when feature_dim > 0 - the configuration of its own tests,
 * holo_diffusion/tests/test_voxel_grid_implicit_function.py:17-41):
 *   vp_features : (n_points, feature_dim) = LeakyReLU(Linear(hidden features)), or NULL (= holo_implicit_eval)
 * Workspace for both entries: holo_implicit_workspace_bytes. */
size_t holo_implicit_workspace_bytes(const HoloRenderer* r, int64_t n_points, int64_t pts_per_dir, int with_features);
int holo_implicit_eval_features(HoloRenderer* r, const float* grid, const float* pts, const float* dirs, int64_t n_points,
                                int64_t pts_per_dir, float* densities, float* colours, float* vp_features, void* workspace,
                                size_t workspace_bytes, void* stream);

/* Normals of the density field at world points.  Replaces RenderMLP.get_normals as used by
 * HoloVoxelGridImplicitFunction.forward with render_normals=True (holo_voxel_grid_implicit_function.py:131-145,
 * 249-263): normalize(d density / d point); the density net is affine up to its final LeakyReLU, so the gradient is
 * evaluated analytically from the trilinear corner values instead of by autograd.
 *   pts : (n_points, 3) world coordinates;  normals : (n_points, 3) */
int holo_implicit_normals(HoloRenderer* r, const float* grid, const float* pts, int64_t n_points, float* normals,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * View pooling: source-view feature maps -> voxel feature grid.  Replaces, fused in one kernel, the image_rgb branch
 * of HoloDiffusionModel.forward behind the image feature extractor (holo_diffusion_model.py:340-373):
 *   VolumeLocator.get_coord_grid, ViewPooler (PyTorch3D ViewSampler: NDC projection + bilinear ndc_grid_sample, masks = 1
 *   with masked_sampling false; AngleWeightedReductionFeatureAggregator with [AVG, STD], configs/apple.yaml:183-196;
 *   _get_point_to_source_camera_ray_dirs, custom_modules.py:279-334), pooled_feature_mapper (:113,368), tanh (:373).
 * The image feature extractor itself (ResNet34) is outside this library: its output maps are the input here.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* feats; /* device, (n_views, channels, height, width) fp32 NCHW: one entry of the extractor's dict */
  int32_t channels, height, width;
} HoloViewFeature;

typedef struct {
  int32_t resol;                   /* HoloDiffusionModel.resol */
  float volume_extent;             /* 8.0 */
  int32_t feature_size;            /* output channels of pooled_feature_mapper = HoloDiffusionModel.feature_size */
  float weight_by_ray_angle_gamma; /* 1.0 */
  float min_ray_angle_weight;      /* 0.1 */
  float projection_eps;            /* |z| clamp of transform_points, 1e-2 */
} HoloViewPoolCfg;

size_t holo_view_pool_workspace_bytes(const HoloViewPoolCfg* cfg, const HoloViewFeature* feats, int n_feats, int n_views);

/* cameras: the n_views SOURCE cameras (host array; view 0 is the reference direction of the angular weights);
 * mapper_weight (feature_size, 2 * sum channels) / mapper_bias (feature_size, may be NULL): pooled_feature_mapper, its
 * input ordered per feature key [AVG | STD] in the order of `feats`;  voxel_features: (1, feature_size, R, R, R). */
int holo_view_pool(HoloCtx* ctx, const HoloViewPoolCfg* cfg, const HoloViewFeature* feats, int n_feats,
                   const HoloCamera* cameras, int n_views, const float* mapper_weight, const float* mapper_bias,
                   float* voxel_features, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of holo_view_pool for a gradient on its output: what autograd computes in the reference behind
 * `tanh(pooled_feature_mapper(view_pooler(...)))` (holo_diffusion_model.py:358-373) when the encoder side is trained -
 * the chain continues from HoloDiffusionModel.training_backward's gradient of the clean grid.  Same inputs as the forward
 * (nothing of it is kept: the aggregation is recomputed), plus
 *   grad_voxel_features : (1, feature_size, R, R, R)
 * Outputs (each may be NULL):
 *   grad_feats[k]       : (n_views, channels_k, height_k, width_k) NCHW - the gradient of entry k of the extractor's dict
 *                         (bilinear scatter by atomic adds, like grid_sample's backward: summation order not fixed)
 *   grad_mapper_weight  : (feature_size, 2 * sum channels);  grad_mapper_bias : (feature_size)   (fixed summation order)
 * feature_size <= 32 (every released configuration).  The angular weights depend on the cameras only: no gradient. */
size_t holo_view_pool_backward_workspace_bytes(HoloCtx* ctx, const HoloViewPoolCfg* cfg, const HoloViewFeature* feats,
                                               int n_feats, int n_views);
int holo_view_pool_backward(HoloCtx* ctx, const HoloViewPoolCfg* cfg, const HoloViewFeature* feats, int n_feats,
                            const HoloCamera* cameras, int n_views, const float* mapper_weight, const float* mapper_bias,
                            const float* grad_voxel_features, float* const* grad_feats, float* grad_mapper_weight,
                            float* grad_mapper_bias, void* workspace, size_t workspace_bytes, void* stream);

/* The same entry with the reference's own LEARNT aggregator, MLPMeanFeatureAggregator (holo_diffusion/custom_modules.py:
 * 162-293 + _get_point_to_source_camera_ray_dirs :296-334; selected by configs/hydrant.yaml:184 and
 * old_base_config.yaml:205): per source view [sampled features | harmonic(ray direction)] -> two LazyLinear layers
 * (sample, mean over views) -> one Linear + LeakyReLU -> Linear; softmax over the views of output 0; then
 * pooled_feature_mapper + tanh (holo_diffusion_model.py:368-373).  Parameters by the reference's state-dict names below
 * `view_pooler.feature_aggregator.`: "_first_sampled.{weight,bias}", "_first_mean.{weight,bias}",
 * "_mlp.mlp.0.0.{weight,bias}", "_last.{weight,bias}", plus "pooled_feature_mapper.{weight,bias}".  Masks are all ones and
 * no view is excluded (holo_diffusion_model.py:114-116 forces both exclusions off; masked_sampling false). */
typedef struct HoloMlpMeanPooler HoloMlpMeanPooler;
typedef struct {
  int32_t resol;
  float volume_extent;
  int32_t feature_size;             /* rows of pooled_feature_mapper (<= 32) */
  int32_t n_hidden;                 /* 128 */
  int32_t dim_out;                  /* 128 */
  int32_t n_layers;                 /* 1 */
  int32_t n_harmonic_functions_ray; /* 3 */
  int32_t n_feats;                  /* feature maps, in the image feature extractor's dict order */
  int32_t channels[8];              /* channels of each map */
  float projection_eps;             /* |z| clamp of camera.transform_points (1e-2) */
} HoloMlpMeanCfg;
int holo_mlp_mean_create(HoloCtx* ctx, const HoloMlpMeanCfg* cfg, HoloMlpMeanPooler** out);
int holo_mlp_mean_destroy(HoloMlpMeanPooler* h);
int holo_mlp_mean_set_param(HoloMlpMeanPooler* h, const char* name, const void* dev_ptr, int ndim, const int64_t* shape,
                            void* stream);
/* Folds the affine stretches of the aggregator + mapper in float64 and uploads them (setup time: synchronises). */
int holo_mlp_mean_commit(HoloMlpMeanPooler* h, void* stream);
size_t holo_mlp_mean_workspace_bytes(const HoloMlpMeanPooler* h, const HoloViewFeature* feats, int n_feats, int n_views);
int holo_mlp_mean_pool(HoloMlpMeanPooler* h, const HoloViewFeature* feats, int n_feats, const HoloCamera* cameras,
                       int n_views, float* voxel_features, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of holo_mlp_mean_pool for a gradient on its output (ABI 4): the gradients autograd leaves on the aggregator's
 * parameters (custom_modules.py:179-196), on pooled_feature_mapper and on the source-view feature maps.  Nothing of the
 * forward is kept: the per-(voxel, view) rows are recomputed into the workspace (about 3 GB at 64^3 with four views).
 *   grad_voxel_features : (1, feature_size, R, R, R);   grad_feats[k] : (n_views, channels_k, height_k, width_k) or NULL
 * The parameter gradients are fetched by reference name with holo_mlp_mean_get_grad (the names of holo_mlp_mean_set_param). */
size_t holo_mlp_mean_backward_workspace_bytes(const HoloMlpMeanPooler* h, const HoloViewFeature* feats, int n_feats, int n_views);
int holo_mlp_mean_backward(HoloMlpMeanPooler* h, const HoloViewFeature* feats, int n_feats, const HoloCamera* cameras, int n_views,
                           const float* grad_voxel_features, float* const* grad_feats, void* workspace, size_t workspace_bytes,
                           void* stream);
int holo_mlp_mean_get_grad(HoloMlpMeanPooler* h, const char* name, float* out_dev, int64_t numel, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement helpers (bench.py): time `iters` back-to-back launches of the dominant kernels with
 * hipEvents recorded on `stream` (torch.cuda.Event only sees torch's current stream).
 * ------------------------------------------------------------------------------------------ */
int holo_event_timer_create(void** timer);
int holo_event_timer_start(void* timer, void* stream);
int holo_event_timer_stop(void* timer, void* stream, float* elapsed_ms); /* synchronises on the stop event */
int holo_event_timer_destroy(void* timer);

/* Time only the conv3d implicit-GEMM launches of the last-planned forward (batch as planned):
 * runs every conv op `iters` times, returns total milliseconds and total algorithmic FLOPs. */
int holo_unet_time_convs(HoloUnet* net, int batch, void* workspace, size_t workspace_bytes, int iters,
                         void* stream, float* total_ms, double* total_flops, int* n_launches);

/* Per-op variant of the above: every op of the planned forward is timed on its own (one untimed launch, then
 * `iters` back-to-back launches between two events on `stream`), in execution order.
 *   op     : 0 memset, 1 layout-in, 2 time-embed, 3 emb linears, 4 GroupNorm stats, 5 GroupNorm finalize,
 *            6 conv, 7 GEMM, 8 softmax, 9 flash attention, 10 layout-out
 *   conv ops: kernel 0 = per-tap gather kernel, 1 = LDS voxel-halo kernel, 2 = small-M weight-streaming kernel, 3 / 4 / 6 = the
 *            Winograd forms (depth / depth + height / all three axes), 5 = bf16 wide-tile kernel, 7 = streaming 1x1x1 kernel,
 *            8 = bf16 wide-tile kernel in its persistent wave-specialised form, 9 = stride-2 bf16 halo kernel,
 *            10 = qkv convolution fused with the bf16 attention's operand packing, 11 = streaming 1x1x1 convolution on bf16 storage;
 *            tile_depth / fused_skip / nsplit describe the variant (they select the template instantiation that
 *            rocprofv3 reports); ms includes the split-K reduce launch when nsplit > 1; flops = 2*MACs incl. the
 *            fused 1x1x1 skip.  Attention/GEMM ops report their 2*MACs as well; other ops report flops 0.
 * x / timesteps / y as in holo_unet_forward (y is overwritten).  Returns the op count in *n_ops (at most `cap`
 * entries are written). */
typedef struct {
  int32_t op, kernel, tile_depth, fused_skip, nsplit;
  int32_t cin, cout, out_dim, stride, upsample, ksz;
  float ms;
  double flops;          /* algorithmic: the reference's multiply-adds x 2 */
  double flops_executed; /* issued to the matrix pipe (kernel 3, the Winograd-in-depth conv, spends 2/3 of `flops`) */
} HoloOpTiming;
int holo_unet_time_ops(HoloUnet* net, int batch, const float* x, const int64_t* timesteps, float* y, void* workspace,
                       size_t workspace_bytes, int iters, void* stream, HoloOpTiming* out, int cap, int* n_ops);

#ifdef __cplusplus
}
#endif
#endif /* HOLO_ABI_H */
