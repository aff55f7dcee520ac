// holo_kernels.h — host-side launchers of the gfx950 kernels (definitions in kernels_*.hip).
// All pointers are device pointers; `stream` is a hipStream_t as void*.
#pragma once
#include <stddef.h>
#include <stdint.h>

// per-device context of the C ABI (holo_ctx_create)
struct HoloCtx {
  int device;
  int num_cus;
  int deterministic = 0;  // holo_ctx_set_deterministic: scatter-adds of the backward entries in fixed point (order-independent)
};

namespace holo {

// ---------------------------------------------------------------------------------------------
// conv3d as implicit GEMM on fp32 MFMA (kernels_conv.hip).
//   out[m][co] = bias[co] + residual[m][co] + sum_{tap,ci} W[tap][co][ci] * in'(m, tap, ci)
// with m = ((n*OD+od)*OH+oh)*OW+ow over channels-last (NDHWC) activations.
// in' = optional per-(n,channel) affine (GroupNorm folded with FiLM) + optional SiLU, applied while
// the tile is staged, zero padding applied AFTER the activation; optional nearest x2 upsample on load;
// optional virtual channel concat of two sources (UNet skip connection, unet.py:829).
// ---------------------------------------------------------------------------------------------
struct ConvParams {
  const float* src0;
  const float* src1;  // may be null
  int C0, C1;         // channels of src0 / src1; C0 % 32 == 0 when src1 != null; (C0+C1) % 4 == 0
  int N, ID, IH, IW;  // logical input dims (AFTER upsampling when ups=1)
  int ups;            // 1: sources have dims ID/2 x IH/2 x IW/2 and are read at (z>>1,y>>1,x>>1)
  int OD, OH, OW;
  int stride, pad, ksz;  // ksz = 3 (27 taps) or 1
  int Cout;
  const float* w;         // packed [ksz^3][CinP/32][CoutP/16][2][4][16][4] (repack_conv_weight_launch), zero padded:
                          // CinP = Cin rounded up to 32, CoutP = Cout rounded up to 64 (32 when Cout < 64)
  int CoutP, CinP;
  const float* coef;      // [N][Cin][2] = (a,b): x' = a*x+b ; null = identity
  int act;                // 1: SiLU after the affine (only with coef)
  const float* bias;      // [Cout] or null
  const float* residual;  // [M][Cout] or null
  float* out;             // [M][Cout]
  // optional fused 1x1x1 skip connection (halo kernel only): out += skip_w . [skip_src0 | skip_src1] + skip_bias
  const float* skip_src0;
  const float* skip_src1;
  int skip_C0, skip_C1;
  const float* skip_w;    // packed like w with one tap
  // bf16-multiply variant of the halo kernel (opt-in, holo_unet_set_compute_dtype): the same weights rounded to
  // bf16 (RNE), packed [ksz^3][CinP/32][CoutP/16][lane 64][8 bf16] = one 1 KB block of B operands of
  // v_mfma_f32_16x16x32_bf16 per (tap, chunk, 16-Cout slice).  Used when bf16 != 0 and the launch is on the halo path.
  // The buffer holds THREE such planes (hi, mid, lo: w = hi + mid + lo exactly); bf16 == 1 uses plane 0 only,
  // bf16 == 2 the fp32-accurate bf16x3 kernel (conv_halo_split_kernel) all three.
  const uint16_t* w_bf;
  const uint16_t* skip_w_bf;
  int bf16;
  int skip_CinP;
  const float* skip_bias;
  double* stats;          // optional: GroupNorm partial sums of `out`, [N][conv_stats_slabs(p)][Cout][2]
  float* partial;         // [nsplit][M][Cout] scratch when nsplit > 1
  int nsplit;             // split-K factor over (tap, cin-chunk) chunks
  int chunks_per_split;
  int skip_chunks_per_split;  // halo kernel with a fused skip: skip chunks are dealt evenly to the same splits
  int tz;                 // halo kernel tile depth (set by conv_plan)
  int grid_x;             // halo kernel: persistent workgroups along x (set by conv_plan)
  int stagger_ticks;      // halo kernel: phase offset (100 MHz ticks) of the workgroup in the odd slot of a CU
  unsigned long long* dbg;  // optional timeline probe (scripts/conv_timeline.cpp): [tile][8] {t_start, t_first_halo,
                            // t_loops_done, t_end (100 MHz wall clock), HW_ID, XCC_ID, 0, 0}; null in production
  int mode;               // set by conv_plan: 0 = per-tap gather kernel, 1 = LDS voxel-halo kernel, 2 = row-tile kernel,
                          // 3 = streaming 1x1x1 kernel (large grids, raw input), 4 = stride-2 bf16 halo kernel (kernels_conv_s2.hip),
                          // 5 = qkv convolution fused with the attention's operand packing, 6 = streaming 1x1x1 convolution on bf16
                          // storage (kernels_conv1x1_bf16.hip)
  // Winograd-in-depth form of the 128-voxel halo kernel (conv_wino_kernel): weights pre-transformed along kz,
  // U_xi = sum_kz G[xi][kz] w[kz], packed like w with 36 pseudo-taps xi*9 + ky*3 + kx; the fused skip as 2 pseudo-taps
  // (+w/2, -w/2).  Null = not prepared for this conv (the direct kernel runs).
  const float* w_wino;
  const float* skip_w_wino;
  // (z,y) Winograd form (conv_wino2_kernel): U = sum_kz,ky G[xi_z][kz] G[xi_y][ky] w[kz][ky][kx], 48 pseudo-taps
  // (xi_z*4 + xi_y)*3 + kx; the fused skip as 4 pseudo-taps (xi_z,xi_y in {1,2}^2: +-w/4)
  const float* w_wino2;
  const float* skip_w_wino2;
  int wino;               // set by conv_plan: 0 direct, 1 Winograd in depth, 2 Winograd in depth and height, 3 all three axes
  // F(2x2x2, 3x3x3) form (conv_wino3_kernel, kernels_conv3.hip): the 64 pseudo-taps in the wave's consumption order
  // (repack_conv_weight_wino3_launch); the fused skip as 8 signed copies of the 1x1x1 weight
  const float* w_wino3;
  const float* skip_w_wino3;
  // bf16 wide-tile kernel (conv_bf16t_kernel, 8x8x8 output tiles, v_mfma_f32_32x32x16_bf16): plane 3 of the bf16
  // weight buffer, packed [ksz^3][CinP/16][CoutP/32][lane 64][8 bf16] = one 1 KB block of B operands per
  // (tap, 16-channel chunk, 32-Cout slice); lane's 8 values are channels 8*(lane>>5) .. +7 of output channel lane&31
  const uint16_t* w_bft;
  const uint16_t* skip_w_bft;
  int bf16t;              // set by conv_plan: the launch runs on conv_bf16t_kernel
  int bf16p;              // set by conv_plan (with bf16t): ... on its persistent wave-specialised form, conv_bf16p_kernel
                          // (kernels_conv_bf16p.hip: grid_x workgroups of 8 waves, 4 consumers + 4 producers, no split-K)
  // bf16 STORAGE (compute mode bf16): the buffers behind these float* are bf16 (uint16_t) channels-last tensors;
  // coefficients, bias, statistics and split-K scratch stay fp32 / double
  int in_bf16;            // src0 / src1 / skip_src0 / skip_src1
  int res_bf16;           // residual
  int out_bf16;           // out
  // the qkv convolution of an AttentionBlock fused with the operand packing of the bf16 attention (kernels_conv1x1_bf16.hip;
  // set by the planner when the launch may take that form, conv_plan decides: mode 5): `out` is then NOT written
  uint16_t* qkv_q;        // bf16 [sample, head][T][CH], scaled by qkv_scale
  uint16_t* qkv_k;        // bf16 [sample, head][T][CH]
  uint16_t* qkv_vt;       // bf16 [sample, head][CH][T]
  float qkv_scale;        // CH^-1/2 * log2 e
  int qkv_T, qkv_CH, qkv_H;
  int qkv_sb, qkv_rows;   // set by conv_plan: 32-channel slices / rows per workgroup
};

// kernels_conv_s2.hip: the stride-2 3x3x3 convolution of a Downsample block on bf16 activation storage (raw input, 2 x 8 x 8
// output tiles x 64 output channels, GroupNorm statistics: one slab per tile)
bool conv_s2_bf16_supported(const ConvParams& p);
int conv_s2_bf16_launch(const ConvParams& p, void* stream);

// kernels_conv1x1_bf16.hip: the qkv convolution of an AttentionBlock (bf16 storage) writing the packed operands of
// flash_attn_bf16v2_kernel (ConvParams::qkv_*) instead of fp32 qkv
bool conv1x1_qkv_bf16_supported(const ConvParams& p);
void conv1x1_qkv_bf16_plan(ConvParams& p, int num_cus);
int conv1x1_qkv_bf16_launch(const ConvParams& p, void* stream);
// ... and the same streaming GEMM with a plain [M][Cout] bf16 output (+ bias, + residual, GroupNorm slabs: one per workgroup row block)
bool conv1x1_bf16_stream_supported(const ConvParams& p);
void conv1x1_bf16_stream_plan(ConvParams& p, int num_cus);
int conv1x1_bf16_stream_slabs(const ConvParams& p);
int conv1x1_bf16_stream_launch(const ConvParams& p, void* stream);

// kernels_conv3.hip
int64_t conv_wino3_weight_floats(int CoutP, int CinP, int src_taps);
int repack_conv_weight_wino3_launch(const float* w, float* out, int Cout, int Cin, int src_taps, int CoutP, int CinP,
                                    void* stream);
int conv_wino3_launch(const ConvParams& p, void* stream);
// kernels_conv_bf16p.hip
int conv_bf16p_launch(const ConvParams& p, void* stream);

// Picks split-K so that the grid fills the chip; returns bytes of `partial` scratch needed (0 if none).
size_t conv_plan(ConvParams& p, int num_cus);
int conv_launch(const ConvParams& p, void* stream);
int conv_stats_slabs(const ConvParams& p);
double conv_flops(const ConvParams& p);       // algorithmic (the reference's multiply-adds x 2)
double conv_exec_flops(const ConvParams& p);  // issued to the matrix pipe (differs for the Winograd-in-depth kernel)

// ---------------------------------------------------------------------------------------------
// batched GEMM on fp32 MFMA (kernels_gemm.hip):  C[b] = alpha * A[b] * B[b]^T
//   A[b][m][k] at A + b*sa + m*lda + k            (k contiguous)
//   B[b][n][k] at B + b*sb + n*ldb + k            (b_kmajor = 0, k contiguous)
//              at B + b*sb + k*ldb + n            (b_kmajor = 1, n contiguous)
//   C[b][m][n] at C + b*sc + m*ldc + n
// batch index b = b0*nb1 + b1 with separate strides for the two levels (sample, head).
// ---------------------------------------------------------------------------------------------
struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  int M, Nn, K;
  int lda, ldb, ldc;
  int nb0, nb1;
  int64_t sa0, sa1, sb0, sb1, sc0, sc1;
  int b_kmajor;
  float alpha;
};
int gemm_launch(const GemmParams& p, void* stream);

// Flash-style attention over the token-major qkv buffer [N][T][3C] (head-major channel order: q,k,v blocks of
// C/H channels per head, unet.py:448); out is token-major [N][T][C].  scale2 = 1/sqrt(head channels).
struct AttnParams {
  const float* qkv;
  float* out;
  int N, T, C, H;
  float scale2;
};
bool flash_attn_supported(int T, int head_channels);
int flash_attn_launch(const AttnParams& p, void* stream);
// bf16-product variant (shared K/V tiles in LDS, v_mfma_f32_16x16x32_bf16), for the opt-in bf16 mode
// Second form of the bf16 attention (bf16 storage mode, long sequences): a pre-pass rewrites qkv once per call as
// bf16 Q (pre-scaled) / K per head and V TRANSPOSED per head, so that the main kernel stages plain 16-byte rows;
// 64-key blocks, 64 queries per wave, v_mfma_f32_32x32x16_bf16, exp2-domain softmax; the key range is split across
// workgroups (and recombined) when one workgroup per 256 queries would leave half of the chip's wave slots empty.
// `work` (flash_attn_bf16v2_workspace_bytes) holds the packed operands and the split partials; out_bf16: `out` is bf16.
bool flash_attn_bf16v2_supported(int T, int head_channels);
size_t flash_attn_bf16v2_workspace_bytes(const AttnParams& p, int num_cus);
// packed != 0: the operands in `work` have been written already (the fused qkv convolution), the packing pre-pass is skipped;
// flash_attn_bf16v2_operands: where they go and the scale folded into Q
int flash_attn_bf16v2_launch(const AttnParams& p, void* work, int out_bf16, int num_cus, void* stream, int packed = 0);
void flash_attn_bf16v2_operands(const AttnParams& p, void* work, uint16_t** q, uint16_t** k, uint16_t** vt, float* qscale);

// in-place row softmax over `rows` rows of length `cols` (unet.py:453, fp32)
int softmax_rows_launch(float* s, int64_t rows, int cols, void* stream);

// ---------------------------------------------------------------------------------------------
// misc (kernels_misc.hip)
// ---------------------------------------------------------------------------------------------
int ncdhw_to_ndhwc_launch(const float* in, float* out, int N, int C, int64_t V, int tanh_flag, void* stream,
                          int out_bf16 = 0);  // out_bf16 / in_bf16 / x_bf16: the channels-last tensor is bf16
int ndhwc_to_ncdhw_launch(const float* in, float* out, int N, int C, int64_t V, void* stream, int in_bf16 = 0);
int f32_to_bf16_launch(const float* in, float* out_bf16, int64_t n, void* stream);  // element-wise, n % 8 == 0

// GroupNorm statistics, stage 1: partial[n][b][c] = (sum, sumsq) in double over voxel slab b (deterministic,
// no atomics).  gn_stats_geometry gives the slab count B(C, V) the buffers must be sized for.
void gn_stats_geometry(int C, int64_t V, int* n_blocks, int* vox_per_block);
int gn_stats_launch(const float* x, double* partial, int N, int C, int64_t V, void* stream, int x_bf16 = 0);

// stage 2: GroupNorm(32 groups, eps) folded to per-channel (a,b), optionally composed with FiLM
// (unet.py:248-250): y = GN(x)*(1+scale)+shift.  Channels [0,C0) use part0 (B0 slabs), [C0,C0+C1) part1.
int gn_finalize_launch(const double* part0, int C0, int B0, const double* part1, int C1, int B1, int N, int64_t V,
                       int groups, float eps, const float* gamma, const float* beta, const float* film,
                       int film_stride, int film_cout, float* coef, void* stream, float* moments = nullptr);
// moments (training forward): [N][C0 + C1][2] = (mean, rstd) of the channel's group, for the backward pass

// emb = Linear2(SiLU(Linear1(timestep_embedding(t, mc))));  writes silu(emb) (all consumers apply SiLU first:
// unet.py:199-205) and emb itself.
int time_embed_launch(const int64_t* t, int N, int mc, int ted, const float* w1, const float* b1, const float* w2,
                      const float* b2, float* emb, float* emb_silu, void* stream);

// out[n][r] = bias[r] + sum_k W[r][k] * in[n][k]      (all ResBlock emb_layers concatenated)
int rows_linear_launch(const float* in, const float* w, const float* bias, float* out, int N, int rows, int K,
                       void* stream);

int ddpm_step_launch(const float* tables, int T, const int64_t* timesteps, int batch, int64_t per, const float* x_t,
                     const float* model_out, const float* noise, int clip, float* sample, float* pred_xstart,
                     void* stream);
int ddpm_step_philox_launch(const float* tables, int T, const int64_t* timesteps, int batch, int64_t per, const float* x_t,
                            const float* model_out, uint64_t seed, uint64_t offset, int clip, float* sample,
                            float* pred_xstart, float* noise_out, int ncdhw_channels, void* stream);
int tanh_launch(const float* x, float* y, int64_t n, void* stream);
// dst = src for a SMALL caller-provided tensor, read with system-scope loads (holo_ld_sys, holo_common.h)
int copy_sys_launch(const float* src, float* dst, int64_t n, void* stream);
// Host -> device upload of a packed parameter image into a buffer that kernels have been reading (a re-commit after a
// parameter update): the bytes land in a staging buffer that only copy_sys_kernel ever reads (system-scope loads), which
// then writes `dst` like any kernel - so no kernel can meet a stale cached line of `dst` (holo_ld_sys).  Synchronises.
// Returns 0 or the hipError_t.
inline int upload_via_stage(float** stage, size_t* stage_floats, float* dst, const float* host, size_t n, void* stream) {
  if (*stage_floats < n) {
    if (*stage) (void)hipFree(*stage);
    *stage = nullptr;
    *stage_floats = 0;
    hipError_t e = hipMalloc((void**)stage, n * sizeof(float));
    if (e != hipSuccess) return (int)e;
    *stage_floats = n;
  }
  hipError_t e = hipMemcpyAsync(*stage, host, n * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  if (copy_sys_launch(*stage, dst, (int64_t)n, stream)) return -1;
  e = hipStreamSynchronize((hipStream_t)stream);
  return e == hipSuccess ? 0 : (int)e;
}
int clip_launch(const float* x, float* y, float lo, float hi, int64_t n, void* stream);

int repack_conv_weight_bf16_launch(const float* w, uint16_t* out, int Cout, int Cin, int taps, int CoutP, int CinP,
                                   void* stream);
// OIDHW [Cout][Cin][taps] -> zero padded MFMA-fragment-packed layout (see ConvParams::w)
// OIDHW 3x3x3 (src_taps = 27) -> 36 Winograd-in-depth pseudo-taps, or a 1x1x1 skip weight (src_taps = 1) -> its 2
// pseudo-taps; same packed layout as repack_conv_weight_launch
int repack_conv_weight_wino_launch(const float* w, float* out, int Cout, int Cin, int src_taps, int CoutP, int CinP,
                                   void* stream, int dims = 1);  // dims 2: (z,y) form, 48 / 4 pseudo-taps
int repack_conv_weight_launch(const float* w, float* out, int Cout, int Cin, int taps, int CoutP, int CinP,
                              void* stream);

// ---------------------------------------------------------------------------------------------
// renderer (kernels_render.hip)
// ---------------------------------------------------------------------------------------------
// packed RenderMLP (built by render_exec.cpp::holo_renderer_commit)
struct MlpParams {
  const float* w_feat;  // [Hd][C]   folded density-net rows 0..Hd-1 (hidden features)
  const float* b_feat;  // [Hd]
  const float* w_dens;  // [C]       folded density row
  float b_dens;
  const float* w_rad;   // [3][Hd]   radiance weights on the hidden features
  const float* u_rad;   // [3][C]    0.6 * W_eff[:Hd]^T w_rad[c]  (linear part of LeakyReLU folded)
  const float* w_dir;   // [3][27]   radiance weights on the direction embedding
  float b_rad[3];
  float k_rad[3];       // 0.6 * w_rad[c] . b_eff[:Hd]
  int Hd;
};

struct RenderKernelParams {
  const float* grid_cl;  // (R,R,R,C) channels-last voxel features
  int R, C;
  float half_extent;  // 0.5*(R-1)*voxel_size (VolumeLocator local->world scale)
  MlpParams mlp;
  // cameras of this launch: gridDim.y = n_cams frames are rendered by ONE launch (outputs of camera i at
  // rgb + i*3*H*W, depth + i*H*W, ...): a single 400^2 frame is only 1.6 waves of workgroups on 256 CUs, so
  // per-frame launches leave the chip half empty for the second half of every frame
  struct Cam {
    float Rm[9], T[3], focal[2], pp[2];
    float zmin, zmax;
  };
  static constexpr int MAX_CAMS = 32;
  Cam cams[MAX_CAMS];
  int n_cams;
  int H, W;
  float range_x, range_y;  // NDC half ranges
  int n_coarse, n_fine;
  float bg[3];
  float background_opacity;
  float pdf_eps;
  // persistent launch: n_tiles = n_cams * ceil(H*W/32) wave tiles, walked by gridDim.x * (waves per workgroup) resident
  // waves; xcd > 1: XCD-aware tile order (workgroup b is taken to sit on XCD b % xcd)
  int64_t n_tiles;
  int xcd;
  // scratch, one slot per RESIDENT wave: val_ws 64*32 float4 (sigma, rgb of the coarse samples); nrm_ws the same for
  // their normals (null: normals are not rendered)
  float* val_ws;
  float* nrm_ws;
  const float* dens_field;  // render2_kernel with normals: S[v] = w_dens . F[v] per voxel (density_field_launch); else null
  // outputs (per camera): CHW planes
  float* rgb;
  float* depth;
  float* mask;
  float* rgb_c;  // may be null
  float* depth_c;
  float* mask_c;
  float* nrm;    // rendered normals sum_i w_i n_i (n_cams,3,H,W), fine / coarse pass; may be null
  float* nrm_c;
  unsigned long long* dbg;  // optional per-slot phase clocks [slot][8] (HOLO_RENDER_TIMELINE=1); null in production
  int split3;  // 1: RenderMLP products on the bf16 matrix cores from an exact 3-term bf16 split (feature_size 32 only)
  int* tile_ctr;  // render2_kernel: 8 zeroed counters (one per XCD range) for the dynamic tile hand-out; null = static stride
  int tail_quads;  // render2_kernel, dynamic hand-out: this many LAST 4-ray tiles of every XCD range go out as single-ray items
  // training-mode rendering (n_rays > 0; render2_kernel only): every camera renders the SAME number of rays given as NDC
  // coordinates; outputs are (n_cams, 3, n_rays) / (n_cams, n_rays) planes.  Optional injected random streams (null =
  // deterministic): stratified coarse depths, stratified importance samples, density noise of both passes.
  struct Train {
    int n_rays;
    const float* xys;           // (n_cams, n_rays, 2) NDC x, y of every ray (PyTorch3D convention: +x left, +y up)
    const float* u_coarse;      // (n_cams, n_rays, n_coarse) uniforms in [0,1): stratified depths (_jiggle_within_stratas)
    const float* u_fine;        // (n_cams, n_rays, n_fine) uniforms: sample_pdf(det = False)
    const float* noise_coarse;  // (n_cams, n_rays, n_coarse) standard normals: density noise of the coarse pass
    const float* noise_fine;    // (n_cams, n_rays, n_coarse + n_fine) standard normals of the fine pass, in DEPTH ORDER
    float noise_std;            // density_noise_std_train
    float* z_merged;            // optional out (n_cams, n_rays, n_coarse + n_fine): the fine pass's depths, in depth order
    unsigned char* new_flags;   // optional out, same shape: 1 where the merged position holds an importance sample
  } train;
};

// ---- backward of the training-mode renderer (kernels_render_bwd.hip) ----
// per ray record (36 floats): origin 3 | direction 3 | W_dir e(dir) + b_rad 3 | harmonic embedding of the normalised direction 27
constexpr int RBWD_REC = 36;      // floats per ray record
constexpr int RBWD_REC_DIR = 3;   // offset of the direction
constexpr int RBWD_REC_RDIR = 6;  // offset of the radiance direction term
constexpr int RBWD_REC_EMB = 9;   // offset of the 27 embedding entries
struct RenderBwdRays {
  RenderKernelParams::Cam cams[RenderKernelParams::MAX_CAMS];
  int n_cams, n_rays;
  int64_t ray0;        // global index of the first ray of this camera group
  const float* xys;    // (all cameras, n_rays, 2)
  float* rays;         // (all rays, 36)
  const float* w_dir;  // [3][27]
  float b_rad[3];
};
// one chunk of whole rays; feature-major buffers have `ld` (= chunk capacity, a multiple of 64) columns
struct RenderBwdChunk {
  const float* grid_cl;
  float* ggrid_cl;
  int R, C;
  float half_extent;
  int Hd, Hp;
  int nm, n_coarse, rays_per_cam;
  int64_t ray0;      // global index of the chunk's first ray
  int n_rays_chunk;
  int64_t n, n_pad, ld;
  const float* rays;
  const float* z_merged;           // (all rays, nm) merged depths in depth order
  const unsigned char* new_flags;  // (all rays, nm) 1 = importance sample, 0 = coarse sample
  float* F;    // [n_pad][C]
  float* YT;   // [Hp][ld]
  float* AT;   // [Hp][ld]
  float* GFT;  // [C][ld]
  float4* val;
  float4* drad;
  float4* gval;
  float4* GR;
  double* tmp;    // [n][4]
  float* gr_ray;  // (all rays, 4)
  const float* be;     // [Hp]
  const float* w_rad;  // [3][Hd]
  const float* noise_fine;
  const float* noise_coarse;
  float noise_std;
  const float *g_rgb, *g_depth, *g_mask, *g_rgb_c, *g_depth_c, *g_mask_c;  // any may be null (= zero)
  float bg[3];
  float background_opacity;
  // deterministic mode (null otherwise): the scatter adds fixed-point integers into gfix [voxel][C], scaled from *gfix_max =
  // bits of max |GFT| of the chunk (rbwd_absmax_launch), and rbwd_fix_flush_launch adds the chunk's sums to ggrid_cl
  long long* gfix;
  uint32_t* gfix_max;
};
int rbwd_absmax_launch(const RenderBwdChunk& p, void* stream);
int rbwd_fix_flush_launch(const RenderBwdChunk& p, void* stream);
int rbwd_rays_launch(const RenderBwdRays& p, void* stream);
int rbwd_gather_launch(const RenderBwdChunk& p, void* stream);
int rbwd_point_fwd_launch(const RenderBwdChunk& p, void* stream);
int rbwd_composite_launch(const RenderBwdChunk& p, void* stream);
int rbwd_point_bwd_launch(const RenderBwdChunk& p, void* stream);
int rbwd_dir_grad_launch(const float* gr_ray, const float* rays, int64_t n_rays_total, float* out, void* stream);
int rbwd_rowsum_launch(const float* YT, int64_t ld, int64_t n, int rows, float* out, int accumulate, void* stream);
int rbwd_scatter_launch(const RenderBwdChunk& p, void* stream);
// dw[i] (+)= sum_s partial[s][i] (kernels_bwd.hip)
int partial_reduce_launch(const float* partial, float* dw, int64_t n, int splits, int accumulate, void* stream);

// stand-alone implicit function: densities[P], colours[P][3] at world points pts[P][3];
// direction of point i is dirs[i / pts_per_dir] (pts_per_dir = 1: one direction per point)
struct ImplicitEvalParams {
  const float* grid_cl;
  int R, C;
  float half_extent;
  MlpParams mlp;
  const float* pts;
  const float* dirs;
  float* rdir;  // scratch [ceil(n_points / pts_per_dir)][3]: radiance direction term per direction
  int64_t n_points;     // end of the point range of this launch
  int64_t point0;       // first point of this launch (implicit_points_launch); hidden rows are relative to it
  int64_t pts_per_dir;
  float* densities;
  float* colours;
  float* hidden;  // optional [n_points][Hd]: hidden features after the LeakyReLU (input of the feature head); null = off
};
int bias_leaky_launch(float* y, const float* bias, int64_t rows, int cols, void* stream);
// view pooling (kernels_viewpool.hip): source-view feature maps -> voxel feature grid
struct ViewPoolParams {
  static constexpr int MAX_VIEWS = 16, MAX_FEATS = 8, MAX_AGG = 512;
  struct Cam {
    float Rm[9], T[3], focal[2], pp[2], centre[3];
  };
  struct Feat {
    const float* data;  // (n_views, H, W, Cp) channels-last, Cp = C rounded up to 4
    int C, Cp, H, W;
    int quad0;          // first channel quad of this map in the concatenated quad list
    int out0;           // first aggregated feature of this key: [AVG (C) | STD (C)]
  };
  Cam cams[MAX_VIEWS];
  Feat feat[MAX_FEATS];
  int n_views, n_feats, n_quads;
  int R;
  float half_extent;
  float gamma, min_weight, proj_eps;
  int A, F;           // aggregated features (2 sum C), output features
  const float* wt;    // (A, F) transposed mapper weight
  const float* bias;  // (F) or null
  float* out;         // (1, F, R, R, R)
};
int view_pool_launch(const ViewPoolParams& p, void* stream);
// backward of view_pool_kernel (kernels_viewpool_bwd.hip): fwd as for the forward launch (feature maps channels-last in the
// workspace, transposed mapper weight), gout = d loss / d voxel_features (1, F, R, R, R)
struct ViewPoolBwdParams {
  ViewPoolParams fwd;
  const float* gout;
  float* gfeat[ViewPoolParams::MAX_FEATS];  // zeroed (n_views, H, W, Cp) gradient maps, or null per key
  int want_feats;
  float* partial;  // [n_wgs][A * F + F]
  float* dW;       // (F, A) or null
  float* dbias;    // (F) or null
  // deterministic mode (holo_ctx_set_deterministic; null otherwise): zeroed 64-bit fixed-point images of the gradient maps
  // and one word per map for the bits of its largest addend (the launch runs twice: measure, then add)
  long long* gfix[ViewPoolParams::MAX_FEATS];
  uint32_t* fix_max;  // [MAX_FEATS], zeroed
};
int view_pool_bwd_launch(const ViewPoolBwdParams& b, int n_wgs, void* stream);
int nhwc_pad_to_nchw_launch(const float* in, float* out, int n, int C, int Cp, int64_t HW, void* stream);
// deterministic mode: in (+)= value of the fixed-point image (binary point from *maxbits), the image zeroed again
int fix_flush_launch(long long* fix, const uint32_t* maxbits, float* out, int64_t n, void* stream);
// MLPMeanFeatureAggregator path (kernels_viewpool.hip): the folded aggregator + mapper (viewpool_exec.cpp); vp carries the
// views, the feature maps (quad0 = first quad in the kernel's padded channel order), R, F, proj_eps and the output
struct MlpMeanParams {
  ViewPoolParams vp;
  const float* a;    // [128][dp]  W1 Ws in the padded channel order
  const float* am;   // [128][dp]  W1 Wm
  const float* cb;   // [128]      W1 (bs + bm) + b1
  const float* g;    // [F][128]   M Wl
  const float* g0;   // [F]        M bl + mapper bias
  const float* l;    // [128]      Wl[0]
  float l0;          // bl[0]
  int dp;            // padded input width (multiple of 8): sum of the maps' padded channels + padded embedding
  int emb0;          // first column of the ray-direction embedding
  int n_harmonic;
};
int mlp_mean_pool_launch(const MlpMeanParams& p, int num_cus, void* stream);
// backward of the MLPMeanFeatureAggregator path (kernels_viewpool_bwd.hip): the row buffers live in the workspace,
// rows = view * P + voxel; NRp / Pp = row counts padded for the split-K products (the padding rows are zero)
struct MlpMeanBwdParams {
  MlpMeanParams fwd;
  const float* gout;  // (1, F, R, R, R)
  int FW;             // width of a DUL row: F values of du, dlogit at column F, zero padding to a multiple of 4
  int64_t NRp, Pp;
  // the row buffers hold ONE chunk of voxels [p0, p0 + Pc) of the Pall = R^3 (rows = view * Pc + local voxel): the workspace
  // does not grow with the grid (holo_mlp_mean_backward walks the chunks, the parameter gradients add up in chunk order)
  int64_t p0, Pc, Pall;
  float *X, *MEAN, *CM, *PRE, *H, *U, *DUL, *DULT, *DPRET, *DC, *DCT, *DX, *DCA;
  float* gfeat[ViewPoolParams::MAX_FEATS];
  long long* gfix[ViewPoolParams::MAX_FEATS];  // deterministic mode, as in ViewPoolBwdParams (per chunk of voxels)
  uint32_t* fix_max;
};
int mm_bwd_step_launch(const MlpMeanBwdParams& b, int step, void* stream);
int mm_colsum_launch(const float* src, int64_t rows, int cols, int ld, float* partial, int n_blocks, float* out, void* stream,
                     int accumulate = 0);
int mm_sum_partials_launch(const float* partial, int S, int64_t n, float* out, void* stream, int accumulate = 0);
int nchw_to_nhwc_pad_launch(const float* in, float* out, int n, int C, int Cp, int64_t HW, void* stream);
int transpose_small_launch(const float* in, float* out, int rows, int cols, void* stream);
// ---------------------------------------------------------------------------------------------
// backward of the denoiser (kernels_bwd.hip)
// ---------------------------------------------------------------------------------------------
struct WgradParams {   // dW[co][ci][tap] = sum_m gy[m][co] act(coef . x)[m + tap][ci]; geometry as ConvParams
  const float* gy;     // [M_out][Cout]
  const float* src0;
  const float* src1;   // virtual concat (may be null); C0 % 32 == 0 then
  int C0, C1;
  int N, ID, IH, IW, ups, OD, OH, OW, stride, pad, ksz, ntaps;
  int Cout;
  const float* coef;   // [N][Cin][2] or null
  int act;
  float* partial;      // [splits][Cout][Cin][ntaps] scratch (wgrad_partial_bytes)
};
int wgrad_splits(const WgradParams& p, int num_cus);
size_t wgrad_partial_bytes(const WgradParams& p, int num_cus);
int conv_wgrad_launch(const WgradParams& p, float* dw, int accumulate, int num_cus, void* stream);
int flip_transpose_weight_launch(const float* in, float* out, int Co, int Ci, int T, void* stream);
int weight_tco_ci_launch(const float* in, float* out, int Co, int Ci, int T, void* stream);
size_t colsum_scratch_bytes(int C);
int colsum_launch(const float* g, int64_t M, int C, double* scratch, float* out, int accumulate, void* stream);
struct GnBwdParams {   // GroupNorm32 (+ FiLM) (+ SiLU) backward over the virtual concat [x0 | x1]
  const float* x0;
  const float* x1;
  int C0, C1, N;
  int64_t V;
  const float* ga;     // [N][V][C0 + C1] gradient w.r.t. the activated tensor
  const float* coef;   // forward (a, b)
  const float* mom;    // forward (mean, rstd) per channel
  const float* gamma;
  const float* beta;
  const float* film;   // [N][film_stride]: scale at +c, shift at +film_cout + c (null: no FiLM)
  int film_stride, film_cout;
  int act;
  double* part;        // scratch (gn_bwd_scratch_bytes)
  float* grp;          // scratch [N][Cin][2]
  float* dgamma;
  float* dbeta;
  float* dfilm;        // [N][film_stride] gradient of the FiLM rows (null without FiLM)
  int acc_params;
  float* gx0;          // gradient w.r.t. x0 / x1 (raw tensors)
  float* gx1;
  int acc0, acc1;      // accumulate into gx0 / gx1 instead of overwriting
};
size_t gn_bwd_scratch_bytes(const GnBwdParams& p);
int gn_bwd_launch(const GnBwdParams& p, void* stream);
int add_launch(float* dst, const float* src, int64_t n, int accumulate, void* stream);
int split_cat_launch(const float* g, float* g0, float* g1, int64_t M, int C0, int C1, int acc0, int acc1, void* stream);
int sumpool2_launch(const float* gup, float* gin, int N, int R, int C, int accumulate, void* stream);
int zero_insert2_launch(const float* gy, float* out, int N, int Rs, int C, void* stream);  // (Rs = the coarse edge, out: (2 Rs)^3)
int conv_dgrad_s2_launch(const float* gy, const float* wt, float* gx, int N, int RI, int RO, int Ci, int Co, int accumulate,
                         void* stream);
int attn_ds_launch(const float* P, float* dP, int64_t rows, int cols, void* stream);
int transpose_launch(const float* in, float* out, int batch, int T, void* stream);
int film_bwd_launch(const float* dfilm, const float* embs, const float* w, float* dw, float* db, float* gembs, int N, int rows,
                    int K, void* stream);
int time_embed_bwd_launch(const int64_t* t, int N, int mc, int ted, const float* w1, const float* b1, const float* w2,
                          const float* b2, const float* gembs, float* dw1, float* db1, float* dw2, float* db2, void* stream);
int fill_launch(float* dst, float v, int64_t n, void* stream);

int implicit_eval_launch(const ImplicitEvalParams& p, void* stream);   // = dirs + points
int implicit_dirs_launch(const ImplicitEvalParams& p, void* stream);
int implicit_points_launch(const ImplicitEvalParams& p, void* stream);
int implicit_normals_launch(const ImplicitEvalParams& p, float* normals, void* stream);  // uses grid_cl, pts, n_points, mlp
int density_field_launch(const float* grid_cl, const float* w_dens, int C, int64_t nvox, float* out, void* stream);
int render_launch(const RenderKernelParams& p, void* stream, int n_workgroups);
int render_waves_per_wg(int C, int n_fine, int with_normals, int split3 = 0, int train = 0);
int render_rays_per_tile(int C, int n_fine, int with_normals, int split3, int train);

}  // namespace holo
