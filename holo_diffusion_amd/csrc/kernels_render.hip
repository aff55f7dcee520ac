// kernels_render.hip — the render side of the hot path.
//
// render_kernel: one fused kernel per rendered frame, replacing the PyTorch3D/Implicitron render path of
// the reference (chunk loop + ~60 ATen launches per chunk, 63 chunks per 400^2 frame):
//   ray generation   NDCMultinomialRaysampler/_xy_to_ray_bundle + AdaptiveRaySampler bounds
//                    (invoked at holo_diffusion_model.py:442-448; configs/apple.yaml:135-146)
//   voxel fetch      VolumeLocator.world_to_local_coords + F.grid_sample(bilinear, zeros, align_corners)
//                    (holo_voxel_grid_implicit_function.py:204-225)
//   RenderMLP        holo_voxel_grid_implicit_function.py:107-129 + custom_modules.py:133-160.  The density
//                    net has no activation between its layers (custom_modules.py:108-112 attaches the
//                    LeakyReLU to the last layer only), so render_exec.cpp folds it to ONE affine map
//                    hidden = W_eff f + b_eff in float64; the kernel evaluates that map on the matrix cores
//   composite        EmissionAbsorptionRaymarcher (holo_multipass_ea.py:96-100)
//   resampling       RayPointRefiner + sample_pdf (det.) + sort (holo_multipass_ea.py:116): inverse CDF of the
//                    coarse weights; the sort of [coarse | new] depths is a merge of two sorted lists
//   fine pass        holo_multipass_ea.py:117-123.  The reference re-evaluates the 64 coarse points inside its
//                    128-point fine pass; those values are bit-identical to the coarse pass, so the kernel
//                    evaluates only the 64 new points and composites the merged list from stored values
//                    (128 instead of 192 implicit-function evaluations per ray, same result)
// implicit_eval_kernel: the stand-alone HoloVoxelGridImplicitFunction.forward (densities, colours) for
// arbitrary points (holo_voxel_grid_implicit_function.py:182-269, incl. the pts_3d entry the reference's
// tests use).
//
// Mapping: wave = 32 rays/points; lanes l and l+32 share item (l&31) and split the
// feature channels in halves (lane half h owns channels [h*CH, h*CH+CH)), which is exactly the k index of
// v_mfma_f32_32x32x2_f32.  Per step a wave evaluates 32 samples:
//   D[hidden][sample] += W_eff[hidden][ch] * f[sample][ch]   (A = weights from LDS, B = the lane's own
//   interpolated features) so every lane ends up with 16 hidden units of ITS sample per 32-row tile; the
//   LeakyReLU + 3x256 radiance dot product is then lane-local and only one cross-half shuffle is needed.
// LeakyReLU_0.2(h) = 0.6 h + 0.4 |h|: the 0.6 h part of the radiance sum is linear in f and is folded on the
// host into a C-vector per colour (u_rad); only sum_rows 0.4 w |h| is accumulated per row (one FMA with a
// free |.| input modifier per colour) — two VALU ops per row fewer than max(h, 0.2h) * w.
// The voxel grid is channels-last so a corner is CH*4 contiguous bytes per lane; the 33.5 MB grid stays
// resident in the 256 MB Infinity Cache / L2 across the frame.  All eight corner fetches of a sample are
// issued unconditionally from clamped addresses (zeros padding = zero weight).
#include <stdlib.h>
#include <string.h>

#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {
namespace {

constexpr int HD = 256;  // RenderMLP.dnet_hidden_dim
constexpr int NTILE = HD / 32;
constexpr int MAXC = 64;  // max coarse samples per ray held in LDS (cdf rows)

__device__ __forceinline__ float lin_space(float start, float end, float step, int i, int steps) {
  // torch.linspace: symmetric evaluation around the midpoint
  return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}
__device__ __forceinline__ float leaky02(float v) { return fmaxf(v, 0.2f * v); }
__device__ __forceinline__ float fast_sigmoid(float v) { return holo_rcp(1.f + __expf(-v)); }
// x / h for a wave-uniform h with rh = 1/h precomputed: one Newton step on the product (Markstein): correctly rounded
// except in rare double-rounding cases (1 ulp), three VALU operations instead of the eleven of an IEEE division
__device__ __forceinline__ float div_uniform(float x, float h, float rh) {
  const float q = x * rh;
  return fmaf(fmaf(-q, h, x), rh, q);
}

// LDS image of the packed MLP shared by the block
// SP (fp32-accurate bf16x3 split, opt-in): W_eff is held as three bf16 planes (hi, mid, lo; 64-byte rows of 4
// XOR-swizzled 16-byte slots, slot = 2*kstep + lane half) and a lane half owns channels 16s + 8g + (0..7) of every
// 16-channel k-step s, which is the operand layout of v_mfma_f32_32x32x16_bf16.
template <int CH, bool SP = false>
struct MlpLds {
  static constexpr int C = 2 * CH;
  static constexpr int LDW = C + 4;
  static constexpr int WPLANE = HD * (C / 2);  // words per bf16 plane
  float w[SP ? 3 * WPLANE : HD * LDW];  // W_eff rows (hidden features): padded fp32 rows, or the three bf16 planes
  float bias[NTILE * 2 * 16];     // b_feat in D-row order: [tile][lane half][r] (the 16 rows a lane owns per tile)
  float wr[3][NTILE * 2 * 16];    // 0.4 * w_rad[colour] in the same order
  float u[2 * 4 * CH];            // per half: {w_dens, u_rad0, u_rad1, u_rad2} slices of CH floats (lane channel order)
};

// channel owned by lane half h at position k of its CH channels
template <int CH, bool SP>
__device__ __forceinline__ int lane_channel(int h, int k) {
  return SP ? 16 * (k >> 3) + 8 * h + (k & 7) : h * CH + k;
}

template <int CH, bool SP>
__device__ __forceinline__ void stage_mlp(MlpLds<CH, SP>& L, const MlpParams& m, int tid, int nthr = 256) {
  constexpr int C = 2 * CH;
  constexpr int LDW = C + 4;
  if (SP) {
    static_assert(!SP || CH == 16, "the split renderer is built for feature_size 32");
    uint32_t* wb = reinterpret_cast<uint32_t*>(L.w);
    constexpr int WPL = HD * (C / 2);
    for (int i = tid; i < HD * (C / 2); i += nthr) {  // one channel pair per step
      const int row = i / (C / 2), cp = i - row * (C / 2);
      const int c = 2 * cp;
      const float x0 = m.w_feat[row * C + c], x1 = m.w_feat[row * C + c + 1];
      uint32_t h, md, l;
      split3_pair(x0, x1, h, md, l);
      const int slot = 2 * (c >> 4) + ((c >> 3) & 1);  // k-step, lane half
      const int word = row * (C / 2) + ((slot ^ ((row >> 2) & 3)) << 2) + ((c & 7) >> 1);
      wb[word] = h;
      wb[WPL + word] = md;
      wb[2 * WPL + word] = l;
    }
  } else {
    for (int i = tid; i < HD * (C / 4); i += nthr) {
      const int row = i / (C / 4), c4 = i - row * (C / 4);
      *reinterpret_cast<float4*>(L.w + row * LDW + c4 * 4) = *reinterpret_cast<const float4*>(m.w_feat + row * C + c4 * 4);
    }
  }
  for (int i = tid; i < NTILE * 2 * 16; i += nthr) {
    const int r = i & 15, h = (i >> 4) & 1, t = i >> 5;
    const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    L.bias[i] = m.b_feat[row];
    L.wr[0][i] = 0.4f * m.w_rad[0 * HD + row];
    L.wr[1][i] = 0.4f * m.w_rad[1 * HD + row];
    L.wr[2][i] = 0.4f * m.w_rad[2 * HD + row];
  }
  for (int i = tid; i < 2 * 4 * CH; i += nthr) {
    const int k = i % CH, j = (i / CH) & 3, h = i / (4 * CH);
    const int ch = lane_channel<CH, SP>(h, k);
    L.u[i] = (j == 0) ? m.w_dens[ch] : m.u_rad[(j - 1) * C + ch];
  }
}

// direction term of the radiance layer (constant along a ray): b_rad + k_rad + W_dir . harmonic(normalize(d))
__device__ __forceinline__ void dir_term(const MlpParams& m, float dx, float dy, float dz, float (&rdir)[3]) {
  const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);  // F.normalize eps
  const float dn[3] = {dx / nrm, dy / nrm, dz / nrm};
  float e[27];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const float arg = dn[a] * (float)(1 << f);
      e[a * 4 + f] = sinf(arg);
      e[12 + a * 4 + f] = cosf(arg);
    }
    e[24 + a] = dn[a];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float s = m.b_rad[c] + m.k_rad[c];
    for (int j = 0; j < 27; ++j) s = fmaf(m.w_dir[c * 27 + j], e[j], s);
    rdir[c] = s;
  }
}

// One sample of the implicit function for the lane's item: world point -> raw density, colour.
// grid is the wave-uniform channels-last grid, lane_off the lane half's first channel (lane_channel(lh, 0)).
// NRM: also the normal of the density field at the point, normalize(d density / d point) (RenderMLP.get_normals,
// holo_voxel_grid_implicit_function.py:131-145): the density pre-activation is affine in the interpolated features, so
// its gradient follows from the eight per-corner scalars s_c = w_dens . F_c and the derivatives of the trilinear
// weights (zero for corners outside the grid, like grid_sample's backward), times LeakyReLU'.
// HID: also store the sample's hidden features AFTER the LeakyReLU (RenderMLP's `mlp_feats`, the input of the
// view-point independent feature head) to hid[0..HD) - the lane's 16 rows of a tile are four groups of 4 consecutive rows.
// NRM = 2: the eight per-corner scalars s_c = w_dens . F_c are read from `sfield` (one float per voxel, density_field_kernel:
// they depend on the grid and the folded density row only, not on the ray) instead of being formed per sample - 8 scalar
// loads in place of 4 CH packed FMAs + the density row from LDS, and both lane halves hold the complete value (no exchange).
template <int CH, bool SP = false, int NRM = 0, bool HID = false>
__device__ __forceinline__ void eval_point(const MlpLds<CH, SP>& L, const float* __restrict__ grid, uint32_t lane_off,
                                           int R, float Rm1, float half_extent, float b_dens, int li, int lh, float px, float py,
                                           float pz, const float (&rdir)[3], float& sigma, float& cr, float& cg,
                                           float& cb, float* nrm = nullptr, float* hid = nullptr,
                                           const float* __restrict__ sfield = nullptr) {
  constexpr int C = 2 * CH;
  constexpr int LDW = C + 4;
  float gx = 0.f, gy = 0.f, gz = 0.f;  // NRM: this lane half's part of d(pre-activation)/d(voxel index)
  // On gfx950 the fp32 MFMA runs on the same FMA lanes as ordinary vector instructions (tools/coexec_probe.cpp:
  // MFMA waves and VALU waves of one SIMD do not overlap), so every VALU instruction here costs MFMA time.  All
  // per-channel / per-row arithmetic is therefore written on register PAIRS (v_pk_fma_f32: two fmaf per instruction).
  f32x2 fv[CH / 2];  // interpolated features, channel pairs
#pragma unroll
  for (int k = 0; k < CH / 2; ++k) fv[k] = f32x2{0.f, 0.f};
  {
    const float rh = holo_rcp_exact(half_extent);  // wave-uniform
    const float lx = div_uniform(px, half_extent, rh), ly = div_uniform(py, half_extent, rh),
                lz = div_uniform(pz, half_extent, rh);
    const float ix = ((lx + 1.f) * 0.5f) * Rm1, iy = ((ly + 1.f) * 0.5f) * Rm1, iz = ((lz + 1.f) * 0.5f) * Rm1;
    const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
    // zeros padding as per-axis weights: a corner outside [0, R-1] gets weight 0.  All 8 corner fetches are
    // issued UNCONDITIONALLY from clamped addresses: a per-corner "if (inside) load" makes the compiler wait
    // for each load before issuing the next, i.e. eight serial L2 round trips per sample.
    const float wxa = (fx0 >= 0.f && fx0 <= Rm1) ? (fx0 + 1.f) - ix : 0.f;
    const float wxb = (fx0 >= -1.f && fx0 <= Rm1 - 1.f) ? ix - fx0 : 0.f;
    const float wya = (fy0 >= 0.f && fy0 <= Rm1) ? (fy0 + 1.f) - iy : 0.f;
    const float wyb = (fy0 >= -1.f && fy0 <= Rm1 - 1.f) ? iy - fy0 : 0.f;
    const float wza = (fz0 >= 0.f && fz0 <= Rm1) ? (fz0 + 1.f) - iz : 0.f;
    const float wzb = (fz0 >= -1.f && fz0 <= Rm1 - 1.f) ? iz - fz0 : 0.f;
    const int x0 = (int)fminf(fmaxf(fx0, -1.f), Rm1), y0 = (int)fminf(fmaxf(fy0, -1.f), Rm1),
              z0 = (int)fminf(fmaxf(fz0, -1.f), Rm1);
    const int xa = max(x0, 0), xb = min(x0 + 1, R - 1);
    const int ya = max(y0, 0), yb = min(y0 + 1, R - 1);
    const int za = max(z0, 0), zb = min(z0 + 1, R - 1);
    // NRM: derivative of the per-axis weights w.r.t. the voxel index (piecewise constant, zero outside the grid)
    const float dxa = (NRM && fx0 >= 0.f && fx0 <= Rm1) ? -1.f : 0.f;
    const float dxb = (NRM && fx0 >= -1.f && fx0 <= Rm1 - 1.f) ? 1.f : 0.f;
    const float dya = (NRM && fy0 >= 0.f && fy0 <= Rm1) ? -1.f : 0.f;
    const float dyb = (NRM && fy0 >= -1.f && fy0 <= Rm1 - 1.f) ? 1.f : 0.f;
    const float dza = (NRM && fz0 >= 0.f && fz0 <= Rm1) ? -1.f : 0.f;
    const float dzb = (NRM && fz0 >= -1.f && fz0 <= Rm1 - 1.f) ? 1.f : 0.f;
    int lhn = lh;
    if (NRM == 1) HOLO_LAUNDER(lhn);
    const float4* wdp = reinterpret_cast<const float4*>(L.u + lhn * 4 * CH);  // the lane half's slice of the density row
    float scf[8];  // NRM = 2: the corners' scalars, requested with the features
    if (NRM == 2) {
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) {
        const int xx = (corner & 1) ? xb : xa, yy = ((corner >> 1) & 1) ? yb : ya, zz = (corner >> 2) ? zb : za;
        scf[corner] = sfield[(uint32_t)((zz * R + yy) * R + xx)];
      }
    }
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
      const float w = ((dx ? wxb : wxa) * (dy ? wyb : wya)) * (dz ? wzb : wza);
      const int xx = dx ? xb : xa, yy = dy ? yb : ya, zz = dz ? zb : za;
      // 32-bit element offset from the wave-uniform grid pointer (a 128^3 x 64 grid is 2^27 elements)
      const float4* g = reinterpret_cast<const float4*>(grid + ((uint32_t)((zz * R + yy) * R + xx) * (uint32_t)C + lane_off));
      const f32x2 w2 = f32x2{w, w};
      f32x2 sc2 = f32x2{0.f, 0.f};
#pragma unroll
      for (int v = 0; v < CH / 4; ++v) {
        const float4 t = g[SP ? 4 * (v >> 1) + (v & 1) : v];  // SP: 8 channels of every 16-channel k-step
        fv[2 * v + 0] = pk_fma(w2, f32x2{t.x, t.y}, fv[2 * v + 0]);
        fv[2 * v + 1] = pk_fma(w2, f32x2{t.z, t.w}, fv[2 * v + 1]);
        if (NRM == 1) {
          const float4 wd = wdp[v];
          sc2 = pk_fma(f32x2{wd.x, wd.y}, f32x2{t.x, t.y}, sc2);
          sc2 = pk_fma(f32x2{wd.z, wd.w}, f32x2{t.z, t.w}, sc2);
        }
      }
      if (NRM) {
        const float sc = NRM == 2 ? scf[corner] : sc2.x + sc2.y;
        gx = fmaf(((dx ? dxb : dxa) * (dy ? wyb : wya)) * (dz ? wzb : wza), sc, gx);
        gy = fmaf(((dx ? wxb : wxa) * (dy ? dyb : dya)) * (dz ? wzb : wza), sc, gy);
        gz = fmaf(((dx ? wxb : wxa) * (dy ? wyb : wya)) * (dz ? dzb : dza), sc, gz);
      }
    }
  }
  // lane-local half dot products: density row and the linear (0.6 h) part of the three radiance sums, each as an
  // (even channel, odd channel) pair of partial sums
  f32x2 dp2 = f32x2{0.f, 0.f}, r0 = dp2, r1 = dp2, r2 = dp2;
  {
    int lhl = lh;
    HOLO_LAUNDER(lhl);  // reloaded per sample: hoisted out of the march loops these rows would pin 4*CH VGPRs
    const float4* up = reinterpret_cast<const float4*>(L.u + lhl * 4 * CH);
#pragma unroll
    for (int v = 0; v < CH / 4; ++v) {
      const float4 a = up[v], b = up[CH / 4 + v], c = up[2 * (CH / 4) + v], d = up[3 * (CH / 4) + v];
      dp2 = pk_fma(f32x2{a.x, a.y}, fv[2 * v], dp2);
      dp2 = pk_fma(f32x2{a.z, a.w}, fv[2 * v + 1], dp2);
      r0 = pk_fma(f32x2{b.x, b.y}, fv[2 * v], r0);
      r0 = pk_fma(f32x2{b.z, b.w}, fv[2 * v + 1], r0);
      r1 = pk_fma(f32x2{c.x, c.y}, fv[2 * v], r1);
      r1 = pk_fma(f32x2{c.z, c.w}, fv[2 * v + 1], r1);
      r2 = pk_fma(f32x2{d.x, d.y}, fv[2 * v], r2);
      r2 = pk_fma(f32x2{d.z, d.w}, fv[2 * v + 1], r2);
    }
  }
  // SP: the lane's features split exactly into three bf16 terms, packed as B operands [plane][k-step]
  float4 fb[3][SP ? CH / 8 : 1];
  if (SP) {
#pragma unroll
    for (int ks = 0; ks < CH / 8; ++ks) {
      uint32_t h[4], md[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split3_pair(fv[4 * ks + e].x, fv[4 * ks + e].y, h[e], md[e], l[e]);
      memcpy(&fb[0][ks], h, 16);
      memcpy(&fb[1][ks], md, 16);
      memcpy(&fb[2][ks], l, 16);
    }
  }
  // hidden features on the matrix cores, tile by tile
#pragma unroll 1
  for (int t = 0; t < NTILE; ++t) {
    // the LDS-resident weights are loop invariant across march steps: keep the compiler from hoisting
    // (and spilling) 8 tiles of operands out of the sample loops
    asm volatile("" ::: "memory");
    const int ro = (t * 2 + lh) * 16;  // this lane's 16 D rows of the tile
    f32x16 acc;  // starts at the bias of the lane's own rows (four 16-byte LDS reads straight into the tuple)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float4 b4 = *reinterpret_cast<const float4*>(L.bias + ro + 4 * v);
      acc[4 * v + 0] = b4.x;
      acc[4 * v + 1] = b4.y;
      acc[4 * v + 2] = b4.z;
      acc[4 * v + 3] = b4.w;
    }
    if (SP) {
      // six leading cross terms of every product on v_mfma_f32_32x32x16_bf16 (smallest first); one accumulator chain
      // per k-step, summed at the end, so that consecutive MFMAs are independent
      constexpr int WPL = HD * (C / 2);
      const int row = t * 32 + li;
      f32x16 acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
      float4 aw[3][2];  // [plane][k-step]: every plane's fragments are read once and used in 3 / 2 / 1 products
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const float* wp = L.w + pl * WPL + row * (C / 2);
        aw[pl][0] = *reinterpret_cast<const float4*>(wp + (((0 + lh) ^ ((row >> 2) & 3)) << 2));
        aw[pl][1] = *reinterpret_cast<const float4*>(wp + (((2 + lh) ^ ((row >> 2) & 3)) << 2));
      }
#pragma unroll
      for (int pr = 0; pr < 6; ++pr) {
        const int pa = (pr == 0) ? 2 : (pr == 2 || pr == 3) ? 1 : 0;  // A plane: lo, hi, mid, mid, hi, hi
        const int pb = (pr == 1) ? 2 : (pr == 2 || pr == 4) ? 1 : 0;  // B plane: hi, lo, mid, hi, mid, hi
        acc = mfma_bf16_32x32x16(aw[pa][0], fb[pb][0], acc);
        acc1 = mfma_bf16_32x32x16(aw[pa][1], fb[pb][SP ? 1 : 0], acc1);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += acc1[r];
    } else {
      const float4* ap = reinterpret_cast<const float4*>(L.w + (t * 32 + li) * LDW + lh * CH);
      float a[CH];
#pragma unroll
      for (int v = 0; v < CH / 4; ++v) {
        const float4 t4 = ap[v];
        a[4 * v + 0] = t4.x;
        a[4 * v + 1] = t4.y;
        a[4 * v + 2] = t4.z;
        a[4 * v + 3] = t4.w;
      }
#pragma unroll
      for (int k = 0; k < CH; ++k)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], (k & 1) ? fv[k >> 1].y : fv[k >> 1].x, acc, 0, 0, 0);
    }
    // the MFMA k index runs over both lane halves, so each lane now holds complete hidden units:
    if (HID && hid) {
#pragma unroll
      for (int v = 0; v < 4; ++v)  // D row of register 4v+e: t*32 + 8v + 4 lh + e
        *reinterpret_cast<float4*>(hid + t * 32 + 8 * v + 4 * lh) =
            make_float4(leaky02(acc[4 * v + 0]), leaky02(acc[4 * v + 1]), leaky02(acc[4 * v + 2]), leaky02(acc[4 * v + 3]));
    }
    // radiance sums  r_c += 0.4 w_c[row] |h[row]|  on row pairs
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float4 w0 = *reinterpret_cast<const float4*>(L.wr[0] + ro + 4 * v);
      const float4 w1 = *reinterpret_cast<const float4*>(L.wr[1] + ro + 4 * v);
      const float4 w2 = *reinterpret_cast<const float4*>(L.wr[2] + ro + 4 * v);
      const f32x2 h01 = f32x2{fabsf(acc[4 * v + 0]), fabsf(acc[4 * v + 1])};
      const f32x2 h23 = f32x2{fabsf(acc[4 * v + 2]), fabsf(acc[4 * v + 3])};
      r0 = pk_fma(f32x2{w0.x, w0.y}, h01, r0);
      r1 = pk_fma(f32x2{w1.x, w1.y}, h01, r1);
      r2 = pk_fma(f32x2{w2.x, w2.y}, h01, r2);
      r0 = pk_fma(f32x2{w0.z, w0.w}, h23, r0);
      r1 = pk_fma(f32x2{w1.z, w1.w}, h23, r1);
      r2 = pk_fma(f32x2{w2.z, w2.w}, h23, r2);
    }
  }
  float dpart = dp2.x + dp2.y, rp0 = r0.x + r0.y, rp1 = r1.x + r1.y, rp2 = r2.x + r2.y;
  dpart += __shfl_xor(dpart, 32);
  rp0 += __shfl_xor(rp0, 32);
  rp1 += __shfl_xor(rp1, 32);
  rp2 += __shfl_xor(rp2, 32);
  sigma = leaky02(dpart + b_dens);
  cr = fast_sigmoid(leaky02(rp0 + rdir[0]));
  cg = fast_sigmoid(leaky02(rp1 + rdir[1]));
  cb = fast_sigmoid(leaky02(rp2 + rdir[2]));
  if (NRM) {
    if (NRM == 1) {  // (the two lane halves hold the two channel halves of every corner's scalar)
      gx += __shfl_xor(gx, 32);
      gy += __shfl_xor(gy, 32);
      gz += __shfl_xor(gz, 32);
    }
    // chain rule: LeakyReLU'(pre-activation) * d(voxel index)/d(world point); both factors positive, kept so that
    // F.normalize's eps acts as in torch
    const float k = ((dpart + b_dens) > 0.f ? 1.f : 0.2f) * (0.5f * Rm1 / half_extent);
    gx *= k;
    gy *= k;
    gz *= k;
    const float nn = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);
    nrm[0] = gx / nn;
    nrm[1] = gy / nn;
    nrm[2] = gz / nn;
  }
}

// ---- the fused renderer -------------------------------------------------------------------------------------
// PERSISTENT kernel: one workgroup of NW waves per CU shares one LDS image of the RenderMLP; every wave is an
// independent worker that walks over "wave tiles" (32 consecutive rays of one frame), tile = slot, slot + nslots, ...
// over ALL frames of the launch, so the tail of a launch is one tile and the scratch is sized for the RESIDENT waves
// only, whatever the number of cameras.
//
// Per-wave state:
//   LDS   cz[32 rays][ZCAP+1]   coarse weights -> CDF (in place) -> importance-sampled depths (in place); row stride
//                               ZCAP+1 words: bank = (ray + j) mod 32, conflict-free for ray-parallel and j-parallel use
//   HBM   cval[nc][32] float4   (sigma_raw, r, g, b) of the coarse samples (+ cnrm[nc][32] normals with NRM), one
//                               32 KB slot per RESIDENT wave, re-used by every tile the wave processes
// The fine pass never stores its values: the merged (sorted) list [coarse | new] is composited INCREMENTALLY inside the
// fine loop - after fine sample k is evaluated, the coarse samples with depth <= z_k and then sample k itself are
// composited; the coarse values stream back through a two-deep register prefetch.
// Reference arithmetic kept: torch's CPU cumsum accumulates fp32 inputs in double (both the raymarcher's
// cumsum(delta * sigma) and sample_pdf's cdf), so both running sums are doubles here.
// waves per workgroup (= per CU): 12 (3 per SIMD, 168 VGPRs) for the released configurations; the normals variant and
// 64-feature grids need more registers per wave (2 per SIMD, 256 VGPRs); 128 new samples per ray need twice the LDS rows
template <int CH, int ZCAP, bool NRM>
constexpr int render_waves() {
  return (CH <= 16 && !NRM) ? (ZCAP <= 64 ? 12 : 6) : (ZCAP <= 64 ? 8 : 4);
}

template <int CH, bool SP, bool NRM, int ZCAP>
__global__ __launch_bounds__((64 * render_waves<CH, ZCAP, NRM>())) void render_kernel(RenderKernelParams p) {
  constexpr int NW = render_waves<CH, ZCAP, NRM>();
  constexpr int ZS = ZCAP + 1;
  // ONE LDS object with the RenderMLP image FIRST: the workgroup uses more than 64 KB of LDS and a ds_read carries a
  // 16-bit offset, so everything the evaluation loop reads (20 reads per 32-row tile) has to sit below 64 KB to be
  // addressed as base + immediate; the per-ray rows behind it are touched twice per evaluation
  struct Smem {
    MlpLds<CH, SP> mlp;
    float cz[NW * 32 * ZS];
  };
  __shared__ __attribute__((aligned(16))) Smem s_mem;
  MlpLds<CH, SP>& s_mlp = s_mem.mlp;
  float* const s_cz = s_mem.cz;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;

  stage_mlp<CH, SP>(s_mlp, p.mlp, tid, 64 * NW);
  __syncthreads();  // the only workgroup-level synchronisation: from here on the waves are independent workers

  float* const cz = s_cz + wave * (32 * ZS) + li * ZS;  // this ray's row
  float* const czw = s_cz + wave * (32 * ZS);           // the wave's rows
  const int npix = p.H * p.W;
  const int R = p.R;
  const float Rm1 = (float)(R - 1);
  const uint32_t lane_off = (uint32_t)lane_channel<CH, SP>(lh, 0);
  const int nc = p.n_coarse, nf = p.n_fine;
  const int nb = nc - 1;

  // worker slot and its scratch
  const int slot = (int)blockIdx.x * NW + wave;
  float4* const cval = reinterpret_cast<float4*>(p.val_ws) + (int64_t)slot * (MAXC * 32) + li;
  float4* const cnrm = NRM ? reinterpret_cast<float4*>(p.nrm_ws) + (int64_t)slot * (MAXC * 32) + li : nullptr;
  unsigned long long* dbg = p.dbg ? p.dbg + (int64_t)slot * 8 : nullptr;

  // tile order: XCD-aware when the launch asks for it (p.xcd > 1): workgroups are dealt to the XCDs round-robin, so
  // workgroup b sits on XCD b % xcd; each XCD works through its OWN contiguous range of tiles (neighbouring rays touch
  // neighbouring voxels: one L2 then sees one region of the volume instead of all eight seeing all of it).  Purely an
  // index permutation - correct whatever the real placement is.
  const int64_t ntiles = p.n_tiles;
  const int tiles_per_cam = (npix + 31) / 32;
  const int xcd = p.xcd > 1 && ((int)gridDim.x % p.xcd) == 0 ? p.xcd : 1;
  const int wg_in_x = (int)blockIdx.x / xcd, wgs_per_x = (int)gridDim.x / xcd, my_x = (int)blockIdx.x % xcd;
  const int64_t x_lo = ntiles * my_x / xcd, x_hi = ntiles * (my_x + 1) / xcd;

  for (int64_t t = x_lo + (int64_t)wg_in_x * NW + wave; t < x_hi; t += (int64_t)wgs_per_x * NW) {
    if (dbg && lane == 0) dbg[6] = HOLO_PROBE_CLOCK();
    // ---- ray setup (pytorch3d NDC grid: +x left, +y up; pixel centres)
    const int cam_i = (int)(t / tiles_per_cam);
    const RenderKernelParams::Cam& cam = p.cams[cam_i];
    const int ray = (int)(t - (int64_t)cam_i * tiles_per_cam) * 32 + li;
    const bool active = ray < npix;
    const int rr = active ? ray : npix - 1;
    const int py = rr / p.W, px = rr - py * p.W;
    const float hx = p.range_x / (float)p.W, hy = p.range_y / (float)p.H;
    const float minx = p.range_x - hx, maxx = -p.range_x + hx;
    const float miny = p.range_y - hy, maxy = -p.range_y + hy;
    const float xn = lin_space(minx, maxx, (maxx - minx) / (float)(p.W - 1), px, p.W);
    const float yn = lin_space(miny, maxy, (maxy - miny) / (float)(p.H - 1), py, p.H);
    const float dc0 = (xn - cam.pp[0]) / cam.focal[0], dc1 = (yn - cam.pp[1]) / cam.focal[1], dc2 = 1.0f;
    float org[3], dir[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float r0 = cam.Rm[j * 3 + 0], r1 = cam.Rm[j * 3 + 1], r2 = cam.Rm[j * 3 + 2];
      const float p1 = (dc0 - cam.T[0]) * r0 + (dc1 - cam.T[1]) * r1 + (dc2 - cam.T[2]) * r2;
      const float p2 = (2.f * dc0 - cam.T[0]) * r0 + (2.f * dc1 - cam.T[1]) * r1 + (2.f * dc2 - cam.T[2]) * r2;
      dir[j] = p2 - p1;
      org[j] = p1 - dir[j];
    }
    float rdir[3];
    dir_term(p.mlp, dir[0], dir[1], dir[2], rdir);
    const float zmin = cam.zmin, zmax = cam.zmax;
    const float zstep = (zmax - zmin) / (float)(nc - 1);
    auto zcoarse = [&](int i) { return lin_space(zmin, zmax, zstep, i, nc); };
    auto eval = [&](float z, float& sigma, float& cr, float& cg, float& cb, float* nv) {
      eval_point<CH, SP, NRM ? 1 : 0>(s_mlp, p.grid_cl, lane_off, R, Rm1, p.half_extent, p.mlp.b_dens, li, lh, org[0] + z * dir[0],
                              org[1] + z * dir[1], org[2] + z * dir[2], rdir, sigma, cr, cg, cb, nv);
    };
    const int64_t ob = (int64_t)cam_i * npix + ray;  // output pixel (1-channel planes); rgb planes at 3*cam*npix + c*npix

    if (dbg && lane == 0) dbg[7] = HOLO_PROBE_CLOCK();
    // ---- coarse pass (all rays of the wave in lock step): evaluate, composite, keep weights + values
    {
      double cum = 0.0;
      float Tr = 1.f, ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, anx = 0.f, any_ = 0.f, anz = 0.f;
      float zi = zcoarse(0);
      for (int i = 0; i < nc; ++i) {
        const float zn = (i + 1 < nc) ? zcoarse(i + 1) : 0.f;
        float sg, cr, cg, cb, nv[3];
        eval(zi, sg, cr, cg, cb, nv);
        if (lh == 0) {
          cval[i * 32] = make_float4(sg, cr, cg, cb);
          if (NRM) cnrm[i * 32] = make_float4(nv[0], nv[1], nv[2], 0.f);
        }
        const float delta = (i + 1 < nc) ? zn - zi : p.background_opacity;
        const float x = delta * fmaxf(sg, 0.f);
        const float cap = 1.f - __expf(-x);
        cum += (double)x;
        const float w = cap * Tr;
        ar = fmaf(w, cr, ar);
        ag = fmaf(w, cg, ag);
        ab = fmaf(w, cb, ab);
        ad = fmaf(w, zi, ad);
        if (NRM) {
          anx = fmaf(w, nv[0], anx);
          any_ = fmaf(w, nv[1], any_);
          anz = fmaf(w, nv[2], anz);
        }
        if (lh == 0) cz[i] = w;  // weights for now; turned into the cdf below
        Tr = 1.f - (1.f - __expf(-(float)cum));  // T = 1 - O, O = 1 - exp(-cumsum) as the raymarcher forms them
        zi = zn;
      }
      const float O = 1.f - __expf(-(float)cum);
      if (p.rgb_c && active && lh == 0) {
        float* o = p.rgb_c + (int64_t)cam_i * 3 * npix + ray;
        o[0 * (int64_t)npix] = ar + (1.f - O) * p.bg[0];
        o[1 * (int64_t)npix] = ag + (1.f - O) * p.bg[1];
        o[2 * (int64_t)npix] = ab + (1.f - O) * p.bg[2];
        p.depth_c[ob] = ad;
        p.mask_c[ob] = O;
        if (NRM && p.nrm_c) {
          float* on = p.nrm_c + (int64_t)cam_i * 3 * npix + ray;
          on[0 * (int64_t)npix] = anx;
          on[1 * (int64_t)npix] = any_;
          on[2 * (int64_t)npix] = anz;
        }
      }
    }

    // ---- weights[1:-1] + eps -> pdf -> cdf, in place in the ray's LDS row (RayPointRefiner / sample_pdf):
    // cdf has nb = nc-1 entries, cdf[0] = 0, cdf[j] = sum_{m<=j} (w[m]+eps)/S; the running sums are doubles like torch's
    if (dbg && lane == 0) dbg[1] += HOLO_PROBE_CLOCK() - dbg[7];
    if (lh == 0) {
      double S = 0.0;
      for (int m = 1; m < nc - 1; ++m) S += (double)(cz[m] + p.pdf_eps);
      const float Sf = (float)S;
      double run = 0.0;
      cz[0] = 0.f;
      for (int j = 1; j < nb; ++j) {
        run += (double)((cz[j] + p.pdf_eps) / Sf);
        cz[j] = (float)run;
      }
    }
    __threadfence();  // cval / cnrm of this tile are read back below (by both lane halves)
    HOLO_WAVE_SYNC();

    // ---- importance-sample depths of all 32 rays, cooperatively: for one ray at a time lane j holds cdf[j] and lane
    //      kk computes the inverse CDF at u_kk = linspace(0,1,nf)[kk] with a binary search over the lanes' values:
    //      searchsorted(right=True), then the lerp of sample_pdf.  The depths overwrite the ray's cdf row.
    {
      const float ustep = 1.0f / (float)(nf - 1);
      auto mid = [&](int i) {
        const float a0 = zcoarse(i), a1 = zcoarse(i + 1);
        return a0 - (a0 - a1) * 0.5f;  // torch.lerp(z[1:], z[:-1], 0.5)
      };
      for (int r0 = 0; r0 < 32; r0 += 8) {
        float cv[8];  // the CDF rows of 8 rays are read together
#pragma unroll
        for (int q = 0; q < 8; ++q) cv[q] = lane < nb ? czw[(r0 + q) * ZS + lane] : 3.0e38f;
        HOLO_WAVE_SYNC();  // every lane holds its CDF entries before the rows are overwritten
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = r0 + q;
          const float c = cv[q];
          for (int kb = 0; kb < nf; kb += 64) {
            const int kk = kb + lane;
            const float u = lin_space(0.f, 1.f, ustep, kk < nf ? kk : nf - 1, nf);
            int lo = 0, hi = nb;  // ind = #{j < nb : cdf[j] <= u}
#pragma unroll
            for (int it = 0; it < 7; ++it) {  // nb <= 63 < 2^6 (+1 closing step); every lane runs all steps (shuffles)
              const int md = (lo + hi) >> 1;
              const float cm = __shfl(c, md < nb ? md : nb - 1);
              const bool go = lo < hi && cm <= u;
              const bool stay = lo < hi && !(cm <= u);
              if (go) lo = md + 1;
              if (stay) hi = md;
            }
            const int ind = lo;
            const int below = ind - 1 > 0 ? ind - 1 : 0;
            const int above = ind < nb - 1 ? ind : nb - 1;
            const float cb_ = __shfl(c, below), ca_ = __shfl(c, above);
            float den = ca_ - cb_;
            if (den < p.pdf_eps) den = 1.f;
            const float tt = (u - cb_) / den;
            const float bb = mid(below), ba = mid(above);
            if (kk < nf) czw[r * ZS + kk] = bb + tt * (ba - bb);
          }
        }
      }
    }
    HOLO_WAVE_SYNC();

    if (dbg && lane == 0) dbg[2] += HOLO_PROBE_CLOCK() - dbg[7];
    // ---- fine pass, in lock step.  The reference re-evaluates the coarse points inside its 128-sample fine pass;
    //      those values are bit-identical to the coarse pass, so only the nf NEW points are evaluated, and the merged
    //      list (== torch.sort of the concatenated depths; a coarse sample goes first on a tie) is composited on the fly.
    {
      double cum = 0.0;
      float Tr = 1.f, ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, anx = 0.f, any_ = 0.f, anz = 0.f;
      // Every depth of the merged list is known before its sample is evaluated (the new depths sit in the LDS row, the
      // coarse ones are analytic), so a sample is composited the moment its values exist: its interval ends at
      // min(next coarse depth, next new depth).  No "pending sample" state is carried across the evaluations.
      auto emit = [&](float z, float delta, const float4& v, const float4& nv4) {
        const float x = delta * fmaxf(v.x, 0.f);
        const float cap = 1.f - __expf(-x);
        cum += (double)x;
        const float w = cap * Tr;
        ar = fmaf(w, v.y, ar);
        ag = fmaf(w, v.z, ag);
        ab = fmaf(w, v.w, ab);
        ad = fmaf(w, z, ad);
        if (NRM) {
          anx = fmaf(w, nv4.x, anx);
          any_ = fmaf(w, nv4.y, any_);
          anz = fmaf(w, nv4.z, anz);
        }
        Tr = 1.f - (1.f - __expf(-(float)cum));  // T = 1 - O, O = 1 - exp(-cumsum) as the raymarcher forms them
      };
      // coarse samples come back through a two-deep prefetch (q0 = entry ci, q1 = entry ci + 1)
      int ci = 0;
      float4 q0 = cval[0], q1 = cval[32];
      float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0;
      if (NRM) {
        m0 = cnrm[0];
        m1 = cnrm[32];
      }
      // composite coarse sample ci; `zlim` = depth of the next NEW sample (the coarse sample's interval ends at the
      // nearer of it and the next coarse depth), `more_fine` = such a sample exists
      auto pop_coarse = [&](float zlim, bool more_fine) {
        const float zc = zcoarse(ci);
        const bool more_coarse = ci + 1 < nc;
        float znext = more_coarse ? zcoarse(ci + 1) : zlim;
        if (more_fine) znext = fminf(znext, zlim);
        const float delta = (more_coarse || more_fine) ? znext - zc : p.background_opacity;
        emit(zc, delta, q0, m0);
        q0 = q1;
        const int nxt = ci + 2 < nc ? ci + 2 : nc - 1;
        q1 = cval[nxt * 32];
        if (NRM) {
          m0 = m1;
          m1 = cnrm[nxt * 32];
        }
        ++ci;
      };
      for (int kk = 0; kk < nf; ++kk) {
        const float zf = cz[kk];
        float sg, cr, cg, cb, nv[3];
        eval(zf, sg, cr, cg, cb, nv);
        for (;;) {  // wave-uniform loop, divergent body: every lane composites its coarse samples with depth <= zf
          const bool need = ci < nc && zcoarse(ci < nc ? ci : nc - 1) <= zf;
          if (!__any(need)) break;
          if (need) pop_coarse(zf, true);
        }
        const bool more_fine = kk + 1 < nf, more_coarse = ci < nc;
        float znext = more_fine ? cz[kk + 1 < nf ? kk + 1 : kk] : 0.f;
        if (more_coarse) {
          const float zc = zcoarse(ci < nc ? ci : nc - 1);
          znext = more_fine ? fminf(znext, zc) : zc;
        }
        emit(zf, (more_fine || more_coarse) ? znext - zf : p.background_opacity, make_float4(sg, cr, cg, cb),
             NRM ? make_float4(nv[0], nv[1], nv[2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f));
      }
      for (;;) {  // coarse samples behind the last new one
        const bool need = ci < nc;
        if (!__any(need)) break;
        if (need) pop_coarse(0.f, false);
      }
      const float O = 1.f - __expf(-(float)cum);
      if (active && lh == 0) {
        float* o = p.rgb + (int64_t)cam_i * 3 * npix + ray;
        o[0 * (int64_t)npix] = ar + (1.f - O) * p.bg[0];
        o[1 * (int64_t)npix] = ag + (1.f - O) * p.bg[1];
        o[2 * (int64_t)npix] = ab + (1.f - O) * p.bg[2];
        p.depth[ob] = ad;
        p.mask[ob] = O;
        if (NRM && p.nrm) {
          float* on = p.nrm + (int64_t)cam_i * 3 * npix + ray;
          on[0 * (int64_t)npix] = anx;
          on[1 * (int64_t)npix] = any_;
          on[2 * (int64_t)npix] = anz;
        }
      }
    }
    HOLO_WAVE_SYNC();  // the rows are rewritten by the next tile's coarse pass
    if (dbg && lane == 0) {
      dbg[3] += HOLO_PROBE_CLOCK() - dbg[7];
      dbg[0] += dbg[7] - dbg[6];
      dbg[4] += 1;
    }
  }
}

// =====================================================================================================================
// render2_kernel - the (ray, depth)-tiled form of the fused renderer (round 3).
//
// render_kernel above makes the 32 MFMA columns of a wave 32 RAYS marching in lock step; a ray's state - the (sigma, rgb)
// of its 64 coarse samples, which the merged fine-pass composite needs again - is then 1 KB x 32 rays per wave and has to
// go through a global scratch slot (328 of the 549 MB a frame moved through the fabric), and a frame is 5 000 32-ray
// tiles: 1.63 rounds on the 3 072 resident waves of a single-frame call.
// Here a wave tile is 4 RAYS and the 32 columns of one implicit-function evaluation are 4 rays x 8 CONSECUTIVE DEPTHS.
// The evaluation (trilinear fetch + folded RenderMLP on the matrix cores: eval_point, unchanged) no longer carries any
// per-ray state: it writes (sigma, r, g, b) of its sample into the wave's LDS rows and is done.  All per-ray arithmetic
// happens afterwards on the rows, one ray at a time with the 64 LANES AS DEPTH INDICES:
//   coarse composite   x_i = delta_i relu(sigma_i), inclusive wave scan of x in DOUBLE (torch's CPU cumsum accumulates fp32
//                      in double - measured), T_i = 1 - (1 - e^-S_{i-1}), w_i = (1 - e^-x_i) T_i, wave reductions
//   cdf                (w_i + eps) / S, second double scan, still in registers: lane j holds cdf[j]
//   inverse cdf        lane k binary-searches u_k over the lanes' cdf values with shuffles (as before) -> new depth row
//   merged composite   every sample (lane c: coarse c; lane k: new k, k + 64) finds its RANK in the merged order by binary
//                      search in the other list (ties: coarse first) and its interval end = the nearer of its own list's
//                      next depth and the other list's first depth behind it; (x, source id) are scattered by rank into
//                      the LDS row, three consecutive ranks per lane, double scan, reductions.
// Per-wave LDS: 4 rays x (64 coarse + ZF new) float4 values + depth / rank rows = 10.6 KB (ZF = 64): 8 waves + the
// 40 KB RenderMLP image per CU; the coarse values never leave the chip and a frame is 40 000 tiles (13 rounds).
// TRAIN (training-mode rendering): rays from an explicit NDC list instead of the full pixel grid
// (mask-sampled rays, configs/apple.yaml:135-146), stratified depths (PyTorch3D _jiggle_within_stratas) and stratified
// importance samples (sample_pdf det = False) from injected uniforms, density noise from injected normals added to the
// raw densities of each pass (holo_multipass_ea.py:87-91: the fine pass draws a NEW value for every one of its sorted
// points - indexed by merged rank here).
// =====================================================================================================================
// ---- wave-wide (64 lanes) scans and reductions.  On the device they run on DPP (data-parallel primitives: the source
// lane is selected inside the VALU instruction, no trip through the LDS crossbar that __shfl = ds_bpermute takes):
// Hillis-Steele inside the 16-lane rows (row_shr 1, 2, 4, 8; a lane without a source inside its row adds the `old` value
// 0), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.  A disabled or source-less lane receives
// `old` = 0 bits = +0.0.  The host emulation of the tests uses shuffles.
#ifndef HOLO_EMU
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double v) {
  unsigned long long b;
  memcpy(&b, &v, 8);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b & 0xffffffffull), CTRL, ROW_MASK, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xf, false);
  b = (unsigned long long)lo | ((unsigned long long)hi << 32);
  double r;
  memcpy(&r, &b, 8);
  return r;
}
#define HOLO_DPP_SCAN(v, F)        \
  v += F<0x111, 0xf>(v);           \
  v += F<0x112, 0xf>(v);           \
  v += F<0x114, 0xf>(v);           \
  v += F<0x118, 0xf>(v);           \
  v += F<0x142, 0xa>(v);           \
  v += F<0x143, 0xc>(v)
__device__ __forceinline__ double wave_scan_incl_d(double v, int) {
  HOLO_DPP_SCAN(v, dpp_d);
  return v;
}
__device__ __forceinline__ int wave_scan_incl_i(int v, int) {
  HOLO_DPP_SCAN(v, dpp_i);
  return v;
}
// totals: the inclusive scan's last lane, broadcast through a scalar register
__device__ __forceinline__ float wave_sum_f(float v) {
  HOLO_DPP_SCAN(v, dpp_f);
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 63));
}
__device__ __forceinline__ double wave_last_d(double incl) {  // value of lane 63
  unsigned long long b;
  memcpy(&b, &incl, 8);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b & 0xffffffffull), 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63);
  b = (unsigned long long)lo | ((unsigned long long)hi << 32);
  double r;
  memcpy(&r, &b, 8);
  return r;
}
// value of the previous lane (wave_shr:1), 0 for lane 0
__device__ __forceinline__ double wave_prev_d(double v, int) { return dpp_d<0x138, 0xf>(v); }
__device__ __forceinline__ int wave_prev_i(int v, int) { return dpp_i<0x138, 0xf>(v); }
#else
__device__ __forceinline__ double shfl_d(double v, int src) {
  unsigned long long b;
  memcpy(&b, &v, 8);
  const float lo = __shfl(__uint_as_float((uint32_t)(b & 0xffffffffull)), src);
  const float hi = __shfl(__uint_as_float((uint32_t)(b >> 32)), src);
  b = (unsigned long long)__float_as_uint(lo) | ((unsigned long long)__float_as_uint(hi) << 32);
  double r;
  memcpy(&r, &b, 8);
  return r;
}
__device__ __forceinline__ double wave_scan_incl_d(double v, int lane) {
  for (int d = 1; d < 64; d <<= 1) {
    const double o = shfl_d(v, lane >= d ? lane - d : lane);
    if (lane >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ int wave_scan_incl_i(int v, int lane) {
  for (int d = 1; d < 64; d <<= 1) {
    const int o = (int)__float_as_uint(__shfl(__uint_as_float((uint32_t)v), lane >= d ? lane - d : lane));
    if (lane >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ double wave_last_d(double incl) { return shfl_d(incl, 63); }
__device__ __forceinline__ double wave_prev_d(double v, int lane) {
  const double r = shfl_d(v, lane > 0 ? lane - 1 : 0);
  return lane > 0 ? r : 0.0;
}
__device__ __forceinline__ int wave_prev_i(int v, int lane) {
  const int r = (int)__float_as_uint(__shfl(__uint_as_float((uint32_t)v), lane > 0 ? lane - 1 : 0));
  return lane > 0 ? r : 0;
}
#endif

template <int ZF, bool NRM>
struct Render2Wave {
  float4 cval[4][64];               // coarse samples (sigma_raw, r, g, b)
  float4 fval[4][ZF];               // new samples
  float zf[4][ZF];                  // new depths, ascending
  unsigned char isnew[64 + ZF];     // merged composite: 1 where the position of the merged list holds a NEW sample
  float rd[4][4];                   // per ray: radiance direction term
  uint32_t cn[NRM ? 4 : 1][NRM ? 64 : 1];  // NRM: the samples' unit normals, octahedral snorm16 x 2 (r2_pack_normal)
  uint32_t fn[NRM ? 4 : 1][NRM ? ZF : 1];
};

// NRM (rendered normals, holo_multipass_ea.py:105-109; the released YAMLs' render_normals: true): a sample's normal
// normalize(d density / d point) (eval_point<NRM>) has to wait in LDS for the sample's composite weight like its colour.
// Three floats per sample would cost 6 KB per wave (12 -> 7 waves per CU); a unit vector is two numbers: the octahedral
// map (project onto |x| + |y| + |z| = 1, fold the lower half over the diagonals) in snorm16 x 2 = ONE word per sample,
// 2 KB per wave, 10 waves per CU.  Worst-case angular error 4e-5 (the composite's tolerance is 5e-4); the all-zero normal
// of a point outside the grid (F.normalize of a zero gradient) has its own code.
constexpr uint32_t R2_ZERO_NORMAL = 0x80008000u;
__device__ __forceinline__ uint32_t r2_pack_normal(const float (&n)[3]) {
  const float l1 = fabsf(n[0]) + fabsf(n[1]) + fabsf(n[2]);
  if (!(l1 > 0.f)) return R2_ZERO_NORMAL;
  const float inv = holo_rcp(l1);
  float px = n[0] * inv, py = n[1] * inv;
  if (n[2] < 0.f) {
    const float qx = (1.f - fabsf(py)) * (px >= 0.f ? 1.f : -1.f), qy = (1.f - fabsf(px)) * (py >= 0.f ? 1.f : -1.f);
    px = qx, py = qy;
  }
  const int ix = (int)rintf(fminf(fmaxf(px, -1.f), 1.f) * 32767.f), iy = (int)rintf(fminf(fmaxf(py, -1.f), 1.f) * 32767.f);
  return ((uint32_t)ix & 0xffffu) | ((uint32_t)iy << 16);
}
__device__ __forceinline__ void r2_unpack_normal(uint32_t c, float (&n)[3]) {
  const float px = (float)(short)(c & 0xffffu) * (1.f / 32767.f), py = (float)(short)(c >> 16) * (1.f / 32767.f);
  const float z = 1.f - fabsf(px) - fabsf(py);
  const float t = fmaxf(-z, 0.f);
  const float x = px + (px >= 0.f ? -t : t), y = py + (py >= 0.f ? -t : t);
  const float k = c == R2_ZERO_NORMAL ? 0.f : rsqrtf(x * x + y * y + z * z);
  n[0] = x * k, n[1] = y * k, n[2] = z * k;
}

// waves per workgroup (= per CU): the per-wave rows are 9.4 KB (64 new samples per ray) / 14.6 KB (128), + 2 KB with
// normals; with the 40 KB RenderMLP image of 32 grid features 12 / 8 waves fit (64 grid features: 73 KB image, 8 / 4 waves)
// NRM: built for <= 32 grid features and <= 64 new samples per ray (the BASELINE / released configurations; anything else
// renders its normals on the ray-per-column kernel, render_rays_per_tile).  10 waves per CU (what the LDS holds): with the
// per-corner scalars read from the density scalar field the evaluation fits the 168 registers of three waves per SIMD
// (84 bytes of scratch outside the MFMA loops; with the scalars formed per sample it was 580 bytes and 25 M rays/s).
// Measured at 400^2, 8 frames: 40.2 M rays/s on 10 waves, 39.0 M on 8 (HOLO_RENDER2_NRM_NW=8, the emulation's choice)
template <int CH, int ZF, bool NRM = false>
constexpr int render2_waves() {
  return NRM ? 10 : (ZF <= 64 ? (CH <= 16 ? 12 : 8) : (CH <= 16 ? 8 : 4));
}

template <int CH, int ZF, bool TRAIN, int NW, bool NRM = false>
__global__ __launch_bounds__((64 * NW)) void render2_kernel(RenderKernelParams p) {
  static_assert(!(TRAIN && NRM), "training-mode rendering returns no normals");
  struct Smem {
    MlpLds<CH, false> mlp;  // FIRST: everything the evaluation loop reads sits below 64 KB (16-bit ds_read offsets)
    Render2Wave<ZF, NRM> w[NW];
  };
  __shared__ __attribute__((aligned(16))) Smem s_mem;
  static_assert(sizeof(Smem) <= 160 * 1024, "render2_kernel: LDS budget");
  MlpLds<CH, false>& s_mlp = s_mem.mlp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  Render2Wave<ZF, NRM>& S = s_mem.w[wave];
  stage_mlp<CH, false>(s_mlp, p.mlp, tid, 64 * NW);
  __syncthreads();  // the only workgroup-level synchronisation: from here on the waves are independent workers

  const int npix = p.H * p.W;
  const int R = p.R;
  const float Rm1 = (float)(R - 1);
  const uint32_t lane_off = (uint32_t)lane_channel<CH, false>(lh, 0);
  const int nc = p.n_coarse, nf = p.n_fine, nb = nc - 1;
  const int64_t ntiles = p.n_tiles;
  const int rays_per_cam = TRAIN ? p.train.n_rays : npix;
  const int tiles_per_cam = (rays_per_cam + 3) / 4;
  const int xcd = p.xcd > 1 && ((int)gridDim.x % p.xcd) == 0 ? p.xcd : 1;
  const int wg_in_x = (int)blockIdx.x / xcd, wgs_per_x = (int)gridDim.x / xcd, my_x = (int)blockIdx.x % xcd;
  const int64_t x_lo = ntiles * my_x / xcd, x_hi = ntiles * (my_x + 1) / xcd;

  unsigned long long* dbg = p.dbg ? p.dbg + ((int64_t)blockIdx.x * NW + wave) * 8 : nullptr;  // development probe
  unsigned long long tk = 0;
  auto stamp = [&](int slot) {
    if (dbg && lane == 0) {
      const unsigned long long now = HOLO_PROBE_CLOCK();
      dbg[slot] += now - tk;
      tk = now;
    }
  };

  // tile hand-out: DYNAMIC when the launch supplies counters (one per XCD range): a wave takes the next tile of its range
  // when it is done with the previous one, so a launch ends within ONE tile time of its average instead of on a whole
  // round of tiles (40 000 tiles on 3 072 resident waves are 13.02 rounds: 14 with a static stride); otherwise static.
  // The END of a range is handed out in finer pieces (dynamic hand-out only): its last `tail_quads` 4-ray tiles go out as
  // four SINGLE-ray tiles each - 32 consecutive depths of one ray per evaluation instead of 8 depths of four rays, the same
  // evaluations per ray, a quarter of the tile time - so a launch ends within a quarter tile of its average.  It is what a
  // single-frame call (the reference's per-camera loop, flyaround.py:247-253) loses most: 13.02 rounds of 290 us tiles.
  // A work item v of the range: v < nq - kf: 4-ray tile x_lo + v; else single ray (v - (nq - kf)) & 3 of a tail tile.
  const int64_t nq = x_hi - x_lo;
  const int64_t kf = p.tile_ctr ? (p.tail_quads < nq ? (int64_t)p.tail_quads : nq) : 0;
  const int64_t nv = nq + 3 * kf;
  auto next_item = [&](int64_t prev) -> int64_t {
    if (!p.tile_ctr) return prev < 0 ? (int64_t)wg_in_x * NW + wave : prev + (int64_t)wgs_per_x * NW;
    int v = 0;
    if (lane == 0) v = atomicAdd(p.tile_ctr + my_x, 1);
    v = (int)__float_as_uint(__shfl(__uint_as_float((uint32_t)v), 0));
    return v;
  };
  for (int64_t v = next_item(-1); v < nv; v = next_item(v)) {
    if (dbg && lane == 0) tk = HOLO_PROBE_CLOCK();
    const bool fine_item = v >= nq - kf;
    const int64_t t = fine_item ? x_lo + (nq - kf) + ((v - (nq - kf)) >> 2) : x_lo + v;
    const int sh = fine_item ? 5 : 3;          // lanes li >> sh share a ray; 1 << sh consecutive depths per evaluation
    const int nr = 32 >> sh;                   // rays of this item: 4 or 1
    const int dstep = 1 << sh;
    const int rq = li >> sh, dq = li & (dstep - 1);  // evaluation mapping: ray of the item, depth inside the column group
    const int cam_i = (int)(t / tiles_per_cam);
    const RenderKernelParams::Cam& cam = p.cams[cam_i];
    const int ray0 = (int)(t - (int64_t)cam_i * tiles_per_cam) * 4 + (fine_item ? (int)((v - (nq - kf)) & 3) : 0);
    // ---- ray setup; every lane computes the ray it needs in each role (the wave pays per instruction, not per lane)
    auto ray_of = [&](int rr, float (&org)[3], float (&dir)[3]) {
      const int ray = min(ray0 + rr, rays_per_cam - 1);
      float xn, yn;
      if (TRAIN) {
        xn = holo_ld_sys(p.train.xys + ((int64_t)cam_i * rays_per_cam + ray) * 2 + 0);  // (small caller tensors: system-scope loads, holo_common.h)
        yn = holo_ld_sys(p.train.xys + ((int64_t)cam_i * rays_per_cam + ray) * 2 + 1);
      } else {
        const int py = ray / p.W, px = ray - py * p.W;
        const float hx = p.range_x / (float)p.W, hy = p.range_y / (float)p.H;
        const float minx = p.range_x - hx, maxx = -p.range_x + hx, miny = p.range_y - hy, maxy = -p.range_y + hy;
        xn = lin_space(minx, maxx, (maxx - minx) / (float)(p.W - 1), px, p.W);
        yn = lin_space(miny, maxy, (maxy - miny) / (float)(p.H - 1), py, p.H);
      }
      const float dc0 = (xn - cam.pp[0]) / cam.focal[0], dc1 = (yn - cam.pp[1]) / cam.focal[1], dc2 = 1.0f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float r0 = cam.Rm[j * 3 + 0], r1 = cam.Rm[j * 3 + 1], r2 = cam.Rm[j * 3 + 2];
        const float p1 = (dc0 - cam.T[0]) * r0 + (dc1 - cam.T[1]) * r1 + (dc2 - cam.T[2]) * r2;
        const float p2 = (2.f * dc0 - cam.T[0]) * r0 + (2.f * dc1 - cam.T[1]) * r1 + (2.f * dc2 - cam.T[2]) * r2;
        dir[j] = p2 - p1;
        org[j] = p1 - dir[j];
      }
    };
    float org[3], dir[3];
    ray_of(rq, org, dir);
    // radiance direction term of the 4 rays: lanes (ray = lane >> 4, q = lane & 15) share the 27 embedding entries
    {
      const int rr = lane >> 4, q = lane & 15;
      float o2[3], d2[3];
      ray_of(rr, o2, d2);
      const float nrm = fmaxf(sqrtf(d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2]), 1e-12f);  // F.normalize eps
      const float dn[3] = {d2[0] / nrm, d2[1] / nrm, d2[2] / nrm};
      float part[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int j = q + 16 * h;  // entry: [0,12) sin(d_a 2^f), [12,24) cos, [24,27) d_a; a = (j % 12) / 4, f = j % 4
        const int jj = j < 24 ? j % 12 : 0;
        const int a = j < 24 ? jj >> 2 : j - 24;
        const float da = a == 0 ? dn[0] : (a == 1 ? dn[1] : dn[2]);
        const float arg = da * (float)(1 << (jj & 3));
        const float e = j < 12 ? sinf(arg) : (j < 24 ? cosf(arg) : da);
        if (j < 27) {
#pragma unroll
          for (int c = 0; c < 3; ++c) part[c] = fmaf(p.mlp.w_dir[c * 27 + j], e, part[c]);
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) part[c] += __shfl_xor(part[c], d);
        if (q == 0) S.rd[rr][c] = part[c] + p.mlp.b_rad[c] + p.mlp.k_rad[c];
      }
    }
    HOLO_WAVE_SYNC();
    const float rdir[3] = {S.rd[rq][0], S.rd[rq][1], S.rd[rq][2]};
    const float zmin = cam.zmin, zmax = cam.zmax;
    const float zstep = (zmax - zmin) / (float)(nc - 1);
    auto zlin = [&](int i) { return lin_space(zmin, zmax, zstep, i, nc); };
    // coarse depth i of ray rr: the linspace, or (TRAIN) jittered inside its stratum: lower + (upper - lower) u with
    // lower = [z_0, mids], upper = [mids, z_last], mids = 0.5 (z_i + z_{i+1})   (PyTorch3D _jiggle_within_stratas)
    auto zcoarse_r = [&](int rr, int i) {
      const float zi = zlin(i);
      if (!TRAIN || !p.train.u_coarse) return zi;
      const float lo = i > 0 ? 0.5f * (zlin(i) + zlin(i - 1)) : zi;
      const float up = i + 1 < nc ? 0.5f * (zlin(i + 1) + zlin(i)) : zi;
      const int ray = min(ray0 + rr, rays_per_cam - 1);
      const float u = holo_ld_sys(p.train.u_coarse + ((int64_t)cam_i * rays_per_cam + ray) * nc + i);
      return lo + (up - lo) * u;
    };
    auto eval = [&](float z, float& sg, float& cr, float& cg, float& cb, uint32_t& ncode) {
      float nv[3] = {0.f, 0.f, 0.f};
      eval_point<CH, false, NRM ? 2 : 0>(s_mlp, p.grid_cl, lane_off, R, Rm1, p.half_extent, p.mlp.b_dens, li, lh, org[0] + z * dir[0],
                                         org[1] + z * dir[1], org[2] + z * dir[2], rdir, sg, cr, cg, cb, nv, nullptr, p.dens_field);
      ncode = NRM ? r2_pack_normal(nv) : 0u;
    };

    stamp(0);
    // ---- coarse evaluation: column groups of 8 (single-ray items: 32) consecutive depths
    for (int j0 = 0; j0 < nc; j0 += dstep) {
      const int i = min(j0 + dq, nc - 1);
      float sg, cr, cg, cb;
      uint32_t ncode;
      eval(zcoarse_r(rq, i), sg, cr, cg, cb, ncode);
      if (lh == 0 && j0 + dq < nc) {
        S.cval[rq][j0 + dq] = make_float4(sg, cr, cg, cb);
        if (NRM) S.cn[rq][j0 + dq] = ncode;
      }
    }
    HOLO_WAVE_SYNC();
    stamp(1);

    // ---- per ray: coarse composite, cdf, inverse cdf (lane = depth index)
    const float ustep = 1.0f / (float)(nf - 1);
#pragma unroll 1
    for (int rr = 0; rr < nr; ++rr) {
      const int ray = ray0 + rr;
      const bool active = ray < rays_per_cam;
      const int ic = min(lane, nc - 1);
      const float4 v = S.cval[rr][ic];
      const float zi = zcoarse_r(rr, ic);
      const float zn = lane + 1 < nc ? zcoarse_r(rr, min(lane + 1, nc - 1)) : 0.f;
      float sraw = v.x;
      if (TRAIN && p.train.noise_coarse)
        sraw += p.train.noise_std * holo_ld_sys(p.train.noise_coarse + ((int64_t)cam_i * rays_per_cam + min(ray, rays_per_cam - 1)) * nc + ic);
      const float delta = lane + 1 < nc ? zn - zi : p.background_opacity;
      const float x = lane < nc ? delta * fmaxf(sraw, 0.f) : 0.f;
      const double Sx = wave_scan_incl_d((double)x, lane);
      const double Sprev = wave_prev_d(Sx, lane);
      const float Tr = lane > 0 ? 1.f - (1.f - __expf(-(float)Sprev)) : 1.f;  // T = 1 - O as the raymarcher forms them
      const float w = lane < nc ? (1.f - __expf(-x)) * Tr : 0.f;
      const float O = 1.f - __expf(-(float)wave_last_d(Sx));
      if (p.rgb_c) {
        const float ar = wave_sum_f(w * v.y), ag = wave_sum_f(w * v.z), ab = wave_sum_f(w * v.w), ad = wave_sum_f(w * zi);
        if (lane == 0 && active) {
          const int64_t ob = (int64_t)cam_i * rays_per_cam + ray;
          float* o = p.rgb_c + (int64_t)cam_i * 3 * rays_per_cam + ray;
          o[0 * (int64_t)rays_per_cam] = ar + (1.f - O) * p.bg[0];
          o[1 * (int64_t)rays_per_cam] = ag + (1.f - O) * p.bg[1];
          o[2 * (int64_t)rays_per_cam] = ab + (1.f - O) * p.bg[2];
          p.depth_c[ob] = ad;
          p.mask_c[ob] = O;
        }
        if (NRM && p.nrm_c) {  // rendered normals of the coarse pass: sum_i w_i n_i
          float nn[3];
          r2_unpack_normal(S.cn[rr][ic], nn);
          const float anx = wave_sum_f(w * nn[0]), any_ = wave_sum_f(w * nn[1]), anz = wave_sum_f(w * nn[2]);
          if (lane == 0 && active) {
            float* on = p.nrm_c + (int64_t)cam_i * 3 * rays_per_cam + ray;
            on[0 * (int64_t)rays_per_cam] = anx;
            on[1 * (int64_t)rays_per_cam] = any_;
            on[2 * (int64_t)rays_per_cam] = anz;
          }
        }
      }
      // weights[1:-1] + eps -> pdf -> cdf (nb = nc-1 entries, cdf[0] = 0); running sums in double like torch's
      const float a = (lane >= 1 && lane <= nc - 2) ? w + p.pdf_eps : 0.f;
      const float Sf = (float)wave_last_d(wave_scan_incl_d((double)a, lane));
      const float pdf = (lane >= 1 && lane <= nc - 2) ? a / Sf : 0.f;
      const double run = wave_scan_incl_d((double)pdf, lane);
      const float c = lane < nb ? (lane == 0 ? 0.f : (float)run) : 3.0e38f;
      // bin mids of this ray (TRAIN: of the jittered depths): lerp(z[1:], z[:-1], 0.5)
      auto mid = [&](int i) {
        const float a0 = zcoarse_r(rr, i), a1 = zcoarse_r(rr, i + 1);
        return a0 - (a0 - a1) * 0.5f;
      };
      for (int kb = 0; kb < nf; kb += 64) {
        const int kk = kb + lane;
        const int kc = kk < nf ? kk : nf - 1;
        float u = lin_space(0.f, 1.f, ustep, kc, nf);
        if (TRAIN && p.train.u_fine)  // sample_pdf(det = False): u ~ U[0,1) per sample (unsorted; the merge sorts)
          u = holo_ld_sys(p.train.u_fine + ((int64_t)cam_i * rays_per_cam + min(ray, rays_per_cam - 1)) * nf + kc);
        int lo = 0, hi = nb;  // ind = #{j < nb : cdf[j] <= u}  (searchsorted right = True)
#pragma unroll
        for (int it = 0; it < 7; ++it) {
          const int md = (lo + hi) >> 1;
          const float cm = __shfl(c, md < nb ? md : nb - 1);
          const bool go = lo < hi && cm <= u;
          const bool stay = lo < hi && !(cm <= u);
          if (go) lo = md + 1;
          if (stay) hi = md;
        }
        const int ind = lo;
        const int below = ind - 1 > 0 ? ind - 1 : 0;
        const int above = ind < nb - 1 ? ind : nb - 1;
        const float cb_ = __shfl(c, below), ca_ = __shfl(c, above);
        float den = ca_ - cb_;
        if (den < p.pdf_eps) den = 1.f;
        const float tt = (u - cb_) / den;
        const float bb = mid(below), ba = mid(above);
        if (kk < nf) S.zf[rr][kk] = bb + tt * (ba - bb);
      }
      if (TRAIN && p.train.u_fine) {
        // stratified importance samples arrive unsorted: odd-even transposition sort of the row inside the wave (the
        // deterministic case is already ascending: the inverse cdf of an ascending u)
        HOLO_WAVE_SYNC();
        for (int pass = 0; pass < nf; ++pass) {
          for (int kb = 0; kb < nf; kb += 128) {
            const int i0 = kb + 2 * lane + (pass & 1);
            if (i0 + 1 < nf) {
              const float a0 = S.zf[rr][i0], a1 = S.zf[rr][i0 + 1];
              if (a1 < a0) {
                S.zf[rr][i0] = a1;
                S.zf[rr][i0 + 1] = a0;
              }
            }
          }
          HOLO_WAVE_SYNC();
        }
      }
    }
    HOLO_WAVE_SYNC();
    stamp(2);

    // ---- evaluation of the new samples
    for (int j0 = 0; j0 < nf; j0 += dstep) {
      const int k = min(j0 + dq, nf - 1);
      float sg, cr, cg, cb;
      uint32_t ncode;
      eval(S.zf[rq][k], sg, cr, cg, cb, ncode);
      if (lh == 0 && j0 + dq < nf) {
        S.fval[rq][j0 + dq] = make_float4(sg, cr, cg, cb);
        if (NRM) S.fn[rq][j0 + dq] = ncode;
      }
    }
    HOLO_WAVE_SYNC();
    stamp(3);

    // ---- per ray: composite of the merged list [coarse | new] in depth order (== torch.sort of the concatenation; a
    //      coarse sample goes first on a tie)
    const int nm = nc + nf;
#pragma unroll 1
    for (int rr = 0; rr < nr; ++rr) {
      const int ray = ray0 + rr;
      const bool active = ray < rays_per_cam;
      const float* zrow = S.zf[rr];
      // positions of the NEW samples in the merged order: rank of new sample k = k + #{coarse c : zc <= z_k}.  The coarse
      // depths are a linspace, so the count is arithmetic (one floor + a fix-up against the exact linspace values);
      // with jittered coarse depths (TRAIN) it is a binary search.
      const bool jitter = TRAIN && p.train.u_coarse != nullptr;
      for (int i = lane; i < (nm + 3) / 4; i += 64) reinterpret_cast<uint32_t*>(S.isnew)[i] = 0u;
      HOLO_WAVE_SYNC();
      for (int kb = 0; kb < nf; kb += 64) {
        const int k = kb + lane;
        if (k < nf) {
          const float zk = zrow[k];
          int b;
          if (jitter) {
            int lo = 0, hi = nc;  // #{c : zc <= zk}
            while (lo < hi) {
              const int md = (lo + hi) >> 1;
              if (zcoarse_r(rr, md) <= zk) lo = md + 1; else hi = md;
            }
            b = lo;
          } else {
            b = (int)fminf(fmaxf(floorf((zk - zmin) / zstep) + 1.f, 0.f), (float)nc);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
              if (b > 0 && zlin(b - 1) > zk) --b;
              if (b < nc && zlin(b) <= zk) ++b;
            }
          }
          S.isnew[k + b] = 1;
        }
      }
      HOLO_WAVE_SYNC();
      // three consecutive positions per lane: which are new, how many new ones lie before (integer wave scan), hence
      // the sample behind every position; its interval ends at the NEXT position's depth
      int f[3], cnt = 0;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int q = 3 * lane + e;
        f[e] = q < nm ? (int)S.isnew[q] : 0;
        cnt += f[e];
      }
      int nbefore = wave_scan_incl_i(cnt, lane) - cnt;
      float xs[3], zz[4];
      float4 vv[3];
      uint32_t ncd[3] = {0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int q = 3 * lane + e;
        const int qc = q < nm ? q : nm - 1;
        const int id = f[e] ? min(nbefore, nf - 1) : min(max(qc - nbefore, 0), nc - 1);
        vv[e] = f[e] ? S.fval[rr][id] : S.cval[rr][id];
        if (NRM) ncd[e] = f[e] ? S.fn[rr][id] : S.cn[rr][id];
        zz[e] = f[e] ? zrow[id] : zcoarse_r(rr, id);
        nbefore += f[e];
      }
      zz[3] = __shfl(zz[0], lane < 63 ? lane + 1 : lane);  // first depth of the next lane
      if (TRAIN && p.train.z_merged && active) {  // the merged list of the ray, for the backward pass (kernels_render_bwd.hip)
        const int64_t mb = ((int64_t)cam_i * rays_per_cam + ray) * nm;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const int q = 3 * lane + e;
          if (q < nm) {
            p.train.z_merged[mb + q] = zz[e];
            p.train.new_flags[mb + q] = (unsigned char)f[e];
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int q = 3 * lane + e;
        float sraw = vv[e].x;
        if (TRAIN && p.train.noise_fine)  // a fresh draw per sorted point of the fine pass
          sraw += p.train.noise_std *
                  holo_ld_sys(p.train.noise_fine + ((int64_t)cam_i * rays_per_cam + min(ray, rays_per_cam - 1)) * nm + (q < nm ? q : nm - 1));
        const float dl = q + 1 < nm ? zz[e + 1] - zz[e] : p.background_opacity;
        xs[e] = q < nm ? dl * fmaxf(sraw, 0.f) : 0.f;
      }
      const double l3 = (double)xs[0] + (double)xs[1] + (double)xs[2];
      const double incl = wave_scan_incl_d(l3, lane);
      double run = wave_prev_d(incl, lane);  // sum of all x before this lane's first position
      float ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, anx = 0.f, any_ = 0.f, anz = 0.f;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int q = 3 * lane + e;
        const float Tr = q > 0 ? 1.f - (1.f - __expf(-(float)run)) : 1.f;
        const float w = q < nm ? (1.f - __expf(-xs[e])) * Tr : 0.f;
        ar = fmaf(w, vv[e].y, ar);
        ag = fmaf(w, vv[e].z, ag);
        ab = fmaf(w, vv[e].w, ab);
        ad = fmaf(w, zz[e], ad);
        if (NRM) {
          float nn[3];
          r2_unpack_normal(ncd[e], nn);
          anx = fmaf(w, nn[0], anx);
          any_ = fmaf(w, nn[1], any_);
          anz = fmaf(w, nn[2], anz);
        }
        run += (double)xs[e];
      }
      const float O = 1.f - __expf(-(float)wave_last_d(incl));
      ar = wave_sum_f(ar);
      ag = wave_sum_f(ag);
      ab = wave_sum_f(ab);
      ad = wave_sum_f(ad);
      if (NRM && p.nrm) {
        anx = wave_sum_f(anx);
        any_ = wave_sum_f(any_);
        anz = wave_sum_f(anz);
      }
      if (lane == 0 && active) {
        const int64_t ob = (int64_t)cam_i * rays_per_cam + ray;
        float* o = p.rgb + (int64_t)cam_i * 3 * rays_per_cam + ray;
        o[0 * (int64_t)rays_per_cam] = ar + (1.f - O) * p.bg[0];
        o[1 * (int64_t)rays_per_cam] = ag + (1.f - O) * p.bg[1];
        o[2 * (int64_t)rays_per_cam] = ab + (1.f - O) * p.bg[2];
        p.depth[ob] = ad;
        p.mask[ob] = O;
        if (NRM && p.nrm) {
          float* on = p.nrm + (int64_t)cam_i * 3 * rays_per_cam + ray;
          on[0 * (int64_t)rays_per_cam] = anx;
          on[1 * (int64_t)rays_per_cam] = any_;
          on[2 * (int64_t)rays_per_cam] = anz;
        }
      }
      HOLO_WAVE_SYNC();  // the flag row is rewritten for the next ray
    }
    stamp(5);
    if (dbg && lane == 0) dbg[4] += 1;
  }
}

// S[v] = w_dens . F[v]: the folded density row applied to every voxel of the channels-last grid (one thread per voxel) - the
// per-corner scalars of the analytic normal (eval_point<NRM = 2>).  33.5 MB read, 1 MB written, once per render call.
__global__ __launch_bounds__(256) void density_field_kernel(const float* __restrict__ grid_cl, const float* __restrict__ w_dens,
                                                            int C, int64_t nvox, float* __restrict__ out) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  const float4* g = reinterpret_cast<const float4*>(grid_cl + v * C);
  const float4* w = reinterpret_cast<const float4*>(w_dens);
  // the same pairing as the in-kernel form (even / odd channel partial sums), so both forms round alike
  float se = 0.f, so = 0.f;
  for (int q = 0; q < C / 4; ++q) {
    const float4 t = g[q], a = w[q];
    se = fmaf(a.x, t.x, se);
    so = fmaf(a.y, t.y, so);
    se = fmaf(a.z, t.z, se);
    so = fmaf(a.w, t.w, so);
  }
  out[v] = se + so;
}

// radiance direction term per ray direction (one thread per direction)
__global__ __launch_bounds__(256) void dir_term_kernel(MlpParams m, const float* __restrict__ dirs, int64_t n_dirs,
                                                       float* __restrict__ rdir_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_dirs) return;
  float rdir[3];
  dir_term(m, dirs[i * 3 + 0], dirs[i * 3 + 1], dirs[i * 3 + 2], rdir);
  rdir_out[i * 3 + 0] = rdir[0];
  rdir_out[i * 3 + 1] = rdir[1];
  rdir_out[i * 3 + 2] = rdir[2];
}

// (densities, colours) for arbitrary points: grid-stride over groups of 128 points per block.  HID: the hidden features
// of every point go to p.hidden [n_points][HD] (input of the view-point independent feature head).
// (128 input features: the LDS image of the folded MLP is 141 KB, one workgroup per CU)
template <int CH, bool HID>
__global__ __launch_bounds__(256, (CH <= 32 ? 2 : 1)) void implicit_eval_kernel(ImplicitEvalParams p) {
  __shared__ __attribute__((aligned(16))) MlpLds<CH, false> s_mlp;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  stage_mlp<CH, false>(s_mlp, p.mlp, tid);
  __syncthreads();
  const float Rm1 = (float)(p.R - 1);
  const uint32_t lane_off = (uint32_t)(lh * CH);
  const int64_t ngroups = (p.n_points - p.point0 + 127) / 128;  // points [point0, n_points)
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t i = p.point0 + g * 128 + wave * 32 + li;
    const bool active = i < p.n_points;
    const int64_t ii = active ? i : p.n_points - 1;
    const float px = p.pts[ii * 3 + 0], py = p.pts[ii * 3 + 1], pz = p.pts[ii * 3 + 2];
    const int64_t di = ii / p.pts_per_dir;
    const float rdir[3] = {p.rdir[di * 3 + 0], p.rdir[di * 3 + 1], p.rdir[di * 3 + 2]};
    float sg, cr, cg, cb;
    eval_point<CH, false, 0, HID>(s_mlp, p.grid_cl, lane_off, p.R, Rm1, p.half_extent, p.mlp.b_dens, li, lh, px, py, pz,
                                      rdir, sg, cr, cg, cb, nullptr, (HID && active) ? p.hidden + (i - p.point0) * HD : nullptr);
    if (active && lh == 0) {
      p.densities[i] = sg;
      p.colours[i * 3 + 0] = cr;
      p.colours[i * 3 + 1] = cg;
      p.colours[i * 3 + 2] = cb;
    }
  }
}


// Normals of the density field at arbitrary points (RenderMLP.get_normals, holo_voxel_grid_implicit_function.py:
// 131-145, 249-263): normalize(d density / d point), density = LeakyReLU(w_dens . f(p) + b_dens) with f the trilinear
// fetch.  The density row is affine in the features (the folded density net), so the gradient is analytic: with
// s_c = w_dens . F_c the scalar of corner c,  d/dx (sum_c w_c(p) s_c) follows from the derivatives of the per-axis
// trilinear weights (zero for corners outside the grid, like grid_sample's backward).  One lane pair per point.
template <int CH>
__global__ __launch_bounds__(256) void implicit_normals_kernel(ImplicitEvalParams p, float* __restrict__ normals) {
  constexpr int C = 2 * CH;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int R = p.R;
  const float Rm1 = (float)(R - 1);
  const float* gbase = p.grid_cl + lh * CH;
  float wd[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) wd[k] = p.mlp.w_dens[lh * CH + k];
  const int64_t ngroups = (p.n_points + 127) / 128;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t i = g * 128 + wave * 32 + li;
    const bool active = i < p.n_points;
    const int64_t ii = active ? i : p.n_points - 1;
    const float px = p.pts[ii * 3 + 0], py = p.pts[ii * 3 + 1], pz = p.pts[ii * 3 + 2];
    const float lx = px / p.half_extent, ly = py / p.half_extent, lz = pz / p.half_extent;
    const float ix = ((lx + 1.f) * 0.5f) * Rm1, iy = ((ly + 1.f) * 0.5f) * Rm1, iz = ((lz + 1.f) * 0.5f) * Rm1;
    const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
    const bool xa_in = fx0 >= 0.f && fx0 <= Rm1, xb_in = fx0 >= -1.f && fx0 <= Rm1 - 1.f;
    const bool ya_in = fy0 >= 0.f && fy0 <= Rm1, yb_in = fy0 >= -1.f && fy0 <= Rm1 - 1.f;
    const bool za_in = fz0 >= 0.f && fz0 <= Rm1, zb_in = fz0 >= -1.f && fz0 <= Rm1 - 1.f;
    const float wx[2] = {xa_in ? (fx0 + 1.f) - ix : 0.f, xb_in ? ix - fx0 : 0.f};
    const float wy[2] = {ya_in ? (fy0 + 1.f) - iy : 0.f, yb_in ? iy - fy0 : 0.f};
    const float wz[2] = {za_in ? (fz0 + 1.f) - iz : 0.f, zb_in ? iz - fz0 : 0.f};
    const float dwx[2] = {xa_in ? -1.f : 0.f, xb_in ? 1.f : 0.f};
    const float dwy[2] = {ya_in ? -1.f : 0.f, yb_in ? 1.f : 0.f};
    const float dwz[2] = {za_in ? -1.f : 0.f, zb_in ? 1.f : 0.f};
    const int x0 = (int)fminf(fmaxf(fx0, -1.f), Rm1), y0 = (int)fminf(fmaxf(fy0, -1.f), Rm1),
              z0 = (int)fminf(fmaxf(fz0, -1.f), Rm1);
    const int xs[2] = {max(x0, 0), min(x0 + 1, R - 1)};
    const int ys[2] = {max(y0, 0), min(y0 + 1, R - 1)};
    const int zs[2] = {max(z0, 0), min(z0 + 1, R - 1)};
    float sc[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
      const float4* gp = reinterpret_cast<const float4*>(gbase + ((int64_t)(zs[dz] * R + ys[dy]) * R + xs[dx]) * C);
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < CH / 4; ++v) {
        const float4 t = gp[v];
        s = fmaf(wd[4 * v], t.x, fmaf(wd[4 * v + 1], t.y, fmaf(wd[4 * v + 2], t.z, fmaf(wd[4 * v + 3], t.w, s))));
      }
      sc[corner] = s + __shfl_xor(s, 32);
    }
    float zval = p.mlp.b_dens, gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
      zval = fmaf((wx[dx] * wy[dy]) * wz[dz], sc[corner], zval);
      gx = fmaf((dwx[dx] * wy[dy]) * wz[dz], sc[corner], gx);
      gy = fmaf((wx[dx] * dwy[dy]) * wz[dz], sc[corner], gy);
      gz = fmaf((wx[dx] * wy[dy]) * dwz[dz], sc[corner], gz);
    }
    // chain rule: LeakyReLU'(z) * d(index)/d(point); both positive, kept so that F.normalize's eps acts as in torch
    const float k = (zval > 0.f ? 1.f : 0.2f) * (0.5f * Rm1 / p.half_extent);
    gx *= k;
    gy *= k;
    gz *= k;
    const float nrm = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);
    if (active && lh == 0) {
      normals[i * 3 + 0] = gx / nrm;
      normals[i * 3 + 1] = gy / nrm;
      normals[i * 3 + 2] = gz / nrm;
    }
  }
}

// y[i][j] = LeakyReLU_0.2(y[i][j] + bias[j]) in place (the LAST-layer activation of the feature head)
__global__ __launch_bounds__(256) void bias_leaky_kernel(float* __restrict__ y, const float* __restrict__ bias, int64_t total,
                                                         int cols) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = leaky02(y[i] + bias[i % cols]);
}

}  // namespace

int bias_leaky_launch(float* y, const float* bias, int64_t rows, int cols, void* stream) {
  const int64_t total = rows * cols;
  if (total <= 0) return 0;
  int64_t blocks = cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  HOLO_LAUNCH(bias_leaky_kernel, dim3((unsigned)blocks), dim3(256), stream, y, bias, total, cols);
  return 0;
}

// waves per workgroup of render2_kernel at run time (HOLO_RENDER2_NW=8: development knob, the 8-wave form of the
// 12-wave configurations)
static int render2_waves_rt(int C, int n_fine, int with_normals = 0) {
  const int ch = C / 2;
  (void)ch;
  if (with_normals) {  // render2_waves<CH, 64, true>(); HOLO_RENDER2_NRM_NW=8: the two-waves-per-SIMD form (development knob)
#ifndef HOLO_EMU
    static const char* en = getenv("HOLO_RENDER2_NRM_NW");
    if (en && atoi(en) == 8) return 8;
    return 10;
#else
    return 8;
#endif
  }
  int nw = n_fine <= 64 ? (ch <= 16 ? 12 : 8) : (ch <= 16 ? 8 : 4);
#ifndef HOLO_EMU
  static const char* e = getenv("HOLO_RENDER2_NW");
  if (e && atoi(e) == 8 && nw == 12) nw = 8;
#endif
  return nw;
}

template <int CH, bool SP>
static int render_launch_t(const RenderKernelParams& p, void* stream, int n_wgs) {
  const bool nrm = p.nrm != nullptr || p.nrm_c != nullptr;
  if (render_rays_per_tile(2 * CH, p.n_fine, nrm ? 1 : 0, SP ? 1 : 0, p.train.n_rays > 0 ? 1 : 0) == 4) {
    // the (ray, depth)-tiled kernel: exact fp32, with or without rendered normals
    const int nw = render2_waves_rt(2 * CH, p.n_fine);
#define HOLO_R2(ZFV, TRV, NWV) HOLO_LAUNCH((render2_kernel<CH, ZFV, TRV, NWV>), dim3((unsigned)n_wgs), dim3(64 * NWV), stream, p)
    if (nrm) {
      if (p.train.n_rays > 0) {
        set_error("render: training-mode rendering returns no normals");
        return -1;
      }
      if constexpr (CH <= 16) {
        if (render2_waves_rt(2 * CH, p.n_fine, 1) == 10) {
          HOLO_LAUNCH((render2_kernel<CH, 64, false, 10, true>), dim3((unsigned)n_wgs), dim3(640), stream, p);
        } else {
          HOLO_LAUNCH((render2_kernel<CH, 64, false, 8, true>), dim3((unsigned)n_wgs), dim3(512), stream, p);
        }
      } else {
        set_error("render: rendered normals of %d grid features run on the ray-per-column kernel", 2 * CH);
        return -1;
      }
    } else if (p.train.n_rays > 0) {
      if (p.n_fine <= 64) {
        if (nw == 12) HOLO_R2(64, true, (render2_waves<CH, 64>())); else HOLO_R2(64, true, 8);
      } else {
        HOLO_R2(128, true, (render2_waves<CH, 128>()));
      }
    } else if (p.n_fine <= 64) {
      if (nw == 12) HOLO_R2(64, false, (render2_waves<CH, 64>())); else HOLO_R2(64, false, 8);
    } else {
      HOLO_R2(128, false, (render2_waves<CH, 128>()));
    }
#undef HOLO_R2
    return 0;
  }
  if (p.n_fine <= 64) {
    if (nrm) {
      HOLO_LAUNCH((render_kernel<CH, SP, true, 64>), dim3((unsigned)n_wgs), dim3(64 * render_waves<CH, 64, true>()), stream, p);
    } else {
      HOLO_LAUNCH((render_kernel<CH, SP, false, 64>), dim3((unsigned)n_wgs), dim3(64 * render_waves<CH, 64, false>()), stream, p);
    }
  } else {
    if (nrm) {
      HOLO_LAUNCH((render_kernel<CH, SP, true, 128>), dim3((unsigned)n_wgs), dim3(64 * render_waves<CH, 128, true>()), stream, p);
    } else {
      HOLO_LAUNCH((render_kernel<CH, SP, false, 128>), dim3((unsigned)n_wgs), dim3(64 * render_waves<CH, 128, false>()), stream, p);
    }
  }
  return 0;
}

// rays of one wave tile: 4 on the (ray, depth)-tiled kernel, 32 on the ray-per-column kernel (the bf16x3 split arithmetic
// and - development knob HOLO_RENDER_V1=1 - everything)
int render_rays_per_tile(int C, int n_fine, int with_normals, int split3, int train) {
  if (train) return 4;
  if (split3) return 32;
  // rendered normals: on the (ray, depth)-tiled kernel for the BASELINE / released shapes (round 5), else ray-per-column
  if (with_normals && (C > 32 || n_fine > 64)) return 32;
#ifndef HOLO_EMU
  static const bool v1 = getenv("HOLO_RENDER_V1") != nullptr;
  if (v1) return 32;
#endif
  return 4;
}

// waves per workgroup of the persistent kernel for this configuration (the scratch has one slot per resident wave)
int render_waves_per_wg(int C, int n_fine, int with_normals, int split3, int train) {
  const bool z64 = n_fine <= 64;
  if (render_rays_per_tile(C, n_fine, with_normals, split3, train) == 4) return render2_waves_rt(C, n_fine, with_normals);
  if (C <= 32) {
    if (with_normals) return z64 ? render_waves<16, 64, true>() : render_waves<16, 128, true>();
    return z64 ? render_waves<16, 64, false>() : render_waves<16, 128, false>();
  }
  return z64 ? render_waves<32, 64, false>() : render_waves<32, 128, false>();
}

int density_field_launch(const float* grid_cl, const float* w_dens, int C, int64_t nvox, float* out, void* stream) {
  if (C & 3) {
    set_error("density_field: feature_size must be a multiple of 4");
    return -1;
  }
  HOLO_LAUNCH(density_field_kernel, dim3((unsigned)cdiv(nvox, 256)), dim3(256), stream, grid_cl, w_dens, C, nvox, out);
  return 0;
}

int render_launch(const RenderKernelParams& p, void* stream, int n_wgs) {
  if (p.mlp.Hd != HD) {
    set_error("render: dnet_hidden_dim must be %d (got %d)", HD, p.mlp.Hd);
    return -1;
  }
  if (p.n_coarse < 3 || p.n_coarse > MAXC || p.n_fine < 2 || p.n_fine > 128) {
    set_error("render: n_pts_coarse must be in [3,%d] and n_pts_fine in [2,128]", MAXC);
    return -1;
  }
  if (p.n_cams < 1 || p.n_cams > RenderKernelParams::MAX_CAMS || n_wgs < 1) {
    set_error("render: %d cameras per launch (1..%d)", p.n_cams, RenderKernelParams::MAX_CAMS);
    return -1;
  }
  switch (p.C) {
    case 16:
      return render_launch_t<8, false>(p, stream, n_wgs);
    case 32:
      return p.split3 ? render_launch_t<16, true>(p, stream, n_wgs) : render_launch_t<16, false>(p, stream, n_wgs);
    case 64:
      return render_launch_t<32, false>(p, stream, n_wgs);
    default:
      set_error("render: the fused renderer is built for feature_size 16, 32 or 64 (got %d; 128 input features are "
                "supported by the stand-alone implicit function only)", p.C);
      return -1;
  }
}

int implicit_normals_launch(const ImplicitEvalParams& p, float* normals, void* stream) {
  if (p.n_points <= 0) return 0;
  int64_t groups = cdiv(p.n_points, 128);
  if (groups > 4096) groups = 4096;
  dim3 grid((unsigned)groups);
  switch (p.C) {
    case 16:
      HOLO_LAUNCH(implicit_normals_kernel<8>, grid, dim3(256), stream, p, normals);
      break;
    case 32:
      HOLO_LAUNCH(implicit_normals_kernel<16>, grid, dim3(256), stream, p, normals);
      break;
    case 64:
      HOLO_LAUNCH(implicit_normals_kernel<32>, grid, dim3(256), stream, p, normals);
      break;
    case 128:
      HOLO_LAUNCH(implicit_normals_kernel<64>, grid, dim3(256), stream, p, normals);
      break;
    default:
      set_error("implicit_normals: feature_size must be 16, 32, 64 or 128 (got %d)", p.C);
      return -1;
  }
  return 0;
}

// radiance direction term of every direction -> p.rdir
int implicit_dirs_launch(const ImplicitEvalParams& p, void* stream) {
  if (p.mlp.Hd != HD) {
    set_error("implicit_eval: dnet_hidden_dim must be %d (got %d)", HD, p.mlp.Hd);
    return -1;
  }
  if (p.n_points <= 0) return 0;
  const int64_t n_dirs = cdiv(p.n_points, p.pts_per_dir);
  HOLO_LAUNCH(dir_term_kernel, dim3((unsigned)cdiv(n_dirs, 256)), dim3(256), stream, p.mlp, p.dirs, n_dirs, p.rdir);
  return 0;
}

int implicit_eval_launch(const ImplicitEvalParams& p, void* stream) {
  if (implicit_dirs_launch(p, stream)) return -1;
  return implicit_points_launch(p, stream);
}

// points [p.point0, p.n_points) with the direction terms already in p.rdir
int implicit_points_launch(const ImplicitEvalParams& p, void* stream) {
  if (p.mlp.Hd != HD) {
    set_error("implicit_eval: dnet_hidden_dim must be %d (got %d)", HD, p.mlp.Hd);
    return -1;
  }
  if (p.n_points - p.point0 <= 0) return 0;
  int64_t groups = cdiv(p.n_points - p.point0, 128);
  if (groups > 4096) groups = 4096;
  dim3 grid((unsigned)groups);
#define HOLO_IMPLICIT_CASE(CHV)                                                              \
  if (p.hidden) {                                                                            \
    HOLO_LAUNCH((implicit_eval_kernel<CHV, true>), grid, dim3(256), stream, p);              \
  } else {                                                                                   \
    HOLO_LAUNCH((implicit_eval_kernel<CHV, false>), grid, dim3(256), stream, p);             \
  }                                                                                          \
  break;
  switch (p.C) {
    case 16:
      HOLO_IMPLICIT_CASE(8)
    case 32:
      HOLO_IMPLICIT_CASE(16)
    case 64:
      HOLO_IMPLICIT_CASE(32)
    case 128:
      HOLO_IMPLICIT_CASE(64)
    default:
      set_error("implicit_eval: feature_size must be 16, 32, 64 or 128 (got %d)", p.C);
      return -1;
  }
#undef HOLO_IMPLICIT_CASE
  return 0;
}

}  // namespace holo
