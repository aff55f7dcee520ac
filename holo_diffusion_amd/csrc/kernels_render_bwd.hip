// kernels_render_bwd.hip — backward of the training-mode renderer: gradients of the rendered rays
// (HoloMultiPassEmissionAbsorptionRenderer, holo_multipass_ea.py:79-125, both passes) with respect to the voxel grid
// (grid_sample's scatter-add) and the RenderMLP parameters (holo_voxel_grid_implicit_function.py:73-129).
//
// The forward kernel (render2_kernel<.., TRAIN>) is run once more and leaves the merged depth list of every ray (coarse
// depths + importance samples in depth order, with a flag per NEW sample); the importance sampling itself carries no
// gradient (PyTorch3D's RayPointRefiner samples under torch.no_grad()).  Every merged point is then evaluated ONCE - the
// coarse pass composites the flagged-off subset of the same points - in feature-major buffers [feature][point] so that
// the per-point kernels are coalesced and the three matrix products are plain GEMMs on the fp32 matrix cores
// (gemm_launch):
//   F   [n][C]    trilinear features                                         (rbwd_gather_kernel)
//   YT  [Hp][n]   = We F^T      pre-activations of the folded density net    (GEMM 1)
//   per point: leaky, radiance head, sigmoid                                  (rbwd_point_fwd_kernel)
//   per ray:   emission-absorption backward of both passes                    (rbwd_composite_kernel)
//   per point: radiance head + LeakyReLU backward, YT <- d pre-activation     (rbwd_point_bwd_kernel)
//   GFT [C][n]    = We^T YT     gradient of the features                      (GEMM 2)
//   dWe [Hp][C]   = YT F        (GEMM 3, split over the points)   dWr_h [Hd][4] = AT GR   (GEMM 4)
//   scatter-add of GFT through the trilinear weights (atomicAdd, as grid_sample's backward; in the deterministic mode of
//   holo_ctx_set_deterministic as order-independent fixed-point sums)                        (rbwd_scatter_kernel)
// The folded weights' gradients are unfolded to the four Linear layers on the host in float64 (render_exec.cpp).
#include <math.h>
#include <string.h>

#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {
namespace {

__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : 0.2f * v; }
__device__ __forceinline__ float dleaky(float v) { return v > 0.f ? 1.f : 0.2f; }

// ---- per ray: origin, direction (as the forward kernel's ray setup), radiance direction term W_dir e(dir) + b_rad and
// the 27 embedding entries (harmonic embedding of the NORMALISED direction, 4 frequencies, [sin | cos | identity])
__global__ __launch_bounds__(256) void rbwd_rays_kernel(RenderBwdRays p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n_cams * p.n_rays) return;
  const int cam_i = i / p.n_rays;
  const RenderKernelParams::Cam& cam = p.cams[cam_i];
  const float xn = holo_ld_sys(p.xys + (int64_t)(p.ray0 + i) * 2 + 0), yn = holo_ld_sys(p.xys + (int64_t)(p.ray0 + i) * 2 + 1);
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
  float* o = p.rays + (int64_t)(p.ray0 + i) * RBWD_REC;
  const float nrm = fmaxf(sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]), 1e-12f);  // F.normalize eps
  const float dn[3] = {dir[0] / nrm, dir[1] / nrm, dir[2] / nrm};
  float rd[3] = {p.b_rad[0], p.b_rad[1], p.b_rad[2]};
  for (int j = 0; j < 27; ++j) {
    const int jj = j < 24 ? j % 12 : 0;
    const int a = j < 24 ? jj >> 2 : j - 24;
    const float arg = dn[a] * (float)(1 << (jj & 3));
    const float e = j < 12 ? sinf(arg) : (j < 24 ? cosf(arg) : dn[a]);
    o[RBWD_REC_EMB + j] = e;
#pragma unroll
    for (int c = 0; c < 3; ++c) rd[c] = fmaf(p.w_dir[c * 27 + j], e, rd[c]);
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    o[j] = org[j];
    o[RBWD_REC_DIR + j] = dir[j];
    o[RBWD_REC_RDIR + j] = rd[j];
  }
}

// trilinear weights / corner indices of a world point (zeros padding, align_corners = True; the forward's eval_point)
struct Tri {
  float w[8];
  uint32_t v[8];
};
__device__ __forceinline__ void tri_setup(float px, float py, float pz, float half_extent, int R, Tri& t) {
  const float Rm1 = (float)(R - 1);
  const float lx = px / half_extent, ly = py / half_extent, lz = pz / half_extent;
  const float ix = ((lx + 1.f) * 0.5f) * Rm1, iy = ((ly + 1.f) * 0.5f) * Rm1, iz = ((lz + 1.f) * 0.5f) * Rm1;
  const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
  const float wxa = (fx0 >= 0.f && fx0 <= Rm1) ? (fx0 + 1.f) - ix : 0.f;
  const float wxb = (fx0 >= -1.f && fx0 <= Rm1 - 1.f) ? ix - fx0 : 0.f;
  const float wya = (fy0 >= 0.f && fy0 <= Rm1) ? (fy0 + 1.f) - iy : 0.f;
  const float wyb = (fy0 >= -1.f && fy0 <= Rm1 - 1.f) ? iy - fy0 : 0.f;
  const float wza = (fz0 >= 0.f && fz0 <= Rm1) ? (fz0 + 1.f) - iz : 0.f;
  const float wzb = (fz0 >= -1.f && fz0 <= Rm1 - 1.f) ? iz - fz0 : 0.f;
  const int x0 = (int)fminf(fmaxf(fx0, -1.f), Rm1), y0 = (int)fminf(fmaxf(fy0, -1.f), Rm1), z0 = (int)fminf(fmaxf(fz0, -1.f), Rm1);
  const int xa = max(x0, 0), xb = min(x0 + 1, R - 1);
  const int ya = max(y0, 0), yb = min(y0 + 1, R - 1);
  const int za = max(z0, 0), zb = min(z0 + 1, R - 1);
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
    t.w[corner] = ((dx ? wxb : wxa) * (dy ? wyb : wya)) * (dz ? wzb : wza);
    t.v[corner] = (uint32_t)(((dz ? zb : za) * R + (dy ? yb : ya)) * R + (dx ? xb : xa));
  }
}
__device__ __forceinline__ void point_of(const RenderBwdChunk& p, int64_t pt, float& px, float& py, float& pz) {
  const int64_t ray = p.ray0 + pt / p.nm;
  const float* rr = p.rays + ray * RBWD_REC;
  const float z = p.z_merged[p.ray0 * p.nm + pt];
  px = rr[0] + z * rr[RBWD_REC_DIR + 0];
  py = rr[1] + z * rr[RBWD_REC_DIR + 1];
  pz = rr[2] + z * rr[RBWD_REC_DIR + 2];
}

// F[pt][c4..c4+3]: one thread per (point, 4 channels); rows [n, n_pad) are zeroed (K padding of the GEMMs)
__global__ __launch_bounds__(256) void rbwd_gather_kernel(RenderBwdChunk p) {
  const int c4n = p.C >> 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n_pad * c4n) return;
  const int64_t pt = i / c4n;
  const int c4 = (int)(i - pt * c4n) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (pt < p.n) {
    float px, py, pz;
    point_of(p, pt, px, py, pz);
    Tri t;
    tri_setup(px, py, pz, p.half_extent, p.R, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 g = *reinterpret_cast<const float4*>(p.grid_cl + (int64_t)t.v[k] * p.C + c4);
      acc.x = fmaf(t.w[k], g.x, acc.x);
      acc.y = fmaf(t.w[k], g.y, acc.y);
      acc.z = fmaf(t.w[k], g.z, acc.z);
      acc.w = fmaf(t.w[k], g.w, acc.w);
    }
  }
  *reinterpret_cast<float4*>(p.F + pt * p.C + c4) = acc;
}

// per point: hidden activations a = leaky(YT + b) -> AT, density = a[Hd], radiance pre-activation, colour.
// val[pt] = (density, r, g, b);  drad[pt] = d colour / d radiance pre-activation
__global__ __launch_bounds__(256) void rbwd_point_fwd_kernel(RenderBwdChunk p) {
  const int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= p.n) return;
  const float* rr = p.rays + (p.ray0 + pt / p.nm) * RBWD_REC + RBWD_REC_RDIR;
  float r0 = rr[0], r1 = rr[1], r2 = rr[2];
  for (int h = 0; h < p.Hd; ++h) {
    const float a = leaky(p.YT[(int64_t)h * p.ld + pt] + p.be[h]);
    p.AT[(int64_t)h * p.ld + pt] = a;
    r0 = fmaf(p.w_rad[h], a, r0);
    r1 = fmaf(p.w_rad[p.Hd + h], a, r1);
    r2 = fmaf(p.w_rad[2 * p.Hd + h], a, r2);
  }
  const float dens = leaky(p.YT[(int64_t)p.Hd * p.ld + pt] + p.be[p.Hd]);
  const float rp[3] = {r0, r1, r2};
  float col[3], dr[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float c = 1.f / (1.f + expf(-leaky(rp[j])));
    col[j] = c;
    dr[j] = c * (1.f - c) * dleaky(rp[j]);
  }
  p.val[pt] = make_float4(dens, col[0], col[1], col[2]);
  p.drad[pt] = make_float4(dr[0], dr[1], dr[2], 0.f);
}

// per ray (one thread): emission-absorption backward of the fine pass (all nm merged points) and the coarse pass (the
// points that are not flagged new), holo_multipass_ea.py:79-103 / EmissionAbsorptionRaymarcher:
//   x_q = delta_q relu(s_q + std noise_q),  w_q = (1 - e^{-x_q}) e^{-cum_{q-1}},  rgb = sum w c + (1 - O) bg,  depth = sum w z,
//   O = 1 - e^{-cum_last}
//   dL/dx_k = gw_k e^{-cum_k} - sum_{i>k} gw_i w_i + gO e^{-cum_last},   gw = g_rgb . c + g_depth z,   gO = g_mask - g_rgb . bg
// gval[pt] = (d density, d colour)
__global__ __launch_bounds__(64) void rbwd_composite_kernel(RenderBwdChunk p) {
  const int rl = blockIdx.x * blockDim.x + threadIdx.x;
  if (rl >= p.n_rays_chunk) return;
  const int64_t ray = p.ray0 + rl;
  const int cam_i = (int)(ray / p.rays_per_cam);
  const int64_t rc = ray - (int64_t)cam_i * p.rays_per_cam;
  const int nm = p.nm, nc = p.n_coarse;
  const float* z = p.z_merged + ray * nm;
  const unsigned char* fl = p.new_flags + ray * nm;
  const float4* val = p.val + (int64_t)rl * nm;
  float4* gval = p.gval + (int64_t)rl * nm;
  for (int pass = 0; pass < 2; ++pass) {
    const bool fine = pass == 0;
    const float* g_rgb = fine ? p.g_rgb : p.g_rgb_c;
    const float* g_dep = fine ? p.g_depth : p.g_depth_c;
    const float* g_msk = fine ? p.g_mask : p.g_mask_c;
    float gr[3] = {0.f, 0.f, 0.f}, gd = 0.f, gm = 0.f;
    if (g_rgb)
      for (int j = 0; j < 3; ++j) gr[j] = holo_ld_sys(g_rgb + ((int64_t)cam_i * 3 + j) * p.rays_per_cam + rc);
    if (g_dep) gd = holo_ld_sys(g_dep + ray);
    if (g_msk) gm = holo_ld_sys(g_msk + ray);
    const float* noise = fine ? p.noise_fine : p.noise_coarse;
    const int np = fine ? nm : nc;
    const float gO = gm - (gr[0] * p.bg[0] + gr[1] * p.bg[1] + gr[2] * p.bg[2]);
    // forward sweep: cum (double), the weights; remember x and e^{-cum} in the gradient buffer's slots
    double cum = 0.0;
    int k = 0;       // index within the pass
    int prev = -1;   // merged position of the previous sample of the pass
    float z_prev = 0.f, s_prev = 0.f;
    // (two sweeps over the merged list; the pass-local scalars live in gval.x .. of the pass's own samples temporarily)
    double total_gw_w = 0.0;
    for (int q = 0; q <= nm; ++q) {
      const bool take = q < nm && (fine || !fl[q]);
      if (!(take || q == nm)) continue;
      if (prev >= 0) {  // close the interval of the previous sample: its delta ends at this sample's depth
        // (double throughout: d colour / d density is the small difference (c_k - colour behind k) of O(1) terms)
        const float dl = q < nm ? z[q] - z_prev : p.background_opacity;
        const double x = (double)dl * (double)fmaxf(s_prev, 0.f);
        const double Tq = exp(-cum);
        cum += x;
        const double w = (1.0 - exp(-x)) * Tq;
        const float4 v = val[prev];
        const double gw = (double)gr[0] * v.y + (double)gr[1] * v.z + (double)gr[2] * v.w + (double)gd * z_prev;
        total_gw_w += gw * w;
        double* t = p.tmp + ((int64_t)rl * nm + prev) * 4;
        t[0] = gw;
        t[1] = w;
        t[2] = exp(-cum);  // e^{-cum_k}
        t[3] = (s_prev > 0.f) ? (double)dl : 0.0;
      }
      if (q < nm) {
        float s = val[q].x;
        if (noise) s += p.noise_std * holo_ld_sys(noise + ray * np + k);
        s_prev = s;
        z_prev = z[q];
        prev = q;
        ++k;
      }
    }
    const double e_last = exp(-cum);
    double prefix = 0.0;  // sum_{i<=k} gw_i w_i
    for (int q = 0; q < nm; ++q) {
      const bool take = fine || !fl[q];
      float4 g = fine ? make_float4(0.f, 0.f, 0.f, 0.f) : gval[q];
      if (take) {
        const double* t = p.tmp + ((int64_t)rl * nm + q) * 4;
        prefix += t[0] * t[1];
        const double dx = t[0] * t[2] - (total_gw_w - prefix) + (double)gO * e_last;
        g.x += (float)(dx * t[3]);
        g.y += (float)(t[1] * gr[0]);
        g.z += (float)(t[1] * gr[1]);
        g.w += (float)(t[1] * gr[2]);
      }
      gval[q] = g;
    }
  }
}

// per point: g_r = d colour * drad; AT holds the hidden activations, YT the pre-activations (without bias);
// YT <- d pre-activation (feature-major, rows Hd+1.. zero), GR[pt] = (g_r, 0)
__global__ __launch_bounds__(256) void rbwd_point_bwd_kernel(RenderBwdChunk p) {
  const int64_t pt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pt >= p.n_pad) return;
  if (pt >= p.n) {  // K padding of the products over the points
    p.GR[pt] = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float4 g = p.gval[pt], dr = p.drad[pt];
  const float g0 = g.y * dr.x, g1 = g.z * dr.y, g2 = g.w * dr.z;
  p.GR[pt] = make_float4(g0, g1, g2, 0.f);
  for (int h = 0; h < p.Hd; ++h) {
    const int64_t o = (int64_t)h * p.ld + pt;
    const float gy = p.w_rad[h] * g0 + p.w_rad[p.Hd + h] * g1 + p.w_rad[2 * p.Hd + h] * g2;
    p.YT[o] = gy * dleaky(p.YT[o] + p.be[h]);
  }
  const int64_t o = (int64_t)p.Hd * p.ld + pt;
  p.YT[o] = g.x * dleaky(p.YT[o] + p.be[p.Hd]);
  for (int h = p.Hd + 1; h < p.Hp; ++h) p.YT[(int64_t)h * p.ld + pt] = 0.f;
}

// per ray sums of g_r (the direction part of the radiance layer sees one embedding per ray): gr_ray[ray] = sum over its points
__global__ __launch_bounds__(64) void rbwd_ray_sum_kernel(RenderBwdChunk p) {
  const int rl = blockIdx.x * blockDim.x + threadIdx.x;
  if (rl >= p.n_rays_chunk) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int q = 0; q < p.nm; ++q) {
    const float4 g = p.GR[(int64_t)rl * p.nm + q];
    s0 += g.x;
    s1 += g.y;
    s2 += g.z;
  }
  float* o = p.gr_ray + (p.ray0 + rl) * 4;
  o[0] = (float)s0;
  o[1] = (float)s1;
  o[2] = (float)s2;
  o[3] = 0.f;
}
// d W_dir[j][e] = sum_ray gr_ray[ray][j] E[ray][e];  d b_rad[j] = sum_ray gr_ray[ray][j]   (one thread per output, 84 of them)
__global__ __launch_bounds__(128) void rbwd_dir_grad_kernel(const float* __restrict__ gr_ray, const float* __restrict__ rays,
                                                           int64_t n_rays_total, float* __restrict__ out) {
  const int i = threadIdx.x;
  if (i >= 84) return;
  const int j = i / 28, e = i - j * 28;
  double s = 0.0;
  for (int64_t r = 0; r < n_rays_total; ++r) s += (double)gr_ray[r * 4 + j] * (e < 27 ? (double)rays[r * RBWD_REC + RBWD_REC_EMB + e] : 1.0);
  out[i] = (float)s;
}

// row sums of YT (d b_eff): one block per row
__global__ __launch_bounds__(256) void rbwd_rowsum_kernel(const float* __restrict__ YT, int64_t ld, int64_t n, float* __restrict__ out,
                                                         int accumulate) {
  __shared__ double red[256];
  const int h = blockIdx.x;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)YT[(int64_t)h * ld + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[h] = (accumulate ? out[h] : 0.f) + (float)red[0];
}

// scatter-add of the feature gradients through the trilinear weights: one thread per (point, channel), channel fastest.
// FIXED (the deterministic mode): the same products added as fixed-point integers (holo_common.h) - the sum of a chunk no
// longer depends on the order the atomics land in; the binary point comes from the chunk's max |GFT| (trilinear weights <= 1).
template <bool FIXED>
__global__ __launch_bounds__(256) void rbwd_scatter_kernel(RenderBwdChunk p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n * p.C) return;
  const int64_t pt = i / p.C;
  const int c = (int)(i - pt * p.C);
  float px, py, pz;
  point_of(p, pt, px, py, pz);
  Tri t;
  tri_setup(px, py, pz, p.half_extent, p.R, t);
  const float g = p.GFT[(int64_t)c * p.ld + pt];
  if (FIXED) {
    const uint32_t mb = *p.gfix_max;
    if (mb == 0u || mb >= 0x7f800000u) return;  // nothing to add / not finite (the flush writes NaN)
    const int shift = holo_fix_shift(mb);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (t.w[k] != 0.f) holo_fix_add(p.gfix + (int64_t)t.v[k] * p.C + c, t.w[k] * g, shift);
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (t.w[k] != 0.f) HOLO_ATOMIC_ADD_F32(p.ggrid_cl + (int64_t)t.v[k] * p.C + c, t.w[k] * g);  // (the hardware instruction, not atomicAdd's compare-and-swap loop)
  }
}

// deterministic mode, stage 1: bits of max |GFT| over the chunk's points (a maximum does not depend on the order either)
__global__ __launch_bounds__(256) void rbwd_absmax_kernel(RenderBwdChunk p) {
  __shared__ uint32_t red[256];
  uint32_t m = 0u;
  const int64_t total = p.n * p.C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = i / p.n, pt = i - c * p.n;
    const uint32_t b = __float_as_uint(p.GFT[c * p.ld + pt]) & 0x7fffffffu;
    m = b > m ? b : m;
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + o] ? red[threadIdx.x] : red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && red[0]) atomicMax(p.gfix_max, red[0]);
}
// deterministic mode, stage 3: the chunk's sums leave the integer buffer (which is zero again afterwards) - chunks are added
// to the gradient in chunk order
__global__ __launch_bounds__(256) void rbwd_fix_flush_kernel(RenderBwdChunk p, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t mb = *p.gfix_max;
  if (mb == 0u) return;
  if (mb >= 0x7f800000u) {
    p.ggrid_cl[i] = __uint_as_float(0x7fc00000u);
    return;
  }
  const long long q = p.gfix[i];
  if (q != 0) {
    p.ggrid_cl[i] += holo_fix_value(q, holo_fix_shift(mb));
    p.gfix[i] = 0;
  }
}

}  // namespace

int rbwd_rays_launch(const RenderBwdRays& p, void* stream) {
  const int n = p.n_cams * p.n_rays;
  HOLO_LAUNCH(rbwd_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), stream, p);
  return 0;
}
int rbwd_gather_launch(const RenderBwdChunk& p, void* stream) {
  HOLO_LAUNCH(rbwd_gather_kernel, dim3((unsigned)((p.n_pad * (p.C >> 2) + 255) / 256)), dim3(256), stream, p);
  return 0;
}
int rbwd_point_fwd_launch(const RenderBwdChunk& p, void* stream) {
  HOLO_LAUNCH(rbwd_point_fwd_kernel, dim3((unsigned)((p.n + 255) / 256)), dim3(256), stream, p);
  return 0;
}
int rbwd_composite_launch(const RenderBwdChunk& p, void* stream) {
  HOLO_LAUNCH(rbwd_composite_kernel, dim3((unsigned)((p.n_rays_chunk + 63) / 64)), dim3(64), stream, p);
  return 0;
}
int rbwd_point_bwd_launch(const RenderBwdChunk& p, void* stream) {
  HOLO_LAUNCH(rbwd_point_bwd_kernel, dim3((unsigned)((p.n_pad + 255) / 256)), dim3(256), stream, p);
  HOLO_LAUNCH(rbwd_ray_sum_kernel, dim3((unsigned)((p.n_rays_chunk + 63) / 64)), dim3(64), stream, p);
  return 0;
}
int rbwd_dir_grad_launch(const float* gr_ray, const float* rays, int64_t n_rays_total, float* out, void* stream) {
  HOLO_LAUNCH(rbwd_dir_grad_kernel, dim3(1), dim3(128), stream, gr_ray, rays, n_rays_total, out);
  return 0;
}
int rbwd_rowsum_launch(const float* YT, int64_t ld, int64_t n, int rows, float* out, int accumulate, void* stream) {
  HOLO_LAUNCH(rbwd_rowsum_kernel, dim3((unsigned)rows), dim3(256), stream, YT, ld, n, out, accumulate);
  return 0;
}
int rbwd_scatter_launch(const RenderBwdChunk& p, void* stream) {
  if (p.gfix)
    HOLO_LAUNCH(rbwd_scatter_kernel<true>, dim3((unsigned)((p.n * p.C + 255) / 256)), dim3(256), stream, p);
  else
    HOLO_LAUNCH(rbwd_scatter_kernel<false>, dim3((unsigned)((p.n * p.C + 255) / 256)), dim3(256), stream, p);
  return 0;
}
int rbwd_absmax_launch(const RenderBwdChunk& p, void* stream) {
  const int64_t blocks = (p.n * p.C + 255) / 256;
  HOLO_LAUNCH(rbwd_absmax_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), stream, p);
  return 0;
}
int rbwd_fix_flush_launch(const RenderBwdChunk& p, void* stream) {
  const int64_t total = (int64_t)p.R * p.R * p.R * p.C;
  HOLO_LAUNCH(rbwd_fix_flush_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), stream, p, total);
  return 0;
}

}  // namespace holo
