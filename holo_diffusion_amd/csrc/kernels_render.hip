// kernels_render.hip — one fused kernel per rendered frame, replacing the PyTorch3D/Implicitron render
// path of the reference (chunk loop + ~60 ATen launches per chunk, 63 chunks per 400^2 frame):
//
//   ray generation   NDCMultinomialRaysampler/_xy_to_ray_bundle + AdaptiveRaySampler bounds
//                    (invoked at holo_diffusion_model.py:442-448; configs/apple.yaml:135-146)
//   voxel fetch      VolumeLocator.world_to_local_coords + F.grid_sample(bilinear, zeros, align_corners)
//                    (holo_voxel_grid_implicit_function.py:204-225)
//   RenderMLP        holo_voxel_grid_implicit_function.py:107-129 + custom_modules.py:133-160.  The density
//                    net has no activation between its layers (custom_modules.py:108-112 attaches the
//                    LeakyReLU to the last layer only), so render_exec.cpp folds it to ONE affine map
//                    hidden = W_eff f + b_eff in float64; the kernel evaluates that map on the matrix cores
//   composite        EmissionAbsorptionRaymarcher (holo_multipass_ea.py:96-100)
//   resampling       RayPointRefiner + sample_pdf (det.) + sort (holo_multipass_ea.py:116), done as a
//                    lazily generated inverse-CDF stream merged with the coarse depths
//   fine pass        holo_multipass_ea.py:117-123
//
// Mapping: block = 4 waves, wave = 32 rays; lanes l and l+32 share ray (l&31) and split the feature
// channels in halves (lane half h owns channels [h*CH, h*CH+CH)), which is exactly the k index of
// v_mfma_f32_32x32x2_f32.  Per march step a wave evaluates 32 samples (one per ray):
//   D[hidden][sample] += W_eff[hidden][ch] * f[sample][ch]   (A = weights from LDS, B = the lane's own
//   interpolated features) so every lane ends up with 16 hidden units of ITS ray per 32-row tile; the
//   LeakyReLU + 3x256 radiance dot product is then lane-local and only one cross-half shuffle is needed.
// The voxel grid is channels-last so a corner is CH*4 contiguous bytes per lane; the 33.5 MB grid stays
// resident in the 256 MB Infinity Cache / L2 across the frame.
#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {
namespace {

constexpr int HD = 256;       // RenderMLP.dnet_hidden_dim
constexpr int NTILE = HD / 32;
constexpr int MAXC = 64;      // max coarse samples per ray held in LDS (cdf rows)

__device__ __forceinline__ float lin_space(float start, float end, float step, int i, int steps) {
  // torch.linspace: symmetric evaluation around the midpoint
  return (i < steps / 2) ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}
__device__ __forceinline__ float leaky02(float v) { return fmaxf(v, 0.2f * v); }

template <int CH>
__global__ __launch_bounds__(256, 2) void render_kernel(RenderKernelParams p) {
  constexpr int C = 2 * CH;
  constexpr int LDW = C + 4;
  __shared__ __attribute__((aligned(16))) float s_w[HD * LDW];           // W_eff rows (hidden features)
  __shared__ __attribute__((aligned(16))) float s_aux[NTILE * 2 * 16 * 4];  // {b_feat, wr0, wr1, wr2} per D row
  __shared__ float s_cdf[4 * MAXC * 32];                                   // per wave: [j][ray]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;

  // ---- stage the packed MLP
  for (int i = tid; i < HD * (C / 4); i += 256) {
    const int row = i / (C / 4), c4 = i - row * (C / 4);
    *reinterpret_cast<float4*>(s_w + row * LDW + c4 * 4) = *reinterpret_cast<const float4*>(p.w_feat + row * C + c4 * 4);
  }
  for (int i = tid; i < NTILE * 2 * 16; i += 256) {
    const int r = i & 15, h = (i >> 4) & 1, t = i >> 5;
    const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    s_aux[i * 4 + 0] = p.b_feat[row];
    s_aux[i * 4 + 1] = p.w_rad[0 * HD + row];
    s_aux[i * 4 + 2] = p.w_rad[1 * HD + row];
    s_aux[i * 4 + 3] = p.w_rad[2 * HD + row];
  }
  float wd[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) wd[k] = p.w_dens[lh * CH + k];
  __syncthreads();

  // ---- ray setup (pytorch3d NDC grid: +x left, +y up; pixel centres)
  const int npix = p.H * p.W;
  const int ray = blockIdx.x * 128 + wave * 32 + li;
  const bool active = ray < npix;
  const int rr = active ? ray : npix - 1;
  const int py = rr / p.W, px = rr - py * p.W;
  const float hx = p.range_x / (float)p.W, hy = p.range_y / (float)p.H;
  const float minx = p.range_x - hx, maxx = -p.range_x + hx;
  const float miny = p.range_y - hy, maxy = -p.range_y + hy;
  const float xn = lin_space(minx, maxx, (maxx - minx) / (float)(p.W - 1), px, p.W);
  const float yn = lin_space(miny, maxy, (maxy - miny) / (float)(p.H - 1), py, p.H);
  const float dc0 = (xn - p.pp[0]) / p.focal[0], dc1 = (yn - p.pp[1]) / p.focal[1], dc2 = 1.0f;
  float org[3], dir[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float r0 = p.Rm[j * 3 + 0], r1 = p.Rm[j * 3 + 1], r2 = p.Rm[j * 3 + 2];
    const float p1 = (dc0 - p.T[0]) * r0 + (dc1 - p.T[1]) * r1 + (dc2 - p.T[2]) * r2;
    const float p2 = (2.f * dc0 - p.T[0]) * r0 + (2.f * dc1 - p.T[1]) * r1 + (2.f * dc2 - p.T[2]) * r2;
    dir[j] = p2 - p1;
    org[j] = p1 - dir[j];
  }
  // view-direction term of the radiance layer: constant along the ray
  float rdir[3];
  {
    const float nrm = fmaxf(sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]), 1e-12f);
    const float dn[3] = {dir[0] / nrm, dir[1] / nrm, dir[2] / nrm};
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
      float s = p.b_rad[c];
      for (int j = 0; j < 27; ++j) s = fmaf(p.w_dir[c * 27 + j], e[j], s);
      rdir[c] = s;
    }
  }

  const int R = p.R;
  const float Rm1 = (float)(R - 1);
  const float* gbase = p.grid_cl + lh * CH;

  // ---- one sample of the implicit function for this lane's ray; returns raw density and colour
  auto eval = [&](float z, float& sigma, float& cr, float& cg, float& cb) {
    float f[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) f[k] = 0.f;
    {
      const float lx = (org[0] + z * dir[0]) / p.half_extent;
      const float ly = (org[1] + z * dir[1]) / p.half_extent;
      const float lz = (org[2] + z * dir[2]) / p.half_extent;
      const float ix = ((lx + 1.f) * 0.5f) * Rm1, iy = ((ly + 1.f) * 0.5f) * Rm1, iz = ((lz + 1.f) * 0.5f) * Rm1;
      const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
      // guard the float->int conversion for far-away points
      const bool near_grid = fx0 >= -2.f && fx0 <= Rm1 + 1.f && fy0 >= -2.f && fy0 <= Rm1 + 1.f && fz0 >= -2.f &&
                             fz0 <= Rm1 + 1.f;
      if (near_grid) {
        const int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
        const float wx1 = ix - fx0, wy1 = iy - fy0, wz1 = iz - fz0;
        const float wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy, wz0 = (fz0 + 1.f) - iz;
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
          const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
          const int xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
          const bool ok = xx >= 0 && xx < R && yy >= 0 && yy < R && zz >= 0 && zz < R;
          if (ok) {
            const float w = ((dx ? wx1 : wx0) * (dy ? wy1 : wy0)) * (dz ? wz1 : wz0);
            const float4* g = reinterpret_cast<const float4*>(gbase + ((int64_t)(zz * R + yy) * R + xx) * C);
#pragma unroll
            for (int v = 0; v < CH / 4; ++v) {
              const float4 t = g[v];
              f[4 * v + 0] = fmaf(w, t.x, f[4 * v + 0]);
              f[4 * v + 1] = fmaf(w, t.y, f[4 * v + 1]);
              f[4 * v + 2] = fmaf(w, t.z, f[4 * v + 2]);
              f[4 * v + 3] = fmaf(w, t.w, f[4 * v + 3]);
            }
          }
        }
      }
    }
    // density row: lane-local half dot product
    float dpart = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) dpart = fmaf(wd[k], f[k], dpart);
    // hidden features on the matrix cores, tile by tile
    float rp0 = 0.f, rp1 = 0.f, rp2 = 0.f;
#pragma unroll 1
    for (int t = 0; t < NTILE; ++t) {
      // the LDS-resident weights are loop invariant across march steps: keep the compiler from hoisting
      // (and spilling) 8 tiles of operands out of the sample loops
      asm volatile("" ::: "memory");
      const float4* aux = reinterpret_cast<const float4*>(s_aux + ((t * 2 + lh) * 16) * 4);
      const float4* ap = reinterpret_cast<const float4*>(s_w + (t * 32 + li) * LDW + lh * CH);
      float a[CH];
#pragma unroll
      for (int v = 0; v < CH / 4; ++v) {
        const float4 t4 = ap[v];
        a[4 * v + 0] = t4.x;
        a[4 * v + 1] = t4.y;
        a[4 * v + 2] = t4.z;
        a[4 * v + 3] = t4.w;
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = aux[r].x;  // accumulator starts at the bias of the lane's own rows
#pragma unroll
      for (int k = 0; k < CH; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], f[k], acc, 0, 0, 0);
      // the MFMA k index runs over both lane halves, so each lane now holds complete hidden units
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float4 w4 = aux[r];
        const float hv = leaky02(acc[r]);
        rp0 = fmaf(w4.y, hv, rp0);
        rp1 = fmaf(w4.z, hv, rp1);
        rp2 = fmaf(w4.w, hv, rp2);
      }
    }
    dpart += __shfl_xor(dpart, 32);
    rp0 += __shfl_xor(rp0, 32);
    rp1 += __shfl_xor(rp1, 32);
    rp2 += __shfl_xor(rp2, 32);
    sigma = leaky02(dpart + p.b_dens);
    cr = 1.f / (1.f + expf(-leaky02(rp0 + rdir[0])));
    cg = 1.f / (1.f + expf(-leaky02(rp1 + rdir[1])));
    cb = 1.f / (1.f + expf(-leaky02(rp2 + rdir[2])));
  };

  // ---- coarse pass
  const int nc = p.n_coarse, nf = p.n_fine;
  const float zstep = (p.zmax - p.zmin) / (float)(nc - 1);
  float* cdf = s_cdf + wave * (MAXC * 32) + li;  // element j at cdf[j*32]
  {
    float cum = 0.f, Tr = 1.f, ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, O = 0.f;
    float zi = lin_space(p.zmin, p.zmax, zstep, 0, nc);
    for (int i = 0; i < nc; ++i) {
      const float zn = (i + 1 < nc) ? lin_space(p.zmin, p.zmax, zstep, i + 1, nc) : 0.f;
      float sg, cr, cg, cb;
      eval(zi, sg, cr, cg, cb);
      const float delta = (i + 1 < nc) ? zn - zi : p.background_opacity;
      const float x = delta * fmaxf(sg, 0.f);
      const float cap = 1.f - expf(-x);
      cum += x;
      O = 1.f - expf(-cum);
      const float w = cap * Tr;
      ar = fmaf(w, cr, ar);
      ag = fmaf(w, cg, ag);
      ab = fmaf(w, cb, ab);
      ad = fmaf(w, zi, ad);
      if (lh == 0) cdf[i * 32] = w;  // weights for now; turned into the cdf below
      Tr = 1.f - O;
      zi = zn;
    }
    if (p.rgb_c && active && lh == 0) {
      p.rgb_c[0 * npix + ray] = ar + (1.f - O) * p.bg[0];
      p.rgb_c[1 * npix + ray] = ag + (1.f - O) * p.bg[1];
      p.rgb_c[2 * npix + ray] = ab + (1.f - O) * p.bg[2];
      p.depth_c[ray] = ad;
      p.mask_c[ray] = O;
    }
  }

  // ---- weights[1:-1] -> pdf -> cdf (in place; both halves write identical values)
  // cdf has nb = nc-1 entries, cdf[0] = 0;  bins (interval mid points) also nb entries
  const int nb = nc - 1;
  __builtin_amdgcn_wave_barrier();
  if (lh == 0) {
    float S = 0.f;
    for (int m = 1; m < nc - 1; ++m) S += cdf[m * 32] + p.pdf_eps;
    float run = 0.f;
    cdf[0] = 0.f;
    for (int j = 1; j < nb; ++j) {  // cdf[j] = cdf[j-1] + (w[j] + eps)/S ; slot j still holds w[j] here
      run += (cdf[j * 32] + p.pdf_eps) / S;
      cdf[j * 32] = run;
    }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- fine pass: merge of the coarse depths with the lazily generated inverse-CDF samples
  {
    const float ustep = 1.0f / (float)(nf - 1);
    int ci = 0, k = 0, ind = 0;
    auto zcoarse = [&](int i) { return lin_space(p.zmin, p.zmax, zstep, i, nc); };
    auto mid = [&](int i) {
      const float a0 = zcoarse(i), a1 = zcoarse(i + 1);
      return a0 - (a0 - a1) * 0.5f;  // torch.lerp(z[1:], z[:-1], 0.5)
    };
    auto gen_fine = [&](int kk) {
      const float u = lin_space(0.f, 1.f, ustep, kk, nf);
      while (ind < nb && cdf[ind * 32] <= u) ++ind;  // searchsorted(right=True)
      const int below = ind - 1 > 0 ? ind - 1 : 0;
      const int above = ind < nb - 1 ? ind : nb - 1;
      const float cb_ = cdf[below * 32], ca_ = cdf[above * 32];
      float den = ca_ - cb_;
      if (den < p.pdf_eps) den = 1.f;
      const float tt = (u - cb_) / den;
      const float bb = mid(below), ba = mid(above);
      return bb + tt * (ba - bb);
    };
    float zc_head = zcoarse(0);
    float zf_head = nf > 0 ? gen_fine(0) : 0.f;
    auto pop = [&]() {
      float v;
      if (ci < nc && (k >= nf || zc_head <= zf_head)) {
        v = zc_head;
        ++ci;
        if (ci < nc) zc_head = zcoarse(ci);
      } else {
        v = zf_head;
        ++k;
        if (k < nf) zf_head = gen_fine(k);
      }
      return v;
    };
    const int total = nc + nf;
    float cum = 0.f, Tr = 1.f, ar = 0.f, ag = 0.f, ab = 0.f, ad = 0.f, O = 0.f;
    float zi = pop();
    for (int s = 0; s < total; ++s) {
      const float zn = (s + 1 < total) ? pop() : 0.f;
      float sg, cr, cg, cb;
      eval(zi, sg, cr, cg, cb);
      const float delta = (s + 1 < total) ? zn - zi : p.background_opacity;
      const float x = delta * fmaxf(sg, 0.f);
      const float cap = 1.f - expf(-x);
      cum += x;
      O = 1.f - expf(-cum);
      const float w = cap * Tr;
      ar = fmaf(w, cr, ar);
      ag = fmaf(w, cg, ag);
      ab = fmaf(w, cb, ab);
      ad = fmaf(w, zi, ad);
      Tr = 1.f - O;
      zi = zn;
    }
    if (active && lh == 0) {
      p.rgb[0 * npix + ray] = ar + (1.f - O) * p.bg[0];
      p.rgb[1 * npix + ray] = ag + (1.f - O) * p.bg[1];
      p.rgb[2 * npix + ray] = ab + (1.f - O) * p.bg[2];
      p.depth[ray] = ad;
      p.mask[ray] = O;
    }
  }
}

}  // namespace

int render_launch(const RenderKernelParams& p, void* stream) {
  if (p.Hd != HD) {
    set_error("render: dnet_hidden_dim must be %d (got %d)", HD, p.Hd);
    return -1;
  }
  if (p.n_coarse < 3 || p.n_coarse > MAXC || p.n_fine < 2) {
    set_error("render: n_pts_coarse must be in [3,%d] and n_pts_fine >= 2", MAXC);
    return -1;
  }
  const int npix = p.H * p.W;
  dim3 grid((unsigned)cdiv(npix, 128));
  switch (p.C) {
    case 16:
      HOLO_LAUNCH(render_kernel<8>, grid, dim3(256), stream, p);
      break;
    case 32:
      HOLO_LAUNCH(render_kernel<16>, grid, dim3(256), stream, p);
      break;
    case 64:
      HOLO_LAUNCH(render_kernel<32>, grid, dim3(256), stream, p);
      break;
    default:
      set_error("render: feature_size must be 16, 32 or 64 (got %d)", p.C);
      return -1;
  }
  return 0;
}

}  // namespace holo
