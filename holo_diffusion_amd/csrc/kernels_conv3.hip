// kernels_conv3.hip — stride-1 3x3x3 convolution in Winograd F(2x2x2, 3x3x3) form on the exact-fp32 matrix cores.
//
// The same conv3d calls as kernels_conv.hip (holo_diffusion/guided_diffusion/unet.py:185,211 ResBlock convs, :89 the
// Upsample conv, :657 the input conv) with the same fused staging (GroupNorm32 apply + FiLM + SiLU, nearest x2
// upsampling, the skip-connection concat, bias, residual, GroupNorm statistics of the output, the ResBlock's 1x1x1
// skip_connection, unet.py:222,256).
//
// Why a third form.  In exact fp32 the matrix pipe runs at the vector rate (157 TFLOP/s, the same FMA lanes), so the only
// way under the multiply count is to multiply less.  conv_wino2_kernel transforms over (depth, height): 48 pseudo-taps
// per 2 x 2 outputs where the direct form spends 108.  This kernel transforms over all three axes: 64 pseudo-taps per
// 2 x 2 x 2 outputs instead of 216 - 8 multiplies per output and input channel instead of 12 (wino2) or 27 (direct).
//
// The price is 64 accumulator sets per tile - 256 accumulator registers per lane - which only fits ONE wave per SIMD
// (512 registers: 256 accumulators + 256 others).  So the kernel is built the other way round from conv_wino2_kernel,
// whose two co-resident workgroups hide each other's staging: ONE persistent 4-wave workgroup per CU walks a list of
// (tile, 64-Cout block, K split) items, and every wave software-pipelines everything itself - the next chunk's raw halo
// is requested (global -> registers) at the start of a chunk, activated / z-transformed / written to the OTHER LDS
// buffer in four pieces between the MFMA groups of the chunk, across tile boundaries too; one barrier per chunk.
//
// Tile: 2 x 8 x 8 output voxels x 64 output channels (wave w: channels 16 w .. +15).  An MFMA row is a (y tile, x tile)
// pair - a 2 x 2 patch of outputs per plane pair - so ONE 16-row v_mfma_f32_16x16x4_f32 tile covers the whole 8 x 8 face.
// LDS: the z-transformed 4 x 10 x 10 halo of one 32-channel chunk, [xi_z][hy][hx][36 words] (the layout of
// conv_wino2_kernel), twice (double buffer, 115 KB).  Per (xi_z, 16-channel half) a lane reads its 4 x 4 (y,x) patch
// (16 ds_read_b128, conflict free: lane's channels are 4 kq + 16 half .. +3, rows yt = lj & 3, xt = lj >> 2), applies
// B^T . B along x then y in registers (64 v_pk_add_f32) and issues the 64 MFMAs of the 16 (xi_y, xi_x) pseudo-taps.
// Weights: U = (G x G x G) g prepared in float64 at set_param, packed in the order the wave consumes them
// ([chunk][16-Cout slice][xi_z][half][xi_y][xi_x][lane][4]: a wave streams 128 KB per chunk linearly, global -> registers
// three 16-MFMA groups ahead).  Output transform A^T . A per axis is lane-local (a lane's accumulators of one register
// index belong to the same 2 x 2 x 2 voxels).
//
// The fused 1x1x1 skip connection needs no Winograd form at all: o0 = m0 + m1 + m2, o1 = m1 - m2 - m3 per axis, so a
// value v destined for output (dz,dy,dx) is accumulated as (-1)^(dz+dy+dx) v into pseudo-tap (3dz, 3dy, 3dx).  The
// skip's A operands are the raw block input at the lane's own 8 voxels, read straight from global memory (no halo, no
// LDS, no activation); its chunks ride at the end of the main chunks' MFMA streams.
#include <stdio.h>
#include <stdlib.h>

#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {

namespace {

constexpr int W3_BK = 32;                      // channels per chunk
constexpr int W3_RS = 36;                      // words per halo column (32 channels + 4 pad: 16-byte aligned rows)
constexpr int W3_HY = 10, W3_HX = 10;          // halo rows / columns of an 8 x 8 face
constexpr int W3_PLANE = W3_HY * W3_HX;        // 100 (y,x) columns
constexpr int W3_HALO = 4 * W3_PLANE * W3_RS;  // words per buffer: four xi_z planes (57.6 KB)
constexpr int W3_WSUB = 4 * 256;               // floats per 16-MFMA group of weights: 4 pseudo-taps x 64 lanes x 4
constexpr int W3_WCHUNK = 32 * W3_WSUB;        // floats per (chunk, slice): 64 pseudo-taps x 2 halves x 1 KB
constexpr int W3_WSKIP = 4 * W3_WSUB;          // floats per (skip chunk, slice): 8 destinations x 2 halves x 1 KB

__device__ __forceinline__ float w3_silu(float v) { return v * holo_rcp(1.0f + __expf(-v)); }

#ifdef HOLO_EMU
struct w3v4 {
  float x, y, z, w;
};
static inline w3v4 operator+(w3v4 a, w3v4 b) { return w3v4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
static inline w3v4 operator-(w3v4 a, w3v4 b) { return w3v4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
#else
typedef float w3v4 __attribute__((ext_vector_type(4)));  // arithmetic on it compiles to v_pk_add_f32 pairs
#endif
__device__ __forceinline__ w3v4 w3_ld(const float* p) { return *reinterpret_cast<const w3v4*>(p); }

// One persistent workgroup per CU.  Work list: item = ((split * ny + cout block) * ntiles + tile), dealt round robin.
// XF: the input passes through the per-(sample, channel) affine (GroupNorm folded with FiLM) and, with p.act, SiLU.
template <bool SKIP, bool XF>
__global__ __launch_bounds__(256, 1) void conv_wino3_kernel(ConvParams p) {
  __shared__ __attribute__((aligned(16))) float s_halo[2 * W3_HALO];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = HOLO_UNIFORM(tid >> 6);  // the wave owns output channels [16 wn, 16 wn + 16) of the item's 64 (scalar)
  const int lj = lane & 15;
  const int kq = lane >> 4;
  const int Cin = p.C0 + p.C1;
  const int ncc = (Cin + W3_BK - 1) / W3_BK;
  const int SCin = p.skip_C0 + p.skip_C1;
  const int ntx = p.OW >> 3, nty = p.OH >> 3, ntz = p.OD >> 1;
  const int ntiles = p.N * ntz * nty * ntx;
  const int ny = (p.Cout + 63) >> 6;
  const int nitems = ntiles * ny * p.nsplit;
  const int SD = p.ups ? (p.ID >> 1) : p.ID;
  const int SH = p.ups ? (p.IH >> 1) : p.IH;
  const int SW = p.ups ? (p.IW >> 1) : p.IW;
  const int wnsl = p.CoutP >> 4;
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;

  // ---- item / stage bookkeeping (wave-uniform).  A stage = one 32-channel chunk of one item.
  struct Item {
    int n, tz0, ty0, tx0, n0, split, cc_begin, cc_end, sk_begin, sk_end;
  };
  auto decode = [&](int it, Item& I) {
    int tile = it % ntiles;
    int rest = it / ntiles;
    I.n0 = (rest % ny) * 64;
    I.split = rest / ny;
    I.tx0 = (tile % ntx) << 3;
    tile /= ntx;
    I.ty0 = (tile % nty) << 3;
    tile /= nty;
    I.tz0 = (tile % ntz) * 2;
    I.n = tile / ntz;
    I.cc_begin = I.split * p.chunks_per_split;
    I.cc_end = min(I.cc_begin + p.chunks_per_split, ncc);
    const int nsk = SKIP ? (SCin + W3_BK - 1) / W3_BK : 0;
    I.sk_begin = min(I.split * p.skip_chunks_per_split, nsk);
    I.sk_end = min(I.sk_begin + p.skip_chunks_per_split, nsk);
  };

  // ---- producer side: item = ((y,x) column, channel quad); a thread holds the column's four planes
  const int q = tid & 7;
  w3v4 hreg[2][4];  // two items in flight: a stage's four items are requested in two halves
  unsigned h_cvalid = 0;  // bit i: column of item i lies inside the volume (y,x)
  unsigned h_zvalid = 0;  // bit pl: plane pl lies inside the volume (z)
  bool h_chvalid = false;
  int h_c = 0, h_n = 0;
  float h_a[4] = {1.f, 1.f, 1.f, 1.f}, h_b[4] = {0.f, 0.f, 0.f, 0.f};  // the chunk's affine coefficients (XF)
  auto halo_issue = [&](const Item& I, int cc, int hpart) {
    int c = cc * W3_BK + q * 4;
    h_c = c;
    h_n = I.n;
    h_chvalid = c < Cin;
    if (!h_chvalid) c = 0;  // clamped, masked at commit
    // virtual concat: C0 is a multiple of the chunk size when there is a second source, so a chunk has ONE source (scalar)
    const bool second = p.src1 != nullptr && cc * W3_BK >= p.C0;
    const float* src = second ? p.src1 : p.src0;
    const int Cs = second ? p.C1 : p.C0;
    const int cs = second ? c - p.C0 : c;
    if (XF && hpart == 0) {
      const float4* cf = reinterpret_cast<const float4*>(p.coef + ((int64_t)I.n * Cin + c) * 2);
      const float4 c01 = cf[0], c23 = cf[1];  // (a,b) interleaved per channel
      h_a[0] = c01.x, h_b[0] = c01.y, h_a[1] = c01.z, h_b[1] = c01.w, h_a[2] = c23.x, h_b[2] = c23.y, h_a[3] = c23.z, h_b[3] = c23.w;
    }
    const char* sbase = reinterpret_cast<const char*>(src + (int64_t)I.n * SD * SH * SW * Cs);
    const unsigned cbytes = (unsigned)Cs * 4u, cofs = (unsigned)cs * 4u;
    unsigned zsrc[4];
    h_zvalid = 0;
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      int z = I.tz0 + pl - 1;
      h_zvalid |= (z >= 0 && z < p.ID ? 1u : 0u) << pl;
      z = min(max(z, 0), p.ID - 1);
      if (p.ups) z >>= 1;
      zsrc[pl] = (unsigned)(z * SH * SW);
    }
    if (hpart == 0) h_cvalid = 0;
#pragma unroll
    for (int i = 2 * hpart; i < 2 * hpart + 2; ++i) {
      const int col = min((tid >> 3) + 32 * i, W3_PLANE - 1);
      const int hy = col / W3_HX, hx = col - hy * W3_HX;
      int y = I.ty0 + hy - 1, x = I.tx0 + hx - 1;
      const bool ok = y >= 0 && y < p.IH && x >= 0 && x < p.IW;
      y = min(max(y, 0), p.IH - 1);
      x = min(max(x, 0), p.IW - 1);
      if (p.ups) {
        y >>= 1;
        x >>= 1;
      }
      h_cvalid |= (ok ? 1u : 0u) << i;
      const unsigned yx = (unsigned)(y * SW + x);
      // unconditional loads from clamped addresses, masked at commit; uniform base + 32-bit byte offsets
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) hreg[i & 1][pl] = w3_ld(reinterpret_cast<const float*>(sbase + ((zsrc[pl] + yx) * cbytes + cofs)));
    }
  };
  // activation, zero padding (AFTER the activation), input transform along z, one item -> four 16-byte LDS writes.
  // Branch free: the lanes of item 3 beyond column 99 hold (and write) a copy of column 99's values.
  auto halo_commit = [&](int i, float* buf) {
    const int col = min((tid >> 3) + 32 * i, W3_PLANE - 1);
    const bool act = p.act != 0;
    float v[4][4];
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      const w3v4 h = hreg[i & 1][pl];
      v[pl][0] = h.x, v[pl][1] = h.y, v[pl][2] = h.z, v[pl][3] = h.w;
      const bool keep = h_chvalid && ((h_cvalid >> i) & 1u) && ((h_zvalid >> pl) & 1u);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = v[pl][e];
        if (XF) {
          t = fmaf(t, h_a[e], h_b[e]);
          const float sl = w3_silu(t);
          t = act ? sl : t;
        }
        v[pl][e] = keep ? t : 0.f;
      }
    }
    // B^T d along z: xi0 = d0 - d2, xi1 = d1 + d2, xi2 = d2 - d1, xi3 = d1 - d3
    float* dst = buf + col * W3_RS + q * 4;
    *reinterpret_cast<float4*>(dst + 0 * W3_PLANE * W3_RS) =
        make_float4(v[0][0] - v[2][0], v[0][1] - v[2][1], v[0][2] - v[2][2], v[0][3] - v[2][3]);
    *reinterpret_cast<float4*>(dst + 1 * W3_PLANE * W3_RS) =
        make_float4(v[1][0] + v[2][0], v[1][1] + v[2][1], v[1][2] + v[2][2], v[1][3] + v[2][3]);
    *reinterpret_cast<float4*>(dst + 2 * W3_PLANE * W3_RS) =
        make_float4(v[2][0] - v[1][0], v[2][1] - v[1][1], v[2][2] - v[1][2], v[2][3] - v[1][3]);
    *reinterpret_cast<float4*>(dst + 3 * W3_PLANE * W3_RS) =
        make_float4(v[1][0] - v[3][0], v[1][1] - v[3][1], v[1][2] - v[3][2], v[1][3] - v[3][3]);
  };

  // ---- consumer side
  f32x4 acc[64];  // [xi_z][xi_y][xi_x]; register r of a lane: y tile r, x tile kq, output channel lj
  // MFMA row lj = (y tile lj & 3, x tile lj >> 2); the lane's k group kq holds channels 4 kq + 16 half .. +3
  const int a_off = ((2 * (lj & 3)) * W3_HX + 2 * (lj >> 2)) * W3_RS + kq * 4;
  w3v4 P[4][4];  // the lane's 4 x 4 (halo row, halo column) patch of one (xi_z, half); x-transformed in place
  auto load_patch = [&](const float* buf, int xz, int half) {
    const float* base = buf + a_off + (xz * W3_PLANE) * W3_RS + half * 16;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) P[a][b] = w3_ld(base + (a * W3_HX + b) * W3_RS);
  };
  auto xform_rows = [&]() {  // B^T along x, in place: 32 v_pk_add_f32
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const w3v4 t0 = P[a][0] - P[a][2], t1 = P[a][1] + P[a][2], t2 = P[a][2] - P[a][1], t3 = P[a][1] - P[a][3];
      P[a][0] = t0, P[a][1] = t1, P[a][2] = t2, P[a][3] = t3;
    }
  };
  auto yform = [&](w3v4 (&Y)[4], int xy) {  // B^T along y for one xi_y: 8 v_pk_add_f32
#pragma unroll
    for (int b = 0; b < 4; ++b)
      Y[b] = xy == 0 ? P[0][b] - P[2][b] : xy == 1 ? P[1][b] + P[2][b] : xy == 2 ? P[2][b] - P[1][b] : P[1][b] - P[3][b];
  };
  // the four accumulators of a group advance together, k-step by k-step: consecutive MFMAs are independent.  The
  // accumulators are TIED to their AGPR tuples (HOLO_MFMA16_ACC, holo_common.h): 64 sets fill the accumulation file.
  auto mfma16 = [&](f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, const w3v4 (&A)[4], const w3v4 (&B)[4]) {
    HOLO_MFMA16_ACC_FIRST(c0, A[0].x, B[0].x);
    HOLO_MFMA16_ACC(c1, A[1].x, B[1].x);
    HOLO_MFMA16_ACC(c2, A[2].x, B[2].x);
    HOLO_MFMA16_ACC(c3, A[3].x, B[3].x);
    HOLO_MFMA16_ACC(c0, A[0].y, B[0].y);
    HOLO_MFMA16_ACC(c1, A[1].y, B[1].y);
    HOLO_MFMA16_ACC(c2, A[2].y, B[2].y);
    HOLO_MFMA16_ACC(c3, A[3].y, B[3].y);
    HOLO_MFMA16_ACC(c0, A[0].z, B[0].z);
    HOLO_MFMA16_ACC(c1, A[1].z, B[1].z);
    HOLO_MFMA16_ACC(c2, A[2].z, B[2].z);
    HOLO_MFMA16_ACC(c3, A[3].z, B[3].z);
    HOLO_MFMA16_ACC(c0, A[0].w, B[0].w);
    HOLO_MFMA16_ACC(c1, A[1].w, B[1].w);
    HOLO_MFMA16_ACC(c2, A[2].w, B[2].w);
    HOLO_MFMA16_ACC(c3, A[3].w, B[3].w);
  };
  w3v4 Bw[4][4];  // ring of weight groups: group g lives in slot g & 3 and is requested two groups (32 MFMAs) ahead (three
                  // groups are live at a time; four names because a chunk's 32 groups must map onto whole ring turns)
  auto load_w = [&](int slot, const float* wp) {
#pragma unroll
    for (int t = 0; t < 4; ++t) Bw[slot][t] = w3_ld(wp + t * 256);
  };

  unsigned long long* dbg = p.dbg ? p.dbg + (int64_t)blockIdx.x * 8 : nullptr;
  if (dbg && tid == 0) dbg[0] = HOLO_PROBE_CLOCK();

  int it = blockIdx.x;
  if (it >= nitems) return;
  Item cur, nxt;
  decode(it, cur);
  // weights of (item, chunk): this wave's 16-Cout slice
  auto w_of = [&](const Item& I, int cc) {
    return p.w_wino3 + ((int64_t)cc * wnsl + (I.n0 >> 4) + wn) * W3_WCHUNK + lane * 4;
  };

  // ---- prologue: the first stage's halo, exposed once per workgroup
  int stage = 0;  // parity = LDS buffer of the current stage
  int cc = cur.cc_begin;
#pragma unroll
  for (int g = 0; g < 2; ++g) load_w(g, w_of(cur, cc) + g * W3_WSUB);
#pragma unroll
  for (int hp = 0; hp < 2; ++hp) {
    halo_issue(cur, cc, hp);
    halo_commit(2 * hp, s_halo);
    halo_commit(2 * hp + 1, s_halo);
  }
  __syncthreads();
  if (dbg && tid == 0) dbg[1] = HOLO_PROBE_CLOCK();

  for (;;) {
    // ---------------- one item
#pragma unroll
    for (int s = 0; s < 64; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[s][r] = 0.f;
    for (cc = cur.cc_begin; cc < cur.cc_end; ++cc, ++stage) {
      // what the producer side prepares during this stage: the next chunk of this item, or the first chunk of the next
      // item.  The stage body is ONE basic block (the 256 accumulators then meet the register allocator at the loop header
      // only): on the very last stage of the workgroup the producer side simply stages the current chunk again.
      const bool last_chunk = cc + 1 == cur.cc_end;
      int ncc_ = cc + 1;
      nxt = cur;
      if (last_chunk) {
        const int nit = it + (int)gridDim.x;
        if (nit < nitems) {
          decode(nit, nxt);
          ncc_ = nxt.cc_begin;
        } else {
          ncc_ = cc;
        }
      }
      const float* buf = s_halo + (stage & 1) * W3_HALO;
      float* obuf = s_halo + ((stage + 1) & 1) * W3_HALO;
      const float* wp = w_of(cur, cc);
      const float* wnext = w_of(nxt, ncc_);
      // fused skip chunk riding on this stage (chunk j of the skip rides on main chunk j; leftovers on the last)
      load_patch(buf, 0, 0);
      halo_issue(nxt, ncc_, 0);
      xform_rows();
#pragma unroll
      for (int st = 0; st < 8; ++st) {  // fully unrolled: accumulator sets and ring slots are compile-time choices
        const int xz = st >> 1;
#pragma unroll
        for (int xy = 0; xy < 4; ++xy) {
          const int g = st * 4 + xy;
          w3v4 Y[4];
          yform(Y, xy);
          if (xy == 3 && st < 7) load_patch(buf, (st + 1) >> 1, (st + 1) & 1);  // P is dead: the next patch flies under 16 MFMAs
          if (g + 2 < 32)
            load_w((g + 2) & 3, wp + (g + 2) * W3_WSUB);
          else
            load_w((g + 2) & 3, wnext + (g + 2 - 32) * W3_WSUB);
          __builtin_amdgcn_sched_barrier(0);  // requests stay AHEAD of the MFMAs that hide them
          mfma16(acc[xz * 16 + xy * 4 + 0], acc[xz * 16 + xy * 4 + 1], acc[xz * 16 + xy * 4 + 2], acc[xz * 16 + xy * 4 + 3], Y, Bw[g & 3]);
          __builtin_amdgcn_sched_barrier(0);
          // the next stage's halo in two halves: items 0,1 (requested at the start of the stage) are committed in steps
          // 2,3, then items 2,3 are requested and committed in steps 5,6 (~130 MFMAs after their request)
          if (xy == 0) {
            if (st == 2) halo_commit(0, obuf);
            if (st == 3) {
              halo_commit(1, obuf);
              halo_issue(nxt, ncc_, 1);
            }
            if (st == 5) halo_commit(2, obuf);
            if (st == 6) halo_commit(3, obuf);
          }
        }
        if (st < 7) xform_rows();
      }
      __syncthreads();  // this stage's buffer is free, the next stage's is complete
    }
    // ---------------- fused 1x1x1 skip connection: raw block input at the lane's own voxels -> pseudo-taps {0,3}^3
    if (SKIP) {
      const int yt = lj & 3, xt = lj >> 2;
      const int64_t vbase = (((int64_t)cur.n * p.OD + cur.tz0) * p.OH + cur.ty0 + 2 * yt) * p.OW + cur.tx0 + 2 * xt;
      // group sg = (skip chunk, half, dz): four voxels (dy,dx) of the lane's patch = one 16-MFMA group; operands of group
      // sg + 1 are requested before the MFMAs of group sg (two buffers; dz = sg & 1 keeps the accumulator choice static)
      const int nsg = (cur.sk_end - cur.sk_begin) * 4;
      w3v4 SA[2][4], SB[2][4];
      auto skip_load = [&](int sg, w3v4 (&A)[4], w3v4 (&B)[4]) {
        const int sc = cur.sk_begin + (sg >> 2), half = (sg >> 1) & 1, dz = sg & 1;
        const float* swp = p.skip_w_wino3 + ((int64_t)sc * wnsl + (cur.n0 >> 4) + wn) * W3_WSKIP + lane * 4;
        int c = sc * W3_BK + half * 16 + kq * 4;
        if (c >= SCin) c = 0;  // (the packed weights of padding channels are zero)
        const bool second = c >= p.skip_C0;
        const float* sp = second ? p.skip_src1 : p.skip_src0;
        const int Cs = second ? p.skip_C1 : p.skip_C0;
        const int cs = second ? c - p.skip_C0 : c;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int64_t v = vbase + ((int64_t)dz * p.OH + (d >> 1)) * p.OW + (d & 1);
          A[d] = w3_ld(sp + v * Cs + cs);
          B[d] = w3_ld(swp + ((half * 2 + dz) * 4 + d) * 256);
        }
      };
      if (nsg > 0) skip_load(0, SA[0], SB[0]);
      for (int sg = 0; sg < nsg; sg += 2) {
        skip_load(sg + 1, SA[1], SB[1]);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(acc[0], acc[3], acc[12], acc[15], SA[0], SB[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (sg + 2 < nsg) skip_load(sg + 2, SA[0], SB[0]);
        __builtin_amdgcn_sched_barrier(0);
        mfma16(acc[48], acc[51], acc[60], acc[63], SA[1], SB[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (dbg && tid == 0) dbg[2] = HOLO_PROBE_CLOCK();
    HOLO_MFMA_DRAIN();  // the last MFMAs' results, before vector instructions read the accumulators

    // ---------------- output transform (lane-local: x, y, z) + epilogue.  D row 4 kq + r = (y tile r, x tile kq)
    {
      const int co = cur.n0 + wn * 16 + lj;
      const int coc = co < p.Cout ? co : p.Cout - 1;
      const bool direct = p.nsplit == 1;
      float bv = (direct && p.bias) ? p.bias[coc] : 0.f;
      if (direct && p.skip_bias) bv += p.skip_bias[coc];
      const int64_t tbase = ((((int64_t)cur.n * p.OD + cur.tz0) * p.OH + cur.ty0) * p.OW + cur.tx0 + 2 * kq) * p.Cout;
      const int64_t zstride = (int64_t)p.OH * p.OW * p.Cout;
      const int ystride = p.OW * p.Cout;
      float ssum = 0.f, ssq = 0.f;
      float* obase = direct ? p.out : p.partial + (int64_t)cur.split * M * p.Cout;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float res[2][2][2];
        const int64_t vo = tbase + (int64_t)(2 * r) * ystride + coc;
        if (direct && p.residual) {
#pragma unroll
          for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
              for (int dx = 0; dx < 2; ++dx) res[dz][dy][dx] = p.residual[vo + dz * zstride + dy * ystride + dx * p.Cout];
        }
        float oz[4][2][2];  // [xi_z][dy][dx]
#pragma unroll
        for (int xz = 0; xz < 4; ++xz) {
          float oy[4][2];  // [xi_y][dx]
#pragma unroll
          for (int xy = 0; xy < 4; ++xy) {
            const float m0 = acc[xz * 16 + xy * 4 + 0][r], m1 = acc[xz * 16 + xy * 4 + 1][r], m2 = acc[xz * 16 + xy * 4 + 2][r],
                        m3 = acc[xz * 16 + xy * 4 + 3][r];
            oy[xy][0] = (m0 + m1) + m2;
            oy[xy][1] = (m1 - m2) - m3;
          }
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            oz[xz][0][dx] = (oy[0][dx] + oy[1][dx]) + oy[2][dx];
            oz[xz][1][dx] = (oy[1][dx] - oy[2][dx]) - oy[3][dx];
          }
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            float o0 = (oz[0][dy][dx] + oz[1][dy][dx]) + oz[2][dy][dx];
            float o1 = (oz[1][dy][dx] - oz[2][dy][dx]) - oz[3][dy][dx];
            if (direct) {
              if (p.residual) {
                o0 += res[0][dy][dx];
                o1 += res[1][dy][dx];
              }
              o0 += bv;
              o1 += bv;
              ssum += o0 + o1;
              ssq += o0 * o0 + o1 * o1;
            }
            if (co < p.Cout) {
              obase[vo + dy * ystride + dx * p.Cout] = o0;
              obase[vo + zstride + dy * ystride + dx * p.Cout] = o1;
            }
          }
      }
      // GroupNorm statistics of the tensor just produced: one slab per tile (conv_stats_slabs)
      if (p.stats && direct) {
        ssum += __shfl_xor(ssum, 16);
        ssq += __shfl_xor(ssq, 16);
        ssum += __shfl_xor(ssum, 32);
        ssq += __shfl_xor(ssq, 32);
        if (kq == 0 && co < p.Cout) {
          const int tiles_per_sample = ntx * nty * ntz;
          const int slab = (it % ntiles) % tiles_per_sample;
          double* d = p.stats + (((int64_t)cur.n * tiles_per_sample + slab) * p.Cout + co) * 2;
          d[0] = (double)ssum;
          d[1] = (double)ssq;
        }
      }
    }
    if (dbg && tid == 0) {
      dbg[3] = HOLO_PROBE_CLOCK();
      dbg[7] += 1;
    }
    it += (int)gridDim.x;
    if (it >= nitems) break;
    cur = nxt;
  }
}

// OIDHW [Cout][Cin][27] -> the 64 pseudo-taps U = (G x G x G) g (float64, rounded once) in the wave's consumption order
//   [chunk][slice][xi_z][half][xi_y][xi_x][lane = 16 kq + lj][e]:  output channel 16 slice + lj, input channel
//   32 chunk + 16 half + 4 kq + e.   src_taps == 1 (a ResBlock's 1x1x1 skip_connection): 8 signed copies,
//   [chunk][slice][half][dz][dy][dx][lane][e] = (-1)^(dz+dy+dx) w.
__global__ __launch_bounds__(256) void repack_conv_weight_wino3_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                                       int Cout, int Cin, int src_taps, int CoutP, int CinP) {
  const int per = src_taps == 27 ? W3_WCHUNK : W3_WSKIP;
  const int nsl = CoutP >> 4;
  const int64_t total = (int64_t)(CinP >> 5) * nsl * per;
  const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3);
    const int lj = (int)((i >> 2) & 15);
    const int kq = (int)((i >> 6) & 3);
    int g = (int)((i % per) >> 8);  // pseudo-tap slot inside the (chunk, slice) block
    const int64_t blk = i / per;
    const int slice = (int)(blk % nsl);
    const int cc = (int)(blk / nsl);
    const int co = slice * 16 + lj;
    float v = 0.f;
    if (src_taps == 27) {
      const int xx = g & 3, xy = (g >> 2) & 3, half = (g >> 4) & 1, xz = g >> 5;
      const int ci = cc * 32 + half * 16 + kq * 4 + e;
      if (ci < Cin && co < Cout) {
        const float* src = w + ((int64_t)co * Cin + ci) * 27;
        double u = 0.0;
        for (int kz = 0; kz < 3; ++kz)
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) u += G[xz][kz] * G[xy][ky] * G[xx][kx] * (double)holo_ld_sys(src + kz * 9 + ky * 3 + kx);
        v = (float)u;
      }
    } else {
      const int d = g & 3, dz = (g >> 2) & 1, half = g >> 3;
      const int ci = cc * 32 + half * 16 + kq * 4 + e;
      if (ci < Cin && co < Cout) {
        const float s = ((dz + (d >> 1) + (d & 1)) & 1) ? -1.f : 1.f;
        v = s * holo_ld_sys(w + (int64_t)co * Cin + ci);
      }
    }
    out[i] = v;
  }
}

}  // namespace

int64_t conv_wino3_weight_floats(int CoutP, int CinP, int src_taps) {
  return (int64_t)(CinP >> 5) * (CoutP >> 4) * (src_taps == 27 ? W3_WCHUNK : W3_WSKIP);
}

int repack_conv_weight_wino3_launch(const float* w, float* out, int Cout, int Cin, int src_taps, int CoutP, int CinP,
                                    void* stream) {
  const int64_t total = conv_wino3_weight_floats(CoutP, CinP, src_taps);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  HOLO_LAUNCH(repack_conv_weight_wino3_kernel, dim3((unsigned)blocks), dim3(256), stream, w, out, Cout, Cin, src_taps, CoutP,
              CinP);
  return 0;
}

// p.wino == 3 (conv_plan): p.grid_x persistent workgroups
int conv_wino3_launch(const ConvParams& p, void* stream) {
  if (!p.w_wino3 || (p.skip_w && !p.skip_w_wino3) || (p.OD & 1) || (p.OH & 7) || (p.OW & 7) || (p.Cout & 63)) {
    set_error("conv_wino3_launch: unsupported shape / weights not prepared");
    return -1;
  }
  const dim3 grid((unsigned)p.grid_x), block(256);
  if (p.skip_w && p.coef) {
    HOLO_LAUNCH((conv_wino3_kernel<true, true>), grid, block, stream, p);
  } else if (p.skip_w) {
    HOLO_LAUNCH((conv_wino3_kernel<true, false>), grid, block, stream, p);
  } else if (p.coef) {
    HOLO_LAUNCH((conv_wino3_kernel<false, true>), grid, block, stream, p);
  } else {
    HOLO_LAUNCH((conv_wino3_kernel<false, false>), grid, block, stream, p);
  }
  return 0;
}

}  // namespace holo
