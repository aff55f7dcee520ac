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

#include <type_traits>
#include <utility>

#include "holo_common.h"
#include "holo_kernels.h"

// Development probes (tools/conv_ab builds extra copies of this file with -DW3_PROBE=<bits> -DW3_ENTRY=<name>): 1 no halo
// requests / commits inside the stage loop, 2 no weight requests, 4 no patch reads / input transforms, 8 no MFMAs, 16 weight
// requests into registers nobody waits for, 32 halo requests but no commits, 64 commits without the activation, 128 the work
// list in eight XCD lanes (see the kernel).
#ifndef W3_PROBE
#define W3_PROBE 0
#endif
#ifndef W3_WDIST
#define W3_WDIST 3  // weight requests run this many 16-MFMA groups ahead (2: measured ~190 cycles of wait per group)
#endif
#ifndef W3_ENTRY
#define W3_ENTRY conv_wino3_launch
#endif

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

// 16 bytes as two register pairs: arithmetic on the pairs is v_pk_add_f32 / v_pk_fma_f32 (one instruction per two values:
// with ONE wave per SIMD every instruction costs an issue slot of ~4-5 cycles, whatever it does)
struct w3q {
  f32x2 lo, hi;
};
__device__ __forceinline__ w3q w3_ld(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return w3q{f32x2{v.x, v.y}, f32x2{v.z, v.w}};
}
// Buffer addressing (scalar resource + 32-bit vector offset + 32-bit scalar offset): a request then needs NO vector
// instruction for its address - a vector instruction between two MFMAs costs ~15 cycles (tools/mfma_shadow_probe.cpp).
// The whole tensor must lie within 4 GB of its base (conv_plan checks).
#ifdef HOLO_EMU
struct w3_rsrc {
  const char* base;
};
static inline w3_rsrc w3_make_rsrc(const void* p) { return w3_rsrc{reinterpret_cast<const char*>(p)}; }
static inline w3q w3_bld(const w3_rsrc& r, unsigned voff, unsigned soff) {
  return w3_ld(reinterpret_cast<const float*>(r.base + (size_t)voff + (size_t)soff));
}
#else
typedef __amdgpu_buffer_rsrc_t w3_rsrc;
typedef unsigned w3u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ w3_rsrc w3_make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ w3q w3_bld(w3_rsrc r, unsigned voff, unsigned soff) {
  const w3u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return w3q{f32x2{__uint_as_float(v.x), __uint_as_float(v.y)}, f32x2{__uint_as_float(v.z), __uint_as_float(v.w)}};
}
#endif
__device__ __forceinline__ w3q w3_add(const w3q& a, const w3q& b) { return w3q{pk_add(a.lo, b.lo), pk_add(a.hi, b.hi)}; }
__device__ __forceinline__ w3q w3_sub(const w3q& a, const w3q& b) { return w3q{pk_sub(a.lo, b.lo), pk_sub(a.hi, b.hi)}; }

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): the 16 MFMA slots of a group as straight-line code with
// compile-time slot numbers (a `#pragma unroll` loop over them inside the two unrolled group loops is refused by the unroller)
template <class F, int... K>
__device__ __forceinline__ void w3_static_for(F&& f, std::integer_sequence<int, K...>) {
  (f(std::integral_constant<int, K>{}), ...);
}

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
  // Order of the work list: item q = (x tile, z tile, y tile, sample, cout block, split), x tile fastest - the 256 items in
  // flight are one row of y tiles through all z.  Tried against it (round 4): y before z (a slab of 8 contiguous planes in flight)
  // and eight XCD lanes (workgroup b, on XCD b % 8 by round-robin dispatch, owns the tile rows ty = b (mod 8), so that x / z
  // neighbours meet in ONE L2): all three give the same step time (150.1 ... 150.6 denoise-steps/s) although the fabric sees
  // every halo voxel 3.8 times (profiles/pmc_traffic.json) - at 1.8 TB/s the re-reads are served by the memory-side cache and
  // hidden behind the MFMAs.  The lanes stay as a probe (W3_PROBE & 128).
  const int X = ((W3_PROBE & 128) && (nty & 7) == 0 && (gridDim.x & 7) == 0) ? 8 : 1;
  const int ntyh = nty / X;
  const int lane_x = (int)blockIdx.x % X, G = (int)gridDim.x / X, nq = nitems / X;
  auto chunks_of = [&](Item& I) {
    I.cc_begin = I.split * p.chunks_per_split;
    I.cc_end = min(I.cc_begin + p.chunks_per_split, ncc);
    const int nsk = SKIP ? (SCin + W3_BK - 1) / W3_BK : 0;
    I.sk_begin = min(I.split * p.skip_chunks_per_split, nsk);
    I.sk_end = min(I.sk_begin + p.skip_chunks_per_split, nsk);
  };
  auto decode = [&](int qi, Item& I, int lane_ofs) {
    int t = qi;
    I.tx0 = (t % ntx) << 3;
    t /= ntx;
    I.tz0 = (t % ntz) * 2;
    t /= ntz;
    I.ty0 = ((t % ntyh) * X + lane_ofs) << 3;
    t /= ntyh;
    I.n = t % p.N;
    t /= p.N;
    I.n0 = (t % ny) * 64;
    I.split = t / ny;
    chunks_of(I);
  };
  // The workgroup's next item = this one + G in its lane's list: added digit by digit in the mixed radix - a handful of scalar
  // adds and selects per item instead of six integer divisions (a uniform division is ~25 instructions, several of them on the
  // vector pipe the MFMAs run on).
  Item stride;  // G in the same digits (scaled like an item's; no lane offset)
  decode(G % (nq > 0 ? nq : 1), stride, 0);  // (a grid >= the work list never advances)
  auto advance = [&](const Item& a, Item& I) {
    int tx = a.tx0 + stride.tx0, c = tx >= p.OW ? 1 : 0;
    I.tx0 = tx - (c ? p.OW : 0);
    int tz = a.tz0 + stride.tz0 + 2 * c;
    c = tz >= p.OD ? 1 : 0;
    I.tz0 = tz - (c ? p.OD : 0);
    int ty = a.ty0 + stride.ty0 + 8 * X * c;
    c = ty >= p.OH ? 1 : 0;
    I.ty0 = ty - (c ? p.OH : 0);
    int n = a.n + stride.n + c;
    c = n >= p.N ? 1 : 0;
    I.n = n - (c ? p.N : 0);
    int nb = a.n0 + stride.n0 + 64 * c;
    c = nb >= 64 * ny ? 1 : 0;
    I.n0 = nb - (c ? 64 * ny : 0);
    I.split = a.split + stride.split + c;  // (callers advance only while the item number stays below nq)
    chunks_of(I);
  };

  // ---- producer side: item = ((y,x) column, channel quad); a thread holds the column's four planes.  A stage's four items
  //      are requested in two halves (two items = 8 x 16 bytes in flight) and committed piece by piece (commit_piece).
  const int q = tid & 7;
  w3q hrA[4], hrB[4];  // items 0, 2 / items 1, 3 (two separate arrays: never indexed by anything but literals)
  unsigned h_off[4];       // byte offset of item i's column and channel quad inside a source plane (clamped)
  unsigned h_cvalid = 0;   // bit i: column of item i lies inside the volume (y,x)
  unsigned h_zvalid = 0;   // bit pl: plane pl lies inside the volume (z)
  unsigned h_zoff[4];      // byte offset of source plane pl of the chunk's sample (clamped; wave-uniform: scalar registers)
  const w3_rsrc rsrc0 = w3_make_rsrc(p.src0), rsrc1 = w3_make_rsrc(p.src1 ? p.src1 : p.src0);
  bool h_second = false;   // the chunk's source
  bool h_chvalid = false;
  float4 h_c01 = make_float4(1.f, 0.f, 1.f, 0.f), h_c23 = h_c01;  // the chunk's affine (XF): (a,b) of the thread's 4 channels
  // Addresses, masks and coefficients of a stage's halo (no requests yet).  Scalar work (source, planes) and the requests of
  // the coefficients first; the vector part is the four column offsets.
  auto halo_setup = [&](const Item& I, int cc) {
    int c = cc * W3_BK + q * 4;
    h_chvalid = c < Cin;
    if (!h_chvalid) c = 0;  // clamped, masked at commit
    // virtual concat: C0 is a multiple of the chunk size when there is a second source, so a chunk has ONE source (scalar)
    const bool second = p.src1 != nullptr && cc * W3_BK >= p.C0;
    const int Cs = second ? p.C1 : p.C0;
    const int cs = second ? c - p.C0 : c;
    if (XF) {
      const float4* cf = reinterpret_cast<const float4*>(p.coef + ((int64_t)I.n * Cin + c) * 2);
      h_c01 = cf[0], h_c23 = cf[1];  // (a,b) interleaved per channel; first used by the commit ~8 groups later
    }
    const unsigned cbytes = (unsigned)Cs * 4u;
    h_second = second;
    h_zvalid = 0;
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) {
      int z = I.tz0 + pl - 1;
      h_zvalid |= (z >= 0 && z < p.ID ? 1u : 0u) << pl;
      z = min(max(z, 0), p.ID - 1) >> p.ups;
      h_zoff[pl] = (unsigned)((I.n * SD + z) * SH * SW) * cbytes;
    }
    h_cvalid = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = min((tid >> 3) + 32 * i, W3_PLANE - 1);
      const int hy = col / W3_HX, hx = col - hy * W3_HX;
      int y = I.ty0 + hy - 1, x = I.tx0 + hx - 1;
      const bool ok = y >= 0 && y < p.IH && x >= 0 && x < p.IW;
      y = min(max(y, 0), p.IH - 1) >> p.ups;
      x = min(max(x, 0), p.IW - 1) >> p.ups;
      h_cvalid |= (ok ? 1u : 0u) << i;
      h_off[i] = (unsigned)(y * SW + x) * cbytes + (unsigned)cs * 4u;
    }
  };
  // one 16-byte request: item i, plane pl (unconditional, from a clamped address; masked at commit): scalar base + 32-bit
  // vector offset, no address arithmetic at the request
  auto halo_load = [&](int i, int pl) {
    const w3q v = h_second ? w3_bld(rsrc1, h_off[i], h_zoff[pl]) : w3_bld(rsrc0, h_off[i], h_zoff[pl]);
    if (i & 1)
      hrB[pl] = v;
    else
      hrA[pl] = v;
  };
  // Commit of item i: affine + SiLU + zero padding (AFTER the activation) of its 16 values in place - sixteen independent
  // chains, so the in-order wave always has something to issue -, then the input transform along z (xi0 = d0 - d2,
  // xi1 = d1 + d2, xi2 = d2 - d1, xi3 = d1 - d3) and four 16-byte LDS writes.  Branch free: the lanes of item 3 beyond
  // column 99 hold (and write) a copy of column 99's values.
  auto commit_item_on = [&](w3q(&H)[4], int i, float* buf) {
    // stage by stage over the eight register pairs, each stage pinned behind the previous one: a dependent instruction is
    // then always eight instructions away from its producer (the in-order wave has no other wave to fill a latency with)
    f32x2* V[8] = {&H[0].lo, &H[0].hi, &H[1].lo, &H[1].hi, &H[2].lo, &H[2].hi, &H[3].lo, &H[3].hi};
    if (XF && !(W3_PROBE & 64)) {
      f32x2 e[8];
      const f32x2 ca[2] = {f32x2{h_c01.x, h_c01.z}, f32x2{h_c23.x, h_c23.z}}, cb[2] = {f32x2{h_c01.y, h_c01.w}, f32x2{h_c23.y, h_c23.w}};
#pragma unroll
      for (int j = 0; j < 8; ++j) *V[j] = pk_fma(*V[j], ca[j & 1], cb[j & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = pk_mul(*V[j], f32x2{-1.4426950408889634f, -1.4426950408889634f});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = f32x2{holo_exp2(e[j].x), holo_exp2(e[j].y)};  // exp(-t), v_exp_f32
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = pk_add(e[j], f32x2{1.f, 1.f});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = f32x2{holo_rcp(e[j].x), holo_rcp(e[j].y)};  // v_rcp_f32 (1 ulp)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) *V[j] = pk_mul(*V[j], e[j]);  // SiLU = t / (1 + exp(-t))  (XF implies p.act: conv_plan)
      __builtin_amdgcn_sched_barrier(0);
    }
    const float kc = (h_chvalid && ((h_cvalid >> i) & 1u)) ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float keep = ((h_zvalid >> (j >> 1)) & 1u) ? kc : 0.f;  // zero padding AFTER the activation
      *V[j] = pk_mul(*V[j], f32x2{keep, keep});
    }
    const int col = min((tid >> 3) + 32 * i, W3_PLANE - 1);
    float* dst = buf + col * W3_RS + q * 4;
    const w3q o0 = w3_sub(H[0], H[2]), o1 = w3_add(H[1], H[2]), o2 = w3_sub(H[2], H[1]), o3 = w3_sub(H[1], H[3]);
    *reinterpret_cast<float4*>(dst + 0 * W3_PLANE * W3_RS) = make_float4(o0.lo.x, o0.lo.y, o0.hi.x, o0.hi.y);
    *reinterpret_cast<float4*>(dst + 1 * W3_PLANE * W3_RS) = make_float4(o1.lo.x, o1.lo.y, o1.hi.x, o1.hi.y);
    *reinterpret_cast<float4*>(dst + 2 * W3_PLANE * W3_RS) = make_float4(o2.lo.x, o2.lo.y, o2.hi.x, o2.hi.y);
    *reinterpret_cast<float4*>(dst + 3 * W3_PLANE * W3_RS) = make_float4(o3.lo.x, o3.lo.y, o3.hi.x, o3.hi.y);
  };
  auto commit_item = [&](int i, float* buf) {
    if (i & 1)
      commit_item_on(hrB, i, buf);
    else
      commit_item_on(hrA, i, buf);
  };

  // ---- consumer side
  f32x4 acc[64];  // [xi_z][xi_y][xi_x]; register r of a lane: y tile r, x tile kq, output channel lj
  // MFMA row lj = (y tile lj & 3, x tile lj >> 2); the lane's k group kq holds channels 4 kq + 16 half .. +3
  const int a_off = ((2 * (lj & 3)) * W3_HX + 2 * (lj >> 2)) * W3_RS + kq * 4;
  w3q P[4][4];  // the lane's 4 x 4 (halo row, halo column) patch of one step = (xi_z, half); x-transformed in place
  w3q Y[2][4];  // A operands of a group (one xi_y, four xi_x): group g reads Y[g & 1] while Y[(g + 1) & 1] is being formed
  auto load_patch_row = [&](const float* buf, int step, int a) {
    const float* base = buf + a_off + ((step >> 1) * W3_PLANE + a * W3_HX) * W3_RS + (step & 1) * 16;
#pragma unroll
    for (int b = 0; b < 4; ++b) P[a][b] = w3_ld(base + b * W3_RS);
  };
  // B^T along x of row a, in place, in two halves: (xi_x 0, 3) then (xi_x 1, 2); 4 v_pk_add_f32 each
  auto xform_row_half = [&](int a, int h) {
    if (h == 0) {
      const w3q t0 = w3_sub(P[a][0], P[a][2]), t3 = w3_sub(P[a][1], P[a][3]);
      P[a][0] = t0, P[a][3] = t3;
    } else {
      const w3q t1 = w3_add(P[a][1], P[a][2]), t2 = w3_sub(P[a][2], P[a][1]);
      P[a][1] = t1, P[a][2] = t2;
    }
  };
  // B^T along y for (xi_y, column b): 2 v_pk_add_f32
  auto yform1 = [&](int xy, int b) {
    return xy == 0 ? w3_sub(P[0][b], P[2][b]) : xy == 1 ? w3_add(P[1][b], P[2][b]) : xy == 2 ? w3_sub(P[2][b], P[1][b]) : w3_sub(P[1][b], P[3][b]);
  };
  w3q Bdummy[4] = {};  // (W3_PROBE & 16)
  w3q Bw[4][4];  // ring of weight groups: group g lives in slot g & 3 and is requested W3_WDIST groups ahead
  // MFMA k of a group: k-step e = k >> 2 (the component of the 16-byte operands), pseudo-tap t = k & 3.  The accumulators are
  // TIED to their AGPR tuples (HOLO_MFMA16_ACC, holo_common.h): 64 sets fill the accumulation file.
  auto mfma1 = [&](f32x4& c, const w3q& A, const w3q& B, int e, bool first) {
#if W3_PROBE & 8
    HOLO_SINK8(A.lo.x, A.lo.y, A.hi.x, A.hi.y, B.lo.x, B.lo.y, B.hi.x, B.hi.y);
    return;
#endif
    const float a = e == 0 ? A.lo.x : e == 1 ? A.lo.y : e == 2 ? A.hi.x : A.hi.y;
    const float b = e == 0 ? B.lo.x : e == 1 ? B.lo.y : e == 2 ? B.hi.x : B.hi.y;
    if (first)
      HOLO_MFMA16_ACC_FIRST(c, a, b);
    else
      HOLO_MFMA16_ACC(c, a, b);
  };

  // a whole group at once (the fused skip): the four accumulators advance together, k-step by k-step
  auto mfma16 = [&](f32x4& c0, f32x4& c1, f32x4& c2, f32x4& c3, const w3q (&A)[4], const w3q (&B)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mfma1(c0, A[0], B[0], e, e == 0);
      mfma1(c1, A[1], B[1], e, false);
      mfma1(c2, A[2], B[2], e, false);
      mfma1(c3, A[3], B[3], e, false);
    }
  };

  unsigned long long* dbg = p.dbg ? p.dbg + (int64_t)blockIdx.x * 8 : nullptr;
  if (dbg && tid == 0) dbg[0] = HOLO_PROBE_CLOCK();

  int it = (int)blockIdx.x / X;  // position in the lane's list
  if (it >= nq) return;
  Item cur, nxt;
  decode(it, cur, lane_x);
  // weights of (item, chunk): this wave's 16-Cout slice
  // (wave-uniform: the lane's 16 bytes are added at the request as a 32-bit vector offset to a scalar base)
  auto w_of = [&](const Item& I, int cc) { return (unsigned)((cc * wnsl + (I.n0 >> 4) + wn) * W3_WCHUNK) * 4u; };  // byte offset
  const w3_rsrc rsrcw = w3_make_rsrc(p.w_wino3);
  const unsigned lane16 = (unsigned)lane * 16u;
  auto w_ld = [&](unsigned ubase, int ofs_floats) { return w3_bld(rsrcw, lane16, ubase + (unsigned)ofs_floats * 4u); };

  // ---- prologue, exposed once per workgroup: the first stage's halo, its first weights, its first patch
  int stage = 0;  // parity = LDS buffer of the current stage
  int cc = cur.cc_begin;
#pragma unroll
  for (int g = 0; g < W3_WDIST; ++g)
#pragma unroll
    for (int t = 0; t < 4; ++t) Bw[g][t] = w_ld(w_of(cur, cc), g * W3_WSUB + t * 256);
  halo_setup(cur, cc);
#pragma unroll
  for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
    for (int i = 2 * hp; i < 2 * hp + 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) halo_load(i, pl);
#pragma unroll
    for (int i = 2 * hp; i < 2 * hp + 2; ++i) commit_item(i, s_halo);
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a) load_patch_row(s_halo, 0, a);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    xform_row_half(a, 0);
    xform_row_half(a, 1);
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) Y[0][b] = yform1(0, b);
  if (dbg && tid == 0) dbg[1] = HOLO_PROBE_CLOCK();

#ifdef W3_TIMELINE
  // (development) shader-clock sums of wave 0: [0] the 16 MFMAs of a group with the requests between them, [1] clumps that only
  // form A operands, [2] clumps with the x transform, [3] commit clumps, [4] the setup clump, [5] barrier, [6] number of stages
  unsigned long long tl[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
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
        const int nit = it + G;
        if (nit < nq) {
          advance(cur, nxt);
          ncc_ = nxt.cc_begin;
        } else {
          ncc_ = cc;
        }
      }
      const float* buf = s_halo + (stage & 1) * W3_HALO;
      float* obuf = s_halo + ((stage + 1) & 1) * W3_HALO;
      const unsigned wp = w_of(cur, cc), wnext = w_of(nxt, ncc_);
#ifdef W3_TIMELINE
      const unsigned long long t_stage = HOLO_PROBE_CLOCK();
#endif
      // ---- 32 groups of 16 MFMAs; group g = (step st = (xi_z, half), xi_y).  With ONE wave per SIMD nothing hides behind
      //      another wave, and what the wave can issue in the shadow of its own exact-fp32 MFMA is measured
      //      (tools/mfma_shadow_probe.cpp, cycles per MFMA with K fillers behind each): scalar instructions and s_nop are free
      //      up to 3; ds_read_b128 up to 2 (LDS bandwidth beyond); one global_load_dwordx4 per FOUR MFMAs (1 KB per wave
      //      instruction against 64 B/clk of L1); a VECTOR instruction is never hidden - the fp32 MFMA runs on the vector
      //      lanes - and the first one after an MFMA costs ~15 cycles, every further one ~4-5 (v_pk_* the same: two values).
      //      Hence: requests ride BETWEEN the MFMAs of a group, vector work sits in ONE clump per group behind its last MFMA:
      //        MFMA k = 0, 4, 8, 12          one of the four weight requests of group g + 2
      //        xi_y == 3, k = 0..7           the next step's patch, two 16-byte LDS reads each
      //        groups 1-4 / 16-19, k = 2, 10 the halo requests of items 0,1 / 2,3 of the next stage (five 16-byte requests
      //                                      per 16 MFMAs with the weights: the L1 takes one per four MFMAs for free)
      //        behind MFMA 15                A operands of group g + 1 (8 v_pk_add); after xi_y == 3 the x transform of the new
      //                                      patch first (32 v_pk_add); after group 0 the next stage's halo addresses; after
      //                                      groups 8 / 12 / 20 / 24 the commit of halo item 0 / 1 / 2 / 3 (~2 500 cycles after
      //                                      its requests)
      //      Group 28 opens with THE barrier of the stage: the last patch of this stage's buffer was read in group 27 and the
      //      last item of the next stage's buffer was written behind group 24, so from here on the other buffer is complete
      //      (group 31 reads the next stage's first patch from it) and this one is free for the next stage's producer.
      auto group = [&](auto gc) {  // straight-line code: accumulator sets, ring slots, slot work are compile-time choices
        constexpr int g = decltype(gc)::value, st = g >> 2, xy = g & 3;
#ifdef W3_TIMELINE
        const unsigned long long tA0 = clock64();
#endif
        if (g == 28) __syncthreads();
#ifdef W3_TIMELINE
        const unsigned long long tA = clock64();
        if (g == 28) tl[5] += tA - tA0;
#endif
        auto slot = [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          mfma1(acc[(st >> 1) * 16 + xy * 4 + (k & 3)], Y[g & 1][k & 3], Bw[g & 3][k & 3], k >> 2, k == 0);
          if ((k & 3) == 0 && !(W3_PROBE & 2)) {
            const unsigned wsrc = g + W3_WDIST < 32 ? wp : wnext;
            constexpr int wofs = (g + W3_WDIST < 32 ? g + W3_WDIST : g + W3_WDIST - 32) * W3_WSUB + (k >> 2) * 256;
            if (W3_PROBE & 16)
              Bdummy[k >> 2] = w_ld(wsrc, wofs);
            else
              Bw[(g + W3_WDIST) & 3][k >> 2] = w_ld(wsrc, wofs);
          }
          if (xy == 3 && k < 8 && !(W3_PROBE & 4)) {
            // the next step's patch: of this buffer, or (last step) the next stage's first patch from the other buffer
            const float* pb = st < 7 ? buf : obuf;
            constexpr int nstep = st < 7 ? st + 1 : 0;
            const float* base = pb + a_off + ((nstep >> 1) * W3_PLANE + (k >> 1) * W3_HX) * W3_RS + (nstep & 1) * 16;
            P[k >> 1][(2 * k) & 3] = w3_ld(base + ((2 * k) & 3) * W3_RS);
            P[k >> 1][(2 * k + 1) & 3] = w3_ld(base + ((2 * k + 1) & 3) * W3_RS);
          }
          if (((g >= 1 && g <= 4) || (g >= 16 && g <= 19)) && (k == 2 || k == 10) && !(W3_PROBE & 1)) {
            constexpr int hl = (g >= 16 ? g - 16 : g >= 1 ? g - 1 : 0) * 2 + (k == 10 ? 1 : 0);  // 0..7: (item of the pair, plane)
            halo_load((g >= 16 ? 2 : 0) + (hl >> 2), hl & 3);
          }
          __builtin_amdgcn_sched_barrier(0);  // requests stay behind THEIR MFMA
        };
        w3_static_for(slot, std::make_integer_sequence<int, 16>{});
#ifdef W3_TIMELINE
        const unsigned long long tB = clock64();
        tl[0] += tB - tA;
#endif
        // ---- the group's vector clump
        if (!(W3_PROBE & 4)) {
          if (xy == 3) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
              xform_row_half(a, 0);
              xform_row_half(a, 1);
            }
          }
#pragma unroll
          for (int b = 0; b < 4; ++b) Y[(g + 1) & 1][b] = yform1((xy + 1) & 3, b);
        }
        if (!(W3_PROBE & 1)) {
          if (g == 0) halo_setup(nxt, ncc_);
          if (!(W3_PROBE & 32)) {
            if (g == 8) commit_item(0, obuf);
            if (g == 12) commit_item(1, obuf);
            if (g == 20) commit_item(2, obuf);
            if (g == 24) commit_item(3, obuf);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef W3_TIMELINE
        tl[g == 0 ? 4 : (g == 8 || g == 12 || g == 20 || g == 24) ? 3 : xy == 3 ? 2 : 1] += clock64() - tB;
        if (g == 0) tl[6] += 1;
#endif
      };
      w3_static_for(group, std::make_integer_sequence<int, 32>{});
#ifdef W3_TIMELINE
      if (dbg && tid == 0) dbg[6] += HOLO_PROBE_CLOCK() - t_stage;  // the stage's own work (barrier wait included)
#endif
    }
    // ---------------- fused 1x1x1 skip connection: raw block input at the lane's own voxels -> pseudo-taps {0,3}^3.
    //   group sg = (skip chunk, half, dz): the four voxels (dy,dx) of the lane's 2 x 2 x 2 patch = one 16-MFMA group with
    //   accumulators (3dz, 3dy, 3dx).  Its operands come straight from global memory (no halo, no LDS, no barrier), THREE
    //   groups ahead, into the registers of the patch (A) and of the weight ring (B) - what the last stage left there for the
    //   next item is simply requested again afterwards, under the epilogue.  Buffer addressing: no vector instruction per
    //   request (the skip tensors lie within 4 GB: conv_plan).
    if (SKIP) {
      const int nsg = (cur.sk_end - cur.sk_begin) * 4;
      if (nsg > 0) {
        const w3_rsrc rs0 = w3_make_rsrc(p.skip_src0), rs1 = w3_make_rsrc(p.skip_src1 ? p.skip_src1 : p.skip_src0);
        const w3_rsrc rsw = w3_make_rsrc(p.skip_w_wino3);
        const int yt = lj & 3, xt = lj >> 2;
        // voxel (dy, dx) of the lane's patch relative to the tile's first voxel, in voxels
        unsigned vrel[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) vrel[d] = (unsigned)((2 * yt + (d >> 1)) * p.OW + 2 * xt + (d & 1));
        const unsigned vtile = (unsigned)(((cur.n * p.OD + cur.tz0) * p.OH + cur.ty0) * p.OW + cur.tx0);
        const unsigned zvox = (unsigned)(p.OH * p.OW);
        const unsigned wbase = (unsigned)((cur.n0 >> 4) + wn) * (unsigned)W3_WSKIP * 4u;
        auto skip_load = [&](int sg, int slot) {
          const int sc = cur.sk_begin + (sg >> 2), half = (sg >> 1) & 1, dz = sg & 1;
          const int c0 = sc * W3_BK + half * 16;        // first channel of the 16-channel half (wave-uniform)
          const bool pad = c0 >= SCin;  // a half of padding channels: its packed weights are zero, read channel 0 instead
          const bool second = !pad && c0 >= p.skip_C0;  // (skip_C0 is a multiple of 16 when there are two sources: conv_plan)
          const unsigned Cs = (unsigned)(second ? p.skip_C1 : p.skip_C0);
          const unsigned cs = pad ? 0u : (unsigned)(second ? c0 - p.skip_C0 : c0);
          const unsigned soff = ((vtile + (unsigned)dz * zvox) * Cs + cs) * 4u;
          const unsigned wsoff = (unsigned)sc * (unsigned)wnsl * (unsigned)W3_WSKIP * 4u + wbase + (unsigned)((half * 2 + dz) * 4) * 1024u;
          // (channels beyond the skip's last one: the packed weights there are zero, the activations any finite value)
          const unsigned koff = (unsigned)min(kq * 4, max((int)Cs - (int)cs - 4, 0)) * 4u;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const unsigned voff = vrel[d] * Cs * 4u + koff;
            P[slot][d] = second ? w3_bld(rs1, voff, soff) : w3_bld(rs0, voff, soff);
            Bw[slot][d] = w3_bld(rsw, lane16, wsoff + (unsigned)d * 1024u);
          }
        };
#pragma unroll
        for (int g0 = 0; g0 < 3; ++g0)
          if (g0 < nsg) skip_load(g0, g0);
        for (int sg = 0; sg < nsg; sg += 4) {  // four groups per turn: ring slots and accumulator choices stay static
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (sg + j + 3 < nsg) skip_load(sg + j + 3, (j + 3) & 3);
            __builtin_amdgcn_sched_barrier(0);
            if (j & 1)
              mfma16(acc[48], acc[51], acc[60], acc[63], P[j], Bw[j]);
            else
              mfma16(acc[0], acc[3], acc[12], acc[15], P[j], Bw[j]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // what the skip overwrote: the next stage's first weights and first patch (its A operands are formed after the
        // epilogue, under which these requests complete)
        const float* nbuf = s_halo + (stage & 1) * W3_HALO;
        const unsigned wn0 = w_of(nxt, nxt.cc_begin);
#pragma unroll
        for (int g0 = 0; g0 < W3_WDIST; ++g0)
#pragma unroll
          for (int t = 0; t < 4; ++t) Bw[g0][t] = w_ld(wn0, g0 * W3_WSUB + t * 256);
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) load_patch_row(nbuf, 0, a4);
      }
    }
    if (dbg && tid == 0) dbg[2] = HOLO_PROBE_CLOCK();
    HOLO_MFMA_DRAIN();  // the last MFMAs' results, before vector instructions read the accumulators
#if W3_PROBE & 16
    HOLO_SINK8(Bdummy[0].lo, Bdummy[0].hi, Bdummy[1].lo, Bdummy[1].hi, Bdummy[2].lo, Bdummy[2].hi, Bdummy[3].lo, Bdummy[3].hi);
#endif
#if W3_PROBE & 32
    HOLO_SINK8(hrA[0].lo, hrA[1].lo, hrA[2].lo, hrA[3].lo, hrB[0].lo, hrB[1].lo, hrB[2].lo, hrB[3].lo);
#endif

    // ---------------- output transform (lane-local: x, y, z) + epilogue.  D row 4 kq + r = (y tile r, x tile kq).
    //                  Two register indices r at a time (v_pk_add_f32); the residual is requested before anything else.
    {
      const int co = cur.n0 + wn * 16 + lj;
      const int coc = co;  // (Cout is a multiple of 64: conv_wino3_launch)
      const bool direct = p.nsplit == 1;
      const bool has_res = direct && p.residual != nullptr;
      float bv = (direct && p.bias) ? p.bias[coc] : 0.f;
      if (direct && p.skip_bias) bv += p.skip_bias[coc];
      const int64_t tbase = ((((int64_t)cur.n * p.OD + cur.tz0) * p.OH + cur.ty0) * p.OW + cur.tx0 + 2 * kq) * p.Cout + coc;
      const int64_t zstride = (int64_t)p.OH * p.OW * p.Cout;
      const int ystride = p.OW * p.Cout;
      float* obase = direct ? p.out : p.partial + (int64_t)cur.split * M * p.Cout;
      // voxel (dz, 2 r + dy, dx) of the lane's x tile
      auto vofs = [&](int r, int dz, int dy, int dx) { return tbase + dz * zstride + (int64_t)(2 * r + dy) * ystride + dx * p.Cout; };
      // one register index r (= y tile) at a time: 64 accumulator reads, x and y transforms on (xi_z, xi_z + 1) register
      // pairs (v_pk_add_f32), the z transform on their halves; the residual of r + 1 is requested before r is transformed
      float res[2][2][2][2];
      auto load_res = [&](int r) {
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
          for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) res[r & 1][dz][dy][dx] = has_res ? p.residual[vofs(r, dz, dy, dx)] : 0.f;
      };
      load_res(0);
      float ssum = 0.f, ssq = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r < 3) load_res(r + 1);
        f32x2 oz[2][2][2];  // [xi_z pair][dy][dx]
#pragma unroll
        for (int zp = 0; zp < 2; ++zp) {
          f32x2 oy[4][2];  // [xi_y][dx]
#pragma unroll
          for (int xy = 0; xy < 4; ++xy) {
            const int s0 = (2 * zp) * 16 + xy * 4, s1 = s0 + 16;
            const f32x2 m0 = f32x2{acc[s0 + 0][r], acc[s1 + 0][r]}, m1 = f32x2{acc[s0 + 1][r], acc[s1 + 1][r]},
                        m2 = f32x2{acc[s0 + 2][r], acc[s1 + 2][r]}, m3 = f32x2{acc[s0 + 3][r], acc[s1 + 3][r]};
            oy[xy][0] = pk_add(pk_add(m0, m1), m2);
            oy[xy][1] = pk_sub(pk_sub(m1, m2), m3);
          }
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            oz[zp][0][dx] = pk_add(pk_add(oy[0][dx], oy[1][dx]), oy[2][dx]);
            oz[zp][1][dx] = pk_sub(pk_sub(oy[1][dx], oy[2][dx]), oy[3][dx]);
          }
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const f32x2 za = oz[0][dy][dx], zb = oz[1][dy][dx];  // (xi_z 0, 1), (xi_z 2, 3)
            float o0 = (za.x + za.y) + zb.x;
            float o1 = (za.y - zb.x) - zb.y;
            if (direct) {
              o0 = (o0 + res[r & 1][0][dy][dx]) + bv;
              o1 = (o1 + res[r & 1][1][dy][dx]) + bv;
              ssum += o0 + o1;
              ssq = fmaf(o0, o0, fmaf(o1, o1, ssq));
            }
            obase[vofs(r, 0, dy, dx)] = o0;
            obase[vofs(r, 1, dy, dx)] = o1;
          }
      }
      // GroupNorm statistics of the tensor just produced: one slab per tile (conv_stats_slabs)
      if (p.stats && direct) {
        float s1 = ssum, s2 = ssq;
        s1 += __shfl_xor(s1, 16);
        s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (kq == 0) {
          const int tiles_per_sample = ntx * nty * ntz;
          const int slab = ((cur.tz0 >> 1) * nty + (cur.ty0 >> 3)) * ntx + (cur.tx0 >> 3);
          double* d = p.stats + (((int64_t)cur.n * tiles_per_sample + slab) * p.Cout + co) * 2;
          d[0] = (double)s1;
          d[1] = (double)s2;
        }
      }
    }
    if (dbg && tid == 0) {
      const unsigned long long t_now = HOLO_PROBE_CLOCK();
      dbg[5] += t_now - dbg[2];  // skip-drain + output transform + stores
      dbg[3] = t_now;
      dbg[7] += 1;
    }
    it += G;
    if (it >= nq) break;
    if (SKIP && cur.sk_end > cur.sk_begin) {  // the patch requested again behind the skip section: its x and y transforms
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) {
        xform_row_half(a4, 0);
        xform_row_half(a4, 1);
      }
#pragma unroll
      for (int b4 = 0; b4 < 4; ++b4) Y[0][b4] = yform1(0, b4);
    }
    cur = nxt;
  }
#ifdef W3_TIMELINE
  if (p.dbg && tid == 0) {
    unsigned long long* d = p.dbg + ((int64_t)gridDim.x + blockIdx.x) * 8;  // second half of the probe buffer
#pragma unroll
    for (int i = 0; i < 7; ++i) d[i] = tl[i];
  }
#endif
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

#if W3_PROBE == 0 && !defined(W3_TIMELINE)
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

#endif  // the library copy

// p.wino == 3 (conv_plan): p.grid_x persistent workgroups
int W3_ENTRY(const ConvParams& p, void* stream) {
  if (!p.w_wino3 || (p.skip_w && !p.skip_w_wino3) || (p.OD & 1) || (p.OH & 7) || (p.OW & 7) || (p.Cout & 63) ||
      (p.coef && !p.act)) {
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
