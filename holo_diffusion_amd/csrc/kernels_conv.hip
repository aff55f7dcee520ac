// kernels_conv.hip — 3D convolution (3x3x3 / 1x1x1, stride 1|2, pad 1|0) as an implicit GEMM on the
// gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 157 TFLOP/s peak).
//
// Replaces the torch/MIOpen conv3d calls of the reference denoiser:
//   holo_diffusion/guided_diffusion/unet.py:185,211 (ResBlock convs), :89 (Upsample conv),
//   :129-131 (Downsample conv, stride (2,2,2)), :222 (1x1x1 skip), :383,392 (attention qkv / proj conv1d),
//   :657 (input conv), :792 (output conv)
// and fuses into the operand staging what the reference runs as separate ATen ops:
//   GroupNorm32 apply + FiLM scale/shift + SiLU (unet.py:183-184,207-208,248-252; nn.py:23-25),
//   nearest x2 upsampling (unet.py:93-97), the skip-connection channel concat (unet.py:829),
//   bias and the residual add (unet.py:256).
//
// Layout: activations are channels-last [n][d][h][w][c]; weights are pre-packed [tap][Cout][Cin].
// GEMM view: M = N*OD*OH*OW output voxels, N = Cout, K = taps*Cin walked in chunks of 32 channels of
// one tap.  Block = 256 threads = 4 waves; block tile 128 voxels x (32*NT) Cout; each wave owns 32
// voxels x 32*NT Cout = NT accumulators of 32x32 (16 VGPRs each).
//
// LDS: per chunk an A tile [128][36] and a B tile [32*NT][36] (row = voxel / Cout, 32 channels + 4 pad
// floats so that rows stay 16-byte aligned and the ds_read_b128 fragment reads are bank-conflict free:
// row*36 mod 64 walks all 16-byte slots).  The K index inside a chunk is permuted so that lane half h
// (= lane>>5, the MFMA k index) owns channels [16h, 16h+16): each lane then fetches its 16 A operands
// (and 16 per B tile) for the 16 MFMA k-steps of the chunk with four ds_read_b128.
// Double buffered: global loads of chunk k+1 are issued before the MFMAs of chunk k and written to the
// other LDS buffer afterwards; one barrier per chunk.
#include <stdio.h>
#include <stdlib.h>

#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDK = 36;

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
// bf16 paths: v_rcp_f32 (1 ulp) instead of the IEEE division sequence (the result is rounded to bf16 anyway)
__device__ __forceinline__ float silu_fast(float v) { return v * holo_rcp(1.0f + __expf(-v)); }

// activation element access for the bf16 storage mode: `bf` = the buffer behind the float* holds bf16
__device__ __forceinline__ float bf16_load(const uint16_t* p) { return __uint_as_float((uint32_t)(*p) << 16); }
__device__ __forceinline__ uint16_t bf16_round(float v) { return (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu); }
__device__ __forceinline__ float4 ld_act4(const float* base, int64_t idx, int bf) {
  if (bf) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + idx);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  }
  return *reinterpret_cast<const float4*>(base + idx);
}
__device__ __forceinline__ void st_act4(float* base, int64_t idx, const float4& v, int bf) {
  if (bf)
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + idx) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  else
    *reinterpret_cast<float4*>(base + idx) = v;
}
__device__ __forceinline__ float ld_act1(const float* base, int64_t idx, int bf) {
  return bf ? bf16_load(reinterpret_cast<const uint16_t*>(base) + idx) : base[idx];
}
__device__ __forceinline__ void st_act1(float* base, int64_t idx, float v, int bf) {
  if (bf)
    reinterpret_cast<uint16_t*>(base)[idx] = bf16_round(v);
  else
    base[idx] = v;
}

// Packed conv weights (repack_conv_weight_kernel): [tap][chunk = cin/32][slice = cout/16][half][kq][lj][4], i.e. one
// 2 KB block per (tap, 32-channel chunk, 16-Cout slice) holding exactly the B fragments of a v_mfma_f32_16x16x4_f32
// wave (lane = 16*kq + lj owns k = 8*kq + 4*half + e of output channel 16*slice + lj): a wave's B load is two fully
// coalesced 1 KB reads and a workgroup's weights for a chunk are one contiguous 8 KB run.
__device__ inline int64_t wpack_block(int tap, int cc, int slice, int ncc, int nslice) {
  return (((int64_t)tap * ncc + cc) * nslice + slice) * 512;
}

template <int NT>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvParams p) {
  constexpr int BN = 32 * NT;
  constexpr int BUF = (BM + BN) * LDK;
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int Cin = p.C0 + p.C1;
  const int ncc = (Cin + BK - 1) / BK;
  const int ntaps = p.ksz * p.ksz * p.ksz;
  const int nchunks = ntaps * ncc;
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int kc_begin = blockIdx.z * p.chunks_per_split;
  int kc_end = kc_begin + p.chunks_per_split;
  if (kc_end > nchunks) kc_end = nchunks;

  // ---- per-thread staging assignment: 8 threads cover the 32 channels (float4 each) of one row
  const int wncc = p.CinP / BK, wnsl = p.CoutP >> 4;  // packed-weight geometry (wpack_block)
  const int q = tid & 7;
  const int r0 = tid >> 3;  // 0..31
  int an[4], az[4], ay[4], ax[4];
  bool av[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int64_t m = m0 + r0 + 32 * j;
    av[j] = m < M;
    if (!av[j]) m = 0;
    int ow = (int)(m % p.OW);
    int64_t t = m / p.OW;
    int oh = (int)(t % p.OH);
    t /= p.OH;
    int od = (int)(t % p.OD);
    an[j] = (int)(t / p.OD);
    az[j] = od * p.stride - p.pad;
    ay[j] = oh * p.stride - p.pad;
    ax[j] = ow * p.stride - p.pad;
  }
  const int SD = p.ups ? (p.ID >> 1) : p.ID;
  const int SH = p.ups ? (p.IH >> 1) : p.IH;
  const int SW = p.ups ? (p.IW >> 1) : p.IW;

  float4 ra[4];
  float4 rb[NT];
  float4 rc01[4], rc23[4];
  unsigned amask = 0;

  // loads are unconditional from clamped addresses (see the halo kernel for why); validity is a mask
  auto load_chunk = [&](int kc) {
    const int tap = kc / ncc;
    const int cc = kc - tap * ncc;
    int kd = 0, kh = 0, kw = 0;
    if (p.ksz == 3) {
      kd = tap / 9;
      kh = (tap - kd * 9) / 3;
      kw = tap - kd * 9 - kh * 3;
    }
    int c = cc * BK + q * 4;  // channel in the concatenated input
    const bool cvalid = c < Cin;
    if (!cvalid) c = 0;
    const float* src = p.src0;
    int Cs = p.C0, cs = c;
    if (c >= p.C0) {
      src = p.src1;
      Cs = p.C1;
      cs = c - p.C0;
    }
    amask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int z = az[j] + kd, y = ay[j] + kh, x = ax[j] + kw;
      const bool ok = av[j] && cvalid && z >= 0 && z < p.ID && y >= 0 && y < p.IH && x >= 0 && x < p.IW;
      z = min(max(z, 0), p.ID - 1);
      y = min(max(y, 0), p.IH - 1);
      x = min(max(x, 0), p.IW - 1);
      if (p.ups) {
        z >>= 1;
        y >>= 1;
        x >>= 1;
      }
      const int64_t idx = ((((int64_t)an[j] * SD + z) * SH + y) * SW + x) * Cs + cs;
      ra[j] = ld_act4(src, idx, p.in_bf16);
      amask |= (ok ? 1u : 0u) << j;
      if (p.coef) {
        const float4* cf = reinterpret_cast<const float4*>(p.coef + ((int64_t)an[j] * Cin + c) * 2);
        rc01[j] = cf[0];
        rc23[j] = cf[1];
      }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int co = n0 + r0 + 32 * j;  // < CoutP by construction of the grid
      rb[j] = *reinterpret_cast<const float4*>(p.w + wpack_block(tap, cc, co >> 4, wncc, wnsl) + (q & 1) * 256 +
                                               ((q >> 1) * 16 + (co & 15)) * 4);
    }
  };

  auto store_chunk = [&](int buf) {
    float* base = lds + buf * BUF;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 v = ra[j];
      if (p.coef) {
        v.x = v.x * rc01[j].x + rc01[j].y;
        v.y = v.y * rc01[j].z + rc01[j].w;
        v.z = v.z * rc23[j].x + rc23[j].y;
        v.w = v.w * rc23[j].z + rc23[j].w;
        if (p.act) {
          v.x = silu_f(v.x);
          v.y = silu_f(v.y);
          v.z = silu_f(v.z);
          v.w = silu_f(v.w);
        }
      }
      const float keep = ((amask >> j) & 1u) ? 1.f : 0.f;
      v.x *= keep;
      v.y *= keep;
      v.z *= keep;
      v.w *= keep;
      *reinterpret_cast<float4*>(base + (r0 + 32 * j) * LDK + q * 4) = v;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) *reinterpret_cast<float4*>(base + (BM + r0 + 32 * j) * LDK + q * 4) = rb[j];
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int li = lane & 31;
  const int lh = lane >> 5;

  auto compute = [&](int buf) {
    const float* base = lds + buf * BUF;
    float a[16];
    float b[NT][16];
    const float4* ap = reinterpret_cast<const float4*>(base + (wave * 32 + li) * LDK + lh * 16);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float4 t4 = ap[v];
      a[4 * v + 0] = t4.x;
      a[4 * v + 1] = t4.y;
      a[4 * v + 2] = t4.z;
      a[4 * v + 3] = t4.w;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float4* bp = reinterpret_cast<const float4*>(base + (BM + t * 32 + li) * LDK + lh * 16);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        float4 t4 = bp[v];
        b[t][4 * v + 0] = t4.x;
        b[t][4 * v + 1] = t4.y;
        b[t][4 * v + 2] = t4.z;
        b[t][4 * v + 3] = t4.w;
      }
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], b[t][ks], acc[t], 0, 0, 0);
  };

  if (kc_begin < kc_end) {
    load_chunk(kc_begin);
    store_chunk(0);
    __syncthreads();
    for (int kc = kc_begin; kc < kc_end; ++kc) {
      const int buf = (kc - kc_begin) & 1;
      const bool more = kc + 1 < kc_end;
      if (more) load_chunk(kc + 1);
      compute(buf);
      if (more) store_chunk(buf ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue: D[row][col]: col = lane&31 (Cout), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (voxel)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int co = n0 + t * 32 + li;
    const int coc = co < p.Cout ? co : p.Cout - 1;
    const float bv = (p.nsplit == 1 && p.bias) ? p.bias[coc] : 0.f;
    int64_t mo[16];
    bool mv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      mv[r] = m < M && co < p.Cout;
      mo[r] = (m < M ? m : M - 1) * p.Cout;
    }
    if (p.nsplit == 1) {
      if (p.residual) {  // one batch of loads (clamped addresses), no per-element branch
        float res[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) res[r] = ld_act1(p.residual, mo[r] + coc, p.res_bf16);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] += res[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (mv[r]) st_act1(p.out, mo[r] + co, acc[t][r] + bv, p.out_bf16);
    } else {
      float* pp = p.partial + (int64_t)blockIdx.z * M * p.Cout + coc;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (mv[r]) pp[mo[r]] = acc[t][r];
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Halo kernel for stride-1 3x3x3 convolutions (95% of the FLOPs).
//
// Block = 4 waves, output tile = 2 x 8 x 8 voxels (z,y,x) = 128 GEMM rows x (16*NWN) output channels.
// For one 32-channel chunk the 4 x 10 x 10 input halo of the tile is staged ONCE in LDS (GroupNorm/FiLM/
// SiLU applied once per element, zero padding after the activation) and all 27 taps read their A
// fragments from it at shifted addresses: global loads, activation math and LDS writes of the A operand
// drop 8.6x (27*128/400) relative to the per-tap gather kernel above.
//
// Work split inside the block: each wave OWNS a 16-wide Cout slice and sweeps the voxels of the tile with
// v_mfma_f32_16x16x4_f32 (A = 16 voxels x 4 channels, B = 4 channels x 16 Cout).  Consequences:
//   * the weights of a tap are needed by exactly one wave, so they go global -> registers directly
//     (two 16-byte loads per lane per tap, prefetched one tap ahead), never through LDS: the 27-tap
//     loop has NO barrier and no wave re-fetches another wave's weights;
//   * the A operand is the block-shared LDS halo; a tap's A fragments for the wave's MT voxel tiles are
//     2*MT ds_read_b128 per lane, software-pipelined in two halves behind the MFMAs.
// K index inside a chunk: lane quarter kq = lane>>4 (the MFMA k index) owns channels [8kq, 8kq+8), k-step
// ks of the 8 per tap uses channel 8kq+ks, so a lane's operands for a whole tap are 8 contiguous floats.
// The halo of the NEXT channel chunk is requested under the last tap and committed after it (two
// barriers per chunk).  Split-K runs over channel chunks.
// LDS: (TZ+2) x 10 x 10 x 36 floats = 57.6 KB (TZ=2) / 43.2 KB (TZ=1).
// ---------------------------------------------------------------------------------------------
constexpr int HY = 10, HX = 10;

// NWN: waves along Cout (4 -> 64 channels per block, 2 -> 32).  TZ: tile depth in voxels (2 -> 128-voxel tiles
// for the 64^3 level; 1 -> 64-voxel tiles so that the 32^3..8^3 levels launch twice the workgroups and the 32^3
// level needs no split-K)
// BF: multiply in bf16 on the matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulate): the halo is rounded to bf16
// (RNE) when it is committed to LDS (80-byte rows: 32 channels + 16 bytes of padding, conflict free for the
// 16-byte A reads), the weights come pre-rounded and packed per lane, and one MFMA covers the 32 channels of a
// chunk for a tap (eight v_mfma_f32_16x16x4_f32 in the fp32 form).  Opt-in (ConvParams::bf16).
template <int NWN, int TZ, bool SKIP, bool BF = false>  // SKIP: a 1x1x1 skip connection is fused as extra K chunks
__global__ __launch_bounds__(256, 2) void conv_halo_kernel(ConvParams p) {
  constexpr int RS = BF ? 20 : LDK;  // LDS row stride of one halo voxel, in 4-byte words
  constexpr int BN = 16 * NWN;
  constexpr int MT = TZ * NWN;  // 16-voxel tiles per wave
  constexpr int HZ = TZ + 2;
  constexpr int HALO_VOX = HZ * HY * HX;
  constexpr int HALO_IT = (HALO_VOX + 31) / 32;  // halo rows per thread (8 threads cover one row's 32 channels)
  __shared__ __attribute__((aligned(16))) float s_halo[HALO_VOX * RS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lj = lane & 15;
  const int kq = lane >> 4;
  const int wn = wave % NWN;
  const int wm = NWN == 4 ? 0 : wave / NWN;  // (4 waves: compile-time 0, so the A addresses fold into immediates)
  const int Cin = p.C0 + p.C1;
  const int ncc = (Cin + BK - 1) / BK;
  // Spatial tiles: one tile per workgroup (gridDim.x = tiles).  A persistent form - each workgroup walking tiles
  // blockIdx.x, +gridDim.x, ... with the next tile's first halo requested under the last tap - was measured with
  // tools/conv_timeline.cpp: it raises the slot occupancy from 1.6 to 1.8 of 2 but loses the natural phase stagger
  // between the two resident workgroups and comes out even, and its cross-tile prefetch costs ~25 VGPRs (spills in
  // the 128-voxel variant), so the loop below runs exactly once.
  const int ntx = p.OW >> 3, nty = p.OH >> 3, ntz = p.OD / TZ;
  const int ntiles = ntx * nty * ntz * p.N;
  int tx0 = 0, ty0 = 0, tz0 = 0, n = 0;      // tile whose halo is being STAGED (halo_issue)
  int ctx0 = 0, cty0 = 0, ctz0 = 0, cn = 0;  // tile being COMPUTED / written (epilogue)
  auto decode_tile = [&](int bt) {
    tx0 = (bt % ntx) << 3;
    bt /= ntx;
    ty0 = (bt % nty) << 3;
    bt /= nty;
    tz0 = (bt % ntz) * TZ;
    n = bt / ntz;
  };
  const int n0 = blockIdx.y * BN;
  // K chunks: [0, ncc) main 3x3x3 chunks, [ncc, ncc + nsk) the fused 1x1x1 skip connection of a ResBlock
  // (unet.py:222,256: skip_connection(x) + h): same output tile, source = the block input, centre tap only
  const int SCin = p.skip_C0 + p.skip_C1;
  const int nsk = SKIP ? (SCin + BK - 1) / BK : 0;
  // split-K: every split takes an equal share of the main chunks AND an equal share of the (27x cheaper) skip chunks
  const int cc_begin = blockIdx.z * p.chunks_per_split;
  int cc_end = cc_begin + p.chunks_per_split;
  if (cc_end > ncc) cc_end = ncc;
  const int sk_begin = ncc + blockIdx.z * p.skip_chunks_per_split;
  int sk_end = sk_begin + p.skip_chunks_per_split;
  if (sk_end > ncc + nsk) sk_end = ncc + nsk;
  const int SD = p.ups ? (p.ID >> 1) : p.ID;
  const int SH = p.ups ? (p.IH >> 1) : p.IH;
  const int SW = p.ups ? (p.IW >> 1) : p.IW;

  const int q = tid & 7;
  const int r0 = tid >> 3;

  // ---- halo staging, split in "issue the loads" / "transform + write LDS"
  float4 hreg[HALO_IT];
  unsigned hmask = 0;   // bit i: element i is inside the volume and a real channel (zero padding otherwise)
  unsigned hvalid = 0;  // spatial part of hmask, per tile
  // clamped source voxel of every halo row this thread stages, per tile: parked in LDS ([i][thread], conflict
  // free) - in registers the 13 values push the 128-voxel variant past 256 VGPRs under the last tap
  __shared__ int s_hvox[HALO_IT * 256];
  int hcoef_c = 0;
  bool h_is_skip = false;
  // On gfx950 the fp32 MFMA runs on the vector FMA lanes (tools/coexec_probe.cpp), so every VALU instruction of the
  // staging path costs matrix time: the halo coordinates are worked out ONCE per tile (halo_prepare), a chunk's
  // loads are then one multiply-add each, and the activation runs on register pairs (v_pk_fma / v_pk_mul).
  auto halo_prepare = [&]() {
    hvalid = 0;
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int hv = min(r0 + 32 * i, HALO_VOX - 1);
      const int hz = hv / (HY * HX);
      const int rem = hv - hz * (HY * HX);
      const int hy = rem / HX;
      const int hx = rem - hy * HX;
      int z = tz0 + hz - 1, y = ty0 + hy - 1, x = tx0 + hx - 1;
      const bool ok = z >= 0 && z < p.ID && y >= 0 && y < p.IH && x >= 0 && x < p.IW;
      z = min(max(z, 0), p.ID - 1);
      y = min(max(y, 0), p.IH - 1);
      x = min(max(x, 0), p.IW - 1);
      if (p.ups) {  // (never set together with a fused skip: ResBlocks do not resample)
        z >>= 1;
        y >>= 1;
        x >>= 1;
      }
      s_hvox[i * 256 + tid] = (z * SH + y) * SW + x;
      hvalid |= (ok ? 1u : 0u) << i;
    }
  };
  auto halo_issue = [&](int cc) {
    h_is_skip = SKIP && cc >= ncc;
    int c = (h_is_skip ? cc - ncc : cc) * BK + q * 4;
    hcoef_c = c;
    const bool cvalid = c < (h_is_skip ? SCin : Cin);
    if (!cvalid) c = 0;  // clamped, masked below
    const float* src = h_is_skip ? p.skip_src0 : p.src0;
    const int C0s = h_is_skip ? p.skip_C0 : p.C0;
    int Cs = C0s, cs = c;
    if (c >= C0s) {
      src = h_is_skip ? p.skip_src1 : p.src1;
      Cs = h_is_skip ? p.skip_C1 : p.C1;
      cs = c - C0s;
    }
    hmask = cvalid ? hvalid : 0u;
    // Every load is issued unconditionally from a clamped (always valid) address and masked afterwards:
    // a "load or zero" branch would make the compiler wait for each load before the next one is issued.
    // uniform 64-bit base + one 32-bit byte offset per load (conv_plan keeps a source volume below 4 GB on this path)
    constexpr unsigned ES = BF ? 2u : 4u;  // bf16 mode = bf16 storage: 8 bytes per (voxel, 4 channels)
    const char* sbase = reinterpret_cast<const char*>(src) + (int64_t)n * SD * SH * SW * Cs * ES;
    const unsigned cbytes = (unsigned)Cs * ES, cofs = (unsigned)cs * ES;
    int tl = tid;
    HOLO_LAUNDER(tl);  // (otherwise the compiler forwards the values stored by halo_prepare and keeps them in VGPRs)
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const char* a = sbase + ((unsigned)s_hvox[i * 256 + tl] * cbytes + cofs);
      if (BF) {  // raw bits; unpacked in halo_commit (no wait on the load here)
        const uint2 u = *reinterpret_cast<const uint2*>(a);
        hreg[i].x = __uint_as_float(u.x);
        hreg[i].y = __uint_as_float(u.y);
      } else {
        hreg[i] = *reinterpret_cast<const float4*>(a);
      }
    }
  };
  auto halo_commit = [&]() {
    f32x2 a01 = f32x2{1.f, 1.f}, b01 = f32x2{0.f, 0.f}, a23 = a01, b23 = b01;
    const bool xform = p.coef && !h_is_skip;  // the skip path reads the raw block input
    if (xform) {
      const int cc4 = hcoef_c < Cin ? hcoef_c : 0;
      const float4* cf = reinterpret_cast<const float4*>(p.coef + ((int64_t)n * Cin + cc4) * 2);
      const float4 c01 = cf[0], c23 = cf[1];  // (a,b) interleaved per channel
      a01 = f32x2{c01.x, c01.z};
      b01 = f32x2{c01.y, c01.w};
      a23 = f32x2{c23.x, c23.z};
      b23 = f32x2{c23.y, c23.w};
    }
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int hv = r0 + 32 * i;
      f32x2 v01 = f32x2{hreg[i].x, hreg[i].y}, v23 = f32x2{hreg[i].z, hreg[i].w};
      if (BF) {
        const uint32_t w0 = __float_as_uint(hreg[i].x), w1 = __float_as_uint(hreg[i].y);
        v01 = f32x2{__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u)};
        v23 = f32x2{__uint_as_float(w1 << 16), __uint_as_float(w1 & 0xffff0000u)};
      }
      if (xform) {
        v01 = pk_fma(v01, a01, b01);
        v23 = pk_fma(v23, a23, b23);
        if (p.act) {
          v01 = f32x2{silu_f(v01.x), silu_f(v01.y)};
          v23 = f32x2{silu_f(v23.x), silu_f(v23.y)};
        }
      }
      const float keep = ((hmask >> i) & 1u) ? 1.f : 0.f;  // zero padding is applied AFTER the activation
      const f32x2 k2 = f32x2{keep, keep};
      v01 = pk_mul(v01, k2);
      v23 = pk_mul(v23, k2);
      if (BF) {
        if (hv < HALO_VOX)
          *reinterpret_cast<uint2*>(s_halo + hv * RS + q * 2) = make_uint2(pack_bf16x2(v01.x, v01.y), pack_bf16x2(v23.x, v23.y));
      } else {
        if (hv < HALO_VOX) *reinterpret_cast<float4*>(s_halo + hv * RS + q * 4) = make_float4(v01.x, v01.y, v23.x, v23.y);
      }
    }
  };

  f32x4 acc[MT];

  // A addressing: lane reads 8 channels starting at 8*kq of its voxel
  // The 16 voxels of MFMA tile T (0 .. 4*TZ-1) are x = 0..7 of rows y = (T&3) and (T&3)+4 of slab z = T>>2: the
  // two rows are 4*HX*LDK = 1440 = 32 (mod 64) floats apart, which makes the 16 lanes of every ds_read_b128
  // group hit 64 distinct banks (adjacent rows, 40 mod 64 apart, gave 2-way conflicts on a quarter of the lanes).
  int a_off[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int T = wm * MT + t;
    a_off[t] = (((T >> 2) * HY + (T & 3) + 4 * (lj >> 3)) * HX + (lj & 7)) * RS + kq * (BF ? 4 : 8);
  }
  // B addressing: weight row of this lane's output channel, 8 channels starting at 8*kq of the chunk
  // (packed layout: the two 16-byte B fragments of a lane for one (tap, chunk) sit at lane*16 B in two 1 KB planes;
  //  bf16: one 16-byte fragment per lane, 1 KB per block)
  const int wncc = p.CinP / BK, wnsl = p.CoutP >> 4;
  constexpr int WBLK = BF ? 256 : 512;  // words per (tap, chunk, slice) block
  const float* w_lane = (BF ? reinterpret_cast<const float*>(p.w_bf) : p.w) + (int64_t)((n0 >> 4) + wn) * WBLK + lane * 4;
  const float* skw = BF ? reinterpret_cast<const float*>(p.skip_w_bf) : p.skip_w;

  auto load_a = [&](float4 (&a)[MT], int tap, int half) {
    const int kd = tap / 9, kh = (tap - kd * 9) / 3, kw = tap - kd * 9 - kh * 3;
    const int toff = ((kd * HY + kh) * HX + kw) * RS + half * 4;
#pragma unroll
    for (int t = 0; t < MT; ++t) a[t] = *reinterpret_cast<const float4*>(s_halo + a_off[t] + toff);
  };
  auto load_b = [&](float4 (&b)[2], int cc, int tap) {
    const float* wp = w_lane + (int64_t)(tap * wncc + cc) * wnsl * WBLK;
    b[0] = *reinterpret_cast<const float4*>(wp);
    if (!BF) b[1] = *reinterpret_cast<const float4*>(wp + 256);
  };
  auto mfma_half = [&](const float4 (&a)[MT], const float4& b) {
    if (BF) {  // one instruction covers the chunk's 32 channels
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = mfma_bf16_16x16x32(a[t], b, acc[t]);
      return;
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, b.x, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, b.y, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, b.z, acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, b.w, acc[t], 0, 0, 0);
  };

  float4 aA[MT] = {}, aB[MT] = {}, aC[MT] = {};  // rotating A buffers: first half of this tap, second half, first half of next tap
  float4 b0[2] = {}, b1[2] = {};

  // one tap: on entry `cur` holds the first-half A operands of `tap` and `bc` its weights
  auto tap_body = [&](float4 (&cur)[MT], float4 (&nxt)[MT], float4 (&bc)[2], float4 (&bn)[2], int cc, int tap,
                      bool prefetch) {
    if (BF) {  // whole tap = MT instructions; the next tap's operands are requested ahead of them
      if (prefetch) {
        load_a(nxt, tap + 1, 0);
        load_b(bn, cc, tap + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(cur, bc[0]);
      __builtin_amdgcn_sched_barrier(0);
      return;
    }
    load_a(aB, tap, 1);
    if (prefetch) load_b(bn, cc, tap + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the requests above AHEAD of the MFMAs that hide their latency
    mfma_half(cur, bc[0]);
    __builtin_amdgcn_sched_barrier(0);
    if (prefetch) load_a(nxt, tap + 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(aB, bc[1]);
    __builtin_amdgcn_sched_barrier(0);
  };

  HOLO_PHASE_DELAY(p.stagger_ticks);
  // Without a fused skip the next halo is prefetched under the last tap of the current chunk.  The SKIP
  // instantiation requests and commits back to back instead: measured on MI355X the prefetch makes it 7% SLOWER
  // (225 instead of 171 VGPRs, and its chunks are short).
  decode_tile(blockIdx.x);
  halo_prepare();
  if (!SKIP) halo_issue(cc_begin);
  for (int tile = blockIdx.x; tile < ntiles; tile += ntiles) {  // (one tile per workgroup, see below)
  ctx0 = tx0, cty0 = ty0, ctz0 = tz0, cn = n;
  unsigned long long* dbg = p.dbg ? p.dbg + ((int64_t)tile * gridDim.y + blockIdx.y) * 8 : nullptr;
  if (dbg && tid == 0) dbg[0] = HOLO_PROBE_CLOCK();
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
  const bool more_tiles = false;
  for (int cc = cc_begin; cc < cc_end; ++cc) {
    if (SKIP) halo_issue(cc);
    halo_commit();
    load_b(b0, cc, 0);
    __syncthreads();  // halo of chunk cc visible
    if (dbg && tid == 0 && cc == cc_begin) dbg[1] = HOLO_PROBE_CLOCK();
    load_a(aA, 0, 0);
    for (int tap = 0; tap < 26; tap += 2) {
      tap_body(aA, aC, b0, b1, cc, tap, true);
      tap_body(aC, aA, b1, b0, cc, tap + 1, true);
    }
    // the next halo flies under the last tap: next chunk of this tile or first chunk of the workgroup's next tile
    if (!SKIP) {
      if (cc + 1 < cc_end) {
        halo_issue(cc + 1);
      } else if (more_tiles) {
        decode_tile(tile + gridDim.x);
        halo_prepare();
        halo_issue(cc_begin);
      }
    }
    tap_body(aA, aC, b0, b1, cc, 26, false);
    __syncthreads();  // everyone done reading this halo before it is overwritten
  }
  if (SKIP) {
    // fused 1x1x1 skip connection: centre tap (13) of the block-input halo, its own packed weights
    for (int cc = sk_begin; cc < sk_end; ++cc) {
      halo_issue(cc);
      halo_commit();
      const float* wp = skw + ((int64_t)(cc - ncc) * wnsl + (n0 >> 4) + wn) * WBLK + lane * 4;
      b0[0] = *reinterpret_cast<const float4*>(wp);
      if (!BF) b0[1] = *reinterpret_cast<const float4*>(wp + 256);
      __syncthreads();
      load_a(aA, 13, 0);
      if (!BF) load_a(aB, 13, 1);
      mfma_half(aA, b0[0]);
      if (!BF) mfma_half(aB, b0[1]);
      __syncthreads();
    }
  }

  if (dbg && tid == 0) dbg[2] = HOLO_PROBE_CLOCK();
  // ---- epilogue: 16x16x4 D layout: col = lane&15 (Cout), row = 4*(lane>>4) + r (voxel inside the tile)
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;
  const int co = n0 + wn * 16 + lj;
  const int coc = co < p.Cout ? co : p.Cout - 1;
  float bv = (p.nsplit == 1 && p.bias) ? p.bias[coc] : 0.f;
  if (p.nsplit == 1 && p.skip_bias) bv += p.skip_bias[coc];
  float ssum = 0.f, ssq = 0.f;
  // Output rows of this lane: a uniform 64-bit tile base plus 32-bit in-tile offsets.  With a residual, ALL its loads
  // are issued as one batch under a uniform branch: a per-element "if (residual) load" makes the compiler wait
  // for every load (and the store before it) in turn.
  const int64_t tbase = ((((int64_t)cn * p.OD + ctz0) * p.OH + cty0) * p.OW + ctx0) * p.Cout;
  int kql = kq;
  HOLO_LAUNDER(kql);  // recompute the offsets per tile instead of keeping them live across the tile loop
  int off[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int T = wm * MT + t;  // D rows 4*kq .. 4*kq+3 of tile T: same z, y, consecutive x (see a_off)
    off[t] = (((T >> 2) * p.OH + (T & 3) + 4 * (kql >> 1)) * p.OW + 4 * (kql & 1)) * p.Cout;
  }
  if (p.nsplit == 1) {
    if (p.residual) {
      float res[MT][4];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) res[t][r] = ld_act1(p.residual, tbase + coc + off[t] + r * p.Cout, BF);
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += res[t][r];
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[t][r] + bv;
        if (co < p.Cout) st_act1(p.out, tbase + coc + off[t] + r * p.Cout, v, BF && p.out_bf16);
        ssum += v;
        ssq += v * v;
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the address arithmetic of later tiles from being hoisted (VGPRs)
    }
  } else if (co < p.Cout) {
    float* pp = p.partial + (int64_t)blockIdx.z * M * p.Cout + tbase + co;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) pp[off[t] + r * p.Cout] = acc[t][r];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // GroupNorm statistics of the tensor just produced (nn.py:23-25): per output channel (sum, sum of squares)
  // over this workgroup's voxels, one slab per (workgroup, wave row) -> stats[n][slab][Cout][2], reduced by
  // gn_finalize.  Saves a full read pass over the activation.
  if (p.stats && p.nsplit == 1) {
    ssum += __shfl_xor(ssum, 16);
    ssq += __shfl_xor(ssq, 16);
    ssum += __shfl_xor(ssum, 32);
    ssq += __shfl_xor(ssq, 32);
    if (kq == 0 && co < p.Cout) {
      const int tiles_per_sample = ntx * nty * ntz;
      const int slab = (tile % tiles_per_sample) * (4 / NWN) + wm;
      const int nslab = tiles_per_sample * (4 / NWN);
      double* d = p.stats + (((int64_t)cn * nslab + slab) * p.Cout + co) * 2;
      d[0] = (double)ssum;
      d[1] = (double)ssq;
    }
  }
  if (dbg && tid == 0) {
    dbg[3] = HOLO_PROBE_CLOCK();
    unsigned hw, xcc;
    HOLO_PROBE_HWID(hw, xcc);
    dbg[4] = hw;
    dbg[5] = xcc;
  }
  if (SKIP && more_tiles) {
    decode_tile(tile + gridDim.x);
    halo_prepare();
  }
  }  // tile loop
}



// ---------------------------------------------------------------------------------------------
// bf16 wide-tile kernel for the stride-1 3x3x3 convolutions of the filled levels (bf16 compute mode).
//
// v_mfma_f32_32x32x16_bf16 moves 16x the flops of the fp32 instruction per operand byte, so the operand traffic that
// the 128-voxel halo kernel affords (every wave re-reads the whole tile's A operand for its 16 output channels: 9
// fragment loads per 8 MFMAs) is 2x the LDS bandwidth at the bf16 rate.  This kernel register-blocks 4 x NT:
//   workgroup = 4 waves, output tile = 8 x 8 x 8 voxels x 32*NT output channels;
//   wave = 2 z-planes = 128 voxels = 4 MFMA row tiles x NT column tiles (128 accumulator registers at NT = 2);
//   per tap and 16-channel chunk a wave reads 4 A fragments from the LDS halo (ds_read_b128) and NT B fragments
//   straight from global/L1 (all four waves read the same 1 KB blocks) for 4*NT MFMAs: LDS at 50 %, L1 at 50 % of their
//   bandwidth when the matrix pipe is saturated.
// The 10^3 halo of ONE 16-channel chunk is staged per pass: 32-byte LDS rows with no padding, the two 16-byte halves
// of a voxel swapped on odd halo rows (slot = half ^ (hy & 1)).  A 32-row MFMA tile is 4 consecutive y rows x 8 x: a
// row's eight voxels cover one 16-byte slot of every 32-byte bank group and the next row covers the other slot, and a
// tap's ky shifts the parity of every lane alike (two precomputed lane bases).  (The first layout - 48-byte rows,
// tile rows {a, a+4, b, b+4} - measured LDS bank-conflict cycles of 45 % of the LDS-active cycles.)
// GroupNorm*FiLM + SiLU, zero padding, nearest-x2 upsampling and the channel concat are applied while staging, as in
// the halo kernel.  The next chunk's raw halo is requested under tap 16 and committed after the last tap (two barriers
// per chunk); the activation arithmetic of one workgroup overlaps the tap loop of the other resident workgroup (the bf16
// matrix pipe does not use the vector lanes).  A fused 1x1x1 skip connection is one more tap whose A operands come
// straight from global memory (a wave's voxels are its own).  The epilogue transposes the accumulators through LDS
// so that residual, statistics and stores are 16-byte operations.  Split-K over 16-channel chunks.
// (A persistent form - each workgroup walking several tiles, the next tile's first halo requested before the
// epilogue - was measured: the prologue drops from 7 to 4 us per tile but the epilogue's own loads and stores then
// queue behind the halo loads (in-order vmcnt) and it doubles to 18 us; one tile per workgroup is faster.  Two halo
// buffers with one barrier per chunk instead of two: no faster either.)
// IOBF: activations / residual / output are bf16 in HBM (bf16 storage mode).
// ---------------------------------------------------------------------------------------------
constexpr int T_H = 10;                 // halo edge of an 8^3 tile
constexpr int T_HV = T_H * T_H * T_H;   // halo voxels
constexpr int T_CK = 16;                // channels per chunk = K of one v_mfma_f32_32x32x16_bf16
constexpr int T_RS = 8;                 // LDS words per halo voxel (16 bf16, no padding: the halves are swizzled)
constexpr int T_IT = 8;                 // staging items (voxel, 8-channel half) per thread: 2000 / 256

__device__ __forceinline__ void unpack_bf16x8(const float4& v, float (&f)[8]) {
  const uint32_t w0 = __float_as_uint(v.x), w1 = __float_as_uint(v.y), w2 = __float_as_uint(v.z), w3 = __float_as_uint(v.w);
  f[0] = __uint_as_float(w0 << 16);
  f[1] = __uint_as_float(w0 & 0xffff0000u);
  f[2] = __uint_as_float(w1 << 16);
  f[3] = __uint_as_float(w1 & 0xffff0000u);
  f[4] = __uint_as_float(w2 << 16);
  f[5] = __uint_as_float(w2 & 0xffff0000u);
  f[6] = __uint_as_float(w3 << 16);
  f[7] = __uint_as_float(w3 & 0xffff0000u);
}
__device__ __forceinline__ float4 pack_bf16x8(const float (&f)[8]) {
  return make_float4(__uint_as_float(pack_bf16x2(f[0], f[1])), __uint_as_float(pack_bf16x2(f[2], f[3])),
                     __uint_as_float(pack_bf16x2(f[4], f[5])), __uint_as_float(pack_bf16x2(f[6], f[7])));
}

// SCHED: 2 = one operand request behind each MFMA of a tap (measured 6-7 % faster than 0 = requests in a clump between
// the taps' MFMA groups; leaving the order to the compiler was 13 % slower than 0).  HT: the tap under which the next
// chunk's halo is requested (8, 16, 20 and 23 measured within run-to-run noise of each other)
template <int NT, bool SKIP, bool IOBF, int SCHED = 2, int HT = 16>
__global__ __launch_bounds__(256, 2) void conv_bf16t_kernel(ConvParams p) {
  constexpr int ES = IOBF ? 2 : 4;  // bytes per activation element in HBM
  constexpr int NV = IOBF ? 1 : 2;  // 16-byte loads per staging item
  constexpr int BN = 32 * NT;
  // (the epilogue re-uses the halo as four 32 x 68-word transposition tiles: 8 704 words)
  __shared__ __attribute__((aligned(16))) float s_halo[T_HV * T_RS > 4 * 32 * 68 ? T_HV * T_RS : 4 * 32 * 68];
  __shared__ int s_hvox[T_IT * 256];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;  // MFMA row (A) / column (B, D)
  const int kg = lane >> 5;  // MFMA k-group: channels 8*kg .. 8*kg+7 of the chunk
  const int hh = tid & 1;    // staging: which 8-channel half of the chunk
  const int Cin = p.C0 + p.C1;
  const int ncc = (Cin + T_CK - 1) / T_CK;
  const int SCin = p.skip_C0 + p.skip_C1;
  const int nsk = SKIP ? (SCin + T_CK - 1) / T_CK : 0;
  const int cc_begin = blockIdx.z * p.chunks_per_split;
  int cc_end = cc_begin + p.chunks_per_split;
  if (cc_end > ncc) cc_end = ncc;
  const int sk_begin = ncc + blockIdx.z * p.skip_chunks_per_split;
  int sk_end = sk_begin + p.skip_chunks_per_split;
  if (sk_end > ncc + nsk) sk_end = ncc + nsk;
  const int SD = p.ups ? (p.ID >> 1) : p.ID;
  const int SH = p.ups ? (p.IH >> 1) : p.IH;
  const int SW = p.ups ? (p.IW >> 1) : p.IW;
  const int ntx = p.OW >> 3, nty = p.OH >> 3, ntz = p.OD >> 3;
  int bt = blockIdx.x;
  const int tx0 = (bt % ntx) << 3;
  bt /= ntx;
  const int ty0 = (bt % nty) << 3;
  bt /= nty;
  const int tz0 = (bt % ntz) << 3;
  const int n = bt / ntz;
  const int n0 = blockIdx.y * BN;
  unsigned long long* dbg = p.dbg ? p.dbg + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 8 : nullptr;
  if (dbg && tid == 0) dbg[0] = HOLO_PROBE_CLOCK();

  // ---- halo staging
  float4 hreg[T_IT][NV];
  unsigned hvalid = 0, hmask = 0;
  int hcoef_c = 0;
#pragma unroll
  for (int i = 0; i < T_IT; ++i) {
    const int id = tid + 256 * i;
    const int hv = min(id >> 1, T_HV - 1);
    const int hz = hv / (T_H * T_H);
    const int rem = hv - hz * (T_H * T_H);
    const int hy = rem / T_H;
    const int hx = rem - hy * T_H;
    int z = tz0 + hz - 1, y = ty0 + hy - 1, x = tx0 + hx - 1;
    const bool ok = z >= 0 && z < p.ID && y >= 0 && y < p.IH && x >= 0 && x < p.IW && id < 2 * T_HV;
    z = min(max(z, 0), p.ID - 1);
    y = min(max(y, 0), p.IH - 1);
    x = min(max(x, 0), p.IW - 1);
    if (p.ups) {
      z >>= 1;
      y >>= 1;
      x >>= 1;
    }
    s_hvox[i * 256 + tid] = (z * SH + y) * SW + x;
    hvalid |= (ok ? 1u : 0u) << i;
  }
  auto halo_issue = [&](int cc) {
    int c = cc * T_CK + hh * 8;
    const bool cvalid = c < Cin;
    if (!cvalid) c = 0;  // clamped, masked in halo_commit
    hcoef_c = c;
    const float* src = p.src0;
    int Cs = p.C0, cs = c;
    if (c >= p.C0) {
      src = p.src1;
      Cs = p.C1;
      cs = c - p.C0;
    }
    hmask = cvalid ? hvalid : 0u;
    // uniform 64-bit base + one 32-bit byte offset per load (conv_plan keeps a source sample below 4 GB on this path)
    const char* sbase = reinterpret_cast<const char*>(src) + (int64_t)n * SD * SH * SW * Cs * ES;
    const unsigned cbytes = (unsigned)Cs * ES, cofs = (unsigned)cs * ES;
    int tl = tid;
    HOLO_LAUNDER(tl);
#pragma unroll
    for (int i = 0; i < T_IT; ++i) {
      const char* a = sbase + ((unsigned)s_hvox[i * 256 + tl] * cbytes + cofs);
#pragma unroll
      for (int v = 0; v < NV; ++v) hreg[i][v] = *reinterpret_cast<const float4*>(a + 16 * v);
    }
  };
  auto halo_commit = [&]() {
    const bool xform = p.coef != nullptr;
    float ca[8], cb[8];
    if (xform) {
      const float4* cf = reinterpret_cast<const float4*>(p.coef + ((int64_t)n * Cin + hcoef_c) * 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 c = cf[j];  // (a, b) interleaved per channel
        ca[2 * j] = c.x;
        cb[2 * j] = c.y;
        ca[2 * j + 1] = c.z;
        cb[2 * j + 1] = c.w;
      }
    }
#pragma unroll
    for (int i = 0; i < T_IT; ++i) {
      float f[8];
      if (IOBF) {
        unpack_bf16x8(hreg[i][0], f);
      } else {
        f[0] = hreg[i][0].x, f[1] = hreg[i][0].y, f[2] = hreg[i][0].z, f[3] = hreg[i][0].w;
        f[4] = hreg[i][NV - 1].x, f[5] = hreg[i][NV - 1].y, f[6] = hreg[i][NV - 1].z, f[7] = hreg[i][NV - 1].w;
      }
      if (xform) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          f[j] = fmaf(f[j], ca[j], cb[j]);
          if (p.act) f[j] = silu_fast(f[j]);
        }
      }
      const bool keep = (hmask >> i) & 1u;  // zero padding is applied AFTER the activation
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = keep ? f[j] : 0.f;
      const int id = tid + 256 * i;
      const int hy_par = (((id >> 1) / T_H) % T_H) & 1;  // halo row parity of the voxel: which slot its halves go to
      if (id < 2 * T_HV) *reinterpret_cast<float4*>(s_halo + (id >> 1) * T_RS + ((hh ^ hy_par) * 4)) = pack_bf16x8(f);
    }
  };

  f32x16 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // A addressing.  MFMA row li of row tile mt (0..3) of this wave: plane z = 2*wave + (mt>>1), y row 4*(mt&1) + ys with
  // ys = li>>3, x = li&7; the lane's 16-byte half sits in slot kg ^ (halo row parity) = kg ^ (ys&1) ^ (kh&1).
  const int ys = li >> 3;
  const int a_vox = (((2 * wave) * T_H + ys) * T_H + (li & 7)) * T_RS;
  const int a_base0 = a_vox + ((kg ^ (ys & 1)) * 4);      // taps with even kh
  const int a_base1 = a_vox + ((kg ^ (ys & 1) ^ 1) * 4);  // taps with odd kh
  auto load_a = [&](float4 (&a)[4], int tap) {
    const int kd = tap / 9, kh = (tap - kd * 9) / 3, kw = tap - kd * 9 - kh * 3;
    const int toff = ((kd * T_H + kh) * T_H + kw) * T_RS;
    const int ab = (kh & 1) ? a_base1 : a_base0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      a[mt] = *reinterpret_cast<const float4*>(s_halo + ab + ((mt >> 1) * T_H * T_H + 4 * (mt & 1) * T_H) * T_RS + toff);
  };
  // B addressing: 1 KB blocks [tap][chunk][32-Cout slice], 16 bytes per lane
  const int nsl = p.CoutP >> 5;
  const int wncc = p.CinP / T_CK;
  const float* w_lane = reinterpret_cast<const float*>(p.w_bft) + (int64_t)(n0 >> 5) * 256 + lane * 4;
  const float* skw_lane = reinterpret_cast<const float*>(p.skip_w_bft) + (int64_t)(n0 >> 5) * 256 + lane * 4;
  auto load_b = [&](float4 (&b)[NT], int cc, int tap) {
    const float* wp = w_lane + (int64_t)(tap * wncc + cc) * nsl * 256;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const float4*>(wp + nt * 256);
  };
  auto load_b_skip = [&](float4 (&b)[NT], int cc) {
    const float* wp = skw_lane + (int64_t)(cc - ncc) * nsl * 256;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const float4*>(wp + nt * 256);
  };
  auto mfma_tap = [&](const float4 (&a)[4], const float4 (&b)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mfma_bf16_32x32x16(a[mt], b[nt], acc[mt][nt]);
  };

  float4 A[2][4];
  float4 B[3][NT];

  HOLO_PHASE_DELAY(p.stagger_ticks);
  halo_issue(cc_begin);
  load_b(B[0], cc_begin, 0);
  load_b(B[1], cc_begin, 1);
  halo_commit();
  __syncthreads();
  if (dbg && tid == 0) dbg[1] = HOLO_PROBE_CLOCK();
  for (int cc = cc_begin; cc < cc_end; ++cc) {
    const bool has_next = cc + 1 < cc_end;
    load_a(A[0], 0);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      if (tap + 1 < 27) load_a(A[(tap + 1) & 1], tap + 1);
      if (tap + 2 < 27) load_b(B[(tap + 2) % 3], cc, tap + 2);
      if (tap == HT && has_next) halo_issue(cc + 1);  // the next chunk's raw halo flies under taps HT..26
      if (SCHED == 0) __builtin_amdgcn_sched_barrier(0);  // (0: all requests AHEAD of the tap's MFMAs)
      mfma_tap(A[tap & 1], B[tap % 3]);
      if (SCHED == 2 && tap != HT) {  // one operand request behind each MFMA instead of a clump after the eighth
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT - 4 - NT, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (has_next) {
      load_b(B[0], cc + 1, 0);
      load_b(B[1], cc + 1, 1);
    }
    const unsigned long long ta = dbg ? HOLO_PROBE_CLOCK() : 0ull;
    __syncthreads();  // everyone done reading this halo before it is overwritten (by the next chunk or the epilogue)
    if (has_next) {
      halo_commit();
      __syncthreads();
    }
    if (dbg && tid == 0) dbg[6] += HOLO_PROBE_CLOCK() - ta;  // barrier + commit + barrier: time outside the tap loop
  }
  if (SKIP) {
    // Fused 1x1x1 skip connection (unet.py:222,256: skip_connection(x) + h): one tap, no halo, no activation - and a
    // wave's 128 voxels are its own, so its A operands come straight from global memory (16 bytes per lane and row
    // tile: channels 8*kg .. +7 of voxel li), four 16-channel k-steps per round trip; no LDS, no barrier.
    const uint16_t* ssrc0 = reinterpret_cast<const uint16_t*>(p.skip_src0);
    const uint16_t* ssrc1 = reinterpret_cast<const uint16_t*>(p.skip_src1);
    const int64_t vbase = (((int64_t)n * p.OD + tz0 + 2 * wave) * p.OH + ty0 + ys) * p.OW + tx0 + (li & 7);
    int vo[4];  // row tile mt: uniform offsets from the lane's voxel of row tile 0
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) vo[mt] = ((mt >> 1) * p.OH + 4 * (mt & 1)) * p.OW;
    constexpr int SKG = 4;
    for (int g = sk_begin; g < sk_end; g += SKG) {
      float4 SA[SKG][4], SB[SKG][NT];
#pragma unroll
      for (int j = 0; j < SKG; ++j) {
        const int cc = min(g + j, sk_end - 1);
        int c = (cc - ncc) * T_CK + kg * 8;
        if (c >= SCin) c = 0;  // (the packed weights of padding channels are zero)
        const bool second = c >= p.skip_C0;
        const int Cs = second ? p.skip_C1 : p.skip_C0;
        const int cs = second ? c - p.skip_C0 : c;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          if (IOBF) {
            SA[j][mt] = *reinterpret_cast<const float4*>((second ? ssrc1 : ssrc0) + (vbase + vo[mt]) * Cs + cs);
          } else {
            const float* fp = (second ? p.skip_src1 : p.skip_src0) + (vbase + vo[mt]) * Cs + cs;
            const float4 f0 = *reinterpret_cast<const float4*>(fp), f1 = *reinterpret_cast<const float4*>(fp + 4);
            const float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
            SA[j][mt] = pack_bf16x8(f);
          }
        }
        load_b_skip(SB[j], cc);
      }
#pragma unroll
      for (int j = 0; j < SKG; ++j)
        if (g + j < sk_end) mfma_tap(SA[j], SB[j]);  // (uniform)
    }
  }
  if (dbg && tid == 0) dbg[2] = HOLO_PROBE_CLOCK();

  // ---- epilogue.  D layout of 32x32: column = li (Cout), row i = (r&3) + 8*(r>>2) + 4*kg of the row tile (x = i&7,
  // y row 4*(mt&1) + (i>>3)).  Written straight from that layout a lane would issue 128 two-byte stores (measured: 17 of a tile's
  // 68 us); instead each wave passes one 32-voxel row tile at a time through its own slice of the (now dead) halo
  // LDS as fp32 [voxel][channel] and leaves with 8 channels of one voxel per lane: bias, residual, GroupNorm
  // statistics and the store are 16-byte operations on that form.
  constexpr int EW = 68;           // words per voxel row of the transposition tile (16-byte reads conflict free)
  constexpr int LPV = BN / 8;      // lanes per voxel
  constexpr int VPP = 64 / LPV;    // voxels per pass
  constexpr int NPASS = 32 / VPP;
  static_assert(VPP % 8 == 0, "a pass covers whole x rows of the tile (uniform per-pass output offsets)");
  float* s_ep = s_halo + wave * (32 * EW);
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;
  const int ch8 = (lane % LPV) * 8;
  const bool cvalid = n0 + ch8 < p.Cout;
  const int co8 = cvalid ? n0 + ch8 : 0;
  float bv[8], es[8], eq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = es[e] = eq[e] = 0.f;
  if (p.nsplit == 1 && p.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co8), b1 = *reinterpret_cast<const float4*>(p.bias + co8 + 4);
    bv[0] = b0.x, bv[1] = b0.y, bv[2] = b0.z, bv[3] = b0.w, bv[4] = b1.x, bv[5] = b1.y, bv[6] = b1.z, bv[7] = b1.w;
  }
  if (p.nsplit == 1 && p.skip_bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(p.skip_bias + co8), b1 = *reinterpret_cast<const float4*>(p.skip_bias + co8 + 4);
    bv[0] += b0.x, bv[1] += b0.y, bv[2] += b0.z, bv[3] += b0.w, bv[4] += b1.x, bv[5] += b1.y, bv[6] += b1.z, bv[7] += b1.w;
  }
  float* pp = p.nsplit > 1 ? p.partial + (int64_t)blockIdx.z * M * p.Cout : nullptr;
  // output offsets: the lane's voxel of (row tile 0, pass 0) + an offset per (row tile, pass) that is uniform over the wave
  const int i0 = lane / LPV;
  // (32-bit lane part: conv_plan keeps an output sample below 4 GB on this path; the sample base and the pass offset are scalars)
  const unsigned obase = (unsigned)((((tz0 + 2 * wave) * p.OH + ty0 + (i0 >> 3)) * p.OW + tx0 + (i0 & 7)) * p.Cout + co8);
  const int64_t nbase = (int64_t)n * p.OD * p.OH * p.OW * p.Cout;
  auto uoff = [&](int mt, int ps) {
    return nbase + (int64_t)(((mt >> 1) * p.OH + 4 * (mt & 1) + ((ps * VPP) >> 3)) * p.OW) * p.Cout;
  };
  // bf16 storage: the residual of ALL four row tiles is requested up front (one memory latency instead of four in a row;
  // the staging and operand registers of the tap loop are dead here)
  constexpr bool HOIST = IOBF;
  float4 res_all[HOIST ? 4 : 1][NPASS];
  const bool with_res = p.nsplit == 1 && p.residual;  // (uniform)
  if (HOIST && with_res) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps)
        res_all[mt][ps] = *reinterpret_cast<const float4*>(reinterpret_cast<const uint16_t*>(p.residual) + uoff(mt, ps) + obase);
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) s_ep[((r & 3) + 8 * (r >> 2) + 4 * kg) * EW + nt * 32 + li] = acc[mt][nt][r];
    HOLO_WAVE_SYNC();  // (the transposition tile is private to the wave)
    int64_t o[NPASS];
    float4 res[NPASS][2];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      o[ps] = uoff(mt, ps) + obase;
      if (with_res) {
        if (HOIST) {
          res[ps][0] = res_all[mt][ps];
        } else if (IOBF) {
          res[ps][0] = *reinterpret_cast<const float4*>(reinterpret_cast<const uint16_t*>(p.residual) + o[ps]);
        } else {
          res[ps][0] = *reinterpret_cast<const float4*>(p.residual + o[ps]);
          res[ps][1] = *reinterpret_cast<const float4*>(p.residual + o[ps] + 4);
        }
      }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int i = ps * VPP + lane / LPV;
      const float4 v0 = *reinterpret_cast<const float4*>(s_ep + i * EW + ch8);
      const float4 v1 = *reinterpret_cast<const float4*>(s_ep + i * EW + ch8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      if (p.nsplit == 1) {
        if (p.residual) {
          float rf[8];
          if (IOBF) {
            unpack_bf16x8(res[ps][0], rf);
          } else {
            rf[0] = res[ps][0].x, rf[1] = res[ps][0].y, rf[2] = res[ps][0].z, rf[3] = res[ps][0].w;
            rf[4] = res[ps][1].x, rf[5] = res[ps][1].y, rf[6] = res[ps][1].z, rf[7] = res[ps][1].w;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rf[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] += bv[e];
          es[e] += v[e];
          eq[e] += v[e] * v[e];
        }
        if (cvalid) {
          if (IOBF && p.out_bf16) {
            *reinterpret_cast<float4*>(reinterpret_cast<uint16_t*>(p.out) + o[ps]) = pack_bf16x8(v);
          } else {
            *reinterpret_cast<float4*>(p.out + o[ps]) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(p.out + o[ps] + 4) = make_float4(v[4], v[5], v[6], v[7]);
          }
        }
      } else if (cvalid) {
        *reinterpret_cast<float4*>(pp + o[ps]) = v0;
        *reinterpret_cast<float4*>(pp + o[ps] + 4) = v1;
      }
    }
    HOLO_WAVE_SYNC();  // the tile is overwritten by the next row tile
  }
  // GroupNorm statistics of the tensor just produced: one slab per tile (512 voxels) -> stats[n][tile][Cout][2]; the
  // four waves' sums meet in LDS in a fixed order (deterministic)
  if (p.stats && p.nsplit == 1) {
    const int tiles_per_sample = ntx * nty * ntz;
    const int slab = blockIdx.x % tiles_per_sample;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int m = LPV; m < 64; m <<= 1) {
        es[e] += __shfl_xor(es[e], m);
        eq[e] += __shfl_xor(eq[e], m);
      }
    }
    __syncthreads();  // every wave is done with its transposition tile
    if (lane < LPV) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s_halo[(wave * LPV + lane) * 16 + e] = es[e];
        s_halo[(wave * LPV + lane) * 16 + 8 + e] = eq[e];
      }
    }
    __syncthreads();
    if (wave == 0 && lane < LPV && cvalid) {
      double* d = p.stats + (((int64_t)n * tiles_per_sample + slab) * p.Cout + co8) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s1 += s_halo[(w * LPV + lane) * 16 + e];
          s2 += s_halo[(w * LPV + lane) * 16 + 8 + e];
        }
        d[2 * e] = (double)s1;
        d[2 * e + 1] = (double)s2;
      }
    }
  }
  if (dbg && tid == 0) {
    dbg[3] = HOLO_PROBE_CLOCK();
    unsigned hw, xcc;
    HOLO_PROBE_HWID(hw, xcc);
    dbg[4] = hw;
    dbg[5] = xcc;
  }
}


// ---------------------------------------------------------------------------------------------
// Winograd-in-depth form of the halo kernel for the 128-voxel tiles (the 64^3 level: 80 % of the FLOPs).
//
// The exact-fp32 MFMA runs at the vector rate, so the only way below the 27-tap multiply count in fp32 is to multiply
// less.  The 2 x 8 x 8 output tile needs exactly the 4 input planes of ONE Winograd F(2,3) tile along z:
//   V_xi = (B^T d)_xi        d = the 4 activated halo planes of a (y,x) column    B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
//   M_xi = sum_{ky,kx,ci} U_xi[ky][kx][ci][co] * V_xi(y+ky, x+kx, ci)              U_xi = sum_kz G[xi][kz] w[kz]  (prepared once)
//   out(z0) = M_0 + M_1 + M_2,   out(z1) = M_1 - M_2 - M_3
// i.e. 4 x 9 = 36 pseudo-taps produce TWO output planes where the direct form spends 2 x 27 = 54: 2/3 of the MFMAs,
// same data movement.  The transform of the inputs happens ONCE per workgroup while the halo is committed to LDS (the
// thread that stages a (y,x) column holds its four planes): the four LDS planes simply hold V_0..V_3 instead of the raw
// planes, and the tap loop below is the direct kernel's with "plane" read as "xi".  The output transform is lane-local
// (a lane's accumulators of the four xi belong to the same voxels).  Arithmetic: fp32 throughout; F(2,3) adds one
// rounding of an add before and after the products (measured against float64: same error as the direct form).
// The fused 1x1x1 skip connection (centre tap) becomes the two pseudo-taps xi = 1, 2 with weights +w/2, -w/2.
// ---------------------------------------------------------------------------------------------
template <bool SKIP>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(ConvParams p) {
  constexpr int RS = LDK;
  constexpr int MT = 4;                // 16-voxel tiles of the 8 x 8 plane (each accumulates four xi)
  constexpr int PLANE = HY * HX;       // 100 (y,x) columns of the halo
  constexpr int HALO_VOX = 4 * PLANE;  // four xi planes
  constexpr int COLS_IT = (PLANE * 8 + 255) / 256;  // (column, channel quad) items per thread: 4
  __shared__ __attribute__((aligned(16))) float s_halo[HALO_VOX * RS];
  __shared__ int s_hcol[COLS_IT * 256];  // clamped source (y,x) offset of every item, per tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = tid >> 6;  // the wave owns output channels [16 wn, 16 wn + 16) of the block's 64
  const int lj = lane & 15;
  const int kq = lane >> 4;
  const int Cin = p.C0 + p.C1;
  const int ncc = (Cin + BK - 1) / BK;
  const int ntx = p.OW >> 3, nty = p.OH >> 3, ntz = p.OD >> 1;
  int tile = blockIdx.x;
  const int tx0 = (tile % ntx) << 3;
  tile /= ntx;
  const int ty0 = (tile % nty) << 3;
  tile /= nty;
  const int tz0 = (tile % ntz) * 2;
  const int n = tile / ntz;
  const int n0 = blockIdx.y * 64;
  const int SCin = p.skip_C0 + p.skip_C1;
  const int nsk = SKIP ? (SCin + BK - 1) / BK : 0;
  const int cc_begin = blockIdx.z * p.chunks_per_split;
  int cc_end = cc_begin + p.chunks_per_split;
  if (cc_end > ncc) cc_end = ncc;
  const int sk_begin = ncc + blockIdx.z * p.skip_chunks_per_split;
  int sk_end = sk_begin + p.skip_chunks_per_split;
  if (sk_end > ncc + nsk) sk_end = ncc + nsk;
  const int SD = p.ups ? (p.ID >> 1) : p.ID;
  const int SH = p.ups ? (p.IH >> 1) : p.IH;
  const int SW = p.ups ? (p.IW >> 1) : p.IW;

  // ---- staging: item = ((y,x) column, channel quad); the thread loads the column's four planes, activates them,
  //      applies the input transform along z and writes the four xi values
  const int q = tid & 7;
  float4 hreg[COLS_IT][4];
  unsigned cvalid = 0;  // bit i: column of item i is inside the volume (y,x)
  unsigned zvalid = 0;  // bit pl: plane pl is inside the volume (z), uniform
  int zsrc[4];
#pragma unroll
  for (int pl = 0; pl < 4; ++pl) {
    int z = tz0 + pl - 1;
    zvalid |= (z >= 0 && z < p.ID ? 1u : 0u) << pl;
    z = min(max(z, 0), p.ID - 1);
    if (p.ups) z >>= 1;
    zsrc[pl] = z * SH * SW;
  }
#pragma unroll
  for (int i = 0; i < COLS_IT; ++i) {
    const int col = min((tid >> 3) + 32 * i, PLANE - 1);
    const int hy = col / HX, hx = col - hy * HX;
    int y = ty0 + hy - 1, x = tx0 + hx - 1;
    const bool ok = y >= 0 && y < p.IH && x >= 0 && x < p.IW;
    y = min(max(y, 0), p.IH - 1);
    x = min(max(x, 0), p.IW - 1);
    if (p.ups) {
      y >>= 1;
      x >>= 1;
    }
    s_hcol[i * 256 + tid] = y * SW + x;
    cvalid |= (ok ? 1u : 0u) << i;
  }
  int hcoef_c = 0;
  bool h_is_skip = false;
  bool h_cvalid = false;
  auto halo_issue = [&](int cc) {
    h_is_skip = SKIP && cc >= ncc;
    int c = (h_is_skip ? cc - ncc : cc) * BK + q * 4;
    hcoef_c = c;
    h_cvalid = c < (h_is_skip ? SCin : Cin);
    if (!h_cvalid) c = 0;  // clamped, masked below
    const float* src = h_is_skip ? p.skip_src0 : p.src0;
    const int C0s = h_is_skip ? p.skip_C0 : p.C0;
    int Cs = C0s, cs = c;
    if (c >= C0s) {
      src = h_is_skip ? p.skip_src1 : p.src1;
      Cs = h_is_skip ? p.skip_C1 : p.C1;
      cs = c - C0s;
    }
    // unconditional loads from clamped addresses, masked afterwards; uniform base + 32-bit byte offsets
    const char* sbase = reinterpret_cast<const char*>(src + (int64_t)n * SD * SH * SW * Cs);
    const unsigned cbytes = (unsigned)Cs * 4u, cofs = (unsigned)cs * 4u;
    int tl = tid;
    HOLO_LAUNDER(tl);
#pragma unroll
    for (int i = 0; i < COLS_IT; ++i) {
      const unsigned yx = (unsigned)s_hcol[i * 256 + tl];
#pragma unroll
      for (int pl = 0; pl < 4; ++pl)
        hreg[i][pl] = *reinterpret_cast<const float4*>(sbase + (((unsigned)zsrc[pl] + yx) * cbytes + cofs));
    }
  };
  auto halo_commit = [&]() {
    f32x2 a01 = f32x2{1.f, 1.f}, b01 = f32x2{0.f, 0.f}, a23 = a01, b23 = b01;
    const bool xform = p.coef && !h_is_skip;  // the skip path reads the raw block input
    if (xform) {
      const int cc4 = hcoef_c < Cin ? hcoef_c : 0;
      const float4* cf = reinterpret_cast<const float4*>(p.coef + ((int64_t)n * Cin + cc4) * 2);
      const float4 c01 = cf[0], c23 = cf[1];  // (a,b) interleaved per channel
      a01 = f32x2{c01.x, c01.z};
      b01 = f32x2{c01.y, c01.w};
      a23 = f32x2{c23.x, c23.z};
      b23 = f32x2{c23.y, c23.w};
    }
#pragma unroll
    for (int i = 0; i < COLS_IT; ++i) {
      const int col = (tid >> 3) + 32 * i;
      f32x2 v01[4], v23[4];
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) {
        v01[pl] = f32x2{hreg[i][pl].x, hreg[i][pl].y};
        v23[pl] = f32x2{hreg[i][pl].z, hreg[i][pl].w};
        if (xform) {
          v01[pl] = pk_fma(v01[pl], a01, b01);
          v23[pl] = pk_fma(v23[pl], a23, b23);
          if (p.act) {
            v01[pl] = f32x2{silu_f(v01[pl].x), silu_f(v01[pl].y)};
            v23[pl] = f32x2{silu_f(v23[pl].x), silu_f(v23[pl].y)};
          }
        }
        // zero padding is applied AFTER the activation
        const float keep = (h_cvalid && ((cvalid >> i) & 1u) && ((zvalid >> pl) & 1u)) ? 1.f : 0.f;
        const f32x2 k2 = f32x2{keep, keep};
        v01[pl] = pk_mul(v01[pl], k2);
        v23[pl] = pk_mul(v23[pl], k2);
      }
      if (col < PLANE) {
        // B^T d: xi0 = d0 - d2, xi1 = d1 + d2, xi2 = d2 - d1, xi3 = d1 - d3
        const f32x2 x0a = pk_sub(v01[0], v01[2]), x0b = pk_sub(v23[0], v23[2]);
        const f32x2 x1a = pk_add(v01[1], v01[2]), x1b = pk_add(v23[1], v23[2]);
        const f32x2 x2a = pk_sub(v01[2], v01[1]), x2b = pk_sub(v23[2], v23[1]);
        const f32x2 x3a = pk_sub(v01[1], v01[3]), x3b = pk_sub(v23[1], v23[3]);
        float* dst = s_halo + col * RS + q * 4;
        *reinterpret_cast<float4*>(dst + 0 * PLANE * RS) = make_float4(x0a.x, x0a.y, x0b.x, x0b.y);
        *reinterpret_cast<float4*>(dst + 1 * PLANE * RS) = make_float4(x1a.x, x1a.y, x1b.x, x1b.y);
        *reinterpret_cast<float4*>(dst + 2 * PLANE * RS) = make_float4(x2a.x, x2a.y, x2b.x, x2b.y);
        *reinterpret_cast<float4*>(dst + 3 * PLANE * RS) = make_float4(x3a.x, x3a.y, x3b.x, x3b.y);
      }
    }
  };

  f32x4 acc[4][MT];  // [xi][16-voxel tile of the plane]
#pragma unroll
  for (int xi = 0; xi < 4; ++xi)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[xi][t][r] = 0.f;

  // A addressing (as in the direct kernel): the 16 voxels of tile t are x = 0..7 of rows y = t and t + 4
  int a_off[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) a_off[t] = ((t + 4 * (lj >> 3)) * HX + (lj & 7)) * RS + kq * 8;
  const int wncc = p.CinP / BK, wnsl = p.CoutP >> 4;
  constexpr int WBLK = 512;
  const float* w_lane = p.w_wino + (int64_t)((n0 >> 4) + wn) * WBLK + lane * 4;

  auto load_a = [&](float4 (&a)[MT], int pt, int half) {  // pseudo-tap pt = xi * 9 + ky * 3 + kx
    const int xi = pt / 9, kh = (pt - xi * 9) / 3, kw = pt - xi * 9 - kh * 3;
    const int toff = ((xi * HY + kh) * HX + kw) * RS + half * 4;
#pragma unroll
    for (int t = 0; t < MT; ++t) a[t] = *reinterpret_cast<const float4*>(s_halo + a_off[t] + toff);
  };
  auto load_b = [&](float4 (&b)[2], int cc, int pt) {
    const float* wp = w_lane + (int64_t)(pt * wncc + cc) * wnsl * WBLK;
    b[0] = *reinterpret_cast<const float4*>(wp);
    b[1] = *reinterpret_cast<const float4*>(wp + 256);
  };
  auto mfma_half = [&](f32x4 (&ac)[MT], const float4 (&a)[MT], const float4& b) {
#pragma unroll
    for (int t = 0; t < MT; ++t) ac[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, b.x, ac[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) ac[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, b.y, ac[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) ac[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, b.z, ac[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < MT; ++t) ac[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, b.w, ac[t], 0, 0, 0);
  };

  float4 aA[MT] = {}, aB[MT] = {}, aC[MT] = {};  // first half of this pseudo-tap, second half, first half of the next
  float4 b0[2] = {}, b1[2] = {};
  auto tap_body = [&](f32x4 (&ac)[MT], float4 (&cur)[MT], float4 (&nxt)[MT], float4 (&bc)[2], float4 (&bn)[2], int cc,
                      int pt, bool prefetch) {
    load_a(aB, pt, 1);
    if (prefetch) load_b(bn, cc, pt + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the requests above AHEAD of the MFMAs that hide their latency
    mfma_half(ac, cur, bc[0]);
    __builtin_amdgcn_sched_barrier(0);
    if (prefetch) load_a(nxt, pt + 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_half(ac, aB, bc[1]);
    __builtin_amdgcn_sched_barrier(0);
  };

  HOLO_PHASE_DELAY(p.stagger_ticks);
  for (int cc = cc_begin; cc < cc_end; ++cc) {
    halo_issue(cc);
    halo_commit();
    load_b(b0, cc, 0);
    __syncthreads();  // transformed halo of chunk cc visible
    load_a(aA, 0, 0);
#pragma unroll
    for (int pt = 0; pt < 36; pt += 2) {  // fully unrolled: the accumulator set acc[pt / 9] is a compile-time choice
      tap_body(acc[pt / 9], aA, aC, b0, b1, cc, pt, true);
      tap_body(acc[(pt + 1) / 9], aC, aA, b1, b0, cc, pt + 1, pt + 1 < 35);
    }
    __syncthreads();  // everyone done reading this halo before it is overwritten
  }
  if (SKIP) {
    // fused 1x1x1 skip connection: centre (ky,kx) of the block-input halo, pseudo-taps xi = 1, 2 (weights +w/2, -w/2)
    for (int cc = sk_begin; cc < sk_end; ++cc) {
      halo_issue(cc);
      halo_commit();
      const float* wp = p.skip_w_wino + ((int64_t)(cc - ncc) * wnsl + (n0 >> 4) + wn) * WBLK + lane * 4;
      const int64_t tap_stride = (int64_t)(p.skip_CinP / BK) * wnsl * WBLK;
      b0[0] = *reinterpret_cast<const float4*>(wp);
      b0[1] = *reinterpret_cast<const float4*>(wp + 256);
      b1[0] = *reinterpret_cast<const float4*>(wp + tap_stride);
      b1[1] = *reinterpret_cast<const float4*>(wp + tap_stride + 256);
      __syncthreads();
      load_a(aA, 1 * 9 + 4, 0);
      load_a(aB, 1 * 9 + 4, 1);
      mfma_half(acc[1], aA, b0[0]);
      mfma_half(acc[1], aB, b0[1]);
      load_a(aA, 2 * 9 + 4, 0);
      load_a(aB, 2 * 9 + 4, 1);
      mfma_half(acc[2], aA, b1[0]);
      mfma_half(acc[2], aB, b1[1]);
      __syncthreads();
    }
  }

  // ---- output transform (lane-local) + epilogue.  16x16x4 D layout: col = lane&15 (Cout), row = 4*(lane>>4) + r
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;
  const int co = n0 + wn * 16 + lj;
  const int coc = co < p.Cout ? co : p.Cout - 1;
  float bv = (p.nsplit == 1 && p.bias) ? p.bias[coc] : 0.f;
  if (p.nsplit == 1 && p.skip_bias) bv += p.skip_bias[coc];
  float ssum = 0.f, ssq = 0.f;
  const int64_t tbase = ((((int64_t)n * p.OD + tz0) * p.OH + ty0) * p.OW + tx0) * p.Cout;
  const int zstride = p.OH * p.OW * p.Cout;
  int off[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) off[t] = ((t + 4 * (kq >> 1)) * p.OW + 4 * (kq & 1)) * p.Cout;
  // out(z0) = M0 + M1 + M2, out(z1) = M1 - M2 - M3, written over acc[0] / acc[3]
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float m0 = acc[0][t][r], m1 = acc[1][t][r], m2 = acc[2][t][r], m3 = acc[3][t][r];
      acc[0][t][r] = (m0 + m1) + m2;
      acc[3][t][r] = (m1 - m2) - m3;
    }
  if (p.nsplit == 1) {
#pragma unroll
    for (int z = 0; z < 2; ++z) {
      f32x4 (&o)[MT] = z ? acc[3] : acc[0];
      if (p.residual) {
        const float* rp = p.residual + tbase + z * (int64_t)zstride + coc;
        float res[MT][4];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) res[t][r] = rp[off[t] + r * p.Cout];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) o[t][r] += res[t][r];
      }
      float* op = p.out + tbase + z * (int64_t)zstride + coc;
#pragma unroll
      for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = o[t][r] + bv;
          if (co < p.Cout) op[off[t] + r * p.Cout] = v;
          ssum += v;
          ssq += v * v;
        }
      }
    }
  } else if (co < p.Cout) {
    float* pp = p.partial + (int64_t)blockIdx.z * M * p.Cout + tbase + co;
#pragma unroll
    for (int z = 0; z < 2; ++z) {
      f32x4 (&o)[MT] = z ? acc[3] : acc[0];
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[z * (int64_t)zstride + off[t] + r * p.Cout] = o[t][r];
    }
  }
  // GroupNorm statistics of the tensor just produced: one slab per workgroup, as the direct 128-voxel kernel
  if (p.stats && p.nsplit == 1) {
    ssum += __shfl_xor(ssum, 16);
    ssq += __shfl_xor(ssq, 16);
    ssum += __shfl_xor(ssum, 32);
    ssq += __shfl_xor(ssq, 32);
    if (kq == 0 && co < p.Cout) {
      const int tiles_per_sample = ntx * nty * ntz;
      const int slab = (int)blockIdx.x % tiles_per_sample;
      double* d = p.stats + (((int64_t)n * tiles_per_sample + slab) * p.Cout + co) * 2;
      d[0] = (double)ssum;
      d[1] = (double)ssq;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) over (z, y): the same kernel with the second transform applied to the A operand at load time
// (see "second Winograd dimension" below).  The LDS halo is the z-transformed one of conv_wino_kernel, unchanged.
// ---------------------------------------------------------------------------------------------
// NWN: waves along Cout.  4: the workgroup covers 64 output channels, every wave both MFMA tiles of the plane pair;
// 2 (32-channel convolutions, e.g. the output conv): waves 0,1 take tile 0 and waves 2,3 tile 1 of the same 32 channels.
template <bool SKIP, int NWN = 4>
__global__ __launch_bounds__(256, 2) void conv_wino2_kernel(ConvParams p) {
  constexpr int RS = LDK;
  constexpr int MT = NWN == 4 ? 2 : 1;  // 16-row MFMA tiles per wave: rows = (y tile, x), y tiles T and T + 2 (see a_off)
  constexpr int PLANE = HY * HX;       // 100 (y,x) columns of the halo
  constexpr int HALO_VOX = 4 * PLANE;  // four xi planes
  constexpr int COLS_IT = (PLANE * 8 + 255) / 256;  // (column, channel quad) items per thread: 4
  __shared__ __attribute__((aligned(16))) float s_halo[HALO_VOX * RS];
  __shared__ int s_hcol[COLS_IT * 256];  // clamped source (y,x) offset of every item, per tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = (tid >> 6) % NWN;             // the wave owns output channels [16 wn, 16 wn + 16) of the block's 16*NWN
  const int wm = NWN == 4 ? 0 : (tid >> 6) / NWN;  // ... and MFMA tiles wm*MT .. wm*MT + MT - 1
  const int lj = lane & 15;
  const int kq = lane >> 4;
  const int Cin = p.C0 + p.C1;
  const int ncc = (Cin + BK - 1) / BK;
  const int ntx = p.OW >> 3, nty = p.OH >> 3, ntz = p.OD >> 1;
  int tile = blockIdx.x;
  const int tx0 = (tile % ntx) << 3;
  tile /= ntx;
  const int ty0 = (tile % nty) << 3;
  tile /= nty;
  const int tz0 = (tile % ntz) * 2;
  const int n = tile / ntz;
  const int n0 = blockIdx.y * (16 * NWN);
  const int SCin = p.skip_C0 + p.skip_C1;
  const int nsk = SKIP ? (SCin + BK - 1) / BK : 0;
  const int cc_begin = blockIdx.z * p.chunks_per_split;
  int cc_end = cc_begin + p.chunks_per_split;
  if (cc_end > ncc) cc_end = ncc;
  const int sk_begin = ncc + blockIdx.z * p.skip_chunks_per_split;
  int sk_end = sk_begin + p.skip_chunks_per_split;
  if (sk_end > ncc + nsk) sk_end = ncc + nsk;
  const int SD = p.ups ? (p.ID >> 1) : p.ID;
  const int SH = p.ups ? (p.IH >> 1) : p.IH;
  const int SW = p.ups ? (p.IW >> 1) : p.IW;

  // ---- staging: item = ((y,x) column, channel quad); the thread loads the column's four planes, activates them,
  //      applies the input transform along z and writes the four xi values
  const int q = tid & 7;
  float4 hreg[COLS_IT][4];
  unsigned cvalid = 0;  // bit i: column of item i is inside the volume (y,x)
  unsigned zvalid = 0;  // bit pl: plane pl is inside the volume (z), uniform
  int zsrc[4];
#pragma unroll
  for (int pl = 0; pl < 4; ++pl) {
    int z = tz0 + pl - 1;
    zvalid |= (z >= 0 && z < p.ID ? 1u : 0u) << pl;
    z = min(max(z, 0), p.ID - 1);
    if (p.ups) z >>= 1;
    zsrc[pl] = z * SH * SW;
  }
#pragma unroll
  for (int i = 0; i < COLS_IT; ++i) {
    const int col = min((tid >> 3) + 32 * i, PLANE - 1);
    const int hy = col / HX, hx = col - hy * HX;
    int y = ty0 + hy - 1, x = tx0 + hx - 1;
    const bool ok = y >= 0 && y < p.IH && x >= 0 && x < p.IW;
    y = min(max(y, 0), p.IH - 1);
    x = min(max(x, 0), p.IW - 1);
    if (p.ups) {
      y >>= 1;
      x >>= 1;
    }
    s_hcol[i * 256 + tid] = y * SW + x;
    cvalid |= (ok ? 1u : 0u) << i;
  }
  int hcoef_c = 0;
  bool h_is_skip = false;
  bool h_cvalid = false;
  auto halo_issue = [&](int cc) {
    h_is_skip = SKIP && cc >= ncc;
    int c = (h_is_skip ? cc - ncc : cc) * BK + q * 4;
    hcoef_c = c;
    h_cvalid = c < (h_is_skip ? SCin : Cin);
    if (!h_cvalid) c = 0;  // clamped, masked below
    const float* src = h_is_skip ? p.skip_src0 : p.src0;
    const int C0s = h_is_skip ? p.skip_C0 : p.C0;
    int Cs = C0s, cs = c;
    if (c >= C0s) {
      src = h_is_skip ? p.skip_src1 : p.src1;
      Cs = h_is_skip ? p.skip_C1 : p.C1;
      cs = c - C0s;
    }
    // unconditional loads from clamped addresses, masked afterwards; uniform base + 32-bit byte offsets
    const char* sbase = reinterpret_cast<const char*>(src + (int64_t)n * SD * SH * SW * Cs);
    const unsigned cbytes = (unsigned)Cs * 4u, cofs = (unsigned)cs * 4u;
    int tl = tid;
    HOLO_LAUNDER(tl);
#pragma unroll
    for (int i = 0; i < COLS_IT; ++i) {
      const unsigned yx = (unsigned)s_hcol[i * 256 + tl];
#pragma unroll
      for (int pl = 0; pl < 4; ++pl)
        hreg[i][pl] = *reinterpret_cast<const float4*>(sbase + (((unsigned)zsrc[pl] + yx) * cbytes + cofs));
    }
  };
  auto halo_commit = [&]() {
    f32x2 a01 = f32x2{1.f, 1.f}, b01 = f32x2{0.f, 0.f}, a23 = a01, b23 = b01;
    const bool xform = p.coef && !h_is_skip;  // the skip path reads the raw block input
    if (xform) {
      const int cc4 = hcoef_c < Cin ? hcoef_c : 0;
      const float4* cf = reinterpret_cast<const float4*>(p.coef + ((int64_t)n * Cin + cc4) * 2);
      const float4 c01 = cf[0], c23 = cf[1];  // (a,b) interleaved per channel
      a01 = f32x2{c01.x, c01.z};
      b01 = f32x2{c01.y, c01.w};
      a23 = f32x2{c23.x, c23.z};
      b23 = f32x2{c23.y, c23.w};
    }
#pragma unroll
    for (int i = 0; i < COLS_IT; ++i) {
      const int col = (tid >> 3) + 32 * i;
      f32x2 v01[4], v23[4];
#pragma unroll
      for (int pl = 0; pl < 4; ++pl) {
        v01[pl] = f32x2{hreg[i][pl].x, hreg[i][pl].y};
        v23[pl] = f32x2{hreg[i][pl].z, hreg[i][pl].w};
        if (xform) {
          v01[pl] = pk_fma(v01[pl], a01, b01);
          v23[pl] = pk_fma(v23[pl], a23, b23);
          if (p.act) {
            v01[pl] = f32x2{silu_f(v01[pl].x), silu_f(v01[pl].y)};
            v23[pl] = f32x2{silu_f(v23[pl].x), silu_f(v23[pl].y)};
          }
        }
        // zero padding is applied AFTER the activation
        const float keep = (h_cvalid && ((cvalid >> i) & 1u) && ((zvalid >> pl) & 1u)) ? 1.f : 0.f;
        const f32x2 k2 = f32x2{keep, keep};
        v01[pl] = pk_mul(v01[pl], k2);
        v23[pl] = pk_mul(v23[pl], k2);
      }
      if (col < PLANE) {
        // B^T d: xi0 = d0 - d2, xi1 = d1 + d2, xi2 = d2 - d1, xi3 = d1 - d3
        const f32x2 x0a = pk_sub(v01[0], v01[2]), x0b = pk_sub(v23[0], v23[2]);
        const f32x2 x1a = pk_add(v01[1], v01[2]), x1b = pk_add(v23[1], v23[2]);
        const f32x2 x2a = pk_sub(v01[2], v01[1]), x2b = pk_sub(v23[2], v23[1]);
        const f32x2 x3a = pk_sub(v01[1], v01[3]), x3b = pk_sub(v23[1], v23[3]);
        float* dst = s_halo + col * RS + q * 4;
        *reinterpret_cast<float4*>(dst + 0 * PLANE * RS) = make_float4(x0a.x, x0a.y, x0b.x, x0b.y);
        *reinterpret_cast<float4*>(dst + 1 * PLANE * RS) = make_float4(x1a.x, x1a.y, x1b.x, x1b.y);
        *reinterpret_cast<float4*>(dst + 2 * PLANE * RS) = make_float4(x2a.x, x2a.y, x2b.x, x2b.y);
        *reinterpret_cast<float4*>(dst + 3 * PLANE * RS) = make_float4(x3a.x, x3a.y, x3b.x, x3b.y);
      }
    }
  };

  // ---- second Winograd dimension (y): an MFMA row is a (y tile, x) pair - two output rows, eight x; its A operand
  //      for pseudo-tap (xi_z, xi_y, kx) is the B^T combination of two of the tile's four halo rows of plane xi_z:
  //      xi_y0 = r0 - r2, xi_y1 = r1 + r2, xi_y2 = r2 - r1, xi_y3 = r1 - r3, formed in registers from four 16-byte
  //      LDS reads that serve all four xi_y.  16 accumulator sets [xi_z][xi_y] per tile, 48 pseudo-taps per chunk
  //      for a 2 x 2 (z,y) block of outputs where the direct form spends 108: 4/9 of the MFMAs.
  f32x4 acc[16][MT];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[xi][t][r] = 0.f;

  // A rows: MFMA row lj of tile t = y tile (t + 2*(lj>>3)), x = lj&7; its halo rows are 2*ytile + a, a = 0..3
  // (the two y tiles of an MFMA tile are 4 halo rows = 32 (mod 64) words apart: conflict-free 16-byte reads)
  int a_off[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) a_off[t] = ((2 * (wm * MT + t) + 4 * (lj >> 3)) * HX + (lj & 7)) * RS + kq * 8;
  const int wncc = p.CinP / BK, wnsl = p.CoutP >> 4;
  constexpr int WBLK = 512;
  const float* w_lane = p.w_wino2 + (int64_t)((n0 >> 4) + wn) * WBLK + lane * 4;
  const int64_t pt_stride = (int64_t)wncc * wnsl * WBLK;  // between pseudo-taps pt = (xi_z*4 + xi_y)*3 + kx

  // one step = (xi_z, kx, half of the chunk's k-steps): rows -> four xi_y operands per tile -> 32 MFMAs
  auto load_rows = [&](float4 (&R)[4], int t, int xz, int kw, int half) {
    const float* base = s_halo + a_off[t] + ((xz * HY) * HX + kw) * RS + half * 4;
#pragma unroll
    for (int a = 0; a < 4; ++a) R[a] = *reinterpret_cast<const float4*>(base + a * HX * RS);
  };
  auto load_w = [&](float4 (&B)[4], int cc, int xz, int kw, int half) {
    const float* wp = w_lane + (int64_t)cc * wnsl * WBLK + half * 256 + (int64_t)((xz * 4) * 3 + kw) * pt_stride;
#pragma unroll
    for (int xy = 0; xy < 4; ++xy) B[xy] = *reinterpret_cast<const float4*>(wp + (int64_t)(xy * 3) * pt_stride);
  };
  auto combine = [&](float4 (&R)[4]) {  // in place: R[xi_y]; eight v_pk_add_f32
    const f32x2 r0a = f32x2{R[0].x, R[0].y}, r0b = f32x2{R[0].z, R[0].w}, r1a = f32x2{R[1].x, R[1].y},
                r1b = f32x2{R[1].z, R[1].w}, r2a = f32x2{R[2].x, R[2].y}, r2b = f32x2{R[2].z, R[2].w},
                r3a = f32x2{R[3].x, R[3].y}, r3b = f32x2{R[3].z, R[3].w};
    const f32x2 y0a = pk_sub(r0a, r2a), y0b = pk_sub(r0b, r2b);
    const f32x2 y1a = pk_add(r1a, r2a), y1b = pk_add(r1b, r2b);
    const f32x2 y2a = pk_sub(r2a, r1a), y2b = pk_sub(r2b, r1b);
    const f32x2 y3a = pk_sub(r1a, r3a), y3b = pk_sub(r1b, r3b);
    R[0] = make_float4(y0a.x, y0a.y, y0b.x, y0b.y);
    R[1] = make_float4(y1a.x, y1a.y, y1b.x, y1b.y);
    R[2] = make_float4(y2a.x, y2a.y, y2b.x, y2b.y);
    R[3] = make_float4(y3a.x, y3a.y, y3b.x, y3b.y);
  };
  // the four xi_y accumulators of a tile advance together, k-step by k-step: consecutive MFMAs are independent (a
  // dependent one would wait out the 8 passes of its predecessor)
  auto mfma_xy = [&](int xz, int t, const float4 (&A)[4], const float4 (&B)[4]) {
#pragma unroll
    for (int xy = 0; xy < 4; ++xy)
      acc[xz * 4 + xy][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[xy].x, B[xy].x, acc[xz * 4 + xy][t], 0, 0, 0);
#pragma unroll
    for (int xy = 0; xy < 4; ++xy)
      acc[xz * 4 + xy][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[xy].y, B[xy].y, acc[xz * 4 + xy][t], 0, 0, 0);
#pragma unroll
    for (int xy = 0; xy < 4; ++xy)
      acc[xz * 4 + xy][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[xy].z, B[xy].z, acc[xz * 4 + xy][t], 0, 0, 0);
#pragma unroll
    for (int xy = 0; xy < 4; ++xy)
      acc[xz * 4 + xy][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[xy].w, B[xy].w, acc[xz * 4 + xy][t], 0, 0, 0);
  };
  auto mfma4 = [&](f32x4& ac, const float4& a, const float4& b) {
    ac = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, ac, 0, 0, 0);
    ac = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, ac, 0, 0, 0);
    ac = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, ac, 0, 0, 0);
    ac = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, ac, 0, 0, 0);
  };

  HOLO_PHASE_DELAY(p.stagger_ticks);
  unsigned long long* dbg = p.dbg ? p.dbg + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 8 : nullptr;
  unsigned long long t_stage = 0;
  if (dbg && tid == 0) dbg[0] = HOLO_PROBE_CLOCK();
  for (int cc = cc_begin; cc < cc_end; ++cc) {
    const unsigned long long ts0 = dbg ? HOLO_PROBE_CLOCK() : 0ull;
    halo_issue(cc);
    halo_commit();
    // weights: global -> registers, requested TWO steps (2 x 32 MFMAs, ~0.85 us of matrix time) ahead of their use -
    // one step does not cover an L2 round trip; the A rows use ONE buffer: a tile's rows are re-requested as soon as
    // the MFMAs that read the previous ones have been issued, their ~100 cycles hide behind those queued MFMAs
    float4 Bw[3][4];
    load_w(Bw[0], cc, 0, 0, 0);
    load_w(Bw[1], cc, 0, 0, 1);
    __syncthreads();  // transformed halo of chunk cc visible
    if (dbg) {
      const unsigned long long ts1 = HOLO_PROBE_CLOCK();
      t_stage += ts1 - ts0;
      if (tid == 0 && cc == cc_begin) dbg[1] = ts1;
    }
    float4 R0[4];
    load_rows(R0, 0, 0, 0, 0);
#pragma unroll
    for (int st = 0; st < 24; ++st) {  // fully unrolled: accumulator sets and ring slots are compile-time choices
      const int xz = st / 6, kw = (st % 6) >> 1, half = st & 1;
      const int nx = st + 1, nxz = nx / 6, nkw = (nx % 6) >> 1, nhalf = nx & 1;
      const int n2 = st + 2, n2xz = n2 / 6, n2kw = (n2 % 6) >> 1, n2half = n2 & 1;
      if (st + 2 < 24) load_w(Bw[(st + 2) % 3], cc, n2xz, n2kw, n2half);
      __builtin_amdgcn_sched_barrier(0);  // requests stay AHEAD of the MFMAs that hide them
      combine(R0);
      mfma_xy(xz, 0, R0, Bw[st % 3]);
      __builtin_amdgcn_sched_barrier(0);
      if (MT == 2) {
        load_rows(R0, 1, xz, kw, half);
        __builtin_amdgcn_sched_barrier(0);
        combine(R0);
        mfma_xy(xz, MT - 1, R0, Bw[st % 3]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (st + 1 < 24) load_rows(R0, 0, nxz, nkw, nhalf);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // everyone done reading this halo before it is overwritten
  }
  if (SKIP) {
    // fused 1x1x1 skip connection: centre tap = pseudo-taps (xi_z, xi_y) in {1,2}^2 at kx = 1, weights +-w/4
    for (int cc = sk_begin; cc < sk_end; ++cc) {
      halo_issue(cc);
      halo_commit();
      const float* wp = p.skip_w_wino2 + ((int64_t)(cc - ncc) * wnsl + (n0 >> 4) + wn) * WBLK + lane * 4;
      const int64_t tap_stride = (int64_t)(p.skip_CinP / BK) * wnsl * WBLK;
      __syncthreads();
#pragma unroll
      for (int xz = 1; xz <= 2; ++xz)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          float4 B1 = *reinterpret_cast<const float4*>(wp + (int64_t)((xz - 1) * 2 + 0) * tap_stride + half * 256);
          float4 B2 = *reinterpret_cast<const float4*>(wp + (int64_t)((xz - 1) * 2 + 1) * tap_stride + half * 256);
#pragma unroll
          for (int t = 0; t < MT; ++t) {
            float4 R[4];
            load_rows(R, t, xz, 1, half);
            combine(R);
            mfma4(acc[xz * 4 + 1][t], R[1], B1);
            mfma4(acc[xz * 4 + 2][t], R[2], B2);
          }
        }
      __syncthreads();
    }
  }

  if (dbg && tid == 0) {
    dbg[2] = HOLO_PROBE_CLOCK();
    dbg[6] = t_stage;
  }
  // ---- output transform (lane-local, y then z) + epilogue.  D row 4*kq + r of tile t = y tile t + 2*(kq>>1), x = 4*(kq&1) + r
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;
  const int co = n0 + wn * 16 + lj;
  const int coc = co < p.Cout ? co : p.Cout - 1;
  float bv = (p.nsplit == 1 && p.bias) ? p.bias[coc] : 0.f;
  if (p.nsplit == 1 && p.skip_bias) bv += p.skip_bias[coc];
  float ssum = 0.f, ssq = 0.f;
  const int64_t tbase = ((((int64_t)n * p.OD + tz0) * p.OH + ty0) * p.OW + tx0) * p.Cout;
  const int zstride = p.OH * p.OW * p.Cout;
  const int ystride = p.OW * p.Cout;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int off = ((2 * (wm * MT + t + 2 * (kq >> 1))) * p.OW + 4 * (kq & 1)) * p.Cout;  // output row 2*ytile, first x of the lane
    float o[2][2][4];  // [z][y][r]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float py[4][2];  // [xi_z][y]
#pragma unroll
      for (int xz = 0; xz < 4; ++xz) {
        const float m0 = acc[xz * 4 + 0][t][r], m1 = acc[xz * 4 + 1][t][r], m2 = acc[xz * 4 + 2][t][r],
                    m3 = acc[xz * 4 + 3][t][r];
        py[xz][0] = (m0 + m1) + m2;
        py[xz][1] = (m1 - m2) - m3;
      }
#pragma unroll
      for (int y = 0; y < 2; ++y) {
        o[0][y][r] = (py[0][y] + py[1][y]) + py[2][y];
        o[1][y][r] = (py[1][y] - py[2][y]) - py[3][y];
      }
    }
    if (p.nsplit == 1) {
      if (p.residual) {
        const float* rp = p.residual + tbase + off + coc;
        float res[2][2][4];
#pragma unroll
        for (int z = 0; z < 2; ++z)
#pragma unroll
          for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) res[z][y][r] = rp[z * (int64_t)zstride + y * ystride + r * p.Cout];
#pragma unroll
        for (int z = 0; z < 2; ++z)
#pragma unroll
          for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[z][y][r] += res[z][y][r];
      }
      float* op = p.out + tbase + off + coc;
#pragma unroll
      for (int z = 0; z < 2; ++z)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = o[z][y][r] + bv;
            if (co < p.Cout) op[z * (int64_t)zstride + y * ystride + r * p.Cout] = v;
            ssum += v;
            ssq += v * v;
          }
    } else if (co < p.Cout) {
      float* pp = p.partial + (int64_t)blockIdx.z * M * p.Cout + tbase + off + co;
#pragma unroll
      for (int z = 0; z < 2; ++z)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int r = 0; r < 4; ++r) pp[z * (int64_t)zstride + y * ystride + r * p.Cout] = o[z][y][r];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // GroupNorm statistics of the tensor just produced: one slab per workgroup, as the direct 128-voxel kernel
  if (p.stats && p.nsplit == 1) {
    ssum += __shfl_xor(ssum, 16);
    ssq += __shfl_xor(ssq, 16);
    ssum += __shfl_xor(ssum, 32);
    ssq += __shfl_xor(ssq, 32);
    if (kq == 0 && co < p.Cout) {
      const int tiles_per_sample = ntx * nty * ntz;
      const int slab = ((int)blockIdx.x % tiles_per_sample) * (4 / NWN) + wm;  // one slab per (workgroup, wave row)
      const int nslab = tiles_per_sample * (4 / NWN);
      double* d = p.stats + (((int64_t)n * nslab + slab) * p.Cout + co) * 2;
      d[0] = (double)ssum;
      d[1] = (double)ssq;
    }
  }
  if (dbg && tid == 0) {
    dbg[3] = HOLO_PROBE_CLOCK();
    unsigned hw, xcc;
    HOLO_PROBE_HWID(hw, xcc);
    dbg[4] = hw;
    dbg[5] = xcc;
  }
}


// ---------------------------------------------------------------------------------------------
// fp32-accurate convolution on the bf16 matrix cores ("bf16x3 split", opt-in: ConvParams::bf16 == 2).
//
// On gfx950 the exact-fp32 MFMA runs on the vector FMA lanes (157 TF, shared with every VALU instruction); the bf16
// MFMA is a separate unit 16x faster.  Every fp32 operand is split EXACTLY into three bf16 terms
//     x = hi + mid + lo,   hi = rne_bf16(x), mid = rne_bf16(x - hi), lo = rne_bf16(x - hi - mid)
// (24 = 3 x 8 mantissa bits; the two subtractions are exact in fp32) and the product x*w is assembled from the six
// leading cross terms  hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid  (each a bf16 x bf16 product, exact in the
// fp32 accumulator); the dropped terms are <= 2^-23 |x w| relative, i.e. the size of ONE fp32 rounding of the
// product.  Six bf16 MFMAs cost 6/16 of the fp32 MFMA they replace and leave the vector lanes to the staging work.
// The activations are split as the halo is committed to LDS (three copies, 64-byte rows, XOR-swizzled 16-byte slots
// so that the A reads are conflict free without padding); the weights are split once at set_param.
// Structure = conv_halo_kernel with 64-voxel tiles (1 x 8 x 8), 4 waves x 16 Cout, no fused skip.
// ---------------------------------------------------------------------------------------------
// TZ = 1: 64-voxel tiles, 4 waves (2 workgroups per CU).  TZ = 2: 128-voxel tiles, 8 waves (wave = 16-Cout slice x
// z-slab; one workgroup per CU, same 8 waves per CU): each weight block is then fetched by two waves of ONE
// workgroup (the second hits L1), halving the L2 -> CU weight traffic that bounds the 64-voxel form at 64^3.
template <int TZ>
__global__ __launch_bounds__(256 * TZ, 2 / TZ) void conv_halo_split_kernel(ConvParams p) {
  constexpr int NT = 256 * TZ;                  // threads per workgroup
  constexpr int MT = 2;                         // 16-voxel tiles per wave   } 2 x 2 register blocking: every A fragment
  constexpr int NS = 2;                         // 16-Cout slices per wave   } feeds two MFMAs, every B fragment two
  constexpr int HALO_VOX = (TZ + 2) * HY * HX;  // 300 / 400
  constexpr int RPT = NT / 8;                   // halo rows staged per pass (8 threads per row)
  constexpr int HALO_IT = (HALO_VOX + RPT - 1) / RPT;
  constexpr int RW = 16;                        // words per halo row (32 bf16), no padding: swizzled slots
  constexpr int PLANE = HALO_VOX * RW;          // words per copy
  __shared__ __attribute__((aligned(16))) uint32_t s_halo[3 * PLANE];
  __shared__ int s_hvox[HALO_IT * NT];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wc = wave & 1;         // Cout half of the 64-wide block: slices 2*wc, 2*wc+1
  const int wv = (wave >> 1) & 1;  // voxel half of the 8 x 8 slab: MFMA tiles 2*wv, 2*wv+1
  const int wz = wave >> 2;        // z-slab of the tile (TZ = 2)
  const int lj = lane & 15;
  const int kq = lane >> 4;
  const int Cin = p.C0 + p.C1;
  const int ncc = (Cin + BK - 1) / BK;
  const int ntx = p.OW >> 3, nty = p.OH >> 3, ntz = p.OD / TZ;
  int bt = blockIdx.x;
  const int tx0 = (bt % ntx) << 3;
  bt /= ntx;
  const int ty0 = (bt % nty) << 3;
  bt /= nty;
  const int tz0 = (bt % ntz) * TZ;
  const int n = bt / ntz;
  const int n0 = blockIdx.y * 64;
  const int cc_begin = blockIdx.z * p.chunks_per_split;
  int cc_end = cc_begin + p.chunks_per_split;
  if (cc_end > ncc) cc_end = ncc;
  const int SD = p.ups ? (p.ID >> 1) : p.ID;
  const int SH = p.ups ? (p.IH >> 1) : p.IH;
  const int SW = p.ups ? (p.IW >> 1) : p.IW;
  const int q = tid & 7;
  const int r0 = tid >> 3;

  // ---- halo coordinates once per tile (parked in LDS), loads from clamped addresses, see conv_halo_kernel
  unsigned hvalid = 0, hmask = 0;
#pragma unroll
  for (int i = 0; i < HALO_IT; ++i) {
    const int hv = min(r0 + RPT * i, HALO_VOX - 1);
    const int hz = hv / (HY * HX);
    const int rem = hv - hz * (HY * HX);
    const int hy = rem / HX;
    const int hx = rem - hy * HX;
    int z = tz0 + hz - 1, y = ty0 + hy - 1, x = tx0 + hx - 1;
    const bool ok = z >= 0 && z < p.ID && y >= 0 && y < p.IH && x >= 0 && x < p.IW;
    z = min(max(z, 0), p.ID - 1);
    y = min(max(y, 0), p.IH - 1);
    x = min(max(x, 0), p.IW - 1);
    if (p.ups) {
      z >>= 1;
      y >>= 1;
      x >>= 1;
    }
    s_hvox[i * NT + tid] = (z * SH + y) * SW + x;
    hvalid |= (ok ? 1u : 0u) << i;
  }
  float4 hreg[HALO_IT];
  int hcoef_c = 0;
  auto halo_issue = [&](int cc) {
    int c = cc * BK + q * 4;
    hcoef_c = c;
    const bool cvalid = c < Cin;
    if (!cvalid) c = 0;
    const float* src = p.src0;
    int Cs = p.C0, cs = c;
    if (c >= p.C0) {
      src = p.src1;
      Cs = p.C1;
      cs = c - p.C0;
    }
    hmask = cvalid ? hvalid : 0u;
    const char* sbase = reinterpret_cast<const char*>(src + (int64_t)n * SD * SH * SW * Cs);
    const unsigned cbytes = (unsigned)Cs * 4u, cofs = (unsigned)cs * 4u;
    int tl = tid;
    HOLO_LAUNDER(tl);
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i)
      hreg[i] = *reinterpret_cast<const float4*>(sbase + ((unsigned)s_hvox[i * NT + tl] * cbytes + cofs));
  };
  auto halo_commit = [&]() {
    f32x2 a01 = f32x2{1.f, 1.f}, b01 = f32x2{0.f, 0.f}, a23 = a01, b23 = b01;
    if (p.coef) {
      const int cc4 = hcoef_c < Cin ? hcoef_c : 0;
      const float4* cf = reinterpret_cast<const float4*>(p.coef + ((int64_t)n * Cin + cc4) * 2);
      const float4 c01 = cf[0], c23 = cf[1];
      a01 = f32x2{c01.x, c01.z};
      b01 = f32x2{c01.y, c01.w};
      a23 = f32x2{c23.x, c23.z};
      b23 = f32x2{c23.y, c23.w};
    }
#pragma unroll
    for (int i = 0; i < HALO_IT; ++i) {
      const int hv = r0 + RPT * i;
      f32x2 v01 = f32x2{hreg[i].x, hreg[i].y}, v23 = f32x2{hreg[i].z, hreg[i].w};
      if (p.coef) {
        v01 = pk_fma(v01, a01, b01);
        v23 = pk_fma(v23, a23, b23);
        if (p.act) {
          v01 = f32x2{silu_f(v01.x), silu_f(v01.y)};
          v23 = f32x2{silu_f(v23.x), silu_f(v23.y)};
        }
      }
      const float keep = ((hmask >> i) & 1u) ? 1.f : 0.f;  // zero padding AFTER the activation
      const f32x2 k2 = f32x2{keep, keep};
      v01 = pk_mul(v01, k2);
      v23 = pk_mul(v23, k2);
      uint32_t h0, m0, l0, h1, m1, l1;
      split3_pair(v01.x, v01.y, h0, m0, l0);
      split3_pair(v23.x, v23.y, h1, m1, l1);
      if (hv < HALO_VOX) {
        // channel quad q -> 16-byte slot q>>1 (8 channels), half q&1; slot XOR-swizzled by the row
        uint32_t* row = s_halo + hv * RW + (((q >> 1) ^ ((hv >> 2) & 3)) << 2) + ((q & 1) << 1);
        *reinterpret_cast<uint2*>(row) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(row + PLANE) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(row + 2 * PLANE) = make_uint2(l0, l1);
      }
    }
  };

  f32x4 acc[MT][NS];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][s2][r] = 0.f;

  // A rows: MFMA tile t = x 0..7 of rows y = t and t+4 (see conv_halo_kernel); halo row index of the lane's voxel
  int a_row[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) a_row[t] = ((wz * HY) + 2 * wv + t + 4 * (lj >> 3)) * HX + (lj & 7);
  const int wncc = p.CinP / BK, wnsl = p.CoutP >> 4;
  const int64_t wplane = (int64_t)p.ksz * p.ksz * p.ksz * wncc * wnsl * 256;  // words per weight copy
  const float* w_lane = reinterpret_cast<const float*>(p.w_bf) + (int64_t)((n0 >> 4) + 2 * wc) * 256 + lane * 4;

  auto load_a = [&](float4 (&a)[MT], int plane, int tap) {  // one bf16 plane (0 hi, 1 mid, 2 lo) of a tap's A operands
    const int kd = tap / 9, kh = (tap - kd * 9) / 3, kw = tap - kd * 9 - kh * 3;
    const int trow = (kd * HY + kh) * HX + kw;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int row = a_row[t] + trow;
      a[t] = *reinterpret_cast<const float4*>(s_halo + plane * PLANE + row * RW + ((kq ^ ((row >> 2) & 3)) << 2));
    }
  };
  auto load_b = [&](float4 (&b)[3][NS], int cc, int tap) {  // [plane][slice]; the two slices are adjacent 1 KB blocks
    const float* wp = w_lane + (int64_t)(tap * wncc + cc) * wnsl * 256;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) b[pl][s2] = *reinterpret_cast<const float4*>(wp + pl * wplane + s2 * 256);
  };
  auto mfma4 = [&](const float4 (&a)[MT], const float4 (&b)[NS]) {  // four independent accumulators
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) acc[t][s2] = mfma_bf16_16x16x32(a[t], b[s2], acc[t][s2]);
  };

  // Operand pipeline.  A tap is only 24 MFMAs (~400 cycles), shorter than an L2 round trip, so the weights run
  // through a 4-deep register ring (requested THREE taps ahead).  The A planes of the next tap are requested inside
  // the current tap as soon as their registers die (lo after the first product group, mid after the fourth), so
  // only the hi plane is double buffered: 64 instead of 96 VGPRs of A operands.
  // Product order (smallest terms first): lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi.
  float4 aHi[2][MT] = {}, aMid[MT] = {}, aLo[MT] = {};
  float4 bRing[4][3][NS] = {};

  halo_issue(cc_begin);
  for (int cc = cc_begin; cc < cc_end; ++cc) {
    halo_commit();
    load_b(bRing[0], cc, 0);
    load_b(bRing[1], cc, 1);
    load_b(bRing[2], cc, 2);
    __syncthreads();
    load_a(aHi[0], 0, 0);
    load_a(aMid, 1, 0);
    load_a(aLo, 2, 0);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const float4(&b)[3][NS] = bRing[tap & 3];
      const float4(&hi)[MT] = aHi[tap & 1];
      if (tap + 1 < 27) load_a(aHi[(tap + 1) & 1], 0, tap + 1);
      if (tap + 3 < 27) load_b(bRing[(tap + 3) & 3], cc, tap + 3);
      if (tap == 26 && cc + 1 < cc_end) halo_issue(cc + 1);  // next chunk's halo flies under the last tap
      __builtin_amdgcn_sched_barrier(0);
      mfma4(aLo, b[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (tap + 1 < 27) load_a(aLo, 2, tap + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma4(hi, b[2]);
      mfma4(aMid, b[1]);
      mfma4(aMid, b[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (tap + 1 < 27) load_a(aMid, 1, tap + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma4(hi, b[1]);
      mfma4(hi, b[0]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // ---- epilogue: per (tile t, slice s2): D col = lane&15 (Cout), row = 4*(lane>>4) + r (voxel of the tile)
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;
  const int64_t tbase = ((((int64_t)n * p.OD + tz0 + wz) * p.OH + ty0) * p.OW + tx0) * p.Cout;
  int off[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) off[t] = ((2 * wv + t + 4 * (kq >> 1)) * p.OW + 4 * (kq & 1)) * p.Cout;
#pragma unroll
  for (int s2 = 0; s2 < NS; ++s2) {
    const int co = n0 + (2 * wc + s2) * 16 + lj;
    const int coc = co < p.Cout ? co : p.Cout - 1;
    const float bv = (p.nsplit == 1 && p.bias) ? p.bias[coc] : 0.f;
    float ssum = 0.f, ssq = 0.f;
    if (p.nsplit == 1) {
      if (p.residual) {  // one batch of loads under a uniform branch (see conv_halo_kernel)
        const float* rp = p.residual + tbase + coc;
        float res[MT][4];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) res[t][r] = rp[off[t] + r * p.Cout];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][s2][r] += res[t][r];
      }
      float* op = p.out + tbase + coc;
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[t][s2][r] + bv;
          if (co < p.Cout) op[off[t] + r * p.Cout] = v;
          ssum += v;
          ssq += v * v;
        }
    } else if (co < p.Cout) {
      float* pp = p.partial + (int64_t)blockIdx.z * M * p.Cout + tbase + co;
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) pp[off[t] + r * p.Cout] = acc[t][s2][r];
    }
    // GroupNorm statistics of the output: one slab per (workgroup, z-slab, voxel half) -> stats[n][slab][Cout][2]
    if (p.stats && p.nsplit == 1) {
      ssum += __shfl_xor(ssum, 16);
      ssq += __shfl_xor(ssq, 16);
      ssum += __shfl_xor(ssum, 32);
      ssq += __shfl_xor(ssq, 32);
      if (kq == 0 && co < p.Cout) {
        const int tiles_per_sample = ntx * nty * ntz;
        const int slab = ((int)(blockIdx.x % tiles_per_sample) * TZ + wz) * 2 + wv;
        double* d = p.stats + (((int64_t)n * tiles_per_sample * TZ * 2 + slab) * p.Cout + co) * 2;
        d[0] = (double)ssum;
        d[1] = (double)ssq;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row-tile kernel for the latency-bound launches: the deepest UNet levels (4^3 / 2^3 voxels: M = 64 rows, K up
// to 27*1024, pure weight streaming: 28-56 MB of weights for ~1 GFLOP) and every 1x1x1 convolution (attention
// qkv / proj_out, un-fused skip connections).  The design goal is round trips, not MFMA rate:
//   * block tile 64 voxels x 64 Cout; each wave OWNS 16 output channels (v_mfma_f32_16x16x4_f32 over the four
//     16-voxel tiles), so its weights go global -> registers with no LDS and no sharing;
//   * K is walked in groups of SG = 8 chunks (256 channels of one tap).  For a group, ALL weights (16 x 16 B per
//     lane) and ALL activation rows (gathered, GroupNorm/FiLM/SiLU applied, zero padded) are requested up
//     front, the activations land in an SG-deep LDS tile, and the 8 x 32 MFMAs of the group then run with no
//     global access and no barrier: one memory round trip per 256 K-channels instead of one per 32;
//   * split-K over (tap, chunk) fills the chip with ~2 workgroups per CU; un-split launches produce the
//     GroupNorm statistics of their output in the epilogue.
// ---------------------------------------------------------------------------------------------
constexpr int SM_ROWS = 64;
constexpr int SG = 8;
constexpr int SGH = 4;  // activation chunks requested per staging batch (register budget)

// BF (bf16 compute mode): the activation rows are rounded to bf16 when they are committed to LDS (80-byte rows), the
// weights are the bf16 plane packed for v_mfma_f32_16x16x32_bf16 (one 1 KB block per wave, tap and chunk), and one
// MFMA per 16-voxel tile covers a chunk's 32 channels (eight fp32 ones otherwise).
template <bool BF>
__global__ __launch_bounds__(256, 2) void conv_small_kernel(ConvParams p) {  // (bf16: 40 KB of LDS, three workgroups per CU)
  constexpr int RW = BF ? 20 : LDK;  // LDS words per activation row
  __shared__ __attribute__((aligned(16))) float s_a[SG * SM_ROWS * RW];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lj = lane & 15;
  const int kq = lane >> 4;
  const int Cin = p.C0 + p.C1;
  const int ncc = (Cin + BK - 1) / BK;
  const int ntaps = p.ksz * p.ksz * p.ksz;
  // a ResBlock's 1x1x1 skip_connection rides along as extra K chunks BEHIND the (tap, chunk) list (the deepest levels, where
  // it used to be a launch + a reduce of its own): raw block input at the output voxel, no GroupNorm / SiLU, its own weights
  const int SCin = p.skip_w ? p.skip_C0 + p.skip_C1 : 0;
  const int nmain = ntaps * ncc;
  const int nchunks = nmain + (SCin + BK - 1) / BK;
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;
  const int64_t m0 = (int64_t)blockIdx.x * SM_ROWS;
  const int n0 = blockIdx.y * 64;
  const int kc_begin = blockIdx.z * p.chunks_per_split;
  int kc_end = kc_begin + p.chunks_per_split;
  if (kc_end > nchunks) kc_end = nchunks;

  const int q = tid & 7;
  const int r0 = tid >> 3;  // rows r0 and r0 + 32
  int an[2], az[2], ay[2], ax[2];
  bool av[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int64_t m64 = m0 + r0 + 32 * j;
    av[j] = m64 < M;
    // (32-bit: M < 2^31, conv_small_launch; a 64-bit division is ~150 instructions and this kernel is all fixed cost)
    const unsigned m = av[j] ? (unsigned)m64 : 0u;
    int ow = (int)(m % (unsigned)p.OW);
    unsigned t = m / (unsigned)p.OW;
    int oh = (int)(t % (unsigned)p.OH);
    t /= (unsigned)p.OH;
    int od = (int)(t % (unsigned)p.OD);
    an[j] = (int)(t / (unsigned)p.OD);
    az[j] = od * p.stride - p.pad;
    ay[j] = oh * p.stride - p.pad;
    ax[j] = ow * p.stride - p.pad;
  }
  const int SD = p.ups ? (p.ID >> 1) : p.ID;
  const int SH = p.ups ? (p.IH >> 1) : p.IH;
  const int SW = p.ups ? (p.IW >> 1) : p.IW;

  float4 ra[SGH][2], rc01[SGH][2], rc23[SGH][2];
  unsigned amask[SGH];
  int g_tap[SG], g_cc[SG];  // (tap, channel chunk) of the group's chunks (wave-uniform; one division per group, not per chunk)
  auto load_a = [&](int i, int slot) {
    const int tap = g_tap[slot];
    const int cc = g_cc[slot];
    int kd = 0, kh = 0, kw = 0;
    if (p.ksz == 3) {
      kd = tap / 9;
      kh = (tap - kd * 9) / 3;
      kw = tap - kd * 9 - kh * 3;
    }
    int c = cc * BK + q * 4;
    const bool cvalid = c < Cin;
    if (!cvalid) c = 0;
    const float* src = p.src0;
    int Cs = p.C0, cs = c;
    if (c >= p.C0) {
      src = p.src1;
      Cs = p.C1;
      cs = c - p.C0;
    }
    amask[i] = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int z = az[j] + kd, y = ay[j] + kh, x = ax[j] + kw;
      const bool ok = av[j] && cvalid && z >= 0 && z < p.ID && y >= 0 && y < p.IH && x >= 0 && x < p.IW;
      z = min(max(z, 0), p.ID - 1);
      y = min(max(y, 0), p.IH - 1);
      x = min(max(x, 0), p.IW - 1);
      if (p.ups) {
        z >>= 1;
        y >>= 1;
        x >>= 1;
      }
      // unconditional load from a clamped address, masked afterwards (see the halo kernel)
      ra[i][j] = ld_act4(src, ((((int64_t)an[j] * SD + z) * SH + y) * SW + x) * Cs + cs, p.in_bf16);
      amask[i] |= (ok ? 1u : 0u) << j;
      if (p.coef) {
        const float4* cf = reinterpret_cast<const float4*>(p.coef + ((int64_t)an[j] * Cin + c) * 2);
        rc01[i][j] = cf[0];
        rc23[i][j] = cf[1];
      }
    }
  };
  auto store_a = [&](int i, int slot, bool plain = false) {  // plain: a chunk of the fused skip (no GroupNorm / SiLU)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float4 v = ra[i][j];
      if (p.coef && !plain) {
        v.x = v.x * rc01[i][j].x + rc01[i][j].y;
        v.y = v.y * rc01[i][j].z + rc01[i][j].w;
        v.z = v.z * rc23[i][j].x + rc23[i][j].y;
        v.w = v.w * rc23[i][j].z + rc23[i][j].w;
        if (p.act) {
          v.x = silu_f(v.x);
          v.y = silu_f(v.y);
          v.z = silu_f(v.z);
          v.w = silu_f(v.w);
        }
      }
      const float keep = ((amask[i] >> j) & 1u) ? 1.f : 0.f;  // zero padding AFTER the activation
      v.x *= keep;
      v.y *= keep;
      v.z *= keep;
      v.w *= keep;
      if (BF)
        *reinterpret_cast<uint2*>(s_a + slot * (SM_ROWS * RW) + (r0 + 32 * j) * RW + q * 2) =
            make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
      else
        *reinterpret_cast<float4*>(s_a + slot * (SM_ROWS * RW) + (r0 + 32 * j) * RW + q * 4) = v;
    }
  };

  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;

  const int wncc = p.CinP / BK, wnsl = p.CoutP >> 4;
  constexpr int WBLK = BF ? 256 : 512;  // words per (tap, chunk, 16-Cout slice) block
  const float* w_lane = (BF ? reinterpret_cast<const float*>(p.w_bf) : p.w) + (int64_t)((n0 >> 4) + wave) * WBLK + lane * 4;  // 1 KB contiguous per wave instruction

  // The main (tap, chunk) list and the skip's chunks behind it are walked by two loops over the same staging tile: the main
  // loop is exactly the kernel without a skip (a variant with both kinds in one load path lost 1.5 - 3 us on EVERY launch).
  const int kc_main_end = kc_end < nmain ? kc_end : nmain;
  // (tap, chunk) of the SG chunks that start at chunk g0
  auto group_chunks = [&](int g0) {
    int tap = p.ksz == 1 ? 0 : g0 / ncc, cc = g0 - tap * ncc;
#pragma unroll
    for (int i = 0; i < SG; ++i) {
      g_tap[i] = tap, g_cc[i] = cc;
      if (g0 + i + 1 < kc_main_end) {  // (chunks beyond the split's last one repeat it: loaded, never used)
        ++cc;
        if (cc == ncc) cc = 0, ++tap;
      }
    }
  };
  // the MFMAs of one staged group: chunks [g, min(g + SG, kend))
  auto mfma_group = [&](int g, int kend, const float4 (&bw)[SG][2]) {
#pragma unroll
    for (int i = 0; i < SG; ++i) {
      if (g + i < kend) {  // uniform
        const float* ab = s_a + i * (SM_ROWS * RW) + lj * RW + kq * (BF ? 4 : 8);
        float4 a0[4], a1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          a0[t] = *reinterpret_cast<const float4*>(ab + t * 16 * RW);
          if (!BF) a1[t] = *reinterpret_cast<const float4*>(ab + t * 16 * RW + 4);
        }
        if (BF) {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16_16x16x32(a0[t], bw[i][0], acc[t]);
          continue;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t].x, bw[i][0].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t].y, bw[i][0].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t].z, bw[i][0].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t].w, bw[i][0].w, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t].x, bw[i][1].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t].y, bw[i][1].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t].z, bw[i][1].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t].w, bw[i][1].w, acc[t], 0, 0, 0);
      }
    }
  };
  for (int g = kc_begin; g < kc_main_end; g += SG) {
    if (g != kc_begin) __syncthreads();  // previous group's activation tile fully consumed
    group_chunks(g);
    // 1. the (small, L2-resident) activation rows of the whole group, in batches of SGH chunks (register budget).
    //    (Requesting the next group's first batch under this group's MFMAs was tried: 96 more live registers, no gain - the
    //    second resident workgroup already fills the gap.)
    // 2. every weight of the group is requested at once, right BEHIND the requests of the last activation batch and before
    //    that batch is waited for: memory returns a wave's loads in order, so the activations are not held behind 28+ MB
    //    of weights, the wait for them overlaps the weights' round trip (a 1x1x1 convolution is then ONE round trip plus
    //    its MFMAs), and the chunk loop below starts on chunk 0 as soon as ITS weights are back while the rest streams.
    float4 bw[SG][2];
    auto load_w = [&]() {
#pragma unroll
      for (int i = 0; i < SG; ++i) {
        const float* wp = w_lane + (int64_t)(g_tap[i] * wncc + g_cc[i]) * wnsl * WBLK;
        bw[i][0] = *reinterpret_cast<const float4*>(wp);
        if (!BF) bw[i][1] = *reinterpret_cast<const float4*>(wp + 256);
      }
    };
#pragma unroll
    for (int h = 0; h < SG; h += SGH) {
      if (g + h < kc_main_end) {  // uniform
#pragma unroll
        for (int i = 0; i < SGH; ++i) load_a(i, h + i);
        if (!BF && (h + SGH >= SG || g + h + SGH >= kc_main_end)) load_w();  // (uniform) the group's last batch
#pragma unroll
        for (int i = 0; i < SGH; ++i) store_a(i, h + i);
      }
    }
    // (bf16 mode: its launches are the large-M ones of the 128^3 net, throughput bound: the weights are requested behind the
    //  stores, which keeps the kernel at 3 - 4 workgroups per CU instead of 2)
    if (BF) load_w();
    __syncthreads();
    mfma_group(g, kc_main_end, bw);
  }
  // ---- the fused skip's chunks (p.skip_w; launches without one never enter): raw block input at the row's OWN voxel (stride 1,
  //      same resolution: conv_launch), zero for rows / channels beyond the ends, its own packed weights
  if (!BF && kc_end > nmain) {  // (fp32 mode only: unet_exec.cpp fuses the skip below 8^3 there)
    const float* skw_lane = (BF ? reinterpret_cast<const float*>(p.skip_w_bf) : p.skip_w) + (int64_t)((n0 >> 4) + wave) * WBLK + lane * 4;
    const int sk_lo = (kc_begin > nmain ? kc_begin : nmain) - nmain, sk_hi = kc_end - nmain;  // skip chunks [sk_lo, sk_hi)
    for (int g = sk_lo; g < sk_hi; g += SG) {
      if (g != sk_lo || kc_begin < nmain) __syncthreads();
      float4 bw[SG][2];
#pragma unroll
      for (int h = 0; h < SG; h += SGH) {
        if (g + h < sk_hi) {  // uniform
#pragma unroll
          for (int i = 0; i < SGH; ++i) {
            const int sc = g + h + i < sk_hi ? g + h + i : sk_hi - 1;
            int c = sc * BK + q * 4;
            const bool cvalid = c < SCin;
            if (!cvalid) c = 0;
            const float* src = p.skip_src0;
            int Cs = p.skip_C0, cs = c;
            if (c >= p.skip_C0) {
              src = p.skip_src1;
              Cs = p.skip_C1;
              cs = c - p.skip_C0;
            }
            amask[i] = 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int z = az[j] + p.pad, y = ay[j] + p.pad, x = ax[j] + p.pad;
              ra[i][j] = ld_act4(src, ((((int64_t)an[j] * p.OD + z) * p.OH + y) * p.OW + x) * Cs + cs, p.in_bf16);
              amask[i] |= (av[j] && cvalid ? 1u : 0u) << j;
            }
          }
          if (h + SGH >= SG || g + h + SGH >= sk_hi) {
#pragma unroll
            for (int i = 0; i < SG; ++i) {
              const int sc = g + i < sk_hi ? g + i : sk_hi - 1;
              const float* wp = skw_lane + (int64_t)sc * wnsl * WBLK;
              bw[i][0] = *reinterpret_cast<const float4*>(wp);
              if (!BF) bw[i][1] = *reinterpret_cast<const float4*>(wp + 256);
            }
          }
#pragma unroll
          for (int i = 0; i < SGH; ++i) store_a(i, h + i, true);
        }
      }
      __syncthreads();
      mfma_group(g, sk_hi, bw);
    }
  }

  // ---- epilogue: col = lane&15 (Cout), row = 4*(lane>>4) + r inside each 16-voxel tile
  const int co = n0 + wave * 16 + lj;
  const int coc = co < p.Cout ? co : p.Cout - 1;
  float bv = (p.nsplit == 1 && p.bias) ? p.bias[coc] : 0.f;
  if (p.nsplit == 1 && p.skip_w && p.skip_bias) bv += p.skip_bias[coc];
  float ssum = 0.f, ssq = 0.f;
  int64_t mo[4][4];  // clamped row offsets (masked at the store): residual loads are issued as ONE batch
  bool mv[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = m0 + t * 16 + 4 * kq + r;
      mv[t][r] = m < M && co < p.Cout;
      mo[t][r] = (m < M ? m : M - 1) * p.Cout;
    }
  if (p.nsplit == 1) {
    if (p.residual) {
      float res[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) res[t][r] = ld_act1(p.residual, mo[t][r] + coc, p.res_bf16);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] += res[t][r];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[t][r] + bv;
        if (mv[t][r]) {
          st_act1(p.out, mo[t][r] + co, v, p.out_bf16);
          ssum += v;
          ssq += v * v;
        }
      }
  } else {
    float* pp = p.partial + (int64_t)blockIdx.z * M * p.Cout + coc;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (mv[t][r]) pp[mo[t][r]] = acc[t][r];
  }
  // GroupNorm statistics of the output (un-split launches whose row tiles do not straddle samples:
  // conv_stats_slabs): one slab per row tile -> stats[n][slab][Cout][2]
  if (p.stats && p.nsplit == 1) {
    ssum += __shfl_xor(ssum, 16);
    ssq += __shfl_xor(ssq, 16);
    ssum += __shfl_xor(ssum, 32);
    ssq += __shfl_xor(ssq, 32);
    if (kq == 0 && co < p.Cout) {
      const int tiles_per_sample = (int)(((int64_t)p.OD * p.OH * p.OW) / SM_ROWS);
      const int n = (int)(blockIdx.x / tiles_per_sample);
      const int slab = (int)(blockIdx.x % tiles_per_sample);
      double* d = p.stats + (((int64_t)n * tiles_per_sample + slab) * p.Cout + co) * 2;
      d[0] = (double)ssum;
      d[1] = (double)ssq;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Streaming 1x1x1 convolution of a LARGE grid: a ResBlock's skip_connection (unet.py:222) on the 64^3 level as a launch
// of its own, its output the residual of the block's second 3x3x3 convolution.  (Fused into that convolution as extra
// pseudo-taps the 128-channel skip costs conv_wino3_kernel ~100 us per launch - its operands are scattered 16-byte reads -;
// here it is a plain GEMM of M = 262 144 rows, K <= 256, N = 64 that runs at the rate its 200 MB cross the fabric.)
// A wave owns ALL 64 output channels of a 16-row tile: the K x 64 weight block (the row-tile kernel's packed layout) sits in
// LDS for the whole launch (32 KB at K = 128; it was first held in registers: 248 of them, which left room for ONE tile of
// rows in flight per wave and 3.1 TB/s), the rows come straight from global memory - lane (lj, kq) reads the 32 contiguous
// bytes (channels 8 kq .. +7 of row lj) of every 32-channel chunk, FOUR tiles ahead of the MFMAs that consume them -, one
// barrier at the start.  Raw input (no GroupNorm / activation), virtual concat of two sources.
// ---------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256, 2) void conv1x1_stream_kernel(ConvParams p) {
  constexpr int DEPTH = NCH <= 4 ? 4 : 2;  // row tiles in flight per wave (the ring of A registers: DEPTH x NCH x 8)
  // the K x 64 weight block of the workgroup, in the row-tile kernel's packed order: [chunk][16-Cout slice][half][lane][4]
  __shared__ __attribute__((aligned(16))) float s_w[NCH * 4 * 2 * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lj = lane & 15, kq = lane >> 4;
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;
  const int64_t ntile = M >> 4;
  const int n0 = blockIdx.y * 64;
  const int wnsl = p.CoutP >> 4;
  for (int i = tid; i < NCH * 4 * 2 * 64; i += 256) {  // 16-byte pieces: (chunk, slice, half, lane)
    const int ln = i & 63, h = (i >> 6) & 1, sl = (i >> 7) & 3, ch = i >> 9;
    *reinterpret_cast<float4*>(s_w + i * 4) =
        *reinterpret_cast<const float4*>(p.w + ((int64_t)ch * wnsl + (n0 >> 4) + sl) * 512 + h * 256 + ln * 4);
  }
  // The product is formed TRANSPOSED (A operand = the weights, rows = output channels; B operand = the tile's rows, columns
  // = voxels): a lane then holds 4 CONSECUTIVE output channels (D rows 4 kq + r) of voxel lj, so the tile leaves as four
  // 16-byte stores per lane instead of sixteen 4-byte ones.
  float4 bv[4];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl)
    bv[sl] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + sl * 16 + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
  // per chunk: source, row stride and channel offset of the lane's 8 channels (C0 is a multiple of 32 with two sources)
  const float* csrc[NCH];
  int cstr[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = i * 32 + kq * 8;
    const bool second = p.src1 != nullptr && c >= p.C0;
    csrc[i] = (second ? p.src1 + (c - p.C0) : p.src0 + c);
    cstr[i] = second ? p.C1 : p.C0;
  }
  auto load_rows = [&](int64_t t, float4 (&a)[NCH][2]) {
    const int64_t m = t * 16 + lj;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const float* r = csrc[i] + m * cstr[i];
      a[i][0] = *reinterpret_cast<const float4*>(r);
      a[i][1] = *reinterpret_cast<const float4*>(r + 4);
    }
  };
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
  float4 A[DEPTH][NCH][2];
#pragma unroll
  for (int j = 0; j < DEPTH - 1; ++j)
    if (gw + j * nw < ntile) load_rows(gw + j * nw, A[j]);
  __syncthreads();  // the weights are in LDS (the only barrier of the kernel)
  for (int64_t t0 = gw; t0 < ntile; t0 += DEPTH * nw) {
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
      const int64_t t = t0 + j * nw;
      if (t >= ntile) break;  // (uniform)
      if (t + (DEPTH - 1) * nw < ntile) load_rows(t + (DEPTH - 1) * nw, A[(j + DEPTH - 1) % DEPTH]);
      int lw = lane * 4;
      HOLO_LAUNDER(lw);  // (the weight reads are loop invariant: hoisted, they would take NCH x 32 registers again)
      const float* wl = s_w + lw;
      f32x4 acc[4];
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[sl][r] = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float4 bw[4];
#pragma unroll
          for (int sl = 0; sl < 4; ++sl) bw[sl] = *reinterpret_cast<const float4*>(wl + ((i * 4 + sl) * 2 + h) * 256);
          const float4 a = A[j][i][h];
          // the four accumulators advance together, k-step by k-step (no MFMA waits on its predecessor's result)
#pragma unroll
          for (int sl = 0; sl < 4; ++sl) acc[sl] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[sl].x, a.x, acc[sl], 0, 0, 0);
#pragma unroll
          for (int sl = 0; sl < 4; ++sl) acc[sl] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[sl].y, a.y, acc[sl], 0, 0, 0);
#pragma unroll
          for (int sl = 0; sl < 4; ++sl) acc[sl] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[sl].z, a.z, acc[sl], 0, 0, 0);
#pragma unroll
          for (int sl = 0; sl < 4; ++sl) acc[sl] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[sl].w, a.w, acc[sl], 0, 0, 0);
        }
      // D: column lj = voxel of the tile, rows 4 kq + r = output channels of the slice
      float* o = p.out + (t * 16 + lj) * p.Cout + n0 + 4 * kq;
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
        *reinterpret_cast<float4*>(o + sl * 16) =
            make_float4(acc[sl][0] + bv[sl].x, acc[sl][1] + bv[sl].y, acc[sl][2] + bv[sl].z, acc[sl][3] + bv[sl].w);
    }
  }
}

// s = b + sum over the splits of partial[k][i .. i+3], in the fixed order both reduce kernels share
__device__ __forceinline__ float4 splitk_sum4(const float* __restrict__ partial, int nsplit, int64_t MC, int64_t i, float4 b) {
  float4 s = b;
  int k = 0;
  for (; k + 16 <= nsplit; k += 16) {  // deep splits (row-tile kernel): sixteen independent loads in flight
    float4 t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = *reinterpret_cast<const float4*>(partial + (int64_t)(k + u) * MC + i);
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      s.x += (t[u].x + t[u + 1].x) + (t[u + 2].x + t[u + 3].x);
      s.y += (t[u].y + t[u + 1].y) + (t[u + 2].y + t[u + 3].y);
      s.z += (t[u].z + t[u + 1].z) + (t[u + 2].z + t[u + 3].z);
      s.w += (t[u].w + t[u + 1].w) + (t[u + 2].w + t[u + 3].w);
    }
  }
  for (; k + 4 <= nsplit; k += 4) {  // four independent loads in flight per thread
    const float4 t0 = *reinterpret_cast<const float4*>(partial + (int64_t)k * MC + i);
    const float4 t1 = *reinterpret_cast<const float4*>(partial + (int64_t)(k + 1) * MC + i);
    const float4 t2 = *reinterpret_cast<const float4*>(partial + (int64_t)(k + 2) * MC + i);
    const float4 t3 = *reinterpret_cast<const float4*>(partial + (int64_t)(k + 3) * MC + i);
    s.x += (t0.x + t1.x) + (t2.x + t3.x);
    s.y += (t0.y + t1.y) + (t2.y + t3.y);
    s.z += (t0.z + t1.z) + (t2.z + t3.z);
    s.w += (t0.w + t1.w) + (t2.w + t3.w);
  }
  for (; k < nsplit; ++k) {
    const float4 t = *reinterpret_cast<const float4*>(partial + (int64_t)k * MC + i);
    s.x += t.x;
    s.y += t.y;
    s.z += t.z;
    s.w += t.w;
  }
  return s;
}

// out = sum_s partial[s] + bias + residual, fused with the GroupNorm statistics of `out`.
// Thread layout of gn_stats_kernel: cq = Cout/4 threads across channels (float4), rows = 256/cq voxels per
// pass, one workgroup per voxel slab of one sample; grid = (B, N) with B, vox_per_block from
// gn_stats_geometry(Cout, V).  stats (optional): [n][B][Cout][2] doubles.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int nsplit,
                                                            int64_t MC, int Cout, int64_t V, int vox_per_block,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ bias2,
                                                            const float* __restrict__ residual,
                                                            float* __restrict__ out, double* __restrict__ stats,
                                                            int res_bf16, int out_bf16) {
  __shared__ double red[256 * 8];
  const int n = blockIdx.y;
  const int c_base = blockIdx.z * 1024;  // column blocks of <= 1024 channels (qkv convs are up to 1536 wide)
  const int Cb = (Cout - c_base) < 1024 ? (Cout - c_base) : 1024;
  const int cq = Cb >> 2;
  const int rows = 256 / cq;
  const int tid = threadIdx.x;
  const int c4 = tid % cq;
  const int vr = tid / cq;
  const int64_t vbeg = (int64_t)blockIdx.x * vox_per_block;
  int64_t vend = vbeg + vox_per_block;
  if (vend > V) vend = V;
  float fs[4] = {0, 0, 0, 0}, fq[4] = {0, 0, 0, 0};
  if (vr < rows) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b = *reinterpret_cast<const float4*>(bias + c_base + c4 * 4);
    if (bias2) {
      const float4 b2 = *reinterpret_cast<const float4*>(bias2 + c_base + c4 * 4);
      b.x += b2.x;
      b.y += b2.y;
      b.z += b2.z;
      b.w += b2.w;
    }
    for (int64_t v = vbeg + vr; v < vend; v += rows) {
      const int64_t i = ((int64_t)n * V + v) * Cout + c_base + c4 * 4;
      float4 s = splitk_sum4(partial, nsplit, MC, i, b);
      if (residual) {
        const float4 r = ld_act4(residual, i, res_bf16);
        s.x += r.x;
        s.y += r.y;
        s.z += r.z;
        s.w += r.w;
      }
      st_act4(out, i, s, out_bf16);
      fs[0] += s.x;
      fs[1] += s.y;
      fs[2] += s.z;
      fs[3] += s.w;
      fq[0] += s.x * s.x;
      fq[1] += s.y * s.y;
      fq[2] += s.z * s.z;
      fq[3] += s.w * s.w;
    }
  }
  if (!stats) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[tid * 8 + e] = (double)fs[e];
    red[tid * 8 + 4 + e] = (double)fq[e];
  }
  __syncthreads();
  if (tid < cq) {
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < rows; ++r)
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += red[(r * cq + tid) * 8 + e];
    double* dst = stats + (((int64_t)n * gridDim.x + blockIdx.x) * Cout + c_base + tid * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dst[e * 2 + 0] = s[e];
      dst[e * 2 + 1] = s[4 + e];
    }
  }
}

}  // namespace

size_t conv_plan(ConvParams& p, int num_cus) {
  const int Cin = p.C0 + p.C1;
  const int ncc = (Cin + BK - 1) / BK;
  const int nchunks = p.ksz * p.ksz * p.ksz * ncc;
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;
  const int bn = p.Cout >= 64 ? 64 : 32;
  const int64_t tiles = cdiv(M, BM) * cdiv(p.Cout, bn);
  int nsplit = 1;
  const int64_t target = 2 * (int64_t)num_cus;
  const int64_t src_vox = (int64_t)p.ID * p.IH * p.IW;  // the halo kernel addresses a source sample with 32-bit byte offsets
  const int cmax = p.C0 > p.C1 ? p.C0 : p.C1;
  const int skmax = p.skip_C0 > p.skip_C1 ? p.skip_C0 : p.skip_C1;
  const bool fits32 = src_vox * (cmax > skmax ? cmax : skmax) * 4 < ((int64_t)1 << 32) &&
                      (int64_t)p.OD * p.OH * p.OW * p.Cout * 4 < ((int64_t)1 << 32);  // (the wide-tile epilogue: output sample too)
  p.mode = (p.ksz == 3 && p.stride == 1 && p.pad == 1 && (p.OD % 2) == 0 && (p.OH % 8) == 0 && (p.OW % 8) == 0 &&  // (TZ=1 tiles need no z divisibility)
            p.ID == p.OD && p.IH == p.OH && p.IW == p.OW && ((p.C0 + p.C1) % 16) == 0 && fits32)
               ? 1
               : 0;
  // the qkv convolution of an AttentionBlock, fused with the operand packing of the bf16 attention (the planner offers it by
  // setting qkv_q; HOLO_CONV_QKV_FUSED=0 keeps the row-tile kernel + attn_pack_kernel)
  if (p.qkv_q) {
    const char* eq = getenv("HOLO_CONV_QKV_FUSED");
    if (!(eq && eq[0] == '0') && conv1x1_qkv_bf16_supported(p)) {
      p.mode = 5;
      p.nsplit = 1;
      p.chunks_per_split = ncc;
      conv1x1_qkv_bf16_plan(p, num_cus);
      if (getenv("HOLO_DEBUG_PLAN"))
        fprintf(stderr, "[plan] qkv conv %d->%d, T %d: fused with the attention's operand packing, %d rows x %d slices per workgroup\n", Cin,
                p.Cout, p.qkv_T, p.qkv_rows, p.qkv_sb);
      return 0;
    }
    p.qkv_q = nullptr;  // (not this launch: the caller packs)
  }
  // any other 1x1x1 convolution of a large grid on bf16 storage (the attention's proj_out): the same streaming GEMM with a plain
  // output (HOLO_CONV1X1_BF16_STREAM=0 keeps the row-tile kernel)
  {
    const char* e5 = getenv("HOLO_CONV1X1_BF16_STREAM");
    if (!(e5 && e5[0] == '0') && conv1x1_bf16_stream_supported(p)) {
      p.mode = 6;
      p.nsplit = 1;
      p.chunks_per_split = ncc;
      conv1x1_bf16_stream_plan(p, num_cus);
      if (getenv("HOLO_DEBUG_PLAN"))
        fprintf(stderr, "[plan] conv1 %d->%d @%d^3: bf16 streaming GEMM, %d rows x %d slices per workgroup\n", Cin, p.Cout, p.OD, p.qkv_rows,
                p.qkv_sb);
      return 0;
    }
  }
  // a 1x1x1 convolution of raw input over a LARGE grid (a ResBlock's skip_connection on the 64^3 level): the streaming GEMM
  // (HOLO_CONV1X1_STREAM_MIN_M=<rows>: development knob, default 131 072 rows; 0 = off)
  {
    const char* e1 = getenv("HOLO_CONV1X1_STREAM_MIN_M");
    const int64_t min_m = e1 ? atoll(e1) : 131072;
    if (p.mode == 0 && min_m > 0 && M >= min_m && p.ksz == 1 && p.stride == 1 && !p.ups && !p.coef && !p.residual && !p.skip_w &&
        p.bf16 == 0 && !p.in_bf16 && !p.out_bf16 && (p.Cout % 64) == 0 && (Cin % 32) == 0 && Cin >= 32 && Cin <= 256 &&
        (!p.src1 || (p.C0 % 32) == 0) && (M % 16) == 0 && p.ID == p.OD && p.IH == p.OH && p.IW == p.OW) {
      p.mode = 3;
      p.nsplit = 1;
      p.chunks_per_split = ncc;
      return 0;
    }
  }
  // the stride-2 convolution of a Downsample block in the bf16 storage mode, where its 128-voxel tiles give the chip at least
  // a workgroup per four CUs (128^3 net: 128^3 -> 64^3 540 -> ~100 us, and the two levels below; deeper the row-tile kernel's
  // split-K fills the chip better).  HOLO_CONV_S2T=0 keeps the row-tile kernel, =1 takes this one wherever it is defined (tests)
  {
    const char* es = getenv("HOLO_CONV_S2T");
    const int64_t wgs = (M / 128) * (p.Cout / 64);
    if (p.mode == 0 && !(es && es[0] == '0') && conv_s2_bf16_supported(p) && (wgs >= num_cus / 4 || (es && es[0] == '1'))) {
      p.mode = 4;
      p.nsplit = 1;
      p.chunks_per_split = Cin / 16;
      if (getenv("HOLO_DEBUG_PLAN"))
        fprintf(stderr, "[plan] conv %d->%d @%d^3 stride 2: bf16 halo kernel, %lld workgroups\n", Cin, p.Cout, p.OD, (long long)wgs);
      return 0;
    }
  }
  // 1x1x1, strided and deepest-level convs: row-tile kernel (also for the 32^3 stride-2 convolution with its 32 768 rows: the
  // per-tap gather kernel takes 118 us there, this one 95)
  if (p.mode == 0 && p.Cout >= 64) {
    p.mode = 2;
    const int64_t t2 = cdiv(M, SM_ROWS) * cdiv(p.Cout, 64);
    const char* st = getenv("HOLO_SMALL_SPLIT_TARGET");  // development knob: workgroups per CU the split-K aims at (default 2)
    const int64_t tgt = (st && atoi(st) > 0 ? atoi(st) : 2) * (int64_t)num_cus;
    nsplit = t2 < tgt ? (int)cdiv(tgt, t2) : 1;
    // (a fused 1x1x1 skip: its chunks follow the (tap, chunk) list; stride 1 and no upsampling there: conv_launch)
    const int nall = nchunks + (p.skip_w ? (int)cdiv(p.skip_C0 + p.skip_C1, BK) : 0);
    int max_split = nall / SG;  // a split below one full staging group only adds a reduce launch
    if (max_split < 1) max_split = 1;
    if (nsplit > max_split) nsplit = max_split;
    int cps = (int)cdiv(nall, nsplit);
    nsplit = (int)cdiv(nall, cps);
    p.nsplit = nsplit;
    p.chunks_per_split = cps;
    return nsplit > 1 ? (size_t)nsplit * M * p.Cout * sizeof(float) : 0;
  }
  p.bf16t = 0;
  p.bf16p = 0;
  if (p.mode == 1 && p.bf16 == 1 && p.in_bf16 && p.w_bft && (!p.skip_w || p.skip_w_bft) && (p.OD % 8) == 0 && (p.OH % 8) == 0 && (p.OW % 8) == 0 && (p.Cout % 32) == 0 &&
      (p.C0 % 8) == 0 && (p.skip_C0 % 8) == 0 && ((p.skip_C0 + p.skip_C1) % 8) == 0 && (p.Cout >= 64 || !p.skip_w)) {
    // bf16 wide-tile kernel (8^3 voxels x 64 | 32 output channels per workgroup): where it fills the chip without
    // split-K, and on the under-filled levels (split over 16-channel chunks) once the K extent is long enough to pay for
    // the partials - measured at 128^3: from 192 input channels (main + fused skip) it beats the 64/128-voxel halo
    // kernel (32^3 .. 8^3 levels together 1.96 -> 1.7 ms), below that it loses.  HOLO_CONV_BF16T=0 disables it, =1
    // forces it everywhere (tests)
    const char* e = getenv("HOLO_CONV_BF16T");
    const int64_t t8 = (M / 512) * cdiv(p.Cout, bn);
    const char* lk = getenv("HOLO_CONV_BF16T_LONGK");  // development knob: channel threshold of the rule below (0 = off)
    const int lk_min = lk ? atoi(lk) : 192;
    const bool long_k = lk_min > 0 && Cin + (p.skip_w ? p.skip_C0 + p.skip_C1 : 0) >= lk_min;
    if (!(e && e[0] == '0') && (t8 >= target || long_k || (e && e[0] == '1'))) {
      const int ncc16 = (Cin + 15) / 16;
      const int nsk16 = p.skip_w ? (p.skip_C0 + p.skip_C1 + 15) / 16 : 0;
      {
        // workgroups per CU the split-K aims at: two on the 32^3 level and above, ONE below it and for raw-input launches
        // (measured at 128^3, profiles/r06_bf16t_split_target.txt: the 16^3 / 8^3 launches 43.6 -> 39.0, 41.5 -> 36.4,
        // 41.2 -> 35.9 us with half the partial sums to write and reduce; the 32^3 launches the other way round, 59.8 vs 66.4;
        // the raw-input Upsample convolution at 32^3 un-split lands on the persistent form: 126.7 -> 91.5 us).
        // HOLO_BF16T_SPLIT_TARGET=<n>: development knob
        const char* st = getenv("HOLO_BF16T_SPLIT_TARGET");
        const int64_t tgt = st && atoi(st) > 0 ? atoi(st) * (int64_t)num_cus : ((p.OD >= 32 && p.coef) ? target : (int64_t)num_cus);
        if (t8 < tgt) {
          nsplit = (int)cdiv(tgt, t8);
          if (nsplit > ncc16) nsplit = ncc16;
        }
      }
      const int cps = (int)cdiv(ncc16, nsplit);
      nsplit = (int)cdiv(ncc16, cps);
      p.bf16t = 1;
      p.tz = 8;
      p.wino = 0;
      p.nsplit = nsplit;
      p.chunks_per_split = cps;
      p.skip_chunks_per_split = (int)cdiv(nsk16, nsplit);
      p.grid_x = (int)(M / 512);
      // the filled levels (every CU gets whole tiles without split-K): the persistent wave-specialised form
      // (kernels_conv_bf16p.hip) - by default only for launches that stage their input RAW (no GroupNorm / SiLU on load: the
      // input convolution, the Upsample convolutions): there its producer waves keep up with the consumers (tools/bf16p_probe:
      // 128^3 32 -> 64 282 vs 293 us); with the activation arithmetic they do not yet (540 vs 479 us) and conv_bf16t_kernel
      // stays.  HOLO_CONV_BF16P=0 keeps conv_bf16t_kernel everywhere, =2 takes the persistent form for activated input too,
      // =1 forces it onto every wide-tile launch, without split-K (tests: small grids)
      {
        const char* ep = getenv("HOLO_CONV_BF16P");
        const bool force = ep && ep[0] == '1';
        const bool any_input = force || (ep && ep[0] == '2');
        const int wgs = num_cus & ~7;
        if (!(ep && ep[0] == '0') && (any_input || !p.coef) && ((nsplit == 1 && wgs >= 8 && t8 >= wgs) || force) &&
            (p.C1 == 0 || (p.C0 % 16) == 0) &&  // (a 16-channel chunk / a 32-channel skip step lies in ONE source)
            (!p.skip_w || (((p.skip_C0 + p.skip_C1) % 32) == 0 && (p.skip_C1 == 0 || (p.skip_C0 % 32) == 0))) &&
            (int64_t)p.ID * p.IH * p.IW < ((int64_t)1 << 24)) {  // (24-bit voxel indices in the producers' address arithmetic)
          p.bf16p = 1;
          p.nsplit = nsplit = 1;
          p.chunks_per_split = ncc16;
          p.skip_chunks_per_split = nsk16;
          const int64_t wcap = wgs >= 8 ? wgs : 8;
          int64_t g = t8 < wcap ? ((t8 + 7) & ~(int64_t)7) : wcap;
          const char* eg = getenv("HOLO_CONV_BF16P_WGS");  // test knob: at most this many persistent workgroups
          if (eg && atoi(eg) > 0 && atoi(eg) < g) g = atoi(eg);
          p.grid_x = (int)(g < 8 && !eg ? 8 : g);
        }
      }
      if (getenv("HOLO_DEBUG_PLAN"))
        fprintf(stderr, "[plan] conv %d->%d @%d^3: bf16 wide-tile kernel%s, %d tiles x %d slices, split-K %d%s\n", Cin, p.Cout,
                p.OD, p.bf16p ? " (persistent, wave-specialised)" : "", (int)(M / 512), (int)cdiv(p.Cout, bn), nsplit,
                p.skip_w ? ", fused skip" : "");
      return nsplit > 1 ? (size_t)nsplit * M * p.Cout * sizeof(float) : 0;
    }
  }
  if (p.mode == 1) {  // halo kernel: split over 32-channel chunks (each split walks all 27 taps)
    p.tz = 2;
    int64_t htiles = tiles;
    const char* f2 = getenv("HOLO_CONV_FORCE_TZ2");  // test knob: 128-voxel tiles (hence the Winograd-in-depth kernel) on small grids
    const bool force_tz2 = f2 && f2[0] == '1';
    // exact-fp32 launches whose Winograd weights exist keep the 128-voxel tile on under-filled levels too: the Winograd
    // kernels do 4/9 (2/3) of the MFMAs per voxel, which buys more than the doubled workgroup count of 64-voxel tiles;
    // the chip is filled by split-K over the channel chunks instead
    const char* ws = getenv("HOLO_CONV_WINO_SMALL");
    const bool wino_small = !(ws && ws[0] == '0') && p.w_wino && p.bf16 == 0 && p.Cout >= 64 && (p.Cout % 64) == 0 &&
                            (!p.skip_w || p.skip_w_wino) && ncc >= 2;
    if (tiles < target && !force_tz2 && !wino_small) {  // under-filled chip: 64-voxel tiles double the workgroups before resorting to split-K
      p.tz = 1;
      htiles = (M / 64) * cdiv(p.Cout, bn);
    }
    const int nsk = p.skip_w ? (p.skip_C0 + p.skip_C1 + BK - 1) / BK : 0;  // fused skip chunks (one tap each)
    if (htiles < target) {
      nsplit = (int)cdiv(target, htiles);
      if (nsplit > ncc) nsplit = ncc;
    }
    int cps = (int)cdiv(ncc, nsplit);
    nsplit = (int)cdiv(ncc, cps);
    p.nsplit = nsplit;
    p.chunks_per_split = cps;
    p.skip_chunks_per_split = (int)cdiv(nsk, nsplit);
    // One workgroup per tile.  (The kernel can also walk several tiles per workgroup - grid_x < tiles - with the
    // next tile's halo prefetched under the last tap; measured on MI355X that is no faster than letting the
    // dispatcher refill the slots, which also balances the load dynamically: tools/conv_timeline.cpp.)
    p.grid_x = (int)(M / (64 * p.tz));
    // exact-fp32 128-voxel tiles: the Winograd-in-depth form (2/3 of the MFMAs) when its weights were prepared
    p.wino = (p.tz == 2 && p.w_wino && p.bf16 == 0 && p.Cout >= 64 && (p.Cout % 64) == 0 && (!p.skip_w || p.skip_w_wino)) ? 1 : 0;
    if (p.wino && p.w_wino2 && (!p.skip_w || p.skip_w_wino2)) p.wino = 2;  // both depth and height in Winograd form
    if (p.tz == 2 && p.w_wino2 && p.bf16 == 0 && p.Cout == 32 && !p.skip_w) p.wino = 2;  // 32-channel (z,y) form, two wave rows
    // F(2x2x2, 3x3x3) form (kernels_conv3.hip): ONE persistent 4-wave workgroup per CU walks (tile, 64-Cout block, split)
    // items, so the chip is full from num_cus items on (the (z,y) form needs 2 workgroups per CU); split-K only up to that.
    // HOLO_CONV_WINO3=0 disables it, HOLO_CONV_WINO3_MIN_ITEMS=<n> moves the threshold (default num_cus / 2)
    {
      const char* e3 = getenv("HOLO_CONV_WINO3");
      const char* m3 = getenv("HOLO_CONV_WINO3_MIN_ITEMS");
      const int64_t t3 = (M / 128) * (p.Cout / 64);
      const int64_t min_items = m3 ? atoi(m3) : num_cus / 2;
      if (!(e3 && e3[0] == '0') && p.tz == 2 && p.w_wino3 && p.bf16 == 0 && !p.in_bf16 && !p.out_bf16 && p.Cout >= 64 &&
          (p.Cout % 64) == 0 && (!p.skip_w || p.skip_w_wino3) && (p.OH % 8) == 0 && (p.OW % 8) == 0 && (p.OD % 2) == 0 &&
          (!p.coef || p.act) &&  // (its staging applies the affine and SiLU together)
          (int64_t)p.N * src_vox * (cmax > skmax ? cmax : skmax) * 4 < ((int64_t)1 << 32) &&  // (buffer addressing of whole tensors)
          (!p.skip_w || ((p.skip_C0 % 16) == 0 && (p.skip_C1 % 4) == 0))) {
        int ns3 = 1;
        if (t3 < num_cus) {
          ns3 = (int)cdiv(num_cus, t3);
          if (ns3 > ncc) ns3 = ncc;
        }
        const int cps3 = (int)cdiv(ncc, ns3);
        ns3 = (int)cdiv(ncc, cps3);
        if (t3 * ns3 >= min_items) {
          p.wino = 3;
          p.nsplit = ns3;
          p.chunks_per_split = cps3;
          p.skip_chunks_per_split = (int)cdiv(nsk, ns3);
          const int64_t items = t3 * ns3;
          // every workgroup gets the same number of items when the list allows it (the last round is then full)
          const int64_t rounds = cdiv(items, num_cus);
          p.grid_x = (int)cdiv(items, rounds);
          if (getenv("HOLO_DEBUG_PLAN"))
            fprintf(stderr, "[plan] conv %d->%d @%d^3: F(2x2x2) Winograd kernel, %lld items on %d workgroups, split-K %d%s\n", Cin,
                    p.Cout, p.OD, (long long)items, p.grid_x, ns3, p.skip_w ? ", fused skip" : "");
          return ns3 > 1 ? (size_t)ns3 * M * p.Cout * sizeof(float) : 0;
        }
      }
    }
    return nsplit > 1 ? (size_t)nsplit * M * p.Cout * sizeof(float) : 0;
  }
  if (tiles < target) {
    nsplit = (int)cdiv(target, tiles);
    int max_split = nchunks / 4;  // keep >= 4 chunks per block
    if (max_split < 1) max_split = 1;
    if (nsplit > max_split) nsplit = max_split;
  }
  int cps = (int)cdiv(nchunks, nsplit);
  nsplit = (int)cdiv(nchunks, cps);
  p.nsplit = nsplit;
  p.chunks_per_split = cps;
  return nsplit > 1 ? (size_t)nsplit * M * p.Cout * sizeof(float) : 0;
}

// Number of GroupNorm-statistics slabs per sample the launch of `p` writes into p.stats (0 = this launch cannot
// produce them: un-split gather kernel; use gn_stats_launch on the output instead).
int conv_stats_slabs(const ConvParams& p) {
  const int64_t V = (int64_t)p.OD * p.OH * p.OW;
  if (p.nsplit > 1) {
    int B, vpb;
    gn_stats_geometry(p.Cout < 1024 ? p.Cout : 1024, V, &B, &vpb);
    return B;
  }
  if (p.mode == 1 && p.bf16t) return (int)(V / 512);  // wide-tile bf16 kernel: one slab per tile
  if (p.mode == 1 && p.bf16 == 2 && p.w_bf && p.Cout >= 64 && !p.skip_w) return (int)(V / 32);  // bf16x3 kernel: per half 8x8 slab
  if (p.mode == 1) return (int)(V / (64 * p.tz)) * (p.Cout >= 64 ? 1 : 2);
  if (p.mode == 2 && V % SM_ROWS == 0) return (int)(V / SM_ROWS);
  if (p.mode == 4) return (int)(V / 128);  // stride-2 bf16 halo kernel: one slab per 2 x 8 x 8 tile
  if (p.mode == 6) return conv1x1_bf16_stream_slabs(p);  // bf16 streaming 1x1x1 kernel: one slab per workgroup row block
  return 0;
}

double conv_flops(const ConvParams& p) {
  const double M = (double)p.N * p.OD * p.OH * p.OW;
  return 2.0 * M * p.Cout * ((double)(p.C0 + p.C1) * p.ksz * p.ksz * p.ksz + (p.skip_w ? p.skip_C0 + p.skip_C1 : 0));
}

// multiply-adds actually issued to the matrix pipe (x2): the Winograd-in-depth kernel spends 36 pseudo-taps where the
// direct form spends 54 (two output planes), and 2 instead of 2 x 1 for the fused skip
double conv_exec_flops(const ConvParams& p) {
  if (!p.wino) return conv_flops(p);
  const double M = (double)p.N * p.OD * p.OH * p.OW;
  if (p.wino == 3)  // 64 pseudo-taps per 2 x 2 x 2 outputs; the fused skip is accumulated directly (1 per output)
    return 2.0 * M * p.Cout * ((double)(p.C0 + p.C1) * 8.0 + (p.skip_w ? p.skip_C0 + p.skip_C1 : 0));
  if (p.wino == 2)  // 48 pseudo-taps per 2 x 2 outputs, the fused skip 4 pseudo-taps per 4 outputs
    return 2.0 * M * p.Cout * ((double)(p.C0 + p.C1) * 12.0 + (p.skip_w ? p.skip_C0 + p.skip_C1 : 0));
  return 2.0 * M * p.Cout * ((double)(p.C0 + p.C1) * 18.0 + (p.skip_w ? p.skip_C0 + p.skip_C1 : 0));
}

int conv_launch(const ConvParams& p, void* stream) {
  const int Cin = p.C0 + p.C1;
  if ((Cin & 3) || (p.C0 & 3) || (p.Cout & 3) || (p.src1 && (p.C0 % BK))) {
    set_error("conv_launch: unsupported channel counts C0=%d C1=%d Cout=%d", p.C0, p.C1, p.Cout);
    return -1;
  }
  if (p.nsplit > 1 && !p.partial) {
    set_error("conv_launch: split-K without scratch");
    return -1;
  }
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;
  const bool wide = p.Cout >= 64;
  const int bn = wide ? 64 : 32;
  dim3 grid((unsigned)cdiv(M, BM), (unsigned)cdiv(p.Cout, bn), (unsigned)p.nsplit);
  dim3 block(256);
  if (p.mode == 1) {
    dim3 hgrid((unsigned)p.grid_x, (unsigned)cdiv(p.Cout, bn), (unsigned)p.nsplit);
    const bool sk = p.skip_w != nullptr;
    if (p.bf16t && p.bf16p) {
      if (conv_bf16p_launch(p, stream)) return -1;
    } else if (p.bf16t) {
#define HOLO_BF16T(NT_, SK_) HOLO_LAUNCH((conv_bf16t_kernel<NT_, SK_, true>), hgrid, block, stream, p)
      if (!p.in_bf16 || (p.residual && !p.res_bf16)) {
        set_error("conv_launch: the wide-tile bf16 kernel runs on bf16 activation storage");
        return -1;
      }
      if (!wide) {
        HOLO_BF16T(1, false);
      } else if (sk) {
        HOLO_BF16T(2, true);
      } else {
        const char* sv = getenv("HOLO_BF16T_SCHED");  // development knob: 0 = operand requests in a clump between the taps
        if (sv && sv[0] == '0') {
          HOLO_LAUNCH((conv_bf16t_kernel<2, false, true, 0>), hgrid, block, stream, p);
        } else {
          HOLO_BF16T(2, false);
        }
      }
#undef HOLO_BF16T
    } else if (p.wino == 3) {
      if (conv_wino3_launch(p, stream)) return -1;
    } else if (p.wino == 2) {
      if (!wide) {
        HOLO_LAUNCH((conv_wino2_kernel<false, 2>), hgrid, block, stream, p);
      } else if (sk) {
        HOLO_LAUNCH((conv_wino2_kernel<true, 4>), hgrid, block, stream, p);
      } else {
        HOLO_LAUNCH((conv_wino2_kernel<false, 4>), hgrid, block, stream, p);
      }
    } else if (p.wino) {
      if (sk) {
        HOLO_LAUNCH(conv_wino_kernel<true>, hgrid, block, stream, p);
      } else {
        HOLO_LAUNCH(conv_wino_kernel<false>, hgrid, block, stream, p);
      }
    } else if (p.bf16 == 2 && p.w_bf && wide && !sk) {
      if (p.tz == 2) {
        HOLO_LAUNCH(conv_halo_split_kernel<2>, hgrid, dim3(512), stream, p);
      } else {
        HOLO_LAUNCH(conv_halo_split_kernel<1>, hgrid, block, stream, p);
      }
    } else {
    const bool bf = p.bf16 == 1 && p.w_bf && (!sk || p.skip_w_bf);
    if ((p.in_bf16 != 0) != bf || (p.residual && (p.res_bf16 != 0) != bf)) {
      set_error("conv_launch: the bf16 halo kernel and bf16 activation storage go together");
      return -1;
    }
    if (!wide && sk) {
      set_error("conv_launch: fused skip needs Cout >= 64");
      return -1;
    }
#define HOLO_HALO(NWN_, TZ_, SK_)                                                                 \
  do {                                                                                            \
    if (bf) {                                                                                     \
      HOLO_LAUNCH((conv_halo_kernel<NWN_, TZ_, SK_, true>), hgrid, block, stream, p);             \
    } else {                                                                                      \
      HOLO_LAUNCH((conv_halo_kernel<NWN_, TZ_, SK_, false>), hgrid, block, stream, p);            \
    }                                                                                             \
  } while (0)
    if (wide && p.tz == 2 && sk) {
      HOLO_HALO(4, 2, true);
    } else if (wide && p.tz == 2) {
      HOLO_HALO(4, 2, false);
    } else if (wide && sk) {
      HOLO_HALO(4, 1, true);
    } else if (wide) {
      HOLO_HALO(4, 1, false);
    } else if (p.tz == 2) {
      HOLO_HALO(2, 2, false);
    } else {
      HOLO_HALO(2, 1, false);
    }
#undef HOLO_HALO
    }
  } else if (p.mode == 4) {
    if (conv_s2_bf16_launch(p, stream)) return -1;
  } else if (p.mode == 5) {
    if (conv1x1_qkv_bf16_launch(p, stream)) return -1;
  } else if (p.mode == 6) {
    if (conv1x1_bf16_stream_launch(p, stream)) return -1;
  } else if (p.mode == 3) {
    if (p.stats || p.residual || p.coef || p.nsplit != 1) {
      set_error("conv_launch: the streaming 1x1x1 kernel takes raw input and produces no statistics");
      return -1;
    }
    // two workgroups per CU, every wave walks 16-row tiles with a stride of the whole grid
    int64_t wgs = cdiv(M >> 4, 4 * 4);  // >= 4 tiles per wave
    if (wgs > 512) wgs = 512;
    if (wgs < 1) wgs = 1;
    dim3 g3((unsigned)wgs, (unsigned)(p.Cout / 64));
    switch (Cin / 32) {
      case 1: HOLO_LAUNCH(conv1x1_stream_kernel<1>, g3, block, stream, p); break;
      case 2: HOLO_LAUNCH(conv1x1_stream_kernel<2>, g3, block, stream, p); break;
      case 3: HOLO_LAUNCH(conv1x1_stream_kernel<3>, g3, block, stream, p); break;
      case 4: HOLO_LAUNCH(conv1x1_stream_kernel<4>, g3, block, stream, p); break;
      case 5: HOLO_LAUNCH(conv1x1_stream_kernel<5>, g3, block, stream, p); break;
      case 6: HOLO_LAUNCH(conv1x1_stream_kernel<6>, g3, block, stream, p); break;
      case 7: HOLO_LAUNCH(conv1x1_stream_kernel<7>, g3, block, stream, p); break;
      case 8: HOLO_LAUNCH(conv1x1_stream_kernel<8>, g3, block, stream, p); break;
      default:
        set_error("conv_launch: streaming 1x1x1 kernel: %d input channels", Cin);
        return -1;
    }
  } else if (p.mode == 2) {
    if (M >= ((int64_t)1 << 31)) {
      set_error("conv_launch: the row-tile kernel indexes output voxels in 32 bits (M = %lld)", (long long)M);
      return -1;
    }
    if (p.skip_w && (p.stride != 1 || p.ups || p.ID != p.OD || (p.bf16 == 1 && p.w_bf && !p.skip_w_bf))) {
      set_error("conv_launch: the row-tile kernel fuses a skip connection at stride 1 without upsampling only");
      return -1;
    }
    dim3 sgrid((unsigned)cdiv(M, SM_ROWS), (unsigned)cdiv(p.Cout, 64), (unsigned)p.nsplit);
    if (p.bf16 == 1 && p.w_bf) {  // bf16 compute mode
      HOLO_LAUNCH(conv_small_kernel<true>, sgrid, block, stream, p);
    } else {
      HOLO_LAUNCH(conv_small_kernel<false>, sgrid, block, stream, p);
    }
  } else if (wide) {
    HOLO_LAUNCH(conv_igemm_kernel<2>, grid, block, stream, p);
  } else {
    HOLO_LAUNCH(conv_igemm_kernel<1>, grid, block, stream, p);
  }
  if (p.nsplit > 1) {
    const int64_t MC = M * p.Cout;
    const int64_t V = (int64_t)p.OD * p.OH * p.OW;
    int B, vpb;
    gn_stats_geometry(p.Cout < 1024 ? p.Cout : 1024, V, &B, &vpb);
    HOLO_LAUNCH(splitk_reduce_kernel, dim3((unsigned)B, (unsigned)p.N, (unsigned)cdiv(p.Cout, 1024)), dim3(256), stream,
                (const float*)p.partial,
                p.nsplit, MC, p.Cout, V, vpb, p.bias, p.skip_bias, p.residual, p.out, p.stats, p.res_bf16, p.out_bf16);
  }
  return 0;
}

}  // namespace holo
