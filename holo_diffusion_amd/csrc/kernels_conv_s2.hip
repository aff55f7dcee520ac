// kernels_conv_s2.hip — the stride-2 3x3x3 convolution of a Downsample block (holo_diffusion/guided_diffusion/unet.py:109-138:
// conv_nd(3, ch, ch, 3, stride=2, padding=1) on the RAW block output) in the bf16 storage mode, as an LDS voxel-halo
// implicit GEMM on v_mfma_f32_32x32x16_bf16.
//
// Until round 6 these launches ran on the row-tile kernel (conv_small_kernel), which gathers every (tap, chunk) operand row
// from global memory: 540 us = 107 TFLOP/s for the 128^3 -> 64^3 launch of the 128^3 net, 4 % of the bf16 pipe and
// 0.5 ms of a 16 ms step.  Here:
//   workgroup = 4 waves, output tile = 2 (z) x 8 (y) x 8 (x) voxels x 64 output channels;
//   wave w = output plane w>>1 x output channels 32*(w&1) .. +31: 64 voxels = 2 MFMA row tiles (4 y rows x 8 x each) x 1
//   column tile: per tap and 16-channel chunk 2 A fragments from LDS + 1 B fragment from global / L1 feed 2 MFMAs - LDS
//   and L1 both at their bandwidth when the matrix pipe is saturated, so the kernel is built for ~half of it, with THREE
//   workgroups per CU (49 KB of LDS each) covering each other's staging.
// The input region of a tile is 5 x 17 x 17 voxels.  One 16-channel chunk of it is staged per pass, DE-INTERLEAVED along x:
// LDS row (hz, hy, x parity) holds the 9 voxels hx = 2 xh + parity, 32 bytes each, so that a tap's eight output x positions
// (input x = 2 x + kx) read eight CONSECUTIVE voxels of one row exactly like the stride-1 wide-tile kernel does; the two
// 16-byte halves of a voxel are swapped on every other PAIR of input rows ((hy >> 1) & 1), which makes the four y rows of an
// MFMA row tile (input rows 2 y + ky: every second one) alternate between the two slots - 16 lanes cover 256 bytes, no bank
// conflict.  Raw input: nothing but zero padding is applied while staging; the next chunk's pieces fly under the taps.
// Epilogue: + bias, bf16 store, GroupNorm statistics of the output (one slab per tile: [n][tile][Cout][2] doubles).
#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {
namespace {

constexpr int S2_HZ = 5, S2_HY = 17, S2_HX = 17, S2_XH = 9;
constexpr int S2_SLOTS = S2_HZ * S2_HY * 2 * S2_XH;   // LDS voxel slots (x parity rows of 9)
constexpr int S2_NV = S2_HZ * S2_HY * S2_HX;          // voxels of the input region
constexpr int S2_IT = (2 * S2_NV + 255) / 256;        // 16-byte staging pieces per thread

__global__ __launch_bounds__(256, 3) void conv_s2_bf16_kernel(ConvParams p) {
  __shared__ __attribute__((aligned(16))) float s_in[S2_SLOTS * 8];  // 8 words (16 bf16) per voxel: 48 960 bytes
  __shared__ float s_st[4 * 32 * 2];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;  // MFMA row (A) / column (B, D)
  const int kg = lane >> 5;  // MFMA k-group: channels 8*kg .. +7 of the chunk
  const int zw = wave >> 1;  // the wave's output plane of the tile
  const int ch = wave & 1;   // ... and its 32-channel half of the 64 output channels
  const int Cin = p.C0;
  const int ncc = Cin / 16;
  const int ntx = p.OW >> 3, nty = p.OH >> 3, ntz = p.OD >> 1;
  int bt = blockIdx.x;
  const int tx0 = (bt % ntx) << 3;
  bt /= ntx;
  const int ty0 = (bt % nty) << 3;
  bt /= nty;
  const int tz0 = (bt % ntz) << 1;
  const int n = bt / ntz;
  const int n0 = blockIdx.y * 64;

  // ---- staging plan of the thread: piece i = (voxel, 8-channel half)
  int goff[S2_IT], ldst[S2_IT];
  unsigned okmask = 0, inmask = 0;
#pragma unroll
  for (int i = 0; i < S2_IT; ++i) {
    const int id = tid + 256 * i;
    const int v = min(id >> 1, S2_NV - 1), half = id & 1;
    const int hz = v / (S2_HY * S2_HX);
    const int rem = v - hz * (S2_HY * S2_HX);
    const int hy = rem / S2_HX;
    const int hx = rem - hy * S2_HX;
    int z = 2 * tz0 - 1 + hz, y = 2 * ty0 - 1 + hy, x = 2 * tx0 - 1 + hx;
    const bool in = id < 2 * S2_NV;
    const bool ok = in && z >= 0 && z < p.ID && y >= 0 && y < p.IH && x >= 0 && x < p.IW;
    z = min(max(z, 0), p.ID - 1);
    y = min(max(y, 0), p.IH - 1);
    x = min(max(x, 0), p.IW - 1);
    goff[i] = ((z * p.IH + y) * p.IW + x) * Cin + half * 8;  // (elements; conv_plan keeps a source sample below 2^31 of them)
    ldst[i] = ((((hz * S2_HY + hy) * 2 + (hx & 1)) * S2_XH + (hx >> 1)) * 8) + ((half ^ ((hy >> 1) & 1)) * 4);
    okmask |= (ok ? 1u : 0u) << i;
    inmask |= (in ? 1u : 0u) << i;
  }
  const uint16_t* src = reinterpret_cast<const uint16_t*>(p.src0) + (int64_t)n * p.ID * p.IH * p.IW * Cin;
  float4 hreg[S2_IT];
  auto issue = [&](int cc) {
#pragma unroll
    for (int i = 0; i < S2_IT; ++i) hreg[i] = *reinterpret_cast<const float4*>(src + goff[i] + cc * 16);
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < S2_IT; ++i) {
      const bool keep = (okmask >> i) & 1u;
      float4 v = hreg[i];
      v.x = keep ? v.x : 0.f, v.y = keep ? v.y : 0.f, v.z = keep ? v.z : 0.f, v.w = keep ? v.w : 0.f;
      if ((inmask >> i) & 1u) *reinterpret_cast<float4*>(s_in + ldst[i]) = v;
    }
  };

  // ---- A addressing.  MFMA row li of row tile mt: output y = 4 mt + ys, x = li & 7 (ys = li >> 3); tap (kd, kh, kw) reads
  // input (hz, hy, hx) = (2 zw + kd, 2 y + kh, 2 x + kw): LDS row (hz, hy, kw & 1), voxel x + (kw >> 1), slot
  // kg ^ ((hy >> 1) & 1) = kg ^ ((ys + (kh >> 1)) & 1)
  const int ys = li >> 3;
  const int a_row = ((2 * zw * S2_HY + 2 * ys) * 2 * S2_XH + (li & 7)) * 8;
  const int a_base0 = a_row + ((kg ^ (ys & 1)) * 4);      // taps with kh < 2
  const int a_base1 = a_row + ((kg ^ (ys & 1) ^ 1) * 4);  // taps with kh = 2
  auto load_a = [&](float4 (&a)[2], int tap) {
    const int kd = tap / 9, kh = (tap - kd * 9) / 3, kw = tap - kd * 9 - kh * 3;
    const int toff = (((kd * S2_HY + kh) * 2 + (kw & 1)) * S2_XH + (kw >> 1)) * 8;
    const int ab = kh == 2 ? a_base1 : a_base0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const float4*>(s_in + ab + toff + mt * (8 * 2 * S2_XH * 8));
  };
  // ---- B addressing: 1 KB blocks [tap][chunk][32-Cout slice] (ConvParams::w_bft), 16 bytes per lane
  const int nsl = p.CoutP >> 5;
  const int wncc = p.CinP / 16;
  const float* w_lane = reinterpret_cast<const float*>(p.w_bft) + (int64_t)((n0 >> 5) + ch) * 256 + lane * 4;
  auto load_b = [&](int cc, int tap) { return *reinterpret_cast<const float4*>(w_lane + (int64_t)(tap * wncc + cc) * nsl * 256); };

  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  float4 A[2][2], B[3];
  issue(0);
  B[0] = load_b(0, 0);
  B[1] = load_b(0, 1);
  commit();
  __syncthreads();
  for (int cc = 0; cc < ncc; ++cc) {
    const bool has_next = cc + 1 < ncc;
    if (has_next) issue(cc + 1);  // the next chunk's raw pieces fly under the taps
    load_a(A[0], 0);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      if (tap + 1 < 27) load_a(A[(tap + 1) & 1], tap + 1);
      if (tap + 2 < 27) B[(tap + 2) % 3] = load_b(cc, tap + 2);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma_bf16_32x32x16(A[tap & 1][mt], B[tap % 3], acc[mt]);
    }
    if (has_next) {
      B[0] = load_b(cc + 1, 0);
      B[1] = load_b(cc + 1, 1);
    }
    __syncthreads();  // everyone is done reading this chunk
    if (has_next) {
      commit();
      __syncthreads();
    }
  }

  // ---- epilogue.  D layout of 32x32: column = li (output channel), row i = (r & 3) + 8 (r >> 2) + 4 kg of the row tile:
  // output y = 4 mt + (i >> 3), x = i & 7
  const int co = n0 + ch * 32 + li;
  const float bv = p.bias ? p.bias[co] : 0.f;
  uint16_t* out = reinterpret_cast<uint16_t*>(p.out) +
                  ((((int64_t)n * p.OD + tz0 + zw) * p.OH + ty0) * p.OW + tx0) * p.Cout + co;
  float ssum = 0.f, ssq = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * kg;
      const float v = acc[mt][r] + bv;
      ssum += v;
      ssq += v * v;
      out[(int64_t)((4 * mt + (i >> 3)) * p.OW + (i & 7)) * p.Cout] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
    }
  // GroupNorm statistics of the tensor just produced: one slab per tile (128 voxels) -> stats[n][tile][Cout][2]; the two
  // k-groups of a wave meet by a shuffle, the two planes of the tile in LDS, in a fixed order (deterministic)
  if (p.stats) {
    ssum += __shfl_xor(ssum, 32);
    ssq += __shfl_xor(ssq, 32);
    if (kg == 0) {
      s_st[(wave * 32 + li) * 2] = ssum;
      s_st[(wave * 32 + li) * 2 + 1] = ssq;
    }
    __syncthreads();
    if (zw == 0 && kg == 0) {
      const int tiles_per_sample = ntx * nty * ntz;
      const int slab = blockIdx.x % tiles_per_sample;
      double* d = p.stats + (((int64_t)n * tiles_per_sample + slab) * p.Cout + co) * 2;
      d[0] = (double)(s_st[(wave * 32 + li) * 2] + s_st[((wave + 2) * 32 + li) * 2]);
      d[1] = (double)(s_st[(wave * 32 + li) * 2 + 1] + s_st[((wave + 2) * 32 + li) * 2 + 1]);
    }
  }
}

}  // namespace

bool conv_s2_bf16_supported(const ConvParams& p) {
  const int Cin = p.C0 + p.C1;
  return p.ksz == 3 && p.stride == 2 && p.pad == 1 && !p.ups && p.bf16 == 1 && p.in_bf16 && p.out_bf16 && p.w_bft && !p.coef &&
         !p.residual && !p.skip_w && p.C1 == 0 && (Cin % 16) == 0 && (p.Cout % 64) == 0 && p.ID == 2 * p.OD && p.IH == 2 * p.OH &&
         p.IW == 2 * p.OW && (p.OD % 2) == 0 && (p.OH % 8) == 0 && (p.OW % 8) == 0 &&
         (int64_t)p.ID * p.IH * p.IW * Cin < ((int64_t)1 << 31);
}

int conv_s2_bf16_launch(const ConvParams& p, void* stream) {
  if (!conv_s2_bf16_supported(p) || p.nsplit != 1) {
    set_error("conv_s2_bf16_launch: unsupported launch (stride %d, %d -> %d channels, %d^3 -> %d^3)", p.stride, p.C0 + p.C1, p.Cout,
              p.ID, p.OD);
    return -1;
  }
  const int64_t tiles = (int64_t)p.N * (p.OD >> 1) * (p.OH >> 3) * (p.OW >> 3);
  HOLO_LAUNCH(conv_s2_bf16_kernel, dim3((unsigned)tiles, (unsigned)(p.Cout / 64)), dim3(256), stream, p);
  return 0;
}

}  // namespace holo
