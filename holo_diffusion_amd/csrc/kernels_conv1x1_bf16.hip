// kernels_conv1x1_bf16.hip — the qkv convolution of an AttentionBlock in the bf16 storage mode, fused with the operand packing
// of the bf16 attention (holo_diffusion/guided_diffusion/unet.py:300-305: qkv = self.qkv(self.norm(x)) -> QKVAttentionLegacy).
//
// Until round 6 the qkv projection of the long-sequence attention ran on the row-tile kernel (conv_small_kernel: 3 072
// workgroups of 64 rows x 64 channels at T = 32 768: 53 us = 61 TFLOP/s, bound by the workgroup turnover), wrote fp32
// [T][3C] (50 MB), and attn_pack_kernel read that back to write the bf16 operands of flash_attn_bf16v2_kernel (another
// 20 us).  Here ONE streaming GEMM reads the bf16 block input once and writes the packed operands directly:
//   Q bf16 [sample, head][T][CH] scaled by CH^-1/2 * log2 e,  K bf16 [..][T][CH],  V^T bf16 [..][CH][T].
// Workgroup = 4 waves; its slice of the weights (a block of output channels x all input channels, <= 48 KB, the 1 KB
// MFMA-operand blocks of ConvParams::w_bft) is staged into LDS once, then it walks its rows 128 at a time: a wave's 32 rows
// are loaded straight from global memory in operand layout (16 bytes per lane and 16-channel k-step), GroupNorm's affine
// (coefficient rows staged in LDS) is applied on the way, and every 32-channel slice of the block is 32 x 32 x Cin on
// v_mfma_f32_32x32x16_bf16.  A lane's 8 values of an operand block are the same whether the block is used as the A or
// the B operand, so the ORIENTATION of a slice's product follows its destination: Q and K slices are formed transposed
// (weights as A: a lane ends up with 4 consecutive channels of one token -> 8-byte stores into [T][CH]), V slices directly
// (tokens as A: 4 consecutive tokens of one channel -> 8-byte stores into [CH][T]).
//
// conv1x1_bf16_stream_kernel is the same streaming GEMM with a plain [M][Cout] bf16 output: the attention's proj_out (+ bias, +
// residual x, unet.py:306) and any other 1x1x1 convolution of a large grid in this mode; products formed directly (a lane = one
// output channel x 16 tokens of the wave's 32), so the GroupNorm statistics of the output are register sums: one slab per
// workgroup row block, [n][block][Cout][2] doubles.
#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {
namespace {

constexpr int Q1_MAXK = 16;  // k-steps of 16 input channels (Cin <= 256)

template <int NK>  // NK = Cin / 16
__global__ __launch_bounds__(256, 2) void conv1x1_qkv_bf16_kernel(ConvParams p) {
  __shared__ __attribute__((aligned(16))) float s_w[12288];          // [NK][SB][256 words]: <= 48 KB (q1_slices_per_block)
  __shared__ __attribute__((aligned(16))) float s_coef[Q1_MAXK * 32];  // [Cin][2] (a, b) of GroupNorm's affine for the workgroup's sample
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int kg = lane >> 5;
  const int Cin = NK * 16;
  const int SB = p.qkv_sb;                  // 32-channel slices per workgroup
  const int nsl = p.CoutP >> 5;             // slices of the whole weight
  const int sl0 = blockIdx.y * SB;
  const int T = p.qkv_T, CH = p.qkv_CH, H = p.qkv_H;
  const int64_t row0 = (int64_t)blockIdx.x * p.qkv_rows;  // first row (token of the whole batch) of the workgroup
  const int n = (int)(row0 / T);                           // (a workgroup's rows lie in one sample: conv_plan)
  const int t00 = (int)(row0 - (int64_t)n * T);

  // ---- weights of the block -> LDS (1 KB blocks [chunk][slice]); GroupNorm coefficients of the sample -> LDS
  {
    const float* wsrc = reinterpret_cast<const float*>(p.w_bft);
    for (int i = tid; i < NK * SB * 64; i += 256) {  // 16-byte pieces
      const int blk = i >> 6, piece = i & 63;
      const int cc = blk / SB, sl = blk - cc * SB;
      *reinterpret_cast<float4*>(s_w + (int64_t)blk * 256 + piece * 4) =
          *reinterpret_cast<const float4*>(wsrc + ((int64_t)cc * nsl + sl0 + sl) * 256 + piece * 4);
    }
    if (p.coef)
      for (int i = tid; i < Cin * 2; i += 256) s_coef[i] = p.coef[(int64_t)n * Cin * 2 + i];
  }
  __syncthreads();

  const uint16_t* src = reinterpret_cast<const uint16_t*>(p.src0);
  for (int r0 = 0; r0 < p.qkv_rows; r0 += 128) {
    const int t = t00 + r0 + wave * 32 + li;  // the lane's token (row of A / column of the transposed product)
    // ---- the wave's 32 rows in operand layout, GroupNorm's affine applied
    float4 xa[NK];
#pragma unroll
    for (int s = 0; s < NK; ++s) xa[s] = *reinterpret_cast<const float4*>(src + ((int64_t)n * T + t) * Cin + s * 16 + kg * 8);
    if (p.coef) {
#pragma unroll
      for (int s = 0; s < NK; ++s) {
        const float4* cf = reinterpret_cast<const float4*>(s_coef + (s * 16 + kg * 8) * 2);
        const uint32_t w[4] = {__float_as_uint(xa[s].x), __float_as_uint(xa[s].y), __float_as_uint(xa[s].z), __float_as_uint(xa[s].w)};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 c = cf[j];  // (a, b) of channels 2j, 2j + 1
          const float v0 = fmaf(__uint_as_float(w[j] << 16), c.x, c.y);
          const float v1 = fmaf(__uint_as_float(w[j] & 0xffff0000u), c.z, c.w);
          o[j] = pack_bf16x2(v0, v1);
        }
        xa[s] = make_float4(__uint_as_float(o[0]), __uint_as_float(o[1]), __uint_as_float(o[2]), __uint_as_float(o[3]));
      }
    }
    // ---- slice by slice
    for (int sl = 0; sl < SB; ++sl) {
      const int c0 = (sl0 + sl) * 32;           // first output channel of the slice
      const int head = c0 / (3 * CH);
      const int j0 = c0 - head * 3 * CH;
      const int part = j0 / CH;                 // 0 q, 1 k, 2 v
      const int chn0 = j0 - part * CH;
      const int64_t hb = (int64_t)n * H + head;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* wl = s_w + (int64_t)sl * 256 + lane * 4;
      if (part < 2) {  // (uniform) transposed product: rows = channels, column = the lane's token
#pragma unroll
        for (int s = 0; s < NK; ++s)
          acc = mfma_bf16_32x32x16(*reinterpret_cast<const float4*>(wl + (int64_t)s * SB * 256), xa[s], acc);
        const float sc = part == 0 ? p.qkv_scale : 1.f;
        uint16_t* dst = (part == 0 ? p.qkv_q : p.qkv_k) + (hb * T + t) * CH + chn0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // registers 4g .. 4g+3 = channels 8g + 4kg .. +3 of the slice
          const int cl = 8 * g + 4 * kg;
          const float4 b = p.bias ? *reinterpret_cast<const float4*>(p.bias + c0 + cl) : make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<uint2*>(dst + cl) = make_uint2(pack_bf16x2((acc[4 * g] + b.x) * sc, (acc[4 * g + 1] + b.y) * sc),
                                                           pack_bf16x2((acc[4 * g + 2] + b.z) * sc, (acc[4 * g + 3] + b.w) * sc));
        }
      } else {  // direct product: rows = tokens, column = the lane's channel
#pragma unroll
        for (int s = 0; s < NK; ++s)
          acc = mfma_bf16_32x32x16(xa[s], *reinterpret_cast<const float4*>(wl + (int64_t)s * SB * 256), acc);
        const float b = p.bias ? p.bias[c0 + li] : 0.f;
        uint16_t* dst = p.qkv_vt + (hb * CH + chn0 + li) * T + t00 + r0 + wave * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // registers 4g .. 4g+3 = tokens 8g + 4kg .. +3 of the wave's 32
          *reinterpret_cast<uint2*>(dst + 8 * g + 4 * kg) =
              make_uint2(pack_bf16x2(acc[4 * g] + b, acc[4 * g + 1] + b), pack_bf16x2(acc[4 * g + 2] + b, acc[4 * g + 3] + b));
        }
      }
    }
  }
}


constexpr int Q1_MAXSB = 12;  // slices per workgroup (48 KB of weights at 64 input channels)

template <int NK>  // NK = Cin / 16
__global__ __launch_bounds__(256, 2) void conv1x1_bf16_stream_kernel(ConvParams p) {
  __shared__ __attribute__((aligned(16))) float s_w[12288];            // [NK][SB][256 words]
  __shared__ __attribute__((aligned(16))) float s_coef[Q1_MAXK * 32];  // [Cin][2]
  __shared__ float s_st[4 * Q1_MAXSB * 32 * 2];                        // [wave][slice][channel][sum, sumsq]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int kg = lane >> 5;
  const int Cin = NK * 16;
  const int SB = p.qkv_sb;
  const int nsl = p.CoutP >> 5;
  const int sl0 = blockIdx.y * SB;
  const int T = p.qkv_T;  // rows (voxels) per sample
  const int64_t row0 = (int64_t)blockIdx.x * p.qkv_rows;
  const int n = (int)(row0 / T);
  {
    const float* wsrc = reinterpret_cast<const float*>(p.w_bft);
    for (int i = tid; i < NK * SB * 64; i += 256) {
      const int blk = i >> 6, piece = i & 63;
      const int cc = blk / SB, sl = blk - cc * SB;
      *reinterpret_cast<float4*>(s_w + (int64_t)blk * 256 + piece * 4) =
          *reinterpret_cast<const float4*>(wsrc + ((int64_t)cc * nsl + sl0 + sl) * 256 + piece * 4);
    }
    if (p.coef)
      for (int i = tid; i < Cin * 2; i += 256) s_coef[i] = p.coef[(int64_t)n * Cin * 2 + i];
  }
  __syncthreads();
  float ssum[Q1_MAXSB], ssq[Q1_MAXSB];
#pragma unroll
  for (int sl = 0; sl < Q1_MAXSB; ++sl) ssum[sl] = ssq[sl] = 0.f;
  const uint16_t* src = reinterpret_cast<const uint16_t*>(p.src0);
  const uint16_t* res = reinterpret_cast<const uint16_t*>(p.residual);
  uint16_t* out = reinterpret_cast<uint16_t*>(p.out);
  for (int r0 = 0; r0 < p.qkv_rows; r0 += 128) {
    const int64_t rw = row0 + r0 + wave * 32;  // first row of the wave
    float4 xa[NK];
#pragma unroll
    for (int s = 0; s < NK; ++s) xa[s] = *reinterpret_cast<const float4*>(src + (rw + li) * Cin + s * 16 + kg * 8);
    if (p.coef) {
#pragma unroll
      for (int s = 0; s < NK; ++s) {
        const float4* cf = reinterpret_cast<const float4*>(s_coef + (s * 16 + kg * 8) * 2);
        const uint32_t w[4] = {__float_as_uint(xa[s].x), __float_as_uint(xa[s].y), __float_as_uint(xa[s].z), __float_as_uint(xa[s].w)};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 c = cf[j];
          o[j] = pack_bf16x2(fmaf(__uint_as_float(w[j] << 16), c.x, c.y), fmaf(__uint_as_float(w[j] & 0xffff0000u), c.z, c.w));
        }
        xa[s] = make_float4(__uint_as_float(o[0]), __uint_as_float(o[1]), __uint_as_float(o[2]), __uint_as_float(o[3]));
      }
    }
#pragma unroll
    for (int sl = 0; sl < Q1_MAXSB; ++sl) {
      if (sl < SB) {  // (uniform)
        const int co = (sl0 + sl) * 32 + li;  // the lane's output channel
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* wl = s_w + (int64_t)sl * 256 + lane * 4;
#pragma unroll
        for (int s = 0; s < NK; ++s)
          acc = mfma_bf16_32x32x16(xa[s], *reinterpret_cast<const float4*>(wl + (int64_t)s * SB * 256), acc);
        const float b = p.bias ? p.bias[co] : 0.f;
        // D rows = tokens (r & 3) + 8 (r >> 2) + 4 kg of the wave's 32
        float rv[16];
        if (res) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            rv[r] = __uint_as_float((uint32_t)res[(rw + (r & 3) + 8 * (r >> 2) + 4 * kg) * p.Cout + co] << 16);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[r] + b + (res ? rv[r] : 0.f);
          ssum[sl] += v;
          ssq[sl] += v * v;
          out[(rw + (r & 3) + 8 * (r >> 2) + 4 * kg) * p.Cout + co] = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
        }
      }
    }
  }
  // GroupNorm statistics of the output: the two k-groups of a wave by a shuffle, the four waves in LDS in wave order (deterministic)
  if (p.stats) {
#pragma unroll
    for (int sl = 0; sl < Q1_MAXSB; ++sl) {
      if (sl < SB) {
        const float a = ssum[sl] + __shfl_xor(ssum[sl], 32), q = ssq[sl] + __shfl_xor(ssq[sl], 32);
        if (kg == 0) {
          s_st[((wave * Q1_MAXSB + sl) * 32 + li) * 2] = a;
          s_st[((wave * Q1_MAXSB + sl) * 32 + li) * 2 + 1] = q;
        }
      }
    }
    __syncthreads();
    const int blocks_per_sample = T / p.qkv_rows;
    const int slab = (int)(blockIdx.x % blocks_per_sample);
    for (int i = tid; i < SB * 32; i += 256) {
      const int sl = i >> 5, c = i & 31;
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a += s_st[((w * Q1_MAXSB + sl) * 32 + c) * 2];
        q += s_st[((w * Q1_MAXSB + sl) * 32 + c) * 2 + 1];
      }
      double* d = p.stats + (((int64_t)n * blocks_per_sample + slab) * p.Cout + (sl0 + sl) * 32 + c) * 2;
      d[0] = (double)a;
      d[1] = (double)q;
    }
  }
}

}  // namespace

// the 32-channel slices a workgroup takes (its weights <= 48 KB of LDS); 0 = the launch is not for this kernel
static int q1_slices_per_block(const ConvParams& p) {
  const int Cin = p.C0 + p.C1;
  int sb = 49152 / (Cin * 64);
  const int nsl = p.Cout / 32;
  while (sb > 1 && nsl % sb) --sb;
  return sb < 1 ? 0 : sb;
}

bool conv1x1_qkv_bf16_supported(const ConvParams& p) {
  const int Cin = p.C0 + p.C1;
  const int64_t M = (int64_t)p.N * p.OD * p.OH * p.OW;
  return p.qkv_q && p.ksz == 1 && p.stride == 1 && !p.ups && p.bf16 == 1 && p.in_bf16 && p.w_bft && !p.residual && !p.skip_w && !p.stats &&
         p.C1 == 0 && (Cin % 16) == 0 && Cin >= 64 && Cin <= 16 * Q1_MAXK && (p.Cout % 32) == 0 && p.Cout == p.CoutP &&
         p.qkv_CH % 32 == 0 && p.Cout == 3 * p.qkv_CH * p.qkv_H && p.qkv_T % 128 == 0 && M == (int64_t)p.N * p.qkv_T &&
         q1_slices_per_block(p) > 0;
}

// rows per workgroup: a multiple of 128 that divides the sample, about two workgroups per CU
void conv1x1_qkv_bf16_plan(ConvParams& p, int num_cus) {
  p.qkv_sb = q1_slices_per_block(p);
  const int nby = (p.Cout / 32) / p.qkv_sb;
  int rows = 128;
  while (rows * 2 <= p.qkv_T && (p.qkv_T % (rows * 2)) == 0 && ((int64_t)p.N * p.qkv_T / (rows * 2)) * nby >= 2 * (int64_t)num_cus) rows *= 2;
  p.qkv_rows = rows;
}

bool conv1x1_bf16_stream_supported(const ConvParams& p) {
  const int Cin = p.C0 + p.C1;
  const int64_t V = (int64_t)p.OD * p.OH * p.OW;
  return !p.qkv_q && p.ksz == 1 && p.stride == 1 && !p.ups && p.bf16 == 1 && p.in_bf16 && p.out_bf16 && (!p.residual || p.res_bf16) &&
         p.w_bft && !p.skip_w && (!p.coef || !p.act) && p.C1 == 0 && (Cin % 16) == 0 && Cin >= 64 && Cin <= 16 * Q1_MAXK &&
         (p.Cout % 32) == 0 && p.Cout == p.CoutP && p.ID == p.OD && p.IH == p.OH && p.IW == p.OW && V >= 4096 && (V % 128) == 0 &&
         V < ((int64_t)1 << 31) && q1_slices_per_block(p) > 0 && q1_slices_per_block(p) <= 12;
}

// (shares ConvParams::qkv_T / qkv_sb / qkv_rows with the fused qkv form: rows per sample, slices and rows per workgroup)
void conv1x1_bf16_stream_plan(ConvParams& p, int num_cus) {
  p.qkv_T = (int)((int64_t)p.OD * p.OH * p.OW);
  conv1x1_qkv_bf16_plan(p, num_cus);
}

int conv1x1_bf16_stream_slabs(const ConvParams& p) { return p.qkv_rows > 0 ? p.qkv_T / p.qkv_rows : 0; }

int conv1x1_bf16_stream_launch(const ConvParams& p, void* stream) {
  if (!conv1x1_bf16_stream_supported(p) || p.qkv_sb < 1 || p.qkv_rows < 128) {
    set_error("conv1x1_bf16_stream_launch: unsupported launch (%d -> %d channels, %d^3)", p.C0 + p.C1, p.Cout, p.OD);
    return -1;
  }
  const int NK = (p.C0 + p.C1) / 16;
  const dim3 grid((unsigned)((int64_t)p.N * p.qkv_T / p.qkv_rows), (unsigned)((p.Cout / 32) / p.qkv_sb));
#define HOLO_S1(NK_)                                                              \
  case NK_:                                                                       \
    HOLO_LAUNCH(conv1x1_bf16_stream_kernel<NK_>, grid, dim3(256), stream, p); \
    break
  switch (NK) {
    HOLO_S1(4);
    HOLO_S1(8);
    HOLO_S1(12);
    HOLO_S1(16);
    default:
      set_error("conv1x1_bf16_stream_launch: %d input channels", NK * 16);
      return -1;
  }
#undef HOLO_S1
  return 0;
}

int conv1x1_qkv_bf16_launch(const ConvParams& p, void* stream) {
  if (!conv1x1_qkv_bf16_supported(p) || p.qkv_sb < 1 || p.qkv_rows < 128) {
    set_error("conv1x1_qkv_bf16_launch: unsupported launch (%d -> %d channels, T %d)", p.C0 + p.C1, p.Cout, p.qkv_T);
    return -1;
  }
  const int Cin = p.C0 + p.C1, NK = Cin / 16;
  const dim3 grid((unsigned)((int64_t)p.N * p.qkv_T / p.qkv_rows), (unsigned)((p.Cout / 32) / p.qkv_sb));
#define HOLO_Q1(NK_)                                                            \
  case NK_:                                                                     \
    HOLO_LAUNCH(conv1x1_qkv_bf16_kernel<NK_>, grid, dim3(256), stream, p); \
    break
  switch (NK) {
    HOLO_Q1(4);
    HOLO_Q1(8);
    HOLO_Q1(12);
    HOLO_Q1(16);
    default:
      set_error("conv1x1_qkv_bf16_launch: %d input channels", Cin);
      return -1;
  }
#undef HOLO_Q1
  return 0;
}

}  // namespace holo
