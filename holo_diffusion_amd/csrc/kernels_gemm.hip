// kernels_gemm.hip — batched fp32-MFMA GEMM and row softmax for the UNet attention blocks.
//
// Replaces QKVAttentionLegacy (holo_diffusion/guided_diffusion/unet.py:436-455):
//   weight = einsum("bct,bcs->bts", q*scale, k*scale)     -> gemm (A=q, B=k, alpha=scale^2), k contiguous
//   weight = softmax(weight.float(), dim=-1)               -> softmax_rows
//   a      = einsum("bts,bcs->bct", weight, v)             -> gemm (A=weight, B=v with n contiguous)
// q/k/v are column slices of the token-major qkv buffer [T][3C] (head-major channel order, unet.py:448).
//
// Same tile machinery as kernels_conv.hip: 128 x 64 block tile, 32-deep K chunks, LDS rows of 36 floats,
// v_mfma_f32_32x32x2_f32, lane half h owns k in [16h,16h+16) of a chunk.
#include <stdlib.h>
#include <string.h>

#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {
namespace {

constexpr int BM = 128;
constexpr int BN = 64;
constexpr int BK = 32;
constexpr int LDK = 36;

__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
  constexpr int BUF = (BM + BN) * LDK;
  __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int bz = blockIdx.z;
  const int b0 = bz / p.nb1, b1 = bz - b0 * p.nb1;
  const float* A = p.A + b0 * p.sa0 + b1 * p.sa1;
  const float* B = p.B + b0 * p.sb0 + b1 * p.sb1;
  float* C = p.C + b0 * p.sc0 + b1 * p.sc1;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int nchunks = (p.K + BK - 1) / BK;

  const int q = tid & 7;
  const int r0 = tid >> 3;
  // k-major B staging: 16 threads cover 64 n (float4 each), 16 k rows per pass
  const int nq = tid & 15;
  const int kr = tid >> 4;

  float4 ra[4], rb[2];
  unsigned amask = 0, bmask = 0;

  // unconditional loads from clamped addresses, masked at LDS-store time (a "load or zero" branch would make
  // the compiler serialise the loads)
  auto load_chunk = [&](int kc) {
    const int k = kc * BK + q * 4;
    const bool kok = k < p.K;
    const int kc4 = kok ? k : 0;
    amask = 0;
    bmask = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + r0 + 32 * j;
      ra[j] = *reinterpret_cast<const float4*>(A + (int64_t)min(m, p.M - 1) * p.lda + kc4);
      amask |= ((m < p.M && kok) ? 1u : 0u) << j;
    }
    if (!p.b_kmajor) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + r0 + 32 * j;
        rb[j] = *reinterpret_cast<const float4*>(B + (int64_t)min(n, p.Nn - 1) * p.ldb + kc4);
        bmask |= ((n < p.Nn && kok) ? 1u : 0u) << j;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int kk = kc * BK + kr + 16 * j;
        const int n = n0 + nq * 4;
        rb[j] = *reinterpret_cast<const float4*>(B + (int64_t)min(kk, p.K - 1) * p.ldb + min(n, p.Nn - 4));
        bmask |= ((kk < p.K && n < p.Nn) ? 1u : 0u) << j;
      }
    }
  };
  auto store_chunk = [&](int buf) {
    float* base = lds + buf * BUF;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float keep = ((amask >> j) & 1u) ? 1.f : 0.f;
      float4 v = ra[j];
      v.x *= keep;
      v.y *= keep;
      v.z *= keep;
      v.w *= keep;
      *reinterpret_cast<float4*>(base + (r0 + 32 * j) * LDK + q * 4) = v;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float keep = ((bmask >> j) & 1u) ? 1.f : 0.f;
      float4 v = rb[j];
      v.x *= keep;
      v.y *= keep;
      v.z *= keep;
      v.w *= keep;
      if (!p.b_kmajor) {
        *reinterpret_cast<float4*>(base + (BM + r0 + 32 * j) * LDK + q * 4) = v;
      } else {
        const int kk = kr + 16 * j;
        float* d = base + (BM + nq * 4) * LDK + kk;
        d[0] = v.x;
        d[LDK] = v.y;
        d[2 * LDK] = v.z;
        d[3 * LDK] = v.w;
      }
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const int li = lane & 31;
  const int lh = lane >> 5;

  auto compute = [&](int buf) {
    const float* base = lds + buf * BUF;
    float a[16], b[2][16];
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
    for (int t = 0; t < 2; ++t) {
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
      for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], b[t][ks], acc[t], 0, 0, 0);
  };

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int kc = 0; kc < nchunks; ++kc) {
    const int buf = kc & 1;
    const bool more = kc + 1 < nchunks;
    if (more) load_chunk(kc + 1);
    compute(buf);
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = n0 + t * 32 + li;
    if (n >= p.Nn) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < p.M) C[(int64_t)m * p.ldc + n] = p.alpha * acc[t][r];
    }
  }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one 256-thread block per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int cols) {
  __shared__ float red[8];
  float* row = s + (int64_t)blockIdx.x * cols;
  const int tid = threadIdx.x;
  float mx = -INFINITY;
  for (int i = tid; i < cols; i += 256) mx = fmaxf(mx, row[i]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = tid; i < cols; i += 256) sum += expf(row[i] - mx);
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  const float inv = 1.0f / sum;
  for (int i = tid; i < cols; i += 256) row[i] = expf(row[i] - mx) * inv;
}


// ---------------------------------------------------------------------------------------------
// Flash-style attention for QKVAttentionLegacy (unet.py:436-455), fp32, no materialised T x T scores.
//
// Transposed formulation so that everything per query is lane-local:
//   S^T[key][query] = sum_ch K[key][ch] * Q[query][ch]      A = K tile, B = Q^T (registers, pre-scaled by ch^-1/2)
//   D layout of a 32x32 tile: col = lane&31 = query, rows = 16 keys per lane  -> online softmax of a query
//   is a max/sum over the lane's own 16 registers + one cross-half shuffle;
//   O^T[c][query] += V[key][c] * P^T[key][query]             A = V (global, one dword per k-step),
//   B = the lane's P register r: k-step r pairs key kappa(r) (lane half 0) with kappa(r)+4 (half 1), which is
//   exactly the MFMA k index, so P never leaves its registers.
// Block = 4 waves = ONE tile of 32 queries of one (sample, head); the waves split the T keys in four
// contiguous quarters (each streams its own K/V rows straight from L2, nothing is shared) and the four
// partial (m, l, O) are merged through LDS at the end.  Grid = N * H * T/32 workgroups.
// ---------------------------------------------------------------------------------------------
template <int CHH>  // head channels / 2  (the two lane halves split the head channels = MFMA k index)
__global__ __launch_bounds__(256, 1) void flash_attn_kernel(AttnParams p) {
  constexpr int CH = 2 * CHH;
  constexpr int NCT = (CH + 31) / 32;  // 32-row tiles of O^T
  __shared__ float s_m[4 * 32], s_l[4 * 32];
  __shared__ float s_o[4 * CH * 33];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int lh = lane >> 5;
  const int qtiles = p.T / 32;
  int b = blockIdx.x;
  const int qt = b % qtiles;
  b /= qtiles;
  const int head = b % p.H;
  const int n = b / p.H;
  const int ld = 3 * p.C;
  const float* base = p.qkv + (int64_t)n * p.T * ld + head * 3 * CH;
  const float* Qp = base;
  const float* Kp = base + CH;
  const float* Vp = base + 2 * CH;

  // Q^T operand: lane (query li, half lh) holds Q[q][lh*CHH + ks], pre-scaled by scale^2
  float qreg[CHH];
  {
    const float4* qp = reinterpret_cast<const float4*>(Qp + (int64_t)(qt * 32 + li) * ld + lh * CHH);
#pragma unroll
    for (int v = 0; v < CHH / 4; ++v) {
      const float4 t = qp[v];
      qreg[4 * v + 0] = t.x * p.scale2;
      qreg[4 * v + 1] = t.y * p.scale2;
      qreg[4 * v + 2] = t.z * p.scale2;
      qreg[4 * v + 3] = t.w * p.scale2;
    }
  }
  f32x16 oacc[NCT];
#pragma unroll
  for (int t = 0; t < NCT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int keys_per_wave = p.T / 4;
  const int kbeg = wave * keys_per_wave;
  const int ntile = keys_per_wave / 32;

  float kreg[CHH];
  auto load_k = [&](int kt) {
    const float4* kp = reinterpret_cast<const float4*>(Kp + (int64_t)(kbeg + kt * 32 + li) * ld + lh * CHH);
#pragma unroll
    for (int v = 0; v < CHH / 4; ++v) {
      const float4 t = kp[v];
      kreg[4 * v + 0] = t.x;
      kreg[4 * v + 1] = t.y;
      kreg[4 * v + 2] = t.z;
      kreg[4 * v + 3] = t.w;
    }
  };
  load_k(0);
  for (int kt = 0; kt < ntile; ++kt) {
    // S^T tile
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < CHH; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kreg[ks], qreg[ks], s, 0, 0, 0);
    // V operands of this tile: row kappa(r) + 4*lh, channel ct*32 + li   (issued before the softmax math).
    // (Round 5 requested them one key tile AHEAD into a second register set - nothing else hides their L2 round trip with
    // one wave per SIMD -: 96.3 us per call at T = 4096 against 85.9 us for this form, 151.7 vs 153.5 steps/s.  Reverted.)
    float vreg[NCT][16];
#pragma unroll
    for (int t = 0; t < NCT; ++t) {
      const int c = t * 32 + li;
      const int cc = c < CH ? c : CH - 1;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbeg + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float v = Vp[(int64_t)key * ld + cc];
        vreg[t][r] = c < CH ? v : 0.f;
      }
    }
    if (kt + 1 < ntile) load_k(kt + 1);
    // online softmax for this lane's query
    float mt = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mt = fmaxf(mt, s[r]);
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __expf(s[r] - m_new);
      psum += s[r];
    }
    l_run = l_run * alpha + psum;  // per-lane partial (own 16 keys per tile); halves are merged at the end
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < NCT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
    // O^T += V^T P^T
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int t = 0; t < NCT; ++t) oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vreg[t][r], s[r], oacc[t], 0, 0, 0);
  }

  // ---- merge the four key ranges
  l_run += __shfl_xor(l_run, 32);
  if (lh == 0) {
    s_m[wave * 32 + li] = m_run;
    s_l[wave * 32 + li] = l_run;
  }
#pragma unroll
  for (int t = 0; t < NCT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (c < CH) s_o[(wave * CH + c) * 33 + li] = oacc[t][r];
    }
  __syncthreads();
  float* out = p.out + ((int64_t)n * p.T + qt * 32) * p.C + head * CH;
  for (int idx = tid; idx < 32 * CH; idx += 256) {
    const int c = idx % CH, q = idx / CH;
    const float m0 = s_m[q], m1 = s_m[32 + q], m2 = s_m[64 + q], m3 = s_m[96 + q];
    const float M = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    const float w0 = __expf(m0 - M), w1 = __expf(m1 - M), w2 = __expf(m2 - M), w3 = __expf(m3 - M);
    const float L = w0 * s_l[q] + w1 * s_l[32 + q] + w2 * s_l[64 + q] + w3 * s_l[96 + q];
    const float o = w0 * s_o[(0 * CH + c) * 33 + q] + w1 * s_o[(1 * CH + c) * 33 + q] +
                    w2 * s_o[(2 * CH + c) * 33 + q] + w3 * s_o[(3 * CH + c) * 33 + q];
    out[(int64_t)q * p.C + c] = o / L;
  }
}


// ---------------------------------------------------------------------------------------------
// bf16 attention, second form (bf16 storage mode, T >= 8192).
//
// Pre-pass (attn_pack_kernel): qkv fp32 [N][T][3C] -> per (sample, head): Q bf16 [T][CH] scaled by CH^-1/2 * log2(e)
// (the softmax then runs on v_exp_f32 = 2^x directly), K bf16 [T][CH], V^T bf16 [CH][T].  25 MB in, 25 MB out at
// T = 32 768: ~20 us, against which the main kernel no longer converts or transposes anything per key block (the first
// form did both, with 2-byte LDS stores, once per 128-query workgroup: 256 times per attention call).
//
// Main kernel: workgroup = 4 waves x 64 queries of one (sample, head, key split); key blocks of 64, double buffered in
// LDS (one barrier per block): K rows [key][CH] (stride CH/2+4 words) and V^T rows [ch][64 keys] (stride 34 words),
// both conflict free for the fragment reads.  Transposed formulation on 32x32x16 tiles:
//   S^T[key][query] = K . Q^T      A = K row li of the 32-key tile, channels 8kg..8kg+7 of the k-step (one ds_read_b128),
//                                  B = Q^T operands resident in registers
//   D: column = query li, rows = keys (r&3) + 8(r>>2) + 4kg: a query's 32 keys are the 16 registers of lanes li and
//   li+32 -> max / sum are register reductions plus ONE cross-half shuffle.
//   O^T[c][query] += V^T[c][key] . P^T[key][query]: the lane's registers 8h..8h+7 of an S tile are keys
//   16h + {4kg..4kg+3, 8+4kg..8+4kg+3}: that IS taken as the k order of the k-step, so P goes register -> operand and
//   the A operand reads the same keys of the V^T row (two ds_read_b64).
// Per 64-key block and wave: 32 MFMAs (1 024 pipe cycles) against ~300 vector instructions of softmax: the two
// waves of a SIMD overlap one's softmax with the other's MFMAs, which is why the grid is sized to two workgroups per CU.
// ---------------------------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(256) void attn_pack_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ qb,
                                                        uint16_t* __restrict__ kb, uint16_t* __restrict__ vt, int T, int C,
                                                        int H, float qscale) {
  __shared__ float tile[64][CH + 1];
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int tb = b % (T / 64);
  b /= (T / 64);
  const int head = b % H;
  const int n = b / H;
  const int t0 = tb * 64;
  const float* base = qkv + ((int64_t)n * T + t0) * 3 * C + head * 3 * CH;
  const int64_t hb = (int64_t)n * H + head;
  for (int i = tid; i < 64 * CH / 4; i += 256) {
    const int tok = i / (CH / 4), c4 = i - tok * (CH / 4);
    const float* rp = base + (int64_t)tok * 3 * C + c4 * 4;
    const float4 q = *reinterpret_cast<const float4*>(rp);
    const float4 k = *reinterpret_cast<const float4*>(rp + CH);
    const float4 v = *reinterpret_cast<const float4*>(rp + 2 * CH);
    const int64_t o = (hb * T + t0 + tok) * CH + c4 * 4;
    *reinterpret_cast<uint2*>(qb + o) =
        make_uint2(pack_bf16x2(q.x * qscale, q.y * qscale), pack_bf16x2(q.z * qscale, q.w * qscale));
    *reinterpret_cast<uint2*>(kb + o) = make_uint2(pack_bf16x2(k.x, k.y), pack_bf16x2(k.z, k.w));
    tile[tok][c4 * 4 + 0] = v.x;
    tile[tok][c4 * 4 + 1] = v.y;
    tile[tok][c4 * 4 + 2] = v.z;
    tile[tok][c4 * 4 + 3] = v.w;
  }
  __syncthreads();
  for (int i = tid; i < CH * 8; i += 256) {
    const int ch = i >> 3, t8 = i & 7;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = pack_bf16x2(tile[t8 * 8 + 2 * e][ch], tile[t8 * 8 + 2 * e + 1][ch]);
    uint16_t* dst = vt + (hb * CH + ch) * T + t0 + t8 * 8;
    *reinterpret_cast<uint2*>(dst) = make_uint2(w[0], w[1]);
    *reinterpret_cast<uint2*>(dst + 4) = make_uint2(w[2], w[3]);
  }
}

struct AttnV2 {
  const uint16_t* qb;
  const uint16_t* kb;
  const uint16_t* vt;
  float* out;     // ksplit == 1: [N][T][C] (fp32 or bf16)
  float* opart;   // ksplit > 1: [ksplit][N][T][C] un-normalised O
  float* ml;      // ksplit > 1: [ksplit][N][H][T][2] (running max in the exp2 domain, running sum)
  int N, T, C, H, ksplit, out_bf16;
};

// QT: 32-query tiles per wave (2; 1 for head channels 128, whose 64-query wave tile would need 320 registers)
template <int CH, int QT, bool PIPE = true>
__global__ __launch_bounds__(256, 2) void flash_attn_bf16v2_kernel(AttnV2 p) {
  constexpr int KB = 64;
  constexpr int KW = CH / 2 + 4;  // words per K row
  constexpr int VW = 34;          // words per V^T row (64 keys + 8 bytes)
  constexpr int NKS = CH / 16;    // k-steps of S^T
  constexpr int NCT = CH / 32;    // channel tiles of O^T
  constexpr int PER = CH / 32;    // 16-byte staging pieces per thread, for K and for V^T
  __shared__ __attribute__((aligned(16))) uint32_t s_k[2][KB * KW];
  __shared__ __attribute__((aligned(16))) uint32_t s_v[2][CH * VW];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int li = lane & 31;
  const int kg = lane >> 5;
  const int qtiles = p.T / (128 * QT);
  int b = blockIdx.x;
  const int qt256 = b % qtiles;
  b /= qtiles;
  const int ks = b % p.ksplit;
  b /= p.ksplit;
  const int head = b % p.H;
  const int n = b / p.H;
  const int64_t hb = (int64_t)n * p.H + head;
  const int q0 = qt256 * (128 * QT) + wave * (32 * QT);
  const int klen = p.T / p.ksplit;
  const int kbeg = ks * klen;
  const int nblk = klen / KB;

  float4 qf[QT][NKS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int s = 0; s < NKS; ++s)
      qf[qt][s] = *reinterpret_cast<const float4*>(p.qb + (hb * p.T + q0 + qt * 32 + li) * CH + s * 16 + kg * 8);

  f32x16 oacc[NCT][QT];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[ct][qt][r] = 0.f;
  float m_run[QT], d_run[QT], l_run[QT];  // exponent reference, running maximum relative to it, running sum
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) m_run[qt] = 0.f, d_run[qt] = -3.0e38f, l_run[qt] = 0.f;

  f32x4 kreg[PER], vreg[PER];  // (native vectors: as float4 structs one of the two arrays stayed in scratch memory, and a
                               //  scratch store of a load still in flight stalls the wave for the whole round trip)
  auto stage_load = [&](int blk) {
    const int k0 = kbeg + blk * KB;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid + 256 * j;
      const int key = i / (CH / 8), c8 = i - key * (CH / 8);
      kreg[j] = *reinterpret_cast<const f32x4*>(p.kb + (hb * p.T + k0 + key) * CH + c8 * 8);
      const int ch = i >> 3, k8 = i & 7;
      vreg[j] = *reinterpret_cast<const f32x4*>(p.vt + (hb * CH + ch) * p.T + k0 + k8 * 8);
    }
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid + 256 * j;
      const int key = i / (CH / 8), c8 = i - key * (CH / 8);
      *reinterpret_cast<f32x4*>(&s_k[buf][key * KW + c8 * 4]) = kreg[j];
      const int ch = i >> 3, k8 = i & 7;
      uint32_t* d = &s_v[buf][ch * VW + k8 * 4];
      *reinterpret_cast<uint2*>(d) = make_uint2(__float_as_uint(vreg[j][0]), __float_as_uint(vreg[j][1]));
      *reinterpret_cast<uint2*>(d + 2) = make_uint2(__float_as_uint(vreg[j][2]), __float_as_uint(vreg[j][3]));
    }
  };

  stage_load(0);
  stage_store(0);
  __syncthreads();
  for (int blk = 0; blk < nblk; ++blk) {
    const int buf = blk & 1;
    // ---- S^T tiles [key tile kt][query tile qt]; online softmax per query column (exp2 domain), P^T operands straight
    // from the registers; O^T += V^T . P^T.  The two query tiles are staggered so that the vector work of one tile's
    // softmax sits between the MFMAs of the other tile (a wave issues in order: 16 MFMAs followed by 150 vector
    // instructions leave the matrix pipe idle for the length of the softmax): S(0) | S(1) + softmax(0) |
    // PV(0) + softmax(1) | PV(1).  (Measured alternatives, all within noise or worse: the plain sequential order; sharing
    // the V^T fragments too; the minimal-register sequential form at three waves per SIMD: 27 % slower; an XCD-aware
    // workgroup order (each XCD one K / V^T stream): 7 % slower.)
    f32x16 sacc[2][QT];
    float4 pf[QT][2][2];  // [qt][kt][h]
    // The exponent reference m_ref of a query is NOT its exact running maximum: S - m_ref comes out of the MFMAs (the
    // accumulators start at -m_ref), P = 2^(S - m_ref) may exceed 1, and O, l are re-referenced only when the running
    // maximum has moved more than 2^32 away from m_ref (or in the first block) - any common reference cancels in O / l.
    // That takes the per-element subtraction and, almost always, the rescaling of O out of the vector work, which is
    // what bounds this kernel (softmax ~2x the matrix time at 64 head channels).
    // the K fragments are shared by the two query tiles (read once per block into registers); the V^T fragments are read
    // per query tile - holding them too costs 32 registers at the point where the kernel then spills its staging
    // registers, and a spilled in-flight load stalls the wave for the whole memory round trip
    float4 kaf[NKS][2];
    auto load_k = [&]() {
#pragma unroll
      for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
          kaf[s][kt] = *reinterpret_cast<const float4*>(&s_k[buf][(kt * 32 + li) * KW + s * 8 + kg * 4]);
    };
    auto s_tile = [&](int qt) {
      const float init = -m_run[qt];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kt][qt][r] = init;
#pragma unroll
      for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) sacc[kt][qt] = mfma_bf16_32x32x16(kaf[s][kt], qf[qt][s], sacc[kt][qt]);
    };
    auto softmax_tile = [&](int qt) {
      float mx = sacc[0][qt][0];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kt][qt][r]);
      mx = holo_max_xor32(mx);
      float d = fmaxf(d_run[qt], mx);  // running maximum relative to m_ref
      if (__any(blk == 0 || d > 32.0f)) {  // re-reference (every lane by its own d: valid for any d)
        const float sc = holo_exp2(-d);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc[kt][qt][r] -= d;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[ct][qt][r] *= sc;
        l_run[qt] *= sc;
        m_run[qt] += d;
        d = 0.f;
      }
      d_run[qt] = d;
      float ls = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sacc[kt][qt][r] = holo_exp2(sacc[kt][qt][r]);
          ls += sacc[kt][qt][r];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
          pf[qt][kt][h] = make_float4(__uint_as_float(pack_bf16x2(sacc[kt][qt][8 * h + 0], sacc[kt][qt][8 * h + 1])),
                                      __uint_as_float(pack_bf16x2(sacc[kt][qt][8 * h + 2], sacc[kt][qt][8 * h + 3])),
                                      __uint_as_float(pack_bf16x2(sacc[kt][qt][8 * h + 4], sacc[kt][qt][8 * h + 5])),
                                      __uint_as_float(pack_bf16x2(sacc[kt][qt][8 * h + 6], sacc[kt][qt][8 * h + 7])));
      }
      l_run[qt] += holo_add_xor32(ls);
    };
    auto pv_tile = [&](int qt) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) {
            const uint32_t* vr = &s_v[buf][(ct * 32 + li) * VW + (kt * 32 + 16 * h + 4 * kg) / 2];
            const uint2 lo = *reinterpret_cast<const uint2*>(vr), hi = *reinterpret_cast<const uint2*>(vr + 4);
            const float4 va = make_float4(__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(hi.x), __uint_as_float(hi.y));
            oacc[ct][qt] = mfma_bf16_32x32x16(va, pf[qt][kt][h], oacc[ct][qt]);
          }
    };
    // one MFMA, then a slice of the other tile's softmax
    auto interleave = [&](int n_mfma) {
#pragma unroll
      for (int i = 0; i < n_mfma; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
        __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
    };
    load_k();
    s_tile(0);
    __builtin_amdgcn_sched_barrier(0);
    if (QT == 2) {
      s_tile(QT - 1);
      softmax_tile(0);
      if (PIPE) interleave(2 * NKS);
      __builtin_amdgcn_sched_barrier(0);
      // the next block's K / V^T pieces are requested only now: their 16 staging registers are not live under the S phases,
      // where the register pressure peaks (requested at the top of the block the kernel spilled Q fragments into the loop)
      if (blk + 1 < nblk) stage_load(blk + 1);
      pv_tile(0);
      softmax_tile(QT - 1);
      if (PIPE) interleave(4 * NCT);
      __builtin_amdgcn_sched_barrier(0);
      pv_tile(QT - 1);
    } else {
      softmax_tile(0);
      __builtin_amdgcn_sched_barrier(0);
      if (blk + 1 < nblk) stage_load(blk + 1);
      pv_tile(0);
    }
    if (blk + 1 < nblk) stage_store(buf ^ 1);
    __syncthreads();
  }

  // ---- D rows = channels ct*32 + (r&3) + 8(r>>2) + 4kg, column = query li
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = q0 + qt * 32 + li;
    if (p.ksplit == 1) {
      const float inv = 1.f / l_run[qt];
      const int64_t o = ((int64_t)n * p.T + q) * p.C + head * CH + 4 * kg;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v0 = oacc[ct][qt][4 * g] * inv, v1 = oacc[ct][qt][4 * g + 1] * inv, v2 = oacc[ct][qt][4 * g + 2] * inv,
                      v3 = oacc[ct][qt][4 * g + 3] * inv;
          const int64_t oo = o + ct * 32 + 8 * g;
          if (p.out_bf16)
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + oo) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
          else
            *reinterpret_cast<float4*>(p.out + oo) = make_float4(v0, v1, v2, v3);
        }
    } else {
      const int64_t o = (((int64_t)ks * p.N + n) * p.T + q) * p.C + head * CH + 4 * kg;
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(p.opart + o + ct * 32 + 8 * g) =
              make_float4(oacc[ct][qt][4 * g], oacc[ct][qt][4 * g + 1], oacc[ct][qt][4 * g + 2], oacc[ct][qt][4 * g + 3]);
      if (kg == 0) {
        float* mlp = p.ml + ((((int64_t)ks * p.N + n) * p.H + head) * p.T + q) * 2;
        mlp[0] = m_run[qt];
        mlp[1] = l_run[qt];
      }
    }
  }
}

// recombination of the key splits: out = sum_s 2^(m_s - m) O_s / sum_s 2^(m_s - m) l_s; one thread per (query, 4 channels)
__global__ __launch_bounds__(256) void attn_combine_kernel(AttnV2 p, int CH) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c4n = p.C / 4;
  const int64_t total = (int64_t)p.N * p.T * c4n;
  if (i >= total) return;
  const int c4 = (int)(i % c4n);
  const int64_t nt = i / c4n;
  const int q = (int)(nt % p.T);
  const int n = (int)(nt / p.T);
  const int head = c4 * 4 / CH;
  float m = -3.0e38f;
  for (int s = 0; s < p.ksplit; ++s) m = fmaxf(m, p.ml[((((int64_t)s * p.N + n) * p.H + head) * p.T + q) * 2]);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float l = 0.f;
  for (int s = 0; s < p.ksplit; ++s) {
    const float* mlp = p.ml + ((((int64_t)s * p.N + n) * p.H + head) * p.T + q) * 2;
    const float w = holo_exp2(mlp[0] - m);
    const float4 o = *reinterpret_cast<const float4*>(p.opart + (((int64_t)s * p.N + n) * p.T + q) * p.C + c4 * 4);
    acc.x += w * o.x;
    acc.y += w * o.y;
    acc.z += w * o.z;
    acc.w += w * o.w;
    l += w * mlp[1];
  }
  const float inv = 1.f / l;
  const int64_t oo = ((int64_t)n * p.T + q) * p.C + c4 * 4;
  if (p.out_bf16)
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + oo) =
        make_uint2(pack_bf16x2(acc.x * inv, acc.y * inv), pack_bf16x2(acc.z * inv, acc.w * inv));
  else
    *reinterpret_cast<float4*>(p.out + oo) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
}

}  // namespace

int gemm_launch(const GemmParams& p, void* stream) {
  if ((p.lda & 3) || (p.ldb & 3) || (p.K & 3) || (p.b_kmajor && (p.Nn & 3))) {
    set_error("gemm_launch: leading dimensions / K must be multiples of 4 (lda=%d ldb=%d K=%d N=%d)", p.lda, p.ldb,
              p.K, p.Nn);
    return -1;
  }
  dim3 grid((unsigned)cdiv(p.M, BM), (unsigned)cdiv(p.Nn, BN), (unsigned)(p.nb0 * p.nb1));
  HOLO_LAUNCH(gemm_kernel, grid, dim3(256), stream, p);
  return 0;
}

bool flash_attn_supported(int T, int ch) { return (T % 128) == 0 && (ch == 16 || ch == 32 || ch == 64 || ch == 128); }

int flash_attn_launch(const AttnParams& p, void* stream) {
  const int ch = p.C / p.H;
  if (!flash_attn_supported(p.T, ch)) {
    set_error("flash_attn: unsupported shape T=%d head channels=%d", p.T, ch);
    return -1;
  }
  dim3 grid((unsigned)((int64_t)p.N * p.H * (p.T / 32)));
  switch (ch) {
    case 16:
      HOLO_LAUNCH(flash_attn_kernel<8>, grid, dim3(256), stream, p);
      break;
    case 32:
      HOLO_LAUNCH(flash_attn_kernel<16>, grid, dim3(256), stream, p);
      break;
    case 64:
      HOLO_LAUNCH(flash_attn_kernel<32>, grid, dim3(256), stream, p);
      break;
    default:
      HOLO_LAUNCH(flash_attn_kernel<64>, grid, dim3(256), stream, p);
      break;
  }
  return 0;
}

bool flash_attn_bf16v2_supported(int T, int ch) { return (T % 256) == 0 && (ch == 32 || ch == 64 || ch == 128); }
static int attn_v2_ksplit(const AttnParams& p, int num_cus) {
  const int64_t wgs = (int64_t)p.N * p.H * (p.T / (p.C / p.H == 128 ? 128 : 256));
  int ks = 1;
  while (wgs * ks < 2 * (int64_t)num_cus && ks < 8 && (p.T / (ks * 2)) % 64 == 0) ks *= 2;
  if (const char* e = getenv("HOLO_FLASH_V2_KSPLIT")) {  // development knob
    const int v = atoi(e);
    if (v >= 1 && v <= 8 && (p.T / v) % 64 == 0) ks = v;
  }
  return ks;
}
size_t flash_attn_bf16v2_workspace_bytes(const AttnParams& p, int num_cus) {
  const size_t ntc = (size_t)p.N * p.T * p.C;
  const int ks = attn_v2_ksplit(p, num_cus);
  size_t b = 3 * ntc * sizeof(uint16_t);
  if (ks > 1) b += (size_t)ks * ntc * sizeof(float) + (size_t)ks * p.N * p.H * p.T * 2 * sizeof(float);
  return b;
}
int flash_attn_bf16v2_launch(const AttnParams& p, void* work, int out_bf16, int num_cus, void* stream) {
  const int ch = p.C / p.H;
  if (!flash_attn_bf16v2_supported(p.T, ch)) {
    set_error("flash_attn_bf16v2: unsupported shape T=%d head channels=%d", p.T, ch);
    return -1;
  }
  const size_t ntc = (size_t)p.N * p.T * p.C;
  AttnV2 a;
  uint16_t* w16 = reinterpret_cast<uint16_t*>(work);
  a.qb = w16;
  a.kb = w16 + ntc;
  a.vt = w16 + 2 * ntc;
  a.out = p.out;
  a.opart = reinterpret_cast<float*>(w16 + 3 * ntc);
  a.ksplit = attn_v2_ksplit(p, num_cus);
  a.ml = a.opart + (size_t)a.ksplit * ntc;
  a.N = p.N, a.T = p.T, a.C = p.C, a.H = p.H;
  a.out_bf16 = out_bf16;
  const float qscale = p.scale2 * 1.4426950408889634f;  // softmax in the exp2 domain
  dim3 pgrid((unsigned)((int64_t)p.N * p.H * (p.T / 64)));
  dim3 grid((unsigned)((int64_t)p.N * p.H * a.ksplit * (p.T / (ch == 128 ? 128 : 256))));
  switch (ch) {
    case 32:
      HOLO_LAUNCH(attn_pack_kernel<32>, pgrid, dim3(256), stream, p.qkv, w16, w16 + ntc, w16 + 2 * ntc, p.T, p.C, p.H, qscale);
      HOLO_LAUNCH((flash_attn_bf16v2_kernel<32, 2>), grid, dim3(256), stream, a);
      break;
    case 64:
      HOLO_LAUNCH(attn_pack_kernel<64>, pgrid, dim3(256), stream, p.qkv, w16, w16 + ntc, w16 + 2 * ntc, p.T, p.C, p.H, qscale);
      HOLO_LAUNCH((flash_attn_bf16v2_kernel<64, 2>), grid, dim3(256), stream, a);
      break;
    default:
      HOLO_LAUNCH(attn_pack_kernel<128>, pgrid, dim3(256), stream, p.qkv, w16, w16 + ntc, w16 + 2 * ntc, p.T, p.C, p.H, qscale);
      HOLO_LAUNCH((flash_attn_bf16v2_kernel<128, 1>), grid, dim3(256), stream, a);
      break;
  }
  if (a.ksplit > 1) {
    const int64_t total = (int64_t)p.N * p.T * (p.C / 4);
    HOLO_LAUNCH(attn_combine_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), stream, a, ch);
  }
  return 0;
}

int softmax_rows_launch(float* s, int64_t rows, int cols, void* stream) {
  HOLO_LAUNCH(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), stream, s, cols);
  return 0;
}

}  // namespace holo
