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

int softmax_rows_launch(float* s, int64_t rows, int cols, void* stream) {
  HOLO_LAUNCH(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), stream, s, cols);
  return 0;
}

}  // namespace holo
