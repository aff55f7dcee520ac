// kernels_conv_bf16p.hip — the stride-1 3x3x3 convolutions of the FILLED levels of the bf16 storage mode (128^3, 64^3:
// BASELINE configs[4], donut.yaml) as a persistent, wave-specialised kernel.  Replaces, like conv_bf16t_kernel
// (kernels_conv.hip), the Conv3d calls of holo_diffusion/guided_diffusion/unet.py:185,211 (ResBlock), :89 (Upsample conv),
// :657 / :792 (input / output convolution) with GroupNorm32 * FiLM + SiLU (:183-184,207-208,248-252), nearest x2
// upsampling (:93-97), the skip concat (:829), the 1x1x1 skip_connection (:222), bias and the residual add (:256) fused.
//
// Why another form.  conv_bf16t_kernel gives every wave all three jobs - request the raw halo, activate it, multiply - and
// relies on the second resident workgroup to cover one wave's vector work with the other's MFMAs; it stays at 0.35 - 0.43 of
// the bf16 pipe.  tools/bf16p_probe (profiles/r06_bf16p_probe.txt) took the work apart on the 128^3 64 -> 64 launch:
//   MFMAs alone 250 us; + A operands from LDS 315; + weights from L1/L2 alone 280; BOTH operands, nobody staging: 325;
//   + waves that stage the next halo: 495 - and of that staging the arithmetic and the LDS writes are nearly free (+45);
//   what costs 170 us is that the halo's LOADS and the weights' LOADS share the CU's vector-memory path, which returns data
//   in order: a weight request (an L2 hit) issued behind a halo request (a miss to the memory side) waits for it, whichever
//   wave issued either, and the waves that multiply starve.  (Each kind of request alone is nearly free: halo loads without
//   weight loads 256 us, weight loads without halo loads 328.)
// So the waves that multiply must not load from memory at all:
//   workgroup = 8 waves, ONE per CU, persistent over (8x8x8-voxel tile, 32*NT-output-channel slice) items;
//   waves 0-3  CONSUMERS: one per SIMD, 2 z-planes of the tile each = 4 x NT register-blocked 32x32 tiles (the blocking of
//              conv_bf16t_kernel); BOTH operands come from LDS: per tap 4 A fragments from the halo buffer and NT B
//              fragments from the weight buffer for 4 * NT v_mfma_f32_32x32x16_bf16; no vector-memory instruction
//              between a tile's first tap and its epilogue;
//   waves 4-7  PRODUCERS: one per SIMD beside a consumer; per 16-channel chunk they request the raw bf16 halo (one chunk
//              ahead of its commit) and the chunk's 27 * NT KB of weights, apply GroupNorm * FiLM + SiLU, zero padding,
//              upsampling and the concat, and write the halo MFMA-ready into the other of two 32 KB LDS buffers.
//              The weights have ONE 54 KB buffer (160 KB of LDS do not hold two): it is refilled in thirds (9 taps) behind
//              the consumers' progress, which every consumer wave publishes in an LDS word after taps 8, 17 and 26.
// Measured (same probe): the consumers' 27-tap loop then takes 3.3 - 3.4 us per chunk (0.9 of the pipe inside the loop) - and
// waits for the producers wherever the halo is ACTIVATED: a producer wave gets its ~350 activation instructions per chunk
// through in ~4.4 us beside a consumer that issues an MFMA every 32 cycles (~1 us alone).  Raw-input launches are faster than
// on conv_bf16t_kernel (128^3 32 -> 64: 236 vs 262 us) and run here by default; activated ones are not yet (540 vs 480 us;
// HOLO_CONV_BF16P=2).  DESIGN.md 4b has the whole ledger.
// One barrier per chunk hands the halo buffer over; the producers run one chunk ahead, across tile boundaries, so a tile's
// first halo is ready when the consumers leave the previous tile's epilogue.  A fused 1x1x1 skip connection is 32-channel
// steps of their own: the producers copy the raw 8^3 centre (no halo, no activation) into the same buffers.
// Halo layout, swizzle, weight layout (w_bft), epilogue (per-wave LDS transposition, 16-byte residual / store, one
// GroupNorm slab per tile) are conv_bf16t_kernel's; no split-K (the planner picks this kernel only where the tiles fill
// the chip).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "holo_common.h"
#include "holo_kernels.h"

// Development probes (tools/bf16p_probe.cpp compiles copies of this file with them; never set in the library build):
//   P_PROBE bits: 1 consumers read no weights, 2 consumers read no A operands, 4 producers stage nothing (barriers only),
//                 8 no MFMAs, 16 producers skip the activation arithmetic, 128 producers load but write nothing,
//                 256 producers write (and compute) but load nothing, 512 the weights are not staged (flags and polls stay)
//   P_TIMELINE:   p.dbg[workgroup][8] = wall-clock ticks (10 ns) of consumer wave 0 {barrier wait, tap loops, epilogue, steps,
//                 items} and of producer wave 4 {all work, barrier wait, of the work: halo commit}
#ifndef P_PROBE
#define P_PROBE 0
#endif
#ifndef P_ENTRY
#define P_ENTRY conv_bf16p_launch
#endif
// P_PHASED 0 (default): the overlapped form (one barrier per chunk, weights refilled in thirds behind the consumers' progress
// flags).  1: producers and consumers take TURNS - the consumers' tap loop and the producers' commit never run at the same time
// (two barriers per chunk).  Measured side by side (tools/bf16p_probe, profiles/r06_bf16p_probe_v4_phased.txt): turns win on
// activated inputs (128^3 64 -> 64: 519 vs 567 us; conv_bf16t_kernel 468) and lose on the raw-input launches this kernel is the
// planner's default for (32 -> 64: 292 vs 279 us) - the producers' chain per chunk is 4.8 us of exposed latency either way.
#ifndef P_PHASED
#define P_PHASED 0
#endif

namespace holo {
namespace {

constexpr int P_H = 10;                 // halo edge of an 8^3 tile
constexpr int P_HV = P_H * P_H * P_H;   // halo voxels
constexpr int P_CK = 16;                // channels per chunk = K of one v_mfma_f32_32x32x16_bf16
constexpr int P_RS = 8;                 // LDS words per halo voxel (16 bf16; the two halves swapped on odd halo rows)
constexpr int P_IT = 8;                 // staging items (voxel, 8-channel half) per producer thread: 2000 / 256
constexpr int P_BUF = P_HV * P_RS + 192;  // words per halo buffer (8 192: a skip step's two 4 096-word k-step planes fit)
constexpr int P_EW = 68;                // words per voxel row of a consumer's transposition tile

__device__ __forceinline__ void unpack8(const float4& v, float (&f)[8]) {
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
__device__ __forceinline__ float4 pack8(const float (&f)[8]) {
  return make_float4(__uint_as_float(pack_bf16x2(f[0], f[1])), __uint_as_float(pack_bf16x2(f[2], f[3])),
                     __uint_as_float(pack_bf16x2(f[4], f[5])), __uint_as_float(pack_bf16x2(f[6], f[7])));
}

// Buffer addressing of the producers' sources: scalar resource (base of the step's source / sample / chunk) + one 32-bit byte
// offset per item; an offset beyond the resource's range (~0) reads zeros.  A source sample must lie within 4 GB (conv_plan).
#ifdef HOLO_EMU
struct pb_rsrc {
  const char* base;
};
static inline pb_rsrc pb_make_rsrc(const void* p) { return pb_rsrc{reinterpret_cast<const char*>(p)}; }
static inline float4 pb_load(const pb_rsrc& r, unsigned voff) {
  if (voff >= 0xfffffff0u) return make_float4(0.f, 0.f, 0.f, 0.f);
  return *reinterpret_cast<const float4*>(r.base + (size_t)voff);
}
#else
typedef __amdgpu_buffer_rsrc_t pb_rsrc;
typedef unsigned pb_u4 __attribute__((ext_vector_type(4)));
// (the base is wave-uniform by construction, but the compiler cannot prove it for values carried through the producers' step
// records: without the two v_readfirstlane it wraps EVERY buffer load in a waterfall loop - cdna_hip_programming.md T20)
__device__ __forceinline__ pb_rsrc pb_make_rsrc(const void* p) {
  const uint64_t a = (uint64_t)p;
  const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(u), 0, 0xfffffff0, 0x00020000);
}
__device__ __forceinline__ float4 pb_load(pb_rsrc r, unsigned voff) {
  const pb_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
#endif

// a * b + c for a, b < 2^24 (voxel indices x bytes per voxel): v_mad_u32_u24, full rate (the 32-bit multiply is quarter rate)
#ifdef HOLO_EMU
static inline unsigned pb_mad24(unsigned a, unsigned b, unsigned c) { return a * b + c; }
#else
__device__ __forceinline__ unsigned pb_mad24(unsigned a, unsigned b, unsigned c) { return __umul24(a, b) + c; }
#endif

// The hand-over barrier of a step: LDS traffic of this wave done (lgkmcnt), then s_barrier.  NOT __syncthreads(): that also
// waits for the wave's outstanding vector-memory loads (vmcnt(0)), and the producers keep the loads of the step after next in
// flight across it on purpose.
#ifdef HOLO_EMU
#define P_STEP_BARRIER() __syncthreads()
#define P_POLL_SLEEP() std::this_thread::yield()
#define P_COMPILER_FENCE() std::atomic_thread_fence(std::memory_order_seq_cst)
#else
#define P_STEP_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define P_POLL_SLEEP() __builtin_amdgcn_s_sleep(4)
#define P_COMPILER_FENCE() asm volatile("" ::: "memory")
#endif

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): register arrays indexed with compile-time constants (a
// `#pragma unroll` loop nested in the unrolled third loop was left rolled and sent the staged weights to scratch memory)
template <class F, int... K>
__device__ __forceinline__ void p_static_for_impl(F&& f, std::integer_sequence<int, K...>) {
  (f(std::integral_constant<int, K>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void p_static_for(F&& f) {
  p_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// one staged step of a producer thread: the raw 16-byte pieces in flight + what the commit needs to know about them
struct Pend {
  float4 h[P_IT];
  unsigned mask;  // main step: per item, the voxel lies inside the tensor
  int kind;       // 0 = nothing (the list is exhausted), 1 = main 16-channel chunk, 2 = 32-channel skip step
  int c;          // main: first channel of this thread's 8-channel half (for the affine coefficients)
  int n;          // sample
  int first;      // first step of an item that is not the workgroup's first: the consumers' statistics barrier comes first
  int ph, slice;  // the step's place in its item and the item's output-channel slice (for its weights)
};

template <int NT, bool SKIP>
__global__ __launch_bounds__(512, 2) void conv_bf16p_kernel(ConvParams p) {
  constexpr int BN = 32 * NT;
  __shared__ __attribute__((aligned(16))) float s_halo[2 * P_BUF];
  __shared__ __attribute__((aligned(16))) float s_ep[4 * 32 * P_EW];
  __shared__ float s_stat[4 * 8 * 16];
  constexpr int WBLK = 256 * NT;                 // floats of a tap's weights: NT 1 KB blocks
  constexpr int NWV = (27 * NT * 64 + 255) / 256;  // float4 per producer thread and step (27 taps)
  constexpr int WT1 = 9 * NT * 64, WT2 = 18 * NT * 64;  // first float4 of the second / last third (taps 9.., 18..)
  __shared__ __attribute__((aligned(16))) float s_bw[27 * WBLK];  // the current step's weights [tap][nt][lane][4]
  __shared__ __attribute__((aligned(16))) int s_flag[4];          // per consumer wave: 3 * (steps done) + thirds of the current one

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int Cin = p.C0 + p.C1;
  const int ncc = (Cin + P_CK - 1) / P_CK;
  const int SCin = SKIP ? p.skip_C0 + p.skip_C1 : 0;
  const int nsp = SKIP ? (SCin + 31) / 32 : 0;  // skip steps of 32 channels (two k-steps)
  const int nst = ncc + nsp;                    // steps per item
  const int ntx = p.OW >> 3, nty = p.OH >> 3, ntz = p.OD >> 3;
  const int tiles_per_sample = ntx * nty * ntz;
  const int nslices = (p.Cout + BN - 1) / BN;
  const int nitems = p.N * tiles_per_sample * nslices;
  const bool with_stats = p.stats != nullptr;  // (uniform)
  // item list of this workgroup.  Workgroup b runs on XCD b % 8 (observed; for speed only): every XCD walks one contiguous
  // range of the item list - neighbouring tiles, whose halos overlap, then meet in one L2.
  const int G = gridDim.x;
  int it_first, it_step, it_end;
  if ((G & 7) == 0) {
    const int per = (nitems + 7) >> 3;
    const int x = blockIdx.x & 7;
    it_first = x * per + (blockIdx.x >> 3);
    it_step = G >> 3;
    it_end = min((x + 1) * per, nitems);
  } else {
    it_first = blockIdx.x;
    it_step = G;
    it_end = nitems;
  }
  // item -> (sample, tile origin, slice): slices of one tile are neighbours in the list (they share the halo)
  auto decode = [&](int item, int& n, int& tz0, int& ty0, int& tx0, int& slice) {
    slice = item % nslices;
    int bt = item / nslices;
    tx0 = (bt % ntx) << 3;
    bt /= ntx;
    ty0 = (bt % nty) << 3;
    bt /= nty;
    tz0 = (bt % ntz) << 3;
    n = bt / ntz;
  };

  if (tid < 4) s_flag[tid] = 0;
  __syncthreads();

  if (wave >= 4) {
    // =============================================== PRODUCERS ===============================================
    // Everything here is written for a LOW INSTRUCTION COUNT: the producers share their SIMD's issue port with a consumer
    // that wants to issue an MFMA every 32 cycles (the first form of this section took ~1 400 - 2 400 instructions per step
    // and wave - 64-bit per-lane address arithmetic, eight selects per item - and cost the consumers 35 % of their time:
    // tools/bf16p_probe).  Sources are addressed through buffer resources (scalar base per step, one 32-bit offset per
    // item; an offset of ~0 reads zeros), per-thread geometry is computed once per kernel / tile, the arithmetic is packed.
    const int ptid = tid - 256;
    const int hh = ptid & 1;  // main steps: which 8-channel half of the 16-channel chunk
#if !defined(HOLO_EMU) && defined(P_PRODUCER_PRIO)
    __builtin_amdgcn_s_setprio(P_PRODUCER_PRIO);  // (probe: the producers' vector work first in the SIMD's issue arbitration)
#endif
    const int SD = p.ups ? (p.ID >> 1) : p.ID;
    const int SH = p.ups ? (p.IH >> 1) : p.IH;
    const int SW = p.ups ? (p.IW >> 1) : p.IW;
    // ---- per-thread constants of its P_IT items.  Main steps: item id = ptid + 256 i = (halo voxel id >> 1, half id & 1);
    // skip steps: id = (centre voxel id >> 2, 8-channel quarter id & 3)
    int hzyx[P_IT], lds_off[P_IT], slds_off[P_IT];
    unsigned svoff[P_IT];
    unsigned live = 0;  // main items that exist (2 000 of 2 048)
#pragma unroll
    for (int i = 0; i < P_IT; ++i) {
      const int id = ptid + 256 * i;
      const int hv = min(id >> 1, P_HV - 1);
      const int hz = hv / (P_H * P_H);
      const int rem = hv - hz * (P_H * P_H);
      const int hy = rem / P_H;
      const int hx = rem - hy * P_H;
      hzyx[i] = hz | (hy << 8) | (hx << 16);
      lds_off[i] = hv * P_RS + ((hh ^ (hy & 1)) * 4);  // the voxel's halves swapped on odd halo rows
      live |= (id < 2 * P_HV ? 1u : 0u) << i;
      const int v = id >> 2, qd = id & 3;
      const int vz = v >> 6, vy = (v >> 3) & 7, vx = v & 7;
      slds_off[i] = (qd >> 1) * 4096 + v * P_RS + (((qd & 1) ^ (vy & 1)) * 4);
      svoff[i] = (unsigned)((vz * p.OH + vy) * p.OW + vx);  // x (channels of the source * 2) + qd * 16 at issue time
    }
    // weights: float4 number f = ptid + 256 j of a step's blocks; block f / 64 = tap * NT + nt, 16 bytes f % 64 of it.  Byte
    // offset of that float4 from the step's base (tap 0, the step's chunk, the item's slice) in w_bft
    // ([tap][16-channel chunk][32-Cout slice][1 KB]); ~0 (reads zeros, never written) beyond the last block
    const int nsl = p.CoutP >> 5;
    const int wncc = p.CinP / P_CK;
    // (f -> f + 256 moves 4 blocks on = 4 / NT taps: the offsets of a thread's float4s are w0 + j wstride)
    const int wblk0 = ptid >> 6, wtap0 = wblk0 / NT;
    const unsigned w0 = (unsigned)(((wtap0 * wncc * nsl + (wblk0 - wtap0 * NT)) * 256 + (ptid & 63) * 4) * 4);
    const unsigned wstride = (unsigned)((4 / NT) * wncc * nsl * 256 * 4);
    // iterator over (item, step) in consumption order
    int item = it_first, ph = 0;
    int cn = 0, ctz0 = 0, cty0 = 0, ctx0 = 0, cslice = 0;
    unsigned hvox[P_IT];
    unsigned hvalid = 0;
    bool very_first = true;
    auto setup_tile = [&]() {
      decode(item, cn, ctz0, cty0, ctx0, cslice);
      hvalid = 0;
#pragma unroll
      for (int i = 0; i < P_IT; ++i) {
        int z = ctz0 + (hzyx[i] & 0xff) - 1, y = cty0 + ((hzyx[i] >> 8) & 0xff) - 1, x = ctx0 + (hzyx[i] >> 16) - 1;
        const bool ok = (unsigned)z < (unsigned)p.ID && (unsigned)y < (unsigned)p.IH && (unsigned)x < (unsigned)p.IW;
        z = min(max(z, 0), p.ID - 1);
        y = min(max(y, 0), p.IH - 1);
        x = min(max(x, 0), p.IW - 1);
        if (p.ups) {
          z >>= 1;
          y >>= 1;
          x >>= 1;
        }
        hvox[i] = (unsigned)((z * SH + y) * SW + x);
        hvalid |= (ok ? 1u : 0u) << i;
      }
      hvalid &= live;
    };
    if (item < it_end) setup_tile();
    // (cf: main step with an affine - (a, b) of the thread's 8 channels)
    auto issue = [&](Pend& q, float4 (&cf)[4]) {
      if (item >= it_end) {
        q.kind = 0;
        return;
      }
      q.first = (ph == 0 && !very_first) ? 1 : 0;
      very_first = false;
      q.ph = ph;
      q.slice = cslice;
      if (P_PROBE & 4) {  // (probe: the step list without its loads)
        q.kind = 3;
      } else if (ph < ncc) {  // ---- a 16-channel chunk of the 10^3 halo.  A chunk lies in ONE source (conv_plan: C0 % 16 == 0)
        q.kind = 1;
        q.n = cn;
        const int c0 = ph * P_CK;
        q.c = c0 + hh * 8;
        const bool second = c0 >= p.C0;  // (uniform)
        const int Cs = second ? p.C1 : p.C0;
        const char* sb = reinterpret_cast<const char*>(second ? p.src1 : p.src0) +
                         ((int64_t)cn * SD * SH * SW * Cs + (second ? c0 - p.C0 : c0)) * 2;
        const pb_rsrc rs = pb_make_rsrc(sb);
        const unsigned cbytes = (unsigned)Cs * 2u, hoff = (unsigned)hh * 16u;
        // raw copies (no affine, no activation) read zeros for the padding through an out-of-range offset; activated chunks
        // read a clamped neighbour and are zeroed AFTER the activation (commit)
        const bool raw = p.coef == nullptr;
        q.mask = hvalid;
#pragma unroll
        for (int i = 0; i < P_IT; ++i) {
          unsigned off = pb_mad24(hvox[i], cbytes, hoff);
          if (raw && !((hvalid >> i) & 1u)) off = 0xffffffffu;
          if (P_PROBE & 256) {  // (probe: no loads, the LDS writes and the arithmetic stay)
            q.h[i] = make_float4(__uint_as_float(off), 0.f, 0.f, 0.f);
          } else {
            q.h[i] = pb_load(rs, off);
          }
        }
        if (!raw) {
          const float4* cfp = reinterpret_cast<const float4*>(p.coef + ((int64_t)cn * Cin + q.c) * 2);
#pragma unroll
          for (int j = 0; j < 4; ++j) cf[j] = cfp[j];
        }

      } else {  // ---- 32 raw channels of the skip connection's input on the tile's 8^3 centre (one source: skip_C0 % 32 == 0)
        q.kind = 2;
        const int c0 = (ph - ncc) * 32;
        const bool second = c0 >= p.skip_C0;  // (uniform)
        const int Cs = second ? p.skip_C1 : p.skip_C0;
        const int64_t org = (((int64_t)cn * p.OD + ctz0) * p.OH + cty0) * p.OW + ctx0;
        const char* sb = reinterpret_cast<const char*>(second ? p.skip_src1 : p.skip_src0) +
                         (org * Cs + (second ? c0 - p.skip_C0 : c0)) * 2;
        const pb_rsrc rs = pb_make_rsrc(sb);
        const unsigned cbytes = (unsigned)Cs * 2u, qoff = (unsigned)(ptid & 3) * 16u;
#pragma unroll
        for (int i = 0; i < P_IT; ++i) q.h[i] = pb_load(rs, pb_mad24(svoff[i], cbytes, qoff));
      }
      if (++ph == nst) {  // the iterator moves on; the next tile's geometry is ready for its first issue
        ph = 0;
        item += it_step;
        if (item < it_end) setup_tile();
      }
    };
    // The step's weights (w: float4 number ptid + 256 j of its 27 * NT (main) / 2 * NT (skip) 1 KB blocks): requested at the
    // END of the previous step's iteration, when its registers are free again (L2 hits, and their first third is needed only
    // a third of a step into the next one).
    auto issue_weights = [&](int kind, int wph, int wslice, float4 (&w)[NWV]) {
      if (P_PROBE & (4 | 128 | 256 | 512)) return;
      if (kind == 1) {
        const pb_rsrc rw = pb_make_rsrc(reinterpret_cast<const float*>(p.w_bft) + ((int64_t)wph * nsl + wslice * NT) * 256);
#pragma unroll
        for (int j = 0; j < NWV; ++j) {
          const bool ok = wtap0 + j * (4 / NT) < 27;  // (the last float4 of some threads lies beyond the 27th tap)
          w[j] = pb_load(rw, ok ? w0 + (unsigned)j * wstride : 0xffffffffu);
        }
      } else if (kind == 2 && ptid < 2 * NT * 64) {  // blocks (k-step e, nt) of [skip chunk 2 sp + e][slice]
        const int e = ptid / (NT * 64), r = ptid - e * (NT * 64);
        const float* wb = reinterpret_cast<const float*>(p.skip_w_bft) +
                          ((int64_t)(2 * (wph - ncc) + e) * nsl + wslice * NT) * 256 + r * 4;
        w[0] = *reinterpret_cast<const float4*>(wb);
      }
    };
    auto commit = [&](const Pend& q, const float4 (&cf)[4], float* buf) {
      if (P_PROBE & 128) {  // (probe: the loads are waited for, nothing is written)
        float acc_ = 0.f;
#pragma unroll
        for (int i = 0; i < P_IT; ++i) acc_ += q.h[i].x;
        if (acc_ == 1.2345e-30f) buf[0] = acc_;
        return;
      }
      if (q.kind == 1) {
        if (p.coef == nullptr) {  // raw: bf16 in, bf16 out (the padding already reads as zero)
#pragma unroll
          for (int i = 0; i < P_IT; ++i)
            if ((live >> i) & 1u) *reinterpret_cast<float4*>(buf + lds_off[i]) = q.h[i];
          return;
        }
        f32x2 ca[4], cb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 c = cf[j];  // (a, b) interleaved per channel
          ca[j] = f32x2{c.x, c.z};
          cb[j] = f32x2{c.y, c.w};
        }
        const bool act = p.act != 0 && !(P_PROBE & 16);
        const bool boundary = q.mask != live;  // (zero padding only where the tile touches the tensor's faces)
        const f32x2 nl2e = f32x2{-1.4426950408889634f, -1.4426950408889634f}, one = f32x2{1.f, 1.f};
#pragma unroll
        for (int i = 0; i < P_IT; ++i) {
          const uint32_t w[4] = {__float_as_uint(q.h[i].x), __float_as_uint(q.h[i].y), __float_as_uint(q.h[i].z),
                                 __float_as_uint(q.h[i].w)};
          uint32_t o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f32x2 v = f32x2{__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xffff0000u)};
            v = pk_fma(v, ca[j], cb[j]);
            if (act) {  // SiLU: v * rcp(1 + exp2(-v log2 e)), packed where the unit has a packed form
              f32x2 t = pk_mul(v, nl2e);
              t = f32x2{holo_exp2(t.x), holo_exp2(t.y)};
              t = pk_add(t, one);
              t = f32x2{holo_rcp(t.x), holo_rcp(t.y)};
              v = pk_mul(v, t);
            }
            o[j] = pack_bf16x2(v.x, v.y);
          }
          if (boundary && !((q.mask >> i) & 1u)) o[0] = o[1] = o[2] = o[3] = 0u;  // zero padding AFTER the activation
          if ((live >> i) & 1u)
            *reinterpret_cast<float4*>(buf + lds_off[i]) =
                make_float4(__uint_as_float(o[0]), __uint_as_float(o[1]), __uint_as_float(o[2]), __uint_as_float(o[3]));
        }
      } else if (q.kind == 2) {  // raw copy: [k-step][centre voxel][8 words], the halves swapped on odd y rows like the halo's
#pragma unroll
        for (int i = 0; i < P_IT; ++i) *reinterpret_cast<float4*>(buf + slds_off[i]) = q.h[i];
      }
    };
#ifdef P_TIMELINE
    unsigned long long pt_commit = 0, pt_wts = 0, pt_last = HOLO_PROBE_CLOCK();
#define P_TL_STAMP(acc_)                                \
  {                                                     \
    const unsigned long long t_ = HOLO_PROBE_CLOCK();   \
    acc_ += t_ - pt_last;                               \
    pt_last = t_;                                       \
  }
    unsigned long long pt_work = 0, pt_wait = 0, pt0 = HOLO_PROBE_CLOCK(), pt1;
#define P_TL_PROD_BAR()                 \
  pt1 = HOLO_PROBE_CLOCK();             \
  pt_work += pt1 - pt0;                 \
  P_STEP_BARRIER();                     \
  pt0 = HOLO_PROBE_CLOCK();             \
  pt_last = pt0;                        \
  pt_wait += pt0 - pt1
#else
#define P_TL_PROD_BAR() P_STEP_BARRIER()
#define P_TL_STAMP(acc_)
#endif
    // The step's weights go into the ONE weight buffer behind the consumers' progress: third k (taps 9k .. 9k + 8) of step s
    // may be overwritten once every consumer wave has published 3 (s - 1) + k + 1 (it has issued the MFMAs of tap 9k + 8 of
    // step s - 1, hence finished reading that third; LDS serves a wave's requests in order).
    auto wait_flags = [&](int target) {
      if (target <= 0) return;
      for (;;) {
        P_COMPILER_FENCE();
        const volatile int* f = s_flag;
        const int a0 = f[0], a1 = f[1], a2 = f[2], a3 = f[3];
        if (min(min(a0, a1), min(a2, a3)) >= target) break;
        P_POLL_SLEEP();
      }
      P_COMPILER_FENCE();
    };
    auto commit_weights = [&](int kind, const float4 (&w)[NWV], int s) {
      if (P_PROBE & (4 | 128 | 256)) return;
      if (kind == 2) {  // a skip step's 2 * NT blocks: the previous step must be over
        wait_flags(3 * s);
        if (ptid < 2 * NT * 64 && !(P_PROBE & 512)) *reinterpret_cast<float4*>(s_bw + ptid * 4) = w[0];
        return;
      }
      p_static_for<3>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int lo = k == 0 ? 0 : (k == 1 ? WT1 : WT2), hi = k == 0 ? WT1 : (k == 1 ? WT2 : 27 * NT * 64);
        wait_flags(3 * (s - 1) + k + 1);
        if (P_PROBE & 512) return;
        p_static_for<NWV>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if (256 * j + 255 >= lo && 256 * j < hi) {  // (compile time: this float4 can lie in third k for some thread)
            const int f = ptid + 256 * j;
            if (f >= lo && f < hi) *reinterpret_cast<float4*>(s_bw + f * 4) = w[j];
          }
        });
      });
    };
    (void)commit_weights;
    Pend P;
    float4 W[NWV], CF[4];
    issue(P, CF);
    issue_weights(P.kind, P.ph, P.slice, W);
    int s = 0;
#if P_PHASED
    // Turns.  A producer wave that activates a halo beside a consumer issuing an MFMA every 32 cycles takes ~4.4 us for what it
    // does in ~1 us alone (measured, tools/bf16p_probe), and the consumers then wait for it: overlapping the two streams on one
    // SIMD costs more than it hides.  So: [hand-over A] the consumers multiply step s while the producers only WAIT - their
    // requests for step s + 1 are in flight - [hand-over B] the producers commit step s + 1 (halo and all of its weights: one
    // buffer each is enough now) while the consumers wait (or run a tile's epilogue) [hand-over A] ...
    while (P.kind != 0) {
      const int kind = P.kind, first = P.first;
      commit(P, CF, s_halo + (s & 1) * P_BUF);
      P_TL_STAMP(pt_commit);
      if (!(P_PROBE & (4 | 128 | 256 | 512))) {
        if (kind == 2) {
          if (ptid < 2 * NT * 64) *reinterpret_cast<float4*>(s_bw + ptid * 4) = W[0];
        } else {
          p_static_for<NWV>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int f = ptid + 256 * j;
            if (f < 27 * NT * 64) *reinterpret_cast<float4*>(s_bw + f * 4) = W[j];
          });
        }
      }
      P_TL_STAMP(pt_wts);
      issue(P, CF);  // (step s + 1: in flight while the consumers multiply step s)
      issue_weights(P.kind, P.ph, P.slice, W);
      if (first && with_stats) __syncthreads();  // (the consumers' statistics hand-over of the previous item)
      P_TL_PROD_BAR();                            // A: step s is staged
      P_TL_PROD_BAR();                            // B: step s has been consumed
      ++s;
    }
#else
    // Overlapped form.  One register set for each kind of request.  Iteration s (between the hand-overs of steps s - 1 and s)
    // commits step s:
    //   halo(s) [requested early in iteration s - 1] -> request halo(s + 1) -> weights(s) in thirds behind the consumers
    //   [requested at the end of iteration s - 1] -> request weights(s + 1) -> hand-over.
    while (P.kind != 0) {
      const int kind = P.kind, first = P.first;
      commit(P, CF, s_halo + (s & 1) * P_BUF);
      P_TL_STAMP(pt_commit);
      issue(P, CF);  // (step s + 1; its registers are free)
      commit_weights(kind, W, s);
      P_TL_STAMP(pt_wts);
      issue_weights(P.kind, P.ph, P.slice, W);
      if (first && with_stats) __syncthreads();  // (the consumers' statistics hand-over of the previous item)
      P_TL_PROD_BAR();                            // step s is ready / step s - 1 has been consumed
      ++s;
    }
#endif
    if (with_stats && s > 0) __syncthreads();  // the last item's statistics hand-over
#ifdef P_TIMELINE
    if (p.dbg && tid == 256) {
      p.dbg[(int64_t)blockIdx.x * 8 + 5] = pt_work;
      p.dbg[(int64_t)blockIdx.x * 8 + 6] = pt_wait;
      p.dbg[(int64_t)blockIdx.x * 8 + 7] = pt_commit;  // (issue of the weights + wait for the halo + activation + LDS writes)
    }
#endif
    return;
  }

  // =============================================== CONSUMERS ===============================================
  const int li = lane & 31;  // MFMA row (A) / column (B, D)
  const int kg = lane >> 5;  // MFMA k-group: channels 8*kg .. 8*kg+7 of the chunk
  const int ys = li >> 3;
  // A addressing (conv_bf16t_kernel): MFMA row li of row tile mt of this wave: plane z = 2*wave + (mt>>1), y row 4*(mt&1) + ys,
  // x = li&7; the lane's 16-byte half sits in slot kg ^ (halo row parity) = kg ^ (ys&1) ^ (kh&1)
  const int a_vox = (((2 * wave) * P_H + ys) * P_H + (li & 7)) * P_RS;
  const int a_base0 = a_vox + ((kg ^ (ys & 1)) * 4);      // taps with even kh
  const int a_base1 = a_vox + ((kg ^ (ys & 1) ^ 1) * 4);  // taps with odd kh
  auto load_a = [&](float4 (&a)[4], int tap, const float* hb) {
    if (P_PROBE & 2) return;
    const int kd = tap / 9, kh = (tap - kd * 9) / 3, kw = tap - kd * 9 - kh * 3;
    const int toff = ((kd * P_H + kh) * P_H + kw) * P_RS;
    const int ab = (kh & 1) ? a_base1 : a_base0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      a[mt] = *reinterpret_cast<const float4*>(hb + ab + ((mt >> 1) * P_H * P_H + 4 * (mt & 1) * P_H) * P_RS + toff);
  };
  // skip steps: centre voxel v = ((2*wave + (mt>>1))*8 + 4*(mt&1) + ys)*8 + (li&7), row parity = ys&1
  const int as_base = (((2 * wave) * 8 + ys) * 8 + (li & 7)) * P_RS + ((kg ^ (ys & 1)) * 4);
  auto load_a_skip = [&](float4 (&a)[4], int kstep, const float* hb) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      a[mt] = *reinterpret_cast<const float4*>(hb + kstep * 4096 + as_base + ((mt >> 1) * 64 + 4 * (mt & 1) * 8) * P_RS);
  };
  // B operands: the step's weights in LDS, [entry][nt][lane][4 words]; entry = tap (main step) / k-step (skip step)
  auto load_b = [&](float4 (&b)[NT], int e) {
    if (P_PROBE & 1) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const float4*>(s_bw + (e * NT + nt) * 256 + lane * 4);
  };
  // progress of this wave through the weight buffer, for the producers that refill it (see commit_weights)
  auto publish = [&](int v) {
    P_COMPILER_FENCE();
    if (lane == 0) *reinterpret_cast<volatile int*>(s_flag + wave) = v;
    P_COMPILER_FENCE();
  };
  f32x16 acc[4][NT];
  auto mfma_tap = [&](const float4 (&a)[4], const float4 (&b)[NT]) {
    if (P_PROBE & 8) {  // (probe: the operands are kept alive, nothing is multiplied)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt][0] += a[mt].x + b[nt].x;
      return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = mfma_bf16_32x32x16(a[mt], b[nt], acc[mt][nt]);
  };

  float4 A[2][4];
  float4 B[3][NT];
  if (P_PROBE & 3) {  // (probes: the operands nobody loads hold something harmless)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) A[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) B[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#ifdef P_TIMELINE
  unsigned long long ct_wait = 0, ct_loop = 0, ct_epi = 0, ct_steps = 0, ct_items = 0, ct0, ct1;
#endif
  int s = 0;
  for (int item = it_first; item < it_end; item += it_step) {
    int n, tz0, ty0, tx0, slice;
    decode(item, n, tz0, ty0, tx0, slice);
    const int n0 = slice * BN;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    for (int st = 0; st < ncc; ++st, ++s) {  // ---- 16-channel chunks of the activated halo: 27 taps
#ifdef P_TIMELINE
      ct0 = HOLO_PROBE_CLOCK();
#endif
      P_STEP_BARRIER();  // step s is staged, halo and weights (and the producers may overwrite the halo buffer of step s - 1)
#ifdef P_TIMELINE
      ct1 = HOLO_PROBE_CLOCK();
      ct_wait += ct1 - ct0;
#endif
      const float* hb = s_halo + (s & 1) * P_BUF;
      load_b(B[0], 0);
      load_b(B[1], 1);
      load_a(A[0], 0, hb);
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        if (tap + 1 < 27) load_a(A[(tap + 1) & 1], tap + 1, hb);
        if (tap + 2 < 27) load_b(B[(tap + 2) % 3], tap + 2);
        mfma_tap(A[tap & 1], B[tap % 3]);
        {  // one operand read behind each MFMA (conv_bf16t_kernel's SCHED = 2)
#pragma unroll
          for (int i = 0; i < 4 + NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT - 4 - NT, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (the reads of taps <= 8 / <= 17 / all have been issued - B two taps ahead - and LDS serves them in order)
        if (!P_PHASED && (tap == 8 || tap == 17 || tap == 26)) publish(3 * s + (tap + 1) / 9);
      }
      if (P_PHASED) P_STEP_BARRIER();  // B: the producers may commit step s + 1
#ifdef P_TIMELINE
      ct_loop += HOLO_PROBE_CLOCK() - ct1;
      ++ct_steps;
#endif
    }
    if (SKIP) {
      for (int st = ncc; st < nst; ++st, ++s) {  // ---- 32 raw channels of the fused 1x1x1 skip connection: two k-steps
        P_STEP_BARRIER();
        const float* hb = s_halo + (s & 1) * P_BUF;
        load_b(B[0], 0);
        load_b(B[1], 1);
        load_a_skip(A[0], 0, hb);
        load_a_skip(A[1], 1, hb);
        mfma_tap(A[0], B[0]);
        mfma_tap(A[1], B[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (P_PHASED) {
          P_STEP_BARRIER();
        } else {
          publish(3 * s + 3);
        }
      }
    }

    // ---- epilogue (conv_bf16t_kernel's): each wave passes one 32-voxel row tile at a time through its own transposition
    // tile as fp32 [voxel][channel] and leaves with 8 channels of one voxel per lane: bias, residual, GroupNorm statistics
    // and the store are 16-byte operations.  D layout of 32x32: column = li (Cout), row i = (r&3) + 8*(r>>2) + 4*kg.
#ifdef P_TIMELINE
    ct0 = HOLO_PROBE_CLOCK();
#endif
    constexpr int LPV = BN / 8;      // lanes per voxel
    constexpr int VPP = 64 / LPV;    // voxels per pass
    constexpr int NPASS = 32 / VPP;
    static_assert(VPP % 8 == 0, "a pass covers whole x rows of the tile (uniform per-pass output offsets)");
    float* ep = s_ep + wave * (32 * P_EW);
    const int ch8 = (lane % LPV) * 8;
    const bool cvalid = n0 + ch8 < p.Cout;
    const int co8 = cvalid ? n0 + ch8 : 0;
    float bv[8], es[8], eq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = es[e] = eq[e] = 0.f;
    if (p.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co8), b1 = *reinterpret_cast<const float4*>(p.bias + co8 + 4);
      bv[0] = b0.x, bv[1] = b0.y, bv[2] = b0.z, bv[3] = b0.w, bv[4] = b1.x, bv[5] = b1.y, bv[6] = b1.z, bv[7] = b1.w;
    }
    if (SKIP && p.skip_bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.skip_bias + co8), b1 = *reinterpret_cast<const float4*>(p.skip_bias + co8 + 4);
      bv[0] += b0.x, bv[1] += b0.y, bv[2] += b0.z, bv[3] += b0.w, bv[4] += b1.x, bv[5] += b1.y, bv[6] += b1.z, bv[7] += b1.w;
    }
    const int i0 = lane / LPV;
    // (32-bit lane part: conv_plan keeps an output sample below 4 GB on this path; the sample base and the pass offset are scalars)
    const unsigned obase = (unsigned)((((tz0 + 2 * wave) * p.OH + ty0 + (i0 >> 3)) * p.OW + tx0 + (i0 & 7)) * p.Cout + co8);
    const int64_t nbase = (int64_t)n * p.OD * p.OH * p.OW * p.Cout;
    auto uoff = [&](int mt, int ps) {
      return nbase + (int64_t)(((mt >> 1) * p.OH + 4 * (mt & 1) + ((ps * VPP) >> 3)) * p.OW) * p.Cout;
    };
    // the residual of TWO row tiles is in flight at any time (all four at once pushed the epilogue into scratch memory)
    float4 res_q[2][NPASS];
    const bool with_res = p.residual != nullptr;  // (uniform)
    auto res_load = [&](int mt) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps)
        res_q[mt & 1][ps] = *reinterpret_cast<const float4*>(reinterpret_cast<const uint16_t*>(p.residual) + uoff(mt, ps) + obase);
    };
    if (with_res) {
      res_load(0);
      res_load(1);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * kg) * P_EW + nt * 32 + li] = acc[mt][nt][r];
      HOLO_WAVE_SYNC();  // (the transposition tile is private to the wave)
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int64_t o = uoff(mt, ps) + obase;
        const int i = ps * VPP + lane / LPV;
        const float4 v0 = *reinterpret_cast<const float4*>(ep + i * P_EW + ch8);
        const float4 v1 = *reinterpret_cast<const float4*>(ep + i * P_EW + ch8 + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (with_res) {
          float rf[8];
          unpack8(res_q[mt & 1][ps], rf);
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
          if (p.out_bf16) {
            *reinterpret_cast<float4*>(reinterpret_cast<uint16_t*>(p.out) + o) = pack8(v);
          } else {
            *reinterpret_cast<float4*>(p.out + o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(p.out + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
          }
        }
      }
      if (with_res && mt + 2 < 4) res_load(mt + 2);
      HOLO_WAVE_SYNC();  // the tile is overwritten by the next row tile
    }
    // GroupNorm statistics of the tensor just produced: one slab per tile (512 voxels) -> stats[n][tile][Cout][2]; the four
    // waves' sums meet in LDS in a fixed order (deterministic).  One barrier of the whole workgroup (the producers take part
    // in it in front of the next item's first hand-over).
    if (with_stats) {
      const int slab = (item / nslices) % tiles_per_sample;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int m = LPV; m < 64; m <<= 1) {
          es[e] += __shfl_xor(es[e], m);
          eq[e] += __shfl_xor(eq[e], m);
        }
      }
      if (lane < LPV) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s_stat[(wave * LPV + lane) * 16 + e] = es[e];
          s_stat[(wave * LPV + lane) * 16 + 8 + e] = eq[e];
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
            s1 += s_stat[(w * LPV + lane) * 16 + e];
            s2 += s_stat[(w * LPV + lane) * 16 + 8 + e];
          }
          d[2 * e] = (double)s1;
          d[2 * e + 1] = (double)s2;
        }
      }
    }
#ifdef P_TIMELINE
    ct_epi += HOLO_PROBE_CLOCK() - ct0;
    ++ct_items;
#endif
  }
#ifdef P_TIMELINE
  if (p.dbg && tid == 0) {
    unsigned long long* d = p.dbg + (int64_t)blockIdx.x * 8;
    d[0] = ct_wait, d[1] = ct_loop, d[2] = ct_epi, d[3] = ct_steps, d[4] = ct_items;
  }
#endif
}

}  // namespace

// Launch of the persistent wave-specialised kernel: p.grid_x workgroups of 512 threads (conv_plan: one per CU, a multiple of 8).
int P_ENTRY(const ConvParams& p, void* stream) {
  if (!p.in_bf16 || (p.residual && !p.res_bf16) || p.nsplit != 1 || !p.w_bft || (p.skip_w && !p.skip_w_bft)) {
    set_error("conv_bf16p_launch: bf16 activation storage, prepared wide-tile weights, no split-K");
    return -1;
  }
  const dim3 grid((unsigned)p.grid_x), block(512);
  const bool sk = p.skip_w != nullptr;
  if (p.Cout < 64) {
    if (sk) {
      set_error("conv_bf16p_launch: a fused skip needs Cout >= 64");
      return -1;
    }
    HOLO_LAUNCH((conv_bf16p_kernel<1, false>), grid, block, stream, p);
  } else if (sk) {
    HOLO_LAUNCH((conv_bf16p_kernel<2, true>), grid, block, stream, p);
  } else {
    HOLO_LAUNCH((conv_bf16p_kernel<2, false>), grid, block, stream, p);
  }
  return 0;
}

}  // namespace holo
