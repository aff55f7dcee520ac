// kernels_attn_bf16.hip — the bf16 attention of the bf16 storage mode (long sequences): attn_pack_kernel +
// flash_attn_bf16v2_kernel + attn_combine_kernel.  Replaces QKVAttentionLegacy
// (holo_diffusion/guided_diffusion/unet.py:436-455) on the 32^3 / 16^3 levels of the 128^3 net.
//
// Compiled TWICE (Makefile): as itself - pack, combine, the exact loop, the launch code - and, through
// kernels_attn_bf16_lazy.hip (-DHOLO_ATTN_LAZY_TU), the LAZY kernels alone with -fno-slp-vectorize: the SLP vectoriser pairs
// the softmax's independent additions / multiplications into v_pk_add_f32 / v_pk_mul_f32, and packed fp32 instructions do
// NOT ride in the shadow of a bf16 MFMA the way plain vector instructions do (tools/mfma_shadow_probe -DPROBE_BF16,
// profiles/r06_mfma_shadow_probe_bf16.txt: one v_pk_add_f32 per MFMA costs 17 cycles, four v_add_f32 cost 3).  Measured on
// the T = 32 768 call (tools/attn_probe): LAZY kernel 593 us without / 653 us with the vectoriser; the exact loop, whose
// softmax runs in clumps BETWEEN its MFMAs, the other way round (687 us with, 814 us without).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {
namespace {

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
  int* redo;      // [workgroups]: set by the LAZY kernel for a workgroup that left its range, read by the exact kernel behind it
  int N, T, C, H, ksplit, out_bf16;
};

#ifndef HOLO_EMU
// The LAZY body's MFMAs as inline asm, because the register CLASS of an accumulator decides whether the vector unit can work
// beside the matrix pipe: v_accvgpr_read_b32 waits for the MFMA in flight (tools/mfma_shadow_probe -DPROBE_BF16: ONE per MFMA
// takes it from 35 to 84 cycles), so accumulators that vector code reads (S) must sit in architectural registers while
// those only MFMAs touch (O) can take the AGPR half of the file - and the compiler picks ONE class for all MFMAs of a
// kernel.  Hazards the compiler cannot see inside an asm statement are padded by hand: s_nop 1 in front of each MFMA (its
// operands may come from vector instructions), attn_mfma_drain() before anything but an MFMA reads an accumulator.
__device__ __forceinline__ void attn_mfma_s(f32x16& c, const float4& a, const float4& b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
               : "+v"(c)
               : "v"(__builtin_bit_cast(f32x4, a)), "v"(__builtin_bit_cast(f32x4, b)));
}
__device__ __forceinline__ void attn_mfma_s0(f32x16& c, const float4& a, const float4& b) {  // accumulator input: the constant 0
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0"
               : "=v"(c)
               : "v"(__builtin_bit_cast(f32x4, a)), "v"(__builtin_bit_cast(f32x4, b)));
}
__device__ __forceinline__ void attn_mfma_o(f32x16& c, const float4& a, const float4& b) {
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
               : "+a"(c)
               : "v"(__builtin_bit_cast(f32x4, a)), "v"(__builtin_bit_cast(f32x4, b)));
}
__device__ __forceinline__ void attn_mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }
__device__ __forceinline__ void attn_mfma_s_pad() { asm volatile("s_nop 3"); }
#else
static inline void attn_mfma_s(f32x16& c, const float4& a, const float4& b) { c = mfma_bf16_32x32x16(a, b, c); }
static inline void attn_mfma_o(f32x16& c, const float4& a, const float4& b) { c = mfma_bf16_32x32x16(a, b, c); }
static inline void attn_mfma_s0(f32x16& c, const float4& a, const float4& b) {
  f32x16 z;
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  c = mfma_bf16_32x32x16(a, b, z);
}
static inline void attn_mfma_drain() {}
static inline void attn_mfma_s_pad() {}
#endif

// QT: 32-query tiles per wave (2; 1 for head channels 128, whose 64-query wave tile would need 320 registers)
#ifndef HOLO_ATTN_LAZY
#define HOLO_ATTN_LAZY 1
#endif
#ifndef HOLO_ATTN_PROBE  // development probes of the LAZY loop (timing only, results wrong): 1 no exponentials, 2 no barrier, 4 no staging, 8 no additions / packing
#define HOLO_ATTN_PROBE 0
#endif
// MODE 0: the exact loop; 1: the LAZY pass alone (a workgroup that leaves its range writes nothing but its redo flag);
// 2: the exact loop for the workgroups whose redo flag is set (launched behind a MODE 1 kernel).  Two kernels rather than
// two loops in one: the compiler picks ONE split of the register file between architectural registers and AGPRs per
// kernel, and the exact loop's 128 accumulator AGPRs would leave the LAZY loop 128 registers (800 bytes of scratch).
template <int CH, int QT, int MODE, bool PIPE = true>
__global__ __launch_bounds__(256, MODE == 1 ? 1 : 2) void flash_attn_bf16v2_kernel(AttnV2 p) {
  static_assert(MODE != 1 || QT == 2, "the LAZY pass is written for two query tiles per wave");
  constexpr bool LAZY = MODE == 1;
  constexpr int KB = 64;
  constexpr int KW = CH / 2 + 4;  // words per K row
  constexpr int VW = 34;          // words per V^T row (64 keys + 8 bytes)
  constexpr int NKS = CH / 16;    // k-steps of S^T
  constexpr int NCT = CH / 32;    // channel tiles of O^T
  constexpr int PER = CH / 32;    // 16-byte staging pieces per thread, for K and for V^T
  __shared__ __attribute__((aligned(16))) uint32_t s_k[2][KB * KW];
  __shared__ __attribute__((aligned(16))) uint32_t s_v[2][CH * VW];
  __shared__ int s_redo;

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

  f32x4 kreg2[PER], vreg2[PER];  // (second set: the LAZY pass requests two blocks ahead)
  f32x4 kreg[PER], vreg[PER];  // (native vectors: as float4 structs one of the two arrays stayed in scratch memory, and a
                               //  scratch store of a load still in flight stalls the wave for the whole round trip)
  auto stage_load_to = [&](int blk, f32x4 (&kr)[PER], f32x4 (&vr)[PER]) {
    const int k0 = kbeg + blk * KB;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid + 256 * j;
      const int key = i / (CH / 8), c8 = i - key * (CH / 8);
      kr[j] = *reinterpret_cast<const f32x4*>(p.kb + (hb * p.T + k0 + key) * CH + c8 * 8);
      const int ch = i >> 3, k8 = i & 7;
      vr[j] = *reinterpret_cast<const f32x4*>(p.vt + (hb * CH + ch) * p.T + k0 + k8 * 8);
    }
  };
  auto stage_store_from = [&](int buf, const f32x4 (&kr)[PER], const f32x4 (&vr)[PER]) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid + 256 * j;
      const int key = i / (CH / 8), c8 = i - key * (CH / 8);
      *reinterpret_cast<f32x4*>(&s_k[buf][key * KW + c8 * 4]) = kr[j];
      const int ch = i >> 3, k8 = i & 7;
      uint32_t* d = &s_v[buf][ch * VW + k8 * 4];
      *reinterpret_cast<uint2*>(d) = make_uint2(__float_as_uint(vr[j][0]), __float_as_uint(vr[j][1]));
      *reinterpret_cast<uint2*>(d + 2) = make_uint2(__float_as_uint(vr[j][2]), __float_as_uint(vr[j][3]));
    }
  };
  auto stage_load = [&](int blk) { stage_load_to(blk, kreg, vreg); };
  auto stage_load2 = [&](int blk) { stage_load_to(blk, kreg2, vreg2); };
  auto stage_store = [&](int buf) { stage_store_from(buf, kreg, vreg); };
  auto stage_store2 = [&](int buf) { stage_store_from(buf, kreg2, vreg2); };
  // ---- the query tiles' rows: D rows = channels ct*32 + (r&3) + 8(r>>2) + 4kg, column = query li
  // (m: exponent reference, l: the query's sum over BOTH halves of its keys)
  auto write_out = [&](const f32x16 (&oacc)[NCT][QT], const float (&m_run)[QT], const float (&l_run)[QT]) {
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
  };

  if constexpr (LAZY) {
    // ---- LAZY pass: the whole key range with NO exponent reference: P = 2^S as it comes out of the MFMAs (the scores are
    // already in the exp2 domain and any common reference cancels in O / l), no maximum, no re-referencing, no branch.
    // That is exact as long as the scores of a query stay inside (-100, +100) (e^-69 .. e^+69 against e^0: every
    // attention map the released nets produce); a sum that is not below 2^100 or not above 2^-100 at the end - overflow,
    // NaN, or everything underflowed - makes the WORKGROUP repeat its range with the exact loop below.
    // With no maximum to wait for, the loop body is ONE basic block, hand scheduled, staggered over the two 32-KEY HALVES
    // of a block instead of the two query tiles:
    //   S(k0) | S(k1) + exponentials(k0) | PV(k0) + exponentials(k1) | PV(k1)
    // (S(k) = both query tiles against key half k: 8 MFMAs; PV(k) likewise).  A key half's K fragments die with its 8
    // MFMAs and each V^T fragment is read ONCE and multiplied into both query tiles back to back (half the V^T traffic of
    // the exact loop).  The order below IS the schedule: every slice is fenced with sched_barrier (left to the scheduler
    // the body came out as 16 MFMAs in a row, the exponentials in a clump behind them, and 150 - 600 bytes of scratch per
    // lane).  A slice behind one MFMA = 4 v_exp_f32 + their additions + packing: what fits the shadow of one bf16 MFMA
    // (profiles/r06_mfma_shadow_probe_bf16.txt).  ONE wave per SIMD (the O accumulators take the AGPR half of the file),
    // so the staging registers run a block AHEAD of the LDS buffers: block b + 2 is requested at the top of block b.
    // The O accumulators of this pass are touched by NOTHING but the asm MFMAs inside the loop: any other definition
    // (a rescaling branch, the exact loop sharing them) made the register allocator copy all 64 of them AGPR -> AGPR at
    // the top of every block, and an AGPR access next to a running MFMA costs ~50 cycles (3 750 cycles per block
    // instead of ~1 300).
    f32x16 oacc[NCT][QT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[ct][qt][r] = 0.f;
    float lsum[QT][4];  // (four chains per query tile: a single one is 64 DEPENDENT additions per block, ~8 cycles each with one wave per SIMD)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) lsum[qt][r] = 0.f;
    if (tid == 0) s_redo = 0;
    stage_load(0);
    stage_store(0);
    stage_load(nblk > 1 ? 1 : 0);
    stage_load2(nblk > 2 ? 2 : nblk - 1);
    __syncthreads();
    constexpr int NS = NKS * QT;      // MFMAs of S(k)
    constexpr int NP = 2 * NCT * QT;  // MFMAs of PV(k)
    constexpr int NE = 4 * QT;        // exponential slices of a key half
    static_assert(NE % NS == 0 || NS % NE == 0, "slices per MFMA");
    static_assert(NE % NP == 0 || NP % NE == 0, "slices per MFMA");
    auto lazy_block = [&](int blk, auto odd) {
      constexpr bool ODD = decltype(odd)::value;  // (which staging register set holds block b + 1)
      const int buf = blk & 1;
      f32x16 sacc[2][QT];
      float4 kh[2][NKS];
      float4 ph[2][QT][2];  // [key half][qt][h]
      float pe[QT][8];      // exponentials waiting to be packed
      float4 vfa[2 * NCT], vfb[2 * NCT];
      // slice i (0 .. 4*QT-1) of a key half's exponentials: registers 4*(i&3) .. +3 of query tile i>>2; the packed operand
      // of 8 registers follows their second slice
      auto exp_slice = [&](int kt, int i) {
        const int qt = i >> 2, r0 = 4 * (i & 3), e0 = r0 & 4;
        if ((i & 3) == 0) attn_mfma_s_pad();  // (the S MFMAs are asm: their write-back is not the compiler's to wait for)
#pragma unroll
        for (int r = 0; r < 4; ++r) pe[qt][e0 + r] = (HOLO_ATTN_PROBE & 1) ? sacc[kt][qt][r0 + r] : holo_exp2(sacc[kt][qt][r0 + r]);
        if (!(HOLO_ATTN_PROBE & 8)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) lsum[qt][r] += pe[qt][e0 + r];
        }
        if ((i & 1) != 0 && (HOLO_ATTN_PROBE & 8) != 0)
          ph[kt][qt][(i & 3) >> 1] = make_float4(pe[qt][0], pe[qt][2], pe[qt][4], pe[qt][6]);
        else if (i & 1)
          ph[kt][qt][(i & 3) >> 1] =
              make_float4(__uint_as_float(pack_bf16x2(pe[qt][0], pe[qt][1])), __uint_as_float(pack_bf16x2(pe[qt][2], pe[qt][3])),
                          __uint_as_float(pack_bf16x2(pe[qt][4], pe[qt][5])), __uint_as_float(pack_bf16x2(pe[qt][6], pe[qt][7])));
      };
      auto k_frag = [&](int kt, int s) { return *reinterpret_cast<const float4*>(&s_k[buf][(kt * 32 + li) * KW + s * 8 + kg * 4]); };
      auto v_frag = [&](int kt, int j) {  // fragment j = (h, ct) of key half kt
        const int h = j / NCT, ct = j - h * NCT;
        const uint32_t* vr = &s_v[buf][(ct * 32 + li) * VW + (kt * 32 + 16 * h + 4 * kg) / 2];
        const uint2 lo = *reinterpret_cast<const uint2*>(vr), hi = *reinterpret_cast<const uint2*>(vr + 4);
        return make_float4(__uint_as_float(lo.x), __uint_as_float(lo.y), __uint_as_float(hi.x), __uint_as_float(hi.y));
      };
      auto s_mfma = [&](int kt, int i) {
        const int s = i / QT, qt = i - s * QT;
        if (s == 0)
          attn_mfma_s0(sacc[kt][qt], kh[kt][s], qf[qt][s]);  // (accumulator input: the constant 0)
        else
          attn_mfma_s(sacc[kt][qt], kh[kt][s], qf[qt][s]);
      };
      // ---- phase 1: S(k0); a k-step's K fragment of k1 is requested when k0's has served its MFMAs.  Behind the K
      // fragments: block b + 1 (requested a block ago) goes from the staging registers to the other buffer, block b + 2 is
      // requested
#pragma unroll
      for (int s = 0; s < NKS; ++s) kh[0][s] = k_frag(0, s);
      if (!(HOLO_ATTN_PROBE & 4)) {  // block b + 1 (requested two blocks ago) -> the other buffer; block b + 3 is requested
        const int nb = blk + 3 < nblk ? blk + 3 : nblk - 1;
        if (ODD) {
          stage_store2(buf ^ 1);
          stage_load2(nb);
        } else {
          stage_store(buf ^ 1);
          stage_load(nb);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        s_mfma(0, i);
        if (i % QT == QT - 1) kh[1][i / QT] = k_frag(1, i / QT);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- phase 2: S(k1) + exponentials(k0); the V^T fragments of PV(k0) are requested behind its last MFMAs
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        s_mfma(1, i);
#pragma unroll
        for (int e = i * NE / NS; e < (i + 1) * NE / NS; ++e) exp_slice(0, e);
#pragma unroll
        for (int j = 0; j < 2 * NCT; ++j)
          if (i == NS - 2 * NCT + j) vfa[j] = v_frag(0, j);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- phase 3: PV(k0) + exponentials(k1)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int j = i / QT, qt = i - j * QT, h = j / NCT, ct = j - h * NCT;
        attn_mfma_o(oacc[ct][qt], vfa[j], ph[0][qt][h]);
#pragma unroll
        for (int e = i * NE / NP; e < (i + 1) * NE / NP; ++e) exp_slice(1, e);
#pragma unroll
        for (int jj = 0; jj < 2 * NCT; ++jj)
          if (i == NP - 2 * NCT + jj) vfb[jj] = v_frag(1, jj);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- phase 4: PV(k1)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int j = i / QT, qt = i - j * QT, h = j / NCT, ct = j - h * NCT;
        attn_mfma_o(oacc[ct][qt], vfb[j], ph[1][qt][h]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(HOLO_ATTN_PROBE & 2)) __syncthreads();
    };
    for (int blk = 0; blk < nblk; blk += 2) {  // (nblk is even: the key range of a workgroup is a multiple of 128)
      lazy_block(blk, std::false_type{});
      lazy_block(blk + 1, std::true_type{});
    }
    attn_mfma_drain();
    bool bad = false;
    float ltot[QT], mzero[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      ltot[qt] = holo_add_xor32((lsum[qt][0] + lsum[qt][1]) + (lsum[qt][2] + lsum[qt][3]));  // the two halves of a query's keys
      mzero[qt] = 0.f;
      bad = bad || !(ltot[qt] < 0x1p100f) || !(ltot[qt] > 0x1p-100f);
    }
#ifdef HOLO_ATTN_NO_REDO  // (development probe: what the fallback pass costs)
    bad = false;
#endif
    if (bad) s_redo = 1;
    __syncthreads();
    if (tid == 0) p.redo[blockIdx.x] = s_redo;
    if (!s_redo) write_out(oacc, mzero, ltot);
  } else {
  if (MODE == 2 && !p.redo[blockIdx.x]) return;  // (uniform)

  // ---- EXACT loop (kernels without LAZY; the fallback of a workgroup whose LAZY pass left its range).
  // S^T tiles [key tile kt][query tile qt]; online softmax per query column (exp2 domain), P^T operands straight from the
  // registers; O^T += V^T . P^T.  The exponent reference m_ref of a query is NOT its exact running maximum: S - m_ref comes
  // out of the MFMAs (the accumulators start at -m_ref), P = 2^(S - m_ref) may exceed 1, and O, l are re-referenced only
  // when the running maximum has moved more than 2^32 away from m_ref (or in the first block) - any common reference
  // cancels in O / l.  That takes the per-element subtraction and, almost always, the rescaling of O out of the vector work.
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
  stage_load(0);
  stage_store(0);
  __syncthreads();
  for (int blk = 0; blk < nblk; ++blk) {
    const int buf = blk & 1;
    // The two query tiles are staggered so that the vector work of one tile's softmax sits between the MFMAs of the
    // other tile: S(0) | S(1) + softmax(0) | PV(0) + softmax(1) | PV(1).  (Measured alternatives, all within noise or
    // worse: the plain sequential order; sharing the V^T fragments too; the minimal-register sequential form at three
    // waves per SIMD: 27 % slower; an XCD-aware workgroup order (each XCD one K / V^T stream): 7 % slower.)
    f32x16 sacc[2][QT];
    float4 pf[QT][2][2];  // [qt][kt][h]
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
  write_out(oacc, m_run, l_run);
  }  // (MODE != 1)
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

#ifdef HOLO_ATTN_LAZY_TU
// ---- the LAZY kernels' translation unit: nothing but their launcher
int flash_attn_bf16v2_lazy_launch(const void* attn_v2, int ch, unsigned grid, void* stream) {
  const AttnV2& a = *reinterpret_cast<const AttnV2*>(attn_v2);
  if (ch == 32)
    HOLO_LAUNCH((flash_attn_bf16v2_kernel<32, 2, 1>), dim3(grid), dim3(256), stream, a);
  else
    HOLO_LAUNCH((flash_attn_bf16v2_kernel<64, 2, 1>), dim3(grid), dim3(256), stream, a);
  return 0;
}
#else
int flash_attn_bf16v2_lazy_launch(const void* attn_v2, int ch, unsigned grid, void* stream);  // (kernels_attn_bf16_lazy.hip)

bool flash_attn_bf16v2_supported(int T, int ch) { return (T % 256) == 0 && (ch == 32 || ch == 64 || ch == 128); }
static int attn_v2_ksplit(const AttnParams& p, int num_cus) {
  const int64_t wgs = (int64_t)p.N * p.H * (p.T / (p.C / p.H == 128 ? 128 : 256));
  int ks = 1;
  // two workgroups per CU for the kernels that run two waves per SIMD; the LAZY kernel (head channels 32 / 64) runs one
  const int per_cu = 2;
  while (wgs * ks < per_cu * (int64_t)num_cus && ks < 8 && (p.T / (ks * 2)) % 128 == 0) ks *= 2;  // (an even number of 64-key blocks per split: the LAZY loop takes two per trip)
  if (const char* e = getenv("HOLO_FLASH_V2_KSPLIT")) {  // development knob
    const int v = atoi(e);
    if (v >= 1 && v <= 8 && (p.T / v) % 128 == 0) ks = v;
  }
  return ks;
}
size_t flash_attn_bf16v2_workspace_bytes(const AttnParams& p, int num_cus) {
  const size_t ntc = (size_t)p.N * p.T * p.C;
  const int ks = attn_v2_ksplit(p, num_cus);
  size_t b = 3 * ntc * sizeof(uint16_t);
  if (ks > 1) b += (size_t)ks * ntc * sizeof(float) + (size_t)ks * p.N * p.H * p.T * 2 * sizeof(float);
  b += (size_t)p.N * p.H * ks * (p.T / 128) * sizeof(int);  // the LAZY kernel's redo flags, one per workgroup
  return b;
}
void flash_attn_bf16v2_operands(const AttnParams& p, void* work, uint16_t** q, uint16_t** k, uint16_t** vt, float* qscale) {
  const size_t ntc = (size_t)p.N * p.T * p.C;
  uint16_t* w16 = reinterpret_cast<uint16_t*>(work);
  *q = w16, *k = w16 + ntc, *vt = w16 + 2 * ntc;
  *qscale = p.scale2 * 1.4426950408889634f;  // softmax in the exp2 domain
}

int flash_attn_bf16v2_launch(const AttnParams& p, void* work, int out_bf16, int num_cus, void* stream, int packed) {
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
  a.redo = reinterpret_cast<int*>(a.ksplit > 1 ? a.ml + (size_t)a.ksplit * p.N * p.H * p.T * 2 : a.opart);
  a.N = p.N, a.T = p.T, a.C = p.C, a.H = p.H;
  a.out_bf16 = out_bf16;
  const float qscale = p.scale2 * 1.4426950408889634f;  // softmax in the exp2 domain
  dim3 pgrid((unsigned)((int64_t)p.N * p.H * (p.T / 64)));
  dim3 grid((unsigned)((int64_t)p.N * p.H * a.ksplit * (p.T / (ch == 128 ? 128 : 256))));
  // head channels 32 / 64: the LAZY kernel, then the exact one for the workgroups it flagged (normally none: they return at once)
  auto pack = [&](auto chc) {
    constexpr int CHC = decltype(chc)::value;
    if (!packed) HOLO_LAUNCH(attn_pack_kernel<CHC>, pgrid, dim3(256), stream, p.qkv, w16, w16 + ntc, w16 + 2 * ntc, p.T, p.C, p.H, qscale);
  };
  const bool lazy = HOLO_ATTN_LAZY != 0 && ch != 128 && !getenv("HOLO_ATTN_EXACT");  // (development knob: the exact loop alone)
  switch (ch) {
    case 32:
      pack(std::integral_constant<int, 32>{});
      if (lazy) {
        flash_attn_bf16v2_lazy_launch(&a, 32, grid.x, stream);
        HOLO_LAUNCH((flash_attn_bf16v2_kernel<32, 2, 2>), grid, dim3(256), stream, a);
      } else {
        HOLO_LAUNCH((flash_attn_bf16v2_kernel<32, 2, 0>), grid, dim3(256), stream, a);
      }
      break;
    case 64:
      pack(std::integral_constant<int, 64>{});
      if (lazy) {
        flash_attn_bf16v2_lazy_launch(&a, 64, grid.x, stream);
        HOLO_LAUNCH((flash_attn_bf16v2_kernel<64, 2, 2>), grid, dim3(256), stream, a);
      } else {
        HOLO_LAUNCH((flash_attn_bf16v2_kernel<64, 2, 0>), grid, dim3(256), stream, a);
      }
      break;
    default:
      pack(std::integral_constant<int, 128>{});
      HOLO_LAUNCH((flash_attn_bf16v2_kernel<128, 1, 0>), grid, dim3(256), stream, a);
      break;
  }
  if (a.ksplit > 1) {
    const int64_t total = (int64_t)p.N * p.T * (p.C / 4);
    HOLO_LAUNCH(attn_combine_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), stream, a, ch);
  }
  return 0;
}

#endif  // HOLO_ATTN_LAZY_TU

}  // namespace holo
