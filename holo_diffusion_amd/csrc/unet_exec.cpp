// unet_exec.cpp — native runtime of the denoiser: builds the static launch plan of
// UNetModel.forward (holo_diffusion/guided_diffusion/unet.py:800-837, block construction :645-798 as
// configured by SimpleUnet3D, holo_diffusion/utils/diffusion_utils.py:56-75) and replays it on a stream.
//
// Everything is static for a given (config, batch): tensor shapes, workspace offsets, split-K factors,
// kernel parameters.  The plan is a flat vector of ops; forward() only patches the three caller pointers
// (x, timesteps, y) and launches.  No allocation, no host synchronisation inside forward().
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/holo_abi.h"
#include "holo_common.h"
#include "holo_kernels.h"

namespace holo {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return HOLO_E_HIP;                                                               \
    }                                                                                  \
  } while (0)

}  // namespace holo

using namespace holo;

// struct HoloCtx { device, num_cus }: holo_kernels.h (shared with render_exec.cpp)

// ---------------------------------------------------------------------------------------------
// structure description
// ---------------------------------------------------------------------------------------------
namespace {

enum BlockKind { B_CONV, B_RES, B_ATTN, B_DOWN, B_UP };
struct Block {
  BlockKind kind;
  std::string prefix;
  int cin, cout;
};

enum ParamKind { P_PLAIN, P_CONV3, P_CONV1, P_EMB_W, P_EMB_B };

// packed conv weights are zero padded to [taps][CoutP][CinP]: CoutP a multiple of the kernel's Cout tile
// (64 when Cout >= 64, else 32), CinP a multiple of the 32-channel K chunk
static inline int pad_cout(int c) { return c >= 64 ? (c + 63) / 64 * 64 : 32; }
static inline int pad_cin(int c) { return (c + 31) / 32 * 32; }
struct ParamSlot {
  std::string name;
  std::vector<int64_t> shape;
  int64_t numel;
  ParamKind kind;
  float* priv;  // private (repacked) device copy
  uint16_t* priv_bf = nullptr;  // conv weights: bf16 (RNE) copy packed for v_mfma_f32_16x16x32_bf16
  float* priv_wino = nullptr;   // conv weights of the wide top levels: Winograd-in-depth pseudo-taps (conv_wino_kernel)
  float* priv_wino2 = nullptr;  // ... and the (z,y) Winograd pseudo-taps (conv_wino2_kernel)
  float* priv_wino3 = nullptr;  // ... and the F(2x2x2, 3x3x3) pseudo-taps (conv_wino3_kernel)
  bool set;
};

struct Arena {
  size_t top = 0, peak = 0;
  bool keep = false;
  std::vector<std::pair<size_t, size_t>> fl;  // free list (off, size), sorted by off
  static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }
  size_t alloc(size_t bytes) {
    bytes = al(bytes);
    for (size_t i = 0; i < fl.size(); ++i) {
      if (fl[i].second >= bytes) {
        size_t off = fl[i].first;
        if (fl[i].second == bytes)
          fl.erase(fl.begin() + i);
        else {
          fl[i].first += bytes;
          fl[i].second -= bytes;
        }
        return off;
      }
    }
    size_t off = top;
    top += bytes;
    if (top > peak) peak = top;
    return off;
  }
  void free(size_t off, size_t bytes) {
    if (keep) return;
    bytes = al(bytes);
    size_t i = 0;
    while (i < fl.size() && fl[i].first < off) ++i;
    fl.insert(fl.begin() + i, std::make_pair(off, bytes));
    // coalesce
    if (i + 1 < fl.size() && fl[i].first + fl[i].second == fl[i + 1].first) {
      fl[i].second += fl[i + 1].second;
      fl.erase(fl.begin() + i + 1);
    }
    if (i > 0 && fl[i - 1].first + fl[i - 1].second == fl[i].first) {
      fl[i - 1].second += fl[i].second;
      fl.erase(fl.begin() + i);
    }
    if (!fl.empty() && fl.back().first + fl.back().second == top) {
      top = fl.back().first;
      fl.pop_back();
    }
  }
};

struct Act {  // channels-last activation [N][R][R][R][C]
  size_t off = 0, bytes = 0;
  size_t stats_off = 0, stats_bytes = 0;  // GroupNorm partial sums [N][stats_B][C][2] doubles
  int stats_B = 0;
  int C = 0, R = 0;
  int refs = 0;
};

struct Tape {  // one layer of the training forward (what its backward needs)
  int kind = 0;  // BlockKind, or 100 = the output head (GroupNorm + SiLU + conv), 101 = input conv
  Block b;
  Act x0, x1, h1, out;
  bool has_x1 = false;
  size_t coefA = 0, coefB = 0, momA = 0, momB = 0;
  const float* film = nullptr;
  size_t qkv = 0, a = 0;
  bool has_skip = false;
};

enum OpKind { OP_MEMSET, OP_IN, OP_TEMB, OP_EMBLIN, OP_STATS, OP_FINAL, OP_CONV, OP_GEMM, OP_SOFTMAX, OP_FLASH, OP_OUT };
struct Op {
  OpKind kind;
  ConvParams conv;
  GemmParams gemm;
  AttnParams attn;
  // generic
  const float* f0 = nullptr;
  const float* f1 = nullptr;
  const float* f2 = nullptr;
  const float* f3 = nullptr;
  const float* f4 = nullptr;
  float* o0 = nullptr;
  float* o1 = nullptr;
  const double* d0 = nullptr;
  const double* d1 = nullptr;
  double* dout = nullptr;
  int i0 = 0, i1 = 0, i2 = 0, i3 = 0, i4 = 0, i5 = 0;
  int64_t l0 = 0, l1 = 0;
  size_t bytes = 0;
};

}  // namespace

struct HoloUnet {
  HoloCtx* ctx;
  HoloUnetCfg cfg;
  std::vector<std::vector<Block>> inputs, outputs;
  std::vector<Block> middle;
  int final_ch;
  int ted;
  std::vector<ParamSlot> params;
  std::map<std::string, int> pindex;
  float* pstore = nullptr;  // one allocation for all private parameter copies
  uint16_t* pstore_bf = nullptr;                       // bf16 copies of the conv weights
  std::map<const float*, const uint16_t*> bf_of;       // fp32 private copy -> bf16 copy
  std::map<const float*, const uint16_t*> bft_of;      // fp32 private copy -> bf16 copy packed for the wide-tile kernel
  float* pstore_wino = nullptr;                        // Winograd-in-depth copies (36 / 2 pseudo-taps)
  std::map<const float*, const float*> wino_of;        // fp32 private copy -> Winograd copy
  std::map<const float*, const float*> wino2_of;       // fp32 private copy -> (z,y) Winograd copy
  std::map<const float*, const float*> wino3_of;       // fp32 private copy -> F(2x2x2, 3x3x3) Winograd copy
  std::map<std::string, float*> dgrad_wino, dgrad_wino2, dgrad_wino3;  // Winograd copies of the transposed (dgrad) weights
  // holo_unet_set_compute_dtype: 0 exact fp32 MFMA; 1 bf16: activations stored as bf16 in HBM, bf16 products with fp32
  // accumulation in the 3x3x3 convolutions and the long-sequence attention, fp32 GroupNorm statistics; 2 bf16x3 split
  // (fp32 storage, fp32-accurate)
  int compute_mode = 0;
  // concatenated emb_layers
  int emb_rows = 0;
  std::map<std::string, int> emb_row_off;  // resblock prefix -> first row
  float* emb_w = nullptr;                  // [emb_rows][ted]
  float* emb_b = nullptr;                  // [emb_rows]
  bool keep_intermediates = false;
  // plan
  int plan_batch = -1;
  void* plan_ws = nullptr;
  std::vector<Op> ops;
  size_t ws_need = 0;
  std::map<std::string, Act> block_outputs;
  std::map<int, size_t> ws_cache;
  // ---- training (holo_unet_backward): weights of the transposed convolutions, packed like the forward ones
  // ([Cin][Cout] flipped taps for the stride-1 convs; [tap][Cout][Cin] for the stride-2 Downsample convs), supplied by
  // holo_unet_set_dgrad_weight; the training plan (forward with every intermediate kept + backward op list)
  std::map<std::string, float*> dgrad_w;
  float* dgrad_tmp = nullptr;
  size_t dgrad_tmp_floats = 0;
  int tplan_batch = -1;
  void* tplan_ws = nullptr;
  std::vector<Op> tops;                                      // training forward
  std::vector<std::function<int(void*)>> bops;               // backward, in execution order
  std::vector<size_t> grad_off;                              // per parameter: byte offset of its gradient in the workspace
  size_t tws_need = 0;
  size_t gy_off = 0, gx_off = 0, y_off = 0;
  const int64_t* t_dev = nullptr;                            // timesteps of the running call (time_embed backward)
  bool tape_valid = false;                                    // holo_unet_forward_train ran and nothing has consumed its tape
  std::map<int, size_t> tws_cache;
};

namespace {

void build_structure(HoloUnet* u) {
  const HoloUnetCfg& c = u->cfg;
  const int mc = c.model_channels;
  int ch = c.channel_mult[0] * mc;
  char buf[128];
  u->inputs.clear();
  u->outputs.clear();
  u->middle.clear();
  u->inputs.push_back({Block{B_CONV, "input_blocks.0.0", c.in_channels, ch}});
  std::vector<int> chans{ch};
  int ds = 1, idx = 1;
  auto has_attn = [&](int d) {
    for (int i = 0; i < c.n_attention_resolutions; ++i)
      if (c.attention_resolutions[i] == d) return true;
    return false;
  };
  for (int level = 0; level < c.n_channel_mult; ++level) {
    const int mult = c.channel_mult[level];
    for (int r = 0; r < c.num_res_blocks; ++r) {
      std::vector<Block> layers;
      snprintf(buf, sizeof buf, "input_blocks.%d.0", idx);
      layers.push_back(Block{B_RES, buf, ch, mult * mc});
      ch = mult * mc;
      if (has_attn(ds)) {
        snprintf(buf, sizeof buf, "input_blocks.%d.1", idx);
        layers.push_back(Block{B_ATTN, buf, ch, ch});
      }
      u->inputs.push_back(layers);
      chans.push_back(ch);
      ++idx;
    }
    if (level != c.n_channel_mult - 1) {
      snprintf(buf, sizeof buf, "input_blocks.%d.0", idx);
      u->inputs.push_back({Block{B_DOWN, buf, ch, ch}});
      chans.push_back(ch);
      ds *= 2;
      ++idx;
    }
  }
  u->middle.push_back(Block{B_RES, "middle_block.0", ch, ch});
  u->middle.push_back(Block{B_ATTN, "middle_block.1", ch, ch});
  u->middle.push_back(Block{B_RES, "middle_block.2", ch, ch});
  int oidx = 0;
  for (int level = c.n_channel_mult - 1; level >= 0; --level) {
    const int mult = c.channel_mult[level];
    for (int i = 0; i < c.num_res_blocks + 1; ++i) {
      const int ich = chans.back();
      chans.pop_back();
      std::vector<Block> layers;
      snprintf(buf, sizeof buf, "output_blocks.%d.0", oidx);
      layers.push_back(Block{B_RES, buf, ch + ich, mc * mult});
      ch = mc * mult;
      if (has_attn(ds)) {
        snprintf(buf, sizeof buf, "output_blocks.%d.%d", oidx, (int)layers.size());
        layers.push_back(Block{B_ATTN, buf, ch, ch});
      }
      if (level && i == c.num_res_blocks) {
        snprintf(buf, sizeof buf, "output_blocks.%d.%d", oidx, (int)layers.size());
        layers.push_back(Block{B_UP, buf, ch, ch});
        ds /= 2;
      }
      u->outputs.push_back(layers);
      ++oidx;
    }
  }
  u->final_ch = ch;
}

void add_param(HoloUnet* u, const std::string& name, std::vector<int64_t> shape, ParamKind kind) {
  ParamSlot s;
  s.name = name;
  s.shape = shape;
  s.numel = 1;
  for (auto d : shape) s.numel *= d;
  s.kind = kind;
  s.priv = nullptr;
  s.set = false;
  u->pindex[name] = (int)u->params.size();
  u->params.push_back(s);
}

void enumerate_params(HoloUnet* u) {
  const HoloUnetCfg& c = u->cfg;
  const int64_t mc = c.model_channels, ted = 4 * mc;
  u->ted = (int)ted;
  add_param(u, "time_embed.0.weight", {ted, mc}, P_PLAIN);
  add_param(u, "time_embed.0.bias", {ted}, P_PLAIN);
  add_param(u, "time_embed.2.weight", {ted, ted}, P_PLAIN);
  add_param(u, "time_embed.2.bias", {ted}, P_PLAIN);
  u->emb_rows = 0;
  auto add_block = [&](const Block& b) {
    const std::string& p = b.prefix;
    const int64_t ci = b.cin, co = b.cout;
    switch (b.kind) {
      case B_CONV:
        add_param(u, p + ".weight", {co, ci, 3, 3, 3}, P_CONV3);
        add_param(u, p + ".bias", {co}, P_PLAIN);
        break;
      case B_RES:
        add_param(u, p + ".in_layers.0.weight", {ci}, P_PLAIN);
        add_param(u, p + ".in_layers.0.bias", {ci}, P_PLAIN);
        add_param(u, p + ".in_layers.2.weight", {co, ci, 3, 3, 3}, P_CONV3);
        add_param(u, p + ".in_layers.2.bias", {co}, P_PLAIN);
        add_param(u, p + ".emb_layers.1.weight", {2 * co, ted}, P_EMB_W);
        add_param(u, p + ".emb_layers.1.bias", {2 * co}, P_EMB_B);
        u->emb_row_off[p] = u->emb_rows;
        u->emb_rows += (int)(2 * co);
        add_param(u, p + ".out_layers.0.weight", {co}, P_PLAIN);
        add_param(u, p + ".out_layers.0.bias", {co}, P_PLAIN);
        add_param(u, p + ".out_layers.3.weight", {co, co, 3, 3, 3}, P_CONV3);
        add_param(u, p + ".out_layers.3.bias", {co}, P_PLAIN);
        if (ci != co) {
          add_param(u, p + ".skip_connection.weight", {co, ci, 1, 1, 1}, P_CONV1);
          add_param(u, p + ".skip_connection.bias", {co}, P_PLAIN);
        }
        break;
      case B_ATTN:
        add_param(u, p + ".norm.weight", {ci}, P_PLAIN);
        add_param(u, p + ".norm.bias", {ci}, P_PLAIN);
        add_param(u, p + ".qkv.weight", {3 * ci, ci, 1}, P_CONV1);
        add_param(u, p + ".qkv.bias", {3 * ci}, P_PLAIN);
        add_param(u, p + ".proj_out.weight", {ci, ci, 1}, P_CONV1);
        add_param(u, p + ".proj_out.bias", {ci}, P_PLAIN);
        break;
      case B_DOWN:
        add_param(u, p + ".op.weight", {co, ci, 3, 3, 3}, P_CONV3);
        add_param(u, p + ".op.bias", {co}, P_PLAIN);
        break;
      case B_UP:
        add_param(u, p + ".conv.weight", {co, ci, 3, 3, 3}, P_CONV3);
        add_param(u, p + ".conv.bias", {co}, P_PLAIN);
        break;
    }
  };
  for (auto& l : u->inputs)
    for (auto& b : l) add_block(b);
  for (auto& b : u->middle) add_block(b);
  for (auto& l : u->outputs)
    for (auto& b : l) add_block(b);
  add_param(u, "out.0.weight", {u->final_ch}, P_PLAIN);
  add_param(u, "out.0.bias", {u->final_ch}, P_PLAIN);
  add_param(u, "out.2.weight", {c.out_channels, u->final_ch, 3, 3, 3}, P_CONV3);
  add_param(u, "out.2.bias", {c.out_channels}, P_PLAIN);
}

const float* P(HoloUnet* u, const std::string& name) {
  auto it = u->pindex.find(name);
  if (it == u->pindex.end()) return nullptr;
  return u->params[it->second].priv;
}

// ---------------------------------------------------------------------------------------------
// plan builder
// ---------------------------------------------------------------------------------------------
struct Planner {
  HoloUnet* u;
  int N;
  char* base;  // workspace base (may be null for a sizing pass)
  Arena arena;                // big activations, after the small regions
  size_t small_top = 0;       // coef / emb buffers
  size_t stats_top = 0;       // GroupNorm statistics (zeroed every forward)
  size_t stats_cap, small_cap;
  size_t stats_base, small_base, arena_base;
  std::vector<Op>& ops;
  std::vector<Tape>* tape = nullptr;  // training forward: every layer is recorded, nothing is released
  size_t last_mom = 0;                // moments buffer of the last emit_finalize (training)

  Planner(HoloUnet* u_, int N_, void* ws, std::vector<Op>& ops_) : u(u_), N(N_), base((char*)ws), ops(ops_) {
    // generous fixed regions for the small buffers
    stats_cap = 0;
    small_cap = Arena::al((size_t)N * 8 * 1024 * 256 * 2 + (size_t)N * (u->emb_rows + 4 * u->ted) * 4 * 2 + 65536);
    stats_base = 0;
    small_base = stats_cap;
    arena_base = stats_cap + small_cap;
    arena.keep = u->keep_intermediates;
  }
  template <class T>
  T* ptr(size_t off) {
    return reinterpret_cast<T*>(base + off);
  }
  int64_t vox(int R) const { return (int64_t)R * R * R; }
  bool bfs() const { return u->compute_mode == 1; }  // bf16 storage of the activations

  Act new_act(int C, int R, bool f32 = false) {
    Act a;
    a.C = C;
    a.R = R;
    a.bytes = (size_t)N * vox(R) * C * ((bfs() && !f32) ? 2 : sizeof(float));
    a.off = arena_base + arena.alloc(a.bytes);
    return a;
  }
  // GroupNorm partial-sum buffer [N][slabs][C][2] doubles; the slab count depends on the producing kernel
  void alloc_stats(Act& a, int slabs) {
    a.stats_B = slabs;
    a.stats_bytes = (size_t)N * slabs * a.C * 2 * sizeof(double);
    a.stats_off = arena_base + arena.alloc(a.stats_bytes);
  }
  void release(Act& a) {
    arena.free(a.off - arena_base, a.bytes);
    if (a.stats_bytes) arena.free(a.stats_off - arena_base, a.stats_bytes);
  }
  size_t small_alloc(size_t bytes) {
    size_t off = small_base + small_top;
    small_top += Arena::al(bytes);
    return off;
  }
  size_t scratch_alloc(size_t bytes) { return arena_base + arena.alloc(bytes); }
  void scratch_free(size_t off, size_t bytes) { arena.free(off - arena_base, bytes); }

  void emit_stats(Act& a) {
    int B, vpb;
    gn_stats_geometry(a.C, vox(a.R), &B, &vpb);
    alloc_stats(a, B);
    Op op;
    op.kind = OP_STATS;
    op.f0 = ptr<float>(a.off);
    op.dout = ptr<double>(a.stats_off);
    op.i0 = a.C;
    op.i1 = bfs() ? 1 : 0;
    op.l0 = vox(a.R);
    ops.push_back(op);
  }
  // returns coef offset
  size_t emit_finalize(const Act& x0, const Act* x1, const float* gamma, const float* beta, const float* film,
                       int film_cout) {
    const int Cin = x0.C + (x1 ? x1->C : 0);
    size_t coef = small_alloc((size_t)N * Cin * 2 * sizeof(float));
    Op op;
    op.kind = OP_FINAL;
    if (tape) {
      last_mom = small_alloc((size_t)N * Cin * 2 * sizeof(float));
      op.o1 = ptr<float>(last_mom);
    }
    op.d0 = ptr<double>(x0.stats_off);
    op.i0 = x0.C;
    op.d1 = x1 ? ptr<double>(x1->stats_off) : nullptr;
    op.i1 = x1 ? x1->C : 0;
    op.i4 = x0.stats_B;
    op.i5 = x1 ? x1->stats_B : 0;
    op.l0 = vox(x0.R);
    op.f0 = gamma;
    op.f1 = beta;
    op.f2 = film;
    op.i2 = u->emb_rows;
    op.i3 = film_cout;
    op.o0 = ptr<float>(coef);
    ops.push_back(op);
    return coef;
  }
  void emit_conv(const Act& x0, const Act* x1, int in_R_logical, int ups, int out_R, int stride, int ksz,
                 const float* w, const float* bias, size_t coef_off, bool has_coef, int act, const float* residual,
                 float* out, int Cout, Act* stats_of = nullptr, const Act* skip0 = nullptr,
                 const Act* skip1 = nullptr, const float* skip_w = nullptr, const float* skip_bias = nullptr,
                 bool in_f32 = false, bool out_f32 = false, const ConvParams* qkv_pack = nullptr) {
    Op op;
    op.kind = OP_CONV;
    ConvParams& p = op.conv;
    memset(&p, 0, sizeof p);
    if (qkv_pack) {  // (an attention block's qkv convolution: conv_plan may fuse the attention's operand packing into it)
      p.qkv_q = qkv_pack->qkv_q, p.qkv_k = qkv_pack->qkv_k, p.qkv_vt = qkv_pack->qkv_vt;
      p.qkv_scale = qkv_pack->qkv_scale, p.qkv_T = qkv_pack->qkv_T, p.qkv_CH = qkv_pack->qkv_CH, p.qkv_H = qkv_pack->qkv_H;
    }
    p.in_bf16 = bfs() && !in_f32;
    p.res_bf16 = bfs();
    p.out_bf16 = bfs() && !out_f32;
    p.src0 = ptr<float>(x0.off);
    p.src1 = x1 ? ptr<float>(x1->off) : nullptr;
    p.C0 = x0.C;
    p.C1 = x1 ? x1->C : 0;
    p.N = N;
    p.ID = p.IH = p.IW = in_R_logical;
    p.ups = ups;
    p.OD = p.OH = p.OW = out_R;
    p.stride = stride;
    p.pad = ksz == 3 ? 1 : 0;
    p.ksz = ksz;
    p.Cout = Cout;
    p.CoutP = pad_cout(Cout);
    p.CinP = pad_cin(p.C0 + p.C1);
    p.w = w;
    if (u->compute_mode) {  // halo-path launches multiply on the bf16 matrix cores (conv_launch checks the pointers)
      auto it = u->bf_of.find(w);
      p.w_bf = it == u->bf_of.end() ? nullptr : it->second;
      auto itt = u->bft_of.find(w);
      p.w_bft = itt == u->bft_of.end() ? nullptr : itt->second;
      p.bf16 = u->compute_mode;
    }
    if (u->compute_mode == 0) {  // exact fp32: the Winograd-in-depth kernel where conv_plan finds 128-voxel tiles
      auto it = u->wino_of.find(w);
      p.w_wino = it == u->wino_of.end() ? nullptr : it->second;
      auto it2 = u->wino2_of.find(w);
      p.w_wino2 = it2 == u->wino2_of.end() ? nullptr : it2->second;
      auto it3 = u->wino3_of.find(w);
      p.w_wino3 = it3 == u->wino3_of.end() ? nullptr : it3->second;
    }
    p.coef = has_coef ? ptr<float>(coef_off) : nullptr;
    p.act = act;
    p.bias = bias;
    p.residual = residual;
    p.out = out;
    if (skip_w) {  // 1x1x1 skip connection fused as extra K chunks (halo kernel)
      p.skip_src0 = ptr<float>(skip0->off);
      p.skip_src1 = skip1 ? ptr<float>(skip1->off) : nullptr;
      p.skip_C0 = skip0->C;
      p.skip_C1 = skip1 ? skip1->C : 0;
      p.skip_w = skip_w;
      if (u->compute_mode) {
        auto it = u->bf_of.find(skip_w);
        p.skip_w_bf = it == u->bf_of.end() ? nullptr : it->second;
        auto itt = u->bft_of.find(skip_w);
        p.skip_w_bft = itt == u->bft_of.end() ? nullptr : itt->second;
      }
      if (u->compute_mode == 0) {
        auto it = u->wino_of.find(skip_w);
        p.skip_w_wino = it == u->wino_of.end() ? nullptr : it->second;
        auto it2 = u->wino2_of.find(skip_w);
        p.skip_w_wino2 = it2 == u->wino2_of.end() ? nullptr : it2->second;
        auto it3 = u->wino3_of.find(skip_w);
        p.skip_w_wino3 = it3 == u->wino3_of.end() ? nullptr : it3->second;
      }
      p.skip_CinP = pad_cin(p.skip_C0 + p.skip_C1);
      p.skip_bias = skip_bias;
    }
    size_t sb = conv_plan(p, u->ctx->num_cus);
    size_t so = 0;
    if (sb) {
      so = scratch_alloc(sb);
      p.partial = ptr<float>(so);
    }
    // GroupNorm statistics of the output: from the conv / split-K-reduce epilogue when the launch can
    // produce them, else by a separate pass over the output
    const int slabs = stats_of ? conv_stats_slabs(p) : 0;
    if (slabs > 0) {
      alloc_stats(*stats_of, slabs);
      p.stats = ptr<double>(stats_of->stats_off);
    }
    // The split-K scratch is released only AFTER the statistics buffer has its place: the reduce kernel of this very
    // launch writes the statistics while other workgroups of it still read partial sums, so the two must not share memory.
    // (Released before, first fit could hand the scratch's own first bytes to the statistics: sample 0 of a split launch then
    // came out wrong in ~25 % of the runs of the bf16 mode at 32^3 - scripts/bf16_batch_check.py, found in round 4.)
    // Stream order protects the scratch against every LATER launch.
    if (sb) scratch_free(so, sb);
    ops.push_back(op);
    if (stats_of && slabs == 0) emit_stats(*stats_of);
  }

  Act resblock(const Block& b, Act& x0, Act* x1) {
    const std::string& p = b.prefix;
    const int R = x0.R;
    size_t coefA = emit_finalize(x0, x1, P(u, p + ".in_layers.0.weight"), P(u, p + ".in_layers.0.bias"), nullptr, 0);
    const size_t momA = last_mom;
    Act h1 = new_act(b.cout, R);
    emit_conv(x0, x1, R, 0, R, 1, 3, P(u, p + ".in_layers.2.weight"), P(u, p + ".in_layers.2.bias"), coefA, true, 1,
              nullptr, ptr<float>(h1.off), b.cout, &h1);
    const float* film = ptr<float>(eml_off) + u->emb_row_off[p];
    size_t coefB =
        emit_finalize(h1, nullptr, P(u, p + ".out_layers.0.weight"), P(u, p + ".out_layers.0.bias"), film, b.cout);
    const size_t momB = last_mom;
    Act s;
    const float* residual;
    const bool has_skip = b.cin != b.cout;
    // the 1x1x1 skip conv rides inside the second 3x3x3 conv (halo kernel) wherever that kernel applies
    // (the bf16x3 kernel has no fused-skip variant: its skip connection runs as a separate fp32 1x1x1 conv)
    // (below 8^3 the convolution runs on the row-tile kernel, which takes the skip's channels as extra K chunks: exact-fp32
    //  mode; four launches + four reduces less at the 4^3 level of the north-star net)
    bool fuse_skip = has_skip && ((R % 8) == 0 || (u->compute_mode == 0 && R < 8)) && b.cout >= 64 && u->compute_mode != 2 &&
                     !getenv("HOLO_NO_SKIP_FUSION");
    // development knob: from this grid size on the skip runs as its own 1x1x1 launch whose output is the second
    // convolution's residual (A/B of the fused form on the wide levels)
    // From 64^3 on (exact-fp32 mode) the skip runs as its own streaming 1x1x1 launch (conv1x1_stream_kernel) whose output is the
    // second convolution's residual: measured on the north-star net, fused 305 us per launch against 208 (plain) + ~45.
    // HOLO_SKIP_FUSION_BELOW_R=<R>: development knob for the threshold (A/B of the two forms)
    {
      const char* mr = getenv("HOLO_SKIP_FUSION_BELOW_R");
      const int below = mr ? atoi(mr) : 64;
      if (fuse_skip && u->compute_mode == 0 && R >= below && (b.cin % 32) == 0 && b.cin <= 256 && (b.cout % 64) == 0) fuse_skip = false;
    }
    if (has_skip && !fuse_skip) {
      s = new_act(b.cout, R);
      emit_conv(x0, x1, R, 0, R, 1, 1, P(u, p + ".skip_connection.weight"), P(u, p + ".skip_connection.bias"), 0,
                false, 0, nullptr, ptr<float>(s.off), b.cout);
      residual = ptr<float>(s.off);
    } else if (has_skip) {
      residual = nullptr;
    } else {
      residual = ptr<float>(x0.off);
    }
    Act out = new_act(b.cout, R);
    if (fuse_skip)
      emit_conv(h1, nullptr, R, 0, R, 1, 3, P(u, p + ".out_layers.3.weight"), P(u, p + ".out_layers.3.bias"), coefB,
                true, 1, nullptr, ptr<float>(out.off), b.cout, &out, &x0, x1, P(u, p + ".skip_connection.weight"),
                P(u, p + ".skip_connection.bias"));
    else
      emit_conv(h1, nullptr, R, 0, R, 1, 3, P(u, p + ".out_layers.3.weight"), P(u, p + ".out_layers.3.bias"), coefB,
                true, 1, residual, ptr<float>(out.off), b.cout, &out);
    release(h1);
    if (has_skip && !fuse_skip) release(s);
    if (tape) {
      Tape t;
      t.kind = B_RES;
      t.b = b;
      t.x0 = x0;
      t.has_x1 = x1 != nullptr;
      if (x1) t.x1 = *x1;
      t.h1 = h1;
      t.out = out;
      t.coefA = coefA;
      t.coefB = coefB;
      t.momA = momA;
      t.momB = momB;
      t.film = film;
      t.has_skip = has_skip;
      tape->push_back(t);
    }
    return out;
  }

  Act attention(const Block& b, Act& x) {
    const std::string& p = b.prefix;
    const int C = x.C, R = x.R, H = u->cfg.num_heads, ch = C / H;
    const int64_t T = vox(R);
    size_t coef = emit_finalize(x, nullptr, P(u, p + ".norm.weight"), P(u, p + ".norm.bias"), nullptr, 0);
    const size_t momX = last_mom;
    const size_t qkv_bytes = (size_t)N * T * 3 * C * sizeof(float);
    const size_t s_bytes = (size_t)N * H * T * T * sizeof(float);
    const size_t a_bytes = (size_t)N * T * C * sizeof(float);
    size_t qkv = scratch_alloc(qkv_bytes);
    size_t a = scratch_alloc(a_bytes);
    size_t v2_work = 0, v2_bytes = 0;
    bool a_is_bf16 = false;
    const bool flash = flash_attn_supported((int)T, ch) && !getenv("HOLO_NO_FLASH_ATTN");
    Op fop;
    fop.kind = OP_FLASH;
    fop.attn.qkv = ptr<float>(qkv);
    fop.attn.out = ptr<float>(a);
    fop.attn.N = N;
    fop.attn.T = (int)T;
    fop.attn.C = C;
    fop.attn.H = H;
    {
      const double sc = 1.0 / sqrt(sqrt((double)ch));
      fop.attn.scale2 = (float)(sc * sc);
    }
    // bf16 mode, sequences of 1 024 tokens and more: the packed-operand bf16 kernel (it splits the key range to fill
    // the chip, so it also serves the shorter of them; below that the exact-fp32 kernel is as fast).
    // HOLO_BF16_FLASH_MIN_T lowers the threshold (tests).  (A first, shared-tile form of the bf16 kernel was removed in
    // round 3: no shape reached it any more, and forced on by the tests it showed a rare dependence on stale memory.)
    ConvParams qp;
    memset(&qp, 0, sizeof qp);
    bool offer_pack = false;
    if (flash) {
      const char* mt = getenv("HOLO_BF16_FLASH_MIN_T");
      const int64_t min_t = mt ? atoll(mt) : 1024;
      fop.i0 = 0;
      if (u->compute_mode == 1 && T >= min_t && flash_attn_bf16v2_supported((int)T, ch)) {
        // packed bf16 operands (V transposed) in scratch, bf16 attention output
        fop.i0 = 2;
        v2_bytes = flash_attn_bf16v2_workspace_bytes(fop.attn, u->ctx->num_cus);
        v2_work = scratch_alloc(v2_bytes);
        fop.o1 = ptr<float>(v2_work);
        fop.i1 = 1;
        a_is_bf16 = true;
        if (!tape && bfs()) {  // the qkv convolution may write the packed operands itself (the backward's tape keeps fp32 qkv)
          flash_attn_bf16v2_operands(fop.attn, fop.o1, &qp.qkv_q, &qp.qkv_k, &qp.qkv_vt, &qp.qkv_scale);
          qp.qkv_T = (int)T, qp.qkv_CH = ch, qp.qkv_H = H;
          offer_pack = true;
        }
      }
    }
    // (the attention internals - qkv and the attention output - stay fp32 in every mode)
    emit_conv(x, nullptr, R, 0, R, 1, 1, P(u, p + ".qkv.weight"), P(u, p + ".qkv.bias"), coef, true, 0, nullptr,
              ptr<float>(qkv), 3 * C, nullptr, nullptr, nullptr, nullptr, nullptr, false, /*out_f32=*/true,
              offer_pack ? &qp : nullptr);
    if (flash) {
      Op op = fop;
      op.i2 = ops.back().kind == OP_CONV && ops.back().conv.mode == 5 ? 1 : 0;  // operands already packed
      if (getenv("HOLO_DEBUG_PLAN"))
        fprintf(stderr, "[plan] attention %s: T=%lld C=%d heads=%d -> %s flash kernel\n", p.c_str(), (long long)T, C, H,
                op.i0 == 2 ? "bf16" : "fp32");
      ops.push_back(op);
    } else {
      size_t S = scratch_alloc(s_bytes);
    {
      Op op;
      op.kind = OP_GEMM;
      GemmParams& g = op.gemm;
      memset(&g, 0, sizeof g);
      g.A = ptr<float>(qkv);
      g.B = ptr<float>(qkv) + ch;
      g.C = ptr<float>(S);
      g.M = (int)T;
      g.Nn = (int)T;
      g.K = ch;
      g.lda = 3 * C;
      g.ldb = 3 * C;
      g.ldc = (int)T;
      g.nb0 = N;
      g.nb1 = H;
      g.sa0 = T * 3 * C;
      g.sa1 = 3 * ch;
      g.sb0 = T * 3 * C;
      g.sb1 = 3 * ch;
      g.sc0 = (int64_t)H * T * T;
      g.sc1 = T * T;
      g.b_kmajor = 0;
      const double sc = 1.0 / sqrt(sqrt((double)ch));
      g.alpha = (float)(sc * sc);
      ops.push_back(op);
    }
    {
      Op op;
      op.kind = OP_SOFTMAX;
      op.o0 = ptr<float>(S);
      op.l0 = (int64_t)N * H * T;
      op.i0 = (int)T;
      ops.push_back(op);
    }
    {
      Op op;
      op.kind = OP_GEMM;
      GemmParams& g = op.gemm;
      memset(&g, 0, sizeof g);
      g.A = ptr<float>(S);
      g.B = ptr<float>(qkv) + 2 * ch;
      g.C = ptr<float>(a);
      g.M = (int)T;
      g.Nn = ch;
      g.K = (int)T;
      g.lda = (int)T;
      g.ldb = 3 * C;
      g.ldc = C;
      g.nb0 = N;
      g.nb1 = H;
      g.sa0 = (int64_t)H * T * T;
      g.sa1 = T * T;
      g.sb0 = T * 3 * C;
      g.sb1 = 3 * ch;
      g.sc0 = T * C;
      g.sc1 = ch;
      g.b_kmajor = 1;
      g.alpha = 1.0f;
      ops.push_back(op);
    }
      scratch_free(S, s_bytes);
    }
    Act out = new_act(C, R);
    Act av;  // view of `a` as an activation for the 1x1 conv
    av.off = a;
    av.C = C;
    av.R = R;
    emit_conv(av, nullptr, R, 0, R, 1, 1, P(u, p + ".proj_out.weight"), P(u, p + ".proj_out.bias"), 0, false, 0,
              ptr<float>(x.off), ptr<float>(out.off), C, &out, nullptr, nullptr, nullptr, nullptr, /*in_f32=*/!a_is_bf16);
    if (v2_bytes) scratch_free(v2_work, v2_bytes);
    scratch_free(qkv, qkv_bytes);
    scratch_free(a, a_bytes);
    if (tape) {
      Tape t;
      t.kind = B_ATTN;
      t.b = b;
      t.x0 = x;
      t.out = out;
      t.coefA = coef;
      t.momA = momX;
      t.qkv = qkv;
      t.a = a;
      tape->push_back(t);
    }
    return out;
  }

  size_t eml_off = 0, embs_off = 0;
  Act x_in, y_out;

  // runs a TimestepEmbedSequential; consumes (releases) the input activation(s)
  Act run_layers(const std::vector<Block>& layers, Act h, Act* skip, bool release_h) {
    bool first = true;
    for (const Block& b : layers) {
      Act out;
      Act* x1 = first ? skip : nullptr;
      switch (b.kind) {
        case B_CONV:
          out = new_act(b.cout, h.R);
          emit_conv(h, nullptr, h.R, 0, h.R, 1, 3, P(u, b.prefix + ".weight"), P(u, b.prefix + ".bias"), 0, false, 0,
                    nullptr, ptr<float>(out.off), b.cout, &out);
          break;
        case B_RES:
          out = resblock(b, h, x1);
          break;
        case B_ATTN:
          out = attention(b, h);
          break;
        case B_DOWN:
          out = new_act(b.cout, h.R / 2);
          emit_conv(h, nullptr, h.R, 0, h.R / 2, 2, 3, P(u, b.prefix + ".op.weight"), P(u, b.prefix + ".op.bias"), 0,
                    false, 0, nullptr, ptr<float>(out.off), b.cout, &out);
          break;
        case B_UP:
          out = new_act(b.cout, h.R * 2);
          emit_conv(h, nullptr, h.R * 2, 1, h.R * 2, 1, 3, P(u, b.prefix + ".conv.weight"),
                    P(u, b.prefix + ".conv.bias"), 0, false, 0, nullptr, ptr<float>(out.off), b.cout, &out);
          break;
      }
      if (tape && (b.kind == B_CONV || b.kind == B_DOWN || b.kind == B_UP)) {
        Tape t;
        t.kind = b.kind;
        t.b = b;
        t.x0 = h;
        t.out = out;
        tape->push_back(t);
      }
      if (!first || release_h) release(h);
      if (first && skip) release(*skip);
      h = out;
      first = false;
    }
    return h;
  }

  void build() {
    const HoloUnetCfg& c = u->cfg;
    const int R = c.image_size;
    ops.clear();
    u->block_outputs.clear();
    // time embedding
    size_t emb = small_alloc((size_t)N * u->ted * 4);
    size_t embs = small_alloc((size_t)N * u->ted * 4);
    eml_off = small_alloc((size_t)N * u->emb_rows * 4);
    embs_off = embs;
    {
      Op op;
      op.kind = OP_TEMB;
      op.f0 = P(u, "time_embed.0.weight");
      op.f1 = P(u, "time_embed.0.bias");
      op.f2 = P(u, "time_embed.2.weight");
      op.f3 = P(u, "time_embed.2.bias");
      op.o0 = ptr<float>(emb);
      op.o1 = ptr<float>(embs);
      ops.push_back(op);
    }
    {
      Op op;
      op.kind = OP_EMBLIN;
      op.f0 = ptr<float>(embs);
      op.f1 = u->emb_w;
      op.f2 = u->emb_b;
      op.o0 = ptr<float>(eml_off);
      ops.push_back(op);
    }
    Act x = new_act(c.in_channels, R);
    x_in = x;
    {
      Op op;
      op.kind = OP_IN;
      op.o0 = ptr<float>(x.off);
      op.i0 = c.in_channels;
      op.i1 = bfs() ? 1 : 0;
      op.l0 = vox(R);
      ops.push_back(op);
    }
    std::vector<Act> hs;
    Act h = x;
    char tag[64];
    for (size_t i = 0; i < u->inputs.size(); ++i) {
      // h is also referenced by hs (except the raw input x): do not release it when consumed
      h = run_layers(u->inputs[i], h, nullptr, /*release_h=*/i == 0);
      hs.push_back(h);
      snprintf(tag, sizeof tag, "input_blocks.%d", (int)i);
      u->block_outputs[tag] = h;
    }
    // middle: h == hs.back(); keep it alive for the skip connection
    h = run_layers(u->middle, h, nullptr, false);
    u->block_outputs["middle_block"] = h;
    for (size_t i = 0; i < u->outputs.size(); ++i) {
      Act skip = hs.back();
      hs.pop_back();
      h = run_layers(u->outputs[i], h, &skip, true);
      snprintf(tag, sizeof tag, "output_blocks.%d", (int)i);
      u->block_outputs[tag] = h;
    }
    size_t coef = emit_finalize(h, nullptr, P(u, "out.0.weight"), P(u, "out.0.bias"), nullptr, 0);
    Act y = new_act(c.out_channels, R, /*f32=*/true);  // the network output stays fp32
    if (tape) {
      Tape t;
      t.kind = 100;
      t.b = Block{B_CONV, "out", u->final_ch, c.out_channels};
      t.x0 = h;
      t.out = y;
      t.coefA = coef;
      t.momA = last_mom;
      tape->push_back(t);
    }
    emit_conv(h, nullptr, R, 0, R, 1, 3, P(u, "out.2.weight"), P(u, "out.2.bias"), coef, true, 1, nullptr,
              ptr<float>(y.off), c.out_channels, nullptr, nullptr, nullptr, nullptr, nullptr, false, /*out_f32=*/true);
    y_out = y;
    release(h);
    {
      Op op;
      op.kind = OP_OUT;
      op.f0 = ptr<float>(y.off);
      op.i0 = c.out_channels;
      op.l0 = vox(R);
      ops.push_back(op);
    }
    release(y);
    if (getenv("HOLO_DEBUG_PLAN")) {  // development: what the plan launches
      int n_fin = 0, n_conv = 0, n_split = 0;
      for (const Op& o : ops) {
        n_fin += o.kind == OP_FINAL;
        if (o.kind != OP_CONV) continue;
        ++n_conv;
        n_split += o.conv.nsplit > 1;
      }
      fprintf(stderr, "[plan] batch %d: %zu ops | %d convs, %d of them split-K (+ a reduce launch) | %d gn_finalize launches\n", N, ops.size(),
              n_conv, n_split, n_fin);
    }
  }
  size_t total_bytes() const { return arena_base + arena.peak; }
  bool regions_ok() const { return stats_top <= stats_cap && small_top <= small_cap; }
};

// ---------------------------------------------------------------------------------------------
// training plan (second half): the forward with every intermediate kept and every layer recorded,
// then the backward as a list of launches in reverse layer order.  fp32 mode only.
//   * dgrad of a stride-1 convolution = the forward conv kernels on the flipped / transposed weights
//     (holo_unet_set_dgrad_weight);  Downsample: conv_dgrad_s2_kernel;  Upsample: dgrad at the fine size + sumpool2
//   * wgrad: conv_wgrad_kernel re-applies GroupNorm . FiLM . SiLU to the raw input like the forward's staging does
//   * GroupNorm (+ FiLM + SiLU): gn_bwd_launch;  attention: batched fp32 MFMA GEMMs around the softmax rows
// Gradients of activations are allocated on first use and ACCUMULATED by later consumers (skip connections, the
// identity branches of ResBlock / AttentionBlock).  Parameter gradients live in the workspace in the reference's layouts.
// ---------------------------------------------------------------------------------------------
struct TrainPlanner {
  HoloUnet* u;
  int N;
  Planner pl;
  std::vector<Tape> tape;
  std::vector<std::function<int(void*)>>& bops;
  std::map<size_t, std::pair<size_t, bool>> grads;  // activation offset -> (gradient offset, already written)
  std::string err;

  TrainPlanner(HoloUnet* u_, int N_, void* ws, std::vector<Op>& fops, std::vector<std::function<int(void*)>>& b)
      : u(u_), N(N_), pl(u_, N_, ws, fops), bops(b) {
    pl.arena.keep = true;
    pl.tape = &tape;
  }
  template <class T>
  T* ptr(size_t off) {
    return pl.ptr<T>(off);
  }
  size_t alloc(size_t bytes) { return pl.scratch_alloc(bytes); }
  int64_t vox(int R) const { return (int64_t)R * R * R; }
  // gradient buffer of an activation; `acc` tells the caller whether to accumulate (a consumer wrote it before)
  // (all bookkeeping is in workspace OFFSETS: the sizing pass runs with a null base, and pointer differences against a
  // null base are undefined behaviour that an optimising compiler does exploit)
  static constexpr size_t NONE = ~(size_t)0;
  size_t grad_of(const Act& a, int* acc) {
    auto it = grads.find(a.off);
    if (it == grads.end()) {
      const size_t off = alloc((size_t)N * vox(a.R) * a.C * sizeof(float));
      grads[a.off] = std::make_pair(off, true);
      *acc = 0;
      return off;
    }
    *acc = 1;
    return it->second.first;
  }
  size_t grad_ready(const Act& a) {  // gradient of a layer OUTPUT: must have been written by its consumers
    auto it = grads.find(a.off);
    if (it == grads.end()) {
      err = "internal: a layer output has no gradient";
      return NONE;
    }
    return it->second.first;
  }
  float* pgrad(const std::string& name) {
    auto it = u->pindex.find(name);
    return it == u->pindex.end() ? nullptr : reinterpret_cast<float*>(pl.base + u->grad_off[it->second]);
  }
  const float* dgw(const std::string& name) {
    auto it = u->dgrad_w.find(name);
    if (it == u->dgrad_w.end()) {
      err = "holo_unet_backward: call holo_unet_set_dgrad_weight for '" + name + "' first";
      return nullptr;
    }
    return it->second;
  }

  // dgrad of a stride-1 conv (3x3x3 pad 1 or 1x1x1): out[M][cin] = conv(gy[M][cout], flipped weights)
  void emit_dgrad(size_t gy_off, int cout, int R, const std::string& wname, int cin, int ksz, size_t out_off,
                  bool accumulate = false) {
    const float* w = dgw(wname);
    if (!w) return;
    Act g;
    g.off = gy_off;
    g.C = cout;
    g.R = R;
    pl.emit_conv(g, nullptr, R, 0, R, 1, ksz, w, nullptr, 0, false, 0, accumulate ? ptr<float>(out_off) : nullptr,
                 ptr<float>(out_off), cin);
    Op op = pl.ops.back();
    pl.ops.pop_back();
    ConvParams cp = op.conv;
    bops.push_back([cp](void* st) { return conv_launch(cp, st); });
  }
  void emit_wgrad(size_t gy_off, int cout, const Act& x0, const Act* x1, int in_R, int ups, int out_R, int stride, int ksz,
                  size_t coef, bool has_coef, int act, const std::string& wname, const std::string& bname) {
    const float* gy = ptr<float>(gy_off);
    WgradParams w;
    memset(&w, 0, sizeof w);
    w.gy = gy;
    w.src0 = ptr<float>(x0.off);
    w.src1 = x1 ? ptr<float>(x1->off) : nullptr;
    w.C0 = x0.C;
    w.C1 = x1 ? x1->C : 0;
    w.N = N;
    w.ID = w.IH = w.IW = in_R;
    w.ups = ups;
    w.OD = w.OH = w.OW = out_R;
    w.stride = stride;
    w.pad = ksz == 3 ? 1 : 0;
    w.ksz = ksz;
    w.ntaps = ksz == 3 ? 27 : 1;
    w.Cout = cout;
    w.coef = has_coef ? ptr<float>(coef) : nullptr;
    w.act = act;
    const size_t pb = wgrad_partial_bytes(w, u->ctx->num_cus);
    w.partial = ptr<float>(alloc(pb));
    float* dw = pgrad(wname);
    float* db = pgrad(bname);
    double* cs = ptr<double>(alloc(colsum_scratch_bytes(cout)));
    const int ncu = u->ctx->num_cus;
    const int64_t M = (int64_t)N * vox(out_R);
    bops.push_back([w, dw, ncu](void* st) { return conv_wgrad_launch(w, dw, 0, ncu, st); });
    bops.push_back([gy, M, cout, cs, db](void* st) { return colsum_launch(gy, M, cout, cs, db, 0, st); });
  }
  // GroupNorm (+FiLM) (+SiLU) backward of the (virtual concat) input of a conv: ga [M][Cin] -> gradients of x0 / x1
  void emit_gn_bwd(const Act& x0, const Act* x1, size_t ga_off, size_t coef, size_t mom, const std::string& gname,
                   const std::string& bname, const float* film, int film_cout, float* dfilm, int act) {
    GnBwdParams g;
    memset(&g, 0, sizeof g);
    g.x0 = ptr<float>(x0.off);
    g.x1 = x1 ? ptr<float>(x1->off) : nullptr;
    g.C0 = x0.C;
    g.C1 = x1 ? x1->C : 0;
    g.N = N;
    g.V = vox(x0.R);
    g.ga = ptr<float>(ga_off);
    g.coef = ptr<float>(coef);
    g.mom = ptr<float>(mom);
    g.gamma = P(u, gname);
    g.beta = P(u, bname);
    g.film = film;
    g.film_stride = u->emb_rows;
    g.film_cout = film_cout;
    g.act = act;
    g.part = ptr<double>(alloc(gn_bwd_scratch_bytes(g)));
    g.grp = ptr<float>(alloc((size_t)N * (g.C0 + g.C1) * 2 * sizeof(float)));
    g.dgamma = pgrad(gname);
    g.dbeta = pgrad(bname);
    g.dfilm = dfilm;
    g.gx0 = ptr<float>(grad_of(x0, &g.acc0));
    if (x1) g.gx1 = ptr<float>(grad_of(*x1, &g.acc1));
    bops.push_back([g](void* st) { return gn_bwd_launch(g, st); });
  }

  void bwd_res(const Tape& t, float* dfilm_base) {
    const std::string& p = t.b.prefix;
    const int R = t.x0.R, cin = t.b.cin, cout = t.b.cout;
    const int64_t M = (int64_t)N * vox(R);
    const size_t gout = grad_ready(t.out);
    if (gout == NONE) return;
    const Act* x1 = t.has_x1 ? &t.x1 : nullptr;
    // second conv: out = skip(x) + conv2(silu(film(gn2(h1))))
    const size_t ga2 = alloc((size_t)M * cout * sizeof(float));
    emit_dgrad(gout, cout, R, p + ".out_layers.3.weight", cout, 3, ga2);
    emit_wgrad(gout, cout, t.h1, nullptr, R, 0, R, 1, 3, t.coefB, true, 1, p + ".out_layers.3.weight", p + ".out_layers.3.bias");
    const int row = u->emb_row_off[p];
    emit_gn_bwd(t.h1, nullptr, ga2, t.coefB, t.momB, p + ".out_layers.0.weight", p + ".out_layers.0.bias", t.film, cout,
                dfilm_base + row, 1);
    const size_t gh1 = grad_ready(t.h1);
    if (gh1 == NONE) return;
    // first conv: h1 = conv1(silu(gn1([x0 | x1])))
    const size_t ga1 = alloc((size_t)M * cin * sizeof(float));
    emit_dgrad(gh1, cout, R, p + ".in_layers.2.weight", cin, 3, ga1);
    emit_wgrad(gh1, cout, t.x0, x1, R, 0, R, 1, 3, t.coefA, true, 1, p + ".in_layers.2.weight", p + ".in_layers.2.bias");
    emit_gn_bwd(t.x0, x1, ga1, t.coefA, t.momA, p + ".in_layers.0.weight", p + ".in_layers.0.bias", nullptr, 0, nullptr, 1);
    // skip connection: identity, or a 1x1x1 conv of the raw input
    int a0 = 0, a1 = 0;
    float* gx0 = ptr<float>(grad_of(t.x0, &a0));
    float* gx1 = x1 ? ptr<float>(grad_of(*x1, &a1)) : nullptr;
    const float* goutp = ptr<float>(gout);
    if (!t.has_skip) {
      const int64_t n = M * cin;
      bops.push_back([gx0, goutp, n, a0](void* st) { return add_launch(gx0, goutp, n, a0, st); });
    } else {
      const size_t gso = alloc((size_t)M * cin * sizeof(float));
      const float* gs = ptr<float>(gso);
      emit_dgrad(gout, cout, R, p + ".skip_connection.weight", cin, 1, gso);
      emit_wgrad(gout, cout, t.x0, x1, R, 0, R, 1, 1, 0, false, 0, p + ".skip_connection.weight", p + ".skip_connection.bias");
      if (x1) {
        const int C0 = t.x0.C, C1 = t.x1.C;
        bops.push_back([gs, gx0, gx1, M, C0, C1, a0, a1](void* st) { return split_cat_launch(gs, gx0, gx1, M, C0, C1, a0, a1, st); });
      } else {
        const int64_t n = M * cin;
        bops.push_back([gx0, gs, n, a0](void* st) { return add_launch(gx0, gs, n, a0, st); });
      }
    }
  }

  void bwd_attn(const Tape& t) {
    const std::string& p = t.b.prefix;
    const int C = t.x0.C, R = t.x0.R, H = u->cfg.num_heads, ch = C / H;
    const int64_t T = vox(R), M = (int64_t)N * T;
    const size_t gout = grad_ready(t.out);
    if (gout == NONE) return;
    int ax = 0;
    float* gx = ptr<float>(grad_of(t.x0, &ax));
    {  // identity branch
      const int64_t n = M * C;
      const float* goutp = ptr<float>(gout);
      bops.push_back([gx, goutp, n, ax](void* st) { return add_launch(gx, goutp, n, ax, st); });
    }
    // proj_out (1x1 over the attention output a)
    Act av;
    av.off = t.a;
    av.C = C;
    av.R = R;
    const size_t ga_off = alloc((size_t)M * C * sizeof(float));
    float* ga = ptr<float>(ga_off);
    emit_dgrad(gout, C, R, p + ".proj_out.weight", C, 1, ga_off);
    emit_wgrad(gout, C, av, nullptr, R, 0, R, 1, 1, 0, false, 0, p + ".proj_out.weight", p + ".proj_out.bias");
    // attention core: P = softmax(s2 q k^T); dP = ga v^T; dS = P (dP - rowsum(dP P)); dv = P^T ga; dq = s2 dS k; dk = s2 dS^T q
    const size_t sb = (size_t)N * H * T * T * sizeof(float);
    float* Pm = ptr<float>(alloc(sb));
    float* dS = ptr<float>(alloc(sb));
    float* Tm = ptr<float>(alloc(sb));
    const size_t gqkv_off = alloc((size_t)M * 3 * C * sizeof(float));
    float* gqkv = ptr<float>(gqkv_off);
    const float* qkv = ptr<float>(t.qkv);
    const double sc = 1.0 / sqrt(sqrt((double)ch));
    const float s2 = (float)(sc * sc);
    auto gemm = [&](const float* A, int lda, int64_t sa0, int64_t sa1, const float* B, int ldb, int64_t sb0, int64_t sb1,
                    int kmajor, float* Cc, int ldc, int64_t sc0, int64_t sc1, int Mm, int Nn, int K, float alpha) {
      GemmParams g;
      memset(&g, 0, sizeof g);
      g.A = A;
      g.B = B;
      g.C = Cc;
      g.M = Mm;
      g.Nn = Nn;
      g.K = K;
      g.lda = lda;
      g.ldb = ldb;
      g.ldc = ldc;
      g.nb0 = N;
      g.nb1 = H;
      g.sa0 = sa0;
      g.sa1 = sa1;
      g.sb0 = sb0;
      g.sb1 = sb1;
      g.sc0 = sc0;
      g.sc1 = sc1;
      g.b_kmajor = kmajor;
      g.alpha = alpha;
      bops.push_back([g](void* st) { return gemm_launch(g, st); });
    };
    const int64_t TT = T * T, q3 = T * 3 * C;
    const int Ti = (int)T;
    gemm(qkv, 3 * C, q3, 3 * ch, qkv + ch, 3 * C, q3, 3 * ch, 0, Pm, Ti, (int64_t)H * TT, TT, Ti, Ti, ch, s2);
    {
      const int64_t rows = (int64_t)N * H * T;
      bops.push_back([Pm, rows, Ti](void* st) { return softmax_rows_launch(Pm, rows, Ti, st); });
    }
    gemm(ga, C, T * C, ch, qkv + 2 * ch, 3 * C, q3, 3 * ch, 0, dS, Ti, (int64_t)H * TT, TT, Ti, Ti, ch, 1.0f);
    {
      const int64_t rows = (int64_t)N * H * T;
      bops.push_back([Pm, dS, rows, Ti](void* st) { return attn_ds_launch(Pm, dS, rows, Ti, st); });
    }
    const int NH = N * H;
    bops.push_back([Pm, Tm, NH, Ti](void* st) { return transpose_launch(Pm, Tm, NH, Ti, st); });
    gemm(Tm, Ti, (int64_t)H * TT, TT, ga, C, T * C, ch, 1, gqkv + 2 * ch, 3 * C, q3, 3 * ch, Ti, ch, Ti, 1.0f);           // dv
    gemm(dS, Ti, (int64_t)H * TT, TT, qkv + ch, 3 * C, q3, 3 * ch, 1, gqkv, 3 * C, q3, 3 * ch, Ti, ch, Ti, s2);           // dq
    bops.push_back([dS, Tm, NH, Ti](void* st) { return transpose_launch(dS, Tm, NH, Ti, st); });
    gemm(Tm, Ti, (int64_t)H * TT, TT, qkv, 3 * C, q3, 3 * ch, 1, gqkv + ch, 3 * C, q3, 3 * ch, Ti, ch, Ti, s2);            // dk
    // qkv conv (1x1, C -> 3C, GroupNorm applied on load, no activation)
    const size_t gxn = alloc((size_t)M * C * sizeof(float));
    emit_dgrad(gqkv_off, 3 * C, R, p + ".qkv.weight", C, 1, gxn);
    emit_wgrad(gqkv_off, 3 * C, t.x0, nullptr, R, 0, R, 1, 1, t.coefA, true, 0, p + ".qkv.weight", p + ".qkv.bias");
    emit_gn_bwd(t.x0, nullptr, gxn, t.coefA, t.momA, p + ".norm.weight", p + ".norm.bias", nullptr, 0, nullptr, 0);
  }

  void bwd_conv(const Tape& t) {  // input conv / Downsample / Upsample / output head
    const int cin = t.b.cin, cout = t.b.cout, Ri = t.x0.R, Ro = t.out.R;
    const size_t gout = grad_ready(t.out);
    if (gout == NONE) return;
    if (t.kind == 100) {  // y = conv(silu(gn(h)))
      const int64_t M = (int64_t)N * vox(Ri);
      const size_t ga = alloc((size_t)M * cin * sizeof(float));
      emit_dgrad(gout, cout, Ri, "out.2.weight", cin, 3, ga);
      emit_wgrad(gout, cout, t.x0, nullptr, Ri, 0, Ri, 1, 3, t.coefA, true, 1, "out.2.weight", "out.2.bias");
      emit_gn_bwd(t.x0, nullptr, ga, t.coefA, t.momA, "out.0.weight", "out.0.bias", nullptr, 0, nullptr, 1);
      return;
    }
    int ax = 0;
    const size_t gx_off = grad_of(t.x0, &ax);
    float* gx = ptr<float>(gx_off);
    const float* goutp = ptr<float>(gout);
    const std::string wn = t.b.prefix + (t.kind == B_DOWN ? ".op.weight" : t.kind == B_UP ? ".conv.weight" : ".weight");
    const std::string bn = t.b.prefix + (t.kind == B_DOWN ? ".op.bias" : t.kind == B_UP ? ".conv.bias" : ".bias");
    if (t.kind == B_CONV) {
      if (ax) {
        err = "internal: the input conv's source already has a gradient";
        return;
      }
      emit_dgrad(gout, cout, Ri, wn, cin, 3, gx_off);
      emit_wgrad(gout, cout, t.x0, nullptr, Ri, 0, Ro, 1, 3, 0, false, 0, wn, bn);
    } else if (t.kind == B_DOWN) {
      const float* wt = dgw(wn);
      if (!wt) return;
      const int Nn = N;
      const char* zi = getenv("HOLO_DGRAD_S2_DIRECT");  // development / test knob: 1 = conv_dgrad_s2_kernel everywhere
      if (!(zi && zi[0] == '1') && u->dgrad_w.count(wn + "#s1") && Ri == 2 * Ro && (Ri % 8) == 0) {
        // zero insertion + the stride-1 transposed convolution on the forward's conv kernels (8x the multiply-adds, on the
        // Winograd kernels: 0.24 instead of 1.35 ms at 64^3 <- 32^3)
        const size_t gz_off = alloc((size_t)N * vox(Ri) * cout * sizeof(float));
        float* gz = ptr<float>(gz_off);
        bops.push_back([goutp, gz, Nn, Ro, cout](void* st) { return zero_insert2_launch(goutp, gz, Nn, Ro, cout, st); });
        emit_dgrad(gz_off, cout, Ri, wn + "#s1", cin, 3, gx_off, ax != 0);
      } else {
        bops.push_back([goutp, wt, gx, Nn, Ri, Ro, cin, cout, ax](void* st) {
          return conv_dgrad_s2_launch(goutp, wt, gx, Nn, Ri, Ro, cin, cout, ax, st);
        });
      }
      emit_wgrad(gout, cout, t.x0, nullptr, Ri, 0, Ro, 2, 3, 0, false, 0, wn, bn);
    } else {  // B_UP: conv at the fine size of the nearest-upsampled input
      const int64_t Mf = (int64_t)N * vox(Ro);
      const size_t gup_off = alloc((size_t)Mf * cin * sizeof(float));
      float* gup = ptr<float>(gup_off);
      emit_dgrad(gout, cout, Ro, wn, cin, 3, gup_off);
      const int Nn = N;
      bops.push_back([gup, gx, Nn, Ri, cin, ax](void* st) { return sumpool2_launch(gup, gx, Nn, Ri, cin, ax, st); });
      emit_wgrad(gout, cout, t.x0, nullptr, Ro, 1, Ro, 1, 3, 0, false, 0, wn, bn);
    }
  }

  int build() {
    const HoloUnetCfg& c = u->cfg;
    pl.build();
    // parameter gradients: one region in the reference's layouts; emb_layers rows alias the concatenated matrix
    u->grad_off.assign(u->params.size(), 0);
    const size_t gembw = alloc((size_t)u->emb_rows * u->ted * sizeof(float));
    const size_t gembb = alloc((size_t)u->emb_rows * sizeof(float));
    for (size_t i = 0; i < u->params.size(); ++i) {
      const ParamSlot& s = u->params[i];
      if (s.kind == P_EMB_W || s.kind == P_EMB_B) {
        const int row = u->emb_row_off[s.name.substr(0, s.name.rfind(".emb_layers"))];
        u->grad_off[i] = s.kind == P_EMB_W ? gembw + (size_t)row * u->ted * sizeof(float) : gembb + (size_t)row * sizeof(float);
      } else {
        u->grad_off[i] = alloc((size_t)s.numel * sizeof(float));
      }
    }
    // the gradient of the output arrives NCDHW and is laid out channels-last as the gradient of y
    const int R = c.image_size;
    const int64_t V = vox(R);
    const size_t gy = alloc((size_t)N * V * c.out_channels * sizeof(float));
    grads[pl.y_out.off] = std::make_pair(gy, true);
    u->gy_off = gy;
    u->y_off = pl.y_out.off;
    const size_t dfilm = alloc((size_t)N * u->emb_rows * sizeof(float));
    float* dfilm_base = ptr<float>(dfilm);
    for (int i = (int)tape.size() - 1; i >= 0 && err.empty(); --i) {
      const Tape& t = tape[i];
      if (t.kind == B_RES)
        bwd_res(t, dfilm_base);
      else if (t.kind == B_ATTN)
        bwd_attn(t);
      else
        bwd_conv(t);
    }
    if (!err.empty()) {
      set_error("%s", err.c_str());
      return HOLO_E_STATE;
    }
    // embedding path
    {
      const float* embs = ptr<float>(pl.embs_off);
      float* gembs = ptr<float>(alloc((size_t)N * u->ted * sizeof(float)));
      float* dw = ptr<float>(gembw);
      float* db = ptr<float>(gembb);
      const float* w = u->emb_w;
      const int rows = u->emb_rows, K = u->ted, Nn = N;
      bops.push_back([dfilm_base, embs, w, dw, db, gembs, Nn, rows, K](void* st) {
        return film_bwd_launch(dfilm_base, embs, w, dw, db, gembs, Nn, rows, K, st);
      });
      HoloUnet* uu = u;
      const int mc = c.model_channels;
      const float *w1 = P(u, "time_embed.0.weight"), *b1 = P(u, "time_embed.0.bias"), *w2 = P(u, "time_embed.2.weight"),
                  *b2 = P(u, "time_embed.2.bias");
      float *dw1 = pgrad("time_embed.0.weight"), *db1 = pgrad("time_embed.0.bias"), *dw2 = pgrad("time_embed.2.weight"),
            *db2 = pgrad("time_embed.2.bias");
      bops.push_back([uu, Nn, mc, K, w1, b1, w2, b2, gembs, dw1, db1, dw2, db2](void* st) {
        return time_embed_bwd_launch(uu->t_dev, Nn, mc, K, w1, b1, w2, b2, gembs, dw1, db1, dw2, db2, st);
      });
    }
    auto gi = grads.find(pl.x_in.off);
    if (gi == grads.end()) {
      set_error("internal: the network input has no gradient");
      return HOLO_E_STATE;
    }
    u->gx_off = gi->second.first;
    return 0;
  }
  size_t total_bytes() const { return pl.total_bytes(); }
};

int ensure_train_plan(HoloUnet* u, int batch, void* ws) {
  if (u->tplan_batch == batch && u->tplan_ws == ws && !u->tops.empty()) return 0;
  if (u->compute_mode != 0) {
    set_error("holo_unet_backward: the backward pass runs in the fp32 mode only");
    return HOLO_E_UNSUPPORTED;
  }
  for (auto& s : u->params)
    if (!s.set) {
      set_error("holo_unet_backward: parameter '%s' has not been set", s.name.c_str());
      return HOLO_E_STATE;
    }
  u->tops.clear();
  u->bops.clear();
  TrainPlanner tp(u, batch, ws, u->tops, u->bops);
  int rc = tp.build();
  if (rc) {
    u->tops.clear();
    u->bops.clear();
    return rc;
  }
  if (!tp.pl.regions_ok()) {
    set_error("internal: small-buffer regions overflow");
    return HOLO_E_INVALID;
  }
  u->tws_need = tp.total_bytes();
  u->tplan_batch = batch;
  u->tplan_ws = ws;
  u->plan_batch = -1;  // block_outputs were rewritten
  return 0;
}

int ensure_plan(HoloUnet* u, int batch, void* ws) {
  if (u->plan_batch == batch && u->plan_ws == ws && !u->ops.empty()) return 0;
  for (auto& s : u->params)
    if (!s.set) {
      set_error("holo_unet_forward: parameter '%s' has not been set", s.name.c_str());
      return HOLO_E_STATE;
    }
  Planner pl(u, batch, ws, u->ops);
  pl.build();
  if (!pl.regions_ok()) {
    set_error("internal: small-buffer regions overflow");
    return HOLO_E_INVALID;
  }
  u->ws_need = pl.total_bytes();
  u->plan_batch = batch;
  u->plan_ws = ws;
  return 0;
}

int run_op(HoloUnet* u, const Op& op, int N, const float* x, const int64_t* t, float* y, void* stream) {
  switch (op.kind) {
    case OP_MEMSET:
      HIP_TRY(hipMemsetAsync(op.o0, 0, op.bytes, (hipStream_t)stream));
      return 0;
    case OP_IN:
      return ncdhw_to_ndhwc_launch(x, op.o0, N, op.i0, op.l0, 0, stream, op.i1);
    case OP_TEMB:
      return time_embed_launch(t, N, u->cfg.model_channels, u->ted, op.f0, op.f1, op.f2, op.f3, op.o0, op.o1, stream);
    case OP_EMBLIN:
      return rows_linear_launch(op.f0, op.f1, op.f2, op.o0, N, u->emb_rows, u->ted, stream);
    case OP_STATS:
      return gn_stats_launch(op.f0, op.dout, N, op.i0, op.l0, stream, op.i1);
    case OP_FINAL:
      return gn_finalize_launch(op.d0, op.i0, op.i4, op.d1, op.i1, op.i5, N, op.l0, 32, 1e-5f, op.f0, op.f1, op.f2,
                                op.i2, op.i3, op.o0, stream, op.o1);
    case OP_CONV:
      return conv_launch(op.conv, stream);
    case OP_GEMM:
      return gemm_launch(op.gemm, stream);
    case OP_SOFTMAX:
      return softmax_rows_launch(op.o0, op.l0, op.i0, stream);
    case OP_FLASH:
      if (op.i0 == 2) return flash_attn_bf16v2_launch(op.attn, op.o1, op.i1, u->ctx->num_cus, stream, op.i2);
      return flash_attn_launch(op.attn, stream);
    case OP_OUT:
      return ndhwc_to_ncdhw_launch(op.f0, y, N, op.i0, op.l0, stream);
  }
  return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int holo_abi_version(void) { return HOLO_ABI_VERSION; }
const char* holo_last_error(void) { return get_error(); }

int holo_ctx_create(int device_id, HoloCtx** out) {
  if (!out) {
    set_error("holo_ctx_create: null out");
    return HOLO_E_INVALID;
  }
  HIP_TRY(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device_id));
  HoloCtx* c = new HoloCtx;
  c->device = device_id;
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (const char* e = getenv("HOLO_NUM_CUS")) {  // test knob: planners size grids / split-K for this many CUs
    const int v = atoi(e);
    if (v > 0) c->num_cus = v;
  }
  *out = c;
  return 0;
}
int holo_ctx_destroy(HoloCtx* ctx) {
  delete ctx;
  return 0;
}
int holo_ctx_set_deterministic(HoloCtx* ctx, int on) {
  if (!ctx) {
    set_error("holo_ctx_set_deterministic: null context");
    return HOLO_E_INVALID;
  }
  ctx->deterministic = on ? 1 : 0;
  return 0;
}
int holo_ctx_get_deterministic(const HoloCtx* ctx) { return ctx ? ctx->deterministic : 0; }

int holo_unet_create(HoloCtx* ctx, const HoloUnetCfg* cfg, HoloUnet** out) {
  if (!ctx || !cfg || !out) {
    set_error("holo_unet_create: null argument");
    return HOLO_E_INVALID;
  }
  if (!cfg->homogeneous_resample) {
    set_error("holo_unet_create: only homogeneous_resample=True is supported");
    return HOLO_E_UNSUPPORTED;
  }
  if (cfg->n_channel_mult < 1 || cfg->n_channel_mult > 8 || cfg->n_attention_resolutions > 8 ||
      cfg->model_channels % 32 || cfg->in_channels % 4 || cfg->out_channels % 4 || cfg->model_channels > 256 ||
      cfg->image_size % (1 << (cfg->n_channel_mult - 1))) {
    set_error("holo_unet_create: unsupported configuration");
    return HOLO_E_UNSUPPORTED;
  }
  HoloUnet* u = new HoloUnet;
  u->ctx = ctx;
  u->cfg = *cfg;
  build_structure(u);
  enumerate_params(u);
  const char* dbg = getenv("HOLO_KEEP_INTERMEDIATES");
  u->keep_intermediates = dbg && dbg[0] == '1';
  // private parameter storage
  int64_t total = 0;
  auto priv_numel = [](const ParamSlot& s) -> int64_t {
    if (s.kind == P_CONV3 || s.kind == P_CONV1)
      return (int64_t)(s.kind == P_CONV3 ? 27 : 1) * pad_cout((int)s.shape[0]) * pad_cin((int)s.shape[1]);
    return s.numel;
  };
  for (auto& s : u->params)
    if (s.kind == P_PLAIN || s.kind == P_CONV3 || s.kind == P_CONV1) total += (priv_numel(s) + 63) & ~(int64_t)63;
  total += ((int64_t)u->emb_rows * u->ted + 63) & ~(int64_t)63;
  total += (u->emb_rows + 63) & ~63;
  if (hipMalloc((void**)&u->pstore, (size_t)total * sizeof(float)) != hipSuccess) {
    set_error("holo_unet_create: hipMalloc of %lld parameter floats failed", (long long)total);
    delete u;
    return HOLO_E_HIP;
  }
  float* cur = u->pstore;
  for (auto& s : u->params)
    if (s.kind == P_PLAIN || s.kind == P_CONV3 || s.kind == P_CONV1) {
      s.priv = cur;
      cur += (priv_numel(s) + 63) & ~(int64_t)63;
    }
  {  // bf16 copies of the conv weights (same padded element counts, 2 bytes each)
    int64_t tb = 0;
    for (auto& s : u->params)
      if (s.kind == P_CONV3 || s.kind == P_CONV1) tb += 4 * ((priv_numel(s) + 63) & ~(int64_t)63);  // hi, mid, lo planes + the 32x32x16 packing of hi
    if (hipMalloc((void**)&u->pstore_bf, (size_t)tb * sizeof(uint16_t)) != hipSuccess) {
      set_error("holo_unet_create: hipMalloc of %lld bf16 weights failed", (long long)tb);
      (void)hipFree(u->pstore);
      delete u;
      return HOLO_E_HIP;
    }
    uint16_t* cb = u->pstore_bf;
    for (auto& s : u->params)
      if (s.kind == P_CONV3 || s.kind == P_CONV1) {
        s.priv_bf = cb;
        u->bf_of[s.priv] = cb;
        u->bft_of[s.priv] = cb + 3 * priv_numel(s);  // plane 3 (the repack kernel lays the planes out back to back)
        cb += 4 * ((priv_numel(s) + 63) & ~(int64_t)63);
      }
  }
  {  // Winograd-in-depth copies for the convolutions that can land on 128-voxel tiles: the wide top levels
     // (a 3x3x3 conv of <= 256 channels: 36 pseudo-taps; a ResBlock's 1x1x1 skip connection: 2 pseudo-taps)
    const char* we = getenv("HOLO_CONV_WINO");
    const bool enable = !(we && we[0] == '0');
    auto wino_numel = [](const ParamSlot& s) -> int64_t {
      const bool c3 = s.kind == P_CONV3;
      const bool sk = s.kind == P_CONV1 && s.name.find("skip_connection") != std::string::npos;
      // (the levels with 8-divisible planes: up to 256 output channels, up to 768 input channels with the skip concat)
      if (!(c3 || sk) || s.shape[0] > 256 || s.shape[1] > 768 || (s.shape[0] % 64 && !(c3 && s.shape[0] == 32))) return 0;
      return (int64_t)(c3 ? 36 : 2) * pad_cout((int)s.shape[0]) * pad_cin((int)s.shape[1]);
    };
    const bool enable2 = enable && !(we && we[0] == '1');  // HOLO_CONV_WINO=1: depth only; default: both forms prepared
    auto wino2_numel = [&](const ParamSlot& s) -> int64_t { return enable2 ? wino_numel(s) / (s.kind == P_CONV3 ? 36 : 2) * (s.kind == P_CONV3 ? 48 : 4) : 0; };
    // F(2x2x2, 3x3x3) copies (conv_wino3_kernel, 64 pseudo-taps / 8 signed skip copies): the levels whose workgroup list
    // can fill the chip - up to 256 output channels (64^3 .. 8^3 in the released nets); HOLO_CONV_WINO3=0: none
    const char* w3e = getenv("HOLO_CONV_WINO3");
    const bool enable3 = enable2 && !(w3e && w3e[0] == '0');
    auto wino3_numel = [&](const ParamSlot& s) -> int64_t {
      if (!enable3 || wino_numel(s) == 0 || s.shape[0] % 64 || s.shape[0] > 256 || s.shape[1] > 768) return 0;
      return conv_wino3_weight_floats(pad_cout((int)s.shape[0]), pad_cin((int)s.shape[1]), s.kind == P_CONV3 ? 27 : 1);
    };
    int64_t tw = 0;
    if (enable)
      for (auto& s : u->params)
        tw += ((wino_numel(s) + 63) & ~(int64_t)63) + ((wino2_numel(s) + 63) & ~(int64_t)63) + ((wino3_numel(s) + 63) & ~(int64_t)63);
    if (tw > 0) {
      if (hipMalloc((void**)&u->pstore_wino, (size_t)tw * sizeof(float)) != hipSuccess) {
        set_error("holo_unet_create: hipMalloc of %lld Winograd weights failed", (long long)tw);
        (void)hipFree(u->pstore);
        (void)hipFree(u->pstore_bf);
        delete u;
        return HOLO_E_HIP;
      }
      float* cw = u->pstore_wino;
      for (auto& s : u->params) {
        const int64_t nw = wino_numel(s);
        if (nw == 0) continue;
        s.priv_wino = cw;
        u->wino_of[s.priv] = cw;
        cw += (nw + 63) & ~(int64_t)63;
        const int64_t nw2 = wino2_numel(s);
        if (nw2) {
          s.priv_wino2 = cw;
          u->wino2_of[s.priv] = cw;
          cw += (nw2 + 63) & ~(int64_t)63;
        }
        const int64_t nw3 = wino3_numel(s);
        if (nw3) {
          s.priv_wino3 = cw;
          u->wino3_of[s.priv] = cw;
          cw += (nw3 + 63) & ~(int64_t)63;
        }
      }
    }
  }
  u->emb_w = cur;
  cur += ((int64_t)u->emb_rows * u->ted + 63) & ~(int64_t)63;
  u->emb_b = cur;
  for (auto& s : u->params) {
    if (s.kind == P_EMB_W || s.kind == P_EMB_B) {
      std::string prefix = s.name.substr(0, s.name.rfind(".emb_layers"));
      int row = u->emb_row_off[prefix];
      s.priv = s.kind == P_EMB_W ? u->emb_w + (int64_t)row * u->ted : u->emb_b + row;
    }
  }
  *out = u;
  return 0;
}

int holo_unet_destroy(HoloUnet* net) {
  if (!net) return 0;
  if (net->pstore) (void)hipFree(net->pstore);
  if (net->pstore_bf) (void)hipFree(net->pstore_bf);
  if (net->pstore_wino) (void)hipFree(net->pstore_wino);
  for (auto& kv : net->dgrad_w)
    if (kv.second) (void)hipFree(kv.second);
  for (auto& kv : net->dgrad_wino)
    if (kv.second) (void)hipFree(kv.second);
  for (auto& kv : net->dgrad_wino3)
    if (kv.second) (void)hipFree(kv.second);
  for (auto& kv : net->dgrad_wino2)
    if (kv.second) (void)hipFree(kv.second);
  if (net->dgrad_tmp) (void)hipFree(net->dgrad_tmp);
  delete net;
  return 0;
}

int holo_unet_num_params(const HoloUnet* net) { return net ? (int)net->params.size() : 0; }

int holo_unet_param_info(const HoloUnet* net, int index, char* name, int name_cap, int64_t shape[8], int* ndim) {
  if (!net || index < 0 || index >= (int)net->params.size()) {
    set_error("holo_unet_param_info: bad index");
    return HOLO_E_INVALID;
  }
  const ParamSlot& s = net->params[index];
  if (name && name_cap > 0) {
    strncpy(name, s.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (ndim) *ndim = (int)s.shape.size();
  if (shape)
    for (size_t i = 0; i < s.shape.size() && i < 8; ++i) shape[i] = s.shape[i];
  return 0;
}

int holo_unet_set_param(HoloUnet* net, const char* name, const void* dev_ptr, int dtype, int ndim,
                        const int64_t* shape, void* stream) {
  if (!net || !name || !dev_ptr) {
    set_error("holo_unet_set_param: null argument");
    return HOLO_E_INVALID;
  }
  if (dtype != HOLO_DTYPE_F32) {
    set_error("holo_unet_set_param: only fp32 parameters are supported");
    return HOLO_E_UNSUPPORTED;
  }
  auto it = net->pindex.find(name);
  if (it == net->pindex.end()) {
    set_error("holo_unet_set_param: unknown parameter '%s'", name);
    return HOLO_E_INVALID;
  }
  ParamSlot& s = net->params[it->second];
  bool ok = ndim == (int)s.shape.size();
  for (int i = 0; ok && i < ndim; ++i) ok = shape[i] == s.shape[i];
  if (!ok) {
    set_error("holo_unet_set_param: shape mismatch for '%s'", name);
    return HOLO_E_INVALID;
  }
  if (s.kind == P_CONV3 || s.kind == P_CONV1) {
    int rc = repack_conv_weight_launch((const float*)dev_ptr, s.priv, (int)s.shape[0], (int)s.shape[1],
                                       s.kind == P_CONV3 ? 27 : 1, pad_cout((int)s.shape[0]), pad_cin((int)s.shape[1]),
                                       stream);
    if (rc) return rc;
    rc = repack_conv_weight_bf16_launch((const float*)dev_ptr, s.priv_bf, (int)s.shape[0], (int)s.shape[1],
                                        s.kind == P_CONV3 ? 27 : 1, pad_cout((int)s.shape[0]), pad_cin((int)s.shape[1]),
                                        stream);
    if (rc) return rc;
    if (s.priv_wino) {
      rc = repack_conv_weight_wino_launch((const float*)dev_ptr, s.priv_wino, (int)s.shape[0], (int)s.shape[1],
                                          s.kind == P_CONV3 ? 27 : 1, pad_cout((int)s.shape[0]),
                                          pad_cin((int)s.shape[1]), stream);
      if (rc) return rc;
    }
    if (s.priv_wino2) {
      rc = repack_conv_weight_wino_launch((const float*)dev_ptr, s.priv_wino2, (int)s.shape[0], (int)s.shape[1],
                                          s.kind == P_CONV3 ? 27 : 1, pad_cout((int)s.shape[0]),
                                          pad_cin((int)s.shape[1]), stream, 2);
      if (rc) return rc;
    }
    if (s.priv_wino3) {
      rc = repack_conv_weight_wino3_launch((const float*)dev_ptr, s.priv_wino3, (int)s.shape[0], (int)s.shape[1],
                                           s.kind == P_CONV3 ? 27 : 1, pad_cout((int)s.shape[0]), pad_cin((int)s.shape[1]),
                                           stream);
      if (rc) return rc;
    }
  } else {  // biases, GroupNorm parameters, Linear layers: a copy kernel with system-scope loads (holo_ld_sys) - like the
            // weight repack kernels, every ingestion of a caller-provided tensor reads it past the L2
    if (copy_sys_launch((const float*)dev_ptr, s.priv, s.numel, stream)) return HOLO_E_INVALID;
  }
  s.set = true;
  return 0;
}

int holo_unet_set_compute_dtype(HoloUnet* net, int dtype) {
  if (!net || (dtype != HOLO_DTYPE_F32 && dtype != HOLO_DTYPE_BF16 && dtype != HOLO_DTYPE_F32_BF16X3)) {
    set_error("holo_unet_set_compute_dtype: HOLO_DTYPE_F32, HOLO_DTYPE_BF16 or HOLO_DTYPE_F32_BF16X3");
    return HOLO_E_INVALID;
  }
  const int mode = dtype == HOLO_DTYPE_BF16 ? 1 : dtype == HOLO_DTYPE_F32_BF16X3 ? 2 : 0;
  if (mode != net->compute_mode) {
    net->compute_mode = mode;
    net->plan_batch = -1;   // re-plan: the conv ops carry the choice
    net->ws_cache.clear();  // ... and the plan's workspace differs between modes (fused skips, statistics slabs)
  }
  return 0;
}

size_t holo_unet_workspace_bytes(HoloUnet* net, int batch) {
  if (!net || batch < 1) return 0;
  auto it = net->ws_cache.find(batch);
  if (it != net->ws_cache.end()) return it->second;
  std::vector<Op> tmp;
  Planner pl(net, batch, nullptr, tmp);
  pl.build();
  size_t b = pl.total_bytes();
  net->ws_cache[batch] = b;
  // the sizing pass overwrote block_outputs with null-based offsets; force a re-plan
  net->plan_batch = -1;
  net->ops.clear();
  return b;
}

int holo_unet_forward(HoloUnet* net, int batch, const float* x, const int64_t* timesteps, float* y, void* workspace,
                      size_t workspace_bytes, void* stream) {
  if (!net || !x || !timesteps || !y || !workspace || batch < 1) {
    set_error("holo_unet_forward: null/invalid argument");
    return HOLO_E_INVALID;
  }
  int rc = ensure_plan(net, batch, workspace);
  if (rc) return rc;
  if (workspace_bytes < net->ws_need) {
    set_error("holo_unet_forward: workspace too small (%zu < %zu)", workspace_bytes, net->ws_need);
    return HOLO_E_WORKSPACE;
  }
  for (const Op& op : net->ops) {
    rc = run_op(net, op, batch, x, timesteps, y, stream);
    if (rc) return rc < 0 ? rc : HOLO_E_INVALID;
  }
  return 0;
}

int holo_unet_forward_cl(HoloUnet* net, int batch, const float* x_cl, const int64_t* timesteps, float* y_cl, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (!net || !x_cl || !timesteps || !y_cl || !workspace || batch < 1) {
    set_error("holo_unet_forward_cl: null/invalid argument");
    return HOLO_E_INVALID;
  }
  const bool bf16_storage = net->compute_mode == 1;  // the plan's input buffer is bf16: a cast replaces the layout pass
  int rc = ensure_plan(net, batch, workspace);
  if (rc) return rc;
  if (workspace_bytes < net->ws_need) {
    set_error("holo_unet_forward_cl: workspace too small (%zu < %zu)", workspace_bytes, net->ws_need);
    return HOLO_E_WORKSPACE;
  }
  // the plan's own input / output buffers: every convolution that reads the one or writes the other is pointed at the
  // caller's channels-last tensors instead, and the two layout passes are skipped
  // (by POSITION in the op list, not by address: the arena hands the input buffer's memory to later activations)
  const float* in_buf = nullptr;
  const float* out_buf = nullptr;
  int first_conv = -1, last_conv = -1;
  for (size_t i = 0; i < net->ops.size(); ++i) {
    const Op& op = net->ops[i];
    if (op.kind == OP_IN) in_buf = op.o0;
    if (op.kind == OP_OUT) out_buf = op.f0;
    if (op.kind == OP_CONV) {
      if (first_conv < 0 && in_buf) first_conv = (int)i;
      last_conv = (int)i;
    }
  }
  if (first_conv < 0 || !in_buf || !out_buf || net->ops[first_conv].conv.src0 != in_buf || net->ops[first_conv].conv.src1 ||
      net->ops[last_conv].conv.out != out_buf || net->ops[last_conv].conv.nsplit != 1 || net->ops[first_conv].conv.nsplit != 1 ||
      first_conv == last_conv) {
    set_error("holo_unet_forward_cl: this plan's first / last convolution cannot take the caller's tensors");
    return HOLO_E_UNSUPPORTED;
  }
  if (bf16_storage && (net->ops[last_conv].conv.out_bf16 || !net->ops[first_conv].conv.in_bf16)) {
    set_error("holo_unet_forward_cl: unexpected storage types at the ends of the bf16 plan");
    return HOLO_E_UNSUPPORTED;
  }
  for (size_t i = 0; i < net->ops.size(); ++i) {
    const Op& op = net->ops[i];
    if (op.kind == OP_IN && bf16_storage) {  // fp32 channels-last -> the plan's bf16 channels-last input buffer
      if (f32_to_bf16_launch(x_cl, op.o0, (int64_t)batch * op.i0 * op.l0, stream)) return HOLO_E_INVALID;
      continue;
    }
    if (op.kind == OP_IN || op.kind == OP_OUT) continue;
    if ((int)i == first_conv || (int)i == last_conv) {
      Op o2 = op;
      if ((int)i == first_conv && !bf16_storage) o2.conv.src0 = x_cl;
      if ((int)i == last_conv) o2.conv.out = y_cl;
      rc = run_op(net, o2, batch, x_cl, timesteps, y_cl, stream);
    } else {
      rc = run_op(net, op, batch, x_cl, timesteps, y_cl, stream);
    }
    if (rc) return rc < 0 ? rc : HOLO_E_INVALID;
  }
  return 0;
}

int holo_unet_fetch_block(HoloUnet* net, const char* tag, float* dst, int64_t dst_capacity, int64_t* numel,
                          void* workspace, void* stream) {
  if (!net || !tag || !dst || !workspace) {
    set_error("holo_unet_fetch_block: null argument");
    return HOLO_E_INVALID;
  }
  if (!net->keep_intermediates) {
    set_error("holo_unet_fetch_block: create the net with HOLO_KEEP_INTERMEDIATES=1");
    return HOLO_E_STATE;
  }
  if (net->plan_ws != workspace || net->ops.empty()) {
    set_error("holo_unet_fetch_block: no forward has run on this workspace");
    return HOLO_E_STATE;
  }
  auto it = net->block_outputs.find(tag);
  if (it == net->block_outputs.end()) {
    set_error("holo_unet_fetch_block: unknown tag '%s'", tag);
    return HOLO_E_INVALID;
  }
  const Act& a = it->second;
  const int64_t V = (int64_t)a.R * a.R * a.R;
  const int64_t n = (int64_t)net->plan_batch * V * a.C;
  if (numel) *numel = n;
  if (dst_capacity < n) {
    set_error("holo_unet_fetch_block: destination too small");
    return HOLO_E_INVALID;
  }
  return ndhwc_to_ncdhw_launch((const float*)((char*)workspace + a.off), dst, net->plan_batch, a.C, V, stream,
                               net->compute_mode == 1 ? 1 : 0);
}

int holo_unet_time_convs(HoloUnet* net, int batch, void* workspace, size_t workspace_bytes, int iters, void* stream,
                         float* total_ms, double* total_flops, int* n_launches) {
  if (!net || !workspace || iters < 1) {
    set_error("holo_unet_time_convs: invalid argument");
    return HOLO_E_INVALID;
  }
  int rc = ensure_plan(net, batch, workspace);
  if (rc) return rc;
  if (workspace_bytes < net->ws_need) {
    set_error("holo_unet_time_convs: workspace too small");
    return HOLO_E_WORKSPACE;
  }
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  double flops = 0.0;
  int launches = 0;
  for (const Op& op : net->ops)
    if (op.kind == OP_CONV && op.conv.ksz == 3) {
      flops += conv_flops(op.conv);
      ++launches;
    }
  HIP_TRY(hipEventRecord(e0, (hipStream_t)stream));
  for (int it = 0; it < iters; ++it)
    for (const Op& op : net->ops)
      if (op.kind == OP_CONV && op.conv.ksz == 3) {
        rc = conv_launch(op.conv, stream);
        if (rc) return HOLO_E_INVALID;
      }
  HIP_TRY(hipEventRecord(e1, (hipStream_t)stream));
  HIP_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (total_ms) *total_ms = ms / iters;
  if (total_flops) *total_flops = flops;
  if (n_launches) *n_launches = launches;
  return 0;
}

int holo_unet_time_ops(HoloUnet* net, int batch, const float* x, const int64_t* timesteps, float* y, void* workspace,
                       size_t workspace_bytes, int iters, void* stream, HoloOpTiming* out, int cap, int* n_ops) {
  if (!net || !workspace || !x || !timesteps || !y || iters < 1 || !n_ops || (cap > 0 && !out)) {
    set_error("holo_unet_time_ops: invalid argument");
    return HOLO_E_INVALID;
  }
  int rc = ensure_plan(net, batch, workspace);
  if (rc) return rc;
  if (workspace_bytes < net->ws_need) {
    set_error("holo_unet_time_ops: workspace too small");
    return HOLO_E_WORKSPACE;
  }
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  int n = 0;
  for (const Op& op : net->ops) {
    rc = run_op(net, op, batch, x, timesteps, y, stream);  // untimed first touch (also keeps the data flow valid)
    if (rc) return HOLO_E_INVALID;
    if (n < cap) {
      HIP_TRY(hipEventRecord(e0, (hipStream_t)stream));
      for (int it = 0; it < iters; ++it) run_op(net, op, batch, x, timesteps, y, stream);
      HIP_TRY(hipEventRecord(e1, (hipStream_t)stream));
      HIP_TRY(hipEventSynchronize(e1));
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
      HoloOpTiming& t = out[n];
      memset(&t, 0, sizeof(t));
      t.op = (int)op.kind;
      t.ms = ms / iters;
      if (op.kind == OP_CONV) {
        const ConvParams& c = op.conv;
        t.kernel = c.mode == 6 ? 11 : c.mode == 5 ? 10 : c.mode == 4 ? 9 : c.mode == 3 ? 7 : (c.bf16t && c.bf16p) ? 8 : c.bf16t ? 5 : c.wino == 3 ? 6 : c.wino == 2 ? 4 : c.wino ? 3 : c.mode;
        t.tile_depth = c.mode == 1 ? c.tz : 0;
        t.fused_skip = c.skip_w ? 1 : 0;
        t.nsplit = c.nsplit;
        t.cin = c.C0 + c.C1;
        t.cout = c.Cout;
        t.out_dim = c.OD;
        t.stride = c.stride;
        t.upsample = c.ups;
        t.ksz = c.ksz;
        t.flops = conv_flops(c);
        t.flops_executed = conv_exec_flops(c);
      } else if (op.kind == OP_FLASH) {
        t.cin = t.cout = op.attn.C;
        t.out_dim = op.attn.T;
        t.flops = 4.0 * op.attn.N * (double)op.attn.T * op.attn.T * op.attn.C;
      } else if (op.kind == OP_GEMM) {
        t.cin = op.gemm.K;
        t.cout = op.gemm.Nn;
        t.out_dim = op.gemm.M;
        t.flops = 2.0 * op.gemm.nb0 * op.gemm.nb1 * (double)op.gemm.M * op.gemm.Nn * op.gemm.K;
      }
    }
    ++n;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *n_ops = n;
  return 0;
}

// ---- training: backward of the denoiser -----------------------------------------------------------------------------
int holo_unet_set_dgrad_weight(HoloUnet* net, const char* name, const void* dev_ptr, void* stream) {
  if (!net || !name || !dev_ptr) {
    set_error("holo_unet_set_dgrad_weight: null argument");
    return HOLO_E_INVALID;
  }
  auto it = net->pindex.find(name);
  if (it == net->pindex.end() || (net->params[it->second].kind != P_CONV3 && net->params[it->second].kind != P_CONV1)) {
    set_error("holo_unet_set_dgrad_weight: '%s' is not a convolution weight", name);
    return HOLO_E_INVALID;
  }
  const ParamSlot& s = net->params[it->second];
  const int Co = (int)s.shape[0], Ci = (int)s.shape[1], T = s.kind == P_CONV3 ? 27 : 1;
  const std::string nm(name);
  const bool down = nm.size() > 10 && nm.compare(nm.size() - 10, 10, ".op.weight") == 0;  // Downsample: stride 2
  // transposed convolution: Cout' = Ci, Cin' = Co
  const size_t packed = down ? (size_t)T * Co * Ci : (size_t)T * pad_cout(Ci) * pad_cin(Co);
  if (down) {  // [tap][co][ci] for conv_dgrad_s2_kernel (the fallback), then the stride-1 form below under "<name>#s1":
               // the transposed stride-2 convolution runs as zero insertion + the stride-1 transposed convolution
    float*& d2 = net->dgrad_w[nm];
    if (!d2) HIP_TRY(hipMalloc((void**)&d2, packed * sizeof(float)));
    if (weight_tco_ci_launch((const float*)dev_ptr, d2, Co, Ci, T, stream)) return HOLO_E_INVALID;
    if ((Co & 3) || (Ci & 3)) return 0;
  }
  const std::string key = down ? nm + "#s1" : nm;
  float*& dst = net->dgrad_w[key];
  if (!dst) HIP_TRY(hipMalloc((void**)&dst, (size_t)T * pad_cout(Ci) * pad_cin(Co) * sizeof(float)));
  if (net->dgrad_tmp_floats < (size_t)s.numel) {
    if (net->dgrad_tmp) {
      HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
      (void)hipFree(net->dgrad_tmp);
    }
    HIP_TRY(hipMalloc((void**)&net->dgrad_tmp, (size_t)s.numel * sizeof(float)));
    net->dgrad_tmp_floats = (size_t)s.numel;
  }
  if (flip_transpose_weight_launch((const float*)dev_ptr, net->dgrad_tmp, Co, Ci, T, stream)) return HOLO_E_INVALID;
  if (repack_conv_weight_launch(net->dgrad_tmp, dst, Ci, Co, T, pad_cout(Ci), pad_cin(Co), stream)) return HOLO_E_INVALID;
  // Winograd copies of the transposed convolution (the dgrad of a wide-level 3x3x3 conv runs on conv_wino2_kernel like
  // the forward conv: 36 + 48 pseudo-taps, same eligibility as holo_unet_create's except that the transposed conv's output
  // channels are the forward conv's INPUT channels, up to 768 with the skip concat)
  static const char* we = getenv("HOLO_CONV_WINO");
  const bool wino_on = !(we && (we[0] == '0' || we[0] == '1'));
  if (wino_on && net->compute_mode == 0 && T == 27 && (Ci % 64) == 0 && Ci <= 768 && Co <= 768) {
    const size_t per_tap = (size_t)pad_cout(Ci) * pad_cin(Co);
    float*& w1 = net->dgrad_wino[key];
    float*& w2 = net->dgrad_wino2[key];
    if (!w1 || !w2) {  // a plan sized before these copies existed chose other kernels (and scratch sizes)
      net->tws_cache.clear();
      net->tplan_batch = -1;
    }
    if (!w1) HIP_TRY(hipMalloc((void**)&w1, 36 * per_tap * sizeof(float)));
    if (!w2) HIP_TRY(hipMalloc((void**)&w2, 48 * per_tap * sizeof(float)));
    if (repack_conv_weight_wino_launch(net->dgrad_tmp, w1, Ci, Co, 27, pad_cout(Ci), pad_cin(Co), stream, 1)) return HOLO_E_INVALID;
    if (repack_conv_weight_wino_launch(net->dgrad_tmp, w2, Ci, Co, 27, pad_cout(Ci), pad_cin(Co), stream, 2)) return HOLO_E_INVALID;
    net->wino_of[dst] = w1;
    net->wino2_of[dst] = w2;
    // ... and on conv_wino3_kernel where the forward convolutions do (transposed: output channels = the forward's inputs)
    static const char* w3e = getenv("HOLO_CONV_WINO3");
    if (!(w3e && w3e[0] == '0') && Ci <= 256 && Co <= 768) {
      float*& w3 = net->dgrad_wino3[key];
      if (!w3) {
        net->tws_cache.clear();
        net->tplan_batch = -1;
        HIP_TRY(hipMalloc((void**)&w3, (size_t)conv_wino3_weight_floats(pad_cout(Ci), pad_cin(Co), 27) * sizeof(float)));
      }
      if (repack_conv_weight_wino3_launch(net->dgrad_tmp, w3, Ci, Co, 27, pad_cout(Ci), pad_cin(Co), stream)) return HOLO_E_INVALID;
      net->wino3_of[dst] = w3;
    }
  }
  return 0;
}

size_t holo_unet_backward_workspace_bytes(HoloUnet* net, int batch) {
  if (!net || batch < 1) return 0;
  auto it = net->tws_cache.find(batch);
  if (it != net->tws_cache.end()) return it->second;
  std::vector<Op> fo;
  std::vector<std::function<int(void*)>> bo;
  std::vector<size_t> keep = net->grad_off;
  TrainPlanner tp(net, batch, nullptr, fo, bo);
  // a sizing pass must not fail on missing transposed weights: it only allocates
  std::map<std::string, float*> saved = net->dgrad_w;
  for (auto& s : net->params)
    if (s.kind == P_CONV3 || s.kind == P_CONV1) {
      if (!net->dgrad_w.count(s.name)) net->dgrad_w[s.name] = (float*)(uintptr_t)256;
      const bool down = s.name.size() > 10 && s.name.compare(s.name.size() - 10, 10, ".op.weight") == 0;
      if (down && !(s.shape[0] & 3) && !(s.shape[1] & 3) && !net->dgrad_w.count(s.name + "#s1"))
        net->dgrad_w[s.name + "#s1"] = (float*)(uintptr_t)256;
    }
  const int rc = tp.build();
  net->dgrad_w = saved;
  net->grad_off = keep;
  if (rc) return 0;  // the message is in holo_last_error()
  const size_t b = tp.total_bytes();
  net->tws_cache[batch] = b;
  net->tplan_batch = -1;
  net->plan_batch = -1;
  net->ops.clear();
  return b;
}

// The two halves of holo_unet_backward as entries of their own (ABI 4): a caller whose cotangent depends on the output - the
// clamp of pred_xstart in HoloDiffusionModel.training_backward - runs the taped forward, forms grad_out from y, then the backward,
// instead of paying a plain forward first.
int holo_unet_forward_train(HoloUnet* net, int batch, const float* x, const int64_t* timesteps, float* y, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (!net || !x || !timesteps || !workspace || batch < 1) {
    set_error("holo_unet_forward_train: null/invalid argument");
    return HOLO_E_INVALID;
  }
  net->tape_valid = false;
  int rc = ensure_train_plan(net, batch, workspace);
  if (rc) return rc;
  if (workspace_bytes < net->tws_need) {
    set_error("holo_unet_forward_train: workspace too small (%zu < %zu)", workspace_bytes, net->tws_need);
    net->tplan_batch = -1;
    return HOLO_E_WORKSPACE;
  }
  net->t_dev = timesteps;
  for (const Op& op : net->tops) {
    if (op.kind == OP_OUT && !y) continue;
    rc = run_op(net, op, batch, x, timesteps, y, stream);
    if (rc) return rc < 0 ? rc : HOLO_E_INVALID;
  }
  net->tape_valid = true;
  return 0;
}

int holo_unet_backward_taped(HoloUnet* net, int batch, const float* grad_out, float* grad_x, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (!net || !grad_out || !workspace || batch < 1) {
    set_error("holo_unet_backward_taped: null/invalid argument");
    return HOLO_E_INVALID;
  }
  if (!net->tape_valid || net->tplan_batch != batch || net->tplan_ws != workspace || workspace_bytes < net->tws_need) {
    set_error("holo_unet_backward_taped: no taped forward of this batch on this workspace (holo_unet_forward_train first)");
    return HOLO_E_STATE;
  }
  net->tape_valid = false;  // the backward consumes the tape (gradient buffers share its workspace)
  int rc;
  const HoloUnetCfg& c = net->cfg;
  const int64_t V = (int64_t)c.image_size * c.image_size * c.image_size;
  if (ncdhw_to_ndhwc_launch(grad_out, (float*)((char*)workspace + net->gy_off), batch, c.out_channels, V, 0, stream))
    return HOLO_E_INVALID;
  for (auto& f : net->bops) {
    rc = f(stream);
    if (rc) return rc < 0 ? rc : HOLO_E_INVALID;
  }
  if (grad_x &&
      ndhwc_to_ncdhw_launch((const float*)((char*)workspace + net->gx_off), grad_x, batch, c.in_channels, V, stream))
    return HOLO_E_INVALID;
  return 0;
}

int holo_unet_backward(HoloUnet* net, int batch, const float* x, const int64_t* timesteps, const float* grad_out, float* y,
                       float* grad_x, void* workspace, size_t workspace_bytes, void* stream) {
  if (!net || !x || !timesteps || !grad_out || !workspace || batch < 1) {
    set_error("holo_unet_backward: null/invalid argument");
    return HOLO_E_INVALID;
  }
  net->tape_valid = false;
  int rc = ensure_train_plan(net, batch, workspace);
  if (rc) return rc;
  if (workspace_bytes < net->tws_need) {
    set_error("holo_unet_backward: workspace too small (%zu < %zu)", workspace_bytes, net->tws_need);
    net->tplan_batch = -1;
    return HOLO_E_WORKSPACE;
  }
  net->t_dev = timesteps;
  float ydummy;
  (void)ydummy;
  for (const Op& op : net->tops) {
    if (op.kind == OP_OUT && !y) continue;
    rc = run_op(net, op, batch, x, timesteps, y, stream);
    if (rc) return rc < 0 ? rc : HOLO_E_INVALID;
  }
  const HoloUnetCfg& c = net->cfg;
  const int64_t V = (int64_t)c.image_size * c.image_size * c.image_size;
  if (ncdhw_to_ndhwc_launch(grad_out, (float*)((char*)workspace + net->gy_off), batch, c.out_channels, V, 0, stream))
    return HOLO_E_INVALID;
  for (auto& f : net->bops) {
    rc = f(stream);
    if (rc) return rc < 0 ? rc : HOLO_E_INVALID;
  }
  if (grad_x &&
      ndhwc_to_ncdhw_launch((const float*)((char*)workspace + net->gx_off), grad_x, batch, c.in_channels, V, stream))
    return HOLO_E_INVALID;
  return 0;
}

int holo_unet_get_grad(HoloUnet* net, const char* name, float* dst, int64_t numel, const void* workspace, void* stream) {
  if (!net || !name || !dst || !workspace) {
    set_error("holo_unet_get_grad: null argument");
    return HOLO_E_INVALID;
  }
  auto it = net->pindex.find(name);
  if (it == net->pindex.end()) {
    set_error("holo_unet_get_grad: unknown parameter '%s'", name);
    return HOLO_E_INVALID;
  }
  if (net->tplan_ws != workspace || net->grad_off.size() != net->params.size()) {
    set_error("holo_unet_get_grad: no backward pass has run on this workspace");
    return HOLO_E_STATE;
  }
  const ParamSlot& s = net->params[it->second];
  if (numel != s.numel) {
    set_error("holo_unet_get_grad: '%s' has %lld elements, not %lld", name, (long long)s.numel, (long long)numel);
    return HOLO_E_INVALID;
  }
  HIP_TRY(hipMemcpyAsync(dst, (const char*)workspace + net->grad_off[it->second], (size_t)numel * sizeof(float),
                         hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

int holo_ddpm_step(HoloCtx* ctx, const float* tables, int num_timesteps, const int64_t* timesteps, int batch,
                   int64_t elems_per_sample, const float* x_t, const float* model_out, const float* noise,
                   int clip_denoised, float* sample, float* pred_xstart, void* stream) {
  if (!tables || !timesteps || !x_t || !model_out || !noise || !sample || !pred_xstart || batch < 1) {
    set_error("holo_ddpm_step: null/invalid argument");
    return HOLO_E_INVALID;
  }
  (void)ctx;
  int rc = ddpm_step_launch(tables, num_timesteps, timesteps, batch, elems_per_sample, x_t, model_out, noise,
                            clip_denoised, sample, pred_xstart, stream);
  return rc ? HOLO_E_INVALID : 0;
}

int holo_ddpm_step_philox(HoloCtx* ctx, const float* tables, int num_timesteps, const int64_t* timesteps, int batch,
                          int64_t elems_per_sample, const float* x_t, const float* model_out, uint64_t seed,
                          uint64_t stream_offset, int clip_denoised, float* sample, float* pred_xstart, float* noise_out,
                          int ncdhw_channels, void* stream) {
  if (!tables || !timesteps || !x_t || !model_out || !sample || batch < 1) {
    set_error("holo_ddpm_step_philox: null/invalid argument");
    return HOLO_E_INVALID;
  }
  (void)ctx;
  int rc = ddpm_step_philox_launch(tables, num_timesteps, timesteps, batch, elems_per_sample, x_t, model_out, seed,
                                   stream_offset, clip_denoised, sample, pred_xstart, noise_out, ncdhw_channels, stream);
  return rc ? HOLO_E_INVALID : 0;
}

int holo_tanh(HoloCtx* ctx, const float* x, float* y, int64_t n, void* stream) {
  (void)ctx;
  return tanh_launch(x, y, n, stream);
}
int holo_clip(HoloCtx* ctx, const float* x, float* y, float lo, float hi, int64_t n, void* stream) {
  (void)ctx;
  return clip_launch(x, y, lo, hi, n, stream);
}

int holo_event_timer_create(void** timer) {
  hipEvent_t* ev = new hipEvent_t[2];
  if (hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess) {
    set_error("hipEventCreate failed");
    return HOLO_E_HIP;
  }
  *timer = ev;
  return 0;
}
int holo_event_timer_start(void* timer, void* stream) {
  HIP_TRY(hipEventRecord(((hipEvent_t*)timer)[0], (hipStream_t)stream));
  return 0;
}
int holo_event_timer_stop(void* timer, void* stream, float* elapsed_ms) {
  hipEvent_t* ev = (hipEvent_t*)timer;
  HIP_TRY(hipEventRecord(ev[1], (hipStream_t)stream));
  HIP_TRY(hipEventSynchronize(ev[1]));
  HIP_TRY(hipEventElapsedTime(elapsed_ms, ev[0], ev[1]));
  return 0;
}
int holo_event_timer_destroy(void* timer) {
  hipEvent_t* ev = (hipEvent_t*)timer;
  (void)hipEventDestroy(ev[0]);
  (void)hipEventDestroy(ev[1]);
  delete[] ev;
  return 0;
}

}  // extern "C"
