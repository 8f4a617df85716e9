// Host side of the C ABI (include/beach_seg_amd.h): workspace plan + launch sequences of the SegGPT forward
// (HF:modeling_seggpt.py:831-951) and of the dgrad-only backward the reference obtains from autograd
// (src/model.py:233-269 with every weight frozen, src/util/ml_util.py:9-10).  No allocation, no
// synchronisation: every launch goes to the caller's stream.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>
#include <unordered_map>
#include <mutex>

#include "../../include/beach_seg_amd.h"
#include <algorithm>
#include "attention.hpp"
#include "decoder.hpp"
#include "frontend.hpp"
#include "gemm.hpp"
#include "loss.hpp"
#include "rowops.hpp"

static thread_local std::string g_err;
static int fail(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}
#define CHECK_LAUNCH()                                                        \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) return fail("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
  } while (0)

struct Region { std::string name; int layer; size_t off, bytes; };
struct Plan {
  std::vector<Region> r;
  size_t total = 0;
  void add(const char* name, int layer, size_t bytes) {
    r.push_back({name, layer, total, bytes});
    total += (bytes + 255) & ~(size_t)255;
  }
  const Region* find(const char* name, int layer) const {
    for (auto& x : r) if (x.name == name && x.layer == layer) return &x;
    return nullptr;
  }
};

enum ProfCat { PC_GEMM = 0, PC_ATTN_FWD, PC_ATTN_BWD_DQ, PC_ATTN_BWD_DKV, PC_CONV, PC_ROW, PC_COUNT };
struct ProfRec { int cat; double flops; hipEvent_t e0, e1; };

struct bsg_model {
  bsg_config c;
  std::vector<const void*> w;
  int hp, wp, N, npad, es, relcat_rows;
  bool prof = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get_event() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
  }
  const void* gw(int i) const { return w[i]; }
  const void* lw(int l, int i) const { return w[BSG_GLOBAL_WEIGHTS + BSG_LAYER_WEIGHTS * l + i]; }
  bool is_tap(int l, int* ti) const {
    for (int i = 0; i < c.num_taps; ++i) if (c.taps[i] == l) { *ti = i; return true; }
    return false;
  }
  int streams(int l, int B) const { return l <= c.merge_index ? 2 * B : B; }
  // first_row of the last saving forward on each workspace: a backward over MORE rows than its forward computed would read
  // decoder activations that were never written (bsg_forward_rows / bsg_backward_rows) -- refused, not computed
  std::mutex fwd_rows_mu;
  std::unordered_map<const void*, int> fwd_rows;
};

static Plan make_plan(const bsg_model* m, int B, int train) {
  Plan p;
  const size_t es = m->es, N = m->N, D = m->c.hidden_size, L = m->c.num_layers, nh = m->c.num_heads,
               mlp = m->c.mlp_dim, npad = m->npad, hp = m->hp, nt = m->c.num_taps;
  const size_t HW = (size_t)m->c.canvas_h * m->c.canvas_w, dc = m->c.decoder_hidden;
  const size_t r2 = 2 * (size_t)B * N, r1 = (size_t)B * N;
  const size_t ek = m->c.embed_split ? 3 : 1;  // split-precision patch embed: K = 3 x 768 ([hi | hi | lo] x [W_hi | W_lo | W_hi])
  p.add("patch_a", -1, r2 * 768 * ek * es);
  if (train) {
    for (size_t l = 0; l < L; ++l) {
      const size_t rows = (size_t)m->streams(l, B) * N, S = m->streams(l, B);
      p.add("x_in", l, rows * D * 4);
      p.add("x_mid", l, rows * D * 4);
      p.add("qkv", l, rows * 3 * D * es);
      p.add("attn_o", l, rows * D * es);
      p.add("lse2", l, S * nh * npad * 4);
      p.add("h_pre", l, rows * mlp * es);
    }
    p.add("x_in", L, r1 * D * 4);
    p.add("x_tmp", -1, r2 * D * 4);
    p.add("conv_out", -1, (size_t)B * HW * dc * es);
  } else {
    p.add("x_a", -1, r2 * D * 4);
    p.add("x_b", -1, r2 * D * 4);
    p.add("x_c", -1, r2 * D * 4);
    p.add("qkv", -1, r2 * 3 * D * es);
    p.add("attn_o", -1, r2 * D * es);
    p.add("lse2", -1, 2 * (size_t)B * nh * npad * 4);
  }
  p.add("ln_out", -1, r2 * D * es);
  p.add("h_act", -1, r2 * mlp * es);
  p.add("vt", -1, 2 * (size_t)B * nh * 64 * npad * es);
  p.add("relh_s", -1, 2 * (size_t)B * nh * hp * npad * 4);  // forward: key-major relh scratch [stream][head][kh][token]
  p.add("taps", -1, r1 * nt * D * es);
  p.add("feat", -1, (size_t)B * HW * dc * es);
  if (train) {
    p.add("feat2", -1, (size_t)B * HW * dc * es);
    p.add("dtaps", -1, r1 * nt * D * es);
    p.add("dx", -1, r1 * D * 4);
    p.add("dx_t", -1, r1 * D * es);
    p.add("dh", -1, r1 * mlp * es);
    p.add("dn_a", -1, r1 * D * es);
    p.add("dn_b", -1, r1 * D * es);
    p.add("dqkv", -1, r1 * 3 * D * es);
    p.add("kt", -1, (size_t)B * nh * 64 * npad * es);
    p.add("qt", -1, (size_t)B * nh * 64 * npad * es);
    p.add("dot", -1, (size_t)B * nh * 64 * npad * es);
    p.add("delta", -1, (size_t)B * nh * npad * 4);
    p.add("relhT", -1, (size_t)B * nh * hp * npad * 4);
    p.add("relwT", -1, (size_t)B * nh * 32 * npad * 4);
    if (m->c.embed_split) p.add("dx_split", -1, (size_t)B * (N / 2) * 3 * D * es);
    p.add("gscale", -1, 256);  // f16 / x3: f32 [0] S, [1] 1/S, [8] max |grad_pred| (bits); i32 [16..22] the overflow guard's state
                               // (rowops.hpp grad_scale_kernel: flag, back-off exponent, clean backwards, overflows, ...)
  }
  return p;
}

// Optional per-launch timing with HIP events on the caller's stream (bench.py's roofline leg).
struct ProfScope {
  bsg_model* m; hipStream_t st; ProfRec r; bool on;
  ProfScope(const bsg_model* m_, hipStream_t st_, int cat, double flops) : m((bsg_model*)m_), st(st_), on(m_->prof) {
    if (on) { r.cat = cat; r.flops = flops; r.e0 = m->get_event(); r.e1 = m->get_event(); (void)hipEventRecord(r.e0, st); }
  }
  ~ProfScope() { if (on) { (void)hipEventRecord(r.e1, st); m->recs.push_back(r); } }
};

template <typename K> static void allow_lds(K kernel, int bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// ------------------------------------------------------------------------------------------- launch helpers
template <typename T, int AM, int EPI> static void gemm(const bsg_model* m, GemmArgs g, hipStream_t st) {
  static bool once = (allow_lds(gemm_nt_kernel<T, AM, EPI>, 65536), allow_lds(gemm_nt_kernel_v2<T, AM, EPI>, 3 * 49152),
                      allow_lds(gemm_nt_kernel_v3<T, AM, EPI>, 131072), true);
  (void)once;
  if constexpr (sizeof(T) == 2 && AM == A_PLAIN && gemm_v5_pick<EPI>()) {
    static bool once5 = (allow_lds(gemm_nt_kernel_v5<T, EPI>, 131072), true);
    (void)once5;
  }
  const bool plain_rows = AM == A_PLAIN && g.a_rpg <= 0;
  if (g.a_rpg <= 0) { g.a_rpg = AM == A_FEAT ? g.tokens : (g.M > 0 ? g.M : 1); g.a_gstride = 0; }
  if constexpr (sizeof(T) == 4) {
    if (m->c.gemm_x3) { g.x3 = 1; g.acc_scale = 1.0f / 32.0f; }  // Linear weights are stored x 2^5 in this mode (header)
  }
  ProfScope ps(m, st, PC_GEMM, 2.0 * g.M * g.N * g.K);
  // Tile quantisation: with 256 x 256 tiles on 256 CUs a launch of R.f rounds pays ceil(R.f).  When the last round is
  // thin (f < 0.3) and the epilogue addresses rows plainly, the rows of that round go to the 128 x 128 kernel
  // instead (2 blocks per CU, 4x more blocks): e.g. M = 100352, N = 1024: 6 full rounds + 128 small tiles.
  constexpr bool kRowPlainEpi = EPI == EPI_PLAIN || EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_GELU_FWD || EPI == EPI_BIAS_RESID || EPI == EPI_GELU_BWD;
  if (kRowPlainEpi && plain_rows && g.o_rpg == 0 && g.N > 192) {
    const long tn = (g.N + 255) / 256, tm = (g.M + 255) / 256, tiles = tm * tn;
    const long full = tiles / 256, rem = tiles % 256;
    const long tm_main = (full * 256) / tn;  // row tiles that fit in the full rounds
    if (full >= 3 && rem > 0 && rem < 77 && tm_main > 0 && tm_main < tm) {
      GemmArgs a = g, b = g;
      a.M = (int)(tm_main * 256);
      b.M = g.M - a.M;
      b.A = (const char*)g.A + (long)a.M * g.lda * sizeof(T);
      const size_t oes = EPI == EPI_BIAS_RESID ? 4 : sizeof(T);
      b.out = (char*)g.out + (long)a.M * g.ldo * oes;
      if (g.out2) b.out2 = (char*)g.out2 + (long)a.M * g.ldo * oes;
      if (g.aux) b.aux = (const char*)g.aux + (long)a.M * g.ldaux * (EPI == EPI_BIAS_RESID ? 4 : sizeof(T));
      a.a_rpg = a.M; b.a_rpg = b.M;
      if constexpr (sizeof(T) == 4) {  // x3: only the X3 form of the tail kernel reads the pre-split (hi | lo) weights
        if (g.x3 && (AM != A_PLAIN || g.K / (8 * Traits<T>::EPC) < 3)) { launch_gemm<T, AM, EPI>(g, st); return; }
      }
      launch_gemm<T, AM, EPI>(a, st);
      const int tiles_b = ((b.M + 127) / 128) * ((b.N + 127) / 128);
      if constexpr (sizeof(T) == 4 && AM == A_PLAIN) {
        if (g.x3) {
          static bool once_x = (allow_lds(gemm_nt_tail_kernel<T, EPI, 128, true>, 131072), allow_lds(gemm_nt_tail_kernel<T, EPI, 64, true>, 98304), true);
          (void)once_x;
          if (tiles_b <= 128)
            hipLaunchKernelGGL((gemm_nt_tail_kernel<T, EPI, 64, true>), dim3(((b.M + 63) / 64) * ((b.N + 127) / 128), 1), dim3(256), 98304, st, b);
          else
            hipLaunchKernelGGL((gemm_nt_tail_kernel<T, EPI, 128, true>), dim3(tiles_b, 1), dim3(256), 131072, st, b);
          return;
        }
      }
      if constexpr (AM == A_PLAIN) {
        if (b.K / (8 * Traits<T>::EPC) >= 3) {  // the four-stage pipeline's prologue requests three K tiles
          static bool once_t = (allow_lds(gemm_nt_tail_kernel<T, EPI, 128>, 131072), allow_lds(gemm_nt_tail_kernel<T, EPI, 64>, 98304), true);
          (void)once_t;
          if (tiles_b <= 128)  // half of the CUs would idle: 64-row tiles, twice the workgroups
            hipLaunchKernelGGL((gemm_nt_tail_kernel<T, EPI, 64>), dim3(((b.M + 63) / 64) * ((b.N + 127) / 128), 1), dim3(256), 98304, st, b);
          else
            hipLaunchKernelGGL((gemm_nt_tail_kernel<T, EPI, 128>), dim3(tiles_b, 1), dim3(256), 131072, st, b);
          return;
        }
      }
      hipLaunchKernelGGL((gemm_nt_kernel<T, AM, EPI>), dim3(tiles_b, 1), dim3(256), 65536, st, b);
      return;
    }
  }
  launch_gemm<T, AM, EPI>(g, st);
}

// 3x3 conv launch: 16-bit dtypes take the persistent ring kernel (decoder.hpp), f32 the halo-tile kernel.
static int conv_row_split(int base_items, int steps_total) {
  int best = 1;
  double best_t = 1e30;
  for (int n = 1; n <= 16 && n <= steps_total; ++n) {
    const long items = (long)base_items * n, rounds = (items + 255) / 256;
    const double t = rounds * ((double)steps_total / n + 1.2);  // + pipeline fill of an item (10 rows before the first MFMA)
    if (t < best_t - 1e-9) { best_t = t; best = n; }
  }
  return best;
}
template <typename T, int MODE, int DC> static void launch_conv_halo(const ConvArgs& a, int B, int ty0, hipStream_t st, bool x3 = false) {
  typedef ConvGeo<DC> G;
  constexpr int lds = G::HALO * 64 * sizeof(T);
  ConvArgs b = a;
  b.ty0 = ty0;
  const dim3 grid(a.W / 32, (a.H - ty0 * CONV_TR) / G::TR, B);
  if constexpr (sizeof(T) == 4) {
    if (x3) {  // float32 storage, three f16 MFMAs per product (bsg_config.gemm_x3)
      static bool once3 = (allow_lds(conv3x3_kernel<T, MODE, DC, true>, lds), true);
      (void)once3;
      hipLaunchKernelGGL((conv3x3_kernel<T, MODE, DC, true>), grid, dim3(256), lds, st, b);
      return;
    }
  }
  static bool once = (allow_lds(conv3x3_kernel<T, MODE, DC>, lds), true);
  (void)once;
  hipLaunchKernelGGL((conv3x3_kernel<T, MODE, DC>), grid, dim3(256), lds, st, b);
}
template <typename T, int MODE> static void launch_conv(const ConvArgs& a, int dc, int B, int ty0, hipStream_t st, bool x3 = false) {
  if (dc == 128) {  // BASELINE config 5: the 288 KB filter bank does not fit the ring kernel's LDS -> halo-tile kernel, two K passes
    launch_conv_halo<T, MODE, 128>(a, B, ty0, st, x3);
    return;
  }
  if constexpr (sizeof(T) == 2) {
    static bool once = (allow_lds(conv3x3_ring8_kernel<T, MODE>, CR_LDS), true);
    (void)once;
    ConvRingArgs r{};
    r.c = a; r.y_begin = ty0 * CONV_TR; r.batch = B;
    const int strips = a.W / 32, steps_total = (a.H - r.y_begin) / CR_ROWS;
    r.nsplit = conv_row_split(B * strips, steps_total);
    const int items = B * strips * r.nsplit;
    hipLaunchKernelGGL((conv3x3_ring8_kernel<T, MODE>), dim3(std::min(items, 256)), dim3(512), CR_LDS, st, r);
    return;
  }
  launch_conv_halo<T, MODE, 64>(a, B, ty0, st, x3);
}

// dK/dV launch: 16-bit dtypes stream 128-query tiles when the padded statistics rows hold the rounded-up length
// the one-wave-per-SIMD form (attention_kv4.hpp, its own translation unit attention_kv4.hip): 16-bit dtypes
bool bsg_dkv4_ok(const AttnBwdKvArgs& k);
void bsg_launch_dkv4(const AttnBwdKvArgs& k, int dtype_bf16, hipStream_t st);
template <typename T> static void launch_dkv4(const AttnBwdKvArgs& k, hipStream_t st) {
  if constexpr (sizeof(T) == 2) bsg_launch_dkv4(k, std::is_same<T, bf16_t>::value, st);
}
// variant: 0 = default (16-bit dtypes: the one-wave-per-SIMD kernel where the token grid allows it, else eight waves),
// 1 = eight waves (one workgroup per CU), 2 = two four-wave workgroups per CU (measured 3 % SLOWER than eight waves: twice
// the LDS-DMA per key row and a barrier per 64 queries; kept for A/B runs through bsg_op_attention only)
template <typename T> static void launch_dkv(const AttnBwdKvArgs& k, hipStream_t st, bool x3 = false, int variant = 0) {
  constexpr bool tr = sizeof(T) == 2;
  if constexpr (tr) {
    if (variant == 0 && bsg_dkv4_ok(k)) {
      launch_dkv4<T>(k, st);
      return;
    }
  }
  const int krows = k.key_rows > 0 && k.key_rows < k.hp ? k.key_rows : k.hp;  // row windows: see AttnBwdKvArgs
  const dim3 kgrid(((krows + 7) / 8) * k.nh * k.S);
  if constexpr (!tr) {
    if (x3) {  // exact-f32 storage, three f16 MFMAs per f32 MFMA quadruple (attention.hpp mma32_x3)
      constexpr int lds3 = 2 * DkvK<T, false, 64>::STAGE;
      static bool once3 = (allow_lds(attn_bwd_dkv_kernel<T, false, 64, true>, lds3), true);
      (void)once3;
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, false, 64, true>), kgrid, dim3(512), lds3, st, k);
      return;
    }
  }
  if constexpr (tr) {
    if (variant == 2 && k.hp % 4 == 0 && ((k.N + 63) & ~63) <= k.hp * 32) {  // two four-wave workgroups per CU (attention.hpp, NW = 4)
      constexpr int lds = 2 * DkvK<T, true, 64>::STAGE;
      static bool once4 = (allow_lds(attn_bwd_dkv_kernel<T, true, 64, false, 4>, lds), true);
      (void)once4;
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, true, 64, false, 4>), dim3(((krows + 3) / 4) * k.nh * k.S), dim3(256), lds, st, k);
      return;
    }
    if (((k.N + 127) & ~127) <= k.hp * 32) {
      constexpr int lds = 2 * DkvK<T, true, 128>::STAGE;
      static bool once = (allow_lds(attn_bwd_dkv_kernel<T, true, 128>, lds), true);
      (void)once;
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, true, 128>), kgrid, dim3(512), lds, st, k);
      return;
    }
  }
  constexpr int lds = 2 * DkvK<T, tr, 64>::STAGE;
  static bool once = (allow_lds(attn_bwd_dkv_kernel<T, tr, 64>, lds), true);
  (void)once;
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, tr, 64>), kgrid, dim3(512), lds, st, k);
}

template <typename T> struct Ctx {
  const bsg_model* m;
  hipStream_t st;
  char* ws;
  Plan plan;
  int B;
  template <typename U = void> U* at(const char* name, int layer = -1) const {
    const Region* r = plan.find(name, layer);
    return r ? (U*)(ws + r->off) : nullptr;
  }
};

template <typename T> static void ln_fwd(const Ctx<T>& c, const float* x, const void* g, const void* b, T* y, long ldy, int rows,
                                         int in_rpg = 0, long in_gstride = 0, long in_off = 0) {
  if (c.m->c.hidden_size <= 1024)  // no masked loads past D: 114.5 -> 109.2 us per call at ViT-L, same box
    hipLaunchKernelGGL((ln_fwd_kernel<T, 4>), dim3((rows + 3) / 4), dim3(256), 0, c.st, x, (const float*)g, (const float*)b, y,
                       ldy, rows, c.m->c.hidden_size, c.m->c.layer_norm_eps, in_rpg, in_gstride, in_off);
  else
    hipLaunchKernelGGL((ln_fwd_kernel<T, 8>), dim3((rows + 3) / 4), dim3(256), 0, c.st, x, (const float*)g, (const float*)b, y,
                       ldy, rows, c.m->c.hidden_size, c.m->c.layer_norm_eps, in_rpg, in_gstride, in_off);
}
template <typename T>
static void ln_bwd(const Ctx<T>& c, const T* dy, long lddy, const float* x, const void* g, const float* dx_in, float* dx_out,
                   T* dx_t, int rows) {
  // NV = 8 at every width: the four-piece form (no masked loads, 68 instead of 102 registers) measured 279 vs 275 us at D = 1024 --
  // this kernel sits on the HBM rate (16 B per element at 6.0 TB/s) and more resident waves only add DRAM page conflicts
  hipLaunchKernelGGL((ln_bwd_kernel<T, 8>), dim3((rows + 3) / 4), dim3(256), 0, c.st, dy, lddy, x, (const float*)g, dx_in, 1.0f,
                     dx_out, dx_t, rows, c.m->c.hidden_size, c.m->c.layer_norm_eps);
}

template <typename T> static int forward_impl(bsg_model* m, hipStream_t st, int B, const float* pix, const float* prm,
                                              const float* pmask, int emb, float* pred, void* ws, int train, int fe = 0,
                                              int first_row = 0) {
  Ctx<T> c{m, st, (char*)ws, make_plan(m, B, train), B};
  const int N = m->N, D = m->c.hidden_size, L = m->c.num_layers, nh = m->c.num_heads, mlp = m->c.mlp_dim;
  const int hp = m->hp, wp = m->wp, nt = m->c.num_taps, dc = m->c.decoder_hidden;
  const float scale = 0.125f;  // head_dim^-0.5, head_dim == 64
  if (train) {
    std::lock_guard<std::mutex> lk(m->fwd_rows_mu);
    m->fwd_rows[ws] = first_row;
  }
  T* patch_a = c.template at<T>("patch_a");
  T* ln_out = c.template at<T>("ln_out");
  T* h_act = c.template at<T>("h_act");
  T* vt = c.template at<T>("vt");
  float* relh_s = c.template at<float>("relh_s");
  T* taps = c.template at<T>("taps");
  T* feat = c.template at<T>("feat");
  // bsg_forward_rows: pred_masks is wanted on canvas rows >= first_row only (the reference loss and decode read the query half,
  // src/model.py:53-57, :251-255).  The decoder then starts at the 16-row tile holding the first pixel row the BACKWARD reads the
  // saved conv output at (backward_impl: hb0 = 16 ph0 - 8 with ph0 = (first_row - 1) / 16), and decoder_embed at the token row
  // whose pixels the 3x3 conv reaches from there (one pixel row above).  The encoder is untouched: every token attends to all.
  const int ph0b = first_row > 0 ? (first_row - 1) / 16 : 0;
  const int ty0f = std::max(0, 16 * ph0b - 8) / CONV_TR;   // first 16-row conv tile of the forward
  const int tr0 = ty0f > 0 ? ty0f - 1 : 0;                 // first token row of the decoder feature map
  const int ntok_f = (hp - tr0) * wp;                      // tokens per image in the window

  {
    const long total = (long)2 * B * N * 96;
    hipLaunchKernelGGL((patchify_kernel<T>), dim3((unsigned)std::min<long>((total + 255) / 256, 65535 * 16)), dim3(256), 0, st,
                       prm, pix, pmask, patch_a, B, hp, wp, m->c.embed_split);
    CHECK_LAUNCH();
  }
  float* pool[3] = {c.template at<float>("x_a"), c.template at<float>("x_b"), c.template at<float>("x_c")};
  float* x_cur = train ? c.template at<float>("x_in", 0) : pool[0];
  {
    GemmArgs g{};
    const int ek = m->c.embed_split ? 3 : 1;
    g.A = patch_a; g.W = m->gw(0); g.M = 2 * B * N; g.N = D; g.K = 768 * ek; g.lda = 768 * ek;
    g.tokens = N; g.batch = B; g.out = x_cur; g.ldo = D; g.aux = m->gw(emb == 0 ? 2 : 3); g.ldaux = D;
    gemm<T, A_PLAIN, EPI_EMBED>(m, g, st);
    CHECK_LAUNCH();
  }
  for (int l = 0; l < L; ++l) {
    const int S = m->streams(l, B), rows = S * N;
    float* x_mid = train ? c.template at<float>("x_mid", l) : (x_cur == pool[1] ? pool[2] : pool[1]);
    T* qkv = c.template at<T>("qkv", train ? l : -1);
    T* attn_o = c.template at<T>("attn_o", train ? l : -1);
    float* lse2 = c.template at<float>("lse2", train ? l : -1);
    T* h_pre = train ? c.template at<T>("h_pre", l) : nullptr;
    ln_fwd<T>(c, x_cur, m->lw(l, 0), m->lw(l, 1), ln_out, D, rows);
    CHECK_LAUNCH();
    {
      GemmArgs g{};
      g.A = ln_out; g.W = m->lw(l, 2); g.M = rows; g.N = 3 * D; g.K = D; g.lda = D;
      g.bias = (const float*)m->lw(l, 4); g.out = qkv; g.ldo = 3 * D;
      if (sizeof(T) == 4 && m->c.gemm_x3) g.pack_hl = 1;  // q / k / v as (f16 hi, f16 lo) pairs for the x3 attention kernels
      gemm<T, A_PLAIN, EPI_BIAS>(m, g, st);
      CHECK_LAUNCH();
    }
    {
      // 16-bit dtypes: the PV operand comes straight from the row-major V tile through transposing LDS reads; f32 (no
      // 32-bit transposing read on gfx950) keeps the row-padded V^T copy.
      constexpr bool tr = sizeof(T) == 2;
      if (!tr) {
        const int hb = nh % 4 == 0 ? 4 : (nh % 2 == 0 ? 2 : 1);
        hipLaunchKernelGGL((head_transpose_kernel<T>), dim3(hp, nh / hb, S), dim3(256), 0, st, qkv + 2 * D, (long)3 * D, vt, N,
                           wp, hp * 32, nh, hb);
        CHECK_LAUNCH();
      }
      AttnArgs a{};
      a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D; a.ld = 3 * D; a.vt = vt; a.rel_cat = m->lw(l, 18); a.relhT = relh_s; a.out = attn_o;
      a.ldo = D; a.lse2 = lse2; a.S = S; a.nh = nh; a.N = N; a.hp = hp; a.wp = wp; a.scale = scale;
      static bool once2 = (allow_lds(attn_fwd_kernel<T, tr>, 160 * 1024), true);
      (void)once2;
      ProfScope ps(m, st, PC_ATTN_FWD, 4.0 * S * nh * (double)N * N * 64);
      const dim3 agrid(((N + 127) / 128) * nh * S);
      const int relh_lds = 4 * 32 * (hp | 1) * 4;  // prologue scratch (32 x HS relh table per wave) aliases the tiles
      const int lds = std::max(4 * AttnK<T>::TILE, relh_lds);
      bool x3_done = false;
      if constexpr (!tr) {
        if (m->c.gemm_x3) {
          static bool once3 = (allow_lds(attn_fwd_kernel<T, false, true>, 160 * 1024), true);
          (void)once3;
          hipLaunchKernelGGL((attn_fwd_kernel<T, false, true>), agrid, dim3(256), lds, st, a);
          x3_done = true;
        }
      }
      if (!x3_done) hipLaunchKernelGGL((attn_fwd_kernel<T, tr>), agrid, dim3(256), lds, st, a);
      CHECK_LAUNCH();
    }
    {
      GemmArgs g{};
      g.A = attn_o; g.W = m->lw(l, 5); g.M = rows; g.N = D; g.K = D; g.lda = D;
      g.bias = (const float*)m->lw(l, 7); g.out = x_mid; g.ldo = D; g.aux = x_cur; g.ldaux = D;
      gemm<T, A_PLAIN, EPI_BIAS_RESID>(m, g, st);
      CHECK_LAUNCH();
    }
    if (fe) {  // HF:414-423
      const int cond = m->c.merge_index > l ? 2 : 1;
      if (S / 2 >= cond) {
        const long n4 = (long)cond * (N / 2) * D / 4;
        hipLaunchKernelGGL(ensemble_fixup_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 65535)), dim3(256), 0, st,
                           (const float*)x_cur, x_mid, S, cond, N, D);
        CHECK_LAUNCH();
      }
    }
    ln_fwd<T>(c, x_mid, m->lw(l, 8), m->lw(l, 9), ln_out, D, rows);
    CHECK_LAUNCH();
    {
      GemmArgs g{};
      g.A = ln_out; g.W = m->lw(l, 10); g.M = rows; g.N = mlp; g.K = D; g.lda = D;
      g.bias = (const float*)m->lw(l, 12); g.out = h_act; g.out2 = h_pre; g.ldo = mlp;
      if (train) gemm<T, A_PLAIN, EPI_BIAS_GELU>(m, g, st);      // + the derivative, saved for the dgrad
      else gemm<T, A_PLAIN, EPI_BIAS_GELU_FWD>(m, g, st);         // inference: gelu alone (same bits of h_act)
      CHECK_LAUNCH();
    }
    float* x_out;
    if (train) x_out = l == m->c.merge_index ? c.template at<float>("x_tmp") : c.template at<float>("x_in", l + 1);
    else {
      x_out = pool[0];
      for (int i = 0; i < 3; ++i) if (pool[i] != x_cur && pool[i] != x_mid) x_out = pool[i];
    }
    {
      GemmArgs g{};
      g.A = h_act; g.W = m->lw(l, 13); g.M = rows; g.N = D; g.K = mlp; g.lda = mlp;
      g.bias = (const float*)m->lw(l, 15); g.out = x_out; g.ldo = D; g.aux = x_mid; g.ldaux = D;
      gemm<T, A_PLAIN, EPI_BIAS_RESID>(m, g, st);
      CHECK_LAUNCH();
    }
    if (l == m->c.merge_index) {
      float* merged = train ? c.template at<float>("x_in", l + 1) : x_cur;  // x_cur (layer input) is dead by now
      const long n4 = (long)B * N * D / 4;
      hipLaunchKernelGGL(merge_halves_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 65535)), dim3(256), 0, st, x_out,
                         merged, n4);
      CHECK_LAUNCH();
      x_cur = merged;
    } else {
      x_cur = x_out;
    }
    int ti;
    if (m->is_tap(l, &ti)) {
      if (l < m->c.merge_index) return fail("tap %d before merge_index %d", l, m->c.merge_index);
      // rows of the window, gathered: taps is [B * ntok_f][nt * D] (the whole (B N) x (nt D) array when first_row = 0)
      ln_fwd<T>(c, x_cur, m->gw(4), m->gw(5), taps + (long)ti * D, (long)nt * D, B * ntok_f, tr0 ? ntok_f : 0, N, (long)tr0 * wp);
      CHECK_LAUNCH();
    }
  }
  {
    GemmArgs g{};
    g.A = taps; g.W = m->gw(6); g.M = B * ntok_f; g.N = 256 * dc; g.K = nt * D; g.lda = (long)nt * D;
    g.bias = (const float*)m->gw(8); g.out = feat; g.tokens = N; g.wp = wp; g.himg = m->c.canvas_h; g.wimg = m->c.canvas_w;
    g.feat_lg = dc == 128 ? 7 : 6;
    if (tr0) { g.o_rpg = ntok_f; g.t_off = tr0 * wp; }
    gemm<T, A_PLAIN, EPI_FEAT>(m, g, st);
    CHECK_LAUNCH();
  }
  {
    ConvArgs a{};
    a.in = feat; a.w = m->gw(9); a.bias = (const float*)m->gw(11); a.out = train ? c.template at<T>("conv_out") : nullptr;
    a.ln_g = (const float*)m->gw(12); a.ln_b = (const float*)m->gw(13); a.head_w = (const float*)m->gw(14);
    a.head_b = (const float*)m->gw(15); a.pred = pred; a.H = m->c.canvas_h; a.W = m->c.canvas_w; a.eps = m->c.layer_norm_eps;
    ProfScope ps(m, st, PC_CONV, 2.0 * B * (a.H - ty0f * CONV_TR) * a.W * 9 * dc * dc);
    launch_conv<T, CONV_FWD_FUSED>(a, dc, B, ty0f, st, m->c.gemm_x3 != 0);
    CHECK_LAUNCH();
  }
  if (train && (std::is_same<T, f16_t>::value || (std::is_same<T, float>::value && m->c.gemm_x3))) {
    // overflow guard of the scaled dgrad chain (rowops.hpp): a forward that already produced non-finite predictions must not
    // count as an overflow of the backward that follows
    const long plane4 = (long)m->c.canvas_h * m->c.canvas_w / 4, keep4 = (long)(m->c.canvas_h - ty0f * CONV_TR) * m->c.canvas_w / 4;
    const long n4 = (long)B * 3 * keep4;
    hipLaunchKernelGGL(nonfinite_flag_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 2048)), dim3(256), 0, st, pred, n4,
                       (int*)(c.template at<float>("gscale") + 16) + 6, ty0f ? plane4 : 0L, keep4);
    CHECK_LAUNCH();
  }
  return 0;
}

template <typename T> static int backward_impl(bsg_model* m, hipStream_t st, int B, const float* dpred, float* gprompt, void* ws,
                                               int first_row = 0) {
  Ctx<T> c{m, st, (char*)ws, make_plan(m, B, 1), B};
  const int N = m->N, D = m->c.hidden_size, L = m->c.num_layers, nh = m->c.num_heads, mlp = m->c.mlp_dim;
  const int hp = m->hp, wp = m->wp, nt = m->c.num_taps, H = m->c.canvas_h, W = m->c.canvas_w, dc = m->c.decoder_hidden;
  {
    std::lock_guard<std::mutex> lk(m->fwd_rows_mu);
    auto it = m->fwd_rows.find(ws);
    if (it != m->fwd_rows.end() && first_row < it->second)
      return fail("backward over canvas rows >= %d on a workspace whose forward (bsg_forward_rows) computed rows >= %d only", first_row, it->second);
  }
  const int rows = B * N;
  const float scale = 0.125f;
  constexpr bool kF32 = sizeof(T) == 4;
  T* dconv = c.template at<T>("feat");
  T* dfeat = c.template at<T>("feat2");
  T* dtaps = c.template at<T>("dtaps");
  float* dx = c.template at<float>("dx");
  T* dx_t = kF32 ? (T*)dx : c.template at<T>("dx_t");
  T* dx_t_out = kF32 ? nullptr : dx_t;  // f32 mode: the fp32 stream itself feeds the dgrad GEMMs
  T* dh = c.template at<T>("dh");
  T* dn_a = c.template at<T>("dn_a");
  T* dn_b = c.template at<T>("dn_b");
  T* dqkv = c.template at<T>("dqkv");
  T* kt = c.template at<T>("kt");
  T* qt = c.template at<T>("qt");
  T* dot = c.template at<T>("dot");
  float* delta = c.template at<float>("delta");
  float* relhT = c.template at<float>("relhT");
  float* relwT = c.template at<float>("relwT");
  // f16, and f32 with x3 GEMMs (f16 hi / lo operand halves): the dgrad chain runs on S * gradient
  float* gscale = (std::is_same<T, f16_t>::value || (std::is_same<T, float>::value && m->c.gemm_x3)) ? c.template at<float>("gscale") : nullptr;

  {
    // grad_pred is zero on canvas rows < first_row (the reference loss only covers the bottom half, src/model.py:53-57):
    // the 3x3 dgrad reaches one row above, so only token rows >= ph0 can carry a gradient into the encoder.
    const int ph0 = first_row > 0 ? (first_row - 1) / 16 : 0;
    const int hb0 = std::max(0, 16 * ph0 - 8), ty0 = ph0, ntok = (hp - ph0) * wp;  // ty0: first 16-row conv tile = first token row
    const long total = (long)B * (H - hb0) * W;
    if (gscale) {  // f16: S = 2^k from max |grad_pred| (device side), so that the half-precision dgrad chain stays in range
      unsigned* amax = (unsigned*)(gscale + 8);
      if (hipMemsetAsync(amax, 0, 4, st) != hipSuccess) return fail("memset failed");
      const long n4 = (long)B * 3 * H * W / 4;
      hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 2048)), dim3(256), 0, st, dpred, n4, amax);
      hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1), 0, st, (const unsigned*)amax, gscale, 8, (int*)(gscale + 16));
      CHECK_LAUNCH();
    }
    if (dc == 128)
      hipLaunchKernelGGL((head_bwd_kernel<T, 128>), dim3((unsigned)((total * 8 + 255) / 256)), dim3(256), 0, st, dpred,
                         c.template at<T>("conv_out"), (const float*)m->gw(12), (const float*)m->gw(13),
                         (const float*)m->gw(14), dconv, B, H, W, m->c.layer_norm_eps, hb0, (const float*)gscale);
    else
      hipLaunchKernelGGL((head_bwd_kernel<T, 64>), dim3((unsigned)((total * 4 + 255) / 256)), dim3(256), 0, st, dpred,
                         c.template at<T>("conv_out"), (const float*)m->gw(12), (const float*)m->gw(13),
                         (const float*)m->gw(14), dconv, B, H, W, m->c.layer_norm_eps, hb0, (const float*)gscale);
    CHECK_LAUNCH();
    ConvArgs a{};
    a.in = dconv; a.w = m->gw(10); a.out = dfeat; a.H = H; a.W = W; a.eps = m->c.layer_norm_eps;
    {
      ProfScope ps(m, st, PC_CONV, 2.0 * B * (H - ty0 * CONV_TR) * W * 9 * dc * dc);
      launch_conv<T, CONV_PLAIN>(a, dc, B, ty0, st, m->c.gemm_x3 != 0);
    }
    CHECK_LAUNCH();
    if (ph0 > 0) {  // token rows < ph0 receive exactly zero
      if (hipMemset2DAsync(dtaps, (size_t)N * nt * D * sizeof(T), 0, (size_t)ph0 * wp * nt * D * sizeof(T), B, st) != hipSuccess)
        return fail("memset2d failed");
    }
    GemmArgs g{};
    g.A = dfeat; g.W = m->gw(7); g.M = B * ntok; g.N = nt * D; g.K = 256 * dc; g.tokens = N; g.wp = wp; g.himg = H; g.wimg = W;
    g.feat_lg = dc == 128 ? 7 : 6;
    g.a_rpg = ntok; g.t_off = ph0 * wp; g.o_rpg = ntok; g.o_gstride = N; g.o_off = ph0 * wp;
    g.out = dtaps; g.ldo = (long)nt * D;
    gemm<T, A_FEAT, EPI_PLAIN>(m, g, st);
    CHECK_LAUNCH();
  }
  bool dx_valid = false;
  // Row windows of the attention backward (exact: only zeros and unread rows are left out).  (1) In the block of the TOP tap the
  // incoming gradient is zero on token rows < ph0 (dtaps above; the MLP branch is row-wise), so dO is: those queries add nothing to
  // dK / dV and their dq is zero.  (2) Block 0 feeds the patch-embed dgrad of the prompt half only (EPI_UNPATCH below reads the top
  // N / 2 tokens): dq / dk / dv of the query half are never read.  dQ still publishes the softmax tables of every query.
  const int zero_tokens = (first_row > 0 ? (first_row - 1) / 16 : 0) * wp;
  for (int l = L - 1; l >= 0; --l) {
    int ti;
    const bool top_block = !dx_valid;
    if (m->is_tap(l, &ti)) {  // d/dx of the shared tap LayerNorm applied to x_in[l+1]  (HF:475-476)
      ln_bwd<T>(c, dtaps + (long)ti * D, (long)nt * D, c.template at<float>("x_in", l + 1), m->gw(4), dx_valid ? dx : nullptr, dx,
                dx_t_out, rows);
      CHECK_LAUNCH();
      dx_valid = true;
    }
    if (!dx_valid) continue;
    if (l == m->c.merge_index) {  // x = (img + mask) / 2: the image stream receives half; the mask stream has no leaf
      const long n4 = (long)rows * D / 4;
      const unsigned gb = (unsigned)std::min<long>((n4 + 255) / 256, 65535);
      hipLaunchKernelGGL((cast_rows_kernel<float>), dim3(gb), dim3(256), 0, st, dx, dx, n4, 0.5f);
      if (!kF32) hipLaunchKernelGGL((cast_rows_kernel<T>), dim3(gb), dim3(256), 0, st, dx, dx_t, n4, 1.0f);
      CHECK_LAUNCH();
    }
    const T* qkv = c.template at<T>("qkv", l);
    const T* attn_o = c.template at<T>("attn_o", l);
    {
      GemmArgs g{};
      g.A = dx_t; g.W = m->lw(l, 14); g.M = rows; g.N = mlp; g.K = D; g.lda = D; g.out = dh; g.ldo = mlp;
      g.aux = c.template at<T>("h_pre", l); g.ldaux = mlp;
      gemm<T, A_PLAIN, EPI_GELU_BWD>(m, g, st);
      CHECK_LAUNCH();
      GemmArgs g2{};
      g2.A = dh; g2.W = m->lw(l, 11); g2.M = rows; g2.N = D; g2.K = mlp; g2.lda = mlp; g2.out = dn_a; g2.ldo = D;
      gemm<T, A_PLAIN, EPI_PLAIN>(m, g2, st);
      CHECK_LAUNCH();
    }
    ln_bwd<T>(c, dn_a, D, c.template at<float>("x_mid", l), m->lw(l, 8), dx, dx, dx_t_out, rows);
    CHECK_LAUNCH();
    {
      GemmArgs g{};
      g.A = dx_t; g.W = m->lw(l, 6); g.M = rows; g.N = D; g.K = D; g.lda = D; g.out = dn_b; g.ldo = D;
      gemm<T, A_PLAIN, EPI_PLAIN>(m, g, st);
      CHECK_LAUNCH();
    }
    {  // attention backward on the B image streams
      constexpr bool tr = sizeof(T) == 2;  // 16-bit: K^T / Q^T / dO^T operands via transposing LDS reads of the row-major tiles
      if (!tr) {
        const int hb = nh % 4 == 0 ? 4 : (nh % 2 == 0 ? 2 : 1);
        const int npad = hp * 32, dg = 2 * ((N + 63) / 64);  // key side: one group per grid row; query side: dense tokens
        hipLaunchKernelGGL((head_transpose_kernel<T>), dim3(hp, nh / hb, B), dim3(256), 0, st, qkv + D, (long)3 * D, kt, N, wp, npad, nh, hb);
        hipLaunchKernelGGL((head_transpose_kernel<T>), dim3(dg, nh / hb, B), dim3(256), 0, st, qkv, (long)3 * D, qt, N, 32, npad, nh, hb);
        hipLaunchKernelGGL((head_transpose_kernel<T>), dim3(dg, nh / hb, B), dim3(256), 0, st, (const T*)dn_b, (long)D, dot, N, 32,
                           npad, nh, hb);
        CHECK_LAUNCH();
      }
      AttnArgs a{};
      a.q = qkv; a.k = qkv + D; a.v = qkv + 2 * D; a.ld = 3 * D; a.kt = kt; a.dout = dn_b; a.ldo = D; a.rel_cat = m->lw(l, 18);
      a.rel_catT = m->lw(l, 19); a.relhT = relhT; a.relwT = relwT; a.lse2 = c.template at<float>("lse2", l); a.delta = delta; a.out = (void*)attn_o; a.dq = dqkv;
      a.S = B; a.nh = nh; a.N = N; a.hp = hp; a.wp = wp; a.scale = scale;
      a.dq_begin = top_block ? zero_tokens / 128 * 128 : 0;
      a.dq_end = l == 0 ? std::min(N, (N / 2 + 127) / 128 * 128) : 0;
      const double dq_frac = (double)((a.dq_end ? a.dq_end : N) - a.dq_begin) / N;
      static bool once = (allow_lds(attn_bwd_dq_kernel<T, tr>, 160 * 1024), true);
      (void)once;
      {
        // ALGORITHMIC work of the attention backward (SURVEY.md section 8 d): 2 x the forward's matmuls = 8 N^2 d per head, credited
        // half to each of the two kernels that share it.  (They EXECUTE 6 + 8 = 14 N^2 d: both recompute S and dP; a one-pass
        // flash backward executes 10.  Rounds 1-3 credited the executed count, which flattered both kernels by 1.75x.)
        ProfScope ps(m, st, PC_ATTN_BWD_DQ, 4.0 * B * nh * (double)N * N * 64 * dq_frac);
        const dim3 qgrid(((N + 127) / 128) * nh * B);
        const int relh_lds = 4 * 32 * (hp | 1) * 4;
        bool x3_done = false;
        if constexpr (!tr) {
          if (m->c.gemm_x3) {
            static bool once3 = (allow_lds(attn_bwd_dq_kernel<T, false, true>, 160 * 1024), true);
            (void)once3;
            hipLaunchKernelGGL((attn_bwd_dq_kernel<T, false, true>), qgrid, dim3(256), 6 * AttnK<T>::TILE + relh_lds, st, a);
            x3_done = true;
          }
        }
        if (!x3_done) hipLaunchKernelGGL((attn_bwd_dq_kernel<T, tr>), qgrid, dim3(256), (tr ? 4 : 6) * AttnK<T>::TILE + relh_lds, st, a);
      }
      CHECK_LAUNCH();
      AttnBwdKvArgs k{};
      k.k = qkv + D; k.v = qkv + 2 * D; k.q = qkv; k.dout = dn_b; k.qt = qt; k.dot = dot; k.ld = 3 * D; k.ldo = D;
      k.relwT = relwT; k.relhT = relhT; k.lse2 = a.lse2; k.delta = delta; k.dk = dqkv + D; k.dv = dqkv + 2 * D;
      k.S = B; k.nh = nh; k.N = N; k.hp = hp; k.wp = wp; k.scale = scale;
      k.q_begin = top_block ? zero_tokens / 64 * 64 : 0;
      k.key_rows = l == 0 ? (hp + 1) / 2 : 0;
      {
        ProfScope ps(m, st, PC_ATTN_BWD_DKV, 4.0 * B * nh * (double)N * N * 64 * ((double)(N - k.q_begin) / N) * (k.key_rows ? (double)k.key_rows / hp : 1.0));
        launch_dkv<T>(k, st, m->c.gemm_x3 != 0);
      }
      CHECK_LAUNCH();
    }
    {
      GemmArgs g{};
      g.A = dqkv; g.W = m->lw(l, 3); g.M = rows; g.N = D; g.K = 3 * D; g.lda = 3 * D; g.out = dn_a; g.ldo = D;
      gemm<T, A_PLAIN, EPI_PLAIN>(m, g, st);
      CHECK_LAUNCH();
    }
    ln_bwd<T>(c, dn_a, D, c.template at<float>("x_in", l), m->lw(l, 0), dx, dx, dx_t_out, rows);
    CHECK_LAUNCH();
  }
  if (!dx_valid) return fail("no tap reaches the loss: gradient is identically zero");
  {  // patch-embed dgrad over the prompt (top) half of the image canvas, un-patchified straight into the output
    GemmArgs g{};
    g.A = dx_t; g.W = m->gw(1); g.M = B * (N / 2); g.N = 768; g.K = D; g.lda = D; g.a_rpg = N / 2; g.a_gstride = N;
    if (m->c.embed_split) {  // the last dgrad at split precision: fp32 dx -> [hi | hi | lo] against [W_hi | W_lo | W_hi]^T
      T* dxs = c.template at<T>("dx_split");
      const long rows_s = (long)B * (N / 2), n4 = rows_s * D / 4;
      hipLaunchKernelGGL((split_rows_kernel<T>), dim3((unsigned)std::min<long>((n4 + 255) / 256, 65535)), dim3(256), 0, st, dx, dxs,
                         rows_s, D, N / 2, (long)N);
      CHECK_LAUNCH();
      g.A = dxs; g.K = 3 * D; g.lda = 3 * D; g.a_rpg = 0; g.a_gstride = 0;
    }
    g.tokens = N; g.wp = wp; g.himg = H; g.wimg = W; g.out = gprompt;
    g.out_scale = gscale ? gscale + 1 : nullptr;  // f16: 1 / S
    gemm<T, A_PLAIN, EPI_UNPATCH>(m, g, st);
    CHECK_LAUNCH();
    if (gscale) {  // f16: non-finite prompt gradient -> overflow flag + more headroom next time (rowops.hpp)
      const long n4 = (long)B * 3 * (H / 2) * W / 4;
      hipLaunchKernelGGL(grad_finite_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 2048)), dim3(256), 0, st,
                         (const float*)gprompt, n4, (int*)(gscale + 16));
      hipLaunchKernelGGL(grad_state_kernel, dim3(1), dim3(1), 0, st, (int*)(gscale + 16));
      CHECK_LAUNCH();
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

const char* bsg_last_error(void) { return g_err.c_str(); }
const char* bsg_build_info(void) { return "beach_seg_amd hip kernels, gfx950, built " __DATE__ " " __TIME__; }

int bsg_create(const bsg_config* cfg, const void* const* weights, int n_weights, bsg_model** out) {
  if (!cfg || !weights || !out) return fail("bsg_create: null argument");
  const bsg_config& c = *cfg;
  if (c.dtype != BSG_DTYPE_F32 && c.dtype != BSG_DTYPE_BF16 && c.dtype != BSG_DTYPE_F16) return fail("dtype must be 0 (f32), 1 (bf16) or 2 (f16)");
  if (c.patch_size != 16) return fail("patch_size must be 16");
  if (c.decoder_hidden != 64 && c.decoder_hidden != 128) return fail("decoder_hidden must be 64 or 128");
  if (c.num_heads <= 0 || c.hidden_size != c.num_heads * 64) return fail("head_dim must be 64 (hidden %d, heads %d)", c.hidden_size, c.num_heads);
  if (c.hidden_size % 64 || c.mlp_dim % 64 || c.hidden_size > 2048) return fail("hidden_size / mlp_dim must be multiples of 64, hidden <= 2048");
  if (c.canvas_h % 32 || c.canvas_w % 32) return fail("canvas must be a multiple of 32 pixels in both axes");
  const int hp = c.canvas_h / 16, wp = c.canvas_w / 16;
  if (wp > 32 || hp > 64) return fail("token grid %d x %d exceeds 64 x 32", hp, wp);
  if (c.num_taps < 1 || c.num_taps > BSG_MAX_TAPS) return fail("num_taps out of range");
  for (int i = 0; i < c.num_taps; ++i)
    if (c.taps[i] < c.merge_index || c.taps[i] >= c.num_layers) return fail("tap index %d out of range", c.taps[i]);
  if (c.merge_index < 0 || c.merge_index >= c.num_layers) return fail("merge_index out of range");
  if (c.embed_split != 0 && c.embed_split != 1) return fail("embed_split must be 0 or 1");
  if (c.embed_split && c.dtype == BSG_DTYPE_F32) return fail("embed_split applies to the 16-bit dtypes only");
  if (c.gemm_x3 != 0 && c.gemm_x3 != 1) return fail("gemm_x3 must be 0 or 1");
  if (c.gemm_x3 && c.dtype != BSG_DTYPE_F32) return fail("gemm_x3 applies to BSG_DTYPE_F32 only");
  const int need = BSG_GLOBAL_WEIGHTS + BSG_LAYER_WEIGHTS * c.num_layers;
  if (n_weights != need) return fail("weight table has %d entries, expected %d", n_weights, need);
  for (int i = 0; i < need; ++i) if (!weights[i]) return fail("weight table entry %d is null", i);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device: the MI355X kernels cannot run here");
  bsg_model* m = new bsg_model();
  m->c = c;
  m->w.assign(weights, weights + need);
  m->hp = hp; m->wp = wp; m->N = hp * wp; m->npad = hp * 32; m->relcat_rows = ((2 * hp - 1 + 2 * wp - 1) + 3) & ~3; m->es = c.dtype == BSG_DTYPE_F32 ? 4 : 2;
  *out = m;
  return 0;
}

void bsg_destroy(bsg_model* m) {
  if (!m) return;
  for (auto& r : m->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto e : m->pool) (void)hipEventDestroy(e);
  delete m;
}

int bsg_profile_enable(bsg_model* m, int enable) {
  if (!m) return fail("null model");
  m->prof = enable != 0;
  return 0;
}

int bsg_profile_read(bsg_model* m, int category, double* total_ms, double* total_flops, long* launches) {
  if (!m || category < 0 || category >= PC_COUNT) return fail("bsg_profile_read: bad argument");
  double ms = 0, fl = 0; long n = 0;
  for (auto& r : m->recs) {
    if (r.cat != category) continue;
    if (hipEventSynchronize(r.e1) != hipSuccess) return fail("event sync failed");
    float t = 0;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return fail("event elapsed failed");
    ms += t; fl += r.flops; ++n;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  if (launches) *launches = n;
  return 0;
}

int bsg_profile_reset(bsg_model* m) {
  if (!m) return fail("null model");
  for (auto& r : m->recs) { m->pool.push_back(r.e0); m->pool.push_back(r.e1); }
  m->recs.clear();
  return 0;
}

size_t bsg_workspace_bytes(const bsg_model* m, int batch, int train) {
  if (!m || batch <= 0) return 0;
  return make_plan(m, batch, train).total;
}

int bsg_workspace_region(const bsg_model* m, int batch, int train, const char* name, int layer, size_t* offset, size_t* bytes) {
  if (!m || !name) return fail("null argument");
  Plan p = make_plan(m, batch, train);
  const Region* r = p.find(name, layer);
  if (!r) return fail("no workspace region '%s' layer %d", name, layer);
  if (offset) *offset = r->off;
  if (bytes) *bytes = r->bytes;
  return 0;
}

int bsg_forward(bsg_model* m, void* stream, int batch, const float* pixel_values, const float* prompt_pixel_values,
                const float* prompt_masks, int embedding_type, float* pred_masks, void* workspace, size_t workspace_bytes,
                int save_for_backward) {
  return bsg_forward_rows(m, stream, batch, pixel_values, prompt_pixel_values, prompt_masks, embedding_type, 0, pred_masks, workspace,
                          workspace_bytes, save_for_backward);
}

int bsg_forward_rows(bsg_model* m, void* stream, int batch, const float* pixel_values, const float* prompt_pixel_values,
                     const float* prompt_masks, int embedding_type, int first_row, float* pred_masks, void* workspace,
                     size_t workspace_bytes, int save_for_backward) {
  if (!m || !pixel_values || !prompt_pixel_values || !prompt_masks || !pred_masks || !workspace) return fail("bsg_forward: null argument");
  if (first_row < 0 || first_row >= m->c.canvas_h) return fail("first_row %d outside the canvas", first_row);
  if (batch <= 0) return fail("batch must be positive");
  if (embedding_type != 0 && embedding_type != 1)
    return fail("Embedding type should be either 'semantic' or 'instance', but got %d", embedding_type);
  if (workspace_bytes < bsg_workspace_bytes(m, batch, save_for_backward))
    return fail("workspace too small: %zu < %zu", workspace_bytes, bsg_workspace_bytes(m, batch, save_for_backward));
  hipStream_t st = (hipStream_t)stream;
#define BSG_FWD(TT) forward_impl<TT>(m, st, batch, pixel_values, prompt_pixel_values, prompt_masks, embedding_type, pred_masks, workspace, save_for_backward, 0, first_row)
  return m->c.dtype == BSG_DTYPE_F32 ? BSG_FWD(float) : m->c.dtype == BSG_DTYPE_BF16 ? BSG_FWD(bf16_t) : BSG_FWD(f16_t);
#undef BSG_FWD
}

int bsg_forward_ensemble(bsg_model* m, void* stream, int batch, const float* pixel_values, const float* prompt_pixel_values,
                         const float* prompt_masks, int embedding_type, float* pred_masks, void* workspace,
                         size_t workspace_bytes) {
  if (!m || !pixel_values || !prompt_pixel_values || !prompt_masks || !pred_masks || !workspace) return fail("bsg_forward_ensemble: null argument");
  if (batch <= 0) return fail("batch must be positive");
  if (embedding_type != 0 && embedding_type != 1)
    return fail("Embedding type should be either 'semantic' or 'instance', but got %d", embedding_type);
  if (workspace_bytes < bsg_workspace_bytes(m, batch, 0)) return fail("workspace too small");
  hipStream_t st = (hipStream_t)stream;
#define BSG_FWD(TT) forward_impl<TT>(m, st, batch, pixel_values, prompt_pixel_values, prompt_masks, embedding_type, pred_masks, workspace, 0, 1)
  return m->c.dtype == BSG_DTYPE_F32 ? BSG_FWD(float) : m->c.dtype == BSG_DTYPE_BF16 ? BSG_FWD(bf16_t) : BSG_FWD(f16_t);
#undef BSG_FWD
}

int bsg_backward_rows(bsg_model* m, void* stream, int batch, const float* grad_pred, int first_row,
                      float* grad_prompt_pixel_values, void* workspace, size_t workspace_bytes) {
  if (!m || !grad_pred || !grad_prompt_pixel_values || !workspace) return fail("bsg_backward: null argument");
  if (workspace_bytes < bsg_workspace_bytes(m, batch, 1)) return fail("workspace too small for backward");
  if (first_row < 0 || first_row >= m->c.canvas_h) return fail("first_row %d outside the canvas", first_row);
  hipStream_t st = (hipStream_t)stream;
#define BSG_BWD(TT) backward_impl<TT>(m, st, batch, grad_pred, grad_prompt_pixel_values, workspace, first_row)
  return m->c.dtype == BSG_DTYPE_F32 ? BSG_BWD(float) : m->c.dtype == BSG_DTYPE_BF16 ? BSG_BWD(bf16_t) : BSG_BWD(f16_t);
#undef BSG_BWD
}

int bsg_backward(bsg_model* m, void* stream, int batch, const float* grad_pred, float* grad_prompt_pixel_values, void* workspace,
                 size_t workspace_bytes) {
  return bsg_backward_rows(m, stream, batch, grad_pred, 0, grad_prompt_pixel_values, workspace, workspace_bytes);
}

static const int kLossBlocks = 1024;
size_t bsg_loss_scratch_bytes(int h, int w) { return (size_t)h * w * 4 + kLossBlocks * 4 + 256; }

int bsg_loss_fwd_bwd(void* stream, int batch, int h, int w, const float* pred, const float* labels, const uint8_t* yesdata,
                     float beta, int variant, float* loss_out, float* grad_pred, void* scratch, size_t scratch_bytes) {
  if (!pred || !labels || !yesdata || !loss_out || !scratch) return fail("bsg_loss_fwd_bwd: null argument");
  if (variant != 0 && variant != 1) return fail("loss variant must be 0 (reference) or 1 (per_sample)");
  if (scratch_bytes < bsg_loss_scratch_bytes(h, w)) return fail("loss scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const long hw = (long)h * w;
  unsigned long long* total = (unsigned long long*)scratch;
  float* partial = (float*)((char*)scratch + 256);
  float* counts = partial + kLossBlocks;
  if (hipMemsetAsync(total, 0, 8, st) != hipSuccess) return fail("memset failed");
  hipLaunchKernelGGL(loss_prep_kernel, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, st, yesdata, counts, total, batch, hw);
  hipLaunchKernelGGL(loss_fwd_bwd_kernel, dim3(kLossBlocks), dim3(256), 0, st, pred, labels, yesdata, (const float*)counts,
                     (const unsigned long long*)total, grad_pred, partial, batch, h, w, beta, variant);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, (const float*)partial, kLossBlocks,
                     (const unsigned long long*)total, loss_out);
  CHECK_LAUNCH();
  return 0;
}

int bsg_loss_fwd_bwd_ids(void* stream, int batch, int h, int w, int K, const float* pred, const uint8_t* class_ids,
                         const float* palette_norm, float beta, int variant, float* loss_out, float* grad_pred, void* scratch,
                         size_t scratch_bytes) {
  if (!pred || !class_ids || !palette_norm || !loss_out || !scratch) return fail("bsg_loss_fwd_bwd_ids: null argument");
  if (variant != 0 && variant != 1) return fail("loss variant must be 0 (reference) or 1 (per_sample)");
  if (K <= 0 || K > 256) return fail("bsg_loss_fwd_bwd_ids: K must be in 1..256");
  if (scratch_bytes < bsg_loss_scratch_bytes(h, w)) return fail("loss scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const long hw = (long)h * w;
  unsigned long long* total = (unsigned long long*)scratch;
  float* partial = (float*)((char*)scratch + 256);
  float* counts = partial + kLossBlocks;
  if (hipMemsetAsync(total, 0, 8, st) != hipSuccess) return fail("memset failed");
  // yesdata = (class id != 0) (src/model.py:255 `mask != 0`): the id plane itself serves as the u8 truth plane
  hipLaunchKernelGGL(loss_prep_kernel, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, st, class_ids, counts, total, batch, hw);
  hipLaunchKernelGGL(loss_fwd_bwd_kernel, dim3(kLossBlocks), dim3(256), 0, st, pred, (const float*)nullptr, class_ids,
                     (const float*)counts, (const unsigned long long*)total, grad_pred, partial, batch, h, w, beta, variant,
                     class_ids, palette_norm, K);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, (const float*)partial, kLossBlocks,
                     (const unsigned long long*)total, loss_out);
  CHECK_LAUNCH();
  return 0;
}

int bsg_mask_rgb_norm(void* stream, int batch, int h, int w, int K, const uint8_t* class_ids, const uint8_t* palette,
                      const float mean[3], const float std[3], float* out) {
  if (!class_ids || !palette || !mean || !std || !out) return fail("bsg_mask_rgb_norm: null argument");
  if (K <= 0 || K > 256) return fail("bsg_mask_rgb_norm: K must be in 1..256");
  if (batch <= 0 || h <= 0 || w <= 0) return fail("bsg_mask_rgb_norm: bad geometry");
  const long hw = (long)h * w;
  hipLaunchKernelGGL(mask_rgb_norm_kernel, dim3((unsigned)std::min<long>((hw / 4 + 255) / 256 + 1, 1024), batch), dim3(256), 0,
                     (hipStream_t)stream, class_ids, palette, out, hw, K, mean[0], mean[1], mean[2], std[0], std[1], std[2]);
  CHECK_LAUNCH();
  return 0;
}

int bsg_decode_argmin(void* stream, int batch, int h, int w, int K, const float* pred, const float* palette_norm,
                      int64_t* out_i64, uint8_t* out_u8) {
  if (!pred || !palette_norm || (!out_i64 && !out_u8)) return fail("bsg_decode_argmin: null argument");
  const long n = (long)batch * h * w;
  hipLaunchKernelGGL(decode_argmin_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred,
                     palette_norm, (long long*)out_i64, out_u8, batch, h, w, K);
  CHECK_LAUNCH();
  return 0;
}

int bsg_prompt_gather(void* stream, int batch, int h, int w, const float* params, const int32_t* idx, const float mean[3],
                      const float std[3], float* out) {
  if (!params || !idx || !out) return fail("bsg_prompt_gather: null argument");
  const long hw = (long)h * w, n = (long)batch * 3 * hw;
  hipLaunchKernelGGL(prompt_gather_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 65535)), dim3(256), 0,
                     (hipStream_t)stream, params, out, (const int*)idx, batch, 3 * hw, hw, mean[0], mean[1], mean[2], std[0],
                     std[1], std[2]);
  CHECK_LAUNCH();
  return 0;
}

int bsg_prompt_grad_scatter(void* stream, int batch, int h, int w, const float* grad_pixels, const int32_t* idx,
                            const float std[3], float* grad_params) {
  if (!grad_pixels || !idx || !grad_params) return fail("bsg_prompt_grad_scatter: null argument");
  const long hw = (long)h * w, n = (long)batch * 3 * hw;
  hipLaunchKernelGGL(prompt_grad_scatter_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 65535)), dim3(256), 0,
                     (hipStream_t)stream, grad_pixels, grad_params, (const int*)idx, batch, 3 * hw, hw, 1.f / std[0],
                     1.f / std[1], 1.f / std[2]);
  CHECK_LAUNCH();
  return 0;
}

int bsg_adamw_step(void* stream, int n_active, long row_elems, float* params, const float* grads, float* exp_avg,
                   float* exp_avg_sq, const int32_t* active, const uint8_t* touched, const float* step_sizes,
                   const float* bc2_sqrts, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float grad_scale) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !active || !step_sizes || !bc2_sqrts) return fail("bsg_adamw_step: null argument");
  if (n_active <= 0) return 0;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)std::min<long>((row_elems + 255) / 256, 4096), n_active), dim3(256), 0,
                     (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, (const int*)active, touched, step_sizes, bc2_sqrts,
                     row_elems, lr, beta1, beta2, eps, weight_decay, grad_scale);
  CHECK_LAUNCH();
  return 0;
}

int bsg_op_gemm(void* stream, int dtype, int M, int N, int K, const void* A, const void* W, const float* bias, void* out) {
  if (!A || !W || !out) return fail("bsg_op_gemm: null argument");
  const bool x3 = dtype == 3;  // f32 operands, W in the pre-split x3 format of bsg_config.gemm_x3 (N > 192: the 256 x 256 kernel)
  if (x3) dtype = BSG_DTYPE_F32;
  if (dtype != BSG_DTYPE_F32 && dtype != BSG_DTYPE_BF16 && dtype != BSG_DTYPE_F16) return fail("bsg_op_gemm: dtype must be 0 (f32), 1 (bf16), 2 (f16) or 3 (f32, x3 weights)");
  if (K % (dtype == BSG_DTYPE_F32 ? 32 : 64) || N % 4) return fail("bsg_op_gemm: K must be a multiple of the 128-byte K tile, N of 4");
  if (x3 && N <= 192) return fail("bsg_op_gemm: the x3 form exists for the 256 x 256 kernel only (N > 192)");
  bsg_model dummy{};
  dummy.c.gemm_x3 = x3 ? 1 : 0;
  GemmArgs g{};
  g.A = A; g.W = W; g.M = M; g.N = N; g.K = K; g.lda = K; g.bias = bias; g.out = out; g.ldo = N;
  hipStream_t st = (hipStream_t)stream;
#ifdef BSG_DIAG  // result-corrupting timing switches exist only in diagnostic builds (-DBSG_DIAG)
  static const bool nostore = getenv("BSG_GEMM_NOSTORE") != nullptr;
  if (getenv("BSG_GEMM_LDO0")) g.ldo = 0;  // every row lands on row 0 (stores stay in L2)
  if (nostore) { gemm<bf16_t, A_PLAIN, EPI_NONE>(&dummy, g, st); CHECK_LAUNCH(); return 0; }
#endif
  if (dtype == BSG_DTYPE_F32) { if (bias) gemm<float, A_PLAIN, EPI_BIAS>(&dummy, g, st); else gemm<float, A_PLAIN, EPI_PLAIN>(&dummy, g, st); }
  else if (dtype == BSG_DTYPE_F16) { if (bias) gemm<f16_t, A_PLAIN, EPI_BIAS>(&dummy, g, st); else gemm<f16_t, A_PLAIN, EPI_PLAIN>(&dummy, g, st); }
#ifdef BSG_DIAG_STAMPS  // the diagnostic build times the GELU epilogue through the bias entry
  else { if (bias) gemm<bf16_t, A_PLAIN, EPI_BIAS_GELU>(&dummy, g, st); else gemm<bf16_t, A_PLAIN, EPI_PLAIN>(&dummy, g, st); }
#else
  else { if (bias) gemm<bf16_t, A_PLAIN, EPI_BIAS>(&dummy, g, st); else gemm<bf16_t, A_PLAIN, EPI_PLAIN>(&dummy, g, st); }
#endif
#ifdef BSG_DIAG_STAMPS
  {
    (void)hipStreamSynchronize(st);
    static long long h[256 * 4];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(bsg_stamps), sizeof(h));
    double a = 0, b = 0, c = 0, d = 0;
    for (int i = 0; i < 256; ++i) { a += h[4 * i]; b += h[4 * i + 1]; c += h[4 * i + 2]; d += h[4 * i + 3]; }
    fprintf(stderr, "[stamps] per tile (second tile of each workgroup, shader cycles): main loop %.0f, epilogue to last store issued %.0f, store drain %.0f; shader clock %.2f GHz\n", a / 256, b / 256, c / 256, d / 256 * 1e-4);
  }
#endif
  CHECK_LAUNCH();
  return 0;
}


size_t bsg_op_attention_scratch_bytes(int S, int nh, int hp) {
  const size_t npad = (size_t)hp * 32;
  // fwd key-major relh | relhT | relwT | delta
  return (size_t)S * nh * npad * 4 * ((size_t)hp + hp + 32 + 1) + 1024;
}

}  // extern "C"
template <typename T>
static int op_gemm_epilogue_impl(hipStream_t st, int epilogue, int M, int N, int K, const void* A, const void* W, const float* bias,
                                 const void* aux, void* out, void* out2) {
  bsg_model dummy{};
  GemmArgs g{};
  g.A = A; g.W = W; g.M = M; g.N = N; g.K = K; g.lda = K; g.bias = bias; g.out = out; g.ldo = N; g.aux = aux; g.ldaux = N; g.out2 = out2;
  if (epilogue == 1) gemm<T, A_PLAIN, EPI_BIAS_GELU>(&dummy, g, st);
  else if (epilogue == 2) gemm<T, A_PLAIN, EPI_BIAS_RESID>(&dummy, g, st);
  else gemm<T, A_PLAIN, EPI_GELU_BWD>(&dummy, g, st);
  CHECK_LAUNCH();
  return 0;
}

// bsg_op_attention_windows: the row windows the NEXT bsg_op_attention call of this thread runs its backward kernels with
static thread_local int g_attn_win[4] = {0, 0, 0, 0};  // dq_begin, dq_end, q_begin, key_rows
template <typename T>
static int op_attention_impl(void* stream, int which, int S, int nh, int hp, int wp, const void* qkv, const void* rel_cat,
                             const void* rel_catT, const void* dout, void* out, float* lse2, void* dqkv, void* scratch,
                             size_t scratch_bytes) {
  if (!qkv || !rel_cat || !out || !lse2 || !scratch) return fail("bsg_op_attention: null argument");
  if ((which & 62) && (!rel_catT || !dout || !dqkv)) return fail("bsg_op_attention: backward needs rel_catT, dout, dqkv");
  if (hp % 2 || hp > 64 || wp > 32 || wp % 4) return fail("bsg_op_attention: bad token grid %d x %d", hp, wp);
  if (scratch_bytes < bsg_op_attention_scratch_bytes(S, nh, hp)) return fail("bsg_op_attention: scratch too small");
  hipStream_t st = (hipStream_t)stream;
  const int N = hp * wp, D = nh * 64;
  const size_t npad = (size_t)hp * 32, per = (size_t)S * nh * npad;
  float* relh_s = (float*)scratch;
  float* relhT = relh_s + per * hp;
  float* relwT = relhT + per * hp;
  float* delta = relwT + per * 32;
  const T* q = (const T*)qkv;
  static bool once = (allow_lds(attn_fwd_kernel<T, true>, 160 * 1024), allow_lds(attn_bwd_dq_kernel<T, true>, 160 * 1024), true);
  (void)once;
  const int relh_lds = 4 * 32 * (hp | 1) * 4;
  if (which & 1) {
    AttnArgs a{};
    a.q = q; a.k = q + D; a.v = q + 2 * D; a.ld = 3 * D; a.rel_cat = rel_cat; a.relhT = relh_s; a.out = out; a.ldo = D;
    a.lse2 = lse2; a.S = S; a.nh = nh; a.N = N; a.hp = hp; a.wp = wp; a.scale = 0.125f;
    hipLaunchKernelGGL((attn_fwd_kernel<T, true>), dim3(((N + 127) / 128) * nh * S), dim3(256),
                       std::max(4 * AttnK<T>::TILE, relh_lds), st, a);
    CHECK_LAUNCH();
  }
  if (which & 2) {
    AttnArgs a{};
    a.q = q; a.k = q + D; a.v = q + 2 * D; a.ld = 3 * D; a.dout = dout; a.ldo = D; a.rel_cat = rel_cat; a.rel_catT = rel_catT;
    a.relhT = relhT; a.relwT = relwT; a.lse2 = lse2; a.delta = delta; a.out = out; a.dq = dqkv;
    a.S = S; a.nh = nh; a.N = N; a.hp = hp; a.wp = wp; a.scale = 0.125f;
    a.dq_begin = g_attn_win[0]; a.dq_end = g_attn_win[1];
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, true>), dim3(((N + 127) / 128) * nh * S), dim3(256),
                       4 * AttnK<T>::TILE + relh_lds, st, a);
    CHECK_LAUNCH();
#ifdef BSG_DIAG_STAMPS_ATTN
    {
      (void)hipStreamSynchronize(st);
      unsigned long long hbuf[8], z[8] = {0};
      (void)hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(bsg_attn_stamps), sizeof(hbuf));
      (void)hipMemcpyToSymbol(HIP_SYMBOL(bsg_attn_stamps), z, sizeof(z));
      const double n = (double)std::max<unsigned long long>(hbuf[7], 1);
      fprintf(stderr, "[dq stamps] per workgroup (wave 0, shader cycles): load+tables %.0f, publish %.0f, delta %.0f, key loop %.0f, rel-pos grad %.0f, store %.0f\n",
              hbuf[0] / n, hbuf[1] / n, hbuf[2] / n, hbuf[3] / n, hbuf[4] / n, hbuf[5] / n);
    }
#endif
  }
  if (which & 4) {
    AttnBwdKvArgs k{};
    k.k = q + D; k.v = q + 2 * D; k.q = q; k.dout = dout; k.ld = 3 * D; k.ldo = D; k.relwT = relwT; k.relhT = relhT;
    k.lse2 = lse2; k.delta = delta; k.dk = (T*)dqkv + D; k.dv = (T*)dqkv + 2 * D;
    k.S = S; k.nh = nh; k.N = N; k.hp = hp; k.wp = wp; k.scale = 0.125f;
    k.q_begin = g_attn_win[2]; k.key_rows = g_attn_win[3];
    launch_dkv<T>(k, st);
    CHECK_LAUNCH();
  }
  if (which & 16) {  // A/B: the eight-wave dK / dV kernel (one workgroup per CU)
    AttnBwdKvArgs k{};
    k.k = q + D; k.v = q + 2 * D; k.q = q; k.dout = dout; k.ld = 3 * D; k.ldo = D; k.relwT = relwT; k.relhT = relhT;
    k.lse2 = lse2; k.delta = delta; k.dk = (T*)dqkv + D; k.dv = (T*)dqkv + 2 * D;
    k.S = S; k.nh = nh; k.N = N; k.hp = hp; k.wp = wp; k.scale = 0.125f;
    k.q_begin = g_attn_win[2]; k.key_rows = g_attn_win[3];
    launch_dkv<T>(k, st, false, 1);
    CHECK_LAUNCH();
  }
  if (which & 32) {  // A/B: two four-wave workgroups per CU
    AttnBwdKvArgs k{};
    k.k = q + D; k.v = q + 2 * D; k.q = q; k.dout = dout; k.ld = 3 * D; k.ldo = D; k.relwT = relwT; k.relhT = relhT;
    k.lse2 = lse2; k.delta = delta; k.dk = (T*)dqkv + D; k.dv = (T*)dqkv + 2 * D;
    k.S = S; k.nh = nh; k.N = N; k.hp = hp; k.wp = wp; k.scale = 0.125f;
    k.q_begin = g_attn_win[2]; k.key_rows = g_attn_win[3];
    launch_dkv<T>(k, st, false, 2);
    CHECK_LAUNCH();
  }
  if (which & 8) {  // A/B: the one-wave-per-SIMD dK / dV kernel on the same tables
    AttnBwdKvArgs k{};
    k.k = q + D; k.v = q + 2 * D; k.q = q; k.dout = dout; k.ld = 3 * D; k.ldo = D; k.relwT = relwT; k.relhT = relhT;
    k.lse2 = lse2; k.delta = delta; k.dk = (T*)dqkv + D; k.dv = (T*)dqkv + 2 * D;
    k.S = S; k.nh = nh; k.N = N; k.hp = hp; k.wp = wp; k.scale = 0.125f;
    launch_dkv4<T>(k, st);
    CHECK_LAUNCH();
  }
  g_attn_win[0] = g_attn_win[1] = g_attn_win[2] = g_attn_win[3] = 0;
  return 0;
}

extern "C" {
int bsg_op_attention_windows(int dq_begin, int dq_end, int q_begin, int key_rows) {
  if (dq_begin < 0 || dq_end < 0 || q_begin < 0 || q_begin % 64 || key_rows < 0) return fail("bsg_op_attention_windows: bad window");
  g_attn_win[0] = dq_begin; g_attn_win[1] = dq_end; g_attn_win[2] = q_begin; g_attn_win[3] = key_rows;
  return 0;
}
int bsg_op_gemm_epilogue(void* stream, int dtype, int epilogue, int M, int N, int K, const void* A, const void* W, const float* bias,
                         const void* aux, void* out, void* out2) {
  if (!A || !W || !out) return fail("bsg_op_gemm_epilogue: null argument");
  if (dtype != BSG_DTYPE_BF16 && dtype != BSG_DTYPE_F16) return fail("bsg_op_gemm_epilogue: dtype must be 1 (bf16) or 2 (f16)");
  if (epilogue < 1 || epilogue > 3) return fail("bsg_op_gemm_epilogue: epilogue must be 1 (bias + GELU), 2 (bias + fp32 residual) or 3 (times saved gelu')");
  if ((epilogue != 3 && !bias) || (epilogue != 1 && !aux)) return fail("bsg_op_gemm_epilogue: the epilogue's bias / aux operand is missing");
  if (K % 64 || N % 16 || N <= 192) return fail("bsg_op_gemm_epilogue: K must be a multiple of 64, N of 16 and above 192");
  hipStream_t st = (hipStream_t)stream;
  return dtype == BSG_DTYPE_BF16 ? op_gemm_epilogue_impl<bf16_t>(st, epilogue, M, N, K, A, W, bias, aux, out, out2)
                                 : op_gemm_epilogue_impl<f16_t>(st, epilogue, M, N, K, A, W, bias, aux, out, out2);
}

int bsg_op_attention(void* stream, int dtype, int which, int S, int nh, int hp, int wp, const void* qkv, const void* rel_cat,
                     const void* rel_catT, const void* dout, void* out, float* lse2, void* dqkv, void* scratch,
                     size_t scratch_bytes) {
  if (dtype == BSG_DTYPE_BF16) return op_attention_impl<bf16_t>(stream, which, S, nh, hp, wp, qkv, rel_cat, rel_catT, dout, out, lse2, dqkv, scratch, scratch_bytes);
  if (dtype == BSG_DTYPE_F16) return op_attention_impl<f16_t>(stream, which, S, nh, hp, wp, qkv, rel_cat, rel_catT, dout, out, lse2, dqkv, scratch, scratch_bytes);
  return fail("bsg_op_attention: dtype must be 1 (bf16) or 2 (f16)");
}

int bsg_tif_image(void* stream, int C, int H, int W, int in_dtype, const void* bands, const uint8_t* nodata, uint8_t* out_rgb,
                  void* scratch) {
  if (!bands || !out_rgb || !scratch) return fail("bsg_tif_image: null argument");
  if (C != 4 && C != 8) return fail("bsg_tif_image: expected 4 or 8 bands, got %d", C);
  if (in_dtype != 0 && in_dtype != 1) return fail("bsg_tif_image: in_dtype must be 0 (f32) or 1 (u16)");
  if (H <= 0 || W <= 0) return fail("bsg_tif_image: bad geometry");
  hipStream_t st = (hipStream_t)stream;
  TifArgs a{};
  a.bands = bands; a.nodata = nodata; a.keys = (int*)scratch; a.out = out_rgb; a.dtype = in_dtype; a.C = C; a.hw = (long)H * W;
  const unsigned grid = (unsigned)std::min<long>((a.hw + 255) / 256, 2048);
  hipLaunchKernelGGL(tif_init_kernel, dim3(1), dim3(64), 0, st, a.keys);
  hipLaunchKernelGGL(tif_min_kernel, dim3(grid), dim3(256), 0, st, a);
  hipLaunchKernelGGL(tif_max_kernel, dim3(grid), dim3(256), 0, st, a);
  hipLaunchKernelGGL(tif_map_kernel, dim3(grid), dim3(256), 0, st, a);
  CHECK_LAUNCH();
  return 0;
}

int bsg_train_aug(void* stream, int batch, int h, int w, const float* img, const uint8_t* mask, const int32_t* params,
                  const float* color, const float* noise, const float mean[3], const float std[3], float* out,
                  uint8_t* mask_out, float* scratch) {
  if (!img || !params || !out) return fail("bsg_train_aug: null argument");
  if ((mask == nullptr) != (mask_out == nullptr)) return fail("bsg_train_aug: mask and mask_out go together");
  if (color && !scratch) return fail("bsg_train_aug: colour parameters need the scratch plane (batch*3*h*w floats)");
  if (color && (h < 3 || w < 3)) return fail("bsg_train_aug: sharpness needs an image of at least 3x3 pixels");
  AugArgs a{};
  a.img = img; a.mask = mask; a.params = (const int*)params; a.color = color; a.noise = noise; a.out = out;
  a.mask_out = mask_out; a.tmp = scratch;
  a.B = batch; a.H = h; a.W = w;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.istd[c] = 1.f / std[c]; }
  const long n = (long)batch * h * w;
  const dim3 grid((unsigned)std::min<long>((n + 255) / 256, 65535));
  if (color) hipLaunchKernelGGL(train_aug_color_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(train_aug_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  CHECK_LAUNCH();
  return 0;
}

int bsg_train_aug_bwd(void* stream, int batch, int h, int w, const float* grad_out, const float* img, const int32_t* params,
                      const float* color, const float std[3], float* grad_img, float* scratch) {
  if (!grad_out || !params || !grad_img) return fail("bsg_train_aug_bwd: null argument");
  const long n = (long)batch * h * w;
  const dim3 grid((unsigned)std::min<long>((n + 255) / 256, 65535));
  if (!color) {
    hipLaunchKernelGGL(train_aug_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, grad_out, (const int*)params, grad_img,
                       batch, h, w, 1.f / std[0], 1.f / std[1], 1.f / std[2]);
    CHECK_LAUNCH();
    return 0;
  }
  if (!img || !scratch) return fail("bsg_train_aug_bwd: colour parameters need img and the scratch planes (2*batch*3*h*w floats)");
  AugArgs f{};
  f.img = img; f.params = (const int*)params; f.color = color; f.tmp = scratch; f.B = batch; f.H = h; f.W = w;
  hipLaunchKernelGGL(train_aug_color_kernel, grid, dim3(256), 0, (hipStream_t)stream, f);
  AugBwdArgs a{};
  a.gout = grad_out; a.img = img; a.params = (const int*)params; a.color = color; a.tmp = scratch; a.g2 = scratch + 3 * n;
  a.gin = grad_img; a.B = batch; a.H = h; a.W = w;
  for (int c = 0; c < 3; ++c) a.istd[c] = 1.f / std[c];
  hipLaunchKernelGGL(train_aug_sharp_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(train_aug_color_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  CHECK_LAUNCH();
  return 0;
}

int bsg_confusion_update(void* stream, long n, int K, int ignore_index, const int64_t* pred_i64, const uint8_t* pred_u8,
                         const uint8_t* target, uint64_t* confmat) {
  if ((!pred_i64 && !pred_u8) || !target || !confmat) return fail("bsg_confusion_update: null argument");
  if (K <= 0 || K > 16) return fail("bsg_confusion_update: K must be in 1..16");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 1024)), dim3(256), 0, (hipStream_t)stream,
                     (const long long*)pred_i64, pred_u8, target, n, K, ignore_index, (unsigned long long*)confmat);
  CHECK_LAUNCH();
  return 0;
}

int bsg_vote_paste(void* stream, int n_crops, const uint8_t* masks, int hin, int win, int crop, const int32_t* crops,
                   uint8_t* counter, int mh, int mw, int K) {
  if (!masks || !crops || !counter) return fail("bsg_vote_paste: null argument");
  if (n_crops <= 0) return 0;
  hipLaunchKernelGGL(vote_paste_kernel, dim3((crop * crop + 255) / 256, n_crops), dim3(256), 0, (hipStream_t)stream, masks,
                     counter, (const int*)crops, n_crops, hin, win, crop, mh, mw, K);
  CHECK_LAUNCH();
  return 0;
}

int bsg_decode_hf(void* stream, int batch, int h, int w, int K, const float* pred, const float* palette,
                  const float* mean3, const float* std3, uint8_t* out_u8) {
  if (!pred || !palette || !mean3 || !std3 || !out_u8) return fail("bsg_decode_hf: null argument");
  if (batch <= 0 || h <= 0 || w <= 0 || K <= 0 || K > 255) return fail("bsg_decode_hf: bad geometry");
  const long n = (long)batch * h * w;
  hipLaunchKernelGGL(decode_hf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred, palette,
                     out_u8, batch, h, w, K, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
  CHECK_LAUNCH();
  return 0;
}

int bsg_tile_frontend(void* stream, const uint8_t* mosaic, int mh, int mw, int n_crops, const int32_t* crops, int crop,
                      int S, const int32_t* coef, const int32_t* bounds, int kmax, const float* mean3,
                      const float* std3, float* out, uint8_t* out_u8) {
  if (!mosaic || !crops || !coef || !bounds || !mean3 || !std3) return fail("bsg_tile_frontend: null argument");
  if (!out && !out_u8) return fail("bsg_tile_frontend: no output buffer");
  if (crop <= 0 || S <= 0 || kmax <= 0 || mh <= 0 || mw <= 0) return fail("bsg_tile_frontend: bad geometry");
  if (n_crops <= 0) return 0;
  hipLaunchKernelGGL(tile_frontend_kernel, dim3((S * S + 255) / 256, n_crops), dim3(256), 0, (hipStream_t)stream, mosaic, mh,
                     mw, (const int*)crops, S, (const int*)coef, (const int*)bounds, kmax, mean3[0], mean3[1], mean3[2],
                     std3[0], std3[1], std3[2], out, out_u8);
  CHECK_LAUNCH();
  return 0;
}

int bsg_vote_argmax(void* stream, const uint8_t* counter, long n_pixels, int K, uint8_t* out) {
  if (!counter || !out) return fail("bsg_vote_argmax: null argument");
  hipLaunchKernelGGL(vote_argmax_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, counter,
                     out, n_pixels, K);
  CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
