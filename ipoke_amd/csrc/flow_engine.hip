// Native executor of the conditional MaCow flow (reference models/modules/INN/INN.py:446-481,
// macow2.py:821-1117).  The engine owns the *topology* -- the ordered layer program, the reference's
// state-dict names, the flat fp32 parameter layout, the matrix-core weight shadows and the workspace
// plan -- and issues the per-layer kernels of elementwise.hip / mcf.hip / gemm.hip back to back on
// the caller's stream: no host synchronisation, no allocation, capturable in a hipGraph.  Weight
// gradient GEMMs are forked onto a side stream so that they overlap the serial data-gradient chain.
//
// Host code only (compiled by hipcc together with the kernels).
#include <functional>
#include <memory>
#include <vector>
#include <string>
#include <cstring>

#include "common.h"

namespace ipoke {

enum OpType { OP_ACTNORM = 0, OP_MCF = 1, OP_NICE = 2, OP_LU = 3 };
enum TensorKind { TK_PARAM = 0, TK_IDX_FWD = 1, TK_IDX_BWD = 2, TK_FLAG = 3, TK_FBUF = 4 };

struct TensorInfo {
  std::string name;
  int64_t offset;            // floats (params) or int32 entries (perm); -1 for flags
  std::vector<int64_t> shape;
  int kind;
};

struct Op {
  int type = 0;
  int C = 0;                                   // active channels
  // actnorm / shuffle
  int c0 = 0, Cn = 0;
  int64_t p_ls = -1, p_bias = -1, idx_fwd = -1, idx_bwd = -1;
  // mcf
  int order = 0;
  int64_t p_w1 = -1, p_b = -1, p_g = -1, p_v = -1;
  int Cp = 0, K1p = 0, K2p = 0, K3p = 0, Hq = 0, H = 0;
  int64_t sh_w1 = -1, sh_w1t = -1, sh_w2 = -1, sh_w2t = -1;
  int64_t wn_off = -1;
  // nice
  int cin = 0, cout = 0, z_off = 0, z_stride = 1, t_off = 0, t_stride = 1;
  int64_t p_c1 = -1, p_c2 = -1;               // conv1.weight, conv2.weight (conv3: p_b, p_g, p_v)
  int Kc1 = 0, Kc3 = 0;
  int64_t sh_c1 = -1, sh_c1t = -1, sh_c2 = -1, sh_c2t = -1, sh_c3 = -1, sh_c3t = -1;
  // workspace (elements resolved per call)
  int64_t ws_a = -1, ws_b = -1, ws_c = -1, ws_d = -1, ws_e = -1, ws_f = -1, ws_g = -1;
  int slot = -1;                               // log-det slot index
  int mcf_idx = -1;                            // running index among the MCF ops (batched weight gradients)
  int nice_idx = -1;                           // running index among the NICE ops (batched weight gradients)
  int hidK = 0;                                // NICE: input channels of conv3 = hidden (+ cond_channels with condition_nice)
  int level = 0;                               // multi-scale level the op belongs to
  int fuse_act = -1;                           // MCF: index of the ActNorm executed inside this layer's kernels
  bool fused = false;                          // ActNorm: executed by the preceding MCF layer (forward / backward)
  bool unit_head = false;                      // MCF: first of the six ops of a MaCowUnit that one fused launch executes
  int unit_of = -1;                            // index of the unit's first op for every op inside such a unit
  int lu_idx = -1;                             // OP_LU: index into the LU job table
  int an_next = -1;                            // NICE: index of the stand-alone ActNorm (+ Shuffle) right behind it, run in the same launches
  int an_prev = -1;                            // ActNorm: index of the coupling whose launches execute it
  int pair_unit = -1;                          // ActNorm / NICE: head of the fused unit whose row-split BACKWARD launch differentiates this
                                               // (coupling, ActNorm) pair in front of it (ipoke_mcf_desc.pair); -1: launches of their own
};

struct RelayoutJobH {     // mirrors RelayoutJob of prep.hip
  long src_off, s_n, s_k, dstA, dstB, scale_off;
  int taps, n_real, k_real, A_rows_pad, A_inner_pad, B_rows_pad, B_inner_pad, B_rows_real, tile, tiles_k, block_start, frag_tiled;
};
struct WnJobH { long v_off, g_off, out_off; int rows, K, row_start; };
struct AdamTileJobH { long src_off, dstA, dstB; int N, K, tile_start, pad; };     // mirrors AdamTileJob of prep.hip
struct AdamSegH { long off, len; };                                               // mirrors AdamSeg of prep.hip
struct LsRefH { long off; int C; };
struct LuJobH { long p_l, p_u, p_logs, b_perm, b_sign, b_lmask, b_umask, b_eye, w_off; int C, pad; };   // mirrors LuJob of lu.hip

}  // namespace ipoke

using namespace ipoke;

struct ipoke_flow {
  ipoke_flow_config cfg;
  int esz = 2, e16 = 8, ks = 32;
  std::vector<TensorInfo> tensors;
  std::vector<Op> ops;
  int64_t n_params = 0, n_perm = 0, n_fbuf = 0;
  std::vector<LuJobH> lujobs; void* d_lujobs = nullptr; const float* fbuf = nullptr;
  int64_t shadow_elems = 0;           // T elements
  int64_t wn_rows = 0;
  int nslots = 0;
  int n_mcf = 0;
  // per-batch-size device tables of the batched weight-gradient / reduction launches
  int tab_B = 0; void* d_w1tab = nullptr; void* d_w2tab = nullptr; void* d_redtab = nullptr; int n_red = 0;
  void* d_ntab[3] = {nullptr, nullptr, nullptr};   // NICE conv1 / conv2 / conv3 weight-gradient batch entries, by nice_idx
  int n_nice = 0;
  std::vector<RelayoutJobH> rjobs; int rblocks = 0;
  // Optimizer-fused shadows (ipoke_flow_adam_range): plain 1x1 weights whose operands the Adam kernel writes itself (ajobs, in
  // tiles of 64 x 64), the relayout table WITHOUT them (rjobs2, own block numbering), and the gaps between them in the flat buffer
  std::vector<AdamTileJobH> ajobs; int atiles = 0;
  std::vector<RelayoutJobH> rjobs2; int rblocks2 = 0;
  std::vector<AdamSegH> asegs;
  void* d_ajobs = nullptr; void* d_rjobs2 = nullptr; void* d_rblockjob2 = nullptr; void* d_asegs = nullptr;
  std::vector<WnJobH> wjobs;
  std::vector<LsRefH> lsrefs;
  void* d_rjobs = nullptr; void* d_wjobs = nullptr; void* d_lsrefs = nullptr; void* d_rblockjob = nullptr;
  hipStream_t side = nullptr;
  std::vector<hipStream_t> lanes;     // extra streams for sub-batch lanes 1..kMaxLanes-1 (lane 0 = caller's stream)
  int n_lanes = 1;
  std::vector<hipEvent_t> events; size_t ev_next = 0;
  bool use_side = true;
  // hipGraph replay of the ~5000-launch layer programs: keyed by every pointer argument + batch + mode
  struct GraphEntry { std::vector<uintptr_t> key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; int state = 0; uint64_t used = 0; };
  std::vector<GraphEntry> graphs; hipStream_t cap = nullptr; bool use_graph = false; uint64_t tick = 0;
  // per level: flat parameter spans / weight-norm job and row ranges of its layers.* and priors.* tensors, op range
  // Units of the piecewise backward, in execution order: every MaCowStep of a level (kind 0, parameters in the layers.*
  // region) and every level's prior + shuffle (kind 1, priors.* region).  Consecutive units of one kind are adjacent in
  // the flat parameter buffer, the weight-norm job table and its row numbering.
  struct Unit { int64_t p_lo, p_hi; int wj_lo, wj_hi, row_lo, row_hi; int kind; int op_lo, op_hi; int64_t p2_lo = -1, p2_hi = -1; };
  std::vector<Unit> units;
  std::vector<int> red_first;          // first reduction-table entry of op i (size nops + 1)
  int last_fwd_B = 0; bool have_saved = false;
  int P = 64;
  // fused MaCowUnit launches on unit_split workgroups per sample (mcf_unit_split.hip; 1 = one workgroup per sample) and the
  // zero-initialised exchange scratch of those launches (sized for max_batch; used by one launch at a time: the chain's stream)
  int unit_split = 1; void* d_xchg = nullptr;
  // conv3 of a coupling net and the coupling transform in one launch (ipoke_conv3x3_coupling; forward and reverse passes) and that
  // launch's exchange scratch (initialised once, left in that state by every launch; the chain's stream only)
  bool coupling_fuse = false; void* d_cxchg = nullptr;
  // Both scratches carry a count of hand-off time-outs in their word 0 (a spin that gave up: the launch finished on garbage).  Every
  // eager pass ends with a one-thread kernel that copies the two words into pinned host memory and an event; the next entry point
  // (a) orders itself behind that event when it runs on ANOTHER stream (the scratches serve one launch at a time) and (b) once the
  // event has completed reads the words: non-zero -> the scratches are re-initialised and the call fails with IPOKE_ERR_STATE.
  // scratch of the deterministic split-K accumulation of the conv1 data gradients (ipoke_conv_desc.acc_scratch; the chain's stream only)
  void* d_acc = nullptr; int64_t acc_bytes = 0;
  unsigned* h_tmo = nullptr; hipEvent_t pass_ev = nullptr; hipStream_t pass_stream = nullptr; bool pass_recorded = false, pass_unchecked = false;
  int64_t xchg_bytes = 0;
  // every masked-conv layer is differentiated inside a fused MaCowUnit launch, which also writes the layer's input in the matrix
  // cores' dtype: the shifted-conv weight gradients then read that copy through the LDS-DMA GEMM instead of the fp32 state
  bool mcf_xop = false;
  // conv2 of every coupling net (plain 1x1, 73 % of the parameters) keeps ONLY its straight bf16 copy: the data gradient reads it
  // K-major (igemm_nn_glds, ipoke_conv_desc.w_kmajor), the optimizer writes it while it holds the updated values (adam_cast_kernel),
  // relayout never touches those tensors (IPOKE_C2_STRAIGHT=0: transposed shadow + relayout as in rounds 1-3)
  bool c2_straight = false;
  // Optimizer applied by the engine itself as soon as a piece of the backward pass is final (ipoke_flow_set_native_adam): the
  // Adam-amsgrad update and the shadow refresh of the piece's parameter ranges are queued on the ready stream by finish_piece
  // without a round trip through the host callback.
  struct NativeAdam { bool on = false; float* m = nullptr; float* v = nullptr; float* vmax = nullptr; float lr = 0, beta1 = 0, beta2 = 0, eps = 0, wd = 0,
                      grad_scale = 1; int step = 0, max_blocks = 0; } nadam;
};

namespace {

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct Builder {
  ipoke_flow& f;
  explicit Builder(ipoke_flow& fl) : f(fl) {}
  int64_t add_param(const std::string& name, std::vector<int64_t> shape) {
    int64_t n = 1; for (auto s : shape) n *= s;
    const int64_t off = f.n_params;
    f.tensors.push_back({name, off, shape, TK_PARAM});
    f.n_params = align_up(off + n, 4);
    return off;
  }
  int64_t add_idx(const std::string& name, int C, int kind) {
    const int64_t off = f.n_perm;
    f.tensors.push_back({name, off, {C}, kind});
    f.n_perm += C;
    return off;
  }
  void add_flag(const std::string& name) { f.tensors.push_back({name, -1, {}, TK_FLAG}); }
  int64_t add_fbuf(const std::string& name, std::vector<int64_t> shape) {
    int64_t n = 1; for (auto s : shape) n *= s;
    const int64_t off = f.n_fbuf;
    f.tensors.push_back({name, off, shape, TK_FBUF});
    f.n_fbuf = align_up(off + n, 4);
    return off;
  }
  // InvertibleConvLU1d (macow2.py:596-649): parameters l, u, log_s; buffers permutated, sign_s, lmask, umask, eye
  void lu(const std::string& pfx, int C) {
    Op op; op.type = OP_LU; op.C = C; op.lu_idx = (int)f.lujobs.size();
    LuJobH j{};
    j.C = C;
    j.p_l = add_param(pfx + ".l", {C, C}); j.p_u = add_param(pfx + ".u", {C, C}); j.p_logs = add_param(pfx + ".log_s", {C});
    j.b_perm = add_fbuf(pfx + ".permutated", {C, C}); j.b_sign = add_fbuf(pfx + ".sign_s", {C});
    j.b_lmask = add_fbuf(pfx + ".lmask", {C, C}); j.b_umask = add_fbuf(pfx + ".umask", {C, C}); j.b_eye = add_fbuf(pfx + ".eye", {C, C});
    j.w_off = (long)op.lu_idx * 4 * 64 * 64;
    f.lujobs.push_back(j);
    f.lsrefs.push_back({j.p_logs, C});          // log-det = P8 * sum(log_s): a constant per sample, like the ActNorms'
    f.ops.push_back(op);
  }
  int64_t add_shadow(int64_t elems) {
    const int64_t off = f.shadow_elems;
    f.shadow_elems = align_up(off + elems, 64);
    return off;
  }
  // W[n][k][tap] -> A[n][tap][k] (padded) and B[k][tap][n] (padded; rows >= B_rows_real zero)
  void relayout_pair(long src, long s_n, long s_k, int taps, int n_real, int k_real, long scale, long dstA, int A_rows_pad,
                     int A_inner_pad, long dstB, int B_rows_pad, int B_inner_pad, int B_rows_real, int frag_tiled = 0) {
    const int tile = taps == 1 ? 64 : 32;
    const int n_ext = std::max(A_rows_pad, B_inner_pad), k_ext = std::max(A_inner_pad, B_rows_pad);
    const int tiles_n = (n_ext + tile - 1) / tile, tiles_k = (k_ext + tile - 1) / tile;
    RelayoutJobH j{src, s_n, s_k, dstA, dstB, scale, taps, n_real, k_real, A_rows_pad, A_inner_pad, B_rows_pad, B_inner_pad,
                   B_rows_real, tile, tiles_k, f.rblocks, frag_tiled};
    f.rjobs.push_back(j);
    f.rblocks += tiles_n * tiles_k;
    // a dense, unpadded, un-normalised [n][k] matrix in whole 64 x 64 tiles: its operands are the tensor's cast and its transpose
    const bool fusable = taps == 1 && scale < 0 && !frag_tiled && s_k == 1 && s_n == k_real && n_real % 64 == 0 && k_real % 64 == 0 &&
                         A_rows_pad == n_real && A_inner_pad == k_real && B_rows_pad == k_real && B_inner_pad == n_real &&
                         B_rows_real == k_real;
    if (fusable) {
      f.ajobs.push_back({src, dstA, dstB, n_real, k_real, f.atiles, 0});
      f.atiles += (n_real / 64) * (k_real / 64);
    } else {
      j.block_start = f.rblocks2;
      f.rjobs2.push_back(j);
      f.rblocks2 += tiles_n * tiles_k;
    }
  }

  void actnorm(const std::string& pfx, int C_active, int c0, int Cn, const std::string* shuffle_pfx) {
    Op op; op.type = OP_ACTNORM; op.C = C_active; op.c0 = c0; op.Cn = Cn;
    if (!pfx.empty()) {
      op.p_ls = add_param(pfx + ".log_scale", {Cn, 1, 1});
      op.p_bias = add_param(pfx + ".bias", {Cn, 1, 1});
      add_flag(pfx + ".initialized");
      f.lsrefs.push_back({(long)op.p_ls, Cn});
    }
    if (shuffle_pfx) {
      op.idx_fwd = add_idx(*shuffle_pfx + ".forward_shuffle_idx", Cn, TK_IDX_FWD);
      op.idx_bwd = add_idx(*shuffle_pfx + ".backward_shuffle_idx", Cn, TK_IDX_BWD);
    }
    f.ops.push_back(op);
  }

  void mcf(const std::string& pfx, int C, int order) {
    const int Cc = f.cfg.cond_channels;
    Op op; op.type = OP_MCF; op.C = C; op.order = order;
    const int kh = order < 2 ? f.cfg.kernel_h : f.cfg.kernel_w, kw = order < 2 ? f.cfg.kernel_w : f.cfg.kernel_h;
    op.H = 4 * C;
    op.Cp = round_up(C, f.ks);
    op.K1p = 6 * op.Cp;
    op.K2p = round_up(op.H + Cc, f.ks);
    op.K3p = round_up(2 * C, f.ks);
    op.Hq = round_up(op.H, f.ks);
    const int K2 = op.H + Cc;
    op.p_w1 = add_param(pfx + ".net.shift_conv.weight", {op.H, C, kh, kw});
    add_flag(pfx + ".net.conv1x1.initialized");
    op.p_b = add_param(pfx + ".net.conv1x1.conv.bias", {2 * C});
    op.p_g = add_param(pfx + ".net.conv1x1.conv.weight_g", {2 * C, 1, 1, 1});
    op.p_v = add_param(pfx + ".net.conv1x1.conv.weight_v", {2 * C, K2, 1, 1});
    op.wn_off = f.wn_rows;
    f.wjobs.push_back({(long)op.p_v, (long)op.p_g, (long)op.wn_off, 2 * C, K2, (int)f.wn_rows});
    f.wn_rows += 2 * C;
    const int Hr = round_up(op.H, 16), N2r = round_up(2 * C, 16), Cr = round_up(C, 16);
    op.sh_w1 = add_shadow((int64_t)Hr * op.K1p);
    op.sh_w1t = add_shadow((int64_t)Cr * 6 * op.Hq);
    op.sh_w2 = add_shadow((int64_t)N2r * op.K2p);
    op.sh_w2t = add_shadow((int64_t)Hr * op.K3p);
    // masked-conv operands in the fragment-tiled order (prep.hip: tiled_offset)
    relayout_pair(op.p_w1, (long)C * 6, 6, 6, op.H, C, -1, op.sh_w1, Hr, op.Cp, op.sh_w1t, Cr, op.Hq, C, 1);
    relayout_pair(op.p_v, K2, 1, 1, 2 * C, K2, op.wn_off, op.sh_w2, N2r, op.K2p, op.sh_w2t, Hr, op.K3p, op.H, 1);
    op.slot = f.nslots++;
    op.mcf_idx = f.n_mcf++;
    f.ops.push_back(op);
  }

  void nice(const std::string& pfx, int C, bool skip, bool up, int factor) {
    const int hid = f.cfg.hidden;
    Op op; op.type = OP_NICE; op.C = C;
    op.cout = C / factor; op.cin = C - op.cout;
    if (skip && (C % 2 == 1)) skip = false;
    if (!skip) {
      const int z1 = up ? op.cin : op.cout;
      if (up) { op.z_off = 0; op.t_off = z1; } else { op.z_off = z1; op.t_off = 0; }
      op.z_stride = op.t_stride = 1;
    } else {
      op.z_stride = op.t_stride = 2;
      if (up) { op.z_off = 0; op.t_off = 1; } else { op.z_off = 1; op.t_off = 0; }
    }
    op.Kc1 = round_up(op.cin, f.e16);
    op.Kc3 = round_up(2 * op.cout, f.e16);
    // condition_nice: torch.cat([conv2 out, h]) in front of the second activation (macow_utils.py:328-332) -- conv3 is built with
    // hidden + h_channels inputs (:275-283); the hidden tile h2 gets that pitch and ELU(h) is copied behind conv2's columns
    const int hidK = hid + (f.cfg.condition_nice ? f.cfg.cond_channels : 0);
    op.hidK = hidK;
    op.p_c1 = add_param(pfx + ".net.conv1.weight", {hid, op.cin, 3, 3});
    op.p_c2 = add_param(pfx + ".net.conv2.weight", {hid, hid, 1, 1});
    add_flag(pfx + ".net.conv3.initialized");
    op.p_b = add_param(pfx + ".net.conv3.conv.bias", {2 * op.cout});
    op.p_g = add_param(pfx + ".net.conv3.conv.weight_g", {2 * op.cout, 1, 1, 1});
    op.p_v = add_param(pfx + ".net.conv3.conv.weight_v", {2 * op.cout, hidK, 3, 3});
    op.wn_off = f.wn_rows;
    f.wjobs.push_back({(long)op.p_v, (long)op.p_g, (long)op.wn_off, 2 * op.cout, hidK * 9, (int)f.wn_rows});
    f.wn_rows += 2 * op.cout;
    const int N3 = 2 * op.cout;
    op.sh_c1 = add_shadow((int64_t)hid * 9 * op.Kc1);
    op.sh_c1t = add_shadow((int64_t)op.cin * 9 * hid);
    op.sh_c2 = add_shadow((int64_t)hid * hid);
    op.sh_c2t = f.c2_straight ? -1 : add_shadow((int64_t)hid * hid);
    op.sh_c3 = add_shadow((int64_t)N3 * 9 * hidK);
    op.sh_c3t = add_shadow((int64_t)hidK * 9 * op.Kc3);      // (the data gradient only needs the first `hid` rows: h has no gradient)
    // conv1.weight [hid][cin][3][3]
    relayout_pair(op.p_c1, (long)op.cin * 9, 9, 9, hid, op.cin, -1, op.sh_c1, hid, op.Kc1, op.sh_c1t, op.cin, hid, op.cin);
    // conv2.weight [hid][hid]
    relayout_pair(op.p_c2, hid, 1, 1, hid, hid, -1, op.sh_c2, hid, hid, op.sh_c2t, hid, hid, hid);
    // conv3 weight_v [N3][hid][3][3] with weight-norm scale
    relayout_pair(op.p_v, (long)hidK * 9, 9, 9, N3, hidK, op.wn_off, op.sh_c3, N3, hidK, op.sh_c3t, hidK, op.Kc3, hidK);
    op.slot = f.nslots++;
    f.ops.push_back(op);
  }

  void unit(const std::string& pfx, int C) {
    mcf(pfx + ".conv1", C, 0);
    mcf(pfx + ".conv2", C, 1);
    actnorm(pfx + ".actnorm1", C, 0, C, nullptr);
    mcf(pfx + ".conv3", C, 2);
    mcf(pfx + ".conv4", C, 3);
    actnorm(pfx + ".actnorm2", C, 0, C, nullptr);
  }
  void step(const std::string& pfx, int C) {
    const std::string sh = pfx + ".conv1x1";
    actnorm(pfx + ".actnorm1", C, 0, C, &sh);
    unit(pfx + ".units1.0", C); unit(pfx + ".units1.1", C);
    nice(pfx + ".coupling1_up", C, false, true, 2);
    nice(pfx + ".coupling1_dn", C, false, false, 2);
    actnorm(pfx + ".actnorm2", C, 0, C, nullptr);
    unit(pfx + ".units2.0", C); unit(pfx + ".units2.1", C);
    nice(pfx + ".coupling2_up", C, true, true, 2);
    nice(pfx + ".coupling2_dn", C, true, false, 2);
  }
};

// The reference registers layers.*, then priors.*, then shuffle_layers.* (macow2.py:835-863); the
// state-dict (and hence the flat parameter) order follows registration, the execution order does not.
int build(ipoke_flow& f) {
  const ipoke_flow_config& c = f.cfg;
  Builder b(f);
  const int L = c.n_levels;
  std::vector<std::vector<Op>> level_ops(L), prior_ops(L), shuf_ops(L);
  int C = c.z_channels, factor = c.factor;
  const int cstep = c.z_channels / c.factor;
  std::vector<int> Cs(L), fs(L);
  for (int l = 0; l < L; ++l) { Cs[l] = C; fs[l] = factor; C -= cstep; --factor; }
  std::vector<std::vector<ipoke_flow::Unit>> step_units(L);
  std::vector<ipoke_flow::Unit> prior_units(L);
  for (int l = 0; l < L; ++l) {
    f.ops.clear();
    for (int s = 0; s < c.num_steps[l]; ++s) {
      ipoke_flow::Unit u{f.n_params, 0, (int)f.wjobs.size(), 0, (int)f.wn_rows, 0, 0, (int)f.ops.size(), 0};
      b.step("flow.layers." + std::to_string(l) + "." + std::to_string(s), Cs[l]);
      u.p_hi = f.n_params; u.wj_hi = (int)f.wjobs.size(); u.row_hi = (int)f.wn_rows; u.op_hi = (int)f.ops.size();
      step_units[l].push_back(u);
    }
    level_ops[l] = f.ops;
  }
  for (int l = 0; l < L; ++l) {
    f.ops.clear();
    ipoke_flow::Unit u{f.n_params, 0, (int)f.wjobs.size(), 0, (int)f.wn_rows, 0, 1, 0, 0};
    const std::string pfx = "flow.priors." + std::to_string(l);
    const std::string sh = pfx + ".conv1x1";
    b.actnorm("", Cs[l], 0, Cs[l], &sh);                         // bare shuffle
    b.nice(pfx + ".coupling", Cs[l], false, true, fs[l]);
    const int cout = Cs[l] / fs[l];
    b.actnorm(pfx + ".actnorm", Cs[l], Cs[l] - cout, cout, nullptr);
    u.p_hi = f.n_params; u.wj_hi = (int)f.wjobs.size(); u.row_hi = (int)f.wn_rows;
    prior_units[l] = u;
    prior_ops[l] = f.ops;
  }
  for (int l = 0; l < L; ++l) {
    f.ops.clear();
    const std::string sh = "flow.shuffle_layers." + std::to_string(l);
    prior_units[l].p2_lo = f.n_params;
    if (c.use1x1) b.lu(sh, Cs[l]); else b.actnorm("", Cs[l], 0, Cs[l], &sh);
    prior_units[l].p2_hi = f.n_params;
    if (prior_units[l].p2_hi == prior_units[l].p2_lo) prior_units[l].p2_lo = prior_units[l].p2_hi = -1;
    shuf_ops[l] = f.ops;
  }
  f.ops.clear();
  f.units.clear();
  for (int l = 0; l < L; ++l) {
    const int base = (int)f.ops.size();
    for (auto& o : level_ops[l]) { f.ops.push_back(o); f.ops.back().level = l; }
    for (auto u : step_units[l]) { u.op_lo += base; u.op_hi += base; f.units.push_back(u); }
    ipoke_flow::Unit pu = prior_units[l];
    pu.op_lo = (int)f.ops.size();
    for (auto& o : prior_ops[l]) { f.ops.push_back(o); f.ops.back().level = l; }
    for (auto& o : shuf_ops[l]) { f.ops.push_back(o); f.ops.back().level = l; }
    pu.op_hi = (int)f.ops.size();
    f.units.push_back(pu);
  }
  int k = 0;
  for (auto& o : f.ops) if (o.type == OP_MCF) o.mcf_idx = k++;     // execution order
  f.n_nice = 0;
  for (auto& o : f.ops) if (o.type == OP_NICE) o.nice_idx = f.n_nice++;
  // MaCowUnit: MCF, MCF, ActNorm, MCF, MCF, ActNorm -- the ActNorm runs inside the preceding MCF kernels
  static const bool nofuse = getenv("IPOKE_NO_ACTNORM_FUSION") != nullptr;
  for (size_t i = 0; !nofuse && i + 1 < f.ops.size(); ++i) {
    Op& a = f.ops[i]; Op& b2 = f.ops[i + 1];
    if (a.type == OP_MCF && b2.type == OP_ACTNORM && b2.idx_fwd < 0 && b2.p_ls >= 0 && b2.c0 == 0 && b2.Cn == a.C &&
        a.C % 4 == 0 && c.z_channels % 4 == 0 && a.level == b2.level) {
      a.fuse_act = (int)i + 1; b2.fused = true;
    }
  }
  // MaCowUnit = MCF, MCF(+ActNorm), ActNorm, MCF, MCF(+ActNorm), ActNorm of one width: one fused launch per direction
  // (mcf_unit.hip) where the matrix-core dtype supports it
  static const bool nounit = getenv("IPOKE_NO_UNIT_FUSION") != nullptr;
  for (size_t i = 0; !nounit && i + 5 < f.ops.size(); ++i) {
    const Op* o = &f.ops[i];
    const bool pat = o[0].type == OP_MCF && o[1].type == OP_MCF && o[2].type == OP_ACTNORM && o[3].type == OP_MCF &&
                     o[4].type == OP_MCF && o[5].type == OP_ACTNORM && o[0].fuse_act < 0 && o[1].fuse_act == (int)i + 2 &&
                     o[3].fuse_act < 0 && o[4].fuse_act == (int)i + 5 && o[2].fused && o[5].fused && o[0].unit_of < 0 &&
                     o[0].C == o[1].C && o[0].C == o[3].C && o[0].C == o[4].C;
    if (pat && ipoke_macow_unit_supported(o[0].C, c.cond_channels, c.dtype)) {
      f.ops[i].unit_head = true;
      for (int k = 0; k < 6; ++k) f.ops[i + k].unit_of = (int)i;
      i += 5;
    }
  }
  // coupling -> ActNorm (+ Shuffle) pairs: one launch per direction (ipoke_affine_actnorm_fwd / ipoke_actnorm_affine_bwd)
  static const bool noan = getenv("IPOKE_NO_AN_FUSION") != nullptr;     // developer A/B
  for (size_t i = 0; !noan && i + 1 < f.ops.size(); ++i) {
    Op& a = f.ops[i]; Op& b2 = f.ops[i + 1];
    // (the pair kernels stage whole rows of a sample: ipoke_actnorm_affine_bwd needs P * ld <= 8192 floats of LDS)
    if (a.type == OP_NICE && b2.type == OP_ACTNORM && !b2.fused && b2.unit_of < 0 && b2.Cn <= 256 && f.P * c.z_channels <= 8192) {
      a.an_next = (int)i + 1; b2.an_prev = (int)i;
    }
  }
  static const bool noxop = getenv("IPOKE_NO_MCF_XOP") != nullptr;      // developer A/B
  f.mcf_xop = c.dtype == IPOKE_BF16 && !noxop;
  for (const Op& op : f.ops) if (op.type == OP_MCF && op.unit_of < 0) f.mcf_xop = false;
  return IPOKE_OK;
}

// ---- workspace plan ---------------------------------------------------------------------------
struct Plan {
  int64_t bytes = 0;
  int64_t cond_act = 0, slots = 0, ls_const = 0, partials = 0, partials_lane = 0, dld = 0, dbias_part = 0;
  int64_t state0 = 0, state_stride = 0;     // saved states S[0..nops]
  int64_t g0 = 0, g1 = 0;                   // gradient ping-pong
  int64_t tmp_h1 = 0, tmp_h2 = 0, tmp_zc = 0;   // shared hidden buffers when nothing is saved
  int64_t lu = 0;                               // [W | W^-1 | wl | wu] of every LU 1x1 conv
};
constexpr int kMaxLanes = 4;
constexpr int kMaxUnitSplit = 4;          // rows of dbias_part per sample
constexpr int kDefaultUnitSplit = 4;
int64_t take(int64_t& cur, int64_t bytes) { const int64_t o = cur; cur = align_up(cur + bytes, 256); return o; }

int max_splitk(const ipoke_flow& f, int B) {
  const int M = B * f.P;
  {   // the stationary-input 3x3 kernel splits over 64-channel chunks and knows its own tile height
    const int s8 = ipoke_conv3x3_skinny_splitk(M, f.cfg.hidden, f.cfg.dtype);
    if (s8 > 0) return s8;
  }
  const int tiles = ceil_div(M, 64);
  const int nkb = ceil_div(9 * f.cfg.hidden, 128 / f.esz);
  static const int target = getenv("IPOKE_SPLITK_TARGET") ? atoi(getenv("IPOKE_SPLITK_TARGET")) : 384;   // workgroups per N tile
  int s = (target + tiles - 1) / tiles;
  if (s < 1) s = 1;
  if (s > nkb) s = nkb;
  if (s > 32) s = 32;
  return s;
}

// mode 0: inference (forward without saves / reverse / init); 1: training forward+backward
Plan make_plan(ipoke_flow& f, int B, int mode) {
  Plan p;
  const int64_t M = (int64_t)B * f.P, ld = f.cfg.z_channels, hid = f.cfg.hidden;
  int64_t cur = 0;
  p.cond_act = take(cur, M * f.cfg.cond_channels * f.esz);
  p.slots = take(cur, (int64_t)f.nslots * B * 4 * 4);
  p.ls_const = take(cur, 256);
  p.dld = take(cur, (int64_t)B * 4);
  p.partials_lane = align_up((int64_t)32 * M * 64 * 4, 256);   // upper bound of splitk * M_lane * 64 floats
  p.partials = take(cur, kMaxLanes * p.partials_lane);
  p.state_stride = align_up(M * ld * 4, 256);
  p.lu = take(cur, (int64_t)f.lujobs.size() * 4 * 64 * 64 * 4 + 256);
  if (mode == 0) {
    p.state0 = take(cur, 2 * p.state_stride);
    p.tmp_h1 = take(cur, M * hid * f.esz);
    p.tmp_h2 = take(cur, M * (hid + (f.cfg.condition_nice ? f.cfg.cond_channels : 0)) * f.esz);
    p.tmp_zc = take(cur, M * 64 * f.esz);
    p.bytes = cur;
    return p;
  }
  p.state0 = take(cur, (int64_t)(f.ops.size() + 1) * p.state_stride);
  p.g0 = take(cur, p.state_stride);
  p.g1 = take(cur, p.state_stride);
  p.dbias_part = take(cur, (int64_t)f.ops.size() * (kMaxUnitSplit * B + 1) * 128 * 4);
  for (auto& op : f.ops) {
    if (op.type == OP_MCF) {
      op.ws_a = take(cur, M * op.K2p * f.esz);        // a2 (ELU(cat[c, h]))
      op.ws_b = take(cur, M * op.C * 4);              // scale
      op.ws_c = take(cur, M * op.K3p * f.esz);        // dparams
      op.ws_d = take(cur, M * op.Hq * f.esz);         // dc
      if (f.mcf_xop) op.ws_e = take(cur, M * op.Cp * f.esz);      // x in the compute dtype (written by the unit's backward kernel)
    } else if (op.type == OP_NICE) {
      op.ws_a = take(cur, M * hid * f.esz);           // h1
      op.ws_b = take(cur, M * op.hidK * f.esz);       // h2 (condition_nice: [h2 | ELU(cond)])
      op.ws_c = take(cur, M * op.cout * 4);           // scale
      op.ws_d = take(cur, M * op.Kc3 * f.esz);        // dparams
      op.ws_e = take(cur, M * hid * f.esz);           // dp2
      op.ws_f = take(cur, M * hid * f.esz);           // dp1
      op.ws_g = take(cur, M * op.Kc1 * f.esz);        // zc: conditioning channels, dense dtype copy
    }
  }
  p.bytes = cur;
  return p;
}

// Execution context of one *lane*: a contiguous range of samples [b0, b0 + B) of the batch, processed on its own stream.
// The flow is strictly sequential per sample, and at the shipped batch sizes every kernel on the chain is bound by its
// fixed launch/prologue/epilogue latency rather than by throughput; independent lanes on separate streams overlap those
// latencies.  All row-indexed workspace buffers are laid out for the full batch, a lane addresses its row range.
struct Ctx {
  ipoke_flow* f; int B; int64_t M; int ld; int dtype; hipStream_t s;
  const float* params; const int32_t* perm; const unsigned char* shadow; unsigned char* ws; Plan plan;
  int b0 = 0, Bfull = 0, lane = 0;
  const float* wn_scale() const { return reinterpret_cast<const float*>(shadow); }
  const float* wn_inv() const { return reinterpret_cast<const float*>(shadow) + align_up(f->wn_rows, 64); }
  const unsigned char* sh(int64_t elem_off) const { return shadow + shadow_base() + elem_off * f->esz; }
  int64_t shadow_base() const { return 2 * align_up(f->wn_rows, 64) * 4; }
  int64_t row0() const { return (int64_t)b0 * f->P; }
  float* state(int i) const { return reinterpret_cast<float*>(ws + plan.state0 + (int64_t)i * plan.state_stride) + row0() * ld; }
  template <typename T> T* at(int64_t off) const { return reinterpret_cast<T*>(ws + off); }
  // row-indexed buffer at byte offset `off` with `row_bytes` per position
  void* rows(int64_t off, int64_t row_bytes) const { return ws + off + row0() * row_bytes; }
  float* rowsf(int64_t off, int64_t row_floats) const { return reinterpret_cast<float*>(ws + off) + row0() * row_floats; }
  void* stream() const { return reinterpret_cast<void*>(s); }
  float* slot(int k) const { return at<float>(plan.slots) + ((int64_t)k * Bfull + b0) * 4; }
  float* partials() const { return reinterpret_cast<float*>(ws + plan.partials + (int64_t)lane * plan.partials_lane); }
  void* cond() const { return rows(plan.cond_act, (int64_t)f->cfg.cond_channels * f->esz); }
  float* dld() const { return at<float>(plan.dld) + b0; }
  // per-sample partial sums of op op_index: `parts` rows of ldp floats per sample (fused unit launches split over workgroups)
  float* dbp(int op_index, int ldp, int parts = 1) const {
    return at<float>(plan.dbias_part) + (int64_t)op_index * (kMaxUnitSplit * Bfull + 1) * 128 + (int64_t)b0 * parts * ldp;
  }
};

void set_conv8(ipoke_conv_desc& d, int B, int k, int pad) {
  std::memset(&d, 0, sizeof(d));
  d.NB = B; d.Di = 1; d.Hi = 8; d.Wi = 8; d.Do = 1; d.Ho = 8; d.Wo = 8;
  d.kd = 1; d.kh = k; d.kw = k; d.sd = d.sh = d.sw = 1; d.pd = 0; d.ph = pad; d.pw = pad;
  d.c_cstride = 1; d.splitk = 1;
}
void set_a_state(ipoke_conv_desc& d, const float* state, int ld, int off, int stride, int creal, int kc) {
  d.A = state; d.a_f32 = 1; d.a_sn = 64L * ld; d.a_sd = 0; d.a_sh = 8L * ld; d.a_sw = ld; d.a_sc = stride;
  d.a_coff = off; d.Kc_real = creal; d.Kc = kc;
}
void set_a_dense(ipoke_conv_desc& d, const void* act, int ld, int kc) {
  d.A = act; d.a_f32 = 0; d.a_sn = 64L * ld; d.a_sd = 0; d.a_sh = 8L * ld; d.a_sw = ld; d.a_sc = 1;
  d.a_coff = 0; d.Kc_real = kc; d.Kc = kc;
}

int nice_splitk(const Ctx& c) { return max_splitk(*c.f, c.B); }
// coupling `a` transforms exactly the channels coupling `b` conditions on (up -> dn pairs): a's transform also writes b's operand
// coupling op `i` follows a fused MaCowUnit (ops i-6 .. i-1): the unit's forward kernel writes the coupling's conditioning operand
// (IPOKE_NO_UNIT_ZC=1: a launch of ipoke_extract_cols instead, as in rounds 1-2)
bool unit_feeds(const ipoke_flow* f, size_t i) {
  static const int on = getenv("IPOKE_NO_UNIT_ZC") ? 0 : 1;
  if (!(on && i >= 6 && i < f->ops.size() && f->ops[i].type == OP_NICE && f->ops[i - 6].unit_head)) return false;
  const Op& nx = f->ops[i];          // the unit kernel gathers the conditioning columns from ITS state tile: they must lie inside its C channels
  return nx.z_off + (nx.cin - 1) * nx.z_stride < f->ops[i - 6].C && nx.Kc1 % 2 == 0;
}
bool nice_feeds(const Op& a, const Op& b) {
  static const int on = getenv("IPOKE_NO_EXTRACT_FUSION") ? 0 : 1;
  return on && a.type == OP_NICE && b.type == OP_NICE && b.z_off == a.t_off && b.z_stride == a.t_stride && b.cin == a.cout;
}

// coupling net forward: conv1 -> ELU -> conv2 -> ELU -> conv3 (split-K partials)
// `zc_ready`: the conditioning operand was already written by the preceding coupling's transform (ipoke_affine_*_ext)
void nice_conv3_desc(const Ctx& c, const Op& op, void* h2, ipoke_conv_desc& d) {
  set_conv8(d, c.B, 3, 1);
  set_a_dense(d, h2, op.hidK, op.hidK);
  d.W = c.sh(op.sh_c3); d.ldw = 9 * op.hidK; d.Nout = 2 * op.cout; d.C = c.partials(); d.c_f32 = 1; d.ldc = 64;
  d.splitk = nice_splitk(c);
}
// conv3 and the coupling transform of this net run as ONE launch (ipoke_conv3x3_coupling) at this batch size
bool nice_fused(const Ctx& c, const Op& op) {
  return c.f->coupling_fuse && c.f->d_cxchg && ipoke_conv3x3_coupling_splitk((int)c.M, op.hidK, c.dtype) > 0;
}
int nice_net(const Ctx& c, const Op& op, const float* in, void* h1, void* h2, void* zc, bool zc_ready = false, bool with_conv3 = true) {
  const int hid = c.f->cfg.hidden;
  ipoke_conv_desc d;
  if (!zc_ready) {
    int rc0 = ipoke_extract_cols(in, c.ld, op.z_off, op.z_stride, op.cin, zc, op.Kc1, c.M, c.dtype, c.stream());
    if (rc0) return rc0;
  }
  set_conv8(d, c.B, 3, 1);
  set_a_dense(d, zc, op.Kc1, op.Kc1);
  d.W = c.sh(op.sh_c1); d.ldw = 9 * op.Kc1; d.Nout = hid; d.act = IPOKE_ACT_ELU; d.C = h1; d.ldc = hid;
  int rc = ipoke_conv_forward(&d, c.dtype, c.stream()); if (rc) return rc;
  set_conv8(d, c.B, 1, 0);
  set_a_dense(d, h1, hid, hid);
  d.W = c.sh(op.sh_c2); d.ldw = hid; d.Nout = hid; d.act = IPOKE_ACT_ELU; d.C = h2; d.ldc = op.hidK;
  rc = ipoke_conv_forward(&d, c.dtype, c.stream()); if (rc) return rc;
  if (op.hidK > hid) {      // condition_nice: ELU(cat[conv2 out, h]) = [ELU(conv2 out) | ELU(h)], the second half is the shared activated map
    rc = ipoke_copy_cols(c.cond(), c.f->cfg.cond_channels, static_cast<unsigned char*>(h2) + (size_t)hid * c.f->esz, op.hidK,
                         c.f->cfg.cond_channels, c.M, c.dtype, c.stream());
    if (rc) return rc;
  }
  if (!with_conv3) return IPOKE_OK;
  nice_conv3_desc(c, op, h2, d);
  return ipoke_conv_forward(&d, c.dtype, c.stream());
}
void nice_affine_desc(const Ctx& c, const Op& op, ipoke_affine_desc& a) {
  a.raw = c.partials(); a.nsplit = nice_splitk(c); a.split_stride = c.M * 64; a.ldraw = 64;
  a.bias = c.params + op.p_b; a.Cp = op.cout; a.t_off = op.t_off; a.t_stride = op.t_stride; a.P = c.f->P; a.ld = c.ld;
}
void mcf_desc(const Ctx& c, const Op& op, ipoke_mcf_desc& d) {
  std::memset(&d, 0, sizeof(d));
  d.ld = c.ld; d.C = op.C; d.B = c.B; d.cond = c.cond(); d.Cc = c.f->cfg.cond_channels;
  d.W1 = c.sh(op.sh_w1); d.W2 = c.sh(op.sh_w2); d.bias2 = c.params + op.p_b; d.order = op.order;
  d.W1T = c.sh(op.sh_w1t); d.W2T = c.sh(op.sh_w2t);
}

// W, W^-1, wl, wu of every LU 1x1 conv into the workspace (one launch; the matrices depend on the current parameters)
int lu_prepare_all(const Ctx& c) {
  ipoke_flow* f = c.f;
  if (f->lujobs.empty()) return IPOKE_OK;
  IPK_REQUIRE(f->fbuf != nullptr, "use1x1: ipoke_flow_set_float_buffers has not been called");
  IPK_REQUIRE(f->n_lanes == 1, "use1x1 does not support sub-batch lanes");
  return ipoke_lu_prepare(c.params, f->fbuf, c.at<float>(c.plan.lu), f->d_lujobs, (int)f->lujobs.size(), c.stream());
}
const float* lu_mat(const Ctx& c, const Op& op, int which) {     // 0 W, 1 W^-1
  return c.at<float>(c.plan.lu) + (int64_t)op.lu_idx * 4 * 64 * 64 + (int64_t)which * op.C * op.C;
}

int common_checks(ipoke_flow* f, int B) {
  IPK_REQUIRE(f != nullptr, "null flow handle");
  IPK_REQUIRE(B >= 1 && B <= f->cfg.max_batch, "batch exceeds max_batch of the flow handle");
  return IPOKE_OK;
}

struct WgEntryH { long a_off, y_off, w_off; int kh, kw, ph, pw; long sh_off = 0; };      // mirrors TnBatchEntry of gemm.hip
struct RedEntryH { long src, dst; int ld, ncols, rmul, pad; };   // rmul: rows per sample (ipoke_reduce_rows_multi sums R * rmul rows)

void drop_graphs(ipoke_flow* f) {
  for (auto& g : f->graphs) {
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (g.graph) (void)hipGraphDestroy(g.graph);
  }
  f->graphs.clear();
}

// (Re)build the device tables for batch size B.  Offsets are relative to the workspace base (bytes) / the
// gradient buffer (floats), so the tables stay valid as long as B does.  Not called under graph capture
// after the first (warm-up) step.
int ensure_tables(ipoke_flow* f, int B, const Plan& plan) {
  if (f->tab_B == B && f->d_w1tab) return IPOKE_OK;
  IPK_REQUIRE((int)sizeof(WgEntryH) == ipoke_wgrad_batch_entry_size() && (int)sizeof(RedEntryH) == ipoke_reduce_entry_size(),
              "batch table layout mismatch");
  std::vector<WgEntryH> w1(f->n_mcf), w2(f->n_mcf);
  std::vector<WgEntryH> nt[3] = {std::vector<WgEntryH>(f->n_nice), std::vector<WgEntryH>(f->n_nice), std::vector<WgEntryH>(f->n_nice)};
  std::vector<RedEntryH> red;
  f->red_first.assign(f->ops.size() + 1, 0);
  for (size_t i = 0; i < f->ops.size(); ++i) {
    const Op& op = f->ops[i];
    f->red_first[i] = (int)red.size();
    const long dbp = (long)(plan.dbias_part / 4) + (long)i * (kMaxUnitSplit * B + 1) * 128;
    // partial sums written by a fused unit launch (its four masked convs and two ActNorms) have unit_split rows per sample
    const int rmul = op.unit_of >= 0 || op.pair_unit >= 0 ? f->unit_split : 1;
    if (op.type == OP_MCF) {
      const McfGeom g = mcf_geom(op.order);
      w1[op.mcf_idx] = {f->mcf_xop ? (long)op.ws_e : (long)(plan.state0 + (int64_t)i * plan.state_stride), (long)op.ws_d, (long)op.p_w1, g.kh,
                        g.kw, -g.oy, -g.ox};
      w2[op.mcf_idx] = {(long)op.ws_a, (long)op.ws_c, (long)op.p_v, 1, 1, 0, 0};
      red.push_back({dbp, (long)op.p_b, 2 * op.C, 2 * op.C, rmul, 0});
    } else if (op.type == OP_NICE) {
      nt[0][op.nice_idx] = {(long)op.ws_g, (long)op.ws_f, (long)op.p_c1, 3, 3, 1, 1};     // conv1: saved conditioning columns x d(pre-act 1)
      nt[1][op.nice_idx] = {(long)op.ws_a, (long)op.ws_e, (long)op.p_c2, 1, 1, 0, 0, (long)op.sh_c2};     // conv2: h1 x d(pre-act 2); its operand copy (fused Adam)
      nt[2][op.nice_idx] = {(long)op.ws_b, (long)op.ws_d, (long)op.p_v, 3, 3, 1, 1};      // conv3: h2 x d(raw shift / scale)
      red.push_back({dbp, (long)op.p_b, 2 * op.cout, 2 * op.cout, rmul, 0});
    } else if (op.p_ls >= 0) {
      red.push_back({dbp, (long)op.p_ls, 2 * op.Cn, op.Cn, rmul, 0});
      red.push_back({dbp + op.Cn, (long)op.p_bias, 2 * op.Cn, op.Cn, rmul, 0});
    }
  }
  drop_graphs(f);    // captured launches hold the old table addresses
  f->red_first[f->ops.size()] = (int)red.size();
  if (f->d_w1tab) { (void)hipFree(f->d_w1tab); (void)hipFree(f->d_w2tab); (void)hipFree(f->d_redtab); }
  for (int k = 0; k < 3; ++k) {
    if (f->d_ntab[k]) (void)hipFree(f->d_ntab[k]);
    IPK_HIP(hipMalloc(&f->d_ntab[k], nt[k].size() * sizeof(WgEntryH)));
    IPK_HIP(hipMemcpy(f->d_ntab[k], nt[k].data(), nt[k].size() * sizeof(WgEntryH), hipMemcpyHostToDevice));
  }
  IPK_HIP(hipMalloc(&f->d_w1tab, w1.size() * sizeof(WgEntryH)));
  IPK_HIP(hipMalloc(&f->d_w2tab, w2.size() * sizeof(WgEntryH)));
  IPK_HIP(hipMalloc(&f->d_redtab, red.size() * sizeof(RedEntryH)));
  IPK_HIP(hipMemcpy(f->d_w1tab, w1.data(), w1.size() * sizeof(WgEntryH), hipMemcpyHostToDevice));
  IPK_HIP(hipMemcpy(f->d_w2tab, w2.data(), w2.size() * sizeof(WgEntryH), hipMemcpyHostToDevice));
  IPK_HIP(hipMemcpy(f->d_redtab, red.data(), red.size() * sizeof(RedEntryH), hipMemcpyHostToDevice));
  f->n_red = (int)red.size();
  f->tab_B = B;
  return IPOKE_OK;
}

// pieces of the piecewise backward: groups of consecutive units (steps / priors), last unit first, of roughly equal parameter counts
std::vector<std::pair<int, int>> backward_pieces(const ipoke_flow* f, int npieces) {
  std::vector<std::pair<int, int>> pieces;            // (lowest unit, highest unit)
  const int U = (int)f->units.size();
  int64_t total = 0;
  for (const auto& u : f->units) total += u.p_hi - u.p_lo;
  const int np = npieces < 1 ? 1 : (npieces > U ? U : npieces);
  // Relative sizes: all pieces equal except the last two (the lowest layers, whose gradients are ready last).  Everything behind the
  // chain's last kernel -- the last weight gradients, then the optimizer and operand refresh of the last piece -- is exposed before
  // the next forward pass can start; a smaller last piece shortens that tail (IPOKE_PIECE_TAPER="a,b": second-to-last, last).
  static double ta = -1, tb = -1;
  if (ta < 0) {
    ta = 0.5; tb = 0.25;           // measured (c2, 24 pieces): 51.14 / 51.63 ms against 51.34 / 51.99 with equal pieces; 0.3 / 0.15 the same
    if (const char* e = getenv("IPOKE_PIECE_TAPER")) { double a = 0, b = 0; if (sscanf(e, "%lf,%lf", &a, &b) == 2 && a > 0 && b > 0) { ta = a; tb = b; } }
  }
  std::vector<double> w((size_t)np, 1.0);
  if (np >= 4) { w[(size_t)np - 2] = ta; w[(size_t)np - 1] = tb; }
  double wsum = 0; for (double x : w) wsum += x;
  int hi = U - 1;
  int64_t acc = 0; double target_acc = 0;
  int64_t done = 0;
  for (int u = U - 1; u >= 0; --u) {
    acc += f->units[u].p_hi - f->units[u].p_lo;
    const int idx = (int)pieces.size();
    const int left = np - idx;
    const double want = (double)total * (target_acc + w[(size_t)(idx < np ? idx : np - 1)]) / wsum;      // cumulative target after this piece
    if (u == 0 || (left > 1 && ((double)(done + acc) >= want || u == left - 1))) {
      pieces.push_back({u, hi}); hi = u - 1; done += acc; acc = 0; target_acc += w[(size_t)(idx < np ? idx : np - 1)];
    }
  }
  return pieces;
}
// flat-parameter ranges [begin, end) of a piece: layers.*, priors.* and (use1x1) shuffle_layers.* are separate regions
void piece_param_ranges(const ipoke_flow* f, int lvl_lo, int lvl_hi, int64_t p0[3], int64_t p1[3]) {
  for (int k = 0; k < 3; ++k) { p0[k] = -1; p1[k] = -1; }
  for (int u = lvl_lo; u <= lvl_hi; ++u) {
    const auto& un = f->units[u];
    const int k = un.kind;
    if (p0[k] < 0) p0[k] = un.p_lo;
    p1[k] = un.p_hi;
    if (un.p2_lo >= 0) { if (p0[2] < 0) p0[2] = un.p2_lo; p1[2] = un.p2_hi; }
  }
}

hipEvent_t next_event(ipoke_flow* f) {
  hipEvent_t e = f->events[f->ev_next];
  f->ev_next = (f->ev_next + 1) % f->events.size();
  return e;
}

}  // namespace

// ================================================================================================
extern "C" int ipoke_flow_create(const ipoke_flow_config* cfg, ipoke_flow** out) {
  IPK_REQUIRE(cfg && out, "null argument");
  IPK_REQUIRE(cfg->dtype == IPOKE_F32 || cfg->dtype == IPOKE_BF16, "bad dtype");
  IPK_REQUIRE(cfg->n_levels >= 1 && cfg->n_levels <= 32 && cfg->n_levels < cfg->factor, "num_steps must be shorter than factor");
  IPK_REQUIRE(cfg->z_channels % cfg->factor == 0 && cfg->z_channels <= 64, "flow_in_channels must be a multiple of factor, <= 64");
  IPK_REQUIRE((cfg->z_channels / cfg->factor) % 2 == 0, "channel step must be even (skip split)");
  IPK_REQUIRE(cfg->kernel_h == 2 && cfg->kernel_w == 3, "kernel_size must be [2,3] (shipped configs)");
  IPK_REQUIRE(cfg->hidden % 32 == 0 && cfg->cond_channels % 32 == 0, "hidden / cond widths must be multiples of 32");
  std::unique_ptr<ipoke_flow> f(new ipoke_flow());
  f->cfg = *cfg;
  f->esz = cfg->dtype == IPOKE_BF16 ? 2 : 4; f->e16 = 16 / f->esz; f->ks = 64 / f->esz;
  {
    const char* cs = getenv("IPOKE_C2_STRAIGHT");
    f->c2_straight = cfg->dtype == IPOKE_BF16 && cfg->hidden % 64 == 0 && !(cs && cs[0] == '0');
  }
  int rc = build(*f); if (rc) return rc;
  {   // gaps between the optimizer-fused tensors (both tables are in parameter order)
    int64_t pos = 0;
    for (size_t k = 0; k < f->ajobs.size(); ++k) {
      IPK_REQUIRE(f->ajobs[k].src_off >= pos, "fused tensors must be in parameter order");
      if (f->ajobs[k].src_off > pos) f->asegs.push_back({(long)pos, (long)(f->ajobs[k].src_off - pos)});
      pos = f->ajobs[k].src_off + (int64_t)f->ajobs[k].N * f->ajobs[k].K;
    }
    if (pos < f->n_params) f->asegs.push_back({(long)pos, (long)(f->n_params - pos)});
    for (size_t k = 1; k < f->rjobs.size(); ++k) IPK_REQUIRE(f->rjobs[k].src_off > f->rjobs[k - 1].src_off, "relayout jobs must be in parameter order");
  }
  const char* env = getenv("IPOKE_NO_SIDE_STREAM");
  f->use_side = !(env && env[0] == '1');
  const char* gr = getenv("IPOKE_GRAPH");
  f->use_graph = gr && gr[0] == '1';     // opt-in: see DESIGN.md (replay is slower than eager launches on ROCm 7.2)
  const char* ln = getenv("IPOKE_LANES");
  f->n_lanes = ln ? atoi(ln) : 1;
  if (f->n_lanes < 1) f->n_lanes = 1;
  if (f->n_lanes > kMaxLanes) f->n_lanes = kMaxLanes;
  {   // rows of a sample dealt to 2 / 4 workgroups in the fused unit launches (IPOKE_UNIT_SPLIT=1: one workgroup per sample)
    const char* us = getenv("IPOKE_UNIT_SPLIT");
    f->unit_split = us ? atoi(us) : kDefaultUnitSplit;
    if (f->unit_split != 2 && f->unit_split != 4) f->unit_split = 1;
    if (f->n_lanes > 1 || cfg->dtype != IPOKE_BF16) f->unit_split = 1;     // (lanes would share the scratch across streams)
  }
  f->coupling_fuse = f->n_lanes == 1 && cfg->dtype == IPOKE_BF16 && ipoke_conv3x3_coupling_splitk(64, cfg->hidden, cfg->dtype) > 0;
  {   // coupling -> ActNorm -> unit (MaCowStep): the pair's backward inside the unit's row-split backward launch (IPOKE_UNIT_PAIR=0: off)
    const char* up = getenv("IPOKE_UNIT_PAIR");
    const bool on = f->unit_split > 1 && f->n_lanes == 1 && (up ? atoi(up) != 0 : true);
    for (size_t h = 2; on && h < f->ops.size(); ++h) {
      const Op& u = f->ops[h]; Op& an = f->ops[h - 1]; Op& cp = f->ops[h - 2];
      if (u.unit_head && an.type == OP_ACTNORM && an.an_prev == (int)h - 2 && cp.type == OP_NICE && an.c0 == 0 && an.Cn == u.C &&
          cp.cout <= 32 && cp.t_off + (cp.cout - 1) * cp.t_stride < u.C) {
        an.pair_unit = (int)h; cp.pair_unit = (int)h;
      }
    }
  }
  *out = f.release();
  return IPOKE_OK;
}

// Device-side tables, the side stream and its events are created on first use so that the topology
// (names, shapes, offsets) can be inspected on a host without a GPU.
static int ensure_device(ipoke_flow* f) {
  if (f->d_rjobs) return IPOKE_OK;
  if (f->unit_split > 1 && !f->d_xchg) {
    const int64_t nb = ipoke_macow_unit_xchg_bytes(f->cfg.max_batch, f->unit_split);
    IPK_HIP(hipMalloc(&f->d_xchg, nb));
    IPK_HIP(hipMemset(f->d_xchg, 0, nb));
    f->xchg_bytes = nb;
  }
  if (!f->d_acc) {
    static const bool atomics = getenv("IPOKE_DGRAD_ATOMICS") != nullptr;      // developer A/B: the order-dependent atomic accumulation of rounds 1-5
    if (!atomics) {
      f->acc_bytes = ipoke_conv_acc_scratch_bytes(f->cfg.max_batch * f->P, 64, 32);
      IPK_HIP(hipMalloc(&f->d_acc, (size_t)f->acc_bytes));
      int rc = ipoke_conv_acc_scratch_init(f->d_acc, nullptr); if (rc) return rc;
      IPK_HIP(hipDeviceSynchronize());
    }
  }
  if (!f->h_tmo) {
    IPK_HIP(hipHostMalloc(reinterpret_cast<void**>(&f->h_tmo), 64, hipHostMallocMapped));
    std::memset(f->h_tmo, 0, 64);
    IPK_HIP(hipEventCreateWithFlags(&f->pass_ev, hipEventDisableTiming));
  }
  if (f->coupling_fuse && !f->d_cxchg) {
    IPK_HIP(hipMalloc(&f->d_cxchg, (size_t)ipoke_conv3x3_coupling_xchg_bytes()));
    int rc = ipoke_conv3x3_coupling_xchg_init(f->d_cxchg, nullptr); if (rc) return rc;
    IPK_HIP(hipDeviceSynchronize());
  }
  IPK_REQUIRE((int)sizeof(RelayoutJobH) == ipoke_relayout_job_size() && (int)sizeof(WnJobH) == ipoke_wn_job_size(),
              "job table layout mismatch");
  IPK_HIP(hipMalloc(&f->d_rjobs, f->rjobs.size() * sizeof(RelayoutJobH)));
  IPK_HIP(hipMemcpy(f->d_rjobs, f->rjobs.data(), f->rjobs.size() * sizeof(RelayoutJobH), hipMemcpyHostToDevice));
  {   // block -> job map of the relayout launch (saves every block a 12-step search through the job table)
    std::vector<int32_t> bj((size_t)f->rblocks);
    for (size_t k = 0; k < f->rjobs.size(); ++k) {
      const int b0 = f->rjobs[k].block_start, b1 = k + 1 < f->rjobs.size() ? f->rjobs[k + 1].block_start : f->rblocks;
      for (int b = b0; b < b1; ++b) bj[b] = (int32_t)k;
    }
    IPK_HIP(hipMalloc(&f->d_rblockjob, bj.size() * sizeof(int32_t)));
    IPK_HIP(hipMemcpy(f->d_rblockjob, bj.data(), bj.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  IPK_REQUIRE((int)sizeof(AdamTileJobH) == ipoke_adam_tile_job_size() && (int)sizeof(AdamSegH) == ipoke_adam_seg_size(), "Adam table layout mismatch");
  if (!f->ajobs.empty()) {
    IPK_HIP(hipMalloc(&f->d_ajobs, f->ajobs.size() * sizeof(AdamTileJobH)));
    IPK_HIP(hipMemcpy(f->d_ajobs, f->ajobs.data(), f->ajobs.size() * sizeof(AdamTileJobH), hipMemcpyHostToDevice));
  }
  if (!f->asegs.empty()) {
    IPK_HIP(hipMalloc(&f->d_asegs, f->asegs.size() * sizeof(AdamSegH)));
    IPK_HIP(hipMemcpy(f->d_asegs, f->asegs.data(), f->asegs.size() * sizeof(AdamSegH), hipMemcpyHostToDevice));
  }
  if (!f->rjobs2.empty()) {
    IPK_HIP(hipMalloc(&f->d_rjobs2, f->rjobs2.size() * sizeof(RelayoutJobH)));
    IPK_HIP(hipMemcpy(f->d_rjobs2, f->rjobs2.data(), f->rjobs2.size() * sizeof(RelayoutJobH), hipMemcpyHostToDevice));
    std::vector<int32_t> bj((size_t)f->rblocks2);
    for (size_t k = 0; k < f->rjobs2.size(); ++k) {
      const int b0 = f->rjobs2[k].block_start, b1 = k + 1 < f->rjobs2.size() ? f->rjobs2[k + 1].block_start : f->rblocks2;
      for (int b = b0; b < b1; ++b) bj[b] = (int32_t)k;
    }
    IPK_HIP(hipMalloc(&f->d_rblockjob2, bj.size() * sizeof(int32_t)));
    IPK_HIP(hipMemcpy(f->d_rblockjob2, bj.data(), bj.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  IPK_HIP(hipMalloc(&f->d_wjobs, f->wjobs.size() * sizeof(WnJobH)));
  IPK_HIP(hipMemcpy(f->d_wjobs, f->wjobs.data(), f->wjobs.size() * sizeof(WnJobH), hipMemcpyHostToDevice));
  if (!f->lujobs.empty()) {
    IPK_REQUIRE((int)sizeof(LuJobH) == ipoke_lu_job_size(), "LU job table layout mismatch");
    IPK_HIP(hipMalloc(&f->d_lujobs, f->lujobs.size() * sizeof(LuJobH)));
    IPK_HIP(hipMemcpy(f->d_lujobs, f->lujobs.data(), f->lujobs.size() * sizeof(LuJobH), hipMemcpyHostToDevice));
  }
  IPK_HIP(hipMalloc(&f->d_lsrefs, f->lsrefs.size() * sizeof(LsRefH)));
  IPK_HIP(hipMemcpy(f->d_lsrefs, f->lsrefs.data(), f->lsrefs.size() * sizeof(LsRefH), hipMemcpyHostToDevice));
  {   // weight gradients are off the critical path: lowest priority so that chain kernels get the CUs first
    int least = 0, greatest = 0;
    IPK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    static const bool flat = getenv("IPOKE_SIDE_FLAT_PRIORITY") != nullptr;
    IPK_HIP(hipStreamCreateWithPriority(&f->side, hipStreamNonBlocking, flat ? 0 : least));
  }
  IPK_HIP(hipStreamCreateWithFlags(&f->cap, hipStreamNonBlocking));
  f->lanes.resize(kMaxLanes - 1);
  for (auto& l : f->lanes) IPK_HIP(hipStreamCreateWithFlags(&l, hipStreamNonBlocking));
  f->events.resize(256);
  for (auto& e : f->events) IPK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return IPOKE_OK;
}

extern "C" void ipoke_flow_destroy(ipoke_flow* f) {
  if (!f) return;
  if (f->d_rjobs) (void)hipFree(f->d_rjobs);
  if (f->d_wjobs) (void)hipFree(f->d_wjobs);
  if (f->d_lsrefs) (void)hipFree(f->d_lsrefs);
  if (f->d_lujobs) (void)hipFree(f->d_lujobs);
  if (f->d_rblockjob) (void)hipFree(f->d_rblockjob);
  if (f->d_ajobs) (void)hipFree(f->d_ajobs);
  if (f->d_asegs) (void)hipFree(f->d_asegs);
  if (f->d_rjobs2) (void)hipFree(f->d_rjobs2);
  if (f->d_rblockjob2) (void)hipFree(f->d_rblockjob2);
  if (f->d_w1tab) (void)hipFree(f->d_w1tab);
  for (int k = 0; k < 3; ++k) if (f->d_ntab[k]) (void)hipFree(f->d_ntab[k]);
  if (f->d_w2tab) (void)hipFree(f->d_w2tab);
  if (f->d_redtab) (void)hipFree(f->d_redtab);
  if (f->d_xchg) (void)hipFree(f->d_xchg);
  if (f->d_cxchg) (void)hipFree(f->d_cxchg);
  if (f->d_acc) (void)hipFree(f->d_acc);
  if (f->pass_ev) (void)hipEventDestroy(f->pass_ev);
  if (f->h_tmo) (void)hipHostFree(f->h_tmo);
  for (auto e : f->events) (void)hipEventDestroy(e);
  drop_graphs(f);
  if (f->side) (void)hipStreamDestroy(f->side);
  if (f->cap) (void)hipStreamDestroy(f->cap);
  for (auto l : f->lanes) if (l) (void)hipStreamDestroy(l);
  delete f;
}

/* Host only (no device access): the flat-parameter ranges ipoke_flow_backward_pieces(npieces) announces through its `ready` callback, in
 * callback order -- ranges[3 * i] = piece, ranges[3 * i + 1] = begin, ranges[3 * i + 2] = end (floats).  Returns the number of ranges
 * (<= max_ranges entries are written), negative on error.  Lets a data-parallel host size its per-slice shards before the first step. */
extern "C" int ipoke_flow_piece_ranges(const ipoke_flow* f, int npieces, int64_t* ranges, int max_ranges) {
  IPK_REQUIRE(f != nullptr && (ranges != nullptr || max_ranges == 0), "bad arguments");
  const std::vector<std::pair<int, int>> pieces = backward_pieces(f, npieces);
  int n = 0;
  for (size_t i = 0; i < pieces.size(); ++i) {
    int64_t p0[3], p1[3];
    piece_param_ranges(f, pieces[i].first, pieces[i].second, p0, p1);
    for (int kind = 0; kind < 3; ++kind) {
      if (p0[kind] < 0 || p1[kind] <= p0[kind]) continue;
      if (n < max_ranges) { ranges[3 * n] = (int64_t)i; ranges[3 * n + 1] = p0[kind]; ranges[3 * n + 2] = p1[kind]; }
      ++n;
    }
  }
  return n;
}

/* The stream the engine issues its weight gradients on during ipoke_flow_backward* (created on first use; owned by the flow).  Hosts
 * that keep further streams busy beside the backward pass (gradient exchange / optimizer, input prefetch) check theirs against it:
 * two busy streams on one hardware queue serialise (ipoke_amd/utils/streams.py). */
extern "C" void* ipoke_flow_side_stream(ipoke_flow* f) {
  if (!f || ensure_device(f) != IPOKE_OK) return nullptr;
  return reinterpret_cast<void*>(f->side);
}
extern "C" int64_t ipoke_flow_param_count(const ipoke_flow* f) { return f ? f->n_params : -1; }
extern "C" int64_t ipoke_flow_index_count(const ipoke_flow* f) { return f ? f->n_perm : -1; }
/* replay the layer programs (forward / reverse / one-shot backward) as captured hipGraphs (1) or launch them eagerly (0) */
extern "C" int ipoke_flow_set_graph(ipoke_flow* f, int enable) {
  IPK_REQUIRE(f != nullptr, "null flow handle");
  f->use_graph = enable != 0;
  return IPOKE_OK;
}
extern "C" int64_t ipoke_flow_float_buffer_count(const ipoke_flow* f) { return f ? f->n_fbuf : -1; }
extern "C" int ipoke_flow_set_float_buffers(ipoke_flow* f, const float* fbuf_dev) {
  IPK_REQUIRE(f != nullptr, "null flow handle");
  f->fbuf = fbuf_dev;
  return IPOKE_OK;
}
extern "C" int32_t ipoke_flow_tensor_count(const ipoke_flow* f) { return f ? (int32_t)f->tensors.size() : -1; }
extern "C" int32_t ipoke_flow_op_count(const ipoke_flow* f) { return f ? (int32_t)f->ops.size() : -1; }
extern "C" int ipoke_flow_tensor_info(const ipoke_flow* f, int i, char* name, int name_cap, int64_t* offset, int32_t* ndim,
                                      int64_t* shape4, int32_t* kind) {
  IPK_REQUIRE(f && i >= 0 && i < (int)f->tensors.size() && name && offset && ndim && shape4 && kind, "bad arguments");
  const TensorInfo& t = f->tensors[i];
  IPK_REQUIRE((int)t.name.size() + 1 <= name_cap, "name buffer too small");
  std::memcpy(name, t.name.c_str(), t.name.size() + 1);
  *offset = t.offset; *ndim = (int32_t)t.shape.size(); *kind = t.kind;
  for (int k = 0; k < 4; ++k) shape4[k] = k < (int)t.shape.size() ? t.shape[k] : 1;
  return IPOKE_OK;
}
// Introspection for tests / documentation: 32 int64 fields describing op i of the layer program.
extern "C" int ipoke_flow_op_info(const ipoke_flow* f, int i, int64_t* out32) {
  IPK_REQUIRE(f && out32 && i >= 0 && i < (int)f->ops.size(), "bad arguments");
  const Op& o = f->ops[i];
  const int64_t v[32] = {o.type, o.C, o.c0, o.Cn, o.p_ls, o.p_bias, o.idx_fwd, o.idx_bwd, o.order, o.p_w1, o.p_b, o.p_g, o.p_v,
                         o.sh_w1, o.sh_w1t, o.sh_w2, o.sh_w2t, o.wn_off, o.cin, o.cout, o.z_off, o.z_stride, o.t_off, o.t_stride,
                         o.p_c1, o.p_c2, o.sh_c1, o.sh_c1t, o.sh_c2, o.sh_c2t, o.sh_c3, o.sh_c3t};
  for (int k = 0; k < 32; ++k) out32[k] = v[k];
  return IPOKE_OK;
}
/* byte offset of the first weight shadow inside the shadow buffer (after the weight-norm row tables) */
extern "C" int64_t ipoke_flow_shadow_base(const ipoke_flow* f) { return f ? 2 * align_up(f->wn_rows, 64) * 4 : -1; }

extern "C" int64_t ipoke_flow_shadow_bytes(const ipoke_flow* f) {
  if (!f) return -1;
  return 2 * align_up(f->wn_rows, 64) * 4 + f->shadow_elems * f->esz;
}
extern "C" int64_t ipoke_flow_workspace_bytes(ipoke_flow* f, int B, int training) {
  if (!f || B < 1) return -1;
  return make_plan(*f, B, training ? 1 : 0).bytes;
}

extern "C" int ipoke_flow_prepare_weights(ipoke_flow* f, const float* params, void* shadow, void* stream) {
  IPK_REQUIRE(f && params && shadow, "null argument");
  { int rc0 = ensure_device(f); if (rc0) return rc0; }
  float* wn_scale = reinterpret_cast<float*>(shadow);
  float* wn_inv = wn_scale + align_up(f->wn_rows, 64);
  int rc = ipoke_wn_scale_multi(params, wn_scale, wn_inv, f->d_wjobs, (int)f->wjobs.size(), (int)f->wn_rows, stream);
  if (rc) return rc;
  void* sh = reinterpret_cast<unsigned char*>(shadow) + 2 * align_up(f->wn_rows, 64) * 4;
  return ipoke_relayout_multi(params, sh, wn_scale, f->d_rjobs, (int)f->rjobs.size(), f->rblocks,
                              reinterpret_cast<const int32_t*>(f->d_rblockjob), f->cfg.dtype, stream);
}

static int prepare_range(ipoke_flow* f, const float* params, void* shadow, int64_t begin, int64_t end, bool skip_fused, void* stream) {
  float* wn_scale = reinterpret_cast<float*>(shadow);
  float* wn_inv = wn_scale + align_up(f->wn_rows, 64);
  // job tables are in parameter order: the jobs whose source tensor starts inside [begin, end) are contiguous
  int w0 = 0, w1 = 0;
  while (w0 < (int)f->wjobs.size() && f->wjobs[w0].v_off < begin) ++w0;
  w1 = w0;
  while (w1 < (int)f->wjobs.size() && f->wjobs[w1].v_off < end) ++w1;
  if (w1 > w0) {
    const int row0 = f->wjobs[w0].row_start;
    const int row1 = w1 < (int)f->wjobs.size() ? f->wjobs[w1].row_start : (int)f->wn_rows;
    int rc = ipoke_wn_scale_multi_range(params, wn_scale, wn_inv, f->d_wjobs, w0, w1 - w0, row0, row1 - row0, stream);
    if (rc) return rc;
  }
  const std::vector<RelayoutJobH>& jobs = skip_fused ? f->rjobs2 : f->rjobs;
  const int nblocks = skip_fused ? f->rblocks2 : f->rblocks;
  int j0 = 0, j1 = 0;
  while (j0 < (int)jobs.size() && jobs[j0].src_off < begin) ++j0;
  j1 = j0;
  while (j1 < (int)jobs.size() && jobs[j1].src_off < end) ++j1;
  if (j1 > j0) {
    const int b0 = jobs[j0].block_start;
    const int b1 = j1 < (int)jobs.size() ? jobs[j1].block_start : nblocks;
    void* sh = reinterpret_cast<unsigned char*>(shadow) + 2 * align_up(f->wn_rows, 64) * 4;
    return ipoke_relayout_multi_range(params, sh, wn_scale, skip_fused ? f->d_rjobs2 : f->d_rjobs, (int)jobs.size(), b0, b1 - b0,
                                      reinterpret_cast<const int32_t*>(skip_fused ? f->d_rblockjob2 : f->d_rblockjob), f->cfg.dtype, stream);
  }
  return IPOKE_OK;
}

/* Optimizer inside the piecewise backward pass (single process): after ipoke_flow_set_native_adam(..., step >= 1) every
 * ipoke_flow_backward_pieces applies Adam-amsgrad to the ranges of a piece and refreshes their shadows on the ready stream as soon as the
 * piece is final -- what the `ready` callback of a data-parallel run does after its gradient exchange.  m = NULL switches it off. */
extern "C" int ipoke_flow_set_native_adam(ipoke_flow* f, float* m, float* v, float* vmax, float lr, float beta1, float beta2, float eps,
                                          float weight_decay, int step, float grad_scale, int max_blocks) {
  IPK_REQUIRE(f != nullptr, "null flow handle");
  ipoke_flow::NativeAdam& A = f->nadam;
  A.on = m != nullptr;
  if (!A.on) return IPOKE_OK;
  IPK_REQUIRE(v && vmax && step >= 1, "bad optimizer state");
  A.m = m; A.v = v; A.vmax = vmax; A.lr = lr; A.beta1 = beta1; A.beta2 = beta2; A.eps = eps; A.wd = weight_decay; A.step = step;
  A.grad_scale = grad_scale; A.max_blocks = max_blocks;
  return IPOKE_OK;
}

extern "C" int ipoke_flow_prepare_weights_range(ipoke_flow* f, const float* params, void* shadow, int64_t begin, int64_t end,
                                                void* stream) {
  IPK_REQUIRE(f && params && shadow && begin >= 0 && end >= begin, "bad arguments");
  { int rc0 = ensure_device(f); if (rc0) return rc0; }
  return prepare_range(f, params, shadow, begin, end, false, stream);
}

/* Adam-amsgrad update of params[begin, end) AND the refresh of every weight shadow derived from it, on `stream`:
 *   1. the plain 1x1 weights (conv2 of the coupling nets) by ipoke_adam_amsgrad_shadow_tiles -- update + both operands in one pass,
 *   2. everything between them by ipoke_adam_amsgrad_segments,
 *   3. weight-norm scales and the relayout of the remaining (3x3, masked, weight-normed) tensors.
 * [begin, end) must cover whole tensors (the ranges ipoke_flow_backward_pieces announces, or [0, param_count)). */
static int adam_range(ipoke_flow* f, float* params, const float* grads, float* m, float* v, float* vmax, void* shadow, int64_t begin, int64_t end,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, int max_blocks, void* stream,
                      bool skip_tiles = false);

extern "C" int ipoke_flow_adam_range(ipoke_flow* f, float* params, const float* grads, float* m, float* v, float* vmax, void* shadow,
                                     int64_t begin, int64_t end, float lr, float beta1, float beta2, float eps, float weight_decay,
                                     int step, float grad_scale, int max_blocks, void* stream) {
  IPK_REQUIRE(f && params && grads && m && v && vmax && shadow && begin >= 0 && end >= begin && end <= f->n_params, "bad arguments");
  { int rc0 = ensure_device(f); if (rc0) return rc0; }
  return adam_range(f, params, grads, m, v, vmax, shadow, begin, end, lr, beta1, beta2, eps, weight_decay, step, grad_scale, max_blocks, stream);
}

// skip_tiles: the plain 1x1 tensors of the range were already updated in the epilogue of their weight-gradient launches (wgrad_adam)
static int adam_range(ipoke_flow* f, float* params, const float* grads, float* m, float* v, float* vmax, void* shadow, int64_t begin, int64_t end,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, int max_blocks, void* stream,
                      bool skip_tiles) {
  if (end == begin) return IPOKE_OK;
  int a0 = 0, a1 = 0;
  while (a0 < (int)f->ajobs.size() && f->ajobs[a0].src_off < begin) ++a0;
  a1 = a0;
  while (a1 < (int)f->ajobs.size() && f->ajobs[a1].src_off < end) ++a1;
  if (a1 > a0 && !skip_tiles) {
    IPK_REQUIRE(f->ajobs[a1 - 1].src_off + (int64_t)f->ajobs[a1 - 1].N * f->ajobs[a1 - 1].K <= end, "range must cover whole tensors");
    const int t0 = f->ajobs[a0].tile_start, t1 = a1 < (int)f->ajobs.size() ? f->ajobs[a1].tile_start : f->atiles;
    void* sh = reinterpret_cast<unsigned char*>(shadow) + 2 * align_up(f->wn_rows, 64) * 4;
    // c2_straight: the tensor's one operand is its own cast, written linearly; else the 64 x 64-tile kernel that also transposes
    int rc = f->c2_straight
        ? ipoke_adam_amsgrad_cast_tiles(params, grads, m, v, vmax, sh, f->d_ajobs, (int)f->ajobs.size(), t0, t1 - t0, lr, beta1, beta2,
                                        eps, weight_decay, step, grad_scale, max_blocks, f->cfg.dtype, stream)
        : ipoke_adam_amsgrad_shadow_tiles(params, grads, m, v, vmax, sh, f->d_ajobs, (int)f->ajobs.size(), t0, t1 - t0, lr, beta1, beta2,
                                          eps, weight_decay, step, grad_scale, max_blocks, f->cfg.dtype, stream);
    if (rc) return rc;
  }
  int s0 = 0, s1 = 0;
  while (s0 < (int)f->asegs.size() && f->asegs[s0].off + f->asegs[s0].len <= begin) ++s0;
  s1 = s0;
  while (s1 < (int)f->asegs.size() && f->asegs[s1].off < end) ++s1;
  if (s1 > s0) {
    int per = max_blocks > 0 ? max_blocks / (s1 - s0) : 16;
    if (per < 2) per = 2; if (per > 32) per = 32;
    int rc = ipoke_adam_amsgrad_segments(params, grads, m, v, vmax, f->d_asegs, s0, s1 - s0, begin, end, lr, beta1, beta2, eps, weight_decay,
                                         step, grad_scale, per, stream);
    if (rc) return rc;
  }
  return prepare_range(f, params, shadow, begin, end, true, stream);
}

// ------------------------------------------------------------------------------------------------
// Lanes: split the batch into up to n_lanes contiguous sample ranges, lane 0 on the caller's stream.
static int make_lanes(const Ctx& full, int want, std::vector<Ctx>& lanes) {
  ipoke_flow* f = full.f;
  int n = want < 1 ? 1 : want;
  if (n > kMaxLanes) n = kMaxLanes;
  if (n > full.B) n = full.B;
  lanes.clear();
  int b0 = 0;
  for (int k = 0; k < n; ++k) {
    const int nb = (full.B - b0 + (n - k) - 1) / (n - k);
    Ctx c = full;
    c.b0 = b0; c.B = nb; c.M = (int64_t)nb * f->P; c.Bfull = full.B; c.lane = k;
    c.s = k == 0 ? full.s : f->lanes[k - 1];
    lanes.push_back(c);
    b0 += nb;
  }
  return IPOKE_OK;
}
// lanes 1.. wait for everything queued so far on the caller's stream
static int fork_lanes(ipoke_flow* f, hipStream_t from, const std::vector<Ctx>& lanes) {
  if (lanes.size() < 2) return IPOKE_OK;
  hipEvent_t e = next_event(f);
  IPK_HIP(hipEventRecord(e, from));
  for (size_t k = 1; k < lanes.size(); ++k) IPK_HIP(hipStreamWaitEvent(lanes[k].s, e, 0));
  return IPOKE_OK;
}
// `to` waits for the work queued so far on every lane (lanes that already run on `to` are skipped)
static int join_lanes(ipoke_flow* f, const std::vector<Ctx>& lanes, hipStream_t to) {
  for (size_t k = 0; k < lanes.size(); ++k) {
    if (lanes[k].s == to) continue;
    hipEvent_t e = next_event(f);
    IPK_HIP(hipEventRecord(e, lanes[k].s));
    IPK_HIP(hipStreamWaitEvent(to, e, 0));
  }
  return IPOKE_OK;
}

// Run `enqueue(stream)` either eagerly on the caller's stream (first call with a given argument set: it also performs
// the one-time allocations / attribute calls that are illegal under capture) or as a replay of the hipGraph captured on
// the second call.  The graph runs on the engine's own stream, ordered after / before the caller's stream by events.
template <typename Fn>
static int with_graph(ipoke_flow* f, std::vector<uintptr_t> key, hipStream_t caller, Fn enqueue) {
  if (!f->use_graph) return enqueue(caller);
  int rc0 = ensure_device(f); if (rc0) return rc0;
  ipoke_flow::GraphEntry* e = nullptr;
  for (auto& g : f->graphs) if (g.key == key) { e = &g; break; }
  if (!e) {
    if (f->graphs.size() >= 12) {
      size_t old = 0;
      for (size_t i = 1; i < f->graphs.size(); ++i) if (f->graphs[i].used < f->graphs[old].used) old = i;
      if (f->graphs[old].exec) (void)hipGraphExecDestroy(f->graphs[old].exec);
      if (f->graphs[old].graph) (void)hipGraphDestroy(f->graphs[old].graph);
      f->graphs.erase(f->graphs.begin() + old);
    }
    ipoke_flow::GraphEntry ne; ne.key = std::move(key); ne.used = ++f->tick;
    f->graphs.push_back(std::move(ne));
    return enqueue(caller);
  }
  e->used = ++f->tick;
  if (e->state < 0) return enqueue(caller);          // capture failed before: stay eager
  if (!e->exec) {
    const std::vector<uintptr_t> k = e->key;
    IPK_HIP(hipStreamBeginCapture(f->cap, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue(f->cap);
    hipGraph_t g = nullptr;
    const hipError_t er = hipStreamEndCapture(f->cap, &g);
    // enqueue() may have rebuilt tables and dropped every entry: look the entry up again
    e = nullptr;
    for (auto& ge : f->graphs) if (ge.key == k) { e = &ge; break; }
    if (rc != IPOKE_OK) { if (g) (void)hipGraphDestroy(g); if (e) e->state = -1; return rc; }
    if (er != hipSuccess || !g || !e) { (void)hipGetLastError(); if (g) (void)hipGraphDestroy(g); if (e) e->state = -1; return enqueue(caller); }
    hipGraphExec_t ex = nullptr;
    if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess) {
      (void)hipGetLastError(); (void)hipGraphDestroy(g); e->state = -1; return enqueue(caller);
    }
    e->graph = g; e->exec = ex; e->state = 1;
  }
  hipEvent_t a = next_event(f);
  IPK_HIP(hipEventRecord(a, caller)); IPK_HIP(hipStreamWaitEvent(f->cap, a, 0));
  IPK_HIP(hipGraphLaunch(e->exec, f->cap));
  hipEvent_t b = next_event(f);
  IPK_HIP(hipEventRecord(b, f->cap)); IPK_HIP(hipStreamWaitEvent(caller, b, 0));
  return IPOKE_OK;
}

// ---- hand-off scratch guard (see ipoke_flow::h_tmo) -------------------------------------------------------------------------------
__global__ void handoff_poll_kernel(const unsigned* unit_xchg, const unsigned* coupling_xchg, unsigned* host_words) {
  if (threadIdx.x == 0) {
    host_words[0] = unit_xchg ? unit_xchg[0] : 0u;
    host_words[1] = coupling_xchg ? coupling_xchg[0] : 0u;
  }
}
static int reinit_scratch(ipoke_flow* f, hipStream_t s) {
  if (f->d_xchg) IPK_HIP(hipMemsetAsync(f->d_xchg, 0, (size_t)f->xchg_bytes, s));
  if (f->d_cxchg) { int rc = ipoke_conv3x3_coupling_xchg_init(f->d_cxchg, reinterpret_cast<void*>(s)); if (rc) return rc; }
  return IPOKE_OK;
}
static bool stream_is_capturing(hipStream_t s) {      // the caller may be capturing the whole pass into a graph of its own (set_sample_graph)
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
  return st != hipStreamCaptureStatusNone;
}
static int pass_begin(ipoke_flow* f, hipStream_t s) {
  int rc = ensure_device(f); if (rc) return rc;
  if (!f->pass_recorded || stream_is_capturing(s)) return IPOKE_OK;
  if (s != f->pass_stream) IPK_HIP(hipStreamWaitEvent(s, f->pass_ev, 0));      // one launch at a time per scratch: order behind the last pass
  if (f->pass_unchecked) {
    const hipError_t q = hipEventQuery(f->pass_ev);
    if (q == hipSuccess) {
      f->pass_unchecked = false;
      const unsigned tu = f->h_tmo[0], tc = f->h_tmo[1];
      if (tu | tc) {
        rc = reinit_scratch(f, s); if (rc) return rc;
        f->h_tmo[0] = f->h_tmo[1] = 0;
        return fail(IPOKE_ERR_STATE, "hand-off time-out in an earlier pass of this flow (row-split unit launches: " + std::to_string(tu) +
                                     ", fused conv3 + coupling launches: " + std::to_string(tc) +
                                     "): the states / gradients of that pass are invalid; the exchange scratches have been re-initialised");
      }
    } else {
      (void)hipGetLastError();       // hipErrorNotReady: the host is ahead, look again at the next entry
    }
  }
  return IPOKE_OK;
}
static int pass_end(ipoke_flow* f, hipStream_t s) {
  if (!f->h_tmo || (!f->d_xchg && !f->d_cxchg) || stream_is_capturing(s)) return IPOKE_OK;
  hipLaunchKernelGGL(handoff_poll_kernel, dim3(1), dim3(64), 0, s, reinterpret_cast<const unsigned*>(f->d_xchg),
                     reinterpret_cast<const unsigned*>(f->d_cxchg), f->h_tmo);
  IPK_LAUNCH_CHECK();
  IPK_HIP(hipEventRecord(f->pass_ev, s));
  f->pass_stream = s; f->pass_recorded = true; f->pass_unchecked = true;
  return IPOKE_OK;
}
/* Test hook (include/ipoke_hip_dev.h): bumps the time-out word of the unit (which = 0) or coupling (which = 1) scratch on `stream`, as a
 * hand-off that gave up would -- the engine must report it at the entry point after the next polled pass. */
__global__ void handoff_inject_kernel(unsigned* word) { if (threadIdx.x == 0) atomicAdd(word, 1u); }
extern "C" int ipoke_flow_test_inject_timeout(ipoke_flow* f, int which, void* stream) {
  IPK_REQUIRE(f && (which == 0 || which == 1), "bad arguments");
  int rc = ensure_device(f); if (rc) return rc;
  unsigned* w = reinterpret_cast<unsigned*>(which == 0 ? f->d_xchg : f->d_cxchg);
  IPK_REQUIRE(w != nullptr, "this flow has no such scratch (unit_split = 1 / coupling fusion off)");
  hipLaunchKernelGGL(handoff_inject_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), w);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
/* Synchronising query: waits for the last pass of this flow and writes the hand-off time-out counts since the scratches were last
 * (re-)initialised: out[0] row-split MaCowUnit launches, out[1] fused conv3 + coupling launches.  Returns IPOKE_OK; non-zero counts
 * mean that a pass finished on garbage (the next entry point fails with IPOKE_ERR_STATE and re-initialises the scratches). */
extern "C" int ipoke_flow_handoff_timeouts(ipoke_flow* f, uint32_t* out) {
  IPK_REQUIRE(f && out, "null argument");
  out[0] = out[1] = 0;
  IPK_HIP(hipDeviceSynchronize());        // (also covers passes replayed from a caller's own graph, which carry no poll)
  if (f->d_xchg) IPK_HIP(hipMemcpy(&out[0], f->d_xchg, sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (f->d_cxchg) IPK_HIP(hipMemcpy(&out[1], f->d_cxchg, sizeof(uint32_t), hipMemcpyDeviceToHost));
  return IPOKE_OK;
}

static int run_forward(ipoke_flow* f, const float* params, const int32_t* perm, const void* shadow, const float* x_nchw,
                       const float* cond_nchw, int B, float* out_nchw, float* logdet, void* workspace, int save, int init,
                       float* params_mut, hipStream_t stream_h) {
  void* stream = reinterpret_cast<void*>(stream_h);
  int rc = common_checks(f, B); if (rc) return rc;
  rc = ensure_device(f); if (rc) return rc;
  IPK_REQUIRE(params && perm && x_nchw && out_nchw && workspace, "null argument");
  IPK_REQUIRE(init || (shadow && cond_nchw), "shadow weights and cond are required");
  Ctx c{f, B, (int64_t)B * f->P, f->cfg.z_channels, f->cfg.dtype, reinterpret_cast<hipStream_t>(stream),
        params, perm, reinterpret_cast<const unsigned char*>(shadow), reinterpret_cast<unsigned char*>(workspace),
        make_plan(*f, B, save ? 1 : 0)};
  c.Bfull = B;
  const int z = f->cfg.z_channels;
  rc = ipoke_nchw_to_state(x_nchw, c.state(0), B, z, f->P, c.ld, stream); if (rc) return rc;
  if (!init) {
    rc = ipoke_cond_prepare(cond_nchw, c.cond(), B, f->cfg.cond_channels, f->P, IPOKE_ACT_ELU, c.dtype, stream);
    if (rc) return rc;
  }
  IPK_HIP(hipMemsetAsync(c.at<void>(c.plan.slots), 0, (size_t)f->nslots * B * 4 * sizeof(float), c.s));
  rc = lu_prepare_all(c); if (rc) return rc;
  // the data-dependent ActNorm init needs whole-batch statistics: one lane
  std::vector<Ctx> lanes;
  rc = make_lanes(c, init ? 1 : f->n_lanes, lanes); if (rc) return rc;
  rc = fork_lanes(f, c.s, lanes); if (rc) return rc;
  static const int rpb = getenv("IPOKE_MCF_ROWS") ? atoi(getenv("IPOKE_MCF_ROWS")) : 16;
  int cur = 0;
  for (size_t i = 0; i < f->ops.size(); ++i) {
    const Op& op = f->ops[i];
    if (init && op.type != OP_ACTNORM && op.type != OP_LU) continue;   // zero-initialised couplings are the identity (macow_utils.py:231-250)
    if (!init && op.unit_head) {                   // whole MaCowUnit (ops i .. i+5) in one launch
      const int nxt = save ? (int)i + 6 : (cur ^ 1);
      static const int lidx[4] = {0, 1, 3, 4};     // the four masked convs among the six ops
      static const int oidx[4] = {1, 3, 4, 6};     // state index (relative to i) each of them writes when states are saved
      for (const Ctx& l : lanes) {
        ipoke_mcf_desc d4[4];
        for (int k = 0; k < 4; ++k) {
          const Op& mk = f->ops[i + lidx[k]];
          mcf_desc(l, mk, d4[k]);
          d4[k].x = k == 0 ? l.state(cur) : nullptr;
          d4[k].y = k == 3 ? l.state(nxt) : (save ? l.state((int)i + oidx[k]) : nullptr);
          d4[k].logdet_slot = l.slot(mk.slot);
          d4[k].rows_per_block = 16;               // slot width 4, as the per-layer launches
          if (mk.fuse_act >= 0) { d4[k].post_log_scale = params + f->ops[mk.fuse_act].p_ls; d4[k].post_bias = params + f->ops[mk.fuse_act].p_bias; }
          if (save) { d4[k].a2_save = l.rows(mk.ws_a, (int64_t)mk.K2p * f->esz); d4[k].scale_save = l.rowsf(mk.ws_b, mk.C); }
        }
        if (unit_feeds(f, i + 6)) {
          const Op& nx = f->ops[i + 6];
          d4[3].zc_out = save ? l.rows(nx.ws_g, (int64_t)nx.Kc1 * f->esz) : l.rows(l.plan.tmp_zc, 64L * f->esz);
          d4[3].zc_off = nx.z_off; d4[3].zc_stride = nx.z_stride; d4[3].zc_cin = nx.cin; d4[3].zc_ld = nx.Kc1;
        }
        if (f->unit_split > 1 && f->d_xchg) { d4[0].split = f->unit_split; d4[0].xchg = f->d_xchg; }
        rc = ipoke_macow_unit_fwd(d4, l.dtype, l.stream()); if (rc) return rc;
      }
      cur = nxt;
      i += 5;
      continue;
    }
    if (!init && op.fused) continue;               // done by the preceding MCF launch, which wrote this op's output state
    const bool fuse = !init && op.type == OP_MCF && op.fuse_act >= 0;
    const bool with_an = !init && op.type == OP_NICE && op.an_next >= 0;       // the ActNorm behind this coupling runs in its affine launch
    const int nxt = save ? (int)i + (fuse || with_an ? 2 : 1) : (cur ^ 1);
    for (const Ctx& l : lanes) {
      const float* in = l.state(cur); float* out = l.state(nxt);
      if (op.type == OP_LU) {
        rc = ipoke_lu_apply(in, out, (int64_t)l.B * l.f->P, l.ld, op.C, lu_mat(l, op, 0), 0, l.stream());
      } else if (op.type == OP_ACTNORM) {
        const float* ls = op.p_ls >= 0 ? params + op.p_ls : nullptr;
        const float* bs = op.p_bias >= 0 ? params + op.p_bias : nullptr;
        const int32_t* idx = op.idx_fwd >= 0 ? perm + op.idx_fwd : nullptr;
        if (init && op.p_ls >= 0) {
          rc = ipoke_actnorm_init(in, (int)l.M, l.ld, op.c0, op.Cn, params_mut + op.p_ls, params_mut + op.p_bias, l.stream());
          if (rc) return rc;
        }
        rc = ipoke_actnorm_fwd(in, out, (int)l.M, l.ld, op.c0, op.Cn, ls, bs, idx, l.stream());
      } else if (op.type == OP_MCF) {
        ipoke_mcf_desc d; mcf_desc(l, op, d);
        d.x = in; d.y = out;
        d.logdet_slot = l.slot(op.slot);
        d.rows_per_block = rpb;   // 16: 4 slices per sample -> slot width 4
        if (fuse) { d.post_log_scale = params + f->ops[op.fuse_act].p_ls; d.post_bias = params + f->ops[op.fuse_act].p_bias; }
        if (save) { d.a2_save = l.rows(op.ws_a, (int64_t)op.K2p * f->esz); d.scale_save = l.rowsf(op.ws_b, op.C); }
        rc = ipoke_mcf_fwd(&d, l.dtype, l.stream());
      } else {
        const int64_t hb = (int64_t)f->cfg.hidden * f->esz;
        void* h1 = l.rows(save ? op.ws_a : l.plan.tmp_h1, hb);
        void* h2 = l.rows(save ? op.ws_b : l.plan.tmp_h2, (int64_t)op.hidK * f->esz);
        void* zc = save ? l.rows(op.ws_g, (int64_t)op.Kc1 * f->esz) : l.rows(l.plan.tmp_zc, 64L * f->esz);
        const bool have_zc = (i > 0 && nice_feeds(f->ops[i - 1], op)) || (!init && unit_feeds(f, i));
        const bool one_launch = !init && nice_fused(l, op);
        rc = nice_net(l, op, in, h1, h2, zc, have_zc, !one_launch); if (rc) return rc;
        ipoke_affine_desc a; nice_affine_desc(l, op, a);
        void* ext = nullptr; int ext_ld = 0;
        if (i + 1 < f->ops.size() && nice_feeds(op, f->ops[i + 1])) {
          const Op& nx = f->ops[i + 1];
          ext = save ? l.rows(nx.ws_g, (int64_t)nx.Kc1 * f->esz) : l.rows(l.plan.tmp_zc, 64L * f->esz);
          ext_ld = nx.Kc1;
        }
        if (one_launch) {
          ipoke_conv_desc d3; nice_conv3_desc(l, op, h2, d3);
          ipoke_coupling_epi e; std::memset(&e, 0, sizeof(e));
          e.mode = with_an ? 1 : 0; e.in = in; e.scale_out = save ? l.rowsf(op.ws_c, op.cout) : nullptr;
          e.logdet_slot = l.slot(op.slot); e.slot_stride = 4; e.xchg = f->d_cxchg;
          if (with_an) {
            const Op& an = f->ops[op.an_next];
            e.out = save ? l.state((int)i + 1) : nullptr; e.out2 = out;
            e.an_c0 = an.c0; e.an_C = an.Cn; e.an_log_scale = an.p_ls >= 0 ? params + an.p_ls : nullptr;
            e.an_bias = an.p_bias >= 0 ? params + an.p_bias : nullptr; e.an_idx = an.idx_fwd >= 0 ? perm + an.idx_fwd : nullptr;
          } else {
            e.out = out; e.ext = ext; e.ext_ld = ext_ld;
          }
          rc = ipoke_conv3x3_coupling(&d3, &a, &e, l.B, l.dtype, l.stream());
        } else if (with_an) {
          const Op& an = f->ops[op.an_next];
          rc = ipoke_affine_actnorm_fwd(&a, in, save ? l.state((int)i + 1) : nullptr, out, save ? l.rowsf(op.ws_c, op.cout) : nullptr,
                                        l.slot(op.slot), 4, l.B, an.c0, an.Cn, an.p_ls >= 0 ? params + an.p_ls : nullptr,
                                        an.p_bias >= 0 ? params + an.p_bias : nullptr, an.idx_fwd >= 0 ? perm + an.idx_fwd : nullptr,
                                        l.stream());
        } else {
          rc = ipoke_affine_fwd_ext(&a, in, out, save ? l.rowsf(op.ws_c, op.cout) : nullptr, l.slot(op.slot), 4, l.B, ext, ext_ld, l.dtype,
                                    l.stream());
        }
      }
      if (rc) return rc;
    }
    cur = nxt;
    if (with_an) ++i;                              // the ActNorm op has been executed
  }
  rc = join_lanes(f, lanes, c.s); if (rc) return rc;
  rc = ipoke_state_to_nchw(c.state(cur), out_nchw, B, z, f->P, c.ld, stream); if (rc) return rc;
  if (logdet) {
    rc = ipoke_actnorm_logdet(params, f->d_lsrefs, (int)f->lsrefs.size(), f->P, c.at<float>(c.plan.ls_const), stream);
    if (rc) return rc;
    rc = ipoke_logdet_finalize(c.at<float>(c.plan.slots), init ? 0 : f->nslots, B, 4, 0.f, c.at<float>(c.plan.ls_const), logdet, stream);
    if (rc) return rc;
  }
  return IPOKE_OK;
}

extern "C" int ipoke_flow_forward(ipoke_flow* f, const float* params, const int32_t* perm, const void* shadow,
                                  const float* x_nchw, const float* cond_nchw, int B, float* out_nchw, float* logdet,
                                  void* workspace, int save_for_backward, void* stream) {
  IPK_REQUIRE(f != nullptr, "null flow handle");
  std::vector<uintptr_t> key = {1, (uintptr_t)params, (uintptr_t)perm, (uintptr_t)shadow, (uintptr_t)x_nchw, (uintptr_t)cond_nchw,
                                (uintptr_t)B, (uintptr_t)out_nchw, (uintptr_t)logdet, (uintptr_t)workspace, (uintptr_t)save_for_backward};
  int rc = pass_begin(f, reinterpret_cast<hipStream_t>(stream)); if (rc) return rc;
  rc = with_graph(f, std::move(key), reinterpret_cast<hipStream_t>(stream), [&](hipStream_t s) {
    return run_forward(f, params, perm, shadow, x_nchw, cond_nchw, B, out_nchw, logdet, workspace, save_for_backward, 0, nullptr, s);
  });
  if (rc == IPOKE_OK) { f->last_fwd_B = B; f->have_saved = save_for_backward != 0; rc = pass_end(f, reinterpret_cast<hipStream_t>(stream)); }
  return rc;
}

extern "C" int ipoke_flow_init_forward(ipoke_flow* f, float* params, const int32_t* perm, const float* x_nchw, int B,
                                       float* out_nchw, float* logdet, void* workspace, void* stream) {
  IPK_REQUIRE(f && params, "null argument");
  // weight-norm layers with zero_init: g <- 0/(std+1e-6) = 0, bias <- -mean*0 = 0 (macow_utils.py:231-250)
  for (const Op& op : f->ops) {
    if (op.type == OP_ACTNORM || op.type == OP_LU) continue;
    const int n = op.type == OP_MCF ? 2 * op.C : 2 * op.cout;
    IPK_HIP(hipMemsetAsync(params + op.p_g, 0, n * sizeof(float), reinterpret_cast<hipStream_t>(stream)));
    IPK_HIP(hipMemsetAsync(params + op.p_b, 0, n * sizeof(float), reinterpret_cast<hipStream_t>(stream)));
  }
  f->have_saved = false;
  return run_forward(f, params, perm, nullptr, x_nchw, nullptr, B, out_nchw, logdet, workspace, 0, 1, params, reinterpret_cast<hipStream_t>(stream));
}

static int run_reverse(ipoke_flow* f, const float* params, const int32_t* perm, const void* shadow,
                       const float* z_nchw, const float* cond_nchw, int B, float* x_nchw, void* workspace, hipStream_t stream_h) {
  void* stream = reinterpret_cast<void*>(stream_h);
  int rc = common_checks(f, B); if (rc) return rc;
  rc = ensure_device(f); if (rc) return rc;
  IPK_REQUIRE(params && perm && shadow && z_nchw && cond_nchw && x_nchw && workspace, "null argument");
  Ctx c{f, B, (int64_t)B * f->P, f->cfg.z_channels, f->cfg.dtype, reinterpret_cast<hipStream_t>(stream),
        params, perm, reinterpret_cast<const unsigned char*>(shadow), reinterpret_cast<unsigned char*>(workspace),
        make_plan(*f, B, 0)};
  c.Bfull = B;
  const int z = f->cfg.z_channels;
  rc = ipoke_nchw_to_state(z_nchw, c.state(0), B, z, f->P, c.ld, stream); if (rc) return rc;
  rc = ipoke_cond_prepare(cond_nchw, c.cond(), B, f->cfg.cond_channels, f->P, IPOKE_ACT_ELU, c.dtype, stream);
  if (rc) return rc;
  rc = lu_prepare_all(c); if (rc) return rc;
  std::vector<Ctx> lanes;
  rc = make_lanes(c, f->n_lanes, lanes); if (rc) return rc;
  rc = fork_lanes(f, c.s, lanes); if (rc) return rc;
  const int64_t hb = (int64_t)f->cfg.hidden * f->esz;
  static const bool no_an_ext = getenv("IPOKE_NO_AN_EXT") != nullptr;       // developer A/B
  // a stand-alone ActNorm (+ Shuffle) whose inverse is directly followed by the inverse of a coupling (ops[k - 1] in this direction)
  auto actnorm_feeds = [&](int k) {
    if (no_an_ext || k < 1 || k >= (int)f->ops.size()) return false;
    const Op& an = f->ops[k]; const Op& nx = f->ops[k - 1];
    return an.type == OP_ACTNORM && an.unit_of < 0 && nx.type == OP_NICE && nx.Kc1 - nx.cin <= c.ld && nx.Kc1 <= 64;
  };
  int cur = 0;
  for (int i = (int)f->ops.size() - 1; i >= 0; --i) {
    const Op& op = f->ops[i];
    if (op.unit_of >= 0 && i == op.unit_of + 5) {     // whole MaCowUnit (ops h .. h+5) inverted by one launch
      const int h = op.unit_of;
      static const int lidx[4] = {0, 1, 3, 4};
      for (const Ctx& l : lanes) {
        ipoke_mcf_desc d4[4];
        for (int k = 0; k < 4; ++k) {
          const Op& mk = f->ops[h + lidx[k]];
          mcf_desc(l, mk, d4[k]);
          if (mk.fuse_act >= 0) { d4[k].post_log_scale = params + f->ops[mk.fuse_act].p_ls; d4[k].post_bias = params + f->ops[mk.fuse_act].p_bias; }
        }
        d4[3].x = l.state(cur); d4[0].y = l.state(cur ^ 1);
        d4[0].x = d4[3].x; d4[3].y = d4[0].y;          // (the shared validator wants input / output on the first / last layer)
        rc = ipoke_macow_unit_inv(d4, l.dtype, l.stream()); if (rc) return rc;
      }
      cur ^= 1;
      i = h;
      continue;
    }
    for (const Ctx& l : lanes) {
      const float* in = l.state(cur); float* out = l.state(cur ^ 1);
      if (op.type == OP_LU) {
        rc = ipoke_lu_apply(in, out, (int64_t)l.B * l.f->P, l.ld, op.C, lu_mat(l, op, 1), 0, l.stream());
      } else if (op.type == OP_ACTNORM) {
        // followed (in this direction) by a coupling: its conditioning operand is written here instead of by an extract_cols launch
        const Op* nx = actnorm_feeds(i) ? &f->ops[i - 1] : nullptr;
        rc = ipoke_actnorm_inv_ext(in, out, (int)l.M, l.ld, op.c0, op.Cn, op.p_ls >= 0 ? params + op.p_ls : nullptr,
                                   op.p_bias >= 0 ? params + op.p_bias : nullptr, op.idx_bwd >= 0 ? perm + op.idx_bwd : nullptr,
                                   nx ? l.rows(l.plan.tmp_zc, 64L * f->esz) : nullptr, nx ? nx->Kc1 : 0, nx ? nx->z_off : 0,
                                   nx ? nx->z_stride : 1, nx ? nx->cin : 0, l.dtype, l.stream());
      } else if (op.type == OP_MCF) {
        ipoke_mcf_desc d; mcf_desc(l, op, d);
        d.x = in; d.y = out;
        rc = ipoke_mcf_inv(&d, l.dtype, l.stream());
      } else {
        // the conditioning channels are untouched by the coupling, so the net sees the same input as in forward
        const bool have_zc = i + 1 < (int)f->ops.size() && (nice_feeds(f->ops[i + 1], op) || actnorm_feeds(i + 1));     // written by the layer inverted just before this one
        const bool one_launch = nice_fused(l, op);
        void* h2 = l.rows(l.plan.tmp_h2, (int64_t)op.hidK * f->esz);
        rc = nice_net(l, op, in, l.rows(l.plan.tmp_h1, hb), h2, l.rows(l.plan.tmp_zc, 64L * f->esz), have_zc, !one_launch);
        if (rc) return rc;
        ipoke_affine_desc a; nice_affine_desc(l, op, a);
        const bool feed = i > 0 && nice_feeds(op, f->ops[i - 1]);
        void* ext = feed ? l.rows(l.plan.tmp_zc, 64L * f->esz) : nullptr;
        const int ext_ld = feed ? f->ops[i - 1].Kc1 : 0;
        if (one_launch) {
          ipoke_conv_desc d3; nice_conv3_desc(l, op, h2, d3);
          ipoke_coupling_epi e; std::memset(&e, 0, sizeof(e));
          e.mode = 2; e.in = in; e.out = out; e.ext = ext; e.ext_ld = ext_ld; e.xchg = f->d_cxchg;
          rc = ipoke_conv3x3_coupling(&d3, &a, &e, l.B, l.dtype, l.stream());
        } else {
          rc = ipoke_affine_inv_ext(&a, in, out, l.B, ext, ext_ld, l.dtype, l.stream());
        }
      }
      if (rc) return rc;
    }
    cur ^= 1;
  }
  rc = join_lanes(f, lanes, c.s); if (rc) return rc;
  return ipoke_state_to_nchw(c.state(cur), x_nchw, B, z, f->P, c.ld, stream);
}

extern "C" int ipoke_flow_reverse(ipoke_flow* f, const float* params, const int32_t* perm, const void* shadow,
                                  const float* z_nchw, const float* cond_nchw, int B, float* x_nchw, void* workspace,
                                  void* stream) {
  IPK_REQUIRE(f != nullptr, "null flow handle");
  std::vector<uintptr_t> key = {2, (uintptr_t)params, (uintptr_t)perm, (uintptr_t)shadow, (uintptr_t)z_nchw, (uintptr_t)cond_nchw,
                                (uintptr_t)B, (uintptr_t)x_nchw, (uintptr_t)workspace};
  int rc = pass_begin(f, reinterpret_cast<hipStream_t>(stream)); if (rc) return rc;
  rc = with_graph(f, std::move(key), reinterpret_cast<hipStream_t>(stream), [&](hipStream_t s) {
    return run_reverse(f, params, perm, shadow, z_nchw, cond_nchw, B, x_nchw, workspace, s);
  });
  return rc ? rc : pass_end(f, reinterpret_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------------
static int run_backward(ipoke_flow* f, const float* params, const int32_t* perm, const void* shadow,
                        const float* d_out_nchw, const float* d_logdet, int B, float* grads, float* dx_nchw,
                        void* workspace, hipStream_t stream_h, int npieces = 1, hipStream_t ready_stream = nullptr,
                        ipoke_grad_ready_fn ready = nullptr, void* user = nullptr) {
  void* stream = reinterpret_cast<void*>(stream_h);
  int rc = common_checks(f, B); if (rc) return rc;
  rc = ensure_device(f); if (rc) return rc;
  IPK_REQUIRE(params && perm && shadow && d_out_nchw && d_logdet && grads && workspace, "null argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Ctx c{f, B, (int64_t)B * f->P, f->cfg.z_channels, f->cfg.dtype, s, params, perm,
        reinterpret_cast<const unsigned char*>(shadow), reinterpret_cast<unsigned char*>(workspace), make_plan(*f, B, 1)};
  c.Bfull = B;
  const int z = f->cfg.z_channels, hid = f->cfg.hidden;
  const int64_t hb = (int64_t)hid * f->esz;
  rc = ipoke_nchw_to_state(d_out_nchw, c.at<float>(c.plan.g0), B, z, f->P, c.ld, stream); if (rc) return rc;
  IPK_HIP(hipMemcpyAsync(c.dld(), d_logdet, B * sizeof(float), hipMemcpyDeviceToDevice, s));
  std::vector<Ctx> lanes;
  rc = make_lanes(c, f->n_lanes, lanes); if (rc) return rc;
  rc = fork_lanes(f, s, lanes); if (rc) return rc;
  // Weight gradients reduce over the whole batch: they run on the side stream (or on lane 0 when it is disabled)
  // after every lane has produced the operands of that layer.
  hipStream_t ws_stream = f->use_side ? f->side : s;
  void* wstream = reinterpret_cast<void*>(ws_stream);
  // IPOKE_SIDE_DELAY_US (developer A/B): a one-wave spin of that many microseconds at the head of every side-stream batch, so that
  // the chain's next kernel (a fused MaCowUnit needs whole CUs: 8 waves x 240 registers) takes its CUs before the batch's
  // hundreds of weight-gradient workgroups are dealt onto every CU
  static const int side_delay = getenv("IPOKE_SIDE_DELAY_US") ? atoi(getenv("IPOKE_SIDE_DELAY_US")) : 0;
  auto wgrads_wait_lanes = [&]() -> int {
    int r0 = join_lanes(f, lanes, ws_stream); if (r0) return r0;
    if (side_delay > 0 && f->use_side) return ipoke_spin_delay(side_delay, wstream);
    return IPOKE_OK;
  };
  // Every parameter gradient is written exactly once per backward (no accumulation, no memset).  The weight gradients
  // of the MCF layers are tiny GEMMs (a handful of output tiles): they are deferred and issued as batched launches,
  // one per run of same-shape layers (= one level), using the per-layer saved operands that stay in the workspace.
  rc = ensure_tables(f, B, c.plan); if (rc) return rc;
  int pend_lo = -1, pend_hi = -1, pend_op = -1;      // pending MCF index range [lo, hi] and a representative op
  auto flush_mcf = [&]() -> int {
    if (pend_lo < 0) return IPOKE_OK;
    const Op& op = f->ops[pend_op];
    const int nb = pend_hi - pend_lo + 1;
    int r = wgrads_wait_lanes(); if (r) return r;
    static const int mcf_cap = getenv("IPOKE_TN_MAX_WGS") ? atoi(getenv("IPOKE_TN_MAX_WGS")) : 0;
    ipoke_wgrad_desc w; std::memset(&w, 0, sizeof(w));
    w.NB = B; w.Di = 1; w.Hi = 8; w.Wi = 8; w.Do = 1; w.Ho = 8; w.Wo = 8; w.kd = w.kh = w.kw = 1; w.sd = w.sh = w.sw = 1;
    w.max_workgroups = f->use_side ? mcf_cap : 0;
    const int K2 = op.H + f->cfg.cond_channels;
    w.a_f32 = 0; w.a_sn = 64L * op.K2p; w.a_sh = 8L * op.K2p; w.a_sw = op.K2p; w.a_sc = 1; w.Kc_real = K2; w.Kc = op.K2p;
    w.ldy = op.K3p; w.Nout = 2 * op.C; w.w_sn = K2; w.w_sc = 1; w.w_st = 0;
    r = ipoke_conv_wgrad_batched(&w, reinterpret_cast<const unsigned char*>(f->d_w2tab) + (size_t)pend_lo * ipoke_wgrad_batch_entry_size(),
                                 nb, c.ws, c.ws, grads, c.dtype, wstream);
    if (r) return r;
    std::memset(&w, 0, sizeof(w));
    w.NB = B; w.Di = 1; w.Hi = 8; w.Wi = 8; w.Do = 1; w.Ho = 8; w.Wo = 8; w.kd = 1; w.kh = 2; w.kw = 3; w.sd = w.sh = w.sw = 1;
    w.max_workgroups = f->use_side ? mcf_cap : 0;
    if (f->mcf_xop) {   // dtype copy [M][Cp] of every layer input, zero padded: the LDS-DMA weight-gradient GEMM
      w.a_f32 = 0; w.a_sn = 64L * op.Cp; w.a_sh = 8L * op.Cp; w.a_sw = op.Cp; w.a_sc = 1; w.Kc_real = op.Cp; w.Kc = op.Cp; w.Kc_store = op.C;
    } else {
      w.a_f32 = 1; w.a_sn = 64L * c.ld; w.a_sh = 8L * c.ld; w.a_sw = c.ld; w.a_sc = 1; w.Kc_real = op.C; w.Kc = op.Cp;
    }
    w.ldy = op.Hq; w.Nout = op.H; w.w_sn = (int64_t)op.C * 6; w.w_sc = 6; w.w_st = 1;
    r = ipoke_conv_wgrad_batched(&w, reinterpret_cast<const unsigned char*>(f->d_w1tab) + (size_t)pend_lo * ipoke_wgrad_batch_entry_size(),
                                 nb, c.ws, c.ws, grads, c.dtype, wstream);
    pend_lo = pend_hi = pend_op = -1;
    return r;
  };
  if (f->use_side) {   // the side stream joins after everything already queued on the main stream
    hipEvent_t e = next_event(f);
    IPK_HIP(hipEventRecord(e, s)); IPK_HIP(hipStreamWaitEvent(f->side, e, 0));
  }
  // pieces: groups of consecutive units (steps / priors), last unit first, of roughly equal parameter counts
  hipStream_t rs = ready_stream ? ready_stream : s;
  const std::vector<std::pair<int, int>> pieces = backward_pieces(f, npieces);            // (lowest unit, highest unit)
  // IPOKE_WGRAD_ADAM=1 (opt-in; =2: the gradient written as well): Adam-amsgrad of conv2 (plain 1x1, 73 % of the parameters) in the
  // epilogue of its weight-gradient GEMM (ipoke_wgrad_desc.adam) instead of adam_cast over the written gradient -- single-process
  // training with the engine-issued optimizer only (a data-parallel run must exchange the gradient first).  Bit-identical
  // parameters and moments (tests/test_full_gpu.py), 8 of 42 bytes per parameter less -- and MEASURED SLOWER on c2 (round 6, one call):
  // 53.2 / 53.2 ms against 49.9 / 49.8.  The launch takes 155 us instead of 53 (the same 2.8 TB/s the stand-alone kernel streams at
  // beside the chain, but on 512 workgroups that hold every CU instead of a 160-workgroup grid), the weight-gradient queue grows from
  // 11.0 to 15.0 ms per half step and the chain's kernels from 22.2 to 24.5 ms; the saved gradient round trip was served by the
  // memory-side cache anyway.  Off by default.
  const int wgrad_adam_mode = getenv("IPOKE_WGRAD_ADAM") ? atoi(getenv("IPOKE_WGRAD_ADAM")) : 0;
  const bool wgrad_adam = wgrad_adam_mode != 0 && f->nadam.on && !getenv("IPOKE_PROBE_SKIP_ADAM") && f->c2_straight &&
                          c.dtype == IPOKE_BF16 && f->use_side && lanes.size() == 1 && hid % 128 == 0 && (int)f->ajobs.size() == f->n_nice && !f->cfg.condition_nice;
  std::function<int()> flush_nice_fn = []() { return (int)IPOKE_OK; };
  auto native_adam_range = [&](int64_t b0, int64_t b1, int max_blocks) -> int {
    const ipoke_flow::NativeAdam& A = f->nadam;
    float* pm = const_cast<float*>(params);
    void* rstream = reinterpret_cast<void*>(rs);
    if (f->c2_straight)      // conv2 tensors: update + their one operand in the same linear pass; the rest: update, then relayout
      return adam_range(f, pm, grads, A.m, A.v, A.vmax, const_cast<void*>(shadow), b0, b1, A.lr, A.beta1, A.beta2, A.eps, A.wd, A.step,
                        A.grad_scale, max_blocks, rstream, wgrad_adam);
    int r = ipoke_adam_amsgrad_step_grid(pm + b0, grads + b0, A.m + b0, A.v + b0, A.vmax + b0, b1 - b0, A.lr, A.beta1, A.beta2, A.eps, A.wd, A.step,
                                         A.grad_scale, max_blocks, rstream);
    if (r) return r;
    return prepare_range(f, params, const_cast<void*>(shadow), b0, b1, false, rstream);
  };
  auto finish_piece = [&](int lvl_lo, int lvl_hi, int piece) -> int {
    int r = flush_nice_fn(); if (r) return r;
    r = flush_mcf(); if (r) return r;
    r = join_lanes(f, lanes, rs); if (r) return r;                  // the chain up to here ...
    if (f->use_side && f->side != rs) {                              // ... and the weight gradients of this piece
      hipEvent_t e = next_event(f);
      IPK_HIP(hipEventRecord(e, f->side)); IPK_HIP(hipStreamWaitEvent(rs, e, 0));
    }
    void* rstream = reinterpret_cast<void*>(rs);
    // bias / ActNorm parameter gradients: multi-tensor reduction over the per-sample partial sums of the piece's layers
    const int r0 = f->red_first[f->units[lvl_lo].op_lo], r1 = f->red_first[f->units[lvl_hi].op_hi];
    if (r1 > r0) {
      r = ipoke_reduce_rows_multi(reinterpret_cast<const float*>(c.ws), grads,
                                  reinterpret_cast<const unsigned char*>(f->d_redtab) + (size_t)r0 * ipoke_reduce_entry_size(), r1 - r0, B,
                                  rstream);
      if (r) return r;
    }
    // layers.* and priors.* are separate flat regions: the units of one kind inside the piece form one contiguous range each
    // (use1x1: the LU convs' parameters form a third region, flow.shuffle_layers.*, without weight-norm rows)
    int64_t p0[3] = {-1, -1, -1}, p1[3] = {-1, -1, -1};
    int wj0[3] = {0, 0, 0}, wj1[3] = {0, 0, 0}, rw0[3] = {0, 0, 0}, rw1[3] = {0, 0, 0};
    for (int u = lvl_lo; u <= lvl_hi; ++u) {
      const auto& un = f->units[u];
      const int k = un.kind;
      if (p0[k] < 0) { p0[k] = un.p_lo; wj0[k] = un.wj_lo; rw0[k] = un.row_lo; }
      p1[k] = un.p_hi; wj1[k] = un.wj_hi; rw1[k] = un.row_hi;
      if (un.p2_lo >= 0) { if (p0[2] < 0) p0[2] = un.p2_lo; p1[2] = un.p2_hi; }
    }
    for (int kind = 0; kind < 2; ++kind) {
      if (p0[kind] < 0) continue;
      r = ipoke_wn_bwd_multi_range(params, grads, c.wn_inv(), f->d_wjobs, wj0[kind], wj1[kind] - wj0[kind], rw0[kind],
                                   rw1[kind] - rw0[kind], rstream);
      if (r) return r;
    }
    static const bool probe_skip_adam = getenv("IPOKE_PROBE_SKIP_ADAM") != nullptr;      // developer probe: what the optimizer costs the step (parameters stay put)
    // (Moving the first k pieces' update + refresh behind the last piece's, into the step-boundary hole, was measured in round 6 and
    // removed: 50.9 / 52.3 / 54.1 ms at k = 4 / 8 / 12 against 49.5 -- profiles/r06_ab_lines.txt.)
    if (f->nadam.on && !probe_skip_adam) {
      // single-GPU training: the update of the piece's ranges and the refresh of their shadows, natively, on the ready stream
      ipoke_flow::NativeAdam A = f->nadam;
      {   // developer A/B (IPOKE_ADAM_EARLY_BLOCKS=n, IPOKE_ADAM_LATE_PIECES=k): the optimizer of all but the last k pieces on a smaller grid --
          // a lower, steadier HBM rate beside the chain; the last pieces (whose parameters the next forward needs first) at full rate
        static const int early = getenv("IPOKE_ADAM_EARLY_BLOCKS") ? atoi(getenv("IPOKE_ADAM_EARLY_BLOCKS")) : 0;
        static const int late = getenv("IPOKE_ADAM_LATE_PIECES") ? atoi(getenv("IPOKE_ADAM_LATE_PIECES")) : 3;
        if (early > 0 && piece < (int)pieces.size() - late) A.max_blocks = early;
      }
      for (int kind = 0; kind < 3; ++kind) {
        if (p0[kind] < 0 || p1[kind] <= p0[kind]) continue;
        r = native_adam_range(p0[kind], p1[kind], A.max_blocks); if (r) return r;
      }
    }
    if (ready)
      for (int kind = 0; kind < 3; ++kind)
        if (p0[kind] >= 0 && p1[kind] > p0[kind]) ready(user, piece, p0[kind], p1[kind]);
    return IPOKE_OK;
  };
  // NICE weight gradients are not started right behind their coupling: the couplings come in pairs whose data-gradient
  // GEMMs want the whole chip, and a weight-gradient kernel that is already running holds its CUs for 30 us.  They are
  // queued when the chain moves on to the (latency-bound, 20..80 workgroup) masked-conv layers.
  std::vector<int> pend_nice;
  auto flush_nice = [&]() -> int {
    if (pend_nice.empty()) return IPOKE_OK;
    int rc = wgrads_wait_lanes(); if (rc) return rc;
    // cap on workgroups per weight-gradient launch (leaving CUs to the chain): measured 79.6 ms uncapped, 81.0 at 128,
    // 88.7 at 96 -- the side stream becomes the critical path, so off by default
    static const int tn_cap = getenv("IPOKE_TN_MAX_WGS") ? atoi(getenv("IPOKE_TN_MAX_WGS")) : 0;
    // The pending couplings (a pair or two pairs of one flow step: same widths) go out as three batched launches, one per
    // convolution, blockIdx.z = coupling: conv1 (48 output tiles) and conv3 (144) are latency chains of 20 reduction
    // stages that leave most of the chip idle when launched one coupling at a time.
    size_t i0 = 0;
    while (i0 < pend_nice.size()) {
      const Op& op = f->ops[pend_nice[i0]];
      size_t i1 = i0 + 1;                                   // pend_nice is in backward order: nice_idx descends by one
      while (i1 < pend_nice.size()) {
        const Op& o2 = f->ops[pend_nice[i1]];
        if (o2.nice_idx != f->ops[pend_nice[i1 - 1]].nice_idx - 1 || o2.Kc1 != op.Kc1 || o2.cin != op.cin || o2.cout != op.cout ||
            o2.Kc3 != op.Kc3) break;
        ++i1;
      }
      const int nb = (int)(i1 - i0), lo = f->ops[pend_nice[i1 - 1]].nice_idx;
      ipoke_wgrad_desc w;
      auto base8 = [&](int k, int pad) {
        std::memset(&w, 0, sizeof(w));
        w.NB = B; w.Di = 1; w.Hi = 8; w.Wi = 8; w.Do = 1; w.Ho = 8; w.Wo = 8; w.kd = 1; w.kh = w.kw = k;
        w.sd = w.sh = w.sw = 1; w.ph = w.pw = pad;
        w.max_workgroups = f->use_side ? tn_cap : 0;
      };
      auto entries = [&](int k) {
        return reinterpret_cast<const unsigned char*>(f->d_ntab[k]) + (size_t)lo * ipoke_wgrad_batch_entry_size();
      };
      // conv3 (effective weight; weight-norm backward runs at the end)
      base8(3, 1);
      w.a_sn = 64L * op.hidK; w.a_sh = 8L * op.hidK; w.a_sw = op.hidK; w.a_sc = 1; w.Kc_real = op.hidK; w.Kc = op.hidK;
      w.ldy = op.Kc3; w.Nout = 2 * op.cout;
      w.w_sn = (int64_t)op.hidK * 9; w.w_sc = 9; w.w_st = 1;
      rc = ipoke_conv_wgrad_batched(&w, entries(2), nb, c.ws, c.ws, grads, c.dtype, wstream); if (rc) return rc;
      // conv2
      base8(1, 0);
      w.a_sn = 64L * hid; w.a_sh = 8L * hid; w.a_sw = hid; w.a_sc = 1; w.Kc_real = hid; w.Kc = hid;
      w.ldy = hid; w.Nout = hid;
      w.w_sn = hid; w.w_sc = 1; w.w_st = 0;
      ipoke_wgrad_adam wa;
      if (wgrad_adam) {
        const ipoke_flow::NativeAdam& A = f->nadam;
        wa.params = const_cast<float*>(params); wa.m = A.m; wa.v = A.v; wa.vmax = A.vmax;
        wa.operand = const_cast<unsigned char*>(c.shadow + c.shadow_base());
        wa.lr = A.lr; wa.beta1 = A.beta1; wa.beta2 = A.beta2; wa.eps = A.eps; wa.weight_decay = A.wd; wa.grad_scale = A.grad_scale; wa.step = A.step;
        wa.keep_grad = wgrad_adam_mode == 2;
        w.adam = &wa;
      }
      rc = ipoke_conv_wgrad_batched(&w, entries(1), nb, c.ws, c.ws, grads, c.dtype, wstream); if (rc) return rc;
      // conv1 (input = conditioning channels of the saved state)
      base8(3, 1);
      w.a_f32 = 0; w.a_sn = 64L * op.Kc1; w.a_sh = 8L * op.Kc1; w.a_sw = op.Kc1; w.a_sc = 1;
      w.Kc_real = op.Kc1; w.Kc = op.Kc1;
      w.ldy = hid; w.Nout = hid;
      w.w_sn = (int64_t)op.cin * 9; w.w_sc = 9; w.w_st = 1; w.Kc_store = op.cin;
      rc = ipoke_conv_wgrad_batched(&w, entries(0), nb, c.ws, c.ws, grads, c.dtype, wstream); if (rc) return rc;
      i0 = i1;
    }
    pend_nice.clear();
    return IPOKE_OK;
  };
  flush_nice_fn = flush_nice;
  // Every flush costs the CHAIN an event record (the side stream must see the couplings' operands): a barrier packet with a completion
  // signal between two chain kernels, ~10 us of queue time each (profiles/r03_bench_trace_gaps.txt: 112 gaps of 12.8 us per step in
  // front of the fused units).  IPOKE_NICE_FLUSH_MIN couplings are collected before the chain pays for one (the end of a piece always
  // flushes: finish_piece).
  // round 6 (with the stationary-input conv1 / conv3 weight gradients: 32 workgroups per problem): four couplings per batch, 48.83 / 48.62
  // against 48.98 / 49.04 ms at two (one call); 3 / 6 / 8: 48.9 / 48.8 / 48.8
  static const int flush_min = getenv("IPOKE_NICE_FLUSH_MIN") ? atoi(getenv("IPOKE_NICE_FLUSH_MIN")) : 4;
  auto maybe_flush_nice = [&]() -> int { return (int)pend_nice.size() >= flush_min ? flush_nice() : (int)IPOKE_OK; };
  size_t pk = 0;
  int cur = 0;
  int pending_an = -1;              // ActNorm op whose backward is carried by the next (lower) coupling's launch
  const int64_t goff[2] = {c.plan.g0, c.plan.g1};
  for (int i = (int)f->ops.size() - 1; i >= 0; --i) {
    const Op& op = f->ops[i];
    if (op.unit_of >= 0 && i == op.unit_of + 5) {     // whole MaCowUnit (ops h .. h+5) differentiated by one launch
      const int h = op.unit_of;
      rc = maybe_flush_nice(); if (rc) return rc;
      static const int lidx[4] = {0, 1, 3, 4};
      for (const Ctx& l : lanes) {
        ipoke_mcf_desc d4[4];
        for (int k = 0; k < 4; ++k) {
          const int j = h + lidx[k];
          const Op& mk = f->ops[j];
          mcf_desc(l, mk, d4[k]);
          d4[k].x = l.state(j);
          d4[k].a2_save = l.rows(mk.ws_a, (int64_t)mk.K2p * f->esz); d4[k].scale_save = l.rowsf(mk.ws_b, mk.C);
          d4[k].dparams_save = l.rows(mk.ws_c, (int64_t)mk.K3p * f->esz); d4[k].dc_save = l.rows(mk.ws_d, (int64_t)mk.Hq * f->esz);
          if (f->mcf_xop) d4[k].x_op_save = l.rows(mk.ws_e, (int64_t)mk.Cp * f->esz);
          d4[k].dbias_part = l.dbp(j, 2 * mk.C, f->unit_split);
          if (mk.fuse_act >= 0) {
            const Op& an = f->ops[mk.fuse_act];
            d4[k].post_log_scale = params + an.p_ls; d4[k].post_bias = params + an.p_bias;
            d4[k].y_post = l.state(j + 2);
            d4[k].post_part = l.dbp(mk.fuse_act, 2 * an.Cn, f->unit_split);
          }
        }
        d4[3].dy = l.rowsf(goff[cur], l.ld); d4[0].dx = l.rowsf(goff[cur ^ 1], l.ld); d4[0].dld = l.dld();
        IPK_REQUIRE(f->unit_split == 1 || f->d_xchg, "row-split unit launches without exchange scratch (ensure_tables)");
        if (f->unit_split > 1) { d4[0].split = f->unit_split; d4[0].xchg = f->d_xchg; }
        ipoke_unit_pair_desc pq;
        if (h >= 2 && f->ops[h - 1].pair_unit == h) {     // the (coupling, ActNorm) pair in front of the unit rides along
          const Op& an = f->ops[h - 1]; const Op& cp = f->ops[h - 2];
          std::memset(&pq, 0, sizeof(pq));
          pq.an_log_scale = an.p_ls >= 0 ? params + an.p_ls : nullptr; pq.an_idx = an.idx_fwd >= 0 ? perm + an.idx_fwd : nullptr;
          pq.an_x = l.state(h - 1); pq.an_part = an.p_ls >= 0 ? l.dbp(h - 1, 2 * an.Cn, f->unit_split) : nullptr;
          pq.Cp = cp.cout; pq.t_off = cp.t_off; pq.t_stride = cp.t_stride; pq.x0 = l.state(h - 2); pq.scale = l.rowsf(cp.ws_c, cp.cout);
          pq.dparams = l.rows(cp.ws_d, (int64_t)cp.Kc3 * f->esz); pq.ldp = cp.Kc3; pq.dbias_part = l.dbp(h - 2, 2 * cp.cout, f->unit_split);
          pq.dx = d4[0].dx;
          d4[0].pair = &pq;
        }
        rc = ipoke_macow_unit_bwd(d4, l.dtype, l.stream()); if (rc) return rc;
      }
      for (int k = 3; k >= 0; --k) {                 // weight gradients: deferred, batched per run of same-width layers
        const int j = h + lidx[k];
        const Op& mk = f->ops[j];
        if (pend_lo >= 0 && (f->ops[pend_op].C != mk.C || pend_hi - pend_lo + 1 >= 256)) { rc = flush_mcf(); if (rc) return rc; }
        if (pend_lo < 0) { pend_hi = mk.mcf_idx; pend_op = j; }
        pend_lo = mk.mcf_idx;
      }
      cur ^= 1;
      i = h;                                         // the loop decrement moves on to op h - 1
      continue;
    }
    if (op.fused) {                                  // differentiated inside the preceding MCF layer's backward kernel
      if (i == f->units[pieces[pk].first].op_lo) { rc = finish_piece(pieces[pk].first, pieces[pk].second, pk); if (rc) return rc; ++pk; }
      continue;
    }
    // An ActNorm right behind a coupling is differentiated by the coupling's launch (ipoke_actnorm_affine_bwd), unless it is the
    // lowest op of the current piece: its parameter-gradient partials must exist when the piece is finished right after this op.
    if (op.type == OP_ACTNORM && op.pair_unit >= 0) {      // differentiated by the unit's launch above (so was its coupling: below)
      if (i == f->units[pieces[pk].first].op_lo) { rc = finish_piece(pieces[pk].first, pieces[pk].second, pk); if (rc) return rc; ++pk; }
      continue;
    }
    if (op.type == OP_ACTNORM && op.an_prev >= 0 && i != f->units[pieces[pk].first].op_lo) { pending_an = i; continue; }
    if (op.type != OP_NICE) { rc = maybe_flush_nice(); if (rc) return rc; }
    for (const Ctx& l : lanes) {
      const bool by_unit = op.type == OP_NICE && op.pair_unit >= 0;      // dx / dparams / partials of this coupling were written by a unit's launch
      const float* gin = l.rowsf(goff[cur], l.ld); float* gout = l.rowsf(goff[by_unit ? cur : cur ^ 1], l.ld);
      const float* xin = l.state(i);                 // saved input of op i
      if (op.type == OP_LU) {
        // dx = dy W (W^T applied per position); parameter gradients straight into the flat buffer (the forward pass of
        // this step left [W | W^-1 | wl | wu] in the workspace)
        rc = ipoke_lu_apply(gin, gout, (int64_t)l.B * l.f->P, l.ld, op.C, lu_mat(l, op, 0), 1, l.stream()); if (rc) return rc;
        rc = ipoke_lu_wgrad(gin, xin, l.B, f->P, l.ld, params, f->fbuf, l.at<float>(l.plan.lu),
                            reinterpret_cast<const unsigned char*>(f->d_lujobs) + (size_t)op.lu_idx * sizeof(LuJobH), l.dld(), grads,
                            l.stream());
        if (rc) return rc;
      } else if (op.type == OP_ACTNORM) {
        rc = ipoke_actnorm_bwd(gin, xin, gout, (int)l.M, l.ld, op.c0, op.Cn, op.p_ls >= 0 ? params + op.p_ls : nullptr,
                               op.idx_fwd >= 0 ? perm + op.idx_fwd : nullptr, l.dld(), l.B, f->P, l.dbp(i, 2 * op.Cn), l.stream());
        if (rc) return rc;
      } else if (op.type == OP_MCF) {
        ipoke_mcf_desc d; mcf_desc(l, op, d);
        d.x = xin; d.dy = gin; d.dx = gout; d.dld = l.dld();
        d.a2_save = l.rows(op.ws_a, (int64_t)op.K2p * f->esz); d.scale_save = l.rowsf(op.ws_b, op.C);
        d.dparams_save = l.rows(op.ws_c, (int64_t)op.K3p * f->esz); d.dc_save = l.rows(op.ws_d, (int64_t)op.Hq * f->esz);
        d.dbias_part = l.dbp(i, 2 * op.C);
        d.y = gout;   // unused by the backward kernel, must be non-null for the shared validator
        if (op.fuse_act >= 0) {
          const Op& an = f->ops[op.fuse_act];
          d.post_log_scale = params + an.p_ls; d.post_bias = params + an.p_bias;
          d.y_post = l.state(i + 2);                 // saved output of the fused pair
          d.post_part = l.dbp(op.fuse_act, 2 * an.Cn);
        }
        rc = ipoke_mcf_bwd(&d, l.dtype, l.stream()); if (rc) return rc;
      } else {
        const void* h1 = l.rows(op.ws_a, hb); const void* h2 = l.rows(op.ws_b, (int64_t)op.hidK * f->esz);
        void* dprm = l.rows(op.ws_d, (int64_t)op.Kc3 * f->esz); void* dp2 = l.rows(op.ws_e, hb); void* dp1 = l.rows(op.ws_f, hb);
        if (by_unit) {
          rc = IPOKE_OK;
        } else if (pending_an == i + 1) {
          const Op& an = f->ops[i + 1];
          rc = ipoke_actnorm_affine_bwd(an.c0, an.Cn, an.p_ls >= 0 ? params + an.p_ls : nullptr, an.idx_fwd >= 0 ? perm + an.idx_fwd : nullptr,
                                        gin, l.state(i + 1), an.p_ls >= 0 ? l.dbp(i + 1, 2 * an.Cn) : nullptr, op.cout, op.t_off, op.t_stride,
                                        f->P, l.ld, xin, l.rowsf(op.ws_c, op.cout), l.dld(), gout, dprm, op.Kc3, l.dbp(i, 2 * op.cout), l.B,
                                        l.dtype, l.stream());
        } else {
          rc = ipoke_affine_bwd(op.cout, op.t_off, op.t_stride, f->P, l.ld, gin, xin, l.rowsf(op.ws_c, op.cout), l.dld(), gout, dprm,
                                op.Kc3, l.dbp(i, 2 * op.cout), l.B, l.dtype, l.stream());
        }
        if (rc) return rc;
        ipoke_conv_desc d;
        // conv3 data gradient, times ELU'(h2)
        set_conv8(d, l.B, 3, 1); d.transposed = 1;
        set_a_dense(d, dprm, op.Kc3, op.Kc3);
        d.W = l.sh(op.sh_c3t); d.ldw = 9 * op.Kc3; d.Nout = hid; d.dact = h2; d.ld_dact = op.hidK; d.dact_act = IPOKE_ACT_ELU;
        d.C = dp2; d.ldc = hid;
        rc = ipoke_conv_forward(&d, l.dtype, l.stream()); if (rc) return rc;
        // conv2 data gradient, times ELU'(h1)
        set_conv8(d, l.B, 1, 0);
        set_a_dense(d, dp2, hid, hid);
        if (f->c2_straight) { d.W = l.sh(op.sh_c2); d.w_kmajor = 1; }      // W2[out][in] read K-major: no transposed copy exists
        else d.W = l.sh(op.sh_c2t);
        d.ldw = hid; d.Nout = hid; d.dact = h1; d.ld_dact = hid; d.dact_act = IPOKE_ACT_ELU;
        d.C = dp1; d.ldc = hid;
        rc = ipoke_conv_forward(&d, l.dtype, l.stream()); if (rc) return rc;
        // conv1 data gradient accumulated into the conditioning channels
        set_conv8(d, l.B, 3, 1); d.transposed = 1;
        set_a_dense(d, dp1, hid, hid);
        d.W = l.sh(op.sh_c1t); d.ldw = 9 * hid; d.Nout = op.cin; d.C = gout; d.c_f32 = 1; d.c_accumulate = 1; d.ldc = l.ld;
        d.c_coff = op.z_off; d.c_cstride = op.z_stride; d.splitk = nice_splitk(l);
        // the K slices meet in the engine's scratch and are summed in a fixed order (bit-reproducible steps); sub-batch lanes
        // (developer switch) run several of these launches at once and keep the atomic form
        if (lanes.size() == 1 && op.cin <= 64) { d.acc_scratch = f->d_acc; d.acc_scratch_bytes = f->acc_bytes; }
        rc = ipoke_conv_forward(&d, l.dtype, l.stream()); if (rc) return rc;
      }
    }
    if (op.type == OP_MCF) {
      // weight gradients: deferred, batched per run of same-width layers
      if (pend_lo >= 0 && (f->ops[pend_op].C != op.C || pend_hi - pend_lo + 1 >= 256)) { rc = flush_mcf(); if (rc) return rc; }
      if (pend_lo < 0) { pend_hi = op.mcf_idx; pend_op = i; }
      pend_lo = op.mcf_idx;
    } else if (op.type == OP_NICE) {
      pend_nice.push_back(i);      // weight gradients: launched when the chain leaves this group of couplings
      if (pending_an == i + 1) pending_an = -1;
    }
    if (!(op.type == OP_NICE && op.pair_unit >= 0)) cur ^= 1;
    if (i == f->units[pieces[pk].first].op_lo) {        // the lowest op of the current piece has been queued
      rc = finish_piece(pieces[pk].first, pieces[pk].second, pk); if (rc) return rc;
      ++pk;
    }
  }
  static const bool probe_no_join = getenv("IPOKE_PROBE_NO_READY_JOIN") != nullptr;      // developer probe (WRONG results): the caller's stream does not wait for the ready stream
  if (rs != s && !probe_no_join) {        // later work on the caller's stream (optimizer step) sees every gradient
    hipEvent_t e = next_event(f);
    IPK_HIP(hipEventRecord(e, rs)); IPK_HIP(hipStreamWaitEvent(s, e, 0));
  }
  if (dx_nchw) { rc = ipoke_state_to_nchw(c.rowsf(goff[cur], c.ld), dx_nchw, B, z, f->P, c.ld, stream); if (rc) return rc; }
  return IPOKE_OK;
}

extern "C" int ipoke_flow_backward(ipoke_flow* f, const float* params, const int32_t* perm, const void* shadow,
                                   const float* d_out_nchw, const float* d_logdet, int B, float* grads, float* dx_nchw,
                                   void* workspace, void* stream) {
  IPK_REQUIRE(f != nullptr, "null flow handle");
  if (!f->have_saved || f->last_fwd_B != B)
    return fail(IPOKE_ERR_STATE, "ipoke_flow_backward needs a preceding ipoke_flow_forward(save_for_backward=1) with the same batch");
  std::vector<uintptr_t> key = {3, (uintptr_t)params, (uintptr_t)perm, (uintptr_t)shadow, (uintptr_t)d_out_nchw, (uintptr_t)d_logdet,
                                (uintptr_t)B, (uintptr_t)grads, (uintptr_t)dx_nchw, (uintptr_t)workspace};
  int rc = pass_begin(f, reinterpret_cast<hipStream_t>(stream)); if (rc) return rc;
  rc = with_graph(f, std::move(key), reinterpret_cast<hipStream_t>(stream), [&](hipStream_t s) {
    return run_backward(f, params, perm, shadow, d_out_nchw, d_logdet, B, grads, dx_nchw, workspace, s);
  });
  return rc ? rc : pass_end(f, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int ipoke_flow_backward_pieces(ipoke_flow* f, const float* params, const int32_t* perm, const void* shadow,
                                          const float* d_out_nchw, const float* d_logdet, int B, float* grads, float* dx_nchw,
                                          void* workspace, int npieces, void* ready_stream, ipoke_grad_ready_fn ready, void* user,
                                          void* stream) {
  IPK_REQUIRE(f != nullptr, "null flow handle");
  if (!f->have_saved || f->last_fwd_B != B)
    return fail(IPOKE_ERR_STATE, "ipoke_flow_backward needs a preceding ipoke_flow_forward(save_for_backward=1) with the same batch");
  // host callbacks cannot be captured: always eager
  int rc = pass_begin(f, reinterpret_cast<hipStream_t>(stream)); if (rc) return rc;
  rc = run_backward(f, params, perm, shadow, d_out_nchw, d_logdet, B, grads, dx_nchw, workspace,
                    reinterpret_cast<hipStream_t>(stream), npieces, reinterpret_cast<hipStream_t>(ready_stream), ready, user);
  return rc ? rc : pass_end(f, reinterpret_cast<hipStream_t>(stream));
}
