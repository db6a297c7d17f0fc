// Whole-path executor: PETRHead.forward (reference models/dense_heads/petr_head.py:366-468, with
// PETRTransformer.forward petr_transformer.py:69-109 and the decoder loop :423-447 inlined) and its
// gradient, each as ONE host call that enqueues every kernel on the caller's stream.
//
// MI355X-first design notes
//  * one flat fp32 parameter buffer (and one flat gradient buffer with the same layout): the host
//    aliases its nn.Parameters onto it, the data-parallel gradient exchange is a handful of large
//    contiguous RCCL all-reduces instead of 222 small ones.  The layout is ordered by backward
//    completion time (branches, decoder layer 5..0, then everything that only becomes final at the
//    end), so stage s of the backward finalises one contiguous range = one all-reduce bucket.
//  * token-major activations [B*L, 256]: the NCHW inputs are read by the GEMM directly as an
//    M-contiguous operand, so the two permute copies of petr_transformer.py:90-91 do not exist.
//  * the key/value projections of all 6 decoder layers are ONE batched contraction each (memory and
//    positional embedding are layer-invariant; petr_transformer.py:343-344,357-362).
//  * every elementwise op of the reference (bias, residual, ReLU, key+pos, query+pos, nan_to_num,
//    sigmoid/box scaling) is an epilogue/prologue of the producing kernel.
#include <string.h>

#include "common.h"

namespace {

struct Dims {
  int B, N, Cin, H, W, HW, Q, NL, NH, C, F, D, ncls, code;
  long L, BL, BQ, R;
};

static Dims make_dims(const petr_head_config* c) {
  Dims d;
  d.B = c->B; d.N = c->N; d.Cin = c->C_in; d.H = c->H; d.W = c->W; d.HW = c->H * c->W;
  d.Q = c->num_query; d.NL = c->num_layers; d.NH = c->num_heads; d.C = c->embed_dims; d.F = c->ffn_dims;
  d.D = c->depth_num; d.ncls = c->num_classes; d.code = c->code_size;
  d.L = (long)d.N * d.HW; d.BL = d.B * d.L; d.BQ = (long)d.B * d.Q; d.R = d.NL * d.BQ;
  return d;
}

static int check_config(const petr_head_config* c) {
  PETR_CHECK(c, PETR_ERR_INVALID, "head: null config");
  PETR_CHECK(c->B > 0 && c->N > 0 && c->C_in > 0 && c->H > 0 && c->W > 0 && c->num_query > 0, PETR_ERR_INVALID,
             "head: bad shape");
  PETR_CHECK(c->embed_dims == 256 && c->num_heads == 8, PETR_ERR_UNSUPPORTED,
             "head: embed_dims must be 256 (petr_head.py:175) with 8 heads of 32");
  PETR_CHECK(c->num_layers >= 1 && c->num_layers <= 8, PETR_ERR_UNSUPPORTED, "head: 1..8 decoder layers");
  PETR_CHECK(c->ffn_dims % 32 == 0 && c->ffn_dims > 0, PETR_ERR_UNSUPPORTED, "head: ffn_dims must be a multiple of 32");
  PETR_CHECK((c->H * c->W) % 4 == 0, PETR_ERR_UNSUPPORTED, "head: H*W must be a multiple of 4");
  PETR_CHECK(c->v2 || (!c->with_fpe && !c->with_time && !c->with_multi && c->shared_branches), PETR_ERR_INVALID,
             "head: with_fpe/with_time/with_multi/deep-copied branches are PETRv2Head switches (set v2)");
  PETR_CHECK(!c->v2 || !c->shared_branches, PETR_ERR_INVALID, "head: PETRv2Head deep-copies its branches (petrv2_head.py:304-307)");
  PETR_CHECK(!c->with_multi || c->code_size == 10, PETR_ERR_UNSUPPORTED, "head: RegLayer groups (2,1,3,2,2) need code_size 10");
  PETR_CHECK(!c->with_time || c->B == 1, PETR_ERR_UNSUPPORTED,
             "head: with_time divides by a per-sample scalar; the reference broadcast only works for B=1 (petrv2_head.py:505,521)");
  PETR_CHECK(c->code_size >= 5 && c->num_classes >= 1, PETR_ERR_INVALID, "head: bad code_size/num_classes");
  return PETR_OK;
}

static long align4(long v) { return (v + 3) & ~3L; }
// parameter slots start on multiples of 8 elements: 16-byte aligned in the fp32 buffer AND in its bf16 copy
static long align8(long v) { return (v + 7) & ~7L; }

// ---------------------------------------------------------------------------------------------
// parameter layout
// ---------------------------------------------------------------------------------------------
struct LayerP {
  long sa_in_w, sa_in_b, sa_out_w, sa_out_b, ca_out_w, ca_out_b, f1_w, f1_b, f2_w, f2_b, n_g[3], n_b[3];
  long ca_in_w, ca_in_b;   // live in the final block (K/V rows are only final after the K/V-projection backward)
};
struct POff {
  long cls_w[3], cls_b[3], cls_g[2], cls_be[2], reg_w[3], reg_b[3];
  long br_stride;                       // distance between the branch blocks of two levels (0: shared, PETRHead)
  long th_w1, th_b1, th_w2, th_b2, th_stride;   // RegLayer task heads (PETRv2 with_multi), uniform slot per head
  long fpe_rw, fpe_rb, fpe_ew, fpe_eb;  // SELayer (PETRv2 with_fpe)
  long post_g, post_b;
  LayerP lay[8];
  long ca_in_stride;
  long qe_w1, qe_b1, qe_w2, qe_b2, ref, pe_w1, pe_b1, pe_w2, pe_b2, ad_w1, ad_b1, ad_w2, ad_b2, in_w, in_b, code_w;
  long total;
  int n_stages;
  long stage_begin[16], stage_end[16];
};

struct LayoutBuilder {
  petr_head_layout_t* out;   // may be null
  long cur = 0;
  int count = 0;
  long add(const char* name, int ndim, int s0, int s1 = 1, int s2 = 1, int s3 = 1, int alias_of = -1) {
    const long n = (long)s0 * s1 * s2 * s3;
    long off = cur;
    if (alias_of >= 0 && out) off = out->offset[alias_of];
    if (out && count < PETR_MAX_PARAMS) {
      snprintf(out->name[count], sizeof(out->name[count]), "%s", name);
      out->offset[count] = off;
      out->ndim[count] = ndim;
      out->shape[count][0] = s0; out->shape[count][1] = s1; out->shape[count][2] = s2; out->shape[count][3] = s3;
      out->alias_of[count] = alias_of;
    }
    ++count;
    if (alias_of < 0) cur = align8(cur + n);
    return off;
  }
};

static void build_layout(const petr_head_config* c, POff* P, petr_head_layout_t* out) {
  const Dims d = make_dims(c);
  LayoutBuilder lb;
  lb.out = out;
  char nm[128];
  int stage = 0;
  // ---- stage 0: branches + post_norm ----
  P->stage_begin[stage] = lb.cur;
  const int cls_idx[3] = {0, 3, 6}, ln_idx[2] = {1, 4}, reg_idx[3] = {0, 2, 4};
  static const int TH_DIMS[5] = {2, 1, 3, 2, 2};   // RegLayer group_reg_dims (petrv2_head.py:66)
  int first_cls[3][2], first_ln[2][2], first_reg[3][2];
  P->br_stride = 0;
  P->th_w1 = P->th_b1 = P->th_w2 = P->th_b2 = P->th_stride = 0;
  long lvl0_begin = lb.cur;
  for (int lvl = 0; lvl < d.NL; ++lvl) {
    // PETRHead: one module in all slots (petr_head.py:244-247) -> aliases; PETRv2Head: deep copies (:304-307)
    const bool alias = c->shared_branches && lvl > 0;
    const bool first = lvl == 0;
    if (lvl == 1 && !c->shared_branches) P->br_stride = lb.cur - lvl0_begin;
    for (int i = 0; i < 3; ++i) {
      const int nout = i == 2 ? d.ncls : d.C;
      snprintf(nm, sizeof nm, "cls_branches.%d.%d.weight", lvl, cls_idx[i]);
      if (first) first_cls[i][0] = lb.count;
      long o = lb.add(nm, 2, nout, d.C, 1, 1, alias ? first_cls[i][0] : -1);
      if (first) P->cls_w[i] = o;
      snprintf(nm, sizeof nm, "cls_branches.%d.%d.bias", lvl, cls_idx[i]);
      if (first) first_cls[i][1] = lb.count;
      o = lb.add(nm, 1, nout, 1, 1, 1, alias ? first_cls[i][1] : -1);
      if (first) P->cls_b[i] = o;
      if (i < 2) {
        snprintf(nm, sizeof nm, "cls_branches.%d.%d.weight", lvl, ln_idx[i]);
        if (first) first_ln[i][0] = lb.count;
        o = lb.add(nm, 1, d.C, 1, 1, 1, alias ? first_ln[i][0] : -1);
        if (first) P->cls_g[i] = o;
        snprintf(nm, sizeof nm, "cls_branches.%d.%d.bias", lvl, ln_idx[i]);
        if (first) first_ln[i][1] = lb.count;
        o = lb.add(nm, 1, d.C, 1, 1, 1, alias ? first_ln[i][1] : -1);
        if (first) P->cls_be[i] = o;
      }
    }
    if (!c->with_multi) {
      for (int i = 0; i < 3; ++i) {
        const int nout = i == 2 ? d.code : d.C;
        snprintf(nm, sizeof nm, "reg_branches.%d.%d.weight", lvl, reg_idx[i]);
        if (first) first_reg[i][0] = lb.count;
        long o = lb.add(nm, 2, nout, d.C, 1, 1, alias ? first_reg[i][0] : -1);
        if (first) P->reg_w[i] = o;
        snprintf(nm, sizeof nm, "reg_branches.%d.%d.bias", lvl, reg_idx[i]);
        if (first) first_reg[i][1] = lb.count;
        o = lb.add(nm, 1, nout, 1, 1, 1, alias ? first_reg[i][1] : -1);
        if (first) P->reg_b[i] = o;
      }
    } else {
      // RegLayer (petrv2_head.py:63-95): reg_branch = Linear,ReLU,Dropout,Linear,ReLU,Dropout ; 5 task heads
      const int sh_idx[2] = {0, 3};
      for (int i = 0; i < 2; ++i) {
        snprintf(nm, sizeof nm, "reg_branches.%d.reg_branch.%d.weight", lvl, sh_idx[i]);
        long o = lb.add(nm, 2, d.C, d.C);
        if (first) P->reg_w[i] = o;
        snprintf(nm, sizeof nm, "reg_branches.%d.reg_branch.%d.bias", lvl, sh_idx[i]);
        o = lb.add(nm, 1, d.C);
        if (first) P->reg_b[i] = o;
      }
      for (int t = 0; t < 5; ++t) {
        const long slot = lb.cur;
        snprintf(nm, sizeof nm, "reg_branches.%d.task_heads.%d.0.weight", lvl, t);
        long o = lb.add(nm, 2, d.C, d.C);
        if (first && t == 0) P->th_w1 = o;
        snprintf(nm, sizeof nm, "reg_branches.%d.task_heads.%d.0.bias", lvl, t);
        o = lb.add(nm, 1, d.C);
        if (first && t == 0) P->th_b1 = o;
        snprintf(nm, sizeof nm, "reg_branches.%d.task_heads.%d.2.weight", lvl, t);
        o = lb.add(nm, 2, TH_DIMS[t], d.C);
        if (first && t == 0) P->th_w2 = o;
        lb.cur = slot + align8((long)d.C * d.C) + align8(d.C) + align8(3L * d.C);   // pad to the widest head (3 rows)
        snprintf(nm, sizeof nm, "reg_branches.%d.task_heads.%d.2.bias", lvl, t);
        o = lb.add(nm, 1, TH_DIMS[t]);
        if (first && t == 0) P->th_b2 = o;
        if (first && t == 1) P->th_stride = slot - (P->th_w1);
      }
    }
  }
  P->post_g = lb.add("transformer.decoder.post_norm.weight", 1, d.C);
  P->post_b = lb.add("transformer.decoder.post_norm.bias", 1, d.C);
  P->stage_end[stage++] = lb.cur;
  // ---- stages 1..NL: decoder layers, last layer first ----
  for (int l = d.NL - 1; l >= 0; --l) {
    P->stage_begin[stage] = lb.cur;
    LayerP& lp = P->lay[l];
    const char* pre = "transformer.decoder.layers";
    snprintf(nm, sizeof nm, "%s.%d.attentions.0.attn.in_proj_weight", pre, l); lp.sa_in_w = lb.add(nm, 2, 3 * d.C, d.C);
    snprintf(nm, sizeof nm, "%s.%d.attentions.0.attn.in_proj_bias", pre, l); lp.sa_in_b = lb.add(nm, 1, 3 * d.C);
    snprintf(nm, sizeof nm, "%s.%d.attentions.0.attn.out_proj.weight", pre, l); lp.sa_out_w = lb.add(nm, 2, d.C, d.C);
    snprintf(nm, sizeof nm, "%s.%d.attentions.0.attn.out_proj.bias", pre, l); lp.sa_out_b = lb.add(nm, 1, d.C);
    snprintf(nm, sizeof nm, "%s.%d.attentions.1.attn.out_proj.weight", pre, l); lp.ca_out_w = lb.add(nm, 2, d.C, d.C);
    snprintf(nm, sizeof nm, "%s.%d.attentions.1.attn.out_proj.bias", pre, l); lp.ca_out_b = lb.add(nm, 1, d.C);
    snprintf(nm, sizeof nm, "%s.%d.ffns.0.layers.0.0.weight", pre, l); lp.f1_w = lb.add(nm, 2, d.F, d.C);
    snprintf(nm, sizeof nm, "%s.%d.ffns.0.layers.0.0.bias", pre, l); lp.f1_b = lb.add(nm, 1, d.F);
    snprintf(nm, sizeof nm, "%s.%d.ffns.0.layers.1.weight", pre, l); lp.f2_w = lb.add(nm, 2, d.C, d.F);
    snprintf(nm, sizeof nm, "%s.%d.ffns.0.layers.1.bias", pre, l); lp.f2_b = lb.add(nm, 1, d.C);
    for (int i = 0; i < 3; ++i) {
      snprintf(nm, sizeof nm, "%s.%d.norms.%d.weight", pre, l, i); lp.n_g[i] = lb.add(nm, 1, d.C);
      snprintf(nm, sizeof nm, "%s.%d.norms.%d.bias", pre, l, i); lp.n_b[i] = lb.add(nm, 1, d.C);
    }
    P->stage_end[stage++] = lb.cur;
  }
  // ---- final stage ----
  P->stage_begin[stage] = lb.cur;
  for (int l = 0; l < d.NL; ++l) {
    snprintf(nm, sizeof nm, "transformer.decoder.layers.%d.attentions.1.attn.in_proj_weight", l);
    P->lay[l].ca_in_w = lb.add(nm, 2, 3 * d.C, d.C);
    snprintf(nm, sizeof nm, "transformer.decoder.layers.%d.attentions.1.attn.in_proj_bias", l);
    P->lay[l].ca_in_b = lb.add(nm, 1, 3 * d.C);
  }
  P->ca_in_stride = d.NL > 1 ? P->lay[1].ca_in_w - P->lay[0].ca_in_w : 0;
  P->qe_w1 = lb.add("query_embedding.0.weight", 2, d.C, d.C * 3 / 2);
  P->qe_b1 = lb.add("query_embedding.0.bias", 1, d.C);
  P->qe_w2 = lb.add("query_embedding.2.weight", 2, d.C, d.C);
  P->qe_b2 = lb.add("query_embedding.2.bias", 1, d.C);
  P->ref = lb.add("reference_points.weight", 2, d.Q, 3);
  P->pe_w1 = lb.add("position_encoder.0.weight", 4, 4 * d.C, 3 * d.D, 1, 1);
  P->pe_b1 = lb.add("position_encoder.0.bias", 1, 4 * d.C);
  P->pe_w2 = lb.add("position_encoder.2.weight", 4, d.C, 4 * d.C, 1, 1);
  P->pe_b2 = lb.add("position_encoder.2.bias", 1, d.C);
  P->ad_w1 = lb.add("adapt_pos3d.0.weight", 4, 4 * d.C, d.C * 3 / 2, 1, 1);
  P->ad_b1 = lb.add("adapt_pos3d.0.bias", 1, 4 * d.C);
  P->ad_w2 = lb.add("adapt_pos3d.2.weight", 4, d.C, 4 * d.C, 1, 1);
  P->ad_b2 = lb.add("adapt_pos3d.2.bias", 1, d.C);
  P->in_w = lb.add("input_proj.weight", 4, d.C, d.Cin, 1, 1);
  P->in_b = lb.add("input_proj.bias", 1, d.C);
  P->fpe_rw = P->fpe_rb = P->fpe_ew = P->fpe_eb = 0;
  if (c->with_fpe) {   // SELayer (petrv2_head.py:48-60)
    P->fpe_rw = lb.add("fpe.conv_reduce.weight", 4, d.C, d.C, 1, 1);
    P->fpe_rb = lb.add("fpe.conv_reduce.bias", 1, d.C);
    P->fpe_ew = lb.add("fpe.conv_expand.weight", 4, d.C, d.C, 1, 1);
    P->fpe_eb = lb.add("fpe.conv_expand.bias", 1, d.C);
  }
  P->stage_end[stage++] = lb.cur;
  P->n_stages = stage;
  P->code_w = lb.add("code_weights", 1, d.code);   // no gradient (petr_head.py:211-212)
  P->total = lb.cur;
  if (out) {
    out->count = lb.count;
    out->total = lb.cur;
  }
}

// ---------------------------------------------------------------------------------------------
// workspace layout (activations kept for backward + scratch), offsets in floats
// ---------------------------------------------------------------------------------------------
struct LayerW {
  long qkv, qkv16, ao_s, lse_s, z0, mean0, rstd0, x1, xe1, qc, ao_c, lse_c, z1, mean1, rstd1, x2, hff, z2, mean2, rstd2, xe_in;
};
struct WOff {
  long posemb, qe_h, qe, vol, sine, mem, h1, h2, pos, pos2, mempos, p16, k_all, v_all, x0;
  long bits, bits_cross_n, bits_self_n;      // [NL][cross | self] key-major words
  LayerW lay[8];
  long xs, mean_p, rstd_p, outs;
  long c1, c1_mean, c1_rstd, c1n, c2, c2_mean, c2_rstd, c2n, r1, r2, reg_raw;
  long br_t;
  long th_h, pe1, fpe_h, fpe_u;          // PETRv2: task-head hiddens [G][5][RG,C]; SELayer buffers [BL,C]
  long d_th_h, d_pe1, d_fpe_h, d_fpe_u;
  long ffn_part; int ffn_split, ffn_fsplit;
  long mha_ws; size_t mha_ws_bytes;
  long mha_sched_n;
  // ---- backward scratch ----
  long zero_begin, zero_end;     // cleared once per backward (atomic / += targets)
  long d_qc, d_qkv, dk_all, dv_all, d_ref_tmp;
  long d_outs, d_xs, d_mempos, d_mem, d_hpe[2], d_e_slab, d_e, d_qe_h, d_posemb;
  // private gradient scratch (never reused inside one backward, so weight-gradient contractions can run on
  // side streams long after the critical path has moved on)
  long s0_raw, s0_r2, s0_r1, s0_c2n, s0_c2, s0_c1n, s0_c1, s0_outs_c;
  struct LayerG { long d_z2, d_h, d_x2, d_z1, d_ao, d_x1, d_z0, d_ao_s, d_zd[3]; } lg[8];
  // transposed copies of the decoder weights for the 900-row input-gradient contractions (see transpose_batch_kernel)
  struct LayerT { long f2, f1, ca_out, ca_q, sa_out, sa_in; } wt[8];
  long total;
};

struct WsBuilder {
  long cur = 0;
  struct Entry { const char* name; long off, n; } table[384];
  int count = 0;
  long add(const char* name, long n) {
    const long off = cur;
    if (count < 384) table[count++] = {name, off, n};
    cur = align4(cur + n);
    return off;
  }
};

static void build_ws(const petr_head_config* c, WOff* Wf, WsBuilder* wb_out) {
  const Dims d = make_dims(c);
  WsBuilder wb;
  WOff& W = *Wf;
  const long C = d.C;
  W.posemb = wb.add("posemb", (long)d.Q * C * 3 / 2);
  W.qe_h = wb.add("query_embed_hidden", (long)d.Q * C);
  W.qe = wb.add("query_embed", (long)d.Q * C);
  W.vol = wb.add("coords3d", (long)d.B * d.N * 3 * d.D * d.HW);
  W.sine = wb.add("sine", (long)d.B * d.N * (C * 3 / 2) * d.HW);
  W.mem = wb.add("memory", d.BL * C);
  W.h1 = wb.add("pe_hidden", d.BL * 4 * C);
  W.h2 = wb.add("sine_hidden", d.BL * 4 * C);
  W.pos = wb.add("pos_embed", d.BL * C);
  W.pos2 = wb.add("pos_embed_adapt", d.BL * C);      // adapt_pos3d half of the key position embedding until the two are joined
  W.mempos = wb.add("mempos", d.BL * C);
  {   // packed attention-dropout masks (training mode), key-major, per layer: the forward kernels leave them, the backward reads
    W.bits_cross_n = (long)d.B * d.NH * cdiv(d.L, 32) * 32 * cdiv(d.Q, 32);
    W.bits_self_n = (long)d.B * d.NH * cdiv(d.Q, 32) * 32 * cdiv(d.Q, 32);
    W.bits = wb.add("dropout_bits", (long)d.NL * (W.bits_cross_n + W.bits_self_n));
  }
  {   // bf16 copy of the flat parameter buffer (bf16 mode: the token-sized contractions read their weights from it)
    POff Pl;
    build_layout(c, &Pl, nullptr);
    W.p16 = wb.add("params_bf16", (Pl.total + 1) / 2);
  }
  W.k_all = wb.add("k_all", (long)d.B * d.NL * d.L * C);
  W.v_all = wb.add("v_all", (long)d.B * d.NL * d.L * C);
  // the attention tile-ticket counters (petr_mha_fwd_args.sched) sit right behind x0 so that the one fill that
  // zeroes the decoder input at the top of every forward also (re)zeroes them
  W.mha_sched_n = (long)d.B * d.NH * ((d.Q + 127) / 128);
  W.x0 = wb.add("x0", d.BQ * C + W.mha_sched_n);
  for (int l = 0; l < d.NL; ++l) {
    LayerW& lw = W.lay[l];
    lw.xe_in = wb.add("xe_in", d.BQ * C);
    lw.qkv = wb.add("qkv_self", d.BQ * 3 * C);
    lw.qkv16 = wb.add("qkv_self_bf16", d.BQ * 3 * C / 2);      // bf16 mode: the self-attention's K / V rows as bf16 (same indexing)
    lw.ao_s = wb.add("attn_self", d.BQ * C);
    lw.lse_s = wb.add("lse_self", (long)d.B * d.NH * d.Q);
    lw.z0 = wb.add("z0", d.BQ * C);
    lw.mean0 = wb.add("mean0", d.BQ);
    lw.rstd0 = wb.add("rstd0", d.BQ);
    lw.x1 = wb.add("x1", d.BQ * C);
    lw.xe1 = wb.add("xe1", d.BQ * C);
    lw.qc = wb.add("q_cross", d.BQ * C);
    lw.ao_c = wb.add("attn_cross", d.BQ * C);
    lw.lse_c = wb.add("lse_cross", (long)d.B * d.NH * d.Q);
    lw.z1 = wb.add("z1", d.BQ * C);
    lw.mean1 = wb.add("mean1", d.BQ);
    lw.rstd1 = wb.add("rstd1", d.BQ);
    lw.x2 = wb.add("x2", d.BQ * C);
    lw.hff = wb.add("ffn_hidden", d.BQ * d.F);
    lw.z2 = wb.add("z2", d.BQ * C);
    lw.mean2 = wb.add("mean2", d.BQ);
    lw.rstd2 = wb.add("rstd2", d.BQ);
  }
  W.xs = wb.add("xs", d.R * C);
  W.mean_p = wb.add("mean_post", d.R);
  W.rstd_p = wb.add("rstd_post", d.R);
  W.outs = wb.add("outs_dec", d.R * C);
  W.c1 = wb.add("c1", d.R * C);
  W.c1_mean = wb.add("c1_mean", d.R);
  W.c1_rstd = wb.add("c1_rstd", d.R);
  W.c1n = wb.add("c1n", d.R * C);
  W.c2 = wb.add("c2", d.R * C);
  W.c2_mean = wb.add("c2_mean", d.R);
  W.c2_rstd = wb.add("c2_rstd", d.R);
  W.c2n = wb.add("c2n", d.R * C);
  W.r1 = wb.add("r1", d.R * C);
  W.r2 = wb.add("r2", d.R * C);
  W.reg_raw = wb.add("reg_raw", d.R * d.code);
  W.th_h = W.pe1 = W.fpe_h = W.fpe_u = W.d_th_h = W.d_pe1 = W.d_fpe_h = W.d_fpe_u = 0;
  if (c->with_multi) W.th_h = wb.add("task_hidden", 5 * d.R * C);
  if (c->with_fpe) {
    W.pe1 = wb.add("pe_pre_gate", d.BL * C);
    W.fpe_h = wb.add("fpe_hidden", d.BL * C);
    W.fpe_u = wb.add("fpe_gate_logit", d.BL * C);
  }
  // split-K of the second FFN contraction (K = F, only 4x15 output tiles otherwise)
  W.ffn_split = d.BQ <= 2048 ? 4 : 1;
  if (d.F / 32 < W.ffn_split) W.ffn_split = 1;
  // petr_ffn_fwd (both contractions in one launch): 32-row blocks x hidden slices, about one workgroup per CU; 0 = not applicable
  W.ffn_fsplit = 0;
  if (C == 256) {
    const long nrb = cdiv(d.BQ, 32);
    for (int ns = 8; ns >= 1 && !W.ffn_fsplit; ns >>= 1)
      if (d.F % (256 * ns) == 0 && (nrb * ns <= 256 || ns == 1)) W.ffn_fsplit = ns;
  }
  W.ffn_part = wb.add("ffn_partials", (long)(W.ffn_fsplit > W.ffn_split ? W.ffn_fsplit : W.ffn_split) * d.BQ * C);
  {
    size_t a = petr_mha_fwd_workspace_bytes(d.B, d.NH, d.Q, (int)d.L, 0);
    size_t b = petr_mha_fwd_workspace_bytes(d.B, d.NH, d.Q, d.Q, 0);
    size_t e = petr_mha_bwd_workspace_bytes(d.B, d.NH, d.Q, (int)d.L);
    size_t h = petr_mha_fwd_bf16_workspace_bytes(d.B, d.NH, d.Q, (int)d.L, 0);
    size_t m = a > b ? a : b;
    if (e > m) m = e;
    if (h > m) m = h;
    W.mha_ws_bytes = m + 16;
    W.mha_ws = wb.add("mha_ws", (long)(W.mha_ws_bytes / 4) + 4);
  }
  // ---- backward ----
  W.zero_begin = wb.cur;
  W.d_qc = wb.add("d_qc", (long)d.NL * d.BQ * C);
  W.d_qkv = wb.add("d_qkv", (long)d.NL * d.BQ * 3 * C);
  W.dk_all = wb.add("dk_all", (long)d.B * d.NL * d.L * C);
  W.dv_all = wb.add("dv_all", (long)d.B * d.NL * d.L * C);
  W.d_ref_tmp = wb.add("d_ref_tmp", (long)d.Q * 3);
  W.zero_end = wb.cur;
  W.d_outs = wb.add("d_outs", d.R * C);
  W.d_xs = wb.add("d_xs", d.R * C);
  if (c->with_multi) W.d_th_h = wb.add("d_task_hidden", 5 * d.R * C);
  if (c->with_fpe) {
    W.d_pe1 = wb.add("d_pe1", d.BL * C);
    W.d_fpe_h = wb.add("d_fpe_h", d.BL * C);
    W.d_fpe_u = wb.add("d_fpe_u", d.BL * C);
  }
  W.s0_raw = wb.add("s0_raw", d.R * d.code);
  W.s0_r2 = wb.add("s0_r2", d.R * C);
  W.s0_r1 = wb.add("s0_r1", d.R * C);
  W.s0_c2n = wb.add("s0_c2n", d.R * C);
  W.s0_c2 = wb.add("s0_c2", d.R * C);
  W.s0_c1n = wb.add("s0_c1n", d.R * C);
  W.s0_c1 = wb.add("s0_c1", d.R * C);
  W.s0_outs_c = wb.add("s0_outs_c", d.R * C);
  for (int l = 0; l < d.NL; ++l) {
    WOff::LayerG& g = W.lg[l];
    g.d_z2 = wb.add("g_d_z2", d.BQ * C);
    g.d_h = wb.add("g_d_h", d.BQ * d.F);
    g.d_x2 = wb.add("g_d_x2", (long)(W.ffn_fsplit > W.ffn_split ? W.ffn_fsplit : W.ffn_split) * d.BQ * C);
    g.d_z1 = wb.add("g_d_z1", d.BQ * C);
    g.d_ao = wb.add("g_d_ao", d.BQ * C);
    g.d_x1 = wb.add("g_d_x1", d.BQ * C);
    g.d_z0 = wb.add("g_d_z0", d.BQ * C);
    g.d_ao_s = wb.add("g_d_ao_s", d.BQ * C);
    for (int i = 0; i < 3; ++i) g.d_zd[i] = wb.add("g_d_z_dropped", d.BQ * C);   // training mode only
  }
  for (int l = 0; l < d.NL; ++l) {
    WOff::LayerT& t = W.wt[l];
    t.f2 = wb.add("wT_ffn2", (long)d.F * C);
    t.f1 = wb.add("wT_ffn1", (long)C * d.F);
    t.ca_out = wb.add("wT_ca_out", (long)C * C);
    t.ca_q = wb.add("wT_ca_q", (long)C * C);
    t.sa_out = wb.add("wT_sa_out", (long)C * C);
    t.sa_in = wb.add("wT_sa_in", (long)C * 3 * C);
  }
  // transposed copies of the two 256 x 256 weights of the class and the box branch, per branch group: [cls 0, cls 1, reg 0, reg 1]
  W.br_t = wb.add("wT_branches", (long)(c->shared_branches ? 1 : d.NL) * 4 * C * C);
  W.d_mempos = wb.add("d_mempos", d.BL * C);
  W.d_mem = wb.add("d_mem", d.BL * C);
  W.d_hpe[0] = wb.add("d_hpe", d.BL * 4 * C);
  W.d_hpe[1] = wb.add("d_hpe", d.BL * 4 * C);
  W.d_e_slab = wb.add("d_e_slab", (long)2 * d.NL * d.BQ * C);      // [2][NL][BQ, C]: cross-attention q rows, self-attention q/k rows
  W.d_e = wb.add("d_e", (long)d.Q * C);
  W.d_qe_h = wb.add("d_qe_h", (long)d.Q * C);
  W.d_posemb = wb.add("d_posemb", (long)d.Q * C * 3 / 2);
  W.total = wb.cur;
  if (wb_out) *wb_out = wb;
}

// ---------------------------------------------------------------------------------------------
// small call builders
// ---------------------------------------------------------------------------------------------
static petr_gemm_args gemm0() {
  petr_gemm_args g;
  memset(&g, 0, sizeof g);
  g.nb0 = g.nb1 = g.split_k = 1;
  g.alpha = 1.f;
  return g;
}
// y[M,N] = x[M,K] @ w[N,K]^T (+bias)
static petr_gemm_args lin_fwd(const float* x, const float* w, const float* bias, float* y, long M, int N, int K) {
  petr_gemm_args g = gemm0();
  g.a = x; g.lda = K; g.a_kcontig = 1;
  g.b = w; g.ldb = K; g.b_kcontig = 1;
  g.c = y; g.ldc = N; g.bias = bias;
  g.M = (int)M; g.N = N; g.K = K;
  return g;
}
// dx[M,K] = dy[M,N] @ w[N,K]
static petr_gemm_args lin_dgrad(const float* dy, const float* w, float* dx, long M, int N, int K) {
  petr_gemm_args g = gemm0();
  g.a = dy; g.lda = N; g.a_kcontig = 1;
  g.b = w; g.ldb = K; g.b_kcontig = 0;
  g.c = dx; g.ldc = K;
  g.M = (int)M; g.N = K; g.K = N;
  return g;
}
// dw[N,K] += dy[M,N]^T @ x[M,K] ; db[N] += colsum(dy)   (float atomics, split over the M rows)
static petr_gemm_args lin_wgrad(const float* dy, long ldy, const float* x, long ldx, float* dw, float* db, long M, int N,
                                int K) {
  petr_gemm_args g = gemm0();
  g.a = dy; g.lda = ldy; g.a_kcontig = 0;
  g.b = x; g.ldb = ldx; g.b_kcontig = 0;
  g.c = dw; g.ldc = K;
  g.M = N; g.N = K; g.K = (int)M;
  g.flags = PETR_GEMM_ATOMIC;
  g.a_colsum = db;
  const long tiles = cdiv(N, 64) * cdiv(K, 64);
  long sk = 512 / (tiles > 0 ? tiles : 1);
  const long ktiles = cdiv(M, 32);
  if (sk > ktiles / 2) sk = ktiles / 2;
  if (sk < 1) sk = 1;
  if (sk > 64) sk = 64;
  g.split_k = (int)sk;
  return g;
}

// 1/(1-p) exactly as the kernels apply it (make_drop rounds p to a 32-bit threshold first)
static float hidden_drop_scale(const petr_dropout& d) { return make_drop(d).scale; }

#define RUN(expr)                 \
  do {                            \
    const int rc__ = (expr);      \
    if (rc__ != PETR_OK) return rc__; \
  } while (0)

static int ln_fwd(const float* x, int np, long pstride, const float* bias, const float* res, const float* g,
                  const float* b, float* y, float* z_out, float* mean, float* rstd, long M, int C, int flags, float* y2,
                  const float* add2, int add2_rows, void* s, const petr_dropout* drop = nullptr) {
  petr_layernorm_args a;
  memset(&a, 0, sizeof a);
  if (drop) a.drop = *drop;
  a.x = x; a.n_partials = np; a.partial_stride = pstride; a.bias = bias; a.residual = res; a.gamma = g; a.beta = b;
  a.y = y; a.z_out = z_out; a.mean = mean; a.rstd = rstd; a.M = (int)M; a.C = C; a.eps = 1e-5f; a.flags = flags;
  a.y2 = y2; a.add2 = add2; a.add2_rows = add2_rows;
  return petr_layernorm_fwd(&a, s);
}

// merge of the attention partials + out-projection + dropout + residual + LayerNorm (+ query_pos add) in one launch
static int attn_out_ln(float* ao, const float* ws, int n_split, const Dims& d, float attn_scale, float* lse, const float* wT,
                       const float* bias, const float* res, const petr_dropout* drop, const float* g, const float* b, float* z,
                       float* mean, float* rstd, float* y, float* y2, const float* add2, int add2_rows, void* s,
                       const float* w2T = nullptr, const float* bias2 = nullptr, float* out2 = nullptr) {
  petr_attn_out_ln_args a;
  memset(&a, 0, sizeof a);
  a.a = ao; a.n_split = n_split; a.B = d.B; a.H = d.NH; a.Q = d.Q; a.attn_scale = attn_scale; a.lse = lse;
  if (n_split > 1) { a.o_part = ws; a.ml_part = ws + (long)n_split * d.B * d.NH * d.Q * 32; }
  a.wT = wT; a.bias = bias; a.residual = res;
  if (drop) a.drop = *drop;
  a.gamma = g; a.beta = b; a.eps = 1e-5f; a.z = z; a.mean = mean; a.rstd = rstd; a.y = y; a.y2 = y2; a.add2 = add2;
  a.add2_rows = add2_rows; a.M = (int)d.BQ; a.w2T = w2T; a.bias2 = bias2; a.out2 = out2;
  return petr_attn_out_ln(&a, s);
}

static int ln_bwd(const float* z, const float* mean, const float* rstd, const float* g, const float* dy, const float* y,
                  float* dz, float* dg, float* db, long M, int C, int flags, int accumulate, void* s, int dy_partials = 1,
                  long dy_pstride = 0, const float* dy_res = nullptr, float* dz_drop = nullptr,
                  const petr_dropout* drop = nullptr) {
  petr_layernorm_bwd_args a;
  memset(&a, 0, sizeof a);
  if (drop && dz_drop) { a.drop = *drop; a.dz_drop = dz_drop; }
  a.dy_partials = dy_partials; a.dy_partial_stride = dy_pstride; a.dy_residual = dy_res;
  a.z = z; a.mean = mean; a.rstd = rstd; a.gamma = g; a.dy = dy; a.y = y; a.dz = dz; a.dgamma = dg; a.dbeta = db;
  a.ws = nullptr; a.M = (int)M; a.C = C; a.flags = flags; a.dz_accumulate = accumulate;
  return petr_layernorm_bwd(&a, s);
}

// LayerNorm backward + the input gradient of the branch's linear layer (dx = dz W: the weight as stored) in one launch
static int ln_bwd_proj(const float* z, const float* mean, const float* rstd, const float* g, const float* dy, int dy_partials,
                       long dy_pstride, const float* dy_res, float* dz, float* dz_drop, const petr_dropout* drop, float* dg,
                       float* db, long M, const float* w, int n2, float alpha, const float* relu_mask, float* out, void* s,
                       const float* pre_a = nullptr, const float* pre_w = nullptr, int pre_n = 0) {
  petr_ln_bwd_proj_args a;
  memset(&a, 0, sizeof a);
  a.z = z; a.mean = mean; a.rstd = rstd; a.gamma = g; a.dy = dy; a.dy_partials = dy_partials; a.dy_partial_stride = dy_pstride;
  a.dy_residual = dy_res; a.dz = dz; a.dz_drop = dz_drop;
  if (drop && dz_drop) a.drop = *drop;
  a.dgamma = dg; a.dbeta = db; a.M = (int)M; a.w = w; a.n2 = n2; a.alpha = alpha; a.relu_mask = relu_mask; a.out = out;
  a.pre_a = pre_a; a.pre_w = pre_w; a.pre_n = pre_n;
  return petr_ln_bwd_proj(&a, s);
}

static int mha_f(const float* q, long q_bs, long q_rs, const float* k, long k_bs, long k_rs, const float* v, float* o,
                 float* lse, const uint8_t* kpm, const Dims& d, int L, float* ws, size_t ws_bytes, int* sched, void* s,
                 const petr_dropout* drop = nullptr, uint32_t* bits = nullptr, int n_split = 0, int defer_merge = 0) {
  petr_mha_fwd_args a;
  memset(&a, 0, sizeof a);
  if (drop) a.drop = *drop;
  a.q = q; a.q_bs = q_bs; a.q_hs = 32; a.q_rs = q_rs;
  a.k = k; a.k_bs = k_bs; a.k_hs = 32; a.k_rs = k_rs;
  a.v = v; a.v_bs = k_bs; a.v_hs = 32; a.v_rs = k_rs;
  a.o = o; a.o_bs = (long)d.Q * d.C; a.o_hs = 32; a.o_rs = d.C;
  a.lse = lse; a.kpm = kpm; a.B = d.B; a.H = d.NH; a.Q = d.Q; a.L = L;
  a.scale = 1.0f / sqrtf(32.f);
  a.n_split = n_split; a.ws = ws; a.ws_bytes = ws_bytes; a.sched = sched; a.drop_bits = bits; a.defer_merge = defer_merge;
  return petr_mha_fwd(&a, s);
}

// cross-attention with bf16 K/V (io->attn_bf16)
static int mha_f_bf16(const float* q, long q_bs, long q_rs, const uint16_t* k, long k_bs, long k_rs, const uint16_t* v,
                      float* o, float* lse, const uint8_t* kpm, const Dims& d, int L, float* ws, size_t ws_bytes, void* s,
                      const petr_dropout* drop = nullptr, uint32_t* bits = nullptr, int n_split = 0, int defer_merge = 0) {
  petr_mha_fwd_bf16_args a;
  memset(&a, 0, sizeof a);
  if (drop) a.drop = *drop;
  a.q = q; a.q_bs = q_bs; a.q_hs = 32; a.q_rs = q_rs;
  a.k = k; a.k_bs = k_bs; a.k_hs = 32; a.k_rs = k_rs;
  a.v = v; a.v_bs = k_bs; a.v_hs = 32; a.v_rs = k_rs;
  a.o = o; a.o_bs = (long)d.Q * d.C; a.o_hs = 32; a.o_rs = d.C;
  a.lse = lse; a.kpm = kpm; a.B = d.B; a.H = d.NH; a.Q = d.Q; a.L = L;
  a.scale = 1.0f / sqrtf(32.f);
  a.n_split = n_split; a.ws = ws; a.ws_bytes = ws_bytes; a.drop_bits = bits; a.defer_merge = defer_merge;
  return petr_mha_fwd_bf16(&a, s);
}

static int mha_b(const float* q, long q_bs, long q_rs, const float* k, long k_bs, long k_rs, const float* v,
                 const float* o, const float* d_o, const float* lse, const uint8_t* kpm, float* dq, float* dk, float* dv,
                 const Dims& d, int L, float* ws, size_t ws_bytes, void* s, const petr_dropout* drop = nullptr,
                 const uint32_t* bits = nullptr) {
  petr_mha_bwd_args a;
  memset(&a, 0, sizeof a);
  if (drop) a.drop = *drop;
  a.q = q; a.q_bs = q_bs; a.q_hs = 32; a.q_rs = q_rs;
  a.k = k; a.k_bs = k_bs; a.k_hs = 32; a.k_rs = k_rs;
  a.v = v; a.v_bs = k_bs; a.v_hs = 32; a.v_rs = k_rs;
  a.o = o; a.o_bs = (long)d.Q * d.C; a.o_hs = 32; a.o_rs = d.C;
  a.d_o = d_o; a.do_bs = (long)d.Q * d.C; a.do_hs = 32; a.do_rs = d.C;
  a.lse = lse; a.kpm = kpm;
  a.dq = dq; a.dq_bs = q_bs; a.dq_hs = 32; a.dq_rs = q_rs;
  a.dk = dk; a.dk_bs = k_bs; a.dk_hs = 32; a.dk_rs = k_rs;
  a.dv = dv; a.dv_bs = k_bs; a.dv_hs = 32; a.dv_rs = k_rs;
  a.B = d.B; a.H = d.NH; a.Q = d.Q; a.L = L;
  a.scale = 1.0f / sqrtf(32.f);
  a.ws = ws; a.ws_bytes = ws_bytes; a.drop_bits = bits;
  return petr_mha_bwd(&a, s);
}

// gradient of mha_f_bf16 (io->attn_bf16 training step)
static int mha_b_bf16(const float* q, long q_bs, long q_rs, const uint16_t* k, long k_bs, long k_rs, const uint16_t* v,
                      const float* o, const float* d_o, const float* lse, const uint8_t* kpm, float* dq, uint16_t* dk,
                      uint16_t* dv, bool dkv16, const Dims& d, int L, void* s, const petr_dropout* drop = nullptr,
                      const uint32_t* bits = nullptr, bool overwrite = true) {
  petr_mha_bwd_bf16_args a;
  memset(&a, 0, sizeof a);
  if (drop) a.drop = *drop;
  a.q = q; a.q_bs = q_bs; a.q_hs = 32; a.q_rs = q_rs;
  a.k = k; a.k_bs = k_bs; a.k_hs = 32; a.k_rs = k_rs;
  a.v = v; a.v_bs = k_bs; a.v_hs = 32; a.v_rs = k_rs;
  a.o = o; a.o_bs = (long)d.Q * d.C; a.o_hs = 32; a.o_rs = d.C;
  a.d_o = d_o; a.do_bs = (long)d.Q * d.C; a.do_hs = 32; a.do_rs = d.C;
  a.lse = lse; a.kpm = kpm;
  a.dq = dq; a.dq_bs = q_bs; a.dq_hs = 32; a.dq_rs = q_rs;
  a.dk = reinterpret_cast<float*>(dk); a.dk_bs = k_bs; a.dk_hs = 32; a.dk_rs = k_rs;
  a.dv = reinterpret_cast<float*>(dv); a.dv_bs = k_bs; a.dv_hs = 32; a.dv_rs = k_rs;
  a.B = d.B; a.H = d.NH; a.Q = d.Q; a.L = L;
  a.scale = 1.0f / sqrtf(32.f);
  a.dkv_overwrite = overwrite ? 1 : 0;   // cross: dK_l / dV_l are stored, not accumulated (the executor does not zero them in this mode)
  a.drop_bits = bits;
  a.dkv_bf16 = dkv16;        // ... and as bf16: the K/V-projection backward rounds them to bf16 anyway (PETR_GEMM_A_BF16)
  return petr_mha_bwd_bf16(&a, s);
}

// ---- transposed decoder weights for the backward ----
// dx = dy W reads the nn.Linear weight W[N_out][K_in] along its rows: as the B operand of the contraction (k = n_out) it is
// "K-major", which the 900-row kernels can only read with 4-byte loads (gemm.hip: the float4 fragment path needs
// K-contiguous operands) - measured 19-22 us per 900 x 256 x 256 input gradient against 12 us for the same-sized forward.
// One batched transposition per backward (36 matrices, ~35 MB read + written, on a side stream beside the branch
// backward) turns every decoder-layer input gradient into a forward-shaped contraction.
struct TrEntry { const float* src; float* dst; long ld; int rows, cols, tile0; };     // dst[c][r] = src[r * ld + c]
struct TrBatch { int n; TrEntry e[64]; };
__global__ __launch_bounds__(256) void transpose_batch_kernel(const TrBatch b) {
  __shared__ float tile[32][33];
  int ei = 0;
  while (ei + 1 < b.n && (int)blockIdx.x >= b.e[ei + 1].tile0) ++ei;
  const TrEntry& e = b.e[ei];
  const int t = blockIdx.x - e.tile0, tiles_c = (e.cols + 31) >> 5;
  const int r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + ty + 8 * j, c = c0 + tx;
    tile[ty + 8 * j][tx] = (r < e.rows && c < e.cols) ? e.src[(long)r * e.ld + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + 8 * j, r = r0 + tx;
    if (r < e.rows && c < e.cols) e.dst[(long)c * e.rows + r] = tile[tx][ty + 8 * j];
  }
}
// transposed copies of the decoder weights into W.wt (one launch): `ffn` = the two FFN matrices (read by petr_ffn_fwd in the
// forward and by the FFN input gradients), `rest` = the attention projections (read by the backward's input gradients)
template <class PL, class WL>
static int launch_weight_transposes(const float* Pm, float* Wm, const PL& P, const WL& W, int NL, int C, int F, bool ffn, bool rest,
                                    hipStream_t st, int branch_groups = 0) {
  TrBatch tb;
  tb.n = 0;
  int tiles = 0;
  auto add = [&](const float* src, long dst_off, int rows, int cols) {
    TrEntry& en = tb.e[tb.n++];
    en.src = src; en.dst = Wm + dst_off; en.ld = cols; en.rows = rows; en.cols = cols; en.tile0 = tiles;
    tiles += (int)(cdiv(rows, 32) * cdiv(cols, 32));
  };
  for (int l = 0; l < NL; ++l) {
    const auto& lp = P.lay[l];
    const auto& t = W.wt[l];
    if (ffn) {
      add(Pm + lp.f2_w, t.f2, C, F);
      add(Pm + lp.f1_w, t.f1, F, C);
    }
    if (rest) {
      add(Pm + lp.ca_out_w, t.ca_out, C, C);
      add(Pm + lp.ca_in_w, t.ca_q, C, C);            // q rows of the cross-attention in_proj
      add(Pm + lp.sa_out_w, t.sa_out, C, C);
      add(Pm + lp.sa_in_w, t.sa_in, 3 * C, C);
    }
  }
  for (int gi = 0; gi < branch_groups; ++gi) {      // petr_branch_fwd's k-major weights (forward only)
    const long o = W.br_t + (long)gi * 4 * C * C, po = (long)gi * P.br_stride;
    add(Pm + P.cls_w[0] + po, o, C, C);
    add(Pm + P.cls_w[1] + po, o + (long)C * C, C, C);
    add(Pm + P.reg_w[0] + po, o + (long)2 * C * C, C, C);
    add(Pm + P.reg_w[1] + po, o + (long)3 * C * C, C, C);
  }
  if (!tiles) return PETR_OK;
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(tiles), dim3(256), 0, st, tb);
  PETR_LAUNCH_CHECK("transpose_batch");
  return PETR_OK;
}
static bool env_on(const char* name) { return petr_tune(name, 1) != 0; }    // default on; product builds: always on (common.h)
// does the forward leave the transposed weight copies in the workspace?  (head_fwd makes them, head_bwd asks)
template <class D>
static bool fwd_transposes(const petr_head_config*, const D& d, int C) {
  static const bool fuse_env = env_on("PETR_FUSE_OUT_LN"), ffn_env = env_on("PETR_FFN_FUSED");
  return C == 256 && ((fuse_env && d.NH == 8) || ffn_env);
}
// dx[M,K] = dy[M,N] @ w[N,K] through wT[K][N]: both operands K-contiguous, like a forward
static petr_gemm_args lin_dgrad_t(const float* dy, const float* wT, float* dx, long M, int N, int K) {
  petr_gemm_args g = gemm0();
  g.a = dy; g.lda = N; g.a_kcontig = 1;
  g.b = wT; g.ldb = N; g.b_kcontig = 1;
  g.c = dx; g.ldc = K;
  g.M = (int)M; g.N = K; g.K = N;
  return g;
}

// bf16 mode stores the two 4C-wide position-embedding hiddens (relu outputs, the largest activations of the step) and their
// gradients as bf16 in the front half of their fp32 buffers: every contraction that touches them rounds to bf16 on load
// anyway, so only the bytes change.  Needs 16-byte K-contiguous rows on the bf16 side (C % 8 == 0 always holds).
static bool hidden_bf16(const petr_head_io* io) {
  static const bool on = env_on("PETR_HID16");       // PETR_HID16=0: fp32 storage (the same-box A/B switch)
  return io->attn_bf16 != 0 && on;
}
// bf16 mode keeps memory (input_proj's output) and key = memory + key_pos as bf16 and reads every token-sized
// contraction's weights from a bf16 copy of the parameter buffer made once per forward: with 128-row tiles a
// contraction re-reads its whole weight panel per tile, so at 24 000 tokens the K/V projections moved as many bytes of
// (L2-resident) weights as of activations - halving both is what the deep-step kernel is bound by (PETR_TOK16=0: fp32)
static bool token_bf16(const petr_head_config* c, const petr_head_io* io) {
  static const bool on = env_on("PETR_TOK16");
  // bf16-source K-contiguous operands are read in 16-byte pieces: every contraction length on the token side must be a
  // multiple of 8 (3 D of the coordinate volume and the input channels can be anything in a toy configuration)
  return io->attn_bf16 != 0 && on && (3 * c->depth_num) % 8 == 0 && c->C_in % 8 == 0 && c->embed_dims % 16 == 0;
}
static bool dkv_bf16(const petr_head_io* io) {       // dK / dV of the cross-attention stored as bf16 (PETR_DKV16=0: fp32)
  static const bool on = env_on("PETR_DKV16");
  return io->attn_bf16 != 0 && on;
}

// route a token-sized gradient contraction to the bf16 matrix cores (gemm_bf16.hip): 128 x 128 output tiles, so the
// K split of a weight gradient is re-derived for that tile size (aim: ~2 workgroups per CU)
static petr_gemm_args to_bf16(petr_gemm_args g) {
  g.flags |= PETR_GEMM_BF16;
  if (g.flags & PETR_GEMM_ATOMIC) {
    const long nb = (long)(g.nb0 > 0 ? g.nb0 : 1) * (g.nb1 > 0 ? g.nb1 : 1);
    const long tiles = cdiv(g.M, 128) * cdiv(g.N, 128) * nb;
    const int kseg = g.k_seg > 0 ? g.k_seg : g.K;
    const long ktiles = (g.k_seg > 0 ? g.K / g.k_seg : 1) * cdiv(kseg, 32);
    long sk = 512 / (tiles > 0 ? tiles : 1);
    if (sk > ktiles / 4) sk = ktiles / 4;
    if (sk < 1) sk = 1;
    if (sk > 128) sk = 128;
    g.split_k = (int)sk;
  }
  return g;
}

}  // namespace

// =============================================================================================
extern "C" int petr_head_layout(const petr_head_config* cfg, petr_head_layout_t* out) {
  RUN(check_config(cfg));
  PETR_CHECK(out, PETR_ERR_INVALID, "head_layout: null output");
  POff P;
  build_layout(cfg, &P, out);
  PETR_CHECK(out->count <= PETR_MAX_PARAMS, PETR_ERR_UNSUPPORTED, "head_layout: too many tensors");
  return PETR_OK;
}

extern "C" size_t petr_head_workspace_bytes(const petr_head_config* cfg) {
  if (check_config(cfg) != PETR_OK) return 0;
  WOff W;
  build_ws(cfg, &W, nullptr);
  return (size_t)W.total * sizeof(float);
}

extern "C" int petr_head_ws_view(const petr_head_config* cfg, const char* name, long* offset_floats, long* numel) {
  RUN(check_config(cfg));
  PETR_CHECK(name && offset_floats && numel, PETR_ERR_INVALID, "head_ws_view: null argument");
  WOff W;
  WsBuilder wb;
  build_ws(cfg, &W, &wb);
  // "<name>" (first occurrence) or "<name>.<layer>" (occurrence index) for per-layer buffers
  char base[64];
  int want = 0;
  snprintf(base, sizeof base, "%s", name);
  char* dot = strrchr(base, '.');
  if (dot && dot[1] >= '0' && dot[1] <= '9') {
    want = atoi(dot + 1);
    *dot = 0;
  }
  int seen = 0;
  for (int i = 0; i < wb.count; ++i) {
    if (strcmp(wb.table[i].name, base) == 0) {
      if (seen == want) {
        *offset_floats = wb.table[i].off;
        *numel = wb.table[i].n;
        return PETR_OK;
      }
      ++seen;
    }
  }
  petr_set_error("head_ws_view: no buffer named '%s'", name);
  return PETR_ERR_INVALID;
}

extern "C" int petr_head_bwd_num_stages(const petr_head_config* cfg) {
  if (check_config(cfg) != PETR_OK) return -1;
  POff P;
  build_layout(cfg, &P, nullptr);
  return P.n_stages;
}

extern "C" int petr_head_bwd_stage_range(const petr_head_config* cfg, int stage, long* begin, long* end) {
  RUN(check_config(cfg));
  POff P;
  build_layout(cfg, &P, nullptr);
  PETR_CHECK(stage >= 0 && stage < P.n_stages && begin && end, PETR_ERR_INVALID, "head_bwd_stage_range: bad stage");
  *begin = P.stage_begin[stage];
  *end = P.stage_end[stage];
  return PETR_OK;
}

// =============================================================================================
// forward
// =============================================================================================
extern "C" int petr_head_fwd(const petr_head_config* cfg, const petr_head_io* io, void* stream) {
  RUN(check_config(cfg));
  PETR_CHECK(io && io->params && (io->feats || io->memory_in) && io->img2lidar && io->depth && io->dim_t && io->all_cls_scores &&
                 io->all_bbox_preds && io->ws,
             PETR_ERR_INVALID, "head_fwd: null pointer");
  PETR_CHECK(!cfg->has_mask || io->mask, PETR_ERR_INVALID, "head_fwd: has_mask without a mask");
  const Dims d = make_dims(cfg);
  POff P;
  build_layout(cfg, &P, nullptr);
  WOff W;
  build_ws(cfg, &W, nullptr);
  PETR_CHECK(io->ws_bytes >= (size_t)W.total * sizeof(float), PETR_ERR_WORKSPACE, "head_fwd: workspace %zu < %zu bytes",
             io->ws_bytes, (size_t)W.total * sizeof(float));
  PETR_CHECK(aligned16(io->params) && aligned16(io->ws) && aligned16(io->feats) && aligned16(io->memory_in), PETR_ERR_INVALID,
             "head_fwd: params / ws / feats / memory_in must be 16-byte aligned");
  PETR_CHECK(!io->memory_in || !(io->dropout_p > 0.f), PETR_ERR_UNSUPPORTED,
             "head_fwd: memory_in (input_proj folded into the producer) is an inference path; dropout_p must be 0");
  const float* Pm = io->params;
  float* Wm = (float*)io->ws;
  const int C = d.C;
  const uint8_t* kpm = cfg->has_mask ? io->mask : nullptr;
  const Lanes ln{(hipStream_t)stream, (petr_ctx*)io->ctx};
  void* s = ln.m();        // critical path
  void* s1 = ln.side(0);   // position-embedding branch A (coords3d MLP) / K projection / reg branch
  void* s2 = ln.side(1);   // position-embedding branch B (sine MLP, input_proj) / V projection
  const int V = d.B * d.N;
  const float* E = Wm + W.qe;
  // io->attn_bf16 (BASELINE configs 3-5): every token-sized contraction runs on the bf16 matrix cores (fp32 operands
  // rounded on load, fp32 accumulation), the K/V projections store bf16 (PETR_GEMM_BF16 | PETR_GEMM_STORE_BF16) into the
  // front half of the fp32 K/V buffers, which stay otherwise unwritten, and the cross-attention (forward AND
  // backward: petr_mha_fwd_bf16 / petr_mha_bwd_bf16) reads them; query-sized work (900 rows) stays fp32
  const bool attn_bf16 = io->attn_bf16 != 0;
  const bool hid16 = hidden_bf16(io);
  const bool tok16 = token_bf16(cfg, io);
  uint16_t* p16 = reinterpret_cast<uint16_t*>(Wm + W.p16);
  // weight at parameter offset `off`: the bf16 copy (element-indexed like the fp32 buffer) or the parameter itself
  auto Wp = [&](long off) -> const float* { return tok16 ? reinterpret_cast<const float*>(p16 + off) : Pm + off; };
  const int wflag = tok16 ? (PETR_GEMM_BF16 | PETR_GEMM_B_BF16) : 0;
  uint16_t* mem16 = reinterpret_cast<uint16_t*>(Wm + W.mem);          // tok16: bf16 images in the front half of the
  uint16_t* mempos16 = reinterpret_cast<uint16_t*>(Wm + W.mempos);    // fp32 buffers, same element indexing
  if (tok16) RUN(petr_cast_bf16(Pm, p16, P.total, s));
  static const bool ffn16_env = petr_tune("PETR_FFN16", 0) != 0;   // opt-in: the fused fp32 FFN kernels measured faster
  const bool ffn16 = tok16 && ffn16_env;
  // bf16 mode: the 1x1 convolutions over NCHW maps take the K-major variant of the bf16 contraction where it applies
  auto bf16_km = [&](const petr_gemm_args& q) {
    return attn_bf16 && q.K % 32 == 0 && (long)q.M * q.N * (q.nb0 > 0 ? q.nb0 : 1) >= 128L * 128 * 64 && !(q.lda & 3) && !(q.ldb & 3);
  };
  uint16_t* k16 = reinterpret_cast<uint16_t*>(Wm + W.k_all);
  uint16_t* v16 = reinterpret_cast<uint16_t*>(Wm + W.v_all);

  // The K / V projections of all layers are layer-invariant in their input, but only layer 0's are needed before the first
  // cross-attention: with side streams they are issued as [layer 0] + [layers 1..NL-1], the main stream waits for the
  // first part only and the second one runs beside decoder layer 0 (whose 900-row kernels leave most CUs idle).
  hipEvent_t ev_k0 = nullptr, ev_v0 = nullptr;
  auto kv_project = [&](petr_gemm_args g, void* side, hipEvent_t* ev0) -> int {
    // opt-in (PETR_KV_FWD_SPLIT=1): same-box A/B on MI355X (scripts/ab_overlap.sh) was neutral at c5 and p4-1600, fp32
    // and bf16 (within 0.3 %), so the single batched contraction stays the default
    static const bool split = petr_tune("PETR_KV_FWD_SPLIT", 0) != 0;
    if (!ln.ctx || d.NL < 2 || !split) return petr_gemm(&g, side);
    const bool st16 = (g.flags & PETR_GEMM_STORE_BF16) != 0;     // bf16 store: c strides count 2-byte elements
    petr_gemm_args g0 = g;
    g0.nb1 = 1;
    RUN(petr_gemm(&g0, side));
    *ev0 = ln.next();
    (void)hipEventRecord(*ev0, (hipStream_t)side);
    petr_gemm_args g1 = g;
    g1.nb1 = d.NL - 1;
    g1.b = (g.flags & PETR_GEMM_B_BF16) ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(g.b) + g.b_bs1)
                                        : g.b + g.b_bs1;
    g1.bias = g.bias + g.bias_bs1;
    g1.c = st16 ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(g.c) + g.c_bs1) : g.c + g.c_bs1;
    return petr_gemm(&g1, side);
  };
  ln.fork(0);
  ln.fork(1);
  // both FFN contractions in one launch (fp32 FFN; PETR_FFN_FUSED=0: two contractions): it reads the weights k-major
  static const bool ffn_fused_env = env_on("PETR_FFN_FUSED");
  const bool ffn_fused = ffn_fused_env && !ffn16 && W.ffn_fsplit > 0;
  // the 16- / 32-row kernels (petr_attn_out_ln, petr_ln_proj, petr_ffn_fwd) stream their weights k-major: transposed copies of
  // the decoder weights, made here once per forward (the backward's input-gradient contractions read them too)
  // one launch per prediction branch (petr_branch_fwd) instead of contraction / LayerNorm chains (PETR_BRANCH_FUSED=0, diagnostic
  // builds: the chains); same condition as the transposed decoder weights it shares the launch with
  const int BG = cfg->shared_branches ? 1 : d.NL;
  const bool branch_fused = fwd_transposes(cfg, d, C) && petr_tune("PETR_BRANCH_FUSED", 1) != 0 && d.ncls <= 16 && d.code <= 16 &&
                            36 + 4 * BG <= 64;
  if (fwd_transposes(cfg, d, C))
    RUN(launch_weight_transposes(Pm, Wm, P, W, d.NL, C, d.F, true, true, (hipStream_t)s, branch_fused ? BG : 0));
  // ---- side 2: input_proj (petr_head.py:390) + sine 3D (positional_encoding.py:58-100) + adapt_pos3d hidden ----
  {
    petr_gemm_args g = gemm0();   // NCHW view [C_in][HW] read as an M-contiguous operand -> token-major memory
    if (io->memory_in) {          // input_proj already applied by the producer (petr_hip.h): take its tokens as they are
      if (tok16) RUN(petr_cast_bf16(io->memory_in, mem16, d.BL * C, s2));
      else PETR_CHECK(hipMemcpyAsync(Wm + W.mem, io->memory_in, (size_t)d.BL * C * 4, hipMemcpyDeviceToDevice, (hipStream_t)s2) == hipSuccess,
                      PETR_ERR_LAUNCH, "head_fwd: copy of memory_in failed");
    } else {
      g.a = io->feats; g.lda = d.HW; g.a_kcontig = 0; g.a_bs0 = (long)d.Cin * d.HW;
      g.b = Wp(P.in_w); g.ldb = d.Cin; g.b_kcontig = 1;
      g.c = Wm + W.mem; g.ldc = C; g.c_bs0 = (long)d.HW * C; g.bias = Pm + P.in_b;
      g.M = d.HW; g.N = C; g.K = d.Cin; g.nb0 = V;
      if (bf16_km(g)) g.flags |= PETR_GEMM_BF16;
      if (tok16) g.flags |= wflag | PETR_GEMM_STORE_BF16;         // memory as bf16
      RUN(petr_gemm(&g, s2));
    }
    petr_sine3d_args b;
    memset(&b, 0, sizeof b);
    b.mask = kpm; b.dim_t = io->dim_t; b.out = Wm + W.sine; b.B = d.B; b.N = d.N; b.H = d.H; b.W = d.W; b.F = C / 2;
    b.normalize = 1; b.scale = 6.283185307179586f; b.eps = 1e-6f; b.offset = 0.f;
    RUN(petr_sine3d_fwd(&b, s2));
    g = gemm0();
    g.a = Wm + W.sine; g.lda = d.HW; g.a_kcontig = 0; g.a_bs0 = (long)(C * 3 / 2) * d.HW;
    g.b = Wp(P.ad_w1); g.ldb = C * 3 / 2; g.b_kcontig = 1;
    g.c = Wm + W.h2; g.ldc = 4 * C; g.c_bs0 = (long)d.HW * 4 * C; g.bias = Pm + P.ad_b1;
    g.M = d.HW; g.N = 4 * C; g.K = C * 3 / 2; g.nb0 = V; g.flags = PETR_GEMM_RELU;
    if (bf16_km(g)) g.flags |= PETR_GEMM_BF16;
    if (hid16) g.flags |= PETR_GEMM_BF16 | PETR_GEMM_STORE_BF16;      // bf16 hidden (same element indexing, 2-byte elements)
    g.flags |= wflag;
    RUN(petr_gemm(&g, s2));
    // adapt_pos3d's second conv (petr_head.py:400-402) right here, into its own buffer: it used to accumulate into pos_embed on
    // side 1 BEHIND the position encoder's second conv - 62 us at c5 in the middle of the chain the first cross-attention waits
    // for (coords3d -> conv -> conv -> [this] -> key = memory + pos -> K projection).  The two halves now meet in petr_add_rows2.
    // (PETR_ADAPT_PARALLEL=0, diagnostic builds: the old order)
    if (petr_tune("PETR_ADAPT_PARALLEL", 1) != 0) {
      g = lin_fwd(Wm + W.h2, Wp(P.ad_w2), Pm + P.ad_b2, Wm + W.pos2, d.BL, C, 4 * C);
      if (attn_bf16) g.flags |= PETR_GEMM_BF16;
      if (hid16) g.flags |= PETR_GEMM_A_BF16;
      g.flags |= wflag;
      RUN(petr_gemm(&g, s2));
    }
  }
  // ---- side 1: 3D position embedding (petr_head.py:286-334): coords3d, conv 3D->4C, ReLU, conv 4C->C ----
  {
    petr_coords3d_args a;
    memset(&a, 0, sizeof a);
    a.img2lidar = io->img2lidar; a.depth = io->depth; a.out = Wm + W.vol; a.cmask = nullptr;
    a.B = d.B; a.N = d.N; a.H = d.H; a.W = d.W; a.D = d.D; a.pad_h = cfg->pad_h; a.pad_w = cfg->pad_w;
    for (int i = 0; i < 6; ++i) a.range[i] = cfg->position_range[i];
    a.eps = 1e-5f;
    RUN(petr_coords3d_fwd(&a, s1));
    petr_gemm_args g = gemm0();
    g.a = Wm + W.vol; g.lda = d.HW; g.a_kcontig = 0; g.a_bs0 = (long)3 * d.D * d.HW;
    g.b = Wp(P.pe_w1); g.ldb = 3 * d.D; g.b_kcontig = 1;
    g.c = Wm + W.h1; g.ldc = 4 * C; g.c_bs0 = (long)d.HW * 4 * C; g.bias = Pm + P.pe_b1;
    g.M = d.HW; g.N = 4 * C; g.K = 3 * d.D; g.nb0 = V; g.flags = PETR_GEMM_RELU;
    if (bf16_km(g)) g.flags |= PETR_GEMM_BF16;
    if (hid16) g.flags |= PETR_GEMM_BF16 | PETR_GEMM_STORE_BF16;
    g.flags |= wflag;
    RUN(petr_gemm(&g, s1));
    g = lin_fwd(Wm + W.h1, Wp(P.pe_w2), Pm + P.pe_b2, Wm + (cfg->with_fpe ? W.pe1 : W.pos), d.BL, C, 4 * C);
    if (attn_bf16) g.flags |= PETR_GEMM_BF16;      // bf16 mode: the K-contiguous L-sized contractions run on bf16 MFMA
    if (hid16) g.flags |= PETR_GEMM_A_BF16;
    g.flags |= wflag;
    RUN(petr_gemm(&g, s1));
    // wait for side 2 (memory, sine hidden)
    if (ln.ctx) {
      hipEvent_t e = ln.next();
      (void)hipEventRecord(e, (hipStream_t)s2);
      (void)hipStreamWaitEvent((hipStream_t)s1, e, 0);
    }
    if (cfg->with_fpe) {
      // feature-guided PE (petrv2_head.py:464-466, SELayer :48-60): pos3d * sigmoid(expand(relu(reduce(x))))
      g = lin_fwd(Wm + W.mem, Wp(P.fpe_rw), Pm + P.fpe_rb, Wm + W.fpe_h, d.BL, C, C);
      g.flags = PETR_GEMM_RELU | (attn_bf16 ? PETR_GEMM_BF16 : 0) | wflag | (tok16 ? PETR_GEMM_A_BF16 : 0);
      RUN(petr_gemm(&g, s1));
      g = lin_fwd(Wm + W.fpe_h, Wp(P.fpe_ew), Pm + P.fpe_eb, Wm + W.fpe_u, d.BL, C, C);
      if (attn_bf16) g.flags |= PETR_GEMM_BF16;
      g.flags |= wflag;
      RUN(petr_gemm(&g, s1));
      RUN(petr_gate_fwd(Wm + W.pe1, Wm + W.fpe_u, Wm + W.pos, d.BL * C, s1));
    }
    // key = memory + key_pos (petr_transformer.py:343-344), once for all layers; key_pos = pos_embed + adapt_pos3d(sine)
    // (petr_head.py:400-402) is formed in the same pass (the sum stays in pos_embed)
    if (petr_tune("PETR_ADAPT_PARALLEL", 1) != 0) {
      if (tok16) RUN(petr_add_rows2_bf16(mem16, Wm + W.pos, Wm + W.pos2, mempos16, d.BL, C, s1));
      else RUN(petr_add_rows2(Wm + W.mem, Wm + W.pos, Wm + W.pos2, Wm + W.mempos, d.BL, C, s1));
    } else {
      g = lin_fwd(Wm + W.h2, Wp(P.ad_w2), Pm + P.ad_b2, Wm + W.pos, d.BL, C, 4 * C);
      if (attn_bf16) { g.flags = PETR_GEMM_BF16; g.r = Wm + W.pos; g.ldr = C; }   // same sum with pos as the residual operand
      else g.flags = PETR_GEMM_ACCUMULATE;
      if (hid16) g.flags |= PETR_GEMM_A_BF16;
      g.flags |= wflag;
      RUN(petr_gemm(&g, s1));
      if (tok16) RUN(petr_add_rows_bf16(mem16, Wm + W.pos, mempos16, d.BL, 0, C, s1));
      else RUN(petr_add_rows(Wm + W.mem, Wm + W.pos, Wm + W.mempos, d.BL, 0, C, s1));
    }
    // K_l = (mem+pos) Wk_l^T + bk_l for ALL layers: [B][NL][L][C]
    g = gemm0();
    g.a = Wm + W.mempos; g.lda = C; g.a_kcontig = 1; g.a_bs0 = d.L * C;
    g.b = Wp(P.lay[0].ca_in_w + (long)C * C); g.ldb = C; g.b_kcontig = 1; g.b_bs1 = P.ca_in_stride;
    g.bias = Pm + P.lay[0].ca_in_b + C; g.bias_bs1 = P.ca_in_stride;
    g.c = Wm + W.k_all; g.ldc = C; g.c_bs0 = (long)d.NL * d.L * C; g.c_bs1 = d.L * C;
    g.M = (int)d.L; g.N = C; g.K = C; g.nb0 = d.B; g.nb1 = d.NL;
    if (attn_bf16) { g.c = reinterpret_cast<float*>(k16); g.flags |= PETR_GEMM_STORE_BF16 | PETR_GEMM_BF16; }   // bf16 MFMA, bf16 store
    if (tok16) g.flags |= wflag | PETR_GEMM_A_BF16;
    RUN(kv_project(g, s1, &ev_k0));
  }
  {
    // V_l = mem Wv_l^T + bv_l on side 2 (memory was produced there)
    petr_gemm_args g = gemm0();
    g.a = Wm + W.mem; g.lda = C; g.a_kcontig = 1; g.a_bs0 = d.L * C;
    g.b = Wp(P.lay[0].ca_in_w + (long)2 * C * C); g.ldb = C; g.b_kcontig = 1; g.b_bs1 = P.ca_in_stride;
    g.bias = Pm + P.lay[0].ca_in_b + 2 * C; g.bias_bs1 = P.ca_in_stride;
    g.c = Wm + W.v_all; g.ldc = C; g.c_bs0 = (long)d.NL * d.L * C; g.c_bs1 = d.L * C;
    g.M = (int)d.L; g.N = C; g.K = C; g.nb0 = d.B; g.nb1 = d.NL;
    if (attn_bf16) { g.c = reinterpret_cast<float*>(v16); g.flags |= PETR_GEMM_STORE_BF16 | PETR_GEMM_BF16; }
    if (tok16) g.flags |= wflag | PETR_GEMM_A_BF16;
    RUN(kv_project(g, s2, &ev_v0));
  }

  // ---- main: query embedding pos2posemb3d + MLP (petr_head.py:422-423) ----
  RUN(petr_posemb3d_fwd(Pm + P.ref, io->dim_t, Wm + W.posemb, d.Q, C / 2, s));
  {
    petr_gemm_args g = lin_fwd(Wm + W.posemb, Pm + P.qe_w1, Pm + P.qe_b1, Wm + W.qe_h, d.Q, C, C * 3 / 2);
    g.flags = PETR_GEMM_RELU;
    RUN(petr_gemm(&g, s));
    g = lin_fwd(Wm + W.qe_h, Pm + P.qe_w2, Pm + P.qe_b2, Wm + W.qe, d.Q, C, C);
    RUN(petr_gemm(&g, s));
  }

  // ---- decoder (petr_transformer.py:95-107,440-446; layer op order A.3) ----
  RUN(petr_fill(Wm + W.x0, 0.f, d.BQ * C + W.mha_sched_n, s));       // target = zeros (:95) + attention tickets
  // Dynamic K/V-tile tickets are opt-in (PETR_MHA_DYNAMIC=1): they make the grouping of the partial sums, and so
  // the low-order bits of the forward, depend on timing; the default static ranges keep the forward bit-reproducible.
  static const bool mha_dynamic = petr_tune("PETR_MHA_DYNAMIC", 0) != 0;
  int* sched = mha_dynamic ? reinterpret_cast<int*>(Wm + W.x0 + d.BQ * C) : nullptr;
  RUN(petr_add_rows(Wm + W.x0, E, Wm + W.lay[0].xe_in, d.BQ, d.Q, C, s));
  const float* x_in = Wm + W.x0;
  float* mws = Wm + W.mha_ws;
  // Training mode (io->dropout_p > 0): the six dropout layers of a decoder layer (petr_hip.h "Dropout"; sites
  // 8*l + 0..5).  The residual dropouts act on the sub-layer output BEFORE the identity is added
  // (petr_transformer.py:367), so in this mode the out-projection / second FFN contraction store the bare
  // sub-layer output and the LayerNorm prologue does drop(out) + identity; z (the LayerNorm input the backward
  // needs) is written back over the same buffer.
  const bool training = io->dropout_p > 0.f;
  PETR_CHECK(io->dropout_p >= 0.f && io->dropout_p < 1.f, PETR_ERR_INVALID, "head_fwd: dropout_p=%g outside [0,1)",
             (double)io->dropout_p);
  auto site = [&](int l, int k) {
    petr_dropout dr;
    dr.seed = io->dropout_seed; dr.site = (uint32_t)(8 * l + k); dr.p = io->dropout_p;
    return dr;
  };
  // Training mode: the attention kernels leave the dropout mask they applied as packed key-major bits; the backward tests
  // one bit per probability instead of hashing it again (PETR_DROP_BITS=0: re-hash).
  static const bool drop_bits_env = env_on("PETR_DROP_BITS");
  const bool use_bits = training && drop_bits_env;
  uint32_t* bits0 = reinterpret_cast<uint32_t*>(Wm + W.bits);
  auto bits_ptr = [&](int l, int self) -> uint32_t* {
    return bits0 + (long)l * (W.bits_cross_n + W.bits_self_n) + (self ? W.bits_cross_n : 0);
  };
  // PETR_FUSE_OUT_LN=0: the four separate launches (merge, out-projection, LayerNorm) per attention
  static const bool fuse_env = env_on("PETR_FUSE_OUT_LN");
  const bool fuse_out = fuse_env && C == 256 && d.NH == 8;
  const int ns_self = petr_mha_choose_split(d.B, d.NH, d.Q, d.Q);
  // (PETR_SELF_BF16=0, diagnostic builds: the self-attention stays on the fp32 kernels in bf16 mode)
  const bool self16 = attn_bf16 && fuse_out && petr_tune("PETR_SELF_BF16", 1) != 0;
  const int ns_cross = attn_bf16 ? petr_mha_fwd_bf16_choose_split(d.B, d.NH, d.Q, (int)d.L) : petr_mha_choose_split(d.B, d.NH, d.Q, (int)d.L);
  for (int l = 0; l < d.NL; ++l) {
    const LayerP& lp = P.lay[l];
    const LayerW& lw = W.lay[l];
    const petr_dropout dr_sp = site(l, 0), dr_so = site(l, 1), dr_cp = site(l, 2), dr_co = site(l, 3), dr_fh = site(l, 4),
                       dr_fo = site(l, 5);
    // self-attention: q = k = x + query_pos, v = x  (multi_atten_decoder_layer.py:223-237)
    petr_gemm_args g = lin_fwd(x_in, Pm + lp.sa_in_w, Pm + lp.sa_in_b, Wm + lw.qkv, d.BQ, 3 * C, C);
    g.a2 = E; g.a2_rows = d.Q; g.a2_ncols = 2 * C;
    if (l == 0 || !fuse_out) RUN(petr_gemm(&g, s));       // layers 1..: done by the previous layer's closing petr_ln_proj
    if (fuse_out) {
      // attention with its L-split partials left in the workspace, then ONE launch: merge + out-projection + dropout +
      // identity (petr_transformer.py:367) + LayerNorm + query_pos add
      if (self16) {     // bf16 mode: the self-attention on the bf16 kernels too (K / V rows of qkv rounded once; q, output, LSE fp32)
        uint16_t* q16 = reinterpret_cast<uint16_t*>(Wm + lw.qkv16);
        if (l == 0) RUN(petr_cast_bf16(Wm + lw.qkv, q16, d.BQ * 3 * C, s));      // layers 1..: written by the previous layer's petr_ln_proj
        RUN(mha_f_bf16(Wm + lw.qkv, (long)d.Q * 3 * C, 3 * C, q16 + C, (long)d.Q * 3 * C, 3 * C, q16 + 2 * C, Wm + lw.ao_s,
                       Wm + lw.lse_s, nullptr, d, d.Q, mws, W.mha_ws_bytes, s, training ? &dr_sp : nullptr,
                       use_bits ? bits_ptr(l, 1) : nullptr, ns_self, 1));
      } else
      RUN(mha_f(Wm + lw.qkv, (long)d.Q * 3 * C, 3 * C, Wm + lw.qkv + C, (long)d.Q * 3 * C, 3 * C, Wm + lw.qkv + 2 * C,
                Wm + lw.ao_s, Wm + lw.lse_s, nullptr, d, d.Q, mws, W.mha_ws_bytes, sched, s, training ? &dr_sp : nullptr,
                use_bits ? bits_ptr(l, 1) : nullptr, ns_self, 1));
      RUN(attn_out_ln(Wm + lw.ao_s, mws, ns_self, d, training ? hidden_drop_scale(dr_sp) : 1.f, Wm + lw.lse_s, Wm + W.wt[l].sa_out,
                      Pm + lp.sa_out_b, x_in, training ? &dr_so : nullptr, Pm + lp.n_g[0], Pm + lp.n_b[0], Wm + lw.z0,
                      Wm + lw.mean0, Wm + lw.rstd0, Wm + lw.x1, Wm + lw.xe1, E, d.Q, s,
                      Wm + W.wt[l].ca_q, Pm + lp.ca_in_b, Wm + lw.qc));    // + the cross-attention's query projection
    } else {
    RUN(mha_f(Wm + lw.qkv, (long)d.Q * 3 * C, 3 * C, Wm + lw.qkv + C, (long)d.Q * 3 * C, 3 * C, Wm + lw.qkv + 2 * C,
              Wm + lw.ao_s, Wm + lw.lse_s, nullptr, d, d.Q, mws, W.mha_ws_bytes, sched, s, training ? &dr_sp : nullptr,
              use_bits ? bits_ptr(l, 1) : nullptr));
    g = lin_fwd(Wm + lw.ao_s, Pm + lp.sa_out_w, Pm + lp.sa_out_b, Wm + lw.z0, d.BQ, C, C);
    if (!training) { g.r = x_in; g.ldr = C; }                          // identity + out (petr_transformer.py:367)
    RUN(petr_gemm(&g, s));
    RUN(ln_fwd(Wm + lw.z0, 1, 0, nullptr, training ? x_in : nullptr, Pm + lp.n_g[0], Pm + lp.n_b[0], Wm + lw.x1,
               training ? Wm + lw.z0 : nullptr, Wm + lw.mean0, Wm + lw.rstd0, d.BQ, C, 0, Wm + lw.xe1, E, d.Q, s,
               training ? &dr_so : nullptr));
    }
    // cross-attention: q = x1 + query_pos, k = mem + pos, v = mem (petr_transformer.py:341-362)
    if (!fuse_out) {
      g = lin_fwd(Wm + lw.xe1, Pm + lp.ca_in_w, Pm + lp.ca_in_b, Wm + lw.qc, d.BQ, C, C);
      RUN(petr_gemm(&g, s));
    }
    if (l == 0) {          // K/V come from the side streams: layer 0's now, the other layers' before layer 1
      if (ev_k0) {
        (void)hipStreamWaitEvent(ln.main, ev_k0, 0);
        (void)hipStreamWaitEvent(ln.main, ev_v0, 0);
      } else {
        ln.join(0);
        ln.join(1);
      }
    } else if (l == 1 && ev_k0) {
      ln.join(0);
      ln.join(1);
    }
    if (attn_bf16)
      RUN(mha_f_bf16(Wm + lw.qc, (long)d.Q * C, C, k16 + (long)l * d.L * C, (long)d.NL * d.L * C, C,
                     v16 + (long)l * d.L * C, Wm + lw.ao_c, Wm + lw.lse_c, kpm, d, (int)d.L, mws, W.mha_ws_bytes, s,
                     training ? &dr_cp : nullptr, use_bits ? bits_ptr(l, 0) : nullptr, fuse_out ? ns_cross : 0, fuse_out ? 1 : 0));
    else
    RUN(mha_f(Wm + lw.qc, (long)d.Q * C, C, Wm + W.k_all + (long)l * d.L * C, (long)d.NL * d.L * C, C,
              Wm + W.v_all + (long)l * d.L * C, Wm + lw.ao_c, Wm + lw.lse_c, kpm, d, (int)d.L, mws, W.mha_ws_bytes, sched, s,
              training ? &dr_cp : nullptr, use_bits ? bits_ptr(l, 0) : nullptr, fuse_out ? ns_cross : 0, fuse_out ? 1 : 0));
    if (fuse_out) {
      RUN(attn_out_ln(Wm + lw.ao_c, mws, ns_cross, d, training ? hidden_drop_scale(dr_cp) : 1.f, Wm + lw.lse_c, Wm + W.wt[l].ca_out,
                      Pm + lp.ca_out_b, Wm + lw.x1, training ? &dr_co : nullptr, Pm + lp.n_g[1], Pm + lp.n_b[1], Wm + lw.z1,
                      Wm + lw.mean1, Wm + lw.rstd1, Wm + lw.x2, nullptr, nullptr, 0, s));
    } else {
    g = lin_fwd(Wm + lw.ao_c, Pm + lp.ca_out_w, Pm + lp.ca_out_b, Wm + lw.z1, d.BQ, C, C);
    if (!training) { g.r = Wm + lw.x1; g.ldr = C; }
    RUN(petr_gemm(&g, s));
    RUN(ln_fwd(Wm + lw.z1, 1, 0, nullptr, training ? Wm + lw.x1 : nullptr, Pm + lp.n_g[1], Pm + lp.n_b[1], Wm + lw.x2,
               training ? Wm + lw.z1 : nullptr, Wm + lw.mean1, Wm + lw.rstd1, d.BQ, C, 0, nullptr, nullptr, 0, s,
               training ? &dr_co : nullptr));
    }
    // FFN (mmcv FFN): x + W2 relu(W1 x + b1) + b2 ; second contraction split over K
    // one launch for both contractions (petr_ffn_fwd, fp32).  PETR_FFN16=1 (bf16 mode only): two contractions on the bf16 matrix
    // cores with the bf16 weight copy instead, as autocast runs them (11.8 us each alone, but the pair, its launch gap and the
    // bf16 input gradients measured 1.4-1.7 % of a step slower than the fused fp32 kernels)
    float* xs_l = Wm + W.xs + (long)l * d.BQ * C;
    const int n_slabs = ffn_fused ? W.ffn_fsplit : W.ffn_split;
    const bool slabs = ffn_fused || !(W.ffn_split == 1 && !training);   // false: the second contraction wrote z2 itself
    if (ffn_fused) {
      petr_ffn_fwd_args f;
      memset(&f, 0, sizeof f);
      f.x = Wm + lw.x2; f.w1t = Wm + W.wt[l].f1; f.b1 = Pm + lp.f1_b; f.w2t = Wm + W.wt[l].f2;
      f.hidden = Wm + lw.hff; f.part = Wm + W.ffn_part; f.part_stride = d.BQ * C;
      if (training) f.drop = dr_fh;                                     // Linear, ReLU, Dropout (mmcv FFN)
      f.M = (int)d.BQ; f.F = (int)d.F; f.n_split = n_slabs;
      RUN(petr_ffn_fwd(&f, s));
    } else {
    g = lin_fwd(Wm + lw.x2, ffn16 ? Wp(lp.f1_w) : Pm + lp.f1_w, Pm + lp.f1_b, Wm + lw.hff, d.BQ, d.F, C);
    g.flags = PETR_GEMM_RELU | (ffn16 ? wflag : 0);
    if (training) g.drop = dr_fh;                                       // Linear, ReLU, Dropout (mmcv FFN)
    RUN(petr_gemm(&g, s));
    g = lin_fwd(Wm + lw.hff, ffn16 ? Wp(lp.f2_w) : Pm + lp.f2_w, nullptr, Wm + W.ffn_part, d.BQ, C, d.F);
    if (ffn16) g.flags |= wflag;
    g.split_k = W.ffn_split; g.c_split_stride = d.BQ * C;
    if (!slabs) { g.bias = Pm + lp.f2_b; g.r = Wm + lw.x2; g.ldr = C; g.c = Wm + lw.z2; }
    RUN(petr_gemm(&g, s));
    }
    float* xe_next = l + 1 < d.NL ? Wm + W.lay[l + 1].xe_in : nullptr;
    if (fuse_out && l + 1 < d.NL) {
      // closing LayerNorm (slab sum + bias + dropout + residual + norm + query_pos add) AND the next layer's self-attention
      // in-projection in one launch
      petr_ln_proj_args q;
      memset(&q, 0, sizeof q);
      q.x = slabs ? Wm + W.ffn_part : Wm + lw.z2; q.n_partials = slabs ? n_slabs : 1; q.partial_stride = d.BQ * C;
      q.bias = slabs ? Pm + lp.f2_b : nullptr; q.residual = slabs ? Wm + lw.x2 : nullptr;
      if (slabs && training) q.drop = dr_fo;
      q.gamma = Pm + lp.n_g[2]; q.beta = Pm + lp.n_b[2]; q.eps = 1e-5f;
      q.z = slabs ? Wm + lw.z2 : nullptr; q.mean = Wm + lw.mean2; q.rstd = Wm + lw.rstd2; q.y = xs_l;
      q.y2 = xe_next; q.add2 = E; q.add2_rows = d.Q; q.M = (int)d.BQ;
      q.w2T = Wm + W.wt[l + 1].sa_in; q.bias2 = Pm + P.lay[l + 1].sa_in_b; q.out2 = Wm + W.lay[l + 1].qkv; q.n2 = 3; q.n2_pos = 2;
      if (self16) q.out2_bf16 = reinterpret_cast<uint16_t*>(Wm + W.lay[l + 1].qkv16);      // the bf16 self-attention's K / V copy
      RUN(petr_ln_proj(&q, s));
    } else if (!slabs) {
      RUN(ln_fwd(Wm + lw.z2, 1, 0, nullptr, nullptr, Pm + lp.n_g[2], Pm + lp.n_b[2], xs_l, nullptr, Wm + lw.mean2,
                 Wm + lw.rstd2, d.BQ, C, 0, xe_next, E, d.Q, s));
    } else {
      RUN(ln_fwd(Wm + W.ffn_part, n_slabs, d.BQ * C, Pm + lp.f2_b, Wm + lw.x2, Pm + lp.n_g[2], Pm + lp.n_b[2], xs_l,
                 Wm + lw.z2, Wm + lw.mean2, Wm + lw.rstd2, d.BQ, C, 0, xe_next, E, d.Q, s, training ? &dr_fo : nullptr));
    }
    x_in = xs_l;
  }
  // post_norm of every intermediate (petr_transformer.py:444, one shared LN) + nan_to_num (petr_head.py:435)
  RUN(ln_fwd(Wm + W.xs, 1, 0, nullptr, nullptr, Pm + P.post_g, Pm + P.post_b, Wm + W.outs, nullptr, Wm + W.mean_p,
             Wm + W.rstd_p, d.R, C, PETR_LN_NAN_TO_NUM, nullptr, nullptr, 0, s));

  // ---- branches (petr_head.py:226-247,440-460 / petrv2_head.py:294-307,513-531); reg chain on a side stream ----
  // PETRHead: ONE cls / reg module for all levels (G = 1 group of R rows); PETRv2Head: deep copies (G = NL groups).
  const int G = cfg->shared_branches ? 1 : d.NL;
  const long RG = d.R / G;
  auto grouped = [&](petr_gemm_args g, long a_row, long c_row) {
    g.nb0 = G; g.M = (int)RG;
    g.a_bs0 = RG * a_row; g.b_bs0 = P.br_stride; g.bias_bs0 = g.bias ? P.br_stride : 0; g.c_bs0 = RG * c_row;
    return g;
  };
  ln.fork(0);
  petr_branch_fwd_args br;
  memset(&br, 0, sizeof br);
  br.x = Wm + W.outs; br.param_gs = P.br_stride; br.wt_gs = (long)4 * C * C; br.rows = (int)RG; br.groups = G; br.eps = 1e-5f;
  {
    petr_gemm_args g;
    if (branch_fused) {       // (Linear, ReLU) x 2 [+ Linear]: r1, r2 are what the backward reads
      petr_branch_fwd_args b = br;
      b.w1t = Wm + W.br_t + (long)2 * C * C; b.b1 = Pm + P.reg_b[0];
      b.w2t = Wm + W.br_t + (long)3 * C * C; b.b2 = Pm + P.reg_b[1];
      b.y1 = Wm + W.r1; b.y2 = Wm + W.r2;
      if (!cfg->with_multi) { b.w3 = Pm + P.reg_w[2]; b.b3 = Pm + P.reg_b[2]; b.out = Wm + W.reg_raw; b.n_out = d.code; }
      RUN(petr_branch_fwd(&b, s1));
    } else {
      g = grouped(lin_fwd(Wm + W.outs, Pm + P.reg_w[0], Pm + P.reg_b[0], Wm + W.r1, RG, C, C), C, C);
      g.flags = PETR_GEMM_RELU;
      RUN(petr_gemm(&g, s1));
      g = grouped(lin_fwd(Wm + W.r1, Pm + P.reg_w[1], Pm + P.reg_b[1], Wm + W.r2, RG, C, C), C, C);
      g.flags = PETR_GEMM_RELU;
      RUN(petr_gemm(&g, s1));
    }
    if (!cfg->with_multi) {
      if (!branch_fused) {
        g = grouped(lin_fwd(Wm + W.r2, Pm + P.reg_w[2], Pm + P.reg_b[2], Wm + W.reg_raw, RG, d.code, C), C, d.code);
        RUN(petr_gemm(&g, s1));
      }
    } else {
      // RegLayer task heads (petrv2_head.py:81-95): 5 x (Linear, ReLU, Linear -> (2,1,3,2,2)), concatenated
      g = grouped(lin_fwd(Wm + W.r2, Pm + P.th_w1, Pm + P.th_b1, Wm + W.th_h, RG, C, C), C, 5 * C);
      g.nb1 = 5; g.b_bs1 = P.th_stride; g.bias_bs1 = P.th_stride; g.c_bs1 = RG * C;
      g.flags = PETR_GEMM_RELU;
      RUN(petr_gemm(&g, s1));
      static const int TH_DIMS[5] = {2, 1, 3, 2, 2}, TH_COL[5] = {0, 2, 3, 6, 8};
      if (C == 256 && petr_tune("PETR_TASK_HEADS_FUSED", 1) != 0) {       // the five 1-3 column Linears in one launch
        petr_task_heads_fwd_args ta;
        memset(&ta, 0, sizeof ta);
        ta.h = Wm + W.th_h; ta.w2 = Pm + P.th_w2; ta.b2 = Pm + P.th_b2; ta.param_gs = P.br_stride; ta.head_stride = P.th_stride;
        ta.out = Wm + W.reg_raw; ta.ld_out = d.code; ta.rows = (int)RG; ta.groups = G; ta.heads = 5;
        for (int t = 0; t < 5; ++t) { ta.dims[t] = TH_DIMS[t]; ta.cols[t] = TH_COL[t]; }
        RUN(petr_task_heads_fwd(&ta, s1));
      } else
      for (int t = 0; t < 5; ++t) {
        g = grouped(lin_fwd(Wm + W.th_h + (long)t * RG * C, Pm + P.th_w2 + t * P.th_stride, Pm + P.th_b2 + t * P.th_stride,
                            Wm + W.reg_raw + TH_COL[t], RG, TH_DIMS[t], C), 5 * C, d.code);
        g.ldc = d.code;
        RUN(petr_gemm(&g, s1));
      }
    }
    petr_bbox_args a;
    memset(&a, 0, sizeof a);
    a.reg = Wm + W.reg_raw; a.ref = Pm + P.ref; a.out = io->all_bbox_preds; a.rows = (int)d.R; a.Q = d.Q; a.code = d.code;
    for (int i = 0; i < 6; ++i) a.pc_range[i] = cfg->pc_range[i];
    a.time_div = cfg->with_time ? io->time_div : 0.f; a.eps = 1e-5f;
    RUN(petr_bbox_epilogue_fwd(&a, s1));
  }
  if (branch_fused) {         // (Linear, LayerNorm, ReLU) x 2 + Linear: the class scores
    petr_branch_fwd_args b = br;
    b.w1t = Wm + W.br_t; b.b1 = Pm + P.cls_b[0]; b.g1 = Pm + P.cls_g[0]; b.be1 = Pm + P.cls_be[0];
    b.w2t = Wm + W.br_t + (long)C * C; b.b2 = Pm + P.cls_b[1]; b.g2 = Pm + P.cls_g[1]; b.be2 = Pm + P.cls_be[1];
    b.w3 = Pm + P.cls_w[2]; b.b3 = Pm + P.cls_b[2]; b.out = io->all_cls_scores; b.n_out = d.ncls;
    b.h1 = Wm + W.c1; b.y1 = Wm + W.c1n; b.h2 = Wm + W.c2; b.y2 = Wm + W.c2n;
    b.mean1 = Wm + W.c1_mean; b.rstd1 = Wm + W.c1_rstd; b.mean2 = Wm + W.c2_mean; b.rstd2 = Wm + W.c2_rstd;
    RUN(petr_branch_fwd(&b, s));
  } else {
    petr_gemm_args g = grouped(lin_fwd(Wm + W.outs, Pm + P.cls_w[0], Pm + P.cls_b[0], Wm + W.c1, RG, C, C), C, C);
    RUN(petr_gemm(&g, s));
    for (int gi = 0; gi < G; ++gi)
      RUN(ln_fwd(Wm + W.c1 + gi * RG * C, 1, 0, nullptr, nullptr, Pm + P.cls_g[0] + gi * P.br_stride,
                 Pm + P.cls_be[0] + gi * P.br_stride, Wm + W.c1n + gi * RG * C, nullptr, Wm + W.c1_mean + gi * RG,
                 Wm + W.c1_rstd + gi * RG, RG, C, PETR_LN_RELU, nullptr, nullptr, 0, s));
    g = grouped(lin_fwd(Wm + W.c1n, Pm + P.cls_w[1], Pm + P.cls_b[1], Wm + W.c2, RG, C, C), C, C);
    RUN(petr_gemm(&g, s));
    for (int gi = 0; gi < G; ++gi)
      RUN(ln_fwd(Wm + W.c2 + gi * RG * C, 1, 0, nullptr, nullptr, Pm + P.cls_g[1] + gi * P.br_stride,
                 Pm + P.cls_be[1] + gi * P.br_stride, Wm + W.c2n + gi * RG * C, nullptr, Wm + W.c2_mean + gi * RG,
                 Wm + W.c2_rstd + gi * RG, RG, C, PETR_LN_RELU, nullptr, nullptr, 0, s));
    g = grouped(lin_fwd(Wm + W.c2n, Pm + P.cls_w[2], Pm + P.cls_b[2], io->all_cls_scores, RG, d.ncls, C), C, d.ncls);
    RUN(petr_gemm(&g, s));
  }
  ln.join(0);
  return PETR_OK;
}

// =============================================================================================
// backward
// =============================================================================================
extern "C" int petr_head_bwd(const petr_head_config* cfg, const petr_head_io* io, const petr_head_grads* gr,
                             int stage_begin, int stage_end, void* stream) {
  RUN(check_config(cfg));
  PETR_CHECK(!(io && io->memory_in), PETR_ERR_UNSUPPORTED, "head_bwd: the forward ran on memory_in (folded input_proj): inference only");
  PETR_CHECK(io && io->params && io->feats && io->ws && io->all_bbox_preds && gr && gr->d_cls && gr->d_bbox && gr->d_params,
             PETR_ERR_INVALID, "head_bwd: null pointer");
  const Dims d = make_dims(cfg);
  POff P;
  build_layout(cfg, &P, nullptr);
  WOff W;
  build_ws(cfg, &W, nullptr);
  PETR_CHECK(io->ws_bytes >= (size_t)W.total * sizeof(float), PETR_ERR_WORKSPACE, "head_bwd: workspace too small");
  PETR_CHECK(stage_begin >= 0 && stage_end <= P.n_stages && stage_begin < stage_end, PETR_ERR_INVALID,
             "head_bwd: bad stage range [%d,%d)", stage_begin, stage_end);
  const float* Pm = io->params;
  float* Gp = gr->d_params;
  float* Wm = (float*)io->ws;
  const int C = d.C;
  const uint8_t* kpm = cfg->has_mask ? io->mask : nullptr;
  float* mws = Wm + W.mha_ws;
  const Lanes ln{(hipStream_t)stream, (petr_ctx*)io->ctx};
  void* s = ln.m();
  // Weight-gradient contractions only feed d_params: they leave the critical path (the chain of input-gradient
  // kernels) and alternate over the side streams.  Every dy they read lives in private scratch (WOff::lg / s0_*).
  // Inside the box / decoder-layer stages they are only QUEUED and issued together at the end of the stage, behind one
  // fork: an event record is a barrier packet in the main queue (~7 us of bubble each, measured in the kernel trace), and
  // there would be one per contraction - seven per decoder layer.  (PETR_WGRAD_DEFER=0: fork per contraction.)
  static const bool defer_env = env_on("PETR_WGRAD_DEFER");
  static const bool dgrad_t = env_on("PETR_DGRAD_T");        // PETR_DGRAD_T=0: input gradients read W itself (K-major operand)
  hipEvent_t ev_tr = nullptr;
  // LayerNorm backward + the following input gradient in one launch (PETR_FUSE_LN_BWD=0: separate)
  static const bool fuse_bwd_env = env_on("PETR_FUSE_LN_BWD");
  const bool fuse_bwd = fuse_bwd_env && C == 256;
  int wg_rr = 0, n_pend = 0;
  bool defer = false;
  petr_gemm_args pend[24];
  auto flush_wgrads = [&]() -> int {
    if (!n_pend) return PETR_OK;
#ifdef PETR_DIAG_SKIP_WGRAD      // diagnostic build only: the main chain's time without any weight-gradient work beside it
    n_pend = 0;
    return PETR_OK;
#endif
    ln.fork_first(2);
    // every 900-row parameter gradient queued so far goes into ONE grouped launch (petr_wgrad_grouped: a workgroup owns an
    // output tile for the whole K range, no atomics) on side stream 0; what does not fit its layout rules (the 10-wide class
    // logits, bf16-routed token contractions, the forward-shaped query_pos slabs) stays a petr_gemm launch, on side stream 1 first
    static const bool grouped_on = env_on("PETR_WGRAD_GROUPED");
    petr_wgrad_item grp[PETR_WGRAD_MAX];
    bool taken[24];
    int ng = 0;
    for (int i = 0; i < n_pend; ++i) {
      const petr_gemm_args& g = pend[i];
      taken[i] = grouped_on && ng < PETR_WGRAD_MAX && !g.a_kcontig && !g.b_kcontig && g.flags == PETR_GEMM_ATOMIC && g.nb0 == 1 &&
                 g.nb1 == 1 && g.k_seg <= 0 && !g.a2 && !g.bias && !g.r && !(g.M & 63) && !(g.N & 63) && !(g.lda & 3) && !(g.ldb & 3) &&
                 aligned16(g.a) && aligned16(g.b);
      if (!taken[i]) continue;
      petr_wgrad_item& w = grp[ng++];
      w.dy = g.a; w.lda = g.lda; w.x = g.b; w.ldb = g.ldb; w.dw = g.c; w.ldc = g.ldc; w.db = g.a_colsum;
      w.M = g.M; w.N = g.N; w.K = g.K;
      // token-sized K (the final stage's K / V projection and position-embedding weights): slices of >= 16 K steps until the
      // item has ~512 workgroups; 900-row items keep one owner per tile
      const long tiles = (long)(g.M / 64) * (g.N / 64);
      long ks = g.K > 4096 ? cdiv(512, tiles) : 1;
      const long max_ks = cdiv(cdiv(g.K, 32), 16);
      w.ksplit = (int)(ks > max_ks ? max_ks : ks < 1 ? 1 : ks);
    }
    if (ng) RUN(petr_wgrad_grouped(grp, ng, ln.side(0)));
    for (int i = 0; i < n_pend; ++i)
      if (!taken[i]) RUN(petr_gemm(&pend[i], ln.side(++wg_rr & 1)));
    n_pend = 0;
    return PETR_OK;
  };
  auto wgrad = [&](petr_gemm_args g) -> int {
    if (n_pend == 24) RUN(flush_wgrads());
    pend[n_pend++] = g;
    return defer ? PETR_OK : flush_wgrads();
  };
  // io->attn_bf16: the token-sized gradient contractions (K/V projections, position-embedding MLPs, input_proj,
  // PETRv2's feature-guided PE) and the cross-attention backward run on the bf16 matrix cores, like their forwards
  const bool bf16 = io->attn_bf16 != 0;
  // the forward's condition for the bf16 self-attention (head_fwd: self16)
  const bool slab_tiled = petr_tune("PETR_SLAB_TILED", bf16 ? 1 : 0) != 0;      // query_pos slab contractions: see the layer stages
  const bool self16_b = bf16 && env_on("PETR_FUSE_OUT_LN") && C == 256 && d.NH == 8 && petr_tune("PETR_SELF_BF16", 1) != 0;
  const bool hid16 = hidden_bf16(io);
  const bool tok16 = token_bf16(cfg, io);
  const uint16_t* p16 = reinterpret_cast<const uint16_t*>(Wm + W.p16);        // made by the forward (same parameters)
  auto Wp = [&](long off) -> const float* { return tok16 ? reinterpret_cast<const float*>(p16 + off) : Pm + off; };
  const int wflag = tok16 ? PETR_GEMM_B_BF16 : 0;
  static const bool ffn16_env = petr_tune("PETR_FFN16", 0) != 0;   // opt-in: the fused fp32 FFN kernels measured faster
  const bool ffn16 = tok16 && ffn16_env;
  static const bool drop_bits_env = env_on("PETR_DROP_BITS");          // the forward generated them (same workspace)
  const bool use_bits = io->dropout_p > 0.f && drop_bits_env;
  const uint32_t* bits0 = reinterpret_cast<const uint32_t*>(Wm + W.bits);
  auto bits_ptr = [&](int l, int self) -> const uint32_t* {
    return bits0 + (long)l * (W.bits_cross_n + W.bits_self_n) + (self ? W.bits_cross_n : 0);
  };
  auto L16 = [&](const petr_gemm_args& g) { return bf16 ? to_bf16(g) : g; };
  // PETR_KV_BWD_OVERLAP=1 (opt-in): the K/V projection backward per layer on the side streams beside the decoder chain
  // instead of two batched contractions in the final stage.  Measured and rejected as a default (same-box A/B,
  // scripts/ab_overlap.sh, two rounds): c5 fp32 5.09 -> 5.21 ms, p4-1600 bf16 6.98 -> 7.52 ms, p4-1600 fp32 neutral - the
  // main queue is ~85 % busy already, so token-sized work beside it takes CUs from the critical chain.
  static const bool kv_overlap_env = petr_tune("PETR_KV_BWD_OVERLAP", 0) != 0;
  const bool kv_overlap = kv_overlap_env && ln.ctx != nullptr;
  const uint16_t* k16 = reinterpret_cast<const uint16_t*>(Wm + W.k_all);
  const uint16_t* v16 = reinterpret_cast<const uint16_t*>(Wm + W.v_all);
  // bf16 mode keeps dK / dV of all layers as bf16 in the front half of their fp32 buffers (same element indexing)
  const bool dkv16 = dkv_bf16(io);
  uint16_t* dk16 = reinterpret_cast<uint16_t*>(Wm + W.dk_all);
  uint16_t* dv16 = reinterpret_cast<uint16_t*>(Wm + W.dv_all);
  auto dkv_ptr = [&](int kv, long off) -> const float* {      // dK (kv = 0) / dV (1) at element offset off, either storage
    if (dkv16) return reinterpret_cast<const float*>((kv == 0 ? dk16 : dv16) + off);
    return Wm + (kv == 0 ? W.dk_all : W.dv_all) + off;
  };
  const int dkv_flag = dkv16 ? PETR_GEMM_A_BF16 : 0;

  for (int stage = stage_begin; stage < stage_end; ++stage) {
    // (stage 0 always defers: its cls-branch gradients are produced on a side stream and only joined at the end)
    defer = stage == 0 || (defer_env && stage <= d.NL);      // the final stage's token-sized weight gradients start as soon as they can
    if (stage == 0) {
      // clear every += / atomic target of this backward pass; in bf16 mode dK / dV of all layers (the bulk of the range:
      // 2 x 6 x L x 256 floats) are stored by petr_mha_bwd_bf16 and need no zero-fill
      // Only d_ref_tmp is touched by this stage: with side streams the bulk (d_qc, d_qkv and, in fp32 mode, dK / dV of all layers:
      // 52 MB at c5, 295 MB at p4-1600) is cleared on side stream 1 beside the branch backward - the layer stages wait for that
      // stream's event anyway (transposed weights) - instead of in front of it on the main stream (PETR_BWD_ZERO_SIDE=0: main)
      hipError_t e;
      const bool zero_side = dgrad_t && ln.ctx && petr_tune("PETR_BWD_ZERO_SIDE", 1) != 0;
      const long bulk_end = io->attn_bf16 ? W.dk_all : W.d_ref_tmp;
      if (dgrad_t) ln.fork(1);
      e = hipMemsetAsync(Wm + W.zero_begin, 0, (size_t)(bulk_end - W.zero_begin) * sizeof(float),
                         zero_side ? (hipStream_t)ln.side(1) : ln.main);
      if (e == hipSuccess)
        e = hipMemsetAsync(Wm + W.d_ref_tmp, 0, (size_t)(W.zero_end - W.d_ref_tmp) * sizeof(float), ln.main);
      PETR_CHECK(e == hipSuccess, PETR_ERR_LAUNCH, "head_bwd: memset failed: %s", hipGetErrorString(e));
      // transposed decoder weights for the layer stages' input gradients: side stream 1, beside the branch backward
      if (dgrad_t) {
        // nothing to do when the forward made them (same parameters, same workspace)
        const bool have = fwd_transposes(cfg, d, C);
        RUN(launch_weight_transposes(Pm, Wm, P, W, d.NL, C, d.F, !have, !have, (hipStream_t)ln.side(1)));
        if (ln.ctx) {
          ev_tr = ln.next();
          (void)hipEventRecord(ev_tr, (hipStream_t)ln.side(1));
        }
      }
      // ---- box epilogue + reg branch ----
      const int G = cfg->shared_branches ? 1 : d.NL;
      const long RG = d.R / G;
      // per-group batching of the branch contractions (G = 1: PETRHead's shared module)
      auto gd = [&](petr_gemm_args g, long a_row, long c_row, long r_row) {   // input gradient / forward-like
        g.nb0 = G; g.M = (int)RG;
        g.a_bs0 = RG * a_row; g.b_bs0 = P.br_stride; g.c_bs0 = RG * c_row; g.r_bs0 = RG * r_row;
        return g;
      };
      auto gw = [&](petr_gemm_args g, long dy_row, long x_row) {               // weight gradient
        g.nb0 = G; g.K = (int)RG;
        g.a_bs0 = RG * dy_row; g.b_bs0 = RG * x_row; g.c_bs0 = P.br_stride; g.cs_bs0 = P.br_stride;
        const long tiles = cdiv(g.M, 64) * cdiv(g.N, 64) * G;
        long sk = 512 / (tiles > 0 ? tiles : 1);
        const long ktiles = cdiv(RG, 32);
        if (sk > ktiles / 2) sk = ktiles / 2;
        if (sk < 1) sk = 1;
        if (sk > 64) sk = 64;
        g.split_k = (int)sk;
        return g;
      };
      petr_bbox_args a;
      memset(&a, 0, sizeof a);
      a.ref = Pm + P.ref; a.out = io->all_bbox_preds; a.rows = (int)d.R; a.Q = d.Q; a.code = d.code;
      for (int i = 0; i < 6; ++i) a.pc_range[i] = cfg->pc_range[i];
      a.time_div = cfg->with_time ? io->time_div : 0.f; a.eps = 1e-5f;
      float* d_raw = Wm + W.s0_raw;   // [R, code]
      // the cls branch below runs on side stream 0 BESIDE the reg branch: its fork is recorded here, in front of the reg
      // branch's kernels (recorded behind them it made the cls branch wait for the whole reg branch: 65 us of the c5 step)
      ln.fork(0);
      RUN(petr_bbox_epilogue_bwd(&a, gr->d_bbox, d_raw, Wm + W.d_ref_tmp, s));
      float* d_r2 = Wm + W.s0_r2;
      petr_gemm_args g;
      // one launch per branch for the input-gradient chain (petr_branch_bwd; PETR_BRANCH_BWD_FUSED=0, diagnostic builds: the
      // contraction / LayerNorm-backward chains below)
      const bool bwd_fused = C == 256 && d.ncls <= 16 && d.code <= 16 && petr_tune("PETR_BRANCH_BWD_FUSED", 1) != 0;
      petr_branch_bwd_args bb;
      memset(&bb, 0, sizeof bb);
      bb.param_gs = P.br_stride; bb.rows = (int)RG; bb.groups = G;
      if (!cfg->with_multi) {
        // (fused: the last Linear's 10 x 256 weight gradient is formed inside petr_branch_bwd - as a contraction of its own it was a
        // 4-tile x 64-slice scalar-load launch of 40-55 us that sat beside the first cross-attention backward)
        if (!bwd_fused)
          RUN(wgrad(gw(lin_wgrad(d_raw, d.code, Wm + W.r2, C, Gp + P.reg_w[2], Gp + P.reg_b[2], RG, d.code, C), d.code, C)));
        if (!bwd_fused) {
          g = gd(lin_dgrad(d_raw, Pm + P.reg_w[2], d_r2, RG, d.code, C), d.code, C, C);
          g.flags = PETR_GEMM_RELU_MASK; g.r = Wm + W.r2; g.ldr = C;
          RUN(petr_gemm(&g, s));
        }
      } else {
        static const int TH_DIMS[5] = {2, 1, 3, 2, 2}, TH_COL[5] = {0, 2, 3, 6, 8};
        float* d_th = Wm + W.d_th_h;
        if (C == 256 && petr_tune("PETR_TASK_HEADS_FUSED", 1) != 0) {     // input, weight and bias gradients of the five in one launch
          petr_task_heads_bwd_args tb;
          memset(&tb, 0, sizeof tb);
          tb.d_out = d_raw; tb.ld_out = d.code; tb.h = Wm + W.th_h; tb.w2 = Pm + P.th_w2; tb.param_gs = P.br_stride;
          tb.head_stride = P.th_stride; tb.d_h = d_th; tb.dw2 = Gp + P.th_w2; tb.db2 = Gp + P.th_b2;
          tb.rows = (int)RG; tb.groups = G; tb.heads = 5;
          for (int t = 0; t < 5; ++t) { tb.dims[t] = TH_DIMS[t]; tb.cols[t] = TH_COL[t]; }
          RUN(petr_task_heads_bwd(&tb, s));
        } else
        for (int t = 0; t < 5; ++t) {
          RUN(wgrad(gw(lin_wgrad(d_raw + TH_COL[t], d.code, Wm + W.th_h + (long)t * RG * C, C, Gp + P.th_w2 + t * P.th_stride,
                                 Gp + P.th_b2 + t * P.th_stride, RG, TH_DIMS[t], C), d.code, 5 * C)));
          g = gd(lin_dgrad(d_raw + TH_COL[t], Pm + P.th_w2 + t * P.th_stride, d_th + (long)t * RG * C, RG, TH_DIMS[t], C),
                 d.code, 5 * C, 5 * C);
          g.lda = d.code;
          g.flags = PETR_GEMM_RELU_MASK; g.r = Wm + W.th_h + (long)t * RG * C; g.ldr = C;
          RUN(petr_gemm(&g, s));
        }
        // first Linear of the 5 task heads: weights batched over (level, head); input gradient summed over the heads
        g = gw(lin_wgrad(d_th, C, Wm + W.r2, C, Gp + P.th_w1, Gp + P.th_b1, RG, C, C), 5 * C, C);
        g.nb1 = 5; g.a_bs1 = RG * C; g.b_bs1 = 0; g.c_bs1 = P.th_stride; g.cs_bs1 = P.th_stride;
        RUN(wgrad(g));
        g = gemm0();      // d_r2[r][c] = sum_{t,o} d_th[t][r][o] * tw1_t[o][c], masked by relu(r2)
        g.a = d_th; g.lda = C; g.a_kcontig = 1; g.a_bs0 = 5 * RG * C;
        g.b = Pm + P.th_w1; g.ldb = C; g.b_kcontig = 0; g.b_bs0 = P.br_stride;
        g.c = d_r2; g.ldc = C; g.c_bs0 = RG * C;
        g.r = Wm + W.r2; g.ldr = C; g.r_bs0 = RG * C; g.flags = PETR_GEMM_RELU_MASK;
        g.M = (int)RG; g.N = C; g.K = 5 * C; g.nb0 = G;
        g.k_seg = C; g.a_seg_stride = RG * C; g.b_seg_stride = P.th_stride;
        RUN(petr_gemm(&g, s));
      }
      float* d_r1 = Wm + W.s0_r1;
      if (bwd_fused) {        // the rest of the chain in one launch: d_r2 (given, or from d_raw), d_r1, d_outs
        petr_branch_bwd_args b = bb;
        if (!cfg->with_multi) { b.d_out = d_raw; b.n_out = d.code; b.w3 = Pm + P.reg_w[2]; b.dw3 = Gp + P.reg_w[2]; b.db3 = Gp + P.reg_b[2]; }
        else b.d_y2 = d_r2;
        b.y2 = Wm + W.r2; b.w2 = Pm + P.reg_w[1]; b.y1 = Wm + W.r1; b.w1 = Pm + P.reg_w[0];
        b.d_h2 = d_r2; b.d_h1 = d_r1; b.d_x = Wm + W.d_outs;
        RUN(petr_branch_bwd(&b, s));
      }
      RUN(wgrad(gw(lin_wgrad(d_r2, C, Wm + W.r1, C, Gp + P.reg_w[1], Gp + P.reg_b[1], RG, C, C), C, C)));
      if (!bwd_fused) {
        g = gd(lin_dgrad(d_r2, Pm + P.reg_w[1], d_r1, RG, C, C), C, C, C);
        g.flags = PETR_GEMM_RELU_MASK; g.r = Wm + W.r1; g.ldr = C;
        RUN(petr_gemm(&g, s));
      }
      RUN(wgrad(gw(lin_wgrad(d_r1, C, Wm + W.outs, C, Gp + P.reg_w[0], Gp + P.reg_b[0], RG, C, C), C, C)));
      if (!bwd_fused) {
        g = gd(lin_dgrad(d_r1, Pm + P.reg_w[0], Wm + W.d_outs, RG, C, C), C, C, C);
        RUN(petr_gemm(&g, s));
      }
      // ---- cls branch: independent of the reg branch until the post-norm, so it runs beside it on side stream 0 and
      // leaves its input gradient in a buffer of its own (summed by the post-norm backward's prologue) ----
      void* sc = ln.side(0);          // (forked before the reg branch was enqueued, see above)
      if (!bwd_fused)
        RUN(wgrad(gw(lin_wgrad(gr->d_cls, d.ncls, Wm + W.c2n, C, Gp + P.cls_w[2], Gp + P.cls_b[2], RG, d.ncls, C), d.ncls, C)));
      float* d_c2 = Wm + W.s0_c2;
      float* d_c1 = Wm + W.s0_c1;
      if (bwd_fused) {
        petr_branch_bwd_args b = bb;
        b.d_out = gr->d_cls; b.n_out = d.ncls; b.w3 = Pm + P.cls_w[2]; b.dw3 = Gp + P.cls_w[2]; b.db3 = Gp + P.cls_b[2];
        b.y2 = Wm + W.c2n; b.h2 = Wm + W.c2; b.mean2 = Wm + W.c2_mean; b.rstd2 = Wm + W.c2_rstd; b.g2 = Pm + P.cls_g[1];
        b.w2 = Pm + P.cls_w[1];
        b.y1 = Wm + W.c1n; b.h1 = Wm + W.c1; b.mean1 = Wm + W.c1_mean; b.rstd1 = Wm + W.c1_rstd; b.g1 = Pm + P.cls_g[0];
        b.w1 = Pm + P.cls_w[0];
        b.d_h2 = d_c2; b.d_h1 = d_c1; b.d_x = Wm + W.s0_outs_c;
        b.dg2 = Gp + P.cls_g[1]; b.dbe2 = Gp + P.cls_be[1]; b.dg1 = Gp + P.cls_g[0]; b.dbe1 = Gp + P.cls_be[0];
        RUN(petr_branch_bwd(&b, sc));
      } else {
        float* d_c2n = Wm + W.s0_c2n;
        g = gd(lin_dgrad(gr->d_cls, Pm + P.cls_w[2], d_c2n, RG, d.ncls, C), d.ncls, C, C);
        RUN(petr_gemm(&g, sc));
        for (int gi = 0; gi < G; ++gi)
          RUN(ln_bwd(Wm + W.c2 + gi * RG * C, Wm + W.c2_mean + gi * RG, Wm + W.c2_rstd + gi * RG, Pm + P.cls_g[1] + gi * P.br_stride,
                     d_c2n + gi * RG * C, Wm + W.c2n + gi * RG * C, d_c2 + gi * RG * C, Gp + P.cls_g[1] + gi * P.br_stride,
                     Gp + P.cls_be[1] + gi * P.br_stride, RG, C, PETR_LN_RELU, 0, sc));
      }
      RUN(wgrad(gw(lin_wgrad(d_c2, C, Wm + W.c1n, C, Gp + P.cls_w[1], Gp + P.cls_b[1], RG, C, C), C, C)));
      if (!bwd_fused) {
        float* d_c1n = Wm + W.s0_c1n;
        g = gd(lin_dgrad(d_c2, Pm + P.cls_w[1], d_c1n, RG, C, C), C, C, C);
        RUN(petr_gemm(&g, sc));
        for (int gi = 0; gi < G; ++gi)
          RUN(ln_bwd(Wm + W.c1 + gi * RG * C, Wm + W.c1_mean + gi * RG, Wm + W.c1_rstd + gi * RG, Pm + P.cls_g[0] + gi * P.br_stride,
                     d_c1n + gi * RG * C, Wm + W.c1n + gi * RG * C, d_c1 + gi * RG * C, Gp + P.cls_g[0] + gi * P.br_stride,
                     Gp + P.cls_be[0] + gi * P.br_stride, RG, C, PETR_LN_RELU, 0, sc));
      }
      RUN(wgrad(gw(lin_wgrad(d_c1, C, Wm + W.outs, C, Gp + P.cls_w[0], Gp + P.cls_b[0], RG, C, C), C, C)));
      if (!bwd_fused) {
        g = gd(lin_dgrad(d_c1, Pm + P.cls_w[0], Wm + W.s0_outs_c, RG, C, C), C, C, C);
        RUN(petr_gemm(&g, sc));
      }
      ln.join(0);
      // ---- post_norm over all levels -> d_xs[l] ----
      RUN(ln_bwd(Wm + W.xs, Wm + W.mean_p, Wm + W.rstd_p, Pm + P.post_g, Wm + W.d_outs, nullptr, Wm + W.d_xs, Gp + P.post_g,
                 Gp + P.post_b, d.R, C, 0, 0, s, 1, 0, Wm + W.s0_outs_c));
      if (ev_tr) (void)hipStreamWaitEvent(ln.main, ev_tr, 0);      // the layer stages read the transposed weights
    } else if (stage <= d.NL) {
      const int l = d.NL - stage;
      const LayerP& lp = P.lay[l];
      const LayerW& lw = W.lay[l];
      const WOff::LayerG& lg = W.lg[l];
      const float* x_in = l == 0 ? Wm + W.x0 : Wm + W.xs + (long)(l - 1) * d.BQ * C;
      const float* G = Wm + W.d_xs + (long)l * d.BQ * C;     // d(x3_l): post-norm path (+ layer l+1's input grad)
      // Training mode: z = drop(f) + identity at each of the three LayerNorms, so the sub-layer branch gets
      // d_z * keep/(1-p) (second output of the LayerNorm backward) while the identity branch keeps d_z.
      const bool training = io->dropout_p > 0.f;
      petr_dropout dr[6];
      for (int k = 0; k < 6; ++k) { dr[k].seed = io->dropout_seed; dr[k].site = (uint32_t)(8 * l + k); dr[k].p = io->dropout_p; }
      // LN2 / FFN
      float* d_z2 = Wm + lg.d_z2;
      float* d_f2 = training ? Wm + lg.d_zd[2] : d_z2;          // gradient of the second FFN contraction's output
      float* d_h = Wm + lg.d_h;                                 // [BQ, F]
      const WOff::LayerT& wt = W.wt[l];
      petr_gemm_args g;
      // stored hidden = relu(.) * keep/(1-p): (hidden > 0) is relu-mask AND keep; the 1/(1-p) rides on alpha
      // PETR_FUSE_LN_BWD: 3 (default) both FFN input gradients in one launch (petr_ffn_bwd: d_h stays on chip between them);
      // 2: LayerNorm backward + the FFN2 input gradient with its ReLU mask in petr_ln_bwd_proj (8 column blocks); 1: separate
      static const int fuse_lvl = petr_tune("PETR_FUSE_LN_BWD", 3);
      static const bool ffn_bwd16_env = env_on("PETR_FFN_BWD_FUSED_BF16");
      const bool ffn_bwd_fused = fuse_bwd && fuse_lvl >= 3 && W.ffn_fsplit > 0 && (!ffn16 || ffn_bwd16_env);
      const bool fuse_ffn2 = fuse_lvl >= 2;
      static const bool fuse_in_env = env_on("PETR_FUSE_IN_DGRAD");
      const bool fuse_in = ffn_bwd_fused && fuse_in_env && C == 256;       // see the end of the stage: must match layer l+1's choice
      float* d_x2 = Wm + lg.d_x2;
      const int sk = ffn_bwd_fused ? W.ffn_fsplit : W.ffn_split;
      if (ffn_bwd_fused) {
        if (fuse_in && l + 1 < d.NL) {
          // the input gradient of layer l+1's self-attention in-projection (K = 3C) is formed here, in front of the LayerNorm
          // backward that consumes it: d(x3_l) = d_qkv(l+1) W_in(l+1) + d_z0(l+1) (identity) + the branches' gradient
          RUN(ln_bwd_proj(Wm + lw.z2, Wm + lw.mean2, Wm + lw.rstd2, Pm + lp.n_g[2], Wm + W.lg[l + 1].d_z0, 1, 0, G, d_z2,
                          training ? d_f2 : nullptr, training ? &dr[5] : nullptr, Gp + lp.n_g[2], Gp + lp.n_b[2], d.BQ, nullptr, 0, 1.f,
                          nullptr, nullptr, s, Wm + W.d_qkv + (long)(l + 1) * d.BQ * 3 * C, Pm + P.lay[l + 1].sa_in_w, 3));
        } else {
        RUN(ln_bwd(Wm + lw.z2, Wm + lw.mean2, Wm + lw.rstd2, Pm + lp.n_g[2], G, nullptr, d_z2, Gp + lp.n_g[2], Gp + lp.n_b[2],
                   d.BQ, C, 0, 0, s, 1, 0, nullptr, training ? d_f2 : nullptr, &dr[5]));
        }
        RUN(wgrad(lin_wgrad(d_f2, C, Wm + lw.hff, d.F, Gp + lp.f2_w, Gp + lp.f2_b, d.BQ, C, d.F)));
        petr_ffn_bwd_args fb;
        memset(&fb, 0, sizeof fb);
        fb.dy = d_f2; fb.w2 = Pm + lp.f2_w; fb.hidden = Wm + lw.hff; fb.alpha = training ? hidden_drop_scale(dr[4]) : 1.f;
        fb.w1 = Pm + lp.f1_w; fb.d_hidden = d_h; fb.part = d_x2; fb.part_stride = d.BQ * C;
        fb.M = (int)d.BQ; fb.F = (int)d.F; fb.n_split = sk;
        RUN(petr_ffn_bwd(&fb, s));
        RUN(wgrad(lin_wgrad(d_h, d.F, Wm + lw.x2, C, Gp + lp.f1_w, Gp + lp.f1_b, d.BQ, d.F, C)));
      } else {
      if (fuse_bwd && fuse_ffn2 && d.F % 256 == 0) {
        RUN(ln_bwd_proj(Wm + lw.z2, Wm + lw.mean2, Wm + lw.rstd2, Pm + lp.n_g[2], G, 1, 0, nullptr, d_z2,
                        training ? d_f2 : nullptr, training ? &dr[5] : nullptr, Gp + lp.n_g[2], Gp + lp.n_b[2], d.BQ, Pm + lp.f2_w,
                        d.F / 256, training ? hidden_drop_scale(dr[4]) : 1.f, Wm + lw.hff, d_h, s));
        RUN(wgrad(lin_wgrad(d_f2, C, Wm + lw.hff, d.F, Gp + lp.f2_w, Gp + lp.f2_b, d.BQ, C, d.F)));
      } else {
      RUN(ln_bwd(Wm + lw.z2, Wm + lw.mean2, Wm + lw.rstd2, Pm + lp.n_g[2], G, nullptr, d_z2, Gp + lp.n_g[2], Gp + lp.n_b[2],
                 d.BQ, C, 0, 0, s, 1, 0, nullptr, training ? d_f2 : nullptr, &dr[5]));
      RUN(wgrad(lin_wgrad(d_f2, C, Wm + lw.hff, d.F, Gp + lp.f2_w, Gp + lp.f2_b, d.BQ, C, d.F)));
      g = dgrad_t ? lin_dgrad_t(d_f2, Wm + wt.f2, d_h, d.BQ, C, d.F) : lin_dgrad(d_f2, Pm + lp.f2_w, d_h, d.BQ, C, d.F);
      if (ffn16) g = lin_dgrad(d_f2, Wp(lp.f2_w), d_h, d.BQ, C, d.F);     // bf16 route: the bf16 weight, read K-major
      g.flags = PETR_GEMM_RELU_MASK | (ffn16 ? PETR_GEMM_BF16 | wflag : 0); g.r = Wm + lw.hff; g.ldr = d.F;
      if (training) g.alpha = hidden_drop_scale(dr[4]);
      RUN(petr_gemm(&g, s));
      }
      RUN(wgrad(lin_wgrad(d_h, d.F, Wm + lw.x2, C, Gp + lp.f1_w, Gp + lp.f1_b, d.BQ, d.F, C)));
      // d_x2 = d_h @ W1 + d_z2 (identity path): K = F is long and there are only BQ/64 x 4 output tiles, so the
      // contraction is split over K into slabs that the LayerNorm backward sums in its prologue
      g = dgrad_t ? lin_dgrad_t(d_h, Wm + wt.f1, d_x2, d.BQ, d.F, C) : lin_dgrad(d_h, Pm + lp.f1_w, d_x2, d.BQ, d.F, C);
      if (ffn16) { g = lin_dgrad(d_h, Wp(lp.f1_w), d_x2, d.BQ, d.F, C); g.flags = PETR_GEMM_BF16 | wflag; }
      if (sk > 1) { g.split_k = sk; g.c_split_stride = d.BQ * C; }
      else { g.r = d_z2; g.ldr = C; }
      RUN(petr_gemm(&g, s));
      }
      const float* id2 = (ffn_bwd_fused || sk > 1) ? d_z2 : nullptr;    // LN2's identity path: not inside the slabs
      // LN1 / cross-attention
      float* d_z1 = Wm + lg.d_z1;
      float* d_f1 = training ? Wm + lg.d_zd[1] : d_z1;          // gradient of the cross-attention out-projection's output
      float* d_ao = Wm + lg.d_ao;
      if (fuse_bwd) {
        RUN(ln_bwd_proj(Wm + lw.z1, Wm + lw.mean1, Wm + lw.rstd1, Pm + lp.n_g[1], d_x2, sk, d.BQ * C, id2, d_z1,
                        training ? d_f1 : nullptr, training ? &dr[3] : nullptr, Gp + lp.n_g[1], Gp + lp.n_b[1], d.BQ, Pm + lp.ca_out_w,
                        1, 1.f, nullptr, d_ao, s));
        RUN(wgrad(lin_wgrad(d_f1, C, Wm + lw.ao_c, C, Gp + lp.ca_out_w, Gp + lp.ca_out_b, d.BQ, C, C)));
      } else {
      RUN(ln_bwd(Wm + lw.z1, Wm + lw.mean1, Wm + lw.rstd1, Pm + lp.n_g[1], d_x2, nullptr, d_z1, Gp + lp.n_g[1],
                 Gp + lp.n_b[1], d.BQ, C, 0, 0, s, sk, d.BQ * C, id2, training ? d_f1 : nullptr, &dr[3]));
      RUN(wgrad(lin_wgrad(d_f1, C, Wm + lw.ao_c, C, Gp + lp.ca_out_w, Gp + lp.ca_out_b, d.BQ, C, C)));
      g = dgrad_t ? lin_dgrad_t(d_f1, Wm + wt.ca_out, d_ao, d.BQ, C, C) : lin_dgrad(d_f1, Pm + lp.ca_out_w, d_ao, d.BQ, C, C);
      RUN(petr_gemm(&g, s));
      }
      float* d_qc = Wm + W.d_qc + (long)l * d.BQ * C;
      if (bf16)
        RUN(mha_b_bf16(Wm + lw.qc, (long)d.Q * C, C, k16 + (long)l * d.L * C, (long)d.NL * d.L * C, C,
                       v16 + (long)l * d.L * C, Wm + lw.ao_c, d_ao, Wm + lw.lse_c, kpm, d_qc,
                       dkv16 ? dk16 + (long)l * d.L * C : reinterpret_cast<uint16_t*>(Wm + W.dk_all + (long)l * d.L * C),
                       dkv16 ? dv16 + (long)l * d.L * C : reinterpret_cast<uint16_t*>(Wm + W.dv_all + (long)l * d.L * C), dkv16, d,
                       (int)d.L, s, training ? &dr[2] : nullptr, use_bits ? bits_ptr(l, 0) : nullptr));
      else
      RUN(mha_b(Wm + lw.qc, (long)d.Q * C, C, Wm + W.k_all + (long)l * d.L * C, (long)d.NL * d.L * C, C,
                Wm + W.v_all + (long)l * d.L * C, Wm + lw.ao_c, d_ao, Wm + lw.lse_c, kpm, d_qc,
                Wm + W.dk_all + (long)l * d.L * C, Wm + W.dv_all + (long)l * d.L * C, d, (int)d.L, mws, W.mha_ws_bytes, s,
                training ? &dr[2] : nullptr, use_bits ? bits_ptr(l, 0) : nullptr));
      // (tried: start the weight gradients queued so far BEHIND the cross-attention backward - that kernel fills the machine and
      // runs 211 -> 240-270 us at 24 000 tokens when they execute beside it)
      // Opt-in (PETR_WGRAD_MIDFLUSH=1): interleaved three-round A/B on one box (scripts/ab_multi.sh) - c5 bf16 4.47 -> 4.54 ms,
      // v2-800 bf16 5.78 -> 5.86, p4-1600 bf16 6.04 -> 6.06, c5 fp32 4.94 -> 5.01: the second fork per layer costs more than
      // the overlap it removes, so the stage keeps its single flush at the end.
      static const bool midflush = petr_tune("PETR_WGRAD_MIDFLUSH", 0) != 0;
      static const bool flush_b = petr_tune("PETR_WGRAD_FLUSH_EARLY", 1) == 2;
      if (defer && (midflush || flush_b)) RUN(flush_wgrads());
      // K / V projection backward of THIS layer (token-sized: the largest contractions of the backward) leaves the
      // critical path: dW_l and d_src (+)= dKV_l W_l go to the side streams right behind the attention backward that
      // produced dK_l / dV_l and run beside the 900-row chain of the remaining layers; the final stage only joins.
      // K path on side 0, V path on side 1 (each accumulates into one buffer, so each stays on one in-order stream).
      for (int kv = 0; kv < 2 && kv_overlap; ++kv) {
        const float* dkv = dkv_ptr(kv, (long)l * d.L * C);
        const float* src = kv == 0 ? Wm + W.mempos : Wm + W.mem;
        void* sd = ln.side(kv);
        ln.fork(kv);
        petr_gemm_args g2 = gemm0();      // dW_l[C,C] += sum_{b,t} dKV[b][l][t][o] * src[b][t][c]
        g2.a = dkv; g2.lda = C; g2.a_kcontig = 0;
        g2.b = src; g2.ldb = C; g2.b_kcontig = 0;
        g2.c = Gp + lp.ca_in_w + (long)(kv + 1) * C * C; g2.ldc = C;
        g2.a_colsum = Gp + lp.ca_in_b + (kv + 1) * C;
        g2.M = C; g2.N = C; g2.K = (int)d.BL;
        g2.k_seg = (int)d.L; g2.a_seg_stride = (long)d.NL * d.L * C; g2.b_seg_stride = d.L * C;
        g2.flags = PETR_GEMM_ATOMIC | dkv_flag | wflag;
        g2.split_k = 32;
        g2 = L16(g2);
        RUN(petr_gemm(&g2, sd));
        g2 = gemm0();      // d_src[b][t][c] (+)= sum_o dKV[b][l][t][o] * W_l[o][c]   (first layer processed overwrites)
        g2.a = dkv; g2.lda = C; g2.a_kcontig = 1; g2.a_bs0 = (long)d.NL * d.L * C;
        g2.b = Wp(lp.ca_in_w + (long)(kv + 1) * C * C); g2.ldb = C; g2.b_kcontig = 0;
        g2.c = Wm + (kv == 0 ? W.d_mempos : W.d_mem); g2.ldc = C; g2.c_bs0 = d.L * C;
        g2.M = (int)d.L; g2.N = C; g2.K = C; g2.nb0 = d.B;
        g2.flags = (l != d.NL - 1 ? PETR_GEMM_ACCUMULATE : 0) | dkv_flag | wflag;
        g2 = L16(g2);
        RUN(petr_gemm(&g2, sd));
      }
      // q projection of the cross-attention (rows 0..C of in_proj): weight grads live in the final block
      RUN(wgrad(lin_wgrad(d_qc, C, Wm + lw.xe1, C, Gp + lp.ca_in_w, Gp + lp.ca_in_b, d.BQ, C, C)));
      // query_pos gradient, this layer's share: d(q-proj input) of the cross-attention into slab [0][l].  It only feeds the
      // query-embedding MLP in the final stage, so it rides with the weight gradients on the side streams instead of
      // standing (NL + 1 launches, ~180 us at 900 queries) at the end of the critical chain.
      {
        petr_gemm_args gs = dgrad_t ? lin_dgrad_t(d_qc, Wm + wt.ca_q, Wm + W.d_e_slab + (long)l * d.BQ * C, d.BQ, C, C)
                                    : lin_dgrad(d_qc, Pm + lp.ca_in_w, Wm + W.d_e_slab + (long)l * d.BQ * C, d.BQ, C, C);
        // bf16 mode: through the TILED contraction (60 workgroups of 256 threads; a single K segment selects it) instead of the
        // 232 x 512-thread latency kernel - beside the bf16 attention backward (one workgroup per CU, three rounds) the slabs' CU
        // slots are what it waits for: every bf16 step 1.6-1.9 % faster, every fp32 step 0.2-0.5 % slower (same-box A/B)
        if (slab_tiled) gs.k_seg = gs.K;
        RUN(wgrad(gs));
      }
      float* d_x1 = Wm + lg.d_x1;
      if (!fuse_bwd) {       // (fused: the leading product of the LayerNorm-0 backward kernel below)
        g = dgrad_t ? lin_dgrad_t(d_qc, Wm + wt.ca_q, d_x1, d.BQ, C, C) : lin_dgrad(d_qc, Pm + lp.ca_in_w, d_x1, d.BQ, C, C);
        g.r = d_z1; g.ldr = C;
        RUN(petr_gemm(&g, s));
      }
      // LN0 / self-attention
      float* d_z0 = Wm + lg.d_z0;
      float* d_f0 = training ? Wm + lg.d_zd[0] : d_z0;          // gradient of the self-attention out-projection's output
      float* d_ao_s = Wm + lg.d_ao_s;
      if (fuse_bwd) {
        // d(x1) = d_qc Wq (the query projection's input gradient, formed in the kernel) + d_z1 (identity path of LN1)
        RUN(ln_bwd_proj(Wm + lw.z0, Wm + lw.mean0, Wm + lw.rstd0, Pm + lp.n_g[0], nullptr, 0, 0, d_z1, d_z0,
                        training ? d_f0 : nullptr, training ? &dr[1] : nullptr, Gp + lp.n_g[0], Gp + lp.n_b[0], d.BQ, Pm + lp.sa_out_w,
                        1, 1.f, nullptr, d_ao_s, s, d_qc, Pm + lp.ca_in_w));
        RUN(wgrad(lin_wgrad(d_f0, C, Wm + lw.ao_s, C, Gp + lp.sa_out_w, Gp + lp.sa_out_b, d.BQ, C, C)));
      } else {
      RUN(ln_bwd(Wm + lw.z0, Wm + lw.mean0, Wm + lw.rstd0, Pm + lp.n_g[0], d_x1, nullptr, d_z0, Gp + lp.n_g[0],
                 Gp + lp.n_b[0], d.BQ, C, 0, 0, s, 1, 0, nullptr, training ? d_f0 : nullptr, &dr[1]));
      RUN(wgrad(lin_wgrad(d_f0, C, Wm + lw.ao_s, C, Gp + lp.sa_out_w, Gp + lp.sa_out_b, d.BQ, C, C)));
      g = dgrad_t ? lin_dgrad_t(d_f0, Wm + wt.sa_out, d_ao_s, d.BQ, C, C) : lin_dgrad(d_f0, Pm + lp.sa_out_w, d_ao_s, d.BQ, C, C);
      RUN(petr_gemm(&g, s));
      }
      // The stage's weight gradients start HERE (one fork per stage): beside the self-attention backward (256 workgroups) and
      // the next stage's first 57-workgroup kernel, where three quarters of the chip idle - not beside the next FFN backward,
      // which fills it (PETR_WGRAD_FLUSH_EARLY=0: at the end of the stage).  The in-projection's weight gradients queued below
      // leave with the next stage's fork, or at the end of the call.
      static const int flush_early = petr_tune("PETR_WGRAD_FLUSH_EARLY", 1);
      if (defer && flush_early == 1) RUN(flush_wgrads());
      float* d_qkv = Wm + W.d_qkv + (long)l * d.BQ * 3 * C;
      if (self16_b) {      // the forward ran the bf16 kernel on the bf16 K / V copy it left in qkv16: its gradient kernel, fp32 += outputs
        const uint16_t* q16 = reinterpret_cast<const uint16_t*>(Wm + lw.qkv16);
        RUN(mha_b_bf16(Wm + lw.qkv, (long)d.Q * 3 * C, 3 * C, q16 + C, (long)d.Q * 3 * C, 3 * C, q16 + 2 * C, Wm + lw.ao_s, d_ao_s,
                       Wm + lw.lse_s, nullptr, d_qkv, reinterpret_cast<uint16_t*>(d_qkv + C), reinterpret_cast<uint16_t*>(d_qkv + 2 * C),
                       false, d, d.Q, s, training ? &dr[0] : nullptr, use_bits ? bits_ptr(l, 1) : nullptr, false));
      } else
      RUN(mha_b(Wm + lw.qkv, (long)d.Q * 3 * C, 3 * C, Wm + lw.qkv + C, (long)d.Q * 3 * C, 3 * C, Wm + lw.qkv + 2 * C,
                Wm + lw.ao_s, d_ao_s, Wm + lw.lse_s, nullptr, d_qkv, d_qkv + C, d_qkv + 2 * C, d, d.Q, mws, W.mha_ws_bytes, s,
                training ? &dr[0] : nullptr, use_bits ? bits_ptr(l, 1) : nullptr));
      // in_proj: q,k rows see x + query_pos, v rows see x
      RUN(wgrad(lin_wgrad(d_qkv, 3 * C, Wm + lw.xe_in, C, Gp + lp.sa_in_w, Gp + lp.sa_in_b, d.BQ, 2 * C, C)));
      {   // query_pos gradient through the self-attention q / k rows: slab [1][l]
        float* slab = Wm + W.d_e_slab + (long)(d.NL + l) * d.BQ * C;
        petr_gemm_args ge = dgrad_t ? lin_dgrad_t(d_qkv, Wm + wt.sa_in, slab, d.BQ, 2 * C, C) : lin_dgrad(d_qkv, Pm + lp.sa_in_w, slab, d.BQ, 2 * C, C);
        ge.lda = 3 * C;
        if (dgrad_t) ge.ldb = 3 * C;          // rows of the transposed [C][3C] in_proj; K = the first 2C columns
        if (slab_tiled) ge.k_seg = ge.K;
        RUN(wgrad(ge));
      }
      RUN(wgrad(lin_wgrad(d_qkv + 2 * C, 3 * C, x_in, C, Gp + lp.sa_in_w + (long)2 * C * C, Gp + lp.sa_in_b + 2 * C, d.BQ, C,
                          C)));
      if (l > 0 && !fuse_in) {       // (fuse_in: layer l-1's first kernel forms it)
        // d(x_in) = d_z0 (identity) + d_qkv @ W_in, added to the post-norm gradient of level l-1
        float* dst = Wm + W.d_xs + (long)(l - 1) * d.BQ * C;
        g = dgrad_t ? lin_dgrad_t(d_qkv, Wm + wt.sa_in, dst, d.BQ, 3 * C, C) : lin_dgrad(d_qkv, Pm + lp.sa_in_w, dst, d.BQ, 3 * C, C);
        g.flags = PETR_GEMM_ACCUMULATE;
        g.r = d_z0; g.ldr = C;                                   // identity path folded into the same epilogue
        RUN(petr_gemm(&g, s));
      }
    } else {
      // ================= final stage =================
      const int V = d.B * d.N;
      // query_pos gradient: sum over layers and batch of the d(q-proj inputs) slabs the layer stages produced on the side
      // streams, then the query_embedding MLP + pos2posemb3d backward.  It depends on the LAYER stages only, so it runs early in
      // this stage (behind the K / V input gradients, by when the layer stages' last weight gradients have drained): ordered
      // behind the side streams as they stand NOW - joined at the end of the stage it waited for this stage's token-sized
      // weight gradients as well and then ran alone, 100 us of the c5 step (PETR_QE_EARLY=0: that order).
      static const bool qe_early = env_on("PETR_QE_EARLY");
      hipEvent_t ev_q[2] = {nullptr, nullptr};
      if (qe_early) {
        RUN(flush_wgrads());
        if (ln.ctx)
          for (int i = 0; i < 2 && i < ln.ctx->n_side; ++i) {
            ev_q[i] = ln.next();
            (void)hipEventRecord(ev_q[i], ln.ctx->side[i]);
          }
      }
      // With side streams the whole chain (seven small launches, 86 us at c5 / 117 us at p4-1600) runs on side stream 0: nothing on
      // the main stream depends on it, and standing in the main stream it delayed the position-embedding input gradients - and
      // with them the token-sized weight gradients that end the step - by its full length (PETR_QE_SIDE=0: main stream)
      static const bool qe_side_env = petr_tune("PETR_QE_SIDE", 1) != 0;
      auto query_embedding_bwd = [&](bool early) -> int {
        petr_gemm_args g;
        const bool on_side = early && qe_side_env && ln.ctx && ln.ctx->n_side >= 2;
        void* sq = on_side ? (void*)ln.ctx->side[0] : s;
        if (on_side) {
          if (ev_q[1]) (void)hipStreamWaitEvent((hipStream_t)sq, ev_q[1], 0);      // (ev_q[0] was recorded on this very stream)
        } else if (early) {
          for (int i = 0; i < 2; ++i)
            if (ev_q[i]) (void)hipStreamWaitEvent(ln.main, ev_q[i], 0);
        } else {
          ln.join(0);
          ln.join(1);
        }
        RUN(petr_reduce_batch(Wm + W.d_e_slab, 2 * d.NL * d.B, d.Q, C, Wm + W.d_e, 0, sq));
        // query_embedding MLP + pos2posemb3d (petr_head.py:422-423); on the side stream its weight gradients follow in order
        g = lin_wgrad(Wm + W.d_e, C, Wm + W.qe_h, C, Gp + P.qe_w2, Gp + P.qe_b2, d.Q, C, C);
        if (on_side) RUN(petr_gemm(&g, sq)); else RUN(wgrad(g));
        g = lin_dgrad(Wm + W.d_e, Pm + P.qe_w2, Wm + W.d_qe_h, d.Q, C, C);
        g.flags = PETR_GEMM_RELU_MASK; g.r = Wm + W.qe_h; g.ldr = C;
        RUN(petr_gemm(&g, sq));
        g = lin_wgrad(Wm + W.d_qe_h, C, Wm + W.posemb, C * 3 / 2, Gp + P.qe_w1, Gp + P.qe_b1, d.Q, C, C * 3 / 2);
        if (on_side) RUN(petr_gemm(&g, sq)); else RUN(wgrad(g));
        g = lin_dgrad(Wm + W.d_qe_h, Pm + P.qe_w1, Wm + W.d_posemb, d.Q, C, C * 3 / 2);
        RUN(petr_gemm(&g, sq));
        RUN(petr_posemb3d_bwd(Pm + P.ref, io->dim_t, Wm + W.d_posemb, Wm + W.d_ref_tmp, d.Q, C / 2, sq));
        RUN(petr_axpy(Gp + P.ref, Wm + W.d_ref_tmp, 1.f, (long)d.Q * 3, sq));
        return PETR_OK;
      };
      // (opt-in schedule) the K/V projection backward of every layer ran on the side streams behind its layer: wait for it
      if (kv_overlap) {
        ln.join(0);
        ln.join(1);
      }
      for (int kv = 0; kv < 2 && !kv_overlap; ++kv) {     // single-stream / opt-out schedule: all layers in one contraction each
        const float* dkv = dkv_ptr(kv, 0);
        const float* src = kv == 0 ? Wm + W.mempos : Wm + W.mem;          // tok16: bf16 images, same element indexing
        petr_gemm_args g = gemm0();      // dW_l[C,C] += sum_{b,t} dKV[b][l][t][o] * src[b][t][c]
        g.a = dkv; g.lda = C; g.a_kcontig = 0; g.a_bs0 = d.L * C;
        g.b = src; g.ldb = C; g.b_kcontig = 0;
        g.c = Gp + P.lay[0].ca_in_w + (long)(kv + 1) * C * C; g.ldc = C; g.c_bs0 = P.ca_in_stride;
        g.a_colsum = Gp + P.lay[0].ca_in_b + (kv + 1) * C; g.cs_bs0 = P.ca_in_stride;
        g.M = C; g.N = C; g.K = (int)d.BL; g.nb0 = d.NL;
        g.k_seg = (int)d.L; g.a_seg_stride = (long)d.NL * d.L * C; g.b_seg_stride = d.L * C;
        g.flags = PETR_GEMM_ATOMIC | dkv_flag | wflag;          // (wflag: the bf16 B operand here is memory / key)
        g.split_k = 8;
        RUN(wgrad(L16(g)));
        g = gemm0();      // d_src[b][t][c] = sum_{l,o} dKV[b][l][t][o] * W_l[o][c]
        g.a = dkv; g.lda = C; g.a_kcontig = 1; g.a_bs0 = (long)d.NL * d.L * C;
        g.b = Wp(P.lay[0].ca_in_w + (long)(kv + 1) * C * C); g.ldb = C; g.b_kcontig = 0;
        g.c = Wm + (kv == 0 ? W.d_mempos : W.d_mem); g.ldc = C; g.c_bs0 = d.L * C;
        g.M = (int)d.L; g.N = C; g.K = d.NL * C; g.nb0 = d.B;
        g.k_seg = C; g.a_seg_stride = d.L * C; g.b_seg_stride = P.ca_in_stride;
        g.flags = dkv_flag | wflag;
        g = L16(g);
        RUN(petr_gemm(&g, s));
      }
      if (qe_early) RUN(query_embedding_bwd(true));
      // d_mem = dV-path + d(mem+pos) ; d_pos = d(mem+pos)
      RUN(petr_axpy(Wm + W.d_mem, Wm + W.d_mempos, 1.f, d.BL * C, s));
      const float* d_pos = Wm + W.d_mempos;
      const float* d_pe = d_pos;          // upstream gradient of position_encoder's output
      if (cfg->with_fpe) {
        // SELayer backward: pos3d = pe1 * sigmoid(u), u = expand(relu(reduce(mem)))
        RUN(petr_gate_bwd(d_pos, Wm + W.pe1, Wm + W.fpe_u, Wm + W.d_pe1, Wm + W.d_fpe_u, d.BL * C, s));
        d_pe = Wm + W.d_pe1;
        RUN(wgrad(L16(lin_wgrad(Wm + W.d_fpe_u, C, Wm + W.fpe_h, C, Gp + P.fpe_ew, Gp + P.fpe_eb, d.BL, C, C))));
        petr_gemm_args g = lin_dgrad(Wm + W.d_fpe_u, Wp(P.fpe_ew), Wm + W.d_fpe_h, d.BL, C, C);
        g.flags = PETR_GEMM_RELU_MASK | wflag; g.r = Wm + W.fpe_h; g.ldr = C;
        g = L16(g);
        RUN(petr_gemm(&g, s));
        g = lin_wgrad(Wm + W.d_fpe_h, C, Wm + W.mem, C, Gp + P.fpe_rw, Gp + P.fpe_rb, d.BL, C, C);
        g.flags |= wflag;                                              // B = memory (bf16 image when tok16)
        RUN(wgrad(L16(g)));
        g = lin_dgrad(Wm + W.d_fpe_h, Wp(P.fpe_rw), Wm + W.d_mem, d.BL, C, C);
        g.flags = PETR_GEMM_ACCUMULATE | wflag;
        g = L16(g);
        RUN(petr_gemm(&g, s));
      }
      // input_proj's weight gradient as soon as d_mem is final: it is the slowest of the stage's token-sized weight gradients
      // (one 256 x 256 tile set over all tokens: 158 us at 24 000 tokens) and used to be queued last, where it ended the step
      {
        petr_gemm_args g = gemm0();        // dW[C, Cin] += sum_{view,hw} d_mem[view*HW+hw][o] * x[view][ci][hw]
        g.a = Wm + W.d_mem; g.lda = C; g.a_kcontig = 0;
        g.b = io->feats; g.ldb = d.HW; g.b_kcontig = 1;
        g.c = Gp + P.in_w; g.ldc = d.Cin; g.a_colsum = Gp + P.in_b;
        g.M = C; g.N = d.Cin; g.K = V * d.HW;
        g.k_seg = d.HW; g.a_seg_stride = (long)d.HW * C; g.b_seg_stride = (long)d.Cin * d.HW;
        g.flags = PETR_GEMM_ATOMIC; g.split_k = 16;
        RUN(wgrad(L16(g)));
      }
      // position_encoder and adapt_pos3d (inputs carry no gradient)
      for (int which = 0; which < 2; ++which) {
        const long w1 = which == 0 ? P.pe_w1 : P.ad_w1, b1 = which == 0 ? P.pe_b1 : P.ad_b1;
        const long w2 = which == 0 ? P.pe_w2 : P.ad_w2, b2 = which == 0 ? P.pe_b2 : P.ad_b2;
        const float* hid = Wm + (which == 0 ? W.h1 : W.h2);
        const float* feat = Wm + (which == 0 ? W.vol : W.sine);
        const int Kin = which == 0 ? 3 * d.D : C * 3 / 2;
        float* d_hpe = Wm + W.d_hpe[which];
        const float* dy = which == 0 ? d_pe : d_pos;
        petr_gemm_args g = lin_wgrad(dy, C, hid, 4 * C, Gp + w2, Gp + b2, d.BL, C, 4 * C);
        if (hid16) g.flags |= PETR_GEMM_B_BF16;
        RUN(wgrad(L16(g)));
        g = lin_dgrad(dy, Wp(w2), d_hpe, d.BL, C, 4 * C);
        g.flags = PETR_GEMM_RELU_MASK | wflag; g.r = hid; g.ldr = 4 * C;
        if (hid16) g.flags |= PETR_GEMM_R_BF16 | PETR_GEMM_STORE_BF16;       // bf16 mask operand, bf16 hidden gradient
        g = L16(g);
        RUN(petr_gemm(&g, s));
        g = gemm0();      // dW1[4C, Kin] += sum_{view, hw} d_h[view*HW+hw][f] * feat[view][c][hw]
        g.a = d_hpe; g.lda = 4 * C; g.a_kcontig = 0;
        g.b = feat; g.ldb = d.HW; g.b_kcontig = 1;
        g.c = Gp + w1; g.ldc = Kin; g.a_colsum = Gp + b1;
        g.M = 4 * C; g.N = Kin; g.K = V * d.HW;
        g.k_seg = d.HW; g.a_seg_stride = (long)d.HW * 4 * C; g.b_seg_stride = (long)Kin * d.HW;
        g.flags = PETR_GEMM_ATOMIC | (hid16 ? PETR_GEMM_A_BF16 : 0); g.split_k = 8;
        RUN(wgrad(L16(g)));
      }
      // input_proj
      {
        petr_gemm_args g;
        if (gr->d_feats) {
          g = gemm0();      // d_x[view][ci][hw] = sum_o W[o][ci] * d_mem[view*HW+hw][o]
          g.a = Wp(P.in_w); g.lda = d.Cin; g.a_kcontig = 0;
          g.b = Wm + W.d_mem; g.ldb = C; g.b_kcontig = 1; g.b_bs0 = (long)d.HW * C;
          g.c = gr->d_feats; g.ldc = d.HW; g.c_bs0 = (long)d.Cin * d.HW;
          g.M = d.Cin; g.N = d.HW; g.K = C; g.nb0 = V;
          if (tok16) g.flags |= PETR_GEMM_A_BF16;
          g = L16(g);
          RUN(petr_gemm(&g, s));
        }
      }
      if (!qe_early) RUN(query_embedding_bwd(false));
    }
    // the stage's queued weight gradients: one fork, alternating side streams (a decoder-layer stage that forked early keeps
    // its last few for the next stage's fork unless the call ends here)
    static const bool flush_early_ = env_on("PETR_WGRAD_FLUSH_EARLY");
    const bool layer_stage = stage >= 1 && stage <= d.NL;
    if (!(defer && flush_early_ && layer_stage && stage + 1 < stage_end && stage + 1 <= d.NL)) RUN(flush_wgrads());
  }
  // The caller's stream is made to wait for the side streams only when the LAST stage has been enqueued.  After an
  // earlier stage range the weight gradients of those stages may still be running on the side streams: whoever
  // consumes them (the gradient exchange) orders itself behind them with petr_ctx_join_into(), and the dependent
  // chain of input-gradient kernels on the caller's stream goes on without stopping at every stage boundary.
  if (stage_end == P.n_stages) ln.join_all();
  return PETR_OK;
}
