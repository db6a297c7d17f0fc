// Training loss of PETRHead on the device: match cost, Hungarian assignment, focal + L1 loss and their gradients
// for all decoder levels and samples in four launches, without a host round trip, plus the
// box part of the NMS-free decode (rank 2).
//
// Replaces (reference projects/mmdet3d_plugin/): PETRHead.loss / loss_single / get_targets / _get_target_single
// (models/dense_heads/petr_head.py:470-728), HungarianAssigner3D.assign (core/bbox/assigners/
// hungarian_assigner_3d.py:61-143: cost on the device, `.cpu()`, scipy linear_sum_assignment, back to the device -
// once per level and sample), FocalLossCost / BBox3DL1Cost (core/bbox/match_costs/match_cost.py:6-27 + mmdet),
// normalize_bbox / denormalize_bbox (core/bbox/util.py:38-87), mmdet FocalLoss / L1Loss, and
// NMSFreeCoder.decode_single's box arithmetic (core/bbox/coders/nms_free_coder.py:62-97).
//
//   loss_cost_kernel   cost[p][g][q] (double) for every problem p = (level, sample): focal cost of query q for the
//                      label of gt g + L1 distance on the 8 normalised box dims, nan_to_num(100, +-100)
//   loss_lsa_kernel    ONE WAVE per problem: shortest-augmenting-path assignment (the algorithm of scipy's
//                      rectangular_lsap: duals u/v, Dijkstra over the columns).  Columns (queries) live on the
//                      lanes, 16 per lane (Q <= 1024), the per-step argmin is a wave reduction: no barrier, no LDS
//                      traffic but the row duals.  Ties on equal path costs prefer an unassigned column, then the
//                      lowest column index (scipy: an order that depends on its internal permutation; exact ties
//                      between float costs do not occur with real predictions).
//   loss_main_kernel   thread per (level, sample, query): focal loss over the classes, L1 over the 10 code dims for
//                      assigned queries with finite normalised targets, their gradients, double-precision sums
//   loss_final_kernel  losses[level][2] = nan_to_num(sum); a non-finite level loss gets zero gradients (the
//                      derivative of torch.nan_to_num at a non-finite input)
// Normalisers: with one assignment per gt, num_total_pos = sum_b min(G_b, Q) is known on the host, so
// avg factors are launch constants: cls: max(pos + neg*bg_cls_weight, 1), bbox: max(pos, 1), both + float eps
// (mmdet weight_reduce_loss).
#include "common.h"

namespace {

constexpr int LSA_COLS = 16;            // columns per lane: Q <= 64 * 16
constexpr double LSA_INF = 1e300;

constexpr int LOSS_MAX_B = 64;

struct LossParams {
  petr_loss_args a;
  int offs[LOSS_MAX_B + 1];   // gt_offsets, copied from HOST memory at launch (no device round trip for metadata)
  double* cost;          // [NL][Gtot][Q]
  double* sums;          // [NL][2]
  float cls_norm, box_norm;   // loss_weight / (avg_factor + eps); superseded by a.avg_factors (device) when given
};

// loss_weight / (max(avg_factor, 1) + eps): from the kernel arguments, or from the caller's cross-rank means in
// device memory (petr_loss_args.avg_factors)
__device__ __forceinline__ float norm_of(const LossParams& p, int which) {
  if (!p.a.avg_factors) return which == 0 ? p.cls_norm : p.box_norm;
  const double w = which == 0 ? (double)p.a.cls_weight : (double)p.a.bbox_weight;
  return (float)(w / ((double)fmaxf(p.a.avg_factors[which], 1.f) + 1.1920928955078125e-07));
}

__device__ __forceinline__ void normalize_gt(const float* g, float* n) {   // util.py:38-58
  n[0] = g[0]; n[1] = g[1]; n[2] = logf(g[3]); n[3] = logf(g[4]); n[4] = g[2]; n[5] = logf(g[5]);
  n[6] = sinf(g[6]); n[7] = cosf(g[6]); n[8] = g[7]; n[9] = g[8];
}

__global__ __launch_bounds__(256) void loss_cost_kernel(const LossParams p) {
  const petr_loss_args& a = p.a;
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;                 // global gt index
  const int lvl = blockIdx.z;
  if (q >= a.Q) return;
  int b = 0;
  while (b + 1 < a.B && g >= p.offs[b + 1]) ++b;
  const float* cls = a.cls + (((long)lvl * a.B + b) * a.Q + q) * a.NC;
  const float* box = a.box + (((long)lvl * a.B + b) * a.Q + q) * a.CS;
  const long label = a.gt_labels[g];
  // A label outside [0, NC) (mmdet3d yields -1 for classes missing from CLASSES when ObjectNameFilter is absent) must
  // not become an address: the cost of such a box is the nan_to_num value 100 for every query, the loss kernel treats
  // its query as matching no class, and a flag word behind the level sums reports it (losses.loss_label_errors()).
  if (label < 0 || label >= a.NC) {
    if (q == 0 && lvl == 0) atomicOr(reinterpret_cast<int*>(p.sums + 2 * a.NL), 1);
    p.cost[((long)lvl * a.Gtot + g) * a.Q + q] = 100.0;
    return;
  }
  float n[10];
  normalize_gt(a.gt_boxes + (long)g * 9, n);
  // FocalLossCost (mmdet): logits in, eps 1e-12
  const float pr = 1.f / (1.f + expf(-cls[label]));
  const float neg = -logf(1.f - pr + 1e-12f) * (1.f - a.alpha) * powf(pr, a.gamma);
  const float pos = -logf(pr + 1e-12f) * a.alpha * powf(1.f - pr, a.gamma);
  float c = (pos - neg) * a.cls_weight;
  float l1 = 0.f;
#pragma unroll
  for (int d = 0; d < 8; ++d) l1 += fabsf(box[d] - n[d]);
  c += l1 * a.bbox_weight;
  if (c != c) c = 100.f;                    // torch.nan_to_num(cost, nan=100, posinf=100, neginf=-100)
  else if (c == INFINITY) c = 100.f;
  else if (c == -INFINITY) c = -100.f;
  p.cost[((long)lvl * a.Gtot + g) * a.Q + q] = (double)c;
}

// wave-wide argmin with the tie rule (value, then unassigned first, then lowest column).  Only the VALUE travels
// through the cross-lane network (six DPP/permute steps of one double); the winner among equal values is found
// with two ballots, and its index comes back with one readlane.
struct Best { double v; int un; int j; };
__device__ __forceinline__ Best better(const Best& x, const Best& y) {
  if (x.v < y.v) return x;
  if (y.v < x.v) return y;
  if (x.un != y.un) return x.un ? x : y;
  return x.j <= y.j ? x : y;
}
__device__ __forceinline__ Best wave_best(Best b) {
  double m = b.v;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmin(m, __shfl_xor(m, o, 64));
  const bool tie = b.v == m;
  const unsigned long long un_mask = __ballot(tie && b.un);
  const bool cand = un_mask ? (tie && b.un) : tie;                     // unassigned columns first
  // lowest column index among the candidates: columns are lane + 64k, so compare indices, one more reduction of ints
  int j = cand ? b.j : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) j = min(j, __shfl_xor(j, o, 64));
  Best r;
  r.v = m; r.un = un_mask != 0; r.j = j;
  return r;
}

__global__ __launch_bounds__(64) void loss_lsa_kernel(const LossParams p) {
  extern __shared__ double lds[];           // u[Gmax] | col4row[Gmax] (int) | sr_list[Gmax] (int)
  const petr_loss_args& a = p.a;
  const int b = blockIdx.x, lvl = blockIdx.y, lane = threadIdx.x;
  const int g0 = p.offs[b], nr = min(p.offs[b + 1] - g0, a.Q), nc = a.Q;
  int* assigned = a.assigned + ((long)lvl * a.B + b) * a.Q;
  for (int j = lane; j < nc; j += 64) assigned[j] = 0;
  if (nr <= 0) return;
  double* u = lds;
  int* col4row = reinterpret_cast<int*>(lds + a.Gmax);
  int* sr_list = col4row + a.Gmax;
  for (int i = lane; i < nr; i += 64) { u[i] = 0.0; col4row[i] = -1; }
  // (one wave per problem: its LDS accesses execute in program order, so no barrier is needed anywhere below)
  const double* cost = p.cost + ((long)lvl * a.Gtot + g0) * a.Q;

  double v[LSA_COLS], sp[LSA_COLS];         // column duals, shortest path costs
  int row4col[LSA_COLS], path[LSA_COLS];
  unsigned sc = 0;                          // bit k: column lane + 64k is in SC (scanned)
#pragma unroll
  for (int k = 0; k < LSA_COLS; ++k) { v[k] = 0.0; row4col[k] = -1; path[k] = -1; }

  for (int cur = 0; cur < nr; ++cur) {
    double min_val = 0.0;
    int i = cur, n_sr = 0, sink = -1;
    sc = 0;
#pragma unroll
    for (int k = 0; k < LSA_COLS; ++k) sp[k] = LSA_INF;
    while (sink < 0 && n_sr < nr) {          // a search visits every row at most once: bounded by construction
      if (lane == 0) sr_list[n_sr] = i;
      ++n_sr;
      const double ui = u[i];
      const double* row = cost + (long)i * nc;
      Best best;
      best.v = LSA_INF; best.un = 0; best.j = 0x7fffffff;
      // the row's costs first, unconditionally and from clamped addresses: a load inside the divergent branch below
      // would be waited for on the spot, i.e. sixteen serial L2 round trips per step instead of one
      double cr[LSA_COLS];
#pragma unroll
      for (int k = 0; k < LSA_COLS; ++k) cr[k] = row[min(lane + 64 * k, nc - 1)];
#pragma unroll
      for (int k = 0; k < LSA_COLS; ++k) {
        const int j = lane + 64 * k;
        if (j < nc && !((sc >> k) & 1u)) {
          const double r = min_val + cr[k] - ui - v[k];
          if (r < sp[k]) { sp[k] = r; path[k] = i; }
          Best c;
          c.v = sp[k]; c.un = row4col[k] < 0; c.j = j;
          best = better(best, c);
        }
      }
      best = wave_best(best);
      min_val = best.v;
      if (!(min_val < LSA_INF)) { sink = -2; break; }   // infeasible: cannot happen after nan_to_num
      const int jb = best.j, kb = jb >> 6, owner = jb & 63;
      // the owner lane's row4col of the chosen column, for everybody
      int r4 = -1;
#pragma unroll
      for (int k = 0; k < LSA_COLS; ++k)
        if (k == kb) r4 = row4col[k];
      r4 = __shfl(r4, owner, 64);
      if (lane == owner) sc |= 1u << kb;
      if (r4 < 0) sink = jb;
      else i = r4;
    }
    if (sink < 0) break;
    // dual updates: u over the scanned rows, v over the scanned columns
    if (lane == 0) u[cur] += min_val;
    // u[i] += min_val - sp[col4row[i]] needs another lane's sp: gather through shuffles, one scanned row at a time
    for (int t = 0; t < n_sr; ++t) {
      const int r = sr_list[t];
      if (r == cur) continue;
      const int jc = col4row[r], kc = jc >> 6, oc = jc & 63;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < LSA_COLS; ++k)
        if (k == kc) s = sp[k];
      s = __shfl(s, oc, 64);
      if (lane == 0) u[r] += min_val - s;
    }
#pragma unroll
    for (int k = 0; k < LSA_COLS; ++k)
      if ((sc >> k) & 1u) v[k] -= min_val - sp[k];
    // augment along the path from the sink back to cur
    int j = sink;
    for (int step = 0; step <= nr; ++step) {   // a path has at most nr edges
      const int kj = j >> 6, oj = j & 63;
      int pi = -1;
#pragma unroll
      for (int k = 0; k < LSA_COLS; ++k)
        if (k == kj) pi = path[k];
      pi = __shfl(pi, oj, 64);
      if (lane == oj) {
#pragma unroll
        for (int k = 0; k < LSA_COLS; ++k)
          if (k == kj) row4col[k] = pi;
      }
      const int prev = col4row[pi];
      if (lane == 0) col4row[pi] = j;
      j = prev;
      if (pi == cur) break;
    }
  }
  // assigned_gt_inds: 0 = background, k = gt k-1 (hungarian_assigner_3d.py:137-139)
#pragma unroll
  for (int k = 0; k < LSA_COLS; ++k) {
    const int j = lane + 64 * k;
    if (j < nc && row4col[k] >= 0) assigned[j] = row4col[k] + 1;
  }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void loss_main_kernel(const LossParams p) {
  const petr_loss_args& a = p.a;
  const long n = (long)a.NL * a.B * a.Q;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  double s_cls = 0.0, s_box = 0.0;
  int lvl = 0;
  const float cls_norm = norm_of(p, 0), box_norm = norm_of(p, 1);
  if (idx < n) {
    lvl = (int)(idx / ((long)a.B * a.Q));
    const int b = (int)((idx / a.Q) % a.B);
    const int asg = a.assigned[idx];
    const int g = asg > 0 ? p.offs[b] + asg - 1 : -1;
    long label = g >= 0 ? a.gt_labels[g] : a.NC;            // num_classes = background (petr_head.py:507-510)
    if (label < 0 || label > a.NC) label = a.NC;            // out-of-range label (flagged by loss_cost_kernel): no positive class
    const float* cls = a.cls + idx * a.NC;
    float* dc = a.d_cls ? a.d_cls + idx * a.NC : nullptr;
    for (int c = 0; c < a.NC; ++c) {
      // mmdet py_sigmoid_focal_loss: BCE_with_logits(x, t) * (alpha t + (1-alpha)(1-t)) * pt^gamma
      const float x = cls[c];
      const float pr = 1.f / (1.f + expf(-x));
      const float sp_neg = x > 0.f ? x + log1pf(expf(-x)) : log1pf(expf(x));     // softplus(x)  = -log(1 - p)
      const float sp_pos = sp_neg - x;                                            // softplus(-x) = -log(p)
      float loss, grad;
      if (c == label) {
        const float w = a.alpha * powf(1.f - pr, a.gamma);
        loss = w * sp_pos;
        grad = w * (a.gamma * pr * (-sp_pos) - (1.f - pr));
      } else {
        const float w = (1.f - a.alpha) * powf(pr, a.gamma);
        loss = w * sp_neg;
        grad = w * (pr + a.gamma * (1.f - pr) * sp_neg);
      }
      s_cls += (double)loss;
      if (dc) dc[c] = grad * cls_norm;
    }
    const float* box = a.box + idx * a.CS;
    float* db = a.d_box ? a.d_box + idx * a.CS : nullptr;
    bool use = false;
    float n10[10];
    if (g >= 0) {
      normalize_gt(a.gt_boxes + (long)g * 9, n10);
      use = true;
#pragma unroll
      for (int d = 0; d < 10; ++d) use = use && isfinite(n10[d]);     // isnotnan row filter (petr_head.py:635-636)
    }
    for (int d = 0; d < a.CS; ++d) {
      float gd = 0.f;
      if (use && d < 10) {
        const float diff = box[d] - n10[d];
        s_box += (double)(fabsf(diff) * a.code_weights[d]);
        gd = (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * a.code_weights[d] * box_norm;
      }
      if (db) db[d] = gd;
    }
  }
  // one block can straddle two levels: reduce per wave only when the whole wave shares a level, else per thread
  const int lvl0 = __shfl(lvl, 0, 64);
  const bool uniform = __all(lvl == lvl0 || idx >= n);
  if (uniform) {
    s_cls = wave_sum_d(s_cls);
    s_box = wave_sum_d(s_box);
    if ((threadIdx.x & 63) == 0 && (s_cls != 0.0 || s_box != 0.0 || idx < n)) {
      atomicAdd(p.sums + 2 * lvl0, s_cls);
      atomicAdd(p.sums + 2 * lvl0 + 1, s_box);
    }
  } else if (idx < n) {
    atomicAdd(p.sums + 2 * lvl, s_cls);
    atomicAdd(p.sums + 2 * lvl + 1, s_box);
  }
}

__global__ __launch_bounds__(256) void loss_final_kernel(const LossParams p) {
  const petr_loss_args& a = p.a;
  const int lvl = blockIdx.x;
  const float lc = (float)(p.sums[2 * lvl] * (double)norm_of(p, 0));
  const float lb = (float)(p.sums[2 * lvl + 1] * (double)norm_of(p, 1));
  const bool okc = isfinite(lc), okb = isfinite(lb);
  if (threadIdx.x == 0) {
    a.losses[2 * lvl] = okc ? lc : (lc != lc ? 0.f : (lc > 0.f ? 3.402823466e+38f : -3.402823466e+38f));
    a.losses[2 * lvl + 1] = okb ? lb : (lb != lb ? 0.f : (lb > 0.f ? 3.402823466e+38f : -3.402823466e+38f));
  }
  const long per = (long)a.B * a.Q;
  if (!okc && a.d_cls)
    for (long i = threadIdx.x; i < per * a.NC; i += 256) a.d_cls[(long)lvl * per * a.NC + i] = 0.f;
  if (!okb && a.d_box)
    for (long i = threadIdx.x; i < per * a.CS; i += 256) a.d_box[(long)lvl * per * a.CS + i] = 0.f;
}

__global__ __launch_bounds__(256) void decode_boxes_kernel(petr_decode_args a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const long q = a.index[i] / a.num_classes;
  const float* s = a.bbox_preds + q * a.code;
  float* o = a.boxes + (long)i * 9;
  // denormalize_bbox (util.py:60-87) + gravity centre -> bottom centre (petr_head.py:745)
  const float w = expf(s[2]), l = expf(s[3]), h = expf(s[5]);
  const float cx = s[0], cy = s[1], cz = s[4];
  o[0] = cx; o[1] = cy; o[2] = a.bottom_center ? cz - h * 0.5f : cz; o[3] = w; o[4] = l; o[5] = h; o[6] = atan2f(s[6], s[7]);
  o[7] = a.code > 8 ? s[8] : 0.f; o[8] = a.code > 9 ? s[9] : 0.f;
  a.labels[i] = a.index[i] % a.num_classes;
  bool keep = cx >= a.post_center_range[0] && cy >= a.post_center_range[1] && cz >= a.post_center_range[2] &&
              cx <= a.post_center_range[3] && cy <= a.post_center_range[4] && cz <= a.post_center_range[5];
  if (a.score_threshold >= 0.f) keep = keep && a.scores[i] > a.score_threshold;
  a.keep[i] = keep ? 1 : 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// NMSFreeCoder.decode_single with the selection on the device (nms_free_coder.py:62-97): sigmoid + top-k of the Q*NC
// class scores + gather + box denormalisation + range filter as ONE launch, one workgroup per sample.
//   selection: 4-pass radix select (8 bits per pass, MSB first, histogram in LDS) over the order-preserving integer image
//   of the LOGITS (sigmoid is monotonic: the top-k of the scores is the top-k of the logits; the score is taken
//   afterwards from the selected logits only), then compaction of everything above the k-th key plus the lowest-index
//   ties, then a bitonic sort of the <= 1024 selected (key descending, index ascending) - torch.topk's sorted order.
// ---------------------------------------------------------------------------------------------------------------
constexpr int TOPK_MAX = 1024;

__device__ __forceinline__ uint32_t ordered_key(float f) {      // larger float <=> larger unsigned key (NaN above +inf)
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void decode_topk_kernel(petr_decode_topk_args a) {
  __shared__ uint32_t hist[256];
  __shared__ unsigned long long sel[TOPK_MAX];
  __shared__ uint32_t sh_prefix, sh_need, sh_count, sh_ties;
  const int t = threadIdx.x;
  const int smp = blockIdx.x;
  const int n = a.Q * a.num_classes;
  const float* logits = a.cls_scores + (long)smp * n;
  const int k = a.k < n ? a.k : n;

  // ---- radix select of the k-th largest key ----
  uint32_t prefix = 0, need = (uint32_t)k;        // need: how many elements we still take among those matching `prefix`
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (t < 256) hist[t] = 0;
    __syncthreads();
    const uint32_t himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = t; i < n; i += 1024) {
      const uint32_t key = ordered_key(logits[i]);
      if ((key & himask) == (prefix & himask)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (t == 0) {
      uint32_t acc = 0;
      int bin = 255;
      for (; bin > 0; --bin) {                   // from the top: the bin in which the need-th largest element lies
        if (acc + hist[bin] >= need) break;
        acc += hist[bin];
      }
      sh_prefix = prefix | ((uint32_t)bin << shift);
      sh_need = need - acc;
    }
    __syncthreads();
    prefix = sh_prefix;
    need = sh_need;
    __syncthreads();
  }
  // prefix = the k-th largest key T; `need` of the elements equal to T are taken (lowest indices first)
  if (t == 0) { sh_count = 0; sh_ties = 0; }
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {         // index order, so that the ties taken are the lowest indices
    const int i = i0 + t;
    const uint32_t key = i < n ? ordered_key(logits[i]) : 0u;
    const bool gt = i < n && key > prefix;
    const bool eq = i < n && key == prefix;
    if (gt) sel[atomicAdd(&sh_count, 1u)] = ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)i);
    // ties: rank inside this 1024-chunk by ballot-free counting through an LDS counter would lose the index order, so
    // one wave-ordered pass: lanes take their tie in index order with a prefix count over the wave
    const unsigned long long m = __ballot(eq);
    __shared__ uint32_t wave_base[16];
    const int wv = t >> 6, ln = t & 63;
    if (ln == 0) wave_base[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (t == 0) {
      uint32_t run = sh_ties;
      for (int w2 = 0; w2 < 16; ++w2) { const uint32_t c = wave_base[w2]; wave_base[w2] = run; run += c; }
      sh_ties = run;
    }
    __syncthreads();
    if (eq) {
      const uint32_t rank = wave_base[wv] + (uint32_t)__popcll(m & ((1ull << ln) - 1ull));
      if (rank < need) sel[atomicAdd(&sh_count, 1u)] = ((unsigned long long)key << 32) | (uint32_t)(~(uint32_t)i);
    }
    __syncthreads();
  }
  // ---- pad to a power of two and sort descending by (key, -index) ----
  int np2 = 1;
  while (np2 < k) np2 <<= 1;
  for (int i = k + t; i < np2; i += 1024) sel[i] = 0ull;
  __syncthreads();
  for (int size = 2; size <= np2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = t; i < np2; i += 1024) {
        const int j = i ^ stride;
        if (j > i) {
          const bool desc = (i & size) == 0;
          const unsigned long long x = sel[i], y = sel[j];
          if ((x < y) == desc) { sel[i] = y; sel[j] = x; }
        }
      }
      __syncthreads();
    }
  // ---- gather + decode (denormalize_bbox util.py:60-87, gravity -> bottom centre petr_head.py:745, range filter) ----
  for (int i = t; i < a.k; i += 1024) {
    const long o_i = (long)smp * a.k + i;
    float* o = a.boxes + o_i * 9;
    if (i >= k) {                                  // fewer candidates than k: padded, never kept
      for (int d = 0; d < 9; ++d) o[d] = 0.f;
      a.scores[o_i] = 0.f; a.labels[o_i] = 0; a.keep[o_i] = 0; a.index[o_i] = -1;
      continue;
    }
    const uint32_t idx = ~(uint32_t)(sel[i] & 0xFFFFFFFFull);
    const float logit = logits[idx];
    const float score = 1.f / (1.f + expf(-logit));
    const long q = idx / a.num_classes;
    const float* s = a.bbox_preds + ((long)smp * a.Q + q) * a.code;
    const float w = expf(s[2]), l = expf(s[3]), h = expf(s[5]);
    const float cx = s[0], cy = s[1], cz = s[4];
    o[0] = cx; o[1] = cy; o[2] = a.bottom_center ? cz - h * 0.5f : cz; o[3] = w; o[4] = l; o[5] = h; o[6] = atan2f(s[6], s[7]);
    o[7] = a.code > 8 ? s[8] : 0.f; o[8] = a.code > 9 ? s[9] : 0.f;
    bool keep = cx >= a.post_center_range[0] && cy >= a.post_center_range[1] && cz >= a.post_center_range[2] &&
                cx <= a.post_center_range[3] && cy <= a.post_center_range[4] && cz <= a.post_center_range[5];
    if (a.score_threshold >= 0.f) keep = keep && score > a.score_threshold;
    a.scores[o_i] = score;
    a.labels[o_i] = idx % a.num_classes;
    a.index[o_i] = idx;
    a.keep[o_i] = keep ? 1 : 0;
  }
}

extern "C" int petr_decode_topk(const petr_decode_topk_args* a, void* stream) {
  PETR_CHECK(a && a->cls_scores && a->bbox_preds && a->boxes && a->scores && a->labels && a->keep && a->index, PETR_ERR_INVALID,
             "decode_topk: null pointer");
  PETR_CHECK(a->B > 0 && a->Q > 0 && a->num_classes > 0 && a->code >= 8 && a->k > 0, PETR_ERR_INVALID, "decode_topk: bad shape");
  PETR_CHECK(a->k <= TOPK_MAX, PETR_ERR_UNSUPPORTED, "decode_topk: at most %d boxes per sample (max_num)", TOPK_MAX);
  PETR_CHECK((long)a->Q * a->num_classes < (1L << 31), PETR_ERR_UNSUPPORTED, "decode_topk: Q * num_classes too large");
  hipLaunchKernelGGL(decode_topk_kernel, dim3((unsigned)a->B), dim3(1024), 0, (hipStream_t)stream, *a);
  PETR_LAUNCH_CHECK("decode_topk");
  return PETR_OK;
}

extern "C" size_t petr_loss_workspace_bytes(int NL, int B, int Q, int Gtot) {
  (void)B;
  return ((size_t)NL * (Gtot > 0 ? Gtot : 1) * Q + 2 * (size_t)NL + 2) * sizeof(double);
}

extern "C" int petr_loss_fwd_bwd(const petr_loss_args* ap, void* stream) {
  PETR_CHECK(ap && ap->cls && ap->box && ap->gt_offsets && ap->losses && ap->assigned && ap->ws, PETR_ERR_INVALID,
             "loss: null pointer");
  PETR_CHECK(ap->Gtot == 0 || (ap->gt_boxes && ap->gt_labels), PETR_ERR_INVALID, "loss: gt pointers missing");
  PETR_CHECK(ap->NL > 0 && ap->B > 0 && ap->Q > 0 && ap->NC > 0 && ap->CS >= 10 && ap->Gtot >= 0 && ap->Gmax >= 0,
             PETR_ERR_INVALID, "loss: bad shape");
  PETR_CHECK(ap->B <= LOSS_MAX_B, PETR_ERR_UNSUPPORTED, "loss: at most %d samples per call", LOSS_MAX_B);
  PETR_CHECK(ap->Q <= 64 * LSA_COLS, PETR_ERR_UNSUPPORTED, "loss: the assignment kernel holds at most %d queries", 64 * LSA_COLS);
  PETR_CHECK(ap->Gmax <= ap->Q, PETR_ERR_UNSUPPORTED, "loss: more ground-truth boxes (%d) than queries (%d) in one sample",
             ap->Gmax, ap->Q);
  PETR_CHECK(ap->num_pos >= 0 && ap->num_pos <= (long)ap->B * ap->Q, PETR_ERR_INVALID, "loss: num_pos out of range");
  const size_t need = petr_loss_workspace_bytes(ap->NL, ap->B, ap->Q, ap->Gtot);
  PETR_CHECK(ap->ws_bytes >= need && ((uintptr_t)ap->ws & 7) == 0, PETR_ERR_WORKSPACE, "loss: workspace %zu < %zu bytes",
             ap->ws_bytes, need);
  const size_t lds = (size_t)(ap->Gmax > 0 ? ap->Gmax : 1) * (sizeof(double) + 2 * sizeof(int));
  PETR_CHECK(lds <= 60000, PETR_ERR_UNSUPPORTED, "loss: %d ground-truth boxes in one sample exceed the LDS budget", ap->Gmax);
  LossParams p;
  p.a = *ap;
  for (int b = 0; b <= ap->B; ++b) p.offs[b] = ap->gt_offsets[b];
  PETR_CHECK(p.offs[0] == 0 && p.offs[ap->B] == ap->Gtot, PETR_ERR_INVALID, "loss: gt_offsets must run from 0 to Gtot");
  for (int b = 0; b < ap->B; ++b)
    PETR_CHECK(p.offs[b + 1] >= p.offs[b] && p.offs[b + 1] - p.offs[b] <= ap->Gmax, PETR_ERR_INVALID,
               "loss: gt_offsets not monotonic or a sample exceeds Gmax");
  p.cost = (double*)ap->ws;
  p.sums = p.cost + (size_t)ap->NL * (ap->Gtot > 0 ? ap->Gtot : 1) * ap->Q;
  const double eps = 1.1920928955078125e-07;   // torch.finfo(float32).eps (mmdet weight_reduce_loss)
  const double num_neg = (double)ap->B * ap->Q - (double)ap->num_pos;
  double cls_avg = (double)ap->num_pos + num_neg * (double)ap->bg_cls_weight;
  if (cls_avg < 1.0) cls_avg = 1.0;
  const double box_avg = ap->num_pos < 1 ? 1.0 : (double)ap->num_pos;
  p.cls_norm = (float)((double)ap->cls_weight / (cls_avg + eps));
  p.box_norm = (float)((double)ap->bbox_weight / (box_avg + eps));
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(p.sums, 0, (2 * (size_t)ap->NL + 1) * sizeof(double), s);   // level sums + the label-error flag word
  PETR_CHECK(e == hipSuccess, PETR_ERR_LAUNCH, "loss: memset failed: %s", hipGetErrorString(e));
  if (ap->Gtot > 0) {
    hipLaunchKernelGGL(loss_cost_kernel, dim3((unsigned)cdiv(ap->Q, 256), (unsigned)ap->Gtot, (unsigned)ap->NL), dim3(256), 0, s, p);
    PETR_LAUNCH_CHECK("loss_cost");
  }
  hipLaunchKernelGGL(loss_lsa_kernel, dim3((unsigned)ap->B, (unsigned)ap->NL), dim3(64), lds, s, p);
  PETR_LAUNCH_CHECK("loss_lsa");
  hipLaunchKernelGGL(loss_main_kernel, dim3((unsigned)cdiv((long)ap->NL * ap->B * ap->Q, 256)), dim3(256), 0, s, p);
  PETR_LAUNCH_CHECK("loss_main");
  hipLaunchKernelGGL(loss_final_kernel, dim3((unsigned)ap->NL), dim3(256), 0, s, p);
  PETR_LAUNCH_CHECK("loss_final");
  return PETR_OK;
}

extern "C" int petr_decode_boxes(const petr_decode_args* a, void* stream) {
  PETR_CHECK(a && a->bbox_preds && a->index && a->scores && a->boxes && a->labels && a->keep, PETR_ERR_INVALID,
             "decode: null pointer");
  PETR_CHECK(a->n > 0 && a->num_classes > 0 && a->code >= 8, PETR_ERR_INVALID, "decode: bad shape");
  hipLaunchKernelGGL(decode_boxes_kernel, dim3((unsigned)cdiv(a->n, 256)), dim3(256), 0, (hipStream_t)stream, *a);
  PETR_LAUNCH_CHECK("decode_boxes");
  return PETR_OK;
}
