// fused.hip — the vectorised rollout frame in 4 launches (v2 of the hot loop), bit-identical to the unfused kernels.
//
// Why: the reference's AoS row (F = 3*Nmax+7 floats, 208 B at Nmax = 15) scatters the ~10 scalars a message needs over
// 2-3 cache lines, and eleven separate launches per frame each re-stream the whole state. Here
//   * a packed per-(env,node) HOT RECORD  rec0 = {head_id, head_dep, n, sel}, rec1 = {tail_id, head_arr, -, -}
//     and a per-node STATIC record st0 = {maxn, ff, road_index, cong} (shared by all environments) mirror the row, so
//     the gather kernels read 16 B per neighbour and never touch x;
//   * the Direction gather also emits post = {n', head', tail', chosen}: the state every row will have after the
//     Direction update, from which the Response "accepted" test is evaluated without a second pass over x;
//   * ONE row pass applies Direction update + Response pop + withdraw to x and refreshes the hot record;
//   * the live policy (logits = emb[road_index(dst)]), segment softmax, inverse-CDF sample, log-prob and the choice
//     phase are one launch; agent bookkeeping scans a 1-byte status + 4-byte departure SoA instead of 36-B AoS rows.
//   * the FIFO contents live in a slot-interleaved store  slots[b][i][s] = {id, arrival, departure}  (row stride padded
//     to 64 B) instead of the reference's three column blocks: the Direction update's unconditional per-row write is then
//     ONE 12-byte store instead of three dwords in three different DRAM sectors (+ counter), and `n` / `sel` live only
//     in the hot record. tarl_fused_export rebuilds the reference's x layout on demand.
//   * LAZY GARBAGE SLOT: a row that receives nobody still gets (0, t, t + tt) written into its first dead slot by the
//     reference. That value is never read by the simulation, is overwritten by the next frame's update (or
//     by an insertion) before anything can move it, and only shows in x. The row pass therefore just records
//     {flag, count-at-write} in rec1 and the export kernel materialises it (same fp32 expression, same slot). The one
//     case where the pop's "last slot keeps its value" rule would duplicate it (count == Nmax-1) is written eagerly.
// The packed state is authoritative between tarl_fused_pack and tarl_fused_export; the exported x and agent_features are
// bit-identical to what the unfused kernels (and the reference) produce after every frame; tests/test_gpu_fused.py
// checks that frame by frame.
//
// Domain: pure road graph (plan nodes == rows of x, ROAD_INDEX(i) == i is NOT assumed: the static record carries it).
// Counts that reach Nmax (outside the reference's defined domain) are not supported by this path.
#include <float.h>
#include <math.h>

#include "tarl_common.h"

#define FB 256
#define ENVB 1024
#define LOG_EPS_P 1e-8f
#define INS_CAP 2048   // LDS candidate list of the insert kernel (entries)

struct FusedBufs {
  float4* rec0;         // [B][N] {head_id, head_dep, n, sel}
  float4* rec1;         // [B][N] {tail_id, head_arr, n0 of the pending garbage write, pending flag}
  float4* post;         // [B][N] {n', head', tail', chosen}
  const float4* st0;    // [N]    {maxn, ff, road_index, cong}
  float* slots;         // [B][N][lds] slot-interleaved FIFO store: slot s at floats 3s..3s+2 = {id, arrival, departure}
  int64_t lds;          // row stride of slots in floats (>= 3*Nmax, multiple of 16)
  int32_t* a_origin;    // [B][A]
  int32_t* a_dest;      // [B][A]
  float* a_dep;         // [B][A]
  uint8_t* a_status;    // [B][A] 0 waiting, 1 on the way, 2 done
};

// ---- pack: build the hot / static records and the agent SoA from x / agent_features ----------------------------------
__global__ __launch_bounds__(FB) void k_pack_nodes(const float* __restrict__ x, Layout L, int64_t B, int64_t N,
                                                   const float* __restrict__ cong, FusedBufs fb, float4* st0_out) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (gid >= B * N) return;
  const int64_t b = gid / N;
  const int32_t i = (int32_t)(gid - b * N);
  const float* xi = x + b * L.bstride + (int64_t)i * L.ldx;
  const int Nmax = L.Nmax;
  const float n = xi[L.col_n()];
  const int q = (int)n;
  const float tail = (q >= 1 && q <= Nmax) ? xi[q - 1] : 0.0f;
  fb.rec0[gid] = make_float4(xi[0], xi[2 * Nmax], n, xi[L.col_sel()]);
  fb.rec1[gid] = make_float4(tail, xi[Nmax], 0.0f, 0.0f);
  fb.post[gid] = make_float4(n, xi[0], tail, 0.0f);
  float* sl = fb.slots + gid * fb.lds;
  for (int sidx = 0; sidx < Nmax; ++sidx) {
    sl[3 * sidx + 0] = xi[sidx];
    sl[3 * sidx + 1] = xi[Nmax + sidx];
    sl[3 * sidx + 2] = xi[2 * Nmax + sidx];
  }
  if (b == 0 && st0_out) {
    const float maxn = xi[L.col_maxn()], ff = xi[L.col_ff()];
    float c;
    if (cong) {
      c = cong[i];
    } else {
      const float critical = xi[L.col_maxflow()] * ff / 3600.0f;
      c = ff * (maxn + 10.0f - critical);
    }
    st0_out[i] = make_float4(maxn, ff, xi[L.col_road()], c);
  }
}

__global__ __launch_bounds__(FB) void k_pack_agents(const float* __restrict__ ag, int64_t B, int64_t A,
                                                    int64_t a_bstride, FusedBufs fb) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (gid >= B * A) return;
  const int64_t b = gid / A, a = gid - b * A;
  const float* row = ag + b * a_bstride + a * AG_COLS;
  fb.a_origin[gid] = (int32_t)(long long)row[AG_ORIGIN];
  fb.a_dest[gid] = (int32_t)(long long)row[AG_DEST];
  fb.a_dep[gid] = row[AG_DEP];
  fb.a_status[gid] = row[AG_DONE] != 0.0f ? 2 : (row[AG_ON_WAY] != 0.0f ? 1 : 0);
}

// ---- fused policy: logits -> segment softmax -> sample -> log_prob (+entropy) -> choice phase ---------------------------
// One workgroup per environment, same arithmetic and the same reduction trees as k_edge_logits_fwd + k_softmax +
// k_sample + k_logprob_entropy_fwd + k_apply_action.
__device__ __forceinline__ float fb_block_sum(float v, float* s_red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  __syncthreads();
  if (lane == 0) s_red[wid] = v;
  __syncthreads();
  float tot = 0.0f;
  for (int w = 0; w < ENVB / 64; ++w) tot += s_red[w];
  return tot;
}

__device__ __forceinline__ float live_logit(const float* __restrict__ emb, int64_t M, const float4* __restrict__ st0,
                                            int32_t dst) {
  const long long idx = (long long)st0[dst].z;
  return (idx >= 0 && idx < M) ? emb[idx] : 0.0f;
}

// The live policy's logits depend only on (emb, static ROAD_INDEX of the target road): they are identical for every
// environment and every frame between two optimiser steps. k_policy_tables therefore evaluates, ONCE per parameter
// update and with exactly the arithmetic / reduction trees of k_softmax + k_sample + k_logprob_entropy_fwd, the per-edge
// tables (CSR order): thr[k] = fp32 inverse-CDF threshold, lg[k] = log(p + 1e-8), plus the entropy; k_fused_choice then
// only draws one uniform per (environment, node), walks <= deg thresholds and reduces the log-prob (same tree).
__global__ __launch_bounds__(ENVB) void k_policy_tables(const int32_t* __restrict__ out_ptr,
                                                        const int32_t* __restrict__ out_dst,
                                                        const int32_t* __restrict__ node_of_group, int64_t N, int64_t G,
                                                        const float* __restrict__ emb, int64_t M, float temperature,
                                                        const float4* __restrict__ st0, double* __restrict__ base,
                                                        float* __restrict__ thr, float* __restrict__ lgt,
                                                        float* __restrict__ entropy_out) {
  __shared__ double s_wave[ENVB / 64];
  __shared__ float s_red[ENVB / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double running = 0.0;
  for (int64_t g0 = 0; g0 < G; g0 += ENVB) {
    const int64_t g = g0 + tid;
    double s = 0.0;
    if (g < G) {
      const int32_t i = node_of_group[g];
      const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];
      float mx = -INFINITY;
      for (int32_t k = k0; k < k1; ++k) mx = fmaxf(mx, live_logit(emb, M, st0, out_dst[k]) / temperature);
      float sum = 0.0f;
      for (int32_t k = k0; k < k1; ++k) sum = sum + expf(live_logit(emb, M, st0, out_dst[k]) / temperature - mx);
      for (int32_t k = k0; k < k1; ++k) s += (double)(expf(live_logit(emb, M, st0, out_dst[k]) / temperature - mx) / sum);
    }
    double inc = s;
    for (int off = 1; off < 64; off <<= 1) {
      const double v = __shfl_up(inc, off);
      if (lane >= off) inc += v;
    }
    if (lane == 63) s_wave[wid] = inc;
    __syncthreads();
    double wbase = 0.0, tot = 0.0;
    for (int w = 0; w < ENVB / 64; ++w) {
      const double v = s_wave[w];
      if (w < wid) wbase += v;
      tot += v;
    }
    double exc = __shfl_up(inc, 1);
    if (lane == 0) exc = 0.0;
    if (g < G) base[g] = running + wbase + exc;
    running += tot;
    __syncthreads();
  }
  float ent = 0.0f;
  for (int64_t i = tid; i < N; i += ENVB) {
    const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];
    if (k0 == k1) continue;
    float mx = -INFINITY;
    for (int32_t k = k0; k < k1; ++k) mx = fmaxf(mx, live_logit(emb, M, st0, out_dst[k]) / temperature);
    float sum = 0.0f;
    for (int32_t k = k0; k < k1; ++k) sum = sum + expf(live_logit(emb, M, st0, out_dst[k]) / temperature - mx);
    int64_t g = i;
    if (G != N) {
      int64_t lo = 0, hi = G - 1;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (node_of_group[mid] < i) lo = mid + 1; else hi = mid;
      }
      g = lo;
    }
    const double bg = base[g];
    const float bg32 = (float)bg;
    double run = bg;
    for (int32_t k = k0; k < k1; ++k) {
      const float p = expf(live_logit(emb, M, st0, out_dst[k]) / temperature - mx) / sum;
      const float lg = logf(p + LOG_EPS_P);
      ent -= p * lg;
      run += (double)p;
      thr[k] = (float)run - bg32;
      lgt[k] = lg;
    }
  }
  const float ent_t = fb_block_sum(ent, s_red);
  if (tid == 0) entropy_out[0] = ent_t;
}

__global__ __launch_bounds__(ENVB) void k_fused_choice(const int32_t* __restrict__ out_ptr,
                                                       const int32_t* __restrict__ out_dst,
                                                       const int32_t* __restrict__ out_eid,
                                                       const int32_t* __restrict__ group_of_node, int64_t N, int64_t G,
                                                       const float* __restrict__ thr, const float* __restrict__ lgt,
                                                       const float* __restrict__ entropy_in,
                                                       const float* __restrict__ uniform, uint64_t seed,
                                                       uint64_t counter, FusedBufs fb, int32_t* __restrict__ choice,
                                                       float* __restrict__ log_prob, float* __restrict__ entropy) {
  __shared__ float s_red[ENVB / 64];
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  float lp = 0.0f;
  bool bad = false;
  for (int64_t i = tid; i < N; i += ENVB) {
    const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];
    if (k0 == k1) {
      if (choice) choice[b * N + i] = -1;
      continue;
    }
    const int64_t g = group_of_node[i];
    const float u = uniform ? uniform[b * G + g] : philox_uniform(seed, counter, (uint64_t)(b * G + g));
    int32_t pick = -1;
    for (int32_t k = k0; k < k1; ++k) {
      if (u < thr[k]) {
        pick = k;
        break;
      }
    }
    if (pick >= 0) {
      lp += lgt[pick];
      float4 r = fb.rec0[b * N + i];                  // SELECTED_ROAD lives in the hot record (whole-record store:
      r.w = (float)out_dst[pick];                     // no partial-sector writes); x gets it at export
      fb.rec0[b * N + i] = r;
      if (choice) choice[b * N + i] = out_eid[pick];
    } else {
      bad = true;
      if (choice) choice[b * N + i] = -1;
    }
  }
  const float bad_t = fb_block_sum(bad ? 1.0f : 0.0f, s_red);
  const float lp_t = fb_block_sum(lp, s_red);
  if (tid == 0) {
    if (log_prob) log_prob[b] = bad_t > 0.0f ? -INFINITY : lp_t;
    if (entropy) entropy[b] = entropy_in[0];
  }
}

// ---- Direction gather on the hot records ---------------------------------------------------------------------------------
__global__ __launch_bounds__(FB) void k_fused_direction(const int32_t* __restrict__ in_ptr,
                                                        const int32_t* __restrict__ in_src,
                                                        const int32_t* __restrict__ in_eid, int64_t E, int64_t B,
                                                        int64_t N, FusedBufs fb, const float* __restrict__ edge_attr,
                                                        const float* __restrict__ log_edge_attr, float log_eps, float t,
                                                        const float* __restrict__ gumbel, uint64_t seed,
                                                        uint64_t counter, float* __restrict__ dtt) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (gid >= B * N) return;
  const int64_t b = gid / N;
  const int32_t i = (int32_t)(gid - b * N);
  const float4* r0b = fb.rec0 + b * N;
  const float4 me = r0b[i];
  const float4 sti = fb.st0[i];
  const float max_i = sti.x, n_i = me.z, road_i = sti.z;
  const float room_i = max_i - n_i;
  const bool has_room = n_i < max_i - TARL_CONGESTION_FILE;
  float P = 0.0f, best = -FLT_MAX, best_id = 0.0f;
  PhiloxRun rng;
  const int32_t k1 = in_ptr[i + 1];
  for (int32_t k = in_ptr[i]; k < k1; ++k) {
    const int32_t j = in_src[k];
    const int32_t e = in_eid[k];
    const float4 rj = r0b[j];
    const float4 stj = fb.st0[j];
    const float id = rj.x, dep = rj.y, n_j = rj.z, sel_j = rj.w, max_j = stj.x;
    const bool heads_here = sel_j == road_i;
    const bool m1 = (dep <= t) && has_room && heads_here && (n_j > 0.0f);
    const bool m2 = ((dep - t) < -10.0f) && ((max_j - TARL_CONGESTION_FILE) <= n_j) && ((max_j - n_j) <= room_i) &&
                    heads_here;
    const bool m = m1 || m2;
    const float prob = edge_attr[e] * (m ? 1.0f : 0.0f);
    P = P + prob;
    const int64_t ge = b * E + e;
    float g;
    if (gumbel) {
      g = gumbel[ge];
    } else {
      const float u = rng.uniform(seed, counter, (uint64_t)(b * E + k));
      g = -logf(-logf(u));
    }
    const float score = (m ? log_edge_attr[e] : log_eps) + g;
    if (score > best) {
      best = score;
      best_id = id;
    }
    if (dtt) {
      const float d = (dep - fb.rec1[b * N + j].y) - stj.y;
      dtt[ge] = d > 0.0f ? d : (d != d ? d : 0.0f);
    }
  }
  const float who = (P > 0.0f) ? best_id : 0.0f;
  const float4 r1 = fb.rec1[gid];
  fb.post[gid] = make_float4(who != 0.0f ? n_i + 1.0f : n_i, n_i == 0.0f ? who : me.x, who != 0.0f ? who : r1.x, who);
}

// ---- the row pass: Direction update + Response pop + withdraw on x, then refresh the hot record -------------------------
__global__ __launch_bounds__(FB) void k_fused_rows(const int32_t* __restrict__ out_ptr,
                                                   const int32_t* __restrict__ out_dst, int Nmax, int64_t B, int64_t N,
                                                   FusedBufs fb, float* __restrict__ ag, int64_t A, int64_t a_bstride,
                                                   float t, uint8_t* __restrict__ popped_out,
                                                   uint8_t* __restrict__ withdrawn_out) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (gid >= B * N) return;
  const int64_t b = gid / N;
  const int32_t i = (int32_t)(gid - b * N);
  float* sl = fb.slots + gid * fb.lds;  // slot s = sl[3s .. 3s+2] = {id, arrival, departure}
  const float4* pb = fb.post + b * N;
  const float4 p = pb[i];
  const float4 r0 = fb.rec0[gid];
  const float4 r1 = fb.rec1[gid];
  const float4 st = fb.st0[i];
  const float n0 = r0.z, who = p.w;
  const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];

  // Response message + max-aggregate from the post records (state after the Direction update of every row)
  bool pop = false;
  if ((long long)p.x > 0) {
    const long long head = (long long)p.y;
    for (int32_t k = k0; k < k1; ++k) {
      const float4 pj = pb[out_dst[k]];
      pop = pop || ((long long)pj.x > 0 && (long long)pj.z == head);
    }
  }
  if (popped_out) popped_out[gid] = pop ? 1 : 0;

  // Direction update (every row, also when nothing was chosen): one 12-byte store — or, for a row that received nobody,
  // a note in the hot record (lazy garbage slot, see the file header)
  const int q = (int)n0;
  const float t_cong = st.w / (st.x + 10.0f - n0);
  const float tt = (t_cong != t_cong) ? t_cong : fmaxf(st.y, t_cong);
  const float dep_new = t + tt;
  const bool lazy = (who == 0.0f) && (q >= 0) && (q < Nmax - 1);
  if (!lazy && q >= 0 && q < Nmax) {
    sl[3 * q + 0] = who;
    sl[3 * q + 1] = t;
    sl[3 * q + 2] = dep_new;
  }
  float n = p.x;  // count after the Direction update

  // head / tail of the row after the Direction update (no memory reads needed)
  float head_id = (n0 == 0.0f) ? who : r0.x;
  float head_dep = (n0 == 0.0f) ? dep_new : r0.y;
  float head_arr = (n0 == 0.0f) ? t : r1.y;
  float tail_id = p.z;

  int shift = 0;
  if (pop) {
    shift = 1;
    n = n - 1.0f;
  }
  // withdraw: leading run of the (popped) row
  int c = 0;
  if (n > 0.0f) {
    const long long road = (long long)st.z;
    int32_t w0 = 0, w1 = 0;
    if (road >= 0 && road < N) {
      w0 = out_ptr[road];
      w1 = out_ptr[road + 1];
    }
    for (int s = 0; s < Nmax && (float)s < n; ++s) {
      int src = s + shift;          // slot of the row as it is in memory right now
      if (src >= Nmax) src = Nmax - 1;  // the pop duplicates the last slot
      float idf, depf;
      if (src == 0) {
        idf = head_id;
        depf = head_dep;
      } else {
        idf = sl[3 * src];
        depf = sl[3 * src + 2];
      }
      const long long id = (long long)idf;
      if (id < 0 || id >= A) break;
      const long long dest = (long long)fb.a_dest[b * A + id];
      bool conn = false;
      for (int32_t k = w0; k < w1; ++k) conn = conn || ((long long)out_dst[k] == dest);
      if (!(conn && depf <= t)) break;
      float* a = ag + b * a_bstride + id * AG_COLS;
      a[AG_DONE] = 1.0f;
      a[AG_ON_WAY] = 0.0f;
      a[AG_ARR] = t;
      fb.a_status[b * A + id] = 2;
      ++c;
    }
  }
  if (withdrawn_out) withdrawn_out[gid] = c > 0 ? 1 : 0;

  if (shift + c > 0) {
    // pop (shift by one, last slot keeps its stale value) followed by withdraw (shift by c, zero fill), in one sweep
    const float l0 = sl[3 * (Nmax - 1)], l1 = sl[3 * (Nmax - 1) + 1], l2 = sl[3 * (Nmax - 1) + 2];
    for (int s = 0; s < Nmax; ++s) {
      float v0, v1, v2;
      int from;
      if (shift == 0) {
        from = (s + c < Nmax) ? s + c : -1;
      } else {
        const int k = s + c;  // index into the popped row
        from = (k < Nmax - 1) ? k + 1 : (k == Nmax - 1 ? -2 : -1);
      }
      if (from >= 0) {
        v0 = sl[3 * from];
        v1 = sl[3 * from + 1];
        v2 = sl[3 * from + 2];
      } else if (from == -2) {
        v0 = l0; v1 = l1; v2 = l2;
      } else {
        v0 = v1 = v2 = 0.0f;
      }
      sl[3 * s] = v0;
      sl[3 * s + 1] = v1;
      sl[3 * s + 2] = v2;
    }
    n = n - (float)c;
    if (lazy && n == 0.0f) {  // the row emptied: its head slot is the (unmaterialised) garbage slot
      head_id = 0.0f;
      head_arr = t;
      head_dep = dep_new;
    } else {
      head_id = sl[0];
      head_arr = sl[1];
      head_dep = sl[2];
    }
    const int qn = (int)n;
    tail_id = (qn >= 1 && qn <= Nmax) ? sl[3 * (qn - 1)] : 0.0f;
  }
  fb.rec0[gid] = make_float4(head_id, head_dep, n, r0.w);
  fb.rec1[gid] = make_float4(tail_id, head_arr, n0, lazy ? 1.0f : 0.0f);
}

// ---- export: rebuild the reference's x layout (three FIFO column blocks + NUMBER_OF_AGENT + SELECTED_ROAD) ----------
__global__ __launch_bounds__(FB) void k_export_rows(float* __restrict__ x, Layout L, int64_t B, int64_t N, FusedBufs fb,
                                                    float t_last) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;
  const int Nmax = L.Nmax;
  if (gid >= B * N * Nmax) return;
  const int64_t row = gid / Nmax;
  const int sidx = (int)(gid - row * Nmax);
  const int64_t b = row / N, i = row - b * N;
  float* xi = x + b * L.bstride + i * L.ldx;
  const float* sl = fb.slots + row * fb.lds + 3 * sidx;
  const float4 r0 = fb.rec0[row];
  const float4 r1 = fb.rec1[row];
  if (r1.w != 0.0f && sidx == (int)r0.z) {  // pending garbage write of the last Direction update -> first dead slot
    const float4 st = fb.st0[i];
    const float t_cong = st.w / (st.x + 10.0f - r1.z);
    const float tt = (t_cong != t_cong) ? t_cong : fmaxf(st.y, t_cong);
    xi[sidx] = 0.0f;
    xi[Nmax + sidx] = t_last;
    xi[2 * Nmax + sidx] = t_last + tt;
  } else {
    xi[sidx] = sl[0];
    xi[Nmax + sidx] = sl[1];
    xi[2 * Nmax + sidx] = sl[2];
  }
  if (sidx == 0) {
    xi[L.col_n()] = r0.z;
    xi[L.col_sel()] = r0.w;
  }
}

// ---- insert + reward + counts on the hot records / agent SoA ----------------------------------------------------------
__device__ __forceinline__ bool fused_target(const FusedBufs& fb, int64_t b, int64_t N, int32_t origin, int32_t* road,
                                             int32_t* cap) {
  if (origin < 0 || origin >= N) return false;
  const long long r = (long long)fb.rec0[b * N + origin].w;
  if (r < 0 || r >= N) return false;
  const long long room = (long long)(fb.st0[r].x - TARL_CONGESTION_FILE - fb.rec0[b * N + r].z);
  *road = (int32_t)r;
  *cap = (int32_t)(room > 0x7fffffff ? 0x7fffffff : room);
  return room > 0;
}

__global__ __launch_bounds__(ENVB) void k_fused_insert(int Nmax, int64_t N, FusedBufs fb,
                                                       float* __restrict__ ag, int64_t A, int64_t a_bstride,
                                                       int use_cong, float t, int32_t* __restrict__ scratch,
                                                       float* __restrict__ reward, float* __restrict__ counts) {
  __shared__ int32_t s_wave[ENVB / 64];
  __shared__ float s_red[ENVB / 64];
  __shared__ int32_t s_cnt;
  __shared__ int32_t s_un_agent[INS_CAP], s_un_road[INS_CAP];
  const int64_t b = blockIdx.x;
  float* agb = ag + b * a_bstride;
  int32_t* cand_agent = scratch + b * 2 * A;
  int32_t* cand_road = cand_agent + A;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  // phase 1: candidates (ready agent whose target road has room). Candidates are rare (a handful per frame), so they
  // are appended unordered to an LDS list with an atomic counter and then ranked by agent id (deterministic: the
  // reference admits agents in stable agent-id order). A backlog larger than the LDS list falls back to the ordered
  // ballot compaction into the global scratch.
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  for (int64_t a0 = tid; a0 < A; a0 += 4 * ENVB) {  // 4 independent (status, departure) loads in flight per thread
    uint8_t stt[4];
    float dp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t a = a0 + (int64_t)j * ENVB;
      stt[j] = a < A ? fb.a_status[b * A + a] : (uint8_t)1;
      dp[j] = a < A ? fb.a_dep[b * A + a] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (stt[j] == 0 && dp[j] <= t) {
        const int64_t a = a0 + (int64_t)j * ENVB;
        int32_t road = 0, cap = 0;
        if (fused_target(fb, b, N, fb.a_origin[b * A + a], &road, &cap)) {
          const int32_t pos = atomicAdd(&s_cnt, 1);
          if (pos < INS_CAP) {
            s_un_agent[pos] = (int32_t)a;
            s_un_road[pos] = road;
          }
        }
      }
    }
  }
  __syncthreads();
  int32_t Lc = s_cnt;
  if (Lc <= INS_CAP) {
    for (int32_t idx = tid; idx < Lc; idx += ENVB) {
      const int32_t a = s_un_agent[idx];
      int32_t pos = 0;
      for (int32_t k = 0; k < Lc; ++k) pos += (s_un_agent[k] < a) ? 1 : 0;
      cand_agent[pos] = a;
      cand_road[pos] = s_un_road[idx];
    }
    __threadfence_block();
    __syncthreads();
  } else {
    int32_t basec = 0;
    for (int64_t a0 = 0; a0 < A; a0 += ENVB) {
      const int64_t a = a0 + tid;
      bool cnd = false;
      int32_t road = 0, cap = 0;
      if (a < A && fb.a_status[b * A + a] == 0 && fb.a_dep[b * A + a] <= t)
        cnd = fused_target(fb, b, N, fb.a_origin[b * A + a], &road, &cap);
      const unsigned long long bal = __ballot(cnd);
      const int lane_off = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) s_wave[wid] = __popcll(bal);
      __syncthreads();
      int32_t wbase = 0, tot = 0;
      for (int w = 0; w < ENVB / 64; ++w) {
        const int32_t v = s_wave[w];
        if (w < wid) wbase += v;
        tot += v;
      }
      if (cnd) {
        cand_agent[basec + wbase + lane_off] = (int32_t)a;
        cand_road[basec + wbase + lane_off] = road;
      }
      basec += tot;
      __syncthreads();
    }
    Lc = basec;
    __threadfence_block();
    __syncthreads();
  }

  for (int32_t idx = tid; idx < Lc; idx += ENVB) {
    const int32_t r = cand_road[idx];
    const int32_t a = cand_agent[idx];
    int32_t rank = 0, total = 0;
    for (int32_t k = 0; k < Lc; ++k) {
      const bool same = cand_road[k] == r;
      total += same ? 1 : 0;
      rank += (same && k < idx) ? 1 : 0;
    }
    const float4 str = fb.st0[r];
    const float n0 = fb.rec0[b * N + r].z;
    const long long cap = (long long)(str.x - TARL_CONGESTION_FILE - n0);
    int32_t commit = 0;
    if (rank < cap) {
      const long long m = total < cap ? total : cap;  // arrivals admitted on this road
      const long long slot = (long long)n0 + rank;
      const float t_cong = use_cong ? str.w / (str.x + 10.0f - (float)(long long)n0) : 0.0f;
      const float tt = (t_cong != t_cong) ? t_cong : fmaxf(str.y, t_cong);
      if (slot >= 0 && slot < Nmax) {
        float* sr = fb.slots + (b * N + r) * fb.lds + 3 * slot;
        sr[0] = (float)a;
        sr[1] = t;
        sr[2] = t + tt;
      }
      agb[(int64_t)a * AG_COLS + AG_ON_WAY] = 1.0f;
      fb.a_status[b * A + a] = 1;
      // hot record: only fields nobody reads in this phase (n is committed after the barrier)
      if (rank == 0 && n0 == 0.0f) {
        fb.rec0[b * N + r].x = (float)a;
        fb.rec0[b * N + r].y = t + tt;
        fb.rec1[b * N + r].y = t;
      }
      if (rank == m - 1) fb.rec1[b * N + r].x = (float)a;
      if (rank == 0) fb.rec1[b * N + r].w = 0.0f;  // the arrivals overwrite a pending garbage slot
      if (rank == 0) commit = (int32_t)m;
    }
    cand_agent[idx] = commit;
  }
  __threadfence_block();
  __syncthreads();
  for (int32_t idx = tid; idx < Lc; idx += ENVB) {
    const int32_t cmt = cand_agent[idx];
    if (cmt > 0) {
      const int32_t r = cand_road[idx];
      fb.rec0[b * N + r].z = fb.rec0[b * N + r].z + (float)cmt;
    }
  }
  __threadfence_block();
  __syncthreads();
  if (reward || counts) {
    float acc = 0.0f;
    for (int64_t i = tid; i < N; i += ENVB) {
      const float v = fb.rec0[b * N + i].z;
      if (counts) counts[b * N + i] = v;
      acc += v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) s_red[wid] = acc;
    __syncthreads();
    if (tid == 0 && reward) {
      float tot = 0.0f;
      for (int w = 0; w < ENVB / 64; ++w) tot += s_red[w];
      reward[b] = -tot;
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
hipEvent_t tarl_prof_event(hipStream_t s);  // sim.hip: live timing of the message-passing gather kernel

static FusedBufs to_bufs(const tarl_fused* f) {
  return FusedBufs{(float4*)f->rec0, (float4*)f->rec1, (float4*)f->post, (const float4*)f->st0, f->slots, f->ld_slots,
                   f->a_origin,      f->a_dest,        f->a_dep,         f->a_status};
}

static int check_fused_core(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax) {
  TARL_REQUIRE(plan && f, "null argument");
  TARL_REQUIRE(f->rec0 && f->rec1 && f->post && f->st0 && f->slots, "fused node buffers missing");
  TARL_REQUIRE(B >= 1 && B < ((int64_t)1 << 31) && Nmax >= 2, "bad sizes");
  TARL_REQUIRE(f->ld_slots >= 3 * (int64_t)Nmax, "slot row stride smaller than 3*Nmax");
  TARL_REQUIRE(((uintptr_t)f->rec0 | (uintptr_t)f->rec1 | (uintptr_t)f->post | (uintptr_t)f->st0) % 16 == 0,
               "fused records must be 16-byte aligned");
  return TARL_OK;
}

static int check_fused(const tarl_plan* plan, const tarl_fused* f, const float* x, int64_t B, int64_t bstride,
                       int64_t ldx, int32_t Nmax) {
  int rc = check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(x != nullptr, "x is null");
  TARL_REQUIRE(ldx >= 3 * (int64_t)Nmax + 7, "row stride smaller than F");
  TARL_REQUIRE(B == 1 || bstride >= plan->N * ldx, "environment stride smaller than one environment");
  return TARL_OK;
}

extern "C" int tarl_fused_pack(const tarl_plan* plan, const tarl_fused* f, const float* x, int64_t B, int64_t x_bstride,
                               int64_t ldx, int32_t Nmax, const float* cong, const float* agent_features, int64_t A,
                               int64_t a_bstride, tarl_stream stream) {
  int rc = check_fused(plan, f, x, B, x_bstride, ldx, Nmax);
  if (rc) return rc;
  const Layout L{Nmax, ldx, x_bstride};
  const FusedBufs fb = to_bufs(f);
  hipStream_t s = (hipStream_t)stream;
  if (plan->N > 0) {
    hipLaunchKernelGGL(k_pack_nodes, dim3((unsigned)ceil_div(B * plan->N, FB)), dim3(FB), 0, s, x, L, B, plan->N, cong,
                       fb, (float4*)f->st0);
    TARL_LAUNCH_CHECK();
  }
  if (agent_features) {
    TARL_REQUIRE(f->a_origin && f->a_dest && f->a_dep && f->a_status && A >= 1, "fused agent buffers missing");
    hipLaunchKernelGGL(k_pack_agents, dim3((unsigned)ceil_div(B * A, FB)), dim3(FB), 0, s, agent_features, B, A,
                       a_bstride, fb);
    TARL_LAUNCH_CHECK();
  }
  return TARL_OK;
}

extern "C" int tarl_fused_policy_prepare(const tarl_plan* plan, const tarl_fused* f, const float* emb,
                                         int64_t num_embeddings, float temperature, double* group_base,
                                         float* thresholds, float* log_probs, float* entropy1, tarl_stream stream) {
  TARL_REQUIRE(plan && f && f->st0 && emb && group_base && thresholds && log_probs && entropy1, "null argument");
  TARL_REQUIRE(num_embeddings >= 1, "bad sizes");
  if (plan->N == 0) return TARL_OK;
  hipLaunchKernelGGL(k_policy_tables, dim3(1), dim3(ENVB), 0, (hipStream_t)stream, plan->out_ptr, plan->out_dst,
                     plan->node_of_group, plan->N, plan->G, emb, num_embeddings, temperature, (const float4*)f->st0,
                     group_base, thresholds, log_probs, entropy1);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_fused_policy_step(const tarl_plan* plan, const tarl_fused* f, int64_t B, const float* thresholds,
                                      const float* log_probs, const float* entropy1, const float* uniform,
                                      uint64_t seed, uint64_t counter, int32_t* choice, float* log_prob, float* entropy,
                                      tarl_stream stream) {
  TARL_REQUIRE(plan && f && f->rec0 && thresholds && log_probs && entropy1, "null argument");
  TARL_REQUIRE(B >= 1 && B < ((int64_t)1 << 31), "bad B");
  if (plan->N == 0) return TARL_OK;
  hipLaunchKernelGGL(k_fused_choice, dim3((unsigned)B), dim3(ENVB), 0, (hipStream_t)stream, plan->out_ptr, plan->out_dst,
                     plan->out_eid, plan->group_of_node, plan->N, plan->G, thresholds, log_probs, entropy1, uniform, seed,
                     counter, to_bufs(f), choice, log_prob, entropy);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_fused_export(const tarl_plan* plan, const tarl_fused* f, float* x, int64_t B, int64_t x_bstride,
                                 int64_t ldx, int32_t Nmax, float last_step_time, tarl_stream stream) {
  int rc = check_fused(plan, f, x, B, x_bstride, ldx, Nmax);
  if (rc) return rc;
  if (plan->N == 0) return TARL_OK;
  const Layout L{Nmax, ldx, x_bstride};
  hipLaunchKernelGGL(k_export_rows, dim3((unsigned)ceil_div(B * plan->N * Nmax, FB)), dim3(FB), 0, (hipStream_t)stream,
                     x, L, B, plan->N, to_bufs(f), last_step_time);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_fused_env_step(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax,
                                   float* agent_features, int64_t A, int64_t a_bstride, const float* edge_attr,
                                   const float* log_edge_attr, float log_eps, int use_cong, float time,
                                   const float* gumbel, uint64_t seed, uint64_t counter, float* delta_travel_time,
                                   uint8_t* popped, uint8_t* withdrawn, int32_t* ins_scratch, float* reward,
                                   float* counts, tarl_stream stream) {
  int rc = check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(agent_features && A >= 1 && ins_scratch, "agents / scratch missing");
  TARL_REQUIRE(f->a_origin && f->a_dest && f->a_dep && f->a_status, "fused agent buffers missing");
  TARL_REQUIRE(B == 1 || a_bstride >= A * AG_COLS, "agent stride smaller than one population");
  TARL_REQUIRE(plan->E == 0 || (edge_attr && log_edge_attr), "edge constants missing");
  if (plan->N == 0) return TARL_OK;
  const FusedBufs fb = to_bufs(f);
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)ceil_div(B * plan->N, FB);
  const bool timed = tarl_prof_event(s) != nullptr;
  hipLaunchKernelGGL(k_fused_direction, dim3(grid), dim3(FB), 0, s, plan->in_ptr, plan->in_src, plan->in_eid, plan->E, B,
                     plan->N, fb, edge_attr, log_edge_attr, log_eps, time, gumbel, seed, counter, delta_travel_time);
  TARL_LAUNCH_CHECK();
  if (timed) (void)tarl_prof_event(s);
  hipLaunchKernelGGL(k_fused_rows, dim3(grid), dim3(FB), 0, s, plan->out_ptr, plan->out_dst, (int)Nmax, B, plan->N, fb,
                     agent_features, A, a_bstride, time, popped, withdrawn);
  TARL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_fused_insert, dim3((unsigned)B), dim3(ENVB), 0, s, (int)Nmax, plan->N, fb, agent_features, A,
                     a_bstride, use_cong, time, ins_scratch, reward, counts);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}
