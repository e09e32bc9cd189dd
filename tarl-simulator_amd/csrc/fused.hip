// fused.hip — the vectorised rollout frame in 4 launches, bit-identical (state, agents, actions, rewards) to the unfused
// kernels and therefore to the reference.
//
// Why: the reference's AoS row (F = 3*Nmax+7 floats, 208 B at Nmax = 15) scatters the ~10 scalars a message needs over
// 2-3 cache lines, and eleven separate launches per frame each re-stream the whole state. Here
//   * a packed per-(node, env) HOT RECORD  rec0 = {head_id, head_dep, n, tail_id} (+ sel [node][env], and the cold half
//     rec1 = {head_arr, pending-garbage n0}) and a per-node STATIC record st0 = {maxn, ff, road_index, cong} (shared by
//     all environments) mirror the row: the gather kernel reads 20 B per neighbour and never touches the FIFO storage;
//   * ENV-MINOR LAYOUT: every per-(node, env) array is stored [node][env]. A workgroup owns a tile of consecutive
//     environments (one per lane) and walks a chunk of nodes: all topology / table / static loads are wave-uniform
//     (scalar loads through the constant cache) and every record gather is a fully coalesced 16 B x 64 = 1 KiB load —
//     there is no dependent index -> address -> data chain left in the vector memory path;
//   * the Direction gather also emits postA = {n', tail'} / postB = chosen agent: the state every row will have after
//     the Direction update, from which the Response "accepted" test is evaluated (8-B gathers of postA) without a second
//     pass over the FIFOs;
//   * the live policy's sample is state-independent (see k_policy_tables): the choice phase is a table walk per
//     (node, env) with Philox blocks shared across consecutive nodes;
//   * ONE row pass applies Direction update + Response pop + withdraw and refreshes the hot record;
//   * the FIFO contents live in a slot-interleaved store  slots[node][env][s] = {id, arrival, departure}: the Direction
//     update's per-row write is ONE 12-byte store instead of three dwords in three DRAM sectors (+ counter);
//   * LAZY GARBAGE SLOT: a row that receives nobody still gets (0, t, t + tt) written into its first dead slot by the
//     reference. That value is never read by the simulation, is overwritten by the next frame's update (or
//     by an insertion) before anything can move it, and only shows in x. The row pass just records {flag, n0} in rec1 and
//     the export kernel materialises it (same fp32 expression, same slot). The one case where the pop's "last slot keeps
//     its value" rule would duplicate it (count == Nmax-1) is written eagerly;
//   * RING-BUFFER FIFOs: the reference pops by shifting all Nmax slots (and withdraws with a zero-filled shift). Here a
//     per-row head offset makes the pop one triple copy (the slot that falls off the front receives the old last slot,
//     which is exactly the reference's "last slot keeps its value") and a withdraw of c agents c zero-writes; the dead
//     slots end up with exactly the reference's contents and tarl_fused_export un-rotates;
//   * agent bookkeeping scans a 1-byte status + 4-byte departure SoA instead of 36-B AoS rows.
// The packed state is authoritative between tarl_fused_pack and tarl_fused_export; the exported x and agent_features are
// bit-identical to what the unfused kernels (and the reference) produce after every frame (tests/test_gpu_fused.py).
//
// Domain: the plan is built on the same node set as x (plan nodes == rows of x): pure road graphs and MATSim graphs
// with SRC/DEST pseudo-nodes alike (tests/test_gpu_fused.py::test_fused_equals_unfused_on_a_matsim_graph_with_pseudo_nodes). Counts that reach Nmax (outside the reference's defined domain,
// DESIGN.md Q25) are not supported by this path.
#include "fused_common.h"

// ---- pack: build the hot / static records, the slot store and the agent SoA from x / agent_features ------------------
__global__ __launch_bounds__(FB) void k_pack_nodes(const float* __restrict__ x, Layout L, int64_t B, int64_t N,
                                                   const float* __restrict__ cong, FusedBufs fb, float4* st0_out) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;  // gid = i * B + b  (env-minor)
  if (gid >= B * N) return;
  const int64_t i = gid / B;
  const int64_t b = gid - i * B;
  const float* xi = x + b * L.bstride + i * L.ldx;
  const int Nmax = L.Nmax;
  const float n = xi[L.col_n()];
  const int q = (int)n;
  const float tail = (q >= 1 && q <= Nmax) ? xi[q - 1] : 0.0f;
  fb.rec0[gid] = make_float4(xi[0], xi[2 * Nmax], n, tail);
  fb.rec1[gid] = make_float2(xi[Nmax], r1_code(-1.0f, 0));
  fb.postA[gid] = make_float2(n, tail);
  fb.postB[gid] = 0.0f;
  fb.sel[gid] = xi[L.col_sel()];
  if (i == 0) {
    for (int64_t sl_ = 0; sl_ < fb.acc_slots; ++sl_) {
      fb.acc_lp[sl_ * B + b] = 0;
      fb.acc_n[sl_ * B + b] = 0.0f;
    }
  }
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
  if (a == 0 && fb.cur_lo) fb.cur_lo[b] = 0;
}

// ---- reset: SimulatorEnv._reset on the packed state (zero FIFOs and counters, clear ON_WAY / DONE, re-arm cursors) -------
__global__ __launch_bounds__(FB) void k_fused_reset_nodes(int64_t B, int64_t N, FusedBufs fb) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (gid >= B * N) return;
  fb.rec0[gid] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  fb.rec1[gid] = make_float2(0.0f, r1_code(-1.0f, 0));
  fb.postA[gid] = make_float2(0.0f, 0.0f);
  fb.postB[gid] = 0.0f;
  if (gid < B) {
    for (int64_t sl_ = 0; sl_ < fb.acc_slots; ++sl_) {
      fb.acc_lp[sl_ * B + gid] = 0;
      fb.acc_n[sl_ * B + gid] = 0.0f;
    }
    if (fb.cur_lo) fb.cur_lo[gid] = 0;
  }
}

__global__ __launch_bounds__(FB) void k_fused_reset_agents(float* __restrict__ ag, int64_t B, int64_t A,
                                                           int64_t a_bstride, FusedBufs fb) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (gid >= B * A) return;
  // only agents that left the "waiting" state have flags to clear (the status SoA mirrors ON_WAY / DONE since the last
  // pack): a sequential 1-byte scan instead of two scattered stores into every 36-byte row
  if (fb.a_status[gid] == 0) return;
  const int64_t b = gid / A, a = gid - b * A;
  float* row = ag + b * a_bstride + a * AG_COLS;
  row[AG_ON_WAY] = 0.0f;
  row[AG_DONE] = 0.0f;
  fb.a_status[gid] = 0;
}

// ---- export: rebuild the reference's x layout (three FIFO column blocks + NUMBER_OF_AGENT + SELECTED_ROAD) ----------
__global__ __launch_bounds__(FB) void k_export_rows(float* __restrict__ x, Layout L, int64_t B, int64_t N, FusedBufs fb,
                                                    float t_last) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;
  const int Nmax = L.Nmax;
  if (gid >= B * N * Nmax) return;
  const int64_t row = gid / Nmax;  // row = i * B + b
  const int sidx = (int)(gid - row * Nmax);
  const int64_t i = row / B, b = row - i * B;
  float* xi = x + b * L.bstride + i * L.ldx;
  const float4 r0 = fb.rec0[row];
  const float2 r1 = fb.rec1[row];
  const float g = r1_g(r1.y);
  const float* sl = fb.slots + row * fb.lds + 3 * phys(r1_hoff(r1.y), sidx, Nmax);  // un-rotate the ring buffer
  if (g >= 0.0f && sidx == (int)r0.z) {  // pending garbage write of the last Direction update -> first dead slot
    const float4 st = fb.st0[i];
    const float t_cong = st.w / (st.x + 10.0f - g);
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
    xi[L.col_sel()] = fb.sel[row];
  }
}

// ---- policy tables ------------------------------------------------------------------------------------------------------
// The live policy's logits depend only on (emb, static ROAD_INDEX of the target road): they are identical for every
// environment and every frame between two optimiser steps. k_policy_tables evaluates, ONCE per parameter update and with
// exactly the arithmetic / reduction trees of k_edge_logits_fwd + k_softmax + k_sample + k_logprob_entropy_fwd, the
// per-edge tables (CSR order): thr[k] = fp32 inverse-CDF threshold, lg[k] = log(p + 1e-8), plus the entropy.
__device__ __forceinline__ float fb_block_sum(float v, float* s_red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  __syncthreads();
  if (lane == 0) s_red[wid] = v;
  __syncthreads();
  float tot = 0.0f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_red[w];
  return tot;
}

__device__ __forceinline__ float live_logit(const float* __restrict__ emb, int64_t M, const float4* __restrict__ st0,
                                            int32_t dst) {
  const long long idx = (long long)st0[dst].z;
  return (idx >= 0 && idx < M) ? emb[idx] : 0.0f;
}

__global__ __launch_bounds__(ENVB) void k_policy_tables(const int32_t* __restrict__ out_ptr,
                                                        const int32_t* __restrict__ out_dst,
                                                        const int32_t* __restrict__ node_of_group, int64_t N, int64_t G,
                                                        const float* __restrict__ emb, int64_t M, float temperature,
                                                        const float4* __restrict__ st0, double* __restrict__ base,
                                                        float* __restrict__ thr, long long* __restrict__ lgt,
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
      lgt[k] = (long long)((double)lg * LP_FIX);   // the frame kernels only ever add it up in 2^-32 fixed point
    }
  }
  const float ent_t = fb_block_sum(ent, s_red);
  if (tid == 0) entropy_out[0] = ent_t;
}

// CSR position of the edge node j picks in this frame: the first out-edge (plan order) whose threshold exceeds j's
// uniform draw; -1 when none does (u >= last threshold through rounding: the action is then infeasible).
__device__ __forceinline__ int32_t sample_node(const int32_t* __restrict__ out_ptr, const float* __restrict__ thr,
                                               int32_t j, float u) {
  const int32_t k1 = out_ptr[j + 1];
  for (int32_t k = out_ptr[j]; k < k1; ++k)
    if (u < thr[k]) return k;
  return -1;
}

__device__ __forceinline__ float node_uniform(const float* __restrict__ uniform, uint64_t seed, uint64_t counter,
                                              int64_t b, int64_t G, int32_t g) {
  return uniform ? uniform[b * G + g] : philox_uniform(seed, counter, (uint64_t)(b * G + g));
}

// ---- choice phase (env-minor: lane = environment, a workgroup walks a chunk of nodes) --------------------------------
// Consecutive nodes of one environment share Philox blocks (index = b*G + g), so a chunk costs ~nchunk/4 + 1 Philox
// evaluations per lane instead of one per node.
__device__ __forceinline__ void fused_choice_body(unsigned bx, unsigned by, const int32_t* __restrict__ out_ptr,
                                                       const int32_t* __restrict__ out_dst,
                                                       const int32_t* __restrict__ out_eid,
                                                       const int32_t* __restrict__ group_of_node, int64_t G, int64_t B,
                                                       int64_t N, FusedBufs fb, const float* __restrict__ thr,
                                                       const long long* __restrict__ lgt,
                                                       const float* __restrict__ uniform, uint64_t pseed,
                                                       uint64_t pcounter, int32_t* __restrict__ choice, int nchunk,
                                                       int want_lp, const float* __restrict__ sel_prev) {
  const int64_t b = (int64_t)bx * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int32_t i0 = by * nchunk;
  const int32_t i1 = (i0 + nchunk < N) ? i0 + nchunk : (int32_t)N;
  long long lp = 0;
  bool bad = false;
  PhiloxRun rng;
  for (int32_t i = i0; i < i1; ++i) {
    const int64_t row = (int64_t)i * B + b;
    int32_t ch = -1;
    const int32_t gi = group_of_node[i];
    if (gi >= 0) {
      const float u = uniform ? uniform[b * G + gi] : rng.uniform(pseed, pcounter, (uint64_t)(b * G + gi));
      // first out-edge (plan order) whose threshold exceeds u. Every table operand is wave-uniform (scalar loads), the
      // per-lane part is compare + select: no dependent vector gathers
      bool found = false;
      float selv = 0.0f;
      long long lpn = 0;
      const int32_t k1 = out_ptr[i + 1];
      for (int32_t k = out_ptr[i]; k < k1; ++k) {
        const bool hit = !found && (u < thr[k]);
        const long long lgk = lgt[k];
        selv = hit ? (float)out_dst[k] : selv;
        ch = hit ? out_eid[k] : ch;
        lpn = hit ? lgk : lpn;
        found = found || hit;
      }
      // a node that picks nothing keeps its previous SELECTED_ROAD (sel_prev = the other buffer when the rollout
      // double-buffers sel: the value is carried over)
      if (found) {
        fb.sel[row] = selv;
        lp += lpn;
      } else {
        bad = true;
        if (sel_prev) fb.sel[row] = sel_prev[row];
      }
    } else if (sel_prev) {
      fb.sel[row] = sel_prev[row];
    }
    if (choice) __builtin_nontemporal_store(ch, &choice[row]);  // write-once stream: keep it out of the caches
  }
  // infeasible action (some node picked nothing): poison the accumulator far beyond any legitimate sum
  if (want_lp)
    atomicAdd((unsigned long long*)&fb.acc_lp[(int64_t)(by % (unsigned)fb.acc_slots) * B + b], (unsigned long long)(bad ? -(1ll << 50) : lp));  // up to 2^13 chunks cannot wrap
}

__global__ __launch_bounds__(TILE) void k_fused_choice(const int32_t* __restrict__ out_ptr,
                                                       const int32_t* __restrict__ out_dst,
                                                       const int32_t* __restrict__ out_eid,
                                                       const int32_t* __restrict__ group_of_node, int64_t G, int64_t B,
                                                       int64_t N, FusedBufs fb, const float* __restrict__ thr,
                                                       const long long* __restrict__ lgt,
                                                       const float* __restrict__ uniform, uint64_t pseed,
                                                       uint64_t pcounter, int32_t* __restrict__ choice, int nchunk,
                                                       int want_lp) {
  fused_choice_body(blockIdx.x, blockIdx.y, out_ptr, out_dst, out_eid, group_of_node, G, B, N, fb, thr, lgt, uniform,
                    pseed, pcounter, choice, nchunk, want_lp, nullptr);
}

// ---- Direction gather on the hot records (env-minor: lane = environment) ---------------------------------------------------
// Two passes inside the workgroup. Pass 1 (every (node, environment) pair of the chunk): admissibility masks and the
// summed turn probability P — no random numbers — and the default post record (nobody chosen). A pair needs the Gumbel
// race only when P > 0, i.e. when some in-edge is admissible; that is a few percent of the pairs, but scattered over all
// lanes, so every wave would still pay for the Philox block and the two logs per edge. The pairs with P > 0 are
// therefore appended to an LDS list and pass 2 walks that list densely (one lane per pair), re-evaluating the pair with
// its noise — same Philox indices, same expressions: the result is bit-identical to evaluating everything.
#define DIR_LIST (TILE * 8)
__global__ __launch_bounds__(TILE) void k_fused_direction(const int32_t* __restrict__ in_ptr,
                                                          const int32_t* __restrict__ in_src,
                                                          const int32_t* __restrict__ in_eid, int64_t E, int64_t B,
                                                          int64_t N, FusedBufs fb, const float* __restrict__ edge_attr,
                                                          const float* __restrict__ log_edge_attr, float log_eps,
                                                          float t, const float* __restrict__ gumbel, uint64_t seed,
                                                          uint64_t counter, float* __restrict__ dtt, int nchunk) {
  __shared__ int32_t s_n;
  __shared__ uint16_t s_item[DIR_LIST];   // (node offset in the chunk) * TILE + lane
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = b < B;
  const int32_t i0 = blockIdx.y * nchunk;
  const int32_t i1 = (i0 + nchunk < N) ? i0 + nchunk : (int32_t)N;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  if (valid) {
    for (int32_t i = i0; i < i1; ++i) {  // i, and everything indexed by it alone, is wave-uniform
      const int64_t row = (int64_t)i * B + b;
      const float4 me = fb.rec0[row];
      const float4 sti = fb.st0[i];
      const float max_i = sti.x, n_i = me.z, road_i = sti.z;
      const float room_i = max_i - n_i;
      const bool has_room = n_i < max_i - TARL_CONGESTION_FILE;
      float P = 0.0f;
      const int32_t k1 = in_ptr[i + 1];
      for (int32_t k = in_ptr[i]; k < k1; ++k) {
        const int32_t j = in_src[k];
        const int32_t e = in_eid[k];
        const int64_t jrow = (int64_t)j * B + b;
        const float4 rj = fb.rec0[jrow];
        const float sel_j = fb.sel[jrow];  // road selected by upstream j in THIS frame's choice phase
        const float4 stj = fb.st0[j];
        const float dep = rj.y, n_j = rj.z, max_j = stj.x;
        const bool heads_here = sel_j == road_i;
        const bool m1 = (dep <= t) && has_room && heads_here && (n_j > 0.0f);
        const bool m2 = ((dep - t) < -10.0f) && ((max_j - TARL_CONGESTION_FILE) <= n_j) && ((max_j - n_j) <= room_i) &&
                        heads_here;
        const float prob = edge_attr[e] * ((m1 || m2) ? 1.0f : 0.0f);
        P = P + prob;
        if (dtt) {
          const float d = (dep - fb.rec1[jrow].x) - stj.y;
          dtt[b * E + e] = d > 0.0f ? d : (d != d ? d : 0.0f);
        }
      }
      fb.postA[row] = make_float2(n_i, me.w);   // nobody chosen; overwritten by pass 2 where P > 0
      fb.postB[row] = 0.0f;
      if (P > 0.0f) s_item[atomicAdd(&s_n, 1)] = (uint16_t)((i - i0) * TILE + threadIdx.x);
    }
  }
  __syncthreads();
  const int32_t cnt = s_n;
  for (int32_t idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
    const int32_t item = s_item[idx];
    const int32_t i = i0 + item / TILE;
    const int64_t bb = (int64_t)blockIdx.x * blockDim.x + (item % TILE);
    const int64_t row = (int64_t)i * B + bb;
    const float4 me = fb.rec0[row];
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
      const int64_t jrow = (int64_t)j * B + bb;
      const float4 rj = fb.rec0[jrow];
      const float sel_j = fb.sel[jrow];
      const float id = rj.x, dep = rj.y, n_j = rj.z, max_j = fb.st0[j].x;
      const bool heads_here = sel_j == road_i;
      const bool m1 = (dep <= t) && has_room && heads_here && (n_j > 0.0f);
      const bool m2 = ((dep - t) < -10.0f) && ((max_j - TARL_CONGESTION_FILE) <= n_j) && ((max_j - n_j) <= room_i) &&
                      heads_here;
      const bool m = m1 || m2;
      const float prob = edge_attr[e] * (m ? 1.0f : 0.0f);
      P = P + prob;
      float g;
      if (gumbel) {
        g = gumbel[bb * E + e];
      } else {
        const float u = rng.uniform(seed, counter, (uint64_t)(bb * E + k));
        g = gumbel_from_u01(u);
      }
      const float score = (m ? log_edge_attr[e] : log_eps) + g;
      if (score > best) {
        best = score;
        best_id = id;
      }
    }
    const float who = (P > 0.0f) ? best_id : 0.0f;
    fb.postA[row] = make_float2(who != 0.0f ? n_i + 1.0f : n_i, who != 0.0f ? who : me.w);
    fb.postB[row] = who;
  }
}

// ---- the row pass: Direction update + Response pop + withdraw on the slot store, then refresh the hot record -------------
__global__ __launch_bounds__(TILE) void k_fused_rows(const int32_t* __restrict__ out_ptr,
                                                     const int32_t* __restrict__ out_dst, int Nmax, int64_t B,
                                                     int64_t N, FusedBufs fb, float* __restrict__ ag, int64_t A,
                                                     int64_t a_bstride, float t, uint8_t* __restrict__ popped_out,
                                                     uint8_t* __restrict__ withdrawn_out, float* __restrict__ counts,
                                                     int nchunk) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int32_t i0 = blockIdx.y * nchunk;
  const int32_t i1 = (i0 + nchunk < N) ? i0 + nchunk : (int32_t)N;
  float nsum = 0.0f;
  for (int32_t i = i0; i < i1; ++i) {
    const int64_t row = (int64_t)i * B + b;
    float* sl = fb.slots + row * fb.lds;  // slot s = sl[3s .. 3s+2] = {id, arrival, departure}
    const float2 pa = fb.postA[row];   // {n', tail'}
    const float who = fb.postB[row];   // chosen
    const float4 r0 = fb.rec0[row];
    const float2 r1 = fb.rec1[row];
    const float4 st = fb.st0[i];
    const float n0 = r0.z;
    const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];

    // Response message + max-aggregate from the post records (state after the Direction update of every row)
    bool pop = false;
    {
      const long long head = (long long)((n0 == 0.0f) ? who : r0.x);   // head after the Direction update
      const bool up = (long long)pa.x > 0;
      for (int32_t k = k0; k < k1; ++k) {  // uniform trip count: the post gathers stay coalesced and unconditional
        const float2 pj = fb.postA[(int64_t)out_dst[k] * B + b];
        pop = pop || (up && (long long)pj.x > 0 && (long long)pj.y == head);
      }
    }
    if (popped_out) popped_out[b * N + i] = pop ? 1 : 0;

    // Direction update (every row, also when nothing was chosen): one 12-byte store — or, for a row that received
    // nobody, a note in the hot record (lazy garbage slot, see the file header). The FIFO is a ring buffer: logical
    // slot s lives at physical slot (hoff + s) mod Nmax.
    int hoff = r1_hoff(r1.y);
    const int q = (int)n0;
    const float t_cong = st.w / (st.x + 10.0f - n0);
    const float tt = (t_cong != t_cong) ? t_cong : fmaxf(st.y, t_cong);
    const float dep_new = t + tt;
    const bool lazy = (who == 0.0f) && (q >= 0) && (q < Nmax - 1);
    if (!lazy && q >= 0 && q < Nmax) {
      float* w = sl + 3 * phys(hoff, q, Nmax);
      w[0] = who;
      w[1] = t;
      w[2] = dep_new;
    }
    float n = pa.x;  // count after the Direction update

    // head / tail of the row after the Direction update (no memory reads needed)
    float head_id = (n0 == 0.0f) ? who : r0.x;
    float head_dep = (n0 == 0.0f) ? dep_new : r0.y;
    float head_arr = (n0 == 0.0f) ? t : r1.x;
    float tail_id = pa.y;

    // Response pop: logical shift by one where the LAST slot keeps its value. Ring form: the slot that falls off the
    // front becomes the new logical last slot, so it receives a copy of the old last slot; then the head advances.
    int shift = 0;
    if (pop) {
      const float* last = sl + 3 * phys(hoff, Nmax - 1, Nmax);
      const float l0 = last[0], l1 = last[1], l2 = last[2];
      float* front = sl + 3 * hoff;
      front[0] = l0;
      front[1] = l1;
      front[2] = l2;
      hoff = phys(hoff, 1, Nmax);
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
      for (int sx = 0; sx < Nmax && (float)sx < n; ++sx) {
        float idf, depf;
        if (sx == 0 && shift == 0) {   // the head is in registers unless the pop just exposed a new one
          idf = head_id;
          depf = head_dep;
        } else {
          const float* rd = sl + 3 * phys(hoff, sx, Nmax);
          idf = rd[0];
          depf = rd[2];
        }
        const long long id = (long long)idf;
        if (id < 0 || id >= A) break;
        if (!(depf <= t)) break;  // tested first: most heads are still travelling, and the lookup below is a gather
        const long long dest = (long long)fb.a_dest[b * A + id];
        bool conn = false;
        for (int32_t k = w0; k < w1; ++k) conn = conn || ((long long)out_dst[k] == dest);
        if (!conn) break;
        float* a = ag + b * a_bstride + id * AG_COLS;
        a[AG_DONE] = 1.0f;
        a[AG_ON_WAY] = 0.0f;
        a[AG_ARR] = t;
        fb.a_status[b * A + id] = 2;
        ++c;
      }
    }
    if (withdrawn_out) withdrawn_out[b * N + i] = c > 0 ? 1 : 0;
    // withdraw = logical shift by c with zero fill: the c slots that fall off the front become the zeroed tail
    for (int k = 0; k < c; ++k) {
      float* z = sl + 3 * phys(hoff, k, Nmax);
      z[0] = 0.0f;
      z[1] = 0.0f;
      z[2] = 0.0f;
    }
    if (c > 0) {
      hoff = phys(hoff, c, Nmax);   // c <= Nmax
      n = n - (float)c;
    }
    if (shift + c > 0) {
      if (lazy && n == 0.0f) {  // the row emptied: its head slot is the (unmaterialised) garbage slot
        head_id = 0.0f;
        head_arr = t;
        head_dep = dep_new;
      } else {
        const float* hd = sl + 3 * hoff;
        head_id = hd[0];
        head_arr = hd[1];
        head_dep = hd[2];
      }
      const int qn = (int)n;
      tail_id = (qn >= 1 && qn <= Nmax) ? sl[3 * phys(hoff, qn - 1, Nmax)] : 0.0f;
    }
    fb.rec0[row] = make_float4(head_id, head_dep, n, tail_id);
    fb.rec1[row] = make_float2(head_arr, r1_code(lazy ? n0 : -1.0f, hoff));
    // per-node count before insertion (the insert kernel adds this frame's arrivals); write-once stream
    if (counts) __builtin_nontemporal_store(n, &counts[row]);
    nsum += n;
  }
  atomicAdd(&fb.acc_n[(int64_t)(blockIdx.y % (unsigned)fb.acc_slots) * B + b], nsum);
}

// ---- insert + reward + log-prob reduction (one workgroup per environment) ----------------------------------------------
__device__ __forceinline__ bool fused_target(const FusedBufs& fb, int64_t b, int64_t B, int64_t N, int32_t origin,
                                             int32_t* road, int32_t* cap) {
  if (origin < 0 || origin >= N) return false;
  const long long r = (long long)fb.sel[(int64_t)origin * B + b];
  if (r < 0 || r >= N) return false;
  const long long room = (long long)(fb.st0[r].x - TARL_CONGESTION_FILE - fb.rec0[r * B + b].z);
  *road = (int32_t)r;
  *cap = (int32_t)(room > 0x7fffffff ? 0x7fffffff : room);
  return room > 0;
}

__device__ __forceinline__ void fused_insert_body(int64_t b, int Nmax, int64_t B, int64_t N, FusedBufs fb,
                                                       float* __restrict__ ag, int64_t A, int64_t a_bstride,
                                                       int use_cong, float t, int32_t* __restrict__ scratch,
                                                       const float* __restrict__ entropy_in,
                                                       float* __restrict__ reward, float* __restrict__ counts,
                                                       float* __restrict__ log_prob, float* __restrict__ entropy) {
  __shared__ int32_t s_wave[INSB / 64];
  __shared__ int32_t s_cnt;
  __shared__ int32_t s_adm;
  __shared__ int32_t s_lo;
  __shared__ int32_t s_un_agent[INS_CAP], s_un_road[INS_CAP];
  float* agb = ag + b * a_bstride;
  int32_t* cand_agent = scratch + b * 2 * A;
  int32_t* cand_road = cand_agent + A;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  // phase 1: candidates (ready agent whose target road has room). Candidates are rare (a handful per frame), so they
  // are appended unordered to an LDS list with an atomic counter and then ranked by agent id (deterministic: the
  // reference admits agents in stable agent-id order). A backlog larger than the LDS list falls back to the ordered
  // ballot compaction into the global scratch.
  if (tid == 0) {
    s_cnt = 0;
    s_adm = 0;
  }
  __syncthreads();
  if (fb.a_order) {
    // Windowed scan: agents sorted by departure time; everything before cur_lo is known not to be waiting any more and
    // everything after the first not-yet-due entry is not due either, so a frame normally looks at one chunk.
    const int32_t* ord = fb.a_order + b * A;
    const float* dsort = fb.a_dep_sorted + b * A;
    const int32_t lo = fb.cur_lo[b];
    if (tid == 0) s_lo = 0x7fffffff;
    __syncthreads();
    for (int64_t k0 = lo; k0 < A; k0 += INSB) {
      const int64_t k = k0 + tid;
      bool notdue = false;
      if (k < A) {
        const bool due = dsort[k] <= t;          // sequential read; per-agent arrays only for entries that are due
        notdue = !due;
        if (!due) {
          atomicMin(&s_lo, (int32_t)k);          // the cursor may not pass this entry
        } else {
          const int32_t a = ord[k];
          if (fb.a_status[b * A + a] == 0) {
            atomicMin(&s_lo, (int32_t)k);
            int32_t road = 0, cap = 0;
            if (fused_target(fb, b, B, N, fb.a_origin[b * A + a], &road, &cap)) {
              const int32_t pos = atomicAdd(&s_cnt, 1);
              if (pos < INS_CAP) {
                s_un_agent[pos] = a;
                s_un_road[pos] = road;
              }
            }
          }
        }
      }
      if (__syncthreads_or(notdue ? 1 : 0)) break;   // sorted by departure: nothing beyond this chunk is due
    }
    __syncthreads();
    if (tid == 0) fb.cur_lo[b] = s_lo == 0x7fffffff ? (int32_t)A : s_lo;
  } else {
    for (int64_t a0 = tid; a0 < A; a0 += 4 * INSB) {  // 4 independent (status, departure) loads in flight per thread
      uint8_t stt[4];
      float dp[4];
  #pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t a = a0 + (int64_t)j * INSB;
        stt[j] = a < A ? fb.a_status[b * A + a] : (uint8_t)1;
        dp[j] = a < A ? fb.a_dep[b * A + a] : 0.0f;
      }
  #pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (stt[j] == 0 && dp[j] <= t) {
          const int64_t a = a0 + (int64_t)j * INSB;
          int32_t road = 0, cap = 0;
          if (fused_target(fb, b, B, N, fb.a_origin[b * A + a], &road, &cap)) {
            const int32_t pos = atomicAdd(&s_cnt, 1);
            if (pos < INS_CAP) {
              s_un_agent[pos] = (int32_t)a;
              s_un_road[pos] = road;
            }
          }
        }
      }
    }
  }
  __syncthreads();
  int32_t Lc = s_cnt;
  if (Lc <= INS_CAP) {
    for (int32_t idx = tid; idx < Lc; idx += INSB) {
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
    for (int64_t a0 = 0; a0 < A; a0 += INSB) {
      const int64_t a = a0 + tid;
      bool cnd = false;
      int32_t road = 0, cap = 0;
      if (a < A && fb.a_status[b * A + a] == 0 && fb.a_dep[b * A + a] <= t)
        cnd = fused_target(fb, b, B, N, fb.a_origin[b * A + a], &road, &cap);
      const unsigned long long bal = __ballot(cnd);
      const int lane_off = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) s_wave[wid] = __popcll(bal);
      __syncthreads();
      int32_t wbase = 0, tot = 0;
      for (int w = 0; w < INSB / 64; ++w) {
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

  // phase 2: rank within road (stable), admit the first min(count, capacity), write slots / hot records
  for (int32_t idx = tid; idx < Lc; idx += INSB) {
    const int32_t r = cand_road[idx];
    const int32_t a = cand_agent[idx];
    int32_t rank = 0, total = 0;
    for (int32_t k = 0; k < Lc; ++k) {
      const bool same = cand_road[k] == r;
      total += same ? 1 : 0;
      rank += (same && k < idx) ? 1 : 0;
    }
    const int64_t rrow = (int64_t)r * B + b;
    const float4 str = fb.st0[r];
    const float n0 = fb.rec0[rrow].z;
    const long long cap = (long long)(str.x - TARL_CONGESTION_FILE - n0);
    int32_t commit = 0;
    bool pend_clear = false;
    if (rank < cap) {
      const long long m = total < cap ? total : cap;  // arrivals admitted on this road
      const long long slot = (long long)n0 + rank;
      const float t_cong = use_cong ? str.w / (str.x + 10.0f - (float)(long long)n0) : 0.0f;
      const float tt = (t_cong != t_cong) ? t_cong : fmaxf(str.y, t_cong);
      const float code = fb.rec1[rrow].y;   // nobody writes rec1.y before the barrier below
      if (slot >= 0 && slot < Nmax) {
        float* sr = fb.slots + rrow * fb.lds + 3 * phys(r1_hoff(code), (int)slot, Nmax);
        sr[0] = (float)a;
        sr[1] = t;
        sr[2] = t + tt;
      }
      if (rank == 0) pend_clear = true;
      agb[(int64_t)a * AG_COLS + AG_ON_WAY] = 1.0f;
      fb.a_status[b * A + a] = 1;
      // hot record: only fields nobody reads in this phase (n is committed after the barrier)
      if (rank == 0 && n0 == 0.0f) {
        fb.rec0[rrow].x = (float)a;
        fb.rec0[rrow].y = t + tt;
        fb.rec1[rrow].x = t;
      }
      if (rank == m - 1) fb.rec0[rrow].w = (float)a;  // new tail
      if (rank == 0) commit = (int32_t)m;
    }
    cand_agent[idx] = commit;
    (void)pend_clear;
  }
  __threadfence_block();
  __syncthreads();
  // phase 3: commit the counters; the arrivals overwrote a pending garbage slot: clear the flag, keep the head offset
  for (int32_t idx = tid; idx < Lc; idx += INSB) {
    const int32_t cmt = cand_agent[idx];
    if (cmt > 0) {
      const int64_t rrow = (int64_t)cand_road[idx] * B + b;
      fb.rec1[rrow].y = r1_code(-1.0f, r1_hoff(fb.rec1[rrow].y));
      const float nn = fb.rec0[rrow].z + (float)cmt;
      fb.rec0[rrow].z = nn;
      if (counts) counts[rrow] = nn;
      atomicAdd(&s_adm, cmt);
    }
  }
  __syncthreads();
  // phase 4: the frame's accumulator banks (filled by the choice kernel and the row pass) -> reward, log-prob; re-arm
  if (wid == 0) {
    long long lpf = 0;
    float nf = 0.0f;
    for (int64_t sl_ = lane; sl_ < fb.acc_slots; sl_ += 64) {
      lpf += fb.acc_lp[sl_ * B + b];
      nf += fb.acc_n[sl_ * B + b];
      fb.acc_lp[sl_ * B + b] = 0;
      fb.acc_n[sl_ * B + b] = 0.0f;
    }
    for (int off = 32; off > 0; off >>= 1) {
      lpf += __shfl_down(lpf, off);
      nf += __shfl_down(nf, off);      // sums of small integers: exact in fp32 in any order
    }
    if (lane == 0) {
      if (reward) reward[b] = -(nf + (float)s_adm);
      if (log_prob) log_prob[b] = (lpf < -(1ll << 49)) ? -INFINITY : (float)((double)lpf / LP_FIX);
      if (entropy) entropy[b] = entropy_in[0];
    }
  }
}

__global__ __launch_bounds__(INSB) void k_fused_insert(int Nmax, int64_t B, int64_t N, FusedBufs fb,
                                                       float* __restrict__ ag, int64_t A, int64_t a_bstride,
                                                       int use_cong, float t, int32_t* __restrict__ scratch,
                                                       const float* __restrict__ entropy_in,
                                                       float* __restrict__ reward, float* __restrict__ counts,
                                                       float* __restrict__ log_prob, float* __restrict__ entropy) {
  fused_insert_body(blockIdx.x, Nmax, B, N, fb, ag, A, a_bstride, use_cong, t, scratch, entropy_in, reward, counts,
                    log_prob, entropy);
}

// One launch, two roles (rollout steady state): the first B workgroups run frame t's insert (one wave each; the other
// three waves of such a workgroup retire at once, so its barriers only count the live wave), the remaining
// `choice_blocks` workgroups draw frame t+1's action into the OTHER half of the double-buffered SELECTED_ROAD / log-prob
// accumulators.
// The insert kernel is a latency chain that leaves the chip idle and the live policy's sample does not depend on the
// state, so the choice work rides in its shadow — without the cross-stream events that made the two-stream variant
// slower — and still completes right before the Direction kernel that consumes it (Infinity-Cache adjacency).
struct ChoiceArgs {
  const int32_t* out_ptr;
  const int32_t* out_dst;
  const int32_t* out_eid;
  const int32_t* group_of_node;
  int64_t G;
  const float* thr;
  const long long* lgt;
  uint64_t pseed, pcounter;
  int32_t* choice;
  int nchunk, want_lp;
  float* sel_next;
  long long* acc_next;
  const float* sel_cur;
  unsigned gx;             // environment tiles (x extent of the choice grid)
  unsigned choice_blocks;  // gx * node chunks
};
__global__ __launch_bounds__(TILE) void k_fused_insert_choice(ChoiceArgs C, int Nmax, int64_t B, int64_t N, FusedBufs fb,
                                                              float* __restrict__ ag, int64_t A, int64_t a_bstride,
                                                              int use_cong, float t, int32_t* __restrict__ scratch,
                                                              const float* __restrict__ entropy_in,
                                                              float* __restrict__ reward, float* __restrict__ counts,
                                                              float* __restrict__ log_prob,
                                                              float* __restrict__ entropy) {
  // insert workgroups first: their dependent-load chains start at once and the choice workgroups fill the chip around them
  if (blockIdx.x < (unsigned)B) {
    if (threadIdx.x >= INSB) return;   // whole waves leave before any barrier
    fused_insert_body((int64_t)blockIdx.x, Nmax, B, N, fb, ag, A, a_bstride, use_cong, t, scratch, entropy_in, reward,
                      counts, log_prob, entropy);
  } else {
    const unsigned cb = blockIdx.x - (unsigned)B;
    FusedBufs fc = fb;
    fc.sel = C.sel_next;
    fc.acc_lp = C.acc_next;
    fused_choice_body(cb % C.gx, cb / C.gx, C.out_ptr, C.out_dst, C.out_eid, C.group_of_node, C.G, B, N, fc, C.thr,
                      C.lgt, nullptr, C.pseed, C.pcounter, C.choice, C.nchunk, C.want_lp, C.sel_cur);
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
FusedBufs tarl_to_bufs(const tarl_fused* f) {
  return FusedBufs{(float4*)f->rec0,         (float2*)f->rec1, (float2*)f->post_a, f->post_b,
                   (const float4*)f->st0,    f->slots,         f->ld_slots,        f->sel,
                   (long long*)f->acc_lp,    f->acc_n,         f->a_origin,        f->a_dest,
                   f->a_dep,                 f->a_status,      f->a_order,         f->cur_lo,
                   f->a_dep_sorted,          f->acc_slots};
}

// nodes walked by one workgroup of the env-minor kernels (tunable: TARL_NCHUNK)
static int nchunk() {
  static int v = 0;
  if (v == 0) {
    const char* e = getenv("TARL_NCHUNK");
    v = e ? atoi(e) : 2;
    if (v < 1) v = 1;
  }
  return v;
}
static int64_t num_chunks(const tarl_plan* plan) { return ceil_div(plan->N, nchunk()); }
// nodes per workgroup pass of the Direction kernel (measured: 1 -> 53.5 us, 2 -> 48.2, 3 -> 49.9, 4 -> 64, 8 -> 85 per
// launch at B = 2048): two give the dense second pass ~25 pairs per workgroup without starving the chip of workgroups;
// at most 8 (capacity of the LDS list and of the 16-bit item code)
static int nchunk_dir() {
  static int v = 0;
  if (v == 0) {
    const char* e = getenv("TARL_NCHUNK_DIR");
    v = e ? atoi(e) : 2;
    if (v < 1) v = 1;
    if (v > 8) v = 8;
  }
  return v;
}
// the choice kernel is light and ends in one accumulator atomic per lane: it walks longer chunks (TARL_NCHUNK_CHOICE)
static int nchunk_choice() {
  static int v = 0;
  if (v == 0) {
    const char* e = getenv("TARL_NCHUNK_CHOICE");
    v = e ? atoi(e) : 8;
    if (v < 1) v = 1;
  }
  return v;
}

int tarl_check_fused_core(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax) {
  TARL_REQUIRE(plan && f, "null argument");
  TARL_REQUIRE(f->rec0 && f->rec1 && f->post_a && f->post_b && f->st0 && f->slots && f->sel && f->acc_lp && f->acc_n,
               "fused node buffers missing");
  TARL_REQUIRE(B >= 1 && B < ((int64_t)1 << 31) && Nmax >= 2, "bad sizes");
  TARL_REQUIRE(f->acc_slots >= 1 && f->acc_slots <= 4096, "acc_slots out of range");
  TARL_REQUIRE(f->ld_slots >= 3 * (int64_t)Nmax, "slot row stride smaller than 3*Nmax");
  TARL_REQUIRE(num_chunks(plan) < 65536 && ceil_div(plan->N, nchunk_choice()) < 65536 &&
                   ceil_div(plan->N, nchunk_dir()) < 65536,
               "too many node chunks for one launch");
  TARL_REQUIRE(((uintptr_t)f->rec0 | (uintptr_t)f->rec1 | (uintptr_t)f->post_a | (uintptr_t)f->post_b |
                (uintptr_t)f->st0) % 16 == 0,
               "fused records must be 16-byte aligned");
  return TARL_OK;
}

static int check_fused(const tarl_plan* plan, const tarl_fused* f, const float* x, int64_t B, int64_t bstride,
                       int64_t ldx, int32_t Nmax) {
  int rc = tarl_check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(x != nullptr, "x is null");
  TARL_REQUIRE(ldx >= 3 * (int64_t)Nmax + 7, "row stride smaller than F");
  TARL_REQUIRE(B == 1 || bstride >= plan->N * ldx, "environment stride smaller than one environment");
  return TARL_OK;
}

static unsigned tile_threads(int64_t B) { return B >= TILE ? TILE : (unsigned)(ceil_div(B, 64) * 64); }

extern "C" int tarl_fused_pack(const tarl_plan* plan, const tarl_fused* f, const float* x, int64_t B, int64_t x_bstride,
                               int64_t ldx, int32_t Nmax, const float* cong, const float* agent_features, int64_t A,
                               int64_t a_bstride, tarl_stream stream) {
  int rc = check_fused(plan, f, x, B, x_bstride, ldx, Nmax);
  if (rc) return rc;
  const Layout L{Nmax, ldx, x_bstride};
  const FusedBufs fb = tarl_to_bufs(f);
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

extern "C" int tarl_fused_reset(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax,
                                float* agent_features, int64_t A, int64_t a_bstride, tarl_stream stream) {
  int rc = tarl_check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(agent_features && A >= 1 && f->a_status, "agents missing");
  const FusedBufs fb = tarl_to_bufs(f);
  hipStream_t s = (hipStream_t)stream;
  TARL_CHECK_HIP(hipMemsetAsync(f->slots, 0, (size_t)(plan->N * B * f->ld_slots) * sizeof(float), s));
  if (plan->N > 0) {
    hipLaunchKernelGGL(k_fused_reset_nodes, dim3((unsigned)ceil_div(B * plan->N, FB)), dim3(FB), 0, s, B, plan->N, fb);
    TARL_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_fused_reset_agents, dim3((unsigned)ceil_div(B * A, FB)), dim3(FB), 0, s, agent_features, B, A,
                     a_bstride, fb);
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
                     x, L, B, plan->N, tarl_to_bufs(f), last_step_time);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_fused_policy_prepare(const tarl_plan* plan, const tarl_fused* f, const float* emb,
                                         int64_t num_embeddings, float temperature, double* group_base,
                                         float* thresholds, int64_t* log_probs, float* entropy1, tarl_stream stream) {
  TARL_REQUIRE(plan && f && f->st0 && emb && group_base && thresholds && log_probs && entropy1, "null argument");
  TARL_REQUIRE(num_embeddings >= 1, "bad sizes");
  if (plan->N == 0) return TARL_OK;
  hipLaunchKernelGGL(k_policy_tables, dim3(1), dim3(ENVB), 0, (hipStream_t)stream, plan->out_ptr, plan->out_dst,
                     plan->node_of_group, plan->N, plan->G, emb, num_embeddings, temperature, (const float4*)f->st0,
                     group_base, thresholds, (long long*)log_probs, entropy1);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_fused_frame(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax,
                                const float* thresholds, const int64_t* log_probs, const float* entropy1,
                                const float* uniform, uint64_t policy_seed, uint64_t policy_counter,
                                float* agent_features, int64_t A, int64_t a_bstride, const float* edge_attr,
                                const float* log_edge_attr, float log_eps, int use_cong, float time, const float* gumbel,
                                uint64_t seed, uint64_t counter, float* delta_travel_time, uint8_t* popped,
                                uint8_t* withdrawn, int32_t* ins_scratch, int32_t* choice, float* log_prob,
                                float* entropy, float* reward, float* counts, tarl_stream stream) {
  int rc = tarl_check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(thresholds && log_probs && entropy1, "policy tables missing (call tarl_fused_policy_prepare)");
  TARL_REQUIRE(agent_features && A >= 1 && ins_scratch, "agents / scratch missing");
  TARL_REQUIRE(f->a_origin && f->a_dest && f->a_dep && f->a_status, "fused agent buffers missing");
  TARL_REQUIRE(f->a_order == nullptr || (f->cur_lo != nullptr && f->a_dep_sorted != nullptr),
               "a_order needs cur_lo and a_dep_sorted");
  TARL_REQUIRE(B == 1 || a_bstride >= A * AG_COLS, "agent stride smaller than one population");
  TARL_REQUIRE(plan->E == 0 || (edge_attr && log_edge_attr), "edge constants missing");
  if (plan->N == 0) return TARL_OK;
  const FusedBufs fb = tarl_to_bufs(f);
  hipStream_t s = (hipStream_t)stream;
  const unsigned threads = tile_threads(B);
  const dim3 grid((unsigned)ceil_div(B, threads), (unsigned)num_chunks(plan));
  const dim3 grid_c((unsigned)ceil_div(B, threads), (unsigned)ceil_div(plan->N, nchunk_choice()));
  hipLaunchKernelGGL(k_fused_choice, grid_c, dim3(threads), 0, s, plan->out_ptr, plan->out_dst, plan->out_eid,
                     plan->group_of_node, plan->G, B, plan->N, fb, thresholds, (const long long*)log_probs, uniform, policy_seed,
                     policy_counter, choice, nchunk_choice(), log_prob != nullptr ? 1 : 0);
  TARL_LAUNCH_CHECK();
  const bool timed = tarl_prof_mark(s, 0) != nullptr;
  const dim3 grid_d((unsigned)ceil_div(B, threads), (unsigned)ceil_div(plan->N, nchunk_dir()));
  hipLaunchKernelGGL(k_fused_direction, grid_d, dim3(threads), 0, s, plan->in_ptr, plan->in_src, plan->in_eid, plan->E,
                     B, plan->N, fb, edge_attr, log_edge_attr, log_eps, time, gumbel, seed, counter, delta_travel_time,
                     nchunk_dir());
  TARL_LAUNCH_CHECK();
  if (timed) (void)tarl_prof_mark(s, 1);
  hipLaunchKernelGGL(k_fused_rows, grid, dim3(threads), 0, s, plan->out_ptr, plan->out_dst, (int)Nmax, B, plan->N, fb,
                     agent_features, A, a_bstride, time, popped, withdrawn, counts, nchunk());
  TARL_LAUNCH_CHECK();
  if (timed) (void)tarl_prof_mark(s, 2);
  hipLaunchKernelGGL(k_fused_insert, dim3((unsigned)B), dim3(INSB), 0, s, (int)Nmax, B, plan->N, fb,
                     agent_features, A, a_bstride, use_cong, time, ins_scratch, entropy1, reward, counts, log_prob,
                     entropy);
  TARL_LAUNCH_CHECK();
  if (timed) (void)tarl_prof_mark(s, 3);
  return TARL_OK;
}

// T consecutive frames with device noise: the collector loop in one foreign call. Each frame is choice -> direction ->
// rows -> insert on the caller's stream; with the scratch pair, frame t+1's choice shares the launch of frame t's insert
// (k_fused_insert_choice). (Two alternatives were measured and rejected, DESIGN.md §4.2: folding
// frame t+1's choice into the row pass, and running it on a side stream into double-buffered SELECTED_ROAD /
// accumulators. Both lose the producer -> consumer adjacency that lets the Direction kernel read the 20 MB the choice
// kernel just wrote from the Infinity Cache.)
extern "C" int tarl_fused_rollout(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax, int64_t T,
                                  const float* times_host, const float* thresholds, const int64_t* log_probs,
                                  const float* entropy1, uint64_t policy_seed, uint64_t policy_counter0,
                                  float* agent_features, int64_t A, int64_t a_bstride, const float* edge_attr,
                                  const float* log_edge_attr, float log_eps, int use_cong, uint64_t seed,
                                  uint64_t counter0, int32_t* ins_scratch, float* sel_scratch, int64_t* acc_scratch,
                                  int32_t* choice, float* log_prob, float* entropy, float* reward, float* counts,
                                  tarl_stream stream) {
  int rc = tarl_check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(T >= 1 && times_host, "bad frame count / times");
  TARL_REQUIRE(thresholds && log_probs && entropy1, "policy tables missing (call tarl_fused_policy_prepare)");
  TARL_REQUIRE(agent_features && A >= 1 && ins_scratch, "agents / scratch missing");
  TARL_REQUIRE(f->a_origin && f->a_dest && f->a_dep && f->a_status, "fused agent buffers missing");
  TARL_REQUIRE(f->a_order == nullptr || (f->cur_lo != nullptr && f->a_dep_sorted != nullptr),
               "a_order needs cur_lo and a_dep_sorted");
  TARL_REQUIRE(B == 1 || a_bstride >= A * AG_COLS, "agent stride smaller than one population");
  TARL_REQUIRE(plan->E == 0 || (edge_attr && log_edge_attr), "edge constants missing");
  if (plan->N == 0) return TARL_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t N = plan->N, NB = N * B;
  const unsigned threads = tile_threads(B);
  const dim3 grid((unsigned)ceil_div(B, threads), (unsigned)num_chunks(plan));
  const dim3 grid_c((unsigned)ceil_div(B, threads), (unsigned)ceil_div(N, nchunk_choice()));
  const dim3 grid_d((unsigned)ceil_div(B, threads), (unsigned)ceil_div(N, nchunk_dir()));
  const int want_lp = log_prob != nullptr ? 1 : 0;
  FusedBufs fb = tarl_to_bufs(f);
  // steady state: frame t's insert and frame t+1's choice share ONE launch (k_fused_insert_choice); SELECTED_ROAD and
  // the log-prob accumulator banks are double-buffered between f->sel / f->acc_lp and the scratch pair
  const char* knob = getenv("TARL_ROLLOUT_MERGE");
  const bool merge = sel_scratch && acc_scratch && T > 1 && !(knob && atoi(knob) == 0);
  float* sel_buf[2] = {f->sel, merge ? sel_scratch : f->sel};
  long long* acc_buf[2] = {(long long*)f->acc_lp, merge ? (long long*)acc_scratch : (long long*)f->acc_lp};
  if (merge) TARL_CHECK_HIP(hipMemsetAsync(acc_scratch, 0, (size_t)(f->acc_slots * B) * sizeof(int64_t), s));
  hipLaunchKernelGGL(k_fused_choice, grid_c, dim3(threads), 0, s, plan->out_ptr, plan->out_dst, plan->out_eid,
                     plan->group_of_node, plan->G, B, N, fb, thresholds, (const long long*)log_probs,
                     (const float*)nullptr, policy_seed, policy_counter0, choice, nchunk_choice(), want_lp);
  TARL_LAUNCH_CHECK();
  for (int64_t t = 0; t < T; ++t) {
    const int cur = merge ? (int)(t & 1) : 0;
    const float time = times_host[t];
    fb.sel = sel_buf[cur];
    fb.acc_lp = acc_buf[cur];
    const bool timed = tarl_prof_mark(s, 0) != nullptr;
    hipLaunchKernelGGL(k_fused_direction, grid_d, dim3(threads), 0, s, plan->in_ptr, plan->in_src, plan->in_eid,
                       plan->E, B, N, fb, edge_attr, log_edge_attr, log_eps, time, (const float*)nullptr, seed,
                       counter0 + (uint64_t)t, (float*)nullptr, nchunk_dir());
    TARL_LAUNCH_CHECK();
    if (timed) (void)tarl_prof_mark(s, 1);
    float* counts_t = counts ? counts + t * NB : nullptr;
    hipLaunchKernelGGL(k_fused_rows, grid, dim3(threads), 0, s, plan->out_ptr, plan->out_dst, (int)Nmax, B, N, fb,
                       agent_features, A, a_bstride, time, (uint8_t*)nullptr, (uint8_t*)nullptr, counts_t, nchunk());
    TARL_LAUNCH_CHECK();
    if (timed) (void)tarl_prof_mark(s, 2);
    float* reward_t = reward ? reward + t * B : nullptr;
    float* lp_t = log_prob ? log_prob + t * B : nullptr;
    float* ent_t = entropy ? entropy + t * B : nullptr;
    if (merge && t + 1 < T) {
      const ChoiceArgs C{plan->out_ptr, plan->out_dst, plan->out_eid, plan->group_of_node, plan->G, thresholds,
                         (const long long*)log_probs, policy_seed, policy_counter0 + (uint64_t)(t + 1),
                         choice ? choice + (t + 1) * NB : nullptr, nchunk_choice(), want_lp, sel_buf[cur ^ 1],
                         acc_buf[cur ^ 1], sel_buf[cur], grid_c.x, grid_c.x * grid_c.y};
      hipLaunchKernelGGL(k_fused_insert_choice, dim3(C.choice_blocks + (unsigned)B), dim3(threads), 0, s, C, (int)Nmax,
                         B, N, fb, agent_features, A, a_bstride, use_cong, time, ins_scratch, entropy1, reward_t,
                         counts_t, lp_t, ent_t);
      TARL_LAUNCH_CHECK();
    } else {
      hipLaunchKernelGGL(k_fused_insert, dim3((unsigned)B), dim3(INSB), 0, s, (int)Nmax, B, N, fb, agent_features, A,
                         a_bstride, use_cong, time, ins_scratch, entropy1, reward_t, counts_t, lp_t, ent_t);
      TARL_LAUNCH_CHECK();
      if (t + 1 < T) {   // unmerged: the next frame's choice in place
        hipLaunchKernelGGL(k_fused_choice, grid_c, dim3(threads), 0, s, plan->out_ptr, plan->out_dst, plan->out_eid,
                           plan->group_of_node, plan->G, B, N, fb, thresholds, (const long long*)log_probs,
                           (const float*)nullptr, policy_seed, policy_counter0 + (uint64_t)(t + 1),
                           choice ? choice + (t + 1) * NB : nullptr, nchunk_choice(), want_lp);
        TARL_LAUNCH_CHECK();
      }
    }
    if (timed) (void)tarl_prof_mark(s, 3);
  }
  if (merge && ((T - 1) & 1) == 1)   // the last frame's SELECTED_ROAD lives in the scratch buffer: bring it home
    TARL_CHECK_HIP(hipMemcpyAsync(f->sel, sel_scratch, (size_t)NB * sizeof(float), hipMemcpyDeviceToDevice, s));
  return TARL_OK;
}
