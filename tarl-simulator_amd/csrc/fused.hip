// fused.hip — the vectorised rollout frame in 3-4 launches, bit-identical (state, agents, actions, rewards) to the unfused
// kernels and therefore to the reference.
//
// Why: the reference's AoS row (F = 3*Nmax+7 floats, 208 B at Nmax = 15) scatters the ~10 scalars a message needs over
// 2-3 cache lines, and eleven separate launches per frame each re-stream the whole state. Here
//   * ENV-MINOR LAYOUT: every per-(node, env) array is stored [node][env]. A workgroup owns a tile of consecutive
//     environments (one per lane) and walks a chunk of nodes: all topology / table / static loads are wave-uniform
//     (scalar loads through the constant cache) and every record gather is fully coalesced — there is no dependent
//     index -> address -> data chain left in the vector memory path;
//   * DENSE WORDS (fused_common.h): what every row reads and rewrites every frame is 12 bytes — hdp = {head_id << 8 | n,
//     head_dep} and tl = tail_id << 8 | flags — plus a per-node STATIC record st0 = {maxn, ff, road_index, cong} shared by
//     all environments. The Direction gather reads 8 B (+ 1 B of SELECTED_ROAD) per neighbour and never touches the FIFOs;
//   * the Direction gather emits ONE post word per row (tail' << 8 | non-empty' | arrived): the state the row will have
//     after the Direction update. The enqueued agent, the new count and the Response "accepted" test (tail of the
//     downstream row == my head, both non-empty) all follow from it — no second pass over the FIFOs, no extra arrays;
//   * SELECTED_ROAD and the action are the same byte: the rank of the chosen out-edge in the node's CSR list (sel8). In a
//     rollout the action buffer's slice of frame t IS the SELECTED_ROAD column the Direction gather of frame t reads;
//   * the live policy's sample is state-independent (see k_policy_tables): the choice phase is a table walk per
//     (node, env) with Philox blocks shared across consecutive nodes;
//   * ONE row pass applies Direction update + Response pop + withdraw. A row where nothing moves (nobody enqueued, no pop,
//     head not due) touches nothing but its dense words; the EVENT-ONLY byte gc8 (pending-garbage count) and the FIFO
//     store are written only by the rows that move something (the head's arrival time is never stored beside the dense
//     words: it is the arrival field of the head's slot record, fused_common.h: head_arrival);
//   * the FIFO contents live in a slot-interleaved store  slots[node][env][s] = {id, arrival, departure}: the Direction
//     update's per-row write is ONE 12-byte store instead of three dwords in three DRAM sectors (+ counter). (32-byte
//     records — a store then fills its sector, no read-modify-write at the memory side — were measured in round 4 and
//     rejected: fused_common.h, TARL_SLW);
//   * LAZY GARBAGE SLOT: a row that receives nobody still gets (0, t, t + tt) written into its first dead slot by the
//     reference. That value is never read by the simulation, is overwritten by the next frame's update (or
//     by an insertion) before anything can move it, and only shows in x. It is never stored: for a row that was idle in
//     the last frame the count at the write is its count, otherwise gc8 holds it (TLF_AUTH), and the export kernel
//     materialises the triple (same fp32 expression, same slot). The one case where the pop's "last slot keeps its value"
//     rule would duplicate it (count == Nmax-1) is written eagerly;
//   * RING-BUFFER FIFOs: the reference pops by shifting all Nmax slots (and withdraws with a zero-filled shift). Here a
//     per-row head offset makes the pop one triple copy (the slot that falls off the front receives the old last slot,
//     which is exactly the reference's "last slot keeps its value") and a withdraw of c agents c zero-writes; the dead
//     slots end up with exactly the reference's contents and tarl_fused_export un-rotates;
//   * agent bookkeeping scans a 1-byte status + 4-byte departure SoA instead of 36-B AoS rows.
// The packed state is authoritative between tarl_fused_pack and tarl_fused_export; the exported x and agent_features are
// bit-identical to what the unfused kernels (and the reference) produce after every frame (tests/test_gpu_fused.py).
//
// Domain: the plan is built on the same node set as x (plan nodes == rows of x): pure road graphs and MATSim graphs
// with SRC/DEST pseudo-nodes alike. Nmax <= 127, out-degree <= 126, agent ids < 2^24 (checked). A count that reaches Nmax
// leaves the reference's defined domain (it raises IndexError one or two steps later): the kernels set
// FLAG_COUNT_AT_NMAX in the device status word and the host raises when it reads it.
#include "fused_common.h"

struct PlanOut {   // CSR by source
  const int32_t* out_ptr;
  const int32_t* out_dst;
};

// ---- pack: build the dense / static words, the slot store and the agent SoA from x / agent_features -------------------
__global__ __launch_bounds__(FB) void k_pack_nodes(const float* __restrict__ x, Layout L, int64_t B, int64_t N,
                                                   const float* __restrict__ cong, FusedBufs fb, float4* st0_out,
                                                   PlanOut P) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;  // gid = i * B + b  (env-minor)
  if (gid >= B * N) return;
  const int64_t i = gid / B;
  const int64_t b = gid - i * B;
  const float* xi = x + b * L.bstride + i * L.ldx;
  const int Nmax = L.Nmax;
  const float n = xi[L.col_n()];
  const int q = (int)n;
  const float tail = (q >= 1 && q <= Nmax) ? xi[q - 1] : 0.0f;
  const float head = xi[0];
  if (q < 0 || q > 127 || !(head >= 0.0f && head < 16777216.0f) || !(tail >= 0.0f && tail < 16777216.0f))
    atomicOr(fb.flags, FLAG_PACK_RANGE);
  // HD_DIRTY: a non-zero dead slot (at or above the count: no garbage is pending after a pack), or a FIFO that has reached
  // its last slot
  bool dirty = q >= Nmax - 1;
  for (int sidx = q < 0 ? 0 : q; sidx < Nmax; ++sidx)
    dirty = dirty || xi[sidx] != 0.0f || xi[Nmax + sidx] != 0.0f || xi[2 * Nmax + sidx] != 0.0f;
  fb.hdp[gid] = make_uint2(((uint32_t)head << 8) | (uint32_t)(q & (int)HD_CNT) | (dirty ? HD_DIRTY : 0u), __float_as_uint(xi[2 * Nmax]));
  fb.tl[gid] = tl_word((uint32_t)tail, 0, TLF_AUTH);
  fb.gc8[gid] = (uint8_t)r1_code(-1);
  fb.post[gid] = ((uint32_t)tail << 8) | (q > 0 ? PF_NONEMPTY : 0u);
  // SELECTED_ROAD: the rank of the out-edge it names, or the raw value when it names none of them
  const float sv = xi[L.col_sel()];
  fb.sel[gid] = sv;
  uint32_t code = SEL_RAW;
  const int32_t k0 = P.out_ptr[i], k1 = P.out_ptr[i + 1];
  for (int32_t k = k1 - 1; k >= k0; --k)
    if ((float)P.out_dst[k] == sv && k - k0 < (int32_t)SEL_RAW) code = (uint32_t)(k - k0);
  fb.sel8[gid] = (uint8_t)code;
  if (i == 0) {
    for (int64_t sl_ = 0; sl_ < fb.acc_slots; ++sl_) {
      fb.acc_lp[sl_ * B + b] = 0;
      fb.acc_n[sl_ * B + b] = 0.0f;
      fb.acc_w[sl_ * B + b] = 0.0f;
    }
  }
  float* sl = fb.slots + gid * fb.lds;
  for (int sidx = 0; sidx < Nmax; ++sidx) slot_store(sl + SLW * sidx, xi[sidx], xi[Nmax + sidx], xi[2 * Nmax + sidx]);
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

// Static records (fused_common.h). in_rec[k].rank for in-edge k = (j -> i): the rank r of j's out-edges with
// (float)out_dst == ROAD_INDEX(i), i.e. the sel8 code of j that makes "SELECTED_ROAD(j) == ROAD_INDEX(i)"
// (src/direction_mpnn.py:77-79) true; INRANK_NONE when no out-edge of j does. Two matching out-edges (parallel dual
// edges) have no unique rank: FLAG_AMBIGUOUS_EDGES. Runs after k_pack_nodes (reads st0).
__global__ __launch_bounds__(FB) void k_pack_static(int64_t N, int64_t E, const int32_t* __restrict__ in_ptr,
                                                    const int32_t* __restrict__ in_src,
                                                    const int32_t* __restrict__ in_eid, PlanOut P,
                                                    const float* __restrict__ edge_attr,
                                                    const float4* __restrict__ st0, NodeRec* nodes, InRec* in_rec,
                                                    int32_t* out_pad, int32_t* flags) {
  const int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (i < 4) {   // padding entries: any valid row, never used
    in_rec[E + i] = InRec{0, (int32_t)INRANK_NONE, 0.0f, 0.0f, 0};
    out_pad[E + i] = 0;
  }
  if (i >= N) return;
  const float4 sti = st0[i];
  const int32_t a0 = in_ptr[i], a1 = in_ptr[i + 1], o0 = P.out_ptr[i], o1 = P.out_ptr[i + 1];
  NodeRec nr{a0, a1 - a0, o0, o1 - o0, sti.x, sti.y, sti.z, sti.w, entry_tt(sti, 0.0f), {0, 0, 0}, {0, 0, 0, 0}, {}};
  for (int q = 0; q < 4; ++q) {
    nr.out4[q] = q < o1 - o0 ? P.out_dst[o0 + q] : (int32_t)i;
    nr.in4[q] = InRec{0, (int32_t)INRANK_NONE, 0.0f, 0.0f, 0};
  }
  for (int32_t k = o0; k < o1; ++k) out_pad[k] = P.out_dst[k];
  for (int32_t k = a0; k < a1; ++k) {
    const int32_t j = in_src[k];
    const int32_t k0 = P.out_ptr[j], k1 = P.out_ptr[j + 1];
    int cnt = 0, r = (int)INRANK_NONE;
    for (int32_t kk = k0; kk < k1; ++kk)
      if ((float)P.out_dst[kk] == sti.z) {
        if (cnt == 0) r = kk - k0;
        ++cnt;
      }
    if (cnt > 1 || (cnt == 1 && r >= (int)SEL_RAW)) {
      atomicOr(flags, FLAG_AMBIGUOUS_EDGES);
      r = (int)INRANK_NONE;
    }
    in_rec[k] = InRec{j, r, edge_attr ? edge_attr[in_eid[k]] : 0.0f, st0[j].x, in_eid[k]};
    if (k - a0 < 4) nr.in4[k - a0] = in_rec[k];
  }
  nodes[i] = nr;
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

// departure-ordered window records of the insert kernel (after k_pack_agents)
__global__ __launch_bounds__(FB) void k_pack_window(int64_t B, int64_t A, FusedBufs fb, uint4* __restrict__ a_win) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;   // gid = b * A + k (sorted position k)
  if (gid >= B * A) return;
  const int64_t b = gid / A;
  const int32_t a = fb.a_order[gid];
  a_win[gid] = make_uint4(__float_as_uint(fb.a_dep_sorted[gid]), (uint32_t)fb.a_origin[b * A + a], (uint32_t)a, 0u);
  fb.a_ins[gid] = fb.a_status[b * A + a] != 0 ? 1 : 0;
}

// ---- reset: SimulatorEnv._reset on the packed state (zero FIFOs and counters, clear ON_WAY / DONE, re-arm cursors) -------
__global__ __launch_bounds__(FB) void k_fused_reset_nodes(int64_t B, int64_t N, FusedBufs fb) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (gid >= B * N) return;
  fb.hdp[gid] = make_uint2(0u, 0u);
  fb.tl[gid] = TLF_AUTH;
  fb.gc8[gid] = (uint8_t)r1_code(-1);
  fb.post[gid] = 0u;
  if (gid < B) {
    for (int64_t sl_ = 0; sl_ < fb.acc_slots; ++sl_) {
      fb.acc_lp[sl_ * B + gid] = 0;
      fb.acc_n[sl_ * B + gid] = 0.0f;
      fb.acc_w[sl_ * B + gid] = 0.0f;
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
  if (fb.a_ins) fb.a_ins[b * A + fb.a_rank[gid]] = 0;
}

// ---- export: rebuild the reference's x layout (three FIFO column blocks + NUMBER_OF_AGENT + SELECTED_ROAD) ----------
__global__ __launch_bounds__(FB) void k_export_rows(float* __restrict__ x, Layout L, int64_t B, int64_t N, FusedBufs fb,
                                                    float t_last, PlanOut P) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;
  const int Nmax = L.Nmax;
  if (gid >= B * N * Nmax) return;
  const int64_t row = gid / Nmax;  // row = i * B + b
  const int sidx = (int)(gid - row * Nmax);
  const int64_t i = row / B, b = row - i * B;
  float* xi = x + b * L.bstride + i * L.ldx;
  const uint32_t hd = fb.hdp[row].x;
  const uint32_t code = fb.gc8[row];
  const int n = (int)(hd & HD_CNT);
  const uint32_t tlw = fb.tl[row];
  const int g = pending_g(tlw, n, code, Nmax);
  const float* sl = fb.slots + row * fb.lds + SLW * phys(tl_hoff(tlw), sidx, Nmax);  // un-rotate the ring buffer
  if (g >= 0 && sidx == n) {  // pending garbage write of the last Direction update -> first dead slot
    const float tt = entry_tt(fb.st0[i], (float)g);
    xi[sidx] = 0.0f;
    xi[Nmax + sidx] = t_last;
    xi[2 * Nmax + sidx] = t_last + tt;
  } else if (!(hd & HD_DIRTY) && sidx >= n) {  // clean row: its dead slots are zero, whatever the store still holds there
    xi[sidx] = 0.0f;
    xi[Nmax + sidx] = 0.0f;
    xi[2 * Nmax + sidx] = 0.0f;
  } else {
    xi[sidx] = sl[0];
    xi[Nmax + sidx] = sl[1];
    xi[2 * Nmax + sidx] = sl[2];
  }
  if (sidx == 0) {
    xi[L.col_n()] = (float)n;
    const float sv = sel_value(fb, P.out_ptr, P.out_dst, i, row);
    xi[L.col_sel()] = sv;
    fb.sel[row] = sv;
  }
}

// ---- clean rows <-> exact slot store (hand-over to / from code that maintains every dead slot physically) ------------------
// tarl_fused_dead_slots(materialise = 1): write the zeros a clean row's dead slots stand for into the store (the slot at the
// count only when no garbage is pending there): afterwards the store is exact for every row. (materialise = 0): set each
// row's HD_DIRTY from the store — a non-zero dead slot, or a FIFO at its last slot — as pack does from x. The LDS-resident
// rollout kernel (rollout_env.hip), which keeps the reference's slot-by-slot bookkeeping, runs between the two.
__global__ __launch_bounds__(FB) void k_dead_slots(int64_t B, int64_t N, int Nmax, FusedBufs fb, int materialise) {
  const int64_t row = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (row >= B * N) return;
  const uint32_t hd = fb.hdp[row].x, tlw = fb.tl[row];
  const int n = (int)(hd & HD_CNT), hoff = tl_hoff(tlw);
  const int g = pending_g(tlw, n, fb.gc8[row], Nmax);
  float* sl = fb.slots + row * fb.lds;
  const int first = (g >= 0) ? n + 1 : n;       // the slot at the count holds the pending garbage (never stored) or is dead
  if (materialise) {
    if (hd & HD_DIRTY) return;
    for (int sidx = first; sidx < Nmax; ++sidx) {
      slot_store(sl + SLW * phys(hoff, sidx, Nmax), 0.0f, 0.0f, 0.0f);
    }
  } else {
    bool dirty = n >= Nmax - 1;
    for (int sidx = first; sidx < Nmax; ++sidx) {
      const float* z = sl + SLW * phys(hoff, sidx, Nmax);
      dirty = dirty || z[0] != 0.0f || z[1] != 0.0f || z[2] != 0.0f;
    }
    fb.hdp[row].x = (hd & ~HD_DIRTY) | (dirty ? HD_DIRTY : 0u);
  }
}

int tarl_fused_dead_slots(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax, int materialise,
                          tarl_stream stream) {
  if (plan->N == 0) return TARL_OK;
  hipLaunchKernelGGL(k_dead_slots, dim3((unsigned)ceil_div(B * plan->N, FB)), dim3(FB), 0, (hipStream_t)stream, B, plan->N,
                     (int)Nmax, tarl_to_bufs(f), materialise);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
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

// ---- choice phase (env-minor: lane = environment, a workgroup walks a chunk of nodes) --------------------------------
// Consecutive nodes of one environment share Philox blocks (index = b*G + g), so a chunk costs ~nchunk/4 + 1 Philox
// evaluations per lane instead of one per node. The action AND the new SELECTED_ROAD are one byte per (node, env): the
// rank of the chosen out-edge (sel_out); a node that draws nothing (no out-edges, or u beyond the last threshold through
// rounding) carries its previous code over from sel_prev with SEL_CARRIED set.
__device__ __forceinline__ void fused_choice_body(unsigned bx, unsigned by, const int32_t* __restrict__ out_ptr,
                                                  const int32_t* __restrict__ out_eid,
                                                  const int32_t* __restrict__ group_of_node, int64_t G, int64_t B,
                                                  int64_t N, long long* __restrict__ acc_lp, int64_t acc_slots,
                                                  const float* __restrict__ thr, const long long* __restrict__ lgt,
                                                  const float* __restrict__ uniform, uint64_t pseed, uint64_t pcounter,
                                                  uint8_t* __restrict__ sel_out, const uint8_t* __restrict__ sel_prev,
                                                  int32_t* __restrict__ choice, int nchunk, int want_lp,
                                                  int64_t env_base) {
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
    uint32_t code = SEL_RAW;
    bool found = false;
    const int32_t gi = group_of_node[i];
    if (gi >= 0) {
      const float u = uniform ? uniform[b * G + gi] : rng.uniform(pseed, pcounter, (uint64_t)((env_base + b) * G + gi));
      // first out-edge (plan order) whose threshold exceeds u. Every table operand is wave-uniform (scalar loads), the
      // per-lane part is compare + select: no dependent vector gathers
      long long lpn = 0;
      const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];
      for (int32_t k = k0; k < k1; ++k) {
        const bool hit = !found && (u < thr[k]);
        const long long lgk = lgt[k];
        code = hit ? (uint32_t)(k - k0) : code;
        ch = hit ? out_eid[k] : ch;
        lpn = hit ? lgk : lpn;
        found = found || hit;
      }
      if (found) lp += lpn; else bad = true;
    }
    if (!found) code = (sel_prev[row] & 0x7Fu) | SEL_CARRIED;   // keeps its previous SELECTED_ROAD
    sel_out[row] = (uint8_t)code;
    if (choice) __builtin_nontemporal_store(ch, &choice[row]);  // frame API: the action as an edge id (-1: none)
  }
  // infeasible action (some node picked nothing): poison the accumulator far beyond any legitimate sum
  if (want_lp)
    atomicAdd((unsigned long long*)&acc_lp[(int64_t)(by % (unsigned)acc_slots) * B + b],
              (unsigned long long)(bad ? -(1ll << 50) : lp));  // up to 2^13 chunks cannot wrap
}

__global__ __launch_bounds__(TILE) void k_fused_choice(const int32_t* __restrict__ out_ptr,
                                                       const int32_t* __restrict__ out_eid,
                                                       const int32_t* __restrict__ group_of_node, int64_t G, int64_t B,
                                                       int64_t N, long long* __restrict__ acc_lp, int64_t acc_slots,
                                                       const float* __restrict__ thr,
                                                       const long long* __restrict__ lgt,
                                                       const float* __restrict__ uniform, uint64_t pseed,
                                                       uint64_t pcounter, uint8_t* __restrict__ sel_out,
                                                       const uint8_t* __restrict__ sel_prev,
                                                       int32_t* __restrict__ choice, int nchunk, int want_lp,
                                                       int64_t env_base) {
  fused_choice_body(blockIdx.x, blockIdx.y, out_ptr, out_eid, group_of_node, G, B, N, acc_lp, acc_slots, thr, lgt,
                    uniform, pseed, pcounter, sel_out, sel_prev, choice, nchunk, want_lp, env_base);
}

// ---- the whole rollout's actions in one pass (state-independent policy) -----------------------------------------------------
// The live policy's distribution does not read the state, so the T actions of a rollout are T independent draws from ONE
// set of tables: instead of a choice launch per frame (instruction-bound: ~16 us of scalar table walks that the frame loop
// had to wait for), the action buffer [T][N][B] is filled by launches of 32 frames each on a side stream, in the shadow of
// the latency-bound frame kernels. One workgroup = one (environment tile, frame): it walks ALL nodes, so a frame's
// log-prob is a register sum (no atomics) — the same Philox indices, thresholds and 2^-32 fixed-point terms as the
// per-frame kernel, hence the same bits.
// A node that draws nothing (u at / beyond its last threshold through rounding, ~1e-7 per draw) keeps the PREVIOUS frame's
// SELECTED_ROAD: frames are concurrent here, so such an entry is marked SEL_UNRESOLVED, listed, and resolved by
// k_choice_fixup (walk back to the last frame that drew something; before frame 0: the packed state's code).
#define SEL_UNRESOLVED 0xFEu      // = SEL_CARRIED | 0x7E: no rank (out-degree <= 126) and not SEL_RAW
#define FIX_CAP 65536
struct __attribute__((aligned(4))) PRec {   // per out-edge (CSR order, padded by 4): inverse-CDF threshold + log-prob
  float thr;
  int32_t lg_lo, lg_hi;                      // log(p + 1e-8) in 2^-32 fixed point
  int32_t pad;
};

__global__ __launch_bounds__(FB) void k_pack_ptab(int64_t E, const float* __restrict__ thr,
                                                  const long long* __restrict__ lgt, PRec* __restrict__ ptab) {
  const int64_t k = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (k >= E + 4) return;
  if (k < E) {
    const long long lg = lgt[k];
    ptab[k] = PRec{thr[k], (int32_t)(uint32_t)(lg & 0xffffffffll), (int32_t)(lg >> 32), 0};
  } else {
    ptab[k] = PRec{2.0f, 0, 0, 0};
  }
}

// per node: everything the all-frames draw needs about it in ONE aligned scalar read whose address depends on the node
// index alone (through the CSR range it would be two dependent scalar rounds per node). The thresholds are kept as
// INTEGERS: a draw is u = f(m) = ((float)m + 0.5f) * 2^-24 of the 24-bit m = word >> 8, f is non-decreasing, so
// u >= thr  <=>  m >= ithr with ithr = min{m : f(m) >= thr} (2^24: never) — found by bisection with f itself, so the
// comparison is the same for every m and the three conversion instructions per draw are not issued. The log-prob terms
// (second half of the record) are copied into LDS per workgroup and looked up by the count.
struct __attribute__((aligned(64))) PNode {
  int32_t gi, deg, out0, pad;
  uint32_t ithr[4];                          // 2^24 beyond the out-degree
  int32_t lg_lo[4], lg_hi[4];
};
static_assert(sizeof(PNode) == 64, "PNode: a 32-byte scalar read + 32 bytes of log-prob terms");
typedef int32_t i32x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(FB) void k_pack_pnode(int64_t N, const NodeRec* __restrict__ nodes,
                                                   const int32_t* __restrict__ group_of_node,
                                                   const float* __restrict__ thr, const long long* __restrict__ lgt,
                                                   PNode* __restrict__ pnode) {
  const int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (i >= N) return;
  PNode pn;
  pn.gi = group_of_node[i];
  pn.deg = nodes[i].out_deg;
  pn.out0 = nodes[i].out0;
  pn.pad = 0;
  for (int q = 0; q < 4; ++q) {
    const bool in = q < pn.deg;
    const long long lg = in ? lgt[pn.out0 + q] : 0ll;
    const float th = in ? thr[pn.out0 + q] : INFINITY;
    uint32_t lo = 0u, hi = 1u << 24;         // the first m with f(m) >= th (NaN / +inf: none)
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (u01_open(mid << 8) >= th)
        hi = mid;
      else
        lo = mid + 1u;
    }
    pn.ithr[q] = lo;
    pn.lg_lo[q] = (int32_t)(uint32_t)(lg & 0xffffffffll);
    pn.lg_hi[q] = (int32_t)(lg >> 32);
  }
  pnode[i] = pn;
}

#define CHOICE_SEG 128    // nodes per workgroup of the all-frames choice (a frame's log-prob = the sum of its segments)
// one node's draw from the first half of its record v and the 24 random bits m: the inverse CDF is a count — thresholds
// are non-decreasing (fp32 roundings of a running double sum), so the first q with u < thr[q] is the number of thresholds
// at or below u. lgn: the node's four log-prob terms in LDS (one entry of slack behind them).
__device__ __forceinline__ void choice_node(const i32x8 v, const uint32_t m, const long long* lgn,
                                            const PRec* __restrict__ ptab, uint32_t row, int64_t t,
                                            uint8_t* __restrict__ out, long long& lp, bool& bad,
                                            int32_t* __restrict__ fix, int32_t* __restrict__ flags) {
  const int32_t deg = v[1];
  const PRec* pr = ptab + v[2];
  uint32_t cnt = (m >= (uint32_t)v[4] ? 1u : 0u) + (m >= (uint32_t)v[5] ? 1u : 0u) + (m >= (uint32_t)v[6] ? 1u : 0u) +
                 (m >= (uint32_t)v[7] ? 1u : 0u);
  long long lpn = lgn[cnt < 4u ? cnt : 4u];
  if (deg > 4) {                         // wave-uniform: the thresholds further down the table, as floats
    const float u = u01_open(m << 8);
    for (int32_t q = 4; q < deg; ++q) cnt += (u >= pr[q].thr) ? 1u : 0u;
    if (cnt >= 4u && cnt < (uint32_t)deg) {
      const PRec px = pr[cnt];
      lpn = ((long long)px.lg_hi << 32) | (long long)(uint32_t)px.lg_lo;
    }
  }
  const bool found = cnt < (uint32_t)deg;
  lp += lpn;                             // a node that drew nothing poisons the whole sum below
  if (!found) {
    bad = true;
    const int32_t pos = atomicAdd(&fix[0], 1);
    if (pos < FIX_CAP) {
      fix[2 + 2 * pos] = (int32_t)t;
      fix[3 + 2 * pos] = (int32_t)row;
    } else {
      atomicOr(flags, FLAG_CHOICE_OVERFLOW);
    }
  }
  out[row] = (uint8_t)(found ? cnt : SEL_UNRESOLVED);
}

// QUAD (every node has out-edges and their number is a multiple of four: draw index = b * N + i, so nodes 4k .. 4k + 3
// of an environment share one Philox block): four nodes per step on the block's four words, without the block compare
// and the word select of the general form. Same indices, same words: the same draws.
template <bool QUAD>
__global__ __launch_bounds__(TILE) void k_fused_choice_all(const PNode* __restrict__ pnode, const PRec* __restrict__ ptab,
                                                           const int32_t* __restrict__ group_of_node, uint32_t G,
                                                           uint32_t B, uint32_t N, int64_t t0, uint64_t pseed,
                                                           uint64_t pcounter0, const uint8_t* __restrict__ sel0,
                                                           uint8_t* __restrict__ choice, long long* __restrict__ lp_acc,
                                                           int32_t* __restrict__ fix, int32_t* __restrict__ flags,
                                                           uint64_t env_base) {
  __shared__ long long s_lg[CHOICE_SEG * 4 + 4];
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t t = t0 + blockIdx.z;
  const uint32_t i0 = blockIdx.y * CHOICE_SEG;
  const uint32_t i1 = (i0 + CHOICE_SEG < N) ? i0 + CHOICE_SEG : N;
  for (uint32_t e = threadIdx.x; e < (i1 - i0) * 4; e += blockDim.x) {
    const PNode& pn = pnode[i0 + (e >> 2)];
    s_lg[e] = ((long long)pn.lg_hi[e & 3] << 32) | (long long)(uint32_t)pn.lg_lo[e & 3];
  }
  __syncthreads();
  if (b >= B) return;
  uint8_t* out = choice + t * (int64_t)N * B;
  long long lp = 0;
  bool bad = false;
  // The node loop issues no vector load at all (a conditional load inside it makes the compiler wait for the previous
  // iteration's store, vmcnt(0), every time round) and ONE scalar read per node, the first half of its PNode record, read
  // as one 8-dword vector (member by member the compiler splits it into dependent reads around the out-degree test).
  // Nodes without out-edges (they keep SELECTED_ROAD) get a loop of their own.
  if (QUAD) {
    const uint64_t blk0 = (env_base + (uint64_t)b) * (G >> 2);
    const uint64_t counter = pcounter0 + (uint64_t)t;
    for (uint32_t i = i0; i < i1; i += 4) {   // i0 and i1 are multiples of four
      const uint64_t blk = blk0 + (i >> 2);
      uint32_t o[4];
      philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)pseed,
                    (uint32_t)(pseed >> 32), o);
      const uint32_t row = i * B + b;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const i32x8 v = *(const i32x8*)(pnode + i + r);
        choice_node(v, o[r] >> 8, s_lg + (i - i0 + r) * 4, ptab, row + r * B, t, out, lp, bad, fix, flags);
      }
    }
  } else {
    PhiloxRun rng;
    for (uint32_t i = i0; i < i1; ++i) {      // i, and everything indexed by it alone, is wave-uniform
      const i32x8 v = *(const i32x8*)(pnode + i);
      if (v[1] == 0) continue;                // no out-edges (== no group)
      const uint32_t w = rng.word(pseed, pcounter0 + (uint64_t)t, (env_base + (uint64_t)b) * G + (uint64_t)v[0]);
      choice_node(v, w >> 8, s_lg + (i - i0) * 4, ptab, i * B + b, t, out, lp, bad, fix, flags);
    }
  }
  if (G != N) {
    for (uint32_t i = i0; i < i1; ++i)
      if (group_of_node[i] < 0) {
        const uint32_t row = i * B + b;
        out[row] = (uint8_t)((sel0[row] & 0x7Fu) | SEL_CARRIED);   // no out-edges: SELECTED_ROAD never changes
      }
  }
  // order-independent 2^-32 fixed-point sum over the node segments; an action with a node that drew nothing is poisoned
  // far beyond any legitimate sum (it is infeasible: log_prob = -inf, src/reinforcement_learning.py:88-92)
  if (lp_acc) atomicAdd((unsigned long long*)&lp_acc[t * B + b], (unsigned long long)(bad ? -(1ll << 50) : lp));
}

__global__ __launch_bounds__(FB) void k_choice_lp_finish(int64_t n, long long* __restrict__ lp_acc,
                                                         float* __restrict__ log_prob) {
  const int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (i >= n) return;
  const long long v = lp_acc[i];
  lp_acc[i] = 0;     // re-armed for the next rollout
  log_prob[i] = (v < -(1ll << 49)) ? -INFINITY : (float)((double)v / LP_FIX);
}

__global__ __launch_bounds__(ENVB) void k_choice_fixup(uint32_t B, uint32_t N, const uint8_t* __restrict__ sel0,
                                                       uint8_t* __restrict__ choice, int32_t* __restrict__ fix) {
  int32_t cnt = fix[0];
  if (cnt > FIX_CAP) cnt = FIX_CAP;
  const int64_t NB = (int64_t)N * B;
  for (int32_t idx = threadIdx.x; idx < cnt; idx += ENVB) {
    const int64_t t = fix[2 + 2 * idx];
    const int64_t row = fix[3 + 2 * idx];
    uint32_t c = SEL_UNRESOLVED;
    for (int64_t tt = t - 1; tt >= 0 && c == SEL_UNRESOLVED; --tt) c = choice[tt * NB + row];
    if (c == SEL_UNRESOLVED) c = sel0[row];
    choice[t * NB + row] = (uint8_t)((c & 0x7Fu) | SEL_CARRIED);   // a concurrent reader sees the marker or this: same walk
  }
  __syncthreads();
  if (threadIdx.x == 0) fix[0] = 0;
}

// element idx of a [node][env] array. O32: through a 32-bit BYTE offset from the (scalar) base — `global_load ... v_off,
// s[base]` instead of a 64-bit address per lane (one to two vector instructions fewer per access; row pass 212 -> 204 us).
// Valid while N * B * sizeof(T) < 2^32, i.e. N * B < 2^29 for the 8-byte words: launch_rows picks the instantiation.
template <bool O32, class T>
__device__ __forceinline__ T& at32(T* base, uint32_t idx) {
  if (O32) return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (size_t)(idx * (uint32_t)sizeof(T)));
  return base[idx];
}
template <bool O32, class T>
__device__ __forceinline__ const T& at32(const T* base, uint32_t idx) {
  if (O32) return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)(idx * (uint32_t)sizeof(T)));
  return base[idx];
}

// ---- Direction gather on the dense words (env-minor: lane = environment) ---------------------------------------------------
// Two passes inside the workgroup. Pass 1 (every (node, environment) pair of the chunk): admissibility masks and the
// summed turn probability P — no random numbers — and the default post word (nobody chosen). A pair needs the Gumbel
// race only when P > 0, i.e. when some in-edge is admissible; that is a few percent of the pairs, but scattered over all
// lanes, so every wave would still pay for the Philox block and the two logs per edge. The pairs with P > 0 are
// therefore appended to an LDS list and pass 2 walks that list densely (one lane per pair), re-evaluating the pair with
// its noise — same Philox indices, same expressions: the result is bit-identical to evaluating everything.
#define DIR_LIST (TILE * 8)
// does upstream row j (dense words hj, SELECTED_ROAD code cj) send its head to the row behind an in-edge of rank code rk?
// EXACT: a raw SELECTED_ROAD (cj == SEL_RAW: a value pack found among none of j's out-edges) is compared as a value.
template <bool EXACT>
__device__ __forceinline__ bool edge_admissible(uint32_t cj, int32_t rk, const float* __restrict__ sel_raw, int64_t jrow,
                                                uint2 hj, float max_j, float road_i, float n_i, float max_i, float t) {
  bool heads_here = cj == (uint32_t)rk;   // ranks are < SEL_RAW: a raw code never matches here
  if (EXACT && cj == SEL_RAW) heads_here = sel_raw[jrow] == road_i;
  const float dep = __uint_as_float(hj.y), n_j = (float)(hj.x & HD_CNT);
  // (bitwise on purpose: lane masks and'ed / or'ed by the scalar unit, no short-circuit control flow)
  const bool m1 = (dep <= t) & (n_i < max_i - TARL_CONGESTION_FILE) & (n_j > 0.0f);
  const bool m2 = ((dep - t) < -10.0f) & ((max_j - TARL_CONGESTION_FILE) <= n_j) & ((max_j - n_j) <= (max_i - n_i));
  return heads_here & (m1 | m2);
}

// Every read-only array is its own `const __restrict__` kernel argument: topology, statics and edge constants are
// wave-uniform and must compile to SCALAR loads (through a struct member the compiler has to assume they alias the post
// stores and falls back to dependent vector loads — measured: 48 -> 85 us per launch).
// CNT: the row's own count comes from a byte per (row, environment) — in a rollout the count buffer's slice of the frame
// before, which the row pass and the insert kernel have just written — instead of its 8-byte head words: of its own row the
// dense pass needs the count and the tail word only (12 -> 5 bytes per pair).
template <int NCH, bool SIB, bool CNT, bool O32>
__global__ __launch_bounds__(TILE) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fused_direction(
    const NodeRec* __restrict__ nodes, const InRec* __restrict__ in_rec, const int32_t* __restrict__ in_eid,
    const float* __restrict__ log_edge_attr, const uint2* __restrict__ hdp, const uint32_t* __restrict__ tl,
    const uint8_t* __restrict__ cnt8, const uint8_t* __restrict__ gc8, const float* __restrict__ slots, int64_t lds,
    int Nmax, const uint8_t* __restrict__ sel8,
    const float* __restrict__ sel_raw,
    const float* __restrict__ gumbel, float* __restrict__ dtt, uint32_t* __restrict__ post, float log_eps, float t,
    float t_prev, uint64_t seed, uint64_t counter, uint32_t E, uint32_t B, uint32_t N, FrameOut out, uint64_t env_base) {
  __shared__ int32_t s_n;
  __shared__ uint16_t s_item[DIR_LIST];   // (node offset in the chunk) * TILE + lane
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;   // row indices are 32-bit: N * B < 2^31 (host check)
  const bool valid = b < B;
  // The chunks are walked from the LAST to the first: the row pass before this launch (and the one after it) walks them
  // from the first to the last, so each of the two kernels starts on the quarter of the state the other has just touched —
  // what the 256 MB Infinity Cache still holds of a 1.7 GB frame. Forward order here: Direction 203 us and row pass 215 us
  // per launch instead of 188 and 205 (same box).
  const uint32_t i0 = (gridDim.y - 1u - blockIdx.y) * NCH;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  // pass 1, branch-free form: a raw SELECTED_ROAD code is only noted; if any lane of the workgroup saw one (never in a
  // rollout: every node with out-edges draws an action in every frame) the pass is repeated in its exact form.
  // NCH rows per lane, ALL their loads (own words + the first four in-edges' gathers) issued before anything is used:
  // the pass is bound by memory latency and by scalar-instruction issue, so requests in flight per wave and address
  // arithmetic per request are what counts: one node record + one base address fetch a row's statics.
  bool raw_seen = false;
  if (valid) {
    uint2 hj[NCH][4];
    uint32_t ncnt[NCH];   // the row's own count
    uint32_t tlw[NCH], cj[NCH][4];
    // SIB: sibling rows — the roads that leave one intersection — have the same upstream rows. When every chunk's rows
    // list the same first four sources (checked once per graph, tarl_plan::siblings4), the upstream words are gathered
    // once per chunk instead of once per row: 16 requests per lane instead of 40.
#pragma unroll
    for (int r = 0; r < NCH; ++r) {
      const uint32_t i = (i0 + r < N) ? i0 + r : N - 1;
      const uint32_t row = i * B + b;
      ncnt[r] = CNT ? (uint32_t)cnt8[row] : (at32<O32>(hdp, row).x & HD_CNT);
      tlw[r] = at32<O32>(tl, row);
      const InRec* ir = nodes[i].in4;   // the first four in-edge records travel in the node record
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (SIB && r > 0) {
          hj[r][q] = hj[0][q];
          cj[r][q] = cj[0][q];
        } else {
          const uint32_t jrow = (uint32_t)ir[q].src * B + b;
          hj[r][q] = at32<O32>(hdp, jrow);
          cj[r][q] = sel8[jrow] & 0x7Fu;
        }
      }
    }
    // SIB: the chunk's rows share their upstream rows, so what depends on the upstream row alone is evaluated once per
    // chunk and folded into two floats per upstream row: x1 = 0 where its head is due and it holds somebody (else NaN),
    // slk = its free slots where its head is overdue by more than 10 s and it is full (else NaN). Per (row, in-edge) the
    // two admissibility tests of edge_admissible are then `n_i + x1 < MAX_i - 3` and `slk <= MAX_i - n_i` (a NaN compares
    // false; n_i + 0 is n_i): the same comparisons on the same values, 6 instead of 14 vector instructions per pair.
    float x1[4], slk[4];
    if (SIB) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint2 hq = hj[0][q];
        const float max_j = nodes[i0].in4[q].max_src;     // (beyond the in-degree: padding, masked by the rank test)
        const float dep = __uint_as_float(hq.y), n_j = (float)(hq.x & HD_CNT);
        const bool a1 = (dep <= t) & (n_j > 0.0f);
        const bool a2 = ((dep - t) < -10.0f) & ((max_j - TARL_CONGESTION_FILE) <= n_j);
        x1[q] = a1 ? 0.0f : __uint_as_float(0x7fc00000u);
        slk[q] = a2 ? (max_j - n_j) : __uint_as_float(0x7fc00000u);
        // a raw SELECTED_ROAD code upstream sends the workgroup to the exact pass; noted per upstream row, whatever the
        // rows' in-degrees (a padding entry names row 0: at worst a repeat that was not needed)
        raw_seen = raw_seen | (cj[0][q] == SEL_RAW);
      }
    }
#pragma unroll
    for (int r = 0; r < NCH; ++r) {
      const uint32_t i = i0 + r;   // i, and everything indexed by it alone, is wave-uniform
      if (i < N) {
        const uint32_t row = i * B + b;
        const NodeRec& nr = nodes[i];
        const InRec* ir4 = nr.in4;
        const InRec* ir = in_rec + nr.in0;
        const float max_i = nr.maxn, n_i = (float)ncnt[r], road_i = nr.road;
        const float lim_i = max_i - TARL_CONGESTION_FILE, room_i = max_i - n_i;
        float P = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (q < nr.in_deg) {   // wave-uniform
            if (!SIB) raw_seen = raw_seen | (cj[r][q] == SEL_RAW);
            bool m;
            if (SIB)
              m = (cj[0][q] == (uint32_t)ir4[q].rank) & (((n_i + x1[q]) < lim_i) | (slk[q] <= room_i));
            else
              m = edge_admissible<false>(cj[r][q], ir4[q].rank, sel_raw, 0, hj[r][q], ir4[q].max_src, road_i, n_i, max_i, t);
            P = P + ir4[q].ea * (m ? 1.0f : 0.0f);
          }
        }
        for (int32_t q = 4; q < nr.in_deg; ++q) {   // in-degree above four: the rest one by one
          const uint32_t jrow = (uint32_t)ir[q].src * B + b;
          const uint2 hx = hdp[jrow];
          const uint32_t cx = sel8[jrow] & 0x7Fu;
          raw_seen = raw_seen | (cx == SEL_RAW);
          const bool m = edge_admissible<false>(cx, ir[q].rank, sel_raw, 0, hx, ir[q].max_src, road_i, n_i, max_i, t);
          P = P + ir[q].ea * (m ? 1.0f : 0.0f);
        }
        // nobody chosen (pass 2 overwrites). A row that idled in the last frame (tail word without TLF_AUTH) still holds
        // exactly this word from the frame before: whatever changes a row's tail, empties or fills it, or hands it an
        // arrival makes it an event row, and event rows and inserts set the flag
        if (tlw[r] & TLF_AUTH) at32<O32>(post, row) = (tlw[r] & ~0xFFu) | (ncnt[r] ? PF_NONEMPTY : 0u) | PF_TLAUTH;
        if (P > 0.0f) s_item[atomicAdd(&s_n, 1)] = (uint16_t)(r * TILE + threadIdx.x);
      }
    }
  }
  const bool redo = __syncthreads_or(raw_seen ? 1 : 0) != 0;
  if (out.dtt_node && valid && b < (uint32_t)out.m_env) {   // metric environments: delta_travel_time once per upstream node
    for (uint32_t i = i0; i < i0 + NCH && i < N; ++i) {
      const uint32_t row = i * B + b;
      const uint2 mw = hdp[row];
      const uint32_t tw = tl[row];
      const bool lazy_i = (mw.x & HD_CNT) == 0u && !(tw & TLF_AUTH);   // empty and idle in the last frame: its garbage
      const float arr_i = head_arrival(slots, lds, gc8, row, mw.x, tw, Nmax, t_prev);   // head arrived at that frame's clock and
      const float dep_i = lazy_i ? t_prev + nodes[i].tt0 : __uint_as_float(mw.y);   // departs tt0 later (never stored)
      const float d = (dep_i - arr_i) - nodes[i].ff;
      out.dtt_node[(int64_t)i * out.m_env + b] = d > 0.0f ? d : (d != d ? d : 0.0f);
    }
  }
  if (redo || dtt) {   // exact form and / or the per-edge delta_travel_time of the frame API (uniform branch)
    if (redo && threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (valid) {
      for (uint32_t i = i0; i < i0 + NCH && i < N; ++i) {
        const uint32_t row = i * B + b;
        const uint2 mw = hdp[row];
        const NodeRec& nr = nodes[i];
        const InRec* ir = in_rec + nr.in0;
        const float max_i = nr.maxn, n_i = (float)(mw.x & HD_CNT), road_i = nr.road;
        float P = 0.0f;
        for (int32_t q = 0; q < nr.in_deg; ++q) {
          const int32_t j = ir[q].src;
          const uint32_t jrow = (uint32_t)j * B + b;
          const uint2 hx = hdp[jrow];
          if (redo) {
            const bool m = edge_admissible<true>(sel8[jrow] & 0x7Fu, ir[q].rank, sel_raw, jrow, hx, ir[q].max_src, road_i,
                                                 n_i, max_i, t);
            P = P + ir[q].ea * (m ? 1.0f : 0.0f);
          }
          if (dtt) {   // per-edge side output of DirectionMPNN.message (src/direction_mpnn.py:94-96): a property of j
            const uint32_t tj = tl[jrow];
            const bool lazy_j = (hx.x & HD_CNT) == 0u && !(tj & TLF_AUTH);
            const float arr_j = head_arrival(slots, lds, gc8, jrow, hx.x, tj, Nmax, t_prev);
            const float dep_j = lazy_j ? t_prev + nodes[j].tt0 : __uint_as_float(hx.y);
            const float d = (dep_j - arr_j) - nodes[j].ff;
            dtt[(int64_t)b * E + in_eid[nr.in0 + q]] = d > 0.0f ? d : (d != d ? d : 0.0f);
          }
        }
        if (redo && P > 0.0f) s_item[atomicAdd(&s_n, 1)] = (uint16_t)((i - i0) * TILE + threadIdx.x);
      }
    }
    __syncthreads();
  }
  const int32_t cnt = s_n;
  for (int32_t idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
    const int32_t item = s_item[idx];
    const uint32_t i = i0 + item / TILE;
    const uint32_t bb = blockIdx.x * blockDim.x + (item % TILE);
    const uint32_t row = i * B + bb;
    const uint2 mw = hdp[row];
    const NodeRec& nr = nodes[i];
    const int32_t in0 = nr.in0, in_deg = nr.in_deg;
    const float max_i = nr.maxn, n_i = (float)(mw.x & HD_CNT), road_i = nr.road;
    float P = 0.0f, best = -FLT_MAX;
    uint32_t best_id = 0u;
    PhiloxRun rng;
    // the first four in-edges: their records travel in the node record, so the gathers (upstream words, edge constant)
    // are ONE dependent round behind the list entry; all four are requested before the first is used
    InRec rc4[4];
    uint2 hx4[4];
    uint32_t cx4[4];
    float le4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rc4[q] = nr.in4[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t jrow = (uint32_t)rc4[q].src * B + bb;     // beyond in_deg: row 0 (valid, ignored)
      hx4[q] = hdp[jrow];
      cx4[q] = sel8[jrow] & 0x7Fu;
      le4[q] = log_edge_attr[rc4[q].eid];
    }
    auto race = [&](int32_t q, const InRec& rc, const uint2 hx, const uint32_t cx, const float le) {
      const int32_t k = in0 + q;
      const uint32_t jrow = (uint32_t)rc.src * B + bb;
      const bool m = edge_admissible<true>(cx, rc.rank, sel_raw, jrow, hx, rc.max_src, road_i, n_i, max_i, t);
      P = P + rc.ea * (m ? 1.0f : 0.0f);
      float g;
      if (gumbel) {
        g = gumbel[(int64_t)bb * E + rc.eid];
      } else {
        const float u = rng.uniform(seed, counter, (env_base + (uint64_t)bb) * E + (uint64_t)k);
        g = gumbel_from_u01(u);
      }
      const float score = (m ? le : log_eps) + g;
      if (score > best) {
        best = score;
        best_id = hx.x >> 8;
      }
    };
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < in_deg) race(q, rc4[q], hx4[q], cx4[q], le4[q]);
    for (int32_t q = 4; q < in_deg; ++q) {
      const InRec rc = in_rec[in0 + q];
      const uint32_t jrow = (uint32_t)rc.src * B + bb;
      race(q, rc, hdp[jrow], sel8[jrow] & 0x7Fu, log_edge_attr[rc.eid]);
    }
    const uint32_t who = (P > 0.0f) ? best_id : 0u;
    if (who != 0u) post[row] = (who << 8) | PF_NONEMPTY | PF_ARRIVED;
  }
}

// ---- the row pass: Direction update + Response pop + withdraw on the slot store, then refresh the dense words -----------
// The pass has two phases inside the workgroup (the trick of the Direction gather, for the same reason). Phase A: every
// (row, environment) pair, on registers only — Response test from the prefetched post words, idle / event decision, and
// for an IDLE row (nothing enqueued, no pop, head not due: ~90 % of the pairs in a filling network) the refreshed dense
// words. EVENT rows (something moves: slot store, event word, agent rows) are only LISTED in LDS. Phase B walks that
// list densely, one lane per event row. The event path is hundreds of instructions long; issued per wave it ran for every
// wave (some lane nearly always has an event) at a few active lanes — the kernel is issue-bound, so that was its cost.
struct RowHead {      // what phase A derives from a row's dense words and hands to phase B through registers / the list
  uint32_t n0i, head_id0, tail0, who, arrived;
  bool pop;
};

// the statics of a row that phase A needs, loaded (scalar) at the head of the kernel instead of where they are used, behind
// the waits for the vector loads (one dependent scalar round per row there: -2 % on the pass)
struct RowStat {
  int32_t out_deg, out0;
  float tt0, maxn;
};
// phase A of one row: returns true when the row is an event row (nothing written), false when it was idle (words written)
// FAPI: the frame API's outputs (fp32 counts, env-major pop / withdraw masks) may be present; rollouts instantiate without
// them (their null tests, addresses and registers leave the idle path)
template <bool FAPI, bool O32>
__device__ __forceinline__ bool row_phase_a(uint32_t i, uint32_t b, const RowStat nr, const uint32_t* __restrict__ post,
                                            uint32_t pa, uint2 hp, uint32_t tlw, const uint32_t (&pj4)[4], int Nmax,
                                            uint32_t B, uint32_t N, const FusedBufs& fb, int64_t A, float t,
                                            const FrameOut& out, bool* pop_out, float* n_out) {
  const uint32_t row = i * B + b;   // 32-bit row indices: N * B < 2^31 (host check)
  const uint32_t n0i = hp.x & HD_CNT, head_id0 = hp.x >> 8;
  const uint32_t arrived = pa & PF_ARRIVED;
  const uint32_t who = arrived ? (pa >> 8) : 0u;   // the agent the Direction update enqueues
  // Response message + max-aggregate from the post words (state after the Direction update of every row)
  bool pop = false;
  {
    const uint32_t head = (n0i == 0u) ? who : head_id0;   // head after the Direction update
    const bool up = (n0i + arrived) > 0u;
    // (bitwise on purpose: four lane masks and'ed / or'ed, no short-circuit control flow)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      pop = pop | ((q < nr.out_deg) & up & ((pj4[q] & PF_NONEMPTY) != 0u) & ((pj4[q] >> 8) == head));
    for (int32_t q = 4; q < nr.out_deg; ++q) {   // out-degree above four: the rest one by one
      const uint32_t pj = at32<O32>(post, (uint32_t)fb.out_pad[nr.out0 + q] * B + b);
      pop = pop | (up & ((pj & PF_NONEMPTY) != 0u) & ((pj >> 8) == head));
    }
  }
  const int q = (int)n0i;
  const bool lazy = (who == 0u) & (q < Nmax - 1);
  const uint32_t ni = n0i + arrived;   // count after the Direction update
  if ((int)ni >= Nmax) atomicOr(fb.flags, FLAG_COUNT_AT_NMAX);
  const uint32_t head_id = (n0i == 0u) ? who : head_id0;
  // an empty row's head is this frame's garbage triple (0, t, t + tt): its departure needs the division only there
  const float head_dep = (n0i == 0u) ? t + nr.tt0 : __uint_as_float(hp.y);
  // is the (unpopped) head a withdraw candidate? (the first two tests of the withdraw scan, on registers)
  const bool due = (!pop) & (ni > 0u) & ((int64_t)head_id < A) & (head_dep <= t);
  *pop_out = pop;
  // the row's count after the pass is ni - pop - (agents withdrawn): all but the last term is known here, so the row's
  // share of the environment's count sum rides on the idle rows' coalesced atomic; the event path adds only what it
  // withdraws (rare) instead of one scattered global atomic per event row
  *n_out = (float)(ni - (pop ? 1u : 0u));
  // The count byte of EVERY row goes out here, from the lane that owns the row (whole-wave, coalesced): an event row's count
  // after the pass is ni - pop - (agents withdrawn), and all but the last term is known; phase B rewrites the byte only
  // when it withdraws somebody (rare). One scattered store per event row less on the event path (an extra 4-byte store
  // per event row was measured at +13 ... +30 % on the pass).
  if (out.counts8) __builtin_nontemporal_store((uint8_t)(ni - (pop ? 1u : 0u)), &at32<O32>(out.counts8, row));
  if (!(lazy & !pop & !due)) return true;
  // IDLE ROW: nothing moves, and (almost) nothing is written. With agents, the row keeps its head and count: its word
  // is already what a refresh would store. Empty, its garbage head departs at t + tt0, which changes every frame — but
  // nobody reads the departure of an empty row that was idle (tail word without TLF_AUTH): this pass recomputes it, the
  // Direction gather's tests need an agent in the row or MAX_NUMBER_OF_AGENT <= 3 (rows that small keep the eager word),
  // export and delta_travel_time derive it from the clock. The tail word changes only when its flag has to go.
  if (n0i == 0u && nr.maxn <= TARL_CONGESTION_FILE)
    at32<O32>(fb.hdp, row) = make_uint2((head_id << 8) | ni | (hp.x & HD_DIRTY), __float_as_uint(head_dep));
  if (tlw & TLF_AUTH) {
    at32<O32>(fb.tl, row) = tlw & ~TLF_AUTH;      // tail and ring offset stay
    // ... and the post word's mirror of the flag goes with it (other workgroups may be gathering this word for their
    // Response test right now: they look at PF_NONEMPTY and the tail id, which do not change)
    at32<O32>(fb.post, row) = pa & ~PF_TLAUTH;
  }
  if (FAPI && out.countsf) __builtin_nontemporal_store((float)ni, &out.countsf[row]);
  if (FAPI && out.popped) out.popped[(int64_t)b * N + i] = 0;
  if (FAPI && out.withdrawn) out.withdrawn[(int64_t)b * N + i] = 0;
  if (out.events && b < out.m_env) out.events[(int64_t)i * out.m_env + b] = 0;
  return false;
}

// phase B of one EVENT row (its dense words travel with the list entry): Direction update on the slot store,
// Response pop, withdraw, refreshed dense words + event word. -> {count after the pass, agents withdrawn}
// Round 4, measured and rejected on this function and its caller (same-box A/B at 16 384 environments, profiles/README.md):
// (1) every load whose address follows from the list entry requested at the head of the function — the record a pop
// exposes, the destination of the scan's first candidate, the withdraw test's out-edge targets from the node record — for
// every event row: row pass +7 % / +15 % (headline / loaded network), 7 M more scattered requests per launch; the same only
// for heads that are due and without re-reading the record the scan stopped at: +2 % / +1 %, 14 spilled VGPRs in the hot
// idle path. (2) DENSE rows: for a wave-row with >= 4 / 8 / 16 event lanes the refreshed words handed back through the
// list and stored by the owning lanes as FULL 64-byte lines after a second barrier (re-read from L2, flag clears of the idle
// lanes folded in): loaded network +14 % / +15 % / +2 % slower, and the extra LDS table and live state put scratch traffic
// into the idle path (headline 181 -> 312 us). (3) 32-byte slot records (full-sector stores): fused_common.h, TARL_SLW.
// What did pay: one store fewer per event row (the 8-byte event word rec1 became the byte gc8: -10 % / -17 %).
template <bool FAPI>
__device__ __forceinline__ float2 row_phase_b(uint32_t i, uint32_t b, bool pop, uint32_t pa, uint2 hp, uint32_t tlw,
                                              const NodeRec& nr, const int32_t* __restrict__ out_ptr,
                                              const int32_t* __restrict__ out_dst, int Nmax, uint32_t B, uint32_t N,
                                              const FusedBufs& fb, float* __restrict__ ag, int64_t A, int64_t a_bstride,
                                              float t, const FrameOut& out) {
  const PlanOut P{out_ptr, out_dst};
  const uint32_t row = i * B + b;
  const float4 st = make_float4(nr.maxn, nr.ff, nr.road, nr.cong);
  const uint32_t n0i = hp.x & HD_CNT, head_id0 = hp.x >> 8, tail0 = tlw >> 8;
  const uint32_t arrived = pa & PF_ARRIVED;
  const uint32_t who = arrived ? (pa >> 8) : 0u;
  const float n0 = (float)n0i;
  const int q = (int)n0i;
  // exact, slot-by-slot bookkeeping of the dead slots: rows that are dirty already, and from the moment the FIFO touches
  // its last slot (count >= Nmax - 1: no dead slot is left above the one this update writes, so nothing has to be
  // materialised at the transition). A clean row's dead slots are zero by the invariant (fused_common.h) whatever the
  // store holds: its pops and withdraws neither read nor write them.
  const bool exact = (hp.x & HD_DIRTY) != 0u || q >= Nmax - 1;
  // Direction update (every row, also when nothing was chosen): one 12-byte store — or, for a row that received
  // nobody, nothing at all (lazy garbage slot, see the file header).
  const float dep_new = t + entry_tt(st, n0);
  const bool lazy = (who == 0u) && (q < Nmax - 1);
  const uint32_t ni = n0i + arrived;   // count after the Direction update
  uint32_t head_id = (n0i == 0u) ? who : head_id0;
  float head_dep = (n0i == 0u) ? dep_new : __uint_as_float(hp.y);
  {
    // The FIFO is a ring buffer: logical slot s lives at physical slot (hoff + s) mod Nmax.
    float* sl = fb.slots + (int64_t)row * fb.lds;  // slot s = sl[3s .. 3s+2] = {id, arrival, departure}
    int hoff = tl_hoff(tlw);     // (the event byte gc8 is write-only here: no load sits between the row and its slots)
    if (!lazy && q < Nmax) {
      slot_store(sl + SLW * phys(hoff, q, Nmax), (float)who, t, dep_new);
    }
    int n = (int)ni;
    uint32_t tail_id = arrived ? who : tail0;

    // Response pop: logical shift by one where the LAST slot keeps its value. Ring form: the slot that falls off the
    // front becomes the new logical last slot, so it receives a copy of the old last slot; then the head advances.
    int shift = 0;
    if (pop) {
      if (exact) {
        const SlotRec last = slot_load(sl + SLW * phys(hoff, Nmax - 1, Nmax));
        slot_store(sl + SLW * hoff, last.id, last.arr, last.dep);
      }
      hoff = phys(hoff, 1, Nmax);
      shift = 1;
      n = n - 1;
    }
    // withdraw: leading run of the (popped) row
    int c = 0;
    if (n > 0) {
      const long long road = (long long)st.z;
      int32_t w0 = -1, w1 = 0;   // out-list of this row's road: fetched only when a head is actually due
      for (int sx = 0; sx < Nmax && sx < n; ++sx) {
        float idf, depf;
        if (sx == 0 && shift == 0) {   // the head is in registers unless the pop just exposed a new one
          idf = (float)head_id;
          depf = head_dep;
        } else {
          const SlotRec rd = slot_load(sl + SLW * phys(hoff, sx, Nmax));
          idf = rd.id;
          depf = rd.dep;
        }
        const long long id = (long long)idf;
        if (id < 0 || id >= A) break;
        if (!(depf <= t)) break;  // tested first: most heads are still travelling, and the lookup below is a gather
        const long long dest = (long long)fb.a_dest[(int64_t)b * A + id];
        if (w0 < 0) {
          w0 = 0;
          if (road >= 0 && road < N) {
            w0 = P.out_ptr[road];
            w1 = P.out_ptr[road + 1];
          }
        }
        bool conn = false;
        for (int32_t k = w0; k < w1; ++k) conn = conn || ((long long)P.out_dst[k] == dest);
        if (!conn) break;
        float* a = ag + (int64_t)b * a_bstride + id * AG_COLS;
        a[AG_DONE] = 1.0f;
        a[AG_ON_WAY] = 0.0f;
        a[AG_ARR] = t;
        fb.a_status[(int64_t)b * A + id] = 2;
        ++c;
      }
    }
    // withdraw = logical shift by c with zero fill: the c slots that fall off the front become the zeroed tail
    for (int k = 0; exact && k < c; ++k) {
      slot_store(sl + SLW * phys(hoff, k, Nmax), 0.0f, 0.0f, 0.0f);
    }
    if (c > 0) {
      hoff = phys(hoff, c, Nmax);   // c <= Nmax
      n = n - c;
    }
    if (shift + c > 0) {
      if (lazy && n == 0) {  // the row emptied: its head slot is the (unmaterialised) garbage slot
        head_id = 0u;
        head_dep = dep_new;
      } else if (!exact && n == 0) {   // a clean row emptied by the pop of the agent it has just received: a dead slot, zero
        head_id = 0u;
        head_dep = 0.0f;
      } else {
        const SlotRec hd = slot_load(sl + SLW * hoff);
        head_id = (uint32_t)(long long)hd.id;
        head_dep = hd.dep;
      }
      // the agents that stay keep their order: the tail is who it was (the arrival, or the tail word's id) unless nobody
      // stays. (A count at Nmax is outside the domain and flagged; the store is re-read there as the reference would.)
      if (n == 0)
        tail_id = 0u;
      else if (n >= Nmax)
        tail_id = (n == Nmax) ? (uint32_t)(long long)sl[SLW * phys(hoff, n - 1, Nmax)] : 0u;
    }
    fb.hdp[row] = make_uint2((head_id << 8) | (uint32_t)n | (exact ? HD_DIRTY : 0u), __float_as_uint(head_dep));
    fb.tl[row] = tl_word(tail_id, hoff, TLF_AUTH);
    if (out.write_gc || b < (uint32_t)out.m_env) fb.gc8[row] = (uint8_t)r1_code(lazy ? q : -1);
    // per-node count before insertion (the insert kernel adds this frame's arrivals)
    if (out.counts8 && c > 0) out.counts8[row] = (uint8_t)n;      // (phase A stored ni - pop for this row already)
    if (FAPI && out.countsf) out.countsf[row] = (float)n;
    if (FAPI && out.popped) out.popped[(int64_t)b * N + i] = pop ? 1 : 0;
    if (FAPI && out.withdrawn) out.withdrawn[(int64_t)b * N + i] = c > 0 ? 1 : 0;
    if (out.events && b < out.m_env) out.events[(int64_t)i * out.m_env + b] = (uint8_t)((pop ? 1 : 0) | (c > 0 ? 2 : 0));
    return make_float2((float)n, (float)c);
  }
}

// NCH rows per lane: ALL their loads (dense words, downstream post words) are issued before the first row is processed,
// so a wave keeps 8 * NCH independent requests in flight instead of one row's dependent phases.
// SIB (NCH == 4): the chunk's rows come from the plan's row-chunk table — rows with the same ordered out-edge target list,
// on a road network the roads that ENTER one intersection — so their four downstream post words are gathered ONCE per
// chunk instead of once per row (those gathers were a quarter of the pass's reads once a frame's words no longer fit the
// Infinity Cache: 20 -> 8 post requests per lane and chunk; 923 -> 691 MB read per launch at 16 384 environments, same
// time: the pass is not bound by its reads). Same rows, same arithmetic, another visiting order.
// Measured and rejected in round 3: the event path as a launch of its own over lists in global memory
// (158 + 94 us instead of 236: its stores then hit lines that have left the L2, and a partial-line store to a cold line is a
// read-modify-write at the memory side, profiles/r03_pmc_calibration.txt); a list for every pair and no in-place fall-back
// (no spilled register left, 18 KB of LDS: 250 us); the event path's pointers and statics through a block in LDS with
// further list rounds instead of the in-place fall-back (no spilled SGPR, 7 KB: 275 us); the pop's two slot reads requested
// before anything is stored (same time).
struct __attribute__((aligned(32))) RowChunk {
  int32_t row[4];    // -1: none (a group's remainder)
  int32_t out4[4];   // the shared first four out-edge targets
};
template <int NCH, bool SIB, bool FAPI, bool O32>
__global__ __launch_bounds__(TILE) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fused_rows(const NodeRec* __restrict__ nodes,
                                                     const int32_t* __restrict__ out_pad,
                                                     const int32_t* __restrict__ out_ptr,
                                                     const int32_t* __restrict__ out_dst,
                                                     const RowChunk* __restrict__ rchunks,
                                                     const uint32_t* __restrict__ post, int Nmax, uint32_t B, uint32_t N,
                                                     const FusedBufs* __restrict__ fbp, float* __restrict__ ag, int64_t A,
                                                     int64_t a_bstride, float t, FrameOut out) {
  // The table of pointers lives in device memory (tarl_fused.bufs_dev, written by k_set_bufs at the head of the call): what
  // the pass needs of it is read where it is needed, through the scalar cache, instead of travelling as 27 kernel arguments
  // that the register allocator keeps alive across the whole pass (round 5, static figures of tools/kernel_resources.sh for
  // this instantiation: 45 -> 13 spilled SGPRs, 173 -> 48 v_readlane / v_writelane, 1 135 -> 1 029 vector instructions).
  const FusedBufs& fb = *fbp;
  static_assert(!SIB || NCH == 4, "row chunks hold four rows");
  __shared__ int32_t s_cnt;
  // the event list holds EV_CAP of the TILE * NCH pairs (a filling network lists ~2 %, a loaded one ~30 %); a pair that
  // finds it full runs its event path in place. 7 KB instead of 18: the workgroups a CU holds are bounded by its wave
  // slots, not by LDS, and the waves that have no list entry leave early
#ifndef TARL_EV_CAP
#define TARL_EV_CAP 384      // (developer override through `make variant EXPFLAGS=-DTARL_EV_CAP=...`: 512 and 768 measure the same)
#endif
  constexpr int EV_CAP = TARL_EV_CAP;
  __shared__ uint16_t s_item[EV_CAP];       // (row offset in the chunk) << 9 | pop << 8 | lane
  __shared__ uint4 s_words[EV_CAP];         // the listed row's {post word, hd, head_dep bits, tl}
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = b < B;
  // the chunk's rows (wave-uniform): consecutive, or the table's
  int32_t ri[NCH];
  bool live[NCH];
  if (SIB) {
    const RowChunk& rc = rchunks[blockIdx.y];
#pragma unroll
    for (int r = 0; r < NCH; ++r) {
      live[r] = rc.row[r] >= 0;
      ri[r] = live[r] ? rc.row[r] : rc.row[0];   // a missing row is loaded as the first one and never used
    }
  } else {
    const uint32_t i0 = blockIdx.y * NCH;
#pragma unroll
    for (int r = 0; r < NCH; ++r) {
      live[r] = i0 + r < N;
      ri[r] = live[r] ? (int32_t)(i0 + r) : (int32_t)(N - 1);   // clamped: the tail rows are loaded twice, used once
    }
  }
  RowStat rs[NCH];
#pragma unroll
  for (int r = 0; r < NCH; ++r) {
    const NodeRec& nr = nodes[ri[r]];
    rs[r] = RowStat{nr.out_deg, nr.out0, nr.tt0, nr.maxn};
  }
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  float nsum = 0.0f;
  if (valid) {
    uint32_t pa[NCH], tlw[NCH], pj[NCH][4];
    uint2 hp[NCH];
    uint32_t ovf = 0u;       // bit r: row r found the event list full; bit 4 + r: its pop flag
#pragma unroll
    for (int r = 0; r < NCH; ++r) {
      const uint32_t row = (uint32_t)ri[r] * B + b;
      pa[r] = at32<O32>(post, row);
      if (SIB) {
        if (r == 0) {
          const int32_t* od = rchunks[blockIdx.y].out4;
#pragma unroll
          for (int q = 0; q < 4; ++q) pj[0][q] = at32<O32>(post, (uint32_t)od[q] * B + b);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) pj[r][q] = pj[0][q];
        }
      } else {
        const int32_t* od = nodes[ri[r]].out4;   // the first four targets travel in the node record
#pragma unroll
        for (int q = 0; q < 4; ++q) pj[r][q] = at32<O32>(post, (uint32_t)od[q] * B + b);
      }
    }
    // The post word already says whether the row holds anybody (PF_NONEMPTY), receives somebody (PF_ARRIVED) or still
    // carries the flag of an event in its tail word (PF_TLAUTH). A row with none of the three is empty and idle: its
    // head words are {0, unused} and its tail word is flag-free whatever its stale tail id — they are not fetched. In a
    // filling network that is most rows; the second round of loads only touches the sectors of the others.
#pragma unroll
    for (int r = 0; r < NCH; ++r) {
      const uint32_t row = (uint32_t)ri[r] * B + b;
      hp[r] = make_uint2(0u, 0u);
      tlw[r] = 0u;
      if (Nmax < 2 || (pa[r] & (PF_ARRIVED | PF_NONEMPTY | PF_TLAUTH))) {   // (a one-slot FIFO has no lazy garbage slot)
        hp[r] = at32<O32>(fb.hdp, row);
        tlw[r] = at32<O32>(fb.tl, row);
      }
    }
#pragma unroll
    for (int r = 0; r < NCH; ++r)
      if (live[r]) {
        bool pop;
        float n = 0.0f;
        const uint32_t i = (uint32_t)ri[r];
        const bool ev = row_phase_a<FAPI, O32>(i, b, rs[r], post, pa[r], hp[r], tlw[r], pj[r], Nmax, B, N, fb, A, t, out, &pop, &n);
        nsum += n;
        if (ev) {
          const int32_t pos = atomicAdd(&s_cnt, 1);
          if (pos < EV_CAP) {
            s_item[pos] = (uint16_t)((r << 9) | (pop ? 256 : 0) | threadIdx.x);
            s_words[pos] = make_uint4(pa[r], hp[r].x, hp[r].y, tlw[r]);
          } else {
            ovf |= (1u << r) | (pop ? (16u << r) : 0u);   // the list is full: served in place below
          }
        }
      }
    // In-place fall-back for the pairs that found the list full (a loaded network; never in a filling one). ONE rolled copy
    // of the event path behind the four rows, not one inside each row's body: with four inlined copies the idle path
    // carried their registers and 145 spilled SGPRs (row pass 222 -> 208 us without them, timing-only build).
    if (ovf != 0u) {
#pragma unroll 1
      for (int r = 0; r < NCH; ++r) {
        if (!((ovf >> r) & 1u)) continue;
        uint32_t pa_r = pa[0], tl_r = tlw[0];
        uint2 hp_r = hp[0];
        int32_t i_r = ri[0];
#pragma unroll
        for (int q = 1; q < NCH; ++q) {
          pa_r = (r == q) ? pa[q] : pa_r;
          tl_r = (r == q) ? tlw[q] : tl_r;
          hp_r.x = (r == q) ? hp[q].x : hp_r.x;
          hp_r.y = (r == q) ? hp[q].y : hp_r.y;
          i_r = (r == q) ? ri[q] : i_r;
        }
        const float2 nc = row_phase_b<FAPI>((uint32_t)i_r, b, ((ovf >> (4 + r)) & 1u) != 0u, pa_r, hp_r, tl_r, nodes[i_r], out_ptr,
                                      out_dst, Nmax, B, N, fb, ag, A, a_bstride, t, out);
        nsum -= nc.y;
        if (nc.y != 0.0f) atomicAdd(&fb.acc_w[(int64_t)(blockIdx.y % (unsigned)fb.acc_slots) * B + b], nc.y);
      }
    }
  }
  // idle rows' share of the environment's count sum goes out now: a wave without a list entry is done after the barrier
  // (its slots go to the next workgroup while the event path, a chain of dependent loads a few lanes wide, runs on)
  const int64_t bank0 = (int64_t)(blockIdx.y % (unsigned)fb.acc_slots) * B;
  if (valid && nsum != 0.0f) atomicAdd(&fb.acc_n[bank0 + b], nsum);
  __syncthreads();
  const int32_t cnt = s_cnt < EV_CAP ? s_cnt : EV_CAP;
  for (int32_t idx = threadIdx.x; idx < cnt; idx += blockDim.x) {
    const uint32_t item = s_item[idx];
    const uint32_t r = item >> 9, lane2 = item & 255u;
    const uint4 wd = s_words[idx];
    const uint32_t b2 = blockIdx.x * blockDim.x + lane2;
    int32_t i2 = ri[0];
#pragma unroll
    for (int q = 1; q < NCH; ++q) i2 = (r == (uint32_t)q) ? ri[q] : i2;
    const float2 nc = row_phase_b<FAPI>((uint32_t)i2, b2, (item & 256u) != 0u, wd.x, make_uint2(wd.y, wd.z), wd.w, nodes[i2],
                                  out_ptr, out_dst, Nmax, B, N, fb, ag, A, a_bstride, t, out);
    if (nc.y != 0.0f) {      // withdrawn agents leave the count sum (small integers: exact in fp32 in any order)
      atomicAdd(&fb.acc_n[bank0 + b2], -nc.y);
      atomicAdd(&fb.acc_w[bank0 + b2], nc.y);
    }
  }
}

// ---- insert + reward + log-prob reduction (one wave per environment) -----------------------------------------------------
__device__ __forceinline__ bool fused_target(const FusedBufs& fb, PlanOut P, const uint8_t* __restrict__ sel8, int64_t b,
                                             int64_t B, int64_t N, int32_t origin, int32_t* road, int32_t* cap) {
  if (origin < 0 || origin >= N) return false;
  const int64_t orow = (int64_t)origin * B + b;
  const uint32_t c = sel8[orow] & 0x7Fu;
  const long long r = (long long)(c == SEL_RAW ? fb.sel[orow] : (float)P.out_dst[P.out_ptr[origin] + (int32_t)c]);
  if (r < 0 || r >= N) return false;
  const long long room = (long long)(fb.st0[r].x - TARL_CONGESTION_FILE - (float)(fb.hdp[r * B + b].x & HD_CNT));
  *road = (int32_t)r;
  *cap = (int32_t)(room > 0x7fffffff ? 0x7fffffff : room);
  return room > 0;
}

// The same test with the words the admission phase needs afterwards: the target row's count word and tail word come back
// with it (they are stashed beside the candidate, so that phase reads nothing dynamic again), and a rank below four
// resolves through the origin's node record (one round: the record and the SELECTED_ROAD byte travel together)
// instead of out_ptr -> out_dst (two).
__device__ __forceinline__ bool fused_target_words(const FusedBufs& fb, PlanOut P, const uint8_t* __restrict__ sel8,
                                                   int64_t b, int64_t B, int64_t N, int32_t origin, int32_t* road,
                                                   uint32_t* hd_out, uint32_t* tl_out) {
  if (origin < 0 || origin >= N) return false;
  const int64_t orow = (int64_t)origin * B + b;
  const uint32_t c = sel8[orow] & 0x7Fu;
  const int32_t* o4 = fb.nodes[origin].out4;
  const int32_t t0 = o4[0], t1 = o4[1], t2 = o4[2], t3 = o4[3];
  long long r;
  if (c < 4u)
    r = (long long)(float)(c == 0u ? t0 : (c == 1u ? t1 : (c == 2u ? t2 : t3)));
  else
    r = (long long)(c == SEL_RAW ? fb.sel[orow] : (float)P.out_dst[P.out_ptr[origin] + (int32_t)c]);
  if (r < 0 || r >= N) return false;
  const uint32_t hd = fb.hdp[r * B + b].x;
  *tl_out = fb.tl[r * B + b];
  const long long room = (long long)(fb.st0[r].x - TARL_CONGESTION_FILE - (float)(hd & HD_CNT));
  *road = (int32_t)r;
  *hd_out = hd;
  return room > 0;
}

// The insert kernels' LDS block (one per workgroup; the one-environment body and the two-environment kernel share it, so
// that the latter's rare fall-back into the former costs no second block: 3.9 KB, 41 workgroups per CU)
struct InsLds {
  int32_t wave[INSB / 64];
  int32_t cnt, adm, lo;
  int32_t cnt2[8], adm2[8], lo2[8];   // k_fused_insert2: one set per environment of the wave
  int32_t un_agent[INS_CAP], un_road[INS_CAP], un_k[INS_CAP];
  uint32_t un_hd[INS_CAP], un_tl[INS_CAP];
};
#define s_wave L.wave
#define s_cnt L.cnt
#define s_adm L.adm
#define s_lo L.lo
#define s_un_agent L.un_agent
#define s_un_road L.un_road
#define s_un_k L.un_k
#define s_un_hd L.un_hd
#define s_un_tl L.un_tl
__device__ __forceinline__ void fused_insert_body(InsLds& L, int64_t b, int Nmax, int64_t B, int64_t N, const FusedBufs& fb, PlanOut P,
                                                  const uint8_t* __restrict__ sel8, float* __restrict__ ag, int64_t A,
                                                  int64_t a_bstride, int use_cong, float t,
                                                  int32_t* __restrict__ scratch, const float* __restrict__ entropy_in,
                                                  float* __restrict__ reward, FrameOut out,
                                                  float* __restrict__ log_prob, float* __restrict__ entropy) {
  // (window path: un_k = the candidate's position in the departure order, un_hd / un_tl = its target row's count / tail words)
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
  // the frame's accumulator banks (filled by the choice kernel and the row pass, complete before this launch) depend on
  // nothing below: they are requested first and reduced last
  long long lpf = 0;
  float nf = 0.0f, wf = 0.0f;
  if (wid == 0) {
    for (int64_t sl_ = lane; sl_ < fb.acc_slots; sl_ += 64) {
      lpf += fb.acc_lp[sl_ * B + b];
      nf += fb.acc_n[sl_ * B + b];
      wf += fb.acc_w[sl_ * B + b];
      fb.acc_lp[sl_ * B + b] = 0;
      fb.acc_n[sl_ * B + b] = 0.0f;
      fb.acc_w[sl_ * B + b] = 0.0f;
    }
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
        bool due, waiting;
        int32_t a, origin;
        if (fb.a_win) {     // one pair of independent loads: {departure, origin, agent} record + "already inserted" byte
          const uint4 w = fb.a_win[b * A + k];
          waiting = fb.a_ins[b * A + k] == 0;
          due = __uint_as_float(w.x) <= t;
          origin = (int32_t)w.y;
          a = (int32_t)w.z;
        } else {            // sequential departures; per-agent arrays only for entries that are due
          due = dsort[k] <= t;
          a = due ? ord[k] : 0;
          waiting = due && fb.a_status[b * A + a] == 0;
          origin = waiting ? fb.a_origin[b * A + a] : 0;
        }
        notdue = !due;
        if (!due) {
          atomicMin(&s_lo, (int32_t)k);          // the cursor may not pass this entry
        } else if (waiting) {
          atomicMin(&s_lo, (int32_t)k);
          int32_t road = 0, cap = 0;
          if (fb.a_win) {
            uint32_t hdw = 0u, tlw = 0u;
            if (fused_target_words(fb, P, sel8, b, B, N, origin, &road, &hdw, &tlw)) {
              const int32_t pos = atomicAdd(&s_cnt, 1);
              if (pos < INS_CAP) {
                s_un_agent[pos] = a;
                s_un_road[pos] = road;
                s_un_k[pos] = (int32_t)k;
                s_un_hd[pos] = hdw;
                s_un_tl[pos] = tlw;
              }
            }
          } else if (fused_target(fb, P, sel8, b, B, N, origin, &road, &cap)) {
            const int32_t pos = atomicAdd(&s_cnt, 1);
            if (pos < INS_CAP) {
              s_un_agent[pos] = a;
              s_un_road[pos] = road;
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
          if (fused_target(fb, P, sel8, b, B, N, fb.a_origin[b * A + a], &road, &cap)) {
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
  const bool stashed = fb.a_order && fb.a_win && Lc <= INS_CAP;
  if (stashed) {
    // Admission straight from the LDS list: a candidate's rank on its road is the number of candidates for the same road
    // with a smaller agent id (the reference admits in stable agent-id order), so the list needs no sorting, and the
    // target row's words were stashed with it: nothing dynamic is read here. The rank-0 candidate of a road owns its
    // count word (the head too when the road was empty), event word and count outputs; the last admitted owns the tail.
    for (int32_t idx = tid; idx < Lc; idx += INSB) {
      const int32_t r = s_un_road[idx];
      const int32_t a = s_un_agent[idx];
      int32_t rank = 0, total = 0;
      for (int32_t k = 0; k < Lc; ++k) {
        const bool same = s_un_road[k] == r;
        total += same ? 1 : 0;
        rank += (same && s_un_agent[k] < a) ? 1 : 0;
      }
      const int64_t rrow = (int64_t)r * B + b;
      const float4 str = fb.st0[r];
      const uint32_t hd = s_un_hd[idx];
      const uint32_t n0i = hd & HD_CNT;
      const float n0 = (float)n0i;
      const long long cap = (long long)(str.x - TARL_CONGESTION_FILE - n0);
      if (rank < cap) {
        const long long m = total < cap ? total : cap;  // arrivals admitted on this road
        const long long slot = (long long)n0i + rank;
        const float t_cong = use_cong ? str.w / (str.x + 10.0f - n0) : 0.0f;
        const float tt = (t_cong != t_cong) ? t_cong : fmaxf(str.y, t_cong);
        const int hoff = tl_hoff(s_un_tl[idx]);
        if (slot >= 0 && slot < Nmax) {
          slot_store(fb.slots + rrow * fb.lds + SLW * phys(hoff, (int)slot, Nmax), (float)a, t, t + tt);
        }
        agb[(int64_t)a * AG_COLS + AG_ON_WAY] = 1.0f;
        fb.a_status[b * A + a] = 1;
        fb.a_ins[b * A + s_un_k[idx]] = 1;
        if (rank == m - 1) fb.tl[rrow] = tl_word((uint32_t)a, hoff, TLF_AUTH);  // new tail; gc8 authoritative from here on
        if (rank == 0) {
          const uint32_t cnt = n0i + (uint32_t)m;      // n0 + m <= MAX - 3 < 255
          if (n0i == 0u) {   // new head: id + departure, arrival
            fb.hdp[rrow] = make_uint2(((uint32_t)a << 8) | cnt | (hd & HD_DIRTY), __float_as_uint(t + tt));
          } else {
            fb.hdp[rrow].x = hd + (uint32_t)m;
          }
          if (out.write_gc || b < out.m_env) fb.gc8[rrow] = (uint8_t)r1_code(-1);   // the arrivals overwrote a pending garbage slot: none pending now
          if (out.counts8) out.counts8[rrow] = (uint8_t)cnt;
          if (out.countsf) out.countsf[rrow] = (float)cnt;
          atomicAdd(&s_adm, (int32_t)m);
        }
      }
    }
    __syncthreads();
  } else {
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
        cnd = fused_target(fb, P, sel8, b, B, N, fb.a_origin[b * A + a], &road, &cap);
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

  // phase 2: rank within road (stable), admit the first min(count, capacity), write slots / dense words.
  // Per admitted road: the rank-0 candidate owns hdp / gc8 (the count's byte of hd is committed after the barrier), the
  // last admitted candidate owns tl. Everybody else only READS the count byte, which nobody changes in this phase.
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
    const uint32_t hd = fb.hdp[rrow].x;
    const uint32_t n0i = hd & HD_CNT;
    const float n0 = (float)n0i;
    const long long cap = (long long)(str.x - TARL_CONGESTION_FILE - n0);
    int32_t commit = 0;
    if (rank < cap) {
      const long long m = total < cap ? total : cap;  // arrivals admitted on this road
      const long long slot = (long long)n0i + rank;
      const float t_cong = use_cong ? str.w / (str.x + 10.0f - n0) : 0.0f;
      const float tt = (t_cong != t_cong) ? t_cong : fmaxf(str.y, t_cong);
      const int hoff = tl_hoff(fb.tl[rrow]);   // the tail word's owner below rewrites it with the same offset
      if (slot >= 0 && slot < Nmax) {
        slot_store(fb.slots + rrow * fb.lds + SLW * phys(hoff, (int)slot, Nmax), (float)a, t, t + tt);
      }
      agb[(int64_t)a * AG_COLS + AG_ON_WAY] = 1.0f;
      fb.a_status[b * A + a] = 1;
      if (fb.a_ins) fb.a_ins[b * A + fb.a_rank[b * A + a]] = 1;
      if (rank == 0 && n0i == 0u) {   // new head: id + departure (count byte unchanged), arrival
        fb.hdp[rrow] = make_uint2(((uint32_t)a << 8) | n0i | (hd & HD_DIRTY), __float_as_uint(t + tt));
      }
      if (rank == m - 1) fb.tl[rrow] = tl_word((uint32_t)a, hoff, TLF_AUTH);  // new tail; gc8 authoritative from here on
      if (rank == 0) commit = (int32_t)m;
    }
    cand_agent[idx] = commit;
  }
  __threadfence_block();
  __syncthreads();
  // phase 3: commit the counters; the arrivals overwrote a pending garbage slot: no garbage pending, keep the head offset
  for (int32_t idx = tid; idx < Lc; idx += INSB) {
    const int32_t cmt = cand_agent[idx];
    if (cmt > 0) {
      const int64_t rrow = (int64_t)cand_road[idx] * B + b;
      if (out.write_gc || b < out.m_env) fb.gc8[rrow] = (uint8_t)r1_code(-1);
      const uint32_t hd = fb.hdp[rrow].x + (uint32_t)cmt;   // count byte: n0 + cmt <= MAX - 3 < 255
      fb.hdp[rrow].x = hd;
      if (out.counts8) out.counts8[rrow] = (uint8_t)(hd & HD_CNT);
      if (out.countsf) out.countsf[rrow] = (float)(hd & HD_CNT);
      atomicAdd(&s_adm, cmt);
    }
  }
  __syncthreads();
  }
  // phase 4: the frame's accumulator banks (requested at the top) -> reward, log-prob
  if (wid == 0) {
    for (int off = 32; off > 0; off >>= 1) {
      lpf += __shfl_down(lpf, off);
      nf += __shfl_down(nf, off);      // sums of small integers: exact in fp32 in any order
      wf += __shfl_down(wf, off);
    }
    if (lane == 0) {
      if (reward) reward[b] = -(nf + (float)s_adm);
      if (log_prob) log_prob[b] = (lpf < -(1ll << 49)) ? -INFINITY : (float)((double)lpf / LP_FIX);
      if (entropy) entropy[b] = entropy_in[0];
      if (out.leg) {
        out.leg[2 * b + 0] = s_adm;
        out.leg[2 * b + 1] = (int32_t)wf;
      }
    }
  }
}

#undef s_wave
#undef s_cnt
#undef s_adm
#undef s_lo
#undef s_un_agent
#undef s_un_road
#undef s_un_k
#undef s_un_hd
#undef s_un_tl

__global__ __launch_bounds__(INSB) void k_fused_insert(int Nmax, int64_t B, int64_t N, const FusedBufs* __restrict__ fbp, PlanOut P,
                                                       const uint8_t* __restrict__ sel8, float* __restrict__ ag,
                                                       int64_t A, int64_t a_bstride, int use_cong, float t,
                                                       int32_t* __restrict__ scratch,
                                                       const float* __restrict__ entropy_in,
                                                       float* __restrict__ reward, FrameOut out,
                                                       float* __restrict__ log_prob, float* __restrict__ entropy) {
  __shared__ InsLds L;
  fused_insert_body(L, blockIdx.x, Nmax, B, N, *fbp, P, sel8, ag, A, a_bstride, use_cong, t, scratch, entropy_in, reward,
                    out, log_prob, entropy);
}

// SEVERAL environments per wave (EPW = 2, 4 or 8: 32, 16 or 8 lanes each; chosen by the population size). With one wave per environment a
// launch of more than 8 192 environments does not fit the chip's wave slots and runs in two rounds of a latency chain —
// and below that, fewer waves walk the same chain faster (18 instead of 23 us at 4 096 environments). A frame's window
// chunk is a handful of entries and the accumulator banks are 32 wide, so a fraction of a wave does an environment's
// work as fast. Same phases, same order, same results as fused_insert_body on its departure-window path (a_order +
// a_win): scan from the cursor, candidates unordered in LDS (each environment its share of the block), admission by
// rank among the candidates of the same road. An environment with more candidates than its share of the list holds
// sits the packed part out and is handed, afterwards, to the one-environment body on the whole wave (which rescans from
// the cursor this kernel has already advanced — the same cursor it would have computed — and reduces the accumulator
// banks itself: nothing else of that environment has been written by then). 16 384 environments: 65 us with one wave
// each, 50 / 40 / 36 us with 2 / 4 / 8 per wave.
template <int EPW>
__global__ __launch_bounds__(INSB) void k_fused_insert2(int Nmax, int64_t B, int64_t N, const FusedBufs* __restrict__ fbp, PlanOut P,
                                                        const uint8_t* __restrict__ sel8, float* __restrict__ ag,
                                                        int64_t A, int64_t a_bstride, int use_cong, float t,
                                                        int32_t* __restrict__ scratch,
                                                        const float* __restrict__ entropy_in,
                                                        float* __restrict__ reward, FrameOut out,
                                                        float* __restrict__ log_prob, float* __restrict__ entropy) {
  __shared__ InsLds L;
  const FusedBufs& fb = *fbp;              // the pointer table in device memory (k_set_bufs): 88 -> 38 spilled SGPRs at EPW = 8
  constexpr int LPE = 64 / EPW;            // lanes per environment
  constexpr int INS_CAP2 = INS_CAP / EPW;  // its share of the candidate list
  const int tid = threadIdx.x, h = tid / LPE, l = tid % LPE;
  const int64_t b0 = EPW * (int64_t)blockIdx.x, b = b0 + h;
  const bool live = b < B;
  const int base = h * INS_CAP2;
  if (l == 0) {
    L.cnt2[h] = 0;
    L.adm2[h] = 0;
    L.lo2[h] = 0x7fffffff;
  }
  // the frame's accumulator banks: requested first, consumed (and re-armed) only once the pair is known to stay here
  long long lpf = 0;
  float nf = 0.0f, wf = 0.0f;
  if (live) {
    for (int64_t sl_ = l; sl_ < fb.acc_slots; sl_ += LPE) {
      lpf += fb.acc_lp[sl_ * B + b];
      nf += fb.acc_n[sl_ * B + b];
      wf += fb.acc_w[sl_ * B + b];
    }
  }
  const int32_t lo = live ? fb.cur_lo[b] : 0;
  __syncthreads();
  // phase 1: the departure window, LPE entries per environment and step
  bool done = !live;
  int64_t k0 = lo;
  if (live && k0 >= A) done = true;
  while (true) {
    bool notdue = false;
    if (!done) {
      const int64_t k = k0 + l;
      if (k < A) {
        const uint4 w = fb.a_win[b * A + k];
        const bool waiting = fb.a_ins[b * A + k] == 0;
        const bool due = __uint_as_float(w.x) <= t;
        const int32_t origin = (int32_t)w.y, a = (int32_t)w.z;
        notdue = !due;
        if (!due) {
          atomicMin(&L.lo2[h], (int32_t)k);          // the cursor may not pass this entry
        } else if (waiting) {
          atomicMin(&L.lo2[h], (int32_t)k);
          int32_t road = 0;
          uint32_t hdw = 0u, tlw = 0u;
          if (fused_target_words(fb, P, sel8, b, B, N, origin, &road, &hdw, &tlw)) {
            const int32_t pos = atomicAdd(&L.cnt2[h], 1);
            if (pos < INS_CAP2) {
              L.un_agent[base + pos] = a;
              L.un_road[base + pos] = road;
              L.un_k[base + pos] = (int32_t)k;
              L.un_hd[base + pos] = hdw;
              L.un_tl[base + pos] = tlw;
            }
          }
        }
      }
    }
    const unsigned long long nd = __ballot(notdue);
    if (!done) {
      if (((nd >> (LPE * h)) & ((1ull << LPE) - 1ull)) != 0ull || k0 + LPE >= A)
        done = true;       // sorted by departure: nothing beyond this chunk is due
      else
        k0 += LPE;
    }
    if (__ballot(!done) == 0ull) break;
  }
  __syncthreads();
  // an environment with more candidates than its share of the list holds sits this part out (nothing of it is written
  // but the cursor) and goes through the one-environment body afterwards
  bool over = false;
#pragma unroll
  for (int e = 0; e < EPW; ++e) over = over || (L.cnt2[e] > INS_CAP2);
  if (live && l == 0) fb.cur_lo[b] = L.lo2[h] == 0x7fffffff ? (int32_t)A : L.lo2[h];
  const bool mine = live && L.cnt2[h] <= INS_CAP2;
  if (mine) {
    for (int64_t sl_ = l; sl_ < fb.acc_slots; sl_ += LPE) {   // the banks are consumed: re-arm them
      fb.acc_lp[sl_ * B + b] = 0;
      fb.acc_n[sl_ * B + b] = 0.0f;
      fb.acc_w[sl_ * B + b] = 0.0f;
    }
    // phase 2: admission straight from the list (see fused_insert_body): rank among the candidates of the same road
    float* agb = ag + b * a_bstride;
    const int32_t Lc = L.cnt2[h];
    for (int32_t idx = l; idx < Lc; idx += LPE) {
      const int32_t r = L.un_road[base + idx];
      const int32_t a = L.un_agent[base + idx];
      int32_t rank = 0, total = 0;
      for (int32_t k = 0; k < Lc; ++k) {
        const bool same = L.un_road[base + k] == r;
        total += same ? 1 : 0;
        rank += (same && L.un_agent[base + k] < a) ? 1 : 0;
      }
      const int64_t rrow = (int64_t)r * B + b;
      const float4 str = fb.st0[r];
      const uint32_t hd = L.un_hd[base + idx];
      const uint32_t n0i = hd & HD_CNT;
      const float n0 = (float)n0i;
      const long long cap = (long long)(str.x - TARL_CONGESTION_FILE - n0);
      if (rank < cap) {
        const long long m = total < cap ? total : cap;  // arrivals admitted on this road
        const long long slot = (long long)n0i + rank;
        const float t_cong = use_cong ? str.w / (str.x + 10.0f - n0) : 0.0f;
        const float tt = (t_cong != t_cong) ? t_cong : fmaxf(str.y, t_cong);
        const int hoff = tl_hoff(L.un_tl[base + idx]);
        if (slot >= 0 && slot < Nmax) {
          slot_store(fb.slots + rrow * fb.lds + SLW * phys(hoff, (int)slot, Nmax), (float)a, t, t + tt);
        }
        agb[(int64_t)a * AG_COLS + AG_ON_WAY] = 1.0f;
        fb.a_status[b * A + a] = 1;
        fb.a_ins[b * A + L.un_k[base + idx]] = 1;
        if (rank == m - 1) fb.tl[rrow] = tl_word((uint32_t)a, hoff, TLF_AUTH);  // new tail; gc8 authoritative from here on
        if (rank == 0) {
          const uint32_t cnt = n0i + (uint32_t)m;      // n0 + m <= MAX - 3 < 255
          if (n0i == 0u) {   // new head: id + departure, arrival
            fb.hdp[rrow] = make_uint2(((uint32_t)a << 8) | cnt | (hd & HD_DIRTY), __float_as_uint(t + tt));
          } else {
            fb.hdp[rrow].x = hd + (uint32_t)m;
          }
          if (out.write_gc || b < out.m_env) fb.gc8[rrow] = (uint8_t)r1_code(-1);   // the arrivals overwrote a pending garbage slot: none pending now
          if (out.counts8) out.counts8[rrow] = (uint8_t)cnt;
          if (out.countsf) out.countsf[rrow] = (float)cnt;
          atomicAdd(&L.adm2[h], (int32_t)m);
        }
      }
    }
  }
  __syncthreads();
  // phase 3: the accumulator banks -> reward, log-prob (LPE-lane reductions)
  for (int off = LPE / 2; off > 0; off >>= 1) {
    lpf += __shfl_down(lpf, off, LPE);
    nf += __shfl_down(nf, off, LPE);      // sums of small integers: exact in fp32 in any order
    wf += __shfl_down(wf, off, LPE);
  }
  if (mine && l == 0) {
    if (reward) reward[b] = -(nf + (float)L.adm2[h]);
    if (log_prob) log_prob[b] = (lpf < -(1ll << 49)) ? -INFINITY : (float)((double)lpf / LP_FIX);
    if (entropy) entropy[b] = entropy_in[0];
    if (out.leg) {
      out.leg[2 * b + 0] = L.adm2[h];
      out.leg[2 * b + 1] = (int32_t)wf;
    }
  }
  if (over) {   // wave-uniform, rare: the long backlogs, one environment after the other on the whole wave
    __syncthreads();
    for (int e = 0; e < EPW; ++e) {
      if (b0 + e < B && L.cnt2[e] > INS_CAP2)
        fused_insert_body(L, b0 + e, Nmax, B, N, fb, P, sel8, ag, A, a_bstride, use_cong, t, scratch, entropy_in, reward,
                          out, log_prob, entropy);
      __syncthreads();
    }
  }
}

// One launch, two roles (rollout steady state): the first B workgroups run frame t's insert (one wave each; the other
// three waves of such a workgroup retire at once, so its barriers only count the live wave), the remaining
// `choice_blocks` workgroups draw frame t+1's action into the NEXT slice of the action buffer and the other half of the
// double-buffered log-prob accumulators.
// The insert kernel is a latency chain that leaves the chip idle and the live policy's sample does not depend on the
// state, so the choice work rides in its shadow — without the cross-stream events that made a two-stream variant
// slower — and still completes right before the Direction kernel that consumes it (Infinity-Cache adjacency).
struct ChoiceArgs {
  const int32_t* out_ptr;
  const int32_t* out_eid;
  const int32_t* group_of_node;
  int64_t G;
  const float* thr;
  const long long* lgt;
  uint64_t pseed, pcounter;
  int nchunk, want_lp;
  uint8_t* sel_next;       // frame t+1's action / SELECTED_ROAD slice
  long long* acc_next;
  unsigned gx;             // environment tiles (x extent of the choice grid)
  unsigned choice_blocks;  // gx * node chunks
};
__global__ __launch_bounds__(TILE) void k_fused_insert_choice(ChoiceArgs C, int Nmax, int64_t B, int64_t N,
                                                              const FusedBufs* __restrict__ fbp,
                                                              PlanOut P, const uint8_t* __restrict__ sel8,
                                                              float* __restrict__ ag, int64_t A, int64_t a_bstride,
                                                              int use_cong, float t, int32_t* __restrict__ scratch,
                                                              const float* __restrict__ entropy_in,
                                                              float* __restrict__ reward, FrameOut out,
                                                              float* __restrict__ log_prob,
                                                              float* __restrict__ entropy) {
  // insert workgroups first: their dependent-load chains start at once and the choice workgroups fill the chip around them
  __shared__ InsLds L;
  if (blockIdx.x < (unsigned)B) {
    if (threadIdx.x >= INSB) return;   // whole waves leave before any barrier
    fused_insert_body(L, (int64_t)blockIdx.x, Nmax, B, N, *fbp, P, sel8, ag, A, a_bstride, use_cong, t, scratch, entropy_in,
                      reward, out, log_prob, entropy);
  } else {
    const unsigned cb = blockIdx.x - (unsigned)B;
    fused_choice_body(cb % C.gx, cb / C.gx, C.out_ptr, C.out_eid, C.group_of_node, C.G, B, N, C.acc_next, fbp->acc_slots,
                      C.thr, C.lgt, nullptr, C.pseed, C.pcounter, C.sel_next, sel8, nullptr, C.nchunk, C.want_lp, fbp->env_base);
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
FusedBufs tarl_to_bufs(const tarl_fused* f) {
  return FusedBufs{(uint2*)f->hdp,        (uint32_t*)f->tl, f->gc8,          (uint32_t*)f->post, (const float4*)f->st0,
                   f->slots,              f->ld_slots,      f->sel8,         f->sel,             (const NodeRec*)f->node_rec,
                   (const InRec*)f->in_rec, f->out_pad,
                   (long long*)f->acc_lp, f->acc_n,         f->acc_w,        f->a_origin,        f->a_dest,
                   f->a_dep,              f->a_status,      f->a_order,      f->cur_lo,          f->a_dep_sorted,
                   (const uint4*)f->a_win, f->a_ins,        f->a_rank,
                   f->acc_slots,          f->flags,         f->env_base};
}

// The device-resident copies of the pointer table (tarl_fused.bufs_dev): slot 0 = the table as the caller gave it, slot 1 =
// the same with the second log-prob accumulator bank (the merged insert + choice launch double-buffers it by frame parity).
// One thread, the tables by value in its kernel arguments: stream-ordered, no host copy, no synchronisation.
__global__ void k_set_bufs(FusedBufs a, FusedBufs b, FusedBufs* __restrict__ dst) {
  dst[0] = a;
  dst[1] = b;
}
extern "C" int64_t tarl_fused_bufs_bytes(void) { return 2 * (int64_t)sizeof(FusedBufs); }
static int upload_bufs(const tarl_fused* f, const FusedBufs& fb, long long* acc_lp_alt, hipStream_t s, const FusedBufs** dev) {
  TARL_REQUIRE(f->bufs_dev && ((uintptr_t)f->bufs_dev) % 16 == 0, "tarl_fused.bufs_dev missing or misaligned (tarl_fused_bufs_bytes)");
  FusedBufs alt = fb;
  if (acc_lp_alt) alt.acc_lp = acc_lp_alt;
  hipLaunchKernelGGL(k_set_bufs, dim3(1), dim3(1), 0, s, fb, alt, (FusedBufs*)f->bufs_dev);
  TARL_LAUNCH_CHECK();
  *dev = (const FusedBufs*)f->bufs_dev;
  return TARL_OK;
}

// rows per lane of the row pass (tunable: TARL_NCHUNK = 1, 2 or 4; measured 63.7 / 57.8 / 56.0 us per launch)
static int nchunk() {
  static int v = 0;
  if (v == 0) {
    const char* e = getenv("TARL_NCHUNK");
    v = e ? atoi(e) : 4;
    v = v >= 4 ? 4 : (v >= 2 ? 2 : 1);
  }
  return v;
}
static int64_t num_chunks(const tarl_plan* plan) { return ceil_div(plan->N, nchunk()); }
// rows per lane of the Direction gather: 1, 2 or 4, all their gathers in flight together (measured at B = 2048, config 4:
// 1 -> 65.7 us, 2 -> 42.4, 4 -> 39.8 per launch)
static int nchunk_dir() {
  static int v = 0;
  if (v == 0) {
    const char* e = getenv("TARL_NCHUNK_DIR");
    v = e ? atoi(e) : 4;
    v = v >= 4 ? 4 : (v >= 2 ? 2 : 1);
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
  TARL_REQUIRE(f->hdp && f->tl && f->gc8 && f->post && f->st0 && f->slots && f->sel8 && f->sel && f->acc_lp &&
                   f->acc_n && f->acc_w && f->flags && f->node_rec && f->in_rec && f->out_pad,
               "fused node buffers missing");
  TARL_REQUIRE(B >= 1 && B < ((int64_t)1 << 31) && Nmax >= 2, "bad sizes");
  TARL_REQUIRE(plan->N * B < ((int64_t)1 << 31), "N * B must stay below 2^31 (32-bit row indices in the frame kernels)");
  TARL_REQUIRE(Nmax <= 127, "the fused path packs NUMBER_OF_AGENT into one byte and the ring offset into seven bits: "
                            "Nmax must be <= 127 (use the unfused entry points for longer FIFOs)");
  TARL_REQUIRE(plan->max_out <= 126, "the fused path packs the chosen out-edge's rank into 7 bits: out-degree must be <= 126");
  TARL_REQUIRE(f->acc_slots >= 1 && f->acc_slots <= 4096, "acc_slots out of range");
  TARL_REQUIRE(f->ld_slots >= SLW * (int64_t)Nmax && f->ld_slots % SLW_ALIGN == 0, "slot row stride smaller than SLW*Nmax or misaligned (tarl_fused_slot_floats)");
  TARL_REQUIRE(num_chunks(plan) < 65536 && plan->num_row_chunks < 65536 && ceil_div(plan->N, nchunk_choice()) < 65536 &&
                   ceil_div(plan->N, nchunk_dir()) < 65536,
               "too many node chunks for one launch");
  TARL_REQUIRE(((uintptr_t)f->hdp | (uintptr_t)f->st0 | (uintptr_t)f->slots) % 32 == 0 &&
                   ((uintptr_t)f->tl | (uintptr_t)f->post) % 4 == 0,
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

extern "C" int64_t tarl_fused_slot_floats(int32_t Nmax) {
  return SLW == 8 ? (int64_t)8 * Nmax : ((int64_t)SLW * Nmax + 15) / 16 * 16;   // 12-byte slots: rows padded to 64 bytes
}

extern "C" int tarl_fused_pack(const tarl_plan* plan, const tarl_fused* f, const float* x, int64_t B, int64_t x_bstride,
                               int64_t ldx, int32_t Nmax, const float* cong, const float* edge_attr,
                               const float* agent_features, int64_t A, int64_t a_bstride, tarl_stream stream) {
  int rc = check_fused(plan, f, x, B, x_bstride, ldx, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(plan->E == 0 || edge_attr, "edge_attr is null");
  const Layout L{Nmax, ldx, x_bstride};
  const FusedBufs fb = tarl_to_bufs(f);
  const PlanOut P{plan->out_ptr, plan->out_dst};
  hipStream_t s = (hipStream_t)stream;
  TARL_CHECK_HIP(hipMemsetAsync(f->flags, 0, sizeof(int32_t), s));
  if (plan->N > 0) {
    hipLaunchKernelGGL(k_pack_nodes, dim3((unsigned)ceil_div(B * plan->N, FB)), dim3(FB), 0, s, x, L, B, plan->N, cong,
                       fb, (float4*)f->st0, P);
    TARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_pack_static, dim3((unsigned)ceil_div(plan->N > 4 ? plan->N : 4, FB)), dim3(FB), 0, s, plan->N,
                       plan->E, plan->in_ptr, plan->in_src, plan->in_eid, P, edge_attr, (const float4*)f->st0,
                       (NodeRec*)f->node_rec, (InRec*)f->in_rec, f->out_pad, f->flags);
    TARL_LAUNCH_CHECK();
  }
  if (agent_features) {
    TARL_REQUIRE(f->a_origin && f->a_dest && f->a_dep && f->a_status && A >= 1, "fused agent buffers missing");
    TARL_REQUIRE(A <= ((int64_t)1 << 24), "agent ids must stay below 2^24 (they are exact fp32 integers in the reference)");
    hipLaunchKernelGGL(k_pack_agents, dim3((unsigned)ceil_div(B * A, FB)), dim3(FB), 0, s, agent_features, B, A,
                       a_bstride, fb);
    TARL_LAUNCH_CHECK();
    if (f->a_order && f->a_win) {
      TARL_REQUIRE(f->a_ins && f->a_rank && f->a_dep_sorted, "a_win needs a_ins, a_rank and a_dep_sorted");
      hipLaunchKernelGGL(k_pack_window, dim3((unsigned)ceil_div(B * A, FB)), dim3(FB), 0, s, B, A, fb, (uint4*)f->a_win);
      TARL_LAUNCH_CHECK();
    }
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
  // The slot store is NOT zeroed: after the reset every row is empty and CLEAN (count byte 0), and a clean row's dead slots
  // are zero by rule whatever the store holds (fused_common.h) — export, head_arrival and tarl_fused_dead_slots apply the
  // rule, the frame kernels never read a clean row's dead slot. (The memset was 1.5 ms of every PPO iteration at 16 384
  // environments with 12-byte slots; with 32-byte records it would be 4 ms.)
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
  const PlanOut P{plan->out_ptr, plan->out_dst};
  hipLaunchKernelGGL(k_export_rows, dim3((unsigned)ceil_div(B * plan->N * Nmax, FB)), dim3(FB), 0, (hipStream_t)stream,
                     x, L, B, plan->N, tarl_to_bufs(f), last_step_time, P);
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

// TARL_ADDR32=0 keeps 64-bit addresses in the frame kernels whatever the batch size (developer knob: the instantiations that
// batches of 2^29 pairs and more run)
static bool addr32_ok() {
  static const bool ok = !(getenv("TARL_ADDR32") && atoi(getenv("TARL_ADDR32")) == 0);
  return ok;
}

// rows per lane of the Direction gather (TARL_NCHUNK_DIR = 1, 2 or 4)
// cnt8: a byte per (row, environment) holding every row's current count (a rollout's count slice of the frame before), or NULL
static int launch_direction(dim3 grid, unsigned threads, hipStream_t s, const tarl_plan* plan, const tarl_fused* f,
                            const float* edge_attr, const float* log_edge_attr, const uint8_t* sel8, const float* gumbel,
                            float* dtt, float log_eps, float time, float prev_time, uint64_t seed, uint64_t counter,
                            int64_t B, int Nmax, const FrameOut& out, const uint8_t* cnt8 = nullptr) {
#define DIR_LAUNCH_(NCH, SIB, CNT, O32)                                                                                   \
  hipLaunchKernelGGL((k_fused_direction<NCH, SIB, CNT, O32>), grid, dim3(threads), 0, s, (const NodeRec*)f->node_rec,          \
                     (const InRec*)f->in_rec, plan->in_eid, log_edge_attr, (const uint2*)f->hdp, (const uint32_t*)f->tl,  \
                     cnt8, (const uint8_t*)f->gc8, (const float*)f->slots, f->ld_slots, Nmax, sel8, (const float*)f->sel, gumbel, \
                     dtt, (uint32_t*)f->post, log_eps, time,   \
                     prev_time, seed, counter, (uint32_t)plan->E, (uint32_t)B, (uint32_t)plan->N, out, (uint64_t)f->env_base)
#define DIR_LAUNCH(NCH, SIB, CNT)                                                                                         \
  do {                                                                                                                     \
    if (o32) {                                                                                                             \
      DIR_LAUNCH_(NCH, SIB, CNT, true);                                                                                    \
    } else {                                                                                                               \
      DIR_LAUNCH_(NCH, SIB, CNT, false);                                                                                   \
    }                                                                                                                      \
  } while (0)
  const bool o32 = addr32_ok() && plan->N * B < ((int64_t)1 << 29);   // 8-byte words through 32-bit byte offsets (at32)
  // gc8 invariant (fused_common.h: head_arrival): a rollout stores the event byte in its LAST frame only (FrameOut::write_gc) —
  // and every frame for the metric environments, whose dtt_node reads it — while the tail word says "gc8 authoritative"
  // (TLF_AUTH) after every event. The per-edge delta_travel_time reads gc8 of EVERY environment: it may only be requested
  // by a caller whose frames all store the byte (the frame API does), and the per-node series only for metric environments.
  TARL_REQUIRE(dtt == nullptr || out.write_gc != 0,
               "per-edge delta_travel_time needs the event byte of every frame (FrameOut::write_gc): not available inside a rollout");
  // TARL_DIR_SIBLINGS=0 keeps the per-row gathers on a sibling graph (developer knob)
  static const bool sib_ok = !(getenv("TARL_DIR_SIBLINGS") && atoi(getenv("TARL_DIR_SIBLINGS")) == 0);
  // TARL_DIR_COUNT_BYTE=0 keeps the head words as the count's source (developer knob)
  static const bool cnt_ok = !(getenv("TARL_DIR_COUNT_BYTE") && atoi(getenv("TARL_DIR_COUNT_BYTE")) == 0);
  if (!cnt_ok) cnt8 = nullptr;
  switch (nchunk_dir()) {
    case 1: DIR_LAUNCH(1, false, false); break;
    case 2: DIR_LAUNCH(2, false, false); break;
    default:
      if (sib_ok && plan->siblings4) {
        if (cnt8) {
          DIR_LAUNCH(4, true, true);
        } else {
          DIR_LAUNCH(4, true, false);
        }
      } else {
        if (cnt8) {
          DIR_LAUNCH(4, false, true);
        } else {
          DIR_LAUNCH(4, false, false);
        }
      }
      break;
  }
#undef DIR_LAUNCH
#undef DIR_LAUNCH_
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

// rows per lane of the row pass (TARL_NCHUNK = 1, 2 or 4); on a graph whose rows group by their out-edge targets (plan
// row_siblings) the four-row form walks the plan's row-chunk table instead of consecutive rows (TARL_ROWS_SIBLINGS=0 keeps
// the consecutive chunks: developer knob)
static bool rows_sib(const tarl_plan* plan) {
  static const bool ok = !(getenv("TARL_ROWS_SIBLINGS") && atoi(getenv("TARL_ROWS_SIBLINGS")) == 0);
  return ok && nchunk() == 4 && plan->row_siblings && plan->row_chunks;
}
static int64_t num_row_chunks(const tarl_plan* plan) { return rows_sib(plan) ? plan->num_row_chunks : num_chunks(plan); }
static int launch_rows(dim3 grid, unsigned threads, hipStream_t s, const tarl_plan* plan, const tarl_fused* f,
                       const FusedBufs* fb, int Nmax, int64_t B, float* agent_features, int64_t A, int64_t a_bstride,
                       float time, const FrameOut& out) {
#define ROWS_LAUNCH_(NCH, SIB, FAPI, O32)                                                                                  \
  hipLaunchKernelGGL((k_fused_rows<NCH, SIB, FAPI, O32>), grid, dim3(threads), 0, s, (const NodeRec*)f->node_rec,               \
                     (const int32_t*)f->out_pad, plan->out_ptr, plan->out_dst, (const RowChunk*)plan->row_chunks,         \
                     (const uint32_t*)f->post, Nmax, (uint32_t)B, (uint32_t)plan->N, fb, agent_features, A, a_bstride,    \
                     time, out)
#define ROWS_LAUNCH(NCH, SIB)                                                                                              \
  do {                                                                                                                     \
    if (fapi) {                                                                                                            \
      ROWS_LAUNCH_(NCH, SIB, true, false);                                                                                 \
    } else if (o32) {                                                                                                      \
      ROWS_LAUNCH_(NCH, SIB, false, true);                                                                                 \
    } else {                                                                                                               \
      ROWS_LAUNCH_(NCH, SIB, false, false);                                                                                \
    }                                                                                                                      \
  } while (0)
  const bool fapi = out.countsf || out.popped || out.withdrawn;
  const bool o32 = addr32_ok() && plan->N * B < ((int64_t)1 << 29);   // 8-byte words through 32-bit byte offsets
  grid.y = (unsigned)num_row_chunks(plan);
  switch (nchunk()) {
    case 1: ROWS_LAUNCH(1, false); break;
    case 2: ROWS_LAUNCH(2, false); break;
    default:
      if (rows_sib(plan)) {
        ROWS_LAUNCH(4, true);
      } else {
        ROWS_LAUNCH(4, false);
      }
      break;
  }
#undef ROWS_LAUNCH
#undef ROWS_LAUNCH_
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

static int check_frame_args(const tarl_plan* plan, const tarl_fused* f, int64_t B, const float* agent_features, int64_t A,
                            int64_t a_bstride, const int32_t* ins_scratch, const float* edge_attr,
                            const float* log_edge_attr) {
  TARL_REQUIRE(agent_features && A >= 1 && ins_scratch, "agents / scratch missing");
  TARL_REQUIRE(f->a_origin && f->a_dest && f->a_dep && f->a_status, "fused agent buffers missing");
  TARL_REQUIRE(f->a_order == nullptr || (f->cur_lo != nullptr && f->a_dep_sorted != nullptr),
               "a_order needs cur_lo and a_dep_sorted");
  TARL_REQUIRE(B == 1 || a_bstride >= A * AG_COLS, "agent stride smaller than one population");
  TARL_REQUIRE(plan->E == 0 || (edge_attr && log_edge_attr), "edge constants missing");
  return TARL_OK;
}

extern "C" int tarl_fused_frame(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax,
                                const float* thresholds, const int64_t* log_probs, const float* entropy1,
                                const float* uniform, uint64_t policy_seed, uint64_t policy_counter,
                                float* agent_features, int64_t A, int64_t a_bstride, const float* edge_attr,
                                const float* log_edge_attr, float log_eps, int use_cong, float time, float prev_time,
                                const float* gumbel, uint64_t seed, uint64_t counter, float* delta_travel_time,
                                uint8_t* popped, uint8_t* withdrawn, int32_t* ins_scratch, int32_t* choice,
                                float* log_prob, float* entropy, float* reward, float* counts, tarl_stream stream) {
  int rc = tarl_check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  // thresholds == NULL: SELECTED_ROAD was set by the caller (tarl_fused_apply_choice); no sample / log-prob in this call
  TARL_REQUIRE((thresholds && log_probs && entropy1) || (!thresholds && !log_prob && !entropy && !choice),
               "policy tables missing (call tarl_fused_policy_prepare), or outputs of the skipped choice phase requested");
  rc = check_frame_args(plan, f, B, agent_features, A, a_bstride, ins_scratch, edge_attr, log_edge_attr);
  if (rc) return rc;
  if (plan->N == 0) return TARL_OK;
  const FusedBufs fb = tarl_to_bufs(f);
  const PlanOut P{plan->out_ptr, plan->out_dst};
  hipStream_t s = (hipStream_t)stream;
  const unsigned threads = tile_threads(B);
  const dim3 grid((unsigned)ceil_div(B, threads), (unsigned)num_chunks(plan));
  const dim3 grid_c((unsigned)ceil_div(B, threads), (unsigned)ceil_div(plan->N, nchunk_choice()));
  if (thresholds) {
    hipLaunchKernelGGL(k_fused_choice, grid_c, dim3(threads), 0, s, plan->out_ptr, plan->out_eid, plan->group_of_node,
                       plan->G, B, plan->N, fb.acc_lp, fb.acc_slots, thresholds, (const long long*)log_probs, uniform,
                       policy_seed, policy_counter, f->sel8, (const uint8_t*)f->sel8, choice, nchunk_choice(),
                       log_prob != nullptr ? 1 : 0, f->env_base);
    TARL_LAUNCH_CHECK();
  }
  const FusedBufs* fbd = nullptr;
  rc = upload_bufs(f, fb, nullptr, s, &fbd);
  if (rc) return rc;
  const bool timed = tarl_prof_mark(s, 0) != nullptr;
  const dim3 grid_d((unsigned)ceil_div(B, threads), (unsigned)ceil_div(plan->N, nchunk_dir()));
  const FrameOut out{nullptr, counts, popped, withdrawn, nullptr, nullptr, 0, nullptr, 1};
  rc = launch_direction(grid_d, threads, s, plan, f, edge_attr, log_edge_attr, (const uint8_t*)f->sel8, gumbel,
                        delta_travel_time, log_eps, time, prev_time, seed, counter, B, (int)Nmax, out);
  if (rc) return rc;
  if (timed) (void)tarl_prof_mark(s, 1);
  rc = launch_rows(grid, threads, s, plan, f, fbd, (int)Nmax, B, agent_features, A, a_bstride, time, out);
  if (rc) return rc;
  if (timed) (void)tarl_prof_mark(s, 2);
  hipLaunchKernelGGL(k_fused_insert, dim3((unsigned)B), dim3(INSB), 0, s, (int)Nmax, B, plan->N, fbd, P,
                     (const uint8_t*)f->sel8, agent_features, A, a_bstride, use_cong, time, ins_scratch, entropy1, reward,
                     out, log_prob, entropy);
  TARL_LAUNCH_CHECK();
  if (timed) (void)tarl_prof_mark(s, 3);
  return TARL_OK;
}

// x[b, src(e), SELECTED_ROAD] = dst(e) for the chosen edge of every node, on the packed state (the choice phase of
// SimulatorEnv._step for an externally sampled action): choice int32 [B][N] = chosen edge id per node, -1 = none.
__global__ __launch_bounds__(FB) void k_apply_choice8(const int32_t* __restrict__ choice, const int32_t* __restrict__ out_ptr,
                                                      const int32_t* __restrict__ out_eid, int64_t B, int64_t N,
                                                      uint8_t* __restrict__ sel8) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;  // gid = i * B + b
  if (gid >= B * N) return;
  const int64_t i = gid / B, b = gid - i * B;
  const int32_t e = choice[b * N + i];
  uint32_t code = (sel8[gid] & 0x7Fu) | SEL_CARRIED;
  if (e >= 0) {
    const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];
    for (int32_t k = k0; k < k1; ++k)
      if (out_eid[k] == e) code = (uint32_t)(k - k0);
  }
  sel8[gid] = (uint8_t)code;
}

extern "C" int tarl_fused_apply_choice(const tarl_plan* plan, const tarl_fused* f, int64_t B, const int32_t* choice,
                                       tarl_stream stream) {
  TARL_REQUIRE(plan && f && f->sel8 && choice && B >= 1, "null argument");
  if (plan->N == 0) return TARL_OK;
  hipLaunchKernelGGL(k_apply_choice8, dim3((unsigned)ceil_div(B * plan->N, FB)), dim3(FB), 0, (hipStream_t)stream, choice,
                     plan->out_ptr, plan->out_eid, B, plan->N, f->sel8);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

// side stream + events of the all-frames choice (one set per device, created on first use, kept for the process lifetime)
#define CHOICE_CHUNK 32          // frames per side-stream block (the first block is CHOICE_FIRST frames: it is waited for)
#define CHOICE_FIRST 4
#define CHOICE_MAX_CHUNKS 256
struct ChoiceSide {
  hipStream_t stream;
  hipEvent_t start;
  hipEvent_t done[CHOICE_MAX_CHUNKS];
  bool ready;
};
static ChoiceSide g_choice_side[64];

static int choice_side(ChoiceSide** out) {
  int dev = 0;
  TARL_CHECK_HIP(hipGetDevice(&dev));
  ChoiceSide* cs = &g_choice_side[dev & 63];
  if (!cs->ready) {
    TARL_CHECK_HIP(hipStreamCreateWithFlags(&cs->stream, hipStreamNonBlocking));
    TARL_CHECK_HIP(hipEventCreateWithFlags(&cs->start, hipEventDisableTiming));
    for (int i = 0; i < CHOICE_MAX_CHUNKS; ++i) TARL_CHECK_HIP(hipEventCreateWithFlags(&cs->done[i], hipEventDisableTiming));
    cs->ready = true;
  }
  *out = cs;
  return TARL_OK;
}

// Environments per wave of the packed insert kernel (k_fused_insert2; TARL_INSERT_EPW = 1, 2, 4 or 8 forces it; 1 = one
// workgroup per environment, k_fused_insert). A frame's window holds the agents due in it, and an environment's share of the
// wave (64 / EPW lanes, INS_CAP / EPW list entries) should take them in one step: an environment with more candidates than
// its share of the list sits the packed part out and is served alone afterwards, one after the other (measured with ~27 due
// per frame and EPW = 8: 260 us per launch instead of 36). With the schedule's due rate known (tarl_fused.due_rate, from
// pack): the largest EPW whose list share holds twice the expected candidates of a frame. Without it, by the size of the
// population: 8 up to 20 000 agents (BASELINE config 4: ~5 due per frame) when the launch is large, 4 up to 32 768, 2 up to
// 65 536, one wave per environment beyond (config 5: 262 144 agents, ~70 per frame).
static int insert_envs_per_wave(const tarl_fused* f, int64_t A, int64_t B, int64_t T, const float* times_host) {
  static const bool pair_ok = !(getenv("TARL_INSERT_PAIR") && atoi(getenv("TARL_INSERT_PAIR")) == 0);
  static const int epw_env = getenv("TARL_INSERT_EPW") ? atoi(getenv("TARL_INSERT_EPW")) : 0;
  int epw = A <= 20000 && B >= 12288 ? 8 : (A <= 32768 ? 4 : (A <= 65536 ? 2 : 1));
  if (f->due_rate > 0.0f) {
    const float dt = T > 1 ? times_host[1] - times_host[0] : 1.0f;
    const float per_frame = f->due_rate * (dt > 0.0f ? dt : 1.0f);
    while (epw > 1 && 2.0f * per_frame > (float)(INS_CAP / epw)) epw >>= 1;
  }
  if (epw_env == 1 || epw_env == 2 || epw_env == 4 || epw_env == 8) epw = epw_env;
  return (pair_ok && f->a_order && f->a_win) ? epw : 1;
}

// the insert launch of a frame whose action is already in `sel_t` (no choice part): packed (epw > 1) or one workgroup per environment
static int launch_insert(int epw, hipStream_t s, int Nmax, int64_t B, int64_t N, const FusedBufs* fbt, PlanOut P,
                         const uint8_t* sel_t, float* agent_features, int64_t A, int64_t a_bstride, int use_cong, float time,
                         int32_t* ins_scratch, const float* entropy1, float* reward_t, const FrameOut& out, float* lp_t,
                         float* ent_t) {
#define INS2_LAUNCH(EPW_)                                                                                                         \
  hipLaunchKernelGGL((k_fused_insert2<EPW_>), dim3((unsigned)ceil_div(B, EPW_)), dim3(INSB), 0, s, Nmax, B, N, fbt, P, sel_t,   \
                     agent_features, A, a_bstride, use_cong, time, ins_scratch, entropy1, reward_t, out, lp_t, ent_t)
  if (epw == 8) {
    INS2_LAUNCH(8);
  } else if (epw == 4) {
    INS2_LAUNCH(4);
  } else if (epw == 2) {
    INS2_LAUNCH(2);
  } else {
    hipLaunchKernelGGL(k_fused_insert, dim3((unsigned)B), dim3(INSB), 0, s, Nmax, B, N, fbt, P, sel_t, agent_features, A,
                       a_bstride, use_cong, time, ins_scratch, entropy1, reward_t, out, lp_t, ent_t);
  }
#undef INS2_LAUNCH
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int64_t tarl_fused_rollout_scratch_ints(const tarl_plan* plan, int64_t T, int64_t B) {
  // unresolved-draw list | packed policy records | per-node draw records (64-byte aligned) | per-(frame, environment)
  // log-prob accumulators (int64; last, so that every other offset is independent of T)
  return plan && T >= 1 && B >= 1 ? (int64_t)(4 + 2 * FIX_CAP) + 4 * (plan->E + 4) + 2 * T * B + 16 * (plan->N + 1) : -1;
}

// T consecutive frames with device noise: the collector loop in one foreign call. Each frame is direction -> rows ->
// insert on the caller's stream. The action of frame t is slice t of the action buffer, which is also the
// SELECTED_ROAD column that frame's Direction gather and insert read. With an action buffer and choice_scratch (the
// default, TARL_ROLLOUT_MERGE=2) the live policy's draws for ALL frames are made on a side stream in blocks of
// CHOICE_CHUNK frames (the sample does not depend on the state: k_fused_choice_all), the first, short block is waited
// for; otherwise frame t+1's choice shares the launch of frame t's insert (k_fused_insert_choice, mode 1) or is a
// launch of its own (mode 0). Per-frame draws on a side stream and folding them into the row pass were measured and
// rejected.
extern "C" int tarl_fused_rollout(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax, int64_t T,
                                  const float* times_host, float prev_time, const float* thresholds,
                                  const int64_t* log_probs, const float* entropy1, uint64_t policy_seed,
                                  uint64_t policy_counter0, float* agent_features, int64_t A, int64_t a_bstride,
                                  const float* edge_attr, const float* log_edge_attr, float log_eps, int use_cong,
                                  uint64_t seed, uint64_t counter0, int32_t* ins_scratch, uint8_t* sel_scratch,
                                  int64_t* acc_scratch, int32_t* choice_scratch, uint8_t* choice, float* log_prob,
                                  float* entropy, float* reward,
                                  uint8_t* counts, int32_t metrics_envs, float* dtt_node, uint8_t* events, int32_t* leg,
                                  tarl_stream stream) {
  int rc = tarl_check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(T >= 1 && times_host, "bad frame count / times");
  TARL_REQUIRE(thresholds && log_probs && entropy1, "policy tables missing (call tarl_fused_policy_prepare)");
  rc = check_frame_args(plan, f, B, agent_features, A, a_bstride, ins_scratch, edge_attr, log_edge_attr);
  if (rc) return rc;
  TARL_REQUIRE(metrics_envs >= 0 && metrics_envs <= B, "metrics_envs out of range");
  TARL_REQUIRE(metrics_envs > 0 || (!dtt_node && !events), "per-node series need metrics_envs > 0");
  if (plan->N == 0) return TARL_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t N = plan->N, NB = N * B;
  const unsigned threads = tile_threads(B);
  const dim3 grid((unsigned)ceil_div(B, threads), (unsigned)num_chunks(plan));
  const dim3 grid_c((unsigned)ceil_div(B, threads), (unsigned)ceil_div(N, nchunk_choice()));
  const dim3 grid_d((unsigned)ceil_div(B, threads), (unsigned)ceil_div(N, nchunk_dir()));
  const int want_lp = log_prob != nullptr ? 1 : 0;
  const FusedBufs fb = tarl_to_bufs(f);
  const PlanOut P{plan->out_ptr, plan->out_dst};
  // slice t of the action buffer (or, without one, the ping-pong pair f->sel8 / sel_scratch)
  auto slice = [&](int64_t t) -> uint8_t* {
    if (choice) return choice + t * NB;
    return (sel_scratch && (t & 1)) ? sel_scratch : f->sel8;
  };
  // steady state: frame t's insert and frame t+1's choice share ONE launch (k_fused_insert_choice); needs two distinct
  // SELECTED_ROAD slices and double-buffered log-prob accumulator banks
  // TARL_ROLLOUT_MERGE: 2 (default) = all actions precomputed on a side stream (needs the action buffer + choice_scratch),
  // 1 = frame t+1's choice shares frame t's insert launch, 0 = a choice launch per frame
  const char* knob = getenv("TARL_ROLLOUT_MERGE");
  const int mode = knob ? atoi(knob) : 2;
  const bool ahead = mode >= 2 && choice && choice_scratch && ceil_div(T, CHOICE_CHUNK) + 1 <= CHOICE_MAX_CHUNKS &&
                     ceil_div(N, CHOICE_SEG) < 65536;
  const bool merge = !ahead && mode >= 1 && (choice || sel_scratch) && acc_scratch && T > 1;
  long long* acc_buf[2] = {(long long*)f->acc_lp, merge ? (long long*)acc_scratch : (long long*)f->acc_lp};
  if (merge) TARL_CHECK_HIP(hipMemsetAsync(acc_scratch, 0, (size_t)(f->acc_slots * B) * sizeof(int64_t), s));
  const FusedBufs* fbd = nullptr;      // [0]: the table with acc_buf[0], [1]: with acc_buf[1]
  rc = upload_bufs(f, fb, acc_buf[1], s, &fbd);
  if (rc) return rc;
  ChoiceSide* side = nullptr;
  if (ahead) {
    rc = choice_side(&side);
    if (rc) return rc;
    int32_t* fix = choice_scratch;
    PRec* ptab = (PRec*)(choice_scratch + 4 + 2 * FIX_CAP);
    TARL_REQUIRE(((uintptr_t)ptab) % 16 == 0, "choice_scratch must be 16-byte aligned");
    // the side stream starts behind everything already queued on the caller's stream (tables, reset, earlier rollouts)
    TARL_CHECK_HIP(hipEventRecord(side->start, s));
    TARL_CHECK_HIP(hipStreamWaitEvent(side->stream, side->start, 0));
    TARL_CHECK_HIP(hipMemsetAsync(fix, 0, 2 * sizeof(int32_t), side->stream));
    hipLaunchKernelGGL(k_pack_ptab, dim3((unsigned)ceil_div(plan->E + 4, FB)), dim3(FB), 0, side->stream, plan->E,
                       thresholds, (const long long*)log_probs, ptab);
    TARL_LAUNCH_CHECK();
    // the per-node records come BEFORE the log-prob accumulators, so that no offset depends on T: a caller may reuse one
    // scratch buffer for rollouts of different lengths (an episode end splits a collector batch), and the accumulators
    // [T][B] — zero between rollouts, k_choice_lp_finish re-arms what a rollout used — must never be overlaid by a record
    // table of another call
    PNode* pnode = (PNode*)(((uintptr_t)(ptab + plan->E + 4) + 63) & ~(uintptr_t)63);
    long long* lp_acc = (long long*)(pnode + N);
    hipLaunchKernelGGL(k_pack_pnode, dim3((unsigned)ceil_div(N, FB)), dim3(FB), 0, side->stream, N,
                       (const NodeRec*)f->node_rec, plan->group_of_node, thresholds, (const long long*)log_probs, pnode);
    TARL_LAUNCH_CHECK();
    for (int64_t c = 0, t0 = 0; t0 < T; ++c) {
      const int64_t want = c == 0 ? CHOICE_FIRST : CHOICE_CHUNK;
      const unsigned nf = (unsigned)(T - t0 < want ? T - t0 : want);
      // TARL_CHOICE_QUAD=0 keeps the one-node-per-step form (developer knob)
      static const bool quad_ok = !(getenv("TARL_CHOICE_QUAD") && atoi(getenv("TARL_CHOICE_QUAD")) == 0);
      const bool quad = quad_ok && plan->G == N && N % 4 == 0;
      hipLaunchKernelGGL(quad ? k_fused_choice_all<true> : k_fused_choice_all<false>,
                         dim3((unsigned)ceil_div(B, threads), (unsigned)ceil_div(N, CHOICE_SEG), nf),
                         dim3(threads), 0, side->stream, (const PNode*)pnode, (const PRec*)ptab,
                         plan->group_of_node, (uint32_t)plan->G, (uint32_t)B, (uint32_t)N, t0, policy_seed,
                         policy_counter0, (const uint8_t*)f->sel8, choice, log_prob ? lp_acc : nullptr, fix, f->flags,
                         (uint64_t)f->env_base);
      TARL_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_choice_fixup, dim3(1), dim3(ENVB), 0, side->stream, (uint32_t)B, (uint32_t)N,
                         (const uint8_t*)f->sel8, choice, fix);
      TARL_LAUNCH_CHECK();
      if (log_prob) {
        hipLaunchKernelGGL(k_choice_lp_finish, dim3((unsigned)ceil_div((int64_t)nf * B, FB)), dim3(FB), 0, side->stream,
                           (int64_t)nf * B, lp_acc + t0 * B, log_prob + t0 * B);
        TARL_LAUNCH_CHECK();
      }
      TARL_CHECK_HIP(hipEventRecord(side->done[c], side->stream));
      t0 += nf;
    }
  } else {
    hipLaunchKernelGGL(k_fused_choice, grid_c, dim3(threads), 0, s, plan->out_ptr, plan->out_eid, plan->group_of_node,
                       plan->G, B, N, acc_buf[0], fb.acc_slots, thresholds, (const long long*)log_probs,
                       (const float*)nullptr, policy_seed, policy_counter0, slice(0), (const uint8_t*)f->sel8,
                       (int32_t*)nullptr, nchunk_choice(), want_lp, f->env_base);
    TARL_LAUNCH_CHECK();
  }
  const int epw_packed = insert_envs_per_wave(f, A, B, T, times_host);
  for (int64_t t = 0; t < T; ++t) {
    const int cur = merge ? (int)(t & 1) : 0;
    const float time = times_host[t];
    const FusedBufs* fbt = fbd + cur;
    const uint8_t* sel_t = slice(t);
    const int64_t m = metrics_envs;
    const FrameOut out{counts ? counts + t * NB : nullptr,
                       nullptr,
                       nullptr,
                       nullptr,
                       events ? events + t * N * m : nullptr,
                       dtt_node ? dtt_node + t * N * m : nullptr,
                       metrics_envs,
                       leg ? leg + t * 2 * B : nullptr,
                       t + 1 == T ? 1 : 0};
    if (ahead && (t == 0 || (t >= CHOICE_FIRST && (t - CHOICE_FIRST) % CHOICE_CHUNK == 0)))
      TARL_CHECK_HIP(hipStreamWaitEvent(s, side->done[t == 0 ? 0 : 1 + (t - CHOICE_FIRST) / CHOICE_CHUNK], 0));
    const bool timed = tarl_prof_mark(s, 0) != nullptr;
    rc = launch_direction(grid_d, threads, s, plan, f, edge_attr, log_edge_attr, sel_t, nullptr, nullptr, log_eps, time,
                          t > 0 ? times_host[t - 1] : prev_time, seed, counter0 + (uint64_t)t, B, (int)Nmax, out,
                          (counts && t > 0) ? counts + (t - 1) * NB : nullptr);   // the counts after frame t - 1
    if (rc) return rc;
    if (timed) (void)tarl_prof_mark(s, 1);
    rc = launch_rows(grid, threads, s, plan, f, fbt, (int)Nmax, B, agent_features, A, a_bstride, time, out);
    if (rc) return rc;
    if (timed) (void)tarl_prof_mark(s, 2);
    float* reward_t = reward ? reward + t * B : nullptr;
    float* lp_t = (log_prob && !ahead) ? log_prob + t * B : nullptr;   // ahead: written by k_fused_choice_all
    float* ent_t = entropy ? entropy + t * B : nullptr;
    if (ahead) {
      rc = launch_insert(epw_packed, s, (int)Nmax, B, N, fbt, P, sel_t, agent_features, A, a_bstride, use_cong, time, ins_scratch,
                         entropy1, reward_t, out, lp_t, ent_t);
      if (rc) return rc;
    } else if (merge && t + 1 < T) {
      const ChoiceArgs C{plan->out_ptr, plan->out_eid, plan->group_of_node, plan->G, thresholds,
                         (const long long*)log_probs, policy_seed, policy_counter0 + (uint64_t)(t + 1), nchunk_choice(),
                         want_lp, slice(t + 1), acc_buf[cur ^ 1], grid_c.x, grid_c.x * grid_c.y};
      hipLaunchKernelGGL(k_fused_insert_choice, dim3(C.choice_blocks + (unsigned)B), dim3(threads), 0, s, C, (int)Nmax,
                         B, N, fbt, P, sel_t, agent_features, A, a_bstride, use_cong, time, ins_scratch, entropy1,
                         reward_t, out, lp_t, ent_t);
      TARL_LAUNCH_CHECK();
    } else {
      hipLaunchKernelGGL(k_fused_insert, dim3((unsigned)B), dim3(INSB), 0, s, (int)Nmax, B, N, fbt, P, sel_t,
                         agent_features, A, a_bstride, use_cong, time, ins_scratch, entropy1, reward_t, out, lp_t, ent_t);
      TARL_LAUNCH_CHECK();
      if (t + 1 < T) {   // unmerged: the next frame's choice as its own launch
        hipLaunchKernelGGL(k_fused_choice, grid_c, dim3(threads), 0, s, plan->out_ptr, plan->out_eid,
                           plan->group_of_node, plan->G, B, N, acc_buf[cur], fb.acc_slots, thresholds,
                           (const long long*)log_probs, (const float*)nullptr, policy_seed,
                           policy_counter0 + (uint64_t)(t + 1), slice(t + 1), sel_t, (int32_t*)nullptr, nchunk_choice(),
                           want_lp, f->env_base);
        TARL_LAUNCH_CHECK();
      }
    }
    if (timed) (void)tarl_prof_mark(s, 3);
  }
  if (slice(T - 1) != f->sel8)   // the last frame's SELECTED_ROAD lives in the action buffer / scratch: bring it home
    TARL_CHECK_HIP(hipMemcpyAsync(f->sel8, slice(T - 1), (size_t)NB, hipMemcpyDeviceToDevice, s));
  return TARL_OK;
}

// ---- T frames with a STATE-DEPENDENT policy (the per-edge MLP head) in one foreign call --------------------------------------
// Nothing of GraphDistribution can be hoisted: per frame observation (tarl_fused_obs16) -> logits (tarl_policy_edge_mlp_fwd)
// -> action + log-prob + SELECTED_ROAD (tarl_graphdist_rollout) -> Direction -> rows -> insert, eight launches queued
// back to back by one host loop in C (the same entry points a caller could queue himself, minus a foreign-call round
// trip per launch). The observations of the frames an optimiser minibatch will use are drawn beforehand (the draw does
// not depend on the data): keep_ptr_host[t] .. keep_ptr_host[t + 1] index the (environment, slot) pairs of frame t.
__global__ __launch_bounds__(FB) void k_obs_keep(const float4* __restrict__ obs, int64_t row4,
                                                 const int32_t* __restrict__ env, const int32_t* __restrict__ slot,
                                                 float4* __restrict__ keep) {
  const int64_t j = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (j >= row4) return;
  keep[(int64_t)slot[blockIdx.y] * row4 + j] = obs[(int64_t)env[blockIdx.y] * row4 + j];
}

// tarl_fused_set_actions: env-major action bytes (choice8 [B][N], what the samplers and the LDS-resident rollout write) ->
// the SELECTED_ROAD column the frame kernels read (sel8 [N][B]): 64 x 64 byte tiles turned through LDS, whole 64-byte runs on
// both sides. A road that drew nothing (bit 7) keeps its previous value: the code is completed from the old sel8 byte and
// written back to the action buffer as well. (Measured as a replacement for the per-frame sampler's own scattered sel8
// stores in tarl_fused_rollout_policy, B = 2048: sampler 102.7 -> 99.5 us, this kernel 8.9 us — not taken there.)
__global__ __launch_bounds__(256) void k_sel8_from_choice8(uint8_t* __restrict__ choice8, uint8_t* __restrict__ sel8,
                                                           int64_t B, int64_t N) {
  __shared__ uint8_t tile[64][68];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t b0 = (int64_t)blockIdx.x * 64, i0 = (int64_t)blockIdx.y * 64;
#pragma unroll 4
  for (int r = w; r < 64; r += 4) {            // row r = environment b0 + r, lanes along the roads
    const int64_t b = b0 + r, i = i0 + lane;
    tile[r][lane] = (b < B && i < N) ? choice8[b * N + i] : (uint8_t)0;
  }
  __syncthreads();
#pragma unroll 4
  for (int r = w; r < 64; r += 4) {            // row r = road i0 + r, lanes along the environments
    const int64_t i = i0 + r, b = b0 + lane;
    if (b < B && i < N) {
      uint32_t c = tile[lane][r];
      if (c & SEL_CARRIED) {
        c = (sel8[i * B + b] & 0x7Fu) | SEL_CARRIED;
        choice8[b * N + i] = (uint8_t)c;
      }
      sel8[i * B + b] = (uint8_t)c;
    }
  }
}

extern "C" int tarl_fused_set_actions(const tarl_plan* plan, const tarl_fused* f, int64_t B, uint8_t* choice8,
                                      tarl_stream stream) {
  TARL_REQUIRE(plan && f && f->sel8 && choice8, "null argument");
  TARL_REQUIRE(B >= 1, "bad batch size");
  if (plan->N == 0) return TARL_OK;
  hipLaunchKernelGGL(k_sel8_from_choice8, dim3((unsigned)ceil_div(B, 64), (unsigned)ceil_div(plan->N, 64)), dim3(256), 0,
                     (hipStream_t)stream, choice8, f->sel8, B, plan->N);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_fused_rollout_policy(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax, int64_t T,
                                         const float* times_host, float prev_time, const float* x, int64_t x_bstride,
                                         int64_t ldx, float* agent_features, int64_t A, int64_t a_bstride,
                                         const float* edge_attr, const float* log_edge_attr, float log_eps, int use_cong,
                                         const float* w1, const float* b1, const float* w2, const float* b2,
                                         const float* w3, const float* b3, int precision, float temperature,
                                         uint64_t policy_seed, uint64_t policy_counter0, uint64_t seed,
                                         uint64_t counter0, const int64_t* keep_ptr_host, const int32_t* keep_env,
                                         const int32_t* keep_slot, float* obs_keep, float* obs_scratch,
                                         float* logits_scratch, void* dist_scratch, int32_t* ins_scratch,
                                         uint8_t* choice8, float* log_prob, float* reward, uint8_t* counts,
                                         int32_t metrics_envs, float* dtt_node, uint8_t* events, int32_t* leg,
                                         tarl_stream stream) {
  int rc = tarl_check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(T >= 1 && times_host, "bad frame count / times");
  TARL_REQUIRE(x && obs_scratch && logits_scratch && dist_scratch, "observation / logits / sampler scratch missing");
  TARL_REQUIRE(!keep_ptr_host || (keep_env && keep_slot && obs_keep), "keep list without its arrays");
  TARL_REQUIRE(precision >= 0 && precision <= 2,
               "precision: 0 = fp32 MFMA logits, 1 = bf16 logits on bf16 observations, 2 = fp32-accurate logits on the bf16 pipe (bf16x3)");
  rc = check_frame_args(plan, f, B, agent_features, A, a_bstride, ins_scratch, edge_attr, log_edge_attr);
  if (rc) return rc;
  TARL_REQUIRE(metrics_envs >= 0 && metrics_envs <= B, "metrics_envs out of range");
  TARL_REQUIRE(metrics_envs > 0 || (!dtt_node && !events), "per-node series need metrics_envs > 0");
  if (plan->N == 0) return TARL_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t N = plan->N, NB = N * B;
  const unsigned threads = tile_threads(B);
  const dim3 grid((unsigned)ceil_div(B, threads), (unsigned)num_chunks(plan));
  const dim3 grid_d((unsigned)ceil_div(B, threads), (unsigned)ceil_div(N, nchunk_dir()));
  const FusedBufs fbh = tarl_to_bufs(f);
  const FusedBufs* fb = nullptr;
  rc = upload_bufs(f, fbh, nullptr, s, &fb);
  if (rc) return rc;
  const PlanOut P{plan->out_ptr, plan->out_dst};
  const int epw_packed = insert_envs_per_wave(f, A, B, T, times_host);
  for (int64_t t = 0; t < T; ++t) {
    const bool keep_t = keep_ptr_host && keep_ptr_host[t + 1] > keep_ptr_host[t];
    const int64_t lo = keep_t ? keep_ptr_host[t] : 0, n = keep_t ? keep_ptr_host[t + 1] - lo : 0;
    TARL_REQUIRE(n < 65536, "more than 65535 kept observations in one frame");
    if (precision == 1) {
      // bf16 logits read bf16 observations (half the bytes written here and gathered by the MLP); the fp32 rows an
      // optimiser step keeps are evaluated for their few environments only
      rc = tarl_fused_obs16_bf16(plan, f, x, B, x_bstride, ldx, Nmax, agent_features, A, a_bstride,
                                 (uint16_t*)obs_scratch, stream);
      if (rc) return rc;
      if (keep_t) {
        rc = tarl_fused_obs16_rows(plan, f, x, B, x_bstride, ldx, Nmax, agent_features, A, a_bstride, keep_env + lo,
                                   keep_slot + lo, n, obs_keep, stream);
        if (rc) return rc;
      }
    } else {
      rc = tarl_fused_obs16(plan, f, x, B, x_bstride, ldx, Nmax, agent_features, A, a_bstride, obs_scratch, stream);
      if (rc) return rc;
      if (keep_t) {
        hipLaunchKernelGGL(k_obs_keep, dim3((unsigned)ceil_div(N * 4, FB), (unsigned)n), dim3(FB), 0, s,
                           (const float4*)obs_scratch, N * 4, keep_env + lo, keep_slot + lo, (float4*)obs_keep);
        TARL_LAUNCH_CHECK();
      }
    }
    // (live timing, bench.py: HIP events around the per-edge MLP of the timed frames — slot 0 of tarl_prof_collect)
    const bool timed = tarl_prof_mark(s, 0) != nullptr;
    rc = tarl_policy_edge_mlp_fwd(plan, obs_scratch, B, edge_attr, w1, b1, w2, b2, w3, b3,
                                  precision == 1 ? 2 : (precision == 2 ? 3 : 0), logits_scratch, stream);
    if (rc) return rc;
    if (timed) (void)tarl_prof_mark(s, 1);
    rc = tarl_graphdist_rollout_at(plan, logits_scratch, B, temperature, nullptr, policy_seed,
                                   policy_counter0 + (uint64_t)t, dist_scratch, nullptr,
                                   choice8 ? choice8 + t * NB : nullptr, f->sel8, log_prob ? log_prob + t * B : nullptr,
                                   f->env_base, stream);
    if (rc) return rc;
    const float time = times_host[t];
    const int64_t m = metrics_envs;
    const FrameOut out{counts ? counts + t * NB : nullptr,
                       nullptr,
                       nullptr,
                       nullptr,
                       events ? events + t * N * m : nullptr,
                       dtt_node ? dtt_node + t * N * m : nullptr,
                       metrics_envs,
                       leg ? leg + t * 2 * B : nullptr,
                       t + 1 == T ? 1 : 0};
    rc = launch_direction(grid_d, threads, s, plan, f, edge_attr, log_edge_attr, (const uint8_t*)f->sel8, nullptr, nullptr,
                          log_eps, time, t > 0 ? times_host[t - 1] : prev_time, seed, counter0 + (uint64_t)t, B, (int)Nmax, out);
    if (rc) return rc;
    rc = launch_rows(grid, threads, s, plan, f, fb, (int)Nmax, B, agent_features, A, a_bstride, time, out);
    if (rc) return rc;
    rc = launch_insert(epw_packed, s, (int)Nmax, B, N, fb, P, (const uint8_t*)f->sel8, agent_features, A, a_bstride, use_cong, time,
                       ins_scratch, (const float*)nullptr, reward ? reward + t * B : nullptr, out, (float*)nullptr, (float*)nullptr);
    if (rc) return rc;
  }
  return TARL_OK;
}

// ---- rollout bytes -> the formats of the unfused entry points (minibatch gather of the PPO update) ------------------------
__global__ __launch_bounds__(FB) void k_rollout_gather(const uint8_t* __restrict__ choice, const uint8_t* __restrict__ counts,
                                                       int64_t B, int64_t N, int env_minor,
                                                       const int64_t* __restrict__ idx, int64_t rows,
                                                       const int32_t* __restrict__ out_ptr,
                                                       const int32_t* __restrict__ out_eid,
                                                       int32_t* __restrict__ choice_eid, float* __restrict__ counts_f) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (gid >= rows * N) return;
  const int64_t r = gid / N, i = gid - r * N;
  const int64_t tb = idx ? idx[r] : r;
  const int64_t t = tb / B, b = tb - t * B;
  const int64_t src = env_minor ? ((t * N + i) * B + b) : ((t * B + b) * N + i);
  if (choice_eid) {
    const uint32_t c = choice[src];
    choice_eid[gid] = (c & SEL_CARRIED) ? -1 : out_eid[out_ptr[i] + (int32_t)c];
  }
  if (counts_f) counts_f[gid] = (float)counts[src];
}

extern "C" int tarl_rollout_gather(const tarl_plan* plan, const uint8_t* choice, const uint8_t* counts, int64_t T,
                                   int64_t B, int env_minor, const int64_t* idx, int64_t rows, int32_t* choice_eid,
                                   float* counts_f, tarl_stream stream) {
  TARL_REQUIRE(plan && T >= 1 && B >= 1 && rows >= 0, "bad arguments");
  TARL_REQUIRE((choice != nullptr) == (choice_eid != nullptr) && (counts != nullptr) == (counts_f != nullptr),
               "each output needs its input");
  TARL_REQUIRE(idx || rows == T * B, "without an index list all T * B rows are converted");
  if (rows == 0 || plan->N == 0) return TARL_OK;
  hipLaunchKernelGGL(k_rollout_gather, dim3((unsigned)ceil_div(rows * plan->N, FB)), dim3(FB), 0, (hipStream_t)stream,
                     choice, counts, B, plan->N, env_minor, idx, rows, plan->out_ptr, plan->out_eid, choice_eid, counts_f);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}
