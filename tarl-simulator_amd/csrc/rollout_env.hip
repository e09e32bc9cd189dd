// rollout_env.hip — the whole T-frame rollout of one environment in ONE workgroup, hot records resident in LDS.
//
// Why a second implementation: the env-minor path (fused.hip) streams every per-(node, environment) record through HBM
// four times per frame and pays four dependent launches per frame; with few environments it is bound by those launches'
// latency (~10 us each), with many by HBM. A CDNA4 CU has 160 KB of LDS — enough for ALL hot state of one environment
// of a few thousand roads (56 B per road: rec0, rec1, the post record, the chosen agent, SELECTED_ROAD, plus the static
// MAX / free-flow / congestion / road-index columns). So: one 1024-thread workgroup per environment loads its records once, runs the T frames with workgroup
// barriers where the env-minor path has kernel boundaries, and writes the records back at the end. HBM then only sees
// the FIFO slot store and the agent table where something actually happens (a few events per frame), and the rollout
// outputs (ENV-MAJOR here: choice / counts [T][B][N], written coalesced by the workgroup). Static topology / tables are
// shared by every environment and stay L2-resident.
//
// The arithmetic, the noise streams (Philox keys, counters, indices) and the order-sensitive rules are those of
// fused.hip's kernels, so the two paths produce identical states, agents, actions, rewards and counts; the per-frame
// log-prob is summed in the same 2^-32 fixed point (bit-identical, order-independent).
//
// Threads own CONTIGUOUS node ranges (consecutive nodes and consecutive CSC positions share Philox blocks), and therefore
// contiguous ranges of in-edges (CSC) and out-edges (CSR). The per-edge statics are packed into 16-byte records once per
// call (k_pack_static) and phases A-C are FLAT loops over the thread's edge range, four records per step: the four
// loads are issued together, so a step costs one L2 round trip instead of a node -> offsets -> edge -> attribute chain
// per edge (measured with clock64: that chain was ~5 000 cycles per road and made a frame ~36 us for 2 500 roads).
#include "fused_common.h"

#define RE_MAX_WAVES 16
#define RE_LAST 0x80000000u
#define RE_NPT_MAX 4        // roads per thread: N <= 4 * block size (the LDS budget allows < 3 * 1024 roads anyway)

// Inside the workgroup the records keep an unpacked fp32 form (r0 = {head_id, head_dep, n, tail_id}, r1 = {head_arr,
// code}); code packs the pending-garbage count g (or -1) and the ring-buffer head offset as an exact fp32 integer. They
// are converted from / to the packed HBM words of fused_common.h when the rollout starts / ends.
__device__ __forceinline__ float l1_code(float g, int hoff) { return (g + 1.0f) * 1024.0f + (float)hoff; }
__device__ __forceinline__ int l1_hoff(float code) { return ((int)code) & 1023; }
__device__ __forceinline__ float l1_g(float code) { return (float)(((int)code) >> 10) - 1.0f; }

struct EnvPlanPtrs {
  const int32_t* in_ptr;
  const int32_t* in_src;
  const int32_t* in_eid;
  const int32_t* out_ptr;
  const int32_t* out_dst;
  const int32_t* out_eid;
  const int32_t* group_of_node;
};

// in-edge k (CSC order): Direction gather
struct __align__(16) InEdge {
  int32_t src;      // upstream road
  uint32_t dst;     // this road; RE_LAST set on the last in-edge of the road
  float ea, lea;    // turn probability and its log (host-evaluated, as in the other paths)
};
// out-edge k (CSR order): choice walk + Response test
struct __align__(16) OutEdge {
  int32_t src;      // this road
  uint32_t dst;     // target road; RE_LAST set on the last out-edge of the road
  float thr;        // inverse-CDF threshold of the policy table
  int32_t gi;       // group (source-node rank) of src: index of its uniform draw
};
// what the choice needs of the edge it picked
struct __align__(16) OutPick {
  long long lg;     // log-prob in 2^-32 fixed point
  int32_t eid;      // original edge id
  int32_t rank;     // rank in the source node's CSR list (the action byte)
};

__global__ __launch_bounds__(256) void k_pack_static(EnvPlanPtrs P, int64_t N, const float* __restrict__ edge_attr,
                                                     const float* __restrict__ log_edge_attr,
                                                     const float* __restrict__ thr, const long long* __restrict__ lgt,
                                                     InEdge* __restrict__ ie, OutEdge* __restrict__ oe,
                                                     OutPick* __restrict__ op) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int32_t a1 = P.in_ptr[i + 1];
  for (int32_t k = P.in_ptr[i]; k < a1; ++k) {
    const int32_t e = P.in_eid[k];
    ie[k] = InEdge{P.in_src[k], (uint32_t)i | (k == a1 - 1 ? RE_LAST : 0u), edge_attr[e], log_edge_attr[e]};
  }
  const int32_t b1 = P.out_ptr[i + 1];
  const int32_t gi = P.group_of_node[i];
  for (int32_t k = P.out_ptr[i]; k < b1; ++k) {
    oe[k] = OutEdge{(int32_t)i, (uint32_t)P.out_dst[k] | (k == b1 - 1 ? RE_LAST : 0u), thr[k], gi};
    op[k] = OutPick{lgt[k], P.out_eid[k], k - P.out_ptr[i]};
  }
}

struct EnvOut {
  uint8_t* choice;   // [T][B][N] or NULL: rank of the chosen out-edge (bit 7: nothing drawn)
  float* log_prob;   // [T][B] or NULL
  float* entropy;    // [T][B] or NULL
  float* reward;     // [T][B] or NULL
  uint8_t* counts;   // [T][B][N] or NULL
  int32_t m_env;     // environments 0 .. m_env-1 keep the per-node series
  float* dtt_node;   // [T][m_env][N] or NULL
  uint8_t* events;   // [T][m_env][N] or NULL: bit 0 popped, bit 1 withdrawn
  int32_t* leg;      // [T][B][2] or NULL: {departed, arrived}
};

struct EnvPlan {
  const int32_t* in_ptr;
  const int32_t* in_src;
  const int32_t* in_eid;
  const int32_t* out_ptr;
  const int32_t* out_dst;
  const int32_t* out_eid;
  const int32_t* group_of_node;
  int64_t N, E, G;
  const InEdge* ie;     // [E] packed in-edge records (CSC order)
  const OutEdge* oe;    // [E] packed out-edge records (CSR order)
  const OutPick* op;    // [E] id / log-prob of each out-edge (read only for the edge a road picks)
};

template <int WAVES>
__device__ __forceinline__ float re_block_sum_f(float v, float* s_red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  __syncthreads();
  if (lane == 0) s_red[wid] = v;
  __syncthreads();
  float tot = 0.0f;
  for (int w = 0; w < WAVES; ++w) tot += s_red[w];
  return tot;
}

template <int WAVES>
__device__ __forceinline__ long long re_block_sum_ll(long long v, long long* s_red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  __syncthreads();
  if (lane == 0) s_red[wid] = v;
  __syncthreads();
  long long tot = 0;
  for (int w = 0; w < WAVES; ++w) tot += s_red[w];
  return tot;
}

// RE_THREADS: 1024 for graphs of thousands of roads, 256 for small ones (several environments share a CU then)
template <int RE_THREADS>
__global__ __launch_bounds__(RE_THREADS) void k_rollout_env(EnvPlan P, int64_t B, int Nmax, FusedBufs fb,
                                                            const float* __restrict__ thr,
                                                            const long long* __restrict__ lgt,
                                                            const float* __restrict__ entropy1,
                                                            const float* __restrict__ edge_attr,
                                                            const float* __restrict__ log_edge_attr, float log_eps,
                                                            int use_cong, uint64_t pseed, uint64_t pcounter0,
                                                            uint64_t seed, uint64_t counter0, int64_t T,
                                                            const float* __restrict__ times, float prev_time,
                                                            float* __restrict__ ag,
                                                            int64_t A, int64_t a_bstride,
                                                            int32_t* __restrict__ scratch, EnvOut out) {
  extern __shared__ float4 re_lds[];
  const int64_t N = P.N;
  float4* r0 = re_lds;                            // [N] {head_id, head_dep, n, tail_id}
  float2* r1 = (float2*)(r0 + N);                 // [N] {head_arr, code}
  float2* pA = r1 + N;                            // [N] {n', tail'}
  float* who_l = (float*)(pA + N);                // [N] chosen agent of the Direction update
  float* sel_l = who_l + N;                       // [N] SELECTED_ROAD
  float* maxn_l = sel_l + N;                      // [N] static MAX_NUMBER_OF_AGENT
  float* ff_l = maxn_l + N;                       // [N] static FREE_FLOW_TIME_TRAVEL
  float* cong_l = ff_l + N;                       // [N] static congestion constant
  float* road_l = cong_l + N;                     // [N] static ROAD_INDEX
  int32_t* s_un_agent = (int32_t*)(road_l + N);   // [INS_CAP]
  int32_t* s_un_road = s_un_agent + INS_CAP;      // [INS_CAP]
  constexpr int RE_WAVES = RE_THREADS / 64;
  __shared__ float s_red_f[RE_MAX_WAVES];
  __shared__ long long s_red_ll[RE_MAX_WAVES];
  __shared__ int32_t s_wave[RE_MAX_WAVES];
  __shared__ int32_t s_cnt, s_lo, s_bad, s_cur, s_dep, s_arr;

  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int32_t npt = (int32_t)((N + RE_THREADS - 1) / RE_THREADS);
  const int32_t i0 = tid * npt < N ? tid * npt : (int32_t)N;
  const int32_t i1 = (i0 + npt < N) ? i0 + npt : (int32_t)N;
  float* agb = ag + b * a_bstride;
  int32_t* cand_agent = scratch + b * 2 * A;
  int32_t* cand_road = cand_agent + A;

  uint32_t prevcode[RE_NPT_MAX];   // SELECTED_ROAD code of each own road in the previous frame (carried when nothing is drawn)
#pragma unroll
  for (int z = 0; z < RE_NPT_MAX; ++z) prevcode[z] = SEL_RAW;
  for (int32_t i = i0; i < i1; ++i) {
    const int64_t row = (int64_t)i * B + b;
    const uint2 hp = fb.hdp[row];
    const uint32_t tlw = fb.tl[row];
    const uint32_t gcode = fb.gc8[row];
    const uint32_t pw = fb.post[row];
    const int n = (int)(hp.x & HD_CNT);   // (HD_DIRTY, bit 7, is re-derived from the store after the rollout: tarl_fused_dead_slots)
    // an idle empty row's head arrived at the previous frame's clock and departs tt0 later (the frame kernels do not store
    // that departure); its pending garbage count is its count
    const bool lazy_row = n == 0 && !(tlw & TLF_AUTH);
    const float dep0 = lazy_row ? prev_time + entry_tt(fb.st0[i], 0.0f) : __uint_as_float(hp.y);
    r0[i] = make_float4((float)(hp.x >> 8), dep0, (float)n, (float)(tlw >> 8));
    const float arr = head_arrival(fb.slots, fb.lds, fb.gc8, row, hp.x, tlw, Nmax, prev_time);
    r1[i] = make_float2(arr, l1_code((float)pending_g(tlw, n, gcode, Nmax), tl_hoff(tlw)));
    const bool arrived = (pw & PF_ARRIVED) != 0u;
    pA[i] = make_float2(arrived ? (float)(n + 1) : (float)n, (float)(pw >> 8));
    who_l[i] = arrived ? (float)(pw >> 8) : 0.0f;
    sel_l[i] = sel_value(fb, P.out_ptr, P.out_dst, i, row);
#pragma unroll
    for (int z = 0; z < RE_NPT_MAX; ++z) prevcode[z] = (z == i - i0) ? (uint32_t)(fb.sel8[row] & 0x7Fu) : prevcode[z];
    const float4 st = fb.st0[i];
    maxn_l[i] = st.x;
    ff_l[i] = st.y;
    road_l[i] = st.z;
    cong_l[i] = st.w;
  }
  // the thread's contiguous edge ranges (fixed for the whole rollout)
  const int32_t ka0 = P.in_ptr[i0], ka1 = P.in_ptr[i1];
  const int32_t kb0 = P.out_ptr[i0], kb1 = P.out_ptr[i1];
  if (tid == 0) s_cur = fb.cur_lo ? fb.cur_lo[b] : 0;
  __syncthreads();

  for (int64_t f = 0; f < T; ++f) {
    const float t = times[f];
    const uint64_t pcounter = pcounter0 + (uint64_t)f, counter = counter0 + (uint64_t)f;

    // ---- A. choice (k_fused_choice): GraphDistribution.sample + log_prob through the policy tables ---------------------
    // Flat walk over the thread's out-edge records, four per step (loads issued together); a road's scan ends at its
    // RE_LAST record. Roads without out-edges have no records: their action stays -1 (pre-filled).
    long long lp = 0;
    bool bad = false;
    {
      PhiloxRun rng;
      bool found = false, fresh = true;
      float u = 0.0f;
      int32_t pick[RE_NPT_MAX];   // CSR position picked by each own road (-1: none); the LDS size caps roads/thread
#pragma unroll
      for (int z = 0; z < RE_NPT_MAX; ++z) pick[z] = -1;
      for (int32_t k4 = kb0; k4 < kb1; k4 += 4) {
        OutEdge rr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rr[q] = P.oe[(k4 + q < kb1) ? k4 + q : kb1 - 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int32_t k = k4 + q;
          if (k < kb1) {
            const OutEdge r = rr[q];
            if (fresh) {   // first out-edge of a road: draw its uniform
              u = rng.uniform(pseed, pcounter, (uint64_t)((fb.env_base + b) * P.G + r.gi));
              found = false;
              fresh = false;
            }
            if (!found && u < r.thr) {   // first out-edge (plan order) whose threshold exceeds u
              found = true;
              sel_l[r.src] = (float)(r.dst & ~RE_LAST);
              const int32_t slot = r.src - i0;
#pragma unroll
              for (int z = 0; z < RE_NPT_MAX; ++z) pick[z] = (z == slot) ? k : pick[z];
            }
            if (r.dst & RE_LAST) {
              if (!found) bad = true;   // the road keeps its previous SELECTED_ROAD; the action is infeasible
              fresh = true;
            }
          }
        }
      }
      OutPick pk[RE_NPT_MAX];   // id / log-prob of the picked edges: independent loads, one round trip
#pragma unroll
      for (int z = 0; z < RE_NPT_MAX; ++z) pk[z] = P.op[pick[z] >= 0 ? pick[z] : 0];
#pragma unroll
      for (int z = 0; z < RE_NPT_MAX; ++z) {
        const int32_t i = i0 + z;
        if (i < i1) {
          if (pick[z] >= 0) lp += pk[z].lg;
          const uint32_t code = pick[z] >= 0 ? (uint32_t)pk[z].rank : (prevcode[z] | SEL_CARRIED);
          prevcode[z] = code & 0x7Fu;
          if (out.choice) __builtin_nontemporal_store((uint8_t)code, &out.choice[(f * B + b) * N + i]);
        }
      }
    }
    if (tid == 0) {
      s_cnt = 0;
      s_lo = 0x7fffffff;
      s_bad = 0;
      s_dep = 0;
      s_arr = 0;
    }
    __syncthreads();
    if (bad) atomicOr(&s_bad, 1);

    // ---- B. Direction gather (k_fused_direction) -------------------------------------------------------------------------
    // Pass 1: a flat walk over the thread's in-edge records (four per step): admissibility and summed turn probability
    // per road, no random numbers; default post record (nobody chosen). Roads with an admissible in-edge (P > 0, a few
    // percent) go to an LDS list (the insert phase's candidate array is free here). Pass 2 runs the Gumbel race for the
    // listed roads densely — same noise indices and expressions: identical to racing everywhere.
    {
      for (int32_t i = i0; i < i1; ++i) {
        const float4 me = r0[i];
        pA[i] = make_float2(me.z, me.w);
        who_l[i] = 0.0f;
        if (out.dtt_node && b < out.m_env) {   // delta_travel_time of this road's out-edges (src/direction_mpnn.py:94-96)
          const float d = (me.y - r1[i].x) - ff_l[i];
          out.dtt_node[(f * out.m_env + b) * N + i] = d > 0.0f ? d : (d != d ? d : 0.0f);
        }
      }
      float Psum = 0.0f;
      for (int32_t k4 = ka0; k4 < ka1; k4 += 4) {
        InEdge rr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rr[q] = P.ie[(k4 + q < ka1) ? k4 + q : ka1 - 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (k4 + q < ka1) {
            const InEdge r = rr[q];
            const int32_t i = (int32_t)(r.dst & ~RE_LAST);
            const float4 me = r0[i];
            const float max_i = maxn_l[i], n_i = me.z, road_i = road_l[i];
            const float4 rj = r0[r.src];
            const float dep = rj.y, n_j = rj.z, max_j = maxn_l[r.src];
            const bool heads_here = sel_l[r.src] == road_i;
            const bool m1 = (dep <= t) && (n_i < max_i - TARL_CONGESTION_FILE) && heads_here && (n_j > 0.0f);
            const bool m2 = ((dep - t) < -10.0f) && ((max_j - TARL_CONGESTION_FILE) <= n_j) &&
                            ((max_j - n_j) <= (max_i - n_i)) && heads_here;
            Psum = Psum + r.ea * ((m1 || m2) ? 1.0f : 0.0f);
            if (r.dst & RE_LAST) {
              if (Psum > 0.0f) {
                const int32_t pos = atomicAdd(&s_cnt, 1);
                if (pos < INS_CAP) s_un_agent[pos] = i; else atomicOr(&s_bad, 2);   // overflow: pass 2 over all roads
              }
              Psum = 0.0f;
            }
          }
        }
      }
    }
    __syncthreads();
    {
      const bool overflow = (s_bad & 2) != 0;
      const int32_t cnt = overflow ? (int32_t)N : s_cnt;
      for (int32_t idx = tid; idx < cnt; idx += RE_THREADS) {
        const int32_t i = overflow ? idx : s_un_agent[idx];
        const float4 me = r0[i];
        const float max_i = maxn_l[i], n_i = me.z, road_i = road_l[i];
        const float room_i = max_i - n_i;
        const bool has_room = n_i < max_i - TARL_CONGESTION_FILE;
        float Psum = 0.0f, best = -FLT_MAX, best_id = 0.0f;
        PhiloxRun rng;
        const int32_t k1 = P.in_ptr[i + 1];
        for (int32_t k = P.in_ptr[i]; k < k1; ++k) {
          const InEdge r = P.ie[k];
          const float4 rj = r0[r.src];
          const float sel_j = sel_l[r.src];
          const float id = rj.x, dep = rj.y, n_j = rj.z, max_j = maxn_l[r.src];
          const bool heads_here = sel_j == road_i;
          const bool m1 = (dep <= t) && has_room && heads_here && (n_j > 0.0f);
          const bool m2 = ((dep - t) < -10.0f) && ((max_j - TARL_CONGESTION_FILE) <= n_j) && ((max_j - n_j) <= room_i) &&
                          heads_here;
          const bool m = m1 || m2;
          const float prob = r.ea * (m ? 1.0f : 0.0f);
          Psum = Psum + prob;
          const float uu = rng.uniform(seed, counter, (uint64_t)((fb.env_base + b) * P.E + k));
          const float g = gumbel_from_u01(uu);
          const float score = (m ? r.lea : log_eps) + g;
          if (score > best) {
            best = score;
            best_id = id;
          }
        }
        const float who = (Psum > 0.0f) ? best_id : 0.0f;
        pA[i] = make_float2(who != 0.0f ? n_i + 1.0f : n_i, who != 0.0f ? who : me.w);
        who_l[i] = who;
      }
    }
    __syncthreads();
    if (tid == 0) s_cnt = 0;   // the list is the insert phase's candidate counter again

    // ---- C. row pass (k_fused_rows): Direction update + Response pop + withdraw ------------------------------------------
    // Response test first: a flat walk over the thread's out-edge records (four per step) sets one bit per own road.
    unsigned long long popbits = 0ull;   // the thread owns at most 64 roads (host check)
    for (int32_t k4 = kb0; k4 < kb1; k4 += 4) {
      OutEdge rr[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) rr[q] = P.oe[(k4 + q < kb1) ? k4 + q : kb1 - 1];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (k4 + q < kb1) {
          const OutEdge r = rr[q];
          const float4 q0 = r0[r.src];
          const long long head = (long long)((q0.z == 0.0f) ? who_l[r.src] : q0.x);   // head after the Direction update
          const bool up = (long long)pA[r.src].x > 0;
          const float2 pj = pA[r.dst & ~RE_LAST];
          if (up && (long long)pj.x > 0 && (long long)pj.y == head) popbits |= 1ull << (r.src - i0);
        }
      }
    }
    for (int32_t i = i0; i < i1; ++i) {
      const int64_t row = (int64_t)i * B + b;
      float* sl = fb.slots + row * fb.lds;
      const float2 pa = pA[i];
      const float who = who_l[i];
      const float4 q0 = r0[i];
      const float2 q1 = r1[i];
      const float4 st = make_float4(maxn_l[i], ff_l[i], road_l[i], cong_l[i]);
      const float n0 = q0.z;
      const bool pop = ((popbits >> (i - i0)) & 1ull) != 0ull;
      int hoff = l1_hoff(q1.y);
      const int q = (int)n0;
      const float t_cong = st.w / (st.x + 10.0f - n0);
      const float tt = (t_cong != t_cong) ? t_cong : fmaxf(st.y, t_cong);
      const float dep_new = t + tt;
      const bool lazy = (who == 0.0f) && (q >= 0) && (q < Nmax - 1);
      if ((int)pa.x >= Nmax) atomicOr(fb.flags, FLAG_COUNT_AT_NMAX);
      if (!lazy && q >= 0 && q < Nmax) {
        slot_store(sl + SLW * phys(hoff, q, Nmax), who, t, dep_new);
      }
      float n = pa.x;
      float head_id = (n0 == 0.0f) ? who : q0.x;
      float head_dep = (n0 == 0.0f) ? dep_new : q0.y;
      float head_arr = (n0 == 0.0f) ? t : q1.x;
      float tail_id = pa.y;
      int shift = 0;
      if (pop) {
        const SlotRec last = slot_load(sl + SLW * phys(hoff, Nmax - 1, Nmax));
        slot_store(sl + SLW * hoff, last.id, last.arr, last.dep);
        hoff = phys(hoff, 1, Nmax);
        shift = 1;
        n = n - 1.0f;
      }
      int c = 0;
      if (n > 0.0f) {
        int32_t w0 = -1, w1 = 0;   // out-list of this row's road: fetched only when a head is actually due
        for (int sx = 0; sx < Nmax && (float)sx < n; ++sx) {
          float idf, depf;
          if (sx == 0 && shift == 0) {
            idf = head_id;
            depf = head_dep;
          } else {
            const SlotRec rd = slot_load(sl + SLW * phys(hoff, sx, Nmax));
            idf = rd.id;
            depf = rd.dep;
          }
          const long long id = (long long)idf;
          if (id < 0 || id >= A) break;
          if (!(depf <= t)) break;
          if (w0 < 0) {
            const long long road = (long long)st.z;
            w0 = 0;
            if (road >= 0 && road < N) {
              w0 = P.out_ptr[road];
              w1 = P.out_ptr[road + 1];
            }
          }
          const long long dest = (long long)fb.a_dest[b * A + id];
          bool conn = false;
          for (int32_t k = w0; k < w1; ++k) conn = conn || ((long long)P.out_dst[k] == dest);
          if (!conn) break;
          float* a = agb + id * AG_COLS;
          a[AG_DONE] = 1.0f;
          a[AG_ON_WAY] = 0.0f;
          a[AG_ARR] = t;
          fb.a_status[b * A + id] = 2;
          ++c;
        }
      }
      for (int k = 0; k < c; ++k) {
        slot_store(sl + SLW * phys(hoff, k, Nmax), 0.0f, 0.0f, 0.0f);
      }
      if (c > 0) {
        hoff = phys(hoff, c, Nmax);
        n = n - (float)c;
      }
      if (shift + c > 0) {
        if (lazy && n == 0.0f) {
          head_id = 0.0f;
          head_arr = t;
          head_dep = dep_new;
        } else {
          const SlotRec hd = slot_load(sl + SLW * hoff);
          head_id = hd.id;
          head_arr = hd.arr;
          head_dep = hd.dep;
        }
        const int qn = (int)n;
        tail_id = (qn >= 1 && qn <= Nmax) ? sl[SLW * phys(hoff, qn - 1, Nmax)] : 0.0f;
      }
      r0[i] = make_float4(head_id, head_dep, n, tail_id);
      r1[i] = make_float2(head_arr, l1_code(lazy ? n0 : -1.0f, hoff));
      if (out.events && b < out.m_env)
        out.events[(f * out.m_env + b) * N + i] = (uint8_t)((pop ? 1 : 0) | (c > 0 ? 2 : 0));
      if (c > 0) atomicAdd(&s_arr, c);
    }
    __threadfence_block();   // the slot-store writes above are read by this workgroup's insert / later frames
    __syncthreads();

    // ---- D. insert (k_fused_insert) -----------------------------------------------------------------------------------------
    {
      auto target = [&](int32_t origin, int32_t* road) -> bool {
        if (origin < 0 || origin >= N) return false;
        const long long r = (long long)sel_l[origin];
        if (r < 0 || r >= N) return false;
        const long long room = (long long)(maxn_l[r] - TARL_CONGESTION_FILE - r0[r].z);
        *road = (int32_t)r;
        return room > 0;
      };
      if (fb.a_order) {
        const int32_t* ord = fb.a_order + b * A;
        const float* dsort = fb.a_dep_sorted + b * A;
        const int32_t lo = s_cur;
        for (int64_t k0 = lo; k0 < A; k0 += RE_THREADS) {
          const int64_t k = k0 + tid;
          bool notdue = false;
          if (k < A) {
            const bool due = dsort[k] <= t;
            notdue = !due;
            if (!due) {
              atomicMin(&s_lo, (int32_t)k);
            } else {
              const int32_t a = ord[k];
              if (fb.a_status[b * A + a] == 0) {
                atomicMin(&s_lo, (int32_t)k);
                int32_t road = 0;
                if (target(fb.a_origin[b * A + a], &road)) {
                  const int32_t pos = atomicAdd(&s_cnt, 1);
                  if (pos < INS_CAP) {
                    s_un_agent[pos] = a;
                    s_un_road[pos] = road;
                  }
                }
              }
            }
          }
          if (__syncthreads_or(notdue ? 1 : 0)) break;
        }
        __syncthreads();
        if (tid == 0) s_cur = s_lo == 0x7fffffff ? (int32_t)A : s_lo;   // read again only after the next barriers
      } else {
        for (int64_t a = tid; a < A; a += RE_THREADS) {
          if (fb.a_status[b * A + a] == 0 && fb.a_dep[b * A + a] <= t) {
            int32_t road = 0;
            if (target(fb.a_origin[b * A + a], &road)) {
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
        for (int32_t idx = tid; idx < Lc; idx += RE_THREADS) {
          const int32_t a = s_un_agent[idx];
          int32_t pos = 0;
          for (int32_t k = 0; k < Lc; ++k) pos += (s_un_agent[k] < a) ? 1 : 0;
          cand_agent[pos] = a;
          cand_road[pos] = s_un_road[idx];
        }
        __threadfence_block();
        __syncthreads();
      } else {   // backlog beyond the LDS list: ordered ballot compaction of all ready agents into the global scratch
        int32_t basec = 0;
        for (int64_t a0 = 0; a0 < A; a0 += RE_THREADS) {
          const int64_t a = a0 + tid;
          bool cnd = false;
          int32_t road = 0;
          if (a < A && fb.a_status[b * A + a] == 0 && fb.a_dep[b * A + a] <= t) cnd = target(fb.a_origin[b * A + a], &road);
          const unsigned long long bal = __ballot(cnd);
          const int lane_off = __popcll(bal & ((1ull << lane) - 1ull));
          if (lane == 0) s_wave[wid] = __popcll(bal);
          __syncthreads();
          int32_t wbase = 0, tot = 0;
          for (int w = 0; w < RE_WAVES; ++w) {
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
      // rank within road (stable), admit the first min(count, capacity), write slots / hot records
      for (int32_t idx = tid; idx < Lc; idx += RE_THREADS) {
        const int32_t r = cand_road[idx];
        const int32_t a = cand_agent[idx];
        int32_t rank = 0, total = 0;
        for (int32_t k = 0; k < Lc; ++k) {
          const bool same = cand_road[k] == r;
          total += same ? 1 : 0;
          rank += (same && k < idx) ? 1 : 0;
        }
        const int64_t rrow = (int64_t)r * B + b;
        const float4 str = make_float4(maxn_l[r], ff_l[r], road_l[r], cong_l[r]);
        const float n0 = r0[r].z;
        const long long cap = (long long)(str.x - TARL_CONGESTION_FILE - n0);
        int32_t commit = 0;
        if (rank < cap) {
          const long long m = total < cap ? total : cap;
          const long long slot = (long long)n0 + rank;
          const float t_cong = use_cong ? str.w / (str.x + 10.0f - (float)(long long)n0) : 0.0f;
          const float tt = (t_cong != t_cong) ? t_cong : fmaxf(str.y, t_cong);
          const float code = r1[r].y;   // nobody writes r1.y before the barrier below
          if (slot >= 0 && slot < Nmax) {
            slot_store(fb.slots + rrow * fb.lds + SLW * phys(l1_hoff(code), (int)slot, Nmax), (float)a, t, t + tt);
          }
          agb[(int64_t)a * AG_COLS + AG_ON_WAY] = 1.0f;
          fb.a_status[b * A + a] = 1;
          if (fb.a_ins) fb.a_ins[b * A + fb.a_rank[b * A + a]] = 1;
          if (rank == 0 && n0 == 0.0f) {
            r0[r].x = (float)a;
            r0[r].y = t + tt;
            r1[r].x = t;
          }
          if (rank == m - 1) r0[r].w = (float)a;  // new tail
          if (rank == 0) commit = (int32_t)m;
        }
        cand_agent[idx] = commit;
      }
      __threadfence_block();
      __syncthreads();
      for (int32_t idx = tid; idx < Lc; idx += RE_THREADS) {
        const int32_t cmt = cand_agent[idx];
        if (cmt > 0) {
          const int32_t r = cand_road[idx];
          r1[r].y = l1_code(-1.0f, l1_hoff(r1[r].y));
          r0[r].z = r0[r].z + (float)cmt;
          atomicAdd(&s_dep, cmt);
        }
      }
      __threadfence_block();
      __syncthreads();
    }

    // ---- E. frame outputs: counts (critic observation), reward = -sum of counts, log-prob, entropy ------------------------
    float nsum = 0.0f;
    for (int32_t i = i0; i < i1; ++i) {
      const float n = r0[i].z;
      if (out.counts) __builtin_nontemporal_store((uint8_t)n, &out.counts[(f * B + b) * N + i]);
      nsum += n;
    }
    const float ntot = re_block_sum_f<RE_WAVES>(nsum, s_red_f);          // sums of small integers: exact in fp32 in any order
    const long long lptot = re_block_sum_ll<RE_WAVES>(lp, s_red_ll);
    if (tid == 0) {
      if (out.reward) out.reward[f * B + b] = -ntot;
      if (out.log_prob) out.log_prob[f * B + b] = (s_bad & 1) ? -INFINITY : (float)((double)lptot / LP_FIX);
      if (out.entropy) out.entropy[f * B + b] = entropy1[0];
      if (out.leg) {
        out.leg[(f * B + b) * 2 + 0] = s_dep;
        out.leg[(f * B + b) * 2 + 1] = s_arr;
      }
    }
    __syncthreads();
  }

  for (int32_t i = i0; i < i1; ++i) {
    const int64_t row = (int64_t)i * B + b;
    const float4 q0 = r0[i];
    const float2 q1 = r1[i];
    fb.hdp[row] = make_uint2(((uint32_t)q0.x << 8) | (uint32_t)q0.z, __float_as_uint(q0.y));
    fb.tl[row] = tl_word((uint32_t)q0.w, l1_hoff(q1.y), TLF_AUTH);
    fb.gc8[row] = (uint8_t)r1_code((int)l1_g(q1.y));
    const float who = who_l[i];
    fb.post[row] = ((uint32_t)pA[i].y << 8) | (pA[i].x > 0.0f ? PF_NONEMPTY : 0u) | (who != 0.0f ? PF_ARRIVED : 0u);
    // SELECTED_ROAD back as a rank of this road's out-list (the raw value where it names none of them, as pack does)
    const float sv = sel_l[i];
    uint32_t code = SEL_RAW;
    const int32_t k0 = P.out_ptr[i], k1 = P.out_ptr[i + 1];
    for (int32_t k = k1 - 1; k >= k0; --k)
      if ((float)P.out_dst[k] == sv && k - k0 < (int32_t)SEL_RAW) code = (uint32_t)(k - k0);
    fb.sel8[row] = (uint8_t)code;
    fb.sel[row] = sv;
  }
  if (tid == 0 && fb.cur_lo) fb.cur_lo[b] = s_cur;
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static size_t re_lds_bytes(int64_t N) { return (size_t)N * 56 + (size_t)INS_CAP * 8; }

extern "C" int tarl_rollout_env_supported(const tarl_plan* plan) {
  return plan && re_lds_bytes(plan->N) + 1024 <= 160 * 1024 ? 1 : 0;
}

// device scratch of tarl_rollout_env: the packed per-edge static records (rebuilt by every call)
extern "C" int64_t tarl_rollout_env_scratch_bytes(const tarl_plan* plan) {
  return plan ? (int64_t)(sizeof(InEdge) + sizeof(OutEdge) + sizeof(OutPick)) * (plan->E > 0 ? plan->E : 1) : -1;
}

extern "C" int tarl_rollout_env(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax, int64_t T,
                                const float* times_dev, float prev_time, const float* thresholds,
                                const int64_t* log_probs, const float* entropy1, uint64_t policy_seed,
                                uint64_t policy_counter0, float* agent_features, int64_t A, int64_t a_bstride,
                                const float* edge_attr, const float* log_edge_attr, float log_eps, int use_cong,
                                uint64_t seed, uint64_t counter0, int32_t* ins_scratch, void* static_scratch,
                                uint8_t* choice, float* log_prob, float* entropy, float* reward, uint8_t* counts,
                                int32_t metrics_envs, float* dtt_node, uint8_t* events, int32_t* leg,
                                tarl_stream stream) {
  int rc = tarl_check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(T >= 1 && times_dev, "bad frame count / times");
  TARL_REQUIRE(thresholds && log_probs && entropy1, "policy tables missing (call tarl_fused_policy_prepare)");
  TARL_REQUIRE(agent_features && A >= 1 && ins_scratch && static_scratch, "agents / scratch missing");
  TARL_REQUIRE(((uintptr_t)static_scratch & 15) == 0, "static_scratch must be 16-byte aligned");
  TARL_REQUIRE(f->a_origin && f->a_dest && f->a_dep && f->a_status, "fused agent buffers missing");
  TARL_REQUIRE(f->a_order == nullptr || (f->cur_lo != nullptr && f->a_dep_sorted != nullptr),
               "a_order needs cur_lo and a_dep_sorted");
  TARL_REQUIRE(B == 1 || a_bstride >= A * AG_COLS, "agent stride smaller than one population");
  TARL_REQUIRE(plan->E == 0 || (edge_attr && log_edge_attr), "edge constants missing");
  TARL_REQUIRE(plan->N <= (plan->N <= 512 ? 256 : 1024) * RE_NPT_MAX, "too many roads per thread");
  TARL_REQUIRE(tarl_rollout_env_supported(plan), "graph too large for the LDS-resident rollout (tarl_rollout_env_supported)");
  TARL_REQUIRE(B < 65536ll * 32768ll, "too many environments for one launch");
  TARL_REQUIRE(metrics_envs >= 0 && metrics_envs <= B, "metrics_envs out of range");
  TARL_REQUIRE(metrics_envs > 0 || (!dtt_node && !events), "per-node series need metrics_envs > 0");
  if (plan->N == 0) return TARL_OK;
  const size_t lds = re_lds_bytes(plan->N);
  const int64_t Ecap = plan->E > 0 ? plan->E : 1;
  InEdge* ie = (InEdge*)static_scratch;
  OutEdge* oe = (OutEdge*)(ie + Ecap);
  OutPick* op = (OutPick*)(oe + Ecap);
  const EnvPlanPtrs PP{plan->in_ptr, plan->in_src, plan->in_eid, plan->out_ptr, plan->out_dst, plan->out_eid,
                       plan->group_of_node};
  hipLaunchKernelGGL(k_pack_static, dim3((unsigned)ceil_div(plan->N, 256)), dim3(256), 0, (hipStream_t)stream, PP,
                     plan->N, edge_attr, log_edge_attr, thresholds, (const long long*)log_probs, ie, oe, op);
  TARL_LAUNCH_CHECK();
  const EnvPlan P{plan->in_ptr, plan->in_src, plan->in_eid, plan->out_ptr, plan->out_dst, plan->out_eid,
                  plan->group_of_node, plan->N, plan->E, plan->G, ie, oe, op};
  const EnvOut out{choice, log_prob, entropy, reward, counts, metrics_envs, dtt_node, events, leg};
  // this kernel keeps the reference's slot-by-slot bookkeeping of the dead slots: the zeros the frame kernels' clean rows
  // stand for (fused_common.h: HD_DIRTY) go into the store first, and the rows' flags are derived from the store afterwards
  int rc_d = tarl_fused_dead_slots(plan, f, B, Nmax, 1, stream);
  if (rc_d) return rc_d;
  if (plan->N <= 512) {
    hipLaunchKernelGGL(k_rollout_env<256>, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, P, B, (int)Nmax,
                       tarl_to_bufs(f), thresholds, (const long long*)log_probs, entropy1, edge_attr, log_edge_attr, log_eps, use_cong,
                       policy_seed, policy_counter0, seed, counter0, T, times_dev, prev_time, agent_features, A,
                       a_bstride, ins_scratch, out);
  } else {
    // the opt-in to > 64 KB of dynamic LDS is a per-device attribute of the function
    static size_t lds_set[64] = {0};
    int devid = 0;
    TARL_CHECK_HIP(hipGetDevice(&devid));
    size_t* set = &lds_set[devid & 63];
    if (lds > 64 * 1024 && lds > *set) {
      TARL_CHECK_HIP(hipFuncSetAttribute((const void*)k_rollout_env<1024>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds));
      *set = lds;
    }
    hipLaunchKernelGGL(k_rollout_env<1024>, dim3((unsigned)B), dim3(1024), lds, (hipStream_t)stream, P, B, (int)Nmax,
                       tarl_to_bufs(f), thresholds, (const long long*)log_probs, entropy1, edge_attr, log_edge_attr, log_eps, use_cong,
                       policy_seed, policy_counter0, seed, counter0, T, times_dev, prev_time, agent_features, A,
                       a_bstride, ins_scratch, out);
  }
  TARL_LAUNCH_CHECK();
  return tarl_fused_dead_slots(plan, f, B, Nmax, 0, stream);
}
