// routing.hip — shortest-path routing: the classical "dijkstra" agent's next-hop table
// (src/agents/base.py:527-584) and the policy's free-flow distance prior (src/agents/mpnn_agent.py:53-113).
//
// The reference calls networkx (all_pairs_dijkstra_path / shortest_path_length). Distances are unique, but the PATH —
// and with it the next hop written into SELECTED_ROAD — depends on how that implementation breaks ties: a node's
// predecessor is the first settled node (heap order = (distance, push counter)) that reaches it at its final distance,
// and successors are pushed in adjacency (= edge_index) order. On a regular grid nearly every pair is tied, so the
// table is only reproducible by replaying that order. k_apsp does exactly that, one 64-lane wave per source:
//   * per-node state (tentative distance in double, push counter, first hop) lives in LDS (16 B/node) or, for graphs
//     beyond 10k nodes, in an L2-resident global scratch row;
//   * extract-min = every lane scans N/64 nodes, then a 6-step DPP/shuffle reduction on the (distance, counter) key —
//     the array form of the heap (a node's live heap entry is always its latest push; stale ones pop later and are
//     skipped, so they never influence the order);
//   * the settled node's out-edges are relaxed in CSR (= edge) order with wave-uniform control flow, lane 0 writing.
// All sources run concurrently: N waves of work, N^2 outputs written coalesced at the end of each source.
#include "tarl_common.h"

#include <math.h>
#include <stdlib.h>

#define RT_BLOCK 256
#define APSP_DONE 0xFFFFFFFFu

// ---- per-edge travel time (src/agents/base.py:541-550) ---------------------------------------------------------------
// w[e] = max(ff[u], cong[v] / (max[u] + 10 - n[u])), u = src(e), v = dst(e); fp32, same operation order as torch.
__global__ __launch_bounds__(RT_BLOCK) void k_edge_travel_time(const int32_t* __restrict__ src,
                                                               const int32_t* __restrict__ dst,
                                                               const float* __restrict__ x, Layout L, int64_t B,
                                                               int64_t E, const float* __restrict__ cong,
                                                               float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
  if (gid >= B * E) return;
  const int64_t b = gid / E;
  const int64_t e = gid - b * E;
  const float* xu = x + b * L.bstride + (int64_t)src[e] * L.ldx;
  const float ff = xu[L.col_ff()];
  const float tc = cong[dst[e]] / ((xu[L.col_maxn()] + 10.0f) - xu[L.col_n()]);
  out[gid] = (tc > ff || tc != tc) ? tc : ff;
}

// ---- all-pairs shortest paths with networkx's tie order ----------------------------------------------------------------
template <typename W>
__global__ __launch_bounds__(64) void k_apsp(const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_dst,
                                             const int32_t* __restrict__ out_eid, const W* __restrict__ w,
                                             int64_t w_bstride, int64_t B, int64_t N, uint8_t* __restrict__ scratch,
                                             int64_t* __restrict__ next_hop, float* __restrict__ dist_out) {
  extern __shared__ double apsp_lds[];
  const int lane = threadIdx.x;
  uint8_t* base = scratch ? scratch + (int64_t)blockIdx.x * 16 * N : (uint8_t*)apsp_lds;
  volatile double* seen = (volatile double*)base;
  volatile uint32_t* cnt = (volatile uint32_t*)(base + 8 * N);
  volatile int32_t* hop = (volatile int32_t*)(base + 12 * N);
  const double INF = __longlong_as_double(0x7FF0000000000000ll);

  for (int64_t job = blockIdx.x; job < B * N; job += gridDim.x) {
    const int64_t b = job / N;
    const int32_t s = (int32_t)(job - b * N);
    const W* wb = w + b * w_bstride;
    for (int64_t v = lane; v < N; v += 64) {
      seen[v] = INF;
      cnt[v] = 0u;
      hop[v] = -1;
    }
    __syncthreads();
    if (lane == 0) {
      seen[s] = 0.0;
      cnt[s] = 1u;
      hop[s] = s;
    }
    __syncthreads();
    uint32_t counter = 1u;   // wave-uniform
    for (;;) {
      // extract-min over the unsettled, reached nodes by (distance, push counter)
      double bd = INF;
      uint32_t bc = APSP_DONE;
      int32_t bv = -1;
      for (int64_t v = lane; v < N; v += 64) {
        const uint32_t c = cnt[v];
        const double d = seen[v];
        if (c != APSP_DONE && c != 0u && (bv < 0 || d < bd || (d == bd && c < bc))) {
          bd = d;
          bc = c;
          bv = (int32_t)v;
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const double od = __shfl_xor(bd, off, 64);
        const uint32_t oc = (uint32_t)__shfl_xor((int)bc, off, 64);
        const int32_t ov = __shfl_xor(bv, off, 64);
        if (ov >= 0 && (bv < 0 || od < bd || (od == bd && oc < bc))) {
          bd = od;
          bc = oc;
          bv = ov;
        }
      }
      if (bv < 0) break;   // uniform: the reduction leaves the same triple in every lane
      const int32_t v = bv;
      const double d = bd;
      const int32_t hv = hop[v];
      __syncthreads();
      if (lane == 0) cnt[v] = APSP_DONE;
      const int32_t k1 = out_ptr[v + 1];
      for (int32_t k = out_ptr[v]; k < k1; ++k) {
        const int32_t u = out_dst[k];
        const double vu = d + (double)wb[out_eid[k]];
        __syncthreads();
        if (cnt[u] != APSP_DONE && vu < seen[u]) {   // unreached nodes hold +inf; same value in all lanes
          ++counter;
          __syncthreads();
          if (lane == 0) {
            seen[u] = vu;
            cnt[u] = counter;
            hop[u] = (v == s) ? u : hv;
          }
        }
      }
      __syncthreads();
    }
    // rows of the two tables, coalesced; unreached nodes: -1 / +inf
    for (int64_t v = lane; v < N; v += 64) {
      const bool reached = cnt[v] == APSP_DONE;
      if (next_hop) next_hop[(b * N + s) * N + v] = reached ? (int64_t)hop[v] : -1ll;
      if (dist_out) dist_out[(b * N + s) * N + v] = reached ? (float)seen[v] : __int_as_float(0x7F800000);
    }
    __syncthreads();
  }
}

// ---- SELECTED_ROAD[i] = next_hop[i, DESTINATION(head agent of i)] for every row (src/agents/base.py:572-580) ---------------
__global__ __launch_bounds__(RT_BLOCK) void k_select_next_hop(float* __restrict__ x, Layout L, int64_t B, int64_t N,
                                                              const float* __restrict__ ag, int64_t A,
                                                              int64_t a_bstride, const int64_t* __restrict__ next_hop,
                                                              int64_t nh_bstride) {
  const int64_t gid = (int64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
  if (gid >= B * N) return;
  const int64_t b = gid / N;
  const int64_t i = gid - b * N;
  float* xi = x + b * L.bstride + i * L.ldx;
  const long long head = (long long)xi[0];
  if (head < 0 || head >= A) return;
  const long long dest = (long long)ag[b * a_bstride + head * AG_COLS + AG_DEST];
  if (dest < 0 || dest >= N) return;
  xi[L.col_sel()] = (float)next_hop[b * nh_bstride + i * N + dest];
}

// ---- C ABI ---------------------------------------------------------------------------------------------------------------
extern "C" int tarl_edge_travel_time(const tarl_plan* plan, const float* x, int64_t B, int64_t x_bstride, int64_t ldx,
                                     int32_t Nmax, const float* congestion_constant, float* travel_time,
                                     tarl_stream stream) {
  TARL_REQUIRE(plan && x && congestion_constant && travel_time, "null argument");
  TARL_REQUIRE(B >= 1 && Nmax >= 1 && ldx >= 3 * (int64_t)Nmax + 7, "bad shape");
  if (plan->E == 0) return TARL_OK;
  Layout L{Nmax, ldx, x_bstride};
  hipLaunchKernelGGL(k_edge_travel_time, dim3((unsigned)ceil_div(B * plan->E, RT_BLOCK)), dim3(RT_BLOCK), 0,
                     (hipStream_t)stream, plan->src, plan->dst, x, L, B, plan->E, congestion_constant, travel_time);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

// largest per-wave state kept in LDS; TARL_APSP_LDS_MAX (bytes) lowers it (test hook for the global-scratch variant)
static int64_t apsp_lds_max() {
  const char* e = getenv("TARL_APSP_LDS_MAX");
  const int64_t cap = 160 * 1024;
  if (!e) return cap;
  const long long v = atoll(e);
  return v < 0 ? 0 : (v > cap ? cap : (int64_t)v);
}

extern "C" int64_t tarl_apsp_scratch_bytes(const tarl_plan* plan, int64_t B) {
  if (!plan || B < 1) return -1;
  if (16 * plan->N <= apsp_lds_max()) return 0;            // state fits the CU's LDS
  int64_t waves = B * plan->N;
  if (waves > 4096) waves = 4096;
  return waves * 16 * plan->N;
}

template <typename W>
static int apsp_launch(const tarl_plan* plan, const W* weights, int64_t B, int64_t w_bstride, void* scratch,
                       int64_t scratch_bytes, int64_t* next_hop, float* dist, tarl_stream stream) {
  TARL_REQUIRE(plan && weights, "null argument");
  TARL_REQUIRE(B >= 1 && (w_bstride == 0 || w_bstride >= plan->E), "bad shape");
  TARL_REQUIRE(next_hop || dist, "no output requested");
  const int64_t N = plan->N;
  if (N == 0) return TARL_OK;
  const int64_t need = tarl_apsp_scratch_bytes(plan, B);
  int64_t waves = B * N;
  size_t lds = 0;
  uint8_t* sc = nullptr;
  if (need == 0) {
    lds = (size_t)(16 * N);
    if (waves > 65536) waves = 65536;
    if (lds > 64 * 1024)
      TARL_CHECK_HIP(hipFuncSetAttribute((const void*)k_apsp<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  } else {
    TARL_REQUIRE(scratch && scratch_bytes >= need, "scratch too small (tarl_apsp_scratch_bytes)");
    if (waves > 4096) waves = 4096;
    sc = (uint8_t*)scratch;
  }
  hipLaunchKernelGGL(k_apsp<W>, dim3((unsigned)waves), dim3(64), lds, (hipStream_t)stream, plan->out_ptr, plan->out_dst,
                     plan->out_eid, weights, w_bstride, B, N, sc, next_hop, dist);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_apsp(const tarl_plan* plan, const float* weights, int64_t B, int64_t w_bstride, void* scratch,
                         int64_t scratch_bytes, int64_t* next_hop, float* dist, tarl_stream stream) {
  return apsp_launch<float>(plan, weights, B, w_bstride, scratch, scratch_bytes, next_hop, dist, stream);
}

// the same with double edge weights (run_msa keeps its link costs in float64)
extern "C" int tarl_apsp_f64(const tarl_plan* plan, const double* weights, int64_t B, int64_t w_bstride, void* scratch,
                             int64_t scratch_bytes, int64_t* next_hop, float* dist, tarl_stream stream) {
  return apsp_launch<double>(plan, weights, B, w_bstride, scratch, scratch_bytes, next_hop, dist, stream);
}

// ---- all-or-nothing assignment of an OD demand along the next-hop table (src/algorithms/user_equilibrium_msa.py:117-131)
// One thread per OD pair: walk o -> d through next_hop and add the pair's volume to every ROAD node entered (the origin
// itself is skipped). fp64 atomics: the summation order over pairs is not fixed (differences ~1e-16 relative).
__global__ __launch_bounds__(RT_BLOCK) void k_msa_assign(const int64_t* __restrict__ next_hop, int64_t N,
                                                         const int64_t* __restrict__ od_o,
                                                         const int64_t* __restrict__ od_d,
                                                         const double* __restrict__ od_vol, int64_t P,
                                                         const uint8_t* __restrict__ is_road,
                                                         double* __restrict__ aux_flow) {
  const int64_t p = (int64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
  if (p >= P) return;
  const int64_t o = od_o[p], d = od_d[p];
  const double vol = od_vol[p];
  if (o < 0 || o >= N || d < 0 || d >= N || !(vol > 0.0)) return;
  if (next_hop[o * N + d] < 0) return;   // no path
  int64_t node = o;
  for (int64_t hops = 0; node != d && hops < N; ++hops) {
    node = next_hop[node * N + d];
    if (node < 0) return;
    if (is_road[node]) atomicAdd(&aux_flow[node], vol);
  }
}

extern "C" int tarl_msa_assign(const int64_t* next_hop, int64_t num_nodes, const int64_t* od_origin,
                               const int64_t* od_dest, const double* od_volume, int64_t num_pairs,
                               const uint8_t* is_road, double* aux_flow, tarl_stream stream) {
  TARL_REQUIRE(next_hop && od_origin && od_dest && od_volume && is_road && aux_flow, "null argument");
  TARL_REQUIRE(num_nodes >= 1 && num_pairs >= 0, "bad sizes");
  if (num_pairs == 0) return TARL_OK;
  hipLaunchKernelGGL(k_msa_assign, dim3((unsigned)ceil_div(num_pairs, RT_BLOCK)), dim3(RT_BLOCK), 0, (hipStream_t)stream,
                     next_hop, num_nodes, od_origin, od_dest, od_volume, num_pairs, is_road, aux_flow);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_select_next_hop(float* x, int64_t B, int64_t x_bstride, int64_t ldx, int32_t Nmax,
                                    int64_t num_nodes, const float* agent_features, int64_t num_agents,
                                    int64_t a_bstride, const int64_t* next_hop, int64_t nh_bstride,
                                    tarl_stream stream) {
  TARL_REQUIRE(x && agent_features && next_hop, "null argument");
  TARL_REQUIRE(B >= 1 && Nmax >= 1 && ldx >= 3 * (int64_t)Nmax + 7 && num_agents >= 1, "bad shape");
  if (num_nodes == 0) return TARL_OK;
  Layout L{Nmax, ldx, x_bstride};
  hipLaunchKernelGGL(k_select_next_hop, dim3((unsigned)ceil_div(B * num_nodes, RT_BLOCK)), dim3(RT_BLOCK), 0,
                     (hipStream_t)stream, x, L, B, num_nodes, agent_features, num_agents, a_bstride, next_hop,
                     nh_bstride);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}
