// agents.hip — the environment step around the core model, entirely on device (no host syncs):
// apply action, withdraw, insert (+ reward and critic observation), reset.
//
// Reference semantics restated: src/reinforcement_learning.py:222-231,256-267 (choice phase, reward),
// src/agents/base.py:247-331 (insert), :348-403 (withdraw), src/transportation_simulator.py:353-358 and
// src/agents/base.py:496-503 (reset). The reference's dense N x N adjacency is replaced by the plan's CSR rows, its
// Python loop over roads and argsort by a deterministic rank-within-road (stable, i.e. agent-id order).
#include "tarl_common.h"

#define AG_BLOCK 256
#define INS_BLOCK 1024

// ---- choice phase ----------------------------------------------------------------------------------------------------
// x[b, src(e), SELECTED_ROAD] = dst(e) for every selected edge; with several selected edges of one source the
// reference's sequential index_put keeps the LAST one in edge order => the largest selected edge id wins.
__global__ __launch_bounds__(AG_BLOCK) void k_apply_action(const int32_t* __restrict__ out_ptr,
                                                           const int32_t* __restrict__ out_dst,
                                                           const int32_t* __restrict__ out_eid,
                                                           const int32_t* __restrict__ dst, float* __restrict__ x,
                                                           Layout L, int64_t B, int64_t N, int64_t E,
                                                           const int64_t* __restrict__ onehot,
                                                           const int32_t* __restrict__ choice) {
  const int64_t gid = (int64_t)blockIdx.x * AG_BLOCK + threadIdx.x;
  if (gid >= B * N) return;
  const int64_t b = gid / N;
  const int32_t i = (int32_t)(gid - b * N);
  int32_t sel = -1;
  if (choice) {
    const int32_t e = choice[gid];
    if (e >= 0 && e < E) sel = dst[e];
  } else {
    int32_t best_e = -1;
    const int32_t k1 = out_ptr[i + 1];
    for (int32_t k = out_ptr[i]; k < k1; ++k) {
      const int32_t e = out_eid[k];
      if (onehot[b * E + e] != 0 && e > best_e) {
        best_e = e;
        sel = out_dst[k];
      }
    }
  }
  if (sel >= 0) x[b * L.bstride + (int64_t)i * L.ldx + L.col_sel()] = (float)sel;
}

// ---- withdraw --------------------------------------------------------------------------------------------------------
// One thread per (environment, row): count the leading run of FIFO slots whose agent may leave here
// (edge road -> destination exists, departure time reached), mark those agents DONE, shift the three FIFO blocks left
// by that count with zero fill, decrement the counter.
__global__ __launch_bounds__(AG_BLOCK) void k_withdraw(const int32_t* __restrict__ out_ptr,
                                                       const int32_t* __restrict__ out_dst, float* __restrict__ x,
                                                       Layout L, int64_t B, int64_t N, int64_t planN,
                                                       float* __restrict__ ag, int64_t A, int64_t a_bstride, float t,
                                                       uint8_t* __restrict__ withdrawn) {
  const int64_t gid = (int64_t)blockIdx.x * AG_BLOCK + threadIdx.x;
  if (gid >= B * N) return;
  const int64_t b = gid / N;
  const int32_t i = (int32_t)(gid - b * N);
  float* xi = x + b * L.bstride + (int64_t)i * L.ldx;
  float* agb = ag + b * a_bstride;
  const int Nmax = L.Nmax;
  const float n = xi[L.col_n()];
  int c = 0;
  if (n > 0.0f) {
    const long long road = (long long)xi[L.col_road()];
    int32_t k0 = 0, k1 = 0;
    if (road >= 0 && road < planN) {
      k0 = out_ptr[road];
      k1 = out_ptr[road + 1];
    }
    for (int s = 0; s < Nmax && (float)s < n; ++s) {
      const long long id = (long long)xi[s];
      if (id < 0 || id >= A) break;
      const long long dest = (long long)agb[id * AG_COLS + AG_DEST];
      bool conn = false;
      for (int32_t k = k0; k < k1; ++k) conn = conn || ((long long)out_dst[k] == dest);
      if (!(conn && xi[2 * Nmax + s] <= t)) break;
      ++c;
    }
  }
  if (withdrawn) withdrawn[gid] = c > 0 ? 1 : 0;
  if (c == 0) return;
  for (int s = 0; s < c; ++s) {
    const long long id = (long long)xi[s];
    float* a = agb + id * AG_COLS;
    a[AG_DONE] = 1.0f;
    a[AG_ON_WAY] = 0.0f;
    a[AG_ARR] = t;
  }
  for (int blk = 0; blk < 3; ++blk) {
    float* q = xi + blk * Nmax;
    for (int s = 0; s < Nmax; ++s) q[s] = (s + c < Nmax) ? q[s + c] : 0.0f;
  }
  xi[L.col_n()] = n - (float)c;
}

// ---- insert + reward + critic observation ---------------------------------------------------------------------------
// One workgroup per environment.
//  phase 1: ordered compaction of the candidates (ready and target road has room) in ascending agent id;
//  phase 2: rank of each candidate among earlier candidates of the same road (stable order); the first
//           min(count, capacity) per road are admitted into consecutive slots n0 + rank; all of a road's arrivals get
//           the same departure time because the reference evaluates it with the pre-insertion count;
//  phase 3: counters += admitted; phase 4: reward = -sum(counts), counts copied out for the critic.
__device__ __forceinline__ bool insert_target(const float* __restrict__ xb, const Layout& L, int64_t N,
                                              const float* __restrict__ a, int32_t* road, int32_t* cap) {
  const long long origin = (long long)a[AG_ORIGIN];
  if (origin < 0 || origin >= N) return false;
  const long long r = (long long)xb[origin * L.ldx + L.col_sel()];
  if (r < 0 || r >= N) return false;
  const float* xr = xb + r * L.ldx;
  const long long room = (long long)(xr[L.col_maxn()] - TARL_CONGESTION_FILE - xr[L.col_n()]);
  *road = (int32_t)r;
  *cap = (int32_t)(room > 0x7fffffff ? 0x7fffffff : room);
  return room > 0;
}

__global__ __launch_bounds__(INS_BLOCK) void k_insert(float* __restrict__ x, Layout L, int64_t N,
                                                      float* __restrict__ ag, int64_t A, int64_t a_bstride,
                                                      const float* __restrict__ cong, float t,
                                                      int32_t* __restrict__ scratch, float* __restrict__ reward,
                                                      float* __restrict__ counts) {
  __shared__ int32_t s_wave[INS_BLOCK / 64];
  __shared__ float s_red[INS_BLOCK / 64];
  __shared__ int32_t s_total;
  const int64_t b = blockIdx.x;
  float* xb = x + b * L.bstride;
  float* agb = ag ? ag + b * a_bstride : nullptr;
  int32_t* cand_agent = scratch ? scratch + b * 2 * A : nullptr;
  int32_t* cand_road = scratch ? cand_agent + A : nullptr;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int Nmax = L.Nmax;

  int32_t base = 0;
  if (agb) {
    for (int64_t a0 = 0; a0 < A; a0 += INS_BLOCK) {
      const int64_t a = a0 + tid;
      bool cnd = false;
      int32_t road = 0, cap = 0;
      if (a < A) {
        const float* row = agb + a * AG_COLS;
        if (row[AG_DEP] <= t && row[AG_ON_WAY] == 0.0f && row[AG_DONE] == 0.0f)
          cnd = insert_target(xb, L, N, row, &road, &cap);
      }
      const unsigned long long bal = __ballot(cnd);
      const int lane_off = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) s_wave[wid] = __popcll(bal);
      __syncthreads();
      int32_t wbase = 0, tot = 0;
      for (int w = 0; w < INS_BLOCK / 64; ++w) {
        const int32_t v = s_wave[w];
        if (w < wid) wbase += v;
        tot += v;
      }
      if (cnd) {
        cand_agent[base + wbase + lane_off] = (int32_t)a;
        cand_road[base + wbase + lane_off] = road;
      }
      base += tot;
      __syncthreads();
    }
  }
  if (tid == 0) s_total = base;
  __threadfence_block();
  __syncthreads();
  const int32_t Lc = s_total;

  // phase 2 (x counters are still the pre-insertion values: nobody writes them before the next barrier)
  for (int32_t idx = tid; idx < Lc; idx += INS_BLOCK) {
    const int32_t r = cand_road[idx];
    const int32_t a = cand_agent[idx];
    int32_t rank = 0, total = 0;
    for (int32_t k = 0; k < Lc; ++k) {
      const bool same = cand_road[k] == r;
      total += same ? 1 : 0;
      rank += (same && k < idx) ? 1 : 0;
    }
    float* xr = xb + (int64_t)r * L.ldx;
    const float n0 = xr[L.col_n()];
    const float maxn = xr[L.col_maxn()];
    const long long cap = (long long)(maxn - TARL_CONGESTION_FILE - n0);
    int32_t commit = 0;
    if (rank < cap) {
      const long long slot = (long long)n0 + rank;
      if (slot >= 0 && slot < Nmax) {
        const float ff = xr[L.col_ff()];
        const float t_cong = cong ? cong[r] / (maxn + 10.0f - (float)(long long)n0) : 0.0f;
        const float tt = (t_cong != t_cong) ? t_cong : fmaxf(ff, t_cong);
        xr[slot] = (float)a;
        xr[Nmax + slot] = t;
        xr[2 * Nmax + slot] = t + tt;
      }
      agb[(int64_t)a * AG_COLS + AG_ON_WAY] = 1.0f;
      if (rank == 0) commit = (int32_t)(total < cap ? total : cap);
    }
    cand_agent[idx] = commit;  // only this thread reads cand_agent[idx]
  }
  __threadfence_block();
  __syncthreads();
  // phase 3
  for (int32_t idx = tid; idx < Lc; idx += INS_BLOCK) {
    const int32_t cmt = cand_agent[idx];
    if (cmt > 0) {
      float* xr = xb + (int64_t)cand_road[idx] * L.ldx;
      xr[L.col_n()] = xr[L.col_n()] + (float)cmt;
    }
  }
  __threadfence_block();
  __syncthreads();
  // phase 4: reward (sum of small integers: exact in fp32 in any order) and the critic's per-node counts
  if (reward || counts) {
    float acc = 0.0f;
    for (int64_t i = tid; i < N; i += INS_BLOCK) {
      const float v = xb[i * L.ldx + L.col_n()];
      if (counts) counts[b * N + i] = v;
      acc += v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0) s_red[wid] = acc;
    __syncthreads();
    if (tid == 0 && reward) {
      float tot = 0.0f;
      for (int w = 0; w < INS_BLOCK / 64; ++w) tot += s_red[w];
      reward[b] = -tot;
    }
  }
}

// ---- reset -----------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AG_BLOCK) void k_reset_rows(float* __restrict__ x, Layout L, int64_t B, int64_t N) {
  const int64_t gid = (int64_t)blockIdx.x * AG_BLOCK + threadIdx.x;
  const int W = 3 * L.Nmax + 1;  // FIFO blocks + one lane for the counter
  if (gid >= B * N * W) return;
  const int64_t row = gid / W;
  const int c = (int)(gid - row * W);
  const int64_t b = row / N, i = row - b * N;
  x[b * L.bstride + i * L.ldx + (c < 3 * L.Nmax ? c : L.col_n())] = 0.0f;
}

__global__ __launch_bounds__(AG_BLOCK) void k_reset_agents(float* __restrict__ ag, int64_t B, int64_t A,
                                                           int64_t a_bstride) {
  const int64_t gid = (int64_t)blockIdx.x * AG_BLOCK + threadIdx.x;
  if (gid >= B * A) return;
  const int64_t b = gid / A, a = gid - b * A;
  float* row = ag + b * a_bstride + a * AG_COLS;
  row[AG_ON_WAY] = 0.0f;
  row[AG_DONE] = 0.0f;
}

// ---- host side -------------------------------------------------------------------------------------------------------
static int check_x(const float* x, int64_t B, int64_t bstride, int64_t ldx, int32_t Nmax, int64_t N) {
  TARL_REQUIRE(x != nullptr, "x is null");
  TARL_REQUIRE(B >= 1 && Nmax >= 1 && N >= 0, "bad sizes");
  TARL_REQUIRE(ldx >= 3 * (int64_t)Nmax + 7, "row stride smaller than F = 3*Nmax+7");
  TARL_REQUIRE(B == 1 || bstride >= N * ldx, "environment stride smaller than one environment");
  return TARL_OK;
}

extern "C" int tarl_apply_action(const tarl_plan* plan, float* x, int64_t B, int64_t x_bstride, int64_t ldx,
                                 int32_t Nmax, const int64_t* onehot, const int32_t* choice, tarl_stream stream) {
  TARL_REQUIRE(plan != nullptr, "plan is null");
  int rc = check_x(x, B, x_bstride, ldx, Nmax, plan->N);
  if (rc) return rc;
  TARL_REQUIRE((onehot != nullptr) != (choice != nullptr), "pass exactly one of action_onehot / choice");
  if (plan->N == 0) return TARL_OK;
  const Layout L{Nmax, ldx, x_bstride};
  hipLaunchKernelGGL(k_apply_action, dim3((unsigned)ceil_div(B * plan->N, AG_BLOCK)), dim3(AG_BLOCK), 0,
                     (hipStream_t)stream, plan->out_ptr, plan->out_dst, plan->out_eid, plan->dst, x, L, B, plan->N,
                     plan->E, onehot, choice);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_withdraw_step(const tarl_plan* plan, float* x, int64_t B, int64_t x_bstride, int64_t ldx,
                                  int32_t Nmax, int64_t N, float* ag, int64_t A, int64_t a_bstride, float time,
                                  uint8_t* withdrawn, tarl_stream stream) {
  TARL_REQUIRE(plan != nullptr, "plan is null");
  int rc = check_x(x, B, x_bstride, ldx, Nmax, N);
  if (rc) return rc;
  TARL_REQUIRE(ag != nullptr && A >= 1, "agent_features is null or empty");
  TARL_REQUIRE(B == 1 || a_bstride >= A * AG_COLS, "agent stride smaller than one population");
  if (N == 0) return TARL_OK;
  const Layout L{Nmax, ldx, x_bstride};
  hipLaunchKernelGGL(k_withdraw, dim3((unsigned)ceil_div(B * N, AG_BLOCK)), dim3(AG_BLOCK), 0, (hipStream_t)stream,
                     plan->out_ptr, plan->out_dst, x, L, B, N, plan->N, ag, A, a_bstride, time, withdrawn);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_insert_step(float* x, int64_t B, int64_t x_bstride, int64_t ldx, int32_t Nmax, int64_t N,
                                float* ag, int64_t A, int64_t a_bstride, const float* cong, float time,
                                int32_t* scratch, float* reward, float* counts, tarl_stream stream) {
  int rc = check_x(x, B, x_bstride, ldx, Nmax, N);
  if (rc) return rc;
  TARL_REQUIRE(ag == nullptr || (A >= 1 && scratch != nullptr), "agents given without scratch");
  TARL_REQUIRE(ag == nullptr || B == 1 || a_bstride >= A * AG_COLS, "agent stride smaller than one population");
  TARL_REQUIRE(B < ((int64_t)1 << 31), "too many environments");
  const Layout L{Nmax, ldx, x_bstride};
  hipLaunchKernelGGL(k_insert, dim3((unsigned)B), dim3(INS_BLOCK), 0, (hipStream_t)stream, x, L, N, ag, A, a_bstride,
                     cong, time, scratch, reward, counts);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_reset_state(float* x, int64_t B, int64_t x_bstride, int64_t ldx, int32_t Nmax, int64_t N,
                                float* ag, int64_t A, int64_t a_bstride, tarl_stream stream) {
  int rc = check_x(x, B, x_bstride, ldx, Nmax, N);
  if (rc) return rc;
  const Layout L{Nmax, ldx, x_bstride};
  const int64_t work = B * N * (3 * (int64_t)Nmax + 1);
  if (work > 0) {
    hipLaunchKernelGGL(k_reset_rows, dim3((unsigned)ceil_div(work, AG_BLOCK)), dim3(AG_BLOCK), 0, (hipStream_t)stream,
                       x, L, B, N);
    TARL_LAUNCH_CHECK();
  }
  if (ag && A > 0) {
    hipLaunchKernelGGL(k_reset_agents, dim3((unsigned)ceil_div(B * A, AG_BLOCK)), dim3(AG_BLOCK), 0,
                       (hipStream_t)stream, ag, B, A, a_bstride);
    TARL_LAUNCH_CHECK();
  }
  return TARL_OK;
}
