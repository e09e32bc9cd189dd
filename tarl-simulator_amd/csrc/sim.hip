// sim.hip — the traffic-flow step as dual-graph message passing (DirectionMPNN, ResponseMPNN, SimulationCoreModel).
//
// Reference semantics restated (never its code): src/direction_mpnn.py:44-196, src/response_mpnn.py:42-127,
// src/simulation_core_model.py:41-83. Both rounds are Jacobi-style: every message reads the pre-round state, then
// every row updates, hence one gather kernel + one row-update kernel per round.
//
// Compiled with -ffp-contract=off and without fast-math: every fp32 expression below is evaluated with the same
// IEEE operations, in the same order, as the reference's torch CPU ops, so the integer state is bit-exact.
#include <float.h>

#include "tarl_common.h"

#define SIM_BLOCK 256

struct PlanView {
  const int32_t* in_ptr;
  const int32_t* in_src;
  const int32_t* in_eid;
  const int32_t* out_ptr;
  const int32_t* out_dst;
  int64_t E;
};

static PlanView view(const tarl_plan* p) { return PlanView{p->in_ptr, p->in_src, p->in_eid, p->out_ptr, p->out_dst, p->E}; }

// ---- Direction: message + aggregate (dst-centric gather; S1-S4 of SURVEY §2.1) --------------------------------------
// One thread per (environment, downstream road i). In-edges are visited in ascending original edge id, which is
// both the sequential order of scatter_add (P) and the first-maximum-wins order of scatter_max.
__global__ __launch_bounds__(SIM_BLOCK) void k_direction_gather(PlanView pv, const float* __restrict__ x, Layout L,
                                                                int64_t B, int64_t R,
                                                                const float* __restrict__ edge_attr,
                                                                const float* __restrict__ log_edge_attr, float log_eps,
                                                                float t, const float* __restrict__ gumbel,
                                                                uint64_t seed, uint64_t counter,
                                                                float* __restrict__ dtt, float* __restrict__ chosen) {
  const int64_t gid = (int64_t)blockIdx.x * SIM_BLOCK + threadIdx.x;
  if (gid >= B * R) return;
  const int64_t b = gid / R;
  const int32_t i = (int32_t)(gid - b * R);
  const float* xb = x + b * L.bstride;
  const float* xi = xb + (int64_t)i * L.ldx;
  const int Nmax = L.Nmax;
  const float max_i = xi[L.col_maxn()];
  const float n_i = xi[L.col_n()];
  const float road_i = xi[L.col_road()];
  const float room_i = max_i - n_i;
  const bool has_room = n_i < max_i - TARL_CONGESTION_FILE;

  float P = 0.0f;
  float best = -FLT_MAX;  // scatter_max starts from numeric_limits::lowest()
  float best_id = 0.0f;
  PhiloxRun rng;          // device noise is keyed by the in-edge's CSC position: one Philox block per node at degree 4
  const int32_t k0 = pv.in_ptr[i], k1 = pv.in_ptr[i + 1];
  for (int32_t k = k0; k < k1; ++k) {
    const int32_t j = pv.in_src[k];
    const int32_t e = pv.in_eid[k];
    const float* xj = xb + (int64_t)j * L.ldx;
    const float id = xj[0];
    const float arr = xj[Nmax];
    const float dep = xj[2 * Nmax];
    const float max_j = xj[L.col_maxn()];
    const float n_j = xj[L.col_n()];
    const float ff_j = xj[L.col_ff()];
    const float sel_j = xj[L.col_sel()];
    const bool heads_here = sel_j == road_i;
    const bool m1 = (dep <= t) && has_room && heads_here && (n_j > 0.0f);
    const bool m2 = ((dep - t) < -10.0f) && ((max_j - TARL_CONGESTION_FILE) <= n_j) && ((max_j - n_j) <= room_i) &&
                    heads_here;
    const bool m = m1 || m2;
    const float prob = edge_attr[e] * (m ? 1.0f : 0.0f);
    P = P + prob;
    const int64_t ge = b * pv.E + e;
    float g;
    if (gumbel) {
      g = gumbel[ge];
    } else {
      const float u = rng.uniform(seed, counter, (uint64_t)(b * pv.E + k));
      g = gumbel_from_u01(u);
    }
    const float score = (m ? log_edge_attr[e] : log_eps) + g;
    if (score > best) {
      best = score;
      best_id = id;
    }
    if (dtt) {
      const float d = (dep - arr) - ff_j;
      dtt[ge] = d > 0.0f ? d : (d != d ? d : 0.0f);  // clamp(min=0), NaN passes through like torch.clamp
    }
  }
  chosen[gid] = (P > 0.0f) ? best_id : 0.0f;
}

// ---- Direction: update (S5). Every row, also when nothing was chosen ----------------------------------
__global__ __launch_bounds__(SIM_BLOCK) void k_direction_update(float* __restrict__ x, Layout L, int64_t B, int64_t R,
                                                                const float* __restrict__ cong, float t,
                                                                const float* __restrict__ chosen,
                                                                int32_t* __restrict__ status) {
  const int64_t gid = (int64_t)blockIdx.x * SIM_BLOCK + threadIdx.x;
  if (gid >= B * R) return;
  const int64_t b = gid / R;
  const int32_t i = (int32_t)(gid - b * R);
  float* xi = x + b * L.bstride + (int64_t)i * L.ldx;
  const int Nmax = L.Nmax;
  const int F = L.F();
  const float who = chosen[gid];
  // The reference performs three unconditional indexed writes at columns q, Nmax+q, 2Nmax+q with q = int(count) and
  // reads the counter through a live view. When a gridlock-relief move has over-filled a FIFO (q >= Nmax) those
  // writes land in the neighbouring blocks / scalar columns; reproduced here write by write, re-reading the row in
  // between exactly as the sequence of torch ops does. (A column >= F raises IndexError there; skipped here.)
  const int q = (int)xi[L.col_n()];  // .to(int64) truncates
  if (q >= 0 && q < F) xi[q] = who;
  if (q >= 0 && Nmax + q < F) xi[Nmax + q] = t;
  const float maxn = xi[L.col_maxn()];
  const float ff = xi[L.col_ff()];
  const float n0 = xi[L.col_n()];
  float c;
  if (cong) {
    c = cong[i];
  } else {  // src/direction_mpnn.py:178-183 (same expression as src/simulation_core_model.py:55-67)
    const float critical = xi[L.col_maxflow()] * ff / 3600.0f;
    c = ff * (maxn + 10.0f - critical);
  }
  const float t_cong = c / (maxn + 10.0f - n0);
  const float tt = (t_cong != t_cong) ? t_cong : fmaxf(ff, t_cong);  // torch.maximum propagates NaN
  if (q >= 0 && 2 * Nmax + q < F) xi[2 * Nmax + q] = t + tt;
  if (who != 0.0f) {
    const float n1 = xi[L.col_n()] + 1.0f;
    xi[L.col_n()] = n1;
    // a count that reaches Nmax leaves the reference's defined domain (its next update raises IndexError)
    if (status && n1 >= (float)Nmax) atomicOr(status, TARL_FLAG_COUNT_AT_NMAX);
  }
}

// ---- Response: message + max-aggregate (src-centric gather; S6-S8) -------------------------------------------------
__global__ __launch_bounds__(SIM_BLOCK) void k_response_gather(PlanView pv, const float* __restrict__ x, Layout L,
                                                               int64_t B, int64_t R, uint8_t* __restrict__ popped,
                                                               int32_t* __restrict__ any_popped) {
  const int64_t gid = (int64_t)blockIdx.x * SIM_BLOCK + threadIdx.x;
  bool acc = false;
  if (gid < B * R) {
    const int64_t b = gid / R;
    const int32_t i = (int32_t)(gid - b * R);
    const float* xb = x + b * L.bstride;
    const float* xi = xb + (int64_t)i * L.ldx;
    const long long cnt_up = (long long)xi[L.col_n()];
    if (cnt_up > 0) {
      const long long head = (long long)xi[0];
      const int32_t k1 = pv.out_ptr[i + 1];
      for (int32_t k = pv.out_ptr[i]; k < k1; ++k) {
        const float* xj = xb + (int64_t)pv.out_dst[k] * L.ldx;
        const long long cnt_dn = (long long)xj[L.col_n()];
        if (cnt_dn > 0 && cnt_dn - 1 < L.F()) {
          const long long tail = (long long)xj[cnt_dn - 1];
          acc = acc || (tail == head);
        }
      }
    }
    popped[gid] = acc ? 1 : 0;
  }
  if (any_popped) {
    const unsigned long long bal = __ballot(acc);
    if (bal != 0ull && (threadIdx.x & 63) == (unsigned)__ffsll((long long)bal) - 1u) atomicOr(any_popped, 1);
  }
}

// ---- Response: update = pop the head (S9). The last slot keeps its stale value -------------------------
__global__ __launch_bounds__(SIM_BLOCK) void k_response_update(float* __restrict__ x, Layout L, int64_t B, int64_t R,
                                                               const uint8_t* __restrict__ popped) {
  const int64_t gid = (int64_t)blockIdx.x * SIM_BLOCK + threadIdx.x;
  if (gid >= B * R) return;
  if (!popped[gid]) return;
  const int64_t b = gid / R;
  const int32_t i = (int32_t)(gid - b * R);
  float* xi = x + b * L.bstride + (int64_t)i * L.ldx;
  const int Nmax = L.Nmax;
  for (int blk = 0; blk < 3; ++blk) {
    float* q = xi + blk * Nmax;
    for (int s = 0; s + 1 < Nmax; ++s) q[s] = q[s + 1];
  }
  xi[L.col_n()] = xi[L.col_n()] - 1.0f;
}

// ---- live kernel timing (bench.py's roofline leg) -------------------------------------------------------------------
// When enabled, the frame kernels are bracketed by HIP events on the launch stream: tag 0 before the Direction gather,
// 1 after it, 2 after the row pass, 3 after the insert (+ next frame's choice) launch. Events come from a pre-created
// pool (no allocation in the launch path); tarl_prof_collect() synchronises them after the timed region and returns the
// summed duration of each kernel slot, over all timed frames and over the frames >= first_late_frame alone.
#include <vector>
static bool g_prof_on = false;
static std::vector<hipEvent_t> g_prof_events;   // marks in launch order
static std::vector<int> g_prof_tags;
static size_t g_prof_used = 0;
#define PROF_MARKS_PER_FRAME 4

extern "C" int tarl_prof_enable(int64_t max_frames) {
  for (hipEvent_t e : g_prof_events) (void)hipEventDestroy(e);
  g_prof_events.clear();
  g_prof_tags.clear();
  g_prof_used = 0;
  g_prof_on = max_frames > 0;
  for (int64_t i = 0; i < PROF_MARKS_PER_FRAME * max_frames; ++i) {
    hipEvent_t e;
    TARL_CHECK_HIP(hipEventCreate(&e));
    g_prof_events.push_back(e);
    g_prof_tags.push_back(-1);
  }
  return TARL_OK;
}

// ms_all[k] / ms_late[k]: summed duration of kernel slot k (0 Direction gather, 1 row pass, 2 insert [+ choice]) over
// all timed frames / over the timed frames with index >= first_late_frame; frames[0..1] = how many frames each sum
// covers. Re-arms the pool.
extern "C" int tarl_prof_collect(int64_t first_late_frame, double* ms_all, double* ms_late, int64_t* frames) {
  TARL_REQUIRE(ms_all && ms_late && frames, "null argument");
  for (int k = 0; k < 3; ++k) ms_all[k] = ms_late[k] = 0.0;
  int64_t frame = -1, n_all = 0, n_late = 0;
  for (size_t i = 0; i + 1 < g_prof_used; ++i) {
    const int ta = g_prof_tags[i], tb = g_prof_tags[i + 1];
    if (ta == 0) {
      ++frame;
      if (tb == 1) {
        ++n_all;
        if (frame >= first_late_frame) ++n_late;
      }
    }
    if (tb != ta + 1 || ta < 0 || ta > 2) continue;
    TARL_CHECK_HIP(hipEventSynchronize(g_prof_events[i + 1]));
    float ms = 0.0f;
    TARL_CHECK_HIP(hipEventElapsedTime(&ms, g_prof_events[i], g_prof_events[i + 1]));
    ms_all[ta] += ms;
    if (frame >= first_late_frame) ms_late[ta] += ms;
  }
  frames[0] = n_all;
  frames[1] = n_late;
  g_prof_used = 0;
  return TARL_OK;
}

hipEvent_t tarl_prof_mark(hipStream_t s, int tag) {
  if (!g_prof_on || g_prof_used >= g_prof_events.size()) return nullptr;
  hipEvent_t e = g_prof_events[g_prof_used];
  g_prof_tags[g_prof_used++] = tag;
  (void)hipEventRecord(e, s);
  return e;
}

// ---- host side -----------------------------------------------------------------------------------------------------
static int check_state(const tarl_plan* plan, const float* x, int64_t B, int64_t bstride, int64_t ldx, int32_t Nmax,
                       int64_t R) {
  TARL_REQUIRE(plan != nullptr, "plan is null");
  TARL_REQUIRE(x != nullptr, "x is null");
  TARL_REQUIRE(B >= 1 && Nmax >= 1, "B and Nmax must be positive");
  TARL_REQUIRE(ldx >= 3 * (int64_t)Nmax + 7, "row stride smaller than F = 3*Nmax+7");
  TARL_REQUIRE(R >= 0 && R <= plan->N, "num_roads exceeds the plan's node count");
  TARL_REQUIRE(R == plan->N, "the plan must be built on the road graph (num_roads == plan nodes)");
  TARL_REQUIRE(B == 1 || bstride >= R * ldx, "environment stride smaller than one environment");
  TARL_REQUIRE(B * R < ((int64_t)1 << 31) * SIM_BLOCK, "grid too large");
  return TARL_OK;
}

extern "C" int tarl_direction_step(const tarl_plan* plan, float* x, int64_t B, int64_t x_bstride, int64_t ldx,
                                   int32_t Nmax, int64_t R, const float* edge_attr, const float* log_edge_attr,
                                   float log_eps, const float* cong, float time, const float* gumbel, uint64_t seed,
                                   uint64_t counter, float* dtt, float* chosen, int32_t* status, tarl_stream stream) {
  int rc = check_state(plan, x, B, x_bstride, ldx, Nmax, R);
  if (rc) return rc;
  TARL_REQUIRE(chosen != nullptr, "chosen scratch is null");
  TARL_REQUIRE(plan->E == 0 || (edge_attr && log_edge_attr), "edge_attr / log_edge_attr is null");
  if (R == 0) return TARL_OK;
  const Layout L{Nmax, ldx, x_bstride};
  const unsigned grid = (unsigned)ceil_div(B * R, SIM_BLOCK);
  hipStream_t s = (hipStream_t)stream;
  const bool timed = tarl_prof_mark(s, 0) != nullptr;
  hipLaunchKernelGGL(k_direction_gather, dim3(grid), dim3(SIM_BLOCK), 0, s, view(plan), x, L, B, R, edge_attr,
                     log_edge_attr, log_eps, time, gumbel, seed, counter, dtt, chosen);
  TARL_LAUNCH_CHECK();
  if (timed) (void)tarl_prof_mark(s, 1);
  hipLaunchKernelGGL(k_direction_update, dim3(grid), dim3(SIM_BLOCK), 0, s, x, L, B, R, cong, time, chosen, status);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_response_step(const tarl_plan* plan, float* x, int64_t B, int64_t x_bstride, int64_t ldx,
                                  int32_t Nmax, int64_t R, uint8_t* popped, int32_t* any_popped, tarl_stream stream) {
  int rc = check_state(plan, x, B, x_bstride, ldx, Nmax, R);
  if (rc) return rc;
  TARL_REQUIRE(popped != nullptr, "popped is null");
  hipStream_t s = (hipStream_t)stream;
  if (any_popped) TARL_CHECK_HIP(hipMemsetAsync(any_popped, 0, sizeof(int32_t), s));
  if (R == 0) return TARL_OK;
  const Layout L{Nmax, ldx, x_bstride};
  const unsigned grid = (unsigned)ceil_div(B * R, SIM_BLOCK);
  hipLaunchKernelGGL(k_response_gather, dim3(grid), dim3(SIM_BLOCK), 0, s, view(plan), x, L, B, R, popped, any_popped);
  TARL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_response_update, dim3(grid), dim3(SIM_BLOCK), 0, s, x, L, B, R, popped);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_core_step(const tarl_plan* plan, float* x, int64_t B, int64_t x_bstride, int64_t ldx, int32_t Nmax,
                              int64_t R, const float* edge_attr, const float* log_edge_attr, float log_eps,
                              const float* cong, float time, const float* gumbel, uint64_t seed, uint64_t counter,
                              float* dtt, float* chosen, uint8_t* popped, int32_t* any_popped, int32_t* status,
                              tarl_stream stream) {
  int rc = tarl_direction_step(plan, x, B, x_bstride, ldx, Nmax, R, edge_attr, log_edge_attr, log_eps, cong, time,
                               gumbel, seed, counter, dtt, chosen, status, stream);
  if (rc) return rc;
  return tarl_response_step(plan, x, B, x_bstride, ldx, Nmax, R, popped, any_popped, stream);
}
