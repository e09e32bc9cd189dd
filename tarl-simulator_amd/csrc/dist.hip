// dist.hip — GraphDistribution: one categorical per source node over its out-edges, on the static CSR plan.
//
// Reference semantics restated: src/reinforcement_learning.py:15-96 (+ torch-scatter 2.1.2 scatter_softmax /
// scatter_max, torch CPU cumsum). The per-step sort/argsort/unique/cumsum/boundary masks of the reference are all
// static topology and live in the plan; what remains per call is segment arithmetic.
#include <math.h>
#include <stdlib.h>

#include "tarl_common.h"

#define SEL_CARRIED 0x80u   // = fused_common.h: rank byte of a node that drew nothing
#define DIST_BLOCK 256
#define ENV_BLOCK 1024
#define LOG_EPS_P 1e-8f  // log(p + 1e-8), src/reinforcement_learning.py:27

// ---- segment softmax ---------------------------------------------------------------------------------------------
// proba = exp(l/T - max_g) / sum_g, the group sum accumulated sequentially in plan order (scatter_softmax, no epsilon).
__global__ __launch_bounds__(DIST_BLOCK) void k_softmax(const int32_t* __restrict__ out_ptr,
                                                        const int32_t* __restrict__ out_eid,
                                                        const float* __restrict__ logits, int64_t B, int64_t N,
                                                        int64_t E, float temperature, float* __restrict__ proba) {
  const int64_t gid = (int64_t)blockIdx.x * DIST_BLOCK + threadIdx.x;
  if (gid >= B * N) return;
  const int64_t b = gid / N;
  const int32_t i = (int32_t)(gid - b * N);
  const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];
  if (k0 == k1) return;
  const float* lb = logits + b * E;
  float* pb = proba + b * E;
  float mx = -INFINITY;
  for (int32_t k = k0; k < k1; ++k) mx = fmaxf(mx, lb[out_eid[k]] / temperature);
  float sum = 0.0f;
  for (int32_t k = k0; k < k1; ++k) sum = sum + expf(lb[out_eid[k]] / temperature - mx);
  for (int32_t k = k0; k < k1; ++k) {
    const int32_t e = out_eid[k];
    pb[e] = expf(lb[e] / temperature - mx) / sum;
  }
}

// ---- sample ---------------------------------------------------------------------------------------------------------
// One workgroup per environment. Phase A: per-group sums in double + exclusive scan over the groups (the reference's
// *global* cumsum over the sorted edges: torch's CPU cumsum accumulates fp32 inputs in double and rounds every output
// to fp32). Phase B: per group, cum_k = fp32(fp32(base + s_k) - fp32(base)), pick the first k with u < cum_k.
__global__ __launch_bounds__(ENV_BLOCK) void k_sample(const int32_t* __restrict__ out_ptr,
                                                      const int32_t* __restrict__ out_eid,
                                                      const int32_t* __restrict__ node_of_group,
                                                      const float* __restrict__ proba, int64_t N, int64_t E, int64_t G,
                                                      const float* __restrict__ uniform, uint64_t seed,
                                                      uint64_t counter, double* __restrict__ base_all,
                                                      int64_t* __restrict__ onehot, int32_t* __restrict__ choice) {
  __shared__ double s_wave[ENV_BLOCK / 64];
  const int64_t b = blockIdx.x;
  const float* pb = proba + b * E;
  double* base = base_all + b * (G + 1);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  double running = 0.0;
  for (int64_t g0 = 0; g0 < G; g0 += ENV_BLOCK) {
    const int64_t g = g0 + tid;
    double s = 0.0;
    if (g < G) {
      const int32_t i = node_of_group[g];
      const int32_t k1 = out_ptr[i + 1];
      for (int32_t k = out_ptr[i]; k < k1; ++k) s += (double)pb[out_eid[k]];
    }
    // inclusive scan inside the wave, then across waves
    double inc = s;
    for (int off = 1; off < 64; off <<= 1) {
      const double v = __shfl_up(inc, off);
      if (lane >= off) inc += v;
    }
    if (lane == 63) s_wave[wid] = inc;
    __syncthreads();
    double wbase = 0.0, tot = 0.0;
    for (int w = 0; w < ENV_BLOCK / 64; ++w) {
      const double v = s_wave[w];
      if (w < wid) wbase += v;
      tot += v;
    }
    double exc = __shfl_up(inc, 1);  // exclusive prefix inside the wave
    if (lane == 0) exc = 0.0;
    if (g < G) base[g] = running + wbase + exc;
    running += tot;
    __syncthreads();
  }
  // every thread reads only the base it wrote itself (same g) => no extra fence needed
  for (int64_t g0 = 0; g0 < G; g0 += ENV_BLOCK) {
    const int64_t g = g0 + tid;
    if (g >= G) break;
    const int32_t i = node_of_group[g];
    const double bg = base[g];
    const float bg32 = (float)bg;
    const float u = uniform ? uniform[b * G + g] : philox_uniform(seed, counter, (uint64_t)(b * G + g));
    double run = bg;
    int32_t pick = -1;
    const int32_t k1 = out_ptr[i + 1];
    for (int32_t k = out_ptr[i]; k < k1; ++k) {
      const int32_t e = out_eid[k];
      run += (double)pb[e];
      const float cum = (float)run - bg32;
      const bool hit = (pick < 0) && (u < cum);
      if (hit) pick = e;
      if (onehot) onehot[b * E + e] = hit ? 1 : 0;
    }
    if (choice) choice[b * N + i] = pick;
  }
}

__global__ __launch_bounds__(DIST_BLOCK) void k_fill_choice(int32_t* __restrict__ choice, int64_t n) {
  const int64_t gid = (int64_t)blockIdx.x * DIST_BLOCK + threadIdx.x;
  if (gid < n) choice[gid] = -1;
}

// ---- mode: per-node argmax, first maximum in ORIGINAL edge order wins (scatter_max on CPU) ---------------------------
__global__ __launch_bounds__(DIST_BLOCK) void k_mode(const int32_t* __restrict__ out_ptr,
                                                     const int32_t* __restrict__ out_eid,
                                                     const float* __restrict__ proba, int64_t B, int64_t N, int64_t E,
                                                     float* __restrict__ onehot, int32_t* __restrict__ choice) {
  const int64_t gid = (int64_t)blockIdx.x * DIST_BLOCK + threadIdx.x;
  if (gid >= B * N) return;
  const int64_t b = gid / N;
  const int32_t i = (int32_t)(gid - b * N);
  const float* pb = proba + b * E;
  const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];
  float best = -INFINITY;
  int32_t best_e = -1;
  for (int32_t k = k0; k < k1; ++k) {
    const int32_t e = out_eid[k];
    const float p = pb[e];
    if (best_e < 0 || p > best || (p == best && e < best_e)) {
      best = p;
      best_e = e;
    }
  }
  if (onehot)
    for (int32_t k = k0; k < k1; ++k) onehot[b * E + out_eid[k]] = (out_eid[k] == best_e) ? 1.0f : 0.0f;
  if (choice) choice[gid] = best_e;
}

// ---- log_prob / entropy forward -------------------------------------------------------------------------------------
// One workgroup per batch row; fixed reduction tree => deterministic.
__device__ __forceinline__ float block_sum(float v, float* s_red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  __syncthreads();
  if (lane == 0) s_red[wid] = v;
  __syncthreads();
  float tot = 0.0f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += s_red[w];
  return tot;
}

__global__ __launch_bounds__(ENV_BLOCK) void k_logprob_entropy_fwd(const int32_t* __restrict__ out_ptr,
                                                                   const int32_t* __restrict__ out_eid,
                                                                   const float* __restrict__ proba, int64_t N,
                                                                   int64_t E, const int64_t* __restrict__ onehot,
                                                                   const int32_t* __restrict__ choice,
                                                                   float* __restrict__ log_prob,
                                                                   float* __restrict__ entropy) {
  __shared__ float s_red[ENV_BLOCK / 64];
  const int64_t b = blockIdx.x;
  const float* pb = proba + b * E;
  float lp = 0.0f, ent = 0.0f, bad = 0.0f;
  for (int64_t i = threadIdx.x; i < N; i += ENV_BLOCK) {
    const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];
    if (k0 == k1) continue;
    long long asum = 0;
    const int32_t ce = choice ? choice[b * N + i] : -1;
    for (int32_t k = k0; k < k1; ++k) {
      const int32_t e = out_eid[k];
      const float p = pb[e];
      const float lg = logf(p + LOG_EPS_P);
      ent -= p * lg;
      const long long a = onehot ? (long long)onehot[b * E + e] : (e == ce ? 1 : 0);
      asum += a;
      lp += (float)a * lg;
    }
    if (asum != 1) bad = 1.0f;
  }
  const float lp_t = block_sum(lp, s_red);
  const float ent_t = block_sum(ent, s_red);
  const float bad_t = block_sum(bad, s_red);
  if (threadIdx.x == 0) {
    if (log_prob) log_prob[b] = bad_t > 0.0f ? -INFINITY : lp_t;
    if (entropy) entropy[b] = ent_t;
  }
}

// ---- the rollout's action draw in one launch: softmax -> sample -> log_prob ------------------------------------------------
// A state-dependent policy has new logits for every (environment, edge) every frame; the three launches above would
// write and re-read the probabilities twice. One workgroup per environment: same arithmetic, same summation orders
// (per-node sequential max / sum / double running sum, the block scan of k_sample for the group bases, the per-thread
// node order and reduction tree of k_logprob_entropy_fwd), so actions and log-probs are bit-identical to the chain
// tarl_graphdist_softmax -> _sample -> _logprob_entropy_fwd; probabilities are recomputed where the chain re-reads them
// (expf / the division are deterministic). The action is written as the edge id ([B][N], -1 = none), as the rank byte of
// the rollout buffers ([B][N], SEL_CARRIED | previous rank where nothing was drawn) and straight into the packed
// state's SELECTED_ROAD byte ([N][B]) — the choice phase of SimulatorEnv._step.
__device__ __forceinline__ void node_softmax_stats(const float* __restrict__ lb, const int32_t* __restrict__ out_eid,
                                                   int32_t k0, int32_t k1, float temperature, float* mx_out,
                                                   float* sum_out) {
  float mx = -INFINITY;
  for (int32_t k = k0; k < k1; ++k) mx = fmaxf(mx, lb[out_eid[k]] / temperature);
  float sum = 0.0f;
  for (int32_t k = k0; k < k1; ++k) sum = sum + expf(lb[out_eid[k]] / temperature - mx);
  *mx_out = mx;
  *sum_out = sum;
}

__global__ __launch_bounds__(ENV_BLOCK) void k_graphdist_rollout(
    const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_eid, const int32_t* __restrict__ node_of_group,
    const float* __restrict__ logits, int64_t B, int64_t N, int64_t E, int64_t G, float temperature,
    const float* __restrict__ uniform, uint64_t seed, uint64_t counter, double* __restrict__ base_all,
    float* __restrict__ u_all, int32_t* __restrict__ choice_eid, uint8_t* __restrict__ choice8,
    uint8_t* __restrict__ sel8, float* __restrict__ log_prob, uint64_t idx_base) {
  __shared__ double s_wave[ENV_BLOCK / 64];
  __shared__ float s_red[ENV_BLOCK / 64];
  const int64_t b = blockIdx.x;
  const float* lb = logits + b * E;
  double* base = base_all + b * N;       // indexed by NODE (written by the group's thread, read by the node's)
  float* un = u_all + b * N;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

  // phase A (k_sample's): per-group probability sums in double, exclusive scan over the groups
  double running = 0.0;
  for (int64_t g0 = 0; g0 < G; g0 += ENV_BLOCK) {
    const int64_t g = g0 + tid;
    double s = 0.0;
    int32_t i = 0;
    if (g < G) {
      i = node_of_group[g];
      const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];
      float mx, sum;
      node_softmax_stats(lb, out_eid, k0, k1, temperature, &mx, &sum);
      for (int32_t k = k0; k < k1; ++k) s += (double)(expf(lb[out_eid[k]] / temperature - mx) / sum);
    }
    double inc = s;
    for (int off = 1; off < 64; off <<= 1) {
      const double v = __shfl_up(inc, off);
      if (lane >= off) inc += v;
    }
    if (lane == 63) s_wave[wid] = inc;
    __syncthreads();
    double wbase = 0.0, tot = 0.0;
    for (int w = 0; w < ENV_BLOCK / 64; ++w) {
      const double v = s_wave[w];
      if (w < wid) wbase += v;
      tot += v;
    }
    double exc = __shfl_up(inc, 1);
    if (lane == 0) exc = 0.0;
    if (g < G) {
      base[i] = running + wbase + exc;
      un[i] = uniform ? uniform[b * G + g] : philox_uniform(seed, counter, idx_base + (uint64_t)(b * G + g));
    }
    running += tot;
    __syncthreads();
  }
  // phase B: pick (k_sample's second loop) and log-prob (k_logprob_entropy_fwd's thread -> node order)
  float lp = 0.0f, bad = 0.0f;
  for (int64_t i = tid; i < N; i += ENV_BLOCK) {
    const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];
    int32_t pick = -1, rank = 0;
    if (k0 != k1) {
      float mx, sum;
      node_softmax_stats(lb, out_eid, k0, k1, temperature, &mx, &sum);
      const double bg = base[i];
      const float bg32 = (float)bg, u = un[i];
      double run = bg;
      float lg_pick = 0.0f;
      for (int32_t k = k0; k < k1; ++k) {
        const int32_t e = out_eid[k];
        const float p = expf(lb[e] / temperature - mx) / sum;
        run += (double)p;
        const float cum = (float)run - bg32;
        if (pick < 0 && u < cum) {
          pick = e;
          rank = k - k0;
          lg_pick = logf(p + LOG_EPS_P);
        }
      }
      if (pick >= 0)
        lp += lg_pick;
      else
        bad = 1.0f;
    }
    if (choice_eid) choice_eid[b * N + i] = pick;
    if (choice8 || sel8) {
      uint32_t code = (uint32_t)rank;
      if (pick < 0) code = ((sel8 ? sel8[i * B + b] : 0u) & 0x7Fu) | SEL_CARRIED;
      if (sel8) sel8[i * B + b] = (uint8_t)code;
      if (choice8) choice8[b * N + i] = (uint8_t)code;
    }
  }
  const float lp_t = block_sum(lp, s_red);
  const float bad_t = block_sum(bad, s_red);
  if (tid == 0 && log_prob) log_prob[b] = bad_t > 0.0f ? -INFINITY : lp_t;
}

// The same draw with every node's logits held in registers: when every node has out-edges (groups == nodes, so a thread
// meets the same nodes in the scan and in the pick), at most 4 (8) of them, and N <= 4096 (2048) nodes — at most J = 4 (2)
// nodes per thread —,
// the three dependent rounds of loads (CSR range -> edge id -> logit) are issued once for all of a thread's nodes and
// everything after them — max, sum, probabilities, the double running sums, the block scans, pick and log-prob — runs
// out of registers. Same operations in the same order as k_graphdist_rollout (the generic path above).
// (J nodes x GDR_DEG edges of registers per thread: instantiated for J * GDR_DEG <= 16)
template <int J, int GDR_DEG, bool SORTED>
__global__ __launch_bounds__(ENV_BLOCK) void k_graphdist_rollout_reg(
    const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_eid, const float* __restrict__ logits, int64_t B,
    int64_t N, int64_t E, float temperature, const float* __restrict__ uniform, uint64_t seed, uint64_t counter,
    int32_t* __restrict__ choice_eid, uint8_t* __restrict__ choice8, uint8_t* __restrict__ sel8,
    float* __restrict__ log_prob, uint64_t idx_base) {
  __shared__ float s_red[ENV_BLOCK / 64];
  const int64_t b = blockIdx.x;
  const float* lb = logits + b * E;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int32_t k0[J], deg[J], eid[J][GDR_DEG];
  float p[J][GDR_DEG];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int64_t i = tid + (int64_t)ENV_BLOCK * j;
    k0[j] = i < N ? out_ptr[i] : 0;
    deg[j] = i < N ? out_ptr[i + 1] - k0[j] : 0;
  }
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int q = 0; q < GDR_DEG; ++q) eid[j][q] = SORTED ? k0[j] + q : (q < deg[j] ? out_eid[k0[j] + q] : 0);
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int q = 0; q < GDR_DEG; ++q) p[j][q] = q < deg[j] ? lb[eid[j][q]] / temperature : 0.0f;
  double base[J], inc[J];
  float un[J];
  // (block-uniform) the wave's uniforms come out of shared Philox blocks: device noise, and the environment's first index a
  // multiple of four (then so is every wave's first index of every chunk)
  const bool share = !uniform && J <= 4 && ((idx_base + (uint64_t)(b * N)) & 3u) == 0;
  __shared__ double s_wave_j[J][ENV_BLOCK / 64];
  // per-node softmax (exp evaluated once per edge: the value the chain recomputes is the same function of the same
  // argument), per-group double sums, wave-level inclusive scans of all J chunks of 1 024 groups ...
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int64_t g = tid + (int64_t)ENV_BLOCK * j;
    double s = 0.0;
    if (g < N) {
      float mx = -INFINITY;
#pragma unroll
      for (int q = 0; q < GDR_DEG; ++q)
        if (q < deg[j]) mx = fmaxf(mx, p[j][q]);
      float sum = 0.0f;
#pragma unroll
      for (int q = 0; q < GDR_DEG; ++q)
        if (q < deg[j]) {
          p[j][q] = expf(p[j][q] - mx);
          sum = sum + p[j][q];
        }
#pragma unroll
      for (int q = 0; q < GDR_DEG; ++q)
        if (q < deg[j]) {
          p[j][q] = p[j][q] / sum;
          s += (double)p[j][q];
        }
    }
    double v_inc = s;
    for (int off = 1; off < 64; off <<= 1) {
      const double v = __shfl_up(v_inc, off);
      if (lane >= off) v_inc += v;
    }
    if (lane == 63) s_wave_j[j][wid] = v_inc;
    inc[j] = v_inc;
    if (!share) un[j] = g < N ? (uniform ? uniform[b * N + g] : philox_uniform(seed, counter, idx_base + (uint64_t)(b * N + g))) : 0.0f;
  }
  if (share) {
    // One Philox block holds the uniforms of four consecutive indices, and chunk j of this wave draws the 64 consecutive
    // indices first + 1024 j + lane: 16 blocks per chunk. With the first index a multiple of four, lane L evaluates block
    // (L & 15) of chunk (L >> 4) — one generator pass per wave instead of one per chunk — and lane l takes word (l & 3) from
    // lane 16 j + (l >> 2). Same blocks, same words as philox_uniform(index).
    const uint64_t first = idx_base + (uint64_t)(b * N) + (uint64_t)(64 * wid);
    const int pj = lane >> 4;
    const uint64_t blk = ((first + (uint64_t)ENV_BLOCK * (uint64_t)(pj < J ? pj : 0)) >> 2) + (uint64_t)(lane & 15);
    uint32_t o[4];
    philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)seed,
                  (uint32_t)(seed >> 32), o);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int src = 16 * j + (lane >> 2);
      const uint32_t w0 = __shfl(o[0], src), w1 = __shfl(o[1], src), w2 = __shfl(o[2], src), w3 = __shfl(o[3], src);
      const int c = lane & 3;
      un[j] = u01_open(c == 0 ? w0 : (c == 1 ? w1 : (c == 2 ? w2 : w3)));
    }
  }
  __syncthreads();
  // ... then the cross-wave part of k_sample's scan, chunk after chunk (same additions in the same order)
  double running = 0.0;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    // wbase = v0 + ... + v(wid-1) and tot = v0 + ... + v15, both left to right: one running sum serves both
    double wbase = 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < ENV_BLOCK / 64; ++w) {
      if (w == wid) wbase = tot;
      tot += s_wave_j[j][w];
    }
    double exc = __shfl_up(inc[j], 1);
    if (lane == 0) exc = 0.0;
    base[j] = running + wbase + exc;
    running += tot;
  }
  float lp = 0.0f, bad = 0.0f;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int64_t i = tid + (int64_t)ENV_BLOCK * j;
    if (i >= N) continue;
    int32_t pick = -1, rank = 0;
    if (deg[j] != 0) {
      const double bg = base[j];
      const float bg32 = (float)bg, u = un[j];
      double run = bg;
      float lg_pick = 0.0f;
#pragma unroll
      for (int q = 0; q < GDR_DEG; ++q)
        if (q < deg[j]) {
          run += (double)p[j][q];
          const float cum = (float)run - bg32;
          if (pick < 0 && u < cum) {
            pick = eid[j][q];
            rank = q;
            lg_pick = logf(p[j][q] + LOG_EPS_P);
          }
        }
      if (pick >= 0)
        lp += lg_pick;
      else
        bad = 1.0f;
    }
    if (choice_eid) choice_eid[b * N + i] = pick;
    if (choice8 || sel8) {
      uint32_t code = (uint32_t)rank;
      if (pick < 0) code = ((sel8 ? sel8[i * B + b] : 0u) & 0x7Fu) | SEL_CARRIED;
      if (sel8) sel8[i * B + b] = (uint8_t)code;
      if (choice8) choice8[b * N + i] = (uint8_t)code;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const float lp_t = block_sum(lp, s_red);
  const float bad_t = block_sum(bad, s_red);
  if (tid == 0 && log_prob) log_prob[b] = bad_t > 0.0f ? -INFINITY : lp_t;
}

// ---- backward through the segment softmax ----------------------------------------------------------------------------
// L = gl*LP + ge*H with LP = sum a log(p+eps), H = -sum p log(p+eps), p = softmax(l/T) per group:
//   q_k = gl * a_k/(p_k+eps) + ge * (-log(p_k+eps) - p_k/(p_k+eps)),  dL/dl_j = p_j (q_j - sum_k p_k q_k) / T.
__global__ __launch_bounds__(DIST_BLOCK) void k_logprob_entropy_bwd(const int32_t* __restrict__ out_ptr,
                                                                    const int32_t* __restrict__ out_eid,
                                                                    const float* __restrict__ proba, int64_t B,
                                                                    int64_t N, int64_t E, float temperature,
                                                                    const int64_t* __restrict__ onehot,
                                                                    const int32_t* __restrict__ choice,
                                                                    const float* __restrict__ g_lp,
                                                                    const float* __restrict__ g_ent,
                                                                    const float* __restrict__ lp_fwd,
                                                                    float* __restrict__ grad) {
  const int64_t gid = (int64_t)blockIdx.x * DIST_BLOCK + threadIdx.x;
  if (gid >= B * N) return;
  const int64_t b = gid / N;
  const int32_t i = (int32_t)(gid - b * N);
  const int32_t k0 = out_ptr[i], k1 = out_ptr[i + 1];
  if (k0 == k1) return;
  const float* pb = proba + b * E;
  float gl = g_lp ? g_lp[b] : 0.0f;
  if (lp_fwd && isinf(lp_fwd[b])) gl = 0.0f;  // masked assignment of -inf in the reference: no gradient
  const float ge = g_ent ? g_ent[b] : 0.0f;
  const int32_t ce = choice ? choice[gid] : -1;
  float dot = 0.0f;
  for (int32_t k = k0; k < k1; ++k) {
    const int32_t e = out_eid[k];
    const float p = pb[e];
    const float a = onehot ? (float)onehot[b * E + e] : (e == ce ? 1.0f : 0.0f);
    const float q = gl * a / (p + LOG_EPS_P) + ge * (-logf(p + LOG_EPS_P) - p / (p + LOG_EPS_P));
    dot += p * q;
  }
  for (int32_t k = k0; k < k1; ++k) {
    const int32_t e = out_eid[k];
    const float p = pb[e];
    const float a = onehot ? (float)onehot[b * E + e] : (e == ce ? 1.0f : 0.0f);
    const float q = gl * a / (p + LOG_EPS_P) + ge * (-logf(p + LOG_EPS_P) - p / (p + LOG_EPS_P));
    grad[b * E + e] = p * (q - dot) / temperature;
  }
}

// ---- host side -------------------------------------------------------------------------------------------------------
extern "C" int tarl_graphdist_softmax(const tarl_plan* plan, const float* logits, int64_t B, float temperature,
                                      float* proba, tarl_stream stream) {
  TARL_REQUIRE(plan && logits && proba, "null argument");
  TARL_REQUIRE(B >= 1, "B must be positive");
  if (plan->N == 0 || plan->E == 0) return TARL_OK;
  hipLaunchKernelGGL(k_softmax, dim3((unsigned)ceil_div(B * plan->N, DIST_BLOCK)), dim3(DIST_BLOCK), 0,
                     (hipStream_t)stream, plan->out_ptr, plan->out_eid, logits, B, plan->N, plan->E, temperature,
                     proba);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_graphdist_sample(const tarl_plan* plan, const float* proba, int64_t B, const float* uniform,
                                     uint64_t seed, uint64_t counter, double* group_sums, int64_t* onehot,
                                     int32_t* choice, tarl_stream stream) {
  TARL_REQUIRE(plan && proba && group_sums, "null argument");
  TARL_REQUIRE(B >= 1 && B < ((int64_t)1 << 31), "bad B");
  if (plan->E == 0) return TARL_OK;
  hipStream_t s = (hipStream_t)stream;
  if (choice && plan->G != plan->N) {  // nodes without out-edges keep -1
    hipLaunchKernelGGL(k_fill_choice, dim3((unsigned)ceil_div(B * plan->N, DIST_BLOCK)), dim3(DIST_BLOCK), 0, s,
                       choice, B * plan->N);
    TARL_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_sample, dim3((unsigned)B), dim3(ENV_BLOCK), 0, s, plan->out_ptr, plan->out_eid,
                     plan->node_of_group, proba, plan->N, plan->E, plan->G, uniform, seed, counter, group_sums, onehot,
                     choice);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int64_t tarl_graphdist_rollout_scratch_bytes(const tarl_plan* plan, int64_t B) {
  return plan && B >= 1 ? B * plan->N * (int64_t)(sizeof(double) + sizeof(float)) : -1;
}

// env_base: global id of the batch's environment 0 (tarl_fused.env_base): the Philox index of (b, group g) is
// (env_base + b) * G + g
int tarl_graphdist_rollout_at(const tarl_plan* plan, const float* logits, int64_t B, float temperature,
                              const float* uniform, uint64_t seed, uint64_t counter, void* scratch, int32_t* choice,
                              uint8_t* choice8, uint8_t* sel8, float* log_prob, int64_t env_base, tarl_stream stream) {
  TARL_REQUIRE(plan && logits && scratch, "null argument");
  const uint64_t idx_base = (uint64_t)env_base * (uint64_t)plan->G;
  TARL_REQUIRE(B >= 1 && B < ((int64_t)1 << 31), "bad B");
  TARL_REQUIRE(temperature > 0.0f, "temperature must be positive");
  TARL_REQUIRE(((uintptr_t)scratch) % 8 == 0, "scratch must be 8-byte aligned");
  if (plan->N == 0) return TARL_OK;
  double* base = (double*)scratch;
  float* un = (float*)(base + B * plan->N);
  const char* knob = getenv("TARL_GRAPHDIST_REG");
  const int J = (int)ceil_div(plan->N, (int64_t)ENV_BLOCK);
  const int D = plan->max_out <= 4 ? 4 : 8;
  const bool reg = (!knob || atoi(knob) != 0) && plan->G == plan->N && plan->max_out <= 8 && J * D <= 16;
#define GDR_LAUNCH(J_, D_, S_)                                                                                          \
  hipLaunchKernelGGL((k_graphdist_rollout_reg<J_, D_, S_>), dim3((unsigned)B), dim3(ENV_BLOCK), 0, (hipStream_t)stream,   \
                     plan->out_ptr, plan->out_eid, logits, B, plan->N, plan->E, temperature, uniform, seed, counter, choice,  \
                     choice8, sel8, log_prob, idx_base)
#define GDR_BY_SORT(J_, D_)         \
  if (plan->src_sorted)             \
    GDR_LAUNCH(J_, D_, true);       \
  else                              \
    GDR_LAUNCH(J_, D_, false)
  if (reg) {
    if (D == 4) {
      switch (J) {
        case 1: GDR_BY_SORT(1, 4); break;
        case 2: GDR_BY_SORT(2, 4); break;
        case 3: GDR_BY_SORT(3, 4); break;
        default: GDR_BY_SORT(4, 4); break;
      }
    } else {
      if (J == 1) {
        GDR_BY_SORT(1, 8);
      } else {
        GDR_BY_SORT(2, 8);
      }
    }
  } else
    hipLaunchKernelGGL(k_graphdist_rollout, dim3((unsigned)B), dim3(ENV_BLOCK), 0, (hipStream_t)stream, plan->out_ptr,
                       plan->out_eid, plan->node_of_group, logits, B, plan->N, plan->E, plan->G, temperature, uniform, seed,
                       counter, base, un, choice, choice8, sel8, log_prob, idx_base);
#undef GDR_BY_SORT
#undef GDR_LAUNCH
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_graphdist_rollout(const tarl_plan* plan, const float* logits, int64_t B, float temperature,
                                      const float* uniform, uint64_t seed, uint64_t counter, void* scratch,
                                      int32_t* choice, uint8_t* choice8, uint8_t* sel8, float* log_prob,
                                      tarl_stream stream) {
  return tarl_graphdist_rollout_at(plan, logits, B, temperature, uniform, seed, counter, scratch, choice, choice8, sel8,
                                   log_prob, 0, stream);
}

extern "C" int tarl_graphdist_mode(const tarl_plan* plan, const float* proba, int64_t B, float* onehot,
                                   int32_t* choice, tarl_stream stream) {
  TARL_REQUIRE(plan && proba, "null argument");
  TARL_REQUIRE(B >= 1, "B must be positive");
  if (plan->N == 0) return TARL_OK;
  hipLaunchKernelGGL(k_mode, dim3((unsigned)ceil_div(B * plan->N, DIST_BLOCK)), dim3(DIST_BLOCK), 0,
                     (hipStream_t)stream, plan->out_ptr, plan->out_eid, proba, B, plan->N, plan->E, onehot, choice);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_graphdist_logprob_entropy_fwd(const tarl_plan* plan, const float* proba, int64_t B,
                                                  const int64_t* onehot, const int32_t* choice, float* log_prob,
                                                  float* entropy, tarl_stream stream) {
  TARL_REQUIRE(plan && proba, "null argument");
  TARL_REQUIRE(B >= 1 && B < ((int64_t)1 << 31), "bad B");
  TARL_REQUIRE(log_prob == nullptr || ((onehot != nullptr) != (choice != nullptr)),
               "log_prob needs exactly one of action_onehot / choice");
  hipLaunchKernelGGL(k_logprob_entropy_fwd, dim3((unsigned)B), dim3(ENV_BLOCK), 0, (hipStream_t)stream, plan->out_ptr,
                     plan->out_eid, proba, plan->N, plan->E, onehot, choice, log_prob, entropy);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_graphdist_logprob_entropy_bwd(const tarl_plan* plan, const float* proba, int64_t B,
                                                  float temperature, const int64_t* onehot, const int32_t* choice,
                                                  const float* grad_log_prob, const float* grad_entropy,
                                                  const float* log_prob_fwd, float* grad_logits, tarl_stream stream) {
  TARL_REQUIRE(plan && proba && grad_logits, "null argument");
  TARL_REQUIRE(B >= 1, "B must be positive");
  TARL_REQUIRE(grad_log_prob == nullptr || ((onehot != nullptr) != (choice != nullptr)),
               "grad_log_prob needs exactly one of action_onehot / choice");
  if (plan->N == 0 || plan->E == 0) return TARL_OK;
  hipLaunchKernelGGL(k_logprob_entropy_bwd, dim3((unsigned)ceil_div(B * plan->N, DIST_BLOCK)), dim3(DIST_BLOCK), 0,
                     (hipStream_t)stream, plan->out_ptr, plan->out_eid, proba, B, plan->N, plan->E, temperature, onehot,
                     choice, grad_log_prob, grad_entropy, log_prob_fwd, grad_logits);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}
