// policy.hip — the live MPNNPolicyNet forward: logits[e] = W_emb[road_index(dst(e))]
// (src/agents/mpnn_agent.py:175-178,215-217). The Dijkstra prior, travel time and norm computed after it in the
// reference (:181-190) are discarded there and therefore not evaluated here. An index outside
// [0, num_embeddings) (ROAD_INDEX = -1 on SRC/DEST pseudo-nodes raises IndexError in the reference) yields
// logit 0 and receives no gradient.
#include "tarl_common.h"

#define POL_BLOCK 256

__global__ __launch_bounds__(POL_BLOCK) void k_edge_logits_fwd(const int32_t* __restrict__ dst,
                                                               const float* __restrict__ road_index,
                                                               int64_t ri_bstride, int64_t ri_nstride, int64_t B,
                                                               int64_t E, const float* __restrict__ emb, int64_t M,
                                                               float* __restrict__ logits) {
  const int64_t gid = (int64_t)blockIdx.x * POL_BLOCK + threadIdx.x;
  if (gid >= B * E) return;
  const int64_t b = gid / E;
  const int32_t e = (int32_t)(gid - b * E);
  const long long idx = (long long)road_index[b * ri_bstride + (int64_t)dst[e] * ri_nstride];
  logits[gid] = (idx >= 0 && idx < M) ? emb[idx] : 0.0f;
}

// one thread per node n: the gradients of n's in-edges (ascending edge id) summed over the batch rows in row order. Rows
// that map n to the same embedding (always, while ROAD_INDEX is the static column it is) are accumulated in registers and
// flushed with ONE add per (node, embedding) run: with a one-to-one ROAD_INDEX every embedding receives a single add, so
// the gradient is bit-reproducible run to run (a per-(row, node) atomic was order-dependent in the last bit).
__global__ __launch_bounds__(POL_BLOCK) void k_edge_logits_bwd(const int32_t* __restrict__ in_ptr,
                                                               const int32_t* __restrict__ in_eid,
                                                               const float* __restrict__ road_index,
                                                               int64_t ri_bstride, int64_t ri_nstride, int64_t B,
                                                               int64_t N, int64_t E,
                                                               const float* __restrict__ grad_logits,
                                                               float* __restrict__ grad_emb, int64_t M) {
  const int64_t n = (int64_t)blockIdx.x * POL_BLOCK + threadIdx.x;
  if (n >= N) return;
  const int32_t k0 = in_ptr[n], k1 = in_ptr[n + 1];
  if (k0 == k1) return;
  long long cur = -1;
  float s = 0.0f;
  for (int64_t b = 0; b < B; ++b) {
    const long long idx = (long long)road_index[b * ri_bstride + n * ri_nstride];
    if (idx != cur) {
      if (cur >= 0 && cur < M) atomicAdd(&grad_emb[cur], s);
      cur = idx;
      s = 0.0f;
    }
    for (int32_t k = k0; k < k1; ++k) s += grad_logits[b * E + in_eid[k]];
  }
  if (cur >= 0 && cur < M) atomicAdd(&grad_emb[cur], s);
}

// MANY batch rows with ONE observation broadcast over them (ri_bstride == 0: the optimiser minibatch of the live policy, whose
// logits read the static ROAD_INDEX column alone): the serial loop over the rows above is cut into chunks of POL_RC rows that run
// in parallel — thread (n, chunk) sums n's in-edge gradients over the chunk's rows — and a second launch adds a node's partial
// sums in chunk order and issues the node's ONE add into its embedding. Same sum in another association; still one add per
// (node, embedding), so still bit-reproducible run to run. (B = 4 096, E = 10 000: 7.7 ms -> see bench.py's update_path.)
#define POL_RC 64
__global__ __launch_bounds__(POL_BLOCK) void k_edge_logits_bwd_part(const int32_t* __restrict__ in_ptr,
                                                                    const int32_t* __restrict__ in_eid, int64_t B,
                                                                    int64_t N, int64_t E,
                                                                    const float* __restrict__ grad_logits,
                                                                    float* __restrict__ part) {
  const int64_t n = (int64_t)blockIdx.x * POL_BLOCK + threadIdx.x;
  if (n >= N) return;
  const int32_t k0 = in_ptr[n], k1 = in_ptr[n + 1];
  const int64_t b0 = (int64_t)blockIdx.y * POL_RC, b1 = b0 + POL_RC < B ? b0 + POL_RC : B;
  float s = 0.0f;
  if (k1 - k0 == 4) {      // the common road-network shape: four in-edges, their ids in registers
    const int32_t e0 = in_eid[k0], e1 = in_eid[k0 + 1], e2 = in_eid[k0 + 2], e3 = in_eid[k0 + 3];
#pragma unroll 4
    for (int64_t b = b0; b < b1; ++b) {
      const float* g = grad_logits + b * E;
      s += g[e0];
      s += g[e1];
      s += g[e2];
      s += g[e3];
    }
  } else {
    for (int64_t b = b0; b < b1; ++b)
      for (int32_t k = k0; k < k1; ++k) s += grad_logits[b * E + in_eid[k]];
  }
  part[(int64_t)blockIdx.y * N + n] = s;
}

__global__ __launch_bounds__(POL_BLOCK) void k_edge_logits_bwd_reduce(const int32_t* __restrict__ in_ptr, int64_t S, int64_t N,
                                                                      const float* __restrict__ road_index, int64_t ri_nstride,
                                                                      const float* __restrict__ part,
                                                                      float* __restrict__ grad_emb, int64_t M) {
  const int64_t n = (int64_t)blockIdx.x * POL_BLOCK + threadIdx.x;
  if (n >= N || in_ptr[n] == in_ptr[n + 1]) return;
  float s = 0.0f;
  for (int64_t c = 0; c < S; ++c) s += part[c * N + n];
  const long long idx = (long long)road_index[n * ri_nstride];
  if (idx >= 0 && idx < M) atomicAdd(&grad_emb[idx], s);
}

extern "C" int64_t tarl_policy_edge_logits_bwd_scratch_floats(const tarl_plan* plan, int64_t B) {
  return plan && B >= 1 ? ceil_div(B, POL_RC) * plan->N : -1;
}

extern "C" int tarl_policy_edge_logits_fwd(const tarl_plan* plan, const float* road_index, int64_t ri_bstride,
                                           int64_t ri_nstride, int64_t B, const float* emb, int64_t M, float* logits,
                                           tarl_stream stream) {
  TARL_REQUIRE(plan && road_index && emb && logits, "null argument");
  TARL_REQUIRE(B >= 1 && M >= 1, "bad sizes");
  if (plan->E == 0) return TARL_OK;
  hipLaunchKernelGGL(k_edge_logits_fwd, dim3((unsigned)ceil_div(B * plan->E, POL_BLOCK)), dim3(POL_BLOCK), 0,
                     (hipStream_t)stream, plan->dst, road_index, ri_bstride, ri_nstride, B, plan->E, emb, M, logits);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_policy_edge_logits_bwd(const tarl_plan* plan, const float* road_index, int64_t ri_bstride,
                                           int64_t ri_nstride, int64_t B, const float* grad_logits, float* grad_emb,
                                           int64_t M, float* scratch, tarl_stream stream) {
  TARL_REQUIRE(plan && road_index && grad_logits && grad_emb, "null argument");
  TARL_REQUIRE(B >= 1 && M >= 1, "bad sizes");
  if (plan->E == 0 || plan->N == 0) return TARL_OK;
  if (scratch && ri_bstride == 0 && B >= 4 * POL_RC) {      // many rows, one broadcast observation: chunked over the rows
    const int64_t S = ceil_div(B, POL_RC);
    TARL_REQUIRE(S < 65536, "too many rows for one launch");
    hipLaunchKernelGGL(k_edge_logits_bwd_part, dim3((unsigned)ceil_div(plan->N, POL_BLOCK), (unsigned)S), dim3(POL_BLOCK), 0,
                       (hipStream_t)stream, plan->in_ptr, plan->in_eid, B, plan->N, plan->E, grad_logits, scratch);
    TARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_edge_logits_bwd_reduce, dim3((unsigned)ceil_div(plan->N, POL_BLOCK)), dim3(POL_BLOCK), 0,
                       (hipStream_t)stream, plan->in_ptr, S, plan->N, road_index, ri_nstride, scratch, grad_emb, M);
    TARL_LAUNCH_CHECK();
    return TARL_OK;
  }
  hipLaunchKernelGGL(k_edge_logits_bwd, dim3((unsigned)ceil_div(plan->N, POL_BLOCK)), dim3(POL_BLOCK), 0,
                     (hipStream_t)stream, plan->in_ptr, plan->in_eid, road_index, ri_bstride, ri_nstride, B, plan->N,
                     plan->E, grad_logits, grad_emb, M);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}
