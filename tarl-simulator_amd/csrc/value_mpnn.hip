// value_mpnn.hip — MPNNValueNet (reference: src/agents/mpnn_agent.py:265-402), the message-passing critic the reference
// defines next to the live MPNNValueNetSimple (its runner never instantiates it; built for API completeness).
//
//   x_v      = [node_features_v (7), agent_features[agent_index_v] (9)]                       (16 per node)
//   m_e      = tanh(W_m . [x_dst(e), edge_attr_e] + b_m)            message of edge e = (u -> v), flow target_to_source
//   a_u      = mean of m_e over the out-edges of u (0 when none)    aggregation at the source, original edge order
//   n_u      = tanh(w_n * a_u + b_n)                                node_mlp
//   time_emb = Linear(1,32)-ReLU-Linear(32,32)-ReLU-Linear(32,1)(time)
//   value    = W_f . [n_0 .. n_{N-1}, time_emb] + b_f
// Evaluation-mode semantics: the three Dropout(0.05) layers are the identity. One 256-thread workgroup per sample;
// floating point: same formulas as the torch modules, summation order of the 17-term dot product and of the final
// reduction differs from a BLAS GEMV => parity to 1e-5 relative, not bit-for-bit (tests/test_gpu_value_mpnn.py).
#include <math.h>

#include "tarl_common.h"

#define VM_THREADS 256
#define VM_WAVES (VM_THREADS / 64)

struct VmParams {
  const float* w_msg;    // [17]
  const float* b_msg;    // [1]
  const float* w_node;   // [1]
  const float* b_node;   // [1]
  const float* w_final;  // [N + 1]
  const float* b_final;  // [1]
  const float* t_w1;     // [32]      time_net.0 (32 x 1)
  const float* t_b1;     // [32]
  const float* t_w2;     // [32][32]  time_net.3
  const float* t_b2;     // [32]
  const float* t_w3;     // [32]      time_net.6 (1 x 32)
  const float* t_b3;     // [1]
};
struct VmGrads {
  float *w_msg, *b_msg, *w_node, *b_node, *w_final, *b_final, *t_w1, *t_b1, *t_w2, *t_b2, *t_w3, *t_b3;
};

__device__ __forceinline__ float vm_block_sum(float v, float* s_red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  __syncthreads();
  if (lane == 0) s_red[wid] = v;
  __syncthreads();
  float tot = 0.0f;
  for (int w = 0; w < VM_WAVES; ++w) tot += s_red[w];
  return tot;
}

// pre-activation of the message of CSR position k (edge u -> v): W_m . [nf_v, af_v, edge_attr] + b_m
__device__ __forceinline__ float vm_message_pre(const VmParams& P, const float* __restrict__ nf,
                                                const float* __restrict__ af, const float* __restrict__ ef, int32_t v,
                                                int32_t e) {
  float pre = P.b_msg[0];
  const float* a = nf + (int64_t)v * 7;
#pragma unroll
  for (int c = 0; c < 7; ++c) pre += P.w_msg[c] * a[c];
  if (af) {
    const float* g = af + (int64_t)v * 9;
#pragma unroll
    for (int c = 0; c < 9; ++c) pre += P.w_msg[7 + c] * g[c];
  }
  pre += P.w_msg[16] * ef[e];
  return pre;
}

// time_net forward by the first wave; h1 / h2 left in LDS, result returned to every thread
__device__ __forceinline__ float vm_time_net(const VmParams& P, float t, float* s_h1, float* s_h2, float* s_te) {
  const int tid = threadIdx.x;
  if (tid < 32) s_h1[tid] = fmaxf(P.t_w1[tid] * t + P.t_b1[tid], 0.0f);
  __syncthreads();
  if (tid < 32) {
    float acc = P.t_b2[tid];
    for (int i = 0; i < 32; ++i) acc += P.t_w2[tid * 32 + i] * s_h1[i];
    s_h2[tid] = fmaxf(acc, 0.0f);
  }
  __syncthreads();
  if (tid == 0) {
    float acc = P.t_b3[0];
    for (int j = 0; j < 32; ++j) acc += P.t_w3[j] * s_h2[j];
    s_te[0] = acc;
  }
  __syncthreads();
  return s_te[0];
}

__global__ __launch_bounds__(VM_THREADS) void k_value_mpnn_fwd(const int32_t* __restrict__ out_ptr,
                                                               const int32_t* __restrict__ out_dst,
                                                               const int32_t* __restrict__ out_eid, int64_t N, int64_t E,
                                                               const float* __restrict__ node_features,
                                                               const float* __restrict__ agent_rows,
                                                               const float* __restrict__ edge_features,
                                                               int64_t ef_mstride, const float* __restrict__ time,
                                                               VmParams P, float* __restrict__ value,
                                                               float* __restrict__ node_act, float* __restrict__ agg) {
  __shared__ float s_red[VM_WAVES], s_h1[32], s_h2[32], s_te[1];
  const int64_t m = blockIdx.x;
  const float* nf = node_features + m * N * 7;
  const float* af = agent_rows ? agent_rows + m * N * 9 : nullptr;
  const float* ef = edge_features + m * ef_mstride;
  float acc = 0.0f;
  for (int64_t u = threadIdx.x; u < N; u += VM_THREADS) {
    const int32_t k0 = out_ptr[u], k1 = out_ptr[u + 1];
    float sum = 0.0f;
    for (int32_t k = k0; k < k1; ++k) sum += tanhf(vm_message_pre(P, nf, af, ef, out_dst[k], out_eid[k]));
    const float a = k1 > k0 ? sum / (float)(k1 - k0) : 0.0f;
    const float nd = tanhf(P.w_node[0] * a + P.b_node[0]);
    if (agg) agg[m * N + u] = a;
    if (node_act) node_act[m * N + u] = nd;
    acc += P.w_final[u] * nd;
  }
  const float te = vm_time_net(P, time[m], s_h1, s_h2, s_te);
  const float tot = vm_block_sum(acc, s_red);
  if (threadIdx.x == 0) value[m] = tot + P.w_final[N] * te + P.b_final[0];
}

__global__ __launch_bounds__(VM_THREADS) void k_value_mpnn_bwd(const int32_t* __restrict__ out_ptr,
                                                               const int32_t* __restrict__ out_dst,
                                                               const int32_t* __restrict__ out_eid, int64_t N, int64_t E,
                                                               const float* __restrict__ node_features,
                                                               const float* __restrict__ agent_rows,
                                                               const float* __restrict__ edge_features,
                                                               int64_t ef_mstride, const float* __restrict__ time,
                                                               VmParams P, const float* __restrict__ dvalue,
                                                               const float* __restrict__ node_act,
                                                               const float* __restrict__ agg, VmGrads G) {
  __shared__ float s_red[VM_WAVES], s_h1[32], s_h2[32], s_te[1], s_dh2[32];
  const int64_t m = blockIdx.x;
  const int tid = threadIdx.x;
  const float* nf = node_features + m * N * 7;
  const float* af = agent_rows ? agent_rows + m * N * 9 : nullptr;
  const float* ef = edge_features + m * ef_mstride;
  const float dv = dvalue[m];
  float g_wm[17];
#pragma unroll
  for (int c = 0; c < 17; ++c) g_wm[c] = 0.0f;
  float g_bm = 0.0f, g_wn = 0.0f, g_bn = 0.0f;
  for (int64_t u = tid; u < N; u += VM_THREADS) {
    const float nd = node_act[m * N + u], a = agg[m * N + u];
    atomicAdd(&G.w_final[u], dv * nd);
    const float dpre_n = dv * P.w_final[u] * (1.0f - nd * nd);
    g_wn += dpre_n * a;
    g_bn += dpre_n;
    const int32_t k0 = out_ptr[u], k1 = out_ptr[u + 1];
    if (k1 > k0) {
      const float dm = dpre_n * P.w_node[0] / (float)(k1 - k0);
      for (int32_t k = k0; k < k1; ++k) {
        const int32_t v = out_dst[k], e = out_eid[k];
        const float me = tanhf(vm_message_pre(P, nf, af, ef, v, e));
        const float dpre = dm * (1.0f - me * me);
        const float* xa = nf + (int64_t)v * 7;
#pragma unroll
        for (int c = 0; c < 7; ++c) g_wm[c] += dpre * xa[c];
        if (af) {
          const float* xg = af + (int64_t)v * 9;
#pragma unroll
          for (int c = 0; c < 9; ++c) g_wm[7 + c] += dpre * xg[c];
        }
        g_wm[16] += dpre * ef[e];
        g_bm += dpre;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 17; ++c) {
    const float tot = vm_block_sum(g_wm[c], s_red);
    if (tid == 0) atomicAdd(&G.w_msg[c], tot);
  }
  {
    const float t0 = vm_block_sum(g_bm, s_red);
    const float t1 = vm_block_sum(g_wn, s_red);
    const float t2 = vm_block_sum(g_bn, s_red);
    if (tid == 0) {
      atomicAdd(&G.b_msg[0], t0);
      atomicAdd(&G.w_node[0], t1);
      atomicAdd(&G.b_node[0], t2);
    }
  }
  // time_net and the final layer's time / bias terms
  const float t = time[m];
  const float te = vm_time_net(P, t, s_h1, s_h2, s_te);
  const float dte = dv * P.w_final[N];
  if (tid == 0) {
    atomicAdd(&G.w_final[N], dv * te);
    atomicAdd(&G.b_final[0], dv);
    atomicAdd(&G.t_b3[0], dte);
  }
  if (tid < 32) {
    atomicAdd(&G.t_w3[tid], dte * s_h2[tid]);
    const float dh2 = s_h2[tid] > 0.0f ? dte * P.t_w3[tid] : 0.0f;
    s_dh2[tid] = dh2;
    atomicAdd(&G.t_b2[tid], dh2);
    for (int i = 0; i < 32; ++i) atomicAdd(&G.t_w2[tid * 32 + i], dh2 * s_h1[i]);
  }
  __syncthreads();
  if (tid < 32) {
    float dh1 = 0.0f;
    for (int j = 0; j < 32; ++j) dh1 += s_dh2[j] * P.t_w2[j * 32 + tid];
    dh1 = s_h1[tid] > 0.0f ? dh1 : 0.0f;
    atomicAdd(&G.t_w1[tid], dh1 * t);
    atomicAdd(&G.t_b1[tid], dh1);
  }
}

static int vm_check(const tarl_plan* plan, const float* nf, int64_t M, const float* ef, const float* time,
                    const float* const* params) {
  TARL_REQUIRE(plan && nf && ef && time && params, "null argument");
  TARL_REQUIRE(M >= 1 && M < ((int64_t)1 << 31), "bad batch size");
  for (int i = 0; i < 12; ++i) TARL_REQUIRE(params[i] != nullptr, "parameter pointer is null");
  return TARL_OK;
}

extern "C" int tarl_value_mpnn_fwd(const tarl_plan* plan, const float* node_features, int64_t M,
                                   const float* agent_rows, const float* edge_features, int64_t ef_mstride,
                                   const float* time, const float* const* params, float* value, float* node_act,
                                   float* agg, tarl_stream stream) {
  int rc = vm_check(plan, node_features, M, edge_features, time, params);
  if (rc) return rc;
  TARL_REQUIRE(value != nullptr, "value is null");
  TARL_REQUIRE(ef_mstride == 0 || ef_mstride >= plan->E, "edge feature stride smaller than the edge count");
  const VmParams P{params[0], params[1], params[2], params[3], params[4], params[5],
                   params[6], params[7], params[8], params[9], params[10], params[11]};
  hipLaunchKernelGGL(k_value_mpnn_fwd, dim3((unsigned)M), dim3(VM_THREADS), 0, (hipStream_t)stream, plan->out_ptr,
                     plan->out_dst, plan->out_eid, plan->N, plan->E, node_features, agent_rows, edge_features, ef_mstride,
                     time, P, value, node_act, agg);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_value_mpnn_bwd(const tarl_plan* plan, const float* node_features, int64_t M,
                                   const float* agent_rows, const float* edge_features, int64_t ef_mstride,
                                   const float* time, const float* const* params, const float* grad_value,
                                   const float* node_act, const float* agg, float* const* grads, tarl_stream stream) {
  int rc = vm_check(plan, node_features, M, edge_features, time, params);
  if (rc) return rc;
  TARL_REQUIRE(grad_value && node_act && agg && grads, "null argument");
  for (int i = 0; i < 12; ++i) TARL_REQUIRE(grads[i] != nullptr, "gradient pointer is null");
  TARL_REQUIRE(ef_mstride == 0 || ef_mstride >= plan->E, "edge feature stride smaller than the edge count");
  const VmParams P{params[0], params[1], params[2], params[3], params[4], params[5],
                   params[6], params[7], params[8], params[9], params[10], params[11]};
  const VmGrads G{grads[0], grads[1], grads[2], grads[3], grads[4], grads[5],
                  grads[6], grads[7], grads[8], grads[9], grads[10], grads[11]};
  hipLaunchKernelGGL(k_value_mpnn_bwd, dim3((unsigned)M), dim3(VM_THREADS), 0, (hipStream_t)stream, plan->out_ptr,
                     plan->out_dst, plan->out_eid, plan->N, plan->E, node_features, agent_rows, edge_features, ef_mstride,
                     time, P, grad_value, node_act, agg, G);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}
