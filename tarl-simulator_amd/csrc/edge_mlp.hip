// edge_mlp.hip — the per-edge MLP head of MPNNPolicyNet (src/agents/mpnn_agent.py:35-41, evaluated as its
// update_edges spells out at :227-231):
//     e_ij   = cat(x[src(e)], x[dst(e)], edge_attr[e])                (33 = 16 + 16 + 1)
//     logit  = W3 relu(W2 relu(W1 e_ij + b1) + b2) + b3               (33 -> 64 -> 32 -> 1)
// with x = cat(node_features (7), agent_features[agent_index] (9)) per node (:166-178). The reference keeps this head
// as parameters (state-dict keys edge_mlp.{0,2,4}.*) and leaves the call commented out; this is the state-DEPENDENT
// policy of the build (`policy_head="edge_mlp"`): logits change with every frame, so nothing of GraphDistribution can be
// hoisted out of the rollout.
//
// Forward = the one GEMM-shaped op of the policy, hence on the matrix cores: a workgroup owns 128 edges of one sample,
// gathers their 33 inputs into LDS, and runs both hidden layers as MFMA tiles out of LDS (the 128 x 64 hidden tile never
// leaves the CU); layer 3 is a 32-long dot product per edge.
//   * fp32: v_mfma_f32_32x32x2_f32 — exact fp32 products, k-ordered fma chain (parity contract 1e-4 on the logits);
//   * bf16: v_mfma_f32_32x32x16_bf16 — inputs, weights and the first hidden activation rounded to bf16 (RNE), fp32
//     accumulation, second hidden layer and output in fp32: BASELINE config 5's "bf16 MPNN features" (tolerance stated
//     in tests/test_gpu_edge_mlp.py).
// Backward (minibatch-sized: sub_batch x E edges, once per optimiser step) recomputes the activations per edge on the
// vector ALU, stores them k-major, and reduces the weight gradients as long dot products in a fixed order
// (deterministic; no atomics). Inputs are observations: they need no gradient.
#include "fused_common.h"

#define EM_TILE 128     // edges per workgroup
#define EM_THREADS 256
#define EM_IN 33
#define EM_H1 64
#define EM_H2 32

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct EdgeMlpW {
  const float* w1;  // [64][33]
  const float* b1;  // [64]
  const float* w2;  // [32][64]
  const float* b2;  // [32]
  const float* w3;  // [32]
  const float* b3;  // [1]
};

// C/D layout of the 32x32 MFMAs: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int em_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// ---- obs16: x = cat(node_features, agent_features[agent_index]) from the packed state of the fused engine ----------------
// node_features = x[:, 3*Nmax:] = {MAX, NUMBER_OF_AGENT, FREE_FLOW, LENGTH, MAX_FLOW, SELECTED_ROAD, ROAD_INDEX}
// (TransportationSimulator.state, src/transportation_simulator.py:360-366); agent_index = the head-of-FIFO id.
__global__ __launch_bounds__(FB) void k_obs16_packed(const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_dst,
                                                     const float* __restrict__ x0, Layout L, int64_t B, int64_t N,
                                                     FusedBufs fb, const float* __restrict__ ag, int64_t A,
                                                     int64_t a_bstride, float* __restrict__ obs) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;   // gid = i * B + b (env-minor reads)
  if (gid >= B * N) return;
  const int64_t i = gid / B, b = gid - i * B;
  const uint32_t hd = fb.hdp[gid].x;
  const float4 st = fb.st0[i];
  const float* xs = x0 + i * L.ldx;     // static columns: environment 0 speaks for all
  float* o = obs + (b * N + i) * 16;
  const long long head = (long long)(hd >> 8);
  const float* arow = ag + b * a_bstride + ((head >= 0 && head < A) ? head : 0) * AG_COLS;
  float4 v0 = make_float4(st.x, (float)(hd & 255u), st.y, xs[L.col_maxn() + 3]);
  float4 v1 = make_float4(xs[L.col_maxflow()], sel_value(fb, out_ptr, out_dst, i, gid), st.z, arow[0]);
  float4 v2 = make_float4(arow[1], arow[2], arow[3], arow[4]);
  float4 v3 = make_float4(arow[5], arow[6], arow[7], arow[8]);
  reinterpret_cast<float4*>(o)[0] = v0;
  reinterpret_cast<float4*>(o)[1] = v1;
  reinterpret_cast<float4*>(o)[2] = v2;
  reinterpret_cast<float4*>(o)[3] = v3;
}

// the same from the reference's tensors: node_features (M, N, >=7 cols, row stride nf_ld) + agent rows gathered by
// agent_index (int64 (M, N)) from agent_features (A, 9) (one population) or (M, A, 9)
__global__ __launch_bounds__(FB) void k_obs16_cat(const float* __restrict__ nf, int64_t nf_ld, const int64_t* __restrict__ aidx,
                                                  const float* __restrict__ ag, int64_t A, int64_t a_mstride, int64_t M,
                                                  int64_t N, float* __restrict__ obs) {
  const int64_t gid = (int64_t)blockIdx.x * FB + threadIdx.x;   // gid = m * N + i
  if (gid >= M * N) return;
  const int64_t m = gid / N;
  const float* r = nf + gid * nf_ld;
  const int64_t head = aidx[gid];
  const float* arow = ag + m * a_mstride + ((head >= 0 && head < A) ? head : 0) * AG_COLS;
  float* o = obs + gid * 16;
  for (int c = 0; c < 7; ++c) o[c] = r[c];
  for (int c = 0; c < 9; ++c) o[7 + c] = arow[c];
}

// ---- forward ------------------------------------------------------------------------------------------------------------
// LDS plan (fp32, dynamic, 67.7 KB): Xs [34][129] (later H2s [32][129]) | Hs [64][129] | W1s [34][65] | W2s [64][33].
// A workgroup stages the weights ONCE and walks EM_TPW consecutive 128-edge tiles of its sample.
#define EM_TPW 8
#define EMF_A (34 * (EM_TILE + 1))
#define EMF_C (EM_H1 * (EM_TILE + 1))
#define EMF_W1 (34 * (EM_H1 + 1))
#define EMF_W2 (EM_H1 * (EM_H2 + 1))
#define EMF_LDS_BYTES ((EMF_A + EMF_C + EMF_W1 + EMF_W2) * sizeof(float))
__global__ __launch_bounds__(EM_THREADS) void k_edge_mlp_fwd_f32(const int32_t* __restrict__ src,
                                                                 const int32_t* __restrict__ dst, int64_t E, int64_t N,
                                                                 const float* __restrict__ obs,
                                                                 const float* __restrict__ edge_attr, EdgeMlpW W,
                                                                 float* __restrict__ logits) {
  extern __shared__ float lds[];
  float* Xs = lds;                   // [k][edge]
  float* Hs = lds + EMF_A;           // [j][edge]
  float* W1s = Hs + EMF_C;           // [k][j]
  float* W2s = W1s + EMF_W1;         // [k][j], stride 33
  float* H2s = Xs;                   // [j][edge] (aliases Xs: dead after the first layer)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m = blockIdx.y;
  const float* om = obs + m * N * 16;
  for (int idx = tid; idx < 34 * EM_H1; idx += EM_THREADS) {
    const int j = idx / 34, k = idx - j * 34;
    W1s[k * (EM_H1 + 1) + j] = k < EM_IN ? W.w1[j * EM_IN + k] : 0.0f;
  }
  for (int idx = tid; idx < EM_H2 * EM_H1; idx += EM_THREADS) {
    const int j = idx >> 6, k = idx & 63;
    W2s[k * (EM_H2 + 1) + j] = W.w2[j * EM_H1 + k];
  }
  const int j0 = lane & 31;
  const float bb0 = W.b1[j0], bb1 = W.b1[j0 + 32], bb2 = W.b2[j0];
  for (int tile = 0; tile < EM_TPW; ++tile) {
    const int64_t e0 = ((int64_t)blockIdx.x * EM_TPW + tile) * EM_TILE;
    if (e0 >= E) break;        // uniform
    __syncthreads();           // the previous tile's readers of H2s (= Xs) are done; the weights are staged
    // gather: 128 edges x 8 float4 (x_i: 4, x_j: 4), transposed into Xs[k][edge]
#pragma unroll
    for (int it = 0; it < (EM_TILE * 8) / EM_THREADS; ++it) {
      const int idx = it * EM_THREADS + tid;
      const int el = idx >> 3, q = idx & 7;     // q < 4: x[src] quarter q ; else x[dst] quarter q - 4
      const int64_t e = e0 + el;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < E) {
        const int32_t node = q < 4 ? src[e] : dst[e];
        v = reinterpret_cast<const float4*>(om + (int64_t)node * 16)[q & 3];
      }
      const int k = 4 * q;
      Xs[(k + 0) * (EM_TILE + 1) + el] = v.x;
      Xs[(k + 1) * (EM_TILE + 1) + el] = v.y;
      Xs[(k + 2) * (EM_TILE + 1) + el] = v.z;
      Xs[(k + 3) * (EM_TILE + 1) + el] = v.w;
    }
    if (tid < EM_TILE) {
      const int64_t e = e0 + tid;
      Xs[32 * (EM_TILE + 1) + tid] = e < E ? edge_attr[e] : 0.0f;
      Xs[33 * (EM_TILE + 1) + tid] = 0.0f;      // K padded to an even number of MFMA k-steps
    }
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
    for (int kk = 0; kk < 34; kk += 2) {
      const int k = kk + (lane >> 5);
      const float a = Xs[k * (EM_TILE + 1) + wave * 32 + (lane & 31)];
      const float b0 = W1s[k * (EM_H1 + 1) + (lane & 31)];
      const float b1 = W1s[k * (EM_H1 + 1) + 32 + (lane & 31)];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {   // each wave writes / reads only its own 32 edge columns of Hs
      const int lr = wave * 32 + em_row(r, lane);
      const float v0 = acc0[r] + bb0, v1 = acc1[r] + bb1;
      Hs[j0 * (EM_TILE + 1) + lr] = v0 > 0.0f ? v0 : 0.0f;
      Hs[(j0 + 32) * (EM_TILE + 1) + lr] = v1 > 0.0f ? v1 : 0.0f;
    }
    __syncthreads();   // Hs complete; everyone is done with Xs: its space becomes H2s
    f32x16 c0 = {0};
#pragma unroll
    for (int kk = 0; kk < EM_H1; kk += 2) {
      const int k = kk + (lane >> 5);
      const float a = Hs[k * (EM_TILE + 1) + wave * 32 + (lane & 31)];
      const float b = W2s[k * (EM_H2 + 1) + (lane & 31)];
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = wave * 32 + em_row(r, lane);
      const float v = c0[r] + bb2;
      H2s[j0 * (EM_TILE + 1) + lr] = v > 0.0f ? v : 0.0f;
    }
    __syncthreads();
    if (tid < EM_TILE) {
      const int64_t e = e0 + tid;
      if (e < E) {
        float sacc = 0.0f;
#pragma unroll 8
        for (int j = 0; j < EM_H2; ++j) sacc += H2s[j * (EM_TILE + 1) + tid] * W.w3[j];
        logits[m * E + e] = sacc + W.b3[0];
      }
    }
  }
}

// bf16 variant. LDS: Xb [128][56] bf16 | W1b [64][56] bf16 | Hb [128][72] bf16 | W2b [32][72] bf16 | H2s [32][129] f32
#define EMB_KX 56    // row stride of the 33-wide operands (48 used: 3 k-steps of 16), 16-byte aligned rows
#define EMB_KH 72    // row stride of the 64-wide operands
__global__ __launch_bounds__(EM_THREADS) void k_edge_mlp_fwd_bf16(const int32_t* __restrict__ src,
                                                                  const int32_t* __restrict__ dst, int64_t E, int64_t N,
                                                                  const float* __restrict__ obs,
                                                                  const float* __restrict__ edge_attr, EdgeMlpW W,
                                                                  float* __restrict__ logits) {
  __shared__ __attribute__((aligned(16))) uint16_t Xb[EM_TILE * EMB_KX];
  __shared__ __attribute__((aligned(16))) uint16_t W1b[EM_H1 * EMB_KX];
  __shared__ __attribute__((aligned(16))) uint16_t Hb[EM_TILE * EMB_KH];
  __shared__ __attribute__((aligned(16))) uint16_t W2b[EM_H2 * EMB_KH];
  __shared__ float H2s[EM_H2 * (EM_TILE + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m = blockIdx.y;
  const float* om = obs + m * N * 16;
  for (int idx = tid; idx < EM_H1 * 48; idx += EM_THREADS) {
    const int j = idx / 48, k = idx - j * 48;
    W1b[j * EMB_KX + k] = k < EM_IN ? f32_to_bf16_rne(W.w1[j * EM_IN + k]) : (uint16_t)0;
  }
  for (int idx = tid; idx < EM_H2 * EM_H1; idx += EM_THREADS) {
    const int j = idx >> 6, k = idx & 63;
    W2b[j * EMB_KH + k] = f32_to_bf16_rne(W.w2[j * EM_H1 + k]);
  }
  // lane l (r = l & 31, h = l >> 5) holds A[row r][k = 16 s + 8 h + 0..7] and B[k = 16 s + 8 h + 0..7][col r]
  const int r32 = lane & 31, h8 = (lane >> 5) * 8;
  const float bb0 = W.b1[r32], bb1 = W.b1[r32 + 32], bb2 = W.b2[r32];
  for (int tile = 0; tile < EM_TPW; ++tile) {
    const int64_t e0 = ((int64_t)blockIdx.x * EM_TPW + tile) * EM_TILE;
    if (e0 >= E) break;        // uniform
    __syncthreads();           // the previous tile's readers are done; the weights are staged
#pragma unroll
    for (int it = 0; it < (EM_TILE * 8) / EM_THREADS; ++it) {
      const int idx = it * EM_THREADS + tid;
      const int el = idx >> 3, q = idx & 7;
      const int64_t e = e0 + el;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < E) {
        const int32_t node = q < 4 ? src[e] : dst[e];
        v = reinterpret_cast<const float4*>(om + (int64_t)node * 16)[q & 3];
      }
      uint16_t* d = Xb + el * EMB_KX + 4 * q;
      d[0] = f32_to_bf16_rne(v.x);
      d[1] = f32_to_bf16_rne(v.y);
      d[2] = f32_to_bf16_rne(v.z);
      d[3] = f32_to_bf16_rne(v.w);
    }
    if (tid < EM_TILE) {
      const int64_t e = e0 + tid;
      uint16_t* d = Xb + tid * EMB_KX;
      d[32] = f32_to_bf16_rne(e < E ? edge_attr[e] : 0.0f);
      for (int k = 33; k < 48; ++k) d[k] = 0;
    }
    __syncthreads();
    f32x16 acc0 = {0}, acc1 = {0};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Xb + (wave * 32 + r32) * EMB_KX + 16 * s + h8);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(W1b + r32 * EMB_KX + 16 * s + h8);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(W1b + (32 + r32) * EMB_KX + 16 * s + h8);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {   // a wave's rows of Hb are its own 32 edges
      const int lr = wave * 32 + em_row(r, lane);
      const float v0 = acc0[r] + bb0, v1 = acc1[r] + bb1;
      Hb[lr * EMB_KH + r32] = f32_to_bf16_rne(v0 > 0.0f ? v0 : 0.0f);
      Hb[lr * EMB_KH + 32 + r32] = f32_to_bf16_rne(v1 > 0.0f ? v1 : 0.0f);
    }
    __syncthreads();
    f32x16 c0 = {0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Hb + (wave * 32 + r32) * EMB_KH + 16 * s + h8);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(W2b + r32 * EMB_KH + 16 * s + h8);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = wave * 32 + em_row(r, lane);
      const float v = c0[r] + bb2;
      H2s[r32 * (EM_TILE + 1) + lr] = v > 0.0f ? v : 0.0f;
    }
    __syncthreads();
    if (tid < EM_TILE) {
      const int64_t e = e0 + tid;
      if (e < E) {
        float sacc = 0.0f;
#pragma unroll 8
        for (int j = 0; j < EM_H2; ++j) sacc += H2s[j * (EM_TILE + 1) + tid] * W.w3[j];
        logits[m * E + e] = sacc + W.b3[0];
      }
    }
  }
}

// ---- backward -------------------------------------------------------------------------------------------------------------
// stage 1: one thread per (sample, edge): recompute h1, h2; dh2 = g W3 (masked), dh1 = W2^T dh2 (masked); everything is
// stored k-major ([k][L], L = M * E) for the reductions of stage 2.
__global__ __launch_bounds__(EM_THREADS) void k_edge_mlp_bwd_edges(const int32_t* __restrict__ src,
                                                                   const int32_t* __restrict__ dst, int64_t E, int64_t N,
                                                                   int64_t M, const float* __restrict__ obs,
                                                                   const float* __restrict__ edge_attr, EdgeMlpW W,
                                                                   const float* __restrict__ grad_logits,
                                                                   float* __restrict__ XT, float* __restrict__ H1T,
                                                                   float* __restrict__ H2T, float* __restrict__ D1T,
                                                                   float* __restrict__ D2T) {
  __shared__ float w1s[EM_H1 * EM_IN], w2s[EM_H2 * EM_H1], w3s[EM_H2], b1s[EM_H1], b2s[EM_H2];
  for (int i = threadIdx.x; i < EM_H1 * EM_IN; i += EM_THREADS) w1s[i] = W.w1[i];
  for (int i = threadIdx.x; i < EM_H2 * EM_H1; i += EM_THREADS) w2s[i] = W.w2[i];
  if (threadIdx.x < EM_H2) {
    w3s[threadIdx.x] = W.w3[threadIdx.x];
    b2s[threadIdx.x] = W.b2[threadIdx.x];
  }
  if (threadIdx.x < EM_H1) b1s[threadIdx.x] = W.b1[threadIdx.x];
  __syncthreads();
  const int64_t L = M * E;
  const int64_t ge = (int64_t)blockIdx.x * EM_THREADS + threadIdx.x;
  if (ge >= L) return;
  const int64_t m = ge / E, e = ge - m * E;
  const float* om = obs + m * N * 16;
  float x[EM_IN];
  {
    const float4* xi = reinterpret_cast<const float4*>(om + (int64_t)src[e] * 16);
    const float4* xj = reinterpret_cast<const float4*>(om + (int64_t)dst[e] * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 a = xi[q], b = xj[q];
      x[4 * q + 0] = a.x; x[4 * q + 1] = a.y; x[4 * q + 2] = a.z; x[4 * q + 3] = a.w;
      x[16 + 4 * q + 0] = b.x; x[16 + 4 * q + 1] = b.y; x[16 + 4 * q + 2] = b.z; x[16 + 4 * q + 3] = b.w;
    }
    x[32] = edge_attr[e];
  }
#pragma unroll
  for (int k = 0; k < EM_IN; ++k) XT[k * L + ge] = x[k];
  float h1[EM_H1];
#pragma unroll
  for (int j = 0; j < EM_H1; ++j) {
    float a = b1s[j];
#pragma unroll
    for (int k = 0; k < EM_IN; ++k) a += w1s[j * EM_IN + k] * x[k];
    h1[j] = a > 0.0f ? a : 0.0f;
    H1T[j * L + ge] = h1[j];
  }
  const float g = grad_logits[ge];
  float dh2[EM_H2];
#pragma unroll
  for (int j = 0; j < EM_H2; ++j) {
    float a = b2s[j];
#pragma unroll
    for (int k = 0; k < EM_H1; ++k) a += w2s[j * EM_H1 + k] * h1[k];
    const float h2 = a > 0.0f ? a : 0.0f;
    H2T[j * L + ge] = h2;
    dh2[j] = a > 0.0f ? g * w3s[j] : 0.0f;
    D2T[j * L + ge] = dh2[j];
  }
#pragma unroll
  for (int k = 0; k < EM_H1; ++k) {
    float a = 0.0f;
#pragma unroll
    for (int j = 0; j < EM_H2; ++j) a += dh2[j] * w2s[j * EM_H1 + k];
    D1T[k * L + ge] = h1[k] > 0.0f ? a : 0.0f;
  }
}

// stage 2: C[p][q] += sum_l A[p][l] * Bm[q][l] for an 8 x 8 block of outputs per workgroup (Bm == NULL: a row of ones,
// i.e. row sums). Fixed order: every thread strides over l, then a fixed LDS tree — deterministic.
#define ND_T 256
__global__ __launch_bounds__(ND_T) void k_nt_dot(const float* __restrict__ Am, int64_t P, const float* __restrict__ Bm,
                                                 int64_t Q, int64_t L, float* __restrict__ Cm, int64_t ldc) {
  __shared__ float red[ND_T];
  const int p0 = blockIdx.x * 8, q0 = blockIdx.y * 8;
  float acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.0f;
  for (int64_t l = threadIdx.x; l < L; l += ND_T) {
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (p0 + i < P) ? Am[(int64_t)(p0 + i) * L + l] : 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (q0 + j < Q) ? (Bm ? Bm[(int64_t)(q0 + j) * L + l] : 1.0f) : 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] += a[i] * b[j];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[threadIdx.x] = acc[i][j];
      __syncthreads();
      for (int s = ND_T / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
      }
      if (threadIdx.x == 0 && p0 + i < P && q0 + j < Q) Cm[(int64_t)(p0 + i) * ldc + q0 + j] += red[0];
      __syncthreads();
    }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
extern "C" int tarl_policy_obs16(const float* node_features, int64_t nf_ld, const int64_t* agent_index,
                                 const float* agent_features, int64_t A, int64_t a_mstride, int64_t M, int64_t N,
                                 float* obs16, tarl_stream stream) {
  TARL_REQUIRE(node_features && agent_index && agent_features && obs16, "null argument");
  TARL_REQUIRE(M >= 1 && N >= 1 && A >= 1 && nf_ld >= 7, "bad sizes");
  hipLaunchKernelGGL(k_obs16_cat, dim3((unsigned)ceil_div(M * N, FB)), dim3(FB), 0, (hipStream_t)stream, node_features,
                     nf_ld, agent_index, agent_features, A, a_mstride, M, N, obs16);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_fused_obs16(const tarl_plan* plan, const tarl_fused* f, const float* x, int64_t B, int64_t x_bstride,
                                int64_t ldx, int32_t Nmax, const float* agent_features, int64_t A, int64_t a_bstride,
                                float* obs16, tarl_stream stream) {
  int rc = tarl_check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(x && agent_features && obs16 && A >= 1, "null argument");
  if (plan->N == 0) return TARL_OK;
  const Layout L{Nmax, ldx, x_bstride};
  hipLaunchKernelGGL(k_obs16_packed, dim3((unsigned)ceil_div(B * plan->N, FB)), dim3(FB), 0, (hipStream_t)stream,
                     plan->out_ptr, plan->out_dst, x, L, B, plan->N, tarl_to_bufs(f), agent_features, A, a_bstride,
                     obs16);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_policy_edge_mlp_fwd(const tarl_plan* plan, const float* obs16, int64_t M, const float* edge_attr,
                                        const float* w1, const float* b1, const float* w2, const float* b2,
                                        const float* w3, const float* b3, int precision, float* logits,
                                        tarl_stream stream) {
  TARL_REQUIRE(plan && obs16 && edge_attr && w1 && b1 && w2 && b2 && w3 && b3 && logits, "null argument");
  TARL_REQUIRE(M >= 1 && M < 65536, "bad batch size");
  TARL_REQUIRE(precision == 0 || precision == 1, "precision: 0 = fp32 MFMA, 1 = bf16 MFMA");
  TARL_REQUIRE(((uintptr_t)obs16) % 16 == 0, "obs16 must be 16-byte aligned");
  if (plan->E == 0) return TARL_OK;
  const EdgeMlpW W{w1, b1, w2, b2, w3, b3};
  const dim3 grid((unsigned)ceil_div(ceil_div(plan->E, EM_TILE), EM_TPW), (unsigned)M);
  if (precision == 0) {
    // the opt-in to > 64 KB of dynamic LDS is a per-device attribute of the function
    static bool lds_set[64] = {false};
    int devid = 0;
    TARL_CHECK_HIP(hipGetDevice(&devid));
    if (!lds_set[devid & 63]) {
      TARL_CHECK_HIP(hipFuncSetAttribute((const void*)k_edge_mlp_fwd_f32, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)EMF_LDS_BYTES));
      lds_set[devid & 63] = true;
    }
    hipLaunchKernelGGL(k_edge_mlp_fwd_f32, grid, dim3(EM_THREADS), EMF_LDS_BYTES, (hipStream_t)stream, plan->src,
                       plan->dst, plan->E, plan->N, obs16, edge_attr, W, logits);
  } else
    hipLaunchKernelGGL(k_edge_mlp_fwd_bf16, grid, dim3(EM_THREADS), 0, (hipStream_t)stream, plan->src, plan->dst, plan->E,
                       plan->N, obs16, edge_attr, W, logits);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int64_t tarl_policy_edge_mlp_bwd_scratch_floats(const tarl_plan* plan, int64_t M) {
  return plan ? (int64_t)(EM_IN + 2 * EM_H1 + 2 * EM_H2) * M * plan->E : -1;
}

extern "C" int tarl_policy_edge_mlp_bwd(const tarl_plan* plan, const float* obs16, int64_t M, const float* edge_attr,
                                        const float* w1, const float* b1, const float* w2, const float* b2,
                                        const float* w3, const float* b3, const float* grad_logits, float* scratch,
                                        float* gw1, float* gb1, float* gw2, float* gb2, float* gw3, float* gb3,
                                        tarl_stream stream) {
  TARL_REQUIRE(plan && obs16 && edge_attr && w1 && b1 && w2 && b2 && w3 && b3 && grad_logits && scratch, "null argument");
  TARL_REQUIRE(gw1 && gb1 && gw2 && gb2 && gw3 && gb3, "null gradient buffer");
  TARL_REQUIRE(M >= 1, "bad batch size");
  if (plan->E == 0) return TARL_OK;
  const int64_t L = M * plan->E;
  float* XT = scratch;
  float* H1T = XT + EM_IN * L;
  float* H2T = H1T + EM_H1 * L;
  float* D1T = H2T + EM_H2 * L;
  float* D2T = D1T + EM_H1 * L;
  const EdgeMlpW W{w1, b1, w2, b2, w3, b3};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_edge_mlp_bwd_edges, dim3((unsigned)ceil_div(L, EM_THREADS)), dim3(EM_THREADS), 0, s, plan->src,
                     plan->dst, plan->E, plan->N, M, obs16, edge_attr, W, grad_logits, XT, H1T, H2T, D1T, D2T);
  TARL_LAUNCH_CHECK();
  auto dot = [&](const float* Am, int64_t P, const float* Bm, int64_t Q, float* Cm, int64_t ldc) {
    hipLaunchKernelGGL(k_nt_dot, dim3((unsigned)ceil_div(P, 8), (unsigned)ceil_div(Q, 8)), dim3(ND_T), 0, s, Am, P, Bm, Q,
                       L, Cm, ldc);
  };
  dot(D1T, EM_H1, XT, EM_IN, gw1, EM_IN);          // dW1 = dh1^T x
  dot(D1T, EM_H1, nullptr, 1, gb1, 1);             // db1
  dot(D2T, EM_H2, H1T, EM_H1, gw2, EM_H1);         // dW2 = dh2^T h1
  dot(D2T, EM_H2, nullptr, 1, gb2, 1);             // db2
  dot(H2T, EM_H2, grad_logits, 1, gw3, 1);         // dW3 = h2^T g
  dot(grad_logits, 1, nullptr, 1, gb3, 1);         // db3
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}
