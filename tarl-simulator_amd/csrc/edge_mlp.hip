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

// One agent row (9 floats, 36 bytes at a 4-byte-aligned address) as two 16-byte loads and one dword instead of nine dwords:
// every lane gathers from a line of its own here, so the texture-address unit pays per INSTRUCTION (64 lines each), not
// per byte — nine scattered dword gathers per (road, environment) pair were the observation kernels' bound (round 5).
struct __attribute__((packed, aligned(4))) AgF4 {
  float x, y, z, w;
};
__device__ __forceinline__ void load_agent_row(const float* __restrict__ arow, float (&a)[9]) {
  const AgF4 v0 = *reinterpret_cast<const AgF4*>(arow);
  const AgF4 v1 = *reinterpret_cast<const AgF4*>(arow + 4);
  a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
  a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
  a[8] = arow[8];
}

// ---- obs16: x = cat(node_features, agent_features[agent_index]) from the packed state of the fused engine ----------------
// node_features = x[:, 3*Nmax:] = {MAX, NUMBER_OF_AGENT, FREE_FLOW, LENGTH, MAX_FLOW, SELECTED_ROAD, ROAD_INDEX}
// (TransportationSimulator.state, src/transportation_simulator.py:360-366); agent_index = the head-of-FIFO id.
// A workgroup owns 8 nodes x 64 environments: the packed words are read env-minor (lanes along the environment:
// coalesced), the 64-byte rows are turned through LDS, and every environment's 8 rows leave as one 512-byte run.
#define OB_TI 8
#define OB_TB 64
#define OB_LD (OB_TI * 16 + 4)     // floats per environment in LDS (+4: the float4 stores of 8 lanes cover all banks)
__global__ __launch_bounds__(256) void k_obs16_packed(const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_dst,
                                                      const float* __restrict__ x0, Layout L, int64_t B, int64_t N,
                                                      FusedBufs fb, const float* __restrict__ ag, int64_t A,
                                                      int64_t a_bstride, float* __restrict__ obs) {
  __shared__ __attribute__((aligned(16))) float sm[OB_TB * OB_LD];
  const int tid = threadIdx.x, bl = tid & 63;
  const int64_t b = (int64_t)blockIdx.x * OB_TB + bl, i0 = (int64_t)blockIdx.y * OB_TI;
#pragma unroll
  for (int r = 0; r < OB_TI / 4; ++r) {
    const int il = (tid >> 6) + 4 * r;
    const int64_t i = i0 + il;
    if (b < B && i < N) {
      const int64_t gid = i * B + b;
      const uint32_t hd = fb.hdp[gid].x;
      const float4 st = fb.st0[i];
      const float* xs = x0 + i * L.ldx;     // static columns: environment 0 speaks for all
      const long long head = (long long)(hd >> 8);
      const float* arow = ag + b * a_bstride + ((head >= 0 && head < A) ? head : 0) * AG_COLS;
      float ar[9];
      load_agent_row(arow, ar);
      float4* o = reinterpret_cast<float4*>(sm + bl * OB_LD + il * 16);
      o[0] = make_float4(st.x, (float)(hd & HD_CNT), st.y, xs[L.col_maxn() + 3]);
      o[1] = make_float4(xs[L.col_maxflow()], sel_value(fb, out_ptr, out_dst, i, gid), st.z, ar[0]);
      o[2] = make_float4(ar[1], ar[2], ar[3], ar[4]);
      o[3] = make_float4(ar[5], ar[6], ar[7], ar[8]);
    }
  }
  __syncthreads();
  const int part = tid & 31;                 // float4 number inside the environment's 8 rows
  const int64_t i = i0 + (part >> 2);
#pragma unroll
  for (int r = 0; r < OB_TB / 8; ++r) {
    const int be = (tid >> 5) + 8 * r;
    const int64_t bb = (int64_t)blockIdx.x * OB_TB + be;
    if (bb < B && i < N)
      reinterpret_cast<float4*>(obs + (bb * N + i0) * 16)[part] = reinterpret_cast<const float4*>(sm + be * OB_LD)[part];
  }
}

// the same observation rounded to bf16 (RNE) — what the bf16 MLP reads ("bf16 MPNN features"): 32-byte rows, half the
// bytes written here and gathered there. Same tile plan (8 nodes x 64 environments, 256-byte runs per environment).
#define OBB_LD (OB_TI * 8 + 4)     // uint32 per environment in LDS
__global__ __launch_bounds__(256) void k_obs16_packed_bf16(const int32_t* __restrict__ out_ptr,
                                                           const int32_t* __restrict__ out_dst,
                                                           const float* __restrict__ x0, Layout L, int64_t B, int64_t N,
                                                           FusedBufs fb, const float* __restrict__ ag, int64_t A,
                                                           int64_t a_bstride, uint16_t* __restrict__ obs) {
  __shared__ __attribute__((aligned(16))) uint32_t sm[OB_TB * OBB_LD];
  const int tid = threadIdx.x, bl = tid & 63;
  const int64_t b = (int64_t)blockIdx.x * OB_TB + bl, i0 = (int64_t)blockIdx.y * OB_TI;
#pragma unroll
  for (int r = 0; r < OB_TI / 4; ++r) {
    const int il = (tid >> 6) + 4 * r;
    const int64_t i = i0 + il;
    if (b < B && i < N) {
      const int64_t gid = i * B + b;
      const uint32_t hd = fb.hdp[gid].x;
      const float4 st = fb.st0[i];
      const float* xs = x0 + i * L.ldx;
      const long long head = (long long)(hd >> 8);
      const float* arow = ag + b * a_bstride + ((head >= 0 && head < A) ? head : 0) * AG_COLS;
      float ar[9];
      load_agent_row(arow, ar);
      const float v[16] = {st.x, (float)(hd & HD_CNT), st.y, xs[L.col_maxn() + 3], xs[L.col_maxflow()],
                           sel_value(fb, out_ptr, out_dst, i, gid), st.z, ar[0], ar[1], ar[2], ar[3], ar[4],
                           ar[5], ar[6], ar[7], ar[8]};
      uint4* o = reinterpret_cast<uint4*>(sm + bl * OBB_LD + il * 8);
      uint32_t w[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        w[q] = (uint32_t)f32_to_bf16_rne(v[2 * q]) | ((uint32_t)f32_to_bf16_rne(v[2 * q + 1]) << 16);
      o[0] = make_uint4(w[0], w[1], w[2], w[3]);
      o[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
  __syncthreads();
  const int part = tid & 15;                 // uint4 number inside the environment's 8 rows (2 per row)
  const int64_t i = i0 + (part >> 1);
#pragma unroll
  for (int r = 0; r < OB_TB / 16; ++r) {
    const int be = (tid >> 4) + 16 * r;
    const int64_t bb = (int64_t)blockIdx.x * OB_TB + be;
    if (bb < B && i < N)
      reinterpret_cast<uint4*>(obs + (bb * N + i0) * 16)[part] = reinterpret_cast<const uint4*>(sm + be * OBB_LD)[part];
  }
}

// fp32 observation rows of a FEW environments (the frames an optimiser minibatch keeps): row (env[j], i) -> keep[slot[j]][i]
__global__ __launch_bounds__(FB) void k_obs16_rows(const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_dst,
                                                   const float* __restrict__ x0, Layout L, int64_t B, int64_t N,
                                                   FusedBufs fb, const float* __restrict__ ag, int64_t A,
                                                   int64_t a_bstride, const int32_t* __restrict__ env,
                                                   const int32_t* __restrict__ slot, float* __restrict__ keep) {
  const int64_t i = (int64_t)blockIdx.x * FB + threadIdx.x;
  if (i >= N) return;
  const int64_t b = env[blockIdx.y];
  const int64_t gid = i * B + b;
  const uint32_t hd = fb.hdp[gid].x;
  const float4 st = fb.st0[i];
  const float* xs = x0 + i * L.ldx;
  const long long head = (long long)(hd >> 8);
  const float* arow = ag + b * a_bstride + ((head >= 0 && head < A) ? head : 0) * AG_COLS;
  float ar[9];
  load_agent_row(arow, ar);
  float4* o = reinterpret_cast<float4*>(keep + ((int64_t)slot[blockIdx.y] * N + i) * 16);
  o[0] = make_float4(st.x, (float)(hd & HD_CNT), st.y, xs[L.col_maxn() + 3]);
  o[1] = make_float4(xs[L.col_maxflow()], sel_value(fb, out_ptr, out_dst, i, gid), st.z, ar[0]);
  o[2] = make_float4(ar[1], ar[2], ar[3], ar[4]);
  o[3] = make_float4(ar[5], ar[6], ar[7], ar[8]);
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
// Register-resident, "transposed" (D^T = W X^T): per 32 edges a wave computes H1^T [64][32] = W1 [64][33+] X^T,
// H2^T [32][32] = W2 H1^T, logit = w3 . H2^T with the WEIGHTS as the A operands (held in registers for the wave's whole
// life) and the EDGES as the columns. In the 32x32 accumulator layout lane l then owns, for ITS edge (column l & 31), the
// hidden units (r & 3) + 8 (r >> 2) + 4 (l >> 5) — sixteen values of one edge per 32-row tile — and the B operand of the
// next layer wants, per lane, consecutive k of its own column: a dot product does not care in which order k runs, so
// the second layer simply walks the hidden units in the order the lanes already hold them (W2's columns are permuted
// to match when its fragments are built). Activations never touch LDS: gather (each half-wave one 32-byte half of the two
// 64-byte node rows) -> registers -> MFMAs -> 16 fma + one cross-half add -> logit. Waves are persistent: each builds
// its weight fragments once and walks a contiguous range of 32-edge chunks, prefetching the next chunk's rows.
#define EMR_WAVES 4
__device__ __forceinline__ int emr_unit(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct EdgeIn {
  float4 s0, s1, d0, d1;
  float ea;
};

__device__ __forceinline__ EdgeIn emr_load(const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                           const float* __restrict__ edge_attr, const float* __restrict__ obs,
                                           int32_t E, int64_t N, int32_t m, int32_t c, int lane) {
  int32_t e = c * 32 + (lane & 31);
  e = e < E ? e : E - 1;                     // the tail chunk's spare lanes repeat the last edge (never stored)
  const float* om = obs + (int64_t)m * N * 16 + 8 * (lane >> 5);
  const float4* ps = reinterpret_cast<const float4*>(om + (int64_t)src[e] * 16);
  const float4* pd = reinterpret_cast<const float4*>(om + (int64_t)dst[e] * 16);
  EdgeIn in;
  in.s0 = ps[0];
  in.s1 = ps[1];
  in.d0 = pd[0];
  in.d1 = pd[1];
  in.ea = edge_attr[e];
  return in;
}

// max(x, 0) as ONE integer instruction: a float is negative exactly when its bit pattern is a negative int32 (fmaxf
// costs two: the IEEE mode quiets signalling NaNs first). -0 and negative NaNs become +0, positive NaNs pass through.
__device__ __forceinline__ float emr_relu(float x) {
  const int32_t i = __float_as_int(x);
  return __int_as_float(i > 0 ? i : 0);
}

// chunk g = m * CH + c (host: M * CH < 2^31); a wave walks [g0, g1) keeping (m, c) by increments
struct ChunkWalk {
  uint32_t g, g1, CH;
  int32_t m, c, mn, cn;
  __device__ __forceinline__ bool init(int64_t E, int64_t M, int wave) {
    CH = (uint32_t)((E + 31) >> 5);
    const uint32_t total = (uint32_t)M * CH, nw = gridDim.x * EMR_WAVES, gw = blockIdx.x * EMR_WAVES + wave;
    const uint32_t per = (total + nw - 1) / nw;
    g = gw * per;
    g1 = (g + per < total) ? g + per : total;
    if (g >= g1) return false;
    mn = (int32_t)(g / CH);
    cn = (int32_t)(g - (uint32_t)mn * CH);
    return true;
  }
  __device__ __forceinline__ void step() {      // (m, c) <- the chunk just prefetched; (mn, cn) <- its successor
    m = mn;
    c = cn;
    if (++cn == (int32_t)CH) {
      cn = 0;
      ++mn;
    }
  }
};

// relu on eight bf16 at once: rounding keeps the sign, and a bf16 is negative exactly when its bit pattern is a negative
// int16 — v_pk_max_i16 does two activations per instruction
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 emr_relu8(const bf16x8 v) {
  s16x8 i = __builtin_bit_cast(s16x8, v);
  i = __builtin_elementwise_max(i, (s16x8){0, 0, 0, 0, 0, 0, 0, 0});
  return __builtin_bit_cast(bf16x8, i);
}
// two floats -> two bf16 (RNE) in one v_cvt_pk_bf16_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t emr_cvt2(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ bf16x8 emr_pack(const float4 a, const float4 b) {
  const u32x4 p = {emr_cvt2(a.x, a.y), emr_cvt2(a.z, a.w), emr_cvt2(b.x, b.y), emr_cvt2(b.z, b.w)};
  return __builtin_bit_cast(bf16x8, p);
}

// ---- the forward kernels' walk over their chunks: a two-stage software pipeline of the gathers -----------------------------
// The gathers are two dependent rounds (edge -> node ids -> node rows) of ~1-2 us each against ~0.2 us (bf16) to ~1 us
// (fp32 MFMA) of arithmetic per chunk. A wave therefore keeps P chunks of rows and P chunks of node ids in flight: a step
// consumes the rows of chunk c and the ids of chunk c + P (both requested P steps ago), then requests the ids of chunk
// c + 2P and — with the ids it just consumed — the rows of chunk c + P into the registers it has just freed. The buffers
// are indexed by step mod P with the loop unrolled P times (a shifting queue would have to MOVE registers whose loads are
// still in flight, i.e. wait for them), and the ids go out before the rows because the memory counter retires in order.
// P = 2 with bf16 observations (9 registers per chunk in flight), 1 with fp32 observations (17). Until round 5 the fp32
// and x3 kernels fetched ids and rows of the NEXT chunk in one go — every chunk then waited a full memory round trip for
// the ids before it could ask for the rows (a quarter of their wave-time parked at that wait, profiles/r05_mlp_counters.txt).
// A wave's range of chunks is cut into per-sample segments (round 5): inside one the chunk walkers are plain counters
// that saturate at the segment's last chunk (66 -> ~35 scalar instructions per chunk against three (sample, chunk) walkers
// with carries; bf16: -4 %); past the end a wave re-requests that last chunk (no divergent control flow inside the loop,
// which would make the compiler's wait-count tracking give up and wait for everything).
struct EIdx {
  int32_t s, d;
  float ea;
};
__device__ __forceinline__ EIdx emr_ldidx(const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                          const float* __restrict__ edge_attr, int32_t E, int32_t c, int lane) {
  c = __builtin_amdgcn_readfirstlane(c);     // wave-uniform by construction: scalar base + 32-bit lane offset
  int32_t e = c * 32 + (lane & 31);
  e = e < E ? e : E - 1;                     // the tail chunk's spare lanes repeat the last edge (never stored)
  EIdx i;
  i.s = src[e];
  i.d = dst[e];
  i.ea = edge_attr[e];
  return i;
}
template <bool OBS_BF16>
struct ERows;
template <>
struct ERows<false> {      // fp32 observations [M][N][16]: this half-wave's 8 floats of both rows
  float4 s0, s1, d0, d1;
  float ea;
  __device__ __forceinline__ void load(const void* __restrict__ obs, int64_t N, int32_t m, const EIdx& i, int lane) {
    const float* om = (const float*)obs + (int64_t)m * N * 16 + 8 * (lane >> 5);
    const float4* ps = reinterpret_cast<const float4*>(om + (int64_t)i.s * 16);
    const float4* pd = reinterpret_cast<const float4*>(om + (int64_t)i.d * 16);
    s0 = ps[0];
    s1 = ps[1];
    d0 = pd[0];
    d1 = pd[1];
    ea = i.ea;
  }
  __device__ __forceinline__ bf16x8 xs() const { return emr_pack(s0, s1); }
  __device__ __forceinline__ bf16x8 xd() const { return emr_pack(d0, d1); }
};
template <>
struct ERows<true> {       // bf16 observations [M][N][16] (tarl_fused_obs16_bf16): 16 bytes of both rows
  uint4 s, d;
  float ea;
  __device__ __forceinline__ void load(const void* __restrict__ obs, int64_t N, int32_t m, const EIdx& i, int lane) {
    // the sample is wave-uniform (scalar base); node rows are 32 bytes: a 32-bit byte offset per lane (host: N < 2^26)
    const char* om = (const char*)obs + (int64_t)__builtin_amdgcn_readfirstlane(m) * N * 32;
    const uint32_t hb = 16u * (uint32_t)(lane >> 5);
    s = *reinterpret_cast<const uint4*>(om + ((uint32_t)i.s * 32u + hb));
    d = *reinterpret_cast<const uint4*>(om + ((uint32_t)i.d * 32u + hb));
    ea = i.ea;
  }
  __device__ __forceinline__ bf16x8 xs() const { return __builtin_bit_cast(bf16x8, s); }
  __device__ __forceinline__ bf16x8 xd() const { return __builtin_bit_cast(bf16x8, d); }
};


// chunk g = m * CH + c (host: M * CH < 2^31): wave gw of nw walks [gw * per, (gw + 1) * per), sample by sample;
// body(rows, m, c, live) computes chunk c of sample m from its gathered rows (live = false: a pipeline step past the end
// of a segment whose length is not a multiple of P — its rows are the last chunk's once more, nothing may be stored)
template <int P, bool OBS_BF16, class Body>
__device__ __forceinline__ void emr_walk(const int32_t* __restrict__ src, const int32_t* __restrict__ dst,
                                         const float* __restrict__ edge_attr, const void* __restrict__ obs, int64_t E,
                                         int64_t N, int64_t M, int wave, int lane, Body&& body) {
  const int32_t CH = (int32_t)((E + 31) >> 5), Ei = (int32_t)E;
  const uint32_t total = (uint32_t)M * (uint32_t)CH, nw = gridDim.x * EMR_WAVES, gw = blockIdx.x * EMR_WAVES + wave;
  const uint32_t per = (total + nw - 1) / nw;
  uint32_t g = gw * per;
  const uint32_t g1 = (g + per < total) ? g + per : total;
  if (g >= g1) return;
  int32_t m = (int32_t)(g / (uint32_t)CH), c0 = (int32_t)(g - (uint32_t)m * (uint32_t)CH);
  ERows<OBS_BF16> R[P];
  EIdx I[P];
  while (g < g1) {
    const int32_t left = (int32_t)(g1 - g), cend = (CH - c0 < left) ? CH : c0 + left, last = cend - 1;
    int32_t c_cur = c0, c_idx = c0;
#pragma unroll
    for (int K = 0; K < P; ++K) {                  // prologue: rows of the segment's first P chunks, ids of the next P
      R[K].load(obs, N, m, emr_ldidx(src, dst, edge_attr, Ei, c_idx, lane), lane);
      c_idx = c_idx < last ? c_idx + 1 : last;
    }
#pragma unroll
    for (int K = 0; K < P; ++K) {
      I[K] = emr_ldidx(src, dst, edge_attr, Ei, c_idx, lane);
      c_idx = c_idx < last ? c_idx + 1 : last;
    }
    for (int32_t c = c0; c < cend; c += P) {
#pragma unroll
      for (int K = 0; K < P; ++K) {
        const ERows<OBS_BF16> cur = R[K];
        const EIdx use = I[K];
        const int32_t cc = c_cur;
        I[K] = emr_ldidx(src, dst, edge_attr, Ei, c_idx, lane);
        R[K].load(obs, N, m, use, lane);
        c_cur = c_cur < last ? c_cur + 1 : last;
        c_idx = c_idx < last ? c_idx + 1 : last;
        body(cur, m, cc, c + K < cend);
      }
    }
    g += (uint32_t)(cend - c0);
    c0 = 0;
    ++m;
  }
}

// fp32: v_mfma_f32_32x32x2_f32, exact fp32 products. k runs [x_src 8h..8h+7] [x_dst 8h..8h+7] for half-wave h (8 + 8
// k-steps of 2), then one k-step {edge_attr, 1} against {W1[:, 32], b1}: the bias enters the accumulation as an exact
// product. 34 + 32 MFMAs per 32 edges: the matrix cores are the bound (66 x 16 passes).
__global__ __launch_bounds__(EMR_WAVES * 64) void k_edge_mlp_fwd_f32(const int32_t* __restrict__ src,
                                                                     const int32_t* __restrict__ dst, int64_t E,
                                                                     int64_t N, int64_t M,
                                                                     const float* __restrict__ obs,
                                                                     const float* __restrict__ edge_attr, EdgeMlpW W,
                                                                     float* __restrict__ logits) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the chunk walker lives in SGPRs
  const int h = lane >> 5, j = lane & 31;
  float w1a[2][17], w2a[32];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int u = 32 * a + j;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      w1a[a][t] = W.w1[u * EM_IN + 8 * h + t];
      w1a[a][8 + t] = W.w1[u * EM_IN + 16 + 8 * h + t];
    }
    w1a[a][16] = h == 0 ? W.w1[u * EM_IN + 32] : W.b1[u];
  }
#pragma unroll
  for (int t = 0; t < 32; ++t) w2a[t] = W.w2[j * EM_H1 + 32 * (t >> 4) + emr_unit(t & 15, h)];
  f32x16 b2r;
  float w3r[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    b2r[r] = W.b2[emr_unit(r, h)];
    w3r[r] = W.w3[emr_unit(r, h)];
  }
  const float b3 = W.b3[0];
  const f32x16 zero = {0};
  emr_walk<1, false>(src, dst, edge_attr, obs, E, N, M, wave, lane, [&](const ERows<false>& cur, int32_t m, int32_t c, bool live) {
    const float xin[17] = {cur.s0.x, cur.s0.y, cur.s0.z, cur.s0.w, cur.s1.x, cur.s1.y, cur.s1.z, cur.s1.w,
                           cur.d0.x, cur.d0.y, cur.d0.z, cur.d0.w, cur.d1.x, cur.d1.y, cur.d1.z, cur.d1.w,
                           h == 0 ? cur.ea : 1.0f};
    f32x16 acc0 = zero, acc1 = zero;
#pragma unroll
    for (int t = 0; t < 17; ++t) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1a[0][t], xin[t], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1a[1][t], xin[t], acc1, 0, 0, 0);
    }
    f32x16 c0 = b2r;
#pragma unroll
    for (int t = 0; t < 16; ++t) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2a[t], emr_relu(acc0[t]), c0, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 16; ++t) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2a[16 + t], emr_relu(acc1[t]), c0, 0, 0, 0);
    float part = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part = fmaf(emr_relu(c0[r]), w3r[r], part);
    const float tot = part + __shfl_xor(part, 32);
    const int32_t e = c * 32 + j;
    if (live && h == 0 && e < (int32_t)E) logits[(int64_t)m * E + e] = tot + b3;
  });
}

// ---- bf16: v_mfma_f32_32x32x16_bf16 ----------------------------------------------------------------------------------------
// layer 1, k-steps of 16: [x_src 0..15] [x_dst 0..15] [edge_attr, 1, 1, 1, 0 ...]: the three ones meet the bias b1 split
// into three bf16 pieces (hi + mid + lo == b1 exactly), i.e. the fp32 bias enters the fp32 accumulation unrounded.
// Inputs, weights and the first hidden activation are rounded to bf16 (RNE); fp32 accumulation, fp32 second activation
// and output: BASELINE config 5's "bf16 MPNN features" (tolerance stated in tests/test_gpu_edge_mlp.py). 6 + 4 MFMAs per
// 32 edges. What bounds it (profiles/r05_mlp_counters.txt, round 5): neither the matrix pipe (busy 0.46 of the launch) nor
// the vector ALU nor their sum nor the gather — a wave's in-order stream of ~190 instructions per chunk (a third of them
// scalar) at four waves per SIMD; five or six waves with the weights left in LDS were slower.
template <bool OBS_BF16>
__global__ __launch_bounds__(EMR_WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_edge_mlp_fwd_bf16(const int32_t* __restrict__ src,
                                                                      const int32_t* __restrict__ dst, int64_t E,
                                                                      int64_t N, int64_t M,
                                                                      const void* __restrict__ obs,
                                                                      const float* __restrict__ edge_attr, EdgeMlpW W,
                                                                      float* __restrict__ logits) {
  // chunks of gathers in flight per wave. Two (round 3; three in round 2): 126 VGPRs instead of 146, i.e. four waves per
  // SIMD instead of three — a fourth wave overlaps more vector-ALU work with the matrix pipe than a third chunk of loads
  // hides latency (205.5 -> 198.7 us per 20.5 M edges)
  constexpr int P = OBS_BF16 ? 2 : 1;
  // fragment-ordered weights: W1f [2 tiles][3 k-steps][64 lanes][8], W2f [4 k-steps][64 lanes][8]
  __shared__ __attribute__((aligned(16))) uint16_t W1f[6 * 64 * 8];
  __shared__ __attribute__((aligned(16))) uint16_t W2f[4 * 64 * 8];
  // b2 / w3 in accumulator-row order per half-wave: re-read from LDS every chunk (two addresses per wave: broadcast) so
  // that 32 registers go to loads in flight instead
  __shared__ __attribute__((aligned(16))) float B2L[2 * 16];
  __shared__ __attribute__((aligned(16))) float W3L[2 * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the chunk walkers below live in SGPRs
  for (int idx = tid; idx < 6 * 64 * 8; idx += EMR_WAVES * 64) {
    const int q = idx & 7, l = (idx >> 3) & 63, f = idx >> 9;
    const int a = f / 3, ks = f - 3 * a, u = 32 * a + (l & 31), h = l >> 5;
    uint16_t v = 0;
    if (ks < 2)
      v = f32_to_bf16_rne(W.w1[u * EM_IN + 16 * ks + 8 * h + q]);
    else if (h == 0) {
      if (q == 0)
        v = f32_to_bf16_rne(W.w1[u * EM_IN + 32]);
      else if (q < 4) {      // b1 = hi + mid + lo, each piece a bf16
        const float b = W.b1[u];
        const uint16_t hi = f32_to_bf16_rne(b);
        const float r1 = b - __uint_as_float((uint32_t)hi << 16);
        const uint16_t mid = f32_to_bf16_rne(r1);
        const float r2 = r1 - __uint_as_float((uint32_t)mid << 16);
        v = q == 1 ? hi : q == 2 ? mid : f32_to_bf16_rne(r2);
      }
    }
    W1f[idx] = v;
  }
  for (int idx = tid; idx < 4 * 64 * 8; idx += EMR_WAVES * 64) {
    const int q = idx & 7, l = (idx >> 3) & 63, ks = idx >> 9;
    const int u = 32 * (ks >> 1) + emr_unit(8 * (ks & 1) + q, l >> 5);
    W2f[idx] = f32_to_bf16_rne(W.w2[(l & 31) * EM_H1 + u]);
  }
  if (tid < 32) {
    B2L[tid] = W.b2[emr_unit(tid & 15, tid >> 4)];
    W3L[tid] = W.w3[emr_unit(tid & 15, tid >> 4)];
  }
  __syncthreads();
  bf16x8 w1f[6], w2f[4];
#pragma unroll
  for (int f = 0; f < 6; ++f) w1f[f] = *reinterpret_cast<const bf16x8*>(W1f + (f * 64 + lane) * 8);
#pragma unroll
  for (int f = 0; f < 4; ++f) w2f[f] = *reinterpret_cast<const bf16x8*>(W2f + (f * 64 + lane) * 8);
  const int h = lane >> 5;
  const float4* b2l = reinterpret_cast<const float4*>(B2L + 16 * h);
  const float4* w3l = reinterpret_cast<const float4*>(W3L + 16 * h);
  const float b3 = W.b3[0];
  const f32x16 zero = {0};

  emr_walk<P, OBS_BF16>(src, dst, edge_attr, obs, E, N, M, wave, lane, [&](const ERows<OBS_BF16>& cur, int32_t m, int32_t c, bool live) {
    const bf16x8 xs = cur.xs(), xd = cur.xd();
    bf16x8 xe = {0};
    if (h == 0) {
      xe[0] = (__bf16)cur.ea;
      xe[1] = (__bf16)1.0f;
      xe[2] = (__bf16)1.0f;
      xe[3] = (__bf16)1.0f;
    }
    f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[0], xs, zero, 0, 0, 0);
    f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[3], xs, zero, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[1], xd, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[4], xd, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[2], xe, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[5], xe, acc1, 0, 0, 0);
    bf16x8 hb[4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      u32x4 p0, p1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        p0[q] = emr_cvt2(acc0[8 * s2 + 2 * q], acc0[8 * s2 + 2 * q + 1]);
        p1[q] = emr_cvt2(acc1[8 * s2 + 2 * q], acc1[8 * s2 + 2 * q + 1]);
      }
      hb[s2] = emr_relu8(__builtin_bit_cast(bf16x8, p0));
      hb[2 + s2] = emr_relu8(__builtin_bit_cast(bf16x8, p1));
    }
    f32x16 c0;
    asm volatile("" ::: "memory");      // keeps the two LDS tables out of loop-invariant registers
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = b2l[q];
      c0[4 * q] = v.x;
      c0[4 * q + 1] = v.y;
      c0[4 * q + 2] = v.z;
      c0[4 * q + 3] = v.w;
    }
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[0], hb[0], c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[1], hb[1], c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[2], hb[2], c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f[3], hb[3], c0, 0, 0, 0);
    float part = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = w3l[q];
      part = fmaf(emr_relu(c0[4 * q]), v.x, part);
      part = fmaf(emr_relu(c0[4 * q + 1]), v.y, part);
      part = fmaf(emr_relu(c0[4 * q + 2]), v.z, part);
      part = fmaf(emr_relu(c0[4 * q + 3]), v.w, part);
    }
    const float tot = part + __shfl_xor(part, 32);
    const int32_t e = __builtin_amdgcn_readfirstlane(c) * 32 + (lane & 31);
    float* lrow = logits + (int64_t)__builtin_amdgcn_readfirstlane(m) * E;
    if (live && h == 0 && e < (int32_t)E) lrow[e] = tot + b3;
  });
}

// ---- fp32 accuracy on the bf16 pipe: v_mfma_f32_32x32x16_bf16 on operands split into exact bf16 pieces -------------------
// An fp32 number is the exact sum of three bf16 numbers (8 + 8 + 8 significand bits: hi = bf16(v), mid = bf16(v - hi),
// lo = bf16(v - hi - mid), every subtraction exact), and the product of two bf16 numbers is exact in fp32. Inputs, weights
// AND the first hidden activation are split; of the nine piece products of x * w the six of order >= 2^-16 are
// accumulated (hi hi, hi mid, mid hi, hi lo, mid mid, lo hi) and three dropped: mid lo, lo mid and lo lo. With pieces
// taken by TRUNCATION |mid| < 2^-7 |v| and |lo| < 2^-14 |v| (not 2^-8 / 2^-16 as for round-to-nearest pieces), so a dropped
// product is up to 2^-21 |w| |x| — a few times the 2^-24 rounding of one fp32 product, NOT below it: per multiply-add the
// kernel carries up to ~2^-20 |w| |x| of truncation error (all of one sign per operand pair; it does not average out
// like rounding), i.e. over the 33 + 64 terms of the head <= 1e-4 of sum |w| |x|. The contract — 1e-4 of the logits' scale
// — therefore holds with a margin of ~100 on benign inputs and ~1 in the worst case of cancellation (large |w| |x| terms
// of mixed sign summing to a small logit): tests/test_gpu_edge_mlp.py holds the kernel to 1e-4 of sum |w| |x| on exactly
// such inputs (clock-time columns ~2e4 with mixed-sign weights) against an fp64 evaluation, at 6 bf16 MFMAs of 8 passes per 16 k where the fp32
// MFMA needs 8 of 16 passes: 2.7x the matrix rate. Layout exactly as k_edge_mlp_fwd_f32 / _bf16: weights as A operands
// in registers for the wave's life, edges as columns, layer 2 walking the hidden units in accumulator order.
//   layer 1, k-steps of 16: [x_src 0..15] [x_dst 0..15], 6 MFMAs each per 32-row tile, then ONE k-step per tile for
//   edge_attr and the bias: B = {ea_h, ea_h, ea_h, ea_m, ea_m, ea_l, 1, 1 | 1, 0 ...} against
//   A = {w_h, w_m, w_l, w_h, w_m, w_h, b_h, b_m | b_l, 0 ...} (w = W1[:, 32], b = b1, both split): 26 MFMAs;
//   layer 2: 4 k-steps x 6 = 24 MFMAs, b2 as the initial fp32 accumulator; layer 3 in fp32 on the vector ALU.
struct Split3 {
  uint32_t h, m, l;      // two bf16 each (low half = first value)
};
// Pieces by TRUNCATION (the top 16 bits of the fp32 word are a bf16): hi = v & 0xFFFF0000, mid = (v - hi) & 0xFFFF0000,
// lo = v - hi - mid — 8 + 8 + 8 significand bits, every subtraction exact, hi + mid + lo == v exactly, and nothing but
// full-rate integer / add instructions (v_and, v_sub, v_perm): with v_cvt_pk_bf16_f32 (round to nearest even) for the
// three pieces: 946 -> 866 us per 20.5 M edges.
__device__ __forceinline__ uint32_t emr_pack_hi16(float a, float b) {      // {bf16 bits of a, bf16 bits of b}, truncated
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ Split3 emr_split2(float a, float b) {
  Split3 r;
  r.h = emr_pack_hi16(a, b);
  const float ra = a - __uint_as_float(__float_as_uint(a) & 0xFFFF0000u), rb = b - __uint_as_float(__float_as_uint(b) & 0xFFFF0000u);
  r.m = emr_pack_hi16(ra, rb);
  const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xFFFF0000u), sb = rb - __uint_as_float(__float_as_uint(rb) & 0xFFFF0000u);
  r.l = emr_pack_hi16(sa, sb);
  return r;
}
__device__ __forceinline__ void emr_split_host3(float v, uint16_t* h, uint16_t* m, uint16_t* l) {
  *h = f32_to_bf16_rne(v);
  const float r1 = v - __uint_as_float((uint32_t)*h << 16);
  *m = f32_to_bf16_rne(r1);
  *l = f32_to_bf16_rne(r1 - __uint_as_float((uint32_t)*m << 16));
}
struct Frag3 {
  bf16x8 h, m, l;
};
// eight floats -> their three piece fragments
__device__ __forceinline__ Frag3 emr_split8(const float (&v)[8]) {
  u32x4 ph, pm, pl;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const Split3 sp = emr_split2(v[2 * q], v[2 * q + 1]);
    ph[q] = sp.h;
    pm[q] = sp.m;
    pl[q] = sp.l;
  }
  return Frag3{__builtin_bit_cast(bf16x8, ph), __builtin_bit_cast(bf16x8, pm), __builtin_bit_cast(bf16x8, pl)};
}
// acc += W x with both operands in pieces: the six products of order >= 2^-16, small ones first
__device__ __forceinline__ f32x16 emr_mma6(const Frag3& w, const Frag3& x, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.l, x.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, x.l, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.m, x.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.m, x.h, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, x.m, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.h, x.h, acc, 0, 0, 0);
  return acc;
}

#define EMX_W1 (2 * 2 * 3)     // W1 fragments: [tile][k-step][piece]
#define EMX_W2 (4 * 3)         // W2 fragments: [k-step][piece]
__global__ __launch_bounds__(EMR_WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_edge_mlp_fwd_x3(const int32_t* __restrict__ src,
                                                                    const int32_t* __restrict__ dst, int64_t E,
                                                                    int64_t N, int64_t M,
                                                                    const float* __restrict__ obs,
                                                                    const float* __restrict__ edge_attr, EdgeMlpW W,
                                                                    float* __restrict__ logits) {
  // fragment-ordered weight pieces, built once per workgroup in LDS and copied to registers
  __shared__ __attribute__((aligned(16))) uint16_t W1f[EMX_W1 * 64 * 8];
  __shared__ __attribute__((aligned(16))) uint16_t WEf[2 * 64 * 8];          // the edge_attr / bias k-step, per tile
  __shared__ __attribute__((aligned(16))) uint16_t W2f[EMX_W2 * 64 * 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, j = lane & 31;
  for (int idx = tid; idx < 2 * 2 * 64 * 8; idx += EMR_WAVES * 64) {
    const int q = idx & 7, l = (idx >> 3) & 63, f = idx >> 9;           // f = tile * 2 + k-step
    const int a = f >> 1, ks = f & 1, u = 32 * a + (l & 31), hh = l >> 5;
    uint16_t ph, pm, pl;
    emr_split_host3(W.w1[u * EM_IN + 16 * ks + 8 * hh + q], &ph, &pm, &pl);
    W1f[((f * 3 + 0) * 64 + l) * 8 + q] = ph;
    W1f[((f * 3 + 1) * 64 + l) * 8 + q] = pm;
    W1f[((f * 3 + 2) * 64 + l) * 8 + q] = pl;
  }
  for (int idx = tid; idx < 2 * 64 * 8; idx += EMR_WAVES * 64) {
    const int q = idx & 7, l = (idx >> 3) & 63, a = idx >> 9;
    const int u = 32 * a + (l & 31), hh = l >> 5;
    uint16_t wh, wm, wl, bh, bm, bl;
    emr_split_host3(W.w1[u * EM_IN + 32], &wh, &wm, &wl);
    emr_split_host3(W.b1[u], &bh, &bm, &bl);
    const uint16_t lo8[8] = {wh, wm, wl, wh, wm, wh, bh, bm};
    WEf[idx] = hh == 0 ? lo8[q] : (q == 0 ? bl : (uint16_t)0);
  }
  for (int idx = tid; idx < 4 * 64 * 8; idx += EMR_WAVES * 64) {
    const int q = idx & 7, l = (idx >> 3) & 63, ks = idx >> 9;
    const int u = 32 * (ks >> 1) + emr_unit(8 * (ks & 1) + q, l >> 5);
    uint16_t ph, pm, pl;
    emr_split_host3(W.w2[(l & 31) * EM_H1 + u], &ph, &pm, &pl);
    W2f[((ks * 3 + 0) * 64 + l) * 8 + q] = ph;
    W2f[((ks * 3 + 1) * 64 + l) * 8 + q] = pm;
    W2f[((ks * 3 + 2) * 64 + l) * 8 + q] = pl;
  }
  __syncthreads();
  // The 26 weight fragments (104 registers) stay in LDS and are read where they are used (a lane's 16 bytes are consecutive:
  // conflict-free ds_read_b128): 119 registers instead of 224, i.e. four waves per SIMD instead of two (866 against 892 us
  // per 20.5 M edges). Measured and rejected (tools/time_edge_mlp.py): a three-chunk software pipeline that
  // puts one chunk's MFMAs beside another's vector work inside the wave (937 us: tools/mfma_valu_overlap.hip shows that
  // independent MFMA and vector streams of one wave take 0.89 of the SUM of their times on this chip, not the maximum).
  auto lw1 = [&](int a, int ks) {
    const int f = a * 2 + ks;
    return Frag3{*reinterpret_cast<const bf16x8*>(W1f + ((f * 3 + 0) * 64 + lane) * 8),
                 *reinterpret_cast<const bf16x8*>(W1f + ((f * 3 + 1) * 64 + lane) * 8),
                 *reinterpret_cast<const bf16x8*>(W1f + ((f * 3 + 2) * 64 + lane) * 8)};
  };
  auto lw2 = [&](int ks) {
    return Frag3{*reinterpret_cast<const bf16x8*>(W2f + ((ks * 3 + 0) * 64 + lane) * 8),
                 *reinterpret_cast<const bf16x8*>(W2f + ((ks * 3 + 1) * 64 + lane) * 8),
                 *reinterpret_cast<const bf16x8*>(W2f + ((ks * 3 + 2) * 64 + lane) * 8)};
  };
  auto lwe = [&](int a) { return *reinterpret_cast<const bf16x8*>(WEf + (a * 64 + lane) * 8); };
  f32x16 b2r;
  float w3r[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    b2r[r] = W.b2[emr_unit(r, h)];
    w3r[r] = W.w3[emr_unit(r, h)];
  }
  const float b3 = W.b3[0];
  const f32x16 zero = {0};
  emr_walk<1, false>(src, dst, edge_attr, obs, E, N, M, wave, lane, [&](const ERows<false>& cur, int32_t m, int32_t c, bool live) {
    asm volatile("" ::: "memory");      // keeps the LDS fragments out of loop-invariant registers
    const float vs[8] = {cur.s0.x, cur.s0.y, cur.s0.z, cur.s0.w, cur.s1.x, cur.s1.y, cur.s1.z, cur.s1.w};
    const float vd[8] = {cur.d0.x, cur.d0.y, cur.d0.z, cur.d0.w, cur.d1.x, cur.d1.y, cur.d1.z, cur.d1.w};
    const Frag3 xs = emr_split8(vs), xd = emr_split8(vd);
    // the edge_attr / bias k-step: {ea_h, ea_h, ea_h, ea_m, ea_m, ea_l, 1, 1} in the lower half-wave, {1, 0 ...} in the upper
    const Split3 se = emr_split2(cur.ea, 0.0f);
    const uint32_t eh = se.h & 0xFFFFu, em = se.m & 0xFFFFu, el = se.l & 0xFFFFu, one = 0x3F80u;
    u32x4 pe;
    pe[0] = h == 0 ? (eh | (eh << 16)) : one;
    pe[1] = h == 0 ? (eh | (em << 16)) : 0u;
    pe[2] = h == 0 ? (em | (el << 16)) : 0u;
    pe[3] = h == 0 ? (one | (one << 16)) : 0u;
    const bf16x8 xe = __builtin_bit_cast(bf16x8, pe);
    f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lwe(0), xe, zero, 0, 0, 0);
    f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lwe(1), xe, zero, 0, 0, 0);
    acc0 = emr_mma6(lw1(0, 0), xs, acc0);
    acc1 = emr_mma6(lw1(1, 0), xs, acc1);
    acc0 = emr_mma6(lw1(0, 1), xd, acc0);
    acc1 = emr_mma6(lw1(1, 1), xd, acc1);
    f32x16 c0 = b2r;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float hv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) hv[q] = emr_relu((ks < 2 ? acc0 : acc1)[8 * (ks & 1) + q]);
      c0 = emr_mma6(lw2(ks), emr_split8(hv), c0);
    }
    float part = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) part = fmaf(emr_relu(c0[r]), w3r[r], part);
    const float tot = part + __shfl_xor(part, 32);
    const int32_t e = c * 32 + j;
    if (live && h == 0 && e < (int32_t)E) logits[(int64_t)m * E + e] = tot + b3;
  });
}

// ---- backward -------------------------------------------------------------------------------------------------------------
// stage 1, on the matrix cores: the forward's register-resident recomputation per 32 edges (exact fp32 products), then
// dh2 = g w3 (masked by the second pre-activation) on the lane's own sixteen units and dh1^T [64][32] = W2^T dh2^T as
// 32 more MFMAs whose B operand is, again, what the lane already holds (A = W2 transposed, its k-order permuted to
// the accumulator layout), masked by h1 > 0. Everything is stored k-major ([k][L], L = M * E: a half-wave writes 32
// consecutive floats per unit) for the reductions of stage 2. 98 MFMAs per 32 edges.
__global__ __launch_bounds__(EMR_WAVES * 64) void k_edge_mlp_bwd_edges(const int32_t* __restrict__ src,
                                                                       const int32_t* __restrict__ dst, int64_t E,
                                                                       int64_t N, int64_t M,
                                                                       const float* __restrict__ obs,
                                                                       const float* __restrict__ edge_attr, EdgeMlpW W,
                                                                       const float* __restrict__ grad_logits,
                                                                       float* __restrict__ XT, float* __restrict__ H1T,
                                                                       float* __restrict__ H2T, float* __restrict__ D1T,
                                                                       float* __restrict__ D2T) {
  __shared__ float W2T[2 * 16 * 64];     // [tile a][k-step s][lane]: W2[unit(s, lane >> 5)][32 a + (lane & 31)]
  __shared__ __attribute__((aligned(16))) float B2L[2 * 16];
  __shared__ __attribute__((aligned(16))) float W3L[2 * 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the chunk walker lives in SGPRs
  const int h = lane >> 5, j = lane & 31;
  for (int idx = tid; idx < 2 * 16 * 64; idx += EMR_WAVES * 64) {
    const int l = idx & 63, ks = (idx >> 6) & 15, a = idx >> 10;
    W2T[idx] = W.w2[emr_unit(ks, l >> 5) * EM_H1 + 32 * a + (l & 31)];
  }
  if (tid < 32) {
    B2L[tid] = W.b2[emr_unit(tid & 15, tid >> 4)];
    W3L[tid] = W.w3[emr_unit(tid & 15, tid >> 4)];
  }
  __syncthreads();
  float w1a[2][17], w2a[32];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int u = 32 * a + j;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      w1a[a][t] = W.w1[u * EM_IN + 8 * h + t];
      w1a[a][8 + t] = W.w1[u * EM_IN + 16 + 8 * h + t];
    }
    w1a[a][16] = h == 0 ? W.w1[u * EM_IN + 32] : W.b1[u];
  }
#pragma unroll
  for (int t = 0; t < 32; ++t) w2a[t] = W.w2[j * EM_H1 + 32 * (t >> 4) + emr_unit(t & 15, h)];
  const float4* b2l = reinterpret_cast<const float4*>(B2L + 16 * h);
  const float4* w3l = reinterpret_cast<const float4*>(W3L + 16 * h);
  const f32x16 zero = {0};
  const int64_t L = M * E;
  ChunkWalk cw;
  if (!cw.init(E, M, wave)) return;
  EdgeIn nxt = emr_load(src, dst, edge_attr, obs, (int32_t)E, N, cw.mn, cw.cn, lane);
  for (; cw.g < cw.g1; ++cw.g) {
    const EdgeIn cur = nxt;
    cw.step();
    if (cw.g + 1 < cw.g1) nxt = emr_load(src, dst, edge_attr, obs, (int32_t)E, N, cw.mn, cw.cn, lane);
    const int32_t e = cw.c * 32 + j;
    const bool live = e < (int32_t)E;
    const int64_t l = (int64_t)cw.m * E + (live ? e : 0);
    const float xin[17] = {cur.s0.x, cur.s0.y, cur.s0.z, cur.s0.w, cur.s1.x, cur.s1.y, cur.s1.z, cur.s1.w,
                           cur.d0.x, cur.d0.y, cur.d0.z, cur.d0.w, cur.d1.x, cur.d1.y, cur.d1.z, cur.d1.w,
                           h == 0 ? cur.ea : 1.0f};
    f32x16 acc0 = zero, acc1 = zero;
#pragma unroll
    for (int t = 0; t < 17; ++t) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1a[0][t], xin[t], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1a[1][t], xin[t], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] = emr_relu(acc0[r]);
      acc1[r] = emr_relu(acc1[r]);
    }
    asm volatile("" ::: "memory");      // keeps the LDS tables out of loop-invariant registers
    f32x16 c0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = b2l[q];
      c0[4 * q] = v.x;
      c0[4 * q + 1] = v.y;
      c0[4 * q + 2] = v.z;
      c0[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2a[t], acc0[t], c0, 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 16; ++t) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w2a[16 + t], acc1[t], c0, 0, 0, 0);
    const float g = grad_logits[l];
    f32x16 dh2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = w3l[q];
      dh2[4 * q] = c0[4 * q] > 0.0f ? g * v.x : 0.0f;
      dh2[4 * q + 1] = c0[4 * q + 1] > 0.0f ? g * v.y : 0.0f;
      dh2[4 * q + 2] = c0[4 * q + 2] > 0.0f ? g * v.z : 0.0f;
      dh2[4 * q + 3] = c0[4 * q + 3] > 0.0f ? g * v.w : 0.0f;
    }
    f32x16 d0 = zero, d1 = zero;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(W2T[t * 64 + lane], dh2[t], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(W2T[(16 + t) * 64 + lane], dh2[t], d1, 0, 0, 0);
    }
    if (live) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        XT[(int64_t)(8 * h + t) * L + l] = xin[t];
        XT[(int64_t)(16 + 8 * h + t) * L + l] = xin[8 + t];
      }
      if (h == 0) XT[(int64_t)32 * L + l] = cur.ea;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t u = emr_unit(r, h);
        H1T[u * L + l] = acc0[r];
        H1T[(32 + u) * L + l] = acc1[r];
        D1T[u * L + l] = acc0[r] > 0.0f ? d0[r] : 0.0f;
        D1T[(32 + u) * L + l] = acc1[r] > 0.0f ? d1[r] : 0.0f;
        H2T[u * L + l] = emr_relu(c0[r]);
        D2T[u * L + l] = dh2[r];
      }
    }
  }
}

// stage 2: C[p][q] += sum_l A[p][l] * Bm[q][l] for an 8 x 8 block of outputs per workgroup (Bm == NULL: a row of ones,
// i.e. row sums), split over ND_SPLIT ranges of l (grid z): a handful of output blocks alone would leave the chip idle
// behind 300 000-long dot products. Fixed order everywhere — every thread strides over its range, a fixed LDS tree, the
// ranges added in order by k_nt_reduce: deterministic.
#define ND_T 256
#define ND_SPLIT 64
__global__ __launch_bounds__(ND_T) void k_nt_dot(const float* __restrict__ Am, int64_t P, const float* __restrict__ Bm,
                                                 int64_t Q, int64_t L, float* __restrict__ part) {
  __shared__ float red[ND_T];
  const int p0 = blockIdx.x * 8, q0 = blockIdx.y * 8;
  const int64_t span = (L + ND_SPLIT - 1) / ND_SPLIT, l0 = (int64_t)blockIdx.z * span;
  const int64_t l1 = l0 + span < L ? l0 + span : L;
  float acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.0f;
  for (int64_t l = l0 + threadIdx.x; l < l1; l += ND_T) {
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
      if (threadIdx.x == 0 && p0 + i < P && q0 + j < Q) part[((int64_t)blockIdx.z * P + p0 + i) * Q + q0 + j] = red[0];
      __syncthreads();
    }
}

__global__ __launch_bounds__(ND_T) void k_nt_reduce(const float* __restrict__ part, int64_t P, int64_t Q,
                                                    float* __restrict__ Cm, int64_t ldc) {
  const int64_t idx = (int64_t)blockIdx.x * ND_T + threadIdx.x;
  if (idx >= P * Q) return;
  float sum = 0.0f;
  for (int z = 0; z < ND_SPLIT; ++z) sum += part[(int64_t)z * P * Q + idx];
  Cm[(idx / Q) * ldc + idx % Q] += sum;
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
  TARL_REQUIRE(((uintptr_t)obs16) % 16 == 0, "obs16 must be 16-byte aligned");
  TARL_REQUIRE(ceil_div(plan->N, OB_TI) < 65536, "too many node tiles for one grid dimension");
  hipLaunchKernelGGL(k_obs16_packed, dim3((unsigned)ceil_div(B, OB_TB), (unsigned)ceil_div(plan->N, OB_TI)), dim3(256), 0,
                     (hipStream_t)stream, plan->out_ptr, plan->out_dst, x, L, B, plan->N, tarl_to_bufs(f), agent_features, A,
                     a_bstride, obs16);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_fused_obs16_bf16(const tarl_plan* plan, const tarl_fused* f, const float* x, int64_t B,
                                     int64_t x_bstride, int64_t ldx, int32_t Nmax, const float* agent_features, int64_t A,
                                     int64_t a_bstride, uint16_t* obs16, tarl_stream stream) {
  int rc = tarl_check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(x && agent_features && obs16 && A >= 1, "null argument");
  if (plan->N == 0) return TARL_OK;
  TARL_REQUIRE(((uintptr_t)obs16) % 16 == 0, "obs16 must be 16-byte aligned");
  TARL_REQUIRE(ceil_div(plan->N, OB_TI) < 65536, "too many node tiles for one grid dimension");
  const Layout L{Nmax, ldx, x_bstride};
  hipLaunchKernelGGL(k_obs16_packed_bf16, dim3((unsigned)ceil_div(B, OB_TB), (unsigned)ceil_div(plan->N, OB_TI)), dim3(256),
                     0, (hipStream_t)stream, plan->out_ptr, plan->out_dst, x, L, B, plan->N, tarl_to_bufs(f), agent_features,
                     A, a_bstride, obs16);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_fused_obs16_rows(const tarl_plan* plan, const tarl_fused* f, const float* x, int64_t B,
                                     int64_t x_bstride, int64_t ldx, int32_t Nmax, const float* agent_features, int64_t A,
                                     int64_t a_bstride, const int32_t* env, const int32_t* slot, int64_t rows,
                                     float* obs_rows, tarl_stream stream) {
  int rc = tarl_check_fused_core(plan, f, B, Nmax);
  if (rc) return rc;
  TARL_REQUIRE(x && agent_features && A >= 1 && rows >= 0 && rows < 65536, "bad arguments");
  if (plan->N == 0 || rows == 0) return TARL_OK;
  TARL_REQUIRE(env && slot && obs_rows && ((uintptr_t)obs_rows) % 16 == 0, "row lists / 16-byte aligned output missing");
  const Layout L{Nmax, ldx, x_bstride};
  hipLaunchKernelGGL(k_obs16_rows, dim3((unsigned)ceil_div(plan->N, FB), (unsigned)rows), dim3(FB), 0, (hipStream_t)stream,
                     plan->out_ptr, plan->out_dst, x, L, B, plan->N, tarl_to_bufs(f), agent_features, A, a_bstride, env, slot,
                     obs_rows);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_policy_edge_mlp_fwd(const tarl_plan* plan, const float* obs16, int64_t M, const float* edge_attr,
                                        const float* w1, const float* b1, const float* w2, const float* b2,
                                        const float* w3, const float* b3, int precision, float* logits,
                                        tarl_stream stream) {
  TARL_REQUIRE(plan && obs16 && edge_attr && w1 && b1 && w2 && b2 && w3 && b3 && logits, "null argument");
  TARL_REQUIRE(M >= 1 && M < 65536, "bad batch size");
  TARL_REQUIRE(precision >= 0 && precision <= 3,
               "precision: 0 = fp32 MFMA, 1 = bf16 MFMA, 2 = bf16 MFMA on bf16 observations, 3 = fp32-accurate bf16x3 MFMA");
  TARL_REQUIRE(((uintptr_t)obs16) % 16 == 0, "obs16 must be 16-byte aligned");
  if (plan->E == 0) return TARL_OK;
  const EdgeMlpW W{w1, b1, w2, b2, w3, b3};
  // persistent waves: every wave builds its weight fragments once and walks a contiguous range of 32-edge chunks
  const int64_t chunks = M * ceil_div(plan->E, 32);
  TARL_REQUIRE(chunks < ((int64_t)1 << 31) && plan->E < ((int64_t)1 << 31) - 32 && plan->N < ((int64_t)1 << 26),
               "edge MLP: batch x edges too large");
  int64_t blocks = ceil_div(chunks, (int64_t)EMR_WAVES * 16);          // >= 16 chunks per wave
  // workgroups the chip holds at once (VGPR-bound): fp32 2 per CU, bf16 on fp32 rows 3, bf16 on bf16 rows 4
  const int64_t resident = 256 * (precision == 0 || precision == 3 ? 2 : (precision == 2 ? 4 : 3));
  if (blocks > resident) blocks = resident;                             // one round: no tail
  if (precision == 3) {
    blocks = ceil_div(chunks, (int64_t)EMR_WAVES * 16);
    if (blocks > 256 * 4) blocks = 256 * 4;      // 119 registers, 26 KB of LDS: four workgroups per CU, one round
    hipLaunchKernelGGL(k_edge_mlp_fwd_x3, dim3((unsigned)blocks), dim3(EMR_WAVES * 64), 0, (hipStream_t)stream,
                       plan->src, plan->dst, plan->E, plan->N, M, obs16, edge_attr, W, logits);
  }
  else if (precision == 0)
    hipLaunchKernelGGL(k_edge_mlp_fwd_f32, dim3((unsigned)blocks), dim3(EMR_WAVES * 64), 0, (hipStream_t)stream,
                       plan->src, plan->dst, plan->E, plan->N, M, obs16, edge_attr, W, logits);
  else if (precision == 1)
    hipLaunchKernelGGL(k_edge_mlp_fwd_bf16<false>, dim3((unsigned)blocks), dim3(EMR_WAVES * 64), 0, (hipStream_t)stream,
                       plan->src, plan->dst, plan->E, plan->N, M, (const void*)obs16, edge_attr, W, logits);
  else
    hipLaunchKernelGGL(k_edge_mlp_fwd_bf16<true>, dim3((unsigned)blocks), dim3(EMR_WAVES * 64), 0, (hipStream_t)stream,
                       plan->src, plan->dst, plan->E, plan->N, M, (const void*)obs16, edge_attr, W, logits);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int64_t tarl_policy_edge_mlp_bwd_scratch_floats(const tarl_plan* plan, int64_t M) {
  // k-major activations and their gradients + the split partial sums of the largest weight gradient
  return plan ? (int64_t)(EM_IN + 2 * EM_H1 + 2 * EM_H2) * M * plan->E + (int64_t)ND_SPLIT * EM_H1 * EM_H1 : -1;
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
  float* part = D2T + EM_H2 * L;          // [ND_SPLIT][P][Q], P * Q <= 64 * 64
  const EdgeMlpW W{w1, b1, w2, b2, w3, b3};
  hipStream_t s = (hipStream_t)stream;
  {
    const int64_t chunks = M * ceil_div(plan->E, 32);
    TARL_REQUIRE(chunks < ((int64_t)1 << 31) && plan->E < ((int64_t)1 << 31) - 32, "edge MLP: batch x edges too large");
    int64_t blocks = ceil_div(chunks, (int64_t)EMR_WAVES * 4);
    if (blocks > 256) blocks = 256;          // one resident workgroup per CU (the kernel is register-heavy)
    hipLaunchKernelGGL(k_edge_mlp_bwd_edges, dim3((unsigned)blocks), dim3(EMR_WAVES * 64), 0, s, plan->src, plan->dst,
                       plan->E, plan->N, M, obs16, edge_attr, W, grad_logits, XT, H1T, H2T, D1T, D2T);
  }
  TARL_LAUNCH_CHECK();
  auto dot = [&](const float* Am, int64_t P, const float* Bm, int64_t Q, float* Cm, int64_t ldc) {
    hipLaunchKernelGGL(k_nt_dot, dim3((unsigned)ceil_div(P, 8), (unsigned)ceil_div(Q, 8), ND_SPLIT), dim3(ND_T), 0, s, Am,
                       P, Bm, Q, L, part);
    hipLaunchKernelGGL(k_nt_reduce, dim3((unsigned)ceil_div(P * Q, ND_T)), dim3(ND_T), 0, s, part, P, Q, Cm, ldc);
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
