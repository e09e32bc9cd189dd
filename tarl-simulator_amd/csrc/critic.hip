// critic.hip — MPNNValueNetSimple: value = MLP(cat(NUMBER_OF_AGENT per node, time)), (N+1) -> 64 -> 64 -> 1 with ReLU
// (src/agents/mpnn_agent.py:420-450). The only GEMM-shaped work on the path, hence the only MFMA user.
//
// Forward: one fused kernel. A workgroup owns 128 rows; each of its 4 waves accumulates a 32 x 64 slab of the first
// layer with v_mfma_f32_32x32x2_f32 (exact fp32 products, k-ordered fma chain: no reduced-precision path is taken, the
// parity contract is 1e-4 on values), staging X and W1 tiles k-major in LDS (padded: conflict-free ds_read_b32 for the
// MFMA operand pattern "lane -> row, half-wave -> k"). The 128 x 64 hidden tile never leaves the CU: layer 2 is a
// second MFMA pass out of LDS, layer 3 a 64-long dot product per row.
// The per-row `time` feature is the last input column in the reference; here it is a rank-1 update after the K loop,
// which keeps the big operand (counts, row stride = N) 16-byte friendly and shared with the rollout buffer.
#include "tarl_common.h"

#define CR_H 64        // hidden width (fixed by the reference architecture)
#define CR_BM 128      // rows per workgroup
#define CR_BK 32       // K chunk
#define CR_THREADS 256

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct CriticParams {
  const float* w1;  // [64][N+1]
  const float* b1;  // [64]
  const float* w2;  // [64][64]
  const float* b2;  // [64]
  const float* w3;  // [64]
  const float* b3;  // [1]
};

// C/D layout of the 32x32 f32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// rps == 0: counts is row-major [M][ldc]. rps > 0 ("slab" mode, rps % CR_BM == 0): counts is [M / rps][N][rps] — the
// env-minor rollout buffer [frame][node][env] — i.e. k-major inside each slab of rps rows, which is exactly the order
// the LDS staging wants (lanes along rows => coalesced).
// XT: float (observations as the reference holds them) or uint8_t (the rollout buffers' count bytes, widened in the staging).
template <typename XT>
__global__ __launch_bounds__(CR_THREADS) void k_critic_fwd(const XT* __restrict__ counts, int64_t ldc, int64_t rps,
                                                           int64_t M, int64_t N,
                                                           const float* __restrict__ time_rows,
                                                           int64_t rows_per_time, CriticParams P,
                                                           float* __restrict__ value, float* __restrict__ h1_out,
                                                           float* __restrict__ h2_out) {
  // one array for all LDS (phases reuse it): Xs [32][129] + Ws [32][65]  |  Hs [64][129] + W2s [64][65]
  __shared__ float lds[CR_H * (CR_BM + 1) + CR_H * (CR_H + 1)];
  float* Xs = lds;                          // [k][row], row stride 129
  float* Ws = lds + CR_BK * (CR_BM + 1);    // [k][j],  stride 65
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * CR_BM;
  const int64_t ldw = N + 1;

  f32x16 acc0 = {0}, acc1 = {0};
  for (int64_t k0 = 0; k0 < N; k0 += CR_BK) {
    if (rps == 0) {
      // stage X tile: 128 rows x 32 k  (row-major input: lanes along k => 128-B coalesced segments)
#pragma unroll
      for (int it = 0; it < (CR_BM * CR_BK) / CR_THREADS; ++it) {
        const int idx = it * CR_THREADS + tid;
        const int r = idx >> 5, k = idx & 31;
        const int64_t gr = row0 + r, gk = k0 + k;
        Xs[k * (CR_BM + 1) + r] = (gr < M && gk < N) ? (float)counts[gr * ldc + gk] : 0.0f;
      }
    } else {
      // slab input: lanes along rows => 256-B coalesced segments, conflict-free LDS stores
      const int64_t slab_base = (row0 / rps) * (N * rps) + (row0 % rps);
#pragma unroll
      for (int it = 0; it < (CR_BM * CR_BK) / CR_THREADS; ++it) {
        const int idx = it * CR_THREADS + tid;
        const int r = idx & (CR_BM - 1), k = idx >> 7;
        const int64_t gr = row0 + r, gk = k0 + k;
        Xs[k * (CR_BM + 1) + r] = (gr < M && gk < N) ? (float)counts[slab_base + gk * rps + r] : 0.0f;
      }
    }
#pragma unroll
    for (int it = 0; it < (CR_H * CR_BK) / CR_THREADS; ++it) {
      const int idx = it * CR_THREADS + tid;
      const int j = idx >> 5, k = idx & 31;
      const int64_t gk = k0 + k;
      Ws[k * (CR_H + 1) + j] = (gk < N) ? P.w1[(int64_t)j * ldw + gk] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < CR_BK; kk += 2) {
      const int k = kk + (lane >> 5);
      const float a = Xs[k * (CR_BM + 1) + wave * 32 + (lane & 31)];
      const float b0 = Ws[k * (CR_H + 1) + (lane & 31)];
      const float b1 = Ws[k * (CR_H + 1) + 32 + (lane & 31)];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
    __syncthreads();
  }

  // epilogue of layer 1: + time * W1[:, N] + b1, ReLU; park h1 k-major in LDS for the second MFMA pass
  float* Hs = lds;                           // [j][row], stride 129
  float* W2s = lds + CR_H * (CR_BM + 1);     // [k][j],  stride 65   (W2s[k][j] = w2[j][k])
  {
    const int j0 = lane & 31;
    const float wt0 = P.w1[(int64_t)j0 * ldw + N], wt1 = P.w1[(int64_t)(j0 + 32) * ldw + N];
    const float bb0 = P.b1[j0], bb1 = P.b1[j0 + 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = wave * 32 + mfma_row(r, lane);
      const int64_t gr = row0 + lr;
      const float tm = (gr < M) ? time_rows[gr / rows_per_time] : 0.0f;
      float v0 = acc0[r] + tm * wt0 + bb0;
      float v1 = acc1[r] + tm * wt1 + bb1;
      v0 = v0 > 0.0f ? v0 : 0.0f;
      v1 = v1 > 0.0f ? v1 : 0.0f;
      Hs[j0 * (CR_BM + 1) + lr] = v0;
      Hs[(j0 + 32) * (CR_BM + 1) + lr] = v1;
      if (h1_out && gr < M) {
        h1_out[gr * CR_H + j0] = v0;
        h1_out[gr * CR_H + j0 + 32] = v1;
      }
    }
  }
#pragma unroll
  for (int it = 0; it < (CR_H * CR_H) / CR_THREADS; ++it) {
    const int idx = it * CR_THREADS + tid;
    const int j = idx >> 6, k = idx & 63;
    W2s[k * (CR_H + 1) + j] = P.w2[j * CR_H + k];
  }
  __syncthreads();
  f32x16 c0 = {0}, c1 = {0};
#pragma unroll
  for (int kk = 0; kk < CR_H; kk += 2) {
    const int k = kk + (lane >> 5);
    const float a = Hs[k * (CR_BM + 1) + wave * 32 + (lane & 31)];
    const float b0 = W2s[k * (CR_H + 1) + (lane & 31)];
    const float b1 = W2s[k * (CR_H + 1) + 32 + (lane & 31)];
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, c1, 0, 0, 0);
  }
  __syncthreads();  // everyone is done reading Hs / W2s
  {
    const int j0 = lane & 31;
    const float bb0 = P.b2[j0], bb1 = P.b2[j0 + 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = wave * 32 + mfma_row(r, lane);
      const int64_t gr = row0 + lr;
      float v0 = c0[r] + bb0, v1 = c1[r] + bb1;
      v0 = v0 > 0.0f ? v0 : 0.0f;
      v1 = v1 > 0.0f ? v1 : 0.0f;
      Hs[j0 * (CR_BM + 1) + lr] = v0;
      Hs[(j0 + 32) * (CR_BM + 1) + lr] = v1;
      if (h2_out && gr < M) {
        h2_out[gr * CR_H + j0] = v0;
        h2_out[gr * CR_H + j0 + 32] = v1;
      }
    }
  }
  __syncthreads();
  if (tid < CR_BM) {
    const int64_t gr = row0 + tid;
    if (gr < M) {
      float s = 0.0f;
#pragma unroll 8
      for (int j = 0; j < CR_H; ++j) s += Hs[j * (CR_BM + 1) + tid] * P.w3[j];
      value[gr] = s + P.b3[0];
    }
  }
}

// ---- slab-input forward, software pipelined ----------------------------------------------------------------------------
// Same arithmetic as k_critic_fwd (same tiles, same k order, same MFMA chain => bit-identical values) for the env-minor
// rollout buffer counts [M / rps][N][rps]. Differences are purely scheduling: the X / W1 tiles of K-chunk c+1 are
// fetched into registers (float4 along the row dimension for X) BEFORE the MFMAs of chunk c and stored into the other
// LDS buffer after them, so global-memory latency hides behind the matrix work and there is one barrier per chunk.
#define CRS_LDX (CR_BM + 4)   // X tile row stride (floats): keeps the float4 LDS stores 16-byte aligned
__device__ __forceinline__ float4 load_x4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load_x4(const uint8_t* p) {   // four count bytes in one dword
  const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
  return make_float4((float)(w & 255u), (float)((w >> 8) & 255u), (float)((w >> 16) & 255u), (float)(w >> 24));
}
template <typename XT>
__global__ __launch_bounds__(CR_THREADS) void k_critic_fwd_slab(const XT* __restrict__ counts, int64_t rps, int64_t M,
                                                                int64_t N, const float* __restrict__ time_rows,
                                                                int64_t rows_per_time, CriticParams P,
                                                                float* __restrict__ value) {
  // [2 buffers] x (Xs [32][132] + Ws [32][65]); the epilogue reuses the space as Hs [64][129] + W2s [64][65]
  __shared__ __attribute__((aligned(16))) float lds[2 * (CR_BK * CRS_LDX + CR_BK * (CR_H + 1))];
  const int XS = CR_BK * CRS_LDX, BUF = XS + CR_BK * (CR_H + 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * CR_BM;
  const int64_t ldw = N + 1;
  const XT* xbase = counts + (row0 / rps) * (N * rps) + (row0 % rps);   // element (r, k) at xbase[k * rps + r]

  float4 xr[4];
  float wr[8];
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = it * CR_THREADS + tid;
      const int r4 = (idx & 31) * 4, k = idx >> 5;
      const int64_t gk = k0 + k;
      xr[it] = (gk < N) ? load_x4(xbase + gk * rps + r4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * CR_THREADS + tid;
      const int j = idx >> 5, k = idx & 31;
      const int64_t gk = k0 + k;
      wr[it] = (gk < N) ? P.w1[(int64_t)j * ldw + gk] : 0.0f;
    }
  };
  auto stash = [&](int buf) {
    float* Xs = lds + buf * BUF;
    float* Ws = Xs + XS;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = it * CR_THREADS + tid;
      const int r4 = (idx & 31) * 4, k = idx >> 5;
      *reinterpret_cast<float4*>(Xs + k * CRS_LDX + r4) = xr[it];
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int idx = it * CR_THREADS + tid;
      const int j = idx >> 5, k = idx & 31;
      Ws[k * (CR_H + 1) + j] = wr[it];
    }
  };

  f32x16 acc0 = {0}, acc1 = {0};
  fetch(0);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int64_t k0 = 0; k0 < N; k0 += CR_BK) {
    const bool more = k0 + CR_BK < N;
    if (more) fetch(k0 + CR_BK);             // global loads in flight during the MFMAs below
    const float* Xs = lds + buf * BUF;
    const float* Ws = Xs + XS;
#pragma unroll
    for (int kk = 0; kk < CR_BK; kk += 2) {
      const int k = kk + (lane >> 5);
      const float a = Xs[k * CRS_LDX + wave * 32 + (lane & 31)];
      const float b0 = Ws[k * (CR_H + 1) + (lane & 31)];
      const float b1 = Ws[k * (CR_H + 1) + 32 + (lane & 31)];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
    }
    if (more) stash(buf ^ 1);                // the other buffer: nobody reads it in this iteration
    __syncthreads();
    buf ^= 1;
  }

  // epilogue: identical to k_critic_fwd
  float* Hs = lds;                           // [j][row], stride 129
  float* W2s = lds + CR_H * (CR_BM + 1);     // [k][j],  stride 65
  {
    const int j0 = lane & 31;
    const float wt0 = P.w1[(int64_t)j0 * ldw + N], wt1 = P.w1[(int64_t)(j0 + 32) * ldw + N];
    const float bb0 = P.b1[j0], bb1 = P.b1[j0 + 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = wave * 32 + mfma_row(r, lane);
      const int64_t gr = row0 + lr;
      const float tm = (gr < M) ? time_rows[gr / rows_per_time] : 0.0f;
      float v0 = acc0[r] + tm * wt0 + bb0;
      float v1 = acc1[r] + tm * wt1 + bb1;
      v0 = v0 > 0.0f ? v0 : 0.0f;
      v1 = v1 > 0.0f ? v1 : 0.0f;
      Hs[j0 * (CR_BM + 1) + lr] = v0;
      Hs[(j0 + 32) * (CR_BM + 1) + lr] = v1;
    }
  }
#pragma unroll
  for (int it = 0; it < (CR_H * CR_H) / CR_THREADS; ++it) {
    const int idx = it * CR_THREADS + tid;
    const int j = idx >> 6, k = idx & 63;
    W2s[k * (CR_H + 1) + j] = P.w2[j * CR_H + k];
  }
  __syncthreads();
  f32x16 c0 = {0}, c1 = {0};
#pragma unroll
  for (int kk = 0; kk < CR_H; kk += 2) {
    const int k = kk + (lane >> 5);
    const float a = Hs[k * (CR_BM + 1) + wave * 32 + (lane & 31)];
    const float b0 = W2s[k * (CR_H + 1) + (lane & 31)];
    const float b1 = W2s[k * (CR_H + 1) + 32 + (lane & 31)];
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, c1, 0, 0, 0);
  }
  __syncthreads();
  {
    const int j0 = lane & 31;
    const float bb0 = P.b2[j0], bb1 = P.b2[j0 + 32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = wave * 32 + mfma_row(r, lane);
      float v0 = c0[r] + bb0, v1 = c1[r] + bb1;
      v0 = v0 > 0.0f ? v0 : 0.0f;
      v1 = v1 > 0.0f ? v1 : 0.0f;
      Hs[j0 * (CR_BM + 1) + lr] = v0;
      Hs[(j0 + 32) * (CR_BM + 1) + lr] = v1;
    }
  }
  __syncthreads();
  if (tid < CR_BM) {
    const int64_t gr = row0 + tid;
    if (gr < M) {
      float s = 0.0f;
#pragma unroll 8
      for (int j = 0; j < CR_H; ++j) s += Hs[j * (CR_BM + 1) + tid] * P.w3[j];
      value[gr] = s + P.b3[0];
    }
  }
}

// ---- slab-input forward on the rollout's count BYTES, first layer on the bf16 matrix cores at fp32 accuracy ----------------
// The observation is a small integer per node (NUMBER_OF_AGENT <= 255): exact in bf16. W1 is split once per call into three
// bf16 pieces with hi + mid + lo == W1 exactly (8 + 8 + 8 significand bits), so
//     h1 = X W1^T = X hi^T + X mid^T + X lo^T
// with every product exact (8-bit count x 8-bit piece) and fp32 accumulation: the same value as the fp32 chain up to the
// order of the fp32 additions (~1e-7 relative; contract 1e-4), at 16x the MFMA rate per piece. The pass then runs at HBM
// speed on 1 byte per (frame, node, environment) instead of at the fp32-MFMA rate on 4.
// Orientation: D^T = W X^T: the count bytes are the B operand — lane (col = env, half h) wants the 8 bytes
// X^T[k = 8h .. 8h+7][env], i.e. 8 consecutive k of ONE environment, while memory (and a straight copy of it in LDS) is
// env-contiguous per k. The staging therefore transposes 4 x 4 byte blocks in registers on the way in (a thread loads 4
// environments of 4 consecutive k, writes 4 dwords [env][4 k]): the fragment is then two dword reads instead of eight
// byte reads per k-step. A = the weight pieces, k-contiguous.
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
#define CB_XLD 36      // bytes per ENVIRONMENT row of the transposed X stage (32 k + 4 pad: odd dword stride, conflict-free)
#define CB_WLD 40      // bf16 per j-row of a W stage (32 k + pad): 80-byte rows, 16-byte aligned

__device__ __forceinline__ uint16_t cr_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// w1 [64][N+1] -> w3 [3][64][Kpad] bf16 (k >= N zero-padded): hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid)
__global__ __launch_bounds__(CR_THREADS) void k_split_w1(const float* __restrict__ w1, int64_t N, int64_t Kpad,
                                                         uint16_t* __restrict__ w3) {
  const int64_t idx = (int64_t)blockIdx.x * CR_THREADS + threadIdx.x;
  if (idx >= CR_H * Kpad) return;
  const int64_t j = idx / Kpad, k = idx - j * Kpad;
  const float w = k < N ? w1[j * (N + 1) + k] : 0.0f;
  const uint16_t hi = cr_bf16_rne(w);
  const float r1 = w - __uint_as_float((uint32_t)hi << 16);
  const uint16_t mid = cr_bf16_rne(r1);
  const float r2 = r1 - __uint_as_float((uint32_t)mid << 16);
  const uint16_t lo = cr_bf16_rne(r2);
  w3[idx] = hi;
  w3[CR_H * Kpad + idx] = mid;
  w3[2 * CR_H * Kpad + idx] = lo;
}

// CT column tiles of 32 environments per wave: a workgroup owns 128 * CT environments of one frame. The three weight pieces
// of a k-tile (12 KB) are read from L2 once per workgroup and k-tile whatever CT is, the count bytes are 4 KB * CT: with
// CT = 1 (round 3) the pass moved 3 bytes of weights through the L2 for every byte of counts from HBM — 32 GB of L2 -> LDS
// traffic beside the 10.5 GB stream at 16 384 environments, 1.9 TB/s of HBM; CT = 4 cuts the weight traffic by four.
template <int CT>
__global__ __launch_bounds__(CR_THREADS) void k_critic_fwd_slab_u8x3(const uint8_t* __restrict__ counts, int64_t rps,
                                                                     int64_t M, int64_t N, int64_t Kpad,
                                                                     const float* __restrict__ time_rows,
                                                                     int64_t rows_per_time,
                                                                     const uint16_t* __restrict__ w3, CriticParams P,
                                                                     float* __restrict__ value) {
  // staging: 2 x (Xs [128 CT envs][36] bytes + Wb [3][64][40] bf16); the epilogue reuses the space as Hs [64][129] + W2s [64][65] f32
  constexpr int XB = CT * CR_BM * CB_XLD, WB = 3 * CR_H * CB_WLD * 2, BUF = XB + WB;   // bytes
  constexpr int EPI = (CR_H * (CR_BM + 1) + CR_H * (CR_H + 1)) * 4;
  __shared__ __attribute__((aligned(16))) uint8_t lds_raw[2 * BUF > EPI ? 2 * BUF : EPI];
  static_assert(CR_BK == 32 && CR_BM == 128 && CR_THREADS == 256 && (CR_BM * CB_XLD) % 16 == 0, "X staging plan");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = (int64_t)blockIdx.x * (CR_BM * CT);
  const int64_t ldw = N + 1;
  const uint8_t* xbase = counts + (row0 / rps) * (N * rps) + (row0 % rps);   // element (env r, node k) at xbase[k * rps + r]
  const int64_t WP = CR_H * Kpad;      // one weight piece

  // Register stages. The WEIGHT pieces of tile i + 2 (L2-resident, 12 KB) are requested while tile i is multiplied and tile
  // i + 1 is written to LDS; the COUNT bytes of tile i + 4 (HBM, 4 KB * CT) at the same moment, i.e. four tiles ahead (round 5;
  // two until then). An iteration is 12 CT MFMAs ≈ 0.6 us and a workgroup had 16 KB of counts in flight: 8 MB on the whole
  // chip against the ≈3 us of a loaded HBM round trip, i.e. 2.3-2.7 TB/s — the rate the pass ran at, whatever its instruction
  // count or occupancy (profiles/r05_ab_update.txt; a plain read of the same bytes in the same 256-byte pieces reaches
  // 6.1 TB/s, tools/dram_pattern.hip). The memory counter retires in order: inside an iteration the weights are requested
  // BEFORE the far-ahead counts, so waiting for them leaves the younger count tiles in flight. Loads are unconditional (a k
  // past the end re-reads the last row: its weight pieces are zero padding; a tile past the end re-reads the last tile): no
  // divergent control flow, exact wait counts.
  struct XStage {
    uint32_t xr[CT][4];    // per 128-environment block c: 4 environments (4 * (tid & 31) ..) of 4 consecutive k (4 * (tid >> 5) ..)
  };
  typedef uint32_t cu32x4 __attribute__((ext_vector_type(4)));      // (a native vector: HIP's uint4 is copied by memcpy, and with
  struct WStage {                                                  // six stages in rotation those copies kept the stage in scratch)
    cu32x4 p0, p1, p2;     // 8 bf16 of each piece: j = tid >> 2, k = 8 * (tid & 3) ..
  };
  const int64_t NIT = (N + CR_BK - 1) / CR_BK;
  // this thread's four count rows of a k-tile: k0 + 4 kg + i. A tile inside the matrix is addressed from one pointer and the
  // row stride; only a tile that reaches past row N - 1 (the last one or two requests) clamps row by row (uniform branch):
  // the clamps and 64-bit products of every row were 50 of the pass's ~150 vector instructions per k-tile
  const uint8_t* xthread = xbase + (int64_t)(4 * (tid >> 5)) * rps + 4 * (tid & 31);
  auto fetch_x = [&](XStage& st, int64_t it) __attribute__((always_inline)) {
    const int64_t k0 = (it < NIT ? it : NIT - 1) * CR_BK;
    if (k0 + CR_BK <= N) {
      const uint8_t* xk = xthread + k0 * rps;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int c = 0; c < CT; ++c) st.xr[c][i] = *reinterpret_cast<const uint32_t*>(xk + CR_BM * c);
        xk += rps;
      }
    } else {
      const int kg = tid >> 5, eg = tid & 31;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t k = k0 + 4 * kg + i;
        const uint8_t* xk = xbase + (k < N ? k : N - 1) * rps + 4 * eg;
#pragma unroll
        for (int c = 0; c < CT; ++c) st.xr[c][i] = *reinterpret_cast<const uint32_t*>(xk + CR_BM * c);
      }
    }
  };
  auto fetch_w = [&](WStage& st, int64_t it) __attribute__((always_inline)) {
    const int64_t k0 = (it < NIT ? it : NIT - 1) * CR_BK;
    const int j = tid >> 2, q = tid & 3;
    const uint16_t* wp = w3 + (int64_t)j * Kpad + k0 + 8 * q;
    st.p0 = *reinterpret_cast<const cu32x4*>(wp);
    st.p1 = *reinterpret_cast<const cu32x4*>(wp + WP);
    st.p2 = *reinterpret_cast<const cu32x4*>(wp + 2 * WP);
  };
  auto stash = [&](const XStage& sx, const WStage& sw, int buf) __attribute__((always_inline)) {
    uint8_t* Xs = lds_raw + buf * BUF;
    uint16_t* Wb = reinterpret_cast<uint16_t*>(Xs + XB);
    const int kg = tid >> 5, eg = tid & 31;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      // 4 x 4 byte transposition in eight v_perm_b32 (two rounds of byte interleaves) instead of 28 shifts / masks / ors:
      // byte e of the four k-rows -> one dword [env 128 c + 4 eg + e][k 4 kg .. 4 kg + 3]
      const uint32_t r0 = sx.xr[c][0], r1 = sx.xr[c][1], r2 = sx.xr[c][2], r3 = sx.xr[c][3];
      const uint32_t a_lo = __builtin_amdgcn_perm(r1, r0, 0x05010400u), a_hi = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
      const uint32_t b_lo = __builtin_amdgcn_perm(r3, r2, 0x05010400u), b_hi = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
      const uint32_t v[4] = {__builtin_amdgcn_perm(b_lo, a_lo, 0x05040100u), __builtin_amdgcn_perm(b_lo, a_lo, 0x07060302u),
                             __builtin_amdgcn_perm(b_hi, a_hi, 0x05040100u), __builtin_amdgcn_perm(b_hi, a_hi, 0x07060302u)};
#pragma unroll
      for (int e = 0; e < 4; ++e) *reinterpret_cast<uint32_t*>(Xs + (CR_BM * c + 4 * eg + e) * CB_XLD + 4 * kg) = v[e];
    }
    const int j = tid >> 2, q = tid & 3;
    *reinterpret_cast<cu32x4*>(Wb + (0 * CR_H + j) * CB_WLD + 8 * q) = sw.p0;
    *reinterpret_cast<cu32x4*>(Wb + (1 * CR_H + j) * CB_WLD + 8 * q) = sw.p1;
    *reinterpret_cast<cu32x4*>(Wb + (2 * CR_H + j) * CB_WLD + 8 * q) = sw.p2;
  };

  // D^T: rows j (0..31 / 32..63), cols = 32 environments; the wave's tile c covers environments 128 c + 32 wave .. + 31
  f32x16 acc0[CT], acc1[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    acc0[c] = (f32x16){0};
    acc1[c] = (f32x16){0};
  }
  const int r32 = lane & 31, h8 = (lane >> 5) * 8;
  auto multiply = [&](int buf) __attribute__((always_inline)) {
    const uint8_t* Xs = lds_raw + buf * BUF;
    const uint16_t* Wb = reinterpret_cast<const uint16_t*>(Xs + XB);
#pragma unroll
    for (int s = 0; s < CR_BK / 16; ++s) {
      // B fragment: lane (col r32 = env, half h) holds X^T[k = 16 s + 8 h + j][env], j = 0..7: small integers, exact in bf16
      cbf16x8 bx[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const uint32_t* xw = reinterpret_cast<const uint32_t*>(Xs + (CR_BM * c + wave * 32 + r32) * CB_XLD + 16 * s + h8);
        const uint32_t x0 = xw[0], x1 = xw[1];
        uint32_t bw[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {      // two counts -> two bf16 (the top halves of their fp32 images: exact below 256)
          const uint32_t xs = j < 2 ? x0 : x1;
          const float c0 = (float)((xs >> (16 * (j & 1))) & 0xFFu), c1 = (float)((xs >> (16 * (j & 1) + 8)) & 0xFFu);
          bw[j] = __builtin_amdgcn_perm(__float_as_uint(c1), __float_as_uint(c0), 0x07060302u);   // {c1.hi16, c0.hi16}: one v_perm_b32
        }
        bx[c] = __builtin_bit_cast(cbf16x8, bw);
      }
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        const cbf16x8 a0 = *reinterpret_cast<const cbf16x8*>(Wb + (pc * CR_H + r32) * CB_WLD + 16 * s + h8);
        const cbf16x8 a1 = *reinterpret_cast<const cbf16x8*>(Wb + (pc * CR_H + 32 + r32) * CB_WLD + 16 * s + h8);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          acc0[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bx[c], acc0[c], 0, 0, 0);
          acc1[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bx[c], acc1[c], 0, 0, 0);
        }
      }
    }
  };
  XStage x0, x1, x2, x3;      // counts of tiles t with t % 4 == 0 .. 3
  WStage w0, w1;              // weight pieces of tiles t with t % 2 == 0, 1
  fetch_w(w0, 0);
  fetch_x(x0, 0);
  fetch_w(w1, 1);
  fetch_x(x1, 1);
  fetch_x(x2, 2);
  fetch_x(x3, 3);
  stash(x0, w0, 0);
  __syncthreads();
  // iteration t: request weights t + 2 and counts t + 4 (into the stages tile t's data have just left), multiply tile t
  // (LDS buffer t % 2), write tile t + 1 into the other buffer
#define CR_STEP(T_, XA_, XB_, WA_, WB_, BUF_)                                           \
  {                                                                                     \
    fetch_w(WA_, (T_) + 2);                                                             \
    fetch_x(XA_, (T_) + 4);                                                             \
    if ((T_) < NIT) multiply(BUF_); /* (uniform condition, no loads inside) */          \
    stash(XB_, WB_, (BUF_) ^ 1);                                                        \
    __syncthreads();                                                                    \
  }
  for (int64_t it = 0; it < NIT; it += 4) {
    CR_STEP(it, x0, x1, w0, w1, 0);
    CR_STEP(it + 1, x1, x2, w1, w0, 1);
    CR_STEP(it + 2, x2, x3, w0, w1, 0);
    CR_STEP(it + 3, x3, x0, w1, w0, 1);
  }
#undef CR_STEP

  // epilogue, one 128-environment block c at a time through the same LDS: + time * W1[:, N] + b1, ReLU -> Hs [j][env]; then
  // the second and third layer exactly as k_critic_fwd_slab
  float* Hs = reinterpret_cast<float*>(lds_raw);     // [j][row], stride 129
  float* W2s = Hs + CR_H * (CR_BM + 1);              // [k][j],  stride 65
#pragma unroll
  for (int it = 0; it < (CR_H * CR_H) / CR_THREADS; ++it) {
    const int idx = it * CR_THREADS + tid;
    const int j = idx >> 6, k = idx & 63;
    W2s[k * (CR_H + 1) + j] = P.w2[j * CR_H + k];
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int64_t rowc = row0 + CR_BM * c;           // first row of this 128-environment block
    {
      const int lr = wave * 32 + r32;                // this lane's environment (column of D^T)
      const int64_t gr = rowc + lr;
      const float tm = (gr < M) ? time_rows[gr / rows_per_time] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = mfma_row(r, lane);             // row of D^T = hidden unit
        float v0 = acc0[c][r] + tm * P.w1[(int64_t)j * ldw + N] + P.b1[j];
        float v1 = acc1[c][r] + tm * P.w1[(int64_t)(j + 32) * ldw + N] + P.b1[j + 32];
        Hs[j * (CR_BM + 1) + lr] = v0 > 0.0f ? v0 : 0.0f;
        Hs[(j + 32) * (CR_BM + 1) + lr] = v1 > 0.0f ? v1 : 0.0f;
      }
    }
    __syncthreads();
    f32x16 c0 = {0}, c1 = {0};
#pragma unroll
    for (int kk = 0; kk < CR_H; kk += 2) {
      const int k = kk + (lane >> 5);
      const float a = Hs[k * (CR_BM + 1) + wave * 32 + (lane & 31)];
      const float b0 = W2s[k * (CR_H + 1) + (lane & 31)];
      const float b1 = W2s[k * (CR_H + 1) + 32 + (lane & 31)];
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, c1, 0, 0, 0);
    }
    __syncthreads();
    {
      const int j0 = lane & 31;
      const float bb0 = P.b2[j0], bb1 = P.b2[j0 + 32];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = wave * 32 + mfma_row(r, lane);
        float v0 = c0[r] + bb0, v1 = c1[r] + bb1;
        v0 = v0 > 0.0f ? v0 : 0.0f;
        v1 = v1 > 0.0f ? v1 : 0.0f;
        Hs[j0 * (CR_BM + 1) + lr] = v0;
        Hs[(j0 + 32) * (CR_BM + 1) + lr] = v1;
      }
    }
    __syncthreads();
    if (tid < CR_BM) {
      const int64_t gr = rowc + tid;
      if (gr < M) {
        float sacc = 0.0f;
#pragma unroll 8
        for (int j = 0; j < CR_H; ++j) sacc += Hs[j * (CR_BM + 1) + tid] * P.w3[j];
        value[gr] = sacc + P.b3[0];
      }
    }
    __syncthreads();      // the next block's activations overwrite Hs
  }
}

// ---- backward (minibatch-sized M) ------------------------------------------------------------------------------------
// stage 1: one workgroup per row m: dh2, dh1 (masked by the ReLUs) -> scratch; stage 2: weight gradients as plain
// reductions over m (deterministic order), dW1 over a (j, k) grid with k-coalesced reads of the counts.
__global__ __launch_bounds__(CR_H) void k_critic_bwd_rows(int64_t M, const float* __restrict__ dv,
                                                          const float* __restrict__ h1, const float* __restrict__ h2,
                                                          CriticParams P, float* __restrict__ dh1,
                                                          float* __restrict__ dh2) {
  __shared__ float s_dh2[CR_H];
  const int64_t m = blockIdx.x;
  const int j = threadIdx.x;
  const float g = dv[m];
  const float d2 = (h2[m * CR_H + j] > 0.0f) ? g * P.w3[j] : 0.0f;
  s_dh2[j] = d2;
  dh2[m * CR_H + j] = d2;
  __syncthreads();
  float s = 0.0f;
  for (int i = 0; i < CR_H; ++i) s += s_dh2[i] * P.w2[i * CR_H + j];  // dh1[j] = sum_i dh2[i] * w2[i][j]
  dh1[m * CR_H + j] = (h1[m * CR_H + j] > 0.0f) ? s : 0.0f;
}

__global__ __launch_bounds__(CR_THREADS) void k_critic_bwd_small(int64_t M, const float* __restrict__ dv,
                                                                 const float* __restrict__ h1,
                                                                 const float* __restrict__ h2,
                                                                 const float* __restrict__ dh1,
                                                                 const float* __restrict__ dh2,
                                                                 const float* __restrict__ time_rows,
                                                                 int64_t rows_per_time, int64_t N,
                                                                 float* __restrict__ gw1, float* __restrict__ gb1,
                                                                 float* __restrict__ gw2, float* __restrict__ gb2,
                                                                 float* __restrict__ gw3, float* __restrict__ gb3) {
  // grid: 64*64 threads for gw2 (+ the small vectors handled by the first threads)
  const int idx = blockIdx.x * CR_THREADS + threadIdx.x;
  if (idx < CR_H * CR_H) {
    const int j = idx >> 6, k = idx & 63;
    float s = 0.0f;
    for (int64_t m = 0; m < M; ++m) s += dh2[m * CR_H + j] * h1[m * CR_H + k];
    gw2[idx] += s;
  }
  if (idx < CR_H) {
    float sb2 = 0.0f, sb1 = 0.0f, sw3 = 0.0f, swt = 0.0f;
    for (int64_t m = 0; m < M; ++m) {
      sb2 += dh2[m * CR_H + idx];
      sb1 += dh1[m * CR_H + idx];
      sw3 += dv[m] * h2[m * CR_H + idx];
      swt += dh1[m * CR_H + idx] * time_rows[m / rows_per_time];
    }
    gb2[idx] += sb2;
    gb1[idx] += sb1;
    gw3[idx] += sw3;
    gw1[(int64_t)idx * (N + 1) + N] += swt;  // the time column of W1
  }
  if (idx == 0) {
    float s = 0.0f;
    for (int64_t m = 0; m < M; ++m) s += dv[m];
    gb3[0] += s;
  }
}

__global__ __launch_bounds__(CR_THREADS) void k_critic_bwd_w1(int64_t M, int64_t N, const float* __restrict__ counts,
                                                              int64_t ldc, const float* __restrict__ dh1,
                                                              float* __restrict__ gw1) {
  const int64_t k = (int64_t)blockIdx.x * CR_THREADS + threadIdx.x;
  const int j = blockIdx.y;
  if (k >= N) return;
  float s = 0.0f;
  for (int64_t m = 0; m < M; ++m) s += dh1[m * CR_H + j] * counts[m * ldc + k];
  gw1[(int64_t)j * (N + 1) + k] += s;
}

// ---- backward for MANY rows (an optimiser minibatch of thousands of frames) ------------------------------------------------
// The kernels above give one thread a whole column of the reduction over m: right for the reference's sub-batch of 32 rows,
// a serial chain of M loads per thread beyond a few hundred (M = 4 096, N = 2 500: 3.7 ms, bench.py's update_path). Here the
// reduction over m is cut into row chunks that run in parallel, each leaving partial sums, and a second launch adds the
// partials of a weight IN CHUNK ORDER: deterministic (no atomics), the same sums in another association.
#define CB_RC 128      // rows per chunk of the dW1 pass (dh1 tile: 32 KB of LDS)
#define CB_RS 32       // rows per chunk of the small-matrix pass (four 8 KB tiles)
#define CB_SMALL (CR_H * CR_H + 4 * CR_H + 1)     // gw2 | gb2 | gb1 | gw3 | gw1's time column | gb3

// dW1 partial: workgroup (kt, s) = 64 input columns x all 64 hidden units x rows [128 s, 128 s + 128)
__global__ __launch_bounds__(CR_THREADS) void k_critic_bwd_w1_part(int64_t M, int64_t N, const float* __restrict__ counts,
                                                                   int64_t ldc, const float* __restrict__ dh1,
                                                                   float* __restrict__ part) {
  __shared__ float4 s_dh[CB_RC * (CR_H / 4)];     // [row][hidden unit]
  const int tid = threadIdx.x;
  const int64_t m0 = (int64_t)blockIdx.y * CB_RC;
  const int rows = (int)((M - m0) < CB_RC ? (M - m0) : CB_RC);
  const float4* src = reinterpret_cast<const float4*>(dh1 + m0 * CR_H);
  for (int e = tid; e < rows * (CR_H / 4); e += CR_THREADS) s_dh[e] = src[e];
  __syncthreads();
  const int kk = tid & 63, jg = tid >> 6;         // a wave = 64 columns of one group of 16 hidden units
  const int64_t k = (int64_t)blockIdx.x * 64 + kk;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  if (k < N) {
    const float* col = counts + m0 * ldc + k;
#pragma unroll 4
    for (int m = 0; m < rows; ++m) {
      const float x = col[(int64_t)m * ldc];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 d = s_dh[m * (CR_H / 4) + jg * 4 + q];      // wave-uniform address: broadcast
        acc[4 * q + 0] = fmaf(d.x, x, acc[4 * q + 0]);
        acc[4 * q + 1] = fmaf(d.y, x, acc[4 * q + 1]);
        acc[4 * q + 2] = fmaf(d.z, x, acc[4 * q + 2]);
        acc[4 * q + 3] = fmaf(d.w, x, acc[4 * q + 3]);
      }
    }
    float* out = part + ((int64_t)blockIdx.y * CR_H + jg * 16) * N + k;
#pragma unroll
    for (int i = 0; i < 16; ++i) out[(int64_t)i * N] = acc[i];
  }
}

__global__ __launch_bounds__(CR_THREADS) void k_critic_bwd_w1_reduce(int64_t S, int64_t N, const float* __restrict__ part,
                                                                     float* __restrict__ gw1) {
  const int64_t idx = (int64_t)blockIdx.x * CR_THREADS + threadIdx.x;     // (j, k) flat over [64][N]
  if (idx >= CR_H * N) return;
  const int64_t j = idx / N, k = idx - j * N;
  float s = 0.0f;
  for (int64_t c = 0; c < S; ++c) s += part[c * CR_H * N + idx];
  gw1[j * (N + 1) + k] += s;
}

// the small gradients' partials: workgroup s = rows [32 s, 32 s + 32)
__global__ __launch_bounds__(CR_THREADS) void k_critic_bwd_small_part(int64_t M, const float* __restrict__ dv,
                                                                      const float* __restrict__ h1,
                                                                      const float* __restrict__ h2,
                                                                      const float* __restrict__ dh1,
                                                                      const float* __restrict__ dh2,
                                                                      const float* __restrict__ time_rows,
                                                                      int64_t rows_per_time, float* __restrict__ part) {
  __shared__ float s_dh2[CB_RS * CR_H], s_h1[CB_RS * CR_H], s_dh1[CB_RS * CR_H], s_h2[CB_RS * CR_H];
  __shared__ float s_dv[CB_RS], s_tm[CB_RS];
  const int tid = threadIdx.x;
  const int64_t m0 = (int64_t)blockIdx.x * CB_RS;
  const int rows = (int)((M - m0) < CB_RS ? (M - m0) : CB_RS);
  for (int e = tid; e < CB_RS * CR_H; e += CR_THREADS) {
    const bool in = e < rows * CR_H;
    s_dh2[e] = in ? dh2[m0 * CR_H + e] : 0.0f;
    s_h1[e] = in ? h1[m0 * CR_H + e] : 0.0f;
    s_dh1[e] = in ? dh1[m0 * CR_H + e] : 0.0f;
    s_h2[e] = in ? h2[m0 * CR_H + e] : 0.0f;
  }
  if (tid < CB_RS) {
    s_dv[tid] = tid < rows ? dv[m0 + tid] : 0.0f;
    s_tm[tid] = tid < rows ? time_rows[(m0 + tid) / rows_per_time] : 0.0f;
  }
  __syncthreads();
  float* out = part + (int64_t)blockIdx.x * CB_SMALL;
  {
    const int j = tid >> 2, k0 = (tid & 3) * 16;       // gw2[j][k0 .. k0 + 15]
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int m = 0; m < CB_RS; ++m) {
      const float a = s_dh2[m * CR_H + j];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaf(a, s_h1[m * CR_H + k0 + i], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) out[j * CR_H + k0 + i] = acc[i];
  }
  if (tid < CR_H) {
    float sb2 = 0.0f, sb1 = 0.0f, sw3 = 0.0f, swt = 0.0f;
    for (int m = 0; m < CB_RS; ++m) {
      sb2 += s_dh2[m * CR_H + tid];
      sb1 += s_dh1[m * CR_H + tid];
      sw3 = fmaf(s_dv[m], s_h2[m * CR_H + tid], sw3);
      swt = fmaf(s_dh1[m * CR_H + tid], s_tm[m], swt);
    }
    out[CR_H * CR_H + tid] = sb2;
    out[CR_H * CR_H + CR_H + tid] = sb1;
    out[CR_H * CR_H + 2 * CR_H + tid] = sw3;
    out[CR_H * CR_H + 3 * CR_H + tid] = swt;
  }
  if (tid == 0) {
    float sd = 0.0f;
    for (int m = 0; m < CB_RS; ++m) sd += s_dv[m];
    out[CB_SMALL - 1] = sd;
  }
}

__global__ __launch_bounds__(CR_THREADS) void k_critic_bwd_small_reduce(int64_t S, int64_t N, const float* __restrict__ part,
                                                                        float* __restrict__ gw1, float* __restrict__ gb1,
                                                                        float* __restrict__ gw2, float* __restrict__ gb2,
                                                                        float* __restrict__ gw3, float* __restrict__ gb3) {
  const int idx = blockIdx.x * CR_THREADS + threadIdx.x;
  if (idx >= CB_SMALL) return;
  float s = 0.0f;
  for (int64_t c = 0; c < S; ++c) s += part[c * CB_SMALL + idx];
  const int v = idx - CR_H * CR_H;
  if (v < 0)
    gw2[idx] += s;
  else if (v < CR_H)
    gb2[v] += s;
  else if (v < 2 * CR_H)
    gb1[v - CR_H] += s;
  else if (v < 3 * CR_H)
    gw3[v - 2 * CR_H] += s;
  else if (v < 4 * CR_H)
    gw1[(int64_t)(v - 3 * CR_H) * (N + 1) + N] += s;      // the time column of W1
  else
    gb3[0] += s;
}

#define CB_MANY_ROWS 512       // from this many rows on the chunked backward is used (needs tarl_critic_mlp_bwd_scratch_floats)

// ---- host side -------------------------------------------------------------------------------------------------------
template <typename XT>
static int critic_fwd(const XT* counts, int64_t ldc, int64_t rps, int64_t M, int64_t N, const float* time_rows,
                      int64_t rows_per_time, const float* w1, const float* b1, const float* w2, const float* b2,
                      const float* w3, const float* b3, float* value, float* h1_out, float* h2_out,
                      tarl_stream stream) {
  TARL_REQUIRE(counts && time_rows && w1 && b1 && w2 && b2 && w3 && b3 && value, "null argument");
  TARL_REQUIRE(M >= 1 && N >= 1 && rows_per_time >= 1, "bad sizes");
  TARL_REQUIRE(ceil_div(M, CR_BM) < ((int64_t)1 << 31), "too many rows");
  const CriticParams P{w1, b1, w2, b2, w3, b3};
  hipLaunchKernelGGL(k_critic_fwd<XT>, dim3((unsigned)ceil_div(M, CR_BM)), dim3(CR_THREADS), 0, (hipStream_t)stream,
                     counts, ldc, rps, M, N, time_rows, rows_per_time, P, value, h1_out, h2_out);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_critic_mlp_fwd(const float* counts, int64_t ldc, int64_t M, int64_t N, const float* time_rows,
                                   int64_t rows_per_time, const float* w1, const float* b1, const float* w2,
                                   const float* b2, const float* w3, const float* b3, float* value, float* h1_out,
                                   float* h2_out, tarl_stream stream) {
  TARL_REQUIRE(ldc >= N, "row stride smaller than N");
  return critic_fwd(counts, ldc, 0, M, N, time_rows, rows_per_time, w1, b1, w2, b2, w3, b3, value, h1_out, h2_out,
                    stream);
}

extern "C" int tarl_critic_mlp_fwd_u8(const uint8_t* counts, int64_t ldc, int64_t M, int64_t N, const float* time_rows,
                                      int64_t rows_per_time, const float* w1, const float* b1, const float* w2,
                                      const float* b2, const float* w3, const float* b3, float* value, float* h1_out,
                                      float* h2_out, tarl_stream stream) {
  TARL_REQUIRE(ldc >= N, "row stride smaller than N");
  return critic_fwd(counts, ldc, 0, M, N, time_rows, rows_per_time, w1, b1, w2, b2, w3, b3, value, h1_out, h2_out,
                    stream);
}

// ---- few rows (the optimiser minibatch): split-K ----------------------------------------------------------------------------
// One 128-row MFMA tile walks all N input columns alone (N / 32 dependent stage + MFMA rounds on one CU). For M <= a few
// hundred rows the first layer is spread over the input columns instead: workgroup (s, rb) multiplies 32 rows by the
// 64 x 64 block of W1 that belongs to columns [64 s, 64 s + 64) on the vector ALU and parks its partial sums; one
// workgroup then adds the partials in chunk order (fixed order: deterministic), applies the time column, bias and ReLU and
// finishes the two small layers. The value differs from k_critic_fwd's only by the order of the fp32 additions.
#define CK_KC 64
#define CK_RB 32

__global__ __launch_bounds__(CR_THREADS) void k_critic_splitk_partial(const float* __restrict__ counts, int64_t ldc,
                                                                      int64_t M, int64_t N,
                                                                      const float* __restrict__ w1,
                                                                      float* __restrict__ partial, int64_t Mpad) {
  __shared__ float Xs[CK_RB * (CK_KC + 1)];     // [row][k]
  __shared__ float Ws[CK_KC * (CR_H + 1)];      // [k][j]
  const int tid = threadIdx.x;
  const int64_t k0 = (int64_t)blockIdx.x * CK_KC, row0 = (int64_t)blockIdx.y * CK_RB, ldw = N + 1;
#pragma unroll
  for (int it = 0; it < (CK_RB * CK_KC) / CR_THREADS; ++it) {
    const int idx = it * CR_THREADS + tid;
    const int r = idx >> 6, k = idx & 63;
    const int64_t gr = row0 + r, gk = k0 + k;
    Xs[r * (CK_KC + 1) + k] = (gr < M && gk < N) ? counts[gr * ldc + gk] : 0.0f;
  }
#pragma unroll
  for (int it = 0; it < (CR_H * CK_KC) / CR_THREADS; ++it) {
    const int idx = it * CR_THREADS + tid;
    const int j = idx >> 6, k = idx & 63;
    const int64_t gk = k0 + k;
    Ws[k * (CR_H + 1) + j] = (gk < N) ? w1[(int64_t)j * ldw + gk] : 0.0f;
  }
  __syncthreads();
  const int r = tid >> 3, jg = tid & 7;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 8
  for (int k = 0; k < CK_KC; ++k) {
    const float x = Xs[r * (CK_KC + 1) + k];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(x, Ws[k * (CR_H + 1) + jg + 8 * i], acc[i]);
  }
  float* out = partial + ((int64_t)blockIdx.x * Mpad + row0 + r) * CR_H;
#pragma unroll
  for (int i = 0; i < 8; ++i) out[jg + 8 * i] = acc[i];
}

__global__ __launch_bounds__(CR_THREADS) void k_critic_splitk_finish(const float* __restrict__ partial, int64_t S,
                                                                     int64_t Mpad, int64_t M, int64_t N,
                                                                     const float* __restrict__ time_rows,
                                                                     int64_t rows_per_time, CriticParams P,
                                                                     float* __restrict__ value,
                                                                     float* __restrict__ h1_out,
                                                                     float* __restrict__ h2_out) {
  __shared__ float H1[CK_RB * (CR_H + 1)];
  __shared__ float H2[CK_RB * (CR_H + 1)];
  __shared__ float W2s[CR_H * (CR_H + 1)];      // [k][j] = w2[j][k]
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * CK_RB, ldw = N + 1;
#pragma unroll
  for (int it = 0; it < (CR_H * CR_H) / CR_THREADS; ++it) {
    const int idx = it * CR_THREADS + tid;
    const int j = idx >> 6, k = idx & 63;
    W2s[k * (CR_H + 1) + j] = P.w2[j * CR_H + k];
  }
  const int r = tid >> 3, jg = tid & 7;
  const int64_t gr = row0 + r;
  const float tm = (gr < M) ? time_rows[gr / rows_per_time] : 0.0f;
  {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t s = 0; s < S; ++s) {
      const float* in = partial + (s * Mpad + gr) * CR_H;
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += in[jg + 8 * i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = jg + 8 * i;
      float v = acc[i] + tm * P.w1[(int64_t)j * ldw + N] + P.b1[j];
      v = v > 0.0f ? v : 0.0f;
      H1[r * (CR_H + 1) + j] = v;
      if (h1_out && gr < M) h1_out[gr * CR_H + j] = v;
    }
  }
  __syncthreads();
  {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 8
    for (int k = 0; k < CR_H; ++k) {
      const float x = H1[r * (CR_H + 1) + k];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fmaf(x, W2s[k * (CR_H + 1) + jg + 8 * i], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = jg + 8 * i;
      float v = acc[i] + P.b2[j];
      v = v > 0.0f ? v : 0.0f;
      H2[r * (CR_H + 1) + j] = v;
      if (h2_out && gr < M) h2_out[gr * CR_H + j] = v;
    }
  }
  __syncthreads();
  if (tid < CK_RB && row0 + tid < M) {
    float s = 0.0f;
#pragma unroll 8
    for (int j = 0; j < CR_H; ++j) s += H2[tid * (CR_H + 1) + j] * P.w3[j];
    value[row0 + tid] = s + P.b3[0];
  }
}

extern "C" int64_t tarl_critic_splitk_scratch_floats(int64_t M, int64_t N) {
  if (M < 1 || N < 1) return -1;
  return ceil_div(N, CK_KC) * ceil_div(M, CK_RB) * CK_RB * CR_H;
}

extern "C" int tarl_critic_mlp_fwd_splitk(const float* counts, int64_t ldc, int64_t M, int64_t N,
                                          const float* time_rows, int64_t rows_per_time, const float* w1,
                                          const float* b1, const float* w2, const float* b2, const float* w3,
                                          const float* b3, float* scratch, float* value, float* h1_out, float* h2_out,
                                          tarl_stream stream) {
  TARL_REQUIRE(counts && time_rows && w1 && b1 && w2 && b2 && w3 && b3 && value && scratch, "null argument");
  TARL_REQUIRE(M >= 1 && N >= 1 && rows_per_time >= 1 && ldc >= N, "bad sizes");
  const int64_t S = ceil_div(N, CK_KC), RB = ceil_div(M, CK_RB);
  TARL_REQUIRE(S < ((int64_t)1 << 31) && RB < 65536, "split-K critic: too many rows (use tarl_critic_mlp_fwd)");
  const CriticParams P{w1, b1, w2, b2, w3, b3};
  hipLaunchKernelGGL(k_critic_splitk_partial, dim3((unsigned)S, (unsigned)RB), dim3(CR_THREADS), 0, (hipStream_t)stream,
                     counts, ldc, M, N, w1, scratch, RB * CK_RB);
  TARL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_critic_splitk_finish, dim3((unsigned)RB), dim3(CR_THREADS), 0, (hipStream_t)stream, scratch, S,
                     RB * CK_RB, M, N, time_rows, rows_per_time, P, value, h1_out, h2_out);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

template <typename XT>
static int critic_fwd_slabs(const XT* counts, int64_t rows_per_slab, int64_t M, int64_t N, const float* time_rows,
                            int64_t rows_per_time, const float* w1, const float* b1, const float* w2, const float* b2,
                            const float* w3, const float* b3, float* value, tarl_stream stream) {
  TARL_REQUIRE(rows_per_slab >= CR_BM && rows_per_slab % CR_BM == 0, "rows_per_slab must be a multiple of 128");
  TARL_REQUIRE(M % rows_per_slab == 0, "M must be a whole number of slabs");
  TARL_REQUIRE(counts && time_rows && w1 && b1 && w2 && b2 && w3 && b3 && value, "null argument");
  TARL_REQUIRE(M >= 1 && N >= 1 && rows_per_time >= 1, "bad sizes");
  TARL_REQUIRE(((uintptr_t)counts) % 16 == 0, "counts must be 16-byte aligned");
  const CriticParams P{w1, b1, w2, b2, w3, b3};
  hipLaunchKernelGGL(k_critic_fwd_slab<XT>, dim3((unsigned)(M / CR_BM)), dim3(CR_THREADS), 0, (hipStream_t)stream, counts,
                     rows_per_slab, M, N, time_rows, rows_per_time, P, value);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_critic_mlp_fwd_slabs(const float* counts, int64_t rows_per_slab, int64_t M, int64_t N,
                                         const float* time_rows, int64_t rows_per_time, const float* w1,
                                         const float* b1, const float* w2, const float* b2, const float* w3,
                                         const float* b3, float* value, tarl_stream stream) {
  return critic_fwd_slabs(counts, rows_per_slab, M, N, time_rows, rows_per_time, w1, b1, w2, b2, w3, b3, value, stream);
}

extern "C" int64_t tarl_critic_split_scratch_bytes(int64_t N) {
  return N >= 1 ? 3 * CR_H * ceil_div(N, CR_BK) * CR_BK * (int64_t)sizeof(uint16_t) : -1;
}

extern "C" int tarl_critic_mlp_fwd_slabs_u8(const uint8_t* counts, int64_t rows_per_slab, int64_t M, int64_t N,
                                            const float* time_rows, int64_t rows_per_time, const float* w1,
                                            const float* b1, const float* w2, const float* b2, const float* w3,
                                            const float* b3, void* split_scratch, float* value, tarl_stream stream) {
  if (!split_scratch)     // fp32 MFMA on the widened bytes
    return critic_fwd_slabs(counts, rows_per_slab, M, N, time_rows, rows_per_time, w1, b1, w2, b2, w3, b3, value, stream);
  TARL_REQUIRE(rows_per_slab >= CR_BM && rows_per_slab % CR_BM == 0, "rows_per_slab must be a multiple of 128");
  TARL_REQUIRE(M % rows_per_slab == 0, "M must be a whole number of slabs");
  TARL_REQUIRE(counts && time_rows && w1 && b1 && w2 && b2 && w3 && b3 && value, "null argument");
  TARL_REQUIRE(M >= 1 && N >= 1 && rows_per_time >= 1, "bad sizes");
  TARL_REQUIRE(((uintptr_t)counts) % 16 == 0 && ((uintptr_t)split_scratch) % 16 == 0, "counts / scratch must be 16-byte aligned");
  const int64_t Kpad = ceil_div(N, CR_BK) * CR_BK;
  const CriticParams P{w1, b1, w2, b2, w3, b3};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_split_w1, dim3((unsigned)ceil_div(CR_H * Kpad, CR_THREADS)), dim3(CR_THREADS), 0, s, w1, N, Kpad,
                     (uint16_t*)split_scratch);
  TARL_LAUNCH_CHECK();
  // environments per workgroup: 256 when the slab divides and the launch still fills the chip several times over, else 128.
  // Measured at 16 384 environments x 257 frames x 2 500 nodes (10.5 GB of count bytes; tools/time_phases.py, whole GAE pass):
  // 5.76 ms with 128, 4.82 ms with 256, 5.22 ms with 512 (368 registers: one workgroup per CU left).
  // (TARL_CRITIC_CT = 1 | 2 | 4: test override, read per call on purpose — tests/test_gpu_fused.py compares the tile widths
  // inside one process; one getenv per GAE pass, i.e. per ~5 ms kernel, is not a cost)
  const char* ct_knob = getenv("TARL_CRITIC_CT");
  const int ct_env = ct_knob ? atoi(ct_knob) : 0;
  int ct = 1;
  if (rows_per_slab % (2 * CR_BM) == 0 && M / (2 * CR_BM) >= 1024) ct = 2;
  if ((ct_env == 1 || ct_env == 2 || ct_env == 4) && rows_per_slab % (ct_env * CR_BM) == 0) ct = ct_env;
  const dim3 grid((unsigned)(M / (CR_BM * ct)));
  if (ct == 4)
    hipLaunchKernelGGL(k_critic_fwd_slab_u8x3<4>, grid, dim3(CR_THREADS), 0, s, counts, rows_per_slab, M, N, Kpad, time_rows,
                       rows_per_time, (const uint16_t*)split_scratch, P, value);
  else if (ct == 2)
    hipLaunchKernelGGL(k_critic_fwd_slab_u8x3<2>, grid, dim3(CR_THREADS), 0, s, counts, rows_per_slab, M, N, Kpad, time_rows,
                       rows_per_time, (const uint16_t*)split_scratch, P, value);
  else
    hipLaunchKernelGGL(k_critic_fwd_slab_u8x3<1>, grid, dim3(CR_THREADS), 0, s, counts, rows_per_slab, M, N, Kpad, time_rows,
                       rows_per_time, (const uint16_t*)split_scratch, P, value);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int64_t tarl_critic_mlp_bwd_scratch_floats(int64_t M, int64_t N) {
  if (M < 1 || N < 1) return -1;
  int64_t n = 2 * M * CR_H;                                   // dh1, dh2
  if (M >= CB_MANY_ROWS) n += ceil_div(M, CB_RC) * CR_H * N + ceil_div(M, CB_RS) * CB_SMALL;     // the chunks' partial sums
  return n;
}

extern "C" int tarl_critic_mlp_bwd(const float* counts, int64_t ldc, int64_t M, int64_t N, const float* time_rows,
                                   int64_t rows_per_time, const float* w1, const float* w2, const float* w3,
                                   const float* h1, const float* h2, const float* grad_value, float* scratch,
                                   float* gw1, float* gb1, float* gw2, float* gb2, float* gw3, float* gb3,
                                   tarl_stream stream) {
  TARL_REQUIRE(counts && time_rows && w1 && w2 && w3 && h1 && h2 && grad_value && scratch, "null argument");
  TARL_REQUIRE(gw1 && gb1 && gw2 && gb2 && gw3 && gb3, "null gradient buffer");
  TARL_REQUIRE(M >= 1 && N >= 1 && ldc >= N && rows_per_time >= 1, "bad sizes");
  TARL_REQUIRE(M < ((int64_t)1 << 31), "too many rows");
  const CriticParams P{w1, nullptr, w2, nullptr, w3, nullptr};
  float* dh1 = scratch;
  float* dh2 = scratch + M * CR_H;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_critic_bwd_rows, dim3((unsigned)M), dim3(CR_H), 0, s, M, grad_value, h1, h2, P, dh1, dh2);
  TARL_LAUNCH_CHECK();
  if (M >= CB_MANY_ROWS) {     // chunked reductions over the rows (the caller sized scratch with tarl_critic_mlp_bwd_scratch_floats)
    const int64_t S1 = ceil_div(M, CB_RC), S2 = ceil_div(M, CB_RS);
    TARL_REQUIRE(S1 < 65536, "too many rows for one launch");
    float* part_w1 = scratch + 2 * M * CR_H;
    float* part_small = part_w1 + S1 * CR_H * N;
    hipLaunchKernelGGL(k_critic_bwd_w1_part, dim3((unsigned)ceil_div(N, 64), (unsigned)S1), dim3(CR_THREADS), 0, s, M, N, counts,
                       ldc, dh1, part_w1);
    TARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_critic_bwd_w1_reduce, dim3((unsigned)ceil_div(CR_H * N, CR_THREADS)), dim3(CR_THREADS), 0, s, S1, N,
                       part_w1, gw1);
    TARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_critic_bwd_small_part, dim3((unsigned)S2), dim3(CR_THREADS), 0, s, M, grad_value, h1, h2, dh1, dh2,
                       time_rows, rows_per_time, part_small);
    TARL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_critic_bwd_small_reduce, dim3((unsigned)ceil_div(CB_SMALL, CR_THREADS)), dim3(CR_THREADS), 0, s, S2, N,
                       part_small, gw1, gb1, gw2, gb2, gw3, gb3);
    TARL_LAUNCH_CHECK();
    return TARL_OK;
  }
  hipLaunchKernelGGL(k_critic_bwd_small, dim3((CR_H * CR_H) / CR_THREADS), dim3(CR_THREADS), 0, s, M, grad_value, h1, h2,
                     dh1, dh2, time_rows, rows_per_time, N, gw1, gb1, gw2, gb2, gw3, gb3);
  TARL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_critic_bwd_w1, dim3((unsigned)ceil_div(N, CR_THREADS), CR_H), dim3(CR_THREADS), 0, s, M, N,
                     counts, ldc, dh1, gw1);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}
