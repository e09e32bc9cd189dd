// tarl_common.h — shared declarations for the libtarl_hip.so translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tarl_hip.h"

// ---- error plumbing (no exceptions cross the ABI) -------------------------------------------------------------------
void tarl_set_error(const char* fmt, ...);

#define TARL_CHECK_HIP(expr)                                                                   \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      tarl_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return TARL_ERR_HIP;                                                                     \
    }                                                                                          \
  } while (0)

#define TARL_REQUIRE(cond, msg)                                         \
  do {                                                                  \
    if (!(cond)) {                                                      \
      tarl_set_error("%s: requirement failed: %s", __func__, msg);      \
      return TARL_ERR_INVALID;                                          \
    }                                                                   \
  } while (0)

#define TARL_LAUNCH_CHECK()                                                              \
  do {                                                                                   \
    hipError_t _e = hipGetLastError();                                                   \
    if (_e != hipSuccess) {                                                              \
      tarl_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return TARL_ERR_HIP;                                                               \
    }                                                                                    \
  } while (0)

// ---- static plan ----------------------------------------------------------------------------------------------------
struct tarl_plan {
  int64_t N, E, G;         // nodes, edges, distinct source nodes ("groups" of GraphDistribution)
  int32_t max_in, max_out;
  int32_t src_sorted;      // 1 when edge_index[0] is non-decreasing and the plan order == original order
  int32_t dst_sorted;      // 1 when CSC order == original order
  int32_t siblings4;       // 1 when nodes 4c .. 4c + 3 have the same first four in-edge sources for every c (N % 4 == 0):
                           // the roads leaving one intersection; the Direction gather then reads the upstream rows once
  int32_t row_siblings;    // 1 when the row-chunk table below is worth using (see row_chunks)
  int64_t num_row_chunks;
  // device arrays (int32)
  int32_t* row_chunks;     // [num_row_chunks][8] = {row[4] (-1: none), out4[4]}: rows with the SAME ordered out-edge target list
                           // (on a road network: the roads that ENTER one intersection) grouped four at a time, so that the row
                           // pass gathers their downstream post words once per chunk instead of once per row
  int32_t* in_ptr;         // [N+1]  CSC by destination
  int32_t* in_src;         // [E]    source node of the k-th in-edge
  int32_t* in_eid;         // [E]    original edge id (ascending inside a destination)
  int32_t* out_ptr;        // [N+1]  CSR by source (the distribution's sorted order)
  int32_t* out_dst;        // [E]
  int32_t* out_eid;        // [E]    original edge id of sorted position k  (== GraphDistribution.index)
  int32_t* src;            // [E]    original order
  int32_t* dst;            // [E]
  int32_t* group_of_node;  // [N]    compact rank among nodes with >=1 out-edge, -1 otherwise
  int32_t* node_of_group;  // [G]
};

// ---- state layout (src/feature_helpers.py:38-54) ------------------------------------------------------------------
struct Layout {
  int Nmax;
  int64_t ldx;      // row stride (floats)
  int64_t bstride;  // environment stride (floats)
  __host__ __device__ int col_maxn() const { return 3 * Nmax + 0; }
  __host__ __device__ int col_n() const { return 3 * Nmax + 1; }
  __host__ __device__ int col_ff() const { return 3 * Nmax + 2; }
  __host__ __device__ int col_maxflow() const { return 3 * Nmax + 4; }
  __host__ __device__ int col_sel() const { return 3 * Nmax + 5; }
  __host__ __device__ int col_road() const { return 3 * Nmax + 6; }
  __host__ __device__ int F() const { return 3 * Nmax + 7; }
};

// agent_features columns (src/feature_helpers.py:59-71)
enum { AG_ORIGIN = 0, AG_DEST = 1, AG_DEP = 2, AG_ARR = 3, AG_ON_WAY = 7, AG_DONE = 8, AG_COLS = 9 };

#define TARL_CONGESTION_FILE 3.0f  // src/feature_helpers.py:54

// ---- Philox4x32-10 (counter-based; one 128-bit block per call) ------------------------------------------------------
#ifndef PHILOX_ROUNDS
#define PHILOX_ROUNDS 10
#endif
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < PHILOX_ROUNDS; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uniform in (0,1): never 0 so that -log(-log(u)) stays finite
__device__ __forceinline__ float u01_open(uint32_t r) { return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// one uniform for (stream id, index) — index is the flat element id, 4 consecutive elements share a Philox block
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t counter, uint64_t index) {
  uint32_t o[4];
  const uint64_t blk = index >> 2;
  philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)seed,
                (uint32_t)(seed >> 32), o);
  return u01_open(o[index & 3]);
}

// uniforms for a run of nearby indices: the 128-bit Philox block is recomputed only when the index leaves it
struct PhiloxRun {
  uint32_t o0, o1, o2, o3;  // named registers: a runtime-indexed array would live in scratch memory
  uint64_t blk;
  __device__ __forceinline__ PhiloxRun() : o0(0), o1(0), o2(0), o3(0), blk(~0ull) {}
  __device__ __forceinline__ float uniform(uint64_t seed, uint64_t counter, uint64_t index) {
    return u01_open(word(seed, counter, index));
  }
  // the raw 32-bit word behind uniform()
  __device__ __forceinline__ uint32_t word(uint64_t seed, uint64_t counter, uint64_t index) {
    const uint64_t b = index >> 2;
    if (b != blk) {
      blk = b;
      uint32_t o[4];
      philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)seed,
                    (uint32_t)(seed >> 32), o);
      o0 = o[0]; o1 = o[1]; o2 = o[2]; o3 = o[3];
    }
    const uint32_t w = (uint32_t)(index & 3);
    return w == 0 ? o0 : (w == 1 ? o1 : (w == 2 ? o2 : o3));
  }
};

// Gumbel(0,1) variate from a device-generated uniform in (0,1): -log(-log(u)) through the hardware log2 (v_log_f32).
// Used ONLY for noise this library draws itself (Philox path). Parity runs pass the reference's own noise in, already
// transformed by torch on the host, so nothing that is compared bit-for-bit goes through this approximation.
__device__ __forceinline__ float gumbel_from_u01(float u) {
  const float ln2 = 0.69314718055994531f;
  const float e = -ln2 * __log2f(u);       // Exp(1) variate, > 0 because u < 1
  return -ln2 * __log2f(e);
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
