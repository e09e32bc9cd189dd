// fused_common.h — definitions shared by the two rollout implementations: fused.hip (env-minor, four launches per frame)
// and rollout_env.hip (one workgroup per environment, LDS-resident records, all T frames in one launch).
#pragma once

#include <float.h>
#include <math.h>
#include <stdlib.h>

#include "tarl_common.h"

#define FB 256          // pack / export kernels
#define ENVB 1024       // one-workgroup-per-environment kernels
#define TILE 256        // environments per workgroup in the env-minor kernels (one per lane)
#define LOG_EPS_P 1e-8f
#define INS_CAP 2048    // LDS candidate list of the insert kernel (entries)
#define INSB 64         // insert kernel: one wave per environment (all resident at once, 3/4 of the wave slots left free)

struct FusedBufs {
  float4* rec0;         // [N][B] {head_id, head_dep, n, tail_id}
  float2* rec1;         // [N][B] {head_arr, pending-garbage n0 (or -1 when nothing is pending)}
  float2* postA;        // [N][B] {n', tail'}   state after the Direction update, gathered by the upstream rows
  float* postB;         // [N][B] chosen: the agent the Direction update enqueues (0: nobody); read by the row itself only
  const float4* st0;    // [N]    {maxn, ff, road_index, cong}
  float* slots;         // [N][B][lds] slot-interleaved FIFO store: slot s at floats 3s..3s+2 = {id, arrival, departure}
  int64_t lds;          // row stride of slots in floats (>= 3*Nmax, multiple of 16)
  float* sel;           // [N][B] SELECTED_ROAD
  long long* acc_lp;    // [S][B] log-prob of this frame's action, 2^-32 fixed point (order-independent => deterministic)
  float* acc_n;         // [S][B] sum of the per-node counts after the row pass (small integers: exact in any order)
                        // S = acc_slots banks spread the atomics of the N/chunk workgroups that serve one environment
  int32_t* a_origin;    // [B][A]
  int32_t* a_dest;      // [B][A]
  float* a_dep;         // [B][A]
  uint8_t* a_status;    // [B][A] 0 waiting, 1 on the way, 2 done
  const int32_t* a_order;  // [B][A] agent ids sorted by departure time (static), or NULL: scan all agents every frame
  int32_t* cur_lo;      // [B] first position of a_order that may still hold a waiting agent
  const float* a_dep_sorted;  // [B][A] departure times in a_order's order (sequential scan instead of a gather)
  int64_t acc_slots;    // accumulator banks: acc_lp / acc_n are [acc_slots][B]; a workgroup adds into bank (chunk % slots)
};

#define LP_FIX 4294967296.0  // 2^32

// rec1.y packs two small integers exactly in fp32: code = (g + 1) * 1024 + hoff, where g = count at the pending
// (unmaterialised) garbage write or -1 when nothing is pending, and hoff = physical slot of logical slot 0 (ring buffer).
__device__ __forceinline__ float r1_code(float g, int hoff) { return (g + 1.0f) * 1024.0f + (float)hoff; }
__device__ __forceinline__ int r1_hoff(float code) { return ((int)code) & 1023; }
__device__ __forceinline__ float r1_g(float code) { return (float)(((int)code) >> 10) - 1.0f; }
// physical slot of logical slot s
__device__ __forceinline__ int phys(int hoff, int s, int Nmax) {
  int p = hoff + s;
  return p >= Nmax ? p - Nmax : p;
}


// fused.hip
FusedBufs tarl_to_bufs(const tarl_fused* f);
int tarl_check_fused_core(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax);
hipEvent_t tarl_prof_mark(hipStream_t s, int tag);  // sim.hip: live timing of the message-passing kernels (tag 0/1/2)
