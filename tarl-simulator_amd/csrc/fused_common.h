// fused_common.h — definitions shared by the two rollout implementations: fused.hip (env-minor, launches per frame)
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
#define INS_CAP 192     // LDS candidate list of the insert kernels (entries; 3.9 KB with the stashed target words: every
                        // workgroup of a 16 384-environment launch — two environments each, half the list each — is resident
                        // at once; a longer backlog takes the ordered global-scratch path)
#define INSB 64         // insert kernel: one wave per environment (all resident at once, 3/4 of the wave slots left free)

// ---- packed per-(node, environment) words (layout "v6") ------------------------------------------------------------------
// Dense words, read by every row every frame (12 + 4 + 1 bytes) and rewritten by the rows where something moved (an
// idle row's words already hold what a refresh would store, with one exception: the departure of an EMPTY row's garbage
// head changes with the clock — it is not stored while the row idles (tl without TLF_AUTH, n == 0): its readers derive it
// as last clock + tt0; rows with MAX_NUMBER_OF_AGENT <= 3, whose empty state the Direction tests can see, keep it eager):
//   hdp  uint2  {hd = head_id << 8 | n, bits of head_dep}   n = NUMBER_OF_AGENT (<= 255: the fused path requires Nmax <= 255),
//                                                            ids < 2^24 (the reference keeps them in fp32: exact below 2^24)
//   tl   u32    tail_id << 8 | hoff << 1 | TLF_AUTH          hoff = physical slot of logical slot 0 (ring buffer; < 128: the
//                                                            fused path requires Nmax <= 127). TLF_AUTH: gc8 (the
//                                                            pending-garbage count) is authoritative for the last frame;
//                                                            clear = the row was idle in the last frame: the pending
//                                                            garbage count is n itself and an empty row's head arrival is
//                                                            the last frame's clock
//   post u32    tail' << 8 | PF_TLAUTH | PF_NONEMPTY | PF_ARRIVED       the row's state after the Direction update: tail' = the agent
//                                                            it enqueues (PF_ARRIVED) or its old tail; written by the
//                                                            Direction gather, gathered by the upstream rows' Response test
//   sel8 u8     SELECTED_ROAD as the rank of the chosen out-edge in the node's CSR list (| SEL_CARRIED when the node drew
//               nothing in this frame and keeps its previous value); SEL_RAW: the fp32 value in `sel` is authoritative
// Event-only byte, written by the rows that move something in a frame:
//   gc8  u8     g + 1                                        g = count at the pending (unmaterialised) garbage write or -1;
//                                                            authoritative while tl carries TLF_AUTH. WRITE-ONLY in the
//                                                            frame kernels. (Layout v11: the head's ARRIVAL time, which
//                                                            the v10 event word rec1 carried beside it, is not stored
//                                                            any more: it is the arrival field of the head's slot record,
//                                                            or follows from the pending garbage / clean-row rules —
//                                                            head_arrival() below; an 8-byte scattered store per event
//                                                            row and frame became one byte.)
// Count byte of hdp.x: NUMBER_OF_AGENT in bits 0..6 (the fused path requires Nmax <= 127) and HD_DIRTY in bit 7.
// A CLEAN row (bit clear) obeys the dead-slot invariant of the reference's own bookkeeping: every logical slot above the
// count is ZERO (the slot at the count itself: the pending garbage triple, or zero when none is pending) — a pop duplicates
// the last slot (zero) into the slot that falls off the front, a withdraw zero-fills, the Direction update and the insert
// only ever write at the count. For a clean row the frame kernels therefore do not maintain the dead slots physically (no
// read of the last slot + store of the vacated one per pop, no zero-fill stores per withdraw: two of the five scattered
// slot accesses of a moving agent) and tarl_fused_export writes zeros for them. A row becomes DIRTY when pack finds a
// non-zero dead slot in x or when its FIFO reaches its last slot (count >= Nmax - 1: the last slot then holds a value a
// later pop has to duplicate); from then on it is maintained exactly, slot by slot, as before.
#define HD_DIRTY 0x80u
#define HD_CNT 0x7Fu
#define TLF_AUTH 1u
#define PF_ARRIVED 1u
#define PF_NONEMPTY 2u
#define PF_TLAUTH 4u      // the row's tail word carried TLF_AUTH when the Direction gather last wrote this post word (the row
                          // pass clears both together): with PF_NONEMPTY / PF_ARRIVED clear too, the row pass needs none of
                          // the row's other words
#define SEL_RAW 0x7Fu
#define SEL_CARRIED 0x80u
#define INRANK_NONE 0xFEu
// device status word (tarl_fused.flags): sticky bits, read by the host at its next synchronisation point
#define FLAG_COUNT_AT_NMAX 1      // a FIFO count reached Nmax: outside the reference's defined domain (it raises IndexError)
#define FLAG_AMBIGUOUS_EDGES 2    // two out-edges of one node lead to the same ROAD_INDEX: SELECTED_ROAD has no unique rank
#define FLAG_PACK_RANGE 4         // pack: a count above 255 / an agent id at or above 2^24
#define FLAG_CHOICE_OVERFLOW 8    // more than 65536 nodes drew nothing in one block of frames (degenerate policy tables)

// Static per-graph records (built by pack; shared by all environments, read through the scalar cache). One node record
// and ONE base address per row give the hot kernels everything static they need: a row's in-edge records and out-edge
// targets are contiguous (CSC / CSR), so four of them are fetched from consecutive addresses without per-edge index
// chains (the arrays are padded by four entries: a row with fewer edges reads its successors' and ignores them).
struct __attribute__((aligned(4))) InRec {
  int32_t src;      // upstream row
  int32_t rank;     // sel8 rank of src that heads for this row (INRANK_NONE: none)
  float ea;         // turn probability edge_attr[eid]
  float max_src;    // MAX_NUMBER_OF_AGENT of src
  int32_t eid;      // the edge's id in the caller's edge order (index of log_edge_attr / gumbel / delta_travel_time)
};
// 144 bytes per node. The first four out-edge targets and in-edge records sit IN the node record: one address (a function
// of the node id alone) reaches them, where out_pad[out0 + q] / in_rec[in0 + q] need the record first — one dependent
// scalar round trip less at the head of every wave of the row pass and of the Direction gather.
struct __attribute__((aligned(16))) NodeRec {
  int32_t in0, in_deg;      // CSC range of the row's in-edges
  int32_t out0, out_deg;    // CSR range of its out-edges
  float maxn, ff, road, cong;   // = st0
  float tt0;                    // travel time assigned at count 0: an empty row's garbage head departs at t + tt0
  int32_t pad_[3];
  int32_t out4[4];          // out_pad[out0 .. out0 + 3] (beyond out_deg: the node itself, ignored by the readers)
  InRec in4[4];             // in_rec[in0 .. in0 + 3]   (beyond in_deg: {0, INRANK_NONE, 0, 0})
};
#define NODE_REC_WORDS 36
#define IN_REC_WORDS 5
static_assert(sizeof(NodeRec) == NODE_REC_WORDS * 4 && sizeof(InRec) == IN_REC_WORDS * 4, "record sizes are part of the buffer contract (tarl_hip/ops.py: FusedState)");

struct FusedBufs {
  uint2* hdp;           // [N][B]
  uint32_t* tl;         // [N][B]
  uint8_t* gc8;         // [N][B] pending-garbage code (g + 1)
  uint32_t* post;       // [N][B]
  const float4* st0;    // [N]    {maxn, ff, road_index, cong}
  float* slots;         // [N][B][lds] slot-interleaved FIFO store: slot s at floats 3s..3s+2 = {id, arrival, departure}
  int64_t lds;          // row stride of slots in floats (>= 3*Nmax, multiple of 16)
  uint8_t* sel8;        // [N][B] SELECTED_ROAD code
  float* sel;           // [N][B] SELECTED_ROAD as fp32 (authoritative where sel8 == SEL_RAW; refreshed by export)
  const NodeRec* nodes;    // [N]
  const InRec* in_rec;     // [E + 4] CSC order
  const int32_t* out_pad;  // [E + 4] CSR order: target row of every out-edge (= plan out_dst, padded)
  long long* acc_lp;    // [S][B] log-prob of this frame's action, 2^-32 fixed point (order-independent => deterministic)
  float* acc_n;         // [S][B] sum of the per-node counts after the row pass (small integers: exact in any order)
  float* acc_w;         // [S][B] agents withdrawn (arrived at their destination) in this frame
                        // S = acc_slots banks spread the atomics of the N/chunk workgroups that serve one environment
  int32_t* a_origin;    // [B][A]
  int32_t* a_dest;      // [B][A]
  float* a_dep;         // [B][A]
  uint8_t* a_status;    // [B][A] 0 waiting, 1 on the way, 2 done
  const int32_t* a_order;  // [B][A] agent ids sorted by departure time (static), or NULL: scan all agents every frame
  int32_t* cur_lo;      // [B] first position of a_order that may still hold a waiting agent
  const float* a_dep_sorted;  // [B][A] departure times in a_order's order (sequential scan instead of a gather)
  // departure-ordered window of the insert kernel (built by pack when a_order is given): ONE 16-byte load per entry gives
  // {departure bits, origin, agent id, 0}; a_ins [B][A] (same order) = 1 once the entry's agent has been inserted (or was
  // not waiting at pack time); a_rank [B][A] = position of agent a in that order
  const uint4* a_win;
  uint8_t* a_ins;
  const int32_t* a_rank;
  int64_t acc_slots;    // accumulator banks: acc_* are [acc_slots][B]; a workgroup adds into bank (chunk % slots)
  int32_t* flags;       // [1] device status word
  int64_t env_base;     // global id of environment 0: the Philox streams are indexed by env_base + b (tarl_hip.h)
};

// per-frame side outputs (all optional)
struct FrameOut {
  uint8_t* counts8;     // [N][B] NUMBER_OF_AGENT after the frame (rollout buffers)
  float* countsf;       // [N][B] the same as fp32 (frame API)
  uint8_t* popped;      // [B][N] Response pop mask      (frame API, env-major like the unfused entry points)
  uint8_t* withdrawn;   // [B][N] withdraw mask
  uint8_t* events;      // [N][m_env] metric environments: bit 0 popped, bit 1 withdrawn
  float* dtt_node;      // [N][m_env] metric environments: delta_travel_time of the node's out-edges
  int32_t m_env;        // environments 0 .. m_env-1 keep the per-node series
  int32_t* leg;         // [B][2] {agents departed (inserted), agents arrived (withdrawn)} in this frame
  int32_t write_gc;     // store the event byte gc8 for every row that moves something. gc8 is read only BETWEEN calls (export,
                        // the LDS-resident rollout, the next call's first delta_travel_time) and is authoritative for the last
                        // frame alone: a rollout sets this in its last frame only, and the frames before store the byte just
                        // for the metric environments (whose delta_travel_time reads it every frame) — one scattered
                        // partial-line store into an array nothing else touches less per event row and frame
};

#define LP_FIX 4294967296.0  // 2^32

// ---- slot records of the FIFO store ------------------------------------------------------------------------------------
// slots[node][env][s] = one record of SLW floats {agent id, arrival, departure [, 0 ...]}. SLW = 3 (default): 12-byte
// triples, 192 B per row at Nmax = 15. SLW = 8 (developer build, -DTARL_SLW=8): a record is ONE aligned 32-byte sector, so
// the scattered store of an enqueue / insert fills its sector (no read-modify-write at the memory side: 13 ps per store
// against 43 ps for a 12-byte triple in isolation, profiles/r03_pmc_calibration.txt). Measured in round 4 at 16 384
// environments (same-box A/B, profiles/README.md): congested regime unchanged within the run-to-run noise, headline regime
// row pass 187 -> 236 us (the store grows from 7.9 to 19.7 GB: the scattered accesses of the event tail pay for the larger
// footprint, and the event tail is a latency chain, not a store-throughput problem). Not taken.
#ifndef TARL_SLW
#define TARL_SLW 3
#endif
#define SLW TARL_SLW
#define SLW_ALIGN (SLW == 8 ? 8 : 1)
__device__ __forceinline__ void slot_store(float* w, float id, float arr, float dep) {
#if TARL_SLW == 8
  reinterpret_cast<float4*>(w)[0] = make_float4(id, arr, dep, 0.0f);
  reinterpret_cast<float4*>(w)[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#else
  w[0] = id;
  w[1] = arr;
  w[2] = dep;
#endif
}
struct SlotRec {
  float id, arr, dep;
};
__device__ __forceinline__ SlotRec slot_load(const float* r) {
#if TARL_SLW == 8
  const float4 v = *reinterpret_cast<const float4*>(r);
  return SlotRec{v.x, v.y, v.z};
#else
  return SlotRec{r[0], r[1], r[2]};
#endif
}
__device__ __forceinline__ uint32_t r1_code(int g) { return (uint32_t)(g + 1); }
__device__ __forceinline__ int r1_g(uint32_t code) { return (int)code - 1; }
__device__ __forceinline__ int tl_hoff(uint32_t tlw) { return (int)((tlw >> 1) & 127u); }
__device__ __forceinline__ uint32_t tl_word(uint32_t tail_id, int hoff, uint32_t auth) {
  return (tail_id << 8) | ((uint32_t)hoff << 1) | auth;
}
// pending garbage count of a row from its dense words (+ gc8 when it is authoritative)
__device__ __forceinline__ int pending_g(uint32_t tlw, int n, uint32_t code, int Nmax) {
  return (tlw & TLF_AUTH) ? r1_g(code) : (n < Nmax - 1 ? n : -1);
}
// physical slot of logical slot s
__device__ __forceinline__ int phys(int hoff, int s, int Nmax) {
  int p = hoff + s;
  return p >= Nmax ? p - Nmax : p;
}
// Arrival time of the row's head slot (the reference's x[i, Nmax + 0]; read by delta_travel_time only) from the packed
// state: the head's slot record while the row holds somebody; the head slot of an EMPTY row is the pending garbage triple
// (arrival = the last frame's clock t_last), a dead slot of a clean row (zero), or whatever the store holds (dirty rows).
// INVARIANT: gc8 must have been stored by the frame that set TLF_AUTH on the row. Inside a rollout that holds for the last
// frame and for the metric environments only (FrameOut::write_gc): in-call readers are restricted to those environments
// (dtt_node, b < m_env), and launch_direction refuses the per-edge dtt of a frame whose predecessor did not store the byte.
__device__ __forceinline__ float head_arrival(const float* __restrict__ slots, int64_t lds, const uint8_t* __restrict__ gc8,
                                              int64_t row, uint32_t hd, uint32_t tlw, int Nmax, float t_last) {
  if ((hd & HD_CNT) == 0u) {
    if (pending_g(tlw, 0, (uint32_t)gc8[row], Nmax) >= 0) return t_last;
    if (!(hd & HD_DIRTY)) return 0.0f;
  }
  return slots[row * lds + SLW * tl_hoff(tlw) + 1];
}

// SELECTED_ROAD value of a sel8 code (node i)
__device__ __forceinline__ float sel_value(const FusedBufs& fb, const int32_t* __restrict__ out_ptr,
                                           const int32_t* __restrict__ out_dst, int64_t i, int64_t row) {
  const uint32_t c = fb.sel8[row] & 0x7Fu;
  return c == SEL_RAW ? fb.sel[row] : (float)out_dst[out_ptr[i] + (int32_t)c];
}
// travel time assigned to an agent entering a row that holds n agents (src/direction_mpnn.py:176-187)
__device__ __forceinline__ float entry_tt(const float4 st, float n) {
  const float t_cong = st.w / (st.x + 10.0f - n);
  return (t_cong != t_cong) ? t_cong : fmaxf(st.y, t_cong);
}

// fused.hip
FusedBufs tarl_to_bufs(const tarl_fused* f);
int tarl_check_fused_core(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax);
int tarl_fused_dead_slots(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax, int materialise,
                          tarl_stream stream);   // clean rows <-> exact slot store (see k_dead_slots)
hipEvent_t tarl_prof_mark(hipStream_t s, int tag);
// dist.hip: tarl_graphdist_rollout with the batch's global environment offset
int tarl_graphdist_rollout_at(const tarl_plan* plan, const float* logits, int64_t B, float temperature,
                              const float* uniform, uint64_t seed, uint64_t counter, void* scratch, int32_t* choice,
                              uint8_t* choice8, uint8_t* sel8, float* log_prob, int64_t env_base, tarl_stream stream);  // sim.hip: live timing of the frame kernels (tags 0..3)
