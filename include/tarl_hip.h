/* tarl_hip.h — C ABI of libtarl_hip.so: hand-written HIP kernels (gfx950 / MI355X) for TARL-simulator's MPNN + PPO
 * routing hot path.
 *
 * The reference (OliBus801/TARL-simulator) is pure Python and has no FFI of its own; the path sits behind Python
 * classes. Each entry point below names the reference interface it replaces (file:line in the
 * reference tree). The host-side mirror of those classes (tarl-simulator_amd/src/...) binds this library with
 * ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - Every data pointer is a DEVICE pointer unless the name ends in `_host`. Buffers are caller-owned (torch
 *    allocations); the library borrows them for the duration of the call and never allocates per call.
 *  - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and the host is never synchronised
 *    (except tarl_plan_create, which uploads the static plan with blocking copies).
 *  - Return value: 0 = ok, negative = tarl_status error; tarl_last_error() returns a thread-local message.
 *  - Batched state: B independent environments over ONE static graph. x is fp32 [B][R][ldx] with
 *    ldx >= F = 3*Nmax+7 (row stride in floats) and x_bstride (env stride in floats); column map as in
 *    src/feature_helpers.py:38-54. agent_features is fp32 [B][A][9] (src/feature_helpers.py:59-71).
 *  - Randomness is an input: pass explicit noise tensors for parity, or NULL to draw Philox4x32-10 on device from
 *    (seed, counter).
 */
#ifndef TARL_HIP_H
#define TARL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TARL_ABI_VERSION 5

typedef enum {
  TARL_OK = 0,
  TARL_ERR_INVALID = -1,   /* bad argument (null pointer, negative size, index out of range) */
  TARL_ERR_HIP = -2,       /* a HIP runtime call failed */
  TARL_ERR_NOMEM = -3,
  TARL_ERR_UNSUPPORTED = -4
} tarl_status;

/* bits of a device status word (sticky; the caller reads it at its next synchronisation point) */
#define TARL_FLAG_COUNT_AT_NMAX 1
#define TARL_FLAG_AMBIGUOUS_EDGES 2
#define TARL_FLAG_PACK_RANGE 4
#define TARL_FLAG_CHOICE_OVERFLOW 8

typedef struct tarl_plan tarl_plan; /* opaque static per-graph plan */
typedef void* tarl_stream;          /* hipStream_t */

int tarl_abi_version(void);
/* The experiment flags (-DTARL_EXP_... of `make variant`) this library was built with; "" for the product build. The test
 * suite refuses to run against a library that reports any (tests/conftest.py): a timing-only developer build must never
 * stand in for the product in a parity run. */
const char* tarl_build_flags(void);
const char* tarl_last_error(void);

/* ---- static plan ---------------------------------------------------------------------------------------------
 * Replaces the per-step sort/argsort/unique of GraphDistribution.__init__ (src/reinforcement_learning.py:21-35)
 * and PyG's per-call gather indices (src/direction_mpnn.py:230, src/response_mpnn.py:40): int32 CSC (in-edges by
 * destination, ascending original edge id) and CSR (out-edges by source) built once.
 * edge_index_host: int64 [2][E] on the HOST. src_order_host (nullable): permutation of 0..E-1 that sorts
 * edge_index[0]; NULL = stable order (the reference's pinned CPU behaviour). */
int tarl_plan_create(const int64_t* edge_index_host, int64_t num_edges, int64_t num_nodes,
                     const int64_t* src_order_host, tarl_plan** out);
void tarl_plan_destroy(tarl_plan* plan);
/* info[0..5] = num_nodes, num_edges, num_groups (distinct sources), max_in_degree, max_out_degree, src_sorted */
int tarl_plan_info(const tarl_plan* plan, int64_t* info6_host);
/* launch-geometry facts of the plan (diagnostics / tests): info3_host = {siblings4: the rows 4c .. 4c+3 share their first
 * four in-edge sources for every c (Direction gather reads them once per chunk), row_siblings: the row pass walks the
 * row-chunk table (rows grouped by their out-edge targets), num_row_chunks} */
int tarl_plan_geometry(const tarl_plan* plan, int64_t* info3_host);

/* ---- traffic-flow step ---------------------------------------------------------------------------------------
 * tarl_direction_step == DirectionMPNN.forward: message + aggregate + update (src/direction_mpnn.py:44-196,210-236).
 *   edge_attr [E]; log_edge_attr [E] = log(edge_attr + 1e-12) and log_eps = log(1e-12), both evaluated by the host
 *   (so the Gumbel scores are bit-identical to the reference's fp32 ones); congestion_constant [R] or NULL
 *   (then recomputed from x as src/simulation_core_model.py:55-67 does); gumbel [B][E] or NULL (device Philox);
 *   delta_travel_time [B][E] or NULL (side output, src/direction_mpnn.py:94-96); chosen [B][R] scratch/out.
 *   x is updated in place (every row, also when nothing was chosen).
 *   status: int32[1] device status word or NULL: TARL_FLAG_COUNT_AT_NMAX is OR-ed in when a count reaches Nmax (a
 *   gridlock-relief move into a full FIFO): the reference raises IndexError at its next update (:172-191). */
int tarl_direction_step(const tarl_plan* plan, float* x, int64_t B, int64_t x_bstride, int64_t ldx, int32_t Nmax,
                        int64_t num_roads, const float* edge_attr, const float* log_edge_attr, float log_eps,
                        const float* congestion_constant, float time, const float* gumbel, uint64_t seed,
                        uint64_t counter, float* delta_travel_time, float* chosen, int32_t* status, tarl_stream stream);

/* tarl_response_step == ResponseMPNN.forward: message + max-aggregate + update (src/response_mpnn.py:25-127).
 *   popped [B][R] uint8 out (the update mask appended to update_history, :125); any_popped: int32[1] or NULL,
 *   set to 1 iff any row of any environment popped (the `.any()` of :106, kept on device). */
int tarl_response_step(const tarl_plan* plan, float* x, int64_t B, int64_t x_bstride, int64_t ldx, int32_t Nmax,
                       int64_t num_roads, uint8_t* popped, int32_t* any_popped, tarl_stream stream);

/* tarl_core_step == SimulationCoreModel.forward (src/simulation_core_model.py:41-83): both rounds, same outputs as
 * the two calls above, fused into fewer launches. */
int tarl_core_step(const tarl_plan* plan, float* x, int64_t B, int64_t x_bstride, int64_t ldx, int32_t Nmax,
                   int64_t num_roads, const float* edge_attr, const float* log_edge_attr, float log_eps,
                   const float* congestion_constant, float time, const float* gumbel, uint64_t seed, uint64_t counter,
                   float* delta_travel_time, float* chosen, uint8_t* popped, int32_t* any_popped, int32_t* status,
                   tarl_stream stream);

/* ---- environment step around the core (src/reinforcement_learning.py:222-309, src/agents/base.py:244-403) -------
 * tarl_apply_action: x[b, src(e), SELECTED_ROAD] = dst(e) for every edge with action[b][e] != 0 (:223-231).
 *   Exactly one of action_onehot (int64 [B][E], the reference's action format) / choice (int32 [B][N]: chosen edge id
 *   per source node, -1 = none) must be non-NULL. */
int tarl_apply_action(const tarl_plan* plan, float* x, int64_t B, int64_t x_bstride, int64_t ldx, int32_t Nmax,
                      const int64_t* action_onehot, const int32_t* choice, tarl_stream stream);

/* tarl_withdraw_step == Agents.withdraw_agent_from_network (src/agents/base.py:348-403) over all num_nodes rows;
 *   adjacency = the plan's edge set (replaces the dense N x N bool matrix). withdrawn [B][num_nodes] uint8 or NULL. */
int tarl_withdraw_step(const tarl_plan* plan, float* x, int64_t B, int64_t x_bstride, int64_t ldx, int32_t Nmax,
                       int64_t num_nodes, float* agent_features, int64_t num_agents, int64_t a_bstride, float time,
                       uint8_t* withdrawn, tarl_stream stream);

/* tarl_insert_step == Agents.insert_agent_into_network (src/agents/base.py:247-331), stable (agent-id) order within
 *   a road. congestion_constant [num_nodes] or NULL (then time_congestion = 0, as :312-313).
 *   ready_scratch: int32 [B][2 * num_agents]. Also emits, when non-NULL, reward [B] = -sum_rows NUMBER_OF_AGENT
 *   (src/reinforcement_learning.py:266-267) and counts [B][num_nodes] (the NUMBER_OF_AGENT column, the critic input). */
int tarl_insert_step(float* x, int64_t B, int64_t x_bstride, int64_t ldx, int32_t Nmax, int64_t num_nodes,
                     float* agent_features, int64_t num_agents, int64_t a_bstride, const float* congestion_constant,
                     float time, int32_t* ready_scratch, float* reward, float* counts, tarl_stream stream);

/* tarl_reset_state == TransportationSimulator.reset + Agents.reset (src/transportation_simulator.py:353-358,
 *   src/agents/base.py:496-503): zero FIFO blocks and NUMBER_OF_AGENT, clear ON_WAY / DONE. agent_features nullable. */
int tarl_reset_state(float* x, int64_t B, int64_t x_bstride, int64_t ldx, int32_t Nmax, int64_t num_nodes,
                     float* agent_features, int64_t num_agents, int64_t a_bstride, tarl_stream stream);

/* ---- GraphDistribution (src/reinforcement_learning.py:15-96) ---------------------------------------------------
 * tarl_graphdist_softmax: proba = scatter_softmax(logits / temperature, src) (:25); logits/proba [B][E] in original
 *   edge order. */
int tarl_graphdist_softmax(const tarl_plan* plan, const float* logits, int64_t B, float temperature, float* proba,
                           tarl_stream stream);
/* tarl_graphdist_sample (:62-80 with the cumsum of :38-42): per source node the first out-edge (plan order) with
 *   u < cum, cum = fp32(fp32(base + s) - fp32(base)) with base/s accumulated in double like torch's CPU cumsum.
 *   uniform [B][num_groups] or NULL (Philox). Outputs (each nullable): action_onehot int64 [B][E]; choice int32 [B][N]
 *   (edge id or -1). group_sums: double scratch [B][num_groups + 1]. */
int tarl_graphdist_sample(const tarl_plan* plan, const float* proba, int64_t B, const float* uniform, uint64_t seed,
                          uint64_t counter, double* group_sums, int64_t* action_onehot, int32_t* choice,
                          tarl_stream stream);
/* tarl_graphdist_rollout: the action draw of one rollout frame in ONE launch for a policy whose logits [B][E] change every
 *   frame: GraphDistribution(logits / temperature).sample() and .log_prob(action) (src/reinforcement_learning.py:16-96),
 *   i.e. tarl_graphdist_softmax -> tarl_graphdist_sample -> tarl_graphdist_logprob_entropy_fwd with identical arithmetic and
 *   summation orders (actions and log-probs are bit-identical to that chain), without materialising the probabilities.
 *   uniform [B][num_groups] or NULL (Philox keyed by seed / counter / b * num_groups + g, as tarl_graphdist_sample).
 *   scratch: tarl_graphdist_rollout_scratch_bytes(plan, B) bytes, 8-byte aligned. Outputs, each nullable: choice int32
 *   [B][N] (edge id, -1 = none); choice8 uint8 [B][N] = the rank byte of the rollout buffers (rank of the chosen
 *   out-edge, or 0x80 | previous rank where nothing was drawn); sel8 uint8 [N][B] = tarl_fused.sel8, updated in place
 *   (== tarl_fused_apply_choice of the drawn action: the choice phase of SimulatorEnv._step, :228-233); log_prob [B]. */
int64_t tarl_graphdist_rollout_scratch_bytes(const tarl_plan* plan, int64_t B);
int tarl_graphdist_rollout(const tarl_plan* plan, const float* logits, int64_t B, float temperature, const float* uniform,
                           uint64_t seed, uint64_t counter, void* scratch, int32_t* choice, uint8_t* choice8,
                           uint8_t* sel8, float* log_prob, tarl_stream stream);
/* tarl_graphdist_mode (:45-55): one-hot (fp32, like zeros_like(proba)) of the per-node argmax, first maximum wins. */
int tarl_graphdist_mode(const tarl_plan* plan, const float* proba, int64_t B, float* mode_onehot, int32_t* choice,
                        tarl_stream stream);
/* tarl_graphdist_logprob_entropy_fwd (:82-96): log_prob [B] = sum_e a_e log(p_e + 1e-8), -inf when the action is not
 *   exactly one edge per source node; entropy [B] = -sum_e p_e log(p_e + 1e-8). Action given as one-hot or choice.
 *   Outputs nullable. One workgroup per batch row, fixed reduction order. */
int tarl_graphdist_logprob_entropy_fwd(const tarl_plan* plan, const float* proba, int64_t B,
                                       const int64_t* action_onehot, const int32_t* choice, float* log_prob,
                                       float* entropy, tarl_stream stream);
/* backward of (log_prob, entropy) w.r.t. logits through the segment softmax: grad_logits [B][E] (overwritten).
 *   grad_log_prob / grad_entropy [B] nullable (= 0). log_prob_fwd [B] nullable: rows whose forward log_prob is -inf
 *   get zero log-prob gradient (the reference assigns -inf through a mask, :91). */
int tarl_graphdist_logprob_entropy_bwd(const tarl_plan* plan, const float* proba, int64_t B, float temperature,
                                       const int64_t* action_onehot, const int32_t* choice,
                                       const float* grad_log_prob, const float* grad_entropy,
                                       const float* log_prob_fwd, float* grad_logits, tarl_stream stream);

/* ---- policy / critic (src/agents/mpnn_agent.py) ------------------------------------------------------------------
 * tarl_policy_edge_logits_fwd: live MPNNPolicyNet.forward (:175-178,215-217):
 *   logits[b][e] = emb[(int) road_index[b][dst(e)]]; road_index points at the ROAD_INDEX observation column with
 *   element strides (ri_bstride, ri_nstride); emb [num_embeddings]. */
int tarl_policy_edge_logits_fwd(const tarl_plan* plan, const float* road_index, int64_t ri_bstride,
                                int64_t ri_nstride, int64_t B, const float* emb, int64_t num_embeddings,
                                float* logits, tarl_stream stream);
/* backward: grad_emb [num_embeddings] += sum_b sum_{e: dst(e)=n} grad_logits[b][e], one fp32 add per (node, embedding) run
 *   of batch rows (bit-reproducible while ROAD_INDEX is one-to-one). scratch (nullable): fp32,
 *   tarl_policy_edge_logits_bwd_scratch_floats(plan, B) elements — with it, a broadcast observation (ri_bstride == 0) and
 *   >= 256 rows the sum over the rows runs in parallel chunks of 64 rows whose partial sums are added in chunk order. */
int64_t tarl_policy_edge_logits_bwd_scratch_floats(const tarl_plan* plan, int64_t B);
int tarl_policy_edge_logits_bwd(const tarl_plan* plan, const float* road_index, int64_t ri_bstride,
                                int64_t ri_nstride, int64_t B, const float* grad_logits, float* grad_emb,
                                int64_t num_embeddings, float* scratch, tarl_stream stream);

/* ---- the per-edge MLP head of MPNNPolicyNet (src/agents/mpnn_agent.py:35-41; evaluation spelled out at :227-231) -----
 * logits[m][e] = W3 relu(W2 relu(W1 cat(x[m][src(e)], x[m][dst(e)], edge_attr[e]) + b1) + b2) + b3, 33 -> 64 -> 32 -> 1,
 * weights in the reference's state-dict layout (edge_mlp.{0,2,4}.{weight,bias}: w1 [64][33], w2 [32][64], w3 [32]).
 * obs16 [M][N][16] = x = cat(node_features (7), agent_features[agent_index] (9)) per node (:166-178), built by
 *   tarl_policy_obs16 (from the reference's observation tensors: node_features [M][N][>=7] with row stride nf_ld,
 *   agent_index int64 [M][N], agent_features [A][9] shared (a_mstride = 0) or per sample) or by tarl_fused_obs16 (from the
 *   packed state of the fused engine, environment b = sample m; x supplies the static LENGTH / MAX_FLOW columns).
 * precision 0: fp32 MFMA (v_mfma_f32_32x32x2_f32; exact fp32 products); 1: bf16 MFMA (v_mfma_f32_32x32x16_bf16; inputs,
 *   weights and the first hidden activation rounded to bf16, fp32 accumulation) — BASELINE config 5's bf16 features;
 *   2: as 1 with obs16 pointing at bf16 observations, uint16 [M][N][16] (tarl_fused_obs16_bf16): same values, half the
 *   bytes gathered, deeper prefetch; 3: fp32 ACCURACY on the bf16 pipe — fp32 observations, weights and the first hidden
 *   activation each split into three exact bf16 pieces, the six piece products of order >= 2^-16 accumulated in fp32
 *   (logits within a few fp32 ulp of precision 0's; 50 bf16 MFMAs instead of 66 fp32 MFMAs of twice the passes).
 * tarl_policy_edge_mlp_bwd ACCUMULATES (+=) the gradients of sum(grad_logits * logits) into gw1 [64][33], gb1 [64],
 *   gw2 [32][64], gb2 [32], gw3 [32], gb3 [1] (fp32, fixed reduction order); scratch: fp32
 *   [tarl_policy_edge_mlp_bwd_scratch_floats(plan, M)]. Observations receive no gradient. */
int tarl_policy_obs16(const float* node_features, int64_t nf_ld, const int64_t* agent_index,
                      const float* agent_features, int64_t num_agents, int64_t a_mstride, int64_t M, int64_t num_nodes,
                      float* obs16, tarl_stream stream);
int tarl_policy_edge_mlp_fwd(const tarl_plan* plan, const float* obs16, int64_t M, const float* edge_attr,
                             const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                             const float* b3, int precision, float* logits, tarl_stream stream);
int64_t tarl_policy_edge_mlp_bwd_scratch_floats(const tarl_plan* plan, int64_t M);
int tarl_policy_edge_mlp_bwd(const tarl_plan* plan, const float* obs16, int64_t M, const float* edge_attr,
                             const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                             const float* b3, const float* grad_logits, float* scratch, float* gw1, float* gb1,
                             float* gw2, float* gb2, float* gw3, float* gb3, tarl_stream stream);

/* tarl_critic_mlp_fwd == MPNNValueNetSimple.forward (:428-450): value = W3 relu(W2 relu(W1 [counts, time] + b1) + b2) + b3
 *   with the reference's state-dict layout: w1 [64][N+1] (last input column = time), w2 [64][64], w3 [64] (= [1][64]).
 *   counts [M][ldc] = the NUMBER_OF_AGENT observation column per node (row stride ldc >= N); time_rows[m / rows_per_time]
 *   is row m's clock (rows_per_time = B for a time-major [T][B] rollout buffer, 1 for per-row times).
 *   value [M]; h1_out / h2_out [M][64] nullable (post-ReLU activations kept for the backward). fp32 MFMA
 *   (v_mfma_f32_32x32x2_f32, exact fp32 products), hidden tile kept in LDS. */
int tarl_critic_mlp_fwd(const float* counts, int64_t ldc, int64_t M, int64_t N, const float* time_rows,
                        int64_t rows_per_time, const float* w1, const float* b1, const float* w2, const float* b2,
                        const float* w3, const float* b3, float* value, float* h1_out, float* h2_out,
                        tarl_stream stream);
/* backward for minibatch-sized M: ACCUMULATES (+=) into gw1 [64][N+1], gb1 [64], gw2 [64][64], gb2 [64], gw3 [64],
 *   gb3 [1]; scratch: fp32, tarl_critic_mlp_bwd_scratch_floats(M, N) elements ([2][M][64] for the reference's sub-batch of a
 *   few dozen rows; from 512 rows on the reductions over the rows run in parallel row chunks that leave partial sums there,
 *   added in chunk order: deterministic, no atomics). */
int64_t tarl_critic_mlp_bwd_scratch_floats(int64_t M, int64_t N);
int tarl_critic_mlp_bwd(const float* counts, int64_t ldc, int64_t M, int64_t N, const float* time_rows,
                        int64_t rows_per_time, const float* w1, const float* w2, const float* w3, const float* h1,
                        const float* h2, const float* grad_value, float* scratch, float* gw1, float* gb1, float* gw2,
                        float* gb2, float* gw3, float* gb3, tarl_stream stream);

/* ---- PPO update (src/rl/ppo_trainer.py:35-37,129-145; torchrl 0.5.0 GAE / ClipPPOLoss formulas) ---------
 * tarl_gae: GAE(gamma, lmbda) over time-major [T][B] tensors; next_value[t] = V(next obs of frame t) (for an unbroken
 *   rollout pass value + B of a [T+1][B] buffer); done / terminated uint8 nullable (= all false).
 *   advantage (un-normalised) and value_target = advantage + value are written. */
int tarl_gae(const float* reward, const float* value, const float* next_value, const uint8_t* done,
             const uint8_t* terminated, int64_t T, int64_t B, float gamma, float lmbda, float* advantage,
             float* value_target, tarl_stream stream);
/* average_gae=True: stats = {sum, sum of squares, count} in double (partial: double scratch [512]); all-reduce stats
 *   across ranks if the statistic must be global, then tarl_advantage_normalize applies
 *   A <- (A - mean) / max(std_unbiased, 1e-6). */
int tarl_advantage_stats(const float* advantage, int64_t n, double* partial, double* stats3, tarl_stream stream);
int tarl_advantage_normalize(float* advantage, int64_t n, const double* stats3, tarl_stream stream);
/* tarl_ppo_loss == ClipPPOLoss(clip_epsilon) forward + gradient seeds in one launch. out6 = {loss_objective,
 *   loss_critic, loss_entropy, clip_fraction, kl_approx, ESS}; grad_* [M] nullable: d(loss_objective + loss_critic +
 *   loss_entropy) / d(log_prob_new | entropy | value), multiplied by grad_scale (1/world_size for averaged DP grads). */
int tarl_ppo_loss(const float* log_prob_new, const float* log_prob_old, const float* advantage, const float* value,
                  const float* value_target, const float* entropy, int64_t M, float clip_epsilon, float entropy_coef,
                  float critic_coef, float grad_scale, float* out6, float* grad_log_prob, float* grad_entropy,
                  float* grad_value, tarl_stream stream);
/* tarl_adam_step == torch.optim.Adam single-tensor update (src/rl/ppo_trainer.py:37,144) on a flat fp32 buffer;
 *   step is 1-based; grad is multiplied by grad_scale first. */
int tarl_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                   double lr, double beta1, double beta2, double eps, float grad_scale, tarl_stream stream);

/* tarl_critic_mlp_fwd_splitk: the same network for FEW rows (the optimiser minibatch: 32 rows x N columns would
 *   otherwise be one MFMA tile walking all columns on one CU). The first layer is split over blocks of 64 input
 *   columns (partial sums in scratch, tarl_critic_splitk_scratch_floats(M, N) floats), added in block order by a second
 *   launch that finishes the network: deterministic; differs from tarl_critic_mlp_fwd only by the order of the fp32
 *   additions. Same arguments otherwise. */
int64_t tarl_critic_splitk_scratch_floats(int64_t M, int64_t N);
int tarl_critic_mlp_fwd_splitk(const float* counts, int64_t ldc, int64_t M, int64_t N, const float* time_rows,
                               int64_t rows_per_time, const float* w1, const float* b1, const float* w2, const float* b2,
                               const float* w3, const float* b3, float* scratch, float* value, float* h1_out,
                               float* h2_out, tarl_stream stream);

/* tarl_critic_mlp_fwd_slabs: same network on the env-minor rollout buffer counts [M / rows_per_slab][N][rows_per_slab]
 *   (= [frame][node][env]); rows_per_slab must be a multiple of 128 and M a whole number of slabs. value [M] is in
 *   (frame, env) order. */
int tarl_critic_mlp_fwd_slabs(const float* counts, int64_t rows_per_slab, int64_t M, int64_t N, const float* time_rows,
                              int64_t rows_per_time, const float* w1, const float* b1, const float* w2, const float* b2,
                              const float* w3, const float* b3, float* value, tarl_stream stream);

/* the same two forwards on the rollout buffers' count bytes (uint8 NUMBER_OF_AGENT, widened to fp32 in the LDS staging:
 *   identical values): tarl_critic_mlp_fwd_u8 reads counts uint8 [M][ldc] (the env-major buffer of tarl_rollout_env),
 *   tarl_critic_mlp_fwd_slabs_u8 counts uint8 [M / rows_per_slab][N][rows_per_slab] (the env-minor buffer of
 *   tarl_fused_rollout). With split_scratch (tarl_critic_split_scratch_bytes(N) bytes, 16-byte aligned) the first layer runs
 *   on the bf16 matrix cores at fp32 accuracy: the counts are exact in bf16 and W1 is split into three bf16 pieces with
 *   hi + mid + lo == W1 exactly, every product exact, fp32 accumulation (the value differs from the fp32 chain only by
 *   the order of the fp32 additions, ~1e-7 relative); NULL: fp32 MFMA on the widened bytes. */
int64_t tarl_critic_split_scratch_bytes(int64_t N);
int tarl_critic_mlp_fwd_u8(const uint8_t* counts, int64_t ldc, int64_t M, int64_t N, const float* time_rows,
                           int64_t rows_per_time, const float* w1, const float* b1, const float* w2, const float* b2,
                           const float* w3, const float* b3, float* value, float* h1_out, float* h2_out,
                           tarl_stream stream);
int tarl_critic_mlp_fwd_slabs_u8(const uint8_t* counts, int64_t rows_per_slab, int64_t M, int64_t N,
                                 const float* time_rows, int64_t rows_per_time, const float* w1, const float* b1,
                                 const float* w2, const float* b2, const float* w3, const float* b3, void* split_scratch,
                                 float* value, tarl_stream stream);

/* ---- MPNNValueNet (src/agents/mpnn_agent.py:265-402; the message-passing critic the reference defines but never
 * instantiates) in evaluation mode (Dropout = identity), M samples:
 *   value[m] = W_f . [tanh(w_n * mean_{e=(u->v)} tanh(W_m . [node_features[m][v] (7), agent_rows[m][v] (9), edge_features[m][e]] + b_m) + b_n) for u, time_net(time[m])] + b_f
 * node_features [M][N][7]; agent_rows [M][N][9] = agent_features[agent_index] gathered by the caller (NULL: zeros);
 * edge_features [M][E] with stride ef_mstride (0 = shared); time [M].
 * params / grads: HOST arrays of 12 device pointers in the order message_mlp.1.{weight [17], bias}, node_mlp.0.{weight,
 * bias}, final_mlp.0.{weight [N+1], bias}, time_net.0.{weight [32], bias [32]}, time_net.3.{weight [32][32], bias [32]},
 * time_net.6.{weight [32], bias}. fwd optionally saves node_act / agg [M][N] (needed by bwd). bwd ACCUMULATES into grads. */
int tarl_value_mpnn_fwd(const tarl_plan* plan, const float* node_features, int64_t M, const float* agent_rows,
                        const float* edge_features, int64_t ef_mstride, const float* time, const float* const* params,
                        float* value, float* node_act, float* agg, tarl_stream stream);
int tarl_value_mpnn_bwd(const tarl_plan* plan, const float* node_features, int64_t M, const float* agent_rows,
                        const float* edge_features, int64_t ef_mstride, const float* time, const float* const* params,
                        const float* grad_value, const float* node_act, const float* agg, float* const* grads,
                        tarl_stream stream);

/* ---- fused rollout frame (vectorised fast path; same results as the entry points above, 3-4 launches per frame) --------
 * ENV-MINOR layout: every per-(node, environment) buffer is stored [node][environment], so that a wavefront holds 64
 * environments of one node: topology / table loads are wave-uniform and record gathers are coalesced.
 * Caller-owned side buffers that mirror x / agent_features (all device memory, 32-byte aligned). Packed words ("v11"):
 *   hdp  uint32 [N][B][2] = {head_id << 8 | NUMBER_OF_AGENT, bits of the head's departure time}
 *   tl   uint32 [N][B]    = tail_id << 8 | ring-buffer head offset << 1 | bit 0: gc8 is authoritative for the last frame
 *   post uint32 [N][B]    = state after the Direction update: tail' << 8 | non-empty' << 1 | arrived
 *   gc8  uint8  [N][B]    = pending-garbage count + 1; only WRITTEN, and only by rows where something moves in a frame
 *                           (ABI version 4; version 3 kept an 8-byte word {head arrival, code} here: the head's arrival
 *                           time is the arrival field of the head's slot record and is no longer stored twice)
 *   sel8 uint8  [N][B]    = SELECTED_ROAD as the rank of the chosen out-edge in the node's CSR list (bit 7: carried over
 *                           from the previous frame; 0x7F: the fp32 value in sel [N][B] is authoritative)
 *   static records built by pack and shared by all environments (read through the scalar cache): node_rec int32/fp32
 *   [N][36] = {CSC start, in-degree, CSR start, out-degree, MAX_NUMBER_OF_AGENT, FREE_FLOW, ROAD_INDEX, congestion_constant,
 *   travel time at count 0, 3 pad words, the target rows of the first four out-edges, the first four in_rec records};
 *   in_rec [E + 4][5] (CSC order) = {upstream row, the sel8 rank of that row which heads for this road (0xFE: none),
 *   edge_attr, MAX_NUMBER_OF_AGENT of the upstream row, edge id}; out_pad int32 [E + 4] (CSR order) = target row of each
 *   out-edge (sizes are checked against csrc/fused_common.h by static_assert; tarl_hip/ops.py:FusedState allocates them)
 *   st0  [N][4]    = {MAX_NUMBER_OF_AGENT, FREE_FLOW_TIME_TRAVEL, ROAD_INDEX, congestion_constant} (static, shared)
 *   slots [N][B][ld_slots]: slot-interleaved FIFO store, slot s at floats 3s..3s+2 = {agent id, arrival, departure};
 *                  ld_slots = tarl_fused_slot_floats(Nmax) (>= 3*Nmax, padded to a multiple of 16 floats)
 *   acc_lp int64 [acc_slots][B], acc_n / acc_w fp32 [acc_slots][B]: per-frame accumulator banks (log-prob in 2^-32 fixed
 *   point, sum of counts, agents withdrawn; zeroed by pack; acc_slots >= 1 banks spread the atomics of the many
 *   workgroups that serve one environment)
 *   a_origin / a_dest int32 [B][A], a_dep fp32 [B][A], a_status uint8 [B][A] (0 waiting, 1 on the way, 2 done);
 *   a_order int32 [B][A] (optional, may be NULL): each environment's agent ids sorted by DEPARTURE_TIME — with it the
 *   insert kernel scans a window of that order from the cursor cur_lo int32 [B] instead of every agent every frame;
 *   a_dep_sorted fp32 [B][A] (required with a_order): DEPARTURE_TIME in that order, so the scan reads departures
 *   sequentially and touches the per-agent arrays only for the few entries that are due. With a_order also a_rank int32
 *   [B][A] (the inverse permutation: position of agent a in that order), a_win uint32 [B][A][4] and a_ins uint8 [B][A]
 *   (both filled by pack, in that order): the window record {departure bits, origin, agent id, 0} and the "already
 *   inserted" flag, so that one scan step is ONE pair of independent loads instead of a chain of four gathers.
 *   flags int32 [1]: sticky device status word (cleared by pack): TARL_FLAG_COUNT_AT_NMAX = a FIFO count reached Nmax
 *   (the reference raises IndexError there, src/direction_mpnn.py:172-191: the state is outside its defined domain),
 *   TARL_FLAG_AMBIGUOUS_EDGES = two out-edges of a node lead to the same ROAD_INDEX, TARL_FLAG_PACK_RANGE = a packed
 *   count above 255 / agent id at or above 2^24. The caller reads it at its next synchronisation point.
 * Domain of the fused path: Nmax <= 127, out-degree <= 126, agent ids < 2^24 (refused / flagged otherwise).
 * tarl_fused_pack imports x / agent_features (call after construction, reset, or any external write to x); between
 * pack and export the packed state is authoritative for the FIFO columns, NUMBER_OF_AGENT and SELECTED_ROAD;
 * tarl_fused_export writes them back into x in the reference's column layout, bit-identical to the unfused path.
 * agent_features is updated in place by every call. */
typedef struct tarl_fused {
  void* hdp;
  void* tl;
  uint8_t* gc8;
  void* post;
  float* st0;
  float* slots;
  int64_t ld_slots;
  uint8_t* sel8;
  float* sel;
  void* node_rec;
  void* in_rec;
  int32_t* out_pad;
  int64_t* acc_lp;
  float* acc_n;
  float* acc_w;
  int32_t* a_origin;
  int32_t* a_dest;
  float* a_dep;
  uint8_t* a_status;
  const int32_t* a_order;
  int32_t* cur_lo;
  const float* a_dep_sorted;
  void* a_win;
  uint8_t* a_ins;
  const int32_t* a_rank;
  int64_t acc_slots;
  int32_t* flags;
  /* GLOBAL id of environment 0 of this batch (default 0). The device noise streams — the action draws of the policy and the
   * Gumbel races of DirectionMPNN.aggregate — are Philox streams indexed by (env_base + b), not by the lane that happens to
   * serve b: a batch is a window into one global population of environments, so a shard of a larger batch (another
   * rank's slice in a data-parallel job, or a few environments run alone) reproduces, bit for bit, the trajectories those
   * environments have inside the larger batch. The reference has a single environment (env_base + b = 0). */
  int64_t env_base;
  /* hint, 0 = unknown: agents that become due per second and environment at the busiest time of the departure schedule
   * (DEPARTURE_TIME column, src/agents/base.py:244-262). The rollout launcher sizes the insert kernel's share of a wave
   * per environment by it (a frame of dt seconds brings ~due_rate * dt candidates). Never changes a result. */
  float due_rate;
  int32_t reserved_;
  /* device scratch of tarl_fused_bufs_bytes() bytes, 16-byte aligned (required by the frame / rollout entry points): the
   * library keeps a device-resident copy of this table of pointers there (written by a one-thread launch at the head of every
   * such call, on the call's stream), and the row pass and the insert kernels read the pointers they need from it through the
   * scalar cache instead of carrying all of them as kernel arguments — 27 pointer arguments cost those kernels a dozen
   * scalar-register pairs each, most of them spilled (csrc/fused.hip: k_set_bufs). */
  void* bufs_dev;
} tarl_fused;

/* floats per (node, environment) row of tarl_fused.slots for FIFOs of Nmax slots */
int64_t tarl_fused_slot_floats(int32_t Nmax);
/* bytes of tarl_fused.bufs_dev */
int64_t tarl_fused_bufs_bytes(void);
int tarl_fused_pack(const tarl_plan* plan, const tarl_fused* f, const float* x, int64_t B, int64_t x_bstride,
                    int64_t ldx, int32_t Nmax, const float* congestion_constant, const float* edge_attr,
                    const float* agent_features, int64_t num_agents, int64_t a_bstride, tarl_stream stream);
/* tarl_fused_reset == tarl_reset_state applied to the packed state (SimulatorEnv._reset): empty every FIFO (count 0, row
 *   CLEAN: its slots are logically zero from here on, the store itself is not touched), zero the
 *   counters, keep SELECTED_ROAD, clear ON_WAY / DONE in agent_features (for the agents the status SoA marks as on the
 *   way / done, i.e. everything that changed since tarl_fused_pack) and the status SoA, re-arm the insert cursor. */
int tarl_fused_reset(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax, float* agent_features,
                     int64_t num_agents, int64_t a_bstride, tarl_stream stream);
/* last_step_time: the clock value passed to the most recent tarl_fused_frame (stamps the pending garbage slots). */
int tarl_fused_export(const tarl_plan* plan, const tarl_fused* f, float* x, int64_t B, int64_t x_bstride, int64_t ldx,
                      int32_t Nmax, float last_step_time, tarl_stream stream);
/* The live policy (MPNNPolicyNet.forward: logits = emb[ROAD_INDEX(dst)]) does not read the dynamic state, so
 * GraphDistribution's probabilities are the same for every environment and frame between two optimiser steps.
 * tarl_fused_policy_prepare evaluates them once per parameter update — same arithmetic and reduction trees as
 * tarl_policy_edge_logits_fwd + tarl_graphdist_softmax + the cumsum of tarl_graphdist_sample +
 * tarl_graphdist_logprob_entropy_fwd — into per-edge tables in plan (CSR) order: thresholds [E] (fp32 inverse-CDF
 * thresholds), log_probs int64 [E] (log(p + 1e-8) in 2^-32 fixed point: the unit the frame kernels accumulate in),
 * entropy1 [1]; group_base: double scratch [num_groups + 1]. */
int tarl_fused_policy_prepare(const tarl_plan* plan, const tarl_fused* f, const float* emb, int64_t num_embeddings,
                              float temperature, double* group_base, float* thresholds, int64_t* log_probs,
                              float* entropy1, tarl_stream stream);
/* tarl_fused_frame == one collector frame for B environments: GraphDistribution.sample() + log_prob() (+ entropy) and
 *   the choice phase, then tarl_core_step + tarl_withdraw_step + tarl_insert_step, in four launches.
 *   uniform [B][num_groups] or NULL (Philox keyed by policy_seed / policy_counter); gumbel [B][E] or NULL (Philox keyed
 *   by seed / counter). prev_time: the clock of the previous frame on this state (an idle empty FIFO's head arrival, used
 *   by delta_travel_time only). Nullable outputs: delta_travel_time [B][E], popped / withdrawn uint8 [B][N] (env-major,
 *   like the unfused entry points); choice int32 [N][B] (chosen edge id, -1: none) and counts fp32 [N][B] (ENV-MINOR);
 *   log_prob, entropy, reward [B].
 *   thresholds == NULL skips the choice phase: SELECTED_ROAD is what tarl_fused_apply_choice (an externally sampled
 *   action, e.g. of a state-dependent policy) left; choice / log_prob / entropy must then be NULL.
 *   log_prob sums the same terms as tarl_graphdist_logprob_entropy_fwd, accumulated in 2^-32 fixed point (order-
 *   independent, hence deterministic): equal to fp32 rounding, not bit-identical. use_cong = 0 reproduces a graph without congestion_constant in insert. */
int tarl_fused_frame(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax, const float* thresholds,
                     const int64_t* log_probs, const float* entropy1, const float* uniform, uint64_t policy_seed,
                     uint64_t policy_counter, float* agent_features, int64_t num_agents, int64_t a_bstride,
                     const float* edge_attr, const float* log_edge_attr, float log_eps, int use_cong, float time,
                     float prev_time, const float* gumbel, uint64_t seed, uint64_t counter, float* delta_travel_time,
                     uint8_t* popped, uint8_t* withdrawn, int32_t* ins_scratch, int32_t* choice, float* log_prob,
                     float* entropy, float* reward, float* counts, tarl_stream stream);
/* tarl_fused_apply_choice == the choice phase of SimulatorEnv._step (src/reinforcement_learning.py:223-231) on the packed
 *   state for an action sampled elsewhere: choice int32 [B][N] = chosen edge id per source node, -1 = none (the node keeps
 *   its SELECTED_ROAD). */
int tarl_fused_apply_choice(const tarl_plan* plan, const tarl_fused* f, int64_t B, const int32_t* choice,
                            tarl_stream stream);

/* tarl_fused_rollout == T consecutive tarl_fused_frame calls with device noise (uniform = gumbel = NULL), frame t at
 *   clock times_host[t] (HOST array of T floats; prev_time = the clock of the frame before the first, if any) with policy
 *   counter policy_counter0 + t and noise counter counter0 + t — the collector loop of ppo_train
 *   (src/rl/ppo_trainer.py:129-133) in one call, same results as the frame-by-frame calls.
 *   Outputs, each frame-major and nullable: choice uint8 [T][N][B] (the action: rank of the chosen out-edge in the source
 *   node's CSR list = tarl_plan order; bit 7 set: the node drew nothing — the action is infeasible there / the node has no
 *   out-edges), counts uint8 [T][N][B] (NUMBER_OF_AGENT after frame t; ENV-MINOR inside a frame), log_prob / entropy /
 *   reward fp32 [T][B].
 *   What SimulatorEnv._step logs per step (src/reinforcement_learning.py:278-294), device-side: leg int32 [T][B][2] =
 *   {agents departed, agents arrived} per frame (the leg histogram's series); for the first metrics_envs environments the
 *   per-node series dtt_node fp32 [T][N][metrics_envs] (delta_travel_time of the node's out-edges,
 *   src/direction_mpnn.py:94-96) and events uint8 [T][N][metrics_envs] (bit 0: Response pop, bit 1: withdraw — the masks
 *   of update_history / withdraw_history).
 *   Scratch (device): ins_scratch int32 [B][2A]; sel_scratch uint8 [N][B] (only used without a choice buffer);
 *   acc_scratch int64 [acc_slots][B]; choice_scratch int32 [tarl_fused_rollout_scratch_ints(plan, T, B)], 16-byte aligned,
 *   ZERO before its first use. Layout: draw-fixup list | policy records | per-node draw records | log-prob accumulators
 *   int64 [T][B] (last: no other offset depends on T, so one buffer sized for the longest rollout serves shorter ones).
 *   The library re-arms (zeroes) exactly the accumulators a call used; the record tables are rebuilt by every call.
 *   Scheduling of the GraphDistribution sample (state-independent for the live policy, so the T actions are T independent
 *   draws from one set of tables; same Philox streams and arithmetic in every mode, identical results):
 *     with a choice buffer and choice_scratch (default, TARL_ROLLOUT_MERGE=2) the whole action buffer is filled in blocks
 *     of 32 frames on a side stream, in the shadow of the latency-bound frame kernels, and each frame is three launches
 *     (Direction gather, row pass, insert); with acc_scratch and two distinct SELECTED_ROAD slices (TARL_ROLLOUT_MERGE=1)
 *     frame t+1's choice shares ONE launch with frame t's insert; otherwise (TARL_ROLLOUT_MERGE=0) a launch per frame. */
int64_t tarl_fused_rollout_scratch_ints(const tarl_plan* plan, int64_t T, int64_t B);
int tarl_fused_rollout(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax, int64_t T,
                       const float* times_host, float prev_time, const float* thresholds, const int64_t* log_probs,
                       const float* entropy1, uint64_t policy_seed, uint64_t policy_counter0, float* agent_features,
                       int64_t num_agents, int64_t a_bstride, const float* edge_attr, const float* log_edge_attr,
                       float log_eps, int use_cong, uint64_t seed, uint64_t counter0, int32_t* ins_scratch,
                       uint8_t* sel_scratch, int64_t* acc_scratch, int32_t* choice_scratch, uint8_t* choice,
                       float* log_prob, float* entropy, float* reward, uint8_t* counts, int32_t metrics_envs,
                       float* dtt_node, uint8_t* events, int32_t* leg, tarl_stream stream);

/* x = cat(node_features, agent_features[head of the FIFO]) [B][N][16] of the packed state (see tarl_policy_obs16) */
int tarl_fused_obs16(const tarl_plan* plan, const tarl_fused* f, const float* x, int64_t B, int64_t x_bstride,
                     int64_t ldx, int32_t Nmax, const float* agent_features, int64_t num_agents, int64_t a_bstride,
                     float* obs16, tarl_stream stream);
/* tarl_fused_obs16_bf16: the same observation rounded to bf16 (RNE), uint16 [B][N][16] — the input of
 *   tarl_policy_edge_mlp_fwd(precision = 2) ("bf16 MPNN features": half the bytes written and gathered).
 * tarl_fused_obs16_rows: the fp32 observation of a FEW environments: row (env[j], i) -> obs_rows[slot[j]][i][16],
 *   j < rows (< 65536); env / slot int32 device arrays. */
int tarl_fused_obs16_bf16(const tarl_plan* plan, const tarl_fused* f, const float* x, int64_t B, int64_t x_bstride,
                          int64_t ldx, int32_t Nmax, const float* agent_features, int64_t num_agents, int64_t a_bstride,
                          uint16_t* obs16, tarl_stream stream);
int tarl_fused_obs16_rows(const tarl_plan* plan, const tarl_fused* f, const float* x, int64_t B, int64_t x_bstride,
                          int64_t ldx, int32_t Nmax, const float* agent_features, int64_t num_agents, int64_t a_bstride,
                          const int32_t* env, const int32_t* slot, int64_t rows, float* obs_rows, tarl_stream stream);

/* tarl_rollout_env == tarl_fused_rollout with the other mapping: ONE workgroup per environment keeps that environment's
 *   hot records and static columns in LDS (56 B per road + 16 KB; tarl_rollout_env_supported(plan) tells whether the
 *   graph fits the CU's 160 KB, i.e. up to ~2 600 roads) and runs all T frames inside a single launch, with workgroup barriers where the env-minor path has kernel
 *   boundaries. Same packed state in / out (tarl_fused), same noise streams, identical states / agents / actions /
 *   rewards / counts / log-probs. Differences at the interface: times_dev is a DEVICE array of T floats, and the
 *   per-frame outputs are ENV-MAJOR: choice uint8 [T][B][N], counts uint8 [T][B][N], dtt_node fp32 [T][metrics_envs][N],
 *   events uint8 [T][metrics_envs][N] (log_prob / entropy / reward [T][B], leg [T][B][2]).
 *   static_scratch: tarl_rollout_env_scratch_bytes(plan) bytes of 16-byte aligned device memory (the per-edge statics
 *   are packed into 16-byte records there by every call). */
int tarl_rollout_env_supported(const tarl_plan* plan);
int64_t tarl_rollout_env_scratch_bytes(const tarl_plan* plan);
int tarl_rollout_env(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax, int64_t T,
                     const float* times_dev, float prev_time, const float* thresholds, const int64_t* log_probs,
                     const float* entropy1, uint64_t policy_seed, uint64_t policy_counter0, float* agent_features,
                     int64_t num_agents, int64_t a_bstride, const float* edge_attr, const float* log_edge_attr,
                     float log_eps, int use_cong, uint64_t seed, uint64_t counter0, int32_t* ins_scratch,
                     void* static_scratch, uint8_t* choice, float* log_prob, float* entropy, float* reward,
                     uint8_t* counts, int32_t metrics_envs, float* dtt_node, uint8_t* events, int32_t* leg,
                     tarl_stream stream);
/* tarl_fused_set_actions: load one frame's ENV-MAJOR action bytes (choice8 uint8 [B][N]: rank of the chosen out-edge in the
 *   road's CSR list, as tarl_graphdist_rollout / tarl_rollout_env / tarl_fused_rollout_policy write them) as the SELECTED_ROAD
 *   column of the packed state (tarl_fused.sel8, env-minor) — apply_action (src/transportation_simulator.py:411-414) for
 *   recorded or externally sampled actions, as 64 x 64 byte tiles turned through LDS. A byte with bit 7 set means "this road
 *   drew nothing": the road keeps its previous SELECTED_ROAD, and the completed code (previous rank | 0x80) is written back
 *   into choice8. */
int tarl_fused_set_actions(const tarl_plan* plan, const tarl_fused* f, int64_t B, uint8_t* choice8, tarl_stream stream);

/* tarl_fused_rollout_policy: T consecutive frames of SimulatorEnv._step under a STATE-DEPENDENT policy — the per-edge MLP
 *   head (MPNNPolicyNet.edge_mlp, src/agents/mpnn_agent.py:35-41, 227-231) — in one foreign call. Nothing of
 *   GraphDistribution can be hoisted out of the frame; per frame the call queues, on the caller's stream,
 *     tarl_fused_obs16 -> tarl_policy_edge_mlp_fwd(precision) -> tarl_graphdist_rollout(temperature; policy counter
 *     policy_counter0 + t; writes the action into tarl_fused.sel8) -> Direction gather -> row pass -> insert (noise counter
 *     counter0 + t),
 *   with the same results as those calls made one by one. x / x_bstride / ldx: the reference state tensor (static node
 *   columns of the observation). Minibatch observations: the (frame, environment) pairs an optimiser step will use are
 *   drawn BEFORE the rollout; keep_ptr_host (HOST int64 [T + 1], nullable) delimits, per frame, the entries of keep_env /
 *   keep_slot (device int32): observation row keep_env[j] of that frame is copied to obs_keep [slot][N][16].
 *   Scratch (device): obs_scratch fp32 [B][N][16] (16-byte aligned), logits_scratch fp32 [B][E], dist_scratch
 *   (tarl_graphdist_rollout_scratch_bytes), ins_scratch int32 [B][2A].
 *   Outputs, frame-major, nullable: choice8 uint8 [T][B][N] (ENV-MAJOR rank bytes, bit 7: nothing drawn), log_prob /
 *   reward fp32 [T][B], counts uint8 [T][N][B] (env-minor, after frame t), and the per-step logs of tarl_fused_rollout.
 *   precision (of the rollout's logits; the PPO update always evaluates the head with exact fp32 products): 0 = fp32 MFMA
 *   (tarl_policy_edge_mlp_fwd precision 0), 1 = bf16 MFMA on bf16 observations (its precision 2), 2 = fp32-accurate on the
 *   bf16 pipe (its precision 3: operands split into exact bf16 pieces). */
int tarl_fused_rollout_policy(const tarl_plan* plan, const tarl_fused* f, int64_t B, int32_t Nmax, int64_t T,
                              const float* times_host, float prev_time, const float* x, int64_t x_bstride, int64_t ldx,
                              float* agent_features, int64_t num_agents, int64_t a_bstride, const float* edge_attr,
                              const float* log_edge_attr, float log_eps, int use_cong, const float* w1, const float* b1,
                              const float* w2, const float* b2, const float* w3, const float* b3, int precision,
                              float temperature, uint64_t policy_seed, uint64_t policy_counter0, uint64_t seed,
                              uint64_t counter0, const int64_t* keep_ptr_host, const int32_t* keep_env,
                              const int32_t* keep_slot, float* obs_keep, float* obs_scratch, float* logits_scratch,
                              void* dist_scratch, int32_t* ins_scratch, uint8_t* choice8, float* log_prob, float* reward,
                              uint8_t* counts, int32_t metrics_envs, float* dtt_node, uint8_t* events, int32_t* leg,
                              tarl_stream stream);
/* the action / count bytes of a rollout back in the formats of the unfused entry points: choice_eid int32 [rows][N] =
 *   chosen edge id (-1: none), counts_f fp32 [rows][N], for `rows` (frame, environment) pairs given as flat indices
 *   idx int64 [rows] = t * B + b (NULL: all T * B pairs in order). env_minor != 0: the buffers are [T][N][B], else
 *   [T][B][N]. Either output / input pair may be NULL. (The minibatch gather of ppo_train, src/rl/ppo_trainer.py:134.) */
int tarl_rollout_gather(const tarl_plan* plan, const uint8_t* choice, const uint8_t* counts, int64_t T, int64_t B,
                        int env_minor, const int64_t* idx, int64_t rows, int32_t* choice_eid, float* counts_f,
                        tarl_stream stream);

/* ---- shortest-path routing -----------------------------------------------------------------------
 * tarl_edge_travel_time == the edge weights of DijkstraAgents.choice (src/agents/base.py:541-550):
 *   travel_time[b][e] = max(FREE_FLOW[u], congestion_constant[v] / (MAX[u] + 10 - N[u])), u = src(e), v = dst(e),
 *   original edge order, fp32.
 * tarl_apsp == nx.all_pairs_dijkstra_path + the next-hop extraction of src/agents/base.py:556-570, and
 *   nx.shortest_path_length of MPNNPolicyNet.refresh_dijkstra (src/agents/mpnn_agent.py:53-79), for B weight sets
 *   (w_bstride = 0 shares one). Distances accumulate in double like networkx's Python floats; ties are broken exactly
 *   as networkx 3.x's heap does ((distance, push order), successors in edge order), so next_hop is reproducible.
 *   next_hop int64 [B][N][N]: first node after s on the path s -> t, s on the diagonal, -1 when unreachable.
 *   dist fp32 [B][N][N]: +inf when unreachable, 0 on the diagonal. Either output may be NULL.
 *   scratch: tarl_apsp_scratch_bytes(plan, B) bytes of device memory (0 => NULL is fine: state kept in LDS).
 *   The graph must be simple (no duplicate (u, v) edges: a DiGraph would merge them).
 * tarl_select_next_hop == src/agents/base.py:572-580: SELECTED_ROAD[i] = next_hop[i][DESTINATION[head agent of i]]
 *   for every row i (rows with an empty FIFO read agent 0). nh_bstride = 0 shares one table between environments. */
int tarl_edge_travel_time(const tarl_plan* plan, const float* x, int64_t B, int64_t x_bstride, int64_t ldx,
                          int32_t Nmax, const float* congestion_constant, float* travel_time, tarl_stream stream);
int64_t tarl_apsp_scratch_bytes(const tarl_plan* plan, int64_t B);
int tarl_apsp(const tarl_plan* plan, const float* weights, int64_t B, int64_t w_bstride, void* scratch,
              int64_t scratch_bytes, int64_t* next_hop, float* dist, tarl_stream stream);
/* tarl_apsp_f64: tarl_apsp with float64 edge weights (run_msa keeps its link costs in double).
 * tarl_msa_assign == the all-or-nothing step of run_msa (src/algorithms/user_equilibrium_msa.py:117-131): every OD pair
 *   p walks od_origin[p] -> od_dest[p] along next_hop [N][N] and adds od_volume[p] to aux_flow[v] (double, ACCUMULATED)
 *   for every node v entered with is_road[v] != 0 (the origin is skipped; unreachable pairs contribute nothing). */
int tarl_apsp_f64(const tarl_plan* plan, const double* weights, int64_t B, int64_t w_bstride, void* scratch,
                  int64_t scratch_bytes, int64_t* next_hop, float* dist, tarl_stream stream);
int tarl_msa_assign(const int64_t* next_hop, int64_t num_nodes, const int64_t* od_origin, const int64_t* od_dest,
                    const double* od_volume, int64_t num_pairs, const uint8_t* is_road, double* aux_flow,
                    tarl_stream stream);
int tarl_select_next_hop(float* x, int64_t B, int64_t x_bstride, int64_t ldx, int32_t Nmax, int64_t num_nodes,
                         const float* agent_features, int64_t num_agents, int64_t a_bstride, const int64_t* next_hop,
                         int64_t nh_bstride, tarl_stream stream);

/* ---- the device noise, written out (test hook; nothing on the product path calls it) ---------------------------------------
 * The rollouts draw their own randomness: per frame one Gumbel value per in-edge for DirectionMPNN.aggregate's race (the
 * reference: torch.rand_like + -log(-log(u)), src/direction_mpnn.py:136-139) and one uniform per source node for
 * GraphDistribution.sample (src/reinforcement_learning.py:66) — Philox4x32-10 streams keyed by (seed, counter of the frame)
 * and indexed by the environment's GLOBAL id (tarl_fused.env_base + b). tarl_noise_export evaluates the same device
 * functions for the listed environments: kind 0 -> out [num_envs][E] fp32 Gumbel values in ORIGINAL edge order (the layout
 * of the `gumbel` argument of tarl_direction_step / tarl_fused_frame), for frame t of a rollout seed = its `seed`, counter =
 * its `counter0 + t`; kind 1 -> out [num_envs][G] uniforms of the action draw (seed = `policy_seed`, counter =
 * `policy_counter0 + t`; G = nodes with out-edges, in node order). env_ids: int64 [num_envs] global environment ids
 * (device). A CPU checker fed these values replays a device rollout bit for bit (tests/test_gpu_bench_geometry.py). */
int tarl_noise_export(const tarl_plan* plan, int kind, uint64_t seed, uint64_t counter, const int64_t* env_ids,
                      int64_t num_envs, float* out, tarl_stream stream);

/* ---- measurement hook (bench.py roofline leg; nothing comparable in the reference) ------------------------------------
 * tarl_prof_enable(n > 0) brackets the frame kernels of the next n fused frames (Direction message+aggregate, the row
 * pass, the insert [+ next frame's choice] launch) with HIP events on their launch stream; tarl_prof_enable(0) turns it
 * off. tarl_prof_collect synchronises those events and returns, per kernel slot k = 0 (Direction gather), 1 (row pass),
 * 2 (insert [+ choice]), the summed kernel time in ms over all timed frames (ms_all[3]) and over the timed frames with
 * index >= first_late_frame (ms_late[3]); frames[0], frames[1] = the number of frames behind each sum. */
int tarl_prof_enable(int64_t max_frames);
int tarl_prof_collect(int64_t first_late_frame, double* ms_all_host, double* ms_late_host, int64_t* frames_host);

#ifdef __cplusplus
}
#endif
#endif /* TARL_HIP_H */
