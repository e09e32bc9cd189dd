// noise.hip — the device noise of the production (Philox) path, written out for a listed set of environments.
//
// The rollout kernels (fused.hip, rollout_env.hip, sim.hip) draw their randomness themselves: a Gumbel value per in-edge
// and frame for DirectionMPNN.aggregate's race (the reference: torch.rand_like + -log(-log(u)), src/direction_mpnn.py:136-139)
// and a uniform per source node and frame for GraphDistribution.sample (src/reinforcement_learning.py:66). Both are pure
// functions of (seed, counter, global environment id, index): tarl_noise_export evaluates exactly those functions — the
// same philox_uniform / gumbel_from_u01 of tarl_common.h the kernels call — into caller buffers, so that a CPU checker
// can be fed the very noise a device rollout consumed and its trajectory compared bit for bit (tests/test_gpu_bench_geometry.py).
// Nothing on the product path calls this.
#include "tarl_common.h"

// kind 0: out[e][eid] = Gumbel value of in-edge `eid` (ORIGINAL edge order, like the `gumbel` argument of the step entry
//         points) of environment env[e]: the kernels index the race's noise by the edge's CSC position k.
// kind 1: out[e][g]   = uniform of group g (compact rank of a node with out-edges) of environment env[e].
__global__ __launch_bounds__(256) void k_noise_export(const int64_t* __restrict__ env, int64_t n, int64_t len,
                                                      const int32_t* __restrict__ in_eid, int kind, uint64_t seed,
                                                      uint64_t counter, float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= n * len) return;
  const int64_t e = gid / len, k = gid - e * len;
  const float u = philox_uniform(seed, counter, (uint64_t)env[e] * (uint64_t)len + (uint64_t)k);
  if (kind == 0)
    out[e * len + in_eid[k]] = gumbel_from_u01(u);
  else
    out[gid] = u;
}

extern "C" int tarl_noise_export(const tarl_plan* plan, int kind, uint64_t seed, uint64_t counter, const int64_t* env_ids,
                                 int64_t num_envs, float* out, tarl_stream stream) {
  TARL_REQUIRE(plan && env_ids && out, "null argument");
  TARL_REQUIRE(kind == 0 || kind == 1, "kind: 0 = Direction Gumbel values per edge, 1 = action uniforms per group");
  TARL_REQUIRE(num_envs >= 0, "bad environment count");
  const int64_t len = kind == 0 ? plan->E : plan->G;
  if (num_envs == 0 || len == 0) return TARL_OK;
  hipLaunchKernelGGL(k_noise_export, dim3((unsigned)ceil_div(num_envs * len, 256)), dim3(256), 0, (hipStream_t)stream,
                     env_ids, num_envs, len, plan->in_eid, kind, seed, counter, out);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}
