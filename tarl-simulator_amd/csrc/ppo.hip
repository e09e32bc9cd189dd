// ppo.hip — the PPO update arithmetic as configured in src/rl/ppo_trainer.py:35-37 (torchrl 0.5.0 GAE / ClipPPOLoss
// formulas; parity unpinned by the reference) and torch.optim.Adam's single-tensor update.
#include <math.h>

#include "tarl_common.h"

#define PPO_BLOCK 256
#define STAT_BLOCKS 256

// ---- GAE: reverse scan over time, one thread per environment ---------------------------------------------------------
// delta_t = r_t + gamma * V'_t * (1 - terminated_t) - V_t ;  A_t = delta_t + gamma*lambda*(1 - done_t) * A_{t+1}
// value_target = A + V (before normalisation). Tensors are time-major [T][B].
__global__ __launch_bounds__(PPO_BLOCK) void k_gae(const float* __restrict__ reward, const float* __restrict__ value,
                                                   const float* __restrict__ next_value,
                                                   const uint8_t* __restrict__ done,
                                                   const uint8_t* __restrict__ terminated, int64_t T, int64_t B,
                                                   float gamma, float lmbda, float* __restrict__ adv,
                                                   float* __restrict__ target) {
  const int64_t b = (int64_t)blockIdx.x * PPO_BLOCK + threadIdx.x;
  if (b >= B) return;
  float run = 0.0f;
  for (int64_t t = T - 1; t >= 0; --t) {
    const int64_t i = t * B + b;
    const float nt = (terminated && terminated[i]) ? 0.0f : 1.0f;
    const float nd = (done && done[i]) ? 0.0f : 1.0f;
    const float v = value[i];
    const float delta = reward[i] + gamma * next_value[i] * nt - v;
    run = delta + gamma * lmbda * nd * run;
    adv[i] = run;
    target[i] = run + v;
  }
}

// ---- advantage statistics (sum, sum of squares, count) in double, fixed two-stage tree => deterministic --------------
__global__ __launch_bounds__(PPO_BLOCK) void k_stats_partial(const float* __restrict__ a, int64_t n,
                                                             double* __restrict__ partial) {
  __shared__ double s1[PPO_BLOCK / 64], s2[PPO_BLOCK / 64];
  double x1 = 0.0, x2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * PPO_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * PPO_BLOCK) {
    const double v = (double)a[i];
    x1 += v;
    x2 += v * v;
  }
  for (int off = 32; off > 0; off >>= 1) {
    x1 += __shfl_down(x1, off);
    x2 += __shfl_down(x2, off);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) {
    s1[wid] = x1;
    s2[wid] = x2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t1 = 0.0, t2 = 0.0;
    for (int w = 0; w < PPO_BLOCK / 64; ++w) {
      t1 += s1[w];
      t2 += s2[w];
    }
    partial[2 * blockIdx.x] = t1;
    partial[2 * blockIdx.x + 1] = t2;
  }
}

__global__ void k_stats_final(const double* __restrict__ partial, int nblocks, int64_t n, double* __restrict__ stats) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double t1 = 0.0, t2 = 0.0;
  for (int i = 0; i < nblocks; ++i) {
    t1 += partial[2 * i];
    t2 += partial[2 * i + 1];
  }
  stats[0] = t1;
  stats[1] = t2;
  stats[2] = (double)n;
}

// Floor of the advantage's standard deviation in average_gae's standardisation (src/rl/ppo_trainer.py:35 configures torchrl
// 0.5.0's GAE(average_gae=True)). SURVEY §3.4 restates it as clamp_min(1e-6); the round-2 review recalls 1e-4 for GAE and
// 1e-6 for ClipPPOLoss(normalize_advantage) in that release — the wheel is not in this image, so neither can be checked
// ("parity unpinned"). Immaterial unless std(A) < 1e-4; one named constant here and in oracle/ppo.py.
#define TARL_ADV_STD_FLOOR 1e-6f
// A <- (A - mean) / max(std, TARL_ADV_STD_FLOOR), unbiased std, from (possibly all-reduced) stats = {sum, sumsq, count}
__global__ __launch_bounds__(PPO_BLOCK) void k_normalize(float* __restrict__ a, int64_t n,
                                                         const double* __restrict__ stats) {
  const int64_t i = (int64_t)blockIdx.x * PPO_BLOCK + threadIdx.x;
  if (i >= n) return;
  const double cnt = stats[2];
  const double mean = stats[0] / cnt;
  double var = (stats[1] - cnt * mean * mean) / (cnt - 1.0);
  if (var < 0.0) var = 0.0;
  float sd = (float)sqrt(var);
  if (sd < TARL_ADV_STD_FLOOR) sd = TARL_ADV_STD_FLOOR;
  a[i] = (a[i] - (float)mean) / sd;
}

// ---- clipped PPO loss: forward values + gradient seeds in one launch (one workgroup; minibatch-sized M) --------------
// out[0..5] = loss_objective, loss_critic, loss_entropy, clip_fraction, kl_approx (= mean(-log_weight)), ESS
__device__ __forceinline__ float blk_sum(float v, float* s_red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < PPO_BLOCK / 64; ++w) t += s_red[w];
  return t;
}

__global__ __launch_bounds__(PPO_BLOCK) void k_ppo_loss(const float* __restrict__ lp_new,
                                                        const float* __restrict__ lp_old,
                                                        const float* __restrict__ adv, const float* __restrict__ value,
                                                        const float* __restrict__ target,
                                                        const float* __restrict__ entropy, int64_t M, float clip_eps,
                                                        float entropy_coef, float critic_coef, float grad_scale,
                                                        float* __restrict__ out, float* __restrict__ g_lp,
                                                        float* __restrict__ g_ent, float* __restrict__ g_val) {
  __shared__ float s_red[PPO_BLOCK / 64];
  const float lo = log1pf(-clip_eps), hi = log1pf(clip_eps);
  const float inv = 1.0f / (float)M;
  float s_obj = 0.0f, s_cr = 0.0f, s_en = 0.0f, s_clip = 0.0f, s_kl = 0.0f, s_w = 0.0f, s_w2 = 0.0f;
  for (int64_t m = threadIdx.x; m < M; m += PPO_BLOCK) {
    const float lw = lp_new[m] - lp_old[m];
    const float a = adv[m];
    const float r = expf(lw);
    const float g1 = r * a;
    const bool inside = (lw >= lo) && (lw <= hi);
    const float lwc = fminf(fmaxf(lw, lo), hi);
    const float g2 = expf(lwc) * a;
    const bool first = g1 <= g2;  // min over the stacked pair keeps the first index on ties
    s_obj += first ? g1 : g2;
    const float dgain = first ? g1 : (inside ? g2 : 0.0f);
    if (g_lp) g_lp[m] = -grad_scale * inv * dgain;
    const float d = value[m] - target[m];
    const float ad = fabsf(d);
    s_cr += ad < 1.0f ? 0.5f * d * d : ad - 0.5f;
    if (g_val) g_val[m] = grad_scale * critic_coef * inv * (ad < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f));
    s_en += entropy[m];
    if (g_ent) g_ent[m] = -grad_scale * entropy_coef * inv;
    s_clip += inside ? 0.0f : 1.0f;
    s_kl += -lw;
    s_w += r;
    s_w2 += r * r;
  }
  const float t_obj = blk_sum(s_obj, s_red), t_cr = blk_sum(s_cr, s_red), t_en = blk_sum(s_en, s_red);
  const float t_clip = blk_sum(s_clip, s_red), t_kl = blk_sum(s_kl, s_red), t_w = blk_sum(s_w, s_red),
              t_w2 = blk_sum(s_w2, s_red);
  if (threadIdx.x == 0) {
    out[0] = -t_obj * inv;
    out[1] = critic_coef * t_cr * inv;
    out[2] = -entropy_coef * t_en * inv;
    out[3] = t_clip * inv;
    out[4] = t_kl * inv;
    out[5] = (t_w2 > 0.0f) ? (t_w * t_w / t_w2) : 0.0f;
  }
}

// ---- Adam (torch.optim.Adam, single-tensor form) over a flat parameter buffer ----------------------------------------
__global__ __launch_bounds__(PPO_BLOCK) void k_adam(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                    float one_minus_b1, float b2, float one_minus_b2, float step_size,
                                                    float bc2_sqrt, float eps, float grad_scale) {
  const int64_t i = (int64_t)blockIdx.x * PPO_BLOCK + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * grad_scale;
  const float mi = m[i] + one_minus_b1 * (gi - m[i]);  // lerp_(grad, 1 - beta1)
  float vi = v[i] * b2;
  vi = vi + (one_minus_b2 * gi) * gi;                  // mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] + (-step_size) * (mi / denom);           // addcdiv_(exp_avg, denom, value = -step_size)
}

// ---- host side -------------------------------------------------------------------------------------------------------
extern "C" int tarl_gae(const float* reward, const float* value, const float* next_value, const uint8_t* done,
                        const uint8_t* terminated, int64_t T, int64_t B, float gamma, float lmbda, float* advantage,
                        float* value_target, tarl_stream stream) {
  TARL_REQUIRE(reward && value && next_value && advantage && value_target, "null argument");
  TARL_REQUIRE(T >= 1 && B >= 1, "bad sizes");
  hipLaunchKernelGGL(k_gae, dim3((unsigned)ceil_div(B, PPO_BLOCK)), dim3(PPO_BLOCK), 0, (hipStream_t)stream, reward,
                     value, next_value, done, terminated, T, B, gamma, lmbda, advantage, value_target);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_advantage_stats(const float* advantage, int64_t n, double* partial, double* stats,
                                    tarl_stream stream) {
  TARL_REQUIRE(advantage && partial && stats && n >= 1, "bad argument");
  const int nb = (int)(ceil_div(n, PPO_BLOCK) < STAT_BLOCKS ? ceil_div(n, PPO_BLOCK) : STAT_BLOCKS);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_stats_partial, dim3(nb), dim3(PPO_BLOCK), 0, s, advantage, n, partial);
  TARL_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(64), 0, s, partial, nb, n, stats);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_advantage_normalize(float* advantage, int64_t n, const double* stats, tarl_stream stream) {
  TARL_REQUIRE(advantage && stats && n >= 1, "bad argument");
  hipLaunchKernelGGL(k_normalize, dim3((unsigned)ceil_div(n, PPO_BLOCK)), dim3(PPO_BLOCK), 0, (hipStream_t)stream,
                     advantage, n, stats);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_ppo_loss(const float* log_prob_new, const float* log_prob_old, const float* advantage,
                             const float* value, const float* value_target, const float* entropy, int64_t M,
                             float clip_epsilon, float entropy_coef, float critic_coef, float grad_scale, float* out6,
                             float* grad_log_prob, float* grad_entropy, float* grad_value, tarl_stream stream) {
  TARL_REQUIRE(log_prob_new && log_prob_old && advantage && value && value_target && entropy && out6, "null argument");
  TARL_REQUIRE(M >= 1, "empty minibatch");
  hipLaunchKernelGGL(k_ppo_loss, dim3(1), dim3(PPO_BLOCK), 0, (hipStream_t)stream, log_prob_new, log_prob_old,
                     advantage, value, value_target, entropy, M, clip_epsilon, entropy_coef, critic_coef, grad_scale,
                     out6, grad_log_prob, grad_entropy, grad_value);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}

extern "C" int tarl_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                              int64_t step, double lr, double beta1, double beta2, double eps, float grad_scale,
                              tarl_stream stream) {
  TARL_REQUIRE(param && grad && exp_avg && exp_avg_sq, "null argument");
  TARL_REQUIRE(n >= 1 && step >= 1, "bad sizes");
  // scalars are formed in double exactly like torch's Python-side arithmetic, then rounded once to fp32
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)ceil_div(n, PPO_BLOCK)), dim3(PPO_BLOCK), 0, (hipStream_t)stream, param,
                     grad, exp_avg, exp_avg_sq, n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), step_size,
                     bc2_sqrt, (float)eps, grad_scale);
  TARL_LAUNCH_CHECK();
  return TARL_OK;
}
