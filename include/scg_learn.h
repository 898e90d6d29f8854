/* scg_learn.h — C ABI of libscg_learn_<obs>_<hidden>_<act_dim>_<activation>.so: the PPO learner's hot kernels on MI355X
 * (gfx950 matrix cores, exact float32) for ONE actor / critic shape, obs -> H -> H -> {act_dim, 1}
 * (the reference's MLPActor / MLPCritic, /root/reference/safe_control_gym/controllers/ppo/ppo_utils.py:149-199 over
 * math_and_models/neural_networks.py:18-54).  The library is compiled per shape from safe_control_gym_amd/csrc/scg_learn.hip
 * with -DSCG_L_NIN= -DSCG_L_H= -DSCG_L_NU= -DSCG_L_ACT= (see safe_control_gym_amd/_learn.py), like the config-specialised
 * simulator libraries.
 *
 * Conventions as in scg_hip.h: plain C types, every d_* pointer is a caller-owned DEVICE pointer (torch tensors), kernels
 * are enqueued on the caller's hipStream_t and never synchronise, 0 = ok / negative = error + scg_learn_last_error().
 * Parameters and gradients are FLAT float32 vectors in the caller's order; scg_mlp_layout gives the offset (in floats) of
 * each tensor of one network inside them, tensors in torch.nn.Linear layout (weight [out][in] row-major).
 */
#ifndef SCG_LEARN_H
#define SCG_LEARN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int32_t W1, b1;     /* fcs.0.weight [H][obs_dim], fcs.0.bias [H] */
    int32_t W2, b2;     /* fcs.1.weight [H][H],       fcs.1.bias [H] */
    int32_t W3, b3;     /* fcs.2.weight [out][H],     fcs.2.bias [out] */
} scg_mlp_layout;

/* The shape this library was compiled for (obs_dim, hidden, act_dim, activation 0 tanh | 1 relu | 2 leaky_relu). */
void scg_learn_shape(int32_t* obs_dim, int32_t* hidden, int32_t* act_dim, int32_t* activation);

/* MLP.forward on a batch (neural_networks.py:45-54): d_x [m][obs_dim] -> d_out [m][nout], nout = act_dim or 1.
 * d_row_mask (nullable, [m] bytes): sparse evaluation — rows are processed in tiles of 32; a tile without a non-zero mask
 * byte is skipped, and every row whose mask byte is 0 is written as 0 (used for the critic's values of the few
 * time-limit-truncated terminal observations of a rollout, ppo.py:276-283: the output IS the bootstrap term). */
int scg_mlp_forward(const float* d_params, const scg_mlp_layout* layout, int nout, const float* d_x, int m, float* d_out,
                    const uint8_t* d_row_mask, void* stream);

/* One minibatch of PPOAgent.update up to (and including) the gradients — replaces compute_policy_loss,
 * compute_value_loss, both backward passes and the approx-KL of ppo_utils.py:82-131:
 *   policy_loss = -mean(min(ratio adv, clip(ratio, 1 -+ clip_param) adv)),  ratio = exp(logp - logp_old),
 *                 logp = Normal(actor(obs), exp(logstd)).log_prob(act).sum(-1)
 *   entropy_loss = -sum_a (0.5 + 0.5 log 2 pi + logstd_a)         (state-independent std: ppo_utils.py:166)
 *   value_loss  = 0.5 mean((critic(obs) - ret)^2)                  (or the clipped form with use_clipped_value)
 * d_grad [n_params + 1] receives d(policy_loss + entropy_coef entropy_loss)/d(actor params), d(value_loss)/d(critic
 * params) at the parameters' own offsets and approx_kl = mean(logp_old - logp) in the last slot (so that ONE data-parallel
 * all-reduce carries gradients and the gate value); d_stats [4] = policy_loss, value_loss, entropy_loss, approx_kl.
 * The minibatch is the rows d_idx[0..batch) of the flat rollout arrays (batch a multiple of 32).
 * d_workspace: scg_ppo_grad_workspace_bytes(n_workgroups) bytes of scratch (per-workgroup partial gradients; the result
 * is a deterministic sum, no global atomics).  n_workgroups: launch width per network, e.g. the CU count / 2. */
typedef struct {
    const float* d_params;
    scg_mlp_layout actor, critic;
    int32_t logstd_off;             /* offset of actor.logstd [act_dim] */
    int32_t n_params;               /* length of the flat parameter vector */
    const float* d_obs;             /* [M][obs_dim] */
    const float* d_act;             /* [M][act_dim] */
    const float* d_logp_old;        /* [M] */
    const float* d_adv;             /* [M] (already normalised, ppo.py:300) */
    const float* d_ret;             /* [M] */
    const float* d_v_old;           /* [M] (clipped value loss only) */
    const int32_t* d_idx;           /* [batch] row indices of this minibatch */
    int32_t batch;
    float clip_param, entropy_coef;
    int32_t use_clipped_value;
    int32_t n_workgroups;
    void* d_workspace;
    float* d_grad;                  /* [n_params + 1] */
    float* d_stats;                 /* [4] */
} scg_ppo_grad_args;
size_t scg_ppo_grad_workspace_bytes(int n_workgroups);
int scg_ppo_grad(const scg_ppo_grad_args* args, void* stream);

/* The two torch.optim.Adam steps of ppo_utils.py:126-138 on flat buffers (defaults: betas 0.9 / 0.999, eps 1e-8): elements
 * [0, n_actor) belong to the actor and step only when d_g[n] (approx_kl, possibly all-reduced) <= 1.5 target_kl (or
 * target_kl <= 0); the critic's always step.  d_steps [2]: step counts (float), advanced by this call.  d_stats_acc [5]
 * (nullable): running sums of d_stats [4] and of the actor steps taken.  d_block_counter: one zero-initialised 32-bit word
 * of device memory owned by the caller (the kernel's last block advances the counts and re-arms it). */
int scg_adam_gated(float* d_p, const float* d_g, float* d_m, float* d_v, int n, int n_actor, float lr_actor, float lr_critic,
                   float* d_steps, float target_kl, float* d_stats_acc, const float* d_stats, uint32_t* d_block_counter,
                   void* stream);
/* The same with every read of d_g (the gradients AND the approx-KL slot d_g[n]) multiplied by grad_scale: data-parallel callers
 * SUM-all-reduce the flat gradient buffer and pass 1 / world here instead of launching a division in between. */
int scg_adam_gated_scaled(float* d_p, const float* d_g, float* d_m, float* d_v, int n, int n_actor, float lr_actor, float lr_critic,
                          float* d_steps, float target_kl, float* d_stats_acc, const float* d_stats, uint32_t* d_block_counter,
                          float grad_scale, void* stream);

/* One optimiser step of PPOAgent.update (ppo_utils.py:113-146) on ONE GPU in two launches: the gradient kernel of scg_ppo_grad, then
 * a kernel that sums the workgroups' partial gradients AND applies the two gated Adam steps of scg_adam_gated element by element (each
 * parameter is summed and stepped by exactly one thread; d_grad / d_stats are filled as by scg_ppo_grad; d_params is updated in place).
 * The step counts are double-buffered: read from d_steps_in [2], written to d_steps_out [2] (two different buffers — the caller
 * alternates them) — which is what lets the kernel do without a device-scope fence.  Data-parallel callers, whose gradients are
 * all-reduced between the two halves, keep scg_ppo_grad + scg_adam_gated. */
int scg_ppo_step(const scg_ppo_grad_args* args, float* d_m, float* d_v, float lr_actor, float lr_critic, const float* d_steps_in,
                 float* d_steps_out, float target_kl, float* d_stats_acc, void* stream);

/* The gradient kernel has two forms: when no wave of the launch has more than one 32-row tile (batch / 32 <= 4 n_workgroups — the shipped
 * minibatch sizes) the waves exchange their transposed tiles and each forms one tile row of dW2 over all the workgroup's samples, otherwise
 * every wave accumulates its own products over its tiles and the waves are summed afterwards.  Same results up to the summation order inside
 * dW2 (everything else bit for bit); this hook forces the accumulating form at any size (A/B runs, tests/test_gpu_learn.py). */
void scg_learn_force_accumulating_form(int on);

/* The collector's post-processing between rollout and update (PPO.train_step, controllers/ppo/ppo.py:276-300) over the [T][N] rollout,
 * four launches in place of ~30 elementwise / reduction kernels:
 *   scg_ppo_returns_prepare    d_trunc = done & (flags & 1) (time truncation is not termination), d_mask = 1 - done, d_rew_out = rew
 *                              (scg_gae adds gamma * terminal_v to it in place), d_v_out = d_v_all[:T]; run it in front of the masked
 *                              critic pass over the terminal observations (scg_mlp_forward with d_row_mask = d_trunc)
 *   scg_ppo_returns_moments    d_moments[3] = {sum adv, sum adv^2, T N} (fixed-order two-stage sum); d_episode_acc (nullable, [N][8]:
 *                              scg_policy_rollout.d_episode_acc): columns 0..3 are added to d_episode_totals[4] and the array is zeroed
 *   scg_ppo_returns_normalise  d_out = (adv - mean) / (std + 1e-6), population std (ppo.py:300), from d_moments — all-reduce (sum) the
 *                              moments between the two calls on several ranks.  d_out may alias d_adv. */
int scg_ppo_returns_prepare(const uint8_t* d_done, const uint8_t* d_flags, const float* d_rew, const float* d_v_all, int T, int N,
                            uint8_t* d_trunc, float* d_mask, float* d_rew_out, float* d_v_out, void* stream);
size_t scg_ppo_returns_scratch_bytes(void);
int scg_ppo_returns_moments(const float* d_adv, int T, int N, float* d_episode_acc, float* d_scratch, float* d_moments,
                            float* d_episode_totals, void* stream);
int scg_ppo_returns_normalise(const float* d_adv, const float* d_moments, int T, int N, float* d_out, void* stream);

/* d_out[i] = pi(i) for i < count, pi a keyed pseudo-random permutation of [0, n) (count <= n): the shuffled row indices of
 * one epoch's minibatches (SubsetRandomSampler + BatchSampler(drop_last=True), ppo_utils.py:358-371), one launch. */
int scg_random_permutation(int32_t* d_out, int n, int count, uint64_t key, void* stream);
/* The same with the key taken from DEVICE memory: d_key_state [2] = {base key, epochs drawn so far}, key = base +
 * 0x9E3779B97F4A7C15 * (drawn + epoch_offset + 1) (mod 2^64) — the sequence scg_random_permutation gives a host that counts its
 * epochs.  A caller that captures its epochs in a HIP graph advances `drawn` on the stream after them (one add), and every replay
 * shuffles afresh; nothing about the shuffle is baked into the graph. */
int scg_random_permutation_keyed(int32_t* d_out, int n, int count, const uint64_t* d_key_state, uint32_t epoch_offset, void* stream);

const char* scg_learn_last_error(void);
const char* scg_learn_source_hash_tag(void);

#ifdef __cplusplus
}
#endif
#endif /* SCG_LEARN_H */
