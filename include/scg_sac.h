/* scg_sac.h — C ABI of libscg_sac_<obs>_<hidden>_<act_dim>_<activation>.so: ONE gradient step of the reference's
 * SACAgent.update (/root/reference/safe_control_gym/controllers/sac/sac_utils.py:110-170) on MI355X, fused.
 *
 * What one scg_sac_update call enqueues (8 kernels; no host synchronisation, exact float32, matrix products on
 * v_mfma_f32_32x32x2_f32, every reduction in a fixed order: the step is bitwise reproducible).  A forward pass that feeds a
 * gradient kernel leaves its activation tiles in the workspace (the actor's at obs; the critics' at (obs, act), evaluated next
 * to the target networks), so the gradient kernels start at the loss derivatives:
 *   sample     batch rows ~ U[0, *d_ring_size) of the device replay ring (SACBuffer.sample, sac_utils.py:399-413)
 *   actor fwd  a, log pi = actor(obs) with the reparameterised tanh-Gaussian (sac_utils.py:185-222; log-prob correction
 *              2 (log 2 - u - softplus(-2u)))
 *   q d/da     q1, q2 (obs, a) and d q / d a                                              (compute_policy_loss, :110-127)
 *   actor grad policy_loss = mean(alpha log pi - min(q1, q2)) back through the actor; [entropy_loss for the temperature]
 *   adam       actor (+ log_alpha) step
 *   actor fwd  a', log pi' = actor(next_obs) with the UPDATED actor                       (compute_q_loss, :129-141)
 *   q targets  q1', q2' of the TARGET networks at (next_obs, a')
 *   q grad     critic_loss = mean((q1 - y)^2) + mean((q2 - y)^2), y = rew + gamma mask (min q' - alpha log pi'), both Q
 *   adam       critic step, then Polyak averaging of ALL actor-critic parameters into the target copy (:163-168)
 * Networks (sac_utils.py:173-262 over neural_networks.py:18-54): actor obs -> H (act) -> H (no activation) -> {mu, log_std}
 * [act_dim each]; q1, q2: (obs, act) -> H (act) -> H (act) -> 1.
 *
 * Conventions as in scg_learn.h: plain C types, d_* = caller-owned DEVICE pointers, kernels go to the caller's hipStream_t,
 * 0 = ok / negative = error + scg_sac_last_error().  Parameters, gradients and Adam moments are FLAT float32 vectors in
 * the caller's order with the offsets below; weights in torch.nn.Linear layout ([out][in] row-major).  The actor's head must
 * be laid out as ONE [2 act_dim][H] matrix (mu_layer.weight rows, then log_std_layer.weight rows) and one [2 act_dim] bias.
 */
#ifndef SCG_SAC_H
#define SCG_SAC_H

#include <stddef.h>
#include <stdint.h>

#include "scg_learn.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    /* ---- parameters: flat vector of n_params + 1 floats, [actor | q1 | q2 | log_alpha] */
    float* d_params;
    float* d_target;                /* target copy of the first n_params floats (ac_targ) */
    float* d_grad;                  /* [n_params + 1] scratch: gradients of the current step */
    float* d_m; float* d_v;         /* [n_params + 1] Adam moments */
    float* d_steps;                 /* [3] Adam step counts: actor, critic, log_alpha */
    scg_mlp_layout actor, q1, q2;   /* offsets inside d_params (actor.W3 / b3 = the stacked head) */
    int32_t n_actor;                /* actor parameters are [0, n_actor), the critics' [n_actor, n_params) */
    int32_t n_params;               /* log_alpha sits at d_params[n_params] */
    /* ---- replay ring (SACBuffer): row-major device arrays of `capacity` rows */
    const float* d_obs;             /* [capacity][obs_dim] */
    const float* d_act;             /* [capacity][act_dim] */
    const float* d_rew;             /* [capacity] */
    const float* d_next_obs;        /* [capacity][obs_dim] */
    const float* d_mask;            /* [capacity] */
    const int32_t* d_ring_size;     /* device scalar: rows currently valid (sampling happens on the device) */
    /* ---- hyper-parameters (sac.yaml) */
    int32_t batch;                  /* train_batch_size, a multiple of 32 */
    float gamma, tau;
    float actor_lr, critic_lr, entropy_lr;
    int32_t use_entropy_tuning;
    float target_entropy;
    float act_low[4], act_high[4];  /* action bounds (the actor's tanh output is rescaled to them) */
    /* ---- randomness: Philox4x32-10 keyed by `seed`, counter = (*d_counter, row, stream); d_counter is advanced by the call */
    uint64_t seed;
    uint32_t* d_counter;            /* device scalar */
    /* ---- tests / replay: when non-NULL these replace the in-kernel draws */
    const int32_t* d_idx_in;        /* [batch] ring rows of the minibatch */
    const float* d_eps_in;          /* [batch][act_dim] N(0,1) noise of the policy-loss action */
    const float* d_eps_next_in;     /* [batch][act_dim] N(0,1) noise of the target action */
    /* ---- scratch + outputs */
    void* d_workspace;              /* scg_sac_workspace_bytes(batch) bytes */
    float* d_stats;                 /* [4] policy_loss, critic_loss, entropy_loss, mean log pi of this step */
    float* d_stats_acc;             /* [4] nullable: running sums over calls */
    /* ---- data-parallel use: which part of the step this call enqueues (0 = all).  The two gradient exchanges of a
     * data-parallel step sit between the parts: SCG_SAC_ACTOR_GRAD leaves d(policy_loss)/d(actor) in d_grad[0, n_actor) and the
     * temperature's gradient in d_grad[n_params]; SCG_SAC_CRITIC_GRAD applies the actor / temperature step from d_grad and leaves
     * d(critic_loss)/d(q1, q2) in d_grad[n_actor, n_params); SCG_SAC_FINISH applies the critic step, the Polyak update and advances
     * the counters.  All-reduce (mean) d_grad between the calls; every rank samples its own minibatch (seed per rank). */
    int32_t phases;
} scg_sac_args;
enum { SCG_SAC_ACTOR_GRAD = 1, SCG_SAC_CRITIC_GRAD = 2, SCG_SAC_FINISH = 4, SCG_SAC_ALL = 7 };

void scg_sac_shape(int32_t* obs_dim, int32_t* hidden, int32_t* act_dim, int32_t* activation);
size_t scg_sac_workspace_bytes(int batch);
/* One-time kernel attributes (dynamic LDS > 64 KB); call before capturing scg_sac_update into a HIP graph (not a stream
 * operation).  scg_sac_update / scg_sac_act call it themselves otherwise. */
int scg_sac_prepare(void);
int scg_sac_update(const scg_sac_args* args, void* stream);
/* n_steps whole gradient steps (args->phases must be 0), bit-identical to n_steps scg_sac_update calls, in 7 n_steps + 1 launches: step
 * k's target-action launch (actor at next_obs) also draws step k + 1's minibatch rows and evaluates the actor at its obs rows — the
 * critics' step between the two touches neither the actor nor the replay ring, and the two 128-workgroup jobs fill the chip together.
 * What SACAgent.update's loop over `n_updates` gradient steps (sac.py:307-311 -> sac_utils.py:143-170) enqueues on one GPU. */
int scg_sac_update_n(const scg_sac_args* args, int n_steps, void* stream);

/* The deterministic actor on a batch (MLPActorCritic.act(obs, deterministic=True), sac_utils.py:258-262):
 * d_act_out[m][act_dim] = low + 0.5 (tanh(mu(obs)) + 1)(high - low).  Evaluation / acting without PyTorch kernels. */
int scg_sac_act(const float* d_params, const scg_mlp_layout* actor, const float* act_low, const float* act_high, const float* d_obs,
                int m, float* d_act_out, void* stream);

/* ---- the collector's two device-side pieces (SAC.train_step, /root/reference/safe_control_gym/controllers/sac/sac.py:273-311)
 * A SAMPLED action per observation (MLPActorCritic.act(obs), sac_utils.py:258-262 with deterministic = False):
 * a = low + 0.5 (tanh(mu + exp(clamp(log_std, -20, 2)) eps) + 1)(high - low), eps ~ N(0, 1) from Philox4x32-10 keyed by `seed` with
 * counter = (*d_counter, row, stream 3) — or the caller's d_eps_in [m][act_dim] (tests).  uniform != 0: the warm-up's
 * action_space.sample(), a ~ U[low, high) per dimension (sac.py:276-277; d_params / actor / d_obs unused).  One launch. */
int scg_sac_sample(const float* d_params, const scg_mlp_layout* actor, const float* act_low, const float* act_high, const float* d_obs,
                   int m, uint64_t seed, const uint32_t* d_counter, int uniform, const float* d_eps_in, float* d_act_out, void* stream);
/* One vectorised env step into the replay ring (SACBuffer.push, sac_utils.py:340-370, with sac.py:287-305's time-limit fix-up): rows
 * pos .. pos + n - 1 (mod capacity) <- (d_cur_obs, d_act, d_reward, next observation = d_terminal_obs where the episode was
 * TRUNCATED by the time limit (done and flags bit 0) else d_next_obs, mask = 1 where truncated else 1 - done); d_cur_obs <- d_next_obs
 * (the persistent current-observation batch of the collector); then *d_pos, *d_size_f, *d_size_i32 advance (every one a device
 * scalar: the whole step is capturable in a HIP graph) and *d_counter (the sample noise's counter word, nullable) is incremented. */
typedef struct {
    float* d_obs; float* d_act; float* d_rew; float* d_next_obs; float* d_mask;     /* ring arrays of `capacity` rows */
    int32_t capacity;
    int64_t* d_pos;                 /* write position */
    float* d_size_f;                /* rows valid, as float (nullable) */
    int32_t* d_size_i32;            /* rows valid (nullable; what scg_sac_args.d_ring_size reads) */
    uint32_t* d_counter;            /* nullable */
} scg_sac_ring;
int scg_sac_push(const scg_sac_ring* ring, float* d_cur_obs, const float* d_act, const float* d_reward, const float* d_next_obs,
                 const float* d_terminal_obs, const uint8_t* d_done, const uint8_t* d_flags, int n, void* stream);

const char* scg_sac_last_error(void);
const char* scg_sac_source_hash_tag(void);

#ifdef __cplusplus
}
#endif
#endif /* SCG_SAC_H */
