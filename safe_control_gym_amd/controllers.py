"""Controller-level drop-in: the reference's controller ids and constructor / method surface on the HIP engine.

    ctrl = make('ppo', env_func, training=True, checkpoint_path=..., output_dir=..., seed=..., **algo_config)
    ctrl.reset(); ctrl.learn(); ctrl.save(path); results = ctrl.run(n_episodes=10); ctrl.close()

is how examples/rl/train_rl_controller.py:32-60 and experiments drive every RL controller
(/root/reference/safe_control_gym/controllers/__init__.py:29-47 registers 'ppo', 'sac', 'rarl', 'rap', 'safe_explorer_ppo';
base_controller.py:12-140 is the surface).  `env_func` is the reference's `partial(make, task, **task_config)`; algo_config
keys and defaults are those of controllers/<algo>/<algo>.yaml (get_config(idx) returns them).

These classes are thin: they resolve env_func to (task id, YAML keys), build the batched HipVecEnv (training env of
`rollout_batch_size` envs, evaluation env of `eval_batch_size`), and delegate to ppo.PPO / sac.SAC / rarl.RARL / rarl.RAP /
safe_explorer.SafeExplorerPPO.  Differences from upstream that a caller can see: logging is a dict per `log_interval`
(no tensorboard / text logger), `run()` evaluates `n_episodes` episodes in parallel (one per eval env) instead of one after
another, and there is no CPU path (`use_gpu=False` raises).
"""
import os

import numpy as np
import torch

from safe_control_gym_amd import _lib as L
from safe_control_gym_amd.record_episode_statistics import resolve_env_func
from safe_control_gym_amd.vec_env import HipVecEnv

_RUNNER = {'max_env_steps': 1000000, 'num_workers': 1, 'rollout_batch_size': 4, 'deque_size': 10, 'eval_batch_size': 10,
           'log_interval': 0, 'save_interval': 0, 'num_checkpoints': 0, 'eval_interval': 0, 'eval_save_best': False,
           'tensorboard': False}
PPO_DEFAULTS = dict(hidden_dim=64, activation='tanh', norm_obs=False, norm_reward=False, clip_obs=10, clip_reward=10, gamma=0.99,
                    use_gae=False, gae_lambda=0.95, use_clipped_value=False, clip_param=0.2, target_kl=0.01, entropy_coef=0.01,
                    opt_epochs=10, mini_batch_size=64, actor_lr=0.0003, critic_lr=0.001, max_grad_norm=0.5, rollout_steps=100,
                    **_RUNNER)                                                  # controllers/ppo/ppo.yaml
SAC_DEFAULTS = dict(hidden_dim=256, activation='relu', norm_obs=False, norm_reward=False, clip_obs=10., clip_reward=10., gamma=0.99,
                    tau=0.005, init_temperature=0.2, use_entropy_tuning=False, target_entropy=None, train_interval=100,
                    train_batch_size=64, actor_lr=0.001, critic_lr=0.001, entropy_lr=0.001, warm_up_steps=1000,
                    max_buffer_size=1000000, **_RUNNER)                         # controllers/sac/sac.yaml
# controllers/rarl/rarl.yaml (its `pretrained`, `train_protagonist`, `train_adversary` keys are read by nothing upstream either)
RARL_DEFAULTS = dict(PPO_DEFAULTS, agent_iterations=10, adversary_iterations=10, pretrained=None, train_protagonist=True,
                     train_adversary=True)
RARL_DEFAULTS.pop('activation')                                                         # (rarl.yaml / rap.yaml have no such key)
RAP_DEFAULTS = dict({k: v for k, v in RARL_DEFAULTS.items() if k not in ('pretrained', 'train_protagonist', 'train_adversary')},
                    num_adversaries=2)                                                  # controllers/rarl/rap.yaml


# controllers/safe_explorer/safe_ppo.yaml
SAFE_EXPLORER_PPO_DEFAULTS = dict({k: v for k, v in PPO_DEFAULTS.items() if k != 'activation'}, pretraining=True, pretrained=None,
                                  constraint_hidden_dim=10, constraint_lr=0.0001, constraint_batch_size=256,
                                  constraint_steps_per_epoch=6000, constraint_epochs=25, constraint_eval_steps=1500,
                                  constraint_eval_interval=5, constraint_buffer_size=1000000, constraint_slack=None)


class _Deterministic:
    def __init__(self, ac):
        self.ac = ac

    def act(self, obs):
        return self.ac.act(obs, deterministic=True)


class HipController:
    """base_controller.py:12-140 for the RL family."""
    DEFAULTS = {}

    def __init__(self, env_func, training=True, checkpoint_path='model_latest.pt', output_dir='temp', use_gpu=True, seed=0, **kwargs):
        if not use_gpu or not torch.cuda.is_available():
            raise L.ScgError('the HIP engine has no CPU path (use_gpu=False / no HIP device)')
        self.env_func, self.training, self.checkpoint_path, self.output_dir, self.seed = env_func, training, checkpoint_path, output_dir, seed
        self.use_gpu, self.device = True, torch.device('cuda', torch.cuda.current_device())
        cfg = dict(self.DEFAULTS)
        cfg.update(kwargs)
        self.algo_config = cfg
        for k, v in cfg.items():                    # base_controller.py:40-41: algo args become attributes
            setattr(self, k, v)
        self.env_id, self.task_config = resolve_env_func(env_func)
        # keys of the reference's task YAMLs that are HipVecEnv's own named parameters or moot here: every
        # examples/rl/config_overrides/*/*.yaml carries `seed:` inside task_config, and env_func is
        # partial(make, task, output_dir=..., **task_config) — the controller's `seed` argument wins, as upstream
        # (make_vec_envs(env_func, None, n, workers, seed), ppo.py:48)
        for k in ('output_dir', 'seed', 'num_envs', 'return_numpy', 'policy', 'device'):
            self.task_config.pop(k, None)
        self.results_dict = {}
        self._build()

    # -- construction helpers
    def _vec(self, n, seed, policy=None, **over):
        return HipVecEnv(self.env_id, n, seed=seed, return_numpy=False, policy=policy, **dict(self.task_config, **over))

    def _policy_shape(self):
        return None

    def _build(self):
        raise NotImplementedError

    # -- the reference surface
    @property
    def total_steps(self):
        return self.impl.total_steps

    @property
    def agent(self):
        return self.impl.agent

    def reset(self):
        """ppo.py:96-110: (re)initialise for training or evaluation.  The batched env was reset at construction and keeps its
        episode statistics on the device; nothing else to do."""
        self.results_dict = {}

    def reset_before_run(self, obs=None, info=None, env=None):
        self.results_dict = {}

    def close(self):
        self.env.close()
        if getattr(self, 'eval_env', None) is not None:
            self.eval_env.close()

    def save(self, path):
        self.impl.save(path)

    def load(self, path):
        self.impl.load(path, training=self.training)

    def train_step(self):
        return self.impl.train_step()

    def _act_module(self):
        return self.impl.agent.ac

    def select_action(self, obs, info=None):
        with torch.inference_mode():
            o = torch.as_tensor(np.asarray(obs), dtype=torch.float32, device=self.device)
            nz = getattr(self.impl, 'obs_normalizer', None)
            if nz is not None and getattr(self, 'norm_obs', False):
                frozen = nz.read_only
                nz.set_read_only()
                o = nz(o)
                nz.read_only = frozen
            return self._act_module().act(o).cpu().numpy()

    def run(self, env=None, render=False, n_episodes=10, verbose=False, **kwargs):
        """ppo.py:210-257: evaluation with the current (deterministic) policy — here `n_episodes` episodes side by side, one
        per env of the evaluation env.  Returns the reference's dict: ep_returns, ep_lengths (+ constraint_violation, mse)."""
        from safe_control_gym_amd.ppo import evaluate
        if render:
            raise NotImplementedError('no renderer: the simulator has no GUI')
        ev = env if isinstance(env, HipVecEnv) else None
        if ev is None:
            if getattr(self, '_run_env', None) is None or self._run_env.num_envs != n_episodes:
                self._run_env = self._vec(n_episodes, self.seed * 111, self._policy_shape())
            ev = self._run_env
        nz = self.impl.obs_normalizer if getattr(self, 'norm_obs', False) and hasattr(self.impl, 'obs_normalizer') else None
        pol = None
        if nz is None and getattr(self.impl, '_fused_rollout', False) and ev.policy_shape is not None:
            pol = self.impl._policy_struct(True)
        evaluate(self._act_module(), ev, obs_normalizer=nz, policy=pol)
        if pol is not None:
            a = ev._eval_fused['acc']
            tot = {'ret': a[:, 1], 'length': a[:, 2], 'viol': a[:, 3], 'mse': a[:, 4]}
        else:
            tot = ev._eval_cache['acc']
        out = {'ep_returns': tot['ret'].double().cpu().numpy(), 'ep_lengths': tot['length'].double().cpu().numpy(),
               'constraint_violation': tot['viol'].double().cpu().numpy(), 'mse': tot['mse'].double().cpu().numpy()}
        self.results_dict = out
        return out

    def learn(self, env=None, **kwargs):
        """ppo.py:150-193: train_step until max_env_steps with the reference's checkpoint / evaluation / logging cadence."""
        hist = []
        while self.total_steps < self.max_env_steps:
            before = self.total_steps
            results = self.train_step()
            crossed = lambda k: bool(k) and (self.total_steps // k) > (before // k)          # noqa: E731
            # (upstream tests `total_steps % interval == 0`, which only fires when the interval is a multiple of the per-iteration
            #  step count; crossing the multiple is the same cadence without that restriction)
            if self.total_steps >= self.max_env_steps or crossed(self.save_interval):
                self.save(self.checkpoint_path)
                self.save(os.path.join(self.output_dir, 'checkpoints', f'model_{self.total_steps}.pt'))
            if crossed(self.eval_interval):
                ev = self.run(n_episodes=self.eval_batch_size)
                results['eval'] = ev
                score = float(ev['ep_returns'].mean())
                if self.eval_save_best and getattr(self, 'eval_best_score', -np.inf) < score:
                    self.eval_best_score = score
                    self.save(os.path.join(self.output_dir, 'model_best.pt'))
            if crossed(self.log_interval):
                hist.append({k: v for k, v in results.items() if not isinstance(v, dict)})
        return hist


class PPO(HipController):
    """controllers/ppo/ppo.py:34-303."""
    DEFAULTS = PPO_DEFAULTS

    def _policy_shape(self):
        return (self.hidden_dim, self.activation) if not (self.norm_obs or self.norm_reward) else None

    def _build(self):
        from safe_control_gym_amd import ppo
        pcfg = ppo.PPOConfig.from_dict(self.algo_config)
        n = self.rollout_batch_size if self.training else self.eval_batch_size
        self.env = self._vec(n, self.seed, self._policy_shape())
        self.eval_env = None
        self.impl = self._make_impl(pcfg)

    def _make_impl(self, pcfg):
        from safe_control_gym_amd import ppo
        return ppo.PPO(self.env, pcfg, seed=self.seed)


class RARL(PPO):
    """controllers/rarl/rarl.py."""
    DEFAULTS = RARL_DEFAULTS

    def _policy_shape(self):
        return None

    def _make_impl(self, pcfg):
        from safe_control_gym_amd import rarl
        return rarl.RARL(self.env, pcfg, seed=self.seed, agent_iterations=self.agent_iterations,
                         adversary_iterations=self.adversary_iterations)


class RAP(RARL):
    """controllers/rarl/rap.py."""
    DEFAULTS = RAP_DEFAULTS

    def _make_impl(self, pcfg):
        from safe_control_gym_amd import rarl
        return rarl.RAP(self.env, pcfg, seed=self.seed, num_adversaries=self.num_adversaries)


class SAC(HipController):
    """controllers/sac/sac.py:35-335."""
    DEFAULTS = SAC_DEFAULTS

    def _build(self):
        from safe_control_gym_amd import sac
        scfg = sac.SACConfig.from_dict(self.algo_config)
        n = self.rollout_batch_size if self.training else self.eval_batch_size
        self.env = self._vec(n, self.seed)
        self.eval_env = None
        self.impl = sac.SAC(self.env, scfg, seed=self.seed)
        self._det = self.impl.agent.deterministic_policy()  # one object: evaluate() caches its captured graph per policy object

    def _act_module(self):
        return self._det

    def save(self, path, save_buffer=False):
        """sac.py:119-141: agent + (training) total_steps, obs, RNG state, env random state and, with save_buffer, the replay ring."""
        self.impl.save(path, training=self.training, save_buffer=save_buffer)

    def load(self, path):
        self.impl.load(path, training=self.training)


class SafeExplorerPPO(PPO):
    """controllers/safe_explorer/safe_ppo.py:32-466 — two phases selected by the config, as upstream:
      pretraining: True    learn() = `constraint_epochs` x pretrain_step (random-action transitions -> constraint models); the
                           checkpoint's 'safety_layer' is what the second phase loads;
      pretraining: False   reset() loads the safety layer from `pretrained` (a checkpoint file, or a directory holding
                           model_latest.pt; safe_ppo.py:96-100) and learn() is PPO with the safety-filtered policy."""
    DEFAULTS = SAFE_EXPLORER_PPO_DEFAULTS

    def _policy_shape(self):
        return None                                 # the safety layer sits inside the actor: no fused collector

    def _build(self):
        self.activation = self.algo_config.setdefault('activation', 'tanh')         # (safe_ppo_utils.py: the PPO networks' default)
        super()._build()
        self.safety_layer = self.impl.safety_layer
        self.num_constraints = self.impl.C
        self.impl.pretrain_steps = 0

    # pre-training epochs done: lives on the implementation object so that it is saved / restored with its checkpoint
    # (a resumed pre-training run continues at epoch k instead of repeating all `constraint_epochs`)
    _pretrain_steps = property(lambda self: self.impl.pretrain_steps,
                               lambda self, v: setattr(self.impl, 'pretrain_steps', int(v)))

    def _make_impl(self, pcfg):
        from safe_control_gym_amd import safe_explorer
        slack = self.constraint_slack
        if not isinstance(slack, (int, float, list, tuple)):        # safe_explorer_utils.py:47 asserts the same (the YAML default is null)
            raise AssertionError('constraint_slack must be a number or a list (one value per state constraint)')
        return safe_explorer.SafeExplorerPPO(self.env, pcfg, seed=self.seed, constraint_hidden_dim=self.constraint_hidden_dim,
                                             constraint_lr=self.constraint_lr, constraint_slack=slack,
                                             constraint_batch_size=self.constraint_batch_size,
                                             constraint_buffer_size=self.constraint_buffer_size)

    @property
    def total_steps(self):
        return self._pretrain_steps if (self.training and self.pretraining) else self.impl.total_steps

    def reset(self):
        super().reset()
        if self.training and not self.pretraining:
            if not self.pretrained:
                raise AssertionError('Must provide a pre-trained model for adaptation.')         # safe_ppo.py:97
            self.impl.load_safety_layer(self.pretrained)

    def pretrain_step(self):
        import time
        t0 = time.perf_counter()
        losses = self.impl.pretrain_step(self.constraint_steps_per_epoch, self.constraint_batch_size)
        self._pretrain_steps += 1
        res = {f'constraint_{i}_loss': v for i, v in enumerate(losses)}
        res.update({'step': self._pretrain_steps, 'elapsed_time': time.perf_counter() - t0})
        return res

    def eval_constraint_models(self):
        return {f'constraint_{i}_loss': v for i, v in
                enumerate(self.impl.eval_constraint_models(self.constraint_eval_steps, self.constraint_batch_size))}

    def select_action(self, obs, info=None):
        """safe_ppo.py:215-228: the current constraint values are the policy's second input."""
        with torch.inference_mode():
            o = torch.as_tensor(np.asarray(obs), dtype=torch.float32, device=self.device)
            c = torch.as_tensor(np.asarray(info['constraint_values'])[..., :self.num_constraints], dtype=torch.float32, device=self.device)
            nz = self.impl.obs_normalizer
            frozen = nz.read_only
            nz.set_read_only()
            o = nz(o)
            nz.read_only = frozen
            batched = o.dim() == 2
            a = self.impl.agent.ac.act(o if batched else o[None], c if batched else c[None])
            return (a if batched else a[0]).cpu().numpy()

    def run(self, env=None, render=False, n_episodes=10, verbose=False, **kwargs):
        if render:
            raise NotImplementedError('no renderer: the simulator has no GUI')
        ev = env if isinstance(env, HipVecEnv) else None
        if ev is None:
            if getattr(self, '_run_env', None) is None or self._run_env.num_envs != n_episodes:
                self._run_env = self._vec(n_episodes, self.seed * 111)
            ev = self._run_env
        tot = self.impl.evaluate(ev)
        out = {'ep_returns': tot['ret'].double().cpu().numpy(), 'ep_lengths': tot['length'].double().cpu().numpy(),
               'constraint_violation': tot['viol'].double().cpu().numpy(), 'mse': tot['mse'].double().cpu().numpy()}
        self.results_dict = out
        return out

    def learn(self, env=None, **kwargs):
        if not self.pretraining:
            return super().learn(env, **kwargs)
        hist = []                                   # safe_ppo.py:178-213 with final_step = constraint_epochs, train_func = pretrain_step
        while self._pretrain_steps < self.constraint_epochs:
            results = self.pretrain_step()
            k = self._pretrain_steps
            if k >= self.constraint_epochs or (self.save_interval and k % self.save_interval == 0):
                self.save(self.checkpoint_path)
            if self.eval_interval and k % self.eval_interval == 0:
                results['eval'] = self.eval_constraint_models()
            if self.log_interval and k % self.log_interval == 0:
                hist.append({q: v for q, v in results.items() if not isinstance(v, dict)})
        return hist


# (the ids 'ppo', 'sac', 'rarl', 'rap', 'safe_explorer_ppo' are registered in registration.py with lazy entry points to these classes)
