"""Registry with the reference's call surface: ``register(idx, entry_point, config_entry_point)``,
``make(idx, *args, **kwargs)``, ``get_config(idx)``
(/root/reference/safe_control_gym/utils/registration.py:118-139), plus ``load_task(name)`` for the
task YAMLs shipped in safe_control_gym_amd/configs/."""
import copy
import importlib
import os

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'configs')


class Spec:
    def __init__(self, idx, entry_point=None, config_entry_point=None):
        self.idx, self.entry_point, self.config_entry_point = idx, entry_point, config_entry_point

    def __repr__(self):
        return f'Spec({self.idx})'

    def _load(self, name):
        mod, attr = name.split(':')
        return getattr(importlib.import_module(mod), attr)

    def get_config(self):
        cep = self.config_entry_point
        if cep is None:
            return {}
        if isinstance(cep, dict):
            return copy.deepcopy(cep)
        if cep.endswith('.yaml'):
            path = cep if os.path.isabs(cep) else os.path.join(CONFIG_DIR, cep)
            with open(path) as f:
                return yaml.safe_load(f)
        return copy.deepcopy(self._load(cep))

    def make(self, *args, **kwargs):
        if self.entry_point is None:
            raise Exception(f'Attempting to make deprecated env {self.idx}.')
        fn = self.entry_point if callable(self.entry_point) else self._load(self.entry_point)
        obj = fn(*args, **kwargs)
        try:
            obj.spec_id = self.idx
        except Exception:                       # noqa: BLE001
            pass
        return obj


class Registry:
    def __init__(self):
        self.specs = {}

    def register(self, idx, **kwargs):
        if idx in self.specs:
            raise Exception(f'Cannot re-register id: {idx}')
        self.specs[idx] = Spec(idx, **kwargs)

    def spec(self, idx):
        try:
            return self.specs[idx]
        except KeyError:
            raise Exception('Key not found in registry.')

    def make(self, idx, *args, **kwargs):
        return self.spec(idx).make(*args, **kwargs)


registry = Registry()


def register(idx, **kwargs):
    return registry.register(idx, **kwargs)


def make(idx, *args, **kwargs):
    return registry.make(idx, *args, **kwargs)


def spec(idx):
    return registry.spec(idx)


def get_config(idx):
    return registry.spec(idx).get_config()


def load_task(name):
    """(env_id, task_config) from safe_control_gym_amd/configs/<name>.yaml.  The YAML's `seed` is only
    the default the reference uses when the caller passes none (train_rl_controller.py:28-36); it is
    returned as task_config['seed'] removed -> use ``load_task_seed`` if you want it."""
    with open(os.path.join(CONFIG_DIR, name + '.yaml')) as f:
        d = yaml.safe_load(f)
    cfg = dict(d['task_config'])
    cfg.pop('seed', None)
    return d['task'], cfg


def load_task_seed(name):
    with open(os.path.join(CONFIG_DIR, name + '.yaml')) as f:
        return yaml.safe_load(f)['task_config'].get('seed')


# env ids of the reference (envs/__init__.py:5-11); entry points build the single-env facade.
register(idx='cartpole', entry_point='safe_control_gym_amd.benchmark_env:CartPole',
         config_entry_point='safe_control_gym_amd.benchmark_env:CARTPOLE_DEFAULT_CONFIG')
register(idx='quadrotor', entry_point='safe_control_gym_amd.benchmark_env:Quadrotor',
         config_entry_point='safe_control_gym_amd.benchmark_env:QUADROTOR_DEFAULT_CONFIG')

# controller ids of the reference (controllers/__init__.py:29-47); classes and YAML defaults load lazily (they import torch)
for _idx, _cls in (('ppo', 'PPO'), ('sac', 'SAC'), ('rarl', 'RARL'), ('rap', 'RAP'), ('safe_explorer_ppo', 'SafeExplorerPPO')):
    register(idx=_idx, entry_point=f'safe_control_gym_amd.controllers:{_cls}',
             config_entry_point=f'safe_control_gym_amd.controllers:{_idx.upper()}_DEFAULTS')
