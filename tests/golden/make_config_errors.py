#!/usr/bin/env python3
"""Error behaviour of the REFERENCE on invalid / unusual task configs -> tests/golden/config_errors.json.

    python tests/golden/make_config_errors.py       (build container only: needs /root/reference)

For every case of tests/golden/error_cases.py the reference's own CartPole / Quadrotor is constructed, reset and stepped twice (on the
stand-ins of tests/golden/ref_stubs.py); recorded: the stage that raised ('init' / 'reset' / 'step' / 'ok'), the exception type and the
start of its message.  tests/test_config_errors.py holds this package's EnvSpec (+ to_c_config, + the seed check) to the same outcome:
same exception TYPE where the reference raises, no exception where it does not — raised at construction here even where upstream only
notices at the first reset / step.
"""
import contextlib
import copy
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import make_golden as MG  # noqa: E402  (installs the stubs, imports the reference)
from tests.golden.error_cases import MUT, SYSTEMS, mutated  # noqa: E402


def run_reference(env_id, cfg):
    stage = 'init'
    try:
        env = MG.ENV_CLS[env_id](**dict(copy.deepcopy(cfg), output_dir='/tmp'))
        stage = 'reset'
        env.reset()
        stage = 'step'
        env.step(env.action_space.sample())
        env.step(env.action_space.sample())
        env.close()
        return ['ok', None, None]
    except BaseException as e:      # noqa: BLE001
        for cid in range(8):        # a failed constructor leaks its Bullet client; base_aviary.py:226 then talks to client 0
            sys.modules['pybullet'].disconnect(cid)
        return [stage, type(e).__name__, str(e)[:120]]


def main():
    out = {}
    for system in SYSTEMS:
        out[system] = {}
        for name in MUT:
            env_id, cfg = mutated(system, name)
            with contextlib.redirect_stdout(io.StringIO()):
                out[system][name] = run_reference(env_id, cfg)
            print(system, name, out[system][name])
    with open(os.path.join(HERE, 'config_errors.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
