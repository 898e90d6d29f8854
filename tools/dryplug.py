"""pytest plugin: a CPU dry run of the HOST side of the `-m gpu` tests, for containers without a GPU.

    PYTHONPATH=tools python -m pytest tests -m gpu -q -p dryplug -p no:cacheprovider --timeout 30 \
        --deselect tests/test_gpu_multirank.py --deselect tests/test_gpu_dropin.py | grep -A40 "==== SUSPECT"

Every GPU test is run until its first HipVecEnv has validated its task config and filled `scg_config` (EnvSpec + to_c_config + seed
check), then stopped with DryRunStop — all tests "fail"; what matters is the SUSPECT list at the end: tests that died INSIDE the host-side
validation (env_config.py / checked_seed), i.e. configs a change to that validation would newly reject on the GPU box.  Used after
tightening EnvSpec's error behaviour with no GPU minutes left (round 3): 0 suspects over 674 tests.
"""
import pytest, torch, sys
torch.cuda.is_available = lambda: True
torch.cuda.current_device = lambda: 0
class DryRunStop(Exception):
    pass
import safe_control_gym_amd.env_config as E
_real = E.EnvSpec.to_c_config
def _wrapped(self, *a, **k):
    _real(self, *a, **k)
    raise DryRunStop()
E.EnvSpec.to_c_config = _wrapped
SUSPECT = []
def pytest_runtest_logreport(report):
    if report.failed:
        txt = str(report.longrepr)
        if 'DryRunStop' in txt:
            return
        if 'env_config.py' in txt or 'checked_seed' in txt:
            SUSPECT.append((report.nodeid, txt[-1500:]))
def pytest_sessionfinish(session):
    print('\n==== SUSPECT host-side config failures:', len(SUSPECT))
    for n, t in SUSPECT[:20]:
        print('----', n); print(t)
