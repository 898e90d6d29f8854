"""The N > 1 launch path on ONE GPU: two ranks started exactly as the driver starts them (python -m torch.distributed.run,
one process per rank) share cuda:0, with gloo carrying the collectives (SCG_BENCH_BACKEND / SCG_DIST_BACKEND = gloo; RCCL
needs one device per rank).  Covers what the CPU gloo tests cannot: the HIP env shards (disjoint env_id_offset streams),
bench.py's barrier / max-over-ranks / whole-job value, and train_ppo.py's gradient all-reduce keeping ranks in lock-step."""
import json
import os
import socket
import subprocess
import sys

import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _torchrun(script_args, extra_env, timeout=600):
    env = dict(os.environ, **extra_env)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port())] + script_args
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    return res.stdout


def test_bench_two_ranks_on_one_gpu():
    out = _torchrun(['bench.py', '--gpus', '2', '--steps', '500', '--warmup', '100', '--no-secondary'], {'SCG_BENCH_BACKEND': 'gloo'})
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out                     # rank 0 only
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['steps'] == 500 and r['scaling'] == 'weak'
    assert r['config']['finite_outputs'] is True and r['config']['parallelism'] == 'env-shard x2'
    # whole-job value = units of ALL ranks / max-over-ranks time
    assert abs(r['value'] - 2 * r['config']['envs_per_gpu'] * r['steps'] / (r['ms_per_step'] * 1e-3 * r['steps'])) <= 1e-6 * r['value']
    assert 'cpu_baseline' not in r                  # rank 0 at N = 1 only


def test_plain_python_bench_gpus_2_spawns_its_own_ranks_and_mismatches_fail():
    """`python bench.py --gpus 2` (the shape of the driver's N = 1 command, no launcher): bench.py becomes the launcher
    (torch.distributed.run, two ranks) instead of silently running one rank and printing n_gpus: 1; a launcher / --gpus mismatch
    and an RCCL run with fewer GPUs than ranks exit non-zero."""
    env = dict(os.environ, SCG_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    res = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '300', '--warmup', '50', '--no-secondary'], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and r['config']['parallelism'] == 'env-shard x2' and r['config']['collective_backend'] == 'gloo'
    assert r['config']['rccl_ranks'] == 0                                           # gloo carried the collectives here, and the line says so
    # launcher says 2 ranks, --gpus says 1
    bad = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(_free_port()), 'bench.py', '--gpus', '1', '--steps', '50', '--warmup', '5', '--no-secondary'],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and 'WORLD_SIZE=2' in bad.stderr and not [l for l in bad.stdout.splitlines() if l.startswith('{')]
    # RCCL needs one device per rank
    if torch.cuda.device_count() < 2:
        env2 = dict(env); env2.pop('SCG_BENCH_BACKEND')
        bad = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '50', '--warmup', '5', '--no-secondary'], cwd=ROOT, env=env2,
                             capture_output=True, text=True, timeout=300)
        assert bad.returncode != 0 and 'need 2 GPUs' in bad.stderr


def test_bench_ppo_leg_two_ranks_on_one_gpu():
    """The PPO wall-clock leg of bench.py with two ranks (what the driver's N > 1 runs execute over RCCL): env shards, one
    flat gradient all-reduce per minibatch, asynchronous evaluation per rank, rank 0's stop flag broadcast every iteration."""
    out = _torchrun(['bench.py', '--gpus', '2', '--steps', '200', '--warmup', '50', '--ppo-seeds', '1', '--ppo-seconds', '3', '--ppo-envs', '16384'],
                    {'SCG_BENCH_BACKEND': 'gloo', 'SCG_BENCH_PPO_GLOO': '1'})
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and 'secondary' not in r and 'cpu_baseline' not in r
    p = r['ppo']
    assert 'error' not in p, p
    assert p['n_gpus'] == 2 and p['seeds'] == [1] and p['iterations'][0] >= 3 and p['envs_per_gpu'] == 16384
    assert 200.0 < p['target_return'] < 250.0 and p['shipped_model_eval']['episodes'] == 1024    # the shipped model, randomised-init protocol
    assert p['best_eval_return'][0] > 0          # evaluations came back (the policy is far from trained in 3 s on a shared GPU)


def test_train_ppo_two_ranks_stay_in_lock_step(tmp_path):
    base = str(tmp_path / 'final')
    _torchrun(['examples/train_ppo.py', '--envs', '2048', '--minibatch', '16384', '--rollout-steps', '32', '--max-env-steps',
               str(3 * 2 * 2048 * 32), '--max-seconds', '1e9', '--target-return', '1e9', '--eval-envs', '64', '--quiet',
               '--save-final', base], {'SCG_DIST_BACKEND': 'gloo'})
    a, b = (torch.load(f'{base}.rank{r}.pt') for r in range(2))
    assert a['iterations'] == b['iterations'] >= 2
    assert (a['env_id_offset'], b['env_id_offset']) == (0, 2048)
    assert not torch.equal(a['first_obs'], b['first_obs'])                  # disjoint Philox streams: different initial states
    assert torch.isfinite(a['params']).all()
    torch.testing.assert_close(a['params'], b['params'], rtol=0, atol=0)    # same init (broadcast) + same reduced gradients
    # and the weights moved: a rank that skipped the optimiser would also be "in lock-step"
    torch.testing.assert_close(a['init_params'], b['init_params'], rtol=0, atol=0)
    assert (a['params'] - a['init_params']).abs().max() > 1e-4
