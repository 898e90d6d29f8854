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


def _torchrun(script_args, extra_env, timeout=600, nproc=2):
    env = dict(os.environ, **extra_env)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
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
    # one diagnostic row per rank: how its data-parallel epoch ran, its iteration time, the spread (what a driver's --gpus N line is read by)
    assert [q['rank'] for q in p['ranks']] == [0, 1] and all(q['dp_path'].startswith('eager') and q['iteration_ms'] > 0 for q in p['ranks'])
    assert p['iteration_ms_max_over_ranks'] >= p['ranks'][0]['iteration_ms'] and p['iteration_ms_skew_max_minus_min'] >= 0
    assert p['roofline']['flops_per_iteration'] > 1e10 and p['roofline']['frac_of_f32_mfma_peak'] > 0


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


def test_bench_sac_leg_two_ranks_on_one_gpu():
    """bench.py's SAC leg with two ranks (BASELINE config #5 is SAC on 8 GPUs; the driver's N > 1 runs execute it over RCCL): env
    shards, replay shards, the data-parallel fused step, rank 0's clock."""
    out = _torchrun(['bench.py', '--gpus', '2', '--steps', '200', '--warmup', '50', '--ppo-seeds', '0', '--sac-seeds', '1', '--sac-seconds', '6'],
                    {'SCG_BENCH_BACKEND': 'gloo', 'SCG_BENCH_SAC_GLOO': '1'})
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    r = json.loads(lines[0])
    assert r['n_gpus'] == 2 and 'watchdog' not in r
    s = r['sac']
    assert 'error' not in s, s
    assert s['n_gpus'] == 2 and s['fused_update'] is True and s['env_steps'][0] > 0 and s['gradient_steps'][0] > 0
    assert 150.0 < s['target_return'] < 250.0
    assert [q['rank'] for q in s['ranks']] == [0, 1] and all(q['dp_path'] == 'eager' for q in s['ranks'])       # (gloo: no captured collectives)
    assert s['roofline']['flops_per_gradient_step'] > 1e9


def test_train_sac_two_ranks_stay_in_lock_step(tmp_path):
    """examples/train_sac.py (BASELINE config #5's shape: SAC, env shards, gradient all-reduce) launched like the driver launches
    bench.py, two ranks on one GPU: the fused data-parallel step keeps the ranks' weights bit-identical while their env shards
    and replay shards differ; the loop ends on the same iteration on both ranks (rank 0's stop flag)."""
    base = str(tmp_path / 'sac')
    _torchrun(['examples/train_sac.py', '--envs', '512', '--batch', '1024', '--updates-per-step', '2', '--warm-up-steps', '2048', '--buffer',
               '100000', '--max-env-steps', str(24 * 2 * 512), '--max-seconds', '1e9', '--quiet', '--save-final', base], {'SCG_DIST_BACKEND': 'gloo'})
    a, b = (torch.load(f'{base}.rank{r}.pt') for r in range(2))
    assert a['fused_update'] and b['fused_update'] and a['vector_steps'] == b['vector_steps'] == 24
    assert (a['env_id_offset'], b['env_id_offset']) == (0, 512) and not torch.equal(a['first_obs'], b['first_obs'])
    torch.testing.assert_close(a['params'], b['params'], rtol=0, atol=0)
    torch.testing.assert_close(a['init_params'], b['init_params'], rtol=0, atol=0)
    assert (a['params'] - a['init_params']).abs().max() > 1e-4


def test_bench_eight_ranks_on_one_gpu_headline_and_ppo_leg():
    """The driver's 8-GPU command line (python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8) with EIGHT ranks
    sharing this box's one GPU over gloo: launcher / port / barrier plumbing, eight env shards, the PPO leg's per-minibatch gradient
    all-reduce, asynchronous evaluation per rank and rank 0's stop-flag broadcast, one JSON line from rank 0.  (No scaling number is
    meaningful here — eight processes time-slice one device; `rccl_ranks: 0` says so in the line.)"""
    out = _torchrun(['bench.py', '--gpus', '8', '--steps', '200', '--warmup', '50', '--envs', '16384', '--ppo-seeds', '1', '--ppo-seconds', '4',
                     '--ppo-envs', '4096', '--sac-seeds', '0'], {'SCG_BENCH_BACKEND': 'gloo', 'SCG_BENCH_PPO_GLOO': '1'}, timeout=900, nproc=8)
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    r = json.loads(lines[0])
    assert r['n_gpus'] == 8 and r['config']['parallelism'] == 'env-shard x8' and r['config']['rccl_ranks'] == 0 and 'watchdog' not in r
    assert r['config']['finite_outputs'] is True and r['config']['envs_per_gpu'] == 16384
    assert abs(r['value'] - 8 * 16384 * r['steps'] / (r['ms_per_step'] * 1e-3 * r['steps'])) <= 1e-6 * r['value']
    p = r['ppo']
    assert 'error' not in p, p
    assert p['n_gpus'] == 8 and p['envs_per_gpu'] == 4096 and p['iterations'][0] >= 2 and p['best_eval_return'][0] > 0


def test_train_ppo_eight_ranks_stay_in_lock_step(tmp_path):
    """examples/train_ppo.py on eight ranks (BASELINE config #4's shape, one GPU shared, gloo): bit-identical weights on all eight
    ranks after three iterations of reduced gradients, eight disjoint env_id_offset streams (pairwise different first observations)."""
    base = str(tmp_path / 'final')
    _torchrun(['examples/train_ppo.py', '--envs', '512', '--minibatch', '16384', '--rollout-steps', '32', '--max-env-steps',
               str(3 * 8 * 512 * 32), '--max-seconds', '1e9', '--target-return', '1e9', '--eval-envs', '64', '--quiet',
               '--save-final', base], {'SCG_DIST_BACKEND': 'gloo'}, timeout=900, nproc=8)
    rs = [torch.load(f'{base}.rank{r}.pt') for r in range(8)]
    assert len({r['iterations'] for r in rs}) == 1 and rs[0]['iterations'] >= 2
    assert [r['env_id_offset'] for r in rs] == [512 * k for k in range(8)]
    for k in range(1, 8):
        torch.testing.assert_close(rs[k]['params'], rs[0]['params'], rtol=0, atol=0)
        torch.testing.assert_close(rs[k]['init_params'], rs[0]['init_params'], rtol=0, atol=0)
        for j in range(k):
            assert not torch.equal(rs[k]['first_obs'], rs[j]['first_obs'])
    assert torch.isfinite(rs[0]['params']).all() and (rs[0]['params'] - rs[0]['init_params']).abs().max() > 1e-4


def _sac_dp_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from safe_control_gym_amd.sac import DeviceReplay, SACAgent, SACConfig
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    low, high = -torch.ones(4, device=dev), torch.ones(4, device=dev)
    torch.manual_seed(100 + rank)                       # different local init: rank 0's weights are broadcast at construction
    ag = SACAgent(24, 4, low, high, SACConfig(hidden_dim=128, activation='relu', use_entropy_tuning=True), dev)
    assert ag.use_fused and not ag.use_graphs
    init = {k: v.clone().cpu() for k, v in ag.ac.state_dict().items()}
    cap, B = 4096, 1024
    g = torch.Generator(device=dev).manual_seed(5 + rank)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)         # noqa: E731
    data = dict(obs=r(cap, 24), act=torch.tanh(r(cap, 4)), rew=r(cap), next_obs=r(cap, 24), mask=(torch.rand(cap, device=dev, generator=g) > 0.05).float())
    buf = DeviceReplay(cap, 24, 4, dev)
    buf.push(data['obs'], data['act'], data['rew'], data['next_obs'], data['mask'])
    steps = []
    for _ in range(2):
        idx = torch.randint(0, cap, (B,), device=dev, generator=g).to(torch.int32)
        eps, eps2 = r(B, 4), r(B, 4)
        F = ag._fused_args(buf, B, idx=idx, eps=eps, eps_next=eps2)
        ag._fused_step_dp(F)
        torch.cuda.synchronize()
        steps.append({'idx': idx.cpu(), 'eps': eps.cpu(), 'eps2': eps2.cpu()})
    torch.save({'init': init, 'params': ag._flat['p'].cpu(), 'targ': ag._flat['targ'].cpu(), 'data': {k: v.cpu() for k, v in data.items()},
                'steps': steps, 'adam_steps': ag._flat['steps'].cpu()}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_fused_sac_step_two_ranks_equal_one_process_on_the_joint_batch(tmp_path):
    """scg_sac_update split at its two gradient exchanges (SCG_SAC_ACTOR_GRAD / CRITIC_GRAD / FINISH, sac.py::_fused_step_dp):
    two ranks on one GPU, each with its own replay shard, minibatch and noise, the flat gradient all-reduced (gloo here, RCCL
    on a node) — both ranks end with the same parameters, and those are the parameters ONE process reaches on the joint
    minibatch (mean over 2 B = mean of the per-rank means)."""
    import torch.multiprocessing as mp
    from safe_control_gym_amd.sac import DeviceReplay, SACAgent, SACConfig
    mp.spawn(_sac_dp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(str(tmp_path / f'rank{r}.pt')) for r in range(2))
    assert torch.equal(a['params'], b['params']) and torch.equal(a['targ'], b['targ'])              # lock-step
    assert a['adam_steps'].tolist() == [2.0, 2.0, 2.0]
    dev = torch.device('cuda', 0)
    low, high = -torch.ones(4, device=dev), torch.ones(4, device=dev)
    one = SACAgent(24, 4, low, high, SACConfig(hidden_dim=128, activation='relu', use_entropy_tuning=True), dev)
    one.ac.load_state_dict(a['init']); one.ac_targ.load_state_dict(a['init'])
    cap = a['data']['obs'].shape[0]
    buf = DeviceReplay(2 * cap, 24, 4, dev)
    for d in (a, b):
        t = {k: v.to(dev) for k, v in d['data'].items()}
        buf.push(t['obs'], t['act'], t['rew'], t['next_obs'], t['mask'])
    for k in range(2):
        idx = torch.cat([a['steps'][k]['idx'], b['steps'][k]['idx'] + cap]).to(dev)
        eps = torch.cat([a['steps'][k]['eps'], b['steps'][k]['eps']]).to(dev).contiguous()
        eps2 = torch.cat([a['steps'][k]['eps2'], b['steps'][k]['eps2']]).to(dev).contiguous()
        F = one._fused_args(buf, idx.numel(), idx=idx, eps=eps, eps_next=eps2)
        one._fused_step(F)
    torch.cuda.synchronize()
    p1, p2 = one._flat['p'].cpu(), a['params']
    assert (p1 - p2).abs().max() <= 2.1e-3 and (p1 - p2).abs().mean() <= 2e-5, ((p1 - p2).abs().max(), (p1 - p2).abs().mean())
    assert (one._flat['targ'].cpu() - a['targ']).abs().max() <= 1e-4
    init_flat_moved = (p2[:-1] - one._flat['p'].cpu()[:-1]).abs().max() < 1.0 and (a['params'] - b['params']).abs().max() == 0
    assert init_flat_moved


def test_rccl_backend_initialises_and_carries_the_collectives_with_one_rank():
    """What a single GPU can show of the `nccl` (= RCCL) branch that the driver's N > 1 runs take: the backend initialises on this box
    (librccl loads, the dmabuf IPC mode is accepted), and the three collectives the learning legs use — all-reduce (gradients), broadcast
    (parameters, stop flag), barrier — run on device tensors through `parallel.py`'s own helpers.  One rank: no peer traffic, but every
    call goes through RCCL's code path instead of the gloo staging the other tests of this file use."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from safe_control_gym_amd import parallel
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
assert dist.get_backend() == 'nccl' and not parallel._through_host(torch.zeros(1, device='cuda'))
g = torch.arange(36742, dtype=torch.float32, device='cuda')            # the PPO pair's flat gradient + approx-KL slot
dist.all_reduce(g); parallel.broadcast_(g, 0); dist.barrier(); torch.cuda.synchronize()
assert float(g[-1]) == 36741.0
m = torch.nn.Linear(4, 3).cuda(); parallel.broadcast_parameters([m])
b = parallel.FlatBucket(list(m.parameters()), n_scalars=1)
for p in m.parameters(): p.grad = torch.ones_like(p)
b.pack([torch.tensor(0.5, device='cuda')]); b.all_reduce_mean(); assert float(b.unpack()[0]) == 0.5
dist.destroy_process_group()
print('RCCL_ONE_RANK_OK')
'''
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    res = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and 'RCCL_ONE_RANK_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


def test_graph_captured_data_parallel_epoch_over_rccl_with_one_rank():
    """The data-parallel PPO optimiser loop as the driver's N > 1 runs execute it over RCCL — per epoch ONE HIP-graph replay of
    n_mb x (gradient kernel, reduction, flat all-reduce, gated Adam reading the sum x 1 / world) — on one GPU: a one-rank `nccl`
    process group with `force_data_parallel`, three iterations (eager warm-up, capture, replay), must leave bit for bit the weights
    of the single-rank scg_ppo_grad + scg_adam_gated loop; and the probe behind bench.py's `multi_gpu` object runs."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from safe_control_gym_amd import parallel
from safe_control_gym_amd.ppo import PPO, PPOConfig
from safe_control_gym_amd.registration import load_task
from safe_control_gym_amd.vec_env import HipVecEnv
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
assert parallel.collectives_capturable()
env_id, cfg = load_task('quadrotor_2D_track')
def run(extra):
    env = HipVecEnv(env_id, 2048, seed=3, return_numpy=False, policy=(128, 'tanh'), **cfg)
    torch.manual_seed(0)
    ppo = PPO(env, PPOConfig(hidden_dim=128, activation='tanh', use_gae=True, rollout_batch_size=2048, rollout_steps=16,
                             mini_batch_size=4096, opt_epochs=2, target_kl=0.03, extra=extra), seed=0)
    for _ in range(3):
        res = ppo.train_step()
    torch.cuda.synchronize()
    p = ppo.agent._flat['p'].clone()
    path = ppo.agent.dp_path
    env.close()
    return p, path, res
p_ref, _, r_ref = run({'fused_step': False})
p_dp, path, r_dp = run({'force_data_parallel': True})
assert path and path.startswith('one graph replay per epoch'), path
assert torch.equal(p_ref, p_dp), float((p_ref - p_dp).abs().max())
assert r_ref['minibatches'] == r_dp['minibatches'] == 16 and abs(r_ref['approx_kl'] - r_dp['approx_kl']) < 1e-6
probe = parallel.allreduce_probe()
assert probe['world'] == 1 and probe['150528']['eager_us'] > 0 and probe['150528']['captured_us'] > 0, probe
print('PROBE', probe)
dist.destroy_process_group()
print('DP_GRAPH_OK')
'''
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    res = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'DP_GRAPH_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]


def test_graph_captured_data_parallel_sac_steps_over_rccl_with_one_rank():
    """SAC's data-parallel gradient steps (actor phase -> all-reduce -> critic phase -> all-reduce -> finish, sac.py::_fused_step_dp) as ONE
    HIP-graph replay per vector step over RCCL: on a one-rank `nccl` group with `force_data_parallel`, four updates of 4 gradient steps
    each (eager warm-up, capture, replays) must leave bit for bit the parameters, target networks and Adam moments of the same steps
    enqueued one by one (`graph_collectives: False`)."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from safe_control_gym_amd import parallel
from safe_control_gym_amd.sac import DeviceReplay, SACAgent, SACConfig
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
dev = torch.device('cuda', 0)
low, high = -torch.ones(4, device=dev), torch.ones(4, device=dev)
g = torch.Generator(device='cpu').manual_seed(5)
n = 8192
data = [torch.randn(n, 24, generator=g), torch.rand(n, 4, generator=g) * 2 - 1, torch.randn(n, generator=g), torch.randn(n, 24, generator=g),
        (torch.rand(n, 1, generator=g) > 0.05).float()]
def run(extra):
    torch.manual_seed(3)
    ag = SACAgent(24, 4, low, high, SACConfig(hidden_dim=128, activation='relu', use_entropy_tuning=True, extra=dict(extra, force_data_parallel=True)), dev)
    assert ag.use_fused
    buf = DeviceReplay(n, 24, 4, dev)
    buf.push(*[t.to(dev) for t in data])
    for _ in range(4):
        res = ag.update_from_buffer(buf, 1024, 4)
    torch.cuda.synchronize()
    fl = ag._flat
    return {k: fl[k].clone() for k in ('p', 'targ', 'm', 'v', 'steps')}, ag.dp_path, res
a, path_a, _ = run({'graph_collectives': False})
b, path_b, res = run({})
assert path_a == 'eager' and path_b.startswith('one graph replay per vector step'), (path_a, path_b)
for k in a:
    assert torch.equal(a[k], b[k]), (k, float((a[k] - b[k]).abs().max()))
assert all(map(lambda v: v == v, res.values()))
dist.destroy_process_group()
print('SAC_DP_GRAPH_OK')
'''
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    res = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and 'SAC_DP_GRAPH_OK' in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
