#!/usr/bin/env python3
"""bench.py — env-steps/s of the HIP simulator on BASELINE.json's headline config.

Workload (config.workload): Quadrotor2D trajectory tracking (the env of BASELINE configs[2];
examples/rl/config_overrides/quadrotor_2D/quadrotor_2D_track.yaml), 65 536 envs per GPU, float32.
One "step" = one control step of every env = ONE launch of the fused step kernel (action
pre-processing, 20 engine substeps, observation / reward / done / info / 16 constraint rows,
episode statistics, auto-reset), driven by synthetic actions ~U(-1,1) already resident in HBM
(the reference's own README benchmark is the same open-loop random-action loop on one env).
The K timed launches are replayed from a HIP graph so the measurement is not Python-bound.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches one rank
per GPU with torch.distributed.run.  Rank 0 prints ONE JSON line.  Env shards are independent
(rank r owns global env ids [r*N, (r+1)*N)), there is no data-path collective: scaling = weak.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# SURVEY.md §8d algorithmic bytes per env-step (fp32): read state 24 + action 8 + counter 4;
# write state 24 + obs 48 + reward 4 + done 1 + flags 1 (trunc+violation+oob packed; survey counts 1+1)
# + counter 4 + c_values 64 + mse 4  => 187 B with the survey's accounting.
ALGO_BYTES_PER_ENV_STEP = {'quadrotor_2D_track': 187, 'cartpole_stab': 111, 'quadrotor_3D_track': 363,
                           'quadrotor_3D_track_disturbed': 379}
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20000)
    ap.add_argument('--warmup', type=int, default=2000)
    ap.add_argument('--envs', type=int, default=65536, help='envs per GPU')
    ap.add_argument('--task', default='quadrotor_2D_track')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f64'])
    ap.add_argument('--no-graph', action='store_true', help='launch every step from Python instead of a HIP graph')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--generic', action='store_true', help='use libscg_hip.so (any config, parameters via LDS) instead of the config-specialised build')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU oracle baseline')
    ap.add_argument('--graph-len', type=int, default=1000, help='steps captured per HIP graph')
    return ap.parse_args()


def usable_cores():
    """Host cores this process may really use: os.cpu_count() capped by the affinity mask and the cgroup CPU quota
    (the GPU box reports 256 logical CPUs but runs the container with a 16-CPU quota: 256 OpenMP threads on it are slower
    than 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(task, cfg, env_id, budget_s, n_envs):
    """CPU path timed on this box's host cores, same workload (task config, env count, random actions):
    oracle/scg_oracle.c — the double-precision C restatement of the control step (validated against the NumPy oracle,
    which is pinned to the reference's own Python) — with OpenMP over all cores, for ~budget_s seconds."""
    import numpy as np
    from oracle.c_port import CPort
    from oracle.envs import make_oracle_env, make_rng
    from oracle.vec import OracleVecEnv
    cores = usable_cores()
    os.environ.setdefault('OMP_NUM_THREADS', str(cores))
    cores = int(os.environ['OMP_NUM_THREADS'])
    small = make_oracle_env(env_id, 8, make_rng('philox', 8, 42), **cfg)            # constants / tables only
    small.num_envs = n_envs
    port = CPort(small, seed=42)
    port.lib.oc_set_threads(cores)          # (torch initialised the OpenMP runtime before OMP_NUM_THREADS was set here)
    cores = int(port.lib.oc_get_threads())
    port.reset()
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(8, n_envs, small.action_dim))
    port.run(acts, 2)                                                               # warm-up (threads, caches)
    # K control steps per call: every thread advances its own envs through all K steps (no barrier per step)
    K, steps = 50, 0
    t0 = time.perf_counter()
    while True:
        port.run(acts, K)
        steps += K
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    c_rate = n_envs * steps / el
    # for scale: the NumPy oracle (the reference's per-env Python structure, batched), 1 core, 2 s
    n_np = 1024
    onp = OracleVecEnv(make_oracle_env(env_id, n_np, make_rng('philox', n_np, 42), **cfg))
    onp.reset()
    k, t1 = 0, time.perf_counter()
    while time.perf_counter() - t1 < 2.0:
        onp.step(rng.uniform(-1, 1, size=(n_np, small.action_dim)))
        k += 1
    np_rate = n_np * k / (time.perf_counter() - t1)
    return {'value': c_rate, 'unit': 'env-steps/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle/scg_oracle.c (float64 C restatement of the control step, gcc -O3 -fopenmp, {cores} threads, '
                      f'each thread runs its envs {K} steps per parallel region), '
                      f'{n_envs} envs x {steps} control steps of {task} with random actions in {el:.1f} s; '
                      f'NumPy oracle on 1 core: {np_rate:.3g} env-steps/s; reference README (1 PyBullet env, '
                      f'i7-1068NG7): 381-464 env-steps/s'}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # SCG_BENCH_BACKEND=gloo lets the N>1 control flow be exercised on a box with fewer GPUs than ranks
    # (ranks then share devices round-robin); the driver's runs use nccl (= RCCL), one rank per GPU.
    backend = os.environ.get('SCG_BENCH_BACKEND', 'nccl')
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print(f'[bench] warning: --gpus {args.gpus} but WORLD_SIZE {world}', file=sys.stderr)
    dev = torch.device('cuda', torch.cuda.current_device())
    dtype = torch.float32 if args.dtype == 'f32' else torch.float64
    env_id, cfg = load_task(args.task)
    if os.environ.get('SCG_BENCH_OVERRIDE'):      # dev only: ablations of the task config
        cfg.update(json.loads(os.environ['SCG_BENCH_OVERRIDE']))
    N = args.envs
    env = HipVecEnv(env_id, N, seed=1337, dtype=dtype, env_id_offset=rank * N, return_numpy=False,
                    specialize=False if args.generic else 'auto', **cfg)
    nu = env.spec.nu
    # synthetic actions resident in HBM: a ring of pre-generated batches
    ring = 64
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    actions = (torch.rand(ring, N, nu, device=dev, dtype=dtype, generator=gen) * 2 - 1)
    env.reset_tensors()
    # outputs a rollout collector consumes (obs, reward, done, flags, constraint values, mse, terminal obs,
    # fused episode statistics); the optional debugging outputs (env.state copy, noisy action) are not bound
    lean_out, lean_c = env.bind_outputs(state=None, noisy_action=None)

    def run_steps(k0, k):
        for t in range(k0, k0 + k):
            env.step_tensors(actions[t % ring], out=lean_out, c_out=lean_c)

    G = max(1, min(args.graph_len, args.steps))
    graph = None
    if not args.no_graph:
        # warm up on a side stream, then capture G consecutive control steps
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run_steps(0, 8)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run_steps(0, G)

    def do(k):
        if graph is None:
            run_steps(0, k)
            return k
        reps = (k + G - 1) // G
        for _ in range(reps):
            graph.replay()
        return reps * G

    do(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    done_steps = do(args.steps)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / done_steps          # avg launch-to-launch period of the step kernel
    el = torch.tensor([elapsed], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    # sanity: the simulator really advanced (episodes finished, finite rewards)
    ok = bool(torch.isfinite(lean_out.reward).all().item()) and int(lean_out.fin_length.max().item()) > 0
    total_env_steps = world * N * done_steps
    value = total_env_steps / elapsed
    if rank == 0:
        algo = ALGO_BYTES_PER_ENV_STEP.get(args.task)
        if dtype == torch.float64 and algo:
            algo = None
        achieved = (algo * N / (kernel_ms * 1e-3)) / 1e9 if algo else None
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this very command
        # (profiles/r01_hbm_traffic.json: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 fetch correction)
        traffic = None
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01_hbm_traffic.json')) as f:
                traffic = json.load(f).get(f'{args.task}/{args.dtype}/{N}', {}).get('traffic_bytes_per_launch')
        except OSError:
            pass
        out = {
            'metric': 'env-steps/sec (whole node), Quadrotor2D-track', 'value': value, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': done_steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / done_steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic',
            'config': {'workload': f'{args.task}: {N} envs/GPU x {world} GPU, 1 launch of the fused step kernel per '
                                   f'control step (20 engine substeps, obs/reward/done/info/constraints, auto-reset), '
                                   f'synthetic U(-1,1) actions resident in HBM, '
                                   f'{"HIP graph of %d steps" % G if graph is not None else "per-step Python launches"}',
                       'envs_per_gpu': N, 'task_yaml': f'safe_control_gym_amd/configs/{args.task}.yaml',
                       'parallelism': f'env-shard x{world}', 'finite_outputs': ok,
                       'kernel_build': 'config-specialised' if env.specialized else 'generic',
                       'ppo_wall_clock_to_reward': 'measured separately: examples/train_ppo.py, profiles/r01_ppo_wallclock_q2track.md'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': (achieved / HBM_PEAK_GBS) if achieved else None, 'traffic': traffic,
                         'traffic_source': 'profiles/r01_hbm_traffic.json (rocprofv3 --pmc, bytes per launch)' if traffic else None,
                         'kernel': 'step_kernel<QUAD_2D,float>', 'avg_launch_us': kernel_ms * 1e3,
                         'algorithmic_bytes_per_env_step': algo},
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args.task, cfg, env_id, args.cpu_seconds, N)
            out['cpu_baseline']['gpu_over_cpu'] = value / out['cpu_baseline']['value']
        print(json.dumps(out))
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
