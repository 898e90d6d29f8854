#!/usr/bin/env python3
"""bench.py — the whole BASELINE.json metric in one JSON line: env-steps/s of the HIP simulator on the headline config
+ PPO wall-clock-to-reward on it, with the per-kernel roofline and the CPU path timed beside it.

Headline (the top-level fields): Quadrotor2D trajectory tracking (BASELINE configs[2]; examples/rl/config_overrides/
quadrotor_2D/quadrotor_2D_track.yaml), 65 536 envs per GPU, float32.  One "step" = one control step of every env = ONE
launch of the fused step kernel (action pre-processing, 20 engine substeps, observation / reward / done / info / 16
constraint rows, episode statistics, auto-reset), synthetic actions ~U(-1,1) resident in HBM (the reference's own README
benchmark is the same open-loop random-action loop on one env).  The K timed launches are one HIP graph; when K is small
the K-step graph is replayed R times, each replay timed between barrier + synchronize, and the MEDIAN replay is reported
(`config.timing`), so that K = 20 is not a single 0.1 ms sample.

Secondary objects (rank 0, N = 1): `f64` (the float64 kernels of the same workload), `secondary` (the other BASELINE
configs' env kernels: cartpole_stab incl. the fused random-action rollout of config #2, quadrotor_3D_track[_disturbed]),
`gae` (scg_gae timing + its own roofline), `sequence` (scg_step_sequence: K control steps per launch, the mode that carries >= 0.40 of the
HBM roofline at this N), `fused_rollout` (K steps per launch with the PPO actor in the loop), `ppo` (budgeted wall-clock-to-reward runs at
BASELINE config #3's batch: 3 partial epochs x 16 minibatches of 16 256 per iteration, with `ppo.full_epochs` and `ppo.envs_16384` beside
it), `sac` (config #5's env; `sac.param_randomised` = with flyable parameter disturbances, target re-measured under them), `cpu_baseline`.
`roofline.traffic`, `roofline.valu_issue`, `f64.traffic`, `sequence.*.traffic*` are quoted from profiles/r06_hbm_traffic.json (rocprofv3
--pmc passes) only while that file names the hash of the kernel sources in this tree.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 the driver launches one rank per GPU with
torch.distributed.run.  Rank 0 prints ONE JSON line.  Env shards are independent (rank r owns global env ids
[r*N, (r+1)*N)), there is no data-path collective: scaling = weak.
"""
import argparse
import json
import os
import statistics
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
KERNEL_NAME = {'quadrotor_2D_track': 'step_kernel<QUAD_2D,float>', 'cartpole_stab': 'step_kernel<CARTPOLE,float>',
               'quadrotor_3D_track': 'step_kernel<QUAD_3D,float>', 'quadrotor_3D_track_disturbed': 'step_kernel<QUAD_3D,float,DIST>'}
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# HBM traffic / executed-instruction counts come from committed rocprofv3 --pmc passes (tools/profile_round5.sh ->
# tools/profile_post.py).  They describe ONE build of the kernels: the file carries the hash of the kernel sources it was measured
# on, and a line printed from other sources drops the number (traffic: null, with the reason) instead of quoting a stale one.
TRAFFIC_FILE = 'r06_hbm_traffic.json'
CHAIN_FILE = 'r06_chain_latency.json'
SEQ_K, SEQ_K_ALL = 32, 8        # control steps per launch of the two scg_step_sequence workloads (sequence_leg, tools/seq_profile.py)
SHADER_CLOCK_GHZ = 2.4          # MI355X_MICROARCH.md
# scg_kernels.hip defaults (scg_set_step_launch): which launch geometry of the step kernel a shard of N envs takes
LAUNCH_WIDE_MIN = int(os.environ.get('SCG_WIDE_MIN_ENVS', 8388608))
LAUNCH_WSBACK = (int(os.environ.get('SCG_WSBACK_MIN_ENVS', 131072)), int(os.environ.get('SCG_WSBACK_MAX_ENVS', 524288)))


def launch_geometry(n, specialised, task='quadrotor_2D_track'):
    if not specialised:
        return 'step_kernel (generic library: 256-thread workgroups, parameters staged in LDS)'
    if LAUNCH_WSBACK[0] <= n <= LAUNCH_WSBACK[1] and n < LAUNCH_WIDE_MIN and not task.startswith('cartpole'):
        return 'step_wsback_kernel (one wave per 64 envs, workspace arrays stored write-back)'
    if n >= LAUNCH_WIDE_MIN:
        return 'step_wide_kernel (256-thread workgroups)'
    return 'step_kernel (one wave per 64 envs, one-wave workgroups)'


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
    ap.add_argument('--no-secondary', action='store_true', help='headline only (no f64 / other configs / gae / ppo legs)')
    ap.add_argument('--generic', action='store_true', help='use libscg_hip.so (any config, parameters via LDS) instead of the config-specialised build')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU oracle baseline')
    ap.add_argument('--graph-len', type=int, default=1000, help='steps captured per HIP graph')
    ap.add_argument('--ppo-seeds', type=int, default=3, help='seeds of the PPO wall-clock-to-reward leg (0 = skip)')
    ap.add_argument('--ppo-seconds', type=float, default=10.0, help='budget per seed')
    ap.add_argument('--ppo-envs', type=int, default=65536, help='envs per GPU of the PPO leg (BASELINE config #3: 65 536)')
    ap.add_argument('--sac-seeds', type=int, default=3, help='seeds of the SAC wall-clock-to-reward leg on config #5 (0 = skip)')
    ap.add_argument('--sac-seconds', type=float, default=40.0, help='budget per seed')
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


_PMC_CACHE = {}


def pmc_entry(key):
    """Entry `key` of profiles/r06_hbm_traffic.json if that file was measured on THESE kernel sources, else (None, why)."""
    if 'file' not in _PMC_CACHE:
        try:
            with open(os.path.join(ROOT, 'profiles', TRAFFIC_FILE)) as f:
                _PMC_CACHE['file'] = json.load(f)
        except (OSError, ValueError):
            _PMC_CACHE['file'] = None
    d = _PMC_CACHE['file']
    if d is None:
        return None, f'profiles/{TRAFFIC_FILE} not present'
    from safe_control_gym_amd import _lib
    have, want = d.get('_meta', {}).get('source_hash'), f'0x{_lib.source_hash():016x}'
    if have != want:
        return None, f'profiles/{TRAFFIC_FILE} was measured on kernel sources {have}, this tree is {want}: dropped'
    e = d.get(key)
    return (e, f'profiles/{TRAFFIC_FILE} (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes, per launch; kernel sources {want})') if e \
        else (None, f'profiles/{TRAFFIC_FILE} has no entry {key}')


def traffic_of(task, dtype, n):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this very command (separate FETCH_SIZE / WRITE_SIZE
    passes, gfx950 x2 fetch correction — tools/profile_round5.sh, tools/profile_post.py)."""
    e, src = pmc_entry(f'{task}/{dtype}/{n}')
    return (e['traffic_bytes_per_launch'] if e else None), src


# Issue intervals measured on MI355X with tools/issue_rate.hip (profiles/r05_issue_rate.txt, 2.396 GHz): ONE wave issues a vector
# instruction every 4.8 clocks at best and a DEPENDENT one every 8.4; the SIMD issues every 2.4 clocks when it has two or more waves to
# choose from (4.3 for packed-fp32 and integer-multiply instructions).  (The guide's "2 clocks per wave64 instruction" is the SIMD's
# rate; round 4's "4 clocks" was one wave's.)
WAVE_ISSUE_CLOCKS, SIMD_ISSUE_CLOCKS, DEPENDENT_ISSUE_CLOCKS, MEASURED_CLOCK_GHZ = 4.8, 2.4, 8.4, 2.396


def valu_issue_of(task, dtype, n, period_us):
    """What the launch period owes to vector-instruction ISSUE: executed VALU instructions per wave (SQ_INSTS_VALU / SQ_WAVES of the PMC
    passes) x the issue interval that applies — with one wave per SIMD the WAVE's own limit (4.8 clocks per instruction: a lower bound of
    the wave's lifetime, dependencies come on top — see `chain_latency`), with W >= 2 waves per SIMD the larger of that and the SIMD's
    W x 2.4 clocks per instruction."""
    e, _ = pmc_entry(f'{task}/{dtype}/{n}')
    if not e or not e.get('valu_instructions_per_wave'):
        return None
    waves_per_simd = max(1.0, (e.get('waves_per_launch') or n / 64) / 1024.0)        # 256 CUs x 4 SIMDs
    per_instr = max(WAVE_ISSUE_CLOCKS, waves_per_simd * SIMD_ISSUE_CLOCKS)
    issue_us = e['valu_instructions_per_wave'] * per_instr / (MEASURED_CLOCK_GHZ * 1e3)
    return {'valu_instructions_per_wave': e['valu_instructions_per_wave'], 'waves_per_simd': waves_per_simd, 'clocks_per_instruction': per_instr,
            'issue_us': issue_us, 'frac': issue_us / period_us, 'source': 'profiles/r05_issue_rate.txt (tools/issue_rate.hip)'}


def chain_latency_of(task, period_us):
    """Serial dependent-instruction chain of the engine-substep loop (tools/chain_latency.py: the built kernel's loop body through the
    in-order model of tools/isa_sim.py with the measured issue / dependent-issue intervals): the bound of kernels whose control step is
    tens of substeps of a dependent chain, where neither the HBM roofline nor the issue rate explains the launch."""
    try:
        with open(os.path.join(ROOT, 'profiles', CHAIN_FILE)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    from safe_control_gym_amd import _lib
    if d.get('_meta', {}).get('source_hash') != f'0x{_lib.source_hash():016x}':
        return {'dropped': f'profiles/{CHAIN_FILE} was computed from other kernel sources'}
    e = d.get(task)
    if not e or 'chain_us_per_control_step' not in e:
        return e
    return {'substep_loop_us': e['chain_us_per_control_step'], 'frac_of_launch': e['chain_us_per_control_step'] / period_us,
            'issue_limit_us': e['issue_limit_us_per_control_step'], 'dependent_instructions_per_substep': e['dependent_instructions_per_substep'],
            'how': 'tools/chain_latency.py (static in-order model, measured intervals)'}


class StepBench:
    """K control steps of one env batch as a HIP graph of step-kernel launches on ring-buffered synthetic actions."""

    def __init__(self, torch, task, n, dtype, rank=0, generic=False, graph_len=1000, use_graph=True):
        from safe_control_gym_amd.registration import load_task
        from safe_control_gym_amd.vec_env import HipVecEnv
        self.torch, self.task, self.n = torch, task, n
        dev = torch.device('cuda', torch.cuda.current_device())
        env_id, cfg = load_task(task)
        if os.environ.get('SCG_BENCH_OVERRIDE'):      # dev only: ablations of the task config
            cfg.update(json.loads(os.environ['SCG_BENCH_OVERRIDE']))
        self.env_id, self.cfg = env_id, cfg
        self.env = HipVecEnv(env_id, n, seed=1337, dtype=dtype, env_id_offset=rank * n, return_numpy=False,
                             specialize=False if generic else 'auto', **cfg)
        ring = 64
        gen = torch.Generator(device=dev)
        gen.manual_seed(1234 + rank)
        self.actions = torch.rand(ring, n, self.env.spec.nu, device=dev, dtype=dtype, generator=gen) * 2 - 1
        self.ring = ring
        self.env.reset_tensors()
        # outputs a rollout collector consumes (obs, reward, done, flags, constraint values, mse, terminal obs, fused episode
        # statistics); the optional debugging outputs (env.state copy, noisy action) are not bound
        self.out, self.c_out = self.env.bind_outputs(state=None, noisy_action=None)
        self.G = graph_len
        self.graph = None
        if use_graph:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self._run(8)
            torch.cuda.current_stream().wait_stream(s)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._run(self.G)

    def _run(self, k):
        for t in range(k):
            self.env.step_tensors(self.actions[t % self.ring], out=self.out, c_out=self.c_out)

    def do(self, k):
        if self.graph is None:
            self._run(k)
            return k
        reps = (k + self.G - 1) // self.G
        for _ in range(reps):
            self.graph.replay()
        return reps * self.G

    def kernel_period_us(self, k):
        """Average launch-to-launch period of the step kernel over >= k launches (HIP events on the launch stream)."""
        torch = self.torch
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        done = self.do(k)
        ev1.record()
        torch.cuda.synchronize()
        return 1e3 * ev0.elapsed_time(ev1) / done

    def sane(self):
        return bool(self.torch.isfinite(self.out.reward).all().item()) and int(self.out.fin_length.max().item()) > 0


def roofline_of(task, dtype_name, n, period_us, driver_step_us=None, geometry=None):
    """`frac` = algorithmic bytes per launch / the launch period measured HERE with HIP events inside back-to-back graph replays.
    `frac_by_clock` puts the other two clocks beside it so that no reader takes one for another: the rocprofv3 --kernel-trace average
    duration of the same kernel (committed under profiles/, quoted while the kernel-source hash matches; the tracer adds ~0.4 us per
    dispatch) and the driver-timed step of this very run (`ms_per_step`: K = 20 steps per graph replay pay the replay floor)."""
    algo = ALGO_BYTES_PER_ENV_STEP.get(task) if dtype_name == 'f32' else None
    achieved = (algo * n / (period_us * 1e-6)) / 1e9 if algo else None
    traffic, src = traffic_of(task, dtype_name, n)
    e, _ = pmc_entry(f'{task}/{dtype_name}/{n}')
    rocprof_us = e.get('rocprof_avg_launch_us') if e else None
    frac_of = lambda us: (algo * n / (us * 1e-6)) / 1e9 / HBM_PEAK_GBS if (algo and us) else None      # noqa: E731
    return {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': (achieved / HBM_PEAK_GBS) if achieved else None, 'traffic': traffic, 'traffic_source': src,
            'frac_by_clock': {'in_graph_hip_events': {'us': period_us, 'frac': frac_of(period_us)},
                              'rocprofv3_kernel_trace_avg': {'us': rocprof_us, 'frac': frac_of(rocprof_us)},
                              'driver_timed_step': {'us': driver_step_us, 'frac': frac_of(driver_step_us)}},
            'kernel': KERNEL_NAME.get(task, 'step_kernel') + (f' launched as {geometry}' if geometry else ''),
            'avg_launch_us': period_us, 'algorithmic_bytes_per_env_step': algo,
            'valu_issue': valu_issue_of(task, dtype_name, n, period_us)}


def secondary_env_kernels(torch, n):
    """The env kernels of the other BASELINE configs (same 65 536 envs, f32): us per launch and roofline fraction."""
    out = {}
    for task in ('cartpole_stab', 'quadrotor_3D_track', 'quadrotor_3D_track_disturbed'):
        try:
            b = StepBench(torch, task, n, torch.float32, graph_len=500)
            b.do(500)
            us = b.kernel_period_us(3000)
            r = roofline_of(task, 'f32', n, us)
            e = {'avg_launch_us': us, 'env_steps_per_s': n / (us * 1e-6), 'frac': r['frac'], 'algorithmic_bytes_per_env_step': r['algorithmic_bytes_per_env_step'],
                 'traffic': r['traffic'], 'valu_issue': r['valu_issue'], 'chain_latency': chain_latency_of(task, us), 'kernel_build': 'config-specialised' if b.env.specialized else 'generic', 'finite_outputs': b.sane()}
            if task == 'cartpole_stab':         # BASELINE config #2: in-kernel random actions, K steps per launch
                K = 1000
                b.env.rollout_random(K)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    b.env.rollout_random(K)
                torch.cuda.synchronize()
                e['rollout_random_env_steps_per_s'] = 3 * K * n / (time.perf_counter() - t0)
            b.env.close()
            out[task] = e
        except Exception as exc:                                    # noqa: BLE001  (a secondary never sinks the headline)
            out[task] = {'error': repr(exc)[:200]}
    return out


def gae_leg(torch):
    """scg_gae (compute_returns_and_advantages): 16 B read + 8 B written per (t, env) + the in-place reward update 8 B."""
    from safe_control_gym_amd.rollout import gae_returns
    out = {}
    for T, N in ((32, 65536), (1000, 4)):
        rew, v = torch.rand(T, N, device='cuda'), torch.rand(T, N, device='cuda')
        mask = (torch.rand(T, N, device='cuda') > 0.01).float()
        tv, last = torch.rand(T, N, device='cuda'), torch.rand(N, device='cuda')
        bufs = (torch.empty_like(rew), torch.empty_like(rew))      # caller-owned outputs: the kernel's own time
        for _ in range(3):
            gae_returns(rew, v, mask, tv, last, out=bufs)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        reps = 50
        ev0.record()
        for _ in range(reps):
            gae_returns(rew, v, mask, tv, last, out=bufs)
        ev1.record()
        torch.cuda.synchronize()
        us = 1e3 * ev0.elapsed_time(ev1) / reps
        byts = 32 * T * N + 4 * N
        out[f'{T}x{N}'] = {'us_per_call': us, 'algorithmic_bytes': byts, 'achieved_GBs': byts / (us * 1e-6) / 1e9,
                           'frac': byts / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                           'note': 'pre-allocated outputs (gae_returns(out=...))' if T * N > 100000 else 'launch-bound (4 envs: wave segmented scan over time)'}
    return out


def sequence_leg(torch, n, K=SEQ_K, task='quadrotor_2D_track', K_all=SEQ_K_ALL):
    """The same control steps with K of them per launch (scg_step_sequence): caller-supplied action sequences resident in
    HBM, per-step outputs written to [K]-stacked arrays, state in registers between steps.  Two output sets: what a PPO-style
    collector keeps (obs, reward, done, flags; terminal observation where done) and every output of scg_step (+ mse,
    constraint values, state, noisy action, finished-episode statistics).  Bytes are this kernel's own: the action read and
    the outputs per env-step; state, counters and running episode statistics move once per launch."""
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = load_task(task)
    env = HipVecEnv(env_id, n, seed=7, return_numpy=False, **cfg)
    spec = env.spec
    res = {'envs': n}
    base = 4 * spec.nu + 4 * spec.obs_dim + 4 + 1 + 1                       # action in; obs, reward, done, flags out
    extra = 4 + 4 * len(spec.con_rows) + 4 * spec.nx + 4 * spec.nu          # mse, constraint values, state, noisy action
    per_launch = 2 * (4 * env._n_state_arrays() + 8) + 32                   # state + counters in and out, episode statistics RMW
    # (K per variant: the [K]-stacked outputs of one launch should stay within the 256 MB Infinity Cache, like the 12 MB
    #  set the per-step launches overwrite; 32 steps of every output are 457 MB and stream to HBM at 5.2 us per step)
    for tag, kw, per_step, K in (('collector_outputs', dict(terminal_obs=True), base, K),
                                 ('all_outputs', dict(terminal_obs=True, mse=True, c_values=True, fin_stats=True, state=True, noisy_action=True),
                                  base + extra, K_all)):
        acts = torch.rand(K, n, spec.nu, device=env.device) * 2 - 1
        env.reset_tensors()
        out = env.step_sequence(acts, **kw)
        for _ in range(3):
            env.step_sequence(acts, out=out)
        reps = 40
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            env.step_sequence(acts, out=out)
        ev[1].record()
        torch.cuda.synchronize()
        us = 1e3 * ev[0].elapsed_time(ev[1]) / reps
        bytes_es = per_step + per_launch / K
        rate = n * K / (us * 1e-6)
        res[tag] = {'steps_per_launch': K, 'us_per_control_step': us / K, 'env_steps_per_s': rate, 'algorithmic_bytes_per_env_step': bytes_es,
                    'stacked_output_MB_per_launch': sum(t.numel() * t.element_size() for t in out.values()) / 1e6,
                    'frac': rate * bytes_es / 1e9 / HBM_PEAK_GBS,
                    'frac_on_per_step_bytes': rate * ALGO_BYTES_PER_ENV_STEP.get(task, 0) / 1e9 / HBM_PEAK_GBS,
                    'finite_outputs': bool(torch.isfinite(out['obs']).all() and torch.isfinite(out['reward']).all())}
        pm, src = pmc_entry(f'sequence_{tag.split("_")[0]}/f32/{n}')           # step_sequence_kernel under rocprofv3 (tools/seq_profile.py)
        res[tag]['traffic'] = pm['traffic_bytes_per_launch'] if pm else None
        res[tag]['traffic_bytes_per_env_step'] = pm['traffic_bytes_per_launch'] / (n * K) if pm else None
        res[tag]['rocprof_avg_launch_us'] = pm.get('rocprof_avg_launch_us') if pm else None
        res[tag]['traffic_source'] = src
        del out
    env.close()
    res['note'] = ('parity: tests/test_gpu_sequence.py (bit-identical to K x scg_step); the headline above stays one launch per control '
                   'step.  frac counts the bytes THIS kernel moves; frac_on_per_step_bytes applies SURVEY 8d / BASELINE.md 3\'s definition '
                   '(per-step algorithmic bytes x env-steps/s) to this rate')
    return res


def fused_rollout_leg(torch, n, T=32):
    """K control steps per launch with the PPO actor (12 -> 128 -> 128 -> 2, tanh, exact f32 MFMA) inside the env kernel."""
    from safe_control_gym_amd.ppo import PPO, PPOConfig
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    env_id, cfg = load_task('quadrotor_2D_track')
    env = HipVecEnv(env_id, n, seed=7, return_numpy=False, policy=(128, 'tanh'), **cfg)
    ppo = PPO(env, PPOConfig(hidden_dim=128, activation='tanh', use_gae=True, rollout_batch_size=n, rollout_steps=T, mini_batch_size=65536), seed=7)
    if not ppo._fused_rollout:
        return {'error': 'fused rollout unavailable'}
    for _ in range(2):
        ppo._collect_fused()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        ppo._collect_fused()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / reps
    env.close()
    pm, src = pmc_entry(f'rollout_policy/f32/{n}')             # rollout_policy_kernel under rocprofv3 (tools/seq_profile.py)
    return {'envs': n, 'rollout_steps': T, 'ms_per_rollout': 1e3 * el, 'env_steps_per_s': n * T / el,
            'what': 'scg_rollout_policy (actor in the loop) + two batched critic passes + scg_gae + advantage moments',
            'rollout_policy_kernel': {'traffic': pm['traffic_bytes_per_launch'] if pm else None,
                                      'traffic_bytes_per_env_step': pm['traffic_bytes_per_launch'] / (n * T) if pm else None,
                                      'rocprof_avg_launch_us': pm.get('rocprof_avg_launch_us') if pm else None, 'traffic_source': src}}


# Evaluation protocol of the learning legs (round 3).  The deterministic policy is scored on `EVAL_ENVS` DISTINCT episodes, one per
# eval env, each from its own randomised initial state (Philox stream per env, fresh draws at every evaluation), and the
# target is the score of the reference's SHIPPED model measured under the very same protocol in the same run.
# The draws are in-bounds versions of upstream's BASE table (quadrotor.py:70-135: x +-0.5, velocities +-0.01; z 1 +- 0.5 and
# pitch +-0.1 here): upstream's own table ADDS U(0.1, 1.5) to z = 1 and U(-0.3, 0.3) to a +-0.2 rad bound, so more than half of
# its episodes start out of bounds and end at step 1 — a score over those measures the draw, not the policy.
EVAL_ENVS = 256
EVAL_INIT_RAND_Q2 = {k: {'distrib': 'uniform', 'low': lo, 'high': hi} for k, (lo, hi) in {
    'init_x': (-0.5, 0.5), 'init_x_dot': (-0.01, 0.01), 'init_z': (-0.5, 0.5), 'init_z_dot': (-0.01, 0.01),
    'init_theta': (-0.1, 0.1), 'init_theta_dot': (-0.01, 0.01)}.items()}
EVAL_INIT_RAND_Q3 = {k: {'distrib': 'uniform', 'low': lo, 'high': hi} for k, (lo, hi) in {
    'init_x': (-0.5, 0.5), 'init_x_dot': (-0.01, 0.01), 'init_y': (-0.5, 0.5), 'init_y_dot': (-0.01, 0.01), 'init_z': (-0.5, 0.5),
    'init_z_dot': (-0.01, 0.01), 'init_phi': (-0.1, 0.1), 'init_theta': (-0.1, 0.1), 'init_psi': (-0.1, 0.1),
    'init_p': (-0.01, 0.01), 'init_q': (-0.01, 0.01), 'init_r': (-0.01, 0.01)}.items()}


# BASELINE config #5 names "parameter + dynamics disturbances".  Upstream's parameter table (quadrotor.py:47-68) holds RANGES
# (M 0.022 .. 0.032 around 0.027) but benchmark_env.py:237-268 ADDS the draw to the nominal value: the mass doubles and nothing flies
# it.  Flyable statement of the same mechanism (additive draws, `respect_randomization_info`): deltas the normalised action space can
# still hover — thrust authority is +-10 % of the nominal hover thrust (norm_act_scale 0.1), so |dM| / M <= 7.4 %.
FLYABLE_PARAM_RAND_Q3 = {'M': {'distrib': 'uniform', 'low': -0.002, 'high': 0.002}, 'Ixx': {'distrib': 'uniform', 'low': -1e-6, 'high': 1e-6},
                         'Iyy': {'distrib': 'uniform', 'low': -1e-6, 'high': 1e-6}, 'Izz': {'distrib': 'uniform', 'low': -1e-6, 'high': 1e-6}}


def sac_task_config(param_rand):
    """(env_id, training config, evaluation config) of the SAC leg; param_rand: None (inertial randomisation off) or an additive table."""
    from safe_control_gym_amd.registration import load_task
    env_id, cfg = load_task('quadrotor_3D_track_disturbed')
    cfg['randomized_inertial_prop'] = False
    ev = eval_task_config(cfg, EVAL_INIT_RAND_Q3)
    if param_rand:
        from safe_control_gym_amd.env_config import QUAD_BASE_INIT_RAND     # (what the training env draws when the YAML tables are ignored, as upstream)
        cfg = dict(cfg, randomized_inertial_prop=True, respect_randomization_info=True, inertial_prop_randomization_info=param_rand,
                   init_state_randomization_info=dict(QUAD_BASE_INIT_RAND))
        ev = dict(ev, randomized_inertial_prop=True, inertial_prop_randomization_info=param_rand)
    return env_id, cfg, ev


def eval_task_config(cfg, table):
    """Task config of the evaluation env: same task, initial states drawn per episode from `table` (additive, like upstream)."""
    return dict(cfg, randomized_init=True, respect_randomization_info=True, init_state_randomization_info=table)


def shipped_ppo_score(torch, eval_env, tag='quadrotor_2D_track', hidden=128, act='tanh', evals=4):
    """Deterministic-policy score of the reference's shipped PPO model (examples/rl/models/ppo/ppo_model_<tag>.pt, weights
    committed as tests/golden/policies.npz) on `eval_env`, `evals` evaluations of EVAL_ENVS episodes each."""
    import numpy as np
    from safe_control_gym_amd.ppo import PPO, PPOConfig, _evaluate_fused_device
    pol = np.load(os.path.join(ROOT, 'tests', 'golden', 'policies.npz'))
    holder = PPO(eval_env, PPOConfig(hidden_dim=hidden, activation=act, use_gae=True, rollout_batch_size=eval_env.num_envs, rollout_steps=1,
                                     mini_batch_size=eval_env.num_envs), seed=0)
    holder.agent.ac.load_state_dict({k[len(tag) + 1:]: torch.as_tensor(pol[k]) for k in pol.files if k.startswith(tag + '/')})
    scores, lengths = [], []
    for _ in range(evals):
        res, _ = _evaluate_fused_device(eval_env, holder._policy_struct(True), 1)
        r = res.tolist()
        scores.append(r[1]); lengths.append(r[2])
    return {'mean_return': sum(scores) / len(scores), 'returns': scores, 'mean_length': sum(lengths) / len(lengths),
            'episodes': evals * eval_env.num_envs}


F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense float32 matrix peak (v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz)
LEARNER_SUMS_FILE = 'r06_learner_kernel_sums.json'


def mlp_flops(nin, h, nout, backward=False):
    """Algorithmic flops of one row through nin -> h -> h -> nout: forward 2 (nin h + h h + h nout); with backward x 3 (forward, data
    gradient, weight gradient — DESIGN.md 4.6's accounting: the first layer's unused data gradient is counted, as the 864 MFMAs per tile are)."""
    f = 2 * (nin * h + h * h + h * nout)
    return 3 * f if backward else f


def learner_kernel_sum(key):
    """Sum of kernel durations of one learner iteration (rocprofv3 --kernel-trace of tools/learner_profile.py, condensed into
    profiles/r06_learner_kernel_sums.json) — quoted only while that file names the hashes of the learner / simulator sources of this tree."""
    try:
        with open(os.path.join(ROOT, 'profiles', LEARNER_SUMS_FILE)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, f'profiles/{LEARNER_SUMS_FILE} not present'
    from safe_control_gym_amd import _learn, _lib, _sac
    want = {'env': f'0x{_lib.source_hash():016x}', 'learn': f'0x{_learn.source_hash():016x}', 'sac': f'0x{_sac.source_hash():016x}'}
    have = d.get('_meta', {}).get('source_hashes', {})
    need = ('env', 'learn') if key.startswith('ppo') else ('env', 'sac')
    if any(have.get(k) != want[k] for k in need):
        return None, f'profiles/{LEARNER_SUMS_FILE} was measured on other kernel sources: dropped'
    e = d.get(key)
    return (e, f'profiles/{LEARNER_SUMS_FILE} (rocprofv3 --kernel-trace --stats, {key})') if e else (None, f'profiles/{LEARNER_SUMS_FILE} has no entry {key}')


def ppo_leg(torch, dist, world, rank, seeds, budget_s, envs=65536, minibatch=None, lr=2e-3, target_kl=0.03, epochs=None, rollout_steps=32,
            target=None, mb_per_epoch='auto'):
    """PPO wall-clock until the deterministic-policy evaluation reaches the reference reward on BASELINE config #3's batch
    (65 536 envs per GPU): fused rollout, fused MFMA update; every iteration's weights are evaluated on a second stream
    (EVAL_ENVS distinct randomised-init episodes, fused deterministic rollout) while training goes on.  Target = the shipped
    reference model's score under the same protocol, measured here first.  Two clocks per seed: until the FIRST evaluation
    >= target is seen on the host, and until TWO CONSECUTIVE evaluations are (training continues until then or the budget).
    With several ranks: env shards + one flat gradient all-reduce per minibatch (RCCL)."""
    from safe_control_gym_amd import parallel
    from safe_control_gym_amd.ppo import PPO, AsyncEvaluator, PPOConfig
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    # re-tuned for the batch (probes: tools/sessions/s43.sh, profiles/r03_ppo_probes_65536.txt): at 65 536 envs
    # 2 epochs x 32 minibatches of 127 x 512 rows reach the target in 1.4 s median where 4 x 64 of 127 x 256 need 2.2-2.3 s
    # (16 384 envs, tools/sessions/s54.sh: 2 epochs x 32 minibatches of 127 x 128 rows 0.62 s median of 4 seeds, 4 x 16 of 127 x 256 0.75 s)
    # round 4 (tools/sessions/s83.sh, 3 seeds each, s to two consecutive evaluations >= target): what the KL-limited policy iteration
    # needs per iteration is a number of optimiser STEPS (~64), not a number of samples seen — at 65 536 envs x 32 steps = 2 M
    # samples, 2 PARTIAL epochs of 32 minibatches of 16 256 (a uniform half of the rollout; PPOConfig.extra minibatches_per_epoch)
    # reach the target in 0.71 s median (0.60-0.77) where 2 full epochs of 32 x 65 024 need 1.30 s; the full-epoch figure stays in
    # the line (`full_epochs`).  Also probed: 1 x 64 of 16 256: 0.98 s; 2 x 16 of 32 512: 0.89; 2 x 32 of 32 512: 0.97; 2 x 16 of 65 024: 0.95.
    # round 5 (tools/sessions/s118.sh, s119.sh; 8 seeds per setting, two boxes; profiles/r05_ppo_probe_*.json): the same optimiser-step budget
    # spent as 3 partial epochs x 16 minibatches (48 steps per iteration instead of 64) needs the same ~ 95 iterations and reaches the target
    # 14 % earlier on both boxes (median 0.649 vs 0.755 s and 0.790 vs 0.937 s); 4 x 12: 0.634; 2 x 24: 0.829 (slow box); 2 x 16: 0.723; 4 x 16:
    # 0.729; lr 3e-3, target_kl 0.05, 16-step rollouts, 8128-row minibatches: all slower.
    if mb_per_epoch == 'auto':
        mb_per_epoch = 16 if (envs >= 65536 and minibatch is None) else None
        if mb_per_epoch and epochs is None:
            epochs = 3
    if minibatch is None:
        minibatch = 16256
    if epochs is None:
        epochs = 2
    env_id, cfg = load_task('quadrotor_2D_track')
    pol = (128, 'tanh')
    ev_cfg = eval_task_config(cfg, EVAL_INIT_RAND_Q2)
    shipped = None
    if target is None:
        e0 = HipVecEnv(env_id, EVAL_ENVS, seed=4242, return_numpy=False, policy=pol, **ev_cfg)
        shipped = shipped_ppo_score(torch, e0)
        e0.close()
        target = shipped['mean_return']
        if world > 1:                                   # one number for every rank
            t = torch.tensor([target], device='cuda', dtype=torch.float64)
            parallel.broadcast_(t, 0)
            target = float(t.item())
    first, both, its, best_all, final = [], [], [], [], []

    def config():
        return PPOConfig(hidden_dim=128, activation='tanh', gamma=0.99, use_gae=True, gae_lambda=0.95, target_kl=target_kl,
                         entropy_coef=0.01, opt_epochs=epochs, mini_batch_size=minibatch, actor_lr=lr, critic_lr=lr,
                         rollout_batch_size=envs, rollout_steps=rollout_steps,
                         extra={'minibatches_per_epoch': mb_per_epoch} if mb_per_epoch else {})
    # One untimed iteration on a scratch instance (seed 0) before the first clock starts: the process's one-time costs — loading the
    # learner / rollout code objects, kernel attributes, first launches — were 0.45 s of the first seed's first iteration
    # (tools/ppo_iter_times.py: 455 ms, then 6.4 ms per iteration; the second seed's first iteration: 6.4 ms) and are not training.
    w_env = HipVecEnv(env_id, envs, seed=0, env_id_offset=rank * envs, return_numpy=False, policy=pol, **cfg)
    w_eval = HipVecEnv(env_id, EVAL_ENVS, seed=0, return_numpy=False, policy=pol, **ev_cfg)
    torch.cuda.synchronize()
    t_cold = time.perf_counter()
    w_ppo = PPO(w_env, config(), seed=0)
    w_aev = AsyncEvaluator(w_ppo, w_eval)
    for _ in range(3):                                  # eager, captured, replayed (PPO._run_iteration) + the evaluator's first launch
        w_ppo.train_step(lazy=(world == 1))
        w_aev.launch()
    w_aev.poll(wait=True)
    torch.cuda.synchronize()
    t_cold = time.perf_counter() - t_cold               # reported beside the clocks: what a fresh process pays once on top of them
    w_env.close(); w_eval.close()
    del w_ppo, w_aev
    RUN_AHEAD = 2                                       # iterations the host may have enqueued beyond the one the GPU is running
    wall_it, dev_it, paths = [], [], []
    for seed in range(1, seeds + 1):
        env = HipVecEnv(env_id, envs, seed=seed, env_id_offset=rank * envs, return_numpy=False, policy=pol, **cfg)
        eval_env = HipVecEnv(env_id, EVAL_ENVS, seed=seed * 111, return_numpy=False, policy=pol, **ev_cfg)
        pcfg = config()
        ppo = PPO(env, pcfg, seed=seed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_first, t_both, best, it, streak, last_ret, cap_err = None, None, -1e30, 0, 0, None, None
        max_it = int(budget_s / 0.004)                  # iteration cap (identical on every rank: no rank leaves a collective alone)
        aev = AsyncEvaluator(ppo, eval_env)

        def seen(ev, el):
            """Book one finished evaluation at host time `el`; True when the stopping rule (two consecutive >= target) holds."""
            nonlocal last_ret, best, streak, t_first
            last_ret = ev['ep_return']
            best = max(best, last_ret)
            streak = streak + 1 if last_ret >= target else 0
            if streak >= 1 and t_first is None:
                t_first = el
            return streak >= 2

        if world == 1:
            # ONE rank: nothing in this loop waits for the GPU.  Each train_step(lazy=True) is one HIP-graph replay (collection, GAE,
            # normalisation, 48 optimiser steps); the host stays at most RUN_AHEAD iterations ahead of the device (an event per
            # iteration, polled), the evaluator's result arrives through its own event, and the clock is read when a finished
            # evaluation is SEEN — the weights that earned it existed by then.  The iterations still queued when the rule holds are
            # drained outside the clock.
            pending, ev_times = [], []
            stop = False
            while it < max_it and not stop:
                res = ppo.train_step(lazy=True)
                it += 1
                ev_times.append(res['events'])
                pending.append(res['events'][2])
                aev.launch(tag=it)
                while True:
                    ev = aev.poll()
                    el = time.perf_counter() - t0
                    if ev is not None and seen(ev, el):
                        t_both, stop = el, True
                        break
                    if el > budget_s:
                        stop = True
                        break
                    while pending and pending[0].query():
                        pending.pop(0)
                    if len(pending) <= RUN_AHEAD:
                        break
            t_loop = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_drained = time.perf_counter() - t0
            # steady-state iteration: device time between the end events of consecutive replays, median (one replay per iteration)
            gaps = sorted(1e-3 * a[2].elapsed_time(b[2]) for a, b in zip(ev_times[2:-1], ev_times[3:]))
            dev_it.append(gaps[len(gaps) // 2] if gaps else None)
            wall_it.append(t_drained / max(it, 1))
            paths.append('one HIP-graph replay per iteration' if getattr(ppo, '_iter_graph', None) and ppo._iter_graph.get('graph') is not None else 'per-launch enqueue')
        else:
            while it < max_it:
                ppo.train_step()
                it += 1
                ev = aev.poll()
                torch.cuda.current_stream().synchronize()
                el = time.perf_counter() - t0
                hit = ev is not None and seen(ev, el)
                flag = torch.tensor([1.0 if (hit or streak >= 2) else 0.0, 1.0 if el > budget_s else 0.0], device=env.device)
                parallel.broadcast_(flag, 0)                # rank 0 decides for everybody
                if flag[0].item() > 0:
                    t_both = el
                    break
                if flag[1].item() > 0:
                    break
                aev.launch(tag=it)
            torch.cuda.synchronize()
            wall_it.append((time.perf_counter() - t0) / max(it, 1)); dev_it.append(None); paths.append(ppo.agent.dp_path)
            cap_err = ppo.agent.dp_capture_error
        last = aev.poll(wait=True)
        if last is not None:
            best = max(best, last['ep_return'])
            last_ret = last['ep_return']
        first.append(t_first); both.append(t_both); its.append(it); best_all.append(best); final.append(last_ret)
        env.close(); eval_env.close()
    ranks = None
    if world > 1:
        # one row per rank, so that a driver line can be read against DESIGN.md section 5's expectation table: how the data-parallel epoch
        # ran on THAT rank (captured graph / eager + why), its iteration time (synchronised loop: wall = device), and the spread
        mine = {'rank': rank, 'dp_path': paths[-1] if paths else None, 'dp_capture_error': cap_err if paths else None,
                'iteration_ms': 1e3 * statistics.median(wall_it) if wall_it else None}
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
    ok1, ok2 = [t for t in first if t is not None], [t for t in both if t is not None]
    # ---- where an iteration's time goes, and what fraction of the float32 matrix peak its algorithmic work is
    n_steps_it = epochs * min(envs * rollout_steps // minibatch, mb_per_epoch or 10 ** 9)
    flops_it = (envs * rollout_steps * mlp_flops(12, 128, 2) + envs * (rollout_steps + 1) * mlp_flops(12, 128, 1)
                + n_steps_it * minibatch * (mlp_flops(12, 128, 2, True) + mlp_flops(12, 128, 1, True)))
    wall_ms = 1e3 * statistics.median(wall_it) if wall_it else None
    dev_ok = [d for d in dev_it if d is not None]
    dev_ms = 1e3 * statistics.median(dev_ok) if dev_ok else None
    ks, ks_src = learner_kernel_sum(f'ppo/{envs}/{n_steps_it}x{minibatch}')
    ks_ms = ks['kernel_sum_ms_per_iteration'] if ks else None
    frac = lambda ms: (flops_it / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS) if ms else None        # noqa: E731
    iteration_ms = {'wall_per_iteration': wall_ms, 'device_steady_state': dev_ms, 'kernel_sum': ks_ms,
                    'gap': (wall_ms - ks_ms) if (wall_ms and ks_ms) else None, 'wall_over_kernel_sum': (wall_ms / ks_ms) if (wall_ms and ks_ms) else None,
                    'kernel_sum_source': ks_src, 'host_path': paths[0] if paths else None,
                    # the same iterations as a plain loop (no evaluation beside them, capture outside the clock): what the host path itself costs
                    'plain_loop_wall': ks.get('wall_ms_per_iteration') if ks else None,
                    'plain_loop_wall_over_kernel_sum': ks.get('wall_over_kernel_sum') if ks else None,
                    'what': 'wall = (clock start .. queue drained) / iterations, median over seeds (includes the capture iteration and the evaluator '
                            'beside it: +0.13 ms device / +0.33 ms wall per iteration, profiles/r06_eval_interference.txt); device_steady_state = median '
                            'distance between consecutive iterations\' end events; kernel_sum = rocprofv3; plain_loop_wall = tools/learner_profile.py, '
                            'the same replays without the asynchronous evaluation (same profile file)'}
    learner_roofline = {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': F32_MFMA_PEAK_TFLOPS, 'flops_per_iteration': flops_it,
                        'kernel_sum_ms': ks_ms, 'wall_ms': wall_ms, 'achieved': (flops_it / (wall_ms * 1e-3) / 1e12) if wall_ms else None,
                        'frac_of_f32_mfma_peak': frac(wall_ms), 'frac_on_kernel_time': frac(ks_ms),
                        'gap_frac': ((wall_ms - ks_ms) / wall_ms) if (wall_ms and ks_ms) else None,
                        'work': f'collector actor passes {envs} x {rollout_steps} rows + critic passes on {rollout_steps + 1} x {envs} rows + '
                                f'{n_steps_it} optimiser steps x {minibatch} rows x (forward + data gradient + weight gradient) of both networks'}
    per_rank = {}
    if ranks:
        ms = [r['iteration_ms'] for r in ranks if r and r.get('iteration_ms')]
        per_rank = {'ranks': ranks, 'iteration_ms_max_over_ranks': max(ms) if ms else None,
                    'iteration_ms_skew_max_minus_min': (max(ms) - min(ms)) if ms else None}
    return {**per_rank, 'iteration_ms': iteration_ms, 'roofline': learner_roofline,
            'target_return': target, 'target_source': 'shipped ppo_model_quadrotor_2D_track.pt under this protocol' if shipped else 'caller',
            'shipped_model_eval': shipped, 'eval_protocol': f'{EVAL_ENVS} distinct episodes per evaluation, one per eval env, initial state = '
            f'config init + U(x +-0.5, z +-0.5, pitch +-0.1, velocities +-0.01) per episode (fresh draws every evaluation), deterministic '
            f'policy, mean return', 'envs_per_gpu': envs, 'rollout_steps': rollout_steps, 'n_gpus': world, 'seeds': list(range(1, seeds + 1)),
            'wall_clock_to_first_hit_s': first, 'wall_clock_to_two_consecutive_s': both, 'iterations': its, 'best_eval_return': best_all,
            'last_eval_return': final, 'reached': len(ok1), 'reached_two_consecutive': len(ok2),
            'median_first_hit_s': statistics.median(ok1) if ok1 else None, 'median_two_consecutive_s': statistics.median(ok2) if ok2 else None,
            'median_s': statistics.median(ok2) if ok2 else None, 'budget_s_per_seed': budget_s,
            'epoch_semantics': ('PARTIAL epochs (not upstream PPO: each epoch visits %d of its %d shuffled minibatches); the upstream-semantics '
                                'figure is `full_epochs.median_s` of this object' % (mb_per_epoch, envs * rollout_steps // minibatch)) if mb_per_epoch
                               else 'full epochs (upstream semantics, ppo_utils.py:113-146)',
            'cold_start_s': t_cold, 'median_s_including_cold_start': (statistics.median(ok2) + t_cold) if ok2 else None,
            'hyper': f'MLP 12-128-128-{{2,1}} tanh, {epochs} epochs x {min(envs * rollout_steps // minibatch, mb_per_epoch or 10 ** 9)} minibatches of {minibatch}'
                     + (f' (PARTIAL epochs: {mb_per_epoch} of the {envs * rollout_steps // minibatch} minibatches of each shuffled epoch)' if mb_per_epoch else '')
                     + f', lr {lr:g}, '
                     f'target_kl {target_kl:g}, GAE 0.95, gamma 0.99, ent 0.01',
            'path': ('scg_rollout_policy + scg_ppo_step (gradient kernel, then reduction + gated Adam in one launch; exact f32 MFMA)' if world == 1 else
                     'scg_rollout_policy + scg_ppo_grad -> all-reduce -> scg_adam_gated (exact f32 MFMA)')
                    + '; every iteration\'s weights evaluated by the fused deterministic rollout on a second stream',
            'untimed_warmup': 'one train_step of a scratch instance (seed 0) before the first clock: code-object loads and first launches, '
                              'paid once per process (`cold_start_s`, measured here)'}


def multi_gpu_readiness(torch, dist, world):
    """What ONE GPU can say about the N > 1 path (DESIGN.md section 5 holds the expectation table these numbers feed): the number
    and size of the collectives per learner iteration and the fixed cost of one such all-reduce on the RCCL path (one rank: enqueue
    + kernel, no wire), eager and captured in a HIP graph — the data-parallel PPO epoch is one graph replay (ppo.py::_dp_epoch)."""
    from safe_control_gym_amd import parallel
    res = {'ppo': {'collectives_per_iteration': 48, 'bucket_bytes': 4 * 36742, 'what': 'flat fp32 gradients of actor + critic (12-128-128-{2,1}) + the approx-KL slot, '
                   'SUM all-reduce, 1 / world folded into scg_adam_gated_scaled; 3 epochs x 16 minibatches', 'host_enqueues_per_iteration_over_rccl': 3,
                   'host_path': 'one HIP-graph replay per epoch: 16 x (gradient kernel, reduction, all-reduce, gated Adam)'},
           'sac': {'collectives_per_vector_step': 32, 'bucket_bytes': 4 * 61451, 'what': 'the flat fp32 gradient vector (actor + log alpha + both critics, 24-128-128 nets) SUM-all-reduced '
                   'twice per gradient step (after the actor phase, after the critic phase), 16 gradient steps per vector step (sac.py::_fused_step_dp)',
                   'host_enqueues_per_vector_step_over_rccl': 1, 'host_path': 'one HIP-graph replay per vector step (16 gradient steps, 32 all-reduces)'},
           'env_shards': 'independent (global env ids [rank N, (rank + 1) N)): no data-path collective, weak scaling',
           'measured_with_more_than_one_gpu': False}
    # The probe runs in a CHILD process (a one-rank "nccl" group of its own): nothing it does — a failed capture, an abort inside the
    # communicator — can cost this process its JSON line.
    import subprocess
    code = ("import os, sys, json, socket, torch, torch.distributed as dist\n"
            "sys.path.insert(0, os.getcwd())\n"
            "from safe_control_gym_amd import parallel\n"
            "s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()\n"
            "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')\n"
            "torch.cuda.set_device(0)\n"
            "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
            "r = parallel.allreduce_probe((4 * 36742, 4 * 61451))\n"
            "print('PROBE_JSON ' + json.dumps(r)); sys.stdout.flush()\n"
            "dist.destroy_process_group()\n")
    try:
        if world != 1:
            raise RuntimeError('measured on the one-GPU run only')
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
            env.pop(k, None)
        p = subprocess.run([sys.executable, '-c', code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith('PROBE_JSON ')]
        if not line:
            raise RuntimeError('probe printed nothing: ' + (p.stderr or '')[-300:])
        pr = json.loads(line[-1][len('PROBE_JSON '):])
        for k, row in pr.items():                       # a one-rank in-place all-reduce is ELIDED by RCCL once captured: not a cost of anything
            if isinstance(row, dict) and 'captured_us' in row:
                row['captured_us_one_rank_elided'] = row.pop('captured_us')
        res['allreduce_us'] = pr
        res['allreduce_us']['note'] = ('ONE rank, child process: eager_us = host enqueue + the no-op kernel of the RCCL path (no wire); '
                                       'captured_us_one_rank_elided is what a graph replay of 50 one-rank in-place all-reduces costs per '
                                       'collective — RCCL elides them, so it measures NOTHING about the captured data-parallel step; the wire terms of '
                                       'DESIGN.md section 5 are estimates until a --gpus N line carries this probe on a real group')
    except Exception as exc:                                            # noqa: BLE001
        res['allreduce_us'] = {'error': repr(exc)[:300]}
    return res


def sac_leg(torch, seeds, budget_s, envs=2048, batch=4096, updates_per_step=16, lr=1e-3, warm_up_steps=65536, eval_every=50,
            buffer=4_000_000, world=1, rank=0, param_rand=None):
    """SAC wall-clock-to-reward on BASELINE config #5's env (Quadrotor3D figure-8 tracking, white-noise dynamics disturbance,
    constraint evaluation; `randomized_inertial_prop` OFF — upstream's additive draw doubles the mass and nothing can fly it,
    DESIGN §7), sac.py:162-335 semantics on the HIP engine.  Target = the score of the reference's SHIPPED SAC model
    (examples/rl/models/sac/sac_model_quadrotor_3D_track.pt, actor committed as tests/golden/sac_actor_quadrotor_3D_track.npz)
    under the same evaluation protocol as the PPO leg (EVAL_ENVS distinct randomised-init episodes per evaluation).  The
    evaluations run inside the clock on the training stream, every `eval_every` vector steps.
    With several ranks (BASELINE config #5 is SAC on 8 GPUs): `envs` envs and one replay shard per rank, the fused step's two
    gradient all-reduces per gradient step (sac.py::_fused_step_dp); every rank evaluates the same weights on the same eval
    seeds, rank 0's clock decides when the loop ends."""
    import numpy as np
    from safe_control_gym_amd import parallel
    from safe_control_gym_amd.ppo import evaluate
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.sac import SAC, MLPActorCritic, SACConfig
    from safe_control_gym_amd.vec_env import HipVecEnv

    class Det:
        def __init__(self, ac):
            self.ac = ac

        def act(self, obs):
            return self.ac.act(obs, deterministic=True)

    env_id, cfg, ev_cfg = sac_task_config(param_rand)
    eval_env = HipVecEnv(env_id, EVAL_ENVS, seed=4242, return_numpy=False, **ev_cfg)
    spec = eval_env.spec
    low = torch.as_tensor(spec.action_space.low, dtype=torch.float32, device=eval_env.device)
    high = torch.as_tensor(spec.action_space.high, dtype=torch.float32, device=eval_env.device)
    f = np.load(os.path.join(ROOT, 'tests', 'golden', 'sac_actor_quadrotor_3D_track.npz'))
    shipped = MLPActorCritic(spec.obs_dim, spec.nu, low, high, [128, 128], 'relu').to(eval_env.device)
    shipped.load_state_dict({k: torch.as_tensor(f[k]) for k in f.files if k.startswith('actor.')}, strict=False)
    det0 = Det(shipped)
    evs = [evaluate(det0, eval_env) for _ in range(4)]
    target = sum(e['ep_return'] for e in evs) / len(evs)
    first, both, best_all, steps_all, grads_all, rate, vstep_ms = [], [], [], [], [], [], []
    # untimed: a scratch instance (seed 0) takes a few vector steps, gradient steps and one evaluation-sized policy call, so that the
    # process's one-time costs (code-object loads, kernel attributes, first launches) are not inside the first seed's clock
    w_env = HipVecEnv(env_id, envs, seed=0, env_id_offset=rank * envs, return_numpy=False, **cfg)
    w = SAC(w_env, SACConfig(hidden_dim=128, activation='relu', train_batch_size=batch, actor_lr=lr, critic_lr=lr, warm_up_steps=envs * world,
                             train_interval=envs * world, max_buffer_size=8 * envs, extra={'updates_per_step': updates_per_step}), seed=0)
    for _ in range(6):
        w.train_step()
    w.agent.deterministic_policy().act(w.obs)
    torch.cuda.synchronize()
    w_env.close()
    del w
    for seed in range(1, seeds + 1):
        env = HipVecEnv(env_id, envs, seed=seed, env_id_offset=rank * envs, return_numpy=False, **cfg)
        scfg = SACConfig(hidden_dim=128, activation='relu', train_batch_size=batch, actor_lr=lr, critic_lr=lr, warm_up_steps=warm_up_steps * world,
                         train_interval=envs * world, max_buffer_size=buffer, extra={'updates_per_step': updates_per_step})
        sac = SAC(env, scfg, seed=seed)
        det = sac.agent.deterministic_policy()          # (fused agents: the library's batched actor, one launch per evaluation step)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_first, t_both, best, it, streak, n_grad = None, None, -1e30, 0, 0, 0
        finished = False
        pending, t_eval, t_learn0, it_learn0 = [], 0.0, None, 0
        while True:
            if world > 1:                               # rank 0's clock: every rank leaves on the same iteration
                go = torch.tensor([1.0 if (time.perf_counter() - t0 < budget_s and not finished) else 0.0], device=env.device)
                parallel.broadcast_(go, 0)
                if go.item() == 0:
                    break
            elif finished or time.perf_counter() - t0 >= budget_s:
                break
            # ONE rank: a vector step is two graph replays (collector; 16 gradient steps) and no read-back (lazy): the host only waits
            # when it is more than 4 vector steps ahead of the device, and at the evaluations (every `eval_every` steps, inside the clock)
            res = sac.train_step(lazy=(world == 1))
            if n_grad == 0 and res.get('updates'):      # first learning step: the clock of the steady-state vector step starts here
                torch.cuda.synchronize()
                t_learn0, it_learn0 = time.perf_counter(), it
            n_grad += int(res.get('updates', 0))
            it += 1
            if world == 1:
                e = torch.cuda.Event()
                e.record()
                pending.append(e)
                if len(pending) > 4:
                    pending.pop(0).synchronize()
            if it % eval_every == 0:
                torch.cuda.synchronize()
                te = time.perf_counter()
                e = evaluate(det, eval_env)
                torch.cuda.synchronize()
                pending.clear()
                t_eval += time.perf_counter() - te
                el = time.perf_counter() - t0
                best = max(best, e['ep_return'])
                streak = streak + 1 if e['ep_return'] >= target else 0
                if streak >= 1 and t_first is None:
                    t_first = el
                if streak >= 2:
                    t_both = el
                    finished = True                     # (leaves at the top of the next iteration, together with the other ranks)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if t_learn0 is not None and it > it_learn0:     # learning vector steps only, evaluations taken out
            evals_learning = it // eval_every - it_learn0 // eval_every
            vstep_ms.append(1e3 * (time.perf_counter() - t_learn0 - t_eval * evals_learning / max(1, it // eval_every)) / (it - it_learn0))
        first.append(t_first); both.append(t_both); best_all.append(best); steps_all.append(sac.total_steps); grads_all.append(n_grad)
        rate.append(sac.total_steps / wall)
        fused = bool(getattr(sac.agent, 'use_fused', False))
        dp_path, dp_err = getattr(sac.agent, 'dp_path', None), getattr(sac.agent, 'dp_capture_error', None)
        env.close()
    eval_env.close()
    ok1, ok2 = [t for t in first if t is not None], [t for t in both if t is not None]
    # ---- the gradient step against the float32 matrix peak: algorithmic flops per minibatch row of one SACAgent.update
    # (sac_utils.py:143-170): actor forward on obs and on next_obs, both Q forward + d q / d a at (obs, a), the actor's backward
    # (data + weight gradients; its forward pass is the first launch's, kept as activation tiles — counted once), both target Q
    # forward, both Q forward + data + weight gradients
    nobs, nu, hd = spec.obs_dim, spec.nu, 128
    a_f, q_f = mlp_flops(nobs, hd, 2 * nu), mlp_flops(nobs + nu, hd, 1)
    flops_step = batch * (2 * a_f + 2 * 2 * q_f + 2 * a_f + 2 * q_f + 2 * 3 * q_f)
    ks, ks_src = learner_kernel_sum(f'sac/{batch}/{updates_per_step}')
    step_us = ks['gradient_step_us'] if ks else None
    vs_ms = statistics.median(vstep_ms) if vstep_ms else None
    sac_roofline = {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': F32_MFMA_PEAK_TFLOPS, 'flops_per_gradient_step': flops_step,
                    'gradient_steps_per_vector_step': updates_per_step, 'gradient_step_us_rocprof': step_us,
                    'kernel_sum_ms_per_vector_step': ks['kernel_sum_ms_per_vector_step'] if ks else None, 'wall_ms_per_vector_step': vs_ms,
                    'frac_of_f32_mfma_peak': (updates_per_step * flops_step / (vs_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS) if vs_ms else None,
                    'frac_on_kernel_time': (flops_step / (step_us * 1e-6) / 1e12 / F32_MFMA_PEAK_TFLOPS) if step_us else None,
                    'gap_frac': ((vs_ms - ks['kernel_sum_ms_per_vector_step']) / vs_ms) if (vs_ms and ks) else None, 'kernel_sum_source': ks_src,
                    'what': 'wall = learning vector steps only (after warm-up), evaluations taken out, median over seeds'}
    per_rank = {}
    if world > 1:
        import torch.distributed as dist
        mine = {'rank': rank, 'dp_path': dp_path, 'dp_capture_error': dp_err, 'vector_step_ms': vs_ms}
        rows = [None] * world
        dist.all_gather_object(rows, mine)
        ms = [r['vector_step_ms'] for r in rows if r and r.get('vector_step_ms')]
        per_rank = {'ranks': rows, 'vector_step_ms_max_over_ranks': max(ms) if ms else None,
                    'vector_step_ms_skew_max_minus_min': (max(ms) - min(ms)) if ms else None}
    return {**per_rank, 'roofline': sac_roofline, 'task': 'quadrotor_3D_track_disturbed ' + ('(randomized_inertial_prop ON: additive draws per episode, training AND evaluation envs, '
                    f'{ {k: (v["low"], v["high"]) for k, v in param_rand.items()} })' if param_rand else '(randomized_inertial_prop off)'),
            'target_return': target,
            'target_source': 'shipped sac_model_quadrotor_3D_track.pt under this protocol',
            'shipped_model_eval': {'returns': [e['ep_return'] for e in evs], 'mean_length': sum(e['ep_length'] for e in evs) / len(evs)},
            'eval_protocol': f'{EVAL_ENVS} distinct randomised-init episodes per evaluation (x, y, z +-0.5, angles +-0.1, rates +-0.01), '
                             f'deterministic policy, every {eval_every} vector steps inside the clock',
            'envs_per_gpu': envs, 'n_gpus': world, 'seeds': list(range(1, seeds + 1)), 'wall_clock_to_first_hit_s': first, 'wall_clock_to_two_consecutive_s': both,
            'best_eval_return': best_all, 'env_steps': steps_all, 'gradient_steps': grads_all, 'env_steps_per_s_incl_learning': rate,
            'reached': len(ok1), 'reached_two_consecutive': len(ok2), 'median_first_hit_s': statistics.median(ok1) if ok1 else None,
            'median_two_consecutive_s': statistics.median(ok2) if ok2 else None, 'median_s': statistics.median(ok2) if ok2 else None,
            'budget_s_per_seed': budget_s, 'fused_update': fused,
            'scope': ("BASELINE config #5's ENV and algorithm (Quadrotor3D lemniscate tracking, dynamics disturbance, constraint evaluation, SAC) on "
                      f'{world} GPU(s) at {envs} envs per GPU; config #5 itself names 8 x MI355X, which has not been run (DESIGN.md section 5)'),
            'hyper': f'MLP 24-128-128 relu (actor + twin Q), batch {batch}, {updates_per_step} gradient steps per vector step of {envs} envs, '
                     f'lr {lr:g}, warm-up {warm_up_steps} env steps, tau 0.005, alpha 0.2'}


def self_spawn(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: become the launcher.  Re-executes this file under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` (one rank per
    GPU, exactly the driver's multi-GPU command), passes the ranks' stdout / stderr through, returns their exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f'[bench] --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks: {" ".join(cmd[1:8])} ...', file=sys.stderr)
    return subprocess.run(cmd).returncode


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_spawn(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # SCG_BENCH_BACKEND=gloo lets the N>1 control flow be exercised on a box with fewer GPUs than ranks
    # (ranks then share devices round-robin); the driver's runs use nccl (= RCCL), one rank per GPU.
    backend = os.environ.get('SCG_BENCH_BACKEND', 'nccl')
    if world != args.gpus:
        # the line's n_gpus must be the number of ranks that really ran: a mismatch is a launch error, not a warning
        print(f'[bench] error: --gpus {args.gpus} but WORLD_SIZE={world} (launch with python -m torch.distributed.run '
              f'--nproc-per-node {args.gpus}, or run plain `python bench.py --gpus {args.gpus}` and let bench.py spawn the ranks)',
              file=sys.stderr)
        sys.exit(2)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            if torch.cuda.device_count() < world:
                print(f'[bench] error: {world} ranks over RCCL need {world} GPUs, this node shows {torch.cuda.device_count()} '
                      f'(SCG_BENCH_BACKEND=gloo shares devices for a control-flow test)', file=sys.stderr)
                sys.exit(2)
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend)
        if dist.get_world_size() != args.gpus:
            print(f'[bench] error: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}', file=sys.stderr)
            sys.exit(2)
    else:
        torch.cuda.set_device(0)
    rccl_ranks = dist.get_world_size() if (world > 1 and backend == 'nccl') else (1 if world == 1 else 0)
    dev = torch.device('cuda', torch.cuda.current_device())
    dtype = torch.float32 if args.dtype == 'f32' else torch.float64
    N = args.envs
    G = max(1, min(args.graph_len, args.steps))
    hb = StepBench(torch, args.task, N, dtype, rank=rank, generic=args.generic, graph_len=G, use_graph=not args.no_graph)

    def device_sync():
        # (measured, tools/sessions/s70.sh: polling an event behind the work before this call is SLOWER — 127.5 vs 125.0 us per
        #  20-step region — hipDeviceSynchronize already spins)
        torch.cuda.synchronize()

    def barrier():
        device_sync()
        if world > 1:
            dist.barrier()
            device_sync()

    def timed_region():
        # barrier + synchronize, clock, K steps, synchronize, clock, barrier.  A rank's clock stops when ITS K steps are done; the
        # closing barrier (N > 1: an RCCL all-reduce of tens of microseconds, comparable to a 20-step region) keeps the ranks
        # together for the next region but is not part of the K steps — the job's time is the MAX over ranks taken below.
        barrier()
        t0 = time.perf_counter()
        n = hb.do(args.steps)
        device_sync()
        el_s = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
        return el_s, n

    hb.do(args.warmup)
    # timed region: EXACTLY K steps between barrier + synchronize.  When K is small (one graph replay), the same K-step
    # region is repeated and the median repeat is reported (every repeat is bracketed the same way).
    repeats = 1 if args.steps >= 5000 else 31
    samples, done_steps = [], args.steps
    for _ in range(repeats):
        el_s, done_steps = timed_region()
        samples.append(el_s)
    elapsed = statistics.median(samples)
    el = torch.tensor([elapsed], device=dev if backend == 'nccl' else 'cpu', dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    period_us = hb.kernel_period_us(max(args.steps, 4000))      # launch period inside long back-to-back replays
    ok = hb.sane()
    value = world * N * done_steps / elapsed
    out = None
    if rank == 0:
        out = {
            'metric': 'env-steps/sec (whole node), Quadrotor2D-track', 'value': value, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': done_steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / done_steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic',
            'config': {'workload': f'{args.task}: {N} envs/GPU x {world} GPU, 1 launch of the fused step kernel per '
                                   f'control step (20 engine substeps, obs/reward/done/info/constraints, auto-reset), '
                                   f'synthetic U(-1,1) actions resident in HBM, '
                                   f'{"HIP graph of %d steps" % G if hb.graph is not None else "per-step Python launches"}',
                       'envs_per_gpu': N, 'task_yaml': f'safe_control_gym_amd/configs/{args.task}.yaml',
                       'parallelism': f'env-shard x{world}', 'rccl_ranks': rccl_ranks, 'collective_backend': backend if world > 1 else None,
                       'finite_outputs': ok,
                       'kernel_build': 'config-specialised' if hb.env.specialized else 'generic',
                       'timing': (f'median of {repeats} timed repeats of the {done_steps}-step region' if repeats > 1 else 'one timed region')
                                 + ('; per rank: barrier + synchronize, clock, K steps, synchronize, clock, barrier; MAX over ranks' if world > 1 else ''),
                       'timed_region_samples_ms': [round(1e3 * s, 4) for s in (min(samples), elapsed, max(samples))]},
            'roofline': roofline_of(args.task, args.dtype, N, period_us, driver_step_us=1e6 * elapsed / done_steps,
                                    geometry=launch_geometry(N, bool(hb.env.specialized), args.task)),
        }
    full = not args.no_secondary and args.task == 'quadrotor_2D_track' and args.dtype == 'f32'
    # The learning legs run collectives (N > 1: RCCL).  A rank that dies or hangs inside one must not cost the run its line:
    # after a generous limit every rank gives up, rank 0 prints what it has (the headline is complete at this point).
    import threading
    limit = 120.0 + 3.0 * (max(args.ppo_seeds, 0) * args.ppo_seconds * 2 + max(args.sac_seeds, 0) * args.sac_seconds)

    def bail():
        if rank == 0 and out is not None:
            out.setdefault('ppo', {'error': 'watchdog: the learning legs did not finish'})
            out['watchdog'] = f'learning legs exceeded {limit:.0f} s: line printed without them'
            print(json.dumps(out), flush=True)
        os._exit(0 if rank == 0 else 3)
    watchdog = threading.Timer(limit, bail)
    watchdog.daemon = True
    if world > 1:
        watchdog.start()
    if rank == 0 and world == 1 and full:
        try:
            fb = StepBench(torch, args.task, N, torch.float64, graph_len=500)
            fb.do(500)
            us = fb.kernel_period_us(3000)
            out['f64'] = {'avg_launch_us': us, 'ms_per_step': us * 1e-3, 'env_steps_per_s': N / (us * 1e-6),
                          'note': 'same workload on the float64 kernels (the reference computes in float64); bytes per env-step double, '
                                  'algorithmic 2 x 187 - 10 = 364 B', 'frac': (364 * N / (us * 1e-6)) / 1e9 / HBM_PEAK_GBS,
                          'traffic': traffic_of(args.task, 'f64', N)[0], 'traffic_source': traffic_of(args.task, 'f64', N)[1],
                          'finite_outputs': fb.sane()}
            fb.env.close()
        except Exception as exc:                                    # noqa: BLE001
            out['f64'] = {'error': repr(exc)[:200]}
        out['secondary'] = secondary_env_kernels(torch, N)
        try:
            out['gae'] = gae_leg(torch)
        except Exception as exc:                                    # noqa: BLE001
            out['gae'] = {'error': repr(exc)[:200]}
        try:
            out['sequence'] = sequence_leg(torch, N)
        except Exception as exc:                                    # noqa: BLE001
            out['sequence'] = {'error': repr(exc)[:200]}
        try:
            out['fused_rollout'] = fused_rollout_leg(torch, N)
        except Exception as exc:                                    # noqa: BLE001
            out['fused_rollout'] = {'error': repr(exc)[:200]}
    hb.env.close()
    if full and args.ppo_seeds > 0 and (world == 1 or backend == 'nccl' or os.environ.get('SCG_BENCH_PPO_GLOO')):
        try:
            res = ppo_leg(torch, dist, world, rank, args.ppo_seeds, args.ppo_seconds, envs=args.ppo_envs)
            if world == 1 and args.ppo_envs != 16384:           # the small-batch point (round 2's leg) under the same protocol and target
                res['envs_16384'] = ppo_leg(torch, dist, world, rank, args.ppo_seeds, args.ppo_seconds, envs=16384, target=res['target_return'])
            if world == 1 and args.ppo_envs >= 65536:           # the same batch with FULL epochs (round 3's configuration), same protocol and target
                res['full_epochs'] = ppo_leg(torch, dist, world, rank, args.ppo_seeds, args.ppo_seconds, envs=args.ppo_envs, minibatch=65024,
                                             mb_per_epoch=None, target=res['target_return'])
                # `median_s` is the UPSTREAM-semantics figure (full shuffled epochs, ppo_utils.py:113-146) so that rounds stay comparable;
                # the partial-epoch configuration's clock has its own name
                res['median_s_partial_epochs'] = res.pop('median_s')
                res['median_s'] = res['full_epochs']['median_s']
                res['median_s_semantics'] = ('median_s = full_epochs.median_s (upstream PPO epochs); median_s_partial_epochs = this object\'s own '
                                             'configuration (3 partial epochs x 16 minibatches per iteration)')
        except Exception as exc:                                    # noqa: BLE001
            import traceback
            res = {'error': repr(exc)[:300], 'trace': traceback.format_exc()[-600:]}
        if rank == 0:
            out['ppo'] = res
    if full and args.sac_seeds > 0 and (world == 1 or backend == 'nccl' or os.environ.get('SCG_BENCH_SAC_GLOO')):
        try:
            res = sac_leg(torch, args.sac_seeds, args.sac_seconds, world=world, rank=rank)
            if world == 1:          # config #5 WITH parameter disturbances (flyable additive deltas), target re-measured under them
                res['param_randomised'] = sac_leg(torch, args.sac_seeds, args.sac_seconds, param_rand=FLYABLE_PARAM_RAND_Q3)
        except Exception as exc:                                    # noqa: BLE001
            import traceback
            res = {'error': repr(exc)[:300], 'trace': traceback.format_exc()[-600:]}
        if rank == 0:
            out['sac'] = res
    probe = None
    if full and world > 1 and backend == 'nccl':            # every rank: the all-reduce of the learners' buckets on the REAL group
        try:
            from safe_control_gym_amd import parallel
            probe = parallel.allreduce_probe((4 * 36742, 4 * 61451))
        except Exception as exc:                                    # noqa: BLE001
            probe = {'error': repr(exc)[:300]}
    watchdog.cancel()
    if rank == 0 and full:
        out['multi_gpu'] = multi_gpu_readiness(torch, dist, world)
        if probe is not None:
            probe['note'] = f'{world} ranks over RCCL, in-place fp32 SUM, microseconds per collective (eager; 50 per HIP-graph replay)'
            out['multi_gpu']['allreduce_us'] = probe
            out['multi_gpu']['measured_with_more_than_one_gpu'] = True
        # the learning legs' rooflines, flattened into the object the driver's parser keeps (depth 2)
        rf = out['roofline']
        pr = (out.get('ppo') or {}).get('roofline') or {}
        pi = (out.get('ppo') or {}).get('iteration_ms') or {}
        sr = (out.get('sac') or {}).get('roofline') or {}
        rf['learners'] = {'peak_TFLOPs_f32_mfma': F32_MFMA_PEAK_TFLOPS,
                          'ppo_flops_per_iteration': pr.get('flops_per_iteration'), 'ppo_wall_ms_per_iteration': pi.get('wall_per_iteration'),
                          'ppo_device_ms_per_iteration': pi.get('device_steady_state'), 'ppo_kernel_sum_ms_per_iteration': pi.get('kernel_sum'),
                          'ppo_wall_over_kernel_sum': pi.get('wall_over_kernel_sum'), 'ppo_plain_loop_wall_ms_per_iteration': pi.get('plain_loop_wall'),
                          'ppo_plain_loop_wall_over_kernel_sum': pi.get('plain_loop_wall_over_kernel_sum'),
                          'ppo_frac_of_peak_on_wall': pr.get('frac_of_f32_mfma_peak'),
                          'ppo_frac_of_peak_on_kernel_time': pr.get('frac_on_kernel_time'),
                          'ppo_median_s_partial_epochs': (out.get('ppo') or {}).get('median_s_partial_epochs'),
                          'ppo_median_s_full_epochs': (out.get('ppo') or {}).get('median_s'),
                          'sac_flops_per_gradient_step': sr.get('flops_per_gradient_step'), 'sac_gradient_step_us': sr.get('gradient_step_us_rocprof'),
                          'sac_wall_ms_per_vector_step': sr.get('wall_ms_per_vector_step'), 'sac_frac_of_peak_on_wall': sr.get('frac_of_f32_mfma_peak'),
                          'sac_frac_of_peak_on_kernel_time': sr.get('frac_on_kernel_time'), 'sac_median_s': (out.get('sac') or {}).get('median_s')}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(args.task, hb.cfg, hb.env_id, args.cpu_seconds, N)
            out['cpu_baseline']['gpu_over_cpu'] = value / out['cpu_baseline']['value']
        print(json.dumps(out, default=str))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
