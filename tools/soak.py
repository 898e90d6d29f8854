#!/usr/bin/env python3
"""Long-run determinism / sanity soak of the step kernels (write-through stores, recurrence integrator, speculative reset draws):
two env batches with the same seed and the same action ring are stepped `--steps` control steps each (HIP graphs of 1000 launches);
after every replay a checksum of every bound output is folded into device accumulators.  At the end: raw simulator state, counters
and all checksums of the two runs must be BITWISE equal (a torn store, a hazard or a race shows up as a difference), everything
finite, finished-episode lengths within [1, CTRL_STEPS].

    python tools/soak.py [--envs 65536] [--steps 100000] [--tasks t1,t2] [--generic]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from safe_control_gym_amd.registration import load_task  # noqa: E402
from safe_control_gym_amd.vec_env import HipVecEnv  # noqa: E402


def run(task, n, steps, generic, seed=11):
    env_id, cfg = load_task(task)
    env = HipVecEnv(env_id, n, seed=seed, return_numpy=False, specialize=False if generic else 'auto', **cfg)
    g = torch.Generator(device='cuda').manual_seed(99)
    acts = torch.rand(64, n, env.spec.nu, device='cuda', generator=g) * 2 - 1
    env.reset_tensors()
    out = env.out
    G = 1000
    acc = torch.zeros(8, dtype=torch.float64, device='cuda')
    lens = torch.zeros(2, device='cuda')                                  # min / max finished-episode length seen
    lens[0] = 1e9

    def body():
        for t in range(G):
            o = env.step_tensors(acts[t % 64])
            d = o.done.to(torch.float64)
            acc[0] += o.obs.double().sum(); acc[1] += o.reward.double().sum(); acc[2] += d.sum()
            acc[3] += o.flags.double().sum(); acc[4] += o.mse.double().sum(); acc[5] += o.c_values.double().sum()
            acc[6] += (o.fin_stats.double().sum(1) * d).sum(); acc[7] += (o.terminal_obs.double().sum(1) * d).sum()
            fl = torch.where(o.done.bool(), o.fin_length.float(), torch.full_like(o.fin_length.float(), float('nan')))
            lens[0] = torch.minimum(lens[0], torch.nan_to_num(fl, nan=1e9).min())
            lens[1] = torch.maximum(lens[1], torch.nan_to_num(fl, nan=0.0).max())
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        body()
    for _ in range(max(1, steps // G) - 1):
        graph.replay()
    torch.cuda.synchronize()
    state = torch.as_tensor(env.get_raw_state())
    step, ep = env.get_counters()
    res = {'acc': acc.cpu(), 'state': state, 'step': torch.as_tensor(step), 'episode': torch.as_tensor(ep.astype('int64')), 'lens': lens.cpu(),
           'ctrl_steps': env.spec.max_episode_steps, 'specialised': bool(env.specialized)}
    env.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--steps', type=int, default=100000)
    ap.add_argument('--tasks', default='quadrotor_2D_track,cartpole_stab,quadrotor_3D_track,quadrotor_3D_track_disturbed')
    ap.add_argument('--generic', action='store_true')
    a = ap.parse_args()
    torch.cuda.set_device(0)
    ok = True
    for task in a.tasks.split(','):
        r1, r2 = run(task, a.envs, a.steps, a.generic), run(task, a.envs, a.steps, a.generic)
        same = all(torch.equal(r1[k], r2[k]) for k in ('acc', 'state', 'step', 'episode', 'lens'))
        finite = bool(torch.isfinite(r1['acc']).all() and torch.isfinite(r1['state']).all())
        lo, hi = r1['lens'].tolist()
        len_ok = 1 <= lo and hi <= r1['ctrl_steps']
        ok &= same and finite and len_ok
        print(json.dumps({'task': task, 'envs': a.envs, 'control_steps': (a.steps // 1000) * 1000, 'specialised': r1['specialised'],
                          'bitwise_repeatable': same, 'finite': finite, 'episodes_finished': float(r1['acc'][2]),
                          'episode_length_min_max': [lo, hi], 'ctrl_steps': r1['ctrl_steps']}))
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
