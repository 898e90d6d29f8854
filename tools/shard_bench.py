#!/usr/bin/env python3
"""Sub-shard launches: one control step of N envs as S launches of N/S envs on S streams (scg_step_range).

At 65 536 envs a launch is a latency chain (dispatch floor, loads, one wave's instruction stream, store drain);
disjoint env ranges are independent, so S chains can overlap.  Two ways to drive them:
  branches : ONE HIP graph, S parallel branches of G steps each (fork / join events during capture)
  streams  : S graphs of G steps, each replayed on its own stream
usage: shard_bench.py [--task T] [--envs N] [--steps K] [--shards 1,2,4,8] [--modes branches,streams]
Prints one JSON line per (S, mode): us per control step of all N envs.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--task', default='quadrotor_2D_track')
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--steps', type=int, default=4000)
    ap.add_argument('--graph-len', type=int, default=500)
    ap.add_argument('--shards', default='1,2,4,8')
    ap.add_argument('--modes', default='branches,streams')
    ap.add_argument('--dtype', default='f32')
    args = ap.parse_args()
    import torch
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.vec_env import HipVecEnv
    dev = torch.device('cuda', 0)
    dtype = torch.float32 if args.dtype == 'f32' else torch.float64
    env_id, cfg = load_task(args.task)
    N = args.envs
    env = HipVecEnv(env_id, N, seed=1337, dtype=dtype, return_numpy=False, **cfg)
    ring = 64
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    actions = torch.rand(ring, N, env.spec.nu, device=dev, dtype=dtype, generator=gen) * 2 - 1
    env.reset_tensors()
    lean_out, lean_c = env.bind_outputs(state=None, noisy_action=None)
    G = args.graph_len
    for S in [int(x) for x in args.shards.split(',')]:
        per = (N // S + 63) // 64 * 64
        ranges = [(k * per, min(per, N - k * per)) for k in range(S) if k * per < N]
        streams = [torch.cuda.Stream() for _ in ranges]

        def chain(k, g):
            first, cnt = ranges[k]
            for t in range(g):
                env.step_range_tensors(first, cnt, actions[t % ring], out=lean_out, c_out=lean_c)

        for mode in args.modes.split(','):
            if S == 1 and mode == 'streams':
                continue
            torch.cuda.synchronize()
            # warm-up outside capture
            for k in range(len(ranges)):
                chain(k, 2)
            torch.cuda.synchronize()
            if mode == 'branches':
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    cur = torch.cuda.current_stream()
                    fork = torch.cuda.Event()
                    fork.record(cur)
                    joins = []
                    for k, s in enumerate(streams):
                        s.wait_event(fork)
                        with torch.cuda.stream(s):
                            chain(k, G)
                            e = torch.cuda.Event()
                            e.record(s)
                            joins.append(e)
                    for e in joins:
                        cur.wait_event(e)

                def run(reps):
                    for _ in range(reps):
                        graph.replay()
            else:
                graphs = []
                for k, s in enumerate(streams):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.stream(s):
                        with torch.cuda.graph(g, stream=s):
                            chain(k, G)
                    graphs.append(g)

                def run(reps):
                    for _ in range(reps):
                        for g, s in zip(graphs, streams):
                            with torch.cuda.stream(s):
                                g.replay()
            reps = max(1, args.steps // G)
            run(2)
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            run(reps)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            us = el / (reps * G) * 1e6
            ok = bool(torch.isfinite(lean_out.reward).all().item())
            print(json.dumps({'task': args.task, 'envs': N, 'shards': len(ranges), 'mode': mode, 'us_per_step': round(us, 3),
                              'env_steps_per_s': N * reps * G / el, 'finite': ok}), flush=True)
    env.close()


if __name__ == '__main__':
    main()
