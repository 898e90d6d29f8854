"""Ablation of the step kernel cost (dev tool)."""
import json, os, subprocess, sys
variants = {
    'base': {},
    'substeps1': {'pyb_freq': 50},
    'no_constraints': {'constraints': None},
    'no_goal_horizon': {'obs_goal_horizon': 0},
    'substeps1_nocon': {'pyb_freq': 50, 'constraints': None},
    'no_random_init': {'randomized_init': False},
    'long_episode_norand': {'randomized_init': False, 'done_on_out_of_bound': False},
}
for name, ov in variants.items():
    env = dict(os.environ, SCG_BENCH_OVERRIDE=json.dumps(ov))
    out = subprocess.run([sys.executable, 'bench.py', '--steps', '5000', '--warmup', '500', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=200, env=env)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print(f"{name:24s} us/launch {d['roofline']['avg_launch_us']:.2f}")
    except Exception:
        print(name, 'FAILED', out.stderr[-400:])
