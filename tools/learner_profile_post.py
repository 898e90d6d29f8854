#!/usr/bin/env python3
"""Condense tools/profile_round6.sh's learner passes: <dir>/kt_{ppo,sac}/**/kernel_stats.csv (rocprofv3 --kernel-trace --stats) and the
LEARNER_PROFILE lines of the traced / untraced runs -> <dir>/r06_learner_kernel_sums.json (+ r06_kernel_stats_{ppo,sac}_iteration.csv).

    ppo/<envs>/<steps>x<minibatch>: kernel_sum_ms_per_iteration = sum of ALL kernel durations of the trace / iterations executed;
                                    the top kernels with their average durations; wall / device ms per iteration of the untraced run
    sac/<batch>/<updates>:          gradient_step_us = sum of the fused step's kernels (scg_sac.hip) / gradient steps executed;
                                    kernel_sum_ms_per_vector_step = updates x that + the collector's kernels per vector step
`_meta.source_hashes` = the kernel sources it was measured on (bench.py drops the entries when the tree's hashes differ)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
D = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'prof6')
SAC_KERNELS = ('wide::', 'reduce_kernel<', 'finish_kernel', 'adam_kernel', 'sac_step')       # (the fused gradient step's launches, scg_sac.hip)


def line_of(path):
    try:
        for ln in open(path):
            if ln.startswith('LEARNER_PROFILE '):
                return json.loads(ln[len('LEARNER_PROFILE '):])
    except OSError:
        pass
    return None


def stats_of(mode):
    f = glob.glob(os.path.join(D, f'kt_{mode}', '**', '*kernel_stats.csv'), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else None


from safe_control_gym_amd import _learn, _lib, _sac       # noqa: E402
out = {'_meta': {'source_hashes': {'env': f'0x{_lib.source_hash():016x}', 'learn': f'0x{_learn.source_hash():016x}', 'sac': f'0x{_sac.source_hash():016x}'},
                 'how': 'tools/profile_round6.sh (rocprofv3 --kernel-trace --stats of tools/learner_profile.py; wall clocks from the same command untraced)'}}
for mode in ('ppo', 'sac'):
    rows, traced, plain = stats_of(mode), line_of(os.path.join(D, f'kt_{mode}.log')), line_of(os.path.join(D, f'plain_{mode}.log'))
    if not rows or not traced:
        print(mode, 'missing', bool(rows), bool(traced))
        continue
    with open(os.path.join(D, f'r06_kernel_stats_{mode}_iteration.csv'), 'w', newline='') as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows([r for r in rows if float(r['Percentage']) > 0.1])
    total_ns = sum(float(r['TotalDurationNs']) for r in rows)
    top = [{'kernel': r['Name'][:90], 'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) * 1e-3, 'pct': float(r['Percentage'])} for r in rows[:8]]
    if mode == 'ppo':
        n = traced['iterations_executed']
        e = {'kernel_sum_ms_per_iteration': total_ns * 1e-6 / n, 'iterations_in_trace': n, 'top_kernels': top,
             'flops_per_iteration': traced['flops_per_iteration'], 'optimiser_steps_per_iteration': traced['optimiser_steps_per_iteration'],
             'traced_wall_ms_per_iteration': traced['wall_ms_per_iteration']}
        if plain:
            e.update(wall_ms_per_iteration=plain['wall_ms_per_iteration'], device_ms_per_iteration=plain['device_ms_per_iteration_median'],
                     wall_over_kernel_sum=plain['wall_ms_per_iteration'] / e['kernel_sum_ms_per_iteration'], host_path=plain['host_path'])
        ng = line_of(os.path.join(D, 'plain_ppo_nograph.log'))
        if ng:
            e['per_launch_enqueue'] = {'wall_ms_per_iteration': ng['wall_ms_per_iteration'], 'device_ms_per_iteration': ng['device_ms_per_iteration_median']}
        step = [r for r in rows if 'ppo_grad_kernel' in r['Name'] or 'ppo_reduce_adam_kernel' in r['Name']]
        e['optimiser_step_us'] = sum(float(r['AverageNs']) for r in step) * 1e-3
        e['frac_of_f32_mfma_peak_on_kernel_time'] = traced['flops_per_iteration'] / (e['kernel_sum_ms_per_iteration'] * 1e-3) / 157.3e12
    else:
        sac_rows = [r for r in rows if any(k in r['Name'] for k in SAC_KERNELS) and 'at::native' not in r['Name']]
        g_ns = sum(float(r['TotalDurationNs']) for r in sac_rows)
        n_g, n_v = traced['gradient_steps_executed'], traced['vector_steps_executed']
        step_us = g_ns * 1e-3 / n_g
        coll_ms = (total_ns - g_ns) * 1e-6 / n_v
        ups = n_g // traced['learning_vector_steps_executed']
        e = {'gradient_step_us': step_us, 'collector_ms_per_vector_step': coll_ms, 'kernel_sum_ms_per_vector_step': ups * step_us * 1e-3 + coll_ms,
             'gradient_steps_in_trace': n_g, 'flops_per_gradient_step': traced['flops_per_gradient_step'],
             'frac_of_f32_mfma_peak_on_kernel_time': traced['flops_per_gradient_step'] / (step_us * 1e-6) / 157.3e12,
             'step_kernels': [{'kernel': r['Name'][:90], 'calls': int(r['Calls']), 'avg_us': float(r['AverageNs']) * 1e-3} for r in sac_rows]}
        if plain:
            e['wall_ms_per_vector_step'] = plain['wall_ms_per_vector_step']
    out[traced['key']] = e
    print(mode, json.dumps({k: v for k, v in e.items() if not isinstance(v, list)}))
with open(os.path.join(D, 'r06_learner_kernel_sums.json'), 'w') as f:
    json.dump(out, f, indent=1)
