#!/usr/bin/env python3
"""Condense gpurun_out/prof (tools/profile_round4.sh) into the committed profiles/<tag>_* artefacts.

    python tools/profile_post.py r04

* <tag>_kernel_stats_<workload>_<N>.csv   rocprofv3 --kernel-trace --stats summary rows of the workload's kernels
* <tag>_hbm_traffic.json                  per workload: FETCH_SIZE x 2 + WRITE_SIZE per launch (separate --pmc passes, gfx950 x2 fetch
                                          correction of MI355X_MICROARCH.md), executed VALU instructions per wave (SQ_INSTS_VALU / SQ_WAVES),
                                          the rocprof average launch duration; `_meta.source_hash` = the kernel sources it was measured on
                                          (bench.py drops the numbers when the tree's hash differs)
Directory names written by the round-4 script: kt_<workload>_<dtype>_<N>, pmc_<COUNTERS>_<workload>_<dtype>_<N> (counters joined by '+');
round-2/3 layouts (kt_<task>_<N>) are still understood."""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, 'gpurun_out', os.environ.get('SCG_PROF_DIR', 'prof'))
DST = os.path.join(ROOT, 'profiles')
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r04'
# workload -> substring of the kernel it is about
KERNEL_OF = {'sequence_all': 'step_sequence_kernel', 'sequence_collector': 'step_sequence_kernel', 'rollout_policy': 'rollout_policy_kernel'}


def kernel_of(workload):
    return KERNEL_OF.get(workload, 'step_kernel')


def is_kernel(kname, name):
    """`kname` in a rocprof kernel name; 'step_kernel' also names the other launch geometries of the same step (step_wsback_kernel,
    step_wide_kernel; round 5's step_split_kernel)."""
    return kname in name or (kname == 'step_kernel' and any(k in name for k in ('step_split_kernel', 'step_wsback_kernel', 'step_wide_kernel')))


def second_half_mean(rows, counter, kname):
    sel = [r for r in rows if is_kernel(kname, r['Kernel_Name']) and r['Counter_Name'] == counter]
    if sel and 'Dispatches_Averaged' in sel[0]:             # condensed on the box by tools/profile_round5.sh: already the second-half mean
        r = max(sel, key=lambda q: int(q['Dispatches_Total']))
        return float(r['Counter_Value']), int(r['Dispatches_Averaged'])
    v = [float(r['Counter_Value']) for r in sel]
    v = v[len(v) // 2:]
    return (sum(v) / len(v), len(v)) if v else None


traffic = {}
for extra in ('ppo_iteration', 'ppo_iteration_65536', 'sac_iteration'):
    stats = glob.glob(f'{SRC}/{extra}/**/*kernel_stats.csv', recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        keep = [r for r in rows if float(r['Percentage']) > 0.2]
        with open(f'{DST}/{TAG}_kernel_stats_{extra}.csv', 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
        print(extra, [(r['Name'][:40], r['Calls'], r['AverageNs']) for r in keep[:4]])
for d in sorted(glob.glob(f'{SRC}/kt_*')):
    if not os.path.isdir(d):
        continue
    m = re.match(r'kt_(.+?)_(f32|f64)_(\d+)$', os.path.basename(d)) or re.match(r'kt_(.+)()_(\d+)$', os.path.basename(d))
    work, dt, N = m.group(1), m.group(2) or 'f32', int(m.group(3))
    kname = kernel_of(work)
    entry = {'envs': N, 'kernel': kname}
    stats = glob.glob(f'{d}/**/*kernel_stats.csv', recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        keep = [r for r in rows if is_kernel(kname, r['Name']) or float(r['Percentage']) > 0.5]
        suffix = '' if dt == 'f32' else f'_{dt}'
        with open(f'{DST}/{TAG}_kernel_stats_{work}{suffix}_{N}.csv', 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
        for r in keep:
            if is_kernel(kname, r['Name']):
                entry['kernel_launched'] = next((k for k in ('step_split_kernel', 'step_wsback_kernel', 'step_wide_kernel') if k in r['Name']), kname)
                entry['rocprof_avg_launch_us'] = float(r['AverageNs']) * 1e-3
                entry['rocprof_calls'] = int(r['Calls'])
                print(work, dt, N, kname, 'calls', r['Calls'], 'avg ns', r['AverageNs'], 'pct', r['Percentage'])
                break
    vals = {}
    for f in glob.glob(f'{SRC}/pmc_*_{work}_{dt}_{N}/**/*counter_collection.csv', recursive=True) + \
            (glob.glob(f'{SRC}/pmc_*_{work}_{N}/**/*counter_collection.csv', recursive=True) if dt == 'f32' else []):
        rows = list(csv.DictReader(open(f)))
        for C in {r['Counter_Name'] for r in rows}:
            got = second_half_mean(rows, C, kname)
            if got:
                vals[C] = got
    if 'FETCH_SIZE' in vals and 'WRITE_SIZE' in vals:
        fetch = vals['FETCH_SIZE'][0] * 1024 * 2          # KB -> B, gfx950 x2 correction (MI355X_MICROARCH.md)
        write = vals['WRITE_SIZE'][0] * 1024
        entry.update({'FETCH_SIZE_KB_per_launch': vals['FETCH_SIZE'][0], 'WRITE_SIZE_KB_per_launch': vals['WRITE_SIZE'][0],
                      'dispatches_averaged': [vals['FETCH_SIZE'][1], vals['WRITE_SIZE'][1]],
                      'fetch_bytes_corrected_x2': fetch, 'write_bytes': write, 'traffic_bytes_per_launch': fetch + write,
                      'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (TCC slot limit); FETCH_SIZE doubled per '
                              'MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)'})
        if kname == 'step_kernel':
            entry['traffic_bytes_per_env_step'] = (fetch + write) / N
        print(work, dt, N, 'traffic B/launch', fetch + write, 'per env', (fetch + write) / N)
    for C in ('SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'TCC_HIT_sum', 'TCC_MISS_sum', 'TCC_EA0_WRREQ_sum',
              'TCC_EA0_WRREQ_64B_sum', 'TCC_EA0_WRREQ_STALL_sum', 'TCC_EA0_RDREQ_sum', 'TCC_EA0_RDREQ_32B_sum', 'GRBM_GUI_ACTIVE', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_VMEM_RD',
              'TCP_PENDING_STALL_CYCLES_sum', 'TCP_TCC_READ_REQ_LATENCY_sum', 'TCP_TCC_READ_REQ_sum', 'TA_BUSY_avr', 'SQ_INST_CYCLES_VMEM'):
        if C in vals:
            entry.setdefault('counters', {})[C] = vals[C][0]
    if 'SQ_INSTS_VALU' in vals and 'SQ_WAVES' in vals and vals['SQ_WAVES'][0] > 0:
        entry['valu_instructions_per_wave'] = vals['SQ_INSTS_VALU'][0] / vals['SQ_WAVES'][0]
        entry['waves_per_launch'] = vals['SQ_WAVES'][0]
        print(work, dt, N, 'VALU / wave', entry['valu_instructions_per_wave'])
    if len(entry) > 2:
        traffic[f'{work}/{dt}/{N}'] = entry
if traffic:
    from safe_control_gym_amd import _lib
    traffic['_meta'] = {'source_hash': f'0x{_lib.source_hash():016x}', 'tag': TAG,
                        'how': 'tools/profile_round5.sh (round 4: profile_round4.sh) on one MI355X (gpurun), condensed by tools/profile_post.py; the hash is of the kernel '
                               'sources in the tree when this file was written — run the two back to back'}
    json.dump(traffic, open(f'{DST}/{TAG}_hbm_traffic.json', 'w'), indent=1)
