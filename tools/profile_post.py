#!/usr/bin/env python3
"""Condense gpurun_out/prof (tools/profile_round.sh) into the committed profiles/<tag>_* artefacts."""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'prof')
DST = os.path.join(ROOT, 'profiles')
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r02'
traffic = {}
for extra in ('ppo_iteration', 'ppo_iteration_65536', 'sac_iteration', 'sequence'):
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
    name = os.path.basename(d)[3:]
    task, N = name.rsplit('_', 1)
    N = int(N)
    stats = glob.glob(f'{d}/**/*kernel_stats.csv', recursive=True)
    if stats:
        rows = list(csv.DictReader(open(stats[0])))
        keep = [r for r in rows if 'step_kernel' in r['Name'] or float(r['Percentage']) > 0.5]
        with open(f'{DST}/{TAG}_kernel_stats_{task}_{N}.csv', 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
        for r in keep:
            if 'step_kernel' in r['Name']:
                print(task, N, 'step_kernel: calls', r['Calls'], 'avg ns', r['AverageNs'], 'pct', r['Percentage'])
    vals = {}
    for C in ('FETCH_SIZE', 'WRITE_SIZE'):
        files = glob.glob(f'{SRC}/pmc_{C}_{task}_{N}/**/*counter_collection.csv', recursive=True)
        if not files: continue
        v = [float(r['Counter_Value']) for r in csv.DictReader(open(files[0])) if 'step_kernel' in r['Kernel_Name'] and r['Counter_Name'] == C]
        v = v[len(v) // 2:]
        if v:
            vals[C] = (sum(v) / len(v), len(v))
    if len(vals) == 2:
        fetch = vals['FETCH_SIZE'][0] * 1024 * 2          # KB -> B, gfx950 x2 correction (MI355X_MICROARCH.md)
        write = vals['WRITE_SIZE'][0] * 1024
        traffic[f'{task}/f32/{N}'] = {
            'envs': N, 'FETCH_SIZE_KB_per_launch': vals['FETCH_SIZE'][0], 'WRITE_SIZE_KB_per_launch': vals['WRITE_SIZE'][0],
            'dispatches_averaged': [vals['FETCH_SIZE'][1], vals['WRITE_SIZE'][1]],
            'fetch_bytes_corrected_x2': fetch, 'write_bytes': write, 'traffic_bytes_per_launch': fetch + write,
            'traffic_bytes_per_env_step': (fetch + write) / N,
            'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (TCC slot limit); FETCH_SIZE doubled per '
                    'MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)'}
        print(task, N, 'traffic B/env-step', (fetch + write) / N)
if traffic:
    json.dump(traffic, open(f'{DST}/{TAG}_hbm_traffic.json', 'w'), indent=1)
# diagnostic counter passes (4 M envs)
diag = {}
for d in sorted(glob.glob(f'{SRC}/diag*')):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(f'{d}/**/*counter_collection.csv', recursive=True):
        acc = {}
        for r in csv.DictReader(open(f)):
            if 'step_kernel' in r['Kernel_Name']:
                acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        for k, v in acc.items():
            v = v[len(v) // 2:]
            diag[k] = sum(v) / len(v)
if diag:
    json.dump(diag, open(f'{DST}/{TAG}_pmc_4m_envs.json', 'w'), indent=1)
    print(json.dumps(diag, indent=1))
