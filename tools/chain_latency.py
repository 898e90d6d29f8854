#!/usr/bin/env python3
"""Serial-chain estimate of the engine-substep loop of a shipped task's specialised step kernel (one wave per SIMD regime).

    python tools/chain_latency.py [task ...]        -> profiles/r06_chain_latency.json

Compiles the float32 specialised build to assembly, takes the step kernel's longest backward-branch loop (the unrolled substep
loop; kernels whose loop is fully unrolled — Quadrotor2D — have none and are reported from the whole integrator block between the
state load and the first output store), and runs tools/isa_sim.py's in-order model on it with the issue / dependent-issue intervals
MEASURED on MI355X (tools/issue_rate.hip -> profiles/r05_issue_rate.txt): per loop pass the number of vector instructions, the cycles
the pass takes with dependencies, and the cycles it would take at the wave's issue limit alone (4.8 clocks per instruction).  The
difference is dependent-instruction latency that only another wave on the SIMD — or a shorter chain — can hide."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import isa_sim  # noqa: E402

CLOCK_GHZ = 2.396          # profiles/r05_issue_rate.txt


def kernel_body(asm, pattern):
    m = re.search(r'^(_ZN3scg11step_kernel\w*%s\w*):[^\n]*\n(.*?)^\.Lfunc_end\d+:' % pattern, asm, re.S | re.M)
    return m.group(2).split('\n')


def loops(lines):
    """(first, last) line index of every backward-branch loop."""
    labels = {ln.split(':')[0].strip(): k for k, ln in enumerate(lines) if re.match(r'^\.LBB\w+:', ln)}
    out = []
    for k, ln in enumerate(lines):
        m = re.match(r'\s*s_cbranch_\w+\s+(\.LBB\w+)', ln)
        if m and labels.get(m.group(1), 10 ** 9) < k:
            out.append((labels[m.group(1)], k))
    return out


def main():
    from safe_control_gym_amd import _lib
    tasks = sys.argv[1:] or ['cartpole_stab', 'quadrotor_3D_track']
    res = {}
    for task in tasks:
        from safe_control_gym_amd.env_config import EnvSpec
        from safe_control_gym_amd.registration import load_task
        env_id, cfg = load_task(task)
        c, _ = EnvSpec(env_id, cfg).to_c_config(1, _lib.F32, 0)
        env = dict(os.environ, SCG_EXTRA_FLAGS=' '.join(_lib.sched_flags(c)[0]))
        subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_stats.py'), '--build', task, 'nothing'], env=env, check=True, stdout=subprocess.DEVNULL)
        asm = open(f'/tmp/isa_{task}_f32.s').read()
        body = kernel_body(asm, 'Lb1E')                    # the one-window variant (what HipVecEnv launches)
        lp = loops(body)
        if not lp:
            res[task] = {'error': 'no loop (fully unrolled integrator)'}
            continue
        a, b = max(lp, key=lambda ab: ab[1] - ab[0])
        seg = body[a:b + 1]
        if sum(1 for ln in seg if re.match(r'\s*v_mad_u64_u32', ln)) > 24:      # a region around the Philox blocks, not a substep loop:
            res[task] = {'error': 'the substep loop of this build is fully unrolled (straight-line integrator, no loop to model); '
                                  'see bench.py wave_issue for its issue-limit time'}
            print(task, res[task]['error'])
            continue
        n_rcp = sum(1 for ln in seg if re.match(r'\s*v_rcp_f32', ln))
        substeps, cfg_sub = int(c.substeps), int(c.substeps)
        t1, n, stall1 = isa_sim.simulate(seg, 1)
        t4, n4, _ = isa_sim.simulate(seg, 4)
        per_pass = (t4 - t1) / 3.0                           # steady-state pass (registers carried from the previous pass)
        n_rsq = sum(1 for ln in seg if re.match(r'\s*v_rsq_f32', ln))      # Quadrotor3D: one quaternion normalisation per substep
        unroll = max(1, n_rcp) if task.startswith('cartpole') else (max(1, n_rsq) if n_rsq else None)
        passes = (substeps / unroll) if unroll else None
        e = {'loop_lines': [a, b], 'valu_per_pass': n, 'cycles_per_pass': per_pass, 'issue_limit_cycles_per_pass': n * 4.8,
             'substeps_per_pass': unroll, 'passes_per_control_step': passes}
        if passes:
            e['chain_us_per_control_step'] = per_pass * passes / (CLOCK_GHZ * 1e3)
            e['issue_limit_us_per_control_step'] = n * 4.8 * passes / (CLOCK_GHZ * 1e3)
            e['dependent_instructions_per_substep'] = (per_pass / unroll) / 8.4
        res[task] = e
        print(task, json.dumps(e))
    res['_meta'] = {'source_hash': f'0x{_lib.source_hash():016x}', 'model': 'tools/isa_sim.py with profiles/r05_issue_rate.txt: one wave issues a VALU '
                    'instruction per 4.8 clocks, a dependent one per 8.4 (v_rcp 12.3, v_mad_u64_u32 8.8)', 'clock_GHz': CLOCK_GHZ}
    json.dump(res, open(os.path.join(ROOT, 'profiles', 'r06_chain_latency.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
