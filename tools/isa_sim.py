#!/usr/bin/env python3
"""Static in-order issue estimate for a straight-line range of gfx950 ISA (one wave per SIMD regime).

Model (measured with tools/issue_rate.hip on MI355X, profiles/r05_issue_rate.txt): ONE wave issues a VALU instruction every 4.8
clocks at best (v_mad_u64_u32: 8.1, transcendentals: 8.5) and its result can feed a dependent VALU instruction 8.4 clocks after
issue (v_mad_u64_u32: 8.8, v_rcp and friends: 12.3); SALU 1/2 cycles.  (The SIMD itself issues every 2.4 clocks when it has two
waves to choose from — 4.3 for packed-fp32 and integer multiplies — which is why a second wave per SIMD is nearly free for scalar
code and not for packed code.)
usage: isa_sim.py file.s first_line last_line [loop_iterations]
"""
import re, sys

def regs(tok):
    out = []
    for m in re.finditer(r'\b([vs])\[(\d+):(\d+)\]|\b([vs])(\d+)\b', tok):
        if m.group(1):
            out += [f'{m.group(1)}{k}' for k in range(int(m.group(2)), int(m.group(3)) + 1)]
        else:
            out.append(f'{m.group(4)}{m.group(5)}')
    if 'vcc' in tok: out.append('vcc')
    return out

def simulate(lines, iters=1):
    ready, t, n_valu, stall = {}, 0, 0, 0
    for _ in range(iters):
        for ln in lines:
            ln = ln.split(';')[0].strip()
            if not ln or ln.endswith(':') or ln.startswith('.'): continue
            op, _, rest = ln.partition(' ')
            ops = [o.strip() for o in rest.split(',')] if rest else []
            if op.startswith('v_'):
                ndst = 2 if op.startswith(('v_div_scale', 'v_mad_u64', 'v_mad_i64', 'v_add_co', 'v_addc_co', 'v_sub_co', 'v_subb_co')) else 1
                dst = [r for o in ops[:ndst] for r in regs(o)]
                src = [r for o in ops[ndst:] for r in regs(o)]
                if op.startswith(('v_fmac', 'v_mac', 'v_pk_fmac')): src += dst
                if op.startswith('v_cmp') and not op.endswith('_e64'): dst, src = ['vcc'], [r for o in ops for r in regs(o)]
                if op.startswith(('v_cndmask_b32_e32', 'v_div_fmas', 'v_addc', 'v_subb')): src.append('vcc')
                trans = bool(re.match(r'v_(rcp|rsq|sqrt|exp|log|sin|cos)_', op))
                wide = op.startswith(('v_mad_u64', 'v_mad_i64'))
                lat = 12.3 if trans else 8.8 if wide else 8.4
                start = max([t] + [ready.get(r, 0) for r in src])
                stall += start - t
                for r in dst: ready[r] = start + lat
                t = start + (8.5 if trans else 8.1 if wide else 4.8)
                n_valu += 1
            elif op.startswith('s_') and not op.startswith(('s_waitcnt', 's_nop', 's_load', 's_cbranch', 's_branch', 's_endpgm')):
                dst = regs(ops[0]) if ops else []
                src = [r for o in ops[1:] for r in regs(o)]
                start = max([t] + [ready.get(r, 0) for r in src])
                stall += start - t
                for r in dst: ready[r] = start + 2
                t = start + 1
            else:
                t += 1
    return t, n_valu, stall

if __name__ == '__main__':
    f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    lines = open(f).read().split('\n')[a - 1:b]
    t, n, stall = simulate(lines, iters)
    print(f'lines {a}-{b} x{iters}: {n} VALU, {t:.0f} cycles ({t / 2.4e3:.2f} us @2.4GHz), dependency stalls {stall:.0f} cycles, {t / max(n, 1):.1f} cycles/VALU')
