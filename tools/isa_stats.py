#!/usr/bin/env python3
"""Per-kernel resource summary of a gfx950 assembly dump (hipcc -S --cuda-device-only).

usage: isa_stats.py file.s [name-substring ...]
       isa_stats.py --build <task> [--dtype f32|f64] [name-substring ...]     (specialised build of a shipped task YAML)
Prints for every kernel (or the ones whose demangled name contains a substring): instruction count, VGPR / AGPR /
SGPR, private segment (scratch) bytes, LDS bytes, and the counts of scratch_load/store, s_waitcnt, branches, packed-fp32
and buffer/global memory instructions.  The numbers that the round reports quote come from here.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'] + names, capture_output=True, text=True).stdout.split('\n')
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def parse(path):
    txt = open(path).read()
    # code bodies:  <name>:  ... s_endpgm ... .Lfunc_end
    bodies = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', txt, re.S | re.M):
        bodies[m.group(1)] = m.group(2)
    meta = {}
    for m in re.finditer(r'^\s*-\s+\.agpr_count:.*?\n(?=\s*-\s+\.agpr_count:|amdhsa\.target|\.\.\.)', txt, re.S | re.M):
        blk = m.group(0)
        name = re.search(r'\.name:\s+(\S+)', blk)
        if not name:
            continue
        g = lambda k: int(re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)) if re.search(r'\.%s:\s+(\d+)' % k, blk) else 0   # noqa: E731
        meta[name.group(1)] = dict(vgpr=g('vgpr_count'), agpr=g('agpr_count'), sgpr=g('sgpr_count'),
                                   scratch=g('private_segment_fixed_size'), lds=g('group_segment_fixed_size'),
                                   vgpr_spill=g('vgpr_spill_count'), sgpr_spill=g('sgpr_spill_count'))
    rows = []
    for name, body in bodies.items():
        ins = [ln.split(';')[0].strip() for ln in body.split('\n')]
        ins = [ln for ln in ins if ln and not ln.endswith(':') and not ln.startswith('.') and not ln.startswith(';')]
        cnt = lambda pat: sum(1 for i in ins if re.match(pat, i))      # noqa: E731
        r = dict(name=name, n=len(ins), scratch_ld=cnt(r'scratch_load'), scratch_st=cnt(r'scratch_store'),
                 waitcnt=cnt(r's_waitcnt'), branch=cnt(r's_c?branch'), pk=cnt(r'v_pk_\w+_f32'),
                 valu=cnt(r'v_'), salu=cnt(r's_(?!waitcnt|c?branch|nop|endpgm|load)'), sload=cnt(r's_load'),
                 vmem_ld=cnt(r'(buffer|global|flat)_load'), vmem_st=cnt(r'(buffer|global|flat)_store'),
                 lds_op=cnt(r'ds_'), trans=cnt(r'v_(rcp|rsq|sqrt|exp|log|sin|cos)_'), mul32=cnt(r'v_mul_(hi|lo)_u32|v_mad_u64_u32'))
        r.update(meta.get(name, {}))
        rows.append(r)
    return rows


def build_asm(task, dtype):
    sys.path.insert(0, ROOT)
    from safe_control_gym_amd import _lib
    from safe_control_gym_amd.env_config import EnvSpec
    from safe_control_gym_amd.registration import load_task
    env_id, cfg = load_task(task)
    c, _ = EnvSpec(env_id, cfg).to_c_config(1, _lib.F64 if dtype == 'f64' else _lib.F32, 0)
    src, h = _lib.spec_source(c)
    hdr, _ = _lib.spec_paths(h)
    os.makedirs(os.path.dirname(hdr), exist_ok=True)
    with open(hdr, 'w') as f:
        f.write(src)
    out = f'/tmp/isa_{task}_{dtype}.s'
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-ffp-contract=on', '-std=c++17', '-DSCG_SPEC', '-include', hdr, '-S',
           '--cuda-device-only', '-o', out, os.path.join(_lib.CSRC_DIR, 'scg_kernels.hip')] + os.environ.get('SCG_EXTRA_FLAGS', '').split()
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode:
        sys.exit(res.stderr)
    return out


def main():
    args = sys.argv[1:]
    if args and args[0] == '--build':
        task = args[1]
        args = args[2:]
        dtype = 'f32'
        if args and args[0] == '--dtype':
            dtype = args[1]
            args = args[2:]
        path = build_asm(task, dtype)
    else:
        path, args = args[0], args[1:]
    rows = parse(path)
    dm = demangle([r['name'] for r in rows])
    print(f'# {path}')
    for r in rows:
        nm = dm.get(r['name'], r['name'])
        nm = re.sub(r'^void scg::', '', nm).split('(')[0]
        if args and not any(a in nm for a in args):
            continue
        print(f"{nm}\n    instr {r['n']:5d}  valu {r['valu']:5d} (pk_f32 {r['pk']}, trans {r['trans']}, mul32 {r['mul32']})  salu {r['salu']}"
              f"  s_load {r['sload']}  vmem ld/st {r['vmem_ld']}/{r['vmem_st']}  ds {r['lds_op']}  waitcnt {r['waitcnt']}  branch {r['branch']}\n"
              f"    vgpr {r.get('vgpr', '?')}  agpr {r.get('agpr', '?')}  sgpr {r.get('sgpr', '?')}  scratch {r.get('scratch', '?')} B"
              f" (ld/st {r['scratch_ld']}/{r['scratch_st']}, vgpr spills {r.get('vgpr_spill', '?')})  lds {r.get('lds', '?')} B")


if __name__ == '__main__':
    main()
