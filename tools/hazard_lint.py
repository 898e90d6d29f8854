#!/usr/bin/env python3
"""Lint gfx950 code objects for the wide-store data hazard LLVM (ROCm 7.2) does not pad:

    buffer_store_dwordx4 v[42:45], v27, s[40:43], s4 offen
    v_mov_b32_e32 v42, s6                    <- overwrites the store's data before the TA has read all of it

A VMEM store of more than 64 bits reads its data VGPRs over several cycles; a VALU write to one of them in the next
1-2 issue slots can win the race (MI355X: seen on the 4th lane quad of each 16-lane row, dword 0, on a cold first launch:
tests/test_gpu_parity_scale.py caught 16-64 torn doubles out of 1.5 M).  GCNHazardRecognizer inserts the wait states only
when soffset is NOT an SGPR ("this hazard only exists if the instruction is not using a register in the soffset field"),
which does not hold on this part.  buf_st128 (csrc/scg_env_core.h) therefore keeps the uniform
array offset out of soffset, so that LLVM sees (and pads) the hazard; this lint proves no wide store of a built library
is followed by such a write.

usage: hazard_lint.py lib.so [...]      exit status 1 if any hazard is found
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
# data operand: first for buffer_ / scratch_ stores, second (after the address) for global_ / flat_ stores
STORE = re.compile(r'^\s*(?:(?:buffer|scratch)_store_dwordx[34]\s+|(?:global|flat)_store_dwordx[34]\s+v(?:\[\d+:\d+\]|\d+),\s*)v\[(\d+):(\d+)\]')
DST = re.compile(r'^\s*(v_\w+|ds_read\w*|ds_load\w*|buffer_load\w*|global_load\w*|flat_load\w*|scratch_load\w*|v_accvgpr_read\w*)\s+(v\[(\d+):(\d+)\]|v(\d+))')
WAIT_STATES = 2


def disassemble(so):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'dev.co')
        subprocess.run([f'{LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={fat}', so, os.devnull], check=True,
                       capture_output=True)
        subprocess.run([f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--input={fat}',
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', f'--output={co}'], check=True, capture_output=True)
        return subprocess.run([f'{LLVM}/llvm-objdump', '-d', '--mcpu=gfx950', co], check=True, capture_output=True,
                              text=True).stdout


def cost(ins):
    """Wait states an instruction provides before the next one issues."""
    m = re.match(r'^\s*s_nop\s+(\d+)', ins)
    return 1 + int(m.group(1)) if m else 1


def lint(text):
    hazards, func, n_wide = [], '?', 0
    lines = text.split('\n')
    for i, ln in enumerate(lines):
        f = re.match(r'^[0-9a-f]+ <(\w+)>:', ln)
        if f:
            func = f.group(1)
            continue
        m = STORE.match(ln)
        if not m:
            continue
        n_wide += 1
        lo, hi = int(m.group(1)), int(m.group(2))
        waited, j = 0, i + 1
        while waited < WAIT_STATES and j < len(lines):
            nxt = lines[j]
            j += 1
            if not nxt.strip() or re.match(r'^[0-9a-f]+ <', nxt):
                break
            d = DST.match(nxt)
            if d and d.group(1).startswith('v_'):           # (memory loads land later than any wait-state window)
                a, b = (int(d.group(3)), int(d.group(4))) if d.group(3) else (int(d.group(5)),) * 2
                if a <= hi and b >= lo:
                    hazards.append((func, ln.split('//')[0].strip(), nxt.split('//')[0].strip()))
                    break
            if re.match(r'^\s*s_(c?branch|endpgm|setpc)', nxt):
                break
            waited += cost(nxt)
    return hazards, n_wide


def main():
    bad = 0
    for so in sys.argv[1:]:
        hz, n = lint(disassemble(so))
        print(f'{so}: {n} wide stores, {len(hz)} unpadded VALU overwrite(s) of store data')
        for func, st, wr in hz[:20]:
            print(f'    {func[:80]}\n        {st}\n        {wr}')
        bad += len(hz)
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
