"""Every built gfx950 library is free of the wide-store data hazard (tools/hazard_lint.py): a > 64-bit VMEM store whose data
VGPRs are overwritten by a VALU instruction within the next two wait states.  LLVM leaves it unpadded when soffset is an
SGPR; MI355X does tear the stored value then (found by tests/test_gpu_parity_scale.py at 65 536 envs)."""
import glob
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('hazard_lint', os.path.join(ROOT, 'tools', 'hazard_lint.py'))
hazard_lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(hazard_lint)


def test_lint_recognises_the_pattern():
    bad = '''0000000000001000 <k>:
	buffer_store_dwordx4 v[42:45], v27, s[40:43], s4 offen     // 0
	v_mov_b32_e32 v42, s6                                      // 8
	s_endpgm
'''
    hz, n = hazard_lint.lint(bad)
    assert n == 1 and len(hz) == 1 and hz[0][0] == 'k'
    padded = bad.replace('\tv_mov_b32_e32 v42', '\ts_nop 1\n\tv_mov_b32_e32 v42')
    assert hazard_lint.lint(padded) == ([], 1)
    one_state = bad.replace('\tv_mov_b32_e32 v42', '\ts_mov_b32 s4, 0x70\n\tv_mov_b32_e32 v42')
    assert len(hazard_lint.lint(one_state)[0]) == 1                     # one wait state is not enough
    other_reg = bad.replace('v_mov_b32_e32 v42', 'v_mov_b32_e32 v46')
    assert hazard_lint.lint(other_reg) == ([], 1)
    pair = bad.replace('v_mov_b32_e32 v42, s6', 'v_fma_f64 v[44:45], v[0:1], v[2:3], v[4:5]')
    assert len(hazard_lint.lint(pair)[0]) == 1
    glob_st = '''0000000000001000 <k>:
	global_store_dwordx4 v[2:3], v[10:13], off offset:48
	v_lshl_add_u64 v[2:3], v[2:3], 0, 64
	global_store_dwordx4 v[2:3], v[10:13], off
	v_mov_b32_e32 v13, 0
'''
    hz, n = hazard_lint.lint(glob_st)                                   # the address pair is not store data
    assert n == 2 and len(hz) == 1 and 'v13' in hz[0][2]


def _libs():
    pkg = os.path.join(ROOT, 'safe_control_gym_amd')
    return sorted(glob.glob(os.path.join(pkg, '*.so')) + glob.glob(os.path.join(pkg, 'spec', '*.so')))


def test_built_libraries_are_hazard_free():
    libs = _libs()
    if not libs:
        pytest.skip('no built libraries in-tree')
    if not os.path.exists(os.path.join(hazard_lint.LLVM, 'llvm-objdump')):
        pytest.skip('llvm-objdump not available')
    bad = {}
    for so in libs:
        hz, _ = hazard_lint.lint(hazard_lint.disassemble(so))
        if hz:
            bad[os.path.basename(so)] = hz[:3]
    assert not bad, bad
