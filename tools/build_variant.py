#!/usr/bin/env python3
"""Tagged variant of a learner / SAC shape library from ANOTHER copy of the sources (e.g. the previous commit's, for same-box A/B runs),
stamped with the CURRENT source hash so that the loaders take it as it is:
    python tools/build_variant.py learn 12 128 2 tanh base /tmp/ab          (SCG_LEARN_TAG=base selects it)
    python tools/build_variant.py sac 24 128 4 relu base /tmp/ab            (tools/sac_step_ab.py base)
<srcroot> mirrors the repo layout (safe_control_gym_amd/csrc, include); extra hipcc flags after it."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    kind, nin, hid, nu, act, tag, root = sys.argv[1:8]
    extra = sys.argv[8:]
    from safe_control_gym_amd import _lib as L
    from safe_control_gym_amd import _learn, _sac
    mod, pre, src = (_learn, 'SCG_L_', 'scg_learn.hip') if kind == 'learn' else (_sac, 'SCG_S_', 'scg_sac.hip')
    names = ('NIN', 'H', 'NU', 'ACT') if kind == 'learn' else ('NOBS', 'H', 'NU', 'ACT')
    vals = (int(nin), int(hid), int(nu), _learn.ACTS[act])
    so = os.path.join(L.SPEC_DIR, f'libscg_{kind}_{nin}_{hid}_{nu}_{act}_{tag}.so')
    cmd = [L._hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared'] + [f'-D{pre}{n}={v}' for n, v in zip(names, vals)] \
        + [f'-DSCG_SRC_HASH=0x{mod.source_hash():016x}ULL', '-o', so, os.path.join(root, 'safe_control_gym_amd', 'csrc', src)] + extra
    subprocess.run(cmd, check=True)
    print(so)


if __name__ == '__main__':
    main()
