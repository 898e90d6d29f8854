#!/usr/bin/env python3
"""Known answers of the reference's observation / reward normalisers, produced by running its own classes
(math_and_models/normalization.py: RunningMeanStd, MeanStdNormalizer, RewardStdNormalizer incl. the
`self.ret[dones.astype(np.long)] = 0` index-array reset, read-only mode, state_dict round trip) on fixed random streams.

    python tests/golden/make_normalizers.py        (build container only: needs /root/reference) -> normalizers.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.golden import ref_stubs  # noqa: E402

ref_stubs.install()

from safe_control_gym.math_and_models import normalization as ref  # noqa: E402

T, N, D = 40, 12, 5


def main():
    rng = np.random.default_rng(17)
    out = {'obs': rng.normal(2.0, 3.0, (T, N, D)), 'rew': rng.normal(0.5, 1.5, (T, N))}
    done = rng.random((T, N)) < 0.15
    done[::4] = False                                   # steps on which no env finishes (row 1 of the index reset stays)
    done[7] = True                                      # a step on which every env finishes (row 0 stays)
    out['done'] = done
    o = ref.MeanStdNormalizer(shape=(D,), clip=2.5)
    r = ref.RewardStdNormalizer(gamma=0.97, clip=3.0)
    on, rn, ret, om, ov, rv, cnt = [], [], [], [], [], [], []
    for t in range(T):
        if t == 30:                                     # frozen statistics from here on (evaluation mode)
            o.set_read_only(); r.set_read_only()
        on.append(o(out['obs'][t]).copy())
        rn.append(r(out['rew'][t], done[t]).copy())
        ret.append(r.ret.copy()); om.append(o.rms.mean.copy()); ov.append(o.rms.var.copy()); rv.append(np.array(r.rms.var)); cnt.append(o.rms.count)
    out.update(obs_norm=np.array(on), rew_norm=np.array(rn), ret=np.array(ret), obs_mean=np.array(om), obs_var=np.array(ov),
               rew_var=np.array(rv), obs_count=np.array(cnt))
    sd = o.state_dict()
    out['sd_mean'], out['sd_var'] = np.asarray(sd['mean']), np.asarray(sd['var'])
    np.savez_compressed(os.path.join(HERE, 'normalizers.npz'), **out)
    print('normalizers.npz written; final obs count', cnt[-1], 'reward var', float(rv[-1]))


if __name__ == '__main__':
    main()
