#!/usr/bin/env python3
"""Actor of the reference's shipped SAC model for Quadrotor3D tracking (examples/rl/models/sac/sac_model_quadrotor_3D_track.pt,
trained with examples/rl/config_overrides/quadrotor_3D/sac_quadrotor_3D.yaml) as a small fixture: its deterministic
evaluation score on this repo's env is the target of the SAC wall-clock-to-reward measurement (tools/sac_time_to_reward.py).

    python tests/golden/make_sac_actor.py           (build container only: needs /root/reference)
Critics and optimiser state are left out (1 MB -> 90 KB)."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
d = torch.load('/root/reference/examples/rl/models/sac/sac_model_quadrotor_3D_track.pt', weights_only=False, map_location='cpu')
out = {k: v.numpy() for k, v in d['agent']['ac'].items() if k.startswith('actor.')}
out['log_alpha'] = np.asarray(float(d['agent']['log_alpha']))
out['total_steps'] = np.asarray(int(d['total_steps']))
np.savez_compressed(os.path.join(HERE, 'sac_actor_quadrotor_3D_track.npz'), **out)
print({k: v.shape for k, v in out.items()})
