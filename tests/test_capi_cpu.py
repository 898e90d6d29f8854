"""CPU-only checks of the C-ABI library and the host-side config compiler (no kernel launches)."""
import ctypes as C
import glob
import json
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='module')
def lib():
    from safe_control_gym_amd import _lib as L
    L.build()
    return L.lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, 'include', 'scg_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = set(re.findall(r'\b(scg_[a-z_0-9]+)\s*\(', hdr))
    assert {'scg_create', 'scg_step', 'scg_reset', 'scg_gae', 'scg_rollout_random'} <= names
    for n in names:
        assert hasattr(lib, n), f'libscg_hip.so does not export {n}'


def test_learner_libraries_export_every_declared_symbol():
    """include/scg_learn.h (PPO learner) and include/scg_sac.h (fused SAC step): per-shape libraries, every declared entry
    point exported, the shape query answers with the shape they were compiled for, the ctypes structs have the C layout
    (offset of the last field + its size == what the header's field list implies)."""
    from safe_control_gym_amd import _learn, _sac
    for mod, hdr_name, shape_fn, shape in ((_learn, 'scg_learn.h', 'scg_learn_shape', (12, 32, 2, 'tanh')), (_sac, 'scg_sac.h', 'scg_sac_shape', (6, 32, 2, 'relu'))):
        so = mod.build(*shape)
        D = C.CDLL(so)
        hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', hdr_name)).read(), flags=re.S)
        names = set(re.findall(r'\b(scg_[a-z_0-9]+)\s*\(', hdr))
        assert len(names) >= 6
        for n in names:
            assert hasattr(D, n), f'{os.path.basename(so)} does not export {n}'
        got = [C.c_int32() for _ in range(4)]
        getattr(D, shape_fn)(*[C.byref(v) for v in got])
        assert tuple(v.value for v in got) == (shape[0], shape[1], shape[2], _learn.ACTS[shape[3]])
    # scg_sac_args: pointers 8-byte aligned, three 24-byte layouts, floats packed as declared
    assert _sac.SacArgs.d_workspace.offset % 8 == 0 and C.sizeof(_sac.SacArgs) % 8 == 0
    assert _sac.SacArgs.actor.offset == 6 * 8 and _sac.SacArgs.n_actor.offset == 6 * 8 + 3 * 24
    assert _sac.SacArgs.act_high.offset - _sac.SacArgs.act_low.offset == 16


def test_struct_layouts_agree(lib):
    from safe_control_gym_amd import _lib as L
    assert lib.scg_sizeof_config() == C.sizeof(L.Config)
    assert lib.scg_sizeof_step_out() == C.sizeof(L.StepOut)
    assert lib.scg_abi_version() == L.SCG_ABI_VERSION


def _cases():
    for p in sorted(glob.glob(os.path.join(GOLDEN, 'rollout_*.npz'))):
        yield os.path.basename(p)[len('rollout_'):-4]


@pytest.mark.parametrize('name', list(_cases()))
def test_env_spec_matches_reference_fixtures(lib, name):
    """Host-side derived quantities (X_GOAL, U_GOAL, spaces, action bounds, #constraints) equal the values
    recorded from the reference's env objects."""
    from safe_control_gym_amd import _lib as L
    from safe_control_gym_amd.env_config import EnvSpec
    g = np.load(os.path.join(GOLDEN, f'rollout_{name}.npz'))
    meta = json.loads(str(g['meta_json']))
    cfg = dict(meta['config'])
    cfg.pop('seed', None)
    spec = EnvSpec(meta['task'], cfg)
    np.testing.assert_allclose(np.atleast_2d(spec.X_GOAL), np.atleast_2d(g['x_goal']), rtol=0, atol=1e-13)
    np.testing.assert_allclose(spec.U_GOAL, g['u_goal'], rtol=1e-15)
    np.testing.assert_array_equal(spec.state_space.low, g['state_space_low'])
    np.testing.assert_array_equal(spec.state_space.high, g['state_space_high'])
    np.testing.assert_array_equal(spec.observation_space.low, g['observation_space_low'])
    np.testing.assert_array_equal(spec.action_space.low, g['action_space_low'])
    np.testing.assert_array_equal(spec.action_space.high, g['action_space_high'])
    np.testing.assert_allclose(np.asarray(spec.physical_action_bounds[0], dtype=float), g['physical_action_low'], rtol=0)
    assert len(spec.con_rows) == g['c_values'].shape[-1]
    assert spec.obs_dim == g['obs'].shape[-1]
    c, xg = spec.to_c_config(64, L.F32, 1)
    dims = [C.c_int32() for _ in range(5)]
    assert lib.scg_dims(C.byref(c), *[C.byref(d) for d in dims]) == 0
    assert dims[0].value == spec.nx and dims[1].value == spec.nu and dims[2].value == spec.obs_dim
    nb = C.c_size_t(0)
    assert lib.scg_workspace_bytes(C.byref(c), C.byref(nb)) == 0 and nb.value > 0


def test_invalid_config_is_reported_not_aborted(lib):
    from safe_control_gym_amd import _lib as L
    from safe_control_gym_amd.env_config import EnvSpec
    spec = EnvSpec('cartpole', {})
    c, _ = spec.to_c_config(8, L.F32, 0)
    nb = C.c_size_t(0)
    c.num_envs = 0
    assert lib.scg_workspace_bytes(C.byref(c), C.byref(nb)) == -1
    assert b'num_envs' in lib.scg_last_error()
    c.num_envs = 8
    c.abi_version = 99
    assert lib.scg_workspace_bytes(C.byref(c), C.byref(nb)) == -1
    assert b'abi_version' in lib.scg_last_error()
    c.abi_version = L.SCG_ABI_VERSION
    c.integrator = 7
    assert lib.scg_workspace_bytes(C.byref(c), C.byref(nb)) == -1
    assert b'integrator' in lib.scg_last_error()
    c.integrator = L.INT_RK4                      # the prior-model integrator refuses dynamics disturbances
    assert lib.scg_workspace_bytes(C.byref(c), C.byref(nb)) == 0
    c.adversary_channel = 1
    assert lib.scg_workspace_bytes(C.byref(c), C.byref(nb)) == -1
    assert b'prior model' in lib.scg_last_error()
    assert lib.scg_step(None, None, None, None, None) == -1
    assert lib.scg_gae(L.F32, None, None, None, None, None, None, None, 4, 4, 0.99, 0.95, 1, None) == -1


def test_env_spec_error_behaviour_mirrors_reference():
    from safe_control_gym_amd.env_config import EnvSpec
    with pytest.raises(ValueError):      # benchmark_env.py:141-142
        EnvSpec('cartpole', {'ctrl_freq': 15, 'pyb_freq': 100})
    with pytest.raises(ValueError):      # quadrotor.py:200-201
        EnvSpec('quadrotor', {'info_mse_metric_state_weight': [1, 2]})
    with pytest.raises(ValueError):      # disturbances.py:211
        EnvSpec('cartpole', {'disturbances': {'action': [{'disturbance_func': 'white_noise', 'std': 1}]}})
    with pytest.raises(AssertionError):  # constraints.py:661
        EnvSpec('quadrotor', {'constraints': [{'constraint_form': 'abs_bound', 'constrained_variable': 'state', 'bound': 1.0}]})
    # quadrotor ignores the YAML randomisation tables (quadrotor.py:208,233) unless the extension is on
    info = {'init_x': {'distrib': 'uniform', 'low': -2, 'high': 2}}
    assert EnvSpec('quadrotor', {'init_state_randomization_info': info}).init_rand_info['init_x']['low'] == -0.5
    assert EnvSpec('quadrotor', {'init_state_randomization_info': info,
                                 'respect_randomization_info': True}).init_rand_info['init_x']['low'] == -2


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under safe_control_gym_amd/ may reference it."""
    for p in glob.glob(os.path.join(ROOT, 'safe_control_gym_amd', '**', '*'), recursive=True):
        if os.path.isfile(p) and p.endswith(('.py', '.h', '.hip', '.cpp')):
            src = open(p, errors='ignore').read()
            assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), p


def test_hip_vec_env_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from safe_control_gym_amd import _lib as L
    from safe_control_gym_amd.vec_env import HipVecEnv
    with pytest.raises(L.ScgError):
        HipVecEnv('cartpole', 4)


def test_rk4_mode_of_the_oracle_is_the_analytic_prior_model():
    """oracle/symbolic.py (checker of the SCG_INT_RK4 kernels) == safe_control_gym_amd.symbolic.AnalyticModel.fd_func
    (the product's NumPy prior model handed to model-based controllers), one RK4 step per control period."""
    import numpy as np
    from oracle.envs import make_oracle_env, make_rng
    from safe_control_gym_amd.env_config import EnvSpec
    from safe_control_gym_amd.registration import load_task
    from safe_control_gym_amd.symbolic import AnalyticModel
    for task in ('quadrotor_2D_track', 'cartpole_stab', 'quadrotor_3D_track'):
        env_id, cfg = load_task(task)
        cfg = dict(cfg, integrator='rk4')
        env = make_oracle_env(env_id, 4, make_rng('philox', 4, 3), **cfg)
        env.reset()
        x0 = env.state.copy()
        env.step(np.random.default_rng(1).uniform(-1, 1, (4, env.action_dim)))
        am = AnalyticModel(env_id, EnvSpec(env_id, cfg))
        u = env.current_clipped_action.reshape(4, -1)
        xf = np.stack([am.fd_func(x0[i], u[i], substeps=1)['xf'].reshape(-1) for i in range(4)])
        # (envs that terminated were not reset: the oracle env has no auto-reset)
        np.testing.assert_allclose(env.state, xf, rtol=1e-12, atol=1e-13)
        spec = EnvSpec(env_id, cfg)
        c, _ = spec.to_c_config(4, 1, 0)
        assert c.integrator == 1 and c.substeps == 1 and abs(c.pyb_dt - spec.CTRL_TIMESTEP) < 1e-15


def test_staged_reference_python_stays_out_of_history_and_out_of_the_product():
    """tools/stage_reference.py can copy the reference's Python to oracle/_ref/reference for LOCAL checker runs (the reference's own
    PPO / SAC classes on HipVecEnv where a GPU and a checkout coexist): untracked scratch — git-ignored like the built oracle library —
    and gpurun-ignored: the reference's Python does not travel to the GPU box in any form.  The compiled oracle library next to it does
    travel (oracle/_ref/ itself must not be gpurun-ignored).  Read only by the checker side."""
    import subprocess
    ignore = open(os.path.join(ROOT, '.gitignore')).read().split()
    assert 'oracle/_ref/' in ignore
    gpurunignore = os.path.join(ROOT, '.gpurunignore')
    lines = [ln.strip() for ln in open(gpurunignore) if ln.strip() and not ln.lstrip().startswith('#')]
    assert 'oracle/_ref/reference/' in lines                                 # the reference's Python stays here
    assert not any(ln.rstrip('/') in ('oracle', 'oracle/_ref') for ln in lines)     # the built oracle library travels
    if os.path.isdir(os.path.join(ROOT, '.git')):
        tracked = subprocess.run(['git', 'ls-files', 'oracle/_ref'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        assert tracked == '', tracked
    for d, _, files in os.walk(os.path.join(ROOT, 'safe_control_gym_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(d, f)).read()
                assert 'oracle/_ref' not in src and 'reference_root' not in src and 'ref_stubs' not in src, os.path.join(d, f)
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    assert 'ref_stubs' not in bench and 'reference_root' not in bench and '/root/reference' not in bench


def test_bench_quotes_pmc_numbers_only_for_the_kernel_sources_they_were_measured_on(tmp_path, monkeypatch):
    """bench.py's roofline.traffic / valu_issue come from a committed rocprofv3 --pmc file that names the hash of the kernel sources it
    was measured on; with other sources in the tree the line must carry null + the reason, never the stale number."""
    import json
    import bench
    from safe_control_gym_amd import _lib
    good = {'_meta': {'source_hash': f'0x{_lib.source_hash():016x}'},
            'quadrotor_2D_track/f32/65536': {'traffic_bytes_per_launch': 14.5e6, 'valu_instructions_per_wave': 900.0}}
    os.makedirs(tmp_path / 'profiles')
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    for meta_hash, expect in ((good['_meta']['source_hash'], 14.5e6), ('0x0123456789abcdef', None)):
        good['_meta']['source_hash'] = meta_hash
        (tmp_path / 'profiles' / bench.TRAFFIC_FILE).write_text(json.dumps(good))
        bench._PMC_CACHE.clear()
        r = bench.roofline_of('quadrotor_2D_track', 'f32', 65536, 4.2)
        assert r['traffic'] == expect
        if expect is None:
            assert 'dropped' in r['traffic_source'] and r['valu_issue'] is None
        else:
            # one wave per SIMD: the wave's own issue limit, 4.8 clocks per instruction at 2.396 GHz (profiles/r05_issue_rate.txt)
            assert r['valu_issue']['frac'] == pytest.approx(900.0 * 4.8 / 2.396e3 / 4.2)
            assert r['frac_by_clock']['in_graph_hip_events']['frac'] == pytest.approx(r['frac'])
        assert r['frac'] == pytest.approx(187 * 65536 / 4.2e-6 / 8e12)
    bench._PMC_CACHE.clear()


def test_bench_names_the_launch_geometry_and_gates_the_chain_model_on_the_source_hash(tmp_path, monkeypatch):
    """bench.py's `roofline.kernel` names the launch geometry scg_step picks for the shard size (scg_kernels.hip defaults, scg_set_step_launch),
    and `chain_latency` is quoted only from a model file computed on THESE kernel sources."""
    import json
    import bench
    from safe_control_gym_amd import _lib
    assert bench.launch_geometry(16384, True).startswith('step_kernel (one wave per 64 envs')          # (no split launch any more)
    assert bench.launch_geometry(65536, True).startswith('step_kernel (one wave per 64 envs')
    assert 'step_wide_kernel' in bench.launch_geometry(16777216, True) and 'step_wide_kernel' not in bench.launch_geometry(4194304, True)
    assert 'generic library' in bench.launch_geometry(65536, False)
    assert 'step_wsback_kernel' in bench.launch_geometry(262144, True) and 'step_wsback_kernel' in bench.launch_geometry(131072, True)
    assert 'step_wsback_kernel' not in bench.launch_geometry(262144, True, 'cartpole_stab') and 'step_wsback_kernel' not in bench.launch_geometry(1048576, True)
    src = open(os.path.join(ROOT, 'safe_control_gym_amd', 'csrc', 'scg_kernels.hip')).read()
    assert 'SCG_SPLIT_MAX_ENVS' not in src and f'#define SCG_WIDE_MIN_ENVS {bench.LAUNCH_WIDE_MIN}' in src
    assert f'#define SCG_WSBACK_MIN_ENVS {bench.LAUNCH_WSBACK[0]}' in src and f'#define SCG_WSBACK_MAX_ENVS {bench.LAUNCH_WSBACK[1]}' in src
    os.makedirs(tmp_path / 'profiles')
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    model = {'_meta': {'source_hash': f'0x{_lib.source_hash():016x}'},
             'cartpole_stab': {'chain_us_per_control_step': 3.7, 'issue_limit_us_per_control_step': 2.7, 'dependent_instructions_per_substep': 21.0}}
    (tmp_path / 'profiles' / bench.CHAIN_FILE).write_text(json.dumps(model))
    got = bench.chain_latency_of('cartpole_stab', 5.5)
    assert got['frac_of_launch'] == pytest.approx(3.7 / 5.5) and got['issue_limit_us'] == 2.7
    model['_meta']['source_hash'] = '0x0123456789abcdef'
    (tmp_path / 'profiles' / bench.CHAIN_FILE).write_text(json.dumps(model))
    assert 'dropped' in bench.chain_latency_of('cartpole_stab', 5.5)


def test_bench_learner_kernel_sums_are_gated_on_the_source_hashes(tmp_path, monkeypatch):
    """bench.py quotes a learner iteration's rocprofv3 kernel sum (ppo.iteration_ms.kernel_sum, sac.roofline.gradient_step_us_rocprof)
    only from a profiles/r06_learner_kernel_sums.json measured on THESE kernel sources; the flop counts are the documented formulas."""
    import json
    import bench
    from safe_control_gym_amd import _learn, _lib, _sac
    assert bench.mlp_flops(12, 128, 2) == 2 * (12 * 128 + 128 * 128 + 128 * 2) and bench.mlp_flops(12, 128, 1, True) == 3 * bench.mlp_flops(12, 128, 1)
    os.makedirs(tmp_path / 'profiles')
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    hashes = {'env': f'0x{_lib.source_hash():016x}', 'learn': f'0x{_learn.source_hash():016x}', 'sac': f'0x{_sac.source_hash():016x}'}
    d = {'_meta': {'source_hashes': hashes}, 'ppo/65536/48x16256': {'kernel_sum_ms_per_iteration': 4.9}, 'sac/4096/16': {'gradient_step_us': 109.0}}
    (tmp_path / 'profiles' / bench.LEARNER_SUMS_FILE).write_text(json.dumps(d))
    e, src = bench.learner_kernel_sum('ppo/65536/48x16256')
    assert e['kernel_sum_ms_per_iteration'] == 4.9 and 'rocprofv3' in src
    assert bench.learner_kernel_sum('ppo/16384/64x16256')[0] is None
    d['_meta']['source_hashes'] = dict(hashes, sac='0x0')               # another SAC library: the PPO entry stays, the SAC entry goes
    (tmp_path / 'profiles' / bench.LEARNER_SUMS_FILE).write_text(json.dumps(d))
    assert bench.learner_kernel_sum('ppo/65536/48x16256')[0] is not None and bench.learner_kernel_sum('sac/4096/16')[0] is None
    d['_meta']['source_hashes'] = dict(hashes, learn='0x0')
    (tmp_path / 'profiles' / bench.LEARNER_SUMS_FILE).write_text(json.dumps(d))
    assert bench.learner_kernel_sum('ppo/65536/48x16256')[0] is None and 'dropped' in bench.learner_kernel_sum('ppo/65536/48x16256')[1]
