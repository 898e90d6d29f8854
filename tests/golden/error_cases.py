"""Invalid / unusual task configs for the error-behaviour comparison (tests/golden/make_config_errors.py runs them through the REFERENCE,
tests/test_config_errors.py through this package's EnvSpec).  Each entry is an update applied to a valid base config of the system."""
import copy

from tests.config_fuzz import fuzz_config

SYSTEMS = ('cartpole', 'quadrotor_1D', 'quadrotor_2D', 'quadrotor_3D')


def base(system):
    env_id, cfg = fuzz_config(system, 1)
    cfg = dict(cfg)
    for k in ('constraints', 'disturbances', 'adversary_disturbance'):
        cfg.pop(k, None)
    if system == 'quadrotor_1D':
        cfg.update(randomized_init=False)                     # (keeps EnvSpec's 1-D lateral-drift guard out of the dynamics-disturbance cases)
    return env_id, cfg


def mutated(system, name):
    env_id, cfg = base(system)
    cfg = copy.deepcopy(cfg)
    cfg.update(copy.deepcopy(MUT[name]))
    return env_id, cfg


MUT = {
 'cost_unknown': dict(cost='foo'),
 'task_unknown': dict(task='foo'),
 'quad_type_5': dict(quad_type=5),
 'traj_type_unknown': dict(task='traj_tracking', task_info={'trajectory_type':'spiral','num_cycles':1,'trajectory_plane':'xz','trajectory_position_offset':[0,1],'trajectory_scale':1}),
 'traj_plane_unknown': dict(task='traj_tracking', task_info={'trajectory_type':'circle','num_cycles':1,'trajectory_plane':'ab','trajectory_position_offset':[0,1],'trajectory_scale':1}),
 'freq_not_divisible': dict(ctrl_freq=70, pyb_freq=1000),
 'pyb_lt_ctrl': dict(ctrl_freq=100, pyb_freq=50),
 'con_no_form': dict(constraints=[{'constrained_variable':'state'}]),
 'con_unknown_form': dict(constraints=[{'constraint_form':'foo','constrained_variable':'state'}]),
 'con_not_dict': dict(constraints=['default_constraint']),
 'con_var_unknown': dict(constraints=[{'constraint_form':'default_constraint','constrained_variable':'foo'}]),
 'con_var_unknown_bounded': dict(constraints=[{'constraint_form':'bounded_constraint','constrained_variable':'foo','lower_bounds':[0],'upper_bounds':[1],'active_dims':[0]}]),
 'con_bounded_int_active': dict(constraints=[{'constraint_form':'bounded_constraint','constrained_variable':'state','lower_bounds':[0,1],'upper_bounds':[1,2],'active_dims':1}]),
 'con_bounded_active_len': dict(constraints=[{'constraint_form':'bounded_constraint','constrained_variable':'state','lower_bounds':[0,1],'upper_bounds':[1,2],'active_dims':[1]}]),
 'con_linear_A_shape': dict(constraints=[{'constraint_form':'linear_constraint','constrained_variable':'state','A':[[1.0,2.0,3.0]],'b':[1.0],'active_dims':[0,1]}]),
 'con_tolerance_ok': dict(constraints=[{'constraint_form':'bounded_constraint','constrained_variable':'state','lower_bounds':[0],'upper_bounds':[1],'active_dims':[1],'tolerance':[0.1,0.1]}]),
 'con_abs_tolerance': dict(cost='rl_reward', constraints=[{'constraint_form':'abs_bound','constrained_variable':'state','bound':1.0,'active_dims':[0],'tolerance':[0.1,0.2]}]),
 'rand_choice_p': dict(randomized_init=True, init_state_randomization_info={'init_x':{'distrib':'choice','a':[0.1,0.2],'p':[0.5,0.5]}}),
 'rand_exponential': dict(randomized_init=True, init_state_randomization_info={'init_x':{'distrib':'exponential','scale':0.1}}),
 'con_var_both': dict(constraints=[{'constraint_form':'default_constraint','constrained_variable':'input_and_state'}]),
 'con_bounded_len': dict(constraints=[{'constraint_form':'bounded_constraint','constrained_variable':'state','lower_bounds':[0,0],'upper_bounds':[1,1,1],'active_dims':[0,1]}]),
 'con_bounded_no_bounds': dict(constraints=[{'constraint_form':'bounded_constraint','constrained_variable':'state','active_dims':[0]}]),
 'con_quad_b_int': dict(constraints=[{'constraint_form':'quadratic_constraint','constrained_variable':'state','P':[[1.0]],'b':1,'active_dims':[0]}]),
 'con_quad_P_shape': dict(constraints=[{'constraint_form':'quadratic_constraint','constrained_variable':'state','P':[[1.0,0,0]],'b':1.0,'active_dims':[0,1]}]),
 'con_active_oor': dict(constraints=[{'constraint_form':'bounded_constraint','constrained_variable':'input','lower_bounds':[0],'upper_bounds':[1],'active_dims':[7]}]),
 'con_active_dup': dict(constraints=[{'constraint_form':'bounded_constraint','constrained_variable':'state','lower_bounds':[0,0],'upper_bounds':[1,1],'active_dims':[1,1]}]),
 'con_active_float': dict(constraints=[{'constraint_form':'bounded_constraint','constrained_variable':'state','lower_bounds':[0],'upper_bounds':[1],'active_dims':[1.0]}]),
 'con_linear_b_len': dict(constraints=[{'constraint_form':'linear_constraint','constrained_variable':'input','A':[[1.0]*1],'b':[1.0,2.0],'active_dims':[0]}]),
 'con_default_active': dict(constraints=[{'constraint_form':'default_constraint','constrained_variable':'state','active_dims':[0]}]),
 'con_default_lb_len': dict(constraints=[{'constraint_form':'default_constraint','constrained_variable':'state','lower_bounds':[0.0]}]),
 'con_abs_bound': dict(constraints=[{'constraint_form':'abs_bound','constrained_variable':'state','bound':[1.0],'active_dims':[0]}]),
 'con_abs_bound_scalar': dict(constraints=[{'constraint_form':'abs_bound','constrained_variable':'state','bound':1.0,'active_dims':[0]}]),
 'con_abs_bound_quadcost': dict(cost='quadratic', constraints=[{'constraint_form':'abs_bound','constrained_variable':'state','bound':1.0,'active_dims':[0]}]),
 'con_extra_kw': dict(constraints=[{'constraint_form':'default_constraint','constrained_variable':'state','foo':1}]),
 'con_tolerance_len': dict(constraints=[{'constraint_form':'default_constraint','constrained_variable':'input','tolerance':[0.1,0.1,0.1,0.1,0.1,0.1,0.1]}]),
 'dist_unknown_mode': dict(disturbances={'wind':[{'disturbance_func':'white_noise','std':0.1}]}),
 'dist_unknown_func': dict(disturbances={'action':[{'disturbance_func':'pink_noise','std':0.1}]}),
 'dist_no_func': dict(disturbances={'action':[{'std':0.1}]}),
 'dist_not_list': dict(disturbances={'action':{'disturbance_func':'white_noise','std':0.1}}),
 'dist_extra_kw': dict(disturbances={'observation':[{'disturbance_func':'white_noise','std':0.1,'foo':2}]}),
 'dist_mask_len': dict(disturbances={'observation':[{'disturbance_func':'white_noise','std':0.1,'mask':[1,0]}]}),
 'dist_std_len': dict(disturbances={'observation':[{'disturbance_func':'white_noise','std':[0.1,0.2,0.3]}]}),
 'dist_impulse': dict(disturbances={'dynamics':[{'disturbance_func':'impulse','magnitude':1.0,'step_offset':2,'duration':3,'decay_rate':0.9}]}),
 'dist_step': dict(disturbances={'action':[{'disturbance_func':'step','magnitude':1.0,'step_offset':2}]}),
 'dist_uniform': dict(disturbances={'action':[{'disturbance_func':'uniform','low':[-0.1],'high':[0.1]}]}),
 'dist_periodic': dict(disturbances={'action':[{'disturbance_func':'periodic','scale':0.1,'frequency':1.0}]}),
 'adv_unknown': dict(adversary_disturbance='wind'),
 'init_state_str': dict(init_state='zero'),
 'init_state_list': dict(init_state=[0.0]*4),
 'init_state_unknown_key': dict(init_state={'init_foo':1.0}),
 'rand_unknown_distrib': dict(randomized_init=True, init_state_randomization_info={'init_x':{'distrib':'foo','low':-1,'high':1}}),
 'rand_missing_arg': dict(randomized_init=True, init_state_randomization_info={'init_x':{'distrib':'uniform','lo':-1}}),
 'rand_unknown_key': dict(randomized_init=True, init_state_randomization_info={'init_foo':{'distrib':'uniform','low':-1,'high':1}}),
 'rand_no_distrib': dict(randomized_init=True, init_state_randomization_info={'init_x':{'low':-1,'high':1}}),
 'inertial_unknown_key': dict(inertial_prop={'foo':1.0}),
 'inertial_rand_unknown_key': dict(randomized_inertial_prop=True, inertial_prop_randomization_info={'foo':{'distrib':'uniform','low':0.1,'high':0.2}}),
 'rew_state_weight_len': dict(rew_state_weight=[1.0,2.0,3.0]),
 'rew_act_weight_len': dict(rew_act_weight=[1.0,2.0,3.0,4.0,5.0]),
 'rew_state_weight_str': dict(rew_state_weight='a'),
 'episode_len_zero': dict(episode_len_sec=0),
 'obs_goal_horizon_neg': dict(task='traj_tracking', obs_goal_horizon=-1),
 'unknown_kwarg': dict(foo=1),
 'physics_dyn': dict(physics='dyn'),
 'physics_unknown': dict(physics='foo'),
 'norm_act_scale_neg': dict(norm_act_scale=-0.1),
 'stab_goal_short': dict(task='stabilization', task_info={'stabilization_goal':[], 'stabilization_goal_tolerance':0.0}),
 'task_info_none_track': dict(task='traj_tracking', task_info=None),
 'task_info_none_stab': dict(task='stabilization', task_info=None),
 'gui': dict(gui=True),
 'seed_neg': dict(seed=-3),
 'seed_str': dict(seed='a'),
 'seed_float': dict(seed=1.5),
}

