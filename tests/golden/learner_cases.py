"""Hyper-parameter corners of the learner fixtures: shared by make_learner_variants.py (runs the reference) and
tests/test_learner_golden.py (runs this repo).  Names of controllers/ppo/ppo.yaml / controllers/sac/sac.yaml."""
PPO_CASES = {
    'clipped_relu': dict(obs=4, act=1, T=12, N=8, use_gae=True, kw=dict(hidden_dim=64, use_clipped_value=True, clip_param=0.1, target_kl=0.5,
                         entropy_coef=0.0, actor_lr=1e-3, critic_lr=3e-3, opt_epochs=4, mini_batch_size=32, activation='relu')),
    'gate_closes_leaky': dict(obs=24, act=4, T=10, N=16, use_gae=False, kw=dict(hidden_dim=32, use_clipped_value=False, clip_param=0.2,
                              target_kl=0.0005, entropy_coef=0.05, actor_lr=2e-2, critic_lr=1e-3, opt_epochs=3, mini_batch_size=40,
                              activation='leaky_relu')),
    'one_whole_batch_epoch': dict(obs=6, act=2, T=8, N=8, use_gae=True, kw=dict(hidden_dim=32, use_clipped_value=True, clip_param=0.3,
                                  target_kl=0.01, entropy_coef=0.01, actor_lr=3e-3, critic_lr=3e-3, opt_epochs=1, mini_batch_size=64,
                                  activation='tanh')),
}
SAC_CASES = {
    'fixed_temperature_tanh': dict(obs=4, low=[-2.0], high=[0.5], kw=dict(hidden_dim=32, gamma=0.99, tau=0.005, init_temperature=0.2,
                                   use_entropy_tuning=False, actor_lr=3e-3, critic_lr=1e-3, entropy_lr=1e-3, activation='tanh')),
    'tuned_four_actions': dict(obs=12, low=[-1.0, -1.0, 0.0, 0.0], high=[1.0, 1.0, 1.0, 3.0], kw=dict(hidden_dim=64, gamma=0.95, tau=0.05,
                               init_temperature=1.0, use_entropy_tuning=True, actor_lr=1e-3, critic_lr=1e-3, entropy_lr=1e-2,
                               activation='relu')),
}
