"""safe_control_gym_amd — MI355X-native batched simulator + rollout engine behind the
safe-control-gym env / vec-env / PPO collector API.  See DESIGN.md."""
__version__ = '0.1.0'
