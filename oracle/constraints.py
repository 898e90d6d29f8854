"""Constraint evaluation, batched.  ORACLE — test infrastructure only.

Follows envs/constraints.py: Constraint.get_value :97-109 (round to ``decimals``),
is_violated :111-131, get_env_constraint_var :155-174 (state, or the NOISY UNCLIPPED
physical action), QuadraticConstraint :181-231, LinearConstraint :234-283 (A and b cast
to float32, evaluated in float64), BoundedConstraint :286-329, DefaultConstraint
:332-397, SymmetricStateConstraint :400-455, ConstraintList :473-636,
create_constraint_list :647-665.

INPUT_AND_STATE constraints are not restated: upstream get_value() builds
``np.array((state, action), ndmin=1)`` from two different-length vectors, which is an
error on NumPy >= 1.24, so no shipped config can use them.
"""
import numpy as np

STATE, INPUT = 'state', 'input'


class _Constraint:
    def __init__(self, env, constrained_variable, strict=False, active_dims=None,
                 tolerance=None, decimals=8):
        self.constrained_variable = str(getattr(constrained_variable, 'value', constrained_variable)).lower()
        if self.constrained_variable == STATE:
            self.dim = env.state_dim
        elif self.constrained_variable == INPUT:
            self.dim = env.action_dim
        else:
            raise NotImplementedError('oracle: only STATE / INPUT constraints are restated.')
        self.strict = strict
        self.decimals = decimals
        if active_dims is not None:
            if isinstance(active_dims, int):
                active_dims = [active_dims]
            self.constraint_filter = np.eye(self.dim)[active_dims]      # :77
            self.dim = len(active_dims)
        else:
            self.constraint_filter = np.eye(self.dim)
        self.tolerance = None if tolerance is None else np.array(tolerance, ndmin=1)

    def raw(self, v):
        raise NotImplementedError

    def get_value(self, state, noisy_action):
        """(N, num_constraints); constraints.py:97-109."""
        v = state if self.constrained_variable == STATE else noisy_action
        return np.round(self.raw(np.asarray(v, dtype=np.float64)), decimals=self.decimals)

    def is_violated(self, c_value):
        """(N,) bool; constraints.py:111-131."""
        if self.strict:
            return np.any(c_value >= 0.0, axis=1)
        return np.any(c_value > 0.0, axis=1)


class LinearConstraint(_Constraint):
    def __init__(self, env, A, b, constrained_variable, **kw):
        super().__init__(env, constrained_variable, **kw)
        self.A = np.asarray(A, dtype=np.float32).reshape(-1, self.dim)      # :267
        self.b = np.asarray(b, dtype=np.float32).reshape(-1)               # :268
        self.num_constraints = self.A.shape[0]

    def raw(self, v):
        # sym_func: A @ filter @ x - b  (:273), float32 operands promoted to float64.
        AF = self.A @ self.constraint_filter
        return v @ AF.T - self.b


class BoundedConstraint(LinearConstraint):
    def __init__(self, env, lower_bounds, upper_bounds, constrained_variable, **kw):
        self.lower_bounds = np.array(lower_bounds, ndmin=1)
        self.upper_bounds = np.array(upper_bounds, ndmin=1)
        dim = self.lower_bounds.shape[0]
        A = np.vstack((-np.eye(dim), np.eye(dim)))                          # :320
        b = np.hstack((-self.lower_bounds, self.upper_bounds))             # :321
        super().__init__(env, A, b, constrained_variable, **kw)


class DefaultConstraint(BoundedConstraint):
    def __init__(self, env, constrained_variable, lower_bounds=None, upper_bounds=None,
                 strict=False, tolerance=None, decimals=8):
        cv = str(getattr(constrained_variable, 'value', constrained_variable)).lower()
        if cv == STATE:
            low, high = env.state_space_low, env.state_space_high          # float32 (:366-369)
        elif cv == INPUT:
            low = np.asarray(env.physical_action_bounds[0], dtype=np.float32)   # :371-373
            high = np.asarray(env.physical_action_bounds[1], dtype=np.float32)
        else:
            raise NotImplementedError('[ERROR] DefaultConstraint can only be of type STATE or INPUT')
        ub = high if upper_bounds is None else np.array(upper_bounds, ndmin=1)
        lb = low if lower_bounds is None else np.array(lower_bounds, ndmin=1)
        assert len(ub) == len(high) and len(lb) == len(low)
        super().__init__(env, lower_bounds=lb.astype(np.float64), upper_bounds=ub.astype(np.float64),
                         constrained_variable=cv, strict=strict, active_dims=None,
                         tolerance=tolerance, decimals=decimals)


class SymmetricStateConstraint(BoundedConstraint):
    """'abs_bound' (cartpole only), constraints.py:400-455."""

    def __init__(self, env, constrained_variable, bound, strict=False, active_dims=None,
                 tolerance=None, decimals=8, **kw):
        bound = np.array(bound, ndmin=1)
        self.bound = bound
        super().__init__(env, lower_bounds=-bound, upper_bounds=bound,
                         constrained_variable=constrained_variable, strict=strict,
                         active_dims=active_dims, tolerance=tolerance, decimals=decimals)
        self.num_constraints = self.bound.shape[0]

    def get_value(self, state, noisy_action):
        # :445-447 — uses the float64 ``bound`` (not the float32 b of the parent).
        return np.round(np.abs(state @ self.constraint_filter.T) - self.bound, decimals=self.decimals)


class QuadraticConstraint(_Constraint):
    def __init__(self, env, P, b, constrained_variable, **kw):
        super().__init__(env, constrained_variable, **kw)
        self.P = np.array(P, ndmin=1, dtype=np.float64).reshape(self.dim, self.dim)
        self.b = float(b)
        self.num_constraints = 1

    def raw(self, v):
        y = v @ self.constraint_filter.T
        return (np.einsum('ni,ij,nj->n', y, self.P, y) - self.b)[:, None]   # :222


CONSTRAINT_FORMS = {
    'linear_constraint': LinearConstraint,
    'quadratic_constraint': QuadraticConstraint,
    'bounded_constraint': BoundedConstraint,
    'default_constraint': DefaultConstraint,
    'abs_bound': SymmetricStateConstraint,      # cartpole.py:68-71
}


class ConstraintList:
    """constraints.py:473-636."""

    def __init__(self, constraints):
        self.constraints = constraints
        self.lengths = [c.num_constraints for c in constraints]
        self.num_constraints = sum(self.lengths)
        self.state_constraints = [c for c in constraints if c.constrained_variable == STATE]
        self.input_constraints = [c for c in constraints if c.constrained_variable == INPUT]
        self.num_state_constraints = sum(c.num_constraints for c in self.state_constraints)

    def get_values(self, state, noisy_action, only_state=False):
        cons = self.state_constraints if only_state else self.constraints
        return np.concatenate([c.get_value(state, noisy_action) for c in cons], axis=1)

    def is_violated(self, c_value):
        """Any constraint violated, per env; :591-609."""
        flags = np.zeros(c_value.shape[0], dtype=bool)
        off = 0
        for c, n in zip(self.constraints, self.lengths):
            flags |= c.is_violated(c_value[:, off:off + n])
            off += n
        return flags


def create_constraint_list(specs, env):
    """constraints.py:647-665."""
    out = []
    for spec in specs:
        form = spec['constraint_form']
        if form == 'abs_bound' and env.NAME != 'cartpole':
            raise AssertionError('abs_bound is a cartpole-only constraint')
        cfg = {k: v for k, v in spec.items() if k != 'constraint_form'}
        out.append(CONSTRAINT_FORMS[form](env, **cfg))
    return ConstraintList(out)
