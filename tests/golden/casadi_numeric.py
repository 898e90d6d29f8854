"""A numeric stand-in for the slice of CasADi the reference's environment code uses.

Fixture generation only (tests/golden/make_golden.py, make_symbolic.py) — never imported at test time or by the product.

CasADi (`casadi ^3.7`, /root/reference/pyproject.toml:23) is not installed here and cannot be (no network).  The
reference builds its prior model as CasADi expressions (envs/gym_pybullet_drones/quadrotor.py:468-604,
envs/gym_control/cartpole.py:390-437, math_and_models/symbolic_systems.py:68-121, controllers/mpc/mpc_utils.py:42-64).
To make THOSE expressions — not a restatement of them — produce numbers, this module implements the same API surface
as a small expression graph over NumPy:

    MX.sym / SX.sym, vertcat, horzcat, blockcat, skew, sin, cos, tan, sqrt, + - * / ** @ .T [], Function (positional and
    keyword calls, symbolic calls = substitution), jacobian (symbolic forward differentiation of the graph, so Hessians
    nest), integrator (the ODE of the graph integrated with SciPy's DOP853 at rtol 1e-12 in place of CVODES).

Semantics follow CasADi's: every value is a 2-D matrix, scalars broadcast, `jacobian(e, v)` is d vec(e) / d vec(v) with
column-major vec, a 1-D NumPy array is a column.
"""
import numpy as np


class DM(np.ndarray):
    """NumPy array with the two conversion methods callers use on casadi.DM."""

    def full(self):
        return np.asarray(self)

    def toarray(self):
        return np.asarray(self)


def _dm(a):
    return np.asarray(a, dtype=float).view(DM)


def _num(x):
    """Numeric operand -> 2-D float array (scalar -> 1x1, 1-D -> column)."""
    a = np.asarray(x, dtype=float)
    if a.ndim == 0:
        return a.reshape(1, 1)
    if a.ndim == 1:
        return a.reshape(-1, 1)
    return a


def _is_zero(v):
    return not isinstance(v, MX) and not np.any(v)


def _shape(v):
    return v.shape if isinstance(v, MX) else _num(v).shape


def _bshape(a, b):
    sa, sb = _shape(a), _shape(b)
    if sa == (1, 1):
        return sb
    if sb == (1, 1) or sa == sb:
        return sa
    raise ValueError(f'shape mismatch {sa} vs {sb}')


class MX:
    __array_ufunc__ = None
    __array_priority__ = 1000

    def __init__(self, op, args=(), shape=(1, 1), name=None, extra=None):
        self.op, self.args, self._shape, self.name, self.extra = op, tuple(args), tuple(shape), name, extra

    @staticmethod
    def sym(name, n=1, m=1):
        return MX('sym', (), (int(n), int(m)), name=name)

    shape = property(lambda self: self._shape)

    def size1(self):
        return self._shape[0]

    def size2(self):
        return self._shape[1]

    def numel(self):
        return self._shape[0] * self._shape[1]

    @property
    def T(self):
        return transpose(self)

    def __repr__(self):
        return f'MX({self.op}{"/" + self.name if self.name else ""}, {self._shape})'

    __hash__ = object.__hash__

    def __add__(self, o): return add(self, o)
    def __radd__(self, o): return add(o, self)
    def __sub__(self, o): return sub(self, o)
    def __rsub__(self, o): return sub(o, self)
    def __mul__(self, o): return mul(self, o)
    def __rmul__(self, o): return mul(o, self)
    def __truediv__(self, o): return div(self, o)
    def __rtruediv__(self, o): return div(o, self)
    def __matmul__(self, o): return matmul(self, o)
    def __rmatmul__(self, o): return matmul(o, self)
    def __neg__(self): return neg(self)
    def __pow__(self, p): return power(self, p)

    def __getitem__(self, idx):
        return index(self, idx)


SX = MX


# ------------------------------------------------------------------ polymorphic operations (MX -> node, numeric -> value)
def _node(op, args, shape, extra=None):
    return MX(op, args, shape, extra=extra)


def add(a, b):
    if _is_zero(a) and _shape(a) in ((1, 1), _shape(b)):
        return b if isinstance(b, MX) else _num(b) + _num(a)
    if _is_zero(b) and _shape(b) in ((1, 1), _shape(a)):
        return a if isinstance(a, MX) else _num(a) + _num(b)
    if isinstance(a, MX) or isinstance(b, MX):
        return _node('add', (a, b), _bshape(a, b))
    return _num(a) + _num(b)


def sub(a, b):
    if _is_zero(b) and _shape(b) in ((1, 1), _shape(a)):
        return a if isinstance(a, MX) else _num(a) - _num(b)
    if isinstance(a, MX) or isinstance(b, MX):
        return _node('sub', (a, b), _bshape(a, b))
    return _num(a) - _num(b)


def mul(a, b):
    if _is_zero(a) or _is_zero(b):
        return np.zeros(_bshape(a, b))
    if isinstance(a, MX) or isinstance(b, MX):
        return _node('mul', (a, b), _bshape(a, b))
    return _num(a) * _num(b)


def div(a, b):
    if _is_zero(a):
        return np.zeros(_bshape(a, b))
    if isinstance(a, MX) or isinstance(b, MX):
        return _node('div', (a, b), _bshape(a, b))
    return _num(a) / _num(b)


def neg(a):
    if isinstance(a, MX):
        return _node('neg', (a,), a.shape)
    return -_num(a)


def power(a, p):
    if isinstance(p, MX):
        raise NotImplementedError('symbolic exponent')
    p = float(np.asarray(p).reshape(-1)[0]) if np.ndim(p) else float(p)
    if isinstance(a, MX):
        return _node('pow', (a,), a.shape, extra=p)
    return _num(a) ** p


def matmul(a, b):
    sa, sb = _shape(a), _shape(b)
    if sa == (1, 1) or sb == (1, 1):            # casadi: a scalar operand makes `@` an element-wise product
        return mul(a, b)
    if sa[1] != sb[0]:
        raise ValueError(f'matmul shape mismatch {sa} @ {sb}')
    if _is_zero(a) or _is_zero(b):
        return np.zeros((sa[0], sb[1]))
    if isinstance(a, MX) or isinstance(b, MX):
        return _node('matmul', (a, b), (sa[0], sb[1]))
    return _num(a) @ _num(b)


mtimes = matmul


def transpose(a):
    if isinstance(a, MX):
        return _node('T', (a,), a.shape[::-1])
    return _num(a).T


def _unary(op, fn):
    def f(a):
        if isinstance(a, MX):
            return _node(op, (a,), a.shape)
        return fn(_num(a))
    f.__name__ = op
    return f


sin = _unary('sin', np.sin)
cos = _unary('cos', np.cos)
tan = _unary('tan', np.tan)
sqrt = _unary('sqrt', np.sqrt)


def vertcat(*xs):
    xs = [x if isinstance(x, MX) else _num(x) for x in xs]
    if not xs:
        return np.zeros((0, 1))
    cols = {_shape(x)[1] for x in xs}
    if len(cols) != 1:
        raise ValueError('vertcat: column mismatch')
    if any(isinstance(x, MX) for x in xs):
        return _node('vertcat', xs, (sum(_shape(x)[0] for x in xs), cols.pop()))
    return np.vstack(xs)


def horzcat(*xs):
    xs = [x if isinstance(x, MX) else _num(x) for x in xs]
    rows = {_shape(x)[0] for x in xs}
    if len(rows) != 1:
        raise ValueError('horzcat: row mismatch')
    if any(isinstance(x, MX) for x in xs):
        return _node('horzcat', xs, (rows.pop(), sum(_shape(x)[1] for x in xs)))
    return np.hstack(xs)


def blockcat(rows):
    return vertcat(*[horzcat(*r) for r in rows])


def skew(v):
    """casadi.skew: the cross-product matrix of a 3-vector."""
    v0, v1, v2 = index(v, 0), index(v, 1), index(v, 2)
    return blockcat([[0.0, neg(v2), v1], [v2, 0.0, neg(v0)], [neg(v1), v0, 0.0]])


def index(a, idx):
    sh = _shape(a)
    if not isinstance(idx, tuple):                      # linear (column-major) index / slice
        flat = np.arange(sh[0] * sh[1]).reshape(sh[::-1]).T     # flat[i, j] = i + j * rows
        sel = flat.reshape(-1, order='F')[idx]
        sel = np.atleast_1d(sel)
        out_shape = (len(sel), 1)
        picks = [(int(k) % sh[0], int(k) // sh[0]) for k in sel]
    else:
        ri = np.atleast_1d(np.arange(sh[0])[idx[0]])
        ci = np.atleast_1d(np.arange(sh[1])[idx[1]])
        out_shape = (len(ri), len(ci))
        picks = [(int(i), int(j)) for j in ci for i in ri]      # column-major order of the result
    if isinstance(a, MX):
        return _node('index', (a,), out_shape, extra=picks)
    a = _num(a)
    return np.array([a[i, j] for i, j in picks]).reshape(out_shape, order='F')


# ------------------------------------------------------------------ evaluation / substitution
_FN = {'add': add, 'sub': sub, 'mul': mul, 'div': div, 'neg': neg, 'matmul': matmul, 'T': transpose,
       'sin': sin, 'cos': cos, 'tan': tan, 'sqrt': sqrt}


def evaluate(node, env, memo=None):
    """Value of `node` with the leaf symbols bound by `env` (values may themselves be MX: substitution)."""
    if not isinstance(node, MX):
        return _num(node)
    memo = {} if memo is None else memo
    if id(node) in memo:
        return memo[id(node)]
    op = node.op
    if op == 'sym':
        if node not in env:
            raise KeyError(f'free symbol {node.name}')
        out = env[node]
    else:
        vals = [evaluate(a, env, memo) for a in node.args]
        if op in _FN:
            out = _FN[op](*vals)
        elif op == 'pow':
            out = power(vals[0], node.extra)
        elif op == 'vertcat':
            out = vertcat(*vals)
        elif op == 'horzcat':
            out = horzcat(*vals)
        elif op == 'index':
            v = vals[0]
            if isinstance(v, MX):
                out = _node('index', (v,), node.shape, extra=node.extra)
            else:
                out = np.array([v[i, j] for i, j in node.extra]).reshape(node.shape, order='F')
        else:
            raise NotImplementedError(op)
    memo[id(node)] = out
    return out


def _leaves(var):
    """(leaf symbol, row, col) for every entry of vec(var); `var` is a symbol or a vertcat of symbols."""
    if not isinstance(var, MX):
        raise TypeError('differentiation variable must be symbolic')
    if var.op == 'sym':
        return [(var, i, j) for j in range(var.shape[1]) for i in range(var.shape[0])]
    if var.op == 'vertcat' and var.shape[1] == 1:
        out = []
        for a in var.args:
            out += _leaves(a)
        return out
    raise NotImplementedError(f'jacobian w.r.t. a {var.op} expression')


def _diff(node, leaf, i, j, memo):
    """d node / d leaf[i, j], same shape as node (MX or numeric zeros)."""
    if not isinstance(node, MX):
        return np.zeros(_shape(node))
    key = id(node)
    if key in memo:
        return memo[key]
    op, a = node.op, node.args
    if op == 'sym':
        out = np.zeros(node.shape)
        if node is leaf:
            out[i, j] = 1.0
    elif op in ('add', 'sub'):
        da, db = _diff(a[0], leaf, i, j, memo), _diff(a[1], leaf, i, j, memo)
        out = add(da, db) if op == 'add' else sub(da, db)
        if _shape(out) != node.shape:               # a scalar operand was broadcast
            out = add(np.zeros(node.shape), out)
    elif op == 'mul':
        out = add(mul(_diff(a[0], leaf, i, j, memo), a[1]), mul(a[0], _diff(a[1], leaf, i, j, memo)))
    elif op == 'div':
        da, db = _diff(a[0], leaf, i, j, memo), _diff(a[1], leaf, i, j, memo)
        out = sub(div(da, a[1]), div(mul(a[0], db), mul(a[1], a[1])))
    elif op == 'neg':
        out = neg(_diff(a[0], leaf, i, j, memo))
    elif op == 'pow':
        p = node.extra
        out = mul(mul(p, power(a[0], p - 1.0)), _diff(a[0], leaf, i, j, memo))
    elif op == 'matmul':
        out = add(matmul(_diff(a[0], leaf, i, j, memo), a[1]), matmul(a[0], _diff(a[1], leaf, i, j, memo)))
    elif op == 'T':
        out = transpose(_diff(a[0], leaf, i, j, memo))
    elif op == 'sin':
        out = mul(cos(a[0]), _diff(a[0], leaf, i, j, memo))
    elif op == 'cos':
        out = neg(mul(sin(a[0]), _diff(a[0], leaf, i, j, memo)))
    elif op == 'tan':
        c = cos(a[0])
        out = div(_diff(a[0], leaf, i, j, memo), mul(c, c))
    elif op == 'sqrt':
        out = div(_diff(a[0], leaf, i, j, memo), mul(2.0, sqrt(a[0])))
    elif op == 'vertcat':
        out = vertcat(*[_diff(x, leaf, i, j, memo) for x in a])
    elif op == 'horzcat':
        out = horzcat(*[_diff(x, leaf, i, j, memo) for x in a])
    elif op == 'index':
        d = _diff(a[0], leaf, i, j, memo)
        if isinstance(d, MX):
            out = _node('index', (d,), node.shape, extra=node.extra)
        else:
            out = np.array([d[r, c] for r, c in node.extra]).reshape(node.shape, order='F')
    else:
        raise NotImplementedError(op)
    memo[key] = out
    return out


def _vec(v):
    """Column-major vec as an (n, 1) value."""
    sh = _shape(v)
    if sh[1] == 1:
        return v
    if isinstance(v, MX):
        return index(v, slice(None))
    return _num(v).reshape(-1, 1, order='F')


def jacobian(expr, var):
    cols = []
    for leaf, i, j in _leaves(var):
        d = _diff(expr, leaf, i, j, {})
        if not isinstance(d, MX) and _shape(d) != _shape(expr):
            d = np.zeros(_shape(expr)) + d
        cols.append(_vec(d))
    return horzcat(*cols)


class Function:
    def __init__(self, name, ins, outs, in_names=None, out_names=None, *a, **k):
        self.name, self.ins, self.outs = name, list(ins), list(outs)
        self.in_names = list(in_names) if in_names else [f'i{k}' for k in range(len(ins))]
        self.out_names = list(out_names) if out_names else [f'o{k}' for k in range(len(outs))]

    def _bind(self, values):
        env = {}
        for sym, val in zip(self.ins, values):
            val = val if isinstance(val, MX) else _num(val)
            leaves = _leaves(sym)
            if sym.op == 'sym':
                if _shape(val) != sym.shape:
                    if isinstance(val, MX) or val.size != sym.numel():
                        raise ValueError(f'{self.name}: argument shape {_shape(val)} for {sym.shape}')
                    val = val.reshape(sym.shape, order='F')
                env[sym] = val
            else:
                if _shape(val)[0] * _shape(val)[1] != len(leaves):
                    raise ValueError(f'{self.name}: argument has {_shape(val)} entries, expected {len(leaves)}')
                flat = _vec(val)
                for k, (leaf, i, j) in enumerate(leaves):
                    assert leaf.shape == (1, 1)
                    env[leaf] = index(flat, k)
        return env

    def __call__(self, *args, **kwargs):
        if kwargs:
            values = [kwargs[n] for n in self.in_names]
        else:
            values = list(args)
        if len(values) != len(self.ins):
            raise TypeError(f'{self.name} takes {len(self.ins)} arguments')
        env = self._bind(values)
        memo = {}
        outs = [evaluate(o, env, memo) for o in self.outs]
        outs = [o if isinstance(o, MX) else _dm(o) for o in outs]
        if kwargs:
            return dict(zip(self.out_names, outs))
        return outs[0] if len(outs) == 1 else outs


def integrator(name, algo, dae, t0=0.0, tf=1.0, *a, **k):
    """x(tf) of  x' = ode(x, p)  from x(t0) = x0 — CVODES in the reference, SciPy DOP853 (rtol 1e-12) here."""
    from scipy.integrate import solve_ivp
    f = Function(name + '_ode', [dae['x'], dae['p']], [dae['ode']])

    def call(x0=None, p=None, **kw):
        x0 = _num(x0).reshape(-1)
        pv = _num(p)
        sol = solve_ivp(lambda t, x: np.asarray(f(x, pv)).reshape(-1), (float(t0), float(tf)), x0, method='DOP853',
                        rtol=1e-12, atol=1e-14)
        return {'xf': _dm(sol.y[:, -1].reshape(-1, 1))}
    return call


def DM_(x=0.0):
    return _dm(_num(x))


def install(modules):
    """Register this file as the `casadi` module in a sys.modules-like dict."""
    import sys
    import types
    cs = types.ModuleType('casadi')
    me = sys.modules[__name__]
    for name in ('MX', 'SX', 'Function', 'jacobian', 'integrator', 'vertcat', 'horzcat', 'blockcat', 'skew', 'sin', 'cos',
                 'tan', 'sqrt', 'mtimes', 'transpose'):
        setattr(cs, name, getattr(me, name))
    cs.DM = DM_
    modules['casadi'] = cs
    return cs
