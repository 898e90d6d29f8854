"""Analytic stand-in for the reference's CasADi `SymbolicModel`
(/root/reference/safe_control_gym/math_and_models/symbolic_systems.py:6-121) for the two robots.

The reference builds its prior model symbolically with CasADi (cartpole.py:390-437, quadrotor.py:468-604); CasADi is not
available in this image, and the model is host-side, single-env bookkeeping for controllers such as LQR — not part of the
data-parallel hot path.  This class evaluates the same continuous-time equations with NumPy and exposes the members
controllers read: ``nx nu ny dt X_EQ U_EQ fc_func df_func fd_func loss`` (Jacobians by central differences of the
analytic right-hand side), plus printable ``x_sym u_sym y_sym x_dot cost_func``.
"""
import numpy as np


class _Dense:
    """Mimics a CasADi DM for the `.toarray()` call sites (lqr_utils.py:22-23)."""

    def __init__(self, a):
        self._a = np.asarray(a, dtype=float)

    def toarray(self):
        return self._a

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    def __float__(self):
        return float(self._a.reshape(-1)[0])

    @property
    def shape(self):
        return self._a.shape


class _Names:
    """Display-only stand-in for a CasADi symbol vector: `str()` reads like CasADi's (`vertcat(x, x_dot, ...)`), `.shape` is (n, 1).
    Not an expression graph — controllers that BUILD on the symbols (MPC, CBF, MPSC: out of scope) need CasADi."""

    def __init__(self, names):
        self.names = tuple(names)
        self.shape = (len(self.names), 1)

    def __str__(self):
        return 'vertcat(' + ', '.join(self.names) + ')'

    __repr__ = __str__

    def __len__(self):
        return len(self.names)

    def __iter__(self):
        return iter(self.names)


# x_dot of `AnalyticModel.f`, entry by entry, in the notation of the reference's expressions (cartpole.py:409-414, quadrotor.py:490, 506-509, 540-562)
_X_DOT = {
    ('cartpole', 4): ('x_dot', 'tmp - m*l*theta_dd*cos(theta)/(m+M)   [tmp = (F + m*l*theta_dot^2*sin(theta))/(m+M)]', 'theta_dot',
                      'theta_dd = (g*sin(theta) - cos(theta)*tmp) / (l*(4/3 - m*cos(theta)^2/(m+M)))'),
    ('quadrotor', 2): ('z_dot', 'T/m - g'),
    ('quadrotor', 6): ('x_dot', 'sin(theta)*(T1+T2)/m', 'z_dot', 'cos(theta)*(T1+T2)/m - g', 'theta_dot', 'l*(T2-T1)/Iyy/sqrt(2)'),
    ('quadrotor', 12): ('x_dot', '(Rob @ [0,0,f1+f2+f3+f4])[0]/m', 'y_dot', '(Rob @ [0,0,f1+f2+f3+f4])[1]/m', 'z_dot',
                        '(Rob @ [0,0,f1+f2+f3+f4])[2]/m - g', '(W(phi,theta) @ [p,q,r])[0]', '(W(phi,theta) @ [p,q,r])[1]', '(W(phi,theta) @ [p,q,r])[2]',
                        '(J^-1 (Mb - [p,q,r] x J [p,q,r]))[0]   [Mb = (l/sqrt2*(f1+f2-f3-f4), l/sqrt2*(-f1+f2+f3-f4), gamma*(-f1+f2-f3+f4))]',
                        '(J^-1 (Mb - [p,q,r] x J [p,q,r]))[1]', '(J^-1 (Mb - [p,q,r] x J [p,q,r]))[2]'),
}
_U_NAMES = {('cartpole', 1): ('F',), ('quadrotor', 1): ('T',), ('quadrotor', 2): ('T1', 'T2'), ('quadrotor', 4): ('f1', 'f2', 'f3', 'f4')}


class AnalyticModel:
    def __init__(self, name, spec, prior_prop=None):
        self.name, self.spec = name, spec
        prior_prop = prior_prop or {}
        self.dt = spec.CTRL_TIMESTEP
        self.nx, self.nu = spec.nx, spec.nu
        self.ny = self.nx
        self.Q, self.R = spec.Q, spec.R
        # what a caller may PRINT of the symbolic model (examples/no_controller/verbose_api.py:52-58)
        self.x_sym = self.y_sym = _Names(spec.state_labels)
        self.u_sym = _Names(_U_NAMES[(name, self.nu)])
        self.x_dot = _Names(_X_DOT[(name, self.nx)])
        self.cost_func = '0.5*(X-Xr).T @ Q @ (X-Xr) + 0.5*(U-Ur).T @ R @ (U-Ur)'
        g = spec.GRAVITY_ACC
        if name == 'cartpole':
            self.params = dict(length=prior_prop.get('pole_length', spec.EFFECTIVE_POLE_LENGTH),
                               m=prior_prop.get('pole_mass', spec.POLE_MASS), M=prior_prop.get('cart_mass', spec.CART_MASS), g=g)
            self.X_EQ = np.zeros(4)
            self.U_EQ = np.atleast_2d(spec.U_GOAL)[0, :]
            self.pole_length, self.pole_mass, self.cart_mass = self.params['length'], self.params['m'], self.params['M']
        else:
            m = prior_prop.get('M', spec.MASS)
            self.params = dict(m=m, Ixx=prior_prop.get('Ixx', spec.J[0, 0]), Iyy=prior_prop.get('Iyy', spec.J[1, 1]),
                               Izz=prior_prop.get('Izz', spec.J[2, 2]), g=g, L=spec.L, gamma=spec.KM / spec.KF)
            self.X_EQ = np.zeros(self.nx)
            self.U_EQ = np.ones(self.nu) * m * g / self.nu
            self.quad_mass, self.quad_Iyy = m, self.params['Iyy']

    # ---- continuous-time dynamics x_dot = f(x, u)
    def f(self, x, u):
        x = np.asarray(x, dtype=float).reshape(-1)
        u = np.asarray(u, dtype=float).reshape(-1)
        p = self.params
        if self.name == 'cartpole':             # cartpole.py:412-414
            _, x_dot, th, th_dot = x
            Mm, ml = p['m'] + p['M'], p['m'] * p['length']
            tmp = (u[0] + ml * th_dot ** 2 * np.sin(th)) / Mm
            th_dd = (p['g'] * np.sin(th) - np.cos(th) * tmp) / (p['length'] * (4.0 / 3.0 - p['m'] * np.cos(th) ** 2 / Mm))
            return np.array([x_dot, tmp - ml * th_dd * np.cos(th) / Mm, th_dot, th_dd])
        if self.nx == 2:                        # quadrotor.py:490
            return np.array([x[1], u[0] / p['m'] - p['g']])
        if self.nx == 6:                        # quadrotor.py:506-509
            _, xd, _, zd, th, thd = x
            T = u[0] + u[1]
            return np.array([xd, np.sin(th) * T / p['m'], zd, np.cos(th) * T / p['m'] - p['g'], thd,
                             p['L'] * (u[1] - u[0]) / p['Iyy'] / np.sqrt(2)])
        # quadrotor.py:552-562
        _, xd, _, yd, _, zd, phi, th, psi, pb, qb, rb = x
        cphi, sphi, cth, sth, cpsi, spsi = np.cos(phi), np.sin(phi), np.cos(th), np.sin(th), np.cos(psi), np.sin(psi)
        Rz = np.array([[cpsi, -spsi, 0], [spsi, cpsi, 0], [0, 0, 1]])
        Ry = np.array([[cth, 0, sth], [0, 1, 0], [-sth, 0, cth]])
        Rx = np.array([[1, 0, 0], [0, cphi, -sphi], [0, sphi, cphi]])
        Rob = Rz @ Ry @ Rx
        acc = Rob @ np.array([0, 0, u.sum()]) / p['m'] - np.array([0, 0, p['g']])
        lsq = p['L'] / np.sqrt(2.0)
        Mb = np.array([lsq * (u[0] + u[1] - u[2] - u[3]), lsq * (-u[0] + u[1] + u[2] - u[3]),
                       p['gamma'] * (-u[0] + u[1] - u[2] + u[3])])
        J = np.diag([p['Ixx'], p['Iyy'], p['Izz']])
        w = np.array([pb, qb, rb])
        rate_dot = np.linalg.solve(J, Mb - np.cross(w, J @ w))
        ang_dot = np.array([[1, sphi * np.tan(th), cphi * np.tan(th)], [0, cphi, -sphi],
                            [0, sphi / cth, cphi / cth]]) @ w
        return np.array([xd, acc[0], yd, acc[1], zd, acc[2], ang_dot[0], ang_dot[1], ang_dot[2],
                         rate_dot[0], rate_dot[1], rate_dot[2]])

    def fc_func(self, x, u):
        return {'f': self.f(x, u).reshape(-1, 1)}

    def df_func(self, x, u, eps=1e-6):
        """(dfdx, dfdu) at (x, u) — symbolic_systems.py:79-85."""
        x = np.asarray(x, dtype=float).reshape(-1)
        u = np.asarray(u, dtype=float).reshape(-1)
        A = np.zeros((self.nx, self.nx))
        B = np.zeros((self.nx, self.nu))
        for k in range(self.nx):
            d = np.zeros(self.nx); d[k] = eps
            A[:, k] = (self.f(x + d, u) - self.f(x - d, u)) / (2 * eps)
        for k in range(self.nu):
            d = np.zeros(self.nu); d[k] = eps
            B[:, k] = (self.f(x, u + d) - self.f(x, u - d)) / (2 * eps)
        return _Dense(A), _Dense(B)

    def fd_func(self, x0, p, substeps=20):
        """Discrete-time prior: RK4 over one control period (the reference integrates with cvodes, :70-73)."""
        x = np.asarray(x0, dtype=float).reshape(-1)
        h = self.dt / substeps
        for _ in range(substeps):
            k1 = self.f(x, p); k2 = self.f(x + 0.5 * h * k1, p); k3 = self.f(x + 0.5 * h * k2, p); k4 = self.f(x + h * k3, p)
            x = x + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
        return {'xf': x.reshape(-1, 1)}

    def loss(self, x, u, Xr, Ur, Q, R):
        """The reference's `loss` Function (symbolic_systems.py:106-121) for its quadratic cost
        l = 0.5 (x-Xr)' Q (x-Xr) + 0.5 (u-Ur)' R (u-Ur)  (quadrotor.py:578, cartpole.py:422): value, gradients and Hessians
        with CasADi's shapes (l_x 1 x nx, l_xx nx x nx, l_u 1 x nu, l_uu nu x nu, l_xu nx x nu)."""
        ex = np.asarray(x, dtype=float).reshape(-1) - np.asarray(Xr, dtype=float).reshape(-1)
        eu = np.asarray(u, dtype=float).reshape(-1) - np.asarray(Ur, dtype=float).reshape(-1)
        Q, R = np.asarray(Q, dtype=float), np.asarray(R, dtype=float)
        Qs, Rs = 0.5 * (Q + Q.T), 0.5 * (R + R.T)
        # DM-like results: the reference's callers take `.toarray()` of every output (ilqr.py:210-247), `l` as a 1 x 1 matrix
        out = {'l': np.array([[0.5 * ex @ Q @ ex + 0.5 * eu @ R @ eu]]), 'l_x': (Qs @ ex).reshape(1, -1), 'l_xx': Qs,
               'l_u': (Rs @ eu).reshape(1, -1), 'l_uu': Rs, 'l_xu': np.zeros((ex.size, eu.size))}
        return {k: _Dense(v) for k, v in out.items()}
