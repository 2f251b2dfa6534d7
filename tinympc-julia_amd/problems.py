"""Problem families used by the benchmark configs and the parity tests.

Numeric problem data only (A, B, Q, R, bounds, initial-state distributions) as
given by the reference's example scripts; BASELINE.json `configs` / SURVEY.md
§8(d) define the batched workloads built from them.

  cartpole   reference: examples/cartpole_example_one_solve.jl:11-17
  quadrotor  reference: examples/quadrotor_hover_codegen.jl:26-58
  rocket     reference: examples/rocket_landing_constraints.jl:17-57
"""
from dataclasses import dataclass, field

import numpy as np


@dataclass
class Problem:
    name: str
    A: np.ndarray
    B: np.ndarray
    Q: np.ndarray
    R: np.ndarray
    rho: float
    N: int
    fdyn: np.ndarray = None
    x_min: np.ndarray = None  # (nx, N) or None
    x_max: np.ndarray = None
    u_min: np.ndarray = None  # (nu, N-1) or None
    u_max: np.ndarray = None
    extra: dict = field(default_factory=dict)

    @property
    def nx(self):
        return self.A.shape[0]

    @property
    def nu(self):
        return self.B.shape[1]

    def has_bounds(self):
        return self.u_min is not None


def _box(nx, nu, N, ulo, uhi, xlo=-1e17, xhi=1e17):
    return (np.full((nx, N), xlo), np.full((nx, N), xhi),
            np.full((nu, N - 1), ulo), np.full((nu, N - 1), uhi))


def cartpole(N=20, rho=1.0, u_bound=None):
    A = np.array([[1.0, 0.01, 0.0, 0.0],
                  [0.0, 1.0, 0.039, 0.0],
                  [0.0, 0.0, 1.002, 0.01],
                  [0.0, 0.0, 0.458, 1.002]])
    B = np.array([[0.0], [0.02], [0.0], [0.067]])
    Q = np.diag([10.0, 1.0, 10.0, 1.0])
    R = np.diag([1.0])
    p = Problem("cartpole", A, B, Q, R, rho, N, fdyn=np.zeros(4))
    if u_bound is not None:
        p.x_min, p.x_max, p.u_min, p.u_max = _box(4, 1, N, -u_bound, u_bound)
    return p


def quadrotor(N=30, rho=5.0, u_bound=0.5):
    A = np.array([
        [1.0, 0.0, 0.0, 0.0, 0.024525, 0.0, 0.05, 0.0, 0.0, 0.0, 0.0002044, 0.0],
        [0.0, 1.0, 0.0, -0.024525, 0.0, 0.0, 0.0, 0.05, 0.0, -0.0002044, 0.0, 0.0],
        [0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.05, 0.0, 0.0, 0.0],
        [0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.025, 0.0, 0.0],
        [0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.025, 0.0],
        [0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.025],
        [0.0, 0.0, 0.0, 0.0, 0.981, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0122625, 0.0],
        [0.0, 0.0, 0.0, -0.981, 0.0, 0.0, 0.0, 1.0, 0.0, -0.0122625, 0.0, 0.0],
        [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0],
        [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0],
        [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0],
        [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]])
    B = np.array([
        [-0.0007069, 0.0007773, 0.0007091, -0.0007795],
        [0.0007034, 0.0007747, -0.0007042, -0.0007739],
        [0.0052554, 0.0052554, 0.0052554, 0.0052554],
        [-0.1720966, -0.1895213, 0.1722891, 0.1893288],
        [-0.1729419, 0.190174, 0.1734809, -0.1907131],
        [0.0123423, -0.0045148, -0.0174024, 0.0095748],
        [-0.056552, 0.0621869, 0.0567283, -0.0623632],
        [0.0562756, 0.0619735, -0.0563386, -0.0619105],
        [0.2102143, 0.2102143, 0.2102143, 0.2102143],
        [-13.7677303, -15.1617018, 13.7831318, 15.1463003],
        [-13.8353509, 15.2139209, 13.8784751, -15.2570451],
        [0.9873856, -0.361182, -1.392188, 0.7659845]])
    Q = np.diag([100.0, 100.0, 100.0, 4.0, 4.0, 400.0, 4.0, 4.0, 4.0, 2.0408163, 2.0408163, 4.0])
    R = np.diag([4.0, 4.0, 4.0, 4.0])
    p = Problem("quadrotor", A, B, Q, R, rho, N, fdyn=np.zeros(12))
    if u_bound is not None:
        p.x_min, p.x_max, p.u_min, p.u_max = _box(12, 4, N, -u_bound, u_bound)
    return p


def rocket(N=50, rho=1.0, box=True):
    A = np.eye(6)
    A[0, 3] = A[1, 4] = A[2, 5] = 0.05
    B = np.zeros((6, 3))
    B[0, 0] = B[1, 1] = B[2, 2] = 0.000125
    B[3, 0] = B[4, 1] = B[5, 2] = 0.005
    Q = np.diag([101.0] * 6)
    R = np.diag([2.0] * 3)
    fdyn = np.array([0.0, 0.0, -0.0122625, 0.0, 0.0, -0.4905])
    p = Problem("rocket", A, B, Q, R, rho, N, fdyn=fdyn)
    if box:
        x_min = np.full((6, N), -1e17)
        x_max = np.full((6, N), 1e17)
        lo = [-5.0, -5.0, -0.5, -10.0, -10.0, -20.0]
        hi = [5.0, 5.0, 100.0, 10.0, 10.0, 20.0]
        for i in range(6):
            x_min[i, :] = lo[i]
            x_max[i, :] = hi[i]
        p.x_min, p.x_max = x_min, x_max
        p.u_min = np.full((3, N - 1), -10.0)
        p.u_max = np.full((3, N - 1), 105.0)
    p.extra = dict(xinit=np.array([4.0, 2.0, 20.0, -3.0, 2.0, -4.5]), xgoal=np.zeros(6),
                   cone_mu_u=0.25, cone_mu_x=0.5, ntotal=100)
    return p


# ---- batched synthetic inputs, SURVEY.md §8(d) ----

def cartpole_x0(batch, seed=0):
    """x0[b] ~ U([-0.5,0.5] x [-0.2,0.2] x [-0.1,0.1] x [-0.2,0.2]); returns (nx, batch) fp64."""
    rng = np.random.default_rng(seed)
    half = np.array([0.5, 0.2, 0.1, 0.2])
    return np.asfortranarray(((rng.random((batch, 4)) * 2.0 - 1.0) * half).T)


def quadrotor_x0(batch, seed=1):
    """x0[b] ~ U(-0.3, 0.3)^12; returns (nx, batch) fp64."""
    rng = np.random.default_rng(seed)
    return np.asfortranarray(((rng.random((batch, 12)) * 2.0 - 1.0) * 0.3).T)


def rocket_x0(batch, seed=2):
    """x0[b] = 1.1 * xinit * (1 + 0.05 U(-1,1)); returns (nx, batch) fp64."""
    rng = np.random.default_rng(seed)
    xinit = np.array([4.0, 2.0, 20.0, -3.0, 2.0, -4.5])
    return np.asfortranarray((1.1 * xinit * (1.0 + 0.05 * (rng.random((batch, 6)) * 2.0 - 1.0))).T)


def rocket_refs(N, ntotal=100):
    """Xref = linear interpolation xinit -> xgoal (rocket_landing_constraints.jl:83-85), Uref[2]=10."""
    xinit = np.array([4.0, 2.0, 20.0, -3.0, 2.0, -4.5])
    xgoal = np.zeros(6)
    xref = np.zeros((6, N), order="F")
    for i in range(N):
        xref[:, i] = xinit + (xgoal - xinit) * i / (ntotal - 1)
    uref = np.zeros((3, N - 1), order="F")
    uref[2, :] = 10.0
    return xref, uref
