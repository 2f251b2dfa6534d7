"""TEST INFRASTRUCTURE ONLY — ctypes access to the CPU checkers.

  CpuSolver("orc64")  our fp64 C restatement      (oracle/libtinympc_oracle.so)
  CpuSolver("orc32")  same loop in fp32 + fp64 cache
  CpuSolver("ref")    the reference's own vendored TinyMPC snapshot compiled from
                      /root/reference (oracle/_ref/libtinympc_ref.so), when built

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PORT_LIB = os.path.join(_HERE, "libtinympc_oracle.so")
REF_LIB = os.path.join(_HERE, "_ref", "libtinympc_ref.so")
REF_ADAPT_LIB = os.path.join(_HERE, "_ref", "libtinympc_ref_adapt.so")

_c_dp = ctypes.POINTER(ctypes.c_double)
_c_ip = ctypes.POINTER(ctypes.c_int)


def build(port=True, ref=True):
    """Compile the checkers (gcc/g++ only).  `_ref` is skipped where /root/reference is absent."""
    if port:
        subprocess.check_call(["make", "-s", "-C", _HERE, "port"])
    if ref and os.path.isdir("/root/reference/src/codegen_src/tinympc"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "_ref"])


def have_ref():
    return os.path.isfile(REF_LIB)


def _dp(a):
    return None if a is None else a.ctypes.data_as(_c_dp)


def _f(a):
    return None if a is None else np.asfortranarray(np.asarray(a, dtype=np.float64))


_libs = {}


def _load(kind):
    path = REF_LIB if kind == "ref" else (REF_ADAPT_LIB if kind == "refa" else PORT_LIB)
    if path not in _libs:
        if not os.path.isfile(path):
            raise FileNotFoundError(f"{path} not built (run oracle.cpu_oracle.build())")
        _libs[path] = ctypes.CDLL(path)
    return _libs[path]


class CpuSolver:
    """One CPU solver instance (single problem), same call sequence as the reference's API."""

    def __init__(self, kind, A, B, Q, R, rho, N):
        assert kind in ("orc64", "orc32", "ref", "refa")
        self.lib = _load(kind)
        # "refa": the same snapshot and driver built with automatic variables zero-initialised, the only way the
        # snapshot's adaptive-rho branch is defined behaviour (see oracle/Makefile)
        kind = "ref" if kind == "refa" else kind
        self.kind = kind
        self.p = kind + "_"
        self.nx, self.nu, self.N = A.shape[0], B.shape[1], N
        fn = getattr(self.lib, self.p + "create")
        fn.restype = ctypes.c_void_p
        A, B, Q, R = _f(A), _f(B), _f(Q), _f(R)
        self.h = ctypes.c_void_p(fn(_dp(A), _dp(B), _dp(Q), _dp(R), ctypes.c_double(rho),
                                    self.nx, self.nu, N))
        if not self.h:
            raise RuntimeError("create failed")

    def _call(self, name, *args):
        fn = getattr(self.lib, self.p + name)
        return fn(self.h, *args)

    def close(self):
        if self.h:
            self._call("destroy")
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update_settings(self, abs_pri_tol=1e-3, abs_dua_tol=1e-3, max_iter=100,
                        check_termination=1, en_state_bound=0, en_input_bound=0):
        self._call("update_settings", ctypes.c_double(abs_pri_tol), ctypes.c_double(abs_dua_tol),
                   int(max_iter), int(check_termination), int(en_state_bound), int(en_input_bound))

    def set_bound_constraints(self, x_min, x_max, u_min, u_max):
        a = [_f(m) for m in (x_min, x_max, u_min, u_max)]
        self._call("set_bound_constraints", *[_dp(m) for m in a])

    def set_x0(self, x0):
        x0 = _f(x0)
        self._call("set_x0", _dp(x0))

    def set_x_ref(self, xr):
        xr = _f(xr)
        self._call("set_x_ref", _dp(xr))

    def set_u_ref(self, ur):
        ur = _f(ur)
        self._call("set_u_ref", _dp(ur))

    def set_fdyn(self, f):
        """UNPINNED extension (not available on the compiled reference snapshot)"""
        assert self.kind != "ref", "the vendored snapshot has no affine term"
        f = np.ascontiguousarray(np.asarray(f, dtype=np.float64))
        self._call("set_fdyn", _dp(f))

    def set_cone_constraints(self, Acu, qcu, cu, Acx, qcx, cx):
        """UNPINNED extension; inputs first (bindings.cpp:453-459)"""
        assert self.kind != "ref", "the vendored snapshot has no cone constraints"
        ia = [np.ascontiguousarray(np.asarray(a, dtype=np.int32)) for a in (Acu, qcu, Acx, qcx)]
        da = [np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in (cu, cx)]
        self._call("set_cone_constraints", ia[0].ctypes.data_as(_c_ip), ia[1].ctypes.data_as(_c_ip), _dp(da[0]),
                   len(da[0]), ia[2].ctypes.data_as(_c_ip), ia[3].ctypes.data_as(_c_ip), _dp(da[1]), len(da[1]))

    def set_linear_constraints(self, Alin_x, blin_x, Alin_u, blin_u):
        """UNPINNED extension; Alin_x x <= blin_x, Alin_u u <= blin_u at every knot (bindings.cpp:413-450)"""
        assert self.kind != "ref", "the vendored snapshot has no linear constraints"
        Ax = np.asfortranarray(np.asarray(Alin_x, dtype=np.float64).reshape(-1, self.nx))
        Au = np.asfortranarray(np.asarray(Alin_u, dtype=np.float64).reshape(-1, self.nu))
        bx = np.ascontiguousarray(np.asarray(blin_x, dtype=np.float64))
        bu = np.ascontiguousarray(np.asarray(blin_u, dtype=np.float64))
        assert Ax.shape[0] == len(bx) and Au.shape[0] == len(bu)
        self._call("set_linear_constraints", _dp(Ax), Ax.shape[0], _dp(bx), _dp(Au), Au.shape[0], _dp(bu))

    def set_cache_terms(self, Kinf, Pinf, Quu_inv, AmBKt):
        a = [_f(m) for m in (Kinf, Pinf, Quu_inv, AmBKt)]
        self._call("set_cache_terms", *[_dp(m) for m in a])

    def set_adaptive_rho(self, enable, rho_min=0.1, rho_max=10.0, clip=True):
        """admm.cpp:147-174; the defaults are TinyMPC.jl:59-61's"""
        self._call("set_adaptive_rho", int(bool(enable)), ctypes.c_double(rho_min), ctypes.c_double(rho_max),
                   int(bool(clip)))

    def set_sensitivity(self, dK, dP):
        a = [_f(m) for m in (dK, dP)]
        self._call("set_sensitivity", _dp(a[0]), _dp(a[1]))

    def get_builtin_sensitivity(self):
        """compiled reference only: the tables tiny_setup hard-codes (tiny_api.cpp:279-329), 12x4 problems only"""
        assert self.kind == "ref"
        dK = np.zeros((self.nu, self.nx), order="F")
        dP = np.zeros((self.nx, self.nx), order="F")
        if self._call("get_sensitivity", _dp(dK), _dp(dP)) != 0:
            raise ValueError("the built-in tables are 4x12 / 12x12")
        return dK, dP

    def get_adapted(self):
        rho = ctypes.c_double()
        K = np.zeros((self.nu, self.nx), order="F")
        P = np.zeros((self.nx, self.nx), order="F")
        self._call("get_adapted", ctypes.byref(rho), _dp(K), _dp(P))
        return dict(rho=rho.value, Kinf=K, Pinf=P)

    def reset(self):
        self._call("reset")

    def set_forced_exit(self, k):
        """test knob (restatement only): 0 off; k > 0: leave as converged exactly at iteration k; -1: never converge"""
        assert self.kind != "ref", "the compiled reference decides by its own residuals"
        self._call("set_forced_exit", int(k))

    def solve(self):
        return int(self._call("solve"))

    def get_solution(self):
        nx, nu, N = self.nx, self.nu, self.N
        x = np.zeros((nx, N), order="F")
        u = np.zeros((nu, N - 1), order="F")
        it, so = ctypes.c_int(), ctypes.c_int()
        res = np.zeros(4)
        self._call("get_solution", _dp(x), _dp(u), ctypes.byref(it), ctypes.byref(so), _dp(res))
        return dict(x=x, u=u, iter=it.value, solved=so.value, res=res)

    def get_cache(self):
        nx, nu = self.nx, self.nu
        K = np.zeros((nu, nx), order="F")
        P = np.zeros((nx, nx), order="F")
        Qi = np.zeros((nu, nu), order="F")
        Am = np.zeros((nx, nx), order="F")
        self._call("get_cache", _dp(K), _dp(P), _dp(Qi), _dp(Am))
        return dict(Kinf=K, Pinf=P, Quu_inv=Qi, AmBKt=Am)

    def set_state(self, d, y, g, v, z):
        a = [_f(m) for m in (d, y, g, v, z)]
        self._call("set_state", *[_dp(m) for m in a])

    def get_cone_state(self):
        """the cone sets' part of the workspace (restatement only): duals gc, yc and previous slack vc, zc"""
        nx, nu, N = self.nx, self.nu, self.N
        gc, vc = np.zeros((nx, N), order="F"), np.zeros((nx, N), order="F")
        yc, zc = np.zeros((nu, N - 1), order="F"), np.zeros((nu, N - 1), order="F")
        self._call("get_cone_state", _dp(gc), _dp(vc), _dp(yc), _dp(zc))
        return dict(gc=gc, vc=vc, yc=yc, zc=zc)

    def set_cone_state(self, gc, vc, yc, zc):
        a = [_f(m) for m in (gc, vc, yc, zc)]
        self._call("set_cone_state", *[_dp(m) for m in a])

    def get_state(self):
        nx, nu, N = self.nx, self.nu, self.N
        d = np.zeros((nu, N - 1), order="F")
        y = np.zeros((nu, N - 1), order="F")
        z = np.zeros((nu, N - 1), order="F")
        g = np.zeros((nx, N), order="F")
        v = np.zeros((nx, N), order="F")
        self._call("get_state", _dp(d), _dp(y), _dp(g), _dp(v), _dp(z))
        return dict(d=d, y=y, g=g, v=v, z=z)


def project_soc(block, mu, kind="orc64"):
    """The oracle's cone projection of one block (head..., axis) — for property tests."""
    lib = _load(kind)
    b = np.ascontiguousarray(np.asarray(block, dtype=np.float64)).copy()
    getattr(lib, kind + "_project_soc_block")(_dp(b), len(b), ctypes.c_double(mu))
    return b


def project_halfspaces(z, A, b, kind="orc64"):
    """The oracle's sequential half-space projection of one vector — for property tests."""
    lib = _load(kind)
    z = np.ascontiguousarray(np.asarray(z, dtype=np.float64)).copy()
    A = np.ascontiguousarray(np.asarray(A, dtype=np.float64).reshape(-1, len(z)))
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float64))
    getattr(lib, kind + "_project_halfspaces_block")(_dp(z), len(z), _dp(A), _dp(b), len(b))
    return z


def solve_batch(kind, prob, x0, xref=None, uref=None, abs_pri_tol=1e-3, abs_dua_tol=1e-3,
                max_iter=100, check_termination=1, nthreads=1, want_outputs=True):
    """Cold-start batch on the CPU checker.  x0: (nx, B).  xref/uref: None (zeros), shared
    (nx,N)/(nu,N-1), or per-instance (nx,N,B)/(nu,N-1,B).  Returns dict incl. wall seconds."""
    lib = _load(kind)
    fn = getattr(lib, kind + "_solve_batch")
    fn.restype = ctypes.c_double
    nx, nu, N = prob.nx, prob.nu, prob.N
    x0 = _f(x0)
    B = x0.shape[1]
    per_inst = int((xref is not None and np.ndim(xref) == 3) or (uref is not None and np.ndim(uref) == 3))
    if per_inst:
        xref = np.zeros((nx, N, B), order="F") if xref is None else xref
        uref = np.zeros((nu, N - 1, B), order="F") if uref is None else uref
        if np.ndim(xref) == 2:
            xref = np.repeat(np.asarray(xref)[:, :, None], B, axis=2)
        if np.ndim(uref) == 2:
            uref = np.repeat(np.asarray(uref)[:, :, None], B, axis=2)
    xref, uref = _f(xref), _f(uref)
    use_bounds = int(prob.has_bounds())
    bnd = [_f(m) for m in (prob.x_min, prob.x_max, prob.u_min, prob.u_max)]
    if want_outputs:
        xo = np.zeros((nx, N, B), order="F")
        uo = np.zeros((nu, N - 1, B), order="F")
        it = np.zeros(B, dtype=np.int32)
        so = np.zeros(B, dtype=np.int32)
        res = np.zeros((B, 4))
    else:
        xo = uo = it = so = res = None
    A, Bm, Q, R = _f(prob.A), _f(prob.B), _f(prob.Q), _f(prob.R)
    secs = fn(_dp(A), _dp(Bm), _dp(Q), _dp(R), ctypes.c_double(prob.rho), nx, nu, N,
              *[_dp(m) for m in bnd], use_bounds, ctypes.c_double(abs_pri_tol),
              ctypes.c_double(abs_dua_tol), int(max_iter), int(check_termination), int(B),
              _dp(x0), _dp(xref), _dp(uref), per_inst, _dp(xo), _dp(uo),
              None if it is None else it.ctypes.data_as(_c_ip),
              None if so is None else so.ctypes.data_as(_c_ip), _dp(res), int(nthreads))
    if secs < 0:
        raise RuntimeError("solve_batch failed")
    return dict(x=xo, u=uo, iter=it, solved=so, res=res, seconds=secs)
