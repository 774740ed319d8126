"""ctypes bindings of oracle/_build/libekf_oracle.so (CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package (slam_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libekf_oracle.so")
SLAM_BIN = os.path.join(_HERE, "_build", "slam_oracle")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


class Params(C.Structure):
    _fields_ = [("Q0", C.c_double), ("R0", C.c_double), ("qbias", C.c_double), ("p0", C.c_double),
                ("pinit", C.c_double), ("alt_override", C.c_double), ("use_alt_override", C.c_int),
                ("gate", C.c_double), ("d_min", C.c_double), ("d_max", C.c_double), ("dinit", C.c_double)]


class Feature(C.Structure):
    _fields_ = [("id", C.c_int), ("pib", C.c_double * 3), ("anchor", C.c_double * 3),
                ("q0", C.c_double * 4), ("pib0", C.c_double * 2), ("isRef", C.c_int), ("noMatch", C.c_int)]


class Match(C.Structure):
    _fields_ = [("id", C.c_int), ("sx", C.c_double), ("sy", C.c_double), ("rx", C.c_double), ("ry", C.c_double)]


class FilterS(C.Structure):
    _fields_ = [("X", C.c_double * 3), ("V", C.c_double * 3), ("b", C.c_double * 3),
                ("nf", C.c_int), ("cap", C.c_int), ("feat", C.POINTER(Feature)),
                ("rf", C.c_int), ("nrf", C.c_int), ("P", C.POINTER(C.c_double)), ("n", C.c_int),
                ("prm", Params)]


class Sensors(C.Structure):
    _fields_ = [("quat", C.c_double * 4), ("acc", C.c_double * 3), ("ang", C.c_double * 3),
                ("alt", C.c_double), ("init", C.c_double * 3), ("vel", C.c_double * 3), ("pos", C.c_double * 3)]


ACC, ANG, QUAT, IMG, ALT, INIT, POS, VEL = (1 << i for i in range(8))

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        L.orc_filter_new.restype = C.POINTER(FilterS)
        L.orc_filter_new.argtypes = [C.POINTER(Params)]
        L.orc_filter_free.argtypes = [C.POINTER(FilterS)]
        L.orc_update_features.argtypes = [C.POINTER(FilterS), C.POINTER(Match), C.c_int, dp]
        L.orc_step.restype = C.c_int
        L.orc_step.argtypes = [C.POINTER(FilterS), C.c_int, C.POINTER(Sensors), C.c_double, dp,
                               C.POINTER(Match), C.c_int, dp, C.POINTER(C.c_int)]
        L.orc_rotation.argtypes = [dp, dp]
        L.orc_jacobian_h.argtypes = [dp, dp, C.POINTER(Feature), dp, dp]
        L.orc_image2body.argtypes = [C.c_double, C.c_double, dp, dp]
        L.orc_euler2quaternion.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.orc_fir.argtypes = [dp, C.c_int, dp, dp, C.c_int]
        L.orc_motion_jacobian.argtypes = [C.POINTER(FilterS), C.POINTER(Sensors), C.c_double, dp]
        L.orc_dynamics_step.argtypes = [C.POINTER(FilterS), C.POINTER(Sensors), C.c_double]
        L.orc_measurement_model.restype = C.c_int
        L.orc_measurement_model.argtypes = [C.POINTER(FilterS), dp, C.c_double, C.POINTER(Match), C.c_int,
                                            dp, dp, dp, dp, C.POINTER(C.c_int)]
        L.orc_init_q_diag.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, dp]
        L.orc_init_r_diag.restype = C.c_int
        L.orc_init_r_diag.argtypes = [C.c_double, C.POINTER(C.c_int), C.c_int, dp]
        L.orc_calc_p.argtypes = [dp, dp, dp, C.c_int]
        L.orc_calc_k.restype = C.c_int
        L.orc_calc_k.argtypes = [dp, dp, dp, dp, C.c_int, C.c_int]
        L.orc_update_p.argtypes = [dp, dp, dp, C.c_int, C.c_int]
        L.orc_pib0_hat.argtypes = [C.POINTER(Feature), dp, dp, dp]
        L.orc_ppb_hat.argtypes = [C.POINTER(Feature), dp, dp, dp]
        L.orc_default_params.argtypes = [C.POINTER(Params)]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vec(v, n):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float64).reshape(n))
    return a


def default_params():
    p = Params()
    lib().orc_default_params(C.byref(p))
    return p


def rotation(q):
    q = _vec(q, 4)
    R = np.zeros(9)
    lib().orc_rotation(_dp(q), _dp(R))
    return R.reshape(3, 3)


def euler2quaternion(phi, theta, psi):
    q = np.zeros(4)
    lib().orc_euler2quaternion(phi, theta, psi, _dp(q))
    return q


def fir(coeffs, x):
    """src/lpf/fir.cpp on one channel."""
    c = np.ascontiguousarray(coeffs, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.zeros_like(x)
    lib().orc_fir(_dp(c), len(c), _dp(x), _dp(y), len(x))
    return y


def image2body(px, py):
    a, b = C.c_double(), C.c_double()
    lib().orc_image2body(px, py, C.byref(a), C.byref(b))
    return a.value, b.value


def make_feature(pib, anchor, q0, pib0=(0, 0), is_ref=0, fid=0):
    f = Feature()
    f.id = fid
    f.pib[:] = list(pib)
    f.anchor[:] = list(anchor)
    f.q0[:] = list(q0)
    f.pib0[:] = list(pib0)
    f.isRef = int(is_ref)
    return f


def jacobian_h(X, qbw, pib, anchor, q0):
    f = make_feature(pib, anchor, q0)
    X = _vec(X, 3)
    qbw = _vec(qbw, 4)
    Hb, Hi = np.zeros(24), np.zeros(12)
    lib().orc_jacobian_h(_dp(X), _dp(qbw), C.byref(f), _dp(Hb), _dp(Hi))
    return Hb.reshape(4, 6), Hi.reshape(4, 3)


def matches_array(matches):
    """matches: iterable of (id, sx, sy, rx, ry)."""
    arr = (Match * max(1, len(matches)))()
    for i, (mid, sx, sy, rx, ry) in enumerate(matches):
        arr[i].id, arr[i].sx, arr[i].sy, arr[i].rx, arr[i].ry = int(mid), sx, sy, rx, ry
    return arr


class Filter:
    """One EKF filter: state + covariance, stepped exactly like the reference loop."""

    def __init__(self, params=None):
        self.L = lib()
        self.p = self.L.orc_filter_new(C.byref(params) if params is not None else None)
        self.old_pos = np.zeros(3)

    def __del__(self):
        if getattr(self, "p", None):
            self.L.orc_filter_free(self.p)
            self.p = None

    # --- accessors -------------------------------------------------------
    @property
    def s(self):
        return self.p.contents

    @property
    def nf(self):
        return self.s.nf

    @property
    def n(self):
        return self.s.n

    @property
    def X(self):
        return np.array(self.s.X[:])

    @X.setter
    def X(self, v):
        self.s.X[:] = list(v)

    @property
    def V(self):
        return np.array(self.s.V[:])

    @V.setter
    def V(self, v):
        self.s.V[:] = list(v)

    @property
    def b(self):
        return np.array(self.s.b[:])

    @property
    def P(self):
        n = self.s.n
        return np.ctypeslib.as_array(self.s.P, shape=(n, n)).copy()

    def set_P(self, P):
        n = self.s.n
        P = np.ascontiguousarray(P, dtype=np.float64)
        assert P.shape == (n, n)
        np.ctypeslib.as_array(self.s.P, shape=(n, n))[:] = P

    def features(self):
        return [self.s.feat[i] for i in range(self.s.nf)]

    def ids(self):
        return [self.s.feat[i].id for i in range(self.s.nf)]

    def pib(self):
        return np.array([self.s.feat[i].pib[:] for i in range(self.s.nf)]).reshape(-1, 3)

    def state_vector(self):
        return np.concatenate([self.X, self.V, self.pib().ravel(), self.b])

    # --- reference operations -------------------------------------------
    def update_features(self, matches, quat):
        arr = matches_array(matches)
        q = _vec(quat, 4)
        self.L.orc_update_features(self.p, arr, len(matches), _dp(q))

    def step(self, u, sensors, dt, matches=()):
        """One loop iteration (main.cpp:157-235).  Returns (rc, (x1, x0), applied)."""
        arr = matches_array(matches)
        xy = np.zeros(2)
        applied = C.c_int(-1)
        rc = self.L.orc_step(self.p, int(u), C.byref(sensors), float(dt), _dp(self.old_pos),
                             arr, len(matches), _dp(xy), C.byref(applied))
        return rc, (xy[0], xy[1]), applied.value

    def motion_jacobian(self, sensors, dt):
        n = self.s.n
        F = np.zeros((n, n))
        self.L.orc_motion_jacobian(self.p, C.byref(sensors), float(dt), _dp(F))
        return F

    def dynamics_step(self, sensors, dt):
        self.L.orc_dynamics_step(self.p, C.byref(sensors), float(dt))

    def measurement_model(self, old_pos, alt, matches, qbw):
        nf, n = self.s.nf, self.s.n
        mcap = 1 + 6 * nf
        meas, hmu = np.zeros(mcap), np.zeros(mcap)
        H = np.zeros((mcap, n))
        ref = (C.c_int * (nf + 1))()
        arr = matches_array(matches)
        op, q = _vec(old_pos, 3), _vec(qbw, 4)
        m = self.L.orc_measurement_model(self.p, _dp(op), float(alt), arr, len(matches), _dp(q),
                                         _dp(meas), _dp(hmu), _dp(H), ref)
        if m < 0:
            raise RuntimeError("Cannot find match.")
        return meas[:m].copy(), hmu[:m].copy(), H[:m].copy(), [ref[i] for i in range(nf)]


def make_sensors(quat=(0, 0, 0, 1), acc=(0, 0, 0), ang=(0, 0, 0), alt=0.0, vel=(0, 0, 0), pos=(0, 0, 0),
                 init=(0, 0, 0)):
    s = Sensors()
    s.quat[:] = list(quat)
    s.acc[:] = list(acc)
    s.ang[:] = list(ang)
    s.alt = alt
    s.vel[:] = list(vel)
    s.pos[:] = list(pos)
    s.init[:] = list(init)
    return s


def init_q_diag(nf, dt, Q0=256.0, qbias=1e-5):
    qd = np.zeros(9 + 3 * nf)
    lib().orc_init_q_diag(nf, Q0, qbias, dt, _dp(qd))
    return qd


def init_r_diag(ref_flags, R0=32.0):
    nfeat = len(ref_flags)
    rd = np.zeros(1 + 6 * nfeat)
    fl = (C.c_int * max(1, nfeat))(*[int(x) for x in ref_flags])
    m = lib().orc_init_r_diag(R0, fl, nfeat, _dp(rd))
    return rd[:m].copy()


def calc_p(P, F, qd):
    P = np.ascontiguousarray(P, dtype=np.float64).copy()
    F = np.ascontiguousarray(F, dtype=np.float64)
    qd = np.ascontiguousarray(qd, dtype=np.float64)
    lib().orc_calc_p(_dp(P), _dp(F), _dp(qd), P.shape[0])
    return P


def calc_k(H, P, rd):
    H = np.ascontiguousarray(H, dtype=np.float64)
    P = np.ascontiguousarray(P, dtype=np.float64)
    rd = np.ascontiguousarray(rd, dtype=np.float64)
    m, n = H.shape
    K = np.zeros((n, m))
    lib().orc_calc_k(_dp(K), _dp(H), _dp(P), _dp(rd), m, n)
    return K


def update_p(P, K, H):
    P = np.ascontiguousarray(P, dtype=np.float64).copy()
    K = np.ascontiguousarray(K, dtype=np.float64)
    H = np.ascontiguousarray(H, dtype=np.float64)
    m, n = H.shape
    lib().orc_update_p(_dp(P), _dp(K), _dp(H), m, n)
    return P


def slam_run(sensors_text, features_path, dump_path=None):
    """Run the oracle's `slam` front end: returns (exit code, stdout, stderr)."""
    if not os.path.exists(SLAM_BIN):
        build()
    cmd = [SLAM_BIN, features_path]
    if dump_path:
        cmd += ["--dump", dump_path]
    r = subprocess.run(cmd, input=sensors_text, capture_output=True, text=True)
    return r.returncode, r.stdout, r.stderr


def parse_dump(path):
    """Parse a --dump file into a list of dicts (iter, X, V, ids, pib, P)."""
    out = []
    with open(path) as fp:
        lines = fp.read().split("\n")
    i = 0
    fh = float.fromhex
    while i < len(lines):
        if not lines[i].startswith("STEP"):
            i += 1
            continue
        _, it, nf = lines[i].split()
        it, nf = int(it), int(nf)
        X = np.array([fh(t) for t in lines[i + 1].split()])
        V = np.array([fh(t) for t in lines[i + 2].split()])
        ids, pib = [], []
        for k in range(nf):
            t = lines[i + 3 + k].split()
            ids.append(int(t[0]))
            pib.append([fh(x) for x in t[1:]])
        n = 9 + 3 * nf
        P = np.array([[fh(t) for t in lines[i + 3 + nf + r].split()] for r in range(n)])
        out.append({"iter": it, "X": X, "V": V, "ids": ids, "pib": np.array(pib).reshape(-1, 3), "P": P})
        i += 3 + nf + n
    return out
