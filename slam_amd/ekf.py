"""Python face of the C ABI (include/slamk.h): a batch of EKF-SLAM filters on one MI355X.

Thin by design: every method is one call through libslamk.so; numpy arrays are host
buffers, integers passed as `*_dev` arguments are device addresses (e.g. torch
`tensor.data_ptr()`).  Method names follow the reference loop (src/slam/main.cpp:150-236):
frame -> (state overrides) -> predict -> update.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Match, Params


class SlamkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("slamk error %d: %s" % (code, msg))
        self.code = code


def default_params():
    p = Params()
    _lib.lib().slamk_default_params(C.byref(p))
    return p


def _f64(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None and a.size != n:
        raise ValueError("expected %d doubles, got %d" % (n, a.size))
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def matches_array(matches):
    arr = (Match * max(1, len(matches)))()
    for i, (mid, sx, sy, rx, ry) in enumerate(matches):
        arr[i].id, arr[i].sx, arr[i].sy, arr[i].rx, arr[i].ry = int(mid), sx, sy, rx, ry
    return arr


class FilterBatch:
    def __init__(self, max_features, batch=1, device=0, params=None):
        self.L = _lib.lib()
        self.B, self.F = int(batch), int(max_features)
        self.h = C.c_void_p()
        p = params if params is not None else default_params()
        self.params = p
        rc = self.L.slamk_create(C.byref(p), self.F, self.B, int(device), C.byref(self.h))
        if rc:
            msg = self.L.slamk_last_error(self.h).decode() if self.h else "create failed"
            raise SlamkError(rc, msg)

    def close(self):
        if getattr(self, "h", None):
            self.L.slamk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise SlamkError(rc, self.L.slamk_last_error(self.h).decode())

    # ---- state ---------------------------------------------------------------------------
    def set_state(self, b, X=None, V=None, bias=None):
        X = None if X is None else _f64(X, 3)
        V = None if V is None else _f64(V, 3)
        bias = None if bias is None else _f64(bias, 3)
        self._chk(self.L.slamk_set_state(self.h, b, _ptr(X), _ptr(V), _ptr(bias)))

    def get_state(self, b):
        X, V, bias = np.zeros(3), np.zeros(3), np.zeros(3)
        self._chk(self.L.slamk_get_state(self.h, b, _ptr(X), _ptr(V), _ptr(bias)))
        return X, V, bias

    def num_features(self, b=0):
        return self.L.slamk_num_features(self.h, b)

    def num_rows(self, b=0):
        return self.L.slamk_num_rows(self.h, b)

    def set_cov(self, b, P):
        P = _f64(P)
        self._chk(self.L.slamk_set_cov(self.h, b, _ptr(P), P.shape[0], P.shape[1]))

    def get_cov(self, b):
        n = 9 + 3 * self.num_features(b)
        P = np.zeros((n, n))
        self._chk(self.L.slamk_get_cov(self.h, b, _ptr(P), n, n))
        return P

    def set_features(self, b, ids, pib, anchor, q0, pib0, is_ref):
        nf = len(ids)
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        is_ref = np.ascontiguousarray(is_ref, dtype=np.uint8)
        pib, anchor, q0, pib0 = _f64(pib, 3 * nf), _f64(anchor, 3 * nf), _f64(q0, 4 * nf), _f64(pib0, 2 * nf)
        self._chk(self.L.slamk_set_features(self.h, b, nf, _ptr(ids), _ptr(pib), _ptr(anchor), _ptr(q0), _ptr(pib0),
                                            _ptr(is_ref)))

    def get_features(self, b):
        nf = self.num_features(b)
        ids = np.zeros(max(nf, 1), dtype=np.int32)
        is_ref = np.zeros(max(nf, 1), dtype=np.uint8)
        pib, anchor = np.zeros((max(nf, 1), 3)), np.zeros((max(nf, 1), 3))
        q0, pib0 = np.zeros((max(nf, 1), 4)), np.zeros((max(nf, 1), 2))
        n = C.c_int(0)
        self._chk(self.L.slamk_get_features(self.h, b, C.byref(n), _ptr(ids), _ptr(pib), _ptr(anchor), _ptr(q0),
                                            _ptr(pib0), _ptr(is_ref)))
        return {"ids": ids[:nf], "pib": pib[:nf], "anchor": anchor[:nf], "q0": q0[:nf], "pib0": pib0[:nf],
                "is_ref": is_ref[:nf]}

    def state_vector(self, b):
        X, V, bias = self.get_state(b)
        return np.concatenate([X, V, self.get_features(b)["pib"].ravel(), bias])

    # ---- loop body -----------------------------------------------------------------------
    def frame(self, b, matches, quat):
        """IMG branch: old_pos, depth clamp, association + covariance surgery, keep the observations."""
        arr = matches_array(matches)
        q = _f64(quat, 4)
        self._chk(self.L.slamk_frame(self.h, b, arr, len(matches), _ptr(q)))

    def frame_batch_async(self, matches, quat, match_off=None, stream=0):
        """Frames of all filters, stream-ordered (slamk_frame_batch_async).  match_off None: one frame for all."""
        arr = matches_array(matches)
        q = _f64(quat, 4 if match_off is None else 4 * self.B)
        off = None if match_off is None else np.ascontiguousarray(match_off, dtype=np.int32)
        self._chk(self.L.slamk_frame_batch_async(self.h, arr, len(matches), _ptr(off), _ptr(q), stream))

    def set_frame_values(self, b, src, refl):
        nf = self.num_features(b)
        src, refl = _f64(src, 2 * nf), _f64(refl, 2 * nf)
        self._chk(self.L.slamk_set_frame_values(self.h, b, _ptr(src), _ptr(refl)))

    def predict(self, qbw, w, acc, dt):
        B = self.B
        qbw, w, acc, dt = _f64(qbw, 4 * B), _f64(w, 3 * B), _f64(acc, 3 * B), _f64(dt, B)
        self._chk(self.L.slamk_predict(self.h, _ptr(qbw), _ptr(w), _ptr(acc), _ptr(dt)))

    def update(self, qbw, alt=None):
        B = self.B
        qbw = _f64(qbw, 4 * B)
        alt = None if alt is None else _f64(alt, B)
        applied = np.zeros(B, dtype=np.int32)
        self._chk(self.L.slamk_update(self.h, _ptr(qbw), _ptr(alt), _ptr(applied)))
        return applied

    # ---- device-pointer variants ------------------------------------------------------------
    def frame_begin_dev(self, stream=0):
        self._chk(self.L.slamk_frame_begin_dev(self.h, stream))

    def predict_dev(self, qbw_dev, w_dev, acc_dev, dt_dev, stream=0):
        self._chk(self.L.slamk_predict_dev(self.h, qbw_dev, w_dev, acc_dev, dt_dev, stream))

    def update_dev(self, qbw_dev, alt_dev=None, src_dev=None, refl_dev=None, applied_dev=None, stream=0):
        self._chk(self.L.slamk_update_dev(self.h, qbw_dev, alt_dev, src_dev, refl_dev, applied_dev, stream))

    def step_dev(self, qbw_dev, w_dev, acc_dev, dt_dev, alt_dev=None, src_dev=None, refl_dev=None, applied_dev=None,
                 stream=0):
        """frame_begin + predict + update of every filter in one call (device pointers)."""
        self._chk(self.L.slamk_step_dev(self.h, qbw_dev, w_dev, acc_dev, dt_dev, alt_dev, src_dev, refl_dev,
                                        applied_dev, stream))

    def set_graph_mode(self, on=True):
        self._chk(self.L.slamk_set_graph_mode(self.h, int(bool(on))))

    def set_pipeline_groups(self, groups):
        self._chk(self.L.slamk_set_pipeline_groups(self.h, int(groups)))

    def sync(self, stream=0):
        self._chk(self.L.slamk_sync(self.h, stream))

    PROFILE_STAGES = ("feature_kernels", "struct_predict", "struct_gain", "gemm_gain", "gemm_cov_update",
                      "gemm_cholesky", "potf2_inv", "gain_solver")

    def profile_enable(self, on=True, stages=None):
        """Stage timers on or off; `stages` (names from PROFILE_STAGES) times only those."""
        v = int(bool(on))
        if on and stages is not None:
            v = sum(1 << (self.PROFILE_STAGES.index(k) + 1) for k in stages)
        self._chk(self.L.slamk_profile_enable(self.h, v))

    def profile_read(self):
        """{stage: (milliseconds, scopes)} accumulated since the last read (HIP events on the launch stream)."""
        n = len(self.PROFILE_STAGES)
        ms = (C.c_double * n)()
        cnt = (C.c_int * n)()
        self._chk(self.L.slamk_profile_read(self.h, ms, cnt, n))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(self.PROFILE_STAGES)}

    # ---- introspection -----------------------------------------------------------------------
    def set_debug(self, on=True):
        """Keep the dense F, H and L of the last predict/update for debug_get / debug_get_F."""
        self._chk(self.L.slamk_set_debug(self.h, int(bool(on))))

    def debug_get(self, b):
        n, m = 9 + 3 * self.num_features(b), self.num_rows(b)
        H, Lm, K, e = np.zeros((m, n)), np.zeros((m, m)), np.zeros((n, m)), np.zeros(m)
        self._chk(self.L.slamk_debug_get(self.h, b, _ptr(H), _ptr(Lm), _ptr(K), _ptr(e)))
        return {"H": H, "L": Lm, "K": K, "e": e}

    def debug_get_F(self, b):
        n = 9 + 3 * self.num_features(b)
        Fm = np.zeros((n, n))
        self._chk(self.L.slamk_debug_get_F(self.h, b, _ptr(Fm)))
        return Fm


def flops_predict(n):
    return _lib.lib().slamk_flops_predict(int(n))


def flops_update(n, m, update_form=0):
    return _lib.lib().slamk_flops_update(int(n), int(m), int(update_form))


def flops_update_stage(n, m, update_form=0, stage=0):
    """MFMA flops of one update by stage: 0 = gain (factorisation + solves), 1 = covariance update."""
    return _lib.lib().slamk_flops_update_stage(int(n), int(m), int(update_form), int(stage))


def set_block_sizes(chol_outer_block=0, inverse_block=0):
    """Process-wide block sizes of the large-matrix path (0 = keep): see slamk_set_block_sizes."""
    rc = _lib.lib().slamk_set_block_sizes(int(chol_outer_block), int(inverse_block))
    if rc:
        raise SlamkError(rc, "set_block_sizes: negative size")


def mfma_f64_probe(iters=20000):
    t = C.c_double(0)
    rc = _lib.lib().slamk_mfma_f64_probe(int(iters), C.byref(t))
    if rc:
        raise SlamkError(rc, "mfma probe failed")
    return t.value
