"""Row-block sharded EKF predict/update for ONE large-map filter over the GPUs of a node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Rank r owns the rows
[r*bn, (r+1)*bn) of the covariance P and of every n-row product (T = P F^T, W = P H^T, K), and
the rows [r*bm, (r+1)*bm) of S.  The O(nf) filter state (X, V, features) is replicated: every rank
runs the per-feature kernels itself and rebuilds F and H locally, so the only traffic is one
all-gather per contraction:

    predict:  T_r = P_r F^T            -> all-gather T ->  P_r = F_r T + Q_r
    update :  W_r = P_r H^T            -> all-gather W ->  S_q = H_q W + R_q        -> all-gather S
              L L^T = S, Li = inv(L)   (replicated: the part that does not scale)
              K_r = (W_r Li^T) Li      -> all-gather K
              P_r <- P_r - K_r W^T - W_r K^T + (K_r S) K^T      (= the Joseph form, row block r)
              dx = K e, gate, mean update (replicated, every rank holds the full K)

which restates `calcP`, `calcK`, `updateP` and the gated mean update of the reference
(src/slam/main.cpp:526-530, 539-552, 561-569, 228-232) for a row-distributed P.

The numerical work goes through a backend object.  `HipBackend` (the product path) drives
libslamk.so through the C ABI and fails without a GPU; tests inject a CPU backend to exercise the
sharding and the collectives under `gloo`.
"""
import ctypes as C

import torch


def _pad16(x):
    return (x + 15) // 16 * 16


class HipBackend:
    """Local operations on the GPU through include/slamk.h."""

    def __init__(self, fb, device, stream=None):
        from . import _lib
        self.fb, self.L, self.device = fb, _lib.lib(), device
        self.stream = stream if stream is not None else torch.cuda.current_stream(device)

    @property
    def sptr(self):
        return self.stream.cuda_stream

    def dims(self):
        return 9 + 3 * self.fb.num_features(0), self.fb.num_rows(0)

    def empty(self, rows, cols):
        return torch.zeros((rows, cols), dtype=torch.float64, device=self.device)

    def _buf(self, which):
        ptr, r, c, ld = C.c_void_p(), C.c_int(), C.c_int(), C.c_int()
        self.fb._chk(self.L.slamk_buffer_dev(self.fb.h, which, C.byref(ptr), C.byref(r), C.byref(c), C.byref(ld)))

        class _Iface:
            pass
        o = _Iface()
        rows = max(r.value, 1)
        o.__cuda_array_interface__ = {"shape": (rows, ld.value), "typestr": "<f8", "data": (ptr.value, False),
                                      "version": 2, "strides": None}
        t = torch.as_tensor(o, device=self.device)
        return t[:, :c.value] if r.value > 1 else t[0, :c.value]

    def motion(self, qbw, w, acc, dt):
        self.fb._chk(self.L.slamk_stage_motion_dev(self.fb.h, qbw.data_ptr(), w.data_ptr(), acc.data_ptr(),
                                                   dt.data_ptr(), self.sptr))
        return self._buf(0), self._buf(4)          # F (n x n view), diag Q

    def measure(self, qbw, src=None, refl=None):
        self.fb._chk(self.L.slamk_stage_measure_dev(self.fb.h, qbw.data_ptr(), None,
                                                    None if src is None else src.data_ptr(),
                                                    None if refl is None else refl.data_ptr(), self.sptr))
        return self._buf(1), self._buf(2), self._buf(3)   # H, e, diag R

    def gemm(self, Cm, A, Bm, trans_b, alpha=1.0, beta=0.0):
        """Cm = alpha * A * op(Bm) + beta * Cm on row-major views with unit inner stride."""
        M, K = A.shape
        N = Cm.shape[1]
        assert A.stride(1) == 1 and Bm.stride(1) == 1 and Cm.stride(1) == 1
        assert M % 16 == 0 and N % 16 == 0 and K % 16 == 0, (M, N, K)
        rc = self.L.slamk_dgemm_dev(int(bool(trans_b)), M, N, K, alpha, A.data_ptr(), A.stride(0), 0, Bm.data_ptr(),
                                    Bm.stride(0), 0, beta, Cm.data_ptr(), Cm.stride(0), 0, 1, self.sptr)
        if rc:
            raise RuntimeError("slamk_dgemm_dev failed: %d" % rc)

    def cholesky_inverse(self, S, Lm, Li):
        rc = self.L.slamk_cholesky_inverse_dev(S.data_ptr(), Lm.data_ptr(), Li.data_ptr(), S.shape[0], S.stride(0),
                                               self.sptr)
        if rc:
            raise RuntimeError("slamk_cholesky_inverse_dev failed: %d" % rc)

    def apply(self, K):
        applied = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.fb._chk(self.L.slamk_stage_apply_dev(self.fb.h, K.data_ptr(), K.stride(0), applied.data_ptr(), self.sptr))
        return applied

    def get_P(self):
        return self._buf(5)

    def sync(self):
        self.stream.synchronize()


class ShardedLargeMapFilter:
    """Covariance of one filter sharded by row blocks over `world` ranks (see module docstring)."""

    def __init__(self, backend, rank=0, world=1, group=None):
        self.be, self.rank, self.world, self.group = backend, rank, world, group
        self.n = self.m = 0
        self.P = None

    # -- layout ------------------------------------------------------------------------------
    def _layout(self):
        n, m = self.be.dims()
        g = self.world
        self.n, self.m = n, m
        self.bn = _pad16((_pad16(n) + g - 1) // g)      # rows of P per rank
        self.bm = _pad16((_pad16(m) + g - 1) // g)      # rows of S per rank
        self.N, self.M = self.bn * g, self.bm * g       # padded global dimensions
        self.r0, self.q0 = self.rank * self.bn, self.rank * self.bm

    def load_cov(self, P_full):
        """Distribute a full n x n covariance (same on every rank): keep rows [r0, r0 + bn)."""
        self._layout()
        be = self.be
        self.P = be.empty(self.bn, self.N)
        hi = min(self.n, self.r0 + self.bn)
        if hi > self.r0:
            self.P[:hi - self.r0, :self.n] = P_full[self.r0:hi, :self.n]
        self.T = be.empty(self.N, self.N)
        self.W = be.empty(self.N, self.M)
        self.K = be.empty(self.N, self.M)
        self.Y = be.empty(self.bn, self.M)
        self.KS = be.empty(self.bn, self.M)
        self.S = be.empty(self.M, self.M)
        self.S2 = be.empty(self.M, self.M)
        self.Lm = be.empty(self.M, self.M)
        self.Li = be.empty(self.M, self.M)
        self.Fp = be.empty(self.N, self.N)
        self.Hp = be.empty(self.M, self.N)

    def _all_gather(self, full, block_rows):
        if self.world == 1:
            return
        import torch.distributed as dist
        mine = full[self.rank * block_rows:(self.rank + 1) * block_rows]
        dist.all_gather_into_tensor(full, mine.contiguous(), group=self.group)

    def gather_cov(self):
        """Full n x n covariance on every rank."""
        full = self.be.empty(self.N, self.N)
        full[self.r0:self.r0 + self.bn] = self.P
        self._all_gather(full, self.bn)
        return full[:self.n, :self.n]

    # -- loop body ---------------------------------------------------------------------------
    def predict(self, qbw, w, acc, dt):
        be, n = self.be, self.n
        F, qd = be.motion(qbw, w, acc, dt)
        self.Fp.zero_()
        self.Fp[:n, :n] = F[:n, :n]
        r = slice(self.r0, self.r0 + self.bn)
        be.gemm(self.T[r], self.P, self.Fp, True)                   # T_r = P_r F^T
        self._all_gather(self.T, self.bn)
        be.gemm(self.P, self.Fp[r], self.T, False)                  # P_r = F_r T
        k = min(self.bn, max(0, n - self.r0))
        if k > 0:
            idx = torch.arange(k, device=self.P.device)
            self.P[idx, self.r0 + idx] += qd[self.r0:self.r0 + k]   # + Q (diagonal, G = I)

    def update(self, qbw, src=None, refl=None):
        be, n, m = self.be, self.n, self.m
        H, e, rd = be.measure(qbw, src, refl)
        self.Hp.zero_()
        self.Hp[:m, :n] = H[:m, :n]
        r = slice(self.r0, self.r0 + self.bn)
        q = slice(self.q0, self.q0 + self.bm)
        be.gemm(self.W[r], self.P, self.Hp, True)                   # W_r = P_r H^T
        self._all_gather(self.W, self.bn)
        be.gemm(self.S[q], self.Hp[q], self.W, False)               # S_q = H_q W
        idx = torch.arange(self.bm, device=self.S.device)
        gq = self.q0 + idx
        rfull = torch.ones(self.bm, dtype=torch.float64, device=self.S.device)   # unit diagonal on padding rows
        live = gq < m
        rfull[live] = rd[gq[live]]
        self.S[gq, gq] += rfull                                     # + R
        self._all_gather(self.S, self.bm)
        self.S2.copy_(self.S)
        self.Lm.zero_()
        self.Li.zero_()
        be.cholesky_inverse(self.S2, self.Lm, self.Li)              # replicated
        be.gemm(self.Y, self.W[r], self.Li, True)                   # Y_r = W_r Li^T
        be.gemm(self.K[r], self.Y, self.Li, False)                  # K_r = Y_r Li
        self._all_gather(self.K, self.bn)
        # Joseph row block: P_r - K_r W^T - W_r K^T + (K_r S) K^T
        be.gemm(self.P, self.K[r], self.W, True, alpha=-1.0, beta=1.0)
        be.gemm(self.P, self.W[r], self.K, True, alpha=-1.0, beta=1.0)
        be.gemm(self.KS, self.K[r], self.S, False)
        be.gemm(self.P, self.KS, self.K, True, alpha=1.0, beta=1.0)
        return be.apply(self.K)                                     # dx = K e, gate, mean update (replicated)
