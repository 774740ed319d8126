"""Row-block sharded EKF predict/update for ONE large-map filter over the GPUs of a node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Rank r owns the rows
[r*bn, (r+1)*bn) of the covariance P and of every n-row product (T = P F^T, W = P H^T, K, A), and
the rows [r*bm, (r+1)*bm) of S.  The O(nf) filter state (X, V, features) is replicated: every rank
runs the per-feature kernels itself and holds the compact rows of F and H (at most six non-zeros
per row), so the products with the Jacobians are local, bandwidth-bound sweeps over the rank's row
block and the only traffic is one all-gather per dense contraction:

    predict:  T_r = P_r F^T                    -> all-gather T ->  P_r = F_r T + Q_r
    update :  W_r = P_r H^T                    -> all-gather W ->  S_q = H_q W + R_q   -> all-gather S
              L L^T = S, Li = inverses of L's diagonal blocks   (replicated: the part that does not scale)
              K_r = W_r L^-T L^-1              -> all-gather K
              A_r = I_r - K_r H                -> all-gather A, all-gather P
              P_r <- (A_r P) A^T + (K_r R) K^T (Joseph form, row block r; three dense GEMMs)
              dx = K e, gate, mean update      (replicated, every rank holds the full K)

which restates `calcP`, `calcK`, `updateP` (Joseph form) and the gated mean update of the reference
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
    """Local operations on the GPU through include/slamk.h (filter 0 of a FilterBatch)."""

    F, H = 0, 1

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
        """Compact rows of F and diag Q at the pre-predict state, then mu += dynamics(mu) dt.  Returns diag Q."""
        self.fb._chk(self.L.slamk_stage_motion_dev(self.fb.h, qbw.data_ptr(), w.data_ptr(), acc.data_ptr(),
                                                   dt.data_ptr(), self.sptr))
        return self._buf(4)

    def measure(self, qbw, src=None, refl=None):
        """Compact rows of H, innovation e and diag R."""
        self.fb._chk(self.L.slamk_stage_measure_dev(self.fb.h, qbw.data_ptr(), None,
                                                    None if src is None else src.data_ptr(),
                                                    None if refl is None else refl.data_ptr(), self.sptr))
        return self._buf(2), self._buf(3)

    def right(self, which, Out, In):
        """Out = In * M^T for a row block In (M = F or H)."""
        assert In.stride(1) == 1 and Out.stride(1) == 1 and In.shape[0] == Out.shape[0]
        self.fb._chk(self.L.slamk_stage_right_dev(self.fb.h, which, In.data_ptr(), In.stride(0), In.shape[0],
                                                  Out.data_ptr(), Out.stride(0), self.sptr))

    def left(self, which, Out, row0, In, add_diag):
        """Out = M[row0 : row0 + rows, :] * In (+ diag Q / R on the operator's diagonal)."""
        assert In.stride(1) == 1 and Out.stride(1) == 1 and In.shape[1] == Out.shape[1]
        self.fb._chk(self.L.slamk_stage_left_dev(self.fb.h, which, row0, Out.shape[0], In.data_ptr(), In.stride(0),
                                                 In.shape[1], Out.data_ptr(), Out.stride(0), int(bool(add_diag)),
                                                 self.sptr))

    def eye_minus_kh(self, A, K, row0):
        """A = I[row0 : row0 + rows, :] - K H for the rank's rows of K."""
        assert A.stride(1) == 1 and K.stride(1) == 1 and A.shape[0] == K.shape[0]
        self.fb._chk(self.L.slamk_stage_eye_minus_kh_dev(self.fb.h, K.data_ptr(), K.stride(0), row0, K.shape[0],
                                                         A.data_ptr(), A.stride(0), self.sptr))

    def gain(self, Y, K, KR, W, Lm, Li, rd):
        """K = W S^-1 (two block triangular solves with L and its diagonal-block inverses Li), KR = K diag(rd), for
        the rank's rows of W; W is overwritten, Y is workspace."""
        m = Li.shape[0]
        assert W.stride(0) == Y.stride(0) == K.stride(0) == KR.stride(0) and rd.shape[0] >= m
        self.fb._chk(self.L.slamk_stage_gain_dev(self.fb.h, W.data_ptr(), W.shape[0], m, W.stride(0), Lm.data_ptr(), Li.data_ptr(),
                                                 Li.stride(0), Y.data_ptr(), K.data_ptr(), KR.data_ptr(),
                                                 rd.data_ptr(), self.sptr))

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
                                               None, self.sptr)
        if rc:
            raise RuntimeError("slamk_cholesky_inverse_dev failed: %d" % rc)

    def apply(self, K):
        applied = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.fb._chk(self.L.slamk_stage_apply_dev(self.fb.h, K.data_ptr(), K.stride(0), applied.data_ptr(), self.sptr))
        return applied

    def sync(self):
        self.stream.synchronize()


class ShardedLargeMapFilter:
    """Covariance of one filter sharded by row blocks over `world` ranks (see module docstring)."""

    F, H = 0, 1

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
        self.G1 = be.empty(self.N, self.N)          # gathered T, then gathered P
        self.G2 = be.empty(self.N, self.N)          # gathered A
        self.Ar = be.empty(self.bn, self.N)         # A_r, then (A_r P)
        self.AP = be.empty(self.bn, self.N)
        self.W = be.empty(self.N, self.M)
        self.K = be.empty(self.N, self.M)
        self.Y = be.empty(self.bn, self.M)
        self.KR = be.empty(self.bn, self.M)
        self.S = be.empty(self.M, self.M)
        self.S2 = be.empty(self.M, self.M)
        self.Lm = be.empty(self.M, self.M)
        self.Li = be.empty(self.M, self.M)
        self.rfull = be.empty(1, self.M)[0]

    def _all_gather(self, full, block_rows, mine=None):
        """Rows [rank * block_rows, ...) of `full` from every rank (`mine`: this rank's block if it lives elsewhere)."""
        own = full[self.rank * block_rows:(self.rank + 1) * block_rows]
        if mine is not None:
            own.copy_(mine)
        if self.world == 1:
            return
        import torch.distributed as dist
        if full.is_cuda and dist.get_backend(self.group) != "nccl":
            # rehearsal on a backend without device collectives (gloo): stage through host memory
            host = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(host, own.cpu().contiguous(), group=self.group)
            full.copy_(host)
            return
        dist.all_gather_into_tensor(full, own.contiguous(), group=self.group)

    def gather_cov(self):
        """Full n x n covariance on every rank."""
        full = self.be.empty(self.N, self.N)
        self._all_gather(full, self.bn, self.P)
        return full[:self.n, :self.n]

    # -- loop body ---------------------------------------------------------------------------
    def predict(self, qbw, w, acc, dt):
        be = self.be
        be.motion(qbw, w, acc, dt)
        r = slice(self.r0, self.r0 + self.bn)
        be.right(self.F, self.G1[r], self.P)                        # T_r = P_r F^T
        self._all_gather(self.G1, self.bn)
        be.left(self.F, self.P, self.r0, self.G1, True)             # P_r = F_r T + Q_r

    def update(self, qbw, src=None, refl=None):
        be, m = self.be, self.m
        e, rd = be.measure(qbw, src, refl)
        r = slice(self.r0, self.r0 + self.bn)
        q = slice(self.q0, self.q0 + self.bm)
        be.right(self.H, self.W[r], self.P)                         # W_r = P_r H^T
        self._all_gather(self.W, self.bn)
        be.left(self.H, self.S[q], self.q0, self.W, True)           # S_q = H_q W + R_q
        ml = min(_pad16(m), rd.shape[0])
        self.rfull.fill_(1.0)                                       # unit diagonal on the rows the operator does not have
        self.rfull[:ml] = rd[:ml]
        lo, hi = max(self.q0, ml), self.q0 + self.bm
        if lo < hi:
            pad = torch.arange(lo, hi, device=self.S.device)
            self.S[lo:hi] = 0.0
            self.S[pad, pad] = 1.0
        self._all_gather(self.S, self.bm)
        self.S2.copy_(self.S)
        self.Lm.zero_()
        self.Li.zero_()
        be.cholesky_inverse(self.S2, self.Lm, self.Li)              # replicated
        be.gain(self.Y, self.K[r], self.KR, self.W[r], self.Lm, self.Li, self.rfull)   # K_r = W_r S^-1, K_r R
        self._all_gather(self.K, self.bn)
        be.eye_minus_kh(self.Ar, self.K[r], self.r0)                # A_r = I_r - K_r H
        self._all_gather(self.G2, self.bn, self.Ar)
        self._all_gather(self.G1, self.bn, self.P)
        be.gemm(self.AP, self.Ar, self.G1, False)                   # A_r P
        be.gemm(self.P, self.AP, self.G2, True)                     # (A_r P) A^T
        be.gemm(self.P, self.KR, self.K, True, alpha=1.0, beta=1.0)  # + (K_r R) K^T
        return be.apply(self.K)                                     # dx = K e, gate, mean update (replicated)
