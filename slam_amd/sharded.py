"""Row-block sharded EKF predict/update for ONE large-map filter over the GPUs of a node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Rank r owns the rows
[r*bn, (r+1)*bn) of the covariance P and of every n-row product (T = P F^T, W = P H^T, K, A), and
the rows [r*bm, (r+1)*bm) of S.  The O(nf) filter state (X, V, features) is replicated: every rank
runs the per-feature kernels itself and holds the compact rows of F and H (at most six non-zeros
per row), so the products with the Jacobians are local, bandwidth-bound sweeps over the rank's row
block and the only traffic is one all-gather per dense contraction:

    predict:  T_r = P_r F^T  -> all-gather of the 13 rows of T per rank that other ranks read ->  P_r = F_r T + Q_r
    update :  W_r = P_r H^T  -> all-gather W
              S = H W + R = L L^T, left-looking by panels of columns: each rank forms its share of the rows of the
              panel from W and the rows of H and subtracts L[rows, :J0] L[J0:J1, :J0]^T -> all-gather of the panel ->
              panel factorisation on every rank; Li = inverses of L's diagonal blocks
              K_r = W_r L^-T L^-1      -> all-gather K;   A = I - K H on every rank;   all-gather P
              P_r <- (A_r P) A^T + (K_r R) K^T (Joseph form, row block r; three dense GEMMs)
              dx = K e, gate, mean update      (replicated, every rank holds the full K)

which restates `calcP`, `calcK`, `updateP` (Joseph form) and the gated mean update of the reference
(src/slam/main.cpp:526-530, 539-552, 561-569, 228-232) for a row-distributed P.

`ShardedFilter` is the product path: the sequence above inside libslamk.so (slamk_sharded_*, RCCL all-gathers on the
caller's stream), which fails without a GPU.  `ShardedLargeMapFilter` is the same sequence written out over a backend
object and `torch.distributed`: tests inject a CPU backend to rehearse the sharding and the collectives with several
ranks under `gloo`, where no GPU exists.
"""
import ctypes as C

import torch


def _pad16(x):
    return (x + 15) // 16 * 16


class ShardedFilter:
    """The product path: row-block sharded filter behind the C ABI (slamk_sharded_* in include/slamk.h).

    The orchestration and the collectives (RCCL all-gathers on the caller's stream) live in libslamk.so; this class
    creates the communicator (the id is made on rank 0 and broadcast through torch.distributed) and forwards
    device pointers.  With a process group that has no device collectives (gloo: rehearsals with several ranks on
    one GPU) the library is given an all-gather callback that stages through host memory instead.
    `fb` is a FilterBatch with batch 1 that every rank feeds with the same frames (the O(nf) state is replicated).
    """

    def __init__(self, fb, rank=0, world=1, group=None, device=None, stream=None, rccl_single_rank=False):
        from . import _lib
        self.fb, self.L, self.rank, self.world, self.group = fb, _lib.lib(), rank, world, group
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        self.s = C.c_void_p()
        self._cb = None
        nccl_id, cb = None, None
        if world == 1 and rccl_single_rank:      # a one-rank RCCL communicator: the collective calls are really made
            nccl_id = (C.c_ubyte * 128)()
            if self.L.slamk_nccl_unique_id(nccl_id):
                raise RuntimeError("slamk_nccl_unique_id failed")
        if world > 1:
            import torch.distributed as dist
            if dist.get_backend(group) == "nccl":
                idbuf = torch.zeros(128, dtype=torch.uint8)
                if rank == 0:
                    raw = (C.c_ubyte * 128)()
                    if self.L.slamk_nccl_unique_id(raw):
                        raise RuntimeError("slamk_nccl_unique_id failed")
                    idbuf = torch.tensor(list(raw), dtype=torch.uint8)
                idbuf = idbuf.to(self.device)
                dist.broadcast(idbuf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                nccl_id = (C.c_ubyte * 128)(*idbuf.cpu().tolist())
            else:
                AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)

                def gather(ctx, buf, bytes_per_rank, rk, wd, strm):
                    try:
                        class _I:
                            pass
                        o = _I()
                        o.__cuda_array_interface__ = {"shape": (wd * bytes_per_rank,), "typestr": "|u1", "data": (buf, False),
                                                      "version": 2, "strides": None}
                        full = torch.as_tensor(o, device=self.device)
                        self.stream.synchronize()
                        mine = full[rk * bytes_per_rank:(rk + 1) * bytes_per_rank].cpu()
                        host = torch.empty(wd * bytes_per_rank, dtype=torch.uint8)
                        dist.all_gather_into_tensor(host, mine, group=self.group)
                        with torch.cuda.stream(self.stream):
                            full.copy_(host)
                        self.stream.synchronize()
                        return 0
                    except Exception:            # never let an exception cross the C boundary
                        import traceback
                        traceback.print_exc()
                        return 1
                cb = AG(gather)
                self._cb = cb
        rc = self.L.slamk_sharded_create(self.fb.h, rank, world, nccl_id, C.cast(cb, C.c_void_p) if cb else None, None,
                                         C.byref(self.s))
        self.fb._chk(rc)

    @property
    def sptr(self):
        return self.stream.cuda_stream

    def close(self):
        if getattr(self, "s", None):
            self.L.slamk_sharded_destroy(self.s)
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dims(self):
        v = [C.c_int() for _ in range(6)]
        self.fb._chk(self.L.slamk_sharded_dims(self.s, *[C.byref(x) for x in v]))
        return dict(zip(("n", "m", "rows_per_rank", "srows_per_rank", "n_padded", "m_padded"), (x.value for x in v)))

    def load_cov(self, P_full):
        """Distribute a full n x n covariance (device tensor, the same on every rank)."""
        assert P_full.is_cuda and P_full.stride(1) == 1
        self.fb._chk(self.L.slamk_sharded_load_cov_dev(self.s, P_full.data_ptr(), P_full.stride(0), self.sptr))

    def predict(self, qbw, w, acc, dt):
        self.fb._chk(self.L.slamk_sharded_predict_dev(self.s, qbw.data_ptr(), w.data_ptr(), acc.data_ptr(), dt.data_ptr(),
                                                      self.sptr))

    def update(self, qbw, src=None, refl=None):
        applied = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.fb._chk(self.L.slamk_sharded_update_dev(self.s, qbw.data_ptr(), None,
                                                     None if src is None else src.data_ptr(),
                                                     None if refl is None else refl.data_ptr(), applied.data_ptr(), self.sptr))
        return applied

    def gather_cov(self):
        n = 9 + 3 * self.fb.num_features(0)
        out = torch.zeros((n, n), dtype=torch.float64, device=self.device)
        self.fb._chk(self.L.slamk_sharded_gather_cov_dev(self.s, out.data_ptr(), n, self.sptr))
        return out

    def sync(self):
        self.fb.sync(self.sptr)


class ShardedLargeMapFilter:
    """The sequence of the module docstring over a backend object (CPU rehearsal of the sharding; the product path is
    ShardedFilter).  `panel`: columns per panel of the factorisation (the library's outer block size)."""

    F, H = 0, 1
    HALO = 13

    def __init__(self, backend, rank=0, world=1, group=None, panel=512):
        self.be, self.rank, self.world, self.group = backend, rank, world, group
        self.panel = panel
        self.n = self.m = 0
        self.P = None

    # -- layout ------------------------------------------------------------------------------
    def _layout(self):
        n, m = self.be.dims()
        g = self.world
        self.n, self.m = n, m
        self.bn = _pad16((_pad16(n) + g - 1) // g)      # rows of P per rank
        self.bm = _pad16((_pad16(m) + g - 1) // g)
        self.N, self.M = self.bn * g, self.bm * g       # padded global dimensions
        self.r0 = self.rank * self.bn

    def load_cov(self, P_full):
        """Distribute a full n x n covariance (same on every rank): keep rows [r0, r0 + bn)."""
        self._layout()
        be = self.be
        self.P = be.empty(self.bn, self.N)
        hi = min(self.n, self.r0 + self.bn)
        if hi > self.r0:
            self.P[:hi - self.r0, :self.n] = P_full[self.r0:hi, :self.n]
        self.G1 = be.empty(self.N, self.N)          # T (own rows and the rows received), then gathered P
        self.G2 = be.empty(self.N, self.N)          # A
        self.AP = be.empty(self.bn, self.N)
        self.W = be.empty(self.N, self.M)
        self.K = be.empty(self.N, self.M)
        self.Y = be.empty(self.bn, self.M)
        self.KR = be.empty(self.bn, self.M)
        self.S = be.empty(self.M, self.M)           # the panels as they are factored (never S as a whole)
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
    def _halo_rows(self, p):
        """Global rows of T in the 13 slots of rank p's pack (-1: not owned by p): the first and the last two rows of
        its block (a feature's triple may lie across a block boundary), rows 0..5 (position, velocity) and the bias
        triple: everything a row of F outside the block can refer to."""
        bn, bias0 = self.bn, self.n - 3
        cand = [p * bn, p * bn + 1, (p + 1) * bn - 2, (p + 1) * bn - 1, 0, 1, 2, 3, 4, 5, bias0, bias0 + 1, bias0 + 2]
        return [r if p * bn <= r < (p + 1) * bn else -1 for r in cand]

    def predict(self, qbw, w, acc, dt):
        be = self.be
        be.motion(qbw, w, acc, dt)
        r = slice(self.r0, self.r0 + self.bn)
        be.right(self.F, self.G1[r], self.P)                        # T_r = P_r F^T
        if self.world > 1:
            pack = be.empty(self.world * self.HALO, self.N)
            for t, row in enumerate(self._halo_rows(self.rank)):
                if row >= 0:
                    pack[self.rank * self.HALO + t] = self.G1[row]
            self._all_gather(pack, self.HALO)
            for p in range(self.world):
                if p != self.rank:
                    for t, row in enumerate(self._halo_rows(p)):
                        if row >= 0:
                            self.G1[row] = pack[p * self.HALO + t]
        be.left(self.F, self.P, self.r0, self.G1, True)             # P_r = F_r T + Q_r

    def _cholesky(self):
        """S = H W + R = L L^T, every rank ending with L and the inverses of its diagonal blocks."""
        be, g, M = self.be, self.world, self.M
        for J0 in range(0, M, self.panel):
            J1 = min(M, J0 + self.panel)
            ob, R = J1 - J0, M - J0
            chunk = _pad16((R + g - 1) // g)            # rows of the panel per rank
            U = be.empty(g * chunk, ob)
            row0 = J0 + self.rank * chunk
            rows = max(0, min(M, row0 + chunk) - row0)
            if rows > 0:
                Ur = U[self.rank * chunk:self.rank * chunk + rows]
                rop = be.left(self.H, Ur, row0, self.W[:, J0:J1], True, col0=J0)    # rows of H W + R
                for q in range(row0 + rop, row0 + rows):                           # rows the operator does not have
                    if J0 <= q < J1:
                        Ur[q - row0, q - J0] = 1.0
                if J0 > 0:
                    be.gemm(Ur, self.Lm[row0:row0 + rows, :J0], self.Lm[J0:J1, :J0], True, alpha=-1.0, beta=1.0)
            self._all_gather(U, chunk)
            self.S[J0:M, J0:J1] = U[:R]
            be.chol_panel(self.S, self.Lm, J0, J1)
        be.block_inverses(self.Lm, self.Li)

    def update(self, qbw, src=None, refl=None):
        be, m = self.be, self.m
        e, rd = be.measure(qbw, src, refl)
        r = slice(self.r0, self.r0 + self.bn)
        be.right(self.H, self.W[r], self.P)                         # W_r = P_r H^T
        self._all_gather(self.W, self.bn)
        ml = min(_pad16(m), rd.shape[0])
        self.rfull.fill_(1.0)                                       # unit diagonal on the rows the operator does not have
        self.rfull[:ml] = rd[:ml]
        self.Lm.zero_()
        self.Li.zero_()
        self._cholesky()
        be.gain(self.Y, self.K[r], self.KR, self.W[r], self.Lm, self.Li, self.rfull)   # K_r = W_r S^-1, K_r R
        self._all_gather(self.K, self.bn)
        be.eye_minus_kh(self.G2, self.K, 0)                         # A = I - K H, all rows, on every rank
        self._all_gather(self.G1, self.bn, self.P)
        be.gemm(self.AP, self.G2[r], self.G1, False)                # A_r P
        be.gemm(self.P, self.AP, self.G2, True)                     # (A_r P) A^T
        be.gemm(self.P, self.KR, self.K, True, alpha=1.0, beta=1.0)  # + (K_r R) K^T
        return be.apply(self.K)                                     # dx = K e, gate, mean update (replicated)
