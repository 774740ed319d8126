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
caller's stream), which fails without a GPU.  (The CPU rehearsal of the same sequence over `gloo` is test
infrastructure and lives in tests/sharded_rehearsal.py.)
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
