"""CPU backend of tests/sharded_rehearsal.ShardedLargeMapFilter (TEST INFRASTRUCTURE): the local operations of the
row-block sharded filter done with the oracle and torch CPU tensors, so that the sharding logic and
the collectives can run under gloo without a GPU."""
import numpy as np
import torch

from oracle import oracle as orc


class OracleBackend:
    def __init__(self, flt, alt=-1.34):
        self.f = flt              # oracle.Filter holding the replicated state
        self.alt = alt
        self.matches = []
        self.old_pos = np.zeros(3)

    def dims(self):
        nf = self.f.nf
        return 9 + 3 * nf, 1 + 6 * self.f.s.rf + 4 * self.f.s.nrf

    def empty(self, rows, cols):
        return torch.zeros((rows, cols), dtype=torch.float64)

    def frame(self, matches, quat):
        self.old_pos = self.f.X.copy()
        self.f.update_features(matches, quat)
        self.matches = matches

    def motion(self, qbw, w, acc, dt):
        s = orc.make_sensors(quat=qbw, acc=acc, ang=w)
        self.Fm = torch.from_numpy(self.f.motion_jacobian(s, float(dt)))
        self.f.dynamics_step(s, float(dt))
        self.qd = torch.from_numpy(orc.init_q_diag(self.f.nf, float(dt)))
        return self.qd

    def measure(self, qbw, src=None, refl=None):
        meas, hmu, H, ref = self.f.measurement_model(self.old_pos, self.alt, self.matches, qbw)
        self.e = meas - hmu
        self.Hm = torch.from_numpy(H)
        self.rd = torch.from_numpy(orc.init_r_diag(ref))
        return torch.from_numpy(self.e), self.rd

    def _op(self, which):
        return (self.Fm, self.qd) if which == 0 else (self.Hm, self.rd)

    def right(self, which, Out, In):
        M, _ = self._op(which)
        Out[:, :M.shape[0]] = In[:, :M.shape[1]] @ M.T

    def left(self, which, Out, row0, In, add_diag, col0=0):
        """Out = M[row0:row0 + rows] In (+ the diagonal term where its column row - col0 lies inside Out); returns the
        number of rows the operator has in this range."""
        M, d = self._op(which)
        rows = max(0, min(Out.shape[0], M.shape[0] - row0))
        if rows == 0:
            return 0
        Out[:rows] = M[row0:row0 + rows] @ In[:M.shape[1]]
        if add_diag:
            for i in range(rows):
                c = row0 + i - col0
                if 0 <= c < Out.shape[1]:
                    Out[i, c] += d[row0 + i]
        return rows

    def eye_minus_kh(self, A, K, row0):
        n, m = self.dims()
        A[:, :n] = -(K[:, :m] @ self.Hm)
        for j in range(A.shape[0]):
            if row0 + j < n:
                A[j, row0 + j] += 1.0

    def gain(self, Y, K, KR, W, Lm, Li, rd):
        Y.copy_(W @ Li.T)
        K.copy_(Y @ Li)
        KR.copy_(K * rd[None, :Li.shape[0]])

    def gemm(self, Cm, A, Bm, trans_b, alpha=1.0, beta=0.0):
        prod = A @ (Bm.T if trans_b else Bm)
        if beta == 0.0:
            Cm.copy_(alpha * prod)
        else:
            Cm.mul_(beta).add_(alpha * prod)

    def chol_panel(self, S, Lm, J0, J1):
        """Columns [J0, J1) of the factor from the same columns of S, which carry the updates of the columns before."""
        L11 = torch.linalg.cholesky(S[J0:J1, J0:J1])
        Lm[J0:J1, J0:J1] = L11
        if J1 < S.shape[0]:
            Lm[J1:, J0:J1] = torch.linalg.solve_triangular(L11, S[J1:, J0:J1].T, upper=False).T

    def block_inverses(self, Lm, Li):
        Li.copy_(torch.linalg.solve_triangular(Lm, torch.eye(Lm.shape[0], dtype=torch.float64), upper=False))

    def apply(self, K):
        n, m = self.dims()
        dx = (K[:n, :m] @ torch.from_numpy(self.e)).numpy()
        f = self.f
        ok = np.abs(f.V - dx[3:6]).sum() < f.s.prm.gate
        if ok:
            f.X = f.X + dx[0:3]
            f.V = f.V + dx[3:6]
            for i in range(f.nf):
                for c in range(3):
                    f.s.feat[i].pib[c] += dx[6 + 3 * i + c]
        return torch.tensor([int(ok)], dtype=torch.int32)

    def sync(self):
        pass
