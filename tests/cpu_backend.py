"""CPU stand-in for slam_amd.sharded.HipBackend (TEST INFRASTRUCTURE): the local operations of the
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
        F = self.f.motion_jacobian(s, float(dt))
        self.f.dynamics_step(s, float(dt))
        qd = orc.init_q_diag(self.f.nf, float(dt))
        return torch.from_numpy(F), torch.from_numpy(qd)

    def measure(self, qbw, src=None, refl=None):
        meas, hmu, H, ref = self.f.measurement_model(self.old_pos, self.alt, self.matches, qbw)
        self.e = meas - hmu
        rd = orc.init_r_diag(ref)
        return torch.from_numpy(H), torch.from_numpy(self.e), torch.from_numpy(rd)

    def gemm(self, Cm, A, Bm, trans_b, alpha=1.0, beta=0.0):
        prod = A @ (Bm.T if trans_b else Bm)
        if beta == 0.0:
            Cm.copy_(alpha * prod)
        else:
            Cm.mul_(beta).add_(alpha * prod)

    def cholesky_inverse(self, S, Lm, Li):
        L = torch.linalg.cholesky(S)
        Lm.copy_(L)
        Li.copy_(torch.linalg.solve_triangular(L, torch.eye(S.shape[0], dtype=torch.float64), upper=False))

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
