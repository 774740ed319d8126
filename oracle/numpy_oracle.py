"""Second form of the CPU oracle for filters too large for the plain C loops: the per-feature parts of one
loop iteration come from the C restatement (oracle/ekf_oracle.c), the dense algebra runs through numpy /
LAPACK in the reference's own formulation (src/slam/main.cpp:526-530, 539-552, 561-569):

    P = F P F^T + Q                      calcP   (G = I)
    S = H P H^T + R,  K^T = LU-solve(S^T, (P H^T)^T)     calcK   (cv::solve default = LU with partial pivoting)
    P = (I - K H) P,  P = (P^T + P) / 2  updateP

TEST INFRASTRUCTURE ONLY, like oracle.py: used by tests/ and by the golden generators in this directory.
`tests/test_oracle_path.py` holds it against the C oracle's own dense loops at sizes both can run.
"""
import numpy as np
import scipy.linalg as sla

from . import oracle as orc


def step(f, u, sensors, dt, matches=()):
    """One loop iteration (src/slam/main.cpp:157-235) of the orc.Filter `f`; returns (xy, applied).
    POS / VEL overrides are left to the caller (set f.X / f.V before the call)."""
    quat = np.array(sensors.quat[:])
    prm = f.s.prm
    if u & orc.IMG:                                            # :167-175
        f.old_pos[:] = f.X
        f.update_features(matches, quat)
    alt = prm.alt_override if prm.use_alt_override else sensors.alt      # :177
    F = f.motion_jacobian(sensors, dt)                         # :192 (pre-predict state)
    f.dynamics_step(sensors, dt)                               # :193-194
    nf, n = f.nf, f.n
    if u & orc.IMG:                                            # :198-202
        meas, hmu, H, ref = f.measurement_model(f.old_pos, alt, matches, quat)
    qd = orc.init_q_diag(nf, dt, prm.Q0, prm.qbias)            # :204-205
    P = F @ f.P @ F.T                                          # :206, :526-530
    P[np.arange(n), np.arange(n)] += qd
    xy = (f.X[1], f.X[0])                                      # :207
    applied = -1
    if u & orc.IMG:                                            # :209-233
        rd = orc.init_r_diag(ref, prm.R0)
        if len(rd) != len(meas):
            raise RuntimeError("H / R row mismatch")
        W = P @ H.T                                            # :545
        S = H @ W
        S[np.arange(len(rd)), np.arange(len(rd))] += rd        # :541-543
        lu, piv = sla.lu_factor(S.T, check_finite=False)       # :547-549
        K = sla.lu_solve((lu, piv), W.T, check_finite=False).T
        A = -K @ H
        A[np.arange(n), np.arange(n)] += 1.0                   # :563-565
        P = A @ P
        P = (P.T + P) / 2                                      # :567
        e = meas - hmu                                         # view.cpp:28-47 (rows present: reflection x != 0)
        kx = K @ e                                             # :228
        V = f.V
        ok = abs(V[0] - kx[3]) + abs(V[1] - kx[4]) + abs(V[2] - kx[5]) < prm.gate      # :230
        if ok:
            f.X = f.X + kx[0:3]
            f.V = V + kx[3:6]
            for i in range(nf):
                for c in range(3):
                    f.s.feat[i].pib[c] += kx[6 + 3 * i + c]
        applied = int(ok)
    f.set_P(P)
    return xy, applied
