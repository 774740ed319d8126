"""CPU: the oracle against reference-authored answers.

* tests/golden/jacobian_h.json  -- values of the reference's own closed-form Jacobian text
  (src/slam/main.cpp:604-623), produced by oracle/gen_jacobian_golden.py.
* tests/golden/motion_model.json -- values of the reference's closed-form motion Jacobian blocks and feature rates
  (src/slam/main.cpp:253-264,290-299,314-317, src/ARC/states.cpp:253-256), produced by oracle/gen_motion_golden.py.
* tests/golden/euler2qbw_*.txt  -- head of src/euler2qbw/{euler,Mquat}.txt (MATLAB answers).
* tests/golden/rotate_acc*.txt  -- head of src/rotate/testing/{acc,acc.rot} (rotate -e 0 0 p).
"""
import json
import os

import numpy as np

from oracle import oracle as orc

G = os.path.join(os.path.dirname(__file__), "golden")
fh = float.fromhex


def test_jacobian_h_matches_reference_expression():
    g = json.load(open(os.path.join(G, "jacobian_h.json")))
    assert len(g["cases"]) == 96
    worst = 0.0
    for c in g["cases"]:
        v = {k: np.array([fh(x) for x in c[k]]) for k in c}
        Hb, Hi = orc.jacobian_h(v["X"], v["qbw"], v["pib"], v["anchor"], v["q0"])
        for got, ref in ((Hb, v["Hb"].reshape(4, 6)), (Hi, v["Hi"].reshape(4, 3))):
            worst = max(worst, np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
        # structural zeros spelled out at main.cpp:608,612-616
        assert np.all(Hb[:, 3:] == 0) and np.all(Hb[2:, :2] == 0)
    assert worst < 1e-12, worst


def motion_case(c):
    """Inputs of one fixture case and the motion Jacobian assembled from the reference's block values the way
    jacobianMotionModel places them (src/slam/main.cpp:266,319,329-340)."""
    v = {k: np.array([fh(x) for x in c[k]]) for k in ("qbw", "w", "V", "Fb1")}
    dt = fh(c["dt"])
    feats = [{k: np.array([fh(x) for x in f[k]]) for k in f} for f in c["features"]]
    nf = len(feats)
    n = 9 + 3 * nf
    F = np.zeros((n, n))
    F[0:6, 0:6] = v["Fb1"].reshape(6, 6)
    for i, f in enumerate(feats):
        F[6 + 3 * i:9 + 3 * i, 0:6] = f["Fib"].reshape(3, 6)
        F[6 + 3 * i:9 + 3 * i, 6 + 3 * i:9 + 3 * i] = f["FiTemp"].reshape(3, 3)
    F = dt * F + np.eye(n)
    for r in range(3):
        F[3 + r, 6 + 3 * nf + r] = -1 * dt
    return v, dt, feats, F


def oracle_filter_for(v, feats, quat):
    f = orc.Filter()
    f.update_features([(100 + i, 0.01 * i, 0.02, 0.0, 0.0) for i in range(len(feats))], quat)
    f.V = v["V"]
    for i, ft in enumerate(feats):
        f.s.feat[i].pib[:] = list(ft["pib"])
    return f


def test_motion_jacobian_and_feature_rates_match_reference_expressions():
    g = json.load(open(os.path.join(G, "motion_model.json")))
    assert len(g["cases"]) == 48
    worst_f = worst_r = 0.0
    for c in g["cases"]:
        v, dt, feats, Fref = motion_case(c)
        f = oracle_filter_for(v, feats, v["qbw"])
        acc = np.array([0.3, -0.2, 0.1])
        s = orc.make_sensors(quat=v["qbw"], acc=acc, ang=v["w"])
        F = f.motion_jacobian(s, dt)
        worst_f = max(worst_f, np.max(np.abs(F - Fref) / np.maximum(1.0, np.abs(Fref))))
        X0, pib0 = f.X, f.pib()
        f.dynamics_step(s, dt)
        want = pib0 + dt * np.array([ft["rate"] for ft in feats])
        worst_r = max(worst_r, np.max(np.abs(f.pib() - want) / np.maximum(1.0, np.abs(want))))
        # the rest of States::dynamics (states.cpp:229-233,244-246,260-262): X' = R(q) V, V' = acc - b
        assert np.allclose(f.X, X0 + dt * (orc.rotation(v["qbw"]) @ v["V"]), rtol=0, atol=1e-15)
        assert np.allclose(f.V, v["V"] + dt * acc, rtol=0, atol=1e-15)
    assert worst_f < 1e-14 and worst_r < 1e-14, (worst_f, worst_r)


def test_euler2quaternion_known_answers():
    eu = np.loadtxt(os.path.join(G, "euler2qbw_euler.txt"), delimiter=",")
    mq = np.loadtxt(os.path.join(G, "euler2qbw_mquat.txt"), delimiter=",")
    got = np.array([orc.euler2quaternion(*e) for e in eu])
    # fixture is printed with 9 significant digits
    assert np.max(np.abs(got - mq)) < 5e-9


def test_rotation_known_answers_yaw_pi():
    acc = np.loadtxt(os.path.join(G, "rotate_acc.txt"), delimiter=",")
    rot = np.loadtxt(os.path.join(G, "rotate_acc_rot.txt"), delimiter=",")
    R = orc.rotation(orc.euler2quaternion(0, 0, np.pi))
    assert np.max(np.abs(acc @ R.T - rot)) < 5e-7      # %f output, 6 decimals


def test_rotation_is_unnormalised_and_orthogonal_for_unit_q():
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        R = orc.rotation(q)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14)
        assert np.isclose(np.linalg.det(R), 1.0)
        # scales with |q|^2: the reference never normalises (Quaternion.cpp:37-45)
        assert np.allclose(orc.rotation(1.5 * q), 2.25 * R, atol=1e-14)


def test_jacobian_h_is_the_derivative_of_the_predictions():
    """Finite differences of Feature::pib0Hat / ppbHat (feature.cpp:143-198) vs jacobian_h."""
    import ctypes as C
    rng = np.random.default_rng(1)
    L = orc.lib()
    for _ in range(10):
        X = rng.standard_normal(3) * 2
        qbw = rng.standard_normal(4); qbw /= np.linalg.norm(qbw)
        q0 = rng.standard_normal(4); q0 /= np.linalg.norm(q0)
        anchor = X + rng.standard_normal(3)
        pib = np.array([0.2 * rng.standard_normal(), 0.2 * rng.standard_normal(), rng.uniform(0.05, 0.5)])
        Hb, Hi = orc.jacobian_h(X, qbw, pib, anchor, q0)

        def h(Xv, pv):
            f = orc.make_feature(pv, anchor, q0)
            a, b = np.zeros(2), np.zeros(2)
            Xc = np.ascontiguousarray(Xv); qc = np.ascontiguousarray(qbw)
            L.orc_pib0_hat(C.byref(f), orc._dp(Xc), orc._dp(qc), orc._dp(a))
            L.orc_ppb_hat(C.byref(f), orc._dp(Xc), orc._dp(qc), orc._dp(b))
            return np.concatenate([a, b])
        eps = 1e-6
        for c in range(3):
            d = np.zeros(3); d[c] = eps
            num = (h(X + d, pib) - h(X - d, pib)) / (2 * eps)
            assert np.allclose(num, Hb[:, c], rtol=1e-5, atol=1e-6)
            num = (h(X, pib + d) - h(X, pib - d)) / (2 * eps)
            assert np.allclose(num, Hi[:, c], rtol=1e-5, atol=1e-6)


def test_fir_ring_buffer_equals_causal_convolution():
    """src/lpf/fir.cpp:39-49: the ring-buffer walk is y[i] = sum_k x[i-k] c[k] with zero history."""
    rng = np.random.default_rng(2)
    for order, n in ((1, 4), (5, 3), (16, 200)):
        c, x = rng.standard_normal(order), rng.standard_normal(n)
        assert np.allclose(orc.fir(c, x), np.convolve(x, c)[:n], rtol=1e-13, atol=1e-13)


# ---- committed end-to-end fixtures (oracle/gen_stream_goldens.py) -------------------------------------------
import pytest   # noqa: E402

import helpers  # noqa: E402


class _S:       # the part of a synth.Stream that run_oracle reads besides the inputs
    def __init__(self, init_pos):
        self.init_pos = init_pos


@pytest.mark.parametrize("nf", [5, 20, 32])
def test_oracle_reproduces_the_committed_stream_fixtures(nf):
    """An edit of the oracle (or another compiler) that moves its results shows up here: the parity target of the
    GPU tests is the committed data, not whatever the oracle computes today."""
    fx = helpers.FixtureStream(os.path.join(G, "stream_nf%d.npz" % nf))
    snaps, _ = helpers.run_oracle(_S(fx.init_pos), inp=fx.inp)
    worst = helpers.compare_with_fixture(fx.expected(), snaps, 1e-12)
    assert len({len(s["ids"]) for s in snaps}) > 1          # the stream has insertions and deletions
    print("nf=%d worst deviation from the fixture %.2e" % (nf, worst))
