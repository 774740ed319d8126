"""CPU: the oracle against reference-authored answers.

* tests/golden/jacobian_h.json  -- values of the reference's own closed-form Jacobian text
  (src/slam/main.cpp:604-623), produced by oracle/gen_jacobian_golden.py.
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
