"""GPU: the HIP path through the C ABI against the CPU oracle.

Bar: feature ids / slot order bit-exact; X, V, pib, P within 1e-9 relative (fp64) of the oracle
on the same inputs; size-independent properties at the full benchmark sizes.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc
from slam_amd import ekf, synth, _lib
import helpers

pytestmark = pytest.mark.gpu
TOL = 1e-9
G = os.path.join(os.path.dirname(__file__), "golden")


def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def compare_snaps(so, sg, tol=TOL):
    assert len(so) == len(sg)
    worst = 0.0
    for k, (a, b) in enumerate(zip(so, sg)):
        assert a["ids"] == b["ids"], "association differs at step %d" % k
        assert a["applied"] == b["applied"], "gate differs at step %d" % k
        for key in ("X", "V", "pib", "P"):
            e = helpers.rel_err(b[key], a[key])
            assert e < tol, "step %d %s rel err %.3e" % (k, key, e)
            worst = max(worst, e)
        # entry-wise as well (every entry of P against sqrt(P_ii P_jj)): a norm-wise bound alone would let the small
        # cross terms, e.g. those of the 1e-4 bias block, be wrong
        e = helpers.entry_err(b["P"], a["P"])
        assert e < tol, "step %d P entry-wise err %.3e" % (k, e)
        worst = max(worst, e)
        assert "%0.5f,%0.5f" % a["xy"] == "%0.5f,%0.5f" % b["xy"]
    return worst


# ------------------------------------------------------------------------------------------
def test_dgemm_kernel_against_numpy():
    torch = _torch()
    L = _lib.lib()
    rng = np.random.default_rng(0)
    for (M, N, K, batch) in [(64, 64, 64, 1), (112, 208, 112, 3), (208, 208, 112, 2), (640, 528, 96, 1),
                             (1024, 512, 512, 1), (16, 16, 16, 5)]:
        for tb in (0, 1):
            A = rng.standard_normal((batch, M, K))
            Bm = rng.standard_normal((batch, N, K) if tb else (batch, K, N))
            C0 = rng.standard_normal((batch, M, N))
            dA, dB, dC = (torch.from_numpy(x).cuda() for x in (A, Bm, C0))
            rc = L.slamk_dgemm_dev(tb, M, N, K, 0.5, dA.data_ptr(), K, M * K, dB.data_ptr(), K if tb else N,
                                   Bm.shape[1] * Bm.shape[2], -2.0, dC.data_ptr(), N, M * N, batch, None)
            assert rc == 0
            torch.cuda.synchronize()
            ref = 0.5 * np.einsum("bmk,bnk->bmn" if tb else "bmk,bkn->bmn", A, Bm) - 2.0 * C0
            assert helpers.rel_err(dC.cpu().numpy(), ref) < 1e-13, (M, N, K, batch, tb)
    # operand-order check with structured data: A = I, asymmetric B
    n = 64
    A = np.eye(n)
    Bm = np.arange(n * n, dtype=np.float64).reshape(n, n)
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(Bm).cuda()
    dC = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    L.slamk_dgemm_dev(0, n, n, n, 1.0, dA.data_ptr(), n, 0, dB.data_ptr(), n, 0, 0.0, dC.data_ptr(), n, 0, 1, None)
    torch.cuda.synchronize()
    assert np.array_equal(dC.cpu().numpy(), Bm)
    L.slamk_dgemm_dev(1, n, n, n, 1.0, dA.data_ptr(), n, 0, dB.data_ptr(), n, 0, 0.0, dC.data_ptr(), n, 0, 1, None)
    torch.cuda.synchronize()
    assert np.array_equal(dC.cpu().numpy(), Bm.T)


@pytest.mark.parametrize("nf,form", [(5, 0), (20, 0), (20, 1), (32, 0), (32, 1), (5, 2), (20, 2), (32, 2)])
def test_single_filter_stream_parity(nf, form):
    st = synth.gen_stream(seed=10 + nf, nf=nf, steps=61, churn=True, drop_at=21, drop_count=min(3, nf - 1),
                          add_at=42, add_count=3)
    so, _ = helpers.run_oracle(st)
    p = ekf.default_params()
    p.update_form = form
    fb = ekf.FilterBatch(max_features=nf + 4, batch=1, params=p)
    sg = helpers.run_gpu(st, fb)
    worst = compare_snaps(so, sg)
    nfs = {len(s["ids"]) for s in sg}
    assert len(nfs) > 1                      # the stream really exercised insert / delete
    print("nf=%d form=%d worst rel err %.2e" % (nf, form, worst))


@pytest.mark.parametrize("nf,ref_fraction,form", [(2, 0.5, 0), (38, 0.0, 0), (38, 0.0, 1), (26, 1.0, 0), (36, 1.0, 0),
                                                  (13, 0.5, 0), (29, 0.5, 0), (1, 1.0, 0), (34, 0.5, 0), (36, 0.5, 0),
                                                  (39, 0.5, 0), (39, 1.0, 0), (39, 1.0, 1), (32, 1.0, 0),
                                                  (2, 0.5, 2), (38, 0.0, 2), (36, 1.0, 2), (13, 0.5, 2), (1, 1.0, 2),
                                                  (50, 0.5, 2), (45, 1.0, 2),
                                                  (30, 0.0, 0), (30, 1.0, 0), (34, 1.0, 0), (34, 1.0, 1), (32, 0.0, 0),
                                                  (32, 0.0, 1), (31, 0.3, 0), (33, 0.7, 0)])
def test_fused_path_tile_count_edges(nf, ref_fraction, form):
    """Smallest filter, eight 16-row blocks of the state (n = 123), all-reflection rows (m = 157), and a filter
    whose S exceeds LDS (m = 171 .. 235: the solver keeps the finished tiles of the factor in an image in global
    memory, 11 to 15 tile columns) while its P still fits the fused covariance update (n <= 126); state
    dimensions that fill their tiles exactly (n = 48, 96: no padding rows); a single reflective feature.  Form 2 (the
    block-sequential update) at the same edges and at its own: n = 159 (ten row blocks, its limit) and m = 271.
    The last line: seven row blocks (n = 99 .. 111), where the covariance update runs with four wavefronts per filter:
    8 to 13 chunks of measurement columns, the altitude row inside the chunks and as the border (m = 129), both forms."""
    st = synth.gen_stream(seed=70 + nf, nf=nf, steps=31, churn=False, ref_fraction=ref_fraction)
    so, _ = helpers.run_oracle(st)
    p = ekf.default_params()
    p.update_form = form
    fb = ekf.FilterBatch(max_features=nf, batch=1, params=p)
    sg = helpers.run_gpu(st, fb)
    worst = compare_snaps(so, sg)
    print("nf=%d ref=%.1f form=%d worst rel err %.2e" % (nf, ref_fraction, form, worst))


@pytest.mark.parametrize("nf,ref_fraction,form", [(40, 0.5, 0), (40, 0.0, 0), (40, 0.0, 1), (40, 1.0, 0), (43, 0.5, 0),
                                                  (45, 0.3, 0), (47, 0.5, 0), (50, 0.0, 0), (52, 0.0, 0), (56, 0.0, 0),
                                                  (58, 0.0, 1), (60, 0.0, 0), (61, 0.0, 0), (48, 0.5, 0)])
def test_wide_gain_solver_edges(nf, ref_fraction, form):
    """Nine to twelve 16-row blocks of the state (n = 129 .. 189) with up to 240 tiled measurement rows: the gain comes
    from the per-filter solver's wide form (row blocks 8.. in a second pass) and feeds the covariance update by GEMMs.
    n = 129 (one row in the ninth block), n = 144 (no padding rows), the altitude row as border with 10, 12, 13, 14
    and 15 tile columns, all-reflection rows (m = 241: 16 tile columns, back on the blocked path), and both sides of
    the range's end (60 features: n = 189, m = 241 as 240 + border; 61: n = 192, m = 245)."""
    st = synth.gen_stream(seed=170 + nf, nf=nf, steps=31, churn=False, ref_fraction=ref_fraction)
    so, _ = helpers.run_oracle(st)
    p = ekf.default_params()
    p.update_form = form
    fb = ekf.FilterBatch(max_features=nf, batch=1, params=p)
    sg = helpers.run_gpu(st, fb)
    worst = compare_snaps(so, sg)
    print("nf=%d ref=%.1f form=%d worst rel err %.2e" % (nf, ref_fraction, form, worst))


def test_map_growing_through_every_path_in_one_handle():
    """One handle whose map grows 30 -> 37 -> 45 features: per-filter kernels, then the GEMM gain with the fused
    covariance update, then the all-GEMM path; the larger paths' workspaces first appear in the middle of the run."""
    st = synth.gen_stream(seed=91, nf=30, steps=46, churn=False, ref_fraction=0.5, add_at=15, add_count=7)
    st2 = synth.gen_stream(seed=91, nf=30, steps=46, churn=False, ref_fraction=0.5, add_at=15, add_count=15)
    for stream, nf_end in ((st, 37), (st2, 45)):
        so, _ = helpers.run_oracle(stream)
        fb = ekf.FilterBatch(max_features=48, batch=1)
        sg = helpers.run_gpu(stream, fb)
        assert len(sg[-1]["ids"]) == nf_end
        worst = compare_snaps(so, sg)
        print("30 -> %d features: worst rel err %.2e" % (nf_end, worst))


def test_map_that_empties_and_refills():
    """Every feature vanishes (altitude-only updates: n = 9, m = 1), later frames bring four new ones."""
    st = synth.gen_stream(seed=5, nf=3, steps=30, churn=False, drop_at=9, drop_count=3, add_at=18, add_count=4)
    so, _ = helpers.run_oracle(st)
    fb = ekf.FilterBatch(max_features=6, batch=1)
    sg = helpers.run_gpu(st, fb)
    assert min(len(s["ids"]) for s in sg) == 0 and len(sg[-1]["ids"]) == 4
    worst = compare_snaps(so, sg)
    print("empty map: worst rel err %.2e" % worst)


def test_altimeter_input_without_the_override():
    """use_alt_override = 0: the first measurement is the altimeter value of the step (the reference overwrites it
    with -1.34, src/slam/main.cpp:177-178; the switch exists so that callers with a live altimeter can use it)."""
    st = synth.gen_stream(seed=41, nf=9, steps=25, churn=False)
    inp = helpers.stream_inputs(st, 7)
    po = orc.default_params()
    po.use_alt_override = 0
    f = orc.Filter(po)
    f.X = st.init_pos
    f.update_features(inp[0][5], inp[0][1])
    f.old_pos[:] = f.X
    pg = ekf.default_params()
    pg.use_alt_override = 0
    fb = ekf.FilterBatch(max_features=9, batch=1, params=pg)
    fb.set_state(0, X=st.init_pos)
    fb.frame(0, inp[0][5], inp[0][1])
    rng = np.random.default_rng(3)
    updates = 0
    for k in range(1, len(inp)):
        u, quat, ang, acc, dt, matches = inp[k]
        alt = -1.3 + 0.05 * rng.standard_normal()
        rc, _, applied = f.step(u | orc.ALT, orc.make_sensors(quat=quat, acc=acc, ang=ang, alt=alt), dt, matches)
        assert rc == 0
        if u & orc.IMG:
            fb.frame(0, matches, quat)
        fb.predict(quat, ang, acc, [dt])
        if u & orc.IMG:
            assert int(fb.update(quat, alt=[alt])[0]) == applied
            updates += 1
        assert helpers.rel_err(fb.get_state(0)[0], f.X) < TOL and helpers.rel_err(fb.get_cov(0), f.P) < TOL
    assert updates >= 5


@pytest.mark.parametrize("seed", [2024, 7, 99])
def test_random_configurations_against_oracle(seed):
    """Seeded sweep over feature counts, reflection fractions, churn patterns and the three update forms."""
    rng = np.random.default_rng(seed)
    worst = 0.0
    for case in range(14):
        nf = int(rng.integers(1, 44))
        ref = float(rng.choice([0.0, 0.25, 0.5, 0.75, 1.0]))
        steps = int(rng.integers(10, 28))
        churn = bool(rng.integers(0, 2))
        kw = {}
        if rng.integers(0, 2) and nf > 2:
            kw.update(drop_at=int(rng.integers(3, steps - 3)), drop_count=int(rng.integers(1, min(nf, 5))))
        if rng.integers(0, 2):
            kw.update(add_at=int(rng.integers(3, steps - 2)), add_count=int(rng.integers(1, 6)))
        st = synth.gen_stream(seed=seed + 500 + case, nf=nf, steps=steps, churn=churn, ref_fraction=ref, **kw)
        so, _ = helpers.run_oracle(st)
        p = ekf.default_params()
        p.update_form = int(rng.integers(0, 3))
        fb = ekf.FilterBatch(max_features=nf + 6, batch=1, params=p)
        sg = helpers.run_gpu(st, fb)
        worst = max(worst, compare_snaps(so, sg))
        fb.close()
    print("worst rel err over the sweep %.2e" % worst)


@pytest.mark.parametrize("nf,ref_fraction", [(14, None), (40, 0.0), (43, 0.5)])
def test_per_function_parity(nf, ref_fraction):
    """F, H, innovation, Cholesky factor and gain of one step against the oracle's functions; also where the factor comes
    from the solver's wide form and its image (129 and 138 states), with the altitude row as border (m = 161) and inside
    the tiles."""
    kw = {} if ref_fraction is None else {"ref_fraction": ref_fraction}
    st = synth.gen_stream(seed=4, nf=nf, steps=7, churn=False, **kw)
    inp = helpers.stream_inputs(st, 7)
    f = orc.Filter()
    f.X = st.init_pos
    fb = ekf.FilterBatch(max_features=nf, batch=1)
    fb.set_debug(True)
    fb.set_state(0, X=st.init_pos)
    u, quat, ang, acc, dt, matches = inp[0]
    f.update_features(matches, quat)
    f.old_pos[:] = f.X
    fb.frame(0, matches, quat)
    for k in (1, 2):
        u, quat, ang, acc, dt, matches = inp[k]
        f.step(u, orc.make_sensors(quat=quat, acc=acc, ang=ang), dt, matches)
        fb.predict(quat, ang, acc, [dt])
    u, quat, ang, acc, dt, matches = inp[3]
    assert u & orc.IMG
    s = orc.make_sensors(quat=quat, acc=acc, ang=ang)
    f.old_pos[:] = f.X
    f.update_features(matches, quat)
    F = f.motion_jacobian(s, dt)
    f.dynamics_step(s, dt)
    meas, hmu, H, ref = f.measurement_model(f.old_pos, -1.34, matches, quat)
    P = orc.calc_p(f.P, F, orc.init_q_diag(f.nf, dt))
    rd = orc.init_r_diag(ref)
    K = orc.calc_k(H, P, rd)
    fb.frame(0, matches, quat)
    fb.predict(quat, ang, acc, [dt])
    assert helpers.rel_err(fb.debug_get_F(0), F) < 1e-14
    assert helpers.rel_err(fb.get_cov(0), P) < 1e-13
    assert helpers.rel_err(fb.state_vector(0), f.state_vector()) < 1e-14
    fb.update(quat)
    d = fb.debug_get(0)
    assert d["H"].shape == H.shape
    assert helpers.rel_err(d["H"], H) < 1e-13
    assert np.array_equal(d["H"] == 0, H == 0)              # same sparsity pattern
    assert helpers.rel_err(d["e"], meas - hmu) < 1e-13
    S = H @ P @ H.T + np.diag(rd)
    perm = np.r_[1:S.shape[0], 0]                            # the library factors S with the altitude row last
    assert helpers.rel_err(np.tril(d["L"]), np.linalg.cholesky(S[np.ix_(perm, perm)])) < 1e-12
    assert helpers.rel_err(d["K"], K) < 1e-11


def test_reference_jacobian_golden_through_the_kernel():
    """tests/golden/jacobian_h.json (values of the reference's closed-form text) vs measure_kernel."""
    g = json.load(open(os.path.join(G, "jacobian_h.json")))
    fh = float.fromhex
    fb = ekf.FilterBatch(max_features=1, batch=1)
    fb.set_debug(True)
    worst = 0.0
    for c in g["cases"][:48]:
        v = {k: np.array([fh(x) for x in c[k]]) for k in c}
        fb.set_features(0, [7], v["pib"], v["anchor"], v["q0"], v["pib"][:2], [1])
        fb.set_state(0, X=v["X"], V=np.zeros(3))
        fb.set_cov(0, np.eye(12))
        fb.set_frame_values(0, [0.1, 0.2], [0.3, 0.4])
        fb.update(v["qbw"])
        H = fb.debug_get(0)["H"]
        Hb, Hi = v["Hb"].reshape(4, 6), v["Hi"].reshape(4, 3)
        got_b, got_i = H[3:7, 0:6], H[3:7, 6:9]
        for got, ref in ((got_b, Hb), (got_i, Hi)):
            worst = max(worst, np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))))
    assert worst < 1e-12, worst


def test_reference_motion_model_golden_through_the_kernels():
    """tests/golden/motion_model.json (values of the reference's closed-form motion Jacobian blocks and feature
    rates) vs motion_rows_kernel / build_F_kernel and dynamics_kernel."""
    from test_oracle_golden import motion_case
    g = json.load(open(os.path.join(G, "motion_model.json")))
    worst_f = worst_r = 0.0
    for c in g["cases"]:
        v, dt, feats, Fref = motion_case(c)
        nf = len(feats)
        fb = ekf.FilterBatch(max_features=nf, batch=1)
        fb.set_debug(True)
        pib = np.array([ft["pib"] for ft in feats])
        fb.set_features(0, list(range(nf)), pib, np.zeros((nf, 3)), np.tile([0.0, 0, 0, 1], (nf, 1)), pib[:, :2],
                        [0] * nf)
        fb.set_state(0, X=np.zeros(3), V=v["V"])
        fb.set_cov(0, np.eye(9 + 3 * nf))
        fb.predict(v["qbw"], v["w"], [0.3, -0.2, 0.1], [dt])
        F = fb.debug_get_F(0)
        worst_f = max(worst_f, np.max(np.abs(F - Fref) / np.maximum(1.0, np.abs(Fref))))
        want = pib + dt * np.array([ft["rate"] for ft in feats])
        got = fb.get_features(0)["pib"]
        worst_r = max(worst_r, np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))))
        # P = F P F^T + Q with P = I: the propagated covariance is built from the same F
        q0 = 256 * dt * dt
        Q = np.diag([q0 * 0.1 * q0] * 3 + [q0] * (3 + 3 * nf) + [1e-5 * dt * dt] * 3)
        assert helpers.rel_err(fb.get_cov(0), Fref @ Fref.T + Q) < 1e-13
        fb.close()
    assert worst_f < 1e-14 and worst_r < 1e-14, (worst_f, worst_r)


def test_batch_of_independent_filters():
    """Filters of one batch see the same sensors but independent pixel noise; each must follow
    its own oracle run."""
    B, nf = 6, 9
    st = synth.gen_stream(seed=21, nf=nf, steps=16, churn=False)
    fb = ekf.FilterBatch(max_features=nf, batch=B)
    oracles, snaps_o = [], []
    for b in range(B):
        so, _ = helpers.run_oracle(st, noise_seed=1000 + b)
        snaps_o.append(so)
    inps = [helpers.stream_inputs(st, 1000 + b) for b in range(B)]
    for b in range(B):
        fb.set_state(b, X=st.init_pos)
        fb.frame(b, inps[b][0][5], inps[b][0][1])
    for k in range(1, len(st.steps)):
        u, quat, ang, acc, dt, _ = inps[0][k]
        if u & orc.IMG:
            for b in range(B):
                fb.frame(b, inps[b][k][5], quat)
        fb.predict(np.tile(quat, B), np.tile(ang, B), np.tile(acc, B), np.full(B, dt))
        if u & orc.IMG:
            applied = fb.update(np.tile(quat, B))
        for b in range(B):
            a = snaps_o[b][k - 1]
            assert helpers.rel_err(fb.get_cov(b), a["P"]) < TOL
            assert helpers.rel_err(fb.state_vector(b)[:-3], np.concatenate([a["X"], a["V"], a["pib"].ravel()])) < TOL
            if u & orc.IMG:
                assert applied[b] == a["applied"]


def test_shared_frame_with_feature_churn_is_applied_to_the_whole_batch():
    """frame(-1, ...) (one association, batched surgery) == frame(b, ...) per filter, with features replaced,
    dropped and added along the stream; filter 0 also follows the oracle."""
    B, nf = 5, 10
    st = synth.gen_stream(seed=41, nf=nf, steps=40, churn=True, drop_at=12, drop_count=3, add_at=24, add_count=4)
    so, _ = helpers.run_oracle(st)
    inp = helpers.stream_inputs(st, 7)
    fbs = [ekf.FilterBatch(max_features=nf + 5, batch=B) for _ in range(2)]
    for fb in fbs:
        for b in range(B):
            fb.set_state(b, X=st.init_pos + np.array([0.0, 0.003 * b, 0.0]))
    for b in range(B):
        fbs[0].frame(b, inp[0][5], inp[0][1])
    fbs[1].frame(-1, inp[0][5], inp[0][1])
    counts = set()
    for k in range(1, len(inp)):
        u, quat, ang, acc, dt, matches = inp[k]
        if u & orc.IMG:
            for b in range(B):
                fbs[0].frame(b, matches, quat)
            fbs[1].frame(-1, matches, quat)
        for fb in fbs:
            fb.predict(np.tile(quat, B), np.tile(ang, B), np.tile(acc, B), np.full(B, dt))
            if u & orc.IMG:
                fb.update(np.tile(quat, B))
        counts.add(fbs[1].num_features(0))
        a = so[k - 1]
        assert helpers.rel_err(fbs[1].get_cov(0), a["P"]) < TOL
        assert list(fbs[1].get_features(0)["ids"]) == list(a["ids"])
    assert len(counts) > 1
    for b in range(B):
        assert np.array_equal(fbs[0].get_cov(b), fbs[1].get_cov(b))
        assert np.array_equal(fbs[0].state_vector(b), fbs[1].state_vector(b))
        assert list(fbs[0].get_features(b)["ids"]) == list(fbs[1].get_features(b)["ids"])
    for fb in fbs:
        fb.close()


@pytest.mark.parametrize("groups,form,nf", [(1, 0, 12), (3, 0, 12), (1, 2, 12), (3, 2, 12), (1, 2, 33), (1, 2, 50), (1, 2, 1),
                                            (3, 0, 44), (2, 0, 36)])
def test_step_call_matches_the_three_calls(groups, form, nf):
    import torch
    """slamk_step_dev (whole iteration, batch cut into ranges on streams of their own) == frame_begin + predict + update.
    Bit for bit in the default form.  In form 2 a small batch takes the whole iteration in ONE launch (seq_step_kernel:
    the covariance is propagated on its lower tiles in LDS, blocks in another order than the stand-alone pass), so the
    two agree to rounding; the gate results are identical."""
    B = 7
    st = synth.gen_stream(seed=31, nf=nf, steps=9, churn=False, image_every=1)
    s0 = st.steps[0]
    first = [(int(s0.ids[i]), s0.src[i, 0], s0.src[i, 1], s0.refl[i, 0], s0.refl[i, 1]) for i in range(nf)]
    fbs = []
    prm = ekf.default_params()
    prm.update_form = form
    for _ in range(2):
        fb = ekf.FilterBatch(max_features=nf, batch=B, params=prm)
        for b in range(B):
            fb.set_state(b, X=st.init_pos + 0.01 * b)
        fb.frame(-1, first, s0.quat)
        fbs.append(fb)
    fbs[1].set_pipeline_groups(groups)
    rng = np.random.default_rng(5)
    dev = torch.device("cuda")
    for k in range(1, 9):
        s, prev = st.steps[k], st.steps[k - 1]
        one = lambda v: torch.tensor(np.tile(np.asarray(v, dtype=np.float64), (B, 1)), device=dev)
        q, w, a = one(s.quat), one(s.ang), one(s.acc)
        dt = torch.full((B,), s.t - prev.t, device=dev, dtype=torch.float64)
        src = torch.tensor(s.src.T[None] + 1e-3 * rng.standard_normal((B, 2, nf)), device=dev)
        refl = torch.tensor(s.refl.T[None] * (1 + 1e-3 * rng.standard_normal((B, 2, nf))), device=dev)
        ap = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
        fbs[0].frame_begin_dev(0)
        fbs[0].predict_dev(q.data_ptr(), w.data_ptr(), a.data_ptr(), dt.data_ptr(), 0)
        fbs[0].update_dev(q.data_ptr(), None, src.data_ptr(), refl.data_ptr(), ap[0].data_ptr(), 0)
        fbs[1].step_dev(q.data_ptr(), w.data_ptr(), a.data_ptr(), dt.data_ptr(), None, src.data_ptr(), refl.data_ptr(),
                        ap[1].data_ptr(), 0)
        fbs[0].sync(0)
        fbs[1].sync(0)
        assert torch.equal(ap[0], ap[1])
    for b in range(B):
        P0, P1 = fbs[0].get_cov(b), fbs[1].get_cov(b)
        if form == 0:
            assert np.array_equal(P0, P1)
            assert np.array_equal(fbs[0].state_vector(b), fbs[1].state_vector(b))
        else:
            assert np.array_equal(P1, P1.T)
            assert helpers.entry_err(P1, P0) < 1e-11 and helpers.rel_err(P1, P0) < 1e-12
            assert helpers.rel_err(fbs[1].state_vector(b), fbs[0].state_vector(b)) < 1e-12
    for fb in fbs:
        fb.close()


def test_graph_mode_replays_the_same_iteration():
    """Graph mode of slamk_step_dev: captured once per buffer parity, replayed, bit-identical to plain launches."""
    import torch
    nf = 20
    st = synth.gen_stream(seed=32, nf=nf, steps=12, churn=False, image_every=1)
    s0 = st.steps[0]
    first = [(int(s0.ids[i]), s0.src[i, 0], s0.src[i, 1], s0.refl[i, 0], s0.refl[i, 1]) for i in range(nf)]
    fbs = []
    for _ in range(2):
        fb = ekf.FilterBatch(max_features=nf, batch=1)
        fb.set_state(0, X=st.init_pos)
        fb.frame(0, first, s0.quat)
        fbs.append(fb)
    fbs[1].set_graph_mode(True)
    dev = torch.device("cuda")
    stream = torch.cuda.Stream(device=dev)
    buf = {k: torch.zeros(n, dtype=torch.float64, device=dev) for k, n in
           (("q", 4), ("w", 3), ("a", 3), ("dt", 1), ("src", 2 * nf), ("refl", 2 * nf))}
    for k in range(1, 12):
        s, prev = st.steps[k], st.steps[k - 1]
        with torch.cuda.stream(stream):
            buf["q"].copy_(torch.tensor(s.quat)); buf["w"].copy_(torch.tensor(s.ang)); buf["a"].copy_(torch.tensor(s.acc))
            buf["dt"].fill_(s.t - prev.t)
            buf["src"].copy_(torch.tensor(s.src.T.copy().ravel())); buf["refl"].copy_(torch.tensor(s.refl.T.copy().ravel()))
        for fb in fbs:
            fb.step_dev(buf["q"].data_ptr(), buf["w"].data_ptr(), buf["a"].data_ptr(), buf["dt"].data_ptr(), None,
                        buf["src"].data_ptr(), buf["refl"].data_ptr(), None, stream.cuda_stream)
            fb.sync(stream.cuda_stream)
        assert np.array_equal(fbs[0].get_cov(0), fbs[1].get_cov(0)), k
        assert np.array_equal(fbs[0].state_vector(0), fbs[1].state_vector(0)), k
    for fb in fbs:
        fb.close()


@pytest.mark.parametrize("nfs", [(4, 11, 7), (32, 5, 20, 33), (45, 12, 40, 33)])
def test_mixed_feature_counts_in_one_batch(nfs):
    """Filters with different nf share one launch (dimensions are padded to the batch maximum); the second set is
    launched with seven row blocks (the covariance update with four wavefronts per filter) for filters of one to
    seven; the third with ten (the gain solver's wide form and its second kernel for row blocks 8 and 9, which the
    smaller filters of the batch do not have)."""
    sts = [synth.gen_stream(seed=30 + i, nf=nf, steps=7, churn=False) for i, nf in enumerate(nfs)]
    sts.append(synth.gen_stream(seed=34, nf=3, steps=7, churn=False, drop_at=3, drop_count=3))   # empties at the 2nd frame
    B = len(sts)
    fb = ekf.FilterBatch(max_features=max(nfs), batch=B)
    inps = [helpers.stream_inputs(st, 5) for st in sts]
    so = [helpers.run_oracle(st, noise_seed=5)[0] for st in sts]
    for b in range(B):
        fb.set_state(b, X=sts[b].init_pos)
        fb.frame(b, inps[b][0][5], inps[b][0][1])
    for k in range(1, 7):
        img = inps[0][k][0] & orc.IMG
        q = np.concatenate([inps[b][k][1] for b in range(B)])
        w = np.concatenate([inps[b][k][2] for b in range(B)])
        a = np.concatenate([inps[b][k][3] for b in range(B)])
        dt = np.array([inps[b][k][4] for b in range(B)])
        if img:
            for b in range(B):
                fb.frame(b, inps[b][k][5], inps[b][k][1])
        fb.predict(q, w, a, dt)
        if img:
            fb.update(q)
        for b in range(B):
            assert helpers.rel_err(fb.get_cov(b), so[b][k - 1]["P"]) < TOL


def test_error_paths():
    fb = ekf.FilterBatch(max_features=3, batch=1)
    q = [0, 0, 0, 1]
    with pytest.raises(ekf.SlamkError) as ei:
        fb.frame(0, [(i, 0.1 * i, 0.1, 0, 0) for i in range(5)], q)
    assert ei.value.code == _lib.SLAMK_ECAPACITY
    # a feature initialised with a reflection whose observation later is (0,0): the reference
    # aborts on the H / R / e row mismatch (main.cpp:211-218,353; view.cpp:17,39)
    fb = ekf.FilterBatch(max_features=3, batch=1)
    fb.frame(0, [(1, 0.1, 0.1, 0.1, 0.3), (2, 0.2, -0.1, 0, 0)], q)
    fb.predict(q, [0, 0, 0], [0, 0, 0], [0.1])
    fb.frame(0, [(1, 0.1, 0.1, 0, 0), (2, 0.2, -0.1, 0, 0)], q)
    with pytest.raises(ekf.SlamkError) as ei:
        fb.update(q)
    assert ei.value.code == _lib.SLAMK_EREFL
    with pytest.raises(ekf.SlamkError):
        fb.set_cov(0, np.eye(9))            # n must match the feature count


@pytest.mark.parametrize("nf,form,batch", [(3, 0, 2), (60, 0, 2), (3, 2, 2), (20, 2, 2), (3, 2, 300), (20, 2, 300)])
def test_indefinite_innovation_covariance_is_reported(nf, form, batch):
    """A covariance that is not positive semi-definite makes S = H P H^T + R indefinite: the Cholesky gain does
    not exist and slamk_sync says so (small fused path and blocked large path; in the block-sequential form, whose
    16 x 16 innovation blocks are factored one by one, through both of its kernels: the separate update kernel and,
    for batches of up to 256 filters, the one-launch step)."""
    p = ekf.default_params()
    p.update_form = form
    fb = ekf.FilterBatch(max_features=nf, batch=batch, params=p)
    q = [0, 0, 0, 1]
    fb.frame(-1, [(i + 1, 0.05 * i - 0.1, 0.02 * i, 0, 0) for i in range(nf)], q)
    n = 9 + 3 * nf
    fb.set_cov(1, -50.0 * np.eye(n))
    fb.predict(np.tile(q, batch), np.zeros(3 * batch), np.zeros(3 * batch), [0.03] * batch)
    with pytest.raises(ekf.SlamkError) as ei:
        fb.update(np.tile(q, batch))
    assert ei.value.code == _lib.SLAMK_ENOTSPD and "filter 1 (1 of %d" % batch in str(ei.value)
    fb.close()


def test_gate_rejects_mean_update_but_not_covariance():
    """main.cpp:230-232: P is updated even when sum|V - dV| >= gate."""
    st = synth.gen_stream(seed=8, nf=6, steps=4, churn=False)
    inp = helpers.stream_inputs(st, 3)
    p = ekf.default_params()
    p.gate = 1e-6
    po = orc.default_params()
    po.gate = 1e-6
    f = orc.Filter(po)
    f.X = st.init_pos
    fb = ekf.FilterBatch(max_features=6, batch=1, params=p)
    fb.set_state(0, X=st.init_pos)
    f.update_features(inp[0][5], inp[0][1])
    f.old_pos[:] = f.X
    fb.frame(0, inp[0][5], inp[0][1])
    for k in (1, 2, 3):
        u, quat, ang, acc, dt, matches = inp[k]
        rc, xy, applied = f.step(u, orc.make_sensors(quat=quat, acc=acc, ang=ang), dt, matches)
        if u & orc.IMG:
            fb.frame(0, matches, quat)
        fb.predict(quat, ang, acc, [dt])
        if u & orc.IMG:
            ap = fb.update(quat)[0]
            assert ap == applied == 0
    assert helpers.rel_err(fb.get_cov(0), f.P) < TOL
    assert helpers.rel_err(fb.state_vector(0), f.state_vector()) < TOL


def test_large_map_against_oracle():
    """nf = 220 (n = 669, m > 1024): 128-wide tiles, 128-wide Cholesky panels, recursive inverse."""
    nf = 220
    st = synth.gen_stream(seed=2, nf=nf, steps=4, churn=False)
    so, _ = helpers.run_oracle(st)
    fb = ekf.FilterBatch(max_features=nf, batch=1)
    sg = helpers.run_gpu(st, fb)
    assert fb.num_rows(0) > 1024
    compare_snaps(so, sg)


def test_large_map_with_small_blocks_runs_the_multi_block_solves():
    """The same filter with 256-column outer panels and 256-row inverse blocks: several outer panels in the
    factorisation and five block steps in each triangular solve, the code path of maps with m > 4096."""
    nf = 220
    st = synth.gen_stream(seed=2, nf=nf, steps=4, churn=False)
    so, _ = helpers.run_oracle(st)
    ekf.set_block_sizes(256, 256)
    try:
        fb = ekf.FilterBatch(max_features=nf, batch=1)
        sg = helpers.run_gpu(st, fb)
    finally:
        ekf.set_block_sizes(512, 4096)
    compare_snaps(so, sg)


def test_full_size_batch_properties():
    """4096 filters x 32 features (the benchmark shape): a sample of filters against the oracle and
    symmetry / finiteness / positive diagonal of every covariance."""
    torch = _torch()
    B, nf = 4096, 32
    st = synth.gen_stream(seed=1, nf=nf, steps=4, churn=False, image_every=1)
    fb = ekf.FilterBatch(max_features=nf, batch=B)
    sample = [0, 1, 2047, 4095]
    inps = {b: helpers.stream_inputs(st, 1000 + b) for b in sample}
    base = helpers.stream_inputs(st, 999)
    for b in sample:
        inps[b][0] = base[0]                  # every filter is initialised from the same first frame
    so = {b: helpers.run_oracle(st, inp=inps[b])[0] for b in sample}
    for b in range(B):
        fb.set_state(b, X=st.init_pos)
    fb.frame(-1, base[0][5], base[0][1])
    for k in range(1, 4):
        u, quat, ang, acc, dt, matches = base[k]
        fb.frame(-1, matches, quat)
        for b in sample:
            fb.frame(b, inps[b][k][5], quat)
        fb.predict(np.tile(quat, B), np.tile(ang, B), np.tile(acc, B), np.full(B, dt))
        fb.update(np.tile(quat, B))
        for b in sample:
            assert helpers.rel_err(fb.get_cov(b), so[b][k - 1]["P"]) < TOL
    for b in (3, 100, 3000):
        P = fb.get_cov(b)
        assert np.all(np.isfinite(P)) and np.array_equal(P, P.T) and np.all(np.diag(P) > 0)
        assert np.linalg.eigvalsh(P).min() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("nf,ref_fraction", [(4, 0.5), (8, 0.5), (20, 0.5), (24, 0.0), (28, 1.0), (32, 0.5), (38, 0.0),
                                             (36, 0.5), (36, 1.0), (39, 1.0), (40, 0.0), (44, 0.5), (47, 1.0)])
def test_identical_filters_give_identical_results(nf, ref_fraction):
    """1024 filters with the same inputs: the per-filter kernels run the same instructions on the same data in every
    workgroup, so every covariance and state must equal those of filter 0 bit for bit, whatever CU a workgroup lands on
    and however its wavefronts interleave; a missing barrier or a read ahead of its producer shows up here.  Filter 0
    is held to the oracle."""
    B = 1024
    st = synth.gen_stream(seed=5, nf=nf, steps=4, churn=False, image_every=1, ref_fraction=ref_fraction)
    inp = helpers.stream_inputs(st, 77)
    so = helpers.run_oracle(st, inp=inp)[0]
    fb = ekf.FilterBatch(max_features=nf, batch=B)
    for b in range(B):
        fb.set_state(b, X=st.init_pos)
    fb.frame(-1, inp[0][5], inp[0][1])
    for k in range(1, 4):
        u, quat, ang, acc, dt, matches = inp[k]
        fb.frame(-1, matches, quat)
        fb.predict(np.tile(quat, B), np.tile(ang, B), np.tile(acc, B), np.full(B, dt))
        fb.update(np.tile(quat, B))
        P0, x0 = fb.get_cov(0), fb.state_vector(0)
        assert helpers.rel_err(P0, so[k - 1]["P"]) < TOL
        for b in range(1, B, 3):
            assert np.array_equal(fb.get_cov(b), P0), (k, b)
            assert np.array_equal(fb.state_vector(b), x0), (k, b)
    fb.close()


def _run_sharded(rank, world, group, nf, steps, dev, rccl_single_rank=False, blocks=None):
    """The stream of the sharded tests through slamk_sharded_* on this rank; returns (P, state vector).
    blocks: (outer panel columns, rows of the inverse blocks) for this run."""
    import torch
    from slam_amd.sharded import ShardedFilter
    ekf.set_block_sizes(*(blocks or (512, 4096)))
    st = synth.gen_stream(seed=17, nf=nf, steps=steps, churn=False)
    inp = helpers.stream_inputs(st, 7)
    fb = ekf.FilterBatch(max_features=nf, batch=1)
    fb.set_state(0, X=st.init_pos)
    fb.frame(0, inp[0][5], inp[0][1])
    stream = torch.cuda.Stream(device=dev)
    sh = ShardedFilter(fb, rank, world, group, device=dev, stream=stream, rccl_single_rank=rccl_single_rank)
    P0 = torch.from_numpy(fb.get_cov(0)).to(dev)
    torch.cuda.synchronize()
    sh.load_cov(P0)
    t = lambda v: torch.tensor(np.asarray(v, dtype=np.float64), device=dev)
    applied = []
    for k in range(1, steps):
        u, quat, ang, acc, dt, matches = inp[k]
        if u & orc.IMG:
            fb.frame(0, matches, quat)
        q, w, a, d = t(quat), t(ang), t(acc), t([dt])
        torch.cuda.synchronize()
        sh.predict(q, w, a, d)
        if u & orc.IMG:
            ap = sh.update(q)
            sh.sync()
            applied.append(int(ap.item()))
    P = sh.gather_cov()
    sh.sync()
    out = P.cpu().numpy(), fb.state_vector(0), applied, sh.dims()
    sh.close()
    fb.close()
    if blocks:
        ekf.set_block_sizes(512, 4096)
    return out


@pytest.mark.parametrize("rccl,blocks", [(False, None), (True, None), (True, (128, 128))])
def test_sharded_filter_single_rank(rccl, blocks):
    """slamk_sharded_* with one rank (the same products and stage calls as with more ranks) against the oracle; with
    `rccl` the library builds a one-rank RCCL communicator and its all-gathers really go through ncclAllGather."""
    torch = _torch()
    nf, steps = 45, 7
    st = synth.gen_stream(seed=17, nf=nf, steps=steps, churn=False)
    so, f = helpers.run_oracle(st)
    P, sv, applied, dims = _run_sharded(0, 1, None, nf, steps, torch.device("cuda", 0), rccl_single_rank=rccl, blocks=blocks)
    assert applied == [s["applied"] for s in so if s["applied"] >= 0]
    assert helpers.rel_err(P, f.P) < TOL and helpers.entry_err(P, f.P) < TOL
    assert np.array_equal(P, P.T)
    assert helpers.rel_err(sv, f.state_vector()) < TOL
    assert dims["n"] == 9 + 3 * nf and dims["rows_per_rank"] == dims["n_padded"]


def _sharded_worker(rank, world, port, nf, steps, out, blocks=None):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, sv, applied, dims = _run_sharded(rank, world, None, nf, steps, torch.device("cuda", 0), blocks=blocks)
        assert dims["rows_per_rank"] * world == dims["n_padded"]
        if rank == world - 1:
            np.save(out, np.concatenate([P.ravel(), sv]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nf,blocks", [(2, 45, None), (3, 45, None), (2, 90, (128, 128)), (3, 100, (128, 256))])
def test_sharded_filter_several_ranks_on_one_gpu(tmp_path, world, nf, blocks):
    """Two and three ranks of slamk_sharded_*, all on GPU 0, the library's all-gathers routed through its callback
    (gloo, staged through the host): row blocks with row0 > 0, padding rows in the last block, against the oracle.
    With 128-column panels the factorisation of S takes four panels, each dealt to the ranks by rows, all-gathered
    and factored on every rank, and the triangular solves take several block steps."""
    import socket
    import torch.multiprocessing as mp
    steps = 7 if nf < 60 else 5
    if blocks:
        os.environ["SLAMK_SPLITK_MIN_SLICE"] = "32"      # the products with a long k range run in slices at this size too
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    out = str(tmp_path / "res.npy")
    try:
        mp.spawn(_sharded_worker, args=(world, port, nf, steps, out, blocks), nprocs=world, join=True)
    finally:
        os.environ.pop("SLAMK_SPLITK_MIN_SLICE", None)
    res = np.load(out)
    st = synth.gen_stream(seed=17, nf=nf, steps=steps, churn=False)
    so, f = helpers.run_oracle(st)
    n = 9 + 3 * nf
    assert helpers.rel_err(res[:n * n].reshape(n, n), f.P) < TOL
    assert helpers.entry_err(res[:n * n].reshape(n, n), f.P) < TOL
    # rows of different ranks come from different products: the gathered covariance is still exactly symmetric
    assert np.array_equal(res[:n * n].reshape(n, n), res[:n * n].reshape(n, n).T)
    assert helpers.rel_err(res[n * n:], f.state_vector()) < TOL
