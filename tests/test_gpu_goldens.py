"""GPU: the HIP path through the C ABI against the COMMITTED fixtures under tests/golden/ (written by
oracle/gen_stream_goldens.py), and the full-size configurations of the benchmark list:

  * streams with feature churn at 5, 20 and 32 features, both covariance update forms;
  * the 2000-feature map (n = 6009, m = 10001) against sampled covariance entries, diagonal, trace and norm;
  * the per-GPU shard of the 65536-filter batch (8192 filters x 32 features): sampled filters against the oracle
    and symmetry / positive definiteness of the rest.

Bar: ids and gate bit-exact; X, V, pib, P within 1e-9 (norm-wise and entry-wise scaled by sqrt(P_ii P_jj)).
"""
import os

import numpy as np
import pytest

from slam_amd import ekf, synth
import helpers

pytestmark = pytest.mark.gpu
TOL = 1e-9
G = os.path.join(os.path.dirname(__file__), "golden")


class _S:
    def __init__(self, init_pos):
        self.init_pos = init_pos


@pytest.mark.parametrize("nf,form", [(5, 0), (20, 0), (20, 1), (32, 0), (32, 1), (5, 2), (20, 2), (32, 2)])
def test_stream_fixtures(nf, form):
    fx = helpers.FixtureStream(os.path.join(G, "stream_nf%d.npz" % nf))
    p = ekf.default_params()
    p.update_form = form
    fb = ekf.FilterBatch(max_features=nf + 4, batch=1, params=p)
    got = helpers.run_gpu(_S(fx.init_pos), fb, inp=fx.inp)
    worst = helpers.compare_with_fixture(fx.expected(), got, TOL)
    print("nf=%d form=%d worst deviation from the fixture %.2e" % (nf, form, worst))
    fb.close()


def test_large_map_2000_features_against_golden():
    """configs[3]: one filter, 2000 features.  Three iterations with an image through the blocked Cholesky, the
    multi-block triangular solves (m > 4096) and the GEMM Joseph update, against the golden values."""
    g = dict(np.load(os.path.join(G, "large_map_nf2000.npz"), allow_pickle=False))
    fx = helpers.FixtureStream(os.path.join(G, "large_map_nf2000.npz"))
    nf = 2000
    fb = ekf.FilterBatch(max_features=nf, batch=1)
    fb.set_state(0, X=fx.init_pos)
    u, quat, ang, acc, dt, matches = fx.inp[0]
    fb.frame(0, matches, quat)
    assert fb.num_features(0) == nf and fb.num_rows(0) == 10001
    si, sj = g["sample_i"], g["sample_j"]
    worst = 0.0
    for k in range(1, len(fx.inp)):
        u, quat, ang, acc, dt, matches = fx.inp[k]
        fb.frame(0, matches, quat)
        fb.predict(quat, ang, acc, [dt])
        applied = int(fb.update(quat)[0])            # raises on SLAMK_ENOTSPD / SLAMK_EREFL
        assert applied == int(g["applied"][k - 1])
        X, V, _ = fb.get_state(0)
        ft = fb.get_features(0)
        assert np.array_equal(ft["ids"], g["ids"])
        P = fb.get_cov(0)
        assert np.all(np.isfinite(P)) and np.array_equal(P, P.T)
        d = g["diagP"][k - 1]
        errs = [helpers.rel_err(X, g["X"][k - 1]), helpers.rel_err(V, g["V"][k - 1]),
                helpers.rel_err(ft["pib"], g["pib"][k - 1]), helpers.rel_err(np.diag(P), d),
                float(np.max(np.abs(np.diag(P) - d) / d)),
                abs(np.trace(P) - g["traceP"][k - 1]) / g["traceP"][k - 1],
                abs(np.linalg.norm(P) - g["frobP"][k - 1]) / g["frobP"][k - 1],
                float(np.max(np.abs(P[si, sj] - g["sampleP"][k - 1]) / np.sqrt(d[si] * d[sj])))]
        assert max(errs) < TOL, (k, errs)
        worst = max(worst, max(errs))
    # positive definite: the Cholesky factorisation of the result exists
    np.linalg.cholesky(P)
    print("2000 features: worst deviation from the golden values %.2e" % worst)
    fb.close()


def test_large_map_2000_features_row_block_sharded_against_golden():
    """configs[3] through slamk_sharded_* at full size (one rank, its all-gathers made on a one-rank RCCL
    communicator): the left-looking panel factorisation with twenty panels, the products whose k range is cut into
    slices at their real sizes, the row-block Joseph update; same golden values as the unsharded path."""
    import torch
    from slam_amd.sharded import ShardedFilter
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    g = dict(np.load(os.path.join(G, "large_map_nf2000.npz"), allow_pickle=False))
    fx = helpers.FixtureStream(os.path.join(G, "large_map_nf2000.npz"))
    nf = 2000
    fb = ekf.FilterBatch(max_features=nf, batch=1)
    fb.set_state(0, X=fx.init_pos)
    u, quat, ang, acc, dt, matches = fx.inp[0]
    fb.frame(0, matches, quat)
    stream = torch.cuda.Stream(device=dev)
    sh = ShardedFilter(fb, 0, 1, None, device=dev, stream=stream, rccl_single_rank=True)
    P0 = torch.from_numpy(fb.get_cov(0)).to(dev)
    torch.cuda.synchronize()
    sh.load_cov(P0)
    t = lambda v: torch.tensor(np.asarray(v, dtype=np.float64), device=dev)
    si, sj = g["sample_i"], g["sample_j"]
    worst = 0.0
    for k in range(1, len(fx.inp)):
        u, quat, ang, acc, dt, matches = fx.inp[k]
        fb.frame(0, matches, quat)
        q, w, a, d_ = t(quat), t(ang), t(acc), t([dt])
        torch.cuda.synchronize()
        sh.predict(q, w, a, d_)
        ap = sh.update(q)
        sh.sync()
        assert int(ap.item()) == int(g["applied"][k - 1])
        P = sh.gather_cov()
        sh.sync()
        P = P.cpu().numpy()
        X, V, _ = fb.get_state(0)
        ft = fb.get_features(0)
        d = g["diagP"][k - 1]
        errs = [helpers.rel_err(X, g["X"][k - 1]), helpers.rel_err(V, g["V"][k - 1]),
                helpers.rel_err(ft["pib"], g["pib"][k - 1]), float(np.max(np.abs(np.diag(P) - d) / d)),
                abs(np.trace(P) - g["traceP"][k - 1]) / g["traceP"][k - 1],
                abs(np.linalg.norm(P) - g["frobP"][k - 1]) / g["frobP"][k - 1],
                float(np.max(np.abs(P[si, sj] - g["sampleP"][k - 1]) / np.sqrt(d[si] * d[sj]))),
                float(np.max(np.abs(P - P.T)) / np.max(d))]
        assert np.all(np.isfinite(P)) and max(errs) < TOL, (k, errs)
        assert np.array_equal(P, P.T)
        worst = max(worst, max(errs))
    print("2000 features, row-block sharded: worst deviation from the golden values %.2e" % worst)
    sh.close()
    fb.close()


def test_shard_of_the_65536_filter_batch():
    """configs[4] on one of its eight GPUs: 8192 filters x 32 features, each with its own pixel noise."""
    from oracle import oracle as orc
    import torch
    B, nf, steps = 8192, 32, 4
    st = synth.gen_stream(seed=1, nf=nf, steps=steps, churn=False, image_every=1)
    fb = ekf.FilterBatch(max_features=nf, batch=B)
    sample = [0, 4095, 4096, 8191]
    base = helpers.stream_inputs(st, 999)
    inps = {b: helpers.stream_inputs(st, 1000 + b) for b in sample}
    for b in sample:
        inps[b][0] = base[0]
    so = {b: helpers.run_oracle(st, inp=inps[b])[0] for b in sample}
    for b in range(B):
        fb.set_state(b, X=st.init_pos)
    fb.frame(-1, base[0][5], base[0][1])
    dev = torch.device("cuda", 0)
    applied = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(device=dev)
    rng = np.random.Generator(np.random.PCG64(5))
    for k in range(1, steps):
        u, quat, ang, acc, dt, matches = base[k]
        s = st.steps[k]
        src = np.tile(s.src.T[None], (B, 1, 1)) + st.noise * rng.standard_normal((B, 2, nf))
        has = (s.refl.T[None] != 0)
        refl = np.tile(s.refl.T[None], (B, 1, 1)) + st.noise * rng.standard_normal((B, 2, nf)) * has
        for b in sample:                              # the sampled filters get the oracle runs' observations
            mb = inps[b][k][5]
            src[b] = np.array([[m[1] for m in mb], [m[2] for m in mb]])
            refl[b] = np.array([[m[3] for m in mb], [m[4] for m in mb]])
        one = lambda v: torch.tensor(np.tile(np.asarray(v, dtype=np.float64), (B, 1)), device=dev)
        d = {"q": one(quat), "w": one(ang), "a": one(acc), "dt": torch.full((B,), dt, device=dev, dtype=torch.float64),
             "src": torch.tensor(np.ascontiguousarray(src), device=dev),
             "refl": torch.tensor(np.ascontiguousarray(refl), device=dev)}
        torch.cuda.synchronize()
        fb.step_dev(d["q"].data_ptr(), d["w"].data_ptr(), d["a"].data_ptr(), d["dt"].data_ptr(), None,
                    d["src"].data_ptr(), d["refl"].data_ptr(), applied.data_ptr(), stream.cuda_stream)
        fb.sync(stream.cuda_stream)
        assert int(applied.sum().item()) == B
        for b in sample:
            a = so[b][k - 1]
            assert helpers.rel_err(fb.get_cov(b), a["P"]) < TOL and helpers.entry_err(fb.get_cov(b), a["P"]) < TOL
            assert helpers.rel_err(fb.state_vector(b)[:-3], np.concatenate([a["X"], a["V"], a["pib"].ravel()])) < TOL
    for b in range(7, B, 911):
        P = fb.get_cov(b)
        assert np.all(np.isfinite(P)) and np.array_equal(P, P.T) and np.all(np.diag(P) > 0)
        assert np.linalg.eigvalsh(P).min() > 0
    fb.close()
