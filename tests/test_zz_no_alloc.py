"""GPU: the entry point that allocates nothing (slamk_stage_gain_static_dev) against slamk_stage_gain_dev.

Same kernels in the same order, so the gains must agree bit for bit.  Two calls in a row, the second with a
smaller input: the static workspace keeps the first call's contents and must not leak into the second.
"""
import numpy as np
import pytest

from slam_amd import ekf, _lib

pytestmark = pytest.mark.gpu


def _problem(torch, rows, m, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, m + 8))
    S = A @ A.T + m * np.eye(m)
    W = rng.standard_normal((rows, m))
    r = rng.uniform(0.5, 2.0, m)
    return S, W, r


def _gain(torch, L, fb, S, W, r, static):
    m, rows = S.shape[0], W.shape[0]
    dev = "cuda"
    dS = torch.from_numpy(np.tril(S)).to(dev)
    dL = torch.zeros((m, m), dtype=torch.float64, device=dev)
    dLi = torch.zeros((m, m), dtype=torch.float64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    assert L.slamk_cholesky_inverse_dev(dS.data_ptr(), dL.data_ptr(), dLi.data_ptr(), m, m, status.data_ptr(), None) == 0
    dW = torch.from_numpy(W.copy()).to(dev)
    dK = torch.full((rows, m), np.nan, dtype=torch.float64, device=dev)
    dKR = torch.full((rows, m), np.nan, dtype=torch.float64, device=dev)
    dr = torch.from_numpy(r).to(dev)
    if static:
        rc = L.slamk_stage_gain_static_dev(fb.h, dW.data_ptr(), rows, m, m, dL.data_ptr(), dLi.data_ptr(), m,
                                           dK.data_ptr(), dKR.data_ptr(), dr.data_ptr(), None)
    else:
        dY = torch.zeros((rows, m), dtype=torch.float64, device=dev)
        rc = L.slamk_stage_gain_dev(fb.h, dW.data_ptr(), rows, m, m, dL.data_ptr(), dLi.data_ptr(), m, dY.data_ptr(),
                                    dK.data_ptr(), dKR.data_ptr(), dr.data_ptr(), None)
    assert rc == 0, fb.L.slamk_last_error(fb.h)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    return dK.cpu().numpy(), dKR.cpu().numpy()


def test_static_workspace_gain_matches_the_allocating_entry_point():
    import torch
    assert torch.cuda.is_available()
    L = _lib.lib()
    fb = ekf.FilterBatch(max_features=4, batch=1)
    # the second problem is smaller in both dimensions; 4352 columns cross the 4096-row inverse blocks
    for rows, m, seed in ((208, 4352, 1), (112, 176, 2)):
        S, W, r = _problem(torch, rows, m, seed)
        K0, KR0 = _gain(torch, L, fb, S, W, r, static=False)
        K1, KR1 = _gain(torch, L, fb, S, W, r, static=True)
        assert np.all(np.isfinite(K0))
        assert np.array_equal(K0, K1) and np.array_equal(KR0, KR1), (rows, m)
        ref = np.linalg.solve(S, W.T).T
        assert np.linalg.norm(K1 - ref) / np.linalg.norm(ref) < 1e-11
        assert np.array_equal(KR1, K1 * r[None, :])
    # beyond the static workspace: refused with a clear error, nothing launched
    rows, m = 4096, 2048
    z = torch.zeros(16, dtype=torch.float64, device="cuda")
    rc = L.slamk_stage_gain_static_dev(fb.h, z.data_ptr(), rows, m, m, z.data_ptr(), z.data_ptr(), m, z.data_ptr(),
                                       z.data_ptr(), z.data_ptr(), None)
    assert rc == _lib.SLAMK_ECAPACITY
    assert b"static workspace" in L.slamk_last_error(fb.h)
    fb.close()
