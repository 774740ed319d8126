"""GPU: the batched pre-processing filters (slamk_pre_*) against the reference-authored fixtures and the oracle.

* euler2qbw: tests/golden/euler2qbw_{euler,mquat}.txt = head of src/euler2qbw/{euler,Mquat}.txt  (pinned)
* rotate   : tests/golden/rotate_acc{,_rot}.txt      = head of src/rotate/testing/{acc,acc.rot}, rotate -e 0 0 p  (pinned)
* img2body : against the oracle only (the reference's fixture was made with other camera constants than the source holds)
* fir      : against the oracle's ring-buffer restatement of src/lpf/fir.cpp (parity unpinned: no fixture in the reference)
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as orc
from slam_amd import _lib

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _dev(a):
    import torch
    return torch.tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")


def test_euler2qbw_kernel_known_answers():
    import torch
    eu = np.loadtxt(os.path.join(G, "euler2qbw_euler.txt"), delimiter=",")
    mq = np.loadtxt(os.path.join(G, "euler2qbw_mquat.txt"), delimiter=",")
    d_eu, d_q = _dev(eu), torch.zeros(len(eu), 4, dtype=torch.float64, device="cuda")
    assert _lib.lib().slamk_pre_euler2qbw_dev(d_eu.data_ptr(), d_q.data_ptr(), len(eu), None) == 0
    torch.cuda.synchronize()
    got = d_q.cpu().numpy()
    assert np.max(np.abs(got - mq)) < 5e-9                      # fixture printed with 9 significant digits
    ref = np.array([orc.euler2quaternion(*e) for e in eu])
    assert np.max(np.abs(got - ref)) < 1e-15


def test_rotate_kernel_known_answers_yaw_pi():
    import torch
    acc = np.loadtxt(os.path.join(G, "rotate_acc.txt"), delimiter=",")
    rot = np.loadtxt(os.path.join(G, "rotate_acc_rot.txt"), delimiter=",")
    q = orc.euler2quaternion(0, 0, np.pi)
    d_v, d_o = _dev(acc), torch.zeros(len(acc), 3, dtype=torch.float64, device="cuda")
    qa = (C.c_double * 4)(*q)
    assert _lib.lib().slamk_pre_rotate_dev(qa, d_v.data_ptr(), d_o.data_ptr(), len(acc), None) == 0
    torch.cuda.synchronize()
    got = d_o.cpu().numpy()
    assert np.max(np.abs(got - rot)) < 5e-7                     # %f output, 6 decimals
    assert np.max(np.abs(got - acc @ orc.rotation(q).T)) < 1e-15
    # never normalised: scales with |q|^2 (Quaternion.cpp:37-45)
    qa2 = (C.c_double * 4)(*(1.5 * np.asarray(q)))
    _lib.lib().slamk_pre_rotate_dev(qa2, d_v.data_ptr(), d_o.data_ptr(), len(acc), None)
    torch.cuda.synchronize()
    assert np.allclose(d_o.cpu().numpy(), 2.25 * got, rtol=1e-14, atol=0)


def test_img2body_kernel_against_oracle():
    import torch
    rng = np.random.default_rng(3)
    px = np.column_stack([rng.uniform(0, 1536, 500), rng.uniform(0, 1180, 500)])
    d_p, d_o = _dev(px), torch.zeros(500, 2, dtype=torch.float64, device="cuda")
    assert _lib.lib().slamk_pre_img2body_dev(d_p.data_ptr(), d_o.data_ptr(), 500, None) == 0
    torch.cuda.synchronize()
    ref = np.array([orc.image2body(*p) for p in px])
    got = d_o.cpu().numpy()
    assert np.max(np.abs(got - ref) / np.maximum(1.0, np.abs(ref))) < 1e-13


@pytest.mark.parametrize("order,n", [(1, 5), (8, 3), (31, 1000)])
def test_fir_kernel_against_oracle(order, n):
    import torch
    rng = np.random.default_rng(order)
    c = rng.standard_normal(order)
    x = rng.standard_normal((n, 3))
    d_c, d_x, d_y = _dev(c), _dev(x), torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    assert _lib.lib().slamk_pre_fir_dev(d_c.data_ptr(), order, d_x.data_ptr(), d_y.data_ptr(), n, None) == 0
    torch.cuda.synchronize()
    ref = np.column_stack([orc.fir(c, x[:, k].copy()) for k in range(3)])
    assert np.array_equal(d_y.cpu().numpy(), ref)               # same products in the same order


def test_preprocess_argument_checks():
    L = _lib.lib()
    assert L.slamk_pre_euler2qbw_dev(None, None, 0, None) == 0
    assert L.slamk_pre_euler2qbw_dev(None, None, 4, None) < 0
    assert L.slamk_pre_fir_dev(None, 0, None, None, 0, None) < 0
