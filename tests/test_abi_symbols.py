"""CPU: the C-ABI library loads and exports every symbol include/slamk.h declares."""
import ctypes
import os
import re

from slam_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "slamk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(slamk_[a-z0-9_]+)\s*\(", txt, flags=re.I)))


def test_header_symbols_are_exported():
    syms = declared_symbols()
    assert len(syms) >= 25
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_python_prototypes_cover_the_header():
    assert sorted(_lib.SIGNATURES) == declared_symbols()


def test_default_params_match_reference_constants():
    p = _lib.Params()
    _lib.lib().slamk_default_params(ctypes.byref(p))
    assert (p.Q0, p.R0, p.qbias, p.p0, p.pinit) == (256, 32, 1e-5, 2, 1e-4)
    assert (p.alt_override, p.use_alt_override, p.gate, p.d_min, p.d_max, p.dinit) == (-1.34, 1, 4, 0.1, 10e3, 100)


def test_flop_model_is_close_to_dense_formula():
    # executed MFMA flops (tile skipping, triangular trimming, padding to 16) against the textbook counts of
    # what runs on MFMA: Cholesky, the two triangular solves and the Joseph products.  The products with the
    # sparse Jacobians F and H are not MFMA work.
    L = _lib.lib()
    assert L.slamk_flops_predict(105) == 0.0
    n, m = 105, 161                                   # fused small-filter path
    model = m**3 / 3 + 2 * n * m * m + 3 * n**3 + n * n * m
    ex = L.slamk_flops_update(n, m, 0)
    assert 0.8 * model < ex < 1.6 * model
    assert ex == L.slamk_flops_update_stage(n, m, 0, 0) + L.slamk_flops_update_stage(n, m, 0, 1)
    assert L.slamk_flops_update_stage(n, m, 1, 1) < L.slamk_flops_update_stage(n, m, 0, 1)
    n, m = 6009, 10001                                # blocked large-map path (explicit triangular inverse)
    model = 2 * m**3 / 3 + 2 * n * m * m + 3 * n**3 + n * n * m
    ex = L.slamk_flops_update(n, m, 0)
    assert 0.8 * model < ex < 1.3 * model


def test_product_never_touches_the_oracle():
    """The CPU restatement is test infrastructure: nothing under slam_amd/ or host/ may import, link or name it."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [f for pat in ("slam_amd/*.py", "slam_amd/csrc/*", "host/*.cpp", "host/include/*", "host/Makefile", "include/*.h")
             for f in glob.glob(os.path.join(root, pat)) if os.path.isfile(f)]
    assert len(files) > 10
    for f in files:
        text = open(f, errors="ignore").read()
        assert "oracle" not in text.lower() and "orc_" not in text, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No CPU fallback: without libslamk.so every entry through the Python face raises."""
    import pytest
    from slam_amd import ekf
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libslamk.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ekf.FilterBatch(max_features=4, batch=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ekf.flops_update(21, 17)
