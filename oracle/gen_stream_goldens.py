#!/usr/bin/env python3
"""Writes the committed end-to-end fixtures tests/golden/stream_nf{5,20,32}.npz and
tests/golden/large_map_nf2000.npz from the CPU oracle (TEST INFRASTRUCTURE, run in the build container).

Stream fixtures: a synthetic stream with feature churn (replacement in place, deletion, append) run
through the C oracle's loop (oracle/ekf_oracle.c orc_step, which restates src/slam/main.cpp:157-235).
Stored per loop iteration: the inputs exactly as main() sees them (flags, quat, ang, acc, dt, the
frame's matches) and the outputs X, V, ids, pib, gate result, stdout pair, diag(P), trace(P), |P|_F;
the full P every sixth iteration and at the end.  All float64, bit-exact (npz, no pickles).

Large-map fixture: one filter with 2000 features (n = 6009, m = 10001), seed 2, every feature visible in
every frame, three iterations with an image, through oracle/numpy_oracle.py (per-feature parts from the C
oracle, dense algebra in the reference's LU / simple-form formulation by LAPACK).  Stored: inputs, X, V,
pib, diag(P), trace, Frobenius norm, and 4096 sampled entries of P with their indices.

    python oracle/gen_stream_goldens.py [streams] [large]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != os.path.dirname(os.path.abspath(__file__))]
sys.path.insert(0, ROOT)                    # `oracle` must resolve to the directory, not to oracle/oracle.py
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as orc            # noqa: E402
from oracle import numpy_oracle as npo      # noqa: E402
from slam_amd import synth                  # noqa: E402
import helpers                              # noqa: E402

G = os.path.join(ROOT, "tests", "golden")

STREAMS = {5: dict(seed=15, nf=5, steps=61, churn=True, drop_at=21, drop_count=3, add_at=42, add_count=3),
           20: dict(seed=30, nf=20, steps=61, churn=True, drop_at=21, drop_count=3, add_at=42, add_count=3),
           32: dict(seed=42, nf=32, steps=61, churn=True, drop_at=21, drop_count=3, add_at=42, add_count=3)}


def pack_inputs(inp):
    """Per-step inputs as flat arrays: matches of all steps concatenated, with offsets."""
    flags = np.array([u for (u, *_r) in inp], dtype=np.int32)
    quat = np.array([q for (_u, q, *_r) in inp])
    ang = np.array([a for (_u, _q, a, *_r) in inp])
    acc = np.array([a for (_u, _q, _w, a, *_r) in inp])
    dt = np.array([d for (*_r, d, _m) in inp])
    off, mid, mval = [0], [], []
    for (*_r, matches) in inp:
        for (i, sx, sy, rx, ry) in matches:
            mid.append(i)
            mval.append([sx, sy, rx, ry])
        off.append(len(mid))
    return {"in_flags": flags, "in_quat": quat, "in_ang": ang, "in_acc": acc, "in_dt": dt,
            "in_match_off": np.array(off, dtype=np.int64), "in_match_id": np.array(mid, dtype=np.int32),
            "in_match_val": np.array(mval, dtype=np.float64).reshape(-1, 4)}


def gen_stream(nf):
    st = synth.gen_stream(**STREAMS[nf])
    inp = helpers.stream_inputs(st, noise_seed=7)
    snaps, _ = helpers.run_oracle(st, inp=inp)
    out = pack_inputs(inp)
    out["init_pos"] = st.init_pos
    T = len(snaps)
    out["X"] = np.array([s["X"] for s in snaps])
    out["V"] = np.array([s["V"] for s in snaps])
    out["applied"] = np.array([s["applied"] for s in snaps], dtype=np.int32)
    out["xy"] = np.array([s["xy"] for s in snaps])
    out["nf"] = np.array([len(s["ids"]) for s in snaps], dtype=np.int32)
    out["ids"] = np.concatenate([np.array(s["ids"], dtype=np.int32) for s in snaps])
    out["pib"] = np.concatenate([s["pib"].reshape(-1, 3) for s in snaps])
    out["diagP"] = np.concatenate([np.diag(s["P"]) for s in snaps])
    out["traceP"] = np.array([np.trace(s["P"]) for s in snaps])
    out["frobP"] = np.array([np.linalg.norm(s["P"]) for s in snaps])
    full = [k for k in range(T) if k % 6 == 5 or k == T - 1]
    out["P_steps"] = np.array(full, dtype=np.int32)
    out["P_full"] = np.concatenate([snaps[k]["P"].ravel() for k in full])
    path = os.path.join(G, "stream_nf%d.npz" % nf)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", T, "iterations; feature counts", sorted(set(out["nf"])))


def gen_large(nf=2000, steps=4):
    st = synth.gen_stream(seed=2, nf=nf, steps=steps, churn=False, image_every=1, ref_fraction=0.5)
    inp = helpers.stream_inputs(st, noise_seed=7)
    f = orc.Filter()
    f.X = st.init_pos
    t0 = time.time()
    f.update_features(inp[0][5], inp[0][1])
    f.old_pos[:] = f.X
    print("first frame: %.1f s, n = %d" % (time.time() - t0, f.n), flush=True)
    n = f.n
    rng = np.random.Generator(np.random.PCG64(2000))
    si = rng.integers(0, n, 4096)
    sj = rng.integers(0, n, 4096)
    si[:6] = [0, 3, 5, 2, n - 1, n - 3]          # a few fixed ones: pose block, pose/bias cross terms
    sj[:6] = [1, 4, 3, n - 2, 0, 4]
    out = pack_inputs(inp)
    out["init_pos"] = st.init_pos
    out["sample_i"], out["sample_j"] = si.astype(np.int32), sj.astype(np.int32)
    Xs, Vs, pibs, diags, trs, frs, samples, aps, ms = [], [], [], [], [], [], [], [], []
    for k in range(1, len(inp)):
        u, quat, ang, acc, dt, matches = inp[k]
        s = orc.make_sensors(quat=quat, acc=acc, ang=ang)
        t0 = time.time()
        xy, ap = npo.step(f, u, s, dt, matches)
        P = f.P
        print("iteration %d: %.1f s, applied %d, asym %.1e, min diag %.3e" %
              (k, time.time() - t0, ap, np.abs(P - P.T).max(), np.diag(P).min()), flush=True)
        Xs.append(f.X); Vs.append(f.V); pibs.append(f.pib()); diags.append(np.diag(P).copy())
        trs.append(np.trace(P)); frs.append(np.linalg.norm(P)); samples.append(P[si, sj].copy()); aps.append(ap)
    out.update(X=np.array(Xs), V=np.array(Vs), pib=np.array(pibs), diagP=np.array(diags), traceP=np.array(trs),
               frobP=np.array(frs), sampleP=np.array(samples), applied=np.array(aps, dtype=np.int32),
               ids=np.array(f.ids(), dtype=np.int32))
    path = os.path.join(G, "large_map_nf%d.npz" % nf)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    what = sys.argv[1:] or ["streams", "large"]
    if "streams" in what:
        for nf_ in (5, 20, 32):
            gen_stream(nf_)
    if "large" in what:
        gen_large()
