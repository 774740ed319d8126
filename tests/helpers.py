"""Shared drivers: run one synthetic stream through the CPU oracle and through the HIP library,
following the reference loop order (src/slam/main.cpp:131-144 start-up, :150-236 loop)."""
import numpy as np

from oracle import oracle as orc
from slam_amd import synth


def frame_matches(stream, k, rng):
    s = stream.steps[k]
    src, refl = stream.noisy_frame(k, rng)
    return [(int(s.ids[i]), src[i, 0], src[i, 1], refl[i, 0], refl[i, 1]) for i in range(len(s.ids))]


def stream_inputs(stream, noise_seed):
    """Per-step (flags, quat, ang, acc, dt, matches) exactly as main() would see them."""
    rng = np.random.Generator(np.random.PCG64(noise_seed))
    out = []
    for k, s in enumerate(stream.steps):
        u = orc.ACC | orc.ANG | orc.QUAT | (orc.IMG if s.image else 0)
        matches = frame_matches(stream, k, rng) if s.image else []
        dt = 0.0 if k == 0 else s.t - stream.steps[k - 1].t
        out.append((u, s.quat, s.ang, s.acc, dt, matches))
    return out


def run_oracle(stream, noise_seed=7, params=None, init_state=True, inp=None):
    """Returns list of snapshots after every loop iteration."""
    if inp is None:
        inp = stream_inputs(stream, noise_seed)
    f = orc.Filter(params)
    u, quat, ang, acc, dt, matches = inp[0]
    if init_state and stream.init_pos is not None:
        f.X = stream.init_pos
    f.update_features(matches, quat)          # main.cpp:142-143
    f.old_pos[:] = f.X                        # :144
    snaps = []
    for k in range(1, len(inp)):
        u, quat, ang, acc, dt, matches = inp[k]
        s = orc.make_sensors(quat=quat, acc=acc, ang=ang)
        rc, xy, applied = f.step(u, s, dt, matches)
        assert rc == 0, rc
        snaps.append({"X": f.X, "V": f.V, "pib": f.pib(), "ids": f.ids(), "P": f.P, "xy": xy, "applied": applied})
    return snaps, f


def run_gpu(stream, fb, noise_seed=7, b=0, init_state=True, inp=None):
    """Same loop through the C ABI (single filter b of a FilterBatch)."""
    if inp is None:
        inp = stream_inputs(stream, noise_seed)
    B = fb.B
    u, quat, ang, acc, dt, matches = inp[0]
    if init_state and stream.init_pos is not None:
        fb.set_state(b, X=stream.init_pos)
    fb.frame(b, matches, quat)
    snaps = []
    for k in range(1, len(inp)):
        u, quat, ang, acc, dt, matches = inp[k]
        if u & orc.IMG:
            fb.frame(b, matches, quat)
        fb.predict(np.tile(quat, B), np.tile(ang, B), np.tile(acc, B), np.full(B, dt))
        X, _, _ = fb.get_state(b)
        applied = -1
        if u & orc.IMG:
            applied = int(fb.update(np.tile(quat, B))[b])
        Xp, V, _ = fb.get_state(b)
        ft = fb.get_features(b)
        snaps.append({"X": Xp, "V": V, "pib": ft["pib"], "ids": list(ft["ids"]), "P": fb.get_cov(b),
                      "xy": (X[1], X[0]), "applied": applied})
    return snaps


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = np.linalg.norm(a - b)
    return d / max(np.linalg.norm(b), 1e-300)


def entry_err(a, b):
    """Worst entry-wise deviation of covariance a from b, each entry scaled by sqrt(b_ii b_jj) (the size an entry
    of a covariance can have): a norm-wise bound alone would let small cross terms be wrong."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = np.sqrt(np.abs(np.diag(b)))
    return float(np.max(np.abs(a - b) / np.maximum(np.outer(d, d), 1e-300)))


class FixtureStream:
    """Inputs and expected outputs of a committed end-to-end fixture (tests/golden/*.npz, written by
    oracle/gen_stream_goldens.py): `inp` in the form of stream_inputs(), `init_pos`, and the arrays in `g`."""

    def __init__(self, path):
        g = dict(np.load(path, allow_pickle=False))
        self.g, self.init_pos = g, g["init_pos"]
        off = g["in_match_off"]
        self.inp = []
        for k in range(len(g["in_flags"])):
            m = [(int(g["in_match_id"][j]), *[float(x) for x in g["in_match_val"][j]]) for j in range(off[k], off[k + 1])]
            self.inp.append((int(g["in_flags"][k]), g["in_quat"][k], g["in_ang"][k], g["in_acc"][k], float(g["in_dt"][k]), m))

    def expected(self):
        """Snapshots in the form run_oracle / run_gpu return; "P" is None where the fixture keeps only its diagonal."""
        g = self.g
        out, o = [], 0
        full = {int(k): i for i, k in enumerate(g["P_steps"])} if "P_steps" in g else {}
        po, pofs = 0, {}
        for k in sorted(full):
            n = 9 + 3 * int(g["nf"][k])
            pofs[k] = (po, n)
            po += n * n
        d = 0
        for k in range(len(g["X"])):
            nf = int(g["nf"][k])
            n = 9 + 3 * nf
            P = None
            if k in pofs:
                a, nn = pofs[k]
                P = g["P_full"][a:a + nn * nn].reshape(nn, nn)
            out.append({"X": g["X"][k], "V": g["V"][k], "pib": g["pib"][o:o + nf], "ids": [int(i) for i in g["ids"][o:o + nf]],
                        "P": P, "diagP": g["diagP"][d:d + n], "traceP": float(g["traceP"][k]), "frobP": float(g["frobP"][k]),
                        "xy": (float(g["xy"][k][0]), float(g["xy"][k][1])), "applied": int(g["applied"][k])})
            o += nf
            d += n
        return out


def compare_with_fixture(exp, got, tol):
    """ids and gate exact; X, V, pib, P (full where kept, else diagonal / trace / Frobenius norm) within tol."""
    assert len(exp) == len(got)
    worst = 0.0
    for k, (a, b) in enumerate(zip(exp, got)):
        assert a["ids"] == [int(i) for i in b["ids"]], "association differs at step %d" % k
        assert a["applied"] == b["applied"], "gate differs at step %d" % k
        errs = [rel_err(b["X"], a["X"]), rel_err(b["V"], a["V"]), rel_err(b["pib"], a["pib"]),
                rel_err(np.diag(b["P"]), a["diagP"]), abs(np.trace(b["P"]) - a["traceP"]) / abs(a["traceP"]),
                abs(np.linalg.norm(b["P"]) - a["frobP"]) / a["frobP"]]
        if a["P"] is not None:
            errs += [rel_err(b["P"], a["P"]), entry_err(b["P"], a["P"])]
        assert max(errs) < tol, "step %d: %s" % (k, errs)
        assert "%0.5f,%0.5f" % a["xy"] == "%0.5f,%0.5f" % tuple(b["xy"])
        worst = max(worst, max(errs))
    return worst
