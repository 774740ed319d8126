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


def run_gpu(stream, fb, noise_seed=7, b=0, init_state=True):
    """Same loop through the C ABI (single filter b of a FilterBatch)."""
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
