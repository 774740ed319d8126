"""Seeded synthetic sensor/feature streams in the formats the estimator reads.

Sensor stream  : lines `time,TYPE,v1,v2,...` (TYPE in ACC ANG QUAT IMG ALT POS VEL INIT),
                 lines closer than 5 ms to the first line of a group form one time step
                 (reference: src/ARC/Sensors.cpp:11-67, include/Sensors.hpp:8).
Feature file   : `id,x,y,rx,ry` per feature, blank line ends a frame, rx=ry=0 means the
                 feature has no reflection (reference: src/ARC/imagesensor.cpp:35-45).

Scenario: a vehicle moves along body-x over a mirror plane z = 0 at height X_z = 1.34 and
sees point features ahead of it plus, for a fraction of them, their mirror images.  Geometry
follows the measurement model of the estimator (reference: src/ARC/feature.cpp:143-198):
a feature at world position f is seen at (y/x, z/x) of Rw2b (f - X), its reflection at
(y/x, z/x) of Rw2b (S f - X) with S = diag(1, 1, -1).

Constraints that keep the filter well-posed and the reference's quirks out of the way:
  * the first time stamp is > 0 (a sensor's dt is forced to 0 while its last stamp == 0,
    reference: include/Sensor.hpp:41-46);
  * every group carries ACC (dt is taken from it, reference: src/slam/main.cpp:158-166);
  * reflection presence is constant per feature id and reflection x is never 0
    (reference: src/ARC/view.cpp:17,39; src/slam/main.cpp:211-218,353);
  * features are retired (dropped from the frame) before they come closer than `retire_dx`.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


def quat_to_rot(q):
    """Scalar-last quaternion -> body-to-world rotation (reference: src/ARC/Quaternion.cpp:32-47)."""
    q1, q2, q3, q4 = q
    return np.array([
        [q4 * q4 + q1 * q1 - q2 * q2 - q3 * q3, 2 * (q1 * q2 - q4 * q3), 2 * (q4 * q2 + q1 * q3)],
        [2 * (q1 * q2 + q4 * q3), q4 * q4 - q1 * q1 + q2 * q2 - q3 * q3, 2 * (q2 * q3 - q4 * q1)],
        [2 * (q1 * q3 - q4 * q2), 2 * (q4 * q1 + q2 * q3), q4 * q4 - q1 * q1 - q2 * q2 + q3 * q3],
    ])


def yaw_pitch_quat(yaw, pitch):
    """Scalar-last quaternion of R = Rz(yaw) Ry(pitch)."""
    cy, sy, cp, sp = np.cos(yaw / 2), np.sin(yaw / 2), np.cos(pitch / 2), np.sin(pitch / 2)
    # q = qz * qy, (x, y, z, w)
    return np.array([-sy * sp, cy * sp, sy * cp, cy * cp])


@dataclass
class Step:
    t: float
    quat: np.ndarray
    ang: np.ndarray
    acc: np.ndarray
    image: bool
    ids: Optional[np.ndarray] = None        # int32 [k]
    src: Optional[np.ndarray] = None        # [k,2] noise-free
    refl: Optional[np.ndarray] = None       # [k,2] noise-free, (0,0) = none
    X_true: Optional[np.ndarray] = None


@dataclass
class Stream:
    seed: int
    dt: float
    steps: List[Step] = field(default_factory=list)
    init_pos: np.ndarray = None
    noise: float = 1e-3

    def noisy_frame(self, k, rng):
        """Measurement of frame k with pixel noise from `rng` (numpy Generator)."""
        s = self.steps[k]
        src = s.src + self.noise * rng.standard_normal(s.src.shape)
        has = (s.refl[:, 0] != 0) | (s.refl[:, 1] != 0)
        refl = s.refl + self.noise * rng.standard_normal(s.refl.shape) * has[:, None]
        return src, refl

    def to_text(self, noise_seed=None, pos_every=0, vel_every=0, alt=False, no_acc_every=0, no_acc_ang_every=0,
                alt_only_after=None):
        """Return (sensors_text, features_text) for the `slam <features> < sensors` front end.

        pos_every / vel_every : every k-th step also carries a POS line (x and y swapped on the wire, as the
                    estimator swaps them back, reference: src/slam/main.cpp:180-185) / a VEL line (world-frame
                    velocity, turned into the body frame by the estimator, :186-191), both slightly off the truth.
        alt       : every step carries an ALT line (the value is overridden by the estimator, :177).
        no_acc_every / no_acc_ang_every : every k-th step lacks its ACC line / its ACC and ANG lines, so that the
                    step length falls back to the ANG / QUAT channel (:158-166); the next ACC line then spans two steps.
        alt_only_after : index of a step after which a group holding nothing but an ALT line is inserted
                    (2 ms later; step length from the ALT channel, a predict-only iteration).
        """
        rng = np.random.Generator(np.random.PCG64(self.seed if noise_seed is None else noise_seed))
        sens, feat = [], []
        for k, s in enumerate(self.steps):
            t = s.t
            if k == 0 and self.init_pos is not None:
                # INIT is stored x/y swapped by the estimator (reference: src/slam/main.cpp:132-137)
                p = self.init_pos
                sens.append("%.6f,INIT,%.17g,%.17g,%.17g" % (t, p[1], p[0], p[2]))
            sens.append("%.6f,QUAT,%.17g,%.17g,%.17g,%.17g" % (t, *s.quat))
            skip_ang = bool(no_acc_ang_every and k > 0 and k % no_acc_ang_every == 0)
            skip_acc = skip_ang or bool(no_acc_every and k > 0 and k % no_acc_every == 0)
            if not skip_ang:
                sens.append("%.6f,ANG,%.17g,%.17g,%.17g" % (t, *s.ang))
            if not skip_acc:
                sens.append("%.6f,ACC,%.17g,%.17g,%.17g" % (t, *s.acc))
            if alt:
                sens.append("%.6f,ALT,%.17g" % (t, -s.X_true[2] + 0.01 * np.sin(k)))
            if pos_every and k > 0 and k % pos_every == 0:
                p = s.X_true + np.array([0.01, -0.02, 0.003])
                sens.append("%.6f,POS,%.17g,%.17g,%.17g" % (t, p[1], p[0], p[2]))
            if vel_every and k > 0 and k % vel_every == 0 and k + 1 < len(self.steps):
                v = (self.steps[k + 1].X_true - s.X_true) / self.dt + np.array([0.01, 0.0, -0.01])
                sens.append("%.6f,VEL,%.17g,%.17g,%.17g" % (t, *v))
            if s.image:
                sens.append("%.6f,IMG,0" % t)
                src, refl = self.noisy_frame(k, rng)
                for i in range(len(s.ids)):
                    feat.append("%d,%.17g,%.17g,%.17g,%.17g" % (s.ids[i], src[i, 0], src[i, 1], refl[i, 0], refl[i, 1]))
                feat.append("")
            if alt_only_after is not None and k == alt_only_after:
                sens.append("%.6f,ALT,%.17g" % (t + 0.012, -s.X_true[2]))
        return "\n".join(sens) + "\n", "\n".join(feat) + "\n"


def gen_stream(seed=1, nf=20, steps=90, ref_fraction=0.5, image_every=3, rate_hz=30.0,
               churn=True, retire_dx=4.0, attitude=True, noise=1e-3, t0=1.0,
               drop_at=None, drop_count=0, add_at=None, add_count=0, spawn_dx=(8.0, 20.0)):
    """Build a stream of `steps` time steps with `nf` live features.

    churn       : retire features closer than `retire_dx` ahead and replace them by a new id
                  (exercises in-place slot replacement of the association step).
    drop_at/add_at : step indices at which `drop_count` features vanish without replacement /
                  `add_count` extra features appear (exercises deletion and append).
    spawn_dx    : range of the distance ahead at which features appear (a narrow range just above
                  `retire_dx` gives a steady turnover of features).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    dt = 1.0 / rate_hz
    v = np.array([1.0, 0.0, 0.0])
    X0 = np.array([0.0, 0.0, 1.34])

    def truth_pos(t):
        return X0 + v * t + np.array([0.0, 0.3 * np.sin(0.5 * t), 0.0])

    def truth_att(t):
        if not attitude:
            return np.array([0.0, 0.0, 0.0, 1.0])
        return yaw_pitch_quat(0.08 * np.sin(0.4 * t), 0.05 * np.sin(0.3 * t + 0.5))

    def spawn(X, k):
        return X + np.stack([rng.uniform(spawn_dx[0], spawn_dx[1], k), rng.uniform(-4, 4, k), rng.uniform(-3, -0.5, k)], axis=1)

    n_ref = int(round(ref_fraction * nf))
    next_id = 0
    world, ids, is_ref = spawn(truth_pos(0.0), nf), [], []
    for i in range(nf):
        ids.append(next_id)
        # alternate ids with / without reflection until the quota is used
        is_ref.append(1 if (i % 2 == 0 and sum(is_ref) < n_ref) or (nf - i <= n_ref - sum(is_ref)) else 0)
        next_id += 1
    ids, is_ref = np.array(ids, dtype=np.int32), np.array(is_ref, dtype=np.int32)

    st = Stream(seed=seed, dt=dt, noise=noise, init_pos=truth_pos(0.0).copy())
    S = np.diag([1.0, 1.0, -1.0])
    T = steps + 2
    ts = np.arange(T) * dt
    Xs = np.array([truth_pos(t) for t in ts])
    qs = np.array([truth_att(t) for t in ts])
    Rs = np.array([quat_to_rot(q) for q in qs])
    # body-frame velocity consistent with X' = Rb2w V (forward differences)
    Vb = np.array([Rs[k].T @ ((Xs[k + 1] - Xs[k]) / dt) for k in range(T - 1)])
    for k in range(steps):
        R = Rs[k]
        # body rate from consecutive attitudes: R_k^T R_{k+1} ~ I + [w]x dt
        dR = R.T @ Rs[k + 1]
        w = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]) / (2 * dt)
        acc = (Vb[k + 1] - Vb[k]) / dt if k + 1 < len(Vb) else np.zeros(3)
        image = (k % image_every == 0)
        s = Step(t=t0 + ts[k], quat=qs[k].copy(), ang=w, acc=acc, image=image, X_true=Xs[k].copy())
        if image:
            if drop_at is not None and k == drop_at and drop_count > 0:
                keep = np.ones(len(ids), bool)
                keep[rng.choice(len(ids), drop_count, replace=False)] = False
                world, ids, is_ref = world[keep], ids[keep], is_ref[keep]
            if add_at is not None and k == add_at and add_count > 0:
                world = np.concatenate([world, spawn(Xs[k], add_count)])
                ids = np.concatenate([ids, np.arange(next_id, next_id + add_count, dtype=np.int32)])
                is_ref = np.concatenate([is_ref, (np.arange(add_count) % 2).astype(np.int32)])
                next_id += add_count
            rel = (world - Xs[k]) @ R          # rows: Rw2b (f - X)
            if churn:
                for i in np.nonzero(rel[:, 0] < retire_dx)[0]:
                    world[i] = spawn(Xs[k], 1)[0]
                    ids[i] = next_id
                    next_id += 1
                rel = (world - Xs[k]) @ R
            mrel = (world @ S - Xs[k]) @ R
            src = np.stack([rel[:, 1] / rel[:, 0], rel[:, 2] / rel[:, 0]], axis=1)
            refl = np.stack([mrel[:, 1] / mrel[:, 0], mrel[:, 2] / mrel[:, 0]], axis=1) * is_ref[:, None]
            s.ids, s.src, s.refl = ids.copy(), src, refl
        st.steps.append(s)
    return st
