import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from slam_amd import ekf, synth
from oracle import oracle as orc
import helpers
st = synth.gen_stream(seed=1, nf=20, steps=31, churn=True)
inp = helpers.stream_inputs(st, 7)
f = orc.Filter(); f.X = st.init_pos
fb = ekf.FilterBatch(max_features=24, batch=1)
fb.set_state(0, X=st.init_pos)
u, quat, ang, acc, dt, matches = inp[0]
f.update_features(matches, quat); f.old_pos[:] = f.X
fb.frame(0, matches, quat)
for k in (1,2):
    u, quat, ang, acc, dt, matches = inp[k]
    s = orc.make_sensors(quat=quat, acc=acc, ang=ang)
    f.step(u, s, dt, matches)
    fb.predict(quat, ang, acc, [dt])
# step 3 has image
u, quat, ang, acc, dt, matches = inp[3]
assert u & orc.IMG
s = orc.make_sensors(quat=quat, acc=acc, ang=ang)
# oracle, manual
f.old_pos[:] = f.X
f.update_features(matches, quat)
F = f.motion_jacobian(s, dt); f.dynamics_step(s, dt)
meas, hmu, H, ref = f.measurement_model(f.old_pos, -1.34, matches, quat)
qd = orc.init_q_diag(f.nf, dt)
P = orc.calc_p(f.P, F, qd)
rd = orc.init_r_diag(ref)
K = orc.calc_k(H, P, rd)
S = H@P@H.T + np.diag(rd)
Lo = np.linalg.cholesky(S)
P2 = orc.update_p(P, K, H)
# gpu
fb.set_debug(True)
fb.frame(0, matches, quat)
fb.predict(quat, ang, acc, [dt])
Pg = fb.get_cov(0)
print('P pred', helpers.rel_err(Pg, P))
fb.update(quat)
d = fb.debug_get(0)
print('H', helpers.rel_err(d['H'], H), 'e', helpers.rel_err(d['e'], meas-hmu))
print('L', helpers.rel_err(np.tril(d['L']), Lo))
print('K', helpers.rel_err(d['K'], K))
Pg2 = fb.get_cov(0)
print('P upd', helpers.rel_err(Pg2, P2))
A = np.eye(P.shape[0]) - K@H
PJ = A@P@A.T + K@np.diag(rd)@K.T
print('P upd vs numpy joseph', helpers.rel_err(Pg2, PJ), 'joseph vs simple', helpers.rel_err(PJ, P2))
Kn = P@H.T@np.linalg.inv(S)
print('K numpy vs oracle', helpers.rel_err(Kn, K))
dL = np.abs(np.tril(d['L'])-Lo); print('L maxerr at', np.unravel_index(dL.argmax(), dL.shape), dL.max(), 'm', S.shape)
Yn = np.linalg.solve(Lo, (P@H.T).T).T
print('cond S', np.linalg.cond(S))
