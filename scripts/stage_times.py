#!/usr/bin/env python3
"""Stage times (HIP events inside the library) of one workload point: python scripts/stage_times.py NF BATCH [STEPS [REF_FRACTION]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from slam_amd import ekf

nf, B = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda", 0)
ref = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
st, steps = bench.build_inputs(torch, dev, B, nf, iters + 2, 0, ref)
fb = ekf.FilterBatch(max_features=nf, batch=B)
for b in range(B):
    fb.set_state(b, X=st.init_pos)
fb.frame(-1, bench.first_frame(st), st.steps[0].quat)
s0 = torch.cuda.Stream(device=dev)
def run(k):
    s = steps[k]
    fb.step_dev(s["qbw"].data_ptr(), s["w"].data_ptr(), s["acc"].data_ptr(), s["dt"].data_ptr(), None,
                s["src"].data_ptr(), s["refl"].data_ptr(), None, s0.cuda_stream)
run(0); run(1); fb.sync(s0.cuda_stream)
fb.profile_enable(True); fb.profile_read()
t0 = time.perf_counter()
for k in range(2, 2 + iters):
    run(k)
fb.sync(s0.cuda_stream)
t = (time.perf_counter() - t0) / iters
prof = fb.profile_read()
n, m = 9 + 3 * nf, fb.num_rows(0)
print("nf=%d n=%d m=%d B=%d: %.3f ms/step, %.0f updates/s" % (nf, n, m, B, t * 1e3, B / t))
print({k: round(v[0] / iters, 3) for k, v in prof.items()})
print("MFMA flops/update %.3e" % ekf.flops_update(n, m, 0))
