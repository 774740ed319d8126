"""Stage timers of the batched step at feature counts behind the per-filter kernels' range (one GPU)."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from slam_amd import ekf

def point(nf, B, iters=4):
    dev = torch.device("cuda:0")
    st, steps = bench.build_inputs(torch, dev, B, nf, iters + 2, 0)
    fb = ekf.FilterBatch(max_features=nf, batch=B, device=0)
    for b in range(B):
        fb.set_state(b, X=st.init_pos)
    fb.frame(-1, bench.first_frame(st), st.steps[0].quat)
    s0 = torch.cuda.Stream(device=dev)
    def run(k):
        s = steps[k]
        fb.step_dev(s["qbw"].data_ptr(), s["w"].data_ptr(), s["acc"].data_ptr(), s["dt"].data_ptr(), None,
                    s["src"].data_ptr(), s["refl"].data_ptr(), None, s0.cuda_stream)
    run(0); run(1); fb.sync(s0.cuda_stream)
    fb.profile_enable(True)
    t0 = time.perf_counter()
    for k in range(2, 2 + iters):
        run(k)
    fb.sync(s0.cuda_stream)
    t = (time.perf_counter() - t0) / iters
    prof = fb.profile_read()
    fb.close()
    return {"nf": nf, "B": B, "ms": round(t * 1e3, 3), "stages": {k: round(v[0] / iters, 3) for k, v in prof.items() if v[0] > 0}}

for nf in [int(a) for a in sys.argv[1:]] or [36, 40, 48, 60]:
    print(json.dumps(point(nf, 1024)), flush=True)
