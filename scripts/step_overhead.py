"""Wall time per batched step (4096 filters x 32 features) with the stage timers on and off, plain and in graph mode."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from slam_amd import ekf
dev = torch.device("cuda:0")
B, nf, K, W = 4096, 32, 100, 5
st, steps = bench.build_inputs(torch, dev, B, nf, K + W, 0)
for prof, graph in ((True, False), (False, False), (False, True)):
    fb = ekf.FilterBatch(max_features=nf, batch=B, device=0)
    for b in range(B):
        fb.set_state(b, X=st.init_pos)
    fb.frame(-1, bench.first_frame(st), st.steps[0].quat)
    if graph:
        fb.set_graph_mode(True)
    s0 = torch.cuda.Stream(device=dev)
    def run(k):
        s = steps[k]
        fb.step_dev(s["qbw"].data_ptr(), s["w"].data_ptr(), s["acc"].data_ptr(), s["dt"].data_ptr(), None,
                    s["src"].data_ptr(), s["refl"].data_ptr(), None, s0.cuda_stream)
    for k in range(W):
        run(k)
    fb.sync(s0.cuda_stream)
    fb.profile_enable(prof)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(W, W + K):
        run(k)
    s0.synchronize()
    t = (time.perf_counter() - t0) / K
    print("stage timers %s, graph mode %s: %.4f ms per step" % (prof, graph, t * 1e3), flush=True)
    fb.profile_enable(False)
    fb.close()
