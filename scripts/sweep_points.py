"""Updates/s of the batched step at chosen feature counts (bench.sweep_point, 1024 filters)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from slam_amd import ekf
dev = torch.device("cuda:0")
for nf in [int(a) for a in sys.argv[1:]] or [36, 40, 44, 48, 60]:
    r = bench.sweep_point(torch, dev, nf, 1024)
    n, m = r["state_dim"], r["meas_dim"]
    fl = m ** 3 / 3 + 2 * n * m * m + 3 * n ** 3 + n * n * m
    r["algorithmic_tflops"] = fl * r["updates_per_s"] / 1e12
    print(json.dumps(r), flush=True)
