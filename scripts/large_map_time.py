"""One-GPU large map (2000 features): update time and stage timers (bench.large_map)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
r = bench.large_map(torch, torch.device("cuda:0"), nf=int(sys.argv[1]) if len(sys.argv) > 1 else 2000, iters=3)
print(json.dumps({k: r[k] for k in ("ms_per_update", "stages_ms", "cov_sane")}))
