#!/usr/bin/env python3
"""Time of the single large-map filter (bench.large_map): python scripts/large_map_time.py [NF]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
r = bench.large_map(torch, torch.device("cuda", 0), nf=nf)
print(json.dumps({k: r[k] for k in ("ms_per_update", "gemm_tflops", "stages_ms", "cov_sane")}))
