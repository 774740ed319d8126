#!/usr/bin/env python3
"""Latency of one small filter (bench.single_filter_latency): python scripts/single_filter_time.py [NF]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print(json.dumps(bench.single_filter_latency(torch, torch.device("cuda", 0), nf=nf, iters=200)))
