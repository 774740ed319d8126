#!/usr/bin/env python3
"""Large-map filter (BASELINE.json configs[3]) sharded by row blocks over the GPUs of one node.

    python scripts/bench_sharded_large_map.py                      # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           scripts/bench_sharded_large_map.py --nf 2000 --steps 3

Every rank holds the replicated O(nf) state and its row block of P; one all-gather (RCCL) per
contraction.  Rank 0 prints one JSON line with updates/s and the achieved fp64 rate per GPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nf", type=int, default=2000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SLAM_BENCH_BACKEND", "nccl")     # gloo: rehearsal with every rank on the GPUs that exist
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from slam_amd import ekf, synth
    from slam_amd.sharded import ShardedFilter
    nf = args.nf
    st = synth.gen_stream(seed=2, nf=nf, steps=args.steps + args.warmup + 1, churn=False, image_every=1)
    fb = ekf.FilterBatch(max_features=nf, batch=1, device=local)
    fb.set_state(0, X=st.init_pos)
    s0 = st.steps[0]
    fb.frame(0, [(int(s0.ids[i]), s0.src[i, 0], s0.src[i, 1], s0.refl[i, 0], s0.refl[i, 1]) for i in range(nf)], s0.quat)
    stream = torch.cuda.Stream(device=dev)
    sh = ShardedFilter(fb, rank, world, None, device=dev, stream=stream)
    P0 = torch.from_numpy(fb.get_cov(0)).to(dev)
    torch.cuda.synchronize(dev)
    sh.load_cov(P0)
    t = lambda v: torch.tensor(np.asarray(v, dtype=np.float64), device=dev)
    rng = np.random.default_rng(5)

    def step(k):
        s = st.steps[k]
        src = s.src + st.noise * rng.standard_normal(s.src.shape)
        refl = s.refl + st.noise * rng.standard_normal(s.refl.shape) * (s.refl != 0)
        fb.set_frame_values(0, src, refl)
        q, w, a, d = t(s.quat), t(s.ang), t(s.acc), t([s.t - st.steps[k - 1].t])
        torch.cuda.synchronize(dev)
        fb.frame_begin_dev(sh.sptr)
        sh.predict(q, w, a, d)
        sh.update(q)

    for k in range(1, 1 + args.warmup):
        step(k)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(1 + args.warmup, 1 + args.warmup + args.steps):
        step(k)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dtm = (time.perf_counter() - t0) / args.steps
    n, m = sh.dims()["n"], sh.dims()["m"]
    dense = 8.0 * n**3 + 6.0 * n * n * m + 4.0 * n * m * m + m**3 / 3.0
    P = sh.gather_cov()
    ok = bool(torch.isfinite(P).all().item() and (torch.diagonal(P) > 0).all().item())
    if rank == 0:
        print(json.dumps({"workload": "single filter, %d features (n=%d, m=%d), row-block sharded x%d" % (nf, n, m, world),
                          "updates_per_s": 1.0 / dtm, "ms_per_update": dtm * 1e3,
                          "dense_model_tflops_total": dense / dtm / 1e12, "n_gpus": world, "cov_sane": ok}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
