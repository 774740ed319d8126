#!/usr/bin/env python3
"""Benchmark of the EKF predict/update hot path on MI355X.

Workload: batched Monte Carlo of independent filters x 32 features (n = 105, m = 161 with half of
the features reflective), synthetic sensor_emu-style data.  One GPU: BASELINE.json configs[2], 4096
filters.  N > 1 GPUs: configs[4], 65536 filters over 8 GPUs = 8192 filters per GPU (the same per-GPU
shard at every N > 1; `config.workload` names what ran).  One step = one loop iteration WITH an image for every filter of the
batch: frame begin (old_pos, depth clamp), predict (F, dynamics, P = F P F^T + Q) and update
(measurement model + Jacobian, S, Cholesky gain, Joseph covariance update, gated mean update).
All inputs of every step are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: independent filters are sharded over ranks (B filters per rank, weak scaling), no
data-path collective; torch.distributed (RCCL) is only used for the barrier and the MAX of the
per-rank times.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6   # MI355X: 256 CU x 2.4 GHz x 128 fp64 FLOP/clk/CU (v_mfma_f64_16x16x4_f64)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E


def kernel_source_sha256():
    """Hash of slam_amd/csrc (the same as scripts/pmc_summarize.py stores beside the counters it collects)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "slam_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def build_inputs(torch, dev, B, nf, nsteps, rank, ref_fraction=0.5):
    """Truth stream shared by all filters, independent pixel noise per filter and step."""
    from slam_amd import synth
    st = synth.gen_stream(seed=1, nf=nf, steps=nsteps + 1, churn=False, image_every=1, ref_fraction=ref_fraction)
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + rank)
    steps = []
    for k in range(1, nsteps + 1):
        s, prev = st.steps[k], st.steps[k - 1]
        one = lambda v: torch.tensor(np.tile(np.asarray(v, dtype=np.float64), (B, 1)), device=dev)
        src = torch.tensor(s.src.T.copy(), device=dev)            # [2][F]
        refl = torch.tensor(s.refl.T.copy(), device=dev)
        has = (refl != 0).to(torch.float64)
        srcb = src.unsqueeze(0) + st.noise * torch.randn((B, 2, nf), generator=g, device=dev, dtype=torch.float64)
        reflb = refl.unsqueeze(0) + st.noise * torch.randn((B, 2, nf), generator=g, device=dev, dtype=torch.float64) * has
        steps.append({
            "qbw": one(s.quat), "w": one(s.ang), "acc": one(s.acc),
            "dt": torch.full((B,), s.t - prev.t, device=dev, dtype=torch.float64),
            "src": srcb.contiguous(), "refl": reflb.contiguous(),
        })
    return st, steps


def first_frame(st):
    s = st.steps[0]
    return [(int(s.ids[i]), s.src[i, 0], s.src[i, 1], s.refl[i, 0], s.refl[i, 1]) for i in range(len(s.ids))]


def cpu_baseline(st, steps, nf, k0, k1, budget_s=12.0, max_filters=256):
    """The CPU restatement (oracle/, 1 thread, dense products + LU as the reference) on a bounded
    sample of the same workload: the first `cnt` filters of rank 0 over the timed steps."""
    from oracle import oracle as orc
    done, t_total, cnt = 0, 0.0, 0
    host = [{k: v[:max_filters].cpu().numpy() for k, v in s.items()} for s in steps]
    while t_total < budget_s and cnt < min(max_filters, host[0]["src"].shape[0]):
        f = orc.Filter()
        f.X = st.init_pos
        f.update_features(first_frame(st), st.steps[0].quat)
        f.old_pos[:] = f.X
        ids = [int(i) for i in st.steps[0].ids]
        for k, s in enumerate(host):
            m = [(ids[i], s["src"][cnt, 0, i], s["src"][cnt, 1, i], s["refl"][cnt, 0, i], s["refl"][cnt, 1, i])
                 for i in range(nf)]
            sens = orc.make_sensors(quat=s["qbw"][cnt], acc=s["acc"][cnt], ang=s["w"][cnt])
            u = orc.ACC | orc.ANG | orc.QUAT | orc.IMG
            t0 = time.perf_counter()
            rc, _, _ = f.step(u, sens, float(s["dt"][cnt]), m)
            t1 = time.perf_counter()
            assert rc == 0
            if k0 <= k < k1:
                t_total += t1 - t0
                done += 1
        cnt += 1
    return {"value": done / t_total, "unit": "EKF updates/s", "cores": 1, "kind": "port",
            "sample": "%d filters x %d steps (nf=%d) of the same batch, oracle/ekf_oracle.c orc_step, 1 thread" % (cnt, k1 - k0, nf)}


def churn_workload(torch, dev, B, nf, K, rank):
    """One step = image frame of every filter (association + covariance surgery) + predict + update, on a stream whose
    features are retired and replaced; the frame is the same for all filters, their pixel noise is not."""
    from slam_amd import ekf, synth
    st = synth.gen_stream(seed=1, nf=nf, steps=K + 4, churn=True, image_every=1, ref_fraction=0.5, retire_dx=8.0,
                          spawn_dx=(8.0, 9.0))
    g = torch.Generator(device=dev)
    g.manual_seed(77 + rank)
    fb = ekf.FilterBatch(max_features=nf, batch=B, device=dev.index or 0)
    for b in range(B):
        fb.set_state(b, X=st.init_pos)
    fb.frame(-1, first_frame(st), st.steps[0].quat)
    stream = torch.cuda.Stream(device=dev)
    steps, frames, replaced = [], [], 0
    for k in range(1, K + 3):
        s, prev = st.steps[k], st.steps[k - 1]
        one = lambda v: torch.tensor(np.tile(np.asarray(v, dtype=np.float64), (B, 1)), device=dev)
        src = torch.tensor(s.src.T.copy(), device=dev)
        refl = torch.tensor(s.refl.T.copy(), device=dev)
        has = (refl != 0).to(torch.float64)
        steps.append({"qbw": one(s.quat), "w": one(s.ang), "acc": one(s.acc),
                      "dt": torch.full((B,), s.t - prev.t, device=dev, dtype=torch.float64),
                      "src": (src.unsqueeze(0) + st.noise * torch.randn((B, 2, nf), generator=g, device=dev, dtype=torch.float64)).contiguous(),
                      "refl": (refl.unsqueeze(0) + st.noise * torch.randn((B, 2, nf), generator=g, device=dev, dtype=torch.float64) * has).contiguous()})
        frames.append(([(int(s.ids[i]), s.src[i, 0], s.src[i, 1], s.refl[i, 0], s.refl[i, 1]) for i in range(nf)], s.quat))
        if k >= 3:
            replaced += int(np.sum(s.ids != prev.ids))
    torch.cuda.synchronize(dev)

    def run(k):
        fb.frame_batch_async(frames[k][0], frames[k][1], None, stream.cuda_stream)
        d = steps[k]
        fb.step_dev(d["qbw"].data_ptr(), d["w"].data_ptr(), d["acc"].data_ptr(), d["dt"].data_ptr(), None,
                    d["src"].data_ptr(), d["refl"].data_ptr(), None, stream.cuda_stream)
    run(0); run(1)
    fb.sync(stream.cuda_stream)
    t0 = time.perf_counter()
    for k in range(2, 2 + K):
        run(k)
    fb.sync(stream.cuda_stream)
    t = time.perf_counter() - t0
    ids_ok = list(fb.get_features(B - 1)["ids"]) == [int(i) for i in st.steps[K + 2].ids]
    P = fb.get_cov(B // 3)
    sane = bool(ids_ok and np.all(np.isfinite(P)) and np.array_equal(P, P.T) and np.all(np.diag(P) > 0))
    fb.close()
    return {"what": "slamk_frame_batch_async (one frame for all filters, features retired and replaced) + slamk_step_dev "
                    "per step, same stream, no host synchronisation inside the timed region",
            "updates_per_s": B * K / t, "ms_per_step": t / K * 1e3, "steps": K,
            "features_replaced_per_step": replaced / K, "outputs_sane": sane}


def ceilings(torch, dev):
    """Measured ceilings of this box: fp64 MFMA issue rate, the library's own DGEMM, and a device-to-device copy."""
    import ctypes as C
    from slam_amd import ekf, _lib
    out = {"fp64_mfma_issue_tflops": ekf.mfma_f64_probe(4096)}
    n = 4096
    A = torch.randn(n, n, dtype=torch.float64, device=dev)
    Bm = torch.randn(n, n, dtype=torch.float64, device=dev)
    Cm = torch.zeros(n, n, dtype=torch.float64, device=dev)
    L = _lib.lib()
    def gemm():
        L.slamk_dgemm_dev(1, n, n, n, 1.0, A.data_ptr(), n, 0, Bm.data_ptr(), n, 0, 0.0, Cm.data_ptr(), n, 0, 1, None)
    gemm()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(5):
        gemm()
    torch.cuda.synchronize(dev)
    out["dgemm_4096_tflops"] = 5 * 2.0 * n ** 3 / (time.perf_counter() - t0) / 1e12
    src = torch.empty(1 << 27, dtype=torch.float64, device=dev)      # 1 GiB
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(5):
        dst.copy_(src)
    torch.cuda.synchronize(dev)
    out["stream_copy_gbs"] = 5 * 2.0 * src.numel() * 8 / (time.perf_counter() - t0) / 1e9
    return out


def single_filter_latency(torch, dev, nf=20, iters=40, form=0):
    """configs[1]: one filter, 20 features, latency of one predict+update on one GPU (host-driven)."""
    from slam_amd import ekf
    st, steps = build_inputs(torch, dev, 1, nf, iters + 5, 0)
    prm = ekf.default_params()
    prm.update_form = form
    fb = ekf.FilterBatch(max_features=nf, batch=1, device=dev.index or 0, params=prm)
    fb.set_state(0, X=st.init_pos)
    fb.frame(0, first_frame(st), st.steps[0].quat)
    s0 = torch.cuda.Stream(device=dev)
    # one fixed set of input buffers, refilled per iteration on the same stream (what a live caller does); the
    # iteration itself is slamk_step_dev in graph mode: one graph submission per update
    buf = {k: torch.empty_like(v) for k, v in steps[0].items()}
    fb.set_graph_mode(True)
    def run(k):
        with torch.cuda.stream(s0):
            for name, v in steps[k].items():
                buf[name].copy_(v, non_blocking=True)
        fb.step_dev(buf["qbw"].data_ptr(), buf["w"].data_ptr(), buf["acc"].data_ptr(), buf["dt"].data_ptr(), None,
                    buf["src"].data_ptr(), buf["refl"].data_ptr(), None, s0.cuda_stream)
    for k in range(5):
        run(k)
    fb.sync(s0.cuda_stream)
    t0 = time.perf_counter()
    for k in range(5, 5 + iters):
        run(k)
        s0.synchronize()
    t = (time.perf_counter() - t0) / iters
    # the same without the input refresh: latency of the filter iteration alone
    t1 = time.perf_counter()
    for k in range(iters):
        fb.step_dev(buf["qbw"].data_ptr(), buf["w"].data_ptr(), buf["acc"].data_ptr(), buf["dt"].data_ptr(), None,
                    buf["src"].data_ptr(), buf["refl"].data_ptr(), None, s0.cuda_stream)
        s0.synchronize()
    t_iter = (time.perf_counter() - t1) / iters
    fb.close()
    return {"workload": "single filter, %d features (n=%d), slamk_step_dev in graph mode, host-synchronised every update"
                        % (nf, 9 + 3 * nf), "update_form": form,
            "latency_us_per_update": t * 1e6, "latency_us_iteration_only": t_iter * 1e6}


def sweep_point(torch, dev, nf, B, iters=6, form=0):
    """Updates/s at another (state dim, batch) point: same step as the headline workload, groups = 1."""
    from slam_amd import ekf
    st, steps = build_inputs(torch, dev, B, nf, iters + 2, 0)
    prm = ekf.default_params()
    prm.update_form = form
    fb = ekf.FilterBatch(max_features=nf, batch=B, device=dev.index or 0, params=prm)
    for b in range(B):
        fb.set_state(b, X=st.init_pos)
    fb.frame(-1, first_frame(st), st.steps[0].quat)
    s0 = torch.cuda.Stream(device=dev)
    def run(k):
        s = steps[k]
        fb.step_dev(s["qbw"].data_ptr(), s["w"].data_ptr(), s["acc"].data_ptr(), s["dt"].data_ptr(), None,
                    s["src"].data_ptr(), s["refl"].data_ptr(), None, s0.cuda_stream)
    run(0); run(1)
    fb.sync(s0.cuda_stream)
    t0 = time.perf_counter()
    for k in range(2, 2 + iters):
        run(k)
    fb.sync(s0.cuda_stream)
    t = (time.perf_counter() - t0) / iters
    n, m = 9 + 3 * nf, fb.num_rows(0)
    P = fb.get_cov(B // 2)
    sane = bool(np.all(np.isfinite(P)) and np.all(np.diag(P)[:n] > 0))
    fb.close()
    return {"features": nf, "state_dim": n, "meas_dim": m, "batch": B, "updates_per_s": B / t, "ms_per_step": t * 1e3,
            "cov_sane": sane}


def large_map(torch, dev, nf=2000, iters=2):
    """configs[3] on one GPU: one filter with 2000 features (n = 6009, m = 10001)."""
    from slam_amd import ekf
    st, steps = build_inputs(torch, dev, 1, nf, iters + 1, 0)
    fb = ekf.FilterBatch(max_features=nf, batch=1, device=dev.index or 0)
    fb.set_state(0, X=st.init_pos)
    fb.frame(0, first_frame(st), st.steps[0].quat)
    n, m = 9 + 3 * nf, fb.num_rows(0)
    s0 = torch.cuda.Stream(device=dev)
    def run(k):
        s = steps[k]
        fb.frame_begin_dev(s0.cuda_stream)
        fb.predict_dev(s["qbw"].data_ptr(), s["w"].data_ptr(), s["acc"].data_ptr(), s["dt"].data_ptr(), s0.cuda_stream)
        fb.update_dev(s["qbw"].data_ptr(), None, s["src"].data_ptr(), s["refl"].data_ptr(), None, s0.cuda_stream)
    run(0)
    fb.sync(s0.cuda_stream)
    fb.profile_enable(True)
    t0 = time.perf_counter()
    for k in range(1, 1 + iters):
        run(k)
    fb.sync(s0.cuda_stream)
    t = (time.perf_counter() - t0) / iters
    prof = fb.profile_read()
    gemm_ms = sum(prof[k][0] for k in prof if k.startswith("gemm")) / iters
    flops = ekf.flops_update(n, m, 0)
    cov_ms = prof["gemm_cov_update"][0] / iters            # A P, (A P) A^T, (K R) K^T
    cov_flops = ekf.flops_update_stage(n, m, 0, 1)
    P = fb.get_cov(0)
    ok = bool(np.all(np.isfinite(P)) and np.abs(P - P.T).max() == 0 and np.all(np.diag(P) > 0))
    fb.close()
    return {"workload": "single filter, %d features (n=%d, m=%d)" % (nf, n, m), "updates_per_s": 1.0 / t,
            "ms_per_update": t * 1e3, "gemm_tflops": flops / (gemm_ms * 1e-3) / 1e12,
            "frac_of_fp64_matrix_peak": flops / (gemm_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
            "covariance_update": {"ms": cov_ms, "tflops": cov_flops / (cov_ms * 1e-3) / 1e12,
                                  "frac_of_fp64_matrix_peak": cov_flops / (cov_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                                  "what": "Joseph form as executed: A P, lower tiles of (A P) A^T + (K R) K^T; HIP events"},
            "stages_ms": {k: round(v[0] / iters, 3) for k, v in prof.items()}, "cov_sane": ok}


def large_map_sharded(torch, dev, dist, rank, world, nf=2000, iters=2):
    """configs[3] over all ranks: ONE filter with `nf` features, its covariance sharded by row blocks behind the C
    boundary (slamk_sharded_*), the all-gathers issued by the library on RCCL (or, on a process group without device
    collectives, through its host-staged callback).  Every rank feeds the same frames (the O(nf) state is replicated).
    Time = MAX over ranks of `iters` iterations (predict + update) between barriers."""
    from slam_amd import ekf
    from slam_amd.sharded import ShardedFilter
    st, steps = build_inputs(torch, dev, 1, nf, iters + 1, 0)      # the same stream and noise on every rank
    fb = ekf.FilterBatch(max_features=nf, batch=1, device=dev.index or 0)
    fb.set_state(0, X=st.init_pos)
    fb.frame(0, first_frame(st), st.steps[0].quat)
    n, m = 9 + 3 * nf, fb.num_rows(0)
    s0 = torch.cuda.Stream(device=dev)
    sh = ShardedFilter(fb, rank, world, None, device=dev, stream=s0)
    P0 = torch.from_numpy(fb.get_cov(0)).to(dev)
    torch.cuda.synchronize(dev)
    sh.load_cov(P0)
    dims = sh.dims()
    cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")

    def run(k):
        s = steps[k]
        fb.frame_begin_dev(s0.cuda_stream)
        sh.predict(s["qbw"], s["w"], s["acc"], s["dt"])
        return sh.update(s["qbw"], s["src"], s["refl"])
    run(0)
    sh.sync()
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(1, 1 + iters):
        applied = run(k)
    sh.sync()
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], device=cdev, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t = float(tt.item()) / iters
    P = sh.gather_cov()
    sh.sync()
    P = P.cpu().numpy()
    ok = bool(np.all(np.isfinite(P)) and np.array_equal(P, P.T) and np.all(np.diag(P) > 0) and int(applied.item()) == 1)
    # every rank must hold the same gathered covariance
    chk = torch.tensor([float(np.trace(P)), float(np.linalg.norm(P))], device=cdev, dtype=torch.float64)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    ok = ok and bool(torch.equal(lo, hi))
    N, M, g = dims["n_padded"], dims["m_padded"], world
    ob = 512
    panels = sum(g * (((M - J0 + g - 1) // g + 15) // 16 * 16) * min(ob, M - J0) for J0 in range(0, M, ob))
    gathered = 8.0 * (13 * g * N + 2 * N * M + panels + N * N)          # rows of T, W, panels of S, K, P
    sh.close()
    fb.close()
    return {"workload": "single filter, %d features (n=%d, m=%d), covariance sharded by row blocks over %d GPUs "
                        "(slamk_sharded_predict_dev + slamk_sharded_update_dev, %s all-gathers)"
                        % (nf, n, m, world, "RCCL" if dist.get_backend() == "nccl" else "host-staged (rehearsal)"),
            "ms_per_update": t * 1e3, "updates_per_s": 1.0 / t, "iters": iters,
            "rows_per_rank": dims["rows_per_rank"], "bytes_gathered_per_update": gathered,
            "bytes_received_per_rank_per_update": gathered * (g - 1) / g, "cov_sane": ok}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch-per-gpu", type=int, default=0, help="default: 4096 on one GPU, 8192 on several")
    ap.add_argument("--nf", type=int, default=32)
    ap.add_argument("--groups", type=int, default=1, help="ranges of the batch pipelined on streams of their own")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the single-filter and large-map side measurements")
    ap.add_argument("--sharded-nf", type=int, default=-1,
                    help="features of the row-block sharded large-map leg of a multi-GPU run (default: 2000 = BASELINE "
                         "configs[3]; 0 or --no-extra without this flag: skip)")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # SLAM_BENCH_BACKEND=gloo rehearses the multi-rank path with every rank on the GPUs that exist
        backend = os.environ.get("SLAM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
            local = local % torch.cuda.device_count()
    dev = torch.device("cuda", local)
    cdev = dev if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")   # where collectives run
    torch.cuda.set_device(dev)

    from slam_amd import ekf
    B = args.batch_per_gpu or (4096 if world == 1 else 8192)
    nf, K, W = args.nf, args.steps, args.warmup
    st, steps = build_inputs(torch, dev, B, nf, W + K, rank)
    fb = ekf.FilterBatch(max_features=nf, batch=B, device=local)
    for b in range(B):
        fb.set_state(b, X=st.init_pos)
    fb.frame(-1, first_frame(st), st.steps[0].quat)
    n, m = 9 + 3 * nf, fb.num_rows(0)
    stream = torch.cuda.Stream(device=dev)
    applied = torch.zeros(B, dtype=torch.int32, device=dev)

    fb.set_pipeline_groups(args.groups)

    def run(k):
        s = steps[k]
        fb.step_dev(s["qbw"].data_ptr(), s["w"].data_ptr(), s["acc"].data_ptr(), s["dt"].data_ptr(), None,
                    s["src"].data_ptr(), s["refl"].data_ptr(), applied.data_ptr(), stream.cuda_stream)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for k in range(W):
        run(k)
    fb.sync(stream.cuda_stream)
    # Inside the timed region only the dominant kernel's stage carries events (two records per step): with all eight
    # stage timers on, their event records cost 0.05 ms of a 2 ms step.  The other stages are timed in a pass of their own
    # behind the timed region.
    fb.profile_enable(True, stages=("gain_solver",))
    fb.profile_read()
    # HIP events on the launch stream between the steps: per-step device times beside the wall clock
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    barrier()
    t0 = time.perf_counter()
    for k in range(W, W + K):
        marks[k - W].record(stream)
        run(k)
    marks[K].record(stream)
    stream.synchronize()
    barrier()
    t = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(K))
    prof = fb.profile_read()
    fb.profile_enable(False)
    fb.sync(stream.cuda_stream)          # also raises on a per-filter status word
    # every stage timed, outside the timed region: the same steps again (the filters just keep running)
    KS = min(K, 20)
    fb.profile_enable(True)
    for k in range(W, W + KS):
        run(k)
    prof_all = fb.profile_read()
    fb.profile_enable(False)
    fb.sync(stream.cuda_stream)
    if dist is not None:
        tt = torch.tensor([t], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t = float(tt.item())

    # multi-rank runs end with the gather of the filter means (outside the timed region)
    if dist is not None:
        mine = torch.tensor(np.stack([fb.get_state(b)[0] for b in range(0, B, max(1, B // 8))]), device=cdev)
        allx = torch.empty((world * mine.shape[0], mine.shape[1]), dtype=mine.dtype, device=cdev)
        dist.all_gather_into_tensor(allx, mine)
        assert bool(torch.isfinite(allx).all())

    # sanity on the product of the timed region (never timed): finite, symmetric, positive diagonal
    P = fb.get_cov(B // 2)
    sane = bool(np.all(np.isfinite(P)) and np.array_equal(P, P.T) and np.all(np.diag(P) > 0)
                and int(applied.sum().item()) == B)

    # predict-only iterations (no image), reported separately
    torch.cuda.synchronize(dev)
    tp0 = time.perf_counter()
    for k in range(W, W + K):
        s_ = steps[k]
        fb.predict_dev(s_["qbw"].data_ptr(), s_["w"].data_ptr(), s_["acc"].data_ptr(), s_["dt"].data_ptr(),
                       stream.cuda_stream)
    stream.synchronize()
    t_predict = (time.perf_counter() - tp0) / K

    # the boundary's host-pointer entry points (PCIe copies of the per-filter sensor inputs + host synchronisation
    # in every call); the observations are the ones kept from the last frame.  Reported beside `value`, never as it.
    hk = min(K, 5)
    hsteps = [{k: v.cpu().numpy() for k, v in steps[W + i].items() if k in ("qbw", "w", "acc", "dt")} for i in range(hk)]
    torch.cuda.synchronize(dev)
    th0 = time.perf_counter()
    for hs in hsteps:
        fb.predict(hs["qbw"], hs["w"], hs["acc"], hs["dt"])
        fb.update(hs["qbw"])
    t_host = (time.perf_counter() - th0) / hk

    # The same step with the image branch in front of it (src/slam/main.cpp:167-175 inside every iteration): a
    # stream whose features are retired and replaced along the way; association on the host, covariance surgery
    # on the device, queued on the same stream without synchronisation (slamk_frame_batch_async).
    churn = churn_workload(torch, dev, B, nf, min(K, 10), rank) if (rank == 0 and not args.no_extra) else None

    # BASELINE configs[3] over the ranks of this run: the one large-map filter, row blocks of its covariance per GPU,
    # RCCL all-gathers inside the library (every rank takes part; rank 0 reports)
    sharded_nf = args.sharded_nf if args.sharded_nf >= 0 else (0 if args.no_extra else 2000)
    sharded = None
    sharded_failed = False
    if dist is not None and sharded_nf > 0:
        for s_ in steps:
            s_.clear()                   # the batched workload's inputs are no longer needed: free their HBM
        torch.cuda.empty_cache()
        # This leg is a side measurement with collectives of its own: whatever happens in it (an exception on one rank,
        # a collective that never completes) must not cost the line of the batched workload measured above.  It runs
        # in a thread with a time limit; a rank whose leg failed or ran out of time skips every later collective, rank 0
        # prints its line, and the process ends without waiting for the others.
        import threading
        box = {}

        def leg():
            try:
                torch.cuda.set_device(dev)
                r = large_map_sharded(torch, dev, dist, rank, world, nf=sharded_nf)
                dist.barrier()
                box["result"] = r
            except BaseException as e:                                   # noqa: BLE001 (reported in the line)
                box["error"] = "%s: %s" % (type(e).__name__, e)
        th = threading.Thread(target=leg, daemon=True)
        th.start()
        th.join(float(os.environ.get("SLAM_BENCH_SHARDED_TIMEOUT_S", "300")))
        if th.is_alive():
            box["error"] = "no result within the time limit"
        if "error" in box:
            sharded_failed = True
            sharded = {"workload": "single filter, %d features, covariance sharded by row blocks over %d GPUs" % (sharded_nf, world),
                       "error": box["error"]}
        else:
            sharded = box["result"]
            if rank == 0 and not args.no_extra:
                try:
                    sharded["one_gpu_unsharded_ms_per_update"] = large_map(torch, dev, nf=sharded_nf)["ms_per_update"]
                except Exception as e:                                   # noqa: BLE001
                    sharded["one_gpu_unsharded_ms_per_update"] = None
                    sharded["one_gpu_unsharded_error"] = "%s: %s" % (type(e).__name__, e)

    if rank == 0:
        # Dominant kernel: gain_solver_kernel (W = P H^T, S = H W + R, tile Cholesky, K = W S^-1), one launch per step, timed
        # with HIP events on the launch stream.  Its algorithmic work is the 16^3 tile products it executes.
        nl, ml = (n + 15) // 16 * 16, (m + 15) // 16 * 16
        solver_ms = prof["gain_solver"][0] / K
        solver_exec_flops = B * ekf.flops_update_stage(n, m, 0, 0)          # 16^3 tile products as executed (padding included)
        # algorithmic work of the stage, no padding: Cholesky m^3/3, two triangular solves with n right-hand sides
        # 2 n m^2, W = P H^T 12 n m and the lower half of S = H W + R 6 m (m + 1) with the six-entry rows of H
        solver_flops = B * (m ** 3 / 3.0 + 2.0 * n * m * m + 12.0 * n * m + 6.0 * m * (m + 1))
        achieved = solver_flops / (solver_ms * 1e-3) / 1e12
        # HBM bytes per launch from the PMC counters (scripts/pmc_traffic.sh): collected in passes of their own on this
        # workload and kept with a hash of the kernel sources they belong to; reported only while that hash still matches
        pmc, pmc_state = {}, "none collected for this workload"
        side = os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")
        if os.path.exists(side) and B == 4096 and nf == 32:      # the counters were collected on this workload
            try:
                rec = json.load(open(side))
                if rec.get("kernel_source_sha256") == kernel_source_sha256():
                    pmc, pmc_state = rec["kernels"], "profiles/r03_pmc_traffic.json (rocprofv3 --pmc, FETCH_SIZE doubled)"
                else:
                    pmc_state = "stale: the kernel sources changed after profiles/r03_pmc_traffic.json was collected"
            except Exception:
                pmc = {}
        def pmc_bytes(name):
            for k, v in pmc.items():
                if name in k:
                    return v["hbm_bytes_per_launch_fetch_doubled"]
            return None
        cov_ms = prof_all["gemm_cov_update"][0] / KS
        cov_exec_flops = B * ekf.flops_update_stage(n, m, 0, 1)
        cov_flops = B * (3.0 * n ** 3 + 1.0 * n * n * m)      # A P 2n^3, lower (A P) A^T n^3, lower K R K^T n^2 m
        pred_ms = prof_all["struct_predict"][0] / KS
        pred_bytes = 2.0 * B * nl * nl * 8
        # which kernel the library takes (launch_joseph, slam_amd/csrc/slamk.hip): seven row blocks and 112..240
        # measurement columns in chunks -> four wavefronts per filter; up to n = 112 -> one pass over K; else two
        mlc = m - 1 if (m - 1) % 16 == 0 else (m + 15) // 16 * 16
        if nl == 112 and 112 <= mlc <= 240:
            cov_name, cov_what = "joseph_quad_kernel", "joseph_quad_kernel (four wavefronts per filter: K R K^T, A = I - K H and dx = K e on one pass over K; A P; (A P) A^T)"
        elif nl <= 112:
            cov_name, cov_what = "joseph_merged_kernel", "joseph_merged_kernel (K R K^T, A = I - K H and dx = K e on one pass over K; A P; (A P) A^T)"
        else:
            cov_name, cov_what = "joseph_kernel", "joseph_kernel (A = I - K H, A P, (A P) A^T + K R K^T)"
        kernels = [
            {"kernel": cov_what, "bound": "mfma", "ms": cov_ms,
             "achieved": cov_flops / (cov_ms * 1e-3) / 1e12, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": cov_flops / (cov_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
             "traffic": pmc_bytes(cov_name),
             "executed_tile_flops_frac": cov_exec_flops / (cov_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS},
            {"kernel": "predict_cov_kernel (P = F P F^T + Q, F in compact row form)", "bound": "hbm", "ms": pred_ms,
             "achieved": pred_bytes / (pred_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": pred_bytes / (pred_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_bytes("predict_cov_band_kernel")},
        ]
        mfma_ms = solver_ms + cov_ms
        dense_flops = 8.0 * n ** 3 + 6.0 * n * n * m + 4.0 * n * m * m + m ** 3 / 3.0 + 2.0 * n * m
        out = {
            "metric": "EKF updates/sec (state dim, batch) + fp64 MFMA % of peak, 1/2/4/8 GPUs",
            "value": world * B * K / t, "unit": "EKF updates/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": t / K * 1e3,
            "step_ms_device": {"min": step_ms[0], "median": step_ms[K // 2], "max": step_ms[-1],
                               "what": "HIP events on the launch stream between consecutive steps (rank 0)"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: batched Monte Carlo, %d independent filters x %d features per GPU x %d GPU%s "
                                   "(state dim n=%d, measurement dim m=%d), one step = predict + image update of "
                                   "every filter" % ("BASELINE configs[2]" if world == 1 else
                                                     "BASELINE configs[4] (65536 filters over 8 GPUs) per-GPU shard",
                                                     B, nf, world, "" if world == 1 else "s", n, m),
                       "batch_per_gpu": B, "features": nf, "state_dim": n, "meas_dim": m,
                       "update_form": "cholesky gain + joseph covariance", "parallelism": "filters sharded x%d" % world,
                       "pipeline_groups": args.groups},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MATRIX_PEAK_TFLOPS, "traffic": pmc_bytes("gain_solver_kernel"),
                         "kernel": "gain_solver_kernel<%d, border>: one launch per step, %d launches timed with HIP events "
                                   "on the launch stream" % ((m - 1) // 16 if (m - 1) % 16 == 0 else ml // 16,
                                                             prof["gain_solver"][1]),
                         "flops_per_launch": solver_flops, "ms_per_launch": solver_ms,
                         "flops_what": "algorithmic, unpadded: m^3/3 + 2 n m^2 + 12 n m + 6 m (m + 1) per filter",
                         "executed_tile_flops_per_launch": solver_exec_flops,
                         "executed_tile_flops_frac": solver_exec_flops / (solver_ms * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
                         "fp64_mfma_frac_all_mfma_kernels": (solver_flops + cov_flops) / (mfma_ms * 1e-3) / 1e12
                                                            / FP64_MATRIX_PEAK_TFLOPS,
                         "traffic_source": pmc_state},
            "other_kernels": kernels,
            # the dense formulation's work (8n^3 + 6n^2 m + 4nm^2 + m^3/3 + 2nm per update) priced at the measured rate:
            # what an all-GEMM pipeline would have to sustain to match `value`; the sparse Jacobian products skip
            # most of it, so this is an equivalence, not executed FLOP
            "dense_model": {"flops_per_update": dense_flops, "equivalent_tflops": world * B * K / t * dense_flops / 1e12 / world,
                            "equivalent_frac_of_fp64_matrix_peak": B * K / t * dense_flops / 1e12 / FP64_MATRIX_PEAK_TFLOPS},
            "predict_only": {"ms_per_step": t_predict * 1e3, "predicts_per_s": world * B / t_predict},
            "host_boundary": {"entry_points": "slamk_predict + slamk_update (host pointers, PCIe-inclusive, synchronous)",
                              "ms_per_step": t_host * 1e3, "updates_per_s": B / t_host, "steps": hk},
            "stages_ms_per_step": {k: round(v[0] / KS, 4) for k, v in prof_all.items()},
            "stages_what": "HIP events around every stage in a pass of %d steps behind the timed region; inside the timed "
                           "region only the gain solver is timed (roofline.ms_per_launch)" % KS,
            "outputs_sane": sane,
        }
        if churn is not None:
            out["churn_workload"] = churn
        if sharded is not None:
            out.setdefault("also", {})["large_map_sharded"] = sharded
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(st, steps, nf, W, W + K)
        fb.close()
        del steps
        torch.cuda.empty_cache()
        if world == 1 and not args.no_extra:
            out["also"] = {"single_filter": single_filter_latency(torch, dev),
                           "single_filter_block_sequential": single_filter_latency(torch, dev, form=2),
                           "large_map": large_map(torch, dev),
                           "ceilings": ceilings(torch, dev),
                           "sweep": [sweep_point(torch, dev, nf_, B_) for nf_, B_ in
                                     ((8, 4096), (20, 4096), (32, 512), (32, 16384), (36, 1024), (40, 1024), (44, 1024),
                                      (48, 1024), (52, 1024), (60, 1024))],
                           # update_form = 2 (not the default, not what `value` measures): the measurement rows 16 at a
                           # time against the covariance in LDS; same posterior up to rounding, n <= 160 only
                           "block_sequential_form": {
                               "what": "slamk_params.update_form = 2, same step and inputs as the sweep points; the "
                                       "default form (Cholesky gain of the whole measurement vector + Joseph update) "
                                       "is what `value` and `roofline` measure",
                               "points": [sweep_point(torch, dev, nf_, B_, form=2) for nf_, B_ in
                                          ((8, 4096), (20, 4096), (32, 4096), (36, 1024), (40, 1024), (48, 1024))]}}
        print(json.dumps(out), flush=True)
    if sharded_failed:
        sys.stdout.flush()
        os._exit(0)                      # (other ranks may sit in a collective that will not complete)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
