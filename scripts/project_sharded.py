#!/usr/bin/env python3
"""Per-rank compute time of the row-block sharded large-map update for world sizes 1, 2, 4, 8, measured on ONE GPU:
rank 0 of a g-rank world runs its share of every stage through slamk_sharded_*; the all-gather is a callback that
moves nothing (the other ranks' blocks keep whatever they held: the numbers are meaningless, the launch sequence and
the shapes, hence the times, are those of a real rank) and adds up the bytes a rank would receive.

    python scripts/project_sharded.py [--nf 2000]

Prints one JSON line per world size: ms per update on the rank, bytes received per update, all-gather calls.
"""
import argparse
import ctypes as C
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
    ap.add_argument("--worlds", type=int, nargs="+", default=[1, 2, 4, 8])
    args = ap.parse_args()
    import torch
    from slam_amd import ekf, synth, _lib
    dev = torch.device("cuda", 0)
    torch.zeros(1, device=dev)
    L = _lib.lib()
    nf = args.nf
    st = synth.gen_stream(seed=2, nf=nf, steps=5, churn=False, image_every=1)
    s0 = st.steps[0]
    first = [(int(s0.ids[i]), s0.src[i, 0], s0.src[i, 1], s0.refl[i, 0], s0.refl[i, 1]) for i in range(nf)]
    AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p)
    for g in args.worlds:
        fb = ekf.FilterBatch(max_features=nf, batch=1, device=0)
        fb.set_state(0, X=st.init_pos)
        fb.frame(0, first, s0.quat)
        stat = {"bytes": 0, "calls": 0}

        def gather(ctx, buf, bytes_per_rank, rk, wd, strm):
            stat["bytes"] += (wd - 1) * bytes_per_rank
            stat["calls"] += 1
            return 0
        cb = AG(gather)
        s = C.c_void_p()
        rc = L.slamk_sharded_create(fb.h, 0, g, None, C.cast(cb, C.c_void_p) if g > 1 else None, None, C.byref(s))
        assert rc == 0, rc
        stream = torch.cuda.Stream(device=dev)
        P0 = torch.from_numpy(fb.get_cov(0)).to(dev)
        torch.cuda.synchronize(dev)
        assert L.slamk_sharded_load_cov_dev(s, P0.data_ptr(), P0.stride(0), stream.cuda_stream) == 0
        t = lambda v: torch.tensor(np.asarray(v, dtype=np.float64), device=dev)
        times = []
        for k in range(1, 4):
            sk = st.steps[k]
            q, w, a, d = t(sk.quat), t(sk.ang), t(sk.acc), t([sk.t - st.steps[k - 1].t])
            applied = torch.zeros(1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize(dev)
            stat["bytes"] = 0
            stat["calls"] = 0
            t0 = time.perf_counter()
            L.slamk_frame_begin_dev(fb.h, C.c_void_p(stream.cuda_stream))
            rc1 = L.slamk_sharded_predict_dev(s, q.data_ptr(), w.data_ptr(), a.data_ptr(), d.data_ptr(), stream.cuda_stream)
            rc2 = L.slamk_sharded_update_dev(s, q.data_ptr(), None, None, None, applied.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            times.append((time.perf_counter() - t0) * 1e3)
            assert rc1 == 0 and rc2 == 0, (rc1, rc2, fb.last_error() if hasattr(fb, "last_error") else "")
        print(json.dumps({"world": g, "ms_per_update_rank0": min(times[1:]), "all": times,
                          "bytes_received_per_update": stat["bytes"], "allgather_calls": stat["calls"]}), flush=True)
        L.slamk_sharded_destroy(s)
        fb.close()


if __name__ == "__main__":
    main()
