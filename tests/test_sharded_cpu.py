"""CPU, world_size 2 over gloo: the row-block sharded large-map sequence (tests/sharded_rehearsal.py) with
the oracle as local backend must reproduce the oracle's own single-process run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nf, steps, panel, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as orc
        from slam_amd import synth
        from sharded_rehearsal import ShardedLargeMapFilter
        from cpu_backend import OracleBackend
        import helpers
        st = synth.gen_stream(seed=11, nf=nf, steps=steps, churn=False)
        inp = helpers.stream_inputs(st, 7)
        f = orc.Filter()
        f.X = st.init_pos
        be = OracleBackend(f)
        be.frame(inp[0][5], inp[0][1])
        sh = ShardedLargeMapFilter(be, rank, world, panel=panel)
        sh.load_cov(torch.from_numpy(f.P))
        for k in range(1, steps):
            u, quat, ang, acc, dt, matches = inp[k]
            if u & orc.IMG:
                be.frame(matches, quat)          # no churn: the layout does not change
            sh.predict(quat, ang, acc, dt)
            if u & orc.IMG:
                sh.update(quat)
        P = sh.gather_cov().numpy()
        if rank == 0:
            np.save(out, np.concatenate([P.ravel(), f.state_vector()]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nf,panel", [(2, 7, 512), (2, 13, 16), (3, 10, 32), (3, 11, 16)])
def test_sharded_matches_oracle(tmp_path, world, nf, panel):
    """panel < m: several panels in the left-looking factorisation (rows of a panel dealt to the ranks, all-gather per
    panel); block boundaries that cut a feature's row triple (the rows of T exchanged in the propagation)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from slam_amd import synth
    steps = 8
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), nf, steps, panel, out), nprocs=world, join=True)
    res = np.load(out)
    st = synth.gen_stream(seed=11, nf=nf, steps=steps, churn=False)
    so, f = helpers.run_oracle(st)
    n = 9 + 3 * nf
    P = res[:n * n].reshape(n, n)
    assert helpers.rel_err(P, f.P) < 1e-9
    assert helpers.rel_err(res[n * n:], f.state_vector()) < 1e-9
