"""GPU: bench.py end to end at a small size: the one-line JSON contract on one GPU, and the rank-sharded path
(one process per rank, filters sharded with no data-path collective, barrier + MAX of times, final gather)
with two ranks on the one GPU there is, collectives over gloo; the same run also times the row-block sharded
large-map filter over its ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [ln for ln in out.strip().split("\n") if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_single_gpu_contract():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch-per-gpu", "128", "--no-cpu",
                        "--no-extra"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["outputs_sane"] and d["value"] > 0
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
    assert d["step_ms_device"]["min"] <= d["step_ms_device"]["median"] <= d["step_ms_device"]["max"]


def test_bench_rank_sharded_two_ranks_over_gloo():
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, SLAM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--batch-per-gpu", "96", "--no-cpu", "--no-extra", "--sharded-nf", "60"],
                       cwd=ROOT, capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["outputs_sane"]
    # the large-map leg of a multi-GPU run: one filter (here 60 features), its covariance sharded by row blocks over the
    # two ranks through slamk_sharded_*, the library's all-gathers routed through its callback (no RCCL between two
    # processes on one GPU)
    sh = d["also"]["large_map_sharded"]
    assert sh["cov_sane"] and sh["ms_per_update"] > 0 and sh["rows_per_rank"] == 96 and "over 2 GPUs" in sh["workload"]
    assert sh["bytes_received_per_rank_per_update"] > 0
    assert d["config"]["batch_per_gpu"] == 96 and "x 2 GPUs" in d["config"]["workload"]
    # whole-job value: both ranks' filters over the slowest rank's time
    assert abs(d["value"] - 2 * 96 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-6


def test_bench_line_survives_a_failed_large_map_leg():
    """The row-block sharded large-map leg of a multi-GPU run is a side measurement with collectives of its own: when it
    does not come back (here: a time limit no run can meet), rank 0 still prints the line of the batched workload, the
    leg's entry carries the reason, and the job ends with status 0."""
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, SLAM_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", SLAM_BENCH_SHARDED_TIMEOUT_S="0.001")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--batch-per-gpu", "96", "--no-cpu", "--no-extra", "--sharded-nf", "60"],
                       cwd=ROOT, capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["outputs_sane"] and d["value"] > 0
    assert "error" in d["also"]["large_map_sharded"]
