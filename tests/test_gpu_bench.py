"""-m gpu: bench.py's contract line, and its N > 1 code path (one seeded stream cut into contiguous shards, barrier,
max-over-ranks time, the counter all-reduce) with two ranks on the one GPU of the test box (gloo for the collectives:
RCCL does not put two ranks on one device; on an 8-GPU node the driver launches the same script over RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(out):
    lines = [ln for ln in out.decode().splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_line_single_gpu():
    out = subprocess.run([sys.executable, "bench.py", "--reads-per-gpu", "60000", "--ref-len", "4000000", "--steps", "4", "--warmup", "1",
                          "--cpu-seconds", "1"], cwd=ROOT, check=True, capture_output=True, timeout=900).stdout
    d = _last_json(out)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "i32"
    assert d["counters"]["reads"] == 4 * 60000
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["value_e2e"]["value"] > 0 and d["value_two_in_flight"]["value"] > 0
    assert abs(d["value"] - 60000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3


def test_bench_two_ranks_sharded_stream():
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--backend", "gloo", "--one-device", "--total-reads", "300000",
           "--ref-len", "4000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, cwd=ROOT, check=True, capture_output=True, timeout=900, env=env).stdout
    d = _last_json(out)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["reads_per_gpu_per_step"] == 150000
    assert d["counters"]["reads"] == 2 * 300000  # both ranks' shards, both steps, summed by the all-reduce
    assert abs(d["value"] - 300000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
