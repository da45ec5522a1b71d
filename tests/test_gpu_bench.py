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


def test_bench_two_ranks_sharded_stream(tmp_path):
    """the N > 1 path end to end: both ranks' ALIGNMENTS (not only the read counter) against the CPU oracle on the same
    shards of the one seeded stream"""
    port = _free_port()
    dig = str(tmp_path / "digest")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--backend", "gloo", "--one-device", "--total-reads", "300000",
           "--ref-len", "4000000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--dump-digest", dig]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, cwd=ROOT, check=True, capture_output=True, timeout=900, env=env).stdout
    d = _last_json(out)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["reads_per_gpu_per_step"] == 150000
    assert d["counters"]["reads"] == 2 * 300000  # both ranks' shards, both steps, summed by the all-reduce
    assert abs(d["value"] - 300000 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    assert [x["rank"] for x in d["per_rank_setup_s"]] == [0, 1]
    # the oracle on each rank's shard
    import hashlib

    import numpy as np

    sys.path.insert(0, ROOT)
    import bench
    from oracle import pyoracle as orc
    from thermite_amd import capi, sharding, synth

    tables = synth.synth_reference(length=4000000)
    oix = orc.Index(tables, sa=capi.build_suffix_array(tables["text"]))
    for rank in (0, 1):
        got = json.load(open("%s.%d.json" % (dig, rank)))
        b, e = sharding.shard_bounds(300000, rank, 2)
        bases, off = bench.stream_reads(synth, tables, b, e, 91)
        r = oix.align_batch(bases, off, capi.CI_OPTS, n_threads=16)
        h = hashlib.sha256()
        h.update(np.ascontiguousarray(r.offsets).tobytes())
        for f in capi.ALN_DT.names:
            if f != "pad_":
                h.update(np.ascontiguousarray(r.alns[f]).tobytes())
        h.update(np.ascontiguousarray(r.ops).tobytes())
        assert got["batches"][0]["reads"] == e - b and got["batches"][0]["alignments"] == len(r.alns)
        assert got["batches"][0]["sha256"] == h.hexdigest(), "rank %d: alignments differ from the oracle" % rank
