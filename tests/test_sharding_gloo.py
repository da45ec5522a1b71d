"""N > 1 path on CPU: world_size-2 gloo.  Each rank takes its contiguous read
shard, the (CPU oracle) checker aligns it, the counter vectors are all-reduced
and must equal the counters of the unsharded run; concatenated shard results
must equal the unsharded result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from thermite_amd import capi, refdata, sharding, synth

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_reads, q):
    from oracle import pyoracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = refdata.load_reference(DATA + "/GRCh38-2020-A-chrM.fasta", DATA + "/GRCh38-2020-A-chrM.gtf")
    bases, off, _ = synth.simulate_reads(t, n_reads, 91)
    sb, so = sharding.shard_reads(bases, off, rank, world)
    oix = orc.Index(t)
    r = oix.align_batch(sb, so, capi.CI_OPTS)
    total = sharding.allreduce_counters(r.counters, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (r.offsets, r.alns, r.ops))
    if rank == 0:
        q.put((total, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_partition():
    for n in (0, 1, 7, 100, 500000):
        for w in (1, 2, 3, 8):
            b = [sharding.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_bounds(10, 2, 2)


def test_two_rank_gloo_counters_and_order():
    from oracle import pyoracle as orc

    n_reads, world = 600, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_reads, q)) for r in range(world)]
    for p in procs:
        p.start()
    total, gathered = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    t = refdata.load_reference(DATA + "/GRCh38-2020-A-chrM.fasta", DATA + "/GRCh38-2020-A-chrM.gtf")
    bases, off, _ = synth.simulate_reads(t, n_reads, 91)
    whole = orc.Index(t).align_batch(bases, off, capi.CI_OPTS)
    assert np.array_equal(total[:13], whole.counters[:13])
    # concatenating the shards in rank order restores the unsharded result
    n_alns = sum(len(g[1]) for g in gathered)
    assert n_alns == len(whole.alns)
    counts = np.concatenate([np.diff(g[0].astype(np.int64)) for g in gathered])
    assert np.array_equal(counts, np.diff(whole.offsets.astype(np.int64)))
    for f in ("score", "ystart", "yend", "ref_id", "aln_type", "ops_len"):
        assert np.array_equal(np.concatenate([g[1][f] for g in gathered]), whole.alns[f])
    assert np.array_equal(np.concatenate([g[2] for g in gathered]), whole.ops)


def test_config4_stream_is_cut_into_contiguous_shards():
    """bench.py's N > 1 workload (BASELINE configs[3]): one seeded stream, each rank generates only its own
    contiguous shard; the shards concatenate to the stream whatever the world size"""
    import bench

    t = refdata.load_reference(DATA + "/GRCh38-2020-A-chrM.fasta", DATA + "/GRCh38-2020-A-chrM.gtf")
    old = bench.STREAM_CHUNK
    bench.STREAM_CHUNK = 700  # several chunks per shard, shard bounds inside chunks
    try:
        total = 5000
        whole, off = bench.stream_reads(synth, t, 0, total, 91)
        assert len(whole) == total * 91 and off[-1] == total * 91
        for world in (2, 3, 8):
            parts = []
            for rank in range(world):
                b, e = sharding.shard_bounds(total, rank, world)
                pb, po = bench.stream_reads(synth, t, b, e, 91)
                assert len(po) == e - b + 1
                parts.append(pb)
            assert np.array_equal(np.concatenate(parts), whole)
    finally:
        bench.STREAM_CHUNK = old
