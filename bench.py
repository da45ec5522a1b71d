#!/usr/bin/env python3
"""bench.py -- headline measurement of the seed-and-extend hot path on MI355X.

Metric (BASELINE.json): aligned reads/sec for synthetic 91 bp reads.  One "step"
= one pass of the hot path (seed kernels -> plan -> scan -> extend kernel ->
compaction) over one batch of reads that is already resident in HBM.

Workloads
  N = 1 (default)   BASELINE configs[2]: 500 000 synthetic 91 bp reads per batch against a
                    chr21-sized synthetic reference (46 709 983 bp; the real chr21 FASTA/GTF and
                    the pbmc10k reads are missing blobs in the reference checkout), flags of the
                    reference's chr21 run `-k20 -s0 --intron-mode` (reference data/Makefile:39).
                    The timed steps rotate over 4 DISTINCT resident batches (4 aligners on the
                    shared index), so no step replays the batch the caches have just seen.
  N > 1 (default)   BASELINE configs[3]: ONE seeded stream of 50 000 000 such reads cut into
                    contiguous shards with thermite_amd.sharding.shard_bounds (6.25 M reads per GPU
                    at N = 8; concatenating the shards restores input order, reference
                    src/aligner.rs:54-115); every rank generates only its own shard.  Index
                    replicated per GPU; the only collective is one all-reduce (RCCL) of the
                    counter vector.  "scaling": "strong" (the total is fixed).
  --reads-per-gpu R weak scaling instead: R reads per rank, each rank its own stream.

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline      -- extend kernel: algorithmic bytes per launch / mean launch time (HIP events on the
                   aligner's stream, inside the timed region), HBM traffic from the committed PMC profile;
  roofline_valu -- the resource that actually binds the kernel: vector-ALU issue (SQ counters from the
                   committed profile) and DP cells / s (live);
  value_e2e     -- reads/s including host<->device transfers (upload + run + fetch of distinct batches,
                   two aligners on two host threads so that transfers overlap kernels); never `value`;
  cpu_baseline  -- the CPU oracle (a port of the reference algorithm; the Rust reference cannot be
                   built here) timed on a bounded sample of the same reads on this box's host cores.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CONFIG4_TOTAL_READS = 50_000_000
STREAM_CHUNK = 250_000  # reads per chunk of the global seeded stream (chunk c = simulate_reads(stream=1000 + c))


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def stream_reads(synth, tables, begin, end, L):
    """Reads [begin, end) of the global seeded stream, generated chunk by chunk (only the chunks the range touches)."""
    parts = []
    c = begin // STREAM_CHUNK
    while c * STREAM_CHUNK < end:
        b, _, _ = synth.simulate_reads(tables, STREAM_CHUNK, L, sub_rate=0.01, indel_rate=0.001, stream=1000 + c)
        lo = max(begin, c * STREAM_CHUNK) - c * STREAM_CHUNK
        hi = min(end, (c + 1) * STREAM_CHUNK) - c * STREAM_CHUNK
        parts.append(b[lo * L: hi * L])
        c += 1
    bases = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    n = end - begin
    return bases, (np.arange(n + 1, dtype=np.uint64) * np.uint64(L)).astype("<u8")


def csrc_hash():
    """hash of the kernel and host sources the library is built from: ties a committed counter profile to a build"""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "thermite_amd", "csrc")
    for f in sorted(os.listdir(d)):
        # the device code and what it is built with (the host-side file I/O, index construction and suffix sorting do not
        # change what a kernel does)
        if (f.endswith((".hip", ".h")) and not f.startswith("io_")) or f == "Makefile":
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def profile_entry(name, match):
    """(summary, reason): a committed profile summary (profiles/*.json) if it was taken on this workload AND on the sources
    this tree is built from (its csrc_hash), else (None, why not).  Counters are collected in separate rocprofv3 passes
    (tools/profile_r03.sh): they cannot be measured inside the timed region, and a stale profile must not pass for one."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, "no committed profile " + name
    try:
        js = json.load(open(path))
    except Exception:
        return None, "unreadable profile " + name
    js = js if isinstance(js, list) else [js]  # one entry per workload
    js = [e for e in js if all(e.get(k) == v for k, v in match.items())]
    if not js:
        return None, "no committed profile of this workload"
    j = js[-1]
    if j.get("csrc_hash") != csrc_hash():
        return None, "committed profile was taken on other sources (csrc_hash %s, tree %s)" % (j.get("csrc_hash"), csrc_hash())
    return j, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)  # 0.27 s of timed region at N = 1 (12 steps were 0.07 s)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--reads-per-gpu", type=int, default=None, help="weak scaling: this many reads per rank (default at N = 1: 500 000)")
    ap.add_argument("--total-reads", type=int, default=None, help="strong scaling: one stream of this many reads sharded over the ranks "
                                                                    "(default at N > 1: 50 000 000, BASELINE configs[3])")
    ap.add_argument("--batches", type=int, default=4, help="distinct resident batches the timed steps rotate over (weak mode)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="batches in flight: 2 launches step i + 1 (another aligner, another HIP stream) before step i is waited for, "
                         "so that the seed kernels of one batch fill the tail of the extend kernel of the other")
    ap.add_argument("--read-len", type=int, default=91)
    ap.add_argument("--ref-len", type=int, default=int(os.environ.get("THM_BENCH_REF_LEN", "0")) or None)
    ap.add_argument("--opts", choices=["ci", "default"], default="ci")
    ap.add_argument("--workload", choices=["chr21syn", "chrM"], default="chr21syn",
                    help="chr21syn: the headline workload (BASELINE configs[2]); chrM: configs[1], the real chrM FASTA/GTF "
                         "that ship with the reference's data/ (copied to tests/golden/data)")
    ap.add_argument("--percent", type=float, default=None, help="override min_aln_score_percent (config 5: 0.574 at 150 bp = band +-64)")
    ap.add_argument("--wide", action="store_true", help="64-bit text coordinates inside the index (the path a GRCh38-sized text takes)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend; gloo + --one-device runs the N > 1 code path on a one-GPU box (tests)")
    ap.add_argument("--one-device", action="store_true", help="every rank on device 0 (tests of the N > 1 path on a one-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the extra figures (value_e2e, value_two_in_flight): profiling passes")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dump-digest", default=None,
                    help="write PATH.<rank>.json: SHA-256 of this rank's alignment records and op streams per batch (tests compare them "
                         "with the CPU oracle's for the same shard)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live

    from thermite_amd import capi, sharding, synth

    # ---------------- reference + index ----------------
    t0 = time.time()
    real_ref = None
    if args.workload == "chrM":
        from thermite_amd import refdata

        d = os.path.join(ROOT, "tests", "golden", "data")
        tables = refdata.load_reference(d + "/GRCh38-2020-A-chrM.fasta", d + "/GRCh38-2020-A-chrM.gtf")
        ref_len = int(tables["refs"][0]["len"])
        tag = "chrM"
    elif os.environ.get("THM_CHR21_FASTA") and os.environ.get("THM_CHR21_GTF"):
        # BASELINE configs[2] on the real inputs, when the caller has them (they are missing blobs in the reference checkout)
        from thermite_amd import refdata

        tables = refdata.load_reference(os.environ["THM_CHR21_FASTA"], os.environ["THM_CHR21_GTF"])
        ref_len = int(tables["refs"][0]["len"])
        tag = "user_%d_%d" % (ref_len, len(tables["text"]))
        real_ref = os.path.basename(os.environ["THM_CHR21_FASTA"])
    else:
        ref_len = args.ref_len or synth.CHR21_LEN
        tables = synth.synth_reference(length=ref_len)
        tag = "%d_%x" % (ref_len, synth.SEED)
    # (no suffix array supplied: every rank's library sorts the suffixes on its own GPU, csrc/sa_gpu.hip -- 0.4 s here)
    index = capi.Index(tables, wide=args.wide)
    t_index = time.time() - t0
    log(rank, "reference: %s %d bp, text n=%d, %d transcripts, %d exons; index (%d-byte coordinates) in %.1fs" % (
        args.workload, ref_len, len(tables["text"]), len(tables["txs"]), len(tables["exons"]), index.coord_bytes, time.time() - t0))
    opts = dict(capi.CI_OPTS if args.opts == "ci" else capi.DEFAULT_OPTS)
    if args.percent is not None:
        opts["min_aln_score_percent"] = args.percent
    L = args.read_len

    # ---------------- reads: this rank's batches ----------------
    t0 = time.time()
    strong = args.reads_per_gpu is None and (args.total_reads is not None or world > 1)
    if strong:
        total = args.total_reads or CONFIG4_TOTAL_READS
        b, e = sharding.shard_bounds(total, rank, world)
        batches = [stream_reads(synth, tables, b, e, L)]
        reads_this_rank = e - b
        desc_reads = "%d synthetic %d bp reads (one seeded stream) in %d contiguous shard(s)" % (total, L, world)
    else:
        per = args.reads_per_gpu or 500000
        nb = max(1, args.batches)
        batches = [synth.simulate_reads(tables, per, L, sub_rate=0.01, indel_rate=0.001, stream=100 + 16 * rank + k)[:2] for k in range(nb)]
        reads_this_rank = per
        desc_reads = "%d synthetic %d bp reads per GPU per step, %d distinct resident batches in rotation" % (per, L, nb)
    NB = len(batches)
    t_reads = time.time() - t0
    log(rank, "reads: %s; rank 0 holds %d x %d reads (generated in %.1fs)" % (desc_reads, NB, reads_this_rank, time.time() - t0))
    aligners = [capi.Aligner(index, opts, device=local_rank) for _ in range(NB)]
    for a, (bases, offsets) in zip(aligners, batches):
        a.upload(bases, offsets)  # inputs resident in HBM before the timed region
        a.run()                   # sizes the device pools (replays on overflow), untimed
        a.sync()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # per-rank set-up times (index build / load, read generation): rank 0 reports them all (a slow rank shows before the run)
    setup = torch.tensor([t_index, t_reads], dtype=torch.float64, device=cdev)
    setup_all = [setup]
    if world > 1:
        setup_all = [torch.zeros_like(setup) for _ in range(world)]
        dist.all_gather(setup_all, setup)
    per_rank_setup = [{"rank": r, "index_s": round(float(v[0]), 2), "reads_s": round(float(v[1]), 2)} for r, v in enumerate(setup_all)]
    log(rank, "set-up seconds per rank (index, reads): " + ", ".join("%d: %.1f / %.1f" % (d["rank"], d["index_s"], d["reads_s"]) for d in per_rank_setup))
    if args.dump_digest:
        import hashlib

        digs = []
        for a in aligners:
            g = a.fetch()
            h = hashlib.sha256()
            h.update(np.ascontiguousarray(g.offsets).tobytes())
            for f in capi.ALN_DT.names:
                if f != "pad_":
                    h.update(np.ascontiguousarray(g.alns[f]).tobytes())
            h.update(np.ascontiguousarray(g.ops).tobytes())
            digs.append({"reads": int(g.n_reads), "alignments": int(len(g.alns)), "sha256": h.hexdigest()})
        json.dump({"rank": rank, "world": world, "batches": digs}, open("%s.%d.json" % (args.dump_digest, rank), "w"))

    # ---------------- warmup ----------------
    for i in range(args.warmup):
        aligners[i % NB].run()
        aligners[i % NB].sync()
    for a in aligners:
        a.reset_counters()

    # ---------------- timed region: exactly K steps ----------------
    K = args.steps
    stage_ms = {k: 0.0 for k in capi.TIMING_NAMES}
    step_wall_ms, step_ext_ms = [], []  # per step: host wall time between completions, extend stage (HIP events)
    barrier()
    t0 = time.perf_counter()
    t_prev = t0
    depth = max(1, min(args.inflight, NB))
    for i in range(K + depth - 1):
        if i < K:
            aligners[i % NB].run()
        j = i - (depth - 1)
        if j >= 0:
            a = aligners[j % NB]
            a.sync()  # stream sync + pool-overflow check; HIP-event stage times of this launch
            t_now = time.perf_counter()
            step_wall_ms.append((t_now - t_prev) * 1e3)
            t_prev = t_now
            tm_ = a.timings()
            step_ext_ms.append(tm_["extend"])
            for k, v in tm_.items():
                stage_ms[k] += v
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---------------- the one collective: counter all-reduce ----------------
    cnt_local = np.zeros(capi.N_COUNTERS, np.uint64)
    for a in aligners:
        cnt_local += a.counters()
    cnt = torch.from_numpy(cnt_local.astype(np.int64)).to(cdev)
    if world > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    cnt = cnt.cpu().numpy().astype(np.uint64)
    c = dict(zip(capi.COUNTER_NAMES, [int(v) for v in cnt]))

    if strong:
        total_reads = (args.total_reads or CONFIG4_TOTAL_READS) * K
    else:
        total_reads = reads_this_rank * world * K
    assert c["reads"] == total_reads, (c["reads"], total_reads)
    value = total_reads / elapsed

    # ---------------- roofline of the dominant kernel (extend) ----------------
    k_seed = int(opts["min_seed_len"])
    lc = dict(zip(capi.COUNTER_NAMES, [int(v) for v in cnt_local]))  # this rank's launches
    per_launch = lambda name: lc[name] / K
    n_r = reads_this_rank
    ext_bytes = (n_r * L + 12 * per_launch("smems") + 4 * per_launch("hits") + per_launch("window_bytes")
                 + 112 * per_launch("alns") + per_launch("op_bytes"))
    seed_bytes = n_r * L + n_r * max(L - k_seed + 1, 0) * 16 + 12 * per_launch("smems")
    ext_ms = stage_ms["extend"] / K
    achieved = ext_bytes / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else 0.0
    profiled_kind = not strong and args.workload == "chr21syn" and real_ref is None
    match = ({"reads_per_gpu": n_r, "ref_len": ref_len, "opts": args.opts, "read_len": L, "percent": args.percent, "wide": bool(args.wide)}
             if profiled_kind else {"reads_per_gpu": -1})
    tj, tj_why = profile_entry("pmc_traffic.json", match)
    sq, sq_why = profile_entry("sq_counters.json", match)
    spread = lambda v: {"min": round(float(np.min(v)), 4), "median": round(float(np.median(v)), 4), "max": round(float(np.max(v)), 4)}
    roofline = {
        "kernel": "extend_kernel", "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": tj.get("extend_kernel_hbm_bytes_per_launch") if tj else None,
        "traffic_source": ("profiles/pmc_traffic.json, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on these sources (csrc_hash %s)" % tj.get("csrc_hash")) if tj else tj_why,
        "algorithmic_bytes_per_launch": int(ext_bytes), "kernel_ms": round(ext_ms, 4),
        "algorithmic_bytes_per_read_whole_path": round((ext_bytes + seed_bytes - n_r * L) / n_r, 1),
        "stage_ms": {k: round(v / K, 4) for k, v in stage_ms.items()},
        "kernel_ms_per_step": spread(step_ext_ms),
        "note": "integer DP + random index probes: bound by HBM latency / VALU issue, not HBM bandwidth (SURVEY.md F7); see roofline_valu",
    }
    # the binding resource: vector-ALU issue.  busy = SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CU_CYCLES-equivalent from the committed
    # SQ-counter profile of this workload (profiles/sq_counters.json); DP cells per second measured here.
    roofline_valu = {
        "kernel": "extend_kernel", "bound": "valu-issue",
        "dp_cells_per_s": round(per_launch("dp_cells") / (ext_ms * 1e-3), 1) if ext_ms > 0 else None,
        "dp_cells_per_read": round(per_launch("dp_cells") / max(n_r, 1), 1),
        "dp_cols_per_read": round(per_launch("dp_cols") / max(n_r, 1), 2),
        "valu_busy_frac": sq.get("extend_valu_busy_frac") if sq else None,
        "valu_insts_per_read": sq.get("extend_valu_insts_per_read") if sq else None,
        "source": sq.get("source") if sq else sq_why,
    }

    # ---------------- PCIe-inclusive rate (not `value`): upload + run + fetch of distinct batches ----------------
    value_e2e = None
    if not args.no_e2e and not strong and NB >= 2:
        n_thr = 2
        per_thread = max(2, K // 2)

        def worker(t):
            a = aligners[t]
            for j in range(per_thread):
                bases, offsets = batches[(t + n_thr * j) % NB]
                a.align_batch(bases, offsets, copy=False)  # H2D of the reads, kernels, D2H of alignments + op streams into the
                # aligner's pinned result set; the views are what a C caller gets (no further copy into numpy arrays)

        barrier()
        t1 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(t,)) for t in range(n_thr)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        if world > 1:
            tm = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt = float(tm.item())
        value_e2e = {"value": round(n_thr * per_thread * reads_this_rank * world / dt, 1), "unit": "reads/s",
                     "what": "host buffers in, host views out: H2D reads + all kernels + D2H alignments and op streams, "
                             "%d batches on %d aligners / host threads per GPU (transfers overlap kernels)" % (n_thr * per_thread, n_thr)}

    # ---------------- two batches in flight (not `value`): step i + 1 is launched (another aligner, another HIP stream)
    # before step i is waited for, so that the tail of one batch's extend kernel is filled by the other batch ----------------
    value_two_in_flight = None
    if not args.no_e2e and NB >= 2 and not strong:
        barrier()
        t1 = time.perf_counter()
        for i in range(K + 1):
            if i < K:
                aligners[i % NB].run()
            if i >= 1:
                aligners[(i - 1) % NB].sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        if world > 1:
            tm = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt = float(tm.item())
        value_two_in_flight = {"value": round(K * reads_this_rank * world / dt, 1), "unit": "reads/s", "ms_per_step": round(dt / K * 1e3, 4),
                               "what": "the same %d steps with two batches in flight per GPU (two aligners, two HIP streams)" % K}

    # ---------------- CPU baseline (rank 0, N = 1 only) ----------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as orc

        bases, offsets = batches[0]
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = min(cores, 16)  # a one-GPU box owns a 16-thread share of the host
        t1 = time.time()
        oix = orc.Index(tables, sa=index.suffix_array())
        log(rank, "oracle index (BWT/Occ/sampled SA) in %.1fs; timing on %d host threads" % (time.time() - t1, cores))
        probe = min(2000 * cores, reads_this_rank)
        t1 = time.perf_counter()
        oix.align_batch(bases[: probe * L], offsets[: probe + 1], opts, n_threads=cores)
        rate = probe / (time.perf_counter() - t1)
        sample = int(min(reads_this_rank, max(probe, rate * args.cpu_seconds)))
        t1 = time.perf_counter()
        r = oix.align_batch(bases[: sample * L], offsets[: sample + 1], opts, n_threads=cores)
        dt = time.perf_counter() - t1
        t1 = time.perf_counter()
        n1 = min(sample, 20000)
        oix.align_batch(bases[: n1 * L], offsets[: n1 + 1], opts, n_threads=1)
        dt1 = time.perf_counter() - t1
        cpu_baseline = {
            "value": round(sample / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "first %d of the %d reads of rank 0's first batch, %.1f s, %d threads" % (sample, reads_this_rank, dt, cores),
            "value_1thread": round(n1 / dt1, 1),
            "note": "CPU restatement of the reference algorithm (oracle/), not the reference Rust binary",
            "aligned_frac": round(float(r.counters[1]) / max(int(r.counters[0]), 1), 4),
        }

    if rank == 0:
        refdesc = ("chr21-sized synthetic transcriptome" if args.workload == "chr21syn" else "GRCh38-2020-A chrM (real FASTA/GTF)") if real_ref is None \
            else "user-supplied reference %s (THM_CHR21_FASTA / THM_CHR21_GTF)" % real_ref
        out = {
            "metric": "aligned reads/sec (%d bp)" % L, "value": round(value, 1), "unit": "reads/s", "n_gpus": world,
            "steps": K, "warmup": args.warmup, "ms_per_step": round(elapsed / K * 1e3, 4), "ms_per_step_spread": spread(step_wall_ms),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "i32", "data": "synthetic" if real_ref is None else "synthetic reads from a user-supplied reference",
            "config": {
                "workload": "%s vs %s (%d bp, %d tx), flags %s%s" % (
                    desc_reads, refdesc, ref_len, len(tables["txs"]),
                    "-k20 -s0 --intron-mode" if args.opts == "ci" else "defaults (-k20 -s0.66)",
                    "" if args.percent is None else " with -s%g" % args.percent),
                "reads_per_gpu_per_step": reads_this_rank, "read_len": L, "ref_len": ref_len, "opts": args.opts, "percent": args.percent,
                "coord_bytes": index.coord_bytes, "batches_in_flight": max(1, min(args.inflight, NB)),
                "parallelism": "reads sharded over %d GPU(s), index replicated, 1 counter all-reduce" % world,
            },
            "per_rank_setup_s": per_rank_setup,
            "roofline": roofline,
            "roofline_valu": roofline_valu,
            "value_is": "batches resident in HBM before the timed region (the task's contract); value_e2e is the rate with host<->device "
                        "transfers inside (BASELINE.md section 2 names both)",
            "value_e2e": value_e2e,
            "value_two_in_flight": value_two_in_flight,
            "cpu_baseline": cpu_baseline,
            "counters": {k: c[k] for k in ("reads", "aligned", "unmapped", "alns", "exonic", "intronic", "intergenic",
                                           "smems", "hits", "swg_calls", "dp_cells")},
        }
        print(json.dumps(out), flush=True)
    for a in aligners:
        a.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
