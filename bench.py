#!/usr/bin/env python3
"""bench.py -- headline measurement of the seed-and-extend hot path on MI355X.

Metric (BASELINE.json): aligned reads/sec for synthetic 91 bp reads.  One "step"
= one pass of the hot path (seed kernel -> scan -> extend kernel -> compaction)
over one batch of reads that is already resident in HBM.  Workload at every N:
`--reads-per-gpu` (default 500 000, BASELINE configs[2]/[3]) synthetic 91 bp
reads per GPU against a chr21-sized synthetic reference (46 709 983 bp; the real
chr21 FASTA/GTF and pbmc10k reads are missing blobs in the reference checkout),
aligned with the flags the reference uses for its chr21 run,
`-k20 -s0 --intron-mode` (reference data/Makefile:39).  Reads shard
embarrassingly: each rank aligns its own reads against its own copy of the
index; the only collective is one all-reduce (RCCL) of the counter vector.

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     -- extend kernel: algorithmic bytes per launch / mean launch time
                  (HIP events on the aligner's stream, inside the timed region)
  cpu_baseline -- the CPU oracle (a port of the reference algorithm; the Rust
                  reference cannot be built here) timed on a bounded sample of
                  the same reads on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def load_suffix_array(capi, tables, rank, world, dist, tag):
    """Rank 0 builds the suffix array once and shares it through /tmp."""
    path = "/tmp/thm_bench_sa_%s.npy" % tag
    if rank == 0 and not os.path.exists(path):
        sa = capi.build_suffix_array(tables["text"])
        tmp = path + ".%d.tmp.npy" % os.getpid()
        np.save(tmp, sa)
        os.replace(tmp, path)
    if world > 1:
        dist.barrier()
    return np.load(path, mmap_mode="r")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads-per-gpu", type=int, default=500000)
    ap.add_argument("--read-len", type=int, default=91)
    ap.add_argument("--ref-len", type=int, default=int(os.environ.get("THM_BENCH_REF_LEN", "0")) or None)
    ap.add_argument("--opts", choices=["ci", "default"], default="ci")
    ap.add_argument("--workload", choices=["chr21syn", "chrM"], default="chr21syn",
                    help="chr21syn: the headline workload (BASELINE configs[2]); chrM: configs[1], the real chrM FASTA/GTF "
                         "that ship with the reference's data/ (copied to tests/golden/data)")
    ap.add_argument("--percent", type=float, default=None, help="override min_aln_score_percent (config 5: 0.574 at 150 bp = band +-64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from thermite_amd import capi, synth

    # ---------------- workload ----------------
    t0 = time.time()
    if args.workload == "chrM":
        from thermite_amd import refdata

        d = os.path.join(ROOT, "tests", "golden", "data")
        tables = refdata.load_reference(d + "/GRCh38-2020-A-chrM.fasta", d + "/GRCh38-2020-A-chrM.gtf")
        ref_len = int(tables["refs"][0]["len"])
        tag = "chrM"
    else:
        ref_len = args.ref_len or synth.CHR21_LEN
        tables = synth.synth_reference(length=ref_len)
        tag = "%d_%x" % (ref_len, synth.SEED)
    sa = load_suffix_array(capi, tables, rank, world, dist, tag)
    index = capi.Index(tables, sa=sa)
    log(rank, "reference: %s %d bp, text n=%d, %d transcripts, %d exons; index in %.1fs" % (
        args.workload, ref_len, len(tables["text"]), len(tables["txs"]), len(tables["exons"]), time.time() - t0))
    opts = dict(capi.CI_OPTS if args.opts == "ci" else capi.DEFAULT_OPTS)
    if args.percent is not None:
        opts["min_aln_score_percent"] = args.percent
    L = args.read_len
    bases, offsets, _ = synth.simulate_reads(tables, args.reads_per_gpu, L, sub_rate=0.01, indel_rate=0.001,
                                             stream=100 + rank)
    aligner = capi.Aligner(index, opts, device=local_rank)
    aligner.upload(bases, offsets)  # inputs resident in HBM before the timed region

    # ---------------- warmup (also sizes the device pools) ----------------
    for _ in range(max(args.warmup, 1)):
        aligner.run()
        aligner.sync()
    aligner.reset_counters()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- timed region: exactly K steps ----------------
    stage_ms = {k: 0.0 for k in capi.TIMING_NAMES}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        aligner.run()
        aligner.sync()  # stream sync + pool-overflow check; HIP-event stage times of this launch
        for k, v in aligner.timings().items():
            stage_ms[k] += v
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---------------- the one collective: counter all-reduce ----------------
    cnt_local = aligner.counters()
    cnt = torch.from_numpy(cnt_local.astype(np.int64)).to(dev)
    if world > 1:
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    cnt = cnt.cpu().numpy().astype(np.uint64)
    c = dict(zip(capi.COUNTER_NAMES, [int(v) for v in cnt]))

    K = args.steps
    total_reads = args.reads_per_gpu * world * K
    assert c["reads"] == total_reads, (c["reads"], total_reads)
    value = total_reads / elapsed

    # ---------------- roofline of the dominant kernel (extend) ----------------
    k_seed = int(opts["min_seed_len"])
    lc = dict(zip(capi.COUNTER_NAMES, [int(v) for v in cnt_local]))  # this rank's launches
    per_launch = lambda name: lc[name] / K
    n_r = args.reads_per_gpu
    ext_bytes = (n_r * L + 12 * per_launch("smems") + 4 * per_launch("hits") + per_launch("window_bytes")
                 + 112 * per_launch("alns") + per_launch("op_bytes"))
    seed_bytes = n_r * L + n_r * max(L - k_seed + 1, 0) * 16 + 12 * per_launch("smems")
    ext_ms = stage_ms["extend"] / K
    achieved = ext_bytes / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if (tj.get("reads_per_gpu") == n_r and tj.get("ref_len") == ref_len and tj.get("opts") == args.opts
                    and args.workload == "chr21syn" and args.percent is None and L == 91):
                traffic = tj.get("extend_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {
        "kernel": "extend_kernel", "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
        "algorithmic_bytes_per_launch": int(ext_bytes), "kernel_ms": round(ext_ms, 4),
        "algorithmic_bytes_per_read_whole_path": round((ext_bytes + seed_bytes - n_r * L) / n_r, 1),
        "stage_ms": {k: round(v / K, 4) for k, v in stage_ms.items()},
        "note": "integer DP + random index probes: bound by HBM latency / VALU, not HBM bandwidth (SURVEY.md F7)",
    }

    # ---------------- CPU baseline (rank 0, N = 1 only) ----------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle as orc

        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = min(cores, 16)  # a one-GPU box owns a 16-thread share of the host
        t1 = time.time()
        oix = orc.Index(tables, sa=sa)
        log(rank, "oracle index (BWT/Occ/sampled SA) in %.1fs; timing on %d host threads" % (time.time() - t1, cores))
        probe = min(2000 * cores, args.reads_per_gpu)
        t1 = time.perf_counter()
        oix.align_batch(bases[: probe * L], offsets[: probe + 1], opts, n_threads=cores)
        rate = probe / (time.perf_counter() - t1)
        sample = int(min(args.reads_per_gpu, max(probe, rate * args.cpu_seconds)))
        t1 = time.perf_counter()
        r = oix.align_batch(bases[: sample * L], offsets[: sample + 1], opts, n_threads=cores)
        dt = time.perf_counter() - t1
        t1 = time.perf_counter()
        n1 = min(sample, 20000)
        oix.align_batch(bases[: n1 * L], offsets[: n1 + 1], opts, n_threads=1)
        dt1 = time.perf_counter() - t1
        cpu_baseline = {
            "value": round(sample / dt, 1), "unit": "reads/s", "cores": cores, "kind": "port",
            "sample": "first %d of the %d reads of rank 0's batch, %.1f s, %d threads" % (sample, args.reads_per_gpu, dt, cores),
            "value_1thread": round(n1 / dt1, 1),
            "note": "CPU restatement of the reference algorithm (oracle/), not the reference Rust binary",
            "aligned_frac": round(float(r.counters[1]) / max(int(r.counters[0]), 1), 4),
        }

    if rank == 0:
        out = {
            "metric": "aligned reads/sec (%d bp)" % L, "value": round(value, 1), "unit": "reads/s", "n_gpus": world,
            "steps": K, "warmup": args.warmup, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {
                "workload": "%d synthetic %d bp reads per GPU vs %s (%d bp, %d tx), flags %s%s" % (
                    args.reads_per_gpu, L,
                    "chr21-sized synthetic transcriptome" if args.workload == "chr21syn" else "GRCh38-2020-A chrM (real FASTA/GTF)",
                    ref_len, len(tables["txs"]),
                    "-k20 -s0 --intron-mode" if args.opts == "ci" else "defaults (-k20 -s0.66)",
                    "" if args.percent is None else " with -s%g" % args.percent),
                "reads_per_gpu": args.reads_per_gpu, "read_len": L, "ref_len": ref_len, "opts": args.opts,
                "parallelism": "reads sharded over %d GPU(s), index replicated, 1 counter all-reduce" % world,
            },
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "counters": {k: c[k] for k in ("reads", "aligned", "unmapped", "alns", "exonic", "intronic", "intergenic",
                                           "smems", "hits", "swg_calls", "dp_cells")},
        }
        print(json.dumps(out), flush=True)
    aligner.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
