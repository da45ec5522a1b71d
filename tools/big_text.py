#!/usr/bin/env python3
"""Builder-side run of the 64-bit-coordinate path on a text of 2^31 symbols or more (BASELINE config 5's
reference is the full GRCh38-2020-A transcriptome: about 6.2 G text symbols).

A synthetic genome of --genome-len bases gives a text of 2 * (len + 1) symbols (both strands,
reference src/index.rs:67-101); from 1 073 741 816 bases on, the library picks 64-bit text positions
and suffix-array ranks by itself.  The CPU oracle is usize-wide like the reference (src/index.rs:103-111), so a
sample of every read shape (--oracle-reads, default 20 000) is compared with it byte for byte on this text -- the
one place where a truncated high half of a coordinate could hide -- with the oracle's FMD index built from the
library's own 64-bit suffix array.  The rest of the batch is checked through size-independent properties: every sampled
alignment is consistent with the sequences at the coordinates it reports (thermite_amd/validate.py: labels of every
op against the text, ends, clips, score bounds) and a replay is idempotent.  Error-free reads: also compared with the
oracle (a few of them have no alignment under the default flags on the repeat-bearing synthetic text; the oracle
leaves the same reads unaligned, so it is the algorithm and not the coordinate width -- the tool prints examples of
them, and what it asserts is equality with the oracle).  The unit-test matrix (tests/test_gpu_align.py, ids c64) compares the same code path with the oracle on
small texts.

    python tools/big_text.py --genome-len 1100000000 --out gpurun_out/big_text.json
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thermite_amd import capi, synth, validate  # noqa: E402


def log(*a):
    print("[big_text %6.0fs]" % (time.time() - T0), *a, file=sys.stderr, flush=True)


T0 = time.time()


def _heartbeat():  # the index build is minutes of host work without output: the GPU pool takes silence for a hang
    import threading

    def beat():
        while True:
            time.sleep(60)
            log("... still working")

    threading.Thread(target=beat, daemon=True).start()


_heartbeat()
ap = argparse.ArgumentParser()
ap.add_argument("--genome-len", type=int, default=1_100_000_000)
ap.add_argument("--reads", type=int, default=500000)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--oracle-reads", type=int, default=20000, help="reads per shape compared with the CPU oracle byte for byte (0: none)")
ap.add_argument("--out", default="gpurun_out/big_text.json")
args = ap.parse_args()

G = args.genome_len
tables = synth.synth_reference(length=G)
n = len(tables["text"])
log("reference: %d bp, text n = %d symbols (2^31 = %d), %d transcripts, %d exons" % (G, n, 1 << 31, len(tables["txs"]), len(tables["exons"])))
t1 = time.time()
ix = capi.Index(tables)  # no suffix array supplied: the library builds it (64-bit induced sorting)
t_index = time.time() - t1
log("index built in %.1f s, coordinate bytes = %d" % (t_index, ix.coord_bytes))
out = {"genome_len": G, "text_symbols": n, "coord_bytes": ix.coord_bytes, "index_build_s": round(t_index, 1),
       "transcripts": int(len(tables["txs"])), "exons": int(len(tables["exons"])), "runs": []}
assert n < (1 << 31) - 16 or ix.coord_bytes == 8
oix = None
if args.oracle_reads:
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as orc  # the checker (tests / tools only)

    t1 = time.time()
    sa = ix.suffix_array()
    log("suffix array fetched from the library (%s, %.1f GB) in %.1f s" % (sa.dtype, sa.nbytes / 1e9, time.time() - t1))
    t1 = time.time()
    oix = orc.Index(tables, sa=sa if sa.dtype.itemsize == 8 else sa.astype("<u8"), verify=False, keep_sa=False)
    del sa
    out["oracle_index_build_s"] = round(time.time() - t1, 1)
    log("oracle FMD index (usize-wide) built in %.1f s" % (time.time() - t1))


def equal_to_oracle(g, bases, off, opts, m, L):
    """the first m reads of the batch against the oracle: whole arrays"""
    r = oix.align_batch(bases[: m * L], off[: m + 1], opts, n_threads=16)
    n_a = int(g.offsets[m])
    same = (np.array_equal(g.offsets[: m + 1], r.offsets) and all(np.array_equal(g.alns[f][:n_a], r.alns[f]) for f in capi.ALN_DT.names if f not in ("pad_", "ops_off", "tx_ops_off")))
    if same:  # op streams of the first m reads are the first bytes of the pool
        n_ops = len(r.ops)
        same = np.array_equal(g.ops[:n_ops], r.ops)
    hi = int((r.alns["ystart"].astype(np.uint64) + tables["refs"]["start_idx"][r.alns["ref_id"]].astype(np.uint64)).max()) if len(r.alns) else 0
    return bool(same), int(len(r.alns)), hi


for L, opts, tag in ((91, capi.CI_OPTS, "91 bp, -k20 -s0 --intron-mode (band +-61)"),
                     (150, dict(capi.CI_OPTS, min_aln_score_percent=0.574), "150 bp, band +-64 (BASELINE config 5 read shape)")):
    bases, off, truth = synth.simulate_reads(tables, args.reads, L, sub_rate=0.01, indel_rate=0.001, stream=100)
    a = capi.Aligner(ix, opts)
    t1 = time.time()
    a.upload(bases, off)
    a.run()
    g = a.fetch()
    log("%s: first pass (index upload + pools) %.1f s" % (tag, time.time() - t1))
    a.reset_counters()
    ms = {k: 0.0 for k in capi.TIMING_NAMES}
    t1 = time.perf_counter()
    for _ in range(args.steps):
        a.run()
        a.sync()
        for k, v in a.timings().items():
            ms[k] += v
    dt = time.perf_counter() - t1
    g2 = a.fetch()
    assert np.array_equal(g.offsets, g2.offsets) and np.array_equal(g.alns, g2.alns) and np.array_equal(g.ops, g2.ops), "replay differs"
    assert g.n_failed == 0
    n_alns = np.diff(g.offsets.astype(np.int64))
    checked, bad = validate.check_batch(tables, bases, off, g, max_alns=4000)
    assert not bad, bad[:5]
    # how far up the text the alignments reach (concatenated coordinates beyond 2^31 / 2^32 exercise the wide fields)
    refs = tables["refs"]
    cat = refs["start_idx"][g.alns["ref_id"]].astype(np.uint64) + g.alns["ystart"]
    cnt = dict(zip(capi.COUNTER_NAMES, [int(v) for v in a.counters()]))
    run = {"workload": "%d synthetic reads, %s" % (args.reads, tag), "reads_per_s": round(args.reads * args.steps / dt, 1),
           "ms_per_step": round(dt / args.steps * 1e3, 3), "stage_ms": {k: round(v / args.steps, 3) for k, v in ms.items()},
           "aligned_frac": round(float((n_alns > 0).mean()), 5), "alignments": int(len(g.alns)),
           "alignments_checked_against_text": int(checked), "violations": len(bad),
           "alignments_on_reverse_strand_copy": int((g.alns["strand"] == 0).sum()),
           "hits_per_read": round(cnt["hits"] / max(cnt["reads"], 1), 3)}
    if oix is not None:
        m = min(args.oracle_reads, args.reads)
        t1 = time.time()
        same, n_or, hi = equal_to_oracle(g, bases, off, opts, m, L)
        run["oracle"] = {"reads_compared": m, "alignments": n_or, "equal_byte_for_byte": same,
                         "highest_text_position_of_an_alignment": hi, "beyond_2^32": hi >= (1 << 32), "seconds": round(time.time() - t1, 1)}
        assert same, "the first %d reads differ from the oracle" % m
    log(json.dumps(run))
    out["runs"].append(run)
    a.close()

# error-free reads align end to end at their origin (default flags: exonic alignments only)
bases, off, truth = synth.simulate_reads(tables, 50000, 91, sub_rate=0.0, indel_rate=0.0, flip_prob=0.0, stream=5)  # transcript strand
a = capi.Aligner(ix, capi.DEFAULT_OPTS)
g = a.align_batch(bases, off)
n_alns = np.diff(g.offsets.astype(np.int64))
zero = np.nonzero(n_alns == 0)[0]
has = np.nonzero(n_alns > 0)[0]
top = g.alns[g.offsets[:-1].astype(np.int64)[has]]
diag = {"reads": 50000, "without_alignment": int(len(zero)), "top_score_not_91": int((top["score"] != 91).sum()),
        "top_not_full_length": int((top["xend"] - top["xstart"] != 91).sum()), "top_not_exonic": int((top["aln_type"] != 0).sum())}
if len(zero):  # what are they?
    mo, mems = a.smems_batch(bases, off, 20)
    h = np.diff(mo.astype(np.int64))
    z = zero[:8]
    diag["zero_examples"] = [{"read": int(r), "hits": int(h[r]), "tx": int(truth["tx"][r]), "tx_strand": int(tables["txs"]["strand"][truth["tx"][r]]),
                              "tx_start": int(truth["start"][r]),
                              "mems": [(int(m["ref_idx"]), int(m["query_idx"]), int(m["len"])) for m in mems[int(mo[r]): int(mo[r]) + 4]],
                              "exon0_start": int(tables["exons"]["start"][tables["txs"]["exon_begin"][truth["tx"][r]]])} for r in z]
    diag["zero_hits_hist"] = {"max": int(h[zero].max()), "median": float(np.median(h[zero]))}
    b2 = a  # the same reads in --intron-mode: what do they align as?
    a2 = capi.Aligner(ix, capi.CI_OPTS)
    zb = np.concatenate([bases[int(off[r]): int(off[r + 1])] for r in z])
    zo = (np.arange(len(z) + 1, dtype=np.uint64) * 91).astype("<u8")
    g2 = a2.align_batch(zb, zo)
    diag["zero_examples_in_intron_mode"] = [(int(x["score"]), int(x["aln_type"]), int(x["ref_id"]), int(x["ystart"])) for x in g2.alns[:16]]
    a2.close()
if oix is not None:
    m = min(args.oracle_reads, 50000)
    same, n_or, hi = equal_to_oracle(g, bases, off, capi.DEFAULT_OPTS, m, 91)
    r0 = oix.align_batch(bases[: m * 91], off[: m + 1], capi.DEFAULT_OPTS, n_threads=16)
    diag["oracle"] = {"reads_compared": m, "equal_byte_for_byte": same,
                      "without_alignment_in_the_oracle_too": int((np.diff(r0.offsets.astype(np.int64)) == 0).sum()),
                      "without_alignment_here": int((n_alns[:m] == 0).sum())}
    assert same
log(json.dumps(diag))
out["error_free_reads"] = diag
# (A handful of error-free reads have no alignment under the default flags: on a repeat-bearing text the oracle leaves the same
# reads unaligned -- the algorithm, not the coordinate width.  The check that counts is equality with the oracle, above.)
out["error_free_reads_equal_the_oracle"] = bool(diag.get("oracle", {}).get("equal_byte_for_byte", False)) if oix is not None else None
a.close()
out["wall_s"] = round(time.time() - T0, 1)
os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
with open(args.out, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
