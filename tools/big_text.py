#!/usr/bin/env python3
"""Builder-side run of the 64-bit-coordinate path on a text of 2^31 symbols or more (BASELINE config 5's
reference is the full GRCh38-2020-A transcriptome: about 6.2 G text symbols).

A synthetic genome of --genome-len bases gives a text of 2 * (len + 1) symbols (both strands,
reference src/index.rs:67-101); from 1 073 741 816 bases on, the library picks 64-bit text positions
and suffix-array ranks by itself.  The CPU oracle cannot hold such a text (its FMD index is 32-bit),
so the results are checked through size-independent properties: every sampled alignment is consistent
with the sequences at the coordinates it reports (thermite_amd/validate.py: labels of every op against
the text, ends, clips, score bounds), error-free reads align end to end with score = L, a replay is
idempotent.  The unit-test matrix (tests/test_gpu_align.py, ids c64) compares the same code path with
the oracle bit for bit on small texts.

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
log(json.dumps(diag))
out["error_free_reads"] = diag
out["error_free_reads_end_to_end"] = bool(len(zero) == 0 and diag["top_score_not_91"] == 0 and diag["top_not_full_length"] == 0 and diag["top_not_exonic"] == 0)
a.close()
out["wall_s"] = round(time.time() - T0, 1)
os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
with open(args.out, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
