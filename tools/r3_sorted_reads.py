#!/usr/bin/env python3
"""What would processing the reads in genome order buy?  The same 500 000 reads twice through the same aligner: in
input (random) order and sorted by the locus they were drawn from -- stage times of five steps each.
   python tools/r3_sorted_reads.py [genome_len] [out.json]
(An experiment for DESIGN.md's 'next' list: a device-side sort of the reads by their first hit would cost ~50 us.)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermite_amd import capi, synth

G = int(sys.argv[1]) if len(sys.argv) > 1 else synth.CHR21_LEN
tables = synth.synth_reference(length=G)
t0 = time.time()
ix = capi.Index(tables)
print("index (%d symbols) in %.0f s" % (len(tables["text"]), time.time() - t0), flush=True)
out = {"text_symbols": int(len(tables["text"]))}
for L, opts, tag in ((91, capi.CI_OPTS, "91bp"), (150, dict(capi.CI_OPTS, min_aln_score_percent=0.574), "150bp_band64")):
    bases, off, truth = synth.simulate_reads(tables, 500000, L, sub_rate=0.01, indel_rate=0.001, stream=100)
    txs, exons = tables["txs"], tables["exons"]
    locus = exons["start"][txs["exon_begin"][truth["tx"]]].astype(np.int64) + truth["start"].astype(np.int64)
    order = np.argsort(locus, kind="stable")
    reads = bases.reshape(-1, L)
    for name, b in (("input_order", reads), ("genome_order", reads[order])):
        a = capi.Aligner(ix, opts)
        a.upload(np.ascontiguousarray(b).reshape(-1), off)
        a.run()
        a.fetch()
        ms = {k: 0.0 for k in capi.TIMING_NAMES}
        for _ in range(5):
            a.run()
            a.sync()
            for k, v in a.timings().items():
                ms[k] += v / 5
        out[tag + "_" + name] = {k: round(v, 3) for k, v in ms.items()}
        print(tag, name, out[tag + "_" + name], flush=True)
        a.close()
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
