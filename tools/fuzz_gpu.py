#!/usr/bin/env python3
"""Randomised differential test on the GPU box (beyond the committed tests): random options, ragged and dirty
reads, against the CPU oracle.   python tools/fuzz_gpu.py [rounds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import pyoracle as orc
from thermite_amd import capi, refdata, synth
from gpu_common import assert_batch_equal

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t = synth.synth_reference(length=600000, n_genes=60)
ix = capi.Index(t)
oix = orc.Index(t, sa=ix.suffix_array())
t0 = time.time()
for r in range(rounds):
    L = int(rng.choice([30, 50, 75, 91, 120, 150, 200, 260]))
    k = int(rng.integers(8, 26))
    pct = float(rng.choice([0.0, 0.3, 0.5, 0.66, 0.8, 0.9]))
    opts = dict(min_seed_len=k, min_aln_score_percent=pct, min_aln_score=int(rng.choice([0, 20, 30])),
                multimap_score_range=int(rng.integers(0, 4)), intron_mode=bool(rng.integers(0, 2)))
    n = int(rng.integers(500, 6000))
    bases, off, _ = synth.simulate_reads(t, n, L, sub_rate=float(rng.choice([0.0, 0.01, 0.03, 0.08])),
                                         indel_rate=float(rng.choice([0.0, 0.002, 0.01])), intronic_frac=0.2, stream=1000 + r)
    b = bases.copy()
    for ch, p in ((ord("N"), 0.002), (ord("x"), 0.0003)):
        b[rng.random(len(b)) < p] = ch
    lower = rng.random(len(b)) < 0.2
    b[lower] = np.where((b[lower] >= 65) & (b[lower] <= 90), b[lower] + 32, b[lower])
    reads = [b[int(off[i]): int(off[i]) + (L if rng.random() < 0.7 else int(rng.integers(0, L + 1)))] for i in range(n)]
    b2, o2 = refdata.pack_reads(reads)
    a = capi.Aligner(ix, opts)
    g = a.align_batch(b2, o2)
    ref = oix.align_batch(b2, o2, opts, n_threads=16)
    assert ref.counters[15] == 0
    assert_batch_equal(g, ref)
    go, gm = a.smems_batch(b2, o2, k)
    rm = oix.all_smems(b2, o2, k)
    assert np.array_equal(go, rm.offsets) and all(np.array_equal(gm[f], rm.mems[f]) for f in ("ref_idx", "query_idx", "len"))
    a.close()
    print("round %d ok: L=%d k=%d pct=%.2f n=%d alns=%d mems=%d (%.0fs)" % (r, L, k, pct, n, len(g.alns), len(gm), time.time() - t0), flush=True)
print("fuzz ok")
