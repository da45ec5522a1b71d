#!/usr/bin/env python3
"""Randomised differential test on the GPU box (beyond the committed tests): random options, ragged and dirty
reads (some rounds: a few reads of 300 - 1500 bases among them, i.e. the any-width kernel beside the fast one),
against the CPU oracle.   python tools/fuzz_gpu.py [rounds] [seed] [syn|heavy]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import pyoracle as orc
from thermite_amd import capi, refdata, synth
from gpu_common import assert_batch_equal

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
which = sys.argv[3] if len(sys.argv) > 3 else "syn"
if which == "heavy":  # a repeat family of thousands of copies: reads with hundreds to thousands of seed hits (team kernel)
    t, _ = synth.heavy_repeat_reference(length=1500000, copies=1200)
else:
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
    if rng.random() < 0.3:
        for j in range(int(rng.integers(1, 6))):
            LL = int(rng.integers(300, 1500))
            lb, lo, _ = synth.simulate_reads(t, 2, LL, sub_rate=0.02, indel_rate=0.004, intronic_frac=0.5, stream=5000 + 10 * r + j)
            for i in range(2):
                reads.insert(int(rng.integers(0, len(reads) + 1)), lb[lo[i]: lo[i + 1]])
    b2, o2 = refdata.pack_reads(reads)
    a = capi.Aligner(ix, opts)
    if rng.random() < 0.25:  # small pools: overflow -> grow -> replay (thm_batch_sync)
        a.debug_set_pool_caps(smem_cap=int(rng.integers(64, 2000)), cand_cap=int(rng.integers(16, 2000)), ops_cap=int(rng.integers(4096, 100000)))
    # either extend path: the fused wave-per-read kernels (default) or the problem-parallel one
    tpr = bool(rng.random() < 0.5)
    a.debug_set_flags(tpr=tpr)
    g = a.align_batch(b2, o2)
    ref = oix.align_batch(b2, o2, opts, n_threads=16)
    assert ref.counters[15] == 0
    try:
        assert_batch_equal(g, ref)
    except AssertionError as e:
        print("MISMATCH round %d: L=%d k=%d opts=%r\n%s" % (r, L, k, opts, str(e)[:300]), flush=True)
        # the first read whose records or op bytes differ, both sides in full
        for rd in range(len(o2) - 1):
            ga, gb = int(g.offsets[rd]), int(g.offsets[rd + 1])
            ra, rb = int(ref.offsets[rd]), int(ref.offsets[rd + 1])
            same = (gb - ga == rb - ra)
            if same:
                for i in range(gb - ga):
                    x, y = g.alns[ga + i], ref.alns[ra + i]
                    gx = bytes(g.ops[int(x["ops_off"]): int(x["ops_off"]) + int(x["ops_len"]) + int(x["tx_ops_len"])])
                    ry = bytes(ref.ops[int(y["ops_off"]): int(y["ops_off"]) + int(y["ops_len"]) + int(y["tx_ops_len"])])
                    if gx != ry or any(x[f] != y[f] for f in capi.ALN_DT.names if f not in ("pad_", "ops_off", "tx_ops_off")):
                        same = False
            if not same:
                print("read %d (%d bases): %s" % (rd, int(o2[rd + 1] - o2[rd]), bytes(b2[int(o2[rd]): int(o2[rd + 1])]).decode("latin1")))
                for nm, res, a0, a1 in (("gpu", g, ga, gb), ("ref", ref, ra, rb)):
                    for i in range(a0, a1):
                        x = res.alns[i]
                        ops = bytes(res.ops[int(x["ops_off"]): int(x["ops_off"]) + int(x["ops_len"])])
                        tops = bytes(res.ops[int(x["tx_ops_off"]): int(x["tx_ops_off"]) + int(x["tx_ops_len"])]) if x["tx_ops_len"] else b""
                        print("  %s aln %d: %s\n      ops %s\n      tx_ops %s" % (nm, i - a0, x, orc.decode_ops(ops), orc.decode_ops(tops)))
                moff, mems = a.smems_batch(b2[int(o2[rd]): int(o2[rd + 1])], np.array([0, int(o2[rd + 1] - o2[rd])], "<u8"), k)
                print("  mems of the read: %d" % len(mems), mems[:12])
                break
        raise
    go, gm = a.smems_batch(b2, o2, k)
    rm = oix.all_smems(b2, o2, k)
    assert np.array_equal(go, rm.offsets) and all(np.array_equal(gm[f], rm.mems[f]) for f in ("ref_idx", "query_idx", "len"))
    a.close()
    print("round %d ok [%s]: L=%d k=%d pct=%.2f n=%d (longest %d) alns=%d mems=%d (%.0fs)" % (r, "problem-parallel" if tpr else "fused", L, k, pct, len(o2) - 1, int(np.diff(o2.astype(np.int64)).max()), len(g.alns), len(gm), time.time() - t0), flush=True)
print("fuzz ok")
