#!/usr/bin/env python3
"""Which reads make a batch's extend kernel finish late?  For the four batches the benchmark rotates over: the hit-count
tail, the extend time as is, without the reads of >= T hits, and with only those reads.   python tools/tail_diag.py [quick]"""
import sys, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from thermite_amd import capi, synth

t = synth.synth_reference()
ix = capi.Index(t)
a = capi.Aligner(ix, capi.CI_OPTS)


def run(b, o):
    a.upload(b, o)
    for _ in range(2):
        a.run(); a.sync()
    acc = 0.0
    for _ in range(5):
        a.run(); a.sync(); acc += a.timings()["extend"]
    return acc / 5


for stream in (100, 101, 102, 103):
    bases, off, _ = synth.simulate_reads(t, 500000, 91, sub_rate=0.01, indel_rate=0.001, stream=stream)
    moff, mems = a.smems_batch(bases, off, 20)
    h = np.diff(moff.astype(np.int64))
    reads = bases.reshape(-1, 91)
    top = np.sort(h)[::-1][:12]
    line = "stream %d: top hit counts %s; reads >=8: %d, >=64: %d, >=128: %d, >=256: %d; extend as is %.3f ms" % (
        stream, top.tolist(), (h >= 8).sum(), (h >= 64).sum(), (h >= 128).sum(), (h >= 256).sum(), run(bases, off))
    for thr in (() if len(sys.argv) > 1 and sys.argv[1] == "quick" else (256, 128, 64)):
        keep = h < thr
        if keep.all():
            continue
        b2 = np.ascontiguousarray(reads[keep]).reshape(-1)
        o2 = (np.arange(keep.sum() + 1, dtype=np.uint64) * 91).astype("<u8")
        line += "; without >=%d: %.3f" % (thr, run(b2, o2))
        b3 = np.ascontiguousarray(reads[~keep]).reshape(-1)
        o3 = (np.arange((~keep).sum() + 1, dtype=np.uint64) * 91).astype("<u8")
        line += " (those %d reads alone: %.3f)" % ((~keep).sum(), run(b3, o3))
    print(line, flush=True)
