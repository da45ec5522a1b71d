#!/usr/bin/env python3
"""Quick stage timing on the GPU box (tuning aid, not the benchmark):
python tools_perf.py [ref_len] [n_reads] [opts]"""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from thermite_amd import capi, synth

ref_len = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
which = sys.argv[3] if len(sys.argv) > 3 else "both"
t = synth.synth_reference(length=ref_len)
ix = capi.Index(t)
bases, off, _ = synth.simulate_reads(t, n, 91, sub_rate=0.01, indel_rate=0.001, stream=100)
for name, opts in (("ci", capi.CI_OPTS), ("default", capi.DEFAULT_OPTS)):
    if which not in ("both", name):
        continue
    a = capi.Aligner(ix, opts)
    a.upload(bases, off)
    for _ in range(2):
        a.run(); a.sync()
    acc = {}
    K = 5
    t0 = time.perf_counter()
    for _ in range(K):
        a.run(); a.sync()
        for k, v in a.timings().items():
            acc[k] = acc.get(k, 0.0) + v / K
    dt = (time.perf_counter() - t0) / K
    print("%-8s ref=%d n=%d  %.2f Mreads/s  wall %.2f ms  " % (name, ref_len, n, n / dt / 1e6, dt * 1e3) +
          " ".join("%s=%.2f" % kv for kv in acc.items()), flush=True)
    c = dict(zip(capi.COUNTER_NAMES, a.counters().tolist()))
    runs = K + 2
    print("         per read: smems %.2f hits %.2f swg_calls %.2f cols %.1f cells %.0f alns %.2f win_bytes %.0f" % tuple(
        c[k] / (n * runs) for k in ("smems", "hits", "swg_calls", "dp_cols", "dp_cells", "alns", "window_bytes")), flush=True)
    if hasattr(a, "debug_tpr_stats"):
        es = a.debug_tpr_stats()
        names = {1: "band", 2: "grid", 3: "lift", 7: "other", 8: "rounds", 9: "candidates", 10: "edits", 11: "introns", 12: "pools", 13: "window", 14: "open"}
        print("         problem-parallel path: %d DP requests (narrow %d, by band class %s), %d reads left to the wave-per-read kernel (%s), "
              "%d still waiting after the last round" % (es[16], es[17], es[18:22].tolist(), es[0],
                                                         " ".join("%s %d" % (names.get(k, str(k)), es[k]) for k in range(1, 16) if es[k]), es[22]), flush=True)
    pr = a.debug_prof()
    if pr.sum() > 0:
        names = ["setup", "stage", "dp", "traceback", "tree", "txprep", "lift", "emit", "final", "other"]
        tot = float(pr[:10].sum())
        pc = lambda k: 100.0 * pr[k] / max(pr[10], 1)
        print("         DP columns: total %d, on <=32 slots %.1f%%, pairable (min of L/R when both <=32) %.1f%%, first hit of the read %.1f%%, "
              "transcript targets %.1f%%, one-mismatch-then-exact extensions %.1f%%" % (pr[10], pc(11), pc(12), pc(13), pc(14), pc(15)), flush=True)
        print("         extend sections: " + " ".join("%s=%.1f%%" % (nm, 100.0 * v / tot) for nm, v in zip(names, pr[:10])), flush=True)
    a.close()
