#!/usr/bin/env python3
"""Micro-benchmark of the operator-level SWG kernel (tuning aid): columns/s of the DP alone."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from thermite_amd import capi, refdata

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
D = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "tests", "golden", "data")
t = refdata.load_reference(D + "/test_ref.fasta", D + "/test_ref.gtf")
ix = capi.Index(t)
a = capi.Aligner(ix, dict(capi.DEFAULT_OPTS, min_seed_len=3, min_aln_score=0))
rng = np.random.default_rng(1)
ACGT = np.frombuffer(b"ACGT", np.uint8)
for xlen, bw in ((45, 61), (45, 31), (45, 3), (20, 61), (70, 61)):
    x = ACGT[rng.integers(0, 4, (n, xlen))]
    y = np.concatenate([x, ACGT[rng.integers(0, 4, (n, 60))]], axis=1)
    m = rng.random(x.shape) < 0.02
    y[:, :xlen][m] = ACGT[rng.integers(0, 4, int(m.sum()))]
    xb = x.reshape(-1); yb = y.reshape(-1)
    xo = np.arange(n + 1, dtype="<u8") * xlen; yo = np.arange(n + 1, dtype="<u8") * (xlen + 60)
    bws = np.full(n, bw, "<u4"); xds = np.full(n, bw, "<i4")
    try:
        a.swg_extend_batch(xb, xo, yb, yo, bws, xds, bw)  # warm
    except Exception as e:  # timing-only experiment builds produce wrong traces
        print('warm:', e)
    a.reset_counters()
    t0 = time.perf_counter()
    try:
        a.swg_extend_batch(xb, xo, yb, yo, bws, xds, bw)
    except Exception as e:
        print('run:', e)
    dt = time.perf_counter() - t0
    c = a.counters()
    print("xlen %d bw %d: %d problems, cols/problem %.1f, wall %.1f ms (includes H2D/D2H)" % (xlen, bw, n, c[11] / n, dt * 1e3), flush=True)
