#!/usr/bin/env python3
"""Heavy reads: a batch of ordinary reads plus reads seeded in a 3 000-copy repeat family
(hundreds to thousands of seed hits each).  Prints the hit histogram, the extend-kernel time with and without
the heavy reads, and checks the heavy batch against the CPU oracle.   python tools/heavy_reads.py [n_heavy]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from thermite_amd import capi, synth, refdata
from oracle import pyoracle as orc
from gpu_common import assert_batch_equal

n_heavy = int(sys.argv[1]) if len(sys.argv) > 1 else 200
which = sys.argv[2] if len(sys.argv) > 2 else "moderate"
if which == "extreme":  # 8 000 near-identical copies: error-free reads have thousands of exact hits, all accepted
    t, pos = synth.heavy_repeat_reference(length=6_000_000, copies=8000, divergence=0.01)
else:
    t, pos = synth.heavy_repeat_reference()
print("reference:", which, "repeat family of", len(pos), "copies")
ix = capi.Index(t)
rng = np.random.default_rng(3)
n_light = 200000 if which != "extreme" else 2000
light_b, light_o, _ = synth.simulate_reads(t, n_light, 91, sub_rate=0.01, indel_rate=0.001, stream=11)
lh = None
starts = pos[rng.integers(0, len(pos), n_heavy)] + rng.integers(0, 300 - 91, n_heavy)
heavy_b, heavy_o = synth.reads_from_positions(t, starts, 91, sub_rate=0.02, stream=12)
a = capi.Aligner(ix, capi.CI_OPTS)
mo, _ = a.smems_batch(heavy_b, heavy_o, 20)
h = np.diff(mo.astype(np.int64))
print("heavy reads: %d, hits/read mean %.0f median %.0f max %d" % (n_heavy, h.mean(), np.median(h), h.max()))


def timed(b, o, label):
    a.upload(b, o)
    a.run(); a.sync()
    acc = 0.0
    for _ in range(3):
        a.run(); a.sync(); acc += a.timings()["extend"] / 3
    print("%-34s n=%d extend %.3f ms" % (label, len(o) - 1, acc), flush=True)
    return a.fetch()


timed(light_b, light_o, "%d ordinary reads" % n_light)
both_b = np.concatenate([light_b, heavy_b])
both_o = np.concatenate([light_o, heavy_o[1:] + light_o[-1]]).astype("<u8")
g = timed(both_b, both_o, "the same + %d heavy reads" % n_heavy)
gh = timed(heavy_b, heavy_o, "the heavy reads alone")
t0 = time.time()
oix = orc.Index(t, sa=ix.suffix_array())
r = oix.align_batch(heavy_b, heavy_o, capi.CI_OPTS, n_threads=16)
print("oracle on the heavy reads: %.1f s" % (time.time() - t0))
assert_batch_equal(gh, r)
print("heavy reads: parity with the oracle ok (%d alignments)" % len(gh.alns))
