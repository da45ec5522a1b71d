#!/usr/bin/env python3
"""When do the waves of the wave-per-read extend kernel run out of work?  Needs the diagnosis build: make -C thermite_amd/csrc timeline;
THM_LIB=thermite_amd/_build/libthermite_amd_timeline.so   python tools/timeline.py [n_reads] [stream]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermite_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
stream = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t = synth.synth_reference()
ix = capi.Index(t)
a = capi.Aligner(ix, capi.CI_OPTS)
bases, off, _ = synth.simulate_reads(t, n, 91, sub_rate=0.01, indel_rate=0.001, stream=stream)
a.upload(bases, off)
for _ in range(3):
    a.run(); a.sync()
a.debug_prof(reset=True)
a.run(); a.sync()
ext = a.timings()["extend"]
p = a.debug_prof()
M = (1 << 64) - 1
start, first_end, last_end = (~int(p[0])) & M, (~int(p[1])) & M, int(p[2])
us = lambda x: x / 100.0
print("n=%d stream=%d extend stage %.3f ms" % (n, stream, ext))
print("waves with work: %d, reads %d (%.1f per wave), mean wave life %.0f us" % (p[5], p[7], p[7] / max(p[5], 1), us(p[6]) / max(p[5], 1)))
print("first wave leaves at %.0f us, last at %.0f us after the first start" % (us(first_end - start), us(last_end - start)))
print("last wave's last read took %d us; longest read of the launch: %.0f us, %d hits, %d transcript targets" % (
    int(p[3]) & 0xfffff, us(int(p[4]) >> 32), int(p[4]) & 0xffff, (int(p[4]) >> 16) & 0xffff))
names = ["<3.0ms", "3.0-3.1", "3.1-3.2", "3.2-3.3", "3.3-3.4", "3.4-3.5", "3.5-3.6", ">=3.6"]
print("waves leaving, on the launch's clock:", " ".join("%s:%d" % (nm, int(x)) for nm, x in zip(names, p[8:16])))
