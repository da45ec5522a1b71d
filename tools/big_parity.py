#!/usr/bin/env python3
"""Whole-batch parity with the CPU oracle at the benchmark's size, beyond the committed full-size tests: the four
batches the benchmark rotates over, chr21 flags and default flags, 500 000 reads each (alignment records and op streams
byte-identical, counters equal).   python tools/big_parity.py [streams...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import pyoracle as orc
from thermite_amd import capi, synth
from gpu_common import assert_batch_equal

streams = [int(x) for x in sys.argv[1:]] or [100, 101, 102, 103]
t = synth.synth_reference()
sa = capi.build_suffix_array(t["text"])
ix = capi.Index(t, sa=sa)
oix = orc.Index(t, sa=sa)
t0 = time.time()
for stream in streams:
    bases, off, _ = synth.simulate_reads(t, 500000, 91, sub_rate=0.01, indel_rate=0.001, stream=stream)
    for name, opts in (("ci", capi.CI_OPTS), ("default", capi.DEFAULT_OPTS)):
        a = capi.Aligner(ix, opts)
        a.reset_counters()
        g = a.align_batch(bases, off)
        c = a.counters()
        r = oix.align_batch(bases, off, opts, n_threads=16)
        assert r.counters[15] == 0
        assert_batch_equal(g, r)
        assert np.array_equal(c[:10], r.counters[:10]) and c[12] == r.counters[12] and c[13] == r.counters[13], (c[:14], r.counters[:14])
        a.close()
        print("stream %d %s: %d alignments, %d op bytes identical (%.0fs)" % (stream, name, len(g.alns), len(g.ops), time.time() - t0), flush=True)
print("big parity ok")
