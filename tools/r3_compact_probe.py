#!/usr/bin/env python3
"""Does the compact stage slow down because of WHERE its pools lie?  (round-3 verdict item 4)
The chr21-sized synthetic text, 500 000 reads of 91 bp, stage times of five steps:
  A  as bench.py runs it;
  B  the same after 40 GB of other device memory have been allocated and touched BEFORE the aligner's pools;
  C  B with the 40 GB being read by a concurrent stream?  no -- C frees the 40 GB again and repeats A (control)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from thermite_amd import capi, synth

tables = synth.synth_reference()
ix = capi.Index(tables)
bases, off, _ = synth.simulate_reads(tables, 500000, 91, sub_rate=0.01, indel_rate=0.001, stream=100)
out = {}


def stage_ms(tag):
    a = capi.Aligner(ix, capi.CI_OPTS)
    a.upload(bases, off)
    a.run()
    a.fetch()
    ms = {k: 0.0 for k in capi.TIMING_NAMES}
    for _ in range(5):
        a.run()
        a.sync()
        for k, v in a.timings().items():
            ms[k] += v / 5
    out[tag] = {k: round(v, 3) for k, v in ms.items()}
    print(tag, out[tag], flush=True)
    a.close()


stage_ms("A_plain")
big = torch.empty(40 << 30, dtype=torch.uint8, device="cuda")
big.zero_()
torch.cuda.synchronize()
stage_ms("B_after_40GB_allocated")
del big
torch.cuda.empty_cache()
stage_ms("C_40GB_freed_again")
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)
