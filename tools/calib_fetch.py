#!/usr/bin/env python3
"""FETCH_SIZE calibration: gathers of known size in the access patterns of the hot kernels (run under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE`; tools/profile_r02.sh collects the counter per launch).
Pattern 0: 8-byte random probes, 1: 4-byte random probes, 2: 256-byte runs of 16-byte lane loads."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermite_amd import capi, synth
import numpy as np

t = synth.synth_reference()  # chr21-sized: a 2.1 GB k-mer table (far beyond the 256 MiB Infinity Cache)
sa_path = "/tmp/thm_bench_sa_%d_%x.npy" % (synth.CHR21_LEN, synth.SEED)
sa = np.load(sa_path, mmap_mode="r") if os.path.exists(sa_path) else None
ix = capi.Index(t, sa=sa)
a = capi.Aligner(ix, capi.CI_OPTS)
n = 64 * 1024 * 1024
out = {}
for pattern in (0, 1, 2):
    for rep in range(3):
        out["pattern%d" % pattern] = {"threads": n, "bytes_requested": a.debug_calib_gather(pattern, n)}
print(json.dumps(out))
