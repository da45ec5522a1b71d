#!/usr/bin/env python3
"""Prototype check (round-3 review item 10): tools/gpu_suffix_sort.hip against the library.
   python tools/gpu_sa_check.py [genome_len ...]
Builds the prototype if needed, sorts the suffixes of the synthetic text on the GPU, and hands the result to
thm_index_create as a supplied suffix array -- which verifies it (a text has one suffix array, so 'verified' is 'equal
to csrc/sais.cpp's'); for texts below 300 M symbols the host builder runs too and the arrays are compared directly."""
import ctypes as C, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from thermite_amd import capi, synth

so = os.path.join(ROOT, "thermite_amd", "_build", "libthm_gpusa.so")
src = os.path.join(ROOT, "tools", "gpu_suffix_sort.hip")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", src, "-o", so])
if len(sys.argv) > 1 and sys.argv[1] == "--build-only":
    sys.exit(0)
L = C.CDLL(so)
L.gpu_suffix_array.restype = C.c_int
L.gpu_suffix_array.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
for G in [int(x) for x in sys.argv[1:]] or [4_000_000]:
    t = synth.synth_reference(length=G)
    text = np.ascontiguousarray(t["text"])
    n = len(text)
    out = np.empty(n, np.uint32)
    secs, rounds = C.c_double(0), C.c_int(0)
    rc = L.gpu_suffix_array(text.ctypes.data, n, out.ctypes.data, C.byref(secs), C.byref(rounds))
    assert rc == 0, rc
    print("n = %d symbols: GPU suffix sort %.2f s (%d rounds, transfers included)" % (n, secs.value, rounds.value), flush=True)
    t0 = time.time()
    ix = capi.Index(t, sa=out.astype("<u8") if n >= (1 << 31) - 16 else out)  # verifies the supplied array
    print("   verified by thm_index_create (supplied suffix array checked, index built) in %.1f s" % (time.time() - t0), flush=True)
    ix.close()
    if n < 300_000_000:
        t0 = time.time()
        ref = capi.build_suffix_array(text)
        print("   host SA-IS %.1f s; arrays equal: %s" % (time.time() - t0, bool(np.array_equal(ref.astype(np.uint64), out.astype(np.uint64)))), flush=True)
        assert np.array_equal(ref.astype(np.uint64), out.astype(np.uint64))
