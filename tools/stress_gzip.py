#!/usr/bin/env python3
"""Randomised stress test of the host-side codecs (csrc/io_inflate.cpp serial and chunk-parallel, csrc/io_deflate.cpp)
against zlib, on the CPU:   python tools/stress_gzip.py [seed] [seconds]
Random data kinds, compression levels, members / flushes, chunk sizes, thread counts; a flipped bit must be an error."""
import sys, os, gzip, zlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from thermite_amd import capi
from test_io_host import _big_fastq
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
fq = _big_fastq(40000, 77)
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 60:
    kind = rng.integers(0, 5)
    size = int(rng.integers(1, len(fq)))
    a = int(rng.integers(0, len(fq) - size + 1))
    if kind == 0: data = fq[a:a+size]
    elif kind == 1: data = rng.integers(0, 256, size // 4 + 1, dtype=np.uint8).tobytes()
    elif kind == 2: data = fq[a:a+size//2] + rng.integers(0, 256, size // 8 + 1, dtype=np.uint8).tobytes() + fq[a:a+size//3]
    elif kind == 3: data = bytes(rng.choice(np.frombuffer(b"ACGT\n", np.uint8), size))
    else: data = b"".join(bytes([int(c)]) * int(k) for c, k in zip(rng.integers(60, 70, 2000), rng.integers(1, 900, 2000)))
    level = int(rng.choice([0, 1, 1, 6, 6, 9]))
    mode = rng.integers(0, 3)
    if mode == 0: z = gzip.compress(data, level)
    elif mode == 1:
        piece = int(rng.integers(1000, 200000)); z = b"".join(gzip.compress(data[s:s+piece], level) for s in range(0, max(len(data),1), piece))
    else:
        co = zlib.compressobj(level, zlib.DEFLATED, 31); z = b""
        step = int(rng.integers(500, 100000))
        for s in range(0, len(data), step):
            z += co.compress(data[s:s+step])
            if rng.random() < 0.3: z += co.flush(zlib.Z_FULL_FLUSH if rng.random() < 0.5 else zlib.Z_SYNC_FLUSH)
        z += co.flush()
    p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "thm_stress_%d.gz" % os.getpid()); open(p, "wb").write(z)
    os.environ["THM_INFLATE_CHUNK_KB"] = str(int(rng.choice([16, 16, 32, 64, 256])))
    threads = int(rng.choice([1, 2, 3, 4, 7]))
    chunk = int(rng.choice([1024, 3000, 65536, 1 << 20]))
    try:
        out = capi.debug_gunzip(p, chunk, threads=threads)
    except capi.ThermiteError:
        print("FAILED case", n, dict(kind=int(kind), level=level, mode=int(mode), threads=threads, chunk=chunk, chunk_kb=os.environ["THM_INFLATE_CHUNK_KB"], n=len(data), z=len(z)), "file kept:", p, flush=True)
        raise
    assert out == data, (n, kind, level, mode, threads, chunk, len(data), len(z))
    # a corrupted copy must be an error or (vanishingly unlikely) equal
    if len(z) > 40 and rng.random() < 0.5:
        zb = bytearray(z); at = int(rng.integers(10, len(z) - 8)); zb[at] ^= 1 << int(rng.integers(0, 8))
        open(p, "wb").write(bytes(zb))
        try:
            out2 = capi.debug_gunzip(p, chunk, threads=threads)
            ok = out2 == data
            if not ok:
                # a flipped bit in a gzip header field that is not checked (mtime, xfl, os, name) changes nothing
                assert at < 30 or False, ("corruption not detected", n, at, len(z), mode)
        except capi.ThermiteError as e:
            assert e.code == capi.ERR_IO
    # the BAM writer's deflate on a block of the same data
    blk = data[: int(rng.integers(0, 65281))]
    assert zlib.decompress(capi.debug_deflate_block(blk), -15) == blk, (n, "deflate", kind, len(blk))
    n += 1
os.remove(p)
print("stress ok:", n, "cases")
