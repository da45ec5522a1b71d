#!/usr/bin/env python3
"""End-to-end file throughput on the GPU box (tuning aid, not the benchmark):
FASTQ on disk -> thm_align_files -> SAM / PAF.   python tools_e2e.py [ref_len] [n_reads] [threads]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from thermite_amd import capi, synth

ref_len = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t = synth.synth_reference(length=ref_len)
ix = capi.Index(t)
bases, off, _ = synth.simulate_reads(t, n, 91, sub_rate=0.01, indel_rate=0.001, stream=100)
t0 = time.time()
path = "/tmp/thm_e2e_%d.fastq" % n
reads = bases.reshape(n, 91)
qual = np.full(91, ord("F"), np.uint8)
with open(path, "wb") as f:
    CH = 100000
    for s in range(0, n, CH):
        e = min(n, s + CH)
        parts = []
        for i in range(s, e):
            parts.append(b"@SYN:%d 1:N:0:ACGT\n" % i)
            parts.append(reads[i].tobytes())
            parts.append(b"\n+\n")
            parts.append(qual.tobytes())
            parts.append(b"\n")
        f.write(b"".join(parts))
print("fastq: %d reads, %.1f MB, written in %.1fs" % (n, os.path.getsize(path) / 1e6, time.time() - t0), flush=True)
a = capi.Aligner(ix, capi.CI_OPTS)
# the plain-FASTQ runs on a file of REP copies as well (a run over 2 M reads lasts 0.2 s: mostly first-touch of buffers)
REP = int(sys.argv[4]) if len(sys.argv) > 4 else 4
big = path + ".x%d" % REP
with open(big, "wb") as f:
    one = open(path, "rb").read()
    for _ in range(REP):
        f.write(one)
    del one
print("plain input: %d reads, %.1f MB" % (n * REP, os.path.getsize(big) / 1e6), flush=True)
for fmt, name in ((capi.FMT_SAM, "sam"), (capi.FMT_PAF, "paf")):
    for rep in range(2):
        out = "/tmp/thm_e2e_out.%s" % name
        st = capi.align_files(a, [big], out, fmt, batch_reads=250000, n_threads=threads)
        print("%s run %d: %.2f Mreads/s wall %.2fs | parse %.2fs gpu %.2fs format %.2fs write %.2fs | %d batches, %.0f MB out" % (
            name, rep, st["n_reads"] / st["wall_s"] / 1e6, st["wall_s"], st["parse_s"], st["gpu_s"], st["format_s"], st["write_s"],
            st["n_batches"], st["n_output_bytes"] / 1e6), flush=True)
# gzip input (the reference's own input format, src/aligner.rs:51-52): one file, then two files (an inflater thread each)
import gzip, shutil, subprocess
gz = path + ".gz"
t0 = time.time()
if shutil.which("gzip"):
    subprocess.check_call("gzip -6 -c %s > %s" % (path, gz), shell=True)
else:
    with open(path, "rb") as fi, gzip.open(gz, "wb", compresslevel=6) as fo:
        shutil.copyfileobj(fi, fo, 1 << 24)
print("gzip -6: %.1f MB in %.1fs" % (os.path.getsize(gz) / 1e6, time.time() - t0), flush=True)
# longer inputs without waiting for gzip: REP copies of the member back to back (a valid gzip file; `cat a.gz a.gz`)
if REP > 1:
    one = open(gz, "rb").read()
    with open(gz, "wb") as f:
        for _ in range(REP):
            f.write(one)
    del one
gz2 = path + ".2.gz"
shutil.copyfile(gz, gz2)
try:
    quota = open("/sys/fs/cgroup/cpu.max").read().strip()
except OSError:
    quota = "?"
print("each .gz: %d reads in %d member(s), %.1f MB; host: %d hardware threads, cpu.max %s" % (n * REP, REP, os.path.getsize(gz) / 1e6, os.cpu_count(), quota), flush=True)


def digest(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


paf_of_plain = digest("/tmp/thm_e2e_out.paf")  # (the last plain run wrote PAF of the REP-copy file)


def run(tag, paths, fmt, out):
    for rep in range(2):
        st = capi.align_files(a, paths, out, fmt, batch_reads=250000, n_threads=threads)
        print("%s run %d: %.2f Mreads/s wall %.2fs | parse+inflate %.2fs gpu %.2fs format %.2fs write %.2fs | %.0f MB out" % (
            tag, rep, st["n_reads"] / st["wall_s"] / 1e6, st["wall_s"], st["parse_s"], st["gpu_s"], st["format_s"], st["write_s"],
            st["n_output_bytes"] / 1e6), flush=True)


run("paf from one .gz", [gz], capi.FMT_PAF, "/tmp/thm_e2e_out.paf")
# the .gz holds the same records as the plain file (REP members of the same reads): same PAF, byte for byte
same = digest("/tmp/thm_e2e_out.paf") == paf_of_plain
print("PAF of the .gz input %s the PAF of the plain input (sha256 over %d reads)" % ("EQUALS" if same else "DIFFERS FROM", n * REP), flush=True)
assert same
run("paf from two .gz", [gz, gz2], capi.FMT_PAF, "/tmp/thm_e2e_out.paf")
run("bam from plain fastq", [big], capi.FMT_BAM, "/tmp/thm_e2e_out.bam")
run("bam from two .gz", [gz, gz2], capi.FMT_BAM, "/tmp/thm_e2e_out.bam")
a.close()
for f in (path, big, gz, gz2, "/tmp/thm_e2e_out.sam", "/tmp/thm_e2e_out.paf", "/tmp/thm_e2e_out.bam"):
    if os.path.exists(f):
        os.remove(f)
