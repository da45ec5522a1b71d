"""-m gpu: the whole-file driver (thm_align_files: FASTQ -> HIP path -> SAM / PAF) against the
oracle's alignments rendered by the oracle's writer, byte for byte."""
import gzip
import os

import numpy as np
import pytest

from oracle import aln_writer as ow
from oracle import pyoracle as orc
from thermite_amd import capi, refdata, synth

pytestmark = pytest.mark.gpu


def test_test_query_files_match_goldens(data_dir, golden_dir, tmp_path):
    """config 1 end to end from the files: data/test_ref.{fasta,gtf} + data/test_query.fastq, -k3 --min-aln-score=0"""
    ix = capi.Index.from_files(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")
    a = capi.Aligner(ix, dict(capi.DEFAULT_OPTS, min_seed_len=3, min_aln_score=0))
    for fmt, name in ((capi.FMT_SAM, "test_query.sam"), (capi.FMT_PAF, "test_query.paf")):
        out = tmp_path / name
        st = capi.align_files(a, [data_dir + "/test_query.fastq"], out, fmt, batch_reads=4, n_threads=2)
        assert open(out, "rb").read() == open(os.path.join(golden_dir, name), "rb").read()
        assert st["n_reads"] == 10 and st["n_batches"] == 3 and st["n_aligned_reads"] == 8
    # BAM: same records, binary, in BGZF blocks
    t = refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")
    names, seqs, quals = refdata.parse_fastq(data_dir + "/test_query.fastq")
    bases, off = refdata.pack_reads(seqs)
    res = orc.Index(t).align_batch(bases, off, dict(capi.DEFAULT_OPTS, min_seed_len=3, min_aln_score=0))
    out = tmp_path / "test_query.bam"
    capi.align_files(a, [data_dir + "/test_query.fastq"], out, capi.FMT_BAM, batch_reads=4, n_threads=2)
    assert ow.bgzf_decompress(open(out, "rb").read()) == ow.bam_stream(
        t, [n.encode() for n in names], [bytes(x) for x in seqs], [bytes(q) for q in quals], res)
    a.close()


@pytest.mark.parametrize("opts_name", ["ci", "default"])
def test_chrM_fastq_to_sam_matches_oracle(data_dir, tmp_path, opts_name):
    opts = capi.CI_OPTS if opts_name == "ci" else capi.DEFAULT_OPTS
    fa, gtf = data_dir + "/GRCh38-2020-A-chrM.fasta", data_dir + "/GRCh38-2020-A-chrM.gtf"
    t = refdata.load_reference(fa, gtf)
    n = 5000
    bases, off, _ = synth.simulate_reads(t, n, 91, sub_rate=0.02, indel_rate=0.004, stream=11)
    rng = np.random.default_rng(5)
    seqs = [bytes(bases[int(off[i]): int(off[i + 1])]) for i in range(n)]
    quals = [bytes(rng.integers(33, 74, len(s)).astype(np.uint8)) for s in seqs]
    names = [("r%d 1:N:0" % i).encode() for i in range(n)]
    # two input files (one gzip), as `thermite align idx a.fastq b.fastq.gz`
    half = n // 2 + 17
    p1, p2 = tmp_path / "a.fastq", tmp_path / "b.fastq.gz"
    rec = lambda i: b"@" + names[i] + b"\n" + seqs[i] + b"\n+\n" + quals[i] + b"\n"
    p1.write_bytes(b"".join(rec(i) for i in range(half)))
    with gzip.open(p2, "wb") as f:
        f.write(b"".join(rec(i) for i in range(half, n)))
    res = orc.Index(t).align_batch(bases, off, opts, n_threads=8)
    ix = capi.Index.from_files(fa, gtf)
    idx_path = tmp_path / "chrM.thmidx"
    ix.save(idx_path)
    a = capi.Aligner(capi.Index.load(idx_path), opts)  # through the index container, like `thermite align <index>`
    for fmt, key in ((capi.FMT_SAM, "sam"), (capi.FMT_PAF, "paf"), (capi.FMT_BAM, "bam")):
        out = tmp_path / ("out." + key)
        st = capi.align_files(a, [p1, p2], out, fmt, batch_reads=777, n_threads=4)
        got = open(out, "rb").read()
        if key == "bam":
            assert ow.bgzf_decompress(got) == ow.bam_stream(t, names, seqs, quals, res)
            assert st["n_output_bytes"] == len(got)
        else:
            want = (ow.sam_header(t) if key == "sam" else b"") + ow.format_batch(t, names, seqs, quals, res, key)
            assert got == want, key
            assert st["n_output_bytes"] == len(want)
        assert st["n_reads"] == n and st["n_aligned_reads"] == int(res.counters[1])
    # several aligners (one per GPU on a multi-GPU node; here three handles on the one device): the batches are dealt
    # over them, the records still leave in input order
    more = [capi.Aligner(a.index, opts) for _ in range(2)]
    out = tmp_path / "multi.sam"
    st = capi.align_files([a] + more, [p1, p2], out, capi.FMT_SAM, batch_reads=311, n_threads=6)
    assert open(out, "rb").read() == ow.sam_header(t) + ow.format_batch(t, names, seqs, quals, res, "sam")
    assert st["n_reads"] == n and st["n_batches"] >= 16
    for x in more:
        x.close()
    a.close()


def test_align_files_reports_errors(data_dir, tmp_path):
    ix = capi.Index.from_files(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")
    a = capi.Aligner(ix, capi.DEFAULT_OPTS)
    with pytest.raises(capi.ThermiteError) as e:
        capi.align_files(a, [tmp_path / "missing.fastq"], tmp_path / "o.sam", capi.FMT_SAM)
    assert e.value.code == capi.ERR_IO
    bad = tmp_path / "bad.fastq"
    bad.write_bytes(b"@r\nACGT\n+\n!!\n")
    with pytest.raises(capi.ThermiteError) as e:
        capi.align_files(a, [bad], tmp_path / "o.sam", capi.FMT_SAM)
    assert e.value.code == capi.ERR_FORMAT
    # a read this build cannot take (longer than 65535 bases) fails the run by name: the reference would align it or panic
    big = tmp_path / "big.fastq"
    big.write_bytes(b"@ok\nACGTACGTACGT\n+\n!!!!!!!!!!!!\n@giant read\n" + b"ACGT" * 17000 + b"\n+\n" + b"!" * 68000 + b"\n")
    with pytest.raises(capi.ThermiteError) as e:
        capi.align_files(a, [big], tmp_path / "o.sam", capi.FMT_SAM)
    assert e.value.code == capi.ERR_UNSUPPORTED and "giant read" in str(e.value)
    with pytest.raises(capi.ThermiteError) as e:
        capi.align_files(a, [data_dir + "/test_query.fastq"], tmp_path / "o.xyz", 7)
    assert e.value.code == capi.ERR_INVALID_ARG
    with pytest.raises(capi.ThermiteError) as e:
        capi.align_files(a, [data_dir + "/test_query.fastq"], tmp_path / "no_such_dir" / "o.sam", capi.FMT_SAM)
    assert e.value.code == capi.ERR_IO
    a.close()


def test_cpp_thermite_aligner_wrapper(data_dir, golden_dir, tmp_path):
    """include/thermite.hpp's ThermiteAligner (src/wrapper.rs:20-123) compiled and run: index file in,
    one read per call, the SAM records of config 1 out"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "wrapper_main"
    libdir = os.path.dirname(capi.SO_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "wrapper_main.cpp"), "-o", str(exe), "-L" + libdir,
                           "-lthermite_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    idx = tmp_path / "test_ref.thmidx"
    capi.Index.from_files(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf").save(idx)
    out = subprocess.run([str(exe), str(idx), "3", "0", data_dir + "/test_query.fastq"], check=True, capture_output=True)
    assert out.stdout == open(os.path.join(golden_dir, "test_query.sam"), "rb").read()
    assert ("est_mem %d" % os.path.getsize(idx)).encode() in out.stderr
    # align_read proper returns the same records without TX / GX / GN / RE (src/wrapper.rs:136-139)
    n_rec = len([ln for ln in out.stdout.split(b"\n") if ln and not ln.startswith(b"@")])
    assert ("stripped_records %d still_tagged 0" % n_rec).encode() in out.stderr


def test_blank_tail_block_and_parallel_gzip(data_dir, tmp_path, monkeypatch):
    """(a) blank lines at the end of the input that fall into a block of their own (batch_reads divides the record
    count) hold no read: nothing is uploaded for them, the output is that of the file without them -- and blank
    lines in the middle of the input are an error in both parsers alike; (b) the same records through a gzip file
    cut into many chunks for the chunk-parallel decoder give the same bytes out"""
    fa, gtf = data_dir + "/GRCh38-2020-A-chrM.fasta", data_dir + "/GRCh38-2020-A-chrM.gtf"
    t = refdata.load_reference(fa, gtf)
    n = 4000
    bases, off, _ = synth.simulate_reads(t, n, 91, sub_rate=0.02, indel_rate=0.004, stream=12)
    seqs = [bytes(bases[int(off[i]): int(off[i + 1])]) for i in range(n)]
    quals = [b"F" * len(s) for s in seqs]
    names = [("q%d" % i).encode() for i in range(n)]
    body = b"".join(b"@" + names[i] + b"\n" + seqs[i] + b"\n+\n" + quals[i] + b"\n" for i in range(n))
    res = orc.Index(t).align_batch(bases, off, capi.CI_OPTS, n_threads=8)
    want = ow.sam_header(t) + ow.format_batch(t, names, seqs, quals, res, "sam")
    a = capi.Aligner(capi.Index.from_files(fa, gtf), capi.CI_OPTS)
    p = tmp_path / "tail.fastq"
    p.write_bytes(body + b"\n\n\n")
    out = tmp_path / "tail.sam"
    st = capi.align_files(a, [p], out, capi.FMT_SAM, batch_reads=1000, n_threads=4)  # 4 full blocks, then the blank one
    assert open(out, "rb").read() == want and st["n_reads"] == n
    mid = tmp_path / "mid.fastq"
    cut = body.index(b"@q2000\n")
    mid.write_bytes(body[:cut] + b"\n\n" + body[cut:])
    with pytest.raises(capi.ThermiteError) as e:
        capi.align_files(a, [mid], tmp_path / "mid.sam", capi.FMT_SAM, batch_reads=1000, n_threads=4)
    assert e.value.code == capi.ERR_FORMAT
    monkeypatch.setenv("THM_INFLATE_CHUNK_KB", "16")
    monkeypatch.setenv("THM_INFLATE_THREADS", "3")
    gz = tmp_path / "reads.fastq.gz"
    gz.write_bytes(gzip.compress(body, 6))
    assert os.path.getsize(gz) > 4 * 16384  # (enough chunks for the parallel decoder to take the file)
    out = tmp_path / "gz.sam"
    st = capi.align_files(a, [gz, p], out, capi.FMT_SAM, batch_reads=700, n_threads=4)
    both = ow.sam_header(t) + ow.format_batch(t, names + names, seqs + seqs, quals + quals, orc.Index(t).align_batch(
        np.concatenate([bases, bases]), np.concatenate([off, off[1:] + off[-1]]), capi.CI_OPTS, n_threads=8), "sam")
    assert open(out, "rb").read() == both and st["n_reads"] == 2 * n
    a.close()
