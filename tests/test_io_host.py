"""Host-side I/O either side of the hot path (include/thermite_io.h), on the CPU:
reference ingestion vs the Python restatement (thermite_amd/refdata.py), the index
container, the FASTQ batcher, and the SAM / PAF writer vs the oracle's restatement
of src/aln_writer.rs (oracle/aln_writer.py) on alignments the CPU oracle produced.
"""
import ctypes
import gzip
import os
import re

import numpy as np
import pytest

from oracle import aln_writer as ow
from oracle import pyoracle as orc
from thermite_amd import capi, refdata, synth

REFS = [("test_ref.fasta", "test_ref.gtf"), ("GRCh38-2020-A-chrM.fasta", "GRCh38-2020-A-chrM.gtf")]


def _same(a, b):
    return np.array_equal(a, b) if isinstance(a, np.ndarray) else a == b


def test_io_header_symbols_exported():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "thermite_io.h")).read()
    declared = set(re.findall(r"\b(thm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.IO_ABI_SYMBOLS)
    L = ctypes.CDLL(capi.SO_PATH)
    for s in sorted(declared):
        assert hasattr(L, s), "missing export: " + s


@pytest.mark.parametrize("fa,gtf", REFS)
def test_native_ingestion_matches_python_restatement(data_dir, fa, gtf):
    ix = capi.Index.from_files(data_dir + "/" + fa, data_dir + "/" + gtf)
    nt = ix.native_tables()
    pt = refdata.load_reference(data_dir + "/" + fa, data_dir + "/" + gtf)
    for k in pt:
        assert _same(pt[k], nt[k]), k
    # and the search structures are the ones the in-memory constructor builds
    assert np.array_equal(ix.suffix_array(), capi.Index(pt).suffix_array())


def test_ingestion_reads_gzip_and_rejects_bad_input(data_dir, tmp_path):
    fa, gtf = data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf"
    gz = tmp_path / "ref.fasta.gz"
    with gzip.open(gz, "wb") as f:
        f.write(open(fa, "rb").read())
    a = capi.Index.from_files(gz, gtf).native_tables()
    b = capi.Index.from_files(fa, gtf).native_tables()
    assert all(_same(a[k], b[k]) for k in a)
    with pytest.raises(capi.ThermiteError) as e:
        capi.Index.from_files(tmp_path / "missing.fasta", gtf)
    assert e.value.code == capi.ERR_IO
    bad = tmp_path / "bad.fasta"
    bad.write_text(">c1\nACGTXACGT\n")
    with pytest.raises(capi.ThermiteError) as e:
        capi.Index.from_files(bad, gtf)
    assert e.value.code == capi.ERR_FORMAT
    badgtf = tmp_path / "bad.gtf"
    badgtf.write_text('some_ref\tx\texon\t1\t5\t.\t+\t.\tgene_id "g"; transcript_id "nope";\n')
    with pytest.raises(capi.ThermiteError) as e:
        capi.Index.from_files(fa, badgtf)
    assert e.value.code == capi.ERR_FORMAT


@pytest.mark.parametrize("fa,gtf", REFS)
def test_index_container_round_trip(data_dir, tmp_path, fa, gtf):
    ix = capi.Index.from_files(data_dir + "/" + fa, data_dir + "/" + gtf)
    p = tmp_path / "ref.thmidx"
    ix.save(p)
    ix2 = capi.Index.load(p)
    a, b = ix.native_tables(), ix2.native_tables()
    assert all(_same(a[k], b[k]) for k in a)
    assert np.array_equal(ix.suffix_array(), ix2.suffix_array())
    # corruption and truncation are detected, not loaded
    raw = bytearray(open(p, "rb").read())
    raw[len(raw) // 2] ^= 0x40
    (tmp_path / "corrupt.thmidx").write_bytes(raw)
    with pytest.raises(capi.ThermiteError) as e:
        capi.Index.load(tmp_path / "corrupt.thmidx")
    assert e.value.code == capi.ERR_FORMAT
    (tmp_path / "short.thmidx").write_bytes(raw[: len(raw) - 7])
    with pytest.raises(capi.ThermiteError) as e:
        capi.Index.load(tmp_path / "short.thmidx")
    assert e.value.code == capi.ERR_FORMAT
    (tmp_path / "other.thmidx").write_bytes(b"not an index at all" * 20)
    with pytest.raises(capi.ThermiteError):
        capi.Index.load(tmp_path / "other.thmidx")


def _collect(reader, batch):
    names, seqs, quals = [], [], []
    while True:
        b = reader.next_batch(batch)
        if b is None:
            break
        for i in range(len(b["offsets"]) - 1):
            s, e = int(b["offsets"][i]), int(b["offsets"][i + 1])
            seqs.append(bytes(b["bases"][s:e]))
            quals.append(None if b["quals"] is None else bytes(b["quals"][s:e]))
            names.append(bytes(b["names"][int(b["name_off"][i]): int(b["name_off"][i + 1])]))
    return names, seqs, quals


@pytest.mark.parametrize("batch", [1, 3, 1000])
def test_fastq_batcher(data_dir, tmp_path, batch):
    path = data_dir + "/test_query.fastq"
    names, seqs, quals = refdata.parse_fastq(path)
    got = _collect(capi.FastqReader(path), batch)
    assert got[0] == [n.encode() for n in names]
    assert got[1] == [bytes(s) for s in seqs]
    assert got[2] == [bytes(q) for q in quals]
    gz = tmp_path / "q.fastq.gz"
    with gzip.open(gz, "wb") as f:
        f.write(open(path, "rb").read().replace(b"\n", b"\r\n"))  # CRLF line ends are stripped too
    assert _collect(capi.FastqReader(gz), batch) == got


def test_fastq_batcher_edge_cases(tmp_path):
    p = tmp_path / "e.fastq"
    p.write_bytes(b"@empty read\n\n+\n\n@r2 extra words\nACGTN\n+r2\n!!!!#\n")
    names, seqs, quals = _collect(capi.FastqReader(p), 10)
    assert names == [b"empty read", b"r2 extra words"] and seqs == [b"", b"ACGTN"] and quals == [b"", b"!!!!#"]
    fa = tmp_path / "q.fasta"
    fa.write_bytes(b">a desc\nACG\nTTA\n>b\nGG\n")
    names, seqs, quals = _collect(capi.FastqReader(fa), 10)
    assert names == [b"a desc", b"b"] and seqs == [b"ACGTTA", b"GG"] and quals == [None, None]
    for bad in (b"@r\nACGT\n+\n!!\n", b"@r\nACGT\nACGT\n!!!!\n", b"ACGT\n", b"@r\nACGT\n"):
        q = tmp_path / "bad.fastq"
        q.write_bytes(bad)
        with pytest.raises(capi.ThermiteError) as e:
            _collect(capi.FastqReader(q), 10)
        assert e.value.code == capi.ERR_FORMAT
    with pytest.raises(capi.ThermiteError) as e:
        capi.FastqReader(tmp_path / "nope.fastq")
    assert e.value.code == capi.ERR_IO


def test_multimapq_table():
    """the table in the doc comment of multimapq, reference src/aln_writer.rs:327-331"""
    assert [ow.multimapq(n) for n in (0, 1, 2, 3, 4, 5, 6, 100)] == [255, 255, 3, 2, 1, 0, 0, 0]


def test_cigar_run_length_rules():
    """to_noodles_cigar, reference src/aln_writer.rs:279-323"""
    assert ow.to_cigar([]) == "*"
    assert ow.to_cigar(["Match", "Subst", "Match", "Ins", "Ins", "Del", "Match", ("Xclip", 7)]) == "3M2I1D1M7S"
    assert ow.to_cigar([("Xclip", 2), "Match", ("Yclip", 40), "Subst", "Subst"]) == "2S1M40N2M"
    # equal neighbouring clips form one run that keeps the clip's own length; unequal ones do not merge
    assert ow.to_cigar([("Yclip", 5), ("Yclip", 5), "Match"]) == "5N1M"
    assert ow.to_cigar([("Yclip", 5), ("Yclip", 6), "Match"]) == "5N6N1M"


def _batch(names, seqs, quals):
    bases, off = refdata.pack_reads(seqs)
    nb, noff = refdata.pack_reads([n if isinstance(n, bytes) else n.encode() for n in names])
    qb, _ = refdata.pack_reads(quals)
    return dict(bases=bases, offsets=off, quals=qb, names=nb, name_off=noff)


def _check_writer(tables, names, seqs, quals, res):
    ix = capi.Index(tables)
    batch = _batch(names, seqs, quals)
    sam = None
    for fmt, key in ((capi.FMT_SAM, "sam"), (capi.FMT_PAF, "paf")):
        want = ow.format_batch(tables, [n if isinstance(n, bytes) else n.encode() for n in names], seqs, quals, res, key)
        for threads in (1, 3):
            w = capi.Writer(ix, fmt, threads)
            got = w.format_batch(batch, res)
            assert got == want, (key, threads)
            if fmt == capi.FMT_SAM:
                assert w.header() == ow.sam_header(tables)
                sam = want
            else:
                assert w.header() == b""
            assert w.trailer() == b""
    # BAM: the decompressed stream (header, records) must equal the oracle's; BGZF framing is checked on the way
    bn = [n if isinstance(n, bytes) else n.encode() for n in names]
    want = ow.bam_stream(tables, bn, seqs, quals, res)
    for threads in (1, 3):
        w = capi.Writer(ix, capi.FMT_BAM, threads)
        blob = w.header() + w.format_batch(batch, res) + w.trailer()
        assert ow.bgzf_decompress(blob) == want, ("bam", threads)
    return sam


def test_writer_on_reference_test_query(data_dir):
    """config 1: data/test_query.fastq vs data/test_ref.*, flags of reference data/Makefile:21"""
    t = refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")
    names, seqs, quals = refdata.parse_fastq(data_dir + "/test_query.fastq")
    bases, off = refdata.pack_reads(seqs)
    opts = dict(capi.DEFAULT_OPTS, min_seed_len=3, min_aln_score=0)
    res = orc.Index(t).align_batch(bases, off, opts)
    text = _check_writer(t, names, [bytes(s) for s in seqs], [bytes(q) for q in quals], res)
    lines = text.split(b"\n")
    assert any(b"\t4\t*\t0\t255\t*\t*\t0\t0\t" in ln for ln in lines)  # the read named "unmapped"
    assert any(b"N" in ln.split(b"\t")[5] and b"RE:A:E" in ln for ln in lines if ln)  # spliced reads carry an N


@pytest.mark.parametrize("opts", [capi.CI_OPTS, capi.DEFAULT_OPTS])
def test_writer_on_chrM_reads(data_dir, opts):
    t = refdata.load_reference(data_dir + "/GRCh38-2020-A-chrM.fasta", data_dir + "/GRCh38-2020-A-chrM.gtf")
    bases, off, _ = synth.simulate_reads(t, 600, 91, sub_rate=0.02, indel_rate=0.004, stream=7)
    seqs = [bytes(bases[int(off[i]): int(off[i + 1])]) for i in range(len(off) - 1)]
    rng = np.random.default_rng(3)
    quals = [bytes(rng.integers(33, 74, len(s)).astype(np.uint8)) for s in seqs]
    names = ["read%d with a comment" % i if i % 3 else "read%d" % i for i in range(len(seqs))]
    res = orc.Index(t).align_batch(bases, off, opts)
    text = _check_writer(t, names, seqs, quals, res)
    assert text.count(b"\n") >= 500


def test_writer_on_constructed_records(data_dir):
    """hand-made results: multimap counts 0..6, both strands, secondary flags, clip runs, an empty read"""
    t = refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")
    op_lists = [["Match"] * 4, [("Xclip", 2), "Match", "Subst", ("Yclip", 4), "Match", "Del", "Ins", ("Xclip", 1)], []]
    alns, ops, offsets = [], bytearray(), [0]
    seqs, quals, names = [], [], []
    for n in range(0, 7):
        seqs.append(b"ACGTNacgtRY"[: 4 + n])
        quals.append(bytes(range(40, 44 + n)))
        names.append("q%d some comment" % n)
        for j in range(n):
            a = np.zeros(1, capi.ALN_DT)[0]
            o = orc.encode_ops(op_lists[(n + j) % 3])
            a["ops_off"], a["ops_len"] = len(ops), len(o)
            ops += o
            to = orc.encode_ops(op_lists[(n + j + 1) % 3])
            a["tx_ops_off"], a["tx_ops_len"] = len(ops), len(to)
            ops += to
            a["ref_id"] = (n + j) % len(t["refs"])
            a["strand"] = t["refs"][a["ref_id"]]["strand"]
            a["primary"] = 1 if j == 0 else 0
            a["aln_type"] = (n + j) % 3
            a["tx_or_gene_idx"] = 0 if a["aln_type"] != 2 else 0xFFFFFFFF
            a["score"], a["ystart"], a["yend"], a["ylen"] = 7 - j, 3 + j, 9 + j, 26
            a["xstart"], a["xend"], a["xlen"], a["tx_ystart"] = j % 2, 4 + n, 4 + n, 11 * j
            alns.append(a)
        offsets.append(len(alns))
    seqs.append(b"")
    quals.append(b"")
    names.append("empty")
    offsets.append(len(alns))

    class R:
        pass

    res = R()
    res.n_reads = len(seqs)
    res.offsets = np.array(offsets, "<u8")
    res.alns = np.array(alns, capi.ALN_DT)
    res.ops = np.frombuffer(bytes(ops), np.uint8).copy()
    text = _check_writer(t, names, seqs, quals, res)
    mapq = [ln.split(b"\t")[4] for ln in text.split(b"\n") if ln and not ln.startswith(b"q0") and not ln.startswith(b"empty")]
    assert mapq[:1] == [b"255"] and mapq[1:3] == [b"3", b"3"] and mapq[-6:] == [b"0"] * 6


def test_oracle_writer_reproduces_committed_sam_paf(data_dir, golden_dir):
    """tests/golden/test_query.{sam,paf} (made by tests/golden/make_goldens.py) pin the restated writer"""
    t = refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")
    names, seqs, quals = refdata.parse_fastq(data_dir + "/test_query.fastq")
    bases, off = refdata.pack_reads(seqs)
    res = orc.Index(t).align_batch(bases, off, dict(capi.DEFAULT_OPTS, min_seed_len=3, min_aln_score=0))
    nb, sb, qb = [n.encode() for n in names], [bytes(s) for s in seqs], [bytes(q) for q in quals]
    assert ow.sam_header(t) + ow.format_batch(t, nb, sb, qb, res, "sam") == open(golden_dir + "/test_query.sam", "rb").read()
    assert ow.format_batch(t, nb, sb, qb, res, "paf") == open(golden_dir + "/test_query.paf", "rb").read()


# ---- the whole-file driver's parallel parsing: the block cutter and the block parser against the sequential parser ----
@pytest.mark.parametrize("per_block", [1, 2, 7, 1000])
def test_fastq_block_parser_equals_sequential_parser(data_dir, tmp_path, per_block):
    rng = np.random.default_rng(5)
    recs = []
    for i in range(533):
        L = int(rng.integers(0, 160))
        seq = bytes(np.frombuffer(b"ACGTNacgt", np.uint8)[rng.integers(0, 9, L)])
        qual = bytes(rng.integers(33, 74, L).astype(np.uint8))  # includes '@' and '+' as quality characters
        recs.append(b"@r%d some words %d\n" % (i, i * 7) + seq + b"\n+\n" + qual + b"\n")
    variants = {"plain": b"".join(recs), "no_final_newline": b"".join(recs)[:-1], "crlf": b"".join(recs).replace(b"\n", b"\r\n"),
                "blank_tail": b"".join(recs) + b"\n\n"}
    for name, data in variants.items():
        for gz in (False, True):
            p = tmp_path / ("%s.fastq%s" % (name, ".gz" if gz else ""))
            (gzip.open(p, "wb") if gz else open(p, "wb")).write(data)
            want = _collect(capi.FastqReader(p), 100)
            b = capi.FastqReader(p).all_by_blocks(per_block)
            got = _collect_one(b)
            assert got == want, (name, gz)
    # malformed records are errors in the block parser too -- also when the sequence line is most of the block (the
    # parser copies a sequence before it has seen the record's '+' and quality lines: round-2 advisor finding, a heap
    # overflow for sequences beyond half the block) -- and in the sequential parser, which accepts the same files
    big = b"A" * 4000000
    for bad in (b"@r\nACGT\n+\n!!\n", b"@r\nACGT\nACGT\n!!!!\n", b"@r\nACGT\n", b"@a\nAC\n+\n!!\n\n@b\nAC\n+\n!!\n",
                b"@r0\n" + big + b"\n+\nII\n",            # quality much shorter than a long sequence
                b"@r0\nAC\n+\nII\n@r1\n" + big + b"\n",  # truncated last record of a long read
                b"@r0\n" + big + b"\n" + big + b"\n",       # no '+' after a long line
                b"@r0\n" + b"A" * 9000 + b"\n+\n" + b"I" * 8999 + b"\n"):
        q = tmp_path / "bad.fastq"
        q.write_bytes(bad)
        with pytest.raises(capi.ThermiteError) as e:
            capi.FastqReader(q).all_by_blocks(per_block)
        assert e.value.code == capi.ERR_FORMAT
        with pytest.raises(capi.ThermiteError) as e:
            _collect(capi.FastqReader(q), 100)
        assert e.value.code == capi.ERR_FORMAT
    # a well-formed record of a long read is fine in both
    q = tmp_path / "long.fastq"
    q.write_bytes(b"@r0\n" + big + b"\n+\n" + b"I" * len(big) + b"\n@r1\nAC\n+\nII\n")
    assert _collect_one(capi.FastqReader(q).all_by_blocks(per_block)) == _collect(capi.FastqReader(q), 100)


def _collect_one(b):
    names, seqs, quals = [], [], []
    for i in range(len(b["offsets"]) - 1):
        s, e = int(b["offsets"][i]), int(b["offsets"][i + 1])
        seqs.append(bytes(b["bases"][s:e]))
        quals.append(bytes(b["quals"][s:e]))
        names.append(bytes(b["names"][int(b["name_off"][i]): int(b["name_off"][i + 1])]))
    return names, seqs, quals


def test_fastq_read_errors_are_not_end_of_file(data_dir, tmp_path):
    """a truncated or corrupt gzip stream is THM_ERR_IO, never a silently shorter input (needletail returns the error,
    reference src/aligner.rs:52-55)"""
    raw = open(data_dir + "/test_query.fastq", "rb").read() * 400
    good = tmp_path / "g.fastq.gz"
    with gzip.open(good, "wb") as f:
        f.write(raw)
    z = open(good, "rb").read()
    cut = tmp_path / "cut.fastq.gz"
    cut.write_bytes(z[: len(z) // 2])
    corrupt = tmp_path / "bad.fastq.gz"
    zb = bytearray(z)
    for k in range(len(zb) // 3, len(zb) // 3 + 64):
        zb[k] ^= 0x5A
    corrupt.write_bytes(bytes(zb))
    n_good = sum(len(x) for x in [_collect(capi.FastqReader(good), 1000)[0]])
    assert n_good == 4000
    for p in (cut, corrupt):
        for how in ("seq", "blocks"):
            with pytest.raises(capi.ThermiteError) as e:
                if how == "seq":
                    _collect(capi.FastqReader(p), 1000)
                else:
                    capi.FastqReader(p).all_by_blocks(500)
            assert e.value.code == capi.ERR_IO, (p, how)


def test_index_load_rejects_absurd_header_counts(data_dir, tmp_path):
    """header counts larger than the file (whose products would wrap around) are a format error, not an allocation"""
    import struct
    t = refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")
    ix = capi.Index(t)
    p = tmp_path / "i.thmidx"
    ix.save(p)
    raw = bytearray(open(p, "rb").read())
    for field in range(1, 9):  # n_text .. names_bytes
        for val in (2**63, 2**61 + 12345, 2**40):
            bad = bytearray(raw)
            bad[8 * field: 8 * field + 8] = struct.pack("<Q", val)
            q = tmp_path / "bad.thmidx"
            q.write_bytes(bytes(bad))
            with pytest.raises(capi.ThermiteError) as e:
                capi.Index.load(q)
            assert e.value.code == capi.ERR_FORMAT


# ---- the library's own inflate (csrc/io_inflate.cpp) against Python's gzip / zlib ----
def _gz_cases():
    rng = np.random.default_rng(11)
    fastq = b"".join(b"@r%d x\n" % i + bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 91)]) + b"\n+\n" + b"F" * 91 + b"\n" for i in range(20000))
    noise = rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()
    skew = bytes(rng.choice(np.arange(256, dtype=np.uint8), 400000, p=np.r_[np.full(4, 0.2), np.full(252, 0.2 / 252)]))  # long and short codes
    runs = b"".join(bytes([int(c)]) * int(n) for c, n in zip(rng.integers(65, 70, 3000), rng.integers(1, 700, 3000)))  # distances 1..8, lengths to 258
    periodic = b"".join((b"abcdefghijklmnopqrstuvwxyz"[: int(k)]) * 40 for k in rng.integers(1, 26, 400))  # every short distance
    return {"fastq": fastq, "noise": noise, "skew": skew, "runs": runs, "periodic": periodic, "empty": b"", "one": b"A", "short": b"hello hello hello hello\n"}


@pytest.mark.parametrize("chunk", [1024, 4096, 1 << 20])
def test_inflate_equals_zlib(tmp_path, chunk):
    import zlib
    for name, data in _gz_cases().items():
        for level in (0, 1, 6, 9):
            p = tmp_path / ("%s_%d.gz" % (name, level))
            p.write_bytes(gzip.compress(data, level))
            assert capi.debug_gunzip(p, chunk) == data, (name, level, chunk)
        # fixed-Huffman blocks, and a stream cut into many blocks by full flushes
        for strategy, tag in ((zlib.Z_FIXED, "fixed"), (zlib.Z_DEFAULT_STRATEGY, "flushed")):
            co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, strategy)
            z = b""
            for s in range(0, len(data), 7001):
                z += co.compress(data[s : s + 7001])
                if tag == "flushed":
                    z += co.flush(zlib.Z_FULL_FLUSH if (s // 7001) % 2 else zlib.Z_SYNC_FLUSH)
            z += co.flush()
            p = tmp_path / ("%s_%s.gz" % (name, tag))
            p.write_bytes(z)
            assert capi.debug_gunzip(p, chunk) == data, (name, tag, chunk)


def test_inflate_members_and_headers(tmp_path):
    import io, struct, zlib
    cases = _gz_cases()
    # several members back to back (bgzip, `cat a.gz b.gz`): the stream is their concatenation
    p = tmp_path / "members.gz"
    p.write_bytes(b"".join(gzip.compress(cases[k], lv) for k, lv in (("fastq", 1), ("empty", 6), ("noise", 6), ("short", 9), ("fastq", 6))))
    want = cases["fastq"] + cases["noise"] + cases["short"] + cases["fastq"]
    for chunk in (1024, 1 << 20):
        assert capi.debug_gunzip(p, chunk) == want
    # every optional header field: FEXTRA, FNAME, FCOMMENT, FHCRC
    body = zlib.compress(cases["fastq"], 6)[2:-4]
    hdr = bytes([0x1F, 0x8B, 8, 4 | 8 | 16, 0, 0, 0, 0, 0, 3]) + struct.pack("<H", 6) + b"BC\x02\x00\x12\x34" + b"name.fastq\0" + b"a comment\0"
    hdr_crc = bytes([0x1F, 0x8B, 8, 2 | 4 | 8 | 16]) + hdr[4:]
    hdr_crc += struct.pack("<H", zlib.crc32(hdr_crc) & 0xFFFF)
    tail = struct.pack("<II", zlib.crc32(cases["fastq"]), len(cases["fastq"]) & 0xFFFFFFFF)
    for k, h in enumerate((hdr, hdr_crc)):
        p = tmp_path / ("hdr%d.gz" % k)
        p.write_bytes(h + body + tail)
        assert gzip.decompress(p.read_bytes()) == cases["fastq"]
        assert capi.debug_gunzip(p) == cases["fastq"]
    # bytes behind the last member that are no member are ignored, as gzread does (zero padding of a tape block)
    p = tmp_path / "padded.gz"
    p.write_bytes(gzip.compress(cases["short"]) + b"\0" * 512)
    assert capi.debug_gunzip(p) == cases["short"]


def test_inflate_bad_streams_are_errors(tmp_path):
    import struct
    data = _gz_cases()["fastq"]
    z = gzip.compress(data, 6)
    bad = {}
    for cut in (5, 10, 11, 40, len(z) // 3, len(z) - 9, len(z) - 8, len(z) - 1):
        bad["cut%d" % cut] = z[:cut]
    bad["crc"] = z[:-8] + struct.pack("<I", (struct.unpack("<I", z[-8:-4])[0] ^ 1)) + z[-4:]
    bad["isize"] = z[:-4] + struct.pack("<I", len(data) + 1)
    bad["method"] = z[:2] + b"\x07" + z[3:]
    bad["reserved_flags"] = z[:3] + b"\x20" + z[4:]
    bad["second_member_cut"] = z + z[: len(z) // 2]
    bad["not_gzip"] = b"@r0\nACGT\n+\nFFFF\n"
    rng = np.random.default_rng(3)
    for k in range(40):  # bit flips anywhere in the deflate stream: caught by the stream's structure or by the CRC
        zb = bytearray(z)
        at = int(rng.integers(10, len(z) - 8))
        zb[at] ^= 1 << int(rng.integers(0, 8))
        bad["flip%d" % k] = bytes(zb)
    for name, b in bad.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(b)
        with pytest.raises(capi.ThermiteError) as e:
            capi.debug_gunzip(p, 4096)
        assert e.value.code == capi.ERR_IO, name
    # a stored block's LEN / NLEN pair
    zs = bytearray(gzip.compress(data[:1000], 0))
    zs[13] ^= 0xFF
    p = tmp_path / "stored.gz"
    p.write_bytes(bytes(zs))
    with pytest.raises(capi.ThermiteError):
        capi.debug_gunzip(p)


def _big_fastq(n, seed):
    rng = np.random.default_rng(seed)
    reads = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, 91))]
    quals = rng.choice(np.frombuffer(b"FFFFFFFF:,#", np.uint8), (n, 91))
    return b"".join(b"@A00123:45:HXX:1:1101:%d:%d 1:N:0:ACGT\n" % (1000 + i % 30000, 1000 + i // 7) + reads[i].tobytes() + b"\n+\n" + quals[i].tobytes() + b"\n"
                    for i in range(n))


@pytest.mark.parametrize("threads", [2, 5])
def test_parallel_inflate_equals_zlib(tmp_path, monkeypatch, threads):
    """the chunk-parallel decoder (several threads on one gzip file): decoding that starts in the middle of the stream,
    markers for the unknown window, resolved in stream order -- on streams cut into many small chunks"""
    import zlib
    monkeypatch.setenv("THM_INFLATE_CHUNK_KB", "64")
    fq = _big_fastq(30000, 21)
    rng = np.random.default_rng(22)
    noise = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()  # incompressible: stored blocks inside level-6 output
    cases = {"fastq": fq, "mixed": fq[: 1 << 20] + noise + fq[1 << 20 :]}
    for name, data in cases.items():
        for level in (1, 6, 9):
            p = tmp_path / ("%s_%d.gz" % (name, level))
            p.write_bytes(gzip.compress(data, level))
            assert os.path.getsize(p) > 4 * 65536
            for chunk in (4096, 1 << 20):
                assert capi.debug_gunzip(p, chunk, threads=threads) == data, (name, level, chunk)
    # members back to back: one per 60 KB of input (what bgzip writes), a few large ones, an empty one in between
    members = b"".join(gzip.compress(fq[s : s + 60000], 6) for s in range(0, len(fq), 60000))
    p = tmp_path / "bgzf_like.gz"
    p.write_bytes(members)
    assert capi.debug_gunzip(p, 1 << 20, threads=threads) == fq
    p = tmp_path / "members.gz"
    p.write_bytes(gzip.compress(fq, 1) + gzip.compress(b"", 6) + gzip.compress(noise, 6) + gzip.compress(fq, 9))
    assert capi.debug_gunzip(p, 1 << 16, threads=threads) == fq + noise + fq
    # no dynamic block anywhere (level 0: stored blocks only): no chunk has a start, one segment takes it all
    p = tmp_path / "stored.gz"
    p.write_bytes(gzip.compress(fq[: 2 << 20], 0))
    assert capi.debug_gunzip(p, 1 << 20, threads=threads) == fq[: 2 << 20]
    # fixed-Huffman blocks only
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    p = tmp_path / "fixed.gz"
    p.write_bytes(co.compress(fq[: 1 << 20]) + co.flush())
    assert capi.debug_gunzip(p, 1 << 20, threads=threads) == fq[: 1 << 20]


def test_parallel_inflate_bad_streams_are_errors(tmp_path, monkeypatch):
    import struct
    monkeypatch.setenv("THM_INFLATE_CHUNK_KB", "64")
    data = _big_fastq(30000, 23)
    z = gzip.compress(data, 6)
    assert len(z) > 6 * 65536
    bad = {"cut_half": z[: len(z) // 2], "cut_tail": z[:-5], "cut_trailer": z[:-8],
           "crc": z[:-8] + struct.pack("<I", struct.unpack("<I", z[-8:-4])[0] ^ 0x10) + z[-4:],
           "isize": z[:-4] + struct.pack("<I", len(data) + 3),
           "second_member_cut": z + z[: len(z) // 3]}
    rng = np.random.default_rng(4)
    for k in range(24):  # a flipped bit anywhere: some segment does not decode, or does not end on the next start, or the CRC differs
        zb = bytearray(z)
        at = int(rng.integers(10, len(z) - 8))
        zb[at] ^= 1 << int(rng.integers(0, 8))
        bad["flip%d" % k] = bytes(zb)
    for name, b in bad.items():
        p = tmp_path / (name + ".gz")
        p.write_bytes(b)
        for threads in (1, 4):
            with pytest.raises(capi.ThermiteError) as e:
                capi.debug_gunzip(p, 1 << 16, threads=threads)
            assert e.value.code == capi.ERR_IO, (name, threads)


def test_parallel_inflate_feeds_the_fastq_parsers(tmp_path, monkeypatch):
    """the FASTQ reader over the chunk-parallel decoder (THM_INFLATE_THREADS) gives the records of the plain file"""
    monkeypatch.setenv("THM_INFLATE_CHUNK_KB", "64")
    monkeypatch.setenv("THM_INFLATE_THREADS", "3")
    fq = _big_fastq(20000, 25)
    plain = tmp_path / "r.fastq"
    plain.write_bytes(fq)
    gz = tmp_path / "r.fastq.gz"
    gz.write_bytes(gzip.compress(fq, 6))
    want = _collect(capi.FastqReader(plain), 3000)
    got = _collect(capi.FastqReader(gz), 3000)
    assert want == got and len(want[0]) == 20000
    blocks = capi.FastqReader(gz).all_by_blocks(1700)
    flat = capi.FastqReader(plain).all_by_blocks(1700)
    assert all(np.array_equal(blocks[k], flat[k]) for k in blocks)


def test_own_deflate_inflates_to_the_input(tmp_path):
    """the BAM writer's deflate (csrc/io_deflate.cpp): every block inflates to its input with zlib and with the
    library's own inflate; incompressible input becomes a stored block; skewed alphabets exercise the length limit
    of the Huffman codes"""
    import zlib
    rng = np.random.default_rng(31)
    fq = _big_fastq(400, 32)
    p = np.r_[np.full(3, 0.3), 0.1 * (0.5 ** np.arange(1, 254))]
    p[-1] += 1.0 - p.sum()  # a geometric tail: Huffman depths beyond 15 before the limit is applied
    cases = {
        "empty": b"", "one": b"A", "short": b"abcabcabcabc", "fastq": fq[:65280], "zeros": bytes(65280),
        "noise": rng.integers(0, 256, 65280, dtype=np.uint8).tobytes(),
        "skew": bytes(rng.choice(np.arange(256, dtype=np.uint8), 65280, p=p)),
        "runs": b"".join(bytes([int(c)]) * int(n) for c, n in zip(rng.integers(65, 70, 400), rng.integers(1, 700, 400)))[:65280],
        "far": (rng.integers(0, 256, 30000, dtype=np.uint8).tobytes() * 3)[:65280],  # matches at distance 30000
        "two_symbols": bytes(rng.choice(np.frombuffer(b"AB", np.uint8), 5000)),
    }
    for k in range(12):  # BAM-like records: binary fields, packed bases, runs of equal qualities
        n = int(rng.integers(1, 65281))
        cases["mixed%d" % k] = bytes(np.where(rng.random(n) < 0.5, rng.integers(0, 256, n), 70).astype(np.uint8))
    for name, data in cases.items():
        z = capi.debug_deflate_block(data)
        assert zlib.decompress(z, -15) == data, name
        if name == "noise":
            assert len(z) == len(data) + 5  # stored
        if name in ("zeros", "fastq", "runs"):
            assert len(z) < len(data) // 2, (name, len(z))
        # ... and through the gzip reader of this library (header + block + trailer)
        import struct
        g = tmp_path / (name + ".gz")
        g.write_bytes(bytes([0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 0, 0xFF]) + z + struct.pack("<II", zlib.crc32(data), len(data)))
        assert capi.debug_gunzip(g) == data, name


def test_inflate_refill_right_before_a_byte_aligned_field(tmp_path):
    """the streaming input buffer is refilled when fewer than 32 compressed bytes are left; a stored block, a tiny
    final block or the member trailer right behind that point make the decoder hand back the whole bytes it holds in
    its bit buffer -- which must still be in the buffer (found by tools/stress_gzip.py: 'invalid stored block
    lengths' on a valid file)"""
    import zlib
    rng = np.random.default_rng(41)
    body = b"".join(bytes([int(c)]) * int(k) for c, k in zip(rng.integers(60, 70, 300), rng.integers(1, 900, 300)))
    for tail in range(0, 24):
        for flush in (zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, None):
            co = zlib.compressobj(6, zlib.DEFLATED, 31)
            z = co.compress(body)
            if flush is not None:
                z += co.flush(flush)
            data = body + bytes(rng.integers(65, 91, tail).astype(np.uint8))
            z += co.compress(data[len(body):]) + co.flush()
            p = tmp_path / "t.gz"
            p.write_bytes(z)
            for chunk in (1024, 1 << 20):
                assert capi.debug_gunzip(p, chunk) == data, (tail, flush, chunk)


def test_parallel_inflate_reader_closed_early(tmp_path, monkeypatch):
    """a reader over the chunk-parallel decoder that is closed after the first batch joins its worker threads and
    leaves nothing behind (finds, decodes and resolves of later chunks are still queued or running at that point)"""
    import threading
    monkeypatch.setenv("THM_INFLATE_CHUNK_KB", "16")
    monkeypatch.setenv("THM_INFLATE_THREADS", "4")
    gz = tmp_path / "r.fastq.gz"
    gz.write_bytes(gzip.compress(_big_fastq(30000, 27), 6))
    before = threading.active_count()
    n_os = len(os.listdir("/proc/self/task"))
    for _ in range(5):
        r = capi.FastqReader(gz)
        b = r.next_batch(100)
        assert len(b["offsets"]) == 101
        r.close()
    assert len(os.listdir("/proc/self/task")) <= n_os and threading.active_count() == before


def test_parallel_inflate_gives_up_on_endless_segments(tmp_path, monkeypatch):
    """a stream in which no block start can be found (stored blocks only, fixed-Huffman blocks only) would be decoded as
    one segment of 16-bit symbols as long as the whole output: beyond a bound the parallel decoder hands the file to
    the serial one -- same bytes, constant memory"""
    import zlib
    monkeypatch.setenv("THM_INFLATE_CHUNK_KB", "16")
    monkeypatch.setenv("THM_INFLATE_MAX_SEGMENT_MB", "1")
    fq = _big_fastq(30000, 29)
    p = tmp_path / "stored.gz"
    p.write_bytes(gzip.compress(fq, 0))
    assert capi.debug_gunzip(p, 1 << 16, threads=3) == fq
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    p = tmp_path / "fixed.gz"
    p.write_bytes(co.compress(fq) + co.flush())
    assert capi.debug_gunzip(p, 1 << 16, threads=3) == fq
    # ... and a normal stream under the same bound still decodes in parallel segments or serially, to the same bytes
    p = tmp_path / "dyn.gz"
    p.write_bytes(gzip.compress(fq, 6))
    assert capi.debug_gunzip(p, 1 << 16, threads=3) == fq
