"""Host-side index construction (no GPU): suffix array by induced sorting vs the
oracle's naive sort; reference table construction on the reference's test data."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from thermite_amd import capi, refdata, synth


def _texts():
    rng = np.random.default_rng(7)
    yield b"$"
    yield b"A$"
    yield b"AAAAAAAAAAAAAAAA$"
    yield b"ACGTACGTACGTACGT$TTTT$"
    yield b"NNNNNNNNNNNNACGTNNNN$NNNNACGTNNNNNNNNNNNN$"
    for n in (10, 100, 1000, 5000):
        for alpha in (b"AC", b"ACGT", b"$ACGNT"):
            a = np.frombuffer(alpha, np.uint8)
            yield bytes(a[rng.integers(0, len(a), n)]) + b"$"
    # periodic and nested repeats
    yield (b"ACGTTGCA" * 300) + b"$" + (b"TGCAACGT" * 300) + b"$"
    yield (b"A" * 700 + b"C" + b"A" * 700) + b"$"


def test_suffix_array_matches_naive():
    for t in _texts():
        sa = capi.build_suffix_array(t)
        assert orc.suffix_array_verify(t, sa), t[:40]
        assert np.array_equal(sa, orc.suffix_array_naive(t)), t[:40]


def test_suffix_array_medium_with_n_run():
    tb = synth.synth_reference(length=200000, n_genes=10)
    sa = capi.build_suffix_array(tb["text"])
    assert orc.suffix_array_verify(tb["text"], sa)


def test_reference_tables_test_ref(data_dir):
    t = refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")
    # text layout of src/index.rs:67-101
    assert bytes(t["text"][:26]) == b"ATATTTCCCGGG$CCCGGGAAATAT$"
    assert len(t["text"]) == 2 * (12 + 12 + 26 + 26 + 4)
    # '-' strand transcript: exons mapped into the revcomp copy and reversed (src/index.rs:149-195)
    tx = t["txs"][4]
    ex = t["exons"][tx["exon_begin"]: tx["exon_begin"] + tx["n_exons"]]
    assert [(int(e["start"]), int(e["end"])) for e in ex] == [(134, 139), (143, 147), (151, 155)]
    seq = bytes(t["tx_seq"][tx["seq_off"]: tx["seq_off"] + tx["seq_len"]])
    assert seq == b"GAAAAGCCGATTG"
    ix = capi.Index(t)
    assert ix.idx_to_ref(0) == (0, 0)
    assert ix.idx_to_ref(12) == (0, 12)
    assert ix.idx_to_ref(13) == (1, 0)
    assert ix.idx_to_ref(159) == (7, 26)
    assert np.array_equal(ix.suffix_array(), orc.suffix_array_naive(t["text"]))


def test_index_rejects_bad_tables(data_dir):
    t = dict(refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf"))
    bad = dict(t)
    bad["refs"] = t["refs"].copy()
    bad["refs"]["start_idx"][1] += 1
    with pytest.raises(capi.ThermiteError):
        capi.Index(bad)
    with pytest.raises(capi.ThermiteError):
        capi.Index(t, sa=np.arange(len(t["text"]), dtype="<u4"))
    with pytest.raises(ValueError):
        refdata.build_tables([("c", np.frombuffer(b"ACGTRYACGT", np.uint8))], [], [])


# ---- 64-bit coordinates (text of 2^31 symbols and more; reference src/index.rs:103-111, 364-388) ----
def test_suffix_array_64_equals_32():
    for t in _texts():
        assert np.array_equal(capi.build_suffix_array(t, wide=True), capi.build_suffix_array(t).astype("<u8")), t[:40]
    tb = synth.synth_reference(length=150000, n_genes=8)
    assert np.array_equal(capi.build_suffix_array(tb["text"], wide=True), capi.build_suffix_array(tb["text"]).astype("<u8"))


@pytest.mark.parametrize("wide", [False, True])
def test_kmer_table_by_counting_equals_table_from_suffix_array(data_dir, wide, monkeypatch):
    """the k-mer prefix table is built by counting over the text (no suffix-array accesses); it must equal
    the table read off the suffix array -- on texts with N runs, '$' separators, both strands, and for
    every table width (THM_KT) up to one wider than the text would pick"""
    tabs = [refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf"),
            refdata.load_reference(data_dir + "/GRCh38-2020-A-chrM.fasta", data_dir + "/GRCh38-2020-A-chrM.gtf"),
            synth.synth_reference(length=120000, n_genes=6),
            refdata.build_tables([("a", np.frombuffer(b"NNNNACGTNNACNNNNNNNNNNNNNNNNGTTTTTNACGTACGTNN", np.uint8)),
                                  ("b", np.frombuffer(b"T", np.uint8)), ("c", np.frombuffer(b"NNNN", np.uint8))], [], [])]
    for kt in (None, 1, 2, 3, 5, 8, 10):
        if kt is None:
            monkeypatch.delenv("THM_KT", raising=False)
        else:
            monkeypatch.setenv("THM_KT", str(kt))
        # the counting runs on several threads for big texts: ranges that start and end inside a run, at a run's
        # first or last symbol, on separators (THM_INDEX_THREADS forces them on these small texts)
        for threads in (1, 2, 3, 7, 16):
            monkeypatch.setenv("THM_INDEX_THREADS", str(threads))
            for t in tabs:
                ix = capi.Index(t, wide=wide)
                assert ix.coord_bytes == (8 if wide else 4)
                assert ix.check_lut(), (kt, threads, len(t["text"]))
                ix.close()


def test_wide_index_tables_and_file_round_trip(data_dir, tmp_path, monkeypatch):
    t = refdata.load_reference(data_dir + "/GRCh38-2020-A-chrM.fasta", data_dir + "/GRCh38-2020-A-chrM.gtf")
    narrow = capi.Index(t)
    wide = capi.Index(t, wide=True)
    assert narrow.coord_bytes == 4 and wide.coord_bytes == 8
    assert np.array_equal(wide.suffix_array(), narrow.suffix_array().astype("<u8"))
    # a supplied suffix array of either width is accepted by either index (checked, converted)
    for sa in (narrow.suffix_array(), wide.suffix_array()):
        for w in (False, True):
            ix = capi.Index(t, sa=sa, wide=w)
            assert np.array_equal(ix.suffix_array().astype("<u8"), wide.suffix_array())
            ix.close()
    with pytest.raises(capi.ThermiteError):
        capi.Index(t, sa=np.arange(len(t["text"]), dtype="<u8"), wide=True)
    assert wide.idx_to_ref(16570) == (1, 0)
    path = tmp_path / "wide.thmidx"
    wide.save(path)
    back = capi.Index.load(path)
    assert back.coord_bytes == 8 and np.array_equal(back.suffix_array(), wide.suffix_array())
    n_path = tmp_path / "narrow.thmidx"
    narrow.save(n_path)
    assert capi.Index.load(n_path).coord_bytes == 4
    # THM_FORCE_WIDE=1 turns every new index wide (runs the wide code path on small texts)
    monkeypatch.setenv("THM_FORCE_WIDE", "1")
    forced = capi.Index(t)
    assert forced.coord_bytes == 8
    assert capi.Index.load(n_path).coord_bytes == 8
