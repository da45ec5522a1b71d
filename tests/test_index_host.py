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
