"""-m gpu: the suffix array built on the device (csrc/sa_gpu.hip, thm_build_suffix_array_gpu) equals the host builder's
(csrc/sais.cpp), and an index created without a supplied suffix array -- which takes the device builder for texts of
4 Mi symbols and more -- holds that same array."""
import numpy as np
import pytest

from thermite_amd import capi, refdata, synth

pytestmark = pytest.mark.gpu


def _texts(data_dir):
    rng = np.random.default_rng(7)
    t = [refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")["text"],
         refdata.load_reference(data_dir + "/GRCh38-2020-A-chrM.fasta", data_dir + "/GRCh38-2020-A-chrM.gtf")["text"],
         synth.synth_reference(length=300000, n_genes=8)["text"],
         np.frombuffer(b"A", np.uint8), np.frombuffer(b"AC", np.uint8), np.frombuffer(b"AAAAAAAAAAAAAAAAAAAAA", np.uint8),
         np.frombuffer(b"NNNNACGTNNACNNNNNNNNNNNNNNNNGTTTTTNACGTACGTNN$TTGCA$", np.uint8),
         np.full(200000, ord("N"), np.uint8),  # one run: log2(n / 8) + 1 rounds
         np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 50000)].repeat(1),
         np.tile(np.frombuffer(b"ACGTTGCAAC", np.uint8), 30000)]  # period 10: every suffix ties until its end
    return [np.ascontiguousarray(x) for x in t]


def test_device_suffix_array_equals_the_host_builder(data_dir):
    for text in _texts(data_dir):
        want = capi.build_suffix_array(text)
        for wide in (False, True):
            got = capi.build_suffix_array_gpu(text, wide=wide)
            assert got.dtype.itemsize == (8 if wide else 4)
            assert np.array_equal(got.astype(np.uint64), want.astype(np.uint64)), (len(text), wide)


def test_wide_device_builder_equals_the_host_builder(data_dir, monkeypatch):
    """texts of 2^32 - 2 symbols and more take a second device builder (64-bit positions, a round = two stable sorts
    because a pair of ranks no longer fits one key); THM_SA_WIDE_SORT=1 sends small texts through it"""
    monkeypatch.setenv("THM_SA_WIDE_SORT", "1")
    for text in _texts(data_dir):
        want = capi.build_suffix_array(text)
        got = capi.build_suffix_array_gpu(text, wide=True)
        assert np.array_equal(got, want.astype(np.uint64)), len(text)


def test_index_without_a_supplied_suffix_array_builds_it_on_the_device(monkeypatch):
    t = synth.synth_reference(length=3_000_000)  # 6 M symbols: above the threshold of the device builder
    assert len(t["text"]) >= (4 << 20)
    want = capi.build_suffix_array(t["text"])
    for wide in (False, True):
        ix = capi.Index(t, wide=wide)
        assert np.array_equal(ix.suffix_array().astype(np.uint64), want.astype(np.uint64))
        assert ix.check_lut()
        ix.close()
    monkeypatch.setenv("THM_SA_HOST", "1")  # and the host builder on request
    ix = capi.Index(t)
    assert np.array_equal(ix.suffix_array().astype(np.uint64), want.astype(np.uint64))
    ix.close()
