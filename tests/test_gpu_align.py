"""-m gpu: seed-level and read-level parity of the HIP path against the CPU
oracle (Index::all_smems src/index.rs:228-255, align_read src/aligner.rs:123-190).
Results are compared as whole arrays: the canonical layouts must be byte-identical."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from thermite_amd import capi, refdata, synth

from gpu_common import assert_batch_equal

pytestmark = pytest.mark.gpu

TEST_OPTS = dict(min_seed_len=3, min_aln_score_percent=0.66, min_aln_score=0, multimap_score_range=1, intron_mode=False)


class World:
    def __init__(self, tables):
        self.t = tables
        self.ix = capi.Index(tables)
        self.oix = orc.Index(tables, sa=self.ix.suffix_array())

    def aligner(self, opts):
        return capi.Aligner(self.ix, opts)


@pytest.fixture(scope="module")
def test_ref(data_dir):
    return World(refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf"))


@pytest.fixture(scope="module")
def chrm(data_dir):
    return World(refdata.load_reference(data_dir + "/GRCh38-2020-A-chrM.fasta", data_dir + "/GRCh38-2020-A-chrM.gtf"))


@pytest.fixture(scope="module")
def syn():
    return World(synth.synth_reference(length=400000, n_genes=40))


def check_smems(w, bases, off, k):
    a = w.aligner(dict(capi.DEFAULT_OPTS, min_seed_len=k))
    g_off, g_mems = a.smems_batch(bases, off, k)
    r = w.oix.all_smems(bases, off, k)
    assert np.array_equal(g_off, r.offsets)
    for f in ("ref_idx", "query_idx", "len"):
        bad = np.nonzero(g_mems[f] != r.mems[f])[0]
        assert len(bad) == 0, "%s differs at mem %d" % (f, bad[0])
    a.close()


def check_align(w, bases, off, opts, n_threads=8):
    a = w.aligner(opts)
    a.reset_counters()
    g = a.align_batch(bases, off)
    r = w.oix.align_batch(bases, off, opts, n_threads=n_threads)
    assert r.counters[15] == 0, "oracle saw reads where the reference would panic"
    assert_batch_equal(g, r)
    c = a.counters()
    assert np.array_equal(c[:10], r.counters[:10]) and c[12] == r.counters[12], (c[:13], r.counters[:13])
    assert c[10] <= r.counters[10] and c[11] <= r.counters[11]  # DP work: exact early exit computes fewer cells
    a.close()
    return g


def test_config1_plumbing(test_ref, data_dir):
    """BASELINE config 1: data/test_query.fastq vs data/test_ref.fasta, -k3 --min-aln-score=0 (data/Makefile:21)."""
    names, seqs, _ = refdata.parse_fastq(data_dir + "/test_query.fastq")
    bases, off = refdata.pack_reads(seqs)
    check_smems(test_ref, bases, off, 3)
    g = check_align(test_ref, bases, off, TEST_OPTS, n_threads=1)
    n_alns = np.diff(g.offsets.astype(np.int64))
    got = dict(zip(names, n_alns))
    assert got["unmapped"] == 0 and got["spliced_tx1"] == 1 and got["all_match"] == 2
    check_align(test_ref, bases, off, dict(TEST_OPTS, intron_mode=True), n_threads=1)


def test_edge_reads(test_ref):
    seqs = [b"", b"A", b"ACGTNNNNACGT", b"attcgtttgatcg", b"NNNNNNNN", b"CCCCCAATCCCCCGGCCCCCTTTTCC", b"AT$TT", b"ATXTT",
            b"GGAAAAGGGGGCCGGGGGATTGGGGG"]
    bases, off = refdata.pack_reads(seqs)
    # '$' in a read matches the sentinel in the reference's FMD walk; here it is out of contract (matches nothing)
    keep = [i for i, s in enumerate(seqs) if b"$" not in s]
    b2, o2 = refdata.pack_reads([seqs[i] for i in keep])
    check_smems(test_ref, b2, o2, 3)
    check_align(test_ref, b2, o2, dict(TEST_OPTS, intron_mode=True), n_threads=1)
    a = test_ref.aligner(TEST_OPTS)
    a.align_batch(bases, off)  # must not fault
    a.close()


def test_chrm_seeds_and_reads(chrm):
    bases, off, _ = synth.simulate_reads(chrm.t, 20000, 91, sub_rate=0.01, indel_rate=0.001)
    check_smems(chrm, bases, off, 20)
    check_align(chrm, bases, off, capi.DEFAULT_OPTS)
    check_align(chrm, bases, off, capi.CI_OPTS)


def test_chrm_noisy_reads(chrm):
    bases, off, _ = synth.simulate_reads(chrm.t, 6000, 91, sub_rate=0.04, indel_rate=0.01, stream=3)
    check_smems(chrm, bases, off, 12)
    check_align(chrm, bases, off, dict(capi.CI_OPTS, min_seed_len=12))


def test_chrm_long_reads_wide_band(chrm):
    """150 bp, +-64 band (BASELINE config 5 shape): percent 0.574 -> ms 86 -> bw 64 -> 3 cells per lane."""
    bases, off, _ = synth.simulate_reads(chrm.t, 3000, 150, sub_rate=0.02, indel_rate=0.004, stream=5)
    opts = dict(min_seed_len=20, min_aln_score_percent=0.574, min_aln_score=30, multimap_score_range=1, intron_mode=True)
    check_align(chrm, bases, off, opts)


def test_reads_longer_than_255_bases(chrm):
    """300 bp reads (ragged down to 0): the position-per-byte paths of the seed stage do not apply --
    SMEM selection runs one read per wavefront over the whole work list, rows of ends exceed 128 positions --
    and reads with many SMEMs (a low k) overflow the per-thread lists of the short-read path."""
    rng = np.random.default_rng(23)
    bases, off, _ = synth.simulate_reads(chrm.t, 1500, 300, sub_rate=0.03, indel_rate=0.005, stream=9)
    reads = [bases[off[i]: off[i] + (300 if i % 3 else int(rng.integers(0, 301)))] for i in range(1500)]
    b2, o2 = refdata.pack_reads(reads)
    check_smems(chrm, b2, o2, 20)
    check_smems(chrm, b2, o2, 9)
    opts = dict(min_seed_len=20, min_aln_score_percent=0.9, min_aln_score=30, multimap_score_range=1, intron_mode=True)
    check_align(chrm, b2, o2, opts)
    # the short-read path with a low k: many SMEMs per read (more than six overflow to the wavefront kernel)
    bases, off, _ = synth.simulate_reads(chrm.t, 3000, 120, sub_rate=0.08, indel_rate=0.01, stream=10)
    check_smems(chrm, bases, off, 8)
    check_align(chrm, bases, off, dict(capi.CI_OPTS, min_seed_len=8, min_aln_score_percent=0.5))


def test_ragged_lengths(chrm):
    rng = np.random.default_rng(11)
    bases, off, _ = synth.simulate_reads(chrm.t, 4000, 120, sub_rate=0.01, stream=7)
    reads = [bases[off[i]: off[i] + int(rng.integers(0, 121))] for i in range(4000)]
    b2, o2 = refdata.pack_reads(reads)
    check_smems(chrm, b2, o2, 20)
    check_align(chrm, b2, o2, capi.CI_OPTS)


def test_synthetic_repeats_multiexon(syn):
    """spliced multi-exon transcripts on both strands + planted repeats (multi-mapping, overlap filter)."""
    bases, off, _ = synth.simulate_reads(syn.t, 20000, 91, sub_rate=0.01, indel_rate=0.001, intronic_frac=0.25)
    check_smems(syn, bases, off, 20)
    check_align(syn, bases, off, capi.DEFAULT_OPTS)
    check_align(syn, bases, off, capi.CI_OPTS)
    check_align(syn, bases, off, dict(capi.CI_OPTS, multimap_score_range=5, min_seed_len=15))


def test_split_api_matches_one_shot(chrm):
    bases, off, _ = synth.simulate_reads(chrm.t, 3000, 91, stream=9)
    a = chrm.aligner(capi.CI_OPTS)
    one = a.align_batch(bases, off)
    a.upload(bases, off)
    a.run()
    a.run()  # replay on resident inputs
    a.sync()
    two = a.fetch()
    assert np.array_equal(one.offsets, two.offsets) and np.array_equal(one.alns, two.alns) and np.array_equal(one.ops, two.ops)
    t = a.timings()
    assert t["total"] > 0 and t["extend"] > 0
    a.close()


@pytest.mark.parametrize("opts", [
    dict(min_seed_len=8, min_aln_score_percent=0.5, min_aln_score=20, multimap_score_range=0, intron_mode=True),
    dict(min_seed_len=25, min_aln_score_percent=0.8, min_aln_score=30, multimap_score_range=2, intron_mode=False),
    dict(min_seed_len=12, min_aln_score_percent=0.0, min_aln_score=0, multimap_score_range=3, intron_mode=True),
])
def test_option_sweep_with_dirty_reads(syn, opts):
    """seed length below the k-mer table width (whole-range search path), score thresholds, multimap
    ranges; reads with N, lower case and bytes outside the alphabet"""
    rng = np.random.default_rng(5)
    bases, off, _ = synth.simulate_reads(syn.t, 4000, 75, sub_rate=0.02, indel_rate=0.005, intronic_frac=0.3, stream=21)
    b = bases.copy()
    m = rng.random(len(b)) < 0.003
    b[m] = ord("N")
    m = rng.random(len(b)) < 0.0005
    b[m] = ord("R")
    lower = rng.random(len(b)) < 0.3
    b[lower] = np.where((b[lower] >= 65) & (b[lower] <= 90), b[lower] + 32, b[lower])
    check_smems(syn, b, off, opts["min_seed_len"])
    check_align(syn, b, off, opts)
