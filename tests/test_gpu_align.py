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
    def __init__(self, tables, wide=False):
        self.t = tables
        self.ix = capi.Index(tables, wide=wide)
        assert self.ix.coord_bytes == (8 if wide else 4)
        self.oix = orc.Index(tables, sa=self.ix.suffix_array())

    def aligner(self, opts):
        return capi.Aligner(self.ix, opts)


# Every test of this file runs twice: on an index with 32-bit text positions and ranks (c32, the default for texts
# below 2^31 symbols) and on one with 64-bit ones (c64: the code path a GRCh38-sized text takes -- BASELINE
# config 5's reference, reference src/index.rs:364-388 -- forced here on small texts).  Results must not
# depend on the width.
WIDTHS = pytest.mark.parametrize("wide", [False, True], ids=["c32", "c64"])
_worlds = {}


def _world(key, make, wide):
    if (key, wide) not in _worlds:
        _worlds[(key, wide)] = World(make(), wide)
    return _worlds[(key, wide)]


@pytest.fixture(params=[False, True], ids=["c32", "c64"])
def test_ref(request, data_dir):
    return _world("test_ref", lambda: refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf"), request.param)


@pytest.fixture(params=[False, True], ids=["c32", "c64"])
def chrm(request, data_dir):
    return _world("chrm", lambda: refdata.load_reference(data_dir + "/GRCh38-2020-A-chrM.fasta", data_dir + "/GRCh38-2020-A-chrM.gtf"),
                  request.param)


@pytest.fixture(params=[False, True], ids=["c32", "c64"])
def syn(request):
    return _world("syn", lambda: synth.synth_reference(length=400000, n_genes=40), request.param)


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
    """Both device paths against the oracle: the problem-parallel path (kernels_tpr.hip: thread-per-read control
    kernel + wave-per-request DP kernel; what it leaves goes to the wave-per-read kernels) and the wave-per-read
    kernels alone."""
    r = w.oix.align_batch(bases, off, opts, n_threads=n_threads)
    assert r.counters[15] == 0, "oracle saw reads where the reference would panic"
    g = None
    for no_tpr in (False, True):
        a = w.aligner(opts)
        a.debug_set_flags(tpr=not no_tpr)
        a.reset_counters()
        g = a.align_batch(bases, off)
        assert g.n_failed == 0 and g.status is None
        assert_batch_equal(g, r)
        c = a.counters()
        assert np.array_equal(c[:10], r.counters[:10]) and c[12] == r.counters[12] and c[13] == r.counters[13], (no_tpr, c[:14], r.counters[:14])
        assert c[10] <= r.counters[10] and c[11] <= r.counters[11], (no_tpr, c[:14], r.counters[:14])  # DP work: exact early exit computes fewer cells
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


# ---------------------------------------------------------------------------------------------
# BASELINE config 5's read shape (150 bp, band +-64 and wider) on spliced, two-strand,
# repeat-bearing input: the 3- and 4-cells-per-lane kernels, the global-memory trace, the
# two-cell dispatch inside a wider kernel, lift_markers over real introns and the
# multi-candidate retain / overlap path (reference src/swg.rs:116-154, src/txome.rs:110-160,
# src/aligner.rs:137-175).
CONFIG5_OPTS = dict(min_seed_len=20, min_aln_score_percent=0.574, min_aln_score=30, multimap_score_range=1, intron_mode=True)


def test_config5_shape_spliced_two_strand_repeats(syn):
    """150 bp, percent 0.574 -> ms 86 -> bw 64 (3 cells per lane) on multi-exon transcripts of both strands"""
    bases, off, _ = synth.simulate_reads(syn.t, 12000, 150, sub_rate=0.02, indel_rate=0.004, intronic_frac=0.25, stream=31)
    check_smems(syn, bases, off, 20)
    check_align(syn, bases, off, CONFIG5_OPTS)
    check_align(syn, bases, off, dict(CONFIG5_OPTS, multimap_score_range=4, min_seed_len=14))
    noisy, off2, _ = synth.simulate_reads(syn.t, 6000, 150, sub_rate=0.06, indel_rate=0.012, intronic_frac=0.25, stream=32)
    check_align(syn, noisy, off2, CONFIG5_OPTS)


@pytest.mark.parametrize("L,pct,bw", [(120, 0.0, 90), (150, 0.0, 120), (157, 0.0, 127), (150, 0.3, 105), (130, 0.5, 65)])
def test_read_level_bands_65_to_127(syn, L, pct, bw):
    """read-level parity for bands +-65..+-127 (the reference's chr21 flags -s0 give bw = L - 30)"""
    opts = dict(capi.CI_OPTS, min_aln_score_percent=pct)
    assert L - max(int(np.float32(pct) * np.float32(L)), 30) == bw
    bases, off, _ = synth.simulate_reads(syn.t, 5000, L, sub_rate=0.03, indel_rate=0.006, intronic_frac=0.25, stream=40 + L)
    check_align(syn, bases, off, opts)


def test_fuzz_bounded(syn):
    """bounded run of the randomised differential test (tools/fuzz_gpu.py): random options, ragged and
    dirty reads, seed-level and read-level parity with the oracle"""
    rng = np.random.default_rng(2)
    for r in range(8):
        L = int(rng.choice([30, 50, 75, 91, 120, 150, 200, 260]))
        k = int(rng.integers(8, 26))
        pct = float(rng.choice([0.0, 0.3, 0.5, 0.66, 0.8, 0.9]))
        opts = dict(min_seed_len=k, min_aln_score_percent=pct, min_aln_score=int(rng.choice([0, 20, 30])),
                    multimap_score_range=int(rng.integers(0, 4)), intron_mode=bool(rng.integers(0, 2)))
        n = int(rng.integers(500, 3000))
        bases, off, _ = synth.simulate_reads(syn.t, n, L, sub_rate=float(rng.choice([0.0, 0.01, 0.03, 0.08])),
                                             indel_rate=float(rng.choice([0.0, 0.002, 0.01])), intronic_frac=0.2, stream=1000 + r)
        b = bases.copy()
        for ch, p in ((ord("N"), 0.002), (ord("x"), 0.0003)):
            b[rng.random(len(b)) < p] = ch
        lower = rng.random(len(b)) < 0.2
        b[lower] = np.where((b[lower] >= 65) & (b[lower] <= 90), b[lower] + 32, b[lower])
        reads = [b[int(off[i]): int(off[i]) + (L if rng.random() < 0.7 else int(rng.integers(0, L + 1)))] for i in range(n)]
        b2, o2 = refdata.pack_reads(reads)
        check_smems(syn, b2, o2, k)
        check_align(syn, b2, o2, opts)


# ---------------------------------------------------------------------------------------------
# pool overflow -> grow -> replay (thm_batch_sync): the faulted attempt must neither write out of bounds
# nor change the result
def test_pool_overflow_replay(syn, chrm):
    bases, off, _ = synth.simulate_reads(syn.t, 6000, 91, sub_rate=0.02, indel_rate=0.004, intronic_frac=0.25, stream=77)
    ref = syn.oix.align_batch(bases, off, capi.CI_OPTS, n_threads=8)
    a = syn.aligner(capi.CI_OPTS)
    for caps in (dict(smem_cap=512), dict(ops_cap=8192), dict(cand_cap=64), dict(smem_cap=300, cand_cap=16, ops_cap=4096)):
        before = a.debug_set_pool_caps(**caps)
        g = a.align_batch(bases, off)
        assert_batch_equal(g, ref)
        assert a.debug_set_pool_caps() > before, "the small pools did not overflow: %r" % (caps,)
    a.close()
    # long reads: op bytes per read far above the pool heuristic (2 * L per exonic alignment)
    lb, lo, _ = synth.simulate_reads(chrm.t, 6000, 600, sub_rate=0.02, indel_rate=0.004, stream=78)
    opts = dict(capi.DEFAULT_OPTS, min_aln_score_percent=0.9)
    check_align(chrm, lb, lo, opts)
    # seed-pool overflow on the seed-only surface
    a = syn.aligner(capi.CI_OPTS)
    a.debug_set_pool_caps(smem_cap=256)
    g_off, g_mems = a.smems_batch(bases, off, 12)
    r = syn.oix.all_smems(bases, off, 12)
    assert np.array_equal(g_off, r.offsets) and np.array_equal(g_mems["ref_idx"], r.mems["ref_idx"])
    a.close()


# ---------------------------------------------------------------------------------------------
# reads the register-resident kernels cannot hold: the reference takes any read length and any band
# (src/swg.rs:17-38, src/aligner.rs:137-141); here they run in the any-width kernel, in the same batch
def test_mixed_batch_short_and_long_reads(syn):
    """91 bp reads with a few 300-1000 bp reads among them, the reference's chr21 flags (-k20 -s0 --intron-mode:
    band = L - 30, i.e. +-270 .. +-970 for the long ones)"""
    rng = np.random.default_rng(41)
    bases, off, _ = synth.simulate_reads(syn.t, 3000, 91, sub_rate=0.01, indel_rate=0.001, intronic_frac=0.2, stream=51)
    reads = [bases[off[i]: off[i + 1]] for i in range(3000)]
    long_len = [300, 333, 450, 512, 640, 777, 1000, 1000]
    for j, L in enumerate(long_len):
        lb, lo, _ = synth.simulate_reads(syn.t, 3, L, sub_rate=0.02, indel_rate=0.004, intronic_frac=0.5, stream=60 + j)
        for i in range(3):
            reads.insert(int(rng.integers(0, len(reads) + 1)), lb[lo[i]: lo[i + 1]])
    b2, o2 = refdata.pack_reads(reads)
    check_smems(syn, b2, o2, 20)
    g = check_align(syn, b2, o2, capi.CI_OPTS)
    lens = np.diff(o2.astype(np.int64))
    n_alns = np.diff(g.offsets.astype(np.int64))
    assert (n_alns[lens > 255] > 0).sum() >= 12  # the long reads do align
    check_align(syn, b2, o2, capi.DEFAULT_OPTS)


def test_all_long_reads_low_threshold(syn):
    """every read beyond the fast kernels' band (+-127): 200 and 260 bp with -s0 (bw 170 / 230), ragged"""
    rng = np.random.default_rng(43)
    for L in (200, 260):
        bases, off, _ = synth.simulate_reads(syn.t, 600, L, sub_rate=0.03, indel_rate=0.006, intronic_frac=0.25, stream=70 + L)
        reads = [bases[off[i]: off[i] + (L if i % 4 else int(rng.integers(0, L + 1)))] for i in range(600)]
        b2, o2 = refdata.pack_reads(reads)
        check_align(syn, b2, o2, capi.CI_OPTS)


def test_very_long_reads(chrm):
    """reads of 2 000 - 6 000 bases (seed-selection lists in global memory, band tiles, default and low thresholds)"""
    from gpu_common import mutate
    rng = np.random.default_rng(47)
    reads = []
    fwd = chrm.t["text"][: int(chrm.t["refs"][0]["len"])]
    for L in (2000, 3500, 6000):
        for flip in (False, True):  # genomic windows (chrM's transcripts are shorter), mutated, either strand
            s0 = int(rng.integers(0, len(fwd) - L))
            r = mutate(rng, fwd[s0: s0 + L], sub=0.02, indel=0.004)
            reads.append(refdata.revcomp(r) if flip else r)
    sb, so, _ = synth.simulate_reads(chrm.t, 500, 91, stream=99)
    reads += [sb[so[i]: so[i + 1]] for i in range(500)]
    b2, o2 = refdata.pack_reads(reads)
    check_smems(chrm, b2, o2, 20)
    check_align(chrm, b2, o2, capi.DEFAULT_OPTS)
    check_align(chrm, b2, o2, dict(capi.CI_OPTS, min_aln_score_percent=0.5))


def test_band_beyond_the_class_is_a_per_read_status(chrm):
    """A read whose band does not fit the launch's class fails alone, with THM_ERR_INTERNAL (the reference asserts
    band_width <= max_band_width per extend() call, src/swg.rs:32).  The classes are cut by read length and the band grows
    with it, so only the debug hook can provoke the condition: the kernels pretend to hold bands up to +-40."""
    rng = np.random.default_rng(5)
    bases, off, _ = synth.simulate_reads(chrm.t, 3000, 91, sub_rate=0.02, indel_rate=0.002, stream=31)
    reads = [bases[off[i]: off[i] + int(rng.integers(45, 92))] for i in range(3000)]
    b2, o2 = refdata.pack_reads(reads)
    lens = np.diff(o2.astype(np.int64))
    r = chrm.oix.align_batch(b2, o2, capi.CI_OPTS, n_threads=8)  # CI flags: band = L - 30
    for tpr in (False, True):
        a = chrm.aligner(capi.CI_OPTS)
        a.debug_set_flags(tpr=tpr)
        a.debug_set_band_clip(40)
        g = a.align_batch(b2, o2)
        bad = lens - 30 > 40
        assert bad.any() and (~bad).any()
        assert g.n_failed == int(bad.sum()) and g.status is not None
        assert (g.status[bad] == capi.ERR_INTERNAL).all() and (g.status[~bad] == 0).all()
        n_g = np.diff(g.offsets.astype(np.int64))
        n_r = np.diff(r.offsets.astype(np.int64))
        assert (n_g[bad] == 0).all() and np.array_equal(n_g[~bad], n_r[~bad])
        keep = np.repeat(~bad, n_r)
        for f in ("score", "ystart", "yend", "xstart", "xend", "ref_id", "strand", "aln_type", "ops_len"):
            assert np.array_equal(g.alns[f], r.alns[f][keep]), f
        a.debug_set_band_clip(None)
        assert_batch_equal(a.align_batch(b2, o2), r)
        a.close()


def test_reads_beyond_every_class_get_a_status(chrm):
    """a read longer than 65535 bases fails alone (per-read status), not the batch"""
    sb, so, _ = synth.simulate_reads(chrm.t, 200, 91, stream=101)
    reads = [sb[so[i]: so[i + 1]] for i in range(200)]
    big = np.frombuffer(b"ACGT" * 17000, np.uint8)  # 68 000 bases
    reads.insert(77, big)
    b2, o2 = refdata.pack_reads(reads)
    a = chrm.aligner(capi.CI_OPTS)
    g = a.align_batch(b2, o2)
    assert g.n_failed == 1 and g.status is not None
    assert g.status[77] == capi.ERR_UNSUPPORTED and (np.delete(g.status, 77) == 0).all()
    assert g.offsets[78] == g.offsets[77]
    keep = [r for i, r in enumerate(reads) if i != 77]
    b3, o3 = refdata.pack_reads(keep)
    ref = chrm.oix.align_batch(b3, o3, capi.CI_OPTS, n_threads=4)
    assert np.array_equal(np.delete(np.diff(g.offsets.astype(np.int64)), 77), np.diff(ref.offsets.astype(np.int64)))
    assert np.array_equal(g.alns["score"], ref.alns["score"]) and np.array_equal(g.ops, ref.ops)
    a.close()


# ---------------------------------------------------------------------------------------------
# reads with hundreds to thousands of seed hits (no cap on seed occurrences, src/index.rs:236-248,
# src/aligner.rs:143-175): a workgroup per read, speculative chunks of hits (kernels_extend.hip, TEAM)
@pytest.fixture(params=[False, True], ids=["c32", "c64"])
def heavy(request):
    def make():
        t, pos = synth.heavy_repeat_reference(length=6_000_000, copies=8000, divergence=0.01)
        t["_copy_pos"] = pos
        return t
    return _world("heavy", make, request.param)


def test_heavy_reads_thousands_of_hits(heavy):
    rng = np.random.default_rng(3)
    pos = heavy.t["_copy_pos"]
    starts = pos[rng.integers(0, len(pos), 90)] + rng.integers(0, 300 - 91, 90)
    hb, ho = synth.reads_from_positions(heavy.t, starts, 91, sub_rate=0.02, stream=12)
    lb, lo, _ = synth.simulate_reads(heavy.t, 3000, 91, sub_rate=0.01, indel_rate=0.001, intronic_frac=0.3, stream=13)
    reads = [lb[lo[i]: lo[i + 1]] for i in range(3000)]
    for i in range(90):
        reads.insert(int(rng.integers(0, len(reads) + 1)), hb[ho[i]: ho[i + 1]])
    b2, o2 = refdata.pack_reads(reads)
    a = heavy.aligner(capi.CI_OPTS)
    mo, _ = a.smems_batch(b2, o2, 20)
    hits = np.diff(mo.astype(np.int64))
    assert hits.max() >= 2000 and (hits >= 256).sum() >= 3  # the team path is taken
    a.close()
    check_align(heavy, b2, o2, capi.CI_OPTS)
    check_align(heavy, b2, o2, capi.DEFAULT_OPTS)
    check_align(heavy, b2, o2, dict(capi.CI_OPTS, multimap_score_range=6, min_seed_len=16))


@pytest.fixture(params=[False, True], ids=["c32", "c64"])
def heavy_small(request):
    return _world("heavy_small", lambda: synth.heavy_repeat_reference(length=1500000, copies=1200)[0], request.param)


def test_team_and_main_kernel_side_by_side_with_global_traces(heavy_small):
    """120-base reads, band +-41 (two cells per lane): extensions of 64 bases and more keep their trace in the wave's
    slice of global scratch, in the team kernel and in the wave-per-read kernel, which run at the same time.  (They
    indexed one scratch array by wave number: a team wave and a main wave overwrote each other's traces -- same
    scores and coordinates, a different CIGAR.  Found by tools/fuzz_gpu.py 12 12 heavy.)"""
    w = heavy_small
    opts = dict(min_seed_len=15, min_aln_score_percent=0.66, min_aln_score=20, multimap_score_range=2, intron_mode=True)
    for rep in range(3):  # a race: more than one chance to show
        bases, off, _ = synth.simulate_reads(w.t, 5000, 120, sub_rate=0.03, indel_rate=0.01, intronic_frac=0.2, stream=1001 + rep)
        check_align(w, bases, off, opts)


def test_result_views_without_copy(chrm):
    """thm_batch_view points into the aligner's pinned result sets (two, used alternately): the binding can hand out
    numpy views of them instead of copies; a view stays valid over the next fetch and is reused by the one after."""
    bases, off, _ = synth.simulate_reads(chrm.t, 3000, 91, sub_rate=0.01, indel_rate=0.001, stream=7)
    b2, o2, _ = synth.simulate_reads(chrm.t, 2000, 91, sub_rate=0.02, stream=8)
    a = chrm.aligner(capi.CI_OPTS)
    want = a.align_batch(bases, off)
    v = a.align_batch(bases, off, copy=False)
    assert not v.alns.flags.owndata and np.array_equal(v.alns, want.alns) and np.array_equal(v.ops, want.ops)
    a.align_batch(b2, o2, copy=False)  # the other result set: v is untouched
    assert np.array_equal(v.alns, want.alns) and np.array_equal(v.ops, want.ops) and np.array_equal(v.offsets, want.offsets)


def test_every_read_length_class_launches(syn):
    """Buffers (and with them the LDS of the 16-wave team workgroup, which carries 17 KB of its own) grow with the read
    length: every length up to the end of the two-cells-per-lane class must find a launchable configuration.  (A length
    whose team workgroup passed the size check without its static part failed the whole batch: found by tools/fuzz_gpu.py.)"""
    opts = dict(capi.DEFAULT_OPTS, min_aln_score_percent=0.8)
    for L in range(96, 316, 12):
        bases, off, _ = synth.simulate_reads(syn.t, 120, L, sub_rate=0.01, indel_rate=0.001, stream=300 + L)
        check_align(syn, bases, off, opts)
