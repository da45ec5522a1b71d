"""-m gpu: BASELINE configs[2]/[3] shape at full size -- 500 000 synthetic 91 bp
reads against the chr21-sized synthetic reference (46 709 983 bp): (1) exact parity
with the CPU oracle on the whole batch (alignment records and op streams
byte-identical; the oracle needs a few seconds on 16 host threads), (2)
size-independent properties (every sampled alignment is consistent with the
sequences, error-free reads align end to end at their origin, a replay is
idempotent, a sub-batch gives the same records)."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from thermite_amd import capi, refdata, synth, validate

from gpu_common import assert_batch_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    t = synth.synth_reference()
    sa = capi.build_suffix_array(t["text"])
    ix = capi.Index(t, sa=sa)
    return t, sa, ix


def test_full_size_properties_and_subset_parity(world):
    t, sa, ix = world
    n = 500000
    bases, off, truth = synth.simulate_reads(t, n, 91, sub_rate=0.01, indel_rate=0.001, stream=100)
    a = capi.Aligner(ix, capi.CI_OPTS)
    a.upload(bases, off)
    a.run()
    g = a.fetch()
    a.run()
    g2 = a.fetch()
    assert np.array_equal(g.offsets, g2.offsets) and np.array_equal(g.alns, g2.alns) and np.array_equal(g.ops, g2.ops)
    assert g.n_reads == n
    n_alns = np.diff(g.offsets.astype(np.int64))
    assert (n_alns > 0).mean() > 0.99  # reads come from the indexed transcripts
    checked, bad = validate.check_batch(t, bases, off, g, max_alns=4000)
    assert checked == 4000 and not bad, bad[:5]
    # exact parity with the oracle on all 500 000 reads
    oix = orc.Index(t, sa=sa)
    r = oix.align_batch(bases, off, capi.CI_OPTS, n_threads=16)
    assert_batch_equal(g, r)
    # a sub-batch gives the same records as the first m reads of the big batch
    m = 20000
    sub = capi.Aligner(ix, capi.CI_OPTS)
    gs = sub.align_batch(bases[: m * 91], off[: m + 1])
    assert np.array_equal(g.offsets[: m + 1], gs.offsets)
    assert np.array_equal(g.alns[: len(gs.alns)]["score"], gs.alns["score"])
    a.close()
    sub.close()


def test_error_free_reads_align_end_to_end(world):
    t, sa, ix = world
    n = 50000
    bases, off, truth = synth.simulate_reads(t, n, 91, sub_rate=0.0, indel_rate=0.0, flip_prob=0.0, stream=5)
    a = capi.Aligner(ix, capi.DEFAULT_OPTS)
    g = a.align_batch(bases, off)
    first = g.offsets[:-1].astype(np.int64)
    assert np.all(np.diff(g.offsets.astype(np.int64)) >= 1)
    top = g.alns[first]
    assert np.all(top["score"] == 91) and np.all(top["xstart"] == 0) and np.all(top["xend"] == 91)
    assert np.all(top["aln_type"] == 0) and np.all(top["primary"] == 1)
    # the transcript the read was drawn from is among the perfect exonic alignments unless an isoform ties
    assert np.all(top["tx_yend"] - top["tx_ystart"] == 91)
    a.close()


def test_config5_shape_full_size(world):
    """BASELINE config 5's read shape at full batch size: 500 000 reads of 150 bp with band +-64
    (percent 0.574 -> 3 cells per lane, wide traces in global memory) against the chr21-sized text
    (the GRCh38-sized text of config 5 is covered by the wide-coordinate tests): exact parity with
    the oracle on every read."""
    t, sa, ix = world
    n = 500000
    opts = dict(min_seed_len=20, min_aln_score_percent=0.574, min_aln_score=30, multimap_score_range=1, intron_mode=True)
    bases, off, _ = synth.simulate_reads(t, n, 150, sub_rate=0.01, indel_rate=0.001, stream=150)
    a = capi.Aligner(ix, opts)
    g = a.align_batch(bases, off)
    oix = orc.Index(t, sa=sa)
    r = oix.align_batch(bases, off, opts, n_threads=16)
    assert_batch_equal(g, r)
    checked, bad = validate.check_batch(t, bases, off, g, max_alns=2000)
    assert checked == 2000 and not bad, bad[:5]
    a.close()


def test_full_size_wide_coordinates(world):
    """the 64-bit-coordinate instantiation (what a text beyond 2^31 - 16 symbols selects) at full batch size: the same
    500 000 reads through an index forced wide give the narrow index's records, which the test above pins to the oracle"""
    t, sa, ix = world
    n = 500000
    bases, off, _ = synth.simulate_reads(t, n, 91, sub_rate=0.01, indel_rate=0.001, stream=100)
    wide = capi.Index(t, sa=sa, wide=True)
    assert wide.coord_bytes == 8 and ix.coord_bytes == 4
    a = capi.Aligner(wide, capi.CI_OPTS)
    g = a.align_batch(bases, off)
    oix = orc.Index(t, sa=sa)
    r = oix.align_batch(bases, off, capi.CI_OPTS, n_threads=16)
    assert_batch_equal(g, r)
    a.close()


def test_config2_chrM_at_size(data_dir):
    """BASELINE config 2's shape: 500 000 reads of 91 bp against the GRCh38-2020-A chrM transcriptome the reference's
    own tests ship (16 569 bp, 37 genes: every read lands in the same few windows, so seed lists and candidate pools
    are as contended as they get).  Exact parity with the oracle on every read, with the default options and with
    the reference CI's (-k20 -s0 --intron-mode)."""
    t = refdata.load_reference(data_dir + "/GRCh38-2020-A-chrM.fasta", data_dir + "/GRCh38-2020-A-chrM.gtf")
    n = 500000
    bases, off, _ = synth.simulate_reads(t, n, 91, sub_rate=0.01, indel_rate=0.001, stream=2, intronic_frac=0.25)
    ix = capi.Index(t)
    oix = orc.Index(t)
    for opts in (capi.DEFAULT_OPTS, capi.CI_OPTS):
        a = capi.Aligner(ix, opts)
        g = a.align_batch(bases, off)
        r = oix.align_batch(bases, off, opts, n_threads=16)
        assert_batch_equal(g, r)
        assert (np.diff(g.offsets.astype(np.int64)) > 0).mean() > (0.95 if opts is capi.CI_OPTS else 0.4)
        a.close()
