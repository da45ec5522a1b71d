"""-m gpu: wave primitives and the operator-level SWG kernel against the oracle
(reference semantics: src/swg.rs:31-240, KATs src/swg.rs:249-317)."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as orc
from thermite_amd import capi, refdata

from gpu_common import assert_swg_equal, swg_fuzz_problems

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aligner(data_dir):
    t = refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")
    ix = capi.Index(t)
    a = capi.Aligner(ix, dict(capi.DEFAULT_OPTS, min_seed_len=3, min_aln_score=0))
    yield a
    a.close()
    ix.close()


def test_wave_primitives(aligner):
    rng = np.random.default_rng(1)
    for _ in range(5):
        v = rng.integers(-100000, 100000, 64).astype("<i4")
        o = aligner.debug_wave_prims(v)
        assert np.array_equal(o[0], np.maximum.accumulate(v))
        neg = -858993459
        ex = np.concatenate([[neg], np.maximum.accumulate(v)[:-1]])
        assert np.array_equal(o[1], ex)
        assert np.all(o[2] == v.max()) and np.all(o[3] == v.min())
        assert np.array_equal(o[4], np.concatenate([[-7], v[:-1]]))
        assert np.array_equal(o[5], np.concatenate([v[1:], [-9]]))


def test_reference_kats(aligner, golden_dir):
    k = json.load(open(os.path.join(golden_dir, "reference_kats.json")))["swg_extend"]
    xs = [c["x"].encode() for c in k["cases"]]
    ys = [c["y"].encode() for c in k["cases"]]
    xb, xo = refdata.pack_reads(xs)
    yb, yo = refdata.pack_reads(ys)
    bw = [c["bw"] for c in k["cases"]]
    xd = [c["xd"] for c in k["cases"]]
    alns, ops = aligner.swg_extend_batch(xb, xo, yb, yo, bw, xd, k["max_band_width"])
    for i, c in enumerate(k["cases"]):
        a = alns[i]
        assert (a["score"], a["xend"], a["yend"]) == (c["score"], c["xend"], c["yend"])
        got = orc.decode_ops(ops[a["ops_off"]: a["ops_off"] + a["ops_len"]])
        assert got == [tuple(o) if isinstance(o, list) else o for o in c["ops"]]


def test_out_of_contract(aligner):
    xb, xo = refdata.pack_reads([b"ACGT"])
    with pytest.raises(capi.ThermiteError) as e:
        aligner.swg_extend_batch(xb, xo, xb, xo, [5], [5], 4)  # bw > max: assert!, src/swg.rs:32
    assert e.value.code == capi.ERR_OUT_OF_CONTRACT
    with pytest.raises(capi.ThermiteError) as e:
        aligner.swg_extend_batch(xb, xo, xb, xo, [3], [2], 4)  # xd < bw: SURVEY A.5
    assert e.value.code == capi.ERR_OUT_OF_CONTRACT


@pytest.mark.parametrize("bw_lo,bw_hi,max_len,n", [(0, 31, 100, 6000), (0, 8, 30, 3000), (32, 63, 150, 1500),
                                                   (64, 95, 150, 600), (96, 127, 160, 400), (31, 31, 91, 4000)])
def test_fuzz_vs_oracle(aligner, bw_lo, bw_hi, max_len, n):
    rng = np.random.default_rng(bw_lo * 1000 + bw_hi)
    xb, xo, yb, yo, bw, xd = swg_fuzz_problems(rng, n, max_len, bw_lo, bw_hi)
    aligner.reset_counters()
    alns, ops = aligner.swg_extend_batch(xb, xo, yb, yo, bw, xd, bw_hi)
    ref = orc.swg_extend_batch(xb, xo, yb, yo, bw, xd, bw_hi)
    assert_swg_equal(alns, ops, ref)
    c = aligner.counters()
    # calls match; cells/columns are <= the reference's because of the exact early exit (swg_device.h)
    assert c[9] == ref.counters[9] and c[10] <= ref.counters[10] and c[11] <= ref.counters[11]


@pytest.mark.parametrize("bw_lo,bw_hi,max_len,n", [(128, 200, 260, 300), (0, 700, 700, 200), (300, 1000, 1200, 60),
                                                   (63, 64, 120, 400), (127, 129, 300, 300)])
def test_fuzz_any_band_width(aligner, bw_lo, bw_hi, max_len, n):
    """bands beyond +-127 (SwgExtend::new takes any band, src/swg.rs:17-26): the tiled any-width kernel; a batch whose
    widest band exceeds +-127 runs all of its problems there, so narrow bands are covered by it as well"""
    rng = np.random.default_rng(bw_lo * 977 + bw_hi)
    xb, xo, yb, yo, bw, xd = swg_fuzz_problems(rng, n, max_len, bw_lo, bw_hi)
    if bw_lo == 0:
        bw[: n // 2] = rng.integers(0, 40, n // 2)  # many narrow problems inside the wide batch
        xd = np.maximum(xd, bw.astype("<i4"))
        bw[-1] = bw_hi
        xd[-1] = bw_hi
    alns, ops = aligner.swg_extend_batch(xb, xo, yb, yo, bw, xd, bw_hi)
    ref = orc.swg_extend_batch(xb, xo, yb, yo, bw, xd, bw_hi)
    assert_swg_equal(alns, ops, ref)


def test_one_mismatch_then_exact_shortcut(aligner):
    """extensions of the shape 'the seed ended on a substitution, the rest matches' take a shortcut without DP
    (swg_device.h: swg_one_mismatch_shortcut); adversarial inputs around its conditions: short x, one repeated
    base, two-letter repeats, y ending exactly at |x|, y continuing the repeat, narrow bands, x_drop 0 / 1"""
    rng = np.random.default_rng(99)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    xs, ys, bws, xds = [], [], [], []
    for _ in range(6000):
        xl = int(rng.integers(1, 70))
        kind = rng.integers(0, 5)
        if kind == 0:
            x = acgt[rng.integers(0, 4, xl)]
        elif kind == 1:
            x = np.full(xl, acgt[rng.integers(0, 4)], np.uint8)  # one repeated base
        elif kind == 2:
            x = np.full(xl, acgt[rng.integers(0, 4)], np.uint8)  # one repeated base except one position
            x[int(rng.integers(0, xl))] = acgt[rng.integers(0, 4)]
        elif kind == 3:
            x = np.tile(acgt[rng.integers(0, 4, 2)], xl)[:xl]  # two-letter repeat
        else:
            x = np.tile(acgt[rng.integers(0, 4, int(rng.integers(1, 5)))], xl)[:xl]
        y = x.copy()
        if rng.random() < 0.9:
            y[0] = acgt[(int(np.searchsorted(acgt, y[0])) + int(rng.integers(1, 4))) % 4]  # substitution at the seed's end
        if rng.random() < 0.15 and xl > 3:
            y[int(rng.integers(1, xl))] = acgt[rng.integers(0, 4)]  # sometimes a second difference: no shortcut
        t = rng.integers(0, 4)
        if t == 1:
            y = np.concatenate([y, [x[-1]]])  # continues the last base
        elif t == 2:
            y = np.concatenate([y, np.full(int(rng.integers(1, 6)), x[0], np.uint8)])
        elif t == 3:
            y = np.concatenate([y, acgt[rng.integers(0, 4, int(rng.integers(1, 30)))]])
        if rng.random() < 0.05:
            y = y[: int(rng.integers(0, len(y) + 1))]
        bw = int(rng.choice([0, 1, 2, 3, 5, 31, 61]))
        xs.append(x)
        ys.append(y.astype(np.uint8))
        bws.append(bw)
        xds.append(bw + int(rng.choice([0, 0, 1, 7])))
    xb, xo = refdata.pack_reads(xs)
    yb, yo = refdata.pack_reads(ys)
    alns, ops = aligner.swg_extend_batch(xb, xo, yb, yo, bws, xds, 61)
    ref = orc.swg_extend_batch(xb, xo, yb, yo, np.array(bws, "<u4"), np.array(xds, "<i4"), 61)
    assert_swg_equal(alns, ops, ref)


def test_empty_inputs(aligner):
    xs = [b"", b"ACG", b"", b"A"]
    ys = [b"ACGT", b"", b"", b"A"]
    xb, xo = refdata.pack_reads(xs)
    yb, yo = refdata.pack_reads(ys)
    alns, ops = aligner.swg_extend_batch(xb, xo, yb, yo, [2, 2, 2, 0], [2, 2, 2, 0], 4)
    ref = orc.swg_extend_batch(xb, xo, yb, yo, [2, 2, 2, 0], [2, 2, 2, 0], 4)
    assert_swg_equal(alns, ops, ref)
