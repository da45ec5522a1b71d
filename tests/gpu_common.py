"""Shared helpers for the -m gpu parity tests (HIP path vs the CPU oracle)."""
import numpy as np

from oracle import pyoracle as orc
from thermite_amd import capi, refdata

_ACGT = np.frombuffer(b"ACGT", np.uint8)


def mutate(rng, a, sub=0.05, indel=0.02):
    out = []
    for ch in a:
        r = rng.random()
        if r < indel / 2:
            continue
        if r < indel:
            out.append(int(_ACGT[rng.integers(0, 4)]))
        if rng.random() < sub:
            out.append(int(_ACGT[rng.integers(0, 4)]))
        else:
            out.append(int(ch))
    return np.array(out, np.uint8)


def swg_fuzz_problems(rng, n, max_len, bw_lo, bw_hi, related=0.8):
    xs, ys, bws, xds = [], [], [], []
    for _ in range(n):
        xl = int(rng.integers(0, max_len + 1))
        x = _ACGT[rng.integers(0, 4, xl)]
        bw = int(rng.integers(bw_lo, bw_hi + 1))
        if rng.random() < related:
            y = mutate(rng, x, sub=rng.random() * 0.15, indel=rng.random() * 0.06)
            tail = _ACGT[rng.integers(0, 4, int(rng.integers(0, 40)))]
            y = np.concatenate([y, tail])
            if rng.random() < 0.2:
                y = y[: int(rng.integers(0, len(y) + 1))]
        else:
            y = _ACGT[rng.integers(0, 4, int(rng.integers(0, max_len + 40)))]
        xs.append(x)
        ys.append(y.astype(np.uint8))
        bws.append(bw)
        xds.append(bw + int(rng.integers(0, 4)) * int(rng.integers(0, 8)))
    xb, xo = refdata.pack_reads(xs)
    yb, yo = refdata.pack_reads(ys)
    return xb, xo, yb, yo, np.array(bws, "<u4"), np.array(xds, "<i4")


def assert_swg_equal(gpu_alns, gpu_ops, ref):
    assert len(gpu_alns) == len(ref.swg)
    for f in ("score", "xend", "yend", "ops_len", "ops_off"):
        bad = np.nonzero(gpu_alns[f] != ref.swg[f])[0]
        assert len(bad) == 0, "field %s differs at problems %s" % (f, bad[:10])
    assert np.array_equal(gpu_ops, ref.ops)


def assert_batch_equal(gpu, ref, max_report=5):
    """gpu: capi.BatchResult, ref: oracle Result('aln') -- canonical layouts must be byte-identical."""
    assert gpu.n_reads == ref.n
    if not np.array_equal(gpu.offsets, ref.offsets):
        d = np.nonzero(np.diff(gpu.offsets.astype(np.int64)) != np.diff(ref.offsets.astype(np.int64)))[0]
        raise AssertionError("alignment counts differ for reads %s" % d[:max_report])
    for f in capi.ALN_DT.names:
        if f == "pad_":
            continue
        bad = np.nonzero(gpu.alns[f] != ref.alns[f])[0]
        if len(bad):
            i = int(bad[0])
            read = int(np.searchsorted(ref.offsets, i, side="right") - 1)
            raise AssertionError("field %s differs at alignment %d (read %d): gpu=%s ref=%s" %
                                 (f, i, read, gpu.alns[i], ref.alns[i]))
    assert np.array_equal(gpu.ops, ref.ops), "op streams differ"
