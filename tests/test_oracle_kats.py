"""Pins the CPU oracle against every known-answer test the reference holds for
the hot path.  Vectors: tests/golden/reference_kats.json,
transcribed from src/swg.rs:249-317, src/aligner.rs:472-639, src/txome.rs:168-341."""
import json
import os

import pytest

from oracle import pyoracle as orc


@pytest.fixture(scope="module")
def kats(golden_dir):
    with open(os.path.join(golden_dir, "reference_kats.json")) as f:
        return json.load(f)


def _ops(j):
    return [tuple(o) if isinstance(o, list) else o for o in j]


def test_swg_extend_kats(kats):
    k = kats["swg_extend"]
    swg = orc.Swg(k["max_band_width"])  # one SwgExtend reused, as in the reference test
    for c in k["cases"]:
        a = swg.extend(c["x"].encode(), c["y"].encode(), c["bw"], c["xd"])
        assert a["score"] == c["score"]
        assert (a["xstart"], a["ystart"]) == (0, 0)
        assert (a["xend"], a["yend"]) == (c["xend"], c["yend"])
        assert (a["xlen"], a["ylen"]) == (len(c["x"]), len(c["y"]))
        assert a["ops"] == _ops(c["ops"])
    assert swg.phase1_breaks == 0


def test_extend_left_right_kat(kats):
    k = kats["extend_left_right"]
    swg = orc.Swg(k["max_band_width"])
    h = k["hit"]
    a = swg.extend_left_right(k["ref"].encode(), (h["ref_idx"], h["query_idx"], h["len"]), k["read"].encode(), k["bw"], k["xd"])
    for f in ("score", "ystart", "xstart", "yend", "xend", "ylen", "xlen"):
        assert a[f] == k[f], f
    assert a["ops"] == _ops(k["ops"])


def test_filter_overlapping_kat(kats):
    k = kats["filter_overlapping"]
    inp = k["input"]
    names = sorted(set(a["ref_name"] for a in inp))
    kept = orc.filter_overlapping(
        [names.index(a["ref_name"]) for a in inp], [int(a["strand"]) for a in inp],
        [a["ystart"] for a in inp], [a["yend"] for a in inp], [a["score"] for a in inp])
    assert [inp[i] for i in kept] == k["expected"]


def test_lift_mem_to_tx_kats(kats):
    k = kats["lift_mem_to_tx"]
    for c in k["cases"]:
        assert list(orc.lift_mem_to_tx(tuple(c["mem"]), [tuple(e) for e in k["exons"]])) == c["expected"]


def test_lift_tx_to_gx_kats(kats):
    for c in kats["lift_tx_to_gx"]["cases"]:
        r = orc.lift_tx_to_gx(_ops(c["ops"]), c["ystart"], c["yend"], [tuple(e) for e in c["exons"]])
        assert r["ops"] == _ops(c["exp_ops"])
        assert (r["ystart"], r["yend"]) == (c["exp_ystart"], c["exp_yend"])


def test_swg_empty_inputs():
    # src/swg.rs:39-55
    swg = orc.Swg(4)
    a = swg.extend(b"", b"ACGT", 2, 2)
    assert a["score"] == 0 and a["ops"] == [] and (a["xend"], a["yend"]) == (0, 0)
    a = swg.extend(b"ACG", b"", 2, 2)
    assert a["score"] == 0 and a["ops"] == [("Xclip", 3)]


def test_intersect():
    # src/txome.rs:77-79
    L = orc.lib()
    assert L.orc_intersect(3, 6, 5, 9) == 1
    assert L.orc_intersect(3, 6, 6, 9) == 0
    assert L.orc_intersect(6, 9, 3, 6) == 0
    assert L.orc_intersect(4, 5, 0, 100) == 1
