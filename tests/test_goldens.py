"""Committed restatement goldens (tests/golden/*.json, made by tests/golden/make_goldens.py):
the oracle (CPU) and the HIP path (-m gpu) must both reproduce them."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle as orc
from thermite_amd import capi, refdata


def _load(golden_dir, data_dir, which):
    g = json.load(open(os.path.join(golden_dir, which)))
    if which.startswith("test_query"):
        t = refdata.load_reference(data_dir + "/test_ref.fasta", data_dir + "/test_ref.gtf")
    else:
        t = refdata.load_reference(data_dir + "/GRCh38-2020-A-chrM.fasta", data_dir + "/GRCh38-2020-A-chrM.gtf")
    bases, off = refdata.pack_reads([r["seq"].encode() for r in g["reads"]])
    return g, t, bases, off


def _as_golden(t, res, i):
    out = []
    for a in res.alns[res.offsets[i]: res.offsets[i + 1]]:
        d = dict(ref=t["names"][t["refs"][a["ref_id"]]["name_id"]], strand="+" if a["strand"] else "-",
                 score=int(a["score"]), ystart=int(a["ystart"]), yend=int(a["yend"]), xstart=int(a["xstart"]),
                 xend=int(a["xend"]), type="ENI"[a["aln_type"]], primary=int(a["primary"]),
                 ops=json.loads(json.dumps(orc.decode_ops(res.ops[a["ops_off"]: a["ops_off"] + a["ops_len"]]))))
        if a["aln_type"] == 0:
            d["tx"] = t["tx_ids"][a["tx_or_gene_idx"]]
            d["tx_ystart"] = int(a["tx_ystart"])
            d["tx_ops"] = json.loads(json.dumps(orc.decode_ops(res.ops[a["tx_ops_off"]: a["tx_ops_off"] + a["tx_ops_len"]])))
        elif a["aln_type"] == 1:
            d["gene"] = t["gene_ids"][a["tx_or_gene_idx"]]
        out.append(d)
    return out


@pytest.mark.parametrize("which", ["test_query_alignments.json", "chrM_200_alignments.json"])
def test_oracle_reproduces_goldens(golden_dir, data_dir, which):
    g, t, bases, off = _load(golden_dir, data_dir, which)
    r = orc.Index(t).align_batch(bases, off, g["opts"])
    for i, rd in enumerate(g["reads"]):
        assert _as_golden(t, r, i) == rd["alignments"], rd["read"]


def test_golden_read_names_document_intent(golden_dir, data_dir):
    """data/test_query.fastq names say what should happen (reference data/test_query.fastq:1-40)."""
    g, _, _, _ = _load(golden_dir, data_dir, "test_query_alignments.json")
    by = {r["read"]: r["alignments"] for r in g["reads"]}
    assert by["unmapped"] == []
    assert by["revcomp"][0]["strand"] == "-" and by["revcomp"][0]["score"] == 4
    assert ["Yclip", 4] in by["spliced_tx1"][0]["ops"] and by["spliced_tx1"][0]["tx"] == "introns_seq_tx1"
    assert ["Yclip", 12] in by["spliced_tx2"][0]["ops"] and by["spliced_tx2"][0]["tx"] == "introns_seq_tx2"
    assert "Subst" in by["spliced_with_err1"][0]["ops"]
    assert by["spliced_revcomp"][0]["ref"] == "introns_revcomp" and by["spliced_revcomp"][0]["strand"] == "-"


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["test_query_alignments.json", "chrM_200_alignments.json"])
def test_hip_path_reproduces_goldens(golden_dir, data_dir, which):
    g, t, bases, off = _load(golden_dir, data_dir, which)
    ix = capi.Index(t)
    a = capi.Aligner(ix, g["opts"])
    r = a.align_batch(bases, off)
    for i, rd in enumerate(g["reads"]):
        assert _as_golden(t, r, i) == rd["alignments"], rd["read"]
    a.close()
    ix.close()
