#!/usr/bin/env python3
"""Regenerates the *restatement* goldens: outputs of the CPU oracle (not of the Rust
reference, which cannot be built here) for the reference's own test inputs.  They
pin the oracle and the HIP path against regressions; parity status per field is
described in oracle/README.md.

    python tests/golden/make_goldens.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import aln_writer as ow  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
from thermite_amd import capi, refdata, synth  # noqa: E402

DATA = os.path.join(HERE, "data")


def dump(tables, names, bases, off, opts):
    ix = orc.Index(tables)
    r = ix.align_batch(bases, off, opts)
    out = []
    for i, nm in enumerate(names):
        alns = []
        for a in r.alns[r.offsets[i]: r.offsets[i + 1]]:
            d = dict(ref=tables["names"][tables["refs"][a["ref_id"]]["name_id"]], strand="+" if a["strand"] else "-",
                     score=int(a["score"]), ystart=int(a["ystart"]), yend=int(a["yend"]), xstart=int(a["xstart"]),
                     xend=int(a["xend"]), type="ENI"[a["aln_type"]], primary=int(a["primary"]),
                     ops=orc.decode_ops(r.ops[a["ops_off"]: a["ops_off"] + a["ops_len"]]))
            if a["aln_type"] == 0:
                d["tx"] = tables["tx_ids"][a["tx_or_gene_idx"]]
                d["tx_ystart"] = int(a["tx_ystart"])
                d["tx_ops"] = orc.decode_ops(r.ops[a["tx_ops_off"]: a["tx_ops_off"] + a["tx_ops_len"]])
            elif a["aln_type"] == 1:
                d["gene"] = tables["gene_ids"][a["tx_or_gene_idx"]]
            alns.append(d)
        out.append(dict(read=nm, seq=bytes(bases[off[i]: off[i + 1]]).decode(), alignments=alns))
    return out


def main():
    t = refdata.load_reference(os.path.join(DATA, "test_ref.fasta"), os.path.join(DATA, "test_ref.gtf"))
    names, seqs, _ = refdata.parse_fastq(os.path.join(DATA, "test_query.fastq"))
    bases, off = refdata.pack_reads(seqs)
    opts = dict(capi.DEFAULT_OPTS, min_seed_len=3, min_aln_score=0)  # data/Makefile:21: -k3 --min-aln-score=0
    g = dict(source="CPU oracle (restatement), reference data/test_query.fastq vs data/test_ref.{fasta,gtf}, -k3 --min-aln-score=0",
             opts=opts, reads=dump(t, names, bases, off, opts))
    json.dump(g, open(os.path.join(HERE, "test_query_alignments.json"), "w"), indent=1)
    # the same run rendered by the restated writer (src/aln_writer.rs): what `thermite align -a` / PAF would print
    names, seqs, quals = refdata.parse_fastq(os.path.join(DATA, "test_query.fastq"))
    r = orc.Index(t).align_batch(bases, off, opts)
    nb, sb, qb = [n.encode() for n in names], [bytes(x) for x in seqs], [bytes(q) for q in quals]
    open(os.path.join(HERE, "test_query.sam"), "wb").write(ow.sam_header(t) + ow.format_batch(t, nb, sb, qb, r, "sam"))
    open(os.path.join(HERE, "test_query.paf"), "wb").write(ow.format_batch(t, nb, sb, qb, r, "paf"))
    tm = refdata.load_reference(os.path.join(DATA, "GRCh38-2020-A-chrM.fasta"), os.path.join(DATA, "GRCh38-2020-A-chrM.gtf"))
    b, o, _ = synth.simulate_reads(tm, 200, 91, sub_rate=0.02, indel_rate=0.005, stream=42)
    g = dict(source="CPU oracle (restatement), 200 synthetic 91 bp reads (thermite_amd.synth, stream 42) vs chrM, -k20 -s0 --intron-mode",
             opts=capi.CI_OPTS, reads=dump(tm, ["r%d" % i for i in range(200)], b, o, capi.CI_OPTS))
    json.dump(g, open(os.path.join(HERE, "chrM_200_alignments.json"), "w"), separators=(",", ":"))
    print("wrote goldens")


if __name__ == "__main__":
    main()
