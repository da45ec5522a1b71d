"""Reference ingestion: FASTA + GTF -> the raw index tables the C ABI takes.

Host-side, offline (not on the per-read path).  Restates the *content* that
Index::create_from_files builds (reference src/index.rs:52-223):

  * text T = for each contig: UPPER(seq) '$' UPPER(revcomp(seq)) '$'  (src/index.rs:67-101)
  * two Ref records per contig, forward then reverse            (src/index.rs:78-100)
  * exons / transcripts lifted into concatenated coordinates, reverse-strand
    features mapped into the revcomp copy (end_idx - 1 - coord) and their exon
    order reversed                                              (src/index.rs:149-195)
  * per-gene span = (min tx_start, max tx_end)                  (src/index.rs:134,159-162)

GTF semantics (gene / transcript / exon rows, 1-based inclusive -> 0-based
half-open, gene_name falling back to gene_id, exons sorted by start) follow the
`transcriptome` crate of 10XDev/cellranger (Cargo.lock:1272-1274), which is not
present in the reference checkout: restated from its published behaviour,
parity unpinned.

The tables are plain numpy arrays with the PODs of include/thermite.h.
"""
import gzip
import re

import numpy as np

REF_DT = np.dtype(
    [("start_idx", "<u8"), ("end_idx", "<u8"), ("len", "<u8"), ("name_id", "<u4"), ("strand", "u1"), ("pad_", "u1", 3)]
)
EXON_DT = np.dtype([("start", "<u8"), ("end", "<u8"), ("tx_idx", "<u4"), ("pad_", "<u4")])
TX_DT = np.dtype(
    [
        ("exon_begin", "<u8"),
        ("seq_off", "<u8"),
        ("seq_len", "<u8"),
        ("n_exons", "<u4"),
        ("gene_idx", "<u4"),
        ("strand", "u1"),
        ("pad_", "u1", 7),
    ]
)
SPAN_DT = np.dtype([("start", "<u8"), ("end", "<u8")])

_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b
_UPPER = np.arange(256, dtype=np.uint8)
_UPPER[ord("a") : ord("z") + 1] -= 32
_VALID = np.zeros(256, dtype=bool)
_VALID[list(b"ACGTN")] = True


def revcomp(a):
    return _COMP[a[::-1]]


def _open(path):
    return gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")


def parse_fasta(path):
    """-> list of (name, uint8 array); name = first word of the header (src/index.rs:69)."""
    contigs, name, chunks = [], None, []
    with _open(path) as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    contigs.append((name, np.frombuffer(b"".join(chunks), np.uint8).copy()))
                name = line[1:].decode().split(" ")[0]
                chunks = []
            elif line:
                chunks.append(line)
    if name is not None:
        contigs.append((name, np.frombuffer(b"".join(chunks), np.uint8).copy()))
    return contigs


def parse_fastq(path):
    """-> (names, list of uint8 arrays, quals)."""
    names, seqs, quals = [], [], []
    with _open(path) as f:
        lines = [ln.rstrip(b"\r\n") for ln in f]
    i = 0
    while i + 3 < len(lines) + 1 and i < len(lines):
        if not lines[i]:
            i += 1
            continue
        names.append(lines[i][1:].decode())
        seqs.append(np.frombuffer(lines[i + 1], np.uint8).copy())
        quals.append(lines[i + 3])
        i += 4
    return names, seqs, quals


_ATTR = re.compile(r'(\S+) "([^"]*)"')


def parse_gtf(path):
    """-> (genes, transcripts) with genes = [dict(id, name)], transcripts =
    [dict(id, gene_idx, chrom, strand(bool, True = '+'), exons=[(start, end)] 0-based half-open, sorted)]."""
    genes, gene_idx = [], {}
    txs, tx_idx = [], {}
    with _open(path) as f:
        for raw in f:
            if raw.startswith(b"#") or not raw.strip():
                continue
            c = raw.decode().rstrip("\n").split("\t")
            if len(c) < 9:
                continue
            chrom, _, feat, start, end, _, strand, _, attr = c[:9]
            if feat not in ("gene", "transcript", "exon"):
                continue
            a = dict(_ATTR.findall(attr))
            if feat == "gene":
                gid = a["gene_id"]
                if gid not in gene_idx:
                    gene_idx[gid] = len(genes)
                    genes.append(dict(id=gid, name=a.get("gene_name", gid)))
            elif feat == "transcript":
                gid = a["gene_id"]
                if gid not in gene_idx:
                    gene_idx[gid] = len(genes)
                    genes.append(dict(id=gid, name=a.get("gene_name", gid)))
                tid = a["transcript_id"]
                tx_idx[tid] = len(txs)
                txs.append(dict(id=tid, gene_idx=gene_idx[gid], chrom=chrom, strand=(strand == "+"), exons=[]))
            else:
                tid = a["transcript_id"]
                txs[tx_idx[tid]]["exons"].append((int(start) - 1, int(end)))
    for t in txs:
        t["exons"].sort()
    return genes, txs


def build_tables(contigs, genes, transcripts):
    """Assemble the raw index tables (see module docstring)."""
    n_total = sum(2 * (len(s) + 1) for _, s in contigs)
    text = np.empty(n_total, np.uint8)
    refs = np.zeros(2 * len(contigs), REF_DT)
    names = []
    ref_of = {}
    pos = 0
    for ci, (name, seq) in enumerate(contigs):
        s = _UPPER[np.ascontiguousarray(seq, np.uint8)]
        if not _VALID[s].all():
            bad = sorted(set(bytes(s[~_VALID[s]])))
            raise ValueError("contig %s has bases outside ACGTN: %r" % (name, bytes(bad)))
        names.append(name)
        L = len(s)
        for strand in (1, 0):
            r = refs[2 * ci + (0 if strand else 1)]
            r["start_idx"] = pos
            text[pos : pos + L] = s if strand else revcomp(s)
            pos += L
            text[pos] = ord("$")
            pos += 1
            r["end_idx"] = pos
            r["len"] = L
            r["name_id"] = ci
            r["strand"] = strand
            ref_of[(name, bool(strand))] = 2 * ci + (0 if strand else 1)
    assert pos == n_total
    # rank of each contig name in byte order (filter_overlapping sorts by ref_name, src/aligner.rs:322-327)
    uniq = sorted(set(n.encode() for n in names))
    name_rank = np.array([uniq.index(n.encode()) for n in names], "<u4")

    n_exons = sum(len(t["exons"]) for t in transcripts)
    exons = np.zeros(n_exons, EXON_DT)
    txs = np.zeros(len(transcripts), TX_DT)
    gene_lo = np.full(len(genes), n_total, np.uint64)
    gene_hi = np.zeros(len(genes), np.uint64)
    seq_chunks, seq_off, eb = [], 0, 0
    for ti, t in enumerate(transcripts):
        strand = bool(t["strand"])
        r = refs[ref_of[(t["chrom"], strand)]]
        s0, e1 = int(r["start_idx"]), int(r["end_idx"])
        ex = t["exons"]
        if not ex:
            raise ValueError("transcript %s has no exons" % t["id"])
        tstart, tend = ex[0][0], ex[-1][1]
        if strand:
            tx_start, tx_end = tstart + s0, tend + s0
            lifted = [(a + s0, b + s0) for a, b in ex]
        else:
            tx_start, tx_end = e1 - 1 - tend, e1 - 1 - tstart
            lifted = [(e1 - 1 - b, e1 - 1 - a) for a, b in ex][::-1]
        g = t["gene_idx"]
        gene_lo[g] = min(int(gene_lo[g]), tx_start)
        gene_hi[g] = max(int(gene_hi[g]), tx_end)
        tl = 0
        for k, (a, b) in enumerate(lifted):
            e = exons[eb + k]
            e["start"], e["end"], e["tx_idx"] = a, b, ti
            seq_chunks.append(text[a:b])
            tl += b - a
        x = txs[ti]
        x["exon_begin"], x["n_exons"], x["gene_idx"], x["strand"] = eb, len(lifted), g, int(strand)
        x["seq_off"], x["seq_len"] = seq_off, tl
        seq_off += tl
        eb += len(lifted)
    tx_seq = np.concatenate(seq_chunks) if seq_chunks else np.zeros(0, np.uint8)
    gspans = np.zeros(len(genes), SPAN_DT)
    gspans["start"], gspans["end"] = gene_lo, gene_hi
    return dict(
        text=text,
        refs=refs,
        names=names,
        name_rank=name_rank,
        txs=txs,
        exons=exons,
        tx_seq=np.ascontiguousarray(tx_seq),
        genes=gspans,
        gene_ids=[g["id"] for g in genes],
        gene_names=[g["name"] for g in genes],
        tx_ids=[t["id"] for t in transcripts],
    )


def load_reference(fasta_path, gtf_path):
    contigs = parse_fasta(fasta_path)
    genes, txs = parse_gtf(gtf_path)
    return build_tables(contigs, genes, txs)


def pack_reads(seqs):
    """list of uint8 arrays / bytes -> (bases, offsets[n+1])"""
    arrs = [np.frombuffer(bytes(s), np.uint8) if not isinstance(s, np.ndarray) else s for s in seqs]
    off = np.zeros(len(arrs) + 1, "<u8")
    if arrs:
        off[1:] = np.cumsum([len(a) for a in arrs])
    bases = np.concatenate(arrs) if arrs and off[-1] > 0 else np.zeros(0, np.uint8)
    return np.ascontiguousarray(bases, np.uint8), off
