"""Synthetic workloads for the BASELINE.json configurations.

The reference's benchmark inputs (pbmc10k FASTQs, chr21 FASTA/GTF) are missing
blobs (reference .MISSING_LARGE_BLOBS:1-6), so reads are simulated from the
transcripts of whatever reference is loaded, and chr21 is replaced by a
"chr21-sized synthetic" contig of the same length
(reference data/GRCh38-2020-A-chr21.fasta.fai:1 -> 46 709 983 bp).
Everything is seeded and deterministic for a given numpy version.
"""
import numpy as np

from . import refdata

SEED = 0x7468726D697465  # "thrmite"
CHR21_LEN = 46709983
_ACGT = np.frombuffer(b"ACGT", np.uint8)


def _rng(seed, stream=0):
    return np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFFFFFFFFFF, stream]))


def simulate_reads(tables, n_reads, read_len=91, sub_rate=0.01, indel_rate=0.001, flip_prob=0.5, seed=SEED, stream=0,
                   intronic_frac=0.0):
    """Reads drawn from transcripts (uniform transcript among those >= read_len,
    uniform start), strand flip, per-base substitutions, 1-base indels.
    Returns (bases uint8[n*L], offsets u64[n+1], truth dict)."""
    rng = _rng(seed, stream)
    L = read_len
    txs, tx_seq = tables["txs"], tables["tx_seq"]
    ok = np.nonzero(txs["seq_len"] >= L + 2)[0]
    if len(ok) == 0:
        raise ValueError("no transcript of length >= %d" % (L + 2))
    pad = 2  # spare bases so a deletion can be back-filled
    ti = ok[rng.integers(0, len(ok), n_reads)]
    tlen = txs["seq_len"][ti].astype(np.int64)
    start = (rng.random(n_reads) * (tlen - (L + pad) + 1)).astype(np.int64)
    base = txs["seq_off"][ti].astype(np.int64) + start
    idx = base[:, None] + np.arange(L + pad, dtype=np.int64)[None, :]
    win = tx_seq[idx]  # [n, L+pad]
    # optional unspliced (genomic / intronic) reads straight from the text
    if intronic_frac > 0:
        text, refs = tables["text"], tables["refs"]
        gi = np.nonzero(rng.random(n_reads) < intronic_frac)[0]
        if len(gi):
            r = refs[rng.integers(0, len(refs), len(gi))]
            span = (r["len"].astype(np.int64) - (L + pad)).clip(min=1)
            s = r["start_idx"].astype(np.int64) + (rng.random(len(gi)) * span).astype(np.int64)
            g = text[s[:, None] + np.arange(L + pad, dtype=np.int64)[None, :]]
            clean = ~(g == ord("N")).any(axis=1)  # windows inside N runs would hit every N position: keep the tx read
            win[gi[clean]] = g[clean]
    # 1-base indels (sparse): handled per affected read
    reads = win[:, :L].copy()
    if indel_rate > 0:
        n_ev = rng.binomial(L, indel_rate, n_reads)
        for r in np.nonzero(n_ev)[0]:
            row = list(win[r])
            for _ in range(min(int(n_ev[r]), pad)):
                p = int(rng.integers(1, L - 1))
                if rng.random() < 0.5:
                    row.insert(p, int(_ACGT[rng.integers(0, 4)]))
                else:
                    del row[p]
            reads[r] = np.array(row[:L], np.uint8)
    # substitutions: uniform over the three other bases (N stays N)
    if sub_rate > 0:
        m = rng.random(reads.shape) < sub_rate
        code = np.full(256, 255, np.uint8)
        code[_ACGT] = np.arange(4, dtype=np.uint8)
        c = code[reads[m]]
        shift = rng.integers(1, 4, c.shape[0]).astype(np.uint8)
        newb = np.where(c < 4, _ACGT[(c + shift) % 4], reads[m])
        reads[m] = newb
    flip = rng.random(n_reads) < flip_prob
    reads[flip] = refdata._COMP[reads[flip][:, ::-1]]
    bases = np.ascontiguousarray(reads.reshape(-1))
    offsets = (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L)).astype("<u8")
    return bases, offsets, dict(tx=ti, start=start, flip=flip)


def synth_annotation(rng, contig_name, contig_len, first_base, n_genes, max_tx_per_gene=4):
    """Multi-exon genes on both strands, alternative isoforms sharing exons."""
    genes, txs = [], []
    usable = contig_len - first_base - 1000
    slot = usable // max(n_genes, 1)
    for g in range(n_genes):
        n_ex = int(rng.integers(1, 12))
        ex_len = rng.integers(60, 420, n_ex)
        in_len = np.minimum(rng.integers(90, 6000, n_ex) * rng.integers(1, 4, n_ex), max(slot // (n_ex + 1), 200))
        span = int(ex_len.sum() + in_len[:-1].sum())
        if span + 200 >= slot:
            scale = (slot - 200) / float(span + 1)
            in_len = np.maximum((in_len * scale).astype(np.int64), 30)
            ex_len = np.maximum((ex_len * min(scale * 2, 1.0)).astype(np.int64), 40)
            span = int(ex_len.sum() + in_len[:-1].sum())
        lo = first_base + g * slot + int(rng.integers(0, max(slot - span - 100, 1)))
        exons, p = [], lo
        for k in range(n_ex):
            exons.append((p, p + int(ex_len[k])))
            p += int(ex_len[k]) + int(in_len[k])
        strand = bool(rng.random() < 0.5)
        gid = "SYNG%06d" % g
        genes.append(dict(id=gid, name="syn%d" % g))
        n_tx = int(rng.integers(1, max_tx_per_gene + 1))
        for t in range(n_tx):
            if t == 0 or n_ex <= 2:
                use = list(range(n_ex))
            else:  # skip some internal exons
                keep = rng.random(n_ex) < 0.7
                keep[0] = keep[-1] = True
                use = [k for k in range(n_ex) if keep[k]]
            txs.append(dict(id="SYNT%06d.%d" % (g, t), gene_idx=g, chrom=contig_name, strand=strand,
                            exons=[exons[k] for k in use]))
    return genes, txs


def synth_contig(rng, length, n_lead_n, n_families=6, copies_per_family=250, family_len=300, divergence=0.12,
                 n_segdups=40, segdup_len=1500, segdup_div=0.01):
    """i.i.d. ACGT with a leading N run, planted dispersed repeat families
    (Alu-like: many diverged copies) and a few near-identical segmental
    duplications, so that hits/read has a tail."""
    seq = _ACGT[rng.integers(0, 4, length, dtype=np.uint8)]
    seq[:n_lead_n] = ord("N")

    def mutate(a, div):
        a = a.copy()
        m = rng.random(len(a)) < div
        a[m] = _ACGT[rng.integers(0, 4, int(m.sum()), dtype=np.uint8)]
        return a

    body = length - n_lead_n - family_len - segdup_len - 10
    for _ in range(n_families):
        cons = _ACGT[rng.integers(0, 4, family_len, dtype=np.uint8)]
        for _ in range(copies_per_family):
            p = n_lead_n + int(rng.integers(0, body))
            c = mutate(cons, divergence * rng.random())
            if rng.random() < 0.5:
                c = refdata.revcomp(c)
            seq[p : p + family_len] = c
    for _ in range(n_segdups):
        a = n_lead_n + int(rng.integers(0, body))
        b = n_lead_n + int(rng.integers(0, body))
        seq[b : b + segdup_len] = mutate(seq[a : a + segdup_len], segdup_div)
    return seq


def synth_reference(length=CHR21_LEN, n_genes=None, seed=SEED, name="chr21syn", lead_n_frac=0.107):
    """chr21-sized synthetic reference + annotation -> index tables."""
    rng = _rng(seed, 1)
    n_lead = int(length * lead_n_frac) if length > 100000 else 0
    if n_genes is None:
        n_genes = max(4, int(800 * length / CHR21_LEN))
    scale = length / CHR21_LEN
    seq = synth_contig(rng, length, n_lead, copies_per_family=max(3, int(250 * scale)), n_segdups=max(2, int(40 * scale)))
    genes, txs = synth_annotation(rng, name, length, n_lead + 500, n_genes)
    return refdata.build_tables([(name, seq)], genes, txs)


def heavy_repeat_reference(length=3_000_000, copies=3000, unit_len=300, divergence=0.02, n_genes=24, seed=SEED, name="heavysyn"):
    """A reference with one Alu-like family: `copies` copies of a `unit_len`-mer, each diverged from the consensus by
    `divergence` per base (either strand).  Two copies share a given 25-mer with probability about
    (1 - divergence)^50, so a read from one copy that carries a sequencing error has SMEMs of 20-60 bases
    with hundreds to thousands of occurrences (the reference extends every one of them,
    src/index.rs:236-248, src/aligner.rs:143-145).  Returns (tables, start positions of the copies on the contig)."""
    rng = _rng(seed, 7)
    seq = _ACGT[rng.integers(0, 4, length, dtype=np.uint8)]
    cons = _ACGT[rng.integers(0, 4, unit_len, dtype=np.uint8)]
    slot = (length - 2000) // copies
    if slot < unit_len + 10:
        raise ValueError("reference too short for %d copies of %d bases" % (copies, unit_len))
    pos = np.zeros(copies, np.int64)
    for c in range(copies):
        u = cons.copy()
        m = rng.random(unit_len) < divergence
        u[m] = _ACGT[rng.integers(0, 4, int(m.sum()), dtype=np.uint8)]
        if rng.random() < 0.5:
            u = refdata.revcomp(u)
        p = 1000 + c * slot + int(rng.integers(0, slot - unit_len))
        seq[p: p + unit_len] = u
        pos[c] = p
    genes, txs = synth_annotation(rng, name, length, 500, n_genes)
    return refdata.build_tables([(name, seq)], genes, txs), pos


def reads_from_positions(tables, starts, read_len=91, sub_rate=0.01, seed=SEED, stream=0, flip_prob=0.5):
    """One read per start position on the first contig's forward strand (substitutions only, strand flip)."""
    rng = _rng(seed, stream)
    fwd = tables["text"][: int(tables["refs"][0]["len"])]
    starts = np.asarray(starts, np.int64)
    reads = fwd[starts[:, None] + np.arange(read_len, dtype=np.int64)[None, :]].copy()
    m = rng.random(reads.shape) < sub_rate
    code = np.full(256, 255, np.uint8)
    code[_ACGT] = np.arange(4, dtype=np.uint8)
    c = code[reads[m]]
    reads[m] = np.where(c < 4, _ACGT[(c + rng.integers(1, 4, c.shape[0]).astype(np.uint8)) % 4], reads[m])
    flip = rng.random(len(starts)) < flip_prob
    reads[flip] = refdata._COMP[reads[flip][:, ::-1]]
    n = len(starts)
    return np.ascontiguousarray(reads.reshape(-1)), (np.arange(n + 1, dtype=np.uint64) * np.uint64(read_len)).astype("<u8")
