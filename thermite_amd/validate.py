"""Size-independent checks of an aligned batch: every reported alignment must be
consistent with the sequences it claims to align (used at full benchmark size,
where the CPU oracle would take too long to replay everything)."""
import numpy as np

from .refdata import _COMP, _UPPER

OPN = ["Match", "Subst", "Del", "Ins", "Xclip", "Yclip"]


def decode_ops(b):
    b = bytes(b)
    out, i = [], 0
    while i < len(b):
        k = b[i]
        i += 1
        if k >= 4:
            out.append((k, int.from_bytes(b[i : i + 4], "little")))
            i += 4
        else:
            out.append((k, 1))
    return out


def check_alignment(tables, read, aln, ops_bytes, tx_ops_bytes=None):
    """Returns None if consistent, else a string describing the violation.
    Walks the op stream over (read or its revcomp) x forward chromosome; introns
    (Yclip) skip reference bases; Match/Subst labels must agree with the bases.
    The reported score is the DP maximum; the op list comes from the reference's
    single-matrix traceback (one direction per cell, src/swg.rs:170-207), which
    can leave the optimal path after a gap, so the path's own affine-gap score
    is only a lower bound of the reported score."""
    refs = tables["refs"]
    ref = refs[int(aln["ref_id"])]
    fwd = refs[int(aln["ref_id"]) & ~1]  # forward copy of the same contig
    text = tables["text"]
    L = len(read)
    x = _UPPER[read]
    xs, xe = int(aln["xstart"]), int(aln["xend"])
    if not ref["strand"]:
        x = _COMP[x[::-1]]
        xs, xe = L - xe, L - xs
    y0 = int(fwd["start_idx"])
    i, j = 0, int(aln["ystart"])
    ops = decode_ops(ops_bytes)
    if int(aln["xlen"]) != L or int(aln["ylen"]) != int(fwd["len"]):
        return "xlen/ylen"
    k = 0
    if ops and ops[0][0] == 4:
        i = ops[0][1]
        k = 1
    if i != xs:
        return "leading clip %d != xstart %d" % (i, xs)
    nm = ns = gaps = gap_open = del_runs = 0
    prev = -1
    while k < len(ops):
        kind, n = ops[k]
        if kind == 4:
            if k != len(ops) - 1 or n != L - i:
                return "trailing clip"
            break
        if kind == 5:
            j += n
        elif kind == 0 or kind == 1:
            same = x[i] == text[y0 + j]
            if same != (kind == 0):
                return "op %d labelled %s but bases %s" % (k, OPN[kind], "equal" if same else "differ")
            nm += kind == 0
            ns += kind == 1
            i += 1
            j += 1
        elif kind == 2:
            gaps += 1
            if prev != 2:
                gap_open += 1
                del_runs += 1
            j += 1
        elif kind == 3:
            gaps += 1
            if prev != 3:
                gap_open += 1
            i += 1
        prev = kind
        k += 1
    if i != xe:
        return "query end %d != xend %d" % (i, xe)
    if j != int(aln["yend"]):
        return "reference end %d != yend %d" % (j, int(aln["yend"]))
    lo = nm - ns - gaps - gap_open
    if int(aln["score"]) < lo or int(aln["score"]) > xe - xs:
        return "score %d outside [%d, %d]" % (int(aln["score"]), lo, xe - xs)
    return None


def check_batch(tables, bases, offsets, result, max_alns=None, rng=None):
    """Check (a sample of) the alignments of a batch; returns (n_checked, violations)."""
    n = len(result.alns)
    idx = np.arange(n)
    if max_alns is not None and n > max_alns:
        idx = (rng or np.random.default_rng(0)).choice(n, max_alns, replace=False)
    read_of = np.searchsorted(result.offsets, idx, side="right") - 1
    bad = []
    for a, r in zip(idx, read_of):
        aln = result.alns[a]
        read = bases[int(offsets[r]) : int(offsets[r + 1])]
        msg = check_alignment(tables, read, aln, result.ops[int(aln["ops_off"]) : int(aln["ops_off"]) + int(aln["ops_len"])])
        if msg:
            bad.append((int(a), int(r), msg))
    # structural invariants of align_read (src/aligner.rs:177-187)
    offs = result.offsets.astype(np.int64)
    first = offs[:-1][np.diff(offs) > 0]
    if len(first) and not np.all(result.alns["primary"][first] == 1):
        bad.append((-1, -1, "first alignment of a read is not primary"))
    if int(result.alns["primary"].sum()) != len(first):
        bad.append((-1, -1, "more than one primary per read"))
    return len(idx), bad
