"""oracle/aln_writer.py -- CPU restatement of the reference's SAM / PAF writer.

TEST INFRASTRUCTURE (see oracle/README.md): only tests/ may import this.  It
restates reference src/aln_writer.rs function by function, and the writer loop
of src/aligner.rs:51-116, in plain Python; the product's writer is
thermite_amd/csrc/io_writer.cpp.

Parity: the reference has no tests for aln_writer.rs and the text layout of a
SAM line comes from noodles-sam 0.1.0 (Cargo.lock:773-776), whose source is not
in the reference checkout -> restated from the SAM specification: **parity
unpinned** for the byte-level layout; the field *values* (flags, MAPQ table,
CIGAR run-length rules, tag set and order, PAF columns) follow aln_writer.rs
line by line.
"""
import numpy as np

from . import pyoracle as orc


# bio::alphabets::dna::complement (bio 0.37.1, recalled): IUPAC table, both cases, others unchanged
_COMP = list(range(256))
for _a, _b in zip(b"AGCTYRWSKMDVHBN", b"TCGARYWSMKHBDVN"):
    _COMP[_a] = _b
    _COMP[_a + 32] = _b + 32
_COMP = bytes(_COMP)


def revcomp(seq):
    """dna::revcomp, used at src/aln_writer.rs:138"""
    return bytes(seq).translate(_COMP)[::-1]


def multimapq(n):
    """src/aln_writer.rs:332-340, with the f32 arithmetic of the original"""
    if n <= 1:
        return 255
    if n >= 5:
        return 0
    one = np.float32(1.0)
    v = np.float32(-10.0) * np.log10(one - one / np.float32(n), dtype=np.float32)
    return int(np.floor(v + np.float32(0.5)))  # f32::round for positive values


def format_read_name(r):
    """src/aln_writer.rs:344-349"""
    r = bytes(r)
    i = r.find(b" ")
    return r[:i] if i >= 0 else r


def format_maybe_empty(s):
    """src/aln_writer.rs:352-358"""
    s = bytes(s)
    return s if s else b"*"


def to_cigar(ops):
    """to_noodles_cigar, src/aln_writer.rs:279-323, rendered the SAM way (`*` when empty)"""
    kind = {"Match": "M", "Subst": "M", "Del": "D", "Ins": "I", "Xclip": "S", "Yclip": "N"}

    def match_op(op, n):
        if isinstance(op, tuple):  # clips carry their own length (:292-294)
            return "%d%s" % (op[1], kind[op[0]])
        return "%d%s" % (n, kind[op])

    v = []
    prev, prev_len = None, 0
    for op in ops:
        if op == "Subst":  # 'M' for both match and mismatch (:303-307)
            op = "Match"
        if prev is None or prev != op:
            if prev is not None:
                v.append(match_op(prev, prev_len))
            prev, prev_len = op, 1
        else:
            prev_len += 1
    if len(ops) > 0:
        v.append(match_op(prev, prev_len))
    return "".join(v) if v else "*"


def sam_header(tables):
    """build_sam_header, src/aln_writer.rs:256-276: one @SQ per contig *name* (the refs are collected into a
    map keyed by name, so the forward and reverse Ref of a contig collapse), then @PG ID:thermite"""
    seen, lines = set(), []
    for r in tables["refs"]:
        name = tables["names"][int(r["name_id"])]
        if name in seen:
            continue
        seen.add(name)
        lines.append("@SQ\tSN:%s\tLN:%d\n" % (name, int(r["len"])))
    lines.append("@PG\tID:thermite\n")
    return "".join(lines).encode()


def _ops(result, off, n):
    return orc.decode_ops(result.ops[int(off): int(off) + int(n)])


def sam_record(tables, name, seq, qual, aln, ops, tx_ops, multimap, hit_index):
    """aln_to_sam_record, src/aln_writer.rs:118-238"""
    strand = bool(aln["strand"])
    qseq = bytes(seq) if strand else revcomp(seq)
    qq = bytes(qual) if strand else bytes(qual)[::-1]
    flags = (0 if strand else 16) | (0 if aln["primary"] else 256)
    n_mismatch = sum(1 for o in ops if o == "Subst")
    data = ["AS:i:%d" % int(aln["score"]), "NH:i:%d" % multimap, "HI:i:%d" % hit_index, "nM:i:%d" % n_mismatch]
    t = int(aln["aln_type"])
    if t == 0:  # Exonic
        tx = int(aln["tx_or_gene_idx"])
        g = int(tables["txs"][tx]["gene_idx"])
        data.append("TX:Z:%s,+%d,%s" % (tables["tx_ids"][tx], int(aln["tx_ystart"]), to_cigar(tx_ops)))
        data.append("GX:Z:%s" % tables["gene_ids"][g])
        data.append("GN:Z:%s" % tables["gene_names"][g])
        data.append("RE:A:E")
    elif t == 1:  # Intronic
        g = int(aln["tx_or_gene_idx"])
        data.append("GX:Z:%s" % tables["gene_ids"][g])
        data.append("GN:Z:%s" % tables["gene_names"][g])
        data.append("RE:A:N")
    else:
        data.append("RE:A:I")
    ref_name = tables["names"][int(tables["refs"][int(aln["ref_id"])]["name_id"])]
    cols = [format_read_name(name), b"%d" % flags, ref_name.encode(), b"%d" % (int(aln["ystart"]) + 1),
            b"%d" % multimapq(multimap), to_cigar(ops).encode(), b"*", b"0", b"0", format_maybe_empty(qseq),
            format_maybe_empty(qq)] + [d.encode() for d in data]
    return b"\t".join(cols) + b"\n"


def unmapped_sam_record(name, seq, qual):
    """unmapped_sam_record, src/aln_writer.rs:241-253: flag 4, everything else at the builder's defaults"""
    return b"\t".join([format_read_name(name), b"4", b"*", b"0", b"255", b"*", b"*", b"0", b"0",
                       format_maybe_empty(seq), format_maybe_empty(qual)]) + b"\n"


def paf_record(tables, name, seq, aln, ops, multimap):
    """PafEntry::new + Display + write_paf, src/aln_writer.rs:47-115 (note the tab before the newline)"""
    num_match = sum(1 for o in ops if o == "Match")
    num_match_gap = sum(1 for o in ops if not (isinstance(o, tuple) and o[0] == "Yclip"))
    ref_name = tables["names"][int(tables["refs"][int(aln["ref_id"])]["name_id"])]
    return b"%s\t%d\t%d\t%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t\n" % (
        bytes(name), len(seq), int(aln["xstart"]), int(aln["xend"]), b"+" if aln["strand"] else b"-", ref_name.encode(),
        int(aln["ylen"]), int(aln["ystart"]), int(aln["yend"]), num_match, num_match_gap, multimapq(multimap))


def format_batch(tables, names, seqs, quals, result, fmt):
    """The writer loop of align_reads_from_file, src/aligner.rs:54-116, for reads already aligned
    (`result` has offsets / alns / ops like the oracle's align_batch)."""
    out = []
    for r in range(len(seqs)):
        a0, a1 = int(result.offsets[r]), int(result.offsets[r + 1])
        if a0 == a1:
            if fmt == "sam":
                out.append(unmapped_sam_record(names[r], seqs[r], quals[r]))
            continue
        for i, a in enumerate(range(a0, a1)):
            aln = result.alns[a]
            ops = _ops(result, aln["ops_off"], aln["ops_len"])
            if fmt == "sam":
                tx_ops = _ops(result, aln["tx_ops_off"], aln["tx_ops_len"]) if int(aln["aln_type"]) == 0 else []
                out.append(sam_record(tables, names[r], seqs[r], quals[r], aln, ops, tx_ops, a1 - a0, i + 1))
            else:
                out.append(paf_record(tables, names[r], seqs[r], aln, ops, a1 - a0))
    return b"".join(out)


# ------------------------------------------------------------------ BAM
# The reference writes BAM with noodles-bam 0.1.0 (`write_sam_record`, src/aligner.rs:69-76,98-108;
# header src/aligner.rs:41-46).  noodles is not in the checkout: the binary layout below is the
# SAM/BAM specification's (parity unpinned); BGZF framing is checked separately by the tests and
# parity is defined on the decompressed stream.
import struct

_SEQ_CODE = {c: i for i, c in enumerate(b"=ACMGRSVTWYHKDBN")}
_CIG_CODE = {"M": 0, "I": 1, "D": 2, "N": 3, "S": 4}


def _reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def _sq_list(tables):
    names, out = {}, []
    for r in tables["refs"]:
        nm = tables["names"][int(r["name_id"])]
        if nm not in names:
            names[nm] = len(out)
            out.append((nm, int(r["len"])))
    return names, out


def bam_header_bytes(tables):
    text = sam_header(tables)
    _, sq = _sq_list(tables)
    out = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(sq))
    for nm, ln in sq:
        out += struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln)
    return out


def _int_tag(tag, v):
    if v >= 0:
        if v <= 0xFF:
            return tag + b"C" + struct.pack("<B", v)
        if v <= 0xFFFF:
            return tag + b"S" + struct.pack("<H", v)
        return tag + b"I" + struct.pack("<I", v)
    if v >= -128:
        return tag + b"c" + struct.pack("<b", v)
    if v >= -32768:
        return tag + b"s" + struct.pack("<h", v)
    return tag + b"i" + struct.pack("<i", v)


def _bam_record(qname, flag, ref_id, pos, mapq, cigar, seq, qual, tags):
    import re

    ops = [] if cigar == "*" else [(int(n), k) for n, k in re.findall(r"(\d+)([MIDNS])", cigar)]
    ref_len = sum(n for n, k in ops if k in "MDN")
    bin_ = 4680 if pos < 0 else _reg2bin(pos, pos + max(ref_len, 1))
    body = struct.pack("<iiBBHHHiiii", ref_id, pos, len(qname) + 1, mapq, bin_, len(ops), flag, len(seq), -1, -1, 0)
    body += qname + b"\0"
    body += b"".join(struct.pack("<I", (n << 4) | _CIG_CODE[k]) for n, k in ops)
    codes = [_SEQ_CODE.get(c, _SEQ_CODE.get(c - 32, 15) if 97 <= c <= 122 else 15) for c in seq]
    if len(codes) % 2:
        codes.append(0)
    body += bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))
    body += bytes([0xFF] * len(seq)) if not qual else bytes(q - 33 for q in qual)
    body += tags
    return struct.pack("<i", len(body)) + body


def bam_stream(tables, names, seqs, quals, result):
    """the uncompressed BAM stream: header, then the records of the writer loop (src/aligner.rs:54-116)"""
    sq_of, _ = _sq_list(tables)
    out = [bam_header_bytes(tables)]
    for r in range(len(seqs)):
        a0, a1 = int(result.offsets[r]), int(result.offsets[r + 1])
        qn = format_read_name(names[r])
        if a0 == a1:
            out.append(_bam_record(qn, 4, -1, -1, 255, "*", bytes(seqs[r]), bytes(quals[r]), b""))
            continue
        for i, a in enumerate(range(a0, a1)):
            aln = result.alns[a]
            ops = _ops(result, aln["ops_off"], aln["ops_len"])
            strand = bool(aln["strand"])
            seq = bytes(seqs[r]) if strand else revcomp(seqs[r])
            qual = bytes(quals[r]) if strand else bytes(quals[r])[::-1]
            tags = (_int_tag(b"AS", int(aln["score"])) + _int_tag(b"NH", a1 - a0) + _int_tag(b"HI", i + 1)
                    + _int_tag(b"nM", sum(1 for o in ops if o == "Subst")))
            t = int(aln["aln_type"])
            if t == 0:
                tx = int(aln["tx_or_gene_idx"])
                g = int(tables["txs"][tx]["gene_idx"])
                tx_ops = _ops(result, aln["tx_ops_off"], aln["tx_ops_len"])
                tags += b"TXZ" + ("%s,+%d,%s" % (tables["tx_ids"][tx], int(aln["tx_ystart"]), to_cigar(tx_ops))).encode() + b"\0"
                tags += b"GXZ" + tables["gene_ids"][g].encode() + b"\0" + b"GNZ" + tables["gene_names"][g].encode() + b"\0" + b"REAE"
            elif t == 1:
                g = int(aln["tx_or_gene_idx"])
                tags += b"GXZ" + tables["gene_ids"][g].encode() + b"\0" + b"GNZ" + tables["gene_names"][g].encode() + b"\0" + b"REAN"
            else:
                tags += b"REAI"
            ref_name = tables["names"][int(tables["refs"][int(aln["ref_id"])]["name_id"])]
            flag = (0 if strand else 16) | (0 if aln["primary"] else 256)
            out.append(_bam_record(qn, flag, sq_of[ref_name], int(aln["ystart"]), multimapq(a1 - a0), to_cigar(ops), seq, qual, tags))
    return b"".join(out)


def bgzf_decompress(data):
    """Checks the BGZF framing block by block (gzip member with the BC extra field, sizes, CRC, the
    28-byte end-of-file block last) and returns the concatenated payload."""
    import zlib

    out, off, last_len = [], 0, None
    while off < len(data):
        hdr = data[off: off + 18]
        assert hdr[:4] == b"\x1f\x8b\x08\x04" and hdr[10:12] == b"\x06\x00" and hdr[12:16] == b"BC\x02\x00", "not a BGZF block"
        bsize = struct.unpack("<H", hdr[16:18])[0] + 1
        block = data[off: off + bsize]
        assert len(block) == bsize, "truncated BGZF block"
        payload = zlib.decompress(block[18:-8], -15)
        crc, isize = struct.unpack("<II", block[-8:])
        assert isize == len(payload) and crc == (zlib.crc32(payload) & 0xFFFFFFFF) and isize <= 65536
        out.append(payload)
        last_len = isize
        off += bsize
    assert last_len == 0 and data[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"), "no EOF block"
    return b"".join(out)
