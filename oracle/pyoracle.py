"""ctypes binding of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  thermite_amd/ never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libthermite_oracle.so")

# numpy mirrors of the PODs in thermite_oracle.h (== include/thermite.h)
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
MEM_DT = np.dtype([("ref_idx", "<u8"), ("query_idx", "<u4"), ("len", "<u4")])
ALN_DT = np.dtype(
    [
        ("ystart", "<u8"),
        ("yend", "<u8"),
        ("ylen", "<u8"),
        ("ops_off", "<u8"),
        ("tx_ystart", "<u8"),
        ("tx_yend", "<u8"),
        ("tx_ylen", "<u8"),
        ("tx_ops_off", "<u8"),
        ("score", "<i4"),
        ("ref_id", "<u4"),
        ("xstart", "<u4"),
        ("xend", "<u4"),
        ("xlen", "<u4"),
        ("ops_len", "<u4"),
        ("tx_or_gene_idx", "<u4"),
        ("tx_score", "<i4"),
        ("tx_xstart", "<u4"),
        ("tx_xend", "<u4"),
        ("tx_ops_len", "<u4"),
        ("strand", "u1"),
        ("primary", "u1"),
        ("aln_type", "u1"),
        ("pad_", "u1"),
    ]
)
SWG_DT = np.dtype([("ops_off", "<u8"), ("ops_len", "<u4"), ("score", "<i4"), ("xend", "<u4"), ("yend", "<u4")])
assert ALN_DT.itemsize == 112 and MEM_DT.itemsize == 16 and SWG_DT.itemsize == 24
assert REF_DT.itemsize == 32 and EXON_DT.itemsize == 24 and TX_DT.itemsize == 40


class Opts(C.Structure):
    _fields_ = [
        ("min_seed_len", C.c_uint64),
        ("min_aln_score_percent", C.c_float),
        ("min_aln_score", C.c_int32),
        ("multimap_score_range", C.c_uint64),
        ("intron_mode", C.c_int32),
        ("reserved", C.c_int32),
    ]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in ("thermite_oracle.cpp", "thermite_oracle.h")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32
    L.orc_swg_new.restype = vp
    L.orc_swg_new.argtypes = [u64, i32, i32, i32, i32]
    L.orc_swg_free.argtypes = [vp]
    L.orc_swg_extend.restype = i32
    L.orc_swg_extend.argtypes = [vp, vp, u64, vp, u64, u64, i32, vp, vp, vp, vp, u64, vp]
    L.orc_swg_phase1_breaks.restype = u64
    L.orc_swg_phase1_breaks.argtypes = [vp]
    L.orc_swg_cells.restype = u64
    L.orc_swg_cells.argtypes = [vp]
    L.orc_swg_extend_batch.restype = vp
    L.orc_swg_extend_batch.argtypes = [vp, vp, vp, vp, vp, vp, u32, u64]
    L.orc_extend_left_right.restype = i32
    L.orc_extend_left_right.argtypes = [vp, vp, u64, u64, u64, u64, vp, u64, u64, i32, vp, vp, vp, vp, vp, vp, u64, vp]
    L.orc_extend_seed_match.argtypes = [vp, u64, vp, vp, u64]
    L.orc_intersect.restype = i32
    L.orc_intersect.argtypes = [u64, u64, u64, u64]
    L.orc_lift_mem_to_tx.restype = i32
    L.orc_lift_mem_to_tx.argtypes = [vp, vp, u64, vp]
    L.orc_lift_tx_to_gx.restype = i32
    L.orc_lift_tx_to_gx.argtypes = [vp, u64, u64, u64, vp, u64, vp, vp, vp, u64, vp]
    L.orc_filter_overlapping.restype = u64
    L.orc_filter_overlapping.argtypes = [vp, vp, vp, vp, vp, u64, vp]
    L.orc_suffix_array_naive.argtypes = [vp, u64, vp]
    L.orc_suffix_array_verify.restype = i32
    L.orc_suffix_array_verify.argtypes = [vp, u64, vp]
    L.orc_index_create.restype = vp
    L.orc_index_create.argtypes = [vp, u64, vp, u32, vp, u32, vp, u64, vp, u64, vp, u32, vp, u32, vp, u32, u32]
    L.orc_index_create64.restype = vp
    L.orc_index_create64.argtypes = [vp, u64, vp, u32, vp, u32, vp, u64, vp, u64, vp, u32, vp, u32, vp, u32, u32, u32]
    L.orc_index_free.argtypes = [vp]
    L.orc_all_smems_batch.restype = vp
    L.orc_all_smems_batch.argtypes = [vp, vp, vp, u64, u64]
    L.orc_all_smems_batch_ms.restype = vp
    L.orc_all_smems_batch_ms.argtypes = [vp, vp, vp, u64, u64]
    L.orc_exon_tree_find.restype = u64
    L.orc_exon_tree_find.argtypes = [vp, u64, u64, vp, u64]
    L.orc_gene_tree_find.restype = u64
    L.orc_gene_tree_find.argtypes = [vp, u64, u64, vp, u64]
    L.orc_align_batch.restype = vp
    L.orc_align_batch.argtypes = [vp, vp, vp, vp, u64, u32]
    L.orc_result_free.argtypes = [vp]
    for f in ("orc_result_n", "orc_result_n_items", "orc_result_n_op_bytes"):
        getattr(L, f).restype = u64
        getattr(L, f).argtypes = [vp]
    for f in (
        "orc_result_offsets",
        "orc_result_alns",
        "orc_result_mems",
        "orc_result_swg",
        "orc_result_ops",
        "orc_result_counters",
    ):
        getattr(L, f).restype = vp
        getattr(L, f).argtypes = [vp]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _bytes_arr(b):
    return np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b, np.uint8)


def _copy(ptr, count, dtype):
    if count == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    nbytes = count * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


# ----------------------------------------------------------------- op streams
OP_NAMES = ["Match", "Subst", "Del", "Ins", "Xclip", "Yclip"]


def decode_ops(b):
    """serialised op stream -> list like ['Match', ('Xclip', 3), ...]"""
    b = bytes(b)
    out, i = [], 0
    while i < len(b):
        k = b[i]
        i += 1
        if k >= 4:
            n = int.from_bytes(b[i : i + 4], "little")
            i += 4
            out.append((OP_NAMES[k], n))
        else:
            out.append(OP_NAMES[k])
    return out


def encode_ops(ops):
    out = bytearray()
    for o in ops:
        if isinstance(o, tuple):
            out.append(OP_NAMES.index(o[0]))
            out += int(o[1]).to_bytes(4, "little")
        else:
            out.append(OP_NAMES.index(o))
    return bytes(out)


# ------------------------------------------------------------------ SwgExtend
class Swg:
    """SwgExtend (src/swg.rs) with the aligner's scoring (src/aligner.rs:140)."""

    def __init__(self, max_band_width, gap_open=-1, gap_extend=-1, match=1, mismatch=-1):
        self.h = lib().orc_swg_new(max_band_width, gap_open, gap_extend, match, mismatch)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_swg_free(self.h)
            self.h = None

    def extend(self, x, y, band_width, x_drop):
        xa, ya = _bytes_arr(x), _bytes_arr(y)
        score, xend, yend, olen = C.c_int32(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        cap = 5 * (len(xa) + len(ya) + 8)
        buf = np.zeros(cap, np.uint8)
        rc = lib().orc_swg_extend(
            self.h, _ptr(xa), len(xa), _ptr(ya), len(ya), band_width, x_drop,
            C.addressof(score), C.addressof(xend), C.addressof(yend), _ptr(buf), cap, C.addressof(olen),
        )
        if rc != 0:
            raise RuntimeError("orc_swg_extend rc=%d" % rc)
        return dict(score=score.value, xend=xend.value, yend=yend.value, ystart=0, xstart=0, ylen=len(ya),
                    xlen=len(xa), ops=decode_ops(buf[: olen.value]))

    def extend_left_right(self, ref_seq, hit, read, band_width, x_drop):
        ra, qa = _bytes_arr(ref_seq), _bytes_arr(read)
        score = C.c_int32()
        ys, xs, ye, xe, olen = (C.c_uint64() for _ in range(5))
        cap = 5 * (len(ra) + len(qa) + 8)
        buf = np.zeros(cap, np.uint8)
        rc = lib().orc_extend_left_right(
            self.h, _ptr(ra), len(ra), hit[0], hit[1], hit[2], _ptr(qa), len(qa), band_width, x_drop,
            C.addressof(score), C.addressof(ys), C.addressof(xs), C.addressof(ye), C.addressof(xe), _ptr(buf), cap,
            C.addressof(olen),
        )
        if rc != 0:
            raise RuntimeError("orc_extend_left_right rc=%d" % rc)
        return dict(score=score.value, ystart=ys.value, xstart=xs.value, yend=ye.value, xend=xe.value, ylen=len(ra),
                    xlen=len(qa), ops=decode_ops(buf[: olen.value]))

    @property
    def phase1_breaks(self):
        return lib().orc_swg_phase1_breaks(self.h)

    @property
    def cells(self):
        return lib().orc_swg_cells(self.h)


class Result:
    def __init__(self, h, kind):
        L = lib()
        self.n = L.orc_result_n(h)
        n_items = L.orc_result_n_items(h)
        self.counters = _copy(L.orc_result_counters(h), 16, "<u8")
        self.ops = _copy(L.orc_result_ops(h), L.orc_result_n_op_bytes(h), np.uint8)
        if kind == "aln":
            self.offsets = _copy(L.orc_result_offsets(h), self.n + 1, "<u8")
            self.alns = _copy(L.orc_result_alns(h), n_items, ALN_DT)
        elif kind == "mem":
            self.offsets = _copy(L.orc_result_offsets(h), self.n + 1, "<u8")
            self.mems = _copy(L.orc_result_mems(h), n_items, MEM_DT)
        elif kind == "swg":
            self.swg = _copy(L.orc_result_swg(h), n_items, SWG_DT)
        L.orc_result_free(h)


def swg_extend_batch(x_bases, x_off, y_bases, y_off, bw, xd, max_bw):
    x_bases, y_bases = _bytes_arr(x_bases), _bytes_arr(y_bases)
    x_off = np.ascontiguousarray(x_off, "<u8")
    y_off = np.ascontiguousarray(y_off, "<u8")
    bw = np.ascontiguousarray(bw, "<u4")
    xd = np.ascontiguousarray(xd, "<i4")
    h = lib().orc_swg_extend_batch(_ptr(x_bases), _ptr(x_off), _ptr(y_bases), _ptr(y_off), _ptr(bw), _ptr(xd), max_bw,
                                   len(bw))
    return Result(h, "swg")


def extend_seed_match(ref_seq, hit, read):
    ra, qa = _bytes_arr(ref_seq), _bytes_arr(read)
    m = np.zeros(1, MEM_DT)
    m["ref_idx"], m["query_idx"], m["len"] = hit
    lib().orc_extend_seed_match(_ptr(ra), len(ra), _ptr(m), _ptr(qa), len(qa))
    return int(m["ref_idx"][0]), int(m["query_idx"][0]), int(m["len"][0])


def _exons(exons):
    e = np.zeros(len(exons), EXON_DT)
    for i, (s, t, tx) in enumerate(exons):
        e[i]["start"], e[i]["end"], e[i]["tx_idx"] = s, t, tx
    return e


def lift_mem_to_tx(mem, exons):
    m = np.zeros(1, MEM_DT)
    m["ref_idx"], m["query_idx"], m["len"] = mem
    o = np.zeros(1, MEM_DT)
    e = _exons(exons)
    rc = lib().orc_lift_mem_to_tx(_ptr(m), _ptr(e), len(e), _ptr(o))
    if rc != 0:
        raise RuntimeError("lift_mem_to_tx rc=%d" % rc)
    return int(o["ref_idx"][0]), int(o["query_idx"][0]), int(o["len"][0])


def lift_tx_to_gx(ops, ystart, yend, exons):
    b = np.frombuffer(encode_ops(ops), np.uint8)
    e = _exons(exons)
    oys, oye, olen = C.c_uint64(), C.c_uint64(), C.c_uint64()
    cap = 5 * (len(b) + len(e) + 8)
    buf = np.zeros(cap, np.uint8)
    rc = lib().orc_lift_tx_to_gx(_ptr(b), len(b), ystart, yend, _ptr(e), len(e), C.addressof(oys), C.addressof(oye),
                                 _ptr(buf), cap, C.addressof(olen))
    if rc != 0:
        raise RuntimeError("lift_tx_to_gx rc=%d" % rc)
    return dict(ystart=oys.value, yend=oye.value, ops=decode_ops(buf[: olen.value]))


def filter_overlapping(name_rank, strand, ystart, yend, score):
    n = len(score)
    kept = np.zeros(n, "<u8")
    k = lib().orc_filter_overlapping(
        _ptr(np.ascontiguousarray(name_rank, "<u4")), _ptr(np.ascontiguousarray(strand, "u1")),
        _ptr(np.ascontiguousarray(ystart, "<u8")), _ptr(np.ascontiguousarray(yend, "<u8")),
        _ptr(np.ascontiguousarray(score, "<i4")), n, _ptr(kept),
    )
    return [int(v) for v in kept[:k]]


def suffix_array_naive(text):
    t = _bytes_arr(text)
    sa = np.zeros(len(t), "<u4")
    lib().orc_suffix_array_naive(_ptr(t), len(t), _ptr(sa))
    return sa


def suffix_array_verify(text, sa):
    t = _bytes_arr(text)
    sa = np.ascontiguousarray(sa, "<u4")
    return bool(lib().orc_suffix_array_verify(_ptr(t), len(t), _ptr(sa)))


class Index:
    """Oracle index: FMD index + AVL interval trees built from raw tables.

    `tables` is the dict produced by thermite_amd.refdata (text, refs, txs,
    exons, tx_seq, genes, name_rank); `sa` an optional precomputed suffix array
    (verified in O(n)).
    """

    def __init__(self, tables, sa=None, sa_sampling_rate=32, occ_sampling_rate=128, verify=True, keep_sa=True):
        """A 64-bit suffix array (dtype u8) takes the usize-wide path of the index -- the reference's own width
        (src/index.rs:103-111) and the only one for texts of 2^32 - 1 symbols and more.  verify=False / keep_sa=False:
        for texts of billions of symbols (the check and the plain array cost 8 n bytes each)."""
        t = tables
        self.tables = t
        wide = sa is not None and np.asarray(sa).dtype.itemsize == 8
        args = (_ptr(t["text"]), len(t["text"]), _ptr(t["refs"]), len(t["refs"]), _ptr(t["txs"]), len(t["txs"]),
                _ptr(t["exons"]), len(t["exons"]), _ptr(t["tx_seq"]), len(t["tx_seq"]), _ptr(t["genes"]), len(t["genes"]),
                _ptr(t["name_rank"]), len(t["name_rank"]))
        if wide:
            sa_arr = np.ascontiguousarray(sa, "<u8")
            self.h = lib().orc_index_create64(*args, _ptr(sa_arr), sa_sampling_rate, occ_sampling_rate,
                                              (0 if verify else 1) | (0 if keep_sa else 2))
        else:
            sa_arr = None if sa is None else np.ascontiguousarray(sa, "<u4")
            self.h = lib().orc_index_create(*args, _ptr(sa_arr), sa_sampling_rate, occ_sampling_rate)
        self.wide = wide
        if not self.h:
            raise RuntimeError("orc_index_create failed (invalid suffix array or text too long)")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_index_free(self.h)
            self.h = None

    def all_smems(self, bases, offsets, min_seed_len, ms=False):
        bases = _bytes_arr(bases)
        offsets = np.ascontiguousarray(offsets, "<u8")
        f = lib().orc_all_smems_batch_ms if ms else lib().orc_all_smems_batch
        return Result(f(self.h, _ptr(bases), _ptr(offsets), len(offsets) - 1, min_seed_len), "mem")

    def exon_tree_find(self, s, e, cap=4096):
        out = np.zeros(cap, "<u4")
        c = lib().orc_exon_tree_find(self.h, s, e, _ptr(out), cap)
        return [int(v) for v in out[: min(c, cap)]]

    def gene_tree_find(self, s, e, cap=4096):
        out = np.zeros(cap, "<u4")
        c = lib().orc_gene_tree_find(self.h, s, e, _ptr(out), cap)
        return [int(v) for v in out[: min(c, cap)]]

    def align_batch(self, bases, offsets, opts, n_threads=1):
        bases = _bytes_arr(bases)
        offsets = np.ascontiguousarray(offsets, "<u8")
        o = Opts(opts["min_seed_len"], opts["min_aln_score_percent"], opts["min_aln_score"],
                 opts["multimap_score_range"], int(bool(opts["intron_mode"])), 0)
        h = lib().orc_align_batch(self.h, C.addressof(o), _ptr(bases), _ptr(offsets), len(offsets) - 1, n_threads)
        return Result(h, "aln")
