"""ctypes binding of the C ABI in include/thermite.h (libthermite_amd.so).

This is host plumbing for tests and bench.py; the product is the shared
library.  There is no fallback: if the HIP library is missing, or no MI355X is
visible when an aligner is created, this raises.
"""
import ctypes as C
import os

import numpy as np

from .refdata import EXON_DT, REF_DT, SPAN_DT, TX_DT  # noqa: F401  (re-exported)

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("THM_LIB") or os.path.join(_HERE, "_build", "libthermite_amd.so")

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

N_COUNTERS = 16
COUNTER_NAMES = ["reads", "aligned", "unmapped", "alns", "exonic", "intronic", "intergenic", "smems", "hits",
                 "swg_calls", "dp_cells", "dp_cols", "op_bytes", "window_bytes"]
N_TIMINGS = 8
TIMING_NAMES = ["seed", "plan", "extend", "compact", "total"]

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_OUT_OF_CONTRACT, ERR_OOM, ERR_INTERNAL = (
    0, -1, -2, -3, -4, -5, -6, -7)


class ThermiteError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("thermite_amd error %d: %s" % (code, msg))
        self.code = code


class Opts(C.Structure):
    """thm_align_opts == AlignOpts (reference src/aligner.rs:452-464)."""

    _fields_ = [
        ("min_seed_len", C.c_uint64),
        ("min_aln_score_percent", C.c_float),
        ("min_aln_score", C.c_int32),
        ("multimap_score_range", C.c_uint64),
        ("intron_mode", C.c_int32),
        ("reserved", C.c_int32),
    ]


# reference defaults: src/main.rs:115-140, src/wrapper.rs:40-46
DEFAULT_OPTS = dict(min_seed_len=20, min_aln_score_percent=0.66, min_aln_score=30, multimap_score_range=1,
                    intron_mode=False)
# flags of the reference's chrM / chr21 runs: -k20 -s0 --intron-mode (data/Makefile:30,39)
CI_OPTS = dict(min_seed_len=20, min_aln_score_percent=0.0, min_aln_score=30, multimap_score_range=1, intron_mode=True)


class BatchView(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_alns", C.c_uint64), ("n_op_bytes", C.c_uint64),
                ("read_aln_off", C.c_void_p), ("alns", C.c_void_p), ("ops", C.c_void_p),
                ("n_failed_reads", C.c_uint64), ("read_status", C.c_void_p)]


class MemsView(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_mems", C.c_uint64), ("read_mem_off", C.c_void_p), ("mems", C.c_void_p)]


class SwgView(C.Structure):
    _fields_ = [("n", C.c_uint64), ("n_op_bytes", C.c_uint64), ("alns", C.c_void_p), ("ops", C.c_void_p)]


# every symbol include/thermite.h declares
ABI_SYMBOLS = [
    "thm_index_create_in_memory", "thm_index_free", "thm_index_text_len", "thm_index_suffix_array",
    "thm_index_idx_to_ref", "thm_build_suffix_array", "thm_aligner_create", "thm_aligner_free", "thm_last_error",
    "thm_aligner_set_opts", "thm_aligner_stream", "thm_align_batch", "thm_batch_upload", "thm_batch_run",
    "thm_batch_sync", "thm_batch_fetch", "thm_smems_batch", "thm_swg_extend_batch", "thm_counters_get",
    "thm_counters_reset", "thm_counters_device_ptr", "thm_timings_get", "thm_version", "thm_device_count",
    "thm_aligner_index", "thm_comm_unique_id", "thm_comm_create", "thm_comm_free", "thm_counters_allreduce",
    "thm_index_create_in_memory_ex", "thm_index_coord_bytes", "thm_index_suffix_array64", "thm_build_suffix_array64", "thm_build_suffix_array_gpu",
]
# every symbol include/thermite_io.h declares
IO_ABI_SYMBOLS = [
    "thm_index_create_from_files", "thm_index_set_names", "thm_index_save", "thm_index_load", "thm_index_tables",
    "thm_index_contig_name", "thm_index_tx_id", "thm_index_gene_id", "thm_index_gene_name", "thm_fastq_open",
    "thm_fastq_next_batch", "thm_fastq_close", "thm_writer_create", "thm_writer_free", "thm_writer_header",
    "thm_writer_format_batch", "thm_writer_trailer", "thm_align_files", "thm_align_files_multi",
]
ERR_IO, ERR_FORMAT = -8, -9
FMT_PAF, FMT_SAM, FMT_BAM = 0, 1, 2

_lib = None


def lib():
    """Load libthermite_amd.so; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ThermiteError(ERR_INTERNAL, "HIP library not built: %s is missing (run __graft_entry__.build())" % SO_PATH)
    L = C.CDLL(SO_PATH)
    vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32
    L.thm_index_create_in_memory.restype = i32
    L.thm_index_create_in_memory.argtypes = [vp, u64, vp, u32, vp, u32, vp, u64, vp, u64, vp, u32, vp, u32, vp, vp]
    L.thm_index_create_in_memory_ex.restype = i32
    L.thm_index_create_in_memory_ex.argtypes = [vp, u64, vp, u32, vp, u32, vp, u64, vp, u64, vp, u32, vp, u32, vp, u32, u32, vp]
    L.thm_index_coord_bytes.restype = u32
    L.thm_index_coord_bytes.argtypes = [vp]
    L.thm_index_suffix_array64.restype = vp
    L.thm_index_suffix_array64.argtypes = [vp]
    L.thm_build_suffix_array64.restype = i32
    L.thm_build_suffix_array64.argtypes = [vp, u64, vp]
    L.thm_build_suffix_array_gpu.restype = i32
    L.thm_build_suffix_array_gpu.argtypes = [vp, u64, vp, u32]
    if hasattr(L, "thm_debug_check_lut"):
        L.thm_debug_check_lut.restype = i32
        L.thm_debug_check_lut.argtypes = [vp]
    L.thm_index_free.argtypes = [vp]
    L.thm_index_text_len.restype = u64
    L.thm_index_text_len.argtypes = [vp]
    L.thm_index_suffix_array.restype = vp
    L.thm_index_suffix_array.argtypes = [vp]
    L.thm_index_idx_to_ref.restype = i32
    L.thm_index_idx_to_ref.argtypes = [vp, u64, vp]
    L.thm_build_suffix_array.restype = i32
    L.thm_build_suffix_array.argtypes = [vp, u64, vp]
    L.thm_aligner_create.restype = i32
    L.thm_aligner_create.argtypes = [vp, vp, i32, vp]
    L.thm_aligner_free.argtypes = [vp]
    L.thm_last_error.restype = C.c_char_p
    L.thm_last_error.argtypes = [vp]
    L.thm_aligner_set_opts.restype = i32
    L.thm_aligner_set_opts.argtypes = [vp, vp]
    L.thm_aligner_stream.restype = vp
    L.thm_aligner_stream.argtypes = [vp]
    L.thm_align_batch.restype = i32
    L.thm_align_batch.argtypes = [vp, vp, vp, u64, vp]
    L.thm_batch_upload.restype = i32
    L.thm_batch_upload.argtypes = [vp, vp, vp, u64]
    for f in ("thm_batch_run", "thm_batch_sync", "thm_counters_reset"):
        getattr(L, f).restype = i32
        getattr(L, f).argtypes = [vp]
    L.thm_batch_fetch.restype = i32
    L.thm_batch_fetch.argtypes = [vp, vp]
    L.thm_smems_batch.restype = i32
    L.thm_smems_batch.argtypes = [vp, vp, vp, u64, u64, vp]
    L.thm_swg_extend_batch.restype = i32
    L.thm_swg_extend_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, u64, vp]
    L.thm_counters_get.restype = i32
    L.thm_counters_get.argtypes = [vp, vp]
    L.thm_counters_device_ptr.restype = vp
    L.thm_counters_device_ptr.argtypes = [vp]
    L.thm_timings_get.restype = i32
    L.thm_timings_get.argtypes = [vp, vp]
    L.thm_version.restype = C.c_char_p
    L.thm_device_count.restype = i32
    L.thm_aligner_index.restype = vp
    L.thm_aligner_index.argtypes = [vp]
    L.thm_comm_unique_id.restype = i32
    L.thm_comm_unique_id.argtypes = [vp]
    L.thm_comm_create.restype = i32
    L.thm_comm_create.argtypes = [vp, i32, i32, i32, vp]
    L.thm_comm_free.argtypes = [vp]
    L.thm_counters_allreduce.restype = i32
    L.thm_counters_allreduce.argtypes = [vp, vp]
    # ---- include/thermite_io.h ----
    cp = C.c_char_p
    L.thm_index_create_from_files.restype = i32
    L.thm_index_create_from_files.argtypes = [cp, cp, vp]
    L.thm_index_set_names.restype = i32
    L.thm_index_set_names.argtypes = [vp, vp, u32, vp, u32, vp, vp, u32]
    L.thm_index_save.restype = i32
    L.thm_index_save.argtypes = [vp, cp]
    L.thm_index_load.restype = i32
    L.thm_index_load.argtypes = [cp, vp]
    L.thm_index_tables.restype = i32
    L.thm_index_tables.argtypes = [vp, vp]
    for f in ("thm_index_contig_name", "thm_index_tx_id", "thm_index_gene_id", "thm_index_gene_name"):
        getattr(L, f).restype = cp
        getattr(L, f).argtypes = [vp, u32]
    L.thm_fastq_open.restype = i32
    L.thm_fastq_open.argtypes = [cp, vp]
    L.thm_fastq_next_batch.restype = i32
    L.thm_fastq_next_batch.argtypes = [vp, u64, vp]
    L.thm_fastq_close.argtypes = [vp]
    if hasattr(L, "thm_debug_deflate_block"):
        L.thm_debug_deflate_block.restype = i32
        L.thm_debug_deflate_block.argtypes = [vp, u64, vp, u64, C.POINTER(u64)]
    if hasattr(L, "thm_debug_gunzip_mt"):
        L.thm_debug_gunzip_mt.restype = i32
        L.thm_debug_gunzip_mt.argtypes = [C.c_char_p, u64, u32, vp, u64, C.POINTER(u64)]
    if hasattr(L, "thm_debug_gunzip"):
        L.thm_debug_gunzip.restype = i32
        L.thm_debug_gunzip.argtypes = [C.c_char_p, u64, vp, u64, C.POINTER(u64)]
    if hasattr(L, "thm_debug_fastq_blocks"):
        L.thm_debug_fastq_blocks.restype = i32
        L.thm_debug_fastq_blocks.argtypes = [vp, u64, vp]
    L.thm_writer_create.restype = i32
    L.thm_writer_create.argtypes = [vp, i32, u32, vp]
    L.thm_writer_free.argtypes = [vp]
    L.thm_writer_header.restype = i32
    L.thm_writer_header.argtypes = [vp, vp]
    L.thm_writer_trailer.restype = i32
    L.thm_writer_trailer.argtypes = [vp, vp]
    L.thm_writer_format_batch.restype = i32
    L.thm_writer_format_batch.argtypes = [vp, vp, vp, vp]
    L.thm_align_files.restype = i32
    L.thm_align_files.argtypes = [vp, vp, u32, cp, i32, u64, u32, vp]
    L.thm_align_files_multi.restype = i32
    L.thm_align_files_multi.argtypes = [vp, u32, vp, u32, cp, i32, u64, u32, vp]
    if hasattr(L, "thm_debug_prof_get"):
        L.thm_debug_prof_get.restype = i32
        L.thm_debug_prof_get.argtypes = [vp, vp, i32]
    if hasattr(L, "thm_debug_set_pool_caps"):
        L.thm_debug_set_pool_caps.restype = i32
        L.thm_debug_set_pool_caps.argtypes = [vp, u64, u64, u64, vp]
    if hasattr(L, "thm_debug_set_flags"):
        L.thm_debug_set_flags.restype = i32
        L.thm_debug_set_flags.argtypes = [vp, C.c_uint32]
        L.thm_debug_set_band_clip.restype = i32
        L.thm_debug_set_band_clip.argtypes = [vp, C.c_uint32]
        L.thm_debug_tpr_stats.restype = i32
        L.thm_debug_tpr_stats.argtypes = [vp, vp]
    if hasattr(L, "thm_debug_calib_gather"):
        L.thm_debug_calib_gather.restype = i32
        L.thm_debug_calib_gather.argtypes = [vp, i32, u64, vp]
    if hasattr(L, "thm_debug_wave_prims"):
        L.thm_debug_wave_prims.restype = i32
        L.thm_debug_wave_prims.argtypes = [vp, vp, vp]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _copy(ptr, count, dtype, copy=True):
    if count == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    nbytes = int(count) * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    a = np.frombuffer(buf, dtype=dtype, count=int(count))
    return a.copy() if copy else a


def _u8(b):
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, np.uint8)
    return np.frombuffer(bytes(b), np.uint8)


def build_suffix_array_gpu(text, wide=False):
    """the suffix array of `text` built on the current HIP device (thm_build_suffix_array_gpu); raises ThermiteError
    with ERR_NO_DEVICE / ERR_OOM / ERR_UNSUPPORTED when the host builder has to do it"""
    text = np.ascontiguousarray(text, np.uint8)
    out = np.empty(len(text), "<u8" if wide else "<u4")
    rc = lib().thm_build_suffix_array_gpu(_ptr(text), len(text), _ptr(out), 8 if wide else 4)
    if rc != 0:
        raise ThermiteError(rc, _last_error())
    return out


def build_suffix_array(text, wide=False):
    """suffix array of `text`: u32 entries, or u64 (`wide`, any text length)"""
    t = _u8(text)
    sa = np.zeros(len(t), "<u8" if wide else "<u4")
    f = lib().thm_build_suffix_array64 if wide else lib().thm_build_suffix_array
    rc = f(_ptr(t), len(t), _ptr(sa))
    if rc != 0:
        raise ThermiteError(rc, "thm_build_suffix_array")
    return sa


INDEX_WIDE = 1


class TablesView(C.Structure):
    _fields_ = [("n_text", C.c_uint64), ("text", C.c_void_p), ("n_refs", C.c_uint32), ("refs", C.c_void_p),
                ("n_txs", C.c_uint32), ("txs", C.c_void_p), ("n_exons", C.c_uint64), ("exons", C.c_void_p),
                ("n_tx_seq", C.c_uint64), ("tx_seq", C.c_void_p), ("n_genes", C.c_uint32), ("genes", C.c_void_p),
                ("name_rank", C.c_void_p), ("n_contigs", C.c_uint32)]


class ReadBatch(C.Structure):
    """thm_read_batch"""

    _fields_ = [("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("bases", C.c_void_p), ("offsets", C.c_void_p),
                ("quals", C.c_void_p), ("names", C.c_void_p), ("name_off", C.c_void_p)]


class Text(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_uint64)]


class RunStats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_aligned_reads", C.c_uint64), ("n_records", C.c_uint64),
                ("n_batches", C.c_uint64), ("n_output_bytes", C.c_uint64), ("parse_s", C.c_double),
                ("gpu_s", C.c_double), ("format_s", C.c_double), ("write_s", C.c_double), ("wall_s", C.c_double)]


def _last_error(h=None):
    return (lib().thm_last_error(h) or b"").decode()


def _cstr_array(strs):
    arr = (C.c_char_p * len(strs))(*[s.encode() if isinstance(s, str) else bytes(s) for s in strs])
    return arr


class Index:
    """thm_index: the in-memory index (stands for reference Index, src/index.rs:39-44)."""

    @classmethod
    def _wrap(cls, h):
        self = cls.__new__(cls)
        self.h = h
        self.tables = None
        return self

    @classmethod
    def from_files(cls, fasta_path, gtf_path):
        """Index::create_from_files (src/index.rs:52-223) in the native library."""
        h = C.c_void_p()
        rc = lib().thm_index_create_from_files(str(fasta_path).encode(), str(gtf_path).encode(), C.byref(h))
        if rc != 0:
            raise ThermiteError(rc, _last_error())
        return cls._wrap(h)

    @classmethod
    def load(cls, path):
        h = C.c_void_p()
        rc = lib().thm_index_load(str(path).encode(), C.byref(h))
        if rc != 0:
            raise ThermiteError(rc, _last_error())
        return cls._wrap(h)

    def save(self, path):
        rc = lib().thm_index_save(self.h, str(path).encode())
        if rc != 0:
            raise ThermiteError(rc, _last_error())

    def set_names(self, contig_names, tx_ids, gene_ids, gene_names):
        a, b, c, d = (_cstr_array(x) for x in (contig_names, tx_ids, gene_ids, gene_names))
        rc = lib().thm_index_set_names(self.h, a, len(contig_names), b, len(tx_ids), c, d, len(gene_ids))
        if rc != 0:
            raise ThermiteError(rc, _last_error())

    def native_tables(self):
        """Copy of the tables the native index holds, in the layout of refdata.build_tables."""
        v = TablesView()
        rc = lib().thm_index_tables(self.h, C.byref(v))
        if rc != 0:
            raise ThermiteError(rc, "thm_index_tables")
        L = lib()
        dec = lambda f, n: [f(self.h, i).decode() for i in range(n)]
        names = dec(L.thm_index_contig_name, v.n_contigs)
        refs = _copy(v.refs, v.n_refs, REF_DT)
        rank_per_ref = _copy(v.name_rank, v.n_refs, "<u4")
        name_rank = np.zeros(v.n_contigs, "<u4")
        if v.n_contigs:
            name_rank[refs["name_id"]] = rank_per_ref
        have = v.n_contigs > 0
        return dict(
            text=_copy(v.text, v.n_text, np.uint8), refs=refs, names=names, name_rank=name_rank,
            txs=_copy(v.txs, v.n_txs, TX_DT), exons=_copy(v.exons, v.n_exons, EXON_DT),
            tx_seq=_copy(v.tx_seq, v.n_tx_seq, np.uint8), genes=_copy(v.genes, v.n_genes, SPAN_DT),
            gene_ids=dec(L.thm_index_gene_id, v.n_genes) if have else [],
            gene_names=dec(L.thm_index_gene_name, v.n_genes) if have else [],
            tx_ids=dec(L.thm_index_tx_id, v.n_txs) if have else [],
        )

    def __init__(self, tables, sa=None, wide=False):
        """`wide`: 64-bit text positions and ranks inside the index even for a small text (a text of 2^31
        symbols or more is wide by itself; THM_FORCE_WIDE=1 in the environment forces it for every index)"""
        t = tables
        self.tables = t
        h = C.c_void_p()
        sa_arr, sa_bytes = None, 0
        if sa is not None:
            sa_bytes = 8 if np.asarray(sa).dtype.itemsize == 8 else 4
            sa_arr = np.ascontiguousarray(sa, "<u8" if sa_bytes == 8 else "<u4")
        rc = lib().thm_index_create_in_memory_ex(
            _ptr(t["text"]), len(t["text"]), _ptr(t["refs"]), len(t["refs"]), _ptr(t["txs"]), len(t["txs"]),
            _ptr(t["exons"]), len(t["exons"]), _ptr(t["tx_seq"]), len(t["tx_seq"]), _ptr(t["genes"]), len(t["genes"]),
            _ptr(t["name_rank"]), len(t["name_rank"]), _ptr(sa_arr), sa_bytes, INDEX_WIDE if wide else 0, C.byref(h),
        )
        if rc != 0:
            raise ThermiteError(rc, (lib().thm_last_error(None) or b"").decode())
        self.h = h
        if t.get("names") and "tx_ids" in t:
            self.set_names(t["names"], t["tx_ids"], t["gene_ids"], t["gene_names"])

    @property
    def coord_bytes(self):
        return lib().thm_index_coord_bytes(self.h)

    def suffix_array(self):
        n = lib().thm_index_text_len(self.h)
        if self.coord_bytes == 8:
            return _copy(lib().thm_index_suffix_array64(self.h), n, "<u8")
        return _copy(lib().thm_index_suffix_array(self.h), n, "<u4")

    def check_lut(self):
        """test hook: the k-mer table (built by counting) equals the one read off the suffix array"""
        return lib().thm_debug_check_lut(self.h) == 0

    def idx_to_ref(self, idx):
        off = C.c_uint64()
        r = lib().thm_index_idx_to_ref(self.h, idx, C.byref(off))
        if r < 0:
            raise ThermiteError(r, "idx_to_ref")
        return r, off.value

    def close(self):
        if getattr(self, "h", None):
            lib().thm_index_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class BatchResult:
    """Canonical host result of one aligned batch: thm_batch_view copied out, or (copy=False) numpy views of the
    aligner's pinned result set itself -- what a C caller gets; valid until the second fetch after this one
    (include/thermite.h, thm_batch_view)."""

    def __init__(self, view, copy=True):
        self.n_reads = view.n_reads
        self.offsets = _copy(view.read_aln_off, view.n_reads + 1, "<u8", copy)
        self.alns = _copy(view.alns, view.n_alns, ALN_DT, copy)
        self.ops = _copy(view.ops, view.n_op_bytes, np.uint8, copy)
        self.n_failed = view.n_failed_reads
        # per-read status (0 = ok); None when every read is ok
        self.status = _copy(view.read_status, view.n_reads, "<i4", copy) if view.read_status else None


class Aligner:
    """thm_aligner: mirrors the reference's per-thread aligner handle
    (ThermiteAligner, src/wrapper.rs:20-27; align_read, src/aligner.rs:123)."""

    def __init__(self, index, opts=None, device=0):
        self.index = index
        o = dict(DEFAULT_OPTS)
        o.update(opts or {})
        self._opts = self._mk(o)
        h = C.c_void_p()
        rc = lib().thm_aligner_create(index.h, C.byref(self._opts), device, C.byref(h))
        if rc != 0:
            raise ThermiteError(rc, (lib().thm_last_error(None) or b"").decode())
        self.h = h

    @staticmethod
    def _mk(o):
        return Opts(o["min_seed_len"], o["min_aln_score_percent"], o["min_aln_score"], o["multimap_score_range"],
                    int(bool(o["intron_mode"])), 0)

    def _chk(self, rc):
        if rc != 0:
            raise ThermiteError(rc, (lib().thm_last_error(self.h) or b"").decode())

    def set_opts(self, opts):
        o = dict(DEFAULT_OPTS)
        o.update(opts)
        self._opts = self._mk(o)
        self._chk(lib().thm_aligner_set_opts(self.h, C.byref(self._opts)))

    @property
    def stream(self):
        return lib().thm_aligner_stream(self.h)

    def align_batch(self, bases, offsets, copy=True):
        bases, offsets = _u8(bases), np.ascontiguousarray(offsets, "<u8")
        v = BatchView()
        self._chk(lib().thm_align_batch(self.h, _ptr(bases), _ptr(offsets), len(offsets) - 1, C.byref(v)))
        return BatchResult(v, copy)

    def upload(self, bases, offsets):
        bases, offsets = _u8(bases), np.ascontiguousarray(offsets, "<u8")
        self._chk(lib().thm_batch_upload(self.h, _ptr(bases), _ptr(offsets), len(offsets) - 1))

    def run(self):
        self._chk(lib().thm_batch_run(self.h))

    def sync(self):
        self._chk(lib().thm_batch_sync(self.h))

    def fetch(self, copy=True):
        v = BatchView()
        self._chk(lib().thm_batch_fetch(self.h, C.byref(v)))
        return BatchResult(v, copy)

    def smems_batch(self, bases, offsets, min_seed_len):
        bases, offsets = _u8(bases), np.ascontiguousarray(offsets, "<u8")
        v = MemsView()
        self._chk(lib().thm_smems_batch(self.h, _ptr(bases), _ptr(offsets), len(offsets) - 1, min_seed_len, C.byref(v)))
        return _copy(v.read_mem_off, v.n_reads + 1, "<u8"), _copy(v.mems, v.n_mems, MEM_DT)

    def swg_extend_batch(self, x_bases, x_off, y_bases, y_off, bw, xd, max_bw):
        xb, yb = _u8(x_bases), _u8(y_bases)
        xo, yo = np.ascontiguousarray(x_off, "<u8"), np.ascontiguousarray(y_off, "<u8")
        bw, xd = np.ascontiguousarray(bw, "<u4"), np.ascontiguousarray(xd, "<i4")
        v = SwgView()
        self._chk(lib().thm_swg_extend_batch(self.h, _ptr(xb), _ptr(xo), _ptr(yb), _ptr(yo), _ptr(bw), _ptr(xd),
                                             max_bw, len(bw), C.byref(v)))
        return _copy(v.alns, v.n, SWG_DT), _copy(v.ops, v.n_op_bytes, np.uint8)

    def counters(self):
        out = np.zeros(N_COUNTERS, "<u8")
        self._chk(lib().thm_counters_get(self.h, _ptr(out)))
        return out

    def reset_counters(self):
        self._chk(lib().thm_counters_reset(self.h))

    def counters_allreduce(self, comm):
        """sum the device counters over all ranks of `comm` (RCCL), in place"""
        self._chk(lib().thm_counters_allreduce(self.h, comm.h))

    def counters_device_ptr(self):
        return lib().thm_counters_device_ptr(self.h)

    def timings(self):
        out = np.zeros(N_TIMINGS, "<f4")
        self._chk(lib().thm_timings_get(self.h, _ptr(out)))
        return dict(zip(TIMING_NAMES, out.tolist()))

    def debug_prof(self, reset=True):
        out = np.zeros(16, "<u8")
        self._chk(lib().thm_debug_prof_get(self.h, _ptr(out), int(reset)))
        return out

    def debug_set_pool_caps(self, smem_cap=0, cand_cap=0, ops_cap=0):
        """test hook: initial pool sizes for the next batches; returns the replay count so far"""
        n = C.c_uint32()
        self._chk(lib().thm_debug_set_pool_caps(self.h, smem_cap, cand_cap, ops_cap, C.byref(n)))
        return n.value

    def debug_set_flags(self, tpr=None, rounds=0):
        """test / tuning hook: tpr = False: every read takes the wave-per-read kernels, True: the problem-parallel path
        in front of them (None: keep); rounds = its request rounds (1..8, 0: keep)"""
        self._chk(lib().thm_debug_set_flags(self.h, (0 if tpr is None else (2 if tpr else 1)) | (int(rounds) << 8)))

    def debug_set_band_clip(self, max_bw=None):
        """test hook: the register-resident kernels pretend to hold bands up to max_bw only (None: off)"""
        self._chk(lib().thm_debug_set_band_clip(self.h, 0 if max_bw is None else int(max_bw) + 1))

    def debug_tpr_stats(self):
        """the last run's problem-parallel path (thm_debug_tpr_stats)"""
        out = np.zeros(32, "<u8")
        self._chk(lib().thm_debug_tpr_stats(self.h, _ptr(out)))
        return out

    def debug_calib_gather(self, pattern, n_threads):
        """profiling hook: a gather of known size (see tools/calib_fetch.py); returns the bytes requested"""
        b = C.c_uint64()
        self._chk(lib().thm_debug_calib_gather(self.h, pattern, n_threads, C.byref(b)))
        return b.value

    def debug_wave_prims(self, v):
        v = np.ascontiguousarray(v, "<i4")
        out = np.zeros(384, "<i4")
        self._chk(lib().thm_debug_wave_prims(self.h, _ptr(v), _ptr(out)))
        return out.reshape(6, 64)

    def close(self):
        if getattr(self, "h", None):
            lib().thm_aligner_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def comm_unique_id():
    out = np.zeros(128, np.uint8)
    rc = lib().thm_comm_unique_id(_ptr(out))
    if rc != 0:
        raise ThermiteError(rc, _last_error())
    return out


class Comm:
    """thm_comm: RCCL communicator for the counter all-reduce (one rank per GPU)."""

    def __init__(self, unique_id, nranks, rank, device=0):
        uid = np.ascontiguousarray(unique_id, np.uint8)
        h = C.c_void_p()
        rc = lib().thm_comm_create(_ptr(uid), nranks, rank, device, C.byref(h))
        if rc != 0:
            raise ThermiteError(rc, _last_error())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().thm_comm_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def debug_deflate_block(data):
    """test hook: at most 65280 bytes through the BAM writer's own deflate -> one raw DEFLATE stream"""
    src = np.frombuffer(bytes(data), np.uint8) if len(data) else np.zeros(1, np.uint8)
    out = np.empty(len(data) + len(data) // 8 + 1024, np.uint8)
    n = C.c_uint64(0)
    rc = lib().thm_debug_deflate_block(src.ctypes.data, len(data), out.ctypes.data, len(out), C.byref(n))
    if rc != 0:
        raise ThermiteError(rc, _last_error())
    return out[: n.value].tobytes()


def debug_gunzip(path, chunk=1 << 20, cap=None, threads=1):
    """test hook: a gzip file through the library's own inflater, `chunk` bytes per call -> bytes
    (threads > 1: the chunk-parallel decoder, for files of at least four chunks of THM_INFLATE_CHUNK_KB)"""
    cap = cap if cap is not None else 64 * os.path.getsize(path) + (1 << 20)
    out = np.empty(cap, np.uint8)
    n = C.c_uint64(0)
    rc = lib().thm_debug_gunzip_mt(os.fsencode(str(path)), chunk, threads, out.ctypes.data, cap, C.byref(n))
    if rc != 0:
        raise ThermiteError(rc, _last_error())
    return out[: n.value].tobytes()


class FastqReader:
    """thm_fastq: FASTQ / FASTA batcher (needletail::parse_fastx_file, src/aligner.rs:51-56)."""

    def __init__(self, path):
        h = C.c_void_p()
        rc = lib().thm_fastq_open(str(path).encode(), C.byref(h))
        if rc != 0:
            raise ThermiteError(rc, _last_error())
        self.h = h

    def all_by_blocks(self, reads_per_block):
        """test hook: the whole file through the parallel driver's block cutter + block parser"""
        v = ReadBatch()
        rc = lib().thm_debug_fastq_blocks(self.h, reads_per_block, C.byref(v))
        if rc != 0:
            raise ThermiteError(rc, _last_error())
        off = _copy(v.offsets, v.n_reads + 1, "<u8")
        noff = _copy(v.name_off, v.n_reads + 1, "<u8")
        return dict(bases=_copy(v.bases, v.n_bases, np.uint8), offsets=off, quals=_copy(v.quals, v.n_bases, np.uint8),
                    names=_copy(v.names, int(noff[-1]), np.uint8), name_off=noff)

    def next_batch(self, max_reads):
        """-> dict(bases, offsets, quals (None for FASTA), names, name_off) or None at end of file"""
        v = ReadBatch()
        rc = lib().thm_fastq_next_batch(self.h, max_reads, C.byref(v))
        if rc != 0:
            raise ThermiteError(rc, _last_error())
        if v.n_reads == 0:
            return None
        off = _copy(v.offsets, v.n_reads + 1, "<u8")
        noff = _copy(v.name_off, v.n_reads + 1, "<u8")
        return dict(bases=_copy(v.bases, v.n_bases, np.uint8), offsets=off,
                    quals=_copy(v.quals, v.n_bases, np.uint8) if v.quals else None,
                    names=_copy(v.names, int(noff[-1]), np.uint8), name_off=noff)

    def close(self):
        if getattr(self, "h", None):
            lib().thm_fastq_close(self.h)
            self.h = None

    def __del__(self):
        self.close()


def read_batch_struct(batch):
    """dict from FastqReader.next_batch (or the same keys) -> (ReadBatch, keep-alive list)"""
    keep = [_u8(batch["bases"]), np.ascontiguousarray(batch["offsets"], "<u8"),
            None if batch.get("quals") is None else _u8(batch["quals"]), _u8(batch["names"]),
            np.ascontiguousarray(batch["name_off"], "<u8")]
    rb = ReadBatch(len(keep[1]) - 1, len(keep[0]), _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), _ptr(keep[3]), _ptr(keep[4]))
    return rb, keep


class Writer:
    """thm_writer: SAM / PAF rendering (src/aln_writer.rs)."""

    def __init__(self, index, fmt=FMT_SAM, n_threads=1):
        h = C.c_void_p()
        rc = lib().thm_writer_create(index.h, fmt, n_threads, C.byref(h))
        if rc != 0:
            raise ThermiteError(rc, _last_error())
        self.h = h
        self.index = index

    def header(self):
        t = Text()
        rc = lib().thm_writer_header(self.h, C.byref(t))
        if rc != 0:
            raise ThermiteError(rc, _last_error())
        return bytes(_copy(t.data, t.len, np.uint8))

    def trailer(self):
        t = Text()
        rc = lib().thm_writer_trailer(self.h, C.byref(t))
        if rc != 0:
            raise ThermiteError(rc, _last_error())
        return bytes(_copy(t.data, t.len, np.uint8))

    def format_batch(self, batch, result):
        """batch: dict as from FastqReader.next_batch; result: BatchResult -> bytes"""
        rb, keep = read_batch_struct(batch)
        offs = np.ascontiguousarray(result.offsets, "<u8")
        v = BatchView(len(offs) - 1, len(result.alns), len(result.ops), _ptr(offs), _ptr(result.alns),
                      _ptr(result.ops), 0, None)
        t = Text()
        rc = lib().thm_writer_format_batch(self.h, C.byref(rb), C.byref(v), C.byref(t))
        if rc != 0:
            raise ThermiteError(rc, _last_error())
        return bytes(_copy(t.data, t.len, np.uint8))

    def close(self):
        if getattr(self, "h", None):
            lib().thm_writer_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def align_files(aligner, fastq_paths, output_path, fmt=FMT_SAM, batch_reads=0, n_threads=0):
    """align_reads_from_file (src/aligner.rs:22-120): FASTQ files -> one SAM / PAF / BAM file; returns the run stats.
    `aligner`: one Aligner, or a list of them (one per GPU, all over the same Index)."""
    paths = _cstr_array([str(p) for p in fastq_paths])
    st = RunStats()
    als = list(aligner) if isinstance(aligner, (list, tuple)) else [aligner]
    arr = (C.c_void_p * len(als))(*[a.h for a in als])
    rc = lib().thm_align_files_multi(arr, len(als), paths, len(fastq_paths), str(output_path).encode(), fmt, batch_reads, n_threads,
                                     C.byref(st))
    if rc != 0:
        raise ThermiteError(rc, _last_error())
    return {k: getattr(st, k) for k, _ in RunStats._fields_}
