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
                ("read_aln_off", C.c_void_p), ("alns", C.c_void_p), ("ops", C.c_void_p)]


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
]

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
    if hasattr(L, "thm_debug_prof_get"):
        L.thm_debug_prof_get.restype = i32
        L.thm_debug_prof_get.argtypes = [vp, vp, i32]
    if hasattr(L, "thm_debug_wave_prims"):
        L.thm_debug_wave_prims.restype = i32
        L.thm_debug_wave_prims.argtypes = [vp, vp, vp]
    _lib = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _copy(ptr, count, dtype):
    if count == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    nbytes = int(count) * np.dtype(dtype).itemsize
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=int(count)).copy()


def _u8(b):
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, np.uint8)
    return np.frombuffer(bytes(b), np.uint8)


def build_suffix_array(text):
    t = _u8(text)
    sa = np.zeros(len(t), "<u4")
    rc = lib().thm_build_suffix_array(_ptr(t), len(t), _ptr(sa))
    if rc != 0:
        raise ThermiteError(rc, "thm_build_suffix_array")
    return sa


class Index:
    """thm_index: the in-memory index (stands for reference Index, src/index.rs:39-44)."""

    def __init__(self, tables, sa=None):
        t = tables
        self.tables = t
        h = C.c_void_p()
        sa_arr = None if sa is None else np.ascontiguousarray(sa, "<u4")
        rc = lib().thm_index_create_in_memory(
            _ptr(t["text"]), len(t["text"]), _ptr(t["refs"]), len(t["refs"]), _ptr(t["txs"]), len(t["txs"]),
            _ptr(t["exons"]), len(t["exons"]), _ptr(t["tx_seq"]), len(t["tx_seq"]), _ptr(t["genes"]), len(t["genes"]),
            _ptr(t["name_rank"]), len(t["name_rank"]), _ptr(sa_arr), C.byref(h),
        )
        if rc != 0:
            raise ThermiteError(rc, (lib().thm_last_error(None) or b"").decode())
        self.h = h

    def suffix_array(self):
        n = lib().thm_index_text_len(self.h)
        return _copy(lib().thm_index_suffix_array(self.h), n, "<u4")

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
    """Canonical host result of one aligned batch (thm_batch_view copied out)."""

    def __init__(self, view):
        self.n_reads = view.n_reads
        self.offsets = _copy(view.read_aln_off, view.n_reads + 1, "<u8")
        self.alns = _copy(view.alns, view.n_alns, ALN_DT)
        self.ops = _copy(view.ops, view.n_op_bytes, np.uint8)


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

    def align_batch(self, bases, offsets):
        bases, offsets = _u8(bases), np.ascontiguousarray(offsets, "<u8")
        v = BatchView()
        self._chk(lib().thm_align_batch(self.h, _ptr(bases), _ptr(offsets), len(offsets) - 1, C.byref(v)))
        return BatchResult(v)

    def upload(self, bases, offsets):
        bases, offsets = _u8(bases), np.ascontiguousarray(offsets, "<u8")
        self._chk(lib().thm_batch_upload(self.h, _ptr(bases), _ptr(offsets), len(offsets) - 1))

    def run(self):
        self._chk(lib().thm_batch_run(self.h))

    def sync(self):
        self._chk(lib().thm_batch_sync(self.h))

    def fetch(self):
        v = BatchView()
        self._chk(lib().thm_batch_fetch(self.h, C.byref(v)))
        return BatchResult(v)

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
