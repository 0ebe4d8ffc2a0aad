"""ctypes binding of fade_amd/libfadehip.so (include/fadehip.h).

The library is the product path; there is no fallback.  Loading fails loudly when the shared
object is missing, and fadehip_create fails when no gfx950 device is present.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FADEHIP_LIB") or os.path.join(HERE, "libfadehip.so")  # FADEHIP_LIB: A/B builds of the same ABI

MAX_OPS = 16
NUM_SLOTS = 4
ABI_VERSION = 3
RULES_DEFAULT = 0x7f
REF_CONSUMING_OPS = (0, 2, 3, 7, 8)  # include/fadehip.h FADEHIP_REF_CONSUMING_OPS (dhtslib Cigar.alignedLength: M, D, N, =, X)
ROW_CLASSES = (4, 6, 8, 10, 12, 14, 16, 20, 24, 32)  # query rows per lane of the wave kernels (fadehip_kernels.hpp class_rows)


def row_class(l_seq):
    """R of the sw_pk_kernel<R, .> instantiation that serves reads of l_seq bases (16 R >= l_seq)."""
    return next(r for r in ROW_CLASSES if 16 * r >= l_seq)

# every symbol include/fadehip.h declares
EXPORTS = [
    "fadehip_params_default", "fadehip_abi_version", "fadehip_create", "fadehip_destroy", "fadehip_last_error",
    "fadehip_host_alloc", "fadehip_host_free", "fadehip_host_register", "fadehip_batch_bytes", "fadehip_batch_bind", "fadehip_sw_batch",
    "fadehip_genome_upload", "fadehip_annotate_upload", "fadehip_annotate_run", "fadehip_annotate_submit",
    "fadehip_annotate_results", "fadehip_annotate_collect",
    "fadehip_sync", "fadehip_last_run_profile", "fadehip_stats_allreduce",
    "fadehip_bgzf_deflate_submit", "fadehip_bgzf_deflate_wait", "fadehip_stats_allreduce_rank", "fadehip_bgzf_inflate",
    "fadehip_bam_open", "fadehip_bam_front", "fadehip_bam_front_raw", "fadehip_bam_back", "fadehip_bam_totals", "fadehip_bam_close",
    "fadehip_bam_prepare",
]
BGZF_BLOCK = 0xff00
BGZF_LANES = 2


class Params(C.Structure):
    _fields_ = [("open", C.c_int32), ("ext", C.c_int32), ("match", C.c_int32), ("mismatch", C.c_int32),
                ("max_ref_len", C.c_int32), ("max_batch_reads", C.c_int32), ("trace_bytes", C.c_int64),
                ("trace_all", C.c_int32), ("rules", C.c_uint32)]


class SwResult(C.Structure):
    _fields_ = [("score", C.c_int32), ("end_query", C.c_int32), ("end_ref", C.c_int32), ("beg_query", C.c_int32),
                ("beg_ref", C.c_int32), ("n_ops", C.c_int32), ("ops", C.c_uint32 * MAX_OPS)]


class Aln(C.Structure):
    _fields_ = [("read_idx", C.c_int32), ("art", C.c_int32), ("win_start", C.c_int64), ("win_len", C.c_int32),
                ("clip_left", C.c_int32), ("clip_right", C.c_int32), ("aligned_len", C.c_int32), ("sw", SwResult)]


SW_DTYPE = np.dtype([("score", "<i4"), ("end_query", "<i4"), ("end_ref", "<i4"), ("beg_query", "<i4"),
                     ("beg_ref", "<i4"), ("n_ops", "<i4"), ("ops", "<u4", (MAX_OPS,))])
ALN_DTYPE = np.dtype([("read_idx", "<i4"), ("art", "<i4"), ("win_start", "<i8"), ("win_len", "<i4"),
                      ("clip_left", "<i4"), ("clip_right", "<i4"), ("aligned_len", "<i4"), ("sw", SW_DTYPE)])
assert SW_DTYPE.itemsize == C.sizeof(SwResult)
assert ALN_DTYPE.itemsize == C.sizeof(Aln)


class ReadBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("tid", C.c_void_p), ("pos", C.c_void_p), ("flag", C.c_void_p),
                ("has_sa", C.c_void_p), ("l_seq", C.c_void_p), ("cigar_off", C.c_void_p), ("cigar_ops", C.c_void_p),
                ("seq_off", C.c_void_p), ("seq_packed", C.c_void_p), ("n_skipped", C.c_int32),
                ("ref_span_bound", C.c_int32), ("n_with_seq", C.c_int32), ("l_seq_min", C.c_int32),
                ("l_seq_max", C.c_int32), ("reserved", C.c_int32)]


class AnnoOut(C.Structure):
    _fields_ = [("rs", C.c_void_p), ("aln", C.c_void_p), ("aln_cap", C.c_int32), ("n_aln", C.c_int32),
                ("stats", C.c_int64 * 8), ("n_oversize", C.c_int32), ("reserved", C.c_int32)]


class AnnoView(C.Structure):
    _fields_ = [("rs", C.c_void_p), ("aln", C.c_void_p), ("n_reads", C.c_int32), ("n_aln", C.c_int32),
                ("stats", C.c_int64 * 8), ("n_oversize", C.c_int32), ("reserved", C.c_int32)]


class BamConfig(C.Structure):
    _fields_ = [("floor_len", C.c_int32), ("window", C.c_int32), ("n_ref", C.c_int32), ("flags", C.c_int32),
                ("ref_names", C.POINTER(C.c_char_p)), ("first_record", C.c_uint32), ("tail_trim", C.c_uint32)]


class FadeHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fadehip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load():
    """Load libfadehip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); fade_amd has no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.fadehip_params_default.argtypes = [C.POINTER(Params)]
    L.fadehip_params_default.restype = None
    L.fadehip_abi_version.restype = C.c_int
    L.fadehip_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(Params)]
    L.fadehip_destroy.argtypes = [vp]
    L.fadehip_destroy.restype = None
    L.fadehip_last_error.argtypes = [vp]
    L.fadehip_last_error.restype = C.c_char_p
    L.fadehip_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.fadehip_host_free.argtypes = [vp, vp]
    L.fadehip_host_register.argtypes = [vp, vp, C.c_size_t]
    L.fadehip_batch_bytes.argtypes = [i32, i64, i64]
    L.fadehip_batch_bytes.restype = C.c_size_t
    L.fadehip_batch_bind.argtypes = [vp, i32, i64, i64, C.POINTER(ReadBatch)]
    L.fadehip_annotate_results.argtypes = [vp, C.c_int, C.POINTER(AnnoView)]
    L.fadehip_sw_batch.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.fadehip_genome_upload.argtypes = [vp, i32, vp, vp]
    L.fadehip_annotate_upload.argtypes = [vp, C.c_int, C.POINTER(ReadBatch)]
    L.fadehip_annotate_run.argtypes = [vp, C.c_int, i32, i32]
    L.fadehip_annotate_submit.argtypes = [vp, C.c_int, C.POINTER(ReadBatch), i32, i32]
    L.fadehip_annotate_collect.argtypes = [vp, C.c_int, C.POINTER(AnnoOut)]
    L.fadehip_sync.argtypes = [vp]
    L.fadehip_last_run_profile.argtypes = [vp, C.c_int, C.POINTER(C.c_float * 4), C.POINTER(i64 * 6)]
    L.fadehip_stats_allreduce.argtypes = [C.POINTER(vp), C.c_int, vp, C.c_int]
    L.fadehip_bgzf_deflate_submit.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.fadehip_stats_allreduce_rank.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, vp, C.c_int]
    L.fadehip_bgzf_deflate_wait.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.fadehip_bgzf_inflate.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.fadehip_bam_open.argtypes = [vp, C.POINTER(BamConfig), C.POINTER(vp)]
    L.fadehip_bam_prepare.argtypes = [vp, C.c_size_t]
    L.fadehip_bam_front.argtypes = [vp, vp, C.c_size_t, C.c_int]
    L.fadehip_bam_front_raw.argtypes = [vp, vp, C.c_size_t, C.c_int]
    L.fadehip_bam_back.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.fadehip_bam_totals.argtypes = [vp, C.POINTER(i64 * 8), C.POINTER(i64), C.POINTER(i64)]
    L.fadehip_bam_close.argtypes = [vp]
    L.fadehip_bam_close.restype = None
    for name in EXPORTS:
        getattr(L, name)  # AttributeError here means the .so is stale against include/fadehip.h
    _lib = L
    return L
