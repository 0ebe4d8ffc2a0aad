"""ctypes binding of oracle/libfadeoracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/fade_oracle.h).  Nothing under fade_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfadeoracle.so")

CIGAR_OPS = "MIDNSHP=X"


def build(force=False):
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class Params(C.Structure):
    _fields_ = [("open", C.c_int), ("ext", C.c_int), ("match", C.c_int), ("mismatch", C.c_int),
                ("alphabet", C.c_char_p), ("rules", C.c_uint32), ("striped", C.c_int)]


class SwResult(C.Structure):
    _fields_ = [("score", C.c_int), ("end_query", C.c_int), ("end_ref", C.c_int),
                ("beg_query", C.c_int), ("beg_ref", C.c_int), ("n_ops", C.c_int)]


class Genome(C.Structure):
    _fields_ = [("n_contigs", C.c_int), ("names", C.POINTER(C.c_char_p)),
                ("lengths", C.POINTER(C.c_int64)), ("seqs", C.POINTER(C.c_char_p))]


class Read(C.Structure):
    _fields_ = [("qname", C.c_char_p), ("flag", C.c_uint16), ("tid", C.c_int32), ("pos", C.c_int64),
                ("n_cigar", C.c_int), ("cigar", C.POINTER(C.c_uint32)), ("l_seq", C.c_int),
                ("seq4", C.POINTER(C.c_uint8)), ("qual", C.POINTER(C.c_uint8)), ("has_sa", C.c_int)]


class Anno(C.Structure):
    _fields_ = [("rs", C.c_uint8), ("has_tags", C.c_int), ("am", C.c_void_p), ("as_", C.c_void_p),
                ("ar", C.c_void_p), ("ab", C.c_void_p), ("n_sw_calls", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.fo_params_default.argtypes = [C.POINTER(Params)]
        L.fo_sw_trace.argtypes = [C.POINTER(Params), C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                  C.POINTER(SwResult), C.POINTER(C.c_uint32), C.c_int]
        L.fo_sw_trace.restype = C.c_int
        L.fo_sw_striped.argtypes = L.fo_sw_trace.argtypes
        L.fo_sw_striped.restype = C.c_int
        L.fo_sw_trace_table.argtypes = L.fo_sw_trace.argtypes + [C.c_void_p]
        L.fo_sw_trace_table.restype = C.c_int
        L.fo_annotate_task.argtypes = [C.POINTER(Params), C.POINTER(Genome), C.POINTER(Read), C.c_int,
                                       C.c_int, C.POINTER(Anno)]
        L.fo_annotate_task.restype = C.c_int
        L.fo_anno_free.argtypes = [C.POINTER(Anno)]
        L.fo_annotate_batch.argtypes = [C.POINTER(Params), C.POINTER(Genome), C.POINTER(Read), C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fo_annotate_batch.restype = C.c_int
        L.fo_annotate_batch_soa.argtypes = [C.POINTER(Params), C.POINTER(Genome), C.c_int] + [C.c_void_p] * 11 + \
            [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fo_annotate_batch_soa.restype = C.c_int
        L.fo_reverse_complement_packed.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
        L.fo_parse_clips.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32)]
        L.fo_stats_parse.argtypes = [C.c_uint8, C.POINTER(C.c_int64)]
        L.fo_sw_batch.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.fo_sw_batch.restype = C.c_int
        _lib = L
    return _lib


def default_params(rules=None, striped=False):
    p = Params()
    lib().fo_params_default(C.byref(p))
    if rules is not None:
        p.rules = rules
    p.striped = 1 if striped else 0
    return p


def cigar_str(ops):
    return "".join("%d%s" % (int(o) >> 4, CIGAR_OPS[int(o) & 0xF]) for o in ops)


def sw(q, r, params=None, striped=False, ops_cap=None):
    """One alignment.  q, r: bytes/str.  Returns dict(score,end_query,end_ref,beg_query,beg_ref,ops)."""
    L = lib()
    p = params or default_params()
    if isinstance(q, str):
        q = q.encode()
    if isinstance(r, str):
        r = r.encode()
    cap = ops_cap or (len(q) + len(r) + 4)
    ops = (C.c_uint32 * cap)()
    res = SwResult()
    fn = L.fo_sw_striped if striped else L.fo_sw_trace
    rc = fn(C.byref(p), q, len(q), r, len(r), C.byref(res), ops, cap)
    if rc != 0:
        raise RuntimeError("oracle sw failed rc=%d" % rc)
    n = min(res.n_ops, cap)
    return dict(score=res.score, end_query=res.end_query, end_ref=res.end_ref, beg_query=res.beg_query,
                beg_ref=res.beg_ref, n_ops=res.n_ops, ops=[int(ops[k]) for k in range(n)])


def sw_batch(q_concat, q_off, r_concat, r_off, threads=1, striped=False, max_ops=16, params=None):
    """Batch of alignments over concatenated ASCII buffers (numpy uint8) with int64 offset arrays
    (n+1 entries).  Returns (res[n,6] int32: score,end_q,end_r,beg_q,beg_r,n_ops ; ops[n,max_ops] uint32)."""
    L = lib()
    p = params or default_params()
    n = len(q_off) - 1
    q_concat = np.ascontiguousarray(q_concat, dtype=np.uint8)
    r_concat = np.ascontiguousarray(r_concat, dtype=np.uint8)
    q_off = np.ascontiguousarray(q_off, dtype=np.int64)
    r_off = np.ascontiguousarray(r_off, dtype=np.int64)
    res = np.zeros((n, 6), dtype=np.int32)
    ops = np.zeros((n, max_ops), dtype=np.uint32)
    rc = L.fo_sw_batch(C.byref(p), n, threads, q_concat.ctypes.data, q_off.ctypes.data, r_concat.ctypes.data,
                       r_off.ctypes.data, res.ctypes.data, ops.ctypes.data, max_ops,
                       1 if striped else 0)
    if rc != 0:
        raise RuntimeError("oracle sw_batch failed rc=%d" % rc)
    return res, ops


class GenomeHolder:
    """Keeps Python-side buffers alive for a fo_genome."""

    def __init__(self, names, seqs):
        self.names = [n.encode() if isinstance(n, str) else n for n in names]
        self.seqs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        n = len(names)
        self._names = (C.c_char_p * n)(*self.names)
        self._lens = (C.c_int64 * n)(*[len(s) for s in self.seqs])
        self._seqs = (C.c_char_p * n)(*self.seqs)
        self.c = Genome(n, self._names, self._lens, self._seqs)


def make_reads(batch):
    """batch: dict of numpy arrays in the layout of fade_amd.synth.ReadBatch.  Returns (Read array, keepalive)."""
    n = len(batch["pos"])
    arr = (Read * n)()
    cig = np.ascontiguousarray(batch["cigar_ops"], dtype=np.uint32)
    seq = np.ascontiguousarray(batch["seq_packed"], dtype=np.uint8)
    qual = np.ascontiguousarray(batch["qual"], dtype=np.uint8)
    cig_p = cig.ctypes.data
    seq_p = seq.ctypes.data
    qual_p = qual.ctypes.data
    coff = batch["cigar_off"]
    soff = batch["seq_off"]
    qoff = batch["qual_off"]
    names = batch.get("qname")
    keep = [cig, seq, qual]
    for i in range(n):
        r = arr[i]
        r.qname = names[i] if names is not None else b"r"
        r.flag = int(batch["flag"][i])
        r.tid = int(batch["tid"][i])
        r.pos = int(batch["pos"][i])
        r.n_cigar = int(coff[i + 1] - coff[i])
        r.cigar = C.cast(cig_p + 4 * int(coff[i]), C.POINTER(C.c_uint32))
        r.l_seq = int(batch["l_seq"][i])
        r.seq4 = C.cast(seq_p + int(soff[i]), C.POINTER(C.c_uint8))
        r.qual = C.cast(qual_p + int(qoff[i]), C.POINTER(C.c_uint8))
        r.has_sa = int(batch["has_sa"][i])
    return arr, keep


def _take_str(ptr):
    return C.string_at(ptr).decode() if ptr else None


def annotate_one(genome, read, floor_len=5, window=300, params=None):
    L = lib()
    p = params or default_params()
    a = Anno()
    L.fo_annotate_task(C.byref(p), C.byref(genome.c), C.byref(read), floor_len, window, C.byref(a))
    out = dict(rs=int(a.rs), has_tags=int(a.has_tags), am=_take_str(a.am), as_=_take_str(a.as_),
               ar=_take_str(a.ar), ab=_take_str(a.ab), n_sw_calls=a.n_sw_calls)
    L.fo_anno_free(C.byref(a))
    return out


def annotate_batch(genome, reads, n, floor_len=5, window=300, threads=1, want_am=True, params=None):
    """Returns (rs uint8[n], am list[str|None])."""
    L = lib()
    p = params or default_params()
    rs = np.zeros(n, dtype=np.uint8)
    am_ptrs = (C.c_void_p * n)() if want_am else None
    L.fo_annotate_batch(C.byref(p), C.byref(genome.c), reads, n, floor_len, window, threads, rs.ctypes.data,
                        C.cast(am_ptrs, C.c_void_p) if want_am else None)
    am = None
    if want_am:
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        am = []
        for i in range(n):
            if am_ptrs[i]:
                am.append(C.string_at(am_ptrs[i]).decode())
                libc.free(am_ptrs[i])
            else:
                am.append(None)
    return rs, am


def annotate_batch_soa(genome, batch, floor_len=5, window=300, threads=1, want_am=True, params=None):
    """annotateTask over a whole SoA batch (dict of numpy arrays).  Returns (rs uint8[n], am list|None)."""
    L = lib()
    p = params or default_params()
    n = len(batch["pos"])
    a = {k: np.ascontiguousarray(batch[k], dtype=dt) for k, dt in
         (("tid", np.int32), ("pos", np.int32), ("flag", np.uint16), ("has_sa", np.uint8), ("l_seq", np.int32),
          ("cigar_off", np.uint32), ("cigar_ops", np.uint32), ("seq_off", np.uint32), ("seq_packed", np.uint8),
          ("qual_off", np.int64), ("qual", np.uint8))}
    rs = np.zeros(n, dtype=np.uint8)
    am_ptrs = (C.c_void_p * n)() if want_am else None
    rc = L.fo_annotate_batch_soa(C.byref(p), C.byref(genome.c), n, *[a[k].ctypes.data for k in
                                 ("tid", "pos", "flag", "has_sa", "l_seq", "cigar_off", "cigar_ops", "seq_off",
                                  "seq_packed", "qual_off", "qual")], floor_len, window, threads, rs.ctypes.data,
                                 C.cast(am_ptrs, C.c_void_p) if want_am else None)
    if rc != 0:
        raise RuntimeError("oracle annotate_batch_soa failed rc=%d" % rc)
    am = None
    if want_am:
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        am = []
        for i in range(n):
            if am_ptrs[i]:
                am.append(C.string_at(am_ptrs[i]).decode())
                libc.free(am_ptrs[i])
            else:
                am.append(None)
    return rs, am


def reverse_complement_packed(seq4, l_seq):
    buf = C.create_string_buffer(l_seq)
    s = np.ascontiguousarray(seq4, dtype=np.uint8)
    lib().fo_reverse_complement_packed(s.ctypes.data, l_seq, buf)
    return buf.raw


def parse_clips(cigar):
    arr = (C.c_uint32 * len(cigar))(*cigar)
    out = (C.c_uint32 * 2)()
    lib().fo_parse_clips(arr, len(cigar), out)
    return int(out[0]), int(out[1])
