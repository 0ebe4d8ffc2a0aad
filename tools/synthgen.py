"""ctypes face of tools/libsynthgen.so: the synthetic workload of SURVEY.md §8(d) generated on all host threads
(measurement tool; the tests and golden fixtures keep using fade_amd/synth.py, whose stream this one does not reproduce)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class SgCfg(C.Structure):
    _fields_ = [("n_contigs", C.c_int32), ("contig_len", C.c_int64), ("read_len", C.c_int32), ("window", C.c_int32),
                ("p_sc", C.c_double), ("clip_min", C.c_int32), ("clip_max", C.c_int32), ("insert_mu", C.c_double),
                ("insert_sd", C.c_double), ("p_unmapped", C.c_double), ("p_sa", C.c_double), ("p_sub", C.c_double),
                ("p_planted", C.c_double)]


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "libsynthgen.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", HERE, "-s", "libsynthgen.so"])
        L = C.CDLL(path)
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
        L.sg_genome.argtypes = [C.c_uint64, i64, vp, C.c_int]
        L.sg_genome.restype = None
        L.sg_reads.argtypes = [vp, C.POINTER(SgCfg), i64, C.c_uint64, i64] + [vp] * 10 + [C.c_int]
        L.sg_reads.restype = i64
        L.sg_compact.argtypes = [i64] + [vp] * 7 + [C.POINTER(i32), C.POINTER(i32)]
        L.sg_compact.restype = i64
        L.sg_write_fasta.argtypes = [C.c_char_p, vp, C.c_int, i64]
        L.sg_bam_open.argtypes = [C.c_char_p, C.c_int, i64, C.c_int]
        L.sg_bam_open.restype = vp
        L.sg_bam_write.argtypes = [vp, i64, i64] + [vp] * 10
        L.sg_bam_close.argtypes = [vp]
        _lib = L
    return _lib


def threads():
    return max(1, min(len(os.sched_getaffinity(0)), 32))


class Genome:
    """Same interface as fade_amd.synth.Genome (names, lengths, offsets, codes, ascii_contigs)."""

    def __init__(self, n_contigs, contig_len, seed, kind="uniform"):
        self.names = ["chr%d" % (k + 1) for k in range(n_contigs)]
        self.lengths = np.full(n_contigs, contig_len, dtype=np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.lengths)]).astype(np.int64)
        self.codes = np.empty(int(self.offsets[-1]), dtype=np.uint8)
        lib().sg_genome(seed, len(self.codes), self.codes.ctypes.data, threads())
        if kind == "repeat_rich":  # (the repeat structure is laid over the uniform bases by fade_amd.synth's routine)
            from fade_amd import synth
            synth.make_repeat_rich(self.codes, n_contigs, contig_len, seed)

    def ascii_contigs(self):
        t = np.frombuffer(b"ACGT", dtype=np.uint8)
        return [t[self.codes[self.offsets[k]:self.offsets[k + 1]]] for k in range(len(self.names))]

    def write_fasta(self, path):
        assert lib().sg_write_fasta(path.encode(), self.codes.ctypes.data, len(self.names), int(self.lengths[0])) == 0


def _cfg(genome, cfg):
    return SgCfg(len(genome.names), int(genome.lengths[0]), cfg["read_len"], cfg["window"], cfg["p_sc"], cfg["clip_min"],
                 cfg["clip_max"], cfg["insert_mu"], cfg["insert_sd"], cfg.get("p_unmapped", 0.01), cfg.get("p_sa", 0.02),
                 cfg.get("p_sub", 0.001), cfg.get("p_planted", 0.5))


def make_reads(genome, n, seed, cfg, idx0=0):
    """The batch dict of fade_amd.synth.make_reads (without qname / _truth)."""
    L = cfg["read_len"]
    nb = (L + 1) // 2
    b = dict(tid=np.empty(n, np.int32), pos=np.empty(n, np.int32), flag=np.empty(n, np.uint16), has_sa=np.empty(n, np.uint8),
             l_seq=np.empty(n, np.int32), cigar_off=np.empty(n + 1, np.uint32), cigar_ops=np.empty(3 * n, np.uint32),
             seq_off=np.empty(n + 1, np.uint32), seq_packed=np.empty(n * nb, np.uint8), qual=np.empty(n * L, np.uint8))
    c = _cfg(genome, cfg)
    n_ops = lib().sg_reads(genome.codes.ctypes.data, C.byref(c), n, seed, idx0, *[b[k].ctypes.data for k in (
        "tid", "pos", "flag", "has_sa", "l_seq", "cigar_off", "cigar_ops", "seq_off", "seq_packed", "qual")], threads())
    b["cigar_ops"] = b["cigar_ops"][:n_ops]
    b["qual_off"] = np.arange(n + 1, dtype=np.int64) * L
    return b


def with_bounds(b):
    """fade_amd.api.Context.with_bounds, natively: every record kept, bases only where the device can align, plus the
    ABI-3 bounds (n_with_seq, l_seq_min / l_seq_max, ref_span_bound)."""
    n = len(b["pos"])
    so = np.empty(n + 1, np.uint32)
    sq = np.empty(len(b["seq_packed"]), np.uint8)
    cnt, span = C.c_int32(0), C.c_int32(0)
    kept = lib().sg_compact(n, b["flag"].ctypes.data, b["cigar_off"].ctypes.data, b["cigar_ops"].ctypes.data, b["seq_off"].ctypes.data,
                            b["seq_packed"].ctypes.data, so.ctypes.data, sq.ctypes.data, C.byref(cnt), C.byref(span))
    out = dict(b)
    out["seq_off"], out["seq_packed"] = so, sq[:kept]
    ls = b["l_seq"]
    out.update(n_with_seq=int(cnt.value), l_seq_min=int(ls.min()), l_seq_max=int(ls.max()), ref_span_bound=int(span.value), n_skipped=0)
    return out


class BamWriter:
    def __init__(self, path, genome):
        self.h = lib().sg_bam_open(path.encode(), len(genome.names), int(genome.lengths[0]), threads())
        assert self.h, "cannot open " + path

    def write(self, b, name_base):
        n = len(b["pos"])
        rc = lib().sg_bam_write(self.h, n, name_base, *[b[k].ctypes.data for k in (
            "tid", "pos", "flag", "has_sa", "l_seq", "cigar_off", "cigar_ops", "seq_off", "seq_packed", "qual")])
        assert rc == 0

    def close(self):
        assert lib().sg_bam_close(self.h) == 0
        self.h = None
