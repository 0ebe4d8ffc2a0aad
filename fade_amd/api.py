"""Host-side mirror of the reference interface for the annotate hot path, over the C ABI.

Reference seam (D):
    auto p = Parasail("ACTGN", 10, 2, 2, -3);           source/anno.d:36
    auto res = p.sw_striped(q_seq, ref_seq);            source/analysis.d:67
    res.score / res.position / res.cigar                source/analysis.d:69-113
    annotateTask(rec, &p, fai, mfai, floor, window)     source/anno.d:55-110

Everything that computes runs in libfadehip.so on the GPU; this module only marshals numpy
arrays and formats the am/as/ar/ab strings exactly as analysis.d:84-92,108-118 and
anno.d:94-107 do.  There is no CPU implementation of the alignment here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ALN_DTYPE, SW_DTYPE, FadeHipError

CIGAR_OPS = "MIDNSHP=X"
NT16 = "=ACMGRSVTWYHKDBN"
_NT16_ARR = np.frombuffer(NT16.encode(), dtype=np.uint8)
# util.d:18-20
_COMP = np.array([0, 8, 4, 12, 2, 10, 6, 14, 1, 9, 5, 13, 3, 11, 7, 15], dtype=np.uint8)


def cigar_str(ops):
    return "".join("%d%s" % (int(o) >> 4, CIGAR_OPS[int(o) & 0xF]) for o in ops)


class Context:
    """One fadehip_ctx (one GPU)."""

    def __init__(self, device=-1, open=10, ext=2, match=2, mismatch=-3, max_ref_len=0, max_batch_reads=0,
                 trace_bytes=0, trace_all=False, rules=None):
        self._L = _lib.load()
        p = _lib.Params()
        self._L.fadehip_params_default(C.byref(p))
        p.open, p.ext, p.match, p.mismatch = open, ext, match, mismatch
        if max_ref_len:
            p.max_ref_len = max_ref_len
        if max_batch_reads:
            p.max_batch_reads = max_batch_reads
        p.trace_bytes = trace_bytes
        p.trace_all = 1 if trace_all else 0
        if rules is not None:
            p.rules = rules
        h = C.c_void_p()
        rc = self._L.fadehip_create(C.byref(h), device, C.byref(p))
        if rc != 0:
            raise FadeHipError(rc, self._L.fadehip_last_error(None).decode())
        self._h = h
        self._keep = {}
        self._views = {}
        self.contig_names = None

    def close(self):
        if getattr(self, "_h", None):
            for ptr in getattr(self, "_pinned", []):
                self._L.fadehip_host_free(self._h, ptr)
            self._pinned = []
            self._L.fadehip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise FadeHipError(rc, self._L.fadehip_last_error(self._h).decode())

    # ---- level 1: the SW seam (analysis.d:67)
    def sw_batch_packed(self, q_concat, q_off, r_concat, r_off):
        q_concat = np.ascontiguousarray(q_concat, dtype=np.uint8)
        r_concat = np.ascontiguousarray(r_concat, dtype=np.uint8)
        q_off = np.ascontiguousarray(q_off, dtype=np.int64)
        r_off = np.ascontiguousarray(r_off, dtype=np.int64)
        n = len(q_off) - 1
        out = np.zeros(n, dtype=SW_DTYPE)
        self._chk(self._L.fadehip_sw_batch(self._h, n, q_concat.ctypes.data, q_off.ctypes.data, r_concat.ctypes.data,
                                           r_off.ctypes.data, out.ctypes.data))
        return out

    def sw_batch(self, queries, refs):
        qs = [q.encode() if isinstance(q, str) else bytes(q) for q in queries]
        rs = [r.encode() if isinstance(r, str) else bytes(r) for r in refs]
        q_off = np.zeros(len(qs) + 1, dtype=np.int64)
        r_off = np.zeros(len(rs) + 1, dtype=np.int64)
        np.cumsum([len(q) for q in qs], out=q_off[1:])
        np.cumsum([len(r) for r in rs], out=r_off[1:])
        qc = np.frombuffer(b"".join(qs), dtype=np.uint8) if q_off[-1] else np.zeros(0, np.uint8)
        rc = np.frombuffer(b"".join(rs), dtype=np.uint8) if r_off[-1] else np.zeros(0, np.uint8)
        return self.sw_batch_packed(qc, q_off, rc, r_off)

    # ---- level 2: annotateTask over a batch (anno.d:55-110)
    def genome_upload(self, names, seqs):
        """seqs: list of bytes / uint8 arrays (raw FASTA residues)."""
        arrs = [np.frombuffer(s.encode() if isinstance(s, str) else s, dtype=np.uint8) if not isinstance(s, np.ndarray)
                else np.ascontiguousarray(s, dtype=np.uint8) for s in seqs]
        n = len(arrs)
        lens = (C.c_int64 * n)(*[len(a) for a in arrs])
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        self._chk(self._L.fadehip_genome_upload(self._h, n, C.cast(lens, C.c_void_p), C.cast(ptrs, C.c_void_p)))
        self.contig_names = [x.decode() if isinstance(x, bytes) else x for x in names]

    _ARRAYS = (("tid", np.int32), ("pos", np.int32), ("l_seq", np.int32), ("cigar_off", np.uint32), ("seq_off", np.uint32),
               ("flag", np.uint16), ("has_sa", np.uint8), ("cigar_ops", np.uint32), ("seq_packed", np.uint8))

    def _c_batch(self, batch):
        keep = {k: np.ascontiguousarray(batch[k], dtype=dt) for k, dt in self._ARRAYS}
        n = len(keep["pos"])
        b = _lib.ReadBatch()
        b.n_reads = n
        for k, _ in self._ARRAYS:
            setattr(b, k, keep[k].ctypes.data)
        self._hints(b, batch)
        return b, keep, n

    @staticmethod
    def _hints(b, batch):
        """What the caller knows about the batch travels with it (include/fadehip.h: n_skipped, ref_span_bound and the
        ABI-3 bounds n_with_seq / l_seq_min / l_seq_max; see with_bounds)."""
        b.n_skipped = int(batch.get("n_skipped", 0))
        b.ref_span_bound = int(batch.get("ref_span_bound", 0))
        b.n_with_seq = int(batch.get("n_with_seq", 0))
        b.l_seq_min = int(batch.get("l_seq_min", 0))
        b.l_seq_max = int(batch.get("l_seq_max", 0))

    def pinned_batch(self, batch):
        """The batch as ONE pinned block in the canonical layout (fadehip_batch_bytes / fadehip_batch_bind), what a
        reader thread would fill in place: annotate_upload of the returned object is a single hipMemcpyAsync.
        `n_skipped` / `ref_span_bound` of the dict travel along (see clipped_only)."""
        arrs = {k: np.ascontiguousarray(batch[k], dtype=dt) for k, dt in self._ARRAYS}
        n = len(arrs["pos"])
        n_cig, n_seq = int(arrs["cigar_off"][n]) if n else 0, int(arrs["seq_off"][n]) if n else 0
        nbytes = self._L.fadehip_batch_bytes(n, n_cig, n_seq)
        ptr = C.c_void_p()
        self._chk(self._L.fadehip_host_alloc(self._h, nbytes, C.byref(ptr)))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(ptr)
        b = _lib.ReadBatch()
        self._chk(self._L.fadehip_batch_bind(ptr, n, n_cig, n_seq, C.byref(b)))
        for k, dt in self._ARRAYS:
            a = arrs[k]
            if a.nbytes:
                C.memmove(getattr(b, k), a.ctypes.data, a.nbytes)
        self._hints(b, batch)
        return PinnedBatch(b, n, ptr, nbytes)

    @staticmethod
    def clipped_only(batch, ref_span_bound=True):
        """anno.d:61-65 gives an unmapped record, or one without an S op, rs = 0 before anything else is looked at:
        such records need not be sent.  Returns (sub_batch, sent_idx): the records that must be sent, with
        `n_skipped` (counted into read_count by the device path) and, optionally, `ref_span_bound` (max
        cigar.alignedLength, which a reader knows from parsing); rs / read_idx of the results index sent_idx."""
        from . import synth
        co = np.asarray(batch["cigar_off"], dtype=np.int64)
        ops = np.asarray(batch["cigar_ops"])
        is_s = np.concatenate([(ops & 15) == 4, [False]]).astype(np.int64)
        cs = np.concatenate([[0], np.cumsum(is_s)])
        has_s = (cs[co[1:]] - cs[co[:-1]]) > 0
        keep = has_s & ((np.asarray(batch["flag"]) & 4) == 0)
        idx = np.nonzero(keep)[0]
        sub = synth.take(batch, idx)
        sub["n_skipped"] = int(len(keep) - len(idx))
        if ref_span_bound:
            ref = np.isin(ops & 15, _lib.REF_CONSUMING_OPS) * (ops >> 4).astype(np.int64)
            cr = np.concatenate([[0], np.cumsum(ref)])
            al = cr[co[1:]] - cr[co[:-1]]
            sub["ref_span_bound"] = int(al[idx].max()) if len(idx) else 1
        return sub, idx

    @staticmethod
    def compact_sequences(batch):
        """Drop the packed bases of records the device never reads (unmapped, or no S op in the CIGAR; anno.d:61):
        they get an empty slice (seq_off[i+1] == seq_off[i]).  A tenth of the upload for 10 % clipped reads."""
        co = np.asarray(batch["cigar_off"], dtype=np.int64)
        is_s = np.concatenate([(np.asarray(batch["cigar_ops"]) & 15) == 4, [False]]).astype(np.int64)
        cs = np.concatenate([[0], np.cumsum(is_s)])
        has_s = (cs[co[1:]] - cs[co[:-1]]) > 0
        keep = has_s & ((np.asarray(batch["flag"]) & 4) == 0)
        so = np.asarray(batch["seq_off"], dtype=np.int64)
        lens = (so[1:] - so[:-1]) * keep
        out = dict(batch)
        out["seq_packed"] = np.asarray(batch["seq_packed"])[np.repeat(keep, so[1:] - so[:-1])]
        out["seq_off"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
        return out

    @staticmethod
    def with_bounds(batch):
        """What a packing thread knows when it has filled a block, passed along so that upload does no per-record host
        work (ABI 3): every record is sent, the ones the device never aligns (unmapped, or no S op; anno.d:61) without
        their bases; `ref_span_bound` = max cigar.alignedLength over the records that carry bases, `n_with_seq` their
        number, `l_seq_min` / `l_seq_max` their shortest / longest read."""
        out = Context.compact_sequences(batch)
        so = np.asarray(out["seq_off"], dtype=np.int64)
        has = (so[1:] - so[:-1]) > 0
        ls = np.asarray(out["l_seq"])[has]
        out["n_with_seq"] = int(has.sum())
        out["l_seq_min"] = int(ls.min()) if len(ls) else 0
        out["l_seq_max"] = int(ls.max()) if len(ls) else 0
        co = np.asarray(out["cigar_off"], dtype=np.int64)
        ops = np.asarray(out["cigar_ops"])
        ref = np.isin(ops & 15, _lib.REF_CONSUMING_OPS) * (ops >> 4).astype(np.int64)
        cr = np.concatenate([[0], np.cumsum(ref)])
        al = (cr[co[1:]] - cr[co[:-1]])[has]
        out["ref_span_bound"] = int(al.max()) if len(al) else 1
        out["n_skipped"] = 0
        return out

    def annotate_upload(self, slot, batch):
        """batch: a dict of numpy arrays (gathered into the slot's staging block) or a PinnedBatch (one DMA).
        Never waits for the slot's run in flight: upload the batch of the slot's NEXT run right after annotate_run."""
        # (the arrays of the batch in flight and of the batch uploaded for the next run are both kept alive)
        if isinstance(batch, PinnedBatch):
            self._keep[slot] = (self._keep.get(slot, (None,))[-1], batch)
            self._chk(self._L.fadehip_annotate_upload(self._h, slot, C.byref(batch.c)))
            return
        b, keep, n = self._c_batch(batch)
        self._keep[slot] = (self._keep.get(slot, (None,))[-1], keep)
        self._chk(self._L.fadehip_annotate_upload(self._h, slot, C.byref(b)))

    def annotate_run(self, slot, floor_len=5, window=300):
        """Enqueues the whole device path of the slot's batch and returns (errors of the batch surface at results)."""
        self._chk(self._L.fadehip_annotate_run(self._h, slot, floor_len, window))

    def _view(self, addr, count, dtype):
        """numpy view of `count` items at a device-library address; the (address, size) pairs of a streaming run repeat,
        and building a ctypes array type per call costs tens of microseconds."""
        key = (addr, count, np.dtype(dtype).itemsize)
        arr = self._views.get(key)
        if arr is None:
            nbytes = count * np.dtype(dtype).itemsize
            arr = np.frombuffer((C.c_uint8 * nbytes).from_address(addr), dtype=dtype)
            if len(self._views) > 256:
                self._views.clear()
            self._views[key] = arr
        return arr

    def annotate_results(self, slot):
        """Waits for the slot; rs [n], alignments [n_aln], stats [8] as views into the slot's pinned result block
        (valid until the slot is run again), plus n_oversize."""
        v = _lib.AnnoView()
        self._chk(self._L.fadehip_annotate_results(self._h, slot, C.byref(v)))
        n, n_aln = v.n_reads, v.n_aln
        rs = self._view(v.rs, n, np.uint8) if n else np.zeros(0, np.uint8)
        aln = self._view(v.aln, n_aln, ALN_DTYPE) if n_aln else np.zeros(0, ALN_DTYPE)
        self.last_oversize = int(v.n_oversize)
        return rs, aln, np.array(list(v.stats), dtype=np.int64)

    def annotate_collect(self, slot, copy=True):
        """rs [n], alignments [n_aln], stats [8]; with copy=False the arrays are views that the slot's next run
        invalidates."""
        rs, aln, stats = self.annotate_results(slot)
        if copy:
            return rs.copy(), aln.copy(), stats
        return rs, aln, stats

    def annotate(self, batch, floor_len=5, window=300, slot=0):
        self.annotate_upload(slot, batch)
        self.annotate_run(slot, floor_len, window)
        return self.annotate_collect(slot)

    # ---- BGZF compression of an output stream (util.d:65-76, anno.d:47-49)
    def bgzf_deflate_submit(self, lane, data):
        """data: bytes / uint8 array (kept alive until bgzf_deflate_wait)."""
        arr = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        self._keep[("bgzf", lane)] = arr
        self._chk(self._L.fadehip_bgzf_deflate_submit(self._h, lane, arr.ctypes.data, arr.nbytes))

    def bgzf_deflate_wait(self, lane):
        """The BGZF members of the submitted bytes (no end-of-file marker), as bytes."""
        ptr, n = C.c_void_p(), C.c_size_t()
        self._chk(self._L.fadehip_bgzf_deflate_wait(self._h, lane, C.byref(ptr), C.byref(n)))
        self._keep.pop(("bgzf", lane), None)
        return C.string_at(ptr.value, n.value)

    def bgzf_deflate(self, data, lane=0):
        self.bgzf_deflate_submit(lane, data)
        return self.bgzf_deflate_wait(lane)

    def bam_stream(self, ref_names, floor_len=5, window=300, first_record=0, stored=False, tail_trim=0, no_output=False):
        """The file path on the device (fadehip_bam_*): BGZF members of a BAM's records in, BGZF members of the annotated
        records out.  ref_names: the BAM header's contigs (the genome must be uploaded)."""
        return BamStream(self, ref_names, floor_len, window, first_record, stored, tail_trim, no_output)

    def bgzf_inflate(self, members, out_cap=None):
        """Whole BGZF members (bytes / uint8 array) -> their payloads, inflated on the device (CRC32 and ISIZE checked)."""
        arr = np.frombuffer(members, dtype=np.uint8) if isinstance(members, (bytes, bytearray, memoryview)) else np.ascontiguousarray(members, dtype=np.uint8)
        if out_cap is None:  # ISIZE of every member: at most 64 KiB each, and a member takes at least 28 bytes
            out_cap = 65536 * (arr.nbytes // 28 + 1)
            out_cap = min(out_cap, max(1 << 16, arr.nbytes * 1100))
        out = np.empty(out_cap, dtype=np.uint8)
        n = C.c_size_t(0)
        self._chk(self._L.fadehip_bgzf_inflate(self._h, arr.ctypes.data, arr.nbytes, out.ctypes.data, out_cap, C.byref(n)))
        return out[:n.value]

    def sync(self):
        self._chk(self._L.fadehip_sync(self._h))

    def last_profile(self, slot=0):
        ms = (C.c_float * 4)()
        cnt = (C.c_int64 * 6)()
        self._chk(self._L.fadehip_last_run_profile(self._h, slot, C.byref(ms), C.byref(cnt)))
        return dict(gate_ms=ms[0], forward_ms=ms[1], traceback_ms=ms[2], total_ms=ms[3], alignments=cnt[0],
                    cells=cnt[1], trace_bytes=cnt[2], algorithmic_bytes=cnt[3], snapshot_bytes=cnt[4], candidates=cnt[5])


class PinnedBatch:
    """A batch in one pinned block (Context.pinned_batch): `c` is the bound fadehip_read_batch."""

    __slots__ = ("c", "n", "ptr", "nbytes")

    def __init__(self, c, n, ptr, nbytes):
        self.c, self.n, self.ptr, self.nbytes = c, n, ptr, nbytes


def stats_allreduce(contexts, counters):
    """Sum per-device counters with one RCCL all-reduce (single process, one Context per device).
    counters: int64 array [n_ctx, count]; returns the reduced copy (every row = the sum)."""
    L = _lib.load()
    arr = np.ascontiguousarray(counters, dtype=np.int64).copy()
    n, count = arr.shape
    assert n == len(contexts)
    hs = (C.c_void_p * n)(*[c._h for c in contexts])
    rc = L.fadehip_stats_allreduce(hs, n, arr.ctypes.data, count)
    if rc != 0:
        raise FadeHipError(rc, L.fadehip_last_error(contexts[0]._h).decode())
    return arr


class SwResult:
    """What FADE reads from a dparasail result (analysis.d:69-113)."""

    __slots__ = ("score", "position", "cigar", "end_query", "end_ref", "beg_query", "n_ops")

    def __init__(self, rec):
        self.score = int(rec["score"])
        self.position = int(rec["beg_ref"])
        self.n_ops = int(rec["n_ops"])
        self.cigar = [int(o) for o in rec["ops"][:min(self.n_ops, _lib.MAX_OPS)]]
        self.end_query = int(rec["end_query"])
        self.end_ref = int(rec["end_ref"])
        self.beg_query = int(rec["beg_query"])

    def cigar_string(self):
        return cigar_str(self.cigar)


class Parasail:
    """Parasail("ACTGN", 10, 2, 2, -3) — anno.d:36 — with the alignment done on the GPU."""

    def __init__(self, alphabet="ACTGN", open=10, ext=2, match=2, mismatch=-3, device=-1, ctx=None):
        if sorted(alphabet.upper()) != sorted("ACTGN"):
            raise ValueError("the gfx950 kernels carry the ACTGN+wildcard matrix of anno.d:36 only")
        self.ctx = ctx or Context(device=device, open=open, ext=ext, match=match, mismatch=mismatch)

    def sw_striped(self, q_seq, ref_seq):
        return SwResult(self.ctx.sw_batch([q_seq], [ref_seq])[0])

    def sw_striped_batch(self, q_seqs, ref_seqs):
        return [SwResult(r) for r in self.ctx.sw_batch(q_seqs, ref_seqs)]


# ---------------------------------------------------------------- tag formatting (host, as in the reference)
def unpack_seq(seq_packed, off, l_seq):
    """BAM 4-bit -> nt16 codes (uint8 array of l_seq)."""
    b = np.asarray(seq_packed[off:off + (l_seq + 1) // 2], dtype=np.uint8)
    codes = np.empty(2 * len(b), dtype=np.uint8)
    codes[0::2] = b >> 4
    codes[1::2] = b & 15
    return codes[:l_seq]


def format_tags(batch, contig_names, rs, aln):
    """am/as/ar/ab for every artifact read: analysis.d:84-92,108-118 + anno.d:98-107.
    Returns {read_idx: dict(rs=, am=, as_=, ar=, ab=)} for reads with art_left|art_right."""
    out = {}
    for a in aln:
        art = int(a["art"])
        if not art:
            continue
        i = int(a["read_idx"])
        lq = int(batch["l_seq"][i])
        codes = unpack_seq(batch["seq_packed"], int(batch["seq_off"][i]), lq)
        seq = _NT16_ARR[codes].tobytes().decode()
        q_seq = _NT16_ARR[_COMP[codes][::-1]].tobytes().decode()  # util.d:23-34
        qo = int(batch["qual_off"][i])
        bq = (np.asarray(batch["qual"][qo:qo + lq], dtype=np.uint8) + 33).astype(np.uint8).tobytes().decode("latin-1")
        sw = a["sw"]
        n_ops = int(sw["n_ops"])
        ops = [int(o) for o in sw["ops"][:n_ops]]
        name = contig_names[int(batch["tid"][i])]
        apos = int(a["win_start"]) + int(sw["beg_ref"])
        pos = int(batch["pos"][i])
        left = ["", "", "", ""]
        right = ["", "", "", ""]
        if art & 1:  # analysis.d:84-92
            clip = int(a["clip_left"])
            overlap = apos - (pos - clip) if apos >= pos - clip else 0
            lead_s = ops[0] >> 4 if (ops[0] & 15) == 4 else 0
            plen = min(lq, (lq - lead_s) + overlap)
            left = ["%s,%d,%s" % (name, apos, cigar_str(ops)), seq[:plen], q_seq[lq - plen:], bq[:plen]]
        if art & 2:  # analysis.d:108-118
            clip = int(a["clip_right"])
            res_aligned = sum(o >> 4 for o in ops if (o & 15) in _lib.REF_CONSUMING_OPS)
            lhs = pos + int(a["aligned_len"]) + clip
            rhs = apos + res_aligned
            overlap = lhs - rhs if lhs >= rhs else 0
            trail_s = ops[-1] >> 4 if (ops[-1] & 15) == 4 else 0
            plen = min(lq, (lq - trail_s) + overlap)
            right = ["%s,%d,%s" % (name, apos, cigar_str(ops)), seq[lq - plen:], q_seq[:plen], bq[lq - plen:]]
        out[i] = dict(rs=int(rs[i]), am=left[0] + ";" + right[0], as_=left[1] + ";" + right[1],
                      ar=left[2] + ";" + right[2], ab=left[3] + ";" + right[3])
    return out


class BamStream:
    def __init__(self, ctx, ref_names, floor_len, window, first_record, stored=False, tail_trim=0, no_output=False):
        self._ctx, self._L = ctx, ctx._L
        names = [n.encode() if isinstance(n, str) else bytes(n) for n in ref_names]
        arr = (C.c_char_p * max(len(names), 1))(*names)
        cfg = _lib.BamConfig(floor_len, window, len(names), (1 if stored else 0) | (2 if no_output else 0), arr, first_record, tail_trim)
        h = C.c_void_p()
        ctx._chk(self._L.fadehip_bam_open(ctx._h, C.byref(cfg), C.byref(h)))
        self._h = h
        self._keep = None

    def prepare(self, call_bytes):
        """fadehip_bam_prepare: the streams and buffers of calls of call_bytes inflated bytes, made ahead of the first call."""
        self._ctx._chk(self._L.fadehip_bam_prepare(self._h, int(call_bytes)))

    def front(self, members, last=False):
        arr = np.frombuffer(members, dtype=np.uint8) if isinstance(members, (bytes, bytearray, memoryview)) else np.ascontiguousarray(members, dtype=np.uint8)
        self._ctx._chk(self._L.fadehip_bam_front(self._h, arr.ctypes.data if arr.nbytes else None, arr.nbytes, 1 if last else 0))

    def front_raw(self, payload, last=False):
        arr = np.frombuffer(payload, dtype=np.uint8) if isinstance(payload, (bytes, bytearray, memoryview)) else np.ascontiguousarray(payload, dtype=np.uint8)
        self._ctx._chk(self._L.fadehip_bam_front_raw(self._h, arr.ctypes.data if arr.nbytes else None, arr.nbytes, 1 if last else 0))

    def front_raw_ptr(self, ptr, nbytes, last=False):
        """front_raw on memory the caller owns (pinned memory from fadehip_host_alloc: the H2D then runs at PCIe speed)."""
        self._ctx._chk(self._L.fadehip_bam_front_raw(self._h, ptr, nbytes, 1 if last else 0))

    def back(self):
        p, n = C.c_void_p(), C.c_size_t(0)
        self._ctx._chk(self._L.fadehip_bam_back(self._h, C.byref(p), C.byref(n)))
        return C.string_at(p, n.value) if n.value else b""

    def totals(self):
        st, nr, no = (C.c_int64 * 8)(), C.c_int64(0), C.c_int64(0)
        self._ctx._chk(self._L.fadehip_bam_totals(self._h, C.byref(st), C.byref(nr), C.byref(no)))
        return [int(x) for x in st], int(nr.value), int(no.value)

    def close(self):
        if self._h:
            self._L.fadehip_bam_close(self._h)
            self._h = None


def annotate_records(ctx, batch, floor_len=5, window=300):
    """annotateTask over a batch: returns (rs uint8[n], {read_idx: tags}) — anno.d:55-110."""
    rs, aln, _ = ctx.annotate(batch, floor_len, window)
    return rs, format_tags(batch, ctx.contig_names, rs, aln)
