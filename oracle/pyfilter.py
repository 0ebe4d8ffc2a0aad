"""Pure-Python restatement of `fade out` (source/filter.d:15-91,127-268, source/stats.d:45-72) —
TEST INFRASTRUCTURE ONLY.  Small cases only.  PARITY UNPINNED (the reference has no fixtures for this path).

Records are the dicts of tests/samutil.parse_sam; output is SAM lines without header."""
import re

_QC = set("MIS=X")   # isQueryConsuming
_RC = set("MDN=X")   # isReferenceConsuming


def _cigar_ops(s):
    return [] if s == "*" else [[int(n), c] for n, c in re.findall(r"(\d+)([MIDNSHP=XB])", s)]


def _aligned(ops):
    return sum(n for n, c in ops if c in _RC)


def _parse_long(s):
    m = re.match(r"[+-]?\d+", s)
    if not m:
        return None, s
    return int(m.group()), s[m.end():]


def numerically_aware_cmp(a, b):
    """filter.d:127-167"""
    while a and b:
        nda, ndb = not a[0].isdigit(), not b[0].isdigit()
        if nda and ndb:
            if a[0] == b[0]:
                a, b = a[1:], b[1:]
                continue
            return -1 if a[0] < b[0] else 1
        ai, a2 = _parse_long(a)
        bi, b2 = _parse_long(b)
        ai, a = (-1, a) if ai is None else (ai, a2)
        bi, b = (-1, b) if bi is None else (bi, b2)
        if ai == bi:
            continue
        return -1 if ai < bi else 1
    return 0 if len(a) == len(b) else (-1 if len(a) < len(b) else 1)


def _fmt(rec):
    tags = []
    for k in rec["tag_order"]:
        ty, v = rec["tags"][k]
        tags.append("%s:%s:%s" % (k, ty, v))
    return "\t".join([rec["qname"], str(rec["flag"]), rec["rname"], str(rec["pos"] + 1), str(rec["mapq"]), rec["cigar"],
                      rec.get("rnext", "*"), str(rec.get("pnext", 0)), str(rec.get("tlen", 0)), rec["seq"], rec["qual"]] + tags)


def clip_read(rec, rs, contig0):
    """filter.d:15-91.  Returns the new record dict."""
    ops = _cigar_ops(rec["cigar"])
    total_aligned = _aligned(ops)
    pos = rec["pos"]
    seq, qual = rec["seq"], rec["qual"]
    am = rec["tags"]["am"][1].split(";")

    def reset():  # a zero-filled bam1_t with name, sequence and qualities (filter.d:47-51)
        return dict(qname=rec["qname"], flag=0, rname=contig0, pos=0, mapq=0, cigar="*", rnext="=", pnext=1, tlen=0,
                    seq=seq, qual=qual, tags={}, tag_order=[])

    if rs & 2:
        to_trim = _aligned(_cigar_ops(am[0].split(",")[2]))
        hard = 0
        if to_trim < total_aligned:
            while to_trim:
                n, c = ops[0]
                if c in _QC:
                    seq, qual, hard = seq[1:], qual[1:], hard + 1
                if c in _RC:
                    pos += 1
                    to_trim -= 1
                ops[0][0] -= 1
                if ops[0][0] == 0:
                    ops.pop(0)
        else:
            return reset()
        ops.insert(0, [hard, "H"])
    if rs & 4:
        to_trim = _aligned(_cigar_ops(am[1].split(",")[2]))
        hard = 0
        if to_trim < _aligned(ops):
            while to_trim:
                n, c = ops[-1]
                if c in _QC:
                    seq, qual, hard = seq[:-1], qual[:-1], hard + 1
                if c in _RC:
                    to_trim -= 1
                ops[-1][0] -= 1
                if ops[-1][0] == 0:
                    ops.pop()
        else:
            return reset()
        ops.append([hard, "H"])
    new = dict(rec)
    new["cigar"] = "".join("%d%s" % (n, c) for n, c in ops)
    new["seq"], new["qual"], new["pos"] = seq, qual, pos
    return new


def _ratio(num, den):
    if den == 0:
        return "nan" if num == 0 else "inf"
    import numpy as np
    return "%g" % float(np.float32(num) / np.float32(den))


def fade_out(records, contig0, clip):
    """Returns (SAM lines, stderr stats text).  filter.d:169-268."""
    st = dict(read_count=0, clipped=0, sup=0, art_sup=0, art=0, aln_l=0, aln_r=0)

    def parse(v):
        sc, al, ar, sup = v & 1, (v >> 1) & 1, (v >> 2) & 1, (v >> 5) & 1
        st["clipped"] += sc
        st["art"] += al | ar
        st["sup"] += sup
        st["art_sup"] += (al | ar) & sup
        st["aln_l"] += al
        st["aln_r"] += ar

    out = []
    if clip:
        for r in records:
            st["read_count"] += 1
            if "rs" not in r["tags"]:
                out.append(_fmt(r))
                continue
            v = int(r["tags"]["rs"][1]) & 0xFF
            parse(v)
            out.append(_fmt(clip_read(r, v, contig0) if v & 6 else r))
    else:
        first = records[:10]
        is_sorted = all(numerically_aware_cmp(first[k]["qname"], first[k - 1]["qname"]) >= 0 for k in range(1, len(first)))
        if is_sorted:
            k = 0
            while k < len(records):
                e = k
                while e < len(records) and records[e]["qname"] == records[k]["qname"]:
                    e += 1
                art = False
                for r in records[k:e]:
                    st["read_count"] += 1
                    if "rs" not in r["tags"]:
                        continue
                    v = int(r["tags"]["rs"][1]) & 0xFF
                    parse(v)
                    art |= bool(v & 6)
                if not art:
                    out.extend(_fmt(r) for r in records[k:e])
                k = e
        else:
            for r in records:
                st["read_count"] += 1
                if "rs" not in r["tags"]:
                    continue
                v = int(r["tags"]["rs"][1]) & 0xFF
                parse(v)
                if not (v & 6):
                    out.append(_fmt(r))
    n = st["read_count"]
    text = ("read count:\t%d\nClipped %%:\t%s\n%% With Supplementary alns:\t%s\nArtifact rate:\t%s\n"
            "%% With Supplementary alns and artifacts:\t%s\nArtifact rate left only:\t%s\nArtifact rate right only:\t%s\n" %
            (n, _ratio(st["clipped"], n), _ratio(st["sup"], n), _ratio(st["art"], n), _ratio(st["art_sup"], n),
             _ratio(st["aln_l"], n), _ratio(st["aln_r"], n)))
    return out, text
