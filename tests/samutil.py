"""Minimal SAM text <-> batch-dict conversion for fixtures and for checking the CLI's output."""
import numpy as np

NT16 = "=ACMGRSVTWYHKDBN"
CIGAR_OPS = "MIDNSHP=X"
_CODE = {c: i for i, c in enumerate(NT16)}


def batch_to_sam(batch, contig_names, contig_lens, qnames=None, extra_header=()):
    lines = ["@HD\tVN:1.6\tSO:unsorted"]
    for n, l in zip(contig_names, contig_lens):
        lines.append("@SQ\tSN:%s\tLN:%d" % (n, l))
    lines.append("@PG\tID:synth\tPN:fade_amd.synth")
    lines.extend(extra_header)
    n = len(batch["pos"])
    for i in range(n):
        lq = int(batch["l_seq"][i])
        so = int(batch["seq_off"][i])
        b = batch["seq_packed"][so:so + (lq + 1) // 2]
        codes = np.empty(2 * len(b), dtype=np.uint8)
        codes[0::2] = b >> 4
        codes[1::2] = b & 15
        seq = "".join(NT16[c] for c in codes[:lq])
        qo = int(batch["qual_off"][i])
        qual = "".join(chr(int(q) + 33) for q in batch["qual"][qo:qo + lq])
        ops = batch["cigar_ops"][batch["cigar_off"][i]:batch["cigar_off"][i + 1]]
        cig = "".join("%d%s" % (int(o) >> 4, CIGAR_OPS[int(o) & 15]) for o in ops) or "*"
        tid = int(batch["tid"][i])
        name = qnames[i] if qnames is not None else "r%d" % (i // 2)
        if isinstance(name, bytes):
            name = name.decode()
        f = [name, str(int(batch["flag"][i])), contig_names[tid] if tid >= 0 else "*", str(int(batch["pos"][i]) + 1),
             "60" if tid >= 0 else "0", cig, "*", "0", "0", seq, qual]
        if int(batch["has_sa"][i]):
            f.append("SA:Z:%s,1,+,50M,60,0;" % contig_names[0])
        lines.append("\t".join(f))
    return "\n".join(lines) + "\n"


def parse_sam(text):
    """Returns (header_lines, records) with records as dicts incl. a `tags` dict of (type, value)."""
    header, recs = [], []
    for line in text.splitlines():
        if not line:
            continue
        if line.startswith("@"):
            header.append(line)
            continue
        f = line.split("\t")
        tags = {}
        for t in f[11:]:
            k, ty, v = t.split(":", 2)
            tags[k] = (ty, v)
        recs.append(dict(qname=f[0], flag=int(f[1]), rname=f[2], pos=int(f[3]) - 1, mapq=int(f[4]), cigar=f[5],
                         rnext=f[6], pnext=int(f[7]), tlen=int(f[8]),
                         seq=f[9], qual=f[10], tags=tags, tag_order=[t.split(":", 1)[0] for t in f[11:]]))
    return header, recs


def sam_to_batch(text):
    """SAM text -> (contig_names, contig_lens, batch dict, qnames)."""
    header, recs = parse_sam(text)
    names, lens = [], []
    for h in header:
        if h.startswith("@SQ"):
            d = dict(x.split(":", 1) for x in h.split("\t")[1:])
            names.append(d["SN"])
            lens.append(int(d["LN"]))
    tid, pos, flag, has_sa, l_seq = [], [], [], [], []
    cigar_off, cigar_ops, seq_off, seq_packed, qual_off, qual, qnames = [0], [], [0], [], [0], [], []
    for r in recs:
        qnames.append(r["qname"])
        tid.append(names.index(r["rname"]) if r["rname"] != "*" else -1)
        pos.append(r["pos"])
        flag.append(r["flag"])
        has_sa.append(1 if "SA" in r["tags"] else 0)
        seq = r["seq"] if r["seq"] != "*" else ""
        l_seq.append(len(seq))
        if r["cigar"] != "*":
            num = ""
            for ch in r["cigar"]:
                if ch.isdigit():
                    num += ch
                else:
                    cigar_ops.append((int(num) << 4) | CIGAR_OPS.index(ch))
                    num = ""
        cigar_off.append(len(cigar_ops))
        codes = [_CODE.get(c.upper(), 15) for c in seq]
        if len(codes) & 1:
            codes.append(0)
        seq_packed.extend((codes[k] << 4) | codes[k + 1] for k in range(0, len(codes), 2))
        seq_off.append(len(seq_packed))
        q = [ord(c) - 33 for c in r["qual"]] if r["qual"] != "*" else [255] * len(seq)
        qual.extend(q)
        qual_off.append(len(qual))
    batch = dict(tid=np.array(tid, np.int32), pos=np.array(pos, np.int32), flag=np.array(flag, np.uint16),
                 has_sa=np.array(has_sa, np.uint8), l_seq=np.array(l_seq, np.int32),
                 cigar_off=np.array(cigar_off, np.uint32), cigar_ops=np.array(cigar_ops, np.uint32),
                 seq_off=np.array(seq_off, np.uint32), seq_packed=np.array(seq_packed, np.uint8),
                 qual_off=np.array(qual_off, np.int64), qual=np.array(qual, np.uint8))
    return names, lens, batch, qnames


def read_fasta(text):
    names, seqs, cur = [], [], []
    for line in text.splitlines():
        if line.startswith(">"):
            if names:
                seqs.append("".join(cur))
            names.append(line[1:].split()[0])
            cur = []
        else:
            cur.append(line.strip())
    if names:
        seqs.append("".join(cur))
    return names, seqs


def bam_to_sam_records(data):
    """Decode a (BGZF) BAM byte string -> (header_text, ref_names, records as parse_sam dicts)."""
    import gzip
    import struct
    raw = gzip.decompress(data)  # BGZF is a series of gzip members
    assert raw[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", raw, 4)
    text = raw[8:8 + l_text].decode()
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, o)
    o += 4
    names = []
    for _ in range(n_ref):
        ln, = struct.unpack_from("<i", raw, o)
        names.append(raw[o + 4:o + 4 + ln - 1].decode())
        o += 4 + ln + 4
    recs = []
    while o < len(raw):
        bs, = struct.unpack_from("<i", raw, o)
        b = raw[o + 4:o + 4 + bs]
        o += 4 + bs
        tid, pos, lqn, mapq, _bin, ncig, flag, lseq, mtid, mpos, tlen = struct.unpack_from("<iiBBHHHiiii", b, 0)
        p = 32
        qname = b[p:p + lqn - 1].decode()
        p += lqn
        cig = "".join("%d%s" % (c >> 4, CIGAR_OPS[c & 15]) for c in struct.unpack_from("<%dI" % ncig, b, p)) or "*"
        p += 4 * ncig
        sq = b[p:p + (lseq + 1) // 2]
        seq = "".join(NT16[(sq[k >> 1] >> (4 if k % 2 == 0 else 0)) & 15] for k in range(lseq)) or "*"
        p += (lseq + 1) // 2
        ql = b[p:p + lseq]
        qual = "*" if (lseq == 0 or ql[0] == 0xFF) else "".join(chr(x + 33) for x in ql)
        p += lseq
        tags, order = {}, []
        while p < len(b):
            tag = b[p:p + 2].decode()
            ty = chr(b[p + 2])
            p += 3
            if ty in "cCsSiI":
                fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[ty]
                v, = struct.unpack_from(fmt, b, p)
                p += struct.calcsize(fmt)
                tags[tag] = ("i", str(v))
                tags[tag + ".bamtype"] = ty
            elif ty == "A":
                tags[tag] = ("A", chr(b[p]))
                p += 1
            elif ty == "f":
                v, = struct.unpack_from("<f", b, p)
                p += 4
                tags[tag] = ("f", "%g" % v)
            elif ty in "ZH":
                e = b.index(b"\0", p)
                tags[tag] = (ty, b[p:e].decode())
                p = e + 1
            elif ty == "B":
                st = chr(b[p])
                n, = struct.unpack_from("<I", b, p + 1)
                fmt = {"c": "b", "C": "B", "s": "h", "S": "H", "i": "i", "I": "I", "f": "f"}[st]
                vals = struct.unpack_from("<%d%s" % (n, fmt), b, p + 5)
                p += 5 + n * struct.calcsize(fmt)
                tags[tag] = ("B", st + "".join(",%g" % v if st == "f" else ",%d" % v for v in vals))
            else:
                raise ValueError("bad aux type " + ty)
            order.append(tag)
        recs.append(dict(qname=qname, flag=flag, rname=names[tid] if tid >= 0 else "*", pos=pos, mapq=mapq, cigar=cig,
                         seq=seq, qual=qual, tags=tags, tag_order=order, mtid=mtid, mpos=mpos, tlen=tlen))
    return text, names, recs
