"""Host BGZF codec: fade_amd/csrc/host/deflate_fast.hpp must emit standard DEFLATE that zlib
inflates back bit-exactly, inflate_fast.hpp must decode zlib's and its sibling's streams, and the `fade` reader / writer
must carry the same payload with either codec."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fade_amd", "csrc")
FADE = os.path.join(ROOT, "fade_amd", "fade")


def test_deflate_selftest_roundtrips_under_sanitizers():
    """1338 blocks (every size boundary, stored / fixed / dynamic, depth-limited Huffman codes, far matches,
    record-like and mixed data, every effort level and an extreme skip rule) compressed by FastDeflate and inflated by
    zlib, and crc32_fast against zlib's crc32; ASan + UBSan."""
    subprocess.run(["make", "-s", "-C", CSRC, "build/deflate_selftest"], check=True, timeout=600)
    p = subprocess.run([os.path.join(CSRC, "build", "deflate_selftest")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    assert p.returncode == 0, p.stdout.decode() + p.stderr.decode()
    assert b"0 failures" in p.stdout


def _sam(path, n, seed):
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        f.write("@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:1000000\n")
        for i in range(n):
            lq = int(rng.integers(30, 200))
            seq = "".join("ACGT"[k] for k in rng.integers(0, 4, lq))
            qual = "".join(chr(33 + int(k)) for k in rng.choice([2, 11, 25, 37], lq, p=[0.05, 0.1, 0.15, 0.7]))
            f.write("read%d\t0\tchr1\t%d\t60\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\n" % (i, int(rng.integers(1, 900000)), lq, seq,
                                                                              qual, int(rng.integers(0, 4))))


def _bgzf_blocks(raw):
    o = 0
    while o < len(raw):
        assert raw[o:o + 4] == b"\x1f\x8b\x08\x04" and raw[o + 12:o + 14] == b"BC"
        bsize = struct.unpack_from("<H", raw, o + 16)[0] + 1
        payload = zlib.decompress(raw[o + 18:o + bsize - 8], -15)
        crc, isize = struct.unpack_from("<II", raw, o + bsize - 8)
        assert isize == len(payload) and crc == (zlib.crc32(payload) & 0xffffffff)
        yield payload
        o += bsize


def test_writer_payload_is_codec_independent(tmp_path):
    sam = tmp_path / "in.sam"
    _sam(sam, 20000, 3)
    outs = {}
    for codec in ("fast", "fast1", "fast3", "fast4", "zlib"):
        env = dict(os.environ, FADE_BGZF_CODEC=codec[:4], FADE_BGZF_EFFORT=codec[4:] or "2")
        p = subprocess.run([FADE, "out", "-b", "-t", "4", str(sam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env,
                           timeout=300)
        assert p.returncode == 0, p.stderr.decode()
        outs[codec] = p.stdout
    blocks = {k: list(_bgzf_blocks(v)) for k, v in outs.items()}  # every block: header, CRC32, ISIZE, inflates
    assert b"".join(blocks["fast"]) == b"".join(blocks["zlib"])
    assert gzip.decompress(outs["fast"]) == b"".join(blocks["fast"])
    assert outs["fast"][-28:] == outs["zlib"][-28:]  # the BGZF EOF marker
    for k in ("fast1", "fast3", "fast4"):
        assert b"".join(blocks[k]) == b"".join(blocks["zlib"])
    # binned qualities with runs are the payload where zlib level 6's deep search pays most (on uniform qualities the
    # default effort's output is smaller than zlib's)
    assert len(outs["fast1"]) <= 1.06 * len(outs["zlib"])
    assert len(outs["fast"]) <= 1.035 * len(outs["zlib"])
    assert len(outs["fast3"]) <= 1.04 * len(outs["zlib"])
    assert len(outs["fast4"]) <= 1.015 * len(outs["zlib"])
    # and the BAM it wrote reads back through the BAM reader to the same records
    bam = tmp_path / "x.bam"
    bam.write_bytes(outs["fast"])
    p = subprocess.run([FADE, "out", "-t", "4", str(bam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    q = subprocess.run([FADE, "out", "-t", "4", str(sam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    strip = lambda t: [l for l in t.decode().splitlines() if not l.startswith("@PG")]
    assert p.returncode == 0 and strip(p.stdout) == strip(q.stdout)
    # both inflaters (inflate_fast.hpp, zlib) read both deflaters' output to the same records
    for writer in ("fast", "zlib"):
        bam.write_bytes(outs[writer])
        for reader in ("fast", "zlib"):
            r = subprocess.run([FADE, "out", "-t", "4", str(bam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                               env=dict(os.environ, FADE_BGZF_CODEC=reader))
            assert r.returncode == 0 and strip(r.stdout) == strip(q.stdout), (writer, reader)


def test_reader_rejects_a_corrupt_block(tmp_path):
    """A flipped payload byte must not pass silently: the reader checks ISIZE and the CRC32 of every BGZF block."""
    sam = tmp_path / "in.sam"
    _sam(sam, 3000, 5)
    p = subprocess.run([FADE, "out", "-b", "-t", "2", str(sam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0
    good = bytearray(p.stdout)
    bam = tmp_path / "bad.bam"
    for pos in (len(good) // 2, len(good) // 3):
        bad = bytearray(good)
        bad[pos] ^= 0x5a
        bam.write_bytes(bytes(bad))
        q = subprocess.run([FADE, "out", "-t", "2", str(bam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert q.returncode != 0, "corruption at byte %d went unnoticed" % pos
    bam.write_bytes(bytes(good))
    assert subprocess.run([FADE, "out", "-t", "2", str(bam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300).returncode == 0


def test_reader_rejects_a_record_whose_fields_overrun_it(tmp_path):
    """A well-formed BGZF stream can still carry a BAM record whose l_read_name / n_cigar_op / l_seq point past its end."""
    sam = tmp_path / "in.sam"
    _sam(sam, 50, 8)
    p = subprocess.run([FADE, "out", "-u", "-t", "1", str(sam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0
    payload = bytearray(b"".join(_bgzf_blocks(p.stdout)))
    # header: magic, l_text, text, n_ref, then per reference l_name, name, l_ref
    o = 4
    l_text = struct.unpack_from("<i", payload, o)[0]
    o += 4 + l_text
    n_ref = struct.unpack_from("<i", payload, o)[0]
    o += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", payload, o)[0]
        o += 4 + l_name + 4
    struct.pack_into("<i", payload, o + 4 + 16, 1 << 20)  # l_seq of the first record: far beyond its block_size

    def bgzf(data):
        out = bytearray()
        for k in range(0, len(data), 0xff00):
            chunk = bytes(data[k:k + 0xff00])
            c = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = c.compress(chunk) + c.flush()
            out += b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body
            out += struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
        return bytes(out) + p.stdout[-28:]

    bad = tmp_path / "bad.bam"
    bad.write_bytes(bgzf(payload))
    q = subprocess.run([FADE, "out", "-t", "1", str(bad)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert q.returncode != 0 and b"corrupt BAM record" in q.stderr


def test_block_writer_with_layout_hints(tmp_path):
    """The path `fade annotate` writes through (Writer::write_block: records framed in place, layout hints to the
    compressor — packed bases skipped, qualities probed mildly) on qualities with long runs, where wrong hints would cost
    most: the payload must come back byte for byte, and the file must not be larger than zlib level 6's."""
    tools = os.path.join(ROOT, "tools")
    subprocess.run(["make", "-s", "-C", tools, "sam2bam"], check=True, timeout=600)
    rng = np.random.default_rng(7)
    sam = tmp_path / "runny.sam"
    with open(sam, "w") as f:
        f.write("@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:1000000\n")
        for i in range(20000):
            lq = int(rng.integers(100, 151))
            seq = "".join("ACGT"[k] for k in rng.integers(0, 4, lq))
            change = rng.random(lq) < 0.08
            vals = rng.choice([2, 11, 25, 37], lq, p=[0.05, 0.1, 0.15, 0.7])
            q, cur = [], 37
            for k in range(lq):
                if change[k]:
                    cur = int(vals[k])
                q.append(chr(33 + cur))
            f.write("read%d\t0\tchr1\t%d\t60\t%dM\t*\t0\t0\t%s\t%s\tNM:i:%d\n" % (i, int(rng.integers(1, 900000)), lq, seq, "".join(q),
                                                                              int(rng.integers(0, 4))))
    outs = {}
    for codec in ("fast", "zlib"):
        p = subprocess.run([os.path.join(tools, "sam2bam"), str(sam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                           env=dict(os.environ, FADE_BGZF_CODEC=codec))
        assert p.returncode == 0, p.stderr.decode()
        outs[codec] = p.stdout
    assert b"".join(_bgzf_blocks(outs["fast"])) == b"".join(_bgzf_blocks(outs["zlib"]))
    assert len(outs["fast"]) <= 1.0 * len(outs["zlib"]), (len(outs["fast"]), len(outs["zlib"]))
    # and the records are the ones `fade out` (the per-record reader / writer) makes of the same SAM
    q = subprocess.run([FADE, "out", "-b", "-t", "4", str(sam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert q.returncode == 0
    a, b = gzip.decompress(outs["fast"]), gzip.decompress(q.stdout)
    la, lb = int.from_bytes(a[4:8], "little"), int.from_bytes(b[4:8], "little")
    skip = lambda x, l: x[8 + l:]  # (the header texts differ by the @PG line `fade out` adds)
    ra, rb = skip(a, la), skip(b, lb)
    assert ra[ra.index(b"read0\0") - 36:] == rb[rb.index(b"read0\0") - 36:]


def test_sam_lines_that_cannot_be_bam_records_are_refused(tmp_path):
    """BAM stores l_read_name in a byte and n_cigar_op in 16 bits.  A SAM line beyond either used to be laid out with
    truncated counts in front of untruncated bytes (every later field read from the wrong place); it is an error now,
    on the per-record path (`fade out`) as on the block path `fade annotate` reads through (tools/sam2bam)."""
    tools = os.path.join(ROOT, "tools")
    subprocess.run(["make", "-s", "-C", tools, "sam2bam"], check=True, timeout=600)
    hdr = "@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:1000000\n"
    long_name = tmp_path / "longname.sam"
    long_name.write_text(hdr + "%s\t0\tchr1\t100\t60\t10M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\n" % ("q" * 300))
    many_ops = tmp_path / "manyops.sam"
    n = 70000
    many_ops.write_text(hdr + "r1\t0\tchr1\t100\t60\t%s\t*\t0\t0\t%s\t%s\n" % ("1M1I" * (n // 2), "A" * n, "I" * n))
    for sam, what in ((long_name, b"QNAME"), (many_ops, b"CIGAR")):
        for cmd in ([FADE, "out", "-b", "-t", "2", str(sam)], [os.path.join(tools, "sam2bam"), str(sam)]):
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            assert p.returncode != 0 and what in p.stderr, (cmd, p.stderr[-300:])
    ok = tmp_path / "ok.sam"
    ok.write_text(hdr + "%s\t0\tchr1\t100\t60\t10M\t*\t0\t0\tACGTACGTAC\tIIIIIIIIII\n" % ("q" * 254))
    p = subprocess.run([FADE, "out", "-b", "-t", "2", str(ok)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0 and (b"q" * 254 + b"\0") in gzip.decompress(p.stdout)


def test_reader_rejects_a_member_whose_trailer_was_zeroed(tmp_path):
    """A member whose CRC32 and ISIZE fields read 0 while its DEFLATE stream holds records: an empty member must BE one (its
    stream ends without a byte of output), else its records would vanish without a word — here every member holds whole
    records, so nothing downstream could notice.  A real empty member in the middle of the file stays legal."""
    sam = tmp_path / "in.sam"
    _sam(sam, 200, 9)
    p = subprocess.run([FADE, "out", "-u", "-t", "1", str(sam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0
    payload = b"".join(_bgzf_blocks(p.stdout))
    o = 4
    l_text = struct.unpack_from("<i", payload, o)[0]
    o += 4 + l_text
    n_ref = struct.unpack_from("<i", payload, o)[0]
    o += 4
    for _ in range(n_ref):
        o += 4 + struct.unpack_from("<i", payload, o)[0] + 4
    pieces, recs = [payload[:o]], []
    while o < len(payload):
        bs = struct.unpack_from("<i", payload, o)[0]
        recs.append(payload[o:o + 4 + bs])
        o += 4 + bs
    pieces += [b"".join(recs[k:k + 10]) for k in range(0, len(recs), 10)]

    def member(chunk):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        return b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(body) + 25) + body + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))

    ms = [member(x) for x in pieces]
    eof = p.stdout[-28:]
    bam = tmp_path / "x.bam"
    count = lambda out: sum(1 for line in out.splitlines() if line and not line.startswith(b"@"))
    bam.write_bytes(b"".join(ms[:5]) + eof + b"".join(ms[5:]) + eof)  # (an empty member in the middle: fine)
    q = subprocess.run([FADE, "out", "-t", "2", str(bam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert q.returncode == 0 and count(q.stdout) == len(recs), q.stderr.decode()[-500:]
    bad = list(ms)
    bad[7] = bad[7][:-8] + bytes(8)
    bam.write_bytes(b"".join(bad) + eof)
    q = subprocess.run([FADE, "out", "-t", "2", str(bam)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert q.returncode != 0 and b"corrupt" in q.stderr, "ten records vanished without a word (%d of %d came out)" % (count(q.stdout), len(recs))
