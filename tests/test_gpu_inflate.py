"""BGZF decompression on the device (include/fadehip.h fadehip_bgzf_inflate; fade_amd/csrc/bgzf_inflate.hpp) — what
htslib's bgzf_read + zlib's inflate do under `bam.allRecords` (anno.d:44).  The checker is zlib: members made by zlib at
every level and strategy (stored, fixed and dynamic Huffman blocks, several DEFLATE blocks per member, long codes,
overlapping matches) must inflate to the bytes zlib was given; members made by the device compressor must come back;
corrupt members must end in an error (a status, never a hang or an out-of-bounds access)."""
import struct
import time
import zlib

import numpy as np
import pytest

import fade_amd
from fade_amd import _lib

pytestmark = pytest.mark.gpu

EOF_MARK = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def member(payload, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, raw=None, extra=b""):
    """One BGZF member around zlib's raw DEFLATE of payload (or the given raw stream); `extra`: further gzip subfields."""
    if raw is None:
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        raw = c.compress(payload) + c.flush()
    xlen = 6 + len(extra)
    bsize = 12 + xlen + len(raw) + 8 - 1
    assert bsize < 65536
    head = struct.pack("<BBBBIBBH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, xlen) + extra + b"BC" + struct.pack("<HH", 2, bsize)
    return head + raw + struct.pack("<II", zlib.crc32(payload) & 0xffffffff, len(payload))


def payloads():
    rng = np.random.default_rng(5)
    text = (b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"@SQ\tSN:chr%d\tLN:%d\n" % (k, 1000 * k) for k in range(1, 2000)))
    out = {
        "empty": b"",
        "one": b"A",
        "two": b"AB",
        "zeros": bytes(65280),
        "max_isize_zeros": bytes(65536),
        "run_then_noise": bytes(3000) + rng.integers(0, 256, 5000, dtype=np.uint8).tobytes() + b"\x07" * 700,
        "random": rng.integers(0, 256, 40000, dtype=np.uint8).tobytes(),
        "text": text[:65280],
        "low_entropy": rng.choice(np.frombuffer(b"ACGT", np.uint8), 65280).tobytes(),
        "skewed": rng.choice(np.arange(256, dtype=np.uint8), 60000, p=np.r_[[0.5], np.full(255, 0.5 / 255)]).tobytes(),
        "period3": (b"abc" * 22000)[:65280],
        "period_300": (rng.integers(0, 256, 300, dtype=np.uint8).tobytes() * 220)[:65280],
    }
    # a geometric symbol law: code lengths up to 15 bits (beyond the 10-bit table)
    p = 0.5 ** np.arange(1, 41)
    out["long_codes"] = rng.choice(np.arange(40, dtype=np.uint8), 65000, p=p / p.sum()).tobytes()
    return out


@pytest.fixture(scope="module")
def ctx():
    c = fade_amd.Context(device=0)
    yield c
    c.close()


@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                            (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE),
                                            (6, zlib.Z_FILTERED)])
def test_members_made_by_zlib(ctx, level, strategy):
    ps = payloads()
    names = [k for k in ps if not (level == 0 and len(ps[k]) > 65000)]  # (a stored 64 KiB payload does not fit a member)
    stream = b"".join(member(ps[k], level, strategy) for k in names)
    got = ctx.bgzf_inflate(stream).tobytes()
    want = b"".join(ps[k] for k in names)
    assert len(got) == len(want)
    assert got == want


def test_several_deflate_blocks_in_one_member_and_other_subfields(ctx):
    rng = np.random.default_rng(8)
    a, b, c = bytes(500), rng.integers(0, 256, 3000, dtype=np.uint8).tobytes(), b"GATTACA" * 900
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(a) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(b) + co.flush(zlib.Z_SYNC_FLUSH) + co.compress(c) + co.flush()
    extra = b"XY" + struct.pack("<H", 3) + b"abc"  # another gzip subfield in front of BC
    stream = member(a + b + c, raw=raw, extra=extra) + EOF_MARK + member(b"tail")
    assert ctx.bgzf_inflate(stream).tobytes() == a + b + c + b"tail"


def test_members_made_by_the_device_compressor_come_back(ctx):
    rng = np.random.default_rng(11)
    data = np.concatenate([rng.integers(0, 256, 100000, dtype=np.uint8), np.zeros(70000, np.uint8),
                           rng.choice(np.frombuffer(b"ACGT#FFF", np.uint8), 300000)]).tobytes()
    comp = bytes(ctx.bgzf_deflate(data))
    assert ctx.bgzf_inflate(comp).tobytes() == data


def test_corrupt_members_are_reported_not_hung_on(ctx):
    ps = payloads()
    good = member(ps["text"]) + member(ps["random"]) + member(ps["period3"])
    assert len(ctx.bgzf_inflate(good)) == len(ps["text"]) + len(ps["random"]) + len(ps["period3"])
    m = bytearray(member(ps["text"]))
    bad_crc = bytes(m[:-8]) + struct.pack("<I", 12345) + bytes(m[-4:])
    bad_isize = bytes(m[:-4]) + struct.pack("<I", len(ps["text"]) - 1)
    big_isize = bytes(m[:-4]) + struct.pack("<I", 70000)
    truncated = bytes(m[:-30])
    bad_magic = b"\x1f\x8c" + bytes(m[2:])
    zero_trailer = bytes(m[:-8]) + bytes(8)  # claims to be empty: its stream says otherwise
    zero_tail = bytes(m[:200]) + bytes(len(m) - 200)
    for what, stream in (("crc", bad_crc), ("isize", bad_isize), ("isize > 64 KiB", big_isize), ("truncated", truncated), ("magic", bad_magic),
                         ("zeroed trailer", zero_trailer), ("zeroed from byte 200 on", zero_tail)):
        with pytest.raises(fade_amd.FadeHipError):
            ctx.bgzf_inflate(member(b"ok") + stream)
    # bit flips anywhere in the DEFLATE streams: every outcome but the original bytes must be an error (the CRC sees to it)
    rng = np.random.default_rng(3)
    want = ps["text"] + ps["random"] + ps["period3"]
    n_err = 0
    for trial in range(60):
        s = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(0, len(s)))
            s[at] ^= 1 << int(rng.integers(0, 8))
        try:
            got = ctx.bgzf_inflate(bytes(s)).tobytes()
            assert got == want, "a corrupted stream inflated to other bytes without an error"
        except fade_amd.FadeHipError:
            n_err += 1
    assert n_err > 30
    # and the context still works
    assert ctx.bgzf_inflate(good).tobytes() == want


def test_large_stream_at_rate(ctx):
    rng = np.random.default_rng(21)
    # BAM-like bytes: 4-letter packed bases (incompressible), qualities with runs, some structure
    n = 64 << 20
    q = np.repeat(rng.choice(np.array([2, 11, 25, 37], np.uint8), n // 16), 8)[:n // 2]
    data = np.concatenate([rng.integers(0, 256, n // 2, dtype=np.uint8), q]).tobytes()
    blocks = [data[o:o + 0xff00] for o in range(0, len(data), 0xff00)]
    stream = b"".join(member(b, 1) for b in blocks)
    ctx.bgzf_inflate(stream[:1 << 20] if False else stream, out_cap=len(data) + 65536)  # warm-up (allocations)
    t = time.perf_counter()
    got = ctx.bgzf_inflate(stream, out_cap=len(data) + 65536)
    dt = time.perf_counter() - t
    assert got.tobytes() == data
    print("device inflate: %d members, %.0f MB -> %.0f MB in %.1f ms incl. H2D/D2H from pageable memory (%.2f GB/s of payload)" % (
        len(blocks), len(stream) / 1e6, len(data) / 1e6, dt * 1e3, len(data) / dt / 1e9))
