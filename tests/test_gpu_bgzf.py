"""BGZF compression on the device (include/fadehip.h fadehip_bgzf_deflate_*; fade_amd/csrc/bgzf_deflate.hpp) — what
htslib's bgzf_write + zlib do for the write at anno.d:47-49 (util.d:65-76, SAMWriterTypes.BAM).
Three properties, on every payload: (1) standard DEFLATE in BGZF members — Python's gzip / zlib inflate them, which also
checks each member's CRC32 and ISIZE; (2) the inflated bytes are the input, byte for byte; (3) on BAM payloads the stream
is no larger than zlib level 6's (htslib's default) over the same 0xff00-byte blocks."""
import gzip
import struct
import zlib

import numpy as np
import pytest

import fade_amd
from fade_amd import _lib

pytestmark = pytest.mark.gpu

BLOCK = _lib.BGZF_BLOCK  # htslib's block size: what the zlib -6 comparator is cut into, and the device's larger geometry
HTS_BLOCK = BLOCK
SMALL = 0x7f00           # the device's smaller geometry (two blocks per CU), taken while the stream hardly compresses
EOF_MARK = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def members(buf):
    """[(payload, crc, isize)] of a BGZF byte stream, checking the framing (SAM spec 4.1)."""
    out, at = [], 0
    while at < len(buf):
        assert buf[at:at + 4] == b"\x1f\x8b\x08\x04" and buf[at + 10:at + 16] == b"\x06\x00BC\x02\x00", at
        bsize = struct.unpack_from("<H", buf, at + 16)[0] + 1
        crc, isize = struct.unpack_from("<II", buf, at + bsize - 8)
        out.append((buf[at + 18:at + bsize - 8], crc, isize))
        at += bsize
    assert at == len(buf)
    return out


def zlib6_size(data):
    total = 0
    for o in range(0, len(data), HTS_BLOCK):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        total += len(c.compress(data[o:o + HTS_BLOCK]) + c.flush()) + 26
    return total


def bam_payload(n_reads, seed, runny):
    """An uncompressed BAM record stream: 100-150 base reads, qualities uniform in [20, 40] or binned with runs (the law of
    tests/test_bgzf_codec.py::test_block_writer_with_layout_hints), an NM tag."""
    rng = np.random.default_rng(seed)
    parts = []
    for i in range(n_reads):
        lq = int(rng.integers(100, 151))
        name = b"read%d\0" % i
        seq = rng.integers(0, 4, lq + (lq & 1))
        nt = np.array([1, 2, 4, 8], dtype=np.uint8)[seq]
        packed = ((nt[0::2] << 4) | nt[1::2]).astype(np.uint8).tobytes()
        if runny:
            change = rng.random(lq) < 0.08
            vals = rng.choice([2, 11, 25, 37], lq, p=[0.05, 0.1, 0.15, 0.7])
            q, cur = np.empty(lq, np.uint8), 37
            for k in range(lq):
                if change[k]:
                    cur = int(vals[k])
                q[k] = cur
            qual = q.tobytes()
        else:
            qual = rng.integers(20, 41, lq, dtype=np.uint8).tobytes()
        body = struct.pack("<iiBBHHHiiii", 0, int(rng.integers(0, 900000)), len(name), 60, 4681, 1, 0, lq, -1, -1, 0) + name + \
            struct.pack("<I", lq << 4) + packed + qual + b"NMC" + bytes([int(rng.integers(0, 4))])
        parts.append(struct.pack("<I", len(body)) + body)
    return b"".join(parts)


@pytest.fixture(scope="module")
def payloads():
    rng = np.random.default_rng(1)
    text = b"".join(b"the quick brown fox jumps over the lazy dog %d\n" % int(x) for x in rng.integers(0, 1000, 9000))
    return {
        "bam, uniform qualities": bam_payload(3000, 2, False),
        "bam, run-heavy qualities": bam_payload(3000, 3, True),
        "random": rng.integers(0, 256, 3 * BLOCK + 17, dtype=np.uint8).tobytes(),
        "zeros": bytes(2 * BLOCK + 5),
        "text": text,
        "short period": bytes((i % 7) * 31 & 255 for i in range(BLOCK)),
        "one block exactly": rng.integers(0, 4, BLOCK, dtype=np.uint8).tobytes(),
        "one block and a byte": rng.integers(0, 4, BLOCK + 1, dtype=np.uint8).tobytes(),
    }


def test_members_inflate_to_the_input_and_are_no_larger_than_zlib6(ctx, payloads):
    for name, data in payloads.items():
        out = ctx.bgzf_deflate(data)
        ms = members(out)
        assert len(ms) in ((len(data) + BLOCK - 1) // BLOCK, (len(data) + SMALL - 1) // SMALL), name
        cut = BLOCK if len(ms) == (len(data) + BLOCK - 1) // BLOCK else SMALL
        at = 0
        for payload, crc, isize in ms:  # each member on its own: raw DEFLATE, CRC32 and ISIZE of its block
            raw = zlib.decompress(payload, -15)
            assert raw == data[at:at + isize] and zlib.crc32(raw) == crc and isize == min(cut, len(data) - at), (name, at)
            at += isize
        assert gzip.decompress(out + EOF_MARK) == data, name  # and as the file a BAM reader sees
        if name.startswith("bam"):
            assert len(out) <= zlib6_size(data), (name, len(out), zlib6_size(data))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 258, 259, 1000, 4099])
def test_tiny_streams(ctx, n):
    rng = np.random.default_rng(n)
    for data in (rng.integers(0, 256, n, dtype=np.uint8).tobytes(), bytes(n), bytes([65 + (i % 3) for i in range(n)])):
        out = ctx.bgzf_deflate(data)
        assert gzip.decompress(out + EOF_MARK) == data
        assert len(members(out)) == 1


def test_two_lanes_in_flight_and_repeated_use(ctx):
    """Two submissions in flight (each lane has its own stream and buffers), lanes reused with other sizes."""
    rng = np.random.default_rng(9)
    datas = [bam_payload(int(k), 20 + i, i % 2 == 1) for i, k in enumerate((1500, 400, 2500, 50, 1200))]
    ref = [ctx.bgzf_deflate(d) for d in datas]
    for d, r in zip(datas, ref):
        assert gzip.decompress(r + EOF_MARK) == d
    got = [None] * len(datas)
    for k in range(len(datas) + 2):
        lane = k % 2
        if k >= 2:
            got[k - 2] = ctx.bgzf_deflate_wait(lane)
        if k < len(datas):
            ctx.bgzf_deflate_submit(lane, datas[k])
    assert got == ref  # (the compressor is deterministic)


def test_large_stream_at_rate(ctx):
    """64 MB of BAM payload: round trip, and the device rate printed (-s shows it)."""
    import time
    base = bam_payload(20000, 77, False)
    data = base * (64 * 1024 * 1024 // len(base))
    ctx.bgzf_deflate(data[:BLOCK * 8])
    t0 = time.perf_counter()
    out = ctx.bgzf_deflate(data)
    dt = time.perf_counter() - t0
    print("bgzf deflate: %.1f MB in %.1f ms = %.2f GB/s (host memory pageable), ratio %.4f" % (len(data) / 1e6, dt * 1e3, len(data) / dt / 1e9, len(out) / len(data)))
    assert gzip.decompress(out + EOF_MARK) == data


@pytest.mark.parametrize("geom", ["64", "32"])
def test_round_trips_of_many_shapes_in_both_geometries(monkeypatch, geom):
    """Both block geometries (FADEHIP_BGZF_GEOM pins one), 120 streams of mixed character — runs, text, noise, periodic
    stretches, sizes around the block boundaries —: zlib inflates what the device wrote, the device inflates it too, and
    both give the input back."""
    monkeypatch.setenv("FADEHIP_BGZF_GEOM", geom)
    c = fade_amd.Context(device=0)
    try:
        rng = np.random.default_rng(int(geom))
        cut = 0xff00 if geom == "64" else 0x7f00
        words = [b"chr1", b"\t", b"150M", b"=", b"NM:i:0", b"read", b"\x00\x00\x00", b"FFFFFFFF", b"ACGT"]
        for trial in range(120):
            parts = []
            n_target = int(rng.choice([1, 2, 3, cut - 1, cut, cut + 1, 2 * cut - 3, int(rng.integers(1, 5 * cut))]))
            while sum(len(p) for p in parts) < n_target:
                kind = int(rng.integers(0, 5))
                m = int(rng.integers(1, 3000))
                if kind == 0:
                    parts.append(bytes([int(rng.integers(0, 256))]) * m)
                elif kind == 1:
                    parts.append(rng.integers(0, 256, m, dtype=np.uint8).tobytes())
                elif kind == 2:
                    parts.append(b"".join(words[int(k)] for k in rng.integers(0, len(words), m // 4 + 1)))
                elif kind == 3:
                    period = rng.integers(0, 256, int(rng.integers(2, 40)), dtype=np.uint8).tobytes()
                    parts.append((period * (m // len(period) + 1))[:m])
                else:
                    parts.append(rng.choice(np.frombuffer(b"ACGT", np.uint8), m).tobytes())
            data = b"".join(parts)[:n_target]
            out = bytes(c.bgzf_deflate(data))
            assert gzip.decompress(out + EOF_MARK) == data, (trial, len(data))
            assert c.bgzf_inflate(out).tobytes() == data, (trial, len(data))
            assert len(members(out)) == (len(data) + cut - 1) // cut
    finally:
        c.close()


def test_the_small_geometry_is_the_cpu_model_byte_for_byte(monkeypatch, tmp_path, payloads):
    """host/selftest/gpu_deflate_model.cpp restates the 0x7f00-byte-block compressor on the CPU (segments, seams, the table's
    ways and buckets, the parse, bgzf_huff.hpp's code lengths and header): every member the device writes must hold exactly the
    model's bytes — whatever the kernel does in registers, ballots or lanes instead of the model's loops."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "fade_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "-s", "build/gpu_deflate_model"])
    monkeypatch.setenv("FADEHIP_BGZF_GEOM", "32")
    c = fade_amd.Context(device=0)
    try:
        for k, name in enumerate(("bam, uniform qualities", "bam, run-heavy qualities", "text", "one block and a byte")):
            data = payloads[name]
            src, dump = tmp_path / ("p%d.bin" % k), tmp_path / ("p%d.model" % k)
            src.write_bytes(data)
            subprocess.run([os.path.join(csrc, "build", "gpu_deflate_model"), str(src)], check=True, stdout=subprocess.DEVNULL,
                           env=dict(os.environ, MODEL_DUMP=str(dump), ASAN_OPTIONS="detect_leaks=0"))
            raw, want, at = dump.read_bytes(), [], 0
            while at < len(raw):
                n = struct.unpack_from("<I", raw, at)[0]
                want.append(raw[at + 4:at + 4 + n])
                at += 4 + n
            got = [m[0] for m in members(bytes(c.bgzf_deflate(data)))]
            assert len(got) == len(want) == (len(data) + SMALL - 1) // SMALL, name
            for j, (g, w) in enumerate(zip(got, want)):
                assert g == w, "%s: member %d differs from the model (%d / %d bytes)" % (name, j, len(g), len(w))
    finally:
        c.close()
