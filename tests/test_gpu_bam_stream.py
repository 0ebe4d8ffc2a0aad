"""The file path on the device (include/fadehip.h fadehip_bam_*; fade_amd/csrc/bam_device.hpp, bgzf_inflate.hpp,
bgzf_deflate.hpp): `fade annotate -b in.bam ref.fa` with inflate, record framing, annotateTask, the tags of
anno.d:63,94-107 and deflate all on the device.  The checker is the host pipeline (FADE_BAM_DEVICE=0), itself held to the
oracle's golden records by tests/test_gpu_cli.py: the two outputs must inflate to the SAME BYTES — header, every record,
every tag, in order."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import fade_amd
import samutil

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FADE = os.path.join(ROOT, "fade_amd", "fade")
GOLD = os.path.join(ROOT, "tests", "golden")


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([FADE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=e)


def _bam_of(sam, path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "-s", "sam2bam"])
    with open(path, "wb") as fo:
        subprocess.check_call([os.path.join(ROOT, "tools", "sam2bam"), str(sam)], stdout=fo)


def _params(tag):
    for line in open(os.path.join(GOLD, tag + ".expected.tsv")):
        if line.startswith("#floor_len"):
            p = dict(kv.split("=") for kv in line[1:].split())
            return int(p["floor_len"]), int(p["window"])
    raise AssertionError(tag)


def _members(buf):
    out, at = [], 0
    while at < len(buf):
        bsize = struct.unpack_from("<H", buf, at + 16)[0] + 1
        out.append(buf[at:at + bsize])
        at += bsize
    return out


@pytest.mark.parametrize("tag", ["anno_c1", "anno_c2", "anno_c5", "anno_floor0"])
def test_device_file_path_equals_the_host_pipeline_on_the_golden_inputs(tmp_path, tag):
    floor_len, window = _params(tag)
    bam = tmp_path / "in.bam"
    _bam_of(os.path.join(GOLD, tag + ".sam"), bam)
    base = ["annotate", "--stats", "--timing", "--min-length", str(floor_len), "-w", str(window), "-b", str(bam), os.path.join(GOLD, tag + ".fa")]
    dev = _run(base, {"FADE_BAM_INFLATE": "device"})
    host = _run(base, {"FADE_BAM_DEVICE": "0"})
    assert dev.returncode == 0, dev.stderr.decode()[-2000:]
    assert host.returncode == 0, host.stderr.decode()[-2000:]
    assert b"file path on the device" in dev.stderr and b"file path on the device" not in host.stderr
    assert gzip.decompress(dev.stdout) == gzip.decompress(host.stdout)
    assert dev.stdout.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    # the file path's OWN bytes against the oracle's golden records (not only through the host pipeline): rs, am, as, ar, ab
    # of every record and their order (anno.d:94-107), inflate on the device and on the host pool
    from test_gpu_cli import _check_records, _expected
    exp = _expected(tag)[0]
    _check_records(samutil.bam_to_sam_records(dev.stdout)[2], exp)
    raw = _run(base, {"FADE_BAM_INFLATE": "host"})
    assert raw.returncode == 0 and b"file path on the device" in raw.stderr
    _check_records(samutil.bam_to_sam_records(raw.stdout)[2], exp)
    few = _run(base[:1] + ["-t", "2"] + base[1:])  # (with fewer than 8 threads the device inflates: nothing is said about it here)
    assert few.returncode == 0 and b"inflate on 0 host threads" in few.stderr, few.stderr.decode()[-800:]
    assert samutil.bam_to_sam_records(few.stdout)[2] == samutil.bam_to_sam_records(dev.stdout)[2]
    stats = lambda err: [l for l in err.decode().splitlines() if l.startswith(("read count", "Clipped", "% With", "Artifact"))]
    assert stats(dev.stderr) == stats(host.stderr) and len(stats(dev.stderr)) == 7
    # a second pass over the annotated file: every record carries rs (and some am / as / ar / ab) already — htslib's
    # update-in-place semantics on the device, again equal to the host's
    again = tmp_path / "anno.bam"
    again.write_bytes(dev.stdout)
    base2 = base[:-2] + [str(again), base[-1]]
    dev2 = _run(base2, {"FADE_BAM_INFLATE": "host"})
    host2 = _run(base2, {"FADE_BAM_DEVICE": "0"})
    assert dev2.returncode == 0 and host2.returncode == 0, dev2.stderr.decode()[-2000:]
    assert gzip.decompress(dev2.stdout) == gzip.decompress(host2.stdout)
    _, _, r1 = samutil.bam_to_sam_records(dev.stdout)
    _, _, r2 = samutil.bam_to_sam_records(dev2.stdout)
    assert r1 == r2
    _check_records(r2, exp)


@pytest.mark.parametrize("seed,floor_len,window", [(11, 5, 100), (12, 0, 40), (13, 7, 300)])
def test_device_file_path_against_the_oracle_on_iupac_reads_and_a_soft_masked_fasta(tmp_path, oracle, seed, floor_len, window):
    """The file path held to the oracle directly (oracle.annotate_one per record: anno.d:55-110, analysis.d:84-92,108-118):
    reads with N and IUPAC codes, every CIGAR op, random flags, a FASTA with lower-case stretches and N / IUPAC letters
    (analysis.d:63 upper-cases the window) — both inflaters, records in input order."""
    from test_gpu_fuzz import _random_batch
    rng = np.random.default_rng(seed)
    contigs = []
    for k in range(3):
        c = bytearray(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(4000, 9000)))].tobytes())
        for q in rng.integers(0, len(c), size=len(c) // 40):
            c[q] = int(rng.choice(list(b"NNNRYKMacgtn")))
        a = int(rng.integers(0, len(c) - 300))
        c[a:a + 200] = bytes(c[a:a + 200]).lower()
        contigs.append(bytes(c).decode())
    names = ["ctgA", "ctgB", "ctgC"]
    b = _random_batch(rng, contigs, 2500, window)
    qn = ["q%d" % i for i in range(len(b["pos"]))]
    b["qname"] = [x.encode() for x in qn]
    sam, fa, bam = tmp_path / "in.sam", tmp_path / "ref.fa", tmp_path / "in.bam"
    sam.write_text(samutil.batch_to_sam(b, names, [len(c) for c in contigs], qn))
    fa.write_text("".join(">%s\n%s\n" % (n, "\n".join(c[o:o + 70] for o in range(0, len(c), 70))) for n, c in zip(names, contigs)))
    _bam_of(sam, bam)
    G = oracle.GenomeHolder(names, contigs)
    reads, keep = oracle.make_reads(b)
    want = [oracle.annotate_one(G, reads[i], floor_len, window) for i in range(len(qn))]
    assert sum(w["has_tags"] for w in want) >= 20
    for inflate in ("device", "host"):
        p = _run(["annotate", "--timing", "--min-length", str(floor_len), "-w", str(window), "-b", str(bam), str(fa)], {"FADE_BAM_INFLATE": inflate})
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert b"file path on the device" in p.stderr
        _, _, recs = samutil.bam_to_sam_records(p.stdout)
        assert [r["qname"] for r in recs] == qn
        for r, w in zip(recs, want):
            t = r["tags"]
            assert int(t["rs"][1]) == w["rs"], (r["qname"], t["rs"], w)
            if w["has_tags"]:
                assert (t["am"][1], t["as"][1], t["ar"][1], t["ab"][1]) == (w["am"], w["as_"], w["ar"], w["ab"]), r["qname"]
                assert [k for k in r["tag_order"] if k in ("rs", "am", "as", "ar", "ab")] == ["rs", "am", "as", "ar", "ab"]
            else:
                assert "am" not in t and "as" not in t


@pytest.fixture(scope="module")
def big(tmp_path_factory):
    """30,000 reads of C5 (30 % soft-clipped) as a BAM of ~130 BGZF members, records straddling them."""
    from fade_amd import synth
    d = tmp_path_factory.mktemp("bamstream")
    cfg, g, b = synth.make_config("C5", 30000, contig_len=400_000)
    names = ["read%d" % (i // 2) for i in range(len(b["pos"]))]
    b["qname"] = names
    sam, fa, bam = d / "in.sam", d / "ref.fa", d / "in.bam"
    sam.write_text(samutil.batch_to_sam(b, g.names, [int(x) for x in g.lengths], names))
    fa.write_bytes(g.fasta_bytes())
    p = _run(["out", "-b", str(sam)])
    assert p.returncode == 0, p.stderr.decode()
    bam.write_bytes(p.stdout)
    host = _run(["annotate", "--stats", "--timing", "-w", "100", "-b", str(bam), str(fa)], {"FADE_BAM_DEVICE": "0"})
    assert host.returncode == 0, host.stderr.decode()[-2000:]
    return dict(bam=bam, fa=fa, host=host, g=g)


@pytest.mark.parametrize("inflate", ["device", "host"])
@pytest.mark.parametrize("chunk_mb", ["1", "64"])
def test_records_that_straddle_members_and_calls(big, chunk_mb, inflate):
    """Small calls (1 MB: records straddle members AND calls; the tail of one call is carried to the next on the device) and
    one large call; the members inflated by the device, or by this process's pool (fadehip_bam_front_raw)."""
    dev = _run(["annotate", "--stats", "--timing", "-w", "100", "-b", str(big["bam"]), str(big["fa"])], {"FADE_BAM_CHUNK_MB": chunk_mb, "FADE_BAM_INFLATE": inflate})
    assert (b"inflate on 0 host threads" in dev.stderr) == (inflate == "device")
    assert dev.returncode == 0, dev.stderr.decode()[-2000:]
    assert b"file path on the device" in dev.stderr
    assert gzip.decompress(dev.stdout) == gzip.decompress(big["host"].stdout)
    stats = lambda err: [l for l in err.decode().splitlines() if l.startswith(("read count", "Clipped", "% With", "Artifact"))]
    assert stats(dev.stderr) == stats(big["host"].stderr)


@pytest.mark.parametrize("share,chunk_mb", [("2", "1"), ("3", "1"), ("2", "64")])
def test_the_pool_and_the_device_share_the_inflating(big, share, chunk_mb):
    """FADE_BAM_DEVICE_SHARE=n: the pool inflates, but every n-th call's members cross PCIe as they are and are inflated by the
    kernel — calls of both kinds alternate on one stream (fadehip_bam_front / fadehip_bam_front_raw), records straddle them.
    Same bytes as the host pipeline."""
    dev = _run(["annotate", "--stats", "--timing", "-w", "100", "-b", str(big["bam"]), str(big["fa"])],
               {"FADE_BAM_CHUNK_MB": chunk_mb, "FADE_BAM_INFLATE": "host", "FADE_BAM_DEVICE_SHARE": share})
    assert dev.returncode == 0, dev.stderr.decode()[-2000:]
    assert b"file path on the device" in dev.stderr
    assert gzip.decompress(dev.stdout) == gzip.decompress(big["host"].stdout)
    stats = lambda err: [l for l in err.decode().splitlines() if l.startswith(("read count", "Clipped", "% With", "Artifact"))]
    assert stats(dev.stderr) == stats(big["host"].stderr)


@pytest.mark.parametrize("inflate", ["device", "host"])
def test_a_slow_reader_of_the_output_gets_the_same_bytes(big, inflate):
    """The output through a pipe that is drained slowly, in many small calls: the back half runs ahead of the writer, and
    its two staging buffers (include/fadehip.h: a call's bytes stay valid during the next call only) must not be compressed
    into again while the writer still holds them."""
    import time
    e = dict(os.environ, FADE_BAM_CHUNK_MB="1", FADE_BAM_INFLATE=inflate)
    # (the fixture's command line: it is in the header's @PG)
    p = subprocess.Popen([FADE, "annotate", "--stats", "--timing", "-w", "100", "-b", str(big["bam"]), str(big["fa"])], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=e)
    got = bytearray()
    while True:
        piece = p.stdout.read(65536)
        if not piece:
            break
        got += piece
        time.sleep(0.004)
    assert p.wait(timeout=120) == 0
    for k, m in enumerate(_members(bytes(got))):
        assert m[:4] == b"\x1f\x8b\x08\x04", "member %d does not begin with a BGZF header" % k
        body = zlib.decompress(m[18:-8], -15)
        assert struct.unpack("<II", m[-8:]) == (zlib.crc32(body) & 0xffffffff, len(body)), "member %d" % k
    assert gzip.decompress(bytes(got)) == gzip.decompress(big["host"].stdout)


def test_stream_api_one_member_per_call(big):
    """The C ABI itself, fed the smallest pieces it takes: one BGZF member per front call (most records then straddle calls),
    front and back alternating on one thread."""
    raw = big["bam"].read_bytes()
    payload = gzip.decompress(raw)
    # the BAM header: magic, l_text, text, n_ref, (l_name, name, l_ref)*
    assert payload[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", payload, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", payload, at)[0]
    at += 4
    names = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", payload, at)[0]
        names.append(payload[at + 4:at + 4 + ln - 1].decode())
        at += 4 + ln + 4
    hdr_bytes = at
    ms = _members(raw)
    cum, k = 0, 0
    while cum + struct.unpack_from("<I", ms[k], len(ms[k]) - 4)[0] <= hdr_bytes:
        cum += struct.unpack_from("<I", ms[k], len(ms[k]) - 4)[0]
        k += 1
    g = big["g"]
    ctx = fade_amd.Context(device=0)
    try:
        ctx.genome_upload(g.names, g.ascii_contigs())
        st = ctx.bam_stream(names, floor_len=5, window=100, first_record=hdr_bytes - cum)
        body = ms[k:]
        if len(body[-1]) == 28:  # the end-of-file member: an empty payload, passed like any other
            pass
        out = []
        for j, m in enumerate(body):
            st.front(m, last=(j == len(body) - 1))
            out.append(st.back())
        with pytest.raises(fade_amd.FadeHipError):
            st.back()  # nothing pending: an error, not a wait
        totals, n_rec, n_over = st.totals()
        st.close()
    finally:
        ctx.close()
    got = gzip.decompress(b"".join(out))
    want = gzip.decompress(big["host"].stdout)
    # the host run's output starts with its header (with the @PG line the CLI adds); the records behind it must be equal
    l_text_w = struct.unpack_from("<i", want, 4)[0]
    w_at = 8 + l_text_w
    n_ref_w = struct.unpack_from("<i", want, w_at)[0]
    w_at += 4
    for _ in range(n_ref_w):
        w_at += 4 + struct.unpack_from("<i", want, w_at)[0] + 4
    assert got == want[w_at:]
    assert n_rec == 30000 and totals[0] == 30000 and n_over == 0


def test_a_stream_opened_and_prepared_before_the_genome_is_there(big):
    """fadehip_bam_open before fadehip_genome_upload, fadehip_bam_prepare on a thread of its own while the genome goes up (what
    `fade annotate` does while the device is brought up): front before the genome is an error, prepare after the first front is
    one, and the bytes are the ones of a stream that was never prepared."""
    import threading
    raw = big["bam"].read_bytes()
    payload = gzip.decompress(raw)
    l_text = struct.unpack_from("<i", payload, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", payload, at)[0]
    at += 4
    names = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", payload, at)[0]
        names.append(payload[at + 4:at + 4 + ln - 1].decode())
        at += 4 + ln + 4
    hdr_bytes = at
    ms = _members(raw)
    cum, k = 0, 0
    while cum + struct.unpack_from("<I", ms[k], len(ms[k]) - 4)[0] <= hdr_bytes:
        cum += struct.unpack_from("<I", ms[k], len(ms[k]) - 4)[0]
        k += 1
    body = ms[k:]
    calls = [b"".join(body[j:j + 40]) for j in range(0, len(body), 40)]
    g = big["g"]
    outs = []
    for prepared in (True, False):
        ctx = fade_amd.Context(device=0)
        try:
            if prepared:
                st = ctx.bam_stream(names, floor_len=5, window=100, first_record=hdr_bytes - cum)
                with pytest.raises(fade_amd.FadeHipError, match="genome"):
                    st.front(calls[0], last=False)
                st.close()
                st = ctx.bam_stream(names, floor_len=5, window=100, first_record=hdr_bytes - cum)
                err = []

                def prep():
                    try:
                        st.prepare(40 * 65280)
                    except Exception as e:  # noqa: BLE001 (reported on the test's thread)
                        err.append(e)
                t = threading.Thread(target=prep)
                t.start()
                ctx.genome_upload(g.names, g.ascii_contigs())
                t.join()
                assert not err, err
            else:
                ctx.genome_upload(g.names, g.ascii_contigs())
                st = ctx.bam_stream(names, floor_len=5, window=100, first_record=hdr_bytes - cum)
            out = []
            for j, c in enumerate(calls):
                st.front(c, last=(j == len(calls) - 1))
                if j == 0:
                    with pytest.raises(fade_amd.FadeHipError, match="before the first front"):
                        st.prepare(1 << 20)
                out.append(st.back())
            totals, n_rec, _ = st.totals()
            st.close()
            outs.append((gzip.decompress(b"".join(out)), totals, n_rec))
        finally:
            ctx.close()
    assert outs[0] == outs[1] and outs[0][2] == 30000


def test_a_tiny_last_call_whose_last_member_is_mostly_another_readers(big):
    """tail_trim (a lane's last member belongs mostly to the next lane): the inflater still writes the member's whole
    ISIZE, so the call's buffer must be sized from the untrimmed length — here the last call inflates to 64 KB of which a
    few hundred bytes are this stream's (a buffer sized from the trimmed length would be overrun by ~60 KB).  The records
    that come out are exactly the ones in front of the cut, byte for byte those of the whole-file run."""
    from test_gpu_inflate import member
    raw = big["bam"].read_bytes()
    payload = gzip.decompress(raw)
    l_text = struct.unpack_from("<i", payload, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", payload, at)[0]
    at += 4
    names = []
    for _ in range(n_ref):
        ln = struct.unpack_from("<i", payload, at)[0]
        names.append(payload[at + 4:at + 4 + ln - 1].decode())
        at += 4 + ln + 4
    # the first 40 records; the cut lies a few hundred bytes into a member of 0xff00 bytes, alone in the last call
    recs_at = at
    for _ in range(40):
        at += 4 + struct.unpack_from("<I", payload, at)[0]
    mine = payload[recs_at:at]
    assert len(mine) > 0x1000
    first = mine[:len(mine) - 300]                      # call 1: ends inside a record
    last = mine[len(mine) - 300:] + payload[at:at + 0xff00 - 300]  # call 2: 300 bytes of ours, the rest another reader's
    assert len(last) == 0xff00
    g = big["g"]
    ctx = fade_amd.Context(device=0)
    try:
        ctx.genome_upload(g.names, g.ascii_contigs())
        st = ctx.bam_stream(names, floor_len=5, window=100, first_record=0, tail_trim=len(last) - 300)
        out = []
        st.front(member(first, 6), last=False)
        out.append(st.back())
        st.front(member(last, 6), last=True)
        out.append(st.back())
        totals, n_rec, n_over = st.totals()
        st.close()
        # (the same stream again on the warmed-up buffers, and a neighbour allocation checked for stray bytes)
        canary = ctx.bgzf_deflate(bytes(range(256)) * 64)
        assert gzip.decompress(canary) == bytes(range(256)) * 64
    finally:
        ctx.close()
    assert n_rec == 40 and totals[0] == 40
    got = gzip.decompress(b"".join(out))
    want = gzip.decompress(big["host"].stdout)
    l_text_w = struct.unpack_from("<i", want, 4)[0]
    w_at = 8 + l_text_w
    n_ref_w = struct.unpack_from("<i", want, w_at)[0]
    w_at += 4
    for _ in range(n_ref_w):
        w_at += 4 + struct.unpack_from("<i", want, w_at)[0] + 4
    assert got == want[w_at:w_at + len(got)] and len(got) > len(mine)


@pytest.mark.parametrize("inflate", ["device", "host"])
def test_a_full_output_device_fails_the_run_instead_of_hanging_it(big, inflate):
    """stdout = /dev/full: every write fails with ENOSPC.  The writer stage gives up, and the stages in front of it — which
    sit in queues nobody serves any more — must see the end: a message and a non-zero exit, not a hang."""
    with open("/dev/full", "wb") as full:
        p = subprocess.run([FADE, "annotate", "-w", "100", "-b", str(big["bam"]), str(big["fa"])], stdout=full, stderr=subprocess.PIPE, timeout=120,
                           env=dict(os.environ, FADE_BAM_INFLATE=inflate, FADE_BAM_CHUNK_MB="1"))
    assert p.returncode != 0
    assert b"write error on the output stream" in p.stderr, p.stderr.decode()[-1500:]


def test_corrupt_inputs_fail_the_run(tmp_path, big):
    raw = bytearray(big["bam"].read_bytes())
    # a flipped bit in the middle of a member's DEFLATE stream
    ms = _members(bytes(raw))
    at = sum(len(m) for m in ms[:40]) + 30
    raw[at] ^= 0x10
    bad = tmp_path / "bad.bam"
    bad.write_bytes(bytes(raw))
    for mode in ("device", "host"):
        p = _run(["annotate", "-b", "-w", "100", str(bad), str(big["fa"])], {"FADE_BAM_INFLATE": mode})
        assert p.returncode != 0 and b"[E::fade annotate]" in p.stderr, mode
    # a file cut inside a member, and one cut between members but inside a record
    whole = big["bam"].read_bytes()
    cut1 = tmp_path / "cut1.bam"
    cut1.write_bytes(whole[:len(whole) // 2])
    p = _run(["annotate", "-b", "-w", "100", str(cut1), str(big["fa"])])
    assert p.returncode != 0
    cut2 = tmp_path / "cut2.bam"
    cut2.write_bytes(b"".join(ms[:50]))
    p = _run(["annotate", "-b", "-w", "100", str(cut2), str(big["fa"])])
    assert p.returncode != 0 and b"inside a record" in p.stderr
    # a member whose CRC32 and ISIZE fields were zeroed, in a file whose members hold whole records: taken at its word the
    # member would be an empty one and its records would vanish without any framing error
    from test_gpu_inflate import member, EOF_MARK
    payload = gzip.decompress(whole)
    at = 8 + struct.unpack_from("<i", payload, 4)[0]
    n_ref = struct.unpack_from("<i", payload, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 4 + struct.unpack_from("<i", payload, at)[0] + 4
    pieces, recs = [payload[:at]], []
    while at < len(payload) and len(recs) < 4000:
        bs = struct.unpack_from("<I", payload, at)[0]
        recs.append(payload[at:at + 4 + bs])
        at += 4 + bs
    pieces += [b"".join(recs[k:k + 50]) for k in range(0, len(recs), 50)]
    ms2 = [member(x) for x in pieces]
    aligned = tmp_path / "aligned.bam"
    aligned.write_bytes(b"".join(ms2[:9]) + EOF_MARK + b"".join(ms2[9:]) + EOF_MARK)  # (a true empty member in the middle is fine)
    for env in ({"FADE_BAM_INFLATE": "device"}, {"FADE_BAM_INFLATE": "host"}, {"FADE_BAM_DEVICE": "0"}):
        p = _run(["annotate", "-b", "-w", "100", str(aligned), str(big["fa"])], env)
        assert p.returncode == 0, (env, p.stderr.decode()[-800:])
        assert len(samutil.bam_to_sam_records(p.stdout)[2]) == len(recs)
    ms2[20] = ms2[20][:-8] + bytes(8)
    aligned.write_bytes(b"".join(ms2) + EOF_MARK)
    for env in ({"FADE_BAM_INFLATE": "device"}, {"FADE_BAM_INFLATE": "host"}, {"FADE_BAM_DEVICE": "0"}):
        p = _run(["annotate", "-b", "-w", "100", str(aligned), str(big["fa"])], env)
        assert p.returncode != 0, "fifty records vanished without a word (%s)" % env


def test_header_only_bam_and_lanes_take_the_file_path(tmp_path, big):
    """A BAM without records gives header + end-of-file block; `--gpus 2` on a BAM file runs the file path in every lane
    (both on this box's one GPU), records and order as with one lane."""
    raw = big["bam"].read_bytes()
    payload = gzip.decompress(raw)
    l_text = struct.unpack_from("<i", payload, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", payload, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 4 + struct.unpack_from("<i", payload, at)[0] + 4
    hdr_only = tmp_path / "hdr.bam"
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = co.compress(payload[:at]) + co.flush()
    bsize = 18 + len(body) + 8 - 1
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    hdr_only.write_bytes(struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, bsize) + body +
                         struct.pack("<II", zlib.crc32(payload[:at]) & 0xffffffff, at) + eof)
    p = _run(["annotate", "--timing", "-b", "-w", "100", str(hdr_only), str(big["fa"])])
    assert p.returncode == 0, p.stderr.decode()[-1500:]
    assert b"file path on the device" in p.stderr
    out = gzip.decompress(p.stdout)
    assert out[:4] == b"BAM\1" and p.stdout.endswith(eof)
    _, _, recs = samutil.bam_to_sam_records(p.stdout)
    assert recs == []
    one = _run(["annotate", "--stats", "--timing", "-w", "100", "-b", str(big["bam"]), str(big["fa"])])
    two = _run(["annotate", "--stats", "--timing", "-w", "100", "-b", "--gpus", "2", str(big["bam"]), str(big["fa"])], {"FADE_DEVICE_MAP": "0,0"})
    assert one.returncode == 0 and two.returncode == 0, two.stderr.decode()[-2000:]
    assert two.stderr.count(b"file path on the device") == 2 and b"lane 1 of 2" in two.stderr
    # the lanes with the device inflating: a lane's last member is cut by the library (tail_trim)
    three = _run(["annotate", "--stats", "--timing", "-w", "100", "-b", "--gpus", "3", str(big["bam"]), str(big["fa"])],
                 {"FADE_DEVICE_MAP": "0,0,0", "FADE_BAM_INFLATE": "device"})
    assert three.returncode == 0 and three.stderr.count(b"inflate on 0 host threads") == 3, three.stderr.decode()[-2000:]
    assert samutil.bam_to_sam_records(three.stdout)[2] == samutil.bam_to_sam_records(one.stdout)[2]
    _, _, r1 = samutil.bam_to_sam_records(one.stdout)
    _, _, r2 = samutil.bam_to_sam_records(two.stdout)
    assert r1 == r2 and len(r1) == 30000
    stats = lambda err: [l for l in err.decode().splitlines() if l.startswith(("read count", "Clipped", "% With", "Artifact"))]
    assert stats(one.stderr) == stats(two.stderr)


def _rec(name, tid=-1, pos=-1, flag=4, seq=b"", qual=b"", cigar=(), aux=b"", mtid=-1, mpos=-1):
    """One BAM record (block_size included); seq as nt16 codes per base."""
    qn = name + b"\0"
    packed = bytes(((seq[k] << 4) | (seq[k + 1] if k + 1 < len(seq) else 0)) for k in range(0, len(seq), 2))
    body = struct.pack("<iiBBHHHiiii", tid, pos, len(qn), 0, 4680, len(cigar), flag, len(seq), mtid, mpos, 0) + qn + \
        b"".join(struct.pack("<I", c) for c in cigar) + packed + bytes(qual) + aux
    return struct.pack("<I", len(body)) + body


def test_a_record_lookalike_inside_a_tag_does_not_derail_the_framing(tmp_path, big):
    """Framing is speculative per 64 KB segment: a wave assumes the first position that looks like a record.  Here a B:C
    array that straddles a segment boundary holds byte-perfect copies of small records right behind the boundary, so the
    segment's wave starts on a lookalike; the resolving wave must notice (the true chain enters elsewhere), walk the
    segment again, and the output must equal the host pipeline's."""
    raw = big["bam"].read_bytes()
    payload = gzip.decompress(raw)
    l_text = struct.unpack_from("<i", payload, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", payload, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 4 + struct.unpack_from("<i", payload, at)[0] + 4
    header = payload[:at]
    out = bytearray(header)
    k = 0
    small = lambda j: _rec(b"r%07d" % j, seq=bytes([1, 2, 4, 8] * 5), qual=bytes([30] * 20))
    for boundary in (1, 3):
        # small records up to ~300 bytes in front of the boundary, then the straddler
        while len(out) + len(small(k)) < 65536 * boundary - 300:
            out += small(k)
            k += 1
        fake = b"".join(_rec(b"fake%03d" % j, seq=bytes([1, 2] * 8), qual=bytes([20] * 16)) for j in range(12))
        lead = 65536 * boundary - len(out)  # bytes of the straddler in front of the boundary
        # straddler: fixed part 36 + name 9 + aux header 8 (tag, 'B', 'C', count); the array = filler up to the boundary (+ 7), then the lookalikes
        fill = lead - (36 + 9 + 8) + 7
        assert fill > 0
        arr = bytes([0xff] * fill) + fake + bytes([0xff] * 50)
        out += _rec(b"straddle", aux=b"zzBC" + struct.pack("<I", len(arr)) + arr)
        for _ in range(40):
            out += small(k)
            k += 1
    blocks = [bytes(out[o:o + 0xff00]) for o in range(0, len(out), 0xff00)]
    from test_gpu_inflate import member, EOF_MARK
    bam = tmp_path / "lookalike.bam"
    bam.write_bytes(b"".join(member(b) for b in blocks) + EOF_MARK)
    args = ["annotate", "--timing", "-w", "100", "-b", str(bam), str(big["fa"])]
    for mode in ("host", "device"):
        dev = _run(args, {"FADE_BAM_INFLATE": mode, "FADEHIP_BAM_PROF": "1"})
        host = _run(args, {"FADE_BAM_DEVICE": "0"})
        assert dev.returncode == 0 and host.returncode == 0, dev.stderr.decode()[-2000:]
        assert gzip.decompress(dev.stdout) == gzip.decompress(host.stdout)
        walked_again = int(dev.stderr.decode().split("segments walked again")[1].split()[0])
        assert walked_again >= 2, dev.stderr.decode()[-600:]
    _, _, recs = samutil.bam_to_sam_records(dev.stdout)
    assert len(recs) == k + 2 and sum(r["qname"] == "straddle" for r in recs) == 2 and not any(r["qname"].startswith("fake") for r in recs)


@pytest.mark.parametrize("read_len,window", [(700, 300), (150, 17000), (251, 300)])
def test_long_reads_and_wide_windows_through_the_file_path(tmp_path, read_len, window):
    """Reads beyond 512 bases (the thread-per-alignment kernel), windows beyond 32,000 columns (likewise), and a row class
    other than C2's: the file path sizes its launches from what the device finds (l_seq range, longest alignedLength), and its
    output must equal the host pipeline's, whose bounds come from the host's own pass over the records."""
    from fade_amd import synth
    cfg = synth.config("C5")
    cfg.update(read_len=read_len, contig_len=300_000, insert_mu=max(cfg["insert_mu"], read_len + 200))
    g = synth.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
    b = synth.make_reads(g, 1500, 11, **cfg)
    names = ["q%d" % (i // 2) for i in range(len(b["pos"]))]
    b["qname"] = names
    sam, fa, bam = tmp_path / "in.sam", tmp_path / "ref.fa", tmp_path / "in.bam"
    sam.write_text(samutil.batch_to_sam(b, g.names, [int(x) for x in g.lengths], names))
    fa.write_bytes(g.fasta_bytes())
    p = _run(["out", "-b", str(sam)])
    assert p.returncode == 0, p.stderr.decode()
    bam.write_bytes(p.stdout)
    args = ["annotate", "--stats", "--timing", "-w", str(window), "-b", str(bam), str(fa)]
    host = _run(args, {"FADE_BAM_DEVICE": "0"})
    assert host.returncode == 0, host.stderr.decode()[-1500:]
    for mode in ("host", "device"):
        dev = _run(args, {"FADE_BAM_INFLATE": mode})
        assert dev.returncode == 0, dev.stderr.decode()[-1500:]
        assert b"file path on the device" in dev.stderr
        assert gzip.decompress(dev.stdout) == gzip.decompress(host.stdout), mode
    _, _, recs = samutil.bam_to_sam_records(dev.stdout)
    assert len(recs) == 1500 and any("am" in r["tags"] for r in recs)


def test_spliced_reads_through_the_file_path(tmp_path, big):
    """Reads with an N op (alignedLength of tens of thousands): the device finds the longest span itself and sizes the
    long-window launches from it (no per-read list as the host's upload keeps): same bytes as the host pipeline."""
    import re
    # back to SAM text through the CLI, a few CIGARs spliced, to BAM again
    p = _run(["out", str(big["bam"])])
    assert p.returncode == 0, p.stderr.decode()
    lines = p.stdout.decode().splitlines()
    done = 0
    for k, l in enumerate(lines):
        if l.startswith("@"):
            continue
        f = l.split("\t")
        m = re.fullmatch(r"(\d+)M(\d+)S", f[5])
        if m and int(m.group(1)) >= 60 and int(f[3]) < 300000 and done < 24:
            a = int(m.group(1))
            gap = 31000 if done % 2 == 0 else 41000  # windows of ~31,300 (wave kernels) and ~41,300 columns (thread kernel)
            f[5] = "%dM%dN%dM%sS" % (a // 2, gap, a - a // 2, m.group(2))
            lines[k] = "\t".join(f)
            done += 1
    assert done == 24
    sam, bam = tmp_path / "spliced.sam", tmp_path / "spliced.bam"
    sam.write_text("\n".join(lines) + "\n")
    p = _run(["out", "-b", str(sam)])
    assert p.returncode == 0, p.stderr.decode()
    bam.write_bytes(p.stdout)
    args = ["annotate", "--stats", "--timing", "-w", "100", "-b", str(bam), str(big["fa"])]
    host = _run(args, {"FADE_BAM_DEVICE": "0"})
    dev = _run(args)
    assert host.returncode == 0 and dev.returncode == 0, dev.stderr.decode()[-1500:]
    assert b"file path on the device" in dev.stderr
    assert gzip.decompress(dev.stdout) == gzip.decompress(host.stdout)


def test_ubam_output_through_the_file_path(big):
    """`fade annotate -u`: uncompressed BGZF (stored DEFLATE blocks, what htslib writes at level 0) made on the device; the
    bytes inside are the host pipeline's."""
    args = ["annotate", "--stats", "--timing", "-w", "100", "-u", str(big["bam"]), str(big["fa"])]
    dev = _run(args)
    host = _run(args, {"FADE_BAM_DEVICE": "0"})
    assert dev.returncode == 0 and host.returncode == 0, dev.stderr.decode()[-1500:]
    assert b"file path on the device" in dev.stderr and b"file path on the device" not in host.stderr
    assert gzip.decompress(dev.stdout) == gzip.decompress(host.stdout)
    ms = _members(dev.stdout)
    body = [m for m in ms[1:-1]]  # (the header's member comes from the CPU writer, the last is the end-of-file block)
    assert len(body) > 100 and all(m[18] == 1 for m in body)  # BFINAL = 1, BTYPE = 00: stored
    assert all(struct.unpack_from("<H", m, 19)[0] == len(m) - 18 - 5 - 8 for m in body)  # LEN = the payload
    assert len(dev.stdout) > len(gzip.decompress(dev.stdout))
    two = _run(args[:1] + ["--gpus", "2"] + args[1:], {"FADE_DEVICE_MAP": "0,0"})
    assert two.returncode == 0, two.stderr.decode()[-1500:]
    assert samutil.bam_to_sam_records(two.stdout)[2] == samutil.bam_to_sam_records(dev.stdout)[2]


def test_records_larger_than_a_framing_segment(tmp_path, big):
    """Unmapped records of 100,000 and 300,000 bases (150 KB and 450 KB: several 64 KB framing segments without a record
    start, several BGZF members, and — with 1 MB calls — a carry-over of most of a call) between ordinary ones."""
    raw = big["bam"].read_bytes()
    payload = gzip.decompress(raw)
    l_text = struct.unpack_from("<i", payload, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", payload, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 4 + struct.unpack_from("<i", payload, at)[0] + 4
    rng = np.random.default_rng(4)
    out = bytearray(payload[:at])
    k = 0
    small = lambda j: _rec(b"r%07d" % j, seq=bytes([1, 2, 4, 8] * 5), qual=bytes([30] * 20))
    for n_big in (100_000, 300_000, 100_001):
        for _ in range(300):
            out += small(k)
            k += 1
        seq = rng.choice(np.array([1, 2, 4, 8], np.uint8), n_big).tobytes()
        out += _rec(b"giant%d" % n_big, seq=seq, qual=bytes(rng.integers(2, 40, n_big, dtype=np.uint8)))
    for _ in range(100):
        out += small(k)
        k += 1
    from test_gpu_inflate import member, EOF_MARK
    blocks = [bytes(out[o:o + 0xff00]) for o in range(0, len(out), 0xff00)]
    bam = tmp_path / "giants.bam"
    bam.write_bytes(b"".join(member(b, 1) for b in blocks) + EOF_MARK)
    args = ["annotate", "--timing", "-w", "100", "-b", str(bam), str(big["fa"])]
    host = _run(args, {"FADE_BAM_DEVICE": "0"})
    assert host.returncode == 0, host.stderr.decode()[-1500:]
    for env in ({"FADE_BAM_INFLATE": "host"}, {"FADE_BAM_INFLATE": "device"}, {"FADE_BAM_INFLATE": "host", "FADE_BAM_CHUNK_MB": "1"},
                {"FADE_BAM_INFLATE": "device", "FADE_BAM_CHUNK_MB": "1"}):
        dev = _run(args, env)
        assert dev.returncode == 0, (env, dev.stderr.decode()[-1500:])
        assert b"file path on the device" in dev.stderr
        assert gzip.decompress(dev.stdout) == gzip.decompress(host.stdout), env
    _, _, recs = samutil.bam_to_sam_records(dev.stdout)
    assert len(recs) == k + 3 and sorted(len(r["seq"]) for r in recs)[-3:] == [100_000, 100_001, 300_000]


def _random_aux(rng, with_ours):
    """A run of valid aux fields of every type; with_ours: some of rs / am / as / ar / ab among them, in odd types."""
    out = b""
    used = set()
    n = int(rng.integers(0, 7))
    ours = [b"rs", b"am", b"as", b"ar", b"ab"]
    for _ in range(n):
        if with_ours and rng.random() < 0.35:
            tag = ours[int(rng.integers(0, 5))]
        else:
            tag = bytes([int(rng.integers(65, 91)), int(rng.integers(48, 58))])  # e.g. X7: never one of ours
        if tag in used:
            continue
        used.add(tag)
        t = "AcCsSiIfZHB"[int(rng.integers(0, 11))]
        if tag == b"rs" and rng.random() < 0.7:
            t = "cCsSiI"[int(rng.integers(0, 6))]
        if t == "A":
            v = bytes([int(rng.integers(33, 127))])
        elif t in "cC":
            v = bytes([int(rng.integers(0, 128))])
        elif t in "sS":
            v = struct.pack("<H", int(rng.integers(0, 30000)))
        elif t in "iI":
            v = struct.pack("<I", int(rng.integers(0, 2 ** 31 - 1)))
        elif t == "f":
            v = struct.pack("<f", float(rng.integers(-1000, 1000)) / 8)
        elif t == "Z":
            v = bytes(rng.integers(33, 127, int(rng.integers(0, 40)), dtype=np.uint8)) + b"\0"
        elif t == "H":
            v = bytes(rng.choice(np.frombuffer(b"0123456789ABCDEF", np.uint8), 2 * int(rng.integers(0, 10)))) + b"\0"
        else:
            sub = "cCsSiIf"[int(rng.integers(0, 7))]
            cnt = int(rng.integers(0, 20))
            es = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
            v = sub.encode() + struct.pack("<I", cnt) + bytes(rng.integers(0, 256, es * cnt, dtype=np.uint8))
        out += tag + t.encode() + v
    return out


@pytest.mark.parametrize("with_ours", [False, True])
def test_records_with_every_kind_of_tag(tmp_path, big, with_ours):
    """Every aux type in front of where the new tags go (the device walks the aux area to check it and to look for SA and for
    its own tags), and — with_ours — records that already carry rs / am / as / ar / ab in assorted types and positions:
    htslib's update semantics field by field.  Same bytes as the host pipeline."""
    rng = np.random.default_rng(12 + int(with_ours))
    payload = gzip.decompress(big["bam"].read_bytes())
    l_text = struct.unpack_from("<i", payload, 4)[0]
    at = 8 + l_text
    n_ref = struct.unpack_from("<i", payload, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 4 + struct.unpack_from("<i", payload, at)[0] + 4
    out = bytearray(payload[:at])
    n = 0
    while at < len(payload) and n < 12000:
        bs = struct.unpack_from("<I", payload, at)[0]
        body = payload[at + 4:at + 4 + bs] + _random_aux(rng, with_ours)
        out += struct.pack("<I", len(body)) + body
        at += 4 + bs
        n += 1
    from test_gpu_inflate import member, EOF_MARK
    blocks = [bytes(out[o:o + 0xff00]) for o in range(0, len(out), 0xff00)]
    bam = tmp_path / "tags.bam"
    bam.write_bytes(b"".join(member(b, 1) for b in blocks) + EOF_MARK)
    args = ["annotate", "--stats", "--timing", "-w", "100", "-b", str(bam), str(big["fa"])]
    host = _run(args, {"FADE_BAM_DEVICE": "0"})
    dev = _run(args)
    assert host.returncode == 0 and dev.returncode == 0, dev.stderr.decode()[-1500:]
    assert b"file path on the device" in dev.stderr
    assert gzip.decompress(dev.stdout) == gzip.decompress(host.stdout)
    _, _, recs = samutil.bam_to_sam_records(dev.stdout)
    assert len(recs) == n and sum("am" in r["tags"] for r in recs) > 100
    # htslib's update semantics, stated on their own (sam.c bam_aux_update_int / bam_aux_update_str): an integer rs keeps its
    # slot and position and leaves with the UNSIGNED type letter of its size; an rs of another type stays as it is (EINVAL)
    # and no second rs is appended; a string tag of ours that is not 'Z' stays as it is; absent tags are appended, rs as 'C'
    _, _, ins = samutil.bam_to_sam_records(b"".join(member(b, 1) for b in blocks) + EOF_MARK)
    seen = dict(int_kept=0, other_kept=0, appended=0, str_kept=0)
    for ri, ro in zip(ins, recs):
        ti, to = ri["tags"], ro["tags"]
        assert ro["tag_order"].count("rs") == 1
        if "rs" in ti:
            assert ro["tag_order"].index("rs") == ri["tag_order"].index("rs")
            if "rs.bamtype" in ti:
                assert to["rs.bamtype"] == {"c": "C", "C": "C", "s": "S", "S": "S", "i": "I", "I": "I"}[ti["rs.bamtype"]]
                assert 0 <= int(to["rs"][1]) < 64
                seen["int_kept"] += 1
            else:
                assert to["rs"] == ti["rs"]
                seen["other_kept"] += 1
        else:
            assert to["rs.bamtype"] == "C" and ro["tag_order"][len(ri["tag_order"])] == "rs"
            seen["appended"] += 1
        for k in ("am", "as", "ar", "ab"):
            if k in ti and ti[k][0] != "Z":
                assert to[k] == ti[k] and ro["tag_order"].count(k) == 1
                seen["str_kept"] += 1
    if with_ours:
        assert min(seen.values()) > 20, seen
