"""The `fade annotate` host driver (fade_amd/fade) end to end on the GPU: same command line as the
reference (app.d:74-101), SAM / uBAM / BAM on stdout, tags rs, am, as, ar, ab in that order."""
import os
import subprocess

import pytest

import samutil

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FADE = os.path.join(ROOT, "fade_amd", "fade")
GOLD = os.path.join(ROOT, "tests", "golden")


def _expected(tag):
    exp, params = [], {}
    for line in open(os.path.join(GOLD, tag + ".expected.tsv")):
        if line.startswith("#floor_len"):
            params = dict(kv.split("=") for kv in line[1:].split())
        elif not line.startswith("#"):
            f = line.rstrip("\n").split("\t")
            exp.append((f[0], int(f[1]), int(f[2]), f[3], f[4], f[5], f[6]))
    return exp, int(params["floor_len"]), int(params["window"])


def _run(args, **kw):
    return subprocess.run([FADE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, **kw)


def _check_records(recs, exp):
    assert len(recs) == len(exp)
    for r, e in zip(recs, exp):  # this driver keeps input order
        t = r["tags"]
        got = (r["qname"], r["flag"], int(t["rs"][1]), t.get("am", ("Z", ""))[1], t.get("as", ("Z", ""))[1],
               t.get("ar", ("Z", ""))[1], t.get("ab", ("Z", ""))[1])
        assert got == e
        mine = [k for k in r["tag_order"] if k in ("rs", "am", "as", "ar", "ab")]
        assert mine == (["rs", "am", "as", "ar", "ab"] if "am" in t else ["rs"])  # anno.d:94-106


@pytest.mark.parametrize("tag", ["anno_c1", "anno_c2", "anno_c5", "anno_floor0"])
def test_cli_sam_output_matches_golden(tag):
    exp, floor_len, window = _expected(tag)
    p = _run(["annotate", "--min-length", str(floor_len), "-w", str(window), os.path.join(GOLD, tag + ".sam"),
              os.path.join(GOLD, tag + ".fa")])
    assert p.returncode == 0, p.stderr.decode()
    assert b"[W::fade annotate] Output SAM/BAM will not be sorted" in p.stderr
    header, recs = samutil.parse_sam(p.stdout.decode())
    pg = [h for h in header if h.startswith("@PG")][-1]
    assert "ID:fade-annotate" in pg and "PN:fade" in pg and "PP:synth" in pg and "CL:" in pg  # anno.d:25-32
    _check_records(recs, exp)


def test_cli_bam_roundtrip_and_ubam():
    tag = "anno_c2"
    exp, floor_len, window = _expected(tag)
    base = ["annotate", "--min-length", str(floor_len), "-w%d" % window]
    sam, fa = os.path.join(GOLD, tag + ".sam"), os.path.join(GOLD, tag + ".fa")
    pb = _run(base + ["-b", sam, fa])
    pu = _run(base + ["-u", sam, fa])
    assert pb.returncode == 0 and pu.returncode == 0, pb.stderr.decode() + pu.stderr.decode()
    assert pb.stdout[-28:] == pu.stdout[-28:] and pb.stdout[-28:-26] == b"\x1f\x8b"  # BGZF EOF block
    assert len(pu.stdout) > len(pb.stdout)
    tb, nb, rb = samutil.bam_to_sam_records(pb.stdout)
    tu, nu, ru = samutil.bam_to_sam_records(pu.stdout)
    assert rb == ru and nb == nu
    _check_records(rb, exp)
    assert all(r["tags"]["rs.bamtype"] == "C" for r in rb)  # bam_aux_update_int of a ubyte
    # BAM in -> SAM out: annotating the annotated BAM again replaces the tags in place, same values
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fade_cli_rt.bam")
    open(tmp, "wb").write(pb.stdout)
    p2 = _run(base + [tmp, fa])
    assert p2.returncode == 0, p2.stderr.decode()
    header, r2 = samutil.parse_sam(p2.stdout.decode())
    _check_records(r2, exp)
    assert sum(1 for h in header if h.startswith("@PG") and "ID:fade-annotate" in h) == 2


def test_cli_flag_errors_and_help():
    assert _run(["annotate", "-b", "-u", "x.bam", "y.fa"]).returncode == 1  # app.d:94-99
    p = _run(["annotate"])
    assert p.returncode == 0 and b"usage: fade annotate" in p.stderr  # app.d:84-89
    assert _run(["annotate", "-h"]).returncode == 0
    assert _run(["bogus"]).returncode == 1  # app.d:218-220
    assert _run([]).returncode == 0
    p = _run(["annotate", "--stats", "--batch", "100", os.path.join(GOLD, "anno_c1.sam"), os.path.join(GOLD, "anno_c1.fa")])
    assert p.returncode == 0 and b"read count:\t600" in p.stderr


@pytest.mark.parametrize("slots", ["1", "2", "switch"])
def test_cli_gpus_2_on_one_device_equals_gpus_1(tmp_path, slots):
    """`fade annotate --gpus 2`: batches dealt round-robin to two fadehip contexts, one or two slots each (FADE_SLOTS:
    one batch in flight per device is the default, two the double-buffered form), all driven by one asynchronous host thread.  On a one-GPU box FADE_DEVICE_MAP=0,0 puts both contexts on device 0 (their stats are
    then summed on the host: RCCL takes one rank per device).  Output identical to --gpus 1, --stats included."""
    from fade_amd import synth
    cfg, g, b = synth.make_config("C5", 6000, contig_len=150_000)
    names = [b"r%d" % (i // 2) for i in range(6000)]
    b["qname"] = names
    sam = tmp_path / "in.sam"
    fa = tmp_path / "ref.fa"
    sam.write_text(samutil.batch_to_sam(b, g.names, [int(x) for x in g.lengths], names))
    fa.write_bytes(g.fasta_bytes())
    base = ["annotate", "--stats", "--batch", "500", "--min-length", "5", "-w", "100"]
    one = _run(base + [str(sam), str(fa)])
    # "switch": the driver starts with one batch in flight per device and takes the second slot into use in mid-run
    # (FADE_SLOT_WAIT=0: as if the host had waited for the device)
    env = dict(os.environ, FADE_DEVICE_MAP="0,0", **({"FADE_SLOT_WAIT": "0"} if slots == "switch" else {"FADE_SLOTS": slots}))
    two = _run(base + ["--gpus", "2", "--timing", str(sam), str(fa)], env=env)
    assert (b"two batches in flight per device from batch" in two.stderr) == (slots == "switch")
    assert one.returncode == 0 and two.returncode == 0, one.stderr.decode() + two.stderr.decode()
    strip_pg = lambda out: [l for l in out.decode().splitlines() if not l.startswith("@PG\tID:fade-annotate")]
    assert strip_pg(one.stdout) == strip_pg(two.stdout) and len(strip_pg(one.stdout)) > 6000
    stats = lambda err: "\n".join(l for l in err.decode().split("read count:")[1].splitlines() if not l.startswith("[timing]")) + "\n"
    assert stats(one.stderr) == stats(two.stderr) and stats(one.stderr).startswith("\t6000\n")


@pytest.mark.parametrize("tag", ["anno_c1", "anno_c2", "anno_c5"])
def test_cli_chain_annotate_extract_out(tmp_path, tag):
    """The path and its consumers chained on device-produced tags: `fade annotate -b` (GPU) -> `fade extract`
    (source/remap.d:29-85) and -> `fade out` / `fade out -c` (source/filter.d:15-91,190-266), each checked line by
    line against its pure-Python restatement (oracle/pyremap.py, oracle/pyfilter.py) run on the annotate output."""
    from oracle import pyfilter, pyremap
    exp, floor_len, window = _expected(tag)
    sam, fa = os.path.join(GOLD, tag + ".sam"), os.path.join(GOLD, tag + ".fa")
    pa = _run(["annotate", "--min-length", str(floor_len), "-w", str(window), sam, fa])
    assert pa.returncode == 0, pa.stderr.decode()
    anno = tmp_path / "anno.sam"
    anno.write_bytes(pa.stdout)
    header, recs = samutil.parse_sam(pa.stdout.decode())
    _check_records(recs, exp)
    names = [h.split("\t")[1][3:] for h in header if h.startswith("@SQ")]
    contig0 = names[0]
    # extract
    pe = _run(["extract", str(anno)])
    assert pe.returncode == 0, pe.stderr.decode()
    got = [l for l in pe.stdout.decode().splitlines() if not l.startswith("@")]
    assert got == pyremap.extract_records(recs, names) and len(got) >= 10
    # out, name-sorted branch (the golden inputs keep mates adjacent) and -c
    for args, clip in (([], False), (["-c"], True)):
        po = _run(["out"] + args + [str(anno)])
        assert po.returncode == 0, po.stderr.decode()
        lines = [l for l in po.stdout.decode().splitlines() if not l.startswith("@")]
        want, stats = pyfilter.fade_out(recs, contig0, clip=clip)
        assert lines == want and po.stderr.decode().endswith(stats)
    # BAM all the way on pipes: annotate -b | out -b, and annotate -b | extract: same records as through the SAM files
    cmd = "%s annotate -b --min-length %d -w %d %s %s | %s out - " % (FADE, floor_len, window, sam, fa, FADE)
    pipe = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert pipe.returncode == 0, pipe.stderr.decode()
    strip = lambda text: [l for l in text.splitlines() if not l.startswith("@")]
    assert strip(pipe.stdout.decode()) == pyfilter.fade_out(recs, contig0, clip=False)[0]
    cmd = "%s annotate -b --min-length %d -w %d %s %s | %s extract -" % (FADE, floor_len, window, sam, fa, FADE)
    pipe = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert pipe.returncode == 0 and strip(pipe.stdout.decode()) == got, pipe.stderr.decode()


@pytest.mark.parametrize("tag", ["anno_c1", "anno_c2", "anno_c5", "anno_floor0"])
def test_cli_bam_input_in_place_records(tmp_path, tag):
    """BAM input takes the in-place path: records are framed in the inflated bytes (no per-record copy), new tags go
    out as a suffix behind the record.  Same records as from SAM input, to SAM and to BAM; a second pass over the
    annotated BAM (tags present: records rebuilt with htslib's update-in-place semantics) changes nothing."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "-s", "sam2bam"])
    exp, floor_len, window = _expected(tag)
    sam, fa = os.path.join(GOLD, tag + ".sam"), os.path.join(GOLD, tag + ".fa")
    bam = tmp_path / "in.bam"
    with open(bam, "wb") as fo:
        subprocess.check_call([os.path.join(ROOT, "tools", "sam2bam"), sam], stdout=fo)
    base = ["annotate", "--min-length", str(floor_len), "-w", str(window), "--batch", "97"]  # several blocks, records straddling them
    from_sam = _run(base + [sam, fa])
    from_bam = _run(base + [str(bam), fa])
    assert from_sam.returncode == 0 and from_bam.returncode == 0, from_bam.stderr.decode()
    strip = lambda out: [l for l in out.decode().splitlines() if not l.startswith("@PG\tID:fade-annotate")]
    assert strip(from_bam.stdout) == strip(from_sam.stdout)
    _check_records(samutil.parse_sam(from_bam.stdout.decode())[1], exp)
    to_bam = _run(base + ["-b", str(bam), fa])
    assert to_bam.returncode == 0, to_bam.stderr.decode()
    _, _, rb = samutil.bam_to_sam_records(to_bam.stdout)
    _check_records(rb, exp)
    assert all(r["tags"]["rs.bamtype"] == "C" for r in rb)
    again = tmp_path / "anno.bam"
    again.write_bytes(to_bam.stdout)
    second = _run(base + ["-b", str(again), fa])
    assert second.returncode == 0, second.stderr.decode()
    _, _, rb2 = samutil.bam_to_sam_records(second.stdout)
    assert rb2 == rb


def test_cli_input_without_soft_clips(tmp_path):
    """Nothing to send to the device (anno.d:61-65 settles every record): every record gets rs = 0, the stats still count
    the reads, and an input with no records at all gives a header-only output."""
    from fade_amd import synth
    cfg = synth.config("C2")
    cfg.update(contig_len=100_000, p_sc=0.0)
    g = synth.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
    b = synth.make_reads(g, 700, 3, **cfg)
    sam, fa, empty = tmp_path / "in.sam", tmp_path / "ref.fa", tmp_path / "empty.sam"
    text = samutil.batch_to_sam(b, g.names, [int(x) for x in g.lengths])
    sam.write_text(text)
    empty.write_text("".join(l + "\n" for l in text.splitlines() if l.startswith("@")))
    fa.write_bytes(g.fasta_bytes())
    p = _run(["annotate", "--stats", "--batch", "300", str(sam), str(fa)])
    assert p.returncode == 0, p.stderr.decode()
    header, recs = samutil.parse_sam(p.stdout.decode())
    assert len(recs) == 700 and all(r["tags"]["rs"][1] == "0" and "am" not in r["tags"] for r in recs)
    assert b"read count:\t700\nClipped %:\t0\n" in p.stderr
    for args in ([], ["-b"]):
        q = _run(["annotate"] + args + [str(empty), str(fa)])
        assert q.returncode == 0, q.stderr.decode()
    assert not [l for l in _run(["annotate", str(empty), str(fa)]).stdout.decode().splitlines() if not l.startswith("@")]


def test_cli_reads_standard_input():
    """`fade annotate - ref.fa`: SAM text and BAM bytes from a pipe (htslib's "-"), same records as from the files."""
    tag = "anno_c5"
    exp, floor_len, window = _expected(tag)
    base = ["annotate", "--min-length", str(floor_len), "-w", str(window)]
    sam, fa = os.path.join(GOLD, tag + ".sam"), os.path.join(GOLD, tag + ".fa")
    from_file = _run(base + [sam, fa])
    assert from_file.returncode == 0, from_file.stderr.decode()
    piped = _run(base + ["-", fa], input=open(sam, "rb").read())
    assert piped.returncode == 0, piped.stderr.decode()
    strip_pg = lambda out: [l for l in out.decode().splitlines() if not l.startswith("@PG\tID:fade-annotate")]
    assert strip_pg(piped.stdout) == strip_pg(from_file.stdout)
    bam = _run(base + ["-b", sam, fa])
    assert bam.returncode == 0
    piped_bam = _run(base + ["-", fa], input=bam.stdout)  # annotated BAM in through the pipe: tags replaced in place
    assert piped_bam.returncode == 0, piped_bam.stderr.decode()
    _, recs = samutil.parse_sam(piped_bam.stdout.decode())
    _check_records(recs, exp)


@pytest.mark.parametrize("fmt", ["-b", "-u", ""])
def test_cli_lanes_one_process_per_gpu_on_disjoint_ranges(tmp_path, fmt):
    """`fade annotate --gpus 2` on a BAM file: two lanes, each a process with its own reader (a BGZF virtual-offset range
    of the input), device context and writer; the parent cuts the file and concatenates the lanes' BGZF blocks.  Both lanes
    sit on this box's one GPU (FADE_DEVICE_MAP=0,0: RCCL wants distinct devices, the parent sums the stats).  The records,
    their order and the stats must be those of the one-process run; a lane's record chain has to end exactly where the
    next lane's guessed start is, so a wrong cut cannot go unnoticed."""
    import gzip
    from fade_amd import synth
    cfg, g, b = synth.make_config("C5", 30000, contig_len=400_000)
    names = ["read%d" % (i // 2) for i in range(len(b["pos"]))]
    b["qname"] = names
    sam = tmp_path / "in.sam"
    fa = tmp_path / "ref.fa"
    sam.write_text(samutil.batch_to_sam(b, g.names, [int(x) for x in g.lengths], names))
    fa.write_bytes(g.fasta_bytes())
    bam = tmp_path / "in.bam"
    p = _run(["out", "-b", str(sam)])  # (no rs tags yet: `out` passes every record through; a BAM of ~130 BGZF blocks)
    assert p.returncode == 0, p.stderr.decode()
    bam.write_bytes(p.stdout)
    base = ["annotate", "--stats", "--timing", "--batch", "4096", "-w", "100"] + ([fmt] if fmt else [])
    one = _run(base + [str(bam), str(fa)])
    two = _run(base + ["--gpus", "2", str(bam), str(fa)], env=dict(os.environ, FADE_DEVICE_MAP="0,0"))
    three = _run(base + ["--gpus", "3", str(bam), str(fa)], env=dict(os.environ, FADE_DEVICE_MAP="0,0,0"))
    assert one.returncode == 0 and two.returncode == 0 and three.returncode == 0, one.stderr.decode()[-500:] + two.stderr.decode()[-1500:]
    assert b"lane 1 of 2" in two.stderr and b"lane 2 of 3" in three.stderr and b"lane 1 of" not in one.stderr

    def records(out):
        if fmt == "":
            return [l for l in out.decode().splitlines() if not l.startswith("@PG\tID:fade-annotate")]
        raw = gzip.decompress(out)
        l_text = int.from_bytes(raw[4:8], "little")
        text = raw[8:8 + l_text].decode()
        return [l for l in text.splitlines() if not l.startswith("@PG\tID:fade-annotate")], raw[8 + l_text:]

    assert records(one.stdout) == records(two.stdout) == records(three.stdout)
    stats = lambda err: [l for l in err.decode().split("read count:")[1].splitlines() if not l.startswith("[timing]") and l][:7]
    assert stats(one.stderr) == stats(two.stderr) == stats(three.stderr) and stats(one.stderr)[0] == "\t30000"
    if fmt:
        assert two.stdout.endswith(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))


@pytest.fixture(scope="module")
def lanes_bam(tmp_path_factory):
    """800,000 reads of C2 as a BAM file (tools/synthgen: the bench's generator), and the one-process run over it."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import synthgen as sg
    from fade_amd import synth
    d = tmp_path_factory.mktemp("lanes")
    cfg = synth.config("C2")
    cfg["contig_len"] = 2_000_000
    g = sg.Genome(cfg["n_contigs"], cfg["contig_len"], 42)
    bam, fa = str(d / "in.bam"), str(d / "ref.fa")
    g.write_fasta(fa)
    w = sg.BamWriter(bam, g)
    for k in range(4):
        w.write(sg.make_reads(g, 200_000, 100 + k, cfg), k * 100_000)
    w.close()
    one_out = str(d / "one.bam")
    with open(one_out, "wb") as fo:
        one = subprocess.run([FADE, "annotate", "--stats", "-w", "100", "-b", bam, fa], stdout=fo, stderr=subprocess.PIPE, timeout=600)
    assert one.returncode == 0, one.stderr.decode()[-1500:]
    return dict(dir=d, bam=bam, fa=fa, one=one_out, one_err=one.stderr)


def _bam_payload(path_or_bytes):
    import gzip
    raw = gzip.decompress(open(path_or_bytes, "rb").read() if isinstance(path_or_bytes, str) else path_or_bytes)
    l_text = int.from_bytes(raw[4:8], "little")
    text = [l for l in raw[8:8 + l_text].decode().splitlines() if not l.startswith("@PG\tID:fade-annotate")]
    return text, raw[8 + l_text:]


def _stats_lines(err):
    return [l for l in err.decode().split("read count:")[1].splitlines() if not l.startswith("[timing]") and l][:7]


@pytest.mark.parametrize("mode", ["placed", "stream", "pipe"])
def test_cli_eight_lanes_merge_while_they_run(lanes_bam, mode):
    """`fade annotate --gpus 8 -b` (BASELINE config 4 rehearsed on one device: FADE_DEVICE_MAP puts every lane
    on GPU 0, FADE_LANES_LIVE=4 keeps within the box's process limit).  The parent forwards the lanes' outputs WHILE they
    run: into their final place when stdout is a file (placed), one after the other when it is a pipe (pipe; `stream` asks
    for that order into a file).  Records and their order = the one-process run's; --timing reports the merge behind the
    last lane, which must be a small part of the run (the serial tail this replaces was a copy of everything)."""
    d = lanes_bam["dir"]
    env = dict(os.environ, FADE_DEVICE_MAP="0,0,0,0,0,0,0,0", FADE_LANES_LIVE="4")
    if mode == "stream":
        env["FADE_LANES_MERGE"] = "stream"
    args = [FADE, "annotate", "--stats", "--timing", "--gpus", "8", "-w", "100", "-b", lanes_bam["bam"], lanes_bam["fa"]]
    if mode == "pipe":
        p = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        out = p.stdout
    else:
        path = str(d / ("eight_%s.bam" % mode))
        with open(path, "wb") as fo:
            p = subprocess.run(args, stdout=fo, stderr=subprocess.PIPE, timeout=900, env=env)
        out = open(path, "rb").read()
    assert p.returncode == 0, p.stderr.decode()[-2500:]
    err = p.stderr.decode()
    assert "lane 7 of 8" in err
    assert ("copied into their final place while the lanes ran" in err) == (mode == "placed")
    assert ("forwarded in order while the lanes ran" in err) == (mode != "placed")
    assert _bam_payload(out) == _bam_payload(lanes_bam["one"])
    assert out.endswith(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
    assert _stats_lines(p.stderr) == _stats_lines(lanes_bam["one_err"]) and _stats_lines(p.stderr)[0] == "\t800000"
    line = [l for l in err.splitlines() if l.startswith("[timing] lanes:")][0]
    last_end = float(line.split("the last lane ended after ")[1].split(" s")[0])
    tail = float(line.split("merge behind the lanes: ")[1].split(" s")[0])
    assert tail < max(0.25, 0.5 * last_end), line
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "lanes_merge_%s.txt" % mode), "w") as f:
        f.write("\n".join(l for l in err.splitlines() if l.startswith("[timing] lane")) + "\n")


def test_cli_out_shards_every_lane_a_complete_file(lanes_bam):
    """`--gpus 6 --out-shards PREFIX`: every lane writes PREFIX.<k>.bam, a complete BAM (header with the @PG line, the lane's
    records, the end-of-file block); nothing goes to stdout, nothing is merged.  The shards' records, in lane order, are the
    one-process run's."""
    d = lanes_bam["dir"]
    prefix = str(d / "shard")
    p = subprocess.run([FADE, "annotate", "--stats", "--timing", "--gpus", "6", "--out-shards", prefix, "-w", "100", "-b", lanes_bam["bam"], lanes_bam["fa"]],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=dict(os.environ, FADE_DEVICE_MAP="0,0,0,0,0,0", FADE_LANES_LIVE="3"))
    assert p.returncode == 0, p.stderr.decode()[-2500:]
    assert p.stdout == b"" and b"a complete file per lane, nothing merged" in p.stderr
    text1, recs1 = _bam_payload(lanes_bam["one"])
    body = b""
    n = 0
    for k in range(6):
        data = open("%s.%d.bam" % (prefix, k), "rb").read()
        assert data.endswith(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
        if k in (0, 5):  # (decoding 270,000 records in Python twice is enough)
            text, names, recs = samutil.bam_to_sam_records(data)
            assert any(l.startswith("@PG\tID:fade-annotate") for l in text.splitlines()) and len(recs) > 50_000
            assert all("rs" in r["tags"] for r in recs[:2000])
        t, b = _bam_payload(data)
        assert t == text1
        # behind l_text + text come n_ref and the references: the same in every shard; the records follow
        import struct
        at = 0
        n_ref = struct.unpack_from("<i", b, at)[0]
        at += 4
        for _ in range(n_ref):
            at += 4 + struct.unpack_from("<i", b, at)[0] + 4
        if k == 0:
            body += b[:at]
        body += b[at:]
        n += 1
    assert body == recs1
    assert _stats_lines(p.stderr) == _stats_lines(lanes_bam["one_err"])


def test_cli_a_failing_lane_ends_the_run(lanes_bam):
    """One lane that cannot work (its device does not exist) fails; the parent must notice whichever lane ends first, stop
    the others and report — not wait for lanes in order, and not leave lanes waiting for a dead peer."""
    p = subprocess.run([FADE, "annotate", "--gpus", "3", "-w", "100", "-b", lanes_bam["bam"], lanes_bam["fa"]], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300, env=dict(os.environ, FADE_DEVICE_MAP="0,0,63"))
    assert p.returncode != 0
    assert b"lane 2 of 3 failed" in p.stderr, p.stderr.decode()[-1500:]
    left = [f for f in os.listdir(os.environ.get("TMPDIR", "/tmp")) if f.startswith("fade_lanes_")]
    assert left == [], left
