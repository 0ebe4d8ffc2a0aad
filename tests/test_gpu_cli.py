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
