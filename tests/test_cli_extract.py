"""`fade extract` (source/remap.d:11-87): host-only consumer of the rs/am tags.
Runs without a GPU.  Input = golden SAM + the golden expected tags; checked against the pure-Python
restatement oracle/pyremap.py and against the FASTA (every '=' run of an extracted record must match the
reference at the am position — the am grammar validated end to end)."""
import os
import subprocess

import pytest

import samutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FADE = os.path.join(ROOT, "fade_amd", "fade")
GOLD = os.path.join(ROOT, "tests", "golden")


def _annotated_sam(tag):
    """golden input SAM with the golden expected tags appended (what `fade annotate` writes)."""
    lines = open(os.path.join(GOLD, tag + ".sam")).read().splitlines()
    exp = [l.rstrip("\n").split("\t") for l in open(os.path.join(GOLD, tag + ".expected.tsv")) if not l.startswith("#")]
    out, k = [], 0
    for l in lines:
        if l.startswith("@"):
            out.append(l)
            continue
        e = exp[k]
        k += 1
        l += "\trs:i:%s" % e[2]
        if e[3]:
            l += "\tam:Z:%s\tas:Z:%s\tar:Z:%s\tab:Z:%s" % (e[3], e[4], e[5], e[6])
        out.append(l)
    return "\n".join(out) + "\n"


@pytest.mark.parametrize("tag", ["anno_c1", "anno_c2", "anno_c5"])
def test_extract_matches_restatement_and_reference(tmp_path, tag):
    import __graft_entry__ as ge
    ge.build()
    from oracle import pyremap
    text = _annotated_sam(tag)
    src = tmp_path / "anno.sam"
    src.write_text(text)
    p = subprocess.run([FADE, "extract", str(src)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    assert b"[W::fade extract] Output SAM/BAM will not be sorted" in p.stderr
    header, got = samutil.parse_sam(p.stdout.decode())
    assert [h for h in header if h.startswith("@PG")][-1].startswith("@PG\tID:fade-extract\tPN:fade")
    in_header, in_recs = samutil.parse_sam(text)
    names, seqs = samutil.read_fasta(open(os.path.join(GOLD, tag + ".fa")).read())
    exp_lines = pyremap.extract_records(in_recs, names)
    got_lines = [l for l in p.stdout.decode().splitlines() if not l.startswith("@")]
    assert got_lines == exp_lines and len(got_lines) >= 10
    # every '=' run matches the (upper-cased) reference, every 'X' run differs base by base
    for r in got:
        ref = seqs[names.index(r["rname"])].upper()
        q, j, num = 0, r["pos"], ""
        for ch in r["cigar"]:
            if ch.isdigit():
                num += ch
                continue
            n = int(num)
            num = ""
            if ch == "=":
                assert r["seq"][q:q + n] == ref[j:j + n]
            if ch == "X":
                assert all(a != b for a, b in zip(r["seq"][q:q + n], ref[j:j + n]))
            if ch in "=XIS":
                q += n
            if ch in "=XD":
                j += n
        assert q == len(r["seq"])
    # BAM output decodes to the same records
    pb = subprocess.run([FADE, "extract", "-b", str(src)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert pb.returncode == 0
    _, _, recs_b = samutil.bam_to_sam_records(pb.stdout)
    assert [(r["qname"], r["flag"], r["rname"], r["pos"], r["cigar"], r["seq"], r["qual"]) for r in recs_b] == \
           [(r["qname"], r["flag"], r["rname"], r["pos"], r["cigar"], r["seq"], r["qual"]) for r in got]


def test_extract_cli_surface():
    assert subprocess.run([FADE, "extract"], stderr=subprocess.PIPE).returncode == 0          # help, app.d:136-141
    assert subprocess.run([FADE, "extract", "-b", "-u", "x"], stderr=subprocess.PIPE).returncode == 1  # app.d:146-151
    p = subprocess.run([FADE, "stats", "x.bam"], stderr=subprocess.PIPE)
    assert p.returncode == 1 and b"outside the MI355X annotate hot path" in p.stderr
