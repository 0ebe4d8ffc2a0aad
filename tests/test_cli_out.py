"""`fade out` (source/filter.d): host-only consumer of the rs/am tags; runs without a GPU.
The CLI is compared line by line with the pure-Python restatement oracle/pyfilter.py on annotated golden
SAMs: the name-sorted branch, the unsorted branch, and the -c hard-clipping branch, plus the stats summary."""
import os
import random
import subprocess

import pytest

import samutil
from test_cli_extract import _annotated_sam

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FADE = os.path.join(ROOT, "fade_amd", "fade")


def _run(args, path):
    p = subprocess.run([FADE, "out"] + args + [path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    lines = [l for l in p.stdout.decode().splitlines() if not l.startswith("@")]
    return lines, p.stderr.decode(), p.stdout.decode()


@pytest.mark.parametrize("tag", ["anno_c1", "anno_c2", "anno_c5"])
def test_out_three_branches(tmp_path, tag):
    import __graft_entry__ as ge
    ge.build()
    from oracle import pyfilter
    text = _annotated_sam(tag)
    header, recs = samutil.parse_sam(text)
    contig0 = [h for h in header if h.startswith("@SQ")][0].split("\t")[1][3:]
    # 1. as generated: mates are adjacent and names increase numerically -> "looks name-sorted"
    src = tmp_path / "sorted.sam"
    src.write_text(text)
    lines, err, out = _run([], str(src))
    exp, stats = pyfilter.fade_out(recs, contig0, clip=False)
    assert "Output looks name-sorted" in err and err.endswith(stats)
    assert lines == exp and 0 < len(lines) < len(recs)
    assert "@PG\tID:fade-extract\tPN:fade" in out  # filter.d:173-180 reuses the extract ID
    # mates of artifact reads are ejected too
    art_names = {r["qname"] for r in recs if int(r["tags"]["rs"][1]) & 6}
    assert not any(l.split("\t")[0] in art_names for l in lines)
    # 2. shuffled: unsorted branch, only the artifact reads themselves go
    rnd = random.Random(3)
    body = [l for l in text.splitlines() if not l.startswith("@")]
    rnd.shuffle(body)
    shuffled = "\n".join([l for l in text.splitlines() if l.startswith("@")] + body) + "\n"
    src2 = tmp_path / "shuffled.sam"
    src2.write_text(shuffled)
    _, recs2 = samutil.parse_sam(shuffled)
    lines2, err2, _ = _run([], str(src2))
    exp2, stats2 = pyfilter.fade_out(recs2, contig0, clip=False)
    assert "doesn't look name-sorted" in err2 and err2.endswith(stats2)
    assert lines2 == exp2 and len(lines2) == len(recs) - len([r for r in recs if int(r["tags"]["rs"][1]) & 6])
    # 3. -c: every record is written, artifact reads hard-clipped
    lines3, err3, _ = _run(["-c"], str(src))
    exp3, stats3 = pyfilter.fade_out(recs, contig0, clip=True)
    assert "Using the -c flag" in err3 and err3.endswith(stats3)
    assert lines3 == exp3 and len(lines3) == len(recs)
    n_clipped = sum(1 for a, b in zip(lines3, [l for l in text.splitlines() if not l.startswith("@")]) if a != b)
    assert n_clipped == len([r for r in recs if int(r["tags"]["rs"][1]) & 6])
    for l in lines3:  # query-consuming CIGAR ops still add up to the sequence length
        f = l.split("\t")
        if f[5] != "*":
            import re
            assert sum(int(n) for n, c in re.findall(r"(\d+)([MIDNSHP=X])", f[5]) if c in "MIS=X") == len(f[9])
    # BAM output of the clip branch decodes to the same records
    pb = subprocess.run([FADE, "out", "-c", "-b", str(src)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert pb.returncode == 0
    _, _, rb = samutil.bam_to_sam_records(pb.stdout)
    assert [(r["qname"], r["flag"], r["pos"], r["cigar"], r["seq"]) for r in rb] == \
           [(l.split("\t")[0], int(l.split("\t")[1]), int(l.split("\t")[3]) - 1, l.split("\t")[5], l.split("\t")[9]) for l in lines3]


def test_numerically_aware_comparison():
    from oracle import pyfilter
    c = pyfilter.numerically_aware_cmp
    assert c("r9", "r10") < 0 and c("r10", "r9") > 0 and c("r10", "r10") == 0
    assert c("a1b2", "a1b10") < 0 and c("abc", "abd") < 0 and c("ab", "abc") < 0


def test_out_cli_surface():
    assert subprocess.run([FADE, "out"], stderr=subprocess.PIPE).returncode == 0          # app.d:109-114
    assert subprocess.run([FADE, "out", "-b", "-u", "x"], stderr=subprocess.PIPE).returncode == 1  # app.d:120-124
    # an option the subcommand does not declare is a getopt error, as in the reference
    assert subprocess.run([FADE, "annotate", "-c", "x", "y"], stderr=subprocess.PIPE).returncode == 1
    assert subprocess.run([FADE, "out", "-w", "5", "x"], stderr=subprocess.PIPE).returncode == 1
    assert subprocess.run([FADE, "extract", "--min-length", "5", "x"], stderr=subprocess.PIPE).returncode == 1
