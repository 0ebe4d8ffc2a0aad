"""One-off: `fade annotate -b` through the file path with odd thread counts (1, 2, 3, 7, 8, 9, 32) and both inflate modes:
same records as the host pipeline every time (the header's @PG differs by the command line).  GPU box."""
import gzip, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synthgen as sg
from fade_amd import synth
from stream_fuzz import header_len
n = 600_000
cfg = synth.config("C5")
g = sg.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
g.write_fasta("/tmp/tc.fa")
w = sg.BamWriter("/tmp/tc.bam", g)
w.write(sg.make_reads(g, n, 5, cfg), 0)
w.close()
fade = os.path.join(ROOT, "fade_amd", "fade")
def run(t, env):
    p = subprocess.run([fade, "annotate", "-t", str(t), "-w", "100", "-b", "/tmp/tc.bam", "/tmp/tc.fa"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env), timeout=200)
    assert p.returncode == 0, (t, env, p.stderr.decode()[-600:])
    d = gzip.decompress(p.stdout)
    return d[header_len(d):]
want = run(16, {"FADE_BAM_DEVICE": "0"})
for t in (1, 2, 3, 7, 8, 9, 32):
    for env in ({"FADE_BAM_INFLATE": "host"}, {"FADE_BAM_INFLATE": "device"}, {"FADE_BAM_INFLATE": "host", "FADE_BAM_CHUNK_MB": "3"}, {}):
        assert run(t, env) == want, (t, env)
    print("-t %d ok" % t, flush=True)
print("thread counts 1, 2, 3, 7, 8, 9, 32 x four settings: the same %d bytes of records as the host pipeline" % len(want))
