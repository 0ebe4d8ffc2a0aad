"""A BAM whose members are zlib's (level 6, as htslib writes them: many matches, dynamic blocks) through the file path, both
inflate modes, against the host pipeline.  python tools/zlib_bam_check.py [n_reads]  (GPU box)"""
import gzip, os, struct, subprocess, sys, time, zlib
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synthgen as sg
from fade_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
cfg = synth.config("C2")
g = sg.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
g.write_fasta("/tmp/zb.fa")
w = sg.BamWriter("/tmp/zb_src.bam", g)
w.write(sg.make_reads(g, n, 5, cfg), 0)
w.close()
payload = gzip.decompress(open("/tmp/zb_src.bam", "rb").read())
def member(b):
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = c.compress(b) + c.flush()
    return struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, 18 + len(raw) + 8 - 1) + raw + struct.pack("<II", zlib.crc32(b) & 0xffffffff, len(b))
blocks = [payload[o:o + 0xff00] for o in range(0, len(payload), 0xff00)]
with ThreadPoolExecutor(16) as ex:
    ms = list(ex.map(member, blocks))
eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
open("/tmp/zb.bam", "wb").write(b"".join(ms) + eof)
print("zlib-6 BAM: %d members, %.0f MB -> %.0f MB" % (len(ms), len(payload) / 1e6, os.path.getsize("/tmp/zb.bam") / 1e6), flush=True)
def walk(d, label):
    """Member by member through zlib, so that a bad member is named."""
    import hashlib
    o, k, h = 0, 0, hashlib.md5()
    while o < len(d):
        bs = struct.unpack_from("<H", d, o + 16)[0] + 1
        try:
            out = zlib.decompress(d[o + 18:o + bs - 8], -15)
            crc, isz = struct.unpack_from("<II", d, o + bs - 8)
            assert len(out) == isz and (zlib.crc32(out) & 0xffffffff) == crc, "crc/isize"
        except Exception as e:
            open(os.path.join(ROOT, "gpurun_out", "bad_member_%s_%d.bin" % (label.split()[-2], k)), "wb").write(d[o:o + bs])
            raise SystemExit("%s: member %d at %d (bsize %d) of %d bytes: %s" % (label, k, o, bs, len(d), e))
        h.update(out); o += bs; k += 1
    return h.hexdigest()


fade = os.path.join(ROOT, "fade_amd", "fade")
outs = {}
for label, env in (("host pipeline", {"FADE_BAM_DEVICE": "0"}), ("file path, host inflate", {"FADE_BAM_INFLATE": "host"}), ("file path, device inflate", {"FADE_BAM_INFLATE": "device"})):
    best = None
    for rep in range(2):
        t = time.perf_counter()
        p = subprocess.run([fade, "annotate", "-t", "16", "-w", "100", "-b", "/tmp/zb.bam", "/tmp/zb.fa"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        dt = time.perf_counter() - t
        assert p.returncode == 0, p.stderr.decode()[-500:]
        best = dt if best is None else min(best, dt)
    print("%-28s %.3f s  %.2f M reads/s" % (label, best, n / best / 1e6), flush=True)
    outs[label] = walk(p.stdout, label)
assert outs["file path, host inflate"] == outs["host pipeline"] and outs["file path, device inflate"] == outs["host pipeline"]
print("all three outputs inflate to the same bytes (md5 %s)" % outs["host pipeline"])
