"""How long does the process take to go away?  `fade annotate -b` on a header-only BAM and on the 10 M-read file: wall time of
the process against its own clock at _exit (python tools/exit_probe.py; needs /tmp/e2eq.bam from tools/e2e_quick.py)."""
import os, struct, subprocess, sys, time, gzip, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FADE = os.path.join(ROOT, "fade_amd", "fade")
raw = open("/tmp/e2eq.bam", "rb").read(1 << 20)
# the header: inflate members until it is whole
at, payload = 0, b""
while True:
    bs = struct.unpack_from("<H", raw, at + 16)[0] + 1
    payload += zlib.decompress(raw[at + 18:at + bs - 8], -15)
    at += bs
    l_text = struct.unpack_from("<i", payload, 4)[0]
    if len(payload) >= 12 + l_text:
        n_ref = struct.unpack_from("<i", payload, 8 + l_text)[0]
        p = 12 + l_text
        ok = True
        for _ in range(n_ref):
            if p + 4 > len(payload): ok = False; break
            p += 4 + struct.unpack_from("<i", payload, p)[0] + 4
        if ok and p <= len(payload):
            break
hdr = payload[:p]
co = zlib.compressobj(6, zlib.DEFLATED, -15)
body = co.compress(hdr) + co.flush()
eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
open("/tmp/hdr_only.bam", "wb").write(struct.pack("<BBBBIBBHBBHH", 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 66, 67, 2, 18 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(hdr) & 0xffffffff, len(hdr)) + eof)
for label, bam, env in (("header only", "/tmp/hdr_only.bam", {}), ("10 M reads", "/tmp/e2eq.bam", {}), ("10 M reads, destructors", "/tmp/e2eq.bam", {"FADE_FAST_EXIT": "0"})):
    for rep in range(2):
        t0 = time.perf_counter()
        p = subprocess.run([FADE, "annotate", "--timing", "-t", "16", "-w", "100", "-b", bam, "/tmp/e2eq.fa"], stdout=open("/tmp/o.bam", "wb"), stderr=subprocess.PIPE, env=dict(os.environ, **env))
        dt = time.perf_counter() - t0
        lines = [l for l in p.stderr.decode().splitlines() if "since process start" in l or "total" in l]
        print(label, "wall %.3f s |" % dt, " | ".join(l.replace("[timing] ", "")[:90] for l in lines), flush=True)
