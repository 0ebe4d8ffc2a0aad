"""One-off: damaged BAM RECORDS (inside sound BGZF members: the CRCs are right) through the file path on the device and
through the host pipeline.  Per case: both refuse the file, or both give the same bytes; the device never hangs or faults.
Damage: stray bytes, a record's fixed fields (block_size, refID, pos, l_read_name, n_cigar_op, flag, l_seq, mate fields) set
to 0 / 1 / -1 / huge / off by one, CIGAR operations with impossible codes or lengths, tag types and B-array counts.
    GPU box: python tools/record_corrupt_fuzz.py [seconds]"""
import gzip
import os
import struct
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import samutil  # noqa: E402
from fade_amd import synth  # noqa: E402
from test_gpu_bam_stream import _random_aux  # noqa: E402
from test_gpu_inflate import EOF_MARK, member  # noqa: E402

FADE = os.path.join(ROOT, "fade_amd", "fade")
TMP = os.environ.get("TMPDIR", "/tmp")


def run(args, env):
    return subprocess.run([FADE] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=dict(os.environ, **env))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
    rng = np.random.default_rng(77)
    cfg, g, b = synth.make_config("C5", 5000, contig_len=200_000)
    names = ["read%d" % (i // 2) for i in range(len(b["pos"]))]
    b["qname"] = names
    sam, fa, bam = (os.path.join(TMP, "rcf." + x) for x in ("sam", "fa", "bam"))
    open(sam, "w").write(samutil.batch_to_sam(b, g.names, [int(x) for x in g.lengths], names))
    open(fa, "wb").write(g.fasta_bytes())
    p = run(["out", "-b", sam], {})
    assert p.returncode == 0
    payload = gzip.decompress(p.stdout)
    at = 8 + struct.unpack_from("<i", payload, 4)[0]
    n_ref = struct.unpack_from("<i", payload, at)[0]
    at += 4
    for _ in range(n_ref):
        at += 4 + struct.unpack_from("<i", payload, at)[0] + 4
    base = bytearray(payload[:at])
    offs = []
    while at < len(payload):
        bs = struct.unpack_from("<I", payload, at)[0]
        body = payload[at + 4:at + 4 + bs] + (_random_aux(rng, True) if rng.random() < 0.5 else b"")
        offs.append(len(base))
        base += struct.pack("<I", len(body)) + body
        at += 4 + bs
    t0, n, both_fail, both_same, kinds = time.time(), 0, 0, 0, {}
    while time.time() - t0 < budget:
        s = bytearray(base)
        kind = int(rng.integers(0, 5))
        r = offs[int(rng.integers(0, len(offs)))]
        bs, tid, pos, l_name, mapq, binv, n_cig, flag, l_seq = struct.unpack_from("<IiiBBHHHi", s, r)
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                s[int(rng.integers(len(payload[:8]), len(s)))] = int(rng.integers(0, 256))
        elif kind == 1:
            field = int(rng.integers(0, 9))
            foff, fmt = [(0, "<I"), (4, "<i"), (8, "<i"), (12, "<B"), (16, "<H"), (18, "<H"), (20, "<i"), (24, "<i"), (28, "<i")][field]
            old = struct.unpack_from(fmt, s, r + foff)[0]
            lim = {"<I": 2 ** 32, "<i": 2 ** 31, "<B": 256, "<H": 65536}[fmt]
            new = int(rng.choice([0, 1, lim - 1, old + 1, old - 1, int(rng.integers(0, lim)), 31, 32, 36]))
            if fmt == "<i":
                new = (new + 2 ** 31) % 2 ** 32 - 2 ** 31
            else:
                new %= lim
            struct.pack_into(fmt, s, r + foff, new)
        elif kind == 2 and n_cig:
            co = r + 36 + l_name + 4 * int(rng.integers(0, n_cig))
            struct.pack_into("<I", s, co, int(rng.choice([0, 9, 15, (2 ** 28 - 1) << 4, int(rng.integers(0, 2 ** 32))])))
        elif kind == 3:
            aux = r + 36 + l_name + 4 * n_cig + (l_seq + 1) // 2 + l_seq
            end = r + 4 + bs
            if aux + 3 <= end:
                k = int(rng.integers(aux, end))
                s[k] = int(rng.choice([0, 0xff, ord("B"), ord("Z"), ord("H"), ord("i"), ord("?"), int(rng.integers(0, 256))]))
        else:
            # grow or shrink a record by a few bytes without telling block_size
            k = int(rng.integers(r + 36, r + 4 + bs))
            if rng.random() < 0.5:
                del s[k:k + int(rng.integers(1, 5))]
            else:
                s[k:k] = bytes(rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8))
        blocks = [bytes(s[o:o + 0xff00]) for o in range(0, len(s), 0xff00)]
        open(bam, "wb").write(b"".join(member(x, 1) for x in blocks) + EOF_MARK)
        args = ["annotate", "-w", "100", "-b", bam, fa]
        host = run(args, {"FADE_BAM_DEVICE": "0"})
        mode = str(rng.choice(["host", "device"]))
        dev = run(args, {"FADE_BAM_INFLATE": mode, "FADE_BAM_CHUNK_MB": str(rng.choice([1, 64]))})
        what = "case %d (damage %d at record offset %d, inflate %s)" % (n, kind, r, mode)
        assert dev.returncode in (0, 1) and host.returncode in (0, 1), what + ": rc %d / %d\n%s" % (dev.returncode, host.returncode, dev.stderr.decode()[-600:])
        if dev.returncode == 0 and host.returncode == 0:
            assert gzip.decompress(dev.stdout) == gzip.decompress(host.stdout), what + ": both accept, bytes differ"
            both_same += 1
        elif dev.returncode and host.returncode:
            both_fail += 1
        else:
            kinds[kind] = kinds.get(kind, 0) + 1
            print(what + ": device rc %d, host rc %d | device: %s | host: %s" % (dev.returncode, host.returncode, dev.stderr.decode().strip().splitlines()[-1][:160] if dev.stderr.strip() else "",
                                                                             host.stderr.decode().strip().splitlines()[-1][:160] if host.stderr.strip() else ""), flush=True)
        n += 1
        if n % 25 == 0:
            print("%d cases: %d refused by both, %d accepted by both with the same bytes, %d one-sided (%.0f s)" % (n, both_fail, both_same, sum(kinds.values()), time.time() - t0), flush=True)
    print("record corruption fuzz: %d cases: %d refused by both, %d accepted by both with the same bytes, %d one-sided %s" % (n, both_fail, both_same, sum(kinds.values()), kinds))


if __name__ == "__main__":
    main()
