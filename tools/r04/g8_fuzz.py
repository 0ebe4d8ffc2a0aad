"""Differential run for the eight-lane score kernels (sw_pk_kernel<R8,1,LG=8>, R8 = 5 / 7 / 10 / 13 / 19): random records of
tests/test_gpu_fuzz.py's generator (every CIGAR op, clips of any length, N / IUPAC, positions past the contig end, lower-case
and IUPAC reference stretches), cut down to the reads of at most 40 / 56 / 80 / 104 / 152 bases so that the batch's top row
class takes the eight-lane geometry; rs and am of every record against the oracle.   GPU box: python tools/r04/g8_fuzz.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["FADEHIP_DEBUG"] = "1"
import fade_amd  # noqa: E402
from fade_amd import format_tags, synth  # noqa: E402
from oracle import pyoracle as O  # noqa: E402
from test_gpu_fuzz import _random_batch  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
O.build()
ctx = fade_amd.Context(device=0)
t0, seed, cases, per = time.time(), 1, 0, {}
while time.time() - t0 < budget:
    rng = np.random.default_rng(70000 + seed)
    top = int(rng.choice([40, 56, 80, 104, 152]))
    contigs = []
    for k in range(3):
        s = bytearray(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(3000, 9000)))].tobytes())
        for p in rng.integers(0, len(s), size=len(s) // 40):
            s[p] = int(rng.choice(list(b"NNNRYKMacgtn")))
        contigs.append(bytes(s).decode())
    names = ["ctgA", "ctgB", "ctgC"]
    window = int(rng.choice([30, 100, 300]))
    b = _random_batch(rng, contigs, 6000, window)
    keep = np.nonzero(np.asarray(b["l_seq"]) <= top)[0]
    if len(keep) < 200:
        seed += 1
        continue
    b = synth.take(b, keep)
    floor_len = int(rng.choice([0, 5, 12]))
    ctx.genome_upload(names, [c.encode() for c in contigs])
    rs, aln, stats = ctx.annotate(b, floor_len, window)
    tags = format_tags(b, names, rs, aln)
    G = O.GenomeHolder(names, contigs)
    ors, oam = O.annotate_batch_soa(G, b, floor_len, window, threads=8)
    assert np.array_equal(rs, ors), (seed, top, np.nonzero(rs != ors)[0][:10])
    for i in range(len(ors)):
        if oam[i] is None:
            assert i not in tags, (seed, i)
        else:
            assert tags[i]["am"] == oam[i], (seed, i)
    cases += len(ors)
    per[top] = per.get(top, 0) + len(ors)
    seed += 1
    if seed % 20 == 0:
        print("%d batches, %d records, 0 mismatches (%.0f s)" % (seed - 1, cases, time.time() - t0), flush=True)
print("eight-lane kernels fuzz: %d batches, %d records (by longest read allowed: %s): 0 mismatches" % (seed - 1, cases, per))
