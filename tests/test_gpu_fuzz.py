"""Differential fuzzing of the level-2 path: records with arbitrary CIGAR structure (H/S/M/=/X/I/D/N/P in any
legal order, soft clips of any length, mid-CIGAR S), positions anywhere including beyond the contig end, mixed
read lengths, random flags, N / IUPAC / lower-case reference stretches — GPU vs the oracle's annotateTask."""
import numpy as np
import pytest

import fade_amd
from fade_amd import format_tags

pytestmark = pytest.mark.gpu

NT16 = np.array([1, 2, 4, 8], dtype=np.uint8)
COMP = {1: 8, 2: 4, 4: 2, 8: 1, 15: 15}


def _random_batch(rng, contigs, n, window):
    tid, pos, flag, has_sa, l_seq = [], [], [], [], []
    cigar_off, cigar_ops, seq_off, seq_packed, qual_off, qual = [0], [], [0], [], [0], []
    code_of = {ord("A"): 1, ord("C"): 2, ord("G"): 4, ord("T"): 8}
    for i in range(n):
        t = int(rng.integers(0, len(contigs)))
        clen = len(contigs[t])
        ops = []
        if rng.random() < 0.1:
            ops.append((int(rng.integers(1, 20)), 5))
        lead = int(rng.integers(0, 60)) if rng.random() < 0.5 else 0
        if lead:
            ops.append((lead, 4))
        body = []
        for _ in range(int(rng.integers(1, 6))):
            op = int(rng.choice([0, 0, 0, 7, 8, 1, 2, 3, 6, 4 if rng.random() < 0.05 else 0]))
            ln = int(rng.integers(1, 60)) if op != 3 else int(rng.integers(1, 400))
            body.append((ln, op))
        if body[0][1] in (1, 2, 3, 6):
            body.insert(0, (int(rng.integers(5, 40)), 0))
        ops += body
        trail = int(rng.integers(0, 60)) if rng.random() < 0.5 else 0
        if trail:
            ops.append((trail, 4))
        if rng.random() < 0.1:
            ops.append((int(rng.integers(1, 20)), 5))
        lq = sum(l for l, o in ops if o in (0, 1, 4, 7, 8))
        if lq > 500:  # stay inside the kernel's read-length limit
            ops = [(100, 0)]
            lq = 100
        a_len = sum(l for l, o in ops if o in (0, 2, 3, 7, 8))
        p = int(rng.integers(0, clen + 50)) if rng.random() < 0.1 else int(rng.integers(0, max(1, clen - a_len)))
        f = int(rng.choice([0, 16, 99, 147, 83, 163, 2048, 256]))
        if rng.random() < 0.04:
            f |= 4
        # sequence: reference-like with a planted reverse-complement clip half of the time
        codes = NT16[rng.integers(0, 4, size=lq)]
        ref = contigs[t]
        if lead >= 6 and rng.random() < 0.6:
            s0 = max(0, min(clen - lead, p - int(rng.integers(0, window + 20))))
            seg = ref[s0:s0 + lead]
            if len(seg) == lead and all(c in code_of for c in seg.upper().encode()):
                codes[:lead] = [COMP[code_of[c]] for c in seg.upper().encode()][::-1]
        if trail >= 6 and rng.random() < 0.6:
            e0 = min(clen, p + a_len + int(rng.integers(0, window + 20)))
            seg = ref[max(0, e0 - trail):e0]
            if len(seg) == trail and all(c in code_of for c in seg.upper().encode()):
                codes[lq - trail:] = [COMP[code_of[c]] for c in seg.upper().encode()][::-1]
        if rng.random() < 0.05:
            codes[rng.integers(0, lq, size=max(1, lq // 10))] = 15
        if rng.random() < 0.02:
            codes[rng.integers(0, lq, size=2)] = int(rng.choice([3, 5, 10, 0]))
        tid.append(t); pos.append(p); flag.append(f); has_sa.append(int(rng.random() < 0.1)); l_seq.append(lq)
        cigar_ops += [(l << 4) | o for l, o in ops]
        cigar_off.append(len(cigar_ops))
        c = list(codes) + ([0] if lq & 1 else [])
        seq_packed += [(c[k] << 4) | c[k + 1] for k in range(0, len(c), 2)]
        seq_off.append(len(seq_packed))
        qual += list(rng.integers(2, 41, size=lq))
        qual_off.append(len(qual))
    return dict(tid=np.array(tid, np.int32), pos=np.array(pos, np.int32), flag=np.array(flag, np.uint16),
                has_sa=np.array(has_sa, np.uint8), l_seq=np.array(l_seq, np.int32),
                cigar_off=np.array(cigar_off, np.uint32), cigar_ops=np.array(cigar_ops, np.uint32),
                seq_off=np.array(seq_off, np.uint32), seq_packed=np.array(seq_packed, np.uint8),
                qual_off=np.array(qual_off, np.int64), qual=np.array(qual, np.uint8))


@pytest.mark.parametrize("seed,floor_len,window", [(1, 5, 100), (2, 5, 300), (3, 0, 30), (4, 12, 700)])
def test_fuzz_records(ctx, oracle, seed, floor_len, window):
    rng = np.random.default_rng(seed)
    contigs = []
    for k in range(3):
        s = bytearray(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(3000, 9000)))].tobytes())
        for p in rng.integers(0, len(s), size=len(s) // 40):
            s[p] = int(rng.choice(list(b"NNNRYKMacgtn")))
        a = int(rng.integers(0, len(s) - 300))
        s[a:a + 200] = bytes(s[a:a + 200]).lower()
        contigs.append(bytes(s).decode())
    names = ["ctgA", "ctgB", "ctgC"]
    b = _random_batch(rng, contigs, 3000, window)
    ctx.genome_upload(names, [c.encode() for c in contigs])
    rs, aln, stats = ctx.annotate(b, floor_len, window)
    tags = format_tags(b, names, rs, aln)
    G = oracle.GenomeHolder(names, contigs)
    ors, oam = oracle.annotate_batch_soa(G, b, floor_len, window, threads=8)
    assert np.array_equal(rs, ors), np.nonzero(rs != ors)[0][:10]
    reads, keep = oracle.make_reads(b)
    n_art = 0
    for i in np.nonzero((ors >> 1) & 3)[0]:
        a = oracle.annotate_one(G, reads[int(i)], floor_len, window)
        t = tags[int(i)]
        assert (a["am"], a["as_"], a["ar"], a["ab"]) == (t["am"], t["as_"], t["ar"], t["ab"]), int(i)
        n_art += 1
    assert set(tags) == set(int(i) for i in np.nonzero((ors >> 1) & 3)[0])
    assert n_art >= 20
