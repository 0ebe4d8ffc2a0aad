"""Shared helpers for the parity tests (random / adversarial sequence pairs)."""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
IUPAC = np.frombuffer(b"ACGTNRYMKSWBDHV", dtype=np.uint8)
COMP = {ord(a): ord(b) for a, b in zip("ACGTNRYMKSWBDHV", "TGCANYRKMSWVHDB")}


def rand_seq(rng, n, alphabet=ACGT):
    return alphabet[rng.integers(0, len(alphabet), size=n)]


def revcomp(a):
    return np.array([COMP[int(x)] for x in a[::-1]], dtype=np.uint8)


def mutate(rng, s, p_sub=0.05, p_indel=0.01):
    out = []
    for c in s:
        u = rng.random()
        if u < p_indel:
            continue
        if u < 2 * p_indel:
            out.append(int(ACGT[rng.integers(0, 4)]))
        if u < 2 * p_indel + p_sub:
            out.append(int(ACGT[rng.integers(0, 4)]))
        else:
            out.append(int(c))
    return np.array(out, dtype=np.uint8)


def make_pairs(rng, n, lq_range=(20, 260), lr_range=(30, 900), kinds=("random", "planted", "homopolymer", "tandem",
                                                                      "nrich", "iupac", "related")):
    """List of (q, r) uint8 arrays covering the adversarial families of SURVEY.md §8c."""
    qs, rs = [], []
    for k in range(n):
        kind = kinds[k % len(kinds)]
        lq = int(rng.integers(lq_range[0], lq_range[1] + 1))
        lr = int(rng.integers(lr_range[0], lr_range[1] + 1))
        if kind == "random":
            q, r = rand_seq(rng, lq), rand_seq(rng, lr)
        elif kind == "planted":
            r = rand_seq(rng, lr)
            q = rand_seq(rng, lq)
            hi = min(lq, lr, 60)
            L = int(rng.integers(min(4, hi), hi + 1))
            s = int(rng.integers(0, lr - L + 1))
            if rng.random() < 0.5:
                q[lq - L:] = r[s:s + L]
            else:
                q[:L] = r[s:s + L]
        elif kind == "homopolymer":
            q = np.full(lq, ACGT[rng.integers(0, 4)], dtype=np.uint8)
            r = rand_seq(rng, lr)
            a = int(rng.integers(0, lr))
            r[a:a + int(rng.integers(5, 80))] = q[0]
        elif kind == "tandem":
            unit = rand_seq(rng, int(rng.integers(1, 7)))
            q = np.resize(unit, lq).copy()
            r = np.resize(np.roll(unit, int(rng.integers(0, len(unit)))), lr).copy()
            for arr in (q, r):
                m = rng.random(len(arr)) < 0.03
                arr[m] = rand_seq(rng, int(m.sum()))
        elif kind == "nrich":
            q, r = rand_seq(rng, lq), rand_seq(rng, lr)
            q[rng.random(lq) < 0.2] = ord("N")
            r[rng.random(lr) < 0.2] = ord("N")
        elif kind == "iupac":
            q, r = rand_seq(rng, lq, IUPAC), rand_seq(rng, lr, IUPAC)
        elif kind == "lowcomplexity":  # two-letter sequences related by block indels: co-optimal paths, i.e. ties between
            # the gap directions and between opening and extending a gap (what Appendix A.4's rules decide)
            two = np.frombuffer(b"AC", dtype=np.uint8)
            lr = int(rng.integers(40, 200))
            r = two[rng.integers(0, 2, size=lr)]
            a = int(rng.integers(0, lr // 2))
            ql = list(r[a:a + int(rng.integers(20, 120))])
            for _ in range(int(rng.integers(1, 4))):
                if len(ql) < 12:
                    break
                p = int(rng.integers(3, len(ql) - 3))
                L = int(rng.integers(1, 6))
                if rng.random() < 0.5:
                    del ql[p:p + L]
                else:
                    ql[p:p] = list(two[rng.integers(0, 2, size=L)])
            q = np.array(ql, dtype=np.uint8)
        elif kind == "refspecial":  # N / IUPAC columns only in the reference: the query stays pure A,C,G,T
            r = rand_seq(rng, lr)
            a = int(rng.integers(0, max(1, lr - 10)))
            q = mutate(rng, r[a:a + lq])
            if len(q) == 0:
                q = rand_seq(rng, 5)
            m = rng.random(lr) < 0.15
            r[m] = rand_seq(rng, int(m.sum()), IUPAC)
            r[rng.random(lr) < 0.05] = ord("N")
        else:  # related: query is a mutated slice of the reference (gaps and mismatches)
            r = rand_seq(rng, lr)
            a = int(rng.integers(0, max(1, lr - 10)))
            q = mutate(rng, r[a:a + lq])
            if len(q) == 0:
                q = rand_seq(rng, 5)
        qs.append(np.ascontiguousarray(q))
        rs.append(np.ascontiguousarray(r))
    return qs, rs


def concat(seqs):
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    cat = np.concatenate(seqs) if len(seqs) and off[-1] else np.zeros(0, np.uint8)
    return cat.astype(np.uint8), off
