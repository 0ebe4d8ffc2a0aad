"""Deterministic synthetic inputs for the annotate hot path.

Genome: iid uniform ACGT contigs.  Reads: paired-end, insert ~ N(350, 50) clamped >= Lq, R1 forward
(flag 99) / R2 reverse (flag 147), 0.1 % substitutions, quals U[20,40], 1 % unmapped, 2 % with an
SA tag.  Soft clips: with probability p_sc a read gets a clip (left 47.5 %, right 47.5 %, both 5 %),
length U[clip_min, clip_max]; half of the clips are *planted artifacts*: the reverse complement of
the window segment that starts d ~ U[0, W/2] bases inside the ±W window on the clip's side, so the
reverse-complemented read matches the forward strand there (what FADE calls an enzymatic
fragmentation artifact); the other half are iid random bases.

Output is the BAM-native structure-of-arrays batch the C ABI takes (include/fadehip.h), as a dict
of numpy arrays, plus `qual`/`qual_off`/`qname` for the tag formatter and the oracle.
The random stream is numpy PCG64 (the survey suggested xoshiro256**; any fixed generator serves).
"""
import numpy as np

# nt16 codes of A, C, G, T  (htslib seq_nt16_str "=ACMGRSVTWYHKDBN")
_ACGT_NT16 = np.array([1, 2, 4, 8], dtype=np.uint8)
_ACGT_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)

CONFIGS = {
    # name: (contigs, contig_len, genome_seed, read_len, window, p_sc, clip_min, clip_max, insert_mu, insert_sd)
    "C1": dict(n_contigs=1, contig_len=1_000_000, genome_seed=1, read_len=150, window=300, p_sc=0.10,
               clip_min=6, clip_max=50, insert_mu=350, insert_sd=50, floor_len=5),
    "C2": dict(n_contigs=4, contig_len=25_000_000, genome_seed=42, read_len=150, window=100, p_sc=0.10,
               clip_min=6, clip_max=50, insert_mu=350, insert_sd=50, floor_len=5),
    "C3": dict(n_contigs=4, contig_len=25_000_000, genome_seed=42, read_len=250, window=300, p_sc=0.10,
               clip_min=6, clip_max=50, insert_mu=600, insert_sd=80, floor_len=5),
    "C5": dict(n_contigs=4, contig_len=25_000_000, genome_seed=42, read_len=150, window=100, p_sc=0.30,
               clip_min=1, clip_max=60, insert_mu=350, insert_sd=50, floor_len=5),
}
CONFIGS["C4"] = dict(CONFIGS["C2"])
# C6 (round 3): C2's reads on a repeat-rich genome — tandem repeats, homopolymer runs and a 10 % segmental duplication —
# where the re-alignment has co-optimal and gapped paths, the forced-diagonal shortcut hits less often and the traced pass
# works for its living (the iid-uniform genome of C1-C5 flatters the shortcut)
CONFIGS["C6"] = dict(CONFIGS["C2"], genome_kind="repeat_rich", p_clip_indel=0.3)


def make_repeat_rich(codes, n_contigs, contig_len, seed):
    """In place, on base codes 0..3 (all contigs concatenated): ~15 % of every contig becomes tandem repeats (unit 1-6 bases,
    30-400 bases long, 2 % of the copies' bases substituted), ~5 % homopolymer runs (10-60 bases), and 10 % of the genome is
    overwritten by copies of other stretches of it (segments of 5-40 kb, 1 % divergence): segmental duplications."""
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    for c in range(n_contigs):
        g = codes[c * contig_len:(c + 1) * contig_len]
        # segmental duplications first (the repeats then differ between the copies a little, as in real genomes)
        left = contig_len // 10
        while left > 0:
            ln = int(min(rng.integers(5_000, 40_001), max(left, 1000), contig_len // 4))
            src_c = int(rng.integers(0, n_contigs))
            src = int(rng.integers(0, contig_len - ln))
            dst = int(rng.integers(0, contig_len - ln))
            seg = codes[src_c * contig_len + src:src_c * contig_len + src + ln].copy()
            m = rng.random(ln) < 0.01
            seg[m] = (seg[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
            g[dst:dst + ln] = seg
            left -= ln
        n_tr = int(0.15 * contig_len / 200)
        for _ in range(n_tr):
            unit = rng.integers(0, 4, int(rng.integers(1, 7)), dtype=np.uint8)
            ln = int(rng.integers(30, 401))
            at = int(rng.integers(0, contig_len - ln))
            rep = np.resize(unit, ln).copy()
            m = rng.random(ln) < 0.02
            rep[m] = rng.integers(0, 4, int(m.sum()), dtype=np.uint8)
            g[at:at + ln] = rep
        n_hp = int(0.05 * contig_len / 35)
        for _ in range(n_hp):
            ln = int(rng.integers(10, 61))
            at = int(rng.integers(0, contig_len - ln))
            g[at:at + ln] = rng.integers(0, 4)


class Genome:
    def __init__(self, n_contigs, contig_len, seed, prefix="chr", kind="uniform"):
        rng = np.random.Generator(np.random.PCG64(seed))
        self.names = ["%s%d" % (prefix, k + 1) for k in range(n_contigs)]
        self.lengths = np.full(n_contigs, contig_len, dtype=np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.lengths)]).astype(np.int64)
        # base codes 0..3 (A,C,G,T), all contigs concatenated
        self.codes = rng.integers(0, 4, size=int(self.offsets[-1]), dtype=np.uint8)
        if kind == "repeat_rich":
            make_repeat_rich(self.codes, n_contigs, contig_len, seed)

    def ascii_contigs(self):
        """list of uint8 arrays of ASCII residues (upper case)."""
        return [_ACGT_ASCII[self.codes[self.offsets[k]:self.offsets[k + 1]]] for k in range(len(self.names))]

    def fasta_bytes(self, width=60):
        out = []
        for k, name in enumerate(self.names):
            seq = self.ascii_contigs()[k].tobytes()
            out.append(b">" + name.encode() + b"\n")
            out.append(b"\n".join(seq[i:i + width] for i in range(0, len(seq), width)) + b"\n")
        return b"".join(out)


def make_reads(genome, n_reads, seed, read_len=150, window=100, p_sc=0.10, clip_min=6, clip_max=50,
               insert_mu=350, insert_sd=50, p_unmapped=0.01, p_sa=0.02, p_sub=0.001, p_planted=0.5,
               with_names=False, chunk=200_000, p_clip_indel=0.0, **_unused):
    """Returns the batch dict.  Deterministic in (genome, all arguments)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    Lq = int(read_len)
    n_pairs = (n_reads + 1) // 2
    nc = len(genome.names)
    clen = int(genome.lengths[0])
    # fragments
    tid_p = rng.integers(0, nc, size=n_pairs, dtype=np.int32)
    ins = np.maximum(np.rint(rng.normal(insert_mu, insert_sd, size=n_pairs)).astype(np.int64), Lq)
    ins = np.minimum(ins, clen)
    fs = (rng.random(n_pairs) * (clen - ins + 1)).astype(np.int64)
    tid = np.repeat(tid_p, 2)[:n_reads]
    span = np.empty(2 * n_pairs, dtype=np.int64)
    span[0::2] = fs
    span[1::2] = fs + ins - Lq
    span = span[:n_reads]
    flag = np.empty(2 * n_pairs, dtype=np.uint16)
    flag[0::2] = 99
    flag[1::2] = 147
    flag = flag[:n_reads]
    n = n_reads
    # clips
    u = rng.random(n)
    has_clip = u < p_sc
    side = rng.random(n)
    left = has_clip & ((side < 0.475) | (side >= 0.95))
    right = has_clip & (side >= 0.475)
    clipL = np.where(left, rng.integers(clip_min, clip_max + 1, size=n), 0).astype(np.int64)
    clipR = np.where(right, rng.integers(clip_min, clip_max + 1, size=n), 0).astype(np.int64)
    both = clipL + clipR >= Lq
    clipR[both] = 0
    plantL = left & (rng.random(n) < p_planted)
    plantR = right & (rng.random(n) < p_planted)
    dL = rng.integers(0, window // 2 + 1, size=n).astype(np.int64)
    dR = rng.integers(0, window // 2 + 1, size=n).astype(np.int64)
    A = Lq - clipL - clipR
    pos = span + clipL
    win_start = np.maximum(pos - window, 0)
    win_end = np.minimum(pos + A + window, clen)
    segL = win_start + dL                 # planted left segment [segL, segL+clipL)
    segR_end = win_end - dR               # planted right segment [segR_end-clipR, segR_end)
    plantL &= (segL + clipL <= win_end)
    plantR &= (segR_end - clipR >= win_start)
    unmapped = rng.random(n) < p_unmapped
    has_sa = (rng.random(n) < p_sa).astype(np.uint8)
    goff = genome.offsets[tid]

    nbytes = (Lq + 1) // 2
    seq_packed = np.empty(n * nbytes, dtype=np.uint8)
    qual = rng.integers(20, 41, size=(n, Lq), dtype=np.uint8).reshape(-1)
    k = np.arange(Lq, dtype=np.int64)[None, :]
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        m = c1 - c0
        sl = slice(c0, c1)
        idx = goff[sl, None] + np.clip(span[sl, None] + k, 0, clen - 1)
        inL = k < clipL[sl, None]
        inR = k >= (Lq - clipR[sl, None])
        # planted clips read the reverse strand of their segment
        pl = inL & plantL[sl, None]
        pr = inR & plantR[sl, None]
        idx = np.where(pl, goff[sl, None] + segL[sl, None] + clipL[sl, None] - 1 - k, idx)
        mR = k - (Lq - clipR[sl, None])
        idx = np.where(pr, goff[sl, None] + segR_end[sl, None] - 1 - mR, idx)
        base = genome.codes[np.clip(idx, 0, len(genome.codes) - 1)]
        base = np.where(pl | pr, 3 - base, base)  # complement in A,C,G,T order
        rnd = rng.integers(0, 4, size=(m, Lq), dtype=np.uint8)
        base = np.where((inL | inR) & ~(pl | pr), rnd, base)
        sub = rng.random((m, Lq)) < p_sub
        shift = rng.integers(1, 4, size=(m, Lq), dtype=np.uint8)
        base = np.where(sub, (base + shift) & 3, base).astype(np.uint8)
        if p_clip_indel > 0:
            # a planted clip of 12 bases or more loses one inner base with this probability (the clip's outer end takes a
            # random one): the re-alignment then has a gap in it, the forced-diagonal shortcut does not apply (C6)
            hit = rng.random(m) < p_clip_indel
            for r in np.nonzero(hit & (plantL[sl] | plantR[sl]))[0]:
                if plantL[c0 + r] and clipL[c0 + r] >= 12:
                    L = int(clipL[c0 + r])
                    at = int(rng.integers(3, L - 3))
                    base[r, 1:at + 1] = base[r, 0:at].copy()
                    base[r, 0] = rng.integers(0, 4)
                elif plantR[c0 + r] and clipR[c0 + r] >= 12:
                    L = int(clipR[c0 + r])
                    at = Lq - L + int(rng.integers(3, L - 3))
                    base[r, at:Lq - 1] = base[r, at + 1:Lq].copy()
                    base[r, Lq - 1] = rng.integers(0, 4)
        codes = _ACGT_NT16[base]
        if Lq & 1:
            codes = np.concatenate([codes, np.zeros((m, 1), dtype=np.uint8)], axis=1)
        seq_packed[c0 * nbytes:c1 * nbytes] = ((codes[:, 0::2] << 4) | codes[:, 1::2]).reshape(-1)

    # CIGARs: [clipL S][A M][clipR S]; unmapped reads carry none
    n_ops = (clipL > 0).astype(np.int64) + 1 + (clipR > 0).astype(np.int64)
    n_ops[unmapped] = 0
    cigar_off = np.concatenate([[0], np.cumsum(n_ops)]).astype(np.uint32)
    cigar_ops = np.zeros(int(cigar_off[-1]), dtype=np.uint32)
    mp = ~unmapped
    o = cigar_off[:-1].astype(np.int64)
    hasL = mp & (clipL > 0)
    cigar_ops[o[hasL]] = (clipL[hasL].astype(np.uint32) << 4) | 4
    om = o + (clipL > 0)
    cigar_ops[om[mp]] = (A[mp].astype(np.uint32) << 4) | 0
    hasR = mp & (clipR > 0)
    cigar_ops[(om + 1)[hasR]] = (clipR[hasR].astype(np.uint32) << 4) | 4

    flag = flag.copy()
    flag[unmapped] |= 4
    tid_out = tid.astype(np.int32).copy()
    pos_out = pos.astype(np.int32).copy()
    tid_out[unmapped] = -1
    pos_out[unmapped] = -1
    batch = dict(
        tid=tid_out, pos=pos_out, flag=flag, has_sa=has_sa, l_seq=np.full(n, Lq, dtype=np.int32),
        cigar_off=cigar_off, cigar_ops=cigar_ops,
        seq_off=(np.arange(n + 1, dtype=np.int64) * nbytes).astype(np.uint32), seq_packed=seq_packed,
        qual_off=(np.arange(n + 1, dtype=np.int64) * Lq), qual=qual,
    )
    if with_names:
        batch["qname"] = [b"r%d" % (i // 2) for i in range(n)]
    # ground truth for planted artifacts (construction-known answers, independent of tie rules
    # whenever the flanking bases mismatch): see tests/test_known_answers.py
    batch["_truth"] = dict(plantL=plantL & mp, plantR=plantR & mp, segL=segL, segR=segR_end - clipR, clipL=clipL,
                           clipR=clipR)
    return batch


def config(name):
    return dict(CONFIGS[name])


def make_config(name, n_reads, read_seed=7, genome=None, contig_len=None):
    """Genome + batch for a BASELINE.json config, optionally with a shorter genome (tests)."""
    cfg = config(name)
    if contig_len is not None:
        cfg["contig_len"] = contig_len
    g = genome or Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"], kind=cfg.get("genome_kind", "uniform"))
    b = make_reads(g, n_reads, read_seed, **cfg)
    return cfg, g, b


def take(batch, idx):
    """Sub-batch of the given read indices (numpy int array), re-packed."""
    idx = np.asarray(idx, dtype=np.int64)
    co, so, qo = batch["cigar_off"].astype(np.int64), batch["seq_off"].astype(np.int64), batch["qual_off"]
    ncig = co[idx + 1] - co[idx]
    nseq = so[idx + 1] - so[idx]
    nq = qo[idx + 1] - qo[idx]

    def gather(src, starts, lens):
        total = int(lens.sum())
        if total == 0:
            return src[:0].copy()
        rep = np.repeat(starts - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)
        return src[rep + np.arange(total)]

    out = dict(tid=batch["tid"][idx], pos=batch["pos"][idx], flag=batch["flag"][idx], has_sa=batch["has_sa"][idx],
               l_seq=batch["l_seq"][idx],
               cigar_off=np.concatenate([[0], np.cumsum(ncig)]).astype(np.uint32),
               cigar_ops=gather(batch["cigar_ops"], co[idx], ncig),
               seq_off=np.concatenate([[0], np.cumsum(nseq)]).astype(np.uint32),
               seq_packed=gather(batch["seq_packed"], so[idx], nseq),
               qual_off=np.concatenate([[0], np.cumsum(nq)]).astype(np.int64),
               qual=gather(batch["qual"], qo[idx], nq))
    if "qname" in batch:
        out["qname"] = [batch["qname"][i] for i in idx]
    return out


def concat(batches):
    """One batch out of several (records in order)."""
    out = {}
    for k in ("tid", "pos", "flag", "has_sa", "l_seq", "cigar_ops", "seq_packed", "qual"):
        out[k] = np.concatenate([b[k] for b in batches])
    for off, data in (("cigar_off", "cigar_ops"), ("seq_off", "seq_packed"), ("qual_off", "qual")):
        parts, base = [np.zeros(1, dtype=np.int64)], 0
        for b in batches:
            parts.append(np.asarray(b[off], dtype=np.int64)[1:] + base)
            base += len(b[data])
        out[off] = np.concatenate(parts).astype(batches[0][off].dtype)
    if all("qname" in b for b in batches):
        out["qname"] = [q for b in batches for q in b["qname"]]
    return out
