"""Pure-Python restatement of `fade extract` (source/remap.d:11-87) — TEST INFRASTRUCTURE ONLY.

Small cases only (Python loops).  PARITY UNPINNED: the reference ships no fixtures for this path; a fresh
dhtslib SAMRecord is a zero-filled bam1_t (bam_init1), hence mapq 0, mate tid 0, mate pos 0, tlen 0."""

_COMP = {"=": "=", "A": "T", "C": "G", "M": "K", "G": "C", "R": "Y", "S": "S", "V": "B", "T": "A", "W": "W",
         "Y": "R", "H": "D", "K": "M", "D": "H", "B": "V", "N": "N"}


def reverse_complement(seq):
    """util.d:23-34 on the nt16 alphabet."""
    return "".join(_COMP[c] for c in reversed(seq))


def extract_records(records, contig_names):
    """records: dicts from tests/samutil.parse_sam.  Returns SAM lines (no header), remap.d:29-85."""
    out = []
    for r in records:
        t = r["tags"]
        if "rs" not in t:                     # remap.d:31-33
            continue
        rs = int(t["rs"][1]) & 0xFF
        if not (rs & 6):                      # remap.d:36-37
            continue
        if "am" not in t:                     # remap.d:38-40
            continue
        am_split = t["am"][1].split(";")      # remap.d:42
        for side, bit in ((0, 2), (1, 4)):    # remap.d:43, 64
            if not (rs & bit):
                continue
            name, pos, cigar = am_split[side].split(",")   # remap.d:46, 67
            tid = contig_names.index(name)
            flag = 0 if (r["flag"] & 0x10) else 0x10        # remap.d:51-58
            seq = reverse_complement(r["seq"])              # remap.d:59
            qual = r["qual"][::-1]                          # remap.d:60
            rnext = "=" if tid == 0 else contig_names[0]    # mate tid 0 of the zero-filled record
            out.append("\t".join([r["qname"], str(flag), name, str(int(pos) + 1), "0", cigar, rnext, "1", "0", seq, qual]))
    return out
