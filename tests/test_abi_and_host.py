"""CPU-side checks: the C-ABI library loads and exports every symbol include/fadehip.h declares;
host-side packing / formatting logic; no compute calls (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "fadehip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fadehip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from fade_amd import _lib
    L = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(L, name), "libfadehip.so lacks %s" % name
    assert sorted(_lib.EXPORTS) == declared
    assert L.fadehip_abi_version() == _lib.ABI_VERSION == 3
    hdr = open(os.path.join(ROOT, "include", "fadehip.h")).read()
    assert "#define FADEHIP_ABI_VERSION %d" % _lib.ABI_VERSION in hdr and "#define FADEHIP_NUM_SLOTS %d" % _lib.NUM_SLOTS in hdr
    p = _lib.Params()
    L.fadehip_params_default(ctypes.byref(p))
    assert (p.open, p.ext, p.match, p.mismatch) == (10, 2, 2, -3)  # anno.d:36
    assert p.rules == _lib.RULES_DEFAULT and p.max_ref_len == 1 << 20


def test_batch_block_layout():
    """fadehip_batch_bytes / fadehip_batch_bind (no GPU needed): nine arrays, 256-byte aligned, inside the block."""
    from fade_amd import _lib
    L = _lib.load()
    n, n_cig, n_seq = 1000, 2345, 75_000
    total = L.fadehip_batch_bytes(n, n_cig, n_seq)
    assert total >= 4 * n * 3 + 8 * (n + 1) + 3 * n + 4 * n_cig + n_seq
    buf = ctypes.create_string_buffer(total + 256)
    base = (ctypes.addressof(buf) + 255) & ~255
    b = _lib.ReadBatch()
    assert L.fadehip_batch_bind(base, n, n_cig, n_seq, ctypes.byref(b)) == 0
    sizes = dict(tid=4 * n, pos=4 * n, l_seq=4 * n, cigar_off=4 * (n + 1), seq_off=4 * (n + 1), flag=2 * n, has_sa=n,
                 cigar_ops=4 * n_cig, seq_packed=n_seq)
    spans = sorted((getattr(b, k), getattr(b, k) + v) for k, v in sizes.items())
    assert spans[0][0] == base and spans[-1][1] <= base + total
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0 and b0 % 256 == 0
    assert b.n_reads == n and b.n_skipped == 0 and b.ref_span_bound == 0 and b.n_with_seq == 0 and b.l_seq_max == 0
    assert L.fadehip_batch_bind(base + 8, n, n_cig, n_seq, ctypes.byref(b)) != 0  # misaligned block


def test_struct_layouts_match_header():
    from fade_amd import _lib
    assert ctypes.sizeof(_lib.SwResult) == 24 + 4 * 16
    assert ctypes.sizeof(_lib.Aln) == 32 + ctypes.sizeof(_lib.SwResult)
    assert ctypes.sizeof(_lib.ReadBatch) == 8 + 9 * 8 + 8 + 16 and ctypes.sizeof(_lib.Params) == 40
    assert ctypes.sizeof(_lib.AnnoView) == 16 + 8 + 64 + 8 and ctypes.sizeof(_lib.AnnoOut) == 16 + 8 + 64 + 8
    assert _lib.ALN_DTYPE.fields["sw"][1] == 32
    assert ctypes.sizeof(_lib.BamConfig) == 16 + 8 + 8  # fadehip_bam_config: four ints, the names pointer, two uints


def test_file_path_entry_points_fail_loudly_without_a_context():
    """The file path (fadehip_bam_*) and the codec calls check their arguments before they touch a device: NULL contexts and
    streams are errors with a message, not crashes — and there is no host fallback behind them."""
    from fade_amd import _lib
    L = _lib.load()
    n = ctypes.c_size_t(0)
    p = ctypes.c_void_p()
    assert L.fadehip_bgzf_inflate(None, b"x", 1, None, 0, ctypes.byref(n)) == -1
    assert L.fadehip_bgzf_deflate_submit(None, 0, b"x", 1) == -1
    assert L.fadehip_bam_open(None, None, ctypes.byref(p)) == -1
    assert L.fadehip_bam_front(None, b"x", 1, 0) == -1 and L.fadehip_bam_front_raw(None, b"x", 1, 0) == -1
    assert L.fadehip_bam_back(None, ctypes.byref(p), ctypes.byref(n)) == -1
    assert L.fadehip_bam_totals(None, None, None, None) == -1
    L.fadehip_bam_close(None)
    assert b"NULL" in L.fadehip_last_error(None)


def test_create_fails_loudly_without_gpu():
    """There is no CPU fallback: without a gfx950 device the product path refuses to start."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import fade_amd
    with pytest.raises(fade_amd.FadeHipError) as e:
        fade_amd.Context(device=0)
    assert e.value.code in (-2, -3)


def test_cli_file_path_fails_loudly_without_gpu(tmp_path):
    """`fade annotate -b in.bam` (the file path on the device) and the host pipeline alike: no device, no output — an error on
    stderr, exit code 1, not a byte on stdout."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "-s", "sam2bam"])
    gold = os.path.join(ROOT, "tests", "golden")
    bam = tmp_path / "in.bam"
    with open(bam, "wb") as fo:
        subprocess.check_call([os.path.join(ROOT, "tools", "sam2bam"), os.path.join(gold, "anno_c1.sam")], stdout=fo)
    for env in ({}, {"FADE_BAM_DEVICE": "0"}, {"FADE_BAM_INFLATE": "device"}):
        p = subprocess.run([os.path.join(ROOT, "fade_amd", "fade"), "annotate", "-b", str(bam), os.path.join(gold, "anno_c1.fa")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=dict(os.environ, **env))
        assert p.returncode == 1 and p.stdout == b"" and b"[E::fade annotate] cannot open the GPU path" in p.stderr, (env, p.stderr[-300:])


def test_cli_out_shards_goes_with_gpus_and_with_an_input_that_can_be_cut(tmp_path):
    """`--out-shards PREFIX` writes a file per device: without `--gpus N` it is refused before anything touches the GPU, and with
    an input that cannot be cut into ranges (SAM text, a file of a few blocks) it says so instead of writing to stdout."""
    import subprocess
    gold = os.path.join(ROOT, "tests", "golden")
    fade = os.path.join(ROOT, "fade_amd", "fade")
    p = subprocess.run([fade, "annotate", "--out-shards", str(tmp_path / "s"), "-b", os.path.join(gold, "anno_c1.sam"), os.path.join(gold, "anno_c1.fa")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 1 and p.stdout == b"" and b"--out-shards PREFIX writes one file per device" in p.stderr
    p = subprocess.run([fade, "annotate", "--gpus", "2", "--out-shards", str(tmp_path / "s"), "-b", os.path.join(gold, "anno_c1.sam"), os.path.join(gold, "anno_c1.fa")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert p.returncode == 1 and p.stdout == b"" and b"--out-shards needs an input that can be cut into ranges" in p.stderr
    assert not list(tmp_path.iterdir())


def test_product_path_does_not_touch_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fade_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "fade_oracle" not in txt and "libfadeoracle" not in txt, f


def test_synth_is_deterministic_and_well_formed():
    from fade_amd import synth
    cfg, g, b = synth.make_config("C2", 5000, contig_len=120_000)
    _, _, b2 = synth.make_config("C2", 5000, contig_len=120_000)
    for k in ("pos", "flag", "cigar_ops", "seq_packed", "qual"):
        assert np.array_equal(b[k], b2[k])
    n = len(b["pos"])
    assert b["cigar_off"][-1] == len(b["cigar_ops"])
    mapped = (b["flag"] & 4) == 0
    # query-consuming ops add up to l_seq
    for i in np.nonzero(mapped)[0][:500]:
        ops = b["cigar_ops"][b["cigar_off"][i]:b["cigar_off"][i + 1]]
        assert sum(int(o) >> 4 for o in ops if (int(o) & 15) in (0, 1, 4, 7, 8)) == int(b["l_seq"][i])
    frac = ((b["cigar_off"][1:] - b["cigar_off"][:-1]) > 1).mean()
    assert 0.07 < frac < 0.13  # p_sc = 0.10
    sub = synth.take(b, np.array([3, 10, 11]))
    assert np.array_equal(sub["pos"], b["pos"][[3, 10, 11]])
    assert np.array_equal(sub["seq_packed"][:75], b["seq_packed"][3 * 75:4 * 75])


def test_format_tags_against_oracle_strings(oracle):
    """fade_amd.format_tags (analysis.d:84-92,108-118) fed with the oracle's own alignment must give
    the oracle's am/as/ar/ab — host formatting logic checked without a GPU."""
    from fade_amd import format_tags, synth
    from fade_amd._lib import ALN_DTYPE
    cfg, g, b = synth.make_config("C2", 3000, contig_len=150_000)
    G = oracle.GenomeHolder(g.names, [a.tobytes() for a in g.ascii_contigs()])
    reads, keep = oracle.make_reads(b)
    contigs = [a.tobytes() for a in g.ascii_contigs()]
    aln, rs_all, exp = [], np.zeros(len(b["pos"]), np.uint8), {}
    for i in range(len(b["pos"])):
        a = oracle.annotate_one(G, reads[i], cfg["floor_len"], cfg["window"])
        rs_all[i] = a["rs"]
        if not a["has_tags"]:
            continue
        exp[i] = dict(rs=a["rs"], am=a["am"], as_=a["as_"], ar=a["ar"], ab=a["ab"])
        # rebuild the alignment record the device would return, from the oracle's SW
        ops = b["cigar_ops"][b["cigar_off"][i]:b["cigar_off"][i + 1]]
        cl, cr = oracle.parse_clips([int(o) for o in ops])
        al = sum(int(o) >> 4 for o in ops if (int(o) & 15) in (0, 2, 3, 7, 8))
        pos, tid = int(b["pos"][i]), int(b["tid"][i])
        start = max(pos - cfg["window"], 0)
        end = min(pos + al + cfg["window"], len(contigs[tid]))
        q = oracle.reverse_complement_packed(b["seq_packed"][b["seq_off"][i]:b["seq_off"][i + 1]], int(b["l_seq"][i]))
        r = oracle.sw(q, contigs[tid][start:end])
        rec = np.zeros(1, dtype=ALN_DTYPE)[0]
        rec["read_idx"], rec["art"], rec["win_start"], rec["win_len"] = i, (a["rs"] >> 1) & 3, start, end - start
        rec["clip_left"], rec["clip_right"], rec["aligned_len"] = cl >> 4, cr >> 4, al
        for k in ("score", "end_query", "end_ref", "beg_query", "beg_ref", "n_ops"):
            rec["sw"][k] = r[k]
        rec["sw"]["ops"][:len(r["ops"])] = r["ops"]
        aln.append(rec)
    got = format_tags(b, g.names, rs_all, np.array(aln, dtype=ALN_DTYPE))
    assert len(exp) > 50 and got == exp


def test_compact_sequences_keeps_exactly_the_records_the_gate_reads():
    """Context.compact_sequences: mapped records with an S op keep their packed bases, every other slice is empty."""
    from fade_amd import synth
    from fade_amd.api import Context
    g = synth.Genome(1, 200_000, 3)
    b = synth.make_reads(g, 5000, 9, read_len=101, window=100, p_sc=0.3)
    c = Context.compact_sequences(b)
    co, so, cs = b["cigar_off"].astype(np.int64), b["seq_off"].astype(np.int64), c["seq_off"].astype(np.int64)
    kept = 0
    for i in range(len(b["pos"])):
        ops = b["cigar_ops"][co[i]:co[i + 1]]
        need = bool(((ops & 15) == 4).any()) and not (int(b["flag"][i]) & 4)
        ln = cs[i + 1] - cs[i]
        assert ln == (so[i + 1] - so[i] if need else 0), i
        if need:
            assert np.array_equal(c["seq_packed"][cs[i]:cs[i + 1]], b["seq_packed"][so[i]:so[i + 1]])
            kept += 1
    assert 0 < kept < len(b["pos"]) // 2 and cs[-1] == len(c["seq_packed"])


def test_with_bounds_describes_the_records_that_carry_bases():
    """Context.with_bounds (the ABI-3 hints a packing thread passes): every record stays, only the ones the gate can
    align keep their bases, and the bounds are those records' count, read lengths and longest alignedLength."""
    from fade_amd import synth
    from fade_amd.api import Context
    g = synth.Genome(1, 200_000, 3)
    b = synth.make_reads(g, 4000, 11, read_len=101, window=100, p_sc=0.25)
    c = Context.with_bounds(b)
    assert len(c["pos"]) == len(b["pos"]) and c["n_skipped"] == 0
    co = b["cigar_off"].astype(np.int64)
    need = np.array([bool(((b["cigar_ops"][co[i]:co[i + 1]] & 15) == 4).any()) and not (int(b["flag"][i]) & 4)
                     for i in range(len(b["pos"]))])
    assert c["n_with_seq"] == int(need.sum()) > 0 and c["l_seq_min"] == c["l_seq_max"] == 101
    al = [sum(int(o) >> 4 for o in b["cigar_ops"][co[i]:co[i + 1]] if (int(o) & 15) in (0, 2, 3, 7, 8)) for i in np.nonzero(need)[0]]
    assert c["ref_span_bound"] == max(al)


def test_corrupt_bam_and_bgzf_inputs_are_rejected_under_sanitizers():
    """host/hts_lite.hpp against truncated / crafted records and blocks, built with ASan + UBSan
    (fade_amd/csrc/host/selftest/hts_selftest.cpp): aux fields that overrun the record, 'B' arrays with huge counts,
    an rs tag cut short, BSIZE smaller than the block's own header, ISIZE beyond 64 KiB."""
    import subprocess
    csrc = os.path.join(ROOT, "fade_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "-s", "build/hts_selftest"])
    p = subprocess.run([os.path.join(csrc, "build", "hts_selftest")], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stdout.decode() + p.stderr.decode()
    assert b"all corrupt-input cases rejected cleanly" in p.stdout


def test_ref_consuming_op_set_has_one_definition():
    """dhtslib's Cigar.alignedLength (M, D, N, =, X) is one constant in include/fadehip.h, mirrored once in _lib.py."""
    from fade_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "fadehip.h")).read()
    mask = int(re.search(r"#define FADEHIP_REF_CONSUMING_OPS (0x[0-9A-Fa-f]+)u", hdr).group(1), 16)
    assert tuple(k for k in range(16) if (mask >> k) & 1) == _lib.REF_CONSUMING_OPS == (0, 2, 3, 7, 8)
    for path in ("fade_amd/csrc/fadehip_kernels.hpp", "fade_amd/csrc/fadehip.hip"):
        assert "op == 0 || op == 2 || op == 3" not in open(os.path.join(ROOT, path)).read(), path


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/fadehip.h is a C header (C99, -pedantic): a C translation unit that takes the address of every declared
    entry point compiles and links against libfadehip.so — the boundary a D / cgo / ctypes binding sees."""
    import subprocess
    src = tmp_path / "abi_c.c"
    names = _declared_symbols()
    src.write_text('#include "fadehip.h"\n#include <stdio.h>\nint main(void) {\n    void *p[] = {' +
                   ", ".join("(void *)%s" % n for n in names) +
                   '};\n    fadehip_params prm;\n    fadehip_params_default(&prm);\n'
                   '    printf("%d %d %u %d\\n", fadehip_abi_version(), (int)(sizeof p / sizeof p[0]), prm.rules, (int)sizeof(fadehip_aln));\n    return 0;\n}\n')
    exe = tmp_path / "abi_c"
    lib_dir = os.path.join(ROOT, "fade_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-Wno-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", lib_dir, "-lfadehip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, timeout=60).stdout.decode().split()
    assert out == ["3", str(len(names)), "127", "120"]
