/// Batched replacement of the loop at source/anno.d:44-50 over the fadehip C ABI.
///
/// NOT COMPILED IN THIS REPOSITORY'S ENVIRONMENT (no D compiler in the image): a faithful sketch of the
/// integration a FADE maintainer would write, kept next to the binding so that the seam is concrete.  The C++
/// driver fade_amd/csrc/host/fade_main.cpp is the tested implementation of exactly this control flow.
module anno_hip;

import std.algorithm : min, max;
import std.array : appender;
import std.conv : to;
import std.exception : enforce;
import std.string : fromStringz;
import dhtslib;
import htslib.hts_log;
import htslib.sam : bam_get_cigar, bam_get_seq;
import fadehip;
import readstatus;
import util;

/// drop-in for `annotate` (source/anno.d:16-52): same arguments, same output
int annotateHip(string cl, string[] args, ubyte con, int artifact_floor_length, int align_buffer_size,
        int batchReads = 1 << 18)
{
    hts_set_log_level(htsLogLevel.HTS_LOG_INFO);
    hts_log_warning("fade annotate", "Output SAM/BAM will not be sorted (regardless of prior sorting)");
    auto bam = SAMReader(args[1]);
    auto fai = IndexedFastaFile(args[2]);
    auto header = bam.header.dup;
    header.addLine(RecordType.PG, "ID", "fade-annotate", "PN", "fade", "VN", VERSION, "PP",
            header.valueByPos(RecordType.PG, header.numRecords(RecordType.PG) - 1, "ID"), "CL", cl);
    auto out_bam = getWriter(con, header);

    // Parasail("ACTGN", 10, 2, 2, -3)  (anno.d:36) -> fadehip_params defaults
    fadehip_ctx* ctx;
    fadehip_params prm;
    fadehip_params_default(&prm);
    prm.max_batch_reads = batchReads;
    enforce(fadehip_create(&ctx, 0, &prm) == 0, fadehip_last_error(null).fromStringz);
    scope (exit)
        fadehip_destroy(ctx);

    // the FASTA that analysis.d:63 fetched per clip under a mutex is uploaded once
    long[] lengths;
    string[] contigs;
    const(ubyte)*[] ptrs;
    foreach (tid; 0 .. bam.header.nTargets)
    {
        auto len = bam.header.targetLength(tid);
        contigs ~= fai.fetchSequence(bam.header.targetName(tid).idup, ZBHO(0, len));
        lengths ~= len;
        ptrs ~= cast(const(ubyte)*) contigs[$ - 1].ptr;
    }
    enforce(fadehip_genome_upload(ctx, cast(int) lengths.length, lengths.ptr, ptrs.ptr) == 0,
            fadehip_last_error(ctx).fromStringz);

    SAMRecord[] chunk;
    int[] tid, pos, lseq;
    ushort[] flag;
    ubyte[] hasSa, seq, rs;
    uint[] cigarOff, cigarOps, seqOff;
    fadehip_aln[] aln;

    void flush()
    {
        if (chunk.length == 0)
            return;
        // pack: straight copies out of each bam1_t, no re-encoding
        tid.length = pos.length = lseq.length = flag.length = hasSa.length = chunk.length;
        cigarOff.length = seqOff.length = chunk.length + 1;
        cigarOps.length = 0;
        seq.length = 0;
        foreach (i, rec; chunk)
        {
            tid[i] = rec.tid;
            pos[i] = cast(int) rec.pos.pos;
            lseq[i] = rec.length;
            flag[i] = rec.flag;
            hasSa[i] = rec["SA"].exists ? 1 : 0; // anno.d:73
            cigarOff[i] = cast(uint) cigarOps.length;
            seqOff[i] = cast(uint) seq.length;
            cigarOps ~= bam_get_cigar(rec.b)[0 .. rec.b.core.n_cigar];
            seq ~= bam_get_seq(rec.b)[0 .. (rec.length + 1) / 2];
        }
        cigarOff[$ - 1] = cast(uint) cigarOps.length;
        seqOff[$ - 1] = cast(uint) seq.length;
        rs.length = chunk.length;
        aln.length = chunk.length;
        // (the arrays may also live in ONE pinned block laid out by fadehip_batch_bind, records without an S op carrying
        // no bases, and the ABI-3 bounds n_with_seq / l_seq_min / l_seq_max filled in -- what fade_main.cpp's pack_chunk
        // does; this sketch keeps the simple form: every record with its bases, arrays from anywhere, bounds unknown = 0)
        fadehip_read_batch b = {
            cast(int) chunk.length, tid.ptr, pos.ptr, flag.ptr, hasSa.ptr, lseq.ptr, cigarOff.ptr,
            cigarOps.ptr, seqOff.ptr, seq.ptr, 0, 0, 0, 0, 0, 0
        };
        fadehip_anno_out o;
        o.rs = rs.ptr;
        o.aln = aln.ptr;
        o.aln_cap = cast(int) aln.length;
        enforce(fadehip_annotate_submit(ctx, 0, &b, artifact_floor_length, align_buffer_size) == 0,
                fadehip_last_error(ctx).fromStringz);
        enforce(fadehip_annotate_collect(ctx, 0, &o) == 0, fadehip_last_error(ctx).fromStringz);

        foreach (i, rec; chunk)
            rec["rs"] = rs[i]; // anno.d:63,94
        foreach (ref a; aln[0 .. o.n_aln])
        {
            if (!a.art)
                continue;
            auto rec = chunk[a.read_idx];
            auto cigar = Cigar(cast(CigarOp[]) a.sw.ops[0 .. a.sw.n_ops]);
            auto q_seq = reverse_complement_sam_record(rec).idup;
            auto start = a.win_start;
            string[4] l, r;
            if (a.art & 1) // analysis.d:84-92
            {
                auto clips = parse_clips(cigar);
                auto apos = start + a.sw.beg_ref;
                auto overlap = apos >= rec.pos - a.clip_left ? apos - (rec.pos - a.clip_left) : 0;
                auto plen = min(cast(long) rec.length, (rec.length - clips[0].length) + overlap);
                l = [
                    rec.h.targetName(rec.tid).idup ~ "," ~ apos.to!string ~ "," ~ cigar.toString,
                    rec.sequence[0 .. plen].idup, q_seq[$ - plen .. $],
                    rec.qscoresPhredScaled[0 .. plen].idup
                ];
            }
            if (a.art & 2) // analysis.d:108-118
            {
                auto clips = parse_clips(cigar);
                auto apos = start + a.sw.beg_ref;
                auto lhs = rec.pos + a.aligned_len + a.clip_right;
                auto rhs = apos + cigar.alignedLength;
                auto overlap = lhs >= rhs ? lhs - rhs : 0;
                auto plen = min(cast(long) rec.length, (rec.length - clips[1].length) + overlap);
                r = [
                    rec.h.targetName(rec.tid).idup ~ "," ~ apos.to!string ~ "," ~ cigar.toString,
                    rec.sequence[$ - plen .. $].idup, q_seq[0 .. plen],
                    rec.qscoresPhredScaled[$ - plen .. $].idup
                ];
            }
            rec["am"] = l[0] ~ ";" ~ r[0]; // anno.d:100
            rec["as"] = l[1] ~ ";" ~ r[1]; // anno.d:102
            rec["ar"] = l[2] ~ ";" ~ r[2]; // anno.d:104
            rec["ab"] = l[3] ~ ";" ~ r[3]; // anno.d:106
        }
        foreach (rec; chunk)
            out_bam.write(rec); // anno.d:47-49
        chunk.length = 0;
    }

    foreach (rec; bam.allRecords)
    {
        chunk ~= rec;
        if (chunk.length == batchReads)
            flush();
    }
    flush();
    return 0;
}
