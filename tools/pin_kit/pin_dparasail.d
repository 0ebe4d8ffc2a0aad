#!/usr/bin/env dub
/+ dub.sdl:
    name "pin_dparasail"
    dependency "dparasail" version="~>0.3.3"
    dependency "dhtslib" repository="git+https://github.com/blachlylab/dhtslib.git" version="c51b842166300796944f786b00f60345face9d48"
+/
// pin_dparasail.d — NOT COMPILED OR RUN IN THE BUILD ENVIRONMENT (no D toolchain, no dparasail / dhtslib there).
// The calls are FADE's own (source/anno.d:36, source/analysis.d:67-113), so whatever this prints IS what FADE sees:
//   auto p = Parasail("ACTGN", 10, 2, 2, -3);  auto res = p.sw_striped(q, r);  res.score, res.position, res.cigar
// Output: the format of tests/golden/sw_pairs.tsv (end_query / end_ref / beg_query are not exposed by this API version as
// far as recollected: they are printed as -1 and compare.py skips them), then the alignedLength probe.
//   dub run --single pin_dparasail.d -- tests/golden/sw_pairs.tsv
import std.stdio, std.array, std.algorithm, std.conv, std.string;
import dparasail;
import dhtslib.cigar;

void main(string[] args)
{
    auto p = Parasail("ACTGN", 10, 2, 2, -3);
    writeln("#query\tref\tscore\tend_query\tend_ref\tbeg_query\tposition\tn_ops\tcigar_front16");
    foreach (line; File(args[1]).byLineCopy)
    {
        if (line.length == 0 || line[0] == '#') continue;
        auto f = line.split('\t');
        auto res = p.sw_striped(f[0], f[1]);
        auto ops = res.cigar.ops;                       // (dhtslib Cigar: the ops FADE indexes, analysis.d:74-80)
        string front;
        foreach (i, op; ops) if (i < 16) front ~= op.length.to!string ~ cast(char) "MIDNSHP=XB"[op.op];
        writefln("%s\t%s\t%d\t-1\t-1\t-1\t%d\t%d\t%s", f[0], f[1], res.score, res.position, ops.length, front);
    }
    // analysis.d:53,111-113 and filter.d:27,61 rest on this:
    stderr.writefln("alignedLength probe: Cigar(\"10M2I5M3D4M\").alignedLength = %d   (22: M,D,N,=,X as this build assumes; 24: I counted too)",
        cigarFromString("10M2I5M3D4M").alignedLength);
}
