// bgzf_deflate.hpp — BGZF (SAM spec §4.1) compression on gfx950: one 512-thread workgroup per BGZF block of up to 0x7f00
// input bytes, the whole block resident in LDS (80 KB: TWO workgroups per CU, 512 blocks in flight).  A block's time is
// set by the serial roles of phase A (one wave each), so what a CU gains from a second block is a second pipeline; with
// 0xff00-byte blocks (160 KB, one workgroup per CU: round 3's first form) the same kernel ran at 9.8 GB/s.  The smaller
// footprint also lets other kernels (the inflater, the record kernels of the file path) share a CU with a compressor.
//
// What it replaces: htslib's bgzf_write -> zlib deflate behind `SAMWriter(..., SAMWriterTypes.BAM)` (source/util.d:65-76),
// i.e. the serialised write at source/anno.d:47-49 — 7 of the 12.7 core-seconds `fade annotate` spent per 10 M reads.
//
// Per block (host/selftest/gpu_deflate_model.cpp is the same algorithm on the CPU, checked with zlib's inflate):
//   A  matches.  Pieces of 64 positions, a wave each, take turns at the hash heads (4-way buckets of 16-bit positions, 4 Ki
//      buckets): a piece's lookups see every earlier piece's inserts.  Nearer than that, distances 1..8 are tried directly
//      (runs, short periods).  The up to five candidates of a position are extended side by side in LDS; the wave then
//      waits for its turn to parse its piece greedily (a match yields to a longer one at the next position), on 64-bit
//      lane masks in scalar registers: token bitmap, match bitmap, match records.
//   B  symbol histograms (8 sub-histograms against same-address LDS atomics), minimum-redundancy code lengths (Moffat &
//      Katajainen in place, the array spread over the lanes of a wave), 15-bit limit, canonical codes.
//   C  the dynamic-block header, one wave (lane arrays again), while the others count their tokens' bits.
//   D  1024 position ranges emit their tokens at scanned bit offsets straight into the block's output slot; a word that
//      two ranges share is OR-ed atomically.  A block that would not shrink is stored.
//   CRC-32 of the input by slicing-by-4 over 1024 pieces, combined with the x^(8n) mod P arithmetic of bgzf_huff.hpp.
// A second kernel scans the block sizes and a third assembles the BGZF members (header, payload, CRC32, ISIZE) into one
// contiguous byte stream: what goes to the file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bgzf_huff.hpp"

// Two geometries of the same kernels.  64: htslib's 0xff00-byte blocks, the block and its tables fill the CU's LDS, one
// workgroup per CU — the smallest output (a block's dynamic-Huffman header and its cold start weigh half as much).  32:
// 0x7f00-byte blocks, 80 KB of LDS, two workgroups per CU — a block's time is set by the serial roles of phase A, so a CU
// gains a second pipeline (9.8 -> 14.6 GB/s, 18 GB/s in the kernel) and other kernels can share a CU with a compressor;
// the output is 0.5 % larger on uniform-quality BAM payload and 1 % larger where qualities run (there a member shrinks
// to 8 KB and its header shows).  fadehip.hip picks per call: 32 while the stream is mostly incompressible bases
// (ratio above 0.45), 64 otherwise, so that the output stays below zlib -6's in both regimes.
#define FADEHIP_BGZF_GEOM 64
#define FADEHIP_BGZF_NS bgzf64
#include "bgzf_deflate_body.hpp"
#undef FADEHIP_BGZF_GEOM
#undef FADEHIP_BGZF_NS
#define FADEHIP_BGZF_GEOM 32
#define FADEHIP_BGZF_NS bgzf32
#include "bgzf_deflate_body.hpp"
#undef FADEHIP_BGZF_GEOM
#undef FADEHIP_BGZF_NS

namespace fadehip {
namespace bgzf {
using bgzf64::claim_ticket;  // (the inflater draws its tickets the same way)
}
}  // namespace fadehip
