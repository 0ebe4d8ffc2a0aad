// fadehip.hip — C ABI (include/fadehip.h) over the gfx950 kernels in fadehip_kernels.hpp.
// Host side of the drop-in boundary for source/anno.d:44-50 / source/analysis.d:67.
// No CPU fallback lives here: every entry point either runs the HIP path or returns an error.
#include "fadehip_kernels.hpp"
#include <rccl/rccl.h>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace fadehip;

namespace {

thread_local std::string g_err = "";

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct Slot {
    hipStream_t stream = nullptr;
    DevBuf tid, pos, lseq, flag, has_sa, cigar_off, cigar_ops, seq_off, seq;
    DevBuf rs, fwd, aln, trace;
    DevBuf ckpt, cand, incomplete;  // two-pass path
    // All small counters of a run live in one block so that one memset clears them and one copy reads the gate's.
    // Counters that different kernels (or different atomics of one kernel) hammer sit in different 128-byte lines:
    // same-line atomics serialise in one L2 channel (the gate kernel took 60 instead of 49 us with them packed).
    //   [0,84) counters | [128,512) counters64, one line each | [512 + 48 c, ...) selection counters of class c |
    //   [1024, 1536) stats: STAT_PARTS partial sums of the 8 stats.d counters
    DevBuf zblock;
    static constexpr size_t ZB_COUNTERS = 0, ZB_C64 = 128, ZB_GATE_BYTES = 512, ZB_SEL = 512, ZB_SEL_STRIDE = 48,
                            ZB_STATS = 1024, ZB_BYTES = 1024 + 8 * 8 * STAT_PARTS;
    unsigned long long *d_counters64() const { return (unsigned long long *)((uint8_t *)zblock.p + ZB_C64); }
    unsigned long long *d_stats() const { return (unsigned long long *)((uint8_t *)zblock.p + ZB_STATS); }
    uint32_t *d_counters() const { return (uint32_t *)((uint8_t *)zblock.p + ZB_COUNTERS); }
    uint32_t *d_sel(int cls) const { return (uint32_t *)((uint8_t *)zblock.p + ZB_SEL + ZB_SEL_STRIDE * (size_t)cls); }
    bool sel_fresh[NUM_CLASSES] = {};  // class's selection counters were cleared by the run's memset and not used yet
    uint32_t *h_sel = nullptr;                   // pinned: NUM_BUCKETS + 1
    DevBuf work[NUM_LISTS], meta[NUM_LISTS];
    DevBuf lrows;  // sw_long_kernel: previous-row H and F-hat
    uint8_t *h_gate = nullptr;                 // pinned: the first ZB_GATE_BYTES of zblock after the gate
    uint32_t *h_counters = nullptr;            // view into h_gate: 2*NC+1
    unsigned long long *h_counters64 = nullptr;  // view into h_gate: 3
    unsigned long long *h_stats = nullptr;       // pinned: 8 * STAT_PARTS partial sums
    int n_reads = 0;
    int state = 0;  // 0 idle, 1 uploaded, 2 ran
    int n_aln = 0;
    std::vector<hipEvent_t> ev;  // event pool
    int ev_used = 0;
    // (start,end) event index pairs of the last run
    std::vector<std::pair<int, int>> fwd_spans, tb_spans;
    int ev_gate0 = -1, ev_gate1 = -1, ev_end = -1;
    int64_t prof_counts[4] = {0, 0, 0, 0};
    int n_fwd_launches = 0;
    int64_t n_cand = 0, n_rerun = 0;  // two-pass: candidates traced / candidates re-run from column 0
};

}  // namespace

struct fadehip_ctx {
    int device = 0;
    fadehip_params prm;
    ScoreTab sc;
    std::string err;
    Slot slots[FADEHIP_NUM_SLOTS];
    // genome
    DevBuf genome, contig_len, contig_base;
    DevBuf l1_q, l1_r, l1_qn, l1_rn, l1_bad, l1_work, l1_aln;  // level 1 (fadehip_sw_batch): kept between calls, grow only
    int n_contigs = 0;
    std::vector<int64_t> h_contig_len;
    std::vector<uint64_t> h_contig_base;
    int cu_count = 0;
    // FADEHIP_KERNEL = twopass (default) | pk (single-pass packed int16) | int32 (single-pass int32): A/B runs
    bool use_packed = true;
    bool two_pass = true;
    int span_slack = 24;  // FADEHIP_SPAN_SLACK overrides (tests: -1 makes almost every path leave its range)
};

namespace {

int set_err(fadehip_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_err = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return set_err(ctx, e_ == hipErrorOutOfMemory ? FADEHIP_E_NOMEM : FADEHIP_E_HIP,      \
                           "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int reserve(fadehip_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap && b.p) return 0;
    if (b.p) {
        HIPCHK(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = std::max<size_t>(bytes, 256);
    want = (want + 255) & ~(size_t)255;
    HIPCHK(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}

void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

int build_score_tab(fadehip_ctx *ctx, const fadehip_params &p, ScoreTab &sc) {
    if (p.open <= 0 || p.ext <= 0 || p.ext > p.open)
        return set_err(ctx, FADEHIP_E_UNSUPPORTED, "gap penalties must satisfy 0 < ext <= open (got open=%d ext=%d)", p.open, p.ext);
    const int lo = std::min(std::min(p.match, p.mismatch), 0) + p.open;
    const int hi = std::max(std::max(p.match, p.mismatch), 0) + p.open;
    if (lo < 0 || hi > 15)
        return set_err(ctx, FADEHIP_E_UNSUPPORTED,
                       "scores + open must fit 4 bits for the profile registers (match=%d mismatch=%d open=%d)",
                       p.match, p.mismatch, p.open);
    for (int cq = 0; cq < 8; cq++) {
        uint32_t v = 0;
        for (int cr = 0; cr < 8; cr++) {
            int w;
            if (cq >= PAD_CLASS || cr >= PAD_CLASS) w = -p.open;  // pad row/column: W' = 0
            else if (cq == 5 || cr == 5) w = 0;                   // parasail wildcard
            else w = (cq == cr) ? p.match : p.mismatch;           // N vs N scores `match` (Appendix A.1)
            v |= (uint32_t)(w + p.open) << (4 * cr);
        }
        sc.prof[cq] = v;
    }
    sc.open = p.open;
    sc.ext = p.ext;
    sc.match = p.match;
    sc.mismatch = p.mismatch;
    return 0;
}

void fill_ascii_table(uint8_t t[256]) {
    memset(t, 0, 256);
    const char *s = "=ACMGRSVTWYHKDBN";
    for (int k = 1; k < 16; k++) {
        t[(unsigned char)s[k]] = (uint8_t)k;
        t[(unsigned char)(s[k] | 0x20)] = (uint8_t)k;  // analysis.d:63 upper-cases the window
    }
}

int new_event(fadehip_ctx *ctx, Slot &s, int *idx) {
    if (s.ev_used == (int)s.ev.size()) {
        hipEvent_t e;
        HIPCHK(ctx, hipEventCreate(&e));
        s.ev.push_back(e);
    }
    *idx = s.ev_used++;
    return 0;
}

int record(fadehip_ctx *ctx, Slot &s, int *idx) {
    int rc = new_event(ctx, s, idx);
    if (rc) return rc;
    HIPCHK(ctx, hipEventRecord(s.ev[*idx], s.stream));
    return 0;
}

template <int C>
int launch_forward_c(fadehip_ctx *ctx, int cls, const SwArgs &a, int quads, size_t lds, hipStream_t st, bool packed) {
    if constexpr (C >= NUM_CLASSES) {
        return set_err(ctx, FADEHIP_E_INVALID, "bad class %d", cls);
    } else {
        if (cls == C) {
            constexpr int R = class_rows(C);
            if (packed)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_pk_kernel<R, 0>), dim3(quads), dim3(64), lds, st, a);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_forward_kernel<R>), dim3(quads), dim3(64), lds, st, a);
            HIPCHK(ctx, hipGetLastError());
            return 0;
        }
        return launch_forward_c<C + 1>(ctx, cls, a, quads, lds, st, packed);
    }
}

template <int C>
int launch_pk_mode(fadehip_ctx *ctx, int cls, int mode, const SwArgs &a, int octets, size_t lds, hipStream_t st) {
    if constexpr (C >= NUM_CLASSES) {
        return set_err(ctx, FADEHIP_E_INVALID, "bad class %d", cls);
    } else {
        if (cls == C) {
            constexpr int R = class_rows(C);
            if (mode == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_pk_kernel<R, 1>), dim3(octets), dim3(64), lds, st, a);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_pk_kernel<R, 2>), dim3(octets), dim3(64), lds, st, a);
            HIPCHK(ctx, hipGetLastError());
            return 0;
        }
        return launch_pk_mode<C + 1>(ctx, cls, mode, a, octets, lds, st);
    }
}

__global__ void make_cand_back_kernel(const Cand *in, int n, int back, Cand *out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) {
        Cand c = in[k];
        c.c0 = (int)c.c0 > back ? c.c0 - (uint32_t)back : 0u;
        out[k] = c;
    }
}

// Two-pass path for one class list: pass 1 scores every alignment and leaves H/E checkpoints
// every CK_COLS columns; the selection keeps the alignments that can still become an artifact call; pass 2
// re-computes, with trace, only columns [c0, end_ref] of each candidate and walks the traceback; a candidate
// whose path leaves that range on the left is re-run from column 0.
int run_class_two_pass(fadehip_ctx *ctx, Slot &s, hipStream_t st, int cls, const Work *work, const Meta *meta,
                       int n_items, int max_lr, const uint8_t *q_nib, const uint8_t *r_nib, fadehip_aln *out,
                       uint8_t *rs, int floor_len, int gate, int64_t budget, bool timed) {
    const int R = class_rows(cls);
    const int n_ck = (max_lr + 15 + CK_COLS - 1) / CK_COLS;
    const uint64_t ck_stride = (uint64_t)n_ck * ck_dwords(R) * 64;  // dwords per pass-1 octet
    auto strides = [&](int steps, int *n_blocks, uint64_t *stride, int *ref_stride, size_t *lds) {
        *n_blocks = (steps + 3) / 4;
        *stride = (uint64_t)*n_blocks * R * 64;
        *ref_stride = (((*n_blocks * 4) * 2 + 15) / 16) * 16;
        *lds = (size_t)*ref_stride * 4;
    };
    int nb1, ref_stride1;
    uint64_t stride_unused;
    size_t lds1;
    strides(max_lr + 15, &nb1, &stride_unused, &ref_stride1, &lds1);
    if (lds1 > 64 * 1024)
        return set_err(ctx, FADEHIP_E_UNSUPPORTED, "reference window of %d bases needs %zu B LDS per wave (max 64 KiB)", max_lr, lds1);
    // chunk so that the checkpoints of a chunk fit half the budget (the other half is pass-2 trace scratch)
    const int64_t ck_bytes = (int64_t)ck_stride * 4;
    const int total_oct = (n_items + 7) / 8;
    const int64_t chunk_oct = std::max<int64_t>(1, std::min<int64_t>(total_oct, (budget / 2) / std::max<int64_t>(ck_bytes, 1)));
    int rc;
    if ((rc = reserve(ctx, s.ckpt, (size_t)(chunk_oct * ck_bytes))) || (rc = reserve(ctx, s.fwd, (size_t)n_items * sizeof(Fwd))) ||
        (rc = reserve(ctx, s.zblock, Slot::ZB_BYTES)))
        return rc;
    uint32_t *const sel_counters = s.d_sel(cls);
    for (int64_t o0 = 0; o0 < total_oct; o0 += chunk_oct) {
        const int octs = (int)std::min<int64_t>(chunk_oct, total_oct - o0);
        const int i0 = (int)(o0 * 8);
        const int n = std::min(n_items - i0, octs * 8);
        if ((rc = reserve(ctx, s.cand, sizeof(Cand) * (size_t)NUM_BUCKETS * (size_t)n)) ||
            (rc = reserve(ctx, s.incomplete, sizeof(Cand) * (size_t)n)))
            return rc;
        if (!s.sel_fresh[cls]) HIPCHK(ctx, hipMemsetAsync(sel_counters, 0, sizeof(uint32_t) * (NUM_BUCKETS + 1), st));
        s.sel_fresh[cls] = false;
        SwArgs a;
        a.work = work + i0;
        a.n_items = n;
        a.q_nib = q_nib;
        a.r_nib = r_nib;
        a.trace = nullptr;
        a.quad_stride = 0;
        a.ref_stride = ref_stride1;
        a.fwd = (Fwd *)s.fwd.p + i0;
        a.sc = ctx->sc;
        a.cand = nullptr;
        a.ckpt = (uint32_t *)s.ckpt.p;
        a.ck_stride = ck_stride;
        a.n_ck = n_ck;
        int e0 = -1, e1 = -1, e2 = -1;
        if (timed && (rc = record(ctx, s, &e0))) return rc;
        if ((rc = launch_pk_mode<0>(ctx, cls, 1, a, octs, lds1, st))) return rc;
        if (timed && (rc = record(ctx, s, &e1))) return rc;
        SelArgs sel;
        sel.work = work + i0;
        sel.meta = meta ? meta + i0 : nullptr;
        sel.fwd = (const Fwd *)s.fwd.p + i0;
        sel.n_items = n;
        sel.floor_len = floor_len;
        sel.trace_all = ctx->prm.trace_all;
        sel.R = R;
        sel.span_slack = ctx->span_slack;
        sel.cand = (Cand *)s.cand.p;
        sel.cap = (uint32_t)n;
        sel.bucket_n = sel_counters;
        sel.out = out + i0;
        sel.q_nib = q_nib;
        sel.r_nib = r_nib;
        sel.rs = rs;
        sel.stats = (rs && gate) ? s.d_stats() : nullptr;
        sel.gate = gate;
        sel.match = getenv("FADEHIP_NO_SHORTCUT") ? 0 : ctx->sc.match;
        hipLaunchKernelGGL(select_kernel, dim3((n + SELECT_BLOCK - 1) / SELECT_BLOCK), dim3(SELECT_BLOCK), 0, st, sel);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(s.h_sel, sel_counters, sizeof(uint32_t) * NUM_BUCKETS, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        uint32_t bucket_n[NUM_BUCKETS];
        for (int b = 0; b < NUM_BUCKETS; b++) bucket_n[b] = s.h_sel[b];
        // pass 2: every bucket in ONE forward launch and ONE traceback launch (separate launches per bucket
        // each run a fraction of a wave-round and pay their own ramp and tail)
        auto pass2 = [&](const Cand *cand, uint32_t cap, const uint32_t *counts, const int *steps_max, int nb,
                         bool may_be_incomplete) -> int {
            P2Table tab;
            memset(&tab, 0, sizeof tab);
            tab.cap = cap;
            uint64_t off = 0;
            uint32_t oct = 0;
            size_t lds2 = 0;
            int ref_stride2 = 16;
            for (int k = 0; k < NUM_BUCKETS; k++) {
                const int b = nb - 1 - k;  // longest bucket first
                tab.oct_first[k] = oct;
                if (b < 0 || !counts[b]) continue;
                int nb2, rs2;
                uint64_t st2;
                size_t l2;
                strides(steps_max[b], &nb2, &st2, &rs2, &l2);
                tab.count[k] = counts[b];
                tab.bucket[k] = (uint32_t)b;
                tab.trace_base[k] = off;
                tab.stride[k] = st2;
                const uint32_t octs_b = (counts[b] + 7) / 8;
                off += (uint64_t)octs_b * st2;
                oct += octs_b;
                lds2 = std::max(lds2, l2);
                ref_stride2 = std::max(ref_stride2, rs2);
            }
            tab.oct_first[NUM_BUCKETS] = oct;
            if (!oct) return 0;
            int r2 = reserve(ctx, s.trace, (size_t)off * 4);
            if (r2) return r2;
            SwArgs b2 = a;
            b2.n_items = 0;
            b2.cand = cand;
            b2.trace = (uint32_t *)s.trace.p;
            b2.quad_stride = 0;
            b2.ref_stride = ref_stride2;
            b2.tab = tab;
            if ((r2 = launch_pk_mode<0>(ctx, cls, 2, b2, (int)oct, lds2, st))) return r2;
            TbArgs t;
            t.work = work + i0;
            t.meta = meta ? meta + i0 : nullptr;
            t.fwd = (const Fwd *)s.fwd.p + i0;
            t.n_items = (int)oct * 8;
            t.R = R;
            t.q_nib = q_nib;
            t.r_nib = r_nib;
            t.trace = (const uint32_t *)s.trace.p;
            t.quad_stride = 0;
            t.sc = ctx->sc;
            t.out = out + i0;
            t.rs = rs;
        t.stats = (rs && gate) ? s.d_stats() : nullptr;
            t.stats = (rs && gate) ? s.d_stats() : nullptr;
            t.floor_len = floor_len;
            t.gate = gate;
            t.early_out = (gate && meta && !ctx->prm.trace_all) ? 1 : 0;
            t.packed = 1;
            t.cand = cand;
            t.incomplete = may_be_incomplete ? (Cand *)s.incomplete.p : nullptr;
            t.incomplete_n = sel_counters + NUM_BUCKETS;
            t.tab = tab;
            hipLaunchKernelGGL(traceback_kernel, dim3((oct * 8 + 63) / 64), dim3(64), 0, st, t);
            HIPCHK(ctx, hipGetLastError());
            s.prof_counts[2] += (int64_t)off * 4;
            return 0;
        };
        int steps_max[NUM_BUCKETS];
        for (int b = 0; b < NUM_BUCKETS; b++) steps_max[b] = std::min(bucket_cols(b), max_lr + 15);
        if ((rc = pass2((const Cand *)s.cand.p, (uint32_t)n, bucket_n, steps_max, NUM_BUCKETS, true))) return rc;
        // candidates whose path left the traced steps: from one snapshot further back (a lone wave is pure latency,
        // ~0.4 us per step, and most paths miss by a few columns), then from eight, then from step 0
        int n_inc = 0;
        for (int round = 0; round < 3; round++) {
            HIPCHK(ctx, hipMemcpyAsync(s.h_sel + NUM_BUCKETS, sel_counters + NUM_BUCKETS, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIPCHK(ctx, hipStreamSynchronize(st));
            const int m = (int)s.h_sel[NUM_BUCKETS];
            if (m == 0) break;
            n_inc += m;
            Cand *again = (Cand *)s.cand.p;  // the bucket lists are consumed, reuse their storage
            hipLaunchKernelGGL(make_cand_back_kernel, dim3((m + 255) / 256), dim3(256), 0, st, (const Cand *)s.incomplete.p, m,
                               round == 0 ? CK_COLS : (round == 1 ? 8 * CK_COLS : (1 << 30)), again);
            HIPCHK(ctx, hipGetLastError());
            HIPCHK(ctx, hipMemsetAsync(sel_counters + NUM_BUCKETS, 0, sizeof(uint32_t), st));
            const uint32_t cnt1[1] = {(uint32_t)m};
            const int st1[1] = {max_lr + 15};
            if ((rc = pass2(again, (uint32_t)m, cnt1, st1, 1, round < 2))) return rc;
        }
        s.n_rerun += n_inc;
        if (getenv("FADEHIP_DEBUG")) {
            fprintf(stderr, "[fadehip] class R=%d chunk n=%d: buckets", R, n);
            for (int b = 0; b < NUM_BUCKETS; b++) fprintf(stderr, " %u", bucket_n[b]);
            fprintf(stderr, " | re-run from step 0: %d\n", n_inc);
        }
        if (timed) {
            if ((rc = record(ctx, s, &e2))) return rc;
            s.fwd_spans.push_back({e0, e1});
            s.tb_spans.push_back({e1, e2});
        }
        s.n_fwd_launches++;
        for (int b = 0; b < NUM_BUCKETS; b++) s.n_cand += bucket_n[b];
    }
    return 0;
}

// Runs forward + traceback for one class list, chunked so the trace fits `trace`.
int run_class(fadehip_ctx *ctx, Slot &s, hipStream_t st, int cls, const Work *work, const Meta *meta, int n_items,
              int max_lr, const uint8_t *q_nib, const uint8_t *r_nib, fadehip_aln *out, uint8_t *rs, int floor_len,
              int gate, int64_t trace_budget, bool timed) {
    if (ctx->two_pass)
        return run_class_two_pass(ctx, s, st, cls, work, meta, n_items, max_lr, q_nib, r_nib, out, rs, floor_len, gate,
                                  trace_budget, timed);
    const int R = class_rows(cls);
    const bool packed = ctx->use_packed;
    const int per_wave = packed ? 8 : 4;  // alignments per wavefront
    const int n_blocks = (max_lr + 15 + 3) / 4;
    // dwords of trace per wave: 4 bits per cell slot either way
    const uint64_t quad_stride = (uint64_t)n_blocks * (packed ? R : R / 2) * 64;
    const int ref_stride = (((n_blocks * 4) * (packed ? 2 : 1) + 15) / 16) * 16;
    const size_t lds = (size_t)ref_stride * 4;
    if (lds > 64 * 1024)
        return set_err(ctx, FADEHIP_E_UNSUPPORTED, "reference window of %d bases needs %zu B LDS per wave (max 64 KiB)", max_lr, lds);
    const int64_t quad_bytes = (int64_t)quad_stride * 4;
    int64_t max_quads = std::max<int64_t>(1, trace_budget / quad_bytes);
    const int total_quads = (n_items + per_wave - 1) / per_wave;
    const int64_t chunk_quads = std::min<int64_t>(max_quads, total_quads);
    int rc = reserve(ctx, s.trace, (size_t)(chunk_quads * quad_bytes));
    if (rc) return rc;
    rc = reserve(ctx, s.fwd, (size_t)n_items * sizeof(Fwd));
    if (rc) return rc;
    for (int64_t q0 = 0; q0 < total_quads; q0 += chunk_quads) {
        const int quads = (int)std::min<int64_t>(chunk_quads, total_quads - q0);
        const int i0 = (int)(q0 * per_wave);
        const int n = std::min(n_items - i0, quads * per_wave);
        SwArgs a;
        a.work = work + i0;
        a.n_items = n;
        a.q_nib = q_nib;
        a.r_nib = r_nib;
        a.trace = (uint32_t *)s.trace.p;
        a.quad_stride = quad_stride;
        a.ref_stride = ref_stride;
        a.fwd = (Fwd *)s.fwd.p + i0;
        a.sc = ctx->sc;
        a.cand = nullptr;
        a.ckpt = nullptr;
        a.ck_stride = 0;
        a.n_ck = 0;
        int e0 = -1, e1 = -1, e2 = -1;
        if (timed && (rc = record(ctx, s, &e0))) return rc;
        rc = launch_forward_c<0>(ctx, cls, a, quads, lds, st, packed);
        if (rc) return rc;
        if (timed && (rc = record(ctx, s, &e1))) return rc;
        TbArgs t;
        t.work = work + i0;
        t.meta = meta ? meta + i0 : nullptr;
        t.fwd = (Fwd *)s.fwd.p + i0;
        t.n_items = n;
        t.R = R;
        t.q_nib = q_nib;
        t.r_nib = r_nib;
        t.trace = (const uint32_t *)s.trace.p;
        t.quad_stride = quad_stride;
        t.sc = ctx->sc;
        t.out = out + i0;
        t.rs = rs;
        t.stats = (rs && gate) ? s.d_stats() : nullptr;
        t.floor_len = floor_len;
        t.gate = gate;
        t.early_out = 0;
        t.packed = packed ? 1 : 0;
        t.cand = nullptr;
        t.incomplete = nullptr;
        t.incomplete_n = nullptr;
        hipLaunchKernelGGL(traceback_kernel, dim3((n + 63) / 64), dim3(64), 0, st, t);
        HIPCHK(ctx, hipGetLastError());
        if (timed) {
            if ((rc = record(ctx, s, &e2))) return rc;
            s.fwd_spans.push_back({e0, e1});
            s.tb_spans.push_back({e1, e2});
        }
        s.prof_counts[2] += (int64_t)quads * quad_bytes;
        s.n_fwd_launches++;
    }
    return 0;
}

// Queries longer than 512 bases: sw_long_kernel (thread per alignment, full trace) + the common traceback.
int run_long(fadehip_ctx *ctx, Slot &s, hipStream_t st, const Work *work, const Meta *meta, int n_items, int max_lr, int max_lq,
             const uint8_t *q_nib, const uint8_t *r_nib, fadehip_aln *out, uint8_t *rs, int32_t floor_len, int gate, int64_t budget,
             bool timed) {
    const int lhalf = (max_lr + 1) / 2;
    const int64_t per_item = (int64_t)max_lq * lhalf + 8 * (int64_t)max_lr;
    const int64_t chunk = std::min<int64_t>(n_items, budget / std::max<int64_t>(per_item, 1));
    if (chunk < 1)
        return set_err(ctx, FADEHIP_E_UNSUPPORTED, "a %d x %d alignment needs %lld B of trace, more than trace_bytes", max_lq, max_lr,
                       (long long)per_item);
    int rc;
    if ((rc = reserve(ctx, s.fwd, (size_t)n_items * sizeof(Fwd)))) return rc;
    for (int64_t i0 = 0; i0 < n_items; i0 += chunk) {
        const int n = (int)std::min<int64_t>(chunk, n_items - i0);
        if ((rc = reserve(ctx, s.trace, (size_t)max_lq * (size_t)lhalf * (size_t)n)) ||
            (rc = reserve(ctx, s.lrows, 8 * (size_t)max_lr * (size_t)n)))
            return rc;
        LongArgs a;
        a.work = work + i0;
        a.n_items = n;
        a.q_nib = q_nib;
        a.r_nib = r_nib;
        a.hrow = (int32_t *)s.lrows.p;
        a.frow = (int32_t *)s.lrows.p + (size_t)max_lr * (size_t)n;
        a.trace = (uint8_t *)s.trace.p;
        a.lhalf = lhalf;
        a.max_lq = max_lq;
        a.max_lr = max_lr;
        a.fwd = (Fwd *)s.fwd.p + i0;
        a.sc = ctx->sc;
        int e0 = -1, e1 = -1, e2 = -1;
        if (timed && (rc = record(ctx, s, &e0))) return rc;
        hipLaunchKernelGGL(sw_long_kernel, dim3((n + 63) / 64), dim3(64), 0, st, a);
        HIPCHK(ctx, hipGetLastError());
        if (timed && (rc = record(ctx, s, &e1))) return rc;
        TbArgs t;
        memset(&t, 0, sizeof t);
        t.work = work + i0;
        t.meta = meta ? meta + i0 : nullptr;
        t.fwd = (Fwd *)s.fwd.p + i0;
        t.n_items = n;
        t.R = 1;
        t.q_nib = q_nib;
        t.r_nib = r_nib;
        t.trace = nullptr;
        t.quad_stride = 0;
        t.sc = ctx->sc;
        t.out = out + i0;
        t.rs = rs;
        t.stats = (rs && gate) ? s.d_stats() : nullptr;
        t.floor_len = floor_len;
        t.gate = gate;
        t.early_out = 0;
        t.packed = 2;
        t.ltrace = (const uint8_t *)s.trace.p;
        t.lhalf = lhalf;
        t.cand = nullptr;
        t.incomplete = nullptr;
        t.incomplete_n = nullptr;
        hipLaunchKernelGGL(traceback_kernel, dim3((n + 63) / 64), dim3(64), 0, st, t);
        HIPCHK(ctx, hipGetLastError());
        if (timed) {
            if ((rc = record(ctx, s, &e2))) return rc;
            s.fwd_spans.push_back({e0, e1});
            s.tb_spans.push_back({e1, e2});
        }
        s.prof_counts[2] += (int64_t)max_lq * lhalf * n;
        s.n_fwd_launches++;
    }
    return 0;
}

int check_slot(fadehip_ctx *ctx, int slot) {
    if (!ctx) return set_err(nullptr, FADEHIP_E_INVALID, "ctx is NULL");
    if (slot < 0 || slot >= FADEHIP_NUM_SLOTS) return set_err(ctx, FADEHIP_E_INVALID, "slot %d out of range", slot);
    return 0;
}

}  // namespace

extern "C" {

void fadehip_params_default(fadehip_params *p) {
    if (!p) return;
    p->open = 10;
    p->ext = 2;
    p->match = 2;
    p->mismatch = -3;
    p->max_ref_len = 8192;
    p->max_batch_reads = 1 << 20;
    p->trace_bytes = 0;
    p->trace_all = 0;
    p->reserved = 0;
}

int fadehip_abi_version(void) { return FADEHIP_ABI_VERSION; }

const char *fadehip_last_error(const fadehip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int fadehip_create(fadehip_ctx **out, int device, const fadehip_params *params) {
    if (!out) return set_err(nullptr, FADEHIP_E_INVALID, "out is NULL");
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return set_err(nullptr, FADEHIP_E_NODEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= n_dev) return set_err(nullptr, FADEHIP_E_NODEVICE, "device %d out of range (%d devices)", device, n_dev);
    fadehip_ctx *ctx = new (std::nothrow) fadehip_ctx();
    if (!ctx) return set_err(nullptr, FADEHIP_E_NOMEM, "out of host memory");
    ctx->device = device;
    fadehip_params_default(&ctx->prm);
    if (params) {
        ctx->prm = *params;
        if (ctx->prm.max_ref_len <= 0) ctx->prm.max_ref_len = 8192;
        if (ctx->prm.max_batch_reads <= 0) ctx->prm.max_batch_reads = 1 << 20;
    }
    int rc = 0;
    auto fail = [&](int code) {
        g_err = ctx->err;
        fadehip_destroy(ctx);
        return code;
    };
    if (ctx->prm.max_ref_len > (1 << 20)) {
        set_err(ctx, FADEHIP_E_UNSUPPORTED, "max_ref_len %d exceeds 2^20", ctx->prm.max_ref_len);
        return fail(FADEHIP_E_UNSUPPORTED);
    }
    if ((rc = build_score_tab(ctx, ctx->prm, ctx->sc))) return fail(rc);
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
        set_err(ctx, FADEHIP_E_NODEVICE, "cannot open HIP device %d", device);
        return fail(FADEHIP_E_NODEVICE);
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(ctx, FADEHIP_E_NODEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
        return fail(FADEHIP_E_NODEVICE);
    }
    ctx->cu_count = prop.multiProcessorCount;
    if (const char *kv = getenv("FADEHIP_KERNEL")) {
        ctx->use_packed = strcmp(kv, "int32") != 0;
        ctx->two_pass = strcmp(kv, "twopass") == 0;
    }
    // Value ranges of the packed kernels at the longest query (512): the score pass keeps 16-bit keys of 32 * score
    // (64 * score per row pair below the f16 infinity pattern for the row classes <= 14), DP values are 8 * score in
    // int16 compared as f16 patterns.  FADE's scoring (match 2) fits everything; larger match scores take the path
    // whose ranges still hold: the single-pass packed kernel (32-bit keys) up to match 7, the int32 kernel beyond.
    if (ctx->prm.match > 2) ctx->two_pass = false;
    if (8 * (ctx->prm.match * FADEHIP_MAX_QUERY + ctx->prm.open + std::max(ctx->prm.match, 0)) >= 0x7c00) ctx->use_packed = false;
    if (const char *kv = getenv("FADEHIP_SPAN_SLACK")) ctx->span_slack = atoi(kv);
    uint8_t table[256];
    fill_ascii_table(table);
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_ascii_code), table, 256) != hipSuccess) {
        set_err(ctx, FADEHIP_E_HIP, "hipMemcpyToSymbol failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(FADEHIP_E_HIP);
    }
    static_assert(sizeof(uint32_t) * (2 * NUM_LISTS + 2) <= Slot::ZB_C64 - Slot::ZB_COUNTERS, "gate counters overflow their slice");
    static_assert(Slot::ZB_SEL + Slot::ZB_SEL_STRIDE * NUM_CLASSES <= Slot::ZB_STATS, "selection counters overlap the stats");
    static_assert(sizeof(uint32_t) * (NUM_BUCKETS + 1) <= Slot::ZB_SEL_STRIDE, "selection counters overflow their slice");
    for (int k = 0; k < FADEHIP_NUM_SLOTS; k++) {
        Slot &s = ctx->slots[k];
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipHostMalloc((void **)&s.h_gate, Slot::ZB_GATE_BYTES) != hipSuccess ||
            hipHostMalloc((void **)&s.h_sel, sizeof(uint32_t) * (NUM_BUCKETS + 1)) != hipSuccess ||
            hipHostMalloc((void **)&s.h_stats, sizeof(unsigned long long) * 8 * STAT_PARTS) != hipSuccess) {
            set_err(ctx, FADEHIP_E_HIP, "stream / pinned allocation failed: %s", hipGetErrorString(hipGetLastError()));
            return fail(FADEHIP_E_HIP);
        }
        s.h_counters64 = (unsigned long long *)(s.h_gate + Slot::ZB_C64);
        s.h_counters = (uint32_t *)(s.h_gate + Slot::ZB_COUNTERS);
    }
    *out = ctx;
    return 0;
}

void fadehip_destroy(fadehip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (int k = 0; k < FADEHIP_NUM_SLOTS; k++) {
        Slot &s = ctx->slots[k];
        for (DevBuf *b : {&s.tid, &s.pos, &s.lseq, &s.flag, &s.has_sa, &s.cigar_off, &s.cigar_ops, &s.seq_off, &s.seq,
                          &s.rs, &s.fwd, &s.aln, &s.zblock, &s.trace, &s.ckpt, &s.cand, &s.incomplete})
            release(*b);
        release(s.lrows);
        for (int c = 0; c < NUM_LISTS; c++) {
            release(s.work[c]);
            release(s.meta[c]);
        }
        for (hipEvent_t e : s.ev) (void)hipEventDestroy(e);
        if (s.h_gate) (void)hipHostFree(s.h_gate);
        if (s.h_sel) (void)hipHostFree(s.h_sel);
        if (s.h_stats) (void)hipHostFree(s.h_stats);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    release(ctx->genome);
    for (DevBuf *b : {&ctx->l1_q, &ctx->l1_r, &ctx->l1_qn, &ctx->l1_rn, &ctx->l1_bad, &ctx->l1_work, &ctx->l1_aln}) release(*b);
    release(ctx->contig_len);
    release(ctx->contig_base);
    delete ctx;
}

int fadehip_host_alloc(fadehip_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return set_err(ctx, FADEHIP_E_INVALID, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostMalloc(out, bytes ? bytes : 1));
    return 0;
}

int fadehip_host_free(fadehip_ctx *ctx, void *p) {
    if (!p) return 0;
    HIPCHK(ctx, hipHostFree(p));
    return 0;
}

// ------------------------------------------------------------------------------- level 1
int fadehip_sw_batch(fadehip_ctx *ctx, int32_t n, const uint8_t *q, const int64_t *q_off, const uint8_t *r,
                     const int64_t *r_off, fadehip_sw_result *out) {
    if (!ctx) return set_err(nullptr, FADEHIP_E_INVALID, "ctx is NULL");
    if (n < 0 || (n > 0 && (!q || !q_off || !r || !r_off || !out))) return set_err(ctx, FADEHIP_E_INVALID, "NULL argument");
    if (n == 0) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Slot &s = ctx->slots[0];
    const int64_t q_total = q_off[n], r_total = r_off[n];
    // class-partitioned work lists, built on the host from the offsets (no sequence is touched here)
    std::vector<Work> lists[NUM_LISTS];
    int max_lr[NUM_LISTS] = {0};
    int max_long_lq = 0;
    std::vector<int> degenerate;
    for (int k = 0; k < n; k++) {
        const int64_t lq = q_off[k + 1] - q_off[k], lr = r_off[k + 1] - r_off[k];
        if (lq < 0 || lr < 0) return set_err(ctx, FADEHIP_E_INVALID, "offsets must be non-decreasing (pair %d)", k);
        if (lq == 0 || lr == 0) { degenerate.push_back(k); continue; }
        const int cls = list_of_len((int)std::min<int64_t>(lq, 1 << 20), lr);
        if (cls < 0) return set_err(ctx, FADEHIP_E_UNSUPPORTED, "query %d has %lld bases (max %d)", k, (long long)lq, FADEHIP_MAX_LONG_QUERY);
        if (cls == LONG_LIST) max_long_lq = std::max(max_long_lq, (int)lq);
        if (lr > ctx->prm.max_ref_len)
            return set_err(ctx, FADEHIP_E_UNSUPPORTED, "reference %d has %lld bases (max_ref_len %d)", k, (long long)lr, ctx->prm.max_ref_len);
        Work w;
        w.r_base = (uint64_t)r_off[k];
        w.q_base = (uint32_t)q_off[k];
        w.lq = (uint32_t)lq;
        w.lr = (uint32_t)lr;
        w.idx = (uint32_t)k;
        w.flags = 0;
        w.pad = 0;
        lists[cls].push_back(w);
        max_lr[cls] = std::max(max_lr[cls], (int)lr);
    }
    if (q_total >= (int64_t)1 << 32) return set_err(ctx, FADEHIP_E_UNSUPPORTED, "query bases per batch must stay below 2^32");
    DevBuf &d_q = ctx->l1_q, &d_r = ctx->l1_r, &d_qn = ctx->l1_qn, &d_rn = ctx->l1_rn, &d_bad = ctx->l1_bad,
           &d_work = ctx->l1_work, &d_aln = ctx->l1_aln;
    int rc = 0;
    auto cleanup = [&]() {};  // the buffers stay with the ctx (hipMalloc / hipFree per call cost more than small batches)
#define L1CHK(call)                                                                                        \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            cleanup();                                                                                     \
            return set_err(ctx, FADEHIP_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_));            \
        }                                                                                                  \
    } while (0)
    hipStream_t st = s.stream;
    if ((rc = reserve(ctx, d_q, (size_t)q_total + 2)) || (rc = reserve(ctx, d_r, (size_t)r_total + 2)) ||
        (rc = reserve(ctx, d_qn, (size_t)q_total / 2 + 8)) || (rc = reserve(ctx, d_rn, (size_t)r_total / 2 + 8)) ||
        (rc = reserve(ctx, d_bad, 4)) || (rc = reserve(ctx, d_aln, (size_t)n * sizeof(fadehip_aln)))) {
        cleanup();
        return rc;
    }
    L1CHK(hipMemcpyAsync(d_q.p, q, (size_t)q_total, hipMemcpyHostToDevice, st));
    L1CHK(hipMemcpyAsync(d_r.p, r, (size_t)r_total, hipMemcpyHostToDevice, st));
    L1CHK(hipMemsetAsync(d_bad.p, 0, 4, st));
    if (q_total > 0)
        hipLaunchKernelGGL(pack_ascii_kernel, dim3((unsigned)((q_total / 2 + 256) / 256)), dim3(256), 0, st,
                           (const uint8_t *)d_q.p, (uint64_t)q_total, (uint64_t)0, (uint8_t *)d_qn.p, 0, (int *)d_bad.p);
    if (r_total > 0)
        hipLaunchKernelGGL(pack_ascii_kernel, dim3((unsigned)((r_total / 2 + 256) / 256)), dim3(256), 0, st,
                           (const uint8_t *)d_r.p, (uint64_t)r_total, (uint64_t)0, (uint8_t *)d_rn.p, 0, (int *)d_bad.p);
    L1CHK(hipGetLastError());
    size_t n_work = 0;
    for (int c = 0; c < NUM_LISTS; c++) n_work += lists[c].size();
    if ((rc = reserve(ctx, d_work, std::max<size_t>(1, n_work) * sizeof(Work)))) {
        cleanup();
        return rc;
    }
    const int64_t budget = ctx->prm.trace_bytes > 0 ? ctx->prm.trace_bytes : ((int64_t)4 << 30);
    size_t base = 0;
    s.fwd_spans.clear();
    s.tb_spans.clear();
    for (int c = 0; c < NUM_LISTS; c++) {
        if (lists[c].empty()) continue;
        Work *dw = (Work *)d_work.p + base;
        L1CHK(hipMemcpyAsync(dw, lists[c].data(), lists[c].size() * sizeof(Work), hipMemcpyHostToDevice, st));
        if (c == LONG_LIST)
            rc = run_long(ctx, s, st, dw, nullptr, (int)lists[c].size(), max_lr[c], max_long_lq, (const uint8_t *)d_qn.p,
                          (const uint8_t *)d_rn.p, (fadehip_aln *)d_aln.p + base, nullptr, 0, 0, budget, false);
        else
            rc = run_class(ctx, s, st, c, dw, nullptr, (int)lists[c].size(), max_lr[c], (const uint8_t *)d_qn.p,
                           (const uint8_t *)d_rn.p, (fadehip_aln *)d_aln.p + base, nullptr, 0, 0, budget, false);
        if (rc) {
            (void)hipStreamSynchronize(st);
            cleanup();
            return rc;
        }
        base += lists[c].size();
    }
    std::vector<fadehip_aln> h_aln(n_work);
    if (n_work) L1CHK(hipMemcpyAsync(h_aln.data(), d_aln.p, n_work * sizeof(fadehip_aln), hipMemcpyDeviceToHost, st));
    L1CHK(hipStreamSynchronize(st));
    for (size_t k = 0; k < n_work; k++) out[h_aln[k].read_idx] = h_aln[k].sw;
    // empty query or reference: nothing to align (no DP): all of the query is soft-clipped
    for (int k : degenerate) {
        fadehip_sw_result o;
        memset(&o, 0, sizeof o);
        o.end_query = o.end_ref = -1;
        const int64_t lq = q_off[k + 1] - q_off[k];
        if (lq > 0) {
            o.n_ops = 1;
            o.ops[0] = ((uint32_t)lq << 4) | 4u;
        }
        out[k] = o;
    }
    cleanup();
#undef L1CHK
    return 0;
}

// ------------------------------------------------------------------------------- level 2
int fadehip_genome_upload(fadehip_ctx *ctx, int32_t n_contigs, const int64_t *lengths, const uint8_t *const *seqs) {
    if (!ctx) return set_err(nullptr, FADEHIP_E_INVALID, "ctx is NULL");
    if (n_contigs <= 0 || !lengths || !seqs) return set_err(ctx, FADEHIP_E_INVALID, "bad genome arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->h_contig_len.assign(lengths, lengths + n_contigs);
    ctx->h_contig_base.resize(n_contigs);
    uint64_t total = 0;
    for (int c = 0; c < n_contigs; c++) {
        if (lengths[c] < 0) return set_err(ctx, FADEHIP_E_INVALID, "contig %d has negative length", c);
        ctx->h_contig_base[c] = total;
        total += ((uint64_t)lengths[c] + 15) & ~(uint64_t)15;  // contigs start on 8-byte boundaries
    }
    int rc;
    if ((rc = reserve(ctx, ctx->genome, (size_t)(total / 2 + 16)))) return rc;
    if ((rc = reserve(ctx, ctx->contig_len, sizeof(int64_t) * n_contigs))) return rc;
    if ((rc = reserve(ctx, ctx->contig_base, sizeof(uint64_t) * n_contigs))) return rc;
    HIPCHK(ctx, hipMemset(ctx->genome.p, 0, ctx->genome.cap));
    HIPCHK(ctx, hipMemcpy(ctx->contig_len.p, lengths, sizeof(int64_t) * n_contigs, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->contig_base.p, ctx->h_contig_base.data(), sizeof(uint64_t) * n_contigs, hipMemcpyHostToDevice));
    const size_t CH = (size_t)64 << 20;  // staging chunk, even
    DevBuf stage, bad;
    if ((rc = reserve(ctx, stage, CH)) || (rc = reserve(ctx, bad, 4))) {
        release(stage);
        release(bad);
        return rc;
    }
    hipError_t e = hipMemset(bad.p, 0, 4);
    for (int c = 0; c < n_contigs && e == hipSuccess; c++) {
        for (int64_t off = 0; off < lengths[c] && e == hipSuccess; off += (int64_t)CH) {
            const size_t nb = (size_t)std::min<int64_t>((int64_t)CH, lengths[c] - off);
            e = hipMemcpy(stage.p, seqs[c] + off, nb, hipMemcpyHostToDevice);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(pack_ascii_kernel, dim3((unsigned)((nb / 2 + 256) / 256)), dim3(256), 0, 0,
                               (const uint8_t *)stage.p, (uint64_t)nb, ctx->h_contig_base[c] + (uint64_t)off,
                               (uint8_t *)ctx->genome.p, 1, (int *)bad.p);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
    }
    int h_bad = 0;
    if (e == hipSuccess) e = hipMemcpy(&h_bad, bad.p, 4, hipMemcpyDeviceToHost);
    release(stage);
    release(bad);
    if (e != hipSuccess) return set_err(ctx, FADEHIP_E_HIP, "genome upload failed: %s", hipGetErrorString(e));
    if (h_bad) return set_err(ctx, FADEHIP_E_RESIDUE, "FASTA contains '=' which is not a residue this encoding can represent");
    ctx->n_contigs = n_contigs;
    return 0;
}

int fadehip_annotate_upload(fadehip_ctx *ctx, int slot, const fadehip_read_batch *b) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!b || b->n_reads < 0) return set_err(ctx, FADEHIP_E_INVALID, "bad batch");
    if (b->n_reads > ctx->prm.max_batch_reads)
        return set_err(ctx, FADEHIP_E_INVALID, "batch of %d reads exceeds max_batch_reads %d", b->n_reads, ctx->prm.max_batch_reads);
    if (ctx->n_contigs == 0) return set_err(ctx, FADEHIP_E_STATE, "fadehip_genome_upload has not been called");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Slot &s = ctx->slots[slot];
    const int n = b->n_reads;
    s.n_reads = n;
    s.state = 0;
    if (n == 0) { s.state = 1; return 0; }
    if (!b->tid || !b->pos || !b->flag || !b->has_sa || !b->l_seq || !b->cigar_off || !b->seq_off ||
        (b->cigar_off[n] && !b->cigar_ops) || (b->seq_off[n] && !b->seq_packed))
        return set_err(ctx, FADEHIP_E_INVALID, "batch has NULL arrays");
    const size_t n_cig = b->cigar_off[n], n_seq = b->seq_off[n];
    // the kernels index with these: offsets must be non-decreasing, lengths non-negative (one pass, ~1 ms per million)
    for (int i = 0; i < n; i++) {
        if (b->cigar_off[i] > b->cigar_off[i + 1] || b->seq_off[i] > b->seq_off[i + 1] || b->l_seq[i] < 0)
            return set_err(ctx, FADEHIP_E_INVALID, "record %d: cigar_off / seq_off must be non-decreasing and l_seq >= 0", i);
    }
    if ((uint64_t)n_seq * 2 >= ((uint64_t)1 << 32)) return set_err(ctx, FADEHIP_E_UNSUPPORTED, "packed sequence bytes per batch must stay below 2^31");
    // classes present decide which work lists exist
    bool present[NUM_LISTS] = {false};
    for (int i = 0; i < n; i++) {
        const int c = list_of_len(b->l_seq[i] > 0 ? b->l_seq[i] : 1);
        if (c >= 0) present[c] = true;
    }
    // windows beyond WAVE_MAX_WINDOW send any read to the long list; they only exist with a large --window-size
    present[LONG_LIST] = present[LONG_LIST] || ctx->prm.max_ref_len > WAVE_MAX_WINDOW;
    if ((rc = reserve(ctx, s.tid, 4 * (size_t)n)) || (rc = reserve(ctx, s.pos, 4 * (size_t)n)) ||
        (rc = reserve(ctx, s.lseq, 4 * (size_t)n)) || (rc = reserve(ctx, s.flag, 2 * (size_t)n)) ||
        (rc = reserve(ctx, s.has_sa, (size_t)n)) || (rc = reserve(ctx, s.cigar_off, 4 * ((size_t)n + 1))) ||
        (rc = reserve(ctx, s.seq_off, 4 * ((size_t)n + 1))) || (rc = reserve(ctx, s.cigar_ops, 4 * n_cig + 4)) ||
        (rc = reserve(ctx, s.seq, n_seq + 8)) || (rc = reserve(ctx, s.rs, (size_t)n)) ||
        (rc = reserve(ctx, s.aln, sizeof(fadehip_aln) * (size_t)n)) ||
        (rc = reserve(ctx, s.zblock, Slot::ZB_BYTES)))
        return rc;
    for (int c = 0; c < NUM_LISTS; c++) {
        if (!present[c]) continue;
        if ((rc = reserve(ctx, s.work[c], sizeof(Work) * (size_t)n)) || (rc = reserve(ctx, s.meta[c], sizeof(Meta) * (size_t)n)))
            return rc;
    }
    hipStream_t st = s.stream;
    HIPCHK(ctx, hipMemcpyAsync(s.tid.p, b->tid, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(s.pos.p, b->pos, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(s.lseq.p, b->l_seq, 4 * (size_t)n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(s.flag.p, b->flag, 2 * (size_t)n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(s.has_sa.p, b->has_sa, (size_t)n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(s.cigar_off.p, b->cigar_off, 4 * ((size_t)n + 1), hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(s.seq_off.p, b->seq_off, 4 * ((size_t)n + 1), hipMemcpyHostToDevice, st));
    if (n_cig) HIPCHK(ctx, hipMemcpyAsync(s.cigar_ops.p, b->cigar_ops, 4 * n_cig, hipMemcpyHostToDevice, st));
    if (n_seq) HIPCHK(ctx, hipMemcpyAsync(s.seq.p, b->seq_packed, n_seq, hipMemcpyHostToDevice, st));
    s.state = 1;
    return 0;
}

int fadehip_annotate_run(fadehip_ctx *ctx, int slot, int32_t floor_len, int32_t window) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    if (s.state < 1) return set_err(ctx, FADEHIP_E_STATE, "slot %d has no uploaded batch", slot);
    if (window < 0) return set_err(ctx, FADEHIP_E_INVALID, "window must be >= 0");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = s.stream;
    const int n = s.n_reads;
    s.ev_used = 0;
    s.fwd_spans.clear();
    s.tb_spans.clear();
    s.n_aln = 0;
    s.n_fwd_launches = 0;
    s.n_cand = 0;
    s.n_rerun = 0;
    memset(s.prof_counts, 0, sizeof s.prof_counts);
    if (n == 0) {
        memset(s.h_stats, 0, sizeof(unsigned long long) * 8 * STAT_PARTS);
        s.ev_gate0 = s.ev_gate1 = s.ev_end = -1;
        s.state = 2;
        return 0;
    }
    HIPCHK(ctx, hipMemsetAsync(s.zblock.p, 0, Slot::ZB_BYTES, st));  // every counter of the run in one fill
    for (int c = 0; c < NUM_CLASSES; c++) s.sel_fresh[c] = true;
    if ((rc = record(ctx, s, &s.ev_gate0))) return rc;
    GateArgs g;
    g.n_reads = n;
    g.tid = (const int32_t *)s.tid.p;
    g.pos = (const int32_t *)s.pos.p;
    g.l_seq = (const int32_t *)s.lseq.p;
    g.flag = (const uint16_t *)s.flag.p;
    g.has_sa = (const uint8_t *)s.has_sa.p;
    g.cigar_off = (const uint32_t *)s.cigar_off.p;
    g.cigar_ops = (const uint32_t *)s.cigar_ops.p;
    g.seq_off = (const uint32_t *)s.seq_off.p;
    g.floor_len = floor_len;
    g.window = window;
    g.n_contigs = ctx->n_contigs;
    g.contig_len = (const int64_t *)ctx->contig_len.p;
    g.contig_base = (const uint64_t *)ctx->contig_base.p;
    g.max_ref_len = ctx->prm.max_ref_len;
    g.rs = (uint8_t *)s.rs.p;
    for (int c = 0; c < NUM_LISTS; c++) {
        g.work[c] = (Work *)s.work[c].p;
        g.meta[c] = (Meta *)s.meta[c].p;
    }
    g.stats = s.d_stats();
    g.counters = s.d_counters();
    g.counters64 = s.d_counters64();
    hipLaunchKernelGGL(gate_kernel, dim3((n + GATE_BLOCK - 1) / GATE_BLOCK), dim3(GATE_BLOCK), 0, st, g);
    HIPCHK(ctx, hipGetLastError());
    if ((rc = record(ctx, s, &s.ev_gate1))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(s.h_gate, s.zblock.p, Slot::ZB_GATE_BYTES, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    const uint32_t errbits = s.h_counters[2 * NUM_LISTS];
    if (errbits) {
        s.state = 1;
        if (errbits & 2u) return set_err(ctx, FADEHIP_E_UNSUPPORTED, "batch has a soft-clipped read longer than %d bases", FADEHIP_MAX_LONG_QUERY);
        if (errbits & 4u) return set_err(ctx, FADEHIP_E_UNSUPPORTED, "batch has a re-alignment window longer than max_ref_len=%d", ctx->prm.max_ref_len);
        if (errbits & 8u) return set_err(ctx, FADEHIP_E_INVALID, "batch has a mapped soft-clipped record whose seq_packed slice is shorter than its l_seq");
        return set_err(ctx, FADEHIP_E_INVALID, "batch has a mapped soft-clipped read whose tid is not a contig of the uploaded genome");
    }
    const int64_t budget = ctx->prm.trace_bytes > 0 ? ctx->prm.trace_bytes : ((int64_t)16 << 30);
    int base = 0;
    for (int c = 0; c < NUM_CLASSES; c++) {
        const int cnt = (int)s.h_counters[c];
        if (!cnt) continue;
        rc = run_class(ctx, s, st, c, (const Work *)s.work[c].p, (const Meta *)s.meta[c].p, cnt,
                       (int)s.h_counters[NUM_LISTS + c], (const uint8_t *)s.seq.p, (const uint8_t *)ctx->genome.p,
                       (fadehip_aln *)s.aln.p + base, (uint8_t *)s.rs.p, floor_len, 1, budget, true);
        if (rc) return rc;
        base += cnt;
    }
    if (const int cnt = (int)s.h_counters[LONG_LIST]) {
        rc = run_long(ctx, s, st, (const Work *)s.work[LONG_LIST].p, (const Meta *)s.meta[LONG_LIST].p, cnt,
                      (int)s.h_counters[NUM_LISTS + LONG_LIST], (int)s.h_counters[2 * NUM_LISTS + 1], (const uint8_t *)s.seq.p,
                      (const uint8_t *)ctx->genome.p, (fadehip_aln *)s.aln.p + base, (uint8_t *)s.rs.p, floor_len, 1, budget, true);
        if (rc) return rc;
        base += cnt;
    }
    s.n_aln = base;
    if ((rc = record(ctx, s, &s.ev_end))) return rc;
    s.prof_counts[0] = base;
    s.prof_counts[1] = (int64_t)s.h_counters64[0 * C64_STRIDE];
    // algorithmic bytes of the forward kernel: packed query + packed window +
    // 16 B descriptor + 64 B result slot + 4-bit trace cell
    // two-pass: the dominant kernel (pass 1) writes H/E checkpoints instead of the trace
    s.prof_counts[3] = (int64_t)s.h_counters64[1 * C64_STRIDE] + (int64_t)base * 80 +
                       (ctx->two_pass ? (int64_t)s.h_counters64[2 * C64_STRIDE] : (int64_t)(s.h_counters64[0 * C64_STRIDE] / 2));
    s.state = 2;
    return 0;
}

int fadehip_annotate_submit(fadehip_ctx *ctx, int slot, const fadehip_read_batch *batch, int32_t floor_len, int32_t window) {
    int rc = fadehip_annotate_upload(ctx, slot, batch);
    if (rc) return rc;
    return fadehip_annotate_run(ctx, slot, floor_len, window);
}

int fadehip_annotate_collect(fadehip_ctx *ctx, int slot, fadehip_anno_out *out) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    if (s.state != 2) return set_err(ctx, FADEHIP_E_STATE, "slot %d has not been run", slot);
    if (!out) return set_err(ctx, FADEHIP_E_INVALID, "out is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = s.stream;
    const int n = s.n_reads;
    out->n_aln = 0;
    memset(out->stats, 0, sizeof out->stats);
    if (n == 0) return 0;
    if (!out->rs) return set_err(ctx, FADEHIP_E_INVALID, "out->rs is NULL");
    if (s.n_aln > 0 && (!out->aln || out->aln_cap < s.n_aln))
        return set_err(ctx, FADEHIP_E_INVALID, "out->aln holds %d entries, %d needed", out->aln ? out->aln_cap : 0, s.n_aln);
    HIPCHK(ctx, hipMemcpyAsync(out->rs, s.rs.p, (size_t)n, hipMemcpyDeviceToHost, st));
    if (s.n_aln) HIPCHK(ctx, hipMemcpyAsync(out->aln, s.aln.p, sizeof(fadehip_aln) * (size_t)s.n_aln, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemcpyAsync(s.h_stats, s.d_stats(), sizeof(unsigned long long) * 8 * STAT_PARTS, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    out->n_aln = s.n_aln;
    for (int k = 0; k < 8; k++)
        for (int q = 0; q < STAT_PARTS; q++) out->stats[k] += (int64_t)s.h_stats[8 * q + k];
    return 0;
}

int fadehip_sync(fadehip_ctx *ctx) {
    if (!ctx) return set_err(nullptr, FADEHIP_E_INVALID, "ctx is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k < FADEHIP_NUM_SLOTS; k++) HIPCHK(ctx, hipStreamSynchronize(ctx->slots[k].stream));
    return 0;
}

int fadehip_last_run_profile(fadehip_ctx *ctx, int slot, float ms[4], int64_t counts[4]) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    if (s.state != 2) return set_err(ctx, FADEHIP_E_STATE, "slot %d has not been run", slot);
    if (!ms || !counts) return set_err(ctx, FADEHIP_E_INVALID, "NULL argument");
    ms[0] = ms[1] = ms[2] = ms[3] = 0.f;
    for (int k = 0; k < 4; k++) counts[k] = s.prof_counts[k];
    if (s.ev_end < 0) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventSynchronize(s.ev[s.ev_end]));
    float t = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&t, s.ev[s.ev_gate0], s.ev[s.ev_gate1]));
    ms[0] = t;
    for (auto &p : s.fwd_spans) {
        HIPCHK(ctx, hipEventElapsedTime(&t, s.ev[p.first], s.ev[p.second]));
        ms[1] += t;
    }
    for (auto &p : s.tb_spans) {
        HIPCHK(ctx, hipEventElapsedTime(&t, s.ev[p.first], s.ev[p.second]));
        ms[2] += t;
    }
    HIPCHK(ctx, hipEventElapsedTime(&t, s.ev[s.ev_gate0], s.ev[s.ev_end]));
    ms[3] = t;
    return 0;
}

int fadehip_stats_allreduce(fadehip_ctx *const *ctxs, int n_ctx, int64_t *counters, int count) {
    if (!ctxs || n_ctx <= 0 || !counters || count <= 0) return set_err(nullptr, FADEHIP_E_INVALID, "bad arguments");
    fadehip_ctx *c0 = ctxs[0];
    std::vector<int> devs(n_ctx);
    for (int k = 0; k < n_ctx; k++) {
        if (!ctxs[k]) return set_err(c0, FADEHIP_E_INVALID, "ctx %d is NULL", k);
        devs[k] = ctxs[k]->device;
    }
    std::vector<ncclComm_t> comms(n_ctx);
    ncclResult_t nr = ncclCommInitAll(comms.data(), n_ctx, devs.data());
    if (nr != ncclSuccess) return set_err(c0, FADEHIP_E_RCCL, "ncclCommInitAll failed: %s", ncclGetErrorString(nr));
    std::vector<void *> bufs(n_ctx, nullptr);
    int rc = 0;
    for (int k = 0; k < n_ctx && !rc; k++) {
        if (hipSetDevice(devs[k]) != hipSuccess || hipMalloc(&bufs[k], sizeof(int64_t) * count) != hipSuccess ||
            hipMemcpy(bufs[k], counters + (size_t)k * count, sizeof(int64_t) * count, hipMemcpyHostToDevice) != hipSuccess)
            rc = set_err(c0, FADEHIP_E_HIP, "staging counters on device %d failed", devs[k]);
    }
    if (!rc) {
        ncclGroupStart();
        for (int k = 0; k < n_ctx; k++) {
            (void)hipSetDevice(devs[k]);
            nr = ncclAllReduce(bufs[k], bufs[k], count, ncclInt64, ncclSum, comms[k], ctxs[k]->slots[0].stream);
            if (nr != ncclSuccess) rc = set_err(c0, FADEHIP_E_RCCL, "ncclAllReduce failed: %s", ncclGetErrorString(nr));
        }
        nr = ncclGroupEnd();
        if (nr != ncclSuccess && !rc) rc = set_err(c0, FADEHIP_E_RCCL, "ncclGroupEnd failed: %s", ncclGetErrorString(nr));
    }
    for (int k = 0; k < n_ctx; k++) {
        (void)hipSetDevice(devs[k]);
        if (!rc) {
            if (hipStreamSynchronize(ctxs[k]->slots[0].stream) != hipSuccess ||
                hipMemcpy(counters + (size_t)k * count, bufs[k], sizeof(int64_t) * count, hipMemcpyDeviceToHost) != hipSuccess)
                rc = set_err(c0, FADEHIP_E_HIP, "reading reduced counters from device %d failed", devs[k]);
        }
        if (bufs[k]) (void)hipFree(bufs[k]);
        ncclCommDestroy(comms[k]);
    }
    return rc;
}

}  // extern "C"
