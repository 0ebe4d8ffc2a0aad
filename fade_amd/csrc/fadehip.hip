// fadehip.hip — C ABI (include/fadehip.h) over the gfx950 kernels in fadehip_kernels.hpp.
// Host side of the drop-in boundary for source/anno.d:44-50 / source/analysis.d:67.
// No CPU fallback lives here: every entry point either runs the HIP path or returns an error.
//
// Level 2 is an asynchronous pipeline: upload is one hipMemcpyAsync of a pinned batch block, run
// enqueues every kernel and the D2H of the results without reading anything back — launches are sized from host-side
// bounds, the real counts stay on the device, persistent waves draw the traced re-computation's work from a table a
// planning kernel builds — and results / collect waits for the slot.
#include "fadehip_kernels.hpp"
#include "bgzf_deflate.hpp"
#include "bgzf_inflate.hpp"
#include "bam_device.hpp"
#include <rccl/rccl.h>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

using namespace fadehip;

namespace {

thread_local std::string g_err = "";
thread_local const fadehip_ctx *g_err_ctx = nullptr;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};
struct PinBuf {  // staging memory (pin_alloc)
    uint8_t *p = nullptr;
    size_t cap = 0;
};

// Canonical layout of a batch block (fadehip_batch_bind): nine arrays, 256-byte aligned, in this order.
enum { A_TID, A_POS, A_LSEQ, A_CIGOFF, A_SEQOFF, A_FLAG, A_SA, A_CIG, A_SEQ, N_ARR };
struct Layout {
    size_t off[N_ARR], bytes[N_ARR], total;
};
Layout batch_layout(int64_t n, int64_t n_cig, int64_t n_seq) {
    Layout L;
    const size_t b[N_ARR] = {4 * (size_t)n, 4 * (size_t)n, 4 * (size_t)n, 4 * ((size_t)n + 1), 4 * ((size_t)n + 1),
                             2 * (size_t)n, (size_t)n,     4 * (size_t)n_cig, (size_t)n_seq};
    size_t at = 0;
    for (int k = 0; k < N_ARR; k++) {
        L.off[k] = at;
        L.bytes[k] = b[k];
        at += (b[k] + 8 + 255) & ~(size_t)255;  // + 8: the kernels read packed sequences as aligned dwords
    }
    L.total = at;
    return L;
}

// upload keeps the reads whose cigar.alignedLength exceeds this (spliced reads, large deletions) for plan_run
constexpr int64_t WIDE_MIN_SPAN = 1024;

struct Slot {
    hipStream_t stream = nullptr;
    // The score pass fills every wave slot it may use for ~0.8 ms; the small, latency-bound kernels of the other
    // slots (gate, selection, plan, pass 2, traceback) would wait behind it for a slot each.  So the score pass runs
    // on a stream whose CU mask leaves a few CUs (one per XCD by default) to everything else.
    hipStream_t score_stream = nullptr;
    // Two input buffers: while a run works on in[cur], the next batch is uploaded into in[1 - cur] on the copy stream
    // (fadehip_annotate_upload never waits for the run in flight, so H2D leaves the slot's critical path).
    DevBuf in[2];                    // device mirrors of a batch block
    int cur = 0;
    bool have_batch = false;         // a batch has been handed to run at least once (it can be run again)
    hipEvent_t ev_copied = nullptr;  // the H2D of the pending batch, recorded on the ctx's copy stream
    Layout L;                        // layout of the batch in flight
    PinBuf stage[2];                 // staging for batches that do not come as one canonical block (one per input buffer)
    const uint8_t *h_base = nullptr; // host block of the batch in flight (the caller's or `stage`)
    struct Pending {                 // the batch uploaded for the NEXT run
        bool valid = false;
        Layout L;
        const uint8_t *h_base = nullptr;
        uint32_t hist[NUM_LISTS] = {};
        int max_lq = 0;
        int64_t span_bound = 0;
        int n_reads = 0, n_skipped = 0;
        int buf = 0;
        uint32_t out_bound = 0;  // alignments the batch can produce at most (records that carry bases)
        // reads whose cigar.alignedLength alone is long (spliced reads, large deletions): (alignedLength, l_seq), so that
        // run can bound the long list for its window size without looking at the caller's arrays again
        std::vector<std::pair<int64_t, int>> wide;
        } next;
    DevBuf rs, fwd, aln, trace;
    DevBuf ckpt, cand;  // two-pass path
    // All small counters of a run live in one block so that one memset clears them and one copy brings them to the host.
    // Counters that different kernels (or different atomics of one kernel) hammer sit in different 128-byte lines:
    // same-line atomics serialise in one L2 channel (the gate kernel took 60 instead of 49 us with them packed).
    //   [0,128) gate counters | [128,512) counters64, one line each | [512 + 48 c, ...) selection counters of class c |
    //   [1024, 1536) stats: STAT_PARTS partial sums of the 8 stats.d counters | [1536, 2560) tickets of the persistent
    //   launches | [2560, ...) PlanOut
    DevBuf zblock;
    static constexpr size_t ZB_COUNTERS = 0, ZB_C64 = 128, ZB_SEL = 512, ZB_SEL_STRIDE = 48, ZB_STATS = 1024,
                            ZB_TICKETS = 1024 + 8 * 8 * STAT_PARTS, N_TICKETS = 256, ZB_PLAN = ZB_TICKETS + 4 * N_TICKETS,
                            ZB_BYTES = ZB_PLAN + 128;
    unsigned long long *d_counters64() const { return (unsigned long long *)((uint8_t *)zblock.p + ZB_C64); }
    unsigned long long *d_stats() const { return (unsigned long long *)((uint8_t *)zblock.p + ZB_STATS); }
    uint32_t *d_counters() const { return (uint32_t *)((uint8_t *)zblock.p + ZB_COUNTERS); }
    uint32_t *d_sel(int cls) const { return (uint32_t *)((uint8_t *)zblock.p + ZB_SEL + ZB_SEL_STRIDE * (size_t)cls); }
    uint32_t *d_ticket(int k) const { return (uint32_t *)((uint8_t *)zblock.p + ZB_TICKETS) + k; }
    PlanOut *d_plan() const { return (PlanOut *)((uint8_t *)zblock.p + ZB_PLAN); }
    bool sel_fresh[NUM_CLASSES] = {};  // class's selection counters were cleared by the run's memset and not used yet
    int tickets_used = 0;
    DevBuf work[NUM_LISTS], meta[NUM_LISTS];
    DevBuf lrows;  // sw_long_kernel: previous-row H and F-hat
    uint8_t *h_zb = nullptr;  // pinned copy of zblock, filled by the D2H that ends a run
    const uint32_t *h_counters() const { return (const uint32_t *)(h_zb + ZB_COUNTERS); }
    const unsigned long long *h_counters64() const { return (const unsigned long long *)(h_zb + ZB_C64); }
    const unsigned long long *h_stats() const { return (const unsigned long long *)(h_zb + ZB_STATS); }
    const PlanOut *h_plan() const { return (const PlanOut *)(h_zb + ZB_PLAN); }
    PinBuf res;               // pinned result block: rs [n_reads] | aln [sum of the class bounds]
    size_t res_aln_off = 0;
    // host-side bounds of the batch in flight (what the launches are sized from)
    uint32_t bound[NUM_LISTS] = {};     // items per work list, at most
    uint32_t hist[NUM_LISTS] = {};      // upload: records per read-length class (gate-passing ones when the CIGARs were scanned)
    int64_t span_bound = 0;             // upload: max cigar.alignedLength (the caller's bound or the scan's)
    int max_lq = 0;                     // upload: longest read
    uint32_t out_bound = 0, out_cap = 0;  // upload: alignments at most; run: entries of the result array
    std::vector<std::pair<int64_t, int>> wide;
    bool use_ckpt = false;              // this run's score passes leave wave snapshots (see run_class_two_pass)
    bool device_only = false;           // the file path: rs and the alignments stay on the device (only the counter block comes back)
    bool wide_all = false;              // the file path: some read's alignedLength is long and which ones is not known on the host
    int wave_lr_bound = 0, long_max_lq = 0, long_max_lr = 0;
    int floor_len = 0, window = 0;
    int n_reads = 0, n_skipped = 0;
    int state = 0;  // 0 nothing run, 2 run enqueued, 3 results on the host
    int n_aln = 0, n_oversize = 0;
    int64_t stats[8] = {};
    std::vector<hipEvent_t> ev;  // event pool
    int ev_used = 0;
    // (start,end) event index pairs of the last run
    std::vector<std::pair<int, int>> fwd_spans, tb_spans;
    int ev_gate0 = -1, ev_gate1 = -1, ev_end = -1;
    int64_t prof_counts[6] = {0, 0, 0, 0, 0, 0};
    int64_t n_cand = 0, n_rerun = 0;  // two-pass: candidates traced / candidates re-run from further back
    int p2_last_octs[NUM_CLASSES];    // octets pass 2 served for this class in the slot's previous run (-1: none yet)
    int64_t last_cand = 0, last_aln = 0;  // previous run of this slot: candidates traced by pass 2 / alignments
    std::vector<void *> trash;        // scratch buffers outgrown while a run was being enqueued (freed after the slot's sync)
};

// One BGZF compression in flight (fadehip_bgzf_deflate_submit / _wait): its own stream, so that the copies of one lane
// run beside the kernels of the other.
struct BgzfLane {
    hipStream_t stream = nullptr;
    DevBuf src, slots, meta, member_off;  // meta: out_size [n] | out_crc [n] | ticket | total (u64)
    PinBuf out;                   // the members, packed: the pack kernel writes them straight into pinned host memory
    uint8_t *h_out = nullptr;     // where the submission in flight packs to (out.p, or a buffer of the caller: the file path's ring)
    hipEvent_t done = nullptr;    // recorded behind the submission's last kernel
    uint64_t *h_total = nullptr;  // pinned
    size_t n_bytes = 0;
    uint32_t n_blocks = 0;
    int geom = 64;  // block geometry of the submission in flight (bgzf_deflate.hpp)
    int state = 0;  // 0 idle, 1 submitted
};

// The synchronous inflate entry (fadehip_bgzf_inflate): buffers kept between calls.
struct InflateLane {
    hipStream_t stream = nullptr;
    DevBuf comp, blocks, out, status, ticket;
    PinBuf h_status;
};

}  // namespace

struct fadehip_ctx {
    int device = 0;
    fadehip_params prm;
    ScoreTab sc;
    std::string err;
    std::mutex err_mu;
    Slot slots[FADEHIP_NUM_SLOTS];
    // genome
    DevBuf genome, contig_len, contig_base;
    DevBuf l1_q, l1_r, l1_qn, l1_rn, l1_bad, l1_work, l1_aln;  // level 1 (fadehip_sw_batch): kept between calls, grow only
    int n_contigs = 0;
    std::vector<int64_t> h_contig_len;
    std::vector<uint64_t> h_contig_base;
    int cu_count = 0;
    // One copy stream for the uploads of every slot (they share the DMA engine anyway).  The device multiplexes streams
    // onto few hardware queues and streams that share one run in order: so streams are few and made when first used
    // (a slot that is never used has none), 2 N + 1 for N slots in use.
    hipStream_t copy_stream = nullptr;
    // FADEHIP_KERNEL = twopass (default) | pk (single-pass packed int16) | int32 (single-pass int32): A/B runs
    bool use_packed = true;
    bool two_pass = true;
    int tail_cus_per_xcd = 1;  // FADEHIP_TAIL_CUS: CUs per XCD the score pass leaves alone (0: no CU mask, one stream per slot)
    int score_g8 = 1;            // the score pass of reads of up to 152 bases in the 160-row class on eight-lane groups (FADEHIP_SCORE_G8=0: sixteen-lane groups; 2: at two waves per SIMD): the score pass of 150-base reads on eight-lane groups (A/B variant)
    bool blocking_sync = false;  // FADEHIP_BLOCKING_SYNC=1: waits for the device sleep
    int split_cus = 0;  // FADEHIP_BAM_SPLIT=j: the file path's record kernels get j CUs of every XCD, the compressor the others
    bool score_persist = false;  // FADEHIP_SCORE_PERSIST=1: the score pass as a persistent launch (A/B variant)
    int p2_waves_fixed = 0;    // FADEHIP_P2_WAVES: waves of the persistent pass-2 launch (0: adaptive, see run_class_two_pass)
    int span_slack = 24;  // FADEHIP_SPAN_SLACK overrides (tests: -1 makes almost every path leave its range)
    bool debug = false;
    BgzfLane bgzf[FADEHIP_BGZF_LANES];
    InflateLane inf;
    bool bgzf_ready = false;           // the compressor's LDS size has been declared to the runtime
    int bgzf_geom_fixed = 0;           // FADEHIP_BGZF_GEOM: 32 / 64 (0: by the ratio of the previous call)
    double bgzf_last_ratio = 0;        // compressed / raw bytes of the ctx's previous compression
    std::map<uint64_t, int> resident;  // (class, mode, LDS bytes) -> waves of that kernel the device holds at once
    std::mutex resident_mu;
};

namespace {

int set_err(fadehip_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) {
        std::lock_guard<std::mutex> l(ctx->err_mu);
        ctx->err = buf;
    }
    g_err = buf;
    g_err_ctx = ctx;
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

// Staging memory the device reaches over PCIe.  Large blocks are ordinary 2 MB-aligned host memory handed to
// hipHostRegister: measured on MI355X (bench/setup_costs.hip) 1.5 ms per 32 MB against 4-7 ms for hipHostMalloc, with the
// same 56 GB/s up and down and the same 53 GB/s for a kernel that stores into it; small ones come from hipHostMalloc.
// The registry says which way a pointer came.
constexpr size_t PIN_REGISTER_MIN = (size_t)1 << 20;
std::mutex g_pin_mu;
std::unordered_map<void *, bool> g_pin_registered;  // pointer -> the library owns the memory (free() it)

int pin_alloc(fadehip_ctx *ctx, size_t bytes, void **out) {
    *out = nullptr;
    if (bytes >= PIN_REGISTER_MIN && !getenv("FADEHIP_PIN_HOSTMALLOC")) {
        const size_t al = (size_t)1 << 21, n = (bytes + al - 1) & ~(al - 1);
        void *p = aligned_alloc(al, n);
        if (!p) return set_err(ctx, FADEHIP_E_NOMEM, "out of host memory (%zu bytes of staging memory)", n);
        void *dp = nullptr;
        if (hipHostRegister(p, n, hipHostRegisterDefault) == hipSuccess && hipHostGetDevicePointer(&dp, p, 0) == hipSuccess && dp == p) {
            std::lock_guard<std::mutex> l(g_pin_mu);
            g_pin_registered[p] = true;
            *out = p;
            return 0;
        }
        // (a stack where registered memory has another address on the device: the kernels are given host pointers)
        (void)hipGetLastError();
        (void)hipHostUnregister(p);
        (void)hipGetLastError();
        free(p);
    }
    HIPCHK(ctx, hipHostMalloc(out, bytes ? bytes : 1));
    return 0;
}

int pin_free(fadehip_ctx *ctx, void *p) {
    if (!p) return 0;
    bool registered = false, owned = false;
    {
        std::lock_guard<std::mutex> l(g_pin_mu);
        auto it = g_pin_registered.find(p);
        if (it != g_pin_registered.end()) {
            registered = true;
            owned = it->second;
            g_pin_registered.erase(it);
        }
    }
    if (!registered) {
        HIPCHK(ctx, hipHostFree(p));
        return 0;
    }
    const hipError_t e = hipHostUnregister(p);
    if (owned) free(p);
    if (e != hipSuccess) return set_err(ctx, FADEHIP_E_HIP, "hipHostUnregister failed: %s", hipGetErrorString(e));
    return 0;
}

int reserve_pinned(fadehip_ctx *ctx, PinBuf &b, size_t bytes) {
    if (bytes <= b.cap && b.p) return 0;
    int rc;
    if (b.p) {
        if ((rc = pin_free(ctx, b.p))) return rc;
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = std::max<size_t>(bytes + bytes / 8, 4096);  // some headroom: batches of a stream differ a little in size
    want = (want + 4095) & ~(size_t)4095;
    if ((rc = pin_alloc(ctx, want, (void **)&b.p))) return rc;
    b.cap = want;
    return 0;
}

// Growing a scratch buffer while a run is being enqueued: kernels already queued may still use the old allocation, so it
// is only parked here and freed once the slot has been waited for (hipFree would also stall every other stream).
int reserve_run(fadehip_ctx *ctx, std::vector<void *> &trash, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap && b.p) return 0;
    if (b.p) trash.push_back(b.p);
    b.p = nullptr;
    b.cap = 0;
    size_t want = std::max<size_t>(bytes + bytes / 8, 256);  // (headroom: a stream's batches differ a little in size)
    want = (want + 255) & ~(size_t)255;
    HIPCHK(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return 0;
}

// buffers of a stream whose sizes differ a little from call to call: a quarter of headroom, so that they settle
int reserve_roomy(fadehip_ctx *ctx, DevBuf &b, size_t bytes) {
    if (bytes <= b.cap && b.p) return 0;
    return reserve(ctx, b, bytes + bytes / 4 + 4096);
}

void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}
void release(PinBuf &b) {
    if (b.p) (void)pin_free(nullptr, b.p);
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
            else if (cq == 4 && cr == 4) w = (p.rules & FADEHIP_RULE_N_MATCHES_N) ? p.match : p.mismatch;  // Appendix A.1
            else w = (cq == cr) ? p.match : p.mismatch;
            v |= (uint32_t)(w + p.open) << (4 * cr);
        }
        sc.prof[cq] = v;
    }
    sc.open = p.open;
    sc.ext = p.ext;
    sc.match = p.match;
    sc.mismatch = p.mismatch;
    sc.rules = p.rules;
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

int record(fadehip_ctx *ctx, Slot &s, int *idx, hipStream_t on = nullptr) {
    int rc = new_event(ctx, s, idx);
    if (rc) return rc;
    HIPCHK(ctx, hipEventRecord(s.ev[*idx], on ? on : s.stream));
    return 0;
}

template <int C>
int launch_forward_c(fadehip_ctx *ctx, int cls, const SwArgs &a, int quads, size_t lds, hipStream_t st, bool packed, bool longw) {
    if constexpr (C >= NUM_CLASSES) {
        return set_err(ctx, FADEHIP_E_INVALID, "bad class %d", cls);
    } else {
        if (cls == C) {
            constexpr int R = class_rows(C);
            if (packed && longw)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_pk_kernel<R, 0, true>), dim3(quads), dim3(64), lds, st, a);
            else if (packed)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_pk_kernel<R, 0>), dim3(quads), dim3(64), lds, st, a);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_forward_kernel<R>), dim3(quads), dim3(64), lds, st, a);
            HIPCHK(ctx, hipGetLastError());
            return 0;
        }
        return launch_forward_c<C + 1>(ctx, cls, a, quads, lds, st, packed, longw);
    }
}

// mode 1: score pass, 2: traced re-computation (default rules), 3: traced re-computation with rule switches
template <int C>
const void *pk_kernel_ptr(int cls, int mode, bool longw) {
    if constexpr (C >= NUM_CLASSES) {
        return nullptr;
    } else {
        if (cls == C) {
            constexpr int R = class_rows(C);
            if (longw) {  // windows beyond one staged chunk (CH_COLS columns) in this launch
                if (mode == 1) return (const void *)sw_pk_kernel<R, 1, true>;
                if (mode == 2) return (const void *)sw_pk_kernel<R, 2, true>;
                return (const void *)sw_pk_kernel<R, 3, true>;
            }
            if (mode == 1) return (const void *)sw_pk_kernel<R, 1>;
            if (mode == 2) return (const void *)sw_pk_kernel<R, 2>;
            return (const void *)sw_pk_kernel<R, 3>;
        }
        return pk_kernel_ptr<C + 1>(cls, mode, longw);
    }
}

int launch_pk_mode(fadehip_ctx *ctx, int cls, int mode, const SwArgs &a, int waves, size_t lds, hipStream_t st, bool longw) {
    const void *fn = pk_kernel_ptr<0>(cls, mode, longw);
    if (!fn) return set_err(ctx, FADEHIP_E_INVALID, "bad class %d", cls);
    SwArgs copy = a;
    void *args[] = {&copy};
    HIPCHK(ctx, hipLaunchKernel(fn, dim3((unsigned)waves), dim3(64), args, lds, st));
    return 0;
}

// waves of a pass-2 kernel the device holds at once: the size of its persistent launch
int resident_waves(fadehip_ctx *ctx, int cls, int mode, size_t lds, bool longw) {
    const uint64_t key = ((uint64_t)cls << 40) | ((uint64_t)mode << 32) | ((uint64_t)longw << 36) | (uint64_t)lds;
    std::lock_guard<std::mutex> l(ctx->resident_mu);
    auto it = ctx->resident.find(key);
    if (it != ctx->resident.end()) return it->second;
    int per_cu = 0;
    const void *fn = pk_kernel_ptr<0>(cls, mode, longw);
    if (!fn || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, lds) != hipSuccess || per_cu <= 0) per_cu = 4;
    const int w = per_cu * std::max(ctx->cu_count, 1);
    ctx->resident[key] = w;
    return w;
}

struct ClassRun {
    int cls;
    const Work *work;
    const Meta *meta;
    int n_bound;                // items on the list, at most
    const uint32_t *count_dev;  // where the device keeps the real count
    int max_lr;                 // longest window, at most
    const uint8_t *q_nib, *r_nib;
    fadehip_aln *out;
    uint8_t *rs;
    int floor_len, gate;
    int64_t budget;
    bool timed;
};

// Sizes of the two-pass path for one class list: launch geometry and scratch.  Computed for every class of a run before
// its first launch, so that the slot's scratch buffers are sized ONCE per run (to the maximum over the classes) and
// nothing is re-allocated between kernels that are already queued.
struct TwoPassPlan {
    int n_ck = 0;
    uint64_t ck_stride = 0;
    int n_blocks1 = 0, ref_stride1 = 0;
    size_t lds1 = 0;
    int mode2 = 2;
    bool longw = false;  // some window of the launch may be longer than one staged chunk
    int64_t chunk_oct = 1;
    int total_oct = 0;
    uint64_t wave_stride = 0;
    int p2_waves = 1;
    size_t ckpt_bytes = 0, fwd_bytes = 0, trace_bytes = 0, cand_bytes = 0;
};

int plan_two_pass(fadehip_ctx *ctx, const Slot &s, const ClassRun &c, TwoPassPlan &p) {
    const int cls = c.cls, R = class_rows(cls), max_lr = c.max_lr, n_items = c.n_bound;
    // Wave snapshots (every CK_COLS steps, so that pass 2 can resume a sweep instead of repeating it) are only worth
    // their stores — four times the algorithmic bytes of the score pass at C2, twelve times at C3 — when pass 2 has many
    // candidates.  The forced-diagonal shortcut leaves it a few hundred per million reads, so a run leaves snapshots
    // only if the slot's previous run sent more than 1/32 of its alignments to pass 2; without them a candidate is
    // traced from step 0 (same bytes out: test_snapshots_on_and_off_give_the_same_bytes).
    p.n_ck = s.use_ckpt ? (max_lr + 15 + CK_COLS - 1) / CK_COLS : 0;
    p.ck_stride = (uint64_t)p.n_ck * ck_dwords(R) * 64;  // dwords per pass-1 octet
    p.n_blocks1 = (max_lr + 15 + 3) / 4;
    // LDS per 16-lane group: the window's columns (2 bytes each), a chunk of CH_COLS at a time when it is longer than that
    p.ref_stride1 = std::min((((p.n_blocks1 * 4) * 2 + 15) / 16) * 16, (CH_COLS + 16) * 2);
    p.lds1 = (size_t)p.ref_stride1 * 4;
    const bool alt_rules = (ctx->sc.rules & (FADEHIP_RULE_HDIR_DIAG_F_E | FADEHIP_RULE_GAP_TIE_EXTENDS)) !=
                           (FADEHIP_RULE_HDIR_DIAG_F_E | FADEHIP_RULE_GAP_TIE_EXTENDS);
    p.mode2 = alt_rules ? 3 : 2;
    // chunk so that the snapshots of a chunk fit half the budget (the other half is pass-2 trace scratch)
    const int64_t ck_bytes = (int64_t)p.ck_stride * 4;
    p.total_oct = (n_items + 7) / 8;
    p.chunk_oct = std::max<int64_t>(1, std::min<int64_t>(p.total_oct, (c.budget / 2) / std::max<int64_t>(ck_bytes, 1)));
    // Pass 2 is a persistent launch; a wave traces into its own scratch region, sized for a whole window (a path that
    // left its steps is traced again from step 0).  How many waves: what the previous batch of this class needed (its
    // octets are known by the time its results are fetched), twice over, within what the device holds at once; before
    // any batch has run, a moderate guess.  Few waves cost time only when candidates abound; many cost time always,
    // because each must find a wave slot next to the other slots' score passes before it can see that nothing is left.
    p.wave_stride = (uint64_t)p.n_blocks1 * R * 64;  // dwords
    p.longw = p.n_blocks1 > CH_BLOCKS;
    const int resident = resident_waves(ctx, cls, p.mode2, p.lds1, p.longw);
    const int bound_waves = (int)std::min<int64_t>(p.chunk_oct + NUM_BUCKETS, resident);
    int p2_waves = s.p2_last_octs[cls] >= 0 ? std::min(2 * s.p2_last_octs[cls] + 64, bound_waves) : std::min(1024, bound_waves);
    if (ctx->p2_waves_fixed > 0) p2_waves = std::min(ctx->p2_waves_fixed, bound_waves);
    // ... and within the scratch the budget allows
    p.p2_waves = (int)std::max<int64_t>(1, std::min<int64_t>(p2_waves, (c.budget / 2) / std::max<int64_t>((int64_t)p.wave_stride * 4, 1)));
    p.ckpt_bytes = (size_t)(p.chunk_oct * ck_bytes);
    p.fwd_bytes = (size_t)n_items * sizeof(Fwd);
    p.trace_bytes = (size_t)p.p2_waves * (size_t)p.wave_stride * 4;
    p.cand_bytes = sizeof(Cand) * (size_t)NUM_BUCKETS * (size_t)std::min<int64_t>(n_items, p.chunk_oct * 8);
    return 0;
}

// Two-pass path for one class list, enqueued without a read-back: pass 1 scores every alignment (and,
// when the run asks for them, leaves wave snapshots every CK_COLS steps); the selection — inside the score pass's waves —
// finishes what needs no DP (non-candidates, forced diagonals) and buckets the remaining candidates by the steps to
// re-compute; pass 2 — ONE persistent launch — turns the bucket counts into its table, re-computes those steps with
// trace, walks the tracebacks and traces again, from further back, the few paths that left their steps.
int run_class_two_pass(fadehip_ctx *ctx, Slot &s, hipStream_t st, const ClassRun &c) {
    const int cls = c.cls, n_items = c.n_bound;
    TwoPassPlan p;
    int rc;
    if ((rc = plan_two_pass(ctx, s, c, p))) return rc;
    const int n_ck = p.n_ck, ref_stride1 = p.ref_stride1, mode2 = p.mode2, total_oct = p.total_oct, p2_waves = p.p2_waves;
    const uint64_t ck_stride = p.ck_stride, wave_stride = p.wave_stride;
    const size_t lds1 = p.lds1;
    const int64_t chunk_oct = p.chunk_oct;
    // (level 2 sized these for all classes before the run's first launch; a buffer that still has to grow here is parked,
    // not freed: earlier launches of this run may be using it)
    if ((rc = reserve_run(ctx, s.trash, s.ckpt, p.ckpt_bytes)) || (rc = reserve_run(ctx, s.trash, s.fwd, p.fwd_bytes)) ||
        (rc = reserve_run(ctx, s.trash, s.trace, p.trace_bytes)) || (rc = reserve_run(ctx, s.trash, s.cand, p.cand_bytes)))
        return rc;
    s.prof_counts[2] = std::max<int64_t>(s.prof_counts[2], (int64_t)p.trace_bytes);
    uint32_t *const sel_counters = s.d_sel(cls);
    for (int64_t o0 = 0; o0 < total_oct; o0 += chunk_oct) {
        const int octs = (int)std::min<int64_t>(chunk_oct, total_oct - o0);
        const int i0 = (int)(o0 * 8);
        const int n = std::min(n_items - i0, octs * 8);
        if (!s.sel_fresh[cls]) HIPCHK(ctx, hipMemsetAsync(sel_counters, 0, sizeof(uint32_t) * NUM_BUCKETS, st));
        s.sel_fresh[cls] = false;
        SwArgs a;
        memset(&a, 0, sizeof a);
        a.work = c.work + i0;
        a.n_items = n;
        a.q_nib = c.q_nib;
        a.r_nib = c.r_nib;
        a.trace = nullptr;
        a.quad_stride = 0;
        a.ref_stride = ref_stride1;
        a.fwd = (Fwd *)s.fwd.p + i0;
        a.sc = ctx->sc;
        a.cand = nullptr;
        a.ckpt = (uint32_t *)s.ckpt.p;
        a.ck_stride = ck_stride;
        a.n_ck = n_ck;
        a.count_dev = c.count_dev;
        a.item_base = (uint32_t)i0;
        // the selection rides in the score pass's waves
        a.sel.enabled = 1;
        a.sel.no_ckpt = n_ck == 0 ? 1 : 0;
        a.sel.meta = c.meta ? c.meta + i0 : nullptr;
        a.sel.floor_len = c.floor_len;
        a.sel.trace_all = ctx->prm.trace_all;
        a.sel.span_slack = ctx->span_slack;
        a.sel.cand = (Cand *)s.cand.p;
        a.sel.cap = (uint32_t)n;
        a.sel.bucket_n = sel_counters;
        a.sel.out = c.out;
        a.sel.rs = c.rs;
        a.sel.stats = (c.rs && c.gate) ? s.d_stats() : nullptr;
        a.sel.gate = c.gate;
        a.sel.match = getenv("FADEHIP_NO_SHORTCUT") ? 0 : ctx->sc.match;  // read per run: a test flips it on a live ctx
        a.sel.mismatch = ctx->sc.mismatch;
        int e0 = -1, e1 = -1, e2 = -1;
        // the score pass goes to the slot's CU-masked stream (fork / join by events); its timing events are recorded there
        hipStream_t sst = s.score_stream ? s.score_stream : st;
        if (sst != st) {
            int ef = -1;
            if ((rc = record(ctx, s, &ef, st))) return rc;
            HIPCHK(ctx, hipStreamWaitEvent(sst, s.ev[ef], 0));
        }
        if (c.timed && (rc = record(ctx, s, &e0, sst))) return rc;
        int waves1 = octs;
        if (ctx->score_persist && class_rows(cls) == 10 && !p.longw && s.tickets_used < (int)Slot::N_TICKETS) {
            // the A/B variant: as many waves as the CUs of the stream hold at once, each drawing octets by ticket
            const int res = resident_waves(ctx, cls, 1, lds1, p.longw);
            const int cus = std::max(ctx->cu_count, 1), mine = s.score_stream ? std::max(cus - 8 * ctx->tail_cus_per_xcd, 1) : cus;
            waves1 = std::max(1, std::min(octs, (int)((int64_t)res * mine / cus)));
            a.ticket = s.d_ticket(s.tickets_used++);
        }
        // the other A/B variant: eight-lane groups, 19 rows per lane (152 rows for reads of up to 152 bases in the 160-row class),
        // sixteen alignments per wavefront; no snapshots in this geometry (pass 2 re-computes from step 0)
        // Eight-lane groups: a lane owns R8 rows of 8 R8 — rows in steps of 8 instead of 16 (36-base reads: 40 rows instead of
        // 64; 50: 56 / 64; 76: 80 / 96; 100, 101: 104 / 128; 150, 151: 152 / 160), sixteen alignments per wavefront, a skew of 7
        // steps.  Taken when every read of the batch fits the rows (the batch's longest read is known: s.max_lq) and they are
        // fewer than the sixteen-lane class's; no snapshots in this geometry (pass 2 re-computes from step 0).
        const void *g8_fn = nullptr;
        if (ctx->score_g8 && !p.longw && n_ck == 0 && !a.ticket) {
            const int rows16 = 16 * class_rows(cls), lq = std::min(s.max_lq, rows16);
            if (lq <= 40 && rows16 > 40) g8_fn = (const void *)sw_pk_kernel<5, 1, false, 8>;
            else if (lq <= 56 && rows16 > 56) g8_fn = (const void *)sw_pk_kernel<7, 1, false, 8>;
            else if (lq <= 80 && rows16 > 80) g8_fn = (const void *)sw_pk_kernel<10, 1, false, 8>;
            else if (lq <= 104 && rows16 > 104) g8_fn = (const void *)sw_pk_kernel<13, 1, false, 8>;
            else if (lq <= 152 && rows16 > 152 && rows16 <= 160)
                g8_fn = ctx->score_g8 == 2 ? (const void *)sw_pk_kernel<19, 1, false, 8, false, 2> : (const void *)sw_pk_kernel<19, 1, false, 8>;
            // (a class below the batch's top class holds reads of ITS row range only, which the kernel chosen for min(max_lq, rows) covers)
            if (g8_fn && s.max_lq > rows16) g8_fn = nullptr;  // (not the top class: its reads may be any length up to rows16)
        }
        if (g8_fn) {
            if (ctx->debug && o0 == 0) fprintf(stderr, "[fadehip] class of %d rows: score pass on eight-lane groups (longest read of the batch: %d)\n", 16 * class_rows(cls), s.max_lq);
            SwArgs copy = a;
            void *args[] = {&copy};
            HIPCHK(ctx, hipLaunchKernel(g8_fn, dim3((unsigned)((n + 15) / 16)), dim3(64), args, 2 * lds1, sst));
        } else if (a.ticket) {
            SwArgs copy = a;
            void *args[] = {&copy};
            HIPCHK(ctx, hipLaunchKernel((const void *)sw_pk_kernel<10, 1, false, 16, true>, dim3((unsigned)waves1), dim3(64), args, lds1, sst));
        } else if ((rc = launch_pk_mode(ctx, cls, 1, a, waves1, lds1, sst, p.longw))) return rc;
        a.ticket = nullptr;
        if ((c.timed || sst != st) && (rc = record(ctx, s, &e1, sst))) return rc;
        if (sst != st) HIPCHK(ctx, hipStreamWaitEvent(st, s.ev[e1], 0));
        // pass 2 + tracebacks + re-traced paths: one persistent launch
        if (s.tickets_used >= (int)Slot::N_TICKETS)
            return set_err(ctx, FADEHIP_E_UNSUPPORTED, "more than %d pass-2 launches in one run (raise trace_bytes)", (int)Slot::N_TICKETS);
        SwArgs b2 = a;
        b2.sel.enabled = 0;
        b2.n_items = 0;
        b2.cand = (const Cand *)s.cand.p;
        b2.cand_cap = (uint32_t)n;
        b2.bucket_n = sel_counters;
        b2.cand_total = &s.d_plan()->cand_total;
        b2.trace = (uint32_t *)s.trace.p;
        b2.quad_stride = wave_stride;
        b2.count_dev = nullptr;
        b2.ticket = s.d_ticket(s.tickets_used++);
        b2.meta = c.meta ? c.meta + i0 : nullptr;
        b2.out = c.out;
        b2.rs = c.rs;
        b2.stats = (c.rs && c.gate) ? s.d_stats() : nullptr;
        b2.floor_len = c.floor_len;
        b2.gate = c.gate;
        b2.early_out = (c.gate && c.meta && !ctx->prm.trace_all) ? 1 : 0;
        b2.rerun_total = &s.d_plan()->rerun_total;
        if ((rc = launch_pk_mode(ctx, cls, mode2, b2, std::min(p2_waves, octs + NUM_BUCKETS), lds1, st, p.longw))) return rc;
        if (c.timed) {
            if ((rc = record(ctx, s, &e2))) return rc;
            s.fwd_spans.push_back({e0, e1});
            s.tb_spans.push_back({e1, e2});
        }
    }
    return 0;
}

// Single-pass kernels (FADEHIP_KERNEL=pk|int32, and scoring schemes beyond the two-pass ranges): forward with full
// trace + traceback for one class list whose item count the host knows, chunked so the trace fits the budget.
int run_class_single(fadehip_ctx *ctx, Slot &s, hipStream_t st, const ClassRun &c) {
    const int cls = c.cls, R = class_rows(cls), n_items = c.n_bound, max_lr = c.max_lr;
    const bool packed = ctx->use_packed;
    const int per_wave = packed ? 8 : 4;  // alignments per wavefront
    const int n_blocks = (max_lr + 15 + 3) / 4;
    // dwords of trace per wave: 4 bits per cell slot either way
    const uint64_t quad_stride = (uint64_t)n_blocks * (packed ? R : R / 2) * 64;
    const int ref_stride = packed ? std::min((((n_blocks * 4) * 2 + 15) / 16) * 16, (CH_COLS + 16) * 2) : (((n_blocks * 4) + 15) / 16) * 16;
    const size_t lds = (size_t)ref_stride * 4;
    if (lds > 64 * 1024)
        return set_err(ctx, FADEHIP_E_UNSUPPORTED, "reference window of %d bases needs %zu B LDS per wave (max 64 KiB)", max_lr, lds);
    const int64_t quad_bytes = (int64_t)quad_stride * 4;
    int64_t max_quads = std::max<int64_t>(1, c.budget / quad_bytes);
    const int total_quads = (n_items + per_wave - 1) / per_wave;
    const int64_t chunk_quads = std::min<int64_t>(max_quads, total_quads);
    int rc = reserve_run(ctx, s.trash, s.trace, (size_t)(chunk_quads * quad_bytes));
    if (rc) return rc;
    rc = reserve_run(ctx, s.trash, s.fwd, (size_t)n_items * sizeof(Fwd));
    if (rc) return rc;
    for (int64_t q0 = 0; q0 < total_quads; q0 += chunk_quads) {
        const int quads = (int)std::min<int64_t>(chunk_quads, total_quads - q0);
        const int i0 = (int)(q0 * per_wave);
        const int n = std::min(n_items - i0, quads * per_wave);
        SwArgs a;
        memset(&a, 0, sizeof a);
        a.work = c.work + i0;
        a.n_items = n;
        a.q_nib = c.q_nib;
        a.r_nib = c.r_nib;
        a.trace = (uint32_t *)s.trace.p;
        a.quad_stride = quad_stride;
        a.ref_stride = ref_stride;
        a.fwd = (Fwd *)s.fwd.p + i0;
        a.sc = ctx->sc;
        int e0 = -1, e1 = -1, e2 = -1;
        if (c.timed && (rc = record(ctx, s, &e0))) return rc;
        rc = launch_forward_c<0>(ctx, cls, a, quads, lds, st, packed, n_blocks > CH_BLOCKS);
        if (rc) return rc;
        if (c.timed && (rc = record(ctx, s, &e1))) return rc;
        TbArgs t;
        memset(&t, 0, sizeof t);
        t.work = c.work + i0;
        t.meta = c.meta ? c.meta + i0 : nullptr;
        t.fwd = (Fwd *)s.fwd.p + i0;
        t.n_items = n;
        t.R = R;
        t.q_nib = c.q_nib;
        t.r_nib = c.r_nib;
        t.trace = (const uint32_t *)s.trace.p;
        t.quad_stride = quad_stride;
        t.sc = ctx->sc;
        t.out = c.out;
        t.rs = c.rs;
        t.stats = (c.rs && c.gate) ? s.d_stats() : nullptr;
        t.floor_len = c.floor_len;
        t.gate = c.gate;
        t.early_out = 0;
        t.packed = packed ? 1 : 0;
        hipLaunchKernelGGL(traceback_kernel, dim3((n + 63) / 64), dim3(64), 0, st, t);
        HIPCHK(ctx, hipGetLastError());
        if (c.timed) {
            if ((rc = record(ctx, s, &e2))) return rc;
            s.fwd_spans.push_back({e0, e1});
            s.tb_spans.push_back({e1, e2});
        }
        s.prof_counts[2] += (int64_t)quads * quad_bytes;
    }
    return 0;
}

// Queries longer than 512 bases (or windows beyond the wave kernels' LDS): sw_long_kernel (thread per alignment, full
// trace) + the common traceback.  n_bound / max_lq / max_lr are upper bounds, the live count stays on the device.
// Long list on the wave kernel with one alignment per wavefront (sw_forward64_kernel): reads of up to 4,096 bases, windows
// of up to LONG_WAVE_MAX_WINDOW columns, under the default end-cell and gap-tie rules (the kernel's flags are those rules').
constexpr int LONG_WAVE_MAX_QUERY = 4096, LONG_WAVE_MAX_WINDOW = 65000;
int run_long_wave(fadehip_ctx *ctx, Slot &s, hipStream_t st, const ClassRun &c, int max_lq) {
    const int n_items = c.n_bound, max_lr = c.max_lr;
    // rows per lane: the smallest of 12 / 16 / 24 / 32 / 48 / 64 that holds the list's longest read (64: 256 VGPRs + 220 AGPRs, one wave per SIMD)
    const int R = max_lq <= 768 ? 12 : max_lq <= 1024 ? 16 : max_lq <= 1536 ? 24 : max_lq <= 2048 ? 32 : max_lq <= 3072 ? 48 : 64;
    const int n_blocks = (max_lr + 63 + 3) / 4;
    const uint64_t quad_stride = (uint64_t)n_blocks * (R / 2) * 64;  // dwords of trace per alignment
    const size_t lds = (size_t)(((n_blocks * 4) + 15) / 16) * 16;
    const int64_t item_bytes = (int64_t)quad_stride * 4;
    const int64_t chunk = std::min<int64_t>(n_items, std::max<int64_t>(1, c.budget / item_bytes));
    int rc;
    if ((rc = reserve_run(ctx, s.trash, s.fwd, (size_t)n_items * sizeof(Fwd))) || (rc = reserve_run(ctx, s.trash, s.trace, (size_t)(chunk * item_bytes)))) return rc;
    for (int64_t i0 = 0; i0 < n_items; i0 += chunk) {
        const int n = (int)std::min<int64_t>(chunk, n_items - i0);
        SwArgs a;
        memset(&a, 0, sizeof a);
        a.work = c.work + i0;
        a.n_items = n;
        a.q_nib = c.q_nib;
        a.r_nib = c.r_nib;
        a.trace = (uint32_t *)s.trace.p;
        a.quad_stride = quad_stride;
        a.ref_stride = (int32_t)lds;
        a.fwd = (Fwd *)s.fwd.p + i0;
        a.sc = ctx->sc;
        a.count_dev = c.count_dev;
        a.item_base = (uint32_t)i0;
        int e0 = -1, e1 = -1, e2 = -1;
        if (c.timed && (rc = record(ctx, s, &e0))) return rc;
        if (R == 12) hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_forward64_kernel<12>), dim3(n), dim3(64), lds, st, a);
        else if (R == 16) hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_forward64_kernel<16>), dim3(n), dim3(64), lds, st, a);
        else if (R == 24) hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_forward64_kernel<24>), dim3(n), dim3(64), lds, st, a);
        else if (R == 32) hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_forward64_kernel<32>), dim3(n), dim3(64), lds, st, a);
        else if (R == 48) hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_forward64_kernel<48>), dim3(n), dim3(64), lds, st, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(sw_forward64_kernel<64>), dim3(n), dim3(64), lds, st, a);
        HIPCHK(ctx, hipGetLastError());
        if (c.timed && (rc = record(ctx, s, &e1))) return rc;
        TbArgs t;
        memset(&t, 0, sizeof t);
        t.work = c.work + i0;
        t.meta = c.meta ? c.meta + i0 : nullptr;
        t.fwd = (Fwd *)s.fwd.p + i0;
        t.n_items = n;
        t.R = R;
        t.q_nib = c.q_nib;
        t.r_nib = c.r_nib;
        t.trace = (const uint32_t *)s.trace.p;
        t.quad_stride = quad_stride;
        t.sc = ctx->sc;
        t.out = c.out;
        t.rs = c.rs;
        t.stats = (c.rs && c.gate) ? s.d_stats() : nullptr;
        t.floor_len = c.floor_len;
        t.gate = c.gate;
        t.early_out = 0;
        t.packed = 3;
        t.count_dev = c.count_dev;
        t.item_base = (uint32_t)i0;
        hipLaunchKernelGGL(traceback_kernel, dim3((n + 63) / 64), dim3(64), 0, st, t);
        HIPCHK(ctx, hipGetLastError());
        if (c.timed) {
            if ((rc = record(ctx, s, &e2))) return rc;
            s.fwd_spans.push_back({e0, e1});
            s.tb_spans.push_back({e1, e2});
        }
        s.prof_counts[2] += (int64_t)n * item_bytes;
    }
    return 0;
}

int run_long(fadehip_ctx *ctx, Slot &s, hipStream_t st, const ClassRun &c, int max_lq) {
    const int n_items = c.n_bound, max_lr = c.max_lr;
    // what the one-alignment-per-wave kernel holds goes there; the thread-per-alignment kernel below keeps the rest (longer
    // reads, wider windows, the non-default end-cell / gap-tie rules: FADEHIP_LONG_THREAD=1 sends everything to it, for A/B runs)
    const uint32_t need_rules = FADEHIP_RULE_END_MIN_REF_THEN_QUERY | FADEHIP_RULE_GAP_TIE_EXTENDS;
    if (max_lq <= LONG_WAVE_MAX_QUERY && max_lr <= LONG_WAVE_MAX_WINDOW && (ctx->sc.rules & need_rules) == need_rules && !getenv("FADEHIP_LONG_THREAD"))
        return run_long_wave(ctx, s, st, c, max_lq);
    const int lhalf = (max_lr + 1) / 2;
    const int64_t per_item = (int64_t)max_lq * lhalf + 8 * (int64_t)max_lr;
    const int64_t chunk = std::min<int64_t>(n_items, c.budget / std::max<int64_t>(per_item, 1));
    if (chunk < 1)
        return set_err(ctx, FADEHIP_E_UNSUPPORTED, "a %d x %d alignment needs %lld B of trace, more than trace_bytes", max_lq, max_lr,
                       (long long)per_item);
    int rc;
    if ((rc = reserve_run(ctx, s.trash, s.fwd, (size_t)n_items * sizeof(Fwd)))) return rc;
    for (int64_t i0 = 0; i0 < n_items; i0 += chunk) {
        const int n = (int)std::min<int64_t>(chunk, n_items - i0);
        if ((rc = reserve_run(ctx, s.trash, s.trace, (size_t)max_lq * (size_t)lhalf * (size_t)n)) ||
            (rc = reserve_run(ctx, s.trash, s.lrows, 8 * (size_t)max_lr * (size_t)n)))
            return rc;
        LongArgs a;
        memset(&a, 0, sizeof a);
        a.work = c.work + i0;
        a.n_items = n;
        a.q_nib = c.q_nib;
        a.r_nib = c.r_nib;
        a.hrow = (int32_t *)s.lrows.p;
        a.frow = (int32_t *)s.lrows.p + (size_t)max_lr * (size_t)n;
        a.trace = (uint8_t *)s.trace.p;
        a.lhalf = lhalf;
        a.max_lq = max_lq;
        a.max_lr = max_lr;
        a.fwd = (Fwd *)s.fwd.p + i0;
        a.sc = ctx->sc;
        a.count_dev = c.count_dev;
        a.item_base = (uint32_t)i0;
        int e0 = -1, e1 = -1, e2 = -1;
        if (c.timed && (rc = record(ctx, s, &e0))) return rc;
        hipLaunchKernelGGL(sw_long_kernel, dim3((n + 63) / 64), dim3(64), 0, st, a);
        HIPCHK(ctx, hipGetLastError());
        if (c.timed && (rc = record(ctx, s, &e1))) return rc;
        TbArgs t;
        memset(&t, 0, sizeof t);
        t.work = c.work + i0;
        t.meta = c.meta ? c.meta + i0 : nullptr;
        t.fwd = (Fwd *)s.fwd.p + i0;
        t.n_items = n;
        t.R = 1;
        t.q_nib = c.q_nib;
        t.r_nib = c.r_nib;
        t.sc = ctx->sc;
        t.out = c.out;
        t.rs = c.rs;
        t.stats = (c.rs && c.gate) ? s.d_stats() : nullptr;
        t.floor_len = c.floor_len;
        t.gate = c.gate;
        t.early_out = 0;
        t.packed = 2;
        t.ltrace = (const uint8_t *)s.trace.p;
        t.lhalf = lhalf;
        t.count_dev = c.count_dev;
        t.item_base = (uint32_t)i0;
        hipLaunchKernelGGL(traceback_kernel, dim3((n + 63) / 64), dim3(64), 0, st, t);
        HIPCHK(ctx, hipGetLastError());
        if (c.timed) {
            if ((rc = record(ctx, s, &e2))) return rc;
            s.fwd_spans.push_back({e0, e1});
            s.tb_spans.push_back({e1, e2});
        }
        s.prof_counts[2] += (int64_t)max_lq * lhalf * n;
    }
    return 0;
}

// A stream on CUs [lo, hi) of every XCD (mask bit i is CU i / 8 of XCD i % 8 on MI355X); nullptr where the stack has no CU masks.
hipStream_t xcd_slice_stream(fadehip_ctx *ctx, int lo, int hi) {
    if (ctx->cu_count < 64 || ctx->cu_count % 32) return nullptr;
    std::vector<uint32_t> mask((size_t)ctx->cu_count / 32, 0u);
    for (int i = 0; i < ctx->cu_count; i++)
        if (i / 8 >= lo && i / 8 < hi) mask[(size_t)i / 32] |= 1u << (i % 32);
    hipStream_t q = nullptr;
    if (hipExtStreamCreateWithCUMask(&q, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return q;
}

// streams, events and the pinned counter block of a slot, made when the slot is first used
int ensure_slot(fadehip_ctx *ctx, Slot &s) {
    if (s.h_zb) return 0;
    if (!s.stream) HIPCHK(ctx, hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));  // (the file path lends slot 0 the ctx's copy stream)
    HIPCHK(ctx, hipEventCreateWithFlags(&s.ev_copied, hipEventDisableTiming));
    HIPCHK(ctx, hipHostMalloc((void **)&s.h_zb, Slot::ZB_BYTES));
    memset(s.h_zb, 0, Slot::ZB_BYTES);
    if (ctx->tail_cus_per_xcd > 0 && ctx->cu_count >= 64 && ctx->cu_count % 32 == 0) {
        // Bits 33 k + 8 j, k = 0..7, j < tail CUs per XCD: one CU in each of the 8 XCDs whether the mask counts CUs
        // XCD-interleaved (bit i -> XCD i % 8; what MI355X does: a mask that clears bit 0 of every word instead takes
        // eight CUs from one XCD and costs the score pass 25 %) or XCD by XCD; everything else is enabled.
        std::vector<uint32_t> mask((size_t)ctx->cu_count / 32, 0xffffffffu);
        for (int x = 0; x < 8 && 33 * x < ctx->cu_count; x++)
            for (int j = 0; j < ctx->tail_cus_per_xcd && j < 3; j++) {
                const int bit = 33 * x + 8 * j;
                if (bit < ctx->cu_count) mask[(size_t)bit / 32] &= ~(1u << (bit % 32));
            }
        if (const char *kv = getenv("FADEHIP_CU_MASK")) {  // experiments: comma-separated hex words, lowest CUs first
            size_t w = 0;
            for (const char *q = kv; *q && w < mask.size(); w++) {
                mask[w] = (uint32_t)strtoul(q, nullptr, 16);
                q = strchr(q, ',');
                if (!q) break;
                q++;
            }
        }
        if (hipExtStreamCreateWithCUMask(&s.score_stream, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
            (void)hipGetLastError();
            s.score_stream = nullptr;  // no CU masks on this stack: the score pass stays on the slot's stream
        }
    }
    return 0;
}

int check_slot(fadehip_ctx *ctx, int slot) {
    if (!ctx) return set_err(nullptr, FADEHIP_E_INVALID, "ctx is NULL");
    if (slot < 0 || slot >= FADEHIP_NUM_SLOTS) return set_err(ctx, FADEHIP_E_INVALID, "slot %d out of range", slot);
    return 0;
}

// typed views of the batch in flight, on the host and on the device
template <class T>
const T *d_arr(const Slot &s, int k) { return (const T *)((const uint8_t *)s.in[s.cur].p + s.L.off[k]); }

// anno.d:61 + util.d:37-62 + dhtslib alignedLength over one record's CIGAR, as gate_kernel computes them
struct CigarSummary {
    int n_soft;
    uint32_t clipL, clipR;
    int64_t aligned;
};
inline CigarSummary summarize_cigar(const uint32_t *ops, uint32_t c0, uint32_t c1) {
    CigarSummary r = {0, 0, 0, 0};
    bool first = true;
    for (uint32_t k = c0; k < c1; k++) {
        const uint32_t op = ops[k] & 15u, len = ops[k] >> 4;
        if (op == 4) r.n_soft++;
        if (FADEHIP_OP_CONSUMES_REF(op)) r.aligned += len;
        if (op == 5) continue;
        const bool is_sc = (op == 4);
        if (first && !is_sc) first = false;
        else if (first && is_sc) r.clipL = len;
        else if (is_sc) r.clipR = len;
    }
    return r;
}

// Host-side bounds of a run (the launches are sized from these; the device keeps the real counts and checks every bound
// before it writes: a record beyond one is reported at results, never trusted).
int plan_run(fadehip_ctx *ctx, Slot &s) {
    const int64_t w = s.window;
    for (int c = 0; c < NUM_LISTS; c++) s.bound[c] = s.hist[c];
    const int64_t lr_bound = std::min<int64_t>(s.span_bound + 2 * w, ctx->prm.max_ref_len);
    s.wave_lr_bound = (int)std::min<int64_t>(lr_bound, WAVE_MAX_WINDOW);
    s.long_max_lq = s.max_lq;
    s.long_max_lr = (int)std::max<int64_t>(lr_bound, 1);
    if (lr_bound > WAVE_MAX_WINDOW) {
        // Some window may exceed what the wave kernels stage in LDS.  Which records go to the long list depends on the
        // window size, which only the run knows; upload kept the reads whose alignedLength alone is long (spliced reads,
        // large deletions), so the count needs no second look at the caller's arrays (they need not outlive results).
        uint32_t n_long = s.hist[LONG_LIST];  // reads beyond 512 bases
        int64_t max_lr = 1;
        int max_lq = s.hist[LONG_LIST] ? std::min(s.max_lq, MAX_LONG_QUERY) : 1;
        if (2 * w + WIDE_MIN_SPAN > WAVE_MAX_WINDOW || s.wide_all) {
            // with this window size a read of ordinary span may have a long window: any record that carries bases may
            // land on the long list
            n_long = s.out_bound;
            max_lr = lr_bound;
            max_lq = std::min(std::max(s.max_lq, 1), MAX_LONG_QUERY);
        } else {
            for (const auto &wr : s.wide)
                if (wr.first + 2 * w > WAVE_MAX_WINDOW) {
                    if (wr.second <= 16 * class_rows(NUM_CLASSES - 1)) n_long++;  // (longer reads are counted already)
                    max_lq = std::max(max_lq, std::min(wr.second, MAX_LONG_QUERY));
                    max_lr = std::max(max_lr, std::min<int64_t>(wr.first + 2 * w, lr_bound));
                }
            if (s.hist[LONG_LIST]) max_lr = lr_bound;
        }
        s.bound[LONG_LIST] = std::min(n_long, s.out_bound);
        s.long_max_lq = max_lq;
        s.long_max_lr = (int)max_lr;
    }
    return 0;
}

// Enqueue one whole run of the slot's uploaded batch on its stream (two-pass path: nothing is read back).
int enqueue_run(fadehip_ctx *ctx, Slot &s) {
    hipStream_t st = s.stream;
    const int n = s.n_reads;
    const int floor_len = s.floor_len;
    s.ev_used = 0;
    s.fwd_spans.clear();
    s.tb_spans.clear();
    s.tickets_used = 0;
    memset(s.prof_counts, 0, sizeof s.prof_counts);
    int rc;
    // (the file path's batches differ a little in size from call to call: its buffers get headroom so that they settle)
    auto rsv = [&](DevBuf &b, size_t bytes) { return s.device_only ? reserve_roomy(ctx, b, bytes) : reserve(ctx, b, bytes); };
    // one result array for all lists: an alignment reports into the entry the gate hands it (Work::out)
    uint32_t total_bound = 0;
    for (int c = 0; c < NUM_LISTS; c++) total_bound += s.bound[c];
    total_bound = std::min(total_bound, s.out_bound);
    s.out_cap = total_bound;
    s.res_aln_off = ((size_t)n + 255) & ~(size_t)255;
    if ((rc = rsv(s.rs, (size_t)n)) || (rc = rsv(s.aln, sizeof(fadehip_aln) * (size_t)std::max<uint32_t>(total_bound, 1))) ||
        (rc = reserve(ctx, s.zblock, Slot::ZB_BYTES)) ||
        (!s.device_only && (rc = reserve_pinned(ctx, s.res, s.res_aln_off + sizeof(fadehip_aln) * (size_t)total_bound))))
        return rc;
    for (int c = 0; c < NUM_LISTS; c++) {
        if (!s.bound[c]) continue;
        if ((rc = rsv(s.work[c], sizeof(Work) * (size_t)s.bound[c])) || (rc = rsv(s.meta[c], sizeof(Meta) * (size_t)s.bound[c])))
            return rc;
    }
    const int64_t budget = ctx->prm.trace_bytes > 0 ? ctx->prm.trace_bytes : ((int64_t)16 << 30);
    // snapshots only when the slot's previous run had many pass-2 candidates (plan_two_pass); FADEHIP_CKPT=0/1 pins it
    s.use_ckpt = s.last_cand * 32 > std::max<int64_t>(s.last_aln, 1);
    if (const char *kv = getenv("FADEHIP_CKPT")) s.use_ckpt = atoi(kv) != 0;
    auto class_run = [&](int c) {
        ClassRun cr;
        cr.cls = c;
        cr.work = (const Work *)s.work[c].p;
        cr.meta = (const Meta *)s.meta[c].p;
        cr.n_bound = (int)s.bound[c];
        cr.count_dev = s.d_counters() + c;
        cr.max_lr = c == LONG_LIST ? s.long_max_lr : std::max(s.wave_lr_bound, 1);
        cr.q_nib = d_arr<uint8_t>(s, A_SEQ);
        cr.r_nib = (const uint8_t *)ctx->genome.p;
        cr.out = (fadehip_aln *)s.aln.p;
        cr.rs = (uint8_t *)s.rs.p;
        cr.floor_len = floor_len;
        cr.gate = 1;
        cr.budget = budget;
        cr.timed = true;
        return cr;
    };
    if (ctx->two_pass) {
        // the scratch of every class of this run, sized once before the first launch (nothing is re-allocated between
        // kernels already queued; the slot is idle here: run waited for its previous results)
        size_t need_ckpt = 0, need_fwd = 0, need_trace = 0, need_cand = 0;
        for (int c = 0; c < NUM_CLASSES; c++) {
            if (!s.bound[c]) continue;
            TwoPassPlan p;
            if ((rc = plan_two_pass(ctx, s, class_run(c), p))) return rc;
            need_ckpt = std::max(need_ckpt, p.ckpt_bytes);
            need_fwd = std::max(need_fwd, p.fwd_bytes);
            need_trace = std::max(need_trace, p.trace_bytes);
            need_cand = std::max(need_cand, p.cand_bytes);
        }
        if ((rc = reserve_run(ctx, s.trash, s.ckpt, need_ckpt)) || (rc = reserve_run(ctx, s.trash, s.fwd, need_fwd)) ||
            (rc = reserve_run(ctx, s.trash, s.trace, need_trace)) || (rc = reserve_run(ctx, s.trash, s.cand, need_cand)))
            return rc;
    }
    HIPCHK(ctx, hipMemsetAsync(s.zblock.p, 0, Slot::ZB_BYTES, st));  // every counter of the run in one fill
    for (int c = 0; c < NUM_CLASSES; c++) s.sel_fresh[c] = true;
    if ((rc = record(ctx, s, &s.ev_gate0))) return rc;
    GateArgs g;
    memset(&g, 0, sizeof g);
    g.n_reads = n;
    g.tid = d_arr<int32_t>(s, A_TID);
    g.pos = d_arr<int32_t>(s, A_POS);
    g.l_seq = d_arr<int32_t>(s, A_LSEQ);
    g.flag = d_arr<uint16_t>(s, A_FLAG);
    g.has_sa = d_arr<uint8_t>(s, A_SA);
    g.cigar_off = d_arr<uint32_t>(s, A_CIGOFF);
    g.cigar_ops = d_arr<uint32_t>(s, A_CIG);
    g.seq_off = d_arr<uint32_t>(s, A_SEQOFF);
    g.floor_len = floor_len;
    g.window = s.window;
    g.n_contigs = ctx->n_contigs;
    g.contig_len = (const int64_t *)ctx->contig_len.p;
    g.contig_base = (const uint64_t *)ctx->contig_base.p;
    g.max_ref_len = ctx->prm.max_ref_len;
    g.wave_lr_bound = s.wave_lr_bound;
    g.long_lr_bound = s.long_max_lr;
    g.long_lq_bound = std::max(s.long_max_lq, 1);
    g.rs = (uint8_t *)s.rs.p;
    for (int c = 0; c < NUM_LISTS; c++) {
        g.work[c] = (Work *)s.work[c].p;
        g.meta[c] = (Meta *)s.meta[c].p;
        g.list_cap[c] = s.bound[c];
    }
    g.out_cap = s.out_cap;
    g.n_cigar_ops = (uint32_t)(s.L.bytes[A_CIG] / 4);
    g.n_seq_bytes = (uint32_t)s.L.bytes[A_SEQ];
    g.stats = s.d_stats();
    g.counters = s.d_counters();
    g.counters64 = s.d_counters64();
    hipLaunchKernelGGL(gate_kernel, dim3((n + GATE_BLOCK - 1) / GATE_BLOCK), dim3(GATE_BLOCK), 0, st, g);
    HIPCHK(ctx, hipGetLastError());
    if ((rc = record(ctx, s, &s.ev_gate1))) return rc;
    uint32_t exact[NUM_LISTS];
    for (int c = 0; c < NUM_LISTS; c++) exact[c] = s.bound[c];
    if (!ctx->two_pass) {
        // single-pass kernels: their launches take the real counts
        HIPCHK(ctx, hipMemcpyAsync(s.h_zb, s.zblock.p, 128, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        for (int c = 0; c < NUM_LISTS; c++) exact[c] = std::min(s.h_counters()[c], s.bound[c]);
    }
    for (int c = 0; c < NUM_CLASSES; c++) {
        if (!exact[c]) continue;
        ClassRun cr = class_run(c);
        cr.n_bound = (int)exact[c];
        rc = ctx->two_pass ? run_class_two_pass(ctx, s, st, cr) : run_class_single(ctx, s, st, cr);
        if (rc) return rc;
    }
    if (exact[LONG_LIST]) {
        ClassRun cr = class_run(LONG_LIST);
        cr.n_bound = (int)exact[LONG_LIST];
        if ((rc = run_long(ctx, s, st, cr, std::max(s.long_max_lq, 1)))) return rc;
    }
    if ((rc = record(ctx, s, &s.ev_end))) return rc;
    // results to the slot's pinned block; the counter block tells the host how many entries of each list are live
    HIPCHK(ctx, hipMemcpyAsync(s.h_zb, s.zblock.p, Slot::ZB_BYTES, hipMemcpyDeviceToHost, st));
    if (s.device_only) return 0;
    HIPCHK(ctx, hipMemcpyAsync(s.res.p, s.rs.p, (size_t)n, hipMemcpyDeviceToHost, st));
    if (total_bound)
        HIPCHK(ctx, hipMemcpyAsync(s.res.p + s.res_aln_off, s.aln.p, sizeof(fadehip_aln) * (size_t)total_bound, hipMemcpyDeviceToHost, st));
    return 0;
}

// Wait for the slot's run and turn the counter block into results (state 2 -> 3).
int finish_run(fadehip_ctx *ctx, Slot &s, int slot) {
    if (s.state == 3) return 0;
    if (s.state != 2) return set_err(ctx, FADEHIP_E_STATE, "slot %d has not been run", slot);
    hipStream_t st = s.stream;
    const int n = s.n_reads;
    if (n == 0) {
        s.n_aln = 0;
        s.n_oversize = 0;
        memset(s.stats, 0, sizeof s.stats);
        s.stats[0] = s.n_skipped;
        s.state = 3;
        return 0;
    }
    HIPCHK(ctx, hipStreamSynchronize(st));
    for (void *q : s.trash) (void)hipFree(q);  // scratch outgrown while this run was enqueued
    s.trash.clear();
    const uint32_t errbits = s.h_counters()[2 * NUM_LISTS];
    if (errbits) {
        s.state = 0;
        if (errbits & 64u) return set_err(ctx, FADEHIP_E_INVALID, "batch has a record whose cigar_off / seq_off are not non-decreasing within the arrays, or l_seq < 0");
        if (errbits & 8u) return set_err(ctx, FADEHIP_E_INVALID, "batch has a mapped soft-clipped record whose seq_packed slice is shorter than its l_seq");
        if (errbits & 16u) return set_err(ctx, FADEHIP_E_INVALID, "batch has a record whose cigar.alignedLength exceeds ref_span_bound=%lld", (long long)s.span_bound);
        if (errbits & 32u) return set_err(ctx, FADEHIP_E_INVALID, "batch has more records to re-align than its bounds said (n_with_seq / l_seq_min / l_seq_max too small?)");
        return set_err(ctx, FADEHIP_E_INVALID, "batch has a mapped soft-clipped read whose tid is not a contig of the uploaded genome");
    }
    for (int c = 0; c < NUM_CLASSES; c++)  // what pass 2 served: sizes the next run's persistent launch
        if (s.bound[c]) {
            const uint32_t *bn = (const uint32_t *)(s.h_zb + Slot::ZB_SEL + Slot::ZB_SEL_STRIDE * (size_t)c);
            int octs = 0;
            for (int b = 0; b < NUM_BUCKETS; b++) octs += (int)((bn[b] + 7) / 8);
            s.p2_last_octs[c] = octs;
        }
    // all lists report into one result array: its live entries are the ones the gate handed out
    const uint32_t at = std::min(s.h_counters()[2 * NUM_LISTS + 3], s.out_cap);
    s.n_aln = (int)at;
    s.n_oversize = (int)s.h_counters()[2 * NUM_LISTS + 2];
    for (int k = 0; k < 8; k++) {
        s.stats[k] = 0;
        for (int q = 0; q < STAT_PARTS; q++) s.stats[k] += (int64_t)s.h_stats()[8 * q + k];
    }
    s.stats[0] += s.n_skipped;  // records the caller left out (anno.d:61-65: rs = 0) are reads all the same
    const PlanOut *po = s.h_plan();
    s.n_cand = (int64_t)po->cand_total;
    s.n_rerun = (int64_t)po->rerun_total;
    s.last_cand = s.n_cand;
    s.last_aln = (int64_t)at;
    s.prof_counts[0] = at;
    s.prof_counts[1] = (int64_t)s.h_counters64()[0 * C64_STRIDE];
    // algorithmic bytes of the dominant kernel: SURVEY §8(d)'s packed query + packed window + 16 B
    // descriptor + 64 B result slot per alignment
    s.prof_counts[3] = (int64_t)s.h_counters64()[1 * C64_STRIDE] + (int64_t)at * 80;
    s.prof_counts[4] = ctx->two_pass ? (s.use_ckpt ? (int64_t)s.h_counters64()[2 * C64_STRIDE] : 0) : (int64_t)(s.h_counters64()[0 * C64_STRIDE] / 2);
    s.prof_counts[5] = s.n_cand;
    if (ctx->debug)
        fprintf(stderr, "[fadehip] slot %d: %d reads, %u alignments, %lld candidates traced, %lld re-run, trace need %lld B\n", slot, n, at,
                (long long)s.n_cand, (long long)s.n_rerun, (long long)s.prof_counts[2]);
    s.state = 3;
    return 0;
}


// The BGZF members of p[0, n): where each one's DEFLATE stream lies, its ISIZE and CRC32 (SAM spec 4.1: gzip member with
// FEXTRA and the subfield 'B','C' holding BSIZE = member size - 1).  Stops in front of a member that is not whole
// (*consumed = bytes of whole members); false + msg for bytes that are not a BGZF member.
bool scan_bgzf_members(const uint8_t *p, size_t n, std::vector<bgzf::InflateBlock> &blocks, size_t *consumed, uint64_t *total_out, std::string &msg) {
    size_t at = 0;
    uint64_t out = *total_out;
    while (n - at >= 18) {
        const uint8_t *m = p + at;
        if (m[0] != 0x1f || m[1] != 0x8b || m[2] != 8 || !(m[3] & 4)) {
            msg = "not a BGZF member at byte " + std::to_string(at) + " (gzip magic / FEXTRA missing)";
            return false;
        }
        const size_t xlen = (size_t)m[10] | ((size_t)m[11] << 8);
        if (n - at < 12 + xlen) break;
        size_t bsize = 0;
        for (size_t x = 12; x + 4 <= 12 + xlen;) {
            const size_t slen = (size_t)m[x + 2] | ((size_t)m[x + 3] << 8);
            if (m[x] == 'B' && m[x + 1] == 'C' && slen == 2 && x + 6 <= 12 + xlen) bsize = ((size_t)m[x + 4] | ((size_t)m[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 2 + 8) {
            msg = "BGZF member at byte " + std::to_string(at) + " has no BC subfield or an impossible BSIZE";
            return false;
        }
        if (n - at < bsize) break;
        bgzf::InflateBlock b;
        b.src_off = at + 12 + xlen;
        b.src_len = (uint32_t)(bsize - 12 - xlen - 8);
        memcpy(&b.crc, m + bsize - 8, 4);
        memcpy(&b.isize, m + bsize - 4, 4);
        b.dst_off = out;
        b.pad = 0;
        if (b.isize > 65536u) {
            msg = "BGZF member at byte " + std::to_string(at) + " claims ISIZE " + std::to_string(b.isize) + " (at most 65536)";
            return false;
        }
        out += b.isize;
        blocks.push_back(b);
        at += bsize;
    }
    *consumed = at;
    *total_out = out;
    return true;
}

const char *inflate_error_name(uint32_t e) {
    static const char *const nm[] = {"ok", "reserved block type", "stored block LEN/NLEN mismatch", "bad dynamic-Huffman header", "invalid code",
                                     "distance beyond the block's start", "more bytes than ISIZE", "stream runs past the member's end",
                                     "fewer bytes than ISIZE", "CRC32 mismatch"};
    return e < sizeof nm / sizeof nm[0] ? nm[e] : "unknown";
}

// the inflate launch: as many waves as the device holds, each drawing members from the ticket
int launch_inflate(fadehip_ctx *ctx, hipStream_t st, const bgzf::InflateArgs &a) {
    HIPCHK(ctx, hipMemsetAsync(a.ticket, 0, 8, st));
    const unsigned wgs = (a.n_blocks + bgzf::INF_WAVES - 1) / bgzf::INF_WAVES;
    const unsigned grid = std::max(1u, std::min<unsigned>(wgs, (unsigned)std::max(ctx->cu_count, 1) * 8u));
    hipLaunchKernelGGL(bgzf::bgzf_inflate_kernel, dim3(grid), dim3(bgzf::INF_WG), 0, st, a);
    HIPCHK(ctx, hipGetLastError());
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
    p->max_ref_len = 1 << 20;
    p->max_batch_reads = 1 << 20;
    p->trace_bytes = 0;
    p->trace_all = 0;
    p->rules = FADEHIP_RULES_DEFAULT;
}

int fadehip_abi_version(void) { return FADEHIP_ABI_VERSION; }

const char *fadehip_last_error(const fadehip_ctx *ctx) {
    if (!ctx || g_err_ctx == ctx) return g_err.c_str();
    thread_local std::string copy;
    {
        std::lock_guard<std::mutex> l(const_cast<fadehip_ctx *>(ctx)->err_mu);
        copy = ctx->err;
    }
    return copy.c_str();
}

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
        if (ctx->prm.max_ref_len <= 0) ctx->prm.max_ref_len = 1 << 20;
        if (ctx->prm.max_batch_reads <= 0) ctx->prm.max_batch_reads = 1 << 20;
        if (ctx->prm.rules == 0) ctx->prm.rules = FADEHIP_RULES_DEFAULT;
    }
    int rc = 0;
    auto fail = [&](int code) {
        {
            std::lock_guard<std::mutex> l(ctx->err_mu);
            g_err = ctx->err;
        }
        g_err_ctx = nullptr;
        fadehip_destroy(ctx);
        return code;
    };
    if (ctx->prm.max_ref_len > (1 << 20)) {
        set_err(ctx, FADEHIP_E_UNSUPPORTED, "max_ref_len %d exceeds 2^20", ctx->prm.max_ref_len);
        return fail(FADEHIP_E_UNSUPPORTED);
    }
    if (ctx->prm.rules & ~(uint32_t)FADEHIP_RULES_DEFAULT) {
        set_err(ctx, FADEHIP_E_INVALID, "unknown rule bits 0x%x", ctx->prm.rules & ~(uint32_t)FADEHIP_RULES_DEFAULT);
        return fail(FADEHIP_E_INVALID);
    }
    if ((rc = build_score_tab(ctx, ctx->prm, ctx->sc))) return fail(rc);
    // FADEHIP_BLOCKING_SYNC=1 (the `fade` driver's file path sets it): a thread that waits for the device sleeps instead of
    // spinning.  A file-to-file run keeps every host core busy inflating; the two threads that wait for the front and the
    // back half of each call would otherwise burn a core each.  Not the default: a wake-up costs tens of microseconds,
    // which the level-2 pipeline (a result every millisecond) does not have to spare.
    if (const char *kv = getenv("FADEHIP_BLOCKING_SYNC")) {
        if (atoi(kv)) {
            ctx->blocking_sync = true;
            (void)hipSetDevice(device);
            if (hipSetDeviceFlags(hipDeviceScheduleBlockingSync) != hipSuccess) (void)hipGetLastError();  // (a device already in use keeps its flags)
        }
    }
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
    // (per row pair and below the f16 infinity pattern 0x7c00: 64 * score for the row classes <= 14, i.e. scores <= 448,
    // 32 * score for 16 .. 24, i.e. scores <= 768), DP values are 8 * score in int16 compared as f16 patterns.  FADE's scoring (match 2) fits everything; larger match scores take the path
    // whose ranges still hold: the single-pass packed kernel (32-bit keys) up to match 7, the int32 kernel beyond.
    if (ctx->prm.match > 2) ctx->two_pass = false;
    if (8 * (ctx->prm.match * FADEHIP_MAX_QUERY + ctx->prm.open + std::max(ctx->prm.match, 0)) >= 0x7c00) ctx->use_packed = false;
    if (!ctx->two_pass && ctx->prm.rules != FADEHIP_RULES_DEFAULT) {
        set_err(ctx, FADEHIP_E_UNSUPPORTED, "the rule switches exist on the two-pass path only (match <= 2, FADEHIP_KERNEL unset)");
        return fail(FADEHIP_E_UNSUPPORTED);
    }
    if (const char *kv = getenv("FADEHIP_SPAN_SLACK")) ctx->span_slack = atoi(kv);
    if (const char *kv = getenv("FADEHIP_TAIL_CUS")) ctx->tail_cus_per_xcd = std::max(0, std::min(atoi(kv), 8));
    if (const char *kv = getenv("FADEHIP_BAM_SPLIT")) ctx->split_cus = std::max(0, std::min(atoi(kv), 31));
    if (const char *kv = getenv("FADEHIP_P2_WAVES")) ctx->p2_waves_fixed = std::max(0, atoi(kv));
    if (const char *kv = getenv("FADEHIP_SCORE_PERSIST")) ctx->score_persist = atoi(kv) != 0;
    if (const char *kv = getenv("FADEHIP_SCORE_G8")) ctx->score_g8 = atoi(kv);
    ctx->debug = getenv("FADEHIP_DEBUG") != nullptr;
    uint8_t table[256];
    fill_ascii_table(table);
    // (everything the library enqueues goes to streams of its own: the null stream would be one more HSA queue, i.e. one
    // more 173 MB context-save area in host memory, for two copies)
    if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMemcpyToSymbolAsync(HIP_SYMBOL(c_ascii_code), table, 256, 0, hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess ||
        hipStreamSynchronize(ctx->copy_stream) != hipSuccess) {
        set_err(ctx, FADEHIP_E_HIP, "hipMemcpyToSymbol failed: %s", hipGetErrorString(hipGetLastError()));
        return fail(FADEHIP_E_HIP);
    }
    static_assert(sizeof(uint32_t) * (2 * NUM_LISTS + 4) <= Slot::ZB_C64 - Slot::ZB_COUNTERS, "gate counters overflow their slice");
    static_assert(Slot::ZB_SEL + Slot::ZB_SEL_STRIDE * NUM_CLASSES <= Slot::ZB_STATS, "selection counters overlap the stats");
    static_assert(sizeof(uint32_t) * NUM_BUCKETS <= Slot::ZB_SEL_STRIDE, "selection counters overflow their slice");
    static_assert(sizeof(PlanOut) <= 128, "PlanOut overflows its slice");
    for (int k = 0; k < FADEHIP_NUM_SLOTS; k++)
        for (int c = 0; c < NUM_CLASSES; c++) ctx->slots[k].p2_last_octs[c] = -1;
    *out = ctx;
    return 0;
}

void fadehip_destroy(fadehip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (int k = 0; k < FADEHIP_NUM_SLOTS; k++) {
        Slot &s = ctx->slots[k];
        for (DevBuf *b : {&s.in[0], &s.in[1], &s.rs, &s.fwd, &s.aln, &s.zblock, &s.trace, &s.ckpt, &s.cand, &s.lrows}) release(*b);
        for (void *q : s.trash) (void)hipFree(q);
        s.trash.clear();
        for (int c = 0; c < NUM_LISTS; c++) {
            release(s.work[c]);
            release(s.meta[c]);
        }
        release(s.stage[0]);
        release(s.stage[1]);
        release(s.res);
        if (s.ev_copied) (void)hipEventDestroy(s.ev_copied);
        for (hipEvent_t e : s.ev) (void)hipEventDestroy(e);
        if (s.h_zb) (void)hipHostFree(s.h_zb);
        if (s.score_stream) (void)hipStreamDestroy(s.score_stream);
        if (s.stream && s.stream != ctx->copy_stream) (void)hipStreamDestroy(s.stream);
    }
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    for (BgzfLane &l : ctx->bgzf) {
        for (DevBuf *b : {&l.src, &l.slots, &l.meta, &l.member_off}) release(*b);
        if (l.done) (void)hipEventDestroy(l.done);
        release(l.out);
        if (l.h_total) (void)hipHostFree(l.h_total);
        if (l.stream && (&l == &ctx->bgzf[0] || l.stream != ctx->bgzf[0].stream)) (void)hipStreamDestroy(l.stream);
    }
    for (DevBuf *b : {&ctx->inf.comp, &ctx->inf.blocks, &ctx->inf.out, &ctx->inf.status, &ctx->inf.ticket}) release(*b);
    release(ctx->inf.h_status);
    if (ctx->inf.stream) (void)hipStreamDestroy(ctx->inf.stream);
    release(ctx->genome);
    for (DevBuf *b : {&ctx->l1_q, &ctx->l1_r, &ctx->l1_qn, &ctx->l1_rn, &ctx->l1_bad, &ctx->l1_work, &ctx->l1_aln}) release(*b);
    release(ctx->contig_len);
    release(ctx->contig_base);
    if (g_err_ctx == ctx) g_err_ctx = nullptr;
    delete ctx;
}

int fadehip_host_alloc(fadehip_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return set_err(ctx, FADEHIP_E_INVALID, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return pin_alloc(ctx, bytes, out);
}

int fadehip_host_free(fadehip_ctx *ctx, void *p) {
    if (!p) return 0;
    return pin_free(ctx, p);
}

int fadehip_host_register(fadehip_ctx *ctx, void *p, size_t bytes) {
    if (!ctx || !p || !bytes) return set_err(ctx, FADEHIP_E_INVALID, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostRegister(p, bytes, hipHostRegisterDefault));
    std::lock_guard<std::mutex> l(g_pin_mu);
    g_pin_registered[p] = false;  // (the caller's memory: fadehip_host_free only takes the registration back)
    return 0;
}

size_t fadehip_batch_bytes(int32_t n_reads, int64_t n_cigar_ops, int64_t n_seq_bytes) {
    if (n_reads < 0 || n_cigar_ops < 0 || n_seq_bytes < 0) return 0;
    return batch_layout(n_reads, n_cigar_ops, n_seq_bytes).total;
}

int fadehip_batch_bind(void *base, int32_t n_reads, int64_t n_cigar_ops, int64_t n_seq_bytes, fadehip_read_batch *b) {
    if (!base || !b || n_reads < 0 || n_cigar_ops < 0 || n_seq_bytes < 0) return set_err(nullptr, FADEHIP_E_INVALID, "bad batch_bind arguments");
    if ((uintptr_t)base & 255u) return set_err(nullptr, FADEHIP_E_INVALID, "a batch block must be 256-byte aligned (fadehip_host_alloc memory is)");
    const Layout L = batch_layout(n_reads, n_cigar_ops, n_seq_bytes);
    uint8_t *p = (uint8_t *)base;
    b->n_reads = n_reads;
    b->tid = (const int32_t *)(p + L.off[A_TID]);
    b->pos = (const int32_t *)(p + L.off[A_POS]);
    b->l_seq = (const int32_t *)(p + L.off[A_LSEQ]);
    b->cigar_off = (const uint32_t *)(p + L.off[A_CIGOFF]);
    b->seq_off = (const uint32_t *)(p + L.off[A_SEQOFF]);
    b->flag = (const uint16_t *)(p + L.off[A_FLAG]);
    b->has_sa = (const uint8_t *)(p + L.off[A_SA]);
    b->cigar_ops = (const uint32_t *)(p + L.off[A_CIG]);
    b->seq_packed = (const uint8_t *)(p + L.off[A_SEQ]);
    b->n_skipped = 0;
    b->ref_span_bound = 0;
    b->n_with_seq = 0;
    b->l_seq_min = b->l_seq_max = 0;
    b->reserved = 0;
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
    {
        const int erc = ensure_slot(ctx, s);
        if (erc) return erc;
    }
    if (s.state == 2) HIPCHK(ctx, hipStreamSynchronize(s.stream));
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
        w.out = 0;
        lists[cls].push_back(w);
        max_lr[cls] = std::max(max_lr[cls], (int)lr);
    }
    if (q_total >= (int64_t)1 << 32) return set_err(ctx, FADEHIP_E_UNSUPPORTED, "query bases per batch must stay below 2^32");
    DevBuf &d_q = ctx->l1_q, &d_r = ctx->l1_r, &d_qn = ctx->l1_qn, &d_rn = ctx->l1_rn, &d_bad = ctx->l1_bad,
           &d_work = ctx->l1_work, &d_aln = ctx->l1_aln;
    int rc = 0;
    hipStream_t st = s.stream;
    size_t n_work = 0;
    for (int c = 0; c < NUM_LISTS; c++) n_work += lists[c].size();
    if ((rc = reserve(ctx, d_q, (size_t)q_total + 2)) || (rc = reserve(ctx, d_r, (size_t)r_total + 2)) ||
        (rc = reserve(ctx, d_qn, (size_t)q_total / 2 + 8)) || (rc = reserve(ctx, d_rn, (size_t)r_total / 2 + 8)) ||
        (rc = reserve(ctx, d_bad, 4)) || (rc = reserve(ctx, d_aln, (size_t)n * sizeof(fadehip_aln))) ||
        (rc = reserve(ctx, d_work, std::max<size_t>(1, n_work) * sizeof(Work))) || (rc = reserve(ctx, s.zblock, Slot::ZB_BYTES)))
        return rc;
    HIPCHK(ctx, hipMemcpyAsync(d_q.p, q, (size_t)q_total, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(d_r.p, r, (size_t)r_total, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemsetAsync(d_bad.p, 0, 4, st));
    if (q_total > 0)
        hipLaunchKernelGGL(pack_ascii_kernel, dim3((unsigned)((q_total / 2 + 256) / 256)), dim3(256), 0, st,
                           (const uint8_t *)d_q.p, (uint64_t)q_total, (uint64_t)0, (uint8_t *)d_qn.p, 0, (int *)d_bad.p);
    if (r_total > 0)
        hipLaunchKernelGGL(pack_ascii_kernel, dim3((unsigned)((r_total / 2 + 256) / 256)), dim3(256), 0, st,
                           (const uint8_t *)d_r.p, (uint64_t)r_total, (uint64_t)0, (uint8_t *)d_rn.p, 0, (int *)d_bad.p);
    HIPCHK(ctx, hipGetLastError());
    {
        size_t base = 0;
        for (int c = 0; c < NUM_LISTS; c++) {
            if (lists[c].empty()) continue;
            for (size_t k = 0; k < lists[c].size(); k++) lists[c][k].out = (uint32_t)(base + k);  // one result array, list after list
            HIPCHK(ctx, hipMemcpyAsync((Work *)d_work.p + base, lists[c].data(), lists[c].size() * sizeof(Work), hipMemcpyHostToDevice, st));
            base += lists[c].size();
        }
    }
    const int64_t budget = ctx->prm.trace_bytes > 0 ? ctx->prm.trace_bytes : ((int64_t)4 << 30);
    {
        size_t base = 0;
        s.fwd_spans.clear();
        s.tb_spans.clear();
        s.tickets_used = 0;
        HIPCHK(ctx, hipMemsetAsync(s.zblock.p, 0, Slot::ZB_BYTES, st));
        for (int c = 0; c < NUM_CLASSES; c++) s.sel_fresh[c] = true;
        // level 1 wants a CIGAR for every pair: whatever the forced-diagonal shortcut leaves goes to pass 2, with snapshots
        s.use_ckpt = true;
        if (const char *kv = getenv("FADEHIP_CKPT")) s.use_ckpt = atoi(kv) != 0;
        for (int c = 0; c < NUM_LISTS; c++) {
            if (lists[c].empty()) continue;
            // the lists come from the host here: their counts go where the gate leaves them at level 2
            hipLaunchKernelGGL(set_counts_kernel, dim3(1), dim3(1), 0, st, s.d_counters() + c, (uint32_t)lists[c].size());
            HIPCHK(ctx, hipGetLastError());
            ClassRun cr;
            cr.cls = c;
            cr.work = (const Work *)d_work.p + base;
            cr.meta = nullptr;
            cr.n_bound = (int)lists[c].size();
            cr.count_dev = s.d_counters() + c;
            cr.max_lr = max_lr[c];
            cr.q_nib = (const uint8_t *)d_qn.p;
            cr.r_nib = (const uint8_t *)d_rn.p;
            cr.out = (fadehip_aln *)d_aln.p;
            cr.rs = nullptr;
            cr.floor_len = 0;
            cr.gate = 0;
            cr.budget = budget;
            cr.timed = false;
            if (c == LONG_LIST) rc = run_long(ctx, s, st, cr, max_long_lq);
            else rc = ctx->two_pass ? run_class_two_pass(ctx, s, st, cr) : run_class_single(ctx, s, st, cr);
            if (rc) {
                (void)hipStreamSynchronize(st);
                return rc;
            }
            base += lists[c].size();
        }
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    s.state = 0;  // slot 0's level-2 buffers were borrowed
    std::vector<fadehip_aln> h_aln(n_work);
    if (n_work) HIPCHK(ctx, hipMemcpy(h_aln.data(), d_aln.p, n_work * sizeof(fadehip_aln), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n_work; k++) out[h_aln[k].read_idx] = h_aln[k].sw;
    // empty query or reference: nothing to align (no DP): all of the query is soft-clipped
    for (int k : degenerate) {
        fadehip_sw_result o;
        memset(&o, 0, sizeof o);
        o.end_query = o.end_ref = -1;
        const int64_t lq = q_off[k + 1] - q_off[k];
        if (lq > 0 && (ctx->sc.rules & FADEHIP_RULE_PAD_SOFTCLIP)) {
            o.n_ops = 1;
            o.ops[0] = ((uint32_t)lq << 4) | 4u;
        }
        out[k] = o;
    }
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
    hipStream_t st = ctx->copy_stream;
    HIPCHK(ctx, hipMemsetAsync(ctx->genome.p, 0, ctx->genome.cap, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->contig_len.p, lengths, sizeof(int64_t) * n_contigs, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipMemcpyAsync(ctx->contig_base.p, ctx->h_contig_base.data(), sizeof(uint64_t) * n_contigs, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st));  // (the two host arrays are pageable)
    const size_t CH = (size_t)64 << 20;  // staging chunk, even
    DevBuf stage, bad;
    if ((rc = reserve(ctx, stage, CH)) || (rc = reserve(ctx, bad, 4))) {
        release(stage);
        release(bad);
        return rc;
    }
    hipError_t e = hipMemsetAsync(bad.p, 0, 4, st);
    for (int c = 0; c < n_contigs && e == hipSuccess; c++) {
        for (int64_t off = 0; off < lengths[c] && e == hipSuccess; off += (int64_t)CH) {
            const size_t nb = (size_t)std::min<int64_t>((int64_t)CH, lengths[c] - off);
            e = hipMemcpyAsync(stage.p, seqs[c] + off, nb, hipMemcpyHostToDevice, st);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(pack_ascii_kernel, dim3((unsigned)((nb / 2 + 256) / 256)), dim3(256), 0, st,
                               (const uint8_t *)stage.p, (uint64_t)nb, ctx->h_contig_base[c] + (uint64_t)off,
                               (uint8_t *)ctx->genome.p, 1, (int *)bad.p);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(st);  // the staging chunk is reused
        }
    }
    int h_bad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, bad.p, 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
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
    if (!b || b->n_reads < 0 || b->n_skipped < 0 || b->ref_span_bound < 0) return set_err(ctx, FADEHIP_E_INVALID, "bad batch");
    if (b->n_reads > ctx->prm.max_batch_reads)
        return set_err(ctx, FADEHIP_E_INVALID, "batch of %d reads exceeds max_batch_reads %d", b->n_reads, ctx->prm.max_batch_reads);
    if (ctx->n_contigs == 0) return set_err(ctx, FADEHIP_E_STATE, "fadehip_genome_upload has not been called");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    Slot &s = ctx->slots[slot];
    if ((rc = ensure_slot(ctx, s))) return rc;
    if (!ctx->copy_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));  // (made on first use: the file path never uploads)
    // The batch goes to the input buffer the run in flight (if any) does not use, on the ctx's copy stream: upload never
    // waits for that run, and the run's results stay valid until the slot is RUN again.
    Slot::Pending &nx = s.next;
    if (nx.valid) HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));  // an earlier upload that was never run: same buffer
    nx.valid = false;
    const int n = b->n_reads;
    nx.n_reads = n;
    nx.n_skipped = b->n_skipped;
    nx.buf = 1 - s.cur;
    memset(nx.hist, 0, sizeof nx.hist);
    nx.out_bound = 0;
    nx.wide.clear();
    nx.max_lq = 0;
    nx.span_bound = b->ref_span_bound;
    nx.h_base = nullptr;
    if (n == 0) { nx.valid = true; return 0; }
    if (!b->tid || !b->pos || !b->flag || !b->has_sa || !b->l_seq || !b->cigar_off || !b->seq_off ||
        (b->cigar_off[n] && !b->cigar_ops) || (b->seq_off[n] && !b->seq_packed))
        return set_err(ctx, FADEHIP_E_INVALID, "batch has NULL arrays");
    const size_t n_cig = b->cigar_off[n], n_seq = b->seq_off[n];
    if ((uint64_t)n_seq * 2 >= ((uint64_t)1 << 32)) return set_err(ctx, FADEHIP_E_UNSUPPORTED, "packed sequence bytes per batch must stay below 2^31");
    // What the launches of the run are sized from: records that carry bases per read-length class, the longest read, the
    // longest cigar.alignedLength.  A caller that packed the block knows them (ABI 3: n_with_seq, l_seq_min / l_seq_max,
    // ref_span_bound) and upload then touches no record: the offsets the kernels index with are checked by the gate
    // kernel before it reads through them.  Otherwise one pass over the records finds them (and checks the offsets
    // here, as ABI 2 did).  The CIGARs are scanned when the caller gave no span bound, or one so long that a window may
    // exceed what the wave kernels stage (then the few reads concerned are remembered for plan_run).
    const bool hinted = b->n_with_seq > 0 && b->ref_span_bound > 0 && b->ref_span_bound <= WIDE_MIN_SPAN &&
                        b->l_seq_min > 0 && b->l_seq_max >= b->l_seq_min;
    uint32_t hist[NUM_LISTS] = {0};
    int max_lq = 0;
    int64_t span = b->ref_span_bound;
    nx.wide.clear();
    if (hinted) {
        if (b->n_with_seq > n) return set_err(ctx, FADEHIP_E_INVALID, "n_with_seq %d exceeds n_reads %d", b->n_with_seq, n);
        const int c_lo = list_of_len(std::min(b->l_seq_min, MAX_LONG_QUERY)), c_hi = list_of_len(std::min(b->l_seq_max, MAX_LONG_QUERY));
        for (int c = c_lo; c <= c_hi; c++) hist[c] = (uint32_t)b->n_with_seq;  // any of them may be of any length in between
        max_lq = std::min(b->l_seq_max, MAX_LONG_QUERY);
        nx.out_bound = (uint32_t)b->n_with_seq;
    } else {
        const bool scan = b->ref_span_bound <= 0, scan_wide = scan || b->ref_span_bound > WIDE_MIN_SPAN;
        uint8_t cls_of[33];  // class of a read of 16 k - 15 .. 16 k bases
        for (int k = 0; k <= 32; k++) cls_of[k] = (uint8_t)list_of_len(std::max(16 * k, 1));
        uint32_t n_out = 0;
        for (int i = 0; i < n; i++) {
            const uint32_t c0 = b->cigar_off[i], c1 = b->cigar_off[i + 1];
            const int lq = b->l_seq[i];
            if (c0 > c1 || c1 > n_cig || b->seq_off[i] > b->seq_off[i + 1] || b->seq_off[i + 1] > n_seq || lq < 0)
                return set_err(ctx, FADEHIP_E_INVALID, "record %d: cigar_off / seq_off must be non-decreasing and l_seq >= 0", i);
            if (b->seq_off[i + 1] == b->seq_off[i]) continue;  // no bases: the record cannot be re-aligned (anno.d:61 settled it)
            if (scan_wide) {
                if (b->flag[i] & 4u) continue;
                const CigarSummary cs = summarize_cigar(b->cigar_ops, c0, c1);
                if (cs.n_soft == 0) continue;
                if (scan) span = std::max(span, cs.aligned);
                if (cs.aligned > WIDE_MIN_SPAN) nx.wide.emplace_back(cs.aligned, lq);
            }
            const int c = lq > 512 ? (lq <= MAX_LONG_QUERY ? LONG_LIST : -1) : (int)cls_of[(std::max(lq, 1) + 15) >> 4];
            if (c >= 0) { hist[c]++; n_out++; }
            max_lq = std::max(max_lq, std::min(lq, MAX_LONG_QUERY));
        }
        nx.out_bound = n_out;
    }
    memcpy(nx.hist, hist, sizeof hist);
    nx.max_lq = max_lq;
    nx.span_bound = std::max<int64_t>(span, 1);
    const Layout L = batch_layout(n, (int64_t)n_cig, (int64_t)n_seq);
    const void *src[N_ARR] = {b->tid, b->pos, b->l_seq, b->cigar_off, b->seq_off, b->flag, b->has_sa, b->cigar_ops, b->seq_packed};
    const uint8_t *base = (const uint8_t *)b->tid;
    bool direct = ((uintptr_t)base & 255u) == 0;
    for (int k = 0; k < N_ARR && direct; k++)
        if (L.bytes[k] && (const uint8_t *)src[k] != base + L.off[k]) direct = false;
    if ((rc = reserve(ctx, s.in[nx.buf], L.total))) return rc;
    if (!direct) {
        // arrays from anywhere: gather them into the pinned staging block of this input buffer (one host copy) for the one DMA
        PinBuf &stage = s.stage[nx.buf];
        if ((rc = reserve_pinned(ctx, stage, L.total))) return rc;
        for (int k = 0; k < N_ARR; k++)
            if (L.bytes[k]) memcpy(stage.p + L.off[k], src[k], L.bytes[k]);
        base = stage.p;
    }
    nx.L = L;
    nx.h_base = base;
    HIPCHK(ctx, hipMemcpyAsync(s.in[nx.buf].p, base, L.off[N_ARR - 1] + L.bytes[N_ARR - 1], hipMemcpyHostToDevice, ctx->copy_stream));
    HIPCHK(ctx, hipEventRecord(s.ev_copied, ctx->copy_stream));
    nx.valid = true;
    return 0;
}

int fadehip_annotate_run(fadehip_ctx *ctx, int slot, int32_t floor_len, int32_t window) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    if (!s.next.valid && !s.have_batch) return set_err(ctx, FADEHIP_E_STATE, "slot %d has no uploaded batch", slot);
    if (window < 0) return set_err(ctx, FADEHIP_E_INVALID, "window must be >= 0");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (s.state == 2) HIPCHK(ctx, hipStreamSynchronize(s.stream));  // results never fetched: the slot's buffers are still in use
    s.state = 0;
    if (s.next.valid) {
        // the batch uploaded last becomes the batch of this (and any repeated) run; its H2D is awaited on the device
        const Slot::Pending &nx = s.next;
        s.cur = nx.buf;
        s.L = nx.L;
        s.h_base = nx.h_base;
        memcpy(s.hist, nx.hist, sizeof s.hist);
        s.max_lq = nx.max_lq;
        s.span_bound = nx.span_bound;
        s.out_bound = nx.out_bound;
        s.wide = nx.wide;
        s.n_reads = nx.n_reads;
        s.n_skipped = nx.n_skipped;
        s.have_batch = true;
        s.device_only = false;
        s.wide_all = false;
        s.next.valid = false;
        if (s.n_reads) HIPCHK(ctx, hipStreamWaitEvent(s.stream, s.ev_copied, 0));
    }
    s.floor_len = floor_len;
    s.window = window;
    if (s.n_reads == 0) {
        s.ev_gate0 = s.ev_gate1 = s.ev_end = -1;
        memset(s.prof_counts, 0, sizeof s.prof_counts);
        s.state = 2;
        return 0;
    }
    if ((rc = plan_run(ctx, s))) return rc;
    if ((rc = enqueue_run(ctx, s))) {
        (void)hipStreamSynchronize(s.stream);
        s.state = 0;
        return rc;
    }
    s.state = 2;
    return 0;
}

int fadehip_annotate_submit(fadehip_ctx *ctx, int slot, const fadehip_read_batch *batch, int32_t floor_len, int32_t window) {
    int rc = fadehip_annotate_upload(ctx, slot, batch);
    if (rc) return rc;
    return fadehip_annotate_run(ctx, slot, floor_len, window);
}

int fadehip_annotate_results(fadehip_ctx *ctx, int slot, fadehip_anno_view *out) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!out) return set_err(ctx, FADEHIP_E_INVALID, "out is NULL");
    Slot &s = ctx->slots[slot];
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = finish_run(ctx, s, slot))) return rc;
    out->rs = s.n_reads ? s.res.p : nullptr;
    out->aln = s.n_aln ? (const fadehip_aln *)(s.res.p + s.res_aln_off) : nullptr;
    out->n_reads = s.n_reads;
    out->n_aln = s.n_aln;
    memcpy(out->stats, s.stats, sizeof out->stats);
    out->n_oversize = s.n_oversize;
    out->reserved = 0;
    return 0;
}

int fadehip_annotate_collect(fadehip_ctx *ctx, int slot, fadehip_anno_out *out) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!out) return set_err(ctx, FADEHIP_E_INVALID, "out is NULL");
    Slot &s = ctx->slots[slot];
    if (s.state < 2) return set_err(ctx, FADEHIP_E_STATE, "slot %d has not been run", slot);
    out->n_aln = 0;
    out->n_oversize = 0;
    out->reserved = 0;
    memset(out->stats, 0, sizeof out->stats);
    if (s.n_reads > 0 && !out->rs) return set_err(ctx, FADEHIP_E_INVALID, "out->rs is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if ((rc = finish_run(ctx, s, slot))) return rc;
    if (s.n_aln > 0 && (!out->aln || out->aln_cap < s.n_aln))
        return set_err(ctx, FADEHIP_E_INVALID, "out->aln holds %d entries, %d needed", out->aln ? out->aln_cap : 0, s.n_aln);
    if (s.n_reads) memcpy(out->rs, s.res.p, (size_t)s.n_reads);
    if (s.n_aln) memcpy(out->aln, s.res.p + s.res_aln_off, sizeof(fadehip_aln) * (size_t)s.n_aln);
    out->n_aln = s.n_aln;
    out->n_oversize = s.n_oversize;
    memcpy(out->stats, s.stats, sizeof out->stats);
    return 0;
}

int fadehip_sync(fadehip_ctx *ctx) {
    if (!ctx) return set_err(nullptr, FADEHIP_E_INVALID, "ctx is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->copy_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
    for (int k = 0; k < FADEHIP_NUM_SLOTS; k++)
        if (ctx->slots[k].stream) HIPCHK(ctx, hipStreamSynchronize(ctx->slots[k].stream));
    return 0;
}

int fadehip_last_run_profile(fadehip_ctx *ctx, int slot, float ms[4], int64_t counts[6]) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot &s = ctx->slots[slot];
    if (s.state != 3) return set_err(ctx, FADEHIP_E_STATE, "slot %d has no collected run", slot);
    if (!ms || !counts) return set_err(ctx, FADEHIP_E_INVALID, "NULL argument");
    ms[0] = ms[1] = ms[2] = ms[3] = 0.f;
    for (int k = 0; k < 6; k++) counts[k] = s.prof_counts[k];
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

// ------------------------------------------------------------------------------- BGZF compression
// the compressor's launches for n_bytes at d_src (device memory with 64 readable bytes behind the end) on the lane's stream
static int bgzf_lane_ready(fadehip_ctx *ctx, int lane, bool one_stream = false) {
    BgzfLane &l = ctx->bgzf[lane];
    if (!l.stream) {
        // (every stream is an HSA queue with a 173 MB context-save area to set up and to give back: FADEHIP_BGZF_ONE_STREAM=1
        // lets the lanes share one — their copies then no longer overlap each other's kernels; the file path's back half
        // uses the lanes one after the other anyway)
        if (lane > 0 && (one_stream || getenv("FADEHIP_BGZF_ONE_STREAM")) && ctx->bgzf[0].stream) l.stream = ctx->bgzf[0].stream;
        else if (ctx->split_cus > 0) l.stream = xcd_slice_stream(ctx, ctx->split_cus, ctx->cu_count / 8);
        else if (const char *kv = getenv("FADEHIP_BGZF_CUS")) l.stream = xcd_slice_stream(ctx, 0, std::max(1, std::min(atoi(kv), ctx->cu_count / 8)));  // the compressor alone on fewer CUs
        if (!l.stream) HIPCHK(ctx, hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
        HIPCHK(ctx, hipHostMalloc((void **)&l.h_total, 64));
        HIPCHK(ctx, hipEventCreateWithFlags(&l.done, hipEventDisableTiming | (ctx->blocking_sync ? hipEventBlockingSync : 0)));
    }
    if (!ctx->bgzf_ready) {
        HIPCHK(ctx, hipFuncSetAttribute((const void *)bgzf64::bgzf_deflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bgzf64::LDS_BYTES));
        HIPCHK(ctx, hipFuncSetAttribute((const void *)bgzf32::bgzf_deflate_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bgzf32::LDS_BYTES));
        ctx->bgzf_ready = true;
        if (const char *g = getenv("FADEHIP_BGZF_GEOM")) ctx->bgzf_geom_fixed = atoi(g) == 32 ? 32 : 64;  // (A/B runs; default: by the stream's ratio)
        if (getenv("FADEHIP_BGZF_PROF")) {
            int p64 = 0, p32 = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&p64, (const void *)bgzf64::bgzf_deflate_kernel, bgzf64::WG, bgzf64::LDS_BYTES);
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&p32, (const void *)bgzf32::bgzf_deflate_kernel, bgzf32::WG, bgzf32::LDS_BYTES);
            fprintf(stderr, "[fadehip bgzf] compressor workgroups per CU: %d (0xff00-byte blocks, %d B of LDS), %d (0x7f00-byte blocks, %d B)\n", p64, bgzf64::LDS_BYTES, p32, bgzf32::LDS_BYTES);
        }
    }
    if (l.state == 1) HIPCHK(ctx, hipEventSynchronize(l.done));  // never waited for: its buffers are still in use
    l.state = 0;
    return 0;
}
// Which geometry (bgzf_deflate.hpp): small blocks, two per CU, while the stream is mostly incompressible (packed bases,
// uniform qualities: ratio above 0.45, where the smaller blocks cost 0.5 %); htslib's block size where it compresses well
// (runs of qualities: a member shrinks to a few KB and a second header per 64 KB would show).  From the ratio of the ctx's
// previous call; the first call takes the large blocks.
static int bgzf_pick_geom(const fadehip_ctx *ctx) {
    if (ctx->bgzf_geom_fixed) return ctx->bgzf_geom_fixed;
    return ctx->bgzf_last_ratio > 0.45 ? 32 : 64;
}
// The same choice for bytes the host can look at (fadehip_bgzf_deflate_submit): the share of bytes equal to their
// predecessor over 64 windows of 4 KB.  Packed bases and uniform qualities: a few per cent; qualities in runs: a third.
static int bgzf_pick_geom_host(const fadehip_ctx *ctx, const uint8_t *p, size_t n) {
    if (ctx->bgzf_geom_fixed) return ctx->bgzf_geom_fixed;
    const size_t win = 4096, nwin = 64;
    size_t eq = 0, seen = 0;
    for (size_t w = 0; w < nwin; w++) {
        const size_t lo = n > win ? (n - win) / nwin * w : 0, hi = std::min(n, lo + win);
        for (size_t k = lo + 1; k < hi; k++) eq += p[k] == p[k - 1];
        seen += hi > lo ? hi - lo - 1 : 0;
        if (n <= win) break;
    }
    return seen && (double)eq < 0.15 * (double)seen ? 32 : 64;
}
// (host_out: pinned memory the members are packed into — the lane's own buffer when NULL.  A member is at most its block's
// bytes + 5 (stored) + 26 of BGZF framing: the buffer is sized for that, the kernel writes through PCIe, and nothing but
// the 8-byte total has to be copied afterwards.)
static size_t bgzf_out_cap(size_t n_bytes, int geom) {
    const size_t block = geom == 32 ? (size_t)bgzf32::BLOCK : (size_t)bgzf64::BLOCK;
    return n_bytes + ((n_bytes + block - 1) / block) * 32 + 64;
}
static int bgzf_enqueue(fadehip_ctx *ctx, int lane, const uint8_t *d_src, size_t n_bytes, int geom, PinBuf *host_out = nullptr) {
    BgzfLane &l = ctx->bgzf[lane];
    const size_t block = geom == 32 ? (size_t)bgzf32::BLOCK : (size_t)bgzf64::BLOCK;
    const uint32_t nb = (uint32_t)((n_bytes + block - 1) / block);
    int rc;
    PinBuf &ob = host_out ? *host_out : l.out;
    if ((rc = reserve(ctx, l.slots, (size_t)nb * bgzf64::SLOT)) || (rc = reserve(ctx, l.meta, (size_t)nb * 8 + 1024)) ||
        (rc = reserve(ctx, l.member_off, (size_t)nb * 8)) || (rc = reserve_pinned(ctx, ob, bgzf_out_cap(n_bytes, geom))))
        return rc;
    l.h_out = ob.p;
    uint32_t *d_size = (uint32_t *)l.meta.p, *d_crc = d_size + nb, *d_ticket = d_crc + nb;
    uint64_t *d_total = (uint64_t *)(((uintptr_t)(d_ticket + 2) + 7) & ~(uintptr_t)7);
    HIPCHK(ctx, hipMemsetAsync(d_ticket, 0, 8, l.stream));
    unsigned long long *prof = nullptr;
    if (getenv("FADEHIP_BGZF_PROF")) {  // shader clocks per phase, printed by wait (development aid)
        prof = (unsigned long long *)(d_total + 1);
        HIPCHK(ctx, hipMemsetAsync(prof, 0, 64 + 8 * 72, l.stream));
    }
    const unsigned cus = (unsigned)std::max(ctx->cu_count, 1);
    auto fill = [&](auto &a) {
        a.src = d_src;
        a.n_bytes = n_bytes;
        a.n_blocks = nb;
        a.slots = (uint8_t *)l.slots.p;
        a.out_size = d_size;
        a.out_crc = d_crc;
        a.ticket = d_ticket;
        a.prof = prof;
    };
    if (geom == 32) {
        bgzf32::DeflateArgs a;
        fill(a);
        hipLaunchKernelGGL(bgzf32::bgzf_deflate_kernel, dim3(std::min<unsigned>(nb, 2u * cus)), dim3(bgzf32::WG), bgzf32::LDS_BYTES, l.stream, a);
        HIPCHK(ctx, hipGetLastError());
        hipLaunchKernelGGL(bgzf32::bgzf_scan_kernel, dim3(1), dim3(1024), 0, l.stream, (const uint32_t *)d_size, nb, (uint64_t *)l.member_off.p, d_total);
        HIPCHK(ctx, hipGetLastError());
        hipLaunchKernelGGL(bgzf32::bgzf_pack_kernel, dim3(nb), dim3(256), 0, l.stream, (const uint8_t *)l.slots.p, (const uint32_t *)d_size,
                           (const uint32_t *)d_crc, (const uint64_t *)l.member_off.p, (uint64_t)n_bytes, nb, l.h_out);
    } else {
        bgzf64::DeflateArgs a;
        fill(a);
        hipLaunchKernelGGL(bgzf64::bgzf_deflate_kernel, dim3(std::min<unsigned>(nb, cus)), dim3(bgzf64::WG), bgzf64::LDS_BYTES, l.stream, a);
        HIPCHK(ctx, hipGetLastError());
        hipLaunchKernelGGL(bgzf64::bgzf_scan_kernel, dim3(1), dim3(1024), 0, l.stream, (const uint32_t *)d_size, nb, (uint64_t *)l.member_off.p, d_total);
        HIPCHK(ctx, hipGetLastError());
        hipLaunchKernelGGL(bgzf64::bgzf_pack_kernel, dim3(nb), dim3(256), 0, l.stream, (const uint8_t *)l.slots.p, (const uint32_t *)d_size,
                           (const uint32_t *)d_crc, (const uint64_t *)l.member_off.p, (uint64_t)n_bytes, nb, l.h_out);
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(l.h_total, d_total, 8, hipMemcpyDeviceToHost, l.stream));
    HIPCHK(ctx, hipEventRecord(l.done, l.stream));
    l.n_bytes = n_bytes;
    l.n_blocks = nb;
    l.geom = geom;
    l.state = 1;
    return 0;
}

int fadehip_bgzf_deflate_submit(fadehip_ctx *ctx, int lane, const void *src, size_t n_bytes) {
    if (!ctx) return set_err(nullptr, FADEHIP_E_INVALID, "ctx is NULL");
    if (lane < 0 || lane >= FADEHIP_BGZF_LANES) return set_err(ctx, FADEHIP_E_INVALID, "bgzf lane %d out of range", lane);
    if (!src || n_bytes == 0 || n_bytes > ((size_t)1 << 31)) return set_err(ctx, FADEHIP_E_INVALID, "bgzf: 1 .. 2^31 bytes per call");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = bgzf_lane_ready(ctx, lane))) return rc;
    BgzfLane &l = ctx->bgzf[lane];
    if ((rc = reserve(ctx, l.src, n_bytes + 64))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(l.src.p, src, n_bytes, hipMemcpyHostToDevice, l.stream));
    return bgzf_enqueue(ctx, lane, (const uint8_t *)l.src.p, n_bytes, bgzf_pick_geom_host(ctx, (const uint8_t *)src, n_bytes));
}

int fadehip_bgzf_deflate_wait(fadehip_ctx *ctx, int lane, const uint8_t **out, size_t *out_bytes) {
    if (!ctx) return set_err(nullptr, FADEHIP_E_INVALID, "ctx is NULL");
    if (lane < 0 || lane >= FADEHIP_BGZF_LANES) return set_err(ctx, FADEHIP_E_INVALID, "bgzf lane %d out of range", lane);
    if (!out || !out_bytes) return set_err(ctx, FADEHIP_E_INVALID, "NULL argument");
    BgzfLane &l = ctx->bgzf[lane];
    if (l.state != 1) return set_err(ctx, FADEHIP_E_STATE, "bgzf lane %d has nothing submitted", lane);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipEventSynchronize(l.done));
    const uint64_t total = *l.h_total;
    l.state = 0;
    if (getenv("FADEHIP_BGZF_PROF")) {
        unsigned long long pr[8 + 72];
        uint32_t *d_ticket = (uint32_t *)l.meta.p + 2 * (size_t)l.n_blocks;
        uint64_t *d_total = (uint64_t *)(((uintptr_t)(d_ticket + 2) + 7) & ~(uintptr_t)7);
        if (hipMemcpy(pr, d_total + 1, sizeof pr, hipMemcpyDeviceToHost) == hipSuccess) {
            if (pr[8]) {
                fprintf(stderr, "[fadehip bgzf] pipeline timeout: wait 0x%llx (saw %llu, wanted %llu, wave %llu) ticket %llu carry %llu n %llu | cand seq/free:", pr[8], pr[11] >> 32, pr[78], pr[79], pr[9], pr[10], pr[11] & 0xffffffffull);
                for (int k = 0; k < 4; k++) fprintf(stderr, " %llu/%llu", pr[12 + 2 * k], pr[13 + 2 * k]);
                fprintf(stderr, " | lens seq/free:");
                for (int k = 0; k < 24; k++) fprintf(stderr, " %llu/%llu", pr[20 + 2 * k], pr[21 + 2 * k]);
                fprintf(stderr, "\n");
            }
            static const char *nm[7] = {"load", "A match+parse", "B hist", "B codes", "C header+count", "D emit", "CRC"};
            unsigned long long sum = 0;
            for (int k = 0; k < 7; k++) sum += pr[k];
            fprintf(stderr, "[fadehip bgzf] %u blocks, shader clocks per block:", l.n_blocks);
            for (int k = 0; k < 7; k++) fprintf(stderr, " %s %.0f (%.0f%%)", nm[k], (double)pr[k] / l.n_blocks, 100.0 * (double)pr[k] / (double)std::max<unsigned long long>(sum, 1));
            fprintf(stderr, "\n");
            if (pr[41])
                fprintf(stderr, "[fadehip bgzf] B codes, clocks per block: ranks %.0f | merge (one lane) %.0f, depths %.0f, histogram + sums %.0f, leaves %.0f | limit %.0f | the others waited for %.0f | lengths, first codes, codes %.0f\n",
                        (double)pr[40] / l.n_blocks, (double)pr[41] / l.n_blocks, (double)pr[42] / l.n_blocks, (double)pr[43] / l.n_blocks, (double)pr[44] / l.n_blocks, (double)pr[45] / l.n_blocks, (double)pr[46] / l.n_blocks, (double)pr[47] / l.n_blocks);
            if (pr[48])
                fprintf(stderr, "[fadehip bgzf] C header + count, clocks per block: runs of lengths into tokens %.0f | thread 0's bit counts %.0f, then waited for the code-length code %.0f | tokens' bits into the header %.0f\n",
                        (double)pr[48] / l.n_blocks, (double)pr[49] / l.n_blocks, (double)pr[50] / l.n_blocks, (double)pr[51] / l.n_blocks);
            if (pr[56])
                fprintf(stderr, "[fadehip bgzf] A (the first wave's segment), clocks per block: position's bytes, bucket read and written %.0f | candidates' four bytes %.0f | extended %.0f | best picked, who yields, ballot %.0f | the piece's matches taken in turn %.0f | bitmaps, records stored %.0f\n",
                        (double)pr[56] / l.n_blocks, (double)pr[57] / l.n_blocks, (double)pr[58] / l.n_blocks, (double)pr[60] / l.n_blocks, (double)pr[61] / l.n_blocks, (double)pr[62] / l.n_blocks);
            if (pr[52])
                fprintf(stderr, "[fadehip bgzf] D emit, clocks per block: scan of the bit counts, the stream's words cleared %.0f | thread 0's tokens placed %.0f, then waited for the others %.0f | copied out %.0f\n",
                        (double)pr[52] / l.n_blocks, (double)pr[53] / l.n_blocks, (double)pr[54] / l.n_blocks, (double)pr[55] / l.n_blocks);
            fprintf(stderr, "[fadehip bgzf] phase A roles, clocks per block waited / in role: hasher %.0f / %.0f, extenders (sum) %.0f / %.0f, parser %.0f / %.0f\n",
                    (double)pr[60] / l.n_blocks, (double)pr[61] / l.n_blocks, (double)pr[62] / l.n_blocks, (double)pr[63] / l.n_blocks, (double)pr[64] / l.n_blocks, (double)pr[65] / l.n_blocks);
        }
    }
    if (total == 0 || total > (uint64_t)l.n_blocks * bgzf64::SLOT) {
        // a block whose pipeline timed out reports size ~0 and, as its CRC, the wait that gave up (role << 28 | piece)
        std::vector<uint32_t> meta(2 * (size_t)l.n_blocks);
        unsigned bad = 0, why = 0;
        if (hipMemcpy(meta.data(), l.meta.p, meta.size() * 4, hipMemcpyDeviceToHost) == hipSuccess)
            for (uint32_t k = 0; k < l.n_blocks; k++)
                if (meta[k] == 0xffffffffu) { if (!bad) why = meta[l.n_blocks + k]; bad++; }
        return set_err(ctx, FADEHIP_E_STATE, "internal: bgzf members add up to %llu bytes (%u blocks timed out, first wait 0x%08x)", (unsigned long long)total, bad, why);
    }
    ctx->bgzf_last_ratio = (double)total / (double)std::max<size_t>(l.n_bytes, 1);
    *out = l.h_out;  // (packed there by the kernel itself)
    *out_bytes = (size_t)total;
    return 0;
}

// ------------------------------------------------------------------------------- BGZF decompression
int fadehip_bgzf_inflate(fadehip_ctx *ctx, const void *members, size_t n_bytes, void *out, size_t out_cap, size_t *out_bytes) {
    if (!ctx) return set_err(nullptr, FADEHIP_E_INVALID, "ctx is NULL");
    if (!out_bytes || (n_bytes && !members)) return set_err(ctx, FADEHIP_E_INVALID, "NULL argument");
    *out_bytes = 0;
    if (n_bytes == 0) return 0;
    std::vector<bgzf::InflateBlock> blocks;
    size_t consumed = 0;
    uint64_t total = 0;
    std::string msg;
    if (!scan_bgzf_members((const uint8_t *)members, n_bytes, blocks, &consumed, &total, msg)) return set_err(ctx, FADEHIP_E_INVALID, "bgzf inflate: %s", msg.c_str());
    if (consumed != n_bytes) return set_err(ctx, FADEHIP_E_INVALID, "bgzf inflate: the last member is not whole (%zu of %zu bytes are whole members)", consumed, n_bytes);
    if (total > out_cap || (total && !out)) return set_err(ctx, FADEHIP_E_INVALID, "bgzf inflate: %llu bytes do not fit out_cap %zu", (unsigned long long)total, out_cap);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    InflateLane &l = ctx->inf;
    if (!l.stream) HIPCHK(ctx, hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
    const uint32_t nb = (uint32_t)blocks.size();
    int rc;
    if ((rc = reserve(ctx, l.comp, n_bytes + 16)) || (rc = reserve(ctx, l.blocks, sizeof(bgzf::InflateBlock) * (size_t)nb)) ||
        (rc = reserve(ctx, l.out, (size_t)total + 64)) || (rc = reserve(ctx, l.status, 4 * (size_t)nb)) || (rc = reserve(ctx, l.ticket, 64)) ||
        (rc = reserve_pinned(ctx, l.h_status, 4 * (size_t)nb + 8)))
        return rc;
    HIPCHK(ctx, hipMemcpyAsync(l.comp.p, members, n_bytes, hipMemcpyHostToDevice, l.stream));
    HIPCHK(ctx, hipMemcpyAsync(l.blocks.p, blocks.data(), sizeof(bgzf::InflateBlock) * (size_t)nb, hipMemcpyHostToDevice, l.stream));
    HIPCHK(ctx, hipStreamSynchronize(l.stream));  // (blocks is a pageable vector about to go out of scope)
    bgzf::InflateArgs a;
    a.comp = (const uint8_t *)l.comp.p;
    a.blocks = (const bgzf::InflateBlock *)l.blocks.p;
    a.n_blocks = nb;
    a.out = (uint8_t *)l.out.p;
    a.out_shift = nullptr;
    a.status = (uint32_t *)l.status.p;
    a.ticket = (uint32_t *)l.ticket.p;
    a.check_crc = 1;
    if ((rc = launch_inflate(ctx, l.stream, a))) return rc;
    HIPCHK(ctx, hipMemcpyAsync(l.h_status.p, l.ticket.p, 8, hipMemcpyDeviceToHost, l.stream));
    HIPCHK(ctx, hipStreamSynchronize(l.stream));
    const uint32_t n_bad = ((const uint32_t *)l.h_status.p)[1];
    if (n_bad) {
        HIPCHK(ctx, hipMemcpy(l.h_status.p, l.status.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
        const uint32_t *stt = (const uint32_t *)l.h_status.p;
        for (uint32_t k = 0; k < nb; k++)
            if (stt[k]) return set_err(ctx, FADEHIP_E_INVALID, "bgzf inflate: member %u of %u: %s (%u members failed)", k, nb, inflate_error_name(stt[k]), n_bad);
    }
    if (total) HIPCHK(ctx, hipMemcpy(out, l.out.p, (size_t)total, hipMemcpyDeviceToHost));
    *out_bytes = (size_t)total;
    return 0;
}

// ------------------------------------------------------------------------------- the file path on the device
struct fadehip_bam_stream {
    fadehip_ctx *ctx = nullptr;
    int32_t floor_len = 0, window = 0, n_ref = 0;
    uint32_t first_record = 0, tail_trim = 0;
    bool stored = false;  // uncompressed BGZF out
    bool no_output = false;  // FADEHIP_BAM_NO_OUTPUT: back gives the call's device bytes back without making BGZF of them
    DevBuf names_text, names_off;
    struct Out {
        DevBuf o;
        size_t bytes = 0;
        hipEvent_t ready = nullptr;
        int state = 0;  // 0 free, 1 its call's kernels are enqueued up to the tag sizes (to be finished), 2 finished: waiting for back
    } ring[FADEHIP_BAM_CHUNKS];
    // The front half, two calls in flight.  Call k lives in set k & 1 and on the ctx's slot k & 1 (a stream each):
    //   A  H2D (or H2D + inflate), carry-over of the previous call's cut-off record, framing, which records go to the gate
    //      and their sizes -> the host waits (buffers are sized from what the device found)
    //   B  packing, gate / score pass / pass 2, tag sizes                      -> enqueued, front returns
    //   C  the host reads the sizes (the one other wait), the rewrite kernel   -> "finishing" the call: done by back when it
    //      takes the call (or by front before the set is used again)
    // so that A of call k + 1 runs on the device beside B and C of call k, and no stream idles while the host waits for
    // another.  Order between calls: A(k + 1) needs where call k's last whole record ended (known once front(k) has waited for
    // its A); everything else of two calls is independent.
    struct Set {
        DevBuf comp, blocks, status, ticket;  // members to inflate on the device
        DevBuf u;                             // the call's inflated bytes, the previous call's cut-off record in front
        DevBuf seg, slots, rec_off, info, sent_of, art_of, out_size, blk32, blk64, counts;
        PinBuf h_blocks, h_counts;
        uint64_t k = ~0ull;
        uint32_t n_rec = 0, n_sent = 0, ntb = 0;
        bool pending = false;  // B is enqueued, C is not
        bam::TagArgs ta;
        Out *out = nullptr;
        std::mutex mu;  // finishing the set's call (front and back may both come to do it; the OTHER set's call is not held up)
    } set[2];
    uint32_t prev_len = 0, prev_consumed = 0;  // of the previous call's u
    double rec_bytes_avg = 0;                  // bytes per record of the previous call (sizes the next call's record-parallel launch)
    uint64_t k_front = 0, k_back = 0;
    uint64_t k_sub = 0;                     // calls handed to the compressor (back may run one ahead of the call it returns)
    PinBuf outbuf[FADEHIP_BAM_CHUNKS];      // the members of call k, packed by the kernel itself: pinned, k % FADEHIP_BAM_CHUNKS
    std::mutex mu;
    std::condition_variable cv;
    int64_t stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t n_records = 0, n_oversize = 0, n_redone = 0;
    bool failed = false, ended = false, closing = false;
    double t_inflate = 0, t_frame = 0, t_run = 0, t_tags = 0;
};

namespace {

int bam_fail(fadehip_bam_stream *st, int rc) {
    {
        std::lock_guard<std::mutex> l(st->mu);
        st->failed = true;
    }
    st->cv.notify_all();
    return rc;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// FADEHIP_BAM_TRACE=1: where the first calls of a stream spend their time (buffers, streams and staging memory are made in them)
struct CallTrace {
    bool on;
    const char *who;
    uint64_t k;
    double t;
    CallTrace(const char *w, uint64_t kk) : on(kk < 4 && getenv("FADEHIP_BAM_TRACE")), who(w), k(kk), t(on ? now_s() : 0) {}
    void mark(const char *what) {
        if (!on) return;
        const double n = now_s();
        fprintf(stderr, "[fadehip trace] %s %llu: %-28s %8.3f ms\n", who, (unsigned long long)k, what, (n - t) * 1e3);
        t = n;
    }
};

// C of call k (see fadehip_bam_stream): waits for the call's run and tag sizes, sizes the output, enqueues the rewrite.
// Idempotent; front and back may both arrive here for the same call.
int bam_finish_call(fadehip_bam_stream *st, uint64_t k) {
    fadehip_ctx *ctx = st->ctx;
    fadehip_bam_stream::Set &S = st->set[k & 1];
    std::lock_guard<std::mutex> pl(S.mu);
    if (S.k != k || !S.pending) return 0;
    Slot &s = ctx->slots[k & 1];
    hipStream_t q = s.stream;
    fadehip_bam_stream::Out *out = S.out;
    const double t0 = now_s();
    int rc;
    out->bytes = 0;
    if (S.n_rec) {
        bam::ChunkCounts *h_counts = (bam::ChunkCounts *)S.h_counts.p;
        if (S.n_sent) {
            if ((rc = finish_run(ctx, s, (int)(k & 1)))) return rc;  // waits for the stream: run and sizes
            std::lock_guard<std::mutex> l(st->mu);
            for (int t = 0; t < 8; t++) st->stats[t] += s.stats[t];
            st->n_oversize += s.n_oversize;
        } else {
            HIPCHK(ctx, hipStreamSynchronize(q));
            std::lock_guard<std::mutex> l(st->mu);
            st->stats[0] += S.n_rec;
        }
        const uint64_t out_bytes = h_counts->out_bytes;
        if (out_bytes > ((uint64_t)1 << 31)) return set_err(ctx, FADEHIP_E_UNSUPPORTED, "bam stream: %llu output bytes in one call (at most 2^31)", (unsigned long long)out_bytes);
        if ((rc = reserve_roomy(ctx, out->o, (size_t)out_bytes + 256))) return rc;
        S.ta.o = (uint8_t *)out->o.p;
        hipLaunchKernelGGL(bam::bam_rewrite_kernel, dim3(S.ntb), dim3(bam::REWRITE_WAVES * 64), 0, q, S.ta);
        HIPCHK(ctx, hipGetLastError());
        out->bytes = (size_t)out_bytes;
        std::lock_guard<std::mutex> l(st->mu);
        st->n_records += S.n_rec;
    }
    if (!out->ready) HIPCHK(ctx, hipEventCreateWithFlags(&out->ready, hipEventDisableTiming | (ctx->blocking_sync ? hipEventBlockingSync : 0)));
    HIPCHK(ctx, hipEventRecord(out->ready, q));
    S.pending = false;
    st->t_tags += now_s() - t0;
    {
        std::lock_guard<std::mutex> l(st->mu);
        out->state = 2;
    }
    st->cv.notify_all();
    return 0;
}

int bam_front_impl(fadehip_bam_stream *st, const uint8_t *members, size_t n_bytes, int last, bool raw) {
    fadehip_ctx *ctx = st->ctx;
    const uint64_t k = st->k_front;
    fadehip_bam_stream::Set &S = st->set[k & 1];
    Slot &s = ctx->slots[k & 1];
    int rc;
    // the set's previous call must have been finished (back has usually done that long ago)
    CallTrace tr("front", k);
    if (ctx->n_contigs == 0) return set_err(ctx, FADEHIP_E_STATE, "fadehip_genome_upload has not been called");
    if (k >= 2 && (rc = bam_finish_call(st, k - 2))) return rc;
    tr.mark("finish call k - 2");
    // every stream is an HSA queue to set up and to give back (tens of ms each): slot 0 works on the ctx's copy stream, which
    // exists anyway and which the file path does not use otherwise; slot 1 gets a stream of its own when the second call comes
    if (ctx->split_cus > 0 && !s.stream && !s.h_zb) {
        ctx->tail_cus_per_xcd = 0;  // (no third set of CUs)
        s.stream = xcd_slice_stream(ctx, 0, ctx->split_cus);
    }
    if (!s.stream && !s.h_zb && (k & 1) == 0 && ctx->copy_stream) s.stream = ctx->copy_stream;
    if ((rc = ensure_slot(ctx, s))) return rc;
    tr.mark("slot (stream, events)");
    hipStream_t q = s.stream;
    const double t0 = now_s();
    // ---- the members, and where their payloads go
    std::vector<bgzf::InflateBlock> blocks;
    size_t consumed = 0;
    uint64_t total = 0;
    std::string msg;
    if (raw) {
        total = n_bytes;  // the caller has inflated the members: these are their payloads
    } else {
        if (n_bytes && !scan_bgzf_members(members, n_bytes, blocks, &consumed, &total, msg)) return set_err(ctx, FADEHIP_E_INVALID, "bam stream: %s", msg.c_str());
        if (consumed != n_bytes) return set_err(ctx, FADEHIP_E_INVALID, "bam stream: front takes whole BGZF members (%zu of %zu bytes are)", consumed, n_bytes);
    }
    const uint32_t carry = st->prev_len - st->prev_consumed;
    if ((uint64_t)carry + total > (uint64_t)bam::MAX_U) return set_err(ctx, FADEHIP_E_UNSUPPORTED, "bam stream: %llu inflated bytes in one call (at most %u)", (unsigned long long)total + carry, bam::MAX_U);
    uint32_t u_len = carry + (uint32_t)total;
    if (last && !raw && st->tail_trim) {  // (the bytes behind this stream's last record, in its last member, belong to another reader)
        if ((uint64_t)st->tail_trim > total) return set_err(ctx, FADEHIP_E_INVALID, "bam stream: tail_trim %u exceeds the last call's %llu bytes", st->tail_trim, (unsigned long long)total);
        u_len -= st->tail_trim;
    }
    // (sized from the UNTRIMMED length: the inflate kernel writes the last member's whole ISIZE, tail_trim only shortens
    // what the framing looks at)
    if ((rc = reserve_roomy(ctx, S.u, (size_t)carry + (size_t)total + 256))) return rc;
    uint8_t *u = (uint8_t *)S.u.p;
    // the cut-off record of the previous call: its bytes are final (front waited for that call's framing), and the rewrite
    // that also reads them does not change them
    if (carry) HIPCHK(ctx, hipMemcpyAsync(u, (const uint8_t *)st->set[(k + 1) & 1].u.p + st->prev_consumed, carry, hipMemcpyDeviceToDevice, q));
    const uint32_t nb = (uint32_t)blocks.size();
    if (raw && n_bytes) HIPCHK(ctx, hipMemcpyAsync(u + carry, members, n_bytes, hipMemcpyHostToDevice, q));
    const bool fine = tr.on && k == 0 && getenv("FADEHIP_BAM_TRACE_FINE");
    if (fine) { tr.mark("  copy enqueued"); (void)hipStreamSynchronize(q); tr.mark("  copy waited for"); }
    if (nb) {
        for (auto &b : blocks) b.dst_off += carry;
        if ((rc = reserve_roomy(ctx, S.comp, n_bytes + 16)) || (rc = reserve_roomy(ctx, S.blocks, sizeof(bgzf::InflateBlock) * (size_t)nb)) ||
            (rc = reserve_roomy(ctx, S.status, 4 * (size_t)nb)) || (rc = reserve_roomy(ctx, S.ticket, 64)) ||
            (rc = reserve_pinned(ctx, S.h_blocks, sizeof(bgzf::InflateBlock) * (size_t)nb)))
            return rc;
        memcpy(S.h_blocks.p, blocks.data(), sizeof(bgzf::InflateBlock) * (size_t)nb);
        HIPCHK(ctx, hipMemcpyAsync(S.comp.p, members, n_bytes, hipMemcpyHostToDevice, q));
        HIPCHK(ctx, hipMemcpyAsync(S.blocks.p, S.h_blocks.p, sizeof(bgzf::InflateBlock) * (size_t)nb, hipMemcpyHostToDevice, q));
        bgzf::InflateArgs ia;
        ia.comp = (const uint8_t *)S.comp.p;
        ia.blocks = (const bgzf::InflateBlock *)S.blocks.p;
        ia.n_blocks = nb;
        ia.out = u;
        ia.out_shift = nullptr;
        ia.status = (uint32_t *)S.status.p;
        ia.ticket = (uint32_t *)S.ticket.p;
        ia.check_crc = 1;
        if ((rc = launch_inflate(ctx, q, ia))) return rc;
    }
    // ---- framing
    const uint32_t n_seg = (u_len + bam::SEG - 1) / bam::SEG;
    const uint32_t rec_cap = u_len / 36u + 2u;
    if ((rc = reserve_roomy(ctx, S.seg, 16 * (size_t)std::max(n_seg, 1u))) || (rc = reserve_roomy(ctx, S.slots, 4 * (size_t)bam::SEG_SLOTS * std::max(n_seg, 1u))) ||
        (rc = reserve_roomy(ctx, S.rec_off, 4 * (size_t)rec_cap)) || (rc = reserve_roomy(ctx, S.counts, sizeof(bam::ChunkCounts))) ||
        (rc = reserve_pinned(ctx, S.h_counts, sizeof(bam::ChunkCounts) + 16)))
        return rc;
    bam::ChunkCounts *d_counts = (bam::ChunkCounts *)S.counts.p;
    bam::ChunkCounts *h_counts = (bam::ChunkCounts *)S.h_counts.p;
    HIPCHK(ctx, hipMemsetAsync(d_counts, 0, sizeof(bam::ChunkCounts), q));
    HIPCHK(ctx, hipMemsetAsync(&d_counts->l_seq_min, 0xff, 4, q));
    bam::FrameArgs fa;
    fa.u = u;
    fa.u_len = u_len;
    fa.first = k == 0 ? st->first_record : 0u;
    fa.n_ref = st->n_ref;
    fa.n_seg_cap = n_seg;
    fa.cand = (uint32_t *)S.seg.p;
    fa.exit_ = fa.cand + std::max(n_seg, 1u);
    fa.cnt = fa.exit_ + std::max(n_seg, 1u);
    fa.base = fa.cnt + std::max(n_seg, 1u);
    fa.slots = (uint32_t *)S.slots.p;
    fa.rec_off = (uint32_t *)S.rec_off.p;
    fa.rec_cap = rec_cap;
    fa.counts = d_counts;
    if (n_seg) {
        hipLaunchKernelGGL(bam::bam_frame_walk_kernel, dim3((n_seg + bam::WALK_SEGS - 1) / bam::WALK_SEGS), dim3(64), 0, q, fa);
        HIPCHK(ctx, hipGetLastError());
    }
    if (fine) { tr.mark("  memsets, walk enqueued"); (void)hipStreamSynchronize(q); tr.mark("  waited for"); }
    hipLaunchKernelGGL(bam::bam_frame_resolve_kernel, dim3(1), dim3(64), 0, q, fa);
    HIPCHK(ctx, hipGetLastError());
    if (fine) { tr.mark("  resolve enqueued"); (void)hipStreamSynchronize(q); tr.mark("  waited for"); }
    hipLaunchKernelGGL(bam::bam_frame_compact_kernel, dim3(std::max(1u, (n_seg + 3) / 4)), dim3(256), 0, q, fa);
    HIPCHK(ctx, hipGetLastError());
    // which records go to the device's gate, their sizes: enqueued behind the framing for as many records as the bytes could
    // hold at most (threads beyond the records that are there return at once), so that ONE wait brings back both counts
    const uint32_t nblk_cap = (rec_cap + bam::PACK_BLOCK - 1) / bam::PACK_BLOCK;
    if ((rc = reserve_roomy(ctx, S.info, 4 * (size_t)rec_cap)) || (rc = reserve_roomy(ctx, S.sent_of, 4 * (size_t)rec_cap)) ||
        (rc = reserve_roomy(ctx, S.out_size, 4 * (size_t)rec_cap)) || (rc = reserve_roomy(ctx, S.blk32, 24 * (size_t)nblk_cap)))
        return rc;
    bam::PackArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.u = u;
    pa.rec_off = fa.rec_off;
    pa.counts_in = d_counts;
    pa.r0 = 0;
    pa.r1_cap = rec_cap;
    pa.info = (uint32_t *)S.info.p;
    pa.blk_sums = (uint32_t *)S.blk32.p;
    pa.blk_base = pa.blk_sums + 3 * (size_t)nblk_cap;
    pa.counts = d_counts;
    pa.sent_of = (int32_t *)S.sent_of.p;
    // (sized from the bytes per record of the calls so far, with a margin; the kernel strides over what is really there)
    const uint32_t nblk_est = st->rec_bytes_avg > 0 ? (uint32_t)((double)u_len / st->rec_bytes_avg * 1.25 / bam::PACK_BLOCK) + 8u : nblk_cap;
    hipLaunchKernelGGL(bam::bam_pack_count_kernel, dim3(std::max(1u, std::min(nblk_cap, nblk_est))), dim3(bam::PACK_BLOCK), 0, q, pa);
    HIPCHK(ctx, hipGetLastError());
    if (fine) { tr.mark("  compact, pack count enqueued"); (void)hipStreamSynchronize(q); tr.mark("  waited for"); }
    hipLaunchKernelGGL(bam::bam_pack_scan_kernel, dim3(1), dim3(1024), 0, q, pa, nblk_cap);
    HIPCHK(ctx, hipGetLastError());
    if (fine) { tr.mark("  pack scan enqueued"); (void)hipStreamSynchronize(q); tr.mark("  waited for"); }
    HIPCHK(ctx, hipMemcpyAsync(h_counts, d_counts, sizeof(bam::ChunkCounts), hipMemcpyDeviceToHost, q));
    uint32_t *h_tick = (uint32_t *)(S.h_counts.p + sizeof(bam::ChunkCounts));
    h_tick[0] = h_tick[1] = 0;
    if (nb) HIPCHK(ctx, hipMemcpyAsync(h_tick, S.ticket.p, 8, hipMemcpyDeviceToHost, q));
    const double t1 = now_s();
    tr.mark("A enqueued");
    HIPCHK(ctx, hipStreamSynchronize(q));  // (1) the records of this call and the sizes of their batch
    const double t2 = now_s();
    tr.mark("A waited for");
    st->t_inflate += t1 - t0;
    st->t_frame += t2 - t1;
    if (nb && h_tick[1]) {
        std::vector<uint32_t> stt(nb);
        HIPCHK(ctx, hipMemcpy(stt.data(), S.status.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
        for (uint32_t b = 0; b < nb; b++)
            if (stt[b]) return set_err(ctx, FADEHIP_E_INVALID, "bam stream: call %llu, member %u of %u: %s (%u members failed)", (unsigned long long)k, b, nb, inflate_error_name(stt[b]), h_tick[1]);
    }
    if (h_counts->frame_err)
        return set_err(ctx, FADEHIP_E_INVALID, "bam stream: call %llu: %s at inflated offset %u", (unsigned long long)k,
                       h_counts->frame_err == 1 ? "a record's block_size is impossible" : "the first record lies beyond the bytes given", h_counts->frame_err_at);
    const uint32_t n_rec = h_counts->n_records, used = h_counts->consumed;
    if (last && used != u_len) return set_err(ctx, FADEHIP_E_INVALID, "bam stream: the input ends inside a record (%u bytes behind the last whole one)", u_len - used);
    st->prev_len = u_len;
    st->prev_consumed = used;
    st->n_redone += h_counts->n_redone;
    if (n_rec) st->rec_bytes_avg = (double)used / (double)n_rec;
    // ---- a place in the ring
    fadehip_bam_stream::Out *out;
    {
        std::unique_lock<std::mutex> l(st->mu);
        out = &st->ring[k % FADEHIP_BAM_CHUNKS];
        st->cv.wait(l, [&] { return out->state == 0 || st->failed || st->closing; });
        if (st->failed || st->closing) return set_err(ctx, FADEHIP_E_STATE, "bam stream: stopped");
    }
    tr.mark("place in the ring");
    out->bytes = 0;
    S.k = k;
    S.n_rec = n_rec;
    S.n_sent = 0;
    S.ntb = 0;
    S.out = out;
    if (n_rec) {
        const uint32_t nblk = (n_rec + bam::PACK_BLOCK - 1) / bam::PACK_BLOCK, ntb = (n_rec + bam::TAG_BLOCK - 1) / bam::TAG_BLOCK;
        if ((rc = reserve_roomy(ctx, S.blk64, 16 * (size_t)ntb))) return rc;
        if (h_counts->n_bad_layout)
            return set_err(ctx, FADEHIP_E_INVALID, "bam stream: call %llu: %u records whose fields do not fit their block_size or whose tags are not whole fields (corrupt BAM)", (unsigned long long)k, h_counts->n_bad_layout);
        const uint32_t n_sent = h_counts->n_sent;
        if ((uint64_t)h_counts->n_seq * 2 >= ((uint64_t)1 << 32)) return set_err(ctx, FADEHIP_E_UNSUPPORTED, "bam stream: packed sequence bytes per call must stay below 2^31");
        if (s.state == 2) HIPCHK(ctx, hipStreamSynchronize(s.stream));
        s.state = 0;
        s.device_only = true;
        s.next.valid = false;
        s.have_batch = false;
        s.cur = 0;
        s.L = batch_layout(n_sent, h_counts->n_cig, h_counts->n_seq);
        if ((rc = reserve_roomy(ctx, s.in[0], s.L.total))) return rc;
        uint8_t *ib = (uint8_t *)s.in[0].p;
        pa.tid = (int32_t *)(ib + s.L.off[A_TID]);
        pa.pos = (int32_t *)(ib + s.L.off[A_POS]);
        pa.lseq = (int32_t *)(ib + s.L.off[A_LSEQ]);
        pa.cigar_off = (uint32_t *)(ib + s.L.off[A_CIGOFF]);
        pa.seq_off = (uint32_t *)(ib + s.L.off[A_SEQOFF]);
        pa.flag = (uint16_t *)(ib + s.L.off[A_FLAG]);
        pa.has_sa = ib + s.L.off[A_SA];
        pa.cigar_ops = (uint32_t *)(ib + s.L.off[A_CIG]);
        pa.seq = ib + s.L.off[A_SEQ];
        hipLaunchKernelGGL(bam::bam_pack_write_kernel, dim3(nblk), dim3(bam::PACK_BLOCK), 0, q, pa);
        HIPCHK(ctx, hipGetLastError());
        tr.mark("pack enqueued");
        // ---- annotateTask on the device (level 2's kernels), results left there
        s.n_reads = (int)n_sent;
        s.n_skipped = (int)(n_rec - n_sent);
        s.floor_len = st->floor_len;
        s.window = st->window;
        memset(s.hist, 0, sizeof s.hist);
        s.wide.clear();
        s.wide_all = false;
        s.out_bound = n_sent;
        s.max_lq = 0;
        s.span_bound = 1;
        if (n_sent) {
            const int lmin = (int)std::min<uint32_t>(h_counts->l_seq_min, (uint32_t)MAX_LONG_QUERY), lmax = (int)std::min<uint32_t>(h_counts->l_seq_max, (uint32_t)MAX_LONG_QUERY);
            const int c_lo = list_of_len(std::max(lmin, 1)), c_hi = list_of_len(std::max(lmax, 1));
            for (int c = c_lo; c <= c_hi; c++) s.hist[c] = n_sent;  // any of them may be of any length in between
            if (h_counts->n_long_q) s.hist[LONG_LIST] = h_counts->n_long_q;  // (the exact number of reads beyond 512 bases)
            s.max_lq = std::max(lmax, 1);
            s.span_bound = std::max<int64_t>(h_counts->span_max, 1);
            s.wide_all = s.span_bound > WIDE_MIN_SPAN;
        }
        if (n_sent) {
            if ((rc = plan_run(ctx, s)) || (rc = enqueue_run(ctx, s))) {
                (void)hipStreamSynchronize(q);
                return rc;
            }
            s.state = 2;
        }
        tr.mark("run planned and enqueued");
        // ---- what anno.d:94-107 adds: sizes, offsets
        if ((rc = reserve_roomy(ctx, S.art_of, 4 * (size_t)std::max(n_sent, 1u)))) return rc;
        if (n_sent) {
            HIPCHK(ctx, hipMemsetAsync(S.art_of.p, 0xff, 4 * (size_t)n_sent, q));
            if (s.out_cap) {
                hipLaunchKernelGGL(bam::bam_art_index_kernel, dim3((s.out_cap + 255) / 256), dim3(256), 0, q, (const fadehip_aln *)s.aln.p,
                                   (const uint32_t *)(s.d_counters() + 2 * NUM_LISTS + 3), s.out_cap, (int32_t *)S.art_of.p, n_sent);
                HIPCHK(ctx, hipGetLastError());
            }
        }
        bam::TagArgs &ta = S.ta;
        memset(&ta, 0, sizeof ta);
        ta.u = u;
        ta.rec_off = fa.rec_off;
        ta.counts_in = d_counts;
        ta.r0 = 0;
        ta.r1_cap = n_rec;
        ta.info = pa.info;
        ta.sent_of = pa.sent_of;
        ta.rs = (const uint8_t *)s.rs.p;
        ta.aln = (const fadehip_aln *)s.aln.p;
        ta.art_of = (const int32_t *)S.art_of.p;
        ta.names.text = (const char *)st->names_text.p;
        ta.names.off = (const uint32_t *)st->names_off.p;
        ta.names.n = st->n_ref;
        ta.out_size = (uint32_t *)S.out_size.p;
        ta.blk_sums = (uint64_t *)S.blk64.p;
        ta.blk_base = ta.blk_sums + ntb;
        ta.counts = d_counts;
        ta.out_base = 0;
        hipLaunchKernelGGL(bam::bam_tag_size_kernel, dim3(ntb), dim3(bam::TAG_BLOCK), 0, q, ta);
        HIPCHK(ctx, hipGetLastError());
        hipLaunchKernelGGL(bam::bam_tag_scan_kernel, dim3(1), dim3(1024), 0, q, ta, ntb);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(h_counts, d_counts, sizeof(bam::ChunkCounts), hipMemcpyDeviceToHost, q));
        S.n_sent = n_sent;
        S.ntb = ntb;
    }
    S.pending = true;
    tr.mark("tag sizes enqueued");
    st->t_run += now_s() - t2;
    // the inflated bytes of this call are read by the next call's carry copy and by this call's rewrite, which change nothing;
    // the set's buffers are written again by call k + 2, whose front finishes this call first.
    {
        std::lock_guard<std::mutex> l(st->mu);
        out->state = 1;
        st->k_front = k + 1;
        if (last) st->ended = true;
    }
    st->cv.notify_all();
    return 0;
}

}  // namespace

int fadehip_bam_open(fadehip_ctx *ctx, const fadehip_bam_config *cfg, fadehip_bam_stream **out) {
    if (!ctx) return set_err(nullptr, FADEHIP_E_INVALID, "ctx is NULL");
    if (!cfg || !out || cfg->n_ref < 0 || (cfg->n_ref && !cfg->ref_names) || cfg->window < 0 || (cfg->flags & ~(FADEHIP_BAM_STORED | FADEHIP_BAM_NO_OUTPUT)))
        return set_err(ctx, FADEHIP_E_INVALID, "bam stream: bad configuration");
    if (!ctx->two_pass) return set_err(ctx, FADEHIP_E_UNSUPPORTED, "bam stream: needs the default kernels (FADEHIP_KERNEL unset)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    fadehip_bam_stream *st = new (std::nothrow) fadehip_bam_stream;
    if (!st) return set_err(ctx, FADEHIP_E_NOMEM, "out of memory");
    st->ctx = ctx;
    st->floor_len = cfg->floor_len;
    st->window = cfg->window;
    st->n_ref = cfg->n_ref;
    st->first_record = cfg->first_record;
    st->tail_trim = cfg->tail_trim;
    st->stored = (cfg->flags & FADEHIP_BAM_STORED) != 0;
    st->no_output = (cfg->flags & FADEHIP_BAM_NO_OUTPUT) != 0;
    std::string text;
    std::vector<uint32_t> off((size_t)cfg->n_ref + 1, 0);
    for (int k = 0; k < cfg->n_ref; k++) {
        if (!cfg->ref_names[k]) { delete st; return set_err(ctx, FADEHIP_E_INVALID, "bam stream: ref_names[%d] is NULL", k); }
        off[(size_t)k] = (uint32_t)text.size();
        text += cfg->ref_names[k];
    }
    off[(size_t)cfg->n_ref] = (uint32_t)text.size();
    int rc;
    if ((rc = reserve(ctx, st->names_text, text.size() + 1)) || (rc = reserve(ctx, st->names_off, 4 * off.size()))) { fadehip_bam_close(st); return rc; }
    // (on the ctx's copy stream: a synchronous hipMemcpy would bring up the null stream, one more queue to set up and to give back)
    if (hipMemcpyAsync(st->names_text.p, text.data(), text.size(), hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess ||
        hipMemcpyAsync(st->names_off.p, off.data(), 4 * off.size(), hipMemcpyHostToDevice, ctx->copy_stream) != hipSuccess ||
        hipStreamSynchronize(ctx->copy_stream) != hipSuccess) {
        fadehip_bam_close(st);
        return set_err(ctx, FADEHIP_E_HIP, "bam stream: copying the contig names failed");
    }
    *out = st;
    return 0;
}

// What the first calls of a stream would otherwise make one after the other, each in its turn holding up the thread that
// came to it — the second slot's stream and the compressor lanes' streams (HSA queues: 7-10 ms each, and a launch on
// another thread waits meanwhile), the three staging buffers of the members, the inflated bytes' buffers — made here side by
// side; then one copy up, one down and one wait per stream (the first of each costs milliseconds).  Optional, and meant
// for a thread of its own beside the caller's own start-up (reading the FASTA, the input's first members).
int fadehip_bam_prepare(fadehip_bam_stream *st, size_t call_bytes) {
    if (!st) return set_err(nullptr, FADEHIP_E_INVALID, "stream is NULL");
    fadehip_ctx *ctx = st->ctx;
    if (call_bytes == 0 || call_bytes > (size_t)bam::MAX_U) return set_err(ctx, FADEHIP_E_INVALID, "bam stream: prepare takes the inflated bytes of a call (1 .. %u)", bam::MAX_U);
    if (st->k_front) return set_err(ctx, FADEHIP_E_STATE, "bam stream: prepare comes before the first front call");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const double t0 = now_s();
    static const bool one_stream = getenv("FADEHIP_BAM_BACK_STREAMS") && atoi(getenv("FADEHIP_BAM_BACK_STREAMS")) == 1;
    int rcs[4] = {0, 0, 0, 0};
    std::string errs[4];
    const bool with_out = !st->no_output;
    // lane 0 first on this thread when the lanes share a stream (lane 1 borrows it); the function attributes are set once, there
    if (with_out && (rcs[0] = bgzf_lane_ready(ctx, 0, one_stream))) return rcs[0];
    const double t_lane0 = now_s();
    std::vector<std::thread> th;
    auto side = [&](int slot_no, auto fn) {
        th.emplace_back([&, slot_no, fn] {
            if (hipSetDevice(ctx->device) != hipSuccess) { rcs[slot_no] = FADEHIP_E_HIP; errs[slot_no] = "hipSetDevice failed"; return; }
            if ((rcs[slot_no] = fn())) errs[slot_no] = fadehip_last_error(ctx);
        });
    };
    side(1, [&]() -> int {  // the second slot: a stream of its own
        Slot &s = ctx->slots[1];
        if (ctx->split_cus > 0 && !s.stream && !s.h_zb) s.stream = xcd_slice_stream(ctx, 0, ctx->split_cus);
        return ensure_slot(ctx, s);
    });
    if (with_out) side(2, [&]() -> int { return bgzf_lane_ready(ctx, 1, one_stream); });
    // this thread: the first slot (on the ctx's copy stream), the buffers
    int rc = 0;
    {
        Slot &s = ctx->slots[0];
        if (ctx->split_cus > 0 && !s.stream && !s.h_zb) {
            ctx->tail_cus_per_xcd = 0;
            s.stream = xcd_slice_stream(ctx, 0, ctx->split_cus);
        }
        if (!s.stream && !s.h_zb && ctx->copy_stream) s.stream = ctx->copy_stream;
        rc = ensure_slot(ctx, s);
    }
    const size_t carry_room = 65536;
    for (int q = 0; q < 2 && !rc; q++) {
        fadehip_bam_stream::Set &S = st->set[q];
        if (!(rc = reserve_roomy(ctx, S.u, call_bytes + carry_room + 256))) rc = reserve_pinned(ctx, S.h_counts, sizeof(bam::ChunkCounts) + 16);
        if (!rc) rc = reserve_roomy(ctx, S.counts, sizeof(bam::ChunkCounts));
    }
    // (annotated records are a few per cent longer than the call's; the members' bound is the compressor's own)
    const size_t out_est = call_bytes + call_bytes / 8;
    for (int q = 0; q < FADEHIP_BAM_CHUNKS && !rc && with_out; q++)
        rc = reserve_pinned(ctx, st->outbuf[q], st->stored ? (out_est / bgzf::STORE_BLOCK + 2) * bgzf::STORE_MEMBER : bgzf_out_cap(out_est, 32));
    const double t_mine = now_s();
    for (auto &t : th) t.join();
    if (getenv("FADEHIP_BAM_TRACE")) fprintf(stderr, "[fadehip trace] prepare: lane 0 %.3f ms, own part (slot 0, buffers) %.3f ms, the side threads %.3f ms more\n", (t_lane0 - t0) * 1e3, (t_mine - t_lane0) * 1e3, (now_s() - t_mine) * 1e3);
    if (rc) return rc;
    for (int q = 1; q < 4; q++)
        if (rcs[q]) return set_err(ctx, rcs[q], "bam stream: prepare: %s", errs[q].c_str());
    // the first copy, the first wait of every stream
    const double t1 = now_s();
    if (with_out && st->outbuf[0].p) {
        for (int q = 0; q < 2; q++) {
            fadehip_bam_stream::Set &S = st->set[q];
            hipStream_t sq = ctx->slots[q].stream;
            const size_t n = std::min(call_bytes, st->outbuf[q].cap);
            HIPCHK(ctx, hipMemcpyAsync(S.u.p, st->outbuf[q].p, n, hipMemcpyHostToDevice, sq));
            HIPCHK(ctx, hipMemsetAsync(S.counts.p, 0, sizeof(bam::ChunkCounts), sq));
            HIPCHK(ctx, hipMemcpyAsync(S.h_counts.p, S.counts.p, sizeof(bam::ChunkCounts), hipMemcpyDeviceToHost, sq));
        }
        for (int q = 0; q < 2; q++) HIPCHK(ctx, hipStreamSynchronize(ctx->slots[q].stream));
        for (int q = 0; q < 2; q++) {
            BgzfLane &l = ctx->bgzf[q];
            HIPCHK(ctx, hipMemcpyAsync(l.h_total, st->set[0].counts.p, 8, hipMemcpyDeviceToHost, l.stream));
            HIPCHK(ctx, hipEventRecord(l.done, l.stream));
            HIPCHK(ctx, hipEventSynchronize(l.done));
        }
    }
    if (getenv("FADEHIP_BAM_TRACE")) fprintf(stderr, "[fadehip trace] prepare: streams and buffers %.3f ms, first copies and waits %.3f ms\n", (t1 - t0) * 1e3, (now_s() - t1) * 1e3);
    return 0;
}

int fadehip_bam_front(fadehip_bam_stream *st, const void *members, size_t n_bytes, int last) {
    if (!st) return set_err(nullptr, FADEHIP_E_INVALID, "stream is NULL");
    fadehip_ctx *ctx = st->ctx;
    if (n_bytes && !members) return set_err(ctx, FADEHIP_E_INVALID, "NULL argument");
    if (st->failed) return set_err(ctx, FADEHIP_E_STATE, "bam stream: an earlier call failed");
    if (st->ended) return set_err(ctx, FADEHIP_E_STATE, "bam stream: front after the last call");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int rc = bam_front_impl(st, (const uint8_t *)members, n_bytes, last, false);
    if (rc) {
        for (int q = 0; q < 2; q++)
            if (ctx->slots[q].stream) (void)hipStreamSynchronize(ctx->slots[q].stream);
        return bam_fail(st, rc);
    }
    return 0;
}

int fadehip_bam_front_raw(fadehip_bam_stream *st, const void *payload, size_t n_bytes, int last) {
    if (!st) return set_err(nullptr, FADEHIP_E_INVALID, "stream is NULL");
    fadehip_ctx *ctx = st->ctx;
    if (n_bytes && !payload) return set_err(ctx, FADEHIP_E_INVALID, "NULL argument");
    if (st->failed) return set_err(ctx, FADEHIP_E_STATE, "bam stream: an earlier call failed");
    if (st->ended) return set_err(ctx, FADEHIP_E_STATE, "bam stream: front after the last call");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int rc = bam_front_impl(st, (const uint8_t *)payload, n_bytes, last, true);
    if (rc) {
        for (int q = 0; q < 2; q++)
            if (ctx->slots[q].stream) (void)hipStreamSynchronize(ctx->slots[q].stream);
        return bam_fail(st, rc);
    }
    return 0;
}

// the back half's first step for call k: the call is finished (sizes read, rewrite enqueued) and its bytes go to the
// compressor on lane k & 1, the members packed into the stream's pinned buffer k % FADEHIP_BAM_CHUNKS
static int bam_submit_back(fadehip_bam_stream *st, uint64_t k) {
    fadehip_ctx *ctx = st->ctx;
    fadehip_bam_stream::Out *o = &st->ring[k % FADEHIP_BAM_CHUNKS];
    int rc;
    CallTrace tr("back", k);
    if ((rc = bam_finish_call(st, k))) return rc;
    tr.mark("finish call");
    st->k_sub = k + 1;
    if (!o->bytes || st->no_output) return 0;
    const int lane = (int)(k & 1);
    // (FADEHIP_BAM_BACK_STREAMS=1: both lanes on one stream — one HSA queue fewer, but call k's members then cross PCIe
    // before call k + 1's compressor starts instead of beside it)
    static const bool one_stream = getenv("FADEHIP_BAM_BACK_STREAMS") && atoi(getenv("FADEHIP_BAM_BACK_STREAMS")) == 1;
    if ((rc = bgzf_lane_ready(ctx, lane, one_stream))) return rc;
    tr.mark("lane (stream)");
    BgzfLane &l = ctx->bgzf[lane];
    if (hipStreamWaitEvent(l.stream, o->ready, 0) != hipSuccess) return set_err(ctx, FADEHIP_E_HIP, "bam stream: hipStreamWaitEvent failed");
    PinBuf &ob = st->outbuf[k % FADEHIP_BAM_CHUNKS];
    if (st->stored) {
        // uncompressed BGZF: the members' sizes are known here; the kernel stores them straight into the pinned buffer
        const uint32_t nb = (uint32_t)((o->bytes + bgzf::STORE_BLOCK - 1) / bgzf::STORE_BLOCK);
        const size_t total = o->bytes + (size_t)nb * (bgzf::STORE_MEMBER - bgzf::STORE_BLOCK);
        if ((rc = reserve_pinned(ctx, ob, (size_t)nb * bgzf::STORE_MEMBER))) return rc;
        hipLaunchKernelGGL(bgzf::bgzf_store_kernel, dim3(nb), dim3(bgzf::STORE_WG), 0, l.stream, (const uint8_t *)o->o.p, (uint64_t)o->bytes, nb, ob.p);
        if (hipGetLastError() != hipSuccess || hipEventRecord(l.done, l.stream) != hipSuccess) return set_err(ctx, FADEHIP_E_HIP, "bam stream: storing the members failed");
        l.h_out = ob.p;
        *l.h_total = total;
        l.n_bytes = o->bytes;
        l.state = 1;
        return 0;
    }
    rc = bgzf_enqueue(ctx, lane, (const uint8_t *)o->o.p, o->bytes, bgzf_pick_geom(ctx), &ob);
    tr.mark("compressor enqueued");
    return rc;
}

int fadehip_bam_back(fadehip_bam_stream *st, const uint8_t **out, size_t *out_bytes) {
    if (!st) return set_err(nullptr, FADEHIP_E_INVALID, "stream is NULL");
    fadehip_ctx *ctx = st->ctx;
    if (!out || !out_bytes) return set_err(ctx, FADEHIP_E_INVALID, "NULL argument");
    *out = nullptr;
    *out_bytes = 0;
    const uint64_t k = st->k_back;
    fadehip_bam_stream::Out *o = &st->ring[k % FADEHIP_BAM_CHUNKS];
    bool next_waiting = false;
    {
        std::unique_lock<std::mutex> l(st->mu);
        if ((o->state == 0 && st->k_sub <= k) || st->failed) return set_err(ctx, FADEHIP_E_STATE, st->failed ? "bam stream: an earlier call failed" : "bam stream: no front call is waiting for back");
        next_waiting = st->k_front > k + 1 && st->ring[(k + 1) % FADEHIP_BAM_CHUNKS].state != 0;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = 0;
    // the call's last step (sizes read, rewrite enqueued) is taken here, beside the front half's work on the next call
    if (st->k_sub <= k && (rc = bam_submit_back(st, k))) return bam_fail(st, rc);
    // The call after this one, if its front half is through on the device: its compressor is enqueued now, so that it
    // starts the moment this call's has left the CUs — while this call's members cross PCIe and the caller gets them.
    if (next_waiting && st->k_sub == k + 1) {
        bool through;
        {
            fadehip_bam_stream::Set &S = st->set[(k + 1) & 1];
            std::lock_guard<std::mutex> pl(S.mu);
            through = S.k != k + 1 || !S.pending || hipStreamQuery(ctx->slots[(k + 1) & 1].stream) == hipSuccess;
        }
        if (through && (rc = bam_submit_back(st, k + 1))) return bam_fail(st, rc);
    }
    if (o->bytes && !st->no_output) {
        const int lane = (int)(k & 1);
        BgzfLane &l = ctx->bgzf[lane];
        if (st->stored) {
            if (hipEventSynchronize(l.done) != hipSuccess) return bam_fail(st, set_err(ctx, FADEHIP_E_HIP, "bam stream: storing the members failed"));
            l.state = 0;
            *out = l.h_out;
            *out_bytes = (size_t)*l.h_total;
        } else if ((rc = fadehip_bgzf_deflate_wait(ctx, lane, out, out_bytes))) return bam_fail(st, rc);
    } else if (o->ready) {
        (void)hipEventSynchronize(o->ready);
    }
    {
        std::lock_guard<std::mutex> l(st->mu);
        o->state = 0;
        st->k_back++;
    }
    st->cv.notify_all();
    return 0;
}

int fadehip_bam_totals(fadehip_bam_stream *st, int64_t stats[8], int64_t *n_records, int64_t *n_oversize) {
    if (!st) return set_err(nullptr, FADEHIP_E_INVALID, "stream is NULL");
    std::lock_guard<std::mutex> l(st->mu);
    if (stats) memcpy(stats, st->stats, sizeof st->stats);
    if (n_records) *n_records = st->n_records;
    if (n_oversize) *n_oversize = st->n_oversize;
    return 0;
}

void fadehip_bam_close(fadehip_bam_stream *st) {
    if (!st) return;
    {
        std::lock_guard<std::mutex> l(st->mu);
        st->closing = true;
    }
    st->cv.notify_all();
    fadehip_ctx *ctx = st->ctx;
    if (getenv("FADEHIP_BAM_PROF"))
        fprintf(stderr, "[fadehip bam] %llu front calls: A enqueue (copy / inflate, frame, pack count) %.3f s, wait for A %.3f | B enqueue (pack, run, tag sizes) %.3f | "
                        "C (wait for B, rewrite enqueued; taken by back or front) %.3f | segments walked again %lld\n", (unsigned long long)st->k_front, st->t_inflate, st->t_frame, st->t_run, st->t_tags,
                (long long)st->n_redone);
    (void)hipSetDevice(ctx->device);
    for (int q = 0; q < 2; q++) {
        if (ctx->slots[q].stream) (void)hipStreamSynchronize(ctx->slots[q].stream);
        ctx->slots[q].device_only = false;
        ctx->slots[q].wide_all = false;
        if (ctx->slots[q].state == 2) ctx->slots[q].state = 0;  // (a call that was never finished: nothing of it is handed out)
    }
    for (BgzfLane &l : ctx->bgzf)
        if (l.stream) (void)hipStreamSynchronize(l.stream);
    release(st->names_text);
    release(st->names_off);
    for (auto &S : st->set) {
        for (DevBuf *b : {&S.comp, &S.blocks, &S.status, &S.ticket, &S.u, &S.seg, &S.slots, &S.rec_off, &S.info, &S.sent_of, &S.art_of, &S.out_size, &S.blk32,
                          &S.blk64, &S.counts})
            release(*b);
        release(S.h_blocks);
        release(S.h_counts);
    }
    for (auto &ob : st->outbuf) release(ob);
    for (auto &o : st->ring) {
        release(o.o);
        if (o.ready) (void)hipEventDestroy(o.ready);
    }
    delete st;
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

int fadehip_stats_allreduce_rank(fadehip_ctx *ctx, int rank, int n_ranks, const char *id_path, int64_t *counters, int count) {
    if (!ctx || !id_path || !counters || count <= 0 || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return set_err(ctx, FADEHIP_E_INVALID, "bad arguments");
    if (n_ranks == 1) return 0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    if (rank == 0) {
        ncclResult_t nr = ncclGetUniqueId(&id);
        if (nr != ncclSuccess) return set_err(ctx, FADEHIP_E_RCCL, "ncclGetUniqueId failed: %s", ncclGetErrorString(nr));
        const std::string tmp = std::string(id_path) + ".tmp";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&id, 1, sizeof id, f) != sizeof id) { if (f) fclose(f); return set_err(ctx, FADEHIP_E_INVALID, "cannot write %s", tmp.c_str()); }
        fclose(f);
        if (rename(tmp.c_str(), id_path) != 0) return set_err(ctx, FADEHIP_E_INVALID, "cannot rename %s", tmp.c_str());
    } else {
        bool got = false;
        for (int tries = 0; tries < 60000 && !got; tries++) {
            if (FILE *f = fopen(id_path, "rb")) {
                got = fread(&id, 1, sizeof id, f) == sizeof id;
                fclose(f);
            }
            if (!got) {
                struct timespec ts = {0, 1000000};
                nanosleep(&ts, nullptr);
            }
        }
        if (!got) return set_err(ctx, FADEHIP_E_RCCL, "rank %d: no ncclUniqueId appeared in %s", rank, id_path);
    }
    ncclComm_t comm;
    ncclResult_t nr = ncclCommInitRank(&comm, n_ranks, id, rank);
    if (nr != ncclSuccess) return set_err(ctx, FADEHIP_E_RCCL, "ncclCommInitRank failed: %s", ncclGetErrorString(nr));
    void *buf = nullptr;
    int rc = 0;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipMalloc(&buf, sizeof(int64_t) * count) != hipSuccess ||
        hipMemcpy(buf, counters, sizeof(int64_t) * count, hipMemcpyHostToDevice) != hipSuccess)
        rc = set_err(ctx, FADEHIP_E_HIP, "staging counters failed");
    if (!rc) {
        nr = ncclAllReduce(buf, buf, count, ncclInt64, ncclSum, comm, st);
        if (nr != ncclSuccess) rc = set_err(ctx, FADEHIP_E_RCCL, "ncclAllReduce failed: %s", ncclGetErrorString(nr));
    }
    if (!rc && (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(counters, buf, sizeof(int64_t) * count, hipMemcpyDeviceToHost) != hipSuccess))
        rc = set_err(ctx, FADEHIP_E_HIP, "reading the reduced counters failed");
    if (buf) (void)hipFree(buf);
    ncclCommDestroy(comm);
    if (st) (void)hipStreamDestroy(st);
    return rc;
}

}  // extern "C"
