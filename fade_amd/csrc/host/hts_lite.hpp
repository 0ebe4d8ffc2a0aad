// hts_lite.hpp — the slice of htslib/dhtslib that `fade annotate` touches, over zlib only
// (htslib is not present in this image: SURVEY.md H2).  Replaces, for this path:
//   SAMReader(args[1]) / bam.allRecords           source/anno.d:22,44      -> Reader
//   IndexedFastaFile(args[2])                     source/anno.d:23         -> load_fasta (whole file, uploaded to HBM once)
//   header.dup + header.addLine(@PG ...)          source/anno.d:24-32      -> Header::add_pg
//   getWriter(con, header): SAM / uBAM / BAM      source/util.d:65-76      -> Writer
//   rec["rs"] = ubyte, rec["am"] = string ...     source/anno.d:63,94-106  -> Rec::aux_update_int / aux_update_str
// BAM records are kept in their on-disk byte layout so CIGAR ops and the 4-bit sequence can be handed
// to the device without re-encoding.
#pragma once
#include <zlib.h>
#include <algorithm>
#include <sys/mman.h>
#include <atomic>
#include <exception>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <memory>
#include <vector>

#include "crc32_fast.hpp"
#include "deflate_fast.hpp"
#include "inflate_fast.hpp"

namespace htsl {

// ------------------------------------------------------------------ small thread pool
// Thread CPU time of the pools' work by kind (annotate --timing): where the host's core-seconds go.
enum CpuKind { CPU_INFLATE, CPU_FRAME, CPU_SAM_PARSE, CPU_PACK, CPU_TAGS, CPU_COPY, CPU_DEFLATE, CPU_SAM_FORMAT, CPU_KINDS };
inline std::atomic<int64_t> *cpu_meter() {
    static std::atomic<int64_t> ns[CPU_KINDS];
    return ns;
}
inline const char *cpu_kind_name(int k) {
    static const char *const names[CPU_KINDS] = {"inflate+crc", "frame", "sam parse", "pack", "tags", "copy", "deflate+crc", "sam format"};
    return names[k];
}
inline int64_t thread_cpu_ns() {
    timespec ts;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    return (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec;
}

// Worker threads shared by every stage of a pipeline: parallel_for may be called from several threads at once; their
// jobs are served first come, first served, at task granularity.  (A pool per stage put twice the threads on the cores:
// every fork-join then waited for whichever of its threads the scheduler had parked for a time slice.)
class Pool {
public:
    explicit Pool(int n) : n_(std::max(1, n)) {
        for (int i = 1; i < n_; i++) th_.emplace_back([this] { worker(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> l(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return n_; }
    // fn(i) for i in [0, count), dynamic distribution; the caller works on its own job too and returns when all are done.
    // kind >= 0: the threads' CPU time inside fn is added to cpu_meter()[kind]
    void parallel_for(size_t count, const std::function<void(size_t)> &fn, int kind = -1) {
        if (count == 0) return;
        if (n_ == 1 || count == 1) {
            const int64_t t0 = kind >= 0 ? thread_cpu_ns() : 0;
            for (size_t i = 0; i < count; i++) fn(i);
            if (kind >= 0) cpu_meter()[kind] += thread_cpu_ns() - t0;
            return;
        }
        // (an exception thrown by fn on any thread ends the job early and is rethrown here, on the caller's thread)
        Job job;
        job.fn = &fn;
        job.count = count;
        job.kind = kind;
        {
            std::lock_guard<std::mutex> l(m_);
            jobs_.push_back(&job);
        }
        cv_.notify_all();
        work_on(job);
        // every index is handed out: take the job off the list (no new helpers), then wait for the helpers still inside fn
        std::unique_lock<std::mutex> l(m_);
        jobs_.erase(std::find(jobs_.begin(), jobs_.end(), &job));
        job.idle.wait(l, [&] { return job.helpers == 0; });
        if (job.error) std::rethrow_exception(job.error);
    }

private:
    struct Job {
        const std::function<void(size_t)> *fn = nullptr;
        size_t count = 0;
        int kind = -1;
        std::atomic<size_t> next{0};
        int helpers = 0;  // workers inside work_on (guarded by m_)
        std::condition_variable idle;
        std::exception_ptr error;  // the first exception out of fn (guarded by err_m)
        std::mutex err_m;
    };
    static void work_on(Job &j) {
        const int64_t t0 = j.kind >= 0 ? thread_cpu_ns() : 0;
        try {
            for (;;) {
                const size_t i = j.next.fetch_add(1);
                if (i >= j.count) break;
                (*j.fn)(i);
            }
        } catch (...) {
            j.next.store(j.count);  // nothing more is handed out
            std::lock_guard<std::mutex> l(j.err_m);
            if (!j.error) j.error = std::current_exception();
        }
        if (j.kind >= 0) cpu_meter()[j.kind] += thread_cpu_ns() - t0;
    }
    void worker() {
        std::unique_lock<std::mutex> l(m_);
        for (;;) {
            Job *j = nullptr;
            cv_.wait(l, [&] {
                if (stop_) return true;
                for (Job *c : jobs_)
                    if (c->next.load() < c->count) {
                        j = c;
                        return true;
                    }
                return false;
            });
            if (stop_) return;
            j->helpers++;
            l.unlock();
            work_on(*j);
            l.lock();
            if (--j->helpers == 0) j->idle.notify_all();
        }
    }
    int n_;
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::vector<Job *> jobs_;
    bool stop_ = false;
};

// ------------------------------------------------------------------ stage hand-off
template <class T>
class BoundedQueue {
public:
    explicit BoundedQueue(size_t cap) : cap_(cap) {}
    void push(T v) {
        std::unique_lock<std::mutex> l(m_);
        not_full_.wait(l, [&] { return q_.size() < cap_ || closed_; });
        q_.push_back(std::move(v));
        not_empty_.notify_one();
    }
    bool pop(T &v) {  // false once closed and drained
        std::unique_lock<std::mutex> l(m_);
        not_empty_.wait(l, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        v = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
    void close() {
        std::lock_guard<std::mutex> l(m_);
        closed_ = true;
        not_empty_.notify_all();
        not_full_.notify_all();
    }

private:
    size_t cap_;
    std::deque<T> q_;
    std::mutex m_;
    std::condition_variable not_full_, not_empty_;
    bool closed_ = false;
};

// A BGZF member that claims ISIZE 0 (the end-of-file marker, or an empty block in the middle of a file): it is what it
// says only if its DEFLATE stream ends without a byte of output and its CRC32 field is that of no bytes.  A damaged trailer
// (ISIZE zeroed) must not make a member's records vanish without a word — htslib goes by what the stream inflates to.
inline bool bgzf_empty_member_ok(const uint8_t *deflate, size_t n, const uint8_t *trailer) {
    uint32_t crc;
    memcpy(&crc, trailer, 4);
    if (crc != 0) return false;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return false;
    uint8_t sink[8];
    zs.next_in = const_cast<uint8_t *>(deflate);
    zs.avail_in = (uInt)n;
    zs.next_out = sink;
    zs.avail_out = sizeof sink;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == 0;
    inflateEnd(&zs);
    return ok;
}

// ------------------------------------------------------------------ header
struct Header {
    std::string text;                // SAM header text (lines end with '\n')
    std::vector<std::string> names;  // @SQ SN
    std::vector<int64_t> lens;       // @SQ LN

    int tid_of(const std::string &n) const {
        for (size_t i = 0; i < names.size(); i++)
            if (names[i] == n) return (int)i;
        return -1;
    }
    void parse_sq_from_text() {
        names.clear();
        lens.clear();
        size_t p = 0;
        while (p < text.size()) {
            size_t e = text.find('\n', p);
            if (e == std::string::npos) e = text.size();
            if (text.compare(p, 4, "@SQ\t") == 0) {
                std::string sn;
                int64_t ln = 0;
                size_t f = p + 4;
                while (f < e) {
                    size_t g = text.find('\t', f);
                    if (g == std::string::npos || g > e) g = e;
                    if (text.compare(f, 3, "SN:") == 0) sn = text.substr(f + 3, g - f - 3);
                    if (text.compare(f, 3, "LN:") == 0) ln = std::strtoll(text.c_str() + f + 3, nullptr, 10);
                    f = g + 1;
                }
                names.push_back(sn);
                lens.push_back(ln);
            }
            p = e + 1;
        }
    }
    // ID of the last @PG line ("" if none) — header.valueByPos(PG, numRecords(PG)-1, "ID"), anno.d:30
    std::string last_pg_id() const {
        std::string id;
        size_t p = 0;
        while (p < text.size()) {
            size_t e = text.find('\n', p);
            if (e == std::string::npos) e = text.size();
            if (text.compare(p, 4, "@PG\t") == 0) {
                size_t f = text.find("\tID:", p);
                if (f != std::string::npos && f < e) {
                    size_t g = text.find_first_of("\t\n", f + 4);
                    if (g == std::string::npos) g = e;
                    id = text.substr(f + 4, g - f - 4);
                }
            }
            p = e + 1;
        }
        return id;
    }
    // anno.d:25-32
    void add_pg(const std::string &id, const std::string &pn, const std::string &vn, const std::string &cl) {
        std::string pp = last_pg_id();
        if (!text.empty() && text.back() != '\n') text += '\n';
        text += "@PG\tID:" + id + "\tPN:" + pn + "\tVN:" + vn;
        if (!pp.empty()) text += "\tPP:" + pp;
        text += "\tCL:" + cl + "\n";
    }
};

// ------------------------------------------------------------------ record (BAM byte layout, no block_size)
static const char NT16_STR[] = "=ACMGRSVTWYHKDBN";
static const char CIGAR_STR[] = "MIDNSHP=XB";

inline int aux_type_size(uint8_t t) {
    switch (t) {
        case 'A': case 'c': case 'C': return 1;
        case 's': case 'S': return 2;
        case 'i': case 'I': case 'f': return 4;
        case 'd': return 8;
        default: return 0;
    }
}

// The read-only view of one BAM record (the bytes after block_size), shared by the owning Rec and the in-place RecView:
// D provides data() / size().
template <class D>
struct RecApi {
    const uint8_t *bytes() const { return static_cast<const D *>(this)->data(); }
    size_t nbytes() const { return static_cast<const D *>(this)->size(); }
    template <class T> T rd(size_t off) const { T v; memcpy(&v, bytes() + off, sizeof(T)); return v; }
    int32_t tid() const { return rd<int32_t>(0); }
    int32_t pos() const { return rd<int32_t>(4); }
    int l_qname() const { return bytes()[8]; }
    int mapq() const { return bytes()[9]; }
    int n_cigar() const { return rd<uint16_t>(12); }
    int flag() const { return rd<uint16_t>(14); }
    int32_t l_seq() const { return rd<int32_t>(16); }
    int32_t mtid() const { return rd<int32_t>(20); }
    int32_t mpos() const { return rd<int32_t>(24); }
    int32_t tlen() const { return rd<int32_t>(28); }
    const char *qname() const { return (const char *)bytes() + 32; }
    size_t cigar_off() const { return 32 + (size_t)l_qname(); }
    const uint8_t *cigar_bytes() const { return bytes() + cigar_off(); }
    uint32_t cigar_op(int k) const { return rd<uint32_t>(cigar_off() + 4 * (size_t)k); }
    size_t seq_off() const { return cigar_off() + 4 * (size_t)n_cigar(); }
    const uint8_t *seq() const { return bytes() + seq_off(); }
    size_t qual_off() const { return seq_off() + ((size_t)l_seq() + 1) / 2; }
    const uint8_t *qual() const { return bytes() + qual_off(); }
    size_t aux_off() const { return qual_off() + (size_t)l_seq(); }
    // The fixed fields must describe a layout that fits the record, and the aux area must be a sequence of whole fields
    // that ends exactly at the end of the record: every accessor, the tag updates and the SAM formatter index with these
    // lengths, so a truncated or crafted record is rejected here, once, when it is read.
    bool layout_ok() const {
        const size_t n = nbytes();
        if (!(n >= 32 && l_seq() >= 0 && l_qname() >= 1 && aux_off() <= n)) return false;
        size_t p = aux_off();
        while (p < n) {
            if (p + 3 > n) return false;
            const size_t fs = aux_field_size(p + 2);
            if (!fs) return false;
            p += 2 + fs;
        }
        return p == n;
    }
    // size in bytes of the aux field whose type byte is at p (type byte included), 0 if the field is malformed or does
    // not fit inside the record
    size_t aux_field_size(size_t p) const {
        const size_t n = nbytes();
        const uint8_t *d = bytes();
        if (p >= n) return 0;
        const uint8_t t = d[p];
        const int s = aux_type_size(t);
        if (s) return p + 1 + (size_t)s <= n ? 1 + (size_t)s : 0;
        if (t == 'Z' || t == 'H') {
            size_t q = p + 1;
            while (q < n && d[q]) q++;
            return q < n ? q - p + 1 : 0;
        }
        if (t == 'B') {
            if (p + 6 > n) return 0;
            const int es = aux_type_size(d[p + 1]);
            const uint64_t cnt = rd<uint32_t>(p + 2);
            if (!es) return 0;
            const uint64_t fs = 6 + (uint64_t)es * cnt;  // 64-bit: a 32-bit count times 8 overflows size_t nowhere, uint32 everywhere
            return fs <= (uint64_t)(n - p) ? (size_t)fs : 0;
        }
        return 0;
    }
    // offset of the tag's 2-byte name, or npos
    size_t aux_find(const char tag[2]) const {
        const size_t n = nbytes();
        const uint8_t *d = bytes();
        size_t p = aux_off();
        while (p + 3 <= n) {
            size_t fs = aux_field_size(p + 2);
            if (!fs) break;
            if (d[p] == (uint8_t)tag[0] && d[p + 1] == (uint8_t)tag[1]) return p;
            p += 2 + fs;
        }
        return std::string::npos;
    }
    bool aux_exists(const char tag[2]) const { return aux_find(tag) != std::string::npos; }
};

// a record framed in place in a block of inflated BAM bytes (RecordBlock): no copy, no allocation
struct RecView : RecApi<RecView> {
    const uint8_t *p = nullptr;
    size_t n = 0;
    RecView() = default;
    RecView(const uint8_t *p_, size_t n_) : p(p_), n(n_) {}
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
};

struct Rec : RecApi<Rec> {
    std::vector<uint8_t> d;
    const uint8_t *data() const { return d.data(); }
    size_t size() const { return d.size(); }
    template <class T> void wr(size_t off, T v) { memcpy(d.data() + off, &v, sizeof(T)); }
    void aux_append(const char tag[2], uint8_t type, const void *data, size_t len) {
        size_t o = d.size();
        d.resize(o + 3 + len);
        d[o] = (uint8_t)tag[0];
        d[o + 1] = (uint8_t)tag[1];
        d[o + 2] = type;
        memcpy(d.data() + o + 3, data, len);
    }
    // htslib bam_aux_update_int (sam.c) for a non-negative value.  Absent: appended with the smallest type.  Present as an
    // integer whose slot is wide enough: the slot is reused and its type letter becomes the UNSIGNED one of that size
    // ("\0CS\0I"[old_sz] — an rs:c / rs:s / rs:i leaves as rs:C / rs:S / rs:I); too narrow: replaced at the same position.
    // Present with a non-integer type: htslib returns EINVAL and leaves the record as it is, and so does this.
    void aux_update_uint(const char tag[2], uint32_t v) {
        uint8_t type = v < 0xff ? 'C' : v < 0xffff ? 'S' : 'I';  // (htslib compares with `<`: 255 already takes 'S')
        uint8_t buf[4];
        memcpy(buf, &v, 4);
        size_t p = aux_find(tag);
        size_t need = type == 'C' ? 1 : type == 'S' ? 2 : 4;
        if (p == std::string::npos) {
            aux_append(tag, type, buf, need);
            return;
        }
        const uint8_t ot = d[p + 2];
        const size_t os = (size_t)aux_type_size(ot);
        const bool is_int = ot == 'c' || ot == 'C' || ot == 's' || ot == 'S' || ot == 'i' || ot == 'I';
        if (!is_int) return;
        if (os >= need) {
            d[p + 2] = (uint8_t)"\0CS\0I"[os];
            memcpy(d.data() + p + 3, buf, os);  // little endian: low bytes first
            return;
        }
        const size_t fs = aux_field_size(p + 2);
        replace_bytes(p + 2, fs, type, buf, need);
    }
    // htslib bam_aux_update_str: replaced at the same position when present as 'Z', appended when absent; present with
    // another type: EINVAL, the record stays as it is
    void aux_update_str(const char tag[2], const std::string &s) {
        size_t p = aux_find(tag);
        if (p == std::string::npos) {
            aux_append(tag, 'Z', s.c_str(), s.size() + 1);
            return;
        }
        if (d[p + 2] != 'Z') return;
        const size_t fs = aux_field_size(p + 2);
        replace_bytes(p + 2, fs, 'Z', s.c_str(), s.size() + 1);
    }

private:
    void replace_bytes(size_t at, size_t old_len, uint8_t type, const void *data, size_t len) {
        std::vector<uint8_t> tail(d.begin() + at + old_len, d.end());
        d.resize(at);
        d.push_back(type);
        d.insert(d.end(), (const uint8_t *)data, (const uint8_t *)data + len);
        d.insert(d.end(), tail.begin(), tail.end());
    }
};

template <class R>
inline int64_t cigar_ref_len(const R &r) {
    int64_t n = 0;
    for (int k = 0; k < r.n_cigar(); k++) {
        uint32_t c = r.cigar_op(k), op = c & 15;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) n += c >> 4;
    }
    return n;
}

inline int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

// ------------------------------------------------------------------ SAM text <-> Rec
struct Nt16Table {
    uint8_t t[256];
    Nt16Table() {
        memset(t, 15, sizeof t);
        for (int k = 0; k < 16; k++) {
            t[(unsigned char)NT16_STR[k]] = (uint8_t)k;
            t[(unsigned char)(NT16_STR[k] | 0x20)] = (uint8_t)k;
        }
    }
};
inline const Nt16Table &nt16_table() {
    static Nt16Table tb;
    return tb;
}

inline void append_int(std::string &s, int64_t v) {
    char buf[24];
    int n = 0;
    bool neg = v < 0;
    uint64_t u = neg ? (uint64_t)(-v) : (uint64_t)v;
    do { buf[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (neg) s += '-';
    while (n) s += buf[--n];
}

// One SAM line (no newline) -> record.  Throws std::runtime_error on malformed input.
inline void sam_parse(const char *line, size_t len, const Header &h, Rec &r) {
    const char *f[12];
    size_t fl[12];
    int nf = 0;
    const char *p = line, *end = line + len;
    while (nf < 11 && p <= end) {
        const char *q = (const char *)memchr(p, '\t', (size_t)(end - p));
        if (!q) q = end;
        f[nf] = p;
        fl[nf] = (size_t)(q - p);
        nf++;
        p = q + 1;
    }
    if (nf < 11) throw std::runtime_error("SAM line has fewer than 11 fields");
    const char *aux = p <= end ? p : end;
    const std::string rname(f[2], fl[2]), rnext(f[6], fl[6]);
    const int tid = rname == "*" ? -1 : h.tid_of(rname);
    if (rname != "*" && tid < 0) throw std::runtime_error("SAM record names unknown reference " + rname);
    const int mtid = rnext == "*" ? -1 : rnext == "=" ? tid : h.tid_of(rnext);
    const int64_t pos = std::strtoll(f[3], nullptr, 10) - 1;
    std::vector<uint32_t> cig;
    if (!(fl[5] == 1 && f[5][0] == '*')) {
        uint64_t num = 0;
        for (size_t k = 0; k < fl[5]; k++) {
            char c = f[5][k];
            if (c >= '0' && c <= '9') num = num * 10 + (uint64_t)(c - '0');
            else {
                const char *o = strchr(CIGAR_STR, c);
                if (!o) throw std::runtime_error("bad CIGAR operator");
                cig.push_back((uint32_t)(num << 4) | (uint32_t)(o - CIGAR_STR));
                num = 0;
            }
        }
    }
    // BAM holds l_read_name in one byte (the name and its NUL) and n_cigar_op in 16 bits: a line beyond either cannot be laid
    // out as a record (htslib moves such a CIGAR into a CG tag; this reader refuses the line rather than write fields that
    // contradict the bytes behind them)
    if (fl[0] > 254) throw std::runtime_error("SAM line with a QNAME longer than 254 characters");
    if (cig.size() > 65535) throw std::runtime_error("SAM line with more than 65535 CIGAR operations");
    const bool noseq = fl[9] == 1 && f[9][0] == '*';
    const size_t lseq = noseq ? 0 : fl[9];
    const size_t lq = fl[0] + 1;
    r.d.assign(32 + lq + 4 * cig.size() + (lseq + 1) / 2 + lseq, 0);
    int64_t reflen = 0;
    for (uint32_t c : cig) {
        uint32_t op = c & 15;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += c >> 4;
    }
    r.wr<int32_t>(0, tid);
    r.wr<int32_t>(4, (int32_t)pos);
    r.d[8] = (uint8_t)lq;
    r.d[9] = (uint8_t)std::strtol(f[4], nullptr, 10);
    r.wr<uint16_t>(10, (uint16_t)reg2bin(pos < 0 ? 0 : pos, (pos < 0 ? 0 : pos) + (reflen > 0 ? reflen : 1)));
    r.wr<uint16_t>(12, (uint16_t)cig.size());
    r.wr<uint16_t>(14, (uint16_t)std::strtol(f[1], nullptr, 10));
    r.wr<int32_t>(16, (int32_t)lseq);
    r.wr<int32_t>(20, mtid);
    r.wr<int32_t>(24, (int32_t)(std::strtoll(f[7], nullptr, 10) - 1));
    r.wr<int32_t>(28, (int32_t)std::strtoll(f[8], nullptr, 10));
    memcpy(r.d.data() + 32, f[0], fl[0]);
    size_t o = 32 + lq;
    if (!cig.empty()) memcpy(r.d.data() + o, cig.data(), 4 * cig.size());
    o += 4 * cig.size();
    const uint8_t *tb = nt16_table().t;
    for (size_t k = 0; k < lseq; k++) r.d[o + (k >> 1)] |= (uint8_t)(tb[(unsigned char)f[9][k]] << ((~k & 1) << 2));
    o += (lseq + 1) / 2;
    if (fl[10] == 1 && f[10][0] == '*') memset(r.d.data() + o, 0xff, lseq);
    else {
        if (fl[10] != lseq) throw std::runtime_error("SEQ and QUAL differ in length");
        for (size_t k = 0; k < lseq; k++) r.d[o + k] = (uint8_t)(f[10][k] - 33);
    }
    // aux
    while (aux < end) {
        const char *q = (const char *)memchr(aux, '\t', (size_t)(end - aux));
        if (!q) q = end;
        const size_t n = (size_t)(q - aux);
        if (n < 5 || aux[2] != ':' || aux[4] != ':') throw std::runtime_error("malformed SAM tag");
        const char tag[2] = {aux[0], aux[1]};
        const char ty = aux[3];
        const char *v = aux + 5;
        const size_t vl = n - 5;
        if (ty == 'A') r.aux_append(tag, 'A', v, 1);
        else if (ty == 'i') {
            const long long x = std::strtoll(v, nullptr, 10);
            if (x < 0) {
                if (x >= -128) { int8_t y = (int8_t)x; r.aux_append(tag, 'c', &y, 1); }
                else if (x >= -32768) { int16_t y = (int16_t)x; r.aux_append(tag, 's', &y, 2); }
                else { int32_t y = (int32_t)x; r.aux_append(tag, 'i', &y, 4); }
            } else {
                if (x <= 0xff) { uint8_t y = (uint8_t)x; r.aux_append(tag, 'C', &y, 1); }
                else if (x <= 0xffff) { uint16_t y = (uint16_t)x; r.aux_append(tag, 'S', &y, 2); }
                else { uint32_t y = (uint32_t)x; r.aux_append(tag, 'I', &y, 4); }
            }
        } else if (ty == 'f') {
            float y = std::strtof(v, nullptr);
            r.aux_append(tag, 'f', &y, 4);
        } else if (ty == 'Z' || ty == 'H') {
            std::string s(v, vl);
            r.aux_append(tag, (uint8_t)ty, s.c_str(), s.size() + 1);
        } else if (ty == 'B') {
            if (vl < 1) throw std::runtime_error("malformed B tag");
            const char st = v[0];
            const int es = aux_type_size((uint8_t)st);
            if (!es) throw std::runtime_error("bad B subtype");
            std::vector<uint8_t> buf;
            uint32_t cnt = 0;
            const char *c = v + 1;
            while (c < v + vl) {
                if (*c == ',') c++;
                char *e2 = nullptr;
                uint8_t tmp[8];
                if (st == 'f') { float y = std::strtof(c, &e2); memcpy(tmp, &y, 4); }
                else { long long y = std::strtoll(c, &e2, 10); memcpy(tmp, &y, 8); }
                if (e2 == c) break;
                buf.insert(buf.end(), tmp, tmp + es);
                cnt++;
                c = e2;
            }
            std::vector<uint8_t> all(5 + buf.size());
            all[0] = (uint8_t)st;
            memcpy(all.data() + 1, &cnt, 4);
            if (!buf.empty()) memcpy(all.data() + 5, buf.data(), buf.size());
            r.aux_append(tag, 'B', all.data(), all.size());
        } else throw std::runtime_error("unknown SAM tag type");
        aux = q + 1;
    }
}

template <class R>
inline void sam_format(const R &r, const Header &h, std::string &s) {
    s.append(r.qname());
    s += '\t';
    append_int(s, r.flag());
    s += '\t';
    const int tid = r.tid(), mtid = r.mtid();
    if (tid >= 0 && tid < (int)h.names.size()) s += h.names[tid]; else s += '*';
    s += '\t';
    append_int(s, (int64_t)r.pos() + 1);
    s += '\t';
    append_int(s, r.mapq());
    s += '\t';
    if (r.n_cigar() == 0) s += '*';
    for (int k = 0; k < r.n_cigar(); k++) {
        uint32_t c = r.cigar_op(k);
        append_int(s, c >> 4);
        s += CIGAR_STR[std::min<uint32_t>(c & 15, 9)];
    }
    s += '\t';
    if (mtid < 0) s += '*';
    else if (mtid == tid) s += '=';
    else if (mtid < (int)h.names.size()) s += h.names[mtid];
    else s += '*';
    s += '\t';
    append_int(s, (int64_t)r.mpos() + 1);
    s += '\t';
    append_int(s, r.tlen());
    s += '\t';
    const int lseq = r.l_seq();
    if (lseq == 0) s += '*';
    else {
        const uint8_t *sq = r.seq();
        for (int k = 0; k < lseq; k++) s += NT16_STR[(sq[k >> 1] >> ((~k & 1) << 2)) & 15];
    }
    s += '\t';
    const uint8_t *ql = r.qual();
    if (lseq == 0 || ql[0] == 0xff) s += '*';
    else for (int k = 0; k < lseq; k++) s += (char)(ql[k] + 33);
    size_t p = r.aux_off();
    char buf[64];
    while (p + 3 <= r.nbytes()) {
        const size_t fs = r.aux_field_size(p + 2);
        if (!fs) break;
        s += '\t';
        s += (char)r.bytes()[p];
        s += (char)r.bytes()[p + 1];
        s += ':';
        const uint8_t t = r.bytes()[p + 2];
        const uint8_t *v = r.bytes() + p + 3;
        switch (t) {
            case 'A': s += "A:"; s += (char)v[0]; break;
            case 'c': s += "i:"; append_int(s, (int8_t)v[0]); break;
            case 'C': s += "i:"; append_int(s, v[0]); break;
            case 's': { int16_t x; memcpy(&x, v, 2); s += "i:"; append_int(s, x); break; }
            case 'S': { uint16_t x; memcpy(&x, v, 2); s += "i:"; append_int(s, x); break; }
            case 'i': { int32_t x; memcpy(&x, v, 4); s += "i:"; append_int(s, x); break; }
            case 'I': { uint32_t x; memcpy(&x, v, 4); s += "i:"; append_int(s, x); break; }
            case 'f': { float x; memcpy(&x, v, 4); snprintf(buf, sizeof buf, "%g", x); s += "f:"; s += buf; break; }
            case 'd': { double x; memcpy(&x, v, 8); snprintf(buf, sizeof buf, "%g", x); s += "d:"; s += buf; break; }
            case 'Z': s += "Z:"; s += (const char *)v; break;
            case 'H': s += "H:"; s += (const char *)v; break;
            case 'B': {
                const uint8_t st = v[0];
                uint32_t n;
                memcpy(&n, v + 1, 4);
                s += "B:";
                s += (char)st;
                const uint8_t *e = v + 5;
                const int es = aux_type_size(st);
                for (uint32_t k = 0; k < n; k++, e += es) {
                    s += ',';
                    switch (st) {
                        case 'c': append_int(s, (int8_t)e[0]); break;
                        case 'C': append_int(s, e[0]); break;
                        case 's': { int16_t x; memcpy(&x, e, 2); append_int(s, x); break; }
                        case 'S': { uint16_t x; memcpy(&x, e, 2); append_int(s, x); break; }
                        case 'i': { int32_t x; memcpy(&x, e, 4); append_int(s, x); break; }
                        case 'I': { uint32_t x; memcpy(&x, e, 4); append_int(s, x); break; }
                        case 'f': { float x; memcpy(&x, e, 4); snprintf(buf, sizeof buf, "%g", x); s += buf; break; }
                    }
                }
                break;
            }
        }
        p += 2 + fs;
    }
    s += '\n';
}

// ------------------------------------------------------------------ BGZF
static const uint8_t BGZF_EOF[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43,
                                     0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// FADE_BGZF_CODEC=zlib selects zlib (level 6 deflate, what htslib does by default, and zlib's inflate) instead of
// deflate_fast.hpp / inflate_fast.hpp;
// FADE_BGZF_EFFORT=1..4 picks FastDeflate's effort (default 2)
inline bool bgzf_use_zlib() {
    static const bool z = [] {
        const char *e = getenv("FADE_BGZF_CODEC");
        return e && strcmp(e, "zlib") == 0;
    }();
    return z;
}

// compress one block of <= 0xff00 bytes into out (appended); level 0 gives stored (uBAM)
inline void bgzf_compress_block(const uint8_t *src, size_t n, int level, std::vector<uint8_t> &out,
                                const FastDeflate::Hint *hints = nullptr, size_t n_hints = 0) {
    uint8_t buf[0x10000 + 64];
    size_t clen;
    if (level > 0 && !bgzf_use_zlib()) {
        static thread_local std::unique_ptr<FastDeflate> fd;
        if (!fd) {
            const char *e = getenv("FADE_BGZF_EFFORT");  // 1 fastest ... 4 smallest (deflate_fast.hpp)
            fd.reset(new FastDeflate(e ? atoi(e) : 2));
        }
        clen = fd->compress(src, n, buf + 18, hints, n_hints);
    } else {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2");
        zs.next_in = const_cast<uint8_t *>(src);
        zs.avail_in = (uInt)n;
        zs.next_out = buf + 18;
        zs.avail_out = sizeof buf - 18 - 8;
        const int rc = deflate(&zs, Z_FINISH);
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) throw std::runtime_error("BGZF deflate overflow");
        clen = zs.total_out;
    }
    const size_t bsize = 18 + clen + 8;
    static const uint8_t hdr[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0};
    memcpy(buf, hdr, 16);
    const uint16_t bs = (uint16_t)(bsize - 1);
    memcpy(buf + 16, &bs, 2);
    const uint32_t crc = crc32_fast(0, src, n), isz = (uint32_t)n;
    memcpy(buf + 18 + clen, &crc, 4);
    memcpy(buf + 18 + clen + 4, &isz, 4);
    out.insert(out.end(), buf, buf + bsize);
}

class ByteSource {  // buffered FILE* with peek
public:
    explicit ByteSource(FILE *f) : f_(f), buf_(1 << 20) {}
    // continue at an absolute file offset (a seekable file; what was buffered is dropped)
    void seek(uint64_t off) {
        if (fseeko(f_, (off_t)off, SEEK_SET) != 0) throw std::runtime_error("cannot seek in the input (lanes need a regular file)");
        beg_ = end_ = 0;
    }
    size_t peek(uint8_t *dst, size_t n) {
        fill(n);
        const size_t k = std::min(n, end_ - beg_);
        memcpy(dst, buf_.data() + beg_, k);
        return k;
    }
    size_t read(uint8_t *dst, size_t n) {
        size_t got = 0;
        while (got < n) {
            if (beg_ == end_) {
                beg_ = end_ = 0;
                if (n - got >= buf_.size()) {
                    const size_t k = fread(dst + got, 1, n - got, f_);
                    got += k;
                    if (k == 0) break;
                    continue;
                }
                end_ = fread(buf_.data(), 1, buf_.size(), f_);
                if (end_ == 0) break;
            }
            const size_t k = std::min(n - got, end_ - beg_);
            memcpy(dst + got, buf_.data() + beg_, k);
            beg_ += k;
            got += k;
        }
        return got;
    }
    // next text line without the newline; false at EOF
    bool getline(std::string &s) {
        s.clear();
        for (;;) {
            if (beg_ == end_) {
                beg_ = 0;
                end_ = fread(buf_.data(), 1, buf_.size(), f_);
                if (end_ == 0) return !s.empty();
            }
            const uint8_t *p = (const uint8_t *)memchr(buf_.data() + beg_, '\n', end_ - beg_);
            if (p) {
                s.append((const char *)buf_.data() + beg_, (size_t)(p - (buf_.data() + beg_)));
                beg_ = (size_t)(p - buf_.data()) + 1;
                if (!s.empty() && s.back() == '\r') s.pop_back();
                return true;
            }
            s.append((const char *)buf_.data() + beg_, end_ - beg_);
            beg_ = end_;
        }
    }

private:
    void fill(size_t n) {
        if (end_ - beg_ >= n) return;
        memmove(buf_.data(), buf_.data() + beg_, end_ - beg_);
        end_ -= beg_;
        beg_ = 0;
        while (end_ < n) {
            const size_t k = fread(buf_.data() + end_, 1, buf_.size() - end_, f_);
            if (!k) break;
            end_ += k;
        }
    }
    FILE *f_;
    std::vector<uint8_t> buf_;
    size_t beg_ = 0, end_ = 0;
};

// Growable byte buffer that does not value-initialise what it grows by (std::vector::resize zero-fills: 0.9 GB per
// million reads between the compressed and the inflated side of the BGZF reader, on its serial thread).
// Large buffers are recycled through a small free list: a batch's 75-300 MB of inflated records would otherwise be
// mapped, page-faulted (by the inflating threads), grown by copying and unmapped once per batch.
class RawBuf {
public:
    RawBuf() = default;
    RawBuf(const RawBuf &) = delete;
    RawBuf &operator=(const RawBuf &) = delete;
    ~RawBuf() {
        if (p_ && cap_ >= kPoolMin) {
            std::lock_guard<std::mutex> l(pool_mutex());
            auto &fl = pool_list();
            if (fl.size() < kPoolMax) {
                fl.push_back({p_, cap_});
                return;
            }
        }
        free(p_);
    }
    // capacity for n bytes without changing the size (a recycled buffer if one is large enough)
    void reserve(size_t n) {
        if (n <= cap_) return;
        if (!p_ && n >= kPoolMin) {
            std::lock_guard<std::mutex> l(pool_mutex());
            auto &fl = pool_list();
            size_t best = fl.size();
            for (size_t k = 0; k < fl.size(); k++)
                if (fl[k].second >= n && (best == fl.size() || fl[k].second < fl[best].second)) best = k;
            if (best < fl.size()) {
                p_ = fl[best].first;
                cap_ = fl[best].second;
                fl.erase(fl.begin() + (long)best);
                return;
            }
            // none large enough: map afresh (growing a recycled one means moving 100 MB of page tables: 10-25 ms) and
            // let the smallest go if the list is full
            if (fl.size() >= kPoolMax) {
                size_t small = 0;
                for (size_t k = 1; k < fl.size(); k++)
                    if (fl[k].second < fl[small].second) small = k;
                free(fl[small].first);
                fl.erase(fl.begin() + (long)small);
            }
        }
        const size_t c = std::max(n, cap_ + cap_ / 2);
        uint8_t *q = (uint8_t *)realloc(p_, c);
        if (!q) throw std::bad_alloc();
        p_ = q;
        cap_ = c;
        if (c >= kPoolMin) {
            // 2 MB pages where the kernel offers them on request: 512x fewer faults for the threads that fill the buffer
            // and as many fewer pages to give back
            const uintptr_t lo = ((uintptr_t)q + 4095) & ~(uintptr_t)4095, hi = ((uintptr_t)q + c) & ~(uintptr_t)4095;
            if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
        }
    }
    size_t capacity() const { return cap_; }
    uint8_t *data() { return p_; }
    const uint8_t *data() const { return p_; }
    size_t size() const { return n_; }
    void clear() { n_ = 0; }
    void resize(size_t n) {
        if (n > cap_) reserve(n_ ? std::max(n, cap_ + cap_ / 2) : n);
        n_ = n;
    }
    void drop_front(size_t k) {  // keep [k, size)
        if (k) memmove(p_, p_ + k, n_ - k);
        n_ -= k;
    }

private:
    static constexpr size_t kPoolMin = (size_t)8 << 20, kPoolMax = 12;
    static std::mutex &pool_mutex() {
        static auto *m = new std::mutex();
        return *m;
    }
    static std::vector<std::pair<uint8_t *, size_t>> &pool_list() {
        // never destroyed: a RawBuf may outlive every other static, and what the list holds at exit goes back with the process
        static auto *v = new std::vector<std::pair<uint8_t *, size_t>>();
        return *v;
    }
    uint8_t *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};

// wall time of the BAM reader's serial thread by step (annotate --timing)
struct ReadProf {
    double wait_io = 0, scan = 0, inflate = 0, frame = 0, carry = 0, layout = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};
inline ReadProf &read_prof() {
    static ReadProf p;
    return p;
}

// Inflates BGZF blocks in parallel batches and serves the uncompressed byte stream, either copied out (read) or
// in place (data / avail / consume / more) so that the BAM reader can frame records without a per-record copy loop.
// One lane's share of a BAM file (`fade annotate --gpus N`: N readers on disjoint BGZF virtual-offset ranges).
// A lane starts at the record at (coff_start, first_rec) — file offset of a BGZF block, offset in that block's inflated
// bytes — and ends exactly in front of the record at (coff_end, end_rec), where the next lane starts; coff_end = 0: at the
// end of the file.  The record chain of a lane must land exactly on its end: a lane whose neighbour guessed its start wrong
// fails with "truncated BAM record" (the driver then falls back to one lane), it never emits a damaged record.
struct LaneRange {
    bool on = false;
    uint64_t coff_start = 0, first_rec = 0, coff_end = 0, end_rec = 0;
};

class BgzfIn {
public:
    BgzfIn(ByteSource *src, Pool *pool) : src_(src), pool_(pool) {}
    size_t read(uint8_t *dst, size_t n) {
        size_t got = 0;
        while (got < n) {
            if (pos_ == out_.size()) {
                if (!more()) break;
            }
            const size_t k = std::min(n - got, out_.size() - pos_);
            memcpy(dst + got, out_.data() + pos_, k);
            pos_ += k;
            got += k;
        }
        return got;
    }
    const uint8_t *data() const { return out_.data() + pos_; }
    size_t avail() const { return out_.size() - pos_; }
    void consume(size_t n) { pos_ += n; }
    // inflate the next batch of blocks behind the unread bytes; false at end of file
    bool more() {
        if (pos_) {
            out_.drop_front(pos_);
            pos_ = 0;
        }
        return inflate_append(out_);
    }
    // hands the bytes inflated but not yet read over to `dst` (appended)
    void take_rest(RawBuf &dst) {
        const size_t k = out_.size() - pos_, o = dst.size();
        dst.resize(o + k);
        if (k) memcpy(dst.data() + o, out_.data() + pos_, k);
        out_.clear();
        pos_ = 0;
    }
    // Inflates the next blocks onto the end of `dst` (the reader's own buffer or a RecordBlock's): whole blocks of the
    // current gulp until about `want` bytes have come out (all of the gulp's by default).  false at end of file.
    bool inflate_append(RawBuf &dst, size_t want = (size_t)-1) {
        while (blk_next_ == offs_.size())
            if (!open_gulp()) return false;
        const RawBuf &comp_ = cbuf_[cur_ ^ 1];  // (open_gulp flipped cur_: the gulp being served is the other buffer)
        const double t1 = ReadProf::now();
        const size_t first = blk_next_, base = dst.size();
        std::vector<size_t> isz, ooff(1, 0);
        size_t last = first;
        while (last < offs_.size() && ooff.back() < want) {
            uint32_t v;
            memcpy(&v, comp_.data() + offs_[last].off + offs_[last].size - 4, 4);
            if (v > 65536) throw std::runtime_error("corrupt BGZF block (ISIZE beyond 64 KiB)");  // the format's limit: 512 blocks cannot ask for more than 32 MiB
            isz.push_back(v);
            ooff.push_back(ooff.back() + v);
            last++;
        }
        blk_next_ = last;
        dst.resize(base + ooff.back());
        const double t0 = ReadProf::now();
        read_prof().scan += t0 - t1;
        // two blocks per task: inflate_fast.hpp decodes a pair in lockstep (their Huffman look-up chains overlap)
        const size_t m = last - first;
        pool_->parallel_for((m + 1) / 2, [&](size_t t) {
            const uint8_t *src[2] = {nullptr, nullptr};
            size_t slen[2] = {0, 0}, jj[2] = {0, 0};
            int n = 0;
            for (size_t j = 2 * t; j < std::min(m, 2 * t + 2); j++) {
                const size_t k = first + j;
                const size_t hl = 12 + offs_[k].xlen;
                if (offs_[k].size < hl + 8) { bad_ = true; return; }
                if (isz[j] == 0) {
                    if (!bgzf_empty_member_ok(comp_.data() + offs_[k].off + hl, offs_[k].size - hl - 8, comp_.data() + offs_[k].off + offs_[k].size - 8)) bad_ = true;
                    continue;
                }
                src[n] = comp_.data() + offs_[k].off + hl;
                slen[n] = offs_[k].size - hl - 8;
                jj[n] = j;
                n++;
            }
            if (!bgzf_use_zlib()) {
                static thread_local std::unique_ptr<FastInflate> fi[2];
                if (!fi[0]) { fi[0].reset(new FastInflate()); fi[1].reset(new FastInflate()); }
                if (n == 2) {
                    if (!FastInflate::inflate2(*fi[0], src[0], slen[0], dst.data() + base + ooff[jj[0]], isz[jj[0]],
                                               *fi[1], src[1], slen[1], dst.data() + base + ooff[jj[1]], isz[jj[1]]))
                        bad_ = true;
                } else if (n == 1) {
                    if (!fi[0]->inflate(src[0], slen[0], dst.data() + base + ooff[jj[0]], isz[jj[0]])) bad_ = true;
                }
            } else {
                for (int q = 0; q < n; q++) {
                    z_stream zs;
                    memset(&zs, 0, sizeof zs);
                    if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("inflateInit2");
                    zs.next_in = const_cast<uint8_t *>(src[q]);
                    zs.avail_in = (uInt)slen[q];
                    zs.next_out = dst.data() + base + ooff[jj[q]];
                    zs.avail_out = (uInt)isz[jj[q]];
                    const int rc = inflate(&zs, Z_FINISH);
                    inflateEnd(&zs);
                    if (rc != Z_STREAM_END || zs.total_out != isz[jj[q]]) bad_ = true;
                }
            }
            // the blocks' CRC32 (RFC 1952 trailer), as htslib checks it
            for (int q = 0; q < n; q++) {
                const size_t k = first + jj[q];
                uint32_t crc;
                memcpy(&crc, comp_.data() + offs_[k].off + offs_[k].size - 8, 4);
                if (crc32_fast(0, dst.data() + base + ooff[jj[q]], isz[jj[q]]) != crc) bad_ = true;
            }
        }, CPU_INFLATE);
        read_prof().inflate += ReadProf::now() - t0;
        if (bad_) throw std::runtime_error("BGZF block does not inflate to its ISIZE / CRC32 (corrupt input)");
        // a lane's last block is served up to the next lane's first record only
        if (trunc_blk_ != (size_t)-1 && trunc_blk_ >= first && trunc_blk_ < last) {
            const size_t j = trunc_blk_ - first;
            if (lim_rec_ > isz[j]) throw std::runtime_error("lane range ends beyond its last block");
            dst.resize(base + ooff[j] + (size_t)lim_rec_);
        }
        return true;
    }

    ~BgzfIn() {
        if (pending_.valid()) pending_.wait();
    }
    // From here on serve only the blocks of a lane's range (the header has been read from the start of the file): blocks
    // at file offsets [coff_start, coff_end), plus the first end_rec inflated bytes of the block at coff_end.
    void restrict(uint64_t coff_start, uint64_t coff_end, uint64_t end_rec) {
        if (pending_.valid()) pending_.wait();
        src_->seek(coff_start);
        cbuf_[0].clear();
        cbuf_[1].clear();
        cbase_[0] = cbase_[1] = coff_start;
        out_.clear();
        pos_ = 0;
        cur_ = 0;
        primed_ = false;
        eof_ = false;
        offs_.clear();
        blk_next_ = 0;
        lim_coff_ = coff_end;
        lim_rec_ = end_rec;
        limited_ = coff_end != 0;
        trunc_blk_ = (size_t)-1;
    }

private:
    struct Blk { size_t off, size, xlen; };
    // Makes the next gulp of compressed bytes the current one: waits for its read, finds its whole blocks (offs_), moves
    // the incomplete block at its end to the other buffer and starts reading the following gulp behind it.  The
    // compressed side is double-buffered: while the blocks of one gulp inflate, a helper thread reads the next.
    // false at end of file.
    bool open_gulp() {
        for (;;) {
            RawBuf &comp_ = cbuf_[cur_];
            const double t0 = ReadProf::now();
            if (!primed_) {
                comp_.clear();
                fetch(comp_);
                primed_ = true;
            } else if (pending_.valid()) {
                pending_.get();
            }
            const double t1 = ReadProf::now();
            read_prof().wait_io += t1 - t0;
            offs_.clear();
            blk_next_ = 0;
            trunc_blk_ = (size_t)-1;
            size_t o = 0;
            const size_t have = comp_.size();
            const uint64_t gbase = cbase_[cur_];  // file offset of this buffer's first byte
            bool lane_done = false;
            while (o + 18 <= have) {
                if (limited_ && (gbase + o > lim_coff_ || (gbase + o == lim_coff_ && lim_rec_ == 0))) { lane_done = true; break; }
                const uint8_t *h = comp_.data() + o;
                if (h[0] != 0x1f || h[1] != 0x8b || !(h[3] & 4)) throw std::runtime_error("not a BGZF block");
                uint16_t xlen;
                memcpy(&xlen, h + 10, 2);
                if (xlen < 6 || h[12] != 'B' || h[13] != 'C') throw std::runtime_error("BGZF block lacks the BC subfield first");
                uint16_t bs;
                memcpy(&bs, h + 16, 2);
                const size_t bsize = (size_t)bs + 1;
                // header (12 + XLEN) + at least an empty deflate stream + CRC32 + ISIZE: a smaller BSIZE would put the
                // trailer reads in front of the block
                if (bsize < 12 + (size_t)xlen + 8) throw std::runtime_error("corrupt BGZF block (BSIZE smaller than its own header and trailer)");
                if (o + bsize > have) break;
                offs_.push_back({o, bsize, (size_t)xlen});
                if (limited_ && gbase + o == lim_coff_) {  // the block the next lane starts in: its first lim_rec_ bytes, then the lane is over
                    trunc_blk_ = offs_.size() - 1;
                    o += bsize;
                    lane_done = true;
                    break;
                }
                o += bsize;
            }
            if (lane_done) {
                if (pending_.valid()) pending_.wait();  // (none is in flight here: the fetch of the next gulp starts below)
                eof_ = true;
                cur_ ^= 1;
                cbuf_[cur_].clear();
                read_prof().scan += ReadProf::now() - t1;
                if (!offs_.empty()) return true;
                return false;
            }
            if (offs_.empty()) {
                if (have && eof_) throw std::runtime_error("truncated BGZF block");
                if (eof_) return false;
            }
            // the incomplete block at the end opens the other buffer, and the next gulp is read behind it meanwhile
            RawBuf &next = cbuf_[cur_ ^ 1];
            cbase_[cur_ ^ 1] = gbase + o;
            next.resize(have - o);
            if (have - o) memcpy(next.data(), comp_.data() + o, have - o);
            if (!eof_) pending_ = std::async(std::launch::async, [this, &next] { fetch(next); });
            cur_ ^= 1;
            read_prof().scan += ReadProf::now() - t1;
            if (!offs_.empty()) return true;  // (a gulp that ended inside its first block: take the next; cannot repeat, a gulp is 16 MiB)
        }
    }
    // appends up to a gulp of file bytes to b; sets eof_ when the file ran out
    void fetch(RawBuf &b) {
        const size_t kGulp = (size_t)16 << 20, o = b.size();
        b.resize(o + kGulp);
        const size_t got = src_->read(b.data() + o, kGulp);
        b.resize(o + got);
        if (got < kGulp) eof_ = true;
    }
    ByteSource *src_;
    Pool *pool_;
    RawBuf cbuf_[2], out_;
    uint64_t cbase_[2] = {0, 0};  // file offset of each compressed buffer's first byte
    bool limited_ = false;        // restrict(): the lane's end
    uint64_t lim_coff_ = 0, lim_rec_ = 0;
    size_t trunc_blk_ = (size_t)-1;  // index in offs_ of the block served only up to lim_rec_
    int cur_ = 0;
    bool primed_ = false;
    std::atomic<bool> eof_{false};
    std::future<void> pending_;
    std::vector<Blk> offs_;  // the whole blocks of the gulp being served
    size_t blk_next_ = 0;    // ... and the first of them not yet inflated
    size_t pos_ = 0;
    std::atomic<bool> bad_{false};
};

// A batch of BAM records framed in place in the inflated bytes they arrived in: no per-record copy or allocation.
struct RecordBlock {
    RawBuf buf;
    std::vector<uint32_t> off, len;  // record i = buf[off[i], off[i] + len[i])  (the bytes after its block_size)
    size_t size() const { return off.size(); }
    RecView view(size_t i) const { return RecView(buf.data() + off[i], len[i]); }
    uint8_t *mut(size_t i) { return buf.data() + off[i]; }
};

// ------------------------------------------------------------------ reader (SAM text or BAM)
class Reader {
public:
    Reader(const std::string &path, Pool *pool) : pool_(pool) {
        f_ = path == "-" ? stdin : fopen(path.c_str(), "rb");
        if (!f_) throw std::runtime_error("cannot open " + path);
        src_.reset(new ByteSource(f_));
        uint8_t m[4] = {0, 0, 0, 0};
        src_->peek(m, 4);
        if (m[0] == 0x1f && m[1] == 0x8b) {
            bam_ = true;
            bgzf_.reset(new BgzfIn(src_.get(), pool_));
            read_bam_header();
        } else {
            read_sam_header();
        }
    }
    ~Reader() {
        if (f_ && f_ != stdin) fclose(f_);
    }
    const Header &header() const { return hdr_; }
    bool is_bam() const { return bam_; }
    // the next up-to-max_n records as one block: BAM records framed in place in the inflated bytes, SAM lines parsed into
    // the same layout; returns their number (0 at end of file)
    size_t read_block(RecordBlock &blk, size_t max_n) {
        blk.buf.clear();
        blk.off.clear();
        blk.len.clear();
        if (!bam_) return read_block_sam(blk, max_n);
        blk.off.reserve(max_n);
        blk.len.reserve(max_n);
        if (!block_mode_) {  // bytes inflated while the header was read
            bgzf_->take_rest(carry_);
            block_mode_ = true;
        }
        const double tc0 = ReadProf::now();
        // room for what the previous batch took (+ a gulp): the buffer is then recycled or mapped once, not grown by copying
        // (rounded up generously: batches of a file differ a little in size, and a recycled buffer must not need growing)
        const size_t guess = last_block_recs_ ? last_block_bytes_ + last_block_bytes_ / 4 : max_n * 400;
        blk.buf.reserve(std::max(carry_.size(), ((guess >> 25) + 2) << 25));
        blk.buf.resize(carry_.size());
        if (carry_.size()) memcpy(blk.buf.data(), carry_.data(), carry_.size());
        carry_.clear();
        read_prof().carry += ReadProf::now() - tc0;
        size_t pos = 0;
        for (;;) {
            const uint8_t *b = blk.buf.data();
            const size_t av = blk.buf.size();
            const double tf = ReadProf::now();
            while (blk.off.size() < max_n && pos + 4 <= av) {
                uint32_t bs;
                memcpy(&bs, b + pos, 4);
                if (bs < 32) throw std::runtime_error("corrupt BAM record");
                if (pos + 4 + (size_t)bs > av) break;
                if (pos + 4 > 0xffffffffull) throw std::runtime_error("record block beyond 4 GiB: lower --batch");
                blk.off.push_back((uint32_t)(pos + 4));
                blk.len.push_back(bs);
                pos += 4 + (size_t)bs;
                // The chain of block_size fields is serial and every link sits in a line another core just wrote (the
                // inflater): ~50 ns each.  Records of a file are of similar size, so the links 6 and 7 records ahead
                // are near pos + 6 * (4 + bs): fetch those lines now.
                const size_t ahead = pos + 6 * (4 + (size_t)bs);
                if (ahead + 384 < av) {
                    __builtin_prefetch(b + ahead);
                    __builtin_prefetch(b + ahead + 64);
                    __builtin_prefetch(b + ahead + 4 + (size_t)bs);
                    __builtin_prefetch(b + ahead + 4 + (size_t)bs + 64);
                }
            }
            read_prof().frame += ReadProf::now() - tf;
            if (blk.off.size() == max_n) break;
            // about as many bytes as the records still missing take (so that little is inflated beyond this batch and has to
            // be carried over), in pieces large enough to keep every inflating thread busy
            const size_t have_n = blk.off.size();
            const size_t per_rec = have_n ? pos / have_n + 1 : (last_block_recs_ ? last_block_bytes_ / last_block_recs_ + 1 : 512);
            const size_t want = std::max<size_t>((max_n - have_n) * per_rec + ((size_t)64 << 10), (size_t)2 << 20);
            if (!bgzf_->inflate_append(blk.buf, want)) {
                if (pos < blk.buf.size()) throw std::runtime_error("truncated BAM record");
                break;
            }
            if (skip_front_) {  // a lane's first block: what lies in front of its first record belongs to the lane before
                if (skip_front_ > blk.buf.size()) throw std::runtime_error("lane range starts beyond its first block");
                blk.buf.drop_front(skip_front_);
                skip_front_ = 0;
            }
        }
        // what follows the last framed record belongs to the next block
        const size_t rest = blk.buf.size() - pos;
        const double tc = ReadProf::now();
        carry_.resize(rest);
        if (rest) memcpy(carry_.data(), blk.buf.data() + pos, rest);
        blk.buf.resize(pos);
        last_block_bytes_ = std::max(pos, (size_t)1 << 20);
        last_block_recs_ = blk.off.size();
        const double tl = ReadProf::now();
        read_prof().carry += tl - tc;
        const size_t cnt = blk.off.size(), nt = (size_t)pool_->size() * 4;
        pool_->parallel_for(nt, [&](size_t t) {
            for (size_t i = cnt * t / nt; i < cnt * (t + 1) / nt; i++)
                if (!blk.view(i).layout_ok()) bad_layout_ = true;
        }, CPU_FRAME);
        read_prof().layout += ReadProf::now() - tl;
        if (bad_layout_) throw std::runtime_error("corrupt BAM record (field lengths exceed the record)");
        return cnt;
    }
    // reads up to max_n records into out (appending); returns number read
    size_t read_chunk(std::vector<Rec> &out, size_t max_n) {
        if (block_mode_) throw std::runtime_error("read_chunk after read_block");
        size_t n = 0;
        if (bam_) {
            // frame the records in the inflated buffer (serial, 4 bytes per record), copy them out in parallel
            std::vector<size_t> off;
            while (n < max_n) {
                const uint8_t *b = bgzf_->data();
                const size_t av = bgzf_->avail();
                off.clear();
                size_t o = 0;
                while (n + off.size() < max_n && o + 4 <= av) {
                    uint32_t bs;
                    memcpy(&bs, b + o, 4);
                    if (bs < 32) throw std::runtime_error("corrupt BAM record");
                    if (o + 4 + bs > av) break;
                    off.push_back(o);
                    o += 4 + (size_t)bs;
                }
                if (off.empty()) {
                    if (bgzf_->more()) continue;
                    if (av) throw std::runtime_error("truncated BAM record");
                    break;
                }
                const size_t base = out.size();
                out.resize(base + off.size());
                const size_t nt = (size_t)pool_->size() * 4, cnt = off.size();
                pool_->parallel_for(nt, [&](size_t t) {
                    for (size_t i = cnt * t / nt; i < cnt * (t + 1) / nt; i++) {
                        uint32_t bs;
                        memcpy(&bs, b + off[i], 4);
                        out[base + i].d.assign(b + off[i] + 4, b + off[i] + 4 + bs);
                        if (!out[base + i].layout_ok()) bad_layout_ = true;
                    }
                }, CPU_FRAME);
                if (bad_layout_) throw std::runtime_error("corrupt BAM record (field lengths exceed the record)");
                bgzf_->consume(o);
                n += cnt;
            }
            return n;
        }
        // SAM: pull text in 16 MiB pieces, frame the lines (memchr), parse them in parallel straight from the buffer
        std::vector<std::pair<size_t, size_t>> lines;  // offset, length in text_
        frame_sam_lines(max_n, lines);
        const size_t base = out.size();
        out.resize(base + lines.size());
        std::atomic<bool> bad{false};
        std::string err;
        std::mutex em;
        const size_t nt = (size_t)pool_->size() * 4, cnt = lines.size();
        pool_->parallel_for(nt, [&](size_t t) {
            for (size_t i = cnt * t / nt; i < cnt * (t + 1) / nt; i++) {
                try {
                    sam_parse((const char *)text_.data() + lines[i].first, lines[i].second, hdr_, out[base + i]);
                } catch (const std::exception &e) {
                    std::lock_guard<std::mutex> l(em);
                    bad = true;
                    err = e.what();
                }
            }
        }, CPU_SAM_PARSE);
        if (bad) throw std::runtime_error(err);
        return lines.size();
    }

private:
    // the next up-to-max_n complete SAM lines of the text buffer (offset, length in text_), reading more text as needed
    void frame_sam_lines(size_t max_n, std::vector<std::pair<size_t, size_t>> &lines) {
        for (;;) {
            lines.clear();
            size_t o = text_pos_;
            while (lines.size() < max_n) {
                const char *b = (const char *)text_.data() + o;
                const char *nl = (const char *)memchr(b, '\n', text_.size() - o);
                if (!nl) break;
                size_t len = (size_t)(nl - b);
                const size_t next = o + len + 1;
                if (len && b[len - 1] == '\r') len--;
                if (len) lines.push_back({o, len});
                o = next;
            }
            if (lines.size() < max_n && !text_eof_) {
                // not enough complete lines buffered: drop the consumed prefix, append more text, frame again
                text_.drop_front(text_pos_);
                text_pos_ = 0;
                const size_t have = text_.size(), want = (size_t)16 << 20;
                text_.resize(have + want);  // (not zero-filled: a RawBuf)
                const size_t got = src_->read(text_.data() + have, want);
                text_.resize(have + got);
                if (got < want) {
                    text_eof_ = true;
                    if (text_.size() && text_.data()[text_.size() - 1] != '\n') {
                        text_.resize(text_.size() + 1);
                        text_.data()[text_.size() - 1] = '\n';
                    }
                }
                continue;
            }
            text_pos_ = o;
            break;
        }
    }
    // SAM input as a RecordBlock: every task parses its share of the lines into one buffer of its own (a reused Rec, no
    // allocation per record), then the pieces are laid end to end in blk.buf
    size_t read_block_sam(RecordBlock &blk, size_t max_n) {
        std::vector<std::pair<size_t, size_t>> lines;
        frame_sam_lines(max_n, lines);
        const size_t cnt = lines.size(), nt = (size_t)pool_->size() * 4;
        if (!cnt) return 0;
        std::vector<std::vector<uint8_t>> piece(nt);
        std::vector<std::vector<uint32_t>> plen(nt);
        std::atomic<bool> bad{false};
        std::string err;
        std::mutex em;
        pool_->parallel_for(nt, [&](size_t t) {
            const size_t lo = cnt * t / nt, hi = cnt * (t + 1) / nt;
            Rec r;
            piece[t].reserve((hi - lo) * 320);
            plen[t].reserve(hi - lo);
            try {
                for (size_t i = lo; i < hi; i++) {
                    sam_parse((const char *)text_.data() + lines[i].first, lines[i].second, hdr_, r);
                    if (!r.layout_ok()) throw std::runtime_error("corrupt SAM record (field lengths exceed the record)");  // as for BAM input
                    piece[t].insert(piece[t].end(), r.d.begin(), r.d.end());
                    plen[t].push_back((uint32_t)r.d.size());
                }
            } catch (const std::exception &e) {
                std::lock_guard<std::mutex> l(em);
                bad = true;
                err = e.what();
            }
        }, CPU_SAM_PARSE);
        if (bad) throw std::runtime_error(err);
        std::vector<size_t> base(nt + 1, 0);
        for (size_t t = 0; t < nt; t++) base[t + 1] = base[t] + piece[t].size();
        if (base[nt] > 0xffffffffull) throw std::runtime_error("record block beyond 4 GiB: lower --batch");
        blk.buf.reserve(base[nt]);
        blk.buf.resize(base[nt]);
        blk.off.resize(cnt);
        blk.len.resize(cnt);
        pool_->parallel_for(nt, [&](size_t t) {
            if (!piece[t].empty()) memcpy(blk.buf.data() + base[t], piece[t].data(), piece[t].size());
            size_t o = base[t], i = cnt * t / nt;
            for (uint32_t l : plen[t]) {
                blk.off[i] = (uint32_t)o;
                blk.len[i] = l;
                o += l;
                i++;
            }
        }, CPU_COPY);
        return cnt;
    }
    void read_sam_header() {
        std::string s;
        while (src_->getline(s)) {
            if (!s.empty() && s[0] == '@') hdr_.text += s + "\n";
            else {
                text_.resize(s.size() + 1);  // first record line: goes back in front of the text buffer
                memcpy(text_.data(), s.data(), s.size());
                text_.data()[s.size()] = '\n';
                break;
            }
        }
        hdr_.parse_sq_from_text();
    }
public:
    // Serve only a lane's records from here on (BAM input from a regular file; the header has been read).
    size_t bam_header_bytes() const { return hdr_bytes_; }
    void restrict_to(const LaneRange &r) {
        if (!bam_) throw std::runtime_error("lanes need BAM input");
        carry_.clear();
        bgzf_->restrict(r.coff_start, r.coff_end, r.end_rec);  // (lane 0 too: it re-enters at the block its first record starts in)
        skip_front_ = (size_t)r.first_rec;
    }

private:
    void read_bam_header() {
        uint8_t m[4];
        if (bgzf_->read(m, 4) != 4 || memcmp(m, "BAM\1", 4) != 0) throw std::runtime_error("not a BAM file");
        int32_t lt;
        bgzf_->read((uint8_t *)&lt, 4);
        hdr_.text.resize((size_t)lt);
        if (lt) bgzf_->read((uint8_t *)&hdr_.text[0], (size_t)lt);
        while (!hdr_.text.empty() && hdr_.text.back() == '\0') hdr_.text.pop_back();
        int32_t nref;
        bgzf_->read((uint8_t *)&nref, 4);
        hdr_bytes_ = 12 + (size_t)lt;
        for (int k = 0; k < nref; k++) {
            int32_t ln;
            bgzf_->read((uint8_t *)&ln, 4);
            std::string nm((size_t)ln, '\0');
            bgzf_->read((uint8_t *)&nm[0], (size_t)ln);
            while (!nm.empty() && nm.back() == '\0') nm.pop_back();
            int32_t l;
            bgzf_->read((uint8_t *)&l, 4);
            hdr_.names.push_back(nm);
            hdr_.lens.push_back(l);
            hdr_bytes_ += 8 + (size_t)ln;
        }
    }
    size_t hdr_bytes_ = 0;  // inflated bytes of the BAM header (the first record starts behind them)
    Pool *pool_;
    FILE *f_ = nullptr;
    std::unique_ptr<ByteSource> src_;
    std::unique_ptr<BgzfIn> bgzf_;
    bool bam_ = false;
    Header hdr_;
    std::atomic<bool> bad_layout_{false};
    RawBuf carry_;          // read_block: inflated bytes behind the last framed record
    size_t skip_front_ = 0; // restrict_to(): inflated bytes in front of the lane's first record, dropped from the first block read
    bool block_mode_ = false;
    size_t last_block_bytes_ = 0, last_block_recs_ = 0;  // inflated size / records of the previous read_block (the next one reserves as much)
    RawBuf text_;           // SAM text not yet handed out (complete lines from text_pos_ on)
    size_t text_pos_ = 0;
    bool text_eof_ = false;
};

// ------------------------------------------------------------------ writer: SAM / uBAM / BAM to a FILE*
enum class OutFmt { SAM = 0, UBAM = 1, BAM = 2 };  // `con` of util.d:65-76

// Output leaves through one I/O thread: the stage that formats / compresses batch k does not wait for the fwrite of
// batch k - 1 (1.6 GB per 10 M reads of BAM).
class OutThread {
public:
    explicit OutThread(FILE *f) : f_(f), th_([this] { run(); }) {}
    ~OutThread() { finish(); }
    void put(std::vector<std::vector<uint8_t>> &&bufs) {
        std::unique_lock<std::mutex> l(m_);
        room_.wait(l, [&] { return q_.size() < 4; });
        q_.push_back(std::move(bufs));
        work_.notify_one();
    }
    void put(std::vector<std::string> &&parts) {
        std::vector<std::vector<uint8_t>> b(parts.size());
        for (size_t k = 0; k < parts.size(); k++) b[k].assign(parts[k].begin(), parts[k].end());
        put(std::move(b));
    }
    // everything handed over so far is in the FILE; throws if a write failed
    void finish() {
        {
            std::unique_lock<std::mutex> l(m_);
            if (done_) return;
            done_ = true;
            work_.notify_one();
        }
        if (th_.joinable()) th_.join();
        fflush(f_);
    }
    bool failed() const { return failed_; }
    double busy_seconds() const { return busy_; }  // inside fwrite (valid after finish())

private:
    void run() {
        for (;;) {
            std::vector<std::vector<uint8_t>> b;
            {
                std::unique_lock<std::mutex> l(m_);
                work_.wait(l, [&] { return !q_.empty() || done_; });
                if (q_.empty()) return;
                b = std::move(q_.front());
                q_.pop_front();
                room_.notify_one();
            }
            const double t0 = ReadProf::now();
            for (auto &o : b)
                if (!o.empty() && fwrite(o.data(), 1, o.size(), f_) != o.size()) failed_ = true;
            busy_ += ReadProf::now() - t0;
        }
    }
    FILE *f_;
    std::mutex m_;
    std::condition_variable work_, room_;
    std::deque<std::vector<std::vector<uint8_t>>> q_;
    bool done_ = false;
    std::atomic<bool> failed_{false};
    double busy_ = 0;
    std::thread th_;
};

// A BGZF compressor on a device (the `fade` driver hands the writer one: fadehip_bgzf_deflate_* of include/fadehip.h).
// This layer knows nothing of HIP; without a device codec the blocks are compressed on the pool (deflate_fast.hpp / zlib).
// Lanes alternate: while one lane's bytes are compressed on the device, the next flush is staged into the other's buffer.
struct BgzfDevice {
    virtual ~BgzfDevice() {}
    virtual int lanes() const = 0;
    virtual uint8_t *stage(int lane, size_t bytes) = 0;  // a pinned buffer of at least `bytes` for the lane's next submit
    virtual void submit(int lane, size_t bytes) = 0;     // compress stage(lane)[0, bytes) into BGZF members; returns at once
    virtual void wait(int lane, const uint8_t **out, size_t *n) = 0;  // the members' bytes, valid until the lane's next submit
};

class Writer {
public:
    // with_header / with_eof = false: a lane of a multi-lane run writes records only (the driver puts the pieces together)
    Writer(FILE *f, OutFmt fmt, const Header &h, Pool *pool, BgzfDevice *dev = nullptr, bool with_header = true, bool with_eof = true)
        : f_(f), fmt_(fmt), hdr_(h), pool_(pool), dev_(fmt == OutFmt::BAM ? dev : nullptr), with_eof_(with_eof), io_(f) {
        if (!with_header) return;
        if (fmt_ == OutFmt::SAM) {
            std::vector<std::string> one(1, hdr_.text);
            io_.put(std::move(one));
        } else {
            std::vector<uint8_t> b;
            auto put32 = [&](int32_t v) { b.insert(b.end(), (uint8_t *)&v, (uint8_t *)&v + 4); };
            b.insert(b.end(), {'B', 'A', 'M', 1});
            put32((int32_t)hdr_.text.size());
            b.insert(b.end(), hdr_.text.begin(), hdr_.text.end());
            put32((int32_t)hdr_.names.size());
            for (size_t k = 0; k < hdr_.names.size(); k++) {
                put32((int32_t)hdr_.names[k].size() + 1);
                b.insert(b.end(), hdr_.names[k].begin(), hdr_.names[k].end());
                b.push_back(0);
                put32((int32_t)hdr_.lens[k]);
            }
            raw_.resize(b.size());
            memcpy(raw_.data(), b.data(), b.size());
            flush_blocks(true);  // htslib flushes the header into its own block(s)
        }
    }
    void write(const std::vector<Rec> &recs) {
        if (fmt_ == OutFmt::SAM) {
            const size_t n = recs.size(), nt = (size_t)pool_->size();
            std::vector<std::string> parts(nt);
            pool_->parallel_for(nt, [&](size_t t) {
                const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
                std::string &s = parts[t];
                s.reserve((hi - lo) * 400);
                for (size_t i = lo; i < hi; i++) sam_format(recs[i], hdr_, s);
            }, CPU_SAM_FORMAT);
            io_.put(std::move(parts));
            return;
        }
        // serial prefix sum of the record sizes, parallel copy
        const size_t n = recs.size();
        std::vector<size_t> off(n + 1);
        off[0] = raw_.size();
        for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + 4 + recs[i].d.size();
        raw_.resize(off[n]);
        const size_t nt = (size_t)pool_->size() * 4;
        pool_->parallel_for(nt, [&](size_t t) {
            for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) {
                const uint32_t bs = (uint32_t)recs[i].d.size();
                memcpy(raw_.data() + off[i], &bs, 4);
                memcpy(raw_.data() + off[i] + 4, recs[i].d.data(), bs);
            }
        }, CPU_COPY);
        flush_blocks(false);
    }
    // What a stage adds to a RecordBlock's records: appended aux bytes per record (sfx_off[i] .. sfx_off[i + 1] of sfx),
    // and whole replacement records for the few that had to be rebuilt (`owned`, any order).
    struct BlockOut {
        std::vector<uint32_t> sfx_off;
        std::vector<uint8_t> sfx;
        std::vector<std::pair<uint32_t, Rec>> owned;
    };
    void write_block(const RecordBlock &blk, const BlockOut &o) {
        const size_t n = blk.size();
        std::vector<int32_t> own(o.owned.empty() ? 0 : n, -1);
        for (size_t k = 0; k < o.owned.size(); k++) own[o.owned[k].first] = (int32_t)k;
        auto sfx_len = [&](size_t i) { return o.sfx_off.empty() ? 0u : o.sfx_off[i + 1] - o.sfx_off[i]; };
        auto sfx_ptr = [&](size_t i) { return o.sfx.data() + (o.sfx_off.empty() ? 0u : o.sfx_off[i]); };
        if (fmt_ == OutFmt::SAM) {
            const size_t nt = (size_t)pool_->size();
            std::vector<std::string> parts(nt);
            pool_->parallel_for(nt, [&](size_t t) {
                const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
                std::string &s = parts[t];
                s.reserve((hi - lo) * 400);
                std::vector<uint8_t> tmp;
                for (size_t i = lo; i < hi; i++) {
                    if (!own.empty() && own[i] >= 0) sam_format(o.owned[(size_t)own[i]].second, hdr_, s);
                    else if (sfx_len(i) == 0) sam_format(blk.view(i), hdr_, s);
                    else {
                        tmp.resize((size_t)blk.len[i] + sfx_len(i));
                        memcpy(tmp.data(), blk.buf.data() + blk.off[i], blk.len[i]);
                        memcpy(tmp.data() + blk.len[i], sfx_ptr(i), sfx_len(i));
                        sam_format(RecView(tmp.data(), tmp.size()), hdr_, s);
                    }
                }
            }, CPU_SAM_FORMAT);
            io_.put(std::move(parts));
            return;
        }
        std::vector<size_t> off(n + 1);
        off[0] = raw_.size();
        for (size_t i = 0; i < n; i++)
            off[i + 1] = off[i] + 4 + ((!own.empty() && own[i] >= 0) ? o.owned[(size_t)own[i]].second.d.size() : (size_t)blk.len[i] + sfx_len(i));
        raw_.resize(off[n]);
        // what the compressor is told about the bytes (FastDeflate::Hint): a record's packed bases hold no repeats worth
        // probing for, its qualities rarely (runs), its tags and the next record's fixed fields do
        const size_t h0 = hints_.size();
        const bool hinted = fmt_ == OutFmt::BAM && !dev_ && off[n] < 0xffffffffull;
        if (hinted) hints_.resize(h0 + 3 * n);
        const size_t nt = (size_t)pool_->size() * 4;
        pool_->parallel_for(nt, [&](size_t t) {
            for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) {
                uint8_t *dst = raw_.data() + off[i];
                const uint32_t bs = (uint32_t)(off[i + 1] - off[i] - 4);
                memcpy(dst, &bs, 4);
                if (!own.empty() && own[i] >= 0) {
                    memcpy(dst + 4, o.owned[(size_t)own[i]].second.d.data(), bs);
                } else {
                    memcpy(dst + 4, blk.buf.data() + blk.off[i], blk.len[i]);
                    if (sfx_len(i)) memcpy(dst + 4 + blk.len[i], sfx_ptr(i), sfx_len(i));
                }
                if (hinted) {
                    const RecView v(dst + 4, bs);  // (layout checked when the record was read / built)
                    const size_t s0 = std::min<size_t>(v.seq_off(), bs), q0 = std::min<size_t>(v.qual_off(), bs), s1 = std::min<size_t>(v.aux_off(), bs);
                    hints_[h0 + 3 * i] = {(uint32_t)(off[i] + 4 + s0), FastDeflate::HINT_SKIP};      // packed bases: nothing to find
                    hints_[h0 + 3 * i + 1] = {(uint32_t)(off[i] + 4 + q0), FastDeflate::HINT_MILD};  // qualities: runs happen
                    hints_[h0 + 3 * i + 2] = {(uint32_t)(off[i] + 4 + s1), 0};                       // tags, then the next record's fixed fields
                }
            }
        }, CPU_COPY);
        flush_blocks(false);
    }
    double io_seconds() const { return io_.busy_seconds(); }
    void close() {
        if (closed_) return;
        closed_ = true;
        if (fmt_ != OutFmt::SAM) {
            flush_blocks(true);
            if (dev_)
                for (int k = 0; k < dev_->lanes(); k++) collect_lane((int)((dev_seq_ + (size_t)k) % (size_t)dev_->lanes()));  // oldest first
            if (with_eof_) {
                std::vector<std::vector<uint8_t>> eof(1, std::vector<uint8_t>(BGZF_EOF, BGZF_EOF + sizeof BGZF_EOF));
                io_.put(std::move(eof));
            }
        }
        io_.finish();
        if (io_.failed()) throw std::runtime_error("write error on the output stream");
    }

private:
    // the members a device lane has finished go to the output thread (copied on the pool: the lane's buffer is reused)
    void collect_lane(int lane) {
        if (dev_busy_.empty() || !dev_busy_[(size_t)lane]) return;
        const uint8_t *out = nullptr;
        size_t n = 0;
        dev_->wait(lane, &out, &n);
        dev_busy_[(size_t)lane] = false;
        const size_t nt = std::max<size_t>(1, std::min<size_t>((size_t)pool_->size(), n / (4u << 20) + 1));
        std::vector<std::vector<uint8_t>> outs(nt);
        pool_->parallel_for(nt, [&](size_t t) {
            const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
            outs[t].assign(out + lo, out + hi);
        }, CPU_COPY);
        io_.put(std::move(outs));
    }
    void flush_blocks_device(bool all) {
        const size_t B = 0xff00;  // (the device cuts a submission into blocks of this size or half of it)
        size_t used = all ? raw_.size() : raw_.size() / B * B;
        if (!used) return;
        if (dev_busy_.empty()) dev_busy_.assign((size_t)dev_->lanes(), false);
        const int lane = (int)(dev_seq_ % (size_t)dev_->lanes());
        collect_lane(lane);  // (the submission before last: its bytes leave before the lane's buffers are reused)
        uint8_t *st = dev_->stage(lane, used);
        const size_t nt = std::max<size_t>(1, std::min<size_t>((size_t)pool_->size() * 2, used / (1u << 20) + 1));
        pool_->parallel_for(nt, [&](size_t t) {
            const size_t lo = used * t / nt, hi = used * (t + 1) / nt;
            memcpy(st + lo, raw_.data() + lo, hi - lo);
        }, CPU_COPY);
        dev_->submit(lane, used);
        dev_busy_[(size_t)lane] = true;
        dev_seq_++;
        raw_.drop_front(used);
        hints_.clear();
    }
    void flush_blocks(bool all) {
        if (dev_) return flush_blocks_device(all);
        const size_t B = 0xff00;
        size_t nblk = raw_.size() / B;
        if (all && raw_.size() % B) nblk++;
        if (!nblk) return;
        std::vector<std::vector<uint8_t>> outs(nblk);
        const int level = fmt_ == OutFmt::UBAM ? 0 : 6;
        pool_->parallel_for(nblk, [&](size_t k) {
            const size_t o = k * B, n = std::min(B, raw_.size() - o);
            outs[k].reserve(n + 64);
            // the block's share of the layout hints, relative to the block; a block that begins inside a hinted stretch
            // opens with it
            static thread_local std::vector<FastDeflate::Hint> bh;
            bh.clear();
            if (!hints_.empty()) {
                auto it = std::lower_bound(hints_.begin(), hints_.end(), o, [](const FastDeflate::Hint &h, size_t v) { return h.pos < v; });
                if (it != hints_.begin() && std::prev(it)->miss) bh.push_back({0, std::prev(it)->miss});
                for (; it != hints_.end() && it->pos < o + n; ++it) bh.push_back({(uint32_t)(it->pos - o), it->miss});
            }
            bgzf_compress_block(raw_.data() + o, n, level, outs[k], bh.data(), bh.size());
        }, CPU_DEFLATE);
        io_.put(std::move(outs));
        const size_t used = std::min(raw_.size(), nblk * B);
        raw_.drop_front(used);
        // the hints follow the bytes that stay: positions move down by `used`; a stretch the cut went through reopens at 0
        if (!hints_.empty()) {
            auto it = std::lower_bound(hints_.begin(), hints_.end(), used, [](const FastDeflate::Hint &h, size_t v) { return h.pos < v; });
            const uint32_t inside = it != hints_.begin() ? std::prev(it)->miss : 0;
            std::vector<FastDeflate::Hint> rest;
            if (raw_.size()) {
                if (inside) rest.push_back({0, inside});
                for (; it != hints_.end(); ++it) rest.push_back({(uint32_t)(it->pos - used), it->miss});
            }
            hints_.swap(rest);
        }
    }
    FILE *f_;
    OutFmt fmt_;
    Header hdr_;
    Pool *pool_;
    BgzfDevice *dev_;
    bool with_eof_ = true;
    std::vector<bool> dev_busy_;
    size_t dev_seq_ = 0;  // flushes handed to the device so far
    RawBuf raw_;
    std::vector<FastDeflate::Hint> hints_;  // layout hints for the bytes in raw_ (positions relative to its start, sorted)
    bool closed_ = false;
    OutThread io_;  // last member: started after, and joined before, everything it writes from
};

// ------------------------------------------------------------------ FASTA
struct Fasta {
    std::vector<std::string> names;
    std::vector<std::string> seqs;  // residues as in the file (case preserved)
};

inline Fasta load_fasta(const std::string &path) {
    gzFile g = gzopen(path.c_str(), "rb");  // transparent for plain, gzip and bgzip files
    if (!g) throw std::runtime_error("cannot open " + path);
    gzbuffer(g, 1 << 20);
    Fasta fa;
    std::vector<char> buf(1 << 20);
    std::string carry;
    bool in_name = false;
    int n;
    while ((n = gzread(g, buf.data(), (unsigned)buf.size())) > 0) {
        const char *p = buf.data(), *e = p + n;
        while (p < e) {
            if (in_name) {
                const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
                carry.append(p, q ? q : e);
                if (!q) { p = e; break; }
                size_t w = carry.find_first_of(" \t\r");
                fa.names.push_back(carry.substr(0, w));
                fa.seqs.emplace_back();
                carry.clear();
                in_name = false;
                p = q + 1;
            } else if (*p == '>') {
                in_name = true;
                p++;
            } else {
                const char *q = (const char *)memchr(p, '\n', (size_t)(e - p));
                const char *stop = q ? q : e;
                // a '>' can only start a line, so anything up to the newline is sequence
                if (fa.seqs.empty()) throw std::runtime_error("FASTA does not start with '>'");
                const char *a = p;
                size_t len = (size_t)(stop - a);
                if (len && a[len - 1] == '\r') len--;
                fa.seqs.back().append(a, len);
                p = q ? q + 1 : e;
            }
        }
    }
    gzclose(g);
    if (fa.names.empty()) throw std::runtime_error("no sequences in " + path);
    return fa;
}

}  // namespace htsl
