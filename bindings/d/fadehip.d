/// extern(C) binding of include/fadehip.h (ABI version 3) for the D host of blachlylab/fade.
///
/// NOT COMPILED IN THIS REPOSITORY'S ENVIRONMENT: the build image has no D compiler (ldc2, dmd, gdc, dub are
/// all absent) and none of FADE's dependencies, so this file has never been compiled or run.  It mirrors
/// include/fadehip.h declaration by declaration; the same ABI is exercised from C++ (fade_amd/csrc/host) and
/// from Python/ctypes (fade_amd/_lib.py) by the test suite.
module fadehip;

extern (C) nothrow @nogc:

enum FADEHIP_ABI_VERSION = 3;
enum FADEHIP_MAX_OPS = 16;
enum FADEHIP_MAX_QUERY = 512;
enum FADEHIP_MAX_LONG_QUERY = 32768;
enum FADEHIP_NUM_SLOTS = 4;

enum : int
{
    FADEHIP_OK = 0,
    FADEHIP_E_INVALID = -1,
    FADEHIP_E_NODEVICE = -2,
    FADEHIP_E_HIP = -3,
    FADEHIP_E_NOMEM = -4,
    FADEHIP_E_UNSUPPORTED = -5,
    FADEHIP_E_STATE = -6,
    FADEHIP_E_RESIDUE = -7,
    FADEHIP_E_RCCL = -8
}

struct fadehip_ctx;

/// Parasail("ACTGN", open, ext, match, mismatch) -- source/anno.d:36
struct fadehip_params
{
    int open = 10;
    int ext = 2;
    int match = 2;
    int mismatch = -3;
    int max_ref_len = 1 << 20;
    int max_batch_reads = 1 << 20;
    long trace_bytes = 0;
    int trace_all = 0;
    uint rules = FADEHIP_RULES_DEFAULT; /// FADEHIP_RULE_* : the assumptions about libparasail that could not be checked
}

/// same bits as oracle/fade_oracle.h FO_RULE_* (SURVEY.md Appendix A)
enum : uint
{
    FADEHIP_RULE_END_MIN_REF_THEN_QUERY = 1u << 0,
    FADEHIP_RULE_HDIR_DIAG_F_E = 1u << 1,
    FADEHIP_RULE_GAP_TIE_EXTENDS = 1u << 2,
    FADEHIP_RULE_EQ_BY_CHAR = 1u << 3,
    FADEHIP_RULE_SAM_GAP_LETTERS = 1u << 4,
    FADEHIP_RULE_PAD_SOFTCLIP = 1u << 5,
    FADEHIP_RULE_N_MATCHES_N = 1u << 6,
    FADEHIP_RULES_DEFAULT = 0x7f
}

/// what source/analysis.d:69-113 reads from a dparasail result
struct fadehip_sw_result
{
    int score;
    int end_query, end_ref;
    int beg_query, beg_ref; /// beg_ref == res.position
    int n_ops;
    uint[FADEHIP_MAX_OPS] ops; /// BAM-encoded: castable to dhtslib CigarOp
}

struct fadehip_read_batch
{
    int n_reads;
    const(int)* tid;
    const(int)* pos;
    const(ushort)* flag;
    const(ubyte)* has_sa;
    const(int)* l_seq;
    const(uint)* cigar_off;
    const(uint)* cigar_ops;
    const(uint)* seq_off;
    const(ubyte)* seq_packed;
    int n_skipped; /// records left out because anno.d:61-65 gives them rs = 0 (unmapped, no S op)
    int ref_span_bound; /// max cigar.alignedLength over the batch, 0 = let the library scan the CIGARs
    int n_with_seq;     /// ABI 3: records whose seq_off slice is not empty (0 = unknown)
    int l_seq_min;      /// ABI 3: bounds of l_seq over the records with bases (0 = unknown)
    int l_seq_max;
    int reserved;
}

struct fadehip_aln
{
    int read_idx;
    int art; /// bit0 art_left, bit1 art_right
    long win_start;
    int win_len;
    int clip_left, clip_right;
    int aligned_len;
    fadehip_sw_result sw;
}

struct fadehip_anno_out
{
    ubyte* rs;
    fadehip_aln* aln;
    int aln_cap;
    int n_aln;
    long[8] stats;
    int n_oversize;
    int reserved;
}

/// zero-copy results: views into the slot's pinned result block, valid until the slot is uploaded again
struct fadehip_anno_view
{
    const(ubyte)* rs;
    const(fadehip_aln)* aln;
    int n_reads, n_aln;
    long[8] stats;
    int n_oversize;
    int reserved;
}

void fadehip_params_default(fadehip_params* p);
int fadehip_abi_version();
int fadehip_create(fadehip_ctx** out_, int device, const(fadehip_params)* params);
void fadehip_destroy(fadehip_ctx* ctx);
const(char)* fadehip_last_error(const(fadehip_ctx)* ctx);
int fadehip_host_alloc(fadehip_ctx* ctx, size_t bytes, void** out_);
int fadehip_host_free(fadehip_ctx* ctx, void* p);
int fadehip_host_register(fadehip_ctx* ctx, void* p, size_t bytes);
size_t fadehip_batch_bytes(int n_reads, long n_cigar_ops, long n_seq_bytes);
int fadehip_batch_bind(void* base, int n_reads, long n_cigar_ops, long n_seq_bytes, fadehip_read_batch* b);
int fadehip_sw_batch(fadehip_ctx* ctx, int n, const(ubyte)* q, const(long)* q_off,
        const(ubyte)* r, const(long)* r_off, fadehip_sw_result* out_);
int fadehip_genome_upload(fadehip_ctx* ctx, int n_contigs, const(long)* lengths, const(ubyte*)* seqs);
int fadehip_annotate_upload(fadehip_ctx* ctx, int slot, const(fadehip_read_batch)* batch);
int fadehip_annotate_run(fadehip_ctx* ctx, int slot, int floor_len, int window);
int fadehip_annotate_submit(fadehip_ctx* ctx, int slot, const(fadehip_read_batch)* batch,
        int floor_len, int window);
int fadehip_annotate_results(fadehip_ctx* ctx, int slot, fadehip_anno_view* out_);
int fadehip_annotate_collect(fadehip_ctx* ctx, int slot, fadehip_anno_out* out_);
int fadehip_sync(fadehip_ctx* ctx);
int fadehip_last_run_profile(fadehip_ctx* ctx, int slot, float* ms4, long* counts6);
int fadehip_stats_allreduce(fadehip_ctx** ctxs, int n_ctx, long* counters, int count);
/// one process per GPU: rank 0 writes the RCCL id to id_path, the others read it (fadehip.h)
int fadehip_stats_allreduce_rank(fadehip_ctx* ctx, int rank, int n_ranks, const(char)* id_path, long* counters, int count);

/// BGZF members made on the device (htslib bgzf_write's deflate under util.d:65-76); lane 0 or 1
enum FADEHIP_BGZF_BLOCK = 0xff00;
enum FADEHIP_BGZF_LANES = 2;
int fadehip_bgzf_deflate_submit(fadehip_ctx* ctx, int lane, const(void)* src, size_t n_bytes);
int fadehip_bgzf_deflate_wait(fadehip_ctx* ctx, int lane, const(ubyte)** out_, size_t* out_bytes);
/// whole BGZF members in, their payloads out (one wavefront per member; CRC32 and ISIZE checked)
int fadehip_bgzf_inflate(fadehip_ctx* ctx, const(void)* members, size_t n_bytes, void* out_, size_t out_cap, size_t* out_bytes);

/// The file path on the device: anno.d:44-50 as a byte stream (fadehip.h, "the file path on the device")
struct fadehip_bam_stream;
struct fadehip_bam_config {
    int floor_len;               /// --min-length
    int window;                  /// -w
    int n_ref;                   /// contigs of the BAM header
    int flags;                   /// 1 (FADEHIP_BAM_STORED): uncompressed BGZF out; 2 (FADEHIP_BAM_NO_OUTPUT): back makes no BGZF (measurement)
    const(char*)* ref_names;     /// [n_ref]
    uint first_record;           /// payload bytes of the first member passed that precede the first record
    uint tail_trim;              /// payload bytes at the end of the last member that belong to the next reader
}
enum FADEHIP_BAM_CHUNKS = 3;
enum FADEHIP_BAM_STORED = 1;     /// fadehip_bam_config.flags: uncompressed BGZF out (`fade annotate -u`)
enum FADEHIP_BAM_NO_OUTPUT = 2;  /// ... back releases the annotated records without compressing them (measurement)
int fadehip_bam_open(fadehip_ctx* ctx, const(fadehip_bam_config)* cfg, fadehip_bam_stream** out_);
int fadehip_bam_prepare(fadehip_bam_stream* st, size_t call_bytes);
int fadehip_bam_front(fadehip_bam_stream* st, const(void)* members, size_t n_bytes, int last);
int fadehip_bam_front_raw(fadehip_bam_stream* st, const(void)* payload, size_t n_bytes, int last);
int fadehip_bam_back(fadehip_bam_stream* st, const(ubyte)** out_, size_t* out_bytes);
int fadehip_bam_totals(fadehip_bam_stream* st, long* stats8, long* n_records, long* n_oversize);
void fadehip_bam_close(fadehip_bam_stream* st);
