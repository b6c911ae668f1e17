// pg_api.hip — host side of the C-ABI declared in include/panagram_hip.h.
// Owns device memory, sizes/grows the tables, builds launch geometry.  No CPU
// compute fallback lives here: without a GPU pg_ctx_create fails.
#include "../../include/panagram_hip.h"
#include "pg_kernels.h"
#include "pg_guard.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

using namespace pg;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(PG_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *pg_last_error(void) { return g_err.c_str(); }
// used by pg_bgzf.cpp so that both translation units share one error slot
int pg_set_error(int code, const char *msg) {
    g_err = msg ? msg : "";
    return code;
}
extern "C" const char *pg_version(void) { return "panagram_hip 0.2 gfx950"; }
extern "C" uint32_t pg_tile_positions(void) { return (uint32_t)PROBE_TILE; }

// ---------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------
struct pg_ctx {
    int device;
    hipStream_t own_stream;
    hipStream_t stream;
    hipStream_t aux_stream;  // statistics kernels run here, event-ordered behind the probe kernels
    // Handles may be destroyed in any order (a garbage collector frees a dropped object graph in no
    // particular order): an object with live dependants is only marked dead and goes when the last
    // dependant does.
    std::atomic<int> refs{0};
    bool dead = false;
    // staging of the GPU BGZF writer (write_bgzf_gpu): four sets, so that four writer threads can run;
    // allocated at first use, kept — pinning 2 x 64 MiB per file would cost more than the compression
    struct DfSet {
        uint8_t *d_slots[2] = {nullptr, nullptr}, *d_packed[2] = {nullptr, nullptr}, *h_slots[2] = {nullptr, nullptr};
        uint32_t *d_sizes[2] = {nullptr, nullptr}, *d_offs[2] = {nullptr, nullptr}, *h_sizes[2] = {nullptr, nullptr};
        uint32_t *d_crc = nullptr, *d_hist = nullptr;
        void *d_code = nullptr;  // the file's Huffman code and block header (k_df_build_code)
        bool ready = false, busy = false;
    } df[4];
    std::mutex df_mu;
    std::condition_variable df_cv;
    // Row buffers of destroyed results, kept for the next result: hipFree of tens of GB costs about 40 ms per
    // GB on this stack (paid inside the NEXT hipMalloc: tools/malloc_time.py), which a run that anchors its
    // genomes in batches would pay for every batch.  At most two buffers; emptied by pg_ctx_trim, when an
    // allocation fails, and with the context.
    struct RowBuf {
        uint8_t *p;
        uint64_t cap;
    };
    std::vector<RowBuf> row_cache;
    std::mutex row_mu;
};

struct SubHost {
    SubTable d;
    uint64_t count;  // distinct keys
};

struct pg_table {
    pg_ctx *ctx;
    int k, ngenomes, ndbs;
    uint32_t m;  // minimizer length of every sub-table (0 = direct hashing)
    bool m_pinned = false;  // set by pg_table_set_minimizer: re-hashing keeps m
    uint32_t cosched = 0;   // anchor genomes a probe launch will co-schedule (pg_table_set_coscheduled; 0: not told — several)
    uint64_t expected = 0;  // pg_table_create's expected_keys (0: unknown)
    double load0 = 0.375;   // keys per slot the table was created for (TARGET_LOAD; PG_TABLE_KEYS_PER_LINE / pg_table_create_dense: denser)
    uint64_t first_len = 0;  // k-mer positions of the first sequence set inserted into the empty table (settle_minimizer)
    uint64_t max_len = 0;    // ... of the longest one inserted so far (what a re-hash settles m from)
    std::vector<SubHost> subs;
    unsigned long long *d_counters;  // [0] newly claimed, [1] overflow flag
    unsigned long long *h_counters = nullptr;  // pinned landing place of d_counters (read_counters)
    uint32_t *d_tile0 = nullptr;     // first tile of every contig of the seqset being inserted (k_tile0), grown on demand
    size_t tile0_cap = 0;
    double spill = 0;                // keys outside their home line / keys, as of the last pg_table_rehash
    std::atomic<int> refs{0};        // results on this table
    bool dead = false;
    // ONE writer at a time: lane_insert's mask update is a plain read-modify-write that is only safe while every
    // concurrent writer of a word ORs in the same bits (pg_device.h) — i.e. one insert call (one genome, its launches
    // serialised on the context's stream and synchronised before the call returns) at a time.  Every entry point
    // that writes the table holds this lock for its whole duration: a second host thread queues up behind the
    // first instead of racing it, whatever stream the context has been pointed at in between.
    std::mutex write_mu;
};
#define TABLE_WRITER(t) std::lock_guard<std::mutex> writer_guard_((t)->write_mu)

struct pg_seqset {
    pg_ctx *ctx;
    uint32_t n;
    std::vector<SeqDesc> desc;
    uint64_t total_words;
    uint64_t *d_seqw;
    uint32_t *d_nmw;
    uint32_t *d_has_n;
    SeqDesc *d_desc;
    void *d_stage;
    size_t stage_cap;
    std::vector<std::string> names;  // record ids when the seqset was parsed from FASTA text
    std::atomic<int> refs{0};        // results on these sequences
    bool dead = false;
};

struct pg_result {
    pg_ctx *ctx;
    pg_table *tbl;  // NULL for a rows container (pg_result_create_rows): rows arrive through pg_result_merge_columns*
    const pg_seqset *seqs;
    uint32_t N;     // genomes per row (the table's, or the container's own)
    int k;
    uint32_t flags;
    uint32_t lowres_step = 100;  // bitmap.<lowres_step> = every lowres_step-th row (index.py:101-106)
    std::vector<AnchorDesc> ad;
    std::vector<uint64_t> nrows100;
    AnchorDesc *d_ad;
    uint32_t *d_tile_contig;
    uint32_t *d_sched = nullptr;  // optional launch order of the tiles (pg_result_coschedule)
    std::vector<uint32_t> sched_bounds;  // tile indices at which independently scheduled ranges begin / end
    uint32_t ntiles;
    uint8_t *d_out1;
    uint64_t out1_bytes;
    uint64_t out1_cap = 0;  // bytes actually allocated behind d_out1 (it may come out of the context's cache)
    uint8_t *d_out100;
    uint64_t out100_bytes;
    uint32_t *d_bins;
    uint64_t total_bins;
    unsigned long long *d_colsums;
    hipEvent_t ev[4];  // last pg_anchor_run: start / after k_probe (main stream), epilogue start / end (side stream)
    bool ev_ok, ev_epi;
    bool rows_valid = false;  // rows were merged in (pg_result_merge_columns*)
    // HIP-event durations of every pg_anchor_run since the last pg_result_timing_reset: a benchmark
    // averages the launches of all its timed steps, not only the last one
    // (every run records into an event set of its own — ev[] is the latest — so that nothing has to be
    // waited for between steps; sets beyond EV_RING are folded into the sums and recycled)
    struct EvSet {
        hipEvent_t e[4];
        bool probe, epi;  // which of the two intervals (e[0]..e[1] probe, e[2]..e[3] statistics) were recorded
    };
    std::vector<EvSet> ev_hist, ev_free;
    // A whole run goes out as a few CHUNKS of its launch order (slices of the co-schedule), the statistics pass of chunk c
    // on the side stream beside the probe of chunk c+1: the pass reads rows at HBM speed while the probe is busy
    // issuing instructions (run_chunks).  A chunk: schedule slice [s0, s1) and the tile ranges it touches.
    struct Chunk {
        uint32_t s0, s1, r0, nr, tiles;
    };
    std::vector<Chunk> chunks;
    uint2 *d_ranges = nullptr;
    bool chunks_ready = false;
    std::vector<hipEvent_t> chunk_ev;
    size_t hist_skip = 0;  // leading sets of ev_hist from before the last pg_result_timing_reset
    double probe_ms_sum = 0, epi_ms_sum = 0;
    uint32_t probe_runs = 0, epi_runs = 0;
    // Fused statistics (round 6, pg_kernels.h: FuseArgs): k_probe leaves per-tile counters, k_tile_reduce adds them up; the
    // statistics pass then only runs over the tiles of contigs whose bins are shorter than a tile (d_small: their ranges).
    int fuse_state = 0;  // 0: not decided yet, 1: this result's whole runs are fused, -1: they are not (row width, layout, memory)
    uint32_t *d_tile_hist = nullptr, *d_tile_cs = nullptr;
    uint2 *d_small = nullptr;
    uint32_t n_small = 0, small_tiles = 0;
    uint32_t fused_runs = 0;  // whole runs that took the fused path (pg_result_fused_runs: tests and bench.py say which path was timed)
};
static constexpr size_t EV_RING = 128;

static constexpr uint32_t MAX_PROBE = 512;  // lines an insert may walk before the table is grown
static constexpr double GROW_AT = 0.55;     // grow when keys > GROW_AT * slots
#ifndef PG_INLINE_LAYOUT
#define PG_INLINE_LAYOUT 1
#endif
static constexpr double TARGET_LOAD = 0.375; // load right after growing (3 keys per 8-slot line)
static constexpr double HARD_LOAD = 0.85;   // worst-case guard before a batch

static int use_device(const pg_ctx *c) {
    HIP_TRY(hipSetDevice(c->device));
    return PG_OK;
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
extern "C" int pg_ctx_create(int device_id, pg_ctx **out) {
    PG_API_BEGIN
    if (!out) return fail(PG_E_INVALID, "pg_ctx_create: out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(PG_E_HIP, "no HIP device visible (%s); libpanagram_hip has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(PG_E_INVALID, "device %d out of range (0..%d)", device_id, n - 1);
    HIP_TRY(hipSetDevice(device_id));
    pg_ctx *c = new pg_ctx();
    c->device = device_id;
    hipError_t se = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete c;
        return fail(PG_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(se));
    }
    c->stream = c->own_stream;
    se = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        hipStreamDestroy(c->own_stream);
        delete c;
        return fail(PG_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(se));
    }
    // load the three code objects now (first-launch cost otherwise: ≈ 5 ms inside the first table insert)
    hipError_t pe = preload_table_kernels();
    if (pe == hipSuccess) pe = preload_anchor_kernels();
    if (pe == hipSuccess) pe = preload_deflate_kernels();
    if (pe != hipSuccess) {
        hipStreamDestroy(c->aux_stream);
        hipStreamDestroy(c->own_stream);
        delete c;
        return fail(PG_E_HIP, "loading the gfx950 kernels failed: %s (is this an MI355X?)", hipGetErrorString(pe));
    }
    *out = c;
    return PG_OK;
    PG_API_END
}

static void df_free_buffers(pg_ctx::DfSet &d);
static void ctx_free(pg_ctx *c) {
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->aux_stream);
    for (auto &b : c->row_cache) hipFree(b.p);
    c->row_cache.clear();
    for (auto &d : c->df) df_free_buffers(d);
    hipStreamDestroy(c->aux_stream);
    hipStreamDestroy(c->own_stream);
    delete c;
}
static void ctx_release(pg_ctx *c) {
    if (--c->refs == 0 && c->dead) ctx_free(c);
}

static constexpr uint64_t ROW_CACHE_MIN = 1ull << 30;  // smaller buffers are not worth keeping
static void row_cache_trim(pg_ctx *c) {
    std::lock_guard<std::mutex> lk(c->row_mu);
    for (auto &b : c->row_cache) hipFree(b.p);
    c->row_cache.clear();
}
// a buffer of at least `bytes` (never more than twice that) out of the cache, or a fresh one
static hipError_t row_alloc(pg_ctx *c, uint64_t bytes, uint8_t **out, uint64_t *cap) {
    {
        std::lock_guard<std::mutex> lk(c->row_mu);
        for (size_t i = 0; i < c->row_cache.size(); ++i)
            if (c->row_cache[i].cap >= bytes && c->row_cache[i].cap <= 2 * bytes) {
                *out = c->row_cache[i].p;
                *cap = c->row_cache[i].cap;
                c->row_cache.erase(c->row_cache.begin() + (long)i);
                return hipSuccess;
            }
    }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(out), bytes);
    if (e != hipSuccess) {  // make room: give the cached buffers back and try once more
        (void)hipGetLastError();
        row_cache_trim(c);
        e = hipMalloc(reinterpret_cast<void **>(out), bytes);
    }
    *cap = bytes;
    return e;
}
static void row_free(pg_ctx *c, uint8_t *p, uint64_t cap) {
    if (!p) return;
    if (cap >= ROW_CACHE_MIN && !c->dead) {
        std::lock_guard<std::mutex> lk(c->row_mu);
        if (c->row_cache.size() < 2) {
            c->row_cache.push_back({p, cap});
            return;
        }
    }
    hipFree(p);
}

extern "C" int pg_host_alloc(pg_ctx *c, uint64_t nbytes, void **out) {
    PG_API_BEGIN
    if (!c || !out) return fail(PG_E_INVALID, "pg_host_alloc: NULL argument");
    if (int r = use_device(c)) return r;
    *out = nullptr;
    hipError_t e = hipHostMalloc(out, std::max<uint64_t>(nbytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return fail(PG_E_HIP, "hipHostMalloc(%llu bytes) failed: %s", (unsigned long long)nbytes, hipGetErrorString(e));
    }
    return PG_OK;
    PG_API_END
}

extern "C" int pg_host_free(pg_ctx *c, void *p) {
    PG_API_BEGIN
    if (!c) return fail(PG_E_INVALID, "ctx is NULL");
    if (!p) return PG_OK;
    if (int r = use_device(c)) return r;
    HIP_TRY(hipHostFree(p));
    return PG_OK;
    PG_API_END
}

extern "C" int pg_ctx_mem_info(pg_ctx *c, uint64_t *free_bytes, uint64_t *total_bytes) {
    PG_API_BEGIN
    if (!c) return fail(PG_E_INVALID, "ctx is NULL");
    if (int r = use_device(c)) return r;
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    {
        std::lock_guard<std::mutex> lk(c->row_mu);  // cached row buffers are as good as free
        for (auto &b : c->row_cache) f += b.cap;
    }
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_ctx_trim(pg_ctx *c) {
    PG_API_BEGIN
    if (!c) return fail(PG_E_INVALID, "ctx is NULL");
    if (int r = use_device(c)) return r;
    row_cache_trim(c);
    return PG_OK;
    PG_API_END
}

extern "C" int pg_ctx_destroy(pg_ctx *c) {
    PG_API_BEGIN
    if (!c || c->dead) return PG_OK;
    c->dead = true;
    if (c->refs == 0) ctx_free(c);
    return PG_OK;
    PG_API_END
}

// plain device buffers for a caller that has no other allocator at hand (the genome-sharded pipeline's exchange
// buffers in a single-process run; with several ranks torch owns them, because the collective takes tensors)
extern "C" int pg_device_alloc(pg_ctx *c, uint64_t bytes, void **out) {
    PG_API_BEGIN
    if (!c || !out) return fail(PG_E_INVALID, "pg_device_alloc: NULL argument");
    if (int r = use_device(c)) return r;
    void *p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<uint64_t>(bytes, 16)));
    hipError_t e = hipMemsetAsync(p, 0, std::max<uint64_t>(bytes, 16), c->stream);
    if (e != hipSuccess) {
        hipFree(p);
        return fail(PG_E_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return PG_OK;
    PG_API_END
}
extern "C" int pg_device_memset(pg_ctx *c, void *p, int value, uint64_t bytes) {
    PG_API_BEGIN
    if (!c || (!p && bytes)) return fail(PG_E_INVALID, "pg_device_memset: NULL argument");
    if (int r = use_device(c)) return r;
    if (bytes) HIP_TRY(hipMemsetAsync(p, value, bytes, c->stream));
    return PG_OK;
    PG_API_END
}
extern "C" int pg_device_free(pg_ctx *c, void *p) {
    PG_API_BEGIN
    if (!c) return fail(PG_E_INVALID, "ctx is NULL");
    if (!p) return PG_OK;
    if (int r = use_device(c)) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipFree(p));
    return PG_OK;
    PG_API_END
}

extern "C" int pg_ctx_set_stream(pg_ctx *c, void *s, int use_own) {
    PG_API_BEGIN
    if (!c) return fail(PG_E_INVALID, "ctx is NULL");
    c->stream = use_own ? c->own_stream : reinterpret_cast<hipStream_t>(s);
    return PG_OK;
    PG_API_END
}

extern "C" int pg_ctx_synchronize(pg_ctx *c) {
    PG_API_BEGIN
    if (!c) return fail(PG_E_INVALID, "ctx is NULL");
    if (int r = use_device(c)) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipStreamSynchronize(c->aux_stream));
    return PG_OK;
    PG_API_END
}

// ---------------------------------------------------------------------------
// table
// ---------------------------------------------------------------------------
// smallest prime >= n: the probe sequence home + i*step (mod nlines) must visit every line
// for every step, which needs gcd(step, nlines) == 1
static uint64_t next_prime(uint64_t n) {
    if (n < 3) return 3;
    n |= 1;
    for (;; n += 2) {
        bool prime = true;
        for (uint64_t d = 3; d * d <= n; d += 2)
            if (n % d == 0) {
                prime = false;
                break;
            }
        if (prime) return n;
    }
}

static void free_sub(SubTable &d) {
    if (d.buckets) hipFree(d.buckets);
    if (d.masks) hipFree(d.masks);
    d.buckets = d.masks = nullptr;
}

static int alloc_sub(pg_ctx *ctx, uint32_t W, uint32_t word0, uint32_t k, uint32_t m, uint32_t slots, uint64_t nbuckets,
                     uint32_t layout, SubTable *out) {
    if (nbuckets < 64) nbuckets = 64;
    nbuckets = next_prime(nbuckets);
    // (next_line adds a line number and a step in 32 bits: both below nbuckets, so nbuckets <= 2^31 keeps the sum exact)
    if (nbuckets > 0x80000000ull) return fail(PG_E_CAPACITY, "sub-table would exceed 2^31 lines (256 GB of key lines)");
    SubTable t;
    t.W = W;
    t.word0 = word0;
    t.k = k;
    t.m = m;
    t.slots = slots;
    t.layout = layout;
    t.nbuckets = nbuckets;
    t.buckets = t.masks = nullptr;
    auto grab = [&](uint64_t bytes, uint8_t **p) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(p), bytes);
        if (e != hipSuccess) {  // the context may be sitting on cached row buffers: give them back and try once more
            (void)hipGetLastError();
            row_cache_trim(ctx);
            e = hipMalloc(reinterpret_cast<void **>(p), bytes);
        }
        return e;
    };
    const uint64_t key_bytes = nbuckets * line_bytes(t), mask_bytes = layout == LAYOUT_SPLIT ? nbuckets * slots * 4ull * W : 0;
    hipError_t e = grab(key_bytes, &t.buckets);
    if (e == hipSuccess && mask_bytes) e = grab(mask_bytes, &t.masks);
    if (e != hipSuccess) {
        free_sub(t);
        return fail(PG_E_HIP, "hipMalloc(%llu bytes) for k-mer table failed: %s", (unsigned long long)(key_bytes + mask_bytes),
                    hipGetErrorString(e));
    }
    e = launch_table_init(ctx->stream, t);
    if (e != hipSuccess) {
        free_sub(t);
        return fail(PG_E_HIP, "table initialisation failed: %s", hipGetErrorString(e));
    }
    *out = t;
    return PG_OK;
}

// Geometry of a table for `ngenomes`: up to 64 genomes ONE sub-table of 16-byte slots {key, mask0, mask1}; more
// genomes ONE sub-table in the split layout (16 bare keys per 128-byte line + W = ceil(N/32) mask words per slot in
// a second array): one line fetch and one probe per position whatever N, where one 64-genome sub-table per pass
// repeated the whole front end per pass.
struct TableGeom {
    uint32_t W, slots, layout;
};
// 65..96 genomes (W = 3): the inline layout — 6 keys per 128-byte line with their mask blocks behind them (5 at W = 4: kept for experiments), no
// dependent gather (PG_WIDE_LAYOUT=split: the split layout for them too, rounds 2-4; PG_WIDE_LAYOUT=inline: the default)
static TableGeom geom_for(int ngenomes) {
    const uint32_t ndbs = (uint32_t)(ngenomes + 31) / 32;
    if (ndbs <= 2) return {ndbs, 8u, LAYOUT_SLOTS};
    static const int want_inline = [] {  // 0: never, 1: W = 3 (the default), 2: W = 4 too (experiments: PG_WIDE_LAYOUT=inline4)
        const char *e = getenv("PG_WIDE_LAYOUT");
        return (e && strcmp(e, "split") == 0) ? 0 : (e && strcmp(e, "inline4") == 0) ? 2 : 1;
    }();
    // (W = 3 only: at W = 4 a line holds 5 keys and the split layout's 16 win — 128 x 10 Mb 10.0 ms against 10.75 at the inline layout's best m)
    if ((ndbs == 3 || (ndbs == 4 && want_inline == 2)) && want_inline && PG_INLINE_LAYOUT) return {ndbs, inline_slots(ndbs), LAYOUT_INLINE};
    return {ndbs, SPLIT_KEYS, LAYOUT_SPLIT};
}

// keys per slot a table is created for: TARGET_LOAD, or PG_TABLE_KEYS_PER_LINE / 8 when the caller knows its key count
static double create_load(uint64_t expected_keys, double keys_per_line = 0.0) {
    if (keys_per_line >= 1.0 && keys_per_line <= 6.4 && expected_keys) return keys_per_line / 8.0;
    if (const char *e = getenv("PG_TABLE_KEYS_PER_LINE")) {
        const double kpl = atof(e);
        if (kpl >= 1.0 && kpl <= 6.4 && expected_keys) return kpl / 8.0;
    }
    return TARGET_LOAD;
}

extern "C" int pg_table_bytes_for(int k, int ngenomes, uint64_t expected_keys, uint64_t *bytes) {
    PG_API_BEGIN
    if (!bytes) return fail(PG_E_INVALID, "pg_table_bytes_for: NULL argument");
    if (k < 1 || k > 32 || ngenomes < 1) return fail(PG_E_INVALID, "pg_table_bytes_for: bad k / ngenomes");
    const TableGeom g = geom_for(ngenomes);
    // (an estimate: one prime search less — the line count itself, not the next prime above it)
    const uint64_t nb = std::max<uint64_t>((uint64_t)((double)std::max<uint64_t>(expected_keys, 1ull << 18) / (create_load(expected_keys) * g.slots)) + 1, 64);
    *bytes = g.layout == LAYOUT_SPLIT ? nb * g.slots * (8ull + 4ull * g.W) : g.layout == LAYOUT_INLINE ? nb * 128ull : nb * 16ull * g.slots;
    return PG_OK;
    PG_API_END
}

static uint32_t window_cap(int ngenomes = 0);  // (PG_TABLE_WMAX, below)

static int table_create(pg_ctx *ctx, int k, int ngenomes, uint64_t expected_keys, double keys_per_line, pg_table **out);
extern "C" int pg_table_create(pg_ctx *ctx, int k, int ngenomes, uint64_t expected_keys, pg_table **out) {
    PG_API_BEGIN
    return table_create(ctx, k, ngenomes, expected_keys, 0.0, out);
    PG_API_END
}
extern "C" int pg_table_create_dense(pg_ctx *ctx, int k, int ngenomes, uint64_t expected_keys, double keys_per_line, pg_table **out) {
    PG_API_BEGIN
    if (!(keys_per_line >= 1.0 && keys_per_line <= 6.4)) return fail(PG_E_INVALID, "pg_table_create_dense: keys_per_line must be in 1 .. 6.4 (of 8 slots), got %g", keys_per_line);
    if (!expected_keys) return fail(PG_E_INVALID, "pg_table_create_dense: expected_keys must be known");
    return table_create(ctx, k, ngenomes, expected_keys, keys_per_line, out);
    PG_API_END
}
extern "C" int pg_table_bytes_for_dense(int k, int ngenomes, uint64_t expected_keys, double keys_per_line, uint64_t *bytes) {
    PG_API_BEGIN
    if (!bytes) return fail(PG_E_INVALID, "pg_table_bytes_for_dense: NULL argument");
    if (k < 1 || k > 32 || ngenomes < 1 || !(keys_per_line >= 1.0 && keys_per_line <= 6.4)) return fail(PG_E_INVALID, "pg_table_bytes_for_dense: bad argument");
    const TableGeom g = geom_for(ngenomes);  // (keys_per_line counts per 8 slots: a load of keys_per_line / 8 whatever the line holds)
    const uint64_t nb = std::max<uint64_t>((uint64_t)((double)std::max<uint64_t>(expected_keys, 1ull << 18) / (keys_per_line / 8.0 * g.slots)) + 1, 64);
    *bytes = g.layout == LAYOUT_SPLIT ? nb * g.slots * (8ull + 4ull * g.W) : g.layout == LAYOUT_INLINE ? nb * 128ull : nb * 16ull * g.slots;
    return PG_OK;
    PG_API_END
}
static int table_create(pg_ctx *ctx, int k, int ngenomes, uint64_t expected_keys, double keys_per_line, pg_table **out) {
    PG_API_BEGIN
    if (!ctx || !out) return fail(PG_E_INVALID, "pg_table_create: NULL argument");
    if (k < 1 || k > 32) return fail(PG_E_INVALID, "k=%d unsupported (1..32)", k);
    if (ngenomes < 1) return fail(PG_E_INVALID, "ngenomes must be >= 1");
    int ndbs = (ngenomes + 31) / 32;
    if (ndbs > (int)MAX_WORDS) return fail(PG_E_INVALID, "ngenomes=%d exceeds the supported %u", ngenomes, MAX_WORDS * 32);
    if (int r = use_device(ctx)) return r;
    pg_table *t = new pg_table();
    t->ctx = ctx;
    ++ctx->refs;
    t->k = k;
    t->ngenomes = ngenomes;
    t->ndbs = ndbs;
    t->m = minimizer_length((uint32_t)k, expected_keys, 0, window_cap(ngenomes), (uint32_t)ngenomes, 0, create_load(expected_keys, keys_per_line));
    t->expected = expected_keys;
    t->d_counters = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&t->d_counters), 2 * sizeof(unsigned long long));
    if (e != hipSuccess) {
        delete t;
        --ctx->refs;
        return fail(PG_E_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    hipMemsetAsync(t->d_counters, 0, 2 * sizeof(unsigned long long), ctx->stream);
    if (hipHostMalloc(reinterpret_cast<void **>(&t->h_counters), 2 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        t->h_counters = nullptr;  // (read_counters then lands in the caller's pageable memory)
    }
    {
        const TableGeom g = geom_for(ngenomes);
        const uint64_t want = expected_keys ? expected_keys : (1ull << 18);
        // keys per line the table is created for: 3 of 8 slots (TARGET_LOAD) unless PG_TABLE_KEYS_PER_LINE says otherwise (a
        // tuning knob, 1..6.4 keys per 8 slots; the genome-sharded planner's block tables: distributed.py)
        const double load = create_load(expected_keys, keys_per_line);
        t->load0 = load;
        // (pg_table_rehash may widen the lines of the slots layout when the keys call for it)
        const uint64_t nb = (uint64_t)((double)want / (load * g.slots)) + 1;
        SubHost sh;
        sh.count = 0;
        int r = alloc_sub(ctx, g.W, 0, (uint32_t)k, t->m, g.slots, nb, g.layout, &sh.d);
        if (r) {
            pg_table_destroy(t);
            return r;
        }
        t->subs.push_back(sh);
    }
    *out = t;
    return PG_OK;
    PG_API_END
}

static void table_free(pg_table *t) {
    hipSetDevice(t->ctx->device);
    hipStreamSynchronize(t->ctx->stream);
    for (auto &s : t->subs) free_sub(s.d);
    if (t->d_counters) hipFree(t->d_counters);
    if (t->h_counters) hipHostFree(t->h_counters);
    if (t->d_tile0) hipFree(t->d_tile0);
    pg_ctx *c = t->ctx;
    delete t;
    ctx_release(c);
}
static void table_release(pg_table *t) {
    if (--t->refs == 0 && t->dead) table_free(t);
}

extern "C" int pg_table_destroy(pg_table *t) {
    PG_API_BEGIN
    if (!t || t->dead) return PG_OK;
    t->dead = true;
    if (t->refs == 0) table_free(t);
    return PG_OK;
    PG_API_END
}

// every line EMPTY again, the allocation kept: the genome-sharded mode builds one block table after the other in the
// same memory (freeing and re-allocating tens of GB costs about 40 ms per GB on this stack)
extern "C" int pg_table_clear(pg_table *t) {
    PG_API_BEGIN
    if (!t) return fail(PG_E_INVALID, "table is NULL");
    if (t->refs > 0) {
        // (results of this table only hold geometry and their own buffers: they stay valid, their rows are stale)
    }
    if (int r = use_device(t->ctx)) return r;
    TABLE_WRITER(t);
    for (auto &s : t->subs) {
        HIP_TRY(launch_table_init(t->ctx->stream, s.d));
        s.count = 0;
    }
    HIP_TRY(hipMemsetAsync(t->d_counters, 0, 2 * sizeof(unsigned long long), t->ctx->stream));
    t->spill = 0;
    t->first_len = t->max_len = 0;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_table_k(const pg_table *t) { return t ? t->k : 0; }
extern "C" int pg_table_ngenomes(const pg_table *t) { return t ? t->ngenomes : 0; }
extern "C" int pg_table_minimizer(const pg_table *t) { return t ? (int)t->m : 0; }

extern "C" int pg_table_set_minimizer(pg_table *t, int m) {
    PG_API_BEGIN
    if (!t) return fail(PG_E_INVALID, "table is NULL");
    const int w = m ? t->k - m + 1 : 0;
    if (m && (t->k < 20 || w < (int)MZ_WMIN || w > (int)MZ_WMAX))
        return fail(PG_E_INVALID, "minimizer length %d: k=%d needs m=0 or a window k-m+1 in %u..%u (k >= 20)", m, t->k,
                    MZ_WMIN, MZ_WMAX);
    for (auto &s : t->subs)
        if (s.count) return fail(PG_E_INVALID, "pg_table_set_minimizer: the table already holds keys");
    t->m = (uint32_t)m;
    t->m_pinned = true;
    for (auto &s : t->subs) s.d.m = (uint32_t)m;
    return PG_OK;
    PG_API_END
}

// The first sequence set that goes into an EMPTY table tells what the expected key count cannot: the non-redundant
// length of the pangenome (distinct loci; a locus of a many-genome pangenome holds a key per variant).  The minimizer
// length is settled again from it (minimizer_length, pg_device.h) — free while no key has a home line yet.
// PG_TABLE_WMAX=3..8 caps the minimizer window the library chooses (default 8).  The wide window wins 10-25 % on
// mostly unique sequence and loses on content dominated by one young high-copy repeat family, whose minimizer groups it
// makes 1.8 x larger (tools/repeat_stress.py, 27 genomes, 2000 x 3 kb copies at 3 %: 26 % of the genome 56 vs 64 G
// k-mers/s at w = 4, 10 % of it 71 vs 76); pg_table_set_minimizer pins m for one table.
// Tables in the inline layout (65..96 genomes: 6 keys per line) take a window of at most 6 m-mers: a minimizer group of w = 7 holds more
// keys than a line and every second look-up of a variant k-mer overflows — k=21: 65 / 80 / 96 x 10 Mb 87 / 79 / 86 G k-mers/s at m = 15,
// 103 / 98 / 106 at m = 16 (the split layout at its best m: 94 / 91 / 90); k=31, 96 genomes: 88 / 91 / 110 / 102 at w = 8 / 7 / 6 / 5
// (profiles/r5i_inline_layout.txt).
static uint32_t window_cap(int ngenomes) {
    static const uint32_t cap = [] {
        const char *e = getenv("PG_TABLE_WMAX");
        const int v = (e && *e) ? atoi(e) : (int)MZ_WMAX;
        return (uint32_t)std::min<int>((int)MZ_WMAX, std::max<int>((int)MZ_WMIN, v));
    }();
    if (ngenomes > 0 && geom_for(ngenomes).layout == LAYOUT_INLINE) return std::min(cap, 6u);
    return cap;
}

extern "C" int pg_minimizer_length(int k, uint64_t expected_keys, uint64_t first_len, int wmax, int ngenomes) {
    PG_API_BEGIN
    if (k < 1 || k > 32) return 0;
    const uint32_t cap = wmax ? (uint32_t)std::min<int>((int)MZ_WMAX, std::max<int>((int)MZ_WMIN, wmax)) : window_cap(ngenomes);
    return (int)minimizer_length((uint32_t)k, expected_keys, first_len, cap, (uint32_t)std::max(0, ngenomes));
    PG_API_END
}

extern "C" int pg_minimizer_length_for(int k, uint64_t expected_keys, uint64_t first_len, int wmax, int ngenomes, int coscheduled) {
    PG_API_BEGIN
    if (k < 1 || k > 32) return 0;
    const uint32_t cap = wmax ? (uint32_t)std::min<int>((int)MZ_WMAX, std::max<int>((int)MZ_WMIN, wmax)) : window_cap(ngenomes);
    return (int)minimizer_length((uint32_t)k, expected_keys, first_len, cap, (uint32_t)std::max(0, ngenomes), (uint32_t)std::max(0, coscheduled));
    PG_API_END
}

extern "C" int pg_minimizer_length_dense(int k, uint64_t expected_keys, uint64_t first_len, int wmax, int ngenomes, int coscheduled,
                                         double keys_per_line) {
    PG_API_BEGIN
    if (k < 1 || k > 32) return 0;
    const uint32_t cap = wmax ? (uint32_t)std::min<int>((int)MZ_WMAX, std::max<int>((int)MZ_WMIN, wmax)) : window_cap(ngenomes);
    const double load = (keys_per_line >= 1.0 && keys_per_line <= 6.4) ? keys_per_line / 8.0 : TARGET_LOAD;
    return (int)minimizer_length((uint32_t)k, expected_keys, first_len, cap, (uint32_t)std::max(0, ngenomes), (uint32_t)std::max(0, coscheduled), load);
    PG_API_END
}

// How the table will be probed (minimizer_length, pg_device.h): the number of anchor genomes one launch co-schedules.  Only
// while the table is empty — the minimizer length decides every key's home line.
extern "C" int pg_table_set_coscheduled(pg_table *t, int anchors) {
    PG_API_BEGIN
    if (!t) return fail(PG_E_INVALID, "table is NULL");
    if (anchors < 0) return fail(PG_E_INVALID, "pg_table_set_coscheduled: %d anchors", anchors);
    for (auto &s : t->subs)
        if (s.count) return fail(PG_E_INVALID, "pg_table_set_coscheduled: the table already holds keys");
    t->cosched = (uint32_t)anchors;
    if (!t->m_pinned) {
        t->m = minimizer_length((uint32_t)t->k, t->expected, t->first_len, window_cap(t->ngenomes), (uint32_t)t->ngenomes, t->cosched, t->load0);
        for (auto &s : t->subs) s.d.m = t->m;
    }
    return PG_OK;
    PG_API_END
}

static void settle_minimizer(pg_table *t, uint64_t positions) {
    if (t->m_pinned || t->first_len || positions == 0) return;
    for (auto &s : t->subs)
        if (s.count) return;
    t->first_len = positions;
    t->m = minimizer_length((uint32_t)t->k, t->expected, positions, window_cap(t->ngenomes), (uint32_t)t->ngenomes, t->cosched, t->load0);
    for (auto &s : t->subs) s.d.m = t->m;
}

static int read_counters(pg_table *t, unsigned long long out[2]) {
    unsigned long long *land = t->h_counters ? t->h_counters : out;
    HIP_TRY(hipMemcpyAsync(land, t->d_counters, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                           t->ctx->stream));
    HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    if (land != out) out[0] = land[0], out[1] = land[1];
    return PG_OK;
}

// replace sub-table si by one with `nb` buckets holding the same content
static int regrow(pg_table *t, int si, uint64_t nb, uint32_t slots = 0) {
    pg_ctx *ctx = t->ctx;
    if (slots == 0) slots = t->subs[si].d.slots;
    for (int attempt = 0; attempt < 8; ++attempt) {
        SubTable nt;
        if (int r = alloc_sub(ctx, t->subs[si].d.W, t->subs[si].d.word0, (uint32_t)t->k, t->m, slots, nb, t->subs[si].d.layout, &nt)) return r;
        HIP_TRY(hipMemsetAsync(t->d_counters, 0, 2 * sizeof(unsigned long long), ctx->stream));
        HIP_TRY(launch_rehash(ctx->stream, t->subs[si].d, nt, t->d_counters, MAX_PROBE, (uint32_t)t->ngenomes));
        unsigned long long c[2];
        if (int r = read_counters(t, c)) {
            free_sub(nt);
            return r;
        }
        if (c[1] == 0) {
            free_sub(t->subs[si].d);
            t->subs[si].d = nt;
            t->subs[si].count = c[0];
            return PG_OK;
        }
        free_sub(nt);
        nb *= 2;
    }
    return fail(PG_E_CAPACITY, "re-hash keeps overflowing");
}

static int ensure_room(pg_table *t, int si, uint64_t incoming) {
    SubHost &s = t->subs[si];
    const int ns = (int)s.d.slots;
    const double slots = (double)s.d.nbuckets * ns;
    // `incoming` counts positions, an upper bound on new keys.  A table created for a known number of
    // distinct keys (pg_table_create's expected_keys, e.g. from a pg_sketch) is trusted while its
    // count stays inside that number: should the estimate have been wrong, the insert's overflow flag
    // and after_insert() still grow it.
    if (t->expected && s.count <= t->expected && (double)t->expected <= HARD_LOAD * slots) return PG_OK;
    if ((double)(s.count + incoming) > HARD_LOAD * slots) {
        uint64_t nb = (uint64_t)((double)(s.count + incoming) / (HARD_LOAD * 0.9 * ns)) + 1;
        nb = std::max(nb, s.d.nbuckets * 2);
        return regrow(t, si, nb);
    }
    return PG_OK;
}

// an insert ran out of probe sequence: whatever sized the table was wrong — fall back to the
// pessimistic bound (every incoming item a new key), or at least double
static int grow_after_overflow(pg_table *t, int si, uint64_t incoming) {
    t->expected = 0;
    const uint64_t before = t->subs[si].d.nbuckets;
    if (int r = ensure_room(t, si, incoming)) return r;
    if (t->subs[si].d.nbuckets != before) return PG_OK;
    return regrow(t, si, before * 2);
}

static int after_insert(pg_table *t, int si) {
    SubHost &s = t->subs[si];
    const int ns = (int)s.d.slots;
    // (a table created denser on purpose — the genome-sharded mode's block tables: fewer passes — is not grown back to 3 keys per line)
    if ((double)s.count > std::max(GROW_AT, t->load0 + 0.1) * (double)s.d.nbuckets * ns) {
        uint64_t nb = (uint64_t)((double)s.count / (TARGET_LOAD * ns)) + 1;
        return regrow(t, si, nb);
    }
    return PG_OK;
}

// enqueue the insertion of every k-mer of `sq` into sub-table `d` (word w, `bits`; count_mode: occurrences are added):
// ONE launch of the wave-cooperative kernel over all contigs for tables of 128-byte lines, else one launch of the
// thread-per-k-mer kernel per contig (256-byte lines: the PG_TABLE_SLOTS=16 tuning knob).  The launch's contig → first
// tile array is computed on the device into `t`'s own scratch (k_tile0): no allocation, upload or wait per call — the
// caller holds t's writer lock and everything is ordered on the context's stream.
static int enqueue_insert(pg_table *t, const SubTable &d, int w, uint32_t bits, int count_mode, const pg_seqset *sq,
                          unsigned long long *counters) {
    hipStream_t st = t->ctx->stream;
    const bool tiles_ok = d.layout != LAYOUT_SLOTS || d.slots == 8;
    if (!tiles_ok || getenv("PG_INSERT_PER_THREAD")) {
        for (uint32_t c = 0; c < sq->n; ++c) {
            const SeqDesc &sd = sq->desc[c];
            if (sd.len < (uint64_t)t->k) continue;
            HIP_TRY(launch_insert_seq(st, d, w, bits, t->k, sq->d_seqw + sd.seq_off, sq->d_nmw + sd.seq_off, sq->d_has_n + c,
                                      sd.len - t->k + 1, counters, MAX_PROBE, count_mode));
        }
        return PG_OK;
    }
    uint64_t tiles = 0;
    for (uint32_t c = 0; c < sq->n; ++c) {
        const uint64_t len = sq->desc[c].len;
        if (len >= (uint64_t)t->k) tiles += (len - t->k + 1 + PROBE_TILE - 1) / PROBE_TILE;
    }
    if (tiles > 0x7FFFFFFFull) return fail(PG_E_INVALID, "too many tiles in one insert launch");
    if (tiles == 0) return PG_OK;
    if (t->tile0_cap < (size_t)sq->n + 1) {
        if (t->d_tile0) {
            HIP_TRY(hipStreamSynchronize(st));  // (an earlier launch may still read the old array)
            hipFree(t->d_tile0);
            t->d_tile0 = nullptr, t->tile0_cap = 0;
        }
        const size_t cap = std::max<size_t>(1024, ((size_t)sq->n + 1) * 2);
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&t->d_tile0), cap * 4));
        t->tile0_cap = cap;
    }
    HIP_TRY(launch_tile0(st, sq->d_desc, sq->n, (uint32_t)t->k, (uint32_t)PROBE_TILE, t->d_tile0));
    HIP_TRY(launch_insert_tiles(st, d, w, bits, count_mode, sq->d_seqw, sq->d_nmw, sq->d_has_n, sq->d_desc, t->d_tile0, sq->n,
                                (uint32_t)tiles, counters, MAX_PROBE));
    return PG_OK;
}

extern "C" int pg_table_insert_seqset(pg_table *t, int g, const pg_seqset *sq) {
    PG_API_BEGIN
    if (!t || !sq) return fail(PG_E_INVALID, "pg_table_insert_seqset: NULL argument");
    if (g < 0 || g >= t->ngenomes) return fail(PG_E_INVALID, "genome index %d out of range (0..%d)", g, t->ngenomes - 1);
    if (t->ctx != sq->ctx) return fail(PG_E_INVALID, "table and seqset belong to different contexts");
    if (int r = use_device(t->ctx)) return r;
    TABLE_WRITER(t);
    const int d = g / 32, si = 0, w = d;  // (one sub-table: the group's mask word)
    const uint32_t bits = 1u << (g % 32);
    uint64_t total = 0;
    for (auto &c : sq->desc)
        if (c.len >= (uint64_t)t->k) total += c.len - t->k + 1;
    settle_minimizer(t, total);
    t->max_len = std::max(t->max_len, total);
    if (int r = ensure_room(t, si, total)) return r;
    for (int attempt = 0; attempt < 8; ++attempt) {
        hipStream_t st = t->ctx->stream;
        HIP_TRY(hipMemsetAsync(t->d_counters, 0, 2 * sizeof(unsigned long long), st));
        const int er = enqueue_insert(t, t->subs[si].d, w, bits, 0, sq, t->d_counters);
        unsigned long long cnt[2] = {0, 0};
        const int rr = er ? er : read_counters(t, cnt);  // (synchronises)
        if (rr) return rr;
        t->subs[si].count += cnt[0];
        if (cnt[1] == 0) return after_insert(t, si);
        // a probe chain exceeded MAX_PROBE buckets: grow and redo (inserts are idempotent)
        if (int r = grow_after_overflow(t, si, total)) return r;
    }
    return fail(PG_E_CAPACITY, "k-mer table keeps overflowing");
    PG_API_END
}

// Bits only: genome g's bit goes into the keys of `sq`'s k-mers that the table ALREADY holds; no key is added.  A table
// built from the genomes a process anchors and then updated with all the others answers every look-up the anchoring
// makes exactly as the table of all genomes does — a position's k-mer is one of the anchors' own — at a fraction of its
// size.  (Tables of 256-byte lines, PG_TABLE_SLOTS=16 / PG_INSERT_PER_THREAD, have no such kernel: they insert.)
extern "C" int pg_table_update_seqset(pg_table *t, int g, const pg_seqset *sq) {
    PG_API_BEGIN
    if (!t || !sq) return fail(PG_E_INVALID, "pg_table_update_seqset: NULL argument");
    if (g < 0 || g >= t->ngenomes) return fail(PG_E_INVALID, "genome index %d out of range (0..%d)", g, t->ngenomes - 1);
    if (t->ctx != sq->ctx) return fail(PG_E_INVALID, "table and seqset belong to different contexts");
    const SubTable &d0 = t->subs[0].d;
    if (!(d0.layout != LAYOUT_SLOTS || d0.slots == 8) || getenv("PG_INSERT_PER_THREAD")) return pg_table_insert_seqset(t, g, sq);
    if (int r = use_device(t->ctx)) return r;
    TABLE_WRITER(t);
    const int w = g / 32;
    const uint32_t bits = 1u << (g % 32);
    hipStream_t st = t->ctx->stream;
    HIP_TRY(hipMemsetAsync(t->d_counters, 0, 2 * sizeof(unsigned long long), st));
    const int er = enqueue_insert(t, t->subs[0].d, w, bits, 2 /* update only */, sq, t->d_counters);
    unsigned long long cnt[2] = {0, 0};
    return er ? er : read_counters(t, cnt);  // (synchronises)
    PG_API_END
}

// kmc -ci<min_count> (workflow/Snakefile:88-89: -ci2 for FASTQ samples): occurrences are counted in
// a private table first; the keys seen at least min_count times then enter the pan table.
extern "C" int pg_table_insert_seqset_min(pg_table *t, int g, const pg_seqset *sq, uint32_t min_count) {
    PG_API_BEGIN
    if (min_count <= 1) return pg_table_insert_seqset(t, g, sq);
    if (!t || !sq) return fail(PG_E_INVALID, "pg_table_insert_seqset_min: NULL argument");
    if (g < 0 || g >= t->ngenomes) return fail(PG_E_INVALID, "genome index %d out of range (0..%d)", g, t->ngenomes - 1);
    if (t->ctx != sq->ctx) return fail(PG_E_INVALID, "table and seqset belong to different contexts");
    if (int r = use_device(t->ctx)) return r;
    const int d = g / 32, si = 0, w = d;  // (one sub-table: the group's mask word)
    const uint32_t bits = 1u << (g % 32);
    uint64_t total = 0;
    for (auto &c : sq->desc)
        if (c.len >= (uint64_t)t->k) total += c.len - t->k + 1;
    if (total == 0) return PG_OK;
    TABLE_WRITER(t);
    hipStream_t st = t->ctx->stream;
    uint64_t expect = total / 3 + 1024;
    for (int attempt = 0; attempt < 8; ++attempt, expect *= 2) {
        pg_table *cnt = nullptr;
        if (int r = pg_table_create(t->ctx, t->k, 1, expect, &cnt)) return r;
        int rc = PG_OK;
        unsigned long long c2[2] = {0, 0};
        do {
            if (hipMemsetAsync(cnt->d_counters, 0, 2 * sizeof(unsigned long long), st) != hipSuccess) {
                rc = fail(PG_E_HIP, "hipMemsetAsync failed");
                break;
            }
            rc = enqueue_insert(t, cnt->subs[0].d, 0, 1u, 1, sq, cnt->d_counters);
            if (!rc) rc = read_counters(cnt, c2);
        } while (0);
        // counting is not idempotent: a table that overflowed (or ran too full) is thrown away and
        // the pass repeated in a bigger one
        const bool redo = !rc && (c2[1] != 0 || (double)c2[0] > HARD_LOAD * (double)cnt->subs[0].d.nbuckets * cnt->subs[0].d.slots);
        if (!rc && !redo) {
            rc = ensure_room(t, si, c2[0]);
            for (int a2 = 0; a2 < 8 && !rc; ++a2) {
                unsigned long long m2[2];
                if (hipMemsetAsync(t->d_counters, 0, 2 * sizeof(unsigned long long), st) != hipSuccess ||
                    launch_merge_min(st, cnt->subs[0].d, t->subs[si].d, w, bits, min_count, t->d_counters, MAX_PROBE) != hipSuccess) {
                    rc = fail(PG_E_HIP, "merge kernel failed");
                    break;
                }
                if ((rc = read_counters(t, m2))) break;
                t->subs[si].count += m2[0];
                if (m2[1] == 0) {
                    rc = after_insert(t, si);
                    break;
                }
                rc = grow_after_overflow(t, si, c2[0]);  // the merge is idempotent: redo it
                if (!rc && a2 == 7) rc = fail(PG_E_CAPACITY, "k-mer table keeps overflowing");
            }
        }
        pg_table_destroy(cnt);
        if (rc || !redo) return rc;
    }
    return fail(PG_E_CAPACITY, "k-mer counting table keeps overflowing");
    PG_API_END
}

static int insert_keys_dev(pg_table *t, int db_idx, const uint64_t *d_keys, const uint32_t *d_vals, uint64_t n) {
    const int si = 0, w = db_idx;
    if (int r = ensure_room(t, si, n)) return r;
    for (int attempt = 0; attempt < 8; ++attempt) {
        hipStream_t st = t->ctx->stream;
        HIP_TRY(hipMemsetAsync(t->d_counters, 0, 2 * sizeof(unsigned long long), st));
        HIP_TRY(launch_insert_keys(st, t->subs[si].d, w, d_keys, d_vals, n, t->d_counters, MAX_PROBE));
        unsigned long long cnt[2];
        if (int r = read_counters(t, cnt)) return r;
        t->subs[si].count += cnt[0];
        if (cnt[1] == 0) return after_insert(t, si);
        if (int r = grow_after_overflow(t, si, n)) return r;
    }
    return fail(PG_E_CAPACITY, "k-mer table keeps overflowing");
}

extern "C" int pg_table_insert_keys(pg_table *t, int db_idx, const uint64_t *keys, const uint32_t *counters,
                                    uint64_t n) {
    PG_API_BEGIN
    if (!t || (n && (!keys || !counters))) return fail(PG_E_INVALID, "pg_table_insert_keys: NULL argument");
    if (db_idx < 0 || db_idx >= t->ndbs) return fail(PG_E_INVALID, "db index %d out of range (0..%d)", db_idx, t->ndbs - 1);
    if (n == 0) return PG_OK;
    if (int r = use_device(t->ctx)) return r;
    TABLE_WRITER(t);
    uint64_t *dk = nullptr;
    uint32_t *dv = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dk), n * 8));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&dv), n * 4);
    if (e != hipSuccess) {
        hipFree(dk);
        return fail(PG_E_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
    }
    int r = PG_OK;
    if (hipMemcpyAsync(dk, keys, n * 8, hipMemcpyHostToDevice, t->ctx->stream) != hipSuccess ||
        hipMemcpyAsync(dv, counters, n * 4, hipMemcpyHostToDevice, t->ctx->stream) != hipSuccess)
        r = fail(PG_E_HIP, "H2D copy of keys failed");
    if (!r) r = insert_keys_dev(t, db_idx, dk, dv, n);
    hipStreamSynchronize(t->ctx->stream);
    hipFree(dk);
    hipFree(dv);
    return r;
    PG_API_END
}

// ---------------------------------------------------------------------------
// KMC databases.  Both layouts CKMCFile::OpenForRA accepts (cpp/anchor.cpp:29, index.py:859-860):
//   KMC1 (kmc_version 0; kmc_tools output):
//       pre = "KMCP" | u64 LUT[4^p] | header(64) | u32 header_offset | "KMCP"
//   KMC2 (kmc_version 0x200; what `kmc` itself writes, workflow/Snakefile:101-104):
//       pre = "KMCP" | u64 LUT[bins][4^p] (+ one guard entry) | u32 signature_map[4^s + 1] | header(68) | u32 header_offset | "KMCP"
//       header = k, mode, counter_size, lut_prefix_length, signature_len, min_count, max_count, u64 total, both_strands, ...
//       the suffix file holds the bins one after the other, each sorted; LUT[b][x] = number of the first record of
//       bin b whose first p symbols are x.  (The signature map only serves random access by signature: a bulk
//       import needs the LUTs alone.)
//   suf = "KMCS" | records (suffix bytes, counter) | "KMCS"   in both.
// The records never pass through host containers: the suffix file image is uploaded in chunks (from wherever the
// caller holds it — a memory map is fine) and k_import_kmc turns records into table inserts on the GPU.
// ---------------------------------------------------------------------------
struct KmcHeader {
    uint32_t k, mode, csz, lut_p, sig_len, minc, maxc, ver, hoff;
    uint64_t total, nlut;  // nlut: LUT entries (bins x 4^lut_p), without the guard
};

static int parse_kmc_pre(const uint8_t *pre, size_t pre_len, KmcHeader *H) {
    if (pre_len < 4 + 8 + 64 + 8 || memcmp(pre, "KMCP", 4) || memcmp(pre + pre_len - 4, "KMCP", 4))
        return fail(PG_E_FORMAT, "kmc_pre: missing KMCP markers");
    memcpy(&H->hoff, pre + pre_len - 8, 4);
    if (H->hoff < 64 || (size_t)H->hoff + 8 + 4 > pre_len) return fail(PG_E_FORMAT, "kmc_pre: bad header offset %u", H->hoff);
    const uint8_t *h = pre + pre_len - 8 - H->hoff;
    memcpy(&H->ver, pre + pre_len - 12, 4);  // last field of the header in both layouts
    if (H->ver != 0 && H->ver != 0x200)
        return fail(PG_E_FORMAT, "kmc_pre: kmc_version=0x%x; the KMC1 (0) and KMC2 (0x200) layouts are supported", H->ver);
    memcpy(&H->k, h, 4);
    memcpy(&H->mode, h + 4, 4);
    memcpy(&H->csz, h + 8, 4);
    memcpy(&H->lut_p, h + 12, 4);
    const uint8_t *q = h + 16;
    H->sig_len = 0;
    if (H->ver == 0x200) {
        if (H->hoff < 68) return fail(PG_E_FORMAT, "kmc_pre: KMC2 header of %u bytes is too short", H->hoff);
        memcpy(&H->sig_len, q, 4);
        q += 4;
    }
    memcpy(&H->minc, q, 4);
    memcpy(&H->maxc, q + 4, 4);
    memcpy(&H->total, q + 8, 8);
    if (H->mode != 0) return fail(PG_E_FORMAT, "kmc_pre: quality-mode databases are not supported");
    if (H->k < 1 || H->k > 32) return fail(PG_E_FORMAT, "kmc_pre: k=%u unsupported (1..32)", H->k);
    if (H->csz > 4) return fail(PG_E_FORMAT, "kmc_pre: counter_size=%u unsupported", H->csz);
    if (H->ver == 0 && H->csz < 1) return fail(PG_E_FORMAT, "kmc_pre: counter_size=0 in a KMC1 database");
    if (H->lut_p < 1 || H->lut_p > 15 || H->lut_p > H->k || (H->k - H->lut_p) % 4)
        return fail(PG_E_FORMAT, "kmc_pre: lut_prefix_length=%u invalid for k=%u", H->lut_p, H->k);
    const uint64_t per_bin = 1ull << (2 * H->lut_p);
    uint64_t lut_bytes = pre_len - 4 - (H->hoff + 8);
    if (H->ver == 0x200) {
        if (H->sig_len < 5 || H->sig_len > 11) return fail(PG_E_FORMAT, "kmc_pre: signature_len=%u out of range (5..11)", H->sig_len);
        const uint64_t map_bytes = ((1ull << (2 * H->sig_len)) + 1) * 4;
        if (lut_bytes < map_bytes + 8) return fail(PG_E_FORMAT, "kmc_pre: truncated (no room for the signature map)");
        lut_bytes -= map_bytes;
        // bins x 4^p entries, with or without one guard entry behind them (the reference's reader — which puts a
        // guard of its own behind whatever it read — accepts both: tests/golden/make_golden.py, kmc2_* fixtures)
        const uint64_t entries = lut_bytes / 8;
        if (lut_bytes % 8 || entries < per_bin || (entries % per_bin != 0 && (entries - 1) % per_bin != 0))
            return fail(PG_E_FORMAT, "kmc_pre: prefix area of %llu bytes is not bins x 4^%u entries (+ guard)", (unsigned long long)lut_bytes, H->lut_p);
        H->nlut = entries % per_bin == 0 ? entries : entries - 1;
    } else {
        if (lut_bytes < per_bin * 8) return fail(PG_E_FORMAT, "kmc_pre: truncated prefix table");
        H->nlut = per_bin;
    }
    return PG_OK;
}

extern "C" int pg_table_load_kmc(pg_table *t, int db_idx, const void *pre_, size_t pre_len, const void *suf_,
                                 size_t suf_len) {
    PG_API_BEGIN
    if (!t || !pre_ || !suf_) return fail(PG_E_INVALID, "pg_table_load_kmc: NULL argument");
    if (db_idx < 0 || db_idx >= t->ndbs) return fail(PG_E_INVALID, "db index %d out of range (0..%d)", db_idx, t->ndbs - 1);
    const uint8_t *pre = static_cast<const uint8_t *>(pre_);
    const uint8_t *suf = static_cast<const uint8_t *>(suf_);
    KmcHeader H;
    if (int r = parse_kmc_pre(pre, pre_len, &H)) return r;
    if (suf_len < 8 || memcmp(suf, "KMCS", 4) || memcmp(suf + suf_len - 4, "KMCS", 4))
        return fail(PG_E_FORMAT, "kmc_suf: missing KMCS markers");
    if ((int)H.k != t->k) return fail(PG_E_FORMAT, "database k=%u but table k=%d", H.k, t->k);
    const uint32_t sb = (H.k - H.lut_p) / 4, rec = sb + H.csz;
    // (a division, not 8 + total * rec: the product wraps for a corrupt total_kmers, the check would pass and the
    // chunk loop would read far past the mapping; checked before anything is allocated for `total` records)
    if (rec ? H.total > (suf_len - 8) / rec : H.total > H.nlut)
        return fail(PG_E_FORMAT, "kmc_suf: truncated (%llu records of %u bytes expected)", (unsigned long long)H.total, rec);
    // the LUT must be monotone from 0 to total: it is what maps a record number to its prefix
    std::vector<uint64_t> lut(H.nlut);
    memcpy(lut.data(), pre + 4, H.nlut * 8);
    uint64_t prev = 0;
    if (H.nlut && lut[0] != 0) return fail(PG_E_FORMAT, "kmc_pre: prefix table does not start at record 0");
    for (uint64_t i = 0; i < H.nlut; ++i) {
        if (lut[i] < prev || lut[i] > H.total) return fail(PG_E_FORMAT, "kmc_pre: prefix table not monotone at entry %llu", (unsigned long long)i);
        prev = lut[i];
    }
    if (H.total == 0) return PG_OK;
    if (rec == 0) return fail(PG_E_FORMAT, "kmc_pre: records of zero bytes (k == lut_prefix_length without counters)");
    if (int r = use_device(t->ctx)) return r;
    TABLE_WRITER(t);
    const int si = 0, w = db_idx;
    {   // the first database into an EMPTY table: its record count is a key count the table was not created with — settle
        // the minimizer length from it (and from how the table will be probed) while no key has a home line yet
        bool empty = !t->m_pinned && !t->first_len;
        for (auto &sh : t->subs) empty = empty && sh.count == 0;
        if (empty && !t->expected) {
            t->m = minimizer_length((uint32_t)t->k, H.total, 0, window_cap(t->ngenomes), (uint32_t)t->ngenomes, t->cosched);
            for (auto &sh : t->subs) sh.d.m = t->m;
        }
    }
    if (int r = ensure_room(t, si, H.total)) return r;
    hipStream_t st = t->ctx->stream;
    // chunks of whole records, about 256 MiB each, through two device buffers: the upload of chunk c+1 (pageable or
    // mapped host memory: HIP stages it) runs behind the import kernel of chunk c
    const uint64_t chunk_recs = std::max<uint64_t>(1, (256ull << 20) / rec);
    uint64_t *d_lut = nullptr;
    uint8_t *d_rec[2] = {nullptr, nullptr};
    hipStream_t up = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    int rc = PG_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_lut), H.nlut * 8);
    const uint64_t buf_bytes = std::min<uint64_t>(chunk_recs, H.total) * rec;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipMalloc(reinterpret_cast<void **>(&d_rec[i]), buf_bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_up[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&up, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpyAsync(d_lut, lut.data(), H.nlut * 8, hipMemcpyHostToDevice, st);
    const uint8_t *recs = suf + 4;
    const uint64_t nchunks = (H.total + chunk_recs - 1) / chunk_recs;
    for (int attempt = 0; attempt < 8 && e == hipSuccess && rc == PG_OK; ++attempt) {
        // (inserts are idempotent: a pass that overflowed the probe bound is simply run again on the grown table)
        e = hipMemsetAsync(t->d_counters, 0, 2 * sizeof(unsigned long long), st);
        // Two passes over the records when the counters are presence masks of several genomes: first the k-mers most of
        // the database's genomes share, then the others — keys that go in first take their minimizer's home line, and the
        // shared ones are the ones most look-ups ask for (a table filled in file order probes 7 % slower).
        const uint32_t db_genomes = (uint32_t)std::min(32, t->ngenomes - 32 * db_idx);
        const uint32_t nphases = db_genomes >= 4 ? 2u : 1u;
        uint64_t seq = 0;  // chunks uploaded so far (over both passes): buffer = seq & 1
        auto upload_seq = [&](uint64_t c, uint64_t sq) {
            const int b = (int)(sq & 1);
            const uint64_t r0 = c * chunk_recs, n = std::min(chunk_recs, H.total - r0);
            hipError_t x = sq >= 2 ? hipStreamWaitEvent(up, ev_done[b], 0) : hipSuccess;  // the buffer's previous kernel
            if (x == hipSuccess) x = hipMemcpyAsync(d_rec[b], recs + r0 * rec, n * rec, hipMemcpyHostToDevice, up);
            if (x == hipSuccess) x = hipEventRecord(ev_up[b], up);
            return x;
        };
        const uint64_t total_chunks = nchunks * nphases;
        if (e == hipSuccess && attempt > 0) e = hipStreamSynchronize(st);  // (the buffers' events of the previous attempt are done)
        if (e == hipSuccess) e = upload_seq(0, 0);
        for (seq = 0; seq < total_chunks && e == hipSuccess; ++seq) {
            const uint64_t c = seq % nchunks;
            const uint32_t phase = nphases == 1 ? 2u : (uint32_t)(seq / nchunks);
            const int b = (int)(seq & 1);
            const uint64_t r0 = c * chunk_recs, n = std::min(chunk_recs, H.total - r0);
            if (seq + 1 < total_chunks) e = upload_seq((seq + 1) % nchunks, seq + 1);
            if (e == hipSuccess) e = hipStreamWaitEvent(st, ev_up[b], 0);
            if (e == hipSuccess)
                e = launch_import_kmc(st, t->subs[si].d, w, d_rec[b], r0, n, d_lut, H.nlut, 1u << (2 * H.lut_p), sb, H.csz,
                                      H.minc, H.maxc, t->d_counters, MAX_PROBE, phase, db_genomes / 2);
            if (e == hipSuccess) e = hipEventRecord(ev_done[b], st);
        }
        if (e != hipSuccess) break;
        unsigned long long cnt[2];
        if ((rc = read_counters(t, cnt))) break;
        t->subs[si].count += cnt[0];
        if (cnt[1] == 0) {
            rc = after_insert(t, si);
            break;
        }
        if ((rc = grow_after_overflow(t, si, H.total))) break;
        if (attempt == 7) rc = fail(PG_E_CAPACITY, "k-mer table keeps overflowing");
        if (up) hipStreamSynchronize(up);
    }
    if (e != hipSuccess) rc = fail(PG_E_HIP, "pg_table_load_kmc: %s", hipGetErrorString(e));
    hipStreamSynchronize(st);
    if (up) {
        hipStreamSynchronize(up);
        hipStreamDestroy(up);
    }
    for (int i = 0; i < 2; ++i) {
        if (d_rec[i]) hipFree(d_rec[i]);
        if (ev_up[i]) hipEventDestroy(ev_up[i]);
        if (ev_done[i]) hipEventDestroy(ev_done[i]);
    }
    if (d_lut) hipFree(d_lut);
    return rc;
    PG_API_END
}

// (kept under its round-1 name: the KMC1 layout was the only one read then)
extern "C" int pg_table_load_kmc1(pg_table *t, int db_idx, const void *pre, size_t pre_len, const void *suf, size_t suf_len) {
    PG_API_BEGIN
    return pg_table_load_kmc(t, db_idx, pre, pre_len, suf, suf_len);
    PG_API_END
}

// k of a KMC database from its .kmc_pre image (either layout): what a caller needs before it can create the table
extern "C" int pg_kmc_kmer_length(const void *pre, size_t pre_len, uint32_t *k) {
    PG_API_BEGIN
    if (!pre || !k) return fail(PG_E_INVALID, "pg_kmc_kmer_length: NULL argument");
    KmcHeader H;
    if (int r = parse_kmc_pre(static_cast<const uint8_t *>(pre), pre_len, &H)) return r;
    *k = H.k;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_table_stats(pg_table *t, uint64_t *nkeys, uint64_t *nslots, uint64_t *nbuckets, uint64_t *bytes) {
    PG_API_BEGIN
    if (!t) return fail(PG_E_INVALID, "table is NULL");
    uint64_t a = 0, b = 0, c = 0;
    for (auto &s : t->subs) {
        a += s.count;
        b += s.d.nbuckets * s.d.slots;
        c += s.d.nbuckets;
    }
    if (nkeys) *nkeys = a;
    if (nslots) *nslots = b;
    if (nbuckets) *nbuckets = c;
    if (bytes) {
        uint64_t by = 0;
        for (auto &s2 : t->subs) by += table_bytes(s2.d);
        *bytes = by;
    }
    return PG_OK;
    PG_API_END
}

// keys of sub-table si that do not sit in their group's home line
static int count_spill(pg_table *t, int si, uint64_t *out) {
    hipStream_t st = t->ctx->stream;
    HIP_TRY(hipMemsetAsync(t->d_counters, 0, 2 * sizeof(unsigned long long), st));
    HIP_TRY(launch_count_spill(st, t->subs[si].d, t->d_counters));
    unsigned long long c[2];
    if (int r = read_counters(t, c)) return r;
    *out = c[0];
    return PG_OK;
}

extern "C" int pg_table_rehash(pg_table *t, double keys_per_bucket) {
    PG_API_BEGIN
    if (!t) return fail(PG_E_INVALID, "table is NULL");
    if (!(keys_per_bucket > 0.05 && keys_per_bucket <= 8.0)) return fail(PG_E_INVALID, "keys_per_bucket must be in (0.05, 8]");
    if (int r = use_device(t->ctx)) return r;
    TABLE_WRITER(t);
    if (!t->m_pinned) {  // the key count is known now: settle the minimizer length for it
        uint64_t most = 0;
        for (auto &s : t->subs) most = std::max<uint64_t>(most, s.count);
        t->m = minimizer_length((uint32_t)t->k, most, t->max_len, window_cap(t->ngenomes), (uint32_t)t->ngenomes, t->cosched);
    }
    // Line width: 128-byte lines of 8 slots.  256-byte lines of 16 slots (PG_TABLE_SLOTS=16, a tuning
    // knob) keep a many-variant locus in ONE place at the price of two requests per line; measured,
    // they only pay for >= 40 genomes at 1 % divergence (+6 %) and cost 10-25 % everywhere else
    // (tools/slots_calib.sh), so they are never chosen automatically.
    uint32_t slots = 8;
    if (const char *e = getenv("PG_TABLE_SLOTS"))
        if (atoi(e) == 16) slots = 16;
    uint64_t keys = 0, spilled = 0;
    for (size_t si = 0; si < t->subs.size(); ++si) {
        if (t->subs[si].d.layout != LAYOUT_SLOTS) slots = t->subs[si].d.slots;  // (split and inline lines keep their geometry)
        const double kpb = std::min(keys_per_bucket * (slots / 8.0), 0.8 * slots);  // keys per line
        if (int r = regrow(t, (int)si, (uint64_t)((double)t->subs[si].count / kpb) + 1, slots)) return r;
        uint64_t sp = 0;
        if (int r = count_spill(t, (int)si, &sp)) return r;
        keys += t->subs[si].count;
        spilled += sp;
    }
    t->spill = keys ? (double)spilled / (double)keys : 0.0;
    return PG_OK;
    PG_API_END
}

// the same figure for the table AS IT STANDS (pg_table_spill answers as of the last re-hash: 0 for a table that was built in
// place and never re-hashed — Index.run()'s, bench.py's): one pass over the table's lines (k_count_spill)
extern "C" int pg_table_measure_spill(pg_table *t, double *fraction) {
    PG_API_BEGIN
    if (!t) return fail(PG_E_INVALID, "table is NULL");
    if (int r = use_device(t->ctx)) return r;
    TABLE_WRITER(t);
    uint64_t keys = 0, spilled = 0;
    for (size_t si = 0; si < t->subs.size(); ++si) {
        uint64_t sp = 0;
        if (int r = count_spill(t, (int)si, &sp)) return r;
        keys += t->subs[si].count;
        spilled += sp;
    }
    // (returned only: pg_table_spill keeps answering "as of the last pg_table_rehash", its documented contract)
    if (fraction) *fraction = keys ? (double)spilled / (double)keys : 0.0;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_table_spill(const pg_table *t, double *fraction, uint32_t *slots) {
    PG_API_BEGIN
    if (!t) return fail(PG_E_INVALID, "table is NULL");
    if (fraction) *fraction = t->spill;
    if (slots) *slots = t->subs.empty() ? 0 : t->subs[0].d.slots;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_table_export(pg_table *t, int db_idx, uint64_t *keys, uint32_t *counters, uint64_t cap, uint64_t *n) {
    PG_API_BEGIN
    if (!t || !n) return fail(PG_E_INVALID, "pg_table_export: NULL argument");
    if (db_idx < 0 || db_idx >= t->ndbs) return fail(PG_E_INVALID, "db index %d out of range", db_idx);
    if (int r = use_device(t->ctx)) return r;
    const int si = 0, w = db_idx;
    hipStream_t st = t->ctx->stream;
    uint64_t *dk = nullptr;
    uint32_t *dv = nullptr;
    if (keys && cap) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&dk), cap * 8));
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&dv), cap * 4);
        if (e != hipSuccess) {
            hipFree(dk);
            return fail(PG_E_HIP, "hipMalloc failed: %s", hipGetErrorString(e));
        }
    }
    int rc = PG_OK;
    unsigned long long cnt[2] = {0, 0};
    do {
        if (hipMemsetAsync(t->d_counters, 0, 16, st) != hipSuccess ||
            launch_export(st, t->subs[si].d, w, dk, dv, dk ? cap : 0, t->d_counters) != hipSuccess) {
            rc = fail(PG_E_HIP, "export kernel failed");
            break;
        }
        if ((rc = read_counters(t, cnt))) break;
        if (dk) {
            uint64_t m = std::min<uint64_t>(cnt[0], cap);
            if (hipMemcpy(keys, dk, m * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(counters, dv, m * 4, hipMemcpyDeviceToHost) != hipSuccess)
                rc = fail(PG_E_HIP, "D2H copy failed");
        }
    } while (0);
    if (dk) hipFree(dk);
    if (dv) hipFree(dv);
    *n = cnt[0];
    return rc;
    PG_API_END
}

// ---------------------------------------------------------------------------
// distinct-k-mer sketch (sizes a table before it is built)
// ---------------------------------------------------------------------------
struct pg_sketch {
    pg_ctx *ctx;
    int k;
    uint32_t *d_regs;
};

extern "C" int pg_sketch_create(pg_ctx *ctx, int k, pg_sketch **out) {
    PG_API_BEGIN
    if (!ctx || !out) return fail(PG_E_INVALID, "pg_sketch_create: NULL argument");
    if (k < 1 || k > 32) return fail(PG_E_INVALID, "k=%d unsupported (1..32)", k);
    if (int r = use_device(ctx)) return r;
    uint32_t *regs = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&regs), sizeof(uint32_t) << SKETCH_BITS));
    hipError_t e = hipMemsetAsync(regs, 0, sizeof(uint32_t) << SKETCH_BITS, ctx->stream);
    if (e != hipSuccess) {
        hipFree(regs);
        return fail(PG_E_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    }
    pg_sketch *sk = new pg_sketch{ctx, k, regs};
    ++ctx->refs;
    *out = sk;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_sketch_destroy(pg_sketch *sk) {
    PG_API_BEGIN
    if (!sk) return PG_OK;
    hipSetDevice(sk->ctx->device);
    hipStreamSynchronize(sk->ctx->stream);
    hipFree(sk->d_regs);
    pg_ctx *c = sk->ctx;
    delete sk;
    ctx_release(c);
    return PG_OK;
    PG_API_END
}

extern "C" int pg_sketch_add_seqset(pg_sketch *sk, const pg_seqset *sq) {
    PG_API_BEGIN
    if (!sk || !sq) return fail(PG_E_INVALID, "pg_sketch_add_seqset: NULL argument");
    if (sk->ctx != sq->ctx) return fail(PG_E_INVALID, "sketch and seqset belong to different contexts");
    if (int r = use_device(sk->ctx)) return r;
    // one launch over all contigs (a launch per contig was 76 ms per genome of 20 000 contigs)
    std::vector<uint2> jobs;
    for (uint32_t c = 0; c < sq->n; ++c) {
        const SeqDesc &sd = sq->desc[c];
        if (sd.len < (uint64_t)sk->k) continue;
        const uint64_t nk = sd.len - sk->k + 1;
        for (uint64_t q = 0; q * SKETCH_JOB < nk; ++q) jobs.push_back(make_uint2(c, (uint32_t)q));
    }
    if (jobs.empty()) return PG_OK;
    hipStream_t st = sk->ctx->stream;
    uint2 *d_jobs = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d_jobs), jobs.size() * sizeof(uint2)));
    hipError_t e = hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(uint2), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = launch_sketch_set(st, sk->k, sq->d_desc, d_jobs, (uint32_t)jobs.size(), sq->d_seqw, sq->d_nmw, sq->d_has_n, sk->d_regs);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // (the job list is freed below)
    hipFree(d_jobs);
    if (e != hipSuccess) return fail(PG_E_HIP, "pg_sketch_add_seqset: %s", hipGetErrorString(e));
    return PG_OK;
    PG_API_END
}

extern "C" int pg_sketch_registers(pg_sketch *sk, uint8_t *out) {
    PG_API_BEGIN
    if (!sk || !out) return fail(PG_E_INVALID, "pg_sketch_registers: NULL argument");
    if (int r = use_device(sk->ctx)) return r;
    std::vector<uint32_t> regs((size_t)1 << SKETCH_BITS);
    HIP_TRY(hipMemcpyAsync(regs.data(), sk->d_regs, regs.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, sk->ctx->stream));
    HIP_TRY(hipStreamSynchronize(sk->ctx->stream));
    for (size_t i = 0; i < regs.size(); ++i) out[i] = (uint8_t)regs[i];
    return PG_OK;
    PG_API_END
}

// HyperLogLog (Flajolet et al. 2007) with the small-range correction; 64-bit hashes need no
// large-range one.  Standard error 1.04 / sqrt(2^16) = 0.4 %.
extern "C" int pg_sketch_estimate_registers(const uint8_t *regs, uint64_t *distinct) {
    PG_API_BEGIN
    if (!regs || !distinct) return fail(PG_E_INVALID, "pg_sketch_estimate_registers: NULL argument");
    const size_t n = (size_t)1 << SKETCH_BITS;
    const double m = (double)n;
    double sum = 0.0;
    size_t zeros = 0;
    for (size_t i = 0; i < n; ++i) {
        sum += std::ldexp(1.0, -(int)regs[i]);
        zeros += regs[i] == 0;
    }
    double est = (0.7213 / (1.0 + 1.079 / m)) * m * m / sum;
    if (est <= 2.5 * m && zeros) est = m * std::log(m / (double)zeros);
    *distinct = (uint64_t)(est + 0.5);
    return PG_OK;
    PG_API_END
}

extern "C" int pg_sketch_estimate(pg_sketch *sk, uint64_t *distinct) {
    PG_API_BEGIN
    if (!sk || !distinct) return fail(PG_E_INVALID, "pg_sketch_estimate: NULL argument");
    std::vector<uint8_t> regs((size_t)1 << SKETCH_BITS);
    if (int r = pg_sketch_registers(sk, regs.data())) return r;
    return pg_sketch_estimate_registers(regs.data(), distinct);
    PG_API_END
}

extern "C" int pg_sketch_reset(pg_sketch *sk) {
    PG_API_BEGIN
    if (!sk) return fail(PG_E_INVALID, "pg_sketch_reset: NULL argument");
    if (int r = use_device(sk->ctx)) return r;
    HIP_TRY(hipMemsetAsync(sk->d_regs, 0, sizeof(uint32_t) << SKETCH_BITS, sk->ctx->stream));
    return PG_OK;
    PG_API_END
}

// ---------------------------------------------------------------------------
// seqset
// ---------------------------------------------------------------------------
extern "C" int pg_seqset_create(pg_ctx *ctx, uint32_t ncontigs, const uint64_t *lens, pg_seqset **out) {
    PG_API_BEGIN
    if (!ctx || !out || (ncontigs && !lens)) return fail(PG_E_INVALID, "pg_seqset_create: NULL argument");
    if (int r = use_device(ctx)) return r;
    pg_seqset *s = new pg_seqset();
    s->ctx = ctx;
    ++ctx->refs;
    s->n = ncontigs;
    s->d_seqw = nullptr;
    s->d_nmw = nullptr;
    s->d_has_n = nullptr;
    s->d_desc = nullptr;
    s->d_stage = nullptr;
    s->stage_cap = 0;
    uint64_t off = 0;
    for (uint32_t i = 0; i < ncontigs; ++i) {
        SeqDesc d;
        d.len = lens[i];
        d.nwords = (lens[i] + 31) / 32 + 2;  // +2 zero words: the kernels read one word past a window
        d.seq_off = off;
        off += d.nwords;
        s->desc.push_back(d);
    }
    s->total_words = off;
    hipStream_t st = ctx->stream;
    hipError_t e = hipSuccess;
    size_t nw = std::max<uint64_t>(off, 1), nc = std::max<uint32_t>(ncontigs, 1);
    if ((e = hipMalloc(reinterpret_cast<void **>(&s->d_seqw), nw * 8)) == hipSuccess &&
        (e = hipMalloc(reinterpret_cast<void **>(&s->d_nmw), nw * 4)) == hipSuccess &&
        (e = hipMalloc(reinterpret_cast<void **>(&s->d_has_n), nc * 4)) == hipSuccess &&
        (e = hipMalloc(reinterpret_cast<void **>(&s->d_desc), nc * sizeof(SeqDesc))) == hipSuccess &&
        (e = hipMemsetAsync(s->d_seqw, 0, nw * 8, st)) == hipSuccess &&
        (e = hipMemsetAsync(s->d_nmw, 0, nw * 4, st)) == hipSuccess &&
        (e = hipMemsetAsync(s->d_has_n, 0, nc * 4, st)) == hipSuccess) {
        if (ncontigs)
            e = hipMemcpyAsync(s->d_desc, s->desc.data(), ncontigs * sizeof(SeqDesc), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) {
        pg_seqset_destroy(s);
        return fail(PG_E_HIP, "seqset allocation failed: %s", hipGetErrorString(e));
    }
    *out = s;
    return PG_OK;
    PG_API_END
}

static void seqset_free(pg_seqset *s) {
    hipSetDevice(s->ctx->device);
    hipStreamSynchronize(s->ctx->stream);
    hipFree(s->d_seqw);
    hipFree(s->d_nmw);
    hipFree(s->d_has_n);
    hipFree(s->d_desc);
    if (s->d_stage) hipFree(s->d_stage);
    pg_ctx *c = s->ctx;
    delete s;
    ctx_release(c);
}
static void seqset_release(pg_seqset *s) {
    if (--s->refs == 0 && s->dead) seqset_free(s);
}

extern "C" int pg_seqset_destroy(pg_seqset *s) {
    PG_API_BEGIN
    if (!s || s->dead) return PG_OK;
    s->dead = true;
    if (s->refs == 0) seqset_free(s);
    return PG_OK;
    PG_API_END
}

extern "C" int pg_seqset_load_dev(pg_seqset *s, uint32_t idx, const void *d_ascii, uint64_t len) {
    PG_API_BEGIN
    if (!s || (len && !d_ascii)) return fail(PG_E_INVALID, "pg_seqset_load_dev: NULL argument");
    if (idx >= s->n) return fail(PG_E_INVALID, "contig %u out of range (0..%u)", idx, s->n ? s->n - 1 : 0);
    const SeqDesc &d = s->desc[idx];
    if (len != d.len) return fail(PG_E_INVALID, "contig %u: length %llu != declared %llu", idx, (unsigned long long)len, (unsigned long long)d.len);
    if (int r = use_device(s->ctx)) return r;
    HIP_TRY(hipMemsetAsync(s->d_has_n + idx, 0, 4, s->ctx->stream));
    HIP_TRY(launch_pack(s->ctx->stream, d_ascii, len, s->d_seqw + d.seq_off, s->d_nmw + d.seq_off,
                        (len + 31) / 32, s->d_has_n + idx));
    return PG_OK;
    PG_API_END
}

extern "C" int pg_seqset_load_host(pg_seqset *s, uint32_t idx, const char *ascii, uint64_t len) {
    PG_API_BEGIN
    if (!s || (len && !ascii)) return fail(PG_E_INVALID, "pg_seqset_load_host: NULL argument");
    if (idx >= s->n) return fail(PG_E_INVALID, "contig %u out of range", idx);
    if (int r = use_device(s->ctx)) return r;
    if (len > s->stage_cap) {
        HIP_TRY(hipStreamSynchronize(s->ctx->stream));
        if (s->d_stage) hipFree(s->d_stage);
        s->d_stage = nullptr;
        s->stage_cap = 0;
        size_t cap = (len + 4095) & ~(size_t)4095;
        HIP_TRY(hipMalloc(&s->d_stage, cap));
        s->stage_cap = cap;
    }
    if (len) HIP_TRY(hipMemcpyAsync(s->d_stage, ascii, len, hipMemcpyHostToDevice, s->ctx->stream));
    if (int r = pg_seqset_load_dev(s, idx, s->d_stage, len)) return r;
    HIP_TRY(hipStreamSynchronize(s->ctx->stream));  // staging buffer is reused by the next call
    return PG_OK;
    PG_API_END
}

// ---------------------------------------------------------------------------
// FASTA text -> seqset.  Host: locate the header lines (memchr over the text, '>' is rare).
// GPU: drop the white space of the sequence lines and pack (k_text_count/scan/pack).
// ---------------------------------------------------------------------------
static inline bool host_is_ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }

extern "C" int pg_seqset_from_fasta(pg_ctx *ctx, const void *text_, uint64_t nbytes, pg_seqset **out) {
    PG_API_BEGIN
    if (!ctx || !out || (nbytes && !text_)) return fail(PG_E_INVALID, "pg_seqset_from_fasta: NULL argument");
    if (int r = use_device(ctx)) return r;
    const unsigned char *text = static_cast<const unsigned char *>(text_);
    // The text goes up while the host looks for the header lines: the copy out of pageable memory (the runtime stages it,
    // ~10 GB/s) and the memchr pass over the same bytes each take 6-10 ms per 100 MB, one after the other they were most
    // of what a genome's load costs.  The upload runs on a helper thread and a stream of its own; the text buffer comes out of the context's buffer cache (one per genome of a pangenome, all about the
    // same size: freeing GBs is paid by the next big allocation).
    const uint64_t tcap = (nbytes + 4095) / 4096 * 4096 + 4096;
    uint8_t *d_text = nullptr;
    uint64_t text_cap = 0;
    hipError_t e_up = row_alloc(ctx, tcap, &d_text, &text_cap);
    if (e_up != hipSuccess) return fail(PG_E_HIP, "FASTA packing failed: %s", hipGetErrorString(e_up));
    struct Upload {  // (joined on every way out, also an exception's)
        std::thread th;
        ~Upload() {
            if (th.joinable()) th.join();
        }
    } up;
    struct TextGuard {  // (given back after the upload has been joined: declared after it would free it first)
        pg_ctx *c;
        uint8_t *p;
        uint64_t cap;
        Upload *u;
        ~TextGuard() {
            if (u->th.joinable()) u->th.join();
            row_free(c, p, cap);
        }
    } text_guard{ctx, d_text, text_cap, &up};
    auto upload = [&]() noexcept {
        if (hipSetDevice(ctx->device) != hipSuccess) {
            e_up = hipErrorInvalidDevice;
            return;
        }
        // (a stream of its own: several genomes may be loading at once — Index.load_inputs parses in its reader threads —
        // and the staging of a pageable copy is host work that runs in the calling thread)
        hipStream_t us = nullptr;
        e_up = hipStreamCreateWithFlags(&us, hipStreamNonBlocking);
        if (e_up != hipSuccess) return;
        e_up = hipMemsetAsync(d_text + nbytes, 0, tcap - nbytes, us);
        if (e_up == hipSuccess && nbytes) e_up = hipMemcpyAsync(d_text, text, nbytes, hipMemcpyHostToDevice, us);
        if (e_up == hipSuccess) e_up = hipStreamSynchronize(us);
        (void)hipStreamDestroy(us);
    };
    if (nbytes < (1u << 20)) {
        upload();  // (a small text: a thread and a stream per call cost more than the overlap brings)
    } else {
        try {
            up.th = std::thread(upload);
        } catch (const std::system_error &) {  // no thread to be had: the copy runs here, before the scan
            upload();
        }
    }
    struct Rec {
        std::string name;
        uint64_t s, e;
    };
    std::vector<Rec> recs;
    // A large text's header lines are looked for ON THE DEVICE once the text is there (k_text_headers): the host's memchr pass
    // over the same bytes runs at 20 GB/s — 10 ms per 200 MB, more than the DMA of a page-locked text takes (4 ms), and beside
    // the staging copy of a pageable one it competes for the same memory.  The few positions come back sorted; more than
    // HDR_CAP of them (a read set passed off as FASTA), or any error: the host looks for itself, as for small texts.
    constexpr uint32_t HDR_CAP = 1u << 16;
    std::vector<uint64_t> hdrs;
    bool have_hdrs = false;
    if (nbytes >= (4u << 20) && !getenv("PG_FASTA_HOST_SCAN")) {
        if (up.th.joinable()) up.th.join();
        uint64_t *d_hdr = nullptr;
        uint32_t *d_nhdr = nullptr, nh = 0;
        if (e_up == hipSuccess && hipMalloc(reinterpret_cast<void **>(&d_hdr), (size_t)HDR_CAP * 8) == hipSuccess &&
            hipMalloc(reinterpret_cast<void **>(&d_nhdr), 4) == hipSuccess) {
            hipStream_t st = ctx->stream;
            if (launch_text_headers(st, d_text, nbytes, d_hdr, HDR_CAP, d_nhdr) == hipSuccess &&
                hipMemcpyAsync(&nh, d_nhdr, 4, hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess && nh <= HDR_CAP) {
                hdrs.resize(nh);
                if (nh == 0 || hipMemcpy(hdrs.data(), d_hdr, (size_t)nh * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                    std::sort(hdrs.begin(), hdrs.end());
                    have_hdrs = true;
                }
            }
        }
        (void)hipGetLastError();
        if (d_hdr) hipFree(d_hdr);
        if (d_nhdr) hipFree(d_nhdr);
    }
    size_t hdr_at = 0;
    // first header: at offset 0 or right after a newline; anything before it is ignored
    auto next_header = [&](uint64_t from) -> uint64_t {
        if (have_hdrs) {  // (asked for in ascending order)
            while (hdr_at < hdrs.size() && hdrs[hdr_at] < from) ++hdr_at;
            return hdr_at < hdrs.size() ? hdrs[hdr_at] : nbytes;
        }
        uint64_t p = from;
        while (p < nbytes) {
            const void *q = memchr(text + p, '>', nbytes - p);
            if (!q) return nbytes;
            p = (uint64_t)(static_cast<const unsigned char *>(q) - text);
            if (p == 0 || text[p - 1] == '\n') return p;
            ++p;
        }
        return nbytes;
    };
    uint64_t h = next_header(0);
    while (h < nbytes) {
        const void *q = memchr(text + h, '\n', nbytes - h);
        const uint64_t eol = q ? (uint64_t)(static_cast<const unsigned char *>(q) - text) : nbytes;
        uint64_t a = h + 1;
        while (a < eol && host_is_ws(text[a])) ++a;
        uint64_t b = a;
        while (b < eol && !host_is_ws(text[b])) ++b;
        Rec r;
        r.name.assign(reinterpret_cast<const char *>(text + a), b - a);
        r.s = std::min<uint64_t>(eol + 1, nbytes);
        const uint64_t hn = next_header(r.s);
        r.e = hn;
        recs.push_back(r);
        h = hn;
    }
    const uint32_t nrec = (uint32_t)recs.size();
    // upper-bound layout (text bytes >= bases): the packed planes can be laid out before counting
    std::vector<uint64_t> ub(nrec);
    std::vector<TextChunk> chunks;
    std::vector<uint64_t> chunk0(nrec + 1, 0);
    for (uint32_t i = 0; i < nrec; ++i) {
        ub[i] = recs[i].e - recs[i].s;
        chunk0[i] = chunks.size();
        for (uint64_t p = recs[i].s; p < recs[i].e;) {
            const uint64_t lim = std::min<uint64_t>(recs[i].e, (p / 4096 + 1) * 4096);
            TextChunk c;
            c.off = p;
            c.len = (uint32_t)(lim - p);
            c.rec = i;
            chunks.push_back(c);
            p = lim;
        }
    }
    chunk0[nrec] = chunks.size();
    pg_seqset *s = nullptr;
    if (int r = pg_seqset_create(ctx, nrec, ub.data(), &s)) return r;
    for (auto &r : recs) s->names.push_back(r.name);
    if (nrec == 0) {
        *out = s;
        return PG_OK;
    }
    hipStream_t st = ctx->stream;
    const uint64_t nch = chunks.size();
    TextChunk *d_chunks = nullptr;
    uint64_t *d_chunk0 = nullptr, *d_base = nullptr, *d_len = nullptr;
    uint32_t *d_counts = nullptr;
    std::vector<uint64_t> lens(nrec, 0);
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t x) {
        if (e == hipSuccess) e = x;
        return e == hipSuccess;
    };
    if (ok(hipMalloc(reinterpret_cast<void **>(&d_chunks), std::max<uint64_t>(nch, 1) * sizeof(TextChunk))) &&
        ok(hipMalloc(reinterpret_cast<void **>(&d_chunk0), (nrec + 1) * 8)) &&
        ok(hipMalloc(reinterpret_cast<void **>(&d_base), std::max<uint64_t>(nch, 1) * 8)) &&
        ok(hipMalloc(reinterpret_cast<void **>(&d_len), nrec * 8)) &&
        ok(hipMalloc(reinterpret_cast<void **>(&d_counts), std::max<uint64_t>(nch, 1) * 4))) {
        if (up.th.joinable()) up.th.join();  // (the text is up — the helper waited for its stream)
        ok(e_up);
        if (nch) ok(hipMemcpyAsync(d_chunks, chunks.data(), nch * sizeof(TextChunk), hipMemcpyHostToDevice, st));
        ok(hipMemcpyAsync(d_chunk0, chunk0.data(), (nrec + 1) * 8, hipMemcpyHostToDevice, st));
        if (e == hipSuccess)
            ok(launch_text_pack(st, d_text, d_chunks, nch, d_chunk0, nrec, d_counts, d_base, d_len, s->d_desc, s->d_seqw,
                                s->d_nmw, s->d_has_n));
        ok(hipMemcpyAsync(lens.data(), d_len, nrec * 8, hipMemcpyDeviceToHost, st));
        ok(hipStreamSynchronize(st));
        if (e == hipSuccess) {
            for (uint32_t i = 0; i < nrec; ++i) s->desc[i].len = lens[i];
            ok(hipMemcpyAsync(s->d_desc, s->desc.data(), nrec * sizeof(SeqDesc), hipMemcpyHostToDevice, st));
            ok(hipStreamSynchronize(st));
        }
    }
    // (an error may have come back with kernels still queued on st that read the text buffer — the upload ran on a stream of
    // its own, nothing else orders them against the buffer's next user once it is back in the context's cache)
    if (e != hipSuccess) (void)hipStreamSynchronize(st);
    hipFree(d_chunks);
    hipFree(d_chunk0);
    hipFree(d_base);
    hipFree(d_len);
    hipFree(d_counts);
    if (e != hipSuccess) {
        pg_seqset_destroy(s);
        return fail(PG_E_HIP, "FASTA packing failed: %s", hipGetErrorString(e));
    }
    *out = s;
    return PG_OK;
    PG_API_END
}

// one seqset holding contigs [first[i], first[i] + count[i]) of sets[i], in order (device-to-device copy of the packed
// planes); first == NULL: every contig of every set
static int seqset_concat(pg_ctx *ctx, const pg_seqset *const *sets, const uint32_t *first, const uint32_t *count,
                         uint32_t nsets, pg_seqset **out) {
    if (!ctx || !out || (nsets && !sets)) return fail(PG_E_INVALID, "pg_seqset_concat: NULL argument");
    std::vector<uint64_t> lens;
    for (uint32_t i = 0; i < nsets; ++i) {
        if (!sets[i] || sets[i]->ctx != ctx) return fail(PG_E_INVALID, "pg_seqset_concat: seqset %u is NULL or of another context", i);
        const uint32_t f = first ? first[i] : 0, n = first ? count[i] : sets[i]->n;
        if ((uint64_t)f + n > sets[i]->n) return fail(PG_E_INVALID, "pg_seqset_concat_ranges: contigs %u..%u of seqset %u out of range", f, f + n, i);
        for (uint32_t j = f; j < f + n; ++j) lens.push_back(sets[i]->desc[j].len);
    }
    pg_seqset *s = nullptr;
    if (int r = pg_seqset_create(ctx, (uint32_t)lens.size(), lens.data(), &s)) return r;
    hipStream_t st = ctx->stream;
    hipError_t e = hipSuccess;
    // one gather launch for all contigs (a copy per plane and contig was 9 us each: 1.3 s for the 160 000 contigs of
    // eight fragmented assemblies); the job list goes up in one piece
    std::vector<SeqCopy> jobs;
    jobs.reserve(lens.size());
    uint64_t max_words = 0;
    uint32_t c = 0;
    for (uint32_t i = 0; i < nsets; ++i) {
        const pg_seqset *src = sets[i];
        const uint32_t f = first ? first[i] : 0, n = first ? count[i] : src->n;
        for (uint32_t j = f; j < f + n; ++j, ++c) {
            SeqCopy q;
            q.src_seqw = src->d_seqw;
            q.src_nmw = src->d_nmw;
            q.src_has_n = src->d_has_n + j;
            q.src_off = src->desc[j].seq_off;
            q.dst_off = s->desc[c].seq_off;
            q.nwords = std::min(src->desc[j].nwords, s->desc[c].nwords);
            max_words = std::max(max_words, q.nwords);
            jobs.push_back(q);
            s->names.push_back(j < src->names.size() ? src->names[j] : std::string());
        }
    }
    SeqCopy *d_jobs = nullptr;
    if (!jobs.empty()) {
        e = hipMalloc(reinterpret_cast<void **>(&d_jobs), jobs.size() * sizeof(SeqCopy));
        if (e == hipSuccess) e = hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(SeqCopy), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = launch_seq_gather(st, d_jobs, (uint32_t)jobs.size(), max_words, s->d_seqw, s->d_nmw, s->d_has_n);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (d_jobs) hipFree(d_jobs);
    if (e != hipSuccess) {
        pg_seqset_destroy(s);
        return fail(PG_E_HIP, "pg_seqset_concat: %s", hipGetErrorString(e));
    }
    *out = s;
    return PG_OK;
}

extern "C" int pg_seqset_concat(pg_ctx *ctx, const pg_seqset *const *sets, uint32_t nsets, pg_seqset **out) {
    PG_API_BEGIN
    return seqset_concat(ctx, sets, nullptr, nullptr, nsets, out);
    PG_API_END
}

extern "C" int pg_seqset_concat_ranges(pg_ctx *ctx, const pg_seqset *const *sets, const uint32_t *first_contig,
                                       const uint32_t *ncontigs, uint32_t nsets, pg_seqset **out) {
    PG_API_BEGIN
    if (nsets && (!first_contig || !ncontigs)) return fail(PG_E_INVALID, "pg_seqset_concat_ranges: NULL argument");
    return seqset_concat(ctx, sets, first_contig, ncontigs, nsets, out);
    PG_API_END
}

extern "C" int pg_seqset_slice(pg_ctx *ctx, const pg_seqset *src, uint32_t n, const uint32_t *contig, const uint64_t *start,
                               const uint64_t *len, pg_seqset **out) {
    PG_API_BEGIN
    if (!ctx || !src || !out || (n && (!contig || !start || !len))) return fail(PG_E_INVALID, "pg_seqset_slice: NULL argument");
    if (src->ctx != ctx) return fail(PG_E_INVALID, "pg_seqset_slice: the seqset belongs to another context");
    for (uint32_t i = 0; i < n; ++i) {
        if (contig[i] >= src->n) return fail(PG_E_INVALID, "pg_seqset_slice: contig %u out of range (0..%u)", contig[i], src->n ? src->n - 1 : 0);
        if (start[i] & 31u) return fail(PG_E_INVALID, "pg_seqset_slice: piece %u starts at base %llu — starts must be multiples of 32", i, (unsigned long long)start[i]);
        if (start[i] > src->desc[contig[i]].len || len[i] > src->desc[contig[i]].len - start[i])
            return fail(PG_E_INVALID, "pg_seqset_slice: piece %u (%llu + %llu) exceeds contig %u of %llu bases", i, (unsigned long long)start[i],
                        (unsigned long long)len[i], contig[i], (unsigned long long)src->desc[contig[i]].len);
    }
    pg_seqset *s = nullptr;
    if (int r = pg_seqset_create(ctx, n, len, &s)) return r;
    hipStream_t st = ctx->stream;
    hipError_t e = hipSuccess;
    for (uint32_t i = 0; i < n && e == hipSuccess; ++i) {
        const SeqDesc &from = src->desc[contig[i]];
        const uint64_t w0 = start[i] >> 5, nw = (len[i] + 31) >> 5;  // (whole words: the piece starts on a word boundary)
        if (nw) {
            e = hipMemcpyAsync(s->d_seqw + s->desc[i].seq_off, src->d_seqw + from.seq_off + w0, nw * 8, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess)
                e = hipMemcpyAsync(s->d_nmw + s->desc[i].seq_off, src->d_nmw + from.seq_off + w0, nw * 4, hipMemcpyDeviceToDevice, st);
        }
        // (the contig's "holds a byte outside ACGT" flag is inherited: a piece without one only reads a zero plane)
        if (e == hipSuccess) e = hipMemcpyAsync(s->d_has_n + i, src->d_has_n + contig[i], 4, hipMemcpyDeviceToDevice, st);
        std::string nm = contig[i] < src->names.size() ? src->names[contig[i]] : std::string();
        s->names.push_back(nm + ":" + std::to_string((unsigned long long)start[i]));
    }
    if (e == hipSuccess) e = launch_seq_tailmask(st, s->d_desc, n, s->d_seqw, s->d_nmw);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        pg_seqset_destroy(s);
        return fail(PG_E_HIP, "pg_seqset_slice: %s", hipGetErrorString(e));
    }
    *out = s;
    return PG_OK;
    PG_API_END
}

extern "C" uint32_t pg_seqset_ncontigs(const pg_seqset *s) { return s ? s->n : 0; }

extern "C" int pg_seqset_contig(const pg_seqset *s, uint32_t idx, const char **name, uint64_t *len) {
    PG_API_BEGIN
    if (!s) return fail(PG_E_INVALID, "seqset is NULL");
    if (idx >= s->n) return fail(PG_E_INVALID, "contig %u out of range", idx);
    if (name) *name = idx < s->names.size() ? s->names[idx].c_str() : "";
    if (len) *len = s->desc[idx].len;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_seqset_describe(const pg_seqset *s, uint64_t *lens, char *names, uint64_t names_cap, uint64_t *names_bytes) {
    PG_API_BEGIN
    if (!s) return fail(PG_E_INVALID, "seqset is NULL");
    uint64_t need = 0;
    for (uint32_t i = 0; i < s->n; ++i) {
        if (lens) lens[i] = s->desc[i].len;
        need += (i < s->names.size() ? s->names[i].size() : 0) + 1;
    }
    if (names_bytes) *names_bytes = need;
    if (names) {
        if (names_cap < need) return fail(PG_E_INVALID, "pg_seqset_describe: %llu bytes of names, room for %llu", (unsigned long long)need, (unsigned long long)names_cap);
        char *p = names;
        for (uint32_t i = 0; i < s->n; ++i) {
            if (i < s->names.size()) {
                memcpy(p, s->names[i].data(), s->names[i].size());
                p += s->names[i].size();
            }
            *p++ = 0;
        }
    }
    return PG_OK;
    PG_API_END
}

extern "C" int pg_seqset_unpack(const pg_seqset *s, uint32_t idx, char *out) {
    PG_API_BEGIN
    if (!s || !out) return fail(PG_E_INVALID, "pg_seqset_unpack: NULL argument");
    if (idx >= s->n) return fail(PG_E_INVALID, "contig %u out of range", idx);
    if (int r = use_device(s->ctx)) return r;
    const SeqDesc &d = s->desc[idx];
    const uint64_t nw = (d.len + 31) / 32;
    std::vector<uint64_t> w(nw);
    std::vector<uint32_t> nm(nw);
    if (nw) {
        HIP_TRY(hipMemcpyAsync(w.data(), s->d_seqw + d.seq_off, nw * 8, hipMemcpyDeviceToHost, s->ctx->stream));
        HIP_TRY(hipMemcpyAsync(nm.data(), s->d_nmw + d.seq_off, nw * 4, hipMemcpyDeviceToHost, s->ctx->stream));
        HIP_TRY(hipStreamSynchronize(s->ctx->stream));
    }
    for (uint64_t i = 0; i < d.len; ++i)
        out[i] = ((nm[i >> 5] >> (i & 31)) & 1u) ? 'N' : "ACGT"[(w[i >> 5] >> (2 * (i & 31))) & 3u];
    return PG_OK;
    PG_API_END
}

extern "C" uint64_t pg_seqset_total_kmers(const pg_seqset *s, int k) {
    uint64_t t = 0;
    if (s)
        for (auto &d : s->desc)
            if (d.len >= (uint64_t)k) t += d.len - k + 1;
    return t;
}

// ---------------------------------------------------------------------------
// anchoring
// ---------------------------------------------------------------------------
static TableDesc make_desc(const pg_table *t) {
    TableDesc T;
    memset(&T, 0, sizeof T);
    T.nsub = (uint32_t)t->subs.size();
    for (uint32_t i = 0; i < T.nsub; ++i) T.sub[i] = t->subs[i].d;
    T.ndbs = t->ndbs;
    T.k = t->k;
    T.ngenomes = t->ngenomes;
    return T;
}

// bitmap.<lowres_step> / bin geometry of a result (index.py:101-106, 1169-1172; cpp/anchor.cpp:114-118 hard-codes
// 100 / 200000 / 100)
struct ResultGeom {
    uint32_t lowres_step = 100;
    uint64_t max_bin_len = 200000;
    uint32_t min_bin_count = 100;
};

static int result_create(pg_ctx *ctx, pg_table *t, int k, uint32_t N, const pg_seqset *sq, uint32_t flags,
                         const ResultGeom &geo, pg_result **out) {
    if (ctx != sq->ctx) return fail(PG_E_INVALID, "table / context and seqset belong to different contexts");
    if (geo.lowres_step < 1 || geo.max_bin_len < 1 || geo.min_bin_count < 1)
        return fail(PG_E_INVALID, "lowres_step, max_bin_len and min_bin_count must be >= 1");
    if (int r = use_device(ctx)) return r;
    const uint32_t nbytes = (N + 7) / 8;
    pg_result *r = new pg_result();
    r->ctx = ctx;
    r->tbl = t;
    r->seqs = sq;
    r->N = N;
    r->k = k;
    if (t) ++t->refs;
    else ++ctx->refs;
    ++const_cast<pg_seqset *>(sq)->refs;
    r->flags = flags;
    r->lowres_step = geo.lowres_step;
    r->d_ad = nullptr;
    r->d_tile_contig = nullptr;
    r->d_out1 = r->d_out100 = nullptr;
    r->d_bins = nullptr;
    r->d_colsums = nullptr;
    r->ev_ok = r->ev_epi = false;
    for (auto &e : r->ev) e = nullptr;
    uint64_t o1 = 0, o100 = 0, bins = 0, tiles = 0;
    std::vector<uint32_t> tile_contig;
    for (uint32_t c = 0; c < sq->n; ++c) {
        const uint64_t len = sq->desc[c].len;
        const uint64_t nk = len >= (uint64_t)k ? len - k + 1 : 0;
        if (nk > 0xFFFFFFF0ull) {
            pg_result_destroy(r);
            return fail(PG_E_INVALID, "contig %u has %llu k-mers; contigs must stay below 2^32 (as in KMC)", c, (unsigned long long)nk);
        }
        AnchorDesc a;
        a.nkmers = (uint32_t)nk;
        // cpp/anchor.cpp:114-118 / index.py:1169-1172; contigs with fewer k-mers than min_bin_count make the
        // reference divide by zero — here they get one bin per k-mer (documented deviation)
        uint64_t binlen = geo.max_bin_len;
        if (nk / binlen < geo.min_bin_count) binlen = nk / geo.min_bin_count;
        if (binlen == 0) binlen = 1;
        a.binlen = (uint32_t)binlen;
        a.nbins = (uint32_t)((nk + binlen - 1) / binlen);
        a.out_off = o1;
        a.out100_off = o100;
        a.bin_off = bins;
        a.tile0 = (uint32_t)tiles;
        const uint64_t n100 = (nk + geo.lowres_step - 1) / geo.lowres_step;
        r->nrows100.push_back(n100);
        o1 += (nk * nbytes + 15) & ~15ull;
        o100 += n100 * nbytes;
        bins += a.nbins;
        const uint64_t nt = (nk + PROBE_TILE - 1) / PROBE_TILE;
        for (uint64_t i = 0; i < nt; ++i) tile_contig.push_back(c);
        tiles += nt;
        r->ad.push_back(a);
    }
    if (tiles > 0x7FFFFFFFull) {
        pg_result_destroy(r);
        return fail(PG_E_INVALID, "too many tiles in one launch");
    }
    r->ntiles = (uint32_t)tiles;
    r->out1_bytes = o1;
    r->out100_bytes = o100;
    r->total_bins = bins;
    hipStream_t st = ctx->stream;
    hipError_t e;
    if ((e = hipMalloc(reinterpret_cast<void **>(&r->d_ad), std::max<size_t>(1, r->ad.size()) * sizeof(AnchorDesc))) == hipSuccess &&
        (e = hipMalloc(reinterpret_cast<void **>(&r->d_tile_contig), std::max<size_t>(1, tile_contig.size()) * 4)) == hipSuccess &&
        // (+ 16: k_epilogue_chunks reads whole 16-byte chunks, the last one of the last row may start at its last byte)
        (e = row_alloc(ctx, (flags & PG_ANCHOR_COLUMNS_ONLY) ? 16 : std::max<uint64_t>(16, o1) + 16, &r->d_out1, &r->out1_cap)) == hipSuccess &&
        (e = hipMalloc(reinterpret_cast<void **>(&r->d_out100), std::max<uint64_t>(16, o100))) == hipSuccess &&
        (e = hipMalloc(reinterpret_cast<void **>(&r->d_bins), std::max<uint64_t>(1, bins) * (N + 1) * 4)) == hipSuccess &&
        (e = hipMalloc(reinterpret_cast<void **>(&r->d_colsums), std::max<size_t>(1, r->ad.size()) * N * 8)) == hipSuccess) {
        if (!r->ad.empty())
            e = hipMemcpyAsync(r->d_ad, r->ad.data(), r->ad.size() * sizeof(AnchorDesc), hipMemcpyHostToDevice, st);
        if (e == hipSuccess && !tile_contig.empty())
            e = hipMemcpyAsync(r->d_tile_contig, tile_contig.data(), tile_contig.size() * 4, hipMemcpyHostToDevice, st);
        // a rows container is filled block by block (pg_result_merge_columns_range with accumulate): start from zero
        if (e == hipSuccess && !t) e = hipMemsetAsync(r->d_out1, 0, std::max<uint64_t>(16, o1), st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) {
        pg_result_destroy(r);
        return fail(PG_E_HIP, "result allocation failed: %s", hipGetErrorString(e));
    }
    *out = r;
    return PG_OK;
}

extern "C" int pg_result_create(pg_table *t, const pg_seqset *sq, uint32_t flags, pg_result **out) {
    PG_API_BEGIN
    if (!t || !sq || !out) return fail(PG_E_INVALID, "pg_result_create: NULL argument");
    return result_create(t->ctx, t, t->k, (uint32_t)t->ngenomes, sq, flags, ResultGeom(), out);
    PG_API_END
}

extern "C" int pg_result_create_ex(pg_table *t, const pg_seqset *sq, uint32_t flags, uint32_t lowres_step,
                                   uint64_t max_bin_len, uint32_t min_bin_count, pg_result **out) {
    PG_API_BEGIN
    if (!t || !sq || !out) return fail(PG_E_INVALID, "pg_result_create_ex: NULL argument");
    ResultGeom g;
    g.lowres_step = lowres_step;
    g.max_bin_len = max_bin_len;
    g.min_bin_count = min_bin_count;
    return result_create(t->ctx, t, t->k, (uint32_t)t->ngenomes, sq, flags, g, out);
    PG_API_END
}

extern "C" int pg_result_create_rows(pg_ctx *ctx, int k, int ngenomes, const pg_seqset *sq, uint32_t flags,
                                     uint32_t lowres_step, uint64_t max_bin_len, uint32_t min_bin_count, pg_result **out) {
    PG_API_BEGIN
    if (!ctx || !sq || !out) return fail(PG_E_INVALID, "pg_result_create_rows: NULL argument");
    if (k < 1 || k > 32) return fail(PG_E_INVALID, "k=%d unsupported (1..32)", k);
    if (ngenomes < 1) return fail(PG_E_INVALID, "ngenomes must be >= 1");
    ResultGeom g;
    g.lowres_step = lowres_step;
    g.max_bin_len = max_bin_len;
    g.min_bin_count = min_bin_count;
    return result_create(ctx, nullptr, k, (uint32_t)ngenomes, sq, flags | PG_ANCHOR_ROWS_ONLY, g, out);
    PG_API_END
}

extern "C" int pg_result_destroy(pg_result *r) {
    PG_API_BEGIN
    if (!r) return PG_OK;
    hipSetDevice(r->ctx->device);
    hipStreamSynchronize(r->ctx->stream);
    hipStreamSynchronize(r->ctx->aux_stream);
    hipFree(r->d_ad);
    hipFree(r->d_tile_contig);
    row_free(r->ctx, r->d_out1, r->out1_cap);
    hipFree(r->d_out100);
    hipFree(r->d_bins);
    hipFree(r->d_colsums);
    if (r->d_sched) hipFree(r->d_sched);
    if (r->d_ranges) hipFree(r->d_ranges);
    if (r->d_tile_hist) hipFree(r->d_tile_hist);
    if (r->d_tile_cs) hipFree(r->d_tile_cs);
    if (r->d_small) hipFree(r->d_small);
    for (auto e : r->chunk_ev) hipEventDestroy(e);
    for (auto *v : {&r->ev_hist, &r->ev_free})
        for (auto &s : *v)
            for (auto &e : s.e)
                if (e) hipEventDestroy(e);
    pg_table *t = r->tbl;
    pg_ctx *c = r->ctx;
    pg_seqset *sq = const_cast<pg_seqset *>(r->seqs);
    delete r;
    if (t) table_release(t);
    else ctx_release(c);
    seqset_release(sq);
    return PG_OK;
    PG_API_END
}

// statistics kernels of a result, on stream `st`
// ---------------------------------------------------------------------------
// co-scheduling: the anchor genomes of one pangenome are homologous, so the same table lines are
// needed at corresponding places of every genome.  Launching the tiles genome by genome fetches each
// line from HBM once per genome; interleaving the genomes piece by piece (pieces of `piece_tiles`
// tiles, every genome traversed at the same relative pace) lets the later genomes find the lines in
// L2 / Infinity Cache.  Only the launch order changes — results are identical for any schedule.
// ---------------------------------------------------------------------------
// The chunks of a whole run (pg_result::Chunk): the launch order — `sched`, or tile order when NULL — is cut into
// slices of at least 32768 tiles (four rounds of waves on the device: shorter probe launches
// lose more in their tails than the overlap wins); per slice the tiles it touches, as sorted disjoint ranges — a slice
// of the co-schedule covers one stretch of every genome.  A slice that touches more than 512 ranges (thousands of short
// contigs) is not worth a launch of its own: the run then stays one probe launch and one statistics pass.
static int build_chunks(pg_result *r, const uint32_t *sched) {
    r->chunks.clear();
    r->chunks_ready = true;
    if (r->d_ranges) {
        hipFree(r->d_ranges);
        r->d_ranges = nullptr;
    }
    // Default: only runs of at least 4 M tiles with 8-byte rows go out in (16) chunks — measured
    // (profiles/r3_ab_chunked_overlap.txt): 64 x 200 Mb, 8-byte rows 105.4 -> 100.8 ms per step; everywhere else the probe
    // slows down by about what the pass takes when the two run side by side (8 x 100 Mb 4.86 -> 5.2 ms at any chunk
    // count, 64 x 20 Mb 8.93 -> 8.93), i.e. the statistics pass cannot be hidden behind the probe.  PG_RUN_CHUNKS pins
    // the count (1: never).
    const char *env = getenv("PG_RUN_CHUNKS");
    const uint32_t nbytes = (r->N + 7) / 8;
    uint32_t K = env ? (uint32_t)atoi(env) : ((nbytes == 8 && r->ntiles >= (4u << 20)) ? 16u : 1u);
    const char *mt = getenv("PG_CHUNK_MIN_TILES");  // (tests: chunks of a few tiles)
    K = std::min(K, r->ntiles / std::max(1u, mt ? (uint32_t)atoi(mt) : 32768u));
    if (K < 2) return PG_OK;
    std::vector<uint2> ranges;
    for (uint32_t c = 0; c < K; ++c) {
        pg_result::Chunk ch;
        ch.s0 = (uint32_t)((uint64_t)r->ntiles * c / K);
        ch.s1 = (uint32_t)((uint64_t)r->ntiles * (c + 1) / K);
        ch.r0 = (uint32_t)ranges.size();
        ch.tiles = ch.s1 - ch.s0;
        if (!sched) {
            ranges.push_back(make_uint2(ch.s0, ch.s1));
        } else {
            std::vector<uint2> runs;  // maximal runs of consecutive tiles in the slice (a piece of a genome each)
            for (uint32_t i = ch.s0; i < ch.s1; ++i) {
                if (!runs.empty() && runs.back().y == sched[i]) ++runs.back().y;
                else runs.push_back(make_uint2(sched[i], sched[i] + 1));
            }
            std::sort(runs.begin(), runs.end(), [](const uint2 &a, const uint2 &b) { return a.x < b.x; });
            for (const uint2 &u : runs) {
                if (ranges.size() > ch.r0 && ranges.back().y == u.x) ranges.back().y = u.y;
                else ranges.push_back(u);
            }
        }
        ch.nr = (uint32_t)ranges.size() - ch.r0;
        if (ch.nr > 512) {
            r->chunks.clear();
            return PG_OK;
        }
        r->chunks.push_back(ch);
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&r->d_ranges), ranges.size() * sizeof(uint2)));
    HIP_TRY(hipMemcpy(r->d_ranges, ranges.data(), ranges.size() * sizeof(uint2), hipMemcpyHostToDevice));
    while (r->chunk_ev.size() < r->chunks.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        r->chunk_ev.push_back(e);
    }
    return PG_OK;
}

static int coschedule(pg_result *r, const uint32_t *contig_group, uint32_t piece_tiles, const uint32_t *range_first,
                      uint32_t nranges, const uint32_t *contig_class = nullptr) {
    if (int e = use_device(r->ctx)) return e;
    hipStream_t st = r->ctx->stream;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipStreamSynchronize(r->ctx->aux_stream));
    if (r->d_sched) {
        hipFree(r->d_sched);
        r->d_sched = nullptr;
    }
    r->sched_bounds.clear();
    r->chunks.clear();
    r->chunks_ready = false;  // (launch order again until a schedule is set below: chunks are cut at the next run)
    if (!contig_group || r->ntiles == 0) return PG_OK;  // NULL: back to launch order
    // Default piece: 64 tiles for up to 32 genomes; 512 / genomes beyond, never under 8.  With 64-tile pieces a locus of 64
    // genomes spanned 4096 tiles of the launch order — half the waves the device holds at once — and the later genomes
    // found fewer of its lines in L2: 64 x 40 Mb 14.5 -> 13.6 ms, 128 x 10 Mb 11.2 -> 10.8, 40 x 30 Mb 7.54 -> 7.37; 8 to 27
    // genomes are best at 64 or indifferent (16 x 60 Mb: 4.35 ms at 64, 4.42 at 16, 4.47 at 8; profiles/r4b_piece_sweep.txt).
    const bool auto_piece = piece_tiles == 0;
    const size_t nc = r->ad.size();
    // contig ranges scheduled independently of each other (default: one range = the whole result)
    std::vector<uint32_t> firsts;
    if (range_first && nranges) {
        for (uint32_t i = 0; i < nranges; ++i) {
            if (range_first[i] >= nc || (i && range_first[i] <= range_first[i - 1]) || (!i && range_first[0] != 0))
                return fail(PG_E_INVALID, "pg_result_coschedule_ranges: range starts must begin at contig 0 and increase");
            firsts.push_back(range_first[i]);
        }
    } else {
        firsts.push_back(0);
    }
    firsts.push_back((uint32_t)nc);
    struct Piece {
        double at;
        uint32_t g, first;
    };
    std::vector<uint32_t> sched;
    sched.reserve(r->ntiles);
    // classes of contigs (homologous chromosomes: the same class in every genome, whatever order a genome lists them
    // in) are scheduled one after the other, each like a range of its own; without classes a range is one class
    std::vector<std::vector<uint32_t>> members;  // contigs of every (range, class), in contig order
    if (contig_class) {
        std::vector<std::pair<uint32_t, uint32_t>> byclass;
        for (size_t c = 0; c < nc; ++c) byclass.emplace_back(contig_class[c], (uint32_t)c);
        std::stable_sort(byclass.begin(), byclass.end(), [](auto &x, auto &y) { return x.first < y.first; });
        for (size_t i = 0; i < byclass.size(); ++i) {
            if (i == 0 || byclass[i].first != byclass[i - 1].first) members.emplace_back();
            members.back().push_back(byclass[i].second);
        }
        r->sched_bounds.push_back(0);
    } else {
        for (size_t ri = 0; ri + 1 < firsts.size(); ++ri) {
            members.emplace_back();
            for (uint32_t c = firsts[ri]; c < firsts[ri + 1]; ++c) members.back().push_back(c);
            r->sched_bounds.push_back(r->ad[firsts[ri]].tile0);
        }
    }
    for (const auto &mem : members) {
        uint32_t ngroups = 0;
        for (uint32_t c : mem) ngroups = std::max(ngroups, contig_group[c] + 1);
        std::vector<std::vector<uint32_t>> tiles(ngroups);  // every group's tiles, contig after contig
        for (uint32_t c : mem) {
            const uint32_t nt = (uint32_t)(((uint64_t)r->ad[c].nkmers + PROBE_TILE - 1) / PROBE_TILE);
            for (uint32_t i = 0; i < nt; ++i) tiles[contig_group[c]].push_back(r->ad[c].tile0 + i);
        }
        if (auto_piece) piece_tiles = ngroups <= 32 ? 64u : std::max(8u, 512u / ngroups);
        std::vector<Piece> pieces;
        for (uint32_t g = 0; g < ngroups; ++g) {
            const size_t np = (tiles[g].size() + piece_tiles - 1) / piece_tiles;
            for (size_t i = 0; i < np; ++i) pieces.push_back({(double)i / (double)np, g, (uint32_t)(i * piece_tiles)});
        }
        std::stable_sort(pieces.begin(), pieces.end(), [](const Piece &x, const Piece &y) { return x.at < y.at; });
        for (const Piece &p : pieces) {
            const size_t end = std::min<size_t>(tiles[p.g].size(), (size_t)p.first + piece_tiles);
            for (size_t i = p.first; i < end; ++i) sched.push_back(tiles[p.g][i]);
        }
    }
    r->sched_bounds.push_back(r->ntiles);
    if (sched.size() != r->ntiles) return fail(PG_E_INVALID, "internal: schedule does not cover the tiles");
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&r->d_sched), (size_t)r->ntiles * 4));
    HIP_TRY(hipMemcpyAsync(r->d_sched, sched.data(), (size_t)r->ntiles * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (!(r->flags & PG_ANCHOR_ROWS_ONLY)) return build_chunks(r, sched.data());
    return PG_OK;
}

extern "C" int pg_result_coschedule(pg_result *r, const uint32_t *contig_group, uint32_t piece_tiles) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    return coschedule(r, contig_group, piece_tiles, nullptr, 0);
    PG_API_END
}

extern "C" int pg_result_coschedule_classes(pg_result *r, const uint32_t *contig_group, const uint32_t *contig_class,
                                            uint32_t piece_tiles) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    if (contig_group && !contig_class) return fail(PG_E_INVALID, "pg_result_coschedule_classes: contig_class is NULL");
    return coschedule(r, contig_group, piece_tiles, nullptr, 0, contig_class);
    PG_API_END
}

extern "C" int pg_result_coschedule_ranges(pg_result *r, const uint32_t *contig_group, uint32_t piece_tiles,
                                           const uint32_t *range_first_contig, uint32_t nranges) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    return coschedule(r, contig_group, piece_tiles, range_first_contig, nranges);
    PG_API_END
}

static int enqueue_epilogue(pg_result *r, hipStream_t st) {
    const uint32_t N = r->N;
    HIP_TRY(hipMemsetAsync(r->d_bins, 0, std::max<uint64_t>(1, r->total_bins) * (N + 1) * 4, st));
    HIP_TRY(hipMemsetAsync(r->d_colsums, 0, std::max<size_t>(1, r->ad.size()) * N * 8, st));
    // the statistics kernels fuse the 1-in-100 rows; any other step is a small gather of its own
    const uint32_t kflags = (r->flags & PG_ANCHOR_COLSUMS) | (r->lowres_step == 100 ? 0u : 2u);
    HIP_TRY(launch_rows_epilogue(st, N, r->d_ad, r->d_tile_contig, r->ntiles, r->d_out1, r->d_out100, r->d_bins,
                                 r->d_colsums, kflags));
    if (r->lowres_step != 100)
        HIP_TRY(launch_lowres(st, N, r->d_ad, r->d_tile_contig, r->ntiles, r->d_out1, r->d_out100, r->lowres_step));
    return PG_OK;
}

// ---- fused statistics: does this result qualify, and its buffers ----
// OFF unless PG_FUSE_STATS=1: measured (profiles/r6f_ab_fuse.txt, r6e_phase_fused.txt) the tile's end costs k_probe more than
// the statistics pass it replaces — rows of 2 / 4 / 8 bytes: step 3.47 -> 3.70, 5.73 -> 5.97, 9.0 -> 9.56 ms — because every
// instruction of it is issued in the one kernel that is short of issue slots, while the pass runs the same arithmetic in the
// shadow of its HBM stream.  Read at every run, so that one process can time both.
static bool fuse_enabled() {
    const char *e = getenv("PG_FUSE_STATS");
    return e && *e == '1';
}
static bool fuse_table_ok(const TableDesc &T, uint32_t N) {
    return T.nsub == 1 && T.ngenomes == N && !(T.sub[0].layout == LAYOUT_SLOTS && T.sub[0].slots != 8);
}
static int prepare_fuse(pg_result *r, const TableDesc &T) {
    if (r->fuse_state) return PG_OK;
    r->fuse_state = -1;
    const uint32_t nbytes = (r->N + 7) / 8;
    // rows of 2..16 bytes written whole by ONE sub-table of 128-byte lines (one-byte rows: k_probe is bound by instruction issue
    // there and their bit-sliced pass costs 8 % of the step; rows beyond 16 bytes: k_epilogue_chunks)
    if (!fuse_rows_ok(nbytes) || !fuse_table_ok(T, r->N)) return PG_OK;
    std::vector<uint2> small;  // tile ranges of the contigs whose bins are shorter than a tile: the statistics pass's
    uint64_t small_tiles = 0, fused_tiles = 0;
    for (const AnchorDesc &a : r->ad) {
        const uint32_t nt = (uint32_t)(((uint64_t)a.nkmers + PROBE_TILE - 1) / PROBE_TILE);
        if (!nt) continue;
        if (a.binlen >= (uint32_t)PROBE_TILE) {
            fused_tiles += nt;
            continue;
        }
        small_tiles += nt;
        if (!small.empty() && small.back().y == a.tile0) small.back().y = a.tile0 + nt;
        else small.push_back(make_uint2(a.tile0, a.tile0 + nt));
    }
    if (!fused_tiles || small.size() > 16384) return PG_OK;  // (thousands of separate short-bin stretches: one pass over everything is simpler)
    const size_t hb = (size_t)r->ntiles * fuse_hist_words(r->N) * 4, cb = (size_t)r->ntiles * fuse_cs_words(r->N) * 4;
    if (hipMalloc(reinterpret_cast<void **>(&r->d_tile_hist), hb) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&r->d_tile_cs), cb) != hipSuccess ||
        (!small.empty() && hipMalloc(reinterpret_cast<void **>(&r->d_small), small.size() * sizeof(uint2)) != hipSuccess)) {
        (void)hipGetLastError();  // no room for the tiles' counters (0.4-6 % of the rows): the unfused path needs none
        if (r->d_tile_hist) hipFree(r->d_tile_hist);
        if (r->d_tile_cs) hipFree(r->d_tile_cs);
        r->d_tile_hist = r->d_tile_cs = nullptr;
        return PG_OK;
    }
    if (!small.empty()) HIP_TRY(hipMemcpy(r->d_small, small.data(), small.size() * sizeof(uint2), hipMemcpyHostToDevice));
    r->n_small = (uint32_t)small.size();
    r->small_tiles = (uint32_t)small_tiles;
    r->fuse_state = 1;
    return PG_OK;
}
static FuseArgs fuse_args(const pg_result *r) {
    FuseArgs fa;
    fa.out100 = r->lowres_step == 100 ? r->d_out100 : nullptr;  // (any other step: k_lowres)
    fa.tile_hist = r->d_tile_hist;
    fa.tile_cs = r->d_tile_cs;
    fa.ngenomes = r->N;
    fa.hw = fuse_hist_words(r->N);
    fa.csw = fuse_cs_words(r->N);
    return fa;
}
// what is left of the statistics after a fused probe: the tiles' counters added up, the short-bin contigs' rows through the pass
static int enqueue_fused_epilogue(pg_result *r, hipStream_t st, const FuseArgs &fa) {
    const uint32_t N = r->N;
    HIP_TRY(hipMemsetAsync(r->d_bins, 0, std::max<uint64_t>(1, r->total_bins) * (N + 1) * 4, st));
    HIP_TRY(hipMemsetAsync(r->d_colsums, 0, std::max<size_t>(1, r->ad.size()) * N * 8, st));
    HIP_TRY(launch_tile_reduce(st, fa, r->d_ad, r->d_tile_contig, r->ntiles, r->d_bins, r->d_colsums, (r->flags & PG_ANCHOR_COLSUMS) ? 1u : 0u));
    const uint32_t kflags = (r->flags & PG_ANCHOR_COLSUMS) | (r->lowres_step == 100 ? 0u : 2u);
    if (r->n_small)
        HIP_TRY(launch_rows_epilogue(st, N, r->d_ad, r->d_tile_contig, r->ntiles, r->d_out1, r->d_out100, r->d_bins, r->d_colsums, kflags,
                                     r->d_small, r->n_small, r->small_tiles));
    if (r->lowres_step != 100)
        HIP_TRY(launch_lowres(st, N, r->d_ad, r->d_tile_contig, r->ntiles, r->d_out1, r->d_out100, r->lowres_step));
    return PG_OK;
}

// make the context's main stream wait for the result's statistics (side stream)
static int join_result(pg_result *r) {
    if (r->ev_epi) HIP_TRY(hipStreamWaitEvent(r->ctx->stream, r->ev[3], 0));
    return PG_OK;
}

// durations of a finished event set into the running sums
static int fold_timing(pg_result *r, const pg_result::EvSet &s) {
    HIP_TRY(hipEventSynchronize(s.e[s.epi ? 3 : 1]));
    float a = 0, b = 0;
    if (s.probe) {
        HIP_TRY(hipEventElapsedTime(&a, s.e[0], s.e[1]));
        r->probe_ms_sum += a;
        ++r->probe_runs;
    }
    if (s.epi) {
        HIP_TRY(hipEventElapsedTime(&b, s.e[2], s.e[3]));
        r->epi_ms_sum += b;
        ++r->epi_runs;
    }
    return PG_OK;
}

// a fresh event set for the work about to be enqueued; r->ev[] point at it
static int next_events(pg_result *r, bool probe) {
    if (r->ev_hist.size() >= EV_RING) {  // the oldest run finished long ago: fold it, reuse its events
        if (r->hist_skip) --r->hist_skip;
        else if (int e = fold_timing(r, r->ev_hist.front())) return e;
        r->ev_free.push_back(r->ev_hist.front());
        r->ev_hist.erase(r->ev_hist.begin());
    }
    pg_result::EvSet s;
    if (!r->ev_free.empty()) {
        s = r->ev_free.back();
        r->ev_free.pop_back();
    } else {
        for (auto &e : s.e) {
            e = nullptr;
            HIP_TRY(hipEventCreate(&e));
        }
    }
    s.probe = probe;
    s.epi = false;
    r->ev_hist.push_back(s);
    for (int i = 0; i < 4; ++i) r->ev[i] = s.e[i];
    return PG_OK;
}

// tiles of contigs [first, first + n)
static int contig_tiles(const pg_result *r, uint32_t first, uint32_t n, uint32_t *t0, uint32_t *nt) {
    if ((uint64_t)first + n > r->ad.size()) return fail(PG_E_INVALID, "contigs %u..%u out of range (the result has %zu)", first, first + n, r->ad.size());
    if (n == 0) {
        *t0 = *nt = 0;
        return PG_OK;
    }
    const AnchorDesc &a = r->ad[first], &z = r->ad[first + n - 1];
    *t0 = a.tile0;
    *nt = z.tile0 + (uint32_t)(((uint64_t)z.nkmers + PROBE_TILE - 1) / PROBE_TILE) - a.tile0;
    return PG_OK;
}

// may the launch over tiles [t0, t0 + nt) follow the schedule?  (it is a permutation within each scheduled range)
static bool sched_covers(const pg_result *r, uint32_t t0, uint32_t nt) {
    if (!r->d_sched) return false;
    const auto &b = r->sched_bounds;
    return std::binary_search(b.begin(), b.end(), t0) && std::binary_search(b.begin(), b.end(), t0 + nt);
}

// A whole run in chunks: probe(chunk c) on the main stream, the statistics of chunk c on the side stream behind it —
// i.e. beside probe(chunk c + 1).  The two kernels want different things of the machine (k_probe: instruction issue and
// random L2 lines; the statistics pass: streaming HBM reads), and the pass's rows are the ones just written.  Timing:
// e[0]..e[1] spans the probe launches (back to back on their stream: the sum of their durations), e[2]..e[3] the
// statistics side from its first kernel's start to its last one's end.
static int run_chunks(pg_result *r, const TableDesc &T) {
    hipStream_t st = r->ctx->stream, aux = r->ctx->aux_stream;
    if (const char *e = getenv("PG_CHUNK_SAME_STREAM"); e && *e == '1') aux = st;  // (experiment: the statistics of a chunk right behind its probe, nothing side by side)
    const uint32_t N = r->N;
    const uint32_t kflags = (r->flags & PG_ANCHOR_COLSUMS) | (r->lowres_step == 100 ? 0u : 2u);
    HIP_TRY(hipEventRecord(r->ev[0], st));
    for (size_t c = 0; c < r->chunks.size(); ++c) {
        const pg_result::Chunk &ch = r->chunks[c];
        HIP_TRY(launch_anchor(st, T, r->seqs->d_seqw, r->seqs->d_nmw, r->seqs->d_has_n, r->seqs->d_desc, r->d_ad, r->d_tile_contig,
                              r->d_sched, ch.s0, ch.s1 - ch.s0, r->d_out1, r->out1_bytes, 0));
        HIP_TRY(hipEventRecord(r->chunk_ev[c], st));
        HIP_TRY(hipStreamWaitEvent(aux, r->chunk_ev[c], 0));
        if (c == 0) {
            HIP_TRY(hipEventRecord(r->ev[2], aux));
            HIP_TRY(hipMemsetAsync(r->d_bins, 0, std::max<uint64_t>(1, r->total_bins) * (N + 1) * 4, aux));
            HIP_TRY(hipMemsetAsync(r->d_colsums, 0, std::max<size_t>(1, r->ad.size()) * N * 8, aux));
        }
        HIP_TRY(launch_rows_epilogue(aux, N, r->d_ad, r->d_tile_contig, r->ntiles, r->d_out1, r->d_out100, r->d_bins, r->d_colsums,
                                     kflags, r->d_ranges + ch.r0, ch.nr, ch.tiles));
    }
    HIP_TRY(hipEventRecord(r->ev[1], st));
    if (r->lowres_step != 100)
        HIP_TRY(launch_lowres(aux, N, r->d_ad, r->d_tile_contig, r->ntiles, r->d_out1, r->d_out100, r->lowres_step));
    HIP_TRY(hipEventRecord(r->ev[3], aux));
    r->ev_ok = r->ev_epi = true;
    r->ev_hist.back().epi = true;
    return PG_OK;
}

static int anchor_run(pg_result *r, uint32_t tile_base, uint32_t ntiles, bool whole, uint32_t columns_width = 0,
                      void *d_columns = nullptr) {
    pg_table *t = r->tbl;
    if (!t) return fail(PG_E_INVALID, "this result is a rows container (pg_result_create_rows): it has no table to probe");
    if (int e = use_device(r->ctx)) return e;
    hipStream_t st = r->ctx->stream, aux = r->ctx->aux_stream;
    if (int e = join_result(r)) return e;  // a previous run's statistics still read the rows we overwrite
    if (int e = next_events(r, true)) return e;
    TableDesc T = make_desc(t);
    const bool stats = whole && !columns_width && !(r->flags & PG_ANCHOR_ROWS_ONLY);
    bool fuse = false;  // the tiles' statistics inside k_probe (round 6): a whole run of a result that qualifies
    if (stats && fuse_enabled()) {
        if (int e = prepare_fuse(r, T)) return e;
        fuse = r->fuse_state == 1 && fuse_table_ok(T, r->N);  // (the table may have been re-hashed into other lines since)
    }
    if (stats && !fuse) {
        if (!r->chunks_ready)
            if (int e = build_chunks(r, nullptr)) return e;
        if (r->chunks.size() > 1) return run_chunks(r, T);
    }
    const FuseArgs fa = fuse_args(r);
    HIP_TRY(hipEventRecord(r->ev[0], st));
    HIP_TRY(launch_anchor(st, T, r->seqs->d_seqw, r->seqs->d_nmw, r->seqs->d_has_n, r->seqs->d_desc, r->d_ad,
                          r->d_tile_contig, sched_covers(r, tile_base, ntiles) ? r->d_sched : nullptr, tile_base, ntiles,
                          columns_width ? static_cast<uint8_t *>(d_columns) : r->d_out1, r->out1_bytes, columns_width, fuse ? &fa : nullptr));
    HIP_TRY(hipEventRecord(r->ev[1], st));
    r->ev_ok = true;
    r->ev_epi = false;
    if (stats) {
        // the streaming statistics pass runs on the side stream, ordered behind the probe kernels by
        // an event, so that it overlaps the next result's probe kernels
        HIP_TRY(hipStreamWaitEvent(aux, r->ev[1], 0));
        HIP_TRY(hipEventRecord(r->ev[2], aux));
        if (fuse) {
            if (int e = enqueue_fused_epilogue(r, aux, fa)) return e;
            ++r->fused_runs;
        } else if (int e = enqueue_epilogue(r, aux)) return e;
        HIP_TRY(hipEventRecord(r->ev[3], aux));
        r->ev_epi = true;
        r->ev_hist.back().epi = true;
    }
    return PG_OK;
}

extern "C" int pg_anchor_run(pg_result *r) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    if (r->flags & PG_ANCHOR_COLUMNS_ONLY) return fail(PG_E_INVALID, "a PG_ANCHOR_COLUMNS_ONLY result has no rows: use pg_anchor_run_columns_range");
    return anchor_run(r, 0, r->ntiles, true);
    PG_API_END
}

extern "C" int pg_anchor_run_range(pg_result *r, uint32_t first_contig, uint32_t ncontigs) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    if (r->flags & PG_ANCHOR_COLUMNS_ONLY) return fail(PG_E_INVALID, "a PG_ANCHOR_COLUMNS_ONLY result has no rows: use pg_anchor_run_columns_range");
    if (!(r->flags & PG_ANCHOR_ROWS_ONLY)) return fail(PG_E_INVALID, "pg_anchor_run_range needs a PG_ANCHOR_ROWS_ONLY result");
    uint32_t t0, nt;
    if (int e = contig_tiles(r, first_contig, ncontigs, &t0, &nt)) return e;
    return anchor_run(r, t0, nt, false);
    PG_API_END
}

extern "C" int pg_result_columns_direct(const pg_result *r, uint32_t width) {
    PG_API_BEGIN
    if (!r || !r->tbl) return 0;
    const pg_table *t = r->tbl;
    return t->subs.size() == 1 && t->subs[0].d.layout == LAYOUT_SLOTS && t->subs[0].d.W == 1 && t->subs[0].d.slots == 8 &&
           t->ngenomes <= 8 && width <= 8 && (int)width >= t->ngenomes;
    PG_API_END
}

extern "C" int pg_anchor_run_columns_range(pg_result *r, uint32_t first_contig, uint32_t ncontigs, uint32_t width, void *d_dst) {
    PG_API_BEGIN
    if (!r || !d_dst) return fail(PG_E_INVALID, "pg_anchor_run_columns_range: NULL argument");
    if (!pg_result_columns_direct(r, width))
        return fail(PG_E_INVALID, "pg_anchor_run_columns_range: needs a table of up to 8 genomes (8-slot lines) and width in ngenomes..8");
    uint32_t t0, nt;
    if (int e = contig_tiles(r, first_contig, ncontigs, &t0, &nt)) return e;
    return anchor_run(r, t0, nt, false, width, d_dst);
    PG_API_END
}

extern "C" int pg_result_timing(pg_result *r, float *probe_ms, float *epilogue_ms) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    if (!r->ev_ok) return fail(PG_E_INVALID, "pg_anchor_run has not been called on this result");
    if (int e = use_device(r->ctx)) return e;
    HIP_TRY(hipEventSynchronize(r->ev[r->ev_epi ? 3 : 1]));
    float a = 0, b = 0;
    HIP_TRY(hipEventElapsedTime(&a, r->ev[0], r->ev[1]));
    if (r->ev_epi) HIP_TRY(hipEventElapsedTime(&b, r->ev[2], r->ev[3]));
    if (probe_ms) *probe_ms = a;
    if (epilogue_ms) *epilogue_ms = b;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_result_fused_runs(const pg_result *r, uint32_t *n) {
    PG_API_BEGIN
    if (!r || !n) return fail(PG_E_INVALID, "pg_result_fused_runs: NULL argument");
    *n = r->fused_runs;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_result_timing_reset(pg_result *r) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    // the latest set stays alive (r->ev[] and the writers' stream waits refer to it) but no longer counts
    while (r->ev_hist.size() > 1) {
        r->ev_free.push_back(r->ev_hist.front());
        r->ev_hist.erase(r->ev_hist.begin());
    }
    r->hist_skip = r->ev_hist.size();
    r->probe_ms_sum = r->epi_ms_sum = 0;
    r->probe_runs = r->epi_runs = 0;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_result_timing_mean(pg_result *r, double *probe_ms, double *epilogue_ms, uint32_t *nruns) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    if (int e = use_device(r->ctx)) return e;
    // sums of the sets that left the ring + the sets still in it (waits for the last of them)
    const double a0 = r->probe_ms_sum, b0 = r->epi_ms_sum;
    const uint32_t pn0 = r->probe_runs, en0 = r->epi_runs;
    int rc = PG_OK;
    for (size_t i = r->hist_skip; i < r->ev_hist.size() && !rc; ++i) rc = fold_timing(r, r->ev_hist[i]);
    if (!rc) {
        if (probe_ms) *probe_ms = r->probe_runs ? r->probe_ms_sum / r->probe_runs : 0.0;
        if (epilogue_ms) *epilogue_ms = r->epi_runs ? r->epi_ms_sum / r->epi_runs : 0.0;
        if (nruns) *nruns = r->probe_runs;
    }
    r->probe_ms_sum = a0;  // (the ring's sets are folded for good only when they leave it)
    r->epi_ms_sum = b0;
    r->probe_runs = pn0;
    r->epi_runs = en0;
    return rc;
    PG_API_END
}

// ---------------------------------------------------------------------------
// genome-sharded exchange: compact bit columns out of / back into the rows
// ---------------------------------------------------------------------------
extern "C" uint64_t pg_result_columns_bytes(const pg_result *r, uint32_t width) {
    return r ? (uint64_t)r->ntiles * (PROBE_TILE / 8) * width : 0;  // (PROBE_TILE / 64 u64 words per genome and tile)
}

extern "C" uint64_t pg_result_columns_bytes_range(const pg_result *r, uint32_t width, uint32_t first_contig, uint32_t ncontigs) {
    uint32_t t0, nt;
    if (!r || contig_tiles(r, first_contig, ncontigs, &t0, &nt)) return 0;
    return (uint64_t)nt * (PROBE_TILE / 8) * width;
}

extern "C" int pg_result_extract_columns_range(pg_result *r, uint32_t g0, uint32_t width, uint32_t first_contig,
                                               uint32_t ncontigs, void *d_dst) {
    PG_API_BEGIN
    if (!r || !d_dst) return fail(PG_E_INVALID, "pg_result_extract_columns: NULL argument");
    if (!r->ev_ok && !r->rows_valid) return fail(PG_E_INVALID, "the result holds no rows yet");
    uint32_t t0, nt;
    if (int e = contig_tiles(r, first_contig, ncontigs, &t0, &nt)) return e;
    if (int e = use_device(r->ctx)) return e;
    HIP_TRY(launch_cols_extract(r->ctx->stream, r->N, r->d_ad, r->d_tile_contig, t0, nt, r->d_out1, g0, width, d_dst));
    return PG_OK;
    PG_API_END
}

extern "C" int pg_result_extract_columns(pg_result *r, uint32_t g0, uint32_t width, void *d_dst) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "pg_result_extract_columns: NULL argument");
    return pg_result_extract_columns_range(r, g0, width, 0, (uint32_t)r->ad.size(), d_dst);
    PG_API_END
}

extern "C" int pg_result_merge_columns_range(pg_result *r, const void *d_src, uint32_t part0, uint32_t nparts, uint32_t per,
                                             uint32_t first_contig, uint32_t ncontigs, int accumulate,
                                             uint64_t part_stride_bytes) {
    PG_API_BEGIN
    if (!r || !d_src) return fail(PG_E_INVALID, "pg_result_merge_columns: NULL argument");
    if (per == 0 || nparts == 0) return fail(PG_E_INVALID, "pg_result_merge_columns: empty partition");
    uint32_t t0, nt;
    if (int e = contig_tiles(r, first_contig, ncontigs, &t0, &nt)) return e;
    const uint64_t own = (uint64_t)nt * (PROBE_TILE / 8) * per;
    if (part_stride_bytes == 0) part_stride_bytes = own;
    if (part_stride_bytes < own || part_stride_bytes % 8) return fail(PG_E_INVALID, "pg_result_merge_columns_range: bad block stride");
    if (int e = use_device(r->ctx)) return e;
    if (int e = join_result(r)) return e;
    HIP_TRY(launch_cols_merge(r->ctx->stream, r->N, r->d_ad, r->d_tile_contig, t0, nt, r->d_out1, d_src, part0, nparts,
                              part_stride_bytes / 8, per, accumulate ? 1u : 0u));
    r->rows_valid = true;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_result_merge_columns(pg_result *r, const void *d_src, uint32_t nparts, uint32_t per) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "pg_result_merge_columns: NULL argument");
    return pg_result_merge_columns_range(r, d_src, 0, nparts, per, 0, (uint32_t)r->ad.size(), 0, 0);
    PG_API_END
}

extern "C" int pg_rows_epilogue(pg_result *r) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    if (int e = use_device(r->ctx)) return e;
    if (int e = join_result(r)) return e;
    hipStream_t st = r->ctx->stream;
    // an event set of its own (a rows container never ran a probe): what the readers / writers wait on
    if (int e = next_events(r, false)) return e;
    HIP_TRY(hipEventRecord(r->ev[0], st));
    HIP_TRY(hipEventRecord(r->ev[1], st));
    HIP_TRY(hipEventRecord(r->ev[2], st));
    if (int e = enqueue_epilogue(r, st)) return e;  // explicit call (genome-sharded mode): main stream
    HIP_TRY(hipEventRecord(r->ev[3], st));
    r->ev_ok = true;
    r->ev_epi = true;
    r->ev_hist.back().epi = true;
    return PG_OK;
    PG_API_END
}

// ---------------------------------------------------------------------------
// GPU-compressed BGZF: k_row_deflate turns every 65280 payload bytes into a finished BGZF block in
// a 64 KiB slot; the host copies the slots back in batches and appends the blocks to the file.
// ---------------------------------------------------------------------------
static constexpr uint32_t CRC_TAB_WORDS = DF_CRC_TAB_WORDS;
// [0..1024): CRC-32 slicing-by-four tables T0..T3 (T0 = the byte table); then DF_CRC_LEVELS sets of 4 x 256: set j = the
// register after DF_CHUNK_BYTES * 2^j more (zero) bytes, as a function of each of its four bytes
static const uint32_t *crc_tables_host() {
    static uint32_t tab[CRC_TAB_WORDS];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            tab[i] = c;
        }
        for (uint32_t k = 1; k < 4; ++k)
            for (uint32_t b = 0; b < 256; ++b) tab[256 * k + b] = (tab[256 * (k - 1) + b] >> 8) ^ tab[tab[256 * (k - 1) + b] & 255u];
        uint32_t *T0 = tab + 1024;
        for (uint32_t k = 0; k < 4; ++k)
            for (uint32_t b = 0; b < 256; ++b) {
                uint32_t s = b << (8 * k);
                for (uint32_t z = 0; z < DF_CHUNK_BYTES; ++z) s = tab[s & 255u] ^ (s >> 8);
                T0[256 * k + b] = s;
            }
        for (uint32_t j = 1; j < DF_CRC_LEVELS; ++j) {  // set j = set j-1 applied twice
            const uint32_t *P = tab + 1024 + 1024 * (j - 1);
            uint32_t *T = tab + 1024 + 1024 * j;
            auto apply = [&](uint32_t x) { return P[x & 255u] ^ P[256 + ((x >> 8) & 255u)] ^ P[512 + ((x >> 16) & 255u)] ^ P[768 + (x >> 24)]; };
            for (uint32_t k = 0; k < 4; ++k)
                for (uint32_t b = 0; b < 256; ++b) T[256 * k + b] = apply(apply(b << (8 * k)));
        }
    });
    return tab;
}

#ifndef PG_DF_BATCH
#define PG_DF_BATCH 1024
#endif
static constexpr uint32_t DF_BATCH = PG_DF_BATCH;  // BGZF blocks per k_row_deflate launch (64 KiB slot each)

static pg_ctx::DfSet *df_acquire(pg_ctx *ctx) {
    std::unique_lock<std::mutex> lk(ctx->df_mu);
    for (;;) {
        for (auto &d : ctx->df)
            if (!d.busy) {
                d.busy = true;
                return &d;
            }
        ctx->df_cv.wait(lk);  // more writer threads than staging sets: wait for one to be released
    }
}
static void df_release(pg_ctx *ctx, pg_ctx::DfSet *D) {
    {
        std::lock_guard<std::mutex> lk(ctx->df_mu);
        D->busy = false;
    }
    ctx->df_cv.notify_one();
}
// device + pinned buffers of a staging set (all or nothing: a partial set is given back at once)
static void df_free_buffers(pg_ctx::DfSet &d) {
    for (int i = 0; i < 2; ++i) {
        if (d.d_slots[i]) hipFree(d.d_slots[i]);
        if (d.d_packed[i]) hipFree(d.d_packed[i]);
        if (d.d_sizes[i]) hipFree(d.d_sizes[i]);
        if (d.d_offs[i]) hipFree(d.d_offs[i]);
        if (d.h_slots[i]) hipHostFree(d.h_slots[i]);
        if (d.h_sizes[i]) hipHostFree(d.h_sizes[i]);
        d.d_slots[i] = d.d_packed[i] = d.h_slots[i] = nullptr;
        d.d_sizes[i] = d.d_offs[i] = d.h_sizes[i] = nullptr;
    }
    if (d.d_crc) hipFree(d.d_crc);
    if (d.d_hist) hipFree(d.d_hist);
    if (d.d_code) hipFree(d.d_code);
    d.d_crc = d.d_hist = nullptr;
    d.d_code = nullptr;
    d.ready = false;
}

static int write_bgzf_gpu(pg_result *r, const uint8_t *src, const std::vector<std::pair<uint64_t, uint64_t>> &segs_in,
                          uint64_t total, uint32_t row, const char *gz_path, const char *gzi_path) {
    static const unsigned char EOF_BLOCK[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43,
                                                0x02, 0x00, 0x1b, 0x00, 0x03, 0x00, 0, 0, 0, 0, 0, 0, 0, 0};
    pg_ctx *ctx = r->ctx;
    const uint64_t nblocks = (total + 65279) / 65280;
    std::vector<PaySeg> segs;
    uint64_t l = 0;
    for (auto &sg : segs_in) {
        segs.push_back({l, sg.first});
        l += sg.second;
    }
    segs.push_back({total, 0});
    FILE *f = fopen(gz_path, "wb");
    if (!f) return fail(PG_E_IO, "cannot open %s for writing", gz_path);
    pg_ctx::DfSet *D = df_acquire(ctx);
    hipStream_t cs = nullptr;
    PaySeg *d_segs = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    auto ok = [&](hipError_t x) {
        if (e == hipSuccess) e = x;
        return e == hipSuccess;
    };
    if (!D->ready) {
        ok(hipMalloc(reinterpret_cast<void **>(&D->d_crc), CRC_TAB_WORDS * 4));
        ok(hipMalloc(reinterpret_cast<void **>(&D->d_hist), DF_HIST_WORDS * 4));
        ok(hipMalloc(&D->d_code, DF_CODE_BYTES));
        for (int i = 0; i < 2; ++i) {
            ok(hipMalloc(reinterpret_cast<void **>(&D->d_slots[i]), (size_t)DF_BATCH * 65536));
            ok(hipMalloc(reinterpret_cast<void **>(&D->d_packed[i]), (size_t)DF_BATCH * 65536));
            ok(hipMalloc(reinterpret_cast<void **>(&D->d_sizes[i]), (size_t)DF_BATCH * 4));
            ok(hipMalloc(reinterpret_cast<void **>(&D->d_offs[i]), (size_t)(DF_BATCH + 1) * 4));
            ok(hipHostMalloc(reinterpret_cast<void **>(&D->h_slots[i]), (size_t)DF_BATCH * 65536, 0));   // packed blocks
            ok(hipHostMalloc(reinterpret_cast<void **>(&D->h_sizes[i]), (size_t)(DF_BATCH + 1) * 4, 0));  // their offsets
        }
        if (e == hipSuccess) {
            ok(hipMemcpyAsync(D->d_crc, crc_tables_host(), CRC_TAB_WORDS * 4, hipMemcpyHostToDevice, cs));
            ok(hipStreamSynchronize(cs));
        }
        D->ready = e == hipSuccess;
        if (!D->ready) df_free_buffers(*D);  // never keep half a set: the next call would overwrite (leak) its pointers
    }
    ok(hipMalloc(reinterpret_cast<void **>(&d_segs), segs.size() * sizeof(PaySeg)));
    for (int i = 0; i < 2; ++i) {
        ok(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        ok(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
    }
    if (e == hipSuccess) {
        ok(hipMemcpyAsync(d_segs, segs.data(), segs.size() * sizeof(PaySeg), hipMemcpyHostToDevice, cs));
        ok(hipStreamWaitEvent(cs, r->ev[r->ev_epi ? 3 : 1], 0));
    }
    // ONE Huffman code for the file, from a sample of its blocks (pg_deflate.hip)
    if (e == hipSuccess && nblocks) ok(launch_deflate_code(cs, src, d_segs, (uint32_t)segs.size() - 1, total, row, D->d_hist, D->d_code));
    std::vector<uint64_t> coffs, uoffs;
    uint64_t cpos = 0;
    int rc = PG_OK;
    // per batch: compress into slots, pack the finished blocks back to back, bring home the offsets first
    // (they say how many packed bytes to fetch), then the bytes
    auto issue = [&](uint64_t b0, int slot) {
        const uint32_t nb = (uint32_t)std::min<uint64_t>(DF_BATCH, nblocks - b0);
        hipError_t x = hipMemsetAsync(D->d_slots[slot], 0, (size_t)nb * 65536, cs);
        if (x == hipSuccess)
            x = launch_row_deflate(cs, src, d_segs, (uint32_t)segs.size() - 1, total, b0, nb, row, D->d_crc, D->d_code, D->d_slots[slot],
                                   D->d_sizes[slot], getenv("PG_DEFLATE_FORCE_STORED") ? (uint32_t)atoi(getenv("PG_DEFLATE_FORCE_STORED")) : 0u, D->d_offs[slot], D->d_packed[slot]);
        if (x == hipSuccess)
            x = hipMemcpyAsync(D->h_sizes[slot], D->d_offs[slot], (size_t)(nb + 1) * 4, hipMemcpyDeviceToHost, cs);
        if (x == hipSuccess) x = hipEventRecord(done[slot], cs);
        return x;
    };
    if (e == hipSuccess && nblocks) ok(issue(0, 0));
    int slot = 0;
    for (uint64_t b0 = 0; e == hipSuccess && rc == PG_OK && b0 < nblocks; b0 += DF_BATCH, slot ^= 1) {
        const uint32_t nb = (uint32_t)std::min<uint64_t>(DF_BATCH, nblocks - b0);
        if (!ok(hipEventSynchronize(done[slot]))) break;
        const uint32_t *offs = D->h_sizes[slot];
        const uint32_t bytes = offs[nb];
        if (bytes < 26u * nb || bytes > nb * 65536ull) {
            rc = fail(PG_E_IO, "GPU deflate produced %u bytes for %u blocks", bytes, nb);
            break;
        }
        if (!ok(hipMemcpyAsync(D->h_slots[slot], D->d_packed[slot], bytes, hipMemcpyDeviceToHost, cs))) break;
        if (!ok(hipEventRecord(copied[slot], cs))) break;
        if (b0 + DF_BATCH < nblocks && !ok(issue(b0 + DF_BATCH, slot ^ 1))) break;  // the next batch runs behind the copy
        if (!ok(hipEventSynchronize(copied[slot]))) break;
        for (uint32_t i = 0; i < nb; ++i) {
            coffs.push_back(cpos + offs[i]);
            uoffs.push_back((b0 + i) * 65280ull);
        }
        if (fwrite(D->h_slots[slot], 1, bytes, f) != bytes) {
            rc = fail(PG_E_IO, "short write to BGZF file");
            break;
        }
        cpos += bytes;
    }
    if (e != hipSuccess) rc = fail(PG_E_HIP, "pg_result_write_bgzf (GPU deflate): %s", hipGetErrorString(e));
    if (cs) hipStreamSynchronize(cs);
    df_release(ctx, D);
    const std::string keep = rc ? g_err : std::string();
    if (!rc && fwrite(EOF_BLOCK, 1, sizeof EOF_BLOCK, f) != sizeof EOF_BLOCK) rc = fail(PG_E_IO, "short write of BGZF EOF block");
    if (fclose(f) != 0 && !rc) rc = fail(PG_E_IO, "fclose failed on BGZF file");
    if (!rc && gzi_path) {
        FILE *g = fopen(gzi_path, "wb");
        if (!g) rc = fail(PG_E_IO, "cannot open %s", gzi_path);
        else {
            const uint64_t ng = coffs.empty() ? 0 : coffs.size() - 1;
            bool good = fwrite(&ng, 8, 1, g) == 1;
            for (size_t i = 1; good && i < coffs.size(); ++i) good = fwrite(&coffs[i], 8, 1, g) == 1 && fwrite(&uoffs[i], 8, 1, g) == 1;
            if (fclose(g) != 0) good = false;
            if (!good) rc = fail(PG_E_IO, "short write to .gzi");
        }
    }
    if (!keep.empty()) g_err = keep;
    hipFree(d_segs);
    for (int i = 0; i < 2; ++i) {
        if (done[i]) hipEventDestroy(done[i]);
        if (copied[i]) hipEventDestroy(copied[i]);
    }
    if (cs) hipStreamDestroy(cs);
    return rc;
}

// ---------------------------------------------------------------------------
// device rows -> BGZF file: D2H through two pinned buffers on a private stream while the previous
// buffer is being deflated by the writer's threads.  Safe to call from a worker thread while the
// context's streams keep running other results.
// ---------------------------------------------------------------------------
extern "C" int pg_result_write_bgzf(pg_result *r, int step, const char *gz_path, const char *gzi_path, int level,
                                    int nthreads) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "pg_result_write_bgzf: NULL argument");
    return pg_result_write_bgzf_range(r, step, 0, (uint32_t)r->ad.size(), gz_path, gzi_path, level, nthreads);
    PG_API_END
}

extern "C" int pg_result_write_bgzf_range(pg_result *r, int step, uint32_t first_contig, uint32_t ncontigs,
                                          const char *gz_path, const char *gzi_path, int level, int nthreads) {
    PG_API_BEGIN
    if (!r || !gz_path) return fail(PG_E_INVALID, "pg_result_write_bgzf: NULL argument");
    if ((uint64_t)first_contig + ncontigs > r->ad.size())
        return fail(PG_E_INVALID, "contigs %u..%u out of range", first_contig, first_contig + ncontigs);
    if (step != 1 && step != 100 && (uint32_t)step != r->lowres_step)
        return fail(PG_E_INVALID, "step must be 1 or the result's low-resolution step (%u; 100 is accepted as its alias)", r->lowres_step);
    if (!r->ev_ok) return fail(PG_E_INVALID, "pg_anchor_run has not been called on this result");
    if (step == 100 && (r->flags & PG_ANCHOR_ROWS_ONLY) && !r->ev_epi)
        return fail(PG_E_INVALID, "rows-only result: bitmap.100 needs pg_rows_epilogue first");
    if (int e = use_device(r->ctx)) return e;
    // the payload is the contigs' segments back to back (their device buffers are padded apart)
    const uint8_t *src = step == 1 ? r->d_out1 : r->d_out100;
    const uint32_t nbytes_row = (r->N + 7) / 8;
    std::vector<std::pair<uint64_t, uint64_t>> segs;  // (device offset, length)
    uint64_t total = 0;
    for (size_t i = first_contig; i < (size_t)first_contig + ncontigs; ++i) {
        const uint64_t len = (step == 1 ? (uint64_t)r->ad[i].nkmers : r->nrows100[i]) * nbytes_row;
        if (len) segs.emplace_back(step == 1 ? r->ad[i].out_off : r->ad[i].out100_off, len);
        total += len;
    }
    // level -2: compress on the GPU (k_row_deflate), the host only writes the blocks
    if (level == -2 && nbytes_row < 256) return write_bgzf_gpu(r, src, segs, total, nbytes_row, gz_path, gzi_path);
    if (nthreads < 1) nthreads = 1;
    pg_bgzf *w = nullptr;
    if (level >= 0) level |= nbytes_row == 1 ? PG_BGZF_RLE : (nbytes_row < 256 ? PG_BGZF_ROWS(nbytes_row) : 0);
    if (int e = pg_bgzf_open(gz_path, level, nthreads, &w)) return e;
    const size_t chunk = (size_t)512 * 65280;  // 32 MiB: 512 BGZF blocks, shared out one by one among the threads
    hipStream_t cs = nullptr;
    uint8_t *pin[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = PG_OK;
    hipError_t e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipHostMalloc(reinterpret_cast<void **>(&pin[i]), std::min<uint64_t>(chunk, std::max<uint64_t>(total, 1)), 0);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipStreamWaitEvent(cs, r->ev[r->ev_epi ? 3 : 1], 0);
    if (e == hipSuccess) {
        size_t seg = 0;
        uint64_t seg_pos = 0;  // cursor of the next byte to fetch
        auto issue = [&](uint64_t off, int b) {  // payload bytes [off, off+n) -> pin[b]
            const uint64_t n = std::min<uint64_t>(chunk, total - off);
            hipError_t x = hipSuccess;
            uint64_t got = 0;
            while (got < n && x == hipSuccess) {
                const uint64_t take = std::min<uint64_t>(n - got, segs[seg].second - seg_pos);
                x = hipMemcpyAsync(pin[b] + got, src + segs[seg].first + seg_pos, take, hipMemcpyDeviceToHost, cs);
                got += take;
                seg_pos += take;
                if (seg_pos == segs[seg].second) {
                    ++seg;
                    seg_pos = 0;
                }
            }
            if (x == hipSuccess) x = hipEventRecord(done[b], cs);
            return x;
        };
        uint64_t off = 0;
        int b = 0;
        if (total) e = issue(0, 0);
        while (e == hipSuccess && off < total) {
            const uint64_t n = std::min<uint64_t>(chunk, total - off);
            e = hipEventSynchronize(done[b]);
            if (e != hipSuccess) break;
            if (off + n < total) {
                e = issue(off + n, b ^ 1);
                if (e != hipSuccess) break;
            }
            if ((rc = pg_bgzf_write(w, pin[b], n))) break;
            off += n;
            b ^= 1;
        }
    }
    if (e != hipSuccess) rc = fail(PG_E_HIP, "pg_result_write_bgzf: %s", hipGetErrorString(e));
    if (cs) hipStreamSynchronize(cs);
    const std::string keep = rc ? g_err : std::string();
    const int rc2 = pg_bgzf_close(w, rc ? nullptr : gzi_path);
    if (rc) g_err = keep;
    for (int i = 0; i < 2; ++i) {
        if (pin[i]) hipHostFree(pin[i]);
        if (done[i]) hipEventDestroy(done[i]);
    }
    if (cs) hipStreamDestroy(cs);
    return rc ? rc : rc2;
    PG_API_END
}

// ---------------------------------------------------------------------------
// window statistics over finished rows resident in HBM
// ---------------------------------------------------------------------------
extern "C" int pg_result_window_stats(pg_result *r, uint32_t idx, int step, uint32_t nwin, const uint64_t *starts,
                                      const uint64_t *ends, uint64_t *hist, uint64_t *colsums) {
    PG_API_BEGIN
    if (!r || (nwin && (!starts || !ends || !hist))) return fail(PG_E_INVALID, "pg_result_window_stats: NULL argument");
    if (idx >= r->ad.size()) return fail(PG_E_INVALID, "contig %u out of range", idx);
    if (step != 1 && step != 100 && (uint32_t)step != r->lowres_step)
        return fail(PG_E_INVALID, "step must be 1 or the result's low-resolution step (%u; 100 is accepted as its alias)", r->lowres_step);
    if (!r->ev_ok) return fail(PG_E_INVALID, "pg_anchor_run has not been called on this result");
    if (nwin == 0) return PG_OK;
    if (int e = use_device(r->ctx)) return e;
    if (int e = join_result(r)) return e;
    hipStream_t st = r->ctx->stream;
    const uint32_t N = r->N;
    const AnchorDesc &a = r->ad[idx];
    const uint8_t *rows = step == 1 ? r->d_out1 + a.out_off : r->d_out100 + a.out100_off;
    const uint64_t nrows = step == 1 ? (uint64_t)a.nkmers : r->nrows100[idx];
    uint64_t longest = 0;
    for (uint32_t i = 0; i < nwin; ++i)
        if (ends[i] > starts[i]) longest = std::max(longest, std::min(ends[i], nrows) - std::min(starts[i], nrows));
    const uint32_t pieces = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, longest / 32768));
    uint64_t *d_se = nullptr;
    unsigned long long *d_out = nullptr;
    const size_t nh = (size_t)nwin * (N + 1), nc = colsums ? (size_t)nwin * N : 0;
    int rc = PG_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_se), (size_t)nwin * 16);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_out), (nh + nc) * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_se, starts, (size_t)nwin * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_se + nwin, ends, (size_t)nwin * 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, (nh + nc) * 8, st);
    if (e == hipSuccess)
        e = launch_window_stats(st, N, rows, nrows, nwin, pieces, d_se, d_se + nwin, d_out, colsums ? d_out + nh : nullptr);
    if (e == hipSuccess) e = hipMemcpyAsync(hist, d_out, nh * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && colsums) e = hipMemcpyAsync(colsums, d_out + nh, nc * 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = fail(PG_E_HIP, "pg_result_window_stats: %s", hipGetErrorString(e));
    if (d_se) hipFree(d_se);
    if (d_out) hipFree(d_out);
    return rc;
    PG_API_END
}

extern "C" int pg_result_contig_info(const pg_result *r, uint32_t idx, uint64_t *nkmers, uint64_t *nrows100,
                                     uint32_t *nbins, uint32_t *binlen) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    if (idx >= r->ad.size()) return fail(PG_E_INVALID, "contig %u out of range", idx);
    if (nkmers) *nkmers = r->ad[idx].nkmers;
    if (nrows100) *nrows100 = r->nrows100[idx];
    if (nbins) *nbins = r->ad[idx].nbins;
    if (binlen) *binlen = r->ad[idx].binlen;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_result_contigs_small(pg_result *r, uint32_t first, uint32_t ncontigs, uint64_t *nkmers, uint64_t *nrows100,
                                       uint32_t *nbins, uint32_t *binlen, uint32_t *bins, uint64_t bins_words) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    if ((uint64_t)first + ncontigs > r->ad.size()) return fail(PG_E_INVALID, "contigs %u..%u out of range", first, first + ncontigs);
    uint64_t rows = 0;
    for (uint32_t i = 0; i < ncontigs; ++i) {
        const AnchorDesc &a = r->ad[first + i];
        if (nkmers) nkmers[i] = a.nkmers;
        if (nrows100) nrows100[i] = r->nrows100[first + i];
        if (nbins) nbins[i] = a.nbins;
        if (binlen) binlen[i] = a.binlen;
        rows += a.nbins;
    }
    if (!bins || ncontigs == 0) return PG_OK;
    const uint64_t N1 = (uint64_t)r->N + 1;
    if (bins_words != rows * N1) return fail(PG_E_INVALID, "pg_result_contigs_small: %llu words for %llu bin rows of %llu", (unsigned long long)bins_words, (unsigned long long)rows, (unsigned long long)N1);
    if (rows == 0) return PG_OK;
    if (int e = use_device(r->ctx)) return e;
    if (int e = join_result(r)) return e;
    hipStream_t st = r->ctx->stream;
    // (the contigs' bin rows follow one another in the result's buffer)
    HIP_TRY(hipMemcpyAsync(bins, r->d_bins + r->ad[first].bin_off * N1, rows * N1 * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PG_OK;
    PG_API_END
}

extern "C" int pg_result_download(pg_result *r, uint32_t idx, uint8_t *bitmap1, uint8_t *bitmap100, uint32_t *bins) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    if (idx >= r->ad.size()) return fail(PG_E_INVALID, "contig %u out of range", idx);
    if (int e = use_device(r->ctx)) return e;
    if (int e = join_result(r)) return e;
    hipStream_t st = r->ctx->stream;
    const AnchorDesc &a = r->ad[idx];
    const uint32_t N = r->N, nbytes = (N + 7) / 8;
    if (bitmap1 && a.nkmers)
        HIP_TRY(hipMemcpyAsync(bitmap1, r->d_out1 + a.out_off, (uint64_t)a.nkmers * nbytes, hipMemcpyDeviceToHost, st));
    if (bitmap100 && r->nrows100[idx])
        HIP_TRY(hipMemcpyAsync(bitmap100, r->d_out100 + a.out100_off, r->nrows100[idx] * nbytes, hipMemcpyDeviceToHost, st));
    if (bins && a.nbins)
        HIP_TRY(hipMemcpyAsync(bins, r->d_bins + a.bin_off * (N + 1), (uint64_t)a.nbins * (N + 1) * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PG_OK;
    PG_API_END
}

extern "C" int pg_result_colsums(pg_result *r, uint64_t *colsums) {
    PG_API_BEGIN
    if (!r || !colsums) return fail(PG_E_INVALID, "pg_result_colsums: NULL argument");
    if (!(r->flags & PG_ANCHOR_COLSUMS)) return fail(PG_E_INVALID, "result was created without PG_ANCHOR_COLSUMS");
    if (int e = use_device(r->ctx)) return e;
    if (int e = join_result(r)) return e;
    hipStream_t st = r->ctx->stream;
    const size_t N = r->N, nc = r->ad.size();
    std::vector<uint64_t> all(std::max<size_t>(1, nc) * N, 0);  // the device keeps them per contig
    if (nc) {
        HIP_TRY(hipMemcpyAsync(all.data(), r->d_colsums, nc * N * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    for (size_t g = 0; g < N; ++g) colsums[g] = 0;
    for (size_t c = 0; c < nc; ++c)
        for (size_t g = 0; g < N; ++g) colsums[g] += all[c * N + g];
    return PG_OK;
    PG_API_END
}

extern "C" int pg_result_contig_colsums(pg_result *r, uint32_t idx, uint32_t ncontigs, uint64_t *colsums) {
    PG_API_BEGIN
    if (!r || !colsums) return fail(PG_E_INVALID, "pg_result_contig_colsums: NULL argument");
    if (!(r->flags & PG_ANCHOR_COLSUMS)) return fail(PG_E_INVALID, "result was created without PG_ANCHOR_COLSUMS");
    if ((uint64_t)idx + ncontigs > r->ad.size()) return fail(PG_E_INVALID, "contigs %u..%u out of range", idx, idx + ncontigs);
    if (ncontigs == 0) return PG_OK;
    if (int e = use_device(r->ctx)) return e;
    if (int e = join_result(r)) return e;
    hipStream_t st = r->ctx->stream;
    const size_t N = r->N;
    HIP_TRY(hipMemcpyAsync(colsums, r->d_colsums + (size_t)idx * N, (size_t)ncontigs * N * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return PG_OK;
    PG_API_END
}

extern "C" int pg_result_device_ptrs(pg_result *r, void **d1, uint64_t *b1, void **d100, uint64_t *b100) {
    PG_API_BEGIN
    if (!r) return fail(PG_E_INVALID, "result is NULL");
    if (d1) *d1 = r->d_out1;
    if (b1) *b1 = r->out1_bytes;
    if (d100) *d100 = r->d_out100;
    if (b100) *b100 = r->out100_bytes;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_anchor_contig(pg_table *t, const char *ascii, uint64_t len, uint8_t *bitmap1, uint8_t *bitmap100,
                                uint32_t *bins, uint64_t *colsums, uint64_t *nkmers) {
    PG_API_BEGIN
    if (!t || (len && !ascii)) return fail(PG_E_INVALID, "pg_anchor_contig: NULL argument");
    if (nkmers) *nkmers = len >= (uint64_t)t->k ? len - t->k + 1 : 0;
    if (colsums) memset(colsums, 0, (size_t)t->ngenomes * 8);
    if (len < (uint64_t)t->k) return PG_OK;
    pg_seqset *sq = nullptr;
    pg_result *res = nullptr;
    int rc = pg_seqset_create(t->ctx, 1, &len, &sq);
    if (!rc) rc = pg_seqset_load_host(sq, 0, ascii, len);
    if (!rc) rc = pg_result_create(t, sq, colsums ? PG_ANCHOR_COLSUMS : 0, &res);
    if (!rc) rc = pg_anchor_run(res);
    if (!rc) rc = pg_result_download(res, 0, bitmap1, bitmap100, bins);
    if (!rc && colsums) rc = pg_result_colsums(res, colsums);
    pg_result_destroy(res);
    pg_seqset_destroy(sq);
    return rc;
    PG_API_END
}

extern "C" int pg_counters_for_read(pg_table *t, int db_idx, const char *ascii, uint64_t len, uint32_t *out) {
    PG_API_BEGIN
    if (!t || (len && !ascii)) return fail(PG_E_INVALID, "pg_counters_for_read: NULL argument");
    if (db_idx < 0 || db_idx >= t->ndbs) return fail(PG_E_INVALID, "db index %d out of range (0..%d)", db_idx, t->ndbs - 1);
    if (len < (uint64_t)t->k) return PG_OK;
    if (!out) return fail(PG_E_INVALID, "pg_counters_for_read: out is NULL");
    const uint64_t nk = len - t->k + 1;
    pg_seqset *sq = nullptr;
    int rc = pg_seqset_create(t->ctx, 1, &len, &sq);
    if (!rc) rc = pg_seqset_load_host(sq, 0, ascii, len);
    uint32_t *d_out = nullptr;
    if (!rc && hipMalloc(reinterpret_cast<void **>(&d_out), nk * 4) != hipSuccess) rc = fail(PG_E_HIP, "hipMalloc failed");
    if (!rc) {
        hipStream_t st = t->ctx->stream;
        const int si = 0, w = db_idx;
        if (launch_counters(st, t->subs[si].d, w, t->k, sq->d_seqw, sq->d_nmw, sq->d_has_n, nk, d_out) != hipSuccess ||
            hipMemcpyAsync(out, d_out, nk * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
            rc = fail(PG_E_HIP, "counters kernel failed: %s", hipGetErrorString(hipGetLastError()));
    }
    if (d_out) hipFree(d_out);
    pg_seqset_destroy(sq);
    return rc;
    PG_API_END
}
