// pg_bgzf.cpp — multi-threaded BGZF + .gzi writer (host, zlib).
//
// Replaces htslib bgzf_open/bgzf_index_build_init/bgzf_write/bgzf_index_dump/
// bgzf_close as used by cpp/anchor.cpp:46-47,53-54,102-106,167,177 and
// bgzip.BGZipWriter + `bgzip -rI` (index.py:1035-1037,1091-1094).
// Format facts (pinned against the reference binary's output):
//   * data blocks hold exactly 65280 uncompressed bytes except the last;
//   * X.gzi = u64 n (= data blocks - 1), then for blocks 1..n the pair
//     (u64 compressed start offset, u64 uncompressed start offset);
//   * the file ends with the 28-byte BGZF EOF block.
// Compressed bytes are NOT part of parity (zlib level/version differ); the
// decompressed payload and the uncompressed offsets are.
#include "../../include/panagram_hip.h"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "pg_guard.h"

extern "C" const char *pg_last_error(void);

namespace {
constexpr size_t BLOCK = 65280;         // htslib BGZF_BLOCK_SIZE
constexpr size_t MAX_CBLOCK = 65536;    // a BGZF block never exceeds 64 KiB
const unsigned char EOF_BLOCK[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43,
                                     0x02, 0x00, 0x1b, 0x00, 0x03, 0x00, 0, 0, 0, 0, 0, 0, 0, 0};

// one deflate state per worker, reset (not re-allocated) between blocks
struct Deflater {
    z_stream zs;
    int level;
    bool ok;
    // level: 0..9, optionally | PG_BGZF_RLE (matches at distance 1 only: for one-byte rows the runs of
    // equal rows ARE byte runs — measured 6.6x faster than the default strategy at the same ratio)
    explicit Deflater(int lvl) : level(lvl & 0xff) {
        memset(&zs, 0, sizeof zs);
        ok = deflateInit2(&zs, level, Z_DEFLATED, -15, 8, (lvl & PG_BGZF_RLE) ? Z_RLE : Z_DEFAULT_STRATEGY) == Z_OK;
    }
    ~Deflater() {
        if (ok) deflateEnd(&zs);
    }
    // raw deflate of src[0..n) into dst; 0 when it does not fit
    size_t run(const unsigned char *src, size_t n, unsigned char *dst, size_t cap) {
        if (!ok || deflateReset(&zs) != Z_OK) return 0;
        zs.next_in = const_cast<unsigned char *>(src);
        zs.avail_in = (uInt)n;
        zs.next_out = dst;
        zs.avail_out = (uInt)cap;
        return deflate(&zs, Z_FINISH) == Z_STREAM_END ? (size_t)zs.total_out : 0;
    }
};

// ---------------------------------------------------------------------------
// Row-aware raw DEFLATE for bitmap payloads whose rows are wider than a byte.  zlib's matcher is
// built for text: on 4- or 8-byte rows it runs at ~60 MB/s/core, and its run-length strategy only
// sees distance 1.  Consecutive bitmap rows are mostly equal (a row changes where a genome's k-mer
// presence flips), so the only match that matters is "same byte as one row earlier": the tokenizer
// below emits literals and (length, distance = row width) matches in one pass, and a dynamic Huffman
// code per BGZF block does the rest.  Output is ordinary DEFLATE (RFC 1951) — any inflater reads it.
// ---------------------------------------------------------------------------
struct BitWriter {
    unsigned char *p, *end;
    uint64_t acc = 0;
    int n = 0;
    bool overflow = false;
    BitWriter(unsigned char *dst, size_t cap) : p(dst), end(dst + cap) {}
    inline void put(uint32_t v, int bits) {  // LSB-first
        acc |= (uint64_t)v << n;
        n += bits;
        while (n >= 8) {
            if (p < end) *p++ = (unsigned char)acc;
            else overflow = true;
            acc >>= 8;
            n -= 8;
        }
    }
    inline void flush() {
        if (n > 0) put(0, 8 - n);
    }
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// code lengths (<= maxlen) of a Huffman code for freq[0..n): heap-free two-queue construction on
// the sorted symbols, then the usual Kraft-sum repair when the tree is deeper than maxlen
static void huff_lengths(const uint32_t *freq, int n, int maxlen, uint8_t *len) {
    struct Node {
        uint64_t w;
        int sym, a, b;
    };
    std::vector<int> used;
    for (int i = 0; i < n; ++i) {
        len[i] = 0;
        if (freq[i]) used.push_back(i);
    }
    if (used.empty()) return;
    if (used.size() == 1) {
        len[used[0]] = 1;
        return;
    }
    std::sort(used.begin(), used.end(), [&](int x, int y) { return freq[x] != freq[y] ? freq[x] < freq[y] : x < y; });
    const int m = (int)used.size();
    std::vector<Node> nodes;
    nodes.reserve(2 * m);
    for (int i = 0; i < m; ++i) nodes.push_back({freq[used[i]], used[i], -1, -1});
    int leaf = 0, inner = m;
    auto take = [&]() {
        if (leaf < m && (inner >= (int)nodes.size() || nodes[leaf].w <= nodes[inner].w)) return leaf++;
        return inner++;
    };
    while ((m - leaf) + ((int)nodes.size() - inner) > 1) {
        const int a = take(), b = take();
        nodes.push_back({nodes[a].w + nodes[b].w, -1, a, b});
    }
    std::vector<int> depth(nodes.size(), 0), cnt(64, 0);
    for (int i = (int)nodes.size() - 1; i >= 0; --i) {
        if (nodes[i].sym < 0) {
            depth[nodes[i].a] = depth[i] + 1;
            depth[nodes[i].b] = depth[i] + 1;
        } else {
            ++cnt[std::min(depth[i], 63)];
        }
    }
    // repair: fold everything deeper than maxlen into maxlen, then restore the Kraft sum
    for (int d = maxlen + 1; d < 64; ++d) {
        cnt[maxlen] += cnt[d];
        cnt[d] = 0;
    }
    uint64_t total = 0;
    for (int d = maxlen; d >= 1; --d) total += (uint64_t)cnt[d] << (maxlen - d);
    while (total > (1ull << maxlen)) {
        --cnt[maxlen];
        for (int d = maxlen - 1; d >= 1; --d)
            if (cnt[d]) {
                --cnt[d];
                cnt[d + 1] += 2;
                break;
            }
        --total;
    }
    // rarest symbols get the longest codes
    int k = 0;
    for (int d = maxlen; d >= 1; --d)
        for (int c = 0; c < cnt[d]; ++c) len[used[k++]] = (uint8_t)d;
}

// canonical codes, bit-reversed for the LSB-first stream
static void huff_codes(const uint8_t *len, int n, uint16_t *code) {
    int bl[16] = {0};
    for (int i = 0; i < n; ++i) ++bl[len[i]];
    bl[0] = 0;
    uint32_t next[16], c = 0;
    for (int b = 1; b < 16; ++b) {
        c = (c + bl[b - 1]) << 1;
        next[b] = c;
    }
    for (int i = 0; i < n; ++i) {
        if (!len[i]) {
            code[i] = 0;
            continue;
        }
        uint32_t v = next[len[i]]++, r = 0;
        for (int b = 0; b < len[i]; ++b) r |= ((v >> b) & 1u) << (len[i] - 1 - b);
        code[i] = (uint16_t)r;
    }
}

struct LenSym {
    uint8_t of[259];
    LenSym() {
        for (int L = 3; L <= 258; ++L) {
            int ls = 28;
            while (LEN_BASE[ls] > L) --ls;
            of[L] = (uint8_t)ls;
        }
    }
};
static const LenSym LEN_SYM;

struct RowDeflater {
    std::vector<uint32_t> tok;  // literal: byte; match: 0x80000000 | length
    // raw DEFLATE of src[0..n) with matches at distance `row` only; 0 when it does not fit
    size_t run(const unsigned char *src, size_t n, unsigned row, unsigned char *dst, size_t cap) {
        tok.clear();
        uint32_t fl[286] = {0};
        size_t i = 0;
        bool any_match = false;
        while (i < n) {
            size_t L = 0;
            if (i >= row) {
                const size_t lim = std::min<size_t>(258, n - i);
                while (L + 8 <= lim) {  // eight bytes at a time
                    uint64_t x, y;
                    memcpy(&x, src + i + L, 8);
                    memcpy(&y, src + i + L - row, 8);
                    if (x != y) {
                        L += (size_t)(__builtin_ctzll(x ^ y) >> 3);
                        goto matched;
                    }
                    L += 8;
                }
                while (L < lim && src[i + L] == src[i + L - row]) ++L;
            matched:;
            }
            if (L >= 3) {
                const int ls = LEN_SYM.of[L];
                ++fl[257 + ls];
                tok.push_back(0x80000000u | (uint32_t)L);
                any_match = true;
                i += L;
            } else {
                ++fl[src[i]];
                tok.push_back(src[i]);
                ++i;
            }
        }
        fl[256] = 1;
        uint8_t ll[286], dl[30] = {0};
        uint16_t lc[286], dc[30] = {0};
        huff_lengths(fl, 286, 15, ll);
        huff_codes(ll, 286, lc);
        int dsym = 0;
        while (dsym < 29 && DIST_BASE[dsym + 1] <= row) ++dsym;
        if (any_match) dl[dsym] = 1;  // a single distance code: one bit, code 0
        int nlit = 286, ndist = any_match ? dsym + 1 : 1;
        while (nlit > 257 && ll[nlit - 1] == 0) --nlit;
        // code lengths of both alphabets, run-length coded with symbols 16 / 17 / 18
        uint8_t all[286 + 30];
        memcpy(all, ll, nlit);
        memcpy(all + nlit, dl, ndist);
        const int nall = nlit + ndist;
        struct CL {
            uint8_t sym, extra;
        };
        std::vector<CL> cl;
        uint32_t cf[19] = {0};
        for (int a = 0; a < nall;) {
            int b = a;
            while (b < nall && all[b] == all[a]) ++b;
            int runlen = b - a;
            if (all[a] == 0) {
                while (runlen >= 11) {
                    const int r = std::min(runlen, 138);
                    cl.push_back({18, (uint8_t)(r - 11)});
                    ++cf[18];
                    runlen -= r;
                }
                if (runlen >= 3) {
                    cl.push_back({17, (uint8_t)(runlen - 3)});
                    ++cf[17];
                    runlen = 0;
                }
            } else {
                cl.push_back({all[a], 0});
                ++cf[all[a]];
                --runlen;
                while (runlen >= 3) {
                    const int r = std::min(runlen, 6);
                    cl.push_back({16, (uint8_t)(r - 3)});
                    ++cf[16];
                    runlen -= r;
                }
            }
            for (; runlen > 0; --runlen) {
                cl.push_back({all[a], 0});
                ++cf[all[a]];
            }
            a = b;
        }
        uint8_t cll[19];
        uint16_t clc[19];
        huff_lengths(cf, 19, 7, cll);
        huff_codes(cll, 19, clc);
        int ncl = 19;
        while (ncl > 4 && cll[CL_ORDER[ncl - 1]] == 0) --ncl;
        BitWriter bw(dst, cap);
        bw.put(1, 1);  // BFINAL
        bw.put(2, 2);  // dynamic Huffman
        bw.put((uint32_t)(nlit - 257), 5);
        bw.put((uint32_t)(ndist - 1), 5);
        bw.put((uint32_t)(ncl - 4), 4);
        for (int a = 0; a < ncl; ++a) bw.put(cll[CL_ORDER[a]], 3);
        for (const CL &c : cl) {
            bw.put(clc[c.sym], cll[c.sym]);
            if (c.sym == 16) bw.put(c.extra, 2);
            else if (c.sym == 17) bw.put(c.extra, 3);
            else if (c.sym == 18) bw.put(c.extra, 7);
        }
        const uint32_t dextra = row - DIST_BASE[dsym];
        for (uint32_t t : tok) {
            if (t & 0x80000000u) {
                const uint32_t L = t & 0xFFFFu;
                const int ls = LEN_SYM.of[L];
                bw.put(lc[257 + ls], ll[257 + ls]);
                if (LEN_EXTRA[ls]) bw.put(L - LEN_BASE[ls], LEN_EXTRA[ls]);
                bw.put(dc[dsym], 1);
                if (DIST_EXTRA[dsym]) bw.put(dextra, DIST_EXTRA[dsym]);
            } else {
                bw.put(lc[t], ll[t]);
            }
        }
        bw.put(lc[256], ll[256]);
        bw.flush();
        return bw.overflow ? 0 : (size_t)(bw.p - dst);
    }
};

// compress one block; returns total BGZF block length or 0 on error
size_t deflate_block(Deflater &d, Deflater &stored, RowDeflater *rd, unsigned row, const unsigned char *src, size_t n,
                     unsigned char *dst) {
    const size_t cap = MAX_CBLOCK - 18 - 8;
    size_t clen = rd ? rd->run(src, n, row, dst + 18, cap) : 0;
    if (!clen) clen = d.run(src, n, dst + 18, cap);
    if (!clen) clen = stored.run(src, n, dst + 18, cap);  // incompressible: level 0 always fits 65280 bytes
    if (!clen) return 0;
    const size_t total = clen + 18 + 8;
    static const unsigned char hdr[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43, 0x02, 0x00};
    memcpy(dst, hdr, 16);
    dst[16] = (unsigned char)((total - 1) & 0xff);
    dst[17] = (unsigned char)((total - 1) >> 8);
    uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), src, (uInt)n);
    uint32_t isz = (uint32_t)n;
    memcpy(dst + 18 + clen, &crc, 4);
    memcpy(dst + 18 + clen + 4, &isz, 4);
    return total;
}

// persistent workers: a "parallel for" over the 65280-byte blocks of one pg_bgzf_write call,
// one block at a time from a shared counter (a block takes ~2 ms at level 6)
struct Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    uint64_t gen = 0;
    bool quit = false;
    int level;
    unsigned row = 0;  // > 0: rows of that many bytes, row-aware deflate
    // current job
    const unsigned char *data = nullptr;
    size_t nbytes = 0, nblk = 0;
    unsigned char *cbuf = nullptr;
    size_t *clen = nullptr;
    std::atomic<size_t> next{0};
    size_t running = 0;

    void work(Deflater &d, Deflater &stored, RowDeflater &rd) {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= nblk) break;
            const size_t off = i * BLOCK, n = std::min(BLOCK, nbytes - off);
            clen[i] = deflate_block(d, stored, row ? &rd : nullptr, row, data + off, n, cbuf + i * MAX_CBLOCK);
        }
    }
    // A worker never lets an exception out of its thread function (that would be std::terminate under the
    // interpreter): a failure — an allocation inside the encoders, say — marks the pool `failed`, the worker keeps
    // taking part in the job hand-shake, and pg_bgzf_write reports PG_E_IO.
    std::atomic<bool> failed{false};
    void loop() noexcept {
        std::unique_ptr<Deflater> d, stored;
        std::unique_ptr<RowDeflater> rd;
        try {
            d.reset(new Deflater(level));
            stored.reset(new Deflater(0));
            rd.reset(new RowDeflater());
        } catch (...) {
            failed.store(true);
        }
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_job.wait(lk, [&] { return quit || gen != seen; });
            if (quit) return;
            seen = gen;
            lk.unlock();
            if (!failed.load()) {
                try {
                    work(*d, *stored, *rd);
                } catch (...) {
                    failed.store(true);
                }
            }
            lk.lock();
            if (--running == 0) cv_done.notify_all();
        }
    }
    Pool(int nthreads, int lvl, unsigned row_) : level(lvl), row(row_) {
        try {
            for (int t = 0; t < nthreads; ++t) th.emplace_back([this] { loop(); });
        } catch (...) {  // (thread creation refused: the ones already running must be joined before the vector dies)
            stop();
            throw;
        }
    }
    void stop() noexcept {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_job.notify_all();
        for (auto &t : th) t.join();
        th.clear();
    }
    ~Pool() { stop(); }
    void run(const unsigned char *d, size_t n, size_t blocks, unsigned char *out, size_t *lens) {
        std::unique_lock<std::mutex> lk(mu);
        data = d;
        nbytes = n;
        nblk = blocks;
        cbuf = out;
        clen = lens;
        next.store(0);
        running = th.size();
        ++gen;
        cv_job.notify_all();
        cv_done.wait(lk, [&] { return running == 0; });
    }
};
}  // namespace

struct pg_bgzf {
    FILE *f;
    int level, nthreads;
    std::vector<unsigned char> pending;            // not yet compressed input (< one batch)
    std::vector<uint64_t> coffs, uoffs;            // start offsets of every data block
    uint64_t cpos, upos;
    size_t batch_blocks;
    std::vector<unsigned char> cbuf;
    std::vector<size_t> clen;
    bool failed;
    unsigned row;  // PG_BGZF_ROWS(width): row-aware deflate
    Pool *pool;
};

static int bfail(int code, const std::string &m) { return pg_set_error(code, m.c_str()); }

// compress + write the blocks of data[0..nbytes) (only the very last block of a file may be short)
static int flush_blocks(pg_bgzf *w, const unsigned char *data, size_t nbytes) {
    const size_t nblk = (nbytes + BLOCK - 1) / BLOCK;
    if (nblk == 0) return PG_OK;
    if (w->cbuf.size() < nblk * MAX_CBLOCK) w->cbuf.resize(nblk * MAX_CBLOCK);
    w->clen.assign(nblk, 0);
    if (w->pool && nblk > 1) {
        w->pool->run(data, nbytes, nblk, w->cbuf.data(), w->clen.data());
        if (w->pool->failed.load()) return bfail(PG_E_IO, "a BGZF worker thread failed (out of memory?)");
    } else {
        Deflater d(w->level), stored(0);
        RowDeflater rd;
        for (size_t i = 0; i < nblk; ++i)
            w->clen[i] = deflate_block(d, stored, w->row ? &rd : nullptr, w->row, data + i * BLOCK,
                                       std::min(BLOCK, nbytes - i * BLOCK), w->cbuf.data() + i * MAX_CBLOCK);
    }
    for (size_t i = 0; i < nblk; ++i) {
        if (w->clen[i] == 0) return bfail(PG_E_IO, "deflate failed");
        w->coffs.push_back(w->cpos);
        w->uoffs.push_back(w->upos);
        if (fwrite(w->cbuf.data() + i * MAX_CBLOCK, 1, w->clen[i], w->f) != w->clen[i])
            return bfail(PG_E_IO, "short write to BGZF file");
        w->cpos += w->clen[i];
        w->upos += std::min(BLOCK, nbytes - i * BLOCK);
    }
    return PG_OK;
}

extern "C" int pg_bgzf_open(const char *path, int level, int nthreads, pg_bgzf **out) {
    PG_API_BEGIN
    if (!path || !out) return bfail(PG_E_INVALID, "pg_bgzf_open: NULL argument");
    FILE *f = fopen(path, "wb");
    if (!f) return bfail(PG_E_IO, std::string("cannot open ") + path + " for writing");
    pg_bgzf *w = nullptr;
    try {
        w = new pg_bgzf();
    } catch (...) {
        fclose(f);
        throw;  // (PG_API_END turns it into an error code)
    }
    w->f = f;
    w->pool = nullptr;
    const int lv = level < 0 ? -1 : (level & 0xff);
    w->level = ((lv < 0 || lv > 9) ? 6 : lv) | (level > 0 ? (level & PG_BGZF_RLE) : 0);
    w->row = level > 0 ? (unsigned)((level >> 16) & 0xff) : 0;
    w->nthreads = nthreads < 1 ? 1 : nthreads;
    w->cpos = w->upos = 0;
    w->batch_blocks = 256;  // 16 MiB of input per parallel batch whatever the thread count
    w->failed = false;
    try {
        if (w->nthreads > 1) w->pool = new Pool(w->nthreads, w->level, w->row);
    } catch (...) {
        fclose(f);
        delete w;
        throw;
    }
    *out = w;
    return PG_OK;
    PG_API_END
}

extern "C" int pg_bgzf_write(pg_bgzf *w, const void *data_, size_t len) {
    PG_API_BEGIN
    if (!w || (len && !data_)) return bfail(PG_E_INVALID, "pg_bgzf_write: NULL argument");
    if (w->failed) return bfail(PG_E_IO, "BGZF writer is in a failed state");
    const unsigned char *data = static_cast<const unsigned char *>(data_);
    const size_t batch = w->batch_blocks * BLOCK;
    // top up the pending buffer to a whole number of blocks first
    if (!w->pending.empty()) {
        size_t need = batch - w->pending.size();
        size_t take = std::min(need, len);
        w->pending.insert(w->pending.end(), data, data + take);
        data += take;
        len -= take;
        if (w->pending.size() == batch) {
            if (int r = flush_blocks(w, w->pending.data(), batch)) {
                w->failed = true;
                return r;
            }
            w->pending.clear();
        }
    }
    while (len >= batch) {  // compress whole blocks straight from the caller's buffer, <= 64 MiB at a time
        const size_t whole = std::min<size_t>(len / BLOCK, 1024) * BLOCK;
        if (int r = flush_blocks(w, data, whole)) {
            w->failed = true;
            return r;
        }
        data += whole;
        len -= whole;
    }
    if (len) w->pending.insert(w->pending.end(), data, data + len);
    return PG_OK;
    PG_API_END
}

extern "C" int pg_bgzf_close(pg_bgzf *w, const char *gzi_path) {
    PG_API_BEGIN
    if (!w) return PG_OK;
    int rc = PG_OK;
    if (!w->failed && !w->pending.empty()) rc = flush_blocks(w, w->pending.data(), w->pending.size());
    if (!rc && !w->failed && fwrite(EOF_BLOCK, 1, sizeof EOF_BLOCK, w->f) != sizeof EOF_BLOCK)
        rc = bfail(PG_E_IO, "short write of BGZF EOF block");
    if (fclose(w->f) != 0 && !rc) rc = bfail(PG_E_IO, "fclose failed on BGZF file");
    if (!rc && !w->failed && gzi_path) {
        FILE *g = fopen(gzi_path, "wb");
        if (!g) rc = bfail(PG_E_IO, std::string("cannot open ") + gzi_path);
        else {
            uint64_t n = w->coffs.empty() ? 0 : w->coffs.size() - 1;
            bool ok = fwrite(&n, 8, 1, g) == 1;
            for (size_t i = 1; ok && i < w->coffs.size(); ++i)
                ok = fwrite(&w->coffs[i], 8, 1, g) == 1 && fwrite(&w->uoffs[i], 8, 1, g) == 1;
            if (fclose(g) != 0) ok = false;
            if (!ok) rc = bfail(PG_E_IO, "short write to .gzi");
        }
    }
    if (w->failed && !rc) rc = PG_E_IO;
    delete w->pool;
    delete w;
    return rc;
    PG_API_END
}

// ---------------------------------------------------------------------------
// bitsum.bins.tsv (cpp/anchor.cpp:57-69, 184-189): plain text, formatted natively — an assembly of 20 000 contigs has
// two million bin rows per anchor genome, and row-by-row formatting in the interpreter took most of the run
// ---------------------------------------------------------------------------
namespace {
inline char *put_u64(char *p, uint64_t v) {
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}
}  // namespace

extern "C" int pg_write_bins_tsv(const char *path, uint32_t ngenomes, uint32_t ncontigs, const uint32_t *nbins,
                                 const uint32_t *binlen, const uint32_t *bins) {
    PG_API_BEGIN
    if (!path || (ncontigs && (!nbins || !binlen))) return bfail(PG_E_INVALID, "pg_write_bins_tsv: NULL argument");
    FILE *f = fopen(path, "wb");
    if (!f) return bfail(PG_E_IO, std::string("cannot open ") + path + " for writing");
    const size_t N1 = (size_t)ngenomes + 1, line_max = 2 * 21 + N1 * 11 + 2;
    std::vector<char> buf(std::max<size_t>(1 << 20, 4 * line_max));
    char *p = buf.data();
    bool good = true;
    auto flush = [&]() {
        good = good && fwrite(buf.data(), 1, (size_t)(p - buf.data()), f) == (size_t)(p - buf.data());
        p = buf.data();
    };
    memcpy(p, "chr\tstart", 9);
    p += 9;
    for (size_t i = 0; i < N1; ++i) {
        if ((size_t)(p - buf.data()) + 24 > buf.size()) flush();
        *p++ = '\t';
        p = put_u64(p, i);
    }
    *p++ = '\n';
    const uint32_t *row = bins;
    for (uint32_t c = 0; c < ncontigs && good; ++c)
        for (uint32_t b = 0; b < nbins[c]; ++b, row += N1) {
            if ((size_t)(p - buf.data()) + line_max > buf.size()) flush();
            p = put_u64(p, c);
            *p++ = '\t';
            p = put_u64(p, (uint64_t)b * binlen[c]);
            for (size_t i = 0; i < N1; ++i) {
                *p++ = '\t';
                p = put_u64(p, row[i]);
            }
            *p++ = '\n';
        }
    flush();
    if (fclose(f) != 0) good = false;
    if (!good) return bfail(PG_E_IO, std::string("short write to ") + path);
    return PG_OK;
    PG_API_END
}
