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

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

extern "C" const char *pg_last_error(void);
int pg_set_error(int code, const char *msg);  // pg_api.hip: one thread-local error slot for the library

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

// compress one block; returns total BGZF block length or 0 on error
size_t deflate_block(Deflater &d, Deflater &stored, const unsigned char *src, size_t n, unsigned char *dst) {
    const size_t cap = MAX_CBLOCK - 18 - 8;
    size_t clen = d.run(src, n, dst + 18, cap);
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
    // current job
    const unsigned char *data = nullptr;
    size_t nbytes = 0, nblk = 0;
    unsigned char *cbuf = nullptr;
    size_t *clen = nullptr;
    std::atomic<size_t> next{0};
    size_t running = 0;

    void work(Deflater &d, Deflater &stored) {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= nblk) break;
            const size_t off = i * BLOCK, n = std::min(BLOCK, nbytes - off);
            clen[i] = deflate_block(d, stored, data + off, n, cbuf + i * MAX_CBLOCK);
        }
    }
    void loop() {
        Deflater d(level), stored(0);
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_job.wait(lk, [&] { return quit || gen != seen; });
            if (quit) return;
            seen = gen;
            lk.unlock();
            work(d, stored);
            lk.lock();
            if (--running == 0) cv_done.notify_all();
        }
    }
    Pool(int nthreads, int lvl) : level(lvl) {
        for (int t = 0; t < nthreads; ++t) th.emplace_back([this] { loop(); });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_job.notify_all();
        for (auto &t : th) t.join();
    }
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
    Pool *pool;
};

static int bfail(int code, const std::string &m) { return pg_set_error(code, m.c_str()); }

// compress + write the blocks of data[0..nbytes) (only the very last block of a file may be short)
static int flush_blocks(pg_bgzf *w, const unsigned char *data, size_t nbytes) {
    const size_t nblk = (nbytes + BLOCK - 1) / BLOCK;
    if (nblk == 0) return PG_OK;
    if (w->cbuf.size() < nblk * MAX_CBLOCK) w->cbuf.resize(nblk * MAX_CBLOCK);
    w->clen.assign(nblk, 0);
    if (w->pool && nblk > 1) w->pool->run(data, nbytes, nblk, w->cbuf.data(), w->clen.data());
    else {
        Deflater d(w->level), stored(0);
        for (size_t i = 0; i < nblk; ++i)
            w->clen[i] = deflate_block(d, stored, data + i * BLOCK, std::min(BLOCK, nbytes - i * BLOCK),
                                       w->cbuf.data() + i * MAX_CBLOCK);
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
    if (!path || !out) return bfail(PG_E_INVALID, "pg_bgzf_open: NULL argument");
    FILE *f = fopen(path, "wb");
    if (!f) return bfail(PG_E_IO, std::string("cannot open ") + path + " for writing");
    pg_bgzf *w = new pg_bgzf();
    w->f = f;
    const int lv = level < 0 ? -1 : (level & 0xff);
    w->level = ((lv < 0 || lv > 9) ? 6 : lv) | (level > 0 ? (level & PG_BGZF_RLE) : 0);
    w->nthreads = nthreads < 1 ? 1 : nthreads;
    w->cpos = w->upos = 0;
    w->batch_blocks = 256;  // 16 MiB of input per parallel batch whatever the thread count
    w->failed = false;
    w->pool = w->nthreads > 1 ? new Pool(w->nthreads, w->level) : nullptr;
    *out = w;
    return PG_OK;
}

extern "C" int pg_bgzf_write(pg_bgzf *w, const void *data_, size_t len) {
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
}

extern "C" int pg_bgzf_close(pg_bgzf *w, const char *gzi_path) {
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
}
