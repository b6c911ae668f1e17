/*
 * anchor_oracle.c — CPU restatement (plain C) of the reference anchor algorithm.
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker / the timed CPU baseline ("kind": "port").  The
 * product (panagram_amd, libpanagram_hip.so) never links or calls this file.
 *
 * Restates, does not copy:
 *   - KMC's random-access lookup as the reference uses it, CKMCFile::GetCountersForRead
 *     (third-party, un-vendored submodule skovaka/KMC; call sites cpp/anchor.cpp:148,
 *     panagram/index.py:935): KMC1 layout = prefix LUT + records sorted by k-mer, each
 *     (suffix bytes, counter); per position: rolling canonical k-mer -> LUT range ->
 *     binary search over the suffix records; counter returned iff
 *     min_count <= c <= max_count; any window with a byte outside ACGTacgt -> 0.
 *     Layout: SURVEY.md Appendix A (validated by the reference binary reading files in it).
 *   - KMCdb::write_bits, cpp/anchor.cpp:112-195: per <=binlen chunk with k-1 overlap,
 *     per DB: lookup, scatter the low n bytes of each u32 at row stride nbytes, popcount;
 *     rows with contig-relative index % 100 == 0 -> bitmap.100; per-chunk histogram.
 *
 * Parity status: PINNED — tests/test_oracle_c.py checks this file against the golden
 * vectors produced by the reference's own run_anchor binary (tests/golden, .npz files).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int k, lut_p, counter_size, suf_bytes, rec_bytes;
    uint32_t min_count, max_count;
    uint64_t total;
    uint64_t *lut;      /* 4^lut_p + 1 entries (last = total) */
    uint8_t *rec;       /* total * rec_bytes */
} odb_t;

static const int8_t CODE[256] = {
    ['A'] = 1, ['C'] = 2, ['G'] = 3, ['T'] = 4, ['a'] = 1, ['c'] = 2, ['g'] = 3, ['t'] = 4};
/* CODE[c]-1 = symbol 0..3, -1 = not a nucleotide */

/* Build the in-memory KMC1 image from sorted unique keys + counters. */
odb_t *odb_from_arrays(const uint64_t *keys, const uint32_t *counters, uint64_t n, int k, int lut_p,
                       uint32_t min_count, uint32_t max_count) {
    if (k < 1 || k > 32 || lut_p < 1 || lut_p > k || (k - lut_p) % 4) return NULL;
    odb_t *db = (odb_t *)calloc(1, sizeof *db);
    db->k = k;
    db->lut_p = lut_p;
    db->counter_size = 4;
    db->suf_bytes = (k - lut_p) / 4;
    db->rec_bytes = db->suf_bytes + 4;
    db->min_count = min_count;
    db->max_count = max_count;
    db->total = n;
    uint64_t nlut = 1ull << (2 * lut_p);
    db->lut = (uint64_t *)calloc(nlut + 1, 8);
    db->rec = (uint8_t *)malloc((size_t)(n ? n : 1) * db->rec_bytes);
    int sshift = 2 * (k - lut_p);
    uint64_t smask = sshift == 64 ? ~0ull : ((1ull << sshift) - 1);
    uint64_t p = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t pre = sshift >= 64 ? 0 : keys[i] >> sshift;
        while (p <= pre) db->lut[p++] = i;
        uint64_t sfx = keys[i] & smask;
        uint8_t *r = db->rec + i * db->rec_bytes;
        for (int b = 0; b < db->suf_bytes; ++b) r[b] = (uint8_t)(sfx >> (8 * (db->suf_bytes - 1 - b)));
        memcpy(r + db->suf_bytes, &counters[i], 4);
    }
    while (p <= nlut) db->lut[p++] = n;
    return db;
}

/* Parse the images of X.kmc_pre / X.kmc_suf (copies what it needs). */
odb_t *odb_from_kmc1(const uint8_t *pre, uint64_t pre_len, const uint8_t *suf, uint64_t suf_len) {
    if (pre_len < 84 || memcmp(pre, "KMCP", 4) || memcmp(pre + pre_len - 4, "KMCP", 4)) return NULL;
    if (suf_len < 8 || memcmp(suf, "KMCS", 4) || memcmp(suf + suf_len - 4, "KMCS", 4)) return NULL;
    uint32_t hoff;
    memcpy(&hoff, pre + pre_len - 8, 4);
    const uint8_t *h = pre + pre_len - 8 - hoff;
    uint32_t k, mode, csz, lp, minc, maxc, ver;
    uint64_t total;
    memcpy(&k, h, 4); memcpy(&mode, h + 4, 4); memcpy(&csz, h + 8, 4); memcpy(&lp, h + 12, 4);
    memcpy(&minc, h + 16, 4); memcpy(&maxc, h + 20, 4); memcpy(&total, h + 24, 8); memcpy(&ver, h + 60, 4);
    if (ver != 0 || mode != 0 || k < 1 || k > 32 || lp < 1 || lp > k || (k - lp) % 4 || csz < 1 || csz > 4) return NULL;
    odb_t *db = (odb_t *)calloc(1, sizeof *db);
    db->k = (int)k; db->lut_p = (int)lp; db->counter_size = (int)csz;
    db->suf_bytes = (int)(k - lp) / 4; db->rec_bytes = db->suf_bytes + (int)csz;
    db->min_count = minc; db->max_count = maxc; db->total = total;
    uint64_t nlut = 1ull << (2 * lp);
    db->lut = (uint64_t *)malloc((nlut + 1) * 8);
    memcpy(db->lut, pre + 4, nlut * 8);
    db->lut[nlut] = total;
    db->rec = (uint8_t *)malloc((size_t)(total ? total : 1) * db->rec_bytes);
    memcpy(db->rec, suf + 4, (size_t)total * db->rec_bytes);
    return db;
}

void odb_free(odb_t *db) {
    if (!db) return;
    free(db->lut);
    free(db->rec);
    free(db);
}

static inline uint32_t odb_lookup(const odb_t *db, uint64_t key) {
    const int sshift = 2 * (db->k - db->lut_p);
    const uint64_t pre = key >> sshift;
    const uint64_t sfx = key & ((1ull << sshift) - 1);
    uint64_t lo = db->lut[pre], hi = db->lut[pre + 1];
    const int sb = db->suf_bytes, rb = db->rec_bytes;
    while (lo < hi) { /* binary search over the suffix records of this prefix */
        uint64_t mid = lo + ((hi - lo) >> 1);
        const uint8_t *r = db->rec + mid * rb;
        uint64_t v = 0;
        for (int b = 0; b < sb; ++b) v = (v << 8) | r[b];
        if (v < sfx) lo = mid + 1;
        else if (v > sfx) hi = mid;
        else {
            uint32_t c = 0;
            for (int b = 0; b < db->counter_size; ++b) c |= (uint32_t)r[sb + b] << (8 * b);
            return (c >= db->min_count && c <= db->max_count) ? c : 0;
        }
    }
    return 0;
}

/* GetCountersForRead: out[i] for i in [0, len-k+1) */
void odb_counters_for_read(const odb_t *db, const uint8_t *seq, uint64_t len, uint32_t *out) {
    const int k = db->k;
    if (len < (uint64_t)k) return;
    const uint64_t kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    uint64_t fwd = 0, rc = 0;
    int run = 0; /* consecutive valid symbols ending here */
    for (uint64_t i = 0; i < len; ++i) {
        int s = CODE[seq[i]] - 1;
        if (s < 0) {
            run = 0; fwd = rc = 0;
        } else {
            fwd = ((fwd << 2) | (uint64_t)s) & kmask;
            rc = (rc >> 2) | ((uint64_t)(3 - s) << (2 * (k - 1)));
            if (run < k) ++run;
        }
        if (i + 1 >= (uint64_t)k) {
            uint64_t key = fwd < rc ? fwd : rc;
            out[i + 1 - k] = (run >= k) ? odb_lookup(db, key) : 0;
        }
    }
}

/* write_bits for one contig.
 *   rows:     nkmers*nbytes      rows100: ceil(nkmers/100)*nbytes
 *   bins:     nbins*(N+1) u64 (row-major), bin_starts: nbins
 * returns the number of bins written (or -1). */
int64_t oracle_write_bits(const odb_t *const *dbs, int ndbs, int ngenomes, const uint8_t *seq, uint64_t len,
                          uint8_t *rows, uint8_t *rows100, uint64_t *bins, uint64_t *bin_starts) {
    const int k = dbs[0]->k;
    const int nbytes = (ngenomes + 7) / 8;
    if (len < (uint64_t)k) return 0;
    const uint64_t nkmers = len - k + 1;
    uint64_t binlen = 200000;
    if (nkmers / binlen < 100) binlen = nkmers / 100;
    if (binlen == 0) return -1; /* the reference divides by zero here */
    const uint64_t nchunks = nkmers / binlen + (nkmers % binlen != 0);
    uint32_t *ints = (uint32_t *)malloc(binlen * 4);
    int *popc = (int *)malloc(binlen * sizeof(int));
    uint64_t chunk_start = 0, i100 = 0;
    for (uint64_t ch = 0; ch < nchunks; ++ch) {
        uint64_t chunk_end = chunk_start + binlen < nkmers ? chunk_start + binlen : nkmers;
        uint64_t cn = chunk_end - chunk_start;
        uint8_t *out = rows + chunk_start * nbytes;
        memset(popc, 0, cn * sizeof(int));
        int offs = 0;
        for (int d = 0; d < ndbs; ++d) {
            int n;
            if (nbytes <= 4) n = nbytes;
            else if (d == ndbs - 1 && nbytes % 4 > 0) n = nbytes % 4;
            else n = 4;
            odb_counters_for_read(dbs[d], seq + chunk_start, cn + k - 1, ints);
            for (uint64_t j = 0; j < cn; ++j) {
                for (int sh = 0; sh < n; ++sh) out[j * nbytes + offs + sh] = (uint8_t)(ints[j] >> (8 * sh));
                popc[j] += __builtin_popcount(ints[j]);
            }
            offs += n;
        }
        for (uint64_t j = (100 - chunk_start % 100) % 100; j < cn; j += 100) {
            memcpy(rows100 + i100 * nbytes, out + j * nbytes, nbytes);
            ++i100;
        }
        uint64_t *b = bins + ch * (uint64_t)(ngenomes + 1);
        memset(b, 0, (size_t)(ngenomes + 1) * 8);
        for (uint64_t j = 0; j < cn; ++j) b[popc[j] <= ngenomes ? popc[j] : ngenomes]++;
        bin_starts[ch] = chunk_start;
        chunk_start += binlen;
    }
    free(ints);
    free(popc);
    return (int64_t)nchunks;
}
