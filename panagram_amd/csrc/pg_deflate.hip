// pg_deflate.hip — BGZF blocks of a bitmap payload compressed ON THE GPU (gfx950).
//
// Replaces, for payloads resident in HBM, the per-block deflate of htslib's bgzf_write
// (cpp/anchor.cpp:167,177) / bgzip.BGZipWriter (index.py:1035-1037): the host only writes the
// finished blocks to the file.  One workgroup = one BGZF block (65280 payload bytes), one thread =
// 255 of them.  Same scheme as the host's row-aware encoder (pg_bgzf.cpp): the only match tried is
// "same byte as one row earlier" — (length, distance = row width) — literals otherwise, one dynamic
// Huffman code per block.  Output is ordinary RFC 1951 / BGZF: header, raw DEFLATE, CRC32, ISIZE.
//
//   stage   the block's 65280 bytes -> LDS (78 KB per workgroup, two per CU), every walk reads LDS
//   pass A  bytes -> equality bits (vs one row earlier), CRC32 of the thread's chunk
//   tokens  a maximal run of equal bytes [s, e) becomes matches of 258, then one of r = (e-s) % 258
//           if r >= 3, else r literals: every position knows its role from (s, e) alone, so threads
//           tokenize their chunks independently once run boundaries crossing chunks are known
//   pass B  symbol histogram (LDS)            thread 0: Huffman lengths / codes / block header
//   pass C  bits per thread -> offsets -> codes OR-ed into the (zeroed) output slot
// Not bit-identical with the host encoders; parity is the decompressed payload and the .gzi geometry.
#include "pg_kernels.h"

namespace pg {

constexpr int DF_THREADS = 256;
constexpr uint32_t DF_BLOCK = 65280, DF_CHUNK = 255;

__constant__ uint16_t DF_DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t DF_DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t DF_CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// ---- the payload as the file sees it: contig segments back to back --------------------------
struct PayCur {
    const uint8_t *p;
    uint64_t left;  // bytes left in the current segment
    uint32_t seg;
};
__device__ __forceinline__ void cur_seek(PayCur &c, const uint8_t *base, const PaySeg *segs, uint32_t nseg, uint64_t L) {
    uint32_t lo = 0, hi = nseg;  // segs[nseg] is a sentinel with lstart = total
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (segs[mid].lstart <= L) lo = mid;
        else hi = mid;
    }
    c.seg = lo;
    c.p = base + segs[lo].doff + (L - segs[lo].lstart);
    c.left = segs[lo + 1].lstart - L;
}
__device__ __forceinline__ uint32_t cur_next(PayCur &c, const uint8_t *base, const PaySeg *segs, uint32_t nseg) {
    const uint32_t v = *c.p++;
    if (--c.left == 0 && c.seg + 1 < nseg) {
        ++c.seg;
        c.p = base + segs[c.seg].doff;
        c.left = segs[c.seg + 1].lstart - segs[c.seg].lstart;
    }
    return v;
}

// ---- Huffman code lengths (<= maxlen): thread-serial part, LDS scratch ------------------------
// order[0..m) = the used symbols sorted by (freq, symbol); wgt[] (2m), kid0[]/kid1[] (2m), dep[] (2m),
// cnt[] (33) live in LDS; fills len[] of the used symbols (the caller zeroed the rest)
__device__ void df_huff_from_sorted(const uint32_t *freq, int m, int maxlen, uint8_t *len, const uint16_t *order,
                                    uint32_t *wgt, uint16_t *kid0, uint16_t *kid1, uint8_t *dep, uint32_t *cnt) {
    if (m == 0) return;
    if (m == 1) {
        len[order[0]] = 1;
        return;
    }
    for (int i = 0; i < m; ++i) wgt[i] = freq[order[i]];
    int leaf = 0, inner = m, total = m;
    while ((m - leaf) + (total - inner) > 1) {  // two-queue merge: leaves and inner nodes are both sorted
        int pick[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (leaf < m && (inner >= total || wgt[leaf] <= wgt[inner])) pick[t] = leaf++;
            else pick[t] = inner++;
        }
        wgt[total] = wgt[pick[0]] + wgt[pick[1]];
        kid0[total] = (uint16_t)pick[0];
        kid1[total] = (uint16_t)pick[1];
        ++total;
    }
    for (int d = 0; d < 33; ++d) cnt[d] = 0;
    dep[total - 1] = 0;
    for (int i = total - 1; i >= m; --i) {
        const uint8_t d = (uint8_t)min((int)dep[i] + 1, 32);
        dep[kid0[i]] = d;
        dep[kid1[i]] = d;
    }
    for (int i = 0; i < m; ++i) ++cnt[min((int)dep[i], maxlen)];
    uint32_t kraft = 0;
    for (int d = maxlen; d >= 1; --d) kraft += cnt[d] << (maxlen - d);
    while (kraft > (1u << maxlen)) {
        --cnt[maxlen];
        for (int d = maxlen - 1; d >= 1; --d)
            if (cnt[d]) {
                --cnt[d];
                cnt[d + 1] += 2;
                break;
            }
        --kraft;
    }
    int k = 0;
    for (int d = maxlen; d >= 1; --d)
        for (uint32_t c = 0; c < cnt[d]; ++c) len[order[k++]] = (uint8_t)d;
}

// small alphabets (the 19 code-length symbols): sort serially, then as above
__device__ void df_huff_lengths(const uint32_t *freq, int n, int maxlen, uint8_t *len, uint16_t *order, uint32_t *wgt,
                                uint16_t *kid0, uint16_t *kid1, uint8_t *dep, uint32_t *cnt) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
        len[i] = 0;
        if (freq[i]) order[m++] = (uint16_t)i;
    }
    for (int i = 1; i < m; ++i) {
        const uint16_t s = order[i];
        const uint32_t f = freq[s];
        int j = i - 1;
        while (j >= 0 && (freq[order[j]] > f || (freq[order[j]] == f && order[j] > s))) {
            order[j + 1] = order[j];
            --j;
        }
        order[j + 1] = s;
    }
    df_huff_from_sorted(freq, m, maxlen, len, order, wgt, kid0, kid1, dep, cnt);
}

// canonical code of symbol i, bit-reversed for the LSB-first stream, from the lengths alone:
// code = sum over shorter used symbols j of 2^(len_i - len_j)  +  #{j < i : len_j == len_i}
__device__ __forceinline__ uint32_t df_canon_code(const uint8_t *len, int n, int i) {
    const uint32_t li = len[i];
    if (!li) return 0;
    uint32_t c = 0;
    for (int j = 0; j < n; ++j) {
        const uint32_t lj = len[j];
        if (lj && lj < li) c += 1u << (li - lj);
        else if (lj == li && j < i) ++c;
    }
    return __brev(c) >> (32 - li);
}

// canonical codes, bit-reversed for the LSB-first stream
__device__ void df_huff_codes(const uint8_t *len, int n, uint16_t *code) {
    for (int i = 0; i < n; ++i) code[i] = (uint16_t)df_canon_code(len, n, i);
}

struct DfBits {  // LSB-first bit writer into a byte array in LDS (the block header)
    uint8_t *p;
    uint32_t acc, n, bits;
    __device__ void put(uint32_t v, uint32_t nb) {
        acc |= v << n;
        n += nb;
        bits += nb;
        while (n >= 8) {
            *p++ = (uint8_t)acc;
            acc >>= 8;
            n -= 8;
        }
    }
};

// DEFLATE length code of a match of L = 3..258 bytes, by arithmetic (a table walk in constant memory
// per token is what the walks would otherwise spend their time on): symbol 257 + ls, nx extra bits of
// value xv
__device__ __forceinline__ int df_len_sym(uint32_t L, uint32_t &nx, uint32_t &xv) {
    const uint32_t x = L - 3;
    if (x < 8) {
        nx = xv = 0;
        return (int)x;
    }
    if (L == 258) {
        nx = xv = 0;
        return 28;
    }
    nx = (31u - (uint32_t)__clz((int)x)) - 2u;
    xv = x & ((1u << nx) - 1u);
    return (int)(4u * nx + 4u + ((x >> nx) & 3u));
}

__global__ __launch_bounds__(DF_THREADS) void k_row_deflate(const uint8_t *__restrict__ base, const PaySeg *__restrict__ segs,
                                                            uint32_t nseg, uint64_t total, uint64_t first_block,
                                                            uint32_t row, const uint32_t *__restrict__ crc_tabs,
                                                            uint8_t *__restrict__ slots, uint32_t *__restrict__ sizes,
                                                            uint32_t force_stored) {
    __shared__ uint32_t hist[288];
    __shared__ uint32_t crc_t[256];
    __shared__ uint16_t lcode[288];
    __shared__ uint8_t llen[288];
    __shared__ int lastNE[DF_THREADS], firstNE[DF_THREADS];
    __shared__ uint32_t tbits[DF_THREADS], crcp[DF_THREADS];
    __shared__ uint8_t hdr[768];
    __shared__ uint32_t hdr_bits, blk_crc, crc_acc;
    // thread-0 scratch of the Huffman builder
    __shared__ uint16_t h_order[288], h_kid0[576], h_kid1[576];
    __shared__ uint32_t h_wgt[576], h_cnt[33], h_m, s_cfl[19];
    __shared__ uint8_t s_cll[19];
    __shared__ uint16_t s_clc[19];
    __shared__ uint8_t h_dep[576], cl_sym[320], cl_extra[320];

    struct __attribute__((packed)) U32 { uint32_t v; };
    struct __attribute__((packed)) U64 { uint64_t v; };
    const int tid = threadIdx.x;
    const uint64_t blk = first_block + blockIdx.x;
    const uint64_t L0 = blk * DF_BLOCK;
    const uint32_t n = (uint32_t)min((uint64_t)DF_BLOCK, total - L0);  // bytes of this block
    uint8_t *slot = slots + (uint64_t)blockIdx.x * 65536;
    for (int i = tid; i < 288; i += DF_THREADS) hist[i] = 0;
    if (tid == 0) crc_acc = 0;
    crc_t[tid] = crc_tabs[tid];
    // ---- the block's bytes into LDS once (gfx950: 160 KB per CU, two of these workgroups fit): a
    // block inside one payload segment is copied coalesced, one that straddles segments chunk-wise ----
    __shared__ uint8_t data[DF_BLOCK];
    __shared__ PayCur blk_cur;
    if (tid == 0) cur_seek(blk_cur, base, segs, nseg, L0);
    __syncthreads();
    const uint32_t c0 = min(n, (uint32_t)tid * DF_CHUNK), c1 = min(n, c0 + DF_CHUNK);
    if (blk_cur.left >= n) {  // block-uniform
        const uint8_t *sp = blk_cur.p;
        const uint32_t head = min(n, (uint32_t)((4 - (reinterpret_cast<uintptr_t>(sp) & 3)) & 3));  // bytes up to 4-alignment
        if ((uint32_t)tid < head) data[tid] = sp[tid];
        const uint32_t nw = (n - head) >> 2;
        // (data + head is generally not 4-aligned in LDS: packed stores)
        const uint32_t *sw = reinterpret_cast<const uint32_t *>(sp + head);
        for (uint32_t i = tid; i < nw; i += DF_THREADS) reinterpret_cast<U32 *>(data + head + 4 * i)->v = sw[i];
        for (uint32_t i = head + 4 * nw + tid; i < n; i += DF_THREADS) data[i] = sp[i];
    } else if (c0 < c1) {
        PayCur cur;
        cur_seek(cur, base, segs, nseg, L0 + c0);
        for (uint32_t i = c0; i < c1; ++i) data[i] = (uint8_t)cur_next(cur, base, segs, nseg);
    }
    __syncthreads();

    // ---- pass A: chunk CRC and the chunk's first / last byte that differs from one row earlier ----
    auto is_eq = [&](uint32_t i) { return i >= row && data[i] == data[i - row]; };
    int fne = 0x7fffffff, lne = -1;
    uint32_t crc = tid == 0 ? 0xFFFFFFFFu : 0u;
    for (uint32_t i = c0; i < c1; ++i) {
        const uint32_t v = data[i];
        crc = crc_t[(crc ^ v) & 255u] ^ (crc >> 8);
        if (!(i >= row && data[i - row] == v)) {
            if (fne == 0x7fffffff) fne = (int)i;
            lne = (int)i;
        }
    }
    firstNE[tid] = fne;
    lastNE[tid] = lne;
    crcp[tid] = crc;
    __syncthreads();
    // run boundaries beyond the chunk
    int prevNE = -1, nextNE = (int)n;
    for (int t = 0; t < tid; ++t) prevNE = max(prevNE, lastNE[t]);
    for (int t = tid + 1; t < DF_THREADS; ++t)
        if (firstNE[t] != 0x7fffffff) {
            nextNE = firstNE[t];
            break;
        }
    // The tokens of this thread's chunk, run by run (not position by position: lanes of a wave are in
    // different runs, and a per-position walk pays for the longest forward scan at every step).  A run of
    // bytes equal to one row earlier, [s, e), is cut into matches of 258 from s, then one match of
    // r = (e - s) % 258 if r >= 3, else r literals — the same for every thread that sees part of the run.
    auto tokens = [&](auto &&on_lit, auto &&on_match) {
        uint32_t i = c0;
        while (i < c1) {
            if (!is_eq(i)) {
                on_lit((uint32_t)data[i]);
                ++i;
                continue;
            }
            const uint32_t s = (i == c0) ? (uint32_t)(prevNE + 1) : i;
            uint32_t e = i + 1;
            // the run's end, eight bytes at a time (unaligned LDS words; e > i >= row here)
            bool open_end = true;
            while (e + 8 <= c1) {
                const uint64_t x = reinterpret_cast<const U64 *>(data + e)->v ^ reinterpret_cast<const U64 *>(data + e - row)->v;
                if (x) {
                    e += (uint32_t)(__ffsll((long long)x) - 1) >> 3;
                    open_end = false;
                    break;
                }
                e += 8;
            }
            if (open_end)
                while (e < c1 && is_eq(e)) ++e;
            const uint32_t hi = e;                     // end of the run inside this chunk
            if (e == c1) e = (uint32_t)nextNE;         // ... and its true end
            const uint32_t R = e - s, q258 = (R / 258u) * 258u, r = R - q258;
            uint32_t p = s + ((i - s + 257u) / 258u) * 258u;  // first match start >= i
            for (; p < min(s + q258, hi); p += 258u) on_match(258u);
            const uint32_t tz = s + q258;              // tail zone [tz, e)
            if (r >= 3) {
                if (tz >= i && tz < hi) on_match(r);
            } else {
                for (uint32_t q = max(i, tz); q < hi; ++q) on_lit((uint32_t)data[q]);
            }
            i = hi;
        }
    };
    // ---- pass B: symbol histogram ----
    tokens([&](uint32_t v) { atomicAdd(&hist[v], 1u); },
           [&](uint32_t L) {
               uint32_t nx, xv;
               atomicAdd(&hist[257 + df_len_sym(L, nx, xv)], 1u);
           });
    __syncthreads();

    // ---- Huffman code of the literal/length alphabet: the used symbols are rank-sorted by the whole
    // workgroup, thread 0 builds the tree over the sorted list, every thread derives its symbols' codes ----
    if (tid == 0) {
        hist[256] = 1;
        h_m = 0;
    }
    __syncthreads();
    for (int sy = tid; sy < 286; sy += DF_THREADS) {
        llen[sy] = 0;
        const uint32_t f = hist[sy];
        if (f) {
            uint32_t rank = 0;
            for (int j = 0; j < 286; ++j) {
                const uint32_t fj = hist[j];
                rank += (fj && (fj < f || (fj == f && j < sy))) ? 1u : 0u;
            }
            h_order[rank] = (uint16_t)sy;
            atomicAdd(&h_m, 1u);
        }
    }
    __syncthreads();
    if (tid == 0) df_huff_from_sorted(hist, (int)h_m, 15, llen, h_order, h_wgt, h_kid0, h_kid1, h_dep, h_cnt);
    __syncthreads();
    for (int sy = tid; sy < 286; sy += DF_THREADS) lcode[sy] = (uint16_t)df_canon_code(llen, 286, sy);
    // ---- thread 0: block header ----
    if (tid == 0) {
        bool any_match = false;
        for (int i = 257; i < 286; ++i) any_match |= hist[i] != 0;
        int dsym = 0;
        while (dsym < 29 && DF_DIST_BASE[dsym + 1] <= row) ++dsym;
        int nlit = 286;
        while (nlit > 257 && llen[nlit - 1] == 0) --nlit;
        const int ndist = any_match ? dsym + 1 : 1;
        // code lengths of both alphabets, run-length coded (16/17/18); hist[] is reused for their counts
        uint32_t *cf = hist;  // 19 counters at hist[0..18]: the literal counts are no longer needed
        auto all_at = [&](int a) -> uint32_t { return a < nlit ? llen[a] : ((any_match && a - nlit == dsym) ? 1u : 0u); };
        const int nall = nlit + ndist;
        int ncl_tok = 0;
        // (counters, lengths and codes of the 19 code-length symbols live in LDS: as thread-local arrays indexed
        // by a run-time symbol they sat in scratch memory, and this thread-serial stretch took 0.54 of the kernel's
        // 2.3 ms)
        uint32_t *cfl = s_cfl;
        for (int i = 0; i < 19; ++i) cfl[i] = 0;
        for (int a = 0; a < nall;) {
            int b = a;
            const uint32_t v = all_at(a);
            while (b < nall && all_at(b) == v) ++b;
            int runlen = b - a;
            if (v == 0) {
                while (runlen >= 11) {
                    const int r = min(runlen, 138);
                    cl_sym[ncl_tok] = 18; cl_extra[ncl_tok++] = (uint8_t)(r - 11); ++cfl[18];
                    runlen -= r;
                }
                if (runlen >= 3) {
                    cl_sym[ncl_tok] = 17; cl_extra[ncl_tok++] = (uint8_t)(runlen - 3); ++cfl[17];
                    runlen = 0;
                }
            } else {
                cl_sym[ncl_tok] = (uint8_t)v; cl_extra[ncl_tok++] = 0; ++cfl[v];
                --runlen;
                while (runlen >= 3) {
                    const int r = min(runlen, 6);
                    cl_sym[ncl_tok] = 16; cl_extra[ncl_tok++] = (uint8_t)(r - 3); ++cfl[16];
                    runlen -= r;
                }
            }
            for (; runlen > 0; --runlen) {
                cl_sym[ncl_tok] = (uint8_t)v; cl_extra[ncl_tok++] = 0; ++cfl[v];
            }
            a = b;
        }
        for (int i = 0; i < 19; ++i) cf[i] = cfl[i];
        uint8_t *cll = s_cll;
        uint16_t *clc = s_clc;
        df_huff_lengths(cf, 19, 7, cll, h_order, h_wgt, h_kid0, h_kid1, h_dep, h_cnt);
        df_huff_codes(cll, 19, clc);
        int ncl = 19;
        while (ncl > 4 && cll[DF_CL_ORDER[ncl - 1]] == 0) --ncl;
        DfBits bw{hdr, 0, 0, 0};
        bw.put(1, 1);
        bw.put(2, 2);
        bw.put((uint32_t)(nlit - 257), 5);
        bw.put((uint32_t)(ndist - 1), 5);
        bw.put((uint32_t)(ncl - 4), 4);
        for (int a = 0; a < ncl; ++a) bw.put(cll[DF_CL_ORDER[a]], 3);
        for (int a = 0; a < ncl_tok; ++a) {
            const uint32_t sy = cl_sym[a];
            bw.put(clc[sy], cll[sy]);
            if (sy == 16) bw.put(cl_extra[a], 2);
            else if (sy == 17) bw.put(cl_extra[a], 3);
            else if (sy == 18) bw.put(cl_extra[a], 7);
        }
        if (bw.n) *bw.p = (uint8_t)bw.acc;  // the last partial byte (its high bits are zero)
        hdr_bits = bw.bits;
        // distance code: the one used symbol gets the 1-bit code 0
        lcode[286] = 0;
        llen[286] = (uint8_t)dsym;  // (slot 286 carries the distance symbol for pass C)
    }
    // CRC32 of the block out of the chunk CRCs.  The register is linear in its state: after a further
    // chunk, state = shift(state) ^ crc(chunk), so the block's CRC is the XOR over the full chunks t of
    // shift^(F-1-t)(crc_t), F = number of full chunks; shift^(2^j) is a table set (crc_tabs[256 +
    // 1024 j ..]), so every thread applies at most 8 of them, in parallel; a short tail chunk (last
    // block of a file) is appended bytewise by thread 0.
    {
        const uint32_t F = n / DF_CHUNK;
        if ((uint32_t)tid < F) {
            uint32_t x = crcp[tid];
            const uint32_t e = F - 1 - (uint32_t)tid;
            for (uint32_t j = 0; j < 8; ++j)
                if ((e >> j) & 1u) {
                    const uint32_t *T = crc_tabs + 256 + 1024 * j;
                    x = T[x & 255u] ^ T[256 + ((x >> 8) & 255u)] ^ T[512 + ((x >> 16) & 255u)] ^ T[768 + (x >> 24)];
                }
            atomicXor(&crc_acc, x);
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t c = crc_acc;
        const uint32_t F = n / DF_CHUNK, tail = n - F * DF_CHUNK;
        if (tail) {
            if (F == 0) c = crcp[0];
            else {
                for (uint32_t z = 0; z < tail; ++z) c = crc_t[c & 255u] ^ (c >> 8);
                c ^= crcp[F];
            }
        }
        blk_crc = c ^ 0xFFFFFFFFu;
    }
    __syncthreads();
    const uint32_t dsym = llen[286];
    const uint32_t dbits = 1u + DF_DIST_EXTRA[dsym];
    const uint32_t dval = (row - DF_DIST_BASE[dsym]) << 1;  // code 0 in bit 0, extra bits above

    // ---- pass C: bit counts, offsets, emission.  One walk = a lambda over the tokens ----
    auto walk = [&](auto &&emit) {
        tokens([&](uint32_t v) { emit(lcode[v], llen[v]); },
               [&](uint32_t L) {
                   uint32_t nx, xv;
                   const int ls = df_len_sym(L, nx, xv);
                   emit(lcode[257 + ls], llen[257 + ls]);
                   if (nx) emit(xv, nx);
                   emit(dval, dbits);
               });
        if (c0 < c1 && c1 == n) emit(lcode[256], llen[256]);  // end of block, by the owner of the last byte
    };
    uint32_t mybits = 0;
    walk([&](uint32_t, uint32_t nb) { mybits += nb; });
    tbits[tid] = mybits;
    __syncthreads();
    // the deflate stream starts behind the 18-byte BGZF header: block header (written by thread 0 in
    // front of its own tokens), then the threads' tokens in order
    uint64_t bitpos = tid == 0 ? 144 : 144 + hdr_bits;
    uint32_t allbits = hdr_bits;
    for (int t = 0; t < DF_THREADS; ++t) {
        if (t < tid) bitpos += tbits[t];
        allbits += tbits[t];
    }
    const uint32_t sbytes = (allbits + 7) / 8;
    const bool fits = sbytes <= 65536 - 18 - 8 && !force_stored;  // (force_stored: test hook for the fallback)
    uint32_t *slotw = reinterpret_cast<uint32_t *>(slot);
    if (fits) {
        // every thread ORs its bits into the zeroed slot: words shared with a neighbour atomically
        uint64_t acc = 0;
        uint32_t nacc = 0;
        uint32_t widx = (uint32_t)(bitpos >> 5);
        const uint32_t first_w = widx;
        nacc = (uint32_t)(bitpos & 31);
        auto flush_word = [&](bool last) {
            const uint32_t wv = (uint32_t)acc;
            if (widx == first_w || last) {
                if (wv) atomicOr(&slotw[widx], wv);
            } else {
                slotw[widx] = wv;
            }
            acc >>= 32;
            nacc -= 32;
            ++widx;
        };
        auto put = [&](uint32_t v, uint32_t nb) {
            acc |= (uint64_t)v << nacc;
            nacc += nb;
            if (nacc >= 32) flush_word(false);
        };
        if (tid == 0) {  // the block header goes first, byte by byte
            acc = 0;
            for (uint32_t b = 0; b * 8 < hdr_bits; ++b) put(hdr[b], min(8u, hdr_bits - 8 * b));
        }
        walk(put);
        if (nacc) {
            nacc += 32;  // flush_word subtracts 32
            flush_word(true);
        }
    }
    __syncthreads();
    if (!fits) {  // incompressible: a stored block (BFINAL=1, BTYPE=00, LEN, NLEN, the bytes)
        if (tid == 0) {
            slot[18] = 1;
            slot[19] = (uint8_t)n;
            slot[20] = (uint8_t)(n >> 8);
            slot[21] = (uint8_t)~n;
            slot[22] = (uint8_t)(~n >> 8);
        }
        for (uint32_t i = tid; i < n; i += DF_THREADS) slot[23 + i] = data[i];
    }
    if (tid == 0) {
        const uint32_t body = fits ? sbytes : 5 + n;
        const uint32_t tot = 18 + body + 8;
        const uint8_t H[16] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0x00, 0x42, 0x43, 0x02, 0x00};
        for (int i = 0; i < 16; ++i) slot[i] = H[i];
        slot[16] = (uint8_t)((tot - 1) & 0xff);
        slot[17] = (uint8_t)((tot - 1) >> 8);
        uint8_t *tr = slot + 18 + body;
        for (int i = 0; i < 4; ++i) tr[i] = (uint8_t)(blk_crc >> (8 * i));
        for (int i = 0; i < 4; ++i) tr[4 + i] = (uint8_t)(n >> (8 * i));
        sizes[blockIdx.x] = tot;
    }
}

// finished blocks of a batch packed back to back (what goes to the file): offsets by one workgroup,
// then one workgroup per block copies its bytes (destination at byte alignment)
__global__ __launch_bounds__(1024) void k_bgzf_offsets(const uint32_t *__restrict__ sizes, uint32_t nb, uint32_t *__restrict__ offs) {
    __shared__ uint32_t part[1024];
    const int tid = threadIdx.x;
    const uint32_t per = (nb + 1023) / 1024;
    uint32_t sum = 0;
    for (uint32_t i = tid * per; i < min(nb, (tid + 1) * per); ++i) sum += sizes[i];
    part[tid] = sum;
    __syncthreads();
    uint32_t base = 0;
    for (int t = 0; t < tid; ++t) base += part[t];
    for (uint32_t i = tid * per; i < min(nb, (tid + 1) * per); ++i) {
        offs[i] = base;
        base += sizes[i];
    }
    if (tid == 1023) offs[nb] = base;
}

__global__ __launch_bounds__(256) void k_bgzf_pack(const uint8_t *__restrict__ slots, const uint32_t *__restrict__ sizes,
                                                   const uint32_t *__restrict__ offs, uint8_t *__restrict__ packed) {
    struct __attribute__((packed)) U32 { uint32_t v; };
    const uint32_t sz = sizes[blockIdx.x];
    const uint32_t *src = reinterpret_cast<const uint32_t *>(slots + (uint64_t)blockIdx.x * 65536);
    uint8_t *dst = packed + offs[blockIdx.x];
    const uint32_t nw = sz >> 2;
    for (uint32_t i = threadIdx.x; i < nw; i += 256) reinterpret_cast<U32 *>(dst + 4 * i)->v = src[i];
    if (threadIdx.x < (sz & 3u)) dst[4 * nw + threadIdx.x] = reinterpret_cast<const uint8_t *>(src)[4 * nw + threadIdx.x];
}

hipError_t preload_deflate_kernels() {
    hipFuncAttributes fa;
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_bgzf_offsets));
}

hipError_t launch_row_deflate(hipStream_t st, const uint8_t *base, const PaySeg *segs, uint32_t nseg, uint64_t total,
                              uint64_t first_block, uint32_t nblocks, uint32_t row, const uint32_t *crc_tabs, uint8_t *slots,
                              uint32_t *sizes, uint32_t force_stored, uint32_t *offs, uint8_t *packed) {
    if (nblocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_row_deflate, dim3(nblocks), dim3(DF_THREADS), 0, st, base, segs, nseg, total, first_block, row,
                       crc_tabs, slots, sizes, force_stored);
    hipLaunchKernelGGL(k_bgzf_offsets, dim3(1), dim3(1024), 0, st, sizes, nblocks, offs);
    hipLaunchKernelGGL(k_bgzf_pack, dim3(nblocks), dim3(256), 0, st, slots, sizes, offs, packed);
    return hipGetLastError();
}

}  // namespace pg
