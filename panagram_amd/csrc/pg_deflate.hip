// pg_deflate.hip — BGZF blocks of a bitmap payload compressed ON THE GPU (gfx950).
//
// Replaces, for payloads resident in HBM, the per-block deflate of htslib's bgzf_write
// (cpp/anchor.cpp:167,177) / bgzip.BGZipWriter (index.py:1035-1037): the host only writes the
// finished blocks to the file.  One workgroup = one BGZF block (65280 payload bytes), one thread =
// 255 of them.  Same scheme as the host's row-aware encoder (pg_bgzf.cpp): the only match tried is
// "same byte as one row earlier" — (length, distance = row width) — literals otherwise, one dynamic
// Huffman code per block.  Output is ordinary RFC 1951 / BGZF: header, raw DEFLATE, CRC32, ISIZE.
//
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

__constant__ uint16_t DF_LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t DF_LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
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

// ---- Huffman code lengths (<= maxlen) for freq[0..n): thread-serial, LDS scratch ----------------
// order[] (n), wgt[] (2n), kid0[]/kid1[] (2n), dep[] (2n) live in LDS; returns nothing, fills len[]
__device__ void df_huff_lengths(const uint32_t *freq, int n, int maxlen, uint8_t *len, uint16_t *order, uint32_t *wgt,
                                uint16_t *kid0, uint16_t *kid1, uint8_t *dep) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
        len[i] = 0;
        if (freq[i]) order[m++] = (uint16_t)i;
    }
    if (m == 0) return;
    if (m == 1) {
        len[order[0]] = 1;
        return;
    }
    for (int i = 1; i < m; ++i) {  // insertion sort by (freq, symbol): rows use few distinct bytes
        const uint16_t s = order[i];
        const uint32_t f = freq[s];
        int j = i - 1;
        while (j >= 0 && (freq[order[j]] > f || (freq[order[j]] == f && order[j] > s))) {
            order[j + 1] = order[j];
            --j;
        }
        order[j + 1] = s;
    }
    for (int i = 0; i < m; ++i) wgt[i] = freq[order[i]];
    int leaf = 0, inner = m, total = m;
    while ((m - leaf) + (total - inner) > 1) {
        int pick[2];
        for (int t = 0; t < 2; ++t) {
            if (leaf < m && (inner >= total || wgt[leaf] <= wgt[inner])) pick[t] = leaf++;
            else pick[t] = inner++;
        }
        wgt[total] = wgt[pick[0]] + wgt[pick[1]];
        kid0[total] = (uint16_t)pick[0];
        kid1[total] = (uint16_t)pick[1];
        ++total;
    }
    int cnt[33];
    for (int d = 0; d < 33; ++d) cnt[d] = 0;
    dep[total - 1] = 0;
    for (int i = total - 1; i >= m; --i) {
        const uint8_t d = (uint8_t)min((int)dep[i] + 1, 32);
        dep[kid0[i]] = d;
        dep[kid1[i]] = d;
    }
    for (int i = 0; i < m; ++i) ++cnt[min((int)dep[i], maxlen)];
    uint32_t kraft = 0;
    for (int d = maxlen; d >= 1; --d) kraft += (uint32_t)cnt[d] << (maxlen - d);
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
        for (int c = 0; c < cnt[d]; ++c) len[order[k++]] = (uint8_t)d;
}

// canonical codes, bit-reversed for the LSB-first stream
__device__ void df_huff_codes(const uint8_t *len, int n, uint16_t *code) {
    uint32_t bl[16], next[16];
    for (int b = 0; b < 16; ++b) bl[b] = 0;
    for (int i = 0; i < n; ++i) ++bl[len[i]];
    bl[0] = 0;
    uint32_t c = 0;
    next[0] = 0;
    for (int b = 1; b < 16; ++b) {
        c = (c + bl[b - 1]) << 1;
        next[b] = c;
    }
    for (int i = 0; i < n; ++i) {
        if (!len[i]) {
            code[i] = 0;
            continue;
        }
        const uint32_t v = next[len[i]]++;
        code[i] = (uint16_t)(__brev(v) >> (32 - len[i]));
    }
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

__device__ __forceinline__ int df_len_sym(uint32_t L) {
    int ls = 28;
    while (DF_LEN_BASE[ls] > L) --ls;
    return ls;
}

// role of position i of a run of equal bytes [s, e):  >0 = a match of that length starts here,
// 0 = literal, -1 = covered by a match that started earlier
__device__ __forceinline__ int df_role(uint32_t i, uint32_t s, uint32_t e) {
    const uint32_t R = e - s, q258 = (R / 258u) * 258u, r = R - q258, o = i - s;
    if (o >= q258) return r >= 3 ? (o == q258 ? (int)r : -1) : 0;
    return (o % 258u == 0) ? 258 : -1;
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
    __shared__ uint32_t hdr_bits, blk_crc;
    // thread-0 scratch of the Huffman builder
    __shared__ uint16_t h_order[288], h_kid0[576], h_kid1[576];
    __shared__ uint32_t h_wgt[576];
    __shared__ uint8_t h_dep[576], cl_sym[320], cl_extra[320];

    const int tid = threadIdx.x;
    const uint64_t blk = first_block + blockIdx.x;
    const uint64_t L0 = blk * DF_BLOCK;
    const uint32_t n = (uint32_t)min((uint64_t)DF_BLOCK, total - L0);  // bytes of this block
    uint8_t *slot = slots + (uint64_t)blockIdx.x * 65536;
    for (int i = tid; i < 288; i += DF_THREADS) hist[i] = 0;
    crc_t[tid] = crc_tabs[tid];
    __syncthreads();

    const uint32_t c0 = min(n, (uint32_t)tid * DF_CHUNK), c1 = min(n, c0 + DF_CHUNK);
    // ---- pass A: equality bits and chunk CRC ----
    uint32_t eq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int fne = 0x7fffffff, lne = -1;
    uint32_t crc = tid == 0 ? 0xFFFFFFFFu : 0u;
    if (c0 < c1) {
        PayCur cur, prv;
        cur_seek(cur, base, segs, nseg, L0 + c0);
        const bool prv_ok = c1 > row;
        if (prv_ok) cur_seek(prv, base, segs, nseg, L0 + (c0 > row ? c0 - row : 0));
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            for (int b = 0; b < 32; ++b) {
                const uint32_t i = c0 + 32 * w + b;
                if (i >= c1) break;
                const uint32_t v = cur_next(cur, base, segs, nseg);
                crc = crc_t[(crc ^ v) & 255u] ^ (crc >> 8);
                bool same = false;
                if (i >= row) same = cur_next(prv, base, segs, nseg) == v;
                if (same) eq[w] |= 1u << b;
                else {
                    if (fne == 0x7fffffff) fne = (int)i;
                    lne = (int)i;
                }
            }
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
    auto is_eq = [&](uint32_t i) { return (eq[(i - c0) >> 5] >> ((i - c0) & 31)) & 1u; };
    auto run_end = [&](uint32_t i) {  // first non-equal position >= i
        for (uint32_t j = i; j < c1; ++j)
            if (!is_eq(j)) return j;
        return (uint32_t)nextNE;
    };

    // ---- pass B: symbol histogram ----
    {
        PayCur cur;
        if (c0 < c1) cur_seek(cur, base, segs, nseg, L0 + c0);
        uint32_t s = 0, e = 0;
        bool in_run = false;
        for (uint32_t i = c0; i < c1; ++i) {
            const uint32_t v = cur_next(cur, base, segs, nseg);
            if (!is_eq(i)) {
                in_run = false;
                atomicAdd(&hist[v], 1u);
                continue;
            }
            if (!in_run) {
                s = (i == c0) ? (uint32_t)(prevNE + 1) : i;
                e = run_end(i);
                in_run = true;
            }
            const int role = df_role(i, s, e);
            if (role > 0) atomicAdd(&hist[257 + df_len_sym((uint32_t)role)], 1u);
            else if (role == 0) atomicAdd(&hist[v], 1u);
        }
    }
    __syncthreads();

    // ---- thread 0: Huffman code, block header, CRC of the block ----
    if (tid == 0) {
        hist[256] = 1;
        bool any_match = false;
        for (int i = 257; i < 286; ++i) any_match |= hist[i] != 0;
        df_huff_lengths(hist, 286, 15, llen, h_order, h_wgt, h_kid0, h_kid1, h_dep);
        df_huff_codes(llen, 286, lcode);
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
        uint32_t cfl[19];
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
        uint8_t cll[19];
        uint16_t clc[19];
        df_huff_lengths(cf, 19, 7, cll, h_order, h_wgt, h_kid0, h_kid1, h_dep);
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
        // CRC32 of the block out of the chunk CRCs: state after |chunk| more bytes = shift(state) ^ crc(chunk)
        uint32_t c = crcp[0];
        for (int t = 1; t < DF_THREADS; ++t) {
            const uint32_t a0 = min(n, (uint32_t)t * DF_CHUNK), a1 = min(n, a0 + DF_CHUNK);
            if (a0 >= a1) break;
            if (a1 - a0 == DF_CHUNK) {
                c = crc_tabs[256 + (c & 255u)] ^ crc_tabs[512 + ((c >> 8) & 255u)] ^ crc_tabs[768 + ((c >> 16) & 255u)] ^
                    crc_tabs[1024 + (c >> 24)];
            } else {
                for (uint32_t z = a0; z < a1; ++z) c = crc_t[c & 255u] ^ (c >> 8);
            }
            c ^= crcp[t];
        }
        blk_crc = c ^ 0xFFFFFFFFu;
    }
    __syncthreads();
    const uint32_t dsym = llen[286];
    const uint32_t dbits = 1u + DF_DIST_EXTRA[dsym];
    const uint32_t dval = (row - DF_DIST_BASE[dsym]) << 1;  // code 0 in bit 0, extra bits above

    // ---- pass C: bit counts, offsets, emission.  One walk = a lambda over the tokens ----
    auto walk = [&](auto &&emit) {
        PayCur cur;
        if (c0 < c1) cur_seek(cur, base, segs, nseg, L0 + c0);
        uint32_t s = 0, e = 0;
        bool in_run = false;
        for (uint32_t i = c0; i < c1; ++i) {
            const uint32_t v = cur_next(cur, base, segs, nseg);
            int role = 0;
            if (is_eq(i)) {
                if (!in_run) {
                    s = (i == c0) ? (uint32_t)(prevNE + 1) : i;
                    e = run_end(i);
                    in_run = true;
                }
                role = df_role(i, s, e);
            } else {
                in_run = false;
            }
            if (role == 0) emit(lcode[v], llen[v]);
            else if (role > 0) {
                const int ls = df_len_sym((uint32_t)role);
                emit(lcode[257 + ls], llen[257 + ls]);
                if (DF_LEN_EXTRA[ls]) emit((uint32_t)role - DF_LEN_BASE[ls], DF_LEN_EXTRA[ls]);
                emit(dval, dbits);
            }
        }
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
        PayCur cur;
        if (c0 < c1) cur_seek(cur, base, segs, nseg, L0 + c0);
        for (uint32_t i = c0; i < c1; ++i) slot[23 + i] = (uint8_t)cur_next(cur, base, segs, nseg);
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
