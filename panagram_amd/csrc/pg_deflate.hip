// pg_deflate.hip — BGZF blocks of a bitmap payload compressed ON THE GPU (gfx950).
//
// Replaces, for payloads resident in HBM, the per-block deflate of htslib's bgzf_write
// (cpp/anchor.cpp:167,177) / bgzip.BGZipWriter (index.py:1035-1037): the host only writes the
// finished blocks to the file.  Same scheme as the host's row-aware encoder (pg_bgzf.cpp): the only match tried is
// "same byte as one row earlier" — (length, distance = row width) — literals otherwise.  Output is ordinary
// RFC 1951 / BGZF: header, raw DEFLATE (one dynamic-Huffman block), CRC32, ISIZE.
//
// Round 6 (second half): ONE Huffman code per FILE instead of one per block.  Per block the old kernel spent 37 % of its
// 1.5 M cycles in thread 0 (Huffman lengths, block header), 17 % on the histogram walk and its rank sort, and the rest in
// three walks of 255 bytes per thread at two waves per SIMD (profiles/r6y_deflate_phases.txt: 50 GB/s at best, the bound of
// every files -> files run).  A bitmap's blocks look alike, so:
//   k_df_sample_hist   up to 512 blocks spread over the file -> symbol counts (the same tokens the encoder emits)
//   k_df_build_code    one workgroup: counts + 1 for EVERY symbol (a block may hold what the sample did not) -> code lengths
//                      (<= 15), canonical codes, the dynamic block header's bits — the same for every block of the file
//   k_row_deflate      one workgroup = one BGZF block (65280 payload bytes), 1024 threads, one thread = 68 bytes (17 dwords:
//                      an odd dword stride keeps the threads' LDS walks off each other's banks), 2 workgroups = 32 waves per CU:
//       stage   the block's bytes -> LDS (every walk reads LDS)
//       pass A  equality with one row earlier, dword-wise; CRC32 of the thread's chunk (slicing by four)
//       scans   last / next unequal byte outside the chunk (wave shuffles + 16 partials)
//       tokens  a maximal run of equal bytes [s, e) becomes matches of 258, then one of r = (e-s) % 258 if r >= 3, else r
//               literals: every position knows its role from (s, e) alone, so threads tokenize their chunks independently
//       pass C  bits per thread -> offsets (scan) -> codes OR-ed into the (zeroed) output slot
// Not bit-identical with the host encoders; parity is the decompressed payload and the .gzi geometry.
#include "pg_kernels.h"

namespace pg {

constexpr int DF_THREADS = 1024, DF_WAVES = DF_THREADS / 64;
constexpr uint32_t DF_BLOCK = 65280, DF_CHUNK = DF_CHUNK_BYTES, DF_NCHUNK = DF_BLOCK / DF_CHUNK;
static_assert(DF_BLOCK % DF_CHUNK == 0 && DF_CHUNK % 4 == 0 && ((DF_CHUNK / 4) & 1) == 1 && DF_NCHUNK <= (uint32_t)DF_THREADS &&
              DF_NCHUNK <= (1u << DF_CRC_LEVELS), "chunks: whole dwords, an odd number of them, one per thread, within the CRC shift tables' reach");
constexpr int DF_CODE_THREADS = 256;

// what k_df_build_code leaves for the blocks of a file (DF_CODE_BYTES of device memory)
struct DfCode {
    uint16_t lcode[288];  // canonical codes, bit-reversed for the LSB-first stream ([286]: the distance code, 0)
    uint8_t llen[288];    // their lengths ([286]: the distance SYMBOL of the row width)
    uint8_t hdr[768];     // the dynamic block header's bits
    uint32_t hdr_bits;
};
static_assert(sizeof(DfCode) <= DF_CODE_BYTES, "DfCode must fit the buffer the host allocates");

// -DPG_DF_PHASE: a measuring build (tools/deflate_phases.py) — thread 0 of every k_row_deflate workgroup stamps the cycle counter at
// the kernel's phase boundaries (the barriers make its timeline the workgroup's) and adds the phases' cycles to pg_df_phase_cycles.
// Slots: 0 stage, 1 pass A, 2 run boundaries (scans), 3 CRC combine, 4 bit counts, 5 offsets (scan), 6 emission, 7 trailer,
// 15 workgroups.
#ifdef PG_DF_PHASE
__device__ unsigned long long pg_df_phase_cycles[256 * 16];
#define DF_PH_DECL uint32_t ph_t = (uint32_t)__builtin_readcyclecounter();
#define DF_PH(i) { if (threadIdx.x == 0) { const uint32_t ph_n = (uint32_t)__builtin_readcyclecounter(); atomicAdd(&pg_df_phase_cycles[(blockIdx.x & 255u) * 16u + (i)], (unsigned long long)(ph_n - ph_t)); ph_t = ph_n; } }
#else
#define DF_PH_DECL
#define DF_PH(i)
#endif

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


// ---- workgroup scans over the threads' values (DF_WAVES waves): wave shuffles, then the waves' partials through LDS.
// Every thread of the workgroup calls them (barriers inside); `part` holds DF_WAVES words. ----
__device__ __forceinline__ uint32_t df_excl_sum(uint32_t v, uint32_t *part, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    uint32_t basev = 0, all = 0;
#pragma unroll
    for (int w = 0; w < DF_WAVES; ++w) {
        const uint32_t p = part[w];
        if (w < wave) basev += p;
        all += p;
    }
    __syncthreads();
    total = all;
    return basev + inc - v;
}
// max over the threads BEFORE this one (-1 if none)
__device__ __forceinline__ int df_excl_max(int v, int *part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc = max(inc, o);
    }
    if (lane == 63) part[wave] = inc;
    int ex = __shfl_up(inc, 1);
    if (lane == 0) ex = -1;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < DF_WAVES; ++w)
        if (w < wave) ex = max(ex, part[w]);
    __syncthreads();
    return ex;
}
// min over the threads BEHIND this one (`none` if there is none)
__device__ __forceinline__ int df_rexcl_min(int v, int none, int *part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_down(inc, d);
        if (lane + d < 64) inc = min(inc, o);
    }
    if (lane == 0) part[wave] = inc;
    int ex = __shfl_down(inc, 1);
    if (lane == 63) ex = none;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < DF_WAVES; ++w)
        if (w > wave) ex = min(ex, part[w]);
    __syncthreads();
    return ex;
}

struct __attribute__((packed)) DfU32 { uint32_t v; };
struct __attribute__((packed)) DfU64 { uint64_t v; };

// ---- the block's bytes into LDS once: a block inside one payload segment is copied coalesced, one that straddles
// segments chunk-wise.  Returns the block's byte count; [c0, c1) is the calling thread's chunk. ----
__device__ __forceinline__ uint32_t df_stage(uint8_t *data, PayCur *blk_cur, const uint8_t *base, const PaySeg *segs, uint32_t nseg,
                                             uint64_t total, uint64_t blk, uint32_t &c0, uint32_t &c1) {
    const int tid = threadIdx.x;
    const uint64_t L0 = blk * DF_BLOCK;
    const uint32_t n = (uint32_t)min((uint64_t)DF_BLOCK, total - L0);
    if (tid == 0) cur_seek(*blk_cur, base, segs, nseg, L0);
    __syncthreads();
    c0 = min(n, (uint32_t)tid * DF_CHUNK);
    c1 = min(n, c0 + DF_CHUNK);
    if (blk_cur->left >= n) {  // block-uniform
        const uint8_t *sp = blk_cur->p;
        const uint32_t head = min(n, (uint32_t)((4 - (reinterpret_cast<uintptr_t>(sp) & 3)) & 3));  // bytes up to 4-alignment
        if ((uint32_t)tid < head) data[tid] = sp[tid];
        const uint32_t nw = (n - head) >> 2;
        // (data + head is generally not 4-aligned in LDS: packed stores)
        const uint32_t *sw = reinterpret_cast<const uint32_t *>(sp + head);
        for (uint32_t i = tid; i < nw; i += DF_THREADS) reinterpret_cast<DfU32 *>(data + head + 4 * i)->v = sw[i];
        for (uint32_t i = head + 4 * nw + tid; i < n; i += DF_THREADS) data[i] = sp[i];
    } else if (c0 < c1) {
        PayCur cur;
        cur_seek(cur, base, segs, nseg, L0 + c0);
        for (uint32_t i = c0; i < c1; ++i) data[i] = (uint8_t)cur_next(cur, base, segs, nseg);
    }
    __syncthreads();
    return n;
}

// the chunk's bytes that differ from one row earlier, one bit each (bit j = byte c0 + j; the bits behind the chunk's end are SET: a
// run of equal bytes never reaches past them) — what pass A leaves for the token walks, which then touch LDS for literals only
struct DfMask {
    uint64_t lo;  // bytes 0..63 of the chunk
    uint32_t hi;  // bytes 64..(DF_CHUNK - 1), and ones above
};
static_assert(DF_CHUNK > 64 && DF_CHUNK <= 96, "DfMask holds a chunk of 65..96 bytes");

// ---- pass A over the thread's chunk, a dword at a time: the bytes that differ from one row earlier (bytes before the block's
// first row differ by definition) as a mask, the first / last of them and, CRC: the chunk's CRC32 register (slicing by four:
// S = T0..T3) ----
template <bool CRC>
__device__ __forceinline__ void df_pass_a(const uint8_t *data, const uint32_t *S, uint32_t c0, uint32_t c1, uint32_t row, int &fne, int &lne,
                                          uint32_t &crc, DfMask &ne) {
    crc = threadIdx.x == 0 ? 0xFFFFFFFFu : 0u;
    uint64_t mlo = 0;
    uint32_t mhi = 0;
    uint32_t i = c0;
#pragma unroll
    for (uint32_t j = 0; j < DF_CHUNK / 4; ++j) {  // (c0 is a multiple of 4; a short chunk — a file's last block — leaves early)
        if (i + 4 > c1) break;
        const uint32_t v = *reinterpret_cast<const uint32_t *>(data + i);
        if (CRC) {
            const uint32_t x = crc ^ v;
            crc = S[768 + (x & 255u)] ^ S[512 + ((x >> 8) & 255u)] ^ S[256 + ((x >> 16) & 255u)] ^ S[x >> 24];
        }
        uint32_t xr;
        if (i >= row) {
            xr = v ^ reinterpret_cast<const DfU32 *>(data + i - row)->v;
        } else {
            xr = 0;
#pragma unroll
            for (uint32_t b = 0; b < 4; ++b) {
                const uint32_t idx = i + b;
                const uint32_t d = idx >= row ? (uint32_t)(data[idx] ^ data[idx - row]) : 0xFFu;
                xr |= d << (8 * b);
            }
        }
        // one bit per non-zero byte of xr: OR every byte's bits down to its bit 0, gather the four bit 0s
        uint32_t t = xr | (xr >> 4);
        t |= t >> 2;
        t |= t >> 1;
        t &= 0x01010101u;
        const uint32_t nib = ((t * 0x01020408u) >> 24) & 0xFu;
        if (j < 16) mlo |= (uint64_t)nib << (4 * j);
        else mhi |= nib << (4 * (j - 16));
        i += 4;
    }
    for (; i < c1; ++i) {  // (a file's last block: up to three bytes behind the last whole dword)
        const uint32_t v = data[i];
        if (CRC) crc = S[(crc ^ v) & 255u] ^ (crc >> 8);
        if (!(i >= row && data[i - row] == v)) {
            const uint32_t j = i - c0;
            if (j < 64) mlo |= 1ull << j;
            else mhi |= 1u << (j - 64);
        }
    }
    fne = mlo ? (int)c0 + (__ffsll((long long)mlo) - 1) : mhi ? (int)c0 + 64 + (__ffs((int)mhi) - 1) : 0x7fffffff;
    lne = mhi ? (int)c0 + 64 + (31 - __clz((int)mhi)) : mlo ? (int)c0 + (63 - __clzll((long long)mlo)) : -1;
    // the bits behind the chunk's end
    const uint32_t len = c1 - c0;
    if (len < 64) {
        mlo |= ~0ull << len;
        mhi = ~0u;
    } else {
        mhi |= ~0u << (len - 64);  // (len <= DF_CHUNK < 96)
    }
    ne.lo = mlo;
    ne.hi = mhi;
}

// The tokens of a thread's chunk [c0, c1), run by run (not position by position: lanes of a wave are in different runs, and a
// per-position walk pays for the longest forward scan at every step).  A run of bytes equal to one row earlier, [s, e), is cut
// into matches of 258 from s, then one match of r = (e - s) % 258 if r >= 3, else r literals — the same for every thread that
// sees part of the run.  prevNE / nextNE: the last unequal byte before the chunk (-1: none), the first one behind it (n: none).
// The walk reads the chunk's mask of unequal bytes (pass A), shifted along as it goes; LDS only for the literals' values.
template <typename Lit, typename Match>
__device__ __forceinline__ void df_tokens(const uint8_t *data, uint32_t c0, uint32_t c1, DfMask ne, int prevNE, int nextNE, Lit &&on_lit,
                                          Match &&on_match) {
    uint64_t lo = ne.lo;
    uint32_t hi = ne.hi;
    auto advance = [&](uint32_t by) {  // the mask moved on by `by` (1..DF_CHUNK) bytes, ones coming in at the top
        if (by >= 64) {
            lo = (by >= 96) ? ~0ull : (((uint64_t)hi >> (by - 64)) | (~0ull << (96 - by)));
            hi = ~0u;
        } else {
            lo = (lo >> by) | ((uint64_t)hi << (64 - by));  // (by >= 1: the shift is below 64)
            if (by > 32) lo |= ~0ull << (96 - by);           // (ones from behind the 96 bits the mask holds)
            hi = by >= 32 ? ~0u : ((hi >> by) | (~0u << (32 - by)));
        }
    };
    uint32_t i = c0;
    while (i < c1) {
        if (lo & 1ull) {
            on_lit((uint32_t)data[i]);
            ++i;
            advance(1u);
            continue;
        }
        // a run of equal bytes: up to the next unequal one (the mask's ones behind the chunk end it at c1 at the latest)
        const uint32_t run = lo ? (uint32_t)(__ffsll((long long)lo) - 1) : 64u + (uint32_t)(__ffs((int)hi) - 1);
        const uint32_t s = (i == c0) ? (uint32_t)(prevNE + 1) : i;
        const uint32_t hiend = i + run;             // end of the run inside this chunk (<= c1)
        const uint32_t e = hiend == c1 ? (uint32_t)nextNE : hiend;  // ... and its true end
        const uint32_t R = e - s, q258 = (R / 258u) * 258u, r = R - q258;
        uint32_t p = s + ((i - s + 257u) / 258u) * 258u;  // first match start >= i
        for (; p < min(s + q258, hiend); p += 258u) on_match(258u);
        const uint32_t tz = s + q258;  // tail zone [tz, e)
        if (r >= 3) {
            if (tz >= i && tz < hiend) on_match(r);
        } else {
            for (uint32_t q = max(i, tz); q < hiend; ++q) on_lit((uint32_t)data[q]);
        }
        i = hiend;
        advance(run);
    }
}

// ---- symbol counts of sampled blocks: block (blockIdx.x * nblocks / nsamp) of the file, tokenised as the encoder will ----
__global__ __launch_bounds__(DF_THREADS, 8) void k_df_sample_hist(const uint8_t *__restrict__ base, const PaySeg *__restrict__ segs, uint32_t nseg,
                                                               uint64_t total, uint64_t nblocks, uint32_t nsamp, uint32_t row,
                                                               uint32_t *__restrict__ ghist) {
    __shared__ __attribute__((aligned(16))) uint8_t data[DF_BLOCK];
    __shared__ uint32_t hist[288];
    __shared__ int part[DF_WAVES];
    __shared__ PayCur blk_cur;
    const int tid = threadIdx.x;
    for (int i = tid; i < 288; i += DF_THREADS) hist[i] = 0;
    const uint64_t blk = (uint64_t)blockIdx.x * nblocks / nsamp;
    uint32_t c0, c1;
    const uint32_t n = df_stage(data, &blk_cur, base, segs, nseg, total, blk, c0, c1);
    int fne, lne;
    uint32_t crc;
    DfMask ne;
    df_pass_a<false>(data, nullptr, c0, c1, row, fne, lne, crc, ne);
    const int prevNE = df_excl_max(lne, part);
    const int nextNE = df_rexcl_min(fne == 0x7fffffff ? (int)n : fne, (int)n, part);
    df_tokens(data, c0, c1, ne, prevNE, nextNE, [&](uint32_t v) { atomicAdd(&hist[v], 1u); },
              [&](uint32_t L) {
                  uint32_t nx, xv;
                  atomicAdd(&hist[257 + df_len_sym(L, nx, xv)], 1u);
              });
    __syncthreads();
    for (int i = tid; i < 286; i += DF_THREADS)
        if (hist[i]) atomicAdd(&ghist[i], hist[i]);
}

// ---- the file's Huffman code and block header out of the sampled counts: one workgroup.  Every symbol gets a code (count
// + 1): a block may hold literals or match lengths the sample did not; the end-of-block symbol counts once per sampled block.
// The used symbols are rank-sorted by the workgroup, thread 0 builds the tree over the sorted list and the header. ----
__global__ __launch_bounds__(DF_CODE_THREADS) void k_df_build_code(const uint32_t *__restrict__ ghist, uint32_t nsamp, uint32_t row,
                                                                   DfCode *__restrict__ code) {
    __shared__ uint32_t hist[288];
    __shared__ uint16_t lcode[288];
    __shared__ uint8_t llen[288];
    __shared__ uint8_t hdr[768];
    __shared__ uint32_t hdr_bits;
    __shared__ uint16_t h_order[288], h_kid0[576], h_kid1[576];
    __shared__ uint32_t h_wgt[576], h_cnt[33], s_cfl[19];
    __shared__ uint8_t s_cll[19];
    __shared__ uint16_t s_clc[19];
    __shared__ uint8_t h_dep[576], cl_sym[320], cl_extra[320];
    const int tid = threadIdx.x;
    for (int i = tid; i < 288; i += DF_CODE_THREADS) hist[i] = i < 286 ? (i == 256 ? max(1u, nsamp) : ghist[i] + 1u) : 0u;
    for (int i = tid; i < 768; i += DF_CODE_THREADS) hdr[i] = 0;
    __syncthreads();
    for (int sy = tid; sy < 286; sy += DF_CODE_THREADS) {
        llen[sy] = 0;
        const uint32_t f = hist[sy];
        uint32_t rank = 0;
        for (int j = 0; j < 286; ++j) {
            const uint32_t fj = hist[j];
            rank += (fj < f || (fj == f && j < sy)) ? 1u : 0u;
        }
        h_order[rank] = (uint16_t)sy;
    }
    __syncthreads();
    if (tid == 0) df_huff_from_sorted(hist, 286, 15, llen, h_order, h_wgt, h_kid0, h_kid1, h_dep, h_cnt);
    __syncthreads();
    for (int sy = tid; sy < 286; sy += DF_CODE_THREADS) lcode[sy] = (uint16_t)df_canon_code(llen, 286, sy);
    if (tid == 0) {
        int dsym = 0;
        while (dsym < 29 && DF_DIST_BASE[dsym + 1] <= row) ++dsym;
        const int nlit = 286, ndist = dsym + 1;
        // code lengths of both alphabets, run-length coded (16/17/18); hist[] is reused for their counts
        uint32_t *cf = hist;  // 19 counters at hist[0..18]: the literal counts are no longer needed
        auto all_at = [&](int a) -> uint32_t { return a < nlit ? llen[a] : ((a - nlit == dsym) ? 1u : 0u); };
        const int nall = nlit + ndist;
        int ncl_tok = 0;
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
        llen[286] = (uint8_t)dsym;  // (slot 286 carries the distance symbol for the encoder)
        lcode[287] = 0;
        llen[287] = 0;
    }
    __syncthreads();
    for (int i = tid; i < 288; i += DF_CODE_THREADS) {
        code->lcode[i] = lcode[i];
        code->llen[i] = llen[i];
    }
    for (int i = tid; i < 768; i += DF_CODE_THREADS) code->hdr[i] = hdr[i];
    if (tid == 0) code->hdr_bits = hdr_bits;
}

// (64 vector and 80 scalar registers: two workgroups = 32 waves per CU, MI355X_MICROARCH.md's occupancy rules)
__global__ __launch_bounds__(DF_THREADS, 8) __attribute__((amdgpu_num_sgpr(80))) void k_row_deflate(const uint8_t *__restrict__ base, const PaySeg *__restrict__ segs,
                                                            uint32_t nseg, uint64_t total, uint64_t first_block,
                                                            uint32_t row, const uint32_t *__restrict__ crc_tabs,
                                                            const DfCode *__restrict__ code, uint8_t *__restrict__ slots,
                                                            uint32_t *__restrict__ sizes, uint32_t force_stored) {
    __shared__ __attribute__((aligned(16))) uint8_t data[DF_BLOCK];
    __shared__ uint32_t S[1024];  // CRC-32, slicing by four
    __shared__ uint32_t lcl[288];  // a symbol's code (low half) and its length (high half): one LDS word per token
    __shared__ uint32_t crcp[DF_THREADS];
    __shared__ int part[DF_WAVES];
    __shared__ uint32_t blk_crc, crc_acc;
    __shared__ PayCur blk_cur;
    const int tid = threadIdx.x;
    DF_PH_DECL
    const uint64_t blk = first_block + blockIdx.x;
    uint8_t *slot = slots + (uint64_t)blockIdx.x * 65536;
    if (tid == 0) crc_acc = 0;
    S[tid] = crc_tabs[tid];
    if (tid < 288) lcl[tid] = (uint32_t)code->lcode[tid] | ((uint32_t)code->llen[tid] << 16);
    const uint32_t hdr_bits = code->hdr_bits;
    uint32_t c0, c1;
    const uint32_t n = df_stage(data, &blk_cur, base, segs, nseg, total, blk, c0, c1);  // bytes of this block
    DF_PH(0)

    // ---- pass A: chunk CRC and the chunk's first / last byte that differs from one row earlier ----
    int fne, lne;
    uint32_t crc;
    DfMask ne;
    df_pass_a<true>(data, S, c0, c1, row, fne, lne, crc, ne);
    crcp[tid] = crc;
    DF_PH(1)
    const int prevNE = df_excl_max(lne, part);
    const int nextNE = df_rexcl_min(fne == 0x7fffffff ? (int)n : fne, (int)n, part);
    DF_PH(2)
    // CRC32 of the block out of the chunk CRCs.  The register is linear in its state: after a further
    // chunk, state = shift(state) ^ crc(chunk), so the block's CRC is the XOR over the full chunks t of
    // shift^(F-1-t)(crc_t), F = number of full chunks; shift^(2^j) is a table set (crc_tabs[1024 +
    // 1024 j ..]), so every thread applies at most DF_CRC_LEVELS of them, in parallel; a short tail chunk (last
    // block of a file) is appended bytewise by thread 0.
    {
        const uint32_t F = n / DF_CHUNK;
        if ((uint32_t)tid < F) {
            uint32_t x = crc;
            const uint32_t e = F - 1 - (uint32_t)tid;
            for (uint32_t j = 0; j < DF_CRC_LEVELS; ++j)
                if ((e >> j) & 1u) {
                    const uint32_t *T = crc_tabs + 1024 + 1024 * j;
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
                for (uint32_t z = 0; z < tail; ++z) c = S[c & 255u] ^ (c >> 8);
                c ^= crcp[F];
            }
        }
        blk_crc = c ^ 0xFFFFFFFFu;
    }
    DF_PH(3)
    const uint32_t dsym = lcl[286] >> 16;
    const uint32_t dbits = 1u + DF_DIST_EXTRA[dsym];
    const uint32_t dval = (row - DF_DIST_BASE[dsym]) << 1;  // code 0 in bit 0, extra bits above

    // ---- pass C: bit counts, offsets, emission.  One walk = a lambda over the tokens ----
    auto walk = [&](auto &&emit) {
        df_tokens(data, c0, c1, ne, prevNE, nextNE,
                  [&](uint32_t v) {
                      const uint32_t cl = lcl[v];
                      emit(cl & 0xFFFFu, cl >> 16);
                  },
                  [&](uint32_t L) {
                      uint32_t nx, xv;
                      const int ls = df_len_sym(L, nx, xv);
                      const uint32_t cl = lcl[257 + ls];
                      const uint32_t lb = cl >> 16;
                      emit((cl & 0xFFFFu) | (xv << lb), lb + nx);  // the length code and its extra bits: 20 bits at most
                      emit(dval, dbits);                           // the distance code (one bit) and its extra bits: 14 at most
                  });
        if (c0 < c1 && c1 == n) {  // end of block, by the owner of the last byte
            const uint32_t cl = lcl[256];
            emit(cl & 0xFFFFu, cl >> 16);
        }
    };
    uint32_t mybits = 0;
    walk([&](uint32_t, uint32_t nb) { mybits += nb; });
    DF_PH(4)
    // the deflate stream starts behind the 18-byte BGZF header: the file's block header, then the threads' tokens in order
    uint32_t tokbits;
    const uint32_t before = df_excl_sum(mybits, reinterpret_cast<uint32_t *>(part), tokbits);
    const uint32_t allbits = hdr_bits + tokbits;
    const uint32_t sbytes = (allbits + 7) / 8;
    const bool fits = sbytes <= 65536 - 18 - 8 && !force_stored;  // (force_stored: test hook for the fallback)
    uint32_t *slotw = reinterpret_cast<uint32_t *>(slot);
    DF_PH(5)
    if (fits) {
        // the header's whole bytes, copied by everybody; its last bits open thread 0's stream
        const uint32_t hb = hdr_bits >> 3, hr = hdr_bits & 7u;
        for (uint32_t b = tid; b < hb; b += DF_THREADS) slot[18 + b] = code->hdr[b];
        // every thread ORs its bits into the zeroed slot: words shared with a neighbour atomically
        uint64_t bitpos = tid == 0 ? 144ull + 8ull * hb : 144ull + hdr_bits + before;
        uint64_t acc = 0;
        uint32_t nacc = (uint32_t)(bitpos & 31);
        uint32_t widx = (uint32_t)(bitpos >> 5);
        const uint32_t first_w = widx;
        if (tid == 0 && hr) {
            acc = (uint64_t)(code->hdr[hb] & ((1u << hr) - 1u)) << nacc;
            nacc += hr;
        }
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
        if (nacc >= 32) flush_word(false);  // (thread 0: the header's last bits may have filled the word)
        walk(put);
        if (nacc) {
            nacc += 32;  // flush_word subtracts 32
            flush_word(true);
        }
    }
    __syncthreads();
    DF_PH(6)
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
    DF_PH(7)
#ifdef PG_DF_PHASE
    if (tid == 0) atomicAdd(&pg_df_phase_cycles[(blockIdx.x & 255u) * 16u + 15], 1ull);
#endif
}

// finished blocks of a batch packed back to back (what goes to the file): offsets by one workgroup,
// then one workgroup per block copies its bytes (destination at byte alignment)
__global__ __launch_bounds__(1024) void k_bgzf_offsets(const uint32_t *__restrict__ sizes, uint32_t nb, uint32_t *__restrict__ offs) {
    __shared__ uint32_t part[DF_WAVES];
    const int tid = threadIdx.x;
    const uint32_t per = (nb + 1023) / 1024;
    uint32_t sum = 0;
    for (uint32_t i = tid * per; i < min(nb, (tid + 1) * per); ++i) sum += sizes[i];
    uint32_t all;
    uint32_t base = df_excl_sum(sum, part, all);  // (a scan by wave shuffles: the thousand-step loop over LDS it replaces took 35 us per batch)
    for (uint32_t i = tid * per; i < min(nb, (tid + 1) * per); ++i) {
        offs[i] = base;
        base += sizes[i];
    }
    if (tid == 0) offs[nb] = all;
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

// the file's code: symbol counts of up to DF_SAMPLE_BLOCKS of its blocks, then lengths / codes / header -> `code`
// (DF_CODE_BYTES), `hist` = DF_HIST_WORDS words of scratch
hipError_t launch_deflate_code(hipStream_t st, const uint8_t *base, const PaySeg *segs, uint32_t nseg, uint64_t total, uint32_t row,
                               uint32_t *hist, void *code) {
    const uint64_t nblocks = (total + DF_BLOCK - 1) / DF_BLOCK;
    if (nblocks == 0) return hipSuccess;
    const uint32_t nsamp = (uint32_t)std::min<uint64_t>(nblocks, DF_SAMPLE_BLOCKS);
    hipError_t e = hipMemsetAsync(hist, 0, DF_HIST_WORDS * 4, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_df_sample_hist, dim3(nsamp), dim3(DF_THREADS), 0, st, base, segs, nseg, total, nblocks, nsamp, row, hist);
    hipLaunchKernelGGL(k_df_build_code, dim3(1), dim3(DF_CODE_THREADS), 0, st, hist, nsamp, row, static_cast<DfCode *>(code));
    return hipGetLastError();
}

hipError_t launch_row_deflate(hipStream_t st, const uint8_t *base, const PaySeg *segs, uint32_t nseg, uint64_t total,
                              uint64_t first_block, uint32_t nblocks, uint32_t row, const uint32_t *crc_tabs, const void *code,
                              uint8_t *slots, uint32_t *sizes, uint32_t force_stored, uint32_t *offs, uint8_t *packed) {
    if (nblocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_row_deflate, dim3(nblocks), dim3(DF_THREADS), 0, st, base, segs, nseg, total, first_block, row,
                       crc_tabs, static_cast<const DfCode *>(code), slots, sizes, force_stored);
    hipLaunchKernelGGL(k_bgzf_offsets, dim3(1), dim3(1024), 0, st, sizes, nblocks, offs);
    hipLaunchKernelGGL(k_bgzf_pack, dim3(nblocks), dim3(256), 0, st, slots, sizes, offs, packed);
    return hipGetLastError();
}

}  // namespace pg

#ifdef PG_DF_PHASE
extern "C" int pg_debug_df_phase_cycles(unsigned long long *out16, int reset) {
    hipDeviceSynchronize();
    static unsigned long long all[256 * 16];
    if (hipMemcpyFromSymbol(all, HIP_SYMBOL(pg::pg_df_phase_cycles), sizeof all) != hipSuccess) return -1;
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    for (int b = 0; b < 256; ++b)
        for (int i = 0; i < 16; ++i) out16[i] += all[b * 16 + i];
    if (reset) {
        for (auto &x : all) x = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(pg::pg_df_phase_cycles), all, sizeof all) != hipSuccess) return -1;
    }
    return 0;
}
#endif
