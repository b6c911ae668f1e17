// pg_kernels.hip — table construction / maintenance kernels and sequence packing for gfx950
// (CDNA4, wave64).  Integer / HBM-bound work: no MFMA.  The anchor kernels live in pg_anchor.hip.
//   k_pack         1 byte in, 0.375 byte out per base, streaming                -> HBM roofline
//   k_insert_seq   one thread per k-mer: canonical key, minimizer, CAS + atomic OR on a random
//                  table line                                                   -> HBM / atomics
//   k_insert_keys  same for (key, counter) pairs (KMC1 import)
//   k_rehash       every occupied slot of a table into a bigger / tighter one
//   k_export       (key, counter) pairs of one 32-genome group (KMC1 export)
//   k_counters     literal GetCountersForRead equivalent (single-lane lookup per position)
//
// Replaces (reference, kjenike/panagram): kmc -ci1 + kmc_tools transform set_counts + kmc_tools
// complex -ocsum (workflow/Snakefile:54-110, index.py:407-426) and CKMCFile::OpenForRA /
// GetCountersForRead (cpp/anchor.cpp:28-31,148; index.py:855-860,934-935).
#include "pg_kernels.h"

namespace pg {

// ---------------------------------------------------------------------------
// table init: every thread writes one 16-byte chunk of a bucket
// ---------------------------------------------------------------------------
// mode 0: slots layout, 1: split layout's key lines, 2 + S: inline layout with S keys per 128-byte line
__global__ __launch_bounds__(256) void k_table_init(uint4 *chunks, uint64_t nchunks, uint32_t mode) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < nchunks; i += stride) {
        // slots layout: {EMPTY key, mask0 = 0, mask1 = 0}; split layout: two EMPTY keys (the mask array is memset);
        // inline layout: the line's first S 8-byte words are EMPTY keys, the mask blocks behind them zero
        if (mode >= 2) {
            const uint32_t S = mode - 2u, k0 = 2u * (uint32_t)(i & 7u);  // this chunk's two 8-byte words of its line
            const uint32_t a = k0 < S ? ~0u : 0u, b = k0 + 1u < S ? ~0u : 0u;
            chunks[i] = make_uint4(a, a, b, b);
        } else {
            chunks[i] = mode ? make_uint4(~0u, ~0u, ~0u, ~0u) : make_uint4(~0u, ~0u, 0u, 0u);
        }
    }
}

// ---------------------------------------------------------------------------
// ASCII -> 2 bit/base words + "not ACGT" plane.  One thread = 32 bases.
// code = ((c>>1)&3) ^ ((c>>2)&1)... : A(0x41)->0 C(0x43)->1 G(0x47)->2 T(0x54)->3,
// same for lower case; valid iff (c & 0xDF) in {A,C,G,T}.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pack_byte(uint32_t c, uint32_t i, uint64_t &w, uint32_t &nm) {
    uint32_t x = (c >> 1) & 3u;       // A:0 C:1 G:3 T:2
    uint32_t code = x ^ (x >> 1);     // A:0 C:1 G:2 T:3
    uint32_t u = c & 0xDFu;
    bool ok = (u == 0x41u) | (u == 0x43u) | (u == 0x47u) | (u == 0x54u);
    w |= (uint64_t)(ok ? code : 0u) << (2 * i);
    nm |= (ok ? 0u : 1u) << i;
}

__global__ __launch_bounds__(256) void k_pack(const uint8_t *ascii, uint64_t len, uint64_t *seqw,
                                              uint32_t *nmw, uint64_t nwords, uint32_t *has_n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwords) return;
    uint64_t w = 0;
    uint32_t nm = 0;
    uint64_t base = i * 32;
    const bool aligned = ((reinterpret_cast<uintptr_t>(ascii) & 15) == 0);
    if (aligned && base + 32 <= len) {
        const uint4 *p = reinterpret_cast<const uint4 *>(ascii + base);
        uint4 a = p[0], b = p[1];
        uint32_t d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pack_byte((d[q] >> (8 * r)) & 0xFFu, 4 * q + r, w, nm);
        }
    } else {
        for (uint32_t r = 0; r < 32; ++r) {
            uint64_t idx = base + r;
            if (idx < len) pack_byte(ascii[idx], r, w, nm);
        }
    }
    seqw[i] = w;
    nmw[i] = nm;
    if (nm) atomicOr(has_n, 1u);
}

// ---------------------------------------------------------------------------
// FASTA text -> packed planes.  The host only locates the header lines; the bytes between two
// headers (sequence lines, any wrapping, any line ending) are compacted here: white space is
// dropped, every other byte becomes one base, in text order.  The text is cut at absolute
// multiples of 4096 bytes (clipped to the record), one workgroup per chunk:
//   k_text_count  bases per chunk            k_text_scan  one wave per record: first base index of
//   k_text_pack   each chunk's bases into LDS words, whole words stored, the two edge words of a
//                 chunk (shared with its neighbours) OR-ed into the pre-zeroed planes
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool text_is_ws(uint32_t c) { return c == 0x20u || (c - 9u) <= 4u; }

// the 16 text bytes of this thread (aligned load) and the mask of those that are bases of the chunk
__device__ __forceinline__ uint32_t text_load16(const uint8_t *text, const TextChunk &ch, int tid, uint32_t d[4]) {
    const uint64_t a0 = (ch.off & ~15ull) + 16ull * tid;
    const uint4 q = *reinterpret_cast<const uint4 *>(text + a0);
    d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint64_t pos = a0 + j;
        const uint32_t c = (d[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        if (pos >= ch.off && pos < ch.off + ch.len && !text_is_ws(c)) m |= 1u << j;
    }
    return m;
}

// a chunk lies inside one aligned 4096-byte window of the text: 256 threads x 16 bytes cover it
constexpr int TEXT_THREADS = 256;

__global__ __launch_bounds__(TEXT_THREADS) void k_text_count(const uint8_t *__restrict__ text,
                                                             const TextChunk *__restrict__ chunks,
                                                             uint32_t *__restrict__ counts) {
    __shared__ uint32_t tot;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    const TextChunk ch = chunks[blockIdx.x];
    uint32_t d[4];
    const uint32_t n = __popc(text_load16(text, ch, threadIdx.x, d));
    uint32_t v = n;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor((int)v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&tot, v);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = tot;
}

__global__ __launch_bounds__(64) void k_text_scan(const uint32_t *__restrict__ counts,
                                                  const uint64_t *__restrict__ rec_chunk0, uint64_t *__restrict__ base,
                                                  uint64_t *__restrict__ rec_len) {
    const uint32_t r = blockIdx.x;
    const int lane = threadIdx.x;
    const uint64_t c0 = rec_chunk0[r], c1 = rec_chunk0[r + 1];
    uint64_t running = 0;
    for (uint64_t i0 = c0; i0 < c1; i0 += 64) {
        const uint64_t i = i0 + lane;
        const uint32_t v = i < c1 ? counts[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= o) incl += up;
        }
        if (i < c1) base[i] = running + incl - v;
        running += (uint32_t)__shfl((int)incl, 63);
    }
    if (lane == 0) rec_len[r] = running;
}

__global__ __launch_bounds__(TEXT_THREADS) void k_text_pack(const uint8_t *__restrict__ text,
                                                            const TextChunk *__restrict__ chunks,
                                                            const uint64_t *__restrict__ base,
                                                            const SeqDesc *__restrict__ sd, uint64_t *__restrict__ seqw,
                                                            uint32_t *__restrict__ nmw, uint32_t *__restrict__ has_n) {
    constexpr int NW = 4096 / 32 + 2;
    __shared__ unsigned long long lw[NW];
    __shared__ uint32_t ln[NW];
    __shared__ uint32_t wave_tot[TEXT_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < NW; i += TEXT_THREADS) {
        lw[i] = 0;
        ln[i] = 0;
    }
    const TextChunk ch = chunks[blockIdx.x];
    uint32_t d[4];
    const uint32_t m = text_load16(text, ch, tid, d);
    const uint32_t n = __popc(m);
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < TEXT_THREADS / 64; ++w) {
        if (w < wv) before += wave_tot[w];
        total += wave_tot[w];
    }
    if (total == 0) return;  // block-uniform
    const uint64_t b_first = base[blockIdx.x];
    const uint64_t w_first = b_first >> 5;
    uint64_t b = b_first + before + incl - n;  // this thread's first base
    uint32_t cur = (uint32_t)((b >> 5) - w_first);
    unsigned long long wacc = 0;
    uint32_t nacc = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (m & (1u << j)) {
            const uint32_t wi = (uint32_t)((b >> 5) - w_first);
            if (wi != cur) {
                if (wacc) atomicOr(&lw[cur], wacc);
                if (nacc) atomicOr(&ln[cur], nacc);
                wacc = 0;
                nacc = 0;
                cur = wi;
            }
            uint64_t w1 = 0;
            uint32_t n1 = 0;
            pack_byte((d[j >> 2] >> (8 * (j & 3))) & 0xFFu, (uint32_t)(b & 31), w1, n1);
            wacc |= w1;
            nacc |= n1;
            ++b;
        }
    }
    if (wacc) atomicOr(&lw[cur], wacc);
    if (nacc) atomicOr(&ln[cur], nacc);
    __syncthreads();
    const uint32_t nw = (uint32_t)(((b_first + total - 1) >> 5) - w_first) + 1;
    const uint64_t g0 = sd[ch.rec].seq_off + w_first;
    bool any_n = false;
    for (uint32_t i = tid; i < nw; i += TEXT_THREADS) {
        const unsigned long long v = lw[i];
        const uint32_t nn = ln[i];
        any_n |= nn != 0;
        if (i == 0 || i == nw - 1) {
            if (v) atomicOr(reinterpret_cast<unsigned long long *>(&seqw[g0 + i]), v);
            if (nn) atomicOr(&nmw[g0 + i], nn);
        } else {
            seqw[g0 + i] = v;
            nmw[g0 + i] = nn;
        }
    }
    if (any_n) atomicOr(&has_n[ch.rec], 1u);
}

// ---------------------------------------------------------------------------
// k-mer set construction: one thread per k-mer position, insert-or-OR.
// counters[0] += newly claimed keys; counters[1] = overflow flag.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_insert_seq(SubTable st, int w, uint32_t bits, int k,
                                                    const uint64_t *seqw, const uint32_t *nmw,
                                                    const uint32_t *has_n, uint64_t nkmers,
                                                    unsigned long long *counters, uint32_t max_probe, int count_mode) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const bool hasn = (*has_n != 0);
    uint32_t claimed = 0;
    for (; p < nkmers; p += stride) {
        if (hasn && extract_nmask(nmw, p, k)) continue;
        uint64_t key = canonical_from_le(extract_bases(seqw, p), k);
        int r = count_mode ? lane_insert<true>(st, key, w, bits, max_probe) : lane_insert<false>(st, key, w, bits, max_probe);
        if (r < 0) atomicOr(reinterpret_cast<unsigned int *>(&counters[1]), 1u);
        else claimed += r;
    }
    if (claimed) atomicAdd(&counters[0], (unsigned long long)claimed);
}

// ---------------------------------------------------------------------------
// Distinct canonical k-mers of the inputs BEFORE any table exists: a HyperLogLog sketch (2^16
// registers).  One streaming pass (no table access); sizes the table once, so that it is neither
// re-hashed while it grows nor held twice in HBM.  register = hash >> 48, value = 1 + leading zeros
// of the remaining 48 bits.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sketch(int k, const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                                                uint64_t nkmers, uint32_t *regs) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const bool hasn = (*has_n != 0);
    for (; p < nkmers; p += stride) {
        if (hasn && extract_nmask(nmw, p, k)) continue;
        const uint64_t h = sketch_hash(canonical_from_le(extract_bases(seqw, p), k));
        const uint32_t idx = (uint32_t)(h >> (64 - SKETCH_BITS));
        const uint64_t rest = h << SKETCH_BITS;
        const uint32_t rho = rest ? (uint32_t)__clzll((long long)rest) + 1u : (65u - SKETCH_BITS);
        if (regs[idx] < rho) atomicMax(&regs[idx], rho);  // (a stale read only costs a redundant atomic)
    }
}

// the same over a whole seqset in ONE launch: job j = positions [start, start + SKETCH_JOB) of contig c (a launch per
// contig cost an assembly of 20 000 contigs 76 ms per genome)
__global__ __launch_bounds__(256) void k_sketch_set(int k, const SeqDesc *__restrict__ sd, const uint2 *__restrict__ jobs,
                                                    const uint64_t *seqw_all, const uint32_t *nmw_all, const uint32_t *has_n,
                                                    uint32_t *regs) {
    const uint2 job = jobs[blockIdx.x];
    const SeqDesc d = sd[job.x];
    const uint64_t nkmers = d.len - (uint64_t)k + 1, p0 = (uint64_t)job.y * SKETCH_JOB, p1 = min(nkmers, p0 + SKETCH_JOB);
    const uint64_t *seqw = seqw_all + d.seq_off;
    const uint32_t *nmw = nmw_all + d.seq_off;
    const bool hasn = has_n[job.x] != 0;
    for (uint64_t p = p0 + threadIdx.x; p < p1; p += 256) {
        if (hasn && extract_nmask(nmw, p, k)) continue;
        const uint64_t h = sketch_hash(canonical_from_le(extract_bases(seqw, p), k));
        const uint32_t idx = (uint32_t)(h >> (64 - SKETCH_BITS));
        const uint64_t rest = h << SKETCH_BITS;
        const uint32_t rho = rest ? (uint32_t)__clzll((long long)rest) + 1u : (65u - SKETCH_BITS);
        if (regs[idx] < rho) atomicMax(&regs[idx], rho);
    }
}

__global__ __launch_bounds__(256) void k_insert_keys(SubTable st, int w, const uint64_t *keys,
                                                     const uint32_t *vals, uint64_t n,
                                                     unsigned long long *counters, uint32_t max_probe) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t claimed = 0;
    for (; i < n; i += stride) {
        uint32_t v = vals[i];
        if (v == 0) continue;  // a zero counter reads the same as an absent key
        int r = lane_insert<false, true>(st, keys[i], w, v, max_probe);
        if (r < 0) atomicOr(reinterpret_cast<unsigned int *>(&counters[1]), 1u);
        else claimed += r;
    }
    if (claimed) atomicAdd(&counters[0], (unsigned long long)claimed);
}

// ---------------------------------------------------------------------------
// KMC database import on the GPU.  A .kmc_suf file is an array of records (suffix bytes, most significant
// symbol first, then the counter, little-endian), sorted inside each bin; the .kmc_pre file gives, for every
// (bin, prefix) pair, the index of its first record — `nlut` monotone entries, bin-major, prefix-minor; the KMC1
// layout (kmc_tools output) is the one-bin case, the KMC2 layout (kmc output) has one LUT per signature bin.
// A thread takes one record: its prefix is (index of the last LUT entry <= record number) mod prefixes-per-bin
// (binary search over the LUT, which stays in L2), key = prefix : suffix, counter filtered by [min, max] as
// CKMCFile::GetCountersForRead does; insert-or-OR into 32-genome group word w.
// Replaces CKMCFile::OpenForRA (cpp/anchor.cpp:29, index.py:859-860), which reads the whole file to host RAM.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_import_kmc(SubTable st, int w, const uint8_t *__restrict__ rec,
                                                    uint64_t first_record, uint64_t nrec,
                                                    const unsigned long long *__restrict__ lut, uint64_t nlut,
                                                    uint32_t prefixes_per_bin, uint32_t suffix_bytes, uint32_t counter_bytes,
                                                    uint32_t min_count, uint32_t max_count,
                                                    unsigned long long *counters, uint32_t max_probe, uint32_t phase,
                                                    uint32_t dense_above) {
    // phase 0: only records whose mask has more than dense_above bits; 1: only the others; 2: all.  The keys most genomes
    // share go in first and take their minimizer's home line — they are the ones most look-ups ask for
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t rb = suffix_bytes + counter_bytes;
    uint32_t claimed = 0;
    for (; i < nrec; i += stride) {
        const uint8_t *q = rec + i * rb;
        uint32_t c = counter_bytes ? 0u : 1u;  // (KMC writes no counter bytes when every count is 1)
        for (uint32_t b = 0; b < counter_bytes; ++b) c |= (uint32_t)q[suffix_bytes + b] << (8 * b);
        if (c == 0 || c < min_count || c > max_count) continue;  // outside [min, max] reads as absent
        if (phase < 2u && ((uint32_t)__popc(c) > dense_above) != (phase == 0u)) continue;
        const uint64_t r = first_record + i;
        uint64_t lo = 0, hi = nlut;  // last entry <= r  (lut[0] = 0 <= r always)
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (lut[mid] <= r) lo = mid;
            else hi = mid;
        }
        uint64_t key = lo & (uint64_t)(prefixes_per_bin - 1);
        for (uint32_t b = 0; b < suffix_bytes; ++b) key = (key << 8) | q[b];
        int res = lane_insert<false, true>(st, key, w, c, max_probe);
        if (res < 0) atomicOr(reinterpret_cast<unsigned int *>(&counters[1]), 1u);
        else claimed += res;
    }
    if (claimed) atomicAdd(&counters[0], (unsigned long long)claimed);
}

// re-hash every occupied slot of `src` into `dst` (same W)
// (phase / dense_above as in k_import_kmc: the keys of most genomes first)
__global__ __launch_bounds__(256) void k_rehash(SubTable src, SubTable dst, unsigned long long *counters,
                                                uint32_t max_probe, uint32_t phase, uint32_t dense_above) {
    const int ns = (int)src.slots;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t nslots = src.nbuckets * ns;
    uint32_t claimed = 0;
    for (; i < nslots; i += stride) {
        uint64_t b = i / ns;
        int s = (int)(i - b * ns);
        uint64_t key = *key_ptr(src, b, (uint32_t)s);
        if (key >= TOMB_KEY) continue;  // empty, or a retired copy
        if (phase < 2u) {
            uint32_t pc = 0;
            for (uint32_t w = 0; w < src.W; ++w) pc += (uint32_t)__popc(*mask_ptr(src, b, (uint32_t)s, w));
            if ((pc > dense_above) != (phase == 0u)) continue;
        }
        for (uint32_t w = 0; w < src.W; ++w) {
            uint32_t m = *mask_ptr(src, b, (uint32_t)s, w);
            if (m == 0 && w > 0) continue;
            int r = lane_insert(dst, key, (int)w, m, max_probe);
            if (r < 0) atomicOr(reinterpret_cast<unsigned int *>(&counters[1]), 1u);
            else claimed += r;
        }
    }
    if (claimed) atomicAdd(&counters[0], (unsigned long long)claimed);
}

// kmc -ci<min_count>: every key of a private occurrence-count table (word 0 = count) that was seen
// at least min_count times gets `bits` OR-ed into word w of the pan table
__global__ __launch_bounds__(256) void k_merge_min(SubTable src, SubTable dst, int w, uint32_t bits, uint32_t min_count,
                                                   unsigned long long *counters, uint32_t max_probe) {
    const int ns = (int)src.slots;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t nslots = src.nbuckets * ns;
    uint32_t claimed = 0;
    for (; i < nslots; i += stride) {
        uint64_t b = i / ns;
        int s = (int)(i - b * ns);
        uint64_t key = *key_ptr(src, b, (uint32_t)s);
        if (key == EMPTY_KEY) continue;
        if (*mask_ptr(src, b, (uint32_t)s, 0) < min_count) continue;
        int r = lane_insert(dst, key, w, bits, max_probe);
        if (r < 0) atomicOr(reinterpret_cast<unsigned int *>(&counters[1]), 1u);
        else claimed += r;
    }
    if (claimed) atomicAdd(&counters[0], (unsigned long long)claimed);
}

// keys that do not sit in the home line of their group (they spilled down the probe sequence)
__global__ __launch_bounds__(256) void k_count_spill(SubTable st, unsigned long long *counters) {
    const int ns = (int)st.slots;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t nslots = st.nbuckets * ns;
    uint32_t spilled = 0;
    for (; i < nslots; i += stride) {
        const uint64_t b = i / ns;
        const int s = (int)(i - b * ns);
        const uint64_t key = *key_ptr(st, b, (uint32_t)s);
        if (key >= TOMB_KEY) continue;
        if (home_of_group(group_of(st, key), st.nbuckets) != (uint32_t)b) ++spilled;
    }
    if (spilled) atomicAdd(&counters[0], (unsigned long long)spilled);
}

// export (key, mask word w) of every slot whose word w is non-zero
__global__ __launch_bounds__(256) void k_export(SubTable st, int w, uint64_t *keys, uint32_t *vals,
                                                uint64_t cap, unsigned long long *count) {
    const int ns = (int)st.slots;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t nslots = st.nbuckets * ns;
    for (; i < nslots; i += stride) {
        uint64_t b = i / ns;
        int s = (int)(i - b * ns);
        uint64_t key = *key_ptr(st, b, (uint32_t)s);
        if (key >= TOMB_KEY) continue;  // empty, or a retired copy (whose mask a racing finder may have written before it was retired)
        uint32_t m = *mask_ptr(st, b, (uint32_t)s, (uint32_t)w);
        if (m == 0) continue;
        unsigned long long idx = atomicAdd(count, 1ull);
        if (keys && idx < cap) {
            keys[idx] = key;
            vals[idx] = m;
        }
    }
}

// GetCountersForRead equivalent for one 32-genome group: one thread per position
__global__ __launch_bounds__(256) void k_counters(SubTable st, int w, int k, const uint64_t *seqw,
                                                  const uint32_t *nmw, const uint32_t *has_n,
                                                  uint64_t nkmers, uint32_t *out) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const bool hasn = (*has_n != 0);
    for (; p < nkmers; p += stride) {
        uint32_t r = 0;
        if (!(hasn && extract_nmask(nmw, p, k))) {
            uint64_t key = canonical_from_le(extract_bases(seqw, p), k);
            lane_lookup(st, key, (uint32_t)w, r);
        }
        out[p] = r;
    }
}

// ---------------------------------------------------------------------------
// host-side launch wrappers
// ---------------------------------------------------------------------------
static inline unsigned grid_for(uint64_t n, unsigned block, unsigned cap) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

hipError_t launch_table_init(hipStream_t st, const SubTable &t) {
    const uint64_t nchunks = t.nbuckets * line_bytes(t) / 16;
    hipLaunchKernelGGL(k_table_init, dim3(grid_for(nchunks, 256, 256 * 32)), dim3(256), 0, st,
                       reinterpret_cast<uint4 *>(t.buckets), nchunks,
                       t.layout == LAYOUT_INLINE ? 2u + t.slots : t.layout == LAYOUT_SPLIT ? 1u : 0u);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && t.layout == LAYOUT_SPLIT)
        e = hipMemsetAsync(t.masks, 0, t.nbuckets * t.slots * 4ull * t.W, st);
    return e;
}

// contigs cut out of longer ones (pg_seqset_slice) were copied word by word: the bases behind a contig's last one are
// cleared from its last word — every packed contig ends in zero bits, as k_pack leaves it
__global__ void k_seq_tailmask(const SeqDesc *__restrict__ sd, uint32_t n, uint64_t *__restrict__ seqw, uint32_t *__restrict__ nmw) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const SeqDesc d = sd[c];
    const uint32_t rem = (uint32_t)(d.len & 31u);
    if (!rem) return;
    const uint64_t w = d.seq_off + (d.len >> 5);
    seqw[w] &= (1ull << (2 * rem)) - 1ull;
    nmw[w] &= (1u << rem) - 1u;
}
hipError_t launch_seq_tailmask(hipStream_t st, const SeqDesc *sd, uint32_t n, uint64_t *seqw, uint32_t *nmw) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_seq_tailmask, dim3((n + 255) / 256), dim3(256), 0, st, sd, n, seqw, nmw);
    return hipGetLastError();
}

// contigs of several seqsets copied into one (pg_seqset_concat): ONE launch whatever their number — a device-to-device
// copy per plane and contig cost an assembly of 20 000 contigs 9 us each.  Block (i, y) copies words [y, y + 1) x 65536 of job i.
constexpr uint32_t GATHER_SPAN = 65536;
__global__ __launch_bounds__(256) void k_seq_gather(const SeqCopy *__restrict__ jobs, uint64_t *__restrict__ seqw,
                                                    uint32_t *__restrict__ nmw, uint32_t *__restrict__ has_n) {
    const SeqCopy j = jobs[blockIdx.x];
    const uint64_t w0 = (uint64_t)blockIdx.y * GATHER_SPAN;
    if (w0 >= j.nwords && !(blockIdx.y == 0)) return;
    if (blockIdx.y == 0 && threadIdx.x == 0) has_n[blockIdx.x] = *j.src_has_n;
    const uint64_t w1 = min(j.nwords, w0 + GATHER_SPAN);
    for (uint64_t w = w0 + threadIdx.x; w < w1; w += 256) {
        seqw[j.dst_off + w] = j.src_seqw[j.src_off + w];
        nmw[j.dst_off + w] = j.src_nmw[j.src_off + w];
    }
}
hipError_t launch_seq_gather(hipStream_t st, const SeqCopy *jobs, uint32_t n, uint64_t max_words, uint64_t *seqw, uint32_t *nmw,
                             uint32_t *has_n) {
    if (!n) return hipSuccess;
    const uint32_t gy = (uint32_t)std::max<uint64_t>(1, (max_words + GATHER_SPAN - 1) / GATHER_SPAN);
    if (gy > 65535u) return hipErrorInvalidValue;  // (a contig of more than 1.4 x 10^11 bases)
    hipLaunchKernelGGL(k_seq_gather, dim3(n, gy), dim3(256), 0, st, jobs, seqw, nmw, has_n);
    return hipGetLastError();
}

hipError_t launch_pack(hipStream_t st, const void *d_ascii, uint64_t len, uint64_t *seqw, uint32_t *nmw,
                       uint64_t nwords, uint32_t *has_n) {
    if (nwords == 0) return hipSuccess;
    uint64_t g = (nwords + 255) / 256;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)g), dim3(256), 0, st, (const uint8_t *)d_ascii, len, seqw, nmw,
                       nwords, has_n);
    return hipGetLastError();
}

// the header lines of a FASTA text on the device: every '>' at offset 0 or right behind a line feed (cpp/anchor.cpp:84: a line
// that starts with '>'), in no particular order — count[0] of them, the first `cap` in out[].  The host's memchr over the same
// bytes reads them at 20 GB/s, half of what a genome's load then costs; here they are read where they already are.
// (text: 16-byte aligned, readable up to the next multiple of 16 behind n)
__global__ __launch_bounds__(256) void k_text_headers(const uint8_t *__restrict__ text, uint64_t n, uint64_t *__restrict__ out, uint32_t cap,
                                                      uint32_t *__restrict__ count) {
    const uint64_t nv = (n + 15) / 16;
    for (uint64_t v = (uint64_t)blockIdx.x * 256u + threadIdx.x; v < nv; v += (uint64_t)gridDim.x * 256u) {
        const uint4 q = reinterpret_cast<const uint4 *>(text)[v];
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t x = w[i] ^ 0x3E3E3E3Eu;  // a zero byte where the text holds '>'
            if (((x - 0x01010101u) & ~x & 0x80808080u) == 0u) continue;
            for (int b = 0; b < 4; ++b) {
                if (((x >> (8 * b)) & 0xFFu) != 0u) continue;
                const uint64_t p = v * 16u + (uint64_t)(4 * i + b);
                if (p >= n || (p != 0 && text[p - 1] != '\n')) continue;
                const uint32_t at = atomicAdd(count, 1u);
                if (at < cap) out[at] = p;
            }
        }
    }
}

hipError_t launch_text_headers(hipStream_t st, const uint8_t *d_text, uint64_t nbytes, uint64_t *d_out, uint32_t cap, uint32_t *d_count) {
    hipError_t e = hipMemsetAsync(d_count, 0, 4, st);
    if (e != hipSuccess || nbytes == 0) return e;
    const uint64_t nv = (nbytes + 15) / 16;
    hipLaunchKernelGGL(k_text_headers, dim3((unsigned)std::min<uint64_t>((nv + 255) / 256, 8192)), dim3(256), 0, st, d_text, nbytes, d_out, cap, d_count);
    return hipGetLastError();
}

hipError_t launch_text_pack(hipStream_t st, const uint8_t *d_text, const TextChunk *d_chunks, uint64_t nchunks,
                            const uint64_t *d_rec_chunk0, uint32_t nrec, uint32_t *d_counts, uint64_t *d_base,
                            uint64_t *d_rec_len, const SeqDesc *sd, uint64_t *seqw, uint32_t *nmw, uint32_t *has_n) {
    if (nrec == 0) return hipSuccess;
    if (nchunks)
        hipLaunchKernelGGL(k_text_count, dim3((unsigned)nchunks), dim3(TEXT_THREADS), 0, st, d_text, d_chunks, d_counts);
    hipLaunchKernelGGL(k_text_scan, dim3(nrec), dim3(64), 0, st, d_counts, d_rec_chunk0, d_base, d_rec_len);
    if (nchunks)
        hipLaunchKernelGGL(k_text_pack, dim3((unsigned)nchunks), dim3(TEXT_THREADS), 0, st, d_text, d_chunks, d_base, sd,
                           seqw, nmw, has_n);
    return hipGetLastError();
}

hipError_t launch_sketch(hipStream_t st, int k, const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                         uint64_t nkmers, uint32_t *regs) {
    if (nkmers == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sketch, dim3(grid_for(nkmers, 256, 256 * 64)), dim3(256), 0, st, k, seqw, nmw, has_n, nkmers, regs);
    return hipGetLastError();
}

hipError_t launch_sketch_set(hipStream_t st, int k, const SeqDesc *sd, const uint2 *jobs, uint32_t njobs, const uint64_t *seqw,
                             const uint32_t *nmw, const uint32_t *has_n, uint32_t *regs) {
    if (njobs == 0) return hipSuccess;
    hipLaunchKernelGGL(k_sketch_set, dim3(njobs), dim3(256), 0, st, k, sd, jobs, seqw, nmw, has_n, regs);
    return hipGetLastError();
}

hipError_t launch_insert_seq(hipStream_t st, const SubTable &t, int w, uint32_t bits, int k,
                             const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                             uint64_t nkmers, unsigned long long *counters, uint32_t max_probe, int count_mode) {
    if (nkmers == 0) return hipSuccess;
    hipLaunchKernelGGL(k_insert_seq, dim3(grid_for(nkmers, 256, 256 * 64)), dim3(256), 0, st, t, w, bits, k,
                       seqw, nmw, has_n, nkmers, counters, max_probe, count_mode);
    return hipGetLastError();
}

hipError_t launch_insert_keys(hipStream_t st, const SubTable &t, int w, const uint64_t *keys,
                              const uint32_t *vals, uint64_t n, unsigned long long *counters,
                              uint32_t max_probe) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_insert_keys, dim3(grid_for(n, 256, 256 * 64)), dim3(256), 0, st, t, w, keys, vals, n,
                       counters, max_probe);
    return hipGetLastError();
}

hipError_t launch_import_kmc(hipStream_t st, const SubTable &t, int w, const uint8_t *rec, uint64_t first_record, uint64_t nrec,
                             const uint64_t *lut, uint64_t nlut, uint32_t prefixes_per_bin, uint32_t suffix_bytes,
                             uint32_t counter_bytes, uint32_t min_count, uint32_t max_count, unsigned long long *counters,
                             uint32_t max_probe, uint32_t phase, uint32_t dense_above) {
    if (nrec == 0) return hipSuccess;
    hipLaunchKernelGGL(k_import_kmc, dim3(grid_for(nrec, 256, 256 * 64)), dim3(256), 0, st, t, w, rec, first_record, nrec,
                       reinterpret_cast<const unsigned long long *>(lut), nlut, prefixes_per_bin, suffix_bytes, counter_bytes,
                       min_count, max_count, counters, max_probe, phase, dense_above);
    return hipGetLastError();
}

hipError_t launch_rehash(hipStream_t st, const SubTable &src, const SubTable &dst,
                         unsigned long long *counters, uint32_t max_probe, uint32_t ngenomes) {
    uint64_t nslots = src.nbuckets * src.slots;
    if (ngenomes < 4) {
        hipLaunchKernelGGL(k_rehash, dim3(grid_for(nslots, 256, 256 * 64)), dim3(256), 0, st, src, dst, counters, max_probe, 2u, 0u);
        return hipGetLastError();
    }
    for (uint32_t phase = 0; phase < 2; ++phase)  // two passes over the old table: the widely shared keys first
        hipLaunchKernelGGL(k_rehash, dim3(grid_for(nslots, 256, 256 * 64)), dim3(256), 0, st, src, dst, counters, max_probe, phase,
                           ngenomes / 2);
    return hipGetLastError();
}

hipError_t launch_merge_min(hipStream_t st, const SubTable &src, const SubTable &dst, int w, uint32_t bits,
                            uint32_t min_count, unsigned long long *counters, uint32_t max_probe) {
    uint64_t nslots = src.nbuckets * src.slots;
    hipLaunchKernelGGL(k_merge_min, dim3(grid_for(nslots, 256, 256 * 64)), dim3(256), 0, st, src, dst, w, bits,
                       min_count, counters, max_probe);
    return hipGetLastError();
}

// tile0[c] = first 512-position tile of contig c in a launch over all contigs, tile0[n] = their number: the exclusive
// prefix sum of ceil(nkmers(c) / tile) — one block, each thread a contiguous share of the contigs, the shares' totals
// scanned through LDS (a build launch's only per-call array: computed where it is used, no allocation or upload)
__global__ void __launch_bounds__(1024) k_tile0(const SeqDesc *__restrict__ sd, uint32_t n, uint32_t k, uint32_t tile, uint32_t *__restrict__ tile0) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (n + 1023) / 1024, lo = min(n, threadIdx.x * per), hi = min(n, lo + per);
    uint32_t mine = 0;
    for (uint32_t c = lo; c < hi; ++c) {
        const uint64_t len = sd[c].len;
        mine += len >= k ? (uint32_t)((len - k + 1 + tile - 1) / tile) : 0u;
    }
    part[threadIdx.x] = mine;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t at = part[threadIdx.x] - mine;
    for (uint32_t c = lo; c < hi; ++c) {
        tile0[c] = at;
        const uint64_t len = sd[c].len;
        at += len >= k ? (uint32_t)((len - k + 1 + tile - 1) / tile) : 0u;
    }
    if (threadIdx.x == 1023) tile0[n] = part[1023];
}

hipError_t preload_table_kernels() {
    hipFuncAttributes fa;
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_tile0));
}

hipError_t launch_tile0(hipStream_t st, const SeqDesc *sd, uint32_t n, uint32_t k, uint32_t tile, uint32_t *tile0) {
    hipLaunchKernelGGL(k_tile0, dim3(1), dim3(1024), 0, st, sd, n, k, tile, tile0);
    return hipGetLastError();
}

hipError_t launch_count_spill(hipStream_t st, const SubTable &t, unsigned long long *counters) {
    uint64_t nslots = t.nbuckets * t.slots;
    hipLaunchKernelGGL(k_count_spill, dim3(grid_for(nslots, 256, 256 * 64)), dim3(256), 0, st, t, counters);
    return hipGetLastError();
}

hipError_t launch_export(hipStream_t st, const SubTable &t, int w, uint64_t *keys, uint32_t *vals,
                         uint64_t cap, unsigned long long *count) {
    uint64_t nslots = t.nbuckets * t.slots;
    hipLaunchKernelGGL(k_export, dim3(grid_for(nslots, 256, 256 * 64)), dim3(256), 0, st, t, w, keys, vals, cap,
                       count);
    return hipGetLastError();
}

hipError_t launch_counters(hipStream_t st, const SubTable &t, int w, int k, const uint64_t *seqw,
                           const uint32_t *nmw, const uint32_t *has_n, uint64_t nkmers, uint32_t *out) {
    if (nkmers == 0) return hipSuccess;
    hipLaunchKernelGGL(k_counters, dim3(grid_for(nkmers, 256, 256 * 64)), dim3(256), 0, st, t, w, k, seqw, nmw,
                       has_n, nkmers, out);
    return hipGetLastError();
}

}  // namespace pg
