// pg_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the anchor path.
//
// Integer / HBM-bound work: no MFMA.  What bounds each kernel:
//   k_anchor       random 64-byte bucket gathers from HBM (1 per position per
//                  sub-table) + nbytes streamed out per position  -> HBM roofline
//   k_insert_seq   random 64-byte read-modify-write per k-mer      -> HBM / atomics
//   k_pack         1 byte in, 0.375 byte out per base, streaming   -> HBM roofline
//
// Replaces (reference, kjenike/panagram): KMC CKMCFile::GetCountersForRead as
// called from KMCdb::write_bits (cpp/anchor.cpp:112-195) and
// Genome._write_bitmap/_query_kmc_bytes/bin_bitsum (index.py:932-969,1169-1183).
#include "pg_kernels.h"

namespace pg {

// ---------------------------------------------------------------------------
// table init: every thread writes one 16-byte chunk of a bucket
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_table_init(uint4 *chunks, uint64_t nchunks, uint32_t W) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < nchunks; i += stride) {
        chunks[i] = make_uint4(~0u, ~0u, 0u, 0u);  // {EMPTY key, mask0 = 0, mask1 = 0}
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
// k-mer set construction: one thread per k-mer position, insert-or-OR.
// counters[0] += newly claimed keys; counters[1] = overflow flag.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_insert_seq(SubTable st, int w, uint32_t bits, int k,
                                                    const uint64_t *seqw, const uint32_t *nmw,
                                                    const uint32_t *has_n, uint64_t nkmers,
                                                    unsigned long long *counters, uint32_t max_probe) {
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const bool hasn = (*has_n != 0);
    uint32_t claimed = 0;
    for (; p < nkmers; p += stride) {
        if (hasn && extract_nmask(nmw, p, k)) continue;
        uint64_t key = canonical_from_le(extract_bases(seqw, p), k);
        int r = lane_insert(st, key, w, bits, max_probe);
        if (r < 0) atomicOr(reinterpret_cast<unsigned int *>(&counters[1]), 1u);
        else claimed += r;
    }
    if (claimed) atomicAdd(&counters[0], (unsigned long long)claimed);
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
        int r = lane_insert(st, keys[i], w, v, max_probe);
        if (r < 0) atomicOr(reinterpret_cast<unsigned int *>(&counters[1]), 1u);
        else claimed += r;
    }
    if (claimed) atomicAdd(&counters[0], (unsigned long long)claimed);
}

// re-hash every occupied slot of `src` into `dst` (same W)
__global__ __launch_bounds__(256) void k_rehash(SubTable src, SubTable dst, unsigned long long *counters,
                                                uint32_t max_probe) {
    const int ns = slots_per_bucket(src.W);
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t nslots = src.nbuckets * ns;
    uint32_t claimed = 0;
    for (; i < nslots; i += stride) {
        uint64_t b = i / ns;
        int s = (int)(i - b * ns);
        const uint8_t *base = src.buckets + b * BUCKET_BYTES;
        uint64_t key = *reinterpret_cast<const uint64_t *>(base + key_off(src.W, s));
        if (key == EMPTY_KEY) continue;
        for (uint32_t w = 0; w < src.W; ++w) {
            uint32_t m = *reinterpret_cast<const uint32_t *>(base + mask_off(src.W, s, w));
            if (m == 0 && w > 0) continue;
            int r = lane_insert(dst, key, (int)w, m, max_probe);
            if (r < 0) atomicOr(reinterpret_cast<unsigned int *>(&counters[1]), 1u);
            else claimed += r;
        }
    }
    if (claimed) atomicAdd(&counters[0], (unsigned long long)claimed);
}

// export (key, mask word w) of every slot whose word w is non-zero
__global__ __launch_bounds__(256) void k_export(SubTable st, int w, uint64_t *keys, uint32_t *vals,
                                                uint64_t cap, unsigned long long *count) {
    const int ns = slots_per_bucket(st.W);
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t nslots = st.nbuckets * ns;
    for (; i < nslots; i += stride) {
        uint64_t b = i / ns;
        int s = (int)(i - b * ns);
        const uint8_t *base = st.buckets + b * BUCKET_BYTES;
        uint64_t key = *reinterpret_cast<const uint64_t *>(base + key_off(st.W, s));
        if (key == EMPTY_KEY) continue;
        uint32_t m = *reinterpret_cast<const uint32_t *>(base + mask_off(st.W, s, w));
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
            uint32_t m0, m1;
            if (lane_lookup(st, key, m0, m1)) r = w ? m1 : m0;
        }
        out[p] = r;
    }
}

// ---------------------------------------------------------------------------
// THE hot kernel.  One workgroup (256 threads = 4 waves = 64 quads) = one tile
// of TILE consecutive k-mer positions of one contig.
//   phase 0  packed bases of the tile (+k-1 halo) -> LDS (coalesced, 0.25 B/pos)
//   phase 1  canonical k-mer + home bucket of every position -> LDS (hashed ONCE)
//   phase 2  LDS-staged probe batches: a quad fetches ONE 64-byte bucket with
//            four 16-byte lanes (a wave instruction = 16 whole buckets, every
//            fetched byte used), UNROLL independent gathers in flight per lane,
//            lane-local slot match + DPP quad-OR; the rare "bucket full, key
//            absent" case goes to an LDS retry queue resolved after the batch
//   phase 3  popcount -> wave-ballot aggregated LDS histogram, column sums by
//            ballot, rows packed into an LDS byte tile, 1-in-100 rows out,
//            then coalesced 16-byte stores of the bitmap.1 tile
// NDBS_C / NBYTES_C: compile-time row shape (0 = runtime, generic path).
// ---------------------------------------------------------------------------
// ---- per-position epilogue pieces shared by k_anchor (fused) and k_rows_epilogue ----
// column sums: one ballot + popcount per genome bit, accumulated in LDS by lane 0
__device__ __forceinline__ void colsum_word(uint32_t wv, uint32_t d, uint32_t N, uint32_t *cs, int lane) {
    const uint32_t ng = min(32u, N - 32 * d);
    for (uint32_t bit = 0; bit < ng; ++bit) {
        const unsigned long long bal = __ballot((wv >> bit) & 1u);
        if (lane == 0 && bal) atomicAdd(&cs[32 * d + bit], (uint32_t)__popcll(bal));
    }
}
// wave-aggregated histogram of (bin, popcount): LDS for the tile's first two bins, global beyond
template <int TILE>
__device__ __forceinline__ void hist_position(bool active, uint32_t pos, uint32_t popc, uint32_t N,
                                              uint32_t binlen, uint32_t bin0, uint32_t bin0_start,
                                              uint32_t *hist, uint32_t *bins, uint64_t bin_off, int lane) {
    if (popc > N) popc = N;  // junk bits beyond ngenomes: the reference indexes out of bounds here
    const uint32_t dpos = pos - bin0_start;
    const uint32_t rel = (binlen >= (uint32_t)TILE) ? (dpos >= binlen ? 1u : 0u) : dpos / binlen;
    const uint32_t hk = rel * (N + 1) + popc;
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lk = __shfl(hk, leader);
        const unsigned long long m = __ballot(active && hk == lk) & todo;
        if (lane == leader) {
            const uint32_t cnt = (uint32_t)__popcll(m);
            if (rel < 2) atomicAdd(&hist[hk], cnt);
            else atomicAdd(&bins[(bin_off + bin0 + rel) * (uint64_t)(N + 1) + popc], cnt);
        }
        todo &= ~m;
    }
}
__device__ __forceinline__ void flush_stats(uint32_t N, const uint32_t *hist, const uint32_t *cs, uint32_t *bins,
                                            unsigned long long *colsums, uint64_t bin_off, uint32_t bin0,
                                            bool want_cs, int tid) {
    for (uint32_t i = tid; i < 2 * (N + 1); i += ANCHOR_WG) {
        const uint32_t hv = hist[i];
        if (hv) {
            const uint32_t rel = i / (N + 1), pc = i - rel * (N + 1);
            atomicAdd(&bins[(bin_off + bin0 + rel) * (uint64_t)(N + 1) + pc], hv);
        }
    }
    if (want_cs) {
        for (uint32_t i = tid; i < N; i += ANCHOR_WG) {
            const uint32_t v = cs[i];
            if (v) atomicAdd(&colsums[i], (unsigned long long)v);
        }
    }
}

template <int TILE>
struct Geo {
    static constexpr int NQUAD = ANCHOR_WG / 4;
    static constexpr int ROUNDS = TILE / NQUAD;
    static constexpr int PER_THREAD = TILE / ANCHOR_WG;
    static constexpr int SEQW = (TILE + 31) / 32 + 3 + ((4 - ((TILE + 31) / 32 + 3) % 4) % 4);  // 16-byte padded
};

__device__ __forceinline__ void quad_chase(const SubTable &st, uint64_t key, uint64_t b, int j,
                                           uint32_t &m0, uint32_t &m1) {
    // follow the probe chain from bucket b until the key or a non-full bucket
    for (uint64_t n = 0; n < st.nbuckets; ++n) {
        uint4 v = *reinterpret_cast<const uint4 *>(st.buckets + b * BUCKET_BYTES + j * 16);
        if (quad_match(v, key, m0, m1)) return;
        if (!quad_full(v)) break;
        b = (b + 1 == st.nbuckets) ? 0 : b + 1;
    }
    m0 = m1 = 0;
}

template <int TILE, int UNROLL>
__device__ __forceinline__ void probe_sub(const SubTable st, const uint64_t *keys, const uint32_t *bkt,
                                          uint32_t *res, uint32_t ndbs, uint32_t *rq_cnt, uint32_t *rq_pi,
                                          uint32_t *rq_b, int tid) {
    using G = Geo<TILE>;
    const int q = tid >> 2, j = tid & 3;
    const uint8_t *lane_base = st.buckets + j * 16;
    const bool two = (st.W == 2);
    for (int r0 = 0; r0 < G::ROUNDS; r0 += UNROLL) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            // UNCONDITIONAL load (invalid positions carry bucket 0): a branch here makes the
            // compiler wait for each gather before issuing the next one
            const int pi = (r0 + u) * G::NQUAD + q;
            v[u] = *reinterpret_cast<const uint4 *>(lane_base + (uint64_t)bkt[pi] * BUCKET_BYTES);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int pi = (r0 + u) * G::NQUAD + q;
            const uint64_t key = keys[pi];  // re-read from LDS: cheaper than holding 2*UNROLL VGPRs
            uint32_t m0, m1;  // an invalid position (key == EMPTY) can only "match" an EMPTY slot: masks 0
            const bool found = quad_match(v[u], key, m0, m1);
            if (!found && key != EMPTY_KEY) {  // rare: absent key, or the key overflowed its bucket
                if (quad_full(v[u])) {
                    uint64_t nb = (uint64_t)bkt[pi] + 1;
                    if (nb == st.nbuckets) nb = 0;
                    uint32_t idx = 0;
                    if (j == 0) idx = atomicAdd(rq_cnt, 1u);
                    idx = quad_perm<QP_BC0>(idx);
                    if (idx < (uint32_t)ANCHOR_RQ) {
                        if (j == 0) {
                            rq_pi[idx] = (uint32_t)pi;
                            rq_b[idx] = (uint32_t)nb;
                        }
                    } else {
                        quad_chase(st, key, nb, j, m0, m1);  // queue full: resolve inline
                    }
                }
            }
            if (j == 0) {
                res[pi * ndbs + st.word0] = m0;
                if (two) res[pi * ndbs + st.word0 + 1] = m1;
            }
        }
    }
    __syncthreads();
    uint32_t n = *rq_cnt;
    if (n > (uint32_t)ANCHOR_RQ) n = ANCHOR_RQ;
    for (uint32_t i = q; i < n; i += G::NQUAD) {
        const uint32_t pi = rq_pi[i];
        uint32_t m0, m1;
        quad_chase(st, keys[pi], rq_b[i], j, m0, m1);
        if (j == 0) {
            res[pi * ndbs + st.word0] = m0;
            if (two) res[pi * ndbs + st.word0 + 1] = m1;
        }
    }
    __syncthreads();
}

template <int TILE, int UNROLL, int NDBS_C, int NBYTES_C>
__global__ __launch_bounds__(ANCHOR_WG) void k_anchor(const TableDesc T, const uint64_t *__restrict__ seqw,
                                                      const uint32_t *__restrict__ nmw,
                                                      const uint32_t *__restrict__ has_n,
                                                      const SeqDesc *__restrict__ sd,
                                                      const AnchorDesc *__restrict__ ad,
                                                      const uint32_t *__restrict__ tile_contig,
                                                      uint8_t *__restrict__ out1, uint8_t *__restrict__ out100,
                                                      uint32_t *__restrict__ bins,
                                                      unsigned long long *__restrict__ colsums, uint32_t flags) {
    using G = Geo<TILE>;
    extern __shared__ uint4 smem[];
    const int tid = threadIdx.x;
    const uint32_t N = T.ngenomes, k = T.k;
    const uint32_t ndbs = NDBS_C ? (uint32_t)NDBS_C : T.ndbs;
    const uint32_t nbytes = NBYTES_C ? (uint32_t)NBYTES_C : (N + 7) / 8;
    // ---- LDS carve-up (all offsets multiples of 16 bytes) ----
    uint8_t *sp = reinterpret_cast<uint8_t *>(smem);
    const uint32_t rows_bytes = TILE * (nbytes > 8 ? nbytes : 8);
    uint64_t *keys = reinterpret_cast<uint64_t *>(sp);          // phase 1-2
    uint8_t *rows = sp;                                         // phase 3 (aliases keys)
    sp += rows_bytes;
    uint32_t *bkt = reinterpret_cast<uint32_t *>(sp);
    sp += TILE * 4;
    uint32_t *res = reinterpret_cast<uint32_t *>(sp);
    sp += TILE * ndbs * 4;
    uint64_t *sw = reinterpret_cast<uint64_t *>(sp);
    sp += G::SEQW * 8;
    uint32_t *nw = reinterpret_cast<uint32_t *>(sp);
    sp += G::SEQW * 4;
    uint32_t *rq_pi = reinterpret_cast<uint32_t *>(sp);
    sp += ANCHOR_RQ * 4;
    uint32_t *rq_b = reinterpret_cast<uint32_t *>(sp);
    sp += ANCHOR_RQ * 4;
    uint32_t *hist = reinterpret_cast<uint32_t *>(sp);
    sp += ((2 * (N + 1) + 3) & ~3u) * 4;
    uint32_t *cs = reinterpret_cast<uint32_t *>(sp);
    sp += ((N + 3) & ~3u) * 4;
    uint32_t *rq_cnt = reinterpret_cast<uint32_t *>(sp);

    const uint32_t c = tile_contig[blockIdx.x];
    const AnchorDesc a = ad[c];
    const SeqDesc s = sd[c];
    const uint32_t tile_start = (blockIdx.x - a.tile0) * TILE;
    const uint32_t npos = min((uint32_t)TILE, a.nkmers - tile_start);
    const bool hasn = has_n[c] != 0;

    // ---- phase 0 ----
    if (tid < G::SEQW) {
        uint64_t wi = (uint64_t)(tile_start >> 5) + tid;
        sw[tid] = wi < s.nwords ? seqw[s.seq_off + wi] : 0ull;
        nw[tid] = (hasn && wi < s.nwords) ? nmw[s.seq_off + wi] : 0u;
    }
    for (uint32_t i = tid; i < 2 * (N + 1); i += ANCHOR_WG) hist[i] = 0;
    for (uint32_t i = tid; i < N; i += ANCHOR_WG) cs[i] = 0;
    if (tid == 0) *rq_cnt = 0;
    __syncthreads();

    // ---- phase 1 + 2 (bucket indices depend on the sub-table's size) ----
    for (uint32_t si = 0; si < T.nsub; ++si) {
        const SubTable st = T.sub[si];
#pragma unroll 2
        for (int jj = 0; jj < G::PER_THREAD; ++jj) {
            const uint32_t pl = jj * ANCHOR_WG + tid;
            uint64_t key;
            if (si == 0) {
                key = EMPTY_KEY;
                if (pl < npos) {
                    key = canonical_from_le(extract_bases(sw, pl), (int)k);
                    if (hasn && extract_nmask(nw, pl, (int)k)) key = EMPTY_KEY;
                }
                keys[pl] = key;
            } else {
                key = keys[pl];
            }
            bkt[pl] = key == EMPTY_KEY ? 0u : (uint32_t)home_bucket(key, st.nbuckets);
        }
        if (si && tid == 0) *rq_cnt = 0;
        __syncthreads();
        probe_sub<TILE, UNROLL>(st, keys, bkt, res, ndbs, rq_cnt, rq_pi, rq_b, tid);
    }

    // ---- phase 3 ----
    const uint32_t binlen = a.binlen;
    const uint32_t bin0 = tile_start / binlen;
    const uint32_t bin0_start = bin0 * binlen;
    const int lane = tid & 63;
    const bool want_cs = (flags & 1u) != 0;
    const bool stats = (flags & 2u) == 0;  // rows-only mode leaves the statistics to k_rows_epilogue
#pragma unroll 1
    for (int jj = 0; jj < G::PER_THREAD; ++jj) {
        const uint32_t pl = jj * ANCHOR_WG + tid;
        const bool active = pl < npos;
        const uint32_t pos = tile_start + pl;
        uint32_t popc = 0;
        const bool is100 = stats && active && (pos % 100u == 0);
#pragma unroll
        for (uint32_t d = 0; d < (NDBS_C ? (uint32_t)NDBS_C : ndbs); ++d) {
            const uint32_t wv = active ? res[pl * ndbs + d] : 0u;
            popc += __popc(wv);
            // low n bytes of this group's u32 at columns 4d.. (cpp/anchor.cpp:139-164)
            const uint32_t nb = min(4u, nbytes - 4 * d);
            if (NBYTES_C == 1) {
                rows[pl] = (uint8_t)wv;
            } else if (NBYTES_C == 2) {
                reinterpret_cast<uint16_t *>(rows)[pl] = (uint16_t)wv;
            } else if (NBYTES_C == 4 || NBYTES_C == 8) {
                reinterpret_cast<uint32_t *>(rows)[pl * (NBYTES_C / 4) + d] = wv;
            } else {
                for (uint32_t bb = 0; bb < nb; ++bb) rows[pl * nbytes + 4 * d + bb] = (uint8_t)(wv >> (8 * bb));
            }
            if (is100) {
                uint8_t *o100 = out100 + a.out100_off + (uint64_t)(pos / 100u) * nbytes + 4 * d;
                for (uint32_t bb = 0; bb < nb; ++bb) o100[bb] = (uint8_t)(wv >> (8 * bb));
            }
            if (stats && want_cs) colsum_word(wv, d, N, cs, lane);
        }
        if (stats) hist_position<TILE>(active, pos, popc, N, binlen, bin0, bin0_start, hist, bins, a.bin_off, lane);
    }
    __syncthreads();
    if (stats) flush_stats(N, hist, cs, bins, colsums, a.bin_off, bin0, want_cs, tid);
    // bitmap.1 tile: coalesced 16-byte stores (tile base is 16-byte aligned)
    {
        uint8_t *g = out1 + a.out_off + (uint64_t)tile_start * nbytes;
        const uint32_t total = npos * nbytes;
        const uint32_t nvec = total >> 4;
        const uint4 *src = reinterpret_cast<const uint4 *>(rows);
        uint4 *dst = reinterpret_cast<uint4 *>(g);
        for (uint32_t i = tid; i < nvec; i += ANCHOR_WG) dst[i] = src[i];
        for (uint32_t i = (nvec << 4) + tid; i < total; i += ANCHOR_WG) g[i] = rows[i];
    }
}

// ---------------------------------------------------------------------------
// Statistics pass over FINISHED rows (genome-sharded mode: every GPU anchors all positions
// against its own genomes' table, the partial rows are combined over xGMI, then each row's
// popcount histogram / column sums / 1-in-100 rows are taken from the combined bytes).
// Streaming: nbytes read per position.
// ---------------------------------------------------------------------------
template <int TILE>
__global__ __launch_bounds__(ANCHOR_WG) void k_rows_epilogue(uint32_t N, const AnchorDesc *__restrict__ ad,
                                                             const uint32_t *__restrict__ tile_contig,
                                                             const uint8_t *__restrict__ out1,
                                                             uint8_t *__restrict__ out100,
                                                             uint32_t *__restrict__ bins,
                                                             unsigned long long *__restrict__ colsums,
                                                             uint32_t flags) {
    extern __shared__ uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t nbytes = (N + 7) / 8, ndbs = (N + 31) / 32;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    uint32_t *cs = hist + ((2 * (N + 1) + 3) & ~3u);
    const uint32_t c = tile_contig[blockIdx.x];
    const AnchorDesc a = ad[c];
    const uint32_t tile_start = (blockIdx.x - a.tile0) * TILE;
    const uint32_t npos = min((uint32_t)TILE, a.nkmers - tile_start);
    for (uint32_t i = tid; i < 2 * (N + 1); i += ANCHOR_WG) hist[i] = 0;
    for (uint32_t i = tid; i < N; i += ANCHOR_WG) cs[i] = 0;
    __syncthreads();
    const uint32_t binlen = a.binlen, bin0 = tile_start / binlen, bin0_start = bin0 * binlen;
    const bool want_cs = (flags & 1u) != 0;
    const uint8_t *g = out1 + a.out_off + (uint64_t)tile_start * nbytes;
    for (uint32_t pl = tid; pl < (uint32_t)TILE; pl += ANCHOR_WG) {
        const bool active = pl < npos;
        const uint32_t pos = tile_start + pl;
        uint32_t popc = 0;
        const bool is100 = active && (pos % 100u == 0);
        for (uint32_t d = 0; d < ndbs; ++d) {
            const uint32_t nb = min(4u, nbytes - 4 * d);
            uint32_t wv = 0;
            if (active)
                for (uint32_t bb = 0; bb < nb; ++bb) wv |= (uint32_t)g[(uint64_t)pl * nbytes + 4 * d + bb] << (8 * bb);
            popc += __popc(wv);
            if (is100) {
                uint8_t *o100 = out100 + a.out100_off + (uint64_t)(pos / 100u) * nbytes + 4 * d;
                for (uint32_t bb = 0; bb < nb; ++bb) o100[bb] = (uint8_t)(wv >> (8 * bb));
            }
            if (want_cs) colsum_word(wv, d, N, cs, lane);
        }
        hist_position<TILE>(active, pos, popc, N, binlen, bin0, bin0_start, hist, bins, a.bin_off, lane);
    }
    __syncthreads();
    flush_stats(N, hist, cs, bins, colsums, a.bin_off, bin0, want_cs, tid);
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
    uint64_t nchunks = t.nbuckets * 4;
    hipLaunchKernelGGL(k_table_init, dim3(grid_for(nchunks, 256, 256 * 32)), dim3(256), 0, st,
                       reinterpret_cast<uint4 *>(t.buckets), nchunks, t.W);
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

hipError_t launch_insert_seq(hipStream_t st, const SubTable &t, int w, uint32_t bits, int k,
                             const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                             uint64_t nkmers, unsigned long long *counters, uint32_t max_probe) {
    if (nkmers == 0) return hipSuccess;
    hipLaunchKernelGGL(k_insert_seq, dim3(grid_for(nkmers, 256, 256 * 64)), dim3(256), 0, st, t, w, bits, k,
                       seqw, nmw, has_n, nkmers, counters, max_probe);
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

hipError_t launch_rehash(hipStream_t st, const SubTable &src, const SubTable &dst,
                         unsigned long long *counters, uint32_t max_probe) {
    uint64_t nslots = src.nbuckets * slots_per_bucket(src.W);
    hipLaunchKernelGGL(k_rehash, dim3(grid_for(nslots, 256, 256 * 64)), dim3(256), 0, st, src, dst, counters,
                       max_probe);
    return hipGetLastError();
}

hipError_t launch_export(hipStream_t st, const SubTable &t, int w, uint64_t *keys, uint32_t *vals,
                         uint64_t cap, unsigned long long *count) {
    uint64_t nslots = t.nbuckets * slots_per_bucket(t.W);
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

size_t anchor_lds_bytes(uint32_t ngenomes) {
    using G = Geo<ANCHOR_TILE>;
    const uint32_t nbytes = (ngenomes + 7) / 8, ndbs = (ngenomes + 31) / 32;
    size_t b = (size_t)ANCHOR_TILE * (nbytes > 8 ? nbytes : 8);
    b += (size_t)ANCHOR_TILE * 4;
    b += (size_t)ANCHOR_TILE * ndbs * 4;
    b += G::SEQW * 8 + G::SEQW * 4;
    b += 2 * ANCHOR_RQ * 4;
    b += ((2 * (ngenomes + 1) + 3) & ~3u) * 4;
    b += ((ngenomes + 3) & ~3u) * 4;
    b += 16;
    return b;
}

template <int NDBS_C, int NBYTES_C>
static hipError_t launch_anchor_t(hipStream_t st, size_t lds, uint32_t ntiles, const TableDesc &T,
                                  const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                                  const SeqDesc *sd, const AnchorDesc *ad, const uint32_t *tile_contig,
                                  uint8_t *out1, uint8_t *out100, uint32_t *bins, unsigned long long *colsums,
                                  uint32_t flags) {
    auto kern = k_anchor<ANCHOR_TILE, ANCHOR_UNROLL, NDBS_C, NBYTES_C>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(ANCHOR_WG), lds, st, T, seqw, nmw, has_n, sd, ad, tile_contig,
                       out1, out100, bins, colsums, flags);
    return hipGetLastError();
}

hipError_t launch_anchor(hipStream_t st, const TableDesc &T, const uint64_t *seqw, const uint32_t *nmw,
                         const uint32_t *has_n, const SeqDesc *sd, const AnchorDesc *ad,
                         const uint32_t *tile_contig, uint32_t ntiles, uint8_t *out1, uint8_t *out100,
                         uint32_t *bins, unsigned long long *colsums, uint32_t flags) {
    if (ntiles == 0) return hipSuccess;
    const size_t lds = anchor_lds_bytes(T.ngenomes);
    const uint32_t nbytes = (T.ngenomes + 7) / 8;
#define PG_ARGS st, lds, ntiles, T, seqw, nmw, has_n, sd, ad, tile_contig, out1, out100, bins, colsums, flags
    if (T.ndbs == 1 && nbytes == 1) return launch_anchor_t<1, 1>(PG_ARGS);
    if (T.ndbs == 1 && nbytes == 2) return launch_anchor_t<1, 2>(PG_ARGS);
    if (T.ndbs == 1 && nbytes == 4) return launch_anchor_t<1, 4>(PG_ARGS);
    if (T.ndbs == 2 && nbytes == 8) return launch_anchor_t<2, 8>(PG_ARGS);
    return launch_anchor_t<0, 0>(PG_ARGS);
#undef PG_ARGS
}

hipError_t launch_rows_epilogue(hipStream_t st, uint32_t ngenomes, const AnchorDesc *ad, const uint32_t *tile_contig,
                                uint32_t ntiles, const uint8_t *out1, uint8_t *out100, uint32_t *bins,
                                unsigned long long *colsums, uint32_t flags) {
    if (ntiles == 0) return hipSuccess;
    size_t lds = (((2 * (ngenomes + 1) + 3) & ~3u) + ((ngenomes + 3) & ~3u)) * 4 + 16;
    hipLaunchKernelGGL(k_rows_epilogue<ANCHOR_TILE>, dim3(ntiles), dim3(ANCHOR_WG), lds, st, ngenomes, ad,
                       tile_contig, out1, out100, bins, colsums, flags);
    return hipGetLastError();
}

}  // namespace pg
