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
// THE hot kernel.  One workgroup = one tile of TILE consecutive k-mer positions of one
// contig; thread t owns the PT = TILE/WG consecutive positions [t*PT, (t+1)*PT).
//   phase 0  packed bases of the tile (+ halo) -> LDS (coalesced, 0.25 B/pos)
//   phase 1  per thread, in registers: roll the forward window X and the reverse-complement
//            window B base by base, canonical key = ~max(X,B), slide the minimizer window,
//            home line = hash(minimizer)   — consecutive positions mostly share a line
//   phase 2  LDS-STAGED PROBE BATCH: positions are grouped into runs of equal home line
//            (prefix scan of "line changed" flags); every distinct line of the tile is
//            fetched ONCE, cooperatively and coalesced (8 lanes x 16 B = one 128-B line,
//            all fetches of the tile in flight together), into an LDS line buffer; then
//            each lane scans the 8 slots of ITS positions' lines out of LDS (no cross-lane
//            traffic).  A position whose key is absent from a full line joins an LDS
//            overflow queue; queue rounds stage line+1, line+2, ... the same way.
//   phase 3  from registers: rows packed per thread (PT consecutive positions -> wide
//            coalesced stores), 1-in-100 rows, popcount histogram (wave ballots -> LDS ->
//            global), column sums (ballots)
// W_C = minimizer window (0: hash the k-mer itself); NDBS_C/NBYTES_C: compile-time row
// shape (0 = runtime, generic path).
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

// A staged line occupies 144 bytes of LDS (128 + 16 pad): with a 128-byte stride every lane's
// ds_read_b128 of "its" line would land on one of two bank groups (32-way conflict); 144 = 4*36
// bytes walks all 16 four-bank groups over 16 consecutive lines.
constexpr int LDS_LINE_U4 = 9;

template <int TILE>
struct Geo {
    static constexpr int PT = TILE / ANCHOR_WG;  // consecutive positions per thread
    static constexpr int NWAVE = ANCHOR_WG / 64;
    // packed words staged per tile: the last thread reads 64 bases past its first position
    static constexpr int SEQW_RAW = (TILE + 31) / 32 + 4;
    static constexpr int SEQW = (SEQW_RAW + 3) & ~3;  // 16-byte padded
    static constexpr int LCAP = ANCHOR_LINES;         // LDS line buffer capacity
    static constexpr int QCAP = TILE / 2;             // overflow queue entries
};

// scan the 8 slots of a line staged in LDS.  Lines fill front to back without holes (an
// insert claims the first EMPTY slot and slots never revert), so "full" == last slot used.
// returns 1 = found, 0 = absent (line not full), -1 = absent from a full line
template <bool TWO>
__device__ __forceinline__ int scan_line(const uint4 *line, uint64_t key, uint32_t &m0, uint32_t &m1) {
    m0 = m1 = 0;
    uint64_t last = 0;
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const uint4 v = line[sl];
        const uint64_t kk = (uint64_t)v.x | ((uint64_t)v.y << 32);
        const bool hit = (kk == key);
        m0 = hit ? v.z : m0;
        if (TWO) m1 = hit ? v.w : m1;
        if (sl == SLOTS - 1) last = kk;
    }
    return (m0 | m1) ? 1 : (last == EMPTY_KEY ? 0 : -1);
}

// single-lane chase through global memory from line b (slow path: chains beyond the queue rounds);
// the 8 slot loads of a line are issued together, so one memory latency per line
__device__ __forceinline__ void lane_chase(const SubTable &st, uint64_t key, uint32_t b, uint32_t step, uint32_t &m0,
                                           uint32_t &m1) {
    m0 = m1 = 0;
    for (uint64_t n = 0; n < st.nbuckets; ++n) {
        const uint4 *line = reinterpret_cast<const uint4 *>(st.buckets + (uint64_t)b * BUCKET_BYTES);
        uint4 v[SLOTS];
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) v[sl] = line[sl];
        uint64_t last = 0;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            const uint64_t kk = (uint64_t)v[sl].x | ((uint64_t)v[sl].y << 32);
            if (kk == key) {
                m0 = v[sl].z;
                m1 = v[sl].w;
            }
            last = kk;
        }
        if ((m0 | m1) != 0 || last == EMPTY_KEY) return;
        b = next_line(b, step, st.nbuckets);
    }
}

// cooperative, coalesced staging of `nl` lines (indices in lines[]) into the LDS buffer:
// 8 consecutive lanes move one 128-byte line; STAGE_U loads per lane in flight
template <int STAGE_U>
__device__ __forceinline__ void stage_lines(const uint8_t *table, const uint32_t *lines, uint32_t nl, uint4 *buf,
                                            int tid) {
    const uint32_t total = nl * 8;
    for (uint32_t i0 = 0; i0 < total; i0 += ANCHOR_WG * STAGE_U) {
        uint4 v[STAGE_U];
#pragma unroll
        for (int u = 0; u < STAGE_U; ++u) {
            uint32_t idx = i0 + u * ANCHOR_WG + tid;
            idx = idx < total ? idx : total - 1;  // clamp: unconditional loads stay back to back
            v[u] = *reinterpret_cast<const uint4 *>(table + (uint64_t)lines[idx >> 3] * BUCKET_BYTES + (idx & 7) * 16);
        }
#pragma unroll
        for (int u = 0; u < STAGE_U; ++u) {
            const uint32_t idx = i0 + u * ANCHOR_WG + tid;
            if (idx < total) buf[(idx >> 3) * LDS_LINE_U4 + (idx & 7)] = v[u];  // padded stride: see LDS_LINE_U4
        }
    }
}

template <int TILE, int W_C, int NDBS_C, int NBYTES_C>
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
    constexpr int PT = G::PT;
    constexpr int MAXD = NDBS_C ? NDBS_C : 2 * MAX_SUB;  // generic shape: dynamic word index (slow path, N > 64)
    extern __shared__ uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t N = T.ngenomes, k = T.k;
    const uint32_t ndbs = NDBS_C ? (uint32_t)NDBS_C : T.ndbs;
    const uint32_t nbytes = NBYTES_C ? (uint32_t)NBYTES_C : (N + 7) / 8;
    // ---- LDS carve-up (all offsets multiples of 16 bytes) ----
    uint8_t *sp = reinterpret_cast<uint8_t *>(smem);
    uint4 *buf = reinterpret_cast<uint4 *>(sp);                 // staged table lines
    sp += (size_t)G::LCAP * LDS_LINE_U4 * 16;
    uint64_t *q_key = reinterpret_cast<uint64_t *>(sp);         // overflow queue: key, later {m0,m1}
    sp += G::QCAP * 8;
    uint32_t *q_line = reinterpret_cast<uint32_t *>(sp);
    sp += G::QCAP * 4;
    uint32_t *q_step = reinterpret_cast<uint32_t *>(sp);
    sp += G::QCAP * 4;
    uint32_t *lines = reinterpret_cast<uint32_t *>(sp);         // line index per buffer slot
    sp += G::LCAP * 4;
    uint16_t *wl0 = reinterpret_cast<uint16_t *>(sp);           // worklists of pending queue entries
    sp += G::QCAP * 2;
    uint16_t *wl1 = reinterpret_cast<uint16_t *>(sp);
    sp += G::QCAP * 2;
    uint64_t *sw = reinterpret_cast<uint64_t *>(sp);
    sp += G::SEQW * 8;
    uint32_t *nw = reinterpret_cast<uint32_t *>(sp);
    sp += G::SEQW * 4;
    uint32_t *lastline = reinterpret_cast<uint32_t *>(sp);      // last home line of every thread
    sp += ANCHOR_WG * 4;
    uint32_t *hist = reinterpret_cast<uint32_t *>(sp);
    sp += ((2 * (N + 1) + 3) & ~3u) * 4;
    uint32_t *cs = reinterpret_cast<uint32_t *>(sp);
    sp += ((N + 3) & ~3u) * 4;
    uint32_t *ctl = reinterpret_cast<uint32_t *>(sp);           // [0] q_cnt [1],[2] worklist counts [4..] wave totals

    const uint32_t c = tile_contig[blockIdx.x];
    const AnchorDesc a = ad[c];
    const SeqDesc s = sd[c];
    const uint32_t tile_start = (blockIdx.x - a.tile0) * TILE;
    const uint32_t npos = min((uint32_t)TILE, a.nkmers - tile_start);
    const bool hasn = has_n[c] != 0;

    // ---- phase 0 ----
    for (uint32_t i = tid; i < (uint32_t)G::SEQW; i += ANCHOR_WG) {
        uint64_t wi = (uint64_t)(tile_start >> 5) + i;
        sw[i] = wi < s.nwords ? seqw[s.seq_off + wi] : 0ull;
        nw[i] = (hasn && wi < s.nwords) ? nmw[s.seq_off + wi] : 0u;
    }
    for (uint32_t i = tid; i < 2 * (N + 1); i += ANCHOR_WG) hist[i] = 0;
    for (uint32_t i = tid; i < N; i += ANCHOR_WG) cs[i] = 0;
    __syncthreads();

    // ---- phase 1: keys + home lines of this thread's PT positions, in registers ----
    const uint32_t p0 = tid * PT;
    uint64_t key[PT];
    uint32_t line[PT], grp[PT];  // home line and group id (minimizer) of every position
    uint32_t words[PT][MAXD];
#pragma unroll
    for (int jj = 0; jj < PT; ++jj)
#pragma unroll
        for (int d = 0; d < MAXD; ++d) words[jj][d] = 0;
    {
        const int kk = (int)k;
        const uint64_t kmask = kmer_mask(kk);
        const uint64_t A = extract_bases(sw, p0), E = extract_bases(sw, p0 + 32);
        const uint64_t NM = hasn ? extract_nmask64(nw, p0) : 0ull;
        const uint32_t kbits = (kk == 32) ? 0xFFFFFFFFu : ((1u << kk) - 1);
        uint64_t X = A & kmask;
        uint64_t B = revcomp_le(X, kk);
        const uint32_t m = k - W_C + 1;
        const uint32_t mm = (m >= 16) ? ~0u : ((1u << (2 * m)) - 1);
        uint32_t h[W_C > 0 ? PT + W_C - 1 : 1];
        if (W_C > 0) {
#pragma unroll
            for (int i = 0; i < W_C - 1; ++i) {
                const uint32_t fa = (uint32_t)(X >> (2 * i)) & mm;
                const uint32_t fb = (uint32_t)(B >> (2 * (W_C - 1 - i))) & mm;
                h[i] = mz_order(fa < fb ? fa : fb);
            }
        }
        const uint64_t nl0 = T.sub[0].nbuckets;
#pragma unroll
        for (int jj = 0; jj < PT; ++jj) {
            if (jj > 0) {  // roll one base: window start p0+jj
                X = ((A >> (2 * jj)) | (E << (64 - 2 * jj))) & kmask;
                const uint64_t nb = (X >> (2 * (kk - 1))) & 3ull;  // the base that entered
                B = ((B << 2) | (nb ^ 3ull)) & kmask;
            }
            bool ok = (p0 + jj) < npos;
            if (hasn) ok = ok && (((uint32_t)(NM >> jj) & kbits) == 0);
            uint32_t ln;
            const uint64_t ck = canonical_from_xb(X, B, kk);
            if (W_C > 0) {
                const uint32_t fa = (uint32_t)(X >> (2 * (W_C - 1))) & mm;  // last m-mer of this k-mer
                const uint32_t fb = (uint32_t)B & mm;                       // its reverse complement
                h[jj + W_C - 1] = mz_order(fa < fb ? fa : fb);
                uint32_t best = h[jj];
#pragma unroll
                for (int i = 1; i < W_C; ++i) best = min(best, h[jj + i]);
                grp[jj] = best;
            } else {
                grp[jj] = group_of_key(ck);
            }
            ln = home_of_group(grp[jj], nl0);
            key[jj] = ok ? ck : EMPTY_KEY;
            line[jj] = ln;
        }
    }

    // ---- phase 2, per sub-table ----
    for (uint32_t si = 0; si < T.nsub; ++si) {
        const SubTable st = T.sub[si];
        const bool two = (st.W == 2);
        if (si > 0) {
#pragma unroll
            for (int jj = 0; jj < PT; ++jj) line[jj] = home_of_group(grp[jj], st.nbuckets);
        }
        // invalid positions inherit their predecessor's line so that they never open a run
#pragma unroll
        for (int jj = 1; jj < PT; ++jj)
            if (key[jj] == EMPTY_KEY) line[jj] = line[jj - 1];
        __syncthreads();  // previous users of lastline / ctl / buf are done
        lastline[tid] = line[PT - 1];
        if (tid < 3) ctl[tid] = 0;
        __syncthreads();
        // run ids: a run = maximal stretch of consecutive positions with the same home line
        uint32_t flag[PT], cnt = 0;
#pragma unroll
        for (int jj = 0; jj < PT; ++jj) {
            const uint32_t prev = jj ? line[jj - 1] : (tid ? lastline[tid - 1] : ~line[0]);
            flag[jj] = (line[jj] != prev) ? 1u : 0u;
            cnt += flag[jj];
        }
        uint32_t incl = cnt;  // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) ctl[4 + wave] = incl;
        __syncthreads();
        uint32_t base = incl - cnt, R = 0;
#pragma unroll
        for (int w2 = 0; w2 < G::NWAVE; ++w2) {
            const uint32_t tw = ctl[4 + w2];
            if (w2 < wave) base += tw;
            R += tw;
        }
        uint32_t rid[PT];
        {
            uint32_t acc = base;
#pragma unroll
            for (int jj = 0; jj < PT; ++jj) {
                acc += flag[jj];
                rid[jj] = acc - 1;  // run id of position jj (>= 0: position 0 of the tile opens run 0)
            }
        }
        int32_t qidx[PT];
#pragma unroll
        for (int jj = 0; jj < PT; ++jj) qidx[jj] = -1;

        // round 0: stage the tile's distinct home lines (LCAP per pass), probe from LDS
        for (uint32_t r0 = 0; r0 < R; r0 += G::LCAP) {
            const uint32_t nl = min((uint32_t)G::LCAP, R - r0);
#pragma unroll
            for (int jj = 0; jj < PT; ++jj)
                if (flag[jj] && rid[jj] - r0 < nl) lines[rid[jj] - r0] = line[jj];
            __syncthreads();
            stage_lines<ANCHOR_UNROLL>(st.buckets, lines, nl, buf, tid);
            __syncthreads();
#pragma unroll
            for (int jj = 0; jj < PT; ++jj) {
                if (key[jj] != EMPTY_KEY && rid[jj] - r0 < nl) {
                    uint32_t m0, m1;
                    const int rc = two ? scan_line<true>(buf + (size_t)(rid[jj] - r0) * LDS_LINE_U4, key[jj], m0, m1)
                                       : scan_line<false>(buf + (size_t)(rid[jj] - r0) * LDS_LINE_U4, key[jj], m0, m1);
                    if (rc < 0) {  // absent from a full line: continue along the probe sequence via the queue
                        const uint32_t stp = step_of_group(grp[jj], st.nbuckets);
                        const uint32_t nx = next_line(line[jj], stp, st.nbuckets);
                        const uint32_t qi = atomicAdd(&ctl[0], 1u);
                        if (qi < (uint32_t)G::QCAP) {
                            q_key[qi] = key[jj];
                            q_line[qi] = nx;
                            q_step[qi] = stp;
                            wl0[qi] = (uint16_t)qi;
                            qidx[jj] = (int32_t)qi;
                        } else {
                            lane_chase(st, key[jj], nx, stp, m0, m1);  // queue full: resolve inline
                        }
                    }
                    words[jj][NDBS_C ? 0 : st.word0] = m0;
                    if (two) words[jj][NDBS_C ? 1 % MAXD : st.word0 + 1] = m1;
                }
            }
            __syncthreads();
        }
        // overflow rounds: pending queue entries try their next line
        {
            uint16_t *cur = wl0, *nxt = wl1;
            uint32_t n = min(ctl[0], (uint32_t)G::QCAP);
            for (uint32_t round = 1; n > 0; ++round) {
                if (tid == 0) ctl[1 + (round & 1)] = 0;
                for (uint32_t e0 = 0; e0 < n; e0 += G::LCAP) {
                    const uint32_t nl = min((uint32_t)G::LCAP, n - e0);
                    for (uint32_t i = tid; i < nl; i += ANCHOR_WG) lines[i] = q_line[cur[e0 + i]];
                    __syncthreads();
                    stage_lines<ANCHOR_UNROLL>(st.buckets, lines, nl, buf, tid);
                    __syncthreads();
                    for (uint32_t i = tid; i < nl; i += ANCHOR_WG) {
                        const uint32_t e = cur[e0 + i];
                        const uint64_t kq = q_key[e];
                        uint32_t m0, m1;
                        const int rc = two ? scan_line<true>(buf + (size_t)i * LDS_LINE_U4, kq, m0, m1)
                                           : scan_line<false>(buf + (size_t)i * LDS_LINE_U4, kq, m0, m1);
                        if (rc < 0) {
                            const uint32_t nx = next_line(q_line[e], q_step[e], st.nbuckets);
                            if (round < (uint32_t)ANCHOR_MAX_ROUNDS) {
                                q_line[e] = nx;
                                nxt[atomicAdd(&ctl[1 + (round & 1)], 1u)] = (uint16_t)e;
                            } else {
                                lane_chase(st, kq, nx, q_step[e], m0, m1);
                                q_key[e] = (uint64_t)m0 | ((uint64_t)m1 << 32);
                            }
                        } else {
                            q_key[e] = (uint64_t)m0 | ((uint64_t)m1 << 32);  // resolved: the entry now holds the masks
                        }
                    }
                    __syncthreads();
                }
                n = ctl[1 + (round & 1)];
                uint16_t *t2 = cur;
                cur = nxt;
                nxt = t2;
                __syncthreads();
            }
        }
#pragma unroll
        for (int jj = 0; jj < PT; ++jj) {
            if (qidx[jj] >= 0) {
                const uint64_t mv = q_key[qidx[jj]];
                words[jj][NDBS_C ? 0 : st.word0] = (uint32_t)mv;
                if (two) words[jj][NDBS_C ? 1 % MAXD : st.word0 + 1] = (uint32_t)(mv >> 32);
            }
        }
    }

    // ---- phase 3 (registers -> global) ----
    const uint32_t binlen = a.binlen;
    const uint32_t bin0 = tile_start / binlen;
    const uint32_t bin0_start = bin0 * binlen;
    const bool want_cs = (flags & 1u) != 0;
    const bool stats = (flags & 2u) == 0;  // rows-only mode leaves the statistics to k_rows_epilogue
    uint8_t *grow = out1 + a.out_off + (uint64_t)(tile_start + p0) * nbytes;
    if (NBYTES_C == 1 && PT % 4 == 0) {
        // PT consecutive 1-byte rows per thread -> 32-bit stores, coalesced across the wave
#pragma unroll
        for (int j4 = 0; j4 < PT; j4 += 4) {
            const uint32_t v = (words[j4][0] & 0xFFu) | ((words[j4 + 1][0] & 0xFFu) << 8) |
                               ((words[j4 + 2][0] & 0xFFu) << 16) | ((words[j4 + 3][0] & 0xFFu) << 24);
            if (p0 + j4 + 3 < npos) *reinterpret_cast<uint32_t *>(grow + j4) = v;
            else
                for (int b2 = 0; b2 < 4; ++b2)
                    if (p0 + j4 + b2 < npos) grow[j4 + b2] = (uint8_t)(v >> (8 * b2));
        }
    } else if (NBYTES_C == 8) {
#pragma unroll
        for (int jj = 0; jj < PT; ++jj)
            if (p0 + jj < npos)
                *reinterpret_cast<uint2 *>(grow + jj * 8) = make_uint2(words[jj][0], words[jj][1]);
    } else {
#pragma unroll
        for (int jj = 0; jj < PT; ++jj) {
            if (p0 + jj < npos) {
                for (uint32_t d = 0; d < ndbs; ++d) {
                    const uint32_t nb = min(4u, nbytes - 4 * d);
                    for (uint32_t bb = 0; bb < nb; ++bb)
                        grow[(uint64_t)jj * nbytes + 4 * d + bb] = (uint8_t)(words[jj][d < MAXD ? d : 0] >> (8 * bb));
                }
            }
        }
    }
    if (stats) {
#pragma unroll
        for (int jj = 0; jj < PT; ++jj) {
            const uint32_t pl = p0 + jj;
            const bool active = pl < npos;
            const uint32_t pos = tile_start + pl;
            uint32_t popc = 0;
            const bool is100 = active && (pos % 100u == 0);
            for (uint32_t d = 0; d < ndbs; ++d) {
                const uint32_t wv = active ? words[jj][d < MAXD ? d : 0] : 0u;
                popc += __popc(wv);
                if (is100) {
                    const uint32_t nb = min(4u, nbytes - 4 * d);
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
    uint64_t nchunks = t.nbuckets * (BUCKET_BYTES / 16);
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
    size_t b = (size_t)G::LCAP * LDS_LINE_U4 * 16;
    b += (size_t)G::QCAP * 8 + 2 * (size_t)G::QCAP * 4;
    b += (size_t)G::LCAP * 4;
    b += 2 * (size_t)G::QCAP * 2;
    b += G::SEQW * 8 + G::SEQW * 4;
    b += ANCHOR_WG * 4;
    b += ((2 * (ngenomes + 1) + 3) & ~3u) * 4;
    b += ((ngenomes + 3) & ~3u) * 4;
    b += 64;
    return (b + 15) & ~(size_t)15;
}

template <int W_C, int NDBS_C, int NBYTES_C>
static hipError_t launch_anchor_t(hipStream_t st, size_t lds, uint32_t ntiles, const TableDesc &T,
                                  const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                                  const SeqDesc *sd, const AnchorDesc *ad, const uint32_t *tile_contig,
                                  uint8_t *out1, uint8_t *out100, uint32_t *bins, unsigned long long *colsums,
                                  uint32_t flags) {
    auto kern = k_anchor<ANCHOR_TILE, W_C, NDBS_C, NBYTES_C>;
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

template <int W_C>
static hipError_t launch_anchor_w(hipStream_t st, size_t lds, uint32_t ntiles, const TableDesc &T,
                                  const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                                  const SeqDesc *sd, const AnchorDesc *ad, const uint32_t *tile_contig,
                                  uint8_t *out1, uint8_t *out100, uint32_t *bins, unsigned long long *colsums,
                                  uint32_t flags) {
    const uint32_t nbytes = (T.ngenomes + 7) / 8;
#define PG_ARGS st, lds, ntiles, T, seqw, nmw, has_n, sd, ad, tile_contig, out1, out100, bins, colsums, flags
    if (T.ndbs == 1 && nbytes == 1) return launch_anchor_t<W_C, 1, 1>(PG_ARGS);
    if (T.ndbs == 1) return launch_anchor_t<W_C, 1, 0>(PG_ARGS);
    if (T.ndbs == 2 && nbytes == 8) return launch_anchor_t<W_C, 2, 8>(PG_ARGS);
    if (T.ndbs == 2) return launch_anchor_t<W_C, 2, 0>(PG_ARGS);
    return launch_anchor_t<W_C, 0, 0>(PG_ARGS);
}

hipError_t launch_anchor(hipStream_t st, const TableDesc &T, const uint64_t *seqw, const uint32_t *nmw,
                         const uint32_t *has_n, const SeqDesc *sd, const AnchorDesc *ad,
                         const uint32_t *tile_contig, uint32_t ntiles, uint8_t *out1, uint8_t *out100,
                         uint32_t *bins, unsigned long long *colsums, uint32_t flags) {
    if (ntiles == 0) return hipSuccess;
    const size_t lds = anchor_lds_bytes(T.ngenomes);
    // the kernel's compile-time minimizer window must be the one the table was built with
    const uint32_t w = T.sub[0].m ? T.k - T.sub[0].m + 1 : 0;
    switch (w) {
        case 0: return launch_anchor_w<0>(PG_ARGS);
        case 8: return launch_anchor_w<8>(PG_ARGS);
        case 12: return launch_anchor_w<12>(PG_ARGS);
        case 16: return launch_anchor_w<16>(PG_ARGS);
        default: return hipErrorInvalidValue;
    }
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
