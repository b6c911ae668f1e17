// pg_anchor.hip — the anchor hot path on gfx950 (CDNA4, wave64): integer / HBM-bound, no MFMA.
//
// Replaces (reference, kjenike/panagram): KMC CKMCFile::GetCountersForRead as called from
// KMCdb::write_bits (cpp/anchor.cpp:112-195) and Genome._write_bitmap / _query_kmc_bytes /
// bin_bitsum (index.py:932-969,1169-1183).
//
// Two lean kernels per (result, sub-table); instruction count per position is what bounds them once
// the table fetches are shared between neighbouring positions (minimizer homes) and between the
// anchor genomes of a pangenome (co-scheduled tiles):
//
//   k_probe      ONE WAVE = one tile of TILE consecutive k-mer positions, no barriers, 4.6 KB of LDS
//                (32 waves per CU).  Per batch of 64 lanes (lane = one position, neighbours =
//                neighbouring positions):
//                  * canonical k-mer from the LDS-staged 2-bit sequence (~0.25 B/pos from HBM)
//                  * minimizer = sliding minimum over the W_C m-mer ranks of the neighbouring
//                    lanes (DPP wave shifts) -> home line; runs of equal home line are found with
//                    one ballot (leaders) and numbered with mbcnt
//                  * LDS-STAGED PROBE BATCH: the batch's distinct table lines (~15 for 58
//                    positions) are fetched cooperatively and coalesced — 8 lanes x 16 B = one
//                    128-byte line — into a per-wave LDS buffer, then every lane scans the 8
//                    slots of ITS line out of LDS
//                  * the presence row goes straight to bitmap.1; a key absent from a FULL line
//                    joins the tile's in-wave overflow queue (position, next line, step), which
//                    drain_queue works off in dense 64-entry batches, level by level
//   k_epilogue   streaming statistics from the finished bitmap.1 rows: bitmap.100 (1-in-100
//                rows), per-bin popcount histogram, per-contig column sums — persistent
//                workgroups, register accumulators, one instantiation per row width (1..8 bytes;
//                16 consecutive rows per thread over 4 full tiles of one bin).
//   k_epilogue_chunks  the same for rows wider than 8 bytes (more than 64 genomes): a lane owns one
//                16-byte chunk of the rows it visits, one launch reads every row once.
//   k_window_stats, k_cols_extract / k_cols_merge: side paths (gene / bin windows; the
//                genome-sharded exchange).
//
// Bytes from HBM per position: 0.25 (sequence) + 128 x (lines missed in L2 per
// position: 0.34 with one launch per genome, 0.095 co-scheduled) + row bytes written + re-read.
#include "pg_kernels.h"
#include <map>
#include <mutex>
#include <utility>
#include <type_traits>

#include <algorithm>

// ---- translation units (round 6) ----
// k_probe and k_insert_tile are instantiated per minimizer window (0, 3..8) x row mode x layout x (fused): 300 kernels, three
// minutes of hipcc in one unit.  panagram_amd/build.py compiles this file THREE times in parallel, each unit instantiating the
// windows of its part (probe_part<N> / insert_part<N> below); part 0 also holds every other kernel and launcher of the file.
// -1 (the default: tools/isa.sh, tools/build_variant.sh, a plain `hipcc pg_anchor.hip`): everything in one unit.
#ifndef PG_ANCHOR_PART
#define PG_ANCHOR_PART -1
#endif
#define PG_HAS_PART(n) (PG_ANCHOR_PART == -1 || PG_ANCHOR_PART == (n))
#define PG_MAIN_PART (PG_ANCHOR_PART <= 0)
#if defined(PG_PHASE_TIMING) && PG_ANCHOR_PART != -1
#error "a -DPG_PHASE_TIMING build keeps its counters in one device variable: compile pg_anchor.hip as ONE unit (PG_ANCHOR_PART unset)"
#endif

namespace pg {

// windows 0, 3, 4 -> part 0; 5, 6 -> part 1; 7, 8 -> part 2
hipError_t probe_part0(hipStream_t s, uint32_t w, uint32_t ntiles, const SubTable &st, const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                       const SeqDesc *sd, const AnchorDesc *ad, const uint32_t *tile_contig, const uint32_t *sched, uint32_t tile_base, uint8_t *out1,
                       uint32_t nbytes, const RowCols &rc, int rowmode, const FuseArgs *fuse);
hipError_t probe_part1(hipStream_t s, uint32_t w, uint32_t ntiles, const SubTable &st, const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                       const SeqDesc *sd, const AnchorDesc *ad, const uint32_t *tile_contig, const uint32_t *sched, uint32_t tile_base, uint8_t *out1,
                       uint32_t nbytes, const RowCols &rc, int rowmode, const FuseArgs *fuse);
hipError_t probe_part2(hipStream_t s, uint32_t w, uint32_t ntiles, const SubTable &st, const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                       const SeqDesc *sd, const AnchorDesc *ad, const uint32_t *tile_contig, const uint32_t *sched, uint32_t tile_base, uint8_t *out1,
                       uint32_t nbytes, const RowCols &rc, int rowmode, const FuseArgs *fuse);
hipError_t insert_part0(hipStream_t s, uint32_t win, uint32_t ntiles, const SubTable &st, int w, uint32_t bits, uint32_t count_mode, const uint64_t *seqw,
                        const uint32_t *nmw, const uint32_t *has_n, const SeqDesc *sd, const uint32_t *tile0, uint32_t ncontigs,
                        unsigned long long *counters, uint32_t max_probe);
hipError_t insert_part1(hipStream_t s, uint32_t win, uint32_t ntiles, const SubTable &st, int w, uint32_t bits, uint32_t count_mode, const uint64_t *seqw,
                        const uint32_t *nmw, const uint32_t *has_n, const SeqDesc *sd, const uint32_t *tile0, uint32_t ncontigs,
                        unsigned long long *counters, uint32_t max_probe);
hipError_t insert_part2(hipStream_t s, uint32_t win, uint32_t ntiles, const SubTable &st, int w, uint32_t bits, uint32_t count_mode, const uint64_t *seqw,
                        const uint32_t *nmw, const uint32_t *has_n, const SeqDesc *sd, const uint32_t *tile0, uint32_t ncontigs,
                        unsigned long long *counters, uint32_t max_probe);
hipError_t preload_part1();
hipError_t preload_part2();

constexpr int PROBE_SEQW = ((PROBE_TILE + 31) / 32 + 6 + 3) & ~3;  // staged 32-base words per tile
// The tile's packed bases are staged TWICE: as they are (sw) and reverse-complemented (rw: base j of rw = complement of
// base 32 PROBE_SEQW - 1 - j of sw).  The reverse complement of the k-mer at base p is then the k bases of rw from
// 32 PROBE_SEQW - p - k on — the same three LDS reads and two funnel shifts as the forward window, instead of a
// 64-bit bit reversal, pair swap, complement and shift per position (14 instructions).
#ifndef PG_ABLATE
#define PG_ABLATE 0  // timing experiments of k_probe (tools/ab_ablate.sh); 0 = the product
#endif
#ifndef PG_RC_LDS
#define PG_RC_LDS 1
#endif
#ifndef PG_PROBE_PIPE
#define PG_PROBE_PIPE 2  // 1: the front end of batch i + 1 ahead of the table look-up of batch i (k_probe); 2: and its fetch issued as soon as batch i's chunks are staged
#endif
constexpr uint32_t PROBE_SEQ_BASES = 32u * PROBE_SEQW;
// k_probe's workgroup is ONE wave: what its lanes hand each other through LDS (the staged lines, the batch's line numbers, the
// overflow queue) needs no barrier — a wave's LDS instructions are executed in the order they were issued, so a read issued
// behind a write sees it.  __syncthreads() costs such a kernel an s_waitcnt lgkmcnt(0) — every LDS operation in flight drained —
// where a wavefront-scope fence emits nothing (38 -> 31 full drains in the one-byte instantiation, the read of the batch's line
// numbers issued right behind their write).  -DPG_WAVE_SYNC=1 builds that; measured (profiles/r6z_ab_wave_sync.txt, two rounds
// on one box): configs[1] 3.26 / 3.18 -> 3.27 / 3.23 ms, 27 x 40 Mb 4.45 -> 4.38, 64 x 20 Mb k = 31 5.79 -> 5.79, one launch per
// genome and d = 5 % unchanged — the round trips it takes off a wave's critical path are slots the seven other waves were already
// using (latency is not what binds).  Off: the barriers say what is meant.
#ifndef PG_WAVE_SYNC
#define PG_WAVE_SYNC 0
#endif
#if PG_WAVE_SYNC
#define PG_WSYNC()                                              \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
    } while (0)
#else
#define PG_WSYNC() __syncthreads()
#endif
// -DPG_PHASE_TIMING: a measuring build (tools/phase_timing.py) — every wave of k_probe stamps s_memtime at its phase
// boundaries and adds the phases' cycles to pg_phase_cycles at the end of its tile: where a wave's time goes, waits for
// the other waves of its SIMD included.  Slots: 0 prologue, 1 front end, 2 wait for the lines, 3 staging, 4 issue of the
// next fetch, 5 slot scan, 6 row store, 7 loop bookkeeping, 8 drain + tail, 9 waves, 10 overflow entries.
#ifdef PG_PHASE_TIMING
__device__ unsigned long long pg_phase_cycles[1024 * 16];  // (1024 sets, by block number: ten same-address atomics per wave serialise the launch)
#define PG_PH_DECL uint32_t ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; uint32_t ph_t = (uint32_t)__builtin_readcyclecounter();
#define PG_PH(i) { const uint32_t ph_n = (uint32_t)__builtin_readcyclecounter(); ph_acc[i] += ph_n - ph_t; ph_t = ph_n; }
#define PG_PH_FLUSH { if (threadIdx.x == 0) { for (int ph_i = 0; ph_i < 9; ++ph_i) atomicAdd(&pg_phase_cycles[(blockIdx.x & 1023u) * 16u + ph_i], (unsigned long long)ph_acc[ph_i]); atomicAdd(&pg_phase_cycles[(blockIdx.x & 1023u) * 16u + 9], 1ull); atomicAdd(&pg_phase_cycles[(blockIdx.x & 1023u) * 16u + 10], (unsigned long long)ph_acc[10]); } }
#else
#define PG_PH_DECL
#define PG_PH(i)
#define PG_PH_FLUSH
#endif
// (rcb = PROBE_SEQ_BASES - k, handed in: written as PROBE_SEQ_BASES - p - k the compiler adds p and k per position first)
__device__ __forceinline__ uint64_t revcomp_window(const uint64_t *rw, uint64_t X, uint32_t p, int k, uint64_t kmask, uint32_t rcb) {
#if PG_RC_LDS
    return extract_bases32(reinterpret_cast<const uint32_t *>(rw), rcb - p) & kmask;
#else
    return revcomp_le(X, k);
#endif
}

// scan the 8 slots of a line staged in LDS.  Lines fill front to back without holes (an insert
// claims the first EMPTY slot and slots never revert), so "full" == last slot used.
// returns 1 = found, 0 = absent (line not full), -1 = absent from a full line
#ifndef PG_LDS_SOA
#define PG_LDS_SOA 1  // staged lines keep their keys and their mask words apart in LDS (stage_chunk / scan_line_lds)
#endif
// A table line {key, m0, m1} x SLOTS is staged into LDS as SLOTS keys followed by SLOTS mask pairs: the scan then takes its
// keys in SLOTS / 2 ds_read_b128 — 4 LDS cycles per 16 bytes and lane — where one ds_read2_b64 per two slots of the
// interleaved line took 8 (MI355X_MICROARCH.md, LDS table): 16 instead of 32 LDS cycles per batch for the 8 keys, the
// largest single item of a batch's ~95.  The staging lane's 16-byte chunk leaves as two 8-byte stores (6 + 6 cycles
// against 13 for the one ds_write_b128).
__device__ __forceinline__ void stage_chunk(uint4 *line, uint32_t slot, int slots, const uint4 v) {
#if PG_LDS_SOA
    uint2 *const p = reinterpret_cast<uint2 *>(line);
    p[slot] = make_uint2(v.x, v.y);
    p[slots + slot] = make_uint2(v.z, v.w);
#else
    line[slot] = v;
#endif
}
template <bool TWO, int SLOTS>
__device__ __forceinline__ int scan_line_lds(const uint4 *line, uint64_t key, uint32_t &m0, uint32_t &m1) {
    // all SLOTS key reads are issued together (one LDS wait); one compare per slot into a scalar lane mask;
    // the scalar unit folds the masks into the three bits of the hit slot's number (at most one slot holds the
    // key), three v_cndmask turn them into the slot's byte offset, and only that slot's masks are read
    uint64_t kk[SLOTS];
#if PG_LDS_SOA
#pragma unroll
    for (int j = 0; j < SLOTS / 2; ++j) {
        const uint4 v = line[j];
        kk[2 * j] = (uint64_t)v.x | ((uint64_t)v.y << 32);
        kk[2 * j + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
    constexpr uint32_t SLOT_BYTES = 8u, MASK0 = 8u * SLOTS;  // a slot's masks: MASK0 + 8 * slot
#else
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) kk[sl] = *reinterpret_cast<const uint64_t *>(line + sl);
    constexpr uint32_t SLOT_BYTES = 16u, MASK0 = 8u;
#endif
    m0 = m1 = 0;
    if constexpr (SLOTS == 8) {
        // (ballot of a compare = v_cmp_eq_u64 with a scalar destination; inverse_ballot = the SGPR pair used as a
        // v_cndmask selector / exec mask: the compiler's lowering of `kk == key ? sl : hit` goes through VCC with
        // two VALU instructions and a wait state per slot)
        unsigned long long e[8];
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) e[sl] = __builtin_amdgcn_ballot_w64(kk[sl] == key);
        const unsigned long long b0 = e[1] | e[3] | e[5] | e[7], b1 = e[2] | e[3] | e[6] | e[7], b2 = e[4] | e[5] | e[6] | e[7];
        const unsigned long long any = b0 | b1 | b2 | e[0];
        const bool hit = __builtin_amdgcn_inverse_ballot_w64(any);
#if PG_ABLATE == 6  // (timing experiment: keys read and compared, the hit slot's mask word NOT read)
        if (hit) m0 = (uint32_t)b0 | 1u;
        return hit ? 1 : (kk[SLOTS - 1] == EMPTY_KEY ? 0 : -1);
#elif PG_ABLATE == 7  // (timing experiment: keys read, one compare instead of eight)
        m0 = (uint32_t)(kk[0] ^ kk[1] ^ kk[2] ^ kk[3] ^ kk[4] ^ kk[5] ^ kk[6] ^ kk[7]) | 1u;
        return (kk[0] ^ kk[3]) == key ? -1 : 1;
#endif
        if (hit) {
            const uint32_t off = (__builtin_amdgcn_inverse_ballot_w64(b0) ? SLOT_BYTES : 0u) | (__builtin_amdgcn_inverse_ballot_w64(b1) ? 2u * SLOT_BYTES : 0u) |
                                 (__builtin_amdgcn_inverse_ballot_w64(b2) ? 4u * SLOT_BYTES : 0u);
            if constexpr (TWO) {
                const uint2 mk = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint8_t *>(line) + off + MASK0);
                m0 = mk.x;
                m1 = mk.y;
            } else {
                m0 = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(line) + off + MASK0);
            }
        }
        return hit ? 1 : (kk[SLOTS - 1] == EMPTY_KEY ? 0 : -1);
    } else {
        int hit = -1;
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) hit = (kk[sl] == key) ? sl : hit;
        if (hit >= 0) {
            const uint2 mk = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint8_t *>(line) + (uint32_t)hit * SLOT_BYTES + MASK0);
            m0 = mk.x;
            if (TWO) m1 = mk.y;
        }
        return hit >= 0 ? 1 : (kk[SLOTS - 1] == EMPTY_KEY ? 0 : -1);
    }
}

// The same scan WITHOUT control flow, for the batches whose every active lane has a staged line (k_probe's cut batches): all
// 64 lanes read and compare — a lane without a k-mer whatever its clamped run number points at — the hit slot's mask
// word is read by every lane (slot 0's where nothing matched) and selected by `any & amask`; the lanes whose key is absent
// from a FULL line come back as a lane mask.  No exec-mask save / restore around the scan and around the hit path, no return
// code rebuilt through 0 / 1 / -1 and compared again: 11 scalar instructions and three branches fewer per batch.  Measured:
// within 0.5 % of the scan with its branches — what a dummy instruction costs in the FRONT END of a batch (profiles/
// r4e_probe_dummy_salu.txt) an instruction saved behind the fetch does not give back; PG_SCAN_FLAT stays 0.
template <bool TWO>
__device__ __forceinline__ unsigned long long scan_line_lds_flat(const uint4 *line, uint64_t key, unsigned long long amask, uint32_t &m0,
                                                                 uint32_t &m1) {
    static_assert(PG_LDS_SOA == 1, "scan_line_lds_flat reads the split staged line");
    constexpr int SLOTS = 8;
    uint64_t kk[SLOTS];
#pragma unroll
    for (int j = 0; j < SLOTS / 2; ++j) {
        const uint4 v = line[j];
        kk[2 * j] = (uint64_t)v.x | ((uint64_t)v.y << 32);
        kk[2 * j + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
    constexpr uint32_t SLOT_BYTES = 8u, MASK0 = 8u * SLOTS;
    unsigned long long e[8];
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) e[sl] = __builtin_amdgcn_ballot_w64(kk[sl] == key);
    const unsigned long long b0 = e[1] | e[3] | e[5] | e[7], b1 = e[2] | e[3] | e[6] | e[7], b2 = e[4] | e[5] | e[6] | e[7];
    const unsigned long long any = b0 | b1 | b2 | e[0];
    const uint32_t off = (__builtin_amdgcn_inverse_ballot_w64(b0) ? SLOT_BYTES : 0u) | (__builtin_amdgcn_inverse_ballot_w64(b1) ? 2u * SLOT_BYTES : 0u) |
                         (__builtin_amdgcn_inverse_ballot_w64(b2) ? 4u * SLOT_BYTES : 0u);
    const bool hit = __builtin_amdgcn_inverse_ballot_w64(any & amask);
    if constexpr (TWO) {
        const uint2 mk = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint8_t *>(line) + off + MASK0);
        m0 = hit ? mk.x : 0u;
        m1 = hit ? mk.y : 0u;
    } else {
        const uint32_t mk = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(line) + off + MASK0);
        m0 = hit ? mk : 0u;
        m1 = 0;
    }
    return amask & ~any & __builtin_amdgcn_ballot_w64(kk[SLOTS - 1] != EMPTY_KEY);
}

// one line straight from global memory: the 8 slot loads are issued together (one latency)
template <bool TWO, int SLOTS>
__device__ __forceinline__ int scan_line_global(const SubTable &st, uint32_t b, uint64_t key, uint32_t &m0, uint32_t &m1) {
    const uint4 *line = reinterpret_cast<const uint4 *>(st.buckets + (uint64_t)b * (16 * SLOTS));
    uint4 v[SLOTS];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) v[sl] = line[sl];
    m0 = m1 = 0;
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const uint64_t kk = (uint64_t)v[sl].x | ((uint64_t)v[sl].y << 32);
        const bool hit = (kk == key);
        m0 = hit ? v[sl].z : m0;
        if (TWO) m1 = hit ? v[sl].w : m1;
    }
    const uint64_t last = (uint64_t)v[SLOTS - 1].x | ((uint64_t)v[SLOTS - 1].y << 32);
    return (m0 | m1) ? 1 : (last == EMPTY_KEY ? 0 : -1);
}

// follow a key's probe sequence from its line number `level` (b, step) to the end
template <bool TWO, int SLOTS>
__device__ __forceinline__ void lane_chase(const SubTable &st, uint64_t key, uint32_t level, uint32_t b, uint32_t step,
                                           uint32_t &m0, uint32_t &m1) {
    for (uint64_t n = 0; n < st.nbuckets + GROUP_CHAIN; ++n) {
        if (scan_line_global<TWO, SLOTS>(st, b, key, m0, m1) >= 0) return;
        level = min(level + 1, GROUP_CHAIN + 1);
        advance_line(key, level, st.nbuckets, b, step);
    }
    m0 = m1 = 0;
}

// ---- split layout (more than 64 genomes: 16 bare keys per line, mask words in a second array) ----
// A hit is reported as (line, slot + 1) — slot1 == 0: absent — and the row is copied from the mask array afterwards.
// scan of a key line staged in LDS: 1 = found, 0 = absent (line not full), -1 = absent from a full line
// FIRST: several slots may hold the key (k_insert_tile's snapshots: a claim that is about to be retired, see
// wave_insert_batch) — report the lowest; otherwise at most one does
template <bool FIRST = false>
__device__ __forceinline__ int scan_keys16_lds(const uint4 *line, uint64_t key, uint32_t &slot1) {
    uint64_t kk[16];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const uint4 v = line[c];
        kk[2 * c] = (uint64_t)v.x | ((uint64_t)v.y << 32);
        kk[2 * c + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
    unsigned long long e[16];
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) e[sl] = __builtin_amdgcn_ballot_w64(kk[sl] == key);
    if constexpr (FIRST) {  // (scalar unit: a lane's match in a lower slot masks its later ones)
        unsigned long long seen = e[0];
#pragma unroll
        for (int sl = 1; sl < 16; ++sl) {
            const unsigned long long m = e[sl];
            e[sl] = m & ~seen;
            seen |= m;
        }
    }
    unsigned long long b[4] = {0, 0, 0, 0}, any = e[0];
#pragma unroll
    for (int sl = 1; sl < 16; ++sl) {
        any |= e[sl];
#pragma unroll
        for (int bit = 0; bit < 4; ++bit)
            if (sl & (1 << bit)) b[bit] |= e[sl];
    }
    const bool hit = __builtin_amdgcn_inverse_ballot_w64(any);
    slot1 = 0;
    if (hit)
        slot1 = 1u + ((__builtin_amdgcn_inverse_ballot_w64(b[0]) ? 1u : 0u) | (__builtin_amdgcn_inverse_ballot_w64(b[1]) ? 2u : 0u) |
                      (__builtin_amdgcn_inverse_ballot_w64(b[2]) ? 4u : 0u) | (__builtin_amdgcn_inverse_ballot_w64(b[3]) ? 8u : 0u));
    return hit ? 1 : (kk[15] == EMPTY_KEY ? 0 : -1);
}

// follow a key's probe sequence through key lines in global memory (16 key loads in flight per line)
__device__ __forceinline__ void lane_chase_wide(const SubTable &st, uint64_t key, uint32_t level, uint32_t b, uint32_t step,
                                                uint32_t &hline, uint32_t &slot1) {
    hline = slot1 = 0;
    for (uint64_t n = 0; n < st.nbuckets + GROUP_CHAIN; ++n) {
        const uint4 *line = reinterpret_cast<const uint4 *>(st.buckets + (uint64_t)b * (8u * SPLIT_KEYS));
        uint4 v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = line[c];
        uint32_t found = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (((uint64_t)v[c].x | ((uint64_t)v[c].y << 32)) == key) found = 2 * c + 1;
            if (((uint64_t)v[c].z | ((uint64_t)v[c].w << 32)) == key) found = 2 * c + 2;
        }
        if (found) {
            hline = b;
            slot1 = found;
            return;
        }
        if (((uint64_t)v[7].z | ((uint64_t)v[7].w << 32)) == EMPTY_KEY) return;  // line not full: absent
        level = min(level + 1, GROUP_CHAIN + 1);
        advance_line(key, level, st.nbuckets, b, step);
    }
}

// the row of a position: the W mask words of (line, slot1 - 1), or zeros (slot1 == 0); nbytes = ceil(N / 8).
// Rows and mask blocks are contiguous, so the copy goes in the widest pieces the width allows — one dwordx4 / x3 /
// x2 access per lane covers the wave's rows back to back (word-by-word stores at a 12- or 20-byte lane stride
// ran 2x slower); only 4-byte alignment is needed for them (gfx950 global memory takes unaligned wide accesses).
template <int NW>
struct __attribute__((packed, aligned(4))) WordsN {
    uint32_t w[NW];
};
#ifndef PG_ROW12_NT
#define PG_ROW12_NT 1  // 12-byte rows (89..96 genomes, inline layout) as one non-temporal store: 96 genomes 6.75 -> 6.12 ms (profiles/r5m_ab_row_fuse.txt)
#endif
#ifndef PG_ROW4_NT
#define PG_ROW4_NT 0  // two- and four-byte rows non-temporal: 27 x 40 Mb 4.96 -> 5.59 ms, 16 x 50 Mb 3.35 -> 3.86 (profiles/r5m_ab_row_fuse.txt): not
#endif
#ifndef PG_NT_ROWS
#define PG_NT_ROWS 1  // non-temporal stores for rows that ONE store instruction covers (8 and 16 bytes): see store_row
#endif
template <int NW, bool NT = false>
__device__ __forceinline__ void copy_words(const uint32_t *src, uint8_t *dst, bool hit) {
    WordsN<NW> v;
#pragma unroll
    for (int i = 0; i < NW; ++i) v.w[i] = 0;
    if (hit) v = *reinterpret_cast<const WordsN<NW> *>(src);
    if constexpr (NW == 4 && NT && PG_NT_ROWS) {  // (non-temporal, like the 8-byte rows of store_row)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 q = {v.w[0], v.w[1], v.w[2], v.w[3]};
        __builtin_nontemporal_store(q, reinterpret_cast<u32x4 *>(dst));
    } else {
        *reinterpret_cast<WordsN<NW> *>(dst) = v;
    }
}
template <int NS>  // NS whole words and `tail` bytes of the next one: NS + 1 words in one load
__device__ __forceinline__ void copy_words_tail(const uint32_t *src, uint8_t *dst, bool hit, uint32_t tail) {
    WordsN<NS + 1> v;
#pragma unroll
    for (int i = 0; i <= NS; ++i) v.w[i] = 0;
    if (hit) v = *reinterpret_cast<const WordsN<NS + 1> *>(src);
    if constexpr (NS > 0) {
        WordsN<NS> o;
#pragma unroll
        for (int i = 0; i < NS; ++i) o.w[i] = v.w[i];
        *reinterpret_cast<WordsN<NS> *>(dst) = o;
    }
    if (NS > 0 && tail > 1) {  // (wave-uniform) the last 4 bytes of the row as one dword that overlaps the words before it — equal
                               // bytes — instead of 2 or 3 single bytes (15-byte rows: 13.6 ps per position, 12- and 16-byte rows 9.5-9.8)
        struct __attribute__((packed)) U32 { uint32_t v; };
        reinterpret_cast<U32 *>(dst + 4 * NS + tail - 4)->v = __builtin_amdgcn_alignbit(v.w[NS], v.w[NS > 0 ? NS - 1 : 0], 8u * tail);
    } else {
        for (uint32_t bb = 0; bb < tail; ++bb) dst[4 * NS + bb] = (uint8_t)(v.w[NS] >> (8 * bb));
    }
}
__device__ __forceinline__ void store_row_wide(const uint8_t *masks, uint32_t W, uint32_t nbytes, uint8_t *row, uint32_t hline,
                                               uint32_t slot1) {
    const bool hit = slot1 != 0;
    const uint32_t *mp = reinterpret_cast<const uint32_t *>(masks) + ((uint64_t)hline * SPLIT_KEYS + (slot1 - 1u)) * W;
    uint32_t d = 0;
    const uint32_t full = nbytes / 4;  // (wave-uniform loop bounds)
    const uint32_t tail = nbytes % 4;  // bytes of a last, partial word: fetched with the piece before it (one request)
    if (nbytes == 16) {  // (wave-uniform) the whole row in one store: non-temporal
        copy_words<4, true>(mp, row, hit);
        return;
    }
    for (; d + 4 <= full; d += 4) copy_words<4>(mp + d, row + 4 * d, hit);
    if (!tail) {
        if (full - d == 3) copy_words<3>(mp + d, row + 4 * d, hit);
        else if (full - d == 2) copy_words<2>(mp + d, row + 4 * d, hit);
        else if (full - d == 1) copy_words<1>(mp + d, row + 4 * d, hit);
    } else {
        if (full - d == 3) copy_words_tail<3>(mp + d, row + 4 * d, hit, tail);
        else if (full - d == 2) copy_words_tail<2>(mp + d, row + 4 * d, hit, tail);
        else if (full - d == 1) copy_words_tail<1>(mp + d, row + 4 * d, hit, tail);
        else copy_words_tail<0>(mp + d, row + 4 * d, hit, tail);
    }
}

// ---- inline layout (65..128 genomes, round 5): a line = S bare keys (S = 6 at W = 3 mask words, 5 at W = 4) followed by the slots'
// mask blocks; staged whole like a split-layout key line, but a hit's mask words come out of the SAME staged line ----
// scan of a staged line: 1 = found, 0 = absent (line not full), -1 = absent from a full line; slot1 = slot + 1 (0: absent).
// The hit's slot number stays on the vector unit (one compare + one select per key; from the last key down, so that the lowest
// match stays — k_insert_tile's snapshots may show a claim that is about to be retired).  Words 5.. of the line that hold mask
// bits (S = 5) are not compared.
__device__ __forceinline__ int scan_keys_inl(const uint4 *line, uint64_t key, uint32_t S, uint32_t &slot1) {
    uint64_t kk[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint4 v = line[c];
        kk[2 * c] = (uint64_t)v.x | ((uint64_t)v.y << 32);
        kk[2 * c + 1] = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
    const bool six = S >= 6u;  // (wave-uniform)
    slot1 = (six && kk[5] == key) ? 6u : 0u;
#pragma unroll
    for (int sl = 4; sl >= 0; --sl) slot1 = (kk[sl] == key) ? (uint32_t)(sl + 1) : slot1;
    const uint64_t last = six ? kk[5] : kk[4];
    return slot1 ? 1 : (last == EMPTY_KEY ? 0 : -1);
}
// the W (3 or 4) mask words of slot `slot` out of a staged line: four words from byte 8 S + 4 W slot on (W = 3: the fourth is the
// next slot's first word or the line's pad — rows of W = 3 tables are at most 12 bytes, it is never stored)
__device__ __forceinline__ void masks_inl(const uint4 *line, uint32_t S, uint32_t W, uint32_t slot, uint32_t (&w)[4]) {
    const WordsN<4> v = *reinterpret_cast<const WordsN<4> *>(reinterpret_cast<const uint8_t *>(line) + 8u * S + 4u * W * slot);
    w[0] = v.w[0], w[1] = v.w[1], w[2] = v.w[2], w[3] = v.w[3];
}
// a row of 9..16 bytes out of registers (zeros for an absent key): the widest pieces the width allows, as store_row_wide
template <int NS>
__device__ __forceinline__ void store_words_tail(const uint32_t (&v)[4], uint8_t *dst, uint32_t tail) {
    if (PG_ROW12_NT && NS == 3 && tail == 0) {  // (wave-uniform) a 12-byte row is one store: non-temporal
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3), aligned(4)));
        u32x3 q = {v[0], v[1], v[2]};
        __builtin_nontemporal_store(q, reinterpret_cast<u32x3 *>(dst));
        return;
    }
    WordsN<NS> o;
#pragma unroll
    for (int i = 0; i < NS; ++i) o.w[i] = v[i];
    *reinterpret_cast<WordsN<NS> *>(dst) = o;
    if (tail > 1) {  // (wave-uniform) the row's last 4 bytes as one dword that overlaps the words before it — equal bytes
        struct __attribute__((packed)) U32 { uint32_t v; };
        reinterpret_cast<U32 *>(dst + 4 * NS + tail - 4)->v = __builtin_amdgcn_alignbit(v[NS & 3], v[NS - 1], 8u * tail);
    } else if (tail == 1) {
        dst[4 * NS] = (uint8_t)v[NS & 3];
    }
}
__device__ __forceinline__ void store_row_regs(uint8_t *row, uint32_t nbytes, const uint32_t (&w)[4]) {
    if (nbytes == 16) {  // (wave-uniform) one store covers the row: non-temporal, as store_row_wide
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 q = {w[0], w[1], w[2], w[3]};
        if (PG_NT_ROWS) __builtin_nontemporal_store(q, reinterpret_cast<u32x4 *>(row));
        else *reinterpret_cast<u32x4 *>(row) = q;
    } else if (nbytes >= 12) {
        store_words_tail<3>(w, row, nbytes - 12u);
    } else {
        store_words_tail<2>(w, row, nbytes - 8u);
    }
}
// A ragged row (9..11 or 13..15 bytes) as ONE store: the 12 or 16 bytes from the row's start on = the row and the first bytes of the
// NEXT row, which the next lane holds (one DPP shift of its first word) — overlapping stores of one instruction write equal bytes.
// A lane whose successor has no row in this batch writes zeros there: the next batch's own store, later in this wave's program
// order, puts them right — so not for the tile's last batch (the bytes behind its last row are another wave's), and not for the
// rows the overflow drain resolves (store_row_regs: exact).  Two stores per row at odd addresses cost the probe of 65 genomes
// 8 % against the one store of 12-byte rows (7.6 against 7.0 ps per position).
#ifndef PG_ROW_FUSE
#define PG_ROW_FUSE 1
#endif
#ifndef PG_ROW_FUSE3
#define PG_ROW_FUSE3 1  // three-byte rows as one unaligned dword per row (0: round 3's aligned-dword scheme)
#endif
#ifndef PG_ROW_FUSE_NT
#define PG_ROW_FUSE_NT 1  // the one-store rows of 9..11 and 13..15 bytes written non-temporally (65 genomes 4.3-4.5 -> 3.98 ms, 72 x 30 Mb 14.7 -> 13.8; profiles/r5m_ab_row_fuse.txt)
#endif
#ifndef PG_ROW_FUSE8
#define PG_ROW_FUSE8 1  // the same for rows of 5..7 bytes (one 8-byte store)
#endif
// nxt: the first row word of the lane above (0 where that lane has no row), taken by the caller with every lane active
__device__ __forceinline__ void store_row_fused(uint8_t *row, uint32_t nbytes, const uint32_t (&w)[4], uint32_t nxt) {
    const uint32_t t = nbytes & 3u;  // (wave-uniform, 1..3) bytes of the row's last word
    const uint32_t keep = (1u << (8u * t)) - 1u;
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));
    typedef uint32_t u32x3u __attribute__((ext_vector_type(3), aligned(1)));
    if (nbytes > 12u) {
        u32x4u q = {w[0], w[1], w[2], (w[3] & keep) | (nxt << (8u * t))};
        if (PG_ROW_FUSE_NT) __builtin_nontemporal_store(q, reinterpret_cast<u32x4u *>(row));
        else *reinterpret_cast<u32x4u *>(row) = q;
    } else {
        u32x3u q = {w[0], w[1], (w[2] & keep) | (nxt << (8u * t))};
        if (PG_ROW_FUSE_NT) __builtin_nontemporal_store(q, reinterpret_cast<u32x3u *>(row));
        else *reinterpret_cast<u32x3u *>(row) = q;
    }
}
// follow a key's probe sequence through inline-layout lines in global memory (all six key words of a line in flight together)
__device__ __forceinline__ void lane_chase_inl(const SubTable &st, uint64_t key, uint32_t level, uint32_t b, uint32_t step, uint32_t (&w)[4]) {
    w[0] = w[1] = w[2] = w[3] = 0;
    for (uint64_t n = 0; n < st.nbuckets + GROUP_CHAIN; ++n) {
        const uint8_t *base = st.buckets + (uint64_t)b * 128u;
        const uint4 *line = reinterpret_cast<const uint4 *>(base);
        uint4 v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = line[c];
        const uint64_t kk[6] = {(uint64_t)v[0].x | ((uint64_t)v[0].y << 32), (uint64_t)v[0].z | ((uint64_t)v[0].w << 32),
                                (uint64_t)v[1].x | ((uint64_t)v[1].y << 32), (uint64_t)v[1].z | ((uint64_t)v[1].w << 32),
                                (uint64_t)v[2].x | ((uint64_t)v[2].y << 32), (uint64_t)v[2].z | ((uint64_t)v[2].w << 32)};
        uint32_t found = 0;
#pragma unroll
        for (int sl = 5; sl >= 0; --sl)
            if ((uint32_t)sl < st.slots && kk[sl] == key) found = (uint32_t)sl + 1u;
        if (found) {
            const uint32_t *mp = reinterpret_cast<const uint32_t *>(base + 8u * st.slots) + (found - 1u) * st.W;
#pragma unroll
            for (int i = 0; i < 4; ++i)  // (static indices: the words stay in registers)
                if ((uint32_t)i < st.W) w[i] = mp[i];
            return;
        }
        if ((st.slots >= 6u ? kk[5] : kk[4]) == EMPTY_KEY) return;  // line not full: absent
        level = min(level + 1, GROUP_CHAIN + 1);
        advance_line(key, level, st.nbuckets, b, step);
    }
}

// write the row bytes this sub-table owns: low nb0 bytes of m0 at column col0, low nb1 bytes of
// m1 at col0+4 (cpp/anchor.cpp:139-164).  ROWMODE 1: one-byte rows; 2: 8-byte rows (N = 64);
// 0: generic byte loop.
template <int ROWMODE>
__device__ __forceinline__ void store_row(uint8_t *row, uint32_t m0, uint32_t m1, const RowCols rc) {
    // Rows of 8 and of 16 bytes are written NON-TEMPORALLY: they are not read again before the statistics pass, and as
    // ordinary stores they push table lines out of the L2 (64 x 20 Mb k=21 8.1-8.4 -> 7.8 ms, k=31 7.22 -> 6.98,
    // 128 genomes 14.0-14.7 -> 13.1-13.9; profiles/r2_ab_nt_rows.txt).  Only where ONE store instruction covers the
    // row, so that a wave writes whole lines: one- and four-byte rows lose 3-5 % that way, rows of two 16-byte pieces
    // 25 % (256 genomes: 24.8 -> 32.2 ms — each piece leaves the L2 as a partial line).
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    if (ROWMODE == 1) {
        row[0] = (uint8_t)m0;
    } else if (ROWMODE == 4) {  // four-byte rows (25..32 genomes): one aligned dword
        if (PG_ROW4_NT) __builtin_nontemporal_store(m0, reinterpret_cast<uint32_t *>(row));
        else *reinterpret_cast<uint32_t *>(row) = m0;
    } else if (ROWMODE == 5) {  // two-byte rows (9..16 genomes)
        if (PG_ROW4_NT) __builtin_nontemporal_store((uint16_t)m0, reinterpret_cast<uint16_t *>(row));
        else *reinterpret_cast<uint16_t *>(row) = (uint16_t)m0;
    } else if (ROWMODE == 6) {  // three-byte rows (17..24 genomes): two stores at byte alignment
        struct __attribute__((packed)) U16 { uint16_t v; };
        reinterpret_cast<U16 *>(row)->v = (uint16_t)m0;
        row[2] = (uint8_t)(m0 >> 16);
    } else if (ROWMODE == 2 || rc.words == 4) {  // (rc.words == 4: both words at an 8-byte aligned column: one store)
        u32x2 q = {m0, m1};
        uint8_t *p = row + (ROWMODE == 2 ? 0u : rc.col0);
        if (PG_NT_ROWS) __builtin_nontemporal_store(q, reinterpret_cast<u32x2 *>(p));
        else *reinterpret_cast<u32x2 *>(p) = q;
    } else if  (rc.words == 1) {  // wave-uniform
        *reinterpret_cast<uint32_t *>(row + rc.col0) = m0;
        if (rc.nb1) *reinterpret_cast<uint32_t *>(row + rc.col0 + 4) = m1;
    } else if (rc.words == 2) {  // two-byte rows (9..16 genomes): one aligned 16-bit store
        *reinterpret_cast<uint16_t *>(row) = (uint16_t)m0;
    } else if (rc.words == 3) {  // rows of 3, 5, 6 or 7 bytes: two stores at byte alignment (gfx950 global
                                 // memory takes unaligned dword / short accesses), not one per byte
        struct __attribute__((packed)) U32 { uint32_t v; };
        struct __attribute__((packed)) U16 { uint16_t v; };
        const uint32_t nb = rc.nb0 + rc.nb1;  // wave-uniform
        if (nb == 3) {
            reinterpret_cast<U16 *>(row)->v = (uint16_t)m0;
            row[2] = (uint8_t)(m0 >> 16);
        } else {
            reinterpret_cast<U32 *>(row)->v = m0;
            if (nb == 5) row[4] = (uint8_t)m1;
            else if (nb == 6) reinterpret_cast<U16 *>(row + 4)->v = (uint16_t)m1;
            else reinterpret_cast<U32 *>(row + 3)->v = (m0 >> 24) | (m1 << 8);  // bytes 3..6 (byte 3 again)
        }
    } else {
        for (uint32_t bb = 0; bb < rc.nb0; ++bb) row[rc.col0 + bb] = (uint8_t)(m0 >> (8 * bb));
        for (uint32_t bb = 0; bb < rc.nb1; ++bb) row[rc.col0 + 4 + bb] = (uint8_t)(m1 >> (8 * bb));
    }
}

// value of lane-1 (lane 0 keeps its own): one DPP move (wave_shr:1) on the VALU instead of a
// ds_bpermute round trip through the LDS crossbar
__device__ __forceinline__ uint32_t lane_up1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// lanes whose value differs from that of the lane below, as a lane mask: an xor whose first source is the DPP shift (a fast
// VOP2 instruction; gfx950 has no DPP form of the compares) and a compare with zero — a move, the shift and a compare
// before.  Lane 0 has no lane below: its bit is arbitrary — callers OR the mask with one that has bit 0 set, or ignore
// lane 0.  (s_nop 1: a DPP source written by the VALU instruction before needs two wait states.)
__device__ __forceinline__ unsigned long long differs_from_lane_below(uint32_t v) {
    uint32_t t;
    asm("s_nop 1\n\tv_xor_b32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "=&v"(t) : "v"(v));
    return __builtin_amdgcn_ballot_w64(t != 0u);
}
// Sliding minimum over the last W lanes (lanes below W-1 see shorter windows): m <- min(own, m of the lane below),
// W-1 times, each step ONE instruction — the DPP shift is the min's own source modifier; lane 0, which has no lane
// below, is left alone and keeps its m.  6 instructions for W = 7 where shuffle-doubling took 6 moves + 3 min +
// 3 selects.  (s_nop 1: a DPP source written by the previous VALU instruction needs two wait states; one asm
// block, so that the compiler does not pad every step once more)
#define PG_DPPMIN "s_nop 1\n\tv_min_u32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
template <int W>
__device__ __forceinline__ uint32_t sliding_min(uint32_t x) {
    uint32_t m = x;
    if constexpr (W == 2) asm(PG_DPPMIN : "+v"(m) : "v"(x));
    if constexpr (W == 3) asm(PG_DPPMIN PG_DPPMIN : "+v"(m) : "v"(x));
    if constexpr (W == 4) asm(PG_DPPMIN PG_DPPMIN PG_DPPMIN : "+v"(m) : "v"(x));
    if constexpr (W == 5) asm(PG_DPPMIN PG_DPPMIN PG_DPPMIN PG_DPPMIN : "+v"(m) : "v"(x));
    if constexpr (W == 6) asm(PG_DPPMIN PG_DPPMIN PG_DPPMIN PG_DPPMIN PG_DPPMIN : "+v"(m) : "v"(x));
    if constexpr (W == 7) asm(PG_DPPMIN PG_DPPMIN PG_DPPMIN PG_DPPMIN PG_DPPMIN PG_DPPMIN : "+v"(m) : "v"(x));
    if constexpr (W == 8) asm(PG_DPPMIN PG_DPPMIN PG_DPPMIN PG_DPPMIN PG_DPPMIN PG_DPPMIN PG_DPPMIN : "+v"(m) : "v"(x));
    static_assert(W >= 1 && W <= 8, "minimizer window");
    return m;
}
#undef PG_DPPMIN
// ---- batches without halo lanes (PG_PROBE_CARRY) -----------------------------------------------------------------
// A position's minimizer is the minimum over the W m-mers of its k-mer: its own (the last one) and the W - 1 before it,
// which the W - 1 lanes below own.  The first W - 1 lanes of a batch have too few lanes below them; round 1-4 made them
// HALO lanes — they only supplied m-mers, and a batch brought 64 - (W - 1) new positions (57 of 64 lanes at W = 7).
// Instead (van Herk's two halves of a window): the window of lane j < W - 1 = the last W - 1 - j m-mers before the batch
// (a suffix minimum over them) and the first j + 1 of this one (what the sliding minimum yields for a lane with fewer
// than W - 1 lanes below).  One ds_bpermute brings the W - 1 m-mer ranks before the NEXT batch's first position — they
// sit in consecutive lanes of this one — down to lanes 0 .. W - 2 (the carry; the other lanes hold ~0 once it is
// used); their suffix minima take three row-local DPP steps there (a window of eight, cut off where the ~0 begin),
// interleaved with the next batch's own sliding minimum, which takes them in with one v_min: up to 64 new positions
// per batch for about eight instructions.
#ifndef PG_RID_ADDC
#define PG_RID_ADDC 1  // k_probe's front end: run ids by mbcnt + v_addc_co_u32 (run_ids), no test for a batch without runs
#endif
#ifndef PG_SCAN_FLAT
#define PG_SCAN_FLAT 0  // 1: k_probe's slot scan without control flow (scan_line_lds_flat) — 11 scalar instructions and three branches fewer per batch, parity green, and no faster: not the default (profiles/r4e_ab_probe_scalar_cuts.txt)
#endif
#ifndef PG_RUNLESS_TESTS
#define PG_RUNLESS_TESTS 0  // 1: k_probe's batches without runs skip fetch, staging and scan (rounds 1-4; see back())
#endif
#ifndef PG_PROBE_CARRY
#define PG_PROBE_CARRY 1
#endif
// -DPG_PRIO=0xABCD: wave priorities (s_setprio) by phase of a batch, an experiment — A from the wait for the batch's lines on
// (staging), B around the issue of the next fetch, C for slot scan / overflow entries / row store, D for the front end of
// the batch after next; 0 = no s_setprio at all (the product; profiles/r4e_ab_setprio.txt)
#ifndef PG_PRIO
#define PG_PRIO 0
#endif
#if PG_PRIO
#define PG_PRIO_AT(i) __builtin_amdgcn_s_setprio((PG_PRIO >> (4 * (4 - (i)))) & 3);
#else
#define PG_PRIO_AT(i)
#endif
#ifndef PG_PROBE_CUT
#define PG_PROBE_CUT 1
#endif
#ifndef PG_EARLY_LINES
#define PG_EARLY_LINES 1  // the first staging step's line numbers pass through LDS at the end of the front end (k_probe: front)
#endif
// r[l] = min(x[l .. min(l + 7, last lane of l's row of 16)])
__device__ __forceinline__ uint32_t suffix_min8_row(uint32_t x) {
    uint32_t r = x;
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf"
        : "+v"(r));
    return r;
}
// the sliding minimum of x AND, in place, the suffix minima of a second vector (the carry): two dependent chains
// interleaved (a DPP source written by a VALU instruction needs two wait states: the other chain's step is one of them)
// (the sliding step as min(m, m of the lane below) — a window of t + 1 from two windows of t — needs no third operand: the
// vector itself can then be the register the suffix chain works in; an input operand equal to a read-write one may be
// given the SAME register by the compiler, and was)
#define PG_F "v_min_u32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define PG_S(n) "v_min_u32_dpp %1, %1, %1 row_shl:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define PG_N0 "s_nop 0\n\t"
#define PG_N1 "s_nop 1\n\t"
template <int W>
__device__ __forceinline__ uint32_t sliding_min_suffix(uint32_t x, uint32_t &suffix) {
    uint32_t m = x, r = suffix;
    static_assert(W >= 2 && W <= 8, "minimizer window");
    if constexpr (W == 2) asm(PG_N1 PG_F PG_S(1) PG_N1 PG_S(2) PG_N1 PG_S(4) : "+v"(m), "+v"(r));
    if constexpr (W == 3) asm(PG_N1 PG_F PG_S(1) PG_N0 PG_F PG_S(2) PG_N1 PG_S(4) : "+v"(m), "+v"(r));
    if constexpr (W == 4) asm(PG_N1 PG_F PG_S(1) PG_N0 PG_F PG_S(2) PG_N0 PG_F PG_S(4) : "+v"(m), "+v"(r));
    if constexpr (W == 5) asm(PG_N1 PG_F PG_S(1) PG_N0 PG_F PG_S(2) PG_N0 PG_F PG_S(4) PG_N0 PG_F : "+v"(m), "+v"(r));
    if constexpr (W == 6) asm(PG_N1 PG_F PG_S(1) PG_N0 PG_F PG_S(2) PG_N0 PG_F PG_S(4) PG_N0 PG_F PG_N1 PG_F : "+v"(m), "+v"(r));
    if constexpr (W == 7) asm(PG_N1 PG_F PG_S(1) PG_N0 PG_F PG_S(2) PG_N0 PG_F PG_S(4) PG_N0 PG_F PG_N1 PG_F PG_N1 PG_F : "+v"(m), "+v"(r));
    if constexpr (W == 8) asm(PG_N1 PG_F PG_S(1) PG_N0 PG_F PG_S(2) PG_N0 PG_F PG_S(4) PG_N0 PG_F PG_N1 PG_F PG_N1 PG_F PG_N1 PG_F : "+v"(m), "+v"(r));
    suffix = r;
    return m;
}
#undef PG_F
#undef PG_S
#undef PG_N0
#undef PG_N1
// base + (set bits of `mask` at lanes <= this lane) - 1: for a lane whose own bit is set, its index among the set lanes
// (a queue slot, counted from the wave-uniform `base`); for any lane, the number of the last set lane at or below it (a
// run id, when the set lanes are the runs' first lanes).  Two VALU instructions: the shift of the mask by one lane —
// which turns mbcnt's "below this lane" into "at or below" — and the constant are scalar work, and the constant rides in
// on mbcnt's own addend.
__device__ __forceinline__ uint32_t lanes_le_index(unsigned long long mask, uint32_t base) {
    const unsigned long long m1 = mask >> 1;
    const uint32_t c = base + (uint32_t)(mask & 1ull) - 1u;
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, c));
}

// run ids: (set lanes of `lmask` at or below this lane) - 1 — lanes_le_index(lmask, 0) — with NO scalar instruction: mbcnt counts
// the set lanes BELOW (its addend -1 is an inline constant), and a lane's own bit comes in as the carry of v_addc_co_u32,
// whose carry-in operand is a lane mask in an SGPR pair
__device__ __forceinline__ uint32_t run_ids(unsigned long long lmask) {
#if PG_RID_ADDC
    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(lmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lmask, ~0u));
    uint32_t rid;
    unsigned long long carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(rid), "=s"(carry_out) : "v"(below), "s"(lmask));
    return rid;
#else
    return lanes_le_index(lmask, 0u);
#endif
}
// for a lane whose own bit of `mask` is set: base + its index among the set lanes = base + the set lanes BELOW it, which is
// what mbcnt counts as it is — none of lanes_le_index's scalar shifts (a scalar instruction costs k_probe's launch what a
// vector one does: profiles/r4e_probe_dummy_salu.txt)
__device__ __forceinline__ uint32_t set_lane_index(unsigned long long mask, uint32_t base) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, base));
}

// ROWMODE 3 — the genome-sharded mode's narrow tables (a block of up to 8 genomes): the probe emits the block's
// COMPACT BIT COLUMNS directly — per tile TILE_SLOTS x u64 per genome, bit l of word s = position 64 s + l — instead of one-byte
// rows that k_cols_extract would read back: the tile's columns are assembled in LDS (one ballot per genome and batch,
// shifted to the batch's place by the scalar unit; positions resolved by the overflow levels OR their bits in) and
// written once, 64 bytes per genome and tile.
constexpr int COLS_G = 8;  // genomes per block in columns mode
constexpr uint32_t TILE_SLOTS = PROBE_TILE / 64;  // u64 column words per genome and tile (a slot = 64 positions)
static_assert(PROBE_TILE % 512 == 0, "the column kernels take a tile in units of 512 positions");
__device__ __forceinline__ void cols_or_position(unsigned long long *cols, uint32_t pl, uint32_t m0) {
    while (m0) {  // (rare path: a lane per resolved position, an LDS atomic per set bit)
        const uint32_t j = (uint32_t)__ffs((int)m0) - 1u;
        m0 &= m0 - 1u;
        atomicOr(&cols[(pl >> 6) * COLS_G + j], 1ull << (pl & 63u));
    }
}

// Overflow levels of a tile: dense 64-entry batches out of the wave's LDS queue, staged exactly
// like the main batches (neighbouring entries belong to the same group and share their next
// line); entries that overflow again are compacted in place for the next level.
// LEVELS: overflow levels worked off with staged lines; entries still unresolved after them chase their sequences lane
// by lane.  Two for many-genome tables (narrow windows, big groups: a third of the overflowing keys overflow again); ONE
// for the wide-window tables of up to 16 genomes, where the second staged level costs more than the few lanes it saves
// (+1.5 % at configs 1-2, tools/ab_one.sh; -1 % at 64 genomes).
template <bool TWO, int ROWMODE, int SLOTS, int MAXRUN, bool WIDE, int LEVELS, bool INL = false>
__device__ __forceinline__ void drain_queue(const SubTable &st, uint32_t qn, const uint64_t *sw, const uint64_t *rw, uint32_t *q_line,
                                            uint32_t *q_step, uint16_t *q_pl, uint32_t *lines_w, uint4 *buf,
                                            uint8_t *tile_rows, uint32_t nbytes, const RowCols rc, int lane) {
    // a queue entry is (position in the tile, next line to try, step of its sequence): the key is
    // taken from the tile's sequence words again — 10 bytes of LDS per entry instead of 18 let the
    // kernel run 32 waves per CU instead of 26
    const int k = (int)st.k;
    constexpr int LDS_LINE_U4 = SLOTS + 1;
    constexpr int STAGE_ITERS = (MAXRUN * SLOTS + 63) / 64;
    const uint8_t *chunk_base = st.buckets + (uint32_t)(lane % SLOTS) * 16u;
    const uint32_t lbytes = INL ? st.layout * 64u : st.slots * (WIDE ? 8u : 16u);  // (= 16 * SLOTS = 128, as a run-time scalar: see k_probe)
    static_assert(LEVELS >= 1, "k_probe's entries (home line, group) are turned into (next line, step) by level 1's staged batches");
    for (int level = 1; qn > 0; ++level) {
        PG_WSYNC();
        if (level > LEVELS) {
            // the few entries still unresolved (long chains) walk their sequences lane by lane:
            // 8 slot loads in flight per line, no staging overhead
            for (uint32_t e = lane; e < qn; e += 64) {
                uint32_t m0, m1;
                const uint64_t key = canonical_from_le(extract_bases(sw, q_pl[e]), k);
                if constexpr (WIDE && INL) {
                    uint32_t rw4[4];
                    lane_chase_inl(st, key, (uint32_t)level, q_line[e], q_step[e], rw4);
                    if (rw4[0] | rw4[1] | rw4[2] | rw4[3]) store_row_regs(tile_rows + (uint64_t)q_pl[e] * nbytes, nbytes, rw4);
                } else if constexpr (WIDE) {
                    lane_chase_wide(st, key, (uint32_t)level, q_line[e], q_step[e], m0, m1);
                    if (m1) store_row_wide(st.masks, st.W, nbytes, tile_rows + (uint64_t)q_pl[e] * nbytes, m0, m1);
                } else {
                    lane_chase<TWO, SLOTS>(st, key, (uint32_t)level, q_line[e], q_step[e], m0, m1);
                    if constexpr (ROWMODE == 3) cols_or_position(reinterpret_cast<unsigned long long *>(tile_rows), q_pl[e], m0);
                    else if (m0 | m1) store_row<ROWMODE>(tile_rows + (uint64_t)q_pl[e] * nbytes, m0, m1, rc);
                }
            }
            break;
        }
        uint32_t kept = 0;
        for (uint32_t i0 = 0; i0 < qn; i0 += 64) {
            const uint32_t e = i0 + lane;
            const bool act = e < qn;
            const uint32_t ec = act ? e : qn - 1;
            uint32_t line = q_line[ec], step = q_step[ec];
            if (GROUP_CHAIN > 1 && level == 1) {  // (k_probe's entries: home line and group — see there)
                step = step_of_group(step, st.nbuckets);
                line = next_line(line, step, st.nbuckets);
            }
            const uint32_t pl = q_pl[ec];
            const uint64_t kmask = kmer_mask(k);
            const uint64_t X = extract_bases32(reinterpret_cast<const uint32_t *>(sw), pl) & kmask;
            const uint64_t key = canonical_from_xb(X, revcomp_window(rw, X, pl, k, kmask, PROBE_SEQ_BASES - (uint32_t)k), k);
            const uint32_t prev_line = lane_up1(line);
            const bool leader = act && (lane == 0 || line != prev_line);
            const unsigned long long lmask = __builtin_amdgcn_ballot_w64(leader);
            const uint32_t rid = lanes_le_index(lmask, 0u);
            const uint32_t nruns = (uint32_t)__popcll(lmask);
            uint32_t m0 = 0, m1 = 0;
            [[maybe_unused]] uint32_t rw4[4] = {0, 0, 0, 0};  // (inline layout: the row's words)
            int rcode = 0;
            // (staging as in k_probe's main batches: the entries behind the last run repeat the first run's line, every lane
            // reads its entries at fixed places, one multiply-add per address)
            const uint32_t padline = lmask ? (uint32_t)__builtin_amdgcn_readlane((int)line, __builtin_ctzll(lmask)) : 0u;
            for (uint32_t r0 = 0; r0 < nruns; r0 += MAXRUN) {
                const uint32_t nl = min((uint32_t)MAXRUN, nruns - r0);
                if (lane < MAXRUN) lines_w[lane] = padline;
                if (leader && rid - r0 < nl) lines_w[rid - r0] = line;
                PG_WSYNC();
                uint4 v[STAGE_ITERS];
                uint32_t ln[STAGE_ITERS];
#pragma unroll
                for (int u = 0; u < STAGE_ITERS; ++u) ln[u] = lines_w[u * (64 / SLOTS) + lane / SLOTS];
#pragma unroll
                for (int u = 0; u < STAGE_ITERS; ++u) v[u] = *reinterpret_cast<const uint4 *>(chunk_base + (uint64_t)ln[u] * lbytes);
#pragma unroll
                for (int u = 0; u < STAGE_ITERS; ++u) {
                    const uint32_t idx = u * 64 + lane;
                    if constexpr (WIDE) buf[(idx / SLOTS) * LDS_LINE_U4 + (idx % SLOTS)] = v[u];  // (bare keys: as they are)
                    else stage_chunk(buf + (idx / SLOTS) * LDS_LINE_U4, idx % SLOTS, SLOTS, v[u]);
                }
                PG_WSYNC();
                if (act && rid - r0 < nl) {
                    if constexpr (WIDE && INL) {
                        rcode = scan_keys_inl(buf + (rid - r0) * LDS_LINE_U4, key, st.slots, m1);
                        if (m1) masks_inl(buf + (rid - r0) * LDS_LINE_U4, st.slots, st.W, m1 - 1u, rw4);
                    } else if constexpr (WIDE) {
                        rcode = scan_keys16_lds(buf + (rid - r0) * LDS_LINE_U4, key, m1);
                        m0 = line;
                    } else {
                        rcode = scan_line_lds<TWO, SLOTS>(buf + (rid - r0) * LDS_LINE_U4, key, m0, m1);
                    }
                }
                PG_WSYNC();
            }
            const bool again = act && rcode < 0;
            if constexpr (WIDE && INL) {
                if (act && m1) store_row_regs(tile_rows + (uint64_t)pl * nbytes, nbytes, rw4);
            } else if constexpr (WIDE) {
                if (act && m1) store_row_wide(st.masks, st.W, nbytes, tile_rows + (uint64_t)pl * nbytes, m0, m1);
            } else {
                if constexpr (ROWMODE == 3) {
                    if (act) cols_or_position(reinterpret_cast<unsigned long long *>(tile_rows), pl, m0);
                } else {
                    if (act && (m0 | m1)) store_row<ROWMODE>(tile_rows + (uint64_t)pl * nbytes, m0, m1, rc);
                }
            }
            const unsigned long long kmask2 = __builtin_amdgcn_ballot_w64(again);
            if (again) {  // in-place compaction: slot <= e, and this batch's reads are already done
                const uint32_t slot = set_lane_index(kmask2, kept);
                uint32_t nl2 = line, ns2 = step;
                // (staged levels end before the group's chain does: no switch to the key's own sequence
                // here, and none of its hashing on this path)
                if constexpr (LEVELS + 1 < (int)GROUP_CHAIN) nl2 = next_line(line, step, st.nbuckets);
                else advance_line(key, (uint32_t)level + 1, st.nbuckets, nl2, ns2);
                q_line[slot] = nl2;
                q_step[slot] = ns2;
                q_pl[slot] = (uint16_t)pl;
            }
            kept += (uint32_t)__popcll(kmask2);
            PG_WSYNC();
        }
        qn = kept;
    }
}

// ---------------------------------------------------------------------------
// Fused statistics (round 6; FuseArgs in pg_kernels.h): the END of a tile inside k_probe.  The statistics pass read every row
// again from HBM — a quarter of the step at 65-128 genomes, 102 GB at BASELINE configs[3] — although each wave has just
// written its tile's rows (at most 16 KB): here the wave reads them back while they are still in the L2 / Infinity Cache and
// leaves the tile's counters, 0.4-5 % of the row bytes, for k_tile_reduce to add up.  Nothing of this touches the batch loop:
// no register, no LDS byte and no instruction of it (the tile's descriptors are loaded again at the end, so that the loop
// keeps no more values alive than the unfused kernel's).  Extra VALU work at the tile's end is cheap where it matters: 48
// extra instructions per batch cost the probe of 65 / 128 genomes 0.6 %, of 27 / 64 genomes 9 % (profiles/r6b_elasticity.txt).
//   rows        16 per lane (row j * 64 + lane of the tile, j = 0..15), each ONE load of ceil(nbytes / 4) dwords at its byte
//               offset (gfx950 takes unaligned dword accesses), L1 bypassed (sc1: the L2 is where the wave's stores are)
//   histogram   popcount -> (bin of the row relative to the tile's first, popcount): one LDS add of 1 << 16 * (index & 1) to a
//               packed pair of u16 counters, four copies by lane & 3 (a tile has 1024 rows: no field overflows)
//   column sums sixteen rows per word through a Harley-Seal carry-save tree (15 adders: 45 instructions) into five bit
//               planes, transposed to byte counters and summed over the lanes by a halving exchange (k_epilogue's vflush)
//   bitmap.100  the tile's rows at multiples of 100 (eleven at most), copied byte by byte by the first lanes
// The reference does all of this in its scatter loop (cpp/anchor.cpp:156-183); the column sums are index.py:1051's.
// ---------------------------------------------------------------------------
#ifndef PG_FUSE_LOAD_SC1
#define PG_FUSE_LOAD_SC1 1  // 0: the row read-back with the default cache policy (experiment)
#endif
// NW (1..4) dwords from byte `off` of the tile's rows (any byte alignment: gfx950 takes unaligned dword accesses), as agent-scope
// loads of 8 and 4 bytes — sc1: served by the L2, where the wave's stores are, whatever the L1 holds — with the tile's base
// address in a scalar register pair.  (A buffer descriptor would do the same in one instruction for three or four words, but its
// four scalar registers do not survive in the wide instantiations, which are held to 80: the compiler then keeps it in vector
// registers and wraps every load in a readfirstlane loop.)
typedef const __attribute__((address_space(1))) uint8_t *GlobalBytes;  // (a pointer the compiler knows to be global memory: global_load with a scalar base, not flat_load)
template <int NW>
__device__ __forceinline__ void fuse_load_row(GlobalBytes base, uint32_t off, uint32_t (&w)[NW]) {
    GlobalBytes p = base + off;
#if PG_FUSE_LOAD_SC1
#pragma unroll
    for (int i = 0; i + 1 < NW; i += 2) {
        const unsigned long long v = __hip_atomic_load(reinterpret_cast<const __attribute__((address_space(1))) unsigned long long *>(p + 4 * i), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        w[i] = (uint32_t)v, w[i + 1] = (uint32_t)(v >> 32);
    }
    if constexpr (NW & 1)
        w[NW - 1] = __hip_atomic_load(reinterpret_cast<const __attribute__((address_space(1))) uint32_t *>(p + 4 * (NW - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    const WordsN<NW> v = *reinterpret_cast<const __attribute__((address_space(1))) WordsN<NW> *>(p);
#pragma unroll
    for (int i = 0; i < NW; ++i) w[i] = v.w[i];
#endif
}
// the five bit planes of one row word (bit c of plane i = bit i of the number of the lane's 16 rows that hold column c) -> the
// tile's 32 column counters of that word, two u16 per output word: entry e (0..15) = columns e and e + 16
__device__ __forceinline__ void fuse_flush_word(const uint32_t (&pl)[5], uint32_t *out16, int lane) {
    uint32_t R[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // byte b of v counts column 8 b + q (16 at most)
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) v |= ((pl[i] >> q) & 0x01010101u) << i;
        R[2 * q] = v & 0x00FF00FFu;
        R[2 * q + 1] = (v >> 8) & 0x00FF00FFu;
    }
    // 16 registers x 64 lanes -> one register per group of four lanes: a lane pair exchanges halves of its register file four times
    // (literal bounds: with the halving written as a loop the compiler indexes R dynamically — a chain of 16 selects per access)
#define PG_XCHG(HALF, BIT)                                                  \
    {                                                                       \
        const bool up = (lane & BIT) != 0;                                  \
        _Pragma("unroll") for (int i = 0; i < HALF; ++i) {                  \
            const uint32_t send = up ? R[i] : R[i + HALF];                  \
            const uint32_t keep = up ? R[i + HALF] : R[i];                  \
            R[i] = keep + (uint32_t)__shfl_xor((int)send, BIT);             \
        }                                                                   \
    }
    PG_XCHG(8, 32)
    PG_XCHG(4, 16)
    PG_XCHG(2, 8)
    PG_XCHG(1, 4)
#undef PG_XCHG
    R[0] += (uint32_t)__shfl_xor((int)R[0], 2);
    R[0] += (uint32_t)__shfl_xor((int)R[0], 1);
    if ((lane & 3) == 0) {  // this lane holds register lane >> 2 = 2 q + odd: columns q + 8 odd (low half) and + 16 (high half)
        const uint32_t idx = (uint32_t)lane >> 2;
        out16[(idx >> 1) + ((idx & 1u) ? 8u : 0u)] = R[0];
    }
}
// One pass over the tile's rows for NWP (1 or 2) of their words, from byte `woff` of the row on: the lane's 16 rows in four
// groups of four (two groups' loads in flight), their popcounts added to `pcs` (one byte per row, four rows per register: a row
// of three or four words takes two passes — the registers of ONE pass of two words are what fits beside the 64 of eight waves per
// SIMD) and, in the row's LAST pass, counted into the histogram; the words' column sums through the carry-save tree and out.
template <int NWP, bool LASTPASS>
__device__ __forceinline__ void tile_statistics_pass(uint32_t *Hc, GlobalBytes rs, uint32_t npos, uint32_t nbytes, uint32_t woff,
                                                     uint32_t keep_last, uint32_t split, uint32_t N, uint32_t (&pcs)[4], uint32_t *cs_out, int lane) {
    const uint32_t N1 = N + 1u;
    uint32_t ones[NWP], twos[NWP], fours[NWP], eights[NWP], sixteens[NWP], hold4[NWP], hold8[NWP];
#pragma unroll
    for (int w = 0; w < NWP; ++w) ones[w] = twos[w] = fours[w] = eights[w] = sixteens[w] = hold4[w] = hold8[w] = 0;
    uint32_t nx[4][NWP];
    auto request = [&](int g, uint32_t (&dst)[4][NWP]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t r = (uint32_t)(4 * g + i) * 64u + (uint32_t)lane;
            fuse_load_row<NWP>(rs, min(r, npos - 1u) * nbytes + woff, dst[i]);  // (rows behind the tile's last: that one again, masked out below)
        }
    };
    request(0, nx);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint32_t rw[4][NWP];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int w = 0; w < NWP; ++w) rw[i][w] = nx[i][w];
        if (g < 3) request(g + 1, nx);
        uint32_t pc4 = pcs[g];  // the four rows' popcounts so far, a byte each
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t r = (uint32_t)(4 * g + i) * 64u + (uint32_t)lane;
            // (validity and the bin as arithmetic, not as lane masks: sixteen rows' masks alive at once are 32 scalar registers the
            // wide instantiations do not have)
            const uint32_t vmask = (uint32_t)((int32_t)(r - npos) >> 31);  // all ones for a row of the tile (r < npos <= 1024)
            if (LASTPASS) rw[i][NWP - 1] &= keep_last;  // (the row's last word holds bytes of the next row)
            uint32_t pc = 0;
#pragma unroll
            for (int w = 0; w < NWP; ++w) {
                rw[i][w] &= vmask;
                pc += (uint32_t)__popc(rw[i][w]);
            }
            if (LASTPASS) {
                pc = min(pc + ((pc4 >> (8 * i)) & 0xFFu), N);  // (junk bits beyond ngenomes: the reference indexes out of bounds there; our rows have none)
                const uint32_t idx = pc + (N1 & (uint32_t)((int32_t)(split - 1u - r) >> 31));  // + N + 1 for a row of the tile's second bin (r >= split)
                atomicAdd(&Hc[idx], vmask & 1u);
            } else {
                pc4 += pc << (8 * i);  // (at most 64 per pass)
            }
        }
        if (!LASTPASS) pcs[g] = pc4;
#pragma unroll
        for (int w = 0; w < NWP; ++w) {  // four rows into the planes: three carry-save adders, the carries of weight 4 and 8 held back in turn
            uint32_t u = ones[w] ^ rw[0][w];
            const uint32_t c2a = (ones[w] & rw[0][w]) | (u & rw[1][w]);
            ones[w] = u ^ rw[1][w];
            u = ones[w] ^ rw[2][w];
            const uint32_t c2b = (ones[w] & rw[2][w]) | (u & rw[3][w]);
            ones[w] = u ^ rw[3][w];
            u = twos[w] ^ c2a;
            const uint32_t c4 = (twos[w] & c2a) | (u & c2b);
            twos[w] = u ^ c2b;
            if ((g & 1) == 0) {
                hold4[w] = c4;
            } else {
                u = fours[w] ^ hold4[w];
                const uint32_t c8 = (fours[w] & hold4[w]) | (u & c4);
                fours[w] = u ^ c4;
                if (g == 1) {
                    hold8[w] = c8;
                } else {
                    u = eights[w] ^ hold8[w];
                    sixteens[w] = (eights[w] & hold8[w]) | (u & c8);
                    eights[w] = u ^ c8;
                }
            }
        }
    }
#pragma unroll
    for (int w = 0; w < NWP; ++w) {
        const uint32_t pl[5] = {ones[w], twos[w], fours[w], eights[w], sixteens[w]};
        fuse_flush_word(pl, cs_out + 16 * w, lane);
    }
}

template <int NW>
__device__ __forceinline__ void tile_statistics(uint32_t *H, const uint8_t *tile_rows, uint32_t npos, uint32_t nbytes, const AnchorDesc &a,
                                                uint32_t tile_start, uint32_t tile, const FuseArgs &fo, int lane) {
    const uint32_t N = fo.ngenomes, N1 = N + 1u, hw = fo.hw;
    const uint32_t nh = 2u * N1;  // histogram counters of a tile: (bin 0 / 1) x (popcount 0..N), four copies of them in LDS
#ifdef PG_PHASE_TIMING  // (slots 11..15 of the phase counters: wait for the stores, zero + 1-in-100 request, rows (first pass), rows (second pass) + 1-in-100 stores, histogram out)
    uint32_t tph[5] = {0, 0, 0, 0, 0}, tph_t = (uint32_t)__builtin_readcyclecounter();
#define PG_TPH(i) { const uint32_t n_ = (uint32_t)__builtin_readcyclecounter(); tph[i] += n_ - tph_t; tph_t = n_; }
#else
#define PG_TPH(i)
#endif
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): every row store of this wave has been acknowledged by the L2
    __syncthreads();                     // (the probe's LDS buffers are dead from here on)
    PG_TPH(0)
    // (the base through readfirstlane: the tile's address is wave-uniform, but the compiler cannot see it)
    const uint64_t base = reinterpret_cast<uint64_t>(tile_rows);
    GlobalBytes rs = reinterpret_cast<GlobalBytes>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
                                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base));
    // bitmap.100: rows whose position in the contig is a multiple of 100 (cpp/anchor.cpp:169-177), eleven at most: requested now,
    // stored behind the rows' passes (a round trip to the L2 that nothing waits for)
    const uint32_t r100 = (100u - tile_start % 100u) % 100u + 100u * (uint32_t)lane;
    const bool want100 = fo.out100 != nullptr && r100 < npos;
    uint32_t w100[NW];
    fuse_load_row<NW>(rs, min(r100, npos - 1u) * nbytes, w100);
    for (uint32_t i = lane; i < 4u * nh; i += 64) H[i] = 0;
    // rows [0, split) of the tile lie in its first bin, the others in the next one (bins are at least a tile long)
    const uint64_t bin_end = ((uint64_t)(tile_start / a.binlen) + 1u) * a.binlen;
    const uint32_t split = (uint32_t)min((uint64_t)npos, bin_end - tile_start);
    const uint32_t keep = (nbytes & 3u) ? ((1u << (8u * (nbytes & 3u))) - 1u) : ~0u;  // bytes of the row in its last word
    uint32_t *const Hc = H + ((uint32_t)lane & 3u) * nh;
    uint32_t *cs_out = fo.tile_cs + (uint64_t)tile * fo.csw;
    uint32_t pcs[4] = {0, 0, 0, 0};
    __syncthreads();
    PG_TPH(1)
    if constexpr (NW <= 2) {
        tile_statistics_pass<NW, true>(Hc, rs, npos, nbytes, 0u, keep, split, N, pcs, cs_out, lane);
        PG_TPH(2)
    } else {
        tile_statistics_pass<2, false>(Hc, rs, npos, nbytes, 0u, ~0u, split, N, pcs, cs_out, lane);
        PG_TPH(2)
        tile_statistics_pass<NW - 2, true>(Hc, rs, npos, nbytes, 8u, keep, split, N, pcs, cs_out + 32, lane);
    }
    if (want100) {
        uint8_t *dst = fo.out100 + a.out100_off + (uint64_t)((tile_start + r100) / 100u) * nbytes;
#pragma unroll
        for (int i = 0; i < NW; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if ((uint32_t)(4 * i + b) < nbytes) dst[4 * i + b] = (uint8_t)(w100[i] >> (8 * b));
    }
    __syncthreads();
    PG_TPH(3)
    uint32_t *th = fo.tile_hist + (uint64_t)tile * hw;
    for (uint32_t i = lane; i < hw; i += 64) {  // two counters per word (a tile has 1024 rows: a u16 holds any count)
        const uint32_t lo = H[2u * i] + H[nh + 2u * i] + H[2u * nh + 2u * i] + H[3u * nh + 2u * i];
        const uint32_t hi = H[2u * i + 1u] + H[nh + 2u * i + 1u] + H[2u * nh + 2u * i + 1u] + H[3u * nh + 2u * i + 1u];
        th[i] = lo | (hi << 16);
    }
    PG_TPH(4)
#ifdef PG_PHASE_TIMING
    if (lane == 0)
        for (int i = 0; i < 5; ++i) atomicAdd(&pg_phase_cycles[(blockIdx.x & 1023u) * 16u + 11 + i], (unsigned long long)tph[i]);
#endif
#undef PG_TPH
}

// M64: m-mers longer than 16 bases.  WIDE: split layout (SLOTS = 8 then counts the 16-byte chunks of a key line: the
// staging geometry is the same, a line holds 16 bare keys; m0 / m1 carry the hit's line and slot + 1)
// (probe_pipelined: the instantiations that run the skewed batch order, see the end of the kernel; they are held to the 64
// registers of 8 waves per SIMD — the only spill that costs them sits around the queue-full call, a cold path)
#ifndef PG_INL_WAVES8
#define PG_INL_WAVES8 1  // the inline-layout probe held to the 64 registers of 8 waves per SIMD
#endif
#ifndef PG_PIPE_MORE
#define PG_PIPE_MORE 5  // which further instantiations run the skewed order: bit 0 two-word slots, 1 split layout, 2 the 6-m-mer window, 3 generic rows
#endif
#ifndef PG_INL_PIPE
#define PG_INL_PIPE 0  // 1: the inline-layout probe in the skewed batch order too (it needs 43 registers in the plain order)
#endif
template <int W_C, bool TWO, int ROWMODE, int SLOTS, bool WIDE, bool INL = false>
constexpr bool probe_pipelined = PG_PROBE_PIPE && SLOTS == 8 && (ROWMODE != 3) && ((ROWMODE == 1 || ROWMODE >= 4) || (PG_PIPE_MORE & 8) || TWO || WIDE) &&
                                 (!TWO || (PG_PIPE_MORE & 1)) && (!WIDE || (PG_PIPE_MORE & 2) || (INL && PG_INL_PIPE)) && (W_C != 6 || (PG_PIPE_MORE & 4));
// The scalar registers count too: a SIMD's 800 SGPRs admit floor(800 / (ceil(sgpr / 16) * 16 + 16)) waves — 8 up to 80, 7 up to
// 96, 6 up to 112 (MI355X_MICROARCH.md) — whatever the compiler's own occupancy figure says.  Left alone the generic-row and
// split-layout instantiations took 92 and 105 (their lane masks live on the scalar unit): 7 and 6 waves.  Held to 80, a dozen
// masks move to vector lanes and the ninth to 63rd genome gain 3-6 % (27 x 40 Mb 140 -> 148 G k-mers/s).
// FUSE: the tile ends with its statistics (tile_statistics above; `fo` says where they go)
template <int W_C, bool TWO, int ROWMODE, int SLOTS, bool M64, bool WIDE = false, bool INL = false, bool FUSE = false>
__global__ __launch_bounds__(64, ((probe_pipelined<W_C, TWO, ROWMODE, SLOTS, WIDE, INL> || (INL && PG_INL_WAVES8) || FUSE) ? 8 : 1))
__attribute__((amdgpu_num_sgpr(80))) void k_probe(const SubTable st, const uint64_t *__restrict__ seqw,
                                              const uint32_t *__restrict__ nmw, const uint32_t *__restrict__ has_n,
                                              const SeqDesc *__restrict__ sd, const AnchorDesc *__restrict__ ad,
                                              const uint32_t *__restrict__ tile_contig,
                                              const uint32_t *__restrict__ sched, uint32_t tile_base,
                                              uint8_t *__restrict__ out1, uint32_t nbytes, const RowCols rc, const FuseArgs fo) {
    // A staged line occupies 16*SLOTS + 16 bytes of LDS: the pad keeps the lanes' ds_read_b128 of
    // "their" lines off a common bank group (a power-of-two stride would be a 32-way conflict)
    constexpr int LDS_LINE_U4 = SLOTS + 1;
    // lines staged per step: 16, whatever the window — a batch ends in front of its 17th run (CUT below).  (While a batch
    // was a fixed 58 - 61 positions, the 6-m-mer window — what k=21 gets on 150-570 Mb genomes, ~17 lines per batch — staged
    // 24 to avoid a second step; with batches cut at 16 runs the smaller LDS footprint wins: 8 x 200 Mb 8.0 -> 7.6 ms,
    // 27 x 160 Mb 21.7 -> 20.3, 64 x 160 Mb 67.3 -> 65.7; profiles/r4b_ab_w6_maxrun.txt)
#ifndef PG_MAXRUN_W6
#define PG_MAXRUN_W6 PROBE_MAXRUN  // (24 until batches were cut at MAXRUN runs, see below: -DPG_MAXRUN_W6=24 with -DPG_PROBE_CUT=0)
#endif
    constexpr int MAXRUN = W_C == 6 ? PG_MAXRUN_W6 : PROBE_MAXRUN;
    constexpr int STAGE_ITERS = (MAXRUN * SLOTS + 63) / 64;  // 16-byte loads per lane per staging step
    // ONE block of LDS, carved up by hand, the tile's sequence words FIRST: the two-dword window reads of every batch
    // (ds_read2_b32, whose offsets reach 1020 bytes) then address them with an immediate instead of an add per window
    // (left to itself the compiler put the staging buffer first and the sequence words at 4.2-4.6 KB).
    constexpr int BUF_U4 = ((STAGE_ITERS * 64 + SLOTS - 1) / SLOTS) * LDS_LINE_U4;  // room for every staged chunk slot
    constexpr int OFF_RW = PROBE_SEQW * 8, OFF_NW = OFF_RW + (PROBE_SEQW + 1) * 8, OFF_LW = OFF_NW + PROBE_SEQW * 4;
    // (the 6-m-mer window stages 24 lines per step: with the full queue its 5968 bytes round up to 6144 = 26 waves per CU,
    // 6 per SIMD; 152 entries bring it to 5568 -> 5632 = 29 waves, 7 per SIMD)
    constexpr int QCAP = (W_C == 6 && MAXRUN > PROBE_MAXRUN && PROBE_QCAP > 152) ? 152 : PROBE_QCAP;
    constexpr int OFF_BUF = (OFF_LW + MAXRUN * 4 + 15) & ~15, OFF_QL = OFF_BUF + BUF_U4 * 16, OFF_QS = OFF_QL + QCAP * 4;
    constexpr int OFF_QP = OFF_QS + QCAP * 4, LDS_BYTES = OFF_QP + QCAP * 2;
    static_assert(OFF_LW <= 1020 || PROBE_TILE > 1024, "the sequence words must stay within reach of ds_read2_b32's offsets");
    __shared__ __attribute__((aligned(16))) uint8_t lds[LDS_BYTES];
    uint64_t *const sw = reinterpret_cast<uint64_t *>(lds);
    uint64_t *const rw = reinterpret_cast<uint64_t *>(lds + OFF_RW);
    uint32_t *const nw = reinterpret_cast<uint32_t *>(lds + OFF_NW);
    uint32_t(*const lines_w)[MAXRUN] = reinterpret_cast<uint32_t(*)[MAXRUN]>(lds + OFF_LW);
    uint4(*const buf)[BUF_U4] = reinterpret_cast<uint4(*)[BUF_U4]>(lds + OFF_BUF);
    uint32_t *const q_line = reinterpret_cast<uint32_t *>(lds + OFF_QL);  // overflow queue of the tile (position order): next line to try,
    uint32_t *const q_step = reinterpret_cast<uint32_t *>(lds + OFF_QS);  // step of the entry's sequence
    uint16_t *const q_pl = reinterpret_cast<uint16_t *>(lds + OFF_QP);    // and position within the tile (the key is re-derived from sw)
    PG_PH_DECL
    const int lane = threadIdx.x;
    const int k = (int)st.k;
    // (a minimizer table has 20 <= k <= 32, minimizer_length: the low word of the k-mer mask is all ones there, and the
    // compiler drops its ANDs once it knows)
    if constexpr (W_C != 0) __builtin_assume(k >= 20 && k <= 32);
    // tiles run in launch order unless the result carries a schedule (co-scheduled anchor genomes:
    // homologous regions of all genomes next to each other, so that table lines are shared in L2)
    // (tile_base: the launch covers a contig range of the result; a schedule is a permutation within such ranges)
    const uint32_t tile = sched ? sched[blockIdx.x + tile_base] : blockIdx.x + tile_base;
    const uint32_t c = tile_contig[tile];
    const AnchorDesc a = ad[c];
    const SeqDesc s = sd[c];
    const uint32_t tile_start = (tile - a.tile0) * PROBE_TILE;
    const uint32_t npos = min((uint32_t)PROBE_TILE, a.nkmers - tile_start);
    const bool hasn = has_n[c] != 0;

    // packed bases of the tile (+ halo), the only sequence traffic: 0.25 B per position
    for (int i = lane; i < PROBE_SEQW; i += 64) {
        const uint64_t wi = (uint64_t)(tile_start >> 5) + i;
        const uint64_t wv = wi < s.nwords ? seqw[s.seq_off + wi] : 0ull;
        sw[i] = wv;
        rw[PROBE_SEQW - 1 - i] = pair_reverse64(~wv);
        nw[i] = (hasn && wi < s.nwords) ? nmw[s.seq_off + wi] : 0u;
    }
    if (lane == 0) rw[PROBE_SEQW] = 0;  // (a window's three dwords may reach one word past the end)
    PG_WSYNC();  // single wave: compiles to a wave-level wait, not an s_barrier

    // (minimizer tables: k >= 20, the mask's low word is all ones — spelt out, the compiler drops the ANDs with it: two per
    // position)
    const uint64_t kmask = W_C ? (((uint64_t)(k == 32 ? ~0u : ((1u << (2 * k - 32)) - 1u)) << 32) | 0xFFFFFFFFull) : kmer_mask(k);
    const uint32_t rcb = PROBE_SEQ_BASES - (uint32_t)k;
    constexpr int HALO = W_C ? W_C - 1 : 0;  // m-mers of a window that lanes below its own supply
    constexpr bool CARRY = PG_PROBE_CARRY && W_C >= 2;  // the first lanes' missing m-mers carried over from the batch before (sliding_min_suffix)
    // CUT: a batch ends in front of its (MAXRUN + 1)-th run — the next batch starts there — so that ONE staging step takes
    // every batch: at w = 7 a quarter of the 58-position batches met more than 16 lines and paid a second, unhidden fetch
    // (not in direct mode, k < 20: every position is a run of its own there, and a batch of 16 positions would run the front
    // end four times where four staging steps share one)
    constexpr bool CUT = PG_PROBE_CUT != 0 && W_C != 0;
    constexpr int LHALO = CARRY ? 0 : HALO;  // lanes of a batch that only supply m-mers to their successors
    [[maybe_unused]] constexpr int STRIDE = 64 - LHALO;  // new positions per batch (at most: CUT)
    const uint32_t m = W_C ? (uint32_t)k - W_C + 1 : 0;
    const uint64_t mm64 = (m >= 32) ? ~0ull : ((1ull << (2 * (m ? m : 1))) - 1);
    // (columns mode: `out1` is the block's column buffer, `nbytes` its width in genomes; the "rows" of the tile are
    // the LDS words the columns are assembled in)
    __shared__ unsigned long long cols[ROWMODE == 3 ? TILE_SLOTS * COLS_G : 1];
    if constexpr (ROWMODE == 3) {
        for (uint32_t i = lane; i < TILE_SLOTS * COLS_G; i += 64) cols[i] = 0;
        __syncthreads();
    }
#if PG_ABLATE == 9 || PG_ABLATE == 10  // (timing experiment, wrong rows: every tile writes into a window of 256 tiles' rows — the L2 takes the stores, HBM sees none)
    uint8_t *tile_rows = out1 + (uint64_t)(blockIdx.x & 255u) * PROBE_TILE * nbytes;
#else
    uint8_t *tile_rows = ROWMODE == 3 ? reinterpret_cast<uint8_t *>(cols) : out1 + a.out_off + (uint64_t)tile_start * nbytes;
#endif
    uint32_t qn = 0;  // wave-uniform: overflow entries of this tile so far

    const uint8_t *chunk_base = st.buckets + (uint32_t)(lane % SLOTS) * 16u;  // this lane's 16-byte chunk of line 0
    const uint32_t lbytes = INL ? st.layout * 64u : st.slots * (WIDE ? 8u : 16u);  // bytes per line (= 16 * SLOTS = 128), as a run-time scalar (inline layout: LAYOUT_INLINE * 64)

    // A batch in three parts, so that the front end of batch i + 1 can run while the table lines of batch i are on
    // their way (PG_PROBE_PIPE): the fetch is the longest wait of a batch — a couple of thousand cycles behind a busy
    // texture-address unit and a TLB that misses on a quarter of these random lines — and the next batch's keys,
    // minimizers and runs need nothing from it.
    struct Front {  // what the front end of a batch leaves behind: no table access so far
        uint64_t key;
        uint32_t grp, line, rid, nruns, padline;
        uint32_t adv;                            // positions the batch covers: the next one starts that many further on (wave-uniform)
        uint32_t ln[PG_EARLY_LINES ? STAGE_ITERS : 1];  // the lines this lane fetches a chunk of in the first staging step
        unsigned long long rmask, amask, lmask;  // lanes with a position / active (no N in the window) / first of a run
    };
    // ---- the carry of a batch that starts at position b0 of the tile (the tile's first, or one the tail loop starts
    // again): lane j < HALO gets the rank of m-mer b0 + j (the m-mers before m-mer b0 + HALO, lane 0's own); the other
    // lanes' values are not used.  An m-mer straight from the staged sequence words: forward strand from sw, reverse
    // complement from rw — the values front() cuts out of X and B.
    auto carry_at = [&](const uint32_t b0) __attribute__((always_inline)) {
        uint32_t y = ~0u;
        if constexpr (CARRY) {
            const uint32_t q = b0 + (uint32_t)min(lane, HALO - 1);  // m-mer number = its first base in the tile
            const uint64_t fa = extract_bases32(reinterpret_cast<const uint32_t *>(sw), q) & mm64;
            const uint64_t fr = extract_bases32(reinterpret_cast<const uint32_t *>(rw), PROBE_SEQ_BASES - q - m) & mm64;
            const uint32_t h = M64 ? mmer_rank(fa < fr ? fa : fr) : mz_order(min((uint32_t)fa, (uint32_t)fr));
            y = h;
        }
        return y;
    };
    uint32_t carry = ~0u;  // (CARRY) what the next front() takes in (lanes 0 .. HALO - 1): written by carry_at or by the front() before
    // ---- keys: lane = position b0 + lane - LHALO; it also owns m-mer number b0 + lane + (HALO - LHALO) ----
    // FIRST: the tile's first batch, whose leading HALO lanes stand before the tile's first k-mer (pl < 0: they take
    // their m-mers out of that k-mer, at their own offsets); in every later batch a lane's m-mer sits at the one fixed
    // offset and its position needs no clamp — instantiated twice so that the later batches carry neither.
    auto front = [&](auto first_tag, const uint32_t b0) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        Front f;
        const int32_t pl = (int32_t)(b0 + lane) - LHALO;
        // (pl >= b0 exactly for the lanes behind the halo: a constant lane mask; a ballot straight off the compare stays
        // a scalar mask, one of a bool that was AND-ed together first is rebuilt through 0 / 1)
        f.rmask = __builtin_amdgcn_ballot_w64(pl < (int32_t)npos) & ~((1ull << LHALO) - 1ull);
        const uint32_t pq = (FIRST && LHALO) ? (uint32_t)max(pl, 0) : (uint32_t)pl;
        const uint64_t X = extract_bases32(reinterpret_cast<const uint32_t *>(sw), pq) & kmask;
        const uint64_t B = revcomp_window(rw, X, pq, k, kmask, rcb);
        f.key = ~(X > B ? X : B) & kmask;  // (canonical_from_xb with this kernel's mask)
        f.amask = f.rmask;
        if (hasn) f.amask &= __builtin_amdgcn_ballot_w64(extract_nmask(nw, pq, k) == 0);
        [[maybe_unused]] uint32_t own = 0;
        if (W_C) {
            // m-mer number b0+lane is the LAST m-mer of this lane's own k-mer (the first lanes of a tile, which have no
            // k-mer, take theirs out of the tile's first k-mer): forward strand from X, reverse complement from B — no
            // second pass over the sequence words
            const uint32_t off = (FIRST && LHALO) ? (uint32_t)(pl + HALO) - pq : (uint32_t)HALO;  // m-mer's offset inside the k-mer, 0..HALO
            if constexpr (!M64) {  // m-mers of up to 32 bits: one funnel shift each, no 64-bit arithmetic
                const uint32_t mm32 = (uint32_t)mm64;
                const uint32_t fa = __builtin_amdgcn_alignbit((uint32_t)(X >> 32), (uint32_t)X, 2 * off) & mm32;
                const uint32_t fr = __builtin_amdgcn_alignbit((uint32_t)(B >> 32), (uint32_t)B, 2 * (HALO - off)) & mm32;
                f.grp = mz_order(min(fa, fr));  // (= mmer_rank: its fold of the high half is the identity here)
            } else {
                const uint64_t fa = (X >> (2 * off)) & mm64;
                const uint64_t fr = (B >> (2 * (HALO - off))) & mm64;
                f.grp = mmer_rank(fa < fr ? fa : fr);
            }
#ifdef PG_DUMMY_VALU
            {   // (experiment: PG_DUMMY_VALU extra VALU instructions per batch, nothing else changed)
                uint32_t dm = f.grp;
#pragma unroll
                for (int i = 0; i < PG_DUMMY_VALU; ++i) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(dm) : "v"((uint32_t)lane));
                if (dm == 0x12345u && lane == 77) out1[0] = 1;
            }
#endif
            // sliding minimum over lanes [lane-W_C+1, lane]: m <- min(own rank, m of the lane below), W_C-1 times
            // (the first lanes of the wave see shorter windows: they are halo lanes, never active)
            if constexpr (CARRY) {
                own = f.grp;  // (this lane's own m-mer rank: the next batch's carry is cut out of these below)
                uint32_t suffix = __builtin_amdgcn_inverse_ballot_w64((1ull << HALO) - 1ull) ? carry : ~0u;
                f.grp = sliding_min_suffix<W_C >= 2 ? W_C : 2>(f.grp, suffix);
                f.grp = min(f.grp, suffix);
            } else {
                f.grp = sliding_min<W_C ? W_C : 1>(f.grp);
            }
        } else {
            f.grp = group_of_key(f.key);
        }
        // ---- runs of equal home line among the active lanes (lane masks on the scalar unit: a run starts at an active
        // lane whose predecessor is inactive or on another line; lane 0 has no predecessor — the shift leaves its bit clear)
        f.line = home_of_group(f.grp, st.nbuckets);
        f.lmask = f.amask & (~(f.amask << 1) | differs_from_lane_below(f.line));
        f.rid = run_ids(f.lmask);  // run id of an active lane
        uint32_t lanes_kept = 64u;
        if constexpr (CUT) {
            // the lanes of the first MAXRUN runs — a prefix of the wave: run ids do not fall from lane to lane; the lanes
            // in front of the first run count -1 — stay; the others' positions are the next batch's
            const unsigned long long keep = __builtin_amdgcn_sicmp((int32_t)f.rid, MAXRUN, 40 /* signed < */);
            f.amask &= keep;
            f.rmask &= keep;
            f.lmask &= keep;
            lanes_kept = (uint32_t)__popcll(keep);  // (>= MAXRUN + LHALO when anything is cut: the batch always advances)
        }
        f.adv = lanes_kept - LHALO;
        if constexpr (CARRY) {
            // the next batch's carry: the HALO m-mer ranks in front of its first position's own = lanes lanes_kept - HALO
            // .. lanes_kept - 1 of this batch (not needed before the next front(): the LDS crossbar's latency is hidden)
            // (lanes 0 .. HALO - 1, the only ones whose result is used, read lanes below 64: no wrap to take care of)
            carry = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane << 2) + ((lanes_kept - HALO) << 2)), (int)own);
        }
        f.nruns = (uint32_t)__popcll(f.lmask);
#if PG_RID_ADDC
        // (a batch without runs: s_ff1 of an empty mask is -1, v_readlane takes it modulo 64 — lane 63's line, a line of the table
        // like any other, fetched and scanned by no lane: no test for the empty mask)
        // (the instruction itself, not __builtin_ctzll: ctz of 0 is undefined to the COMPILER — it may fold anything — while
        // s_ff1_i32_b64 is defined by the ISA to return -1; __ffsll - 1 is defined too and costs two scalar instructions more)
        int first_run;
        asm("s_ff1_i32_b64 %0, %1" : "=s"(first_run) : "s"(f.lmask));
        f.padline = (uint32_t)__builtin_amdgcn_readlane((int)f.line, first_run & 63);  // (wave-uniform)
#else
        f.padline = f.lmask ? (uint32_t)__builtin_amdgcn_readlane((int)f.line, __builtin_ctzll(f.lmask)) : 0u;  // (wave-uniform)
#endif
#if PG_ABLATE == 3  // (timing experiment: keys, minimizers and runs only — no table access)
        if (f.line != 0xDEADBEEFu) f.nruns = 0;
#endif
        if constexpr (PG_EARLY_LINES != 0) {
            // The first staging step's line numbers go through lines_w HERE — one per run, padded with the first run's
            // (see issue) — and every lane reads back the ones it will fetch a chunk of: two LDS round trips that used to
            // stand between "the chunks of the batch before are staged" and this batch's fetch; the read is in flight
            // while the caller does something else (the skewed order: the whole staging of the batch before).
            const bool leader = __builtin_amdgcn_inverse_ballot_w64(f.lmask);
            if (lane < MAXRUN) lines_w[0][lane] = f.padline;
            if (leader && f.rid < (uint32_t)MAXRUN) lines_w[0][f.rid] = f.line;
            PG_WSYNC();
#pragma unroll
            for (int it = 0; it < STAGE_ITERS; ++it) f.ln[it] = lines_w[0][it * (64 / SLOTS) + lane / SLOTS];
        }
        return f;
    };
    // ---- the fetch of a staging step: runs r0 .. r0 + MAXRUN - 1 of the batch, MAXRUN lines = MAXRUN * SLOTS chunks of 16
    // bytes, coalesced, all loads of the step in flight.  The step's line numbers go through lines_w, one per run; the
    // entries behind the last run repeat the first run's line (the same address again inside one load: no second
    // fetch), so that a lane reads its entries at fixed places — no clamp of the entry's number, one LDS instruction for
    // all of a lane's entries.  Loads are unconditional: any predicate makes the compiler sink each load into its own
    // branch and wait for it there.  A lane always fetches chunk lane % SLOTS of a line: its address = the lane's own
    // base + line x line bytes, one multiply-add (the line bytes come from the table's descriptor, a scalar the
    // compiler cannot turn into a 64-bit shift and a 64-bit add).
    struct Lines {  // the step's 16-byte chunks on their way into this lane's registers
        uint4 v[STAGE_ITERS];
    };
    auto issue = [&](const Front &f, const uint32_t r0) __attribute__((always_inline)) {
        Lines L;
        uint32_t ln[STAGE_ITERS];
        if (PG_EARLY_LINES == 0 || r0 != 0u) {
            const uint32_t nl = min((uint32_t)MAXRUN, f.nruns - r0);  // (0 for a batch without runs: r0 is 0 then)
            const bool leader = __builtin_amdgcn_inverse_ballot_w64(f.lmask);
            if (lane < MAXRUN) lines_w[0][lane] = f.padline;
            if (leader && f.rid - r0 < nl) lines_w[0][f.rid - r0] = f.line;
            PG_WSYNC();
        }
#pragma unroll
        for (int it = 0; it < STAGE_ITERS; ++it) {
            if (PG_EARLY_LINES == 0 || r0 != 0u) ln[it] = lines_w[0][it * (64 / SLOTS) + lane / SLOTS];
            else ln[it] = f.ln[it];
#if PG_ABLATE == 1  // (timing experiment, wrong rows: every fetch a cache hit — the lines of one 64 KB window)
            ln[it] &= 511u;
#endif
        }
#pragma unroll
        for (int it = 0; it < STAGE_ITERS; ++it) L.v[it] = *reinterpret_cast<const uint4 *>(chunk_base + (uint64_t)ln[it] * lbytes);
        return L;
    };
    // ---- the rest of the batch: its lines into LDS, every lane scans its own; overflow entries; the rows ----
    auto back = [&](const Front &f, const Lines &L, const uint32_t b0, auto &&after_staging) __attribute__((always_inline)) {
        const bool act = __builtin_amdgcn_inverse_ballot_w64(f.amask), inrange = __builtin_amdgcn_inverse_ballot_w64(f.rmask);
        const int32_t pl = (int32_t)(b0 + lane) - LHALO;
        uint32_t m0 = 0, m1 = 0;
        [[maybe_unused]] uint32_t rw4[4] = {0, 0, 0, 0};  // (inline layout: the row's words, zeros for an absent key)
        int rcode = 0;
        // a staging step's chunks into LDS (wave-uniform placement: chunk idx of the step -> line idx / SLOTS, slot
        // idx % SLOTS), then every lane of the step's runs scans its own line
        auto stage = [&](const Lines &S) __attribute__((always_inline)) {
            // (chunk it * 64 + lane of the step: line it * (64 / SLOTS) + lane / SLOTS, slot lane % SLOTS — one address per
            // lane, the iterations at constant offsets from it)
            uint4 *const mine = buf[0] + ((uint32_t)lane / SLOTS) * LDS_LINE_U4;
#pragma unroll
            for (int it = 0; it < STAGE_ITERS; ++it) {
                if constexpr (WIDE) mine[it * (64 / SLOTS) * LDS_LINE_U4 + ((uint32_t)lane % SLOTS)] = S.v[it];  // (bare keys: as they are)
                else stage_chunk(mine + it * (64 / SLOTS) * LDS_LINE_U4, (uint32_t)lane % SLOTS, SLOTS, S.v[it]);
            }
            PG_WSYNC();
        };
        constexpr bool FLAT = PG_SCAN_FLAT && PG_ABLATE == 0 && CUT && !WIDE && SLOTS == 8;  // (CUT: an active lane's run is one of the step's)
        unsigned long long ovf_flat = 0;
        auto scan = [&](const uint32_t r0) __attribute__((always_inline)) {
            if constexpr (FLAT) {
                ovf_flat = scan_line_lds_flat<TWO>(buf[0] + min(f.rid, (uint32_t)MAXRUN - 1u) * LDS_LINE_U4, f.key, f.amask, m0, m1);
                PG_WSYNC();
                return;
            }
            const uint32_t nl = min((uint32_t)MAXRUN, f.nruns - r0);
#if PG_ABLATE == 5  // (timing experiment: lines fetched and staged, no slot scan: every lane "hits" with one LDS word)
            if (act && f.rid - r0 < nl) {
                rcode = 1;
                m0 = m1 = reinterpret_cast<const uint32_t *>(buf[0] + (f.rid - r0) * LDS_LINE_U4)[2];
            }
#else
            if (act && f.rid - r0 < nl) {
                if constexpr (WIDE && INL) {
                    rcode = scan_keys_inl(buf[0] + (f.rid - r0) * LDS_LINE_U4, f.key, st.slots, m1);
                    if (m1) masks_inl(buf[0] + (f.rid - r0) * LDS_LINE_U4, st.slots, st.W, m1 - 1u, rw4);
                } else if constexpr (WIDE) {
                    rcode = scan_keys16_lds(buf[0] + (f.rid - r0) * LDS_LINE_U4, f.key, m1);
                    m0 = f.line;
                } else {
                    rcode = scan_line_lds<TWO, SLOTS>(buf[0] + (f.rid - r0) * LDS_LINE_U4, f.key, m0, m1);
                }
            }
#endif
            PG_WSYNC();
        };
        // The batch's lines are waited for HERE, on every path (a batch without runs has none in flight).  Left to the
        // compiler, the wait sits inside the branch below, the chunks count as possibly still on their way where the
        // paths meet, and the first instruction that re-uses one of their registers — an address of the NEXT fetch,
        // issued right behind this batch's row store — gets a vmcnt(0) that waits for that store to be acknowledged.
        PG_PH(1)
        PG_PRIO_AT(1)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), nothing else
        PG_PH(2)
        // (a batch without runs — a stretch of N — is not singled out: its lines_w hold line 0 throughout (padline), which is fetched,
        // staged and scanned by no lane; three wave-uniform tests and their branches per batch were 10 scalar instructions)
        if (PG_RUNLESS_TESTS == 0 || f.nruns) stage(L);
        PG_PH(3)
        PG_PRIO_AT(2)
        after_staging();  // (the chunks' registers are free from here on: the skewed order starts the NEXT batch's fetch now)
        PG_PRIO_AT(3)
        PG_PH(4)
        if (PG_RUNLESS_TESTS == 0 || f.nruns) scan(0u);
        PG_PH(5)
        if constexpr (!CUT)
        for (uint32_t r0 = MAXRUN; r0 < f.nruns; r0 += MAXRUN) {  // (a batch with more than MAXRUN lines: a quarter of the batches at w = 7)
            const Lines X = issue(f, r0);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            stage(X);
            scan(r0);
        }
        // ---- overflow: absent from a full line -> queue entry for the next line of its sequence ----
        // (sicmp = the compare as a lane mask, predicate 40 = signed less-than: a ballot of `rcode < 0` is sunk into the
        // blocks rcode comes from and its bool rebuilt here through 0 / 1)
        const unsigned long long omask = FLAT ? ovf_flat : (f.amask & __builtin_amdgcn_sicmp(rcode, 0, 40));
        const bool ovf = __builtin_amdgcn_inverse_ballot_w64(omask);
        if (omask) {
            // (the entry carries the home line and the GROUP: its step and next line — two multiplies and the wrap — are
            // worked out by the drain's dense batches (drain_queue, level 1), not here for the few overflowing lanes of
            // every batch: nine VALU instructions per batch)
            uint32_t step, nx;
            if constexpr (GROUP_CHAIN == 1) {  // (tuning build: the group owns its home line only — level 1 is the key's own sequence)
                key_sequence(f.key, st.nbuckets, nx, step);
            } else {
                step = f.grp;
                nx = f.line;
            }
            // (the queue always has room for a whole batch: the batch loop leaves for an early drain before it could not)
            const uint32_t slot = set_lane_index(omask, qn);
            if (ovf) {
                q_line[slot] = nx;
                q_step[slot] = step;
                q_pl[slot] = (uint16_t)pl;
            }
            qn += (uint32_t)__popcll(omask);
        }
        PG_PH(10)
        // (32-bit offset from the tile's uniform base: one store with a scalar base address)
        if constexpr (WIDE && INL) {
            // (ragged rows: one store per row — not in the tile's last batch, see store_row_fused; positions without a row hold zeros)
            const bool fuse = PG_ROW_FUSE && (nbytes & 3u) != 0u && b0 + (uint32_t)__popcll(f.rmask) < npos;  // (wave-uniform)
            if (fuse) {
                const uint32_t mine = inrange ? rw4[0] : 0u;
                const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);  // (lane 63: 0)
                if (inrange) store_row_fused(tile_rows + (uint64_t)(uint32_t)pl * nbytes, nbytes, rw4, nxt);
            } else if (inrange) {
                store_row_regs(tile_rows + (uint64_t)(uint32_t)pl * nbytes, nbytes, rw4);
            }
        } else if constexpr (WIDE) {
#if PG_ABLATE == 2  // (timing experiment: no mask gather, no row store)
            if (inrange && m0 == 0xDEADBEEFu && m1 == 77u)
#elif PG_ABLATE == 8 || PG_ABLATE == 10  // (timing experiment, wrong rows: the row store without a DIVERGENT mask gather — every hit copies slot 0 of line 0's block)
            if (m1) m0 = 0, m1 = 1;
            if (inrange)
#else
            // (rows of 13..15 bytes, W = 4: the hit's four mask words into registers and ONE 16-byte store per row, as the inline
            // layout's ragged rows — store_row_fused; not in the tile's last batch)
            if (PG_ROW_FUSE && st.W == 4u && (nbytes & 3u) != 0u && b0 + (uint32_t)__popcll(f.rmask) < npos) {  // (wave-uniform)
                uint32_t v4[4] = {0, 0, 0, 0};
                if (inrange && m1) {
                    const WordsN<4> g = *reinterpret_cast<const WordsN<4> *>(reinterpret_cast<const uint32_t *>(st.masks) + ((uint64_t)m0 * SPLIT_KEYS + (m1 - 1u)) * 4u);
                    v4[0] = g.w[0], v4[1] = g.w[1], v4[2] = g.w[2], v4[3] = g.w[3];
                }
                const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v4[0], 0x130 /* wave_shl:1 */, 0xf, 0xf, true);  // (lanes without a row hold 0)
                if (inrange) store_row_fused(tile_rows + (uint64_t)(uint32_t)pl * nbytes, nbytes, v4, nxt);
            } else if (inrange)
#endif
                store_row_wide(st.masks, st.W, nbytes, tile_rows + (uint64_t)(uint32_t)pl * nbytes, m0, m1);
        } else if constexpr (ROWMODE == 3) {
            const uint32_t w0 = b0 >> 6, sh = b0 & 63u;
            for (uint32_t j = 0; j < rc.col0; ++j) {  // (uniform) one ballot per genome of the block
                const unsigned long long shifted = __ballot(inrange && ((m0 >> j) & 1u)) >> LHALO;  // bit i = position b0 + i
                if (lane == 0 && shifted) {
                    cols[w0 * COLS_G + j] |= shifted << sh;
                    if (sh && (shifted >> (64u - sh))) cols[(w0 + 1) * COLS_G + j] |= shifted >> (64u - sh);
                }
            }
        } else {
            if (PG_ROW_FUSE3 && ROWMODE == 6 && b0 + (uint32_t)__popcll(f.rmask) < npos) {  // (wave-uniform; ROWMODE a constant)
                // (three-byte rows as ONE unaligned dword per row — the row and the next lane's first byte — as the
                // 5..7-byte rows; not in the tile's last batch)
                const uint32_t mine = inrange ? m0 : 0u;
                const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
                struct __attribute__((packed)) U32 { uint32_t v; };
                if (inrange) reinterpret_cast<U32 *>(tile_rows + (uint32_t)pl * 3u)->v = (m0 & 0xFFFFFFu) | (nxt << 24);
            } else if constexpr (ROWMODE == 6) {
                // Three-byte rows as ALIGNED dwords: a u16 and a u8 per lane at odd addresses cost the launch a third
                // of its time (20 x 40 Mb: 6.7 ps per position against 5.1 with the four-byte rows of 27 genomes).
                // The batch's rows are 3 x 58 consecutive bytes; the aligned dword that starts inside a lane's row
                // holds the row's last 3 - j bytes and the next lane's first j + 1 (j = 0..2 by the row's address
                // mod 4; a row that starts at byte 1 of a dword starts none) — the next lane's mask comes over one DPP
                // shift, and ~43 lanes write whole dwords.  The lane without a successor in the batch (lane 63, or the
                // tile's last position) and the batch's first position, whose leading bytes no lane of this batch
                // covers, write their own three bytes as before; stores that overlap write equal bytes.
                const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m0, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
                const uint32_t o4 = (uint32_t)reinterpret_cast<uintptr_t>(tile_rows) & 3u;  // (a tile starts at a multiple of 512 rows)
                const uint32_t j = ((uint32_t)pl - o4) & 3u;
                const uint32_t lo = (nxt << 24) | (m0 & 0xFFFFFFu);
                const uint32_t dw = __builtin_amdgcn_alignbit(nxt >> 8, lo, 8u * j);
                const unsigned long long nextin = f.rmask >> 1;
                const unsigned long long dmask = f.rmask & nextin & __builtin_amdgcn_ballot_w64(j != 3u);
                const unsigned long long fmask = f.rmask & ((1ull << LHALO) | ~nextin);
                uint8_t *row = tile_rows + (uint32_t)pl * 3u;
                if (__builtin_amdgcn_inverse_ballot_w64(dmask)) *reinterpret_cast<uint32_t *>(row + j) = dw;
                if (__builtin_amdgcn_inverse_ballot_w64(fmask)) store_row<ROWMODE>(row, m0, m1, rc);
            } else
#if PG_ABLATE == 2  // (timing experiment: no row store unless a value no mask has turns up)
            if (inrange && m0 == 0xDEADBEEFu)
#else
            // (rows of 5..7 bytes, 33..56 genomes: ONE 8-byte store per row — the row and the first bytes of the next lane's, as
            // store_row_fused does for the wider ragged rows; not in the tile's last batch)
            if (PG_ROW_FUSE8 && ROWMODE == 0 && TWO && rc.words == 3u && nbytes >= 5u && b0 + (uint32_t)__popcll(f.rmask) < npos) {  // (wave-uniform)
                const uint32_t mine = inrange ? m0 : 0u;
                const uint32_t nxt = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
                if (inrange) {
                    typedef uint32_t u32x2u __attribute__((ext_vector_type(2), aligned(1)));
                    const uint32_t t8 = 8u * (nbytes - 4u);  // bits of the row in its second word (8, 16 or 24)
                    u32x2u q = {m0, (m1 & ((1u << t8) - 1u)) | (nxt << t8)};
                    *reinterpret_cast<u32x2u *>(tile_rows + (uint32_t)pl * nbytes) = q;  // (non-temporal: 56 genomes +3 %, 40 genomes mixed — plain)
                }
            } else if (inrange)
#endif
                store_row<ROWMODE>(tile_rows + (ROWMODE == 1 ? (uint32_t)pl : ROWMODE == 4 ? (uint32_t)pl * 4u : ROWMODE == 5 ? (uint32_t)pl * 2u : ROWMODE == 6 ? (uint32_t)pl * 3u : (uint32_t)pl * nbytes), m0, m1, rc);
        }
        PG_PH(6)
        PG_PRIO_AT(4)
    };
    // The skewed order — front end of batch i + 1 between the fetch of batch i and its use — keeps the fetched chunks
    // and two batches' keys in registers at once: it pays where that still fits 64 registers (8 waves per SIMD: this
    // kernel's speed goes with its occupancy — 7 waves cost 7 %, 5 waves 28 %), i.e. for the one-byte rows of up to 8
    // genomes (configs[1]: 4.31 -> 4.17 ms); the wider instantiations spill there and keep the plain order.
    constexpr bool PIPE = probe_pipelined<W_C, TWO, ROWMODE, SLOTS, WIDE, INL>;
    constexpr int LEVELS = (W_C >= 6 && !TWO && !WIDE && PROBE_STAGED_LEVELS > 1) ? 1 : PROBE_STAGED_LEVELS;  // (W_C >= 6: the wide-window tables of up to 16 genomes)
    // The batch loop runs until the tile is through OR the overflow queue could not take another full batch (tiles inside
    // a repeat family: most of their keys sit outside their home lines): the queue is drained then and the loop goes on
    // where it stopped.  So no batch ever meets a full queue — the lane-by-lane chase that served it was a function
    // call inside the loop, whose clobbered registers the allocator answered with spills on the hot path.
    static_assert(QCAP >= 128, "an early drain must leave room for a batch");
    constexpr uint32_t QROOM = QCAP - 64;  // entries the queue may hold when a batch starts
    // (The hot loop stands on its own: what it keeps in registers is dead when it ends.  The tail — a plain batch loop
    // around the one drain — only ever runs batches for the tiles whose queue filled up.)
    uint32_t b0 = 0;
    if constexpr (PIPE) {
        // The skewed order — front end of batch i + 1 between the fetch of batch i and its use — keeps the fetched chunks
        // and two batches' keys in registers at once: it pays where that still fits 64 registers (8 waves per SIMD: this
        // kernel's speed goes with its occupancy — 7 waves cost 7 %, 5 waves 28 %)
        Lines L;
#pragma unroll
        for (int it = 0; it < STAGE_ITERS; ++it) L.v[it] = make_uint4(0, 0, 0, 0);
        PG_PH(0)
        carry = carry_at(0u);
        Front cur = front(std::true_type{}, 0u);
        if (PG_RUNLESS_TESTS == 0 || cur.nruns) L = issue(cur, 0u);
        for (;;) {
            PG_PH(7)
            const uint32_t b1 = b0 + cur.adv;
            const bool more = b1 < npos;  // (wave-uniform)
            Front nxt = cur;
            __builtin_amdgcn_sched_barrier(0);  // (the parts stay apart: interleaved by the scheduler they keep both batches' temporaries alive)
            if (more) nxt = front(std::false_type{}, b1);  // while the lines of `cur` are on their way
            __builtin_amdgcn_sched_barrier(0);
#if PG_PROBE_PIPE >= 2
            // the next batch's fetch goes out as soon as this batch's chunks have left their registers for LDS: it is in
            // flight during this batch's slot scan, row store and overflow entries AND the front end after next
            back(cur, L, b0, [&]() __attribute__((always_inline)) {
                if (more && (PG_RUNLESS_TESTS == 0 || nxt.nruns)) L = issue(nxt, 0u);
            });
            __builtin_amdgcn_sched_barrier(0);
            b0 = b1;
            if (!more || qn > QROOM) break;  // (rare way out with a fetch in flight: nobody reads it; the tail starts the batch again)
            cur = nxt;
#else
            back(cur, L, b0, [] {});
            __builtin_amdgcn_sched_barrier(0);
            b0 = b1;
            // (the room is checked AFTER a batch's look-up, which is what fills the queue: the front end of the next batch
            // is thrown away on the rare way out — the tail starts it again)
            if (!more || qn > QROOM) break;
            cur = nxt;
            if (PG_RUNLESS_TESTS == 0 || cur.nruns) L = issue(cur, 0u);
#endif
        }
    } else {
        {
            carry = carry_at(0u);
            const Front f = front(std::true_type{}, 0u);
            const Lines L = issue(f, 0u);  // (also for a batch without runs — a stretch of N: line 0 is fetched and ignored)
            back(f, L, 0u, [] {});
            b0 = f.adv;
        }
        while (b0 < npos && qn <= QROOM) {
            const Front f = front(std::false_type{}, b0);
            const Lines L = issue(f, 0u);
            back(f, L, b0, [] {});
            b0 += f.adv;
        }
    }
    PG_PH(7)
    for (;;) {
        drain_queue<TWO, ROWMODE, SLOTS, MAXRUN, WIDE, LEVELS, INL>(st, qn, sw, rw, q_line, q_step, q_pl, lines_w[0], buf[0], tile_rows, nbytes, rc, lane);
        qn = 0;
        if (b0 >= npos) break;
        PG_WSYNC();
        carry = carry_at(b0);  // (the hot loop may have left with the front end of a batch it did not finish)
        while (b0 < npos && qn <= QROOM) {
            const Front f = front(std::false_type{}, b0);
            const Lines L = issue(f, 0u);
            back(f, L, b0, [] {});
            b0 += f.adv;
        }
    }
    PG_PH(8)
    PG_PH_FLUSH
    if constexpr (FUSE) {
        static_assert(ROWMODE != 1 && ROWMODE != 3 && LDS_BYTES >= 4 * 4 * 2 * 129, "fused statistics: rows of 2..16 bytes; four copies of 2 x 129 histogram counters in LDS");
        // Everything about the tile is looked up AGAIN (through pointers the compiler cannot identify with the ones the kernel
        // began with): kept alive across the batch loop these values would cost it scalar registers it does not have.
        const AnchorDesc *ad2 = ad;
        const uint32_t *tc2 = tile_contig, *sched2 = sched;
        asm volatile("" : "+s"(ad2), "+s"(tc2), "+s"(sched2));
        const uint32_t tile2 = sched2 ? sched2[blockIdx.x + tile_base] : blockIdx.x + tile_base;
        const AnchorDesc a2 = ad2[tc2[tile2]];
        if (a2.binlen >= (uint32_t)PROBE_TILE) {  // (wave-uniform) contigs with shorter bins are the statistics pass's (launched over their tiles only)
            const uint32_t ts2 = (tile2 - a2.tile0) * PROBE_TILE;
            const uint32_t np2 = min((uint32_t)PROBE_TILE, a2.nkmers - ts2);
            const uint8_t *rows2 = out1 + a2.out_off + (uint64_t)ts2 * nbytes;
            uint32_t *H = reinterpret_cast<uint32_t *>(lds);
            if constexpr (WIDE) {
                if (nbytes <= 12u) tile_statistics<3>(H, rows2, np2, nbytes, a2, ts2, tile2, fo, lane);
                else tile_statistics<4>(H, rows2, np2, nbytes, a2, ts2, tile2, fo, lane);
            } else {
                tile_statistics<TWO ? 2 : 1>(H, rows2, np2, nbytes, a2, ts2, tile2, fo, lane);
            }
        }
    }
    if constexpr (ROWMODE == 3) {
        // the tile's columns: TILE_SLOTS slots x `width` (= nbytes) genomes, slot-major — what k_cols_extract would have written
        __syncthreads();
        const uint32_t width = nbytes, trel = tile - tile_base;
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(out1) + (uint64_t)trel * TILE_SLOTS * width;
        for (uint32_t i = lane; i < TILE_SLOTS * width; i += 64) dst[i] = cols[(i / width) * COLS_G + i % width];
    }
}

// ---------------------------------------------------------------------------
// k-mer set construction, wave-cooperative (replaces one thread per k-mer with a per-thread minimizer loop and eight
// volatile key loads each): the SAME front end as k_probe — one wave per tile of 512 positions, per 64-lane batch
// canonical keys, minimizer by the sliding DPP minimum, runs of equal home line, the batch's lines staged into LDS
// with coalesced 128-byte fetches — then every lane scans its line's snapshot:
//   key found (every later genome's common case: 85-99 % of its k-mers exist already)  ->  OR the genome's bit into
//       the slot's mask word (a plain store when the snapshot shows the bit missing: see lane_insert_grp's note on
//       the one-writer invariant) or, in counting mode, atomicAdd
//   key absent in the snapshot  ->  (position, group) goes to the wave's LDS queue; the queue is worked off densely,
//       64 entries at a time, by the CAS-claiming insert (wave_insert_batch: lane_insert_grp's protocol — claims race
//       correctly against other waves — with one compare-and-swap per address and round; the group id is handed over)
// counters[0] += newly claimed keys; counters[1] = overflow flag (a probe sequence exceeded max_probe lines).
// ---------------------------------------------------------------------------
constexpr int INSERT_QCAP = 256;
#ifndef PG_INSERT_SPREAD
#define PG_INSERT_SPREAD 1
#endif

// The claiming insert for a batch of 64 queue entries, wave-cooperative.  The protocol is lane_insert_grp's (a key may
// be claimed in a slot only by someone who has seen every earlier slot of its sequence hold OTHER keys; slots never
// revert), but the lanes of a run — neighbouring queue entries of one minimizer group walk the same lines in step —
// no longer each throw their own compare-and-swap at the run's next free slot: per round only the FIRST of the lanes
// that aim at an address issues the CAS, and what the slot holds afterwards (its own key if it won, else the key it
// found there) is handed to the others, who compare and move on.  A device-scope CAS costs far more than anything
// else here (3.2 of them per new key were 11 of the first genome's 13 ms at config 2); now about one per new key.
// A line is read with 8 relaxed atomic loads in flight together (volatile loads are waited for one by one).
// r: 0 = existed, 1 = newly claimed, -1 = gave up after max_probe lines.
// CLAIM = false (update-only builds, pg_table_update_seqset): a key the table does not hold is left out — the walk ends at
// the first line that is not full, nothing is claimed.
template <bool COUNT, bool CLAIM = true>
__device__ __forceinline__ int wave_insert_batch(const SubTable &st, bool valid, uint64_t key, uint32_t grp, int w, uint32_t bits,
                                                 uint32_t max_probe, int lane) {
    const uint32_t kstride = key_stride(st), slots = st.slots;
    uint32_t b = home_of_group(grp, st.nbuckets), step = step_of_group(grp, st.nbuckets), probes = 0;
    int s = -1;  // slot to try next in line b; -1: the line has not been read yet; slots: the line is full
    bool done = !valid, fresh = false;  // fresh: the line has just been read, s is its first empty slot
    int r = 0;
    auto shfl64 = [](uint64_t v, int src) __attribute__((always_inline)) {
        return (uint64_t)(uint32_t)__shfl((int)(uint32_t)v, src) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src) << 32);
    };
    auto resolve = [&](uint32_t sl, bool claimed) __attribute__((always_inline)) {
        uint32_t *mp = mask_ptr(st, b, sl, (uint32_t)w);
        if (COUNT) {
            if (*mp < 0xFFFFFF00u) atomicAdd(mp, bits);  // saturates far above any -ci threshold
        } else if (claimed) {
            *reinterpret_cast<volatile uint32_t *>(mp) = bits;  // a fresh slot's mask words are zero (lane_insert_grp's note on racing writers applies)
        } else {
            const uint32_t cur = *mp;
            if ((cur & bits) != bits) *reinterpret_cast<volatile uint32_t *>(mp) = cur | bits;
        }
        r = claimed ? 1 : 0;
        done = true;
    };
    while (__ballot(!done)) {
        if (!done && s >= (int)slots) {  // on to the next line of the sequence
            if (++probes >= max_probe) {
                done = true;
                r = -1;
            } else {
                advance_line(key, probes, st.nbuckets, b, step);
                s = -1;
            }
        }
        if (!done && s < 0) {  // read the line: the key itself, or the first empty slot (lines fill front to back)
            uint8_t *base = st.buckets + (uint64_t)b * line_bytes(st);
            int hit = -1, fr = -1;
            for (uint32_t s0 = 0; s0 < slots; s0 += 8) {
                uint64_t kk[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    kk[i] = __hip_atomic_load(reinterpret_cast<unsigned long long *>(base + kstride * (s0 + i)), __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
                    if (s0 + (uint32_t)i >= slots) kk[i] = TOMB_KEY;  // (inline layout: mask words behind the line's 5 or 6 keys — neither a key nor empty)
                }
                int h8 = -1, f8 = -1;
#pragma unroll
                for (int i = 7; i >= 0; --i) {
                    h8 = (kk[i] == key) ? i : h8;
                    f8 = (kk[i] == EMPTY_KEY) ? i : f8;
                }
                if (hit < 0 && h8 >= 0) hit = (int)s0 + h8;
                if (fr < 0 && f8 >= 0) fr = (int)s0 + f8;
            }
            if (hit >= 0) resolve((uint32_t)hit, false);
            else if (!CLAIM && fr >= 0) done = true;  // not in a line that is not full: the table does not hold it
            else {
                s = fr >= 0 ? fr : (int)slots;
                fresh = true;
            }
        }
        if constexpr (!CLAIM) continue;  // (s is -1 or slots here: read the next line, or finished)
#if PG_INSERT_SPREAD
        // The lanes of a run that have just read their line claim DISTINCT slots in one round: first empty slot + rank
        // in the run.  A claim made this way has not yet seen the slots below it; it sees them right after — every slot
        // of the run's range is occupied once the round is over, and what each holds comes back with its lane's CAS —
        // and if a LOWER one turns out to hold the same key (an equal key further up in the run, or another wave's
        // claim that landed in between), the claim is retired (TOMB_KEY) and the key counts as found below.  Two
        // racing claims of one key in one line always see each other from the higher slot (its range reaches down to
        // what was the first empty slot when it read the line, and the lower claim was not there yet or it would
        // have been found), so exactly the lower copy survives.  Not in counting mode: a count added to a copy that
        // is retired afterwards would be lost.
        if (!COUNT) {
            const bool sp = !done && fresh && s < (int)slots;
            const unsigned long long spmask = __ballot(sp);
            if (spmask) {
                const uint64_t gaddr = sp ? reinterpret_cast<uint64_t>(key_ptr(st, b, (uint32_t)s)) : 0;  // (line, first empty slot)
                const uint64_t gprev = (uint64_t)(uint32_t)__shfl_up((int)(uint32_t)gaddr, 1) | ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(gaddr >> 32), 1) << 32);
                const bool gl = sp && (lane == 0 || gprev != gaddr);
                const unsigned long long glmask = __ballot(gl);
                const unsigned long long below = glmask & ((2ull << lane) - 1ull);
                const int g0 = below ? 63 - __builtin_clzll(below) : lane;  // first lane of this lane's group
                const unsigned long long ends = (glmask | ~spmask) & ~((2ull << g0) - 1ull);
                const int gend = ends ? __builtin_ctzll(ends) : 64;
                const uint32_t mm = sp ? min((uint32_t)(gend - g0), slots - (uint32_t)s) : 0u;  // slots the group goes for
                const int t = s + (lane - g0);
                const bool part = sp && t < (int)slots;
                uint64_t content = 0;
                bool won = false;
                if (part) {
                    const unsigned long long cur = atomicCAS(key_ptr(st, b, (uint32_t)t), (unsigned long long)EMPTY_KEY, (unsigned long long)key);
                    won = cur == EMPTY_KEY;
                    content = won ? key : cur;
                }
                int pos = -1;  // lowest slot of the group's range that holds this lane's key
                for (uint32_t i = 0; i < slots; ++i) {  // (wave-uniform trip count)
                    const uint64_t c = shfl64(content, min(g0 + (int)i, 63));
                    if (i < mm && pos < 0 && c == key) pos = s + (int)i;
                }
                if (sp) {
                    if (pos >= 0) {
                        const bool mine = part && won && pos == t;
                        if (part && won && pos != t) *reinterpret_cast<volatile unsigned long long *>(key_ptr(st, b, (uint32_t)t)) = TOMB_KEY;
                        resolve((uint32_t)pos, mine);
                    } else {
                        s += (int)mm;  // other keys all the way: on behind them, one slot per round from here
                    }
                }
            }
        }
        fresh = false;
#endif
        // one compare-and-swap round among the lanes that have a slot to try
        const bool want = !done && s >= 0 && s < (int)slots;
        unsigned long long *kp = want ? key_ptr(st, b, (uint32_t)s) : nullptr;
        const uint64_t addr = reinterpret_cast<uint64_t>(kp);
        const uint64_t prev = (uint64_t)(uint32_t)__shfl_up((int)(uint32_t)addr, 1) | ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(addr >> 32), 1) << 32);
        const bool leader = want && (lane == 0 || prev != addr);  // (a lane without a slot to try carries address 0)
        uint64_t content = 0;
        bool won = false;
        if (leader) {
            const unsigned long long cur = atomicCAS(kp, (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            won = cur == EMPTY_KEY;
            content = won ? key : cur;
        }
        const unsigned long long lmask = __builtin_amdgcn_ballot_w64(leader);
        if (__ballot(want)) {
            const unsigned long long below = lmask & ((2ull << lane) - 1ull);
            const int mine = below ? 63 - __builtin_clzll(below) : lane;  // the leader this lane follows
            const uint64_t got = (uint64_t)(uint32_t)__shfl((int)(uint32_t)content, mine) | ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(content >> 32), mine) << 32);
            if (want) {
                if (got == key) resolve((uint32_t)s, leader && won);
                else ++s;
            }
        }
    }
    return r;
}

template <int W_C, bool M64>
__global__ __launch_bounds__(64) void k_insert_tile(const SubTable st, int w, uint32_t bits, uint32_t count_mode,
                                                    const uint64_t *__restrict__ seqw, const uint32_t *__restrict__ nmw,
                                                    const uint32_t *__restrict__ has_n, const SeqDesc *__restrict__ sd,
                                                    const uint32_t *__restrict__ tile0, uint32_t ncontigs,
                                                    unsigned long long *__restrict__ counters, uint32_t max_probe) {
    constexpr int SLOTS = 8;  // 16-byte chunks of a 128-byte line (8 slots, or 16 bare keys in the split layout)
    constexpr int LDS_LINE_U4 = SLOTS + 1;
    constexpr int STAGE_ITERS = (PROBE_MAXRUN * SLOTS + 63) / 64;
    __shared__ uint64_t sw[PROBE_SEQW];
    __shared__ uint64_t rw[PROBE_SEQW + 1];
    __shared__ uint32_t nw[PROBE_SEQW];
    __shared__ uint32_t lines_w[PROBE_MAXRUN];
    __shared__ uint4 buf[((STAGE_ITERS * 64 + SLOTS - 1) / SLOTS) * LDS_LINE_U4];
    __shared__ uint32_t q_grp[INSERT_QCAP];
    __shared__ uint16_t q_pl[INSERT_QCAP];
    const int lane = threadIdx.x;
    const int k = (int)st.k;
    const bool split = st.layout == LAYOUT_SPLIT;  // (uniform)
    const bool update_only = (count_mode & 2u) != 0;  // pg_table_update_seqset: bits for keys already there, no new keys
    count_mode &= 1u;
    const uint32_t tile = blockIdx.x;
    uint32_t c = 0;  // contig of the tile: last c with tile0[c] <= tile (uniform binary search)
    {
        uint32_t lo = 0, hi = ncontigs;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (tile0[mid] <= tile) lo = mid;
            else hi = mid;
        }
        c = lo;
    }
    const SeqDesc s = sd[c];
    const uint32_t nkmers = (uint32_t)(s.len - (uint64_t)k + 1);
    const uint32_t tile_start = (tile - tile0[c]) * PROBE_TILE;
    const uint32_t npos = min((uint32_t)PROBE_TILE, nkmers - tile_start);
    const bool hasn = has_n[c] != 0;
    for (int i = lane; i < PROBE_SEQW; i += 64) {
        const uint64_t wi = (uint64_t)(tile_start >> 5) + i;
        const uint64_t wv = wi < s.nwords ? seqw[s.seq_off + wi] : 0ull;
        sw[i] = wv;
        rw[PROBE_SEQW - 1 - i] = pair_reverse64(~wv);
        nw[i] = (hasn && wi < s.nwords) ? nmw[s.seq_off + wi] : 0u;
    }
    if (lane == 0) rw[PROBE_SEQW] = 0;  // (a window's three dwords may reach one word past the end)
    __syncthreads();
    const uint64_t kmask = W_C ? (((uint64_t)(k == 32 ? ~0u : ((1u << (2 * k - 32)) - 1u)) << 32) | 0xFFFFFFFFull) : kmer_mask(k);  // (as in k_probe)
    const uint32_t rcb = PROBE_SEQ_BASES - (uint32_t)k;
    constexpr int HALO = W_C ? W_C - 1 : 0;
    constexpr int STRIDE = 64 - HALO;
    const uint32_t m = W_C ? (uint32_t)k - W_C + 1 : 0;
    const uint64_t mm64 = (m >= 32) ? ~0ull : ((1ull << (2 * (m ? m : 1))) - 1);
    uint32_t qn = 0, claimed = 0;  // (wave-uniform)
    bool overflowed = false;

    auto drain = [&]() {  // the queued (absent) keys through the claiming insert, 64 at a time
        __syncthreads();
        for (uint32_t e0 = 0; e0 < qn; e0 += 64) {
            const uint32_t e = e0 + lane;
            const bool valid = e < qn;
            const uint32_t ec = valid ? e : qn - 1;
            const uint64_t key = canonical_from_le(extract_bases32(reinterpret_cast<const uint32_t *>(sw), q_pl[ec]), k);
            const int r = update_only ? wave_insert_batch<false, false>(st, valid, key, q_grp[ec], w, bits, max_probe, lane)
                          : count_mode ? wave_insert_batch<true>(st, valid, key, q_grp[ec], w, bits, max_probe, lane)
                                       : wave_insert_batch<false>(st, valid, key, q_grp[ec], w, bits, max_probe, lane);
            overflowed |= __ballot(r < 0) != 0;
            claimed += (uint32_t)__popcll(__ballot(r > 0));
        }
        qn = 0;
        __syncthreads();
    };

    uint32_t adv = STRIDE;  // positions the batch covers (k_probe's CUT: a batch ends in front of its (PROBE_MAXRUN + 1)-th run)
    for (uint32_t b = 0; b < npos; b += adv) {
        const int32_t pl = (int32_t)(b + lane) - HALO;
        const bool inrange = pl >= (int32_t)b && pl < (int32_t)npos;
        const uint32_t pq = (uint32_t)max(pl, 0);
        const uint64_t X = extract_bases32(reinterpret_cast<const uint32_t *>(sw), pq) & kmask;
        const uint64_t B = revcomp_window(rw, X, pq, k, kmask, rcb);
        const uint64_t key = ~(X > B ? X : B) & kmask;
        bool act = inrange;
        if (hasn) act = act && (extract_nmask(nw, pq, k) == 0);
        uint32_t grp;
        if (W_C) {
            const uint32_t off = (uint32_t)(pl + HALO) - pq;
            if constexpr (!M64) {
                const uint32_t mm32 = (uint32_t)mm64;
                const uint32_t fa = __builtin_amdgcn_alignbit((uint32_t)(X >> 32), (uint32_t)X, 2 * off) & mm32;
                const uint32_t fr = __builtin_amdgcn_alignbit((uint32_t)(B >> 32), (uint32_t)B, 2 * (HALO - off)) & mm32;
                grp = mz_order(min(fa, fr));
            } else {
                const uint64_t fa = (X >> (2 * off)) & mm64;
                const uint64_t fr = (B >> (2 * (HALO - off))) & mm64;
                grp = mmer_rank(fa < fr ? fa : fr);
            }
            grp = sliding_min<W_C ? W_C : 1>(grp);
        } else {
            grp = group_of_key(key);
        }
        const uint32_t line = home_of_group(grp, st.nbuckets);
        const uint32_t prev_line = lane_up1(line);
        unsigned long long amask = __builtin_amdgcn_ballot_w64(act);  // (run starts by lane-mask arithmetic on the scalar unit, as in k_probe)
        unsigned long long lmask = amask & (~(amask << 1) | __builtin_amdgcn_ballot_w64(line != prev_line));
        const uint32_t rid = lanes_le_index(lmask, 0u);
        if constexpr (PG_PROBE_CUT && W_C != 0) {  // one staging step per batch: the lanes behind the PROBE_MAXRUN-th run are the next batch's
            const unsigned long long keep = __builtin_amdgcn_sicmp((int32_t)rid, PROBE_MAXRUN, 40 /* signed < */);
            amask &= keep;
            lmask &= keep;
            act = __builtin_amdgcn_inverse_ballot_w64(amask);
            adv = (uint32_t)__popcll(keep) - HALO;
        }
        const bool leader = __builtin_amdgcn_inverse_ballot_w64(lmask);
        const uint32_t nruns = (uint32_t)__popcll(lmask);
        bool found = false, full = true;  // full: the staged home line had no empty slot (or was not reached)
        for (uint32_t r0 = 0; r0 < nruns; r0 += PROBE_MAXRUN) {
            const uint32_t nl = min((uint32_t)PROBE_MAXRUN, nruns - r0);
            if (leader && rid - r0 < nl) lines_w[rid - r0] = line;
            __syncthreads();
            uint4 v[STAGE_ITERS];
#pragma unroll
            for (int it = 0; it < STAGE_ITERS; ++it) {
                const uint32_t ls = min((uint32_t)(it * (64 / SLOTS) + lane / SLOTS), nl - 1u);
                v[it] = *reinterpret_cast<const uint4 *>(st.buckets + (((uint64_t)lines_w[ls] * 128u) | ((lane % SLOTS) * 16u)));
            }
#pragma unroll
            for (int it = 0; it < STAGE_ITERS; ++it) {
                const uint32_t idx = it * 64 + lane;
                buf[(idx / SLOTS) * LDS_LINE_U4 + (idx % SLOTS)] = v[it];
            }
            __syncthreads();
            if (act && rid - r0 < nl) {
                const uint4 *ln = buf + (rid - r0) * LDS_LINE_U4;
                uint32_t *mp = nullptr;
                uint32_t cur = 0;
                if (st.layout == LAYOUT_INLINE) {  // (uniform)
                    uint32_t slot1;
                    full = scan_keys_inl(ln, key, st.slots, slot1) < 0;
                    if (slot1) {
                        found = true;
                        mp = mask_ptr(st, line, slot1 - 1u, (uint32_t)w);
                        cur = *reinterpret_cast<const volatile uint32_t *>(mp);
                    }
                } else if (split) {
                    uint32_t slot1;
                    full = scan_keys16_lds<true>(ln, key, slot1) < 0;
                    if (slot1) {
                        found = true;
                        mp = reinterpret_cast<uint32_t *>(st.masks) + ((uint64_t)line * SPLIT_KEYS + (slot1 - 1u)) * st.W + (uint32_t)w;
                        cur = *reinterpret_cast<const volatile uint32_t *>(mp);
                    }
                } else {
                    uint32_t m0, m1;
                    // (the slot's byte offset is what is needed here: scan the keys like scan_line_lds does)
                    uint64_t kk[8];
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl) kk[sl] = *reinterpret_cast<const uint64_t *>(ln + sl);
                    unsigned long long e[8];
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl) e[sl] = __builtin_amdgcn_ballot_w64(kk[sl] == key);
                    {  // a snapshot taken during this launch may show a key twice (a claim about to be retired, see
                       // wave_insert_batch): the LOWEST slot counts — a lane's match there masks its later ones (scalar unit)
                        unsigned long long seen = e[0];
#pragma unroll
                        for (int sl = 1; sl < 8; ++sl) {
                            const unsigned long long m = e[sl];
                            e[sl] = m & ~seen;
                            seen |= m;
                        }
                    }
                    const unsigned long long b0 = e[1] | e[3] | e[5] | e[7], b1 = e[2] | e[3] | e[6] | e[7], b2 = e[4] | e[5] | e[6] | e[7];
                    full = kk[7] != EMPTY_KEY;
                    (void)m0;
                    (void)m1;
                    if (__builtin_amdgcn_inverse_ballot_w64(b0 | b1 | b2 | e[0])) {
                        found = true;
                        const uint32_t off = (__builtin_amdgcn_inverse_ballot_w64(b0) ? 16u : 0u) | (__builtin_amdgcn_inverse_ballot_w64(b1) ? 32u : 0u) |
                                             (__builtin_amdgcn_inverse_ballot_w64(b2) ? 64u : 0u);
                        cur = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(ln) + off + 8 + 4 * w);
                        mp = reinterpret_cast<uint32_t *>(st.buckets + (uint64_t)line * 128u + off + 8 + 4 * w);
                    }
                }
                if (found) {
                    if (count_mode) {
                        if (cur < 0xFFFFFF00u) atomicAdd(mp, bits);
                    } else if ((cur & bits) != bits) {
                        *reinterpret_cast<volatile uint32_t *>(mp) = cur | bits;
                    }
                }
            }
            __syncthreads();
        }
        // absent from the snapshot of its line (or the line was not reached): queue for the claiming insert
        // (update-only: a key missing from a home line that is not full is not in the table — lines fill front to back)
        const bool todo = act && !found && (!update_only || full);
        const unsigned long long tmask = __builtin_amdgcn_ballot_w64(todo);
        if (tmask) {
            if (qn + 64 > (uint32_t)INSERT_QCAP) drain();
            if (todo) {
                const uint32_t slot = lanes_le_index(tmask, qn);
                q_grp[slot] = grp;
                q_pl[slot] = (uint16_t)pl;
            }
            qn += (uint32_t)__popcll(tmask);
        }
    }
    if (qn) drain();
    if (lane == 0) {
        if (claimed) atomicAdd(&counters[0], (unsigned long long)claimed);
        if (overflowed) atomicOr(reinterpret_cast<unsigned int *>(&counters[1]), 1u);
    }
}

#if PG_MAIN_PART  // ======== every other kernel of the file: part 0 (or the one unit) only ========
// ---------------------------------------------------------------------------
// statistics from finished rows: bitmap.100, per-bin popcount histogram, column sums.
// A workgroup (256 threads) walks a CONTIGUOUS range of tiles (PT consecutive positions per
// thread and tile); histogram counters stay in LDS until the bin changes and column sums until
// the end, so that global atomics on the few shared counters stay rare.
// (also the second half of the genome-sharded mode: rows combined over xGMI first)
// ---------------------------------------------------------------------------
// A workgroup keeps the histograms of EPI_MAXB consecutive bins in LDS at a time (rows relative to
// cur_row0): contigs of a few kb .. Mb have bins of nkmers/100 positions, far shorter than a tile.
// The window is 16..128 bins wide, as many as about 12 KB of LDS hold at N + 1 counters per bin (chosen by the launcher,
// handed over in bits 8..15 of `flags`): with 16 bins a contig of a few kb — bins of 50 rows, a tile spans 11 of them —
// flushed its window to global memory after nearly every tile (4 x 100 Mb in 20 000 contigs: 1.45 ms for 4 x 10^8 rows).
constexpr uint32_t EPI_MAXB = 16;  // (the least)
__host__ __device__ __forceinline__ uint32_t epi_maxb_for(uint32_t ngenomes) {
    const uint32_t b = 3072u / (ngenomes + 1u);
    return b < EPI_MAXB ? EPI_MAXB : (b > 128u ? 128u : b);
}
__host__ __device__ __forceinline__ uint32_t epi_minbin(uint32_t maxb) { return ((uint32_t)PROBE_TILE + maxb - 3u) / (maxb - 2u); }  // a tile then spans <= maxb bins
// column sums: one ballot + popcount per genome bit, accumulated in LDS by lane 0
__device__ __forceinline__ void colsum_word(uint32_t wv, uint32_t d, uint32_t N, uint32_t *cs, int lane) {
    const uint32_t ng = min(32u, N - 32 * d);
    for (uint32_t bit = 0; bit < ng; ++bit) {
        const unsigned long long bal = __ballot((wv >> bit) & 1u);
        if (lane == 0 && bal) atomicAdd(&cs[32 * d + bit], (uint32_t)__popcll(bal));
    }
}
// wave-aggregated histogram of (bin, popcount): LDS for the first EPI_MAXB bins from bin0, global beyond
__device__ __forceinline__ void hist_position(bool active, uint32_t pos, uint32_t popc, uint32_t N, uint32_t binlen,
                                              uint32_t bin0, uint32_t bin0_start, uint32_t rel_base, uint32_t *hist,
                                              uint32_t *bins, uint64_t bin_off, int lane, uint32_t maxb) {
    if (popc > N) popc = N;  // junk bits beyond ngenomes: the reference indexes out of bounds here
    const uint32_t dpos = pos - bin0_start;
    const uint32_t rel = (binlen >= (uint32_t)PROBE_TILE) ? (dpos >= binlen ? 1u : 0u) : dpos / binlen;
    const uint32_t hk = rel * (N + 1) + popc;
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t lk = __shfl(hk, leader);
        const unsigned long long mk = __ballot(active && hk == lk) & todo;
        if (lane == leader) {
            const uint32_t cnt = (uint32_t)__popcll(mk);
            if (rel_base + rel < maxb) atomicAdd(&hist[rel_base * (N + 1) + hk], cnt);
            else atomicAdd(&bins[(bin_off + bin0 + rel) * (uint64_t)(N + 1) + popc], cnt);
        }
        todo &= ~mk;
    }
}

// four consecutive rows of NB bytes = NB aligned 32-bit words (a thread's first row starts at a
// multiple of 4 rows): load the words (all in flight together), cut the rows out with static shifts
template <int NB>
__device__ __forceinline__ void load_row_words(const uint8_t *g4, uint32_t raw[8]) {
#pragma unroll
    for (int i = 0; i < NB; ++i) raw[i] = reinterpret_cast<const uint32_t *>(g4)[i];
}
template <int NB>
__device__ __forceinline__ void cut4_rows(const uint32_t raw[8], uint32_t w0[4], uint32_t w1[4]) {
    uint32_t w[NB + 1];
#pragma unroll
    for (int i = 0; i < NB; ++i) w[i] = raw[i];
    w[NB] = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        constexpr uint64_t keep = NB >= 8 ? ~0ull : ((1ull << (8 * (NB & 7))) - 1);
        const int off = j * NB, idx = off >> 2, sh = 8 * (off & 3);
        uint64_t v = (uint64_t)w[idx] >> sh;
        if (idx + 1 <= NB) v |= (uint64_t)w[idx + 1] << (32 - sh);
        if (sh && idx + 2 <= NB) v |= (uint64_t)w[idx + 2] << (64 - sh);
        v &= keep;
        w0[j] = (uint32_t)v;
        w1[j] = (uint32_t)(v >> 32);
    }
}
#ifndef PG_EPI_MIN_TILES
#define PG_EPI_MIN_TILES 128
#endif
constexpr int EPI_THREADS = PROBE_TILE / 4;

static_assert(EPI_THREADS >= 64 && EPI_THREADS % 64 == 0, "PROBE_TILE must be a multiple of 256");

__device__ __forceinline__ void flush_hist(uint32_t N, uint32_t *hist, uint32_t *bins, uint64_t bin_row0, int tid, uint32_t maxb) {
    for (uint32_t i = tid; i < maxb * (N + 1); i += EPI_THREADS) {
        const uint32_t hv = hist[i];
        if (hv) {
            const uint32_t rel = i / (N + 1), pc2 = i - rel * (N + 1);
            atomicAdd(&bins[(bin_row0 + rel) * (uint64_t)(N + 1) + pc2], hv);
            hist[i] = 0;
        }
    }
}

// MODE 0: one-byte rows (N <= 8), 1: rows of 2..8 bytes (N <= 64); wider rows go through
// k_epilogue_chunks below.  One instantiation per mode so that each carries only its own accumulators
// in registers.
// The tiles a statistics workgroup takes: a contiguous range, cut in units of `gt` tiles (the group paths' granule).
// ranges == NULL: the launch covers tiles [0, ntiles), split evenly over the grid.  Otherwise (a CHUNK of a run whose
// probe launches are interleaved with their statistics passes, pg_api.hip: anchor_run): the launch covers the tile ranges
// ranges[0 .. gridDim.x / wpr) — what one slice of the co-schedule touches of every genome — with wpr workgroups each.
struct EpiRange {
    uint32_t begin, end;
};
__device__ __forceinline__ EpiRange epi_range(uint32_t gt, uint32_t ntiles, const uint2 *ranges, uint32_t wpr) {
    uint32_t lo = 0, hi = ntiles, j = blockIdx.x, n = gridDim.x;
    if (ranges) {
        const uint2 rg = ranges[blockIdx.x / wpr];
        lo = rg.x;
        hi = rg.y;
        j = blockIdx.x % wpr;
        n = wpr;
    }
    const uint32_t ngroups = (hi - lo + gt - 1) / gt;
    EpiRange e;
    e.begin = lo + gt * (uint32_t)((uint64_t)ngroups * j / n);
    e.end = min(hi, lo + gt * (uint32_t)((uint64_t)ngroups * (j + 1) / n));
    return e;
}

// (one-byte rows: held to the registers of 7 waves per SIMD — 72 VGPRs and 20 bytes of scratch on a cold path instead of 79,
// 96 SGPRs instead of 106: 6 -> 7 workgroups per CU, the pass 0.362 -> 0.353 ms on 8 x 10^8 rows, 0.616 -> 0.588 on
// 1.6 x 10^9; 8 waves (64 VGPRs, 40 bytes of scratch) are slower, 0.392; profiles/r4b_ab_epilogue_waves.txt)
#ifndef PG_EPI_HS
#define PG_EPI_HS 1  // k_epilogue, rows of 2..8 bytes: column sums through eight counter planes behind a Harley-Seal tree
#endif
#ifndef PG_EPI_STREAK
#define PG_EPI_STREAK 1  // k_epilogue, one-byte rows: the bookkeeping of whole one-bin groups once per streak (see the tile loop)
#endif
#ifndef PG_EPI_WAIT_HERE
#define PG_EPI_WAIT_HERE 1  // k_epilogue, one-byte rows: see the group path's prefetch
#endif
#ifndef PG_EPI_WAVES0
#define PG_EPI_WAVES0 7
#endif
// (rows of 2 to 5 bytes: 6 waves per SIMD instead of the 4-5 their 96-106 VGPRs allowed — 12 x 60 Mb 0.578 -> 0.509 ms,
// 20 x 40 Mb 0.647 -> 0.57, 27 x 40 Mb and 40 x 30 Mb 2-6 %; 8-byte rows lose with it, 1.95 -> 2.1-2.3 ms, and stay as they were)
#ifndef PG_EPI_WAVES1
#define PG_EPI_WAVES1 6
#endif
#ifndef PG_EPI_SGPRS0
#define PG_EPI_SGPRS0 96
#endif
template <int MODE, int NBT>  // NBT = bytes per row (1..8): one instantiation, and one register allocation, per width
__global__ __launch_bounds__(EPI_THREADS, (MODE == 0 ? PG_EPI_WAVES0 : NBT <= 5 ? PG_EPI_WAVES1 : 1))
__attribute__((amdgpu_num_sgpr(PG_EPI_SGPRS0))) void k_epilogue(uint32_t N, const AnchorDesc *__restrict__ ad,
                                                          const uint32_t *__restrict__ tile_contig, uint32_t ntiles,
                                                          const uint8_t *__restrict__ out1, uint8_t *__restrict__ out100,
                                                          uint32_t *__restrict__ bins,
                                                          unsigned long long *__restrict__ colsums, uint32_t flags,
                                                          const uint2 *__restrict__ ranges, uint32_t wpr) {
    extern __shared__ uint4 smem[];
    constexpr int PT = 4;  // rows per thread and tile: EPI_THREADS = PROBE_TILE / 4 threads per workgroup
    constexpr bool WIDE = MODE == 1;
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr uint32_t nbytes = NBT;
    const uint32_t Nw = N;
    const uint32_t ndbs = (N + 31) / 32;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    const uint32_t MAXB = max(EPI_MAXB, (flags >> 8) & 0xFFu), MINBIN = epi_minbin(MAXB);  // bins in the LDS window (launcher's choice)
    uint32_t *cs = hist + ((MAXB * (N + 1) + 3) & ~3u);
    for (uint32_t i = tid; i < MAXB * (N + 1); i += EPI_THREADS) hist[i] = 0;
    for (uint32_t i = tid; i < N; i += EPI_THREADS) cs[i] = 0;
    __syncthreads();
    const bool want_cs = (flags & 1u) != 0;
    const bool want100 = (flags & 2u) == 0;  // bit 1: the low-resolution rows are taken by k_lowres (step != 100)
    // contiguous tile ranges, cut in units of the group paths' 4 tiles (16 rows per thread; one-byte rows: 8 tiles, 32 rows)
    constexpr uint32_t GT = MODE == 0 ? 8u : 4u;
    const EpiRange er = epi_range(GT, ntiles, ranges, wpr);
    const uint32_t t_begin = er.begin, t_end = er.end;
    uint64_t cur_row0 = ~0ull;  // bins row the accumulators currently stand for
    uint32_t cur_c = ~0u;
    AnchorDesc a;
    a.out_off = a.out100_off = a.bin_off = 0;
    a.nkmers = a.binlen = a.tile0 = a.nbins = 0;
    const uint32_t p0 = tid * PT;
    // fast path (N <= 8) per-thread accumulators, reduced over the workgroup only when the bin
    // changes / at the end: 9 popcount classes as 7-bit fields of one u64 (spilled to the u32 counters
    // below every 31 tiles), 8 column counters
    unsigned long long hacc = 0;
    // ... and of the group path: thr[i] = rows seen with MORE than i bits set (the histogram classes are their
    // differences), grows = rows seen
    uint32_t thr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, grows = 0;
    uint32_t cacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t since_spill = 0;
    uint32_t next_packed = 0;  // software prefetch of the next tile's rows
    bool next_valid = false;
    uint4 wp_a = make_uint4(0, 0, 0, 0), wp_b = make_uint4(0, 0, 0, 0);  // ... and of 4- / 8-byte rows on the wide path
    bool wp_valid = false;
    // wide path (8 < N <= 64): column sums in three levels, all in registers until the very end —
    //  L1  per-thread VERTICAL counters: bit g of plane p is bit p of the number of rows seen with
    //      genome g set; the 4 rows of a tile enter through carry-save adders (12 ALU ops per 4 rows);
    //  L2  every 12 rows the planes are transposed into nibbles and added to byte-sliced
    //      accumulators: byte b of bacc[w][q] counts genome 32w + 8b + q (up to 252 rows);
    //  L3  a halving exchange over the wave (17 shuffles per word) leaves each total in one lane,
    //      which adds it to the workgroup's LDS counters.
#if PG_EPI_HS
    // (round 4: L1 and L2 as in k_epilogue_chunks — EIGHT bit planes per word behind a Harley-Seal carry-save tree: four rows
    // enter the ones / twos planes per call (9 instructions), their carry of weight 4 is held back every other call and enters
    // the fours plane with the next one (3), likewise the eights, and every 16 rows one carry ripples through the upper four
    // planes (8): 3.3 instructions per row and word where four planes emptied every 12 rows into byte-sliced accumulators took
    // 8.3 — a quarter of the 22 vector instructions per row that bound the pass for rows of 2 and 3 bytes
    // (profiles/r4e_pmc_n12_n27.txt); the planes are emptied every 240 rows straight into L3's exchange)
    uint32_t vp[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}}, pf[2] = {0, 0}, pe[2] = {0, 0};
    uint32_t vrows = 0;  // rows in the planes (block-uniform, a multiple of 4): the carries held back follow from it
    constexpr uint32_t VROWS_FLUSH = 240;
    auto ripple = [&](uint32_t (&p)[8], uint32_t cw, int q0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q >= q0) {
                const uint32_t n = p[q] & cw;
                p[q] ^= cw;
                cw = n;
            }
    };
    auto vadd4 = [&](int ws, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3) __attribute__((always_inline)) {
        const bool odd4 = (vrows & 4u) != 0, odd8 = (vrows & 8u) != 0;  // (the caller advances vrows once all words of the four rows are in)
        uint32_t (&p)[8] = vp[ws];
        const uint32_t x = p[0];
        const uint32_t t1 = x ^ r0, s1 = t1 ^ r1, ca = (t1 & r1) | (~t1 & x);      // x + r0 + r1
        const uint32_t t2 = s1 ^ r2, s2 = t2 ^ r3, cb = (t2 & r3) | (~t2 & s1);    // .. + r2 + r3
        p[0] = s2;
        const uint32_t y = p[1];
        const uint32_t t3 = y ^ ca, cc = (t3 & cb) | (~t3 & y);                     // twos + ca + cb -> a carry of weight 4
        p[1] = t3 ^ cb;
        if (!odd4) {
            pf[ws] = cc;
            return;
        }
        const uint32_t z = p[2], t4 = z ^ pf[ws], c8 = (t4 & cc) | (~t4 & z);      // fours + both carries -> weight 8
        p[2] = t4 ^ cc;
        if (!odd8) {
            pe[ws] = c8;
            return;
        }
        const uint32_t u = p[3], t5 = u ^ pe[ws], c16 = (t5 & c8) | (~t5 & u);     // eights + both carries -> weight 16
        p[3] = t5 ^ c8;
        ripple(p, c16, 4);
    };
    auto vflush = [&]() {  // L1 -> L3, wave-uniform call sites only
        const bool odd4 = (vrows & 4u) != 0, odd8 = (vrows & 8u) != 0;  // carries still held back (a flush between whole 16-row blocks)
        for (uint32_t ws = 0; ws < ndbs && ws < 2; ++ws) {
            if (odd4) ripple(vp[ws], pf[ws], 2);
            if (odd8) ripple(vp[ws], pe[ws], 3);
            uint32_t R[16];
#pragma unroll
            for (int q = 0; q < 8; ++q) {  // byte b of v counts genome 32 ws + 8 b + q (up to 255 rows)
                uint32_t v = 0;
#pragma unroll
                for (int pl = 0; pl < 8; ++pl) v |= ((vp[ws][pl] >> q) & 0x01010101u) << pl;
                R[2 * q] = v & 0x00FF00FFu;
                R[2 * q + 1] = (v >> 8) & 0x00FF00FFu;
            }
#pragma unroll
            for (int pl = 0; pl < 8; ++pl) vp[ws][pl] = 0;
#pragma unroll
            for (int half = 8, bit = 32; half >= 1; half >>= 1, bit >>= 1) {
                const bool up = (lane & bit) != 0;
#pragma unroll
                for (int i = 0; i < half; ++i) {
                    const uint32_t send = up ? R[i] : R[i + half];
                    const uint32_t keep = up ? R[i + half] : R[i];
                    R[i] = keep + (uint32_t)__shfl_xor((int)send, bit);
                }
            }
            R[0] += (uint32_t)__shfl_xor((int)R[0], 2);
            R[0] += (uint32_t)__shfl_xor((int)R[0], 1);
            if ((lane & 3) == 0) {  // this lane holds register (lane >> 2): q = idx / 2, odd idx = bytes 1 and 3
                const uint32_t idx = (uint32_t)lane >> 2;
                const uint32_t g0 = 32 * ws + (idx >> 1) + ((idx & 1) ? 8u : 0u);
                if (g0 < Nw && (R[0] & 0xFFFFu)) atomicAdd(&cs[g0], R[0] & 0xFFFFu);
                if (g0 + 16 < Nw && (R[0] >> 16)) atomicAdd(&cs[g0 + 16], R[0] >> 16);
            }
        }
        vrows = 0;
    };
    [[maybe_unused]] uint32_t brounds = 0;
    auto wave_colsums = [&]() {};
#else
    constexpr uint32_t VROWS_FLUSH = 12;
    uint32_t vp[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    uint32_t bacc[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
    uint32_t vrows = 0, brounds = 0;
    auto vadd4 = [&](int ws, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3) {
        const uint32_t x = vp[ws][0];
        const uint32_t t1 = x ^ r0, s1 = t1 ^ r1, ca = (t1 & r1) | (~t1 & x);      // x + r0 + r1
        const uint32_t t2 = s1 ^ r2, s2 = t2 ^ r3, cb = (t2 & r3) | (~t2 & s1);    // .. + r2 + r3
        vp[ws][0] = s2;
        const uint32_t y = vp[ws][1];
        const uint32_t t3 = y ^ ca, cc = (t3 & cb) | (~t3 & y);                     // twos + ca + cb
        vp[ws][1] = t3 ^ cb;
        const uint32_t c4 = vp[ws][2] & cc;
        vp[ws][2] ^= cc;
        vp[ws][3] ^= c4;
    };
    auto wave_colsums = [&]() {  // L3, wave-uniform call sites only
        for (uint32_t ws = 0; ws < ndbs && ws < 2; ++ws) {
            uint32_t R[16];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                R[2 * q] = bacc[ws][q] & 0x00FF00FFu;
                R[2 * q + 1] = (bacc[ws][q] >> 8) & 0x00FF00FFu;
                bacc[ws][q] = 0;
            }
#pragma unroll
            for (int half = 8, bit = 32; half >= 1; half >>= 1, bit >>= 1) {
                const bool up = (lane & bit) != 0;
#pragma unroll
                for (int i = 0; i < half; ++i) {
                    const uint32_t send = up ? R[i] : R[i + half];
                    const uint32_t keep = up ? R[i + half] : R[i];
                    R[i] = keep + (uint32_t)__shfl_xor((int)send, bit);
                }
            }
            R[0] += (uint32_t)__shfl_xor((int)R[0], 2);
            R[0] += (uint32_t)__shfl_xor((int)R[0], 1);
            if ((lane & 3) == 0) {  // this lane holds register (lane >> 2): q = idx / 2, odd idx = bytes 1 and 3
                const uint32_t idx = (uint32_t)lane >> 2;
                const uint32_t g0 = 32 * ws + (idx >> 1) + ((idx & 1) ? 8u : 0u);
                if (g0 < Nw && (R[0] & 0xFFFFu)) atomicAdd(&cs[g0], R[0] & 0xFFFFu);
                if (g0 + 16 < Nw && (R[0] >> 16)) atomicAdd(&cs[g0 + 16], R[0] >> 16);
            }
        }
        brounds = 0;
    };
    auto vflush = [&]() {  // L1 -> L2, wave-uniform call sites only
        for (uint32_t ws = 0; ws < ndbs && ws < 2; ++ws) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t M = 0x11111111u;
                const uint32_t nib = ((vp[ws][0] >> j) & M) | (((vp[ws][1] >> j) & M) << 1) |
                                     (((vp[ws][2] >> j) & M) << 2) | (((vp[ws][3] >> j) & M) << 3);
                bacc[ws][j] += nib & 0x0F0F0F0Fu;
                bacc[ws][4 + j] += (nib >> 4) & 0x0F0F0F0Fu;
            }
#pragma unroll
            for (int pln = 0; pln < 4; ++pln) vp[ws][pln] = 0;
        }
        vrows = 0;
        if (++brounds == 21) wave_colsums();  // 21 x 12 rows: the byte counters are about to fill
    };

#endif
    auto spill = [&]() {  // the per-tile path's classes (7-bit fields of hacc) join the thresholds: thr[i] += rows of class > i
        uint32_t run = 0;
#pragma unroll
        for (int v = 8; v >= 1; --v) {
            run += (uint32_t)(hacc >> (7 * v)) & 127u;
            thr[v - 1] += run;
        }
        grows += run + ((uint32_t)hacc & 127u);
        hacc = 0;
        since_spill = 0;
    };
    auto reduce_hist = [&]() {  // per-thread thresholds -> classes -> LDS histogram (bin-relative row 0)
        if constexpr (MODE != 0) return;
        spill();
        uint32_t hc[9];
        hc[0] = grows - thr[0];
#pragma unroll
        for (int v = 1; v < 8; ++v) hc[v] = thr[v - 1] - thr[v];
        hc[8] = thr[7];
#pragma unroll
        for (int v = 8; v >= 1; --v)  // (junk bits beyond ngenomes count as class N, as on the per-tile path)
            if ((uint32_t)v > N) {
                hc[v - 1] += hc[v];
                hc[v] = 0;
            }
        grows = 0;
#pragma unroll
        for (int v = 0; v < 8; ++v) thr[v] = 0;
#pragma unroll
        for (int v = 0; v < 9; ++v)
            if ((uint32_t)v <= N && hc[v]) atomicAdd(&hist[v], hc[v]);
    };

    // column sums are kept per contig (colsums[contig][N]): register / LDS accumulators are emptied
    // whenever the workgroup's tile range moves on to another contig (block-uniform, rare)
    auto flush_colsums = [&](uint32_t contig) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int gb = 0; gb < 8; ++gb) {
                if ((uint32_t)gb < N && cacc[gb]) atomicAdd(&cs[gb], cacc[gb]);
                cacc[gb] = 0;
            }
        }
        if constexpr (WIDE) {
            if (vrows) vflush();
            if (brounds) wave_colsums();
        }
        __syncthreads();
        for (uint32_t i = tid; i < N; i += EPI_THREADS) {
            const uint32_t v = cs[i];
            if (v) {
                atomicAdd(&colsums[(uint64_t)contig * N + i], (unsigned long long)v);
                cs[i] = 0;
            }
        }
        __syncthreads();
    };

    // The group path of the 2..8-byte rows adds ONE to an LDS counter per row, and the lanes of a wave mostly ask for the
    // same few counters (the popcount classes near N): the LDS works a wave's atomic off one lane per cycle and address —
    // 0.9-1.0 ps per row whatever the row width, twice what 2-byte rows need otherwise.  While the groups lie inside one
    // or two long bins (any contig of more than 200 kb) the window is therefore used as EPI_REPL copies of those two bin
    // rows, lane l adding to copy l % EPI_REPL (copy c at c * repl_stride, an odd stride: the copies of one class sit in
    // different banks); `unreplicate` folds the copies into the window's ordinary form — rows 0 and 1 — before anything
    // else reads or flushes it.  `repl` is block-uniform.
    constexpr uint32_t EPI_REPL = 8;  // (16 copies: no further gain, profiles/r3_ab_stats_hist_copies.txt)
    const uint32_t repl_stride = (2u * (N + 1u)) | 1u;  // (EPI_REPL * repl_stride <= MAXB * (N + 1): MAXB >= 47 for N <= 64)
    bool repl = false;
    auto unreplicate = [&]() {
        if constexpr (MODE != 1) return;
        if (!repl) return;
        __syncthreads();
        for (uint32_t i = tid; i < 2u * (N + 1u); i += EPI_THREADS) {
            uint32_t v = hist[i];
#pragma unroll
            for (uint32_t cpy = 1; cpy < EPI_REPL; ++cpy) {
                v += hist[cpy * repl_stride + i];
                hist[cpy * repl_stride + i] = 0;
            }
            hist[i] = v;
        }
        __syncthreads();
        repl = false;
    };
    uint4 gq_next = make_uint4(0, 0, 0, 0), gq_next2 = make_uint4(0, 0, 0, 0);  // group path: prefetched rows of the next group
    bool gq_valid = false;
    // (one-byte rows) STREAK: whole one-bin groups known to follow the current one inside its bin, contig and tile range.  This
    // pass is bound by its SCALAR instructions — a SIMD issues at most one per four cycles, and the block-uniform bookkeeping of
    // a group (two tile_contig look-ups, six divisions by the bin length at 11 instructions each, the window checks) came to
    // 330 of them against 195 vector instructions (100 dummy s_add per group: +0.07 ms on 8 x 10^8 rows, 100 dummy VALU: +0.04;
    // profiles/r4e_stats_scalar_bound.txt).  Worked out ONCE when a group turns out whole and inside one bin; the groups of
    // the streak then take nothing of that: same contig, same bin row, no window check, the next group's prefetch certain.
    uint32_t streak = 0, nk_ba = 0, nk_bz = 0;
    int nk_kind = 0;
    for (uint32_t tile = t_begin; tile < t_end; ++tile) {
        uint32_t c = cur_c;
        if (!(MODE == 0 && PG_EPI_STREAK && (streak || gq_valid))) {  // (a group the one before has announced lies in its contig)
            c = tile_contig[tile];
            if (c != cur_c) {  // block-uniform; consecutive tiles nearly always share their contig
                if (want_cs && cur_c != ~0u) flush_colsums(cur_c);
                a = ad[c];
                cur_c = c;
            }
        }
        // ---- group path (N <= 8): 8 full tiles of one contig inside one bin = 32 one-byte rows per thread in two
        // 16-byte loads, worked on BIT-SLICED: a three-stage butterfly between the 8 words regroups their 256 bits so
        // that word g holds bit g of all 32 rows (same row, same bit position in every word: 4 instructions per word
        // pair and stage); an 8-input sorting network on those planes (19 compare-exchanges = AND / OR pairs) turns
        // them into thresholds "row has more than i bits"; popcounts of the planes are the column sums, popcounts of
        // the thresholds the cumulative histogram.  3.2 instructions per row, where one-hot adds per row took 13 ----
        // Two kinds of group: (1) all 4096 rows inside ONE bin — thresholds and rows counted in registers, reduced when the
        // bin changes; (2) several bins (contigs of a few kb .. Mb have bins of nkmers / 100 rows): bins of at least 32
        // rows, so that a thread's 32 rows meet at most one bin boundary, and all of the group's bins inside the LDS window —
        // the thread splits its threshold popcounts at the boundary (a mask over the planes' bit positions) and adds the
        // classes of its one or two bins to the window with up to 9 LDS atomics each, where the per-tile path does one per row.
        // A group need not be whole: a contig's last tiles (and a workgroup's last ones) form a group of fewer rows — a thread
        // then holds nv < 32 valid rows (possibly none) and masks the planes with the same kind of position mask.
        if constexpr (MODE == 0) {
            const uint32_t ts = (tile - a.tile0) * PROBE_TILE;
            const uint32_t span = 8u * PROBE_TILE;
            // rows of this contig from ts on that belong to this workgroup's tile range, at most a whole group's
            auto rows_at = [&](uint32_t tl, uint32_t t0) -> uint32_t {  // (block-uniform)
                if (tl >= t_end || tile_contig[tl] != c || t0 >= a.nkmers) return 0u;
                return min(min(span, a.nkmers - t0), (t_end - tl) * (uint32_t)PROBE_TILE);
            };
            // (block-uniform) first and last bin of the rows [t0, t0 + rows) and the kind of group they make — 1: one bin, 2: several
            // bins, 0: not a group
            auto group_kind = [&](uint32_t t0, uint32_t rows, uint32_t &ba, uint32_t &bz) -> int {
                if (rows == 0) return 0;
                const uint32_t bl = a.binlen;
                ba = t0 / bl;
                bz = (t0 + rows - 1) / bl;
                if (ba == bz) return 1;
                return (bl >= 32u && bz - ba + 1u <= MAXB) ? 2 : 0;
            };
            const bool fast = PG_EPI_STREAK && streak != 0;  // (block-uniform) a group of a streak: whole, one bin, the bin of the group before
            const bool known = PG_EPI_STREAK && !fast && gq_valid;  // the group before worked this one out (whole; nk_kind, nk_ba .. nk_bz) when it asked for its rows
            uint32_t grows_n = span, ba = nk_ba, bz = nk_bz;
            int kind = fast ? 1 : nk_kind;
            if (fast) {
                --streak;
            } else if (!known) {
                grows_n = rows_at(tile, ts);  // (>= 1: this tile has rows)
                kind = group_kind(ts, grows_n, ba, bz);
            }
            if (kind != 0) {
                uint64_t row0g = cur_row0;
                if (!fast) {
                    row0g = a.bin_off + ba;
                    const uint32_t nbg = bz - ba + 1u;  // bins of the group
                    const bool keep = cur_row0 != ~0ull && (kind == 1 ? row0g == cur_row0 : (row0g >= cur_row0 && row0g + nbg <= cur_row0 + MAXB));
                    if (!keep) {
                        if (cur_row0 != ~0ull) {
                            reduce_hist();
                            __syncthreads();
                            flush_hist(N, hist, bins, cur_row0, tid, MAXB);
                            __syncthreads();
                        }
                        cur_row0 = row0g;
                    }
                    if (PG_EPI_STREAK && kind == 1 && grows_n == span) {
                        // whole groups from ts on that end inside this bin, this contig and this workgroup's range (this one included)
                        const uint64_t bin_end = min((uint64_t)(ba + 1u) * a.binlen, (uint64_t)a.nkmers);
                        streak = min((uint32_t)(bin_end - ts) / span, (t_end - tile) / 8u) - 1u;
                    }
                }
                const uint32_t nv = 32u * tid < grows_n ? min(32u, grows_n - 32u * tid) : 0u;  // this thread's valid rows
                // (a 16-byte load that begins on a valid row ends inside the contig's 16-byte padded region)
                const uint4 *gg = reinterpret_cast<const uint4 *>(out1 + a.out_off + (uint64_t)ts + 32u * tid);
                const uint4 z4 = make_uint4(0, 0, 0, 0);
                const uint4 qa = gq_valid ? gq_next : (nv > 0u ? gg[0] : z4);
                const uint4 qb = gq_valid ? gq_next2 : (nv > 16u ? gg[1] : z4);
#if PG_EPI_WAIT_HERE
                // This group's rows are waited for HERE, before the next group's loads go out.  Left to the compiler the wait sat
                // at the rows' first use — BEHIND the prefetch — and, the paths above having merged, as vmcnt(0): it waited for
                // the prefetch as well, so that a group's load latency and its arithmetic ran one after the other (0.35 ms for
                // 8 x 10^8 rows where a kernel of the same geometry that only loads and counts takes 0.19).
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), nothing else
#endif
                // prefetch the next group when this one is whole and a whole group follows right behind
                const uint32_t tiles_here = (grows_n + PROBE_TILE - 1) / PROBE_TILE;
                gq_valid = streak != 0;
                if (!gq_valid && grows_n == span && rows_at(tile + 8, ts + span) == span) {
                    nk_kind = group_kind(ts + span, span, nk_ba, nk_bz);
                    gq_valid = nk_kind != 0;
                }
                if (gq_valid) {
                    gq_next = gg[span / 16u];
                    gq_next2 = gg[span / 16u + 1];
                }
                uint32_t w[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
                const uint32_t pos0 = ts + 32u * tid;  // at most one multiple of 100 among 32 positions
                const uint32_t r100 = (pos0 + 99u) / 100u;
                const uint32_t first = r100 * 100u - pos0;
                if (want100 && first < nv) {
                    uint32_t sel = w[0];
#pragma unroll
                    for (uint32_t i = 1; i < 8; ++i) sel = (first >> 2) == i ? w[i] : sel;
                    out100[a.out100_off + r100] = (uint8_t)(sel >> (8 * (first & 3)));
                }
#pragma unroll
                for (int kb = 0; kb < 3; ++kb) {
                    const uint32_t m0 = kb == 0 ? 0x55555555u : kb == 1 ? 0x33333333u : 0x0F0F0F0Fu;
                    const int sh = 1 << kb;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (i & sh) continue;
                        const uint32_t x = w[i], y = w[i | sh];
                        w[i] = (x & m0) | ((y << sh) & ~m0);       // the pair's bits with bit kb of g clear
                        w[i | sh] = ((x >> sh) & m0) | (y & ~m0);  // ... and set
                    }
                }
                // row j = 4 * word + byte of the thread sits at bit 8 * byte + word of every plane: the rows below j
                auto rows_below = [](uint32_t j) __attribute__((always_inline)) -> uint32_t {
                    uint32_t m = 0;
#pragma unroll
                    for (uint32_t rb = 0; rb < 4; ++rb) {
                        const uint32_t nw = j > rb ? min(8u, (j - rb + 3u) >> 2) : 0u;  // words whose byte rb is below row j
                        m |= ((1u << nw) - 1u) << (8u * rb);
                    }
                    return m;
                };
                const bool whole = grows_n == span;  // (block-uniform)
                const uint32_t mval = whole ? 0xFFFFFFFFu : rows_below(nv);
                if (!whole) {
#pragma unroll
                    for (int gb = 0; gb < 8; ++gb) w[gb] &= mval;  // rows past the group hold whatever follows in memory
                }
                if (want_cs) {
#pragma unroll
                    for (int gb = 0; gb < 8; ++gb) cacc[gb] += __popc(w[gb]);
                }
                auto cx = [&](int i, int j) __attribute__((always_inline)) {
                    const uint32_t lo = w[i] & w[j], hi = w[i] | w[j];
                    w[i] = lo;
                    w[j] = hi;
                };
                cx(0, 1), cx(2, 3), cx(4, 5), cx(6, 7);
                cx(0, 2), cx(1, 3), cx(4, 6), cx(5, 7);
                cx(1, 2), cx(5, 6), cx(0, 4), cx(3, 7);
                cx(1, 5), cx(2, 6);
                cx(1, 4), cx(3, 6);
                cx(2, 4), cx(3, 5);
                cx(3, 4);
                if (kind == 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) thr[i] += __popc(w[7 - i]);  // ascending order: w[7] = any bit set
                    grows += nv;
                } else {
                    // this thread's rows pos0 .. pos0 + nv - 1: bin of the first one (relative to the group's first bin) and
                    // rows until the next bin boundary
                    const uint32_t bl = a.binlen, bin0s = ba * bl, d0 = pos0 - bin0s;  // (ba = ts / bl)
                    const uint32_t rel0 = bl >= span ? (d0 >= bl ? 1u : 0u) : __umulhi(d0, 0xFFFFFFFFu / bl + 1u);  // (d0 < 2^16)
                    const uint32_t jb = min(nv, (rel0 + 1u) * bl - d0);  // valid rows of the first bin
                    const uint32_t mlo = rows_below(jb);
                    uint32_t *h0 = hist + ((uint32_t)(row0g - cur_row0) + rel0) * (N + 1);
                    auto add_classes = [&](uint32_t *h, uint32_t mask, uint32_t rows) __attribute__((always_inline)) {
                        uint32_t cl[9], above = rows;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const uint32_t t = (uint32_t)__popc(w[7 - i] & mask);  // rows of this bin with more than i bits
                            cl[i] = above - t;
                            above = t;
                        }
                        cl[8] = above;
#pragma unroll
                        for (int v = 8; v >= 1; --v)  // (junk bits beyond ngenomes count as class N)
                            if ((uint32_t)v > N) {
                                cl[v - 1] += cl[v];
                                cl[v] = 0;
                            }
#pragma unroll
                        for (int v = 0; v < 9; ++v)
                            if ((uint32_t)v <= N && cl[v]) atomicAdd(&h[v], cl[v]);
                    };
                    if (jb) add_classes(h0, mlo, jb);
                    if (jb < nv) add_classes(h0 + (N + 1), mval & ~mlo, nv - jb);
                }
                next_valid = false;
                tile += tiles_here - 1;
                continue;
            }
            gq_valid = false;
        }
        // ---- group path (2..8-byte rows): 4 full tiles of one contig inside one bin = 16 consecutive
        // rows per thread (4 x nbytes aligned words).  The per-tile bookkeeping (bin arithmetic, window
        // check, 1-in-100 search) is paid once per 16 rows instead of once per 4 — it was two thirds of
        // the instructions of this pass — and the histogram index needs no bin lookup ----
        if constexpr (MODE == 1) {
            const uint32_t ts = (tile - a.tile0) * PROBE_TILE;
            const uint32_t span = 4u * PROBE_TILE;
            // (the group may span several bins — contigs under 20 Mb have bins of nkmers / 100 rows — as long as a bin holds
            // at least 16 rows, so that a thread's 16 rows meet at most one boundary, and the group's bins fit the LDS window)
            const uint32_t nbg = (ts + span - 1) / a.binlen - ts / a.binlen + 1u;  // bins of the group
            const bool grp_ok = tile + 3 < t_end && tile_contig[tile + 3] == c && ts + span <= a.nkmers &&
                                (nbg == 1u || (a.binlen >= 16u && nbg <= MAXB));
            if (grp_ok) {
                const uint64_t row0g = a.bin_off + ts / a.binlen;
                const bool want_repl = nbg <= 2u && EPI_REPL * repl_stride <= MAXB * (N + 1u);  // (block-uniform; the copies must fit the window: the last one would run into `cs` otherwise)
                if (want_repl) {
                    if (!(repl && row0g >= cur_row0 && row0g + nbg <= cur_row0 + 2u)) {
                        if (cur_row0 != ~0ull) {
                            unreplicate();
                            __syncthreads();
                            flush_hist(N, hist, bins, cur_row0, tid, MAXB);
                            __syncthreads();
                        }
                        cur_row0 = row0g;
                        repl = true;
                    }
                } else {
                    unreplicate();
                    if (cur_row0 == ~0ull || row0g < cur_row0 || row0g + nbg > cur_row0 + MAXB) {
                        if (cur_row0 != ~0ull) {
                            __syncthreads();
                            flush_hist(N, hist, bins, cur_row0, tid, MAXB);
                            __syncthreads();
                        }
                        cur_row0 = row0g;
                    }
                }
                // this thread's 16 rows: bin of the first one (relative to the group's first bin) and rows until the boundary
                uint32_t rel0 = 0, jb = 16;
                if (nbg > 1u) {
                    const uint32_t bl = a.binlen, d0 = ts + 16u * tid - (ts / bl) * bl;
                    rel0 = bl >= span ? (d0 >= bl ? 1u : 0u) : __umulhi(d0, 0xFFFFFFFFu / bl + 1u);  // (d0 < 2^16)
                    jb = min(16u, (rel0 + 1u) * bl - d0);
                }
                uint32_t *hrow = hist + ((uint32_t)(row0g - cur_row0) + rel0) * (N + 1) + (want_repl ? ((uint32_t)lane % EPI_REPL) * repl_stride : 0u);
                const uint8_t *gt = out1 + a.out_off + ((uint64_t)ts + 16u * tid) * nbytes;
                auto rows16 = [&](auto nbc) {
                    constexpr int NB = decltype(nbc)::value;
                    uint32_t raw[4][8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) load_row_words<NB>(gt + q * 4 * NB, raw[q]);  // all 16 rows in flight
                    // (fewer in flight saves registers but measured slower: 2.6 / 2.74 / 2.78 ms at N=64 for 4 / 2 / 1
                    // groups ahead; requesting the NEXT group's rows as well costs a wave of occupancy: 2.2 vs 1.4 ms at N=27)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t w0[4], w1[4];
                        cut4_rows<NB>(raw[q], w0, w1);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            atomicAdd(&hrow[((uint32_t)(4 * q + j) >= jb ? N + 1 : 0u) +
                                            min((uint32_t)(__popc(w0[j]) + (NB > 4 ? __popc(w1[j]) : 0)), N)], 1u);
                        if (want_cs) {
                            vadd4(0, w0[0], w0[1], w0[2], w0[3]);
                            if (NB > 4) vadd4(1, w1[0], w1[1], w1[2], w1[3]);
                            vrows += PT;
                            if (vrows == VROWS_FLUSH) vflush();
                        }
                    }
                };
                switch (nbytes) {  // block-uniform
                    case 2: rows16(std::integral_constant<int, 2>{}); break;
                    case 3: rows16(std::integral_constant<int, 3>{}); break;
                    case 4: rows16(std::integral_constant<int, 4>{}); break;
                    case 5: rows16(std::integral_constant<int, 5>{}); break;
                    case 6: rows16(std::integral_constant<int, 6>{}); break;
                    case 7: rows16(std::integral_constant<int, 7>{}); break;
                    default: rows16(std::integral_constant<int, 8>{}); break;
                }
                // 1-in-100 rows: at most one multiple of 100 among 16 consecutive positions; its row is read
                // again (a cache hit) rather than selected out of 16 register pairs
                const uint32_t pos0 = ts + 16u * tid;
                const uint32_t r100 = (pos0 + 99u) / 100u;
                const uint32_t first = r100 * 100u - pos0;
                if (want100 && first < 16u) {
                    const uint8_t *pr = gt + first * nbytes;
                    uint8_t *o100 = out100 + a.out100_off + (uint64_t)r100 * nbytes;
                    for (uint32_t bb = 0; bb < nbytes; ++bb) o100[bb] = pr[bb];
                }
                wp_valid = false;
                tile += 3;
                continue;
            }
        }
        unreplicate();  // (the per-tile paths read the window in its ordinary form)
        const uint32_t tile_start = (tile - a.tile0) * PROBE_TILE;
        const uint32_t npos = min((uint32_t)PROBE_TILE, a.nkmers - tile_start);
        const uint32_t binlen = a.binlen, bin0 = tile_start / binlen, bin0_start = bin0 * binlen;
        const uint64_t row0 = a.bin_off + bin0;
        const bool onebin = (tile_start + npos) <= (bin0_start + binlen);  // block-uniform
        const bool big = binlen >= (uint32_t)PROBE_TILE;                    // a tile spans at most 2 bins
        const bool windowed = binlen >= MINBIN;                             // ... at most MAXB bins
        const uint32_t last_rel = (tile_start + npos - 1 - bin0_start) / binlen;
        // (bin - bin0) of a position for short bins: exact for pos - bin0_start < 2^16 > tile + bin
        const uint32_t binv = big ? 0u : 0xFFFFFFFFu / binlen + 1u;
        // block-uniform: may this tile add to the LDS window as it stands?  The per-thread one-byte
        // accumulators stand for the window's first bin, so a one-bin tile needs row0 == cur_row0.
        const bool reg_tile = MODE == 0 && big && onebin;
        const bool fits = cur_row0 != ~0ull && (reg_tile || !windowed ? row0 == cur_row0
                                                : (row0 >= cur_row0 && row0 + last_rel < cur_row0 + MAXB));
        if (!fits) {
            if (cur_row0 != ~0ull) {
                reduce_hist();
                __syncthreads();
                flush_hist(N, hist, bins, cur_row0, tid, MAXB);
                __syncthreads();
            }
            cur_row0 = row0;
        }
        const uint32_t rel_base = (uint32_t)(row0 - cur_row0);
        auto rel_of = [&](uint32_t pos) -> uint32_t {  // bin of a position of this tile, relative to bin0
            const uint32_t dpos = pos - bin0_start;
            return big ? (dpos >= binlen ? 1u : 0u) : __umulhi(dpos, binv);
        };
        const uint8_t *g = out1 + a.out_off + (uint64_t)tile_start * nbytes;
        if (MODE == 0 && windowed) {
            // ---- fast path (N <= 8): 4 one-byte rows per thread in one 32-bit word ----
            uint32_t packed = 0;
            if (next_valid) packed = next_packed;
            else if (p0 + 3 < npos) packed = *reinterpret_cast<const uint32_t *>(g + p0);
            else
                for (uint32_t j = 0; j < 4; ++j)
                    if (p0 + j < npos) packed |= (uint32_t)g[p0 + j] << (8 * j);
            // prefetch: the next tile of the same contig is a full tile right behind this one
            next_valid = (tile + 1 < t_end) && (npos == (uint32_t)PROBE_TILE) &&
                         (tile_start + 2u * PROBE_TILE <= a.nkmers) && (tile_contig[tile + 1] == c);
            if (next_valid) next_packed = *reinterpret_cast<const uint32_t *>(g + PROBE_TILE + p0);
            const uint32_t nact = p0 < npos ? min(4u, npos - p0) : 0u;
            if (want_cs) {  // bit g of the 4 rows = bits g, g+8, g+16, g+24 of the word
#pragma unroll
                for (int gb = 0; gb < 8; ++gb) cacc[gb] += __popc(packed & (0x01010101u << gb));
            }
            if (reg_tile) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t pcj = min((uint32_t)__popc((packed >> (8 * j)) & 0xFFu), N);
                    if ((uint32_t)j < nact) hacc += 1ull << (7 * pcj);
                }
                if (++since_spill == 31) spill();
            } else {  // the tile spans several bins: one LDS counter per (bin, popcount)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if ((uint32_t)j < nact)
                        atomicAdd(&hist[(rel_base + rel_of(tile_start + p0 + j)) * (N + 1) +
                                        min((uint32_t)__popc((packed >> (8 * j)) & 0xFFu), N)], 1u);
            }
            // 1-in-100 rows: at most one of 4 consecutive positions is a multiple of 100
            if (nact) {
                const uint32_t pos0 = tile_start + p0;
                const uint32_t r100 = (pos0 + 99u) / 100u;
                const uint32_t first = r100 * 100u;
                if (want100 && first < pos0 + nact) out100[a.out100_off + r100] = (uint8_t)(packed >> (8 * (first - pos0)));
            }
        } else if (WIDE && windowed) {
            // ---- wide path: the row bytes this launch sums, as one or two 32-bit words ----
            next_valid = false;
            uint32_t w0[PT], w1[PT];
            if (npos == (uint32_t)PROBE_TILE && (nbytes == 4 || nbytes == 8)) {  // (block-uniform) a full tile of 4- or 8-byte rows
                // the thread's 4 rows are 16 / 32 aligned bytes.  The next tile's are requested before this
                // tile is worked on when it is an equally regular one right behind: loads issued only when
                // their tile starts leave the memory latency exposed (2.2-3.4 TB/s of the 6.3 a plain
                // streaming read reaches with this geometry)
                const bool two = nbytes == 8;
                uint4 qa, qb = make_uint4(0, 0, 0, 0);
                const uint8_t *g4 = g + (uint64_t)p0 * nbytes;
                if (wp_valid) {
                    qa = wp_a;
                    qb = wp_b;
                } else {
                    qa = *reinterpret_cast<const uint4 *>(g4);
                    if (two) qb = *reinterpret_cast<const uint4 *>(g4 + 16);
                }
                wp_valid = (tile + 1 < t_end) && (tile_start + 2u * PROBE_TILE <= a.nkmers) && (tile_contig[tile + 1] == c);
                if (wp_valid) {
                    const uint8_t *gn = g4 + (uint64_t)PROBE_TILE * nbytes;
                    wp_a = *reinterpret_cast<const uint4 *>(gn);
                    if (two) wp_b = *reinterpret_cast<const uint4 *>(gn + 16);
                }
                if (two) {
                    w0[0] = qa.x; w1[0] = qa.y; w0[1] = qa.z; w1[1] = qa.w;
                    w0[2] = qb.x; w1[2] = qb.y; w0[3] = qb.z; w1[3] = qb.w;
                } else {
                    w0[0] = qa.x; w0[1] = qa.y; w0[2] = qa.z; w0[3] = qa.w;
                    w1[0] = w1[1] = w1[2] = w1[3] = 0;
                }
            } else if (npos == (uint32_t)PROBE_TILE) {  // other widths: nbytes aligned words, rows cut out with static shifts
                wp_valid = false;
                const uint8_t *g4 = g + (uint64_t)p0 * nbytes;
                uint32_t raw[8];
                switch (nbytes) {  // block-uniform
                    case 2: load_row_words<2>(g4, raw); cut4_rows<2>(raw, w0, w1); break;
                    case 3: load_row_words<3>(g4, raw); cut4_rows<3>(raw, w0, w1); break;
                    case 5: load_row_words<5>(g4, raw); cut4_rows<5>(raw, w0, w1); break;
                    case 6: load_row_words<6>(g4, raw); cut4_rows<6>(raw, w0, w1); break;
                    default: load_row_words<7>(g4, raw); cut4_rows<7>(raw, w0, w1); break;
                }
            } else {
                wp_valid = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint64_t r = 0;
                    if (p0 + j < npos)
                        for (uint32_t bb = 0; bb < nbytes; ++bb) r |= (uint64_t)g[(uint64_t)(p0 + j) * nbytes + bb] << (8 * bb);
                    w0[j] = (uint32_t)r;
                    w1[j] = (uint32_t)(r >> 32);
                }
            }
            const uint32_t nact = p0 < npos ? min(4u, npos - p0) : 0u;
            {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if ((uint32_t)j < nact) {
                        const uint32_t pc = __popc(w0[j]) + __popc(w1[j]);
                        atomicAdd(&hist[(rel_base + rel_of(tile_start + p0 + j)) * (N + 1) + min(pc, N)], 1u);
                    }
                }
            }
            if (nact) {  // 1-in-100 rows: at most one of 4 consecutive positions is a multiple of 100
                const uint32_t pos0 = tile_start + p0;
                const uint32_t r100 = (pos0 + 99u) / 100u;
                const uint32_t jsel = r100 * 100u - pos0;
                if (want100 && jsel < nact) {
                    const uint32_t a0 = jsel == 0 ? w0[0] : jsel == 1 ? w0[1] : jsel == 2 ? w0[2] : w0[3];
                    const uint32_t a1 = jsel == 0 ? w1[0] : jsel == 1 ? w1[1] : jsel == 2 ? w1[2] : w1[3];
                    uint8_t *o100 = out100 + a.out100_off + (uint64_t)r100 * nbytes;
                    if (nbytes == 4) *reinterpret_cast<uint32_t *>(o100) = a0;
                    else if (nbytes == 8) *reinterpret_cast<uint2 *>(o100) = make_uint2(a0, a1);
                    else {
                        const uint64_t r = (uint64_t)a0 | ((uint64_t)a1 << 32);
                        for (uint32_t bb = 0; bb < nbytes; ++bb) o100[bb] = (uint8_t)(r >> (8 * bb));
                    }
                }
            }
            if (want_cs) {  // rows beyond npos are zero: adding them is harmless
                vadd4(0, w0[0], w0[1], w0[2], w0[3]);
                if (ndbs > 1) vadd4(1, w1[0], w1[1], w1[2], w1[3]);
                vrows += PT;
                if (vrows == VROWS_FLUSH) vflush();
            }
        } else {
            next_valid = false;
            const uint32_t ndbs_all = (N + 31) / 32;
#pragma unroll
            for (int jj = 0; jj < PT; ++jj) {
                const uint32_t pl = p0 + jj;
                const bool active = pl < npos;
                const uint32_t pos = tile_start + pl;
                uint32_t popc = 0;
                const bool is100 = want100 && active && (pos % 100u == 0);
                for (uint32_t d = 0; d < ndbs_all; ++d) {
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
                hist_position(active, pos, popc, N, binlen, bin0, bin0_start, rel_base, hist, bins, a.bin_off, lane, MAXB);
            }
        }
    }
    if (want_cs && cur_c != ~0u) flush_colsums(cur_c);
    unreplicate();
    reduce_hist();
    __syncthreads();
    if (cur_row0 != ~0ull) flush_hist(N, hist, bins, cur_row0, tid, MAXB);
}

// ---------------------------------------------------------------------------
// Rows wider than 8 bytes (more than 64 genomes): the same statistics CHUNK-PARALLEL.  A lane owns one
// 16-byte chunk c of the rows it visits (C = ceil(nbytes / 16) consecutive lanes share a row, 64 / C
// rows per wave and step; one 16-byte load per lane and row, contiguous over the wave, at whatever byte
// alignment the row stride gives), so the per-row work (addressing, tail masking, histogram, the
// 1-in-100 test) is paid once per 16 bytes and every lane carries the vertical counters of FOUR words
// whatever the row width.  (Round 2's first version gave every lane one 32-bit word: 45 VALU
// instructions per word, 2.8 wave-instructions per 16-byte row, issue-bound at 2.7 TB/s.)
//   popcount of a row   4 v_bcnt per lane, C - 1 shuffles to the row's first lane, one LDS atomic
//   bitmap.100          each lane copies its chunk of the 1-in-100 rows
//   column sums         carry-save vertical counters per word: eight bit planes behind a Harley-Seal
//                       tree (4 rows at a time), LDS atomics every 240 rows, per contig to global
// EXACT: nbytes == 16 C (N a multiple of 128): aligned loads, no tail mask.
// ---------------------------------------------------------------------------
// (it runs 3-4 waves per SIMD on 104-149 VGPRs; round 4's first attempt to hold it to 5 or 6: 65-128 genomes 2.7-5.5 -> 5.0-13.9 ms,
// profiles/r4b_ab_epilogue_waves.txt)
template <int C_T, bool EXACT>  // chunks per row known at compile time (1..4), or 0: any
// (held to the registers of 5 waves per SIMD — 96 — it still spills in the row loop: 65-128 genomes 2.67-5.23 -> 3.04-5.55 ms,
// profiles/r4e_ab_stats_harley_seal.txt)
#ifndef PG_EPI_REPLC
#define PG_EPI_REPLC 1  // k_epilogue_chunks: the histogram of a tile inside one or two long bins in 8 (4) copies
#endif
#ifndef PG_EPI_WAVESC
#define PG_EPI_WAVESC 1
#endif
__global__ __launch_bounds__(EPI_THREADS, PG_EPI_WAVESC) void k_epilogue_chunks(uint32_t N, const AnchorDesc *__restrict__ ad,
                                                                 const uint32_t *__restrict__ tile_contig, uint32_t ntiles,
                                                                 const uint8_t *__restrict__ out1, uint8_t *__restrict__ out100,
                                                                 uint32_t *__restrict__ bins,
                                                                 unsigned long long *__restrict__ colsums, uint32_t flags,
                                                                 const uint2 *__restrict__ ranges, uint32_t wpr) {
    extern __shared__ uint4 smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nbytes = (N + 7) / 8, C = C_T ? (uint32_t)C_T : (nbytes + 15) / 16;
    const uint32_t RPW = 64u / C, RPS = RPW * (EPI_THREADS / 64);  // rows per wave / per workgroup and step
    const uint32_t c = (uint32_t)lane % C, rsub = (uint32_t)wave * RPW + (uint32_t)lane / C;
    const bool lane_on = (uint32_t)lane < RPW * C;
    const uint32_t vb = EXACT ? 16u : min(16u, nbytes - 16u * c);  // bytes of this lane's chunk (the row's last one may be short)
    // column counters in LDS: one copy per 16 rows of a wave's step, so that at most 16 lanes add to a word
    // at a time (and the address is lane-dependent: on a wave-uniform address the compiler's atomic
    // optimizer would sum the lanes' values one by one, 64 rounds per counter)
    const uint32_t K = (RPW + 15u) / 16u, cs_words = 128u * C;
    uint32_t wm[4];                                  // ... as masks of its four words
#pragma unroll
    for (uint32_t w = 0; w < 4; ++w) {
        const uint32_t nb = vb > 4u * w ? min(4u, vb - 4u * w) : 0u;
        wm[w] = nb == 4 ? 0xFFFFFFFFu : (1u << (8 * nb)) - 1u;
    }
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    const uint32_t MAXB = max(EPI_MAXB, (flags >> 8) & 0xFFu), MINBIN = epi_minbin(MAXB);  // bins in the LDS window (launcher's choice)
    uint32_t *cs = hist + ((MAXB * (N + 1) + 3) & ~3u);
    for (uint32_t i = tid; i < MAXB * (N + 1); i += EPI_THREADS) hist[i] = 0;
    for (uint32_t i = tid; i < K * cs_words; i += EPI_THREADS) cs[i] = 0;
    uint32_t *cs_mine = cs + ((uint32_t)lane / C / 16u) * cs_words + 128u * c;
    __syncthreads();
    const bool want_cs = (flags & 1u) != 0;
    const bool want100 = (flags & 2u) == 0;
    const EpiRange er = epi_range(4u, ntiles, ranges, wpr);
    const uint32_t t_begin = er.begin, t_end = er.end;
    uint64_t cur_row0 = ~0ull;
    uint32_t cur_c = ~0u;
    AnchorDesc a;
    a.out_off = a.out100_off = a.bin_off = 0;
    a.nkmers = a.binlen = a.tile0 = a.nbins = 0;
    // Column sums: per lane and word EIGHT bit planes of vertical counters (bit g of plane p = bit p of the number of rows seen
    // with genome g set: up to 255 rows between flushes) fed through a Harley-Seal carry-save tree — four rows enter the
    // ones / twos planes per iteration (9 instructions), their carry of weight 4 is held back every other iteration and
    // enters the fours plane together with the next one (3), likewise the eights (3 per 8 rows), and only every 16 rows a
    // carry ripples through the four upper planes (8): 3.3 instructions per row and word.  (Rounds 2-4 kept four planes,
    // emptied every 12 rows into byte-sliced accumulators — 64 instructions per word — and those every 252 rows into LDS:
    // 8.3 per row and word, a third of this pass's instructions, and 48 registers where this takes 40.)
    uint32_t vp[4][8], pf[4], pe[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
#pragma unroll
        for (int q = 0; q < 8; ++q) vp[w][q] = 0;
        pf[w] = pe[w] = 0;
    }
    uint32_t vrows = 0;  // rows in the planes (block-uniform, a multiple of 4): the carries held back follow from it
    // add a word of carries of weight 2^q0 to the planes q0 .. 7 (no carry leaves plane 7: fewer than 256 rows)
    auto ripple = [&](uint32_t (&p)[8], uint32_t cw, int q0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q >= q0) {
                const uint32_t n = p[q] & cw;
                p[q] ^= cw;
                cw = n;
            }
    };
    auto vadd4 = [&](int w, uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, bool odd4, bool odd8) __attribute__((always_inline)) {
        uint32_t (&p)[8] = vp[w];
        const uint32_t x = p[0];
        const uint32_t t1 = x ^ r0, s1 = t1 ^ r1, ca = (t1 & r1) | (~t1 & x);      // x + r0 + r1
        const uint32_t t2 = s1 ^ r2, s2 = t2 ^ r3, cb = (t2 & r3) | (~t2 & s1);    // .. + r2 + r3
        p[0] = s2;
        const uint32_t y = p[1];
        const uint32_t t3 = y ^ ca, cc = (t3 & cb) | (~t3 & y);                     // twos + ca + cb -> a carry of weight 4
        p[1] = t3 ^ cb;
        if (!odd4) {  // (block-uniform) held back: the next four rows' carry joins it
            pf[w] = cc;
            return;
        }
        const uint32_t z = p[2], t4 = z ^ pf[w], c8 = (t4 & cc) | (~t4 & z);       // fours + both carries -> weight 8
        p[2] = t4 ^ cc;
        if (!odd8) {
            pe[w] = c8;
            return;
        }
        const uint32_t u = p[3], t5 = u ^ pe[w], c16 = (t5 & c8) | (~t5 & u);      // eights + both carries -> weight 16
        p[3] = t5 ^ c8;
        ripple(p, c16, 4);
    };
    auto vflush = [&]() __attribute__((always_inline)) {  // planes -> the workgroup's LDS counters of this lane's words
        const bool odd4 = (vrows & 4u) != 0, odd8 = (vrows & 8u) != 0;  // carries still held back (a flush between whole 16-row blocks)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (odd4) ripple(vp[w], pf[w], 2);
            if (odd8) ripple(vp[w], pe[w], 3);
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // bits j, 8 + j, 16 + j, 24 + j of the word: their four counts as the bytes of v
                uint32_t v = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) v |= ((vp[w][q] >> j) & 0x01010101u) << q;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t cnt = (v >> (8 * b)) & 255u;
                    if (cnt) atomicAdd(&cs_mine[32 * w + 8 * b + j], cnt);
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) vp[w][q] = 0;
        }
        vrows = 0;
    };
    auto flush_colsums = [&](uint32_t contig) __attribute__((always_inline)) {
        if (vrows) vflush();
        __syncthreads();
        for (uint32_t i = tid; i < N; i += EPI_THREADS) {
            uint32_t v = 0;
            for (uint32_t kk = 0; kk < K; ++kk) {
                v += cs[kk * cs_words + i];
                cs[kk * cs_words + i] = 0;
            }
            if (v) atomicAdd(&colsums[(uint64_t)contig * N + i], (unsigned long long)v);
        }
        __syncthreads();
    };
    struct __attribute__((packed)) U32 { uint32_t v; };
    struct __attribute__((packed)) U128 { uint32_t x, y, z, w; };
    // popcount of a whole row from its lanes' chunks, valid (at least) in the row's first lane
    auto row_popc = [&](uint32_t pc) __attribute__((always_inline)) -> uint32_t {
        if (C_T == 1) return pc;
        if (C_T == 2) return pc + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pc, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
        if (C_T == 4) {
            pc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pc, 0xB1, 0xF, 0xF, false);
            return pc + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)pc, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
        }
        uint32_t tot = pc;
        for (uint32_t q = 1; q < C; ++q) tot += (uint32_t)__shfl_down((int)pc, q);
        return tot;
    };
    // The rows are streamed in iterations of 4 steps (4 x RPS rows), the loads of iteration i + 1 issued
    // before iteration i is worked on — also across tiles: 8 x 16 bytes per lane in flight.
    constexpr uint32_t NJ = 4;
    const uint32_t iter_rows = NJ * RPS, IPT = ((uint32_t)PROBE_TILE + iter_rows - 1) / iter_rows;
    const uint32_t nit = t_end > t_begin ? (t_end - t_begin) * IPT : 0u;
    auto issue = [&](uint32_t it, uint4 (&out)[NJ]) {
        const uint32_t tile = t_begin + it / IPT, r0 = (it % IPT) * iter_rows;
        const AnchorDesc A = ad[tile_contig[tile]];  // (uniform: scalar loads)
        const uint32_t ts = (tile - A.tile0) * PROBE_TILE;
        const uint32_t npos = min((uint32_t)PROBE_TILE, A.nkmers - ts);
        const uint8_t *g = out1 + A.out_off + (uint64_t)ts * nbytes + 16u * c;
        // branch-free: every lane always loads 16 bytes from a valid row (a predicated load per row makes
        // the compiler wait for each load in turn); a short last chunk reads into the next row — or, at the
        // very end, into the 16 bytes of slack every row buffer carries
#pragma unroll
        for (uint32_t j = 0; j < NJ; ++j) {
            const uint32_t pl = r0 + j * RPS + rsub;
            const uint8_t *q = g + min(pl, npos - 1u) * nbytes;
            if (EXACT) out[j] = *reinterpret_cast<const uint4 *>(q);
            else {
                const U128 t = *reinterpret_cast<const U128 *>(q);
                out[j] = make_uint4(t.x, t.y, t.z, t.w);
            }
        }
    };
    uint4 v[NJ], vn[NJ];
    if (nit) issue(0, v);
    // per-tile state (block-uniform), set when an iteration starts a tile
    uint32_t tile_start = 0, npos = 0, binlen = 1, bin0 = 0, bin0_start = 0, binv = 0, rel_base = 0;
    bool big = false, windowed = false;
    // One LDS counter per row and (bin, class): the lanes of a wave mostly ask for the same few — on a real pangenome most rows
    // carry ALL genomes — and the LDS works lanes that add to one word off one after the other (SQ_LDS_BANK_CONFLICT was 94 %
    // of this pass's LDS cycles, a wave waited for the LDS 28 % of its time; d = 0.0001: +15 % on the whole pass).  While the
    // tiles lie inside one or two long bins (any contig of more than 100 tiles) the window is therefore used as repl_n COPIES of
    // those two bin rows, row-lane l / C adding to copy (l / C) % repl_n (copy c at c * repl_stride, an odd stride: the copies
    // of one class sit in different banks) — k_epilogue's scheme for rows of 2..8 bytes; `unreplicate` folds the copies into
    // the window's ordinary form, rows 0 and 1, before anything else reads or flushes it.  `repl` is block-uniform.
    // Measured (profiles/r4e_ab_stats_hist_copies_wide.txt): core-heavy rows (d = 0.0001) 65 genomes 2.95 -> 2.73 ms, 128 genomes
    // 6.44 -> 5.29 (89.9 -> 98.8 G k-mers/s for the step); the bench's d = 0.01 +1 %.  Rows of two or three chunks (more than
    // 128 genomes: only every second or third lane adds) LOSE 3-6 % with it at either divergence and keep the plain window.
    const uint32_t repl_stride = (2u * (N + 1u)) | 1u;
    const uint32_t repl_n = 8u * repl_stride <= MAXB * (N + 1u) ? 8u : 4u;  // (MAXB >= 16: four copies always fit)
    const uint32_t repl_off = (((uint32_t)lane / C) % repl_n) * repl_stride;
    bool repl = false;
    auto unreplicate = [&]() __attribute__((always_inline)) {
        if (!repl) return;
        __syncthreads();
        for (uint32_t i = tid; i < 2u * (N + 1u); i += EPI_THREADS) {
            uint32_t hv = hist[i];
            for (uint32_t cpy = 1; cpy < repl_n; ++cpy) {
                hv += hist[cpy * repl_stride + i];
                hist[cpy * repl_stride + i] = 0;
            }
            hist[i] = hv;
        }
        __syncthreads();
        repl = false;
    };
    for (uint32_t it = 0; it <= nit; ++it) {  // (one more round: the last contig's column sums, flushed at ONE site)
        const bool fin = it == nit;
        if (it + 1 < nit) issue(it + 1, vn);
        const uint32_t tile = t_begin + it / IPT, r0 = fin ? 0u : (it % IPT) * iter_rows;
        if (r0 == 0) {
            const uint32_t cg = fin ? ~0u : tile_contig[tile];
            if (cg != cur_c) {
                if (want_cs && cur_c != ~0u) flush_colsums(cur_c);
                if (!fin) a = ad[cg];
                cur_c = cg;
            }
            if (fin) break;
            tile_start = (tile - a.tile0) * PROBE_TILE;
            npos = min((uint32_t)PROBE_TILE, a.nkmers - tile_start);
            binlen = a.binlen;
            bin0 = tile_start / binlen;
            bin0_start = bin0 * binlen;
            const uint64_t row0 = a.bin_off + bin0;
            big = binlen >= (uint32_t)PROBE_TILE;
            windowed = binlen >= MINBIN;
            const uint32_t last_rel = (tile_start + npos - 1 - bin0_start) / binlen;
            binv = big ? 0u : 0xFFFFFFFFu / binlen + 1u;
            if (PG_EPI_REPLC && big && C == 1u) {  // (block-uniform) the tile's one or two bins as copies
                if (!(repl && row0 >= cur_row0 && row0 + last_rel < cur_row0 + 2u)) {
                    if (cur_row0 != ~0ull) {
                        const uint32_t held = repl ? 2u : MAXB;  // (as copies the window held two rows: nothing beyond them to look at)
                        unreplicate();
                        __syncthreads();
                        flush_hist(N, hist, bins, cur_row0, tid, held);
                        __syncthreads();
                    }
                    cur_row0 = row0;
                    repl = true;
                }
            } else {
                unreplicate();
                const bool fits = cur_row0 != ~0ull && (!windowed ? row0 == cur_row0 : (row0 >= cur_row0 && row0 + last_rel < cur_row0 + MAXB));
                if (!fits) {
                    if (cur_row0 != ~0ull) {
                        __syncthreads();
                        flush_hist(N, hist, bins, cur_row0, tid, MAXB);
                        __syncthreads();
                    }
                    cur_row0 = row0;
                }
            }
            rel_base = (uint32_t)(row0 - cur_row0);
        }
        if (r0 < npos) {  // (block-uniform)
            uint32_t *hrow = hist + rel_base * (N + 1) + (repl ? repl_off : 0u);
            const bool full = EXACT && C_T && (64 % (C_T ? C_T : 1) == 0) && r0 + iter_rows <= npos;  // (block-uniform) no row to mask
#pragma unroll
            for (uint32_t j = 0; j < NJ; ++j) {
                const uint32_t pl = r0 + j * RPS + rsub;
                const bool on = full || (lane_on && pl < npos);
                if (!full) {
                    const uint32_t keep = on ? 0xFFFFFFFFu : 0u;
                    v[j].x &= EXACT ? keep : keep & wm[0];
                    v[j].y &= EXACT ? keep : keep & wm[1];
                    v[j].z &= EXACT ? keep : keep & wm[2];
                    v[j].w &= EXACT ? keep : keep & wm[3];
                }
                const uint32_t tot = row_popc(__popc(v[j].x) + __popc(v[j].y) + __popc(v[j].z) + __popc(v[j].w));
                const uint32_t pos = tile_start + pl;
                if (windowed) {
                    if (on && c == 0) {
                        const uint32_t dpos = pos - bin0_start;
                        const uint32_t rel = big ? (dpos >= binlen ? 1u : 0u) : __umulhi(dpos, binv);
                        atomicAdd(&hrow[rel * (N + 1) + min(tot, N)], 1u);
                    }
                } else {
                    hist_position(on && c == 0, pos, tot, N, binlen, bin0, bin0_start, rel_base, hist, bins, a.bin_off, lane, MAXB);
                }
                if (want100 && on && pos % 100u == 0) {  // 1-in-100 rows: every lane copies its chunk
                    uint8_t *o = out100 + a.out100_off + (uint64_t)(pos / 100u) * nbytes + 16u * c;
                    if (EXACT) *reinterpret_cast<uint4 *>(o) = v[j];
                    else {
                        const uint32_t xw[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                        for (uint32_t w = 0; w < 4; ++w) {
                            if (vb >= 4u * w + 4u) reinterpret_cast<U32 *>(o + 4u * w)->v = xw[w];
                            else if (vb > 4u * w)
                                for (uint32_t bb = 0; bb < vb - 4u * w; ++bb) o[4u * w + bb] = (uint8_t)(xw[w] >> (8 * bb));
                        }
                    }
                }
            }
            if (want_cs) {  // rows beyond npos are zero: adding them is harmless
                const bool odd4 = (vrows & 4u) != 0, odd8 = (vrows & 8u) != 0;
                vadd4(0, v[0].x, v[1].x, v[2].x, v[3].x, odd4, odd8);
                vadd4(1, v[0].y, v[1].y, v[2].y, v[3].y, odd4, odd8);
                vadd4(2, v[0].z, v[1].z, v[2].z, v[3].z, odd4, odd8);
                vadd4(3, v[0].w, v[1].w, v[2].w, v[3].w, odd4, odd8);
                vrows += 4;
                if (vrows == 240) vflush();  // (the 8 planes count to 255)
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < NJ; ++j) v[j] = vn[j];
    }
    unreplicate();
    __syncthreads();
    if (cur_row0 != ~0ull) flush_hist(N, hist, bins, cur_row0, tid, MAXB);
}

// ---------------------------------------------------------------------------
// Rows of 9..16 bytes (65..128 genomes): k_epilogue's scheme for 2..8-byte rows at THREE or FOUR words per row.
// k_epilogue_chunks gives every row one lane and one (unaligned) 16-byte load whatever its width — ≈ 38 instructions per
// row and lane, 4.1 ps per row at 9 bytes as at 16: the 65th genome paid for 128 (0.98 -> 2.68 ms for one more row byte).
// Here a thread owns RPT consecutive rows of a group of full tiles (NBT aligned words per four rows, cut into rows with
// static funnel shifts; RPT = 8, a group = two tiles: 16 rows per thread as for the narrower widths cost a wave of occupancy
// and 2-7 %, profiles/r5g_ab_stats_w_rows.txt), the group's bookkeeping (bins, window, the 1-in-100 row) is paid once per group,
// the histogram takes one LDS atomic per row (in 8 copies while the group lies inside one or two long bins), and the
// column sums go through eight counter planes per word behind the Harley-Seal tree (3.3 instructions per row and word).
// Tiles that form no group — a contig's last ones, contigs of a few tiles, bins shorter than 16 rows — take the same
// four rows per thread one tile at a time.  Reference: the per-bin histogram and rows of cpp/anchor.cpp:150-189,
// index.py:1169-1183; column sums: index.py:1051,1068-1074.
// ---------------------------------------------------------------------------
#ifndef PG_EPI_W
#define PG_EPI_W 1  // 0: rows of 9..16 bytes through k_epilogue_chunks (rounds 2-4)
#endif
#ifndef PG_EPI_W_GQ12
#define PG_EPI_W_GQ12 2  // k_epilogue_w, rows of 9..12 bytes: tiles per group (4: 16 consecutive rows per thread, 2: 8)
#endif
#ifndef PG_EPI_W_GQ16
#define PG_EPI_W_GQ16 2  // ... rows of 13..16 bytes
#endif
#ifndef PG_EPI_WAVESW12
#define PG_EPI_WAVESW12 1  // waves per SIMD the instantiations of 9..12-byte rows are held to (1: the compiler's choice)
#endif
#ifndef PG_EPI_WAVESW16
#define PG_EPI_WAVESW16 4  // ... of 13..16-byte rows
#endif
template <int NB>
__device__ __forceinline__ void cut4_rows_w(const uint32_t (&raw)[NB], uint32_t (&w)[(NB + 3) / 4][4]) {
    constexpr int NW = (NB + 3) / 4;
    constexpr uint32_t last_keep = (NB % 4) ? ((1u << (8 * (NB % 4))) - 1u) : 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int off = j * NB, idx = off >> 2, sh = 8 * (off & 3);  // (constants once unrolled)
#pragma unroll
        for (int t = 0; t < NW; ++t) {
            const uint32_t lo = raw[idx + t];  // (idx + NW - 1 <= NB - 1: the row ends inside the four rows' words)
            const uint32_t hi = (idx + t + 1 < NB) ? raw[idx + t + 1] : 0u;
            uint32_t v = sh ? __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh) : lo;
            if (t == NW - 1) v &= last_keep;
            w[t][j] = v;
        }
    }
}
// one row of NBT bytes copied to the low-resolution bitmap: one unaligned 16-byte read (it may reach into the rows that follow, or
// into the 16 bytes of slack every row buffer carries), whole words and the tail's bytes written
template <int NBT>
__device__ __forceinline__ void copy_row_w(const uint8_t *pr, uint8_t *o) {
    struct __attribute__((packed)) U32 { uint32_t v; };
    struct __attribute__((packed)) U128 { uint32_t x, y, z, w; };
    const U128 t = *reinterpret_cast<const U128 *>(pr);
    const uint32_t xw[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int w = 0; w < NBT / 4; ++w) reinterpret_cast<U32 *>(o + 4 * w)->v = xw[w];
#pragma unroll
    for (int bb = 0; bb < NBT % 4; ++bb) o[4 * (NBT / 4) + bb] = (uint8_t)(xw[(NBT / 4) & 3] >> (8 * bb));
}
template <int NBT>
__global__ __launch_bounds__(EPI_THREADS, (NBT <= 12 ? PG_EPI_WAVESW12 : PG_EPI_WAVESW16)) void k_epilogue_w(uint32_t N, const AnchorDesc *__restrict__ ad,
                                                                          const uint32_t *__restrict__ tile_contig, uint32_t ntiles,
                                                                          const uint8_t *__restrict__ out1, uint8_t *__restrict__ out100,
                                                                          uint32_t *__restrict__ bins,
                                                                          unsigned long long *__restrict__ colsums, uint32_t flags,
                                                                          const uint2 *__restrict__ ranges, uint32_t wpr) {
    static_assert(NBT >= 9 && NBT <= 16, "rows of 9..16 bytes");
    extern __shared__ uint4 smem[];
    constexpr int PT = 4;             // rows per thread and tile
    constexpr int NW = (NBT + 3) / 4;  // words per row
    constexpr int GQ = (NBT <= 12 ? PG_EPI_W_GQ12 : PG_EPI_W_GQ16);  // tiles per group = blocks of four rows per thread and group
    constexpr uint32_t RPT = 4u * GQ;   // consecutive rows of a group per thread
    constexpr uint32_t nbytes = NBT;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t Nw = N;
    uint32_t *hist = reinterpret_cast<uint32_t *>(smem);
    const uint32_t MAXB = max(EPI_MAXB, (flags >> 8) & 0xFFu), MINBIN = epi_minbin(MAXB);
    uint32_t *cs = hist + ((MAXB * (N + 1) + 3) & ~3u);
    for (uint32_t i = tid; i < MAXB * (N + 1); i += EPI_THREADS) hist[i] = 0;
    for (uint32_t i = tid; i < N; i += EPI_THREADS) cs[i] = 0;
    __syncthreads();
    const bool want_cs = (flags & 1u) != 0;
    const bool want100 = (flags & 2u) == 0;
    const EpiRange er = epi_range(4u, ntiles, ranges, wpr);
    const uint32_t t_begin = er.begin, t_end = er.end;
    uint64_t cur_row0 = ~0ull;
    uint32_t cur_c = ~0u;
    AnchorDesc a;
    a.out_off = a.out100_off = a.bin_off = 0;
    a.nkmers = a.binlen = a.tile0 = a.nbins = 0;
    const uint32_t p0 = tid * PT;
    // ---- column sums: eight counter planes per word behind a Harley-Seal tree (as k_epilogue, rows of 2..8 bytes) ----
    uint32_t vp[NW][8], pf[NW], pe[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) {
#pragma unroll
        for (int q = 0; q < 8; ++q) vp[w][q] = 0;
        pf[w] = pe[w] = 0;
    }
    uint32_t vrows = 0;  // rows in the planes (block-uniform, a multiple of 4)
    auto ripple = [&](uint32_t (&p)[8], uint32_t cw, int q0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q >= q0) {
                const uint32_t n = p[q] & cw;
                p[q] ^= cw;
                cw = n;
            }
    };
    auto vadd4 = [&](uint32_t (&p)[8], uint32_t &pfw, uint32_t &pew, const uint32_t (&r)[4], bool odd4, bool odd8) __attribute__((always_inline)) {
        const uint32_t x = p[0];
        const uint32_t t1 = x ^ r[0], s1 = t1 ^ r[1], ca = (t1 & r[1]) | (~t1 & x);      // x + r0 + r1
        const uint32_t t2 = s1 ^ r[2], s2 = t2 ^ r[3], cb = (t2 & r[3]) | (~t2 & s1);    // .. + r2 + r3
        p[0] = s2;
        const uint32_t y = p[1];
        const uint32_t t3 = y ^ ca, cc = (t3 & cb) | (~t3 & y);                           // twos + ca + cb -> a carry of weight 4
        p[1] = t3 ^ cb;
        if (!odd4) {
            pfw = cc;
            return;
        }
        const uint32_t z = p[2], t4 = z ^ pfw, c8 = (t4 & cc) | (~t4 & z);               // fours + both carries -> weight 8
        p[2] = t4 ^ cc;
        if (!odd8) {
            pew = c8;
            return;
        }
        const uint32_t u = p[3], t5 = u ^ pew, c16 = (t5 & c8) | (~t5 & u);              // eights + both carries -> weight 16
        p[3] = t5 ^ c8;
        ripple(p, c16, 4);
    };
    auto vadd_rows = [&](const uint32_t (&w)[NW][4]) __attribute__((always_inline)) {  // four rows into the planes of every word
        const bool odd4 = (vrows & 4u) != 0, odd8 = (vrows & 8u) != 0;
#pragma unroll
        for (int t = 0; t < NW; ++t) vadd4(vp[t], pf[t], pe[t], w[t], odd4, odd8);
        vrows += PT;
    };
    auto vflush = [&]() __attribute__((always_inline)) {  // planes -> the workgroup's LDS counters: ONE (wave-uniform) call site, at the top of the tile loop
        const bool odd4 = (vrows & 4u) != 0, odd8 = (vrows & 8u) != 0;
#pragma unroll
        for (int ws = 0; ws < NW; ++ws) {
            if (odd4) ripple(vp[ws], pf[ws], 2);
            if (odd8) ripple(vp[ws], pe[ws], 3);
            uint32_t R[16];
#pragma unroll
            for (int q = 0; q < 8; ++q) {  // byte b of v counts genome 32 ws + 8 b + q (up to 255 rows)
                uint32_t v = 0;
#pragma unroll
                for (int pl = 0; pl < 8; ++pl) v |= ((vp[ws][pl] >> q) & 0x01010101u) << pl;
                R[2 * q] = v & 0x00FF00FFu;
                R[2 * q + 1] = (v >> 8) & 0x00FF00FFu;
            }
#pragma unroll
            for (int pl = 0; pl < 8; ++pl) vp[ws][pl] = 0;
#pragma unroll
            for (int half = 8, bit = 32; half >= 1; half >>= 1, bit >>= 1) {
                const bool up = (lane & bit) != 0;
#pragma unroll
                for (int i = 0; i < half; ++i) {
                    const uint32_t send = up ? R[i] : R[i + half];
                    const uint32_t keep = up ? R[i + half] : R[i];
                    R[i] = keep + (uint32_t)__shfl_xor((int)send, bit);
                }
            }
            R[0] += (uint32_t)__shfl_xor((int)R[0], 2);
            R[0] += (uint32_t)__shfl_xor((int)R[0], 1);
            if ((lane & 3) == 0) {  // this lane holds register (lane >> 2): q = idx / 2, odd idx = bytes 1 and 3
                const uint32_t idx = (uint32_t)lane >> 2;
                const uint32_t g0 = 32 * ws + (idx >> 1) + ((idx & 1) ? 8u : 0u);
                if (g0 < Nw && (R[0] & 0xFFFFu)) atomicAdd(&cs[g0], R[0] & 0xFFFFu);
                if (g0 + 16 < Nw && (R[0] >> 16)) atomicAdd(&cs[g0 + 16], R[0] >> 16);
            }
        }
        vrows = 0;
    };
    auto flush_colsums = [&](uint32_t contig) __attribute__((always_inline)) {  // (one call site)
        __syncthreads();
        for (uint32_t i = tid; i < N; i += EPI_THREADS) {
            const uint32_t v = cs[i];
            if (v) {
                atomicAdd(&colsums[(uint64_t)contig * N + i], (unsigned long long)v);
                cs[i] = 0;
            }
        }
        __syncthreads();
    };
    // ---- the histogram window as EPI_REPL copies of two bin rows while the groups lie inside one or two long bins ----
    constexpr uint32_t EPI_REPL = 8;
    const uint32_t repl_stride = (2u * (N + 1u)) | 1u;  // (8 copies need 16 (N + 1) + 8 words: they fit from MAXB = 17 on — epi_maxb_for gives >= 23 for N <= 128; want_repl checks)
    bool repl = false;
    auto unreplicate = [&]() __attribute__((always_inline)) {
        if (!repl) return;
        __syncthreads();
        for (uint32_t i = tid; i < 2u * (N + 1u); i += EPI_THREADS) {
            uint32_t v = hist[i];
#pragma unroll
            for (uint32_t cpy = 1; cpy < EPI_REPL; ++cpy) {
                v += hist[cpy * repl_stride + i];
                hist[cpy * repl_stride + i] = 0;
            }
            hist[i] = v;
        }
        __syncthreads();
        repl = false;
    };
    auto popc_row = [&](const uint32_t (&w)[NW][4], int j) __attribute__((always_inline)) -> uint32_t {
        uint32_t pc = (uint32_t)__popc(w[0][j]);
#pragma unroll
        for (int t = 1; t < NW; ++t) pc += (uint32_t)__popc(w[t][j]);
        return min(pc, N);  // (junk bits beyond ngenomes count as class N)
    };
    for (uint32_t tile = t_begin; tile <= t_end; ++tile) {  // (one more round: the last contig's column sums leave at the one site)
        const bool fin = tile >= t_end;
        const uint32_t c = fin ? ~0u : tile_contig[tile];
        // the planes count to 255 rows and a group brings 16: emptied here when they could not take another group, and when
        // the range moves on to another contig (column sums are kept per contig)
        if (want_cs && vrows && (vrows + RPT > 255u || c != cur_c)) vflush();
        if (c != cur_c) {  // block-uniform
            if (want_cs && cur_c != ~0u) flush_colsums(cur_c);
            if (!fin) a = ad[c];
            cur_c = c;
        }
        if (fin) break;
        // ---- group path: GQ full tiles of one contig = 4 GQ consecutive rows per thread (GQ = 2: 8 rows) ----
        {
            const uint32_t ts = (tile - a.tile0) * PROBE_TILE;
            const uint32_t span = (uint32_t)GQ * PROBE_TILE;
            const uint32_t nbg = (ts + span - 1) / a.binlen - ts / a.binlen + 1u;  // bins of the group
            const bool grp_ok = tile + (GQ - 1) < t_end && tile_contig[tile + (GQ - 1)] == c && ts + span <= a.nkmers &&
                                (nbg == 1u || (a.binlen >= RPT && nbg <= MAXB));
            if (grp_ok) {
                const uint64_t row0g = a.bin_off + ts / a.binlen;
                const bool want_repl = nbg <= 2u && EPI_REPL * repl_stride <= MAXB * (N + 1u);  // (block-uniform; the copies must fit the window: the last one would run into `cs` otherwise)
                if (want_repl) {
                    if (!(repl && row0g >= cur_row0 && row0g + nbg <= cur_row0 + 2u)) {
                        if (cur_row0 != ~0ull) {
                            unreplicate();
                            __syncthreads();
                            flush_hist(N, hist, bins, cur_row0, tid, MAXB);
                            __syncthreads();
                        }
                        cur_row0 = row0g;
                        repl = true;
                    }
                } else {
                    unreplicate();
                    if (cur_row0 == ~0ull || row0g < cur_row0 || row0g + nbg > cur_row0 + MAXB) {
                        if (cur_row0 != ~0ull) {
                            __syncthreads();
                            flush_hist(N, hist, bins, cur_row0, tid, MAXB);
                            __syncthreads();
                        }
                        cur_row0 = row0g;
                    }
                }
                uint32_t rel0 = 0, jb = RPT;  // bin of the thread's first row (relative to the group's first bin), rows until the boundary
                if (nbg > 1u) {
                    const uint32_t bl = a.binlen, d0 = ts + RPT * tid - (ts / bl) * bl;
                    rel0 = bl >= span ? (d0 >= bl ? 1u : 0u) : __umulhi(d0, 0xFFFFFFFFu / bl + 1u);  // (d0 < 2^16)
                    jb = min(RPT, (rel0 + 1u) * bl - d0);
                }
                uint32_t *hrow = hist + ((uint32_t)(row0g - cur_row0) + rel0) * (N + 1) + (want_repl ? ((uint32_t)lane % EPI_REPL) * repl_stride : 0u);
                const uint8_t *gt = out1 + a.out_off + ((uint64_t)ts + RPT * tid) * nbytes;
                uint32_t raw[GQ][NBT];
#pragma unroll
                for (int q = 0; q < GQ; ++q) {  // all of the thread's rows in flight
#pragma unroll
                    for (int i = 0; i < NBT; ++i) raw[q][i] = reinterpret_cast<const uint32_t *>(gt + q * 4 * NBT)[i];
                }
#pragma unroll
                for (int q = 0; q < GQ; ++q) {
                    uint32_t w[NW][4];
                    cut4_rows_w<NBT>(raw[q], w);
#pragma unroll
                    for (int j = 0; j < 4; ++j) atomicAdd(&hrow[((uint32_t)(4 * q + j) >= jb ? N + 1 : 0u) + popc_row(w, j)], 1u);
                    if (want_cs) vadd_rows(w);
                }
                // 1-in-100 rows: at most one multiple of 100 among 16 consecutive positions; its row is read again (a cache hit)
                const uint32_t pos0 = ts + RPT * tid;
                const uint32_t r100 = (pos0 + 99u) / 100u;
                const uint32_t first = r100 * 100u - pos0;
                if (want100 && first < RPT) copy_row_w<NBT>(gt + first * nbytes, out100 + a.out100_off + (uint64_t)r100 * nbytes);
                tile += GQ - 1;
                continue;
            }
        }
        // ---- one tile: the thread's four rows ----
        unreplicate();
        const uint32_t tile_start = (tile - a.tile0) * PROBE_TILE;
        const uint32_t npos = min((uint32_t)PROBE_TILE, a.nkmers - tile_start);
        const uint32_t binlen = a.binlen, bin0 = tile_start / binlen, bin0_start = bin0 * binlen;
        const uint64_t row0 = a.bin_off + bin0;
        const bool big = binlen >= (uint32_t)PROBE_TILE;  // a tile spans at most 2 bins
        const bool windowed = binlen >= MINBIN;           // ... at most MAXB bins
        const uint32_t last_rel = (tile_start + npos - 1 - bin0_start) / binlen;
        const uint32_t binv = big ? 0u : 0xFFFFFFFFu / binlen + 1u;
        const bool fits = cur_row0 != ~0ull && (!windowed ? row0 == cur_row0 : (row0 >= cur_row0 && row0 + last_rel < cur_row0 + MAXB));
        if (!fits) {
            if (cur_row0 != ~0ull) {
                __syncthreads();
                flush_hist(N, hist, bins, cur_row0, tid, MAXB);
                __syncthreads();
            }
            cur_row0 = row0;
        }
        const uint32_t rel_base = (uint32_t)(row0 - cur_row0);
        const uint8_t *g = out1 + a.out_off + (uint64_t)tile_start * nbytes;
        const uint32_t nact = p0 < npos ? min(4u, npos - p0) : 0u;
        uint32_t w[NW][4];
        if (npos == (uint32_t)PROBE_TILE) {  // (block-uniform) a full tile: NBT aligned words
            uint32_t raw[NBT];
#pragma unroll
            for (int i = 0; i < NBT; ++i) raw[i] = reinterpret_cast<const uint32_t *>(g + (uint64_t)p0 * nbytes)[i];
            cut4_rows_w<NBT>(raw, w);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int t = 0; t < NW; ++t) {
                    uint32_t v = 0;
                    if ((uint32_t)j < nact) {
#pragma unroll
                        for (int bb = 0; bb < (NBT - 4 * t < 4 ? NBT - 4 * t : 4); ++bb) v |= (uint32_t)g[(uint64_t)(p0 + j) * nbytes + 4 * t + bb] << (8 * bb);
                    }
                    w[t][j] = v;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t pos = tile_start + p0 + j;
            const uint32_t pc = popc_row(w, j);
            if (windowed) {
                if ((uint32_t)j < nact) {
                    const uint32_t dpos = pos - bin0_start;
                    const uint32_t rel = big ? (dpos >= binlen ? 1u : 0u) : __umulhi(dpos, binv);
                    atomicAdd(&hist[(rel_base + rel) * (N + 1) + pc], 1u);
                }
            } else {
                hist_position((uint32_t)j < nact, pos, pc, N, binlen, bin0, bin0_start, rel_base, hist, bins, a.bin_off, lane, MAXB);
            }
        }
        if (nact) {  // 1-in-100 rows: at most one of 4 consecutive positions is a multiple of 100
            const uint32_t pos0 = tile_start + p0;
            const uint32_t r100 = (pos0 + 99u) / 100u;
            const uint32_t jsel = r100 * 100u - pos0;
            if (want100 && jsel < nact) copy_row_w<NBT>(g + (uint64_t)(p0 + jsel) * nbytes, out100 + a.out100_off + (uint64_t)r100 * nbytes);
        }
        if (want_cs) vadd_rows(w);  // (rows beyond npos are zero — a partial tile's words are built that way: adding them is harmless)
    }
    unreplicate();
    __syncthreads();
    if (cur_row0 != ~0ull) flush_hist(N, hist, bins, cur_row0, tid, MAXB);
}

// ---------------------------------------------------------------------------
// statistics of arbitrary row windows of a finished bitmap (genes, bins of any length): per window
// the histogram of row popcounts and, optionally, the per-genome column sums.  Not on the hot
// path: LDS atomics for the histogram, one ballot per genome bit and 64 rows for the columns.
// grid = (windows, pieces): piece p of a window takes its 256-row groups p, p + pieces, ...
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_window_stats(uint32_t N, const uint8_t *__restrict__ rows, uint64_t nrows,
                                                      const uint64_t *__restrict__ starts, const uint64_t *__restrict__ ends,
                                                      unsigned long long *__restrict__ hist_out,
                                                      unsigned long long *__restrict__ cs_out) {
    extern __shared__ uint32_t wsm[];
    uint32_t *hist = wsm, *cs = wsm + (N + 1);
    const int tid = threadIdx.x, lane = tid & 63;
    for (uint32_t i = tid; i < 2 * N + 1; i += 256) wsm[i] = 0;
    __syncthreads();
    const uint32_t nbytes = (N + 7) / 8, ndbs = (N + 31) / 32;
    const uint64_t s = starts[blockIdx.x], e = min(ends[blockIdx.x], nrows);
    if (s < e) {
        for (uint64_t g0 = s + 256ull * blockIdx.y; g0 < e; g0 += 256ull * gridDim.y) {
            const uint64_t r = g0 + tid;
            const bool active = r < e;
            uint32_t popc = 0;
            for (uint32_t d = 0; d < ndbs; ++d) {
                const uint32_t nb = min(4u, nbytes - 4 * d);
                uint32_t wv = 0;
                if (active)
                    for (uint32_t bb = 0; bb < nb; ++bb) wv |= (uint32_t)rows[r * nbytes + 4 * d + bb] << (8 * bb);
                popc += __popc(wv);
                if (cs_out) colsum_word(wv, d, N, cs, lane);
            }
            if (active) atomicAdd(&hist[min(popc, N)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i <= N; i += 256)
        if (hist[i]) atomicAdd(&hist_out[(uint64_t)blockIdx.x * (N + 1) + i], (unsigned long long)hist[i]);
    if (cs_out)
        for (uint32_t i = tid; i < N; i += 256)
            if (cs[i]) atomicAdd(&cs_out[(uint64_t)blockIdx.x * N + i], (unsigned long long)cs[i]);
}

// ---------------------------------------------------------------------------
// genome-sharded exchange (tables too big for one GPU): a rank's partial rows hold only the bits of
// the genomes it owns, so what crosses xGMI is a COMPACT block of bit columns — for every 64
// positions, one u64 per owned genome (bit l = position l) — all-gathered over RCCL and merged back
// into full rows: (n-1)/n row bytes received per position instead of the 2(n-1)/n of an all-reduce.
// Layout: tile t (PROBE_TILE positions) owns TILE_SLOTS = PROBE_TILE / 64 slots of `width` u64 words: word
// (TILE_SLOTS t + s) * width + j = genome g0 + j at positions 64 s .. 64 s + 63 of the tile.
// ---------------------------------------------------------------------------
// One-byte rows (a block of up to 8 genomes — config 5: ONE genome per GPU): a lane takes 16 consecutive positions (one
// aligned 16-byte load, the wave a whole tile) and gathers bit g of 8 row bytes into one byte with a multiply — two bytes per
// genome and lane: bytes 2 (lane % 4), 2 (lane % 4) + 1 of word (slot = lane / 4, genome j), one 16-bit store.  A wave takes
// COLS_TPW tiles, all of their loads in flight together.  (Rounds 2-4: a wave per 512 positions, 8 per lane — 4.7 x 10^7 waves
// of one load and one byte store each for config 5's 2.4 x 10^10 positions: 15 ms per pass at 1.6 TB/s, the launch's waves,
// not its bytes.)
constexpr uint32_t COLS_TPW = 4;  // tiles per wave of the one-byte-row column kernels
static_assert(PROBE_TILE == 1024, "k_cols_extract_b1 / k_cols_merge_b1: a wave's 64 lanes x 16 positions are one tile");
__global__ __launch_bounds__(256) void k_cols_extract_b1(uint32_t N, const AnchorDesc *__restrict__ ad,
                                                         const uint32_t *__restrict__ tile_contig, uint32_t tile_base,
                                                         uint32_t ntiles, const uint8_t *__restrict__ out1, uint32_t g0,
                                                         uint32_t width, uint8_t *__restrict__ dst) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t t0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * COLS_TPW;
    if (t0 >= ntiles) return;  // wave-uniform
    uint4 v[COLS_TPW];
    uint32_t valid[COLS_TPW];
#pragma unroll
    for (uint32_t k = 0; k < COLS_TPW; ++k) {
        v[k] = make_uint4(0, 0, 0, 0);
        valid[k] = 0;
        if (t0 + k < ntiles) {  // (wave-uniform)
            const uint32_t tile = tile_base + t0 + k;
            const AnchorDesc a = ad[tile_contig[tile]];
            const uint32_t p0 = (tile - a.tile0) * PROBE_TILE + 16 * lane;
            if (p0 < a.nkmers) {  // (rows are padded to 16 bytes per contig: the aligned 16-byte load stays inside)
                v[k] = *reinterpret_cast<const uint4 *>(out1 + a.out_off + p0);
                valid[k] = min(16u, a.nkmers - p0);
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < COLS_TPW; ++k) {
        if (t0 + k >= ntiles) break;  // (wave-uniform)
        uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {  // bytes of positions past the contig's end hold whatever follows: cleared
            const uint32_t nv = valid[k] > 4 * i ? min(4u, valid[k] - 4 * i) : 0u;
            w[i] &= nv >= 4 ? 0xFFFFFFFFu : (1u << (8 * nv)) - 1u;
        }
        uint8_t *o = dst + ((uint64_t)(t0 + k) * TILE_SLOTS + (lane >> 2)) * width * 8 + 2 * (lane & 3);
        for (uint32_t j = 0; j < width; ++j) {
            const uint32_t g = g0 + j;
            uint32_t b = 0;
            if (g < N) {  // bit g of rows 0..3 / 4..7 -> bits 0..3 / 4..7 (0x01020408: the four bits meet in bits 24..27)
                const uint32_t a0 = ((w[0] >> g) & 0x01010101u) * 0x01020408u, a1 = ((w[1] >> g) & 0x01010101u) * 0x01020408u;
                const uint32_t a2 = ((w[2] >> g) & 0x01010101u) * 0x01020408u, a3 = ((w[3] >> g) & 0x01010101u) * 0x01020408u;
                b = ((a0 >> 24) & 0xFu) | ((a1 >> 20) & 0xF0u) | ((a2 >> 16) & 0xF00u) | ((a3 >> 12) & 0xF000u);
            }
            *reinterpret_cast<uint16_t *>(o + 8 * j) = (uint16_t)b;
        }
    }
}

// the reverse for one-byte rows (N <= 8): a lane rebuilds the rows of 16 consecutive positions — its two bytes of each genome's
// word spread over 16 row bytes — and stores (or ORs) them as one aligned 16-byte word; COLS_TPW tiles per wave, their loads
// (the genomes' bytes and, when accumulating, the rows as they are) in flight together
__global__ __launch_bounds__(256) void k_cols_merge_b1(uint32_t N, const AnchorDesc *__restrict__ ad,
                                                       const uint32_t *__restrict__ tile_contig, uint32_t tile_base,
                                                       uint32_t ntiles, uint8_t *__restrict__ out1,
                                                       const uint8_t *__restrict__ src, uint32_t part0, uint32_t nparts,
                                                       uint64_t part_bytes, uint32_t per, uint32_t accumulate) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t t0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * COLS_TPW;
    if (t0 >= ntiles) return;  // wave-uniform
    const uint32_t gfirst = part0 * per, gend = min(N, (part0 + nparts) * per);
    uint4 old[COLS_TPW];
    uint4 *rowp[COLS_TPW];
#pragma unroll
    for (uint32_t k = 0; k < COLS_TPW; ++k) {
        old[k] = make_uint4(0, 0, 0, 0);
        rowp[k] = nullptr;
        if (t0 + k < ntiles) {  // (wave-uniform)
            const uint32_t tile = tile_base + t0 + k;
            const AnchorDesc a = ad[tile_contig[tile]];
            const uint32_t p0 = (tile - a.tile0) * PROBE_TILE + 16 * lane;
            if (p0 < a.nkmers) {
                rowp[k] = reinterpret_cast<uint4 *>(out1 + a.out_off + p0);
                if (accumulate) old[k] = *rowp[k];
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < COLS_TPW; ++k) {
        if (t0 + k >= ntiles) break;  // (wave-uniform)
        const uint8_t *in = src + ((uint64_t)(t0 + k) * TILE_SLOTS + (lane >> 2)) * per * 8 + 2 * (lane & 3);
        uint32_t w[4] = {old[k].x, old[k].y, old[k].z, old[k].w};
        for (uint32_t g = gfirst; g < gend; ++g) {
            const uint32_t part = g / per - part0, j = g % per;
            const uint32_t b2 = *reinterpret_cast<const uint16_t *>(in + (uint64_t)part * part_bytes + 8 * j);
            const uint32_t r0 = (b2 & 0xFFu) * 0x01010101u, r1 = (b2 >> 8) * 0x01010101u;  // bit i of a byte -> bit 0 of byte i
            w[0] |= ((((r0 & 0x08040201u) + 0x7F7F7F7Fu) >> 7) & 0x01010101u) << g;
            w[1] |= ((((r0 & 0x80402010u) + 0x7F7F7F7Fu) >> 7) & 0x01010101u) << g;
            w[2] |= ((((r1 & 0x08040201u) + 0x7F7F7F7Fu) >> 7) & 0x01010101u) << g;
            w[3] |= ((((r1 & 0x80402010u) + 0x7F7F7F7Fu) >> 7) & 0x01010101u) << g;
        }
        if (rowp[k]) *rowp[k] = make_uint4(w[0], w[1], w[2], w[3]);  // (bits of positions past nkmers are zero in the blocks: the padding stays zero)
    }
}

// Wider rows.  A lane owns one position; the wave one slot of 64.  The row bytes are read a 32-bit word at a time (one
// access per 32 genomes: the first version read a byte per genome in a loop that waited for each load in turn — 20 ps
// per row at 32 genomes per block, ten times the probe), a ballot per genome turns the word's bit into the slot's u64,
// kept by lane j and stored coalesced.
__global__ __launch_bounds__(256) void k_cols_extract(uint32_t N, const AnchorDesc *__restrict__ ad,
                                                      const uint32_t *__restrict__ tile_contig, uint32_t tile_base,
                                                      uint32_t ntiles, const uint8_t *__restrict__ out1, uint32_t g0,
                                                      uint32_t width, unsigned long long *__restrict__ dst) {
    struct __attribute__((packed)) U32 { uint32_t v; };
    const int lane = threadIdx.x & 63;
    const uint64_t slot = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // relative to the range's first tile
    if (slot >= (uint64_t)ntiles * TILE_SLOTS) return;  // wave-uniform
    const uint32_t tile = tile_base + (uint32_t)(slot / TILE_SLOTS), sub = (uint32_t)(slot % TILE_SLOTS);
    const AnchorDesc a = ad[tile_contig[tile]];
    const uint32_t nbytes = (N + 7) / 8;
    const uint32_t p = (tile - a.tile0) * PROBE_TILE + sub * 64 + lane;
    const bool active = p < a.nkmers;
    const uint8_t *row = out1 + a.out_off + (uint64_t)p * nbytes;
    const uint32_t gend = min(N, g0 + width);
    for (uint32_t j0 = 0; j0 < width; j0 += 64) {
        unsigned long long mine = 0;
        const uint32_t ga = g0 + j0, gz = min(gend, ga + 64u);  // genomes of this group of (up to) 64 columns
        for (uint32_t wd = ga >> 5; 32u * wd < gz; ++wd) {
            uint32_t v = 0;
            if (active) {  // (the word may reach past the row's last byte: byte loads there)
                if (4u * wd + 4u <= nbytes) v = reinterpret_cast<const U32 *>(row + 4u * wd)->v;
                else
                    for (uint32_t bb = 0; 4u * wd + bb < nbytes; ++bb) v |= (uint32_t)row[4u * wd + bb] << (8u * bb);
            }
            const uint32_t b_lo = max(ga, 32u * wd) - 32u * wd, b_hi = min(gz, 32u * wd + 32u) - 32u * wd;
            for (uint32_t b = b_lo; b < b_hi; ++b) {
                const unsigned long long m = __ballot((v >> b) & 1u);
                if ((uint32_t)lane == 32u * wd + b - ga) mine = m;
            }
        }
        if ((uint32_t)lane < min(64u, width - j0)) dst[slot * width + j0 + lane] = mine;  // (columns past N stay zero)
    }
}

// src = nparts blocks of part_words u64 each (block i = genomes (part0 + i) * per ...); the bits of those genomes are
// set in the rows from the blocks.  accumulate == 0: the rows are written whole (bits of genomes outside the
// blocks become 0); != 0: the blocks' bits are OR-ed into what the rows hold (genome blocks arriving pass by pass).
// The u64 of a genome and slot is the same for the whole wave: lane l FETCHES the one of genome 32 d + l (one coalesced
// access per 32 genomes) and the wave reads them lane by lane — as wave-uniform loads inside the genome loop every one
// of them was waited for in turn (0.4-0.7 ms per call at 64 genomes).
__global__ __launch_bounds__(256) void k_cols_merge(uint32_t N, const AnchorDesc *__restrict__ ad,
                                                    const uint32_t *__restrict__ tile_contig, uint32_t tile_base,
                                                    uint32_t ntiles, uint8_t *__restrict__ out1,
                                                    const unsigned long long *__restrict__ src, uint32_t part0,
                                                    uint32_t nparts, uint64_t part_words, uint32_t per, uint32_t accumulate) {
    const int lane = threadIdx.x & 63;
    const uint64_t slot = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (slot >= (uint64_t)ntiles * TILE_SLOTS) return;
    const uint32_t tile = tile_base + (uint32_t)(slot / TILE_SLOTS), sub = (uint32_t)(slot % TILE_SLOTS);
    const AnchorDesc a = ad[tile_contig[tile]];
    const uint32_t nbytes = (N + 7) / 8, ndbs = (N + 31) / 32;
    const uint32_t p = (tile - a.tile0) * PROBE_TILE + sub * 64 + lane;
    uint8_t *row = out1 + a.out_off + (uint64_t)p * nbytes;
    const uint32_t gfirst = part0 * per, gend = min(N, (part0 + nparts) * per);  // genomes the blocks cover
    for (uint32_t d = 0; d < ndbs; ++d) {
        if (accumulate && (32 * d + 32 <= gfirst || 32 * d >= gend)) continue;  // (uniform) word untouched by these blocks
        const uint32_t gl = 32u * d + ((uint32_t)lane & 31u);
        unsigned long long mine = 0;
        if (lane < 32 && gl >= gfirst && gl < gend) mine = src[(uint64_t)(gl / per - part0) * part_words + slot * per + gl % per];
        uint32_t w = 0;
        // bits of this word that the blocks cover (none: the word is written as zeros)
        const bool any = 32u * d < gend && 32u * d + 32u > gfirst;
        const uint32_t b_lo = any ? max(gfirst, 32u * d) - 32u * d : 0u, b_hi = any ? min(gend, 32u * d + 32u) - 32u * d : 0u;
        for (uint32_t b = b_lo; b < b_hi; ++b) {  // (uniform)
            const unsigned long long word = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), (int)b) << 32) |
                                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, (int)b);
            w |= (__builtin_amdgcn_inverse_ballot_w64(word) ? 1u : 0u) << b;
        }
        if (p < a.nkmers) {
            const uint32_t n = min(4u, nbytes - 4 * d);
            if (n == 4 && (nbytes & 3u) == 0) {  // (uniform) whole words of rows that start on word boundaries: one access, not four
                uint32_t *rw = reinterpret_cast<uint32_t *>(row + 4 * d);
                if (!accumulate) *rw = w;
                else if (w) *rw |= w;
            } else if (accumulate) {
                for (uint32_t bb = 0; bb < n; ++bb) {
                    const uint8_t add = (uint8_t)(w >> (8 * bb));
                    if (add) row[4 * d + bb] |= add;
                }
            } else {
                for (uint32_t bb = 0; bb < n; ++bb) row[4 * d + bb] = (uint8_t)(w >> (8 * bb));
            }
        }
    }
}

// every step-th row of bitmap.1 -> the low-resolution bitmap, for steps other than the 100 the statistics
// kernels fuse (index.py:101-106 lowres_step): one wave per tile, a handful of rows each
__global__ __launch_bounds__(64) void k_lowres(uint32_t N, const AnchorDesc *__restrict__ ad,
                                               const uint32_t *__restrict__ tile_contig, const uint8_t *__restrict__ out1,
                                               uint8_t *__restrict__ outlow, uint32_t step) {
    const uint32_t tile = blockIdx.x;
    const AnchorDesc a = ad[tile_contig[tile]];
    const uint32_t nbytes = (N + 7) / 8;
    const uint32_t ts = (tile - a.tile0) * PROBE_TILE, te = min(a.nkmers, ts + (uint32_t)PROBE_TILE);
    const uint64_t r0 = ((uint64_t)ts + step - 1) / step;
    for (uint64_t r = r0 + threadIdx.x; r * step < te; r += 64) {
        const uint8_t *src = out1 + a.out_off + r * step * nbytes;
        uint8_t *dst = outlow + a.out100_off + r * nbytes;
        for (uint32_t bb = 0; bb < nbytes; ++bb) dst[bb] = src[bb];
    }
}

#endif  // PG_MAIN_PART

// ---------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------
template <int W_C, bool TWO, int ROWMODE, int SLOTS>
static hipError_t probe_t(hipStream_t s, uint32_t ntiles, const SubTable &st, const uint64_t *seqw, const uint32_t *nmw,
                          const uint32_t *has_n, const SeqDesc *sd, const AnchorDesc *ad, const uint32_t *tile_contig,
                          const uint32_t *sched, uint32_t tile_base, uint8_t *out1, uint32_t nbytes, const RowCols &rc, const FuseArgs *fuse) {
    const bool m64 = W_C && st.m > 16;
    if constexpr (SLOTS == 8 && ROWMODE != 1 && ROWMODE != 3) {  // (the fused instantiations: rows of 2..8 bytes, 8-slot lines)
        if (fuse) {
            if (m64)
                hipLaunchKernelGGL((k_probe<W_C, TWO, ROWMODE, SLOTS, true, false, false, true>), dim3(ntiles), dim3(64), 0, s, st, seqw, nmw, has_n, sd, ad,
                                   tile_contig, sched, tile_base, out1, nbytes, rc, *fuse);
            else
                hipLaunchKernelGGL((k_probe<W_C, TWO, ROWMODE, SLOTS, false, false, false, true>), dim3(ntiles), dim3(64), 0, s, st, seqw, nmw, has_n, sd, ad,
                                   tile_contig, sched, tile_base, out1, nbytes, rc, *fuse);
            return hipGetLastError();
        }
    }
    if (fuse) return hipErrorInvalidValue;  // (launch_anchor only asks for what is instantiated)
    const FuseArgs none = {nullptr, nullptr, nullptr, 0, 0, 0};
    if (m64)
        hipLaunchKernelGGL((k_probe<W_C, TWO, ROWMODE, SLOTS, true>), dim3(ntiles), dim3(64), 0, s, st, seqw, nmw, has_n, sd, ad,
                           tile_contig, sched, tile_base, out1, nbytes, rc, none);
    else
        hipLaunchKernelGGL((k_probe<W_C, TWO, ROWMODE, SLOTS, false>), dim3(ntiles), dim3(64), 0, s, st, seqw, nmw, has_n, sd, ad,
                           tile_contig, sched, tile_base, out1, nbytes, rc, none);
    return hipGetLastError();
}

template <int W_C>
static hipError_t probe_w(hipStream_t s, uint32_t ntiles, const SubTable &st, const uint64_t *seqw, const uint32_t *nmw,
                          const uint32_t *has_n, const SeqDesc *sd, const AnchorDesc *ad, const uint32_t *tile_contig,
                          const uint32_t *sched, uint32_t tile_base, uint8_t *out1, uint32_t nbytes, const RowCols &rc, int rowmode,
                          const FuseArgs *fuse = nullptr) {
#define PG_A s, ntiles, st, seqw, nmw, has_n, sd, ad, tile_contig, sched, tile_base, out1, nbytes, rc, fuse
#define PG_K s, st, seqw, nmw, has_n, sd, ad, tile_contig, sched, tile_base, out1, nbytes, rc
    const FuseArgs none = {nullptr, nullptr, nullptr, 0, 0, 0};
    const bool m64 = W_C && st.m > 16;
    if (st.layout == LAYOUT_INLINE || st.layout == LAYOUT_SPLIT) {
        // more than 64 genomes: keys and their mask blocks in one line (inline, 65..96 genomes), or key lines + mask array (split)
        const bool inl = st.layout == LAYOUT_INLINE;
        if (fuse && (nbytes > 16u || !PG_FUSE_WIDE)) return hipErrorInvalidValue;
        if (inl) {
            if (fuse) {
#if PG_FUSE_WIDE
                if (m64) hipLaunchKernelGGL((k_probe<W_C, false, 0, 8, true, true, true, true>), dim3(ntiles), dim3(64), 0, PG_K, *fuse);
                else hipLaunchKernelGGL((k_probe<W_C, false, 0, 8, false, true, true, true>), dim3(ntiles), dim3(64), 0, PG_K, *fuse);
#endif
            } else {
                if (m64) hipLaunchKernelGGL((k_probe<W_C, false, 0, 8, true, true, true>), dim3(ntiles), dim3(64), 0, PG_K, none);
                else hipLaunchKernelGGL((k_probe<W_C, false, 0, 8, false, true, true>), dim3(ntiles), dim3(64), 0, PG_K, none);
            }
        } else {
            if (fuse) {
#if PG_FUSE_WIDE
                if (m64) hipLaunchKernelGGL((k_probe<W_C, false, 0, 8, true, true, false, true>), dim3(ntiles), dim3(64), 0, PG_K, *fuse);
                else hipLaunchKernelGGL((k_probe<W_C, false, 0, 8, false, true, false, true>), dim3(ntiles), dim3(64), 0, PG_K, *fuse);
#endif
            } else {
                if (m64) hipLaunchKernelGGL((k_probe<W_C, false, 0, 8, true, true>), dim3(ntiles), dim3(64), 0, PG_K, none);
                else hipLaunchKernelGGL((k_probe<W_C, false, 0, 8, false, true>), dim3(ntiles), dim3(64), 0, PG_K, none);
            }
        }
        return hipGetLastError();
    }
    if (st.slots == 16) {
        if (st.W == 2) {
            if (rowmode == 2) return probe_t<W_C, true, 2, 16>(PG_A);
            return probe_t<W_C, true, 0, 16>(PG_A);
        }
        if (rowmode == 1) return probe_t<W_C, false, 1, 16>(PG_A);
        return probe_t<W_C, false, 0, 16>(PG_A);
    }
    if (st.W == 2) {
        if (rowmode == 2) return probe_t<W_C, true, 2, 8>(PG_A);
        return probe_t<W_C, true, 0, 8>(PG_A);
    }
    if (rowmode == 3) return probe_t<W_C, false, 3, 8>(PG_A);
    if (rowmode == 1) return probe_t<W_C, false, 1, 8>(PG_A);
    if (rowmode == 4) return probe_t<W_C, false, 4, 8>(PG_A);
    if (rowmode == 5) return probe_t<W_C, false, 5, 8>(PG_A);
    if (rowmode == 6) return probe_t<W_C, false, 6, 8>(PG_A);
    return probe_t<W_C, false, 0, 8>(PG_A);
#undef PG_A
#undef PG_K
}

template <int W_C>
static hipError_t insert_tiles_w(hipStream_t s, uint32_t ntiles, const SubTable &st, int w, uint32_t bits, uint32_t count_mode,
                                 const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n, const SeqDesc *sd,
                                 const uint32_t *tile0, uint32_t ncontigs, unsigned long long *counters, uint32_t max_probe) {
    if (W_C && st.m > 16)
        hipLaunchKernelGGL((k_insert_tile<W_C, true>), dim3(ntiles), dim3(64), 0, s, st, w, bits, count_mode, seqw, nmw, has_n, sd,
                           tile0, ncontigs, counters, max_probe);
    else
        hipLaunchKernelGGL((k_insert_tile<W_C, false>), dim3(ntiles), dim3(64), 0, s, st, w, bits, count_mode, seqw, nmw, has_n, sd,
                           tile0, ncontigs, counters, max_probe);
    return hipGetLastError();
}

// the parts' entry points: the windows a unit instantiates (see the top of the file)
#define PG_PROBE_ARGS hipStream_t s, uint32_t w, uint32_t ntiles, const SubTable &st, const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n, \
                      const SeqDesc *sd, const AnchorDesc *ad, const uint32_t *tile_contig, const uint32_t *sched, uint32_t tile_base, uint8_t *out1,          \
                      uint32_t nbytes, const RowCols &rc, int rowmode, const FuseArgs *fuse
#define PG_PROBE_CASE(W) case W: return probe_w<W>(s, ntiles, st, seqw, nmw, has_n, sd, ad, tile_contig, sched, tile_base, out1, nbytes, rc, rowmode, fuse);
#define PG_INSERT_ARGS hipStream_t s, uint32_t win, uint32_t ntiles, const SubTable &st, int w, uint32_t bits, uint32_t count_mode, const uint64_t *seqw, \
                       const uint32_t *nmw, const uint32_t *has_n, const SeqDesc *sd, const uint32_t *tile0, uint32_t ncontigs,                          \
                       unsigned long long *counters, uint32_t max_probe
#define PG_INSERT_CASE(W) case W: return insert_tiles_w<W>(s, ntiles, st, w, bits, count_mode, seqw, nmw, has_n, sd, tile0, ncontigs, counters, max_probe);
// (each unit is a code object of its own, loaded by the HIP runtime when the first of its kernels is asked for: preload_anchor_kernels asks)
#if PG_HAS_PART(1)
hipError_t preload_part1() {
    hipFuncAttributes fa;
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_insert_tile<6, false>));
}
#endif
#if PG_HAS_PART(2)
hipError_t preload_part2() {
    hipFuncAttributes fa;
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_insert_tile<7, false>));
}
#endif
#if PG_HAS_PART(0)
hipError_t probe_part0(PG_PROBE_ARGS) {
    switch (w) { PG_PROBE_CASE(0) PG_PROBE_CASE(3) PG_PROBE_CASE(4) default: return hipErrorInvalidValue; }
}
hipError_t insert_part0(PG_INSERT_ARGS) {
    switch (win) { PG_INSERT_CASE(0) PG_INSERT_CASE(3) PG_INSERT_CASE(4) default: return hipErrorInvalidValue; }
}
#endif
#if PG_HAS_PART(1)
hipError_t probe_part1(PG_PROBE_ARGS) {
    switch (w) { PG_PROBE_CASE(5) PG_PROBE_CASE(6) default: return hipErrorInvalidValue; }
}
hipError_t insert_part1(PG_INSERT_ARGS) {
    switch (win) { PG_INSERT_CASE(5) PG_INSERT_CASE(6) default: return hipErrorInvalidValue; }
}
#endif
#if PG_HAS_PART(2)
hipError_t probe_part2(PG_PROBE_ARGS) {
    switch (w) { PG_PROBE_CASE(7) PG_PROBE_CASE(8) default: return hipErrorInvalidValue; }
}
hipError_t insert_part2(PG_INSERT_ARGS) {
    switch (win) { PG_INSERT_CASE(7) PG_INSERT_CASE(8) default: return hipErrorInvalidValue; }
}
#endif
#undef PG_PROBE_ARGS
#undef PG_PROBE_CASE
#undef PG_INSERT_ARGS
#undef PG_INSERT_CASE

#if PG_MAIN_PART  // ======== the launchers: part 0 (or the one unit) only ========
// the kernel's compile-time window must be the one the table was built with: the part that instantiates it
static hipError_t probe_window(hipStream_t s, uint32_t w, uint32_t ntiles, const SubTable &st, const uint64_t *seqw, const uint32_t *nmw,
                               const uint32_t *has_n, const SeqDesc *sd, const AnchorDesc *ad, const uint32_t *tile_contig, const uint32_t *sched,
                               uint32_t tile_base, uint8_t *out1, uint32_t nbytes, const RowCols &rc, int rowmode, const FuseArgs *fuse) {
    auto *part = (w == 0 || w == 3 || w == 4) ? probe_part0 : (w == 5 || w == 6) ? probe_part1 : (w == 7 || w == 8) ? probe_part2 : nullptr;
    if (!part) return hipErrorInvalidValue;
    return part(s, w, ntiles, st, seqw, nmw, has_n, sd, ad, tile_contig, sched, tile_base, out1, nbytes, rc, rowmode, fuse);
}

static int row_mode(uint32_t nbytes, const RowCols &rc) {
    if (nbytes == 1) return 1;
    if (nbytes == 8 && rc.col0 == 0 && rc.nb0 == 4 && rc.nb1 == 4) return 2;
    if (nbytes == 4 && rc.col0 == 0 && rc.nb0 == 4 && rc.nb1 == 0) return 4;
    if (nbytes == 2 && rc.col0 == 0 && rc.nb1 == 0) return 5;
    if (nbytes == 3 && rc.col0 == 0 && rc.nb1 == 0) return 6;
    return 0;
}

// columns_width != 0: the genome-sharded mode's narrow tables (up to 8 genomes, 8-slot lines) emit the block's bit
// columns (`columns_width` genomes wide) into `out1` instead of rows
hipError_t launch_anchor(hipStream_t s, const TableDesc &T, const uint64_t *seqw, const uint32_t *nmw,
                         const uint32_t *has_n, const SeqDesc *sd, const AnchorDesc *ad, const uint32_t *tile_contig,
                         const uint32_t *sched, uint32_t tile_base, uint32_t ntiles, uint8_t *out1, uint64_t out1_bytes,
                         uint32_t columns_width, const FuseArgs *fuse) {
    if (ntiles == 0) return hipSuccess;
    const uint32_t nbytes = (T.ngenomes + 7) / 8;
    hipError_t e = hipSuccess;
    (void)out1_bytes;
    // fused statistics: one sub-table writes whole rows of 2..16 bytes, 8-slot / inline / split lines (pg_api.hip asks only then)
    if (fuse && (columns_width || T.nsub != 1 || !fuse_rows_ok(nbytes) || fuse->ngenomes != T.ngenomes ||
                 (T.sub[0].layout == LAYOUT_SLOTS && T.sub[0].slots != 8)))
        return hipErrorInvalidValue;
    if (columns_width) {
        const SubTable &st = T.sub[0];
        if (T.nsub != 1 || st.layout != LAYOUT_SLOTS || st.W != 1 || st.slots != 8 || T.ngenomes > (uint32_t)COLS_G ||
            columns_width > (uint32_t)COLS_G || columns_width < T.ngenomes)
            return hipErrorInvalidValue;
        RowCols rc;
        rc.col0 = T.ngenomes;  // (genomes of the block)
        rc.nb0 = rc.nb1 = rc.words = 0;
        const uint32_t w = st.m ? st.k - st.m + 1 : 0;
        return probe_window(s, w, ntiles, st, seqw, nmw, has_n, sd, ad, tile_contig, sched, tile_base, out1, columns_width, rc, 3, nullptr);
    }
    for (uint32_t si = 0; si < T.nsub; ++si) {
        const SubTable &st = T.sub[si];
        RowCols rc;  // every sub-table writes all bytes of the columns it owns, so rows need no zero fill
        rc.col0 = 4 * st.word0;
        rc.nb0 = min(4u, nbytes - rc.col0);
        rc.nb1 = (st.W == 2 && nbytes > rc.col0 + 4) ? min(4u, nbytes - rc.col0 - 4) : 0;
        rc.words = (nbytes % 4 == 0 && rc.nb0 == 4 && (rc.nb1 == 0 || rc.nb1 == 4)) ? 1u : (nbytes == 2 ? 2u : 0u);
        if (T.nsub == 1 && (nbytes == 3 || (nbytes >= 5 && nbytes <= 7))) rc.words = 3u;
        if (rc.words == 1 && rc.nb1 == 4 && rc.col0 % 8 == 0 && nbytes % 8 == 0) rc.words = 4u;  // (one 8-byte store per row: -9 % probe time at N=128)
        const int rm = (T.nsub == 1) ? row_mode(nbytes, rc) : 0;
        const uint32_t w = st.m ? st.k - st.m + 1 : 0;
        e = probe_window(s, w, ntiles, st, seqw, nmw, has_n, sd, ad, tile_contig, sched, tile_base, out1, nbytes, rc, rm, fuse);
        if (e != hipSuccess) return e;
    }
    return e;
}

hipError_t launch_rows_epilogue(hipStream_t st, uint32_t ngenomes, const AnchorDesc *ad, const uint32_t *tile_contig,
                                uint32_t ntiles, const uint8_t *out1, uint8_t *out100, uint32_t *bins,
                                unsigned long long *colsums, uint32_t flags, const uint2 *d_ranges, uint32_t nranges,
                                uint32_t range_tiles) {
    // d_ranges: the launch covers these nranges tile ranges (range_tiles tiles in all) instead of [0, ntiles)
    if (ntiles == 0 || (d_ranges && (nranges == 0 || range_tiles == 0))) return hipSuccess;
    const uint32_t work_tiles = d_ranges ? range_tiles : ntiles;
    const uint32_t maxb = epi_maxb_for(ngenomes);
    flags = (flags & 0xFFFF00FFu) | (maxb << 8);
    size_t lds = (((maxb * (ngenomes + 1) + 3) & ~3u) + ((ngenomes + 3) & ~3u)) * 4 + 16;
    // contiguous tile ranges per workgroup: enough workgroups to fill every CU, but no fewer than
    // a minimum number of tiles each so that the end-of-range reductions stay amortised
    const uint32_t maxg = 256u * (2048u / EPI_THREADS);
    // long ranges (PG_EPI_MIN_TILES tiles) keep the pass light beside a concurrent k_probe; but never
    // fewer than ~1024 workgroups (4 per CU) as long as each still gets 16 tiles, or small inputs
    // turn latency-bound
    uint32_t grid = (work_tiles + PG_EPI_MIN_TILES - 1) / PG_EPI_MIN_TILES;
    grid = std::max(grid, std::min(1024u, work_tiles / 16u));
    grid = grid < 1 ? 1 : (grid > maxg ? maxg : grid);
    uint32_t wpr = 0;
    auto per_range = [&]() {  // (ranges: the same number of workgroups for each, the grid a multiple of the range count)
        if (!d_ranges) return;
        wpr = std::max(1u, grid / nranges);
        grid = wpr * nranges;
    };
    // persistent workgroups: no more of them than the device holds at once (a second, partly filled round of
    // workgroups would leave CUs idle at the end: 8192 waves over 5120 slots cost the one-byte kernel 20 %)
    auto fit = [&](const void *kern, size_t lds_bytes) {
        static std::mutex mu;
        static std::map<std::pair<const void *, size_t>, uint32_t> caps;  // (the query is a driver call: once per kernel and LDS size)
        std::lock_guard<std::mutex> lk(mu);
        auto it = caps.find({kern, lds_bytes});
        if (it == caps.end()) {
            int per_cu = 0, dev = 0, cus = 0;
            uint32_t cap = ~0u;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, EPI_THREADS, lds_bytes) == hipSuccess && per_cu >= 1 &&
                hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
                cap = (uint32_t)per_cu * (uint32_t)cus;
            it = caps.emplace(std::make_pair(kern, lds_bytes), cap).first;
        }
        if (grid > it->second) grid = it->second;
    };
    const uint32_t nbytes = (ngenomes + 7) / 8;
    if (nbytes <= 8) {
        auto kern = k_epilogue<0, 1>;
        switch (nbytes) {
            case 2: kern = k_epilogue<1, 2>; break;
            case 3: kern = k_epilogue<1, 3>; break;
            case 4: kern = k_epilogue<1, 4>; break;
            case 5: kern = k_epilogue<1, 5>; break;
            case 6: kern = k_epilogue<1, 6>; break;
            case 7: kern = k_epilogue<1, 7>; break;
            case 8: kern = k_epilogue<1, 8>; break;
            default: break;
        }
        fit(reinterpret_cast<const void *>(kern), lds);
        per_range();
        hipLaunchKernelGGL(kern, dim3(grid), dim3(EPI_THREADS), lds, st, ngenomes, ad, tile_contig, ntiles, out1, out100, bins,
                           colsums, flags, d_ranges, wpr);
    }
    else if (PG_EPI_W && nbytes <= 16) {  // 65..128 genomes: 16 rows per thread, three or four words per row
        auto kern = k_epilogue_w<16>;
        switch (nbytes) {
            case 9: kern = k_epilogue_w<9>; break;
            case 10: kern = k_epilogue_w<10>; break;
            case 11: kern = k_epilogue_w<11>; break;
            case 12: kern = k_epilogue_w<12>; break;
            case 13: kern = k_epilogue_w<13>; break;
            case 14: kern = k_epilogue_w<14>; break;
            case 15: kern = k_epilogue_w<15>; break;
            default: break;
        }
        fit(reinterpret_cast<const void *>(kern), lds);
        per_range();
        hipLaunchKernelGGL(kern, dim3(grid), dim3(EPI_THREADS), lds, st, ngenomes, ad, tile_contig, ntiles, out1, out100, bins,
                           colsums, flags, d_ranges, wpr);
    }
    else {  // chunk-parallel: one launch, every row read once
        const uint32_t C = (nbytes + 15) / 16;
        if (C > 64) return hipErrorInvalidValue;
        const size_t lds_c = (((maxb * (ngenomes + 1) + 3) & ~3u) + ((64u / C + 15u) / 16u) * 128u * C) * 4 + 16;
        const bool exact = nbytes == 16u * C;
        auto kern = k_epilogue_chunks<0, false>;  // (compile-time C: the lanes-per-row shuffles and index arithmetic unroll)
        switch (C) {
            case 1: kern = exact ? k_epilogue_chunks<1, true> : k_epilogue_chunks<1, false>; break;
            case 2: kern = exact ? k_epilogue_chunks<2, true> : k_epilogue_chunks<2, false>; break;
            case 3: kern = k_epilogue_chunks<3, false>; break;
            case 4: kern = exact ? k_epilogue_chunks<4, true> : k_epilogue_chunks<4, false>; break;
            default: break;
        }
        fit(reinterpret_cast<const void *>(kern), lds_c);
        per_range();
        hipLaunchKernelGGL(kern, dim3(grid), dim3(EPI_THREADS), lds_c, st, ngenomes, ad, tile_contig, ntiles, out1, out100, bins,
                           colsums, flags, d_ranges, wpr);
    }
    return hipGetLastError();
}

// The tiles' counters of a fused launch (FuseArgs) into the bins and the per-contig column sums.  A workgroup takes TR_TILES
// consecutive tiles; a thread owns an item — a histogram field (bin 0 / 1 of the tile, popcount) or a genome's column — adds its
// values up while the item's destination (the bin's row, the contig's row) stays the same, and flushes one atomic add when it
// changes: a bin of 200 000 rows is 195 tiles long.  Tiles of contigs whose bins are shorter than a tile carry no counters (the
// statistics pass did them).  Reads (4 (N + 1) + 64 ceil(nbytes / 4)) bytes per tile where the pass read 1024 rows.
constexpr uint32_t TR_TILES = 32;
__global__ __launch_bounds__(256) void k_tile_reduce(const FuseArgs fo, const AnchorDesc *__restrict__ ad, const uint32_t *__restrict__ tile_contig,
                                                     uint32_t ntiles, uint32_t *__restrict__ bins, unsigned long long *__restrict__ colsums,
                                                     uint32_t want_cs) {
    const uint32_t N = fo.ngenomes, N1 = N + 1u;
    const uint32_t nitems = 2u * N1 + (want_cs ? N : 0u);
    // a workgroup = as many groups of TR_TILES tiles as its threads hold items for (N = 12: 38 items, 6 groups per 256 threads)
    const uint32_t per = max(1u, blockDim.x / nitems), slot = threadIdx.x / nitems;
    const uint32_t it0 = nitems <= blockDim.x ? threadIdx.x - slot * nitems : threadIdx.x;
    const uint32_t grp = blockIdx.x * per + (nitems <= blockDim.x ? slot : 0u);
    if (nitems <= blockDim.x && slot >= per) return;
    const uint32_t t0 = grp * TR_TILES;
    if (t0 >= ntiles) return;
    const uint32_t t1 = min(ntiles, t0 + TR_TILES);
    for (uint32_t it = it0; it < nitems; it += blockDim.x) {
        const bool is_hist = it < 2u * N1;
        const uint32_t rel = is_hist ? it / N1 : 0u, pc = it - rel * N1, g = it - 2u * N1;
        // where the item's u16 sits in a tile's counters
        const uint32_t word = is_hist ? (it >> 1) : (g >> 5) * 16u + (g & 15u), shift = is_hist ? 16u * (it & 1u) : ((g & 16u) ? 16u : 0u);
        const uint32_t *src = (is_hist ? fo.tile_hist : fo.tile_cs) + word;
        const uint32_t stride = is_hist ? fo.hw : fo.csw;
        uint32_t v[TR_TILES];  // (every tile's word requested at once: the walk below is serial)
#pragma unroll
        for (uint32_t j = 0; j < TR_TILES; ++j) v[j] = src[(uint64_t)min(t0 + j, t1 - 1u) * stride];
        unsigned long long acc = 0, key = ~0ull;
        uint32_t cur_c = ~0u;
        AnchorDesc a;
        a.binlen = 0, a.tile0 = 0, a.bin_off = 0;
        auto flush = [&]() {
            if (acc) {
                if (is_hist) atomicAdd(&bins[key], (uint32_t)acc);
                else atomicAdd(&colsums[key], acc);
            }
            acc = 0;
        };
#pragma unroll
        for (uint32_t j = 0; j < TR_TILES; ++j) {
            const uint32_t t = t0 + j;
            if (t >= t1) break;
            const uint32_t c = tile_contig[t];
            if (c != cur_c) {
                cur_c = c;
                a = ad[c];
            }
            if (a.binlen < (uint32_t)PROBE_TILE) continue;  // (not fused: its counters were never written)
            const unsigned long long k2 = is_hist ? (a.bin_off + (uint64_t)(t - a.tile0) * PROBE_TILE / a.binlen + rel) * N1 + pc
                                                  : (unsigned long long)c * N + g;
            if (k2 != key) {
                flush();
                key = k2;
            }
            acc += (v[j] >> shift) & 0xFFFFu;
        }
        flush();
    }
}

hipError_t launch_tile_reduce(hipStream_t st, const FuseArgs &fo, const AnchorDesc *ad, const uint32_t *tile_contig, uint32_t ntiles,
                              uint32_t *bins, unsigned long long *colsums, uint32_t want_colsums) {
    if (ntiles == 0) return hipSuccess;
    const uint32_t nitems = 2u * (fo.ngenomes + 1u) + (want_colsums ? fo.ngenomes : 0u);
    const uint32_t per = std::max(1u, 256u / nitems), groups = (ntiles + TR_TILES - 1) / TR_TILES;
    hipLaunchKernelGGL(k_tile_reduce, dim3((groups + per - 1) / per), dim3(256), 0, st, fo, ad, tile_contig, ntiles, bins, colsums,
                       want_colsums);
    return hipGetLastError();
}

hipError_t launch_window_stats(hipStream_t st, uint32_t ngenomes, const uint8_t *rows, uint64_t nrows, uint32_t nwin,
                               uint32_t pieces, const uint64_t *starts, const uint64_t *ends, unsigned long long *hist,
                               unsigned long long *cs) {
    if (nwin == 0) return hipSuccess;
    hipLaunchKernelGGL(k_window_stats, dim3(nwin, pieces), dim3(256), (2 * ngenomes + 1) * 4, st, ngenomes, rows, nrows,
                       starts, ends, hist, cs);
    return hipGetLastError();
}

hipError_t launch_cols_extract(hipStream_t st, uint32_t ngenomes, const AnchorDesc *ad, const uint32_t *tile_contig,
                               uint32_t tile_base, uint32_t ntiles, const uint8_t *out1, uint32_t g0, uint32_t width, void *dst) {
    if (ntiles == 0 || width == 0) return hipSuccess;
    if (ngenomes <= 8 && g0 < 8)  // one-byte rows
        hipLaunchKernelGGL(k_cols_extract_b1, dim3((unsigned)(((uint64_t)ntiles + 4 * COLS_TPW - 1) / (4 * COLS_TPW))), dim3(256), 0, st, ngenomes, ad, tile_contig, tile_base,
                           ntiles, out1, g0, width, static_cast<uint8_t *>(dst));
    else
        hipLaunchKernelGGL(k_cols_extract, dim3((unsigned)(((uint64_t)ntiles * TILE_SLOTS + 3) / 4)), dim3(256), 0, st, ngenomes, ad,
                           tile_contig, tile_base, ntiles, out1, g0, width, static_cast<unsigned long long *>(dst));
    return hipGetLastError();
}

hipError_t launch_cols_merge(hipStream_t st, uint32_t ngenomes, const AnchorDesc *ad, const uint32_t *tile_contig,
                             uint32_t tile_base, uint32_t ntiles, uint8_t *out1, const void *src, uint32_t part0,
                             uint32_t nparts, uint64_t part_words, uint32_t per, uint32_t accumulate) {
    if (ntiles == 0 || per == 0 || nparts == 0) return hipSuccess;
    if (ngenomes <= 8)  // one-byte rows
        hipLaunchKernelGGL(k_cols_merge_b1, dim3((unsigned)(((uint64_t)ntiles + 4 * COLS_TPW - 1) / (4 * COLS_TPW))), dim3(256), 0, st, ngenomes, ad, tile_contig, tile_base,
                           ntiles, out1, static_cast<const uint8_t *>(src), part0, nparts, part_words * 8, per, accumulate);
    else
        hipLaunchKernelGGL(k_cols_merge, dim3((unsigned)(((uint64_t)ntiles * TILE_SLOTS + 3) / 4)), dim3(256), 0, st, ngenomes, ad,
                           tile_contig, tile_base, ntiles, out1, static_cast<const unsigned long long *>(src), part0, nparts,
                           part_words, per, accumulate);
    return hipGetLastError();
}

hipError_t launch_lowres(hipStream_t st, uint32_t ngenomes, const AnchorDesc *ad, const uint32_t *tile_contig,
                         uint32_t ntiles, const uint8_t *out1, uint8_t *outlow, uint32_t step) {
    if (ntiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_lowres, dim3(ntiles), dim3(64), 0, st, ngenomes, ad, tile_contig, out1, outlow, step);
    return hipGetLastError();
}

#ifdef PG_PHASE_TIMING
}  // namespace pg
extern "C" int pg_debug_phase_cycles(unsigned long long *out16, int reset) {
    hipDeviceSynchronize();
    static unsigned long long all[1024 * 16];
    if (hipMemcpyFromSymbol(all, HIP_SYMBOL(pg::pg_phase_cycles), sizeof all) != hipSuccess) return -1;
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    for (int b = 0; b < 1024; ++b)
        for (int i = 0; i < 16; ++i) out16[i] += all[b * 16 + i];
    if (reset) {
        for (auto &x : all) x = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(pg::pg_phase_cycles), all, sizeof all) != hipSuccess) return -1;
    }
    return 0;
}
namespace pg {
#endif
hipError_t preload_anchor_kernels() {  // (any kernel of this unit loads its code object; the other parts' load with their first launch)
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k_lowres));
    if (e == hipSuccess) e = preload_part1();
    if (e == hipSuccess) e = preload_part2();
    return e;
}

// every k-mer of the contigs described by sd / tile0 (tile0[c] = first tile of contig c; tile0[ncontigs] = ntiles)
hipError_t launch_insert_tiles(hipStream_t s, const SubTable &st, int w, uint32_t bits, int count_mode, const uint64_t *seqw,
                               const uint32_t *nmw, const uint32_t *has_n, const SeqDesc *sd, const uint32_t *tile0,
                               uint32_t ncontigs, uint32_t ntiles, unsigned long long *counters, uint32_t max_probe) {
    if (ntiles == 0) return hipSuccess;
    const uint32_t win = st.m ? st.k - st.m + 1 : 0;
    auto *part = (win == 0 || win == 3 || win == 4) ? insert_part0 : (win == 5 || win == 6) ? insert_part1 : (win == 7 || win == 8) ? insert_part2 : nullptr;
    if (!part) return hipErrorInvalidValue;
    return part(s, win, ntiles, st, w, bits, (uint32_t)count_mode, seqw, nmw, has_n, sd, tile0, ncontigs, counters, max_probe);
}
#endif  // PG_MAIN_PART

}  // namespace pg
