// pg_device.h — device-side building blocks shared by the gfx950 kernels.
//
// Table layout (up to 64 genomes: one sub-table of W = 1 or 2 mask words per slot, described here; 65..96 genomes: ONE table in
// the inline layout — 6 bare keys and their mask blocks in one line; more genomes: ONE table in the split layout — bare keys
// in the lines, all mask words of a slot in a second array — see SubTable):
//   line   = 128 bytes, 128-byte aligned: MI355X moves 128 B per random HBM access whatever
//            the request size (tools/gather_bench.hip: 32/64/128-B random gathers all run at
//            ~50 G requests/s), so a probe fetches — and uses — a whole line
//          = 8 slots x { u64 key ; u32 mask0 ; u32 mask1 }   (mask1 unused when W == 1);
//            256-byte lines of 16 slots exist as a tuning knob (PG_TABLE_SLOTS=16)
//   k_probe fetches a line once per run of positions (8 or 16 lanes x 16 B, coalesced) into LDS
//   and every lane scans its line's slots there.
//   EMPTY key = ~0 (never a canonical k-mer for k <= 32: the all-T k-mer's reverse complement
//   is 0).  Keys only ever go EMPTY -> key, masks only gain bits: inserts are one 64-bit CAS +
//   one 32-bit OR, no locks.
//
// Home line = LOCALITY hash.  For 20 <= k <= 32 the home of a k-mer is a hash of its
// MINIMIZER: the smallest (in a scrambled order) canonical m-mer among its w = k-m+1 m-mers
// (w = 3..8, see minimizer_length).  Consecutive k-mers of a sequence share their minimizer for
// ~(w+1)/2 positions, so consecutive anchor positions probe the SAME line: one HBM fetch
// serves a run of positions.  Other k fall back to hashing the k-mer itself (m = 0).
// Collisions: double hashing by line — the group's own sequence for its first GROUP_CHAIN lines, the
// key's own sequence beyond (advance_line); a lookup moves to the next line only when the key is
// absent AND the line has no EMPTY slot.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace pg {

constexpr uint64_t EMPTY_KEY = ~0ull;
// A retired slot: holds no key and is not empty either (lines keep filling front to back behind it).  Written by
// wave_insert_batch over the later of two copies of a key that two racing claims put into one line.  Like EMPTY_KEY it
// can never be a canonical k-mer: ~0 is TT..T and ~0 - 1 is TT..TG, whose reverse complements AA..A / CAA..A are smaller.
constexpr uint64_t TOMB_KEY = ~0ull - 1;
// slots per table line: 8 (128-byte lines) or 16 (256-byte lines, tuning knob); a property of the
// sub-table (SubTable::slots)
constexpr int MAX_SUB = 1;           // a pan table is ONE sub-table (slots layout up to 64 genomes, split layout beyond)
constexpr uint32_t MAX_WORDS = 64;   // mask words per slot of the split layout => up to 2048 genomes

// Two layouts of a sub-table:
//   LAYOUT_SLOTS (up to 64 genomes per sub-table): line = `slots` 16-byte slots {u64 key, u32 mask0, u32 mask1}
//   LAYOUT_SPLIT (more than 64 genomes, ONE table whatever N): line = 16 bare keys (128 bytes) in `buckets`; the
//                W = ceil(N/32) mask words of slot s of line l live in a second array, at word (l * 16 + s) * W —
//                a probe fetches (and scans) key lines only and then reads exactly the mask words of its hit
struct SubTable {
    uint8_t *buckets;
    uint64_t nbuckets;  // lines
    uint32_t W;         // mask words per slot in use
    uint32_t word0;     // first 32-genome group covered
    uint32_t k;
    uint32_t m;         // minimizer length (0 = hash the whole k-mer)
    uint32_t slots;     // slots (keys) per line: 8 or 16
    uint32_t layout;    // LAYOUT_SLOTS / LAYOUT_SPLIT
    uint8_t *masks;     // LAYOUT_SPLIT only
};
constexpr uint32_t LAYOUT_SLOTS = 0, LAYOUT_SPLIT = 1, LAYOUT_INLINE = 2;
constexpr uint32_t SPLIT_KEYS = 16;  // keys per line of the split layout
//   LAYOUT_INLINE (65..128 genomes, round 5): a 128-byte line = S bare keys followed by the S slots' mask blocks of W words —
//                S = 6 at W = 3 (48 + 72 bytes), S = 5 at W = 4 (40 + 80 bytes): ONE fetch brings a hit's mask words with its
//                key, where the split layout's probe follows its key line with a dependent, divergent gather
__host__ __device__ __forceinline__ uint32_t inline_slots(uint32_t W) { return 120u / (8u + 4u * W); }

__host__ __device__ __forceinline__ uint32_t line_bytes(const SubTable &st) {
    return st.layout == LAYOUT_INLINE ? 128u : (st.layout == LAYOUT_SPLIT ? 8u : 16u) * st.slots;
}
// bytes from one key of a line to the next
__host__ __device__ __forceinline__ uint32_t key_stride(const SubTable &st) { return st.layout == LAYOUT_SLOTS ? 16u : 8u; }
__host__ __device__ __forceinline__ uint64_t table_bytes(const SubTable &st) {
    return st.nbuckets * line_bytes(st) + (st.layout == LAYOUT_SPLIT ? st.nbuckets * st.slots * 4ull * st.W : 0ull);
}
__device__ __forceinline__ unsigned long long *key_ptr(const SubTable &st, uint64_t line, uint32_t s) {
    return reinterpret_cast<unsigned long long *>(st.buckets + line * line_bytes(st) + key_stride(st) * s);
}
__device__ __forceinline__ uint32_t *mask_ptr(const SubTable &st, uint64_t line, uint32_t s, uint32_t w) {
    if (st.layout == LAYOUT_SPLIT) return reinterpret_cast<uint32_t *>(st.masks) + (line * st.slots + s) * st.W + w;
    if (st.layout == LAYOUT_INLINE) return reinterpret_cast<uint32_t *>(st.buckets + line * 128u + 8u * st.slots) + s * st.W + w;
    return reinterpret_cast<uint32_t *>(st.buckets + line * (16u * st.slots) + 16u * s + 8u) + w;
}

struct TableDesc {
    SubTable sub[MAX_SUB];
    uint32_t nsub;
    uint32_t ndbs;  // total 32-genome groups
    uint32_t k;
    uint32_t ngenomes;
};


// Minimizer geometry of a table: w m-mers of m = k-w+1 bases per k-mer (m = 0: hash the k-mer
// itself, k < 20).  Two forces (measured on MI355X):
//  * a wide window gives long runs of positions per fetched line — at w = 7 / 8 a 64-lane batch
//    meets about 13 distinct lines, which ONE staging step (16 lines) takes; w = 4 meets 24 (two
//    steps), and w = 5 / 6 sit just above 16 and pay the second step for a few lines (measured
//    worse than w = 4) — a narrow one gives small minimizer groups (a group = all variants of a
//    locus, from every genome) that fit their home line.  Since the overflow levels are staged and
//    the build claims a run's slots in one round, the wide window wins whatever the genome count
//    (tools/w_sweep_wide.sh, profiles/r2_w_sweep.txt; G k-mers/s at w = 8 / 7 / 6 / 5 / 4 —
//    k=31: 27 genomes 136 / 133 / 120 / 114 / 118, 64: 123 / 126 / 113 / 110 / 116, 128: 69 / 69 / 65 / 63 / 63;
//    k=21 (m >= 15: w <= 7): 27 genomes - / 133 / 123 / 116 / 119, 64: - / 118 / 112 / 109 / 111).  Round 1's kernels
//    had it the other way round for more than 16 genomes (64 genomes k=21: 89 G at w = 4, 61 at w = 7).
//  * the m-mers must stay long enough that distinct loci rarely share one: with 4^m below ~4x the
//    number of keys the groups merge and throughput collapses (k=21, 100 Mb genomes: m=15 100 G
//    k-mers/s, m=14 84 G, m=13 23 G; 27 x 40 Mb: m=14 123 G against 133 at m=15).
//    What merges groups is the number of distinct LOCI per m-mer, i.e. the non-redundant length L of the
//    pangenome, not its key count (a locus of a many-genome pangenome contributes a key per variant): at full size
//    (tools/m_sweep_full.sh, profiles/r2_m_sweep_full.txt, k=21, G k-mers/s at m = 15 / 16 / 18) 27 x 135 Mb
//    (7.8e8 keys, 4^15 = 8 L) 130 / 123 / 121; 64 x 200 Mb (2.4e9 keys, 4^15 = 5.4 L) 100 / 114 / 115;
//    8 x 3 Gb (3.4e9 keys; m = 16 / 17 / 18) 120 / 134 / 140.  L is known to the library as the length of the FIRST
//    sequence set inserted into an empty table (pg_table_insert_seqset settles m again then).
// Round 4 (k_probe's batches end in front of their 17th run, no halo lanes: profiles/r4b_m_sweep.txt) moved the balance:
// a batch now holds 16 runs whatever the window, so a narrow window means a short batch (w = 4: 40 positions of 64
// lanes) and costs more than it did — w 8 -> 7 nothing, 7 -> 6 about 3.5 %, 6 -> 5 another 7 %, 5 -> 4 another 9 % —
// while merged groups cost what they always did: with r = 4^m / L (L = the pangenome's non-redundant length) a launch
// loses about 50 % / r (r = 1.5: a third; 5: 10 %; 10: 5 %; 40: 1 %).  Measured on the new kernel (k=21, G k-mers/s at
// m = 15 / 16 / 17 / 18): 8 x 100 Mb 197 / 199 / 182 / -, 8 x 200 Mb 187 / 200 / - / 165, 8 x 300 Mb 174 / 199 / - / 166,
// 27 x 135 Mb (config 3) 162 / 177, 27 x 160 Mb 162 / 187 / - / 152, 4 x 700 Mb 135 / 179 / 174 / 159, 8 x 30 Mb 180 / 174;
// at 3 % divergence 8 x 100 Mb 153 / 165 / - / 142, 27 x 40 Mb 133 / 149 / - / 134; k=20 (w = 6 / 5 / 4 / 3) 193 / 185 / 166 /
// 141; k=22 (w = 8 / 7 / 6 / 5) 192 / 206 / 197 / 184.  More variants per locus (divergence, genome count) act like a longer
// pangenome, but less than in proportion: L_eff = L * sqrt(keys / (2.3 L)) (2.3 keys per locus is what 8 genomes at 1 % have;
// 64 x 160 Mb, 12 keys per locus: 139 / 137 / 126 at m = 16 / 17 / 18), the factor held to 4 (an expected key count far above
// what the first sequence set supports is an estimate gone wrong); 8 x 3 Gb: 153 / 169 / 156 at m = 16 / 17 / 18.
// So: m = the candidate (window 3..wmax, m >= 15) with the smallest  window_cost(w) + 50 / r  [percent];
//   L_eff = 1.5e8 while neither a key count nor a sequence is known (m = 16 for k = 21);
//   once a key count is known m stays within two bases of ceil(log4(4 * keys)) from below (a first sequence set much
//   shorter than the genomes that follow must not talk m down).
// m-mers longer than 16 bases use 64-bit arithmetic.
constexpr uint32_t MZ_WMIN = 3, MZ_WMAX = 8;
// `cosched`: how many anchor genomes a probe launch co-schedules against this table (0: not known — several, the product's
// default; 1: the table will be probed one genome at a time — a single-anchor `panagram index`, the run_anchor CLI with one
// FASTA, py_kmc_api-style GetCountersForRead callers, unrelated sequences).  Co-scheduled launches take most table lines
// from L2 and are bound by instruction issue: the window's price is the shorter batch (the table above).  A launch WITHOUT a
// partner fetches every line from HBM at the random-line rate of the memory system, and its time goes with the lines per
// position, about 2 / (w + 1): the window's price is steeper there and the same rule picked the narrower window for both —
// round 4 gained 1 % co-scheduled at configs[1] with m = 16 and lost 9 % per genome (125 -> 115 G k-mers/s).
// `load` (round 6): keys per slot the table is created for (0.375 = the library's 3 keys per 8-slot line).  In a SPARSER table —
// what Index.build_table asks for where HBM is plentiful — a group that outgrows its home line finds the next lines of its
// sequence empty more often, and merged groups cost less: at 1.5 keys per line 8 x 100 Mb runs 3.20 ms at m = 15 against 3.33 at
// m = 16 and 3.71 at m = 17 (at 3 keys per line m = 16 was 1 % ahead), 27 x 40 Mb the same at m = 15 and 16
// (profiles/r6n_m_sweep_roomy.txt): the penalty goes with the square of the density, down to a quarter — where the m-mers are
// long enough for the pangenome (r >= 10 below; profiles/r6o_lines_roomy_m.txt).
__host__ __device__ __forceinline__ uint32_t minimizer_length(uint32_t k, uint64_t expected_keys, uint64_t first_len = 0,
                                                              uint32_t wmax = MZ_WMAX, uint32_t ngenomes = 0, uint32_t cosched = 0,
                                                              double load = 0.375) {
    if (k < 20 || k > 32) return 0;
    // (more than 64 genomes: the split layout's lines hold 16 keys, a merged group fits more often — half the penalty:
    // 128 x 40 Mb 83.6 / 80.7 / 76.6 G k-mers/s at m = 15 / 16 / 17, 128 x 10 Mb 74.5 / 72.6 / 69.0)
    const double merged = ngenomes > 64 ? 25.0 : 50.0;
    double sparse = 1.0;  // what is left of the penalty in a sparser table, where the m-mers are long enough (r >= 10: see the loop)
    if (load > 0.0 && load < 0.375) {
        sparse = (load / 0.375) * (load / 0.375);
        if (sparse < 0.25) sparse = 0.25;
    }
    double leff = 1.5e8;
    if (first_len) {
        leff = (double)first_len;
        double v = (double)expected_keys / (2.3 * leff);  // variants per locus, in units of what 8 genomes at 1 % have
        if (v > 1.0) leff *= sqrt(v < 16.0 ? v : 16.0);
    } else if (expected_keys) {
        leff = (double)expected_keys / 2.3;
    }
    uint32_t m_lo = 15;  // (k=21, 100 Mb genomes: m = 14 84 G k-mers/s against 100 at m = 15, round 1)
    if (k - (wmax - 1) > m_lo) m_lo = k - (wmax - 1);  // (wmax: PG_TABLE_WMAX caps the window, pg_api.hip — repeat-rich genomes)
    if (expected_keys) {
        uint32_t mk = 15;
        while (mk < 27 && (1ull << (2 * mk)) < 4 * expected_keys) ++mk;
        if (mk - 2 > m_lo) m_lo = mk - 2;
    }
    const uint32_t m_hi = k - (MZ_WMIN - 1);
    if (m_lo > m_hi) m_lo = m_hi;
    const double wcost_co[9] = {0, 0, 0, 35.0, 19.5, 10.5, 3.5, 0.5, 0.0};   // percent, by window: co-scheduled launches
    const double wcost_one[9] = {0, 0, 0, 62.0, 42.0, 26.0, 12.0, 2.0, 0.0};  // one launch per genome (profiles/r5_m_sweep_pergenome.txt)
    const double *wcost = cosched == 1 ? wcost_one : wcost_co;
    uint32_t best = m_lo;
    double best_cost = 1e30;
    for (uint32_t m = m_lo; m <= m_hi; ++m) {
        const double r = (m >= 31 ? 4.6e18 : (double)(1ull << (2 * m))) / leff;
        // (the relief of a sparse table fades where groups merge in earnest: 27 x 135 Mb, r = 5 at m = 15, runs 178 G at m = 15
        // against 194 at m = 16 whatever the density, 8 x 300 Mb, r = 3.6, 197 against 206 — none of it below r = 5, all of it from 10 on)
        const double relief = r >= 10.0 ? sparse : r <= 5.0 ? 1.0 : 1.0 + (sparse - 1.0) * (r - 5.0) / 5.0;
        const double c = wcost[k - m + 1] + merged * relief / r;
        if (c < best_cost) {  // (ties: the wider window)
            best_cost = c;
            best = m;
        }
    }
    return best;
}

// ---- hashing ------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
// scrambled order of canonical m-mers (a bijection on 32 bits: ties only between equal m-mers)
__host__ __device__ __forceinline__ uint32_t mz_order(uint32_t c) { return c * 0x9E3779B1u; }

__device__ __forceinline__ uint32_t range32(uint32_t h, uint64_t n) { return __umulhi(h, (uint32_t)n); }

// Probe sequence of a group = home, home+step, home+2*step, ... (mod nlines): DOUBLE hashing on
// the group id (minimizer, or the k-mer itself in direct mode).  Every k-mer of a group walks
// the same sequence, so a group that spills stays together; and because spilled groups jump
// to unrelated lines, full lines do not pile up into long chains the way +1 probing does.
__device__ __forceinline__ uint32_t group_of_key(uint64_t key) {
    return (uint32_t)key ^ fmix32((uint32_t)(key >> 32) + 0x9E3779B9u);
}
// one multiply + xor-shift each (32-bit multiplies are quarter rate on CDNA): range32 takes the TOP
// bits of the hash, and the top bits of g * odd depend on every bit of g
__device__ __forceinline__ uint32_t mix1(uint32_t g, uint32_t mul) {
    const uint32_t h = g * mul;
    return h ^ (h >> 15);
}
// (home: no xor-shift behind the multiply — range32 reads the top bits of g * odd, which depend on every bit of g already, and
// the shift only stirs the low 17: two VALU instructions per batch of k_probe / k_insert_tile for nothing)
__device__ __forceinline__ uint32_t home_of_group(uint32_t g, uint64_t nlines) { return range32(g * 0x85ebca6bu, nlines); }
__device__ __forceinline__ uint32_t step_of_group(uint32_t g, uint64_t nlines) {
    return 1u + range32(mix1(g ^ 0x5bd1e995u, 0xc2b2ae35u), nlines - 1);
}
// (line, step < nlines <= 2^31, alloc_sub: the sum is exact in 32 bits; n - nlines wraps to a huge number exactly when
// n < nlines, so the smaller of the two is the answer — add, subtract, min instead of a 64-bit add, compare and select)
__device__ __forceinline__ uint32_t next_line(uint32_t line, uint32_t step, uint64_t nlines) {
    const uint32_t n = line + step;
    return min(n, n - (uint32_t)nlines);
}
// A minimizer group owns only the first GROUP_CHAIN lines of its sequence.  A key that finds them
// all full continues on a sequence of ITS OWN (double hashing on the key): repeat families put
// thousands of distinct k-mers behind one minimizer, and without this bound their chain — walked
// by every insert and every lookup of the family — grows to hundreds of lines.
// Line number `level` (0 = home) of a key's probe sequence, given the line before it:
#ifndef PG_GROUP_CHAIN
#define PG_GROUP_CHAIN 4
#endif
constexpr uint32_t GROUP_CHAIN = PG_GROUP_CHAIN;
static_assert(GROUP_CHAIN >= 1, "a group owns at least its home line");
__device__ __forceinline__ void key_sequence(uint64_t key, uint64_t nlines, uint32_t &home, uint32_t &step) {
    const uint32_t g2 = fmix32(group_of_key(key) ^ 0x7feb352du);
    home = home_of_group(g2, nlines);
    step = step_of_group(g2, nlines);
}
__device__ __forceinline__ void advance_line(uint64_t key, uint32_t level, uint64_t nlines, uint32_t &line,
                                             uint32_t &step) {
    if (level == GROUP_CHAIN) {  // first line of the key's own sequence
        key_sequence(key, nlines, line, step);
    } else {
        line = next_line(line, step, nlines);
    }
}

// ---- packed sequence ------------------------------------------------------
// base i of a contig lives in bits [2*(i%32), 2*(i%32)+1] of u64 word i/32 (little-endian in
// the word); the "not ACGT" plane has bit i%32 of u32 word i/32.
//
// Notation used everywhere:  X = LE window of a k-mer (first base in the low bits),
//   B = LE window of its reverse complement = pairreverse(~X) >> (64-2k).
//   value() = first base most significant (KMC order):
//   value(fwd) = ~B & kmask,  value(revcomp) = ~X & kmask,
//   canonical key = min of the two = ~max(X, B) & kmask.
__device__ __forceinline__ uint64_t pair_reverse64(uint64_t x) {
    // bit reversal, then the two bits of every base swapped back — per 32-bit half (the masks drop
    // the bits that would cross), one bit-field insert each
    const uint32_t lo = __brev((uint32_t)(x >> 32)), hi = __brev((uint32_t)x);
    const uint32_t lo2 = ((lo >> 1) & 0x55555555u) | ((lo << 1) & ~0x55555555u);
    const uint32_t hi2 = ((hi >> 1) & 0x55555555u) | ((hi << 1) & ~0x55555555u);
    return (uint64_t)lo2 | ((uint64_t)hi2 << 32);
}
__device__ __forceinline__ uint64_t kmer_mask(int k) { return (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1); }

__device__ __forceinline__ uint64_t revcomp_le(uint64_t X, int k) {
    return (pair_reverse64(~X) >> (64 - 2 * k)) & kmer_mask(k);
}
__device__ __forceinline__ uint64_t canonical_from_xb(uint64_t X, uint64_t B, int k) {
    return ~(X > B ? X : B) & kmer_mask(k);
}
__device__ __forceinline__ uint64_t canonical_from_le(uint64_t x, int k) {
    x &= kmer_mask(k);
    return canonical_from_xb(x, revcomp_le(x, k), k);
}

// hash of a key for the distinct-count sketch (splitmix64 finalizer: a bijection on 64 bits)
constexpr uint32_t SKETCH_BITS = 16;
__host__ __device__ __forceinline__ uint64_t sketch_hash(uint64_t x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// rank of a canonical m-mer (up to 30 bases = 60 bits) in the scrambled order: the bits above 32
// are folded in with rotates (no multiply), then a multiplicative scramble.  For m <= 16 the fold is
// the identity and the rank a bijection; longer m-mers may tie, which only merges two groups.
__host__ __device__ __forceinline__ uint32_t mmer_rank(uint64_t c) {
    const uint32_t hi = (uint32_t)(c >> 32);
    return mz_order((uint32_t)c ^ ((hi << 11) | (hi >> 21)) ^ ((hi << 23) | (hi >> 9)));
}

// minimizer of a k-mer given X and B (generic, runtime w).  Symmetric in X <-> B.
__device__ __forceinline__ uint32_t minimizer_from_xb(uint64_t X, uint64_t B, uint32_t m, uint32_t w) {
    const uint64_t mm = (m >= 32) ? ~0ull : ((1ull << (2 * m)) - 1);
    uint32_t best = ~0u;
    for (uint32_t i = 0; i < w; ++i) {
        const uint64_t a = (X >> (2 * i)) & mm;
        const uint64_t b = (B >> (2 * (w - 1 - i))) & mm;
        const uint32_t h = mmer_rank(a < b ? a : b);
        best = h < best ? h : best;
    }
    return best;
}

// group id of a canonical key (insert / single-lane lookup / rehash path)
__device__ __forceinline__ uint32_t group_of(const SubTable &st, uint64_t key) {
    if (st.m == 0) return group_of_key(key);
    const int k = (int)st.k;
    const uint64_t X = pair_reverse64(key) >> (64 - 2 * k);  // LE window of the canonical strand
    const uint64_t B = ~key & kmer_mask(k);                  // LE window of the other strand
    return minimizer_from_xb(X, B, st.m, st.k - st.m + 1);
}

// 64 bits (32 bases) starting at base p of a word array (works for LDS or global)
template <typename P>
__device__ __forceinline__ uint64_t extract_bases(P words, uint64_t p) {
    uint64_t w = p >> 5;
    uint32_t sh = (uint32_t)(p & 31) * 2;
    uint64_t lo = words[w];
    uint64_t hi = words[w + 1];
    return (lo >> sh) | ((hi << 1) << (63 - sh));  // branch-free also for sh == 0 (both reads always issue)
}

// the same out of a 32-bit view of the words: two funnel shifts (v_alignbit) instead of three 64-bit
// shifts; reads three consecutive dwords
__device__ __forceinline__ uint64_t extract_bases32(const uint32_t *words32, uint32_t p) {
    const uint32_t d = p >> 4, sh = (p & 15u) * 2u;
    const uint32_t d0 = words32[d], d1 = words32[d + 1], d2 = words32[d + 2];
    const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh);
    const uint32_t hi = __builtin_amdgcn_alignbit(d2, d1, sh);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// k "not ACGT" bits starting at base p
template <typename P>
__device__ __forceinline__ uint32_t extract_nmask(P words, uint64_t p, int k) {
    uint64_t w = p >> 5;
    uint32_t sh = (uint32_t)(p & 31);
    uint64_t v = (uint64_t)words[w] | ((uint64_t)words[w + 1] << 32);
    v >>= sh;
    uint32_t km = (k == 32) ? 0xFFFFFFFFu : ((1u << k) - 1);
    return (uint32_t)v & km;
}
// 64 "not ACGT" bits starting at base p (reads three u32 words)
template <typename P>
__device__ __forceinline__ uint64_t extract_nmask64(P words, uint64_t p) {
    uint64_t w = p >> 5;
    uint32_t sh = (uint32_t)(p & 31);
    uint64_t lo = (uint64_t)words[w] | ((uint64_t)words[w + 1] << 32);
    uint64_t hi = (uint64_t)words[w + 2];
    return (lo >> sh) | ((hi << 1) << (63 - sh));
}

// Single-lane lookup (GetCountersForRead kernel): mask word w of `key`, 0 if absent.
__device__ __forceinline__ bool lane_lookup(const SubTable &st, uint64_t key, uint32_t w, uint32_t &out) {
    const uint32_t grp = group_of(st, key);
    uint32_t b = home_of_group(grp, st.nbuckets);
    uint32_t step = step_of_group(grp, st.nbuckets);
    for (uint64_t probes = 0; probes < st.nbuckets + GROUP_CHAIN; ++probes) {
        bool empty_seen = false;
        for (uint32_t s = 0; s < st.slots; ++s) {
            const uint64_t cur = *key_ptr(st, b, s);
            if (cur == key) {
                out = *mask_ptr(st, b, s, w);
                return true;
            }
            empty_seen |= (cur == EMPTY_KEY);
        }
        if (empty_seen) break;
        advance_line(key, (uint32_t)min(probes + 1, (uint64_t)GROUP_CHAIN + 1), st.nbuckets, b, step);
    }
    out = 0;
    return false;
}

// Insert-or-find `key`, OR `bits` into mask word w.  Returns 0 = existed,
// 1 = newly claimed, -1 = gave up after max_probe lines (table must grow).
// COUNT: mask word w is an occurrence counter (bits is added) instead of a presence mask (OR-ed)
template <bool COUNT = false, bool ATOMIC_OR = false>
__device__ __forceinline__ int lane_insert_grp(const SubTable &st, uint64_t key, int w, uint32_t bits, uint32_t max_probe,
                                               uint32_t grp) {
    uint32_t b = home_of_group(grp, st.nbuckets);
    uint32_t step = step_of_group(grp, st.nbuckets);
    const uint32_t kstride = key_stride(st);  // bytes from one key of a line to the next
    for (uint32_t probes = 0; probes < max_probe; ++probes) {
        uint8_t *base = st.buckets + (uint64_t)b * line_bytes(st);
        for (uint32_t s0 = 0; s0 < st.slots; s0 += 8) {  // 8 slots at a time: all key loads in flight together
            uint8_t *grp8 = base + kstride * s0;
            uint64_t kk[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {  // (relaxed atomic loads: all eight in flight together; volatile ones are waited for one by one)
                kk[s] = __hip_atomic_load(reinterpret_cast<unsigned long long *>(grp8 + kstride * s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (s0 + s >= st.slots) kk[s] = TOMB_KEY;  // (inline layout: 5 or 6 keys, mask words behind them — neither a key nor empty)
            }
            int hit = -1, free_s = -1;
#pragma unroll
            for (int s = 7; s >= 0; --s) {
                hit = (kk[s] == key) ? s : hit;
                free_s = (kk[s] == EMPTY_KEY) ? s : free_s;  // first empty slot (lines fill front to back)
            }
            int claimed = 0;
            // slots before the first empty one hold other keys for good; from there on a slot may be
            // taken by a concurrent insert between our read and our CAS: walk on, one CAS per slot
            for (int s = (hit >= 0 ? hit : free_s); hit < 0 && s >= 0 && s < 8 && s0 + (uint32_t)s < st.slots; ++s) {
                unsigned long long *kp = reinterpret_cast<unsigned long long *>(grp8 + kstride * s);
                const unsigned long long cur = atomicCAS(kp, (unsigned long long)EMPTY_KEY, (unsigned long long)key);
                if (cur == EMPTY_KEY) {
                    hit = s;
                    claimed = 1;
                } else if (cur == key) {
                    hit = s;
                }
            }
            if (hit >= 0) {
                uint32_t *mp = mask_ptr(st, b, s0 + (uint32_t)hit, (uint32_t)w);
                if (COUNT) {
                    if (*mp < 0xFFFFFF00u) atomicAdd(mp, bits);  // saturates far above any -ci threshold
                } else if (claimed && !ATOMIC_OR) {
                    // a slot claimed just now has zero mask words (tables start zeroed, slots never revert), and whoever
                    // finds the key there meanwhile ORs in the same bits (see below): no read needed
                    *reinterpret_cast<volatile uint32_t *>(mp) = bits;
                } else {
                    // A plain store, not an atomic OR (7 % of the table build): every writer of this word during
                    // one launch ORs in the SAME bits — one genome per insert launch, one writer per key in
                    // re-hash / merge / import, launches of one table serialised on its stream — so two racing
                    // read-modify-writes store the same value and no other bit can be lost.
                    // (ATOMIC_OR: callers that cannot promise it — key arrays handed in through the ABI.)
                    const uint32_t cur = *mp;
                    if ((cur & bits) != bits) {
                        if (ATOMIC_OR) atomicOr(mp, bits);
                        else *reinterpret_cast<volatile uint32_t *>(mp) = cur | bits;
                    }
                }
                return claimed;
            }
        }
        advance_line(key, probes + 1, st.nbuckets, b, step);
    }
    return -1;
}
template <bool COUNT = false, bool ATOMIC_OR = false>
__device__ __forceinline__ int lane_insert(const SubTable &st, uint64_t key, int w, uint32_t bits, uint32_t max_probe) {
    return lane_insert_grp<COUNT, ATOMIC_OR>(st, key, w, bits, max_probe, group_of(st, key));
}

}  // namespace pg
