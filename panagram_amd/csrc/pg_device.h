// pg_device.h — device-side building blocks shared by the gfx950 kernels.
//
// Table layout (one sub-table covers W = 1 or 2 consecutive 32-genome groups):
//   bucket = 64 bytes, 64-byte aligned = one HBM fetch per probe
//          = 4 slots x { u64 key ; u32 mask0 ; u32 mask1 }   (mask1 unused when W == 1)
//   so that the four lanes of a quad each hold ONE complete slot of the bucket after a
//   single 16-byte load: the match is lane-local, no cross-lane traffic for the masks.
//   EMPTY key = ~0 (never a canonical k-mer for k <= 32: the all-T k-mer's
//   reverse complement is 0).  Keys only ever go EMPTY -> key, masks only gain
//   bits, so inserts need one 64-bit CAS + one 32-bit OR and no locks.
//   Collision policy: bucketed linear probing — a lookup moves to the next
//   bucket only when the key is absent AND the bucket has no EMPTY slot.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pg {

constexpr uint64_t EMPTY_KEY = ~0ull;
constexpr int BUCKET_BYTES = 64;
constexpr int MAX_SUB = 8;  // sub-tables per pan table => up to 512 genomes

struct SubTable {
    uint8_t *buckets;
    uint64_t nbuckets;
    uint32_t W;      // mask words per slot
    uint32_t word0;  // first 32-genome group covered
};

struct TableDesc {
    SubTable sub[MAX_SUB];
    uint32_t nsub;
    uint32_t ndbs;  // total 32-genome groups
    uint32_t k;
    uint32_t ngenomes;
};

constexpr int SLOTS = 4;
__host__ __device__ __forceinline__ int slots_per_bucket(uint32_t) { return SLOTS; }
__host__ __device__ __forceinline__ uint32_t key_off(uint32_t, int s) { return 16u * s; }
__host__ __device__ __forceinline__ uint32_t mask_off(uint32_t, int s, int w) { return 16u * s + 8u + 4u * w; }

// murmur3 finaliser: a bijection on u64, so distinct keys never alias before the
// range reduction.
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

__device__ __forceinline__ uint64_t home_bucket(uint64_t key, uint64_t nbuckets) {
    return __umul64hi(mix64(key), nbuckets);
}

// ---- packed sequence ------------------------------------------------------
// base i of a contig lives in bits [2*(i%32), 2*(i%32)+1] of u64 word i/32
// (little-endian in the word); the "not ACGT" plane has bit i%32 of u32 word i/32.
//
// For a window x = bases [p, p+k) extracted little-endian (first base in the low
// bits):   value(revcomp) = ~x & kmask          (complement, order already reversed)
//          value(fwd)     = pairreverse(x) >> (64-2k)
// with value() = first base most significant (KMC order).
__device__ __forceinline__ uint64_t pair_reverse64(uint64_t x) {
    uint64_t r = __brevll(x);
    return ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
}

__device__ __forceinline__ uint64_t canonical_from_le(uint64_t x, int k) {
    const uint64_t kmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    x &= kmask;
    uint64_t rc = (~x) & kmask;
    uint64_t fw = pair_reverse64(x) >> (64 - 2 * k);
    return fw < rc ? fw : rc;
}

// 64 bits (32 bases) starting at base p of a word array (works for LDS or global)
template <typename P>
__device__ __forceinline__ uint64_t extract_bases(P words, uint64_t p) {
    uint64_t w = p >> 5;
    uint32_t sh = (uint32_t)(p & 31) * 2;
    uint64_t lo = words[w];
    uint64_t hi = words[w + 1];
    return sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
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

// ---- DPP quad permutes: lanes 4q..4q+3 cooperate on one bucket ---------------
template <int CTRL>
__device__ __forceinline__ uint32_t quad_perm(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
constexpr int QP_XOR1 = 0xB1;   // [1,0,3,2]
constexpr int QP_XOR2 = 0x4E;   // [2,3,0,1]
constexpr int QP_BC0 = 0x00, QP_BC1 = 0x55, QP_BC2 = 0xAA, QP_BC3 = 0xFF;

__device__ __forceinline__ uint32_t quad_or(uint32_t v) {
    v |= quad_perm<QP_XOR1>(v);
    v |= quad_perm<QP_XOR2>(v);
    return v;
}

// Cooperative match of one 64-byte bucket held one 16-byte slot per lane of a quad:
// v = {key.lo, key.hi, mask0, mask1}.  Returns found; masks are quad-uniform.
__device__ __forceinline__ bool quad_match(const uint4 v, uint64_t key, uint32_t &m0, uint32_t &m1) {
    const uint64_t kk = (uint64_t)v.x | ((uint64_t)v.y << 32);
    const bool hit = (kk == key);
    m0 = quad_or(hit ? v.z : 0u);
    m1 = quad_or(hit ? v.w : 0u);
    return quad_or(hit ? 1u : 0u) != 0;
}
// no EMPTY slot among the quad's four (only needed on the rare not-found path)
__device__ __forceinline__ bool quad_full(const uint4 v) {
    const uint64_t kk = (uint64_t)v.x | ((uint64_t)v.y << 32);
    return quad_or(kk == EMPTY_KEY ? 1u : 0u) == 0;
}

// Single-lane lookup (used by the GetCountersForRead kernel, export and rehash).
__device__ __forceinline__ bool lane_lookup(const SubTable &st, uint64_t key, uint32_t &m0, uint32_t &m1) {
    uint64_t b = home_bucket(key, st.nbuckets);
    const int ns = slots_per_bucket(st.W);
    for (uint64_t probes = 0; probes < st.nbuckets; ++probes) {
        const uint8_t *base = st.buckets + b * BUCKET_BYTES;
        bool empty_seen = false;
        for (int s = 0; s < ns; ++s) {
            uint64_t cur = *reinterpret_cast<const uint64_t *>(base + key_off(st.W, s));
            if (cur == key) {
                m0 = *reinterpret_cast<const uint32_t *>(base + mask_off(st.W, s, 0));
                m1 = st.W == 2 ? *reinterpret_cast<const uint32_t *>(base + mask_off(st.W, s, 1)) : 0u;
                return true;
            }
            empty_seen |= (cur == EMPTY_KEY);
        }
        if (empty_seen) break;
        b = (b + 1 == st.nbuckets) ? 0 : b + 1;
    }
    m0 = m1 = 0;
    return false;
}

// Insert-or-find `key`, OR `bits` into mask word w.  Returns 0 = existed,
// 1 = newly claimed, -1 = gave up after max_probe buckets (table must grow).
__device__ __forceinline__ int lane_insert(const SubTable &st, uint64_t key, int w, uint32_t bits,
                                           uint32_t max_probe) {
    uint64_t b = home_bucket(key, st.nbuckets);
    const int ns = slots_per_bucket(st.W);
    for (uint32_t probes = 0; probes < max_probe; ++probes) {
        uint8_t *base = st.buckets + b * BUCKET_BYTES;
        for (int s = 0; s < ns; ++s) {
            unsigned long long *kp = reinterpret_cast<unsigned long long *>(base + key_off(st.W, s));
            unsigned long long cur = *kp;  // a stale EMPTY only costs a failed CAS
            int claimed = 0;
            if (cur == EMPTY_KEY) {
                cur = atomicCAS(kp, (unsigned long long)EMPTY_KEY, (unsigned long long)key);
                if (cur == EMPTY_KEY) {
                    cur = key;
                    claimed = 1;
                }
            }
            if (cur == key) {
                uint32_t *mp = reinterpret_cast<uint32_t *>(base + mask_off(st.W, s, w));
                if ((*mp & bits) != bits) atomicOr(mp, bits);
                return claimed;
            }
        }
        b = (b + 1 == st.nbuckets) ? 0 : b + 1;
    }
    return -1;
}

}  // namespace pg
