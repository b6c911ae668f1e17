// pg_kernels.h — launch wrappers of the gfx950 kernels (see pg_kernels.hip)
#pragma once
#include "pg_device.h"

namespace pg {

#ifndef PG_ANCHOR_TILE
#define PG_ANCHOR_TILE 1024
#endif
#ifndef PG_ANCHOR_UNROLL
#define PG_ANCHOR_UNROLL 4
#endif
constexpr int ANCHOR_TILE = PG_ANCHOR_TILE;      // k-mer positions per workgroup
constexpr int ANCHOR_UNROLL = PG_ANCHOR_UNROLL;  // independent bucket gathers in flight per lane
#ifndef PG_ANCHOR_WG
#define PG_ANCHOR_WG 256
#endif
constexpr int ANCHOR_WG = PG_ANCHOR_WG;          // threads per workgroup (64 = one wave: barriers vanish)
#ifndef PG_ANCHOR_LINES
#define PG_ANCHOR_LINES (PG_ANCHOR_TILE * 5 / 16)
#endif
constexpr int ANCHOR_LINES = PG_ANCHOR_LINES;    // LDS line buffer (128-B table lines) per workgroup
constexpr int ANCHOR_MAX_ROUNDS = 24;            // queued overflow rounds before chasing a chain inline

// one packed contig of a seqset (offsets in 32-base words, shared by both planes)
struct SeqDesc {
    uint64_t seq_off;
    uint64_t nwords;
    uint64_t len;
};

// per-contig output geometry of one anchor run
struct AnchorDesc {
    uint64_t out_off;     // byte offset into the bitmap.1 buffer (16-byte aligned)
    uint64_t out100_off;  // byte offset into the bitmap.100 buffer
    uint64_t bin_off;     // first row of this contig in the bins buffer
    uint32_t nkmers;
    uint32_t binlen;
    uint32_t tile0;  // index of the contig's first tile in the launch
    uint32_t nbins;
};

hipError_t launch_table_init(hipStream_t st, const SubTable &t);
hipError_t launch_pack(hipStream_t st, const void *d_ascii, uint64_t len, uint64_t *seqw, uint32_t *nmw,
                       uint64_t nwords, uint32_t *has_n);
hipError_t launch_insert_seq(hipStream_t st, const SubTable &t, int w, uint32_t bits, int k,
                             const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                             uint64_t nkmers, unsigned long long *counters, uint32_t max_probe);
hipError_t launch_insert_keys(hipStream_t st, const SubTable &t, int w, const uint64_t *keys,
                              const uint32_t *vals, uint64_t n, unsigned long long *counters,
                              uint32_t max_probe);
hipError_t launch_rehash(hipStream_t st, const SubTable &src, const SubTable &dst,
                         unsigned long long *counters, uint32_t max_probe);
hipError_t launch_export(hipStream_t st, const SubTable &t, int w, uint64_t *keys, uint32_t *vals,
                         uint64_t cap, unsigned long long *count);
hipError_t launch_counters(hipStream_t st, const SubTable &t, int w, int k, const uint64_t *seqw,
                           const uint32_t *nmw, const uint32_t *has_n, uint64_t nkmers, uint32_t *out);
size_t anchor_lds_bytes(uint32_t ngenomes);
hipError_t launch_anchor(hipStream_t st, const TableDesc &T, const uint64_t *seqw, const uint32_t *nmw,
                         const uint32_t *has_n, const SeqDesc *sd, const AnchorDesc *ad,
                         const uint32_t *tile_contig, uint32_t ntiles, uint8_t *out1, uint8_t *out100,
                         uint32_t *bins, unsigned long long *colsums, uint32_t flags);

hipError_t launch_rows_epilogue(hipStream_t st, uint32_t ngenomes, const AnchorDesc *ad, const uint32_t *tile_contig,
                                uint32_t ntiles, const uint8_t *out1, uint8_t *out100, uint32_t *bins,
                                unsigned long long *colsums, uint32_t flags);

}  // namespace pg
