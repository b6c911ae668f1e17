// pg_kernels.h — launch wrappers of the gfx950 kernels (see pg_kernels.hip)
#pragma once
#include "pg_device.h"

namespace pg {

// Positions per wave-tile.  1024 since round 4: a wave spends a twelfth of a 512-position tile's time before its first
// batch (descriptor and sequence loads, one after the other) and a fifth in the overflow drain behind its last, both
// per TILE — tools/phase_timing.py — and with twice the positions the drain's batches are twice as dense: configs[1]
// 4.19 -> 4.01 ms, 27 x 160 Mb +6 %, the probe of 64 x 160 Mb -10 %.  2048 needs the queue cut to 128 entries to keep
// 32 waves per CU in LDS, and loses more to early drains than it saves (profiles/r4_ab_tile.txt).
#ifndef PG_PROBE_TILE
#define PG_PROBE_TILE 1024
#endif
#ifndef PG_PROBE_MAXRUN
#define PG_PROBE_MAXRUN 16
#endif
constexpr int PROBE_TILE = PG_PROBE_TILE;      // k-mer positions per wave-tile (k_probe) / per block (k_epilogue)
constexpr int PROBE_MAXRUN = PG_PROBE_MAXRUN;  // table lines staged in LDS per step of a 64-lane batch
#ifndef PG_PROBE_STAGED_LEVELS
#define PG_PROBE_STAGED_LEVELS 2
#endif
constexpr int PROBE_STAGED_LEVELS = PG_PROBE_STAGED_LEVELS;  // LDS-staged overflow levels; beyond: lanes chase inline
#ifndef PG_PROBE_QCAP
#define PG_PROBE_QCAP 192  // (with the 1024-position tile's sequence words: 5104 bytes of LDS per wave, 32 waves per CU)
#endif
constexpr int PROBE_QCAP = PG_PROBE_QCAP;      // per-tile LDS overflow queue (beyond: resolved inline)

// which row bytes a sub-table writes: low nb0 bytes of mask word 0 at column col0, low nb1 bytes
// of mask word 1 at col0+4; words: 1 = rows are a whole number of 32-bit words, so a full mask word
// goes out as one aligned store; 2 = two-byte rows, one aligned 16-bit store; 3 = whole rows of 3/5/6/7 bytes in two byte-aligned stores;
// 0 = byte stores
struct RowCols {
    uint32_t col0, nb0, nb1, words;
};

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
// bytes [off, off+len) of a FASTA text are sequence lines of record rec (len <= 4096, never
// crossing an absolute multiple of 4096)
struct TextChunk {
    uint64_t off;
    uint32_t len, rec;
};
hipError_t launch_text_pack(hipStream_t st, const uint8_t *d_text, const TextChunk *d_chunks, uint64_t nchunks,
                            const uint64_t *d_rec_chunk0, uint32_t nrec, uint32_t *d_counts, uint64_t *d_base,
                            uint64_t *d_rec_len, const SeqDesc *sd, uint64_t *seqw, uint32_t *nmw, uint32_t *has_n);
// header lines of a FASTA text on the device ('>' at offset 0 or behind a line feed), unordered: count[0] of them, the first cap in d_out
hipError_t launch_text_headers(hipStream_t st, const uint8_t *d_text, uint64_t nbytes, uint64_t *d_out, uint32_t cap, uint32_t *d_count);
hipError_t launch_seq_tailmask(hipStream_t st, const SeqDesc *sd, uint32_t n, uint64_t *seqw, uint32_t *nmw);
// packed contigs gathered into another seqset's planes: job i copies nwords words of both planes and the contig's flag
struct SeqCopy {
    const uint64_t *src_seqw;
    const uint32_t *src_nmw;
    const uint32_t *src_has_n;  // (this contig's flag)
    uint64_t src_off, dst_off, nwords;
};
hipError_t launch_seq_gather(hipStream_t st, const SeqCopy *jobs, uint32_t n, uint64_t max_words, uint64_t *seqw, uint32_t *nmw,
                             uint32_t *has_n);
hipError_t launch_pack(hipStream_t st, const void *d_ascii, uint64_t len, uint64_t *seqw, uint32_t *nmw,
                       uint64_t nwords, uint32_t *has_n);
hipError_t launch_sketch(hipStream_t st, int k, const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                         uint64_t nkmers, uint32_t *regs);
// ... of all contigs of a seqset in one launch: jobs[j] = (contig, chunk number) — positions [chunk, chunk + 1) x SKETCH_JOB
constexpr uint32_t SKETCH_JOB = 1u << 16;
hipError_t launch_sketch_set(hipStream_t st, int k, const SeqDesc *sd, const uint2 *jobs, uint32_t njobs, const uint64_t *seqw,
                             const uint32_t *nmw, const uint32_t *has_n, uint32_t *regs);
hipError_t launch_insert_seq(hipStream_t st, const SubTable &t, int w, uint32_t bits, int k,
                             const uint64_t *seqw, const uint32_t *nmw, const uint32_t *has_n,
                             uint64_t nkmers, unsigned long long *counters, uint32_t max_probe, int count_mode = 0);
// the wave-cooperative build (pg_anchor.hip: k_insert_tile); tables of 128-byte lines only (8 slots, or the split layout)
hipError_t launch_insert_tiles(hipStream_t st, const SubTable &t, int w, uint32_t bits, int count_mode, const uint64_t *seqw,
                               const uint32_t *nmw, const uint32_t *has_n, const SeqDesc *sd, const uint32_t *tile0,
                               uint32_t ncontigs, uint32_t ntiles, unsigned long long *counters, uint32_t max_probe);
// Each .hip file is one code object, loaded by the HIP runtime when the first of its kernels is launched — for the anchor
// kernels' file ≈ 5 ms, which used to land on the first genome's insert.  pg_ctx_create calls these instead.
hipError_t preload_table_kernels();
hipError_t preload_anchor_kernels();
hipError_t preload_deflate_kernels();
// first tile of every contig (+ the total) of a launch over all contigs of a seqset, computed on the device
hipError_t launch_tile0(hipStream_t st, const SeqDesc *sd, uint32_t n, uint32_t k, uint32_t tile, uint32_t *tile0);
hipError_t launch_count_spill(hipStream_t st, const SubTable &t, unsigned long long *counters);
hipError_t launch_merge_min(hipStream_t st, const SubTable &src, const SubTable &dst, int w, uint32_t bits,
                            uint32_t min_count, unsigned long long *counters, uint32_t max_probe);
hipError_t launch_insert_keys(hipStream_t st, const SubTable &t, int w, const uint64_t *keys,
                              const uint32_t *vals, uint64_t n, unsigned long long *counters,
                              uint32_t max_probe);
// records [first_record, first_record + nrec) of a KMC suffix file (resident at `rec`) into group word w
hipError_t launch_import_kmc(hipStream_t st, const SubTable &t, int w, const uint8_t *rec, uint64_t first_record, uint64_t nrec,
                             const uint64_t *lut, uint64_t nlut, uint32_t prefixes_per_bin, uint32_t suffix_bytes,
                             uint32_t counter_bytes, uint32_t min_count, uint32_t max_count, unsigned long long *counters,
                             uint32_t max_probe, uint32_t phase = 2, uint32_t dense_above = 0);
hipError_t launch_rehash(hipStream_t st, const SubTable &src, const SubTable &dst,
                         unsigned long long *counters, uint32_t max_probe, uint32_t ngenomes);
hipError_t launch_export(hipStream_t st, const SubTable &t, int w, uint64_t *keys, uint32_t *vals,
                         uint64_t cap, unsigned long long *count);
hipError_t launch_counters(hipStream_t st, const SubTable &t, int w, int k, const uint64_t *seqw,
                           const uint32_t *nmw, const uint32_t *has_n, uint64_t nkmers, uint32_t *out);
// Fused statistics (round 6): a k_probe launch that is handed these buffers ends every tile of a contig whose bins are at least
// a tile long (AnchorDesc::binlen >= PROBE_TILE: a tile then touches at most two bins) by reading its finished rows back —
// while they are still in the L2 / Infinity Cache, not from HBM — and leaves, per tile:
//   tile_hist[tile][hw]   hw = N + 1 words = 2 (N + 1) u16 counters: (bin of the row relative to the tile's first: 0 / 1) x (popcount 0..N)
//   tile_cs[tile][csw]    csw = 16 * ceil(nbytes / 4) words: word 16 w + e = rows of the tile holding genome 32 w + e (low half) and
//                         genome 32 w + e + 16 (high half)
// and the tile's 1-in-100 rows in out100 (NULL: another step, k_lowres copies them).  k_tile_reduce sums the tiles' counters into
// the bins and the per-contig column sums: the statistics pass's re-read of every row from HBM (k_epilogue*) is gone.  Contigs
// with shorter bins are left to that pass, launched over their tile ranges only.  Replaces the popcount / histogram / 1-in-100
// part of the reference's scatter loop (cpp/anchor.cpp:156-183) and the column sums of index.py:1051.
struct FuseArgs {
    uint8_t *out100;
    uint32_t *tile_hist;
    uint32_t *tile_cs;
    uint32_t ngenomes, hw, csw;
};
constexpr uint32_t fuse_hist_words(uint32_t ngenomes) { return ngenomes + 1u; }
constexpr uint32_t fuse_cs_words(uint32_t ngenomes) { return 16u * ((((ngenomes + 7u) / 8u) + 3u) / 4u); }
// rows the fused instantiations exist for: 2..8 bytes; 9..16 bytes (the inline and split layouts) only in a -DPG_FUSE_WIDE=1
// build — held to the 64 vector and 80 scalar registers of eight waves per SIMD their tile end spills, and the probe of 65 /
// 128 genomes takes 11.0 / 20.7 ms instead of 4.4 / 11.4 (profiles/r6f_ab_fuse.txt).  One-byte rows keep their bit-sliced pass.
#ifndef PG_FUSE_WIDE
#define PG_FUSE_WIDE 0
#endif
constexpr bool fuse_rows_ok(uint32_t nbytes) { return nbytes >= 2u && nbytes <= (PG_FUSE_WIDE ? 16u : 8u); }
hipError_t launch_anchor(hipStream_t st, const TableDesc &T, const uint64_t *seqw, const uint32_t *nmw,
                         const uint32_t *has_n, const SeqDesc *sd, const AnchorDesc *ad,
                         const uint32_t *tile_contig, const uint32_t *sched, uint32_t tile_base, uint32_t ntiles, uint8_t *out1,
                         uint64_t out1_bytes, uint32_t columns_width = 0, const FuseArgs *fuse = nullptr);
// the tiles' counters (FuseArgs) of tiles [0, ntiles) into bins / colsums (atomic adds: both zeroed by the caller)
hipError_t launch_tile_reduce(hipStream_t st, const FuseArgs &fo, const AnchorDesc *ad, const uint32_t *tile_contig, uint32_t ntiles,
                              uint32_t *bins, unsigned long long *colsums, uint32_t want_colsums);
// genome-sharded exchange over the tiles [tile_base, tile_base + ntiles) of a result (a contig range)
hipError_t launch_cols_extract(hipStream_t st, uint32_t ngenomes, const AnchorDesc *ad, const uint32_t *tile_contig,
                               uint32_t tile_base, uint32_t ntiles, const uint8_t *out1, uint32_t g0, uint32_t width, void *dst);
hipError_t launch_cols_merge(hipStream_t st, uint32_t ngenomes, const AnchorDesc *ad, const uint32_t *tile_contig,
                             uint32_t tile_base, uint32_t ntiles, uint8_t *out1, const void *src, uint32_t part0,
                             uint32_t nparts, uint64_t part_words, uint32_t per, uint32_t accumulate);
// every step-th row of bitmap.1 -> low-resolution bitmap (steps other than 100)
hipError_t launch_lowres(hipStream_t st, uint32_t ngenomes, const AnchorDesc *ad, const uint32_t *tile_contig,
                         uint32_t ntiles, const uint8_t *out1, uint8_t *outlow, uint32_t step);
// GPU-side BGZF compression of a payload (pg_deflate.hip): the payload is the concatenation of
// segments of device memory; segs[nseg] is a sentinel with lstart = total
struct PaySeg {
    uint64_t lstart, doff;
};
// (a thread of k_row_deflate takes DF_CHUNK_BYTES of a block; crc_tabs: 1024 words of CRC-32 slicing-by-four tables, then
// DF_CRC_LEVELS sets of 4 x 256: set j = the register after DF_CHUNK_BYTES * 2^j more zero bytes, by each of its four bytes)
constexpr uint32_t DF_CHUNK_BYTES = 68, DF_CRC_LEVELS = 10, DF_CRC_TAB_WORDS = 1024 + DF_CRC_LEVELS * 1024;
constexpr uint32_t DF_CODE_BYTES = 2048, DF_HIST_WORDS = 288, DF_SAMPLE_BLOCKS = 512;
// one Huffman code per file: symbol counts of up to DF_SAMPLE_BLOCKS blocks -> code + block header in `code` (DF_CODE_BYTES)
hipError_t launch_deflate_code(hipStream_t st, const uint8_t *base, const PaySeg *segs, uint32_t nseg, uint64_t total, uint32_t row,
                               uint32_t *hist, void *code);
hipError_t launch_row_deflate(hipStream_t st, const uint8_t *base, const PaySeg *segs, uint32_t nseg, uint64_t total,
                              uint64_t first_block, uint32_t nblocks, uint32_t row, const uint32_t *crc_tabs, const void *code,
                              uint8_t *slots, uint32_t *sizes, uint32_t force_stored, uint32_t *offs, uint8_t *packed);
hipError_t launch_window_stats(hipStream_t st, uint32_t ngenomes, const uint8_t *rows, uint64_t nrows, uint32_t nwin,
                               uint32_t pieces, const uint64_t *starts, const uint64_t *ends, unsigned long long *hist,
                               unsigned long long *cs);
hipError_t launch_rows_epilogue(hipStream_t st, uint32_t ngenomes, const AnchorDesc *ad, const uint32_t *tile_contig,
                                uint32_t ntiles, const uint8_t *out1, uint8_t *out100, uint32_t *bins,
                                unsigned long long *colsums, uint32_t flags, const uint2 *d_ranges = nullptr,
                                uint32_t nranges = 0, uint32_t range_tiles = 0);

}  // namespace pg
