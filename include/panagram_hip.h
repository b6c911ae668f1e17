/*
 * panagram_hip.h — C-ABI of libpanagram_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the `panagram index` anchor hot path.  Each entry point
 * names the reference interface it replaces (paths relative to the reference
 * repo kjenike/panagram).  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - every function returns 0 on success, <0 (PG_E_*) on failure;
 *     pg_last_error() returns a thread-local, human-readable message.
 *   - handles are opaque; the library owns all device memory behind them.
 *   - host output buffers are allocated by the caller.
 *   - all work of a context is enqueued on ONE HIP stream (pg_ctx_set_stream), except that
 *     pg_anchor_run puts its statistics pass on a library-owned side stream, event-ordered
 *     behind the probe kernels so that it overlaps the next result's probes; every function
 *     that reads a result makes the main stream wait for that pass first.
 *     Functions documented "async" do not synchronise.
 *   - a context is not re-entrant; different contexts are independent.
 *
 * Bit/byte conventions (identical to the reference):
 *   nbytes = ceil(ngenomes/8) (cpp/anchor.cpp:34, index.py:484); genome g of a
 *   position = byte g/8, bit g%8 of its row (index.py:824-825); a row is the
 *   concatenation over 32-genome groups ("bitvec DBs", index.py:391-401) of the
 *   low n bytes of that group's u32 mask (cpp/anchor.cpp:139-164).
 */
#ifndef PANAGRAM_HIP_H
#define PANAGRAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_OK 0
#define PG_E_INVALID (-1)   /* bad argument */
#define PG_E_HIP (-2)       /* HIP runtime error (no GPU, OOM, launch failure) */
#define PG_E_FORMAT (-3)    /* ill-formed KMC database */
#define PG_E_CAPACITY (-4)  /* table cannot grow further */
#define PG_E_IO (-5)        /* file I/O (BGZF writer) */

typedef struct pg_ctx pg_ctx;
typedef struct pg_table pg_table;
typedef struct pg_seqset pg_seqset;
typedef struct pg_result pg_result;
typedef struct pg_bgzf pg_bgzf;

/* thread-local message of the last failing call on this thread */
const char *pg_last_error(void);
/* library version string, e.g. "panagram_hip 0.2 gfx950" */
const char *pg_version(void);
/* k-mer positions per tile: the unit of a launch, of pg_result_coschedule's pieces and of the bit-column blocks
 * (pg_tile_positions() / 8 bytes per tile and genome — one u64 per 64 positions; 128 bytes with today's tiles of 1024.
 * Size buffers with pg_result_columns_bytes / pg_result_columns_bytes_range, never with a constant) */
uint32_t pg_tile_positions(void);

/* ---- context ---------------------------------------------------------- */
/* One context per GPU / per process rank.  Fails with PG_E_HIP when no
 * gfx950-class device is visible: there is NO CPU fallback in this library. */
int pg_ctx_create(int device_id, pg_ctx **out);
/* *_destroy calls may come in any order (garbage-collected callers cannot promise one): a handle
 * with live dependants — a context with tables / seqsets, a table or seqset with results — is
 * retired at once for the caller but freed only when its last dependant is destroyed.  Destroy
 * each handle exactly once and do not use it afterwards. */
int pg_ctx_destroy(pg_ctx *ctx);
/* give back device memory the context keeps for reuse (row buffers of destroyed results: freeing and
 * re-allocating tens of GB per batch of anchors costs seconds) */
int pg_ctx_trim(pg_ctx *ctx);
/* device memory free for new tables / results (cached row buffers counted as free), and the device's total */
int pg_ctx_mem_info(pg_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes);
/* page-locked host memory for the buffers a caller hands to the library (ownership: "caller allocates all host
 * buffers, ideally pinned"): a FASTA image read straight into one goes up by DMA instead of through the runtime's staging
 * copy of pageable memory, and re-used it costs no page faults per file (Index.load_inputs: index.py:922-930's reading of
 * the sample files).  Free with pg_host_free, any time before the context is destroyed. */
int pg_host_alloc(pg_ctx *ctx, uint64_t nbytes, void **out);
int pg_host_free(pg_ctx *ctx, void *p);
/* adopt an external hipStream_t (e.g. torch.cuda.current_stream().cuda_stream; NULL is
 * HIP's legacy default stream, which is what torch's default stream is); use_own != 0
 * restores the context's own non-blocking stream instead. */
int pg_ctx_set_stream(pg_ctx *ctx, void *hip_stream, int use_own);
int pg_ctx_synchronize(pg_ctx *ctx);
/* plain device buffers (zeroed; memset is async on the context's stream) for callers without an allocator of their
 * own: the genome-sharded pipeline's exchange buffers in a single-process run */
int pg_device_alloc(pg_ctx *ctx, uint64_t bytes, void **out);
int pg_device_memset(pg_ctx *ctx, void *ptr, int value, uint64_t bytes);
int pg_device_free(pg_ctx *ctx, void *ptr);

/* ---- pan-kmer table: replaces the merged KMC "bitvec" databases --------
 * Reference: KMCdb::KMCdb opens root/kmc/bitvec{i} with CKMCFile::OpenForRA
 * (cpp/anchor.cpp:21-35); Genome._load_kmc (index.py:847-863).
 * ONE GPU-resident open-addressed table whatever the genome count, in 128-byte lines: up to 64 genomes 8 slots
 * {u64 key, u32 mask, u32 mask}; 65..96 genomes 6 bare keys followed by their 12-byte mask blocks (inline layout);
 * more: 16 bare keys, the mask words in a second array (split layout).  Home line = hash of the k-mer's minimizer;
 * key = canonical k-mer (2k-bit integer, first base most significant), value = the group's u32
 * one-hot-OR mask(s).  k in 1..32.  One writer at a time: the calls that add keys (insert_*, load_kmc1,
 * rehash) serialise on a per-table lock; lookups (pg_anchor_run ...) must not overlap them. */
int pg_table_create(pg_ctx *ctx, int k, int ngenomes, uint64_t expected_keys, pg_table **out);
/* the same with the table's density chosen by the caller: keys_per_line (1 .. 6.4) keys per 128-byte line of 8 slots instead
 * of the library's 3 (a load of keys_per_line / 8 in the layouts of more than 64 genomes, whose lines hold 6 or 16 keys), for a
 * known expected_keys (> 0).  SPARSER (1.5): what Index.build_table asks for where HBM is plentiful — fewer keys outside their
 * home lines, k_probe 3-8 % faster for twice the table bytes (profiles/r6k2_density_sweep.txt, r6q_wide_density.txt).  DENSER:
 * the genome-sharded mode's block
 * tables (BASELINE configs[4]) — a denser table holds the union of TWO genomes' k-mers in one GPU's HBM, the
 * job takes half the passes over the anchors' positions, and a pass against 3.4-4.5 keys per line is 5-25 % slower, not 100 %
 * (profiles/r6g_config5_blocks.txt).  The table is not grown back to 3 keys per line while it fills.  The reference has one
 * density: KMC's sorted suffix arrays (call site cpp/anchor.cpp:28-31). */
int pg_table_create_dense(pg_ctx *ctx, int k, int ngenomes, uint64_t expected_keys, double keys_per_line, pg_table **out);
/* device bytes pg_table_create_dense would allocate (host only; the planner's arithmetic) */
int pg_table_bytes_for_dense(int k, int ngenomes, uint64_t expected_keys, double keys_per_line, uint64_t *bytes);
/* device bytes pg_table_create would allocate for that many keys (host only): lets the caller decide between one
 * replicated table and the genome-sharded mode before allocating anything */
int pg_table_bytes_for(int k, int ngenomes, uint64_t expected_keys, uint64_t *bytes);
int pg_table_destroy(pg_table *tbl);
/* empty the table, keeping its allocation and geometry (async): the genome-sharded mode builds its genome blocks'
 * tables one after the other in the same memory */
int pg_table_clear(pg_table *tbl);

/* k-mer set construction from sequence, replacing `kmc -ci1 -fm` +
 * `kmc_tools transform set_counts` + `kmc_tools complex -ocsum`
 * (workflow/Snakefile:54-110, index.py:407-426): every canonical k-mer of every
 * contig of `seqs` gets bit genome_idx%32 set in group genome_idx/32.
 * Grows the table as needed.  Synchronises. */
int pg_table_insert_seqset(pg_table *tbl, int genome_idx, const pg_seqset *seqs);
/* bits only: bit genome_idx goes into the k-mers of `seqs` that the table ALREADY holds, no key is added.  Building the
 * table from the genomes a process anchors (pg_table_insert_seqset) and updating it with all the others gives every
 * look-up of the anchor step (cpp/anchor.cpp:148: the k-mers of the anchor FASTA itself) the answer the table of all
 * genomes gives, in a fraction of its memory.  Synchronises. */
int pg_table_update_seqset(pg_table *tbl, int genome_idx, const pg_seqset *seqs);
/* same with KMC's -ci<min_count> cut-off: only canonical k-mers occurring at least min_count times
 * in the seqset enter the table (the reference counts FASTQ samples with -ci2,
 * workflow/Snakefile:88-89; min_count <= 1 is pg_table_insert_seqset) */
int pg_table_insert_seqset_min(pg_table *tbl, int genome_idx, const pg_seqset *seqs, uint32_t min_count);

/* bulk insert of (canonical key, u32 counter) pairs into group db_idx; counters
 * of equal keys are OR-ed.  Host pointers.  Synchronises. */
int pg_table_insert_keys(pg_table *tbl, int db_idx, const uint64_t *keys,
                         const uint32_t *counters, uint64_t n);

/* load a real KMC database as group db_idx: the images of X.kmc_pre / X.kmc_suf (host memory; a read-only
 * memory map of the files is fine, they are only read once, chunk by chunk).  Both layouts
 * CKMCFile::OpenForRA accepts (cpp/anchor.cpp:29, index.py:859-860): KMC1 (kmc_version 0: kmc_tools output,
 * SURVEY.md Appendix A) and KMC2 (kmc_version 0x200: what `kmc` itself writes, workflow/Snakefile:101-104 —
 * one prefix LUT per signature bin).  The records are turned into table inserts on the GPU (k_import_kmc);
 * counters outside the header's [min_count,max_count] read as 0 as in KMC.  Returns PG_E_FORMAT on an
 * ill-formed database instead of silently yielding zeros like the reference.  pg_table_load_kmc1 is the same
 * function under its first name. */
int pg_table_load_kmc(pg_table *tbl, int db_idx, const void *pre, size_t pre_len,
                      const void *suf, size_t suf_len);
int pg_table_load_kmc1(pg_table *tbl, int db_idx, const void *pre, size_t pre_len,
                       const void *suf, size_t suf_len);
/* k-mer length of a KMC database from its .kmc_pre image (host only) */
int pg_kmc_kmer_length(const void *pre, size_t pre_len, uint32_t *k);

/* statistics: distinct keys, slot capacity, bucket count, bytes, summed over sub-tables */
int pg_table_stats(pg_table *tbl, uint64_t *nkeys, uint64_t *nslots, uint64_t *nbuckets,
                   uint64_t *bytes);
/* Distinct canonical k-mers of a set of inputs before a table exists (HyperLogLog, 2^16 registers,
 * standard error 0.4 %): sizes pg_table_create's expected_keys, so the table is neither re-hashed
 * while it grows nor held twice in HBM.  Replaces nothing in the reference (KMC is only given a memory cap:
 * panagram/workflow/Snakefile:101 -m{config[kmc][memory]}); pg_sketch_registers exposes the
 * 65536 register values for parity with the CPU restatement. */
typedef struct pg_sketch pg_sketch;
int pg_sketch_create(pg_ctx *ctx, int k, pg_sketch **out);
int pg_sketch_add_seqset(pg_sketch *sk, const pg_seqset *seqs);
int pg_sketch_estimate(pg_sketch *sk, uint64_t *distinct);
int pg_sketch_registers(pg_sketch *sk, uint8_t *out65536);
/* sketches merge by register-wise maximum: the estimate for 65536 register values held by the caller (host only,
 * no device needed) — the distinct k-mers of any union of inputs whose registers were kept; and a sketch emptied
 * for the next input */
int pg_sketch_estimate_registers(const uint8_t *regs65536, uint64_t *distinct);
int pg_sketch_reset(pg_sketch *sk);
int pg_sketch_destroy(pg_sketch *sk);

/* re-hash into the smallest table whose mean occupancy is <= keys_per_bucket keys per 128 bytes;
 * also settles the minimizer length for the keys actually present */
int pg_table_rehash(pg_table *tbl, double keys_per_bucket);
/* as of the last pg_table_rehash: fraction of keys outside their home line, slots per line */
int pg_table_spill(const pg_table *tbl, double *fraction, uint32_t *slots);
/* the same fraction measured NOW, on the table as it stands (one pass over its lines): a table built in place and never
 * re-hashed — what Index.run() anchors against — has no "last re-hash".  Diagnostic; no reference counterpart. */
int pg_table_measure_spill(pg_table *tbl, double *fraction);
/* export group db_idx as (key, counter) pairs, unsorted; *n receives the count
 * (call with keys==NULL to query).  Lets the caller write a KMC1 database the
 * reference can open. */
int pg_table_export(pg_table *tbl, int db_idx, uint64_t *keys, uint32_t *counters,
                    uint64_t cap, uint64_t *n);
int pg_table_k(const pg_table *tbl);
int pg_table_ngenomes(const pg_table *tbl);
/* tuning knob with no reference counterpart: the table places a k-mer by its minimizer
 * (smallest canonical m-mer); m is picked from k, the key count and the genome length
 * (at creation from expected_keys; again from the length of the first sequence set inserted into the
 * empty table, and at pg_table_rehash; the environment variable PG_TABLE_WMAX=3..8 caps the window k-m+1 the
 * library chooses — 4 suits genomes dominated by one young high-copy repeat family).  set_minimizer pins m for
 * an EMPTY table: 0 = hash the k-mer itself, else k >= 20 and 3 <= k-m+1 <= 8. */
int pg_table_minimizer(const pg_table *tbl);
int pg_table_set_minimizer(pg_table *tbl, int m);
/* the minimizer length the library would choose (pure host arithmetic, no device needed): k, the expected key count
 * (0: unknown), the k-mer positions of the first sequence set (0: unknown), the widest window allowed (3..8; 0: the
 * PG_TABLE_WMAX / default cap), the genome count (0: unknown; more than 64 = the split layout).  What it encodes — window cost against merged minimizer groups — is measured in
 * profiles/r4b_m_sweep.txt; no reference counterpart (KMC's own signature length is fixed at 9, kmc_file.h). */
int pg_minimizer_length(int k, uint64_t expected_keys, uint64_t first_len, int wmax, int ngenomes);
/* How the table will be PROBED decides the window too: `coscheduled` = the number of anchor genomes one launch anchors side
 * by side (pg_result_coschedule*; 0: not known, taken as several — what pg_minimizer_length assumes; 1: one genome per
 * launch, no partner: a single-anchor `panagram index` (index.py:1012, one Snakemake job per anchor), `run_anchor` with one
 * FASTA (cpp/anchor.cpp:217-223 with argc == 5), a py_kmc_api-style GetCountersForRead caller (index.py:934-935)).  Without a
 * partner every table line comes from HBM and the launch's time follows the lines per position, which a wider window
 * lowers (profiles/r5_m_sweep_pergenome.txt).  pg_table_set_coscheduled tells an EMPTY table (before
 * the first insert / load); pg_minimizer_length_for is the rule itself. */
int pg_table_set_coscheduled(pg_table *tbl, int anchors);
int pg_minimizer_length_for(int k, uint64_t expected_keys, uint64_t first_len, int wmax, int ngenomes, int coscheduled);
/* ... and the table's DENSITY (round 6): in a table created sparser than the library's 3 keys per line (pg_table_create_dense with
 * keys_per_line < 3) merged minimizer groups cost less — a group that outgrows its home line finds the next lines empty more often —
 * and the rule may take the wider window: configs[1] m = 15 at 1.25 keys per line, 16 at 3 (profiles/r6n_m_sweep_roomy.txt,
 * r6p_lines_roomy_m.txt).  A table applies the rule with its own density; this is the rule itself (host arithmetic). */
int pg_minimizer_length_dense(int k, uint64_t expected_keys, uint64_t first_len, int wmax, int ngenomes, int coscheduled,
                              double keys_per_line);

/* ---- sequences: 2-bit packed contigs resident in HBM -------------------
 * Reference: the FASTA record strings handed to write_bits / _write_bitmap
 * (cpp/anchor.cpp:77-100, index.py:1044-1046).  A seqset holds the contigs of
 * one FASTA, each packed 2 bits/base (A=0 C=1 G=2 T=3, case-insensitive) plus a
 * 1 bit/base "not ACGT" plane; packing runs on the GPU. */
int pg_seqset_create(pg_ctx *ctx, uint32_t ncontigs, const uint64_t *lens, pg_seqset **out);
int pg_seqset_destroy(pg_seqset *s);
/* upload + pack contig `idx` from host ASCII (async w.r.t. the host only after
 * the internal staging copy; synchronises the stream before returning) */
int pg_seqset_load_host(pg_seqset *s, uint32_t idx, const char *ascii, uint64_t len);
/* pack contig `idx` from ASCII already in device memory (16-byte aligned); async */
int pg_seqset_load_dev(pg_seqset *s, uint32_t idx, const void *d_ascii, uint64_t len);
uint64_t pg_seqset_total_kmers(const pg_seqset *s, int k);
/* FASTA text (a whole file image in host memory) -> seqset: replaces the line-based parse of
 * KMCdb::anchor_fasta (cpp/anchor.cpp:77-100) and Genome.iter_fasta (index.py:922-930).  A line
 * starting with '>' opens a record whose id is the header up to the first white space; the other
 * lines are joined with all white space removed (the Python path's Bio.SeqIO behaviour; the C++
 * path differs only in keeping '\r').  The host only locates the header lines; the GPU strips the
 * line breaks and packs.  Text before the first header is ignored.  Synchronises. */
int pg_seqset_from_fasta(pg_ctx *ctx, const void *text, uint64_t nbytes, pg_seqset **out);
/* one seqset holding the contigs of all `sets`, in order (packed planes copied device to device):
 * lets ONE result anchor every anchor genome of a pangenome, see pg_result_coschedule */
int pg_seqset_concat(pg_ctx *ctx, const pg_seqset *const *sets, uint32_t nsets, pg_seqset **out);
/* the same from contig ranges: contigs first_contig[i] .. first_contig[i]+ncontigs[i]-1 of sets[i], in order (a set may
 * appear several times): the genome-sharded pipeline lays the anchors' contigs out chunk group by chunk group, so
 * that one launch covers the homologous pieces of every anchor */
int pg_seqset_concat_ranges(pg_ctx *ctx, const pg_seqset *const *sets, const uint32_t *first_contig,
                            const uint32_t *ncontigs, uint32_t nsets, pg_seqset **out);
/* a seqset of n PIECES of src's contigs: piece i = bases [start[i], start[i] + len[i]) of contig contig[i] (packed planes
 * copied device to device; start[i] must be a multiple of 32).  The contig-sharded multi-GPU mode cuts long chromosomes
 * into pieces with a k-1 base overlap, so that a piece's k-mer positions are rows [start, start + len - k + 1) of the
 * whole contig's bitmap (cpp/anchor.cpp:127 cuts its chunks the same way); record ids become "<id>:<start>" */
int pg_seqset_slice(pg_ctx *ctx, const pg_seqset *src, uint32_t n, const uint32_t *contig, const uint64_t *start,
                    const uint64_t *len, pg_seqset **out);
uint32_t pg_seqset_ncontigs(const pg_seqset *s);
/* record id ("" unless parsed from FASTA; owned by the seqset) and length in bases of contig idx */
int pg_seqset_contig(const pg_seqset *s, uint32_t idx, const char **name, uint64_t *len);
/* all contigs at once: lens[ncontigs] (may be NULL) and the names, each followed by a NUL, into names[names_cap] (may be
 * NULL); *names_bytes = the bytes the names take.  (A call per contig costs the interpreter 1.6 us: a quarter of a
 * second for an assembly of 20 000 contigs times eight genomes.) */
int pg_seqset_describe(const pg_seqset *s, uint64_t *lens, char *names, uint64_t names_cap, uint64_t *names_bytes);
/* diagnostic: contig idx back as len ASCII bytes — ACGT upper case, 'N' for every byte the
 * packing treats as "not ACGT" (what GetCountersForRead's window test sees); synchronises */
int pg_seqset_unpack(const pg_seqset *s, uint32_t idx, char *out);

/* ---- anchoring ---------------------------------------------------------
 * Reference: KMCdb::write_bits (cpp/anchor.cpp:112-195) per contig =
 * Genome._write_bitmap + bin_bitsum + paircount sums (index.py:949-969,
 * 1048-1051, 1169-1183).  For every k-mer position of every contig: the
 * nbytes-byte presence row (bitmap.1), every row whose contig-relative index is
 * a multiple of 100 (bitmap.100), the per-bin popcount histogram (N+1 counters
 * per bin; bin length 200000 or nkmers/100) and per-genome column sums. */
#define PG_ANCHOR_COLSUMS 1u   /* also accumulate per-genome column sums */
#define PG_ANCHOR_COLUMNS_ONLY 4u /* (with ROWS_ONLY) no row buffer: the result only emits bit columns (pg_anchor_run_columns_range) */
#define PG_ANCHOR_ROWS_ONLY 2u /* genome-sharded mode: pg_anchor_run writes only the bitmap.1 rows (this
                                * GPU's genomes' bits); after the rows of all GPUs have been combined in
                                * place (RCCL) pg_rows_epilogue derives the rest */
int pg_result_create(pg_table *tbl, const pg_seqset *seqs, uint32_t flags, pg_result **out);
/* the same with the Python path's parameters (index.py:101-106, 1169-1172; cpp/anchor.cpp:114-118,169-176
 * hard-codes 100 / 200000 / 100): the low-resolution bitmap keeps every lowres_step-th row (bitmap.<lowres_step>;
 * "bitmap100" / step 100 in the calls below then mean that bitmap), bins are max_bin_len positions long unless the
 * contig would get fewer than min_bin_count of them (then nkmers / min_bin_count) */
int pg_result_create_ex(pg_table *tbl, const pg_seqset *seqs, uint32_t flags, uint32_t lowres_step,
                        uint64_t max_bin_len, uint32_t min_bin_count, pg_result **out);
/* a result WITHOUT a table: an ngenomes-wide row buffer over the contigs of `seqs` (zeroed) that is filled
 * through pg_result_merge_columns_range and finished with pg_rows_epilogue — the writer's side of the
 * genome-sharded mode, where no single GPU holds a table of all genomes */
int pg_result_create_rows(pg_ctx *ctx, int k, int ngenomes, const pg_seqset *seqs, uint32_t flags, uint32_t lowres_step,
                          uint64_t max_bin_len, uint32_t min_bin_count, pg_result **out);
int pg_result_destroy(pg_result *r);
/* run the anchor kernels for all contigs of the result's seqset; async */
int pg_anchor_run(pg_result *r);
/* PG_ANCHOR_ROWS_ONLY results: probe contigs first_contig .. first_contig+ncontigs-1 only (bitmap.1 rows, no
 * statistics); async.  The unit of the genome-sharded pipeline: probe a chunk, extract its columns, exchange them
 * while the next chunk is probed. */
int pg_anchor_run_range(pg_result *r, uint32_t first_contig, uint32_t ncontigs);
/* the same, emitting the contig range's compact bit columns (`width` genomes wide, the layout of
 * pg_result_extract_columns_range) STRAIGHT from the probe into d_dst — no rows are written or read back.  For the
 * narrow block tables of the genome-sharded mode: pg_result_columns_direct() says whether the result's table
 * qualifies (up to 8 genomes, width in ngenomes..8); config 5 — one genome per GPU — does.  A result created with
 * PG_ANCHOR_COLUMNS_ONLY allocates no row buffer at all and only takes this call.  async */
int pg_result_columns_direct(const pg_result *r, uint32_t width);
int pg_anchor_run_columns_range(pg_result *r, uint32_t first_contig, uint32_t ncontigs, uint32_t width, void *d_dst);
/* HIP-event durations of the last pg_anchor_run on this result, measured on the context's
 * stream: the probe kernels (k_probe, one per sub-table) and the statistics kernel
 * (k_epilogue; 0 in rows-only mode).  Synchronises on the run's last event. */
int pg_result_timing(pg_result *r, float *probe_ms, float *epilogue_ms);
/* mean durations over EVERY run since pg_result_timing_reset (each run records into events of its own, nothing
 * is waited for between runs): what a benchmark quotes as its average launch.  nruns = probe launches counted. */
int pg_result_timing_reset(pg_result *r);
int pg_result_timing_mean(pg_result *r, double *probe_ms, double *epilogue_ms, uint32_t *nruns);
/* Fused statistics (round 6, an opt-in experiment: PG_FUSE_STATS=1 in the environment): a pg_anchor_run over rows of 2..8
 * bytes ends every tile INSIDE k_probe with the tile's popcount histogram, column sums and 1-in-100 rows — what
 * cpp/anchor.cpp:156-183 does in its scatter loop — read back from the rows while the cache still holds them; a small kernel
 * adds the tiles' counters up (k_tile_reduce) and the statistics pass (k_epilogue*) only visits contigs whose bins are
 * shorter than a tile.  The outputs are the same bytes either way (tests/test_gpu_parity.py); the default stays the pass over
 * every row, which is faster.  *n = whole runs of this result that took the fused path so far (0: none did —
 * not asked for, one-byte rows, rows wider than 8 bytes, several sub-tables, no contig with long enough bins, no memory for the
 * tiles' counters: 2-6 % of the rows).  No reference counterpart: a diagnostic for tests and bench.py. */
int pg_result_fused_runs(const pg_result *r, uint32_t *n);
/* Genome-sharded exchange (union of tables > one GPU's HBM): rank i owns genomes
 * [i*per, (i+1)*per) and its PG_ANCHOR_ROWS_ONLY rows hold only their bits.  extract writes the
 * compact block of bit columns of genomes [g0, g0+width) — pg_result_columns_bytes(width) bytes,
 * one u64 per genome per 64 positions — into device memory; the blocks of all ranks, all-gathered
 * over RCCL/xGMI into one buffer (nparts blocks of equal size, block i = genomes from i*per), are
 * merged back into full rows by merge.  Both are asynchronous on the context's stream. */
uint64_t pg_result_columns_bytes(const pg_result *r, uint32_t width);
int pg_result_extract_columns(pg_result *r, uint32_t g0, uint32_t width, void *d_dst);
int pg_result_merge_columns(pg_result *r, const void *d_src, uint32_t nparts, uint32_t per);
/* the same over a contig range (the pipeline's chunk): the block then holds only that range's tiles.  merge:
 * the nparts blocks are genome blocks part0 .. part0+nparts-1 of `per` genomes each; accumulate != 0 ORs their bits
 * into the rows (blocks arriving pass by pass — more genome blocks than GPUs), 0 writes the rows whole.
 * part_stride_bytes: distance between consecutive blocks in d_src (0: the range's own block size) — larger when the
 * range is a slice of a bigger exchanged chunk (d_src then points at the slice inside the first block). */
uint64_t pg_result_columns_bytes_range(const pg_result *r, uint32_t width, uint32_t first_contig, uint32_t ncontigs);
int pg_result_extract_columns_range(pg_result *r, uint32_t g0, uint32_t width, uint32_t first_contig, uint32_t ncontigs,
                                    void *d_dst);
int pg_result_merge_columns_range(pg_result *r, const void *d_src, uint32_t part0, uint32_t nparts, uint32_t per,
                                  uint32_t first_contig, uint32_t ncontigs, int accumulate, uint64_t part_stride_bytes);
/* bitmap.100 rows, bin histograms and column sums from the (combined) bitmap.1 rows in the
 * result's device buffer; async.  Same outputs as the fused pg_anchor_run path. */
int pg_rows_epilogue(pg_result *r);
/* geometry of contig idx: k-mer count, bitmap.100 row count, bin count, bin length */
int pg_result_contig_info(const pg_result *r, uint32_t idx, uint64_t *nkmers, uint64_t *nrows100,
                          uint32_t *nbins, uint32_t *binlen);
/* copy contig idx's outputs to host (any pointer may be NULL); synchronises.
 *   bitmap1:   nkmers  * nbytes bytes        bitmap100: nrows100 * nbytes bytes
 *   bins:      nbins * (ngenomes+1) u32      (row b covers [b*binlen, ...)) */
/* statistics of nwin row windows [starts[i], ends[i]) (clipped to the contig) of contig idx's
 * bitmap.<step> rows in HBM: hist[i*(ngenomes+1) + c] = rows of the window with popcount c,
 * colsums[i*ngenomes + g] = rows with genome g's bit (colsums may be NULL).  One kernel for the
 * reference's per-gene occupancy tabulation (index.py:1055-1064), bin_bitsum with any bin length
 * (index.py:1169-1183) and bitmap_to_bins / paircount bins (index.py:438-449).  Synchronises. */
int pg_result_window_stats(pg_result *r, uint32_t idx, int step, uint32_t nwin, const uint64_t *starts,
                           const uint64_t *ends, uint64_t *hist, uint64_t *colsums);
/* stream the whole bitmap.1 (step 1) or bitmap.100 (step 100) payload of the result — every
 * contig, in order — from HBM into a BGZF file + .gzi index (gzi_path may be NULL): D2H through
 * pinned double buffers on a private stream overlapped with multi-threaded deflate.  Replaces
 * bgzf_open/bgzf_write/bgzf_index_dump/bgzf_close of cpp/anchor.cpp:46-55,102-106,167,177.  Waits
 * for the run's kernels; may be called from another host thread than the one enqueueing work.
 * level: a zlib level 0..9 for the host path (Z_RLE for one-byte rows, the library's row-aware
 * encoder for wider ones) — or -2: the BGZF blocks are compressed ON THE GPU (k_row_deflate: one
 * workgroup per 65280-byte block, matches one row back, dynamic Huffman, CRC32), the host only
 * appends the finished blocks to the file (nthreads unused). */
int pg_result_write_bgzf(pg_result *r, int step, const char *gz_path, const char *gzi_path, int level,
                         int nthreads);
/* the same for contigs first_contig .. first_contig+ncontigs-1 only: one anchor genome of a result
 * that spans several (pg_seqset_concat / pg_result_coschedule) */
int pg_result_write_bgzf_range(pg_result *r, int step, uint32_t first_contig, uint32_t ncontigs,
                               const char *gz_path, const char *gzi_path, int level, int nthreads);
/* Launch-order hint for a result over SEVERAL anchor genomes (contig_group[c] = genome of contig
 * c; NULL restores launch order): the reference anchors its FASTAs in parallel threads
 * (cpp/anchor.cpp:217-223); here all of them share one kernel launch whose tiles interleave the
 * genomes piece by piece (piece_tiles tiles, 0 = default), each genome traversed at the same
 * relative pace, so that homologous regions — which need the same table lines — run side by side
 * and the lines are fetched from HBM once instead of once per genome.  Results do not depend on
 * the schedule. */
int pg_result_coschedule(pg_result *r, const uint32_t *contig_group, uint32_t piece_tiles);
/* the same when the genomes do not list their contigs in corresponding order: contig_class[c] names the homology
 * class of contig c (e.g. the chromosome: equal record ids in different FASTAs); the classes are scheduled one after
 * the other, within a class the genomes' contigs side by side as above.  (With shuffled contig order the plain
 * co-schedule degrades to the one-launch-per-genome rate: tools/indel_cosched.py --shuffle-contigs.) */
int pg_result_coschedule_classes(pg_result *r, const uint32_t *contig_group, const uint32_t *contig_class,
                                 uint32_t piece_tiles);
/* the same with the contigs cut into nranges consecutive ranges (range i starts at contig range_first_contig[i];
 * the first at 0) that are scheduled independently: pg_anchor_run_range over one or several whole ranges then
 * follows the schedule too (any other range runs in launch order) */
int pg_result_coschedule_ranges(pg_result *r, const uint32_t *contig_group, uint32_t piece_tiles,
                                const uint32_t *range_first_contig, uint32_t nranges);
/* column sums of contigs idx .. idx+ncontigs-1 (ncontigs x ngenomes u64); pg_result_colsums is
 * their total */
int pg_result_contig_colsums(pg_result *r, uint32_t idx, uint32_t ncontigs, uint64_t *colsums);
int pg_result_download(pg_result *r, uint32_t idx, uint8_t *bitmap1, uint8_t *bitmap100,
                       uint32_t *bins);
/* The small outputs of contigs first .. first+ncontigs-1 in ONE call (an assembly of 20 000 contigs paid 26 us per
 * contig for pg_result_contig_info + pg_result_download): the geometry into four arrays of ncontigs entries each (any
 * may be NULL) and, when bins != NULL, the contigs' bin rows back to back — bins_words must be the sum of
 * nbins[i] * (ngenomes + 1) — in one device-to-host copy; synchronises.  What KMCdb::write_bits accumulates per chunk
 * (cpp/anchor.cpp:179-189), per contig. */
int pg_result_contigs_small(pg_result *r, uint32_t first, uint32_t ncontigs, uint64_t *nkmers, uint64_t *nrows100,
                            uint32_t *nbins, uint32_t *binlen, uint32_t *bins, uint64_t bins_words);
/* per-genome column sums over ALL contigs of the seqset (ngenomes u64); synchronises */
int pg_result_colsums(pg_result *r, uint64_t *colsums);
/* device pointers (for benchmarking / checksums without PCIe traffic) */
int pg_result_device_ptrs(pg_result *r, void **d_bitmap1, uint64_t *bitmap1_bytes,
                          void **d_bitmap100, uint64_t *bitmap100_bytes);

/* one-shot convenience for a single contig held in host memory: upload, pack,
 * anchor, download.  nkmers = len-k+1 (0 and a warning-free no-op when len<k:
 * the reference underflows here, cpp/anchor.cpp:115).  Any output may be NULL. */
int pg_anchor_contig(pg_table *tbl, const char *ascii, uint64_t len, uint8_t *bitmap1,
                     uint8_t *bitmap100, uint32_t *bins, uint64_t *colsums, uint64_t *nkmers);

/* literal equivalent of CKMCFile::GetCountersForRead(seq, vector<uint32>&)
 * (call sites cpp/anchor.cpp:148, index.py:934-935) for group db_idx:
 * out[i] = counter of the canonical k-mer of ascii[i:i+k], 0 if absent or if the
 * window holds a byte outside ACGTacgt.  out has len-k+1 entries. */
int pg_counters_for_read(pg_table *tbl, int db_idx, const char *ascii, uint64_t len,
                         uint32_t *out);

/* ---- BGZF + .gzi writer (host, zlib, multi-threaded) -------------------
 * Reference: htslib bgzf_open/bgzf_index_build_init/bgzf_write/
 * bgzf_index_dump/bgzf_close (cpp/anchor.cpp:46-47,53-54,102-106,167,177) and
 * bgzip.BGZipWriter + `bgzip -rI` (index.py:1035-1037,1091-1094).  Blocks hold
 * <= 65280 uncompressed bytes; X.gzi = u64 n, n x (u64 compressed_off, u64
 * uncompressed_off) for blocks 1..n (read by index.py:793-799).
 * level: zlib level 0..9 (other values: 6), optionally | PG_BGZF_RLE = zlib's Z_RLE strategy
 * (matches at distance 1 only).  For ONE-byte rows (N <= 8) runs of equal rows are byte runs:
 * same compression ratio as the default strategy, 6.6x faster (tools/zstrategy.py); useless for
 * wider rows.  pg_result_write_bgzf adds it by itself for one-byte rows.
 * | PG_BGZF_ROWS(w): the payload is rows of w bytes (2..255): a row-aware DEFLATE encoder (matches
 * "same byte as one row earlier" only, dynamic Huffman per block) replaces zlib's matcher, which
 * runs at ~60 MB/s/core on such data; falls back to zlib per block if its output does not fit.
 * pg_result_write_bgzf uses it for rows wider than one byte. */
#define PG_BGZF_RLE 0x100
#define PG_BGZF_ROWS(w) (((w) & 0xff) << 16)
int pg_bgzf_open(const char *path, int level, int nthreads, pg_bgzf **out);
int pg_bgzf_write(pg_bgzf *w, const void *data, size_t len);
/* writes the EOF block, closes the file and, if gzi_path != NULL, the index */
int pg_bgzf_close(pg_bgzf *w, const char *gzi_path);

/* bitsum.bins.tsv as KMCdb::anchor_fasta / write_bits write it (cpp/anchor.cpp:57-69,184-189): the header line
 * "chr\tstart\t0\t1...\tN", then for contig c (numbered from 0) and bin b the line
 * "c\t(b * binlen[c])\tcount_0...\tcount_N"; bins = the contigs' rows back to back, ngenomes + 1 counts each.
 * Host code, no GPU; the text of millions of bins is formatted here rather than row by row in the interpreter. */
int pg_write_bins_tsv(const char *path, uint32_t ngenomes, uint32_t ncontigs, const uint32_t *nbins, const uint32_t *binlen,
                      const uint32_t *bins);

#ifdef __cplusplus
}
#endif
#endif /* PANAGRAM_HIP_H */
