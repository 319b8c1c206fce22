/* ============================================================================
 * sage2ov.h -- C ABI of the MI355X-native SAGE2 read-overlap path (CLI steps 1-3; step 4 at the end of the file).
 *
 * SAGE2 has no plugin/FFI interface; its replaceable seam is (a) the step/prefix file
 * API (<outdir>/<prefix>.reads + <prefix>.graph3, consumed by `SAGE2 -m 4 -i <prefix>`,
 * main.cpp:141-148) and (b) in process, the four classes main.cpp:44-131 drives:
 * ReadLoader -> HashTable -> EconomyGraph -> OverlapGraph.  Each entry point below
 * replaces the member function cited next to it (paths relative to the reference tree).
 *
 * Conventions: opaque context, plain pointers and sizes, no exceptions across the ABI,
 * `int` status (0 ok, <0 error; text via sage2ov_last_error), caller-owned output
 * buffers, one host thread per context (distinct contexts may live on distinct threads).
 * All device work is hand-written HIP for gfx950 on the context's own stream; there is no
 * CPU fallback: every compute call fails with SAGE2OV_ERR_DEVICE when no GPU is usable.
 * ========================================================================== */
#ifndef SAGE2OV_H_
#define SAGE2OV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAGE2OV_OK            0
#define SAGE2OV_ERR_ARG      -1   /* bad argument / call order */
#define SAGE2OV_ERR_IO       -2   /* file could not be opened / parsed */
#define SAGE2OV_ERR_DEVICE   -3   /* HIP error or no GPU */
#define SAGE2OV_ERR_NOMEM    -4
#define SAGE2OV_ERR_LIMIT    -5   /* input exceeds a documented limit */
/* The documented limits of a context (SAGE2OV_ERR_LIMIT; the reference's widths next to them):
 *   - read length <= 1018 bases (32 words of 2-bit bases with the 11-bit length field in the last one; inputReader/readLoader.h:28 holds any uint16_t length:
 *     reads of 1019 .. 65535 bases are refused by the reads_add_* / reads_organize calls, not truncated);
 *   - fewer than 2^30 unique reads per context (position * 4 + type is a 32-bit entry; economyGraph/hashTable.h:13-18 carries 40-bit ids);
 *   - at most 2^32 slot pairs of the table (8 slots per read: never reached below the previous limit). */
#define SAGE2OV_ERR_INTERNAL -6

typedef struct sage2ov_ctx sage2ov_ctx;

#define SAGE2OV_DEVICE_CURRENT (-1)
#define SAGE2OV_DEVICE_NONE    (-2)  /* step-1-only context (what `-M 1` needs): parsing, canonical order, P.reads.
                                        Every step 2/3 call on it fails with SAGE2OV_ERR_DEVICE. */

typedef struct sage2ov_config {
    uint32_t min_overlap;   /* -k (main.cpp:424); hash string length h = min(k,64) (hashTable.cpp:78-81) */
    int32_t  device;        /* HIP device ordinal; SAGE2OV_DEVICE_CURRENT; SAGE2OV_DEVICE_NONE */
    uint32_t rank;          /* multi-GPU: this context's rank ...            */
    uint32_t world;         /* ... of `world` ranks (0 or 1 = single GPU)    */
    uint32_t host_threads;  /* OpenMP threads for host-side step 1 (0 = default) */
    uint32_t flags;         /* SAGE2OV_FLAG_* */
} sage2ov_config;

#define SAGE2OV_FLAG_HOST_REDUCE 1u  /* force the exact serial BFS replay (economyGraph.cpp:513-564) on the host
                                        even when the order-independent device form would be exact */
#define SAGE2OV_FLAG_ASYNC_DEVICE 2u /* open the device on a helper thread: sage2ov_ctx_create returns at once and the caller parses its
                                        input files meanwhile (the HIP runtime takes ~0.2 s to start).  The first call that needs the
                                        device waits for it; a device that cannot be opened is reported there (SAGE2OV_ERR_DEVICE) */

/* ---- lifetime: replaces new/delete of the four classes (main.cpp:44,76,108,116) ---- */
int  sage2ov_ctx_create(const sage2ov_config* cfg, sage2ov_ctx** out);
void sage2ov_ctx_destroy(sage2ov_ctx* ctx);
const char* sage2ov_last_error(const sage2ov_ctx* ctx);   /* ctx may be NULL: error of the last failed create */
/* The library's SAGE2OV_* environment switches (test-only route overrides, grid sweeps, diagnostics -- the reference has none,
 * main.cpp:384-521 takes flags only) are read ONCE, when the context is created; no step reads the environment.  This call takes a new snapshot: for tests
 * and diagnostics that change a switch between two steps of one context. */
int sage2ov_options_reload(sage2ov_ctx* ctx);
const char* sage2ov_version(void);

/* ---- STEP 1: ReadLoader (inputReader/readLoader.h:44-55) ---- */
typedef struct sage2ov_read_stats {
    uint64_t total_reads;        /* ReadLoader::totalReads                      */
    uint64_t good_reads;         /* ReadLoader::numberOfReads (incl. duplicates) */
    uint64_t unique_reads;       /* ReadLoader::numberOfUniqueReads = N          */
    uint64_t total_bp;           /* ReadLoader::totalBP                          */
    uint64_t average_read_length;/* global averageReadLength = totalBP/numberOfReads (readLoader.cpp:161) */
    uint32_t max_read_length;
    uint32_t words_per_read;     /* u64 words per read slot in HBM */
} sage2ov_read_stats;

/* readDatasetInBytes' loop body (readLoader.cpp:145-158): filter (len<=k, non-ACGT; utils.cpp:144),
 * canonical orientation (readLoader.cpp:179-213), 2-bit pack (utils.cpp:96).  Read r is
 * bases[offsets[r] .. offsets[r+1]). */
int sage2ov_reads_add_ascii(sage2ov_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n);
/* ReadLoader::readDatasetInBytes(mateFile1, mateFile2) (readLoader.cpp:133): FASTA/FASTQ, optionally
 * gzip; with two files reads alternate file1/file2 (inputReader.cpp:26-49). path2 may be NULL. */
int sage2ov_reads_add_file(sage2ov_ctx* ctx, const char* path1, const char* path2);
/* ReadLoader::loadFromList (readLoader.cpp:73): list grammar f1=/f2=/f=, '#' comments. */
int sage2ov_reads_add_list(sage2ov_ctx* ctx, const char* list_path);
/* ReadLoader::organizeReads (readLoader.cpp:215) + the orientation choice of insertReadIntoList (readLoader.cpp:195):
 * canonical orientation, sort by stringCompareInBytes order (utils.cpp:224), unique with frequency (u16 wrap), ids 1..N.
 * With a GPU context all of it runs on the device (the staged forward reads are uploaded once) and the read store stays
 * resident in HBM; a device-less context (SAGE2OV_DEVICE_NONE) uses the host organiser. */
int sage2ov_reads_organize(sage2ov_ctx* ctx);
int sage2ov_reads_stats(const sage2ov_ctx* ctx, sage2ov_read_stats* out);
/* class Read fields (readLoader.h:21-30) for ids 1..N, arrays indexed [0..N] (entry 0 unused):
 * packed = MSB-first 2-bit bytes (utils.cpp:96) of the forward strand, `stride` bytes apart. */
int sage2ov_reads_export(const sage2ov_ctx* ctx, uint8_t* packed, uint64_t stride, uint16_t* length, uint16_t* frequency);
int sage2ov_reads_save(sage2ov_ctx* ctx, const char* path);   /* saveReadsInFile  (readLoader.cpp:270) -> P.reads */
int sage2ov_reads_load(sage2ov_ctx* ctx, const char* path);   /* loadReadsFromFile (readLoader.cpp:289) + upload  */
/* graph3 header fields when reads came from P.reads (good_reads / average length are not in that file;
 * main.cpp:141-148 likewise takes them from the graph file) */
int sage2ov_reads_set_totals(sage2ov_ctx* ctx, uint64_t good_reads, uint64_t total_bp);
/* The organised read store in its HBM word layout ((N+1) x words_per_read u64): one rank
 * organises, the others import the image (multi-GPU: every rank needs all reads). */
int sage2ov_reads_export_words(const sage2ov_ctx* ctx, uint64_t* words, uint64_t cap_words, uint16_t* frequency);
int sage2ov_reads_import_words(sage2ov_ctx* ctx, const uint64_t* words, uint64_t n_unique, uint32_t words_per_read,
                               uint32_t max_read_length, const uint16_t* frequency, uint64_t good_reads, uint64_t total_bp);

/* ---- STEP 2: HashTable (economyGraph/hashTable.h:20-43) ---- */
typedef struct sage2ov_index_stats {
    uint64_t slots;          /* open-addressed 8-byte slots in HBM           */
    uint64_t keys;           /* occupied slots (distinct prefix/suffix keys, up to tag merges) */
    uint64_t csr_entries;    /* bucket entries stored out of line            */
    uint64_t long_buckets;   /* buckets with >= 100 entries (hashTable.cpp:111-123): never found */
    uint32_t hash_string_length;
    uint32_t rebuilds;       /* reseeds after an impure long bucket */
    uint32_t minimiser_groups; /* 1: the fast probe kernel's second access path (keys grouped by minimiser) was built and is used; the library decides
                                * by the share of reads without a predecessor in the locality order */
    uint32_t reserved;
} sage2ov_index_stats;
int sage2ov_index_build(sage2ov_ctx* ctx);                      /* hashPrefixesAndSuffix (hashTable.cpp:70) */
int sage2ov_index_stats_get(const sage2ov_ctx* ctx, sage2ov_index_stats* out);
/* hashTableSearch (hashTable.cpp:193) + the bucket walk of economyGraph.cpp:89-93 for one key
 * (key[0]=v0 leading bases, key[1]=v1 last 32 bases, utils.cpp:171-187).  Entries are id*4+type in
 * bucket order; *count = 0 for "not found" (absent or long).  Test/diagnostic entry point. */
int sage2ov_index_lookup(sage2ov_ctx* ctx, const uint64_t key[2], uint64_t* entries, uint32_t cap, uint32_t* count);
/* saveHashTableInFile (hashTable.cpp:256-273, :12-20) -> P.hashTable: the slot-by-slot text dump of the REFERENCE's double-hashed table (what
 * `SAGE2 -m 3` loads).  Our index has another shape, so the reference's serial insertion (hashTable.cpp:94-109, :133-187: table size, start
 * slot, probe step, 101-entry cap, long-bucket flag) is replayed on the host for this file alone.  Needs organised reads only (no GPU). */
int sage2ov_hashtable_save(sage2ov_ctx* ctx, const char* path);

/* ---- STEP 3: EconomyGraph + OverlapGraph::convertGraph ---- */
typedef struct sage2ov_overlap_stats {
    uint64_t verified_overlaps;   /* sum of `connections` (economyGraph.cpp:96,189,281,361) = N_ov */
    uint64_t contained_extension; /* "Total contained by extension" (economyGraph.cpp:485) */
    uint64_t contained_size;      /* "Total contained by size"      (economyGraph.cpp:486) */
    uint64_t left_to_explore;     /* economyGraph.cpp:487 */
    uint64_t edges_inserted;      /* "Total edges inserted"   (economyGraph.cpp:569), twins counted */
    uint64_t transitive_removed;  /* "Transitive edge removed"(economyGraph.cpp:571) */
    uint64_t edges;               /* undirected edges in the canonical list (= record pairs in P.graph3) */
    uint64_t unresolved_hits;     /* directional hits of status-0 reads fed to the reduce phase */
} sage2ov_overlap_stats;
/* buildInitialOverlapGraph (economyGraph.cpp:37): probe+verify+extension kernel over this rank's read
 * range, then the reciprocal pass (economyGraph.cpp:455-480). */
int sage2ov_overlap_initial(sage2ov_ctx* ctx);
/* buildOverlapGraphEconomy (economyGraph.cpp:495): all-edges + transitive reduction of unresolved reads.
 * On the device; when the index hides keys (>= 100 entries, hashTable.cpp:111-123) the order in which the serial BFS explores
 * the reads decides which edges exist (:605) and is walked on the host over device-built lists. */
int sage2ov_overlap_reduce(sage2ov_ctx* ctx);
/* sortEconomyGraph (economyGraph.cpp:896) + OverlapGraph::convertGraph (overlapGraph.cpp:84):
 * canonical edge list, ascending (from,to,type,length), from<to, one per (from,to,type). */
int sage2ov_overlap_convert(sage2ov_ctx* ctx);
int sage2ov_overlap_stats_get(const sage2ov_ctx* ctx, sage2ov_overlap_stats* out);
/* per-read results of the initial pass, arrays [0..N]: ExtensionTable records packed exactly like
 * economyGraph.h:24-30 (readId:40 | type:2 | length:22), exploredReads value (0/4/5/6), connections. */
int sage2ov_overlap_export_initial(sage2ov_ctx* ctx, uint64_t* right_ext, uint64_t* left_ext, uint8_t* status, uint32_t* connections);

typedef struct sage2ov_edge {      /* one record pair of P.graph3 (overlapGraph.cpp:12-20,136-163) */
    uint64_t from, to;             /* from < to */
    uint32_t length;               /* Edge::lengthOfEdge of from->to  */
    uint32_t length_twin;          /* lengthOfEdge of the twin to->from */
    uint8_t  type;                 /* typeOfEdge of from->to; twin = reverseEdgeType (utils.cpp:212) */
    uint8_t  pad[7];
} sage2ov_edge;
int sage2ov_edges_count(const sage2ov_ctx* ctx, uint64_t* n);
int sage2ov_edges_export(sage2ov_ctx* ctx, sage2ov_edge* out, uint64_t cap);
int sage2ov_graph_save(sage2ov_ctx* ctx, const char* path);     /* saveOverlapGraphInFile (overlapGraph.cpp:338) -> P.graph3 */
/* loadOverlapGraphFromFile's role for a graph of simple edges (overlapGraph.cpp:371): take this edge list (the layout edges_export
 * writes; pairs are pushed in the order given) instead of computing it -- the entry for step 4 on an existing graph. */
int sage2ov_edges_import(sage2ov_ctx* ctx, const sage2ov_edge* edges, uint64_t n);
int sage2ov_graph_load(sage2ov_ctx* ctx, const char* path);     /* loadOverlapGraphFromFile (overlapGraph.cpp:371) <- P.graph3 (simple edges) */

/* ---- step 4: simplification of the overlap graph on the device, main.cpp:139-172 over overlapGraph/simplification.cpp:
 * contractCompositePaths (:14), removeDeadEnds (:58), removeBubbles (:118) in the reference's loop, with its growing thresholds.
 * Needs sage2ov_overlap_convert.  The result is the graph the reference holds in memory after its step 4; sage2ov_graph4_save writes
 * it with saveOverlapGraphInFile's format (overlapGraph.cpp:338-369, :12-20) as `P.graph4`, the file the reference's step 5 reads
 * (main.cpp:196) but none of its steps writes. */
typedef struct sage2ov_simplify_stats {
    uint64_t nodes_contracted;     /* sum of contractCompositePaths' "Nodes removed" */
    uint64_t removed;              /* sum of removeDeadEnds' and removeBubbles' return values */
    uint64_t loop_iterations;      /* passes of the while(1) loop, main.cpp:158 */
    uint64_t edges;                /* surviving edge pairs */
    uint64_t reads_on_edges;       /* entries of the surviving forward read lists */
    double   device_ms;
} sage2ov_simplify_stats;
int sage2ov_graph_simplify(sage2ov_ctx* ctx);
int sage2ov_simplify_stats_get(const sage2ov_ctx* ctx, sage2ov_simplify_stats* out);
int sage2ov_graph4_save(sage2ov_ctx* ctx, const char* path);

/* diagnostic: table census {occupied, inline, claimed-but-unfilled, zero-tag, entries in short CSR buckets} */
int sage2ov_debug_table(sage2ov_ctx* ctx, uint64_t* out5);
/* diagnostic: device memory {free now, total, LOWEST free seen at the library's sampling points since the context was created (end of step 1, the index
 * build's peak, end of the probe pass, reduce, convert), bytes held by the context's grow-only workspace arena}: the high-water mark of a run is
 * total - out[2] when the context is the only user of the GPU */
int sage2ov_debug_meminfo(sage2ov_ctx* ctx, uint64_t* out4);
/* diagnostic: the four index keys (hashTable.cpp:96-104) of every read as the device computes them: 8N u64 (hi,lo per entry) */
int sage2ov_debug_keys(sage2ov_ctx* ctx, uint64_t* out);
/* diagnostic: the directional hit list (economyGraph.cpp:591-633: to, edge type, length) of EVERY read, rows of
 * 5 x u32 {from, to, type, length, sequence-number}, sorted by (from, sequence).  out may be NULL to size. */
int sage2ov_debug_all_hits(sage2ov_ctx* ctx, uint32_t* out, uint64_t cap_rows, uint64_t* n_rows);

/* whole timed region of SURVEY 8(d): index build + initial + reduce + convert, reads already in HBM */
int sage2ov_run_steps23(sage2ov_ctx* ctx);

/* ---- multi-GPU exchange points (one process per GPU; the caller owns the communicator and the
 * collective -- RCCL through torch.distributed or directly -- and hands device buffers in and out).
 * Rank r probes read ids [lo_r, hi_r); the reciprocal test reads the neighbours' records
 * (economyGraph.cpp:460), hence the all-gather of fixed-size per-read records, then of edge buckets. */
int sage2ov_shard_range(const sage2ov_ctx* ctx, uint64_t* lo, uint64_t* hi);          /* ids, hi exclusive */
int sage2ov_shard_record_bytes(const sage2ov_ctx* ctx, uint64_t* bytes_per_read);     /* 16: right and left extension (position:30, type:2, length:22 each), connection count:18, containment flags:2 */
int sage2ov_overlap_probe_shard(sage2ov_ctx* ctx);                                    /* kernel only, own range */
int sage2ov_shard_export_records(sage2ov_ctx* ctx, void* dev_dst, uint64_t max_reads);/* own range -> device buffer */
int sage2ov_shard_import_records(sage2ov_ctx* ctx, const void* dev_src, uint64_t first_id, uint64_t n_reads);
/* containment marks (economyGraph.cpp:735) land on reads of ANY rank: 2*(N+1) bytes (two bit planes) that the
 * caller MAX-all-reduces (= bitwise OR) between export and import */
int sage2ov_shard_flags_bytes(const sage2ov_ctx* ctx, uint64_t* bytes);
int sage2ov_shard_export_flags(sage2ov_ctx* ctx, void* dev_dst);
int sage2ov_shard_import_flags(sage2ov_ctx* ctx, const void* dev_src);
/* reciprocal test for every read (cheap, replicated), edge buckets only for this rank's reads */
int sage2ov_overlap_reciprocal(sage2ov_ctx* ctx);
/* per-rank edge buckets (16-byte records {from,to,len,type}): all-gather counts, then the padded buckets,
 * then hand every rank the concatenation */
int sage2ov_shard_edges_count(const sage2ov_ctx* ctx, uint64_t* n);
int sage2ov_shard_edges_export(sage2ov_ctx* ctx, void* dev_dst, uint64_t cap_edges);
int sage2ov_shard_edges_set(sage2ov_ctx* ctx, const void* dev_src, uint64_t n_edges);

/* Sharded reduce phase (round 3).  On a context with world > 1, sage2ov_overlap_reduce (buildOverlapGraphEconomy, economyGraph.cpp:495-574) builds the hit
 * lists and the adjacency of ALL unresolved reads -- every rank's marks read them -- but runs markTransitiveEdge / removeTransitiveEdges (:643-707) and the
 * re-emission of the surviving edges only for THIS RANK'S SHARE of the unresolved reads (a contiguous part of their list; any cut is exact: a read's marks
 * are a function of the lists).  The survivors of the share are this rank's survivor bucket (16-byte records like the edge buckets): all-gather the buckets,
 * sum `removed_partial` over the ranks, and hand every rank the concatenation before sage2ov_overlap_convert (which refuses to run without it).
 * Where the phase runs replicated (the serial replay for a handful of reads), rank 0's bucket carries everything and the others are empty. */
int sage2ov_shard_survivors_count(const sage2ov_ctx* ctx, uint64_t* n, uint64_t* removed_partial);
int sage2ov_shard_survivors_export(sage2ov_ctx* ctx, void* dev_dst, uint64_t cap_edges);
int sage2ov_shard_survivors_set(sage2ov_ctx* ctx, const void* dev_src, uint64_t n_total, uint64_t removed_total);

/* ---- profiling hooks: HIP-event timings of the last run, milliseconds ---- */
typedef struct sage2ov_timings {
    double index_ms, probe_ms, reciprocal_ms, reduce_ms, convert_ms, total_ms;
    double probe_kernel_ms;      /* the dominant kernel alone (HIP events on the context stream), summed over its launches */
    uint64_t probe_kernel_launches;   /* probe PASSES timed (one per initial pass): probe_kernel_ms / this = kernel time per pass */
    uint64_t sequential_reads;   /* reads the fast kernel handed to the sequential state-machine kernel */
    double organize_ms;          /* step 1 on the device: upload of the staged reads .. organised read store resident (HIP events) */
    uint64_t probe_fast_launches;     /* launches of the fast kernel behind probe_kernel_ms: a pass is a sample launch, the rest of the range
                                         and, if reads were listed, one launch over the list */
    /* what a multi-rank run keeps replicated / shards inside the two phases above (bench.py's scaling model): */
    double reciprocal_cond_ms;        /* the part of reciprocal_ms every rank spends on ALL reads (cond(i), economyGraph.cpp:460); the rest is the emit half, sharded */
    double reduce_marks_ms;           /* the part of reduce_ms spent in the marks + removals + re-emission (economyGraph.cpp:643-707): sharded over the ranks */
} sage2ov_timings;
int sage2ov_timings_get(const sage2ov_ctx* ctx, sage2ov_timings* out);
int sage2ov_timings_reset(sage2ov_ctx* ctx);
void* sage2ov_stream(sage2ov_ctx* ctx);   /* hipStream_t the context launches on */

/* ---- deterministic synthetic reads (repo-owned, not part of the reference) ---- */
typedef struct sage2ov_synth_params {
    uint64_t seed;
    uint64_t genome_len;
    uint64_t n_reads;            /* reads, i.e. 2 x pairs; interleaved mate1, mate2 */
    uint32_t read_len;           /* L (max) */
    uint32_t read_len_min;       /* 0 or >= read_len: fixed length; else uniform in [min, L] */
    uint32_t err_ppm;            /* substitution rate, parts per million */
    uint32_t n_repeat_families;  /* planted repeats: families x copies x length */
    uint32_t repeat_copies;
    uint32_t repeat_len;
} sage2ov_synth_params;
int      sage2ov_synth_genome(const sage2ov_synth_params* p, uint8_t* genome /* genome_len codes 0..3 */);
uint32_t sage2ov_synth_read_len(const sage2ov_synth_params* p, uint64_t read_index);
int      sage2ov_synth_reads_ascii(const sage2ov_synth_params* p, const uint8_t* genome, uint64_t first, uint64_t n,
                                   char* bases, uint64_t* offsets /* n+1 */);
int      sage2ov_synth_write_fasta(const sage2ov_synth_params* p, const char* path);
/* generate reads [first, first+n) and feed them through the same filter/canonical/pack path as
 * sage2ov_reads_add_ascii, in parallel on the host, without materialising ASCII */
int      sage2ov_reads_add_synth(sage2ov_ctx* ctx, const sage2ov_synth_params* p, const uint8_t* genome, uint64_t first, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif /* SAGE2OV_H_ */
