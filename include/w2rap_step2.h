/* w2rap_step2.h -- C ABI of libw2rap_step2.so: the MI355X-native replacement for
 * w2rap-contigger's Step 2 (k=60 de Bruijn graph build + per-read pathing).
 *
 * Drop-in boundary.  Everything below replaces exactly one reference call pair,
 *     buildReadQGraph(bases, quals, false, false, minQual, minFreq, .75, 0,
 *                     &hbv, &paths, 60, out_dir, tmp_dir, disk_batches);
 *     FixPaths(hbv, paths);
 * (src/modules/w2rap-contigger.cc:338,340; declared at
 * src/paths/long/BuildReadQGraph.h:24-29 and src/paths/long/large/GapToyTools.cc:322).
 * Inputs are the flattened contents of `vecbvec` / `VecPQVec` (equivalently of
 * frag_reads_orig.fastb / .qualp, w2rap-contigger.cc:315-316,326-327); outputs are
 * what `HyperBasevector::writeBinary` (src/paths/HyperBasevector.cc:121-125) and
 * `WriteReadPathVec` (src/paths/long/ReadPath.cc:6-20) serialise, plus the
 * small_K.freqs histogram (BuildReadQGraph.cc:1094-1112).
 *
 * Plain pointers and sizes only; no C++/torch types; never throws.  Every entry
 * point returns 0 on success or a nonzero W2RAP_E_* code and a message in the
 * context (w2rap_step2_last_error).  The library is NOT re-entrant per context; use
 * one context per GPU / per host thread.  All kernels are HIP for gfx950; there is
 * no CPU fallback: without a usable device every compute entry point fails with
 * W2RAP_E_NO_DEVICE.
 */
#ifndef W2RAP_STEP2_H_
#define W2RAP_STEP2_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define W2RAP_STEP2_ABI_VERSION 3

enum {
    W2RAP_OK = 0,
    W2RAP_E_ARG = 1,        /* bad argument (K != 60, null pointer, ...) */
    W2RAP_E_NO_DEVICE = 2,  /* no HIP device / wrong architecture */
    W2RAP_E_HIP = 3,        /* a HIP runtime call failed */
    W2RAP_E_STATE = 4,      /* entry points called out of order */
    W2RAP_E_LIMIT = 5,      /* an implementation limit was exceeded (read too long, ...) */
    W2RAP_E_GRAPH = 6,      /* reference-fatal condition (BuildReadQGraph.cc:265,303: failed neighbour lookup / preoccupied k-mers) */
    W2RAP_E_HINT = 7,       /* edge_order_hint does not match the unipath set */
    W2RAP_E_IO = 8
};

/* Where the read arrays live. */
enum { W2RAP_MEM_HOST = 0, W2RAP_MEM_DEVICE = 1 };

/* ---- inputs: the flattened vecbvec / VecPQVec (BuildReadQGraph.h:24) ------------- */
typedef struct w2rap_reads {
    uint64_t n_reads;
    /* .fastb variable data: read r occupies bytes [base_byte_off[r], base_byte_off[r+1]),
     * base i at bits 2*(i%4) of byte i/4 (src/feudal/FieldVec.h:768). */
    const uint8_t*  bases_packed;
    const uint64_t* base_byte_off;   /* [n_reads+1] */
    const uint32_t* read_len;        /* [n_reads] bases per read */
    /* qualities, exactly one of the two forms:
     *  raw : one u8 per base, concatenated, read r at qual_off[r] (qual_off = prefix sum of read_len)
     *  pq  : PQVec byte strings (src/feudal/PQVec.cc:87-127), element r at [pq_off[r], pq_off[r+1]) */
    const uint8_t*  quals;           /* or NULL */
    const uint64_t* qual_off;        /* [n_reads+1] or NULL */
    const uint8_t*  pq;              /* or NULL */
    const uint64_t* pq_off;          /* [n_reads+1] or NULL */
    int32_t mem;                     /* W2RAP_MEM_HOST: arrays are copied to the GPU;
                                        W2RAP_MEM_DEVICE: device pointers, used in place (must outlive the context's use);
                                        set_reads synchronises the device once, so work queued on the caller's own streams
                                        that fills the arrays is complete before the library's (non-blocking) streams read them */
} w2rap_reads;

/* Optional replay of a reference run's (arbitrary) unipath numbering: the canonical
 * edge sequences in the order the ids must follow.  Host memory,
 * .fastb-style packing. */
typedef struct w2rap_edge_hint {
    uint64_t n_edges;
    const uint8_t*  packed;
    const uint64_t* byte_off;        /* [n_edges+1] */
    const uint32_t* len;             /* [n_edges] */
} w2rap_edge_hint;

typedef struct w2rap_step2_params {
    uint32_t K;                      /* must be 60 (BuildReadQGraph.cc:51) */
    uint32_t min_qual;               /* --min_qual, default 7 (w2rap-contigger.cc) */
    uint32_t min_freq;               /* --min_freq, default 4 */
    int32_t  device;                 /* HIP device ordinal */
    const w2rap_edge_hint* edge_order_hint;   /* NULL = canonical (lexicographic) edge order */
    const char* freqs_path;          /* NULL or path for small_K.freqs */
    /* the GPUs of this node the one in-process call may use (the reference's call is one in-process call too,
     * w2rap-contigger.cc:338, parallel inside).  0 or 1: the device `device`.  N > 1: devices device .. device+N-1, or the ordinals
     * listed in `devices` (an ordinal may repeat -- several contexts on one GPU -- which is how the tests run it on a 1-GPU box);
     * reads are sharded by rank, k-mer buckets by owner, the super-k-mer records travel by peer copies; dictionary, prune and unipaths stay
     * sharded by owner (flags & W2RAP_F_REPLICATED_GRAPH: gathered and replicated instead), the E-sized rest is built on every rank alike.
     * Where the driver grants no peer access between two of the GPUs, their exchanges are staged through pinned host memory (a warning on
     * stderr, w2rap_step2_last_peer_mode() == 2) instead of failing.  reads.mem may be W2RAP_MEM_DEVICE here too: the arrays live on ONE
     * device (any: where Step 1 left them) and every rank takes its shard from there. */
    int32_t  n_gpus;
    uint32_t n_passes;               /* counting in this many hash-range passes over the reads (the GPU analogue of --disk_batches,
                                        BuildReadQGraph.cc:1120-1250, MapReduceEngine.h:288-299): 0 = chosen from the free HBM, 1 = one pass */
    const int32_t* devices;          /* NULL or [n_gpus] device ordinals */
    uint32_t flags;                  /* W2RAP_F_* */
} w2rap_step2_params;
/* n_gpus > 1: gather the solid k-mers of every owner on every GPU and build the job's dictionary and graph everywhere (rounds 1-4)
 * instead of keeping dictionary, prune and unipaths sharded by bucket owner (row e-3, the default) */
#define W2RAP_F_REPLICATED_GRAPH 1u
/* the caller wants the graph only (pPaths == nullptr, BuildReadQGraph.cc:1300-1307): no read is pathed, out->n_paths = 0 */
#define W2RAP_F_GRAPH_ONLY 2u

/* ---- outputs (library-allocated HOST memory; free with w2rap_step2_free) ---------- */
typedef struct w2rap_step2_out {
    int32_t  K;
    /* HyperBasevector: vertices, edge objects in id order, adjacency (DigraphTemplate.h:1829-1839 order) */
    uint64_t n_vertices;
    uint64_t n_edge_objs;
    uint8_t*  edge_packed;           /* each object ceil(len/4) bytes, .fastb packing */
    uint64_t* edge_byte_off;         /* [n_edge_objs+1] */
    uint32_t* edge_len;              /* [n_edge_objs] bases */
    int32_t*  vleft;                 /* [n_edge_objs] */
    int32_t*  vright;                /* [n_edge_objs] */
    uint64_t* from_off;              /* [n_vertices+1]  CSR of from_ / from_edge_obj_ */
    int32_t*  from_v;                /* [n_edge_objs] */
    int32_t*  from_e;                /* [n_edge_objs] */
    uint64_t* to_off;                /* [n_vertices+1]  CSR of to_ / to_edge_obj_ */
    int32_t*  to_v;                  /* [n_edge_objs] */
    int32_t*  to_e;                  /* [n_edge_objs] */
    /* unipath -> object translation (HBVFromEdges.cc:137-151) */
    uint64_t n_unipaths;
    int32_t*  fwd_xlat;              /* [n_unipaths] */
    int32_t*  rev_xlat;              /* [n_unipaths] */
    /* ReadPathVec after FixPaths */
    uint64_t n_paths;
    int32_t*  path_offset;           /* [n_paths] */
    uint64_t* path_off;              /* [n_paths+1] */
    int32_t*  path_edges;            /* [path_off[n_paths]] */
    /* statistics the reference prints (BuildReadQGraph.cc:1091,1106,325,1323) + small_K.freqs */
    uint64_t hist[101];
    uint64_t n_kmer_instances;       /* M */
    uint64_t n_kmers_distinct;       /* D */
    uint64_t n_kmers_solid;          /* S */
    uint64_t n_reads_pathed;
    uint64_t n_reads_multipathed;
    /* device time of the phases, milliseconds (hipEvent) */
    float ms_count, ms_graph, ms_path;
} w2rap_step2_out;

/* ---- one-shot entry point (what a buildReadQGraph shim binds) --------------------- */
int  w2rap_step2_run(const w2rap_reads* reads, const w2rap_step2_params* params,
                     w2rap_step2_out* out, char* err, size_t errlen);
void w2rap_step2_free(w2rap_step2_out* out);
int  w2rap_step2_device_count(void);
int  w2rap_step2_abi_version(void);
/* how the ranks of this process's latest n_gpus > 1 run reached each other: 0 no such run yet, 1 peer copies, 2 host-staged copies for at
 * least one pair of GPUs (hipDeviceCanAccessPeer refused, or W2RAP_TEST_NO_PEER=1) */
int  w2rap_step2_last_peer_mode(void);

/* w2rap_step2_run (and the one-shot entry points of Steps 1 and 3 and the GFA dump) keep their context -- streams and a pool of device
 * blocks -- in a process-wide cache between calls; this destroys the idle ones and hands their device memory back to the driver.
 * Returns the number of contexts destroyed.  W2RAP_NO_CTX_CACHE=1 in the environment disables the cache. */
int  w2rap_step2_trim_cached(void);
/* a context from that cache (or a new one) / back into it; what the one-shot entry points do internally */
struct w2rap_step2_ctx* w2rap_step2_acquire(int device, char* err, size_t errlen);
void w2rap_step2_release(struct w2rap_step2_ctx*);

/* ---- staged entry points (same work, phase by phase; used by bench.py, the tests and
 *      the multi-GPU host logic).  Order: create -> set_reads -> count_kmers ->
 *      build_graph -> path_reads -> fetch -> destroy. ------------------------------- */
typedef struct w2rap_step2_ctx w2rap_step2_ctx;

w2rap_step2_ctx* w2rap_step2_create(int device, char* err, size_t errlen);
void        w2rap_step2_destroy(w2rap_step2_ctx*);
const char* w2rap_step2_last_error(const w2rap_step2_ctx*);
int w2rap_step2_set_reads(w2rap_step2_ctx*, const w2rap_reads* reads);
/* a1-a5: quality windows, canonical 60-mers + context, count, min_freq filter, histogram,
 * lookup table over the solid k-mers.  Fills hist/M/D/S of *stats if non-NULL. */
int w2rap_step2_count_kmers(w2rap_step2_ctx*, uint32_t min_qual, uint32_t min_freq, w2rap_step2_out* stats);
/* the same with the counting phase in n_passes hash-range passes over the reads (0 = chosen from the free HBM): every pass cuts the
 * reads again and keeps only the super-k-mer records of ITS bucket range (MapReduceEngine.h:288-299: "keys of other passes are dropped
 * and re-mapped later"), so the record buffer is 1/n_passes of the whole; the results are identical */
int w2rap_step2_count_kmers_passes(w2rap_step2_ctx*, uint32_t min_qual, uint32_t min_freq, uint32_t n_passes, w2rap_step2_out* stats);
/* a6-a8: adjacency prune, unipaths, edge order (hint or canonical), vertices + adjacency.
 * ONE call per count: the list ranking rewrites the prune's neighbour links in place and the phase hands them back, so a second
 * build_graph on the same count (another edge order, say) returns W2RAP_E_STATE -- count again (count_kmers / dict_end) first. */
int w2rap_step2_build_graph(w2rap_step2_ctx*, const w2rap_edge_hint* hint);
/* a9-a12: seed pathing, heuristics, quality-scored extension, FixPaths */
int w2rap_step2_path_reads(w2rap_step2_ctx*);
/* copies graph + paths + statistics to host arrays */
int w2rap_step2_fetch(w2rap_step2_ctx*, w2rap_step2_out* out);
/* the HIP stream all kernels of this context are launched on (hipStream_t as void*) */
void* w2rap_step2_stream(w2rap_step2_ctx*);

/* A context recycles its device buffers between runs; trim returns the idle ones to the driver. */
int w2rap_step2_trim(w2rap_step2_ctx*);

/* measurement aid (bench.py): the rate of a plain device-to-device copy kernel (16 B per lane) on this GPU, GB/s counting
 * bytes read + bytes written */
int w2rap_step2_copy_bench(w2rap_step2_ctx*, uint64_t nbytes, uint32_t reps, double* gb_per_s);
/* dst[0, nbytes) = src[0, nbytes) on this context's device, by the library's own copy kernel, complete on return: what a host layer uses
 * for the part of an exchange that stays on the rank (a rank's own share of an all-to-all, the "exchanges" of a world-1 run) */
int w2rap_step2_device_copy(w2rap_step2_ctx*, void* dst, const void* src, uint64_t nbytes);
/* which form of the copy kernel the last copy_bench found fastest: 0 grid-stride, 1 non-temporal loads / stores, 2 a contiguous stretch per block */
int w2rap_step2_copy_bench_form(void);

/* per-kernel device time, measured with hipEvents on the context's stream.  Writes
 * "kernel_name total_ms launches\n" lines into buf; returns the bytes needed. */
int    w2rap_step2_set_profiling(w2rap_step2_ctx*, int on);
size_t w2rap_step2_profile(w2rap_step2_ctx*, char* buf, size_t len, int reset);

/* the sizes of what the context holds, without fetching it: k-mer instances, distinct, solid; unipaths, edge objects, vertices
 * (after build_graph); reads pathed, path elements (after path_reads) */
int w2rap_step2_counts(w2rap_step2_ctx*, uint64_t out[8]);

/* ---- stage-level read-back for the parity tests ---------------------------------- */
int w2rap_step2_get_good_len(w2rap_step2_ctx*, uint16_t* out /* [n_reads] */);
/* solid k-mer table in device order (unsorted): hi/lo = bases 0..29 / 30..59 as 60-bit
 * words (base 0 most significant), count = min(255, occurrences), ctx = (pred<<4)|succ.
 * After build_graph ctx is the pruned context and edge/off the unipath placement. */
int w2rap_step2_get_table(w2rap_step2_ctx*, uint64_t* hi, uint64_t* lo, uint8_t* count, uint8_t* ctx,
                          int32_t* edge, uint32_t* off /* each [S] or NULL */);

/* ---- multi-GPU building blocks -------------------------------------------
 * Reads are sharded by rank; every k-mer bucket has one owner rank (buckets are split into
 * `n_parts` equal contiguous ranges).  count_kmers == quality_windows + partition +
 * count_records + set_solid on one rank; between the steps the host code exchanges
 *   - the super-k-mer records and their per-bucket counts (all_to_all_v over RCCL/xGMI),
 *   - the solid k-mers of every owner (all_gather_v),
 * using the device pointers exposed here.  A super-k-mer record is 32 B (byte 0: bits 5:0
 * k-mers-1, bit 6/7 left/right flank valid; from bit 8: 2-bit bases, LSB first -- up to 63 k-mers and their two flank bases);
 * w2rap_step2_record_bytes() says so (36 in a -DW2RAP_REC36 build). */
int      w2rap_step2_quality_windows(w2rap_step2_ctx*, uint32_t min_qual, uint64_t* n_kmers /* this rank's M */);
uint32_t w2rap_step2_default_buckets(uint64_t total_kmers, uint32_t multiple_of);
uint32_t w2rap_step2_record_bytes(void);          /* bytes of one super-k-mer record (32) */
/* extract + scatter this rank's reads into n_buckets buckets; recs_per_part[n_parts] / kmers_per_part[n_parts]
 * (either may be NULL) = records / k-mer instances destined to each owner */
int w2rap_step2_partition(w2rap_step2_ctx*, uint32_t n_buckets, uint32_t n_parts, uint64_t* recs_per_part,
                          uint64_t* kmers_per_part);
/* the same for ONE HASH-RANGE PASS (MapReduceEngine.h:286-299: "keys of other passes are dropped and re-mapped later"; SURVEY.md 8e "if HBM
 * is short"): only the records of buckets [first_bucket, end_bucket) of n_buckets are kept, the n_parts owners divide that range, bucket
 * numbers in the outputs are relative to first_bucket.  On the owner's side w2rap_step2_count_pass(pass, n_passes) before the pass's
 * count_records / count_records_begin makes a later pass append to the solid k-mers, chunks, counters and histogram of the earlier ones
 * (pass the job's estimate of the owner's k-mer instances over ALL passes as total_kmers of pass 0: it sizes the solid arrays). */
int w2rap_step2_partition_range(w2rap_step2_ctx*, uint32_t n_buckets, uint32_t first_bucket, uint32_t end_bucket, uint32_t n_parts,
                                uint64_t* recs_per_part, uint64_t* kmers_per_part);
int w2rap_step2_count_pass(w2rap_step2_ctx*, uint32_t pass, uint32_t n_passes);
/* device pointers: records grouped by bucket (w2rap_step2_record_bytes() each), u32 records-per-bucket [n_buckets] */
int w2rap_step2_partition_buffers(w2rap_step2_ctx*, void** d_records, void** d_bucket_counts, uint64_t* n_records);
/* count n_local_buckets buckets whose records arrive as n_segments bucket-grouped segments laid back to
 * back in d_records; d_counts[s*n_local_buckets + b] (u32) = records of bucket b in segment s.
 * total_kmers = k-mer instances in those records (bounds the solid set: S <= total_kmers / min_freq).
 * Fills hist/D/S of *stats. */
int w2rap_step2_count_records(w2rap_step2_ctx*, uint32_t min_freq, uint32_t n_local_buckets, uint32_t n_segments,
                              const void* d_records, const void* d_counts, uint64_t total_kmers, w2rap_step2_out* stats);
/* The same in slices, so that the exchange of a slice's solid k-mers overlaps the counting of the next one:
 * count_records_begin plans n_slices (<= 16; fewer for tiny inputs, see count_records_slices) consecutive bucket ranges
 * (count_records_bounds; equal ranges, in deferred mode a first one of half the size) and, unless `deferred`, launches them
 * all and returns at once; count_records_slice(k) blocks until
 * slice k is complete and reports how many solid k-mers / chunks slices 0..k have appended to the arrays of solid_buffers /
 * chunk_buffers; count_records_end == the rest of count_records.  With `deferred` only d_counts must be final at begin: the
 * caller launches slice k with count_records_launch(k) (in order) once the records of ITS buckets have arrived in
 * d_records, so that the record exchange of slice k+1 overlaps the counting of slice k. */
int w2rap_step2_count_records_begin(w2rap_step2_ctx*, uint32_t min_freq, uint32_t n_local_buckets, uint32_t n_segments,
                                    const void* d_records, const void* d_counts, uint64_t total_kmers, uint32_t n_slices,
                                    int deferred);
int w2rap_step2_count_records_slices(w2rap_step2_ctx*);
int w2rap_step2_count_records_bounds(w2rap_step2_ctx*, uint32_t k, uint32_t* first_bucket, uint32_t* end_bucket);
int w2rap_step2_count_records_launch(w2rap_step2_ctx*, uint32_t k);
int w2rap_step2_count_records_slice(w2rap_step2_ctx*, uint32_t k, uint64_t* n_solid, uint64_t* n_chunks);
int w2rap_step2_count_records_end(w2rap_step2_ctx*, w2rap_step2_out* stats);
/* device pointers of this rank's solid k-mers: hi, lo (u64 each), cc (u32: count | ctx<<8) */
int w2rap_step2_solid_buffers(w2rap_step2_ctx*, void** d_hi, void** d_lo, void** d_cc, uint64_t* n);
/* install the gathered solid set (device arrays are copied) and build the lookup table + pruned contexts;
 * M, D, hist101 are the job-wide statistics to report */
int w2rap_step2_set_solid(w2rap_step2_ctx*, const void* d_hi, const void* d_lo, const void* d_cc, uint64_t n,
                          uint64_t M, uint64_t D, const uint64_t* hist101);
/* The solid k-mers of one minimizer bucket lie contiguously in an owner's output ("chunks": first k-mer, count; device
 * pointers u64 / u32).  Handing the chunk list of the gathered dictionary (starts shifted by every owner's offset) to
 * set_solid_chunked lets the adjacency prune find most neighbours bucket-locally in LDS; set_solid == no chunk list. */
int w2rap_step2_chunk_buffers(w2rap_step2_ctx*, void** d_chunk_start, void** d_chunk_count, uint64_t* n_chunks);
int w2rap_step2_set_solid_chunked(w2rap_step2_ctx*, const void* d_hi, const void* d_lo, const void* d_cc, uint64_t n,
                                  uint64_t M, uint64_t D, const uint64_t* hist101,
                                  const void* d_chunk_start, const void* d_chunk_count, uint64_t n_chunks);

/* set_solid in pieces: dict_begin reserves room for the job-wide dictionary (and clears its lookup table on the library's
 * side stream), every dict_append copies one gathered block of solid k-mers (+ its chunk list, starts relative to the
 * block) behind the earlier ones and inserts it into the table on the side stream -- the caller's arrays must stay alive
 * until dict_end --, dict_end == the rest of set_solid (adjacency prune).  Exceeding the capacity is W2RAP_E_LIMIT; the
 * caller then falls back to dict_abort + set_solid. */
int w2rap_step2_dict_begin(w2rap_step2_ctx*, uint64_t kmer_capacity, uint64_t chunk_capacity);
int w2rap_step2_dict_append(w2rap_step2_ctx*, const void* d_hi, const void* d_lo, const void* d_cc, uint64_t n,
                            const void* d_chunk_start, const void* d_chunk_count, uint64_t n_chunks);
/* dict_append for a SLICE of an owner's solid arrays given in place (n k-mers from d_hi on; peer memory is fine): the chunk starts handed
 * over are positions in the owner's WHOLE array, the slice begins at its k-mer `chunk_bias` */
int w2rap_step2_dict_append_slice(w2rap_step2_ctx*, const void* d_hi, const void* d_lo, const void* d_cc, uint64_t n,
                                  const void* d_chunk_start, const void* d_chunk_count, uint64_t n_chunks, uint64_t chunk_bias);
int w2rap_step2_dict_end(w2rap_step2_ctx*, uint64_t M, uint64_t D, const uint64_t* hist101);
int w2rap_step2_dict_abort(w2rap_step2_ctx*);

/* ---- the dictionary, the adjacency prune and the unipath phase SHARDED by bucket owner ------------------
 * The reference's dictionary has no ceiling (new BRQ_Dict(kmers.size()), BuildReadQGraph.cc:1092) and its prune and unipath walks run over
 * all of it (kmers/ReadPather.h:317-346, BuildReadQGraph.cc:314-339).  Instead of gathering the solid k-mers of every owner on every
 * GPU (set_solid / dict_*), each rank keeps the solid k-mers of ITS buckets -- what count_records left in its context -- and the
 * phases that follow run as a state machine between exchanges:
 *     shard_begin(rank, world, solid k-mers of every rank, bucket geometry, job statistics, hint)
 *     loop:  shard_next(&x)            -- computes up to the next exchange and describes it in x; x.op == W2RAP_X_DONE: the graph is built
 *            the host layer performs x (RCCL in dist.py, peer copies in w2rap_step2_run): all counts are in ELEMENTS of x.elem_bytes
 *              ALLTOALL       x.send holds send_count[r] elements for rank r, rank blocks back to back; exchange the counts, ask
 *                             shard_recv(counts received, &buffer) for room, deliver rank r's block at offset sum(counts[0..r))
 *              ALLGATHER      every rank contributes send_count[0] elements; shard_recv(every rank's count, &buffer); blocks in rank order
 *              ALLGATHER_HOST one 8-byte word per rank in HOST memory (x.send); hand all world words to shard_host_words()
 *              ALLREDUCE_U8 / ALLREDUCE_U32   sum of send_count[0] elements in place on x.send (device)
 * Afterwards the context is as after build_graph (the graph identical on every rank; path_reads then paths THIS rank's reads against the
 * minimizer-sampled index over the edge sequences), except that get_table covers the rank's own k-mers only. */
enum { W2RAP_X_DONE = 0, W2RAP_X_ALLTOALL = 1, W2RAP_X_ALLGATHER = 2, W2RAP_X_ALLGATHER_HOST = 3, W2RAP_X_ALLREDUCE_U8 = 4, W2RAP_X_ALLREDUCE_U32 = 5 };
typedef struct w2rap_xchg {
    int32_t  op;                 /* W2RAP_X_* */
    uint32_t elem_bytes;
    void*    send;               /* device memory (host memory for ALLGATHER_HOST) */
    uint64_t send_count[64];
} w2rap_xchg;
int w2rap_step2_shard_begin(w2rap_step2_ctx*, uint32_t rank, uint32_t world, const uint64_t* solid_per_rank /* [world] */,
                            uint32_t n_buckets, uint32_t n_passes, uint64_t M, uint64_t D, const uint64_t* hist101,
                            const w2rap_edge_hint* hint /* host memory, alive until DONE; or NULL */);
/* optional, between count_records_slice(k) and the next one: insert the solid k-mers counted so far (n_solid, as count_records_slice reports)
 * into this owner's own dictionary on the library's side stream, under the counting of the next slice; expected_total lays the table out
 * at the first call (the caller's extrapolation from the first slice).  shard_begin uses the table if it is complete, rebuilds it otherwise. */
int w2rap_step2_local_dict_slice(w2rap_step2_ctx*, uint64_t n_solid, uint64_t expected_total);
int w2rap_step2_shard_next(w2rap_step2_ctx*, w2rap_xchg* x);
int w2rap_step2_shard_recv(w2rap_step2_ctx*, const uint64_t* recv_count /* [world] */, uint32_t elem_bytes, void** d_recv);
int w2rap_step2_shard_host_words(w2rap_step2_ctx*, const uint64_t* words /* [world] */);
/* out: this rank's solid k-mers, the job's, this rank's chain segments, the job's, unipaths, edge bases, index entries, phase */
int w2rap_step2_shard_info(w2rap_step2_ctx*, uint64_t out[8]);
/* test aid: the library's own device-wide primitives (scans, maximum, stable radix sort of pairs; csrc/step2_prims.hip) on n pseudo-random
 * elements against host-side references; 0 = all equal, k > 0 = check k failed, < 0 = could not run */
int w2rap_step2_selftest_prims(w2rap_step2_ctx*, uint64_t n, uint64_t seed, int key_bits);
/* device bytes the context holds at the moment (live blocks of its pool): what the per-rank share of the dictionary is measured by */
uint64_t w2rap_step2_device_bytes(w2rap_step2_ctx*);
/* ... and their maximum since the context was created or since the last call with reset != 0 (which restarts the maximum at the current
 * value): the measured memory peak of a phase -- what the "fits 288 GB" sizing of BASELINE configs[4] is checked by.
 * Memory the CALLER owns (device-resident reads handed over with W2RAP_MEM_DEVICE) is not in it. */
uint64_t w2rap_step2_device_peak_bytes(w2rap_step2_ctx*, int reset);

#ifdef __cplusplus
}
#endif
#endif /* W2RAP_STEP2_H_ */
