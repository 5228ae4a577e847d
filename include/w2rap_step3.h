/* w2rap_step3.h -- C ABI of the MI355X-native replacement for w2rap-contigger's Step 3,
 * "Repathing to second (large K) graph" (SURVEY.md 8f, rows N1 + N2).  Exported by the same
 * shared library as Step 2 (w2rap_contigger_amd/libw2rap_step2.so).
 *
 * Drop-in boundary.  w2rap_step3_run replaces exactly this block of the reference's main
 * (src/modules/w2rap-contigger.cc:359-378):
 *     vecbvec edges(hbv.Edges().begin(), hbv.Edges().end());
 *     hbv.Involution(inv);                                   // src/paths/HyperBasevector.cc:648-660
 *     FragDist(hbv, inv, paths, <prefix>.first.frags.dist);  // src/paths/long/large/GapToyTools3.cc:616-646 (the counts)
 *     RepathInMemory(hbv, edges, inv, paths, hbv.K(), large_K, hbvr, pathsr, True, True, extend_paths);
 *                                                            // src/paths/long/large/Repath.cc:23-251, declared Repath.h
 * Inputs are what BinaryReader::readFile(<prefix>.small_K.hbv) and LoadReadPathVec(<prefix>.small_K.paths) hold
 * (w2rap-contigger.cc:352-356); outputs are what BinaryWriter::writeFile(<prefix>.large_K.hbv, hbvr) and
 * WriteReadPathVec(pathsr, <prefix>.large_K.paths) serialise (:376-377), plus the fragment-size counts FragDist prints.
 *
 * Plain pointers and sizes; no C++/torch types; never throws; returns 0 or a W2RAP_E_* code (w2rap_step2.h) with a
 * message in `err`.  All compute runs in HIP kernels for gfx950; there is no CPU fallback (W2RAP_E_NO_DEVICE).
 *
 * Edge numbering of the large-K graph: the reference's is arbitrary (BigKEdgeBuilder::addEdge takes ids under a spin lock
 * while a hash set is walked in parallel, src/kmers/BigKPather.cc:275-292).  Default = lexicographic order of the unipath
 * sequences; `edge_order_hint` replays a given order (then .large_K.paths is byte-identical to the reference's and
 * .large_K.hbv identical up to the padding bits of each edge's last byte, which the reference leaves uninitialised).
 */
#ifndef W2RAP_STEP3_H_
#define W2RAP_STEP3_H_

#include "w2rap_step2.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- inputs (host memory): the small-K HyperBasevector's edge objects and the ReadPathVec -------------------------------- */
typedef struct w2rap_step3_in {
    int32_t  K;                      /* hbv.K(): 60 */
    uint64_t n_edge_objs;            /* hbv.EdgeObjectCount() */
    const uint8_t*  edge_packed;     /* each object ceil(len/4) bytes, base i at bits 2*(i%4) of byte i/4 (.hbv edges_ section) */
    const uint64_t* edge_byte_off;   /* [n_edge_objs+1] */
    const uint32_t* edge_len;        /* [n_edge_objs] bases */
    uint64_t n_paths;                /* paths.size() (reads; mates interleaved R1,R2 -- FragDist pairs 2i with 2i+1) */
    const int32_t*  path_offset;     /* [n_paths]  ReadPath::getOffset() */
    const uint64_t* path_off;        /* [n_paths+1] */
    const int32_t*  path_edges;      /* [path_off[n_paths]] edge-object ids */
    /* the vertices of the small-K graph, needed by params.extend_paths only (NULL otherwise): hbv.ToLeft / hbv.ToRight, Repath.cc:76-77 */
    uint64_t n_vertices;             /* hbv.N() */
    const int32_t*  vleft;           /* [n_edge_objs] the vertex an edge object leaves */
    const int32_t*  vright;          /* [n_edge_objs] the vertex it enters */
} w2rap_step3_in;

typedef struct w2rap_step3_params {
    uint32_t K2;                     /* -K / --large_k, default 200; even, K < K2 <= 640.  (The reference runs the values that are both in its command
                                        line's list, modules/w2rap-contigger.cc:60-62, and in BigK's, paths/long/LargeKDispatcher.h:22-27: 72 ... 640;
                                        any even value in the range works here) */
    int32_t  device;                 /* HIP device ordinal */
    int32_t  extend_paths;           /* --extend_paths (w2rap-contigger.cc:371 -> Repath.cc:72-96; experimental in the reference, default false):
                                        every unique place is also entered with the sole edge into its first vertex in front and the sole
                                        edge out of its last vertex behind.  Needs in->vleft / vright (W2RAP_E_ARG without them) */
    const w2rap_edge_hint* edge_order_hint;   /* NULL = canonical (lexicographic) order of the large-K unipaths */
    uint32_t flags;                  /* W2RAP_STEP3_NO_FETCH: compute everything, copy only the counters back (timing runs) */
    /* multi-GPU (reads sharded by rank, graph replicated): read paths of OTHER ranks that stand for their unique places.  They take part
       in the places / large-K graph exactly like local reads (so every rank builds the same graph) and produce no output path. */
    uint64_t n_extra_paths;
    const uint64_t* extra_path_off;  /* [n_extra_paths+1], host */
    const int32_t*  extra_path_edges;
} w2rap_step3_params;
#define W2RAP_STEP3_NO_FETCH    1u
#define W2RAP_STEP3_PLACES_ONLY 2u   /* stop behind the unique places; return one representative read path per unique place (place_path_*) */
#define W2RAP_STEP3_UNIQUE_KMERS 4u  /* the caller vouches that every K-mer of the input graph occurs ONCE in it, up to the involution -- the unipath graph
                                        buildReadQGraph builds (BuildReadQGraph.cc:287-339), the only input RepathInMemory ever gets (w2rap-contigger.cc:338-371).
                                        Then a K2-mer strictly inside an edge that no place of three or more edges holds in its middle has no second
                                        occurrence, and the dictionary (BigKPather.cc:40-55) leaves it out of the hashing: same result, ~78 % fewer keys.
                                        w2rap_step3_run_after_step2 sets it by itself (the graph is Step 2's own).  Without the flag every K2-mer is
                                        grouped by content, whatever the graph */

/* ---- outputs (library-allocated HOST memory; free with w2rap_step3_free) ------------------------------------------------- */
typedef struct w2rap_step3_out {
    int32_t  K2;
    /* hbv.Involution(inv) of the INPUT graph: inv[e] = the object holding e's reverse complement */
    int32_t*  inv;                   /* [in->n_edge_objs] */
    /* FragDist: count[len / 10] over read pairs on one edge of >= 10000 bases, 0 <= len < 1000 (GapToyTools3.cc:622-634) */
    uint64_t frag_count[100];
    /* hbvr: vertices, edge objects in id order, adjacency (DigraphTemplate.h:1829-1839 order) */
    uint64_t n_vertices;
    uint64_t n_edge_objs;
    uint8_t*  edge_packed;
    uint64_t* edge_byte_off;         /* [n_edge_objs+1] */
    uint32_t* edge_len;              /* [n_edge_objs] */
    int32_t*  vleft;                 /* [n_edge_objs] */
    int32_t*  vright;                /* [n_edge_objs] */
    uint64_t* from_off;              /* [n_vertices+1] */
    int32_t*  from_v;
    int32_t*  from_e;
    uint64_t* to_off;                /* [n_vertices+1] */
    int32_t*  to_v;
    int32_t*  to_e;
    int32_t*  inv2;                  /* [n_edge_objs] hb2.Involution (Repath.cc:137-138) */
    /* pathsr */
    uint64_t n_paths;
    int32_t*  path_offset;           /* [n_paths] */
    uint64_t* path_off;              /* [n_paths+1] */
    int32_t*  path_edges;
    /* what the reference prints (Repath.cc:36-72) and the sizes of the large-K dictionary */
    uint64_t n_reads_pathed, n_reads_multipathed;
    uint64_t n_places, n_unique_places;
    uint64_t n_place_bases;          /* sum of the lengths of `all` (Repath.cc:101) */
    uint64_t n_kmer_instances;       /* K2-mers of `all` */
    uint64_t n_kmers_distinct;       /* BigDict size */
    uint64_t n_unipaths;
    float ms_places, ms_dict, ms_graph, ms_paths;     /* device time of the phases, milliseconds */
    /* W2RAP_STEP3_PLACES_ONLY: a read path for every unique place of this rank's reads (what the other ranks get as extra_path_*) */
    uint64_t n_place_paths;
    uint64_t* place_path_off;        /* [n_place_paths+1] */
    int32_t*  place_path_edges;
    void* _owner;                    /* internal */
} w2rap_step3_out;

int  w2rap_step3_run(const w2rap_step3_in* in, const w2rap_step3_params* params, w2rap_step3_out* out, char* err, size_t errlen);
void w2rap_step3_free(w2rap_step3_out* out);

/* Step 3 straight behind Step 2 in one process -- the reference's default flow (w2rap-contigger.cc:338-371: buildReadQGraph, FixPaths,
 * Involution, FragDist, RepathInMemory on the same in-memory objects): `ctx` is a Step-2 context that has run build_graph and
 * path_reads; its graph and read paths are used where they lie in HBM, only the large-K result is copied to the host.  The
 * context is left intact (w2rap_step2_fetch still works afterwards). */
int  w2rap_step3_run_after_step2(w2rap_step2_ctx* ctx, const w2rap_step3_params* params, w2rap_step3_out* out, char* err, size_t errlen);

/* per-kernel device time of the last w2rap_step3_run in this process: "kernel_name total_ms launches\n" lines; returns the bytes needed */
size_t w2rap_step3_profile(char* buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* W2RAP_STEP3_H_ */
