/* w2rap_gfa.h -- C ABI of the MI355X-native replacement for w2rap-contigger's graph dump tool hbv2gfa without line finding (SURVEY.md 8f,
 * row N4).  Exported by w2rap_contigger_amd/libw2rap_step2.so.
 *
 * Drop-in boundary.  w2rap_gfa_dump replaces, in the reference's hbv2gfa main (src/modules/hbv2gfa.cc:50-99),
 *     hbv.Involution(inv); TestInvolution(hbv, inv);          // paths/HyperBasevector.cc:648-660
 *     the "=== Graph stats ===" block                        // hbv2gfa.cc:57-92: canonical size, N10..N90, NG10..NG90
 *     GFADump(out_prefix, hbv, inv, paths, 50, 10, false);   // src/GFADump.cc:228-286: <out_prefix>_raw.gfa
 * Input is what BinaryReader::readFile(<prefix>.hbv) holds (any K: .small_K.hbv and .large_K.hbv alike); output is the text of
 * <out_prefix>_raw.gfa, byte for byte, the numbers of the statistics block, and the involution.  `find_lines` (-l, default false; FindLines
 * + SortLines, paths/long/large/Lines.cc, serial graph surgery) is out of scope.  All compute runs in HIP kernels; no CPU fallback.
 */
#ifndef W2RAP_GFA_H_
#define W2RAP_GFA_H_

#include "w2rap_step2.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct w2rap_gfa_in {
    int32_t  K;
    uint64_t n_vertices, n_edge_objs;
    const uint8_t*  edge_packed;     /* each object ceil(len/4) bytes, base i at bits 2*(i%4) of byte i/4 (.hbv edges_ section) */
    const uint64_t* edge_byte_off;   /* [n_edge_objs+1] */
    const uint32_t* edge_len;        /* [n_edge_objs] bases */
    const uint64_t* from_off;        /* [n_vertices+1] */
    const int32_t*  from_e;          /* from_edge_obj_: edge objects leaving each vertex */
    const uint64_t* to_off;          /* [n_vertices+1] */
    const int32_t*  to_e;            /* to_edge_obj_: edge objects entering each vertex */
} w2rap_gfa_in;

#define W2RAP_GFA_STATS_ONLY 1u      /* --stats_only: no text */
#define W2RAP_GFA_NO_FETCH   2u      /* build the text, copy only the numbers back (timing runs) */

typedef struct w2rap_gfa_params {
    int32_t  device;
    uint32_t flags;
    uint64_t genome_size;            /* -g in bases (the tool multiplies its Kbp argument by 1000); 0 = no NGxx */
} w2rap_gfa_params;

/* library-allocated HOST memory; free with w2rap_gfa_free */
typedef struct w2rap_gfa_out {
    char*    gfa;                    /* the text of <out_prefix>_raw.gfa: S lines, then L lines */
    uint64_t gfa_len;
    uint64_t n_segments, n_links, segment_bytes;
    int32_t* inv;                    /* [n_edge_objs] */
    uint64_t canonical_size;         /* summed length of the objects that are not REV-canonical */
    uint64_t n_canonical;
    uint64_t nxx[9];                 /* N10 .. N90 */
    int64_t  ngxx[9];                /* NG10 .. NG90, -1 = "n/a" (only with genome_size) */
    float ms_involution, ms_dump;
} w2rap_gfa_out;

int  w2rap_gfa_dump(const w2rap_gfa_in* in, const w2rap_gfa_params* params, w2rap_gfa_out* out, char* err, size_t errlen);
void w2rap_gfa_free(w2rap_gfa_out* out);

/* per-kernel device time of the last w2rap_gfa_dump in this process: "kernel_name total_ms launches\n" lines; returns the bytes needed */
size_t w2rap_gfa_profile(char* buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* W2RAP_GFA_H_ */
