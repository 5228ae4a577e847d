/* w2rap_step1.h -- C ABI of the MI355X-native replacement for w2rap-contigger's Step 1 ("Reading input files") for a pair of
 * fastq files.  Exported by w2rap_contigger_amd/libw2rap_step2.so.
 *
 * Drop-in boundary.  w2rap_step1_run replaces
 *     ExtractReads(read_files, out_dir, subsam_names, subsam_starts, &bases, &quals);      // src/modules/w2rap-contigger.cc:308
 * for `-r r1.fastq,r2.fastq` (one frag library, frac = 1: src/paths/long/large/ExtractReads.cc:350-474, the paired-fastq branch; with
 * W2RAP_STEP1_INTERLEAVED :481-568, one fastq file with alternating mates), and
 * its outputs are the flattened contents of `bases` / `quals`, i.e. what bases.WriteAll / quals.WriteAll put into
 * frag_reads_orig.fastb / .qualp (w2rap-contigger.cc:315-316) -- exactly the arrays w2rap_reads (w2rap_step2.h) takes.
 *   * four lines per record, both files in lock step; 'N' -> 'A'; bases ACGTacgt (Base::char2Val, src/dna/Bases.h:226); q = char - 33;
 *     mates interleaved R1, R2;
 *   * PQVec bytes as PQVecEncoder writes them (src/feudal/PQVec.cc:17-127) -- in effect one 3-byte block per run of equal qualities,
 *     runs cut at 255 (the encoder's log2 lookup table, src/math/PowerOf2.h:33-43, makes every mixed block look 58..63 bits wide);
 *   * the reference's fatal conditions (different record counts, incomplete record, base/quality length mismatch, a quality above 63,
 *     a character that is not a base) return W2RAP_E_ARG with the reference's wording in `err`.
 * Input is the (decompressed) TEXT of the two files in host memory; all parsing and encoding runs in HIP kernels (no CPU fallback).
 */
#ifndef W2RAP_STEP1_H_
#define W2RAP_STEP1_H_

#include "w2rap_step2.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct w2rap_step1_in {
    const char* fastq1; uint64_t len1;      /* text of the first file  (reads /1) */
    const char* fastq2; uint64_t len2;      /* text of the second file (reads /2) */
    int32_t mem;                            /* W2RAP_MEM_HOST: the text is copied to the GPU; W2RAP_MEM_DEVICE: device pointers, read in place */
} w2rap_step1_in;

#define W2RAP_STEP1_NO_PQ    1u             /* skip the PQVec encoding (Step 2 follows in-process and takes raw qualities) */
#define W2RAP_STEP1_NO_FETCH 2u             /* compute everything, copy only the counters back (timing runs) */
#define W2RAP_STEP1_INTERLEAVED 4u          /* fastq1 alone holds both mates, alternating (ExtractReads.cc:481-568, the "unpaired" fastq branch:
                                               what the reference does with a fastq file whose first read name no other file shares); an odd
                                               number of records is fatal there and W2RAP_E_ARG here; fastq2 is ignored */

typedef struct w2rap_step1_params {
    int32_t  device;
    uint32_t flags;
} w2rap_step1_params;

/* library-allocated HOST memory; free with w2rap_step1_free.  The first five arrays are a w2rap_reads in its "raw qualities" form, the
 * last two its PQVec form. */
typedef struct w2rap_step1_out {
    uint64_t n_reads;
    uint8_t*  bases_packed;                 /* read r at [base_byte_off[r], base_byte_off[r+1]), base i at bits 2*(i%4) of byte i/4 */
    uint64_t* base_byte_off;                /* [n_reads+1] */
    uint32_t* read_len;                     /* [n_reads] */
    uint8_t*  quals;                        /* one byte per base */
    uint64_t* qual_off;                     /* [n_reads+1] */
    uint8_t*  pq;                           /* PQVec byte strings (NULL with W2RAP_STEP1_NO_PQ) */
    uint64_t* pq_off;                       /* [n_reads+1] */
    uint64_t n_bases;                       /* = qual_off[n_reads] */
    uint64_t n_packed_bytes, n_pq_bytes;    /* = base_byte_off[n_reads], pq_off[n_reads] */
    float ms_upload, ms_index, ms_encode;   /* host->device copy of the text; device time of the line index; of validation + packing + PQVec */
} w2rap_step1_out;

int  w2rap_step1_run(const w2rap_step1_in* in, const w2rap_step1_params* params, w2rap_step1_out* out, char* err, size_t errlen);
void w2rap_step1_free(w2rap_step1_out* out);

/* Step 1 straight into Step 2 in one process -- the reference's default flow (w2rap-contigger.cc:308-346: ExtractReads fills `bases` and
 * `quals`, buildReadQGraph takes them): runs on `ctx`'s device and leaves the reads in HBM as that context's reads, as after
 * w2rap_step2_set_reads (count_kmers / build_graph / path_reads follow).  `out` receives the counters and, unless W2RAP_STEP1_NO_FETCH is
 * set, host copies for frag_reads_orig.fastb/.qualp. */
int  w2rap_step1_run_into_step2(w2rap_step2_ctx* ctx, const w2rap_step1_in* in, const w2rap_step1_params* params, w2rap_step1_out* out,
                                char* err, size_t errlen);

/* per-kernel device time of the last Step-1 run in this process: "kernel_name total_ms launches\n" lines; returns the bytes needed */
size_t w2rap_step1_profile(char* buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* W2RAP_STEP1_H_ */
