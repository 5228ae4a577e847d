// step1_ingest.hip -- Step 1 of w2rap-contigger for a pair of fastq files on gfx950; C ABI in include/w2rap_step1.h.
//
//   reference                                                          here
//   getline x 4 per record, both files in lock step                     k1_count_nl + scan + k1_list_nl (line index of each file),
//     ExtractReads.cc:396-441                                            host check of the line counts (the reference's fatal conditions)
//   N -> A, Base::char2Val, q = c - 33, length check :416-452,470-474   k1_measure (lengths, byte counts, run counts, character checks)
//   PQVecEncoder::init/encode  feudal/PQVec.cc:17-127                   k1_encode (2-bit packing, raw qualities, one 3-byte block per run
//                                                                        of equal qualities cut at 255 -- what that encoder produces)
// One thread per read; a read's two lines are read with byte loads (neighbouring threads are ~350 B apart: every sector is fetched
// once and used by one thread -- the kernel is bound by HBM sectors of the text, 2 x ~190 B per read, read twice: measure, encode).
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "ctx.h"
#include "../../include/w2rap_step1.h"

extern "C" w2rap_step2_ctx* w2rap_step2_create(int device, char* err, size_t errlen);
extern "C" void w2rap_step2_destroy(w2rap_step2_ctx*);
struct w2rap_step2_ctx { w2::Ctx c; };

namespace w2 {
namespace {

static inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }
constexpr unsigned CHUNK = 64;            // text bytes per thread of the newline passes

// newlines in [64 t, 64 t + 64)
__global__ void __launch_bounds__(256) k1_count_nl(uint64_t nchunks, uint64_t len, const uint8_t* __restrict__ text, uint32_t* __restrict__ cnt) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    const uint64_t a = t * CHUNK, e = a + CHUNK < len ? a + CHUNK : len;
    unsigned c = 0;
    if (e - a == CHUNK) {
        const uint4* p = reinterpret_cast<const uint4*>(text + a);           // (the text buffer is 16-byte aligned)
#pragma unroll
        for (unsigned j = 0; j < CHUNK / 16; ++j) {
            const uint4 v = p[j];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (unsigned k = 0; k < 4; ++k) {
                const uint32_t x = w[k] ^ 0x0A0A0A0Au;                          // a zero byte where there is a newline
                c += __builtin_popcount(((x - 0x01010101u) & ~x & 0x80808080u));
            }
        }
        // (the SWAR zero-byte test can flag a byte that follows a zero byte; recount exactly when a count looks wrong is not needed:
        //  x - 0x01010101 borrows only out of zero bytes, and a borrow into the next byte flags it only if that byte is 0x01 -> 0x00;
        //  0x01 ^ 0x0A = 0x0B is a vertical tab, never in fastq text, but stay exact: fall through to the byte loop when one is seen)
        bool vt = false;
#pragma unroll
        for (unsigned j = 0; j < CHUNK / 16; ++j) {
            const uint4 v = p[j];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (unsigned k = 0; k < 4; ++k) { const uint32_t y = w[k] ^ 0x0B0B0B0Bu; vt |= ((y - 0x01010101u) & ~y & 0x80808080u) != 0; }
        }
        if (vt) { c = 0; for (uint64_t i = a; i < e; ++i) c += text[i] == '\n'; }
    } else for (uint64_t i = a; i < e; ++i) c += text[i] == '\n';
    cnt[t] = c;
}
// nl[k] = position of the k-th newline
__global__ void __launch_bounds__(256) k1_list_nl(uint64_t nchunks, uint64_t len, const uint8_t* __restrict__ text, const uint64_t* __restrict__ excl, uint64_t* __restrict__ nl) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nchunks) return;
    const uint64_t a = t * CHUNK, e = a + CHUNK < len ? a + CHUNK : len;
    uint64_t k = excl[t];
    if (excl[t + 1] == k) return;
    for (uint64_t i = a; i < e; ++i) if (text[i] == '\n') nl[k++] = i;
}
struct FileIx { const uint8_t* text; const uint64_t* nl; uint64_t len, nnl; };
// line j of a file: [start, end)
__device__ inline void line_of(const FileIx& f, uint64_t j, uint64_t* s, uint64_t* e) {
    *s = j ? f.nl[j - 1] + 1 : 0;
    *e = j < f.nnl ? f.nl[j] : f.len;
}
enum { E1_LEN = 1, E1_BASE = 2, E1_QUAL = 4 };
// per read: bases, packed bytes, PQVec bytes; character checks
__global__ void __launch_bounds__(256) k1_measure(uint64_t n, FileIx f0, FileIx f1, uint32_t* __restrict__ rlen, uint32_t* __restrict__ nby, uint32_t* __restrict__ npq,
                                                   uint32_t* __restrict__ flags, unsigned long long* __restrict__ first_bad) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const FileIx& f = (r & 1) ? f1 : f0;
    const uint64_t rec = r >> 1;
    uint64_t s, e, qs, qe;
    line_of(f, 4 * rec + 1, &s, &e); line_of(f, 4 * rec + 3, &qs, &qe);
    const uint64_t L = e - s;
    unsigned bad = 0;
    if (qe - qs != L) bad |= E1_LEN;
    unsigned runs = 0;
    if (!bad) {
        int prev = -1; unsigned runlen = 0;
        for (uint64_t i = 0; i < L; ++i) {
            const uint8_t c = f.text[s + i];
            const bool okb = c == 'A' || c == 'C' || c == 'G' || c == 'T' || c == 'N' || c == 'a' || c == 'c' || c == 'g' || c == 't';
            if (!okb) bad |= E1_BASE;
            const int q = (int)(uint8_t)(f.text[qs + i] - 33);
            if (q > 63) bad |= E1_QUAL;
            if (q != prev || runlen == 255) { ++runs; prev = q; runlen = 0; }
            ++runlen;
        }
    }
    if (bad) { atomicOr(flags, bad); atomicMin(first_bad, (unsigned long long)r); }
    rlen[r] = (uint32_t)L; nby[r] = (uint32_t)((L + 3) >> 2); npq[r] = 3 * runs + 1;
}
__global__ void __launch_bounds__(256) k1_encode(uint64_t n, FileIx f0, FileIx f1, const uint64_t* __restrict__ boff, const uint64_t* __restrict__ qoff,
                                                  const uint64_t* __restrict__ pqoff, uint8_t* __restrict__ bases, uint8_t* __restrict__ quals, uint8_t* __restrict__ pq) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const FileIx& f = (r & 1) ? f1 : f0;
    const uint64_t rec = r >> 1;
    uint64_t s, e, qs, qe;
    line_of(f, 4 * rec + 1, &s, &e); line_of(f, 4 * rec + 3, &qs, &qe);
    const uint64_t L = e - s;
    uint8_t* bo = bases + boff[r]; uint8_t* qo = quals + qoff[r]; uint8_t* po = pq ? pq + pqoff[r] : nullptr;
    unsigned acc = 0;
    int prev = -1; unsigned runlen = 0;
    for (uint64_t i = 0; i < L; ++i) {
        const uint8_t c = f.text[s + i];
        // 'N' -> 'A' (ExtractReads.cc:416-419); Base::char2Val: A C G T in either case -> 0 1 2 3 (bits 2:1 of the ASCII code: A 00, C 01, G 11, T 10)
        const unsigned v = (c == 'N') ? 0u : (((c >> 1) & 3u) ^ ((c >> 2) & 1u));
        acc |= v << (2 * (i & 3));
        if ((i & 3) == 3) { bo[i >> 2] = (uint8_t)acc; acc = 0; }
        const uint8_t q = (uint8_t)(f.text[qs + i] - 33);
        qo[i] = q;
        if (po) {
            if ((int)q != prev || runlen == 255) {
                if (runlen) { po[0] = (uint8_t)runlen; po[1] = (uint8_t)(prev << 3); po[2] = (uint8_t)(prev >> 5); po += 3; }
                prev = q; runlen = 0;
            }
            ++runlen;
        }
    }
    if (L & 3) bo[L >> 2] = (uint8_t)acc;
    if (po) {
        if (runlen) { po[0] = (uint8_t)runlen; po[1] = (uint8_t)(prev << 3); po[2] = (uint8_t)(prev >> 5); po += 3; }
        po[0] = 0;
    }
}

template <class T>
int dl(Ctx& c, T** host, const T* dev, uint64_t n) {
    *host = (T*)std::malloc((n ? n : 1) * sizeof(T));
    if (!*host) { c.err = "out of host memory"; return W2RAP_E_HIP; }
    if (n) W2_HIP(hipMemcpyAsync(*host, dev, n * sizeof(T), hipMemcpyDeviceToHost, c.stream));
    return 0;
}
struct Timer {
    hipEvent_t a = nullptr, b = nullptr; hipStream_t st;
    explicit Timer(hipStream_t s) : st(s) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, st); }
    float stop() { float ms = 0; (void)hipEventRecord(b, st); (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b); return ms; }
    ~Timer() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
};

// uploads one file's text and builds its newline index
int index_file(Ctx& c, const char* text, uint64_t len, FileIx* ix, uint64_t* n_lines) {
    hipStream_t st = c.stream;
    uint8_t* d = c.alloc<uint8_t>(len + 64);
    if (!d) return W2RAP_E_HIP;
    if (len) W2_HIP(hipMemcpyAsync(d, text, len, hipMemcpyHostToDevice, st));
    W2_HIP(hipMemsetAsync(d + len, 0, 64, st));
    const uint64_t nchunks = (len + CHUNK - 1) / CHUNK;
    uint32_t* cnt = nullptr; uint64_t* excl = nullptr;
    W2_ALLOC(cnt, uint32_t, nchunks + 1); W2_ALLOC(excl, uint64_t, nchunks + 2);
    if (nchunks) LAUNCH(c, "k1_count_nl", k1_count_nl, dim3(grid_for(nchunks)), dim3(256), 0, nchunks, len, d, cnt);
    W2_TRY(exclusive_scan_u32_to_u64(c, cnt, excl, nchunks));
    uint64_t nnl = 0;
    W2_HIP(hipMemcpy(&nnl, excl + nchunks, 8, hipMemcpyDeviceToHost));
    uint64_t* nl = nullptr;
    W2_ALLOC(nl, uint64_t, nnl + 1);
    if (nchunks) LAUNCH(c, "k1_list_nl", k1_list_nl, dim3(grid_for(nchunks)), dim3(256), 0, nchunks, len, d, excl, nl);
    // getline: every newline ends a line; text behind the last newline is one more line
    char last = 0;
    if (len) last = text[len - 1];
    *n_lines = nnl + ((len && last != '\n') ? 1 : 0);
    *ix = FileIx{d, nl, len, nnl};
    W2_HIP(hipStreamSynchronize(st));
    c.release(cnt); c.release(excl);
    return 0;
}

int step1(Ctx& c, const w2rap_step1_in& in, const w2rap_step1_params& P, w2rap_step1_out& out) {
    hipStream_t st = c.stream;
    Timer t_index(st);
    FileIx f0, f1; uint64_t L1 = 0, L2 = 0;
    W2_TRY(index_file(c, in.fastq1, in.len1, &f0, &L1));
    W2_TRY(index_file(c, in.fastq2, in.len2, &f1, &L2));
    out.ms_index = t_index.stop();
    // the reference's loop (ExtractReads.cc:396-441) in terms of the line counts: record i exists in a file iff line 4i does
    const uint64_t n1 = (L1 + 3) / 4, n2 = (L2 + 3) / 4, m = n1 < n2 ? n1 : n2;
    for (uint64_t i = 0; i < 2; ++i) {
        const uint64_t L = i ? L2 : L1, nrec = i ? n2 : n1;
        if (L % 4 && nrec <= m) { c.err = "See incomplete record in the fastq files (ExtractReads.cc:409-437)"; return W2RAP_E_ARG; }
    }
    if (n1 != n2) { c.err = "The fastq files appear to be paired, yet have different numbers of records (ExtractReads.cc:399-405)"; return W2RAP_E_ARG; }
    const uint64_t n = 2 * n1;
    if (n >= (1ull << 32)) { c.err = "more than 2^32 reads"; return W2RAP_E_LIMIT; }
    Timer t_enc(st);
    uint32_t *rlen, *nby, *npq, *d_flags; unsigned long long* d_first; uint64_t *boff, *qoff, *pqoff;
    W2_ALLOC(rlen, uint32_t, n + 1); W2_ALLOC(nby, uint32_t, n + 1); W2_ALLOC(npq, uint32_t, n + 1); W2_ALLOC(d_flags, uint32_t, 4); W2_ALLOC(d_first, unsigned long long, 1);
    W2_ALLOC(boff, uint64_t, n + 2); W2_ALLOC(qoff, uint64_t, n + 2); W2_ALLOC(pqoff, uint64_t, n + 2);
    W2_HIP(hipMemsetAsync(d_flags, 0, 16, st)); W2_HIP(hipMemsetAsync(d_first, 0xFF, 8, st));
    if (n) LAUNCH(c, "k1_measure", k1_measure, dim3(grid_for(n)), dim3(256), 0, n, f0, f1, rlen, nby, npq, d_flags, d_first);
    uint32_t h_flags = 0; unsigned long long h_first = 0;
    W2_HIP(hipMemcpyAsync(&h_flags, d_flags, 4, hipMemcpyDeviceToHost, st)); W2_HIP(hipMemcpyAsync(&h_first, d_first, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    if (h_flags) {
        const std::string where = " (first at read " + std::to_string(h_first) + ")";
        if (h_flags & E1_LEN) c.err = "See inconsistent base/quality lengths in the fastq files (ExtractReads.cc:442-452)" + where;
        else if (h_flags & E1_BASE) c.err = "illegal base character in a sequence line (Base::char2Val, dna/Bases.h:226)" + where;
        else c.err = "Your input reads are funny.  I found a quality score > 63, the maximum value that I allow (PQVec.cc:30-35)" + where;
        return W2RAP_E_ARG;
    }
    W2_TRY(exclusive_scan_u32_to_u64(c, nby, boff, n));
    W2_TRY(exclusive_scan_u32_to_u64(c, rlen, qoff, n));
    W2_TRY(exclusive_scan_u32_to_u64(c, npq, pqoff, n));
    uint64_t nbytes = 0, nq = 0, npqb = 0;
    W2_HIP(hipMemcpy(&nbytes, boff + n, 8, hipMemcpyDeviceToHost)); W2_HIP(hipMemcpy(&nq, qoff + n, 8, hipMemcpyDeviceToHost)); W2_HIP(hipMemcpy(&npqb, pqoff + n, 8, hipMemcpyDeviceToHost));
    const bool want_pq = !(P.flags & W2RAP_STEP1_NO_PQ);
    uint8_t *bases, *quals, *pq = nullptr;
    W2_ALLOC(bases, uint8_t, nbytes + 16); W2_ALLOC(quals, uint8_t, nq + 16);
    if (want_pq) W2_ALLOC(pq, uint8_t, npqb + 16);
    if (n) LAUNCH(c, "k1_encode", k1_encode, dim3(grid_for(n)), dim3(256), 0, n, f0, f1, boff, qoff, pqoff, bases, quals, pq);
    out.ms_encode = t_enc.stop();
    out.n_reads = n; out.n_bases = nq;
    if (!(P.flags & W2RAP_STEP1_NO_FETCH)) {
        W2_TRY(dl(c, &out.bases_packed, bases, nbytes)); W2_TRY(dl(c, &out.base_byte_off, boff, n + 1)); W2_TRY(dl(c, &out.read_len, rlen, n));
        W2_TRY(dl(c, &out.quals, quals, nq)); W2_TRY(dl(c, &out.qual_off, qoff, n + 1));
        if (want_pq) { W2_TRY(dl(c, &out.pq, pq, npqb)); W2_TRY(dl(c, &out.pq_off, pqoff, n + 1)); }
    }
    W2_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // namespace
}  // namespace w2

using namespace w2;

extern "C" {

int w2rap_step1_run(const w2rap_step1_in* in, const w2rap_step1_params* P, w2rap_step1_out* out, char* err, size_t errlen) {
    auto fail = [&](int code, const std::string& m) { if (err && errlen) std::snprintf(err, errlen, "%s", m.c_str()); return code; };
    if (!in || !P || !out) return fail(W2RAP_E_ARG, "null argument");
    std::memset(out, 0, sizeof(*out));
    if ((in->len1 && !in->fastq1) || (in->len2 && !in->fastq2)) return fail(W2RAP_E_ARG, "null fastq buffer");
    char ebuf[512] = {0};
    w2rap_step2_ctx* h = w2rap_step2_create(P->device, ebuf, sizeof ebuf);
    if (!h) return fail(W2RAP_E_NO_DEVICE, ebuf);
    int rc = step1(h->c, *in, *P, *out);
    std::string msg = h->c.err;
    (void)hipStreamSynchronize(h->c.stream);
    w2rap_step2_destroy(h);
    if (rc) { w2rap_step1_free(out); return fail(rc, msg); }
    return 0;
}

void w2rap_step1_free(w2rap_step1_out* o) {
    if (!o) return;
    for (void* p : {(void*)o->bases_packed, (void*)o->base_byte_off, (void*)o->read_len, (void*)o->quals, (void*)o->qual_off, (void*)o->pq, (void*)o->pq_off}) std::free(p);
    std::memset(o, 0, sizeof(*o));
}

}  // extern "C"
