// gfa_dump.hip -- the reference's hbv2gfa (without line finding) on gfx950; C ABI in include/w2rap_gfa.h.
//
//   reference                                                        here
//   HyperBasevector::Involution  paths/HyperBasevector.cc:648-660    kg_hash (128-bit hash of every object and of its reverse complement),
//     (two sorts of all edge sequences, paired by rank)                two stable sorts, kg_pair (paired by rank, hashes compared),
//                                                                      kg_verify (every pair compared base by base, 32 at a time)
//   bvec::getCanonicalForm       dna/CanonicalForm.h:34-46           kg_form
//   graph statistics             modules/hbv2gfa.cc:57-92            sort of the canonical lengths on the device, the 9-step walk on the host
//   GFADump, find_lines = false  src/GFADump.cc:228-286              kg_seg_len + scan + kg_seg_write (16 output bytes per thread, line by
//                                                                      binary search), kg_links<false> + scan + kg_links<true>
// Sequence text dominates: 1 byte written per base, 1/4 byte read.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "ctx.h"
#include "../../include/w2rap_gfa.h"

extern "C" w2rap_step2_ctx* w2rap_step2_create(int device, char* err, size_t errlen);
extern "C" void w2rap_step2_destroy(w2rap_step2_ctx*);

namespace w2 {
namespace {

static inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }
// 32 bases from base position pos of a 2-bit stream (base p at bits 2(p&3) of byte p>>2); buffers are padded by 32 readable bytes
__device__ inline uint64_t stream64(const uint8_t* __restrict__ s, uint64_t pos) {
    const uint64_t b = pos >> 2; const unsigned sh = 2 * (unsigned)(pos & 3);
    uint64_t x = reinterpret_cast<const U64u*>(s + b)->v;
    if (sh) x = (x >> sh) | ((uint64_t)s[b + 8] << (64 - sh));
    return x;
}
__device__ inline unsigned stream1(const uint8_t* __restrict__ s, uint64_t pos) { return (s[pos >> 2] >> (2 * (pos & 3))) & 3u; }
// bases [32 j, 32 j + n) of an object's forward strand / of its reverse complement, base t of the word at bits 2t; n = min(32, L - 32 j)
__device__ inline uint64_t word_f(const uint8_t* s, uint64_t g0, uint32_t L, uint32_t j) {
    const uint32_t n = L - 32 * j < 32 ? L - 32 * j : 32;
    const uint64_t x = stream64(s, g0 + 32ull * j);
    return n < 32 ? x & ((1ull << (2 * n)) - 1) : x;
}
__device__ inline uint64_t word_r(const uint8_t* s, uint64_t g0, uint32_t L, uint32_t j) {
    const uint32_t n = L - 32 * j < 32 ? L - 32 * j : 32;
    if (n == 32) return rev2_64(~stream64(s, g0 + L - 32ull * (j + 1)));
    return rev2_64(~(stream64(s, g0) & ((1ull << (2 * n)) - 1))) >> (2 * (32 - n));       // the object's first n bases, reversed and complemented
}
__device__ inline uint64_t mix64(uint64_t h, uint64_t w) { h = (h ^ w) * 0x9E3779B97F4A7C15ull; return h ^ (h >> 29); }
template <class T>
__device__ inline uint64_t upper_index(const T* __restrict__ a, uint64_t n, T x) {       // largest i in [0, n) with a[i] <= x
    uint64_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint64_t m = (lo + hi) >> 1; if (a[m] <= x) lo = m; else hi = m; }
    return lo;
}
__device__ inline unsigned ndigits(uint32_t v) { unsigned d = 1; while (v >= 10) { v /= 10; ++d; } return d; }

__global__ void __launch_bounds__(256) kg_mul4(uint64_t n, const uint64_t* __restrict__ in, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 4;
}
__global__ void __launch_bounds__(256) kg_iota(uint64_t n, uint32_t* __restrict__ a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(256) kg_gather_u64(uint64_t n, const uint64_t* __restrict__ src, const uint32_t* __restrict__ perm, uint64_t* __restrict__ dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
// CanonicalForm of an object (CanonicalForm.h:34-46): odd length: the middle base decides; even: the object against its reverse complement
__global__ void __launch_bounds__(256) kg_form(uint64_t NO, const uint8_t* __restrict__ bits, const uint64_t* __restrict__ base0, const uint32_t* __restrict__ len,
                                                uint8_t* __restrict__ form) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= NO) return;
    const uint32_t L = len[o]; const uint64_t g0 = base0[o];
    if (L & 1) { form[o] = (stream1(bits, g0 + L / 2) & 2) ? 1 : 0; return; }
    unsigned f = 2;
    const uint32_t nw = (L + 31) / 32;
    for (uint32_t j = 0; j < nw; ++j) {                     // (the second half repeats the first with the roles swapped: harmless)
        const uint64_t a = rev2_64(word_f(bits, g0, L, j)), b = rev2_64(word_r(bits, g0, L, j));     // first base most significant
        if (a != b) { f = a < b ? 0 : 1; break; }
    }
    form[o] = (uint8_t)f;
}
// 128-bit hashes of an object's sequence and of its reverse complement
__global__ void __launch_bounds__(256) kg_hash(uint64_t NO, const uint8_t* __restrict__ bits, const uint64_t* __restrict__ base0, const uint32_t* __restrict__ len,
                                                uint64_t* __restrict__ f_hi, uint64_t* __restrict__ f_lo, uint64_t* __restrict__ r_hi, uint64_t* __restrict__ r_lo) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= NO) return;
    const uint32_t L = len[o]; const uint64_t g0 = base0[o];
    uint64_t a1 = L, a2 = 0x243F6A8885A308D3ull ^ L, b1 = a1, b2 = a2;
    const uint32_t nw = (L + 31) / 32;
    for (uint32_t j = 0; j < nw; ++j) {
        const uint64_t f = word_f(bits, g0, L, j), r = word_r(bits, g0, L, j);
        a1 = mix64(a1, f); a2 = mix64(a2 + j, f); b1 = mix64(b1, r); b2 = mix64(b2 + j, r);
    }
    f_hi[o] = a1; f_lo[o] = a2; r_hi[o] = b1; r_lo[o] = b2;
}
// rank i of the objects by hash pairs with rank i of the reverse complements by hash (Involution's x1[i] -> x2[i])
__global__ void __launch_bounds__(256) kg_pair(uint64_t NO, const uint32_t* __restrict__ p1, const uint32_t* __restrict__ p2, const uint64_t* __restrict__ f_hi,
                                                const uint64_t* __restrict__ f_lo, const uint64_t* __restrict__ r_hi, const uint64_t* __restrict__ r_lo,
                                                int32_t* __restrict__ inv, uint32_t* __restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NO) return;
    const uint32_t a = p1[i], b = p2[i];
    if (f_hi[a] != r_hi[b] || f_lo[a] != r_lo[b]) atomicOr(flags, 1u);
    inv[a] = (int32_t)b;
}
__global__ void __launch_bounds__(256) kg_wordcount(uint64_t NO, const uint32_t* __restrict__ len, uint32_t* __restrict__ nw) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o < NO) nw[o] = (len[o] + 31) / 32;
}
__global__ void __launch_bounds__(256) kg_verify(uint64_t nwords, uint64_t NO, const uint64_t* __restrict__ wordoff, const uint8_t* __restrict__ bits,
                                                  const uint64_t* __restrict__ base0, const uint32_t* __restrict__ len, const int32_t* __restrict__ inv, uint32_t* __restrict__ flags) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one 32-base word of one object
    if (w >= nwords) return;
    const uint64_t o = upper_index(wordoff, NO, w);
    const int32_t p = inv[o];
    const uint32_t L = len[o];
    if (len[p] != L) { atomicOr(flags, 2u); return; }
    const uint32_t j = (uint32_t)(w - wordoff[o]);
    if (word_f(bits, base0[o], L, j) != word_r(bits, base0[p], L, j)) atomicOr(flags, 2u);
}
// to_left / to_right of every object from the per-vertex lists
__global__ void __launch_bounds__(256) kg_ends(uint64_t NO, uint64_t NV, const uint64_t* __restrict__ off, const int32_t* __restrict__ e, int32_t* __restrict__ at,
                                                uint32_t* __restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NO) return;
    const int32_t x = e[i];
    if (x < 0 || (uint64_t)x >= NO) { atomicOr(flags, 4u); return; }
    at[x] = (int32_t)upper_index(off, NV, i);
}
// S lines: "S\tedge<id>\t<bases>\tCL:z:black\n" for the objects that are not REV
constexpr unsigned S_HEAD = 6, S_TAIL = 12;
__global__ void __launch_bounds__(256) kg_seg_len(uint64_t NO, const uint8_t* __restrict__ form, const uint32_t* __restrict__ len, uint64_t* __restrict__ bytes,
                                                   uint32_t* __restrict__ is_seg) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= NO) return;
    const bool s = form[o] != 1;
    is_seg[o] = s;
    bytes[o] = s ? S_HEAD + ndigits((uint32_t)o) + 1 + (uint64_t)len[o] + S_TAIL : 0;
}
__global__ void __launch_bounds__(256) kg_compact(uint64_t NO, const uint32_t* __restrict__ is_seg, const uint64_t* __restrict__ rank, const uint64_t* __restrict__ soff_all,
                                                   const uint32_t* __restrict__ len, uint32_t* __restrict__ cobj, uint64_t* __restrict__ soff, uint64_t* __restrict__ clen) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= NO || !is_seg[o]) return;
    const uint64_t k = rank[o];
    cobj[k] = (uint32_t)o; soff[k] = soff_all[o]; clen[k] = len[o];
}
__global__ void __launch_bounds__(256) kg_seg_write(uint64_t total, uint64_t NC, const uint32_t* __restrict__ cobj, const uint64_t* __restrict__ soff,
                                                     const uint8_t* __restrict__ bits, const uint64_t* __restrict__ base0, const uint32_t* __restrict__ len,
                                                     uint8_t* __restrict__ text) {
    const uint64_t p0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (p0 >= total) return;
    uint64_t k = upper_index(soff, NC, p0);
    uint32_t o = cobj[k], L = len[o], nd = ndigits(o);
    uint64_t within = p0 - soff[k], g0 = base0[o];
    const unsigned nb = total - p0 < 16 ? (unsigned)(total - p0) : 16u;
    uint32_t w[4] = {0, 0, 0, 0};
    const uint64_t seq0 = S_HEAD + nd + 1;
    if (nb == 16 && within >= seq0 && within + 16 <= seq0 + L) {           // 16 bases of one object: the common case
        const uint32_t x = (uint32_t)stream64(bits, g0 + (within - seq0));
#pragma unroll
        for (unsigned t = 0; t < 16; ++t) w[t >> 2] |= ((0x54474341u >> (8 * ((x >> (2 * t)) & 3u))) & 0xFFu) << (8 * (t & 3));
    } else {
        const char head[] = "S\tedge", tail[] = "\tCL:z:black\n";
#pragma unroll
        for (unsigned t = 0; t < 16; ++t) {
            if (t < nb) {
                const uint64_t line_len = S_HEAD + nd + 1 + (uint64_t)L + S_TAIL;
                if (within == line_len) { ++k; o = cobj[k]; L = len[o]; nd = ndigits(o); g0 = base0[o]; within = 0; }
                uint32_t ch;
                if (within < S_HEAD) ch = (uint8_t)head[within];
                else if (within < S_HEAD + nd) {
                    uint32_t v = o;
                    for (unsigned d = (unsigned)(S_HEAD + nd - 1 - within); d; --d) v /= 10;
                    ch = '0' + v % 10;
                } else if (within == S_HEAD + nd) ch = '\t';
                else if (within < S_HEAD + nd + 1 + (uint64_t)L) ch = (0x54474341u >> (8 * stream1(bits, g0 + (within - (S_HEAD + nd + 1))))) & 0xFFu;
                else ch = (uint8_t)tail[within - (S_HEAD + nd + 1 + (uint64_t)L)];
                w[t >> 2] |= ch << (8 * (t & 3));
                ++within;
            }
        }
    }
    if (nb == 16) *reinterpret_cast<uint4*>(text + p0) = make_uint4(w[0], w[1], w[2], w[3]);
    else for (unsigned t = 0; t < nb; ++t) text[p0 + t] = (uint8_t)(w[t >> 2] >> (8 * (t & 3)));
}
// L lines of object e (GFADump.cc:252-284): followers = the objects leaving e's right vertex and the inverses of those entering inv[e]'s
// left vertex, in ascending id without repeats, each named by its canonical object cn, kept when cn >= e; then the predecessors likewise.
struct Graph { const int32_t *to_left, *to_right, *inv; const uint8_t* form; const uint64_t *from_off, *to_off; const int32_t *from_e, *to_e; };
__device__ inline unsigned put_uint(uint8_t* p, uint32_t v) {
    const unsigned nd = ndigits(v);
    for (unsigned d = nd; d; --d) { p[d - 1] = (uint8_t)('0' + v % 10); v /= 10; }
    return nd;
}
template <bool WRITE>
__global__ void __launch_bounds__(256) kg_links(uint64_t NO, Graph g, const uint64_t* __restrict__ loff, uint32_t* __restrict__ lbytes, uint32_t* __restrict__ lcount,
                                                 uint8_t* __restrict__ text) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NO) return;
    if (g.form[e] == 1) { if (!WRITE) { lbytes[e] = 0; lcount[e] = 0; } return; }
    const int32_t ie = g.inv[e];
    uint8_t* out = WRITE ? text + loff[e] : nullptr;
    unsigned bytes = 0, count = 0;
    for (unsigned side = 0; side < 2; ++side) {
        // side 0: own list A = objects leaving to_right[e]; mirrored list B = objects entering to_left[inv e], each through inv
        // side 1: A = objects entering to_left[e]; B = objects leaving to_right[inv e], each through inv
        const uint64_t *offA = side ? g.to_off : g.from_off, *offB = side ? g.from_off : g.to_off;
        const int32_t *lstA = side ? g.to_e : g.from_e, *lstB = side ? g.from_e : g.to_e;
        const int32_t vA = side ? g.to_left[e] : g.to_right[e], vB = side ? g.to_right[ie] : g.to_left[ie];
        const uint64_t a0 = offA[vA], a1 = offA[vA + 1], b0 = offB[vB], b1 = offB[vB + 1];
        int64_t last = -1;
        for (;;) {
            int64_t m = INT64_MAX;
            for (uint64_t i = a0; i < a1; ++i) { const int64_t x = lstA[i]; if (x > last && x < m) m = x; }
            for (uint64_t i = b0; i < b1; ++i) { const int64_t x = g.inv[lstB[i]]; if (x > last && x < m) m = x; }
            if (m == INT64_MAX) break;
            last = m;
            const bool same = g.form[m] != 1;
            const uint32_t cn = same ? (uint32_t)m : (uint32_t)g.inv[m];
            if (cn < e) continue;
            const unsigned n = 6 + ndigits((uint32_t)e) + 7 + ndigits(cn) + 6;
            if (WRITE) {
                uint8_t* p = out + bytes;
                const char h[] = "L\tedge"; for (unsigned t = 0; t < 6; ++t) p[t] = h[t];
                p += 6; p += put_uint(p, (uint32_t)e);
                *p++ = '\t'; *p++ = side ? '-' : '+'; *p++ = '\t'; *p++ = 'e'; *p++ = 'd'; *p++ = 'g'; *p++ = 'e';
                p += put_uint(p, cn);
                *p++ = '\t'; *p++ = (same != (side == 1)) ? '+' : '-'; *p++ = '\t'; *p++ = '0'; *p++ = 'M'; *p++ = '\n';
            }
            bytes += n; ++count;
        }
    }
    if (!WRITE) { lbytes[e] = bytes; lcount[e] = count; }
}
__global__ void __launch_bounds__(256) kg_u32_to_u64(uint64_t n, const uint32_t* __restrict__ in, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// stable sort of perm by (hi, lo)
int sort128(Ctx& c, const uint64_t* hi, const uint64_t* lo, uint64_t n, uint32_t* perm, uint64_t* tmp) {
    LAUNCH(c, "kg_iota", kg_iota, dim3(grid_for(n)), dim3(256), 0, n, perm);
    W2_HIP(hipMemcpyAsync(tmp, lo, n * 8, hipMemcpyDeviceToDevice, c.stream));
    W2_TRY(sort_pairs_u64(c, tmp, perm, n, 0, 64));
    LAUNCH(c, "kg_gather_u64", kg_gather_u64, dim3(grid_for(n)), dim3(256), 0, n, hi, perm, tmp);
    W2_TRY(sort_pairs_u64(c, tmp, perm, n, 0, 64));
    return 0;
}

int gfa(Ctx& c, const w2rap_gfa_in& in, const w2rap_gfa_params& P, w2rap_gfa_out& out) {
    hipStream_t st = c.stream;
    const uint64_t NO = in.n_edge_objs, NV = in.n_vertices;
    const uint8_t* bits; const uint64_t* obyte; const uint32_t* len; const uint64_t *from_off, *to_off; const int32_t *from_e, *to_e;
    uint8_t* b0; uint64_t* b1; uint32_t* b2; uint64_t *b3, *b4; int32_t *b5, *b6;
    W2_TRY(up_pooled(c, &b0, in.edge_packed, NO ? in.edge_byte_off[NO] : 0, 32)); W2_TRY(up_pooled(c, &b1, in.edge_byte_off, NO + 1)); W2_TRY(up_pooled(c, &b2, in.edge_len, NO));
    W2_TRY(up_pooled(c, &b3, in.from_off, NV + 1)); W2_TRY(up_pooled(c, &b4, in.to_off, NV + 1)); W2_TRY(up_pooled(c, &b5, in.from_e, NO)); W2_TRY(up_pooled(c, &b6, in.to_e, NO));
    bits = b0; obyte = b1; len = b2; from_off = b3; to_off = b4; from_e = b5; to_e = b6;
    uint32_t* d_flags; W2_ALLOC(d_flags, uint32_t, 4);
    W2_HIP(hipMemsetAsync(d_flags, 0, 16, st));
    auto check = [&]() -> int {
        uint32_t f = 0;
        W2_HIP(hipMemcpyAsync(&f, d_flags, 4, hipMemcpyDeviceToHost, st)); W2_HIP(hipStreamSynchronize(st));
        if (f & 3) { c.err = "Involution: the edge objects do not pair up with their reverse complements (HyperBasevector.cc:648-660; TestInvolution would abort)"; return W2RAP_E_GRAPH; }
        if (f & 4) { c.err = "an adjacency list names an edge object that does not exist"; return W2RAP_E_ARG; }
        return 0;
    };
    Timer t_inv(st);
    uint64_t* base0; W2_ALLOC(base0, uint64_t, NO + 1);
    LAUNCH(c, "kg_mul4", kg_mul4, dim3(grid_for(NO + 1)), dim3(256), 0, NO + 1, obyte, base0);
    uint8_t* form; W2_ALLOC(form, uint8_t, NO + 1);
    int32_t *inv, *to_left, *to_right; W2_ALLOC(inv, int32_t, NO + 1); W2_ALLOC(to_left, int32_t, NO + 1); W2_ALLOC(to_right, int32_t, NO + 1);
    if (NO) {
        LAUNCH(c, "kg_form", kg_form, dim3(grid_for(NO)), dim3(256), 0, NO, bits, base0, len, form);
        uint64_t *f_hi, *f_lo, *r_hi, *r_lo, *tmp; uint32_t *p1, *p2;
        W2_ALLOC(f_hi, uint64_t, NO); W2_ALLOC(f_lo, uint64_t, NO); W2_ALLOC(r_hi, uint64_t, NO); W2_ALLOC(r_lo, uint64_t, NO); W2_ALLOC(tmp, uint64_t, NO);
        W2_ALLOC(p1, uint32_t, NO); W2_ALLOC(p2, uint32_t, NO);
        LAUNCH(c, "kg_hash", kg_hash, dim3(grid_for(NO)), dim3(256), 0, NO, bits, base0, len, f_hi, f_lo, r_hi, r_lo);
        W2_TRY(sort128(c, f_hi, f_lo, NO, p1, tmp)); W2_TRY(sort128(c, r_hi, r_lo, NO, p2, tmp));
        LAUNCH(c, "kg_pair", kg_pair, dim3(grid_for(NO)), dim3(256), 0, NO, p1, p2, f_hi, f_lo, r_hi, r_lo, inv, d_flags);
        W2_TRY(check());
        uint32_t* nw; uint64_t* wordoff; W2_ALLOC(nw, uint32_t, NO); W2_ALLOC(wordoff, uint64_t, NO + 1);
        LAUNCH(c, "kg_wordcount", kg_wordcount, dim3(grid_for(NO)), dim3(256), 0, NO, len, nw);
        W2_TRY(exclusive_scan_u32_to_u64(c, nw, wordoff, NO));
        uint64_t nwords = 0; W2_HIP(hipMemcpy(&nwords, wordoff + NO, 8, hipMemcpyDeviceToHost));
        if (nwords) LAUNCH(c, "kg_verify", kg_verify, dim3(grid_for(nwords)), dim3(256), 0, nwords, NO, wordoff, bits, base0, len, inv, d_flags);
        W2_HIP(hipMemsetAsync(to_left, 0xFF, NO * 4, st)); W2_HIP(hipMemsetAsync(to_right, 0xFF, NO * 4, st));
        LAUNCH(c, "kg_ends", kg_ends, dim3(grid_for(NO)), dim3(256), 0, NO, NV, from_off, from_e, to_left, d_flags);
        LAUNCH(c, "kg_ends", kg_ends, dim3(grid_for(NO)), dim3(256), 0, NO, NV, to_off, to_e, to_right, d_flags);
        W2_TRY(check());
        for (void* p : {(void*)f_hi, (void*)f_lo, (void*)r_hi, (void*)r_lo, (void*)tmp, (void*)p1, (void*)p2, (void*)nw, (void*)wordoff}) c.release(p);
    }
    out.ms_involution = t_inv.stop();
    Timer t_dump(st);
    // ---- S lines and the statistics
    uint64_t *sbytes, *soff_all, *rank64, *soff, *clen; uint32_t *is_seg, *cobj;
    W2_ALLOC(sbytes, uint64_t, NO + 1); W2_ALLOC(soff_all, uint64_t, NO + 2); W2_ALLOC(rank64, uint64_t, NO + 2); W2_ALLOC(is_seg, uint32_t, NO + 1);
    if (NO) LAUNCH(c, "kg_seg_len", kg_seg_len, dim3(grid_for(NO)), dim3(256), 0, NO, form, len, sbytes, is_seg);
    W2_TRY(exclusive_scan_u64(c, sbytes, soff_all, NO));
    W2_TRY(exclusive_scan_u32_to_u64(c, is_seg, rank64, NO));
    uint64_t S_total = 0, NC = 0;
    W2_HIP(hipMemcpy(&S_total, soff_all + NO, 8, hipMemcpyDeviceToHost)); W2_HIP(hipMemcpy(&NC, rank64 + NO, 8, hipMemcpyDeviceToHost));
    W2_ALLOC(cobj, uint32_t, NC + 1); W2_ALLOC(soff, uint64_t, NC + 1); W2_ALLOC(clen, uint64_t, NC + 1);
    if (NO) LAUNCH(c, "kg_compact", kg_compact, dim3(grid_for(NO)), dim3(256), 0, NO, is_seg, rank64, soff_all, len, cobj, soff, clen);
    out.n_canonical = NC; out.n_segments = NC; out.segment_bytes = S_total;
    {   // N10 .. N90 over the canonical lengths, longest first (hbv2gfa.cc:70-92)
        uint32_t* idx; W2_ALLOC(idx, uint32_t, NC + 1);
        if (NC) { LAUNCH(c, "kg_iota", kg_iota, dim3(grid_for(NC)), dim3(256), 0, NC, idx); W2_TRY(sort_pairs_u64(c, clen, idx, NC, 0, 64)); }
        std::vector<uint64_t> sizes(NC);
        W2_HIP(hipStreamSynchronize(st));
        if (NC) W2_HIP(hipMemcpy(sizes.data(), clen, NC * 8, hipMemcpyDeviceToHost));
        uint64_t canonical = 0;
        for (uint64_t s : sizes) canonical += s;
        out.canonical_size = canonical;
        auto walk = [&](uint64_t denom, bool stop_at_end, auto&& put) {
            uint64_t k = 0; int64_t cs = 0;                                  // sizes[NC-1-k] = the k-th longest
            for (int i = 10, j = 0; i < 100; i += 10, ++j) {
                while ((double)(cs * 100.0) / denom < i && (!stop_at_end || k < NC)) { if (k >= NC) break; cs += sizes[NC - 1 - k]; ++k; }
                put(j, (stop_at_end && k == NC) ? -1 : (k ? (int64_t)sizes[NC - k] : 0));
            }
        };
        if (canonical) walk(canonical, false, [&](int j, int64_t v) { out.nxx[j] = (uint64_t)v; });
        if (P.genome_size && NC) walk(P.genome_size, true, [&](int j, int64_t v) { out.ngxx[j] = v; });
        c.release(idx);
    }
    uint8_t* text = nullptr; uint64_t total = 0, n_links = 0;
    if (!(P.flags & W2RAP_GFA_STATS_ONLY)) {
        // ---- L lines: sizes, offsets
        Graph g{to_left, to_right, inv, form, from_off, to_off, from_e, to_e};
        uint32_t *lbytes, *lcount; uint64_t *loff, *lcoff;
        W2_ALLOC(lbytes, uint32_t, NO + 1); W2_ALLOC(lcount, uint32_t, NO + 1); W2_ALLOC(loff, uint64_t, NO + 2); W2_ALLOC(lcoff, uint64_t, NO + 2);
        if (NO) LAUNCH(c, "kg_links_count", kg_links<false>, dim3(grid_for(NO)), dim3(256), 0, NO, g, loff, lbytes, lcount, text);
        W2_TRY(exclusive_scan_u32_to_u64(c, lbytes, loff, NO)); W2_TRY(exclusive_scan_u32_to_u64(c, lcount, lcoff, NO));
        uint64_t L_total = 0;
        W2_HIP(hipMemcpy(&L_total, loff + NO, 8, hipMemcpyDeviceToHost)); W2_HIP(hipMemcpy(&n_links, lcoff + NO, 8, hipMemcpyDeviceToHost));
        total = S_total + L_total;
        W2_ALLOC(text, uint8_t, total + 32);
        if (S_total) LAUNCH(c, "kg_seg_write", kg_seg_write, dim3(grid_for((S_total + 15) / 16)), dim3(256), 0, S_total, NC, cobj, soff, bits, base0, len, text);
        if (NO) LAUNCH(c, "kg_links_write", kg_links<true>, dim3(grid_for(NO)), dim3(256), 0, NO, g, loff, lbytes, lcount, text + S_total);
    }
    out.ms_dump = t_dump.stop();
    out.gfa_len = total; out.n_links = n_links;
    if (!(P.flags & W2RAP_GFA_NO_FETCH)) {
        if (text) { uint8_t* h = nullptr; W2_TRY(dl(c, &h, text, total)); out.gfa = (char*)h; }
        W2_TRY(dl(c, &out.inv, inv, NO));
    }
    W2_HIP(hipStreamSynchronize(st));
    return 0;
}

std::string g_profile_gfa;

}  // namespace
}  // namespace w2

using namespace w2;

extern "C" {

int w2rap_gfa_dump(const w2rap_gfa_in* in, const w2rap_gfa_params* P, w2rap_gfa_out* out, char* err, size_t errlen) {
    auto fail = [&](int code, const std::string& m) { if (err && errlen) std::snprintf(err, errlen, "%s", m.c_str()); return code; };
    if (!in || !P || !out) return fail(W2RAP_E_ARG, "null argument");
    std::memset(out, 0, sizeof(*out));
    if (in->n_edge_objs >= (1ull << 31) || in->n_vertices >= (1ull << 31)) return fail(W2RAP_E_LIMIT, "more than 2^31 edge objects or vertices");
    if ((in->n_edge_objs && (!in->edge_packed || !in->edge_byte_off || !in->edge_len || !in->from_e || !in->to_e)) || !in->from_off || !in->to_off)
        return fail(W2RAP_E_ARG, "null input array");
    if (in->from_off[in->n_vertices] != in->n_edge_objs || in->to_off[in->n_vertices] != in->n_edge_objs) return fail(W2RAP_E_ARG, "the adjacency lists do not hold every edge object once");
    for (uint64_t v = 0; v < in->n_vertices; ++v)
        if (in->from_off[v + 1] < in->from_off[v] || in->to_off[v + 1] < in->to_off[v]) return fail(W2RAP_E_ARG, "adjacency offsets are not ascending");
    for (uint64_t o = 0; o < in->n_edge_objs; ++o) {
        if (in->edge_len[o] == 0) return fail(W2RAP_E_ARG, "an empty edge object");
        if (in->edge_byte_off[o + 1] - in->edge_byte_off[o] != (in->edge_len[o] + 3ull) / 4) return fail(W2RAP_E_ARG, "edge_byte_off does not match edge_len");
    }
    char ebuf[512] = {0};
    w2rap_step2_ctx* h = w2rap_step2_acquire(P->device, ebuf, sizeof ebuf);
    if (!h) return fail(W2RAP_E_NO_DEVICE, ebuf);
    int rc = gfa(h->c, *in, *P, *out);
    std::string msg = h->c.err;
    (void)hipStreamSynchronize(h->c.stream);
    h->c.presolve();
    g_profile_gfa.clear();
    for (auto& s : h->c.prof_sums) { char line[256]; std::snprintf(line, sizeof line, "%s %.4f %llu\n", s.name.c_str(), s.ms, (unsigned long long)s.launches); g_profile_gfa += line; }
    if (rc) w2rap_step2_destroy(h); else w2rap_step2_release(h);     // (a failed context is not cached)
    if (rc) { w2rap_gfa_free(out); return fail(rc, msg); }
    return 0;
}

size_t w2rap_gfa_profile(char* buf, size_t len) {
    if (buf && len) std::snprintf(buf, len, "%s", g_profile_gfa.c_str());
    return g_profile_gfa.size() + 1;
}

void w2rap_gfa_free(w2rap_gfa_out* o) {
    if (!o) return;
    std::free(o->gfa); std::free(o->inv);
    std::memset(o, 0, sizeof(*o));
}

}  // extern "C"
