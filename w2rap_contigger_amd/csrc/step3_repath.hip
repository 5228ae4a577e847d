// step3_repath.hip -- Step 3 of w2rap-contigger ("Repathing to second (large K) graph") on gfx950; C ABI in include/w2rap_step3.h.
//
//   reference                                                     here
//   HyperBasevector::Involution  paths/HyperBasevector.cc:648-660  k3_obj_ends + sort + k3_inv_match / k3_inv_verify
//   FragDist                     paths/long/large/GapToyTools3.cc:616-634   inside k3_place_keys (the mates sit in neighbouring lanes)
//   RepathInMemory               paths/long/large/Repath.cc:23-251
//     places  :40-71                                               k3_place_keys, sort, k3_place_heads, k3_place_index
//     all     :101-123                                             k3_place_layout, k3_all_fill
//     LongReadsToPaths -> readsToHBV  kmers/BigKPather.cc:461-536
//       BigKMerizer::kmerize :40-55 (BigDict = HashSet by content)  k3_kmer_keys, sort, k3_group / k3_group_fix, k3_ids
//       BigKEdgeBuilder :110-310                                   k3_nbr, k3_links, run_ranking (step2_graph.hip), k3_mid, circles,
//                                                                  k3_heads, k3_edge_order / k3_edge_from_hint, k3_assign
//       buildHBVFromEdges  paths/long/HBVFromEdges.cc:76-154       k3_objs, k3_ends + sorts, k3_end_vertices, adjacency
//       Pather :311-405 + translation Repath.cc:140-214            k3_occ, k3_place_paths
//     final translation :216-249                                   k3_read_counts, k3_read_paths
//
// Design (MI355X-first; integer / byte work bound by HBM transactions, no MFMA):
//  * a K2-mer is never materialised: it is a position in the 2-bit stream of the place sequences (`all`), its content is read
//    with unaligned 8-byte loads (32 bases per load) and compared 32 bases at a time;
//  * the dictionary is a SORT, not a hash table: every K2-mer position gets the 64-bit hash of its canonical form, the
//    (hash, position) pairs are radix-sorted, equal neighbours are verified by content (hash collisions between different
//    K2-mers are resolved exactly, k3_group_fix), and a distinct K2-mer is numbered by a scan -- no atomics, no probing;
//  * adjacency needs no lookups either: the successor of an occurrence is the next position of the same place;
//  * unipaths are chains over the 2D oriented nodes, resolved by the list ranking Step 2 uses (pointer jumping with
//    tile-local splitters), circles by min-jumping over K2-mer contents;
//  * everything that depends on the (arbitrary) edge numbering comes after one small sort of the unipath heads, so the
//    canonical and the replayed order share every kernel.
// The oracle (oracle/step3_oracle.cc) is the checker in tests/ only; nothing here calls it.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "ctx.h"
#include "../../include/w2rap_step3.h"

extern "C" w2rap_step2_ctx* w2rap_step2_create(int device, char* err, size_t errlen);
extern "C" void w2rap_step2_destroy(w2rap_step2_ctx*);

namespace w2 {
namespace {

constexpr unsigned MAXW = 20;                 // K2 <= 640 (the largest -K the reference's command line admits): at most 20 words of 32 bases
constexpr uint32_t NONE = 0xFFFFFFFFu;

// ---- 2-bit streams (LSB first: base p at bits 2(p&3) of byte p>>2); every buffer is padded by 16 readable bytes ----------
__device__ inline uint64_t stream64(const uint8_t* __restrict__ s, uint64_t pos) {       // 32 bases from base position pos
    const uint64_t b = pos >> 2; const unsigned sh = 2 * (unsigned)(pos & 3);
    uint64_t x = reinterpret_cast<const U64u*>(s + b)->v;
    if (sh) x = (x >> sh) | ((uint64_t)s[b + 8] << (64 - sh));
    return x;
}
__device__ inline unsigned stream1(const uint8_t* __restrict__ s, uint64_t pos) { return (s[pos >> 2] >> (2 * (pos & 3))) & 3u; }

// A K2-mer = K2 bases from position g of a stream.  Its canonical form (CF<K>::getForm, dna/CanonicalForm.h:58-66, even K:
// the smaller of the k-mer and its reverse complement) is handled as NW words of 32 bases, base 0 most significant, the
// last word left-aligned.  Forward word j = reversed groups of the 64 stream bits at g+32j; reverse-complement word j = the
// complement of the 64 stream bits at g+K2-32(j+1) (their top group is the complement of the k-mer's last base).
struct KGeom { unsigned K2, NW, tail, sbits; };       // tail = bases in the last word (1..32); sbits = hash bits the dictionary sort orders by
__device__ inline uint64_t kword_f(const uint8_t* s, uint64_t g, const KGeom& q, unsigned j) {
    uint64_t x = rev2_64(stream64(s, g + 32 * j));
    if (j == q.NW - 1 && q.tail < 32) x &= ~0ull << (64 - 2 * q.tail);
    return x;
}
__device__ inline uint64_t kword_r(const uint8_t* s, uint64_t g, const KGeom& q, unsigned j) {
    if (j == q.NW - 1 && q.tail < 32) return (~stream64(s, g) << (64 - 2 * q.tail));              // bases tail-1 .. 0, complemented
    return ~stream64(s, g + q.K2 - 32 * (j + 1));
}
__device__ inline uint64_t kword(const uint8_t* s, uint64_t g, const KGeom& q, bool rc, unsigned j) { return rc ? kword_r(s, g, q, j) : kword_f(s, g, q, j); }
// -1 / 0 / +1: oriented k-mer a against oriented k-mer b, lexicographic
__device__ inline int kcmp(const uint8_t* s, uint64_t ga, bool ra, uint64_t gb, bool rb, const KGeom& q) {
    for (unsigned j = 0; j < q.NW; ++j) {
        const uint64_t a = kword(s, ga, q, ra, j), b = kword(s, gb, q, rb, j);
        if (a != b) return a < b ? -1 : 1;
    }
    return 0;
}
// are two oriented k-mers equal?  Four words of each side are fetched together before they are compared (kcmp's one word at a time is a
// chain of dependent HBM round trips: what k3_group waits for)
__device__ inline bool kequal(const uint8_t* s, uint64_t ga, bool ra, uint64_t gb, bool rb, const KGeom& q) {
    for (unsigned j0 = 0; j0 < q.NW; j0 += 4) {
        uint64_t a[4], b[4];
#pragma unroll
        for (unsigned t = 0; t < 4; ++t) {
            const unsigned j = j0 + t < q.NW ? j0 + t : q.NW - 1;
            a[t] = kword(s, ga, q, ra, j); b[t] = kword(s, gb, q, rb, j);
        }
        if ((a[0] ^ b[0]) | (a[1] ^ b[1]) | (a[2] ^ b[2]) | (a[3] ^ b[3])) return false;
    }
    return true;
}
__device__ inline unsigned kbase(const uint8_t* s, uint64_t g, const KGeom& q, bool rc, unsigned t) {      // base t of the oriented k-mer
    return rc ? 3u - stream1(s, g + q.K2 - 1 - t) : stream1(s, g + t);
}
__device__ inline uint64_t mix64(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

// (The stored keys are the hashes ROTATED so that the sort key sits in the low bits: rocPRIM's radix sort returned unsorted
// data on small inputs when asked for a bit range that does not start at bit 0 -- seen with [24, 64) on 19 k pairs.)
constexpr unsigned SORT_BITS = 40;             // (W2RAP_TEST_SORT_BITS lowers it in the tests: many runs with several contents, same results)
__host__ __device__ inline uint64_t sort_rot(uint64_t h, unsigned sb) { return (h << sb) | (h >> (64 - sb)); }      // top sb bits -> low bits
__host__ __device__ inline uint64_t sort_mask(unsigned sb) { return (1ull << sb) - 1; }
__device__ inline bool same_run(uint64_t a, uint64_t b, unsigned sb) { return ((a ^ b) & sort_mask(sb)) == 0; }
static inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }
__global__ void __launch_bounds__(256) k3_iota(uint64_t n, uint32_t* __restrict__ a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(256) k3_gather_u64(uint64_t n, const uint64_t* __restrict__ src, const uint32_t* __restrict__ perm, uint64_t* __restrict__ dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
__global__ void __launch_bounds__(256) k3_gather_u32(uint64_t n, const uint32_t* __restrict__ src, const uint32_t* __restrict__ perm, uint32_t* __restrict__ dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
// largest i in [0, n) with a[i] <= x (a ascending, a[0] <= x)
template <class T>
__device__ inline uint64_t upper_index(const T* __restrict__ a, uint64_t n, T x) {
    uint64_t lo = 0, hi = n;
    while (hi - lo > 1) { const uint64_t m = (lo + hi) >> 1; if (a[m] <= x) lo = m; else hi = m; }
    return lo;
}

// The same for kernels whose threads hold CONSECUTIVE x (occurrences or words in position order): one binary search per block for its
// first element, then a short walk (a block of 256 spans a place or two).  Every thread of the block must call it (it synchronises).
// `blk` (k3_block_index): the searches of ALL blocks done beforehand, one per thread -- a block that starts with one thread's nineteen
// dependent loads while 255 wait spends ~10 us there, and that, not its work, was the time of every kernel over the occurrences
// (k3_kmer_keys 2.9 ms, k3_occ 1.4, k3_nbr 1.3, k3_pack_objs 1.3, k3_place_paths 1.1 at 149 M occurrences).
template <class T>
__device__ inline uint64_t upper_index_seq(const T* __restrict__ a, uint64_t n, T x, T x_block0, const uint32_t* __restrict__ blk = nullptr) {
    __shared__ uint64_t s_u0;
    uint64_t u;
    if (blk) u = blk[blockIdx.x];
    else {
        if (threadIdx.x == 0) s_u0 = upper_index(a, n, x_block0);
        __syncthreads();
        u = s_u0;
    }
    while (u + 1 < n && a[u + 1] <= x) ++u;
    return u;
}
// out[b] = upper_index(a, n, 256 b) for the blocks 0 .. nblocks (one more than there are: a block's last element starts from there)
template <class T>
__global__ void __launch_bounds__(256) k3_block_index(uint64_t nblocks, const T* __restrict__ a, uint64_t n, uint32_t* __restrict__ out) {
    const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b <= nblocks) out[b] = (uint32_t)upper_index(a, n, (T)(b * 256));
}

// ============================================================================= Involution (HyperBasevector.cc:648-660)
// The objects of a unipath graph start with distinct K-mers, so e's partner is the object whose first K-mer is the reverse
// complement of e's last one; the match is then VERIFIED base by base (any graph whose objects do not pair up is rejected).
__global__ void __launch_bounds__(256) k3_obj_ends(uint64_t NO, unsigned K, const uint8_t* __restrict__ bits, const uint64_t* __restrict__ base0,
                                                    const uint32_t* __restrict__ len, uint64_t* __restrict__ f_hi, uint64_t* __restrict__ f_lo,
                                                    uint64_t* __restrict__ r_hi, uint64_t* __restrict__ r_lo) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= NO) return;
    const KGeom q{K, (K + 31) / 32, K - 32 * ((K + 31) / 32 - 1), 0};
    const uint64_t g0 = base0[o], g1 = base0[o] + len[o] - K;
    f_hi[o] = kword_f(bits, g0, q, 0); f_lo[o] = q.NW > 1 ? kword_f(bits, g0, q, 1) : 0;
    r_hi[o] = kword_r(bits, g1, q, 0); r_lo[o] = q.NW > 1 ? kword_r(bits, g1, q, 1) : 0;
}
// (the objects are sorted by ONE 64-bit mix of their first K-mer; equal mixes -- 2^-64 per pair -- are walked through, the match itself is exact)
__device__ inline uint64_t inv_mix(uint64_t hi, uint64_t lo) { return mix64(mix64(0x452821E638D01377ull, hi), lo); }
__global__ void __launch_bounds__(256) k3_inv_mix(uint64_t NO, const uint64_t* __restrict__ f_hi, const uint64_t* __restrict__ f_lo, uint64_t* __restrict__ out) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o < NO) out[o] = inv_mix(f_hi[o], f_lo[o]);
}
__global__ void __launch_bounds__(256) k3_inv_match(uint64_t NO, const uint64_t* __restrict__ s_mix /* ascending */, const uint32_t* __restrict__ perm,
                                                     const uint64_t* __restrict__ f_hi, const uint64_t* __restrict__ f_lo,
                                                     const uint64_t* __restrict__ r_hi, const uint64_t* __restrict__ r_lo,
                                                     int32_t* __restrict__ inv, uint32_t* __restrict__ flags) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= NO) return;
    const uint64_t hi = r_hi[o], lo = r_lo[o], mx = inv_mix(hi, lo);
    uint64_t a = 0, b = NO;                                   // first sorted index with s_mix >= mx
    while (a < b) { const uint64_t m = (a + b) >> 1; if (s_mix[m] < mx) a = m + 1; else b = m; }
    int32_t found = -1;
    for (; a < NO && s_mix[a] == mx; ++a) { const uint32_t x = perm[a]; if (f_hi[x] == hi && f_lo[x] == lo) { found = (int32_t)x; break; } }
    inv[o] = found;
    if (found < 0) atomicOr(&flags[1], 1u);
}
__global__ void __launch_bounds__(256) k3_inv_verify(uint64_t nwords, uint64_t NO, const uint64_t* __restrict__ wordoff, const uint8_t* __restrict__ bits,
                                                      const uint64_t* __restrict__ base0, const uint32_t* __restrict__ len, const int32_t* __restrict__ inv,
                                                      uint32_t* __restrict__ flags) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // one 32-base word of one object
    if (w >= nwords) return;
    const uint64_t o = upper_index(wordoff, NO, w);
    const int32_t p = inv[o];
    if (p < 0) return;
    const uint32_t L = len[o];
    if (len[p] != L) { atomicOr(&flags[1], 2u); return; }
    const uint32_t t0 = (uint32_t)(w - wordoff[o]) * 32, n = L - t0 < 32 ? L - t0 : 32;
    uint64_t a = stream64(bits, base0[o] + t0);                                // bases t0 .. t0+n-1 of o
    // RC of the partner at the same places: partner bases L-1-t0 down to L-t0-n, complemented
    uint64_t b = rev2_64(~stream64(bits, base0[p] + (L - t0 >= 32 ? L - t0 - 32 : 0)));
    if (n < 32) { a &= (1ull << (2 * n)) - 1; b = (L - t0 >= 32) ? b : (b >> (2 * (32 - (L - t0)))); b &= (1ull << (2 * n)) - 1; }
    if (a != b) atomicOr(&flags[1], 2u);
}

// ============================================================================= places (Repath.cc:40-71) and FragDist (GapToyTools3.cc:622-634)
// per read: does its path imply >= K2 bases (:56-59); is the inverse path smaller (:60-62); two 64-bit hashes of the chosen one
constexpr unsigned FRAG_STRIPES = 32;
__global__ void __launch_bounds__(256) k3_place_keys(uint64_t n, uint64_t n_local, unsigned K, unsigned K2, const uint64_t* __restrict__ p_off, const int32_t* __restrict__ p_edges,
                                                      const int32_t* __restrict__ inv, const uint32_t* __restrict__ len,
                                                      uint64_t* __restrict__ keyA, uint64_t* __restrict__ keyB, uint8_t* __restrict__ state /*0 none, 1 as is, 2 inverse*/,
                                                      uint32_t* __restrict__ first1 /* per edge object: the first read whose place is that one edge */,
                                                      unsigned long long* __restrict__ counters /*0 pathed 1 multipathed 2 placed*/,
                                                      const int32_t* __restrict__ p_offset, unsigned long long* __restrict__ frag_count /* FragDist of the pairs, or null */) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned pathed = 0, multi = 0, placed = 0;
    // FragDist (GapToyTools3.cc:616-634) rides along: the mates 2i, 2i+1 sit in neighbouring lanes, the odd one hands its first edge down
    __shared__ uint32_t s_frag[100];
    if (frag_count) { for (unsigned i = threadIdx.x; i < 100; i += blockDim.x) s_frag[i] = 0; __syncthreads(); }
    {
        int first = -1; uint32_t mm = 0;
        if (r < n_local) { const uint64_t a = p_off[r]; mm = (uint32_t)(p_off[r + 1] - a); if (mm) first = p_edges[a]; }
        const int first2 = __shfl_down(first, 1);
        if (frag_count && !(r & 1) && r + 1 < n_local && first >= 0 && first2 >= 0) {
            const int e1 = first, e2 = inv[first2];
            if (e1 == e2 && (int)len[e1] >= 10000) {
                const int d = ((int)len[e2] - p_offset[r + 1]) - p_offset[r];
                if (d >= 0 && d < 1000) atomicAdd(&s_frag[d / 10], 1u);
            }
        }
    }
    if (r < n) {
        const uint64_t a = p_off[r]; const uint32_t m = (uint32_t)(p_off[r + 1] - a);
        pathed = m > 0 && r < n_local; multi = m > 2 && r < n_local;          // Repath.cc:38-41 (this rank's reads only)
        uint8_t st = 0; uint64_t hA = 0, hB = 0;
        if (m == 1) {
            // a one-edge path (nearly every read): the place IS min(e, inv e) -- an exact key, no hashing (top bit set; hashed keys clear it)
            const int x = p_edges[a], y = inv[x];
            if ((long long)len[x] >= (long long)K2) {
                st = y < x ? 2 : 1; hA = (1ull << 63) | (uint32_t)(y < x ? y : x); hB = 1;
                // nearly every read ends here: such places are told apart by direct addressing (the smallest read id = the representative a
                // stable sort would pick); only multi-edge places go through the sort below
                // (the reads come in ascending order, so after an edge's first few the entry is below r already: a plain load -- it can only be
                // stale upwards, the entries fall -- spares 50 M atomics on 70 k addresses, which were 1.8 of this kernel's 2.2 ms)
                uint32_t* f1 = &first1[y < x ? y : x];
                if (__hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (uint32_t)r) atomicMin(f1, (uint32_t)r);
            }
        } else if (m > 1) {
            // one pass over the path: the bases it implies, x against y = the inverse path (std::vector<int> order, decided at the first
            // difference), and the hashes of both
            long long nk = 0;
            int cmp = 0;
            uint64_t xA = 0x243F6A8885A308D3ull ^ m, xB = 0x13198A2E03707344ull + m, yA = xA, yB = xB;
            for (uint32_t j = 0; j < m; ++j) {
                const int x = p_edges[a + j], y = inv[p_edges[a + m - 1 - j]];
                nk += (long long)len[x] - ((int)K - 1);
                if (cmp == 0 && x != y) cmp = y < x ? -1 : 1;
                xA = mix64(xA, (uint32_t)x); xB = (xB ^ ((uint32_t)x + 0x9E3779B9u)) * 0xD6E8FEB86659FD93ull; xB ^= xB >> 32;
                yA = mix64(yA, (uint32_t)y); yB = (yB ^ ((uint32_t)y + 0x9E3779B9u)) * 0xD6E8FEB86659FD93ull; yB ^= yB >> 32;
            }
            if (nk + ((int)K - 1) >= (long long)K2) {
                const bool rc = cmp < 0;
                st = rc ? 2 : 1; hA = (rc ? yA : xA) & ~(1ull << 63); hB = rc ? yB : xB;
            }
        }
        state[r] = st; keyA[r] = hA; keyB[r] = hB;
        placed = st != 0;
    }
    // (block totals into 64 slot triples: one address takes ~11 ns per atomic -- a per-wave add of two totals was 13 of this kernel's 13 ms)
    __shared__ uint32_t s_cnt[3];
    if (threadIdx.x < 3) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long mp = __ballot(pathed), mm = __ballot(multi), ml = __ballot(placed);
    if ((threadIdx.x & 63) == 0) {
        if (mp) atomicAdd(&s_cnt[0], (uint32_t)__builtin_popcountll(mp));
        if (mm) atomicAdd(&s_cnt[1], (uint32_t)__builtin_popcountll(mm));
        if (ml) atomicAdd(&s_cnt[2], (uint32_t)__builtin_popcountll(ml));
    }
    __syncthreads();
    if (threadIdx.x < 3 && s_cnt[threadIdx.x]) atomicAdd(&counters[3 * (blockIdx.x & 63u) + threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    // (FRAG_STRIPES copies of the hundred counters, 1 KB apart: every block has the same few bins -- the insert sizes --, ~200 k blocks on one address each)
    if (frag_count) for (unsigned j = threadIdx.x; j < 100; j += blockDim.x) if (s_frag[j]) atomicAdd(&frag_count[(blockIdx.x & (FRAG_STRIPES - 1u)) * 128u + j], (unsigned long long)s_frag[j]);
}
__global__ void __launch_bounds__(128) k3_frag_sum(const unsigned long long* __restrict__ stripes, unsigned long long* __restrict__ out) {
    if (threadIdx.x >= 100) return;
    unsigned long long t = 0;
    for (unsigned k = 0; k < FRAG_STRIPES; ++k) t += stripes[k * 128u + threadIdx.x];
    out[threadIdx.x] = t;
}
// the reads that go through the sort: a place of several edges (hashed key: top bit of keyA clear)
__global__ void __launch_bounds__(256) k3_flag_multi(uint64_t n, const uint8_t* __restrict__ st, const uint64_t* __restrict__ keyA, uint32_t* __restrict__ f) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f[i] = (st[i] && !(keyA[i] >> 63)) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k3_compact_multi(uint64_t n, const uint32_t* __restrict__ f, const uint64_t* __restrict__ excl, const uint64_t* __restrict__ keyA,
                                                         const uint64_t* __restrict__ keyB, uint32_t* __restrict__ ids, uint64_t* __restrict__ kA, uint64_t* __restrict__ kB) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && f[r]) { const uint64_t j = excl[r]; ids[j] = (uint32_t)r; kA[j] = keyA[r]; kB[j] = keyB[r]; }
}
// one-edge places: numbered behind the U_multi sorted places in edge-object order (the order their keys would sort in)
__global__ void __launch_bounds__(256) k3_flag_first1(uint64_t NO, const uint32_t* __restrict__ first1, uint32_t* __restrict__ f) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < NO) f[e] = first1[e] != NONE ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k3_rep_first1(uint64_t NO, const uint32_t* __restrict__ first1, const uint64_t* __restrict__ rank1, uint64_t U_multi, uint32_t* __restrict__ rep) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < NO && first1[e] != NONE) rep[U_multi + rank1[e]] = first1[e];
}
__global__ void __launch_bounds__(256) k3_place_of_one(uint64_t n, const uint8_t* __restrict__ st, const uint64_t* __restrict__ keyA, const uint64_t* __restrict__ rank1,
                                                        uint64_t U_multi, uint32_t* __restrict__ place_of_read) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n && st[r] && (keyA[r] >> 63)) place_of_read[r] = (uint32_t)(U_multi + rank1[(uint32_t)keyA[r]]);
}
__device__ inline int place_elem(const int32_t* p_edges, const int32_t* inv, uint64_t a, uint32_t m, bool rc, uint32_t j) {
    return rc ? inv[p_edges[a + m - 1 - j]] : p_edges[a + j];
}
// sorted by (keyA, keyB): an element starts a new place unless it equals its predecessor -- verified element by element
__global__ void __launch_bounds__(256) k3_place_heads(uint64_t np, const uint64_t* __restrict__ kA, const uint64_t* __restrict__ kB, const uint32_t* __restrict__ ids,
                                                       const uint8_t* __restrict__ st, const uint64_t* __restrict__ p_off, const int32_t* __restrict__ p_edges,
                                                       const int32_t* __restrict__ inv, uint32_t* __restrict__ head, uint32_t* __restrict__ flags) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= np) return;
    uint32_t h = 1;
    // (sorted by kA alone: a neighbour with the same kA and another kB is a second place under one 64-bit key -- 2^-64 per pair; the run may
    // then hold its places interleaved, and the caller sorts again by both keys)
    if (j > 0 && kA[j] == kA[j - 1] && kB[j] != kB[j - 1]) atomicOr(&flags[5], 1u);
    if (j > 0 && kA[j] == kA[j - 1] && kB[j] == kB[j - 1]) {
        const uint32_t r1 = ids[j], r0 = ids[j - 1];
        const uint64_t a1 = p_off[r1], a0 = p_off[r0];
        const uint32_t m1 = (uint32_t)(p_off[r1 + 1] - a1), m0 = (uint32_t)(p_off[r0 + 1] - a0);
        bool same = m1 == m0;
        for (uint32_t t = 0; same && t < m1; ++t) same = place_elem(p_edges, inv, a1, m1, st[r1] == 2, t) == place_elem(p_edges, inv, a0, m0, st[r0] == 2, t);
        if (same) h = 0; else atomicOr(&flags[1], 4u);           // two different places under one 128-bit key: rejected, never merged
    }
    head[j] = h;
}
__global__ void __launch_bounds__(256) k3_place_index(uint64_t np, const uint32_t* __restrict__ head, const uint64_t* __restrict__ excl, const uint32_t* __restrict__ ids,
                                                       uint32_t* __restrict__ place_of_read, uint32_t* __restrict__ rep) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= np) return;
    const uint32_t u = (uint32_t)(excl[j] + head[j]) - 1;       // inclusive scan - 1
    place_of_read[ids[j]] = u;
    if (head[j]) rep[u] = ids[j];
}
// ============================================================================= --extend_paths (Repath.cc:72-96, experimental in the reference)
// Every unique place gets the SOLE edge that enters its first vertex in front and the sole edge that leaves its last vertex behind,
// unless the place already holds that edge; the longer places are ADDED to the list (as they are), which is sorted and made unique again.
// As written the reference's loops never advance v / w (`while (hb.To(v).solo())` pushes e once, finds it a member on the next round
// and breaks): one edge per side at most, and the right side tests membership against the place WITH its new front edge.  Here the
// extended places go back into the path set as additional paths (an empty path for a place that did not grow), and the places are
// computed again: that canonicalises an extended place against its reverse complement, which the reference does not -- the same
// K2-mers either way, so the same graph and, the reads' own places being untouched, the same read paths.
__global__ void __launch_bounds__(256) k3_vertex_deg(uint64_t NO, const int32_t* __restrict__ to_left, const int32_t* __restrict__ to_right, uint64_t NV,
                                                      uint32_t* __restrict__ indeg, uint32_t* __restrict__ outdeg, int32_t* __restrict__ in_e, int32_t* __restrict__ out_e,
                                                      uint32_t* __restrict__ flags) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NO) return;
    const int32_t v = to_left[e], w = to_right[e];
    if (v < 0 || w < 0 || (uint64_t)v >= NV || (uint64_t)w >= NV) { atomicOr(&flags[1], 512u); return; }
    atomicAdd(&outdeg[v], 1u); out_e[v] = (int32_t)e;            // (with degree 1 there is one writer)
    atomicAdd(&indeg[w], 1u); in_e[w] = (int32_t)e;
}
__global__ void __launch_bounds__(256) k3_extend_len(uint64_t U, const uint32_t* __restrict__ rep, const uint8_t* __restrict__ st, const uint64_t* __restrict__ p_off,
                                                      const int32_t* __restrict__ p_edges, const int32_t* __restrict__ inv, const int32_t* __restrict__ to_left,
                                                      const int32_t* __restrict__ to_right, const uint32_t* __restrict__ indeg, const uint32_t* __restrict__ outdeg,
                                                      const int32_t* __restrict__ in_e, const int32_t* __restrict__ out_e, int32_t* __restrict__ ext /*[2U]*/,
                                                      uint32_t* __restrict__ xlen) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const uint32_t r = rep[u]; const uint64_t a = p_off[r]; const uint32_t m = (uint32_t)(p_off[r + 1] - a); const bool rc = st[r] == 2;
    const int v = to_left[place_elem(p_edges, inv, a, m, rc, 0)], w = to_right[place_elem(p_edges, inv, a, m, rc, m - 1)];
    int eL = indeg[v] == 1 ? in_e[v] : -1, eR = outdeg[w] == 1 ? out_e[w] : -1;
    for (uint32_t j = 0; j < m && (eL >= 0 || eR >= 0); ++j) {
        const int x = place_elem(p_edges, inv, a, m, rc, j);
        if (x == eL) eL = -1;
        if (x == eR) eR = -1;
    }
    if (eR >= 0 && eR == eL) eR = -1;                             // (the front edge is a member by then)
    ext[2 * u] = eL; ext[2 * u + 1] = eR;
    xlen[u] = (eL >= 0 || eR >= 0) ? m + (eL >= 0) + (eR >= 0) : 0u;
}
__global__ void __launch_bounds__(256) k3_extend_fill(uint64_t U, const uint32_t* __restrict__ rep, const uint8_t* __restrict__ st, const uint64_t* __restrict__ p_off,
                                                       const int32_t* __restrict__ p_edges, const int32_t* __restrict__ inv, const int32_t* __restrict__ ext,
                                                       const uint64_t* __restrict__ xoff, int32_t* __restrict__ out) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U || xoff[u + 1] == xoff[u]) return;
    const uint32_t r = rep[u]; const uint64_t a = p_off[r]; const uint32_t m = (uint32_t)(p_off[r + 1] - a); const bool rc = st[r] == 2;
    uint64_t o = xoff[u];
    if (ext[2 * u] >= 0) out[o++] = ext[2 * u];
    for (uint32_t j = 0; j < m; ++j) out[o++] = place_elem(p_edges, inv, a, m, rc, j);
    if (ext[2 * u + 1] >= 0) out[o++] = ext[2 * u + 1];
}

// Repath.cc:101-123: bases of a place = its edges overlapped by K-1, first and last edge cut to at most K2 bases
__global__ void __launch_bounds__(256) k3_place_layout(uint64_t U, unsigned K, unsigned K2, const uint32_t* __restrict__ rep, const uint8_t* __restrict__ st,
                                                        const uint64_t* __restrict__ p_off, const int32_t* __restrict__ p_edges, const int32_t* __restrict__ inv,
                                                        const uint32_t* __restrict__ len, uint32_t* __restrict__ plen /*edges*/, uint32_t* __restrict__ nbases,
                                                        uint32_t* __restrict__ nwords, uint32_t* __restrict__ nkm, int32_t* __restrict__ ltrunc, int32_t* __restrict__ rtrunc,
                                                        uint32_t* __restrict__ flags) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const uint32_t r = rep[u]; const uint64_t a = p_off[r]; const uint32_t m = (uint32_t)(p_off[r + 1] - a); const bool rc = st[r] == 2;
    unsigned long long L = 0;
    for (uint32_t j = 0; j < m; ++j) L += len[place_elem(p_edges, inv, a, m, rc, j)] - (j + 1 < m ? K - 1 : 0);
    int lt = 0, rt = 0;
    if (m > 1) {
        const uint32_t l0 = len[place_elem(p_edges, inv, a, m, rc, 0)], l1 = len[place_elem(p_edges, inv, a, m, rc, m - 1)];
        if (l0 > K2) lt = (int)(l0 - K2);
        if (l1 > K2) rt = (int)(l1 - K2);
    }
    L -= (unsigned long long)lt + rt;
    if (L >= (1ull << 31)) { atomicOr(&flags[1], 8u); L = K2; }
    plen[u] = m; nbases[u] = (uint32_t)L; nwords[u] = (uint32_t)((L + 31) / 32) + 1;      // (+1: a word of slack so that 64-bit reads past the end stay inside the place's own zeros)
    nkm[u] = L >= K2 ? (uint32_t)(L - K2 + 1) : 0;
    ltrunc[u] = lt; rtrunc[u] = rt;
}
// the oriented edge vector of every place, and where each piece starts in the place's local coordinates
__global__ void __launch_bounds__(256) k3_place_vec(uint64_t U, unsigned K, const uint32_t* __restrict__ rep, const uint8_t* __restrict__ st,
                                                     const uint64_t* __restrict__ p_off, const int32_t* __restrict__ p_edges, const int32_t* __restrict__ inv,
                                                     const uint32_t* __restrict__ len, const uint64_t* __restrict__ voff, const int32_t* __restrict__ ltrunc,
                                                     int32_t* __restrict__ vec, int64_t* __restrict__ pstart /* local base at which piece j's base 0 would lie */) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const uint32_t r = rep[u]; const uint64_t a = p_off[r]; const uint32_t m = (uint32_t)(p_off[r + 1] - a); const bool rc = st[r] == 2;
    int64_t at = -(int64_t)ltrunc[u];
    for (uint32_t j = 0; j < m; ++j) {
        const int e = place_elem(p_edges, inv, a, m, rc, j);
        vec[voff[u] + j] = e; pstart[voff[u] + j] = at;
        at += (int64_t)len[e] - (K - 1);
    }
}
// one thread per 32-base word of `all`
__global__ void __launch_bounds__(256) k3_all_fill(uint64_t nwords_total, uint64_t U, const uint64_t* __restrict__ woff, const uint32_t* __restrict__ nbases,
                                                    const uint64_t* __restrict__ voff, const int32_t* __restrict__ vec, const int64_t* __restrict__ pstart,
                                                    const uint8_t* __restrict__ obits, const uint64_t* __restrict__ base0, const uint32_t* __restrict__ len,
                                                    uint64_t* __restrict__ all, const uint32_t* __restrict__ blk) {
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, w0 = (uint64_t)blockIdx.x * blockDim.x;
    const uint64_t u = upper_index_seq(woff, U, w < nwords_total ? w : nwords_total - 1, w0, blk);
    if (w >= nwords_total) return;
    const uint32_t L = nbases[u];
    const uint64_t t0 = (w - woff[u]) * 32;
    uint64_t out = 0;
    if (t0 < L) {
        const uint32_t n = L - t0 < 32 ? (uint32_t)(L - t0) : 32;
        const uint64_t v0 = voff[u]; const uint32_t m = (uint32_t)(voff[u + 1] - v0);
        // piece holding local base t0: the last one whose start <= t0 (piece j holds [pstart[j] .. pstart[j+1]) except the last, which holds its whole rest)
        uint32_t j = 0;
        { uint32_t lo = 0, hi = m; while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (pstart[v0 + mid] <= (int64_t)t0) lo = mid; else hi = mid; } j = lo; }
        const int e = vec[v0 + j];
        const uint64_t src = (uint64_t)((int64_t)t0 - pstart[v0 + j]);
        const uint64_t room = (j + 1 < m) ? (uint64_t)(pstart[v0 + j + 1] - (int64_t)t0) : (uint64_t)len[e] - src;     // bases of this piece from t0 on
        if (room >= n) out = stream64(obits, base0[e] + src);
        else {                                                    // the word straddles pieces: base by base
            for (uint32_t i = 0; i < n; ++i) {
                const int64_t t = (int64_t)t0 + i;
                while (j + 1 < m && pstart[v0 + j + 1] <= t) ++j;
                out |= (uint64_t)stream1(obits, base0[vec[v0 + j]] + (uint64_t)(t - pstart[v0 + j])) << (2 * i);
            }
        }
        if (n < 32) out &= (1ull << (2 * n)) - 1;
    }
    all[w] = out;
}

// ---- K2-mers that need no dictionary.  When every K-mer of the small-K graph occurs ONCE in it (up to the involution) -- the unipath graph
// BuildReadQGraph builds: W2RAP_STEP3_UNIQUE_KMERS -- two K2-mer occurrences with one content lie on the same walk through the graph (their
// K-mers pin them to the same edges at the same offsets).  A one-edge place holds its edge whole; every other place that has this edge (or its
// inverse) at an END holds only K2 bases of it (k3_place_layout: the first K2-mer or the last), so a K2-mer STRICTLY inside the edge can occur
// a second time only where the edge is a MIDDLE element of a longer place, or in the edge itself when it is its own inverse.  Where neither is
// the case the K2-mers strictly inside a one-edge place are their own representatives with both neighbours beside them: k3_kmer_keys
// decides their orientation and nothing else, and only the others (~22 % at 1 SNP per 2 kb) are hashed, partitioned and grouped.
__global__ void __launch_bounds__(256) k3_mid_edges(uint64_t U, uint64_t NO, const uint64_t* __restrict__ voff, const int32_t* __restrict__ vec, const int32_t* __restrict__ inv,
                                                     uint8_t* __restrict__ shared_edge) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NO && inv[i] == (int32_t)i) shared_edge[i] = 1;
    if (i >= U) return;
    const uint64_t v0 = voff[i]; const uint32_t m = (uint32_t)(voff[i + 1] - v0);
    for (uint32_t j = 1; j + 1 < m; ++j) { const int32_t e = vec[v0 + j]; shared_edge[e] = 1; shared_edge[inv[e]] = 1; }
}
__global__ void __launch_bounds__(256) k3_lone_places(uint64_t U, const uint64_t* __restrict__ voff, const int32_t* __restrict__ vec, const uint8_t* __restrict__ shared_edge,
                                                       uint8_t* __restrict__ lone) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const uint64_t v0 = voff[u];
    lone[u] = (voff[u + 1] - v0 == 1 && !shared_edge[vec[v0]]) ? 1 : 0;
}

// ============================================================================= the K2-mer dictionary (BigKPather.cc:40-55, 96-108)
// one thread per K2-mer occurrence: canonical orientation, palindrome flag, context, hash of the canonical form.
// The 2 NW words a thread needs (forward and reverse-complement form) lie within K2 + 2 bases of its position, and the 256 occurrences of a
// block are consecutive positions of one place or a few: the block's stretch of `all` is staged in LDS once (coalesced 8-B loads) and
// every word is cut from two aligned LDS words -- 14 unaligned 8-B + 14 one-byte global loads per thread before (K2 = 200), none now.
// A block that spans many short places (its stretch does not fit KW_WORDS) reads the rest from global memory, aligned.
constexpr unsigned KW_WORDS = 512;                                // 32 bases each: 16 k bases per block
struct StreamWin {
    const uint64_t* lds; const uint64_t* glob; uint64_t w0; uint32_t nw;
    __device__ inline uint64_t at(uint64_t pos) const {           // 32 bases from base position pos (= stream64 on the bytes)
        const uint64_t w = pos >> 5; const unsigned sh = 2u * (unsigned)(pos & 31u);
        const uint64_t rel = w - w0;
        uint64_t lo, hi;
        if (rel + 1 < nw) { lo = lds[rel]; hi = lds[rel + 1]; } else { lo = glob[w]; hi = glob[w + 1]; }
        return sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
    }
    __device__ inline unsigned base(uint64_t pos) const {
        const uint64_t w = pos >> 5, rel = w - w0;
        const uint64_t v = rel < nw ? lds[rel] : glob[w];
        return (unsigned)(v >> (2u * (unsigned)(pos & 31u))) & 3u;
    }
};
constexpr unsigned KK_PER = 4;                                    // occurrences per thread: ONE prologue (place search, window) per 1024 occurrences
__global__ void __launch_bounds__(256) k3_kmer_keys(uint64_t N2, uint64_t U, KGeom q, const uint64_t* __restrict__ koff, const uint64_t* __restrict__ woff,
                                                     const uint32_t* __restrict__ nbases, const uint64_t* __restrict__ allw, uint64_t* __restrict__ key,
                                                     uint32_t* __restrict__ val /* the position itself (sorted dictionary only), or null */,
                                                     uint16_t* __restrict__ meta /* ctx | rc << 8 | pal << 9 */,
                                                     uint64_t* __restrict__ gpos /* stream position of every occurrence */,
                                                     uint32_t* __restrict__ grp_rep, uint32_t* __restrict__ ctx_by_x /* = k3_rep_init: every occurrence its own representative */,
                                                     const uint8_t* __restrict__ lone /* per place: its K2-mers strictly inside occur nowhere else (k3_lone_places), or null */,
                                                     unsigned long long* __restrict__ npairs /* with lone: key / val become the LIST of the other occurrences */,
                                                     const uint32_t* __restrict__ blk /* k3_block_index of koff, one entry per 256 occurrences */) {
    __shared__ uint64_t s_win[KW_WORDS];
    __shared__ uint64_t s_u1, s_w0; __shared__ uint32_t s_nw;
    __shared__ uint32_t s_cnt; __shared__ unsigned long long s_base;
    constexpr uint64_t SPAN = 256ull * KK_PER;
    const uint64_t x0 = (uint64_t)blockIdx.x * SPAN;
    const uint64_t x1 = x0 + SPAN - 1 < N2 ? x0 + SPAN - 1 : N2 - 1;              // the block's last occurrence
    if (threadIdx.x == 64) {                                      // its place
        const uint64_t nb256 = (N2 + 255) / 256, bi = ((uint64_t)blockIdx.x + 1) * KK_PER;
        const uint64_t c = blk[bi < nb256 ? bi : nb256];
        s_u1 = koff[c] <= x1 ? c : upper_index(koff, U, x1);
    }
    if (threadIdx.x == 0) s_cnt = 0;
    uint64_t u = blk[(uint64_t)blockIdx.x * KK_PER];              // the place of the block's first occurrence
    while (u + 1 < U && koff[u + 1] <= x0) ++u;
    __syncthreads();                                              // (s_u1)
    if (threadIdx.x == 0) {
        const uint64_t u1 = s_u1;
        const uint64_t gA = woff[u] * 32 + (x0 - koff[u]), gB = woff[u1] * 32 + (x1 - koff[u1]) + q.K2 + 1;
        const uint64_t wa = (gA ? gA - 1 : 0) >> 5, nw = (gB >> 5) - wa + 2;
        s_w0 = wa; s_nw = nw < KW_WORDS ? (uint32_t)nw : KW_WORDS;
    }
    __syncthreads();
    const StreamWin W{s_win, allw, s_w0, s_nw};
    for (unsigned i = threadIdx.x; i < W.nw; i += 256) s_win[i] = allw[W.w0 + i];
    __syncthreads();
    uint64_t pk[KK_PER]; uint32_t pat[KK_PER];                    // (lone: this thread's entries of the list)
#pragma unroll
    for (unsigned it = 0; it < KK_PER; ++it) {
        pat[it] = NONE; pk[it] = 0;
        const uint64_t x = x0 + (uint64_t)it * 256 + threadIdx.x;
        if (x >= N2) continue;
        while (u + 1 < U && koff[u + 1] <= x) ++u;
        const uint32_t t = (uint32_t)(x - koff[u]), L = nbases[u];
        const uint64_t g = woff[u] * 32 + t;
        gpos[x] = g;
        // strictly inside an edge no longer place shares: the one occurrence of its K2-mer, both neighbours in the place -- orientation only
        const bool alone = lone && lone[u] && t > 0 && t + q.K2 < L;
        uint64_t hf = 0x6A09E667F3BCC908ull, hr = hf;
        int cmp = 0;                                              // rc against forward, decided at the first differing word
        for (unsigned j = 0; j < q.NW; ++j) {
            const bool lastw = j == q.NW - 1 && q.tail < 32;
            uint64_t f = rev2_64(W.at(g + 32 * j));               // = kword_f / kword_r
            if (lastw) f &= ~0ull << (64 - 2 * q.tail);
            const uint64_t r = lastw ? (~W.at(g) << (64 - 2 * q.tail)) : ~W.at(g + q.K2 - 32 * (j + 1));
            if (cmp == 0 && f != r) cmp = r < f ? -1 : 1;
            if (alone) { if (cmp) break; }
            else { hf = mix64(hf, f); hr = mix64(hr, r); }
        }
        const bool rc = cmp < 0, pal = cmp == 0;                  // REV iff the reverse complement is smaller; a palindrome stays forward
        unsigned ctx = 0;
        if (L > q.K2) {                                           // a place of exactly K2 bases has no context (BigKPather.cc:45)
            if (t > 0) ctx |= 1u << (4 + W.base(g - 1));
            if (t + q.K2 < L) ctx |= 1u << W.base(g + q.K2);
        }
        if (rc) ctx = brev8(ctx);
        const uint32_t m = ctx | (rc ? 256u : 0u) | (pal ? 512u : 0u);
        meta[x] = (uint16_t)m;
        grp_rep[x] = (uint32_t)x; ctx_by_x[x] = m & 0x2FFu;
        const uint64_t k = sort_rot(rc ? hr : hf, q.sbits);
        if (!lone) { key[x] = k; if (val) val[x] = (uint32_t)x; }
        else if (!alone) { pk[it] = k; pat[it] = atomicAdd(&s_cnt, 1u); }
    }
    if (!lone) return;
    // ---- the occurrences the dictionary has to group, as a dense list
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) s_base = atomicAdd(npairs, (unsigned long long)s_cnt);
    __syncthreads();
#pragma unroll
    for (unsigned it = 0; it < KK_PER; ++it)
        if (pat[it] != NONE) { key[s_base + pat[it]] = pk[it]; val[s_base + pat[it]] = (uint32_t)(x0 + (uint64_t)it * 256 + threadIdx.x); }
}
// The pairs are sorted by the top SORT_BITS bits of the hash only (5 radix passes instead of 8); a RUN = neighbours with equal sort
// keys.  An occurrence starts a new group unless its canonical form equals its predecessor's; a run that holds more than one
// content (different K2-mers under one sort key: expected N^2 / 2^(SORT_BITS+1) pairs) is flagged and settled exactly by k3_group_fix.
__global__ void __launch_bounds__(256) k3_group(uint64_t N2, uint64_t U, KGeom q, const uint64_t* __restrict__ key, const uint32_t* __restrict__ val,
                                                 const uint16_t* __restrict__ meta, const uint64_t* __restrict__ gpos,
                                                 const uint8_t* __restrict__ all, uint32_t* __restrict__ head, uint32_t* __restrict__ coll, unsigned long long* __restrict__ ncoll) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N2) return;
    uint32_t h = 1, cflag = 0;
    if (j > 0 && same_run(key[j], key[j - 1], q.sbits)) {
        bool same = key[j] == key[j - 1];
        if (same) {
            const uint32_t xa = val[j], xb = val[j - 1];
            same = kequal(all, gpos[xa], (meta[xa] >> 8) & 1, gpos[xb], (meta[xb] >> 8) & 1, q);
        }
        if (same) h = 0; else { cflag = 1; atomicAdd(ncoll, 1ull); }
    }
    head[j] = h ? (uint32_t)j + 1 : 0;             // (j + 1 for heads: an inclusive max-scan then gives every element its group head)
    coll[j] = cflag;
}
// a run that contains a collision: the first flagged element of the run regroups the whole run serially and exactly -- an element is a
// head iff no earlier element of the run has its content; every other element gets its head recorded in over[]
__global__ void __launch_bounds__(256) k3_group_fix(uint64_t N2, uint64_t U, KGeom q, const uint64_t* __restrict__ key, const uint32_t* __restrict__ val,
                                                     const uint16_t* __restrict__ meta, const uint64_t* __restrict__ gpos,
                                                     const uint8_t* __restrict__ all, const uint32_t* __restrict__ coll, uint32_t* __restrict__ head, uint32_t* __restrict__ over) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N2 || !coll[j]) return;
    uint64_t a = j;
    while (a > 0 && same_run(key[a - 1], key[j], q.sbits)) { --a; if (coll[a]) return; }   // an earlier flagged element owns this run
    uint64_t b = j + 1;
    while (b < N2 && same_run(key[b], key[j], q.sbits)) ++b;
    for (uint64_t i = a + 1; i < b; ++i) {
        const uint32_t xi = val[i];
        const uint64_t gi = gpos[xi]; const bool ri = (meta[xi] >> 8) & 1;
        uint32_t found = NONE;
        for (uint64_t e = a; e < i; ++e) {
            if (e != a && !head[e]) continue;                                       // compare with the heads found so far only
            if (key[e] != key[i]) continue;
            const uint32_t xe = val[e];
            if (kequal(all, gi, ri, gpos[xe], (meta[xe] >> 8) & 1, q)) { found = (uint32_t)e; break; }
        }
        head[i] = found == NONE ? (uint32_t)i + 1 : 0u;
        over[i] = found;                                                              // NONE for heads
    }
}
// ---- the dictionary WITHOUT the library sort (the default; the sorted form above stays for the replay of a given edge order, whose
// lookup wants the distinct K2-mers in hash order, and as the fallback).  What the dictionary has to deliver is, for every occurrence,
// the FIRST occurrence with the same canonical content (BigDict is a set by content; "first" makes the numbering independent of any
// order of insertion) and the OR of the group's contexts.  Grouping needs no order, only co-location:
//   k3_dict_part   one to three radix-PARTITION passes over (hash, position) pairs, <= 512 bins each, until a partition holds 320..640 pairs.
//                  A block takes 4096 pairs through LDS: bin histogram by LDS atomics (whose return value is the pair's rank in its bin),
//                  one global atomic per (block, bin) reserves the bin's next run, the pairs are grouped by bin in LDS and leave as
//                  runs (coalesced 8-B / 4-B stores).  Bins have a FIXED capacity (the hashes are uniform: mean + >20 sigma), so no
//                  counting pre-pass is needed; only a K2-mer with hundreds of places can overflow one, and then the sorted form runs.
//   k3_dict_group  a block per partition: an LDS table of (hash tag | position) words.  First an OPTIMISTIC grouping that reads no content:
//                  an empty slot is claimed by CAS, a slot with the same tag takes ds_min_u64 (equal tags, so the minimum is the smallest
//                  position), anything else probes on.  Then every pair reads its slot -- the representative --, the ~9 % that are not
//                  their own representative are gathered in LDS and verified by content, all lanes busy, one round of loads per block.
//                  Only those write anything (their grp_rep entry, an atomicOr into the representative's context).  A partition in
//                  which a verification fails (two K2-mers under one 32-bit tag: 2^-32 per pair) is regrouped with every tag match
//                  verified before it is trusted.
// Traffic: 12 B read + 12 B written per pair and pass, 12 B read by the grouping; against five library passes of the same 24 B plus the
// neighbour-verification pass (k3_group) that re-read both contents of EVERY adjacent pair.
constexpr unsigned DP_CH = 4096, DP_T = 1024, DP_MAXB = 512, DP_CAP = 2048, DP_SLOTS = 4096, DP_AVG = 1280, DP_GT = 512;
template <bool FIRST>
__global__ void __launch_bounds__(DP_T) k3_dict_part(uint64_t n_first, const uint64_t* __restrict__ skey, const uint32_t* __restrict__ sx,
                                                     const uint32_t* __restrict__ scnt, uint64_t scap, unsigned bps /* blocks per source bin */,
                                                     unsigned shift, unsigned nb, uint64_t dcap, uint64_t* __restrict__ dkey, uint32_t* __restrict__ dx,
                                                     uint32_t* __restrict__ dcnt, uint32_t* __restrict__ ovf) {
    __shared__ uint64_t st_key[DP_CH];
    __shared__ uint32_t st_x[DP_CH];
    __shared__ uint32_t hist[DP_MAXB], lbase[DP_MAXB], gb[DP_MAXB], wsum[DP_T / 64];
    const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t seg = blockIdx.x / bps, chunk = blockIdx.x % bps;
    uint64_t n_src = n_first;
    if (!FIRST) { n_src = scnt[seg]; if (n_src > scap) n_src = scap; }
    const uint64_t i0 = chunk * DP_CH;
    if (i0 >= n_src) return;                                      // (block-uniform)
    const unsigned nitems = n_src - i0 < DP_CH ? (unsigned)(n_src - i0) : DP_CH;
    const uint64_t sbase = seg * scap + i0;
    if (tid < DP_MAXB) hist[tid] = 0;
    __syncthreads();
    uint64_t k[DP_CH / DP_T]; uint32_t xx[DP_CH / DP_T], r[DP_CH / DP_T];
#pragma unroll
    for (unsigned j = 0; j < DP_CH / DP_T; ++j) {
        const unsigned i = j * DP_T + tid;
        r[j] = NONE; k[j] = 0; xx[j] = 0;
        if (i < nitems) {
            k[j] = skey[sbase + i]; xx[j] = FIRST && !sx ? (uint32_t)(i0 + i) : sx[sbase + i];
            r[j] = atomicAdd(&hist[(unsigned)(k[j] >> shift) & (nb - 1)], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the bin counts (nb <= 512: one per thread of the first eight waves) + the global reservation of every bin's run
    const uint32_t cnt = tid < nb ? hist[tid] : 0u;
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    if (tid < nb) {
        uint32_t woff = 0;
        for (unsigned w = 0; w < wv; ++w) woff += wsum[w];
        lbase[tid] = woff + incl - cnt;
        uint32_t g = 0;
        if (cnt) { g = atomicAdd(&dcnt[seg * nb + tid], cnt); if ((uint64_t)g + cnt > dcap) *ovf = 1u; }
        gb[tid] = g;
    }
    __syncthreads();
#pragma unroll
    for (unsigned j = 0; j < DP_CH / DP_T; ++j)
        if (r[j] != NONE) { const unsigned pos = lbase[(unsigned)(k[j] >> shift) & (nb - 1)] + r[j]; st_key[pos] = k[j]; st_x[pos] = xx[j]; }
    __syncthreads();
    const uint64_t dbase = seg * nb * dcap;
    for (unsigned i = tid; i < nitems; i += DP_T) {
        const uint64_t key = st_key[i];
        const unsigned bin = (unsigned)(key >> shift) & (nb - 1);
        const uint64_t d = (uint64_t)gb[bin] + (i - lbase[bin]);
        if (d < dcap) { dkey[dbase + bin * dcap + d] = key; dx[dbase + bin * dcap + d] = st_x[i]; }
    }
}
__global__ void __launch_bounds__(DP_GT) k3_dict_group(uint64_t cap, const uint64_t* __restrict__ pkey, const uint32_t* __restrict__ px, const uint32_t* __restrict__ pcnt,
                                                       unsigned slot_shift, uint32_t tagmask, KGeom q, const uint8_t* __restrict__ all, const uint64_t* __restrict__ gpos,
                                                       const uint16_t* __restrict__ meta, uint32_t* __restrict__ grp_rep, uint32_t* __restrict__ ctx_by_x,
                                                       int mode /* 0: verify the duplicates here; 1: trust the tags, k3_verify_runs checks behind; 2: verify every tag match */) {
    __shared__ unsigned long long tab[DP_SLOTS];
    __shared__ uint32_t pend_x[DP_CAP], pend_rep[DP_CAP];
    __shared__ uint32_t npend, bad;
    constexpr unsigned long long EMPTY = ~0ull;
    constexpr unsigned IPT = DP_CAP / DP_GT;
    const unsigned tid = threadIdx.x;
    uint32_t cnt = pcnt[blockIdx.x];
    if (cnt == 0) return;
    if (cnt > cap) cnt = (uint32_t)cap;                           // (overflow: flagged by the partition pass, the caller starts over)
    const uint64_t base = (uint64_t)blockIdx.x * cap;
    uint64_t key[IPT]; uint32_t x[IPT]; unsigned slot[IPT];
#pragma unroll
    for (unsigned j = 0; j < IPT; ++j) {
        const unsigned i = j * DP_GT + tid;
        key[j] = 0; x[j] = NONE; slot[j] = 0;
        if (i < cnt) { key[j] = pkey[base + i]; x[j] = px[base + i]; }
    }
    for (unsigned i = tid; i < DP_SLOTS; i += DP_GT) tab[i] = EMPTY;
    if (tid == 0) { npend = 0; bad = mode == 2 ? 1u : 0u; }
    __syncthreads();
    // ---- optimistic grouping by (tag, probe position): no content is read
#pragma unroll
    for (unsigned j = 0; j < IPT; ++j) {
        if (x[j] == NONE || mode == 2) continue;
        const uint32_t tag = (uint32_t)(key[j] >> 32) & tagmask;
        const unsigned long long mine = ((unsigned long long)tag << 32) | x[j];
        unsigned s = (unsigned)(key[j] >> slot_shift) & (DP_SLOTS - 1);
        for (;;) {
            unsigned long long v = __hip_atomic_load(&tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v == EMPTY) { v = atomicCAS(&tab[s], EMPTY, mine); if (v == EMPTY) break; }
            if ((uint32_t)(v >> 32) == tag) { atomicMin(&tab[s], mine); break; }
            s = (s + 1) & (DP_SLOTS - 1);
        }
        slot[j] = s;
    }
    __syncthreads();
    // ---- the occurrences that are not their own representative (duplicates, ~9 %), gathered, then verified by content all lanes at once
#pragma unroll
    for (unsigned j = 0; j < IPT; ++j) {
        if (x[j] == NONE || mode == 2) continue;
        const uint32_t rep = (uint32_t)tab[slot[j]];
        if (rep != x[j]) { const uint32_t p = atomicAdd(&npend, 1u); pend_x[p] = x[j]; pend_rep[p] = rep; }
    }
    __syncthreads();
    const uint32_t np = npend;
    if (mode == 0)
        for (unsigned p = tid; p < np; p += DP_GT) {
            const uint32_t xa = pend_x[p], xb = pend_rep[p];
            if (!kequal(all, gpos[xa], (meta[xa] >> 8) & 1, gpos[xb], (meta[xb] >> 8) & 1, q)) bad = 1u;
        }
    __syncthreads();
    if (!bad) {
        for (unsigned p = tid; p < np; p += DP_GT) {
            const uint32_t xa = pend_x[p], xb = pend_rep[p];
            grp_rep[xa] = xb; atomicOr(&ctx_by_x[xb], (uint32_t)(meta[xa] & 0x2FFu));
        }
        return;
    }
    // ---- two different K2-mers met under one tag (2^-32 per pair; the tests narrow the tag): this partition again, every tag match verified
    //      before it is trusted -- a slot then stands for ONE content, and different contents with one tag sit in different slots
    __syncthreads();
    for (unsigned i = tid; i < DP_SLOTS; i += DP_GT) tab[i] = EMPTY;
    __syncthreads();
#pragma unroll
    for (unsigned j = 0; j < IPT; ++j) {
        if (x[j] == NONE) continue;
        const uint32_t tag = (uint32_t)(key[j] >> 32) & tagmask;
        const unsigned long long mine = ((unsigned long long)tag << 32) | x[j];
        const uint64_t gx = gpos[x[j]]; const bool rx = (meta[x[j]] >> 8) & 1;
        unsigned s = (unsigned)(key[j] >> slot_shift) & (DP_SLOTS - 1);
        for (;;) {
            unsigned long long v = __hip_atomic_load(&tab[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (v == EMPTY) { v = atomicCAS(&tab[s], EMPTY, mine); if (v == EMPTY) break; }
            if ((uint32_t)(v >> 32) == tag) {
                const uint32_t xv = (uint32_t)v;
                if (kequal(all, gx, rx, gpos[xv], (meta[xv] >> 8) & 1, q)) { atomicMin(&tab[s], mine); break; }
            }
            s = (s + 1) & (DP_SLOTS - 1);
        }
        slot[j] = s;
    }
    __syncthreads();
#pragma unroll
    for (unsigned j = 0; j < IPT; ++j) {
        if (x[j] == NONE) continue;
        const uint32_t rep = (uint32_t)tab[slot[j]];
        if (rep != x[j]) { grp_rep[x[j]] = rep; atomicOr(&ctx_by_x[rep], (uint32_t)(meta[x[j]] & 0x2FFu)); }
    }
}

// ---- the duplicates verified in RUNS.  Checking every duplicate against its representative inside k3_dict_group reads both K2-mers from wherever
// they lie: 13 M duplicates x 2 x (position, orientation, 57 bytes of sequence) = 12 GB fetched as 128-byte lines for 0.4 GB of pairs, 3.5 ms
// (profiles/r06_pmc_step3_before_runs.md).  But duplicates come in runs: the K2-mers x, x+1, ... of one place repeat y, y+1, ... (or y, y-1, ... read the
// other way) of another.  So k3_dict_group only proposes (mode 1: grouping by tag, no content read) and this kernel, a thread per occurrence in
// POSITION order, checks: a duplicate x of y whose predecessor x-1 is, in the same relative orientation, the duplicate of y's neighbour in the
// stream needs ONE base compared -- the base x adds behind x-1 against the base y adds on that side; every other duplicate (the first of its run)
// is compared in full.  If every check holds, every proposal is exact by induction along the runs; the loads of neighbouring lanes are neighbours.
// A failed check (two K2-mers under one tag) sends the caller to the grouping that verifies every tag match (mode 2).
__global__ void __launch_bounds__(256) k3_verify_runs(uint64_t N2, KGeom q, const uint8_t* __restrict__ all, const uint64_t* __restrict__ gpos, const uint16_t* __restrict__ meta,
                                                       const uint32_t* __restrict__ grp_rep, uint32_t* __restrict__ bad) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= N2) return;
    const uint32_t y = grp_rep[x];
    if (y == (uint32_t)x) return;
    const uint64_t gx = gpos[x], gy = gpos[y];
    const bool rx = (meta[x] >> 8) & 1, ry = (meta[y] >> 8) & 1;
    bool ok = false, decided = false;
    if (x > 0 && gpos[x - 1] + 1 == gx) {                           // x-1 is the K2-mer in front of x in the same place
        const uint32_t yp = grp_rep[x - 1];
        const bool rel = rx == ry, relp = (((meta[x - 1] >> 8) & 1) != 0) == (((meta[yp] >> 8) & 1) != 0);
        const uint64_t gp = gpos[yp];
        if (rel == relp && (rel ? gp + 1 == gy : gp == gy + 1)) {   // ... and (the duplicate of) y's neighbour on the matching side
            const unsigned bx = stream1(all, gx + q.K2 - 1);
            ok = rel ? bx == stream1(all, gy + q.K2 - 1) : bx == 3u - stream1(all, gy);
            decided = true;
        }
    }
    if (!decided) ok = kequal(all, gx, rx, gy, ry, q);
    if (!ok) *bad = 1u;
}

// every occurrence learns the representative (first) occurrence of its group and the contexts are ORed into the representative's word.
// An occurrence that is alone in its group (most of them) is its own representative: k3_rep_init writes that in position order
// (streaming), and only members of larger groups pay a random 4-byte scatter and an atomic.
__global__ void __launch_bounds__(256) k3_rep_init(uint64_t N2, const uint16_t* __restrict__ meta, uint32_t* __restrict__ grp_rep, uint32_t* __restrict__ ctx_by_x) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x < N2) { grp_rep[x] = (uint32_t)x; ctx_by_x[x] = meta[x] & 0x2FFu; }
}
__global__ void __launch_bounds__(256) k3_scatter_rep(uint64_t N2, const uint32_t* __restrict__ head, const uint32_t* __restrict__ hidx, const uint32_t* __restrict__ over,
                                                       const uint32_t* __restrict__ val, const uint16_t* __restrict__ meta, uint32_t* __restrict__ grp_rep,
                                                       uint32_t* __restrict__ ctx_by_x) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N2) return;
    uint64_t hj = (uint64_t)hidx[j] - 1;
    if (over && over[j] != NONE) hj = over[j];
    if (hj == j) return;                                     // a representative: initialised in place; its members OR into it
    const uint32_t x = val[j], rx = val[hj];
    grp_rep[x] = rx;
    atomicOr(&ctx_by_x[rx], (uint32_t)(meta[x] & 0x2FFu));
}
// ids in POSITION order of the representatives (see below); streaming except for the duplicates' gather of their representative's id
__global__ void __launch_bounds__(256) k3_finish_ids(uint64_t N2, const uint32_t* __restrict__ grp_rep, const uint64_t* __restrict__ pid, const uint32_t* __restrict__ ctx_by_x,
                                                      uint32_t* __restrict__ id_of, uint32_t* __restrict__ rep, uint32_t* __restrict__ dctx) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= N2) return;
    const uint32_t rx = grp_rep[x];
    const uint32_t id = (uint32_t)pid[rx];
    id_of[x] = id;
    if (rx == (uint32_t)x) { rep[id] = (uint32_t)x; dctx[id] = ctx_by_x[x]; }
}
// replay mode only: (hash, id) of the distinct K2-mers in sorted order, for the lookup of the hinted edges
__global__ void __launch_bounds__(256) k3_head_list(uint64_t N2, const uint32_t* __restrict__ head, const uint64_t* __restrict__ hex, const uint64_t* __restrict__ key,
                                                     const uint32_t* __restrict__ val, const uint32_t* __restrict__ id_of, uint64_t* __restrict__ dhash, uint32_t* __restrict__ did) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N2 || !head[j]) return;
    dhash[hex[j]] = key[j]; did[hex[j]] = id_of[val[j]];
}
__global__ void __launch_bounds__(256) k3_nonzero(uint64_t n, const uint32_t* __restrict__ a, uint32_t* __restrict__ f) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f[i] = a[i] ? 1u : 0u;
}

// Distinct K2-mers are numbered by the POSITION of their first occurrence, not by their hash: consecutive K2-mers of a place then
// get consecutive ids, a unipath mostly runs through one tile of the list ranking (k_rank_tiles resolves it in LDS; numbered by
// hash every node was a splitter and the pointer jumping over all of them took 20 ms for 49 M K2-mers), and the per-k-mer
// arrays are read and written almost sequentially along the places.
// ============================================================================= unipaths (BigKPather.cc:110-310)
// successor of every oriented occurrence = the next occurrence of its place
__global__ void __launch_bounds__(256) k3_nbr(uint64_t N2, uint64_t U, const uint64_t* __restrict__ koff, const uint32_t* __restrict__ id_of, const uint16_t* __restrict__ meta,
                                               uint32_t* __restrict__ nbr, const uint32_t* __restrict__ blk) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t u = upper_index_seq(koff, U, x < N2 ? x : N2 - 1, (uint64_t)blockIdx.x * blockDim.x, blk);
    if (x + 1 >= N2) return;
    if (x + 1 >= koff[u + 1]) return;                                                 // last K2-mer of its place
    const uint32_t a = 2 * id_of[x] + ((meta[x] >> 8) & 1), b = 2 * id_of[x + 1] + ((meta[x + 1] >> 8) & 1);
    nbr[a] = b; nbr[b ^ 1u] = a ^ 1u;
}
// buildEdge :181-199 with upstream/downstreamExtensionPossible :201-224 (the same port rule as Step 2's k_links)
__global__ void __launch_bounds__(256) k3_links(uint64_t D, const uint32_t* __restrict__ dctx, const uint32_t* __restrict__ nbr, uint32_t* __restrict__ nxt0) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    uint32_t n0 = NONE, n1 = NONE;
    const uint32_t ci = dctx[i];
    if (!(ci & 512u)) {                                                                // a palindrome is a 1-k-mer edge
        const unsigned c = ci & 0xFF;
        if (popc4(c & 15) == 1) {                                                      // downstreamExtensionPossible
            const uint32_t s = nbr[2 * i];
            if (s != NONE) {
                const uint32_t cs = dctx[s >> 1]; unsigned cj = cs & 0xFF; if (s & 1) cj = brev8(cj);
                if (!(cs & 512u) && popc4(cj >> 4) == 1) n0 = s;
            }
        }
        if (popc4(c >> 4) == 1) {                                                      // upstreamExtensionPossible
            const uint32_t w = nbr[2 * i + 1];                                         // successor of the flipped node = flip(predecessor)
            if (w != NONE) {
                const uint32_t cp = dctx[w >> 1]; unsigned cj = cp & 0xFF; if (!(w & 1)) cj = brev8(cj);     // context of the predecessor node w^1
                if (!(cp & 512u) && popc4(cj & 15) == 1) n1 = w;
            }
        }
    }
    nxt0[2 * i] = n0; nxt0[2 * i + 1] = n1;
}
// oriented node v = 2*id + o: the K2-mer content is the representative occurrence, flipped when o differs from its orientation
struct KSrc { const uint8_t* all; const uint64_t* gpos /* stream position of every occurrence */; const uint32_t* rep; const uint16_t* meta; KGeom q; };
__device__ inline void node_loc(const KSrc& S, uint32_t v, uint64_t* g, bool* rc) {
    const uint32_t x = S.rep[v >> 1];
    *g = S.gpos[x];
    *rc = (((S.meta[x] >> 8) & 1u) != 0) != ((v & 1u) != 0);
}
// middle base of the odd-length unipaths (even number of K2-mers), as seen from each head (bvec::getCanonicalForm, feudal/BaseVec.h:326)
__global__ void __launch_bounds__(256) k3_mid(uint64_t D, KSrc S, const uint32_t* __restrict__ nxt, const uint32_t* __restrict__ rnk, uint8_t* __restrict__ mid) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    const uint32_t r0 = rnk[2 * i], r1 = rnk[2 * i + 1];
    const uint64_t n = (uint64_t)r0 + r1 + 1;
    if (n & 1) return;                                   // even number of bases (K2 is even): decided by the end k-mers
    const uint64_t qm = n / 2 + (S.q.K2 / 2 - 1);        // (n + K2 - 1) / 2
    const uint64_t x = qm < n - 1 ? qm : n - 1;
    if (r1 != x && r0 != x) return;
    const unsigned off = (unsigned)(qm - x);
    uint64_t g; bool rc;
    node_loc(S, (uint32_t)(2 * i), &g, &rc);
    if (r1 == x) mid[nxt[2 * i + 1] ^ 1u] = (uint8_t)kbase(S.all, g, S.q, rc, off);         // traversed forward
    if (r0 == x) mid[nxt[2 * i] ^ 1u] = (uint8_t)kbase(S.all, g, S.q, !rc, off);           // traversed reversed
}
// circles (simpleCircle :126-153, canonicalizeCircle :156-180): min-jumping over the canonical contents
__global__ void __launch_bounds__(256) k3_minjump_init(uint64_t N, const uint32_t* __restrict__ nxt0, const uint8_t* __restrict__ cyc, uint32_t* __restrict__ nx, uint32_t* __restrict__ mn) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    nx[v] = cyc[v] ? nxt0[v] : (uint32_t)v;
    mn[v] = (uint32_t)(v >> 1);
}
__global__ void __launch_bounds__(256) k3_minjump(uint64_t N, KSrc S, const uint32_t* __restrict__ nx, const uint32_t* __restrict__ mn, uint32_t* __restrict__ nx2, uint32_t* __restrict__ mn2) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const uint32_t a = nx[v];
    const uint32_t m0 = mn[v], m1 = mn[a];
    uint32_t best = m0;
    if (m0 != m1) {
        uint64_t g0, g1; bool r0, r1;
        node_loc(S, 2 * m0, &g0, &r0); node_loc(S, 2 * m1, &g1, &r1);
        if (kcmp(S.all, g1, r1, g0, r0, S.q) < 0) best = m1;
    }
    mn2[v] = best; nx2[v] = nx[a];
}
__global__ void __launch_bounds__(256) k3_cycle_cut(uint64_t D, const uint8_t* __restrict__ cyc, const uint32_t* __restrict__ mn, uint32_t* __restrict__ nxt0) {
    const uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= D) return;
    if (cyc[2 * m] && mn[2 * m] == (uint32_t)m) {
        const uint32_t u = nxt0[2 * m + 1];
        if (u != NONE) nxt0[u ^ 1u] = NONE;
        nxt0[2 * m + 1] = NONE;
    }
}
// canonical heads (extend :253-258 keeps an edge iff its sequence is not REV) with the words of their first K2-mer as sort key
// Two kernels: the heads are one node in ~270 and the decision is a chain of ~10 dependent loads (two nodes located, K2-mers compared word by
// word) -- inside the streaming kernel a wavefront waited for its one or two heads (2.0 ms at 272 M nodes); listed first and decided by a dense
// launch, every lane of a wavefront waits at once.
constexpr unsigned HC_PER = 16;                                   // nodes per thread: one reservation per block of 4096 nodes (one per wavefront: ~2 M additions to one address)
__global__ void __launch_bounds__(256) k3_head_cands(uint64_t N, const uint32_t* __restrict__ nxt0, uint8_t* __restrict__ is_head, uint32_t* __restrict__ cand,
                                                      unsigned long long* __restrict__ n_cand, uint64_t cap) {
    __shared__ uint32_t s_n; __shared__ unsigned long long s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint64_t v0 = (uint64_t)blockIdx.x * (256 * HC_PER) + threadIdx.x;
    uint32_t at[HC_PER];
#pragma unroll
    for (unsigned i = 0; i < HC_PER; ++i) {
        const uint64_t v = v0 + (uint64_t)i * 256;
        at[i] = NONE;
        if (v < N) {
            is_head[v] = 0;
            if (nxt0[v ^ 1] == NONE) at[i] = atomicAdd(&s_n, 1u);          // the reverse of v is a chain end <=> v is a head
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_n) s_base = atomicAdd(n_cand, (unsigned long long)s_n);
    __syncthreads();
#pragma unroll
    for (unsigned i = 0; i < HC_PER; ++i)
        if (at[i] != NONE && s_base + at[i] < cap) cand[s_base + at[i]] = (uint32_t)(v0 + (uint64_t)i * 256);
}
__global__ void __launch_bounds__(256) k3_heads(const unsigned long long* __restrict__ n_cand, uint64_t cap, const uint32_t* __restrict__ cand, KSrc S,
                                                 const uint32_t* __restrict__ dctx, const uint32_t* __restrict__ nxt, const uint32_t* __restrict__ rnk,
                                                 const uint8_t* __restrict__ mid, uint8_t* __restrict__ is_head, uint32_t* __restrict__ head_v,
                                                 unsigned long long* __restrict__ n_heads) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nc = *n_cand < cap ? *n_cand : cap;
    bool canon = false; uint32_t v = 0;
    if (i < nc) {
        v = cand[i];
        const uint64_t n = (uint64_t)rnk[v] + 1;
        if (dctx[v >> 1] & 512u) canon = !(v & 1);                // PALINDROME: one object
        else if (n & 1) {                                         // even number of bases: first K2-mer against the first K2-mer of the RC
            uint64_t g0, g1; bool r0, r1;
            node_loc(S, v, &g0, &r0); node_loc(S, nxt[v] ^ 1u, &g1, &r1);
            canon = kcmp(S.all, g0, r0, g1, r1, S.q) < 0;
        } else canon = !(mid[v] & 2);                             // odd number of bases: middle base A/C
        if (canon) is_head[v] = 1;
    }
    const unsigned long long m = __ballot(canon);
    if (m) {
        const unsigned lane = threadIdx.x & 63;
        unsigned long long base = 0;
        if (lane == (unsigned)__builtin_ctzll(m)) base = atomicAdd(n_heads, (unsigned long long)__builtin_popcountll(m));
        base = __shfl(base, __builtin_ctzll(m));
        if (canon) head_v[base + (unsigned long long)__builtin_popcountll(m & ((1ull << lane) - 1))] = v;       // (heads <= candidates <= cap)
    }
}
__global__ void __launch_bounds__(256) k3_head_word(uint64_t E, KSrc S, const uint32_t* __restrict__ head_v, const uint32_t* __restrict__ perm, unsigned j, uint64_t* __restrict__ out) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    uint64_t g; bool rc; node_loc(S, head_v[perm[e]], &g, &rc);
    out[e] = kword(S.all, g, S.q, rc, j);
}
// the heads sorted by the FIRST word of their K2-mers: runs of equal first words (unipaths that leave one vertex share K2-1 bases) are put
// in full lexicographic order by insertion, one thread per run; a run longer than `max_run` is left to the caller's word-by-word sort
__global__ void __launch_bounds__(256) k3_edge_tie_sort(uint64_t E, KSrc S, const uint32_t* __restrict__ head_v, const uint64_t* __restrict__ w0, uint32_t* __restrict__ perm,
                                                         unsigned max_run, uint32_t* __restrict__ flags) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= E || (j > 0 && w0[j] == w0[j - 1])) return;          // not the first of its run
    uint64_t b = j + 1;
    while (b < E && w0[b] == w0[j]) ++b;
    const uint64_t n = b - j;
    if (n == 1) return;
    if (n > max_run) { atomicOr(&flags[6], 1u); return; }
    for (uint64_t i = 1; i < n; ++i) {
        const uint32_t x = perm[j + i];
        uint64_t gx; bool rx; node_loc(S, head_v[x], &gx, &rx);
        uint64_t t = i;
        while (t > 0) {
            uint64_t gy; bool ry; node_loc(S, head_v[perm[j + t - 1]], &gy, &ry);
            if (kcmp(S.all, gy, ry, gx, rx, S.q) <= 0) break;
            perm[j + t] = perm[j + t - 1]; --t;
        }
        perm[j + t] = x;
    }
}
__global__ void __launch_bounds__(256) k3_edge_from_sorted(uint64_t E, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ head_v, const uint32_t* __restrict__ rnk,
                                                            uint32_t* __restrict__ head_edge, uint32_t* __restrict__ edge_head, uint32_t* __restrict__ edge_nk) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const uint32_t v = head_v[perm[e]];
    head_edge[v] = (uint32_t)e; edge_head[e] = v; edge_nk[e] = rnk[v] + 1;
}
// replay: hint edge e -> the head whose first K2-mer it starts with (dictionary lookup = binary search in the sorted hashes)
__global__ void __launch_bounds__(256) k3_edge_from_hint(uint64_t E, uint64_t D, KSrc S, const uint8_t* __restrict__ hbits, const uint64_t* __restrict__ hbase0,
                                                          const uint32_t* __restrict__ hlen, const uint64_t* __restrict__ dhash /* ascending */,
                                                          const uint32_t* __restrict__ did /* id of the dhash entries */, const uint8_t* __restrict__ is_head,
                                                          const uint32_t* __restrict__ rnk, uint32_t* __restrict__ head_edge, uint32_t* __restrict__ edge_head,
                                                          uint32_t* __restrict__ edge_nk, uint32_t* __restrict__ flags) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    edge_head[e] = 0; edge_nk[e] = 1;
    const uint64_t g = hbase0[e];
    uint64_t hf = 0x6A09E667F3BCC908ull, hr = hf; int cmp = 0;
    for (unsigned j = 0; j < S.q.NW; ++j) {
        const uint64_t f = kword_f(hbits, g, S.q, j), r = kword_r(hbits, g, S.q, j);
        if (cmp == 0 && f != r) cmp = r < f ? -1 : 1;
        hf = mix64(hf, f); hr = mix64(hr, r);
    }
    const bool hrc = cmp < 0; const uint64_t h = sort_rot(hrc ? hr : hf, S.q.sbits), smask = sort_mask(S.q.sbits);
    uint64_t a = 0, b = D;
    while (a < b) { const uint64_t m = (a + b) >> 1; if ((dhash[m] & smask) < (h & smask)) a = m + 1; else b = m; }
    uint32_t id = NONE;
    for (; a < D && same_run(dhash[a], h, S.q.sbits); ++a) {                                           // the run of this sort key: equal hashes, then contents
        if (dhash[a] != h) continue;
        uint64_t gd; bool rd; node_loc(S, 2 * did[a], &gd, &rd);
        bool same = true;
        for (unsigned j = 0; same && j < S.q.NW; ++j) same = kword(hbits, g, S.q, hrc, j) == kword(S.all, gd, S.q, rd, j);
        if (same) { id = did[a]; break; }
    }
    if (id == NONE) { atomicOr(&flags[1], 16u); return; }
    const uint32_t v = 2 * id + (hrc ? 1u : 0u);
    if (!is_head[v]) { atomicOr(&flags[1], 16u); return; }
    if (hlen[e] != rnk[v] + S.q.K2) { atomicOr(&flags[1], 64u); return; }
    const uint32_t old = atomicExch(&head_edge[v], (uint32_t)e);
    if (old != NONE) atomicOr(&flags[1], 32u);
    edge_head[e] = v; edge_nk[e] = rnk[v] + 1;
}
__global__ void __launch_bounds__(256) k3_edge_len(uint64_t E, unsigned K2, const uint32_t* __restrict__ edge_nk, uint32_t* __restrict__ len) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) len[e] = edge_nk[e] + (K2 - 1);
}
// addEdge :275-306 / updateDict :78-88: every K2-mer learns (edge, offset, lies reversed on it) and deposits its base(s)
__global__ void __launch_bounds__(256) k3_assign(uint64_t D, KSrc S, const uint32_t* __restrict__ nxt, const uint32_t* __restrict__ rnk, const uint32_t* __restrict__ head_edge,
                                                  const uint64_t* __restrict__ edge_off, uint32_t* __restrict__ k_edge /* edge | rev << 31 */, uint32_t* __restrict__ k_off,
                                                  uint8_t* __restrict__ codes, uint32_t* __restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    const uint32_t h0 = nxt[2 * i + 1] ^ 1u, h1 = nxt[2 * i] ^ 1u;
    uint32_t e = head_edge[h0], off = rnk[2 * i + 1];
    bool rev = false;
    if (e == NONE) { e = head_edge[h1]; off = rnk[2 * i]; rev = true; }
    if (e == NONE) { atomicOr(&flags[1], 128u); k_edge[i] = NONE; k_off[i] = 0; return; }
    k_edge[i] = e | (rev ? 0x80000000u : 0u); k_off[i] = off;
    uint64_t g; bool rc; node_loc(S, (uint32_t)(2 * i + (rev ? 1 : 0)), &g, &rc);
    uint8_t* dst = codes + edge_off[e];
    if (off == 0) { for (unsigned t = 0; t < S.q.K2; ++t) dst[t] = (uint8_t)kbase(S.all, g, S.q, rc, t); }
    else dst[S.q.K2 - 1 + off] = (uint8_t)kbase(S.all, g, S.q, rc, S.q.K2 - 1);
}

// ============================================================================= buildHBVFromEdges (HBVFromEdges.cc:76-154)
__global__ void __launch_bounds__(256) k3_edge_nobj(uint64_t E, const uint32_t* __restrict__ edge_head, const uint32_t* __restrict__ edge_nk, const uint32_t* __restrict__ dctx,
                                                     uint32_t* __restrict__ nobj) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    nobj[e] = (edge_nk[e] == 1 && (dctx[edge_head[e] >> 1] & 512u)) ? 1u : 2u;           // :94,142: a palindromic edge is one object
}
__global__ void __launch_bounds__(256) k3_edge_xlat(uint64_t E, const uint32_t* __restrict__ nobj, const uint64_t* __restrict__ ooff, int32_t* __restrict__ fwdX,
                                                     int32_t* __restrict__ revX, uint32_t* __restrict__ obj_edge, int32_t* __restrict__ inv2) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const uint64_t o = ooff[e];
    fwdX[e] = (int32_t)o; obj_edge[o] = (uint32_t)(e << 1);
    if (nobj[e] == 2) { revX[e] = (int32_t)(o + 1); obj_edge[o + 1] = (uint32_t)(e << 1) | 1u; inv2[o] = (int32_t)(o + 1); inv2[o + 1] = (int32_t)o; }
    else { revX[e] = (int32_t)o; inv2[o] = (int32_t)o; }
}
__device__ inline unsigned obj_base(const uint8_t* codes, uint64_t eoff, uint32_t len, bool rc, uint32_t t) { return rc ? 3u - codes[eoff + (len - 1 - t)] : codes[eoff + t]; }
// eight consecutive base codes of an oriented object, positions t .. t+7 (all inside the object), code t in the low byte: one unaligned
// 8-byte load; against the stored orientation the bytes are reversed and complemented
__device__ inline uint64_t obj_base8(const uint8_t* codes, uint64_t eoff, uint32_t len, bool rc, uint32_t t) {
    uint64_t w;
    if (!rc) { __builtin_memcpy(&w, codes + eoff + t, 8); return w; }
    __builtin_memcpy(&w, codes + eoff + (len - 8 - t), 8);
    return 0x0303030303030303ull - __builtin_bswap64(w);
}
// one thread per edge end: FNV1a over the K2-1 base codes (math/Hash.h:26-35)
__global__ void __launch_bounds__(256) k3_end_hash(uint64_t NO, unsigned K2, const uint32_t* __restrict__ obj_edge, const uint64_t* __restrict__ edge_off,
                                                    const uint32_t* __restrict__ edge_nk, const uint8_t* __restrict__ codes, uint64_t* __restrict__ ehash) {
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 2 * NO) return;
    const uint64_t o = id >> 1; const bool distal = id & 1;
    const uint32_t oe = obj_edge[o], e = oe >> 1; const bool rc = oe & 1;
    const uint32_t len = edge_nk[e] + (K2 - 1);
    const uint32_t t0 = distal ? len - (K2 - 1) : 0;
    const uint64_t eo = edge_off[e];
    uint64_t h = 14695981039346656037ull;
    unsigned t = 0;
    for (; t + 8 <= K2 - 1; t += 8) {
        uint64_t w = obj_base8(codes, eo, len, rc, t0 + t);
#pragma unroll
        for (unsigned k = 0; k < 8; ++k) { h = 1099511628211ull * (h ^ (w & 0xFFu)); w >>= 8; }
    }
    for (; t < K2 - 1; ++t) h = 1099511628211ull * (h ^ obj_base(codes, eo, len, rc, t0 + t));
    ehash[id] = h;
}
// all words (32 bases each, MSB first, zero padded) of every end's K2-1 bases: words[j * n + id]
__global__ void __launch_bounds__(256) k3_end_words(uint64_t n, unsigned K2, unsigned EW, const uint32_t* __restrict__ obj_edge, const uint64_t* __restrict__ edge_off,
                                                     const uint32_t* __restrict__ edge_nk, const uint8_t* __restrict__ codes, uint64_t* __restrict__ words) {
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const uint64_t o = id >> 1; const bool distal = id & 1;
    const uint32_t oe = obj_edge[o], e = oe >> 1; const bool rc = oe & 1;
    const uint32_t len = edge_nk[e] + (K2 - 1);
    const uint32_t t0 = distal ? len - (K2 - 1) : 0;
    const uint64_t eo = edge_off[e];
    for (unsigned j = 0; j < EW; ++j) {
        uint64_t w = 0;
        for (unsigned g = 0; g < 4; ++g) {                       // eight bases at a time
            const unsigned p = 32 * j + 8 * g;
            if (p + 8 <= K2 - 1) {
                uint64_t b = obj_base8(codes, eo, len, rc, t0 + p);
                // codes 0..3 in 8 bytes -> 16 bits, first base most significant
                b = (b | (b >> 6)) & 0x000F000F000F000Full;              // pairs: byte 2i | byte 2i+1 << 2  (low base in the low bits)
                b = (b | (b >> 12)) & 0x000000FF000000FFull;
                b = (b | (b >> 24)) & 0xFFFFull;                          // 16 bits, base p in bits 1:0 ... base p+7 in bits 15:14
                uint32_t r16 = (uint32_t)b;                               // reverse the eight 2-bit groups: base p most significant
                r16 = ((r16 & 0x3333u) << 2) | ((r16 >> 2) & 0x3333u);
                r16 = ((r16 & 0x0F0Fu) << 4) | ((r16 >> 4) & 0x0F0Fu);
                r16 = ((r16 & 0x00FFu) << 8) | ((r16 >> 8) & 0x00FFu);
                w = (w << 16) | r16;
            } else {
                for (unsigned t = 0; t < 8; ++t) { const unsigned q = p + t; w = (w << 2) | (q < K2 - 1 ? obj_base(codes, eo, len, rc, t0 + q) : 0u); }
            }
        }
        words[(uint64_t)j * n + id] = w;
    }
}
__global__ void __launch_bounds__(256) k3_end_differs(uint64_t n, const uint64_t* __restrict__ w, uint32_t* __restrict__ flag, bool first) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t d = (j > 0 && w[j] != w[j - 1]) ? 1u : 0u;
    flag[j] = first ? d : (flag[j] | d);
}
// the ends sorted by their hash alone: a vertex boundary wherever the hash changes; equal hashes are checked word by word, and ends of
// different content under one hash (2^-64 per pair; the run may hold them interleaved) send the caller to the sort by (hash, sequence)
__global__ void __launch_bounds__(256) k3_end_group(uint64_t n, unsigned EW, const uint64_t* __restrict__ shash, const uint32_t* __restrict__ perm,
                                                     const uint64_t* __restrict__ words /* [EW][n] by end id */, uint32_t* __restrict__ flag, uint32_t* __restrict__ flags,
                                                     bool pretend_collision) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t d = 0;
    if (j > 0) {
        if (shash[j] != shash[j - 1]) d = 1;
        else {
            const uint32_t a = perm[j], b = perm[j - 1];
            bool same = !pretend_collision;
            for (unsigned w = 0; same && w < EW; ++w) same = words[(uint64_t)w * n + a] == words[(uint64_t)w * n + b];
            if (!same) atomicOr(&flags[7], 1u);
        }
    }
    flag[j] = d;
}
__global__ void __launch_bounds__(256) k3_end_vertices(uint64_t n, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ flag, const uint64_t* __restrict__ excl,
                                                        int32_t* __restrict__ left, int32_t* __restrict__ right) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t vid = (int32_t)(excl[j] + flag[j]);
    const uint32_t id = perm[j];
    if (id & 1) right[id >> 1] = vid; else left[id >> 1] = vid;
}
__global__ void __launch_bounds__(256) k3_adj_keys(uint64_t NO, const int32_t* __restrict__ a, const int32_t* __restrict__ b, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                    uint32_t* __restrict__ deg) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= NO) return;
    keys[o] = ((uint64_t)(uint32_t)a[o] << 32) | (uint32_t)b[o];
    vals[o] = (uint32_t)o;
    atomicAdd(&deg[a[o]], 1u);
}
__global__ void __launch_bounds__(256) k3_adj_out(uint64_t NO, const uint32_t* __restrict__ vals, const int32_t* __restrict__ other, int32_t* __restrict__ out_v, int32_t* __restrict__ out_e) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= NO) return;
    const uint32_t o = vals[j];
    out_e[j] = (int32_t)o; out_v[j] = other[o];
}
__global__ void __launch_bounds__(256) k3_obj_len(uint64_t NO, unsigned K2, const uint32_t* __restrict__ obj_edge, const uint32_t* __restrict__ edge_nk, uint32_t* __restrict__ len, uint32_t* __restrict__ nbytes) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= NO) return;
    const uint32_t l = edge_nk[obj_edge[o] >> 1] + (K2 - 1);
    len[o] = l; nbytes[o] = (l + 3) >> 2;
}
__global__ void __launch_bounds__(256) k3_pack_objs(uint64_t total_bytes, uint64_t NO, unsigned K2, const uint64_t* __restrict__ byte_off, const uint32_t* __restrict__ obj_edge,
                                                     const uint32_t* __restrict__ edge_nk, const uint64_t* __restrict__ edge_off, const uint8_t* __restrict__ codes, uint8_t* __restrict__ out,
                                                     const uint32_t* __restrict__ blk) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t o = upper_index_seq(byte_off, NO, i < total_bytes ? i : total_bytes - 1, (uint64_t)blockIdx.x * blockDim.x, blk);
    if (i >= total_bytes) return;
    const uint32_t oe = obj_edge[o], e = oe >> 1; const bool rc = oe & 1;
    const uint32_t len = edge_nk[e] + (K2 - 1);
    const uint32_t t0 = (uint32_t)(i - byte_off[o]) * 4;
    const uint64_t eo = edge_off[e];
    unsigned v = 0;
    if (t0 + 4 <= len) {                                       // four codes in one unaligned load (reversed and complemented against the stored strand)
        uint32_t w;
        if (!rc) __builtin_memcpy(&w, codes + eo + t0, 4);
        else { __builtin_memcpy(&w, codes + eo + (len - 4 - t0), 4); w = 0x03030303u - __builtin_bswap32(w); }
        v = (w | (w >> 6) | (w >> 12) | (w >> 18)) & 0xFFu;
    } else for (unsigned j = 0; j < 4; ++j) if (t0 + j < len) v |= obj_base(codes, eo, len, rc, t0 + j) << (2 * j);
    out[i] = (uint8_t)v;
}

// ============================================================================= places through the graph (Pather :321-357; Repath.cc:150-214)
// every occurrence: the edge object it lies on in the place's direction and its offset there; an occurrence starts a path element
// iff it is the place's first or lies at offset 0 of its object
__global__ void __launch_bounds__(256) k3_occ(uint64_t N2, uint64_t U, const uint64_t* __restrict__ koff, const uint32_t* __restrict__ id_of, const uint16_t* __restrict__ meta,
                                               const uint32_t* __restrict__ k_edge, const uint32_t* __restrict__ k_off, const uint32_t* __restrict__ edge_nk,
                                               const int32_t* __restrict__ fwdX, const int32_t* __restrict__ revX, uint32_t* __restrict__ start, int32_t* __restrict__ obj,
                                               int32_t* __restrict__ starts, int32_t* __restrict__ stops, const uint32_t* __restrict__ blk) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t u = upper_index_seq(koff, U, x < N2 ? x : N2 - 1, (uint64_t)blockIdx.x * blockDim.x, blk);
    if (x >= N2) return;
    const uint32_t id = id_of[x], ke = k_edge[id], e = ke & 0x7FFFFFFFu;
    const bool against = (((meta[x] >> 8) & 1u) != 0) != ((ke >> 31) != 0);            // the place runs against the edge's stored orientation
    const uint32_t nk = edge_nk[e];
    const uint32_t off = against ? nk - 1 - k_off[id] : k_off[id];
    const bool first = x == koff[u], last = x + 1 == koff[u + 1];
    start[x] = (first || off == 0) ? 1u : 0u;
    obj[x] = against ? revX[e] : fwdX[e];
    if (first) starts[u] = (int32_t)off;
    if (last) stops[u] = (int32_t)(nk - 1 - off);
}
__global__ void __launch_bounds__(256) k3_place_paths(uint64_t N2, uint64_t U, const uint64_t* __restrict__ koff, const uint32_t* __restrict__ start, const uint64_t* __restrict__ excl,
                                                       const int32_t* __restrict__ obj, int32_t* __restrict__ ipath, uint64_t* __restrict__ ioff, const uint32_t* __restrict__ blk) {
    const uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t u = upper_index_seq(koff, U, x < N2 ? x : N2 - 1, (uint64_t)blockIdx.x * blockDim.x, blk);
    if (x >= N2) return;
    if (start[x]) ipath[excl[x]] = obj[x];
    if (x == koff[u]) ioff[u] = excl[x];
}
// Repath.cc:216-249
__global__ void __launch_bounds__(256) k3_read_counts(uint64_t n, const uint8_t* __restrict__ st, const uint32_t* __restrict__ place_of_read, const uint64_t* __restrict__ ioff,
                                                       uint32_t* __restrict__ cnt) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint32_t c = 0;
    if (st[r]) { const uint32_t u = place_of_read[r]; c = (uint32_t)(ioff[u + 1] - ioff[u]); }
    cnt[r] = c;
}
__global__ void __launch_bounds__(256) k3_read_paths(uint64_t n, const uint8_t* __restrict__ st, const uint32_t* __restrict__ place_of_read, const uint64_t* __restrict__ ioff,
                                                      const int32_t* __restrict__ ipath, const int32_t* __restrict__ inv2, const int32_t* __restrict__ p_offset,
                                                      const int32_t* __restrict__ starts, const int32_t* __restrict__ stops, const int32_t* __restrict__ ltrunc,
                                                      const int32_t* __restrict__ rtrunc, const uint64_t* __restrict__ ooff, int32_t* __restrict__ o_offset, int32_t* __restrict__ o_edges) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int32_t off = 0;
    if (st[r]) {
        const uint32_t u = place_of_read[r];
        const uint64_t a = ioff[u]; const uint32_t m = (uint32_t)(ioff[u + 1] - a);
        const bool rc = st[r] == 2;
        off = !rc ? p_offset[r] + starts[u] - ltrunc[u] : p_offset[r] + stops[u] - rtrunc[u];
        int32_t* dst = o_edges + ooff[r];
        if (!rc) for (uint32_t j = 0; j < m; ++j) dst[j] = ipath[a + j];
        else for (uint32_t j = 0; j < m; ++j) dst[j] = inv2[ipath[a + m - 1 - j]];
    }
    o_offset[r] = off;
}
__global__ void __launch_bounds__(256) k3_obj_wordcount(uint64_t NO, const uint32_t* __restrict__ len, uint32_t* __restrict__ nw) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o < NO) nw[o] = (len[o] + 31) / 32;
}
__global__ void __launch_bounds__(256) k3_mul4(uint64_t n, const uint64_t* __restrict__ in, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * 4;
}

// multi-GPU: the other ranks' place paths appended behind the local reads' (offsets shifted by the local edge total)
__global__ void __launch_bounds__(256) k3_shift_off(uint64_t n, const uint64_t* __restrict__ in, uint64_t add, uint64_t* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + add;
}
__global__ void __launch_bounds__(256) k3_check_edges(uint64_t n, const int32_t* __restrict__ e, uint64_t NO, uint32_t* __restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (e[i] < 0 || (uint64_t)e[i] >= NO)) atomicOr(&flags[1], 256u);
}
// W2RAP_STEP3_PLACES_ONLY: the path of every unique place's representative read
__global__ void __launch_bounds__(256) k3_rep_len(uint64_t U, const uint32_t* __restrict__ rep_read, const uint64_t* __restrict__ p_off, uint32_t* __restrict__ len) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u < U) len[u] = (uint32_t)(p_off[rep_read[u] + 1] - p_off[rep_read[u]]);
}
__global__ void __launch_bounds__(256) k3_rep_copy(uint64_t U, const uint32_t* __restrict__ rep_read, const uint64_t* __restrict__ p_off, const int32_t* __restrict__ p_edges,
                                                    const uint64_t* __restrict__ o_off, int32_t* __restrict__ o_edges) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const uint64_t a = p_off[rep_read[u]], m = p_off[rep_read[u] + 1] - a, o = o_off[u];
    for (uint64_t j = 0; j < m; ++j) o_edges[o + j] = p_edges[a + j];
}


std::string g_profile;          // per-kernel times of the last run

// stable LSD sort of `perm` by multi-word keys: word(j) fills `tmp` for the current order, least significant word first
template <class F>
int sort_by_words(Ctx& c, uint32_t* perm, uint64_t n, unsigned nwords, uint64_t* tmp, F word) {
    for (int j = (int)nwords - 1; j >= 0; --j) {
        W2_TRY(word((unsigned)j, tmp));
        W2_TRY(sort_pairs_u64(c, tmp, perm, n, 0, 64));
    }
    return 0;
}

// the per-block starting points of a kernel whose threads are consecutive positions under the offsets a[0 .. n] (k3_block_index)
template <class T>
int block_index(Ctx& c, const T* a, uint64_t n, uint64_t total, uint32_t** out) {
    const uint64_t nblocks = (total + 255) / 256;
    uint32_t* b = nullptr;
    W2_ALLOC(b, uint32_t, nblocks + 2);
    LAUNCH(c, "k3_block_index", k3_block_index<T>, dim3(grid_for(nblocks + 1)), dim3(256), 0, nblocks, a, n, b);
    *out = b;
    return 0;
}

// The dictionary by hash partition (kernels above): grp_rep[x] = first occurrence with x's canonical content, ctx_by_x[rep] = OR of the
// group's contexts.  `overflow` = a bin received more pairs than its fixed capacity: nothing usable was written, the caller sorts instead.
int dict_by_partition(Ctx& c, uint64_t N2 /* pairs */, const KGeom& q, const uint64_t* key, const uint32_t* pos /* null: pair i = occurrence i */, const uint8_t* allb,
                      const uint64_t* gpos, const uint16_t* meta, uint32_t* grp_rep, uint32_t* ctx_by_x, bool& overflow, uint64_t n_occ /* all occurrences */) {
    hipStream_t st = c.stream;
    overflow = false;
    uint64_t avg = DP_AVG, fcap = DP_CAP;
    uint32_t tagmask = 0xFFFFFFFFu;
    if (test_hook("W2RAP_TEST_DICT_AVG")) { const long v = atol(getenv("W2RAP_TEST_DICT_AVG")); if (v >= 1 && v <= (long)DP_AVG) avg = (uint64_t)v; }    // more, smaller partitions
    if (test_hook("W2RAP_TEST_DICT_CAP")) { const long v = atol(getenv("W2RAP_TEST_DICT_CAP")); if (v >= 1 && v <= (long)DP_CAP) fcap = (uint64_t)v; }    // forces the overflow
    if (test_hook("W2RAP_TEST_SORT_BITS") && q.sbits < 32) tagmask = (1u << q.sbits) - 1u;          // few tag bits: nearly every probe meets a foreign K2-mer with its tag
    unsigned maxb = 9;                                             // 512 bins per pass
    if (test_hook("W2RAP_TEST_DICT_PASS_BITS")) { const int v = atoi(getenv("W2RAP_TEST_DICT_PASS_BITS")); if (v >= 1 && v <= 9) maxb = (unsigned)v; }   // three passes on small inputs
    unsigned bits = 0;
    while (((uint64_t)avg << bits) < N2) ++bits;
    if (bits > 3 * maxb) { overflow = true; return 0; }            // (beyond three passes of nine bits: N2 > 2^37, not reachable with 32-bit occurrence ids)
    unsigned nbits[3] = {0, 0, 0}, npass = 1;
    if (bits <= maxb) nbits[0] = bits;
    else if (bits <= 2 * maxb) { nbits[0] = bits - maxb; nbits[1] = maxb; npass = 2; }
    else { nbits[0] = bits - 2 * maxb; nbits[1] = maxb; nbits[2] = maxb; npass = 3; }
    uint32_t* d_ovf = nullptr;
    W2_ALLOC(d_ovf, uint32_t, 4);
    W2_HIP(hipMemsetAsync(d_ovf, 0, 16, st));
    const uint64_t* skey = key; const uint32_t* sx = pos; const uint32_t* scnt = nullptr;
    uint64_t scap = N2, nseg = 1;
    unsigned shift = 0;
    void* to_free[9]; unsigned nfree = 0;
    for (unsigned p = 0; p < npass; ++p) {
        const unsigned nb = 1u << nbits[p];
        const uint64_t nbins = nseg * nb;
        // the last level has the grouping kernel's capacity; the levels before it hold their mean + 3 % + 8192 (uniform hashes: > 20 sigma)
        const uint64_t dcap = p + 1 == npass ? fcap : (N2 / nbins) + (N2 / nbins) / 32 + 8192;
        if ((uint64_t)nb * dcap >= (1ull << 32)) { overflow = true; break; }
        // (no room for the partitions -- they take up to 38 B per occurrence at the last level --: the sorted form needs a third of that)
        uint64_t* dkey = c.alloc<uint64_t>(nbins * dcap + 1);
        uint32_t* dx = dkey ? c.alloc<uint32_t>(nbins * dcap + 1) : nullptr;
        uint32_t* dcnt = dx ? c.alloc<uint32_t>(nbins + 1) : nullptr;
        if (dkey) to_free[nfree++] = dkey;
        if (dx) to_free[nfree++] = dx;
        if (dcnt) to_free[nfree++] = dcnt;
        if (!dcnt) { (void)hipGetLastError(); c.err.clear(); overflow = true; break; }
        W2_HIP(hipMemsetAsync(dcnt, 0, (nbins + 1) * 4, st));
        const uint64_t bps = ((p == 0 ? N2 : scap) + DP_CH - 1) / DP_CH;
        if (nseg * bps >= (1ull << 31)) { overflow = true; break; }
        if (p == 0) LAUNCH(c, "k3_dict_part", k3_dict_part<true>, dim3((unsigned)bps), dim3(DP_T), 0, N2, skey, sx, scnt, scap, (unsigned)bps, shift, nb, dcap, dkey, dx, dcnt, d_ovf);
        else LAUNCH(c, "k3_dict_part", k3_dict_part<false>, dim3((unsigned)(nseg * bps)), dim3(DP_T), 0, N2, skey, sx, scnt, scap, (unsigned)bps, shift, nb, dcap, dkey, dx, dcnt, d_ovf);
        skey = dkey; sx = dx; scnt = dcnt; scap = dcap; nseg = nbins; shift += nbits[p];
    }
    if (!overflow) {
        // the duplicates are proposed by tag and verified in runs behind (k3_verify_runs); W2RAP_STEP3_NO_RUNS=1: verified one by one inside the grouping
        const bool runs = !getenv("W2RAP_STEP3_NO_RUNS");
        LAUNCH(c, "k3_dict_group", k3_dict_group, dim3((unsigned)nseg), dim3(DP_GT), 0, scap, skey, sx, scnt, shift, tagmask, q, allb, gpos, meta, grp_rep, ctx_by_x, runs ? 1 : 0);
        if (runs) LAUNCH(c, "k3_verify_runs", k3_verify_runs, dim3(grid_for(n_occ)), dim3(256), 0, n_occ, q, allb, gpos, meta, (const uint32_t*)grp_rep, d_ovf + 1);
        uint32_t h[2] = {0, 0};
        W2_HIP(hipMemcpyAsync(h, d_ovf, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        overflow = h[0] != 0;
        if (!overflow && h[1]) {                                   // a proposal was wrong (two K2-mers under one tag): from the start, every tag match verified
            if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] step 3 dictionary: a run check failed, regrouping with every tag match verified\n");
            LAUNCH(c, "k3_rep_init", k3_rep_init, dim3(grid_for(n_occ)), dim3(256), 0, n_occ, meta, grp_rep, ctx_by_x);
            LAUNCH(c, "k3_dict_group", k3_dict_group, dim3((unsigned)nseg), dim3(DP_GT), 0, scap, skey, sx, scnt, shift, tagmask, q, allb, gpos, meta, grp_rep, ctx_by_x, 2);
            W2_HIP(hipStreamSynchronize(st));
        }
    } else W2_HIP(hipStreamSynchronize(st));
    for (unsigned i = 0; i < nfree; ++i) c.release(to_free[i]);
    c.release(d_ovf);
    return 0;
}

// the small-K graph's edge objects and the read paths, on the device
struct DevIn { unsigned K; uint64_t NO; const uint8_t* obits /* +32 readable bytes */; const uint64_t* obyte; const uint32_t* olen;
               uint64_t n; const int32_t* p_offset; const uint64_t* p_off; const int32_t* p_edges;
               uint64_t NV; const int32_t* vleft; const int32_t* vright; /* the vertices an edge object leaves / enters: --extend_paths only */
               bool unique_kmers = false; /* every K-mer of the graph occurs once in it (Step 2's own graph, or W2RAP_STEP3_UNIQUE_KMERS) */ };

int step3(Ctx& c, const DevIn& in, const w2rap_step3_params& P, w2rap_step3_out& out) {
    hipStream_t st = c.stream;
    const unsigned K = in.K, K2 = P.K2;
    unsigned sbits = SORT_BITS;
    if (test_hook("W2RAP_TEST_SORT_BITS")) { const int v = atoi(getenv("W2RAP_TEST_SORT_BITS")); if (v >= 1 && v <= 63) sbits = (unsigned)v; }    // results do not depend on it
    const KGeom q{K2, (K2 + 31) / 32, K2 - 32 * ((K2 + 31) / 32 - 1), sbits};
    const uint64_t NO = in.NO, n = in.n;
    uint32_t* d_flags = nullptr;                   // [0] ranking "changed"  [1] error bits  [2] has cycles
    W2_ALLOC(d_flags, uint32_t, 8);
    W2_HIP(hipMemsetAsync(d_flags, 0, 32, st));
    uint32_t h_flags[4] = {0, 0, 0, 0};
    auto check = [&]() -> int {
        W2_HIP(hipMemcpyAsync(h_flags, d_flags, 16, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        const uint32_t f = h_flags[1];
        if (f & 1) { c.err = "Involution: an edge object has no reverse complement in the graph (HyperBasevector.cc:648-660 needs every edge's RC)"; return W2RAP_E_GRAPH; }
        if (f & 2) { c.err = "Involution: the objects of the small-K graph do not pair up with their reverse complements"; return W2RAP_E_GRAPH; }
        if (f & 4) { c.err = "places: two different paths under one 128-bit key"; return W2RAP_E_LIMIT; }
        if (f & 8) { c.err = "a place longer than 2^31 bases"; return W2RAP_E_LIMIT; }
        if (f & 16) { c.err = "edge_order_hint: a hinted edge is not a unipath of this graph"; return W2RAP_E_HINT; }
        if (f & 32) { c.err = "edge_order_hint: an edge is listed twice"; return W2RAP_E_HINT; }
        if (f & 64) { c.err = "edge_order_hint: a hinted edge has the wrong length"; return W2RAP_E_HINT; }
        if (f & 128) { c.err = "K2-mer left without an edge (BigKPather.cc:303)"; return W2RAP_E_GRAPH; }
        if (f & 256) { c.err = "extra_path_edges names an edge object that does not exist"; return W2RAP_E_ARG; }
        if (f & 512) { c.err = "extend_paths: vleft / vright name a vertex that does not exist"; return W2RAP_E_ARG; }
        return 0;
    };
    // ---------------------------------------------------------------- inputs
    Timer t_places(st);
    const uint8_t* obits = in.obits; const uint64_t* obyte = in.obyte; const uint32_t* olen = in.olen;
    const int32_t *p_offset = in.p_offset, *p_edges = in.p_edges; const uint64_t* p_off = in.p_off;
    // multi-GPU: `na` paths enter the places (this rank's n reads, then the other ranks' place paths); only the n reads are translated
    uint64_t na = n;
    if (P.n_extra_paths) {
        const uint64_t ne = P.n_extra_paths, ee = P.extra_path_off[ne];
        if (n + ne >= (1ull << 32) - 2) { c.err = "more than 2^32 paths with the other ranks' places"; return W2RAP_E_LIMIT; }
        uint64_t e_local = 0;
        W2_HIP(hipStreamSynchronize(st));            // (p_off may still be on its way up: the stream does not block the null stream's copies)
        if (n) W2_HIP(hipMemcpy(&e_local, p_off + n, 8, hipMemcpyDeviceToHost));
        uint64_t *off2 = nullptr, *xoff = nullptr; int32_t* edges2 = nullptr;
        W2_ALLOC(off2, uint64_t, n + ne + 2); W2_ALLOC(edges2, int32_t, e_local + ee + 1); W2_ALLOC(xoff, uint64_t, ne + 2);
        if (n) W2_HIP(hipMemcpyAsync(off2, p_off, n * 8, hipMemcpyDeviceToDevice, st));
        if (e_local) W2_HIP(hipMemcpyAsync(edges2, p_edges, e_local * 4, hipMemcpyDeviceToDevice, st));
        W2_HIP(hipMemcpyAsync(xoff, P.extra_path_off, (ne + 1) * 8, hipMemcpyHostToDevice, st));
        if (ee) W2_HIP(hipMemcpyAsync(edges2 + e_local, P.extra_path_edges, ee * 4, hipMemcpyHostToDevice, st));
        LAUNCH(c, "k3_shift_off", k3_shift_off, dim3(grid_for(ne + 1)), dim3(256), 0, ne + 1, xoff, e_local, off2 + n);
        if (ee) LAUNCH(c, "k3_check_edges", k3_check_edges, dim3(grid_for(ee)), dim3(256), 0, ee, edges2 + e_local, NO, d_flags);
        W2_TRY(check());                             // before anything indexes by these edge ids
        p_off = off2; p_edges = edges2; na = n + ne;
    }
    uint64_t* obase0 = nullptr;
    W2_ALLOC(obase0, uint64_t, NO + 1);
    LAUNCH(c, "k3_mul4", k3_mul4, dim3(grid_for(NO + 1)), dim3(256), 0, NO + 1, obyte, obase0);
    // ---------------------------------------------------------------- Involution
    int32_t* inv = nullptr;
    W2_ALLOC(inv, int32_t, NO + 1);
    if (NO) {
        uint64_t *f_hi, *f_lo, *r_hi, *r_lo, *tmp; uint32_t* perm;
        W2_ALLOC(f_hi, uint64_t, NO); W2_ALLOC(f_lo, uint64_t, NO); W2_ALLOC(r_hi, uint64_t, NO); W2_ALLOC(r_lo, uint64_t, NO); W2_ALLOC(tmp, uint64_t, NO);
        W2_ALLOC(perm, uint32_t, NO);
        LAUNCH(c, "k3_obj_ends", k3_obj_ends, dim3(grid_for(NO)), dim3(256), 0, NO, K, obits, obase0, olen, f_hi, f_lo, r_hi, r_lo);
        LAUNCH(c, "k3_iota", k3_iota, dim3(grid_for(NO)), dim3(256), 0, NO, perm);
        LAUNCH(c, "k3_inv_mix", k3_inv_mix, dim3(grid_for(NO)), dim3(256), 0, NO, f_hi, f_lo, tmp);
        W2_TRY(sort_pairs_u64(c, tmp, perm, NO, 0, 64));
        LAUNCH(c, "k3_inv_match", k3_inv_match, dim3(grid_for(NO)), dim3(256), 0, NO, tmp, perm, f_hi, f_lo, r_hi, r_lo, inv, d_flags);
        uint32_t* nw = nullptr; uint64_t* wordoff = nullptr;
        W2_ALLOC(nw, uint32_t, NO); W2_ALLOC(wordoff, uint64_t, NO + 1);
        LAUNCH(c, "k3_obj_wordcount", k3_obj_wordcount, dim3(grid_for(NO)), dim3(256), 0, NO, olen, nw);
        W2_TRY(exclusive_scan_u32_to_u64(c, nw, wordoff, NO));
        uint64_t nwords = 0;
        W2_HIP(hipMemcpy(&nwords, wordoff + NO, 8, hipMemcpyDeviceToHost));
        if (nwords) LAUNCH(c, "k3_inv_verify", k3_inv_verify, dim3(grid_for(nwords)), dim3(256), 0, nwords, NO, wordoff, obits, obase0, olen, inv, d_flags);
        W2_TRY(check());
        for (void* p : {(void*)f_hi, (void*)f_lo, (void*)r_hi, (void*)r_lo, (void*)tmp, (void*)perm, (void*)nw, (void*)wordoff}) c.release(p);
    }
    // ---------------------------------------------------------------- FragDist
    unsigned long long* d_cnt = nullptr;             // [0..99] fragment counts  [100] pathed [101] multipathed [102] heads [103] collisions
    W2_ALLOC(d_cnt, unsigned long long, 112);
    W2_HIP(hipMemsetAsync(d_cnt, 0, 112 * 8, st));
    unsigned long long* d_frag = nullptr;            // FRAG_STRIPES x 128 partial fragment counts
    W2_ALLOC(d_frag, unsigned long long, FRAG_STRIPES * 128);
    W2_HIP(hipMemsetAsync(d_frag, 0, FRAG_STRIPES * 128 * 8, st));
    bool frag_done = false;                          // (FragDist rides in the first k3_place_keys launch)
    // ---------------------------------------------------------------- places
    // (a function of the path set: with --extend_paths it runs twice, the second time with the extended places among the paths)
    uint64_t U = 0, np = 0;                          // unique places; reads (and other ranks' place paths) that have a place
    uint32_t* place_of_read = nullptr; uint32_t* rep_read = nullptr; uint8_t* state = nullptr;
    unsigned long long* d_pcnt = nullptr;            // 64 x (pathed, multipathed, placed)
    auto compute_places = [&](uint64_t na, const uint64_t* p_off, const int32_t* p_edges) -> int {
    uint64_t *keyA, *keyB;
    W2_ALLOC(keyA, uint64_t, na + 1); W2_ALLOC(keyB, uint64_t, na + 1); W2_ALLOC(state, uint8_t, na + 1);
    W2_ALLOC(d_pcnt, unsigned long long, 192);
    W2_HIP(hipMemsetAsync(d_pcnt, 0, 192 * 8, st));
    uint32_t* first1 = nullptr;
    W2_ALLOC(first1, uint32_t, NO + 1);
    W2_HIP(hipMemsetAsync(first1, 0xFF, (NO + 1) * 4, st));
    if (na) LAUNCH(c, "k3_place_keys", k3_place_keys, dim3(grid_for(na)), dim3(256), 0, na, n, K, K2, p_off, p_edges, inv, olen, keyA, keyB, state, first1, d_pcnt,
                   p_offset, frag_done ? (unsigned long long*)nullptr : d_frag);
    if (!frag_done) LAUNCH(c, "k3_frag_sum", k3_frag_sum, dim3(1), dim3(128), 0, (const unsigned long long*)d_frag, d_cnt);
    frag_done = true;
    // ---- places of several edges: compacted, sorted by their 128-bit keys, neighbours verified element by element
    uint32_t* f32 = nullptr; uint64_t* ex = nullptr;
    W2_ALLOC(f32, uint32_t, na + 1); W2_ALLOC(ex, uint64_t, na + 2);
    if (na) LAUNCH(c, "k3_flag_multi", k3_flag_multi, dim3(grid_for(na)), dim3(256), 0, na, state, keyA, f32);
    W2_TRY(exclusive_scan_u32_to_u64(c, f32, ex, na));
    uint64_t npm = 0;
    W2_HIP(hipMemcpy(&npm, ex + na, 8, hipMemcpyDeviceToHost));
    uint32_t* ids = nullptr; uint64_t *kA, *kB;
    W2_ALLOC(ids, uint32_t, npm + 1); W2_ALLOC(kA, uint64_t, npm + 1); W2_ALLOC(kB, uint64_t, npm + 1);
    if (na) LAUNCH(c, "k3_compact_multi", k3_compact_multi, dim3(grid_for(na)), dim3(256), 0, na, f32, ex, keyA, keyB, ids, kA, kB);
    uint64_t U_multi = 0;
    W2_ALLOC(place_of_read, uint32_t, na + 1);
    uint32_t *sid = nullptr, *head = nullptr; uint64_t* hex = nullptr;
    if (npm) {
        // stable sort by kA alone (equal places are then neighbours, the smallest read id first); by (kB, kA) only if two places share a kA
        uint64_t* tmpk = nullptr; W2_ALLOC(tmpk, uint64_t, npm);
        uint32_t* perm = nullptr; W2_ALLOC(perm, uint32_t, npm);
        uint64_t* sB = nullptr; W2_ALLOC(sB, uint64_t, npm);
        W2_ALLOC(sid, uint32_t, npm); W2_ALLOC(head, uint32_t, npm); W2_ALLOC(hex, uint64_t, npm + 1);
        bool both = test_hook("W2RAP_TEST_STEP3_FULL_SORTS");
        for (;;) {
            LAUNCH(c, "k3_iota", k3_iota, dim3(grid_for(npm)), dim3(256), 0, npm, perm);
            if (both) {
                W2_HIP(hipMemcpyAsync(tmpk, kB, npm * 8, hipMemcpyDeviceToDevice, st));
                W2_TRY(sort_pairs_u64(c, tmpk, perm, npm, 0, 64));
                LAUNCH(c, "k3_gather_u64", k3_gather_u64, dim3(grid_for(npm)), dim3(256), 0, npm, kA, perm, tmpk);
            } else W2_HIP(hipMemcpyAsync(tmpk, kA, npm * 8, hipMemcpyDeviceToDevice, st));
            W2_TRY(sort_pairs_u64(c, tmpk, perm, npm, 0, 64));                 // tmpk = sorted kA; perm = order
            LAUNCH(c, "k3_gather_u64", k3_gather_u64, dim3(grid_for(npm)), dim3(256), 0, npm, kB, perm, sB);
            LAUNCH(c, "k3_gather_u32", k3_gather_u32, dim3(grid_for(npm)), dim3(256), 0, npm, ids, perm, sid);
            W2_HIP(hipMemsetAsync(d_flags + 5, 0, 4, st));
            LAUNCH(c, "k3_place_heads", k3_place_heads, dim3(grid_for(npm)), dim3(256), 0, npm, tmpk, sB, sid, state, p_off, p_edges, inv, head, d_flags);
            W2_TRY(exclusive_scan_u32_to_u64(c, head, hex, npm));
            uint32_t shared_key = 0;
            W2_HIP(hipMemcpyAsync(&shared_key, d_flags + 5, 4, hipMemcpyDeviceToHost, st));
            W2_HIP(hipMemcpy(&U_multi, hex + npm, 8, hipMemcpyDeviceToHost));
            W2_HIP(hipStreamSynchronize(st));
            if (shared_key && !both) { both = true; continue; }
            break;
        }
        W2_TRY(check());
        for (void* p : {(void*)tmpk, (void*)perm, (void*)sB}) c.release(p);
    }
    // ---- one-edge places, by direct addressing, numbered behind them
    uint32_t* f1 = nullptr; uint64_t* rank1 = nullptr;
    W2_ALLOC(f1, uint32_t, NO + 1); W2_ALLOC(rank1, uint64_t, NO + 2);
    if (NO) LAUNCH(c, "k3_flag_first1", k3_flag_first1, dim3(grid_for(NO)), dim3(256), 0, NO, first1, f1);
    W2_TRY(exclusive_scan_u32_to_u64(c, f1, rank1, NO));
    uint64_t U1 = 0;
    W2_HIP(hipMemcpy(&U1, rank1 + NO, 8, hipMemcpyDeviceToHost));
    U = U_multi + U1;
    W2_ALLOC(rep_read, uint32_t, U + 1);
    if (npm) LAUNCH(c, "k3_place_index", k3_place_index, dim3(grid_for(npm)), dim3(256), 0, npm, head, hex, sid, place_of_read, rep_read);
    if (NO) LAUNCH(c, "k3_rep_first1", k3_rep_first1, dim3(grid_for(NO)), dim3(256), 0, NO, first1, rank1, U_multi, rep_read);
    if (na) LAUNCH(c, "k3_place_of_one", k3_place_of_one, dim3(grid_for(na)), dim3(256), 0, na, state, keyA, rank1, U_multi, place_of_read);
    np = 0;
    {
        unsigned long long h_pc[192];
        W2_HIP(hipMemcpyAsync(h_pc, d_pcnt, sizeof h_pc, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        for (unsigned k = 0; k < 64; ++k) np += h_pc[3 * k + 2];
    }
    for (void* p : {(void*)keyA, (void*)keyB, (void*)f32, (void*)ex, (void*)ids, (void*)kA, (void*)kB, (void*)sid, (void*)head, (void*)hex, (void*)f1, (void*)rank1, (void*)first1})
        if (p) c.release(p);
    return 0;
    };
    W2_TRY(compute_places(na, p_off, p_edges));
    const uint64_t U_printed = U, np_printed = np;   // what the reference prints (Repath.cc:67-71), before any extension
    if (P.extend_paths && U && !(P.flags & W2RAP_STEP3_PLACES_ONLY)) {
        // ---- Repath.cc:72-96: the extended places join the paths (an empty path for a place that did not grow), the places are computed again
        if (na + U >= (1ull << 32) - 2) { c.err = "more than 2^32 paths with the extended places"; return W2RAP_E_LIMIT; }
        const uint64_t NV = in.NV;
        uint32_t *indeg, *outdeg, *xlen; int32_t *in_e, *out_e, *ext; uint64_t* xoff;
        W2_ALLOC(indeg, uint32_t, NV + 1); W2_ALLOC(outdeg, uint32_t, NV + 1); W2_ALLOC(in_e, int32_t, NV + 1); W2_ALLOC(out_e, int32_t, NV + 1);
        W2_ALLOC(xlen, uint32_t, U + 1); W2_ALLOC(ext, int32_t, 2 * U + 2); W2_ALLOC(xoff, uint64_t, U + 2);
        W2_HIP(hipMemsetAsync(indeg, 0, (NV + 1) * 4, st)); W2_HIP(hipMemsetAsync(outdeg, 0, (NV + 1) * 4, st));
        if (NO) LAUNCH(c, "k3_vertex_deg", k3_vertex_deg, dim3(grid_for(NO)), dim3(256), 0, NO, in.vleft, in.vright, NV, indeg, outdeg, in_e, out_e, d_flags);
        W2_TRY(check());                             // (vertex ids in range) before anything indexes by them
        LAUNCH(c, "k3_extend_len", k3_extend_len, dim3(grid_for(U)), dim3(256), 0, U, rep_read, state, p_off, p_edges, inv, in.vleft, in.vright, indeg, outdeg, in_e, out_e, ext, xlen);
        W2_TRY(exclusive_scan_u32_to_u64(c, xlen, xoff, U));
        uint64_t x_total = 0, e_total = 0;
        W2_HIP(hipMemcpy(&x_total, xoff + U, 8, hipMemcpyDeviceToHost));
        if (na) W2_HIP(hipMemcpy(&e_total, p_off + na, 8, hipMemcpyDeviceToHost));
        uint64_t* off3 = nullptr; int32_t* edges3 = nullptr;
        W2_ALLOC(off3, uint64_t, na + U + 2); W2_ALLOC(edges3, int32_t, e_total + x_total + 1);
        if (na) W2_HIP(hipMemcpyAsync(off3, p_off, na * 8, hipMemcpyDeviceToDevice, st));
        if (e_total) W2_HIP(hipMemcpyAsync(edges3, p_edges, e_total * 4, hipMemcpyDeviceToDevice, st));
        LAUNCH(c, "k3_shift_off", k3_shift_off, dim3(grid_for(U + 1)), dim3(256), 0, U + 1, xoff, e_total, off3 + na);
        LAUNCH(c, "k3_extend_fill", k3_extend_fill, dim3(grid_for(U)), dim3(256), 0, U, rep_read, state, p_off, p_edges, inv, ext, xoff, edges3 + e_total);
        W2_HIP(hipStreamSynchronize(st));
        for (void* q : {(void*)indeg, (void*)outdeg, (void*)in_e, (void*)out_e, (void*)xlen, (void*)ext, (void*)xoff, (void*)state, (void*)place_of_read, (void*)rep_read, (void*)d_pcnt})
            c.release(q);
        p_off = off3; p_edges = edges3; na += U;
        W2_TRY(compute_places(na, p_off, p_edges));
    }
    if (P.flags & W2RAP_STEP3_PLACES_ONLY) {
        uint32_t* rl = nullptr; uint64_t* ro = nullptr;
        W2_ALLOC(rl, uint32_t, U + 1); W2_ALLOC(ro, uint64_t, U + 2);
        if (U) LAUNCH(c, "k3_rep_len", k3_rep_len, dim3(grid_for(U)), dim3(256), 0, U, rep_read, p_off, rl);
        W2_TRY(exclusive_scan_u32_to_u64(c, rl, ro, U));
        uint64_t tot = 0;
        W2_HIP(hipMemcpy(&tot, ro + U, 8, hipMemcpyDeviceToHost));
        int32_t* re = nullptr; W2_ALLOC(re, int32_t, tot + 1);
        if (U) LAUNCH(c, "k3_rep_copy", k3_rep_copy, dim3(grid_for(U)), dim3(256), 0, U, rep_read, p_off, p_edges, ro, re);
        W2_TRY(dl(c, &out.place_path_off, ro, U + 1)); W2_TRY(dl(c, &out.place_path_edges, re, tot));
        unsigned long long h_pc[192];
        W2_HIP(hipMemcpyAsync(h_pc, d_pcnt, sizeof h_pc, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        for (unsigned k = 0; k < 64; ++k) { out.n_reads_pathed += h_pc[3 * k]; out.n_reads_multipathed += h_pc[3 * k + 1]; }
        out.K2 = (int32_t)K2; out.n_place_paths = U; out.n_unique_places = U_printed; out.n_places = np_printed; out.ms_places = t_places.stop();
        return 0;
    }
    // ---------------------------------------------------------------- all
    uint32_t *plen, *nbases, *nwordsU, *nkm; int32_t *ltrunc, *rtrunc;
    W2_ALLOC(plen, uint32_t, U + 1); W2_ALLOC(nbases, uint32_t, U + 1); W2_ALLOC(nwordsU, uint32_t, U + 1); W2_ALLOC(nkm, uint32_t, U + 1);
    W2_ALLOC(ltrunc, int32_t, U + 1); W2_ALLOC(rtrunc, int32_t, U + 1);
    uint64_t *voff, *woff, *koff;
    W2_ALLOC(voff, uint64_t, U + 2); W2_ALLOC(woff, uint64_t, U + 2); W2_ALLOC(koff, uint64_t, U + 2);
    if (U) LAUNCH(c, "k3_place_layout", k3_place_layout, dim3(grid_for(U)), dim3(256), 0, U, K, K2, rep_read, state, p_off, p_edges, inv, olen, plen, nbases, nwordsU, nkm,
                             ltrunc, rtrunc, d_flags);
    W2_TRY(exclusive_scan_u32_to_u64(c, plen, voff, U));
    W2_TRY(exclusive_scan_u32_to_u64(c, nwordsU, woff, U));
    W2_TRY(exclusive_scan_u32_to_u64(c, nkm, koff, U));
    uint64_t nvec = 0, nwords_all = 0, N2 = 0;
    W2_HIP(hipMemcpy(&nvec, voff + U, 8, hipMemcpyDeviceToHost));
    W2_HIP(hipMemcpy(&nwords_all, woff + U, 8, hipMemcpyDeviceToHost));
    W2_HIP(hipMemcpy(&N2, koff + U, 8, hipMemcpyDeviceToHost));
    W2_TRY(check());
    if (N2 >= (1ull << 32) - 2 || nwords_all * 32 >= (1ull << 40)) { c.err = "more than 2^32 K2-mer occurrences on one GPU (32-bit occurrence ids)"; return W2RAP_E_LIMIT; }
    int32_t* pvec = nullptr; int64_t* pstart = nullptr; uint64_t* all = nullptr;
    W2_ALLOC(pvec, int32_t, nvec + 1); W2_ALLOC(pstart, int64_t, nvec + 1); W2_ALLOC(all, uint64_t, nwords_all + 4);
    W2_HIP(hipMemsetAsync(all + nwords_all, 0, 32, st));
    if (U) LAUNCH(c, "k3_place_vec", k3_place_vec, dim3(grid_for(U)), dim3(256), 0, U, K, rep_read, state, p_off, p_edges, inv, olen, voff, ltrunc, pvec, pstart);
    if (nwords_all) {
        uint32_t* wblk = nullptr;
        W2_TRY(block_index(c, (const uint64_t*)woff, U, nwords_all, &wblk));
        LAUNCH(c, "k3_all_fill", k3_all_fill, dim3(grid_for(nwords_all)), dim3(256), 0, nwords_all, U, woff, nbases, voff, pvec, pstart, obits, obase0, olen, all, (const uint32_t*)wblk);
        c.release(wblk);
    }
    uint32_t* kblk = nullptr;                                      // where every block of 256 occurrences starts among the places
    W2_TRY(block_index(c, (const uint64_t*)koff, U, N2 ? N2 : 1, &kblk));
    const uint8_t* allb = reinterpret_cast<const uint8_t*>(all);
    out.ms_places = t_places.stop();
    // ---------------------------------------------------------------- dictionary
    Timer t_dict(st);
    uint64_t* key = nullptr; uint32_t* val = nullptr; uint16_t* meta = nullptr;
    W2_ALLOC(key, uint64_t, N2 + 1); W2_ALLOC(val, uint32_t, N2 + 1); W2_ALLOC(meta, uint16_t, N2 + 2);
    uint64_t* gpos = nullptr; W2_ALLOC(gpos, uint64_t, N2 + 1);
    // the replay of a given edge order looks its edges up in the hash-ORDERED list of the distinct K2-mers: the sorted form
    bool sorted_dict = P.edge_order_hint != nullptr || getenv("W2RAP_STEP3_SORT_DICT") != nullptr;
    uint32_t *grp_rep, *ctx_by_x; uint64_t* pid;
    W2_ALLOC(grp_rep, uint32_t, N2 + 1); W2_ALLOC(ctx_by_x, uint32_t, N2 + 1); W2_ALLOC(pid, uint64_t, N2 + 2);
    // the K2-mers strictly inside a one-edge place whose edge no longer place shares stay out of the dictionary (k3_lone_places)
    const bool lone_on = !sorted_dict && N2 && (in.unique_kmers || (P.flags & W2RAP_STEP3_UNIQUE_KMERS)) && !getenv("W2RAP_STEP3_NO_LONE");
    uint8_t* lone = nullptr; unsigned long long n_pairs = N2;
    if (lone_on) {
        uint8_t* shared_edge = nullptr; unsigned long long* d_np = nullptr;
        W2_ALLOC(shared_edge, uint8_t, NO + 1); W2_ALLOC(lone, uint8_t, U + 1); W2_ALLOC(d_np, unsigned long long, 1);
        W2_HIP(hipMemsetAsync(shared_edge, 0, NO + 1, st)); W2_HIP(hipMemsetAsync(d_np, 0, 8, st));
        LAUNCH(c, "k3_mid_edges", k3_mid_edges, dim3(grid_for(std::max<uint64_t>(U, NO))), dim3(256), 0, U, NO, voff, pvec, inv, shared_edge);
        LAUNCH(c, "k3_lone_places", k3_lone_places, dim3(grid_for(U)), dim3(256), 0, U, voff, pvec, shared_edge, lone);
        LAUNCH(c, "k3_kmer_keys", k3_kmer_keys, dim3((unsigned)((N2 + 256 * KK_PER - 1) / (256 * KK_PER))), dim3(256), 0, N2, U, q, koff, woff, nbases, (const uint64_t*)all, key, val, meta, gpos, grp_rep, ctx_by_x,
               (const uint8_t*)lone, d_np, (const uint32_t*)kblk);
        W2_HIP(hipMemcpyAsync(&n_pairs, d_np, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        c.release(shared_edge); c.release(d_np);
        if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] step 3 dictionary: %llu of %llu occurrences lie strictly inside an edge no longer place shares\n",
                                           (unsigned long long)(N2 - n_pairs), (unsigned long long)N2);
    }
    else if (N2) LAUNCH(c, "k3_kmer_keys", k3_kmer_keys, dim3((unsigned)((N2 + 256 * KK_PER - 1) / (256 * KK_PER))), dim3(256), 0, N2, U, q, koff, woff, nbases, (const uint64_t*)all, key, sorted_dict ? val : (uint32_t*)nullptr, meta, gpos,
                        grp_rep, ctx_by_x, (const uint8_t*)nullptr, (unsigned long long*)nullptr, (const uint32_t*)kblk);
    uint32_t *ghead = nullptr, *gcoll = nullptr, *gover = nullptr, *hidx = nullptr;
    unsigned long long ncoll = 0;
    if (!sorted_dict && N2) {
        bool overflow = false;
        if (n_pairs) W2_TRY(dict_by_partition(c, n_pairs, q, key, lone_on ? val : (const uint32_t*)nullptr, allb, gpos, meta, grp_rep, ctx_by_x, overflow, N2));
        if (overflow) {
            if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] step 3 dictionary: a hash partition overflowed, sorting instead\n");
            sorted_dict = true;
            // (the sorted form wants the key of EVERY occurrence in position order)
            if (lone_on) LAUNCH(c, "k3_kmer_keys", k3_kmer_keys, dim3((unsigned)((N2 + 256 * KK_PER - 1) / (256 * KK_PER))), dim3(256), 0, N2, U, q, koff, woff, nbases, (const uint64_t*)all, key, val, meta, gpos, grp_rep, ctx_by_x,
                                (const uint8_t*)nullptr, (unsigned long long*)nullptr, (const uint32_t*)kblk);
            else LAUNCH(c, "k3_iota", k3_iota, dim3(grid_for(N2)), dim3(256), 0, N2, val);
        }
    }
    if (lone) c.release(lone);
    if (sorted_dict) {
        W2_TRY(sort_pairs_u64(c, key, val, N2, 0, (int)q.sbits));
        W2_ALLOC(ghead, uint32_t, N2 + 1); W2_ALLOC(gcoll, uint32_t, N2 + 1); W2_ALLOC(hidx, uint32_t, N2 + 1);
        if (N2) LAUNCH(c, "k3_group", k3_group, dim3(grid_for(N2)), dim3(256), 0, N2, U, q, key, val, meta, gpos, allb, ghead, gcoll, d_cnt + 103);
        W2_HIP(hipMemcpyAsync(&ncoll, d_cnt + 103, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        if (ncoll) {
            W2_ALLOC(gover, uint32_t, N2 + 1);
            W2_HIP(hipMemsetAsync(gover, 0xFF, (N2 + 1) * 4, st));
            LAUNCH(c, "k3_group_fix", k3_group_fix, dim3(grid_for(N2)), dim3(256), 0, N2, U, q, key, val, meta, gpos, allb, gcoll, ghead, gover);
        }
        W2_TRY(inclusive_max_scan_u32(c, ghead, hidx, N2));
        if (getenv("W2RAP_TRACE") && N2 && N2 < (1u << 22)) {
            std::vector<uint32_t> hh(N2), hx(N2); std::vector<uint64_t> hk(N2);
            W2_HIP(hipMemcpy(hh.data(), ghead, N2 * 4, hipMemcpyDeviceToHost)); W2_HIP(hipMemcpy(hx.data(), hidx, N2 * 4, hipMemcpyDeviceToHost));
            W2_HIP(hipMemcpy(hk.data(), key, N2 * 8, hipMemcpyDeviceToHost));
            uint64_t nheads = 0, badscan = 0, unsorted = 0; uint32_t m = 0;
            for (uint64_t j = 0; j < N2; ++j) { if (hh[j]) ++nheads; m = std::max(m, hh[j]); if (hx[j] != m) ++badscan; if (j && (hk[j] & sort_mask(q.sbits)) < (hk[j - 1] & sort_mask(q.sbits))) ++unsorted; }
            fprintf(stderr, "[w2rap]   group heads %llu, max-scan mismatches %llu, sort-key inversions %llu\n", (unsigned long long)nheads, (unsigned long long)badscan, (unsigned long long)unsorted);
        }
        if (N2) {
            LAUNCH(c, "k3_rep_init", k3_rep_init, dim3(grid_for(N2)), dim3(256), 0, N2, meta, grp_rep, ctx_by_x);
            LAUNCH(c, "k3_scatter_rep", k3_scatter_rep, dim3(grid_for(N2)), dim3(256), 0, N2, ghead, hidx, (const uint32_t*)gover, val, meta, grp_rep, ctx_by_x);
        }
    }
    W2_TRY(exclusive_scan_is_self(c, grp_rep, pid, N2));          // (the flags "is its own representative" scanned without being written)
    uint64_t D = 0;
    W2_HIP(hipMemcpy(&D, pid + N2, 8, hipMemcpyDeviceToHost));
    if (2 * D >= (1ull << 32) - 2) { c.err = "more than 2^31 distinct K2-mers on one GPU (32-bit node ids)"; return W2RAP_E_LIMIT; }
    uint32_t *id_of, *krep, *dctx; uint64_t* dhash = nullptr; uint32_t* did = nullptr;
    W2_ALLOC(id_of, uint32_t, N2 + 1); W2_ALLOC(krep, uint32_t, D + 1); W2_ALLOC(dctx, uint32_t, D + 1);
    if (N2) LAUNCH(c, "k3_finish_ids", k3_finish_ids, dim3(grid_for(N2)), dim3(256), 0, N2, grp_rep, pid, ctx_by_x, id_of, krep, dctx);
    if (P.edge_order_hint && N2) {                   // replay: the sorted (hash, id) list of the distinct K2-mers
        uint32_t* hf = hidx; uint64_t* hex = pid;    // (reused)
        W2_ALLOC(dhash, uint64_t, D + 1); W2_ALLOC(did, uint32_t, D + 1);
        LAUNCH(c, "k3_nonzero", k3_nonzero, dim3(grid_for(N2)), dim3(256), 0, N2, ghead, hf);
        W2_TRY(exclusive_scan_u32_to_u64(c, hf, hex, N2));
        LAUNCH(c, "k3_head_list", k3_head_list, dim3(grid_for(N2)), dim3(256), 0, N2, ghead, hex, key, val, id_of, dhash, did);
    }
    W2_HIP(hipStreamSynchronize(st));
    for (void* p : {(void*)key, (void*)val, (void*)ghead, (void*)gcoll, (void*)hidx, (void*)grp_rep, (void*)ctx_by_x, (void*)pid, (void*)gover})
        if (p) c.release(p);
    if (getenv("W2RAP_TRACE")) {
        fprintf(stderr, "[w2rap] step 3 dictionary: %llu occurrences, %llu distinct, %llu neighbours with one sort key but different content\n",
                (unsigned long long)N2, (unsigned long long)D, ncoll);
        if (N2 && N2 < (1u << 22)) {                     // small inputs: consistency of the id arrays
            std::vector<uint32_t> hid(N2), hrep(D), hctx(D);
            W2_HIP(hipMemcpy(hid.data(), id_of, N2 * 4, hipMemcpyDeviceToHost)); W2_HIP(hipMemcpy(hrep.data(), krep, D * 4, hipMemcpyDeviceToHost));
            W2_HIP(hipMemcpy(hctx.data(), dctx, D * 4, hipMemcpyDeviceToHost));
            uint64_t bad_id = 0, bad_rep = 0, zero_ctx = 0;
            for (uint64_t x = 0; x < N2; ++x) if (hid[x] >= D) ++bad_id;
            for (uint64_t i = 0; i < D; ++i) { if (hrep[i] >= N2 || hid[hrep[i]] != i) ++bad_rep; if (!(hctx[i] & 0xFF)) ++zero_ctx; }
            fprintf(stderr, "[w2rap]   ids out of range %llu, representatives inconsistent %llu, empty contexts %llu\n", (unsigned long long)bad_id,
                    (unsigned long long)bad_rep, (unsigned long long)zero_ctx);
        }
    }
    out.ms_dict = t_dict.stop();
    // ---------------------------------------------------------------- unipaths
    Timer t_graph(st);
    const uint64_t N = 2 * D;
    const KSrc S{allb, gpos, krep, meta, q};
    uint32_t *nbr, *nxt0, *nxt, *rnk; unsigned long long* rankw; uint8_t *cyc, *mid, *is_head;
    W2_ALLOC(nbr, uint32_t, N + 2); W2_ALLOC(nxt0, uint32_t, N + 2); W2_ALLOC(nxt, uint32_t, N + 2); W2_ALLOC(rnk, uint32_t, N + 2);
    W2_ALLOC(rankw, unsigned long long, N + 2); W2_ALLOC(cyc, uint8_t, N + 2); W2_ALLOC(mid, uint8_t, N + 2); W2_ALLOC(is_head, uint8_t, N + 2);
    W2_HIP(hipMemsetAsync(nbr, 0xFF, (N + 2) * 4, st));
    W2_HIP(hipMemsetAsync(d_flags, 0, 32, st));
    if (D) {
        if (N2 > 1) LAUNCH(c, "k3_nbr", k3_nbr, dim3(grid_for(N2)), dim3(256), 0, N2, U, koff, id_of, meta, nbr, (const uint32_t*)kblk);
        LAUNCH(c, "k3_links", k3_links, dim3(grid_for(D)), dim3(256), 0, D, dctx, nbr, nxt0);
        W2_TRY(run_ranking(c, N, nxt0, nxt, rnk, rankw, cyc, mid, d_flags, nullptr, nullptr, false));
        W2_TRY(check());
        if (h_flags[2]) {                            // smooth circles
            uint32_t *nx, *mn, *nx2, *mn2;
            W2_ALLOC(nx, uint32_t, N); W2_ALLOC(mn, uint32_t, N); W2_ALLOC(nx2, uint32_t, N); W2_ALLOC(mn2, uint32_t, N);
            LAUNCH(c, "k3_minjump_init", k3_minjump_init, dim3(grid_for(N)), dim3(256), 0, N, nxt0, cyc, nx, mn);
            for (int round = 0; round < 33; ++round) {
                LAUNCH(c, "k3_minjump", k3_minjump, dim3(grid_for(N)), dim3(256), 0, N, S, nx, mn, nx2, mn2);
                std::swap(nx, nx2); std::swap(mn, mn2);
            }
            LAUNCH(c, "k3_cycle_cut", k3_cycle_cut, dim3(grid_for(D)), dim3(256), 0, D, cyc, mn, nxt0);
            W2_HIP(hipStreamSynchronize(st));
            c.release(nx); c.release(mn); c.release(nx2); c.release(mn2);
            W2_HIP(hipMemsetAsync(d_flags, 0, 32, st));
            W2_TRY(run_ranking(c, N, nxt0, nxt, rnk, rankw, cyc, mid, d_flags, nullptr, nullptr, false));
            W2_TRY(check());
            if (h_flags[2]) { c.err = "failed to close circle (BigKPather.cc:141)"; return W2RAP_E_GRAPH; }
        }
        W2_HIP(hipMemsetAsync(mid, 0, N, st));
        LAUNCH(c, "k3_mid", k3_mid, dim3(grid_for(D)), dim3(256), 0, D, S, nxt, rnk, mid);
    }
    const uint64_t head_cap = D ? c.rank_ends + 1 : 1;
    uint32_t* head_v = nullptr;
    W2_ALLOC(head_v, uint32_t, head_cap + 1);
    if (N) {
        uint32_t* cand = nullptr;
        W2_ALLOC(cand, uint32_t, head_cap + 1);
        LAUNCH(c, "k3_head_cands", k3_head_cands, dim3((unsigned)((N + 256 * HC_PER - 1) / (256 * HC_PER))), dim3(256), 0, N, nxt0, is_head, cand, d_cnt + 101, head_cap);
        LAUNCH(c, "k3_heads", k3_heads, dim3(grid_for(head_cap)), dim3(256), 0, (const unsigned long long*)(d_cnt + 101), head_cap, (const uint32_t*)cand, S, dctx, nxt, rnk, mid, is_head,
               head_v, d_cnt + 102);
        c.release(cand);
    }
    unsigned long long E = 0;
    W2_HIP(hipMemcpyAsync(&E, d_cnt + 102, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    if (E > head_cap) { c.err = "more canonical heads than chain ends"; return W2RAP_E_GRAPH; }
    uint32_t *perm, *head_edge, *edge_head, *edge_nk; uint64_t* wtmp;
    W2_ALLOC(perm, uint32_t, E + 1); W2_ALLOC(head_edge, uint32_t, N + 2); W2_ALLOC(edge_head, uint32_t, E + 1); W2_ALLOC(edge_nk, uint32_t, E + 1); W2_ALLOC(wtmp, uint64_t, E + 1);
    W2_HIP(hipMemsetAsync(head_edge, 0xFF, (N + 2) * 4, st));
    const w2rap_edge_hint* hint = P.edge_order_hint;
    if (hint) {
        if (hint->n_edges != E) { c.err = "edge_order_hint has " + std::to_string(hint->n_edges) + " edges, the graph has " + std::to_string(E); return W2RAP_E_HINT; }
        for (uint64_t e = 0; e < E; ++e) {
            if (hint->len[e] < K2) { c.err = "edge_order_hint: edge shorter than K2"; return W2RAP_E_HINT; }
            if (hint->byte_off[e + 1] < hint->byte_off[e] || hint->byte_off[e + 1] - hint->byte_off[e] != ((uint64_t)hint->len[e] + 3) / 4) {
                c.err = "edge_order_hint: byte_off does not match len"; return W2RAP_E_HINT;
            }
        }
        uint8_t* hbits = nullptr; uint64_t *hbyte = nullptr, *hbase0 = nullptr; uint32_t* hlen = nullptr;
        W2_TRY(up_pooled(c, &hbits, hint->packed, E ? hint->byte_off[E] : 0, 32));
        W2_TRY(up_pooled(c, &hbyte, hint->byte_off, E + 1));
        W2_TRY(up_pooled(c, &hlen, hint->len, E));
        W2_ALLOC(hbase0, uint64_t, E + 1);
        LAUNCH(c, "k3_mul4", k3_mul4, dim3(grid_for(E + 1)), dim3(256), 0, E + 1, hbyte, hbase0);
        if (E) LAUNCH(c, "k3_edge_from_hint", k3_edge_from_hint, dim3(grid_for(E)), dim3(256), 0, E, D, S, hbits, hbase0, hlen, dhash, did, is_head, rnk, head_edge, edge_head, edge_nk, d_flags);
        W2_TRY(check());
        for (void* p : {(void*)hbits, (void*)hbyte, (void*)hlen, (void*)hbase0}) c.release(p);
    } else if (E) {
        // canonical order: the unipaths by their sequences = by their first K2-mers (distinct), NW words, least significant first
        // ONE sort by the first word, then the (short) runs of equal first words ordered by full comparison; NW sorts, least significant
        // word first, only if a run is too long for that
        bool by_words = test_hook("W2RAP_TEST_STEP3_FULL_SORTS");
        unsigned max_run = 64;
        if (test_hook("W2RAP_TEST_TIE_RUN")) max_run = (unsigned)atoi(getenv("W2RAP_TEST_TIE_RUN"));
        LAUNCH(c, "k3_iota", k3_iota, dim3(grid_for(E)), dim3(256), 0, E, perm);
        if (!by_words) {
            LAUNCH(c, "k3_head_word", k3_head_word, dim3(grid_for(E)), dim3(256), 0, E, S, head_v, perm, 0u, wtmp);
            W2_TRY(sort_pairs_u64(c, wtmp, perm, E, 0, 64));
            W2_HIP(hipMemsetAsync(d_flags + 6, 0, 4, st));
            LAUNCH(c, "k3_edge_tie_sort", k3_edge_tie_sort, dim3(grid_for(E)), dim3(256), 0, E, S, head_v, wtmp, perm, max_run, d_flags);
            uint32_t long_run = 0;
            W2_HIP(hipMemcpyAsync(&long_run, d_flags + 6, 4, hipMemcpyDeviceToHost, st));
            W2_HIP(hipStreamSynchronize(st));
            if (long_run) { by_words = true; LAUNCH(c, "k3_iota", k3_iota, dim3(grid_for(E)), dim3(256), 0, E, perm); }
        }
        if (by_words)
            W2_TRY(sort_by_words(c, perm, E, q.NW, wtmp, [&](unsigned j, uint64_t* tmp) -> int {
                LAUNCH(c, "k3_head_word", k3_head_word, dim3(grid_for(E)), dim3(256), 0, E, S, head_v, perm, j, tmp);
                return 0; }));
        LAUNCH(c, "k3_edge_from_sorted", k3_edge_from_sorted, dim3(grid_for(E)), dim3(256), 0, E, perm, head_v, rnk, head_edge, edge_head, edge_nk);
    }
    // ---- edge sequences, K2-mer placements
    uint32_t* elen = nullptr; uint64_t* edge_off = nullptr;
    W2_ALLOC(elen, uint32_t, E + 1); W2_ALLOC(edge_off, uint64_t, E + 2);
    if (E) LAUNCH(c, "k3_edge_len", k3_edge_len, dim3(grid_for(E)), dim3(256), 0, E, K2, edge_nk, elen);
    W2_TRY(exclusive_scan_u32_to_u64(c, elen, edge_off, E));
    uint64_t edge_bases = 0;
    W2_HIP(hipMemcpy(&edge_bases, edge_off + E, 8, hipMemcpyDeviceToHost));
    uint8_t* codes = nullptr; uint32_t *k_edge, *k_off;
    W2_ALLOC(codes, uint8_t, edge_bases + 64); W2_ALLOC(k_edge, uint32_t, D + 1); W2_ALLOC(k_off, uint32_t, D + 1);
    if (D) LAUNCH(c, "k3_assign", k3_assign, dim3(grid_for(D)), dim3(256), 0, D, S, nxt, rnk, head_edge, edge_off, k_edge, k_off, codes, d_flags);
    W2_TRY(check());
    // ---- objects, vertices, adjacency
    uint32_t* nobj = nullptr; uint64_t* ooff = nullptr;
    W2_ALLOC(nobj, uint32_t, E + 1); W2_ALLOC(ooff, uint64_t, E + 2);
    if (E) LAUNCH(c, "k3_edge_nobj", k3_edge_nobj, dim3(grid_for(E)), dim3(256), 0, E, edge_head, edge_nk, dctx, nobj);
    W2_TRY(exclusive_scan_u32_to_u64(c, nobj, ooff, E));
    uint64_t NO2 = 0;
    W2_HIP(hipMemcpy(&NO2, ooff + E, 8, hipMemcpyDeviceToHost));
    int32_t *fwdX, *revX, *inv2, *left, *right; uint32_t* obj_edge;
    W2_ALLOC(fwdX, int32_t, E + 1); W2_ALLOC(revX, int32_t, E + 1); W2_ALLOC(inv2, int32_t, NO2 + 1); W2_ALLOC(obj_edge, uint32_t, NO2 + 1);
    W2_ALLOC(left, int32_t, NO2 + 1); W2_ALLOC(right, int32_t, NO2 + 1);
    if (E) LAUNCH(c, "k3_edge_xlat", k3_edge_xlat, dim3(grid_for(E)), dim3(256), 0, E, nobj, ooff, fwdX, revX, obj_edge, inv2);
    uint64_t NV = 0;
    const uint64_t nends = 2 * NO2;
    if (nends) {
        uint64_t *ehash, *etmp, *eex; uint32_t *eperm, *eflag;
        W2_ALLOC(ehash, uint64_t, nends); W2_ALLOC(etmp, uint64_t, nends); W2_ALLOC(eex, uint64_t, nends + 1); W2_ALLOC(eperm, uint32_t, nends); W2_ALLOC(eflag, uint32_t, nends);
        LAUNCH(c, "k3_end_hash", k3_end_hash, dim3(grid_for(nends)), dim3(256), 0, NO2, K2, obj_edge, edge_off, edge_nk, codes, ehash);
        LAUNCH(c, "k3_iota", k3_iota, dim3(grid_for(nends)), dim3(256), 0, nends, eperm);
        const unsigned EW = (K2 - 1 + 31) / 32;
        uint64_t* ewords = nullptr;
        W2_ALLOC(ewords, uint64_t, (uint64_t)EW * nends);
        LAUNCH(c, "k3_end_words", k3_end_words, dim3(grid_for(nends)), dim3(256), 0, nends, K2, EW, obj_edge, edge_off, edge_nk, codes, ewords);
        // ONE sort by the hash, boundaries where it changes, equal hashes verified by content (k3_end_group); by (hash, sequence) -- LSD over
        // the sequence words, then the hash -- only if two contents share a hash.  Same vertex numbers either way: the groups are ordered by hash.
        bool by_words = test_hook("W2RAP_TEST_STEP3_FULL_SORTS");
        const bool pretend = test_hook("W2RAP_TEST_ENDS_COLLISION");
        for (;;) {
            if (by_words)
                W2_TRY(sort_by_words(c, eperm, nends, EW, etmp, [&](unsigned j, uint64_t* tmp) -> int {
                    LAUNCH(c, "k3_gather_u64", k3_gather_u64, dim3(grid_for(nends)), dim3(256), 0, nends, ewords + (uint64_t)j * nends, eperm, tmp);
                    return 0; }));
            LAUNCH(c, "k3_gather_u64", k3_gather_u64, dim3(grid_for(nends)), dim3(256), 0, nends, ehash, eperm, etmp);
            W2_TRY(sort_pairs_u64(c, etmp, eperm, nends, 0, 64));
            if (by_words) {
                // vertex boundaries: the hash or any sequence word differs from the predecessor's
                LAUNCH(c, "k3_end_differs", k3_end_differs, dim3(grid_for(nends)), dim3(256), 0, nends, etmp, eflag, true);
                for (unsigned j = 0; j < EW; ++j) {
                    LAUNCH(c, "k3_gather_u64", k3_gather_u64, dim3(grid_for(nends)), dim3(256), 0, nends, ewords + (uint64_t)j * nends, eperm, etmp);
                    LAUNCH(c, "k3_end_differs", k3_end_differs, dim3(grid_for(nends)), dim3(256), 0, nends, etmp, eflag, false);
                }
            } else {
                W2_HIP(hipMemsetAsync(d_flags + 7, 0, 4, st));
                LAUNCH(c, "k3_end_group", k3_end_group, dim3(grid_for(nends)), dim3(256), 0, nends, EW, etmp, eperm, ewords, eflag, d_flags, pretend);
            }
            W2_TRY(exclusive_scan_u32_to_u64(c, eflag, eex, nends));
            uint32_t shared_hash = 0;
            if (!by_words) W2_HIP(hipMemcpyAsync(&shared_hash, d_flags + 7, 4, hipMemcpyDeviceToHost, st));
            W2_HIP(hipMemcpy(&NV, eex + nends, 8, hipMemcpyDeviceToHost));
            W2_HIP(hipStreamSynchronize(st));
            if (shared_hash && !by_words) { by_words = true; LAUNCH(c, "k3_iota", k3_iota, dim3(grid_for(nends)), dim3(256), 0, nends, eperm); continue; }
            break;
        }
        c.release(ewords);
        NV += 1;
        LAUNCH(c, "k3_end_vertices", k3_end_vertices, dim3(grid_for(nends)), dim3(256), 0, nends, eperm, eflag, eex, left, right);
        W2_HIP(hipStreamSynchronize(st));
        for (void* p : {(void*)ehash, (void*)etmp, (void*)eex, (void*)eperm, (void*)eflag}) c.release(p);
    }
    uint64_t *from_off, *to_off; int32_t *from_v, *from_e, *to_v, *to_e;
    W2_ALLOC(from_off, uint64_t, NV + 2); W2_ALLOC(to_off, uint64_t, NV + 2);
    W2_ALLOC(from_v, int32_t, NO2 + 1); W2_ALLOC(from_e, int32_t, NO2 + 1); W2_ALLOC(to_v, int32_t, NO2 + 1); W2_ALLOC(to_e, int32_t, NO2 + 1);
    {
        uint64_t* akeys = nullptr; uint32_t *avals = nullptr, *deg = nullptr;
        W2_ALLOC(akeys, uint64_t, NO2 + 1); W2_ALLOC(avals, uint32_t, NO2 + 1); W2_ALLOC(deg, uint32_t, NV + 1);
        for (int dir = 0; dir < 2; ++dir) {          // AddEdge keeps from_[v] sorted by target with ties in insertion (= object id) order: stable sort by (v, w)
            W2_HIP(hipMemsetAsync(deg, 0, (NV + 1) * 4, st));
            if (NO2) {
                LAUNCH(c, "k3_adj_keys", k3_adj_keys, dim3(grid_for(NO2)), dim3(256), 0, NO2, dir ? right : left, dir ? left : right, akeys, avals, deg);
                W2_TRY(sort_pairs_u64(c, akeys, avals, NO2, 0, 64));
                LAUNCH(c, "k3_adj_out", k3_adj_out, dim3(grid_for(NO2)), dim3(256), 0, NO2, avals, dir ? left : right, dir ? to_v : from_v, dir ? to_e : from_e);
            }
            W2_TRY(exclusive_scan_u32_to_u64(c, deg, dir ? to_off : from_off, NV));
        }
        c.release(akeys); c.release(avals); c.release(deg);
    }
    out.ms_graph = t_graph.stop();
    // ---------------------------------------------------------------- places through the graph, read paths
    Timer t_paths(st);
    uint32_t* ostart = nullptr; int32_t *oobj, *starts, *stops; uint64_t* oex;
    W2_ALLOC(ostart, uint32_t, N2 + 1); W2_ALLOC(oobj, int32_t, N2 + 1); W2_ALLOC(oex, uint64_t, N2 + 2); W2_ALLOC(starts, int32_t, U + 1); W2_ALLOC(stops, int32_t, U + 1);
    if (N2) LAUNCH(c, "k3_occ", k3_occ, dim3(grid_for(N2)), dim3(256), 0, N2, U, koff, id_of, meta, k_edge, k_off, edge_nk, fwdX, revX, ostart, oobj, starts, stops, (const uint32_t*)kblk);
    W2_TRY(exclusive_scan_u32_to_u64(c, ostart, oex, N2));
    uint64_t nip = 0;
    W2_HIP(hipMemcpy(&nip, oex + N2, 8, hipMemcpyDeviceToHost));
    int32_t* ipath = nullptr; uint64_t* ioff = nullptr;
    W2_ALLOC(ipath, int32_t, nip + 1); W2_ALLOC(ioff, uint64_t, U + 2);
    if (N2) LAUNCH(c, "k3_place_paths", k3_place_paths, dim3(grid_for(N2)), dim3(256), 0, N2, U, koff, ostart, oex, oobj, ipath, ioff, (const uint32_t*)kblk);
    W2_HIP(hipMemcpyAsync(ioff + U, &nip, 8, hipMemcpyHostToDevice, st));
    uint32_t* rcnt = nullptr; uint64_t* o_off = nullptr; int32_t *o_offset = nullptr, *o_edges = nullptr;
    W2_ALLOC(rcnt, uint32_t, n + 1); W2_ALLOC(o_off, uint64_t, n + 2); W2_ALLOC(o_offset, int32_t, n + 1);
    if (n) LAUNCH(c, "k3_read_counts", k3_read_counts, dim3(grid_for(n)), dim3(256), 0, n, state, place_of_read, ioff, rcnt);
    W2_TRY(exclusive_scan_u32_to_u64(c, rcnt, o_off, n));
    uint64_t npath_ints = 0;
    W2_HIP(hipMemcpy(&npath_ints, o_off + n, 8, hipMemcpyDeviceToHost));
    W2_ALLOC(o_edges, int32_t, npath_ints + 1);
    if (n) LAUNCH(c, "k3_read_paths", k3_read_paths, dim3(grid_for(n)), dim3(256), 0, n, state, place_of_read, ioff, ipath, inv2, p_offset, starts, stops, ltrunc, rtrunc, o_off,
                         o_offset, o_edges);
    out.ms_paths = t_paths.stop();
    // ---------------------------------------------------------------- results
    uint32_t *d_olen = nullptr, *d_nby = nullptr; uint64_t* d_byoff = nullptr; uint8_t* d_packed = nullptr;
    W2_ALLOC(d_olen, uint32_t, NO2 + 1); W2_ALLOC(d_nby, uint32_t, NO2 + 1); W2_ALLOC(d_byoff, uint64_t, NO2 + 2);
    if (NO2) LAUNCH(c, "k3_obj_len", k3_obj_len, dim3(grid_for(NO2)), dim3(256), 0, NO2, K2, obj_edge, edge_nk, d_olen, d_nby);
    W2_TRY(exclusive_scan_u32_to_u64(c, d_nby, d_byoff, NO2));
    uint64_t total_bytes = 0;
    W2_HIP(hipMemcpy(&total_bytes, d_byoff + NO2, 8, hipMemcpyDeviceToHost));
    W2_ALLOC(d_packed, uint8_t, total_bytes + 1);
    if (total_bytes) {
        uint32_t* bblk = nullptr;
        W2_TRY(block_index(c, (const uint64_t*)d_byoff, NO2, total_bytes, &bblk));
        LAUNCH(c, "k3_pack_objs", k3_pack_objs, dim3(grid_for(total_bytes)), dim3(256), 0, total_bytes, NO2, K2, d_byoff, obj_edge, edge_nk, edge_off, codes, d_packed, (const uint32_t*)bblk);
        c.release(bblk);
    }
    unsigned long long h_cnt[112];
    W2_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
    out.K2 = (int32_t)K2; out.n_vertices = NV; out.n_edge_objs = NO2; out.n_paths = n;
    if (!(P.flags & W2RAP_STEP3_NO_FETCH)) {
        W2_TRY(dl(c, &out.inv, inv, NO));
        W2_TRY(dl(c, &out.edge_packed, d_packed, total_bytes)); W2_TRY(dl(c, &out.edge_byte_off, d_byoff, NO2 + 1)); W2_TRY(dl(c, &out.edge_len, d_olen, NO2));
        W2_TRY(dl(c, &out.vleft, left, NO2)); W2_TRY(dl(c, &out.vright, right, NO2));
        W2_TRY(dl(c, &out.from_off, from_off, NV + 1)); W2_TRY(dl(c, &out.from_v, from_v, NO2)); W2_TRY(dl(c, &out.from_e, from_e, NO2));
        W2_TRY(dl(c, &out.to_off, to_off, NV + 1)); W2_TRY(dl(c, &out.to_v, to_v, NO2)); W2_TRY(dl(c, &out.to_e, to_e, NO2));
        W2_TRY(dl(c, &out.inv2, inv2, NO2));
        W2_TRY(dl(c, &out.path_offset, o_offset, n)); W2_TRY(dl(c, &out.path_off, o_off, n + 1)); W2_TRY(dl(c, &out.path_edges, o_edges, npath_ints));
    }
    W2_HIP(hipStreamSynchronize(st));
    if (!NV && out.from_off) { out.from_off[0] = 0; out.to_off[0] = 0; }
    for (int i = 0; i < 100; ++i) out.frag_count[i] = h_cnt[i];
    {
        unsigned long long hp[192];
        W2_HIP(hipMemcpy(hp, d_pcnt, sizeof(hp), hipMemcpyDeviceToHost));
        out.n_reads_pathed = out.n_reads_multipathed = 0;
        for (int i = 0; i < 64; ++i) { out.n_reads_pathed += hp[3 * i]; out.n_reads_multipathed += hp[3 * i + 1]; }
    }
    out.n_places = np_printed; out.n_unique_places = U_printed; out.n_place_bases = 0;
    {   // sum of the place lengths (what the reference calls `all`)
        std::vector<uint32_t> hb(U);
        if (U) W2_HIP(hipMemcpy(hb.data(), nbases, U * 4, hipMemcpyDeviceToHost));
        for (uint32_t b : hb) out.n_place_bases += b;
    }
    out.n_kmer_instances = N2; out.n_kmers_distinct = D; out.n_unipaths = E;
    return 0;
}

}  // namespace
}  // namespace w2

using namespace w2;

static void save_profile(Ctx& c) {      // per-kernel times of this run -> w2rap_step3_profile
    (void)hipStreamSynchronize(c.stream);
    c.presolve();
    g_profile.clear();
    for (auto& s : c.prof_sums) { char line[256]; std::snprintf(line, sizeof line, "%s %.4f %llu\n", s.name.c_str(), s.ms, (unsigned long long)s.launches); g_profile += line; }
}

extern "C" {

int w2rap_step3_run(const w2rap_step3_in* in, const w2rap_step3_params* P, w2rap_step3_out* out, char* err, size_t errlen) {
    auto fail = [&](int code, const std::string& m) { if (err && errlen) std::snprintf(err, errlen, "%s", m.c_str()); return code; };
    if (!in || !P || !out) return fail(W2RAP_E_ARG, "null argument");
    std::memset(out, 0, sizeof(*out));
    if (in->K < 16 || in->K > 64) return fail(W2RAP_E_ARG, "small K must be in [16, 64] (the reference runs Step 2 at K = 60)");
    if (P->K2 & 1 || P->K2 <= (uint32_t)in->K || P->K2 > 32 * MAXW) return fail(W2RAP_E_ARG, "K2 must be even, larger than K and at most 640");
    if (P->extend_paths && in->n_edge_objs && (!in->vleft || !in->vright)) return fail(W2RAP_E_ARG, "extend_paths needs the vertices of the small-K graph (vleft, vright)");
    if (P->n_extra_paths && (!P->extra_path_off || (P->extra_path_off[P->n_extra_paths] && !P->extra_path_edges))) return fail(W2RAP_E_ARG, "null extra path array");
    if (P->n_extra_paths && P->extra_path_off[0] != 0) return fail(W2RAP_E_ARG, "extra_path_off must start at 0");
    for (uint64_t r = 0; r < P->n_extra_paths; ++r) if (P->extra_path_off[r + 1] < P->extra_path_off[r]) return fail(W2RAP_E_ARG, "extra_path_off is not ascending");
    if (in->n_edge_objs >= (1ull << 31) || in->n_paths >= (1ull << 32) - 2) return fail(W2RAP_E_LIMIT, "more than 2^31 edge objects or 2^32 reads");
    if ((in->n_edge_objs && (!in->edge_packed || !in->edge_byte_off || !in->edge_len)) || (in->n_paths && (!in->path_offset || !in->path_off)))
        return fail(W2RAP_E_ARG, "null input array");
    if (in->n_paths && in->path_off[0] != 0) return fail(W2RAP_E_ARG, "path_off must start at 0");
    if (in->n_paths && in->path_off[in->n_paths] && !in->path_edges) return fail(W2RAP_E_ARG, "null path_edges");
    if (in->n_edge_objs && in->edge_byte_off[0] != 0) return fail(W2RAP_E_ARG, "edge_byte_off must start at 0");
    for (uint64_t r = 0; r < in->n_paths; ++r) {
        if (in->path_off[r + 1] < in->path_off[r]) return fail(W2RAP_E_ARG, "path_off is not ascending");
    }
    const uint64_t npe = in->n_paths ? in->path_off[in->n_paths] : 0;
    for (uint64_t i = 0; i < npe; ++i) if (in->path_edges[i] < 0 || (uint64_t)in->path_edges[i] >= in->n_edge_objs) return fail(W2RAP_E_ARG, "a path names an edge object that does not exist");
    for (uint64_t o = 0; o < in->n_edge_objs; ++o) {
        if (in->edge_len[o] < (uint32_t)in->K) return fail(W2RAP_E_ARG, "an edge object shorter than K bases");
        if (in->edge_byte_off[o + 1] < in->edge_byte_off[o] || in->edge_byte_off[o + 1] - in->edge_byte_off[o] != ((uint64_t)in->edge_len[o] + 3) / 4)
            return fail(W2RAP_E_ARG, "edge_byte_off does not match edge_len");
    }
    char ebuf[512] = {0};
    w2rap_step2_ctx* h = w2rap_step2_acquire(P->device, ebuf, sizeof ebuf);
    if (!h) return fail(W2RAP_E_NO_DEVICE, ebuf);
    Ctx& c = h->c;
    auto body = [&]() -> int {
        uint8_t* obits = nullptr; uint64_t* obyte = nullptr; uint32_t* olen = nullptr; int32_t *p_offset = nullptr, *p_edges = nullptr; uint64_t* p_off = nullptr;
        const uint64_t NO = in->n_edge_objs, n = in->n_paths;
        W2_TRY(up_pooled(c, &obits, in->edge_packed, NO ? in->edge_byte_off[NO] : 0, 32));
        W2_TRY(up_pooled(c, &obyte, in->edge_byte_off, in->edge_byte_off ? NO + 1 : 0));
        W2_TRY(up_pooled(c, &olen, in->edge_len, NO));
        W2_TRY(up_pooled(c, &p_offset, in->path_offset, n));
        W2_TRY(up_pooled(c, &p_off, in->path_off, in->path_off ? n + 1 : 0));
        W2_TRY(up_pooled(c, &p_edges, in->path_edges, npe));
        if (!n) { const uint64_t z = 0; W2_HIP(hipMemcpyAsync(p_off, &z, 8, hipMemcpyHostToDevice, c.stream)); }
        if (!NO) { const uint64_t z = 0; W2_HIP(hipMemcpyAsync(obyte, &z, 8, hipMemcpyHostToDevice, c.stream)); W2_HIP(hipStreamSynchronize(c.stream)); }
        int32_t *vleft = nullptr, *vright = nullptr;
        if (P->extend_paths && NO) { W2_TRY(up_pooled(c, &vleft, in->vleft, NO)); W2_TRY(up_pooled(c, &vright, in->vright, NO)); }
        return step3(c, DevIn{(unsigned)in->K, NO, obits, obyte, olen, n, p_offset, p_off, p_edges, in->n_vertices, vleft, vright}, *P, *out);
    };
    int rc = body();
    std::string msg = c.err;
    save_profile(c);
    if (rc) w2rap_step2_destroy(h); else w2rap_step2_release(h);     // (a failed context is not cached)
    if (rc) { w2rap_step3_free(out); return fail(rc, msg); }
    return 0;
}

// Step 3 straight behind Step 2 in one process (the reference's default flow, w2rap-contigger.cc:338-371): the graph and the read
// paths stay in HBM; only the large-K result comes back to the host.  The context must have run path_reads; it is left intact.
int w2rap_step3_run_after_step2(w2rap_step2_ctx* h, const w2rap_step3_params* P, w2rap_step3_out* out, char* err, size_t errlen) {
    auto fail = [&](int code, const std::string& m) { if (err && errlen) std::snprintf(err, errlen, "%s", m.c_str()); return code; };
    if (!h || !P || !out) return fail(W2RAP_E_ARG, "null argument");
    std::memset(out, 0, sizeof(*out));
    if (P->K2 & 1 || P->K2 <= K || P->K2 > 32 * MAXW) return fail(W2RAP_E_ARG, "K2 must be even, larger than K and at most 640");
    if (P->n_extra_paths && (!P->extra_path_off || (P->extra_path_off[P->n_extra_paths] && !P->extra_path_edges))) return fail(W2RAP_E_ARG, "null extra path array");
    if (P->n_extra_paths && P->extra_path_off[0] != 0) return fail(W2RAP_E_ARG, "extra_path_off must start at 0");
    for (uint64_t r = 0; r < P->n_extra_paths; ++r) if (P->extra_path_off[r + 1] < P->extra_path_off[r]) return fail(W2RAP_E_ARG, "extra_path_off is not ascending");
    Ctx& c = h->c;
    if (!c.graphed || !c.pathed_done) return fail(W2RAP_E_STATE, "w2rap_step3_run_after_step2: the context has not run build_graph and path_reads");
    if (hipSetDevice(c.device) != hipSuccess) return fail(W2RAP_E_HIP, "hipSetDevice failed");
    c.prof_sums.clear();
    auto body = [&]() -> int {
        const uint64_t NO = c.NO;
        uint32_t *d_len = nullptr, *d_nb = nullptr; uint64_t* d_boff = nullptr; uint8_t* d_packed = nullptr;
        W2_ALLOC(d_len, uint32_t, NO + 1); W2_ALLOC(d_nb, uint32_t, NO + 1); W2_ALLOC(d_boff, uint64_t, NO + 2);
        if (NO) LAUNCH(c, "k3_obj_len", k3_obj_len, dim3(grid_for(NO)), dim3(256), 0, NO, K, c.d_obj_edge, c.d_edge_nk, d_len, d_nb);
        W2_TRY(exclusive_scan_u32_to_u64(c, d_nb, d_boff, NO));
        uint64_t total = 0;
        W2_HIP(hipMemcpy(&total, d_boff + NO, 8, hipMemcpyDeviceToHost));
        W2_ALLOC(d_packed, uint8_t, total + 64);
        W2_HIP(hipMemsetAsync(d_packed + total, 0, 64, c.stream));
        if (total) {
            uint32_t* bblk = nullptr;
            W2_TRY(block_index(c, (const uint64_t*)d_boff, NO, total, &bblk));
            LAUNCH(c, "k3_pack_objs", k3_pack_objs, dim3(grid_for(total)), dim3(256), 0, total, NO, K, d_boff, c.d_obj_edge, c.d_edge_nk, c.d_edge_off, c.d_edge_codes, d_packed, (const uint32_t*)bblk);
            c.release(bblk);
        }
        const int rc = step3(c, DevIn{K, NO, d_packed, d_boff, d_len, c.n, c.d_path_offset, c.d_path_off, c.d_path_edges, c.NV, c.d_left, c.d_right, true}, *P, *out);
        return rc;
    };
    // everything Step 3 allocates is tracked behind this mark and released (parked in the context's pool) afterwards
    const size_t mark = c.owned.size();
    int rc = body();
    (void)hipStreamSynchronize(c.stream);
    while (c.owned.size() > mark) { void* p = c.owned.back(); c.owned.pop_back(); c.park(p); }
    std::string msg = c.err;
    save_profile(c);
    if (rc) { w2rap_step3_free(out); return fail(rc, msg); }
    return 0;
}

void w2rap_step3_free(w2rap_step3_out* o) {
    if (!o) return;
    for (void* p : {(void*)o->inv, (void*)o->edge_packed, (void*)o->edge_byte_off, (void*)o->edge_len, (void*)o->vleft, (void*)o->vright, (void*)o->from_off,
                    (void*)o->from_v, (void*)o->from_e, (void*)o->to_off, (void*)o->to_v, (void*)o->to_e, (void*)o->inv2, (void*)o->path_offset, (void*)o->path_off,
                    (void*)o->path_edges, (void*)o->place_path_off, (void*)o->place_path_edges})
        std::free(p);
    std::memset(o, 0, sizeof(*o));
}

size_t w2rap_step3_profile(char* buf, size_t len) {
    if (buf && len) std::snprintf(buf, len, "%s", g_profile.c_str());
    return g_profile.size() + 1;
}

}  // extern "C"
