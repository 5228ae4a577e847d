// step2_graph.hip -- phases a7 (unipaths) and a8 (vertices + adjacency) on gfx950.
//
// Unipaths (EdgeBuilder / buildEdges, BuildReadQGraph.cc:99-339) are NOT walked one
// k-mer at a time: each solid k-mer is a node with two ports (up / down in its
// canonical orientation); a port is linked iff the reference's
// upstream/downstreamExtensionPossible (:192-214) holds, which is symmetric between
// the two k-mers of an adjacency.  Unipaths are then the chains of that graph and are
// resolved by pointer jumping (list ranking) over the 2S oriented nodes, so a
// 16M-k-mer edge costs 24 rounds, not 16M dependent probes.  Smooth circles
// (simpleCircle/canonicalizeCircle :126-180) are the nodes that never reach a chain
// end: their minimum k-mer is found by min-jumping, the circle is cut in front of it
// and the ranking is repeated.
//
// Orientation of every unipath follows bvec::getCanonicalForm (feudal/BaseVec.h:326 ->
// dna/CanonicalForm.h:34-46): odd length -> middle base A/C is FWD; even length ->
// lexicographic vs. its RC, which two distinct end k-mers always decide.
//
// Edge numbering: the reference's is arbitrary (spin-locked push_back under a parallel
// hash-set walk, :275-286); we number by the lexicographic order of the sequences
// (== order of their first 60-mers, which are unique) or replay a given order.
#include <algorithm>
#include "ctx.h"

namespace w2 {

// error bits reported through d_flags[1]
enum { GE_LOOKUP = 1, GE_OFFSET = 2, GE_HINT_MISS = 4, GE_HINT_DUP = 8, GE_ASSIGN = 16, GE_HINT_LEN = 32 };

template <class Id>
__device__ inline Kmer oriented(const uint64_t* shi, const uint64_t* slo, Id v) {
    Kmer k{shi[v >> 1], slo[v >> 1]};
    return (v & 1) ? kmer_rc(k) : k;
}

// ------------------------------------------------------------------------------ links
// IN PLACE: nbr[2i], nbr[2i+1] (the single surviving successor / predecessor, k_prune) become the chain links nxt0[2i], nxt0[2i+1]
// (a thread reads its own two words before it writes them; the neighbours' CONTEXTS come from sctx): no second 2S-word array
template <class Id>
__global__ void __launch_bounds__(256) k_links(uint64_t S, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo,
                                                const uint8_t* __restrict__ sctx, Id* nbr_nxt0) {
    constexpr Id NONE = NodeId<Id>::NONE, PAL = NodeId<Id>::PAL;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    Kmer k{shi[i], slo[i]};
    Id n0 = NONE, n1 = NONE;
    if (!kmer_is_pal(k)) {                                               // :105-106
        // nbr[] (from k_prune) = the single surviving successor / predecessor as an oriented node, or
        // PAL if that neighbour is a palindrome (:198,210), or NONE if there is not exactly one
        const Id s = nbr_nxt0[2 * i], p = nbr_nxt0[2 * i + 1];
        if (s < PAL) {                                                   // downstreamExtensionPossible :204-214
            unsigned cj = sctx[s >> 1]; if (s & 1) cj = brev8(cj);
            if (popc4(cj >> 4) == 1) n0 = s;
        }
        if (p < PAL) {                                                   // upstreamExtensionPossible :192-202
            unsigned cj = sctx[p >> 1]; if (p & 1) cj = brev8(cj);
            if (popc4(cj & 15) == 1) n1 = p ^ (Id)1;                     // walking backwards flips the traversal direction
        }
    }
    nbr_nxt0[2 * i] = n0; nbr_nxt0[2 * i + 1] = n1;
}

// ------------------------------------------------------------------------------ list ranking
// Every oriented node needs (chain end, distance to it).  Plain Wyllie pointer jumping moves all 2S
// nodes log2(longest chain) times through HBM; instead (Helman-JaJa style, with the splitters chosen by LOCALITY):
//   1. the solid k-mers of one minimizer bucket lie next to each other in K3's output (one emit CHUNK, in LDS-table order)
//      and consecutive k-mers of a unipath share their minimizer with probability ~45/47, so a chain mostly runs inside
//      a chunk.  Splitters = chain heads, chain ends and every node whose predecessor lies outside its chunk;
//   2. a block takes one chunk at a time (tiles of <= RT k-mers): the links are loaded into LDS (one coalesced read),
//      every node jumps BACKWARDS in LDS to the splitter that starts its segment (owner, steps from it) -- no HBM
//      traffic, <= log2(2 RT) rounds -- and the last node of a segment hands the segment's length and the next splitter to
//      its owner: w[owner] = (distance, next) (RankW, common.h);
//   3. pointer jumping IN PLACE on the packed words of the splitters only (~1/23 of the nodes; an 8-byte word is
//      read/written atomically and any version a lane sees is a consistent (next, distance) pair);
//   4. a node's (chain end, distance) = its owner's word minus its steps from the owner: rank_of() below, evaluated where it is
//      needed (same chunk: the sector is shared by its neighbours) -- Step 2 keeps no per-node end / rank arrays.
// Without a chunk list (or one that does not cover every k-mer) the tiles are simply RT consecutive k-mers: any cut of
// the node array into ranges is correct, locality only decides how many splitters there are.
// Nodes on a circle that lies inside one tile have no splitter and keep themselves as "end"; they are picked up by the
// circle detection, as are circles whose splitter ring never reaches a real chain end.
// (RT, RT_NODES, OWN_CIRCLE, rank_of, edge_of_end: common.h -- the sharded graph phase reads the same rank words)
constexpr unsigned RT_BUF = 3072;                  // splitter ids collected in LDS between two reservations of list space
// LDS word of a node during the backward jumping: bits 9:0 current target (local node), 29:10 steps to it, bit 31 = the
// target is the segment's splitter (final)
template <class Id>
__global__ void __launch_bounds__(256) k_rank_tiles(uint64_t S, uint64_t nchunks, const uint64_t* __restrict__ cstart,
                                                     const uint32_t* __restrict__ ccnt, const Id* __restrict__ nxt0,
                                                     unsigned long long* __restrict__ w, uint32_t* __restrict__ own,
                                                     Id* __restrict__ spl, unsigned long long* __restrict__ counters, uint64_t spl_cap) {
    constexpr Id NONE = NodeId<Id>::NONE;
    __shared__ __attribute__((aligned(16))) Id s_nx[RT_NODES];             // nxt0 of the tile's nodes
    __shared__ uint32_t s_w[RT_NODES];
    __shared__ Id s_buf[RT_BUF];
    __shared__ uint32_t s_nbuf, s_run;
    __shared__ unsigned long long s_base;
    const unsigned tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) { s_nbuf = 0; s_run = 0; }
    unsigned long long covered = 0;
    uint32_t ends = 0;                             // chain ends among this thread's nodes (as many heads as ends: bounds the edge list)
    auto flush = [&]() {                           // all threads; s_nbuf is stable on entry
        const uint32_t n = s_nbuf;
        if (tid == 0) s_base = n ? atomicAdd(&counters[0], (unsigned long long)n) : 0ull;
        __syncthreads();
        const unsigned long long gb = s_base;
        for (unsigned i = tid; i < n; i += 256) if (gb + i < spl_cap) spl[gb + i] = s_buf[i];
        __syncthreads();
        if (tid == 0) s_nbuf = 0;
        __syncthreads();
    };
    const uint64_t ntiles = cstart ? nchunks : (S + RT - 1) / RT;
    for (uint64_t ch = blockIdx.x; ch < ntiles; ch += gridDim.x) {
        uint64_t c_start; uint32_t c_cnt;
        if (cstart) { c_start = cstart[ch]; c_cnt = ccnt[ch]; if (c_start + c_cnt > S) c_cnt = 0; }
        else { c_start = ch * RT; c_cnt = (uint32_t)(S - c_start < RT ? S - c_start : RT); }
        covered += c_cnt;
        for (uint32_t sub = 0; sub < c_cnt; sub += RT) {               // an oversized chunk goes tile by tile
            const uint64_t base = 2 * (c_start + sub);
            const uint32_t nloc = 2 * (c_cnt - sub < RT ? c_cnt - sub : RT);
            __syncthreads();                                           // the previous tile's LDS words are no longer read
            {   // node x = 256 q + tid: the lanes of a wavefront take consecutive LDS words (with x = 4 tid + q every access of the tile's
                // words was a four-way bank conflict: 55 % of the kernel's LDS cycles)
                Id v4[4];
#pragma unroll
                for (unsigned q = 0; q < 4; ++q) v4[q] = 256 * q + tid < nloc ? nxt0[base + 256 * q + tid] : NONE;
#pragma unroll
                for (unsigned q = 0; q < 4; ++q) s_nx[256 * q + tid] = v4[q];
            }
            __syncthreads();
            bool listed[4];
            uint32_t mine = 0;
#pragma unroll
            for (unsigned q = 0; q < 4; ++q) {
                const unsigned x = 256 * q + tid;
                const Id nx = s_nx[x], px = s_nx[x ^ 1];                   // px = flip(predecessor)
                const bool inside = px != NONE && (uint64_t)px - base < (uint64_t)nloc;
                const bool split = x < nloc && (nx == NONE || !inside);
                ends += x < nloc && nx == NONE;
                listed[q] = split && nx != NONE;                           // chain ends never jump: they stay off the list
                mine += listed[q];
                // a splitter is its own owner; everybody else starts one step behind its predecessor
                s_w[x] = (split || x >= nloc) ? (0x80000000u | x) : ((1u << 10) | (uint32_t)(((uint64_t)px ^ 1ull) - base));
            }
            // backward jumping; asynchronous in place: any word a lane reads is a consistent (target, steps) pair
            __syncthreads();
            for (unsigned round = 0; round < 11; ++round) {
                bool open = false;
#pragma unroll
                for (unsigned q = 0; q < 4; ++q) {
                    const unsigned x = 256 * q + tid;
                    uint32_t wx = s_w[x];
                    if (!(wx >> 31)) {
                        const uint32_t wp = s_w[wx & 1023u];
                        uint32_t steps = ((wx >> 10) & 0xFFFFFu) + ((wp >> 10) & 0xFFFFFu);   // real segments: < 2 RT; a circle saturates
                        if (steps > 0xFFFFFu) steps = 0xFFFFFu;
                        wx = (wp & 0x800003FFu) | (steps << 10);
                        s_w[x] = wx;
                        open |= !(wx >> 31);
                    }
                }
                if (!__syncthreads_or(open)) break;
            }
            // room for this tile's splitters in the LDS list?
            for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
            if (lane == 0 && mine) atomicAdd(&s_run, mine);
            __syncthreads();
            if (s_nbuf + s_run > RT_BUF) flush();                          // uniform decision (both words stable here)
            const uint32_t lbase = s_nbuf;
            __syncthreads();
            if (tid == 0) { s_nbuf = lbase + s_run; s_run = 0; }
#pragma unroll
            for (unsigned q = 0; q < 4; ++q) {
                const unsigned x = 256 * q + tid;
                if (x < nloc) {
                    const uint32_t wx = s_w[x]; const Id nx = s_nx[x];
                    const bool done = wx >> 31;
                    const uint32_t o = wx & 1023u, j = (wx >> 10) & 0xFFFFFu;
                    own[base + x] = done ? ((j << 12) | (x + RT_NODES - o)) : OWN_CIRCLE;
                    if (done) {
                        if (nx == NONE) w[base + x] = RankW<Id>::pack(0, (Id)(base + x));              // chain end: next = itself, distance 0
                        else {
                            const uint64_t rel = (uint64_t)nx - base;
                            const bool next_split = rel >= (uint64_t)nloc || (s_w[rel] & 0xBFFFFFFFu) == (0x80000000u | (uint32_t)rel);
                            if (next_split) w[base + o] = RankW<Id>::pack(j + 1, nx);
                        }
                    }
                }
            }
            __syncthreads();                                               // s_run reset visible; s_w reads done
#pragma unroll
            for (unsigned q = 0; q < 4; ++q) {
                const unsigned long long m = __ballot(listed[q]);
                if (m) {
                    const int leader = __builtin_ctzll(m);
                    uint32_t wbase = 0;
                    if ((int)lane == leader) wbase = atomicAdd(&s_run, (uint32_t)__builtin_popcountll(m));
                    wbase = __shfl(wbase, leader);
                    if (listed[q]) s_buf[lbase + wbase + __builtin_popcountll(m & ((1ull << lane) - 1))] = (Id)(base + 256 * q + tid);
                }
            }
            __syncthreads();
            if (tid == 0) s_run = 0;
        }
    }
    __syncthreads();
    flush();
    if (tid == 0 && covered) atomicAdd(&counters[1], covered);            // (every thread counted the same chunks)
    for (int d = 32; d > 0; d >>= 1) ends += __shfl_down(ends, d);
    if (lane == 0 && ends) atomicAdd(&counters[2], (unsigned long long)ends);
}
constexpr int JUMPS_PER_LAUNCH = 16;
template <class Id>
__global__ void __launch_bounds__(256) k_split_jump(uint64_t n, const Id* __restrict__ spl, unsigned long long* __restrict__ w,
                                                     uint32_t* __restrict__ flags) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Id v = spl[i];
    unsigned long long wv = __hip_atomic_load(&w[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Id a = (Id)RankW<Id>::next(wv);
    if (a == v) return;                                            // chain end
    // several jumps per launch: every 8-byte word is a consistent (distance, next) pair whenever it is read, so the
    // jumping needs no barrier between rounds -- only the host's "nothing changed" test does
    bool changed = false, arrived = false;
    for (int round = 0; round < JUMPS_PER_LAUNCH; ++round) {
        const unsigned long long wa = __hip_atomic_load(&w[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const Id b = (Id)RankW<Id>::next(wa);
        if (b == a) { arrived = true; break; }                     // already points at its chain end
        wv = RankW<Id>::pack(RankW<Id>::dist(wv) + RankW<Id>::dist(wa), b);
        a = b;
        changed = true;
        // publish every fourth jump: lanes that come later in this launch then jump over what has been gathered so far
        if ((round & 3) == 3) __hip_atomic_store(&w[v], wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (changed) __hip_atomic_store(&w[v], wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // another launch is needed only if some lane has not arrived: its target is not a chain end (w[a] != a)
    if (!arrived) flags[0] = 1;
}
// ---- the splitter chains in two levels (round 6).  Pointer jumping over ALL listed splitters (6 % of the nodes) costs ~log2(splitters per chain)
// random words per splitter and launch -- nine for a 7000-k-mer unipath: 14 GB of random sectors at 50 M reads, the kernel runs at the rate such
// traffic reaches.  One splitter in sixteen (by a hash of its node number) is a SUPER-splitter: it walks to the next super-splitter (walk 1: every
// splitter's word is read once), the super-splitters alone jump (a sixteenth of the chain), every super-splitter walks its stretch again and gives
// the splitters on it their final words (walk 2).  Only valid (distance, next) pairs are ever written, so whatever the walks leave out -- the
// splitters in front of a chain's first super-splitter, chains without one, stretches cut by the step limit, circles -- is finished by the plain
// jumping behind them, which finds everything else arrived.
constexpr unsigned SUPER_STEPS = 64;
template <class Id> __device__ inline bool is_super(Id v) { return (((uint32_t)v * 0x9E3779B1u) >> 28) == 0u; }
// the super-splitters of the list, dense (one reservation per block of 4096 splitters -- one per wavefront is 380 k additions to one
// address, 4.7 ms): the walks run with every lane busy
constexpr unsigned SUPERS_PER_THREAD = 16;
template <class Id>
__global__ void __launch_bounds__(256) k_split_supers(uint64_t n, const Id* __restrict__ spl, Id* __restrict__ sup, unsigned long long* __restrict__ nsup, uint64_t cap) {
    __shared__ unsigned cnt; __shared__ unsigned long long base;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const uint64_t i0 = (uint64_t)blockIdx.x * (256 * SUPERS_PER_THREAD) + threadIdx.x;
    Id mine[SUPERS_PER_THREAD]; unsigned at[SUPERS_PER_THREAD];
#pragma unroll
    for (unsigned u = 0; u < SUPERS_PER_THREAD; ++u) {
        const uint64_t i = i0 + (uint64_t)u * 256;
        at[u] = ~0u;
        if (i < n) { mine[u] = spl[i]; if (is_super<Id>(mine[u])) at[u] = atomicAdd(&cnt, 1u); }
    }
    __syncthreads();
    if (threadIdx.x == 0) base = cnt ? atomicAdd(nsup, (unsigned long long)cnt) : 0;
    __syncthreads();
#pragma unroll
    for (unsigned u = 0; u < SUPERS_PER_THREAD; ++u)
        if (at[u] != ~0u && base + at[u] < cap) sup[base + at[u]] = mine[u];
}
template <class Id>
__global__ void __launch_bounds__(256) k_split_walk1(uint64_t n, const Id* __restrict__ spl, unsigned long long* __restrict__ w, unsigned long long* __restrict__ w0) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Id v = spl[i];                                           // (spl: the dense list of super-splitters)
    const unsigned long long wv = w[v];
    w0[i] = wv;                                                    // the word as the tiles left it: walk 2 starts from it
    Id t = (Id)RankW<Id>::next(wv);
    if (t == v) return;                                            // a chain end
    uint64_t d = RankW<Id>::dist(wv);
    unsigned steps = 0;
    while (!is_super<Id>(t) && steps < SUPER_STEPS) {              // (a non-super splitter's word is not written before walk 2)
        const unsigned long long wt = w[t];
        const Id y = (Id)RankW<Id>::next(wt);
        if (y == t) break;                                         // t is the chain end
        d += RankW<Id>::dist(wt); t = y; ++steps;
    }
    if (steps && t != v) w[v] = RankW<Id>::pack(d, t);             // (t == v: a circle with this one super-splitter on it -- left to the jumping)
}
template <class Id>
__global__ void __launch_bounds__(256) k_split_jump_super(uint64_t n, const Id* __restrict__ spl, unsigned long long* __restrict__ w, uint32_t* __restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Id v = spl[i];                                           // (spl: the dense list of super-splitters)
    unsigned long long wv = __hip_atomic_load(&w[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Id a = (Id)RankW<Id>::next(wv);
    if (a == v) return;
    bool changed = false, arrived = false;
    for (int round = 0; round < JUMPS_PER_LAUNCH; ++round) {
        const unsigned long long wa = __hip_atomic_load(&w[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const Id b = (Id)RankW<Id>::next(wa);
        if (b == a) { arrived = true; break; }
        wv = RankW<Id>::pack(RankW<Id>::dist(wv) + RankW<Id>::dist(wa), b);
        a = b;
        changed = true;
        if ((round & 3) == 3) __hip_atomic_store(&w[v], wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (changed) __hip_atomic_store(&w[v], wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!arrived) flags[0] = 1;
}
template <class Id>
__global__ void __launch_bounds__(256) k_split_walk2(uint64_t n, const Id* __restrict__ spl, unsigned long long* __restrict__ w, const unsigned long long* __restrict__ w0) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Id v = spl[i];                                           // (spl: the dense list of super-splitters)
    const unsigned long long wv = w[v];
    const Id e = (Id)RankW<Id>::next(wv);
    if (e == v) return;
    const uint64_t D = RankW<Id>::dist(wv);
    if (RankW<Id>::next(w[e]) != e) return;                        // this super-splitter has not arrived (a circle, a very long chain): the jumping finishes it
    if (sizeof(Id) == 8 && D >= (1ull << 31) - 1) return;          // (a saturated distance: reported as too long downstream)
    const unsigned long long o = w0[i];
    Id x = (Id)RankW<Id>::next(o);
    uint64_t acc = RankW<Id>::dist(o);
    unsigned steps = 0;
    while (!is_super<Id>(x) && x != e && steps < SUPER_STEPS && acc <= D) {
        const unsigned long long wx = w[x];                        // still the tiles' word: only this walk writes it
        const Id y = (Id)RankW<Id>::next(wx);
        if (y == x) break;                                         // (the chain end itself: x == e was tested above)
        w[x] = RankW<Id>::pack(D - acc, e);
        acc += RankW<Id>::dist(wx); x = y; ++steps;
    }
}
// every k-mer evaluates the finished ranks of its two nodes once: the circle test (a node whose "end" still has a successor lies on a
// circle) and the middle base of odd-length unipaths, as seen from each of the two heads (orientation by getCanonicalForm,
// feudal/BaseVec.h:326); Step 3 also wants the ranks as arrays (nxt, rnk; null for Step 2, which re-evaluates rank_of where needed)
template <class Id>
__global__ void __launch_bounds__(256) k_rank_finish(uint64_t S, const unsigned long long* __restrict__ w,
                                                      const uint32_t* __restrict__ own, const Id* __restrict__ nxt0,
                                                      const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo,
                                                      Id* __restrict__ nxt, uint32_t* __restrict__ rnk, uint8_t* __restrict__ cyc,
                                                      uint8_t* __restrict__ mid, uint32_t* __restrict__ flags) {
    constexpr Id NONE = NodeId<Id>::NONE;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    Id nx[2]; uint32_t rk[2];
#pragma unroll
    for (unsigned q = 0; q < 2; ++q) rank_of<Id>(own, w, (Id)(2 * i + q), nx[q], rk[q]);
    if (nxt) { nxt[2 * i] = nx[0]; nxt[2 * i + 1] = nx[1]; }
    if (rnk) *reinterpret_cast<uint2*>(&rnk[2 * i]) = make_uint2(rk[0], rk[1]);
    const bool c0 = nxt0[nx[0]] != NONE, c1 = nxt0[nx[1]] != NONE;
    *reinterpret_cast<uchar2*>(&cyc[2 * i]) = make_uchar2(c0, c1);
    if (c0 || c1) flags[2] = 1;
    // middle base
    if (!shi) return;
    const uint32_t r0 = rk[0], r1 = rk[1];
    const uint64_t n = (uint64_t)r0 + r1 + 1;
    if (n & 1) return;                               // even number of bases: decided by the end k-mers
    const uint64_t q = n / 2 + 29;                   // (n+59)/2
    const uint64_t x = q < n - 1 ? q : n - 1;
    if (r1 != x && r0 != x) return;
    const unsigned off = (unsigned)(q - x);
    const Kmer k{shi[i], slo[i]};
    if (r1 == x) mid[nx[1] ^ (Id)1] = (uint8_t)kmer_base(k, off);                    // traversed forward
    if (r0 == x) mid[nx[0] ^ (Id)1] = (uint8_t)kmer_base(kmer_rc(k), off);          // traversed reversed
}
// (kernels over the N = 2S oriented nodes run with one thread per K-MER, two nodes each: a grid has fewer than 2^32 threads, N may not)
template <class Id>
__global__ void __launch_bounds__(256) k_minjump_init(uint64_t S, const Id* __restrict__ nxt0, const uint8_t* __restrict__ cyc,
                                                       Id* __restrict__ nx, Id* __restrict__ mn) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
#pragma unroll
    for (unsigned q = 0; q < 2; ++q) {
        const uint64_t v = 2 * i + q;
        nx[v] = cyc[v] ? nxt0[v] : (Id)v;
        mn[v] = (Id)i;
    }
}
template <class Id>
__global__ void __launch_bounds__(256) k_minjump(uint64_t S, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo,
                                                  const Id* __restrict__ nx, const Id* __restrict__ mn,
                                                  Id* __restrict__ nx2, Id* __restrict__ mn2) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
#pragma unroll
    for (unsigned q = 0; q < 2; ++q) {
        const uint64_t v = 2 * i + q;
        const Id a = nx[v];
        const Id m0 = mn[v], m1 = mn[a];
        const Kmer k0{shi[m0], slo[m0]}, k1{shi[m1], slo[m1]};
        mn2[v] = kmer_lt(k1, k0) ? m1 : m0;
        nx2[v] = nx[a];
    }
}
// canonicalizeCircle :156-180: the circle starts at its minimum k-mer, traversed in canonical orientation
template <class Id>
__global__ void __launch_bounds__(256) k_cycle_cut(uint64_t S, const uint8_t* __restrict__ cyc, const Id* __restrict__ mn,
                                                    Id* __restrict__ nxt0) {
    constexpr Id NONE = NodeId<Id>::NONE;
    uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= S) return;
    if (cyc[2 * m] && mn[2 * m] == (Id)m) {
        Id u = nxt0[2 * m + 1];                  // reverse traversal leaves (m,1) towards flip(pred of (m,0))
        if (u != NONE) nxt0[u ^ (Id)1] = NONE;   // pred(m,0) -> (m,0) is cut
        nxt0[2 * m + 1] = NONE;
    }
}

// ------------------------------------------------------------------------------ orientation
// canonical heads -> unordered edge list with their first 60-mer as sort key
// A block takes HEADS_PER x 256 k-mers and gathers its canonical heads in LDS: ONE addition to the counter per block.  (One per head -- the
// compiler makes it one per wavefront -- is 356 k additions to one address on the planted workload's graph, 2.2 of this kernel's 3.0 ms there;
// a real genome has millions of unipaths.)
constexpr unsigned HEADS_PER = 16, HEADS_LDS = 512;
template <class Id>
__global__ void __launch_bounds__(256) k_heads(uint64_t S, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo,
                                                const Id* __restrict__ nxt0, const uint32_t* __restrict__ own,
                                                const unsigned long long* __restrict__ w, const uint8_t* __restrict__ mid,
                                                uint8_t* __restrict__ is_head, Id* __restrict__ head_v,
                                                uint64_t* __restrict__ key_hi, uint64_t* __restrict__ key_lo,
                                                unsigned long long* __restrict__ n_heads, uint64_t cap, uint32_t* __restrict__ flags, bool write) {
    constexpr Id NONE = NodeId<Id>::NONE;
    __shared__ uint32_t s_n; __shared__ unsigned long long s_base;
    __shared__ Id s_v[HEADS_LDS]; __shared__ uint64_t s_hi[HEADS_LDS], s_lo[HEADS_LDS];
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (unsigned it = 0; it < HEADS_PER; ++it) {
        const uint64_t i = ((uint64_t)blockIdx.x * HEADS_PER + it) * 256 + threadIdx.x;
        if (i >= S) break;
#pragma unroll
        for (unsigned q = 0; q < 2; ++q) {
            const uint64_t v = 2 * i + q;
            bool canon = false;
            Kmer F{0, 0};
            if (nxt0[v ^ 1] == NONE) {                       // the reverse of v is a chain end <=> v is a head
                F = oriented<Id>(shi, slo, (Id)v);
                Id end; uint32_t rk;
                rank_of<Id>(own, w, (Id)v, end, rk);
                uint64_t n = (uint64_t)rk + 1;
                if (n - 1 > 0xFFFFFFull) atomicOr(&flags[1], (uint32_t)GE_OFFSET);        // ReadPather.h:122 (24-bit offset)
                if (kmer_is_pal(F)) canon = !(v & 1);                                      // PALINDROME: one object (:247-249)
                else if (n & 1) {                                                          // even #bases
                    Kmer Fr = oriented<Id>(shi, slo, end ^ (Id)1);                         // first 60-mer of the RC sequence
                    canon = kmer_lt(F, Fr);
                } else canon = !(mid[v] & 2);                                              // odd #bases: middle base A/C
            }
            is_head[v] = canon;
            if (canon) {
                const uint32_t at = atomicAdd(&s_n, 1u);
                if (at < HEADS_LDS) { s_v[at] = (Id)v; s_hi[at] = F.hi; s_lo[at] = F.lo; }
                else {                                                                     // (more heads than the block's list holds: one by one)
                    const unsigned long long pos = atomicAdd(n_heads, 1ull);
                    if (write && pos < cap) { head_v[pos] = (Id)v; key_hi[pos] = F.hi; key_lo[pos] = F.lo; }
                }
            }
        }
    }
    __syncthreads();
    const uint32_t nl = s_n < HEADS_LDS ? s_n : HEADS_LDS;
    if (threadIdx.x == 0 && nl) s_base = atomicAdd(n_heads, (unsigned long long)nl);
    __syncthreads();
    if (write)
        for (uint32_t j = threadIdx.x; j < nl; j += 256) {
            const unsigned long long pos = s_base + j;
            if (pos < cap) { head_v[pos] = s_v[j]; key_hi[pos] = s_hi[j]; key_lo[pos] = s_lo[j]; }
        }
}
__global__ void __launch_bounds__(256) k_iota(uint64_t n, uint32_t* __restrict__ a) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(256) k_gather_u64(uint64_t n, const uint64_t* __restrict__ src, const uint32_t* __restrict__ perm,
                                                     uint64_t* __restrict__ dst) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
// canonical mode: edge e = e-th head in sorted order
template <class Id>
__global__ void __launch_bounds__(256) k_edge_from_sorted(uint64_t E, const uint32_t* __restrict__ perm, const Id* __restrict__ head_v,
                                                           const uint32_t* __restrict__ own, unsigned long long* __restrict__ w,
                                                           Id* __restrict__ edge_head, uint32_t* __restrict__ edge_nk) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const Id v = head_v[perm[e]];
    Id end; uint32_t rk;
    rank_of<Id>(own, w, v, end, rk);
    edge_head[e] = v; edge_nk[e] = rk + 1;
    __hip_atomic_store(&w[v ^ (Id)1], RankW<Id>::pack(e + 1, (Id)(v ^ (Id)1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // v^1 is a chain end
}
// replay mode: edge e = the unipath whose canonical first 60-mer is hint e's
template <class Id>
__global__ void __launch_bounds__(256) k_edge_from_hint(uint64_t E, const uint64_t* __restrict__ hk_hi, const uint64_t* __restrict__ hk_lo,
                                                         const uint32_t* __restrict__ hk_len, const Slot* __restrict__ table, uint64_t mask,
                                                         const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo,
                                                         const uint8_t* __restrict__ is_head, const uint32_t* __restrict__ own,
                                                         unsigned long long* __restrict__ w, Id* __restrict__ edge_head,
                                                         uint32_t* __restrict__ edge_nk, uint32_t* __restrict__ flags) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    Kmer k{hk_hi[e], hk_lo[e]};
    bool r = kmer_canon(k);
    int64_t s = table_find(table, mask, shi, slo, k);
    edge_head[e] = 0; edge_nk[e] = 1;
    if (s < 0) { atomicOr(&flags[1], (uint32_t)GE_HINT_MISS); return; }
    const Id v = (Id)(2 * (uint64_t)s + (r ? 1u : 0u));
    if (!is_head[v]) { atomicOr(&flags[1], (uint32_t)GE_HINT_MISS); return; }
    Id end; uint32_t rk;
    rank_of<Id>(own, w, v, end, rk);
    if (hk_len[e] != rk + K) { atomicOr(&flags[1], (uint32_t)GE_HINT_LEN); return; }
    const unsigned long long old = atomicExch(&w[v ^ (Id)1], RankW<Id>::pack(e + 1, (Id)(v ^ (Id)1)));
    if (RankW<Id>::dist(old) != 0) atomicOr(&flags[1], (uint32_t)GE_HINT_DUP);
    edge_head[e] = v; edge_nk[e] = rk + 1;
}
__global__ void __launch_bounds__(256) k_edge_len(uint64_t E, const uint32_t* __restrict__ edge_nk, uint32_t* __restrict__ len) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) len[e] = edge_nk[e] + (K - 1);
}
// every k-mer learns (edge, offset) (addEdge :287-301) and deposits its base(s) of the edge sequence.  srec (may be null): the 32-B
// record {key, KDef} per k-mer that read pathing through the dictionary reads (one GPU); with the pathing index (sharded dictionary) a
// k-mer's (edge, offset) is wherever its 60 bases lie in the edge sequences and nothing is written per k-mer.
template <class Id>
__global__ void __launch_bounds__(256) k_assign(uint64_t S, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo,
                                                 const uint32_t* __restrict__ own, const unsigned long long* __restrict__ w,
                                                 const uint64_t* __restrict__ edge_off, KRec* __restrict__ srec,
                                                 uint8_t* __restrict__ codes, uint32_t* __restrict__ flags) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    Id e0, e1; uint32_t rk0, rk1;
    rank_of<Id>(own, w, (Id)(2 * i), e0, rk0);
    rank_of<Id>(own, w, (Id)(2 * i + 1), e1, rk1);
    // the chain through node 2i starts at head e1^1, the one through 2i+1 at e0^1: exactly one of the two is a canonical head
    uint32_t e = edge_of_end<Id>(w, e1), off = rk1;
    bool rev = false;
    if (e == NONE32) { e = edge_of_end<Id>(w, e0); off = rk0; rev = true; }
    Kmer k{shi[i], slo[i]};
    if (e == NONE32) { atomicOr(&flags[1], (uint32_t)GE_ASSIGN); if (srec) srec[i] = KRec{k.hi, k.lo, make_uint4(NONE32, 0, 0, 0)}; return; }
    const uint64_t eo = edge_off[e];
    if (srec) srec[i] = KRec{k.hi, k.lo, make_uint4(e | (rev ? 0x80000000u : 0u), off, (uint32_t)eo, (uint32_t)(eo >> 32) | ((rk0 + rk1 + 1u) << 8))};
    if (rev) k = kmer_rc(k);
    uint8_t* dst = codes + eo;
    if (off == 0) {
        for (unsigned t = 0; t < K; ++t) dst[t] = (uint8_t)kmer_base(k, t);
    } else dst[K - 1 + off] = (uint8_t)kmer_last(k);
}

// all edge bases as one 2-bit stream (for 16-bases-per-load comparisons in read pathing)
__global__ void __launch_bounds__(256) k_pack_codes(uint64_t nbytes, uint64_t nbases, const uint8_t* __restrict__ codes, uint8_t* __restrict__ bits) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbytes) return;
    unsigned v = 0;
#pragma unroll
    for (unsigned j = 0; j < 4; ++j) { uint64_t g = 4 * i + j; if (g < nbases) v |= (unsigned)(codes[g] & 3) << (2 * j); }
    bits[i] = (uint8_t)v;
}

// absence filter over every 31-mer of the edge stream (common.h).  31-mers that straddle two edges in the concatenated
// stream are inserted as well: harmless, a filter may only err towards "maybe present".  Lane = position; the bits of
// lanes that fall into the same word are ORed together with four shuffle steps and only the first lane of a run
// issues the atomic (equal words that are not neighbours in a run may be merged too: they ARE the same word).
// (w_lo, w_hi): only the words of that range are written, at filter[word - w_lo] -- the sharded graph phase has every rank build ITS
// slice of the words (the whole stream is scanned, the atomics -- what bounds this kernel -- are the slice's) and gathers the slices
__global__ void __launch_bounds__(256) k_filter32(uint64_t npos, const uint8_t* __restrict__ bits, unsigned long long* __restrict__ filter,
                                                   uint32_t fmask, uint32_t w_lo, uint32_t w_hi) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x & 63;
    uint32_t word = 0xFFFFFFFFu; unsigned long long mask = 0;
    if (g < npos) {
        const uint64_t b0 = g >> 2; const unsigned sh = 2 * (unsigned)(g & 3);
        uint64_t x = reinterpret_cast<const U64u*>(bits + b0)->v >> sh;
        if (sh) x |= (uint64_t)bits[b0 + 8] << (64 - sh);
        const FmerKey k = fmer_key(x);
        word = k.word & fmask; mask = k.mask;
    }
    // positions g, g+4, g+8, .. share windows: a run's lanes are 4 apart.  Gather along distance 4 (4, 8, 16, 32), then a
    // lane is the head of its run iff the lane 4 before it has another word.
    const uint32_t prev = __shfl_up(word, 4);
    const bool head = g < npos && (lane < 4 || prev != word);
#pragma unroll
    for (unsigned d = 4; d < 64; d <<= 1) {
        const uint32_t ow = __shfl_down(word, d);
        const unsigned long long om = __shfl_down(mask, d);
        if (lane + d < 64 && ow == word) mask |= om;
    }
    if (head && word - w_lo < w_hi - w_lo) atomicOr(&filter[word - w_lo], mask);
}

// ------------------------------------------------------------------------------ the pathing index (common.h EdgeIndex)
// positions of the edge stream where no 15-mer of an edge starts: the last MMER-1 bases of every edge (one bit per position)
__global__ void __launch_bounds__(256) k_index_tails(uint64_t E, const uint64_t* __restrict__ edge_off, const uint32_t* __restrict__ edge_nk,
                                                      uint32_t* __restrict__ bad) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const uint64_t g1 = edge_off[e] + edge_nk[e] + (K - 1), g0 = g1 - (MMER - 1);
    for (uint64_t wd = g0 >> 5; wd <= (g1 - 1) >> 5; ++wd) {
        const uint64_t lo = wd << 5;
        const unsigned a = g0 > lo ? (unsigned)(g0 - lo) : 0u, b = g1 - lo < 32 ? (unsigned)(g1 - lo) : 32u;       // bits [a, b)
        const uint32_t m = (b == 32 ? 0xFFFFFFFFu : (1u << b) - 1u) & ~((1u << a) - 1u);
        atomicOr(&bad[wd], m);
    }
}
// A block takes the IS = 2048 k-mer starts S0 .. S0+IS-1 (S0 = g0 - 45) and emits the entries of the IT = IS - 45 positions g0 .. g0+IT-1
// (every window that contains one of them starts in the block's range).  The 26-bit keys (idx_key >> 6, common.h) of the 15-mers at S0 .. S0+IS+44 go to LDS
// as key + 1, 0 where no 15-mer of an edge starts (the last 14 bases of an edge, outside the
// stream): a window that holds a 0 is not a k-mer of an edge -- its minimum is 0 and matches no position.  A thread takes EIGHT
// consecutive starts: 53 keys in registers, the 39 keys common to its eight windows reduced once, each window finished with the suffix
// / prefix minima of the other 14 -- 8.5 v_min per window instead of 45.  A position is an entry iff its key equals the minimum of some
// window that contains it; every such minimum is <= the key, so: iff the key equals the LARGEST of the minima of the thread's windows
// that contain it (ties: every position that attains a window minimum is kept).  WRITE = false only counts.
constexpr unsigned IS = 2048, IT = IS - (WIN - 1);
constexpr unsigned IT_EDGES = IT / K + 3;                               // edges that begin inside one block's positions, and the one before
// the edge that holds the first position of every block (a thread per edge writes the blocks that begin inside it)
__global__ void __launch_bounds__(256) k_index_tile_edge(uint64_t E, const uint64_t* __restrict__ edge_off, uint32_t* __restrict__ tile_edge) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const uint64_t a = edge_off[e], b = edge_off[e + 1];
    for (uint64_t t = (a + IT - 1) / IT; t * IT < b; ++t) tile_edge[t] = (uint32_t)e;
}
// MODE 1: entries go straight into the table `slots`; MODE 2: into a list (slots[position], mask = its capacity) -- the sharded graph phase
// builds the entries of ITS blocks of the stream (blk0: the first of them), gathers every rank's list and inserts them all (k_index_insert)
template <int MODE>
__global__ void __launch_bounds__(256) k_index_build(uint64_t nbases, const uint8_t* __restrict__ ebits, const uint32_t* __restrict__ bad, uint64_t E,
                                                      const uint64_t* __restrict__ edge_off, const uint32_t* __restrict__ tile_edge,
                                                      uint4* __restrict__ slots, uint64_t mask, unsigned long long* __restrict__ counter, uint64_t blk0) {
    constexpr bool WRITE = MODE != 0;
    constexpr unsigned NKEY = IS + (WIN - 1) + 3;                       // (+3: the last thread's 53 keys are read as 14 quads)
    __shared__ __attribute__((aligned(16))) uint32_t s_key[NKEY];
    __shared__ uint8_t s_flag[NKEY];
    const unsigned tid = threadIdx.x;
    const uint64_t blk = blk0 + blockIdx.x;
    const int64_t g0 = (int64_t)blk * IT, base = g0 - (int64_t)(WIN - 1);
    for (unsigned i = tid; i < NKEY; i += 256) {
        const int64_t g = base + i;
        uint32_t key = 0;
        if (g >= 0 && (uint64_t)g + MMER <= nbases && !((bad[(uint64_t)g >> 5] >> ((uint64_t)g & 31)) & 1u)) {
            const uint32_t f = stream16_global(ebits, (uint64_t)g) & 0x3FFFFFFFu, r = rc15(f);
            key = (idx_key(f < r ? f : r) >> 6) + 1u;
        }
        s_key[i] = key; s_flag[i] = 0;
    }
    __syncthreads();
    {
        uint32_t k[56];
#pragma unroll
        for (unsigned q = 0; q < 14; ++q) {
            const uint4 v = *reinterpret_cast<const uint4*>(&s_key[8 * tid + 4 * q]);
            k[4 * q] = v.x; k[4 * q + 1] = v.y; k[4 * q + 2] = v.z; k[4 * q + 3] = v.w;
        }
        uint32_t common = k[7];
#pragma unroll
        for (unsigned i = 8; i <= 45; ++i) common = min(common, k[i]);
        uint32_t M[8];
        {
            uint32_t sfx = 0xFFFFFFFFu, s7[8];
            s7[7] = 0xFFFFFFFFu;
#pragma unroll
            for (int jj = 6; jj >= 0; --jj) { sfx = min(sfx, k[jj]); s7[jj] = sfx; }
            uint32_t pfx = 0xFFFFFFFFu;
#pragma unroll
            for (unsigned jj = 0; jj < 8; ++jj) { if (jj) pfx = min(pfx, k[45 + jj]); M[jj] = min(min(common, s7[jj]), pfx); }
        }
        uint32_t PM[8], SM[8];                                         // max(M[0..i]), max(M[i..7])
        PM[0] = M[0];
#pragma unroll
        for (unsigned jj = 1; jj < 8; ++jj) PM[jj] = max(PM[jj - 1], M[jj]);
        SM[7] = M[7];
#pragma unroll
        for (int jj = 6; jj >= 0; --jj) SM[jj] = max(SM[jj + 1], M[jj]);
        if (PM[7]) {                                                   // some window of this thread is a k-mer
#pragma unroll
            for (unsigned i = 0; i < 53; ++i) {
                const uint32_t mx = i < 7 ? PM[i] : i <= 45 ? PM[7] : SM[i - 45];
                if (k[i] && k[i] == mx) s_flag[8 * tid + i] = 1;
            }
        }
    }
    __syncthreads();
    // The marked positions of g0 .. g0+IT-1 are gathered into a dense list (most iterations of a loop over the positions would have one
    // or two busy lanes in each wavefront, and every entry costs a claim), counted and reserved in the table's budget (half its slots): a
    // block that finds the budget spent inserts nothing -- the claims below always find an empty slot, and the host repeats the pass
    // with the table the count asks for.
    __shared__ unsigned s_cnt; __shared__ bool s_ok;
    __shared__ uint16_t s_list[IT];
    __shared__ uint64_t s_eoff[IT_EDGES];
    __shared__ uint32_t s_e0;
    if (tid == 0) { s_cnt = 0; s_e0 = tile_edge[blk]; }
    __syncthreads();
    const uint32_t e0 = s_e0;
    if (tid < IT_EDGES) s_eoff[tid] = (uint64_t)e0 + tid <= E ? edge_off[e0 + tid] : ~0ull;          // (edge_off[E] = nbases ends the last edge)
    for (unsigned t0 = 0; t0 < IT; t0 += 256) {
        const unsigned t = t0 + tid;
        const bool f = t < IT && s_flag[(WIN - 1) + t];
        const unsigned long long m = __ballot(f);
        unsigned wbase = 0;
        if ((tid & 63) == 0 && m) wbase = atomicAdd(&s_cnt, (unsigned)__builtin_popcountll(m));
        wbase = __shfl(wbase, 0);
        if (f) s_list[wbase + __builtin_popcountll(m & ((1ull << (tid & 63)) - 1))] = (uint16_t)t;
    }
    __syncthreads();
    const unsigned cnt = s_cnt;
    // the sides an entry is made for: the 16 bases before / behind the 15-mer must lie inside its edge
    __shared__ unsigned s_ent;
    if (tid == 0) s_ent = 0;
    __syncthreads();
    unsigned mine_e = 0;
    for (unsigned q = tid; q < cnt; q += 256) {
        const uint64_t g = (uint64_t)g0 + s_list[q];
        unsigned lo = 0, hi = IT_EDGES;                                 // the edge that holds g: the last one of the block's that begins at or before g
        while (hi - lo > 1) { const unsigned md = (lo + hi) >> 1; if (s_eoff[md] <= g) lo = md; else hi = md; }
        const unsigned sides = (g >= s_eoff[lo] + 16 ? 1u : 0u) | (g + MMER + 16 <= s_eoff[lo + 1] ? 2u : 0u);          // bit 0: bases before, bit 1: bases behind (edge orientation)
        s_list[q] = (uint16_t)(s_list[q] | (sides << 11) | (0u));       // positions < 2048: bits 10:0; sides in bits 12:11
        s_flag[q] = (uint8_t)lo;                                        // (the flags have been consumed: the edge's place in s_eoff rides here)
        mine_e += (sides & 1) + (sides >> 1);
    }
    for (int d = 32; d > 0; d >>= 1) mine_e += __shfl_down(mine_e, d);
    if ((tid & 63) == 0 && mine_e) atomicAdd(&s_ent, mine_e);
    __syncthreads();
    const unsigned ent = s_ent;
    __shared__ unsigned long long s_before;
    __shared__ unsigned s_at;
    if (tid == 0) {
        const unsigned long long before = ent ? atomicAdd(counter, (unsigned long long)ent) : 0ull;
        s_ok = MODE == 2 ? before + ent <= mask : 2 * (before + ent) <= mask + 1;
        s_before = before; s_at = 0;
    }
    __syncthreads();
    if (!WRITE || !s_ok) return;
    for (unsigned q = tid; q < 2 * cnt; q += 256) {                     // (position, side) pairs
        const unsigned e_ = q >> 1, side_edge = q & 1;                  // side_edge 0: the bases before the 15-mer, 1: behind it (edge orientation)
        const unsigned t = s_list[e_] & 2047u, sides = s_list[e_] >> 11;
        if (!((sides >> side_edge) & 1)) continue;
        const uint64_t g = (uint64_t)g0 + t;
        const uint32_t f = stream16_global(ebits, g) & 0x3FFFFFFFu, r = rc15(f);
        const bool sb = !(f < r);                                       // the reverse complement is the canonical strand
        const uint32_t cfwd = side_edge ? stream16_global(ebits, g + MMER) : stream16_global(ebits, g - 16);
        // in the canonical orientation of the 15-mer: the bases behind it are its RIGHT context on the forward strand, the (reverse-complemented)
        // bases before it are its right context on the other
        const bool right = (side_edge == 1) != sb;
        const uint32_t key = idx_hash(sb ? r : f, sb ? rc32(cfwd) : cfwd, right);
        const unsigned long long claim = (unsigned long long)((key & ~1u) | (sb ? 1u : 0u)) | ((unsigned long long)(e0 + s_flag[e_]) << 32);
        if (MODE == 2) {                                                 // the entry as it will stand in a table, and the full key in its place's stead: bucket_mix(key) is needed again
            const unsigned long long at = s_before + atomicAdd(&s_at, 1u);
            slots[at] = make_uint4(key, (uint32_t)(claim >> 32), (uint32_t)g, (uint32_t)(g >> 32) | ((sb ? 1u : 0u) << 31));
            continue;
        }
        uint64_t sl = bucket_mix(key) & mask;
        for (;;) {                                                       // the (x, y) half is the claim; y == NONE32: empty
            unsigned long long* p = reinterpret_cast<unsigned long long*>(&slots[sl]);
            const unsigned long long old = atomicCAS(p, 0xFFFFFFFFFFFFFFFFull, claim);
            if (old == 0xFFFFFFFFFFFFFFFFull) { p[1] = g | ((unsigned long long)(key & 1u) << 62); break; }      // (bit 30 of w: the key's bit 0, which x gave to the strand)
            sl = (sl + 1) & mask;
        }
    }
}
// the listed entries of every rank -> the table (list entry: x = full key, y = unipath, z | (w & 0x7FFFFFFF) << 32 = stream position, w bit 31 = strand)
__global__ void __launch_bounds__(256) k_index_insert(uint64_t n, const uint4* __restrict__ list, uint4* __restrict__ slots, uint64_t mask) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 e = list[i];
    const unsigned long long claim = (unsigned long long)((e.x & ~1u) | (e.w >> 31)) | ((unsigned long long)e.y << 32);
    const unsigned long long g = (unsigned long long)e.z | ((unsigned long long)(e.w & 0x7FFFFFFFu) << 32);
    uint64_t sl = bucket_mix(e.x) & mask;
    for (;;) {
        unsigned long long* p = reinterpret_cast<unsigned long long*>(&slots[sl]);
        const unsigned long long old = atomicCAS(p, 0xFFFFFFFFFFFFFFFFull, claim);
        if (old == 0xFFFFFFFFFFFFFFFFull) { p[1] = g | ((unsigned long long)(e.x & 1u) << 62); break; }
        sl = (sl + 1) & mask;
    }
}
// per solid k-mer: where the index finds it (tests; the sharded graph phase's self check)
__global__ void __launch_bounds__(256) k_index_probe(uint64_t S, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo, EdgeIndex X,
                                                      int32_t* __restrict__ edge, uint32_t* __restrict__ off) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const Kmer k{shi[i], slo[i]};
    const uint64_t a = rev2_64(k.hi << 4), b = rev2_64(k.lo << 4);
    IdxHit h;
    const bool ok = index_find(X, a | (b << 60), b >> 4, h);
    if (edge) edge[i] = ok ? (int32_t)h.e : -1;
    if (off) off[i] = ok ? h.off : 0u;
}

// ------------------------------------------------------------------------------ a8: HBVFromEdges.cc:76-154
// a unipath of ONE palindromic k-mer is one object (:94,142): read off its 60 base codes
__global__ void __launch_bounds__(256) k_edge_nobj(uint64_t E, const uint32_t* __restrict__ edge_nk, const uint64_t* __restrict__ edge_off,
                                                    const uint8_t* __restrict__ codes, uint32_t* __restrict__ nobj) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    bool pal = edge_nk[e] == 1;
    if (pal) {
        const uint8_t* q = codes + edge_off[e];
        for (unsigned t = 0; t < K / 2; ++t) pal = pal && q[t] == 3u - q[K - 1 - t];
    }
    nobj[e] = pal ? 1u : 2u;
}
__global__ void __launch_bounds__(256) k_edge_xlat(uint64_t E, const uint32_t* __restrict__ nobj, const uint64_t* __restrict__ ooff,
                                                    int32_t* __restrict__ fwdX, int32_t* __restrict__ revX, uint32_t* __restrict__ obj_edge) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    uint64_t o = ooff[e];
    fwdX[e] = (int32_t)o; obj_edge[o] = (uint32_t)(e << 1);
    if (nobj[e] == 2) { revX[e] = (int32_t)(o + 1); obj_edge[o + 1] = (uint32_t)(e << 1) | 1u; }
    else revX[e] = (int32_t)o;
}
__device__ inline unsigned obj_base(const uint8_t* codes, uint64_t eoff, uint32_t len, bool rc, uint32_t t) {
    return rc ? 3u - codes[eoff + (len - 1 - t)] : codes[eoff + t];
}
// one thread per edge end: FNV1a over the 59 base codes (math/Hash.h:26-35) + the 118-bit sequence
__global__ void __launch_bounds__(256) k_ends(uint64_t NO, const uint32_t* __restrict__ obj_edge, const uint64_t* __restrict__ edge_off,
                                               const uint32_t* __restrict__ edge_nk, const uint8_t* __restrict__ codes,
                                               uint64_t* __restrict__ ehash, uint64_t* __restrict__ ehi, uint64_t* __restrict__ elo) {
    uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 2 * NO) return;
    uint64_t o = id >> 1; bool distal = id & 1;
    uint32_t oe = obj_edge[o], e = oe >> 1; bool rc = oe & 1;
    uint32_t len = edge_nk[e] + (K - 1);
    uint64_t eoff = edge_off[e];
    uint32_t t0 = distal ? len - (K - 1) : 0;
    uint64_t h = 14695981039346656037ull, hi = 0, lo = 0;
    for (unsigned t = 0; t < K - 1; ++t) {
        unsigned b = obj_base(codes, eoff, len, rc, t0 + t);
        h = 1099511628211ull * (h ^ b);
        if (t < 30) hi = (hi << 2) | b; else lo = (lo << 2) | b;
    }
    ehash[id] = h; ehi[id] = hi; elo[id] = lo;
}
__global__ void __launch_bounds__(256) k_end_flags(uint64_t n, const uint32_t* __restrict__ perm, const uint64_t* __restrict__ ehash,
                                                    const uint64_t* __restrict__ ehi, const uint64_t* __restrict__ elo, uint32_t* __restrict__ flag,
                                                    uint32_t* __restrict__ shared_hash /* set if two different ends carry one hash */) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    uint32_t f = 0;
    if (j > 0) {
        uint32_t a = perm[j - 1], b = perm[j];
        const bool same_seq = ehi[a] == ehi[b] && elo[a] == elo[b];
        f = (ehash[a] != ehash[b] || !same_seq) ? 1u : 0u;
        if (ehash[a] == ehash[b] && !same_seq) *shared_hash = 1u;
    }
    flag[j] = f;
}
// the unipaths sorted by the first 30 bases of their first k-mer: runs of equal words are ordered by the other 30 bases (one thread per run)
__global__ void __launch_bounds__(256) k_tie_sort_lo(uint64_t E, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ lo, uint32_t* __restrict__ perm,
                                                      unsigned max_run, uint32_t* __restrict__ long_run /* set if a run is longer: the caller sorts by both words */) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= E || (j > 0 && shi[j] == shi[j - 1])) return;
    uint64_t b = j + 1;
    while (b < E && shi[b] == shi[j]) ++b;
    if (b - j > max_run) { *long_run = 1u; return; }              // (thousands of unipaths that start with the same 30 bases: a low-complexity genome)
    for (uint64_t i = j + 1; i < b; ++i) {
        const uint32_t x = perm[i]; const uint64_t lx = lo[x];
        uint64_t t = i;
        while (t > j && lo[perm[t - 1]] > lx) { perm[t] = perm[t - 1]; --t; }
        perm[t] = x;
    }
}
__global__ void __launch_bounds__(256) k_end_vertices(uint64_t n, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ flag,
                                                       const uint64_t* __restrict__ excl, int32_t* __restrict__ left, int32_t* __restrict__ right) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    int32_t vid = (int32_t)(excl[j] + flag[j]);
    uint32_t id = perm[j];
    if (id & 1) right[id >> 1] = vid; else left[id >> 1] = vid;
}
__global__ void __launch_bounds__(256) k_adj_keys(uint64_t NO, const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                                                   uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ deg) {
    uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= NO) return;
    keys[o] = ((uint64_t)(uint32_t)a[o] << 32) | (uint32_t)b[o];
    vals[o] = (uint32_t)o;
    atomicAdd(&deg[a[o]], 1u);
}
__global__ void __launch_bounds__(256) k_adj_out(uint64_t NO, const uint32_t* __restrict__ vals, const int32_t* __restrict__ other,
                                                  int32_t* __restrict__ out_v, int32_t* __restrict__ out_e) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= NO) return;
    uint32_t o = vals[j];
    out_e[j] = (int32_t)o; out_v[j] = other[o];
}

// ------------------------------------------------------------------------------ object table for read pathing
// One 32-B record per edge object: the object that FOLLOWS it for each of the four possible next bases (the out-edges of its right
// vertex all begin with the vertex's 59-mer and differ in their 60th base), and where the object itself lies in the packed edge
// stream.  When a read runs off the END of a unipath, its next 60-mer starts with that 59-mer: the successor is picked by one read
// base -- no dictionary probe -- or, if there is none for that base, the k-mer is not solid (step2_path.hip).
__device__ inline unsigned obj_base(const uint8_t* codes, uint64_t eoff, uint32_t len, bool rc, uint32_t t);
__global__ void __launch_bounds__(256) k_obj_table(uint64_t NO, const uint32_t* __restrict__ obj_edge, const uint32_t* __restrict__ edge_nk,
                                                    const uint64_t* __restrict__ edge_off, const uint8_t* __restrict__ codes,
                                                    const int32_t* __restrict__ right, const uint64_t* __restrict__ from_off,
                                                    const int32_t* __restrict__ from_e, ObjRec* __restrict__ tab) {
    const uint64_t o = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= NO) return;
    const uint32_t oe = obj_edge[o], e = oe >> 1;
    const uint64_t eo = edge_off[e];
    ObjRec r;
    r.succ[0] = r.succ[1] = r.succ[2] = r.succ[3] = -1;
    r.eo_lo = (uint32_t)eo; r.eo_hi = (uint32_t)(eo >> 32); r.elen = edge_nk[e] + (K - 1); r.edge_rc = oe;
    const int32_t v = right[o];
    for (uint64_t k = from_off[v]; k < from_off[v + 1]; ++k) {
        const int32_t o2 = from_e[k];
        const uint32_t oe2 = obj_edge[o2], e2 = oe2 >> 1, len2 = edge_nk[e2] + (K - 1);
        const unsigned b = obj_base(codes, edge_off[e2], len2, oe2 & 1, K - 1);
        r.succ[b] = o2;
    }
    tab[o] = r;
}

// ------------------------------------------------------------------------------ driver
static inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }     // n < 2^32 - 256: a grid holds fewer than 2^32 threads

// the pathing index over c.d_edge_bits (E edges: d_edge_off, d_edge_nk): a counting pass sizes the table (load 0.2 .. 0.4), a second pass fills it
int build_index(Ctx& c) {
    hipStream_t st = c.stream;
    if (c.d_index) { c.release(c.d_index); c.d_index = nullptr; }
    c.index_cap = 0; c.index_entries = 0;
    const uint64_t nb = c.edge_bases, E = c.E;
    uint32_t* bad = nullptr; unsigned long long* d_n = nullptr;
    const uint64_t nwords = nb / 32 + 2;
    W2_ALLOC(bad, uint32_t, nwords); W2_ALLOC(d_n, unsigned long long, 1);
    W2_HIP(hipMemsetAsync(bad, 0, nwords * 4, st));
    W2_HIP(hipMemsetAsync(d_n, 0, 8, st));
    const uint64_t nblk = (nb + IT - 1) / IT;
    if (nblk >= (1ull << 31)) { c.err = "edge stream too long for the pathing index"; return W2RAP_E_LIMIT; }
    unsigned long long n_ent = 0;
    uint32_t* tile_edge = nullptr;
    W2_ALLOC(tile_edge, uint32_t, nblk + 1);
    if (E) {
        LAUNCH(c, "k_index_tails", k_index_tails, dim3(grid_for(E)), dim3(256), 0, E, c.d_edge_off, c.d_edge_nk, bad);
        LAUNCH(c, "k_index_tile_edge", k_index_tile_edge, dim3(grid_for(E)), dim3(256), 0, E, c.d_edge_off, tile_edge);
    }
    // ONE filling pass into a table laid out for the usual density (4 / 47 entries per base -- two sides per position: load 0.17 .. 0.35); the pass counts its
    // entries, and a sequence that makes more than half a table of them (low complexity: every position of a run of equal keys is kept)
    // gets a second pass with the table its count asks for
    uint64_t cap = 1024;
    while (cap < nb / 4) cap <<= 1;
    if (test_hook("W2RAP_TEST_INDEX_SMALL")) cap = 1024;
    for (int attempt = 0;; ++attempt) {
        W2_ALLOC(c.d_index, uint4, cap);
        W2_HIP(hipMemsetAsync(c.d_index, 0xFF, cap * sizeof(uint4), st));
        W2_HIP(hipMemsetAsync(d_n, 0, 8, st));
        if (!E) break;
        LAUNCH(c, "k_index_fill", k_index_build<1>, dim3((unsigned)nblk), dim3(256), 0, nb, c.d_edge_bits, bad, E, c.d_edge_off, tile_edge, c.d_index, cap - 1, d_n, (uint64_t)0);
        W2_HIP(hipMemcpyAsync(&n_ent, d_n, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        if (2 * n_ent <= cap) break;
        if (attempt >= 2) { c.err = "pathing index: table sizing failed"; return W2RAP_E_LIMIT; }
        c.release(c.d_index); c.d_index = nullptr;
        cap = 1024;
        while (2 * cap < 5 * n_ent) cap <<= 1;
    }
    c.index_cap = cap; c.index_entries = n_ent;
    if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] pathing index: %llu entries for %llu edge bases (%.3f per base), %llu slots of 16 B\n", n_ent, (unsigned long long)nb,
                                       nb ? (double)n_ent / (double)nb : 0.0, (unsigned long long)cap);
    c.release(bad); c.release(d_n); c.release(tile_edge);
    return index_harden(c);
}
// The same in two steps for the sharded graph phase: (1) the entries of the blocks [nblk r / world, nblk (r+1) / world) of the stream as a
// list (*d_list, *n_list; the caller releases it), (2) the table from the gathered lists of all ranks.
int index_entries_slice(Ctx& c, unsigned rank, unsigned world, uint4** d_list, uint64_t* n_list) {
    hipStream_t st = c.stream;
    const uint64_t nb = c.edge_bases, E = c.E;
    *d_list = nullptr; *n_list = 0;
    const uint64_t nblk = (nb + IT - 1) / IT;
    const uint64_t b0 = nblk * rank / world, b1 = nblk * (rank + 1) / world;
    if (nblk >= (1ull << 31)) { c.err = "edge stream too long for the pathing index"; return W2RAP_E_LIMIT; }
    uint32_t* bad = nullptr; unsigned long long* d_n = nullptr; uint32_t* tile_edge = nullptr;
    const uint64_t nwords = nb / 32 + 2;
    W2_ALLOC(bad, uint32_t, nwords); W2_ALLOC(d_n, unsigned long long, 1); W2_ALLOC(tile_edge, uint32_t, nblk + 1);
    W2_HIP(hipMemsetAsync(bad, 0, nwords * 4, st));
    if (E) {
        LAUNCH(c, "k_index_tails", k_index_tails, dim3(grid_for(E)), dim3(256), 0, E, c.d_edge_off, c.d_edge_nk, bad);
        LAUNCH(c, "k_index_tile_edge", k_index_tile_edge, dim3(grid_for(E)), dim3(256), 0, E, c.d_edge_off, tile_edge);
    }
    uint64_t cap = (b1 - b0) * IT / 8 + 4096;                              // 4 / 47 entries per base is the usual density; an overflow is followed by the exact size
    unsigned long long n_ent = 0;
    uint4* list = nullptr;
    for (int attempt = 0;; ++attempt) {
        W2_ALLOC(list, uint4, cap);
        W2_HIP(hipMemsetAsync(d_n, 0, 8, st));
        if (E && b1 > b0) LAUNCH(c, "k_index_list", k_index_build<2>, dim3((unsigned)(b1 - b0)), dim3(256), 0, nb, c.d_edge_bits, bad, E, c.d_edge_off, tile_edge, list, cap, d_n, b0);
        W2_HIP(hipMemcpyAsync(&n_ent, d_n, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        if (n_ent <= cap) break;
        if (attempt) { c.err = "pathing index: entry list overflow after resizing"; return W2RAP_E_LIMIT; }
        c.release(list);
        cap = n_ent + 16;
    }
    c.release(bad); c.release(d_n); c.release(tile_edge);
    *d_list = list; *n_list = n_ent;
    return 0;
}
int index_from_entries(Ctx& c, const uint4* d_all, uint64_t n_all, bool harden) {
    hipStream_t st = c.stream;
    if (c.d_index) { c.release(c.d_index); c.d_index = nullptr; }
    uint64_t cap = 1024;
    while (2 * cap < 5 * n_all) cap <<= 1;
    W2_ALLOC(c.d_index, uint4, cap);
    W2_HIP(hipMemsetAsync(c.d_index, 0xFF, cap * sizeof(uint4), st));
    if (n_all) LAUNCH(c, "k_index_insert", k_index_insert, dim3(grid_for(n_all)), dim3(256), 0, n_all, d_all, c.d_index, cap - 1);
    W2_HIP(hipGetLastError());
    c.index_cap = cap; c.index_entries = n_all; c.index_prebuilt = true;
    if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] pathing index: %llu entries gathered for %llu edge bases, %llu slots of 16 B\n", (unsigned long long)n_all,
                                       (unsigned long long)c.edge_bases, (unsigned long long)cap);
    if (c.d_xindex) { c.release(c.d_xindex); c.d_xindex = nullptr; }
    c.xindex_cap = 0; c.xindex_kmers = 0;
    return harden ? index_harden(c) : 0;
}
// the absence filter's geometry for the current edge stream (0 words: no filter), and one rank's slice of its words
uint64_t filter32_words(const Ctx& c) {
    if (c.edge_bases < FMER || getenv("W2RAP_NO_FILTER32") || c.edge_bases > (1ull << 33)) return 0;
    uint64_t fw = 1024;
    while (fw * 4 < c.edge_bases) fw <<= 1;
    return fw;
}
// (on the SIDE stream, behind what the main stream has queued so far: the caller synchronises c.stream2 before it uses the slice -- the sharded
//  graph phase lists, gathers and inserts the index entries meanwhile)
int filter32_slice(Ctx& c, unsigned rank, unsigned world, unsigned long long** d_slice, uint64_t* n_words) {
    hipStream_t st = c.stream2 ? c.stream2 : c.stream;
    const uint64_t fw = filter32_words(c);
    *d_slice = nullptr; *n_words = 0;
    if (!fw) return 0;
    const uint64_t lo = fw * rank / world, hi = fw * (rank + 1) / world;
    unsigned long long* p = nullptr;
    W2_ALLOC(p, unsigned long long, hi - lo + 1);
    if (st != c.stream) {
        hipEvent_t ev;
        W2_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        W2_HIP(hipEventRecord(ev, c.stream));
        W2_HIP(hipStreamWaitEvent(st, ev, 0));
        (void)hipEventDestroy(ev);
    }
    W2_HIP(hipMemsetAsync(p, 0, (hi - lo + 1) * 8, st));
    const uint64_t npos = c.edge_bases - (FMER - 1);
    LAUNCH_ON(c, st, "k_filter32", k_filter32, dim3(grid_for(npos)), dim3(256), 0, npos, c.d_edge_bits, p, (uint32_t)(fw - 1), (uint32_t)lo, (uint32_t)hi);
    W2_HIP(hipGetLastError());
    *d_slice = p; *n_words = hi - lo;
    return 0;
}
EdgeIndex edge_index(const Ctx& c) {
    return EdgeIndex{c.d_index, c.index_cap - 1, c.d_edge_bits, c.d_edge_off, c.d_edge_nk, c.edge_bases, c.xindex_cap ? c.d_xindex : nullptr, c.xindex_cap ? c.xindex_cap - 1 : 0};
}
// ---- the exact table (common.h).  k_index_mark: a slot whose key has more than IDX_HARD entries in its probe sequence gets bit 31 of w.
__global__ void __launch_bounds__(256) k_index_mark(uint64_t cap, uint4* __restrict__ slots, uint64_t mask, unsigned long long* __restrict__ n_hard /* [64] striped */,
                                                     uint32_t* __restrict__ flags /* [1]: an occupied run longer than the walk's bound */, unsigned run_max) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hard = false;
    if (i < cap) {
        const uint4 me = slots[i];
        if (me.y != NONE32) {
            // (a stored entry no longer knows its home slot -- bit 0 of its key gave way to the strand --, but linear probing keeps the entries of one
            //  key in ONE run of occupied slots: the whole run around this slot is counted, back to the empty slot in front of it and on to the next)
            const uint32_t key = me.x & ~1u, kb0 = (me.w >> 30) & 1u;
            unsigned n = 0;
            uint64_t s0 = i;
            // (the walk is bounded; a run longer than the bound would be counted from different starting points by different entries of one
            //  key, and index_find relies on "all entries of a marked key are marked": reported, the host answers W2RAP_E_LIMIT)
            unsigned guard = 0;
            for (; guard < run_max; ++guard) { const uint64_t b = (s0 - 1) & mask; if (slots[b].y == NONE32) break; s0 = b; }
            if (guard == run_max) flags[1] = 1u;
            for (uint64_t s = s0;; s = (s + 1) & mask) {
                const uint4 v = slots[s];
                if (v.y == NONE32) break;
                if ((v.x & ~1u) == key && ((v.w >> 30) & 1u) == kb0 && ++n > IDX_HARD) break;
            }
            hard = n > IDX_HARD;
            if (hard) atomicOr(&slots[i].w, 0x80000000u);
        }
    }
    const unsigned long long m = __ballot(hard);
    if (m && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(m)) atomicAdd(&n_hard[blockIdx.x & 63u], (unsigned long long)__builtin_popcountll(m));
}
// every k-mer that covers the 15-mer of a marked entry and lies inside the entry's unipath -> the exact table (idempotent: word 0 holds tag and position)
__global__ void __launch_bounds__(256) k_exact_insert(uint64_t cap, const uint4* __restrict__ slots, uint4* __restrict__ xs, uint64_t xmask, const uint8_t* __restrict__ ebits,
                                                       const uint64_t* __restrict__ edge_off, const uint32_t* __restrict__ edge_nk, uint32_t* __restrict__ flags /* [0] table full */,
                                                       unsigned long long* __restrict__ n_in /* [64] striped: k-mers inserted */) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const uint4 me = slots[i];
    if (me.y == NONE32 || !(me.w >> 31)) return;
    const uint64_t g = (uint64_t)me.z | ((uint64_t)(me.w & 0x3FFFFFFFu) << 32), eo = edge_off[me.y];
    const uint32_t nk = edge_nk[me.y];
    unsigned mine = 0;
    for (unsigned d = 0; d < WIN; ++d) {
        if (g < eo + d) break;                                             // the k-mer would start in front of the unipath
        const uint64_t P = g - d;
        if (P - eo >= nk) continue;                                        // ... or end behind it
        const U128u w = *reinterpret_cast<const U128u*>(ebits + (P >> 2));
        const unsigned sh = 2 * (unsigned)(P & 3);
        const uint64_t lo = sh ? (w.a >> sh) | (w.b << (64 - sh)) : w.a, hi = (w.b >> sh) & ((1ull << 56) - 1);
        const uint64_t ra = rev2_64(hi), rb = rev2_64(lo);
        const uint64_t rlo = ~((ra >> 8) | (rb << 56)), rhi = ~(rb >> 8) & ((1ull << 56) - 1);
        const uint64_t h = exact_hash(lo, hi, rlo, rhi);
        const unsigned long long claim = ((h >> 34) << 34) | P;
        uint64_t s = h & xmask;
        for (unsigned probes = 0;; ++probes, s = (s + 1) & xmask) {
            if (probes > 4096) { flags[0] = 1u; break; }                   // the table is too small for its load: the caller doubles it
            unsigned long long* p0 = reinterpret_cast<unsigned long long*>(&xs[s]);
            const unsigned long long old = atomicCAS(p0, XEMPTY, claim);
            if (old == XEMPTY) { xs[s].z = me.y; ++mine; break; }
            if (old == claim) break;                                       // the same k-mer, from the entry of the other side or of a neighbouring minimizer
        }
    }
    if (mine) atomicAdd(&n_in[blockIdx.x & 63u], (unsigned long long)mine);
}
// ---- the same SHARDED (step2_shard.hip): a rank examines ITS part of the gathered entry list -- an entry knows its home slot, its key's entries are
// counted along the probe sequence --, the hard entries of all ranks are gathered (16 B each), every rank marks them in its table and builds the exact
// table from them: the table scan above, proportional to the job on every rank, becomes 1/N of the entries per rank.
__global__ void __launch_bounds__(256) k_index_hard_list(uint64_t lo, uint64_t hi, const uint4* __restrict__ list, const uint4* __restrict__ slots, uint64_t mask,
                                                          unsigned long long* __restrict__ n_out, uint64_t cap, uint4* __restrict__ out) {
    const uint64_t i = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hard = false;
    uint4 e = make_uint4(0, 0, 0, 0);
    if (i < hi) {
        e = list[i];
        const uint32_t key = e.x & ~1u;
        unsigned n = 0;
        for (uint64_t s = bucket_mix(e.x) & mask;; s = (s + 1) & mask) {
            const uint4 v = slots[s];
            if (v.y == NONE32) break;
            if ((v.x & ~1u) == key && ((v.w >> 30) & 1u) == (e.x & 1u) && ++n > IDX_HARD) break;
        }
        hard = n > IDX_HARD;
    }
    const unsigned long long m = __ballot(hard);
    if (!m) return;
    const unsigned lane = threadIdx.x & 63u, leader = (unsigned)__builtin_ctzll(m);
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(&n_out[blockIdx.x & 63u], (unsigned long long)__builtin_popcountll(m));
    base = __shfl(base, (int)leader);
    // (64 striped sub-lists of cap entries each; the host packs them)
    if (hard) { const unsigned long long at = base + (unsigned long long)__builtin_popcountll(m & ((1ull << lane) - 1ull)); if (at < cap) out[(uint64_t)(blockIdx.x & 63u) * cap + at] = e; }
}
__global__ void __launch_bounds__(256) k_index_hard_pack(uint64_t cap, const unsigned long long* __restrict__ cnt, const unsigned long long* __restrict__ pre, const uint4* __restrict__ in,
                                                          uint4* __restrict__ out) {
    const unsigned k = blockIdx.y;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt[k]) out[pre[k] + i] = in[(uint64_t)k * cap + i];
}
// the gathered hard entries: mark each in this rank's table, put the k-mers around it into the exact table
__global__ void __launch_bounds__(256) k_index_hard_apply(uint64_t n, const uint4* __restrict__ hard, uint4* __restrict__ slots, uint64_t mask, uint4* __restrict__ xs, uint64_t xmask,
                                                           const uint8_t* __restrict__ ebits, const uint64_t* __restrict__ edge_off, const uint32_t* __restrict__ edge_nk,
                                                           uint32_t* __restrict__ flags, unsigned long long* __restrict__ n_in) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 e = hard[i];
    const uint32_t sx = (e.x & ~1u) | (e.w >> 31);                        // the entry as the table holds it
    const uint64_t g = (uint64_t)e.z | ((uint64_t)(e.w & 0x7FFFFFFFu) << 32);
    for (uint64_t s = bucket_mix(e.x) & mask;; s = (s + 1) & mask) {
        const uint4 v = slots[s];
        if (v.y == NONE32) break;
        if (v.x == sx && v.y == e.y && v.z == (uint32_t)g && (v.w & 0x3FFFFFFFu) == (uint32_t)(g >> 32)) { atomicOr(&slots[s].w, 0x80000000u); break; }
    }
    const uint64_t eo = edge_off[e.y];
    const uint32_t nk = edge_nk[e.y];
    unsigned mine = 0;
    for (unsigned d = 0; d < WIN; ++d) {
        if (g < eo + d) break;
        const uint64_t P = g - d;
        if (P - eo >= nk) continue;
        const U128u w = *reinterpret_cast<const U128u*>(ebits + (P >> 2));
        const unsigned sh = 2 * (unsigned)(P & 3);
        const uint64_t lo = sh ? (w.a >> sh) | (w.b << (64 - sh)) : w.a, hi = (w.b >> sh) & ((1ull << 56) - 1);
        const uint64_t ra = rev2_64(hi), rb = rev2_64(lo);
        const uint64_t rlo = ~((ra >> 8) | (rb << 56)), rhi = ~(rb >> 8) & ((1ull << 56) - 1);
        const uint64_t h = exact_hash(lo, hi, rlo, rhi);
        const unsigned long long claim = ((h >> 34) << 34) | P;
        uint64_t s = h & xmask;
        for (unsigned probes = 0;; ++probes, s = (s + 1) & xmask) {
            if (probes > 4096) { flags[0] = 1u; break; }
            unsigned long long* p0 = reinterpret_cast<unsigned long long*>(&xs[s]);
            const unsigned long long old = atomicCAS(p0, XEMPTY, claim);
            if (old == XEMPTY) { xs[s].z = e.y; ++mine; break; }
            if (old == claim) break;
        }
    }
    if (mine) atomicAdd(&n_in[blockIdx.x & 63u], (unsigned long long)mine);
}
int index_hard_slice(Ctx& c, const uint4* d_all, uint64_t n_all, unsigned rank, unsigned world, uint4** d_hard, uint64_t* n_hard) {
    hipStream_t st = c.stream;
    *d_hard = nullptr; *n_hard = 0;
    if (!c.index_cap || !n_all || getenv("W2RAP_NO_EXACT_INDEX") || c.edge_bases > XPOS_MASK) { uint4* p = nullptr; W2_ALLOC(p, uint4, 1); *d_hard = p; return 0; }
    const uint64_t lo = n_all * rank / world, hi = n_all * (rank + 1) / world;
    unsigned long long* d_n = nullptr;
    W2_ALLOC(d_n, unsigned long long, 130);
    uint64_t cap = (hi - lo) / 64 / 8 + 1024;                              // an eighth of the slice's entries hard; more: the exact sizes
    uint4 *stripes = nullptr, *packed = nullptr;
    unsigned long long h_n[64];
    for (int attempt = 0;; ++attempt) {
        W2_ALLOC(stripes, uint4, 64 * cap);
        W2_HIP(hipMemsetAsync(d_n, 0, 130 * 8, st));
        if (hi > lo) LAUNCH(c, "k_index_mark", k_index_hard_list, dim3(grid_for(hi - lo)), dim3(256), 0, lo, hi, d_all, (const uint4*)c.d_index, c.index_cap - 1, d_n, cap, stripes);
        W2_HIP(hipMemcpyAsync(h_n, d_n, sizeof(h_n), hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        uint64_t mx = 0;
        for (unsigned k = 0; k < 64; ++k) mx = std::max<uint64_t>(mx, h_n[k]);
        if (mx <= cap) break;
        if (attempt) { c.err = "pathing index: hard-entry list overflow after resizing"; return W2RAP_E_LIMIT; }
        c.release(stripes);
        cap = mx + 64;
    }
    unsigned long long pre[65]; uint64_t tot = 0, mx = 0;
    for (unsigned k = 0; k < 64; ++k) { pre[k] = tot; tot += h_n[k]; mx = std::max<uint64_t>(mx, h_n[k]); }
    pre[64] = tot;
    W2_ALLOC(packed, uint4, tot + 1);
    W2_HIP(hipMemcpyAsync(d_n + 65, pre, 65 * 8, hipMemcpyHostToDevice, st));
    if (tot) LAUNCH(c, "k_index_hard_pack", k_index_hard_pack, dim3(grid_for(mx), 64), dim3(256), 0, cap, (const unsigned long long*)d_n, (const unsigned long long*)(d_n + 65), (const uint4*)stripes, packed);
    W2_HIP(hipStreamSynchronize(st));
    W2_HIP(hipGetLastError());
    c.release(stripes); c.release(d_n);
    *d_hard = packed; *n_hard = tot;
    return 0;
}
int index_hard_apply(Ctx& c, const uint4* d_hard, uint64_t n_hard) {
    hipStream_t st = c.stream;
    if (c.d_xindex) { c.release(c.d_xindex); c.d_xindex = nullptr; }
    c.xindex_cap = 0; c.xindex_kmers = 0;
    if (!n_hard || !c.index_cap) return 0;
    unsigned long long* d_n = nullptr; uint32_t* d_f = nullptr;
    W2_ALLOC(d_n, unsigned long long, 64); W2_ALLOC(d_f, uint32_t, 4);
    uint64_t cap = 1024;
    while (cap < n_hard * 24) cap <<= 1;
    if (test_hook("W2RAP_TEST_EXACT_SMALL")) cap = 1024;
    for (int attempt = 0;; ++attempt) {
        W2_ALLOC(c.d_xindex, uint4, cap);
        W2_HIP(hipMemsetAsync(c.d_xindex, 0xFF, cap * sizeof(uint4), st));
        W2_HIP(hipMemsetAsync(d_f, 0, 16, st));
        W2_HIP(hipMemsetAsync(d_n, 0, 64 * 8, st));
        LAUNCH(c, "k_exact_insert", k_index_hard_apply, dim3(grid_for(n_hard)), dim3(256), 0, n_hard, d_hard, c.d_index, c.index_cap - 1, c.d_xindex, cap - 1, (const uint8_t*)c.d_edge_bits,
               (const uint64_t*)c.d_edge_off, (const uint32_t*)c.d_edge_nk, d_f, d_n);
        uint32_t h_f[4] = {0, 0, 0, 0}; unsigned long long h_n[64];
        W2_HIP(hipMemcpyAsync(h_f, d_f, 16, hipMemcpyDeviceToHost, st));
        W2_HIP(hipMemcpyAsync(h_n, d_n, sizeof(h_n), hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        uint64_t nin = 0;
        for (unsigned i = 0; i < 64; ++i) nin += h_n[i];
        if (!h_f[0] && 2 * nin <= cap) { c.xindex_cap = cap; c.xindex_kmers = nin; break; }
        if (attempt >= 24) { c.err = "pathing index: the exact table's sizing failed"; return W2RAP_E_LIMIT; }
        c.release(c.d_xindex); c.d_xindex = nullptr;
        cap <<= 2;
        while (cap < 2 * nin) cap <<= 1;
    }
    if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] pathing index: %llu gathered entries of keys with more than %u entries, %llu k-mers around them in the exact table (%llu slots)\n",
                                       (unsigned long long)n_hard, IDX_HARD, (unsigned long long)c.xindex_kmers, (unsigned long long)c.xindex_cap);
    c.release(d_n); c.release(d_f);
    return 0;
}
int index_harden(Ctx& c) {
    hipStream_t st = c.stream;
    if (c.d_xindex) { c.release(c.d_xindex); c.d_xindex = nullptr; }
    c.xindex_cap = 0; c.xindex_kmers = 0;
    if (!c.index_cap || !c.d_index || getenv("W2RAP_NO_EXACT_INDEX")) return 0;
    if (c.edge_bases > XPOS_MASK) return 0;                                // (positions beyond 34 bits: the plain index alone)
    unsigned long long* d_n = nullptr; uint32_t* d_f = nullptr;
    W2_ALLOC(d_n, unsigned long long, 128); W2_ALLOC(d_f, uint32_t, 4);
    W2_HIP(hipMemsetAsync(d_n, 0, 128 * 8, st));
    W2_HIP(hipMemsetAsync(d_f, 0, 16, st));
    unsigned run_max = 1u << 20;
    if (const char* v = getenv("W2RAP_TEST_INDEX_RUN_MAX")) { if (test_hook("W2RAP_TEST_INDEX_RUN_MAX")) run_max = (unsigned)std::max(1, atoi(v)); }
    LAUNCH(c, "k_index_mark", k_index_mark, dim3(grid_for(c.index_cap)), dim3(256), 0, c.index_cap, c.d_index, c.index_cap - 1, d_n, d_f, run_max);
    unsigned long long h_n[128];
    uint32_t h_run[4] = {0, 0, 0, 0};
    W2_HIP(hipMemcpyAsync(h_n, d_n, sizeof(h_n), hipMemcpyDeviceToHost, st));
    W2_HIP(hipMemcpyAsync(h_run, d_f, 16, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    W2_HIP(hipGetLastError());
    if (h_run[1]) { c.release(d_n); c.release(d_f); c.err = "pathing index: a run of occupied slots longer than the marking pass walks (a degenerate key distribution)"; return W2RAP_E_LIMIT; }
    uint64_t hard = 0;
    for (unsigned i = 0; i < 64; ++i) hard += h_n[i];
    if (hard) {
        uint64_t cap = 1024;
        while (cap < hard * 24) cap <<= 1;                                 // ~12 distinct k-mers per marked entry (two sides, overlapping neighbours): load ~0.5 at most
        if (test_hook("W2RAP_TEST_EXACT_SMALL")) cap = 1024;
        for (int attempt = 0;; ++attempt) {
            W2_ALLOC(c.d_xindex, uint4, cap);
            W2_HIP(hipMemsetAsync(c.d_xindex, 0xFF, cap * sizeof(uint4), st));
            W2_HIP(hipMemsetAsync(d_f, 0, 16, st));
            W2_HIP(hipMemsetAsync(d_n + 64, 0, 64 * 8, st));
            LAUNCH(c, "k_exact_insert", k_exact_insert, dim3(grid_for(c.index_cap)), dim3(256), 0, c.index_cap, (const uint4*)c.d_index, c.d_xindex, cap - 1, (const uint8_t*)c.d_edge_bits,
                   (const uint64_t*)c.d_edge_off, (const uint32_t*)c.d_edge_nk, d_f, d_n + 64);
            uint32_t h_f[4] = {0, 0, 0, 0};
            W2_HIP(hipMemcpyAsync(h_f, d_f, 16, hipMemcpyDeviceToHost, st));
            W2_HIP(hipMemcpyAsync(h_n, d_n, sizeof(h_n), hipMemcpyDeviceToHost, st));
            W2_HIP(hipStreamSynchronize(st));
            W2_HIP(hipGetLastError());
            uint64_t nin = 0;
            for (unsigned i = 64; i < 128; ++i) nin += h_n[i];
            if (!h_f[0] && 2 * nin <= cap) { c.xindex_cap = cap; c.xindex_kmers = nin; break; }
            if (attempt >= 24) { c.err = "pathing index: the exact table's sizing failed"; return W2RAP_E_LIMIT; }
            c.release(c.d_xindex); c.d_xindex = nullptr;
            cap <<= 2;                                                     // (a full table stopped the pass early: what it counted is a lower bound)
            while (cap < 2 * nin) cap <<= 1;
        }
    }
    if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] pathing index: %llu entries of keys with more than %u entries, %llu k-mers around them in the exact table (%llu slots)\n",
                                       (unsigned long long)hard, IDX_HARD, (unsigned long long)c.xindex_kmers, (unsigned long long)c.xindex_cap);
    c.release(d_n); c.release(d_f);
    return 0;
}
// (edge, offset) of every solid k-mer, looked up through the index
int index_probe_all(Ctx& c, int32_t* d_edge, uint32_t* d_off) {
    if (c.S) LAUNCH(c, "k_index_probe", k_index_probe, dim3(grid_for(c.S)), dim3(256), 0, c.S, c.d_shi, c.d_slo, edge_index(c), d_edge, d_off);
    W2_HIP(hipGetLastError());
    return 0;
}

// list ranking over N oriented nodes linked by nxt0: afterwards rank_of(own, w, v) = (the chain end v reaches, its distance), cyc = lies on
// a circle, mid = middle bases; nxt / rnk (may be null) receive the ranks as arrays.  shi == nullptr skips the middle-base part.
template <class Id>
static int run_ranking_t(Ctx& c, uint64_t N, const Id* nxt0, Id* nxt, uint32_t* rnk, unsigned long long* w, uint32_t* own, uint8_t* cyc, uint8_t* mid,
                         uint32_t* d_flags, const uint64_t* shi, const uint64_t* slo, bool use_chunks) {
    hipStream_t st = c.stream;
    unsigned long long* d_cnt = nullptr; Id* spl = nullptr;
    const uint64_t S = N / 2;
    // splitters: chain heads and nodes whose predecessor lies in another tile -- normally a few percent of the nodes.  64-bit ids start
    // with room for a quarter of them (the array is 8 B x that at more than 2^31 k-mers); the kernel counts them all, so an overflow
    // is followed by one more pass with the exact size
    uint64_t spl_cap = sizeof(Id) == 4 ? N : N / 4 + 4096;
    if (const char* v = getenv("W2RAP_SPL_CAP")) spl_cap = (uint64_t)atoll(v);              // (tests: force the second pass)
    W2_ALLOC(spl, Id, spl_cap); W2_ALLOC(d_cnt, unsigned long long, 4);
    unsigned long long h_cnt[4] = {0, 0, 0, 0};
    bool chunks = use_chunks && c.nchunks != 0 && !getenv("W2RAP_NO_RANK_CHUNKS");
    for (;;) {
        const uint64_t ntiles = chunks ? c.nchunks : (S + RT - 1) / RT;
        W2_HIP(hipMemsetAsync(d_cnt, 0, 32, st));
        LAUNCH(c, "k_rank_tiles", k_rank_tiles<Id>, dim3((unsigned)std::min<uint64_t>(ntiles ? ntiles : 1, (uint64_t)c.sm_count * 64)), dim3(256), 0,
               S, chunks ? c.nchunks : 0, chunks ? c.d_chunk_start : (const uint64_t*)nullptr, chunks ? c.d_chunk_cnt : (const uint32_t*)nullptr,
               nxt0, w, own, spl, d_cnt, spl_cap);
        W2_HIP(hipMemcpyAsync(h_cnt, d_cnt, 32, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        if (h_cnt[1] != S) {
            if (!chunks) { c.err = "list ranking: tiles do not cover the k-mers"; return W2RAP_E_GRAPH; }
            chunks = false;                          // the chunk list does not cover every k-mer exactly once: plain tiles
            continue;
        }
        if (h_cnt[0] <= spl_cap) break;
        c.release(spl);                              // more splitters than room: their number is known now
        spl_cap = h_cnt[0] + 4096;
        W2_ALLOC(spl, Id, spl_cap);
    }
    const unsigned long long nspl = h_cnt[0];
    c.rank_ends = h_cnt[2];
    if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] list ranking: %llu nodes, %llu listed splitters (%s tiles, %u-bit ids)\n", (unsigned long long)N, nspl, chunks ? "chunk" : "plain", (unsigned)(8 * sizeof(Id)));
    int rounds = 0;
    {   // two levels (above); W2RAP_RANK_HIER=0: the plain jumping alone
        const char* hv = getenv("W2RAP_RANK_HIER");
        if ((hv ? atoi(hv) != 0 : nspl >= (1u << 16)) && nspl) {
            const uint64_t sup_cap = nspl / 8 + 4096;                                      // (a sixteenth of them, by a hash: twice that is room enough; more: the plain jumping)
            Id* sup = nullptr; unsigned long long* d_ns = nullptr; unsigned long long ns = 0;
            W2_ALLOC(sup, Id, sup_cap); W2_ALLOC(d_ns, unsigned long long, 1);
            W2_HIP(hipMemsetAsync(d_ns, 0, 8, st));
            LAUNCH(c, "k_split_supers", k_split_supers<Id>, dim3((unsigned)((nspl + 256 * SUPERS_PER_THREAD - 1) / (256 * SUPERS_PER_THREAD))), dim3(256), 0, (uint64_t)nspl, spl, sup, d_ns, sup_cap);
            W2_HIP(hipMemcpyAsync(&ns, d_ns, 8, hipMemcpyDeviceToHost, st));
            W2_HIP(hipStreamSynchronize(st));
            if (ns && ns <= sup_cap) {
                unsigned long long* w0 = nullptr;
                W2_ALLOC(w0, unsigned long long, ns);
                LAUNCH(c, "k_split_walk", k_split_walk1<Id>, dim3(grid_for(ns)), dim3(256), 0, (uint64_t)ns, (const Id*)sup, w, w0);
                // Two launches of sixteen jumps each, nobody asks whether they arrived: a stretch cut by the step limit keeps its super-splitter
                // from ever arriving (one position in sixty: at scale some always are), so a convergence test ran all its eight rounds -- 0.1 ms
                // of launch, copy and wait each for 3 us of work -- and whatever is not final here the plain jumping below finishes anyway.
                for (int round = 0; round < 2; ++round) {
                    LAUNCH(c, "k_split_super_jump", k_split_jump_super<Id>, dim3(grid_for(ns)), dim3(256), 0, (uint64_t)ns, (const Id*)sup, w, d_flags);
                    ++rounds;
                }
                LAUNCH(c, "k_split_walk", k_split_walk2<Id>, dim3(grid_for(ns)), dim3(256), 0, (uint64_t)ns, (const Id*)sup, w, (const unsigned long long*)w0);
                c.release(w0);                                                             // (parked; the stream orders any reuse behind the walk)
            }
            c.release(sup); c.release(d_ns);
        }
    }
    for (int round = 0; round < 40 && nspl; ++round) {
        W2_HIP(hipMemsetAsync(d_flags, 0, 4, st));
        LAUNCH(c, "k_split_jump", k_split_jump<Id>, dim3(grid_for(nspl)), dim3(256), 0, (uint64_t)nspl, spl, w, d_flags);
        uint32_t changed = 0;
        W2_HIP(hipMemcpyAsync(&changed, d_flags, 4, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        ++rounds;
        if (!changed) break;
    }
    if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] list ranking: %d jump launches\n", rounds);
    c.release(spl); c.release(d_cnt);
    if (mid) W2_HIP(hipMemsetAsync(mid, 0, N, st));
    LAUNCH(c, "k_rank_finish", k_rank_finish<Id>, dim3(grid_for(S)), dim3(256), 0, S, w, own, nxt0, shi, slo, nxt, rnk, cyc, mid, d_flags);
    W2_HIP(hipStreamSynchronize(st));
    return 0;
}
// (Step 3, step3_repath.hip: 32-bit ids, the ranks wanted as arrays, no chunk list unless asked for)
int run_ranking(Ctx& c, uint64_t N, const uint32_t* nxt0, uint32_t* nxt, uint32_t* rnk, unsigned long long* w, uint8_t* cyc, uint8_t* mid,
                uint32_t* d_flags, const uint64_t* shi, const uint64_t* slo, bool use_chunks) {
    uint32_t* own = nullptr;
    W2_ALLOC(own, uint32_t, N);
    const int rc = run_ranking_t<uint32_t>(c, N, nxt0, nxt, rnk, w, own, cyc, mid, d_flags, shi, slo, use_chunks);
    c.release(own);
    return rc;
}

static int graph_error(Ctx& c, uint32_t f) {
    if (f & GE_LOOKUP) { c.err = "neighbour k-mer lookup failed (ForceAssert, BuildReadQGraph.cc:265)"; return W2RAP_E_GRAPH; }
    if (f & GE_OFFSET) { c.err = "unipath longer than 16,777,215 k-mers (ForceAssertLe, ReadPather.h:122)"; return W2RAP_E_GRAPH; }
    if (f & GE_ASSIGN) { c.err = "k-mer left without an edge (BuildReadQGraph.cc:303)"; return W2RAP_E_GRAPH; }
    if (f & GE_HINT_MISS) { c.err = "edge_order_hint: a hinted edge is not a unipath of this graph"; return W2RAP_E_HINT; }
    if (f & GE_HINT_DUP) { c.err = "edge_order_hint: an edge is listed twice"; return W2RAP_E_HINT; }
    if (f & GE_HINT_LEN) { c.err = "edge_order_hint: a hinted edge has the wrong length"; return W2RAP_E_HINT; }
    return 0;
}

// list ranking of the chains of nxt0 with the smooth circles (simpleCircle / canonicalizeCircle :126-180) resolved: a circle is cut in
// front of its minimum k-mer and the ranking repeated.  (with_mid: the middle bases of odd-length chains as seen from their heads)
template <class Id>
static int rank_resolve_t(Ctx& c, uint64_t N, Id* nxt0, unsigned long long* rankw, uint32_t* own, uint8_t* cyc, uint8_t* mid, uint32_t* d_flags,
                          const uint64_t* shi, const uint64_t* slo, bool with_mid, bool* had_circles = nullptr) {
    hipStream_t st = c.stream;
    const uint64_t S = N / 2;
    uint32_t h_flags[4] = {0, 0, 0, 0};
    W2_TRY(run_ranking_t<Id>(c, N, nxt0, (Id*)nullptr, (uint32_t*)nullptr, rankw, own, cyc, mid, d_flags, with_mid ? shi : nullptr, slo, true));
    W2_HIP(hipMemcpyAsync(h_flags, d_flags, 16, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    W2_TRY(graph_error(c, h_flags[1]));
    if (had_circles) *had_circles = h_flags[2] != 0;
    if (h_flags[2]) {                        // smooth circles
        Id *nx, *mn, *nx2, *mn2;
        W2_ALLOC(nx, Id, N); W2_ALLOC(mn, Id, N); W2_ALLOC(nx2, Id, N); W2_ALLOC(mn2, Id, N);
        LAUNCH(c, "k_minjump_init", k_minjump_init<Id>, dim3(grid_for(S)), dim3(256), 0, S, nxt0, cyc, nx, mn);
        for (int round = 0; round < 33; ++round) {
            LAUNCH(c, "k_minjump", k_minjump<Id>, dim3(grid_for(S)), dim3(256), 0, S, shi, slo, nx, mn, nx2, mn2);
            std::swap(nx, nx2); std::swap(mn, mn2);
        }
        LAUNCH(c, "k_cycle_cut", k_cycle_cut<Id>, dim3(grid_for(S)), dim3(256), 0, S, cyc, mn, nxt0);
        W2_HIP(hipStreamSynchronize(st));
        c.release(nx); c.release(mn); c.release(nx2); c.release(mn2);
        W2_HIP(hipMemsetAsync(d_flags, 0, 32, st));
        W2_TRY(run_ranking_t<Id>(c, N, nxt0, (Id*)nullptr, (uint32_t*)nullptr, rankw, own, cyc, mid, d_flags, with_mid ? shi : nullptr, slo, true));
        W2_HIP(hipMemcpyAsync(h_flags, d_flags, 16, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        if (h_flags[2]) { c.err = "failed to close circle (BuildReadQGraph.cc:141)"; return W2RAP_E_GRAPH; }
    }
    return 0;
}
// (the sharded graph phase, step2_shard.hip: 64-bit ids, no middle bases)
int rank_resolve32(Ctx& c, uint64_t N, uint32_t* nxt0, unsigned long long* rankw, uint32_t* own, uint8_t* cyc, uint8_t* mid, uint32_t* d_flags,
                   const uint64_t* shi, const uint64_t* slo, bool* had_circles) {
    return rank_resolve_t<uint32_t>(c, N, nxt0, rankw, own, cyc, mid, d_flags, shi, slo, false, had_circles);
}
int rank_resolve64(Ctx& c, uint64_t N, uint64_t* nxt0, unsigned long long* rankw, uint32_t* own, uint8_t* cyc, uint8_t* mid, uint32_t* d_flags,
                   const uint64_t* shi, const uint64_t* slo, bool* had_circles) {
    return rank_resolve_t<uint64_t>(c, N, nxt0, rankw, own, cyc, mid, d_flags, shi, slo, false, had_circles);
}

// Everything behind the edge sequences: the packed edge stream, read pathing's dictionary substitute (index) and absence filter, and
// a8 -- objects, vertices, adjacency (HBVFromEdges.cc:76-154) -- from c.E, c.d_edge_nk, c.d_edge_off, c.edge_bases, c.d_edge_codes.
// A pure function of the ordered edge list: the sharded graph phase (step2_shard.hip) runs it replicated on every rank.
int graph_finish(Ctx& c) {
    hipStream_t st = c.stream;
    const uint64_t E = c.E;
    uint32_t* d_flags = nullptr;
    W2_ALLOC(d_flags, uint32_t, 8);
    W2_HIP(hipMemsetAsync(d_flags, 0, 32, st));
    uint32_t h_flags[4] = {0, 0, 0, 0};
    if (!c.bits_ready) {
        const uint64_t nby = (c.edge_bases + 3) / 4;
        W2_ALLOC(c.d_edge_bits, uint8_t, nby + 16);
        W2_HIP(hipMemsetAsync(c.d_edge_bits + nby, 0, 16, st));
        if (nby) LAUNCH(c, "k_pack_codes", k_pack_codes, dim3(grid_for(nby)), dim3(256), 0, nby, c.edge_bases, c.d_edge_codes, c.d_edge_bits);
    }
    c.bits_ready = false;
    // with the pathing index the dictionary has done its work (prune, edge hints)
    if (c.use_index && c.d_table) { c.release(c.d_table); c.d_table = nullptr; }
    // ---- read pathing's dictionary when the k-mer dictionary is not at hand: the minimizer-sampled index over the edge stream
    if (c.use_index && !c.index_prebuilt) W2_TRY(build_index(c));
    c.index_prebuilt = false;
    // ---- the 31-mer absence filter of read pathing, on the side stream beside the vertex / adjacency kernels below
    if (c.filter_prebuilt) c.filter_prebuilt = false;                  // (sharded graph phase: gathered from the ranks' slices)
    else {
    if (c.d_filter32) { c.release(c.d_filter32); c.d_filter32 = nullptr; }
    c.f32words = 0;
    if (c.edge_bases >= FMER && c.stream2 && !getenv("W2RAP_NO_FILTER32") && c.edge_bases <= (1ull << 33)) {
        uint64_t fw = 1024;
        while (fw * 4 < c.edge_bases) fw <<= 1;                        // one 64-bit word per 2-4 positions (a run of ~9 shares a word)
        W2_ALLOC(c.d_filter32, unsigned long long, fw);
        c.f32words = fw;
        hipEvent_t ev;
        W2_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        W2_HIP(hipEventRecord(ev, st));
        W2_HIP(hipStreamWaitEvent(c.stream2, ev, 0));
        (void)hipEventDestroy(ev);
        W2_HIP(hipMemsetAsync(c.d_filter32, 0, fw * 8, c.stream2));
        const uint64_t npos = c.edge_bases - (FMER - 1);
        LAUNCH_ON(c, c.stream2, "k_filter32", k_filter32, dim3(grid_for(npos)), dim3(256), 0, npos, c.d_edge_bits, c.d_filter32, (uint32_t)(fw - 1), 0u, (uint32_t)fw);
    }
    }
    // ---- a8: objects
    uint32_t* d_nobj = nullptr; uint64_t* d_ooff = nullptr;
    W2_ALLOC(d_nobj, uint32_t, E); W2_ALLOC(d_ooff, uint64_t, E + 1);
    if (E) LAUNCH(c, "k_edge_nobj", k_edge_nobj, dim3(grid_for(E)), dim3(256), 0, E, c.d_edge_nk, c.d_edge_off, c.d_edge_codes, d_nobj);
    W2_TRY(exclusive_scan_u32_to_u64(c, d_nobj, d_ooff, E));
    W2_HIP(hipMemcpyAsync(&c.NO, d_ooff + E, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipMemcpyAsync(h_flags, d_flags, 16, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    W2_TRY(graph_error(c, h_flags[1]));
    const uint64_t NO = c.NO;
    if (NO >= (1ull << 31)) { c.err = "more than 2^31 edge objects"; return W2RAP_E_LIMIT; }
    W2_ALLOC(c.d_fwdX, int32_t, E); W2_ALLOC(c.d_revX, int32_t, E); W2_ALLOC(c.d_obj_edge, uint32_t, NO);
    W2_ALLOC(c.d_left, int32_t, NO); W2_ALLOC(c.d_right, int32_t, NO);
    if (E) LAUNCH(c, "k_edge_xlat", k_edge_xlat, dim3(grid_for(E)), dim3(256), 0, E, d_nobj, d_ooff, c.d_fwdX, c.d_revX, c.d_obj_edge);
    // ---- ends -> vertices
    const uint64_t NE = 2 * NO;
    uint64_t *ehash, *ehi, *elo, *ktmp, *excl;
    uint32_t *eperm, *eflag;
    W2_ALLOC(ehash, uint64_t, NE); W2_ALLOC(ehi, uint64_t, NE); W2_ALLOC(elo, uint64_t, NE); W2_ALLOC(ktmp, uint64_t, NE);
    W2_ALLOC(excl, uint64_t, NE + 1); W2_ALLOC(eperm, uint32_t, NE); W2_ALLOC(eflag, uint32_t, NE);
    c.NV = 0;
    if (NE) {
        LAUNCH(c, "k_ends", k_ends, dim3(grid_for(NE)), dim3(256), 0, NO, c.d_obj_edge, c.d_edge_off, c.d_edge_nk, c.d_edge_codes, ehash, ehi, elo);
        // ONE sort by the hash of the end's K-1 bases; vertex boundaries where hash or bases change.  Two different ends under one hash
        // (2^-64 per pair; the run may then hold them interleaved): the sort by (hash, bases) of round 1.  The groups are ordered by hash
        // either way: same vertex numbers.
        bool by_bases = test_hook("W2RAP_TEST_ENDS_FULL_SORT");
        uint64_t nflag = 0;
        for (;;) {
            LAUNCH(c, "k_iota", k_iota, dim3(grid_for(NE)), dim3(256), 0, NE, eperm);
            if (by_bases) {
                LAUNCH(c, "k_gather_u64", k_gather_u64, dim3(grid_for(NE)), dim3(256), 0, NE, elo, eperm, ktmp);
                W2_TRY(sort_pairs_u64(c, ktmp, eperm, NE, 0, 58));
                LAUNCH(c, "k_gather_u64", k_gather_u64, dim3(grid_for(NE)), dim3(256), 0, NE, ehi, eperm, ktmp);
                W2_TRY(sort_pairs_u64(c, ktmp, eperm, NE, 0, 60));
            }
            LAUNCH(c, "k_gather_u64", k_gather_u64, dim3(grid_for(NE)), dim3(256), 0, NE, ehash, eperm, ktmp);
            W2_TRY(sort_pairs_u64(c, ktmp, eperm, NE, 0, 64));
            W2_HIP(hipMemsetAsync(d_flags + 3, 0, 4, st));
            LAUNCH(c, "k_end_flags", k_end_flags, dim3(grid_for(NE)), dim3(256), 0, NE, eperm, ehash, ehi, elo, eflag, d_flags + 3);
            W2_TRY(exclusive_scan_u32_to_u64(c, eflag, excl, NE));
            uint32_t shared_hash = 0;
            W2_HIP(hipMemcpyAsync(&nflag, excl + NE, 8, hipMemcpyDeviceToHost, st));
            W2_HIP(hipMemcpyAsync(&shared_hash, d_flags + 3, 4, hipMemcpyDeviceToHost, st));
            W2_HIP(hipStreamSynchronize(st));
            if (shared_hash && !by_bases) { by_bases = true; continue; }
            break;
        }
        c.NV = nflag + 1;
        LAUNCH(c, "k_end_vertices", k_end_vertices, dim3(grid_for(NE)), dim3(256), 0, NE, eperm, eflag, excl, c.d_left, c.d_right);
    }
    // ---- adjacency (digraphE::AddEdge order, DigraphTemplate.h:1829-1839): per vertex sorted by
    //      (other vertex, object id) == stable sort of the objects by (this vertex, other vertex)
    const uint64_t NV = c.NV;
    W2_ALLOC(c.d_from_off, uint64_t, NV + 1); W2_ALLOC(c.d_to_off, uint64_t, NV + 1);
    W2_ALLOC(c.d_from_v, int32_t, NO); W2_ALLOC(c.d_from_e, int32_t, NO);
    W2_ALLOC(c.d_to_v, int32_t, NO); W2_ALLOC(c.d_to_e, int32_t, NO);
    uint32_t* deg = nullptr; uint64_t* akeys = nullptr; uint32_t* avals = nullptr;
    W2_ALLOC(deg, uint32_t, NV); W2_ALLOC(akeys, uint64_t, NO); W2_ALLOC(avals, uint32_t, NO);
    for (int dir = 0; dir < 2; ++dir) {
        const int32_t* a = dir == 0 ? c.d_left : c.d_right;
        const int32_t* b = dir == 0 ? c.d_right : c.d_left;
        W2_HIP(hipMemsetAsync(deg, 0, (NV ? NV : 1) * 4, st));
        if (NO) {
            LAUNCH(c, "k_adj_keys", k_adj_keys, dim3(grid_for(NO)), dim3(256), 0, NO, a, b, akeys, avals, deg);
            W2_TRY(sort_pairs_u64(c, akeys, avals, NO, 0, 64));
            LAUNCH(c, "k_adj_out", k_adj_out, dim3(grid_for(NO)), dim3(256), 0, NO, avals, b, dir == 0 ? c.d_from_v : c.d_to_v,
                               dir == 0 ? c.d_from_e : c.d_to_e);
        }
        W2_TRY(exclusive_scan_u32_to_u64(c, deg, dir == 0 ? c.d_from_off : c.d_to_off, NV));
    }
    if (c.d_otab) { c.release(c.d_otab); c.d_otab = nullptr; }
    W2_ALLOC(c.d_otab, ObjRec, NO);
    if (NO) LAUNCH(c, "k_obj_table", k_obj_table, dim3(grid_for(NO)), dim3(256), 0, NO, c.d_obj_edge, c.d_edge_nk, c.d_edge_off, c.d_edge_codes,
                   c.d_right, c.d_from_off, c.d_from_e, c.d_otab);
    W2_HIP(hipStreamSynchronize(st));
    if (c.stream2) W2_HIP(hipStreamSynchronize(c.stream2));
    W2_HIP(hipGetLastError());
    for (void* p : {(void*)d_nobj, (void*)d_ooff, (void*)ehash, (void*)ehi, (void*)elo, (void*)ktmp, (void*)excl,
                    (void*)eperm, (void*)eflag, (void*)deg, (void*)akeys, (void*)avals, (void*)d_flags})
        c.release(p);
    c.graphed = true;
    return 0;
}

// Memory: the phase's large arrays (over the N = 2S oriented nodes) live only as long as they are needed -- the links overwrite
// k_prune's neighbour array, no per-node end / rank arrays exist (rank_of), the link array and the flags go before head_edge and the
// 32-B k-mer records come -- so that S = 2.5 G solid k-mers (BASELINE configs[2] replicated) stay inside 288 GB.
template <class Id>
static int phase_graph_t(Ctx& c, const w2rap_edge_hint* hint) {
    constexpr Id NONE = NodeId<Id>::NONE; (void)NONE;
    hipStream_t st = c.stream;
    const uint64_t S = c.S, N = 2 * S;
    const uint64_t mask = c.tcap - 1;
    uint32_t* d_flags = nullptr;                 // [0] changed  [1] error bits  [2] has cycles
    W2_ALLOC(d_flags, uint32_t, 8);
    W2_HIP(hipMemsetAsync(d_flags, 0, 32, st));
    Id* nxt0 = reinterpret_cast<Id*>(c.d_nbr);   // k_links works in place
    unsigned long long* rankw = nullptr; uint32_t* own = nullptr;
    W2_ALLOC(rankw, unsigned long long, N); W2_ALLOC(own, uint32_t, N);
    uint8_t *cyc = nullptr, *mid = nullptr, *is_head = nullptr;
    W2_ALLOC(cyc, uint8_t, N); W2_ALLOC(mid, uint8_t, N);
    uint32_t h_flags[4] = {0, 0, 0, 0};
    if (S) {
        LAUNCH(c, "k_links", k_links<Id>, dim3(grid_for(S)), dim3(256), 0, S, c.d_shi, c.d_slo, c.d_sctx, nxt0);
        W2_TRY(rank_resolve_t<Id>(c, N, nxt0, rankw, own, cyc, mid, d_flags, c.d_shi, c.d_slo, true));
    }
    c.release(cyc); cyc = nullptr;
    // ---- heads in ONE pass: there are as many heads as chain ends (counted by the ranking), which bounds the canonical ones
    unsigned long long* d_nheads = nullptr;
    W2_ALLOC(d_nheads, unsigned long long, 1);
    W2_HIP(hipMemsetAsync(d_nheads, 0, 8, st));
    const uint64_t head_cap = S ? c.rank_ends + 1 : 1;
    Id *head_v, *edge_head; uint32_t* perm;
    uint64_t *key_hi, *key_lo, *key_tmp;
    W2_ALLOC(is_head, uint8_t, N);
    W2_ALLOC(head_v, Id, head_cap); W2_ALLOC(key_hi, uint64_t, head_cap); W2_ALLOC(key_lo, uint64_t, head_cap);
    if (S) LAUNCH(c, "k_heads", k_heads<Id>, dim3((unsigned)((S + 256 * HEADS_PER - 1) / (256 * HEADS_PER))), dim3(256), 0, S, c.d_shi, c.d_slo, nxt0, own, rankw, mid, is_head,
                              head_v, key_hi, key_lo, d_nheads, head_cap, d_flags, hint == nullptr);
    W2_HIP(hipGetLastError());
    unsigned long long E = 0;
    W2_HIP(hipMemcpyAsync(&E, d_nheads, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipMemcpyAsync(h_flags, d_flags, 16, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    W2_TRY(graph_error(c, h_flags[1]));
    if (E > head_cap) { c.err = "more canonical heads than chain ends"; return W2RAP_E_GRAPH; }
    if (E >= (1ull << 31)) { c.err = "more than 2^31 unipaths (edge ids are int, paths/long/ReadPath.h)"; return W2RAP_E_LIMIT; }
    c.E = E;
    // the links and the middle bases have done their work
    c.release(c.d_nbr); c.d_nbr = nullptr; nxt0 = nullptr;
    c.release(mid); mid = nullptr;
    W2_ALLOC(perm, uint32_t, E); W2_ALLOC(edge_head, Id, E);
    W2_ALLOC(key_tmp, uint64_t, E);
    W2_ALLOC(c.d_edge_nk, uint32_t, E);
    if (hint) {
        if (hint->n_edges != E) {
            c.err = "edge_order_hint has " + std::to_string(hint->n_edges) + " edges, the graph has " + std::to_string(E);
            return W2RAP_E_HINT;
        }
        std::vector<uint64_t> hh(E), hl(E);
        for (uint64_t e = 0; e < E; ++e) {
            if (hint->len[e] < K) { c.err = "edge_order_hint: edge shorter than K"; return W2RAP_E_HINT; }
            const uint8_t* p = hint->packed + hint->byte_off[e];
            uint64_t hi = 0, lo = 0;
            for (unsigned t = 0; t < 30; ++t) hi = (hi << 2) | ((p[t >> 2] >> (2 * (t & 3))) & 3);
            for (unsigned t = 30; t < 60; ++t) lo = (lo << 2) | ((p[t >> 2] >> (2 * (t & 3))) & 3);
            hh[e] = hi; hl[e] = lo;
        }
        uint32_t* d_hlen = nullptr;
        W2_ALLOC(d_hlen, uint32_t, E);
        W2_HIP(hipMemcpyAsync(key_hi, hh.data(), E * 8, hipMemcpyHostToDevice, st));
        W2_HIP(hipMemcpyAsync(key_lo, hl.data(), E * 8, hipMemcpyHostToDevice, st));
        W2_HIP(hipMemcpyAsync(d_hlen, hint->len, E * 4, hipMemcpyHostToDevice, st));
        if (E) LAUNCH(c, "k_edge_from_hint", k_edge_from_hint<Id>, dim3(grid_for(E)), dim3(256), 0, E, key_hi, key_lo, d_hlen, c.d_table, mask, c.d_shi, c.d_slo, is_head,
                                  own, rankw, edge_head, c.d_edge_nk, d_flags);
        W2_HIP(hipMemcpyAsync(h_flags, d_flags, 16, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        W2_TRY(graph_error(c, h_flags[1]));
        c.release(d_hlen);
    } else {
        if (E) {
            // ONE sort by the first 30 bases, the runs of equal words ordered by the other 30 in place; a run too long for that: both sorts
            unsigned max_run = 64;
            if (test_hook("W2RAP_TEST_TIE_RUN")) max_run = (unsigned)atoi(getenv("W2RAP_TEST_TIE_RUN"));
            LAUNCH(c, "k_iota", k_iota, dim3(grid_for(E)), dim3(256), 0, E, perm);
            W2_HIP(hipMemcpyAsync(key_tmp, key_hi, E * 8, hipMemcpyDeviceToDevice, st));
            W2_TRY(sort_pairs_u64(c, key_tmp, perm, E, 0, 60));
            W2_HIP(hipMemsetAsync(d_flags + 3, 0, 4, st));
            LAUNCH(c, "k_tie_sort_lo", k_tie_sort_lo, dim3(grid_for(E)), dim3(256), 0, E, key_tmp, key_lo, perm, max_run, d_flags + 3);
            uint32_t long_run = 0;
            W2_HIP(hipMemcpyAsync(&long_run, d_flags + 3, 4, hipMemcpyDeviceToHost, st));
            W2_HIP(hipStreamSynchronize(st));
            if (long_run) {
                LAUNCH(c, "k_iota", k_iota, dim3(grid_for(E)), dim3(256), 0, E, perm);
                W2_TRY(sort_pairs_u64(c, key_lo, perm, E, 0, 60));
                LAUNCH(c, "k_gather_u64", k_gather_u64, dim3(grid_for(E)), dim3(256), 0, E, key_hi, perm, key_tmp);
                W2_TRY(sort_pairs_u64(c, key_tmp, perm, E, 0, 60));
            }
            LAUNCH(c, "k_edge_from_sorted", k_edge_from_sorted<Id>, dim3(grid_for(E)), dim3(256), 0, E, perm, head_v, own, rankw, edge_head, c.d_edge_nk);
        }
    }
    W2_HIP(hipStreamSynchronize(st));
    c.release(is_head); is_head = nullptr;
    // ---- edge sequences
    uint32_t* d_elen = nullptr;
    W2_ALLOC(d_elen, uint32_t, E);
    W2_ALLOC(c.d_edge_off, uint64_t, E + 1);
    if (E) LAUNCH(c, "k_edge_len", k_edge_len, dim3(grid_for(E)), dim3(256), 0, E, c.d_edge_nk, d_elen);
    W2_TRY(exclusive_scan_u32_to_u64(c, d_elen, c.d_edge_off, E));
    W2_HIP(hipMemcpyAsync(&c.edge_bases, c.d_edge_off + E, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    W2_ALLOC(c.d_edge_codes, uint8_t, c.edge_bases + 64);
    {   // W2RAP_PATH_INDEX=1: read pathing through the index on one GPU too (an A/B switch: profiles/r05_index_ab.txt)
        const char* iv = getenv("W2RAP_PATH_INDEX");
        c.use_index = iv && atoi(iv) != 0;
    }
    if (c.d_srec) { c.release(c.d_srec); c.d_srec = nullptr; }
    if (!c.use_index) W2_ALLOC(c.d_srec, KRec, S);
    // (Tried in round 6: the bases first and the 8 GB of records as a second pass from graph_finish, beside the 31-mer filter -- which needs
    // only the bases and is 3.9 ms of nothing else running in front of read pathing.  Beside each other the filter takes 8.3 instead of 4.1 ms
    // and the two passes 7.9 instead of 5.3: graph phase +2 ms.  One pass it stays.)
    if (S) LAUNCH(c, "k_assign", k_assign<Id>, dim3(grid_for(S)), dim3(256), 0, S, c.d_shi, c.d_slo, own, rankw,
                              c.d_edge_off, c.d_srec, c.d_edge_codes, d_flags);
    W2_HIP(hipStreamSynchronize(st));
    c.release(rankw); c.release(own); rankw = nullptr; own = nullptr;
    for (void* p : {(void*)d_nheads, (void*)head_v, (void*)perm, (void*)edge_head, (void*)key_hi, (void*)key_lo, (void*)key_tmp, (void*)d_elen, (void*)d_flags})
        c.release(p);
    return graph_finish(c);
}

int phase_graph(Ctx& c, const w2rap_edge_hint* hint) {
    if (!c.counted) { c.err = "build_graph called before count_kmers"; return W2RAP_E_STATE; }
    if (!c.d_nbr && c.S) { c.err = "build_graph: the neighbour links of this count have been consumed; call count_kmers again"; return W2RAP_E_STATE; }
    c.graphed = false;
    return c.wide_ids ? phase_graph_t<uint64_t>(c, hint) : phase_graph_t<uint32_t>(c, hint);
}

}  // namespace w2
