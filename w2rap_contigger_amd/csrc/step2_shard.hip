// step2_shard.hip -- SURVEY.md 8(e), row e-3: the dictionary, the adjacency prune and the unipath phase SHARDED by bucket owner.
//
//   new BRQ_Dict(kmers.size())            BuildReadQGraph.cc:1092      one dictionary without a ceiling -- here every owner keeps ITS solid k-mers
//   KmerDict::recomputeAdjacencies        kmers/ReadPather.h:317-346   membership of every neighbour k-mer
//   buildEdges                            BuildReadQGraph.cc:99-339    unipaths = chains of the k-mer graph, circles cut at their minimum k-mer
//
// Rounds 1-4 all-gathered the solid k-mers and rebuilt the JOB's dictionary and graph on every GPU.  Here a k-mer lives on ONE rank -- the
// owner of the bucket of its canonical minimizer, where the counting left it -- and keeps the job-wide number  base[rank] + local index
// (owners in rank order: the numbering of the replicated path), an oriented node is 2 * number + (reverse-complemented).  What crosses
// ranks, each a batched query / response pair of all-to-alls keyed by the owner of the k-mer asked about:
//   A  neighbour k-mers that are not in the asker's own dictionary and belong to another owner: solid? -> its number;
//   B  the pruned context of a single surviving neighbour on another rank (the link condition of buildEdges :192-214 is symmetric);
//   C  the segment number of the chain head a local chain continues into.
// Unipaths are ranked on TWO levels.  Level 1: every rank ranks the chains of ITS nodes with the links to other ranks taken as chain ends
// (the list ranking of step2_graph.hip, chunk-local tiles and all).  A maximal local chain is a SEGMENT; consecutive k-mers share their
// minimizer -- hence their owner -- with probability ~45/47, so the segments are ~4 % of the k-mers.  Level 2: the segment records
// (length, next segment, head k-mer: 32 B) are all-gathered and the segment chains are ranked by pointer jumping, REPLICATED on every
// rank: the only replicated per-k-mer-proportional work of the phase, at 1/25 of the k-mers.  Canonical heads, the unipath order (sorted
// or replayed), offsets and the edge table follow from the segment arrays on every rank alike; every rank then writes the bases of ITS
// k-mers into a zeroed edge stream, the streams are summed (all-reduce; the bits are disjoint), and vertices, adjacency, the pathing index
// and the absence filter are built from the stream by graph_finish() -- a pure function of the ordered edge list, E-sized.  Read pathing
// asks the minimizer-sampled index over the replicated edge sequences (common.h EdgeIndex) instead of a dictionary.
//
// The phase is a state machine: w2rap_step2_shard_next() computes up to the next exchange and describes it (w2rap_xchg); the host layer --
// dist.py over RCCL, or the threads of w2rap_step2_run over peer copies -- performs it and calls again.  Nothing here knows how bytes travel.
#include <algorithm>
#include <cstring>
#include <chrono>
#include "ctx.h"

namespace w2 {

int rank_resolve64(Ctx& c, uint64_t N, uint64_t* nxt0, unsigned long long* rankw, uint32_t* own, uint8_t* cyc, uint8_t* mid, uint32_t* d_flags,
                   const uint64_t* shi, const uint64_t* slo, bool* had_circles);      // step2_graph.hip
int rank_resolve32(Ctx& c, uint64_t N, uint32_t* nxt0, unsigned long long* rankw, uint32_t* own, uint8_t* cyc, uint8_t* mid, uint32_t* d_flags,
                   const uint64_t* shi, const uint64_t* slo, bool* had_circles);
int graph_finish(Ctx& c);                                                             // step2_graph.hip
int table_build_plain(Ctx& c);                                                        // step2_count.hip: table over c.d_shi[0..S), all at once
int prune_local_chunks64(Ctx& c, uint8_t* sctx, uint64_t* nbr, uint8_t* unres);       // step2_count.hip: k_prune_local over the chunk list
int prune_local_chunks32(Ctx& c, uint8_t* sctx, uint32_t* nbr, uint8_t* unres);

typedef uint64_t Id;
constexpr Id NONE = NodeId<Id>::NONE, PAL = NodeId<Id>::PAL;
constexpr uint64_t ABSENT = ~0ull;

struct ShardMap { uint64_t base[65]; uint32_t world, me, NB, per_pass, nbl, test_cut, test_virtual; };           // owner of bucket b = (b % per_pass) / nbl
// Is the link between the job-wide nodes a and b one that stays inside its rank's local chains?  Always, unless the test hook
// W2RAP_TEST_SHARD_CUT = n is set: then one local link in n (by a hash of the two k-mers' numbers: the same answer for the link and for its
// mirror) is handed to the cross-rank machinery -- segment queries, level 2 -- as if its far end lived on another rank.  Results do not change;
// a single GPU can then measure and test level 2 at the segment counts of a many-rank job.
__device__ inline bool link_stays_local(const ShardMap& M, uint64_t a, uint64_t b) {
    if (!M.test_cut) return true;
    const uint64_t x = a >> 1, y = b >> 1, lo = x < y ? x : y, hi = x < y ? y : x;
    return (((lo * 0x9E3779B97F4A7C15ull) ^ (hi * 0xC2B2AE3D27D4EB4Full)) >> 17) % M.test_cut != 0;
}
__device__ inline unsigned owner_of_kmer(const ShardMap& M, Kmer canon) {
    const MinHit m = minimizer_of(canon);
    const uint32_t b = bucket_of(m.key, M.NB);
    const uint32_t o = (b % M.per_pass) / M.nbl;
    return o < M.world ? o : M.world - 1;
}
__device__ inline unsigned rank_of_index(const ShardMap& M, uint64_t gidx) {
    unsigned r = 0;
    while (r + 1 < M.world && gidx >= M.base[r + 1]) ++r;
    return r;
}
static inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }
// One place in a list for every lane that calls (a lane appends ONE item): the calling lanes of the wavefront reserve their places with ONE
// atomic.  (A per-lane atomicAdd on one counter is serialised at the L2's atomic unit -- ~24 ns each: 23 M appends took 550 ms, found with
// W2RAP_TEST_SHARD_CUT, which is what gives a single GPU the list sizes of a many-rank job.)
__device__ inline unsigned long long wave_slot(unsigned long long* counter) {
    const unsigned long long m = __ballot(1);
    const unsigned lane = threadIdx.x & 63u, leader = (unsigned)__builtin_ctzll(m);
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned long long)__builtin_popcountll(m));
    base = __shfl(base, (int)leader);
    return base + (unsigned long long)__builtin_popcountll(m & ((1ull << lane) - 1ull));
}

// ---------------------------------------------------------------------------------------------- lists built by many blocks
// Atomics on ONE address are serialised at their L2 channel: ~24 ns each on this part, whatever the wavefront does with the result
// (measured with W2RAP_TEST_SHARD_CUT: 23 M single appends 550 ms; one atomic per wavefront, 360 k of them: 8.6 ms).  The lists of a
// many-rank job have 10^7 - 10^8 entries, so a list here is STRIPED: up to NSTRIPE sub-lists, each with its own counter in its own 128-byte
// line and its own region of `cap` entries; a block appends to sub-list blockIdx.x % k with one atomic per wavefront and append site.
// Consumers walk the sub-lists (grid.y = sub-list), or the entries are numbered densely from the prefix sums of the counts.
constexpr unsigned NSTRIPE = 512, STRIPE_PAD = 16;
struct Stripes { unsigned long long* cnt; uint64_t cap; uint32_t k; };
__device__ inline uint64_t stripe_slot(const Stripes& L) {                // ~0: the sub-list is full (its counter keeps counting: the host sees by how much)
    const unsigned k = blockIdx.x % L.k;
    const unsigned long long i = wave_slot(&L.cnt[k * STRIPE_PAD]);
    return i < L.cap ? (uint64_t)k * L.cap + i : ~0ull;
}
__device__ inline uint64_t stripe_count(const Stripes& L, unsigned k) { const uint64_t n = L.cnt[k * STRIPE_PAD]; return n < L.cap ? n : L.cap; }
// striped u64 entries -> dense, sub-list after sub-list (pre: exclusive prefix sums of the counts)
__global__ void __launch_bounds__(256) k_stripes_compact(Stripes L, const uint64_t* __restrict__ pre, const uint64_t* __restrict__ in, uint64_t* __restrict__ out) {
    const unsigned k = blockIdx.y;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < stripe_count(L, k)) out[pre[k] + i] = in[(uint64_t)k * L.cap + i];
}

// ---------------------------------------------------------------------------------------------- generic routing of tagged items
// tag bits 63:58 = destination rank.  A block takes RT_ITEMS entries of one sub-list: k_route_count leaves its count per destination at
// bh[dest][block]; ONE exclusive scan over that array is every (destination, block)'s place in the send buffer -- no global atomics --;
// k_route_scatter ranks the entries inside the block (LDS) and writes them (the order inside a destination's part is arbitrary: the responses
// come back in the order the queries went).
constexpr unsigned RT_ITEMS = 2048;
__global__ void __launch_bounds__(256) k_route_count(Stripes L, const uint64_t* __restrict__ tag, uint32_t world, uint64_t NB, uint32_t* __restrict__ bh) {
    __shared__ unsigned s_h[64];
    if (threadIdx.x < 64) s_h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned k = blockIdx.y;
    const uint64_t nk = stripe_count(L, k), b = (uint64_t)k * gridDim.x + blockIdx.x;
    for (unsigned j = 0; j < RT_ITEMS / 256; ++j) {
        const uint64_t i = (uint64_t)blockIdx.x * RT_ITEMS + j * 256 + threadIdx.x;
        if (i < nk) atomicAdd(&s_h[tag[(uint64_t)k * L.cap + i] >> 58], 1u);
    }
    __syncthreads();
    if (threadIdx.x < world) bh[(uint64_t)threadIdx.x * NB + b] = s_h[threadIdx.x];
}
__global__ void __launch_bounds__(256) k_route_scatter(Stripes L, const uint64_t* __restrict__ tag, const uint64_t* __restrict__ p0, const uint64_t* __restrict__ p1,
                                                        uint64_t NB, const uint64_t* __restrict__ boff, uint64_t* __restrict__ out_tag,
                                                        uint64_t* __restrict__ out /* 1 or 2 words per item */) {
    __shared__ unsigned s_c[64];
    if (threadIdx.x < 64) s_c[threadIdx.x] = 0;
    __syncthreads();
    const unsigned k = blockIdx.y;
    const uint64_t nk = stripe_count(L, k), b = (uint64_t)k * gridDim.x + blockIdx.x;
    for (unsigned j = 0; j < RT_ITEMS / 256; ++j) {
        const uint64_t i = (uint64_t)blockIdx.x * RT_ITEMS + j * 256 + threadIdx.x;
        if (i >= nk) continue;
        const uint64_t src = (uint64_t)k * L.cap + i, t = tag[src];
        const unsigned dest = (unsigned)(t >> 58);
        const uint64_t at = boff[(uint64_t)dest * NB + b] + atomicAdd(&s_c[dest], 1u);
        out_tag[at] = t;
        if (p1) { out[2 * at] = p0[src]; out[2 * at + 1] = p1[src]; } else out[at] = p0[src];
    }
}

// ---------------------------------------------------------------------------------------------- A: the adjacency prune across owners
// The global step of the prune on an owner's k-mers (k_prune of step2_count.hip) with three outcomes per open context bit: the neighbour
// is in THIS rank's dictionary (settled, its node remembered); it is not and its bucket is this rank's (settled: not solid); its bucket is
// another rank's: a query (tag: dest | pal | rc | bit | k-mer) is appended and the bit stays set until the answer comes.
// (LId: the type of the rank's LOCAL node numbers -- 32-bit words while it owns fewer than 2^31 k-mers, as on one GPU; job-wide numbers are 64-bit)
template <class LId>
__global__ void __launch_bounds__(256) k_prune_shard(uint64_t S, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo, const uint32_t* __restrict__ scc,
                                                      const Slot* __restrict__ table, uint64_t mask, const uint8_t* __restrict__ sctx_in,
                                                      const LId* __restrict__ nbr_in, const uint8_t* __restrict__ unres, ShardMap M,
                                                      uint8_t* __restrict__ sctx, Id* __restrict__ nbrG,
                                                      Stripes Q, uint64_t* __restrict__ q_tag, uint64_t* __restrict__ q_hi, uint64_t* __restrict__ q_lo) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const uint64_t base2 = 2 * M.base[M.me];
    unsigned c, todo;
    Id ns = NONE, np = NONE;
    if (unres) {
        todo = unres[i];
        if (todo == 0xFFu && sctx_in[i] == 0xFFu) { c = (scc[i] >> 8) & 0xFF; todo = c; }        // never visited (oversized chunk)
        else {
            c = sctx_in[i];
            const LId ls = nbr_in[2 * i], lp = nbr_in[2 * i + 1];
            ns = ls < NodeId<LId>::PAL ? base2 + ls : ls == NodeId<LId>::PAL ? PAL : NONE;
            np = lp < NodeId<LId>::PAL ? base2 + lp : lp == NodeId<LId>::PAL ? PAL : NONE;
        }
    } else { c = (scc[i] >> 8) & 0xFF; todo = c; }
    const Kmer k{shi[i], slo[i]}, rk = kmer_rc(k);
    for (unsigned rest = todo & c; rest;) {
        const unsigned t = (unsigned)__builtin_ctz(rest);
        rest &= rest - 1;
        const unsigned b = t & 3;
        const Kmer fw = t < 4 ? kmer_succ(k, b) : kmer_pred(k, b);
        const Kmer rv = t < 4 ? kmer_pred(rk, 3u - b) : kmer_succ(rk, 3u - b);
        const bool r = kmer_lt(rv, fw);
        const Kmer nk = r ? rv : fw;
        const bool pal = kmer_eq(rv, fw);
        unsigned o = M.me;
        bool ask = false;                                             // (test hook W2RAP_TEST_SHARD_VIRTUAL = V: the neighbours whose bucket would belong to
        if (M.test_virtual) {                                         //  another of V owners are ASKED for, as on V ranks, although this rank could look)
            const uint32_t b = bucket_of(minimizer_of(nk).key, M.NB) % M.per_pass;
            ask = b / (M.per_pass / M.test_virtual) != 0;
            const uint32_t ro = b / M.nbl;
            o = ro < M.world ? ro : M.world - 1;
        }
        if (!ask) {
            const int64_t s = table_find(table, mask, shi, slo, nk);
            if (s >= 0) {
                const Id id = pal ? PAL : (Id)(base2 + 2 * (uint64_t)s + (r ? 1u : 0u));
                if (t < 4) ns = id; else np = id;
                continue;
            }
            if (!M.test_virtual) o = M.world > 1 ? owner_of_kmer(M, nk) : M.me;
            if (o == M.me) { c &= ~(1u << t); continue; }
        }
        const uint64_t at = stripe_slot(Q);
        if (at != ~0ull) {
            q_tag[at] = ((uint64_t)o << 58) | ((uint64_t)pal << 57) | ((uint64_t)r << 56) | ((uint64_t)t << 53) | i;
            q_hi[at] = nk.hi; q_lo[at] = nk.lo;
        }
    }
    sctx[i] = (uint8_t)c;
    nbrG[2 * i] = ns; nbrG[2 * i + 1] = np;                         // raw: "exactly one survives" is decided when every answer is in
}
// owner side: is the k-mer solid here? -> its job-wide number
__global__ void __launch_bounds__(256) k_answer_member(uint64_t n, const uint64_t* __restrict__ q /* hi, lo pairs */, const Slot* __restrict__ table, uint64_t mask,
                                                        const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo, uint64_t base_me,
                                                        uint64_t* __restrict__ resp) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t s = table_find(table, mask, shi, slo, Kmer{q[2 * j], q[2 * j + 1]});
    resp[j] = s < 0 ? ABSENT : base_me + (uint64_t)s;
}
__global__ void __launch_bounds__(256) k_apply_member(uint64_t n, const uint64_t* __restrict__ tag, const uint64_t* __restrict__ resp,
                                                       uint8_t* __restrict__ sctx, Id* __restrict__ nbrG) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t tg = tag[j], i = tg & ((1ull << 53) - 1);
    const unsigned t = (unsigned)(tg >> 53) & 7u;
    const uint64_t r = resp[j];
    if (r == ABSENT) {
        // clear bit t of byte i: the bits of one k-mer may be answered by several threads
        uint32_t* wd = reinterpret_cast<uint32_t*>(sctx + (i & ~3ull));
        atomicAnd(wd, ~((1u << t) << (8 * (unsigned)(i & 3))));
    } else {
        const Id id = ((tg >> 57) & 1) ? PAL : (Id)(2 * r + ((tg >> 56) & 1));
        nbrG[2 * i + (t >= 4)] = id;
    }
}
// the single surviving successor / predecessor (or NONE), and the B queries: the pruned context of such a neighbour on another rank
__global__ void __launch_bounds__(256) k_prune_final(uint64_t S, const uint8_t* __restrict__ sctx, Id* __restrict__ nbrG, ShardMap M,
                                                      Stripes Q, uint64_t* __restrict__ q_tag, uint64_t* __restrict__ q_p0) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const unsigned c = sctx[i];
    const uint64_t lo = M.base[M.me], hi = M.base[M.me + 1];
#pragma unroll
    for (unsigned d = 0; d < 2; ++d) {
        Id g = nbrG[2 * i + d];
        if (popc4(d ? c >> 4 : c) != 1) g = NONE;
        nbrG[2 * i + d] = g;
        if (g < PAL && ((g >> 1) < lo || (g >> 1) >= hi)) {
            const uint64_t at = stripe_slot(Q);
            if (at != ~0ull) { q_tag[at] = ((uint64_t)rank_of_index(M, g >> 1) << 58) | (2 * i + d); q_p0[at] = g >> 1; }
        }
    }
}
__global__ void __launch_bounds__(256) k_answer_ctx(uint64_t n, const uint64_t* __restrict__ q, const uint8_t* __restrict__ sctx, uint64_t base_me, uint64_t S,
                                                     uint64_t* __restrict__ resp) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t x = q[j] - base_me;
    resp[j] = x < S ? sctx[x] : 0u;
}
__global__ void __launch_bounds__(256) k_apply_ctx(uint64_t n, const uint64_t* __restrict__ tag, const uint64_t* __restrict__ resp, uint8_t* __restrict__ nctx) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    nctx[tag[j] & ((1ull << 58) - 1)] = (uint8_t)resp[j];
}
// chain links (k_links of step2_graph.hip, :192-214) in job-wide node numbers; nxtL: the same with the links to other ranks as chain ends
template <class LId>
__global__ void __launch_bounds__(256) k_links_shard(uint64_t S, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo, const uint8_t* __restrict__ sctx,
                                                      const uint8_t* __restrict__ nctx, ShardMap M, Id* __restrict__ nbr_nxtG, LId* __restrict__ nxtL) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const uint64_t lo = M.base[M.me], hi = M.base[M.me + 1];
    const Kmer k{shi[i], slo[i]};
    Id n0 = NONE, n1 = NONE;
    if (!kmer_is_pal(k)) {
        const Id s = nbr_nxtG[2 * i], p = nbr_nxtG[2 * i + 1];
        if (s < PAL) {
            const bool loc = (s >> 1) >= lo && (s >> 1) < hi;
            unsigned cj = loc ? sctx[(s >> 1) - lo] : nctx[2 * i]; if (s & 1) cj = brev8(cj);
            if (popc4(cj >> 4) == 1) n0 = s;
        }
        if (p < PAL) {
            const bool loc = (p >> 1) >= lo && (p >> 1) < hi;
            unsigned cj = loc ? sctx[(p >> 1) - lo] : nctx[2 * i + 1]; if (p & 1) cj = brev8(cj);
            if (popc4(cj & 15) == 1) n1 = p ^ (Id)1;
        }
    }
    nbr_nxtG[2 * i] = n0; nbr_nxtG[2 * i + 1] = n1;
    auto local = [&](Id g, unsigned d) -> LId {
        return g != NONE && (g >> 1) >= lo && (g >> 1) < hi && link_stays_local(M, 2 * (lo + i) + d, g) ? (LId)(g - 2 * lo) : NodeId<LId>::NONE;
    };
    nxtL[2 * i] = local(n0, 0); nxtL[2 * i + 1] = local(n1, 1);
}
template <class LId>
__global__ void __launch_bounds__(256) k_local_links(uint64_t S, const Id* __restrict__ nxtG, ShardMap M, LId* __restrict__ nxtL) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= 2 * S) return;
    const uint64_t lo = M.base[M.me], hi = M.base[M.me + 1];
    const Id g = nxtG[v];
    nxtL[v] = g != NONE && (g >> 1) >= lo && (g >> 1) < hi && link_stays_local(M, 2 * lo + v, g) ? (LId)(g - 2 * lo) : NodeId<LId>::NONE;
}

// ---------------------------------------------------------------------------------------------- segments (level 1 -> level 2)
// A local chain has two heads, v and the flip of its other end: the smaller one numbers the pair (2c, 2c + 1).  The number of the segment
// whose head is the flip of a local chain end t rides in the distance field of t's own rank word (as the unipath number does on one GPU).
template <class LId>
__global__ void __launch_bounds__(256) k_seg_list(uint64_t S, const LId* __restrict__ nxtL, const uint32_t* __restrict__ own, const unsigned long long* __restrict__ w,
                                                   Stripes L, uint64_t* __restrict__ heads) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= 2 * S || nxtL[v ^ 1] != NodeId<LId>::NONE) return;         // not a local head
    LId t; uint32_t d;
    rank_of<LId>(own, w, (LId)v, t, d);
    const uint64_t u = (uint64_t)t ^ 1ull;                                // the other head
    if (u < v) return;                                                    // (u != v: a chain never runs from a node to its own flip)
    const uint64_t at = stripe_slot(L);
    if (at != ~0ull) heads[at] = v;
}
// the listed chains numbered sub-list after sub-list (pre: exclusive prefix sums of the sub-lists' counts)
template <class LId>
__global__ void __launch_bounds__(256) k_seg_number(Stripes L, const uint64_t* __restrict__ pre, const uint64_t* __restrict__ heads, const uint32_t* __restrict__ own,
                                                     unsigned long long* __restrict__ w, Id* __restrict__ seg_head, uint32_t* __restrict__ seg_len) {
    const unsigned k = blockIdx.y;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= stripe_count(L, k)) return;
    const uint64_t v = heads[(uint64_t)k * L.cap + i], ch = pre[k] + i;
    LId t; uint32_t d;
    rank_of<LId>(own, w, (LId)v, t, d);
    const uint64_t u = (uint64_t)t ^ 1ull;
    seg_head[2 * ch] = (Id)v; seg_head[2 * ch + 1] = (Id)u;
    seg_len[2 * ch] = d + 1; seg_len[2 * ch + 1] = d + 1;
    // t is the end of segment 2c and the flip of the head of 2c+1; v^1 is the end of 2c+1 and the flip of the head of 2c
    // (the words of chain ENDS only: no other chain's rank_of reads them -- a node's owner word lies on its own chain)
    __hip_atomic_store(&w[t], RankW<LId>::pack(2 * ch + 2, t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&w[v ^ 1], RankW<LId>::pack(2 * ch + 1, (LId)(v ^ 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// local segment of the head whose flip is the local chain end t
template <class LId>
__device__ inline uint64_t seg_of_end_flip(const unsigned long long* __restrict__ w, uint64_t t) { return RankW<LId>::dist(w[t]) - 1; }
// the C queries: the segment a chain continues into on another rank
__global__ void __launch_bounds__(256) k_seg_queries(uint64_t nseg, const Id* __restrict__ seg_head, const uint32_t* __restrict__ own, const unsigned long long* __restrict__ w,
                                                      const Id* __restrict__ nxtG, ShardMap M, Stripes Q,
                                                      uint64_t* __restrict__ q_tag, uint64_t* __restrict__ q_p0, uint64_t* __restrict__ seg_next) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const Id tail = seg_head[s ^ 1] ^ (Id)1;                              // the segment ends where its reverse begins
    const Id g = nxtG[tail];
    seg_next[s] = ABSENT;
    if (g == NONE) return;
    const uint64_t at = stripe_slot(Q);
    if (at != ~0ull) { q_tag[at] = ((uint64_t)rank_of_index(M, g >> 1) << 58) | s; q_p0[at] = g; }
}
template <class LId>
__global__ void __launch_bounds__(256) k_answer_seg(uint64_t n, const uint64_t* __restrict__ q, const unsigned long long* __restrict__ w, uint64_t base2_me, uint64_t N,
                                                     uint64_t segbase_me, uint64_t* __restrict__ resp) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t v = q[j] - base2_me;                                   // a local head: its flip is a local chain end
    resp[j] = v < N ? segbase_me + seg_of_end_flip<LId>(w, v ^ 1) : ABSENT;
}
__global__ void __launch_bounds__(256) k_apply_seg(uint64_t n, const uint64_t* __restrict__ tag, const uint64_t* __restrict__ resp, uint64_t* __restrict__ seg_next) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    seg_next[tag[j] & ((1ull << 58) - 1)] = resp[j];
}
// ---------------------------------------------------------------------------------------------- level 2: ranking the segment chains
// What travels: (1) every segment's word (length, next segment or itself: a chain END points at itself -- the jumping never adds an end's
// distance field, which is free to carry ITS length too), 8 bytes, all-gathered: the linked lists of the JOB, replicated; (2) 32-byte
// records, all-gathered: a chain HEAD's oriented k-mer and node (a = segment), and a splitter's first walk (a = bit 63 | segment);
// (3) the second walk's results (segment, its chain's end, its distance to it), 16 bytes, routed to the segment's owner.
// Level 2 is a list ranking over RANDOM-ACCESS lists (a segment's successor is anywhere in the gathered array), so plain pointer jumping
// would move every segment's word log(chain) times, a 64-B sector per 8-B word.  Work-efficient instead (Helman-JaJa): SPLITTERS = the chain
// heads + one segment in 64 by a hash of its number; every splitter WALKS to the next splitter or the chain's end, summing lengths
// (k_seg_walk1); pointer jumping ranks the splitters alone (1/64 of the words); every splitter walks its stretch again and hands every segment
// its chain's end and its distance to it (k_seg_walk2).  The two walks -- the dependent random loads, all of the cost -- are SHARDED: a rank
// walks from the splitters among ITS OWN segments (1/N of the job's), so that the replicated part of level 2 is streaming only (the word
// array, the splitter marks) plus the jumping over 1/64 of the segments.  A circle that has no splitter is never visited, one that has some
// never reaches an end: both stay marked ABSENT.
struct SegMap { uint64_t b[65]; uint32_t world, me; };                    // job-wide segment numbers: rank r holds [b[r], b[r + 1])
__device__ inline unsigned rank_of_seg(const SegMap& G, uint64_t s) {
    unsigned r = 0;
    while (r + 1 < G.world && s >= G.b[r + 1]) ++r;
    return r;
}
struct alignas(32) L2Rec { uint64_t a, b, c, d; };                        // head: segment, k-mer hi, lo, job-wide node; splitter: L2_SPL | segment, word, T, Fend
constexpr uint64_t L2_SPL = 1ull << 63;
constexpr unsigned L2_TSHIFT = 34;                                        // routed result: segment (34 bits) | distance (30 bits, saturating: a unipath beyond
constexpr uint64_t L2_TMAX = (1ull << 30) - 1;                            //   2^24 k-mers is an error anyway, raised at the heads, which are splitters)
__global__ void __launch_bounds__(256) k_seg_words(uint64_t nseg, const uint32_t* __restrict__ seg_len, const uint64_t* __restrict__ seg_next, uint64_t segbase_me,
                                                    unsigned long long* __restrict__ out) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const uint64_t nx = seg_next[s];
    out[s] = RankW<Id>::pack(seg_len[s], nx == ABSENT ? segbase_me + s : nx);
}
constexpr int SEG_JUMPS = 16;
__device__ inline bool seg_sampled(uint64_t s) { return ((s * 0x9E3779B97F4A7C15ull) >> 58) == 0; }
// every rank alike: the splitter marks and the list of ALL splitters (for the jumping); this rank's own splitters apart (for its walks)
__global__ void __launch_bounds__(256) k_seg_mark(uint64_t NS, const unsigned long long* __restrict__ w2o, uint8_t* __restrict__ sp, Stripes LA, uint64_t* __restrict__ spl,
                                                   uint64_t* __restrict__ Fend, uint64_t own_lo, uint64_t own_hi, Stripes LM, uint64_t* __restrict__ spl_me) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= NS) return;
    Fend[s] = ABSENT;
    const bool head = RankW<Id>::next(w2o[s ^ 1]) == (s ^ 1);             // the reverse of s is a chain end: s is a chain head
    const bool is = head || seg_sampled(s);
    sp[s] = is;
    if (is) {
        const uint64_t at = stripe_slot(LA);
        if (at != ~0ull) spl[at] = s;
        if (s >= own_lo && s < own_hi) { const uint64_t am = stripe_slot(LM); if (am != ~0ull) spl_me[am] = s; }
    }
}
// own splitter s -> record (word = (k-mers from its head up to the next splitter's head, that splitter); a stretch that reaches the chain's end F:
// word = (0, s) -- an end of the splitter list --, T = k-mers from its head to the end of the chain, Fend = F), and the number of segments
// between the two splitters
__global__ void __launch_bounds__(256) k_seg_walk1(uint64_t n, const uint64_t* __restrict__ spl_me, const unsigned long long* __restrict__ w2o, const uint8_t* __restrict__ sp,
                                                    L2Rec* __restrict__ out, uint32_t* __restrict__ steps, uint64_t max_steps) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = spl_me[i];
    uint64_t cur = s, acc = 0, st = 0;
    L2Rec r{L2_SPL | s, 0, 0, ABSENT};
    for (;; ++st) {
        const unsigned long long w = w2o[cur];
        const uint64_t nx = RankW<Id>::next(w);
        acc += RankW<Id>::dist(w);
        if (nx == cur) { r.b = RankW<Id>::pack(0, s); r.c = acc; r.d = cur; break; }
        if (sp[nx] || st >= max_steps) { r.b = RankW<Id>::pack(acc, nx); break; }
        cur = nx;
    }
    out[i] = r;
    steps[i] = (uint32_t)st;
}
// own chain heads -> records behind the splitters' (any order)
__global__ void __launch_bounds__(256) k_head_recs(uint64_t nseg, const Id* __restrict__ seg_head, const uint64_t* __restrict__ seg_next, const uint64_t* __restrict__ shi,
                                                    const uint64_t* __restrict__ slo, uint64_t base2_me, uint64_t segbase_me, L2Rec* __restrict__ out,
                                                    unsigned long long* __restrict__ n_out) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg || seg_next[s ^ 1] != ABSENT) return;                   // the reverse of s is not a chain end
    const Id v = seg_head[s];
    Kmer k{shi[v >> 1], slo[v >> 1]};
    if (v & 1) k = kmer_rc(k);
    out[wave_slot(n_out)] = L2Rec{segbase_me + s, k.hi, k.lo, base2_me + v};
}
// the gathered records: the splitters' words into the replicated arrays; where a head's record is
__global__ void __launch_bounds__(256) k_l2_apply(uint64_t n, const L2Rec* __restrict__ R, unsigned long long* __restrict__ w2, uint64_t* __restrict__ Fend, uint64_t* __restrict__ T,
                                                   uint32_t* __restrict__ hr_of_seg) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const L2Rec r = R[j];
    if (r.a & L2_SPL) {
        const uint64_t s = r.a & ~L2_SPL;
        w2[s] = r.b;
        if (r.d != ABSENT) { T[s] = r.c; Fend[s] = r.d; }
    } else hr_of_seg[r.a] = (uint32_t)j;
}
__global__ void __launch_bounds__(256) k_seg_jump(uint64_t n, const uint64_t* __restrict__ spl, unsigned long long* __restrict__ w, uint32_t* __restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t v = spl[i];
    unsigned long long wv = __hip_atomic_load(&w[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Id a = (Id)RankW<Id>::next(wv);
    if (a == v) return;
    bool changed = false, arrived = false;
    for (int round = 0; round < SEG_JUMPS; ++round) {
        const unsigned long long wa = __hip_atomic_load(&w[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const Id b = (Id)RankW<Id>::next(wa);
        if (b == a) { arrived = true; break; }
        wv = RankW<Id>::pack(RankW<Id>::dist(wv) + RankW<Id>::dist(wa), b);
        a = b; changed = true;
    }
    if (changed) __hip_atomic_store(&w[v], wv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!arrived) flags[0] = 1;
}
// the second walk of an own splitter: every segment of its stretch learns the chain's end and its own distance to it -- as an item for the
// segment's owner, at the place the first walk's step counts reserve (a stretch on a circle of splitters hands out ABSENT)
__global__ void __launch_bounds__(256) k_seg_walk2(uint64_t n, const uint64_t* __restrict__ spl_me, const unsigned long long* __restrict__ w2o, const uint8_t* __restrict__ sp,
                                                    const unsigned long long* __restrict__ w2, const uint64_t* __restrict__ Fend, const uint64_t* __restrict__ T,
                                                    const uint64_t* __restrict__ off, SegMap G, uint64_t* __restrict__ q_tag, uint64_t* __restrict__ q_p0,
                                                    uint64_t* __restrict__ q_p1) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = spl_me[i];
    const unsigned long long ws = w2[s];
    const uint64_t e = RankW<Id>::next(ws);                               // the last splitter of the chain, if the splitter list ended
    uint64_t F = RankW<Id>::next(w2[e]) == e ? Fend[e] : ABSENT;          // (a circle of splitters: its segments stay ABSENT)
    uint64_t t = F != ABSENT ? (e == s ? 0 : RankW<Id>::dist(ws)) + T[e] : 0;
    uint64_t cur = s, o = off[i];
    const uint64_t o_end = off[i + 1];
    for (;;) {
        const unsigned long long w = w2o[cur];
        const uint64_t nx = RankW<Id>::next(w);
        if (cur != s && o < o_end) {
            q_tag[o] = ((uint64_t)rank_of_seg(G, cur) << 58) | cur;
            q_p0[o] = cur | ((t < L2_TMAX ? t : L2_TMAX) << L2_TSHIFT);
            q_p1[o] = F;
            ++o;
        }
        if (F != ABSENT) t -= RankW<Id>::dist(w);
        if (nx == cur || sp[nx] || (cur != s && o >= o_end)) break;
        cur = nx;
    }
}
__global__ void __launch_bounds__(256) k_seg_splitters_done(uint64_t n, const uint64_t* __restrict__ spl, const unsigned long long* __restrict__ w2, const uint64_t* __restrict__ Fend,
                                                             const uint64_t* __restrict__ T, uint64_t* __restrict__ Fsp, uint64_t* __restrict__ Tsp) {
    // (two steps so that nothing reads an end splitter's T / Fend while another thread rewrites them: first into side arrays ...)
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = spl[i];
    const unsigned long long ws = w2[s];
    const uint64_t e = RankW<Id>::next(ws);
    if (RankW<Id>::next(w2[e]) != e || Fend[e] == ABSENT) { Fsp[i] = ABSENT; Tsp[i] = 0; return; }
    Fsp[i] = Fend[e]; Tsp[i] = (e == s ? 0 : RankW<Id>::dist(ws)) + T[e];
}
__global__ void __launch_bounds__(256) k_seg_splitters_store(uint64_t n, const uint64_t* __restrict__ spl, const uint64_t* __restrict__ Fsp, const uint64_t* __restrict__ Tsp,
                                                              uint64_t* __restrict__ Fend, uint64_t* __restrict__ T) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (... then into place)
    if (i >= n) return;
    Fend[spl[i]] = Fsp[i]; T[spl[i]] = Tsp[i];
}
// the routed results of the other ranks' walks over MY segments
__global__ void __launch_bounds__(256) k_l2_results(uint64_t n, const uint64_t* __restrict__ r, uint64_t* __restrict__ Fend, uint64_t* __restrict__ T) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t p0 = r[2 * j], sg = p0 & ((1ull << L2_TSHIFT) - 1);
    Fend[sg] = r[2 * j + 1]; T[sg] = p0 >> L2_TSHIFT;
}
// per segment: its own length (every rank alike); MY segments on a circle (never reached from a chain's end): marked and counted
__global__ void __launch_bounds__(256) k_seg_finish(uint64_t NS, const unsigned long long* __restrict__ w2o, uint32_t* __restrict__ len) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < NS) len[s] = (uint32_t)RankW<Id>::dist(w2o[s]);
}
__global__ void __launch_bounds__(256) k_seg_circles(uint64_t nseg, uint64_t segbase_me, const uint64_t* __restrict__ Fend, uint64_t* __restrict__ T, uint8_t* __restrict__ cyc2,
                                                      uint32_t* __restrict__ flags) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const bool cyc = Fend[segbase_me + s] == ABSENT;
    cyc2[segbase_me + s] = cyc;
    if (cyc) { flags[2] = 1; T[segbase_me + s] = 0; }
}

// ---------------------------------------------------------------------------------------------- circles that cross ranks
// the minimum canonical k-mer of every local chain that lies on a level-2 circle (a thread walks its chain: circles are rare and short)
struct alignas(8) MinRec { uint64_t hi, lo, idx; };
template <class LId>
__global__ void __launch_bounds__(256) k_seg_min(uint64_t nchains, const Id* __restrict__ seg_head, const uint32_t* __restrict__ seg_len, const LId* __restrict__ nxtL,
                                                  const uint8_t* __restrict__ cyc2_me, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo, uint64_t base_me,
                                                  MinRec* __restrict__ out) {
    const uint64_t ch = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= nchains) return;
    MinRec m{~0ull, ~0ull, ~0ull};
    if (cyc2_me[2 * ch]) {
        Id v = seg_head[2 * ch];
        for (uint32_t t = 0; t < seg_len[2 * ch]; ++t) {
            const uint64_t i = v >> 1;
            const Kmer k{shi[i], slo[i]};
            if (kmer_lt(k, Kmer{m.hi, m.lo})) { m.hi = k.hi; m.lo = k.lo; m.idx = base_me + i; }
            const LId nv = nxtL[v];
            if (nv == NodeId<LId>::NONE) break;
            v = nv;
        }
    }
    out[ch] = m;
}
__global__ void __launch_bounds__(256) k_segmin_init(uint64_t NS, const unsigned long long* __restrict__ w2o, const MinRec* __restrict__ mr, uint64_t* __restrict__ nx, uint64_t* __restrict__ mn) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= NS) return;
    nx[s] = mr[s >> 1].idx != ~0ull ? RankW<Id>::next(w2o[s]) : s;       // (a chain on a circle has a minimum; k_seg_min leaves ~0 for every other chain)
    mn[s] = s >> 1;                                                       // index into the gathered per-chain minima
}
__global__ void __launch_bounds__(256) k_segmin_jump(uint64_t NS, const MinRec* __restrict__ mr, const uint64_t* __restrict__ nx, const uint64_t* __restrict__ mn,
                                                      uint64_t* __restrict__ nx2, uint64_t* __restrict__ mn2) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= NS) return;
    const uint64_t a = nx[s], m0 = mn[s], m1 = mn[a];
    mn2[s] = kmer_lt(Kmer{mr[m1].hi, mr[m1].lo}, Kmer{mr[m0].hi, mr[m0].lo}) ? m1 : m0;
    nx2[s] = nx[a];
}
// canonicalizeCircle :156-180 on a circle that crosses ranks: the owner of the minimum k-mer m cuts in front of (m, 0): its own word of
// (m, 1), and -- listed for whoever owns it -- the link of the predecessor
__global__ void __launch_bounds__(256) k_seg_cuts(uint64_t nseg_me, uint64_t segbase_me, const uint8_t* __restrict__ cyc2, const uint64_t* __restrict__ mn, const MinRec* __restrict__ mr,
                                                   ShardMap M, Id* __restrict__ nxtG, unsigned long long* __restrict__ ncut, uint64_t cap, uint64_t* __restrict__ cuts) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg_me || !cyc2[segbase_me + s]) return;
    const uint64_t m = mr[mn[segbase_me + s]].idx;
    if (m < M.base[M.me] || m >= M.base[M.me + 1]) return;
    const uint64_t i = m - M.base[M.me];
    const Id u = (Id)atomicExch(reinterpret_cast<unsigned long long*>(&nxtG[2 * i + 1]), (unsigned long long)NONE);      // the first segment to come does the cut
    if (u == NONE) return;
    const unsigned long long at = atomicAdd(ncut, 1ull);
    if (at < cap) cuts[at] = u ^ (Id)1;
}
__global__ void __launch_bounds__(256) k_apply_cuts(uint64_t n, const uint64_t* __restrict__ cuts, ShardMap M, Id* __restrict__ nxtG) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t g = cuts[j];
    if ((g >> 1) >= M.base[M.me] && (g >> 1) < M.base[M.me + 1]) nxtG[g - 2 * M.base[M.me]] = NONE;
}

// ---------------------------------------------------------------------------------------------- unipaths from the segment arrays
// k-mers from node v to the end of its unipath chain (inclusive), and the end segment
template <class LId>
__device__ inline uint64_t to_end(const uint32_t* __restrict__ own, const unsigned long long* __restrict__ w, uint64_t segbase_me, const uint32_t* __restrict__ len,
                                  const uint64_t* __restrict__ Fend, const uint64_t* __restrict__ T, uint64_t v, uint64_t& F) {
    LId t; uint32_t d;
    rank_of<LId>(own, w, (LId)v, t, d);
    const uint64_t sg = segbase_me + (seg_of_end_flip<LId>(w, t) ^ 1);    // the segment that ends at t
    F = Fend[sg];
    return (uint64_t)d + 1 + T[sg] - len[sg];
}
// the middle base of the unipaths with an odd number of bases, as seen from the head segment of each orientation (k_rank_finish of step2_graph.hip)
template <class LId>
__global__ void __launch_bounds__(256) k_mid_shard(uint64_t S, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo, const uint32_t* __restrict__ own,
                                                    const unsigned long long* __restrict__ w, uint64_t segbase_me, const uint32_t* __restrict__ len,
                                                    const uint64_t* __restrict__ Fend, const uint64_t* __restrict__ T, uint8_t* __restrict__ mid) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    // One walk to the end, not two: the chain through (i, 0) ends in segment F0, whose flip is the HEAD of the reverse chain, and a head's T is
    // the chain's length -- n, the same for both orientations -- so the distance to the other end is n - 1 - r0.  The second walk (its end
    // names the slot) is left to the one k-mer in the middle of a unipath (six random words per walk: 4.6 -> 2.3 ms for 312 M k-mers).
    uint64_t F0, F1;
    const uint64_t r0 = to_end<LId>(own, w, segbase_me, len, Fend, T, 2 * i, F0) - 1;
    const uint64_t n = T[F0 ^ 1];
    if (n & 1) return;
    const uint64_t r1 = n - 1 - r0;
    const uint64_t q = n / 2 + 29, x = q < n - 1 ? q : n - 1;
    if (r1 != x && r0 != x) return;
    const unsigned off = (unsigned)(q - x);
    const Kmer k{shi[i], slo[i]};
    if (r1 == x) { (void)to_end<LId>(own, w, segbase_me, len, Fend, T, 2 * i + 1, F1); mid[F1 ^ 1] = (uint8_t)(4u | kmer_base(k, off)); }    // the head segment of the chain through (i, 0) is the flip of (i, 1)'s end
    if (r0 == x) mid[F0 ^ 1] = (uint8_t)(4u | kmer_base(kmer_rc(k), off));
}
// canonical heads (k_heads of step2_graph.hip, on segments): bvec::getCanonicalForm
__global__ void __launch_bounds__(256) k_heads_shard(uint64_t nR, const L2Rec* __restrict__ R, const uint32_t* __restrict__ hr_of_seg, const uint64_t* __restrict__ Fend,
                                                      const uint64_t* __restrict__ T, const uint8_t* __restrict__ mid, uint64_t* __restrict__ head_seg,
                                                      uint64_t* __restrict__ key_hi, uint64_t* __restrict__ key_lo, unsigned long long* __restrict__ n_heads, uint64_t cap,
                                                      uint32_t* __restrict__ flags) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nR) return;
    const L2Rec r = R[j];
    if (r.a & L2_SPL) return;                                             // (a splitter's record)
    const uint64_t H = r.a;                                               // a chain head: a splitter, its Fend / T are on every rank
    const uint64_t F = Fend[H];
    if (F == ABSENT) return;
    const uint64_t n = T[H];
    if (n - 1 > 0xFFFFFFull) atomicOr(&flags[1], 2u);                     // GE_OFFSET, ReadPather.h:122
    const Kmer Fk{r.b, r.c};
    bool canon;
    if (kmer_is_pal(Fk)) canon = !(r.d & 1);
    else if (n & 1) { const L2Rec o = R[hr_of_seg[F ^ 1]]; canon = kmer_lt(Fk, Kmer{o.b, o.c}); }      // the head of the reverse chain
    else canon = !(mid[H] & 2);
    if (!canon) return;
    const unsigned long long pos = wave_slot(n_heads);
    if (pos < cap) { head_seg[pos] = H; key_hi[pos] = Fk.hi; key_lo[pos] = Fk.lo; }
}
__global__ void __launch_bounds__(256) k_iota32(uint64_t n, uint32_t* __restrict__ a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(256) k_tie_sort(uint64_t E, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ lo, uint32_t* __restrict__ perm) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= E || (j > 0 && shi[j] == shi[j - 1])) return;
    uint64_t b = j + 1;
    while (b < E && shi[b] == shi[j]) ++b;
    for (uint64_t i = j + 1; i < b; ++i) {
        const uint32_t x = perm[i]; const uint64_t lx = lo[x];
        uint64_t t = i;
        while (t > j && lo[perm[t - 1]] > lx) { perm[t] = perm[t - 1]; --t; }
        perm[t] = x;
    }
}
// unipath e = the rank-th canonical head in sorted order (canonical mode): its length, and its number at its head segment
__global__ void __launch_bounds__(256) k_edges_sorted(uint64_t E, const uint32_t* __restrict__ perm, const uint64_t* __restrict__ head_seg, const uint64_t* __restrict__ T,
                                                       uint32_t* __restrict__ edge_nk, uint32_t* __restrict__ edge_of_head) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const uint64_t H = head_seg[perm[e]];
    edge_nk[e] = (uint32_t)T[H];
    edge_of_head[H] = (uint32_t)e + 1;
}
// replay: unipath e = the canonical head whose first 60-mer is hint e's (binary search in the sorted heads)
__global__ void __launch_bounds__(256) k_edges_hint(uint64_t E, const uint64_t* __restrict__ hk_hi, const uint64_t* __restrict__ hk_lo, const uint32_t* __restrict__ hk_len,
                                                     const uint32_t* __restrict__ perm, const uint64_t* __restrict__ key_hi, const uint64_t* __restrict__ key_lo,
                                                     const uint64_t* __restrict__ head_seg, const uint64_t* __restrict__ T, uint32_t* __restrict__ edge_nk,
                                                     uint32_t* __restrict__ edge_of_head, uint32_t* __restrict__ flags) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const Kmer k{hk_hi[e], hk_lo[e]};
    edge_nk[e] = 1;
    uint64_t lo = 0, hi = E;                                             // first sorted head with key >= k
    while (lo < hi) { const uint64_t md = (lo + hi) >> 1; const uint32_t x = perm[md]; if (kmer_lt(Kmer{key_hi[x], key_lo[x]}, k)) lo = md + 1; else hi = md; }
    if (lo >= E || key_hi[perm[lo]] != k.hi || key_lo[perm[lo]] != k.lo) { atomicOr(&flags[1], 4u); return; }     // GE_HINT_MISS
    const uint64_t H = head_seg[perm[lo]];
    if (hk_len[e] != T[H] + (K - 1)) { atomicOr(&flags[1], 32u); return; }                                      // GE_HINT_LEN
    if (atomicExch(&edge_of_head[H], (uint32_t)e + 1) != 0) atomicOr(&flags[1], 8u);                              // GE_HINT_DUP
    edge_nk[e] = (uint32_t)T[H];
}
__global__ void __launch_bounds__(256) k_edge_len2(uint64_t E, const uint32_t* __restrict__ edge_nk, uint32_t* __restrict__ len) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) len[e] = edge_nk[e] + (K - 1);
}
// every local k-mer deposits its base(s) of its unipath's sequence (k_assign of step2_graph.hip) as byte codes into a zeroed array -- plain
// stores: no two k-mers write the same base --, which is then packed 16 bases per word; the ranks' packed streams are summed (the 2-bit
// groups a rank does not own stay zero).  (Round 5, first form: atomicOr straight into the packed words -- 9.2 ms for 312 M k-mers, the
// device-atomic rate; bytes + pack: 3.4 ms.)
template <class LId>
__global__ void __launch_bounds__(256) k_assign_shard(uint64_t S, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo, const uint32_t* __restrict__ own,
                                                       const unsigned long long* __restrict__ w, uint64_t segbase_me, const uint32_t* __restrict__ len,
                                                       const uint64_t* __restrict__ Fend, const uint64_t* __restrict__ T, const uint32_t* __restrict__ edge_of_head,
                                                       const uint64_t* __restrict__ edge_off, uint8_t* __restrict__ codes, uint32_t* __restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    // One walk to the end of the chain through (i, 0): the flip of its end segment F0 is the head of the reverse chain, and a head knows its
    // chain's end and length -- the end of the chain through (i, 1) and both distances follow without the second walk (k_mid_shard; six random
    // words per walk).  A head without its end (never seen here) walks as before.
    uint64_t F0, F1;
    const uint64_t r0 = to_end<LId>(own, w, segbase_me, len, Fend, T, 2 * i, F0) - 1;
    uint64_t r1;
    F1 = Fend[F0 ^ 1];
    if (F1 != ABSENT) r1 = T[F0 ^ 1] - 1 - r0;
    else r1 = to_end<LId>(own, w, segbase_me, len, Fend, T, 2 * i + 1, F1) - 1;
    // the chain through node 2i starts at the head segment F1^1, the one through 2i+1 at F0^1: exactly one of the two is a canonical head
    uint32_t e = edge_of_head[F1 ^ 1] - 1u; uint64_t off = r1; bool rev = false;
    if (e == NONE32) { e = edge_of_head[F0 ^ 1] - 1u; off = r0; rev = true; }
    if (e == NONE32) { atomicOr(&flags[1], 16u); return; }               // GE_ASSIGN
    Kmer k{shi[i], slo[i]};
    if (rev) k = kmer_rc(k);
    uint8_t* dst = codes + edge_off[e];
    if (off == 0) { for (unsigned t = 0; t < K; ++t) dst[t] = (uint8_t)kmer_base(k, t); }
    else dst[K - 1 + off] = (uint8_t)kmer_last(k);
}
__global__ void __launch_bounds__(256) k_pack_words(uint64_t nwords, uint64_t nbases, const uint8_t* __restrict__ codes, uint32_t* __restrict__ bits) {
    const uint64_t wd = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wd >= nwords) return;
    uint32_t v = 0;
    const uint64_t g0 = 16 * wd;
    if (g0 + 16 <= nbases) {
        const uint4 q = *reinterpret_cast<const uint4*>(codes + g0);     // (the array is 256-byte aligned)
        const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (unsigned j = 0; j < 4; ++j)
#pragma unroll
            for (unsigned b = 0; b < 4; ++b) v |= ((d[j] >> (8 * b)) & 3u) << (2 * (4 * j + b));
    } else for (unsigned j = 0; j < 16 && g0 + j < nbases; ++j) v |= (uint32_t)(codes[g0 + j] & 3) << (2 * j);
    bits[wd] = v;
}
__global__ void __launch_bounds__(256) k_unpack_codes(uint64_t nbases, const uint32_t* __restrict__ bits, uint8_t* __restrict__ codes) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < nbases) codes[g] = (uint8_t)((bits[g >> 4] >> (2 * (unsigned)(g & 15))) & 3u);
}

// ============================================================================================== the state machine
enum Phase { PH_BEGIN = 0, PH_A_ANSWER, PH_A_APPLY, PH_B_ANSWER, PH_B_APPLY, PH_SEGBASE, PH_C_ANSWER, PH_C_APPLY, PH_LEVEL2, PH_L2_JUMP, PH_L2_RESULTS, PH_L2_CIRCLES, PH_CIRC_MIN, PH_CIRC_CUT,
             PH_HEADS, PH_STREAM, PH_INDEX, PH_INDEX_HARD, PH_FILTER, PH_DONE };

struct Shard {
    ShardMap M{};
    const w2rap_edge_hint* hint = nullptr;
    int phase = PH_BEGIN;
    int circle_rounds = 0;
    uint64_t S = 0;
    // pending routed queries
    uint64_t nq = 0; uint64_t* q_tag = nullptr; uint64_t* q_send = nullptr; uint64_t q_counts[64] = {0};
    // what arrived
    void* recv = nullptr; uint64_t recv_total = 0; uint64_t recv_counts[64] = {0};
    uint64_t* resp = nullptr;
    uint64_t recv_host[64] = {0};                                         // the words of a host all-gather
    // prune / links
    uint8_t* nctx = nullptr; Id* nxtG = nullptr; void* nxtL = nullptr;
    bool local32 = false;                                                 // the rank's own node numbers are 32-bit words (fewer than 2^31 - 1 owned k-mers)
    unsigned long long* rankw = nullptr; uint32_t* own = nullptr; uint32_t* d_flags = nullptr;
    // segments
    uint64_t nseg = 0, segbase[65] = {0}, NS = 0;
    Id* seg_head = nullptr; uint32_t* seg_len = nullptr; uint64_t* seg_next = nullptr; unsigned long long* seg_w = nullptr;
    uint64_t* h_small = nullptr;                                          // pinned host words for the tiny all-gathers
    // level 2: the gathered 32-B records (heads, splitters' first walks), where a head's record is, the splitter marks and lists
    L2Rec* R = nullptr; uint64_t nR = 0; uint32_t* hr_of_seg = nullptr; L2Rec* l2_send = nullptr;
    uint8_t* sp = nullptr; uint64_t *spl = nullptr, *spl_me = nullptr, *l2_off = nullptr; uint64_t nspl = 0, nspl_me = 0; uint32_t* l2_steps = nullptr;
    unsigned long long *w2 = nullptr, *w2o = nullptr; uint64_t *Fend = nullptr, *T = nullptr; uint8_t *cyc2 = nullptr, *mid = nullptr;
    uint32_t* lenS = nullptr;
    MinRec* minrec = nullptr; uint64_t* cuts = nullptr; uint64_t *mn = nullptr;
    uint32_t* edge_of_head = nullptr; uint32_t* bits = nullptr; uint32_t* bits_keep = nullptr; uint64_t nwords = 0;
    uint4* idx_list = nullptr; unsigned long long* flt_slice = nullptr; uint64_t flt_words = 0;
};

static Shard& sh(Ctx& c) { return *static_cast<Shard*>(c.shard); }
// a kernel templated on the local id type, launched for the width this rank uses
// (the arguments may name the type as LId: `(const LId*)s.nxtL`)
#define LAUNCH_L(c, s, name, kern, grid, block, ...)                                                   \
    do {                                                                                              \
        if ((s).local32) { using LId = uint32_t; LAUNCH(c, name, (kern<LId>), grid, block, 0, __VA_ARGS__); }   \
        else { using LId = uint64_t; LAUNCH(c, name, (kern<LId>), grid, block, 0, __VA_ARGS__); }               \
    } while (0)

// ---- a striped list on the host side: counters, counts, prefix sums
struct StripeList {
    Stripes L{nullptr, 0, 1};
    std::vector<uint64_t> n;                                              // entries per sub-list (capped at cap)
    uint64_t total = 0, maxn = 0, max_wanted = 0;                         // max_wanted: the largest counter, beyond cap if the sub-list overflowed
    bool overflow = false;
    uint64_t* d_pre = nullptr;                                            // [k + 1] exclusive prefix sums, if asked for
};
static void stripes_free(Ctx& c, StripeList& sl) {
    if (sl.L.cnt) c.release(sl.L.cnt);
    if (sl.d_pre) c.release(sl.d_pre);
    sl.L.cnt = nullptr; sl.d_pre = nullptr;
}
// `blocks`: of the kernel that appends (a block appends to sub-list blockIdx.x % k); cap: entries per sub-list
static int stripes_begin(Ctx& c, StripeList& sl, uint64_t blocks, uint64_t cap) {
    stripes_free(c, sl);
    sl.L.k = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(NSTRIPE, blocks));
    sl.L.cap = cap ? cap : 1;
    W2_ALLOC(sl.L.cnt, unsigned long long, (uint64_t)NSTRIPE * STRIPE_PAD);
    W2_HIP(hipMemsetAsync(sl.L.cnt, 0, (uint64_t)NSTRIPE * STRIPE_PAD * 8, c.stream));
    return 0;
}
// a dense array of n entries as a list of one sub-list
static int stripes_dense(Ctx& c, StripeList& sl, uint64_t n) {
    W2_TRY(stripes_begin(c, sl, 1, n));
    const unsigned long long v = n;
    W2_HIP(hipMemcpyAsync(sl.L.cnt, &v, 8, hipMemcpyHostToDevice, c.stream));
    W2_HIP(hipStreamSynchronize(c.stream));
    sl.n.assign(1, n); sl.total = n; sl.maxn = n; sl.max_wanted = n; sl.overflow = false;
    return 0;
}
static int stripes_counts(Ctx& c, StripeList& sl, bool want_prefix) {
    std::vector<unsigned long long> h((size_t)sl.L.k * STRIPE_PAD);
    W2_HIP(hipMemcpyAsync(h.data(), sl.L.cnt, h.size() * 8, hipMemcpyDeviceToHost, c.stream));
    W2_HIP(hipStreamSynchronize(c.stream));
    W2_HIP(hipGetLastError());
    sl.n.assign(sl.L.k, 0); sl.total = 0; sl.maxn = 0; sl.max_wanted = 0; sl.overflow = false;
    std::vector<uint64_t> pre(sl.L.k + 1, 0);
    for (unsigned k = 0; k < sl.L.k; ++k) {
        const uint64_t w = h[(size_t)k * STRIPE_PAD];
        sl.max_wanted = std::max(sl.max_wanted, w);
        if (w > sl.L.cap) sl.overflow = true;
        sl.n[k] = std::min<uint64_t>(w, sl.L.cap);
        pre[k] = sl.total;
        sl.total += sl.n[k]; sl.maxn = std::max(sl.maxn, sl.n[k]);
    }
    pre[sl.L.k] = sl.total;
    if (want_prefix && !sl.overflow) {
        if (sl.d_pre) c.release(sl.d_pre);
        W2_ALLOC(sl.d_pre, uint64_t, sl.L.k + 1);
        W2_HIP(hipMemcpyAsync(sl.d_pre, pre.data(), pre.size() * 8, hipMemcpyHostToDevice, c.stream));
        W2_HIP(hipStreamSynchronize(c.stream));                           // (pre is a local)
    }
    return 0;
}
static dim3 stripes_grid(const StripeList& sl, unsigned per_block) { return dim3((unsigned)std::max<uint64_t>(1, (sl.maxn + per_block - 1) / per_block), sl.L.k); }

// items (tag, p0[, p1]) of a striped list -> parts by destination; fills x for the all-to-all; keeps the tags (in send order) for the answers
static int route(Ctx& c, Shard& s, const StripeList& sl, const uint64_t* tag, const uint64_t* p0, const uint64_t* p1, w2rap_xchg* x) {
    hipStream_t st = c.stream;
    const uint64_t n = sl.total;
    const unsigned W = s.M.world, words = p1 ? 2 : 1;
    const dim3 grid = stripes_grid(sl, RT_ITEMS);
    const uint64_t NB = (uint64_t)grid.x * grid.y;
    uint32_t* bh = nullptr; uint64_t* boff = nullptr;
    W2_ALLOC(bh, uint32_t, W * NB + 1); W2_ALLOC(boff, uint64_t, W * NB + 2);
    LAUNCH(c, "k_route_count", k_route_count, grid, dim3(256), 0, sl.L, tag, (uint32_t)W, NB, bh);
    W2_TRY(exclusive_scan_u32_to_u64(c, bh, boff, W * NB));
    uint64_t h_off[65];
    for (unsigned r = 0; r <= W; ++r) W2_HIP(hipMemcpyAsync(&h_off[r], boff + (uint64_t)r * NB, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    for (unsigned r = 0; r < 64; ++r) s.q_counts[r] = r < W ? h_off[r + 1] - h_off[r] : 0;
    if (h_off[W] != n) { c.err = "sharded graph: routed " + std::to_string(h_off[W]) + " of " + std::to_string(n) + " items"; return W2RAP_E_STATE; }
    if (s.q_tag) c.release(s.q_tag);
    if (s.q_send) c.release(s.q_send);
    W2_ALLOC(s.q_tag, uint64_t, n + 1); W2_ALLOC(s.q_send, uint64_t, n * words + 1);
    if (n) LAUNCH(c, "k_route_scatter", k_route_scatter, grid, dim3(256), 0, sl.L, tag, p0, p1, NB, (const uint64_t*)boff, s.q_tag, s.q_send);
    W2_HIP(hipStreamSynchronize(st));
    W2_HIP(hipGetLastError());
    c.release(bh); c.release(boff);
    s.nq = n;
    std::memset(x, 0, sizeof(*x));
    x->op = W2RAP_X_ALLTOALL; x->elem_bytes = 8 * words; x->send = s.q_send;
    for (unsigned r = 0; r < s.M.world; ++r) x->send_count[r] = s.q_counts[r];
    return 0;
}
// the answers travel back: what came from rank r goes to rank r, in the order it came
static void respond(Shard& s, w2rap_xchg* x) {
    std::memset(x, 0, sizeof(*x));
    x->op = W2RAP_X_ALLTOALL; x->elem_bytes = 8; x->send = s.resp;
    for (unsigned r = 0; r < s.M.world; ++r) x->send_count[r] = s.recv_counts[r];
}

static int shard_error(Ctx& c, uint32_t f) {
    if (f & 2) { c.err = "unipath longer than 16,777,215 k-mers (ForceAssertLe, ReadPather.h:122)"; return W2RAP_E_GRAPH; }
    if (f & 16) { c.err = "k-mer left without an edge (BuildReadQGraph.cc:303)"; return W2RAP_E_GRAPH; }
    if (f & 4) { c.err = "edge_order_hint: a hinted edge is not a unipath of this graph"; return W2RAP_E_HINT; }
    if (f & 8) { c.err = "edge_order_hint: an edge is listed twice"; return W2RAP_E_HINT; }
    if (f & 32) { c.err = "edge_order_hint: a hinted edge has the wrong length"; return W2RAP_E_HINT; }
    return 0;
}

void shard_free(Ctx& c) {
    if (!c.shard) return;
    Shard& s = sh(c);
    if (s.h_small) (void)hipHostFree(s.h_small);
    delete &s;
    c.shard = nullptr;
}

int shard_begin(Ctx& c, unsigned rank, unsigned world, const uint64_t* solid_per_rank, uint32_t n_buckets, uint32_t n_passes, const w2rap_edge_hint* hint) {
    if (!world || world > 64 || rank >= world || !n_buckets || !n_passes || n_buckets % (world * n_passes)) { c.err = "shard_begin: bad rank / world / bucket geometry"; return W2RAP_E_ARG; }
    if (c.cs_planned) { c.err = "shard_begin while a sliced count is pending (count_records_end first)"; return W2RAP_E_STATE; }
    if (!c.d_shi && c.S) { c.err = "shard_begin before the owner's count"; return W2RAP_E_STATE; }
    shard_free(c);
    Shard* sp = new Shard;
    c.shard = sp;
    Shard& s = *sp;
    s.M.world = world; s.M.me = rank; s.M.NB = n_buckets; s.M.per_pass = n_buckets / n_passes; s.M.nbl = s.M.per_pass / world;
    s.M.base[0] = 0;
    for (unsigned r = 0; r < world; ++r) s.M.base[r + 1] = s.M.base[r] + solid_per_rank[r];
    for (unsigned r = world + 1; r < 65; ++r) s.M.base[r] = s.M.base[world];
    if (solid_per_rank[rank] != c.S) { c.err = "shard_begin: this rank's solid count does not match its context"; return W2RAP_E_ARG; }
    if (s.M.base[world] >= (1ull << 52)) { c.err = "more than 2^52 solid k-mers"; return W2RAP_E_LIMIT; }
    if (c.S >= MAX_SOLID_KMERS) { c.err = "more than 2^32 solid k-mers on one GPU"; return W2RAP_E_LIMIT; }
    s.M.test_virtual = 0;
    if (test_hook("W2RAP_TEST_SHARD_VIRTUAL")) { const uint32_t v = (uint32_t)atoi(getenv("W2RAP_TEST_SHARD_VIRTUAL")); if (v >= 2 && v <= s.M.per_pass) s.M.test_virtual = v; }
    s.M.test_cut = test_hook("W2RAP_TEST_SHARD_CUT") ? (uint32_t)std::max(2, atoi(getenv("W2RAP_TEST_SHARD_CUT"))) : 0u;
    s.hint = hint; s.S = c.S; s.phase = PH_BEGIN;
    // Memory that the counting needed and this phase does not (measured at the per-GPU share of BASELINE configs[4], tests/test_gpu_scale.py:
    // the phase's peak is ~70 B per owned k-mer on top of what is live here):
    //  * the super-k-mer records this rank cut for the last pass -- every owner has pulled its part by now;
    //  * the slack of the solid arrays, which the count sized by its only hard bound, instances / min_freq: 4x the k-mers a 30x data set
    //    really has (20 B each) -- the arrays are copied into blocks of their exact size, one after the other.
    if (c.d_recs) { c.release(c.d_recs); c.d_recs = nullptr; c.nrec = 0; }
    if (c.d_shi && c.solid_cap > c.S && (c.solid_cap - c.S) * 20 > (2ull << 30) && !getenv("W2RAP_NO_SOLID_SHRINK")) {
        if (c.stream2) W2_HIP(hipStreamSynchronize(c.stream2));           // (the owner's dictionary may still be reading the arrays on the side stream)
        const uint64_t n = c.S + 1;
        uint64_t* a = nullptr; uint32_t* b = nullptr;
        W2_ALLOC(a, uint64_t, n); W2_HIP(hipMemcpyAsync(a, c.d_shi, c.S * 8, hipMemcpyDeviceToDevice, c.stream)); W2_HIP(hipStreamSynchronize(c.stream)); c.release(c.d_shi); c.d_shi = a;
        a = nullptr;
        W2_ALLOC(a, uint64_t, n); W2_HIP(hipMemcpyAsync(a, c.d_slo, c.S * 8, hipMemcpyDeviceToDevice, c.stream)); W2_HIP(hipStreamSynchronize(c.stream)); c.release(c.d_slo); c.d_slo = a;
        W2_ALLOC(b, uint32_t, n); W2_HIP(hipMemcpyAsync(b, c.d_scc, c.S * 4, hipMemcpyDeviceToDevice, c.stream)); W2_HIP(hipStreamSynchronize(c.stream)); c.release(c.d_scc); c.d_scc = b;
        c.solid_cap = n;
        // (the pool parks the three big blocks for the next count of this context; a caller short of memory trims: w2rap_step2_trim)
        if (c.parked.size()) { size_t free_b = 0, total_b = 0; if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < 90 * c.S) c.trim(); }
    }
    W2_HIP(hipHostMalloc((void**)&s.h_small, 64 * 8, hipHostMallocDefault));
    c.use_index = true; c.wide_ids = true; c.counted = false; c.graphed = false;
    return 0;
}

// room for what an exchange delivers
int shard_recv(Ctx& c, const uint64_t* recv_count, uint32_t elem_bytes, void** d_recv) {
    if (!c.shard) { c.err = "shard_recv before shard_begin"; return W2RAP_E_STATE; }
    Shard& s = sh(c);
    uint64_t tot = 0;
    for (unsigned r = 0; r < s.M.world; ++r) { s.recv_counts[r] = recv_count[r]; tot += recv_count[r]; }
    if (s.recv) { c.release(s.recv); s.recv = nullptr; }
    uint8_t* p = nullptr;
    W2_ALLOC(p, uint8_t, tot * elem_bytes + 64);
    s.recv = p; s.recv_total = tot;
    *d_recv = p;
    return 0;
}

// ---- the pieces between the exchanges
static int prune_emit(Ctx& c, Shard& s, w2rap_xchg* x) {                   // PH_BEGIN: dictionary of the owned k-mers, local prune, A queries
    hipStream_t st = c.stream;
    const uint64_t S = s.S;
    // the dictionary of the owned k-mers: built slice by slice under the counting (local_dict_slice), or here in one go
    // (the last slice's inserts may still be running on the side stream: the chunk-local prune below does not need the table and runs beside
    //  them, as on one GPU; the side stream is waited for in front of the first probe)
    const bool prebuilt = c.table_built && c.d_table && c.ld_done == S && 10 * c.tcap >= 13 * S;
    if (!prebuilt) {
        if (c.stream2) W2_HIP(hipStreamSynchronize(c.stream2));           // (an abandoned table may still be taking inserts: they end before its block is reused)
        W2_TRY(table_build_plain(c));
    }
    uint8_t* sctx0 = nullptr; void* nbrL = nullptr; uint8_t* unres = nullptr;
    {
        const char* wv = getenv("W2RAP_WIDE_IDS");
        s.local32 = S < (1ull << 31) - 1 && !(wv && atoi(wv) != 0);
    }
    W2_ALLOC(sctx0, uint8_t, S + 4); W2_ALLOC(unres, uint8_t, S + 4);
    if (s.local32) { uint32_t* q = nullptr; W2_ALLOC(q, uint32_t, 2 * S); nbrL = q; } else { uint64_t* q = nullptr; W2_ALLOC(q, uint64_t, 2 * S); nbrL = q; }
    bool have_local = false;
    if (S && c.nchunks) { W2_TRY(s.local32 ? prune_local_chunks32(c, sctx0, (uint32_t*)nbrL, unres) : prune_local_chunks64(c, sctx0, (uint64_t*)nbrL, unres)); have_local = true; }
    if (prebuilt) {                                                       // the owner's dictionary, built slice by slice under the counting: complete before the probes
        if (c.stream2) {
            hipEvent_t ev;
            W2_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            W2_HIP(hipEventRecord(ev, c.stream2));
            W2_HIP(hipStreamWaitEvent(st, ev, 0));
            (void)hipEventDestroy(ev);
        }
        c.table_built = false;
    }
    for (void* p : {(void*)c.d_sctx, (void*)c.d_nbr}) if (p) c.release(p);
    c.d_nbr = nullptr;
    W2_ALLOC(c.d_sctx, uint8_t, S + 4);
    W2_HIP(hipMemsetAsync(c.d_sctx, 0, S + 4, st));
    W2_ALLOC(s.nxtG, Id, 2 * S);
    // the queries: a striped list (~0.7 (N-1)/N per k-mer); a sub-list that turns out too small is followed by the exact size
    StripeList ql;
    const uint64_t blocks = grid_for(S);
    uint64_t qcap = S / 2 + 4096;                                       // (measured: 0.26 per k-mer at 7/8 of the neighbours on other ranks -- the chunk-local prune settles most)
    if (test_hook("W2RAP_TEST_SHARD_QCAP")) qcap = (uint64_t)atoll(getenv("W2RAP_TEST_SHARD_QCAP"));
    uint64_t cap = qcap / std::max<uint64_t>(1, std::min<uint64_t>(NSTRIPE, blocks)) + 1;
    uint64_t *q_tag = nullptr, *q_hi = nullptr, *q_lo = nullptr;
    for (int attempt = 0;; ++attempt) {
        W2_TRY(stripes_begin(c, ql, blocks, cap));
        const uint64_t room = (uint64_t)ql.L.k * ql.L.cap;
        W2_ALLOC(q_tag, uint64_t, room); W2_ALLOC(q_hi, uint64_t, room); W2_ALLOC(q_lo, uint64_t, room);
        if (S) {
            if (s.local32) LAUNCH(c, "k_prune_shard", k_prune_shard<uint32_t>, dim3(grid_for(S)), dim3(256), 0, S, c.d_shi, c.d_slo, c.d_scc, c.d_table, c.tcap - 1, (const uint8_t*)sctx0,
                                  (const uint32_t*)nbrL, have_local ? (const uint8_t*)unres : (const uint8_t*)nullptr, s.M, c.d_sctx, s.nxtG, ql.L, q_tag, q_hi, q_lo);
            else LAUNCH(c, "k_prune_shard", k_prune_shard<uint64_t>, dim3(grid_for(S)), dim3(256), 0, S, c.d_shi, c.d_slo, c.d_scc, c.d_table, c.tcap - 1, (const uint8_t*)sctx0,
                        (const uint64_t*)nbrL, have_local ? (const uint8_t*)unres : (const uint8_t*)nullptr, s.M, c.d_sctx, s.nxtG, ql.L, q_tag, q_hi, q_lo);
        }
        W2_TRY(stripes_counts(c, ql, false));
        if (!ql.overflow) break;
        if (attempt) { c.err = "sharded prune: query list overflow after resizing"; return W2RAP_E_LIMIT; }
        c.release(q_tag); c.release(q_hi); c.release(q_lo);
        cap = ql.max_wanted + 64;                                         // more open neighbours than room: their number is known now
    }
    const unsigned long long nq = ql.total;
    c.release(sctx0); c.release(nbrL); c.release(unres);
    W2_TRY(route(c, s, ql, q_tag, q_hi, q_lo, x));
    c.release(q_tag); c.release(q_hi); c.release(q_lo);
    stripes_free(c, ql);
    if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] shard %u/%u: %llu solid k-mers, %llu neighbour queries to other owners\n", s.M.me, s.M.world, (unsigned long long)S, nq);
    return 0;
}

template <class LId>
__global__ void __launch_bounds__(256) k_mirror_cuts(uint64_t N, const LId* __restrict__ nxtL, ShardMap M, Id* __restrict__ nxtG) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const Id g = nxtG[v];
    if (g != NONE && nxtL[v] == NodeId<LId>::NONE && (g >> 1) >= M.base[M.me] && (g >> 1) < M.base[M.me + 1] && link_stays_local(M, 2 * M.base[M.me] + v, g)) nxtG[v] = NONE;
}

static int links_and_segments(Ctx& c, Shard& s, w2rap_xchg* x) {           // local ranking, segments; -> tiny all-gather of the segment counts
    hipStream_t st = c.stream;
    const uint64_t S = s.S, N = 2 * S;
    if (!s.rankw) { W2_ALLOC(s.rankw, unsigned long long, N); W2_ALLOC(s.own, uint32_t, N); }
    if (!s.d_flags) W2_ALLOC(s.d_flags, uint32_t, 8);
    W2_HIP(hipMemsetAsync(s.d_flags, 0, 32, st));
    uint8_t* cyc = nullptr;
    W2_ALLOC(cyc, uint8_t, N + 4);
    bool had_circles = false;
    if (S) W2_TRY(s.local32 ? rank_resolve32(c, N, (uint32_t*)s.nxtL, s.rankw, s.own, cyc, nullptr, s.d_flags, c.d_shi, c.d_slo, &had_circles)
                            : rank_resolve64(c, N, (uint64_t*)s.nxtL, s.rankw, s.own, cyc, nullptr, s.d_flags, c.d_shi, c.d_slo, &had_circles));
    c.release(cyc);
    // a circle inside the rank was cut in nxtL: the job-wide links follow (wherever nxtL is NONE and nxtG is a local link, nxtG becomes NONE)
    if (had_circles) LAUNCH_L(c, s, "k_mirror_cuts", k_mirror_cuts, dim3(grid_for(N)), dim3(256), N, (const LId*)s.nxtL, s.M, s.nxtG);
    // the local chains, listed (striped) and then numbered sub-list after sub-list
    const uint64_t bound = S ? c.rank_ends + 2 : 2;                       // chains <= chain ends
    StripeList hl;
    uint64_t* heads = nullptr;
    uint64_t cap = bound / std::max<uint64_t>(1, std::min<uint64_t>(NSTRIPE, grid_for(N))) + bound / 4096 + 256;
    for (int attempt = 0;; ++attempt) {
        W2_TRY(stripes_begin(c, hl, grid_for(N), cap));
        W2_ALLOC(heads, uint64_t, (uint64_t)hl.L.k * hl.L.cap);
        if (S) LAUNCH_L(c, s, "k_seg_list", k_seg_list, dim3(grid_for(N)), dim3(256), S, (const LId*)s.nxtL, (const uint32_t*)s.own, (const unsigned long long*)s.rankw, hl.L, heads);
        W2_TRY(stripes_counts(c, hl, true));
        if (!hl.overflow) break;
        if (attempt) { c.err = "sharded graph: chain list overflow after resizing"; return W2RAP_E_LIMIT; }
        c.release(heads);
        cap = hl.max_wanted + 64;
    }
    const unsigned long long nch = hl.total;
    if (nch > bound) { c.err = "sharded graph: more local chains than chain ends"; return W2RAP_E_GRAPH; }
    for (void* p : {(void*)s.seg_head, (void*)s.seg_len, (void*)s.seg_next}) if (p) c.release(p);
    W2_ALLOC(s.seg_head, Id, 2 * nch + 2); W2_ALLOC(s.seg_len, uint32_t, 2 * nch + 2); W2_ALLOC(s.seg_next, uint64_t, 2 * nch + 2);
    if (nch) LAUNCH_L(c, s, "k_seg_number", k_seg_number, stripes_grid(hl, 256), dim3(256), hl.L, (const uint64_t*)hl.d_pre, (const uint64_t*)heads, (const uint32_t*)s.own, s.rankw,
                      s.seg_head, s.seg_len);
    W2_HIP(hipStreamSynchronize(st));
    W2_HIP(hipGetLastError());
    c.release(heads);
    stripes_free(c, hl);
    s.nseg = 2 * nch;
    s.h_small[0] = s.nseg;
    std::memset(x, 0, sizeof(*x));
    x->op = W2RAP_X_ALLGATHER_HOST; x->elem_bytes = 8; x->send = s.h_small; x->send_count[0] = 1;
    return 0;
}

static int level2_walk1(Ctx& c, Shard& s, w2rap_xchg* x);
static int level2_walk2(Ctx& c, Shard& s, w2rap_xchg* x);
static int level2_done(Ctx& c, Shard& s, w2rap_xchg* x, bool circles);
static int heads_and_stream(Ctx& c, Shard& s, w2rap_xchg* x);

static int shard_step(Ctx& c, w2rap_xchg* x);
int shard_next(Ctx& c, w2rap_xchg* x) {
    if (!c.shard) { c.err = "shard_next before shard_begin"; return W2RAP_E_STATE; }
    static const bool trace = getenv("W2RAP_TRACE_SHARD") != nullptr;
    if (!trace) return shard_step(c, x);
    const int ph = sh(c).phase;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = shard_step(c, x);
    (void)hipStreamSynchronize(c.stream);
    fprintf(stderr, "[w2rap] shard phase %d: %.2f ms -> exchange %u (%u-byte items)\n", ph, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
            (unsigned)x->op, (unsigned)x->elem_bytes);
    return rc;
}
static int shard_step(Ctx& c, w2rap_xchg* x) {
    Shard& s = sh(c);
    hipStream_t st = c.stream;
    const uint64_t S = s.S, N = 2 * S;
    std::memset(x, 0, sizeof(*x));
    switch (s.phase) {
    case PH_BEGIN: {
        W2_TRY(prune_emit(c, s, x));
        s.phase = PH_A_ANSWER;
        return 0;
    }
    case PH_A_ANSWER: {                                                   // the other owners' questions about MY k-mers
        if (s.resp) c.release(s.resp);
        W2_ALLOC(s.resp, uint64_t, s.recv_total);
        if (s.recv_total) LAUNCH(c, "k_answer_member", k_answer_member, dim3(grid_for(s.recv_total)), dim3(256), 0, s.recv_total, (const uint64_t*)s.recv, c.d_table, c.tcap - 1,
                                 c.d_shi, c.d_slo, s.M.base[s.M.me], s.resp);
        W2_HIP(hipStreamSynchronize(st));
        respond(s, x);
        s.phase = PH_A_APPLY;
        return 0;
    }
    case PH_A_APPLY: {
        if (s.nq != s.recv_total) { c.err = "sharded prune: the answers do not match the questions"; return W2RAP_E_STATE; }
        if (s.nq) LAUNCH(c, "k_apply_member", k_apply_member, dim3(grid_for(s.nq)), dim3(256), 0, s.nq, (const uint64_t*)s.q_tag, (const uint64_t*)s.recv, c.d_sctx, s.nxtG);
        // the dictionary has done its work; B: contexts of single neighbours on other ranks
        if (c.d_table) { c.release(c.d_table); c.d_table = nullptr; }
        StripeList ql;
        const uint64_t blocks = grid_for(S);
        uint64_t cap = (S / 4 + 4096) / std::max<uint64_t>(1, std::min<uint64_t>(NSTRIPE, blocks)) + 64;
        uint64_t *q_tag = nullptr, *q_p0 = nullptr;
        for (int attempt = 0;; ++attempt) {
            W2_TRY(stripes_begin(c, ql, blocks, cap));
            W2_ALLOC(q_tag, uint64_t, (uint64_t)ql.L.k * ql.L.cap); W2_ALLOC(q_p0, uint64_t, (uint64_t)ql.L.k * ql.L.cap);
            if (S) LAUNCH(c, "k_prune_final", k_prune_final, dim3(grid_for(S)), dim3(256), 0, S, (const uint8_t*)c.d_sctx, s.nxtG, s.M, ql.L, q_tag, q_p0);
            W2_TRY(stripes_counts(c, ql, false));
            if (!ql.overflow) break;
            if (attempt) { c.err = "sharded prune: context query list overflow after resizing"; return W2RAP_E_LIMIT; }
            c.release(q_tag); c.release(q_p0);
            cap = ql.max_wanted + 64;                                     // (k_prune_final is idempotent: a second pass rewrites the same words)
        }
        W2_TRY(route(c, s, ql, q_tag, q_p0, nullptr, x));
        c.release(q_tag); c.release(q_p0);
        stripes_free(c, ql);
        s.phase = PH_B_ANSWER;
        return 0;
    }
    case PH_B_ANSWER: {
        if (s.resp) c.release(s.resp);
        W2_ALLOC(s.resp, uint64_t, s.recv_total);
        if (s.recv_total) LAUNCH(c, "k_answer_ctx", k_answer_ctx, dim3(grid_for(s.recv_total)), dim3(256), 0, s.recv_total, (const uint64_t*)s.recv, (const uint8_t*)c.d_sctx,
                                 s.M.base[s.M.me], S, s.resp);
        W2_HIP(hipStreamSynchronize(st));
        respond(s, x);
        s.phase = PH_B_APPLY;
        return 0;
    }
    case PH_B_APPLY: {
        if (s.nq != s.recv_total) { c.err = "sharded links: the answers do not match the questions"; return W2RAP_E_STATE; }
        W2_ALLOC(s.nctx, uint8_t, N + 4);
        W2_HIP(hipMemsetAsync(s.nctx, 0, N + 4, st));
        if (s.nq) LAUNCH(c, "k_apply_ctx", k_apply_ctx, dim3(grid_for(s.nq)), dim3(256), 0, s.nq, (const uint64_t*)s.q_tag, (const uint64_t*)s.recv, s.nctx);
        if (s.local32) { uint32_t* q = nullptr; W2_ALLOC(q, uint32_t, N); s.nxtL = q; } else { uint64_t* q = nullptr; W2_ALLOC(q, uint64_t, N); s.nxtL = q; }
        if (S) LAUNCH_L(c, s, "k_links_shard", k_links_shard, dim3(grid_for(S)), dim3(256), S, c.d_shi, c.d_slo, (const uint8_t*)c.d_sctx, (const uint8_t*)s.nctx, s.M, s.nxtG, (LId*)s.nxtL);
        W2_HIP(hipStreamSynchronize(st));
        c.release(s.nctx); s.nctx = nullptr;
        W2_TRY(links_and_segments(c, s, x));
        s.phase = PH_SEGBASE;
        return 0;
    }
    case PH_SEGBASE: {                                                    // every rank's segment count has arrived (host words)
        const uint64_t* cnt = s.recv_host;
        s.segbase[0] = 0;
        for (unsigned r = 0; r < s.M.world; ++r) s.segbase[r + 1] = s.segbase[r] + cnt[r];
        s.NS = s.segbase[s.M.world];
        if (s.NS >= (1ull << 33) - 2) { c.err = "more than 2^33 chain segments (rank words hold 33-bit numbers)"; return W2RAP_E_LIMIT; }
        StripeList ql;
        const uint64_t blocks = grid_for(s.nseg);
        W2_TRY(stripes_begin(c, ql, blocks, 256));                         // (a block appends one query per segment at most: its sub-list takes every kb-th block)
        ql.L.cap = ((blocks + ql.L.k - 1) / ql.L.k) * 256;
        uint64_t *q_tag = nullptr, *q_p0 = nullptr;
        W2_ALLOC(q_tag, uint64_t, (uint64_t)ql.L.k * ql.L.cap); W2_ALLOC(q_p0, uint64_t, (uint64_t)ql.L.k * ql.L.cap);
        if (s.nseg) LAUNCH(c, "k_seg_queries", k_seg_queries, dim3(grid_for(s.nseg)), dim3(256), 0, s.nseg, (const Id*)s.seg_head, (const uint32_t*)s.own,
                           (const unsigned long long*)s.rankw, (const Id*)s.nxtG, s.M, ql.L, q_tag, q_p0, s.seg_next);
        W2_TRY(stripes_counts(c, ql, false));
        if (ql.overflow) { c.err = "sharded segments: query list overflow"; return W2RAP_E_STATE; }
        W2_TRY(route(c, s, ql, q_tag, q_p0, nullptr, x));
        c.release(q_tag); c.release(q_p0);
        stripes_free(c, ql);
        s.phase = PH_C_ANSWER;
        return 0;
    }
    case PH_C_ANSWER: {
        if (s.resp) c.release(s.resp);
        W2_ALLOC(s.resp, uint64_t, s.recv_total);
        if (s.recv_total) LAUNCH_L(c, s, "k_answer_seg", k_answer_seg, dim3(grid_for(s.recv_total)), dim3(256), s.recv_total, (const uint64_t*)s.recv, (const unsigned long long*)s.rankw,
                                   2 * s.M.base[s.M.me], N, s.segbase[s.M.me], s.resp);
        W2_HIP(hipStreamSynchronize(st));
        respond(s, x);
        s.phase = PH_C_APPLY;
        return 0;
    }
    case PH_C_APPLY: {
        if (s.nq != s.recv_total) { c.err = "sharded segments: the answers do not match the questions"; return W2RAP_E_STATE; }
        if (s.nq) LAUNCH(c, "k_apply_seg", k_apply_seg, dim3(grid_for(s.nq)), dim3(256), 0, s.nq, (const uint64_t*)s.q_tag, (const uint64_t*)s.recv, s.seg_next);
        if (s.seg_w) c.release(s.seg_w);
        W2_ALLOC(s.seg_w, unsigned long long, s.nseg + 1);
        if (s.nseg) LAUNCH(c, "k_seg_words", k_seg_words, dim3(grid_for(s.nseg)), dim3(256), 0, s.nseg, (const uint32_t*)s.seg_len, (const uint64_t*)s.seg_next, s.segbase[s.M.me], s.seg_w);
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        x->op = W2RAP_X_ALLGATHER; x->elem_bytes = 8; x->send = s.seg_w; x->send_count[0] = s.nseg;
        s.phase = PH_LEVEL2;
        return 0;
    }
    case PH_LEVEL2: {                                                     // the words of every rank's segments: splitters, this rank's first walks
        W2_TRY(level2_walk1(c, s, x));
        s.phase = PH_L2_JUMP;
        return 0;
    }
    case PH_L2_JUMP: {                                                    // every rank's head and splitter records: the splitters ranked, this rank's second walks
        W2_TRY(level2_walk2(c, s, x));
        s.phase = PH_L2_RESULTS;
        return 0;
    }
    case PH_L2_RESULTS: {                                                 // what the walks of all ranks found for MY segments; is any of them on a circle?
        if (s.recv_total) LAUNCH(c, "k_l2_results", k_l2_results, dim3(grid_for(s.recv_total)), dim3(256), 0, s.recv_total, (const uint64_t*)s.recv, s.Fend, s.T);
        W2_HIP(hipMemsetAsync(s.d_flags, 0, 32, st));
        if (s.NS) LAUNCH(c, "k_seg_finish", k_seg_finish, dim3(grid_for(s.NS)), dim3(256), 0, s.NS, (const unsigned long long*)s.w2o, s.lenS);
        if (s.nseg) LAUNCH(c, "k_seg_circles", k_seg_circles, dim3(grid_for(s.nseg)), dim3(256), 0, s.nseg, s.segbase[s.M.me], (const uint64_t*)s.Fend, s.T, s.cyc2, s.d_flags);
        uint32_t h_flags[4] = {0, 0, 0, 0};
        W2_HIP(hipMemcpyAsync(h_flags, s.d_flags, 16, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        for (void* p : {(void*)s.q_tag, (void*)s.q_send}) if (p) c.release(p);
        s.q_tag = s.q_send = nullptr;
        s.h_small[0] = h_flags[2];
        x->op = W2RAP_X_ALLGATHER_HOST; x->elem_bytes = 8; x->send = s.h_small; x->send_count[0] = 1;
        s.phase = PH_L2_CIRCLES;
        return 0;
    }
    case PH_L2_CIRCLES: {                                                 // every rank's "a segment of mine lies on a circle"
        bool circles = false;
        for (unsigned r = 0; r < s.M.world; ++r) circles = circles || s.recv_host[r] != 0;
        W2_TRY(level2_done(c, s, x, circles));
        s.phase = circles ? PH_CIRC_MIN : PH_HEADS;
        return 0;
    }
    case PH_CIRC_MIN: {                                                   // every chain's minimum k-mer has arrived: the circles' minima, the cuts
        const uint64_t NC = s.NS / 2;
        const MinRec* mr = (const MinRec*)s.recv;
        uint64_t *nx, *mn, *nx2, *mn2;
        W2_ALLOC(nx, uint64_t, s.NS); W2_ALLOC(mn, uint64_t, s.NS); W2_ALLOC(nx2, uint64_t, s.NS); W2_ALLOC(mn2, uint64_t, s.NS);
        (void)NC;
        LAUNCH(c, "k_segmin_init", k_segmin_init, dim3(grid_for(s.NS)), dim3(256), 0, s.NS, (const unsigned long long*)s.w2o, mr, nx, mn);
        for (int round = 0; round < 34; ++round) {
            LAUNCH(c, "k_segmin_jump", k_segmin_jump, dim3(grid_for(s.NS)), dim3(256), 0, s.NS, mr, (const uint64_t*)nx, (const uint64_t*)mn, nx2, mn2);
            std::swap(nx, nx2); std::swap(mn, mn2);
        }
        unsigned long long* d_n = nullptr;
        W2_ALLOC(d_n, unsigned long long, 1);
        W2_HIP(hipMemsetAsync(d_n, 0, 8, st));
        if (s.cuts) c.release(s.cuts);
        W2_ALLOC(s.cuts, uint64_t, s.nseg + 1);
        if (s.nseg) LAUNCH(c, "k_seg_cuts", k_seg_cuts, dim3(grid_for(s.nseg)), dim3(256), 0, s.nseg, s.segbase[s.M.me], (const uint8_t*)s.cyc2, (const uint64_t*)mn, mr, s.M, s.nxtG,
                           d_n, s.nseg + 1, s.cuts);
        unsigned long long ncut = 0;
        W2_HIP(hipMemcpyAsync(&ncut, d_n, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        c.release(nx); c.release(mn); c.release(nx2); c.release(mn2); c.release(d_n);
        x->op = W2RAP_X_ALLGATHER; x->elem_bytes = 8; x->send = s.cuts; x->send_count[0] = ncut;
        s.phase = PH_CIRC_CUT;
        return 0;
    }
    case PH_CIRC_CUT: {                                                   // the cut links of every rank: apply mine, rank again
        if (s.recv_total) LAUNCH(c, "k_apply_cuts", k_apply_cuts, dim3(grid_for(s.recv_total)), dim3(256), 0, s.recv_total, (const uint64_t*)s.recv, s.M, s.nxtG);
        if (S) LAUNCH_L(c, s, "k_local_links", k_local_links, dim3(grid_for(N)), dim3(256), S, (const Id*)s.nxtG, s.M, (LId*)s.nxtL);
        W2_HIP(hipStreamSynchronize(st));
        if (++s.circle_rounds > 1) { c.err = "failed to close circle (BuildReadQGraph.cc:141)"; return W2RAP_E_GRAPH; }
        W2_TRY(links_and_segments(c, s, x));
        s.phase = PH_SEGBASE;
        return 0;
    }
    case PH_HEADS: {                                                      // the middle bases of every rank are summed
        W2_TRY(heads_and_stream(c, s, x));
        s.phase = PH_STREAM;
        return 0;
    }
    case PH_STREAM: {                                                     // the edge stream is complete: everything E-sized, replicated
        if (c.edge_bases) LAUNCH(c, "k_unpack_codes", k_unpack_codes, dim3(grid_for(c.edge_bases)), dim3(256), 0, c.edge_bases, (const uint32_t*)s.bits, c.d_edge_codes);
        W2_HIP(hipStreamSynchronize(st));
        s.bits_keep = s.bits; s.bits = nullptr;                           // (owned by the context from here on: c.d_edge_bits)
        for (void* p : {(void*)s.nxtG, (void*)s.nxtL, (void*)s.rankw, (void*)s.own, (void*)s.seg_head, (void*)s.seg_len, (void*)s.seg_next, (void*)s.seg_w, (void*)s.R, (void*)s.hr_of_seg, (void*)s.w2,
                        (void*)s.w2o, (void*)s.Fend, (void*)s.T, (void*)s.cyc2, (void*)s.mid, (void*)s.lenS, (void*)s.edge_of_head, (void*)s.q_tag, (void*)s.q_send, (void*)s.resp,
                        (void*)s.d_flags, (void*)s.cuts})
            if (p) c.release(p);
        s.nxtG = nullptr; s.nxtL = nullptr; s.rankw = nullptr; s.own = nullptr; s.seg_head = nullptr; s.seg_len = nullptr; s.seg_next = nullptr; s.seg_w = nullptr; s.R = nullptr; s.hr_of_seg = nullptr;
        s.w2 = s.w2o = nullptr; s.Fend = s.T = nullptr; s.cyc2 = s.mid = nullptr; s.lenS = nullptr; s.edge_of_head = nullptr; s.q_tag = s.q_send = s.resp = nullptr;
        s.d_flags = nullptr; s.cuts = nullptr;
        c.counted = true;
        // the packed stream stays as it was summed (+ 16 bytes of slack behind it, which the word array has); the index entries of this rank's
        // share of its positions are listed and gathered -- every rank then inserts all of them: the window minima, which are the cost of the
        // index, are computed once per position in the JOB, not once per rank
        c.d_edge_bits = reinterpret_cast<uint8_t*>(s.bits_keep);
        c.bits_ready = true;
        if (s.idx_list) { c.release(s.idx_list); s.idx_list = nullptr; }
        // the absence filter the same way -- every rank scans the stream and sets the bits of ITS range of words, the ranges are gathered --, started
        // now on the side stream: it runs beside the listing, the gathering and the insertion of the index entries
        if (s.flt_slice) { c.release(s.flt_slice); s.flt_slice = nullptr; }
        s.flt_words = 0;
        if (filter32_words(c)) W2_TRY(filter32_slice(c, s.M.me, s.M.world, &s.flt_slice, &s.flt_words));
        uint64_t n_list = 0;
        W2_TRY(index_entries_slice(c, s.M.me, s.M.world, &s.idx_list, &n_list));
        x->op = W2RAP_X_ALLGATHER; x->elem_bytes = 16; x->send = s.idx_list; x->send_count[0] = n_list;
        s.phase = PH_INDEX;
        return 0;
    }
    case PH_INDEX: {
        // every rank inserts all entries; which of them belong to keys with many entries (repeat boundaries: common.h, the exact table) is found out
        // for ITS part of the gathered list by every rank, and those entries are gathered in their turn
        W2_TRY(index_from_entries(c, (const uint4*)s.recv, s.recv_total, false));
        W2_HIP(hipStreamSynchronize(st));
        if (s.idx_list) { c.release(s.idx_list); s.idx_list = nullptr; }
        uint64_t n_hard = 0;
        W2_TRY(index_hard_slice(c, (const uint4*)s.recv, s.recv_total, s.M.me, s.M.world, &s.idx_list, &n_hard));
        if (s.recv) { c.release(s.recv); s.recv = nullptr; }
        x->op = W2RAP_X_ALLGATHER; x->elem_bytes = 16; x->send = s.idx_list; x->send_count[0] = n_hard;
        s.phase = PH_INDEX_HARD;
        return 0;
    }
    case PH_INDEX_HARD: {
        W2_TRY(index_hard_apply(c, (const uint4*)s.recv, s.recv_total));
        W2_HIP(hipStreamSynchronize(st));
        if (s.idx_list) { c.release(s.idx_list); s.idx_list = nullptr; }
        if (s.recv) { c.release(s.recv); s.recv = nullptr; }
        if (s.flt_slice) {                                                // this rank's range of the filter's words (started in the step before)
            if (c.stream2) W2_HIP(hipStreamSynchronize(c.stream2));
            x->op = W2RAP_X_ALLGATHER; x->elem_bytes = 8; x->send = s.flt_slice; x->send_count[0] = s.flt_words;
            s.phase = PH_FILTER;
            return 0;
        }
        W2_TRY(graph_finish(c));
        s.phase = PH_DONE;
        x->op = W2RAP_X_DONE;
        return 0;
    }
    case PH_FILTER: {
        const uint64_t fw = filter32_words(c);
        if (s.recv_total != fw) { c.err = "sharded graph: gathered filter has the wrong size"; return W2RAP_E_STATE; }
        if (c.d_filter32) { c.release(c.d_filter32); c.d_filter32 = nullptr; }
        c.d_filter32 = (unsigned long long*)s.recv; s.recv = nullptr;          // the gathered words ARE the filter
        c.f32words = fw; c.filter_prebuilt = true;
        if (s.flt_slice) { c.release(s.flt_slice); s.flt_slice = nullptr; }
        W2_TRY(graph_finish(c));
        s.phase = PH_DONE;
        x->op = W2RAP_X_DONE;
        return 0;
    }
    default:
        x->op = W2RAP_X_DONE;
        return 0;
    }
}

static void level2_release(Ctx& c, Shard& s) {
    for (void* p : {(void*)s.sp, (void*)s.spl, (void*)s.spl_me, (void*)s.l2_off, (void*)s.l2_steps, (void*)s.l2_send}) if (p) c.release(p);
    s.sp = nullptr; s.spl = s.spl_me = s.l2_off = nullptr; s.l2_steps = nullptr; s.l2_send = nullptr;
}
// the gathered segment words: the splitters (every rank alike), the first walk from this rank's own; -> all-gather of the head and splitter records
static int level2_walk1(Ctx& c, Shard& s, w2rap_xchg* x) {
    hipStream_t st = c.stream;
    const uint64_t NS = s.NS;
    if (s.recv_total != NS) { c.err = "sharded graph: gathered " + std::to_string(s.recv_total) + " segment words, expected " + std::to_string(NS); return W2RAP_E_STATE; }
    for (void* p : {(void*)s.R, (void*)s.hr_of_seg, (void*)s.w2, (void*)s.w2o, (void*)s.Fend, (void*)s.T, (void*)s.cyc2, (void*)s.mid, (void*)s.lenS}) if (p) c.release(p);
    s.R = nullptr; s.hr_of_seg = nullptr;
    level2_release(c, s);
    s.w2o = (unsigned long long*)s.recv; s.recv = nullptr;                 // the gathered words ARE the lists (+ 64 bytes of slack behind them)
    W2_ALLOC(s.w2, unsigned long long, NS + 1); W2_ALLOC(s.Fend, uint64_t, NS + 1); W2_ALLOC(s.T, uint64_t, NS + 1);
    W2_ALLOC(s.cyc2, uint8_t, NS + 4); W2_ALLOC(s.mid, uint8_t, NS + 4); W2_ALLOC(s.lenS, uint32_t, NS + 1); W2_ALLOC(s.hr_of_seg, uint32_t, NS + 1);
    W2_ALLOC(s.sp, uint8_t, NS + 4);
    W2_HIP(hipMemsetAsync(s.cyc2, 0, NS + 4, st));
    W2_HIP(hipMemsetAsync(s.hr_of_seg, 0, (NS + 1) * 4, st));
    unsigned long long* d_n = nullptr;
    W2_ALLOC(d_n, unsigned long long, 4);
    {   // the splitters: ~1/64 sampled + the heads, listed striped, then packed (an overflow is followed by the exact sizes)
        StripeList la, lm;
        const uint64_t blocks = grid_for(NS), kk = std::max<uint64_t>(1, std::min<uint64_t>(NSTRIPE, blocks));
        uint64_t cap = (NS / 32 + NS / 16) / kk + 256, cap_me = (s.nseg / 32 + s.nseg / 16) / kk + 256;
        uint64_t *a = nullptr, *m = nullptr;
        for (int attempt = 0;; ++attempt) {
            W2_TRY(stripes_begin(c, la, blocks, cap)); W2_TRY(stripes_begin(c, lm, blocks, cap_me));
            W2_ALLOC(a, uint64_t, (uint64_t)la.L.k * la.L.cap); W2_ALLOC(m, uint64_t, (uint64_t)lm.L.k * lm.L.cap);
            if (NS) LAUNCH(c, "k_seg_mark", k_seg_mark, dim3(grid_for(NS)), dim3(256), 0, NS, (const unsigned long long*)s.w2o, s.sp, la.L, a, s.Fend, s.segbase[s.M.me],
                           s.segbase[s.M.me + 1], lm.L, m);
            W2_TRY(stripes_counts(c, la, true)); W2_TRY(stripes_counts(c, lm, true));
            if (!la.overflow && !lm.overflow) break;
            if (attempt) { c.err = "sharded graph: splitter list overflow after resizing"; return W2RAP_E_LIMIT; }
            c.release(a); c.release(m);
            cap = la.max_wanted + 64; cap_me = lm.max_wanted + 64;
        }
        s.nspl = la.total; s.nspl_me = lm.total;
        W2_ALLOC(s.spl, uint64_t, s.nspl + 1); W2_ALLOC(s.spl_me, uint64_t, s.nspl_me + 1);
        if (s.nspl) LAUNCH(c, "k_stripes_compact", k_stripes_compact, stripes_grid(la, 256), dim3(256), 0, la.L, (const uint64_t*)la.d_pre, (const uint64_t*)a, s.spl);
        if (s.nspl_me) LAUNCH(c, "k_stripes_compact", k_stripes_compact, stripes_grid(lm, 256), dim3(256), 0, lm.L, (const uint64_t*)lm.d_pre, (const uint64_t*)m, s.spl_me);
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        c.release(a); c.release(m);
        stripes_free(c, la); stripes_free(c, lm);
    }
    unsigned long long h_n[4] = {0, 0, 0, 0};
    // records: [0, nspl_me) the own splitters' first walks, behind them the own chain heads
    W2_ALLOC(s.l2_send, L2Rec, s.nspl_me + s.nseg + 1);
    W2_ALLOC(s.l2_steps, uint32_t, s.nspl_me + 1);
    if (s.nspl_me) LAUNCH(c, "k_seg_walk1", k_seg_walk1, dim3(grid_for(s.nspl_me)), dim3(256), 0, s.nspl_me, (const uint64_t*)s.spl_me, (const unsigned long long*)s.w2o,
                          (const uint8_t*)s.sp, s.l2_send, s.l2_steps, NS + 1);
    W2_HIP(hipMemsetAsync(d_n, 0, 8, st));
    if (s.nseg) LAUNCH(c, "k_head_recs", k_head_recs, dim3(grid_for(s.nseg)), dim3(256), 0, s.nseg, (const Id*)s.seg_head, (const uint64_t*)s.seg_next, c.d_shi, c.d_slo,
                       2 * s.M.base[s.M.me], s.segbase[s.M.me], s.l2_send + s.nspl_me, d_n);
    W2_HIP(hipMemcpyAsync(h_n, d_n, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    W2_HIP(hipGetLastError());
    c.release(d_n);
    std::memset(x, 0, sizeof(*x));
    x->op = W2RAP_X_ALLGATHER; x->elem_bytes = sizeof(L2Rec); x->send = s.l2_send; x->send_count[0] = s.nspl_me + h_n[0];
    return 0;
}
// the gathered records: the splitters' chains ranked (every rank alike: 1/64 of the segments + the heads), the second walk from this rank's own;
// -> all-to-all of the results to the segments' owners
static int level2_walk2(Ctx& c, Shard& s, w2rap_xchg* x) {
    hipStream_t st = c.stream;
    if (s.recv_total >= (1ull << 32)) { c.err = "more than 2^32 level-2 records"; return W2RAP_E_LIMIT; }
    s.R = (L2Rec*)s.recv; s.nR = s.recv_total; s.recv = nullptr;
    if (s.nR) LAUNCH(c, "k_l2_apply", k_l2_apply, dim3(grid_for(s.nR)), dim3(256), 0, s.nR, (const L2Rec*)s.R, s.w2, s.Fend, s.T, s.hr_of_seg);
    const uint64_t nspl = s.nspl, nme = s.nspl_me;
    uint64_t total = 0;
    uint64_t *q_tag = nullptr, *q_p0 = nullptr, *q_p1 = nullptr;
    if (nspl) {
        for (int round = 0; round < 12; ++round) {                         // the splitter chains: log_17 launches
            W2_HIP(hipMemsetAsync(s.d_flags, 0, 4, st));
            LAUNCH(c, "k_seg_jump", k_seg_jump, dim3(grid_for(nspl)), dim3(256), 0, nspl, (const uint64_t*)s.spl, s.w2, s.d_flags);
            uint32_t changed = 0;
            W2_HIP(hipMemcpyAsync(&changed, s.d_flags, 4, hipMemcpyDeviceToHost, st));
            W2_HIP(hipStreamSynchronize(st));
            if (!changed) break;
        }
        uint64_t *Fsp = nullptr, *Tsp = nullptr;
        W2_ALLOC(Fsp, uint64_t, nspl + 1); W2_ALLOC(Tsp, uint64_t, nspl + 1);
        LAUNCH(c, "k_seg_splitters_done", k_seg_splitters_done, dim3(grid_for(nspl)), dim3(256), 0, nspl, (const uint64_t*)s.spl, (const unsigned long long*)s.w2, (const uint64_t*)s.Fend,
               (const uint64_t*)s.T, Fsp, Tsp);
        // this rank's second walks read the END splitters' first-walk values: before the store
        W2_ALLOC(s.l2_off, uint64_t, nme + 2);
        W2_TRY(exclusive_scan_u32_to_u64(c, s.l2_steps, s.l2_off, nme));
        W2_HIP(hipMemcpyAsync(&total, s.l2_off + nme, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        W2_ALLOC(q_tag, uint64_t, total + 1); W2_ALLOC(q_p0, uint64_t, total + 1); W2_ALLOC(q_p1, uint64_t, total + 1);
        SegMap G{};
        for (unsigned r = 0; r < 65; ++r) G.b[r] = s.segbase[r <= s.M.world ? r : s.M.world];
        G.world = s.M.world; G.me = s.M.me;
        if (nme) LAUNCH(c, "k_seg_walk2", k_seg_walk2, dim3(grid_for(nme)), dim3(256), 0, nme, (const uint64_t*)s.spl_me, (const unsigned long long*)s.w2o, (const uint8_t*)s.sp,
                        (const unsigned long long*)s.w2, (const uint64_t*)s.Fend, (const uint64_t*)s.T, (const uint64_t*)s.l2_off, G, q_tag, q_p0, q_p1);
        LAUNCH(c, "k_seg_splitters_store", k_seg_splitters_store, dim3(grid_for(nspl)), dim3(256), 0, nspl, (const uint64_t*)s.spl, (const uint64_t*)Fsp, (const uint64_t*)Tsp, s.Fend, s.T);
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        c.release(Fsp); c.release(Tsp);
    }
    {
        StripeList dl;
        W2_TRY(stripes_dense(c, dl, total));
        W2_TRY(route(c, s, dl, q_tag, q_p0, q_p1, x));
        stripes_free(c, dl);
    }
    for (void* p : {(void*)q_tag, (void*)q_p0, (void*)q_p1}) if (p) c.release(p);
    level2_release(c, s);
    return 0;
}
// no circle crosses ranks: the middle bases; else the circles' minima
static int level2_done(Ctx& c, Shard& s, w2rap_xchg* x, bool circles) {
    hipStream_t st = c.stream;
    const uint64_t NS = s.NS, S = s.S;
    std::memset(x, 0, sizeof(*x));
    if (circles) {
        // the minimum k-mer of every local chain on a circle -> all-gather (chain ch of rank r is entry segbase[r] / 2 + ch)
        const uint64_t nch = s.nseg / 2;
        if (s.minrec) c.release(s.minrec);
        W2_ALLOC(s.minrec, MinRec, nch + 1);
        if (nch) LAUNCH_L(c, s, "k_seg_min", k_seg_min, dim3(grid_for(nch)), dim3(256), nch, (const Id*)s.seg_head, (const uint32_t*)s.seg_len, (const LId*)s.nxtL,
                          (const uint8_t*)(s.cyc2 + s.segbase[s.M.me]), c.d_shi, c.d_slo, s.M.base[s.M.me], s.minrec);
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        x->op = W2RAP_X_ALLGATHER; x->elem_bytes = sizeof(MinRec); x->send = s.minrec; x->send_count[0] = nch;
        if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] shard %u: a circle crosses ranks\n", s.M.me);
        return 0;
    }
    // the middle bases of the odd-length unipaths, written by whoever holds the middle k-mer; summed over the ranks
    W2_HIP(hipMemsetAsync(s.mid, 0, NS + 4, st));
    if (S) LAUNCH_L(c, s, "k_mid_shard", k_mid_shard, dim3(grid_for(S)), dim3(256), S, c.d_shi, c.d_slo, (const uint32_t*)s.own, (const unsigned long long*)s.rankw, s.segbase[s.M.me],
                    (const uint32_t*)s.lenS, (const uint64_t*)s.Fend, (const uint64_t*)s.T, s.mid);
    W2_HIP(hipStreamSynchronize(st));
    W2_HIP(hipGetLastError());
    x->op = W2RAP_X_ALLREDUCE_U8; x->elem_bytes = 1; x->send = s.mid; x->send_count[0] = (NS + 3) & ~3ull;
    return 0;
}

static int heads_and_stream(Ctx& c, Shard& s, w2rap_xchg* x) {             // canonical heads, unipath order, offsets (replicated); this rank's bases
    hipStream_t st = c.stream;
    const uint64_t NS = s.NS, S = s.S;
    unsigned long long* d_nheads = nullptr;
    W2_ALLOC(d_nheads, unsigned long long, 1);
    W2_HIP(hipMemsetAsync(d_nheads, 0, 8, st));
    W2_HIP(hipMemsetAsync(s.d_flags, 0, 32, st));
    const uint64_t head_cap = NS / 2 + 1;                                  // every chain has two heads, at most one of them canonical
    uint64_t *head_seg, *key_hi, *key_lo, *key_tmp; uint32_t* perm;
    W2_ALLOC(head_seg, uint64_t, head_cap); W2_ALLOC(key_hi, uint64_t, head_cap); W2_ALLOC(key_lo, uint64_t, head_cap);
    if (s.nR) LAUNCH(c, "k_heads_shard", k_heads_shard, dim3(grid_for(s.nR)), dim3(256), 0, s.nR, (const L2Rec*)s.R, (const uint32_t*)s.hr_of_seg, (const uint64_t*)s.Fend,
                     (const uint64_t*)s.T, (const uint8_t*)s.mid, head_seg, key_hi, key_lo, d_nheads, head_cap, s.d_flags);
    unsigned long long E = 0; uint32_t h_flags[4] = {0, 0, 0, 0};
    W2_HIP(hipMemcpyAsync(&E, d_nheads, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipMemcpyAsync(h_flags, s.d_flags, 16, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    W2_HIP(hipGetLastError());
    W2_TRY(shard_error(c, h_flags[1]));
    if (E > head_cap) { c.err = "more canonical heads than chains"; return W2RAP_E_GRAPH; }
    if (E >= (1ull << 31)) { c.err = "more than 2^31 unipaths (edge ids are int, paths/long/ReadPath.h)"; return W2RAP_E_LIMIT; }
    c.E = E;
    W2_ALLOC(perm, uint32_t, E + 1); W2_ALLOC(key_tmp, uint64_t, E + 1);
    if (c.d_edge_nk) c.release(c.d_edge_nk);
    W2_ALLOC(c.d_edge_nk, uint32_t, E + 1);
    if (s.edge_of_head) c.release(s.edge_of_head);
    W2_ALLOC(s.edge_of_head, uint32_t, NS + 1);
    W2_HIP(hipMemsetAsync(s.edge_of_head, 0, (NS + 1) * 4, st));
    if (E) {
        // the heads in the lexicographic order of their first 60-mers (the atomic append above leaves them in any order -- and in a different one
        // on every rank): one sort by the first 30 bases, runs of equal words by the other 30
        LAUNCH(c, "k_iota32", k_iota32, dim3(grid_for(E)), dim3(256), 0, E, perm);
        W2_HIP(hipMemcpyAsync(key_tmp, key_hi, E * 8, hipMemcpyDeviceToDevice, st));
        W2_TRY(sort_pairs_u64(c, key_tmp, perm, E, 0, 60));
        LAUNCH(c, "k_tie_sort", k_tie_sort, dim3(grid_for(E)), dim3(256), 0, E, (const uint64_t*)key_tmp, (const uint64_t*)key_lo, perm);
    }
    if (s.hint) {
        const w2rap_edge_hint* hint = s.hint;
        if (hint->n_edges != E) { c.err = "edge_order_hint has " + std::to_string(hint->n_edges) + " edges, the graph has " + std::to_string(E); return W2RAP_E_HINT; }
        std::vector<uint64_t> hh(E), hl(E);
        for (uint64_t e = 0; e < E; ++e) {
            if (hint->len[e] < K) { c.err = "edge_order_hint: edge shorter than K"; return W2RAP_E_HINT; }
            const uint8_t* p = hint->packed + hint->byte_off[e];
            uint64_t hi = 0, lo = 0;
            for (unsigned t = 0; t < 30; ++t) hi = (hi << 2) | ((p[t >> 2] >> (2 * (t & 3))) & 3);
            for (unsigned t = 30; t < 60; ++t) lo = (lo << 2) | ((p[t >> 2] >> (2 * (t & 3))) & 3);
            hh[e] = hi; hl[e] = lo;
        }
        uint64_t *d_hh, *d_hl; uint32_t* d_hlen;
        W2_ALLOC(d_hh, uint64_t, E + 1); W2_ALLOC(d_hl, uint64_t, E + 1); W2_ALLOC(d_hlen, uint32_t, E + 1);
        if (E) {
            W2_HIP(hipMemcpyAsync(d_hh, hh.data(), E * 8, hipMemcpyHostToDevice, st));
            W2_HIP(hipMemcpyAsync(d_hl, hl.data(), E * 8, hipMemcpyHostToDevice, st));
            W2_HIP(hipMemcpyAsync(d_hlen, hint->len, E * 4, hipMemcpyHostToDevice, st));
            LAUNCH(c, "k_edges_hint", k_edges_hint, dim3(grid_for(E)), dim3(256), 0, E, (const uint64_t*)d_hh, (const uint64_t*)d_hl, (const uint32_t*)d_hlen, (const uint32_t*)perm,
                   (const uint64_t*)key_hi, (const uint64_t*)key_lo, (const uint64_t*)head_seg, (const uint64_t*)s.T, c.d_edge_nk, s.edge_of_head, s.d_flags);
        }
        W2_HIP(hipMemcpyAsync(h_flags, s.d_flags, 16, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        W2_TRY(shard_error(c, h_flags[1]));
        c.release(d_hh); c.release(d_hl); c.release(d_hlen);
    } else if (E) {
        LAUNCH(c, "k_edges_sorted", k_edges_sorted, dim3(grid_for(E)), dim3(256), 0, E, (const uint32_t*)perm, (const uint64_t*)head_seg, (const uint64_t*)s.T, c.d_edge_nk, s.edge_of_head);
    }
    uint32_t* d_elen = nullptr;
    W2_ALLOC(d_elen, uint32_t, E + 1);
    if (c.d_edge_off) c.release(c.d_edge_off);
    W2_ALLOC(c.d_edge_off, uint64_t, E + 1);
    if (E) LAUNCH(c, "k_edge_len2", k_edge_len2, dim3(grid_for(E)), dim3(256), 0, E, (const uint32_t*)c.d_edge_nk, d_elen);
    W2_TRY(exclusive_scan_u32_to_u64(c, d_elen, c.d_edge_off, E));
    W2_HIP(hipMemcpyAsync(&c.edge_bases, c.d_edge_off + E, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    // this rank's bases into a zeroed packed stream; the ranks' streams are then summed (disjoint 2-bit groups: sum = or)
    s.nwords = (c.edge_bases + 15) / 16 + 4;
    W2_ALLOC(s.bits, uint32_t, s.nwords);
    if (c.d_edge_codes) c.release(c.d_edge_codes);
    W2_ALLOC(c.d_edge_codes, uint8_t, c.edge_bases + 64);
    W2_HIP(hipMemsetAsync(c.d_edge_codes, 0, c.edge_bases + 64, st));
    if (S) LAUNCH_L(c, s, "k_assign_shard", k_assign_shard, dim3(grid_for(S)), dim3(256), S, c.d_shi, c.d_slo, (const uint32_t*)s.own, (const unsigned long long*)s.rankw, s.segbase[s.M.me],
                    (const uint32_t*)s.lenS, (const uint64_t*)s.Fend, (const uint64_t*)s.T, (const uint32_t*)s.edge_of_head, (const uint64_t*)c.d_edge_off, c.d_edge_codes, s.d_flags);
    LAUNCH(c, "k_pack_words", k_pack_words, dim3(grid_for(s.nwords)), dim3(256), 0, s.nwords, c.edge_bases, (const uint8_t*)c.d_edge_codes, s.bits);
    W2_HIP(hipMemcpyAsync(h_flags, s.d_flags, 16, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    W2_HIP(hipGetLastError());
    W2_TRY(shard_error(c, h_flags[1]));
    for (void* p : {(void*)d_nheads, (void*)head_seg, (void*)key_hi, (void*)key_lo, (void*)key_tmp, (void*)perm, (void*)d_elen}) c.release(p);
    if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] shard %u: %llu segments job-wide (%.3f of the solid k-mers), %llu unipaths, %llu edge bases\n", s.M.me, (unsigned long long)NS,
                                       s.M.base[s.M.world] ? (double)NS / 2.0 / (double)s.M.base[s.M.world] : 0.0, E, (unsigned long long)c.edge_bases);
    std::memset(x, 0, sizeof(*x));
    x->op = W2RAP_X_ALLREDUCE_U32; x->elem_bytes = 4; x->send = s.bits; x->send_count[0] = s.nwords;
    return 0;
}

// the words of a host all-gather (one u64 per rank)
int shard_host_words(Ctx& c, const uint64_t* words) {
    if (!c.shard) { c.err = "shard_host_words before shard_begin"; return W2RAP_E_STATE; }
    Shard& s = sh(c);
    for (unsigned r = 0; r < s.M.world; ++r) s.recv_host[r] = words[r];
    return 0;
}
// job-wide numbers of the sharded dictionary and what this rank holds of it (tests: the per-rank share)
int shard_info(Ctx& c, uint64_t out[8]) {
    if (!c.shard) { c.err = "shard_info before shard_begin"; return W2RAP_E_STATE; }
    Shard& s = sh(c);
    out[0] = s.S; out[1] = s.M.base[s.M.world]; out[2] = s.nseg; out[3] = s.NS; out[4] = c.E; out[5] = c.edge_bases; out[6] = c.index_entries; out[7] = (uint64_t)s.phase;
    return 0;
}

}  // namespace w2
