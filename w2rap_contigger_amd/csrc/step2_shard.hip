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

struct ShardMap { uint64_t base[65]; uint32_t world, me, NB, per_pass, nbl; };           // owner of bucket b = (b % per_pass) / nbl
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

// ---------------------------------------------------------------------------------------------- generic routing of tagged items
// tag bits 63:58 = destination rank.  k_route_hist counts per destination, k_route_scatter writes item j to out[off[dest] + running cursor]
// (the order inside a destination's block is arbitrary: the responses come back in the order the queries went).
__global__ void __launch_bounds__(256) k_route_hist(uint64_t n, const uint64_t* __restrict__ tag, unsigned long long* __restrict__ hist) {
    __shared__ unsigned s_h[64];
    if (threadIdx.x < 64) s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) atomicAdd(&s_h[tag[j] >> 58], 1u);
    __syncthreads();
    if (threadIdx.x < 64 && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s_h[threadIdx.x]);
}
__global__ void __launch_bounds__(256) k_route_scatter(uint64_t n, const uint64_t* __restrict__ tag, const uint64_t* __restrict__ p0, const uint64_t* __restrict__ p1,
                                                        unsigned long long* __restrict__ cursor /* [64], preset to the block offsets */,
                                                        uint64_t* __restrict__ out_tag, uint64_t* __restrict__ out /* 1 or 2 words per item */) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint64_t t = tag[j];
    const unsigned long long at = atomicAdd(&cursor[t >> 58], 1ull);
    out_tag[at] = t;
    if (p1) { out[2 * at] = p0[j]; out[2 * at + 1] = p1[j]; } else out[at] = p0[j];
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
                                                      unsigned long long* __restrict__ qn, uint64_t qcap, uint64_t* __restrict__ q_tag, uint64_t* __restrict__ q_hi,
                                                      uint64_t* __restrict__ q_lo) {
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
        const int64_t s = table_find(table, mask, shi, slo, nk);
        if (s >= 0) {
            const Id id = pal ? PAL : (Id)(base2 + 2 * (uint64_t)s + (r ? 1u : 0u));
            if (t < 4) ns = id; else np = id;
            continue;
        }
        const unsigned o = M.world > 1 ? owner_of_kmer(M, nk) : M.me;
        if (o == M.me) { c &= ~(1u << t); continue; }
        const unsigned long long at = atomicAdd(qn, 1ull);
        if (at < qcap) {
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
                                                      unsigned long long* __restrict__ qn, uint64_t qcap, uint64_t* __restrict__ q_tag, uint64_t* __restrict__ q_p0) {
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
            const unsigned long long at = atomicAdd(qn, 1ull);
            if (at < qcap) { q_tag[at] = ((uint64_t)rank_of_index(M, g >> 1) << 58) | (2 * i + d); q_p0[at] = g >> 1; }
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
    auto local = [&](Id g) -> LId { return g != NONE && (g >> 1) >= lo && (g >> 1) < hi ? (LId)(g - 2 * lo) : NodeId<LId>::NONE; };
    nxtL[2 * i] = local(n0); nxtL[2 * i + 1] = local(n1);
}
template <class LId>
__global__ void __launch_bounds__(256) k_local_links(uint64_t S, const Id* __restrict__ nxtG, ShardMap M, LId* __restrict__ nxtL) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= 2 * S) return;
    const uint64_t lo = M.base[M.me], hi = M.base[M.me + 1];
    const Id g = nxtG[v];
    nxtL[v] = g != NONE && (g >> 1) >= lo && (g >> 1) < hi ? (LId)(g - 2 * lo) : NodeId<LId>::NONE;
}

// ---------------------------------------------------------------------------------------------- segments (level 1 -> level 2)
// A local chain has two heads, v and the flip of its other end: the smaller one numbers the pair (2c, 2c + 1).  The number of the segment
// whose head is the flip of a local chain end t rides in the distance field of t's own rank word (as the unipath number does on one GPU).
template <class LId>
__global__ void __launch_bounds__(256) k_seg_number(uint64_t S, const LId* __restrict__ nxtL, const uint32_t* __restrict__ own, unsigned long long* __restrict__ w,
                                                     unsigned long long* __restrict__ nchains, uint64_t cap, Id* __restrict__ seg_head, uint32_t* __restrict__ seg_len) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= 2 * S || nxtL[v ^ 1] != NodeId<LId>::NONE) return;         // not a local head
    LId t; uint32_t d;
    rank_of<LId>(own, w, (LId)v, t, d);
    const uint64_t u = (uint64_t)t ^ 1ull;                                // the other head
    if (u < v) return;                                                    // (u != v: a chain never runs from a node to its own flip)
    const unsigned long long ch = atomicAdd(nchains, 1ull);
    if (ch >= cap) return;
    seg_head[2 * ch] = (Id)v; seg_head[2 * ch + 1] = (Id)u;
    seg_len[2 * ch] = d + 1; seg_len[2 * ch + 1] = d + 1;
    // t is the end of segment 2c and the flip of the head of 2c+1; v^1 is the end of 2c+1 and the flip of the head of 2c
    __hip_atomic_store(&w[t], RankW<LId>::pack(2 * ch + 2, t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&w[v ^ 1], RankW<LId>::pack(2 * ch + 1, (LId)(v ^ 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// local segment of the head whose flip is the local chain end t
template <class LId>
__device__ inline uint64_t seg_of_end_flip(const unsigned long long* __restrict__ w, uint64_t t) { return RankW<LId>::dist(w[t]) - 1; }
// the C queries: the segment a chain continues into on another rank
__global__ void __launch_bounds__(256) k_seg_queries(uint64_t nseg, const Id* __restrict__ seg_head, const uint32_t* __restrict__ own, const unsigned long long* __restrict__ w,
                                                      const Id* __restrict__ nxtG, ShardMap M, unsigned long long* __restrict__ qn, uint64_t qcap,
                                                      uint64_t* __restrict__ q_tag, uint64_t* __restrict__ q_p0, uint64_t* __restrict__ seg_next) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const Id tail = seg_head[s ^ 1] ^ (Id)1;                              // the segment ends where its reverse begins
    const Id g = nxtG[tail];
    seg_next[s] = ABSENT;
    if (g == NONE) return;
    const unsigned long long at = atomicAdd(qn, 1ull);
    if (at < qcap) { q_tag[at] = ((uint64_t)rank_of_index(M, g >> 1) << 58) | s; q_p0[at] = g; }
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
struct alignas(32) SegRec { unsigned long long w; uint64_t hi, lo, head; };     // (length, next segment or itself), the head's oriented k-mer, its job-wide node
__global__ void __launch_bounds__(256) k_seg_records(uint64_t nseg, const Id* __restrict__ seg_head, const uint32_t* __restrict__ seg_len, const uint64_t* __restrict__ seg_next,
                                                      const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo, uint64_t base2_me, uint64_t segbase_me,
                                                      SegRec* __restrict__ out) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const Id v = seg_head[s];
    Kmer k{shi[v >> 1], slo[v >> 1]};
    if (v & 1) k = kmer_rc(k);
    const uint64_t nx = seg_next[s];
    SegRec r;
    // (length, next): a chain END points at itself -- the jumping never adds an end's distance field, which is free to carry ITS length too
    r.w = RankW<Id>::pack(seg_len[s], nx == ABSENT ? segbase_me + s : nx);
    r.hi = k.hi; r.lo = k.lo; r.head = base2_me + v;
    out[s] = r;
}
// level 2, replicated: the gathered records -> rank words, lengths
__global__ void __launch_bounds__(256) k_seg_unpack(uint64_t NS, const SegRec* __restrict__ g, unsigned long long* __restrict__ w2, unsigned long long* __restrict__ w2o) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= NS) return;
    w2[s] = g[s].w; w2o[s] = g[s].w;
}
// Level 2 is a list ranking over RANDOM-ACCESS lists (a segment's successor is anywhere in the gathered array), so plain pointer jumping
// would move every segment's word log(chain) times, a 64-B sector per 8-B word (the one-GPU splitter jumping does: 0.08 ns per word and
// launch).  Work-efficient instead (Helman-JaJa): SPLITTERS = the chain heads + one segment in 64 by a hash of its number; every splitter
// WALKS to the next splitter or the chain's end, summing lengths (k_seg_walk1); pointer jumping ranks the splitters alone (1/64 of the
// words); every splitter walks its stretch again and hands every segment its chain's end and its distance to it (k_seg_walk2).  Two passes of
// dependent loads over the segments instead of ~3 launches x ~8 jumps; a circle that has no splitter is never visited, one that has some
// never reaches an end: both stay marked ABSENT.
constexpr int SEG_JUMPS = 16;
__device__ inline bool seg_sampled(uint64_t s) { return ((s * 0x9E3779B97F4A7C15ull) >> 58) == 0; }
__global__ void __launch_bounds__(256) k_seg_mark(uint64_t NS, const unsigned long long* __restrict__ w2o, uint8_t* __restrict__ sp, uint64_t* __restrict__ spl,
                                                   unsigned long long* __restrict__ nspl, uint64_t cap, uint64_t* __restrict__ Fend) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= NS) return;
    Fend[s] = ABSENT;
    const bool head = RankW<Id>::next(w2o[s ^ 1]) == (s ^ 1);             // the reverse of s is a chain end: s is a chain head
    const bool is = head || seg_sampled(s);
    sp[s] = is;
    if (is) { const unsigned long long at = atomicAdd(nspl, 1ull); if (at < cap) spl[at] = s; }
}
// splitter s -> w2[s] = (k-mers from its head up to the next splitter's head, that splitter); a stretch that reaches the chain's end F:
// w2[s] = (0, s) -- an end of the splitter list --, T[s] = k-mers from its head to the end of the chain, Fend[s] = F
__global__ void __launch_bounds__(256) k_seg_walk1(uint64_t n, const uint64_t* __restrict__ spl, const unsigned long long* __restrict__ w2o, const uint8_t* __restrict__ sp,
                                                    unsigned long long* __restrict__ w2, uint64_t* __restrict__ Fend, uint64_t* __restrict__ T, uint64_t max_steps) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = spl[i];
    uint64_t cur = s, acc = 0;
    for (uint64_t step = 0; step < max_steps; ++step) {
        const unsigned long long w = w2o[cur];
        const uint64_t nx = RankW<Id>::next(w);
        acc += RankW<Id>::dist(w);
        if (nx == cur) { w2[s] = RankW<Id>::pack(0, s); T[s] = acc; Fend[s] = cur; return; }
        if (sp[nx]) { w2[s] = RankW<Id>::pack(acc, nx); return; }
        cur = nx;
    }
    w2[s] = RankW<Id>::pack(acc, cur);                                     // (a stretch longer than any chain can be: left unfinished, its segments stay ABSENT)
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
// the second walk: every segment of a splitter's stretch learns the chain's end and its own distance to it
__global__ void __launch_bounds__(256) k_seg_walk2(uint64_t n, const uint64_t* __restrict__ spl, const unsigned long long* __restrict__ w2o, const uint8_t* __restrict__ sp,
                                                    const unsigned long long* __restrict__ w2, uint64_t* __restrict__ Fend, uint64_t* __restrict__ T, uint64_t max_steps) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t s = spl[i];
    const unsigned long long ws = w2[s];
    const uint64_t e = RankW<Id>::next(ws);                               // the last splitter of the chain, if the splitter list ended
    if (RankW<Id>::next(w2[e]) != e) return;                              // a circle of splitters: its segments stay ABSENT
    const uint64_t F = Fend[e];
    if (F == ABSENT) return;
    uint64_t t = (e == s ? 0 : RankW<Id>::dist(ws)) + T[e];
    uint64_t cur = s;
    for (uint64_t step = 0; step < max_steps; ++step) {
        const unsigned long long w = w2o[cur];
        const uint64_t nx = RankW<Id>::next(w);
        if (cur != s && cur != e) { Fend[cur] = F; T[cur] = t; }          // (the splitters' own entries are written below: e's are being read by other walks)
        t -= RankW<Id>::dist(w);
        if (nx == cur || sp[nx]) break;
        cur = nx;
    }
}
__global__ void __launch_bounds__(256) k_seg_splitters_done(uint64_t n, const uint64_t* __restrict__ spl, const unsigned long long* __restrict__ w2, uint64_t* __restrict__ Fend,
                                                             uint64_t* __restrict__ T, uint64_t* __restrict__ Fsp, uint64_t* __restrict__ Tsp) {
    // (two steps so that no walk reads an end splitter's T / Fend while another thread rewrites them: first into side arrays ...)
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
// per segment: its own length; on a circle (never reached from a chain's end): flag
__global__ void __launch_bounds__(256) k_seg_finish(uint64_t NS, const unsigned long long* __restrict__ w2o, const uint64_t* __restrict__ Fend, uint64_t* __restrict__ T,
                                                     uint32_t* __restrict__ len, uint8_t* __restrict__ cyc2, uint32_t* __restrict__ flags) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= NS) return;
    len[s] = (uint32_t)RankW<Id>::dist(w2o[s]);
    const bool cyc = Fend[s] == ABSENT;
    cyc2[s] = cyc;
    if (cyc) { flags[2] = 1; T[s] = 0; }
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
__global__ void __launch_bounds__(256) k_segmin_init(uint64_t NS, const unsigned long long* __restrict__ w2o, const uint8_t* __restrict__ cyc2, uint64_t* __restrict__ nx, uint64_t* __restrict__ mn) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= NS) return;
    nx[s] = cyc2[s] ? RankW<Id>::next(w2o[s]) : s;
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
    uint64_t F0, F1;
    const uint64_t r0 = to_end<LId>(own, w, segbase_me, len, Fend, T, 2 * i, F0) - 1, r1 = to_end<LId>(own, w, segbase_me, len, Fend, T, 2 * i + 1, F1) - 1;
    const uint64_t n = r0 + r1 + 1;
    if (n & 1) return;
    const uint64_t q = n / 2 + 29, x = q < n - 1 ? q : n - 1;
    if (r1 != x && r0 != x) return;
    const unsigned off = (unsigned)(q - x);
    const Kmer k{shi[i], slo[i]};
    if (r1 == x) mid[F1 ^ 1] = (uint8_t)(4u | kmer_base(k, off));                     // the head segment of the chain through (i, 0) is the flip of (i, 1)'s end
    if (r0 == x) mid[F0 ^ 1] = (uint8_t)(4u | kmer_base(kmer_rc(k), off));
}
// canonical heads (k_heads of step2_graph.hip, on segments): bvec::getCanonicalForm
__global__ void __launch_bounds__(256) k_heads_shard(uint64_t NS, const SegRec* __restrict__ g, const unsigned long long* __restrict__ w2o, const uint64_t* __restrict__ Fend,
                                                      const uint64_t* __restrict__ T, const uint8_t* __restrict__ mid, uint64_t* __restrict__ head_seg,
                                                      uint64_t* __restrict__ key_hi, uint64_t* __restrict__ key_lo, unsigned long long* __restrict__ n_heads, uint64_t cap,
                                                      uint32_t* __restrict__ flags) {
    const uint64_t H = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (H >= NS) return;
    if (RankW<Id>::next(w2o[H ^ 1]) != (H ^ 1)) return;                   // the reverse of H is not a chain end: H is not a chain head
    const uint64_t F = Fend[H];
    if (F == ABSENT) return;
    const uint64_t n = T[H];
    if (n - 1 > 0xFFFFFFull) atomicOr(&flags[1], 2u);                     // GE_OFFSET, ReadPather.h:122
    const Kmer Fk{g[H].hi, g[H].lo};
    bool canon;
    if (kmer_is_pal(Fk)) canon = !(g[H].head & 1);
    else if (n & 1) canon = kmer_lt(Fk, Kmer{g[F ^ 1].hi, g[F ^ 1].lo});
    else canon = !(mid[H] & 2);
    if (!canon) return;
    const unsigned long long pos = atomicAdd(n_heads, 1ull);
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
    uint64_t F0, F1;
    const uint64_t r0 = to_end<LId>(own, w, segbase_me, len, Fend, T, 2 * i, F0) - 1, r1 = to_end<LId>(own, w, segbase_me, len, Fend, T, 2 * i + 1, F1) - 1;
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
enum Phase { PH_BEGIN = 0, PH_A_ANSWER, PH_A_APPLY, PH_B_ANSWER, PH_B_APPLY, PH_SEGBASE, PH_C_ANSWER, PH_C_APPLY, PH_LEVEL2, PH_CIRC_MIN, PH_CIRC_CUT,
             PH_HEADS, PH_STREAM, PH_INDEX, PH_FILTER, PH_DONE };

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
    Id* seg_head = nullptr; uint32_t* seg_len = nullptr; uint64_t* seg_next = nullptr; SegRec* seg_rec = nullptr;
    uint64_t* h_small = nullptr;                                          // pinned host words for the tiny all-gathers
    SegRec* G = nullptr; unsigned long long *w2 = nullptr, *w2o = nullptr; uint64_t *Fend = nullptr, *T = nullptr; uint8_t *cyc2 = nullptr, *mid = nullptr;
    uint32_t* lenS = nullptr;
    MinRec* minrec = nullptr; uint64_t* cuts = nullptr; uint64_t *mn = nullptr;
    uint32_t* edge_of_head = nullptr; uint32_t* bits = nullptr; uint32_t* bits_keep = nullptr; uint64_t nwords = 0;
    uint4* idx_list = nullptr; unsigned long long* flt_slice = nullptr;
};

static Shard& sh(Ctx& c) { return *static_cast<Shard*>(c.shard); }
// a kernel templated on the local id type, launched for the width this rank uses
// (the arguments may name the type as LId: `(const LId*)s.nxtL`)
#define LAUNCH_L(c, s, name, kern, grid, block, ...)                                                   \
    do {                                                                                              \
        if ((s).local32) { using LId = uint32_t; LAUNCH(c, name, (kern<LId>), grid, block, 0, __VA_ARGS__); }   \
        else { using LId = uint64_t; LAUNCH(c, name, (kern<LId>), grid, block, 0, __VA_ARGS__); }               \
    } while (0)

// items (tag, p0[, p1]) -> blocks by destination; fills x for the all-to-all; keeps the tags (in send order) for the answers
static int route(Ctx& c, Shard& s, uint64_t n, uint64_t* tag, uint64_t* p0, uint64_t* p1, w2rap_xchg* x) {
    hipStream_t st = c.stream;
    unsigned long long* d_h = nullptr;
    W2_ALLOC(d_h, unsigned long long, 64);
    W2_HIP(hipMemsetAsync(d_h, 0, 64 * 8, st));
    if (n) hipLaunchKernelGGL(k_route_hist, dim3(grid_for(n)), dim3(256), 0, st, n, tag, d_h);
    unsigned long long h[64];
    W2_HIP(hipMemcpyAsync(h, d_h, sizeof(h), hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    unsigned long long off[64]; uint64_t run = 0;
    for (unsigned r = 0; r < 64; ++r) { off[r] = run; run += h[r]; s.q_counts[r] = h[r]; }
    W2_HIP(hipMemcpyAsync(d_h, off, sizeof(off), hipMemcpyHostToDevice, st));
    const unsigned words = p1 ? 2 : 1;
    if (s.q_tag) c.release(s.q_tag);
    if (s.q_send) c.release(s.q_send);
    W2_ALLOC(s.q_tag, uint64_t, n); W2_ALLOC(s.q_send, uint64_t, n * words);
    if (n) hipLaunchKernelGGL(k_route_scatter, dim3(grid_for(n)), dim3(256), 0, st, n, tag, p0, p1, d_h, s.q_tag, s.q_send);
    W2_HIP(hipStreamSynchronize(st));
    W2_HIP(hipGetLastError());
    c.release(d_h);
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
    s.hint = hint; s.S = c.S; s.phase = PH_BEGIN;
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
    if (c.table_built && c.d_table && c.ld_done == S && 10 * c.tcap >= 13 * S) { if (c.stream2) W2_HIP(hipStreamSynchronize(c.stream2)); c.table_built = false; }
    else W2_TRY(table_build_plain(c));
    uint8_t* sctx0 = nullptr; void* nbrL = nullptr; uint8_t* unres = nullptr;
    {
        const char* wv = getenv("W2RAP_WIDE_IDS");
        s.local32 = S < (1ull << 31) - 1 && !(wv && atoi(wv) != 0);
    }
    W2_ALLOC(sctx0, uint8_t, S + 4); W2_ALLOC(unres, uint8_t, S + 4);
    if (s.local32) { uint32_t* q = nullptr; W2_ALLOC(q, uint32_t, 2 * S); nbrL = q; } else { uint64_t* q = nullptr; W2_ALLOC(q, uint64_t, 2 * S); nbrL = q; }
    bool have_local = false;
    if (S && c.nchunks) { W2_TRY(s.local32 ? prune_local_chunks32(c, sctx0, (uint32_t*)nbrL, unres) : prune_local_chunks64(c, sctx0, (uint64_t*)nbrL, unres)); have_local = true; }
    for (void* p : {(void*)c.d_sctx, (void*)c.d_nbr}) if (p) c.release(p);
    c.d_nbr = nullptr;
    W2_ALLOC(c.d_sctx, uint8_t, S + 4);
    W2_HIP(hipMemsetAsync(c.d_sctx, 0, S + 4, st));
    W2_ALLOC(s.nxtG, Id, 2 * S);
    unsigned long long* d_qn = nullptr;
    W2_ALLOC(d_qn, unsigned long long, 1);
    uint64_t qcap = S + S / 2 + 4096;
    if (test_hook("W2RAP_TEST_SHARD_QCAP")) qcap = (uint64_t)atoll(getenv("W2RAP_TEST_SHARD_QCAP"));
    uint64_t *q_tag = nullptr, *q_hi = nullptr, *q_lo = nullptr;
    unsigned long long nq = 0;
    for (int attempt = 0;; ++attempt) {
        W2_ALLOC(q_tag, uint64_t, qcap); W2_ALLOC(q_hi, uint64_t, qcap); W2_ALLOC(q_lo, uint64_t, qcap);
        W2_HIP(hipMemsetAsync(d_qn, 0, 8, st));
        if (S) {
            if (s.local32) LAUNCH(c, "k_prune_shard", k_prune_shard<uint32_t>, dim3(grid_for(S)), dim3(256), 0, S, c.d_shi, c.d_slo, c.d_scc, c.d_table, c.tcap - 1, (const uint8_t*)sctx0,
                                  (const uint32_t*)nbrL, have_local ? (const uint8_t*)unres : (const uint8_t*)nullptr, s.M, c.d_sctx, s.nxtG, d_qn, qcap, q_tag, q_hi, q_lo);
            else LAUNCH(c, "k_prune_shard", k_prune_shard<uint64_t>, dim3(grid_for(S)), dim3(256), 0, S, c.d_shi, c.d_slo, c.d_scc, c.d_table, c.tcap - 1, (const uint8_t*)sctx0,
                        (const uint64_t*)nbrL, have_local ? (const uint8_t*)unres : (const uint8_t*)nullptr, s.M, c.d_sctx, s.nxtG, d_qn, qcap, q_tag, q_hi, q_lo);
        }
        W2_HIP(hipMemcpyAsync(&nq, d_qn, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        if (nq <= qcap) break;
        if (attempt) { c.err = "sharded prune: query list overflow after resizing"; return W2RAP_E_LIMIT; }
        c.release(q_tag); c.release(q_hi); c.release(q_lo);
        qcap = nq + 1024;                                                 // more open neighbours than room: their number is known now
    }
    c.release(sctx0); c.release(nbrL); c.release(unres); c.release(d_qn);
    W2_TRY(route(c, s, nq, q_tag, q_hi, q_lo, x));
    c.release(q_tag); c.release(q_hi); c.release(q_lo);
    if (getenv("W2RAP_TRACE")) fprintf(stderr, "[w2rap] shard %u/%u: %llu solid k-mers, %llu neighbour queries to other owners\n", s.M.me, s.M.world, (unsigned long long)S, nq);
    return 0;
}

template <class LId>
__global__ void __launch_bounds__(256) k_mirror_cuts(uint64_t N, const LId* __restrict__ nxtL, ShardMap M, Id* __restrict__ nxtG) {
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= N) return;
    const Id g = nxtG[v];
    if (g != NONE && nxtL[v] == NodeId<LId>::NONE && (g >> 1) >= M.base[M.me] && (g >> 1) < M.base[M.me + 1]) nxtG[v] = NONE;
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
    unsigned long long* d_n = nullptr;
    W2_ALLOC(d_n, unsigned long long, 1);
    W2_HIP(hipMemsetAsync(d_n, 0, 8, st));
    const uint64_t cap = S ? c.rank_ends + 2 : 2;                         // chains <= chain ends
    for (void* p : {(void*)s.seg_head, (void*)s.seg_len, (void*)s.seg_next}) if (p) c.release(p);
    W2_ALLOC(s.seg_head, Id, 2 * cap); W2_ALLOC(s.seg_len, uint32_t, 2 * cap); W2_ALLOC(s.seg_next, uint64_t, 2 * cap);
    if (S) LAUNCH_L(c, s, "k_seg_number", k_seg_number, dim3(grid_for(N)), dim3(256), S, (const LId*)s.nxtL, (const uint32_t*)s.own, s.rankw, d_n, cap, s.seg_head, s.seg_len);
    unsigned long long nch = 0;
    W2_HIP(hipMemcpyAsync(&nch, d_n, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    W2_HIP(hipGetLastError());
    c.release(d_n);
    if (nch > cap) { c.err = "sharded graph: more local chains than chain ends"; return W2RAP_E_GRAPH; }
    s.nseg = 2 * nch;
    s.h_small[0] = s.nseg;
    std::memset(x, 0, sizeof(*x));
    x->op = W2RAP_X_ALLGATHER_HOST; x->elem_bytes = 8; x->send = s.h_small; x->send_count[0] = 1;
    return 0;
}

static int level2(Ctx& c, Shard& s, w2rap_xchg* x, bool* circles);
static int heads_and_stream(Ctx& c, Shard& s, w2rap_xchg* x);

int shard_next(Ctx& c, w2rap_xchg* x) {
    if (!c.shard) { c.err = "shard_next before shard_begin"; return W2RAP_E_STATE; }
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
        unsigned long long* d_qn = nullptr;
        W2_ALLOC(d_qn, unsigned long long, 1);
        uint64_t qcap = S / 4 + 4096;
        uint64_t *q_tag = nullptr, *q_p0 = nullptr;
        unsigned long long nq = 0;
        for (int attempt = 0;; ++attempt) {
            W2_ALLOC(q_tag, uint64_t, qcap); W2_ALLOC(q_p0, uint64_t, qcap);
            W2_HIP(hipMemsetAsync(d_qn, 0, 8, st));
            if (S) LAUNCH(c, "k_prune_final", k_prune_final, dim3(grid_for(S)), dim3(256), 0, S, (const uint8_t*)c.d_sctx, s.nxtG, s.M, d_qn, qcap, q_tag, q_p0);
            W2_HIP(hipMemcpyAsync(&nq, d_qn, 8, hipMemcpyDeviceToHost, st));
            W2_HIP(hipStreamSynchronize(st));
            W2_HIP(hipGetLastError());
            if (nq <= qcap) break;
            if (attempt) { c.err = "sharded prune: context query list overflow after resizing"; return W2RAP_E_LIMIT; }
            c.release(q_tag); c.release(q_p0);
            qcap = nq + 1024;                                             // (k_prune_final is idempotent: a second pass rewrites the same words)
        }
        c.release(d_qn);
        W2_TRY(route(c, s, nq, q_tag, q_p0, nullptr, x));
        c.release(q_tag); c.release(q_p0);
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
        unsigned long long* d_qn = nullptr;
        W2_ALLOC(d_qn, unsigned long long, 1);
        W2_HIP(hipMemsetAsync(d_qn, 0, 8, st));
        const uint64_t qcap = s.nseg + 1;
        uint64_t *q_tag = nullptr, *q_p0 = nullptr;
        W2_ALLOC(q_tag, uint64_t, qcap); W2_ALLOC(q_p0, uint64_t, qcap);
        if (s.nseg) LAUNCH(c, "k_seg_queries", k_seg_queries, dim3(grid_for(s.nseg)), dim3(256), 0, s.nseg, (const Id*)s.seg_head, (const uint32_t*)s.own,
                           (const unsigned long long*)s.rankw, (const Id*)s.nxtG, s.M, d_qn, qcap, q_tag, q_p0, s.seg_next);
        unsigned long long nq = 0;
        W2_HIP(hipMemcpyAsync(&nq, d_qn, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        c.release(d_qn);
        W2_TRY(route(c, s, nq, q_tag, q_p0, nullptr, x));
        c.release(q_tag); c.release(q_p0);
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
        if (s.seg_rec) c.release(s.seg_rec);
        W2_ALLOC(s.seg_rec, SegRec, s.nseg);
        if (s.nseg) LAUNCH(c, "k_seg_records", k_seg_records, dim3(grid_for(s.nseg)), dim3(256), 0, s.nseg, (const Id*)s.seg_head, (const uint32_t*)s.seg_len, (const uint64_t*)s.seg_next,
                           c.d_shi, c.d_slo, 2 * s.M.base[s.M.me], s.segbase[s.M.me], s.seg_rec);
        W2_HIP(hipStreamSynchronize(st));
        W2_HIP(hipGetLastError());
        x->op = W2RAP_X_ALLGATHER; x->elem_bytes = sizeof(SegRec); x->send = s.seg_rec; x->send_count[0] = s.nseg;
        s.phase = PH_LEVEL2;
        return 0;
    }
    case PH_LEVEL2: {
        bool circles = false;
        W2_TRY(level2(c, s, x, &circles));
        s.phase = circles ? PH_CIRC_MIN : PH_HEADS;
        return 0;
    }
    case PH_CIRC_MIN: {                                                   // every chain's minimum k-mer has arrived: the circles' minima, the cuts
        const uint64_t NC = s.NS / 2;
        const MinRec* mr = (const MinRec*)s.recv;
        uint64_t *nx, *mn, *nx2, *mn2;
        W2_ALLOC(nx, uint64_t, s.NS); W2_ALLOC(mn, uint64_t, s.NS); W2_ALLOC(nx2, uint64_t, s.NS); W2_ALLOC(mn2, uint64_t, s.NS);
        (void)NC;
        LAUNCH(c, "k_segmin_init", k_segmin_init, dim3(grid_for(s.NS)), dim3(256), 0, s.NS, (const unsigned long long*)s.w2o, (const uint8_t*)s.cyc2, nx, mn);
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
        for (void* p : {(void*)s.nxtG, (void*)s.nxtL, (void*)s.rankw, (void*)s.own, (void*)s.seg_head, (void*)s.seg_len, (void*)s.seg_next, (void*)s.seg_rec, (void*)s.G, (void*)s.w2,
                        (void*)s.w2o, (void*)s.Fend, (void*)s.T, (void*)s.cyc2, (void*)s.mid, (void*)s.lenS, (void*)s.edge_of_head, (void*)s.q_tag, (void*)s.q_send, (void*)s.resp,
                        (void*)s.d_flags, (void*)s.cuts})
            if (p) c.release(p);
        s.nxtG = nullptr; s.nxtL = nullptr; s.rankw = nullptr; s.own = nullptr; s.seg_head = nullptr; s.seg_len = nullptr; s.seg_next = nullptr; s.seg_rec = nullptr; s.G = nullptr;
        s.w2 = s.w2o = nullptr; s.Fend = s.T = nullptr; s.cyc2 = s.mid = nullptr; s.lenS = nullptr; s.edge_of_head = nullptr; s.q_tag = s.q_send = s.resp = nullptr;
        s.d_flags = nullptr; s.cuts = nullptr;
        c.counted = true;
        // the packed stream stays as it was summed (+ 16 bytes of slack behind it, which the word array has); the index entries of this rank's
        // share of its positions are listed and gathered -- every rank then inserts all of them: the window minima, which are the cost of the
        // index, are computed once per position in the JOB, not once per rank
        c.d_edge_bits = reinterpret_cast<uint8_t*>(s.bits_keep);
        c.bits_ready = true;
        if (s.idx_list) { c.release(s.idx_list); s.idx_list = nullptr; }
        uint64_t n_list = 0;
        W2_TRY(index_entries_slice(c, s.M.me, s.M.world, &s.idx_list, &n_list));
        x->op = W2RAP_X_ALLGATHER; x->elem_bytes = 16; x->send = s.idx_list; x->send_count[0] = n_list;
        s.phase = PH_INDEX;
        return 0;
    }
    case PH_INDEX: {
        W2_TRY(index_from_entries(c, (const uint4*)s.recv, s.recv_total));
        W2_HIP(hipStreamSynchronize(st));
        if (s.idx_list) { c.release(s.idx_list); s.idx_list = nullptr; }
        if (s.recv) { c.release(s.recv); s.recv = nullptr; }
        // the absence filter the same way: every rank scans the stream, sets the bits of ITS range of words, the ranges are gathered
        if (filter32_words(c)) {
            uint64_t nw = 0;
            W2_TRY(filter32_slice(c, s.M.me, s.M.world, &s.flt_slice, &nw));
            x->op = W2RAP_X_ALLGATHER; x->elem_bytes = 8; x->send = s.flt_slice; x->send_count[0] = nw;
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

static int level2(Ctx& c, Shard& s, w2rap_xchg* x, bool* circles) {        // the gathered segment records: ranks of the segment chains, replicated
    hipStream_t st = c.stream;
    const uint64_t NS = s.NS, S = s.S;
    if (s.recv_total != NS) { c.err = "sharded graph: gathered " + std::to_string(s.recv_total) + " segment records, expected " + std::to_string(NS); return W2RAP_E_STATE; }
    for (void* p : {(void*)s.G, (void*)s.w2, (void*)s.w2o, (void*)s.Fend, (void*)s.T, (void*)s.cyc2, (void*)s.mid, (void*)s.lenS}) if (p) c.release(p);
    s.G = (SegRec*)s.recv; s.recv = nullptr;                               // the gathered records stay (head k-mers, head nodes)
    W2_ALLOC(s.w2, unsigned long long, NS + 1); W2_ALLOC(s.w2o, unsigned long long, NS + 1); W2_ALLOC(s.Fend, uint64_t, NS + 1); W2_ALLOC(s.T, uint64_t, NS + 1);
    W2_ALLOC(s.cyc2, uint8_t, NS + 4); W2_ALLOC(s.mid, uint8_t, NS + 4); W2_ALLOC(s.lenS, uint32_t, NS + 1);
    W2_HIP(hipMemsetAsync(s.d_flags, 0, 32, st));
    if (NS) LAUNCH(c, "k_seg_unpack", k_seg_unpack, dim3(grid_for(NS)), dim3(256), 0, NS, (const SegRec*)s.G, s.w2, s.w2o);
    if (NS) {
        uint8_t* sp = nullptr; uint64_t *spl = nullptr, *Fsp = nullptr, *Tsp = nullptr; unsigned long long* d_n = nullptr;
        W2_ALLOC(sp, uint8_t, NS + 4); W2_ALLOC(d_n, unsigned long long, 1);
        uint64_t cap = NS / 32 + s.NS / 2 / 8 + 4096;                      // ~NS / 64 sampled + the heads; an overflow is followed by the exact size
        unsigned long long nspl = 0;
        for (int attempt = 0;; ++attempt) {
            W2_ALLOC(spl, uint64_t, cap);
            W2_HIP(hipMemsetAsync(d_n, 0, 8, st));
            LAUNCH(c, "k_seg_mark", k_seg_mark, dim3(grid_for(NS)), dim3(256), 0, NS, (const unsigned long long*)s.w2o, sp, spl, d_n, cap, s.Fend);
            W2_HIP(hipMemcpyAsync(&nspl, d_n, 8, hipMemcpyDeviceToHost, st));
            W2_HIP(hipStreamSynchronize(st));
            if (nspl <= cap) break;
            if (attempt) { c.err = "sharded graph: splitter list overflow after resizing"; return W2RAP_E_LIMIT; }
            c.release(spl);
            cap = nspl + 16;
        }
        W2_ALLOC(Fsp, uint64_t, nspl + 1); W2_ALLOC(Tsp, uint64_t, nspl + 1);
        const uint64_t max_steps = NS + 1;
        if (nspl) {
            LAUNCH(c, "k_seg_walk1", k_seg_walk1, dim3(grid_for(nspl)), dim3(256), 0, (uint64_t)nspl, (const uint64_t*)spl, (const unsigned long long*)s.w2o, (const uint8_t*)sp, s.w2, s.Fend, s.T, max_steps);
            for (int round = 0; round < 12; ++round) {                     // the splitter chains: 1/64 of the segments, log_17 launches
                W2_HIP(hipMemsetAsync(s.d_flags, 0, 4, st));
                LAUNCH(c, "k_seg_jump", k_seg_jump, dim3(grid_for(nspl)), dim3(256), 0, (uint64_t)nspl, (const uint64_t*)spl, s.w2, s.d_flags);
                uint32_t changed = 0;
                W2_HIP(hipMemcpyAsync(&changed, s.d_flags, 4, hipMemcpyDeviceToHost, st));
                W2_HIP(hipStreamSynchronize(st));
                if (!changed) break;
            }
            LAUNCH(c, "k_seg_walk2", k_seg_walk2, dim3(grid_for(nspl)), dim3(256), 0, (uint64_t)nspl, (const uint64_t*)spl, (const unsigned long long*)s.w2o, (const uint8_t*)sp,
                   (const unsigned long long*)s.w2, s.Fend, s.T, max_steps);
            LAUNCH(c, "k_seg_splitters_done", k_seg_splitters_done, dim3(grid_for(nspl)), dim3(256), 0, (uint64_t)nspl, (const uint64_t*)spl, (const unsigned long long*)s.w2, s.Fend, s.T, Fsp, Tsp);
            LAUNCH(c, "k_seg_splitters_store", k_seg_splitters_store, dim3(grid_for(nspl)), dim3(256), 0, (uint64_t)nspl, (const uint64_t*)spl, (const uint64_t*)Fsp, (const uint64_t*)Tsp, s.Fend, s.T);
        }
        W2_HIP(hipMemsetAsync(s.d_flags, 0, 32, st));
        LAUNCH(c, "k_seg_finish", k_seg_finish, dim3(grid_for(NS)), dim3(256), 0, NS, (const unsigned long long*)s.w2o, (const uint64_t*)s.Fend, s.T, s.lenS, s.cyc2, s.d_flags);
        W2_HIP(hipStreamSynchronize(st));
        c.release(sp); c.release(spl); c.release(Fsp); c.release(Tsp); c.release(d_n);
    }
    uint32_t h_flags[4] = {0, 0, 0, 0};
    W2_HIP(hipMemcpyAsync(h_flags, s.d_flags, 16, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    W2_HIP(hipGetLastError());
    *circles = h_flags[2] != 0;
    std::memset(x, 0, sizeof(*x));
    if (*circles) {
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
    if (NS) LAUNCH(c, "k_heads_shard", k_heads_shard, dim3(grid_for(NS)), dim3(256), 0, NS, (const SegRec*)s.G, (const unsigned long long*)s.w2o, (const uint64_t*)s.Fend,
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
