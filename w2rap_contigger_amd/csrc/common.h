// common.h -- shared device helpers and the context of libw2rap_step2.so (gfx950 only).
//
// K-mer layout on the device: a 60-mer is two 60-bit words, hi = bases 0..29,
// lo = bases 30..59, base 0 most significant.  Unsigned (hi,lo) order is the
// lexicographic order A<C<G<T, i.e. the order of the reference's KMer<60>::operator<
// (src/kmers/KMer.h:289-319), so "canonical" = min(k, rc(k)) matches
// CF<60>::getForm (src/dna/CanonicalForm.h:58-66).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace w2 {

constexpr unsigned K = 60;
constexpr unsigned MMER_F = 15;               // window length of the filter keys' minimizers
constexpr uint64_t M60 = (1ull << 60) - 1;
constexpr uint64_t EMPTY_HI = ~0ull;          // no canonical 60-mer has hi == all ones
constexpr uint32_t NONE32 = 0xFFFFFFFFu;

struct Kmer { uint64_t hi, lo; };

// reverse the 32 2-bit groups of a u64
__host__ __device__ inline uint64_t rev2_64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    x = __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    x = __builtin_bswap64(x);
#endif
    return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
}
// reverse complement of a 30-base word
__host__ __device__ inline uint64_t rc60(uint64_t w) { return rev2_64(~w & M60) >> 4; }
// an LSB-first 60-bit chunk (base p at bits 1:0) -> MSB-first word (base p at bits 59:58)
__host__ __device__ inline uint64_t lsb2msb60(uint64_t a) { return rev2_64(a & M60) >> 4; }

__host__ __device__ inline Kmer kmer_rc(Kmer k) { return Kmer{rc60(k.lo), rc60(k.hi)}; }             // KMer.h:205-227
__host__ __device__ inline Kmer kmer_succ(Kmer k, unsigned b) {                                       // KMer.h toSuccessor
    return Kmer{((k.hi << 2) | (k.lo >> 58)) & M60, ((k.lo << 2) | b) & M60};
}
__host__ __device__ inline Kmer kmer_pred(Kmer k, unsigned b) {                                       // KMer.h toPredecessor
    return Kmer{(k.hi >> 2) | ((uint64_t)b << 58), (k.lo >> 2) | ((k.hi & 3) << 58)};
}
__host__ __device__ inline bool kmer_lt(Kmer a, Kmer b) { return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo; }
__host__ __device__ inline bool kmer_eq(Kmer a, Kmer b) { return a.hi == b.hi && a.lo == b.lo; }
__host__ __device__ inline unsigned kmer_base(Kmer k, unsigned i) {
    return i < 30 ? (unsigned)(k.hi >> (2 * (29 - i))) & 3 : (unsigned)(k.lo >> (2 * (59 - i))) & 3;
}
__host__ __device__ inline unsigned kmer_first(Kmer k) { return (unsigned)(k.hi >> 58) & 3; }
__host__ __device__ inline unsigned kmer_last(Kmer k) { return (unsigned)k.lo & 3; }
// canonicalise: returns true when the RC was taken (CanonicalForm REV); palindromes stay forward
__host__ __device__ inline bool kmer_canon(Kmer& k) {
    Kmer r = kmer_rc(k);
    if (kmer_lt(r, k)) { k = r; return true; }
    return false;
}
__host__ __device__ inline bool kmer_is_pal(Kmer k) { return kmer_eq(kmer_rc(k), k); }

// KMerContext::rc == bit reversal of the byte (src/kmers/KMerContext.cc:18-36)
__host__ __device__ inline unsigned brev8(unsigned c) {
    c = ((c >> 4) | (c << 4)) & 0xFF;
    c = ((c >> 2) & 0x33) | ((c & 0x33) << 2);
    c = ((c >> 1) & 0x55) | ((c & 0x55) << 1);
    return c;
}
__host__ __device__ inline unsigned popc4(unsigned m) { return __builtin_popcount(m & 15u); }
__host__ __device__ inline unsigned single4(unsigned m) { return __builtin_ctz(m | 16u); }

// 64-bit mix of a k-mer (our choice; not observable in any output): ONE 64-bit multiply of the folded halves, then the
// high half folded into the low one (the dictionary indexes with the low bits and keeps the high half as fingerprint)
__host__ __device__ inline uint64_t kmer_hash(Kmer k) {
    uint64_t h = (k.hi ^ ((k.lo << 32) | (k.lo >> 32))) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 32);
}

// 60 stream bits starting at base position p of an LSB-first 2-bit stream held in u32 words
template <class W>
__device__ inline uint64_t stream60(const W* w, unsigned p) {
    unsigned o = 2 * p, i = o >> 5, s = o & 31;
    uint64_t lo = (uint64_t)w[i] | ((uint64_t)w[i + 1] << 32);
    uint64_t x = lo >> s;
    if (s) x |= (uint64_t)w[i + 2] << (64 - s);
    return x & M60;
}
template <class W>
__device__ inline unsigned stream_base(const W* w, unsigned p) { return (w[p >> 4] >> (2 * (p & 15))) & 3u; }
template <class W>
__device__ inline Kmer stream_kmer(const W* w, unsigned p) {
    return Kmer{lsb2msb60(stream60(w, p)), lsb2msb60(stream60(w, p + 30))};
}
// 16 bases (32 bits) of an LSB-first 2-bit stream starting at base position pos: ONE unaligned 8-byte
// global load (gfx950 global memory handles unaligned dwordx2), so a comparison of 16 bases is one
// round trip instead of 16.  The stream must be readable 8 bytes past its last used byte.
struct __attribute__((packed, aligned(1))) U64u { uint64_t v; };
__device__ inline uint32_t stream16_global(const uint8_t* __restrict__ s, uint64_t pos) {
    const uint64_t x = reinterpret_cast<const U64u*>(s + (pos >> 2))->v;
    return (uint32_t)(x >> (2 * (pos & 3)));
}
__device__ inline uint32_t rc32(uint32_t x) {      // reverse-complement of 16 packed bases
    x = ~x;
    x = __brev(x);
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}
// byte-addressed variants for global-memory reads / edges (.fastb packing, read starts on a byte)
__device__ inline unsigned packed_base(const uint8_t* b, uint64_t i) { return (b[i >> 2] >> (2 * (i & 3))) & 3u; }

// ---- dictionary over the solid k-mers ----------------------------------------------------------------
// Open addressing over 8-BYTE slots: fingerprint (top 24 bits of the k-mer's hash) << 40 | index of the k-mer in the solid
// arrays (40 bits: the reference's dictionary has no 2^31 ceiling, BuildReadQGraph.cc:1092, README.md:3 "17 Gbp"); ~0 = empty.
// An insert is ONE 64-bit CAS (keys are distinct, so the claim needs no comparison and there is no payload to store behind
// it), the table is 8 B x 4 S (a quarter of the bytes of slots that hold the key), four slots share a 32-B sector so that
// linear probing stays inside it, and a lookup verifies the key where the k-mer lives: in the SoA arrays (adjacency prune,
// edge hints) or in the 32-B record {hi, lo, KDef} that read pathing reads on ONE GPU -- a seed costs two dependent sectors (slot ->
// record), an absent k-mer one.  With the dictionary SHARDED over several GPUs (row e-3) read pathing asks the minimizer-sampled index
// over the replicated edge sequences below (EdgeIndex) instead.  (KmerDict / KmerDictEntry, kmers/ReadPather.h:104-169)
typedef unsigned long long Slot;
constexpr Slot SLOT_EMPTY = ~0ull;
constexpr unsigned SLOT_IDX_BITS = 40;
constexpr uint64_t SLOT_IDX_MASK = (1ull << SLOT_IDX_BITS) - 1;
constexpr uint64_t MAX_SOLID_KMERS = (1ull << 32) - (1ull << 20);   // a grid holds fewer than 2^32 threads (one per k-mer); node ids 2 i + 1 < 2^33 fit the rank words
__host__ __device__ inline uint64_t slot_fp(uint64_t h) { return h >> SLOT_IDX_BITS; }          // 24 bits of the hash (index bits come from the low end)
__host__ __device__ inline Slot slot_make(uint64_t h, uint64_t i) { return (slot_fp(h) << SLOT_IDX_BITS) | i; }   // never ~0: i < 2^32
__host__ __device__ inline uint64_t slot_index(Slot v) { return v & SLOT_IDX_MASK; }

// (two loops: the lanes of a wavefront first ALL walk their slots to a fingerprint match or an empty slot, then fetch the key
// together -- verifying inside the probing loop would cost a second dependent trip in every round some lane matches)
__device__ inline int64_t table_find_h(const Slot* __restrict__ t, uint64_t mask, const uint64_t* __restrict__ shi,
                                       const uint64_t* __restrict__ slo, Kmer k, uint64_t h) {
    uint64_t s = h & mask;
    const uint64_t fp = slot_fp(h);
    for (;;) {
        Slot v;
        for (;;) { v = t[s]; if (v == SLOT_EMPTY || (v >> SLOT_IDX_BITS) == fp) break; s = (s + 1) & mask; }
        if (v == SLOT_EMPTY) return -1;
        const uint64_t i = slot_index(v);
        if (shi[i] == k.hi && slo[i] == k.lo) return (int64_t)i;
        s = (s + 1) & mask;                                    // another k-mer with the same fingerprint (2^-24 per occupied slot passed)
    }
}
__device__ inline int64_t table_find(const Slot* __restrict__ t, uint64_t mask, const uint64_t* __restrict__ shi,
                                     const uint64_t* __restrict__ slo, Kmer k) {
    return table_find_h(t, mask, shi, slo, k, kmer_hash(k));
}
// read pathing through the dictionary: key and KDef come from the k-mer's 32-B record in one trip (the two halves are requested together)
struct alignas(32) KRec { uint64_t hi, lo; uint4 kdef; };    // kdef: see ctx.h d_srec
__device__ inline int64_t table_find_rec(const Slot* __restrict__ t, uint64_t mask, const KRec* __restrict__ rec, Kmer k, uint64_t h, uint4& kdef) {
    uint64_t s = h & mask;
    const uint64_t fp = slot_fp(h);
    for (;;) {
        Slot v;
        for (;;) { v = t[s]; if (v == SLOT_EMPTY || (v >> SLOT_IDX_BITS) == fp) break; s = (s + 1) & mask; }
        if (v == SLOT_EMPTY) return -1;
        const uint64_t i = slot_index(v);
        const ulonglong2 key = *reinterpret_cast<const ulonglong2*>(&rec[i]);
        kdef = rec[i].kdef;
        if (key.x == k.hi && key.y == k.lo) return (int64_t)i;
        s = (s + 1) & mask;
    }
}

// ---- oriented node ids of the unipath graph: 2 * (index of the solid k-mer) + (traversed reverse-complemented).  32 bits while
// S < 2^31; 64-bit words beyond (the reference's dictionary has no such ceiling: new BRQ_Dict(kmers.size()), BuildReadQGraph.cc:1092).
template <class Id> struct NodeId {
    static constexpr Id NONE = (Id)~(Id)0;          // no neighbour / not exactly one
    static constexpr Id PAL = (Id)(NONE - 1);       // the one neighbour is a palindrome (:198,210)
};
// list-ranking word of a node: (distance to `next`, next).  32-bit ids: 32 | 32.  Wide ids: 31 | 33 -- a unipath has at most 2^24 - 1
// k-mers (ForceAssertLe, kmers/ReadPather.h:122), distances SATURATE at 2^31 - 1 so that a longer chain is still reported as too long.
// The word of a CHAIN END is (0, itself); its distance field is free and later carries (unipath id + 1) of the canonical head v^1.
template <class Id> struct RankW;
template <> struct RankW<uint32_t> {
    __host__ __device__ static inline unsigned long long pack(uint64_t dist, uint32_t next) { return ((unsigned long long)dist << 32) | next; }
    __host__ __device__ static inline uint32_t next(unsigned long long w) { return (uint32_t)w; }
    __host__ __device__ static inline uint64_t dist(unsigned long long w) { return w >> 32; }
};
template <> struct RankW<uint64_t> {
    static constexpr unsigned NB = 33; static constexpr uint64_t DMAX = (1ull << 31) - 1;
    __host__ __device__ static inline unsigned long long pack(uint64_t dist, uint64_t next) { return ((dist < DMAX ? dist : DMAX) << NB) | next; }
    __host__ __device__ static inline uint64_t next(unsigned long long w) { return w & ((1ull << NB) - 1); }
    __host__ __device__ static inline uint64_t dist(unsigned long long w) { return w >> NB; }
};

// ---- list-ranking words of step2_graph.hip (see there)
constexpr unsigned RT = 512;                       // k-mers per tile (2 RT oriented nodes, 4 per thread: x = 256 q + tid)
constexpr unsigned RT_NODES = 2 * RT;
constexpr uint32_t OWN_CIRCLE = 0xFFFFFFFFu;      // own[v]: steps from the owner (bits 31:12) | v - owner + RT_NODES (bits 11:0; same tile)
template <class Id>
__device__ inline void rank_of(const uint32_t* __restrict__ own, const unsigned long long* __restrict__ w, Id v, Id& end, uint32_t& dist) {
    const uint32_t o = own[v];
    if (o == OWN_CIRCLE) { end = v; dist = 0; return; }                // a circle without splitters
    const unsigned long long x = w[v + RT_NODES - (o & 0xFFFu)];        // the owner lies in the same tile
    end = (Id)RankW<Id>::next(x);
    dist = end == v ? 0u : (uint32_t)RankW<Id>::dist(x) - (o >> 12);    // (a chain end's distance field is not a distance: edge_of_end)
}
// the unipath whose canonical head is the flip of chain end t (k_edge_from_sorted / k_edge_from_hint put id + 1 into the distance field
// of the end's own word, which is (0, t) after the ranking); NONE32: t^1 is not a canonical head
template <class Id>
__device__ inline uint32_t edge_of_end(const unsigned long long* __restrict__ w, Id t) {
    return (uint32_t)RankW<Id>::dist(w[t]) - 1u;
}

// ---- per edge object: its successors by next base and its place in the packed edge stream (step2_graph.hip k_obj_table, read pathing)
struct alignas(32) ObjRec { int32_t succ[4]; uint32_t eo_lo, eo_hi, elen, edge_rc /* unipath << 1 | reverse-complemented */; };

// ---- absence filter over the 31-mers of all unipath sequences ---------------------------------------
// A solid 60-mer lies inside its unipath, so every 31-mer of it occurs in some edge sequence.  Conversely a read 31-mer
// that occurs in NO edge proves that all (up to 30) 60-mers of the read containing it are absent from the dictionary:
// after a sequencing error at base e, TWO probes (31-mers at e-30 and e: 2 x 30 k-mers) replace the 60 per-k-mer probes
// that BRQ_Pather::path's base-by-base slide costs (31 is the largest length for which two suffice).  No false negatives.
// Layout: 64-bit words.  The WORD of a 31-mer is chosen by a canonical minimizer -- the smallest hashed 15-mer, over both
// strands, among the five windows at offsets 0, 4, .., 16 (a set that maps onto itself under reverse complement) -- its two
// BITS by the hash of the canonical 31-mer.  31-mers four positions apart share four of their five windows, so the 31-mers
// of an edge fall into runs with a common word; the builder ORs the bits of a wavefront's equal words together and issues
// ONE atomic per run (a quarter of the positions' count or less; device atomics run at 27 G/s whatever the table size).
// 31 bases as one u64, LSB first (base t at bits 2t+1:2t; the callers hand over 64 stream bits, the top group is dropped).
constexpr unsigned FMER = 31;                  // filter-mer length
constexpr unsigned FSPAN = K - FMER;           // the FMER-mer at base t lies in the k-mers t-FSPAN .. t
struct FmerKey { uint32_t word; uint64_t mask; };        // word: index before masking with the table size
__host__ __device__ inline FmerKey fmer_key(uint64_t x) {
    x &= (1ull << (2 * FMER)) - 1;
    const uint64_t rx = rev2_64(~x) >> (64 - 2 * FMER);   // reverse complement: its 15-mer at 16-j is the RC of x's 15-mer at j
    uint32_t mu = 0xFFFFFFFFu;
#pragma unroll
    for (unsigned j = 0; j <= FMER - MMER_F; j += 4) {
        const uint32_t f = (uint32_t)(x >> (2 * j)) & 0x3FFFFFFFu, r = (uint32_t)(rx >> (2 * (FMER - MMER_F - j))) & 0x3FFFFFFFu;
        const uint32_t key = (f < r ? f : r) * 0x9E3779B1u;                         // odd multiplier: a bijection, no ties
        mu = key < mu ? key : mu;
    }
    uint32_t w = mu ^ (mu >> 15); w *= 0x85EBCA6Bu; w ^= w >> 13;
    const uint64_t h = (rx < x ? rx : x) * 0x9E3779B97F4A7C15ull;
    return FmerKey{w, (1ull << ((h >> 58) & 63)) | (1ull << ((h >> 40) & 63))};
}

// ---- super-k-mer records (extract -> count hand-off) -------------------------------
// One fixed 32-B record (one HBM sector, never straddling two) = up to 63 consecutive k-mers of one read that share a bucket.
// byte 0: bits 5:0 nk-1, bit 6 hasL, bit 7 hasR.  From bit REC_HB = 8 the LSB-first 2-bit stream,
// t=0 left flank base (valid iff hasL), t=1..nk+59 the bases, t=nk+60 right flank (iff hasR).
// (-DW2RAP_REC36 builds the 36-B variant of rounds 1-3 -- a header DWORD, <= 64 k-mers, an odd record stride in the LDS tiles.  Round 3, with
// k_count_buckets: 36 B won, 49.5 against 52.0 ms -- stride 8 puts the stream words of neighbouring records on the same banks.  Round 4, with
// k_count_fp and the lane-per-read K1, both builds back to back on one box, twice: k_count_fp 39.2 -> 39.6 ms, but k_scatter_records alone
// 13.5 -> 10.0 ms and k_table_insert beside the counting kernel 31 -> 16.6 ms; the step 117.8 -> 113.6 and 126.3 -> 122.9 ms.)
#ifndef W2RAP_REC36
constexpr unsigned REC_DWORDS = 8;
constexpr unsigned REC_MAXK = 63;             // k-mers per record
constexpr unsigned REC_HB = 8;                // header bits in front of the stream
#else
constexpr unsigned REC_DWORDS = 9;
constexpr unsigned REC_MAXK = 64;
constexpr unsigned REC_HB = 32;
#endif
constexpr unsigned REC_BYTES = 4 * REC_DWORDS;
constexpr unsigned MMER = 15;                 // minimizer length
constexpr unsigned WIN = K - MMER + 1;        // m-mers per k-mer window (46)

// ---- canonical minimizers ---------------------------------------------------------------------------
// A 15-mer is held LSB first (base j at bits 2j+1:2j), its reverse complement the same way; the canonical one is the smaller INTEGER.
// key of a canonical m-mer: an odd multiplier is a bijection of 2^32, so distinct m-mers never tie and the window
// minimum depends only on the SET of canonical m-mers (strand-symmetric)
__host__ __device__ inline uint32_t mmer_key(uint32_t canon) { return canon * 0x9E3779B1u; }
// bucket of a minimizer key; re-mixed because the minimum of 46 keys is biased towards small values
__host__ __device__ inline uint32_t bucket_mix(uint32_t key) {
    uint32_t h = (key ^ (key >> 15)) * 0x85EBCA6Bu;
    return h ^ (h >> 13);
}
__device__ inline uint32_t bucket_of(uint32_t key, uint32_t nb) { return __umulhi(bucket_mix(key), nb); }
// reverse complement of the 15 bases in the low 30 bits of x (LSB first), LSB first again
__host__ __device__ inline uint32_t rc15(uint32_t x) {
    x = ~x;
#if defined(__HIP_DEVICE_COMPILE__)
    x = __brev(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = __builtin_bswap32(x);
#endif
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);      // the 16 groups reversed and complemented; group 0 is the unused one
    return x >> 2;
}
// The canonical minimizer of a 60-mer given as 120 stream bits (LSB first: lo = bases 0..31, hi = bases 32..59; higher bits ignored):
// key = the smallest mmer_key over its 46 15-mers (both strands), pos = the LEFTMOST 15-mer that attains it, fwd = that 15-mer itself
// (not its reverse complement) is the canonical one.  The reverse complement of the 60-mer has the same key, at position 45 - (its
// rightmost attaining 15-mer here).  ~500 lane operations, no loop-carried dependency but the running minimum.
struct MinHit { uint32_t key, pos; bool fwd; };
__host__ __device__ inline MinHit minimizer60(uint64_t lo, uint64_t hi) {
    // rolled like K1's cutter (small register state: this runs inside read pathing): f = the 15-mer LSB first, r = its reverse complement
    uint32_t f = ((uint32_t)lo << 2) & 0x3FFFFFFCu;                     // bases 0..13 wait in groups 1..14
    uint32_t r = rc15((uint32_t)lo & 0x0FFFFFFFu) >> 2;                 // their complements, base 13 first
    uint64_t W = (lo >> 28) | (hi << 36);                               // bases 14..45; 46..59 follow from hi
    uint32_t best = 0xFFFFFFFFu, pos = 0;
#pragma unroll 2
    for (unsigned j = 0; j < WIN; ++j) {
        if (j == 32) W = hi >> 28;
        const uint32_t b = (uint32_t)W & 3u; W >>= 2;
        f = (f >> 2) | (b << 28);
        r = ((r << 2) & 0x3FFFFFFFu) | (b ^ 3u);
        const uint32_t key = mmer_key(f < r ? f : r);
        if (key < best) { best = key; pos = j; }
    }
    // the strand of the chosen 15-mer, from the stream again
    const unsigned o = 2 * pos;
    const uint32_t fm = (uint32_t)(o < 64 ? (lo >> o) | (o ? hi << (64 - o) : 0ull) : hi >> (o - 64)) & 0x3FFFFFFFu;
    return MinHit{best, pos, fm < rc15(fm)};
}
// the same for a k-mer in the dictionary's layout (two 60-bit words, base 0 most significant)
__host__ __device__ inline MinHit minimizer_of(Kmer k) {
    const uint64_t a = rev2_64(k.hi << 4), b = rev2_64(k.lo << 4);       // bases 0..29 / 30..59, LSB first, 60 bits each
    return minimizer60(a | (b << 60), b >> 4);
}

// ---- minimizer-sampled index over the unipath sequences (read pathing's dictionary) -------------------
// Every solid k-mer lies on exactly one unipath at one offset (buildEdges :287-301), so the answer of the reference's dictionary lookup
// (KmerDict::findEntry -> KDef, kmers/ReadPather.h:104-169, BuildReadQGraph.cc:510-513) for a read k-mer is: the place in the packed
// edge stream where those 60 bases (or their reverse complement) occur inside ONE edge, or nothing.  The index holds one 16-B entry per
// (edge position whose 15-mer is the canonical minimizer of at least one of the edge's k-mers that contain it -- ties all kept --, side):
//   x = idx_hash(canonical 15-mer, the 16 bases on that side of it in its canonical orientation, side), bit 0 = the strand (1: the reverse
//       complement of the edge's 15-mer is the canonical one).  A 15-mer alone is not unique in a genome (250 Mbp: ~1.5 selected positions
//       per 15-mer, 17 Gbp: dozens -- measured with the 15-mer as key: 3.5 candidates verified per lookup at 250 Mbp); a k-mer that contains
//       it has at least 16 more bases on one side, so every position is entered once per side that the edge has, and a lookup asks for the
//       side its k-mer has: 31 bases identify the place.
//   y = unipath id (NONE32: empty slot)     z | w << 32 = position of the 15-mer in the edge stream
// in an open-addressing table probed from bucket_mix(x): ~4/47 entries per edge base instead of a 8-B slot + a 32-B record per solid
// k-mer.  A lookup takes the read k-mer's minimizer (leftmost on ties), walks the slots with that key and verifies each candidate's 60
// bases against the edge stream: two dependent trips (slot -> stream + the unipath's offset and length, fetched together), into a table
// of ~0.7 B per genome base and the 0.25 B per base stream, which stay cache resident where the dictionary's 64 B per k-mer never did.
// The index has minimizers of its own, cheaper than the buckets' (a lookup runs inside read pathing, which is bound by VALU issue): the
// canonical 15-mer -- held at bits 31:2, so that it comes out of the k-mer's stream words with ONE funnel shift -- is ordered by its top
// 26 bits XOR a constant (no multiply), and the window position rides in the six low bits of the compared word, so the running minimum is
// one v_min per position and picks the LEFTMOST of equal keys: five instructions per position.  Equal 26-bit keys of different 15-mers
// are ties like any other (the builder keeps every position that attains a window minimum, the lookup verifies 60 bases).
constexpr uint32_t IDX_XOR = 0x5A3C96E7u & ~63u;
__host__ __device__ inline uint32_t idx_key(uint32_t canon30) { return ((canon30 << 2) & ~63u) ^ IDX_XOR; }     // low six bits 0
__host__ __device__ inline uint32_t funnel_r(uint32_t lo, uint32_t hi, unsigned s) {      // bits s .. s+31 of hi:lo, 0 <= s < 32
#if defined(__HIP_DEVICE_COMPILE__)
    return __funnelshift_r(lo, hi, s);
#else
    return s ? (lo >> s) | (hi << (32 - s)) : lo;
#endif
}
struct IdxMin { uint32_t key, pos; bool fwd; uint64_t rlo, rhi; };      // key: idx_key of the chosen canonical 15-mer; (rlo, rhi): the reverse complement of the 60 bases
// the 46 positions, from the stream words of the 60 bases (w) and of their reverse complement (v): -> packed (key | position), strand.
// NOT inlined on the device: one copy of the unrolled code per kernel (three lookup sites in k_path), a register-only interface, and --
// what matters most -- the compiler cannot hoist this branch-free block above the conditions that guard a lookup and run it speculatively
// in every iteration of the pathing loop (measured: k_path 22 -> 30 ms with the inlined form).
__device__ __attribute__((noinline)) inline uint2 idx_min60_words(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    const uint32_t w[5] = {w0, w1, w2, w3, 0u}, v[5] = {v0, v1, v2, v3, 0u};
    uint32_t best = 0xFFFFFFFFu;
#pragma unroll
    for (unsigned j = 0; j < WIN; ++j) {
        // the 15-mer at j with its first base at bits 3:2 (bits 1:0: the base before it, or nothing -- never decides a comparison, the two
        // strands of an odd-length word differ above them); the reverse strand's 15-mer is the one at 45 - j of the reversed stream
        const unsigned of = 2 * j, orv = 2 * (WIN - 1 - j);
        const uint32_t f = of ? funnel_r(w[(of - 2) >> 5], w[((of - 2) >> 5) + 1], (of - 2) & 31) : w[0] << 2;
        const uint32_t r = orv ? funnel_r(v[(orv - 2) >> 5], v[((orv - 2) >> 5) + 1], (orv - 2) & 31) : v[0] << 2;
        const uint32_t c = f < r ? f : r;
        const uint32_t key = ((c & ~63u) ^ IDX_XOR) | j;
        best = key < best ? key : best;
    }
    // the strand of the chosen 15-mer, from the stream again
    const unsigned o = 2 * (best & 63u), i = o >> 5, sh = o & 31;
    uint32_t a = w0, b = w1;
    if (i == 1) { a = w1; b = w2; } else if (i == 2) { a = w2; b = w3; }
    const uint32_t fm = funnel_r(a, b, sh) & 0x3FFFFFFFu;
    uint2 out; out.x = best; out.y = fm < rc15(fm) ? 1u : 0u;
    return out;
}
// (lo, hi): 60 bases LSB first, hi's bits above 55 zero
__device__ inline IdxMin idx_min60(uint64_t lo, uint64_t hi) {
    // the reverse complement of the 120 bits: all 64 groups of (hi:lo) reversed leave the 60 on top, >> 8
    const uint64_t ra = rev2_64(hi), rb = rev2_64(lo);
    IdxMin m;
    m.rlo = ~((ra >> 8) | (rb << 56)); m.rhi = ~(rb >> 8) & ((1ull << 56) - 1);
    const uint2 r = idx_min60_words((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32), (uint32_t)m.rlo, (uint32_t)(m.rlo >> 32),
                                    (uint32_t)m.rhi, (uint32_t)(m.rhi >> 32));
    m.key = r.x & ~63u; m.pos = r.x & 63u; m.fwd = r.y != 0;
    return m;
}
struct __attribute__((packed, aligned(1))) U128u { uint64_t a, b; };
struct IdxHit { uint32_t e, off, nk; uint64_t eo; bool rc; };     // unipath, the k-mer's offset on the FORWARD unipath, its k-mers, its first base in the stream; read runs against it
struct EdgeIndex {
    const uint4* slots; uint64_t mask; const uint8_t* ebits; const uint64_t* edge_off; const uint32_t* edge_nk; uint64_t nbases;
    const uint4* xslots; uint64_t xmask;       // the exact table for the k-mers around keys with many entries (null: none)
};
// ---- the exact table beside the index.  A key of 31 bases is shared by every copy of a repeat whose boundary a k-mer straddles with its minimizer and
// that context inside the repeat: a lookup would verify the copies' entries one after the other (measured on the planted workload: 14 candidates per
// lookup, profiles/r05_planted_kernels.txt).  The entries of a key with more than IDX_HARD of them are marked (bit 31 of w), and every k-mer that
// covers such an entry's 15-mer is in a table of its own, keyed by a hash of its 60 bases: word 0 = tag (30 bits) | stream position (34), word 1 = unipath.
constexpr unsigned IDX_HARD = 3;
constexpr uint64_t XPOS_MASK = (1ull << 34) - 1;
constexpr unsigned long long XEMPTY = ~0ull;
// the hash of a k-mer given as 60 bases LSB first and their reverse complement: of the smaller of the two as integers (any strand-symmetric rule does)
__host__ __device__ inline uint64_t exact_hash(uint64_t lo, uint64_t hi, uint64_t rlo, uint64_t rhi) {
    const bool r = rhi != hi ? rhi < hi : rlo < lo;
    uint64_t a = r ? rlo : lo, b = r ? rhi : hi;
    a *= 0x9E3779B97F4A7C15ull; b = (b ^ (a >> 29)) * 0xC2B2AE3D27D4EB4Full;
    a ^= b >> 31; a *= 0x165667B19E3779F9ull;
    return a ^ (a >> 32) ^ b;
}
// (lo, hi): the read's 60 bases from the k-mer's first one, LSB first (lo = bases 0..31, hi = bases 32..59; higher bits ignored)
#ifdef W2RAP_IDX_STATS
static __device__ unsigned long long g_idx_stats[8];      // [0] lookups (lanes) [1] lookups (wavefront executions) [2] slots visited [3] candidates verified [4] hits
#define IDX_STAT(i, v) atomicAdd(&g_idx_stats[i], (unsigned long long)(v))
#else
#define IDX_STAT(i, v) ((void)0)
#endif
// the key of an entry / a lookup: the canonical 15-mer, the 16 bases on ONE side of it in its canonical orientation, which side
__host__ __device__ inline uint32_t idx_hash(uint32_t c30, uint32_t ctx, bool right) {
    uint32_t h = (c30 ^ (right ? 0x40000000u : 0u)) * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + ctx * 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    return h ^ (h >> 16);
}
__device__ inline bool exact_find(const EdgeIndex& X, uint64_t lo, uint64_t hi, uint64_t rlo, uint64_t rhi, IdxHit& out) {
    const uint64_t h = exact_hash(lo, hi, rlo, rhi), tag = h >> 34;
    for (uint64_t s = h & X.xmask;; s = (s + 1) & X.xmask) {
        const uint4 v = X.xslots[s];
        const uint64_t w0 = (uint64_t)v.x | ((uint64_t)v.y << 32);
        if (w0 == XEMPTY) return false;
        if ((w0 >> 34) != tag) continue;
        const uint64_t P = w0 & XPOS_MASK;
        const U128u w = *reinterpret_cast<const U128u*>(X.ebits + (P >> 2));
        const unsigned sh = 2 * (unsigned)(P & 3);
        const uint64_t a = sh ? (w.a >> sh) | (w.b << (64 - sh)) : w.a, b = (w.b >> sh) & ((1ull << 56) - 1);
        const bool same = a == lo && b == hi, opp = a == rlo && b == rhi;
        if (same || opp) {
            const uint64_t eo = X.edge_off[v.z];
            out.e = v.z; out.off = (uint32_t)(P - eo); out.nk = X.edge_nk[v.z]; out.eo = eo; out.rc = !same;
            return true;
        }
    }
}
__device__ inline bool index_find(const EdgeIndex& X, uint64_t lo, uint64_t hi, IdxHit& out) {
#ifdef W2RAP_IDX_STATS
    { const unsigned long long am = __ballot(1); if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(am)) { IDX_STAT(0, __builtin_popcountll(am)); IDX_STAT(1, 1); } }
#endif
    hi &= (1ull << 56) - 1;
    const IdxMin m = idx_min60(lo, hi);
    const uint64_t rlo = m.rlo, rhi = m.rhi;
    // the k-mer in the canonical orientation of its minimizer: the 15-mer sits at mc, with mc bases to its left and 45 - mc to its right;
    // the context is taken on the side that has sixteen of them
    const unsigned mc = m.fwd ? m.pos : (WIN - 1) - m.pos;
    const uint64_t slo = m.fwd ? lo : rlo, shi = m.fwd ? hi : rhi;
    auto bits32_at = [&](unsigned base) -> uint32_t { const unsigned o = 2 * base; return (uint32_t)(o < 64 ? (slo >> o) | (o ? shi << (64 - o) : 0ull) : shi >> (o - 64)); };
    const bool right = mc <= 29;
    const uint32_t key = idx_hash(bits32_at(mc) & 0x3FFFFFFFu, bits32_at(right ? mc + MMER : mc - 16), right);
    uint64_t s = bucket_mix(key) & X.mask;
    for (;;) {
        const uint4 v = X.slots[s];
        IDX_STAT(2, 1);
        if (v.y == NONE32) return false;
        if (((v.x ^ key) & ~1u) == 0 && ((v.w >> 30) & 1u) == (key & 1u)) {       // (the key's bit 0 lives in bit 30 of w: x gave its own to the strand)
            IDX_STAT(3, 1);
            if (v.w >> 31) return exact_find(X, lo, hi, rlo, rhi, out);    // a key with many entries: all of its entries are marked, its k-mers are in the exact table
            const bool same = ((v.x & 1u) == 0) == m.fwd;                  // the read k-mer lies on the edge as it is / reverse-complemented
            const uint64_t g = (uint64_t)v.z | ((uint64_t)(v.w & 0x3FFFFFFFu) << 32);
            const uint64_t shift = same ? m.pos : (WIN - 1) - m.pos;       // the 15-mer's offset inside the k-mer in EDGE orientation
            if (g >= shift && g - shift + K <= X.nbases) {
                const uint64_t P = g - shift;
                const U128u w = *reinterpret_cast<const U128u*>(X.ebits + (P >> 2));
                const uint64_t eo = X.edge_off[v.y]; const uint32_t nk = X.edge_nk[v.y];
                const unsigned sh = 2 * (unsigned)(P & 3);
                const uint64_t a = sh ? (w.a >> sh) | (w.b << (64 - sh)) : w.a, b = (w.b >> sh) & ((1ull << 56) - 1);
                if (a == (same ? lo : rlo) && b == (same ? hi : rhi) && P >= eo && P - eo < nk) {
                    out.e = v.y; out.off = (uint32_t)(P - eo); out.nk = nk; out.eo = eo; out.rc = !same;
                    IDX_STAT(4, 1);
                    return true;
                }
            }
        }
        s = (s + 1) & X.mask;
    }
}

}  // namespace w2
