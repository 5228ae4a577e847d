// step2_path.hip -- phases a9..a12 of Step 2 on gfx950: one read per lane.
//   BRQ_Pather::path            BuildReadQGraph.cc:500-550   seed lookups + matchLen along the edge
//   path_reads_OMP heuristics   :845-918                     (hanging-seed deletion is dead code, Q9/Q12)
//   pathPartsToReadPath         :804-827
//   ExtendReadPath left/right   paths/long/ExtendReadPath.cc:15-348 (with toRight := toLeft, :836-838)
//   FixPaths                    paths/long/large/GapToyTools.cc:322-335
// The kernel is bound by the NUMBER of divergent memory instructions a wavefront issues (every lane walks its own read; the
// texture addresser serialises their 64 addresses), not by bytes: so the block's 256 consecutive reads are staged in LDS with
// coalesced dword loads and every k-mer, 31-mer and comparison word of a read comes from there; matchLen compares 60 bases per
// ONE unaligned 16-byte load of the packed edge stream; the first parts and path elements of a read live in LDS (the rare rest
// in a small lane-interleaved spill area).  Blocks are persistent (chunks of 256 reads from an atomic queue), so the spill area
// is sized by the resident lanes, not by the reads.  A read's final path goes straight to its place: up to two elements inline
// in a per-read record, longer paths into a pool reserved with one atomic per wavefront; one scan + one gather build the CSR.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "ctx.h"

namespace w2 {

#ifdef W2RAP_IDX_STATS
#define SITE_STAT(k) do { const unsigned long long am_ = __ballot(1); if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(am_)) { atomicAdd(&g_site_stats[2 * (k)], (unsigned long long)__builtin_popcountll(am_)); atomicAdd(&g_site_stats[2 * (k) + 1], 1ull); } } while (0)
static __device__ unsigned long long g_site_stats[8];
#else
#define SITE_STAT(k) ((void)0)
#endif
constexpr unsigned PCS = 256;          // counter slots
constexpr unsigned LP = 2;             // parts of a read kept in LDS (most reads end with <= 4: seed, gap, seed, ...)
constexpr unsigned PL = 4;             // path elements of a read kept in LDS: logical positions pmid-1 .. pmid+PL-2
constexpr unsigned PATH_THREADS = 256;
#ifndef W2RAP_PATH_TICKETS
#define W2RAP_PATH_TICKETS 1
#endif
constexpr uint32_t PATH_TICKETS = W2RAP_PATH_TICKETS;      // chunks of 256 reads a block takes from the queue per atomic (a power of two; measured in
                                                           // round 5: four at a time 14.60 ms against 14.45 -- the uneven tail costs more than the tickets)
struct PathArgs {
    uint64_t n;                      // reads r_first .. n-1 (the lane-per-read first pass), or the entries 0 .. n-1 of `list`
    uint64_t r_first;
    // reads
    const uint8_t* bases; const uint64_t* boff; const uint32_t* len; const uint8_t* quals; const uint64_t* qoff;
    // dictionary + edges
    const Slot* table; uint64_t mask; const KRec* srec;     // the dictionary (one GPU) ...
    EdgeIndex X;                     // ... or the minimizer-sampled index over the edge sequences (common.h): read k-mer -> (unipath, offset, orientation)
    const unsigned long long* filter32; uint32_t f32mask;
    const uint8_t* codes; const uint8_t* ebits; const uint64_t* edge_off; const uint32_t* edge_nk;
    const int32_t* fwdX; const int32_t* revX; const uint32_t* obj_edge; const ObjRec* otab;
    const int32_t* left; const int32_t* right;
    const uint64_t* from_off; const int32_t* from_v; const int32_t* from_e;
    const uint64_t* to_off; const int32_t* to_v; const int32_t* to_e;
    // spill of the RESIDENT lanes (lane interleaved: element j of lane t at [j*T + t]): parts LP.., path elements outside the LDS window
    uint4* parts; int32_t* pbuf; uint32_t T; uint32_t maxparts; uint32_t pcap; uint32_t pmid;
    uint32_t rd_dwords;              // LDS dwords of the block's read staging area; 0: reads too long to stage, read from global memory
    // Reads that cut into many parts (high-copy repeats: dozens of tiny unipaths per read) are a few percent of the reads but would keep
    // nearly every wavefront waiting for its one slow lane.  The first pass gives up on a read at `part_budget` parts and lists it; a
    // second pass over the listed reads alone runs with every lane equally busy.
    uint32_t part_budget;            // first pass: parts per read before it is deferred (0: no limit)
    uint32_t* defer; uint64_t defer_cap;      // first pass: the deferred reads
    const uint32_t* list;            // second pass: the reads to path (A.n of them); nullptr: reads 0 .. n-1
    // per-read outputs
    uint32_t* plen; int2* inl; int32_t* pool; uint64_t pool_cap; int32_t* poffset;
    // k_path_dyn (lanes refilled as they finish): striped read cursors, the lane's LDS slot, the size of the packed-base array
    unsigned long long* stripes; uint32_t slot_dwords; uint64_t bases_bytes; uint32_t fin_thresh;
    unsigned long long* counters;    // [0] chunk queue, [1] pool cursor, [2] deferred reads, [3] -, then PCS slots of {pathed, multipathed} (a slot
                                     // per block residue: one address would serialise the wave-level atomics at ~11 ns each)
};

// part encoding: x = edge (unipath id) or 0xFFFFFFFF for a gap, y = offset, z = length, w = edge k-mers | rc<<31
__device__ inline bool part_gap(const uint4& p) { return p.x == NONE32; }
__device__ inline bool part_rc(const uint4& p) { return p.w >> 31; }
__device__ inline uint32_t part_elen(const uint4& p) { return p.w & 0x7FFFFFFFu; }
__device__ inline uint4 make_gap(uint32_t len) { return make_uint4(NONE32, 0, len, 0); }

// ---- the bases of the lane's read.  Callers never use bits of bases beyond the read (they mask by its length); the accessors only
// have to stay inside memory that may be read.
// (a) staged: the block's reads lie back to back in LDS (dword array w, this read's first byte at byte offset `off`, 16 B of slack behind)
struct RdLds {
    const uint32_t* w; uint32_t off;
    __device__ inline uint64_t bits64(uint32_t pos) const {                     // 32 bases from base pos
        const uint32_t byte = off + (pos >> 2), dw = byte >> 2, sh = 8 * (byte & 3) + 2 * (pos & 3);
        const uint32_t d0 = w[dw], d1 = w[dw + 1], d2 = w[dw + 2];
        return (uint64_t)__funnelshift_r(d0, d1, sh) | ((uint64_t)__funnelshift_r(d1, d2, sh) << 32);
    }
    __device__ inline void bits120(uint32_t pos, uint64_t& lo, uint64_t& hi) const {    // 60 bases: lo = bases 0..31, hi = bases 32..
        const uint32_t byte = off + (pos >> 2), dw = byte >> 2, sh = 8 * (byte & 3) + 2 * (pos & 3);
        const uint32_t d0 = w[dw], d1 = w[dw + 1], d2 = w[dw + 2], d3 = w[dw + 3], d4 = w[dw + 4];
        lo = (uint64_t)__funnelshift_r(d0, d1, sh) | ((uint64_t)__funnelshift_r(d1, d2, sh) << 32);
        hi = (uint64_t)__funnelshift_r(d2, d3, sh) | ((uint64_t)__funnelshift_r(d3, d4, sh) << 32);
    }
};
// (b) from global memory (reads too long to stage): 16 bases per unaligned 8-byte load, pulled back inside the read's bytes near its end
struct RdGlb {
    const uint8_t* rb; uint32_t nby;                                            // nby >= 15 (L >= K)
    __device__ inline uint32_t bits32(uint32_t pos) const {
        const uint32_t b0 = pos >> 2;
        if (b0 >= nby) return 0;
        const uint32_t b0c = b0 + 8 <= nby ? b0 : nby - 8, sh = 8 * (b0 - b0c) + 2 * (pos & 3);
        return (uint32_t)(reinterpret_cast<const U64u*>(rb + b0c)->v >> sh);
    }
    __device__ inline uint64_t bits64(uint32_t pos) const { return (uint64_t)bits32(pos) | ((uint64_t)bits32(pos + 16) << 32); }
    __device__ inline void bits120(uint32_t pos, uint64_t& lo, uint64_t& hi) const { lo = bits64(pos); hi = bits64(pos + 32); }
};
template <class RD>
__device__ inline Kmer read_kmer(const RD& rd, uint32_t p) {                    // the 60-mer at base p (p + 60 <= L)
    uint64_t lo, hi;
    rd.bits120(p, lo, hi);
    return Kmer{lsb2msb60(lo & M60), lsb2msb60(((lo >> 60) | (hi << 4)) & M60)};
}

// the dictionary's answer for the read k-mer at base p (KmerDict::findEntry -> KDef, ReadPather.h:104-145, BuildReadQGraph.cc:510-513):
// INDEX = false through the table and the k-mer's 32-B record, INDEX = true through the minimizer-sampled index (common.h)
template <bool INDEX, class RD>
__device__ inline bool dict_lookup(const PathArgs& A, const RD& rd, uint32_t p, IdxHit& ih) {
    if constexpr (INDEX) {
        uint64_t kl, kh;
        rd.bits120(p, kl, kh);
        return index_find(A.X, kl, kh, ih);
    } else {
        Kmer kc = read_kmer(rd, p);
        const bool r_ = kmer_canon(kc);
        uint4 kdef = make_uint4(0, 0, 0, 0);
        if (table_find_rec(A.table, A.mask, A.srec, kc, kmer_hash(kc), kdef) < 0) return false;
        ih.e = kdef.x & 0x7FFFFFFFu; ih.rc = r_ != (bool)(kdef.x >> 31);                            // CF<K>::isRC, CanonicalForm.h:84-91
        ih.off = kdef.y; ih.nk = kdef.w >> 8; ih.eo = (uint64_t)kdef.z | ((uint64_t)(kdef.w & 0xFFu) << 32);
        return true;
    }
}

// 60 bases of a unipath in PATH orientation from position j (< elen), LSB first: one unaligned 16-byte load of the packed edge
// stream (16 B of slack behind it).  Reverse-complemented edges: the 60 forward bases ENDING at the mirrored position, moved so that
// the mirrored base is the 60th, their 60 groups reversed and complemented.  Bits of bases beyond the edge are undefined.
__device__ inline void edge120(const uint8_t* __restrict__ ebits, uint64_t eo, uint32_t elen, bool rc, uint32_t j, uint64_t& lo, uint64_t& hi) {
    const uint32_t h = elen - 1 - j, back = h >= K - 1 ? K - 1 : h;
    const uint64_t pos = eo + (rc ? h - back : j);
    const U128u v = *reinterpret_cast<const U128u*>(ebits + (pos >> 2));
    const unsigned sh = 2 * (unsigned)(pos & 3);
    uint64_t a = sh ? (v.a >> sh) | (v.b << (64 - sh)) : v.a, b = v.b >> sh;
    if (rc) {
        const unsigned sl = 2 * (K - 1 - back);                                // 0 .. 118
        if (sl >= 64) { b = a << (sl - 64); a = 0; } else if (sl) { b = (b << sl) | (a >> (64 - sl)); a <<= sl; }
        // reverse the 60 groups of the 120-bit value b:a (reversing all 64 groups of 128 bits leaves them on top: >> 8), complement
        const uint64_t ra = rev2_64(b), rb_ = rev2_64(a);                      // rb_:ra = reversed 128 bits
        a = ~((ra >> 8) | (rb_ << 56)); b = ~(rb_ >> 8);
    }
    lo = a; hi = b;
}

__device__ inline uint32_t obj_kmers(const PathArgs& A, uint32_t o) { return A.edge_nk[A.obj_edge[o] >> 1]; }

// scoreLeftOverlap / scoreRightOverlap, ExtendReadPath.cc:15-109 (pDecay .2, mapQ2 20, leftOver 10).  The walk is base by base (the
// fp64 decay is sequential), but the bases come 32 (read) / 60 (edge object, from the packed edge stream) per load instead of one
// byte load each per step; a quality is fetched only at a mismatch.
template <class RD>
__device__ unsigned score_overlap(const PathArgs& A, const RD& rd, const uint8_t* q, uint32_t L, uint32_t start, uint32_t o, bool leftward) {
    const uint32_t oe = A.obj_edge[o], e = oe >> 1; const bool rc = oe & 1;
    const uint32_t elen = A.edge_nk[e] + (K - 1);
    const uint64_t eo = A.edge_off[e];
    const uint32_t nb = start, ne = elen - (K - 1), m = nb < ne ? nb : ne;
    unsigned qSum = 0, penalty = 0;
    uint32_t rlo = 0, elo = 0; uint64_t rw = 0, ewl = 0, ewh = 0; bool have = false;
    for (uint32_t j = 0; j < m; ++j) {
        const uint32_t rp = leftward ? start - 1 - j : L - start + j;          // read position
        const uint32_t ep = leftward ? elen - K - j : (K - 1) + j;             // position on the edge OBJECT (its own orientation)
        if (!have || rp < rlo || rp >= rlo + 32) { rlo = leftward ? (rp >= 31 ? rp - 31 : 0) : rp; rw = rd.bits64(rlo); }
        if (!have || ep < elo || ep >= elo + K) { elo = leftward ? (ep >= K - 1 ? ep - (K - 1) : 0) : ep; edge120(A.ebits, eo, elen, rc, elo, ewl, ewh); }
        have = true;
        const unsigned rbase = (unsigned)(rw >> (2 * (rp - rlo))) & 3u;
        const uint32_t ei = ep - elo;
        const unsigned ebase = (unsigned)((ei < 32 ? ewl >> (2 * ei) : ewh >> (2 * (ei - 32)))) & 3u;
        if (rbase != ebase) {
            unsigned qs = q[rp];
            penalty += (qs == 2 ? 20u : qs);
            qSum += penalty;
        } else if (penalty > 0) {
            double dp = (double)penalty;
            double pr = __dmul_rn(0.2, dp);              // penalty -= (pDecay*penalty): no FMA contraction
            penalty = (unsigned)__dsub_rn(dp, pr);
        }
    }
    qSum += 10u * (nb - m);
    return qSum;
}

// one extension attempt; leftward: ExtendReadPath.cc:124-230, rightward: :233-348
template <class RD>
__device__ bool extend_once(const PathArgs& A, bool leftward, uint64_t lastGap, uint32_t v, const RD& rd, const uint8_t* q,
                            uint32_t L, int32_t& pick) {
    const uint64_t* coff = leftward ? A.to_off : A.from_off;
    const int32_t* cand = leftward ? A.to_e : A.from_e;
    const int32_t* vd = leftward ? A.to_v : A.from_v;
    uint64_t c0 = coff[v], c1 = coff[v + 1];
    uint32_t nc = (uint32_t)(c1 - c0);
    uint32_t nlong = 0, nshort = 0; int32_t short_first = -1; bool short_same = true;
    for (uint32_t i = 0; i < nc; ++i) {
        int32_t d = vd[c0 + i];
        uint64_t ts = A.to_off[d + 1] - A.to_off[d], fs = A.from_off[d + 1] - A.from_off[d];
        bool hanging = leftward ? (ts == 0 && fs == 1) : (fs == 0 && ts == 1);
        bool lng = (uint64_t)obj_kmers(A, cand[c0 + i]) >= lastGap;
        if (lng) ++nlong;
        if (!lng && !hanging) {
            if (nshort == 0) short_first = d; else if (d != short_first) short_same = false;
            ++nshort;
        }
    }
    if (nc != 1 && nshort > 0) {
        if (nlong > 0) return false;
        if (!short_same) return false;
        uint64_t deg = leftward ? A.to_off[short_first + 1] - A.to_off[short_first] : A.from_off[short_first + 1] - A.from_off[short_first];
        if (deg != 1) return false;
    }
    int32_t least_edge = -1; unsigned least = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < nc; ++i) {
        int32_t d = vd[c0 + i];
        uint64_t ts = A.to_off[d + 1] - A.to_off[d], fs = A.from_off[d + 1] - A.from_off[d];
        bool hanging = leftward ? (ts == 0 && fs == 1) : (fs == 0 && ts == 1);
        if (!hanging || nc == 1) {
            unsigned s = score_overlap(A, rd, q, L, (uint32_t)lastGap, cand[c0 + i], leftward);
            if (s < least) { least_edge = cand[c0 + i]; least = s; }
        }
    }
    if (least_edge == -1 || (uint64_t)least > lastGap * 10) return false;
    pick = least_edge;
    return true;
}


// Everything behind the seed stage of one read, on its parts 0 .. np-1: the heuristics of path_reads_OMP (:845-918), pathPartsToReadPath
// (:804-827), ExtendReadPath left / right (ExtendReadPath.cc:115-348 with toRight := toLeft), the pathed / multipathed counters
// (:1319-1322, before FixPaths) and FixPaths (GapToyTools.cc:322-335).  Parts and path elements are reached through the caller's
// accessors (LDS + spill in the lane-per-read kernel, LDS alone in the wave-per-read kernel).  -> the path in [lo, hi), its offset.
template <class GP, class SP, class GB, class SB, class RD>
__device__ inline void finish_read(const PathArgs& A, const GP& getp, const SP& setp, const GB& getb, const SB& setb, const RD& rd, const uint8_t* q,
                                   uint32_t L, uint32_t np, uint32_t& lo, uint32_t& hi, int32_t& offset, uint32_t& plen,
                                   unsigned long long& my_pathed, unsigned long long& my_multi) {
    int64_t sumk = 0;                                                          // k-mers of the path's edges (ExtendReadPath.cc:243-249 sums them per attempt)
    // ---------------- heuristics :848-918
    {   // merge adjacent gaps (:865-868); hanging-seed deletion :849-862 is unreachable (vleft==vright)
        uint32_t w = 0;
        for (uint32_t j = 0; j < np; ++j) {
            uint4 pj = getp(j);
            if (part_gap(pj) && w > 0) {
                uint4 pw = getp(w - 1);
                if (part_gap(pw)) { pw.z += pj.z; setp(w - 1, pw); continue; }
            }
            if (w != j) setp(w, pj);
            ++w;
        }
        np = w;
    }
    if (np >= 3) {                                                        // :875-898
        uint32_t seeds = part_gap(getp(0)) ? 0 : 1;
        for (uint32_t j = 1; j + 1 < np; ++j) {
            uint4 pj = getp(j);
            if (!part_gap(pj)) { ++seeds; continue; }
            uint4 prev = getp(j - 1), next = getp(j + 1);
            uint32_t graphDist = next.y - (prev.y + prev.z);              // :467-474
            bool same = prev.x == next.x && part_rc(prev) == part_rc(next);
            if (!same) graphDist += part_elen(prev);
            int32_t d = (int32_t)(pj.z - graphDist);
            bool ok = (uint32_t)(d < 0 ? -d : d) <= 3u;
            if (ok && prev.x != next.x) {                                 // isJoinable :552-558: equal trailing 59-mers
                const uint32_t l1 = part_elen(prev) + (K - 1), l2 = part_elen(next) + (K - 1);
                uint64_t a0, a1, b0, b1;                                  // the last 59 bases of both unipaths in path orientation
                edge120(A.ebits, A.edge_off[prev.x], l1, part_rc(prev), l1 - (K - 1), a0, a1);
                edge120(A.ebits, A.edge_off[next.x], l2, part_rc(next), l2 - (K - 1), b0, b1);
                ok = a0 == b0 && ((a1 ^ b1) & ((1ull << (2 * (K - 1) - 64)) - 1)) == 0;
            }
            if (!ok) {
                if (seeds > 1) {
                    uint32_t tot = prev.z;
                    for (uint32_t qn = j; qn < np; ++qn) tot += getp(qn).z;
                    np = j - 1;
                    setp(np, make_gap(tot)); ++np;
                } else {
                    for (uint32_t qn = j + 1; qn < np; ++qn) pj.z += getp(qn).z;
                    setp(j, pj);
                    np = j + 1;
                }
                break;
            }
        }
    }
    {   // tail back-off :904-918
        uint4 lastp = getp(np - 1);
        if (part_gap(lastp) && np > 1) {
            uint4 l2 = getp(np - 2);
            if (l2.y == 0 && l2.z <= 5) { lastp.z += l2.z; np -= 2; setp(np, lastp); ++np; }
        } else if (!part_gap(lastp)) {
            if (lastp.y == 0 && lastp.z <= 5) setp(np - 1, make_gap(lastp.z));
        }
    }
    // ---------------- pathPartsToReadPath :804-827
    {
        bool have_last = false; uint32_t le = 0; bool lrc = false;
        for (uint32_t j = 0; j < np; ++j) {
            uint4 pj = getp(j);
            if (part_gap(pj)) continue;
            if (have_last && le == pj.x && lrc == part_rc(pj)) continue;
            setb(hi, part_rc(pj) ? A.revX[pj.x] : A.fwdX[pj.x]); ++hi;
            sumk += part_elen(pj);
            have_last = true; le = pj.x; lrc = part_rc(pj);
        }
        if (hi != lo) {
            uint4 p0 = getp(0);
            if (!part_gap(p0)) offset = (int32_t)p0.y;
            else offset = (int32_t)getp(1).y - (int32_t)p0.z;
        }
    }
#ifdef W2RAP_IDX_STATS
    const unsigned long long tx0 = __builtin_amdgcn_s_memtime();
#endif
    // ---------------- extension, ExtendReadPath.cc:115-120
    while (hi != lo && offset < 0) {                                       // leftward :124-230
        uint64_t lastGap = (uint64_t)(-(int64_t)offset);
        if (lastGap < 10) break;
        if (lo == 0) break;                                                // scratch exhausted (cannot happen: lastGap shrinks by >=1)
        int32_t pick;
        uint32_t v = (uint32_t)A.left[getb(lo)];
        if (!extend_once(A, true, lastGap, v, rd, q, L, pick)) break;
        const uint32_t pk = obj_kmers(A, pick);
        offset += (int32_t)pk; sumk += pk;
        --lo; setb(lo, pick);
    }
    while (hi != lo) {                                                     // rightward :233-348
        const int64_t g = (int64_t)L + offset - sumk - (int64_t)(K - 1);
        if (g < 10) break;
        if (hi >= A.pcap) break;
        int32_t pick;
        uint32_t v = (uint32_t)A.left[getb(hi - 1)];                       // sic: toRight is built with ToLeft (:838)
        if (!extend_once(A, false, (uint64_t)g, v, rd, q, L, pick)) break;
        setb(hi, pick); ++hi;
        sumk += obj_kmers(A, pick);
    }
#ifdef W2RAP_IDX_STATS
    if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(1))) IDX_STAT(6, __builtin_amdgcn_s_memtime() - tx0);
#endif
    plen = hi - lo;
    if (plen > 0) ++my_pathed;                                             // :1319-1322 (before FixPaths)
    if (plen > 2) ++my_multi;
    // ---------------- FixPaths, GapToyTools.cc:322-335 (the correct to_right)
    for (uint32_t j = lo; j + 1 < hi; ++j) {
        if (A.right[getb(j)] != A.left[getb(j + 1)]) { hi = j + 1; break; }
    }
    plen = hi - lo;
}

template <bool STAGED, bool LISTED, bool INDEX>
__global__ void __launch_bounds__(PATH_THREADS) __attribute__((amdgpu_waves_per_eu(7, 7))) k_path(PathArgs A) {
    static_assert(!(STAGED && LISTED), "listed reads are not contiguous: they are read from global memory");
    extern __shared__ __attribute__((aligned(16))) uint32_t s_rd[];          // [rd_dwords] the block's reads, back to back
    __shared__ uint4 s_parts[LP][PATH_THREADS];
    __shared__ int32_t s_path[PL][PATH_THREADS];
    __shared__ unsigned long long s_chunk;
    const unsigned tid = threadIdx.x, lane = tid & 63;
    const uint32_t T = A.T;
    const uint32_t tg = blockIdx.x * PATH_THREADS + tid;                     // this lane's spill column
    uint4* parts = A.parts + tg;
    int32_t* pbs = A.pbuf + tg;
    auto getp = [&](uint32_t j_) -> uint4 { return j_ < LP ? s_parts[j_][tid] : parts[(uint64_t)(j_ - LP) * T]; };
    auto setp = [&](uint32_t j_, const uint4& v_) { if (j_ < LP) s_parts[j_][tid] = v_; else parts[(uint64_t)(j_ - LP) * T] = v_; };
    // logical path position j (the path grows from pmid in both directions): a window of PL positions in LDS, the rest spilled
    const uint32_t pw0 = A.pmid - 1;
    auto getb = [&](uint32_t j_) -> int32_t { return j_ - pw0 < PL ? s_path[j_ - pw0][tid] : pbs[(uint64_t)j_ * T]; };
    auto setb = [&](uint32_t j_, int32_t v_) { if (j_ - pw0 < PL) s_path[j_ - pw0][tid] = v_; else pbs[(uint64_t)j_ * T] = v_; };
    unsigned long long my_pathed = 0, my_multi = 0;
    const uint64_t nchunks = (A.n - A.r_first + PATH_THREADS - 1) / PATH_THREADS;      // (reads r_first .. n-1; a list: entries 0 .. n-1)
    // (the chunk queue is ONE address: ~24 ns of its L2 channel's atomic unit per ticket, 195 k tickets per step -- a third of what it can serve)
    for (uint32_t iter = 0;; ++iter) {
        __syncthreads();                                                     // the previous chunk's LDS contents are no longer read
        if (tid == 0 && (iter & (PATH_TICKETS - 1u)) == 0) s_chunk = atomicAdd(&A.counters[0], (unsigned long long)PATH_TICKETS);
        __syncthreads();
        const uint64_t chunk = s_chunk + (iter & (PATH_TICKETS - 1u));
        if (chunk >= nchunks) break;
        const uint64_t r0 = A.r_first + chunk * PATH_THREADS;
        const uint32_t nr = (uint32_t)(A.n - r0 < PATH_THREADS ? A.n - r0 : PATH_THREADS);
        const bool live = tid < nr;
        const uint64_t r = LISTED ? (uint64_t)A.list[live ? r0 + tid : r0] : r0 + tid;
        const uint64_t bo = A.boff[live ? r : (LISTED ? r : r0)];
        bool deferred = false;
        uint32_t my_off = 0;
        if (STAGED) {
            // the chunk's packed bytes [boff[r0], boff[r0+nr]) -> LDS, whole dwords from the dword below the first byte (inside the
            // array: base pointers are at least dword aligned); the last, partial dword byte by byte (nothing behind the array is read)
            const uint64_t b_lo = A.boff[r0], b_hi = A.boff[r0 + nr], a0 = b_lo & ~3ull;
            const uint32_t nbytes = (uint32_t)(b_hi - a0), nfull = nbytes >> 2;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(A.bases + a0);
            for (uint32_t i = tid; i < nfull; i += PATH_THREADS) s_rd[i] = src[i];
            if (tid < 4) s_rd[nfull + 1 + tid] = 0;
            if (tid == 0) {
                uint32_t v = 0;
                for (uint32_t t2 = 0; t2 < (nbytes & 3); ++t2) v |= (uint32_t)A.bases[a0 + 4ull * nfull + t2] << (8 * t2);
                s_rd[nfull] = v;
            }
            my_off = (uint32_t)(bo - a0);
            __syncthreads();
        }
        uint32_t plen = 0, lo = A.pmid, hi = A.pmid;
        int32_t offset = 0;
        if (live) {
#ifdef W2RAP_IDX_STATS
            const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
            const uint8_t* rb = A.bases + bo;
            const uint8_t* q = A.quals + A.qoff[r];
            const uint32_t L = A.len[r];
            uint32_t np = 0;
            typename std::conditional<STAGED, RdLds, RdGlb>::type rd;
            if constexpr (STAGED) { rd.w = s_rd; rd.off = my_off; } else { rd.rb = rb; rd.nby = (L + 3) >> 2; }
            // ---------------- seed pathing, BRQ_Pather::path :500-550 (whole read, not good_len)
            if (L < K) { setp(0, make_gap(L)); np = 1; }
            else {
                uint32_t p = 0;
                const uint32_t end = L - K + 1;
                auto mer32_at = [&](uint32_t tt) -> FmerKey { return fmer_key(rd.bits64(tt)); };   // the 31-mer at base tt <= L-31 (common.h)
                auto f32_absent = [&](const FmerKey& k, unsigned long long w) -> bool { return (w & k.mask) != k.mask; };
                const uint32_t last = L - K, tmax = L - FMER;
                // Up to three 31-mers that contain base e, fetched together: how many k-mers from `cur` on do they prove absent?  (A
                // sequencing error at e spoils the k-mers e-59 .. e; the 31-mers at min(cur+29, e) and, 30 further, at e cover cur .. e.)
                auto probe3 = [&](uint32_t cur0, uint32_t e) -> uint32_t {
                    const uint32_t qmax = e < last ? e : last;
                    auto target = [&](uint32_t c_) -> uint32_t { uint32_t tt = c_ + FSPAN < e ? c_ + FSPAN : e; return tt > tmax ? tmax : tt; };
                    uint32_t cur = cur0;
                    const bool v0 = cur <= qmax; const uint32_t t0 = target(cur), q0 = t0 < last ? t0 : last; if (v0) cur = q0 + 1;
                    const bool v1 = v0 && cur <= qmax; const uint32_t t1 = target(cur), q1 = t1 < last ? t1 : last; if (v1) cur = q1 + 1;
                    const bool v2 = v1 && cur <= qmax; const uint32_t t2 = target(cur), q2 = t2 < last ? t2 : last;
                    FmerKey k0{0, 0}, k1{0, 0}, k2{0, 0}; unsigned long long w0 = 0, w1 = 0, w2 = 0;
                    if (v0) { k0 = mer32_at(t0); w0 = A.filter32[k0.word & A.f32mask]; }
                    if (v1) { k1 = mer32_at(t1); w1 = A.filter32[k1.word & A.f32mask]; }
                    if (v2) { k2 = mer32_at(t2); w2 = A.filter32[k2.word & A.f32mask]; }
                    const bool a0 = v0 && f32_absent(k0, w0), a1 = a0 && v1 && f32_absent(k1, w1), a2 = a1 && v2 && f32_absent(k2, w2);
                    return a2 ? q2 + 1 - cur0 : a1 ? q1 + 1 - cur0 : a0 ? q0 + 1 - cur0 : 0u;     // the 31-mer at t lies in the k-mers t-29 .. t
                };
                bool mism = false;                       // the previous part ended at a mismatching base (then k-mer p very likely does not exist)
                // the seed that ended there: its unipath, and the read / edge position of the mismatching base (same diagonal)
                uint32_t pv_e = 0, pv_elen = 0, pv_i = 0, pv_j = 0; uint64_t pv_eo = 0; bool pv_rc = false;
                bool at_end = false;                     // ... or it ended with the END of its unipath (the read goes on)
                int32_t pv_obj = 0;                      // its edge object
                while (p != end) {
                    if (!LISTED && A.part_budget && np >= A.part_budget) { deferred = true; break; }    // a many-part read: second pass
                    // Absence tests use the 31-mer filter (common.h): a read 31-mer that occurs in no edge proves every 60-mer around
                    // it absent.  Behind a mismatch at base e = p+59 k-mer p and the 59 behind it are most likely spoilt: the three
                    // probes start at k-mer p itself.  At the start of a read (or behind the end of an edge) the k-mer is probably
                    // there and the dictionary is asked directly.
                    bool hit = false; IdxHit ih{};                                  // the dictionary's answer for k-mer p (KDef, ReadPather.h:104-145)
                    auto lookup = [&](uint32_t pp) -> bool { return dict_lookup<INDEX>(A, rd, pp, ih); };
                    uint32_t gapLen = 0;                 // k-mers proven absent so far (slide one base at a time until one is found, :513-527)
                    bool probed = false;
                    const bool after_mism = mism;
                    if (mism && A.filter32) { gapLen = probe3(p, p + (K - 1)); p += gapLen; probed = gapLen != 0; }
                    mism = false;
                    // set when the k-mer is recognised WITHOUT the dictionary (see below): its unipath and offset in path orientation
                    bool diag_hit = false; uint32_t dg_off = 0;
                    bool ask_dict = !gapLen;
                    if (at_end) {
                        // The read ran off the END of a unipath: its next 60-mer begins with the 59-mer of that object's right vertex, and
                        // the out-edges of a vertex differ in their 60th base -- the read's base at p+59 names the one successor whose
                        // first k-mer this is (k_obj_table): two small records (L2 / Infinity Cache resident) instead of two dependent
                        // random sectors of the dictionary.  No successor for that base does NOT mean the k-mer is absent: adjacencies
                        // come from the contexts seen inside quality windows (:1062-1078), so a solid k-mer in the INTERIOR of another
                        // unipath can follow this end in a read's low-quality tail; the reference looks every read k-mer up (:510-513) and
                        // starts a part there, so the dictionary is asked on this (rare) miss.
                        const unsigned nb_ = (unsigned)(rd.bits64(p + (K - 1)) & 3u);
                        const int32_t o2 = A.otab[pv_obj].succ[nb_];
                        if (o2 >= 0) {
                            const ObjRec r2 = A.otab[o2];
                            diag_hit = true; dg_off = 0; ask_dict = false;
                            pv_e = r2.edge_rc >> 1; pv_rc = r2.edge_rc & 1u; pv_elen = r2.elen; pv_eo = (uint64_t)r2.eo_lo | ((uint64_t)r2.eo_hi << 32);
                        }
                        at_end = false;
                    }
                    if (ask_dict) {
                        SITE_STAT(1); hit = lookup(p);
                        if (!hit) { gapLen = 1; ++p; }
                    }
                    if (!hit && !diag_hit) {
                        uint32_t j = p + (K - 1);                                  // invariant: k-mer p ends at base j = p+59; j == L <=> no k-mer left
                        if (!probed && A.filter32 && j != L) {                     // the miss came from the dictionary: suspect base j-1
                            const uint32_t adv = probe3(p, j - 1);
                            gapLen += adv; p += adv; j += adv; probed = adv != 0;
                        }
                        if (probed && j != L) {                                    // the first k-mer behind the proven stretch: usually the hit that ends the gap
                            // Behind a single substitution the read goes on along the SAME unipath on the same diagonal.  Every 60-mer of a
                            // unipath is a solid k-mer whose dictionary entry names exactly that unipath and offset (buildEdges :287-301), so
                            // if the read's k-mer p equals the unipath's 60 bases at its diagonal position, the lookup's answer is known: one
                            // load of the edge stream next to the ones just compared, instead of two dependent random sectors (slot, record).
                            if (after_mism) {
                                const int64_t jp = (int64_t)pv_j + ((int64_t)p - (int64_t)pv_i);
                                if (jp >= 0 && jp + (int64_t)K <= (int64_t)pv_elen) {
                                    uint64_t el, eh, rl, rh;
                                    edge120(A.ebits, pv_eo, pv_elen, pv_rc, (uint32_t)jp, el, eh);
                                    rd.bits120(p, rl, rh);
                                    if (rl == el && ((rh ^ eh) & ((1ull << 56) - 1)) == 0) { diag_hit = true; dg_off = (uint32_t)jp; }
                                }
                            }
                            if (!diag_hit) {
                                SITE_STAT(2); hit = lookup(p);
                                if (!hit) { ++gapLen; ++p; ++j; }
                            }
                        }
                        // Whatever is left (the error was not where the mismatch suggested: start of the read, several errors, a false
                        // positive): a LADDER of 31-mers at p+29, p+14, p+7, p+3, p+1, p, fetched together -- the one at p+d proves
                        // p .. p+d absent if the spoiling base lies in it -- and the largest absent one is taken; only when no rung
                        // helps is k-mer p itself looked up in the dictionary.
                        while (!hit && !diag_hit && j != L) {
                            if (A.filter32) {
                                constexpr unsigned NR = 6;
                                const uint32_t rung[NR] = {FSPAN, 14, 7, 3, 1, 0};
                                FmerKey hr[NR]; unsigned long long wr[NR]; uint32_t tr[NR];
#pragma unroll
                                for (unsigned i = 0; i < NR; ++i) {
                                    tr[i] = p + rung[i] < tmax ? p + rung[i] : tmax;           // p <= last <= tmax
                                    hr[i] = mer32_at(tr[i]);
                                    wr[i] = A.filter32[hr[i].word & A.f32mask];
                                }
                                uint32_t adv = 0;
#pragma unroll
                                for (unsigned i = 0; i < NR; ++i)
                                    if (!adv && f32_absent(hr[i], wr[i])) adv = (tr[i] < last ? tr[i] : last) + 1 - p;
                                if (adv) { gapLen += adv; p += adv; j += adv; continue; }
                            }
                            SITE_STAT(3); hit = lookup(p);
                            if (hit) break;
                            ++gapLen; ++p; ++j;
                        }
                        setp(np, make_gap(gapLen)); ++np;
                    }
                    if (hit || diag_hit) {
                        // the index's answer = KDef (ReadPather.h:104-145) + the unipath's place and length; or the same facts from the diagonal
                        const uint32_t e = diag_hit ? pv_e : ih.e;
                        const bool rc = diag_hit ? pv_rc : ih.rc;                                 // CF<K>::isRC, CanonicalForm.h:84-91
                        const uint32_t elen = diag_hit ? pv_elen : ih.nk + (K - 1);
                        const uint64_t eo = diag_hit ? pv_eo : ih.eo;
                        uint32_t off = diag_hit ? (rc ? elen - dg_off - K : dg_off) : ih.off;    // offset of the k-mer on the FORWARD unipath
                        // matchLen (:341-350), 60 bases per step: the read's 120 bits against ONE 16-byte load of the packed edge stream
                        // in path orientation; the loads of two steps (120 bases: what is left of a PE150 read behind its first k-mer) are
                        // in flight together -- they do not depend on the outcome of the comparison, only the decision where to stop does
                        uint32_t len = 1, i = p + K;
                        uint32_t j = rc ? elen - off : off + K;                // position on the (forward or reverse-complemented) edge just past the k-mer
                        bool stop = false;
                        while (!stop && i < L && j < elen) {
                            uint64_t el[2], eh[2];
#pragma unroll
                            for (unsigned u = 0; u < 2; ++u) {
                                const uint32_t jj = j + K * u < elen ? j + K * u : elen - 1;
                                edge120(A.ebits, eo, elen, rc, jj, el[u], eh[u]);
                            }
#pragma unroll
                            for (unsigned u = 0; u < 2; ++u) {
                                if (stop || !(i < L && j < elen)) break;
                                uint32_t nn = L - i < elen - j ? L - i : elen - j; if (nn > K) nn = K;
                                uint64_t rl, rh;
                                rd.bits120(i, rl, rh);
                                uint64_t x = rl ^ el[u], y = rh ^ eh[u];
                                if (nn <= 32) { y = 0; if (nn < 32) x &= (1ull << (2 * nn)) - 1; }
                                else y &= (1ull << (2 * (nn - 32))) - 1;
                                if (x | y) {                                     // (i, j) move on to the differing base
                                    const uint32_t m = x ? (uint32_t)__builtin_ctzll(x) >> 1 : 32u + ((uint32_t)__builtin_ctzll(y) >> 1);
                                    len += m; i += m; j += m; stop = true;
                                }
                                else { len += nn; i += nn; j += nn; }
                            }
                        }
                        if (rc) off = (elen - off) - K;
                        mism = stop;                                             // stopped by a differing base (not by the end of the edge or read)
                        pv_e = e; pv_elen = elen; pv_eo = eo; pv_rc = rc; pv_i = i; pv_j = j;       // (i, j: the differing base, when stop)
                        at_end = !stop && j >= elen && i < L;                    // the unipath ended, the read did not
                        if (at_end) pv_obj = rc ? A.revX[e] : A.fwdX[e];
                        setp(np, make_uint4(e, off, len, (elen - K + 1) | (rc ? 0x80000000u : 0u))); ++np;
                        p += len;
                    }
                }
            }
#ifdef W2RAP_IDX_STATS
            const unsigned long long tf0 = __builtin_amdgcn_s_memtime();
            if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(1))) IDX_STAT(5, tf0 - ts0);
#endif
            if (!deferred) finish_read(A, getp, setp, getb, setb, rd, q, L, np, lo, hi, offset, plen, my_pathed, my_multi);
#ifdef W2RAP_IDX_STATS
            if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(1))) IDX_STAT(7, __builtin_amdgcn_s_memtime() - tf0);
#endif
        }
        if (!LISTED) {                                                          // the deferred reads of this wavefront -> list (one reservation)
            const unsigned long long dm = __ballot(deferred);
            if (dm) {
                unsigned long long dbase = 0;
                const int leader = __builtin_ctzll(dm);
                if ((int)lane == leader) dbase = atomicAdd(&A.counters[2], (unsigned long long)__builtin_popcountll(dm));
                dbase = __shfl(dbase, leader);
                const unsigned long long at = dbase + (unsigned)__builtin_popcountll(dm & ((1ull << lane) - 1));
                if (deferred && at < A.defer_cap) A.defer[at] = (uint32_t)r;
            }
        }
        // ---------------- the read's path to its place: <= 2 elements inline, longer ones in the pool (one reservation per wavefront)
        uint32_t need = plen > 2 ? plen : 0, incl = need;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
        const uint32_t wtot = __shfl(incl, 63);
        unsigned long long wbase = 0;
        if (wtot) { if (lane == 63) wbase = atomicAdd(&A.counters[1], (unsigned long long)wtot); wbase = __shfl(wbase, 63); }
        if (live && !deferred) {
            int2 rec = make_int2(0, 0);
            if (plen > 2) {
                const unsigned long long at = wbase + incl - need;
                rec = make_int2((int)(uint32_t)at, (int)(uint32_t)(at >> 32));
                if (at + plen <= A.pool_cap) for (uint32_t j = 0; j < plen; ++j) A.pool[at + j] = getb(lo + j);
            } else {
                if (plen > 0) rec.x = getb(lo);
                if (plen > 1) rec.y = getb(lo + 1);
            }
            A.plen[r] = plen; A.inl[r] = rec; A.poffset[r] = offset;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { my_pathed += __shfl_down(my_pathed, o); my_multi += __shfl_down(my_multi, o); }
    if (lane == 0) {
        const unsigned slot = 4 + 2 * ((blockIdx.x * 4 + (tid >> 6)) & (PCS - 1));
        if (my_pathed) atomicAdd(&A.counters[slot], my_pathed);
        if (my_multi) atomicAdd(&A.counters[slot + 1], my_multi);
    }
}

// ------------------------------------------------------------------------------------------------ the same, lanes refilled as they finish
// k_path gives every lane ONE read of its block's chunk and the wavefront runs as many part iterations as its slowest read needs: on the
// uniform workload a read has 1.9 parts on average and a wavefront's slowest ~5, on the planted one the 4 % many-part reads sit in 93 %
// of the wavefronts -- the lookups of a 50 M-read step execute with 8.7 of 64 lanes active (profiles/r05_index_ab.txt).  Here a lane that
// has finished the seed stage of its read PARKS it and takes the next read from the queue; the parked reads of a wavefront go through
// finish_read together (when a quarter of the lanes are parked, or nothing else is left to do).  The part iteration itself, finish_read and
// every result are k_path's, word for word; what differs is who does what when:
//  * a lane keeps its read in an LDS slot of its own (three unaligned 16-byte loads for a PE150 read) instead of the block staging 256
//    consecutive reads together -- no __syncthreads anywhere, the wavefronts of a block are independent;
//  * reads are handed out by tickets of as many consecutive reads as a wavefront has free lanes, from NSTR striped cursors (a cursor per
//    128-byte line; a block starts at stripe blockIdx % NSTR and moves on when its stripe is exhausted: one address would serialise ~1.6 M
//    tickets per step at 24 ns each);
//  * a read's results go to its own places as before (<= 2 elements inline, longer paths in the pool: one reservation per group of parked reads).
constexpr unsigned NSTR = 64;                     // striped read cursors
constexpr unsigned STRIPE_PAD = 16;               // u64 words per cursor: a 128-byte line each
template <bool INDEX>
__global__ void __launch_bounds__(PATH_THREADS) __attribute__((amdgpu_waves_per_eu(6, 6))) k_path_dyn(PathArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_rd[];          // [PATH_THREADS][A.slot_dwords]: the lane's read, from its first byte
    __shared__ uint4 s_parts[LP][PATH_THREADS];
    __shared__ int32_t s_path[PL][PATH_THREADS];
    const unsigned tid = threadIdx.x, lane = tid & 63;
    const uint32_t T = A.T;
    const uint32_t tg = blockIdx.x * PATH_THREADS + tid;                     // this lane's spill column
    uint4* parts = A.parts + tg;
    int32_t* pbs = A.pbuf + tg;
    auto getp = [&](uint32_t j_) -> uint4 { return j_ < LP ? s_parts[j_][tid] : parts[(uint64_t)(j_ - LP) * T]; };
    auto setp = [&](uint32_t j_, const uint4& v_) { if (j_ < LP) s_parts[j_][tid] = v_; else parts[(uint64_t)(j_ - LP) * T] = v_; };
    const uint32_t pw0 = A.pmid - 1;
    auto getb = [&](uint32_t j_) -> int32_t { return j_ - pw0 < PL ? s_path[j_ - pw0][tid] : pbs[(uint64_t)j_ * T]; };
    auto setb = [&](uint32_t j_, int32_t v_) { if (j_ - pw0 < PL) s_path[j_ - pw0][tid] = v_; else pbs[(uint64_t)j_ * T] = v_; };
    unsigned long long my_pathed = 0, my_multi = 0;
    const uint64_t n_all = A.n - A.r_first;
    const uint64_t per_stripe = (n_all + NSTR - 1) / NSTR;
    uint32_t* const slot = s_rd + (size_t)tid * A.slot_dwords;
    RdLds rd; rd.w = s_rd; rd.off = tid * A.slot_dwords * 4;
    // ---- the lane's read and the state of its seed stage (what k_path keeps across the iterations of its part loop)
    bool has = false, ready = false, deferred = false;
    uint64_t r = 0; uint32_t L = 0, end = 0, p = 0, np = 0;
    bool mism = false, at_end = false, pv_rc = false;
    uint32_t pv_e = 0, pv_elen = 0, pv_i = 0, pv_j = 0; uint64_t pv_eo = 0; int32_t pv_obj = 0;
    unsigned stripe = blockIdx.x % NSTR, stripes_done = 0;                   // wave-uniform
    for (;;) {
        // ---- (1) free lanes take the next reads of the stripe
        const unsigned long long free_m = __ballot(!has && !ready);
        if (free_m && stripes_done < NSTR) {
            const unsigned want = (unsigned)__builtin_popcountll(free_m);
            const int leader = __builtin_ctzll(free_m);
            unsigned long long base = 0;
            if ((int)lane == leader) base = atomicAdd(&A.stripes[(size_t)stripe * STRIPE_PAD], (unsigned long long)want);
            base = __shfl(base, leader);
            const uint64_t s_lo = (uint64_t)stripe * per_stripe, s_hi = s_lo + per_stripe < n_all ? s_lo + per_stripe : n_all;
            const uint64_t idx = s_lo + base + (unsigned)__builtin_popcountll(free_m & ((1ull << lane) - 1));
            if (!has && !ready && idx < s_hi) {
                r = A.r_first + idx;
                const uint64_t bo = A.boff[r];
                L = A.len[r];
                // the read's packed bytes -> the lane's slot, 16 bytes at a time (nothing behind the array is read)
                const uint32_t nby = (L + 3) >> 2;
                for (uint32_t o = 0; o < A.slot_dwords * 4; o += 16) {
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (o < nby + 8) {
                        if (bo + o + 16 <= A.bases_bytes) { const U128u w = *reinterpret_cast<const U128u*>(A.bases + bo + o); v = make_uint4((uint32_t)w.a, (uint32_t)(w.a >> 32), (uint32_t)w.b, (uint32_t)(w.b >> 32)); }
                        else {
                            uint32_t t4[4] = {0, 0, 0, 0};
                            for (uint32_t t2 = 0; t2 < 16 && bo + o + t2 < A.bases_bytes; ++t2) t4[t2 >> 2] |= (uint32_t)A.bases[bo + o + t2] << (8 * (t2 & 3));
                            v = make_uint4(t4[0], t4[1], t4[2], t4[3]);
                        }
                    }
                    *reinterpret_cast<uint4*>(slot + (o >> 2)) = v;
                }
                has = true; deferred = false; np = 0; p = 0; mism = false; at_end = false;
                if (L < K) { setp(0, make_gap(L)); np = 1; end = 0; }
                else end = L - K + 1;
            }
            if (s_lo + base + want >= s_hi) { ++stripes_done; stripe = (stripe + 1) % NSTR; }      // (wave-uniform: this stripe has no read left)
        }
        // ---- (2) one part iteration for every lane whose read has k-mers left (k_path's loop body)
        if (has && p != end) {
            if (A.part_budget && np >= A.part_budget) { deferred = true; p = end; }
            else {
                auto mer32_at = [&](uint32_t tt) -> FmerKey { return fmer_key(rd.bits64(tt)); };   // the 31-mer at base tt <= L-31 (common.h)
                auto f32_absent = [&](const FmerKey& k, unsigned long long w) -> bool { return (w & k.mask) != k.mask; };
                const uint32_t last = L - K, tmax = L - FMER;
                auto probe3 = [&](uint32_t cur0, uint32_t e) -> uint32_t {
                    const uint32_t qmax = e < last ? e : last;
                    auto target = [&](uint32_t c_) -> uint32_t { uint32_t tt = c_ + FSPAN < e ? c_ + FSPAN : e; return tt > tmax ? tmax : tt; };
                    uint32_t cur = cur0;
                    const bool v0 = cur <= qmax; const uint32_t t0 = target(cur), q0 = t0 < last ? t0 : last; if (v0) cur = q0 + 1;
                    const bool v1 = v0 && cur <= qmax; const uint32_t t1 = target(cur), q1 = t1 < last ? t1 : last; if (v1) cur = q1 + 1;
                    const bool v2 = v1 && cur <= qmax; const uint32_t t2 = target(cur), q2 = t2 < last ? t2 : last;
                    FmerKey k0{0, 0}, k1{0, 0}, k2{0, 0}; unsigned long long w0 = 0, w1 = 0, w2 = 0;
                    if (v0) { k0 = mer32_at(t0); w0 = A.filter32[k0.word & A.f32mask]; }
                    if (v1) { k1 = mer32_at(t1); w1 = A.filter32[k1.word & A.f32mask]; }
                    if (v2) { k2 = mer32_at(t2); w2 = A.filter32[k2.word & A.f32mask]; }
                    const bool a0 = v0 && f32_absent(k0, w0), a1 = a0 && v1 && f32_absent(k1, w1), a2 = a1 && v2 && f32_absent(k2, w2);
                    return a2 ? q2 + 1 - cur0 : a1 ? q1 + 1 - cur0 : a0 ? q0 + 1 - cur0 : 0u;
                };
                // Absence tests use the 31-mer filter (common.h): a read 31-mer that occurs in no edge proves every 60-mer around
                // it absent.  Behind a mismatch at base e = p+59 k-mer p and the 59 behind it are most likely spoilt: the three
                // probes start at k-mer p itself.  At the start of a read (or behind the end of an edge) the k-mer is probably
                // there and the dictionary is asked directly.
                bool hit = false; IdxHit ih{};                                  // the dictionary's answer for k-mer p (KDef, ReadPather.h:104-145)
                auto lookup = [&](uint32_t pp) -> bool { return dict_lookup<INDEX>(A, rd, pp, ih); };
                uint32_t gapLen = 0;                 // k-mers proven absent so far (slide one base at a time until one is found, :513-527)
                bool probed = false;
                const bool after_mism = mism;
                if (mism && A.filter32) { gapLen = probe3(p, p + (K - 1)); p += gapLen; probed = gapLen != 0; }
                mism = false;
                // set when the k-mer is recognised WITHOUT the dictionary (see below): its unipath and offset in path orientation
                bool diag_hit = false; uint32_t dg_off = 0;
                bool ask_dict = !gapLen;
                if (at_end) {
                    // The read ran off the END of a unipath: its next 60-mer begins with the 59-mer of that object's right vertex, and
                    // the out-edges of a vertex differ in their 60th base -- the read's base at p+59 names the one successor whose
                    // first k-mer this is (k_obj_table): two small records (L2 / Infinity Cache resident) instead of two dependent
                    // random sectors of the dictionary.  No successor for that base does NOT mean the k-mer is absent: adjacencies
                    // come from the contexts seen inside quality windows (:1062-1078), so a solid k-mer in the INTERIOR of another
                    // unipath can follow this end in a read's low-quality tail; the reference looks every read k-mer up (:510-513) and
                    // starts a part there, so the dictionary is asked on this (rare) miss.
                    const unsigned nb_ = (unsigned)(rd.bits64(p + (K - 1)) & 3u);
                    const int32_t o2 = A.otab[pv_obj].succ[nb_];
                    if (o2 >= 0) {
                        const ObjRec r2 = A.otab[o2];
                        diag_hit = true; dg_off = 0; ask_dict = false;
                        pv_e = r2.edge_rc >> 1; pv_rc = r2.edge_rc & 1u; pv_elen = r2.elen; pv_eo = (uint64_t)r2.eo_lo | ((uint64_t)r2.eo_hi << 32);
                    }
                    at_end = false;
                }
                if (ask_dict) {
                    SITE_STAT(1); hit = lookup(p);
                    if (!hit) { gapLen = 1; ++p; }
                }
                if (!hit && !diag_hit) {
                    uint32_t j = p + (K - 1);                                  // invariant: k-mer p ends at base j = p+59; j == L <=> no k-mer left
                    if (!probed && A.filter32 && j != L) {                     // the miss came from the dictionary: suspect base j-1
                        const uint32_t adv = probe3(p, j - 1);
                        gapLen += adv; p += adv; j += adv; probed = adv != 0;
                    }
                    if (probed && j != L) {                                    // the first k-mer behind the proven stretch: usually the hit that ends the gap
                        // Behind a single substitution the read goes on along the SAME unipath on the same diagonal.  Every 60-mer of a
                        // unipath is a solid k-mer whose dictionary entry names exactly that unipath and offset (buildEdges :287-301), so
                        // if the read's k-mer p equals the unipath's 60 bases at its diagonal position, the lookup's answer is known: one
                        // load of the edge stream next to the ones just compared, instead of two dependent random sectors (slot, record).
                        if (after_mism) {
                            const int64_t jp = (int64_t)pv_j + ((int64_t)p - (int64_t)pv_i);
                            if (jp >= 0 && jp + (int64_t)K <= (int64_t)pv_elen) {
                                uint64_t el, eh, rl, rh;
                                edge120(A.ebits, pv_eo, pv_elen, pv_rc, (uint32_t)jp, el, eh);
                                rd.bits120(p, rl, rh);
                                if (rl == el && ((rh ^ eh) & ((1ull << 56) - 1)) == 0) { diag_hit = true; dg_off = (uint32_t)jp; }
                            }
                        }
                        if (!diag_hit) {
                            SITE_STAT(2); hit = lookup(p);
                            if (!hit) { ++gapLen; ++p; ++j; }
                        }
                    }
                    // Whatever is left (the error was not where the mismatch suggested: start of the read, several errors, a false
                    // positive): a LADDER of 31-mers at p+29, p+14, p+7, p+3, p+1, p, fetched together -- the one at p+d proves
                    // p .. p+d absent if the spoiling base lies in it -- and the largest absent one is taken; only when no rung
                    // helps is k-mer p itself looked up in the dictionary.
                    while (!hit && !diag_hit && j != L) {
                        if (A.filter32) {
                            constexpr unsigned NR = 6;
                            const uint32_t rung[NR] = {FSPAN, 14, 7, 3, 1, 0};
                            FmerKey hr[NR]; unsigned long long wr[NR]; uint32_t tr[NR];
                #pragma unroll
                            for (unsigned i = 0; i < NR; ++i) {
                                tr[i] = p + rung[i] < tmax ? p + rung[i] : tmax;           // p <= last <= tmax
                                hr[i] = mer32_at(tr[i]);
                                wr[i] = A.filter32[hr[i].word & A.f32mask];
                            }
                            uint32_t adv = 0;
                #pragma unroll
                            for (unsigned i = 0; i < NR; ++i)
                                if (!adv && f32_absent(hr[i], wr[i])) adv = (tr[i] < last ? tr[i] : last) + 1 - p;
                            if (adv) { gapLen += adv; p += adv; j += adv; continue; }
                        }
                        SITE_STAT(3); hit = lookup(p);
                        if (hit) break;
                        ++gapLen; ++p; ++j;
                    }
                    setp(np, make_gap(gapLen)); ++np;
                }
                if (hit || diag_hit) {
                    // the index's answer = KDef (ReadPather.h:104-145) + the unipath's place and length; or the same facts from the diagonal
                    const uint32_t e = diag_hit ? pv_e : ih.e;
                    const bool rc = diag_hit ? pv_rc : ih.rc;                                 // CF<K>::isRC, CanonicalForm.h:84-91
                    const uint32_t elen = diag_hit ? pv_elen : ih.nk + (K - 1);
                    const uint64_t eo = diag_hit ? pv_eo : ih.eo;
                    uint32_t off = diag_hit ? (rc ? elen - dg_off - K : dg_off) : ih.off;    // offset of the k-mer on the FORWARD unipath
                    // matchLen (:341-350), 60 bases per step: the read's 120 bits against ONE 16-byte load of the packed edge stream
                    // in path orientation; the loads of two steps (120 bases: what is left of a PE150 read behind its first k-mer) are
                    // in flight together -- they do not depend on the outcome of the comparison, only the decision where to stop does
                    uint32_t len = 1, i = p + K;
                    uint32_t j = rc ? elen - off : off + K;                // position on the (forward or reverse-complemented) edge just past the k-mer
                    bool stop = false;
                    while (!stop && i < L && j < elen) {
                        uint64_t el[2], eh[2];
                #pragma unroll
                        for (unsigned u = 0; u < 2; ++u) {
                            const uint32_t jj = j + K * u < elen ? j + K * u : elen - 1;
                            edge120(A.ebits, eo, elen, rc, jj, el[u], eh[u]);
                        }
                #pragma unroll
                        for (unsigned u = 0; u < 2; ++u) {
                            if (stop || !(i < L && j < elen)) break;
                            uint32_t nn = L - i < elen - j ? L - i : elen - j; if (nn > K) nn = K;
                            uint64_t rl, rh;
                            rd.bits120(i, rl, rh);
                            uint64_t x = rl ^ el[u], y = rh ^ eh[u];
                            if (nn <= 32) { y = 0; if (nn < 32) x &= (1ull << (2 * nn)) - 1; }
                            else y &= (1ull << (2 * (nn - 32))) - 1;
                            if (x | y) {                                     // (i, j) move on to the differing base
                                const uint32_t m = x ? (uint32_t)__builtin_ctzll(x) >> 1 : 32u + ((uint32_t)__builtin_ctzll(y) >> 1);
                                len += m; i += m; j += m; stop = true;
                            }
                            else { len += nn; i += nn; j += nn; }
                        }
                    }
                    if (rc) off = (elen - off) - K;
                    mism = stop;                                             // stopped by a differing base (not by the end of the edge or read)
                    pv_e = e; pv_elen = elen; pv_eo = eo; pv_rc = rc; pv_i = i; pv_j = j;       // (i, j: the differing base, when stop)
                    at_end = !stop && j >= elen && i < L;                    // the unipath ended, the read did not
                    if (at_end) pv_obj = rc ? A.revX[e] : A.fwdX[e];
                    setp(np, make_uint4(e, off, len, (elen - K + 1) | (rc ? 0x80000000u : 0u))); ++np;
                    p += len;
                }
            }
        }
        if (has && p == end) { has = false; ready = true; }
        // ---- (3) the parked reads of the wavefront: finish_read and the results, together
        const unsigned long long ready_m = __ballot(ready), busy_m = __ballot(has);
        if (ready_m && ((unsigned)__builtin_popcountll(ready_m) >= A.fin_thresh || !busy_m)) {
            const bool fin = ready;
            uint32_t plen = 0, lo = A.pmid, hi = A.pmid;
            int32_t offset = 0;
            if (fin && !deferred) finish_read(A, getp, setp, getb, setb, rd, A.quals + A.qoff[r], L, np, lo, hi, offset, plen, my_pathed, my_multi);
            const unsigned long long dm = __ballot(fin && deferred);
            if (dm) {
                unsigned long long dbase = 0;
                const int leader = __builtin_ctzll(dm);
                if ((int)lane == leader) dbase = atomicAdd(&A.counters[2], (unsigned long long)__builtin_popcountll(dm));
                dbase = __shfl(dbase, leader);
                const unsigned long long at = dbase + (unsigned)__builtin_popcountll(dm & ((1ull << lane) - 1));
                if (fin && deferred && at < A.defer_cap) A.defer[at] = (uint32_t)r;
            }
            uint32_t need = (fin && !deferred && plen > 2) ? plen : 0, incl = need;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
            const uint32_t wtot = __shfl(incl, 63);
            unsigned long long wbase = 0;
            if (wtot) { if (lane == 63) wbase = atomicAdd(&A.counters[1], (unsigned long long)wtot); wbase = __shfl(wbase, 63); }
            if (fin && !deferred) {
                int2 rec = make_int2(0, 0);
                if (plen > 2) {
                    const unsigned long long at = wbase + incl - need;
                    rec = make_int2((int)(uint32_t)at, (int)(uint32_t)(at >> 32));
                    if (at + plen <= A.pool_cap) for (uint32_t j = 0; j < plen; ++j) A.pool[at + j] = getb(lo + j);
                } else {
                    if (plen > 0) rec.x = getb(lo);
                    if (plen > 1) rec.y = getb(lo + 1);
                }
                A.plen[r] = plen; A.inl[r] = rec; A.poffset[r] = offset;
            }
            if (fin) ready = false;
        }
        if (!__ballot(has || ready) && stripes_done >= NSTR) break;
    }
    for (int o = 32; o > 0; o >>= 1) { my_pathed += __shfl_down(my_pathed, o); my_multi += __shfl_down(my_multi, o); }
    if (lane == 0) {
        const unsigned slot_c = 4 + 2 * ((blockIdx.x * 4 + (tid >> 6)) & (PCS - 1));
        if (my_pathed) atomicAdd(&A.counters[slot_c], my_pathed);
        if (my_multi) atomicAdd(&A.counters[slot_c + 1], my_multi);
    }
}

// finish_read for the wavefront-per-read kernel, with the LANES on the parts: the joinability of every interior gap (two dependent loads of
// the edge stream each -- what made the sequential version 100 k clocks per many-part read), the compaction of the seeds into the path and
// FixPaths are evaluated for all parts at once; only the extension attempts (each depends on the one before) stay with lane 0.  Same
// results as finish_read on parts WITHOUT ADJACENT GAPS -- the run-length encoding of the per-position answers never has them, so the
// gap-merging pass (:865-868) has nothing to do.  parts / pth: this wave's LDS arrays.  All outputs are wave-uniform.
__device__ inline void wave_fence_() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <class T> __device__ inline T wave_sum_(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <class RD>
__device__ inline void finish_read_wave(const PathArgs& A, uint4* parts, int32_t* pth, const RD& rd, const uint8_t* q, uint32_t L, uint32_t np,
                                        uint32_t& lo, uint32_t& hi, int32_t& offset, uint32_t& plen, unsigned long long& my_pathed,
                                        unsigned long long& my_multi) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned long long lt = (1ull << lane) - 1;
    auto getb = [&](uint32_t j_) -> int32_t { return pth[j_]; };
    auto setb = [&](uint32_t j_, int32_t v_) { pth[j_] = v_; };
    int64_t sumk = 0;
    // ---------------- :875-898: the first interior gap that does not fit the graph ends the path there
    if (np >= 3) {
        uint32_t first_bad = NONE32;
        for (uint32_t j0 = 0; j0 < np && first_bad == NONE32; j0 += 64) {
            const uint32_t j = j0 + lane;
            bool bad = false;
            if (j >= 1 && j + 1 < np) {
                const uint4 pj = parts[j];
                if (part_gap(pj)) {
                    const uint4 prev = parts[j - 1], next = parts[j + 1];
                    uint32_t graphDist = next.y - (prev.y + prev.z);              // :467-474
                    const bool same = prev.x == next.x && part_rc(prev) == part_rc(next);
                    if (!same) graphDist += part_elen(prev);
                    const int32_t d = (int32_t)(pj.z - graphDist);
                    bool ok = (uint32_t)(d < 0 ? -d : d) <= 3u;
                    if (ok && prev.x != next.x && !part_gap(prev) && !part_gap(next)) {   // isJoinable :552-558: equal trailing 59-mers (a gap's neighbours are seeds)
                        const uint32_t l1 = part_elen(prev) + (K - 1), l2 = part_elen(next) + (K - 1);
                        uint64_t a0, a1, b0, b1;
                        edge120(A.ebits, A.edge_off[prev.x], l1, part_rc(prev), l1 - (K - 1), a0, a1);
                        edge120(A.ebits, A.edge_off[next.x], l2, part_rc(next), l2 - (K - 1), b0, b1);
                        ok = a0 == b0 && ((a1 ^ b1) & ((1ull << (2 * (K - 1) - 64)) - 1)) == 0;
                    }
                    bad = !ok;
                }
            }
            const unsigned long long bm = __ballot(bad);
            if (bm) first_bad = j0 + (uint32_t)__builtin_ctzll(bm);
        }
        if (first_bad != NONE32) {
            const uint32_t j = first_bad;
            uint32_t seeds = 0, zge = 0, zgt = 0;                                  // seeds before j; lengths from j on / behind j
            for (uint32_t j0 = 0; j0 < np; j0 += 64) {
                const uint32_t t = j0 + lane;
                const bool in = t < np;
                const uint4 pt = in ? parts[t] : make_gap(0);
                seeds += (uint32_t)__builtin_popcountll(__ballot(in && t < j && !part_gap(pt)));
                zge += wave_sum_<uint32_t>(in && t >= j ? pt.z : 0u);
                zgt += wave_sum_<uint32_t>(in && t > j ? pt.z : 0u);
            }
            const uint4 prev = parts[j - 1]; uint4 pj = parts[j];
            wave_fence_();
            if (seeds > 1) { np = j - 1; if (lane == 0) parts[np] = make_gap(prev.z + zge); ++np; }
            else { pj.z += zgt; if (lane == 0) parts[j] = pj; np = j + 1; }
            wave_fence_();
        }
    }
    {   // tail back-off :904-918
        uint4 lastp = parts[np - 1];
        if (part_gap(lastp) && np > 1) {
            const uint4 l2 = parts[np - 2];
            wave_fence_();
            if (l2.y == 0 && l2.z <= 5) { lastp.z += l2.z; np -= 2; if (lane == 0) parts[np] = lastp; ++np; }
        } else if (!part_gap(lastp)) {
            wave_fence_();
            if (lastp.y == 0 && lastp.z <= 5 && lane == 0) parts[np - 1] = make_gap(lastp.z);
        }
        wave_fence_();
    }
    // ---------------- pathPartsToReadPath :804-827: a seed enters the path unless the seed before it (across one gap) is on the same unipath
    for (uint32_t j0 = 0; j0 < np; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool in = j < np;
        const uint4 pj = in ? parts[j] : make_gap(0);
        bool emit = in && !part_gap(pj);
        if (emit && j >= 1) {
            uint4 pp = parts[j - 1];
            if (part_gap(pp) && j >= 2) pp = parts[j - 2];
            if (!part_gap(pp) && pp.x == pj.x && part_rc(pp) == part_rc(pj)) emit = false;
        }
        const unsigned long long em = __ballot(emit);
        if (emit) pth[hi + (uint32_t)__builtin_popcountll(em & lt)] = part_rc(pj) ? A.revX[pj.x] : A.fwdX[pj.x];
        sumk += wave_sum_<long long>(emit ? (long long)part_elen(pj) : 0ll);
        hi += (uint32_t)__builtin_popcountll(em);
    }
    wave_fence_();
    if (hi != lo) {
        const uint4 p0 = parts[0];
        offset = !part_gap(p0) ? (int32_t)p0.y : (int32_t)parts[1].y - (int32_t)p0.z;
    }
    // ---------------- extension, ExtendReadPath.cc:115-348: every attempt depends on the one before: lane 0
    if (lane == 0) {
        while (hi != lo && offset < 0) {                                       // leftward :124-230
            const uint64_t lastGap = (uint64_t)(-(int64_t)offset);
            if (lastGap < 10) break;
            if (lo == 0) break;
            int32_t pick;
            const uint32_t v = (uint32_t)A.left[getb(lo)];
            if (!extend_once(A, true, lastGap, v, rd, q, L, pick)) break;
            const uint32_t pk = obj_kmers(A, pick);
            offset += (int32_t)pk; sumk += pk;
            --lo; setb(lo, pick);
        }
        while (hi != lo) {                                                     // rightward :233-348
            const int64_t g = (int64_t)L + offset - sumk - (int64_t)(K - 1);
            if (g < 10) break;
            if (hi >= A.pcap) break;
            int32_t pick;
            const uint32_t v = (uint32_t)A.left[getb(hi - 1)];                 // sic: toRight is built with ToLeft (:838)
            if (!extend_once(A, false, (uint64_t)g, v, rd, q, L, pick)) break;
            setb(hi, pick); ++hi;
            sumk += obj_kmers(A, pick);
        }
    }
    lo = (uint32_t)__shfl((int)lo, 0); hi = (uint32_t)__shfl((int)hi, 0); offset = __shfl(offset, 0);
    wave_fence_();
    plen = hi - lo;
    if (lane == 0) { if (plen > 0) ++my_pathed; if (plen > 2) ++my_multi; }   // :1319-1322 (before FixPaths)
    // ---------------- FixPaths, GapToyTools.cc:322-335 (the correct to_right)
    for (uint32_t j0 = lo; j0 + 1 < hi; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool bad = j + 1 < hi && A.right[pth[j]] != A.left[pth[j + 1]];
        const unsigned long long bm = __ballot(bad);
        if (bm) { hi = j0 + (uint32_t)__builtin_ctzll(bm) + 1; break; }
    }
    plen = hi - lo;
}

// ---- the many-part reads, a WAVEFRONT per read.  A read that cuts into dozens of parts (high-copy repeats) makes dozens of dependent
// dictionary round trips in the lane-per-read kernel, and the 64 reads of a wavefront each follow their own control flow.  Here the lanes of
// a wavefront look up ALL k-mer positions of ONE read at once: the parts of BRQ_Pather::path (:500-550) are the run-length encoding of the
// per-position answers -- a seed that starts at p with (unipath, orientation, offset o) runs on exactly as long as position p+t answers
// (same unipath, same orientation, o+t), because every 60-mer of a unipath is in the dictionary with that unipath and offset (buildEdges
// :287-301) and a solid k-mer lies on one unipath only; a gap is a run of positions without an answer.  The parts go to LDS, lane 0 runs the
// sequential rest (finish_read) on them without another lane in its way.
constexpr unsigned WAVE_PARTS = 192;       // parts (= k-mer positions + 2) of a read this kernel holds: reads up to 249 bases
constexpr unsigned WAVE_PATH = 768;        // path elements (pcap of phase_path)
constexpr unsigned WAVE_SLAB = 2048;       // pool elements a wavefront reserves at a time (one atomic per slab instead of one per read)
template <bool PAR, bool INDEX>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) k_path_wave(PathArgs A) {
    __shared__ uint4 s_parts[4][WAVE_PARTS];
    __shared__ int32_t s_path[4][WAVE_PATH];
    __shared__ uint32_t s_start[4][WAVE_PARTS + 1];
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint4* parts = s_parts[wv]; int32_t* pth = s_path[wv]; uint32_t* starts = s_start[wv];
    auto getp = [&](uint32_t j_) -> uint4 { return parts[j_]; };
    auto setp = [&](uint32_t j_, const uint4& v_) { parts[j_] = v_; };
    auto getb = [&](uint32_t j_) -> int32_t { return pth[j_]; };
    auto setb = [&](uint32_t j_, int32_t v_) { pth[j_] = v_; };
    unsigned long long my_pathed = 0, my_multi = 0, slab_at = 0;
    unsigned long long t_hits = 0, t_fin = 0;                                   // shader clocks of the two stages (W2RAP_TRACE prints them)
    uint32_t slab_left = 0;
    const uint64_t nwaves = (uint64_t)gridDim.x * 4;
    for (uint64_t it = (uint64_t)blockIdx.x * 4 + wv; it < A.n; it += nwaves) {
        const uint64_t r = A.list ? (uint64_t)A.list[it] : it;
        const uint8_t* rb = A.bases + A.boff[r];
        const uint8_t* q = A.quals + A.qoff[r];
        const uint32_t L = A.len[r];
        RdGlb rd; rd.rb = rb; rd.nby = (L + 3) >> 2;
        uint32_t np = 0;
        const unsigned long long tc0 = __builtin_amdgcn_s_memtime();
        if (L < K) { if (lane == 0) parts[0] = make_gap(L); np = 1; }
        else {
            const uint32_t npos = L - K + 1;
            uint32_t carry_a = 0xFFFFFFFDu, carry_b = 0;                           // the answer at the last position of the previous 64
            for (uint32_t c0 = 0; c0 < npos; c0 += 64) {
                const uint32_t p = c0 + lane;
                const bool valid = p < npos;
                uint32_t a = NONE32, bd = 0, e = 0, offp = 0, wfield = 0;
                if (valid) {
                    IdxHit ih;
                    if (dict_lookup<INDEX>(A, rd, p, ih)) {
                        e = ih.e;
                        const bool rc = ih.rc;                                      // CF<K>::isRC, CanonicalForm.h:84-91
                        const uint32_t elen = ih.nk + (K - 1);
                        offp = rc ? (elen - ih.off) - K : ih.off;                  // offset of the k-mer on the unipath in PATH orientation
                        a = (e << 1) | (rc ? 1u : 0u);
                        bd = offp - p;                                             // the diagonal
                        wfield = (elen - K + 1) | (rc ? 0x80000000u : 0u);
                    }
                }
                const uint32_t pa = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_a, (int)a, 0x138, 0xF, 0xF, false);      // wave_shr:1 (lane 0: the carry)
                const uint32_t pb = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_b, (int)bd, 0x138, 0xF, 0xF, false);
                const bool same = a == pa && (a == NONE32 || bd == pb);
                const bool start = valid && (p == 0 || !same);
                const unsigned long long sm = __ballot(start);
                if (start) {
                    const uint32_t idx = np + (uint32_t)__builtin_popcountll(sm & ((1ull << lane) - 1));
                    parts[idx] = a == NONE32 ? make_uint4(NONE32, 0, 0, 0) : make_uint4(e, offp, 0, wfield);
                    starts[idx] = p;
                }
                np += (uint32_t)__builtin_popcountll(sm);
                carry_a = (uint32_t)__builtin_amdgcn_readlane((int)a, 63); carry_b = (uint32_t)__builtin_amdgcn_readlane((int)bd, 63);
            }
            if (lane == 0) starts[np] = npos;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (uint32_t j = lane; j < np; j += 64) parts[j].z = starts[j + 1] - starts[j];      // a seed's k-mers, a gap's missing k-mers
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const unsigned long long tc1 = __builtin_amdgcn_s_memtime();
        uint32_t plen = 0, lo = A.pmid, hi = A.pmid;
        int32_t offset = 0;
        if (PAR) finish_read_wave(A, parts, pth, rd, q, L, np, lo, hi, offset, plen, my_pathed, my_multi);
        if (lane == 0) {
            if (!PAR) finish_read(A, getp, setp, getb, setb, rd, q, L, np, lo, hi, offset, plen, my_pathed, my_multi);
            int2 rec = make_int2(0, 0);
            if (plen > 2) {
                if (slab_left < plen) { slab_at = atomicAdd(&A.counters[1], (unsigned long long)WAVE_SLAB); slab_left = WAVE_SLAB; }
                const unsigned long long at = slab_at;
                slab_at += plen; slab_left -= plen;
                rec = make_int2((int)(uint32_t)at, (int)(uint32_t)(at >> 32));
                if (at + plen <= A.pool_cap) for (uint32_t j = 0; j < plen; ++j) A.pool[at + j] = pth[lo + j];
            } else {
                if (plen > 0) rec.x = pth[lo];
                if (plen > 1) rec.y = pth[lo + 1];
            }
            A.plen[r] = plen; A.inl[r] = rec; A.poffset[r] = offset;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        t_hits += tc1 - tc0; t_fin += __builtin_amdgcn_s_memtime() - tc1;
    }
    if (lane == 0) {
        atomicAdd(&A.counters[0], t_hits); atomicAdd(&A.counters[3], t_fin);
        const unsigned slot = 4 + 2 * ((blockIdx.x * 4 + wv) & (PCS - 1));
        if (my_pathed) atomicAdd(&A.counters[slot], my_pathed);
        if (my_multi) atomicAdd(&A.counters[slot + 1], my_multi);
    }
}

// CSR of the read paths: element j of read r from its inline record or from the pool
__global__ void __launch_bounds__(256) k_path_gather(uint64_t n, const uint32_t* __restrict__ plen, const int2* __restrict__ inl,
                                                      const int32_t* __restrict__ pool, const uint64_t* __restrict__ path_off, int32_t* __restrict__ out) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t len = plen[r];
    if (!len) return;
    const int2 rec = inl[r];
    const uint64_t o = path_off[r];
    if (len <= 2) { out[o] = rec.x; if (len > 1) out[o + 1] = rec.y; }
    else {
        const uint64_t at = (uint64_t)(uint32_t)rec.x | ((uint64_t)(uint32_t)rec.y << 32);
        for (uint32_t j = 0; j < len; ++j) out[o + j] = pool[at + j];
    }
}

int phase_path(Ctx& c) {
    if (!c.graphed) { c.err = "path_reads called before build_graph"; return W2RAP_E_STATE; }
    c.pathed_done = false;
    hipStream_t st = c.stream;
    const uint64_t n = c.n;
    const uint32_t maxL = c.max_len;
    const uint32_t maxparts = (maxL >= K ? maxL - K + 1 : 1) + 2;
    const uint32_t pmid = maxL + 1;
    const uint32_t pcap = pmid + maxparts + maxL + 1;
    // LDS: the block's reads (256 x ceil(maxL/4) bytes + slack), 16 KB of parts, 4 KB of path elements; reads too long for that are
    // read from global memory.  Blocks are persistent: as many as fit on the GPU by LDS (at most 8 per CU).
    const uint64_t rd_bytes = (uint64_t)PATH_THREADS * ((maxL + 3) / 4) + 64;
    const bool staged = rd_bytes <= 96 * 1024 && !getenv("W2RAP_PATH_NO_STAGE");
    const uint32_t rd_dwords = staged ? (uint32_t)((rd_bytes + 3) / 4) : 0;
    const size_t lds_static = LP * PATH_THREADS * 16 + PL * PATH_THREADS * 4 + 64;
    const size_t lds_dyn = (size_t)rd_dwords * 4;
    unsigned per_cu = (unsigned)std::min<size_t>(8, (160 * 1024) / (lds_static + lds_dyn + 512));
    if (const char* v = getenv("W2RAP_PATH_BLOCKS")) per_cu = (unsigned)std::max(1, atoi(v));
    const uint64_t nchunks = (n + PATH_THREADS - 1) / PATH_THREADS;
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nchunks, (uint64_t)c.sm_count * per_cu));
    // k_path_dyn (lanes refilled as they finish; W2RAP_PATH_DYN=0: k_path): a lane's read in an LDS slot of its own, whole 16-byte pieces + 8 bytes
    // of slack for the accessors; six blocks per CU by LDS at PE150 (24 KB each), which is what its 6 waves per SIMD allow
    const uint32_t slot_dwords = (((maxL + 3) / 4 + 8 + 15) / 16) * 4;
    const size_t lds_dyn2 = (size_t)PATH_THREADS * slot_dwords * 4;
    const char* dv = getenv("W2RAP_PATH_DYN");
    const bool dyn = (dv ? atoi(dv) != 0 : true) && lds_static + lds_dyn2 <= 64 * 1024 && n >= 4096;
    const unsigned per_cu2 = (unsigned)std::min<size_t>(6, (160 * 1024) / (lds_static + lds_dyn2 + 512));
    const unsigned grid2 = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nchunks, (uint64_t)c.sm_count * per_cu2));
    const uint32_t T = std::max(grid, dyn ? grid2 : 0u) * PATH_THREADS;
    PathArgs A{};
    A.n = n;
    A.bases = c.d_bases; A.boff = c.d_boff; A.len = c.d_len; A.quals = c.d_quals; A.qoff = c.d_qoff;
    A.table = c.d_table; A.mask = c.tcap - 1; A.srec = c.d_srec;
    if (c.use_index) A.X = edge_index(c);
    const bool idx = c.use_index;
    A.filter32 = c.d_filter32; A.f32mask = c.f32words ? (uint32_t)(c.f32words - 1) : 0;
    A.codes = c.d_edge_codes; A.ebits = c.d_edge_bits; A.edge_off = c.d_edge_off; A.edge_nk = c.d_edge_nk;
    A.fwdX = c.d_fwdX; A.revX = c.d_revX; A.obj_edge = c.d_obj_edge; A.otab = c.d_otab; A.left = c.d_left; A.right = c.d_right;
    A.from_off = c.d_from_off; A.from_v = c.d_from_v; A.from_e = c.d_from_e;
    A.to_off = c.d_to_off; A.to_v = c.d_to_v; A.to_e = c.d_to_e;
    A.T = T; A.maxparts = maxparts; A.pcap = pcap; A.pmid = pmid; A.rd_dwords = rd_dwords;
    W2_ALLOC(A.parts, uint4, (uint64_t)(maxparts > LP ? maxparts - LP : 1) * T);
    W2_ALLOC(A.pbuf, int32_t, (uint64_t)pcap * T);
    W2_ALLOC(A.plen, uint32_t, n); W2_ALLOC(A.inl, int2, n);
    W2_ALLOC(c.d_path_offset, int32_t, n);
    W2_ALLOC(c.d_path_off, uint64_t, n + 1);
    W2_ALLOC(A.counters, unsigned long long, 4 + 2 * PCS);
    A.poffset = c.d_path_offset;
    A.slot_dwords = slot_dwords;
    A.fin_thresh = getenv("W2RAP_PATH_FIN") ? (uint32_t)std::max(1, atoi(getenv("W2RAP_PATH_FIN"))) : 16u;
    if (dyn) {
        W2_ALLOC(A.stripes, unsigned long long, (uint64_t)NSTR * STRIPE_PAD);
        uint64_t bb = 0;
        if (n) W2_HIP(hipMemcpyAsync(&bb, c.d_boff + n, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        A.bases_bytes = bb;
    }
    // The many-part reads go to the wave-per-read kernel when a read's parts and path fit its LDS arrays (reads up to 249 bases), the lanes on
    // the parts in both of its stages (W2RAP_PATH_WAVE: 0 = the lane-per-read listed kernel instead, 1 = the wave kernel with its second stage
    // on lane 0 -- exact and slower than either, kept for the comparison).  With it a read leaves the first pass at FOUR parts: planted
    // workload 44.0 -> 36.1 ms of pathing (12 parts: 40.5), the uniform one +0.14 ms (profiles/r04_planted_wave2.txt).
    const int wave_mode = getenv("W2RAP_PATH_WAVE") ? atoi(getenv("W2RAP_PATH_WAVE")) : 2;
    const bool wave_ok = maxparts <= WAVE_PARTS && pcap <= WAVE_PATH && (wave_mode == 1 || wave_mode == 2);
    A.part_budget = wave_ok && wave_mode == 2 ? 4 : 12;
    if (const char* v = getenv("W2RAP_PATH_BUDGET")) A.part_budget = (uint32_t)atoi(v);   // (0: everything in one pass)
    A.defer_cap = A.part_budget ? n : 0;
    W2_ALLOC(A.defer, uint32_t, A.defer_cap);
    uint64_t pool_cap = 2 * n + (1u << 20) + (wave_ok ? (uint64_t)c.sm_count * 4 * 4 * WAVE_SLAB : 0);
    if (const char* v = getenv("W2RAP_PATH_POOL")) pool_cap = (uint64_t)atoll(v);        // (tests: force the retry)
    unsigned long long h_all[4 + 2 * PCS];
    auto launch = [&](const PathArgs& B, bool listed) -> int {
        const uint64_t nch = (B.n - B.r_first + PATH_THREADS - 1) / PATH_THREADS;
        const unsigned g = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nch, (uint64_t)grid));
        if (listed && wave_ok) {
            // a wavefront per read: as many blocks as stay resident (four per CU by registers), reads dealt out by stride
            const unsigned gw = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((B.n + 3) / 4, (uint64_t)c.sm_count * 5));
            if (wave_mode == 2) { if (idx) LAUNCH(c, "k_path_wave", (k_path_wave<true, true>), dim3(gw), dim3(256), 0, B); else LAUNCH(c, "k_path_wave", (k_path_wave<true, false>), dim3(gw), dim3(256), 0, B); }
            else { if (idx) LAUNCH(c, "k_path_wave", (k_path_wave<false, true>), dim3(gw), dim3(256), 0, B); else LAUNCH(c, "k_path_wave", (k_path_wave<false, false>), dim3(gw), dim3(256), 0, B); }
        } else if (listed) {
            if (idx) LAUNCH(c, "k_path_deferred", (k_path<false, true, true>), dim3(g), dim3(PATH_THREADS), 0, B);
            else LAUNCH(c, "k_path_deferred", (k_path<false, true, false>), dim3(g), dim3(PATH_THREADS), 0, B);
        } else if (dyn) {
            W2_HIP(hipMemsetAsync(B.stripes, 0, (size_t)NSTR * STRIPE_PAD * 8, st));
            const unsigned g2 = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nch, (uint64_t)grid2));
            if (idx) {
                W2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_path_dyn<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dyn2));
                LAUNCH(c, "k_path", (k_path_dyn<true>), dim3(g2), dim3(PATH_THREADS), lds_dyn2, B);
            } else {
                W2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_path_dyn<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dyn2));
                LAUNCH(c, "k_path", (k_path_dyn<false>), dim3(g2), dim3(PATH_THREADS), lds_dyn2, B);
            }
        } else if (staged) {
            if (idx) {
                W2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_path<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dyn));
                LAUNCH(c, "k_path", (k_path<true, false, true>), dim3(g), dim3(PATH_THREADS), lds_dyn, B);
            } else {
                W2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_path<true, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dyn));
                LAUNCH(c, "k_path", (k_path<true, false, false>), dim3(g), dim3(PATH_THREADS), lds_dyn, B);
            }
        } else {
            if (idx) LAUNCH(c, "k_path", (k_path<false, false, true>), dim3(g), dim3(PATH_THREADS), 0, B);
            else LAUNCH(c, "k_path", (k_path<false, false, false>), dim3(g), dim3(PATH_THREADS), 0, B);
        }
        W2_HIP(hipGetLastError());
        return 0;
    };
    // Qualities that are still travelling (w2rap_step2_run's late upload): the first part of the reads is pathed as soon as ITS qualities
    // are up -- the extension's scores are all that reads them --, the rest behind the upload's end.
    uint64_t split = 0;
    if (c.quals_job) {
        W2_TRY(quals_wait_prefix(c, &split));
        if (split == 0 || split >= n) { split = 0; W2_TRY(quals_wait(c)); }
    }
    for (int attempt = 0;; ++attempt) {
        A.pool = c.alloc<int32_t>(pool_cap);
        if (!A.pool) return W2RAP_E_HIP;
        A.pool_cap = pool_cap;
        W2_HIP(hipMemsetAsync(A.counters, 0, (4 + 2 * PCS) * 8, st));
        for (int part = 0; part < (split ? 2 : 1); ++part) {
            A.r_first = part ? split : 0; A.n = split && !part ? split : n;
            if (part) {
                W2_TRY(quals_wait(c));                                   // everything is up (and c.stream waits for the last copy)
                W2_HIP(hipMemsetAsync(A.counters, 0, 8, st)); W2_HIP(hipMemsetAsync(A.counters + 2, 0, 8, st));      // chunk queue, deferred reads: again from 0
            }
            if (A.n > A.r_first) W2_TRY(launch(A, false));
            W2_HIP(hipMemcpyAsync(h_all, A.counters, sizeof(h_all), hipMemcpyDeviceToHost, st));
            W2_HIP(hipStreamSynchronize(st));
            if (h_all[2]) {                                  // the many-part reads, on their own
                PathArgs B = A;
                B.r_first = 0; B.n = h_all[2]; B.list = A.defer; B.part_budget = 0; B.defer = nullptr; B.defer_cap = 0;
                W2_HIP(hipMemsetAsync(A.counters, 0, 8, st));                 // the chunk queue starts again; pool cursor and statistics go on
                W2_TRY(launch(B, true));
                W2_HIP(hipMemcpyAsync(h_all, A.counters, sizeof(h_all), hipMemcpyDeviceToHost, st));
                W2_HIP(hipStreamSynchronize(st));
                if (wave_ok && getenv("W2RAP_TRACE"))
                    fprintf(stderr, "[w2rap] k_path_wave: %llu reads, shader clocks per read: positional lookups %.0f, sequential rest %.0f\n",
                            (unsigned long long)B.n, (double)h_all[0] / (double)B.n, (double)h_all[3] / (double)B.n);
            }
        }
        split = 0;                                           // (a second attempt finds every quality in place)
        A.r_first = 0; A.n = n;
        if (h_all[1] <= pool_cap) break;
        if (attempt) { c.err = "read pathing: path pool overflow after resizing"; return W2RAP_E_LIMIT; }
        c.release(A.pool);                               // longer paths than the pool was sized for: the exact need is known now
        pool_cap = h_all[1] + 1024 + (wave_ok ? (uint64_t)c.sm_count * 4 * 4 * WAVE_SLAB : 0);       // (the wave kernel reserves by slabs)
    }
#ifdef W2RAP_IDX_STATS
    {
        unsigned long long hs[8] = {0};
        (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_idx_stats), sizeof(hs));
        fprintf(stderr, "[w2rap] index lookups: %llu (lanes), %llu wavefront executions, %llu slots visited, %llu candidates, %llu hits; %llu reads; wave clocks (100 MHz): seeds %llu, finish_read %llu of which extension %llu\n", hs[0], hs[1], hs[2], hs[3], hs[4],
                (unsigned long long)n, hs[5], hs[7], hs[6]);
        unsigned long long z[8] = {0}, ss[8] = {0};
        (void)hipMemcpyFromSymbol(ss, HIP_SYMBOL(g_site_stats), sizeof(ss));
        fprintf(stderr, "[w2rap] lookup sites (lanes / executions): top %llu / %llu, behind a proven gap %llu / %llu, ladder %llu / %llu\n", ss[2], ss[3], ss[4], ss[5], ss[6], ss[7]);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_site_stats), z, sizeof(z));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_idx_stats), z, sizeof(z));
    }
#endif
    W2_TRY(exclusive_scan_u32_to_u64(c, A.plen, c.d_path_off, n));
    uint64_t total = 0;
    W2_HIP(hipMemcpyAsync(&total, c.d_path_off + n, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    int32_t* d_out = c.alloc<int32_t>(total + 1);
    if (!d_out) return W2RAP_E_HIP;
    if (n) LAUNCH(c, "k_path_gather", k_path_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, n, A.plen, A.inl, A.pool, c.d_path_off, d_out);
    W2_HIP(hipGetLastError());
    W2_HIP(hipStreamSynchronize(st));
    c.n_pathed = 0; c.n_multipathed = 0;
    for (unsigned i = 0; i < PCS; ++i) { c.n_pathed += h_all[4 + 2 * i]; c.n_multipathed += h_all[5 + 2 * i]; }
    c.d_path_edges = d_out; c.path_total = total;
    c.release(A.parts); c.release(A.pbuf); c.release(A.plen); c.release(A.inl); c.release(A.pool); c.release(A.counters); c.release(A.defer);
    c.pathed_done = true;
    return 0;
}

}  // namespace w2
