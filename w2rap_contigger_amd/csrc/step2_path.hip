// step2_path.hip -- phases a9..a12 of Step 2 on gfx950: one read per lane.
//   BRQ_Pather::path            BuildReadQGraph.cc:500-550   seed lookups + matchLen along the edge
//   path_reads_OMP heuristics   :845-918                     (hanging-seed deletion is dead code, Q9/Q12)
//   pathPartsToReadPath         :804-827
//   ExtendReadPath left/right   paths/long/ExtendReadPath.cc:15-348 (with toRight := toLeft, :836-838)
//   FixPaths                    paths/long/large/GapToyTools.cc:322-335
// Variable-length per-read state (parts, path) lives in lane-interleaved HBM scratch so
// that the lanes of a wavefront touch consecutive addresses; paths are compacted by a
// scan + copy per chunk of reads.
#include <algorithm>
#include <cstdlib>
#include "ctx.h"

namespace w2 {

constexpr unsigned PCS = 256;          // counter slots
struct PathArgs {
    // reads
    const uint8_t* bases; const uint64_t* boff; const uint32_t* len; const uint8_t* quals; const uint64_t* qoff;
    // dictionary + edges
    const Slot* table; uint64_t mask; const KRec* srec;
    const unsigned long long* filter32; uint32_t f32mask;
    const uint8_t* codes; const uint8_t* ebits; const uint64_t* edge_off; const uint32_t* edge_nk;
    const int32_t* fwdX; const int32_t* revX; const uint32_t* obj_edge;
    const int32_t* left; const int32_t* right;
    const uint64_t* from_off; const int32_t* from_v; const int32_t* from_e;
    const uint64_t* to_off; const int32_t* to_v; const int32_t* to_e;
    // scratch (lane interleaved: element j of thread t at [j*T + t])
    uint4* parts; int32_t* pbuf; uint32_t T; uint32_t maxparts; uint32_t pcap; uint32_t pmid;
    // per-read outputs of this chunk
    uint32_t* plen; uint32_t* pstart; int32_t* poffset;
    unsigned long long* counters;    // PCS slots of {pathed, multipathed} (a slot per block residue: one address would serialise
                                     // 1.5 M wave-level atomics at ~11 ns each), then 8 profile words
};

// part encoding: x = edge (unipath id) or 0xFFFFFFFF for a gap, y = offset, z = length, w = edge k-mers | rc<<31
__device__ inline bool part_gap(const uint4& p) { return p.x == NONE32; }
__device__ inline bool part_rc(const uint4& p) { return p.w >> 31; }
__device__ inline uint32_t part_elen(const uint4& p) { return p.w & 0x7FFFFFFFu; }
__device__ inline uint4 make_gap(uint32_t len) { return make_uint4(NONE32, 0, len, 0); }

// the 60-mer starting at base p of a .fastb-packed read (unaligned bytes)
__device__ inline Kmer read_kmer(const uint8_t* rb, uint32_t nbytes, uint32_t p) {
    uint32_t b0 = p >> 2, sh = 2 * (p & 3);
    uint64_t w0, w1 = 0;
    if (b0 + 16 <= nbytes) {                       // two unaligned 8-byte loads
        w0 = reinterpret_cast<const U64u*>(rb + b0)->v;
        w1 = reinterpret_cast<const U64u*>(rb + b0 + 8)->v;
    } else {                                       // tail of the read: stay inside its bytes
        w0 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { if (b0 + i < nbytes) w0 |= (uint64_t)rb[b0 + i] << (8 * i); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { if (b0 + 8 + i < nbytes) w1 |= (uint64_t)rb[b0 + 8 + i] << (8 * i); }
    }
    // 128-bit little-endian value >> sh
    uint64_t x0 = sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
    uint64_t x1 = w1 >> sh;
    uint64_t a = x0 & M60;                        // bases p..p+29 LSB-first
    uint64_t b = ((x0 >> 60) | (x1 << 4)) & M60;  // bases p+30..p+59
    return Kmer{lsb2msb60(a), lsb2msb60(b)};
}

__device__ inline unsigned obj_base_at(const PathArgs& A, uint32_t o, uint32_t t, uint32_t& len_out) {
    uint32_t oe = A.obj_edge[o], e = oe >> 1;
    uint32_t len = A.edge_nk[e] + (K - 1);
    len_out = len;
    uint64_t eo = A.edge_off[e];
    return (oe & 1) ? 3u - A.codes[eo + (len - 1 - t)] : A.codes[eo + t];
}
__device__ inline uint32_t obj_kmers(const PathArgs& A, uint32_t o) { return A.edge_nk[A.obj_edge[o] >> 1]; }

// scoreLeftOverlap / scoreRightOverlap, ExtendReadPath.cc:15-109 (pDecay .2, mapQ2 20, leftOver 10)
__device__ unsigned score_overlap(const PathArgs& A, const uint8_t* rb, const uint8_t* q, uint32_t L, uint32_t start, uint32_t o, bool leftward) {
    uint32_t oe = A.obj_edge[o], e = oe >> 1; bool rc = oe & 1;
    uint32_t elen = A.edge_nk[e] + (K - 1);
    const uint8_t* ec = A.codes + A.edge_off[e];
    uint32_t nb = start, ne = elen - (K - 1), m = nb < ne ? nb : ne;
    unsigned qSum = 0, penalty = 0;
    for (uint32_t j = 0; j < m; ++j) {
        uint32_t rp = leftward ? start - 1 - j : L - start + j;
        uint32_t ep = leftward ? elen - K - j : (K - 1) + j;
        unsigned rbase = packed_base(rb, rp);
        unsigned ebase = rc ? 3u - ec[elen - 1 - ep] : ec[ep];
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
__device__ bool extend_once(const PathArgs& A, bool leftward, uint64_t lastGap, uint32_t v, const uint8_t* rb, const uint8_t* q,
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
            unsigned s = score_overlap(A, rb, q, L, (uint32_t)lastGap, cand[c0 + i], leftward);
            if (s < least) { least_edge = cand[c0 + i]; least = s; }
        }
    }
    if (least_edge == -1 || (uint64_t)least > lastGap * 10) return false;
    pick = least_edge;
    return true;
}

// PROF: shader clocks of every wave per phase (gap slides, dictionary probes, edge compares, the rest of the seed loop,
// heuristics + path, extension + FixPaths), summed into counters[2..]; W2RAP_PATH_PROF=1 + W2RAP_TRACE=1 prints them
template <bool PROF, int ABL = 0>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) k_path(PathArgs A, uint64_t r0, uint64_t nreads) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nreads) return;
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, tp = PROF ? __builtin_amdgcn_s_memtime() : 0;
    auto tick = [&](int ph) { if (PROF) { const unsigned long long now = __builtin_amdgcn_s_memtime(); pt[ph] += now - tp; tp = now; } };
    const uint64_t r = r0 + t;
    const uint32_t T = A.T;
    const uint8_t* rb = A.bases + A.boff[r];
    const uint8_t* q = A.quals + A.qoff[r];
    const uint32_t L = A.len[r];
    uint4* parts = A.parts + t;            // parts[j*T] in HBM scratch for part LP and beyond; the first LP live in LDS
    constexpr unsigned LP = 4;             // (most reads end with <= 4 parts: seed, gap, seed, ...)
    __shared__ uint4 s_parts[LP][256];
    auto getp = [&](uint32_t j_) -> uint4 { return j_ < LP ? s_parts[j_][threadIdx.x] : parts[(uint64_t)j_ * T]; };
    auto setp = [&](uint32_t j_, const uint4& v_) { if (j_ < LP) s_parts[j_][threadIdx.x] = v_; else parts[(uint64_t)j_ * T] = v_; };
    uint32_t np = 0;
    // ---------------- seed pathing, BRQ_Pather::path :500-550 (whole read, not good_len)
    if (L < K) { setp(0, make_gap(L)); np = 1; }
    else {
        uint32_t p = 0, end = L - K + 1;
        const uint32_t nby_ = (L + 3) >> 2;
        // filter key of the 31-mer at base tt <= L-31 (common.h: 31 bases LSB first)
        auto mer32_at = [&](uint32_t tt) -> FmerKey {
            const uint32_t b0 = tt >> 2, sh = 2 * (tt & 3);
            uint64_t x = reinterpret_cast<const U64u*>(rb + b0)->v >> sh;              // bytes b0..b0+7 (+8 if sh > 2) hold bases tt..tt+30 <= L-1
            if (sh > 2) x |= (uint64_t)rb[b0 + 8] << (64 - sh);
            return fmer_key(x);
        };
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
        while (p != end) {
            // Absence tests use the 31-mer filter (common.h): a read 31-mer that occurs in no edge proves every 60-mer around
            // it absent.  Behind a mismatch at base e = p+59 k-mer p and the 59 behind it are most likely spoilt: the three
            // probes start at k-mer p itself.  At the start of a read (or behind the end of an edge) the k-mer is probably
            // there and the dictionary is asked directly.
            Kmer kc; bool r = false; int64_t s = -1;
            uint32_t gapLen = 0;                 // k-mers proven absent so far (slide one base at a time until one is found, :513-527)
            bool probed = false;
            if (mism && A.filter32 && ABL == 0) { gapLen = probe3(p, p + (K - 1)); p += gapLen; probed = gapLen != 0; }
            mism = false;
            if (!gapLen) {
                kc = read_kmer(rb, nby_, p); r = kmer_canon(kc);
                s = table_find_rec(A.table, A.mask, A.srec, kc, kmer_hash(kc));
                if (s < 0) { gapLen = 1; ++p; }
            }
            tick(1);
            if (s < 0) {
                uint32_t j = p + (K - 1);                                  // invariant: k-mer p ends at base j = p+59; j == L <=> no k-mer left
                if (ABL >= 2) { gapLen += L - j; p += L - j; j = L; }
                if (!probed && A.filter32 && ABL == 0 && j != L) {         // the miss came from the dictionary: suspect base j-1
                    const uint32_t adv = probe3(p, j - 1);
                    gapLen += adv; p += adv; j += adv; probed = adv != 0;
                }
                if (probed && j != L) {                                    // the first k-mer behind the proven stretch: usually the hit that ends the gap
                    kc = read_kmer(rb, nby_, p); r = kmer_canon(kc);
                    s = table_find_rec(A.table, A.mask, A.srec, kc, kmer_hash(kc));
                    if (s < 0) { ++gapLen; ++p; ++j; }
                }
                // Whatever is left (the error was not where the mismatch suggested: start of the read, several errors, a false
                // positive): a LADDER of 31-mers at p+29, p+14, p+7, p+3, p+1, p, fetched together -- the one at p+d proves
                // p .. p+d absent if the spoiling base lies in it -- and the largest absent one is taken; only when no rung
                // helps is k-mer p itself looked up in the dictionary.
                while (s < 0 && j != L) {
                    if (ABL >= 1) { gapLen += L - j; p += L - j; j = L; break; }
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
                    kc = read_kmer(rb, nby_, p); r = kmer_canon(kc);
                    s = table_find_rec(A.table, A.mask, A.srec, kc, kmer_hash(kc));
                    if (s >= 0) break;
                    ++gapLen; ++p; ++j;
                }
                setp(np, make_gap(gapLen)); ++np;
                tick(0);
            }
            if (s >= 0) {
                const uint4 kdef = A.srec[s].kdef;                      // KDef (ReadPather.h:104-145) + the unipath's place and length
                uint32_t e = kdef.x & 0x7FFFFFFFu, off = kdef.y;
                bool rc = r != (bool)(kdef.x >> 31);                    // CF<K>::isRC, CanonicalForm.h:84-91
                uint32_t elen = (kdef.w >> 8) + (K - 1);
                // matchLen (:341-350) 16 bases per step: read word vs edge word (forward), or vs the
                // reverse complement of the 16 edge bases ending at the mirrored position
                uint32_t len = 1, i = p + K;
                const uint64_t eo = (uint64_t)kdef.z | ((uint64_t)(kdef.w & 0xFFu) << 32);
                const uint32_t nby = (L + 3) >> 2;
                // 16-base words of the read (from pos < L) and of the edge in path orientation (from jj < elen), branch-free
                // and split into address / load / decode so that the eight loads of four steps are in flight together.
                // Read: the 8-byte load is pulled back inside the read's bytes near its end (the bits that drop out belong
                // to bases beyond the read).  Edge: forward, or the mirrored 16 bases reverse-complemented.
                auto read_addr = [&](uint32_t pos, uint32_t& sh) -> const uint8_t* {
                    const uint32_t b0 = pos >> 2, b0c = b0 + 8 <= nby ? b0 : nby - 8;      // nby >= 15 here (L >= K)
                    sh = 8 * (b0 - b0c) + 2 * (pos & 3);
                    return rb + b0c;
                };
                auto edge_addr = [&](uint32_t jj, uint32_t& sh, uint32_t& shl) -> const uint8_t* {
                    const uint32_t qhi = elen - 1 - jj, back = qhi >= 15 ? 15 : qhi;     // forward position mirrored to rc position jj
                    const uint64_t pos = eo + (rc ? qhi - back : jj);
                    sh = 2 * (uint32_t)(pos & 3); shl = rc ? 2 * (15 - back) : 0;
                    return A.ebits + (pos >> 2);
                };
                // matchLen (:341-350): four 16-base steps are fetched before the first of them is compared -- the loads of a
                // step do not depend on the outcome of the previous one, only the decision where to stop does
                uint32_t j = rc ? elen - off : off + K;                  // position on the (forward or reverse-complemented) edge just past the k-mer
                tick(1);
                bool stop = false;
                while (!stop && i < L && j < elen) {
                    uint32_t xs[4], rsh[4], esh[4], eshl[4];
                    const uint8_t *ra[4], *ea[4];
                    uint64_t rw[4], ew[4];
#pragma unroll
                    for (unsigned u = 0; u < 4; ++u) {
                        const uint32_t ii = i + 16 * u < L ? i + 16 * u : L - 1, jj = j + 16 * u < elen ? j + 16 * u : elen - 1;
                        ra[u] = read_addr(ii, rsh[u]); ea[u] = edge_addr(jj, esh[u], eshl[u]);
                    }
#pragma unroll
                    for (unsigned u = 0; u < 4; ++u) { rw[u] = reinterpret_cast<const U64u*>(ra[u])->v; ew[u] = reinterpret_cast<const U64u*>(ea[u])->v; }
#pragma unroll
                    for (unsigned u = 0; u < 4; ++u) {
                        const uint32_t w = (uint32_t)(ew[u] >> esh[u]);
                        xs[u] = (uint32_t)(rw[u] >> rsh[u]) ^ (rc ? rc32(w << eshl[u]) : w);
                    }
#pragma unroll
                    for (unsigned u = 0; u < 4; ++u) {
                        if (stop || !(i < L && j < elen)) break;
                        uint32_t n = L - i < elen - j ? L - i : elen - j; if (n > 16) n = 16;
                        uint32_t x = xs[u];
                        if (n < 16) x &= (1u << (2 * n)) - 1;
                        if (x) { len += (uint32_t)__builtin_ctz(x) >> 1; stop = true; }
                        else { len += n; i += n; j += n; }
                    }
                }
                if (rc) off = (elen - off) - K;
                mism = stop;                                             // stopped by a differing base (not by the end of the edge or read)
                tick(2);
                setp(np, make_uint4(e, off, len, (elen - K + 1) | (rc ? 0x80000000u : 0u))); ++np;
                p += len;
                tick(3);
            }
        }
    }
    tick(3);
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
                uint32_t l1 = part_elen(prev) + (K - 1), l2 = part_elen(next) + (K - 1);
                const uint8_t* e1 = A.codes + A.edge_off[prev.x];
                const uint8_t* e2 = A.codes + A.edge_off[next.x];
                bool rc1 = part_rc(prev), rc2 = part_rc(next);
                for (uint32_t i = 0; i < K - 1 && ok; ++i) {
                    unsigned b1 = rc1 ? 3u - e1[(K - 2) - i] : e1[l1 - (K - 1) + i];
                    unsigned b2 = rc2 ? 3u - e2[(K - 2) - i] : e2[l2 - (K - 1) + i];
                    ok = b1 == b2;
                }
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
        uint4 last = getp(np - 1);
        if (part_gap(last) && np > 1) {
            uint4 l2 = getp(np - 2);
            if (l2.y == 0 && l2.z <= 5) { last.z += l2.z; np -= 2; setp(np, last); ++np; }
        } else if (!part_gap(last)) {
            if (last.y == 0 && last.z <= 5) setp(np - 1, make_gap(last.z));
        }
    }
    // ---------------- pathPartsToReadPath :804-827
    int32_t* pb = A.pbuf + t;              // pb[j*T], logical path = pb[lo..hi)
    uint32_t lo = A.pmid, hi = A.pmid;
    int32_t offset = 0;
    {
        bool have_last = false; uint32_t le = 0; bool lrc = false;
        for (uint32_t j = 0; j < np; ++j) {
            uint4 pj = getp(j);
            if (part_gap(pj)) continue;
            if (have_last && le == pj.x && lrc == part_rc(pj)) continue;
            pb[(uint64_t)hi * T] = part_rc(pj) ? A.revX[pj.x] : A.fwdX[pj.x]; ++hi;
            have_last = true; le = pj.x; lrc = part_rc(pj);
        }
        if (hi != lo) {
            uint4 p0 = getp(0);
            if (!part_gap(p0)) offset = (int32_t)p0.y;
            else offset = (int32_t)getp(1).y - (int32_t)p0.z;
        }
    }
    tick(4);
    // ---------------- extension, ExtendReadPath.cc:115-120
    while (hi != lo && offset < 0) {                                       // leftward :124-230
        uint64_t lastGap = (uint64_t)(-(int64_t)offset);
        if (lastGap < 10) break;
        if (lo == 0) break;                                                // scratch exhausted (cannot happen: lastGap shrinks by >=1)
        int32_t pick;
        uint32_t v = (uint32_t)A.left[pb[(uint64_t)lo * T]];
        if (!extend_once(A, true, lastGap, v, rb, q, L, pick)) break;
        offset += (int32_t)obj_kmers(A, pick);
        --lo; pb[(uint64_t)lo * T] = pick;
    }
    while (hi != lo) {                                                     // rightward :233-348
        int64_t g = (int64_t)L + offset;
        for (uint32_t j = lo; j < hi; ++j) g -= obj_kmers(A, pb[(uint64_t)j * T]);
        g -= (K - 1);
        if (g < 10) break;
        if (hi >= A.pcap) break;
        int32_t pick;
        uint32_t v = (uint32_t)A.left[pb[(uint64_t)(hi - 1) * T]];        // sic: toRight is built with ToLeft (:838)
        if (!extend_once(A, false, (uint64_t)g, v, rb, q, L, pick)) break;
        pb[(uint64_t)hi * T] = pick; ++hi;
    }
    uint32_t plen = hi - lo;
    if (plen > 0) atomicAdd(&A.counters[2 * (blockIdx.x & (PCS - 1))], 1ull);       // :1319-1322 (before FixPaths)
    if (plen > 2) atomicAdd(&A.counters[2 * (blockIdx.x & (PCS - 1)) + 1], 1ull);
    // ---------------- FixPaths, GapToyTools.cc:322-335 (the correct to_right)
    for (uint32_t j = lo; j + 1 < hi; ++j) {
        if (A.right[pb[(uint64_t)j * T]] != A.left[pb[(uint64_t)(j + 1) * T]]) { hi = j + 1; break; }
    }
    A.plen[t] = hi - lo; A.pstart[t] = lo; A.poffset[r] = offset;
    tick(5);
    if (PROF && (threadIdx.x & 63) == 0) for (int i = 0; i < 6; ++i) atomicAdd(&A.counters[2 * PCS + i], pt[i]);
}

__global__ void __launch_bounds__(256) k_path_copy(uint32_t nreads, uint32_t T, const int32_t* __restrict__ pbuf,
                                                    const uint32_t* __restrict__ plen, const uint32_t* __restrict__ pstart,
                                                    const uint64_t* __restrict__ off, uint64_t base, uint64_t* __restrict__ path_off,
                                                    uint64_t r0, int32_t* __restrict__ out) {
    uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nreads) return;
    uint64_t o = base + off[t];
    path_off[r0 + t] = o;
    uint32_t n = plen[t], s = pstart[t];
    for (uint32_t j = 0; j < n; ++j) out[o + j] = pbuf[(uint64_t)(s + j) * T + t];
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
    // scratch budget ~26 GiB = 8 M lanes per launch for PE150: every launch ends with a tail of slow waves and a host
    // round trip for the chunk's path total, so fewer, larger launches pay (2 M lanes: 38.8 ms, 8 M: 34.8 ms, 16 M: same)
    uint64_t per_thread = (uint64_t)maxparts * 16 + (uint64_t)pcap * 4;
    const char* tv = getenv("W2RAP_PATH_LANES");             // lanes per launch (scratch: ~3 KB per lane for PE150)
    uint64_t T64 = tv ? (uint64_t)atoll(tv) : (26ull << 30) / per_thread;
    T64 = std::max<uint64_t>(1024, std::min<uint64_t>(T64, tv ? (1u << 25) : (1u << 23))) & ~255ull;
    if (T64 > ((n + 255) & ~255ull)) T64 = std::max<uint64_t>(256, (n + 255) & ~255ull);
    const uint32_t T = (uint32_t)T64;
    PathArgs A{};
    A.bases = c.d_bases; A.boff = c.d_boff; A.len = c.d_len; A.quals = c.d_quals; A.qoff = c.d_qoff;
    A.table = c.d_table; A.mask = c.tcap - 1; A.srec = c.d_srec;
    A.filter32 = c.d_filter32; A.f32mask = c.f32words ? (uint32_t)(c.f32words - 1) : 0;
    A.codes = c.d_edge_codes; A.ebits = c.d_edge_bits; A.edge_off = c.d_edge_off; A.edge_nk = c.d_edge_nk;
    A.fwdX = c.d_fwdX; A.revX = c.d_revX; A.obj_edge = c.d_obj_edge; A.left = c.d_left; A.right = c.d_right;
    A.from_off = c.d_from_off; A.from_v = c.d_from_v; A.from_e = c.d_from_e;
    A.to_off = c.d_to_off; A.to_v = c.d_to_v; A.to_e = c.d_to_e;
    A.T = T; A.maxparts = maxparts; A.pcap = pcap; A.pmid = pmid;
    W2_ALLOC(A.parts, uint4, (uint64_t)maxparts * T);
    W2_ALLOC(A.pbuf, int32_t, (uint64_t)pcap * T);
    W2_ALLOC(A.plen, uint32_t, T); W2_ALLOC(A.pstart, uint32_t, T);
    W2_ALLOC(c.d_path_offset, int32_t, n);
    W2_ALLOC(c.d_path_off, uint64_t, n + 1);
    W2_ALLOC(A.counters, unsigned long long, 2 * PCS + 8);
    A.poffset = c.d_path_offset;
    W2_HIP(hipMemsetAsync(A.counters, 0, (2 * PCS + 8) * 8, st));
    const bool prof = getenv("W2RAP_PATH_PROF") != nullptr;
    uint64_t* d_off = nullptr;
    W2_ALLOC(d_off, uint64_t, (uint64_t)T + 1);
    uint64_t cap = n * 2 + 1024, total = 0;
    int32_t* d_out = c.alloc<int32_t>(cap);
    if (!d_out) return W2RAP_E_HIP;
    for (uint64_t r0 = 0; r0 < n; r0 += T) {
        uint32_t nr = (uint32_t)std::min<uint64_t>(T, n - r0);
#ifdef W2RAP_TESTING
        const char* ab = test_hook("W2RAP_PATH_ABLATE") ? getenv("W2RAP_PATH_ABLATE") : nullptr;   // timing experiments only (results are wrong)
        if (ab && atoi(ab) == 1) LAUNCH(c, "k_path", (k_path<false, 1>), dim3((nr + 255) / 256), dim3(256), 0, A, r0, (uint64_t)nr);
        else if (ab && atoi(ab) == 2) LAUNCH(c, "k_path", (k_path<false, 2>), dim3((nr + 255) / 256), dim3(256), 0, A, r0, (uint64_t)nr);
        else
#endif
        if (prof) LAUNCH(c, "k_path", k_path<true>, dim3((nr + 255) / 256), dim3(256), 0, A, r0, (uint64_t)nr);
        else LAUNCH(c, "k_path", k_path<false>, dim3((nr + 255) / 256), dim3(256), 0, A, r0, (uint64_t)nr);
        W2_HIP(hipGetLastError());
        W2_TRY(exclusive_scan_u32_to_u64(c, A.plen, d_off, nr));
        uint64_t chunk = 0;
        W2_HIP(hipMemcpyAsync(&chunk, d_off + nr, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        if (total + chunk > cap) {
            uint64_t ncap = std::max(cap * 2, total + chunk + 1024);
            int32_t* d_new = c.alloc<int32_t>(ncap);
            if (!d_new) return W2RAP_E_HIP;
            W2_HIP(hipMemcpyAsync(d_new, d_out, total * 4, hipMemcpyDeviceToDevice, st));
            W2_HIP(hipStreamSynchronize(st));
            c.release(d_out); d_out = d_new; cap = ncap;
        }
        LAUNCH(c, "k_path_copy", k_path_copy, dim3((nr + 255) / 256), dim3(256), 0, nr, T, A.pbuf, A.plen, A.pstart, d_off, total,
                           c.d_path_off, r0, d_out);
        W2_HIP(hipGetLastError());
        total += chunk;
    }
    W2_HIP(hipMemcpyAsync(c.d_path_off + n, &total, 8, hipMemcpyHostToDevice, st));
    unsigned long long h_all[2 * PCS + 8], h_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    W2_HIP(hipMemcpyAsync(h_all, A.counters, sizeof(h_all), hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    for (unsigned i = 0; i < PCS; ++i) { h_cnt[0] += h_all[2 * i]; h_cnt[1] += h_all[2 * i + 1]; }
    for (unsigned i = 0; i < 6; ++i) h_cnt[2 + i] = h_all[2 * PCS + i];
    if (prof && getenv("W2RAP_TRACE")) {
        const double nw = (double)((n + 63) / 64);
        fprintf(stderr, "[w2rap] k_path clocks per wave: gap slides %.0f, seed probes %.0f, edge compares %.0f, seed-loop rest %.0f, heuristics+path %.0f, "
                        "extension+FixPaths %.0f\n", h_cnt[2] / nw, h_cnt[3] / nw, h_cnt[4] / nw, h_cnt[5] / nw, h_cnt[6] / nw, h_cnt[7] / nw);
    }
    c.n_pathed = h_cnt[0]; c.n_multipathed = h_cnt[1];
    c.d_path_edges = d_out; c.path_total = total;
    c.release(A.parts); c.release(A.pbuf); c.release(A.plen); c.release(A.pstart); c.release(A.counters); c.release(d_off);
    c.pathed_done = true;
    return 0;
}

}  // namespace w2
