// oracle/step2_oracle.cc -- TEST INFRASTRUCTURE ONLY.
//
// A single-threaded CPU restatement of the reference's Step-2 hot path
// (buildReadQGraph + FixPaths).  It is the checker the HIP path is compared
// against; nothing in the product (w2rap_contigger_amd/, bench.py's timed
// region) may call it.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg load this library.
//
// Parity status: PINNED.  The reference holds no golden vectors or tests for
// this path, so the pin is the reference itself run in the build
// container: oracle/Makefile compiles the unmodified reference translation
// units into oracle/_ref/ref_step2, tests/golden/make_golden.py runs it on the
// adversarial fixtures, and tests/test_oracle_vs_golden.py requires this
// restatement to reproduce small_K.freqs, .small_K.hbv and .small_K.paths
// byte-for-byte (edge order replayed from the reference's own output, because
// the reference's edge numbering is arbitrary -- SURVEY.md 8c).
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference/src).  No reference source text is copied.
//
// K-mer layout used here (ours, not the reference's): a 60-mer is two 60-bit
// words, hi = bases 0..29, lo = bases 30..59, base i of a word at bits
// 2*(29-i)+1 : 2*(29-i).  (hi,lo) unsigned order == lexicographic order on
// A<C<G<T == the reference's KMer<60> operator< (kmers/KMer.h:289-319).

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr unsigned K = 60;                       // BuildReadQGraph.cc:51
constexpr uint64_t M60 = (1ull << 60) - 1;

struct Kmer {
    uint64_t hi, lo;
    bool operator<(Kmer const& o) const { return hi != o.hi ? hi < o.hi : lo < o.lo; }
    bool operator==(Kmer const& o) const { return hi == o.hi && lo == o.lo; }
    bool operator!=(Kmer const& o) const { return !(*this == o); }
};

inline uint64_t rev2(uint64_t x) {   // reverse the 32 2-bit groups of a u64
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
}
inline uint64_t rc60(uint64_t w) { return rev2(~w & M60) >> 4; }   // RC of a 30-base word

inline Kmer kmer_rc(Kmer k) { return Kmer{rc60(k.lo), rc60(k.hi)}; }            // kmers/KMer.h:205-227
inline Kmer kmer_succ(Kmer k, unsigned b) {                                      // KMer.h:191-197 toSuccessor
    return Kmer{((k.hi << 2) | (k.lo >> 58)) & M60, ((k.lo << 2) | b) & M60};
}
inline Kmer kmer_pred(Kmer k, unsigned b) {                                      // KMer.h:199-203 toPredecessor
    return Kmer{(k.hi >> 2) | ((uint64_t)b << 58), (k.lo >> 2) | ((k.hi & 3) << 58)};
}
inline unsigned kmer_base(Kmer k, unsigned i) {
    return i < 30 ? (k.hi >> (2 * (29 - i))) & 3 : (k.lo >> (2 * (59 - i))) & 3;
}
inline Kmer kmer_from(const uint8_t* b) {
    Kmer k{0, 0};
    for (unsigned i = 0; i < 30; ++i) k.hi = (k.hi << 2) | b[i];
    for (unsigned i = 30; i < 60; ++i) k.lo = (k.lo << 2) | b[i];
    return k;
}
inline bool kmer_is_rev(Kmer k) { return kmer_rc(k) < k; }         // dna/CanonicalForm.h:58-66 (K even)
inline bool kmer_is_pal(Kmer k) { return kmer_rc(k) == k; }

// kmers/KMerContext.cc:18-36: rc of a context byte == bit reversal of the byte
inline uint8_t brev8(uint8_t c) {
    c = (c >> 4) | (c << 4);
    c = ((c >> 2) & 0x33) | ((c & 0x33) << 2);
    c = ((c >> 1) & 0x55) | ((c & 0x55) << 1);
    return c;
}
inline unsigned popc4(unsigned m) { return __builtin_popcount(m & 15); }
inline unsigned single(unsigned m) { return __builtin_ctz(m); }

struct Entry {          // one distinct canonical k-mer
    Kmer k;
    uint32_t count;     // min(255, occurrences)   BuildReadQGraph.cc:943-949
    uint8_t ctx;        // (pred mask << 4) | succ mask, in the canonical orientation
    int32_t edge;       // unipath id (-1 = unassigned)  kmers/ReadPather.h:104-145
    uint32_t off;       // k-mer offset on that edge
};

struct Part {           // BuildReadQGraph.cc:432-492 PathPart
    bool gap; int edge; bool rc; uint32_t off, len, elen;
};

struct Oracle {
    unsigned minQual = 7, minFreq = 4;
    // inputs (unpacked: one base code / one qual per byte)
    uint64_t n = 0;
    const uint8_t* bases = nullptr;
    const uint8_t* quals = nullptr;
    const uint64_t* roff = nullptr;

    std::vector<uint16_t> good_len;
    uint64_t n_instances = 0, n_distinct = 0;
    uint64_t hist[101];
    std::vector<Entry> solid;                 // sorted by k-mer
    std::vector<uint32_t> index;              // radix index on the top 20 bits of hi
    std::vector<std::vector<uint8_t>> edges;  // unipaths, canonical orientation, final order
    // HBV
    std::vector<std::vector<uint8_t>> objs;   // edge objects (fwd, rc, fwd, rc ...)
    std::vector<int32_t> fwdX, revX;
    std::vector<int32_t> left, right;         // per object
    uint64_t n_vertices = 0;
    std::vector<std::vector<int32_t>> from_v, from_e, to_v, to_e;
    // paths
    std::vector<int32_t> path_offset;
    std::vector<uint64_t> path_off;
    std::vector<int32_t> path_edges;
    std::vector<int32_t> path_edges_prefix;   // before FixPaths (debug)
    uint64_t pathed = 0, multipathed = 0;
    std::string err;

    // ---------------------------------------------------------------- a1
    // count_good_lengths, BuildReadQGraph.cc:962-987
    void goodLengths() {
        good_len.assign(n, 0);
        for (uint64_t r = 0; r < n; ++r) {
            const uint8_t* q = quals + roff[r];
            uint64_t L = roff[r + 1] - roff[r];
            unsigned good = 0;
            for (uint64_t i = L; i-- > 0;) {
                if (q[i] < minQual) good = 0;
                else if (++good == K) { good_len[r] = (uint16_t)(i + K); break; }
            }
        }
    }

    // ---------------------------------------------------------------- a2-a5
    struct Rec { Kmer k; uint8_t ctx; };
    static void pushRec(std::vector<Rec>& v, Kmer k, uint8_t c) {
        if (kmer_is_rev(k)) { k = kmer_rc(k); c = brev8(c); }    // BuildReadQGraph.cc:68-76,1069
        v.push_back(Rec{k, c});
    }
    void countKmers() {
        std::vector<Rec> recs;
        for (uint64_t r = 0; r < n; ++r) {
            unsigned len = good_len[r];
            if (!(len > K)) continue;                               // :1064 (strict)
            const uint8_t* b = bases + roff[r];
            Kmer k = kmer_from(b);
            pushRec(recs, k, (uint8_t)(1u << b[K]));                 // :1067 initialContext(succ)
            unsigned last = len - 1, itr = K;
            while (itr != last) {                                   // :1070-1075
                unsigned pred = kmer_base(k, 0);
                k = kmer_succ(k, b[itr]); ++itr;
                pushRec(recs, k, (uint8_t)((1u << (4 + pred)) | (1u << b[itr])));
            }
            unsigned pred = kmer_base(k, 0);                        // :1076 finalContext(pred)
            k = kmer_succ(k, b[last]);
            pushRec(recs, k, (uint8_t)(1u << (4 + pred)));
        }
        n_instances = recs.size();
        std::sort(recs.begin(), recs.end(), [](Rec const& a, Rec const& b) { return a.k < b.k; });   // :1081
        for (auto& h : hist) h = 0;
        solid.clear(); n_distinct = 0;
        for (size_t i = 0; i < recs.size();) {                      // collapse_entries :1002-1013
            size_t j = i; unsigned c = 0; uint8_t ctx = 0;
            while (j < recs.size() && recs[j].k == recs[i].k) { ctx |= recs[j].ctx; c = std::min(255u, c + 1); ++j; }
            ++n_distinct;
            ++hist[std::min(100u, c)];                              // :1097
            if (c >= minFreq) solid.push_back(Entry{recs[i].k, c, ctx, -1, 0});   // :1098-1100
            i = j;
        }
        buildIndex();
    }
    static constexpr unsigned IDX_BITS = 20;
    void buildIndex() {
        index.assign((1u << IDX_BITS) + 1, 0);
        for (auto const& e : solid) ++index[(e.k.hi >> (60 - IDX_BITS)) + 1];
        for (size_t i = 1; i < index.size(); ++i) index[i] += index[i - 1];
    }
    // KmerDict::findEntryCanonical, kmers/ReadPather.h:190-201
    Entry* findCanonical(Kmer k) {
        uint32_t b = (uint32_t)(k.hi >> (60 - IDX_BITS));
        auto beg = solid.begin() + index[b], end = solid.begin() + index[b + 1];
        auto it = std::lower_bound(beg, end, k, [](Entry const& e, Kmer const& kk) { return e.k < kk; });
        return (it != end && it->k == k) ? &*it : nullptr;
    }
    Entry* find(Kmer k) { return findCanonical(kmer_is_rev(k) ? kmer_rc(k) : k); }

    // ---------------------------------------------------------------- a6
    // KmerDict::recomputeAdjacencies, kmers/ReadPather.h:307,317-346
    void pruneAdjacency() {
        std::vector<uint8_t> nctx(solid.size());
        for (size_t i = 0; i < solid.size(); ++i) {
            Entry const& e = solid[i]; uint8_t c = e.ctx;
            for (unsigned b = 0; b < 4; ++b) {
                if ((c & (1u << b)) && !find(kmer_succ(e.k, b))) c &= ~(1u << b);
                if ((c & (1u << (4 + b))) && !find(kmer_pred(e.k, b))) c &= ~(1u << (4 + b));
            }
            nctx[i] = c;
        }
        for (size_t i = 0; i < solid.size(); ++i) solid[i].ctx = nctx[i];
    }

    // ---------------------------------------------------------------- a7
    // EdgeBuilder::lookup, BuildReadQGraph.cc:261-273: entry + context in the
    // orientation of the query k-mer
    Entry* lookupCtx(Kmer k, uint8_t* ctx) {
        Entry* e;
        if (kmer_is_rev(k)) { e = findCanonical(kmer_rc(k)); *ctx = e ? brev8(e->ctx) : 0; }
        else { e = findCanonical(k); *ctx = e ? e->ctx : 0; }
        if (!e) err = "oracle: neighbour lookup failed (ForceAssert BuildReadQGraph.cc:265)";
        return e;
    }
    bool upPossible(Kmer k, uint8_t ctx) {          // :192-202
        unsigned pm = ctx >> 4;
        if (popc4(pm) != 1) return false;
        Kmer p = kmer_pred(k, single(pm));
        if (kmer_is_pal(p)) return false;
        uint8_t c; if (!lookupCtx(p, &c)) return false;
        return popc4(c & 15) == 1;
    }
    bool downPossible(Kmer k, uint8_t ctx) {        // :204-214
        unsigned sm = ctx & 15;
        if (popc4(sm) != 1) return false;
        Kmer s = kmer_succ(k, single(sm));
        if (kmer_is_pal(s)) return false;
        uint8_t c; if (!lookupCtx(s, &c)) return false;
        return popc4(c >> 4) == 1;
    }
    // bvec::getCanonicalForm, feudal/BaseVec.h:326-327 -> dna/CanonicalForm.h:34-46
    // 0 FWD, 1 REV, 2 PALINDROME
    static int eform(std::vector<uint8_t> const& s) {
        size_t len = s.size();
        if (len & 1) return (s[len / 2] & 2) ? 1 : 0;
        for (size_t i = 0, j = len; i < j;) {
            unsigned f = s[i], r = s[--j] ^ 3u;
            if (f < r) return 0;
            if (r < f) return 1;
            ++i;
        }
        return 2;
    }
    static void rcSeq(std::vector<uint8_t>& s) {
        std::reverse(s.begin(), s.end());
        for (auto& b : s) b ^= 3;
    }
    std::vector<std::vector<Entry*>> edge_members;   // parallel to raw edge list
    void addEdge(std::vector<uint8_t>& seq, std::vector<Entry*>& mem) {       // :275-306
        if (eform(seq) == 1) { rcSeq(seq); std::reverse(mem.begin(), mem.end()); }
        int id = (int)edges.size();
        unsigned off = 0;
        for (Entry* e : mem) {
            if (e->edge != -1) err = "oracle: preoccupied kmers (BuildReadQGraph.cc:303)";
            e->edge = id; e->off = off++;
        }
        edges.push_back(seq);
        edge_members.push_back(mem);
    }
    void extend(Kmer k, uint8_t ctx, std::vector<uint8_t>& seq, std::vector<Entry*>& mem) {   // :234-259
        while (popc4(ctx & 15) == 1) {
            unsigned b = single(ctx & 15);
            Kmer nk = kmer_succ(k, b);
            if (kmer_is_pal(nk)) break;
            uint8_t c; Entry* e = lookupCtx(nk, &c);
            if (!e) return;
            if (popc4(c >> 4) != 1) break;
            seq.push_back((uint8_t)b); mem.push_back(e);
            k = nk; ctx = c;
        }
        if (eform(seq) != 1) addEdge(seq, mem);      // REV copy is produced from the other end
    }
    static void kmerBases(Kmer k, std::vector<uint8_t>& s) {
        s.resize(K);
        for (unsigned i = 0; i < K; ++i) s[i] = (uint8_t)kmer_base(k, i);
    }
    void buildEdges() {                              // buildEdges :314-339, buildEdge :104-115
        edges.clear(); edge_members.clear();
        std::vector<uint8_t> seq; std::vector<Entry*> mem;
        for (auto& ent : solid) {
            if (ent.edge != -1) continue;
            seq.clear(); mem.clear();
            if (kmer_is_pal(ent.k)) { kmerBases(ent.k, seq); mem.push_back(&ent); addEdge(seq, mem); }
            else if (upPossible(ent.k, ent.ctx)) {
                if (downPossible(ent.k, ent.ctx)) continue;
                Kmer r = kmer_rc(ent.k);              // extendUpstream :222-226
                kmerBases(r, seq); mem.push_back(&ent);
                extend(r, brev8(ent.ctx), seq, mem);
            } else if (downPossible(ent.k, ent.ctx)) {
                kmerBases(ent.k, seq); mem.push_back(&ent);   // extendDownstream :228-232
                extend(ent.k, ent.ctx, seq, mem);
            } else { kmerBases(ent.k, seq); mem.push_back(&ent); addEdge(seq, mem); }
            if (!err.empty()) return;
        }
        // smooth circles: simpleCircle :126-153 + canonicalizeCircle :156-180
        for (auto& ent : solid) {
            if (ent.edge != -1) continue;
            seq.clear(); mem.clear();
            kmerBases(ent.k, seq); mem.push_back(&ent);
            Kmer k = ent.k; uint8_t ctx = ent.ctx;
            while (true) {
                if (popc4(ctx >> 4) != 1 || popc4(ctx & 15) != 1) { err = "oracle: circle context (BuildReadQGraph.cc:133)"; return; }
                unsigned b = single(ctx & 15);
                k = kmer_succ(k, b);
                Entry* e = lookupCtx(k, &ctx);
                if (!e) return;
                if (e == &ent) break;
                if (e->edge != -1) { err = "oracle: failed to close circle (BuildReadQGraph.cc:141)"; return; }
                seq.push_back((uint8_t)b); mem.push_back(e);
            }
            // canonicalizeCircle
            size_t idx = 0;
            for (size_t i = 1; i < mem.size(); ++i) if (mem[i]->k < mem[idx]->k) idx = i;
            Kmer at = kmer_from(&seq[idx]);
            if (kmer_is_rev(at)) {                     // CF<K>::getForm(...) == REV
                rcSeq(seq); std::reverse(mem.begin(), mem.end());
                idx = seq.size() - idx - K;
            }
            if (idx) {
                std::vector<uint8_t> bv(seq.begin() + idx, seq.end());
                bv.insert(bv.end(), seq.begin() + (K - 1), seq.begin() + (K + idx - 1));
                seq = bv;
                std::rotate(mem.begin(), mem.begin() + idx, mem.end());
            }
            addEdge(seq, mem);
        }
    }
    // Reorder the unipaths.  The reference's order is arbitrary (a parallel
    // hash-set walk under a spin lock, :275-286, :317-321); we fix it either to
    // the order given by a hint (replay of a reference run) or to the
    // lexicographic order of the sequences (canonical mode).
    bool orderEdges(uint64_t n_hint, const uint8_t* hint, const uint64_t* hoff) {
        size_t E = edges.size();
        std::vector<size_t> perm(E);    // perm[new] = old
        if (hint) {
            if (n_hint != E) { err = "oracle: edge hint count " + std::to_string(n_hint) + " != " + std::to_string(E); return false; }
            std::vector<size_t> byseq(E);
            for (size_t i = 0; i < E; ++i) byseq[i] = i;
            std::sort(byseq.begin(), byseq.end(), [&](size_t a, size_t b) { return edges[a] < edges[b]; });
            for (size_t i = 0; i < E; ++i) {
                std::vector<uint8_t> s(hint + hoff[i], hint + hoff[i + 1]);
                auto it = std::lower_bound(byseq.begin(), byseq.end(), s, [&](size_t a, std::vector<uint8_t> const& v) { return edges[a] < v; });
                if (it == byseq.end() || edges[*it] != s) { err = "oracle: hinted edge " + std::to_string(i) + " not in our edge set"; return false; }
                perm[i] = *it;
            }
        } else {
            for (size_t i = 0; i < E; ++i) perm[i] = i;
            std::sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return edges[a] < edges[b]; });
        }
        std::vector<std::vector<uint8_t>> ne(E);
        std::vector<int> newid(E, -1);
        for (size_t i = 0; i < E; ++i) { ne[i] = edges[perm[i]]; if (newid[perm[i]] != -1) { err = "oracle: duplicate hinted edge"; return false; } newid[perm[i]] = (int)i; }
        edges.swap(ne);
        for (auto& e : solid) e.edge = newid[e.edge];
        edge_members.clear();
        return true;
    }

    // ---------------------------------------------------------------- a8
    // buildHBVFromEdges, paths/long/HBVFromEdges.cc:76-154
    struct End { uint64_t hash; std::vector<uint8_t> seq; uint32_t obj; bool distal; };
    void buildHBV() {
        objs.clear(); fwdX.assign(edges.size(), -1); revX.assign(edges.size(), -1);
        for (size_t i = 0; i < edges.size(); ++i) {           // :137-151 (object ids)
            fwdX[i] = (int)objs.size(); objs.push_back(edges[i]);
            if (eform(edges[i]) == 2) revX[i] = fwdX[i];
            else { revX[i] = (int)objs.size(); auto r = edges[i]; rcSeq(r); objs.push_back(r); }
        }
        std::vector<End> ends; ends.reserve(2 * objs.size());
        for (size_t o = 0; o < objs.size(); ++o)
            for (int d = 0; d < 2; ++d) {                       // :91-98
                End e; e.obj = (uint32_t)o; e.distal = d;
                auto const& s = objs[o];
                e.seq.assign(d ? s.end() - (K - 1) : s.begin(), d ? s.end() : s.begin() + (K - 1));
                uint64_t h = 14695981039346656037ull;           // math/Hash.h:26-35 FNV1a over base codes
                for (uint8_t b : e.seq) h = 1099511628211ull * (h ^ b);
                e.hash = h; ends.push_back(e);
            }
        std::stable_sort(ends.begin(), ends.end(), [](End const& a, End const& b) {   // :33-37,99
            if (a.hash != b.hash) return a.hash < b.hash;
            return a.seq < b.seq; });
        left.assign(objs.size(), -1); right.assign(objs.size(), -1);
        int64_t vid = 0;
        for (size_t i = 0; i < ends.size(); ++i) {               // :109-125
            if (i > 0 && !(ends[i - 1].hash == ends[i].hash && ends[i - 1].seq == ends[i].seq)) ++vid;
            (ends[i].distal ? right : left)[ends[i].obj] = (int32_t)vid;
        }
        n_vertices = ends.empty() ? 0 : (uint64_t)vid + 1;
        from_v.assign(n_vertices, {}); from_e.assign(n_vertices, {}); to_v.assign(n_vertices, {}); to_e.assign(n_vertices, {});
        for (size_t o = 0; o < objs.size(); ++o) {               // digraphE::AddEdge, graph/DigraphTemplate.h:1829-1839
            int v = left[o], w = right[o];
            size_t i = std::upper_bound(from_v[v].begin(), from_v[v].end(), w) - from_v[v].begin();
            from_v[v].insert(from_v[v].begin() + i, w); from_e[v].insert(from_e[v].begin() + i, (int)o);
            size_t j = std::upper_bound(to_v[w].begin(), to_v[w].end(), v) - to_v[w].begin();
            to_v[w].insert(to_v[w].begin() + j, v); to_e[w].insert(to_e[w].begin() + j, (int)o);
        }
    }

    // ---------------------------------------------------------------- a9
    // BRQ_Pather::path, BuildReadQGraph.cc:500-550
    void seedPath(const uint8_t* rd, uint32_t L, std::vector<Part>& parts) {
        parts.clear();
        if (L < K) { parts.push_back(Part{true, -1, false, 0, L, 0}); return; }
        uint32_t p = 0, end = L - K + 1;
        while (p != end) {
            Kmer kmer = kmer_from(rd + p);
            Entry* e = find(kmer);
            if (!e) {
                uint32_t gapLen = 1, j = p + K; ++p;
                while (j != L) {
                    kmer = kmer_succ(kmer, rd[j]); ++j;
                    if ((e = find(kmer))) break;
                    ++gapLen; ++p;
                }
                parts.push_back(Part{true, -1, false, 0, gapLen, 0});
            }
            if (e) {
                auto const& edge = edges[e->edge];
                uint32_t offset = e->off, len = 1;
                bool rc = std::memcmp(rd + p, &edge[offset], K) != 0;      // CF<K>::isRC, CanonicalForm.h:84-91
                if (!rc) {
                    uint32_t i = p + K, j = offset + K;
                    while (i < L && j < edge.size() && rd[i] == edge[j]) { ++i; ++j; ++len; }
                } else {
                    uint32_t ro = (uint32_t)edge.size() - offset;         // position in rc(edge) just past the k-mer
                    uint32_t i = p + K, j = ro;
                    while (i < L && j < edge.size() && rd[i] == (edge[edge.size() - 1 - j] ^ 3u)) { ++i; ++j; ++len; }
                    offset = ro - K;
                }
                parts.push_back(Part{false, e->edge, rc, offset, len, (uint32_t)edge.size() - K + 1});
                p += len;
            }
        }
    }
    // last 59 bases of the edge in path orientation; BRQ_Pather::isJoinable :552-558
    bool joinable(Part const& a, Part const& b) {
        if (a.edge == b.edge) return true;
        auto tail = [&](Part const& p, uint8_t* out) {
            auto const& e = edges[p.edge];
            if (!p.rc) std::memcpy(out, &e[e.size() - (K - 1)], K - 1);
            else for (unsigned i = 0; i < K - 1; ++i) out[i] = e[(K - 2) - i] ^ 3u;   // last 59 of rc(e)
        };
        uint8_t t1[K], t2[K]; tail(a, t1); tail(b, t2);
        return std::memcmp(t1, t2, K - 1) == 0;
    }
    // path_reads_OMP body, BuildReadQGraph.cc:845-920
    void heuristics(std::vector<Part>& parts) {
        // (i) hanging-seed deletion :849-862 is unreachable: toRight is built with
        // hbv.ToLeft (:838) so vleft==vright and ToSize(v)==0 && ToSize(v)>1 is false.
        std::vector<Part> np;
        for (auto const& p : parts) {                                     // :865-868
            if (p.gap && !np.empty() && np.back().gap) np.back().len += p.len;
            else np.push_back(p);
        }
        parts.swap(np);
        if (parts.size() >= 3) {                                          // :875-898
            size_t seeds = parts[0].gap ? 0 : 1;
            for (size_t j = 1; j + 1 < parts.size(); ++j) {
                if (!parts[j].gap) { ++seeds; continue; }
                Part const& prev = parts[j - 1]; Part const& next = parts[j + 1];
                uint32_t graphDist = next.off - (prev.off + prev.len);    // :467-474
                bool same = prev.edge == next.edge && prev.rc == next.rc;
                if (!same) graphDist += prev.elen;
                int32_t d = (int32_t)(parts[j].len - graphDist);
                bool conforming = (uint32_t)(d < 0 ? -d : d) <= 3u;
                if (!conforming || !joinable(prev, next)) {
                    if (seeds > 1) {
                        uint32_t tot = parts[j - 1].len;
                        for (size_t q = j; q < parts.size(); ++q) tot += parts[q].len;
                        parts.resize(j - 1);
                        parts.push_back(Part{true, -1, false, 0, tot, 0});
                    } else {
                        for (size_t q = j + 1; q < parts.size(); ++q) parts[j].len += parts[q].len;
                        parts.resize(j + 1);
                    }
                    break;
                }
            }
        }
        if (parts.back().gap && parts.size() > 1) {                      // :904-912
            Part const& l2 = parts[parts.size() - 2];
            if (l2.off == 0 && l2.len <= 5) {
                Part last = parts.back(); last.len += l2.len;
                parts.pop_back(); parts.pop_back(); parts.push_back(last);
            }
        } else if (!parts.back().gap) {                                   // :913-918
            Part& last = parts.back();
            if (last.off == 0 && last.len <= 5) last = Part{true, -1, false, 0, last.len, 0};
        }
    }
    // pathPartsToReadPath, BuildReadQGraph.cc:804-827
    void toReadPath(std::vector<Part> const& parts, int32_t& offset, std::vector<int32_t>& path) {
        path.clear();
        Part const* last = nullptr;
        for (auto const& p : parts) {
            if (p.gap) continue;
            if (last && last->edge == p.edge && last->rc == p.rc) continue;
            path.push_back(p.rc ? revX[p.edge] : fwdX[p.edge]);
            last = &p;
        }
        if (path.empty()) offset = 0;
        else if (!parts[0].gap) offset = (int32_t)parts[0].off;
        else offset = (int32_t)parts[1].off - (int32_t)parts[0].len;
    }

    // ---------------------------------------------------------------- a11
    // scoreLeftOverlap / scoreRightOverlap, paths/long/ExtendReadPath.cc:15-109
    // (pDecay .2, mapQ2 20, leftOver 10: ExtendReadPath.h:57-59)
    static unsigned score(const uint8_t* rd, const uint8_t* q, uint32_t L, uint32_t start,
                          std::vector<uint8_t> const& e, bool leftward) {
        unsigned qSum = 0, penalty = 0;
        uint32_t nb = start, ne = (uint32_t)e.size() - (K - 1), m = std::min(nb, ne);
        for (uint32_t j = 0; j < m; ++j) {
            unsigned rb, qb, eb;
            if (leftward) { rb = rd[start - 1 - j]; qb = q[start - 1 - j]; eb = e[e.size() - K - j]; }
            else { rb = rd[L - start + j]; qb = q[L - start + j]; eb = e[(K - 1) + j]; }
            if (rb != eb) { penalty += (qb == 2 ? 20u : qb); qSum += penalty; }
            else if (penalty > 0) {
                volatile double dp = (double)penalty;        // penalty -= (pDecay*penalty), no FMA
                volatile double prod = 0.2 * dp;
                penalty = (unsigned)(dp - prod);
            }
        }
        qSum += 10u * (nb - m);
        return qSum;
    }
    uint32_t elk(int o) const { return (uint32_t)objs[o].size() - K + 1; }
    // ExtendReadPath::attemptLeftwardExtension, ExtendReadPath.cc:124-230
    bool extendLeft(int32_t& offset, std::vector<int32_t>& path, const uint8_t* rd, const uint8_t* q, uint32_t L) {
        if (path.empty() || offset >= 0) return false;
        uint64_t lastGap = (uint64_t)(-(int64_t)offset);
        if (lastGap < 10) return false;
        int v = left[path.front()];
        auto const& cand = to_e[v]; auto const& src = to_v[v];
        return pick(cand, src, true, lastGap, offset, path, rd, q, L);
    }
    // ExtendReadPath::attemptRightwardExtension, ExtendReadPath.cc:233-348; "to_right" is
    // hbv.ToLeft (BuildReadQGraph.cc:838), reproduced deliberately.
    bool extendRight(int32_t& offset, std::vector<int32_t>& path, const uint8_t* rd, const uint8_t* q, uint32_t L) {
        if (path.empty()) return false;
        int64_t g = (int64_t)L + offset;
        for (int e : path) g -= elk(e);
        g -= (K - 1);
        if (g < 10) return false;
        int v = left[path.back()];       // sic
        auto const& cand = from_e[v]; auto const& dst = from_v[v];
        return pick(cand, dst, false, (uint64_t)g, offset, path, rd, q, L);
    }
    bool pick(std::vector<int32_t> const& cand, std::vector<int32_t> const& vd, bool leftward, uint64_t lastGap,
              int32_t& offset, std::vector<int32_t>& path, const uint8_t* rd, const uint8_t* q, uint32_t L) {
        size_t nc = cand.size();
        std::vector<char> hanging(nc, 0), lng(nc, 0);
        std::vector<int32_t> short_dest;
        for (size_t i = 0; i < nc; ++i) {
            size_t ts = to_e[vd[i]].size(), fs = from_e[vd[i]].size();
            if (leftward ? (ts == 0 && fs == 1) : (fs == 0 && ts == 1)) hanging[i] = 1;
            if ((uint64_t)elk(cand[i]) >= lastGap) lng[i] = 1;
            if (!lng[i] && !hanging[i]) short_dest.push_back(vd[i]);
        }
        if (nc != 1) {
            size_t nlong = 0; for (char c : lng) nlong += c;
            if (!short_dest.empty()) {
                if (nlong > 0) return false;
                std::sort(short_dest.begin(), short_dest.end());
                short_dest.erase(std::unique(short_dest.begin(), short_dest.end()), short_dest.end());
                if (short_dest.size() != 1) return false;
                size_t deg = leftward ? to_e[short_dest.back()].size() : from_e[short_dest.back()].size();
                if (deg != 1) return false;
            }
        }
        int least_edge = -1; unsigned least = 0xFFFFFFFFu;
        for (size_t i = 0; i < nc; ++i)
            if (!hanging[i] || nc == 1) {
                unsigned s = score(rd, q, L, (uint32_t)lastGap, objs[cand[i]], leftward);
                if (s < least) { least_edge = cand[i]; least = s; }
            }
        if (least_edge == -1 || (uint64_t)least > lastGap * 10) return false;
        if (leftward) { offset += (int32_t)elk(least_edge); path.insert(path.begin(), least_edge); }
        else path.push_back(least_edge);
        return true;
    }

    // ---------------------------------------------------------------- a9-a12
    void pathReads() {
        path_offset.assign(n, 0); path_off.assign(n + 1, 0); path_edges.clear(); path_edges_prefix.clear();
        pathed = multipathed = 0;
        std::vector<Part> parts; std::vector<int32_t> path;
        for (uint64_t r = 0; r < n; ++r) {
            const uint8_t* rd = bases + roff[r]; const uint8_t* q = quals + roff[r];
            uint32_t L = (uint32_t)(roff[r + 1] - roff[r]);
            seedPath(rd, L, parts);
            heuristics(parts);
            int32_t offset; toReadPath(parts, offset, path);
            while (extendLeft(offset, path, rd, q, L)) {}              // ExtendReadPath.cc:115-120
            while (extendRight(offset, path, rd, q, L)) {}
            if (path.size() > 0) ++pathed;                             // BuildReadQGraph.cc:1319-1322
            if (path.size() > 2) ++multipathed;
            // FixPaths, paths/long/large/GapToyTools.cc:322-335 (correct to_right)
            for (size_t i = 0; i + 1 < path.size(); ++i)
                if (right[path[i]] != left[path[i + 1]]) { path.resize(i + 1); break; }
            path_offset[r] = offset;
            path_edges.insert(path_edges.end(), path.begin(), path.end());
            path_off[r + 1] = path_edges.size();
        }
    }
};

}  // namespace

// ---------------------------------------------------------------------------
// C interface (ctypes).  All arrays are caller-visible until oracle_free().
extern "C" {

void* oracle_run(uint64_t n_reads, const uint8_t* bases, const uint8_t* quals, const uint64_t* read_off,
                 unsigned min_qual, unsigned min_freq,
                 uint64_t n_hint, const uint8_t* hint_bases, const uint64_t* hint_off,
                 int stop_after /*0 all, 1 table only, 2 graph only*/) {
    Oracle* o = new Oracle;
    o->n = n_reads; o->bases = bases; o->quals = quals; o->roff = read_off;
    o->minQual = min_qual; o->minFreq = min_freq;
    o->goodLengths();
    o->countKmers();
    if (stop_after == 1) return o;
    o->pruneAdjacency();
    o->buildEdges();
    if (!o->err.empty()) return o;
    if (!o->orderEdges(n_hint, hint_bases, hint_off)) return o;
    o->buildHBV();
    if (stop_after == 2) return o;
    o->pathReads();
    return o;
}
const char* oracle_error(void* h) { auto* o = (Oracle*)h; return o->err.empty() ? nullptr : o->err.c_str(); }
void oracle_free(void* h) { delete (Oracle*)h; }

// sizes: [0] n_instances [1] n_distinct [2] n_solid [3] n_edges [4] n_objs [5] n_vertices
//        [6] total path edges [7] pathed [8] multipathed [9] total edge bases [10] total obj bases
void oracle_sizes(void* h, uint64_t* out) {
    auto* o = (Oracle*)h;
    out[0] = o->n_instances; out[1] = o->n_distinct; out[2] = o->solid.size(); out[3] = o->edges.size();
    out[4] = o->objs.size(); out[5] = o->n_vertices; out[6] = o->path_edges.size();
    out[7] = o->pathed; out[8] = o->multipathed;
    uint64_t eb = 0; for (auto& e : o->edges) eb += e.size(); out[9] = eb;
    uint64_t ob = 0; for (auto& e : o->objs) ob += e.size(); out[10] = ob;
}
void oracle_good_len(void* h, uint16_t* out) { auto* o = (Oracle*)h; std::memcpy(out, o->good_len.data(), o->n * 2); }
void oracle_hist(void* h, uint64_t* out) { std::memcpy(out, ((Oracle*)h)->hist, 101 * 8); }
// solid table sorted by k-mer: hi, lo (60-bit words), count, ctx (after pruning if the run got that far), edge, off
void oracle_table(void* h, uint64_t* hi, uint64_t* lo, uint8_t* count, uint8_t* ctx, int32_t* edge, uint32_t* off) {
    auto* o = (Oracle*)h;
    for (size_t i = 0; i < o->solid.size(); ++i) {
        auto const& e = o->solid[i];
        hi[i] = e.k.hi; lo[i] = e.k.lo; count[i] = (uint8_t)e.count; ctx[i] = e.ctx; edge[i] = e.edge; off[i] = e.off;
    }
}
static void flat(std::vector<std::vector<uint8_t>> const& v, uint8_t* b, uint64_t* off) {
    uint64_t p = 0; off[0] = 0;
    for (size_t i = 0; i < v.size(); ++i) { std::memcpy(b + p, v[i].data(), v[i].size()); p += v[i].size(); off[i + 1] = p; }
}
void oracle_edges(void* h, uint8_t* b, uint64_t* off) { flat(((Oracle*)h)->edges, b, off); }
void oracle_objs(void* h, uint8_t* b, uint64_t* off, int32_t* left, int32_t* right, int32_t* fwdX, int32_t* revX) {
    auto* o = (Oracle*)h; flat(o->objs, b, off);
    std::memcpy(left, o->left.data(), o->left.size() * 4); std::memcpy(right, o->right.data(), o->right.size() * 4);
    std::memcpy(fwdX, o->fwdX.data(), o->fwdX.size() * 4); std::memcpy(revX, o->revX.data(), o->revX.size() * 4);
}
// adjacency in CSR form; which: 0 from_ (targets), 1 from_edge_obj_, 2 to_ (sources), 3 to_edge_obj_
void oracle_adj(void* h, int which, uint64_t* off, int32_t* vals) {
    auto* o = (Oracle*)h;
    auto const& a = which == 0 ? o->from_v : which == 1 ? o->from_e : which == 2 ? o->to_v : o->to_e;
    uint64_t p = 0; off[0] = 0;
    for (size_t v = 0; v < a.size(); ++v) { for (int x : a[v]) vals[p++] = x; off[v + 1] = p; }
}
void oracle_paths(void* h, int32_t* offset, uint64_t* off, int32_t* edges) {
    auto* o = (Oracle*)h;
    std::memcpy(offset, o->path_offset.data(), o->n * 4);
    std::memcpy(off, o->path_off.data(), (o->n + 1) * 8);
    std::memcpy(edges, o->path_edges.data(), o->path_edges.size() * 4);
}

}  // extern "C"
