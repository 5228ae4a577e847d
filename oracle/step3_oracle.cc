// oracle/step3_oracle.cc -- TEST INFRASTRUCTURE (checker), not product code.
//
// Single-threaded CPU restatement of the reference's Step 3 ("Repathing to second (large K) graph"),
// src/modules/w2rap-contigger.cc:359-378:
//     hbv.Involution(inv)                         paths/HyperBasevector.cc:648-660
//     FragDist(hbv, inv, paths, file)              paths/long/large/GapToyTools3.cc:616-646 (the counts; the plot is not ours)
//     RepathInMemory(hbv, edges, inv, paths, 60, K2, hbvr, pathsr, True, True, extend_paths)
//                                                  paths/long/large/Repath.cc:23-251
//       -> LongReadsToPaths -> buildBigKHBVFromReads  paths/long/LongReadsToPaths.cc:263-283, kmers/BigKPather.cc:461-556
//          (BigKMerizer::kmerize :40-55, BigKEdgeBuilder :110-310, buildHBVFromEdges paths/long/HBVFromEdges.cc:76-154,
//           Pather :311-405)
//       -> path translation                         Repath.cc:140-249
//
// The translation through HyperKmerPath / KmerPath databases (Repath.cc:140-214) is restated by what it computes: every
// large-K edge object owns a k-mer id range of its own (buildHKPFromHBV, HBVFromEdges.cc:157-213), so the run of
// database hits of a place's KmerPath is exactly the edge list the Pather walked, `starts` is the offset of the place's
// first K2-mer on its first edge and `stops` the number of K2-mers of the last edge behind the place's last one; the
// "bad"/"incomplete" branches cannot fire for paths the Pather itself produced.  This is pinned, like everything here,
// by byte-for-byte comparison with the reference's own Step-3 output on the fixtures (tests/test_step3_oracle.py).
//
// Edge numbering: BigKEdgeBuilder numbers its edges under a spin lock in hash-set order (:286-292) -- arbitrary, as in
// Step 2.  Replay mode (a hint = the reference's canonical edges in its order) reproduces the reference's files byte for
// byte; canonical mode numbers the edges in lexicographic order of their sequences.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

typedef std::vector<uint8_t> Seq;          // base codes 0..3

static void rcSeq(Seq& s) { std::reverse(s.begin(), s.end()); for (auto& b : s) b ^= 3; }
// bvec::getCanonicalForm, feudal/BaseVec.h:326-327 -> dna/CanonicalForm.h:34-46: 0 FWD, 1 REV, 2 PALINDROME
static int eform(const uint8_t* s, size_t len) {
    if (len & 1) return (s[len / 2] & 2) ? 1 : 0;
    for (size_t i = 0, j = len; i < j; ++i) {
        unsigned f = s[i], r = s[--j] ^ 3u;
        if (f < r) return 0;
        if (r < f) return 1;
    }
    return 2;
}
static int eform(Seq const& s) { return eform(s.data(), s.size()); }
static unsigned popc4(unsigned m) { return __builtin_popcount(m & 15u); }
static unsigned single(unsigned m) { return __builtin_ctz(m | 16u); }
static uint8_t brev8(unsigned c) {        // KMerContext::rc, kmers/KMerContext.cc:18-36
    c = ((c >> 4) | (c << 4)) & 0xFF; c = ((c >> 2) & 0x33) | ((c & 0x33) << 2); c = ((c >> 1) & 0x55) | ((c & 0x55) << 1);
    return (uint8_t)c;
}

struct Oracle3 {
    std::string err;
    unsigned K = 60, K2 = 200;
    // ---- input: the small-K graph and paths
    std::vector<Seq> e1;                   // edge objects of the small-K HBV
    std::vector<int> inv;                  // Involution
    uint64_t n_reads = 0;
    const int32_t* p_offset = nullptr; const uint64_t* p_off = nullptr; const int32_t* p_edges = nullptr;
    // ---- FragDist
    double frag[100];
    // ---- places
    std::vector<std::vector<int>> places;
    size_t n_unique_places = 0;
    std::vector<int> to_left, to_right; size_t n_vertices1 = 0;    // the small-K graph's vertices (needed by --extend_paths only)
    std::vector<Seq> all; std::vector<int> left_trunc, right_trunc;
    // ---- big-K dictionary: canonical K2-mers by content; an entry remembers one occurrence
    struct Ent { uint32_t seq, off; bool rc; uint8_t ctx; int edge; uint32_t eoff; bool erc; };
    std::vector<Ent> ents;
    std::unordered_map<std::string, uint32_t> dict;     // canonical K2-mer (codes) -> index into ents
    uint64_t n_instances = 0;
    std::vector<Seq> edges;                // unipaths, canonical orientation
    std::vector<std::vector<uint32_t>> members;
    // ---- HBV 2
    std::vector<Seq> objs; std::vector<int> fwdX, revX, left, right, inv2;
    uint64_t n_vertices = 0;
    std::vector<std::vector<int>> from_v, from_e, to_v, to_e;
    // ---- places through the graph, translated paths
    std::vector<std::vector<int>> ipaths2; std::vector<int> starts, stops;
    std::vector<int32_t> o_offset; std::vector<uint64_t> o_off; std::vector<int32_t> o_edges;

    // HyperBasevector::Involution, paths/HyperBasevector.cc:648-660: ranks of the sequences against ranks of their RCs;
    // sequences of a unipath graph are distinct, so this pairs every object with the object holding its RC
    void involution(std::vector<Seq> const& objs_, std::vector<int>& out) {
        std::unordered_map<std::string, int> id;
        for (size_t i = 0; i < objs_.size(); ++i) id.emplace(std::string(objs_[i].begin(), objs_[i].end()), (int)i);
        out.assign(objs_.size(), -1);
        for (size_t i = 0; i < objs_.size(); ++i) {
            Seq r = objs_[i]; rcSeq(r);
            auto it = id.find(std::string(r.begin(), r.end()));
            if (it == id.end()) { err = "oracle3: edge object " + std::to_string(i) + " has no reverse complement in the graph (Involution)"; return; }
            out[i] = it->second;
        }
    }
    int plen(uint64_t r) const { return (int)(p_off[r + 1] - p_off[r]); }
    const int32_t* pbeg(uint64_t r) const { return p_edges + p_off[r]; }

    // FragDist, GapToyTools3.cc:616-646
    void fragDist() {
        const int width = 10, max_sep = 1000, min_edge = 10000;
        for (double& c : frag) c = 0;
        for (uint64_t id1 = 0; id1 + 1 < n_reads; id1 += 2) {
            const uint64_t id2 = id1 + 1;
            if (plen(id1) == 0 || plen(id2) == 0) continue;
            const int e1_ = pbeg(id1)[0], e2_ = inv[pbeg(id2)[0]];
            const int epos1 = p_offset[id1];
            if (e1_ != e2_) continue;
            if ((int)e1[e1_].size() < min_edge) continue;
            const int epos2 = (int)e1[e2_].size() - p_offset[id2];
            const int len = epos2 - epos1;
            if (len < 0 || len >= max_sep) continue;
            frag[len / width] += 1;
        }
    }

    // Repath.cc:40-71: places
    bool placeOf(uint64_t r, std::vector<int>& x, bool* rc) const {
        const int n = plen(r);
        x.assign(pbeg(r), pbeg(r) + n);
        int nkmers = 0;
        for (int e : x) nkmers += (int)e1[e].size() - ((int)K - 1);
        if (nkmers + ((int)K - 1) < (int)K2) return false;
        std::vector<int> y;
        for (int j = n - 1; j >= 0; --j) y.push_back(inv[x[j]]);
        *rc = y < x;
        if (y < x) x = y;
        return true;
    }
    void buildPlaces() {
        std::vector<int> x; bool rc;
        for (uint64_t r = 0; r < n_reads; ++r) if (placeOf(r, x, &rc)) places.push_back(x);
        std::sort(places.begin(), places.end());
        places.erase(std::unique(places.begin(), places.end()), places.end());
        n_unique_places = places.size();                       // what the reference prints (:71), before any extension
    }
    // Repath.cc:72-96 (EXTEND_PATHS, "--extend_paths", experimental): every place gets the sole edge entering its first vertex in front
    // and the sole edge leaving its last vertex behind, unless the place already holds that edge.  As written the loops never advance
    // v / w: `while (hb.To(v).solo())` pushes e once, finds it a member the next time round and breaks -- ONE edge per side at most, and
    // the right side tests membership against the place with its new front edge.  The extended places are added as they are (not
    // re-canonicalised against their reverse complement), the whole list sorted and made unique again.
    void extendPlaces() {
        const size_t nv = n_vertices1;
        std::vector<int> indeg(nv, 0), outdeg(nv, 0), in_e(nv, -1), out_e(nv, -1);
        for (size_t e = 0; e < to_left.size(); ++e) { ++indeg[to_right[e]]; in_e[to_right[e]] = (int)e; ++outdeg[to_left[e]]; out_e[to_left[e]] = (int)e; }
        std::vector<std::vector<int>> eplaces;
        for (size_t i = 0; i < places.size(); ++i) {
            std::vector<int> p = places[i];
            const int v = to_left[p.front()], w = to_right[p.back()];
            if (indeg[v] == 1) { const int e = in_e[v]; if (std::find(p.begin(), p.end(), e) == p.end()) p.insert(p.begin(), e); }
            if (outdeg[w] == 1) { const int e = out_e[w]; if (std::find(p.begin(), p.end(), e) == p.end()) p.push_back(e); }
            if (p.size() > places[i].size()) eplaces.push_back(p);
        }
        places.insert(places.end(), eplaces.begin(), eplaces.end());
        std::sort(places.begin(), places.end());
        places.erase(std::unique(places.begin(), places.end()), places.end());
    }
    // Repath.cc:101-123: bases of a place, first and last edge cut to at most K2 bases
    void buildAll() {
        all.resize(places.size()); left_trunc.assign(places.size(), 0); right_trunc.assign(places.size(), 0);
        for (size_t i = 0; i < places.size(); ++i) {
            auto const& e = places[i];
            Seq b = e1[e[0]];
            for (size_t l = 1; l < e.size(); ++l) { b.resize(b.size() - (K - 1)); b.insert(b.end(), e1[e[l]].begin(), e1[e[l]].end()); }
            if (e.size() > 1) {
                int x = e.back();
                if ((int)e1[x].size() > (int)K2) { b.resize(b.size() - (e1[x].size() - K2)); right_trunc[i] = (int)e1[x].size() - (int)K2; }
                x = e.front();
                if ((int)e1[x].size() > (int)K2) { b.erase(b.begin(), b.begin() + (e1[x].size() - K2)); left_trunc[i] = (int)e1[x].size() - (int)K2; }
            }
            all[i] = b;
        }
    }

    // ---- BigKMerizer::kmerize, BigKPather.cc:40-55 (+ canonicalAdd :104-108)
    static std::string canon(const uint8_t* s, unsigned k, bool* rev) {      // CF<K>::getForm: REV iff the RC is smaller
        std::string f((const char*)s, k);
        *rev = eform(s, k) == 1;
        if (*rev) { std::reverse(f.begin(), f.end()); for (auto& c : f) c ^= 3; }
        return f;
    }
    uint32_t lookupCanon(std::string const& key) const { auto it = dict.find(key); return it == dict.end() ? ~0u : it->second; }
    void add(uint32_t seq, uint32_t off, uint8_t ctx) {
        bool rev; std::string key = canon(&all[seq][off], K2, &rev);
        if (rev) ctx = brev8(ctx);
        auto it = dict.find(key);
        ++n_instances;
        if (it == dict.end()) { dict.emplace(key, (uint32_t)ents.size()); ents.push_back(Ent{seq, off, rev, ctx, -1, 0, false}); }
        else ents[it->second].ctx |= ctx;
    }
    void kmerize() {
        for (uint32_t i = 0; i < all.size(); ++i) {
            Seq const& bv = all[i];
            if (bv.size() < K2) continue;
            if (bv.size() == K2) { add(i, 0, 0); continue; }
            add(i, 0, (uint8_t)(1u << bv[K2]));                                       // initialContext: successor only
            uint32_t last = (uint32_t)(bv.size() - K2);
            for (uint32_t p = 1; p < last; ++p) add(i, p, (uint8_t)((1u << (4 + bv[p - 1])) | (1u << bv[p + K2])));
            add(i, last, (uint8_t)(1u << (4 + bv[last - 1])));                         // finalContext: predecessor only
        }
    }
    // ---- BigKEdgeBuilder, BigKPather.cc:110-310 (same construction as Step 2's EdgeBuilder)
    static bool isPal(const uint8_t* s, unsigned k) { return eform(s, k) == 2; }      // even K2 only (CF<BIGK>::isPalindrome :188-190)
    // entry + context in the orientation of the query k-mer (lookup :262-273)
    uint32_t lookupCtx(const uint8_t* s, uint8_t* ctx) {
        bool rev; std::string key = canon(s, K2, &rev);
        uint32_t e = lookupCanon(key);
        if (e == ~0u) { err = "oracle3: neighbour lookup failed (ForceAssert BigKPather.cc:266)"; return e; }
        *ctx = rev ? brev8(ents[e].ctx) : ents[e].ctx;
        return e;
    }
    Seq entSeq(uint32_t e, bool flip) const {          // the entry's K2-mer in canonical orientation (flip: its RC)
        Ent const& x = ents[e];
        Seq s(all[x.seq].begin() + x.off, all[x.seq].begin() + x.off + K2);
        if (x.rc != flip) rcSeq(s);
        return s;
    }
    bool upPossible(Seq const& k, uint8_t ctx) {        // :192-202
        unsigned pm = ctx >> 4;
        if (popc4(pm) != 1) return false;
        Seq p; p.push_back((uint8_t)single(pm)); p.insert(p.end(), k.begin(), k.end() - 1);
        if (isPal(p.data(), K2)) return false;
        uint8_t c; if (lookupCtx(p.data(), &c) == ~0u) return false;
        return popc4(c & 15) == 1;
    }
    bool downPossible(Seq const& k, uint8_t ctx) {      // :204-214
        unsigned sm = ctx & 15;
        if (popc4(sm) != 1) return false;
        Seq s(k.begin() + 1, k.end()); s.push_back((uint8_t)single(sm));
        if (isPal(s.data(), K2)) return false;
        uint8_t c; if (lookupCtx(s.data(), &c) == ~0u) return false;
        return popc4(c >> 4) == 1;
    }
    void addEdge(Seq& seq, std::vector<uint32_t>& mem) {      // :275-306
        if (eform(seq) == 1) { rcSeq(seq); std::reverse(mem.begin(), mem.end()); }
        int id = (int)edges.size();
        for (uint32_t m : mem) {
            if (ents[m].edge != -1) err = "oracle3: preoccupied kmers (BigKPather.cc:303)";
            ents[m].edge = id;
        }
        edges.push_back(seq); members.push_back(mem);
    }
    void extend(uint8_t ctx, Seq& seq, std::vector<uint32_t>& mem) {     // :234-259
        while (popc4(ctx & 15) == 1) {
            seq.push_back((uint8_t)single(ctx & 15));
            const uint8_t* nk = seq.data() + seq.size() - K2;
            if (isPal(nk, K2)) { seq.pop_back(); break; }
            uint8_t c; uint32_t e = lookupCtx(nk, &c);
            if (e == ~0u) return;
            if (popc4(c >> 4) != 1) { seq.pop_back(); break; }
            mem.push_back(e); ctx = c;
        }
        if (eform(seq) != 1) addEdge(seq, mem);            // the REV copy is produced from the other end
    }
    void buildEdges() {
        Seq seq; std::vector<uint32_t> mem;
        for (uint32_t i = 0; i < ents.size(); ++i) {       // buildEdge :104-115
            if (ents[i].edge != -1) continue;
            seq.clear(); mem.clear();
            Seq k = entSeq(i, false);
            const uint8_t ctx = ents[i].ctx;
            if (isPal(k.data(), K2)) { seq = k; mem.push_back(i); addEdge(seq, mem); }
            else if (upPossible(k, ctx)) {
                if (downPossible(k, ctx)) continue;
                seq = entSeq(i, true); mem.push_back(i);
                extend(brev8(ctx), seq, mem);
            } else if (downPossible(k, ctx)) { seq = k; mem.push_back(i); extend(ctx, seq, mem); }
            else { seq = k; mem.push_back(i); addEdge(seq, mem); }
            if (!err.empty()) return;
        }
        for (uint32_t i = 0; i < ents.size(); ++i) {       // smooth circles: simpleCircle :126-153, canonicalizeCircle :156-180
            if (ents[i].edge != -1) continue;
            seq = entSeq(i, false); mem.assign(1, i);
            uint8_t ctx = ents[i].ctx;
            while (true) {
                if (popc4(ctx >> 4) != 1 || popc4(ctx & 15) != 1) { err = "oracle3: circle context (BigKPather.cc:136)"; return; }
                seq.push_back((uint8_t)single(ctx & 15));
                uint32_t e = lookupCtx(seq.data() + seq.size() - K2, &ctx);
                if (e == ~0u) return;
                if (e == i) { seq.pop_back(); break; }
                if (ents[e].edge != -1) { err = "oracle3: failed to close circle (BigKPather.cc:141)"; return; }
                mem.push_back(e);
            }
            size_t idx = 0;                                  // the minimum k-mer (entries compare in canonical orientation)
            { Seq best = entSeq(mem[0], false);
              for (size_t j = 1; j < mem.size(); ++j) { Seq c = entSeq(mem[j], false); if (c < best) { best = c; idx = j; } }
              if (!std::equal(best.begin(), best.end(), seq.begin() + idx)) {
                  rcSeq(seq); std::reverse(mem.begin(), mem.end());
                  idx = seq.size() - idx - K2;
              } }
            if (idx) {
                Seq bv(seq.begin() + idx, seq.end());
                bv.insert(bv.end(), seq.begin() + (K2 - 1), seq.begin() + (K2 - 1 + idx));
                seq = bv;
                std::rotate(mem.begin(), mem.begin() + idx, mem.end());
            }
            addEdge(seq, mem);
            if (!err.empty()) return;
        }
    }
    bool orderEdges(uint64_t n_hint, const uint8_t* hint, const uint64_t* hoff) {
        size_t E = edges.size();
        std::vector<size_t> perm(E);
        std::vector<size_t> byseq(E);
        for (size_t i = 0; i < E; ++i) byseq[i] = i;
        std::sort(byseq.begin(), byseq.end(), [&](size_t a, size_t b) { return edges[a] < edges[b]; });
        if (hint) {
            if (n_hint != E) { err = "oracle3: edge hint count " + std::to_string(n_hint) + " != " + std::to_string(E); return false; }
            for (size_t i = 0; i < E; ++i) {
                Seq s(hint + hoff[i], hint + hoff[i + 1]);
                auto it = std::lower_bound(byseq.begin(), byseq.end(), s, [&](size_t a, Seq const& v) { return edges[a] < v; });
                if (it == byseq.end() || edges[*it] != s) { err = "oracle3: hinted edge " + std::to_string(i) + " not in our edge set"; return false; }
                perm[i] = *it;
            }
        } else perm = byseq;
        std::vector<Seq> ne(E); std::vector<std::vector<uint32_t>> nm(E); std::vector<int> seen(E, 0);
        for (size_t i = 0; i < E; ++i) { if (seen[perm[i]]++) { err = "oracle3: duplicate hinted edge"; return false; } ne[i] = edges[perm[i]]; nm[i] = members[perm[i]]; }
        edges.swap(ne); members.swap(nm);
        // updateDict :78-88: every entry learns (edge, offset, orientation on the edge)
        for (size_t e = 0; e < E; ++e)
            for (uint32_t o = 0; o + K2 <= edges[e].size(); ++o) {
                bool rev; std::string key = canon(&edges[e][o], K2, &rev);
                uint32_t x = lookupCanon(key);
                if (x == ~0u) { err = "oracle3: edge k-mer not in the dictionary"; return false; }
                ents[x].edge = (int)e; ents[x].eoff = o; ents[x].erc = rev;
            }
        return true;
    }
    // buildHBVFromEdges, paths/long/HBVFromEdges.cc:76-154, with K2
    struct End { uint64_t hash; Seq seq; uint32_t obj; bool distal; };
    void buildHBV() {
        objs.clear(); fwdX.assign(edges.size(), -1); revX.assign(edges.size(), -1);
        for (size_t i = 0; i < edges.size(); ++i) {
            fwdX[i] = (int)objs.size(); objs.push_back(edges[i]);
            if (eform(edges[i]) == 2) revX[i] = fwdX[i];
            else { revX[i] = (int)objs.size(); Seq r = edges[i]; rcSeq(r); objs.push_back(r); }
        }
        std::vector<End> ends; ends.reserve(2 * objs.size());
        for (size_t o = 0; o < objs.size(); ++o)
            for (int d = 0; d < 2; ++d) {
                End e; e.obj = (uint32_t)o; e.distal = d;
                auto const& s = objs[o];
                e.seq.assign(d ? s.end() - (K2 - 1) : s.begin(), d ? s.end() : s.begin() + (K2 - 1));
                uint64_t h = 14695981039346656037ull;           // math/Hash.h:26-35 FNV1a over base codes
                for (uint8_t b : e.seq) h = 1099511628211ull * (h ^ b);
                e.hash = h; ends.push_back(e);
            }
        std::stable_sort(ends.begin(), ends.end(), [](End const& a, End const& b) { if (a.hash != b.hash) return a.hash < b.hash; return a.seq < b.seq; });
        left.assign(objs.size(), -1); right.assign(objs.size(), -1);
        int64_t vid = 0;
        for (size_t i = 0; i < ends.size(); ++i) {
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
        inv2.assign(objs.size(), -1);
        for (size_t i = 0; i < edges.size(); ++i) { inv2[fwdX[i]] = revX[i]; inv2[revX[i]] = fwdX[i]; }
    }
    // Pather :321-357 for every place; the translation Repath.cc:150-214 reads the same edges back (see the header)
    void pathPlaces() {
        ipaths2.assign(all.size(), {}); starts.assign(all.size(), 0); stops.assign(all.size(), 0);
        for (size_t i = 0; i < all.size(); ++i) {
            Seq const& read = all[i];
            if (read.size() < K2) continue;
            size_t pos = 0, remaining = read.size() - K2 + 1;           // K2-mers still to place
            bool first = true; int last_edge = -1; size_t last_end = 0;
            while (remaining) {
                bool rev; std::string key = canon(&read[pos], K2, &rev);
                uint32_t x = lookupCanon(key);
                if (x == ~0u) { err = "oracle3: place k-mer not in the dictionary (ForceAssert BigKPather.cc:363)"; return; }
                Ent const& en = ents[x];
                const bool rc = rev != en.erc;                          // the read runs against the edge's stored orientation
                const size_t nk = edges[en.edge].size() - K2 + 1;
                const size_t off = rc ? nk - 1 - en.eoff : en.eoff;     // offset on the edge OBJECT the read follows
                if (!first && off != 0) { err = "oracle3: next entry not at offset 0 (ForceAssertEq BigKPather.cc:338)"; return; }
                const int obj = rc ? revX[en.edge] : fwdX[en.edge];
                if (first) { starts[i] = (int)off; first = false; }
                ipaths2[i].push_back(obj);
                const size_t take = std::min(remaining, nk - off);
                last_edge = obj; last_end = off + take;                 // K2-mers of the last edge used so far
                pos += take; remaining -= take;
            }
            stops[i] = (int)(objs[last_edge].size() - K2 + 1 - last_end);
        }
    }
    // Repath.cc:216-249
    void translate() {
        o_offset.assign(n_reads, 0); o_off.assign(n_reads + 1, 0); o_edges.clear();
        std::vector<int> x; bool rc;
        for (uint64_t id = 0; id < n_reads; ++id) {
            o_off[id] = o_edges.size();
            if (plen(id) == 0) continue;
            if (!placeOf(id, x, &rc)) continue;
            const size_t pos = std::lower_bound(places.begin(), places.end(), x) - places.begin();      // BinPosition
            auto const& ip = ipaths2[pos];
            const int n = (int)ip.size();
            o_offset[id] = !rc ? p_offset[id] + starts[pos] - left_trunc[pos] : p_offset[id] + stops[pos] - right_trunc[pos];
            if (!rc) for (int j = 0; j < n; ++j) o_edges.push_back(ip[j]);
            else for (int j = 0; j < n; ++j) o_edges.push_back(inv2[ip[n - j - 1]]);
        }
        o_off[n_reads] = o_edges.size();
    }
};

static void flat(std::vector<Seq> const& v, uint8_t* b, uint64_t* off) {
    uint64_t p = 0; off[0] = 0;
    for (size_t i = 0; i < v.size(); ++i) { std::memcpy(b + p, v[i].data(), v[i].size()); p += v[i].size(); off[i + 1] = p; }
}

}  // namespace

extern "C" {

// the small-K graph = its edge objects (codes + offsets); paths in CSR form; hint: the large-K canonical edges to replay
void* oracle3_run(unsigned K, unsigned K2, uint64_t n_obj, const uint8_t* obj_codes, const uint64_t* obj_off,
                  uint64_t n_reads, const int32_t* p_offset, const uint64_t* p_off, const int32_t* p_edges,
                  uint64_t n_hint, const uint8_t* hint_bases, const uint64_t* hint_off, int stop_after /*0 all, 1 places + all*/,
                  int extend_paths, uint64_t n_vertices, const int32_t* to_left, const int32_t* to_right /* [n_obj], extend_paths only */) {
    Oracle3* o = new Oracle3;
    o->K = K; o->K2 = K2;
    if (K2 & 1 || K2 <= K) { o->err = "oracle3: K2 must be even and larger than K"; return o; }
    o->e1.resize(n_obj);
    for (uint64_t i = 0; i < n_obj; ++i) o->e1[i].assign(obj_codes + obj_off[i], obj_codes + obj_off[i + 1]);
    o->n_reads = n_reads; o->p_offset = p_offset; o->p_off = p_off; o->p_edges = p_edges;
    o->involution(o->e1, o->inv);
    if (!o->err.empty()) return o;
    o->fragDist();
    o->buildPlaces();
    if (extend_paths) {
        if (!to_left || !to_right) { o->err = "oracle3: extend_paths needs the vertices of the small-K graph"; return o; }
        o->n_vertices1 = n_vertices; o->to_left.assign(to_left, to_left + n_obj); o->to_right.assign(to_right, to_right + n_obj);
        o->extendPlaces();
    }
    o->buildAll();
    if (stop_after == 1) return o;
    o->kmerize();
    o->buildEdges();
    if (!o->err.empty()) return o;
    if (!o->orderEdges(n_hint, hint_bases, hint_off)) return o;
    o->buildHBV();
    o->pathPlaces();
    if (!o->err.empty()) return o;
    o->translate();
    return o;
}
const char* oracle3_error(void* h) { auto* o = (Oracle3*)h; return o->err.empty() ? nullptr : o->err.c_str(); }
void oracle3_free(void* h) { delete (Oracle3*)h; }
// sizes: [0] places [1] place ints total [2] all bases [3] K2-mer instances [4] distinct [5] edges [6] objs [7] vertices
//        [8] obj bases [9] path ints
void oracle3_sizes(void* h, uint64_t* out) {
    auto* o = (Oracle3*)h;
    out[0] = o->places.size(); uint64_t t = 0; for (auto& p : o->places) t += p.size(); out[1] = t;
    t = 0; for (auto& s : o->all) t += s.size(); out[2] = t;
    out[3] = o->n_instances; out[4] = o->ents.size(); out[5] = o->edges.size(); out[6] = o->objs.size(); out[7] = o->n_vertices;
    t = 0; for (auto& s : o->objs) t += s.size(); out[8] = t;
    out[9] = o->o_edges.size();
}
void oracle3_inv(void* h, int32_t* inv) { auto* o = (Oracle3*)h; std::memcpy(inv, o->inv.data(), o->inv.size() * 4); }
void oracle3_frag(void* h, double* f) { std::memcpy(f, ((Oracle3*)h)->frag, 100 * 8); }
void oracle3_places(void* h, uint64_t* off, int32_t* vals, int32_t* ltrunc, int32_t* rtrunc) {
    auto* o = (Oracle3*)h; uint64_t p = 0; off[0] = 0;
    for (size_t i = 0; i < o->places.size(); ++i) { for (int x : o->places[i]) vals[p++] = x; off[i + 1] = p; ltrunc[i] = o->left_trunc[i]; rtrunc[i] = o->right_trunc[i]; }
}
void oracle3_all(void* h, uint8_t* b, uint64_t* off) { flat(((Oracle3*)h)->all, b, off); }
void oracle3_objs(void* h, uint8_t* b, uint64_t* off, int32_t* left, int32_t* right, int32_t* inv2) {
    auto* o = (Oracle3*)h; flat(o->objs, b, off);
    std::memcpy(left, o->left.data(), o->left.size() * 4); std::memcpy(right, o->right.data(), o->right.size() * 4);
    std::memcpy(inv2, o->inv2.data(), o->inv2.size() * 4);
}
void oracle3_adj(void* h, int which, uint64_t* off, int32_t* vals) {
    auto* o = (Oracle3*)h;
    auto const& a = which == 0 ? o->from_v : which == 1 ? o->from_e : which == 2 ? o->to_v : o->to_e;
    uint64_t p = 0; off[0] = 0;
    for (size_t v = 0; v < a.size(); ++v) { for (int x : a[v]) vals[p++] = x; off[v + 1] = p; }
}
void oracle3_paths(void* h, int32_t* offset, uint64_t* off, int32_t* edges) {
    auto* o = (Oracle3*)h;
    std::memcpy(offset, o->o_offset.data(), o->o_offset.size() * 4); std::memcpy(off, o->o_off.data(), o->o_off.size() * 8);
    if (!o->o_edges.empty()) std::memcpy(edges, o->o_edges.data(), o->o_edges.size() * 4);
}

}  // extern "C"
