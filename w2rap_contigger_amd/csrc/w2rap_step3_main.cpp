// w2rap-step3 -- standalone Step 3 with the reference's file names and flags.
//
// Drop-in for `w2rap-contigger --from_step 3 --to_step 3` (src/modules/w2rap-contigger.cc:352-378): reads
// <out_dir>/<prefix>.small_K.{hbv,paths} written by Step 2 (the reference's or w2rap-step2), writes
// <out_dir>/<prefix>.large_K.{hbv,paths} that Step 4 (`--from_step 4`) loads, and <out_dir>/<prefix>.first.frags.dist
// (FragDist, paths/long/large/GapToyTools3.cc:636-646; the PNG rendering is not ours).  All compute happens in
// libw2rap_step2.so (HIP).
//
//   w2rap-step3 -o <out_dir> -p <prefix> [-K 200] [--device 0] [--edge_order_from <file.hbv>]
//
// File layouts: include/w2rap_step2.h, include/w2rap_step3.h and w2rap_contigger_amd/formats.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "w2rap_step3.h"

namespace {

bool slurp(const std::string& path, std::vector<uint8_t>& buf) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    std::streamsize n = f.tellg();
    f.seekg(0);
    buf.resize((size_t)n);
    return n == 0 || (bool)f.read((char*)buf.data(), n);
}

// the edges_ section of a BINWRITE .hbv (HyperBasevector::writeBinary, paths/HyperBasevector.cc:121-125); every count is checked
// against the bytes that are there
struct HbvEdges {
    int32_t K = 0;
    std::vector<uint8_t> packed; std::vector<uint64_t> byte_off{0}; std::vector<uint32_t> len;
    uint64_t n_vertices = 0; std::vector<int32_t> from_v, from_e; std::vector<uint64_t> from_deg;     // from_ / from_edge_obj_ (for --extend_paths)
    // hbv.ToLeft / ToRight from the adjacency sections; false if they do not describe every edge object once
    bool to_left_right(std::vector<int32_t>& tl, std::vector<int32_t>& tr) const {
        tl.assign(len.size(), -1); tr.assign(len.size(), -1);
        if (from_v.size() != from_e.size() || from_deg.size() != n_vertices) return false;
        size_t j = 0;
        for (uint64_t v = 0; v < n_vertices; ++v)
            for (uint64_t t = 0; t < from_deg[v]; ++t, ++j) {
                if (j >= from_e.size() || from_e[j] < 0 || (size_t)from_e[j] >= len.size() || tl[from_e[j]] >= 0) return false;
                tl[from_e[j]] = (int32_t)v; tr[from_e[j]] = from_v[j];
            }
        for (int32_t x : tl) if (x < 0) return false;
        return true;
    }
    bool load(const std::string& path, std::string& err) {
        std::vector<uint8_t> hb;
        if (!slurp(path, hb) || hb.size() < 12 || std::memcmp(hb.data(), "BINWRITE", 8)) { err = "cannot read " + path + " (not a BINWRITE .hbv)"; return false; }
        std::memcpy(&K, &hb[8], 4);
        size_t p = 12;
        auto need = [&](uint64_t bytes) { return bytes <= hb.size() - p; };
        for (int t = 0; t < 3; ++t) {                              // from_, from_edge_obj_, to_edge_obj_
            if (!need(8)) goto bad;
            { uint64_t nv; std::memcpy(&nv, &hb[p], 8); p += 8;
              if (t == 0) n_vertices = nv;
              for (uint64_t v = 0; v < nv; ++v) {
                  if (!need(8)) goto bad;
                  uint64_t d; std::memcpy(&d, &hb[p], 8); p += 8;
                  if (d > (hb.size() - p) / 4) goto bad;
                  if (t == 0) from_deg.push_back(d);
                  if (t < 2) { std::vector<int32_t>& dst = t == 0 ? from_v : from_e; const size_t at = dst.size(); dst.resize(at + d); if (d) std::memcpy(&dst[at], &hb[p], 4 * d); }
                  p += 4 * d;
              } }
        }
        if (!need(8)) goto bad;
        { uint64_t E; std::memcpy(&E, &hb[p], 8); p += 8;
          for (uint64_t e = 0; e < E; ++e) {
              if (!need(4)) goto bad;
              uint32_t nb; std::memcpy(&nb, &hb[p], 4); p += 4;
              const size_t nby = ((size_t)nb + 3) / 4;
              if (!need(nby)) goto bad;
              packed.insert(packed.end(), hb.begin() + p, hb.begin() + p + nby); p += nby;
              byte_off.push_back(packed.size()); len.push_back(nb);
          } }
        return true;
    bad:
        err = "cannot read " + path + ": truncated or not a .hbv file";
        return false;
    }
};

void put(std::vector<uint8_t>& b, const void* p, size_t n) { b.insert(b.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
template <class T> void put(std::vector<uint8_t>& b, T v) { put(b, &v, sizeof(T)); }
void put_csr(std::vector<uint8_t>& b, uint64_t nv, const uint64_t* off, const int32_t* vals) {
    put<uint64_t>(b, nv);
    for (uint64_t v = 0; v < nv; ++v) { put<uint64_t>(b, off[v + 1] - off[v]); put(b, vals + off[v], (off[v + 1] - off[v]) * 4); }
}

}  // namespace

int main(int argc, char** argv) {
    std::string out_dir, prefix, hint_path;
    w2rap_step3_params P{};
    P.K2 = 200; P.device = 0; P.extend_paths = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "-o" || a == "--out_dir") out_dir = next();
        else if (a == "-p" || a == "--prefix") prefix = next();
        else if (a == "-K" || a == "--large_k") P.K2 = (uint32_t)std::atoi(next());
        else if (a == "--extend_paths") { const std::string v = next(); P.extend_paths = (v == "1" || v == "true" || v == "True") ? 1 : 0; }      // (TCLAP::ValueArg<bool>, w2rap-contigger.cc:110)
        else if (a == "--device") P.device = std::atoi(next());
        else if (a == "--unique_kmers") { if (std::atoi(next())) P.flags |= W2RAP_STEP3_UNIQUE_KMERS; }      // the .hbv is Step 2's unipath graph (include/w2rap_step3.h)
        else if (a == "--edge_order_from") hint_path = next();
        else if (a == "-t" || a == "-m" || a == "-d" || a == "--disk_batches" || a == "--tmp_dir" || a == "-r") next();   // accepted, unused
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (out_dir.empty() || prefix.empty()) { std::fprintf(stderr, "usage: w2rap-step3 -o out_dir -p prefix [-K large_k] [--extend_paths 0|1] [--device d] [--unique_kmers 0|1] [--edge_order_from x.hbv]\n"); return 2; }
    std::string err;
    HbvEdges hb;
    if (!hb.load(out_dir + "/" + prefix + ".small_K.hbv", err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    // <prefix>.small_K.paths: {u64 n; n x {i32 offset; u16 len; i32[len]}} (WriteReadPathVec, paths/long/ReadPath.cc:6-20)
    std::vector<uint8_t> pb;
    const std::string ppath = out_dir + "/" + prefix + ".small_K.paths";
    if (!slurp(ppath, pb) || pb.size() < 8) { std::fprintf(stderr, "cannot read %s\n", ppath.c_str()); return 1; }
    uint64_t n; std::memcpy(&n, pb.data(), 8);
    std::vector<int32_t> p_offset; std::vector<uint64_t> p_off{0}; std::vector<int32_t> p_edges;
    {
        size_t p = 8;
        for (uint64_t r = 0; r < n; ++r) {
            if (pb.size() - p < 6) { std::fprintf(stderr, "cannot read %s: truncated\n", ppath.c_str()); return 1; }
            int32_t o; uint16_t l; std::memcpy(&o, &pb[p], 4); std::memcpy(&l, &pb[p + 4], 2); p += 6;
            if ((pb.size() - p) / 4 < l) { std::fprintf(stderr, "cannot read %s: truncated\n", ppath.c_str()); return 1; }
            p_offset.push_back(o);
            for (unsigned j = 0; j < l; ++j) { int32_t e; std::memcpy(&e, &pb[p + 4 * j], 4); p_edges.push_back(e); }
            p += 4 * (size_t)l; p_off.push_back(p_edges.size());
        }
    }
    std::printf("--== Step 3: Repathing to second (large K) graph ==--\n");
    w2rap_step3_in I{};
    I.K = hb.K; I.n_edge_objs = hb.len.size(); I.edge_packed = hb.packed.data(); I.edge_byte_off = hb.byte_off.data(); I.edge_len = hb.len.data();
    I.n_paths = n; I.path_offset = p_offset.data(); I.path_off = p_off.data(); I.path_edges = p_edges.data();
    std::vector<int32_t> tl, tr;
    if (P.extend_paths) {
        if (!hb.to_left_right(tl, tr)) { std::fprintf(stderr, "%s.small_K.hbv: the adjacency sections do not name every edge object once\n", prefix.c_str()); return 1; }
        I.n_vertices = hb.n_vertices; I.vleft = tl.data(); I.vright = tr.data();
    }
    // optional: replay the unipath order of an existing .large_K.hbv (its non-REV-canonical edge objects, in id order)
    w2rap_edge_hint H{}; HbvEdges hh; std::vector<uint8_t> hpacked; std::vector<uint64_t> hoff{0}; std::vector<uint32_t> hlen;
    if (!hint_path.empty()) {
        if (!hh.load(hint_path, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
        for (size_t e = 0; e < hh.len.size(); ++e) {
            const uint8_t* s = hh.packed.data() + hh.byte_off[e]; const uint32_t nb = hh.len[e];
            auto base = [&](uint32_t i) { return (s[i >> 2] >> (2 * (i & 3))) & 3; };
            int form = 2;                                   // bvec::getCanonicalForm (dna/CanonicalForm.h:34-46)
            if (nb & 1) form = (base(nb / 2) & 2) ? 1 : 0;
            else for (uint32_t i = 0, j = nb; i < j; ++i) { unsigned f = base(i), r = base(--j) ^ 3u; if (f < r) { form = 0; break; } if (r < f) { form = 1; break; } }
            if (form == 1) continue;
            hpacked.insert(hpacked.end(), s, s + (nb + 3) / 4); hoff.push_back(hpacked.size()); hlen.push_back(nb);
        }
        H.n_edges = hlen.size(); H.packed = hpacked.data(); H.byte_off = hoff.data(); H.len = hlen.data();
        P.edge_order_hint = &H;
    }
    w2rap_step3_out O{};
    char ebuf[1024] = {0};
    int rc = w2rap_step3_run(&I, &P, &O, ebuf, sizeof ebuf);
    if (rc) { std::fprintf(stderr, "w2rap_step3_run failed (%d): %s\n", rc, ebuf); return 1; }
    std::printf("beginning repathing %llu edges from K=%d to K2=%u\nconstructing places from %llu paths\n%llu / %llu reads pathed, %llu spanning junctions\n"
                "sorting %llu places\n%llu unique places\n", (unsigned long long)I.n_edge_objs, I.K, P.K2, (unsigned long long)n, (unsigned long long)O.n_reads_pathed,
                (unsigned long long)n, (unsigned long long)O.n_reads_multipathed, (unsigned long long)O.n_places, (unsigned long long)O.n_unique_places);
    if (P.extend_paths) std::printf("begin extending paths\nresorting\ndone extending paths\n");          // Repath.cc:75,91,95
    std::printf("GPU ms: places %.2f dictionary %.2f graph %.2f paths %.2f\n", O.ms_places, O.ms_dict, O.ms_graph, O.ms_paths);
    std::vector<uint8_t> b;
    put(b, "BINWRITE", 8); put<int32_t>(b, O.K2);
    put_csr(b, O.n_vertices, O.from_off, O.from_v);
    put_csr(b, O.n_vertices, O.from_off, O.from_e);
    put_csr(b, O.n_vertices, O.to_off, O.to_e);
    put<uint64_t>(b, O.n_edge_objs);
    for (uint64_t e = 0; e < O.n_edge_objs; ++e) { put<uint32_t>(b, O.edge_len[e]); put(b, O.edge_packed + O.edge_byte_off[e], O.edge_byte_off[e + 1] - O.edge_byte_off[e]); }
    { std::ofstream f(out_dir + "/" + prefix + ".large_K.hbv", std::ios::binary); f.write((const char*)b.data(), (std::streamsize)b.size()); if (!f) { std::fprintf(stderr, "cannot write .hbv\n"); return 1; } }
    b.clear();
    put<uint64_t>(b, O.n_paths);
    for (uint64_t r = 0; r < O.n_paths; ++r) {
        const uint64_t m = O.path_off[r + 1] - O.path_off[r];
        put<int32_t>(b, O.path_offset[r]); put<uint16_t>(b, (uint16_t)m); put(b, O.path_edges + O.path_off[r], m * 4);
    }
    { std::ofstream f(out_dir + "/" + prefix + ".large_K.paths", std::ios::binary); f.write((const char*)b.data(), (std::streamsize)b.size()); if (!f) { std::fprintf(stderr, "cannot write .paths\n"); return 1; } }
    {   // <prefix>.first.frags.dist: iostream's default double formatting == %g
        FILE* f = std::fopen((out_dir + "/" + prefix + ".first.frags.dist").c_str(), "w");
        if (!f) { std::fprintf(stderr, "cannot write .first.frags.dist\n"); return 1; }
        double total = 0; for (int j = 0; j < 100; ++j) total += (double)O.frag_count[j];
        std::fprintf(f, "# fragment library size distribution\n# bins have diameter 10\n# line format:\n# bin_center mass\n");
        for (int j = 0; j < 100; ++j) { if (total == 0) std::fprintf(f, "%d -nan\n", j * 10 + 5); else std::fprintf(f, "%d %g\n", j * 10 + 5, (double)O.frag_count[j] / total); }
        std::fclose(f);
    }
    std::printf("Repathing to second graph DONE!\n");
    w2rap_step3_free(&O);
    return 0;
}
