// w2rap-step2 -- standalone Step 2 with the reference's file names and flags.
//
// Drop-in for `w2rap-contigger --from_step 2 --to_step 2` (src/modules/w2rap-contigger.cc:
// 326-346): reads <out_dir>/frag_reads_orig.{fastb,qualp} written by the reference's
// Step 1, writes <out_dir>/<prefix>.small_K.{hbv,paths} and <out_dir>/small_K.freqs that
// its Step 3 (`--from_step 3`) loads.  All compute happens in libw2rap_step2.so (HIP).
//
//   w2rap-step2 -o <out_dir> -p <prefix> [--min_freq 4] [--min_qual 7] [--device 0]
//               [--edge_order_from <file.hbv>]
//
// File layouts: see include/w2rap_step2.h and w2rap_contigger_amd/formats.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "w2rap_step2.h"

namespace {

bool slurp(const std::string& path, std::vector<uint8_t>& buf) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    std::streamsize n = f.tellg();
    f.seekg(0);
    buf.resize((size_t)n);
    return n == 0 || (bool)f.read((char*)buf.data(), n);
}

struct Feudal {               // feudal/FeudalControlBlock.h:157-166
    std::vector<uint8_t> raw;
    uint64_t n = 0;
    const uint8_t* var = nullptr;
    std::vector<uint64_t> off;         // relative to var
    const uint8_t* fixed = nullptr;
    uint64_t fixed_bytes = 0;
    bool load(const std::string& path, std::string& err) {
        if (!slurp(path, raw) || raw.size() < 24) { err = "cannot read " + path; return false; }
        uint32_t n32; uint8_t flags; uint64_t var_off, fixed_off;
        std::memcpy(&n32, &raw[0], 4); flags = raw[4];
        std::memcpy(&var_off, &raw[8], 8); std::memcpy(&fixed_off, &raw[16], 8);
        if ((flags & 3) != 1 || fixed_off > raw.size() || var_off > fixed_off || (fixed_off - var_off) % 8) { err = path + ": not a single-file feudal file"; return false; }
        if (var_off < 24 || fixed_off - var_off < 8) { err = path + ": not a single-file feudal file"; return false; }
        n = (fixed_off - var_off) / 8 - 1;
        if ((n & 0xFFFFFFFFull) != n32) { err = path + ": element count mismatch"; return false; }
        off.resize(n + 1);
        std::memcpy(off.data(), &raw[var_off], (n + 1) * 8);        // var_off + (n+1)*8 == fixed_off <= raw.size(), checked above
        for (size_t i = 0; i <= n; ++i) {                            // absolute file offsets into [24, var_off], ascending
            if (off[i] < 24 || off[i] > var_off || (i && off[i] < off[i - 1])) { err = path + ": not a single-file feudal file (offset table)"; return false; }
            off[i] -= 24;
        }
        var = raw.data() + 24; fixed = raw.data() + fixed_off; fixed_bytes = raw.size() - fixed_off;
        return true;
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
    w2rap_step2_params P{};
    P.K = 60; P.min_qual = 7; P.min_freq = 4; P.device = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "-o" || a == "--out_dir") out_dir = next();
        else if (a == "-p" || a == "--prefix") prefix = next();
        else if (a == "--min_freq") P.min_freq = (uint32_t)std::atoi(next());
        else if (a == "--min_qual") P.min_qual = (uint32_t)std::atoi(next());
        else if (a == "--device") P.device = std::atoi(next());
        else if (a == "--gpus") P.n_gpus = std::atoi(next());                       // devices device .. device+gpus-1
        else if (a == "--passes") P.n_passes = (uint32_t)std::atoi(next());         // hash-range passes of the counting phase (0 = automatic)
        else if (a == "--edge_order_from") hint_path = next();
        else if (a == "-t" || a == "-m" || a == "-d" || a == "--disk_batches" || a == "--tmp_dir" || a == "-K" || a == "-r") next();   // accepted, unused
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (out_dir.empty() || prefix.empty()) { std::fprintf(stderr, "usage: w2rap-step2 -o out_dir -p prefix [--min_freq f] [--min_qual q] [--device d] [--gpus n] [--passes p] [--edge_order_from x.hbv]\n"); return 2; }
    std::string err;
    Feudal fb, qp;
    if (!fb.load(out_dir + "/frag_reads_orig.fastb", err) || !qp.load(out_dir + "/frag_reads_orig.qualp", err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    if (fb.n != qp.n || fb.fixed_bytes != fb.n * 4) { std::fprintf(stderr, "fastb/qualp mismatch\n"); return 1; }
    std::printf("--== Step 2: Building first (small K) graph ==--\n");
    w2rap_reads R{};
    R.n_reads = fb.n; R.bases_packed = fb.var; R.base_byte_off = fb.off.data(); R.read_len = (const uint32_t*)fb.fixed;
    R.pq = qp.var; R.pq_off = qp.off.data(); R.mem = W2RAP_MEM_HOST;
    // optional: replay the unipath order of an existing .hbv (its non-REV-canonical edge objects, in id order)
    w2rap_edge_hint H{}; std::vector<uint8_t> hpacked; std::vector<uint64_t> hoff{0}; std::vector<uint32_t> hlen;
    if (!hint_path.empty()) {
        std::vector<uint8_t> hb;
        if (!slurp(hint_path, hb) || hb.size() < 12 || std::memcmp(hb.data(), "BINWRITE", 8)) { std::fprintf(stderr, "cannot read %s\n", hint_path.c_str()); return 1; }
        size_t p = 12;
        // every count in the file is checked against the bytes that are there (a truncated or foreign file must not be read past its end)
        auto need = [&](uint64_t bytes) { if (bytes > hb.size() - p) { std::fprintf(stderr, "cannot read %s: truncated or not a .hbv file\n", hint_path.c_str()); std::exit(1); } };
        for (int t = 0; t < 3; ++t) {
            need(8); uint64_t nv; std::memcpy(&nv, &hb[p], 8); p += 8;
            for (uint64_t v = 0; v < nv; ++v) { need(8); uint64_t d; std::memcpy(&d, &hb[p], 8); p += 8; if (d > (hb.size() - p) / 4) need(~0ull); p += 4 * d; }
        }
        need(8); uint64_t E; std::memcpy(&E, &hb[p], 8); p += 8;
        for (uint64_t e = 0; e < E; ++e) {
            need(4); uint32_t nb; std::memcpy(&nb, &hb[p], 4); p += 4;
            size_t nby = ((size_t)nb + 3) / 4; need(nby);
            const uint8_t* s = hb.data() + p; p += nby;
            auto base = [&](uint32_t i) { return (s[i >> 2] >> (2 * (i & 3))) & 3; };
            int form = 2;                                   // bvec::getCanonicalForm (dna/CanonicalForm.h:34-46)
            if (nb & 1) form = (base(nb / 2) & 2) ? 1 : 0;
            else for (uint32_t i = 0, j = nb; i < j; ++i) { unsigned f = base(i), r = base(--j) ^ 3u; if (f < r) { form = 0; break; } if (r < f) { form = 1; break; } }
            if (form == 1) continue;
            hpacked.insert(hpacked.end(), s, s + nby); hoff.push_back(hpacked.size()); hlen.push_back(nb);
        }
        H.n_edges = hlen.size(); H.packed = hpacked.data(); H.byte_off = hoff.data(); H.len = hlen.data();
        P.edge_order_hint = &H;
    }
    std::string freqs = out_dir + "/small_K.freqs";
    P.freqs_path = freqs.c_str();
    w2rap_step2_out O{};
    char ebuf[1024] = {0};
    int rc = w2rap_step2_run(&R, &P, &O, ebuf, sizeof ebuf);
    if (rc) { std::fprintf(stderr, "w2rap_step2_run failed (%d): %s\n", rc, ebuf); return 1; }
    std::printf("%llu kmers counted, filtering...\n%llu / %llu kmers with Freq >= %u\n", (unsigned long long)O.n_kmers_distinct,
                (unsigned long long)O.n_kmers_solid, (unsigned long long)O.n_kmers_distinct, P.min_freq);
    std::printf("%llu / %llu reads pathed, %llu spanning junctions\n", (unsigned long long)O.n_reads_pathed, (unsigned long long)O.n_paths,
                (unsigned long long)O.n_reads_multipathed);
    std::printf("GPU ms: count %.2f graph %.2f path %.2f\n", O.ms_count, O.ms_graph, O.ms_path);
    // <prefix>.small_K.hbv  (HyperBasevector::writeBinary, paths/HyperBasevector.cc:121-125)
    std::vector<uint8_t> b;
    put(b, "BINWRITE", 8); put<int32_t>(b, 60);
    put_csr(b, O.n_vertices, O.from_off, O.from_v);
    put_csr(b, O.n_vertices, O.from_off, O.from_e);
    put_csr(b, O.n_vertices, O.to_off, O.to_e);
    put<uint64_t>(b, O.n_edge_objs);
    for (uint64_t e = 0; e < O.n_edge_objs; ++e) { put<uint32_t>(b, O.edge_len[e]); put(b, O.edge_packed + O.edge_byte_off[e], O.edge_byte_off[e + 1] - O.edge_byte_off[e]); }
    { std::ofstream f(out_dir + "/" + prefix + ".small_K.hbv", std::ios::binary); f.write((const char*)b.data(), (std::streamsize)b.size()); if (!f) { std::fprintf(stderr, "cannot write .hbv\n"); return 1; } }
    // <prefix>.small_K.paths  (WriteReadPathVec, paths/long/ReadPath.cc:6-20)
    b.clear();
    put<uint64_t>(b, O.n_paths);
    for (uint64_t r = 0; r < O.n_paths; ++r) {
        uint64_t n = O.path_off[r + 1] - O.path_off[r];
        put<int32_t>(b, O.path_offset[r]); put<uint16_t>(b, (uint16_t)n); put(b, O.path_edges + O.path_off[r], n * 4);
    }
    { std::ofstream f(out_dir + "/" + prefix + ".small_K.paths", std::ios::binary); f.write((const char*)b.data(), (std::streamsize)b.size()); if (!f) { std::fprintf(stderr, "cannot write .paths\n"); return 1; } }
    std::printf("Building first graph DONE!\n");
    w2rap_step2_free(&O);
    return 0;
}
