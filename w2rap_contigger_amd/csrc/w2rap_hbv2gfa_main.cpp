// w2rap-hbv2gfa -- the reference's hbv2gfa tool (src/modules/hbv2gfa.cc) without line finding, with its flags and its output.
//
//   w2rap-hbv2gfa -i <in_prefix> -o <out_prefix> [-g <genome size in Kbp>] [--stats_only 1] [--device 0]
//
// Reads <in_prefix>.hbv (a .small_K.hbv or .large_K.hbv alike; <in_prefix>.paths, which the reference loads and does not use without
// -l, is not needed), prints the reference's statistics block, writes <out_prefix>_raw.gfa.  -l / --find_lines 1 (FindLines, serial
// graph surgery) is rejected.  Involution, canonical forms and all text are made in libw2rap_step2.so (HIP).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "w2rap_gfa.h"

namespace {

bool slurp(const std::string& path, std::vector<uint8_t>& buf) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    std::streamsize n = f.tellg();
    f.seekg(0);
    buf.resize((size_t)n);
    return n == 0 || (bool)f.read((char*)buf.data(), n);
}

// "BINWRITE", i32 K, from_, from_edge_obj_, to_edge_obj_ (each {u64 N; N x {u64 deg; i32[deg]}}), edges_ {u64 E; E x {u32 nbases; u8[ceil(nbases/4)]}}
struct Hbv {
    int32_t K = 0;
    uint64_t nv = 0;
    std::vector<uint64_t> from_off{0}, to_off{0}, byte_off{0};
    std::vector<int32_t> from_e, to_e;
    std::vector<uint8_t> packed; std::vector<uint32_t> len;
    bool load(const std::string& path, std::string& err) {
        std::vector<uint8_t> hb;
        if (!slurp(path, hb) || hb.size() < 12 || std::memcmp(hb.data(), "BINWRITE", 8)) { err = "cannot read " + path + " (not a BINWRITE .hbv)"; return false; }
        std::memcpy(&K, &hb[8], 4);
        size_t p = 12;
        auto need = [&](uint64_t bytes) { return bytes <= hb.size() - p; };
        for (int t = 0; t < 3; ++t) {
            if (!need(8)) goto bad;
            { uint64_t n; std::memcpy(&n, &hb[p], 8); p += 8;
              if (t == 0) nv = n; else if (n != nv) goto bad;
              std::vector<uint64_t>* off = t == 1 ? &from_off : t == 2 ? &to_off : nullptr;
              std::vector<int32_t>* lst = t == 1 ? &from_e : t == 2 ? &to_e : nullptr;
              for (uint64_t v = 0; v < n; ++v) {
                  if (!need(8)) goto bad;
                  uint64_t d; std::memcpy(&d, &hb[p], 8); p += 8;
                  if (d > (hb.size() - p) / 4) goto bad;
                  if (lst) { const size_t o = lst->size(); lst->resize(o + d); if (d) std::memcpy(lst->data() + o, &hb[p], 4 * d); off->push_back(lst->size()); }
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
        if (from_e.size() != len.size() || to_e.size() != len.size()) goto bad;
        return true;
    bad:
        err = "cannot read " + path + ": truncated or not a .hbv file";
        return false;
    }
};

}  // namespace

int main(int argc, char** argv) {
    std::string in_prefix, out_prefix;
    uint64_t genome_kb = 0; bool stats_only = false, find_lines = false; int device = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
        const char* v = nullptr;
        if ((a == "-i" || a == "--in_prefix") && (v = next())) in_prefix = v;
        else if ((a == "-o" || a == "--out_prefix") && (v = next())) out_prefix = v;
        else if ((a == "-g" || a == "--genome_size") && (v = next())) genome_kb = std::strtoull(v, nullptr, 10);
        else if (a == "--stats_only" && (v = next())) stats_only = std::atoi(v) != 0;
        else if ((a == "-l" || a == "--find_lines") && (v = next())) find_lines = std::atoi(v) != 0;
        else if (a == "--device" && (v = next())) device = std::atoi(v);
        else { std::fprintf(stderr, "usage: w2rap-hbv2gfa -i <in_prefix> -o <out_prefix> [-g Kbp] [--stats_only 1] [--device 0]\n"); return 1; }
    }
    if (in_prefix.empty() || out_prefix.empty()) { std::fprintf(stderr, "w2rap-hbv2gfa: -i and -o are required\n"); return 1; }
    if (find_lines) { std::fprintf(stderr, "w2rap-hbv2gfa: --find_lines is not implemented (the raw dump only)\n"); return 1; }
    std::printf("hbv2gfa from w2rap-contigger\nReading graph and paths...\n");
    Hbv g; std::string e;
    if (!g.load(in_prefix + ".hbv", e)) { std::fprintf(stderr, "w2rap-hbv2gfa: %s\n", e.c_str()); return 1; }
    std::printf("   DONE!\n=== Graph stats === \n");
    w2rap_gfa_in in{g.K, g.nv, g.len.size(), g.packed.data(), g.byte_off.data(), g.len.data(), g.from_off.data(), g.from_e.data(), g.to_off.data(), g.to_e.data()};
    w2rap_gfa_params P{device, stats_only ? W2RAP_GFA_STATS_ONLY : 0u, 1000 * genome_kb};
    w2rap_gfa_out out;
    char err[1024] = {0};
    const int rc = w2rap_gfa_dump(&in, &P, &out, err, sizeof err);
    if (rc) { std::fprintf(stderr, "w2rap-hbv2gfa: %s (code %d)\n", err, rc); return 1; }
    std::printf("Canonical graph sequences size: %llu\n", (unsigned long long)out.canonical_size);
    for (int j = 0; j < 9; ++j) std::printf("N%d: %llu\n", 10 * (j + 1), (unsigned long long)out.nxx[j]);
    if (P.genome_size) {
        std::printf("\nUser provided size: %llu\n", (unsigned long long)P.genome_size);
        for (int j = 0; j < 9; ++j) { if (out.ngxx[j] < 0) std::printf("NG%d: n/a\n", 10 * (j + 1)); else std::printf("NG%d: %lld\n", 10 * (j + 1), (long long)out.ngxx[j]); }
    }
    bool ok = true;
    if (!stats_only) {
        std::printf("Dumping gfa\n\n\n\n============GFA DUMP STARTING============\nGraph has %llu edges\nDumping edges\nDumping connections\n============GFA DUMP ENDED============\n\n\n\n",
                    (unsigned long long)g.len.size());
        std::ofstream f(out_prefix + "_raw.gfa", std::ios::binary);
        ok = (bool)f && (bool)f.write(out.gfa, (std::streamsize)out.gfa_len);
    }
    w2rap_gfa_free(&out);
    if (!ok) { std::fprintf(stderr, "w2rap-hbv2gfa: cannot write %s_raw.gfa\n", out_prefix.c_str()); return 1; }
    return 0;
}
