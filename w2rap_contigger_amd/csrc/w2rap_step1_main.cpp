// w2rap-step1 -- standalone Step 1 with the reference's file names and flags, for a pair of fastq files.
//
// Drop-in for `w2rap-contigger -r r1.fastq,r2.fastq -o <out_dir> --from_step 1 --to_step 1` (src/modules/w2rap-contigger.cc:300-323):
// writes <out_dir>/frag_reads_orig.fastb and <out_dir>/frag_reads_orig.qualp, which Step 2 (the reference's `--from_step 2`, or
// w2rap-step2) loads.  Files ending in .gz are inflated on the host with zlib (the reference reads them through its gzstream wrapper,
// ExtractReads.cc:372-389); all parsing, packing and quality compression happens in libw2rap_step2.so (HIP).
//
//   w2rap-step1 -r <r1.fastq[.gz]>,<r2.fastq[.gz]> -o <out_dir> [--device 0]
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "w2rap_step1.h"

namespace {

bool ends_with(const std::string& s, const char* suf) { const size_t n = std::strlen(suf); return s.size() >= n && !s.compare(s.size() - n, n, suf); }

bool slurp(const std::string& path, std::vector<char>& buf) {
    if (ends_with(path, ".gz")) {
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) return false;
        gzbuffer(f, 1u << 20);
        std::vector<char> chunk(1u << 24);
        for (;;) {
            const int n = gzread(f, chunk.data(), (unsigned)chunk.size());
            if (n < 0) { gzclose(f); return false; }
            if (n == 0) break;
            buf.insert(buf.end(), chunk.begin(), chunk.begin() + n);
        }
        gzclose(f);
        return true;
    }
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const std::streamsize n = f.tellg();
    f.seekg(0);
    buf.resize((size_t)n);
    return n == 0 || (bool)f.read(buf.data(), n);
}

// single-file feudal container (feudal/FeudalControlBlock.h:157-166): header, variable data, n+1 absolute offsets, fixed data
bool write_feudal(const std::string& path, uint64_t n, const uint8_t* var, const uint64_t* off, const void* fixed, uint64_t fixed_bytes, uint8_t szf, uint8_t szx, uint8_t sza) {
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    const uint64_t var_bytes = off[n], var_off = 24 + var_bytes, fixed_off = var_off + (n + 1) * 8;
    const uint32_t n32 = (uint32_t)(n & 0xFFFFFFFFull);
    const uint8_t hdr4[4] = {1, szf, szx, sza};
    f.write((const char*)&n32, 4); f.write((const char*)hdr4, 4); f.write((const char*)&var_off, 8); f.write((const char*)&fixed_off, 8);
    f.write((const char*)var, (std::streamsize)var_bytes);
    std::vector<uint64_t> abs(off, off + n + 1);
    for (auto& o : abs) o += 24;
    f.write((const char*)abs.data(), (std::streamsize)((n + 1) * 8));
    if (fixed_bytes) f.write((const char*)fixed, (std::streamsize)fixed_bytes);
    return (bool)f;
}

}  // namespace

int main(int argc, char** argv) {
    std::string reads, out_dir;
    int device = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
        const char* v = nullptr;
        if ((a == "-r" || a == "--read_files") && (v = next())) reads = v;
        else if ((a == "-o" || a == "--out_dir") && (v = next())) out_dir = v;
        else if (a == "--device" && (v = next())) device = std::atoi(v);
        else { std::fprintf(stderr, "usage: w2rap-step1 -r <r1.fastq[.gz]>,<r2.fastq[.gz]> -o <out_dir> [--device 0]\n"); return 2; }
    }
    const size_t comma = reads.find(',');
    if (reads.empty() || out_dir.empty() || comma == std::string::npos || reads.find(',', comma + 1) != std::string::npos) {
        std::fprintf(stderr, "w2rap-step1: -r takes one pair of fastq files (r1,r2) and -o an output directory\n");
        return 2;
    }
    std::vector<char> t1, t2;
    const std::string p1 = reads.substr(0, comma), p2 = reads.substr(comma + 1);
    if (!slurp(p1, t1)) { std::fprintf(stderr, "w2rap-step1: cannot read %s\n", p1.c_str()); return 1; }
    if (!slurp(p2, t2)) { std::fprintf(stderr, "w2rap-step1: cannot read %s\n", p2.c_str()); return 1; }
    w2rap_step1_in in{t1.data(), t1.size(), t2.data(), t2.size(), W2RAP_MEM_HOST};
    w2rap_step1_params P{device, 0};
    w2rap_step1_out out;
    char err[1024] = {0};
    const int rc = w2rap_step1_run(&in, &P, &out, err, sizeof err);
    if (rc) { std::fprintf(stderr, "w2rap-step1: %s (code %d)\n", err, rc); return 1; }
    std::printf("Reading input files: %llu reads, %llu bases; device ms: upload %.2f, line index %.2f, encode %.2f\n", (unsigned long long)out.n_reads,
                (unsigned long long)out.n_bases, out.ms_upload, out.ms_index, out.ms_encode);
    bool ok = write_feudal(out_dir + "/frag_reads_orig.fastb", out.n_reads, out.bases_packed, out.base_byte_off, out.read_len, out.n_reads * 4, 4, 16, 1)
           && write_feudal(out_dir + "/frag_reads_orig.qualp", out.n_reads, out.pq, out.pq_off, nullptr, 0, 0, 8, 1);
    w2rap_step1_free(&out);
    if (!ok) { std::fprintf(stderr, "w2rap-step1: cannot write into %s\n", out_dir.c_str()); return 1; }
    return 0;
}
