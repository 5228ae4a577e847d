// w2rap-step1 -- standalone Step 1 with the reference's file names and flags, for a pair of fastq files.
//
// Drop-in for `w2rap-contigger -r r1.fastq,r2.fastq -o <out_dir> --from_step 1 --to_step 1` (src/modules/w2rap-contigger.cc:300-323):
// writes <out_dir>/frag_reads_orig.fastb and <out_dir>/frag_reads_orig.qualp, which Step 2 (the reference's `--from_step 2`, or
// w2rap-step2) loads.  Files ending in .gz are inflated on the host with zlib (the reference reads them through its gzstream wrapper,
// ExtractReads.cc:372-389); all parsing, packing and quality compression happens in libw2rap_step2.so (HIP).
//
//   w2rap-step1 -r <r1.fastq[.gz]>,<r2.fastq[.gz]> -o <out_dir> [--device 0]
//
// Any number of fastq files: as in the reference (ExtractReads.cc:218-258) they are sorted by first read name, two files that share it are
// a pair (mates interleaved R1, R2), every other file is read on its own with alternating mates.
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
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

// ExtractReads.cc:230-243: the first line must start with '@', be longer than one character and not go on with ' ' or '/';
// the read name is what lies between the '@' and the first ' ' or '/'
bool first_read_name(const std::vector<char>& t, std::string* name) {
    size_t e = 0;
    while (e < t.size() && t[e] != '\n') ++e;
    if (e < 2 || t[0] != '@' || t[1] == ' ' || t[1] == '/') return false;
    size_t p = 0;
    while (p < e && t[p] != ' ' && t[p] != '/') ++p;
    name->assign(t.data() + 1, p - 1);
    return true;
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
        else { std::fprintf(stderr, "usage: w2rap-step1 -r <a.fastq[.gz]>[,<b.fastq[.gz]>...] -o <out_dir> [--device 0]\n"); return 2; }
    }
    std::vector<std::string> files;
    for (size_t p = 0; p <= reads.size();) {
        const size_t c = reads.find(',', p);
        const std::string f = reads.substr(p, c == std::string::npos ? std::string::npos : c - p);
        if (!f.empty()) files.push_back(f);
        if (c == std::string::npos) break;
        p = c + 1;
    }
    if (files.empty() || out_dir.empty()) { std::fprintf(stderr, "w2rap-step1: -r takes fastq files (a pair r1,r2; or files with alternating mates) and -o an output directory\n"); return 2; }
    // the reference's grouping (ExtractReads.cc:218-258): files sorted by first read name, two files with one name are a pair
    std::vector<std::vector<char>> text(files.size());
    std::vector<std::string> rn(files.size());
    for (size_t i = 0; i < files.size(); ++i) {
        if (!slurp(files[i], text[i])) { std::fprintf(stderr, "w2rap-step1: cannot read %s\n", files[i].c_str()); return 1; }
        if (!first_read_name(text[i], &rn[i])) { std::fprintf(stderr, "w2rap-step1: Something is wrong with the first line of your fastq file %s\n", files[i].c_str()); return 1; }
    }
    std::vector<size_t> order(files.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return rn[x] < rn[y]; });
    // outputs of all groups, concatenated
    std::vector<uint8_t> bases, pq; std::vector<uint64_t> boff{0}, pqoff{0}; std::vector<uint32_t> rlen;
    uint64_t n_reads = 0, n_bases = 0; float ms_up = 0, ms_ix = 0, ms_enc = 0;
    for (size_t j = 0; j < order.size();) {
        size_t k = j;
        while (k < order.size() && rn[order[k]] == rn[order[j]]) ++k;
        if (k - j > 2) { std::fprintf(stderr, "w2rap-step1: There are more than two fastq files that start with the read name %s: it's not clear how to pair the files\n", rn[order[j]].c_str()); return 1; }
        const bool pair = k - j == 2;
        const std::vector<char>& t1 = text[order[j]];
        static const std::vector<char> none;
        const std::vector<char>& t2 = pair ? text[order[j + 1]] : none;
        w2rap_step1_in in{t1.data(), t1.size(), t2.data(), t2.size(), W2RAP_MEM_HOST};
        w2rap_step1_params P{device, pair ? 0u : W2RAP_STEP1_INTERLEAVED};
        w2rap_step1_out out;
        char err[1024] = {0};
        const int rc = w2rap_step1_run(&in, &P, &out, err, sizeof err);
        if (rc) { std::fprintf(stderr, "w2rap-step1: %s (code %d)\n", err, rc); return 1; }
        const uint64_t b0 = bases.size(), q0 = pq.size();
        bases.insert(bases.end(), out.bases_packed, out.bases_packed + out.n_packed_bytes);
        pq.insert(pq.end(), out.pq, out.pq + out.n_pq_bytes);
        rlen.insert(rlen.end(), out.read_len, out.read_len + out.n_reads);
        for (uint64_t r = 1; r <= out.n_reads; ++r) { boff.push_back(b0 + out.base_byte_off[r]); pqoff.push_back(q0 + out.pq_off[r]); }
        n_reads += out.n_reads; n_bases += out.n_bases; ms_up += out.ms_upload; ms_ix += out.ms_index; ms_enc += out.ms_encode;
        w2rap_step1_free(&out);
        j = k;
    }
    std::printf("Reading input files: %llu reads, %llu bases; device ms: upload %.2f, line index %.2f, encode %.2f\n", (unsigned long long)n_reads,
                (unsigned long long)n_bases, ms_up, ms_ix, ms_enc);
    const bool ok = write_feudal(out_dir + "/frag_reads_orig.fastb", n_reads, bases.data(), boff.data(), rlen.data(), n_reads * 4, 4, 16, 1)
                 && write_feudal(out_dir + "/frag_reads_orig.qualp", n_reads, pq.data(), pqoff.data(), nullptr, 0, 0, 8, 1);
    if (!ok) { std::fprintf(stderr, "w2rap-step1: cannot write into %s\n", out_dir.c_str()); return 1; }
    return 0;
}
