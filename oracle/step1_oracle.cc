// oracle/step1_oracle.cc -- TEST INFRASTRUCTURE (checker), not product code.
//
// Single-threaded CPU restatement of the reference's Step 1 for fastq files ("-r r1.fastq,r2.fastq", one frag library,
// frac = 1), src/paths/long/large/ExtractReads.cc:350-474 (the paired-fastq branch), :481-568 (one file with alternating mates; which
// branch a file takes is decided by first read names, :218-258, restated in oracle1.py) + feudal/PQVec.cc:17-127 (PQVecEncoder):
//   * four lines per record, read in lock step from both files (:396-441); a missing line is fatal ("incomplete record", :409-436),
//     different record counts are fatal (:399-405), base and quality lines must have equal length (:442-452);
//   * 'N' -> 'A' (:416-419), then Base::char2Val (dna/Bases.h:226: ACGTacgt only); q = char - 33 (:470-473), q > 63 is fatal
//     (PQVec.cc:30-35); mates interleaved R1, R2 (:474);
//   * every quality vector is compressed by PQVecEncoder::init (the block partition it finds, PQVec.cc:17-85 -- restated operation by
//     operation: its result is NOT the cheapest partition, and the files must match byte for byte) and ::encode (:87-127).
// Pinned against the reference's own frag_reads_orig.fastb/.qualp (tests/golden/*.step1.*, written by oracle/_ref/ref_step1): 1000-read
// inputs.  NOT restated on purpose (NOTES.md quirk Q18): the reference flushes its 10 M-entry quality buffer when it is full, BEFORE the
// pair that filled it has been stored (ExtractReads.cc:385-478), so in a run of 5 M pairs or more the reads 10 M k - 2 and 10 M k - 1
// are encoded from stale buffer contents.  This restatement (and the GPU path) encode the true qualities: byte-identical to the
// reference for every read except those two per 10 M.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Block { unsigned nQs, bits, minQ; };
// PowerOf2::ceilLg2lkp (math/PowerOf2.h:33-43) AS IT IS: the table holds 64 - ceil(log2 v) for v >= 2 (0 for v = 1), so every block
// that spans two different qualities looks 58..63 bits wide to the cost model and is never chosen -- what the encoder really produces is
// a run-length code, one 3-byte block per run of equal qualities (<= 255 long).  Quirk Q17: reproduced, or .qualp would not match.
static unsigned ceilLg2(unsigned v) {
    if (v <= 1) return 0;
    unsigned b = 0; while ((1u << b) < v) ++b;                  // the true ceil(log2 v), 1..6 for v in 2..64
    return 64 - b;
}
static unsigned blockSize(unsigned nQs, unsigned bits) { return (nQs * bits + 17 + 7) >> 3; }       // PQVec.h:53-59

// PQVecEncoder::init, PQVec.cc:17-85
static bool pq_blocks(const uint8_t* q, size_t n, std::vector<Block>& blocks, std::string& err) {
    blocks.clear();
    std::vector<unsigned> costs; costs.reserve(n + 1);
    costs.push_back(1);
    size_t itr = 0;
    while (itr != n) {
        if (q[itr] > 63) { err = "quality score " + std::to_string((unsigned)q[itr]) + " > 63 (PQVec.cc:30-35)"; return false; }
        size_t iCost = costs.size();
        unsigned minVal = q[itr] < 63u ? q[itr] : 63u, maxVal = q[itr];
        unsigned bits = ceilLg2(maxVal + 1u - minVal);
        unsigned prevCost = costs[--iCost];
        unsigned nQs = 1;
        unsigned bestCost = prevCost + blockSize(nQs, bits);
        Block best{1, bits, minVal};
        size_t itr2 = itr;
        ++itr;
        while (itr2 != 0 && nQs < 255) {
            unsigned val = q[--itr2];
            if (val > maxVal) maxVal = val;
            if (val < minVal) minVal = val;
            bits = ceilLg2(maxVal + 1u - minVal);
            prevCost = costs[--iCost];
            unsigned curCost = prevCost + blockSize(++nQs, bits);
            if (curCost < bestCost) { bestCost = curCost; best = Block{nQs, bits, minVal}; }
        }
        costs.push_back(bestCost);
        unsigned toRemove = best.nQs - 1;
        if (!toRemove) blocks.push_back(best);
        else {
            while (toRemove > blocks.back().nQs) { toRemove -= blocks.back().nQs; blocks.pop_back(); }
            if (toRemove == blocks.back().nQs) blocks.back() = best;
            else { blocks.back().nQs -= toRemove; blocks.push_back(best); }
        }
    }
    return true;
}
// PQVecEncoder::encode, PQVec.cc:87-127
static void pq_encode(const uint8_t* q, std::vector<Block> const& blocks, std::vector<uint8_t>& out) {
    size_t itr = 0;
    for (Block const& b : blocks) {
        uint64_t nQs = b.nQs, nBits = b.bits, minQ = b.minQ;
        out.push_back((uint8_t)nQs);
        uint64_t bits = nBits | (minQ << 3);
        out.push_back((uint8_t)bits); bits >>= 8;
        if (!nBits) { out.push_back((uint8_t)bits); itr += nQs; }
        else {
            uint64_t off = 1;
            while (nQs--) {
                uint64_t val = q[itr++] - minQ;
                bits |= val << off;
                if ((off += nBits) >= 8) { out.push_back((uint8_t)bits); off -= 8; bits >>= 8; }
            }
            if (off) out.push_back((uint8_t)bits);
        }
    }
    out.push_back(0);
}

struct Lines {                      // getline over a buffer
    const char* p; const char* end;
    bool next(const char** s, size_t* n) {
        if (p >= end) return false;                      // nothing left: getline fails
        const char* e = (const char*)memchr(p, '\n', end - p);
        *s = p; *n = (e ? e : end) - p;
        p = e ? e + 1 : end;
        return true;
    }
};

struct Oracle1 {
    std::string err;
    std::vector<uint8_t> bases;     // .fastb variable data: each read ceil(len/4) bytes
    std::vector<uint64_t> boff{0};
    std::vector<uint32_t> len;
    std::vector<uint8_t> quals;     // raw, concatenated
    std::vector<uint8_t> pq;        // PQVec byte strings
    std::vector<uint64_t> pqoff{0};
    bool addRead(const char* s, size_t n, const char* qs, size_t qn) {
        if (n != qn) { err = "inconsistent base/quality lengths (ExtractReads.cc:442-452)"; return false; }
        size_t o = bases.size();
        bases.resize(o + (n + 3) / 4, 0);
        for (size_t i = 0; i < n; ++i) {
            char c = s[i] == 'N' ? 'A' : s[i];
            unsigned v;
            switch (c) { case 'A': case 'a': v = 0; break; case 'C': case 'c': v = 1; break; case 'G': case 'g': v = 2; break; case 'T': case 't': v = 3; break;
                         default: err = std::string("illegal base character '") + c + "' (Base::char2Val, dna/Bases.h:226)"; return false; }
            bases[o + i / 4] |= (uint8_t)(v << (2 * (i % 4)));
        }
        boff.push_back(bases.size()); len.push_back((uint32_t)n);
        size_t qo = quals.size();
        for (size_t i = 0; i < n; ++i) quals.push_back((uint8_t)(qs[i] - 33));
        std::vector<Block> blocks;
        if (!pq_blocks(quals.data() + qo, n, blocks, err)) return false;
        pq_encode(quals.data() + qo, blocks, pq);
        pqoff.push_back(pq.size());
        return true;
    }
    void run(const char* f1, size_t n1, const char* f2, size_t n2) {
        Lines a{f1, f1 + n1}, b{f2, f2 + n2};
        const char *s1, *s2, *q1, *q2, *t; size_t l1, l2, m1, m2, u;
        while (true) {
            bool ok1 = a.next(&t, &u), ok2 = b.next(&t, &u);                       // header lines (:396)
            if (!ok1 && !ok2) break;
            if (ok1 != ok2) { err = "the files appear to be paired, yet have different numbers of records (ExtractReads.cc:399-405)"; return; }
            if (!a.next(&s1, &l1) | !b.next(&s2, &l2)) { err = "incomplete record (ExtractReads.cc:409-413)"; return; }
            if (!a.next(&t, &u) | !b.next(&t, &u)) { err = "incomplete record (ExtractReads.cc:424-428)"; return; }
            if (!a.next(&q1, &m1) | !b.next(&q2, &m2)) { err = "incomplete record (ExtractReads.cc:433-437)"; return; }
            if (!addRead(s1, l1, q1, m1) || !addRead(s2, l2, q2, m2)) return;
        }
    }
    // the "unpaired" fastq branch (ExtractReads.cc:481-568): one file, records taken in order (mates alternate); checked record by record,
    // and in the end the number of records must be even (:556-563)
    void run_single(const char* f, size_t n) {
        Lines a{f, f + n};
        const char *s1, *q1, *t; size_t l1, m1, u;
        size_t nrec = 0;
        while (a.next(&t, &u)) {
            if (!a.next(&s1, &l1)) { err = "incomplete record (ExtractReads.cc:498-502)"; return; }
            if (!a.next(&t, &u)) { err = "incomplete record (ExtractReads.cc:510-514)"; return; }
            if (!a.next(&q1, &m1)) { err = "incomplete record (ExtractReads.cc:519-523)"; return; }
            if (!addRead(s1, l1, q1, m1)) return;
            ++nrec;
        }
        if (nrec % 2) err = "should be interlaced and hence have an even number of entries (ExtractReads.cc:556-563)";
    }
};

}  // namespace

extern "C" {

void* oracle1_run(const char* f1, uint64_t n1, const char* f2, uint64_t n2) { auto* o = new Oracle1; o->run(f1, n1, f2, n2); return o; }
void* oracle1_run_single(const char* f, uint64_t n) { auto* o = new Oracle1; o->run_single(f, n); return o; }
const char* oracle1_error(void* h) { auto* o = (Oracle1*)h; return o->err.empty() ? nullptr : o->err.c_str(); }
void oracle1_free(void* h) { delete (Oracle1*)h; }
// sizes: [0] reads [1] base bytes [2] qualities [3] pq bytes
void oracle1_sizes(void* h, uint64_t* out) { auto* o = (Oracle1*)h; out[0] = o->len.size(); out[1] = o->bases.size(); out[2] = o->quals.size(); out[3] = o->pq.size(); }
void oracle1_get(void* h, uint8_t* bases, uint64_t* boff, uint32_t* len, uint8_t* quals, uint8_t* pq, uint64_t* pqoff) {
    auto* o = (Oracle1*)h;
    memcpy(bases, o->bases.data(), o->bases.size()); memcpy(boff, o->boff.data(), o->boff.size() * 8); memcpy(len, o->len.data(), o->len.size() * 4);
    memcpy(quals, o->quals.data(), o->quals.size()); memcpy(pq, o->pq.data(), o->pq.size()); memcpy(pqoff, o->pqoff.data(), o->pqoff.size() * 8);
}
// one quality vector -> PQVec bytes (for the encoder's own tests); returns the length, -1 on a quality > 63
int64_t oracle1_pq_encode(const uint8_t* q, uint64_t n, uint8_t* out) {
    std::vector<Block> blocks; std::string err; std::vector<uint8_t> v;
    if (!pq_blocks(q, n, blocks, err)) return -1;
    pq_encode(q, blocks, v);
    memcpy(out, v.data(), v.size());
    return (int64_t)v.size();
}

}  // extern "C"
