// step1_ingest.hip -- Step 1 of w2rap-contigger for a pair of fastq files on gfx950; C ABI in include/w2rap_step1.h.
//
//   reference                                                          here
//   getline x 4 per record, both files in lock step                     k1_count_nl + scan + k1_list_nl (line index of each file),
//     ExtractReads.cc:396-441                                            host check of the line counts (the reference's fatal conditions)
//   N -> A, Base::char2Val, q = c - 33, length check :416-452,470-474   k1_measure (lengths, byte counts, run counts, character checks)
//   PQVecEncoder::init/encode  feudal/PQVec.cc:17-127                   k1_encode (2-bit packing, raw qualities, one 3-byte block per run
//                                                                        of equal qualities cut at 255 -- what that encoder produces)
// One thread per read; a read's two lines are read with byte loads (neighbouring threads are ~350 B apart: every sector is fetched
// once and used by one thread -- the kernel is bound by HBM sectors of the text, 2 x ~190 B per read, read twice: measure, encode).
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "ctx.h"
#include "../../include/w2rap_step1.h"

extern "C" w2rap_step2_ctx* w2rap_step2_create(int device, char* err, size_t errlen);
extern "C" void w2rap_step2_destroy(w2rap_step2_ctx*);

namespace w2 {
namespace {

static inline unsigned grid_for(uint64_t n) { return (unsigned)((n + 255) / 256); }
constexpr unsigned TILE = 4 * 256 * 16;   // text bytes per block of the newline passes: four rounds of 16 per thread

// 0x80 in every byte of w that is a newline -- exact (no carries between bytes)
__device__ inline uint32_t nl_bytes(uint32_t w) {
    const uint32_t x = w ^ 0x0A0A0A0Au;
    return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
// thread t of a block: newline marks of text bytes [16 t, 16 t + 16) of the block's tile, bytes behind `len` masked out
__device__ inline void nl_marks(const uint8_t* __restrict__ text, uint64_t a, uint64_t len, uint32_t m[4]) {
    m[0] = m[1] = m[2] = m[3] = 0;
    if (a >= len) return;
    const uint4 v = *reinterpret_cast<const uint4*>(text + a);      // (the buffer is 16-byte aligned and padded to a multiple of 16)
    m[0] = nl_bytes(v.x); m[1] = nl_bytes(v.y); m[2] = nl_bytes(v.z); m[3] = nl_bytes(v.w);
    if (len - a < 16) {
        const unsigned keep = (unsigned)(len - a);
#pragma unroll
        for (unsigned k = 0; k < 4; ++k) {
            const int nb = (int)keep - 4 * (int)k;                  // bytes of word k in front of `len`
            if (nb <= 0) m[k] = 0; else if (nb < 4) m[k] &= (1u << (8 * nb)) - 1u;
        }
    }
}
__device__ inline unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// 16 newline marks (bit j: byte j is a newline) from the four 0x80-per-byte words
__device__ inline uint32_t pack_marks(const uint32_t m[4]) {
    uint32_t r = 0;
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) {
        const uint32_t x = m[k] >> 7;                                   // bit 0, 8, 16, 24
        r |= ((x & 1u) | ((x >> 7) & 2u) | ((x >> 14) & 4u) | ((x >> 21) & 8u)) << (4 * k);
    }
    return r;
}
// newlines per 16 KiB tile; the marks of every 16 bytes are kept (1/8 of the text) so that the listing pass does not read the text again
__global__ void __launch_bounds__(256) k1_count_nl(uint64_t len, const uint8_t* __restrict__ text, uint32_t* __restrict__ cnt, uint16_t* __restrict__ marks) {
    __shared__ unsigned part[4];
    unsigned c = 0;
#pragma unroll
    for (unsigned j = 0; j < 4; ++j) {
        uint32_t m[4];
        nl_marks(text, (uint64_t)blockIdx.x * TILE + j * 4096u + threadIdx.x * 16u, len, m);
        const uint32_t mk = pack_marks(m);
        marks[(uint64_t)blockIdx.x * 1024 + j * 256u + threadIdx.x] = (uint16_t)mk;
        c += (unsigned)__builtin_popcount(mk);
    }
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// nl[k] = position of the k-th newline, from the marks alone: a block takes one tile (16 KiB of text, 64 bytes = one u64 of marks per
// thread), scans the counts inside the block and writes the positions in order behind the tile's prefix
__global__ void __launch_bounds__(256) k1_list_nl(uint64_t nmarks, const uint16_t* __restrict__ marks, const uint64_t* __restrict__ excl, uint64_t* __restrict__ nl) {
    __shared__ unsigned part[4];
    const uint64_t i4 = (uint64_t)blockIdx.x * 1024 + threadIdx.x * 4u;       // first of this thread's four u16 marks (nmarks is a multiple of 4)
    uint64_t mm = i4 < nmarks ? *reinterpret_cast<const uint64_t*>(marks + i4) : 0ull;
    const unsigned c = (unsigned)__builtin_popcountll(mm);
    unsigned incl = c;                                             // inclusive scan over the wave
    const unsigned lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= (unsigned)o) incl += t; }
    if (lane == 63) part[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned before = incl - c;
    for (unsigned w = 0; w < (threadIdx.x >> 6); ++w) before += part[w];
    if (!c) return;
    uint64_t k = excl[blockIdx.x] + before;
    const uint64_t a = i4 * 16;
    while (mm) { nl[k++] = a + (unsigned)__builtin_ctzll(mm); mm &= mm - 1; }
}
struct FileIx { const uint8_t* text; const uint64_t* nl; uint64_t len, nnl; };
// line j of a file: [start, end)
__device__ inline void line_of(const FileIx& f, uint64_t j, uint64_t* s, uint64_t* e) {
    *s = j ? f.nl[j - 1] + 1 : 0;
    *e = j < f.nnl ? f.nl[j] : f.len;
}
// the file and record a read comes from.  Paired files: mates are interleaved, R1 from the first file, R2 from the second, read r is
// record r / 2 of file r & 1.  One interleaved file (f1.text == nullptr): read r is record r.  (Member-wise select: no struct in scratch.)
__device__ inline FileIx file_of(uint64_t r, const FileIx& f0, const FileIx& f1, uint64_t* rec) {
    const bool paired = f1.text != nullptr;
    const bool odd = paired && (r & 1);
    *rec = paired ? r >> 1 : r;
    return FileIx{odd ? f1.text : f0.text, odd ? f1.nl : f0.nl, odd ? f1.len : f0.len, odd ? f1.nnl : f0.nnl};
}
enum { E1_LEN = 1, E1_BASE = 2, E1_QUAL = 4, E1_LONG = 8 };
// per read: bases and packed bytes from the line index alone; the base/quality length check (ExtractReads.cc:442-452)
__global__ void __launch_bounds__(256) k1_lens(uint64_t n, FileIx f0, FileIx f1, uint32_t* __restrict__ rlen, uint32_t* __restrict__ nby,
                                                uint32_t* __restrict__ flags, unsigned long long* __restrict__ first_bad) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint64_t rec;
    const FileIx f = file_of(r, f0, f1, &rec);
    uint64_t s, e, qs, qe;
    line_of(f, 4 * rec + 1, &s, &e); line_of(f, 4 * rec + 3, &qs, &qe);
    const uint64_t L = e - s;
    if (qe - qs != L) { atomicOr(flags, (unsigned)E1_LEN); atomicMin(first_bad, (unsigned long long)r); }
    if (L > 0xFFFFFFF0ull) { atomicOr(flags, (unsigned)E1_LONG); atomicMin(first_bad, (unsigned long long)r); }      // lengths are 32-bit from here on
    rlen[r] = (uint32_t)L; nby[r] = (uint32_t)((L + 3) >> 2);
}
constexpr unsigned RPW = 4;               // reads per wave of the two wave-per-read kernels
// PQVecEncoder (feudal/PQVec.cc:17-127) on one read's qualities by one wave, one quality per lane and round.  What that encoder writes:
// one block [n, (v << 3) & 0xFF, v >> 5] per run of equal values, runs cut after 255, then a 0 byte.  A block opens where the value
// changes or 255 values after the run's first; it is written by the lane that opens it when the next opening is in the same round, else
// carried (start, value, block number) until a later round or the end of the read closes it.  -> the number of blocks; WRITE = false only
// counts.  SUB = 33 reads quality characters of the fastq text instead of raw qualities.  Four rounds' loads are issued together.
// (Staging the bytes in LDS to store whole dwords was measured slower: the kernel is bound by instruction issue, not by the byte stores.)
template <bool WRITE, unsigned SUB>
__device__ inline unsigned pq_read(const uint8_t* __restrict__ qp, uint32_t L, uint8_t* __restrict__ po, unsigned lane) {
    unsigned carry_prev = 0x100;                       // the value in front of the round (none: every value differs from it)
    uint32_t run_first = 0;                            // where the value last changed
    uint32_t open_at = 0; unsigned open_val = 0;       // the carried block, number nblocks - 1
    unsigned nblocks = 0;
    const bool cuts = L > 255;                         // only then can a run reach 256 values
    for (uint32_t sbase = 0; sbase < L; sbase += 256) {
        unsigned qq[4];
#pragma unroll
        for (unsigned t = 0; t < 4; ++t) { const uint32_t i = sbase + 64 * t + lane; qq[t] = i < L ? (unsigned)qp[i] - SUB : 0x200u; }
#pragma unroll
        for (unsigned t = 0; t < 4; ++t) {
            const uint32_t base = sbase + 64 * t;
            if (base >= L) break;
            const uint32_t i = base + lane;
            const bool valid = i < L;
            const unsigned q = qq[t];
            unsigned prev = __shfl_up(q, 1);
            if (lane == 0) prev = carry_prev;
            const bool chg = valid && q != prev;
            const uint64_t cm = __ballot(chg);
            uint64_t om = cm;
            bool open = chg;
            if (cuts) {
                const uint64_t le = cm & ((2ull << lane) - 1ull);                   // changes at or below this lane
                const uint32_t first = le ? base + (63 - __builtin_clzll(le)) : run_first;
                open = chg || (valid && !chg && (i - first) % 255u == 0);
                om = __ballot(open);
            }
            if (om) {
                const unsigned lowest = __builtin_ctzll(om);
                const uint64_t above = (lane == 63) ? 0ull : (om >> (lane + 1));
                if (WRITE) {
                    if (nblocks && lane == lowest) {                                 // this opening closes the carried block
                        uint8_t* b = po + 3u * (nblocks - 1);
                        b[0] = (uint8_t)(i - open_at); b[1] = (uint8_t)(open_val << 3); b[2] = (uint8_t)(open_val >> 5);
                    }
                    if (open && above) {
                        uint8_t* b = po + 3u * (nblocks + __builtin_popcountll(om & ((1ull << lane) - 1ull)));
                        b[0] = (uint8_t)(__builtin_ctzll(above) + 1); b[1] = (uint8_t)(q << 3); b[2] = (uint8_t)(q >> 5);
                    }
                }
                const unsigned highest = 63 - __builtin_clzll(om);
                open_at = base + highest; open_val = __shfl(q, highest);
                nblocks += __builtin_popcountll(om);
            }
            if (cm) run_first = base + (63 - __builtin_clzll(cm));
            carry_prev = __shfl(q, 63);
        }
    }
    if (WRITE && lane == 0) {
        if (nblocks) { uint8_t* b = po + 3u * (nblocks - 1); b[0] = (uint8_t)(L - open_at); b[1] = (uint8_t)(open_val << 3); b[2] = (uint8_t)(open_val >> 5); }
        po[3u * nblocks] = 0;
    }
    return nblocks;
}
// One wave per read, 4 characters per lane and round: the sequence line -> 2-bit codes ('N' -> 'A', ExtractReads.cc:416-419;
// Base::char2Val, dna/Bases.h:226), the quality line -> q = c - 33 (ExtractReads.cc:470-474); both checked (PQVec.cc:30-35).  Also the
// size of the read's PQVec: 3 bytes per run of equal qualities + 1 (reads of more than 255 bases, whose runs may be cut: pq_read).
// A lane's 4 characters are one unaligned dword load; the lane over the end of the line loads the line's LAST four bytes instead and
// shifts (no byte behind the line is touched), and stores those four qualities where they belong -- overlapping its neighbour's store
// with equal values.  Reads shorter than 4 bases take a byte-wise path.
__device__ inline uint32_t zero_bytes(uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu); }   // 0x80 per zero byte, exact
__device__ inline uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
struct Round1 { uint32_t v, qv; };
// the (up to) 4 characters at line offset i0 of a line of L >= 4 characters; *a = where the loaded dword starts
__device__ inline void ld_round(const uint8_t* sp, const uint8_t* qp, uint64_t L, uint64_t i0, Round1* in) {
    const uint64_t a = i0 + 4 <= L ? i0 : L - 4;
    in->v = ld32(sp + a); in->qv = ld32(qp + a);
}
__device__ inline unsigned unpack_round(uint64_t L, uint64_t base, unsigned lane, Round1 in, unsigned carry_q, uint8_t* __restrict__ bo, uint8_t* __restrict__ qo,
                                        unsigned* bad, unsigned* top_q) {
    const uint64_t i0 = base + 4u * lane;
    const bool act = i0 < L;
    const unsigned nch = !act ? 0u : (L - i0 < 4 ? (unsigned)(L - i0) : 4u);
    const uint64_t a = i0 + 4 <= L ? i0 : L - 4;
    const unsigned sh = 8u * (unsigned)(i0 - a);                                   // 0 for a whole lane, 8..24 for the one over the end
    uint32_t q = 0;
    if (act) {
        const uint32_t keep = nch == 4 ? 0xFFFFFFFFu : (1u << (8 * nch)) - 1u;
        const uint32_t v = ((in.v >> sh) & keep) | (0x41414141u & ~keep);          // 'A' behind the end
        const uint32_t up = v & 0xDFDFDFDFu;                                        // upper case
        const uint32_t isN = zero_bytes(v ^ 0x4E4E4E4Eu);
        const uint32_t ok = zero_bytes(up ^ 0x41414141u) | zero_bytes(up ^ 0x43434343u) | zero_bytes(up ^ 0x47474747u) | zero_bytes(up ^ 0x54545454u) | isN;
        *bad |= ok == 0x80808080u ? 0u : (unsigned)E1_BASE;
        // bits 2:1 of the ASCII code: A 00, C 01, G 11, T 10 -> 0 1 2 3; 'N' -> 0
        uint32_t code = ((v >> 1) & 0x03030303u) ^ ((v >> 2) & 0x01010101u);
        code &= ~((isN >> 7) * 3u);
        bo[i0 >> 2] = (uint8_t)(code | (code >> 6) | (code >> 12) | (code >> 18));
        const uint32_t qraw = in.qv - 0x21212121u;                                  // per byte exact when every character is >= 33 ...
        const uint32_t low = ~(((in.qv & 0x7F7F7F7Fu) + 0x5F5F5F5Fu) | in.qv) & 0x80808080u;   // ... 0x80 where a character is < 33
        *bad |= ((qraw & 0xC0C0C0C0u) || low) ? (unsigned)E1_QUAL : 0u;
        __builtin_memcpy(qo + a, &qraw, 4);
        q = (qraw >> sh) & keep;
    }
    // positions whose quality differs from the one in front of it (the read's first position always does)
    unsigned before = __shfl_up(q >> 24, 1);
    if (lane == 0) before = carry_q & 0xFF;
    const uint32_t x = q ^ ((q << 8) | before);
    uint32_t nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    if (lane == 0 && carry_q > 0xFF) nz |= 0x80u;
    if (nch < 4) nz &= (1u << (8 * nch)) - 1u;
    unsigned changes = 0;
#pragma unroll
    for (unsigned j = 0; j < 4; ++j) changes += __builtin_popcountll(__ballot((nz >> (8 * j + 7)) & 1u));
    *top_q = __shfl(q >> 24, 63);
    return changes;
}
#define PICK4(a, k) ((k) == 3 ? (a)[3] : (k) == 2 ? (a)[2] : (k) == 1 ? (a)[1] : (a)[0])
static_assert(RPW == 4, "PICK4");
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k1_unpack(uint64_t n, FileIx f0, FileIx f1, const uint64_t* __restrict__ boff, const uint64_t* __restrict__ qoff,
                                                  uint8_t* __restrict__ bases, uint8_t* __restrict__ quals, uint32_t* __restrict__ npq, uint32_t* __restrict__ flags,
                                                  unsigned long long* __restrict__ first_bad) {
    const unsigned lane = threadIdx.x & 63;
    const uint64_t r0 = (__builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6)) + (uint64_t)blockIdx.x * 4u) * RPW;
    // staged over the wave's RPW reads: all line offsets, then all first-round text loads, then the arithmetic and the stores
    const uint8_t *sp[RPW], *qp[RPW]; uint8_t *bo[RPW], *qo[RPW]; uint64_t L[RPW];
#pragma unroll
    for (unsigned k = 0; k < RPW; ++k) {
        const uint64_t r = r0 + k;
        L[k] = 0; sp[k] = qp[k] = nullptr; bo[k] = qo[k] = nullptr;
        if (r < n) {
            uint64_t rec;
            const FileIx f = file_of(r, f0, f1, &rec);
            uint64_t s, e, qs, qe;
            line_of(f, 4 * rec + 1, &s, &e); line_of(f, 4 * rec + 3, &qs, &qe);
            L[k] = e - s; sp[k] = f.text + s; qp[k] = f.text + qs;
            bo[k] = bases + boff[r]; qo[k] = quals + qoff[r];
        }
    }
    Round1 in[RPW];
#pragma unroll
    for (unsigned k = 0; k < RPW; ++k) {
        in[k].v = in[k].qv = 0;
        if (L[k] >= 4 && 4u * lane < L[k]) ld_round(sp[k], qp[k], L[k], 4u * lane, &in[k]);
    }
    unsigned slow = 0;
#pragma unroll
    for (unsigned k = 0; k < RPW; ++k) {
        const uint64_t r = r0 + k;
        if (r >= n) break;
        if (L[k] < 4 || L[k] > 255) { slow |= 1u << k; continue; }
        unsigned bad = 0, top = 0;
        const unsigned changes = unpack_round(L[k], 0, lane, in[k], 0x100u, bo[k], qo[k], &bad, &top);
        if (npq && lane == 0) npq[r] = 3 * changes + 1;
        if (bad) { atomicOr(flags, bad); atomicMin(first_bad, (unsigned long long)r); }
    }
    if (!slow) return;
#pragma unroll 1
    for (unsigned k = 0; k < RPW; ++k) {                                           // very short and long reads
        if (!(slow >> k & 1)) continue;
        const uint64_t Lk = PICK4(L, k);
        const uint8_t *spk = PICK4(sp, k), *qpk = PICK4(qp, k); uint8_t *bok = PICK4(bo, k), *qok = PICK4(qo, k);
        unsigned bad = 0, changes = 0;
        if (Lk < 4) {
            if (lane == 0) {
                unsigned packed = 0, prev = 0x100;
                for (unsigned j = 0; j < (unsigned)Lk; ++j) {
                    const unsigned c = spk[j], u = c & 0xDFu;
                    bad |= (u == 'A' || u == 'C' || u == 'G' || u == 'T' || c == 'N') ? 0u : (unsigned)E1_BASE;
                    packed |= ((c == 'N') ? 0u : (((c >> 1) & 3u) ^ ((c >> 2) & 1u))) << (2 * j);
                    const unsigned q = (uint8_t)(qpk[j] - 33);
                    bad |= q > 63 ? (unsigned)E1_QUAL : 0u;
                    qok[j] = (uint8_t)q;
                    changes += q != prev; prev = q;
                }
                if (Lk) bok[0] = (uint8_t)packed;
            }
            changes = __shfl(changes, 0);
        } else {
            unsigned top = 0x100u;
            for (uint64_t base = 0; base < Lk; base += 256) {
                Round1 rin{0, 0};
                if (base + 4u * lane < Lk) ld_round(spk, qpk, Lk, base + 4u * lane, &rin);
                unpack_round(Lk, base, lane, rin, top, bok, qok, &bad, &top);
            }
            changes = pq_read<false, 33>(qpk, (uint32_t)Lk, nullptr, lane);                  // runs may be longer than 255: count the cuts too
        }
        if (npq && lane == 0) npq[r0 + k] = 3 * changes + 1;
        if (bad) { atomicOr(flags, bad); atomicMin(first_bad, (unsigned long long)(r0 + k)); }
    }
}
// The same bytes for a read of 4..255 qualities in ONE round, four qualities per lane: no run can reach 256 values (no cuts) and nothing
// is carried.  Block openings are the bytes that differ from the byte in front of them (four ballots, one per byte position); a lane's
// blocks are numbered behind the openings of the lower lanes (v_mbcnt) and of its own lower bytes; a block ends at the next opening in
// the lane, else at the lowest opening of the next lane that has one (one cross-lane read), else at the end of the read.
// the lane's four qualities of a read of 4..255 (the lane over the end loads the last four and shifts)
__device__ inline uint32_t pq_load_short(const uint8_t* __restrict__ qp, uint32_t L, unsigned lane) {
    const uint32_t i0 = 4u * lane;
    if (i0 >= L) return 0;
    const uint32_t a = i0 + 4 <= L ? i0 : L - 4;
    uint32_t v; __builtin_memcpy(&v, qp + a, 4);
    return v >> (8u * (i0 - a));
}
__device__ inline void pq_write_short(uint32_t q, uint32_t L, uint8_t* __restrict__ po, unsigned lane) {
    const uint32_t i0 = 4u * lane;
    const bool act = i0 < L;
    const uint32_t nch = !act ? 0u : (L - i0 < 4 ? L - i0 : 4u);
    const uint32_t before = __shfl_up(q >> 24, 1);
    const uint32_t x = q ^ ((q << 8) | (before & 0xFFu));
    uint32_t nz = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u;
    if (lane == 0) nz |= 0x80u;                                                     // the first quality always opens a block
    nz = nch == 4 ? nz : nz & ((1u << (8 * nch)) - 1u);
    const uint32_t m = ((nz >> 7) & 1u) | ((nz >> 14) & 2u) | ((nz >> 21) & 4u) | ((nz >> 28) & 8u);    // bit j: byte j opens a block
    unsigned below = 0, total = 0;
    uint64_t any = 0;
#pragma unroll
    for (unsigned j = 0; j < 4; ++j) {
        const uint64_t b = __ballot((m >> j) & 1u);
        below = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, below));
        total += (unsigned)__builtin_popcountll(b);
        any |= b;
    }
    const uint64_t above = lane == 63 ? 0ull : any >> (lane + 1);
    const unsigned nl = lane + 1 + (above ? (unsigned)__builtin_ctzll(above) : 0u);
    const uint32_t low_next = (uint32_t)__shfl((int)(m ? __builtin_ctz(m) : 0), (int)(nl & 63u));
    const uint32_t end_lane = above ? 4u * nl + low_next : L;                       // where this lane's last block ends
    // the lane's blocks are neighbours in the output: their 3, 6, 9 or 12 bytes are put together in registers and leave as dwords
    // (+ one short / one byte), not as twelve byte stores
    uint64_t lo = 0; uint32_t hi = 0; unsigned cnt = 0;
#pragma unroll
    for (unsigned j = 0; j < 4; ++j) {
        const bool o = (m >> j) & 1u;
        const uint32_t rest = m >> (j + 1);
        const uint32_t len = rest ? 1u + (uint32_t)__builtin_ctz(rest) : end_lane - (i0 + j);
        const uint32_t v = (q >> (8 * j)) & 0xFFu;
        const uint32_t w = o ? (len & 0xFFu) | ((v << 11) & 0xFF00u) | ((v >> 5) << 16) : 0u;    // [len, v << 3, v >> 5]
        const unsigned pos = 24u * cnt;                                               // 0, 24, 48 or 72
        lo |= pos < 64 ? (uint64_t)w << pos : 0ull;
        hi |= pos == 48 ? w >> 16 : pos == 72 ? w << 8 : 0u;
        cnt += o ? 1u : 0u;
    }
    uint8_t* b = po + 3u * below;
    const uint32_t d0 = (uint32_t)lo, d1 = (uint32_t)(lo >> 32), d2 = hi;
    if (cnt >= 2) __builtin_memcpy(b, &d0, 4);
    if (cnt >= 3) __builtin_memcpy(b + 4, &d1, 4);
    if (cnt == 4) __builtin_memcpy(b + 8, &d2, 4);
    if (cnt == 1) { const uint16_t h = (uint16_t)d0; __builtin_memcpy(b, &h, 2); b[2] = (uint8_t)(d0 >> 16); }
    if (cnt == 2) { const uint16_t h = (uint16_t)d1; __builtin_memcpy(b + 4, &h, 2); }
    if (cnt == 3) b[8] = (uint8_t)d2;
    if (lane == 0) po[3u * total] = 0;
}
// the PQVec bytes of every read from the raw qualities
__global__ void __launch_bounds__(256) k1_pq_write(uint64_t n, const uint8_t* __restrict__ quals, const uint64_t* __restrict__ qoff, const uint64_t* __restrict__ pqoff,
                                                    uint8_t* __restrict__ pq) {
    const unsigned lane = threadIdx.x & 63;
    const uint64_t r0 = (__builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6)) + (uint64_t)blockIdx.x * 4u) * RPW;
    // staged over the wave's RPW reads like k1_unpack: all offsets, then all quality loads, then the arithmetic and the stores
    uint64_t q0[RPW], p0[RPW]; uint32_t L[RPW], qv[RPW];
#pragma unroll
    for (unsigned k = 0; k < RPW; ++k) {
        const uint64_t r = r0 + k;
        q0[k] = 0; p0[k] = 0; L[k] = 0;
        if (r < n) { q0[k] = qoff[r]; L[k] = (uint32_t)(qoff[r + 1] - q0[k]); p0[k] = pqoff[r]; }
    }
#pragma unroll
    for (unsigned k = 0; k < RPW; ++k) qv[k] = (L[k] >= 4 && L[k] <= 255) ? pq_load_short(quals + q0[k], L[k], lane) : 0u;
#pragma unroll
    for (unsigned k = 0; k < RPW; ++k) {
        if (r0 + k >= n) break;
        if (L[k] >= 4 && L[k] <= 255) pq_write_short(qv[k], L[k], pq + p0[k], lane);
        else pq_read<true, 0>(quals + q0[k], L[k], pq + p0[k], lane);
    }
}


// one file's text on the device (uploaded, or used in place) and its newline index
int stage_file(Ctx& c, const char* text, uint64_t len, int mem, const uint8_t** d_text, char* last) {
    hipStream_t st = c.stream;
    *last = 0;
    // (the newline passes read whole 16-byte words: in place only if the text is aligned and ends on a word boundary)
    if (mem == W2RAP_MEM_DEVICE && (reinterpret_cast<uintptr_t>(text) & 15) == 0 && len % 16 == 0) {
        *d_text = reinterpret_cast<const uint8_t*>(text);
        if (len) W2_HIP(hipMemcpy(last, text + len - 1, 1, hipMemcpyDeviceToHost));
        return 0;
    }
    uint8_t* d = c.alloc<uint8_t>(len + 64);
    if (!d) return W2RAP_E_HIP;
    if (len) W2_HIP(hipMemcpyAsync(d, text, len, mem == W2RAP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    if (len) { if (mem == W2RAP_MEM_DEVICE) W2_HIP(hipMemcpy(last, text + len - 1, 1, hipMemcpyDeviceToHost)); else *last = text[len - 1]; }
    *d_text = d;
    return 0;
}
int index_file(Ctx& c, const uint8_t* d, uint64_t len, char last, FileIx* ix, uint64_t* n_lines) {
    const uint64_t ntiles = (len + TILE - 1) / TILE;
    uint32_t* cnt = nullptr; uint64_t* excl = nullptr; uint16_t* marks = nullptr;
    W2_ALLOC(cnt, uint32_t, ntiles + 1); W2_ALLOC(excl, uint64_t, ntiles + 2); W2_ALLOC(marks, uint16_t, ntiles * 1024 + 8);
    if (ntiles) LAUNCH(c, "k1_count_nl", k1_count_nl, dim3((unsigned)ntiles), dim3(256), 0, len, d, cnt, marks);
    W2_TRY(exclusive_scan_u32_to_u64(c, cnt, excl, ntiles));
    uint64_t nnl = 0;
    W2_HIP(hipMemcpy(&nnl, excl + ntiles, 8, hipMemcpyDeviceToHost));
    uint64_t* nl = nullptr;
    W2_ALLOC(nl, uint64_t, nnl + 1);
    if (ntiles) LAUNCH(c, "k1_list_nl", k1_list_nl, dim3((unsigned)ntiles), dim3(256), 0, ntiles * 1024, marks, excl, nl);
    // getline: every newline ends a line; text behind the last newline is one more line
    *n_lines = nnl + ((len && last != '\n') ? 1 : 0);
    *ix = FileIx{d, nl, len, nnl};
    W2_HIP(hipStreamSynchronize(c.stream));
    c.release(cnt); c.release(excl); c.release(marks);
    return 0;
}

// device arrays that outlive this call when the reads are handed to Step 2 (freed with the context's reads)
template <class T>
T* alloc_out(Ctx& c, uint64_t count, bool install) {
    T* p = c.alloc<T>(count, !install);
    if (p && install) c.owned_reads.push_back(p);
    return p;
}

int step1(Ctx& c, const w2rap_step1_in& in, const w2rap_step1_params& P, w2rap_step1_out& out, bool install) {
    hipStream_t st = c.stream;
    const bool interleaved = (P.flags & W2RAP_STEP1_INTERLEAVED) != 0;        // one file, mates alternating (the reference's "unpaired" fastq branch)
    const uint8_t *t0 = nullptr, *t1 = nullptr; char last0 = 0, last1 = 0;
    {
        Timer t_up(st);
        W2_TRY(stage_file(c, in.fastq1, in.len1, in.mem, &t0, &last0));
        if (!interleaved) W2_TRY(stage_file(c, in.fastq2, in.len2, in.mem, &t1, &last1));
        out.ms_upload = t_up.stop();
    }
    FileIx f0, f1{nullptr, nullptr, 0, 0}; uint64_t L1 = 0, L2 = 0;
    {
        Timer t_index(st);
        W2_TRY(index_file(c, t0, in.len1, last0, &f0, &L1));
        if (!interleaved) W2_TRY(index_file(c, t1, in.len2, last1, &f1, &L2));
        out.ms_index = t_index.stop();
    }
    // the reference's loops (ExtractReads.cc:396-441, 493-530) in terms of the line counts: record i exists in a file iff line 4i does
    const uint64_t n1 = (L1 + 3) / 4, n2 = interleaved ? n1 : (L2 + 3) / 4, m = n1 < n2 ? n1 : n2;
    for (uint64_t i = 0; i < (interleaved ? 1u : 2u); ++i) {
        const uint64_t L = i ? L2 : L1, nrec = i ? n2 : n1;
        if (L % 4 && nrec <= m) { c.err = "See incomplete record in the fastq files (ExtractReads.cc:409-437, 498-523)"; return W2RAP_E_ARG; }
    }
    if (n1 != n2) { c.err = "The fastq files appear to be paired, yet have different numbers of records (ExtractReads.cc:399-405)"; return W2RAP_E_ARG; }
    const uint64_t n = interleaved ? n1 : 2 * n1;
    if (n > (1ull << 32) - 1024) { c.err = "more than 2^32 reads"; return W2RAP_E_LIMIT; }
    Timer t_enc(st);
    uint32_t *rlen, *nby, *npq = nullptr, *d_flags; unsigned long long* d_first; uint64_t *boff, *qoff, *pqoff = nullptr;
    W2_ALLOC(nby, uint32_t, n + 1); W2_ALLOC(d_flags, uint32_t, 4); W2_ALLOC(d_first, unsigned long long, 1);
    rlen = alloc_out<uint32_t>(c, n + 1, install); boff = alloc_out<uint64_t>(c, n + 2, install); qoff = alloc_out<uint64_t>(c, n + 2, install);
    if (!rlen || !boff || !qoff) return W2RAP_E_HIP;
    W2_HIP(hipMemsetAsync(d_flags, 0, 16, st)); W2_HIP(hipMemsetAsync(d_first, 0xFF, 8, st));
    auto check = [&]() -> int {                                                      // the reference's fatal inputs
        uint32_t h_flags = 0; unsigned long long h_first = 0;
        W2_HIP(hipMemcpyAsync(&h_flags, d_flags, 4, hipMemcpyDeviceToHost, st)); W2_HIP(hipMemcpyAsync(&h_first, d_first, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        if (!h_flags) return 0;
        const std::string where = " (first at read " + std::to_string(h_first) + ")";
        if (h_flags & E1_LONG) { c.err = "a read of 2^32 bases or more" + where; return W2RAP_E_LIMIT; }
        if (h_flags & E1_LEN) c.err = "See inconsistent base/quality lengths in the fastq files (ExtractReads.cc:442-452)" + where;
        else if (h_flags & E1_BASE) c.err = "illegal base character in a sequence line (Base::char2Val, dna/Bases.h:226)" + where;
        else if (interleaved && (n & 1)) c.err = "The file should be interlaced and hence have an even number of entries.  It does not. (ExtractReads.cc:556-563)";
        else c.err = "Your input reads are funny.  I found a quality score > 63, the maximum value that I allow (PQVec.cc:30-35)" + where;
        return W2RAP_E_ARG;
    };
    if (n) LAUNCH(c, "k1_lens", k1_lens, dim3(grid_for(n)), dim3(256), 0, n, f0, f1, rlen, nby, d_flags, d_first);
    W2_TRY(check());
    W2_TRY(exclusive_scan_u32_to_u64(c, nby, boff, n));
    W2_TRY(exclusive_scan_u32_to_u64(c, rlen, qoff, n));
    uint64_t nbytes = 0, nq = 0, npqb = 0;
    W2_HIP(hipMemcpy(&nbytes, boff + n, 8, hipMemcpyDeviceToHost)); W2_HIP(hipMemcpy(&nq, qoff + n, 8, hipMemcpyDeviceToHost));
    uint8_t *bases, *quals, *pq = nullptr;
    bases = alloc_out<uint8_t>(c, nbytes + 32, install); quals = alloc_out<uint8_t>(c, nq + 32, install);
    if (!bases || !quals) return W2RAP_E_HIP;
    const unsigned wgrid = (unsigned)((n + 4 * RPW - 1) / (4 * RPW));                 // 4 waves per block, RPW reads per wave
    const bool want_pq = !(P.flags & W2RAP_STEP1_NO_PQ);
    if (want_pq) { W2_ALLOC(npq, uint32_t, n + 1); W2_ALLOC(pqoff, uint64_t, n + 2); }
    if (n) LAUNCH(c, "k1_unpack", k1_unpack, dim3(wgrid), dim3(256), 0, n, f0, f1, boff, qoff, bases, quals, npq, d_flags, d_first);
    W2_TRY(check());
    if (interleaved && (n & 1)) { c.err = "The file should be interlaced and hence have an even number of entries.  It does not. (ExtractReads.cc:556-563)"; return W2RAP_E_ARG; }
    if (want_pq) {
        W2_TRY(exclusive_scan_u32_to_u64(c, npq, pqoff, n));
        W2_HIP(hipMemcpy(&npqb, pqoff + n, 8, hipMemcpyDeviceToHost));
        W2_ALLOC(pq, uint8_t, npqb + 16);
        if (n) LAUNCH(c, "k1_pq_write", k1_pq_write, dim3(wgrid), dim3(256), 0, n, quals, qoff, pqoff, pq);
    }
    out.ms_encode = t_enc.stop();
    out.n_reads = n; out.n_bases = nq; out.n_packed_bytes = nbytes; out.n_pq_bytes = want_pq ? npqb : 0;
    if (!(P.flags & W2RAP_STEP1_NO_FETCH)) {
        W2_TRY(dl(c, &out.bases_packed, bases, nbytes)); W2_TRY(dl(c, &out.base_byte_off, boff, n + 1)); W2_TRY(dl(c, &out.read_len, rlen, n));
        W2_TRY(dl(c, &out.quals, quals, nq)); W2_TRY(dl(c, &out.qual_off, qoff, n + 1));
        if (want_pq) { W2_TRY(dl(c, &out.pq, pq, npqb)); W2_TRY(dl(c, &out.pq_off, pqoff, n + 1)); }
    }
    W2_HIP(hipStreamSynchronize(st));
    if (install) { c.d_bases = bases; c.d_boff = boff; c.d_len = rlen; c.d_quals = quals; c.d_qoff = qoff; c.n = n; }
    return 0;
}

std::string g_profile1;

}  // namespace
}  // namespace w2

using namespace w2;

namespace w2 { void drop_reads(Ctx& c); void drop_results(Ctx& c); }

static int check_args(const w2rap_step1_in* in, const w2rap_step1_params* P, w2rap_step1_out* out, std::string* m) {
    if (!in || !P || !out) { *m = "null argument"; return W2RAP_E_ARG; }
    std::memset(out, 0, sizeof(*out));
    if ((in->len1 && !in->fastq1) || (in->len2 && !in->fastq2 && !(P->flags & W2RAP_STEP1_INTERLEAVED))) { *m = "null fastq buffer"; return W2RAP_E_ARG; }
    if (in->mem != W2RAP_MEM_HOST && in->mem != W2RAP_MEM_DEVICE) { *m = "bad mem kind"; return W2RAP_E_ARG; }
    return 0;
}
// per-kernel times of this run = the context's sums now minus the sums before
static void save_profile(Ctx& c, const std::vector<Ctx::ProfSum>& before) {
    (void)hipStreamSynchronize(c.stream);
    c.presolve();
    g_profile1.clear();
    for (auto& s : c.prof_sums) {
        double ms = s.ms; unsigned long long k = s.launches;
        for (auto& b : before) if (b.name == s.name) { ms -= b.ms; k -= b.launches; }
        if (!k) continue;
        char line[256]; std::snprintf(line, sizeof line, "%s %.4f %llu\n", s.name.c_str(), ms, k); g_profile1 += line;
    }
}

extern "C" {

int w2rap_step1_run(const w2rap_step1_in* in, const w2rap_step1_params* P, w2rap_step1_out* out, char* err, size_t errlen) {
    auto fail = [&](int code, const std::string& m) { if (err && errlen) std::snprintf(err, errlen, "%s", m.c_str()); return code; };
    std::string m;
    if (int rc = check_args(in, P, out, &m)) return fail(rc, m);
    char ebuf[512] = {0};
    w2rap_step2_ctx* h = w2rap_step2_acquire(P->device, ebuf, sizeof ebuf);
    if (!h) return fail(W2RAP_E_NO_DEVICE, ebuf);
    if (in->mem == W2RAP_MEM_DEVICE) (void)hipDeviceSynchronize();          // the caller's kernels may still be writing the text
    int rc = step1(h->c, *in, *P, *out, false);
    std::string msg = h->c.err;
    save_profile(h->c, {});
    if (rc) w2rap_step2_destroy(h); else w2rap_step2_release(h);     // (a failed context is not cached)
    if (rc) { w2rap_step1_free(out); return fail(rc, msg); }
    return 0;
}

int w2rap_step1_run_into_step2(w2rap_step2_ctx* h, const w2rap_step1_in* in, const w2rap_step1_params* P, w2rap_step1_out* out, char* err, size_t errlen) {
    auto fail = [&](int code, const std::string& m) { if (err && errlen) std::snprintf(err, errlen, "%s", m.c_str()); return code; };
    if (!h) return fail(W2RAP_E_ARG, "null context");
    std::string m;
    if (int rc = check_args(in, P, out, &m)) return fail(rc, m);
    Ctx& c = h->c;
    if (hipSetDevice(c.device) != hipSuccess) return fail(W2RAP_E_HIP, "hipSetDevice failed");
    drop_results(c);
    drop_reads(c);
    if (in->mem == W2RAP_MEM_DEVICE) (void)hipDeviceSynchronize();
    c.presolve();
    const std::vector<Ctx::ProfSum> before = c.prof_sums;
    int rc = step1(c, *in, *P, *out, true);
    save_profile(c, before);
    c.free_all();                                                             // the text copies, the line index, the PQVec bytes
    if (rc) { drop_reads(c); w2rap_step1_free(out); return fail(rc, c.err); }
    return 0;
}

void w2rap_step1_free(w2rap_step1_out* o) {
    if (!o) return;
    for (void* p : {(void*)o->bases_packed, (void*)o->base_byte_off, (void*)o->read_len, (void*)o->quals, (void*)o->qual_off, (void*)o->pq, (void*)o->pq_off}) std::free(p);
    std::memset(o, 0, sizeof(*o));
}

size_t w2rap_step1_profile(char* buf, size_t len) {
    if (buf && len) std::snprintf(buf, len, "%s", g_profile1.c_str());
    return g_profile1.size() + 1;
}

}  // extern "C"
