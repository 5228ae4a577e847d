// step2_count.hip -- phases a1..a6 of Step 2 on gfx950:
//   K0  k_good_len        quality window per read                 (BuildReadQGraph.cc:962-987)
//   K1  k_superkmers      canonical-minimizer super-k-mers: bucket histogram + one descriptor per record
//   K2  k_scatter_records descriptor -> 32-B record in its bucket  (K1+K2 replace the leaf loop :1062-1080 and
//                                                                   std::sort's partitioning :1081)
//   K3  k_count_buckets   per-bucket LDS hash count/merge          (collapse_entries :1002-1013, combine_Entries :943-949,
//                         + min_freq filter + histogram             filter :1094-1104)
//   K4  k_table_insert    lookup table over solid k-mers           (new BRQ_Dict + insertEntryNoLocking :1092-1099)
//   K5  k_prune_local,    adjacency prune, bucket-local in LDS,    (kmers/ReadPather.h:317-346)
//       k_prune           then the open bits against the table
//
// Design (MI355X-first, integer/HBM work, no MFMA):
//  * reads are consumed wavefront-per-read, lane = k-mer position: no divergence, the
//    packed read block is loaded once (coalesced byte loads) into LDS;
//  * instead of shipping one 17-B record per k-mer instance to HBM and back, consecutive
//    k-mers that share a canonical minimizer bucket travel as ONE 32-B super-k-mer
//    record (<= 64 k-mers): ~1.6 B/k-mer of partition traffic instead of 34 B;
//  * every bucket is sized to fit an LDS hash table, so counting (count saturating add,
//    context OR) never touches HBM; only distinct solid k-mers are written back;
//  * bucket sizes are data dependent: a bucket whose distinct set overflows the LDS table
//    is re-processed in 2,4,.. hash sub-passes (same semantics as MapReduceEngine.h:288-291);
//  * the kernels are bound by instruction issue (K1, K3), device atomics (K2, K4: 27 G/s) or random 32-B sectors (K5), not
//    by HBM bytes: K3 (issue) and K4 (atomics) therefore run side by side on two streams, bucket slice by bucket slice.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "ctx.h"

namespace w2 {

// =============================================================================== K0
constexpr unsigned QSLOTS = 256;       // partial sums / maxima of k_good_len
// One read per lane; scans the raw qualities backwards for the rightmost window of K
// consecutive q >= min_qual.  good_len is stored as uint16 like the reference (:1056).
// MASK: the bits are given (qmask: bit i = quality i >= min_qual, made on the host side of an upload that sends the raw bytes later).
template <bool MASK>
__global__ void __launch_bounds__(256) k_good_len(uint64_t n, const uint8_t* __restrict__ quals, const uint32_t* __restrict__ qmask,
                                                   const uint64_t* __restrict__ qoff, const uint32_t* __restrict__ len,
                                                   uint32_t min_qual, uint16_t* __restrict__ good,
                                                   unsigned long long* __restrict__ total_kmers,
                                                   uint32_t* __restrict__ max_len) {
    // The qualities of the block's 256 consecutive reads are contiguous.  They are read once with 16-byte loads and
    // reduced on the fly to ONE BIT per base (q >= min_qual) in LDS -- 5 KB per block instead of the 38 KB of raw bytes,
    // so the CU holds enough blocks to keep HBM busy -- and every lane then finds the rightmost run of K set bits of its
    // own read with shifted ANDs on 128-bit windows (a 150-base read is three windows).
    constexpr unsigned CH = 4096;                       // 16-byte chunks per block: 64 KB of qualities (256 reads x 256 bases)
    __shared__ __attribute__((aligned(16))) uint16_t gbits[CH + 16];
    __shared__ unsigned long long s_sum[4];
    __shared__ uint32_t s_max[4];
    const uint64_t r0 = (uint64_t)blockIdx.x * blockDim.x;
    const uint64_t r = r0 + threadIdx.x;
    const uint64_t rend = r0 + blockDim.x < n ? r0 + blockDim.x : n;
    const uint64_t base0 = qoff[r0], endo = qoff[rend];
    const unsigned shift = MASK ? (unsigned)(base0 & 31) : (unsigned)((reinterpret_cast<uintptr_t>(quals) + base0) & 15);
    const unsigned n16 = (unsigned)((endo - base0 + shift + 15) >> 4);
    const bool in_lds = (endo - base0 + shift + 15) >> 4 <= CH;
    if (MASK && in_lds) {
        uint32_t* gw32 = reinterpret_cast<uint32_t*>(gbits);
        const uint32_t* src = qmask + (base0 >> 5);                  // (the mask has 17 zeroed words behind its last one)
        for (unsigned i = threadIdx.x; i < (n16 + 10 + 1) / 2; i += blockDim.x) gw32[i] = 2 * i < n16 + 1 ? src[i] : 0u;
    } else if (in_lds) {
        const uint4* src = reinterpret_cast<const uint4*>(quals + base0 - shift);
        for (unsigned i = threadIdx.x; i < n16 + 10; i += blockDim.x) {
            unsigned bits = 0;
            if (i < n16) {
                const uint4 v = src[i];
                const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (unsigned j = 0; j < 16; ++j) bits |= (((d[j >> 2] >> (8 * (j & 3))) & 0xFFu) >= min_qual ? 1u : 0u) << j;
            }
            gbits[i] = (uint16_t)bits;                  // zero chunks behind the data: the window loads below read defined bits
        }
    }
    __syncthreads();
    unsigned long long mine = 0;
    uint32_t L = 0;
    if (r < n) {
        L = len[r];
        const uint64_t o = qoff[r];
        uint32_t g = 0;
        if (in_lds) {
            const uint32_t* gw = reinterpret_cast<const uint32_t*>(gbits);
            const uint32_t b0 = shift + (uint32_t)(o - base0);                 // the read's first bit
            // windows of 64 start positions, from the top: Y[i] = X[i..i+59] all set needs the 123 bits from i on
            for (int32_t s0 = (int32_t)((L - 1) & ~63u); L >= K && s0 >= 0; s0 -= 64) {
                const uint32_t bit = b0 + (uint32_t)s0, wi = bit >> 5, sh = bit & 31;
                const uint32_t w0 = gw[wi], w1 = gw[wi + 1], w2 = gw[wi + 2], w3 = gw[wi + 3], w4 = gw[wi + 4];
                uint64_t lo = ((uint64_t)w0 | ((uint64_t)w1 << 32)) >> sh, hi = ((uint64_t)w2 | ((uint64_t)w3 << 32)) >> sh;
                if (sh) { lo |= (uint64_t)w2 << (64 - sh); hi |= (uint64_t)w4 << (64 - sh); }
                const uint32_t valid = L - (uint32_t)s0;                         // bits of this read from s0 on
                if (valid < 128) { if (valid <= 64) { hi = 0; lo &= valid == 64 ? ~0ull : ((1ull << valid) - 1); } else hi &= (1ull << (valid - 64)) - 1; }
                // runs of >= 2, 4, 8, 16, 32, 60 set bits starting at each position (128-bit shifts)
                auto step = [&](unsigned sft) { const uint64_t l2 = (lo >> sft) | (hi << (64 - sft)), h2 = hi >> sft; lo &= l2; hi &= h2; };
                step(1); step(2); step(4); step(8); step(16); step(28);
                if (lo) { g = (uint32_t)s0 + (63u - (uint32_t)__builtin_clzll(lo)) + K; break; }
            }
        } else {
            uint32_t run = 0;
            for (uint32_t i = L; i-- > 0;) {
                const bool ok = MASK ? ((qmask[(o + i) >> 5] >> ((o + i) & 31)) & 1u) != 0 : quals[o + i] >= min_qual;
                if (!ok) run = 0;
                else if (++run == K) { g = i + K; break; }
            }
        }
        uint16_t g16 = (uint16_t)g;
        good[r] = g16;
        if (g16 > K) mine = g16 - (K - 1);
    }
    // block reduce
    for (int o = 32; o > 0; o >>= 1) {
        mine += __shfl_down(mine, o);
        uint32_t m2 = __shfl_down(L, o);
        L = m2 > L ? m2 : L;
    }
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[w] = mine; s_max[w] = L; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
        uint32_t m = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        // a slot per block residue: 2 x 195 k atomics on ONE address serialise at ~11 ns each (4 ms, the whole kernel)
        if (t) atomicAdd(&total_kmers[blockIdx.x & (QSLOTS - 1)], t);
        if (m) atomicMax(&max_len[blockIdx.x & (QSLOTS - 1)], m);
    }
}

// PQVec -> raw qualities, one read per lane (feudal/PQVec.cc:129-188 semantics).
__global__ void __launch_bounds__(256) k_decode_pq(uint64_t n, const uint8_t* __restrict__ pq, const uint64_t* __restrict__ pqoff,
                                                    uint8_t* __restrict__ quals, const uint64_t* __restrict__ qoff) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint8_t* b = pq + pqoff[r];
    const uint8_t* bend = pq + pqoff[r + 1];
    uint8_t* out = quals + qoff[r];
    uint8_t* oend = quals + qoff[r + 1];
    while (b < bend) {
        unsigned nqs = *b++;
        if (!nqs) break;
        unsigned nbits = b[0] & 7;
        unsigned minq = ((b[0] >> 3) | ((unsigned)b[1] << 5)) & 63;
        unsigned nbytes = (nqs * nbits + 9 + 7) >> 3;
        unsigned bitpos = 9;
        unsigned mask = (1u << nbits) - 1;
        for (unsigned i = 0; i < nqs && out < oend; ++i) {
            unsigned v = 0;
            if (nbits) {
                unsigned by = bitpos >> 3, sh = bitpos & 7;
                unsigned w = b[by] | ((by + 1 < nbytes ? (unsigned)b[by + 1] : 0u) << 8);
                v = (w >> sh) & mask;
                bitpos += nbits;
            }
            *out++ = (uint8_t)(minq + v);
        }
        b += nbytes;
    }
    while (out < oend) *out++ = 0;           // a PQVec that holds fewer values than the read has bases (a damaged file): defined, quality 0
}

int decode_pq(Ctx& c, const uint8_t* d_pq, const uint64_t* d_pqoff, uint8_t* d_quals, const uint64_t* d_qoff) {
    if (!c.n) return 0;
    LAUNCH(c, "k_decode_pq", k_decode_pq, dim3((unsigned)((c.n + 255) / 256)), dim3(256), 0, c.n, d_pq, d_pqoff, d_quals, d_qoff);
    W2_HIP(hipGetLastError());
    return 0;
}

// =============================================================================== K1+K2
// orders the LDS accesses of ONE wavefront (LDS executes a wave's operations in order; this only stops the compiler
// from reordering them) -- unlike __syncthreads() it does not wait for outstanding global memory operations
__device__ inline void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// (mmer_key, bucket_of: common.h -- the sharded graph phase and the pathing index use the same minimizers)

// One wavefront per read; a pass covers 128 k-mer positions (two per lane), i.e. a whole PE150 read.
// The canonical-minimizer of every k-mer is a sliding-window minimum over 46 m-mer keys; the window
// minimum is computed by doubling (2,4,8,16,32, then 32+16) entirely in registers with ds_bpermute
// lane shifts, so a pass is six cross-lane round trips deep and needs no barrier.  The m-mer key is
// mix32(canonical 15-mer): mix32 is a bijection, so distinct m-mers never tie and the choice depends
// only on the SET of canonical m-mers in the window (strand-symmetric, as it must be for a k-mer
// and its reverse complement to land in the same bucket).
__device__ inline uint32_t lane_shift(uint32_t lo, uint32_t hi, unsigned lane, unsigned d) {
    // value at position (lane + d) of the concatenation lo[0..63] ++ hi[0..63]
    const int idx = (int)(((lane + d) & 63u) << 2);
    const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)lo);
    const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)hi);
    return lane + d < 64 ? a : b;
}

// Output of K1: the number of records per bucket (fire-and-forget atomics) and one 8-B descriptor per record
// (bucket; start | nk-1 << 16 | hasL << 22 | hasR << 23).  Descriptors live at FIXED positions -- `spp` slots per
// (read, pass of 128 k-mer positions), unused slots keep the 0xFFFFFFFF the array was filled with -- so the wave
// neither reserves anything nor waits for memory; the rare pass with more than `spp` records appends the surplus
// to a small overflow list.  K2 turns every descriptor into a 32-B record with one thread per slot, so the slot
// reservations of a whole wavefront are in flight together instead of one read's at a time.
template <unsigned MINW>
__global__ void __launch_bounds__(256, MINW) k_superkmers(uint64_t n, uint32_t chunk, const uint8_t* __restrict__ bases,
                                                    const uint64_t* __restrict__ boff, const uint16_t* __restrict__ good,
                                                    uint32_t nb, uint32_t pb_lo, uint32_t pb_hi /* this pass keeps buckets [pb_lo, pb_hi) */,
                                                    uint32_t* __restrict__ bcount /* [pb_hi - pb_lo] */,
                                                    uint32_t nbl_part, uint32_t inv_nbl, unsigned long long* __restrict__ part_kmers,
                                                    uint2* __restrict__ s_desc, uint32_t spp, uint32_t npass,
                                                    uint32_t* __restrict__ o_read, uint32_t* __restrict__ o_bkt, uint32_t* __restrict__ o_meta,
                                                    uint32_t* __restrict__ o_rank,
                                                    uint64_t ov_cap, unsigned long long* __restrict__ ov_cursor /*[0] entries*/) {
    // four independent wavefronts per block (a CU holds more 256-thread blocks than 64-thread ones); no block barriers
    __shared__ uint2 dbuf_[4][128];       // the descriptors of one pass, in record order
    __shared__ uint32_t spart_[4][64];    // multi-GPU: k-mers this wave sends to each owner rank (owner = bucket / nbl_part)
    const unsigned lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
    if (part_kmers) { spart_[wv_][lane] = 0; wave_lds_fence(); }
    uint2* dbuf = dbuf_[wv_];
    // The window of a read that a pass needs (<= 206 bases = 52 bytes) lives in REGISTERS: lanes 0..15 hold the sixteen aligned dwords from
    // the dword that contains the window's first byte (an aligned word that holds a valid byte never leaves the allocation), and a lane
    // fetches the two dwords of its 15-mer from them with ds_bpermute.  (Rounds 1-3 staged the bytes in LDS: three fences, a clearing
    // store, a byte store -- 60 of the pass's 300 VALU instructions.)
    auto win_load = [&](uint64_t first_byte, unsigned nbytes) -> uint32_t {          // nbytes: valid bytes from first_byte on
        const uintptr_t a = reinterpret_cast<uintptr_t>(bases) + first_byte;         // (aligned by ADDRESS: the array itself may start anywhere)
        const unsigned sb = (unsigned)(a & 3);
        return 4 * lane < sb + nbytes ? reinterpret_cast<const uint32_t*>(a & ~uintptr_t(3))[lane & 15u] : 0u;
    };
    // A wave takes `chunk` CONSECUTIVE reads and the grid has one wave per chunk: blocks are dispatched in order, so the reads
    // -- and with them the ranks the histogram atomic hands out inside a bucket -- advance roughly in read order, the order in
    // which the scatter pass writes the records (neighbouring slots of a bucket are then written close in time and meet in L2).
    const uint64_t stride = 1;
    uint64_t r = ((uint64_t)blockIdx.x * 4 + wv_) * chunk;
    const uint64_t r_end = r + chunk < n ? r + chunk : n;
    // Software pipeline over this wave's reads: the quality window and byte offset of the read after next and the
    // first 60 packed bytes of the next read are loaded while the current read is cut, so no read waits for HBM.
    unsigned gl_c = 0, gl_n = 0; uint64_t off_c = 0, off_n = 0; uint32_t win_c = 0;
    if (r < r_end) { gl_c = good[r]; off_c = boff[r]; }
    if (r + stride < r_end) { gl_n = good[r + stride]; off_n = boff[r + stride]; }
    if (r < r_end && lane < 16 && gl_c > K) win_c = win_load(off_c, (gl_c + 3) >> 2);
    for (; r < r_end; r += stride) {
        const unsigned gl = gl_c;
        const uint64_t off_cur = off_c;
        const uint32_t win0 = win_c;
        // look ahead
        unsigned gl_nn = 0; uint64_t off_nn = 0;
        if (r + 2 * stride < r_end) { gl_nn = good[r + 2 * stride]; off_nn = boff[r + 2 * stride]; }
        win_c = 0;
        if (r + stride < r_end && lane < 16 && gl_n > K) win_c = win_load(off_n, (gl_n + 3) >> 2);
        gl_c = gl_n; off_c = off_n; gl_n = gl_nn; off_n = off_nn;
        if (gl <= K) {                                           // strict, BuildReadQGraph.cc:1064: no records, empty slots
            for (unsigned i = lane; i < spp * npass; i += 64) s_desc[r * npass * spp + i] = make_uint2(0u, NONE32);
            continue;
        }
        const unsigned nk_total = gl - (K - 1);
        const unsigned nbytes_read = (gl + 3) >> 2;
        for (unsigned c0 = 0; c0 < nk_total; c0 += 128) {
            // ---- stage the window of the read this pass needs: bases [c0-1, c0+189) ----
            const unsigned first_base = c0 ? c0 - 1 : 0;
            const unsigned b0a = (first_base >> 2) & ~3u;        // window start byte, dword aligned in the read
            // this pass's window: the prefetched dwords (first pass) or sixteen dwords from byte b0a of the read on
            uint32_t wdw = win0;
            if (c0) wdw = lane < 16 ? win_load(off_cur + b0a, nbytes_read - b0a) : 0u;
            const unsigned sbit = 8u * (unsigned)((reinterpret_cast<uintptr_t>(bases) + off_cur) & 3);   // the window's first byte inside its first dword (b0a is a multiple of 4)
            const unsigned sbase = 4 * b0a;                      // read base of the window's first byte
            // ---- canonical m-mer keys at m-mer positions c0+lane, c0+64+lane, c0+128+lane ----
            uint32_t k0, k1, k2;
            {
                uint32_t kk[3];
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    const unsigned j = c0 + h * 64 + lane;
                    uint32_t key = 0xFFFFFFFFu;
                    // (every lane takes part in the two cross-lane fetches; lanes without an m-mer read dword 0)
                    const bool has = j + MMER <= gl && (h < 2 || lane < WIN - 1);
                    const unsigned o = has ? sbit + 2 * (j - sbase) : 0u, wi = o >> 5, sh = o & 31;
                    const uint32_t w0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(wi << 2), (int)wdw);
                    const uint32_t w1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((wi + 1) << 2), (int)wdw);
                    if (has) {
                        const uint32_t f = __funnelshift_r(w0, w1, sh) & 0x3FFFFFFFu;                 // 15 bases, LSB first
                        // reverse complement in 32-bit ops: complement, reverse the 16 two-bit groups, drop the padding group
                        uint32_t rc = __brev(~f & 0x3FFFFFFFu);
                        rc = (((rc & 0x55555555u) << 1) | ((rc >> 1) & 0x55555555u)) >> 2;
                        key = mmer_key(f < rc ? f : rc);
                    }
                    kk[h] = key;
                }
                k0 = kk[0]; k1 = kk[1]; k2 = kk[2];
            }
            // ---- sliding-window minimum over WIN=46 keys, by 16-lane rows: the window [p, p+45] is the suffix of p's
            //      row from p, the prefix of (p+45)'s row up to p+45, and the one or two whole rows in between.
            //      Suffix/prefix minima inside rows are DPP scans (VALU only), row minima travel through SGPRs
            //      (v_readlane), and only the prefix at p+45 is a cross-lane fetch: 2 ds_bpermute per half-pass.
            constexpr uint32_t INF = 0xFFFFFFFFu;
            auto row_pref = [](uint32_t x) {
                x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x111, 0xF, 0xF, false));
                x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x112, 0xF, 0xF, false));
                x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x114, 0xF, 0xF, false));
                x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x118, 0xF, 0xF, false));
                return x;
            };
            auto row_suff = [](uint32_t x) {
                x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x101, 0xF, 0xF, false));
                x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x102, 0xF, 0xF, false));
                x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x104, 0xF, 0xF, false));
                x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, 0x108, 0xF, 0xF, false));
                return x;
            };
            const uint32_t P0 = row_pref(k0), P1 = row_pref(k1), P2 = row_pref(k2), S0 = row_suff(k0), S1 = row_suff(k1);
            uint32_t F[12];                                          // row minima, rows 0..11 of the 192 keys (wave-uniform)
#pragma unroll
            for (int R = 0; R < 4; ++R) {
                F[R] = (uint32_t)__builtin_amdgcn_readlane((int)P0, 16 * R + 15);
                F[4 + R] = (uint32_t)__builtin_amdgcn_readlane((int)P1, 16 * R + 15);
                F[8 + R] = (uint32_t)__builtin_amdgcn_readlane((int)P2, 16 * R + 15);
            }
            const unsigned rowi = lane >> 4;
            const bool two = (lane & 15u) >= 3;                      // (p+45) lies three rows after p's row: two whole rows between
            uint32_t mkh[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t g1a = F[4 * h + 1], g1b = F[4 * h + 2], g1c = F[4 * h + 3], g1d = F[4 * h + 4];
                const uint32_t g2a = min(g1a, F[4 * h + 2]), g2b = min(g1b, F[4 * h + 3]), g2c = min(g1c, F[4 * h + 4]), g2d = min(g1d, F[4 * h + 5]);
                const uint32_t one_row = rowi == 0 ? g1a : rowi == 1 ? g1b : rowi == 2 ? g1c : g1d;
                const uint32_t two_rows = rowi == 0 ? g2a : rowi == 1 ? g2b : rowi == 2 ? g2c : g2d;
                const uint32_t pend = h ? lane_shift(P1, P2, lane, WIN - 1) : lane_shift(P0, P1, lane, WIN - 1);
                mkh[h] = min(min(h ? S1 : S0, pend), two ? two_rows : one_row);
            }
            const uint32_t mk0 = mkh[0], mk1 = mkh[1];
            // ---- two half-passes of 64 k-mer positions.  All global traffic of both halves (slot
            //      reservation, bucket base) is issued before any of it is consumed, so a read pays
            //      one atomic round trip, not two. ----
            unsigned nkh[2] = {0, 0}; uint32_t bkh[2] = {0, 0}; bool sth[2] = {false, false};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned cc0 = c0 + 64 * h;
                const unsigned p = cc0 + lane;
                const bool valid = p < nk_total;
                const uint32_t bkt = valid ? bucket_of(h ? mk1 : mk0, nb) : 0xFFFFFFFFu;
                const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bkt, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane 0 unused)
                bool start = valid && (lane == 0 || prev != bkt);
                unsigned long long smask = __ballot(start);
                unsigned nvalid = cc0 < nk_total ? nk_total - cc0 : 0; if (nvalid > 64) nvalid = 64;
                // a record holds at most 63 k-mers (32 B): a half-pass that is ONE run of 64 is cut before its last k-mer
                if (REC_MAXK < 64 && smask == 1ull && nvalid == 64) { start |= lane == 63; smask |= 1ull << 63; }
                unsigned long long rest = lane < 63 ? (smask >> (lane + 1)) : 0ull;
                unsigned nxt = rest ? lane + 1 + __builtin_ctzll(rest) : nvalid;
                // multi-pass counting (MapReduceEngine.h:288-291): records of buckets outside this pass's range are dropped here and
                // cut again in their own pass; bucket numbers are relative to the range
                const bool mine = bkt - pb_lo < pb_hi - pb_lo;
                sth[h] = start && mine; bkh[h] = bkt - pb_lo; nkh[h] = start ? nxt - lane : 0;
            }
            unsigned cnt = 0;
            const uint64_t slot0 = (r * npass + c0 / 128) * spp;
            // the histogram atomic RETURNS the record's rank inside its bucket: it travels in the descriptor (16 bits; the rare
            // bucket that gets more than 65535 records from these reads sends the surplus through the overflow list), so the
            // scatter pass needs no atomic of its own.  Both halves' atomics are in flight before either result is used.
            uint32_t rk[2] = {0, 0};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (sth[h]) {
                    rk[h] = atomicAdd(&bcount[bkh[h]], 1u);
                    if (part_kmers) {                                        // bucket / nbl_part by reciprocal (+1 correction)
                        uint32_t pt = nbl_part > 1 ? __umulhi(bkh[h], inv_nbl) : bkh[h];
                        if ((pt + 1) * nbl_part <= bkh[h]) ++pt;
                        atomicAdd(&spart_[wv_][pt & 63], nkh[h]);
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned long long m = __ballot(sth[h]);
                if (sth[h]) {
                    const unsigned p = c0 + 64 * h + lane, nk = nkh[h];
                    const uint32_t meta = p | ((nk - 1) << 16) | (p > 0 ? 1u << 22 : 0u) | ((p + nk - 1) < (nk_total - 1) ? 1u << 23 : 0u);
                    const unsigned j = cnt + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1));
                    if (j < spp && rk[h] < 65536u) dbuf[j] = make_uint2(bkh[h] | (rk[h] << 24), meta | ((rk[h] >> 8) << 24));
                    else {
                        if (j < spp) dbuf[j] = make_uint2(0u, NONE32);
                        const unsigned long long o = atomicAdd(ov_cursor, 1ull);
                        if (o < ov_cap) { o_read[o] = (uint32_t)r; o_bkt[o] = bkh[h]; o_meta[o] = meta; o_rank[o] = rk[h]; }
                    }
                }
                cnt += (unsigned)__builtin_popcountll(m);
            }
            // the pass's slots as whole, coalesced 8-B stores (unused ones marked empty)
            wave_lds_fence();
            for (unsigned i = lane; i < spp; i += 64) s_desc[slot0 + i] = i < cnt ? dbuf[i] : make_uint2(0u, NONE32);
            wave_lds_fence();
        }
        // passes this read does not have (shorter than the longest read)
        for (unsigned i = ((nk_total + 127) / 128) * spp + lane; i < spp * npass; i += 64) s_desc[r * npass * spp + i] = make_uint2(0u, NONE32);
    }
    if (part_kmers) {
        wave_lds_fence();
        const uint32_t v = spart_[wv_][lane];
        // 64 copies of the 64 totals: millions of waves adding to ONE address would serialise at ~11 ns each
        if (v) atomicAdd(&part_kmers[(blockIdx.x & 63u) * 64u + lane], (unsigned long long)v);
    }
}


// ------------------------------------------------------------------------------- K1, one LANE per read (round 4)
// The wavefront-per-read kernel above spends ~300 VALU instructions of a whole wavefront on the 136 m-mers of a PE150 read (cross-lane
// scans for the window minimum, wave-wide control flow per record) -- measured in round 4: with its atomics AND its descriptor stores
// removed it still takes 19.2 of its 22.9 ms (profiles/r04_k1_diag.txt), it is bound by its own instruction stream.  Here a lane walks
// ITS read base by base: the forward and the reverse-complement 15-mer roll (five operations), the sliding-window minimum over 46 keys is
// the block decomposition of van Herk / Gil-Werman with blocks of 16 keys held in registers -- a window is the suffix of its first block,
// one or two whole blocks and the running prefix of the current one: one v_min3 per position --, and a run of equal buckets ends with ~10 instructions of the few
// lanes it concerns.  Descriptors (bucket, meta) are staged in LDS at the lane's S slots; when the 64 reads are done the wave turns the
// staged descriptors into final ones 64 at a time -- the histogram atomic returns the rank -- and stores them coalesced: the 64 reads'
// slots are one contiguous piece of s_desc.  Runs are cut at bucket changes and when a record is full (REC_MAXK = 63 k-mers);
// the kernel above also cuts at every multiple of 64 (its half-passes).
template <unsigned SMAX, bool ALIGN64>
__global__ void __launch_bounds__(256, 4) k_superkmers_lane(uint64_t n, const uint8_t* __restrict__ bases, const uint64_t* __restrict__ boff,
                                                            const uint16_t* __restrict__ good, uint32_t nb, uint32_t pb_lo, uint32_t pb_hi,
                                                            uint32_t* __restrict__ bcount, uint32_t nbl_part, uint32_t inv_nbl,
                                                            unsigned long long* __restrict__ part_kmers, uint2* __restrict__ s_desc, uint32_t S,
                                                            uint32_t* __restrict__ o_read, uint32_t* __restrict__ o_bkt, uint32_t* __restrict__ o_meta,
                                                            uint32_t* __restrict__ o_rank, uint64_t ov_cap, unsigned long long* __restrict__ ov_cursor) {
    __shared__ uint2 stage_[4][64 * SMAX];
    __shared__ uint32_t spart_[4][64];
    const unsigned lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
    const uint64_t r0 = ((uint64_t)blockIdx.x * 4 + wv_) * 64;
    if (r0 >= n) return;                                     // (the whole wavefront; the kernel has no block barrier)
    uint2* stage = stage_[wv_];
    if (part_kmers) spart_[wv_][lane] = 0;
    for (unsigned u = 0; u < S; ++u) stage[lane + 64 * u] = make_uint2(0u, NONE32);
    wave_lds_fence();
    constexpr uint32_t INF = 0xFFFFFFFFu;
    const uint64_t r = r0 + lane;
    unsigned gl = r < n ? good[r] : 0;
    if (gl <= K) gl = 0;                                     // strict, BuildReadQGraph.cc:1064
    const unsigned nk = gl ? gl - (K - 1) : 0;
    unsigned maxgl = gl;
#pragma unroll
    for (int d = 32; d; d >>= 1) maxgl = max(maxgl, (unsigned)__shfl_xor((int)maxgl, d));
    maxgl = (unsigned)__builtin_amdgcn_readfirstlane((int)maxgl);
    if (maxgl) {
        // the read as ALIGNED dwords (an aligned word that holds a valid byte never leaves the allocation), sixteen bases per refill, one
        // refill ahead; words past the read's last one are not loaded (the last one again: bases past the good length are never used)
        const uint64_t off = gl ? boff[r] : 0;
        const unsigned mis = (unsigned)((reinterpret_cast<uintptr_t>(bases) + off) & 3);
        const unsigned sb8 = 8u * mis;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(bases + ((int64_t)off - (int64_t)mis));
        const unsigned ulast = gl ? (mis + ((gl + 3) >> 2) - 1) >> 2 : 0;
        uint32_t dA, dB, W; unsigned un = 3;
        { const uint32_t d0 = q[0]; dA = q[min(1u, ulast)]; dB = q[min(2u, ulast)]; W = __funnelshift_r(d0, dA, sb8); }
        uint32_t f = 0, rc = 0;
        auto roll = [&]() {                                  // takes the next base into the two 15-mers
            const uint32_t b = W & 3u; W >>= 2;
            f = (f >> 2) | (b << 28);
            rc = ((rc << 2) & 0x3FFFFFFFu) | (b ^ 3u);
        };
#pragma unroll
        for (unsigned s = 0; s < MMER - 1; ++s) roll();
        // Sliding-window minimum over WIN = 46 keys, blocks of 16 m-mers: at m-mer 16c + i the window of k-mer 16c + i - 45 is the suffix of
        // block c-3 from i+3, the blocks c-2 and c-1 and the prefix of block c (i <= 12), or the suffix of block c-2 from i-13, block c-1
        // and the prefix of block c (i >= 13): one v_min3 per position.  X3, X2, X1: suffix minima of the three blocks before the current one.
        uint32_t X3[16], X2[16], X1[16], C[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) { X3[t] = INF; X2[t] = INF; X1[t] = INF; C[t] = INF; }
        uint32_t cur = 0;
        unsigned p0 = 0, jrec = 0;
        // one block of 16 m-mers from m-mer jb on (base jb + 14 + i is taken at step i: the refill of sixteen bases falls on step 2 of every
        // block).  FIRST: its first step that has a k-mer position -- blocks 0 and 1 have none, block 2 has k-mer 0 at step 13.  Steps past
        // the longest read of the wave do nothing (no key, no k-mer), so the block needs no guard -- a guard makes every array a phi of
        // the branch and costs a register move per element and step.
        auto block = [&](auto first, unsigned jb) {
            constexpr int FIRST = decltype(first)::value;
            // stage 1, branch-free: the sixteen keys of the block and the buckets of its k-mer positions -- sixteen independent chains
            // (three multiplications each) in one basic block, so that a wavefront alone keeps its SIMD issuing
            uint32_t pref = INF;
            const uint32_t w21 = min(X2[0], X1[0]);
            uint32_t bk[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned s = jb + (MMER - 1) + i;      // base taken in this step; m-mer s-14 is complete, k-mer s-59 gets its window
                if (i == 2) { W = __funnelshift_r(dA, dB, sb8); dA = dB; dB = q[min(un, ulast)]; ++un; }
                roll();
                const uint32_t key = s < gl ? mmer_key(f < rc ? f : rc) : INF;
                C[i] = key; pref = min(pref, key);
                bk[i] = 0;
                if (i >= FIRST) {
                    const uint32_t mk = i <= 12 ? min(min(X3[i <= 12 ? i + 3 : 0], w21), pref) : min(min(X2[i >= 13 ? i - 13 : 0], X1[0]), pref);
                    bk[i] = bucket_of(mk, nb);
                }
            }
            // stage 2: where runs end (the few lanes it concerns; nearly every step has some)
#pragma unroll
            for (int i = FIRST; i < 16; ++i) {
                const unsigned p = jb + i - (WIN - 1);       // (= s - 59)
                const bool isk = p < nk;
                const uint32_t bkt = bk[i];
                // a run ends where the bucket changes and when the record is full (ALIGN64: at every multiple of 64 instead, the cuts
                // of the wavefront-per-read kernel -- 17 % more records; the parity test of the two kernels asks for it)
                const bool brk = isk && (!p || bkt != cur || (ALIGN64 ? !(p & 63u) || (REC_MAXK < 64 && p - p0 == REC_MAXK) : p - p0 == REC_MAXK));
                if ((brk && p) || (p == nk && nk)) {                                  // the run [p0, p) ends here
                    const uint32_t rb = cur - pb_lo;
                    if (rb < pb_hi - pb_lo) {                // multi-pass counting: other ranges' records are cut again in their own pass
                        const uint32_t meta = p0 | ((p - p0 - 1) << 16) | (p0 ? 1u << 22 : 0u) | (isk ? 1u << 23 : 0u);
                        if (jrec < S) stage[lane * S + jrec] = make_uint2(rb, meta);
                        else {                               // more records than slots: the overflow list, with its rank
                            const uint32_t rk = atomicAdd(&bcount[rb], 1u);
                            if (part_kmers) {
                                uint32_t pt = nbl_part > 1 ? __umulhi(rb, inv_nbl) : rb;
                                if ((pt + 1) * nbl_part <= rb) ++pt;
                                atomicAdd(&spart_[wv_][pt & 63], p - p0);
                            }
                            const unsigned long long o = atomicAdd(ov_cursor, 1ull);
                            if (o < ov_cap) { o_read[o] = (uint32_t)r; o_bkt[o] = rb; o_meta[o] = meta; o_rank[o] = rk; }
                        }
                        ++jrec;
                    }
                }
                if (brk) { p0 = p; cur = bkt; }
            }
#pragma unroll
            for (int t = 14; t >= 0; --t) C[t] = min(C[t], C[t + 1]);                     // suffix minima of the block just filled
#pragma unroll
            for (int t = 0; t < 16; ++t) { X3[t] = X2[t]; X2[t] = X1[t]; X1[t] = C[t]; }
        };
        block(std::integral_constant<int, 16>{}, 0);
        block(std::integral_constant<int, 16>{}, 16);
        block(std::integral_constant<int, 13>{}, 32);
        for (unsigned jb = 48; jb + (MMER - 1) <= maxgl; jb += 16) block(std::integral_constant<int, 0>{}, jb);
    }
    wave_lds_fence();
    // ---- staged descriptors -> final ones, 64 at a time in slot order (conflict-free LDS reads, coalesced stores): the histogram atomic
    //      RETURNS the record's rank inside its bucket (16 bits in the descriptor; beyond that the overflow list); eight atomics are in
    //      flight per lane before the first result is used
    const uint64_t nflat = (uint64_t)((n - r0 < 64 ? n - r0 : 64)) * S;
    for (unsigned u0 = 0; u0 < S; u0 += 8) {
        uint2 d[8]; uint32_t rk[8];
#pragma unroll
        for (unsigned k = 0; k < 8; ++k) {
            const unsigned i = lane + 64 * (u0 + k);
            d[k] = u0 + k < S ? stage[i] : make_uint2(0u, NONE32);
            rk[k] = 0;
            if (d[k].y != NONE32) {
                rk[k] = atomicAdd(&bcount[d[k].x], 1u);
                if (part_kmers) {                            // bucket / nbl_part by reciprocal (+1 correction)
                    uint32_t pt = nbl_part > 1 ? __umulhi(d[k].x, inv_nbl) : d[k].x;
                    if ((pt + 1) * nbl_part <= d[k].x) ++pt;
                    atomicAdd(&spart_[wv_][pt & 63], ((d[k].y >> 16) & 63u) + 1u);
                }
            }
        }
#pragma unroll
        for (unsigned k = 0; k < 8; ++k) {
            const unsigned i = lane + 64 * (u0 + k);
            if (u0 + k < S && i < nflat) {
                uint2 out = make_uint2(0u, NONE32);
                if (d[k].y != NONE32) {
                    if (rk[k] < 65536u) out = make_uint2(d[k].x | (rk[k] << 24), d[k].y | ((rk[k] >> 8) << 24));
                    else {
                        const unsigned long long o = atomicAdd(ov_cursor, 1ull);
                        if (o < ov_cap) { o_read[o] = (uint32_t)(r0 + i / S); o_bkt[o] = d[k].x; o_meta[o] = d[k].y; o_rank[o] = rk[k]; }
                    }
                }
                s_desc[r0 * S + i] = out;
            }
        }
    }
    if (part_kmers) {
        wave_lds_fence();
        const uint32_t v = spart_[wv_][lane];
        if (v) atomicAdd(&part_kmers[(blockIdx.x & 63u) * 64u + lane], (unsigned long long)v);
    }
}

// =============================================================================== K2
// One thread per descriptor slot (then per overflow entry): the record's place is its bucket's base + the rank K1's
// histogram atomic returned; cut the 2*(nk+61) stream bits [left flank][k-mers' bases][right flank] out of the read
// (unaligned 8-byte loads), store the 32-B record.  No atomics.
__global__ void __launch_bounds__(256) k_scatter_records(uint64_t nslots, uint32_t slots_per_read, const uint2* __restrict__ s_desc,
                                                          uint64_t nov, const uint32_t* __restrict__ o_read,
                                                          const uint32_t* __restrict__ o_bkt, const uint32_t* __restrict__ o_meta,
                                                          const uint32_t* __restrict__ o_rank,
                                                          const uint8_t* __restrict__ bases,
                                                          const uint64_t* __restrict__ boff, uint64_t bases_bytes,
                                                          const uint64_t* __restrict__ bbase,
                                                          uint32_t* __restrict__ recs) {
    // A wavefront's 64 records leave through LDS: built one per lane, stored REC_DWORDS lanes per record, so that every store
    // instruction carries eight whole 32-B records (seven of 36 B in a W2RAP_REC36 build) as contiguous bursts instead of 64 scattered dwords.
    __shared__ __attribute__((aligned(16))) uint32_t s_rec[4][64 * REC_DWORDS];
    __shared__ uint64_t s_dst[4][64];
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t r = 0, b = 0, meta = NONE32, slot = 0;
    if (i < nslots) {
        const uint2 d = s_desc[i];
        if (d.y != NONE32) { meta = d.y & 0xFFFFFFu; b = d.x & 0xFFFFFFu; slot = (d.x >> 24) | ((d.y >> 24) << 8); }
        r = (uint32_t)(i / slots_per_read);
    } else if (i - nslots < nov) { r = o_read[i - nslots]; b = o_bkt[i - nslots]; meta = o_meta[i - nslots]; slot = o_rank[i - nslots]; }
    const bool valid = meta != NONE32;
    uint64_t dst = ~0ull;
    uint32_t out[REC_DWORDS] = {};
    if (valid) {
        const uint64_t base = bbase[b];
        const unsigned p = meta & 0xFFFFu, nk = ((meta >> 16) & 63u) + 1u;
        const bool hasL = (meta >> 22) & 1u, hasR = (meta >> 23) & 1u;
        const uint64_t ro = boff[r];
        const unsigned q = p ? p - 1 : 0;                          // first base taken from the read
        const uint64_t byte0 = ro + ((2 * q) >> 3);
        const unsigned sh = (2 * q) & 7;
        uint64_t W[5];
#pragma unroll
        for (unsigned j = 0; j < 5; ++j) {
            const uint64_t a = byte0 + 8 * j;
            uint64_t w = 0;
            if (a + 8 <= bases_bytes) w = reinterpret_cast<const U64u*>(bases + a)->v;
            else for (unsigned t = 0; t < 8; ++t) if (a + t < bases_bytes) w |= (uint64_t)bases[a + t] << (8 * t);
            W[j] = w;
        }
        uint64_t O[4];
#pragma unroll
        for (unsigned j = 0; j < 4; ++j) O[j] = sh ? (W[j] >> sh) | (W[j + 1] << (64 - sh)) : W[j];
        if (!p) {                                                  // no base before base 0: a zero left flank
            O[3] = (O[3] << 2) | (O[2] >> 62); O[2] = (O[2] << 2) | (O[1] >> 62); O[1] = (O[1] << 2) | (O[0] >> 62); O[0] <<= 2;
        }
        unsigned nbits = 2 * (nk + 61);
        if (!hasR) nbits -= 2;                                     // the right flank is not part of the k-mer run
#pragma unroll
        for (unsigned j = 0; j < 4; ++j) {
            if (nbits <= 64 * j) O[j] = 0;
            else if (nbits < 64 * (j + 1)) O[j] &= (1ull << (nbits - 64 * j)) - 1;
        }
        const uint32_t hdr = (nk - 1) | (hasL ? 64u : 0u) | (hasR ? 128u : 0u);
        if constexpr (REC_HB == 8) {                               // 256 bits: the header byte, the stream behind it
            O[3] = (O[3] << 8) | (O[2] >> 56); O[2] = (O[2] << 8) | (O[1] >> 56); O[1] = (O[1] << 8) | (O[0] >> 56);
            O[0] = (O[0] << 8) | (uint64_t)hdr;
#pragma unroll
            for (unsigned j = 0; j < 4; ++j) { out[2 * j] = (uint32_t)O[j]; out[2 * j + 1] = (uint32_t)(O[j] >> 32); }
        } else {
            out[0] = hdr;
#pragma unroll
            for (unsigned j = 0; j < 4; ++j) { out[(1 + 2 * j) % REC_DWORDS] = (uint32_t)O[j]; out[(2 + 2 * j) % REC_DWORDS] = (uint32_t)(O[j] >> 32); }
        }
        dst = (base + slot) * REC_DWORDS;
    }
    if constexpr (REC_DWORDS == 8) {
        reinterpret_cast<uint4*>(s_rec[wv])[2 * lane] = make_uint4(out[0], out[1], out[2], out[3]);
        reinterpret_cast<uint4*>(s_rec[wv])[2 * lane + 1] = make_uint4(out[4], out[5], out[6], out[7]);
    } else {
#pragma unroll
        for (unsigned j = 0; j < REC_DWORDS; ++j) s_rec[wv][lane * REC_DWORDS + j] = out[j];
    }
    s_dst[wv][lane] = dst;
    wave_lds_fence();
    if constexpr (REC_DWORDS == 8) {                               // 32-B records, 32-B aligned: two lanes per record, 16 B each
#pragma unroll
        for (unsigned j = 0; j < 2; ++j) {
            const unsigned idx = j * 64 + lane, rec = idx >> 1;
            const uint64_t d = s_dst[wv][rec];
            if (d != ~0ull) reinterpret_cast<uint4*>(recs + d)[idx & 1u] = reinterpret_cast<const uint4*>(s_rec[wv])[idx];
        }
    } else {
#pragma unroll
        for (unsigned j = 0; j < REC_DWORDS; ++j) {
            const unsigned idx = j * 64 + lane, rec = idx / REC_DWORDS, qd = idx - rec * REC_DWORDS;
            const uint64_t d = s_dst[wv][rec];
            if (d != ~0ull) recs[d + qd] = s_rec[wv][idx];
        }
    }
}

// =============================================================================== K3
// Per-bucket LDS hash table: 16-B keys (hi, lo; hi == ~0 <=> empty, lo == ~0 <=> key being written)
// + one dword (bits 23:0 occurrence count, bits 31:24 OR of contexts) per slot.
//
// The kernel is a persistent, software-pipelined loop over buckets; nothing on the per-bucket
// critical path waits for HBM:
//  * bucket ids come from an atomic queue three buckets ahead, the record ranges of bucket i+2 and
//    the first record tile of bucket i+1 are loaded while bucket i is counted;
//  * a bucket's records (from all `nseg` source segments, as one logical dword stream) go through
//    LDS in tiles of TILE records with coalesced dword loads;
//  * inside a tile the k-mers of a wavefront's records are flattened (lane = k-mer): the record of
//    every lane comes from a per-wave bit vector of record starts (one uniform 64-bit word per
//    64-k-mer window, v_readlane + popcount; no search), the 124 stream bits of the k-mer and its
//    two flank bases are five LDS dwords, the reverse complement is the complement of the two
//    LSB-first halves swapped, and one probe is {key, count} read together followed by
//    fire-and-forget ds_add / ds_or;
//  * emit compacts the solid entries into an LDS staging area, reserves their output range with
//    ONE global atomic whose result is only consumed while the next bucket is emitted, and resets
//    exactly the slots that were occupied, so the table never needs a clearing pass.
// A bucket whose distinct set overflows the table is split by hash bits and recounted
// (= MapReduceEngine.h:288-291).
template <unsigned CAP, unsigned THREADS, bool PARK = true>
struct K3Cfg {
    static constexpr unsigned NW = THREADS / 64;
    static constexpr unsigned RPL = !PARK ? 16 : NW >= 16 ? 32 : 64;   // records per wave per tile (one per lane 0..RPL-1)
    static constexpr unsigned TILE = NW * RPL;                // records per tile
    static constexpr unsigned NPF = (TILE * REC_DWORDS + THREADS - 1) / THREADS;
    static constexpr unsigned SC = CAP / 8;                   // staging entries
    static constexpr unsigned QCAP = PARK ? 128 : 0;          // parked k-mers per wave (PARK = false: collisions are probed at once, no queue)
    static constexpr unsigned MAXSEG = 64;
    static constexpr unsigned LIMIT = CAP - THREADS - 8;
    static constexpr unsigned PER = CAP / THREADS;
    static constexpr unsigned LDS = CAP * 16 + SC * 16 + 3 * MAXSEG * 8 + NW * QCAP * 16 +
                                    (CAP + SC + TILE * REC_DWORDS + 4 + 3 * TILE + NW + NW * QCAP + 3 * (MAXSEG + 1) + 4 + 104 + 16 + 40 + 2 + 128) * 4;
};
enum { K3_FILL = 0, K3_OVF, K3_DEPTH, K3_CNT, K3_NPREV, K3_BASELO, K3_BASEHI, K3_B2LO, K3_B2HI, K3_WIN = 11 /* [10]: PROF */ };

__device__ inline uint32_t ld32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void st32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline uint64_t ld64(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void st64(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// inclusive prefix sum over the lanes of each 16-lane row (DPP row_shr, no LDS round trips) ...
__device__ inline uint32_t row_scan16(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
    return x;
}
// ... and over lanes 0..31 (row 0's total is added to row 1: row_bcast15 into rows 1 and 3)
__device__ inline uint32_t half_scan32(uint32_t x) {
    x = row_scan16(x);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
    return x;
}
// ... and over the whole wavefront (row_bcast31: lane 31's total into rows 2 and 3)
__device__ inline uint32_t wave_scan64(uint32_t x) {
    x = half_scan32(x);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
    return x;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// 16-B LDS read that the compiler neither splits nor caches (the tag groups are polled)
__device__ inline u32x4 lds_read_b128(const uint32_t* p) {
    u32x4 v;
    const uint32_t a = (uint32_t)(uintptr_t)p;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
    return v;
}

// (PARK = false, MINW = 8: the round-3 experiment -- 2048 slots, 256-record tiles, no parking queue: 75 KB of LDS and <= 64 VGPRs, so that
//  TWO 1024-thread blocks share a CU, eight waves per SIMD instead of four; buckets of half the size)
template <unsigned CAP, unsigned THREADS, bool PROF = false, bool PARK = true, unsigned MINW = 1>
__global__ void __launch_bounds__(THREADS, MINW) k_count_buckets(uint32_t nb, uint32_t b_lo, uint32_t b_hi /* this launch counts buckets [b_lo, b_hi) of nb */,
                                                            uint32_t nseg, const uint64_t* __restrict__ roff,
                                                            const uint32_t* __restrict__ recs, uint32_t min_freq,
                                                            uint32_t* __restrict__ queue,
                                                            uint64_t* __restrict__ shi, uint64_t* __restrict__ slo,
                                                            uint32_t* __restrict__ scc, uint64_t solid_cap,
                                                            unsigned long long* __restrict__ counters /*0 solid | emits << 40,1 distinct,2 overflow passes,3 error*/,
                                                            unsigned long long* __restrict__ ghist,
                                                            uint64_t* __restrict__ chunk_start, uint32_t* __restrict__ chunk_cnt, uint32_t chunk_cap,
                                                            const uint32_t* __restrict__ blist /* null, or [0] = number of listed buckets, [2..] their ids:
                                                            the buckets k_count_fp deferred; `queue` then deals out list positions */, uint32_t blist_cap) {
    // Every emit reserves its output range AND a chunk number with one 64-bit atomic (count in bits 39:0, chunks above):
    // the solid k-mers of one bucket (class) lie contiguously, and the list of those chunks lets the adjacency prune work
    // bucket by bucket in LDS (k_prune_local) instead of probing the dictionary in HBM for every neighbour.
    constexpr unsigned long long SMASK = (1ull << 40) - 1;
    using C = K3Cfg<CAP, THREADS, PARK>;
    constexpr unsigned NW = C::NW, RPL = C::RPL, TILE = C::TILE, NPF = C::NPF, SC = C::SC, MAXSEG = C::MAXSEG, PER = C::PER, QCAP = C::QCAP;
    constexpr unsigned LOG_CAP = CAP == 4096 ? 12 : CAP == 2048 ? 11 : CAP == 1024 ? 10 : 13;
    static_assert((1u << LOG_CAP) == CAP, "CAP");
    constexpr uint64_t EMPTY = ~0ull;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);            // [CAP][2]: hi, lo
    uint64_t* sthi = keys + 2 * CAP;                               // staging
    uint64_t* stlo = sthi + SC;
    uint64_t* segbase = stlo + SC;                                 // [3][MAXSEG] dword index of a segment's share minus its logical start
    uint64_t* qhi = segbase + 3 * MAXSEG + (threadIdx.x >> 6) * QCAP;   // this wave's queue of parked k-mers
    uint64_t* qlo = qhi + NW * QCAP;
    uint32_t* cc = reinterpret_cast<uint32_t*>(segbase + 3 * MAXSEG + 2 * NW * QCAP);
    uint32_t* stcc = cc + CAP;
    uint32_t* tile = stcc + SC;                                    // TILE*9 (+4 pad)
    uint32_t* bv32 = tile + TILE * REC_DWORDS + 4;                 // [2*TILE] record-start bit vector of the tile's flattened k-mers
    uint32_t* Bw = bv32 + 2 * TILE;                                // [TILE] record covering the first position of each window
    uint32_t* wtot = Bw + TILE;                                    // [NW] k-mers per wave's records
    uint32_t* qmeta = wtot + NW + (threadIdx.x >> 6) * QCAP;       // parked k-mers: next slot to look at | ctx << 16
    uint32_t* segdpre = wtot + NW + NW * QCAP;                     // [3][MAXSEG+1] logical dword prefix of the segments
    uint32_t* bq = segdpre + 3 * (MAXSEG + 1);                     // ring of bucket ids
    uint32_t* lhist = bq + 4;                                      // 104
    uint32_t* misc = lhist + 104;                                  // 16
    uint32_t* stk = misc + 16;                                     // (class, P) pairs, depth <= 18
    uint64_t* dummy64 = reinterpret_cast<uint64_t*>((reinterpret_cast<uintptr_t>(stk + 40) + 7) & ~uintptr_t(7));   // [64] sink of the lanes that did not claim a slot
    const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the bucket behind ticket t of the queue: the t-th bucket of the range, or the t-th listed one
    const uint32_t nlist = blist ? (blist[0] < blist_cap ? blist[0] : blist_cap) : 0u;
    auto ticket = [&](uint32_t t) -> uint32_t { return blist ? (t < nlist ? blist[2 + t] : NONE32) : (t < b_hi - b_lo ? b_lo + t : NONE32); };

    // ---- segment table of bucket bb -> registers (wave 0, lane = segment), and from registers -> LDS ring slot
    auto seg_load = [&](uint32_t bb, uint64_t& r0, uint32_t& cnt) {
        r0 = 0; cnt = 0;
        if (tid < nseg && bb < b_hi) {
            const uint64_t a = roff[(uint64_t)tid * nb + bb], e = roff[(uint64_t)tid * nb + bb + 1];
            r0 = a; cnt = (uint32_t)(e - a);
            if (e - a >= (1ull << 26)) { cnt = 0; counters[3] = 2; }     // keeps 24-bit counts and 32-bit dword indices exact
        }
    };
    auto seg_store = [&](unsigned q, uint64_t r0, uint32_t cnt) {       // wave 0 only
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
        uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (tot >= (1u << 26)) { tot = 0; incl = 0; cnt = 0; if (lane == 0) counters[3] = 2; }
        const uint32_t ex = incl - cnt;
        if (lane < nseg) { segdpre[q * (MAXSEG + 1) + lane] = ex * REC_DWORDS; segbase[q * MAXSEG + lane] = (r0 - ex) * REC_DWORDS; }
        if (lane == 0) segdpre[q * (MAXSEG + 1) + nseg] = tot * REC_DWORDS;
    };
    // ---- tile t of the bucket in ring slot q -> registers (coalesced dwords of the logical record stream)
    auto tile_load = [&](unsigned q, uint32_t t, uint32_t (&v)[NPF]) {
        const uint32_t* dp = segdpre + q * (MAXSEG + 1);
        const uint32_t dall = dp[nseg];
        const uint32_t d0 = t * (TILE * REC_DWORDS);
        const uint32_t dend = dall - d0 < TILE * REC_DWORDS ? dall : d0 + TILE * REC_DWORDS;
        // the segment of a dword = the number of segment boundaries at or below it: up to 8 segments the boundaries are read
        // once (independent LDS loads) and compared in registers instead of walked load by load
        constexpr unsigned NB_REG = 7;
        uint32_t bnd[NB_REG];
#pragma unroll
        for (unsigned i = 0; i < NB_REG; ++i) bnd[i] = (i + 1 < nseg && nseg <= NB_REG + 1) ? dp[i + 1] : 0xFFFFFFFFu;
#pragma unroll
        for (unsigned j = 0; j < NPF; ++j) {
            const uint32_t d = d0 + j * THREADS + tid;
            uint32_t x = 0;
            if (d0 < dall && d < dend) {
                unsigned s = 0;
                if (nseg <= NB_REG + 1) {
#pragma unroll
                    for (unsigned i = 0; i < NB_REG; ++i) s += d >= bnd[i] ? 1u : 0u;
                } else while (d >= dp[s + 1]) ++s;
                x = recs[segbase[q * MAXSEG + s] + d];
            }
            v[j] = x;
        }
    };
    auto tile_store = [&](const uint32_t (&v)[NPF]) {
#pragma unroll
        for (unsigned j = 0; j < NPF; ++j) { const unsigned i = j * THREADS + tid; if (i < TILE * REC_DWORDS) tile[i] = v[j]; }
    };

    // ---- init: empty table, first three bucket ids, segment tables of the first two, first tile
    for (unsigned i = tid; i < CAP; i += THREADS) { keys[2 * i] = EMPTY; keys[2 * i + 1] = EMPTY; cc[i] = 0; }
    for (unsigned i = tid; i < 104; i += THREADS) lhist[i] = 0;
    if (tid < 4) tile[TILE * REC_DWORDS + tid] = 0;
    if (tid < 16) misc[tid] = 0;
    if (tid == 0) { const uint32_t t0 = atomicAdd(queue, 3u); bq[0] = ticket(t0); bq[1] = ticket(t0 + 1); bq[2] = ticket(t0 + 2); bq[3] = NONE32; }
    __syncthreads();
    if (wv == 0) {
        uint64_t r0; uint32_t cnt;
        seg_load(bq[0], r0, cnt); seg_store(0, r0, cnt);
        seg_load(bq[1], r0, cnt); seg_store(1, r0, cnt);
    }
    __syncthreads();
    uint32_t pf[NPF];
    tile_load(0, 0, pf);
    unsigned long long pend_base = 0, my_distinct = 0;
    // PROF: shader-clock time of wave 0 per phase (stage-in, count, barrier A, flush+scan, barrier B, staging), summed into counters[106..]
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0}, tp = 0, ptmax = 0, wt[5] = {0, 0, 0, 0, 0}, wtp = 0, wcount = 0;
    auto wtick = [&](int ph) { if (PROF && wv == 0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); if (ph >= 0) wt[ph] += now - wtp; wtp = now; } };
    auto tick = [&](int ph) { if (PROF) { const unsigned long long now = __builtin_amdgcn_s_memtime(); if (ph >= 0) pt[ph] += now - tp; tp = now; } };

    // ---- the full probe sequence for the lanes with `live`: find the key or claim a free slot from slot s on; count + context.
    //      Returns whether this lane inserted a NEW key.
    unsigned qn = 0;
    bool big = false;
    auto probe = [&](bool live, Kmer k, unsigned s, unsigned ctx) -> bool {
        bool isnew = false;
        if (live) {
            bool ok = false;
            int budget = 2 * (int)CAP;
            for (;;) {
                bool emp;
                for (;;) {                                           // search only: tight single-exit loop
                    const uint64_t khi = ld64(&keys[2 * s]), klo = ld64(&keys[2 * s + 1]);
                    const bool same_hi = khi == k.hi;
                    emp = khi == EMPTY;
                    const bool hit = same_hi & (klo == k.lo), busy = same_hi & (klo == EMPTY);
                    --budget;
                    if (hit | emp | (budget <= 0)) break;
                    s = busy ? s : ((s + 1) & (CAP - 1));
                }
                if (budget <= 0) break;                              // table full
                if (!emp) { ok = true; break; }
                const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[2 * s]), (unsigned long long)EMPTY,
                                               (unsigned long long)k.hi);
                if (old == EMPTY) { st64(&keys[2 * s + 1], k.lo); isnew = true; ok = true; break; }
            }
            if (ok) {
                if (!big || (ld32(&cc[s]) & 0xFFFFFFu) < 0xFFF000u) atomicAdd(&cc[s], 1u);
                atomicOr(&cc[s], ctx << 24);
            } else st32(&misc[K3_OVF], 1u);                          // table full: recount in two classes
        }
        return isnew;
    };
    // ---- finish the top `cnt` (<= 64) parked k-mers of this wave; returns the new keys
    auto drain = [&](unsigned cnt) -> unsigned {
        wave_lds_fence();
        bool isnew = false;
        if (PARK) {
            const bool live = lane < cnt;
            const unsigned e = live ? qn - cnt + lane : 0u;
            Kmer k{0, 0}; uint32_t meta = 0;
            if (live) { k = Kmer{qhi[e], qlo[e]}; meta = qmeta[e]; }
            isnew = probe(live, k, meta & 0xFFFFu, meta >> 16);
        }
        qn -= cnt;
        wave_lds_fence();
        return (unsigned)__builtin_popcountll(__ballot(isnew));
    };

    for (uint32_t it = 0;; ++it) {
        const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld32(&bq[it & 3]));       // block-uniform values are kept in SGPRs: scalar branches
        if (b >= b_hi) break;
        const unsigned q = it % 3;
        tick(-1);
        const uint32_t nrec = (uint32_t)__builtin_amdgcn_readfirstlane((int)segdpre[q * (MAXSEG + 1) + nseg]) / REC_DWORDS;
        const uint32_t ntiles = (nrec + TILE - 1) / TILE;
        big = nrec >= (1u << 18);
        // ---- stage in this bucket's first tile; start the look-ahead loads (consumed before barrier A)
        tile_store(pf);
        uint32_t la_b = 0;
        if (tid == 0) la_b = ticket(atomicAdd(queue, 1u));
        uint64_t la_r0 = 0; uint32_t la_cnt = 0;
        if (wv == 0) seg_load(ld32(&bq[(it + 2) & 3]), la_r0, la_cnt);
        tile_load((it + 1) % 3, 0, pf);
        if (tid == 0) { stk[0] = 0; stk[1] = 1; misc[K3_DEPTH] = 1; }
        bool first_pass = true;
        __syncthreads();                                             // S1
        tick(0);
        unsigned long long tp0 = 0; if (PROF) tp0 = __builtin_amdgcn_s_memtime();
        for (;;) {                                                   // (class, P) work stack; normally one pass
            const unsigned sp = ld32(&misc[K3_DEPTH]) - 1;
            const uint32_t cls = stk[2 * sp], P = stk[2 * sp + 1];
            for (uint32_t t = 0; t < ntiles; ++t) {
                if (t > 0 || (!first_pass && ntiles > 1)) {          // rare: bucket longer than one tile / recount
                    uint32_t tmp[NPF];
                    tile_load(q, t, tmp);
                    __syncthreads();
                    tile_store(tmp);
                    __syncthreads();
                }
                const unsigned nrec_tile = nrec - t * TILE < TILE ? nrec - t * TILE : TILE;
                // Flatten the tile's k-mers: record G = wv*32 + l (scanned by lane l of wave wv) occupies positions
                // [e, e+nk) of the tile-wide k-mer numbering.  bv = one bit per
                // position where a record starts; Bw[w] = (G << 16 | e) of the record covering position 64*w.
                // Window w (64 consecutive positions) is counted by wave w % NW, so every wave gets the same
                // number of k-mers whatever the records' lengths are.
                const unsigned myrec = wv * RPL + lane;
                const unsigned nk = (lane < RPL && myrec < nrec_tile) ? (tile[myrec * REC_DWORDS] & 63u) + 1u : 0u;
                const unsigned incl = RPL == 32 ? half_scan32(nk) : wave_scan64(nk);
                if (lane == RPL - 1) wtot[wv] = incl;
                for (unsigned i = tid; i < 2 * TILE; i += THREADS) bv32[i] = 0;
                if (tid == 0) misc[K3_WIN] = 2 * NW;                 // windows 0 .. 2 NW - 1 are dealt out statically, the rest on demand
                __syncthreads();                                     // X1
                const unsigned wsum = lane < NW ? wtot[lane] : 0u, wsc = row_scan16(wsum);
                const int wvu = __builtin_amdgcn_readfirstlane((int)wv);
                const unsigned base = (unsigned)__builtin_amdgcn_readlane((int)wsc, wvu) - (unsigned)__builtin_amdgcn_readlane((int)wsum, wvu);
                const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)wsc, NW - 1);
                if (nk) {
                    const unsigned e = base + incl - nk;
                    atomicOr(&bv32[e >> 5], 1u << (e & 31));
                    const unsigned w1 = (e + 63) >> 6;
                    if (64 * w1 < e + nk) Bw[w1] = (myrec << 16) | e;
                }
                __syncthreads();                                     // X2
                // ---- the windows of this wave, as three stages:
                //   A(w): locate every lane's k-mer (record, index) and fetch its 6 stream dwords;
                //   B(w): cut out the k-mer, canonicalise, hash, fetch the key at its home slot;
                //   C(w): hit -> count / free -> claim / anything else -> park.
                // (Interleaving the stages of three consecutive windows inside one wave -- B(w+1), A(w+2), C(w) -- was measured:
                // 165 VGPRs at 512 threads, 65 ms against 45 ms for this form with 16 waves per CU.)
                // The loop body is STRAIGHT-LINE, predicated code.  A wave issues one instruction per ~5 clocks whatever its
                // kind, and a divergent `if` (v_cmp -> s_and_saveexec -> ... -> s_or exec) costs ~37 clocks against ~18 for
                // v_cmp + v_cndmask (tools/issue_ubench.hip), so: the window counter and everything derived from it are
                // SCALAR (readfirstlane), LDS reads of lanes without a k-mer are clamped instead of masked, the claim is an
                // unconditional CAS (a lane that does not want the slot swaps EMPTY for EMPTY), lanes that did not win
                // store their low word into a per-lane dummy, count/context go through `add 0` / `or 0`, and every lane
                // writes one queue entry (parked lanes first, the others behind the new top).
                uint32_t fill_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld32(&misc[K3_FILL])),
                         ovf_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld32(&misc[K3_OVF]));
                const unsigned nwin = (total + 63) / 64;
                const unsigned wv_s = (unsigned)wvu;                 // this wave's number as a scalar
                uint32_t nM0 = 0, nM1 = 0, nB = 0;                   // start bits / covering record of the window handled next
                { const unsigned w0 = wv_s < nwin ? wv_s : 0u; nM0 = bv32[2 * w0]; nM1 = bv32[2 * w0 + 1]; nB = Bw[w0]; }
                const uint64_t lane_le = ~0ull >> (63 - lane);
                const unsigned dummy_at = (unsigned)(dummy64 - keys);
                const uint32_t lane_lt_lo = lane < 32 ? (1u << lane) - 1u : 0xFFFFFFFFu, lane_lt_hi = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;
                // Windows are dealt out ON DEMAND: the hardware issues from the oldest wave first, so with a fixed share per wave
                // the waves of a block finish 30 % apart and everyone waits at barrier A for the youngest (measured: 52 M clocks
                // for waves 0-3, 72 M for waves 12-15).  Each wave starts with windows wv and wv + NW and draws the index of the
                // window after next from an LDS counter while it works on the current one.
                unsigned wnext = wv_s + NW;
                for (unsigned w = wv_s; w < nwin;) {
                    // the (one window stale) overflow checks, on scalars
                    if (ovf_seen) break;
                    if (fill_seen >= C::LIMIT) { if (lane == 0) st32(&misc[K3_OVF], 1u); break; }
                    const uint32_t fill_ld = ld32(&misc[K3_FILL]), ovf_ld = ld32(&misc[K3_OVF]);
                    uint32_t wdraw = 0;
                    if (lane == 0) wdraw = atomicAdd(&misc[K3_WIN], 1u);
                    wtick(-1);
                    // ---- A: locate every lane's k-mer (record, index) and fetch its 6 stream dwords
                    const uint32_t M0 = nM0, M1 = nM1, Bv = nB;
                    { const unsigned wn = wnext < nwin ? wnext : w; nM0 = bv32[2 * wn]; nM1 = bv32[2 * wn + 1]; nB = Bw[wn]; }
                    const unsigned g = w * 64 + lane;
                    bool active = g < total;
                    const uint32_t mle0 = M0 & (uint32_t)lane_le, mle1 = M1 & (uint32_t)(lane_le >> 32);
                    const unsigned c = (unsigned)__builtin_popcount(mle0 & ~1u) + (unsigned)__builtin_popcount(mle1);
                    // highest record start at or below this lane (bit 0 forced: defined for c == 0, where it is not used)
                    const unsigned top = mle1 ? 63u - (unsigned)__builtin_clz(mle1) : 31u - (unsigned)__builtin_clz(mle0 | 1u);
                    const unsigned idx = lane - (c ? top : (Bv & 0xFFFFu) - w * 64);
                    const unsigned rec = active ? (Bv >> 16) + c : 0u;
                    const uint32_t* wp = tile + rec * REC_DWORDS;
                    const unsigned q0 = active ? (idx + REC_HB / 2) >> 4 : 0u;      // the stream starts at bit REC_HB of the record: base t at bit REC_HB + 2 t
                    const uint32_t hdr = wp[0], d0 = wp[q0], d1 = wp[1 + q0], d2 = wp[2 + q0], d3 = wp[3 + q0], d4 = wp[4 + q0];
                    if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); wtick(0); }
                    // ---- B: cut out the k-mer, canonicalise, hash, fetch the key at its home slot
                    const unsigned sh = ((idx + REC_HB / 2) & 15u) * 2u;
                    // 128 stream bits from base idx: 1:0 left flank, 2..121 the k-mer, 123:122 right flank (32-bit ops only)
                    const uint32_t e0 = __funnelshift_r(d0, d1, sh), e1 = __funnelshift_r(d1, d2, sh),
                                   e2 = __funnelshift_r(d2, d3, sh), e3 = __funnelshift_r(d3, d4, sh);
                    const uint32_t s0l = __funnelshift_r(e0, e1, 2), s0h = (e1 >> 2) & 0x0FFFFFFFu;       // bases 0..29, LSB first
                    const uint32_t s1l = __funnelshift_r(e1, e2, 30), s1h = __funnelshift_r(e2, e3, 30) & 0x0FFFFFFFu;
                    auto rev2_32 = [](uint32_t x) { x = __brev(x); return ((x & 0x55555555u) << 1) | ((x >> 1) & 0x55555555u); };
                    // MSB-first words: reverse the 30 groups of each half
                    const uint32_t a0 = rev2_32(s0l), b0 = rev2_32(s0h), a1 = rev2_32(s1l), b1 = rev2_32(s1h);
                    const uint32_t khh = a0 >> 4, khl = __funnelshift_r(b0, a0, 4), klh = a1 >> 4, kll = __funnelshift_r(b1, a1, 4);
                    // reverse complement = complemented LSB-first halves, swapped
                    const uint32_t rhh = ~s1h & 0x0FFFFFFFu, rhl = ~s1l, rlh = ~s0h & 0x0FFFFFFFu, rll = ~s0l;
                    const uint64_t khi_f = ((uint64_t)khh << 32) | khl, klo_f = ((uint64_t)klh << 32) | kll;
                    const uint64_t khi_r = ((uint64_t)rhh << 32) | rhl, klo_r = ((uint64_t)rlh << 32) | rll;
                    const bool rc = khi_r != khi_f ? khi_r < khi_f : klo_r < klo_f;
                    const Kmer k{rc ? khi_r : khi_f, rc ? klo_r : klo_f};
                    // context (KMerContext: bits 0..3 successors, 4..7 predecessors); under reverse complement the predecessor L
                    // becomes the successor 3-L and the successor R the predecessor 3-R: the shift amounts are L^4 | R forward, L^3 | R^7 reversed
                    const unsigned rnk_ = (hdr & 63u) + 1u;
                    const bool hasp = (idx > 0) | ((hdr & 64u) != 0), hass = (idx + 1 < rnk_) | ((hdr & 128u) != 0);
                    const unsigned shl_p = (e0 & 3u) ^ (rc ? 3u : 4u), shl_s = ((e3 >> 26) & 3u) ^ (rc ? 7u : 0u);
                    const unsigned ctx = (hasp ? 1u << shl_p : 0u) | (hass ? 1u << shl_s : 0u);
                    const uint32_t fa = (uint32_t)k.hi ^ (uint32_t)(k.lo >> 32), fb = (uint32_t)(k.hi >> 32) ^ (uint32_t)k.lo;
                    const uint32_t h1 = (fa + ((fb << 16) | (fb >> 16))) * 0x9E3779B1u;
                    const unsigned s = h1 >> (32 - LOG_CAP);
                    active &= ((h1 >> 2) & (P - 1)) == cls;
                    const uint64_t h0 = ld64(&keys[2 * s]), l0 = ld64(&keys[2 * s + 1]);
                    wtick(1);
                    // ---- C: ONE look at the key's home slot.  Hit -> count; free -> claim (64-bit CAS on hi, then lo); anything
                    //      else (another key there, a lost claim, an owner still writing) is parked in the wave's private queue and
                    //      finished later 64 at a time, so the data-dependent probe sequences never run with a handful of live lanes.
                    const bool hit = active & (h0 == k.hi) & (l0 == k.lo);
                    const bool want = active & !hit & (h0 == EMPTY);
                    const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long*>(&keys[2 * s]), (unsigned long long)EMPTY,
                                                   (unsigned long long)(want ? k.hi : EMPTY));
                    const bool won = want & (old == EMPTY);
                    st64(&keys[won ? 2 * s + 1 : dummy_at + lane], k.lo);                  // (an index select keeps it a ds_write; a pointer select becomes a FLAT store)
                    const bool done = hit | won;
                    // only min(255, count) is ever used (:943-949): 24 bits cannot wrap while the bucket has < 2^18 records
                    uint32_t inc = done ? 1u : 0u;
                    if (big) inc = (done && (ld32(&cc[s]) & 0xFFFFFFu) < 0xFFF000u) ? 1u : 0u;
                    atomicAdd(&cc[s], inc);
                    atomicOr(&cc[s], done ? ctx << 24 : 0u);
                    const bool parked = active & !done;
                    unsigned nnew = (unsigned)__builtin_popcountll(__ballot(won));
                    if (PARK) {
                        const unsigned long long pm = __ballot(parked);
                        const unsigned below = (unsigned)__builtin_popcount((uint32_t)pm & lane_lt_lo) + (unsigned)__builtin_popcount((uint32_t)(pm >> 32) & lane_lt_hi);
                        const unsigned npark = (unsigned)__builtin_popcountll(pm);
                        {   // parked lanes take qn .. qn+npark-1, the others the (unused) entries behind them: qn + 63 <= 126 < QCAP
                            const unsigned e = qn + (parked ? below : npark + lane - below);
                            qhi[e] = k.hi; qlo[e] = k.lo; qmeta[e] = s | (ctx << 16);
                        }
                        qn += npark;
                        if (qn >= 64) nnew += drain(64);
                    } else if (__any(parked)) {                      // (eight waves per SIMD cover the divergent probe sequences)
                        wave_lds_fence();
                        nnew += (unsigned)__builtin_popcountll(__ballot(probe(parked, k, s, ctx)));
                        wave_lds_fence();
                    }
                    if (nnew && lane == 0) atomicAdd(&misc[K3_FILL], nnew);
                    fill_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)fill_ld); ovf_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)ovf_ld);
                    w = wnext; wnext = (uint32_t)__builtin_amdgcn_readfirstlane((int)wdraw);
                    if (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); wtick(3); if (wv == 0) wt[4] += 1; }
                }
                {   // leftovers of this tile
                    unsigned nnew = 0;
                    while (qn) nnew += drain(qn < 64 ? qn : 64);
                    if (nnew && lane == 0) atomicAdd(&misc[K3_FILL], nnew);
                }
            }
            if (PROF) { if (lane == 0) { const uint32_t dt = (uint32_t)(__builtin_amdgcn_s_memtime() - tp0); atomicMax(&misc[10], dt); wcount += dt; } }
            tick(1);
            // ---- publish the look-ahead results and last emit's output base (their loads had the whole count phase)
            if (first_pass) {
                if (wv == 0) seg_store((it + 2) % 3, la_r0, la_cnt);
                if (tid == 0) st32(&bq[(it + 3) & 3], la_b);
                first_pass = false;
            }
            if (tid == 0) { misc[K3_BASELO] = (uint32_t)pend_base; misc[K3_BASEHI] = (uint32_t)(pend_base >> 32); misc[K3_CNT] = 0; }
            __syncthreads();                                         // A: all inserts done
            tick(2);
            if (PROF && tid == 0) { ptmax += misc[10]; misc[10] = 0; }
            {   // flush the previous emit's staging area: coalesced 8-B / 4-B stores
                const uint32_t nprev = misc[K3_NPREV];
                const unsigned long long pk = (unsigned long long)misc[K3_BASELO] | ((unsigned long long)misc[K3_BASEHI] << 32);
                const unsigned long long gb = pk & SMASK;
                for (unsigned i = tid; i < nprev; i += THREADS)
                    if (gb + i < solid_cap) { shi[gb + i] = sthi[i]; slo[gb + i] = stlo[i]; scc[gb + i] = stcc[i]; }
                if (tid == 0 && nprev && chunk_start && (pk >> 40) < chunk_cap) { chunk_start[pk >> 40] = gb; chunk_cnt[pk >> 40] = nprev; }
            }
            if (ld32(&misc[K3_OVF])) {                               // distinct set does not fit: refine this class and retry
                __syncthreads();
                for (unsigned i = tid; i < CAP; i += THREADS) { keys[2 * i] = EMPTY; keys[2 * i + 1] = EMPTY; cc[i] = 0; }
                if (tid == 0) {
                    misc[K3_NPREV] = 0; misc[K3_FILL] = 0; misc[K3_OVF] = 0;
                    atomicAdd(&counters[2], 1ull);
                    if (P >= (1u << 16)) { counters[3] = 1; misc[K3_DEPTH] = sp; }
                    else { stk[2 * sp] = cls + P; stk[2 * sp + 1] = 2 * P; stk[2 * sp + 2] = cls; stk[2 * sp + 3] = 2 * P; misc[K3_DEPTH] = sp + 2; }
                }
                __syncthreads();
                if (ld32(&misc[K3_DEPTH]) == 0) break;
                continue;
            }
            // ---- emit: histogram over ALL distinct k-mers (:1097); solid ones (:1098-1100) -> staging
            uint32_t vals[PER];
            unsigned long long sm[PER];
            unsigned nsolid = 0;
#pragma unroll
            for (unsigned j = 0; j < PER; ++j) {
                const unsigned i = j * THREADS + tid;
                const uint32_t v = cc[i];
                const bool occ = (v & 0xFFFFFFu) != 0;
                uint32_t cnt = v & 0xFFFFFFu; if (cnt > 255) cnt = 255;          // :943-949 saturating u8
                const unsigned long long m1 = __ballot(occ && cnt == 1);           // singletons (sequencing errors) dominate
                if (m1 && lane == (unsigned)__builtin_ctzll(m1)) atomicAdd(&lhist[1], (uint32_t)__builtin_popcountll(m1));
                if (occ && cnt != 1) atomicAdd(&lhist[cnt > 100 ? 100 : cnt], 1u);
                if (occ) ++my_distinct;
                const bool solid = occ && cnt >= min_freq;
                if (occ) { cc[i] = 0; }
                if (occ && !solid) { keys[2 * i] = EMPTY; keys[2 * i + 1] = EMPTY; }
                vals[j] = solid ? (cnt | ((v >> 24) << 8) | 0x80000000u) : 0u;
                sm[j] = __ballot(solid);
                nsolid += (unsigned)__builtin_popcountll(sm[j]);
            }
            uint32_t wbase = 0;
            if (nsolid && lane == 0) wbase = atomicAdd(&misc[K3_CNT], nsolid);
            wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
            tick(3);
            __syncthreads();                                         // B: staging flushed, solid total known
            tick(4);
            const uint32_t tot = ld32(&misc[K3_CNT]);
            if (tot <= SC) {
                if (tid == 0) {
                    misc[K3_NPREV] = tot; misc[K3_FILL] = 0; misc[K3_DEPTH] = sp;
                    pend_base = tot ? atomicAdd(&counters[0], (1ull << 40) | tot) : 0ull;          // consumed at the next barrier A
                }
                unsigned run = wbase;
#pragma unroll
                for (unsigned j = 0; j < PER; ++j) {
                    if (vals[j] >> 31) {
                        const unsigned i = j * THREADS + tid;
                        const unsigned pos = run + (unsigned)__builtin_popcountll(sm[j] & ((1ull << lane) - 1));
                        sthi[pos] = keys[2 * i]; stlo[pos] = keys[2 * i + 1]; stcc[pos] = vals[j] & 0xFFFFu;
                        keys[2 * i] = EMPTY; keys[2 * i + 1] = EMPTY;
                    }
                    run += (unsigned)__builtin_popcountll(sm[j]);
                }
            } else {                                                 // more solid k-mers than the staging area holds: direct
                if (tid == 0) {
                    const unsigned long long pk = atomicAdd(&counters[0], (1ull << 40) | tot), base = pk & SMASK;
                    if (chunk_start && (pk >> 40) < chunk_cap) { chunk_start[pk >> 40] = base; chunk_cnt[pk >> 40] = tot; }
                    misc[K3_B2LO] = (uint32_t)base; misc[K3_B2HI] = (uint32_t)(base >> 32);
                    misc[K3_NPREV] = 0; misc[K3_FILL] = 0; misc[K3_DEPTH] = sp; pend_base = 0;
                }
                __syncthreads();
                const unsigned long long gb = (unsigned long long)misc[K3_B2LO] | ((unsigned long long)misc[K3_B2HI] << 32);
                unsigned run = wbase;
#pragma unroll
                for (unsigned j = 0; j < PER; ++j) {
                    if (vals[j] >> 31) {
                        const unsigned i = j * THREADS + tid;
                        const unsigned long long pos = gb + run + (unsigned)__builtin_popcountll(sm[j] & ((1ull << lane) - 1));
                        if (pos < solid_cap) { shi[pos] = keys[2 * i]; slo[pos] = keys[2 * i + 1]; scc[pos] = vals[j] & 0xFFFFu; }
                        keys[2 * i] = EMPTY; keys[2 * i + 1] = EMPTY;
                    }
                    run += (unsigned)__builtin_popcountll(sm[j]);
                }
            }
            tick(5);
            if (sp == 0) break;                                      // common case: the next bucket's barrier S1 closes this emit
            __syncthreads();                                         // C
        }
    }
    // ---- drain: last staging area, histogram, distinct count
    __syncthreads();
    if (tid == 0) { misc[K3_BASELO] = (uint32_t)pend_base; misc[K3_BASEHI] = (uint32_t)(pend_base >> 32); }
    __syncthreads();
    {
        const uint32_t nprev = misc[K3_NPREV];
        const unsigned long long pk = (unsigned long long)misc[K3_BASELO] | ((unsigned long long)misc[K3_BASEHI] << 32);
        const unsigned long long gb = pk & SMASK;
        for (unsigned i = tid; i < nprev; i += THREADS)
            if (gb + i < solid_cap) { shi[gb + i] = sthi[i]; slo[gb + i] = stlo[i]; scc[gb + i] = stcc[i]; }
        if (tid == 0 && nprev && chunk_start && (pk >> 40) < chunk_cap) { chunk_start[pk >> 40] = gb; chunk_cnt[pk >> 40] = nprev; }
    }
    for (unsigned i = tid; i < 101; i += THREADS) if (lhist[i]) atomicAdd(&ghist[i], (unsigned long long)lhist[i]);
    for (int o = 32; o > 0; o >>= 1) my_distinct += __shfl_down(my_distinct, o);
    if (lane == 0 && my_distinct) atomicAdd(&counters[1], my_distinct);
    if (PROF && tid == 0) { for (int i = 0; i < 6; ++i) atomicAdd(&counters[106 + i], pt[i]); atomicAdd(&counters[112], ptmax);
                            for (int i = 0; i < 5; ++i) atomicAdd(&counters[113 + i], wt[i]); }
    if (PROF && lane == 0) atomicAdd(&counters[124 + wv], wcount);          // count-phase clocks of every wave
}

// =============================================================================== K3, round 4: fingerprint + reference slots
// The same counting (collapse_entries :1002-1013, combine_Entries :943-949, filter + histogram :1094-1104) with an LDS table of
// 8 B per slot instead of 20: a slot holds a 16-bit TAG of the k-mer's hash and a 16-bit REFERENCE (record, index) to the first
// instance of that k-mer among the bucket's super-k-mer records, which stay resident in LDS for the whole bucket, plus the usual
// count | context word.  Equality is still decided by CONTENT: a tag match is verified against the referenced instance's 120 bits
// (five LDS dwords, compared with this instance as extracted and as its reverse complement -- the referenced instance needs no
// canonicalisation).  4096 slots + 640 records + staging are 76 KB, so TWO blocks share a CU at the bucket size that one 157-KB
// block of k_count_buckets needed: one block's barriers, stage-in and emit run under the other block's window loop.
// What does not fit this shape -- more records than the resident tile, more than MAXK k-mers, a distinct set beyond the table, more
// segments than MAXSEG -- is DEFERRED: the bucket's id goes to a list that k_count_buckets (list mode) counts afterwards.
// The resident tile keeps the records at their own stride, 8 dwords.  An ODD stride (-DW2RAP_FP_TS9: the five stream words of neighbouring
// records on different LDS banks, 30 % conflict cycles instead of 41 %) was measured in round 4 and is WORSE for the step: k_count_fp
// 39.6 -> 39.0 ms, but k_table_insert beside it on the side stream 17 -> 31 ms (its solo speed against half of it) and the step 108.6 -> 114.2
// ms -- the same pair of numbers the 36-B records (stride 9) gave.  Not the LDS footprint (W2RAP_K3=22 with stride 9 is as large as the
// default with stride 8 and as slow), not the registers, not the stream priorities (the other way round and level: unchanged); unexplained.
#ifdef W2RAP_FP_TS9
constexpr unsigned FP_TS = REC_DWORDS | 1u;
#else
constexpr unsigned FP_TS = REC_DWORDS;
#endif
__device__ inline unsigned fp_tile_index(unsigned d) { return FP_TS == REC_DWORDS ? d : d + d / REC_DWORDS; }   // dword d of the record stream -> its place in the tile
template <unsigned THREADS, unsigned TILE_, unsigned SC_>
struct FpCfg {
    static constexpr unsigned CAP = 4096, LOG_CAP = 12, NW = THREADS / 64;
    static constexpr unsigned TILE = TILE_;                                   // records resident per bucket (a reference holds 10 bits of record)
    static constexpr unsigned ROUNDS = (TILE + THREADS - 1) / THREADS;        // records per thread in the flatten scan
    static constexpr unsigned MAXK = 16384, MAXWIN = MAXK / 64;               // flattened k-mers per bucket
    static constexpr unsigned PFREC = TILE < 512 ? TILE : 512;                // records whose dwords are prefetched into registers one bucket ahead ...
    static constexpr unsigned NPF = (PFREC * REC_DWORDS + THREADS - 1) / THREADS;   // ... the rare rest of a big bucket is loaded when its turn comes
    static constexpr unsigned SC = SC_;                                       // solid k-mers per bucket (references staged in LDS; more: deferred)
    static constexpr unsigned QCAP = 128;                                     // parked references per wave
    static constexpr unsigned MAXSEG = 16;
    static constexpr unsigned LIMIT = CAP - THREADS - 8 < 3072 ? CAP - THREADS - 8 : 3072;
    static constexpr unsigned PER = CAP / THREADS;
    static constexpr unsigned OCAP = CAP;                                     // occupied slots a bucket can end with
    static constexpr unsigned LDS = 3 * MAXSEG * 8 +
                                    (2 * CAP + TILE * FP_TS + 8 + MAXK / 32 + 2 + MAXWIN + 2 + 2 * SC + NW * QCAP + 3 * (MAXSEG + 1) + 32 + 4 + 104 + 16 + 40 + (OCAP + 1) / 2) * 4;
    static_assert(TILE < 1023 && ROUNDS * NW <= 16 && (1u << LOG_CAP) == CAP, "FpCfg");
};
enum { FP_FILL = 0, FP_OVF, FP_CNT, FP_BASELO, FP_BASEHI, FP_WIN, FP_DEPTH };
#ifndef W2RAP_FP_TICKETS
#define W2RAP_FP_TICKETS 4
#endif
constexpr uint32_t FP_TICKETS = W2RAP_FP_TICKETS;          // buckets a block takes from the queue per atomic (a power of two)
static_assert((FP_TICKETS & (FP_TICKETS - 1)) == 0, "FP_TICKETS");

// one k-mer instance of the resident tile: the words every stage needs
struct FpInst {
    uint32_t s[4];                  // as extracted, LSB first: bases 0..15 | 16..29 | 30..45 | 46..59 (32 + 28 + 32 + 28 bits)
    uint32_t e0, e3;                // the stream words that carry the flanks
};
__device__ inline FpInst fp_fetch(const uint32_t* tile, unsigned rec, unsigned idx) {
    const uint32_t* wp = tile + rec * FP_TS;
    const unsigned q0 = (idx + REC_HB / 2) >> 4, sh = ((idx + REC_HB / 2) & 15u) * 2u;       // base t of the record at bit REC_HB + 2 t
    const uint32_t d0 = wp[q0], d1 = wp[1 + q0], d2 = wp[2 + q0], d3 = wp[3 + q0], d4 = wp[4 + q0];
    FpInst x;
    const uint32_t e0 = __funnelshift_r(d0, d1, sh), e1 = __funnelshift_r(d1, d2, sh), e2 = __funnelshift_r(d2, d3, sh), e3 = __funnelshift_r(d3, d4, sh);
    // 128 stream bits from base idx: 1:0 left flank, 2..121 the k-mer, 123:122 right flank
    x.s[0] = __funnelshift_r(e0, e1, 2); x.s[1] = (e1 >> 2) & 0x0FFFFFFFu;
    x.s[2] = __funnelshift_r(e1, e2, 30); x.s[3] = __funnelshift_r(e2, e3, 30) & 0x0FFFFFFFu;
    x.e0 = e0; x.e3 = e3;
    return x;
}
// the canonical form of an instance (MSB-first 2 x 60 bits as everywhere else), whether the reverse complement was taken, and the
// instance as extracted in MSB-first words (what a verification compares a reference's reverse complement with)
struct FpKey { uint64_t hi, lo; bool rc; uint32_t f[4]; };
__device__ inline FpKey fp_key(const FpInst& x) {
    auto rev2_32 = [](uint32_t v) { v = __brev(v); return ((v & 0x55555555u) << 1) | ((v >> 1) & 0x55555555u); };
    const uint32_t a0 = rev2_32(x.s[0]), b0 = rev2_32(x.s[1]), a1 = rev2_32(x.s[2]), b1 = rev2_32(x.s[3]);
    FpKey k;
    k.f[0] = a0 >> 4; k.f[1] = __funnelshift_r(b0, a0, 4); k.f[2] = a1 >> 4; k.f[3] = __funnelshift_r(b1, a1, 4);
    // reverse complement = complemented LSB-first halves, swapped
    const uint64_t khi_f = ((uint64_t)k.f[0] << 32) | k.f[1], klo_f = ((uint64_t)k.f[2] << 32) | k.f[3];
    const uint64_t khi_r = ((uint64_t)(~x.s[3] & 0x0FFFFFFFu) << 32) | (uint32_t)~x.s[2], klo_r = ((uint64_t)(~x.s[1] & 0x0FFFFFFFu) << 32) | (uint32_t)~x.s[0];
    k.rc = khi_r != khi_f ? khi_r < khi_f : klo_r < klo_f;
    k.hi = k.rc ? khi_r : khi_f; k.lo = k.rc ? klo_r : klo_f;
    return k;
}
__device__ inline uint32_t fp_hash(const FpKey& k) {
    const uint32_t fa = (uint32_t)k.hi ^ (uint32_t)(k.lo >> 32), fb = (uint32_t)(k.hi >> 32) ^ (uint32_t)k.lo;
    return (fa + ((fb << 16) | (fb >> 16))) * 0x9E3779B1u;
}
// is the instance (rec, idx) of the tile the same k-mer as (x, k), on either strand?
__device__ inline bool fp_same(const uint32_t* tile, unsigned ref, const FpInst& x, const FpKey& k) {
    const FpInst r = fp_fetch(tile, ref >> 6, ref & 63u);
    const uint32_t same = (r.s[0] ^ x.s[0]) | (r.s[1] ^ x.s[1]) | (r.s[2] ^ x.s[2]) | (r.s[3] ^ x.s[3]);
    // the reference's reverse complement, MSB first, is (~s3, ~s2, ~s1, ~s0): equal to this instance's forward words iff every XOR is all ones
    const uint32_t opp = ((r.s[3] ^ k.f[0]) | 0xF0000000u) & (r.s[2] ^ k.f[1]) & ((r.s[1] ^ k.f[2]) | 0xF0000000u) & (r.s[0] ^ k.f[3]);
    return same == 0u || opp == 0xFFFFFFFFu;
}

template <unsigned THREADS, unsigned MINW, unsigned TILE_, unsigned SC_>
__global__ void __launch_bounds__(THREADS, MINW) k_count_fp(uint32_t nb, uint32_t b_lo, uint32_t b_hi, uint32_t nseg, const uint64_t* __restrict__ roff,
                                                            const uint32_t* __restrict__ recs, uint32_t min_freq, uint32_t* __restrict__ queue,
                                                            uint64_t* __restrict__ shi, uint64_t* __restrict__ slo, uint32_t* __restrict__ scc, uint64_t solid_cap,
                                                            unsigned long long* __restrict__ counters, unsigned long long* __restrict__ ghist,
                                                            uint64_t* __restrict__ chunk_start, uint32_t* __restrict__ chunk_cnt, uint32_t chunk_cap,
                                                            uint32_t* __restrict__ defer /* [0] count, [2..] deferred bucket ids */, uint32_t defer_cap,
                                                            uint32_t fill_limit /* <= LIMIT */, uint32_t solid_limit /* <= SC: smaller values are test hooks */,
                                                            uint8_t* __restrict__ sctx, uint8_t* __restrict__ unres, uint32_t* __restrict__ nbr /* all three or none:
                                                            the chunk-local adjacency prune (k_prune_local's outputs) done here, while the table still holds every
                                                            distinct k-mer of the bucket with its count */) {
    constexpr unsigned long long SMASK = (1ull << 40) - 1;
    using C = FpCfg<THREADS, TILE_, SC_>;
    constexpr unsigned CAP = C::CAP, NW = C::NW, TILE = C::TILE, ROUNDS = C::ROUNDS, NPF = C::NPF, SC = C::SC, MAXSEG = C::MAXSEG, PER = C::PER, QCAP = C::QCAP,
                       MAXK = C::MAXK, MAXWIN = C::MAXWIN;
    constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* segbase = reinterpret_cast<uint64_t*>(smem);         // [3][MAXSEG]
    uint32_t* tab = reinterpret_cast<uint32_t*>(segbase + 3 * MAXSEG);   // [CAP] tag << 16 | record << 6 | index; ~0 = empty
    uint32_t* cc = tab + CAP;                                      // [CAP] count (23:0) | context (31:24)
    uint32_t* tile = cc + CAP;                                     // the bucket's records (+8 dwords of slack)
    uint32_t* bv32 = tile + TILE * FP_TS + 8;                 // [MAXK/32 + 2] record-start bits of the flattened k-mers
    uint32_t* Bw = bv32 + MAXK / 32 + 2;                           // [MAXWIN + 2] record covering the first position of each window
    uint32_t* stcc = Bw + MAXWIN + 2;                              // [SC] count | context of the bucket's solid k-mers, compacted ...
    uint32_t* stref = stcc + SC;                                   // [SC] ... and the instance each of them refers to
    uint32_t* qref = stref + SC + (threadIdx.x >> 6) * QCAP;       // this wave's parked instances
    uint32_t* segdpre = stref + SC + NW * QCAP;                    // [3][MAXSEG + 1]
    uint32_t* wtot = segdpre + 3 * (MAXSEG + 1);                   // [32] per-wave partial sums (flatten: k-mers; emit: occupied and solid slots)
    uint32_t* bq = wtot + 32;                                      // ring of bucket ids
    uint32_t* lhist = bq + 4;                                      // 104
    uint32_t* misc = lhist + 104;                                  // 16
    uint32_t* stk = misc + 16;                                     // (class, P) pairs of a bucket counted in hash classes, depth <= 18
    uint16_t* olist = reinterpret_cast<uint16_t*>(stk + 40);       // [OCAP] the occupied slots of the bucket being emitted
    const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    auto seg_load = [&](uint32_t bb, uint64_t& r0, uint32_t& cnt) {
        r0 = 0; cnt = 0;
        if (tid < nseg && bb < b_hi) {
            const uint64_t a = roff[(uint64_t)tid * nb + bb], e = roff[(uint64_t)tid * nb + bb + 1];
            r0 = a; cnt = e - a < (1ull << 20) ? (uint32_t)(e - a) : (1u << 20);       // (anything beyond the tile is deferred)
        }
    };
    auto seg_store = [&](unsigned q, uint64_t r0, uint32_t cnt) {       // wave 0 only
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
        const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        const uint32_t ex = incl - cnt;
        if (lane < nseg) { segdpre[q * (MAXSEG + 1) + lane] = ex * REC_DWORDS; segbase[q * MAXSEG + lane] = (r0 - ex) * REC_DWORDS; }
        if (lane == 0) segdpre[q * (MAXSEG + 1) + nseg] = tot * REC_DWORDS;
    };
    // dword d of the logical record stream of the bucket in ring slot q (all segments as one stream)
    auto stream_dword = [&](unsigned q, const uint32_t* dp, const uint32_t (&bnd)[7], uint32_t d) -> uint32_t {
        unsigned s = 0;
        if (nseg <= 8) {
#pragma unroll
            for (unsigned i = 0; i < 7; ++i) s += d >= bnd[i] ? 1u : 0u;
        } else while (d >= dp[s + 1]) ++s;
        return recs[segbase[q * MAXSEG + s] + d];
    };
    // the first PFREC records of the bucket in ring slot q -> registers (coalesced dwords of the logical record stream)
    auto tile_load = [&](unsigned q, uint32_t (&v)[NPF]) {
        const uint32_t* dp = segdpre + q * (MAXSEG + 1);
        const uint32_t dall = dp[nseg];
        const uint32_t dend = dall < C::PFREC * REC_DWORDS ? dall : C::PFREC * REC_DWORDS;
        uint32_t bnd[7];
#pragma unroll
        for (unsigned i = 0; i < 7; ++i) bnd[i] = (i + 1 < nseg && nseg <= 8) ? dp[i + 1] : 0xFFFFFFFFu;
        // (16 B per lane and load -- a quarter of the loads and of the segment searches -- was measured in round 4: k_count_fp alone 36.9 -> 36.4 ms,
        //  but 39.6 -> 41.0 ms beside k_table_insert; the step 108.8 -> 110.2 ms.  Dword loads stay.)
#pragma unroll
        for (unsigned j = 0; j < NPF; ++j) {
            const uint32_t d = j * THREADS + tid;
            v[j] = d < dend ? stream_dword(q, dp, bnd, d) : 0u;
        }
    };
    auto tile_store = [&](const uint32_t (&v)[NPF]) {
#pragma unroll
        for (unsigned j = 0; j < NPF; ++j) { const unsigned i = j * THREADS + tid; if (i < C::PFREC * REC_DWORDS) tile[fp_tile_index(i)] = v[j]; }
    };
    // records PFREC .. TILE-1 of a bucket that has them: straight from global memory (its latency is not hidden: one bucket in six at most)
    auto tile_rest = [&](unsigned q) {
        if (TILE <= C::PFREC) return;
        const uint32_t* dp = segdpre + q * (MAXSEG + 1);
        const uint32_t dall = dp[nseg];
        const uint32_t dend = dall < TILE * REC_DWORDS ? dall : TILE * REC_DWORDS;
        uint32_t bnd[7];
#pragma unroll
        for (unsigned i = 0; i < 7; ++i) bnd[i] = (i + 1 < nseg && nseg <= 8) ? dp[i + 1] : 0xFFFFFFFFu;
        for (uint32_t d = C::PFREC * REC_DWORDS + tid; d < dend; d += THREADS) tile[fp_tile_index(d)] = stream_dword(q, dp, bnd, d);
    };
    // context bits of an instance (KMerContext: bits 0..3 successors, 4..7 predecessors); see k_count_buckets
    auto ctx_of = [&](const FpInst& x, bool rc, uint32_t hdr, unsigned idx) -> unsigned {
        const unsigned rnk_ = (hdr & 63u) + 1u;
        const bool hasp = (idx > 0) | ((hdr & 64u) != 0), hass = (idx + 1 < rnk_) | ((hdr & 128u) != 0);
        const unsigned shl_p = (x.e0 & 3u) ^ (rc ? 3u : 4u), shl_s = ((x.e3 >> 26) & 3u) ^ (rc ? 7u : 0u);
        return (hasp ? 1u << shl_p : 0u) | (hass ? 1u << shl_s : 0u);
    };

    // ---- init
    for (unsigned i = tid; i < CAP; i += THREADS) { tab[i] = EMPTY; cc[i] = 0; }
    for (unsigned i = tid; i < MAXK / 32 + 2; i += THREADS) bv32[i] = 0;
    for (unsigned i = tid; i < TILE * FP_TS + 8; i += THREADS) tile[i] = 0;
    for (unsigned i = tid; i < 104; i += THREADS) lhist[i] = 0;
    if (tid < 16) misc[tid] = 0;
    if (tid == 0) {
        const uint32_t t0 = atomicAdd(queue, 3u), nbk = b_hi - b_lo;
        bq[0] = t0 < nbk ? b_lo + t0 : NONE32; bq[1] = t0 + 1 < nbk ? b_lo + t0 + 1 : NONE32; bq[2] = t0 + 2 < nbk ? b_lo + t0 + 2 : NONE32; bq[3] = NONE32;
    }
    __syncthreads();
    if (wv == 0) {
        uint64_t r0; uint32_t cnt;
        seg_load(bq[0], r0, cnt); seg_store(0, r0, cnt);
        seg_load(bq[1], r0, cnt); seg_store(1, r0, cnt);
    }
    __syncthreads();
    uint32_t pf[NPF];
    tile_load(0, pf);
    unsigned long long my_distinct = 0;
    unsigned qn = 0;
    uint32_t tick4 = 0;

    // ---- finish the top `cnt` (<= 64) parked instances of this wave: the full probe sequence; returns the new keys
    auto drain = [&](unsigned cnt) -> unsigned {
        wave_lds_fence();
        const bool live = lane < cnt;
        const unsigned ref = live ? qref[qn - cnt + lane] : 0u;
        const unsigned rec = ref >> 6, idx = ref & 63u;
        const FpInst x = fp_fetch(tile, rec, idx);
        const FpKey k = fp_key(x);
        const unsigned ctx = ctx_of(x, k.rc, tile[rec * FP_TS], idx);
        const uint32_t h1 = fp_hash(k), tag = (h1 >> 5) & 0xFFFFu;
        unsigned s = (h1 >> (33 - C::LOG_CAP)) << 1;                  // the probe sequence starts at an EVEN slot: the window loop looks at a pair
        bool isnew = false;
        if (live) {
            bool ok = false;
            int budget = (int)CAP;
            for (;;) {
                uint32_t a = ld32(&tab[s]);
                if (a == EMPTY) {
                    a = atomicCAS(&tab[s], EMPTY, (tag << 16) | ref);
                    if (a == EMPTY) { isnew = true; ok = true; break; }
                }
                if ((a >> 16) == tag && fp_same(tile, a & 0xFFFFu, x, k)) { ok = true; break; }
                s = (s + 1) & (CAP - 1);
                if (--budget <= 0) break;
            }
            if (ok) { atomicAdd(&cc[s], 1u); atomicOr(&cc[s], ctx << 24); }
            else st32(&misc[FP_OVF], 1u);
        }
        qn -= cnt;
        wave_lds_fence();
        return (unsigned)__builtin_popcountll(__ballot(isnew));
    };

    for (uint32_t it = 0;; ++it) {
        const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld32(&bq[it & 3]));
        if (b >= b_hi) break;
        const unsigned q = it % 3;
        const uint32_t nrec = (uint32_t)__builtin_amdgcn_readfirstlane((int)segdpre[q * (MAXSEG + 1) + nseg]) / REC_DWORDS;
        bool skip = nrec > TILE;                                      // block-uniform
        // ---- stage in this bucket's records; start the look-ahead loads (consumed before barrier A)
        tile_store(pf);
        if (!skip) tile_rest(q);
        uint32_t la_b = 0;
        if (tid == 0) {
            // Tickets FP_TICKETS at a time: an atomic on ONE address costs ~24 ns at its L2 channel whoever waits for it, and ~940 k buckets per
            // step each take a ticket here and an output range below -- 58 % of one channel's atomic unit for the kernel's duration.  Four at a
            // time: k_count_fp 38.9 -> 37.9 ms, the step -0.8 ms (three alternating runs each, profiles/r05_tickets_ab.txt).
            if ((it & (FP_TICKETS - 1u)) == 0) tick4 = atomicAdd(queue, FP_TICKETS);
            const uint32_t t = tick4 + (it & (FP_TICKETS - 1u));
            la_b = t < b_hi - b_lo ? b_lo + t : NONE32;
        }
        uint64_t la_r0 = 0; uint32_t la_cnt = 0;
        if (wv == 0) seg_load(ld32(&bq[(it + 2) & 3]), la_r0, la_cnt);
        tile_load((it + 1) % 3, pf);
        __syncthreads();                                             // S1
        // ---- flatten: record rec occupies positions [e, e + nk) of the bucket-wide k-mer numbering (see k_count_buckets)
        unsigned nkr[ROUNDS], inclr[ROUNDS];
#pragma unroll
        for (unsigned r = 0; r < ROUNDS; ++r) {
            const unsigned rec = r * THREADS + tid;
            nkr[r] = (!skip && rec < nrec) ? (tile[rec * FP_TS] & 63u) + 1u : 0u;
            inclr[r] = wave_scan64(nkr[r]);
            if (lane == 63) wtot[r * NW + wv] = inclr[r];
        }
        if (tid == 0) { misc[FP_WIN] = 2 * NW; stk[0] = 0; stk[1] = 1; misc[FP_DEPTH] = 1; }
        __syncthreads();                                             // X1
        const unsigned wsum = lane < ROUNDS * NW ? wtot[lane] : 0u, wsc = row_scan16(wsum);
        const int wvu = __builtin_amdgcn_readfirstlane((int)wv);
        const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)wsc, ROUNDS * NW - 1);
        skip |= total > MAXK;
#pragma unroll
        for (unsigned r = 0; r < ROUNDS; ++r) {
            const unsigned base = (unsigned)__builtin_amdgcn_readlane((int)wsc, r * NW + wvu) - (unsigned)__builtin_amdgcn_readlane((int)wsum, r * NW + wvu);
            if (nkr[r] && !skip) {
                const unsigned e = base + inclr[r] - nkr[r], rec = r * THREADS + tid;
                atomicOr(&bv32[e >> 5], 1u << (e & 31));
                const unsigned w1 = (e + 63) >> 6;
                if (64 * w1 < e + nkr[r]) Bw[w1] = (rec << 16) | e;
            }
        }
        __syncthreads();                                             // X2
        const unsigned nwin = skip ? 0u : (total + 63) / 64;
        bool first_pass = true;
        // A bucket whose distinct set does not fit the table (or whose solid set does not fit the staging area) is counted in hash classes:
        // (class, P) work stack as in k_count_buckets (= MapReduceEngine.h:288-291); the records stay where they are.  Normally one pass.
        for (;;) {
        const unsigned sp = (unsigned)__builtin_amdgcn_readfirstlane((int)ld32(&misc[FP_DEPTH])) - 1u;
        const uint32_t cls = (uint32_t)__builtin_amdgcn_readfirstlane((int)stk[2 * sp]), P = (uint32_t)__builtin_amdgcn_readfirstlane((int)stk[2 * sp + 1]);
        {
            // ---- the windows of this wave (dealt out on demand); straight-line, predicated code as in k_count_buckets
            uint32_t fill_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld32(&misc[FP_FILL])),
                     ovf_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)ld32(&misc[FP_OVF]));
            const unsigned wv_s = (unsigned)wvu;
            uint32_t nM0 = 0, nM1 = 0, nB = 0;
            { const unsigned w0 = wv_s < nwin ? wv_s : 0u; nM0 = bv32[2 * w0]; nM1 = bv32[2 * w0 + 1]; nB = Bw[w0]; }
            const uint64_t lane_le = ~0ull >> (63 - lane);
            const uint32_t lane_lt_lo = lane < 32 ? (1u << lane) - 1u : 0xFFFFFFFFu, lane_lt_hi = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;
            unsigned wnext = wv_s + NW;
            for (unsigned w = wv_s; w < nwin;) {
                if (ovf_seen) break;
                if (fill_seen >= fill_limit) { if (lane == 0) st32(&misc[FP_OVF], 1u); break; }
                const uint32_t fill_ld = ld32(&misc[FP_FILL]), ovf_ld = ld32(&misc[FP_OVF]);
                uint32_t wdraw = 0;
                if (lane == 0) wdraw = atomicAdd(&misc[FP_WIN], 1u);
                // ---- A: locate every lane's k-mer (record, index)
                const uint32_t M0 = nM0, M1 = nM1, Bv = nB;
                { const unsigned wn = wnext < nwin ? wnext : w; nM0 = bv32[2 * wn]; nM1 = bv32[2 * wn + 1]; nB = Bw[wn]; }
                const unsigned g = w * 64 + lane;
                const bool active = g < total;
                const uint32_t mle0 = M0 & (uint32_t)lane_le, mle1 = M1 & (uint32_t)(lane_le >> 32);
                const unsigned c = (unsigned)__builtin_popcount(mle0 & ~1u) + (unsigned)__builtin_popcount(mle1);
                const unsigned top = mle1 ? 63u - (unsigned)__builtin_clz(mle1) : 31u - (unsigned)__builtin_clz(mle0 | 1u);
                const unsigned idx = active ? lane - (c ? top : (Bv & 0xFFFFu) - w * 64) : 0u;
                const unsigned rec = active ? (Bv >> 16) + c : 0u;
                const uint32_t hdr = tile[rec * FP_TS];
                // ---- B: cut out the k-mer, canonicalise, hash, look at its home slot
                const FpInst x = fp_fetch(tile, rec, idx);
                const FpKey k = fp_key(x);
                const unsigned ctx = ctx_of(x, k.rc, hdr, idx);
                const uint32_t h1 = fp_hash(k), tag = (h1 >> 5) & 0xFFFFu;
                const unsigned s0 = (h1 >> (33 - C::LOG_CAP)) << 1;
                const unsigned myref = (rec << 6) | idx;
                // ---- C: ONE 8-byte look at the first TWO slots of the k-mer's probe sequence.  A tag match is verified against the instance it
                //      refers to (lanes without one compare with themselves); no match and a free slot among the two -> claim the first free one;
                //      anything else (two other keys there, a lost claim, a tag match that is another k-mer) is parked and finished by drain().
                const uint64_t ab = ld64(reinterpret_cast<const uint64_t*>(tab + s0));
                const uint32_t a0 = (uint32_t)ab, a1 = (uint32_t)(ab >> 32);
                const bool m0 = ((a0 >> 16) == tag) & (a0 != EMPTY), m1 = ((a1 >> 16) == tag) & (a1 != EMPTY);
                const bool second = !m0 & (m1 | ((a0 != EMPTY) & (a1 == EMPTY)));        // act on the second slot: it matches, or it is the first free one
                const uint32_t a = second ? a1 : a0;
                const unsigned s = s0 + (second ? 1u : 0u);
                const bool mine = active & (((h1 >> 2) & (P - 1)) == cls);                 // (a bucket counted in hash classes: this pass's class only)
                const bool tagm = mine & (m0 | m1);
                const bool hit = tagm & fp_same(tile, tagm ? (a & 0xFFFFu) : myref, x, k);
                const bool want = mine & !tagm & (a == EMPTY);
                const uint32_t old = atomicCAS(&tab[s], EMPTY, want ? ((tag << 16) | myref) : EMPTY);
                const bool won = want & (old == EMPTY);
                const bool done = hit | won;
                atomicAdd(&cc[s], done ? 1u : 0u);
                atomicOr(&cc[s], done ? ctx << 24 : 0u);
                const bool parked = mine & !done;
                unsigned nnew = (unsigned)__builtin_popcountll(__ballot(won));
                {
                    const unsigned long long pm = __ballot(parked);
                    const unsigned below = (unsigned)__builtin_popcount((uint32_t)pm & lane_lt_lo) + (unsigned)__builtin_popcount((uint32_t)(pm >> 32) & lane_lt_hi);
                    const unsigned npark = (unsigned)__builtin_popcountll(pm);
                    // parked lanes take qn .. qn+npark-1, the others the (unused) entries behind them: qn + 63 <= 126 < QCAP
                    qref[qn + (parked ? below : npark + lane - below)] = myref;
                    qn += npark;
                    if (qn >= 64) nnew += drain(64);
                }
                if (nnew && lane == 0) atomicAdd(&misc[FP_FILL], nnew);
                fill_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)fill_ld); ovf_seen = (uint32_t)__builtin_amdgcn_readfirstlane((int)ovf_ld);
                w = wnext; wnext = (uint32_t)__builtin_amdgcn_readfirstlane((int)wdraw);
            }
            unsigned nnew = 0;
            while (qn) nnew += drain(qn < 64 ? qn : 64);
            if (nnew && lane == 0) atomicAdd(&misc[FP_FILL], nnew);
        }
        // ---- publish the look-ahead results
        if (first_pass) {
            if (wv == 0) seg_store((it + 2) % 3, la_r0, la_cnt);
            if (tid == 0) st32(&bq[(it + 3) & 3], la_b);
            first_pass = false;
        }
        if (tid == 0) {
            misc[FP_CNT] = 0;
            if (skip) misc[FP_OVF] = 1;
        }
        __syncthreads();                                             // A: all inserts done
        bool deferred = ld32(&misc[FP_OVF]) != 0;
        // ---- emit, pass 0: every thread looks at PER neighbouring slots (16-byte LDS reads): which are occupied, how many are solid
        static_assert(PER % 4 == 0, "PER");
        uint32_t occm = 0, nsol = 0;
        if (!deferred) {
            const u32x4* c4 = reinterpret_cast<const u32x4*>(cc + tid * PER);
#pragma unroll
            for (unsigned j4 = 0; j4 < PER / 4; ++j4) {
                const u32x4 v4 = c4[j4];
#pragma unroll
                for (unsigned j = 0; j < 4; ++j) {
                    const uint32_t cnt = v4[j] & 0xFFFFFFu;
                    occm |= (cnt != 0 ? 1u : 0u) << (4 * j4 + j);
                    nsol += (cnt != 0 && cnt >= min_freq) ? 1u : 0u;
                }
            }
        }
        const unsigned nocc = (unsigned)__builtin_popcount(occm);
        const unsigned i_occ = wave_scan64(nocc), i_sol = wave_scan64(nsol);
        if (lane == 63) { wtot[wv] = i_occ; wtot[16 + wv] = i_sol; }
        __syncthreads();                                             // P0
        const unsigned ws_o = lane < NW ? wtot[lane] : 0u, wc_o = row_scan16(ws_o), ws_s = lane < NW ? wtot[16 + lane] : 0u, wc_s = row_scan16(ws_s);
        const unsigned tot_occ = (unsigned)__builtin_amdgcn_readlane((int)wc_o, NW - 1), tot = (unsigned)__builtin_amdgcn_readlane((int)wc_s, NW - 1);
        deferred |= tot > solid_limit;                               // (nothing has been written yet)
        if (deferred) {
            // the table or the staging area is too small for this class: refine it and count again; a bucket that is not this kernel's shape
            // at all (more records than the tile, more k-mers than the start bits) goes to the list kernel, which counts it from scratch
            for (unsigned i = tid; i < CAP; i += THREADS) { tab[i] = EMPTY; cc[i] = 0; }
            if (tid == 0) {
                if (skip) {
                    const uint32_t at = atomicAdd(&defer[0], 1u);
                    if (at < defer_cap) defer[2 + at] = b; else counters[3] = 3;
                    misc[FP_DEPTH] = sp;
                } else if (P >= (1u << 16)) { counters[3] = 1; misc[FP_DEPTH] = sp; }
                else { atomicAdd(&counters[2], 1ull); stk[2 * sp] = cls + P; stk[2 * sp + 1] = 2 * P; stk[2 * sp + 2] = cls; stk[2 * sp + 3] = 2 * P; misc[FP_DEPTH] = sp + 2; }
                misc[FP_FILL] = 0; misc[FP_OVF] = 0; misc[FP_WIN] = 2 * NW;
            }
            __syncthreads();
            if (ld32(&misc[FP_DEPTH]) == 0) break;
            continue;
        }
        {
            unsigned at = (unsigned)__builtin_amdgcn_readlane((int)wc_o, wvu) - (unsigned)__builtin_amdgcn_readlane((int)ws_o, wvu) + i_occ - nocc;
            for (uint32_t m = occm; m; m &= m - 1) olist[at++] = (uint16_t)(tid * PER + (unsigned)__builtin_ctz(m));
        }
        if (tid == 0) {
            // the bucket's output range and chunk number: ONE global atomic (count in bits 39:0, chunks above); its result is needed behind
            // barrier B2 only -- this lane waits for it there while the block runs pass 1, and the CU's other block does not wait at all
            misc[FP_FILL] = 0; misc[FP_DEPTH] = sp; misc[FP_WIN] = 2 * NW;
            const unsigned long long pk = tot ? atomicAdd(&counters[0], (1ull << 40) | tot) : 0ull;
            misc[FP_BASELO] = (uint32_t)pk; misc[FP_BASEHI] = (uint32_t)(pk >> 32);
            if (tot && chunk_start && (pk >> 40) < chunk_cap) { chunk_start[pk >> 40] = pk & SMASK; chunk_cnt[pk >> 40] = tot; }
        }
        __syncthreads();                                             // P1: the list of occupied slots is complete
        // ---- pass 1, dense over the occupied slots: histogram over ALL distinct k-mers (:1097), reset; the solid ones (:1098-1100) go to the
        //      staging area -- as references first, as keys after one dense extraction round
        for (unsigned i0 = 0; i0 < tot_occ; i0 += THREADS) {
            const unsigned i = i0 + tid;
            const bool live = i < tot_occ;
            const unsigned slot = live ? olist[i] : 0u;
            const uint32_t v = cc[slot];
            uint32_t cnt = v & 0xFFFFFFu; if (cnt > 255) cnt = 255;                  // :943-949 saturating u8
            const unsigned long long m1 = __ballot(live && cnt == 1);                  // singletons (sequencing errors) dominate
            if (m1 && lane == (unsigned)__builtin_ctzll(m1)) atomicAdd(&lhist[1], (uint32_t)__builtin_popcountll(m1));
            if (live && cnt != 1) atomicAdd(&lhist[cnt > 100 ? 100 : cnt], 1u);
            if (live) ++my_distinct;
            const bool solid = live && cnt >= min_freq;
            const unsigned long long sb = __ballot(solid);
            if (sb) {
                uint32_t base = 0;
                if (lane == (unsigned)__builtin_ctzll(sb)) base = atomicAdd(&misc[FP_CNT], (uint32_t)__builtin_popcountll(sb));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(sb));
                if (solid) {
                    const unsigned pos = base + (unsigned)__builtin_popcountll(sb & ((1ull << lane) - 1));
                    stref[pos] = tab[slot] & 0xFFFFu; stcc[pos] = cnt | ((v >> 24) << 8);
                    if (sctx) cc[slot] = 0x80000000u | pos;                             // (fused prune: a solid k-mer's slot names its place in the chunk ...
                }
            }
            if (live && sctx && !solid) cc[slot] = 0x40000000u;                         //  ... any other occupied slot says "not solid"; the table is reset behind the probes)
            if (live && !sctx) { cc[slot] = 0; tab[slot] = EMPTY; }
        }
        __syncthreads();                                             // B2: the references are in place, the output range is known
        {   // the keys of the solid k-mers, one dense extraction round, straight to their places (coalesced 8-B / 4-B stores)
            const unsigned long long gb = ((unsigned long long)misc[FP_BASELO] | ((unsigned long long)misc[FP_BASEHI] << 32)) & SMASK;
            for (unsigned i0 = 0; i0 < tot; i0 += THREADS) {
                const unsigned i = i0 + tid;
                const bool have = i < tot;
                const uint32_t ref = have ? stref[i] : 0u;
                const FpKey k = fp_key(fp_fetch(tile, ref >> 6, ref & 63u));
                const uint32_t ccv = have ? stcc[i] : 0u;
                if (have && gb + i < solid_cap) { shi[gb + i] = k.hi; slo[gb + i] = k.lo; scc[gb + i] = ccv; }
                if (sctx) {
                    // KmerDict::recomputeAdjacencies (ReadPather.h:317-346) as far as this bucket can tell: a neighbour k-mer shares the minimizer
                    // with probability 45/47 and then lies in THIS table with its whole count -- solid: the bit stays and the neighbour's node id
                    // is kept for the unipath links; present but not solid: the bit goes; absent: the bit stays open for k_prune's global probe
                    constexpr uint32_t NONE = 0xFFFFFFFFu, PAL = 0xFFFFFFFEu;
                    unsigned c = (ccv >> 8) & 0xFFu, un = 0;
                    uint32_t ns = NONE, np_ = NONE;
                    const Kmer kk{k.hi, k.lo};
                    for (unsigned rest = have ? c : 0u; __any(rest != 0);) {
                        if (rest) {
                            const unsigned t = (unsigned)__builtin_ctz(rest);
                            rest &= rest - 1;
                            Kmer nk = t < 4 ? kmer_succ(kk, t & 3) : kmer_pred(kk, t & 3);
                            const bool r = kmer_canon(nk);
                            FpInst nx; FpKey nkk;
                            const uint64_t lh = rev2_64(nk.hi) >> 4, ll = rev2_64(nk.lo) >> 4;     // LSB-first halves (base 0 at bits 1:0)
                            nx.s[0] = (uint32_t)lh; nx.s[1] = (uint32_t)(lh >> 32); nx.s[2] = (uint32_t)ll; nx.s[3] = (uint32_t)(ll >> 32); nx.e0 = 0; nx.e3 = 0;
                            nkk.hi = nk.hi; nkk.lo = nk.lo; nkk.rc = false;
                            nkk.f[0] = (uint32_t)(nk.hi >> 32); nkk.f[1] = (uint32_t)nk.hi; nkk.f[2] = (uint32_t)(nk.lo >> 32); nkk.f[3] = (uint32_t)nk.lo;
                            const uint32_t h1 = fp_hash(nkk), tag = (h1 >> 5) & 0xFFFFu;
                            unsigned s = (h1 >> (33 - C::LOG_CAP)) << 1;
                            int found = -1;
                            for (int budget = (int)CAP; budget > 0; --budget) {
                                const uint32_t a = tab[s];
                                if (a == EMPTY) break;
                                if ((a >> 16) == tag && fp_same(tile, a & 0xFFFFu, nx, nkk)) { found = (int)s; break; }
                                s = (s + 1) & (CAP - 1);
                            }
                            if (found < 0) un |= 1u << t;
                            else {
                                const uint32_t v = cc[found];
                                if (v & 0x80000000u) {
                                    const uint32_t id = kmer_is_pal(nk) ? PAL : (uint32_t)(2 * (gb + (v & 0xFFFFu)) + (r ? 1u : 0u));
                                    if (t < 4) ns = id; else np_ = id;
                                } else c &= ~(1u << t);
                            }
                        }
                    }
                    if (have && gb + i < solid_cap) {
                        const unsigned long long gi = gb + i;
                        sctx[gi] = (uint8_t)c; unres[gi] = (uint8_t)un;
                        // only meaningful when exactly one successor / predecessor survives (then it is the last one found): as k_prune_local
                        nbr[2 * gi] = (!(un & 15u) && popc4(c & 15) != 1) ? NONE : ns;
                        nbr[2 * gi + 1] = (!(un >> 4) && popc4(c >> 4) != 1) ? NONE : np_;
                    }
                }
            }
        }
        __syncthreads();                                             // C: the tile is free for the next bucket (or the next class starts)
        if (sctx) {                                                  // fused prune: the table was kept for the probes; reset it now
            for (unsigned i = tid; i < tot_occ; i += THREADS) { const unsigned slot = olist[i]; cc[slot] = 0; tab[slot] = EMPTY; }
            if (sp != 0) __syncthreads();                            // (the next class starts at once; the next bucket has barriers of its own)
        }
        if (sp == 0) break;
        }
        for (unsigned i = tid; i < 2 * nwin + 2; i += THREADS) bv32[i] = 0;            // the next bucket's start bits (behind its barriers S1 and X1)
    }
    // ---- histogram, distinct count
    __syncthreads();
    for (unsigned i = tid; i < 101; i += THREADS) if (lhist[i]) atomicAdd(&ghist[i], (unsigned long long)lhist[i]);
    for (int o = 32; o > 0; o >>= 1) my_distinct += __shfl_down(my_distinct, o);
    if (lane == 0 && my_distinct) atomicAdd(&counters[1], my_distinct);
}

// =============================================================================== K4
__global__ void __launch_bounds__(256) k_table_insert(uint64_t i0, uint64_t S, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo,
                                                       Slot* __restrict__ table, uint64_t mask) {
    uint64_t i = i0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    const uint64_t h = kmer_hash(Kmer{shi[i], slo[i]});
    const Slot v = slot_make(h, i);                                  // fingerprint | index: the whole entry is the claim
    uint64_t s = h & mask;
    while (atomicCAS(&table[s], SLOT_EMPTY, v) != SLOT_EMPTY) s = (s + 1) & mask;
}

// =============================================================================== K5
// KmerDict::recomputeAdjacencies (ReadPather.h:317-346): clear every context bit whose
// neighbour k-mer is not in the solid set.  Membership only, so it is order-free.
//
// Two steps.  k_prune_local works on one K3 emit chunk (the solid k-mers of one minimizer bucket) at a time: their keys
// go into an LDS hash table and every neighbour is looked for THERE first -- consecutive k-mers share their minimizer
// with probability ~45/47, so most surviving neighbours are found without touching HBM.  A bit whose neighbour is not in
// the chunk (another bucket, or not solid at all) stays set and is recorded in unres[i]; k_prune then probes the global
// table for exactly those bits.  Without chunks (multi-GPU: the gathered dictionary is renumbered) k_prune does it all.
// A chunk holds the solid k-mers of one bucket: ~4500 k-mer INSTANCES' worth, i.e. ~250 k-mers at 30x coverage, ~420 at 17x (the per-GPU share of
// BASELINE configs[4]), more below that.  A thread takes U of them (U = 2, 4, 8: chunks of up to 512, 1024, 2048 k-mers in a table of twice
// as many slots); the launcher picks U from the mean chunk size (PL_LAUNCH).  An oversized chunk is still exact -- its bits stay open for the
// global step -- but at U = 2 a 17x data set left a third of its chunks to it: 0.87 routed queries per k-mer instead of 0.26 in the sharded phase.
template <unsigned U> struct PlCfg { static constexpr unsigned CAP = 256 * U, SLOTS = 512 * U, BITS = U == 2 ? 10 : U == 4 ? 11 : 12; };
template <unsigned BITS>
__device__ inline unsigned pl_hash(Kmer k) {
    const uint32_t fa = (uint32_t)k.hi ^ (uint32_t)(k.lo >> 32), fb = (uint32_t)(k.hi >> 32) ^ (uint32_t)k.lo;
    return ((fa + ((fb << 16) | (fb >> 16))) * 0x9E3779B1u) >> (32 - BITS);
}
template <class Id, unsigned U = 2>
__global__ void __launch_bounds__(256) k_prune_local(uint64_t nchunks, const uint64_t* __restrict__ cstart, const uint32_t* __restrict__ ccnt,
                                                      const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo,
                                                      const uint32_t* __restrict__ scc, uint8_t* __restrict__ sctx,
                                                      Id* __restrict__ nbr, uint8_t* __restrict__ unres) {
    constexpr Id NONE = NodeId<Id>::NONE, PAL = NodeId<Id>::PAL;
    constexpr unsigned PL_CAP = PlCfg<U>::CAP, PL_SLOTS = PlCfg<U>::SLOTS, PL_BITS = PlCfg<U>::BITS;
    __shared__ uint64_t khi[PL_SLOTS], klo[PL_SLOTS];
    __shared__ uint16_t kix[PL_SLOTS];
    const unsigned tid = threadIdx.x;
    // software pipeline over this block's chunks: the descriptor of the chunk after next and the keys of the next chunk are
    // loaded while the current one is worked on in LDS
    auto load_desc = [&](uint64_t ch, uint64_t& st_, uint32_t& cn_) {
        st_ = 0; cn_ = 0;
        if (ch < nchunks) { st_ = cstart[ch]; cn_ = ccnt[ch]; if (cn_ > PL_CAP) cn_ = 0; }   // oversized chunk: its k-mers stay fully unresolved
    };
    auto load_keys = [&](uint64_t st_, uint32_t cn_, Kmer (&k_)[U], unsigned (&c_)[U]) {
#pragma unroll
        for (unsigned u = 0; u < U; ++u) {
            const unsigned j = tid + 256 * u;
            k_[u] = Kmer{0, 0}; c_[u] = 0;
            if (j < cn_) { k_[u] = Kmer{shi[st_ + j], slo[st_ + j]}; c_[u] = (scc[st_ + j] >> 8) & 0xFF; }
        }
    };
    uint64_t st_c, st_n; uint32_t cn_c, cn_n;
    Kmer nx[U]; unsigned ncx[U];
    load_desc(blockIdx.x, st_c, cn_c);
    load_keys(st_c, cn_c, nx, ncx);
    load_desc((uint64_t)blockIdx.x + gridDim.x, st_n, cn_n);
    for (uint64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const uint64_t start = st_c;
        const uint32_t cnt = cn_c;
        Kmer mine[U]; unsigned cm[U];
#pragma unroll
        for (unsigned u = 0; u < U; ++u) { mine[u] = nx[u]; cm[u] = ncx[u]; }
        st_c = st_n; cn_c = cn_n;
        load_keys(st_c, cn_c, nx, ncx);                           // next chunk's keys
        load_desc(ch + 2 * (uint64_t)gridDim.x, st_n, cn_n);      // the one after's descriptor
        if (cnt == 0) continue;
        __syncthreads();
        for (unsigned s = tid; s < PL_SLOTS; s += 256) khi[s] = EMPTY_HI;
        __syncthreads();
#pragma unroll
        for (unsigned u = 0; u < U; ++u) {
            const unsigned j = tid + 256 * u;
            if (j < cnt) {
                unsigned s = pl_hash<PL_BITS>(mine[u]);
                for (;;) {                                        // the keys of a chunk are distinct
                    const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&khi[s]), (unsigned long long)EMPTY_HI,
                                                             (unsigned long long)mine[u].hi);
                    if (old == EMPTY_HI) { klo[s] = mine[u].lo; kix[s] = (uint16_t)j; break; }
                    s = (s + 1) & (PL_SLOTS - 1);
                }
            }
        }
        __syncthreads();
        auto find = [&](Kmer nk) -> int {
            unsigned s = pl_hash<PL_BITS>(nk);
            for (;;) {
                const uint64_t h = khi[s];
                if (h == EMPTY_HI) return -1;
                if (h == nk.hi && klo[s] == nk.lo) return (int)kix[s];
                s = (s + 1) & (PL_SLOTS - 1);
            }
        };
#pragma unroll
        for (unsigned u = 0; u < U; ++u) {
            if (256 * u + (tid & ~63u) >= cnt) continue;          // this wavefront has no k-mer in this part of the chunk (wave-uniform)
            const unsigned j = tid + 256 * u;
            const Kmer k = mine[u];
            const unsigned c = j < cnt ? cm[u] : 0;
            unsigned un = 0;
            Id ns = NONE, np = NONE;
            // The reverse complement of a neighbour is a neighbour of the reverse complement: rc(k + b) = (3-b) + rc(k) without its last base,
            // rc(b + k) = rc(k) without its first base + (3-b).  One 120-bit reversal per k-mer instead of two per context bit.
            const Kmer rk = kmer_rc(k);
            // one set bit per trip (3-4 trips for a wave instead of eight branches): bits 0..3 successors, 4..7 predecessors
            for (unsigned rest = c; __any(rest != 0);) {
                if (rest) {
                    const unsigned t = (unsigned)__builtin_ctz(rest);
                    rest &= rest - 1;
                    const unsigned b = t & 3;
                    const Kmer fw = t < 4 ? kmer_succ(k, b) : kmer_pred(k, b);
                    const Kmer rv = t < 4 ? kmer_pred(rk, 3u - b) : kmer_succ(rk, 3u - b);
                    const bool r = kmer_lt(rv, fw);                          // kmer_canon: the reverse complement only if it is smaller
                    const Kmer nk = r ? rv : fw;
                    const int f = find(nk);
                    if (f < 0) un |= 1u << t;
                    else {
                        const Id id = kmer_eq(rv, fw) ? PAL : (Id)(2 * (start + (unsigned)f) + (r ? 1u : 0u));
                        if (t < 4) ns = id; else np = id;
                    }
                }
            }
            if (j < cnt) {
                const uint64_t i = start + j;
                sctx[i] = (uint8_t)c; unres[i] = (uint8_t)un;
                // only meaningful when exactly one successor / predecessor survives (then it is the last one found)
                nbr[2 * i] = (!(un & 15u) && popc4(c & 15) != 1) ? NONE : ns;
                nbr[2 * i + 1] = (!(un >> 4) && popc4(c >> 4) != 1) ? NONE : np;
            }
        }
    }
}
// the launcher: k-mers per thread from the mean chunk size (W2RAP_PL_U forces 2, 4 or 8: the parity tests run all three)
#define PL_LAUNCH(c, IdT, gl, S_, nch_, ...)                                                                                          \
    do {                                                                                                                              \
        unsigned u__ = (nch_) ? ((S_) / (nch_) > 600 ? 8u : (S_) / (nch_) > 300 ? 4u : 2u) : 2u;                                       \
        if (const char* v__ = getenv("W2RAP_PL_U")) { const int x__ = atoi(v__); if (x__ == 2 || x__ == 4 || x__ == 8) u__ = (unsigned)x__; }  \
        if (u__ == 8) LAUNCH(c, "k_prune_local", (k_prune_local<IdT, 8>), dim3(gl), dim3(256), 0, nch_, __VA_ARGS__);                  \
        else if (u__ == 4) LAUNCH(c, "k_prune_local", (k_prune_local<IdT, 4>), dim3(gl), dim3(256), 0, nch_, __VA_ARGS__);             \
        else LAUNCH(c, "k_prune_local", (k_prune_local<IdT, 2>), dim3(gl), dim3(256), 0, nch_, __VA_ARGS__);                           \
    } while (0)
// the global step: every context bit recorded in unres[i] (all set bits when unres == nullptr) is looked up in the table
template <class Id>
__global__ void __launch_bounds__(256) k_prune(uint64_t i0, uint64_t S /* k-mers [i0, S) */, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo,
                                                const uint32_t* __restrict__ scc, const Slot* __restrict__ table, uint64_t mask,
                                                uint8_t* __restrict__ sctx, Id* __restrict__ nbr, const uint8_t* __restrict__ unres) {
    constexpr Id NONE = NodeId<Id>::NONE, PAL = NodeId<Id>::PAL;
    uint64_t i = i0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    unsigned c, todo;
    // the neighbour found for each context bit is remembered (oriented node id 2*idx + reversed) so that the
    // unipath linking step does not have to probe the dictionary again
    Id ns = NONE, np = NONE;
    if (unres) {
        todo = unres[i];
        if (!todo) return;                                        // settled by k_prune_local
        if (todo == 0xFFu && sctx[i] == 0xFFu) { c = (scc[i] >> 8) & 0xFF; todo = c; }        // never visited (oversized chunk)
        else { c = sctx[i]; ns = nbr[2 * i]; np = nbr[2 * i + 1]; }
    } else { c = (scc[i] >> 8) & 0xFF; todo = c; }
    const Kmer k{shi[i], slo[i]}, rk = kmer_rc(k);                // (the reverse complement of a neighbour is a neighbour of rk: see k_prune_local)
    // one open bit per trip: a wave makes as many dependent table probes as its busiest lane has open bits (2-3), not eight
    for (unsigned rest = todo; rest;) {
        const unsigned t = (unsigned)__builtin_ctz(rest);
        rest &= rest - 1;
        const unsigned b = t & 3;
        const Kmer fw = t < 4 ? kmer_succ(k, b) : kmer_pred(k, b);
        const Kmer rv = t < 4 ? kmer_pred(rk, 3u - b) : kmer_succ(rk, 3u - b);
        const bool r = kmer_lt(rv, fw);
        const Kmer nk = r ? rv : fw;
        const int64_t s = table_find(table, mask, shi, slo, nk);
        if (s < 0) c &= ~(1u << t);
        else {
            const Id id = kmer_eq(rv, fw) ? PAL : (Id)(2 * (uint64_t)s + (r ? 1u : 0u));
            if (t < 4) ns = id; else np = id;
        }
    }
    sctx[i] = (uint8_t)c;
    // only meaningful when exactly one successor / predecessor survives (then it is the last one found)
    nbr[2 * i] = popc4(c & 15) == 1 ? ns : NONE;
    nbr[2 * i + 1] = popc4(c >> 4) == 1 ? np : NONE;
}

// =============================================================================== driver
static constexpr unsigned COUNT_CAP = 4096, COUNT_THREADS = 1024;
static constexpr unsigned KMERS_PER_BUCKET = 4500;      // (round 4: 5000 -> 4500 with k_count_fp: fewer buckets beyond its resident tile; 4000 .. 5500 are within 1 %)

// ---- K0: quality windows; sets c.M (k-mer instances of this rank's reads) and c.max_len
int count_quality(Ctx& c, uint32_t min_qual) {
    c.min_qual = min_qual;
    c.counted = false;
    hipStream_t st = c.stream;
    const uint64_t n = c.n;
    if (c.d_good) c.release(c.d_good);
    W2_ALLOC(c.d_good, uint16_t, n);
    unsigned long long* d_cnt = nullptr;                 // [0..QSLOTS) partial M  [QSLOTS..) partial max_len (as u32)
    W2_ALLOC(d_cnt, unsigned long long, 2 * QSLOTS);
    W2_HIP(hipMemsetAsync(d_cnt, 0, 2 * QSLOTS * sizeof(unsigned long long), st));
    const bool masked = c.d_qmask && c.qmask_min_qual == (int)std::min<uint32_t>(min_qual, 255u);
    if (!masked && c.quals_absent) { c.err = "quality windows: the reads were installed by a graph-only call for another min_qual, their raw qualities were not uploaded"; return W2RAP_E_STATE; }
    if (!masked) W2_TRY(quals_wait(c));                  // (another threshold than the mask was made for: the raw qualities, once they are up)
    if (n) {
        if (masked) LAUNCH(c, "k_good_len", k_good_len<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, n, c.d_quals, c.d_qmask, c.d_qoff, c.d_len, min_qual,
                           c.d_good, d_cnt, reinterpret_cast<uint32_t*>(d_cnt + QSLOTS));
        else LAUNCH(c, "k_good_len", k_good_len<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, n, c.d_quals, (const uint32_t*)nullptr, c.d_qoff, c.d_len, min_qual,
                    c.d_good, d_cnt, reinterpret_cast<uint32_t*>(d_cnt + QSLOTS));
        W2_HIP(hipGetLastError());
    }
    unsigned long long h_cnt[2 * QSLOTS];
    W2_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    c.M = 0; c.max_len = 0;
    for (unsigned i = 0; i < QSLOTS; ++i) {
        c.M += h_cnt[i];
        const uint32_t* mx = reinterpret_cast<const uint32_t*>(h_cnt + QSLOTS);
        c.max_len = std::max(c.max_len, mx[i]);
    }
    c.release(d_cnt);
    // (device arrays are not swept on the host: the longest read shows here; its good length would not fit the 16-bit words written above)
    if (c.max_len > 65535u) { c.err = "a read of more than 65,535 bases: good lengths are 16-bit words"; return W2RAP_E_LIMIT; }
    c.quality_done = true;
    return 0;
}

uint32_t default_buckets(uint64_t total_kmers, uint32_t multiple_of) {
    const char* v = getenv("W2RAP_KPB");               // tuning knob: k-mers per bucket
    uint64_t kpb = v ? (uint64_t)atoll(v) : KMERS_PER_BUCKET;
    uint64_t nb64 = total_kmers / kpb + 1;
    if (nb64 > (1u << 24)) nb64 = 1u << 24;
    if (multiple_of > 1) nb64 = (nb64 + multiple_of - 1) / multiple_of * multiple_of;
    return (uint32_t)nb64;
}

static uint32_t k1_chunk_reads() {                      // consecutive reads per wavefront of K1
    const char* v = getenv("W2RAP_K1_CHUNK");
    const int x = v ? atoi(v) : 16;
    return x > 0 ? (uint32_t)x : 16;
}


// K1 on reads [0, nr) of (boff, good): the lane-per-read kernel when a read's slots fit its LDS staging (spp * npass <= 16: reads up to
// ~315 good bases at the default 8 slots per pass), the wavefront-per-read kernel otherwise or with W2RAP_K1=wave
// (W2RAP_K1_MINW: 6 or 7 waves per SIMD instead of 8 for the latter -- 80 / 72 VGPRs, no spills -- an A/B knob)
static int launch_k1(Ctx& c, uint64_t nr, const uint64_t* boff, const uint16_t* good, uint32_t pb_lo, uint32_t pb_hi, uint32_t* bcount,
                     uint32_t nbl_part, uint32_t inv_nbl, unsigned long long* d_part, uint2* s_desc, uint32_t spp, uint32_t npass,
                     uint32_t* o_read, uint32_t* o_bkt, uint32_t* o_meta, uint32_t* o_rank, uint64_t ov_cap, unsigned long long* d_ov_cur) {
    const char* k1v = getenv("W2RAP_K1");
    const bool wave_kernel = k1v && !strcmp(k1v, "wave");
    if (!wave_kernel && spp * npass <= 16) {
        // (four waves per SIMD: with five -- the staging area of 8 slots per read would leave room -- the scatter pass beside it loses more
        // than this kernel gains, partition 20.6 -> 23.8 ms; W2RAP_K1_ALIGN64: the cuts of the wavefront-per-read kernel, for the test that
        // compares the two)
        const bool a64 = getenv("W2RAP_K1_ALIGN64") != nullptr;
        const dim3 grid((unsigned)((nr + 255) / 256));
#define W2_K1L(A64) LAUNCH(c, "k_superkmers_lane", (k_superkmers_lane<16, A64>), grid, dim3(256), 0, nr, c.d_bases, boff, good, c.NB, pb_lo, pb_hi, bcount, \
                           nbl_part, inv_nbl, d_part, s_desc, spp * npass, o_read, o_bkt, o_meta, o_rank, ov_cap, d_ov_cur)
        if (a64) W2_K1L(true); else W2_K1L(false);
#undef W2_K1L
        W2_HIP(hipGetLastError());
        return 0;
    }
    // many more blocks than fit at once (8 resident per CU with __launch_bounds__(256, 8)): the dispatcher refills freed slots,
    // so no CU waits for a straggler block (a grid of exactly "8 per CU" ran 25 ms instead of 20 -- the occupancy API answers 7,
    // the hardware admits 6, and the surplus blocks start when the others are done)
    const uint32_t k1_chunk = k1_chunk_reads();
    const unsigned grid = (unsigned)((nr + 4ull * k1_chunk - 1) / (4ull * k1_chunk) + 1);
    static const int k1_minw = getenv("W2RAP_K1_MINW") ? atoi(getenv("W2RAP_K1_MINW")) : 8;
    if (k1_minw == 6)
        LAUNCH(c, "k_superkmers", k_superkmers<6>, dim3(grid), dim3(256), 0, nr, k1_chunk, c.d_bases, boff, good, c.NB, pb_lo, pb_hi, bcount, nbl_part, inv_nbl, d_part,
               s_desc, spp, npass, o_read, o_bkt, o_meta, o_rank, ov_cap, d_ov_cur);
    else if (k1_minw == 7)
        LAUNCH(c, "k_superkmers", k_superkmers<7>, dim3(grid), dim3(256), 0, nr, k1_chunk, c.d_bases, boff, good, c.NB, pb_lo, pb_hi, bcount, nbl_part, inv_nbl, d_part,
               s_desc, spp, npass, o_read, o_bkt, o_meta, o_rank, ov_cap, d_ov_cur);
    else
        LAUNCH(c, "k_superkmers", k_superkmers<8>, dim3(grid), dim3(256), 0, nr, k1_chunk, c.d_bases, boff, good, c.NB, pb_lo, pb_hi, bcount, nbl_part, inv_nbl, d_part,
               s_desc, spp, npass, o_read, o_bkt, o_meta, o_rank, ov_cap, d_ov_cur);
    W2_HIP(hipGetLastError());
    return 0;
}

// ---- K1/K2: super-k-mer records of this rank's reads, grouped by bucket (descriptor pass, scan, scatter pass)
// (pb_lo, pb_hi): this hash-range pass keeps the records of buckets [pb_lo, pb_hi) of nb only (MapReduceEngine.h:288-299); bucket numbers
// in the outputs are relative to pb_lo, the n_parts owners divide the RANGE
int count_partition(Ctx& c, uint32_t nb, uint32_t n_parts, uint32_t pb_lo, uint32_t pb_hi) {
    if (pb_hi > nb || pb_lo >= pb_hi) { c.err = "partition: bad bucket range"; return W2RAP_E_ARG; }
    const uint32_t nbr = pb_hi - pb_lo;                  // buckets of this pass
    if (!c.quality_done) { c.err = "partition before quality_windows"; return W2RAP_E_STATE; }
    hipStream_t st = c.stream;
    const uint64_t n = c.n;
    if (n >= (1ull << 32)) { c.err = "more than 2^32 reads on one GPU (32-bit read ids in the record descriptors)"; return W2RAP_E_LIMIT; }
    c.NB = nb;
    if (c.d_bcount) c.release(c.d_bcount);
    if (c.d_bbase) c.release(c.d_bbase);
    if (c.d_recs) c.release(c.d_recs);
    W2_ALLOC(c.d_bcount, uint32_t, nbr);
    W2_ALLOC(c.d_bbase, uint64_t, (uint64_t)nbr + 1);
    unsigned long long* d_ov_cur = nullptr;
    W2_ALLOC(d_ov_cur, unsigned long long, 2);
    // multi-GPU: k-mer instances destined to each of the n_parts owners (their solid sets are bounded by it)
    unsigned long long* d_part = nullptr;
    if (n_parts > 64 || (n_parts && nbr % n_parts)) { c.err = "partition: at most 64 parts, dividing the bucket count"; return W2RAP_E_LIMIT; }
    if (n_parts) W2_ALLOC(d_part, unsigned long long, 64 * 64);
    const uint32_t nbl_part = n_parts ? nbr / n_parts : 0, inv_nbl = nbl_part > 1 ? (uint32_t)((1ull << 32) / nbl_part) : 0;
    // descriptor slots: spp per (read, pass of 128 k-mer positions).  A pass of a PE150 read cuts into ~4 records,
    // 8 slots hold all but ~1 % of them; the surplus goes to the overflow list.  If even that list is too small the
    // pass is repeated with twice the slots (128 cannot overflow).
    const char* sv = getenv("W2RAP_SPP");
    uint32_t spp = sv ? (uint32_t)atoi(sv) : 8;
    const uint32_t npass = c.max_len > K + 127 ? (c.max_len - (K - 1) + 127) / 128 : 1;
    uint2* s_desc = nullptr; uint32_t *o_read = nullptr, *o_bkt = nullptr, *o_meta = nullptr, *o_rank = nullptr;
    uint64_t ov_cap = n / 8 + 1024;
    W2_ALLOC(o_read, uint32_t, ov_cap); W2_ALLOC(o_bkt, uint32_t, ov_cap); W2_ALLOC(o_meta, uint32_t, ov_cap); W2_ALLOC(o_rank, uint32_t, ov_cap);
    uint64_t nslots = 0, nov = 0;
    for (;;) {
        nslots = n * npass * spp;
        W2_ALLOC(s_desc, uint2, nslots);
        W2_HIP(hipMemsetAsync(c.d_bcount, 0, (size_t)nbr * 4, st));
        W2_HIP(hipMemsetAsync(d_ov_cur, 0, 16, st));
        if (d_part) W2_HIP(hipMemsetAsync(d_part, 0, 64 * 64 * 8, st));
        if (n) W2_TRY(launch_k1(c, n, c.d_boff, c.d_good, pb_lo, pb_hi, c.d_bcount, nbl_part, inv_nbl, d_part, s_desc, spp, npass, o_read, o_bkt, o_meta, o_rank, ov_cap, d_ov_cur));
        W2_TRY(exclusive_scan_u32_to_u64(c, c.d_bcount, c.d_bbase, nbr));
        unsigned long long h_ov = 0;
        W2_HIP(hipMemcpyAsync(&c.nrec, c.d_bbase + nbr, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipMemcpyAsync(&h_ov, d_ov_cur, 8, hipMemcpyDeviceToHost, st));
        unsigned long long h_part[64 * 64];
        if (d_part) W2_HIP(hipMemcpyAsync(h_part, d_part, sizeof(h_part), hipMemcpyDeviceToHost, st));
        W2_HIP(hipStreamSynchronize(st));
        if (d_part) for (unsigned g = 0; g < 64; ++g) { c.part_kmers[g] = 0; for (unsigned k = 0; k < 64; ++k) c.part_kmers[g] += h_part[k * 64 + g]; }
        nov = h_ov;
        if (nov <= ov_cap) break;
        // the list was too small: its exact need is known now (records beyond rank 65535 of a heavy bucket do not depend on
        // spp; with many more entries than reads the slots are too few as well)
        c.release(s_desc);
        if (nov > n && spp < 128) spp *= 2;
        for (uint32_t* q : {o_read, o_bkt, o_meta, o_rank}) c.release(q);
        ov_cap = nov + nov / 8 + 1024;
        W2_ALLOC(o_read, uint32_t, ov_cap); W2_ALLOC(o_bkt, uint32_t, ov_cap); W2_ALLOC(o_meta, uint32_t, ov_cap); W2_ALLOC(o_rank, uint32_t, ov_cap);
    }
    W2_ALLOC(c.d_recs, uint32_t, c.nrec * REC_DWORDS);
    if (c.nrec) {
        uint64_t bases_bytes = 0;
        W2_HIP(hipMemcpy(&bases_bytes, c.d_boff + n, 8, hipMemcpyDeviceToHost));
        const uint64_t nthreads = nslots + nov;
        LAUNCH(c, "k_scatter_records", k_scatter_records, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, nslots, npass * spp, s_desc,
               nov, o_read, o_bkt, o_meta, o_rank, c.d_bases, c.d_boff, bases_bytes, c.d_bbase, c.d_recs);
        W2_HIP(hipGetLastError());
    }
    W2_HIP(hipStreamSynchronize(st));
    c.release(d_ov_cur); if (d_part) c.release(d_part); c.release(s_desc); c.release(o_read); c.release(o_bkt); c.release(o_meta); c.release(o_rank);
    return 0;
}

// ---- K1/K2 in batches of reads (single GPU).  K1 is bound by instruction issue, K2 by device atomics and scattered stores:
// batch k's records are scattered on the side stream while batch k+1 is being cut.  Every batch becomes one SEGMENT of
// records (grouped by bucket, segments back to back) -- the layout K3 already consumes for the records of several source
// ranks -- so nothing is merged: c.d_bcount is [n_batches][NB], c.d_recs the segments, *n_seg the number of batches.
int count_partition_batched(Ctx& c, uint32_t nb, unsigned n_batches, unsigned* n_seg, uint32_t pb_lo, uint32_t pb_hi) {
    const uint32_t nbl = pb_hi - pb_lo;                  // buckets of this pass: c.d_bcount is [n_batches][nbl]
    if (!c.quality_done) { c.err = "partition before quality_windows"; return W2RAP_E_STATE; }
    hipStream_t st = c.stream, st2 = c.stream2;
    const uint64_t n = c.n;
    if (n >= (1ull << 32)) { c.err = "more than 2^32 reads on one GPU (32-bit read ids in the record descriptors)"; return W2RAP_E_LIMIT; }
    if (n_batches < 1) n_batches = 1; if (n_batches > 16) n_batches = 16;
    if (n < (1u << 20) || !st2) n_batches = 1;
    c.NB = nb;
    if (c.d_bcount) c.release(c.d_bcount);
    if (c.d_bbase) { c.release(c.d_bbase); c.d_bbase = nullptr; }
    if (c.d_recs) c.release(c.d_recs);
    c.d_recs = nullptr;
    W2_ALLOC(c.d_bcount, uint32_t, (uint64_t)n_batches * nbl);
    uint64_t* d_bbase[2] = {nullptr, nullptr};
    uint2* s_desc[2] = {nullptr, nullptr};
    uint32_t *o_read[2] = {nullptr, nullptr}, *o_bkt[2] = {nullptr, nullptr}, *o_meta[2] = {nullptr, nullptr}, *o_rank[2] = {nullptr, nullptr};
    unsigned long long* d_ov_cur = nullptr;
    W2_ALLOC(d_ov_cur, unsigned long long, 2);
    const char* sv = getenv("W2RAP_SPP");
    uint32_t spp = sv ? (uint32_t)atoi(sv) : 8;
    const uint32_t npass = c.max_len > K + 127 ? (c.max_len - (K - 1) + 127) / 128 : 1;
    // equal batches (W2RAP_LAST_BATCH < 1: a shorter last one, whose scatter pass is the one nothing hides -- measured in round 4: no difference)
    const double last_frac = getenv("W2RAP_LAST_BATCH") ? std::min(1.0, std::max(0.1, atof(getenv("W2RAP_LAST_BATCH")))) : 1.0;
    const uint64_t per_batch = n_batches > 1 ? (((uint64_t)((double)n / ((double)n_batches - 1.0 + last_frac)) + 2) & ~1ull) : ((n + 1) & ~1ull);
    uint64_t ov_cap[2] = {per_batch / 8 + 1024, per_batch / 8 + 1024};
    uint64_t slots_alloc[2] = {0, 0};
    for (int b = 0; b < 2; ++b) {
        W2_ALLOC(d_bbase[b], uint64_t, (uint64_t)nbl + 1);
        W2_ALLOC(o_read[b], uint32_t, ov_cap[b]); W2_ALLOC(o_bkt[b], uint32_t, ov_cap[b]); W2_ALLOC(o_meta[b], uint32_t, ov_cap[b]);
        W2_ALLOC(o_rank[b], uint32_t, ov_cap[b]);
    }
    uint64_t bases_bytes = 0;
    if (n) W2_HIP(hipMemcpy(&bases_bytes, c.d_boff + n, 8, hipMemcpyDeviceToHost));
    hipEvent_t ev_k2[16] = {};
    uint64_t rec_cap = 0, seg_base = 0;
    c.nrec = 0;
    unsigned nseg = 0;
    for (unsigned k = 0; k < n_batches; ++k) {
        const uint64_t r0 = std::min<uint64_t>(n, k * per_batch), r1 = std::min<uint64_t>(n, r0 + per_batch), nr = r1 - r0;
        const int b = k & 1;
        uint32_t* bcount = c.d_bcount + (uint64_t)k * nbl;
        if (k >= 2) W2_HIP(hipEventSynchronize(ev_k2[k - 2]));            // the scatter that read these double buffers is done
        uint64_t nslots = 0, nov = 0, nrec_k = 0;
        for (;;) {
            nslots = nr * npass * spp;
            if (slots_alloc[b] < nslots) { if (s_desc[b]) c.release(s_desc[b]); W2_ALLOC(s_desc[b], uint2, nslots); slots_alloc[b] = nslots; }
            W2_HIP(hipMemsetAsync(bcount, 0, (size_t)nbl * 4, st));
            W2_HIP(hipMemsetAsync(d_ov_cur, 0, 16, st));
            if (nr) W2_TRY(launch_k1(c, nr, c.d_boff + r0, c.d_good + r0, pb_lo, pb_hi, bcount, 0u, 0u, nullptr, s_desc[b], spp, npass, o_read[b], o_bkt[b], o_meta[b], o_rank[b], ov_cap[b], d_ov_cur));
            W2_TRY(exclusive_scan_u32_to_u64(c, bcount, d_bbase[b], nbl));
            unsigned long long h_ov = 0;
            W2_HIP(hipMemcpyAsync(&nrec_k, d_bbase[b] + nbl, 8, hipMemcpyDeviceToHost, st));
            W2_HIP(hipMemcpyAsync(&h_ov, d_ov_cur, 8, hipMemcpyDeviceToHost, st));
            W2_HIP(hipStreamSynchronize(st));
            nov = h_ov;
            if (nov <= ov_cap[b]) break;
            // the list was too small: its exact need is known now (see count_partition); this batch's buffers are free to grow
            if (nov > nr && spp < 128) spp *= 2;                           // (this and the later batches; earlier ones keep their slots)
            for (uint32_t* q : {o_read[b], o_bkt[b], o_meta[b], o_rank[b]}) c.release(q);
            ov_cap[b] = nov + nov / 8 + 1024;
            W2_ALLOC(o_read[b], uint32_t, ov_cap[b]); W2_ALLOC(o_bkt[b], uint32_t, ov_cap[b]); W2_ALLOC(o_meta[b], uint32_t, ov_cap[b]);
            W2_ALLOC(o_rank[b], uint32_t, ov_cap[b]);
        }
        // room for this segment: the first batch predicts the total (batches are equal samples of the reads)
        if (seg_base + nrec_k > rec_cap) {
            const uint64_t want = k == 0 ? (uint64_t)((double)nrec_k * ((double)n / (double)std::max<uint64_t>(nr, 1))) + nrec_k / 16 + (1u << 20) : (seg_base + nrec_k) * 2;
            uint32_t* bigger = c.alloc<uint32_t>(want * REC_DWORDS);
            if (!bigger) return W2RAP_E_HIP;
            if (c.d_recs) {
                W2_HIP(hipStreamSynchronize(st2));
                W2_HIP(hipMemcpyAsync(bigger, c.d_recs, seg_base * REC_BYTES, hipMemcpyDeviceToDevice, st2));
                W2_HIP(hipStreamSynchronize(st2));
                c.release(c.d_recs);
            }
            c.d_recs = bigger; rec_cap = want;
        }
        hipStream_t sk = n_batches > 1 ? st2 : st;
        if (nrec_k) {
            const uint64_t nthreads = nslots + nov;
            LAUNCH_ON(c, sk, "k_scatter_records", k_scatter_records, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, nslots, npass * spp, s_desc[b],
                      nov, o_read[b], o_bkt[b], o_meta[b], o_rank[b], c.d_bases, c.d_boff + r0, bases_bytes, d_bbase[b], c.d_recs + seg_base * REC_DWORDS);
            W2_HIP(hipGetLastError());
        }
        W2_HIP(hipEventCreateWithFlags(&ev_k2[k], hipEventDisableTiming));
        W2_HIP(hipEventRecord(ev_k2[k], sk));
        seg_base += nrec_k;
        nseg = k + 1;
    }
    for (unsigned k = 0; k < nseg; ++k) { W2_HIP(hipEventSynchronize(ev_k2[k])); (void)hipEventDestroy(ev_k2[k]); }
    c.nrec = seg_base;
    if (!c.d_recs) W2_ALLOC(c.d_recs, uint32_t, REC_DWORDS);
    for (int b = 0; b < 2; ++b) {
        c.release(d_bbase[b]); if (s_desc[b]) c.release(s_desc[b]);
        c.release(o_read[b]); c.release(o_bkt[b]); c.release(o_meta[b]); c.release(o_rank[b]);
    }
    c.release(d_ov_cur);
    *n_seg = nseg;
    return 0;
}

// ---- K3: count `nbl` buckets whose records arrive in `nseg` segments (one per source rank), each
// segment grouped by bucket; d_counts[s*nbl + b] = records of bucket b in segment s, d_recs = the
// segments back to back.  total_kmers bounds the solid set (S <= kmers / min_freq).
// lookup-table geometry for S solid k-mers: 4 slots per k-mer (load <= 0.25: ~1.3 probes per miss instead of ~2.3)
static void table_geometry(uint64_t S, uint64_t& tcap) {
    const char* lf = getenv("W2RAP_TABLE_X");
    // beyond 2^30 k-mers the table is kept at 1.3 slots per k-mer at least (a power of two: load 0.38 .. 0.77) -- at 2.5 G solid
    // k-mers (BASELINE configs[2] replicated, capacity estimate + 15 %) that is 34 GB instead of 137
    const uint64_t mult10 = lf ? 10 * (uint64_t)atoll(lf) : (S < (1ull << 30) ? 40 : 13);
    tcap = 1024;
    while (10 * tcap < mult10 * S) tcap <<= 1;
    // (there is no per-k-mer absence filter in front of it any more -- a second atomic per k-mer in K4: read pathing proves
    // absence through the 31-mer filter built with the graph, step2_graph.hip k_filter32)
}

// The buckets are counted in NS launches (slices of the bucket range); after each one the running totals (solid k-mers,
// chunks) are copied to pinned memory and an event is recorded, so that a caller can consume slice k -- insert its solid
// k-mers into the lookup table on the side stream (single GPU), or exchange them with the other ranks (multi-GPU) -- while
// slice k+1 is being counted: the insert kernel is bound by device atomics and leaves the SIMDs idle, the counting kernel is
// bound by instruction issue and leaves the memory system idle.
int count_buckets_launch(Ctx& c, uint32_t min_freq, uint32_t nbl, uint32_t nseg, const uint32_t* d_recs, const uint32_t* d_counts,
                         uint64_t total_kmers, unsigned NS, bool deferred) {
    hipStream_t st = c.stream;
    if (c.cs_planned) { c.err = "count_records_begin: a sliced count is already pending"; return W2RAP_E_STATE; }
    c.min_freq = min_freq;
    if (!c.pass) c.table_built = false;                  // (a later pass goes on filling the owner's own dictionary of the earlier ones: local_dict_slice)
    if (NS < 1) NS = 1; if (NS > 16) NS = 16;
    if (nbl < 4096) NS = 1;
    if (nseg > 64) { c.err = "count_records: more than 64 segments"; return W2RAP_E_LIMIT; }
    const uint64_t nflat = (uint64_t)nbl * nseg;
    uint64_t* d_off = nullptr;
    W2_ALLOC(d_off, uint64_t, nflat + 1);
    W2_TRY(exclusive_scan_u32_to_u64(c, d_counts, d_off, nflat));
    // A later pass of a multi-pass count (c.pass > 0) appends to the solid arrays, the chunk list, the counters and the histogram of
    // the passes before it; everything else is this pass's own.
    unsigned long long* d_cnt = c.pass ? c.pass_cnt : nullptr;   // [4..7] counters  [8..108] hist  [144] queue (a line of its own)
    uint32_t chunk_cap = c.pass ? c.cs_chunk_cap : 0;
    if (!c.pass) {
        W2_ALLOC(d_cnt, unsigned long long, 160);
        W2_HIP(hipMemsetAsync(d_cnt, 0, 160 * sizeof(unsigned long long), st));
        c.solid_cap = total_kmers / (min_freq ? min_freq : 1) + 1;
        for (void* p : {(void*)c.d_shi, (void*)c.d_slo, (void*)c.d_scc}) if (p) c.release(p);
        W2_ALLOC(c.d_shi, uint64_t, c.solid_cap);
        W2_ALLOC(c.d_slo, uint64_t, c.solid_cap);
        W2_ALLOC(c.d_scc, uint32_t, c.solid_cap);
        // chunk list for the bucket-local prune and the chunk-local list ranking (multi-GPU: exchanged with the solid k-mers)
        if (c.d_chunk_start) { c.release(c.d_chunk_start); c.release(c.d_chunk_cnt); c.d_chunk_start = nullptr; c.d_chunk_cnt = nullptr; }
        c.nchunks = 0;
        if (!getenv("W2RAP_NO_LOCAL_PRUNE")) {
            const uint64_t nb_all = c.npass > 1 ? c.NB : nbl;                      // all passes share the list
            chunk_cap = (uint32_t)std::min<uint64_t>(nb_all * 2 + 4096, 1u << 24);
            W2_ALLOC(c.d_chunk_start, uint64_t, chunk_cap);
            W2_ALLOC(c.d_chunk_cnt, uint32_t, chunk_cap);
            W2_HIP(hipMemsetAsync(c.d_chunk_cnt, 0, (size_t)chunk_cap * 4, st));
        }
    }
    c.cs_cnt = d_cnt; c.cs_off = d_off; c.cs_chunk_cap = chunk_cap;
    W2_ALLOC(c.cs_defer, uint32_t, (uint64_t)nbl + 2);
    c.cs_planned = NS; c.cs_ns = 0; c.cs_nbl = nbl; c.cs_nseg = nseg; c.cs_recs = d_recs;
    c.cs_short_first = deferred && NS >= 3;        // deferred: slice 0's records are exchanged before anything can be counted -- keep it short
    if (!deferred) for (unsigned k = 0; k < NS; ++k) W2_TRY(count_buckets_launch_slice(c, k));
    return 0;
}

// bucket range of slice k: equal slices, or (cs_short_first) a first slice of half the size of the others
void count_slice_bounds(const Ctx& c, unsigned k, uint32_t* b_lo, uint32_t* b_hi) {
    const uint64_t NS = c.cs_planned, nbl = c.cs_nbl;
    auto cut = [&](uint64_t j) -> uint32_t {
        if (!c.cs_short_first) return (uint32_t)(nbl * j / NS);
        return (uint32_t)(j == 0 ? 0 : nbl * (2 * j - 1) / (2 * NS - 1));       // weights 1, 2, 2, ..
    };
    *b_lo = cut(k); *b_hi = cut(k + 1);
}

// launches bucket slice k of a prepared count (slices go in order; deferred mode: the caller launches slice k once its
// records have arrived)
int count_buckets_launch_slice(Ctx& c, unsigned k) {
    hipStream_t st = c.stream;
    if (!c.cs_planned || k != c.cs_ns || k >= c.cs_planned) { c.err = "count_records_launch: slices are launched once each, in order"; return W2RAP_E_STATE; }
    const unsigned NS = c.cs_planned;
    const uint32_t nbl = c.cs_nbl, nseg = c.cs_nseg;
    unsigned long long* d_cnt = c.cs_cnt;
    // (the bucket queue in a 128-byte line of its own: the counters behind d_cnt + 4 take one same-address atomic per bucket too -- its output range --,
    //  and two hot words in one line share one L2 channel's atomic unit)
#ifndef W2RAP_QUEUE_AT
#define W2RAP_QUEUE_AT 144
#endif
    uint32_t* d_queue = reinterpret_cast<uint32_t*>(d_cnt + W2RAP_QUEUE_AT);
    uint32_t b_lo, b_hi;
    count_slice_bounds(c, k, &b_lo, &b_hi);
    auto slice_done = [&]() -> int {
        W2_HIP(hipMemcpyAsync(c.h_pinned + k, d_cnt + 4, 8, hipMemcpyDeviceToHost, st));
        W2_HIP(hipEventCreateWithFlags(&c.cs_ev[k], hipEventDisableTiming));
        W2_HIP(hipEventRecord(c.cs_ev[k], st));
        c.cs_ns = k + 1;
        return 0;
    };
    auto launch = [&](auto kern, unsigned lds, unsigned threads, unsigned blocks_per_cu) -> int {
        W2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        unsigned grid = (unsigned)std::min<uint64_t>(b_hi - b_lo, (uint64_t)c.sm_count * blocks_per_cu);
        if (k || c.pass) W2_HIP(hipMemsetAsync(d_queue, 0, 4, st));
        LAUNCH(c, "k_count_buckets", kern, dim3(grid ? grid : 1), dim3(threads), lds, nbl, b_lo, b_hi, nseg, c.cs_off, c.cs_recs, c.min_freq, d_queue,
               c.d_shi, c.d_slo, c.d_scc, c.solid_cap, d_cnt + 4, d_cnt + 8, c.d_chunk_start, c.d_chunk_cnt, c.cs_chunk_cap, (const uint32_t*)nullptr, 0u);
        W2_HIP(hipGetLastError());
        return slice_done();
    };
    // the round-4 shape: fingerprint + reference slots over the bucket's resident records, two (or more) blocks per CU; what it defers
    // (a bucket beyond its tile or table) is counted behind it by the round-1..3 kernel in list mode
    auto launch_fp = [&](auto kern, unsigned lds, unsigned threads, unsigned blocks_per_cu, uint32_t fp_limit, uint32_t fp_sc) -> int {
        W2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        auto list_kern = k_count_buckets<COUNT_CAP, COUNT_THREADS>;
        W2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(list_kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)K3Cfg<COUNT_CAP, COUNT_THREADS>::LDS));
        const unsigned grid = (unsigned)std::min<uint64_t>(b_hi - b_lo, (uint64_t)c.sm_count * blocks_per_cu);
        if (k || c.pass) W2_HIP(hipMemsetAsync(d_queue, 0, 4, st));
        if (k == 0) W2_HIP(hipMemsetAsync(c.cs_defer, 0, 8, st));                 // the deferred buckets of ALL slices collect in one list ...
        // (test hooks: a tiny table / staging area sends nearly every bucket through the hash-class path)
        uint32_t fill_limit = fp_limit, solid_limit = fp_sc;
        if (const char* t = getenv("W2RAP_TEST_FP_LIMIT")) { if (test_hook("W2RAP_TEST_FP_LIMIT")) fill_limit = std::min<uint32_t>(fill_limit, (uint32_t)std::max(1, atoi(t))); }
        if (const char* t = getenv("W2RAP_TEST_FP_SC")) { if (test_hook("W2RAP_TEST_FP_SC")) solid_limit = std::min<uint32_t>(solid_limit, (uint32_t)std::max(1, atoi(t))); }
        LAUNCH(c, "k_count_fp", kern, dim3(grid ? grid : 1), dim3(threads), lds, nbl, b_lo, b_hi, nseg, c.cs_off, c.cs_recs, c.min_freq, d_queue,
               c.d_shi, c.d_slo, c.d_scc, c.solid_cap, d_cnt + 4, d_cnt + 8, c.d_chunk_start, c.d_chunk_cnt, c.cs_chunk_cap, c.cs_defer, nbl, fill_limit, solid_limit,
               c.fused_prune ? c.d_sctx : (uint8_t*)nullptr, c.fused_prune ? c.d_unres : (uint8_t*)nullptr, c.fused_prune ? (uint32_t*)c.d_nbr : (uint32_t*)nullptr);
        W2_HIP(hipGetLastError());
        if (k + 1 == NS && c.fused_prune) W2_HIP(hipMemcpyAsync(c.h_pinned + 20, d_cnt + 4, 8, hipMemcpyDeviceToHost, st));      // chunks so far: the list kernel's come behind
        if (k + 1 == NS) {
            // ... which the list kernel counts behind the last slice: one launch with every block busy instead of one thin launch per slice
            // (its buckets are the heavy ones: several tiles, hash classes -- 50-80 us each)
            const unsigned lgrid = (unsigned)std::min<uint64_t>(nbl, (uint64_t)c.sm_count);
            constexpr unsigned list_lds = K3Cfg<COUNT_CAP, COUNT_THREADS>::LDS;
            LAUNCH(c, "k_count_buckets", list_kern, dim3(lgrid ? lgrid : 1), dim3(COUNT_THREADS), list_lds, nbl, 0u, nbl, nseg, c.cs_off,
                   c.cs_recs, c.min_freq, c.cs_defer + 1, c.d_shi, c.d_slo, c.d_scc, c.solid_cap, d_cnt + 4, d_cnt + 8, c.d_chunk_start, c.d_chunk_cnt, c.cs_chunk_cap,
                   (const uint32_t*)c.cs_defer, nbl);
            W2_HIP(hipGetLastError());
        }
        return slice_done();
    };
    const char* v = getenv("W2RAP_K3");            // tuning knob: table/block shape
    int cfg = v ? atoi(v) : 22;
    if (cfg >= 20 && nseg > FpCfg<512, 512, 1024>::MAXSEG) cfg = 0;
    // (threads, min waves per SIMD, resident records, solid k-mers per bucket): two 512-thread blocks per CU is the shipped shape.  Default
    // 22 since K1 cuts 17 % fewer records (a 4500-k-mer bucket holds ~240): 512 resident records are enough -- 0.2 % of the buckets go to
    // the list kernel instead of 0.004 % -- and k_count_fp is 0.7 ms faster than with shape 20's 640 (three alternating runs, and the
    // planted workload within 0.2 ms: profiles/r04_sched_ab.txt)
    if (cfg == 20) W2_TRY(launch_fp(k_count_fp<512, 4, 640, 768>, FpCfg<512, 640, 768>::LDS, 512, 2, FpCfg<512, 640, 768>::LIMIT, 768));
    else if (cfg == 21) W2_TRY(launch_fp(k_count_fp<1024, 8, 576, 512>, FpCfg<1024, 576, 512>::LDS, 1024, 2, FpCfg<1024, 576, 512>::LIMIT, 512));
    else if (cfg == 22) W2_TRY(launch_fp(k_count_fp<512, 4, 512, 1024>, FpCfg<512, 512, 1024>::LDS, 512, 2, FpCfg<512, 512, 1024>::LIMIT, 1024));
    else if (cfg == 23) W2_TRY(launch_fp(k_count_fp<1024, 4, 576, 1024>, FpCfg<1024, 576, 1024>::LDS, 1024, 1, FpCfg<1024, 576, 1024>::LIMIT, 1024));
    else if (cfg == 24) W2_TRY(launch_fp(k_count_fp<512, 4, 576, 1024>, FpCfg<512, 576, 1024>::LDS, 512, 2, FpCfg<512, 576, 1024>::LIMIT, 1024));
    else if (cfg == 1) W2_TRY(launch(k_count_buckets<2048, 512>, K3Cfg<2048, 512>::LDS, 512, 2));
    else if (cfg == 2) W2_TRY(launch(k_count_buckets<4096, 512>, K3Cfg<4096, 512>::LDS, 512, 1));
    else if (cfg == 3) W2_TRY(launch(k_count_buckets<1024, 256>, K3Cfg<1024, 256>::LDS, 256, 4));
    // round-3 experiments (profiles/r03_k3_variants.txt; results identical, times at 50 M reads against 50.3 ms for the shipped form):
    //   5: no parking queue, 2048 slots, 256-record tiles -> 62 KB of LDS, 64 VGPRs (39 spilled): TWO blocks per CU, eight waves per SIMD: 69.2 ms
    //   8: the shipped shape without the parking queue (collisions probed at once): 61.9 ms
    else if (cfg == 5) W2_TRY(launch(k_count_buckets<2048, 1024, false, false, 8>, K3Cfg<2048, 1024, false>::LDS, 1024, 2));
    else if (cfg == 8) W2_TRY(launch(k_count_buckets<4096, 1024, false, false, 1>, K3Cfg<4096, 1024, false>::LDS, 1024, 1));
    else if (cfg == 9) W2_TRY(launch(k_count_buckets<COUNT_CAP, COUNT_THREADS, true>, K3Cfg<COUNT_CAP, COUNT_THREADS>::LDS, COUNT_THREADS, 1));
    else W2_TRY(launch(k_count_buckets<COUNT_CAP, COUNT_THREADS>, K3Cfg<COUNT_CAP, COUNT_THREADS>::LDS, COUNT_THREADS, 1));
    return 0;
}

// waits for slice k; -> solid k-mers / chunks emitted by slices 0..k (all of them written)
int count_buckets_slice(Ctx& c, unsigned k, uint64_t* n_solid, uint64_t* n_chunks) {
    if (k >= c.cs_ns) { c.err = "count_records_slice: slice not launched"; return W2RAP_E_ARG; }
    W2_HIP(hipEventSynchronize(c.cs_ev[k]));
    const uint64_t w = c.h_pinned[k];
    if (n_solid) *n_solid = std::min<uint64_t>(w & ((1ull << 40) - 1), c.solid_cap);
    if (n_chunks) *n_chunks = std::min<uint64_t>(w >> 40, c.cs_chunk_cap);
    return 0;
}

int count_buckets_finish(Ctx& c) {
    hipStream_t st = c.stream;
    if (!c.cs_planned || c.cs_ns != c.cs_planned) { c.err = "count_records_end: not every slice has been launched"; return W2RAP_E_STATE; }
    unsigned long long h_all[160];
    W2_HIP(hipMemcpyAsync(h_all, c.cs_cnt, sizeof(h_all), hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    for (unsigned k = 0; k < c.cs_ns; ++k) (void)hipEventDestroy(c.cs_ev[k]);
    c.cs_ns = 0; c.cs_planned = 0;
    if (c.pass + 1 < c.npass) c.pass_cnt = c.cs_cnt;                     // the next pass goes on counting into these
    else { c.release(c.cs_cnt); c.pass_cnt = nullptr; c.pass = 0; c.npass = 1; }
    c.release(c.cs_off); c.cs_cnt = nullptr; c.cs_off = nullptr;
    if (getenv("W2RAP_TRACE") && c.cs_defer) {
        uint32_t nd = 0;
        (void)hipMemcpy(&nd, c.cs_defer, 4, hipMemcpyDeviceToHost);
        fprintf(stderr, "[w2rap] k_count_fp deferred %u buckets (all slices) to k_count_buckets\n", nd);
    }
    if (c.cs_defer) { c.release(c.cs_defer); c.cs_defer = nullptr; }
    if (getenv("W2RAP_TRACE") && h_all[111])
        fprintf(stderr, "[w2rap] k_count_buckets wave-0 clocks per block: stage-in %.0f, count %.0f, barrier A %.0f, flush+scan %.0f, barrier B %.0f, staging %.0f; slowest wave's count %.0f (x%u blocks, %u buckets)\n",
                (double)h_all[110] / c.sm_count, (double)h_all[111] / c.sm_count, (double)h_all[112] / c.sm_count, (double)h_all[113] / c.sm_count,
                (double)h_all[114] / c.sm_count, (double)h_all[115] / c.sm_count, (double)h_all[116] / c.sm_count, (unsigned)c.sm_count, 0u);
    if (getenv("W2RAP_TRACE") && h_all[128]) {
        fprintf(stderr, "[w2rap] k_count_buckets count-phase clocks per block, by wave:");
        for (int w = 0; w < 16; ++w) fprintf(stderr, " %.1fM", (double)h_all[128 + w] / c.sm_count / 1e6);
        fprintf(stderr, "\n");
    }
    if (getenv("W2RAP_TRACE") && h_all[121])
        fprintf(stderr, "[w2rap] k_count_buckets wave-0 clocks per window: loads %.0f, extract+hash %.0f, probe %.0f, atomics %.0f (%.0f windows per block)\n",
                (double)h_all[117] / h_all[121], (double)h_all[118] / h_all[121], (double)h_all[119] / h_all[121], (double)h_all[120] / h_all[121],
                (double)h_all[121] / c.sm_count);
    if (h_all[7]) { c.err = "k_count_buckets: a bucket did not fit the LDS table after 2^16-way splitting"; return W2RAP_E_LIMIT; }
    c.S = h_all[4] & ((1ull << 40) - 1); c.D = h_all[5];
    c.nchunks = c.cs_chunk_cap ? std::min<uint64_t>(h_all[4] >> 40, c.cs_chunk_cap) : 0;
    for (int i = 0; i < 101; ++i) c.hist[i] = h_all[8 + i];
    if (c.S > c.solid_cap) { c.err = "solid k-mer count exceeds its bound"; return W2RAP_E_LIMIT; }
    return 0;
}

// lookup-table + absence-filter storage for `S` solid k-mers, cleared on stream `on`
static int table_alloc(Ctx& c, uint64_t S, hipStream_t on) {
    uint64_t tcap;
    table_geometry(S, tcap);
    c.tcap = tcap;
    W2_ALLOC(c.d_table, Slot, tcap);
    W2_HIP(hipMemsetAsync(c.d_table, 0xFF, tcap * sizeof(Slot), on));
    return 0;
}

// single-GPU counting: all buckets of this GPU; with build_table the solid k-mers of a finished slice are inserted into the
// lookup table on the side stream while the next slice counts.  The table is sized from the first slice (buckets are
// hash-uniform, so S ~ NS * S_1); if the guess turns out too small the table is rebuilt the plain way.
int count_buckets(Ctx& c, uint32_t min_freq, uint32_t nbl, uint32_t nseg, const uint32_t* d_recs, const uint32_t* d_counts,
                  uint64_t total_kmers, bool build_table) {
    hipStream_t st2 = c.stream2;
    const char* nsv = getenv("W2RAP_SLICES");
    unsigned NS = (build_table && !getenv("W2RAP_NO_OVERLAP") && nbl >= 4096 && st2) ? (nsv ? (unsigned)atoi(nsv) : 4) : 1;
    // The chunk-local adjacency prune inside k_count_fp's emit (W2RAP_FUSED_PRUNE=1): only where the k-mer numbering of this count is final
    // (one GPU, one pass: build_table), node ids are 32-bit words, and the round-4 kernel counts (its deferred buckets, which the list kernel
    // counts, keep k_prune_local).  The prune's arrays exist before the first launch then, sized by the bound of the solid set.
    {
        const char* fv = getenv("W2RAP_FUSED_PRUNE"); const char* kv = getenv("W2RAP_K3"); const char* wv = getenv("W2RAP_WIDE_IDS");
        const uint64_t cap = total_kmers / (min_freq ? min_freq : 1) + 1;
        c.fused_prune = build_table && fv && atoi(fv) == 1 && !(kv && atoi(kv) < 20) && !(wv && atoi(wv) != 0) && cap < (1ull << 31) - 1 && c.npass == 1 &&
                        nseg <= 16 && !getenv("W2RAP_NO_LOCAL_PRUNE");
        c.unfused_chunks.clear();
        if (c.fused_prune) {
            for (void* p : {(void*)c.d_sctx, (void*)c.d_nbr, (void*)c.d_unres}) if (p) c.release(p);
            W2_ALLOC(c.d_sctx, uint8_t, cap); W2_ALLOC(c.d_unres, uint8_t, cap);
            uint32_t* nb32 = nullptr; W2_ALLOC(nb32, uint32_t, 2 * cap); c.d_nbr = nb32;
            W2_HIP(hipMemsetAsync(c.d_sctx, 0xFF, cap, c.stream));      // unvisited k-mers (the list kernel's chunks until k_prune_local): every bit open
            W2_HIP(hipMemsetAsync(c.d_unres, 0xFF, cap, c.stream));
        }
    }
    // (Round 5, measured and removed: the bucket-local prune slice by slice on the side stream, under the counting of the next slice --
    //  count phase 70.2 -> 77.7 ms: like the table inserts, it takes from k_count_fp more than it saves behind it.)
    W2_TRY(count_buckets_launch(c, min_freq, nbl, nseg, d_recs, d_counts, total_kmers, NS, false));
    NS = c.cs_ns;
    uint64_t s_cap = 0;
    if (NS > 1) {
        uint64_t s_prev = 0;
        for (unsigned k = 0; k < NS; ++k) {
            uint64_t s_k = 0;
            W2_TRY(count_buckets_slice(c, k, &s_k, nullptr));
            if (k == 0) {
                s_cap = s_k * NS + s_k / 2 + 1024;          // 12 % head room over the extrapolation
                if (test_hook("W2RAP_TEST_SMALL_SCAP")) s_cap = s_k + 1;   // test hook: make the extrapolation fail
                W2_TRY(table_alloc(c, s_k * NS, st2));      // the table itself is laid out for the extrapolation (a power of two)
            }
            const uint64_t s_hi = s_k < s_cap ? s_k : s_cap;
            if (s_hi > s_prev) {
                LAUNCH_ON(c, st2, "k_table_insert", k_table_insert, dim3((unsigned)((s_hi - s_prev + 255) / 256)), dim3(256), 0, s_prev, s_hi,
                          c.d_shi, c.d_slo, c.d_table, c.tcap - 1);
                W2_HIP(hipGetLastError());
                s_prev = s_hi;
            }
        }
    }
    // the last slice's insert is still running on the side stream: the bucket-local prune does not need the table and
    // runs beside it (count_table waits for the side stream before the first global probe)
    W2_TRY(count_buckets_finish(c));
    if (c.fused_prune) c.unfused_chunks.emplace_back(std::min<uint64_t>(c.h_pinned[20] >> 40, c.nchunks), c.nchunks);
    if (NS > 1) {
        if (c.S <= s_cap && 10 * c.tcap >= 13 * c.S) c.table_built = true;  // load <= 0.77 at worst; normally the intended 0.25
        else {                                                            // the extrapolation was too small: build it the plain way
            W2_HIP(hipStreamSynchronize(st2));
            c.release(c.d_table);
            c.d_table = nullptr;
        }
    }
    return 0;
}

// ---- dictionary built incrementally from gathered solid k-mers (multi-GPU): every append is copied behind the ones
// before it and inserted into the table on the side stream, while the main stream keeps counting the next bucket slice.
__global__ void __launch_bounds__(256) k_shift_u64(uint64_t n, const uint64_t* __restrict__ in, uint64_t add, uint64_t* __restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] + add;
}
void dict_abort(Ctx& c) {
    if (!c.g_open) return;
    if (c.stream2) (void)hipStreamSynchronize(c.stream2);
    for (void* p : {(void*)c.g_hi, (void*)c.g_lo, (void*)c.g_cc, (void*)c.g_cstart, (void*)c.g_ccnt, (void*)c.d_table}) if (p) c.release(p);
    c.g_hi = c.g_lo = nullptr; c.g_cc = nullptr; c.g_cstart = nullptr; c.g_ccnt = nullptr; c.d_table = nullptr;
    c.g_open = false; c.g_n = c.g_nc = 0;
}
int dict_begin(Ctx& c, uint64_t kmer_cap, uint64_t chunk_cap) {
    if (c.g_open) dict_abort(c);
    if (!c.stream2) { c.err = "dict_begin: no side stream"; return W2RAP_E_STATE; }
    if (kmer_cap >= MAX_SOLID_KMERS) { c.err = "more than 2^32 solid k-mers on one GPU"; return W2RAP_E_LIMIT; }
    if (c.d_table) c.release(c.d_table);
    c.d_table = nullptr;
    W2_ALLOC(c.g_hi, uint64_t, kmer_cap); W2_ALLOC(c.g_lo, uint64_t, kmer_cap); W2_ALLOC(c.g_cc, uint32_t, kmer_cap);
    c.g_cstart = nullptr; c.g_ccnt = nullptr;
    if (chunk_cap) { W2_ALLOC(c.g_cstart, uint64_t, chunk_cap); W2_ALLOC(c.g_ccnt, uint32_t, chunk_cap); }
    c.g_cap = kmer_cap; c.g_ccap = chunk_cap; c.g_n = 0; c.g_nc = 0;
    W2_TRY(table_alloc(c, kmer_cap, c.stream2));
    c.g_open = true;
    return 0;
}
// (chunk_bias: the chunk starts handed over count from that k-mer of the source array on -- a slice of an owner's arrays, given in place)
int dict_append(Ctx& c, const uint64_t* d_hi, const uint64_t* d_lo, const uint32_t* d_cc, uint64_t n, const uint64_t* d_cstart, const uint32_t* d_ccnt,
                uint64_t nc, uint64_t chunk_bias) {
    hipStream_t st2 = c.stream2;
    if (!c.g_open) { c.err = "dict_append before dict_begin"; return W2RAP_E_STATE; }
    if (c.g_n + n > c.g_cap || (nc && c.g_nc + nc > c.g_ccap)) { c.err = "dict_append: capacity of dict_begin exceeded"; return W2RAP_E_LIMIT; }
    // first everything that is COPIED (k-mers, chunk list), then an event, then the insert kernel: dict_end lets the main stream
    // wait for the last append's copies only -- the bucket-local prune reads the k-mers and the chunk list but not the table, and
    // runs beside the last insert
    if (n) {
        W2_HIP(hipMemcpyAsync(c.g_hi + c.g_n, d_hi, n * 8, hipMemcpyDeviceToDevice, st2));
        W2_HIP(hipMemcpyAsync(c.g_lo + c.g_n, d_lo, n * 8, hipMemcpyDeviceToDevice, st2));
        W2_HIP(hipMemcpyAsync(c.g_cc + c.g_n, d_cc, n * 4, hipMemcpyDeviceToDevice, st2));
    }
    if (nc) {
        hipLaunchKernelGGL(k_shift_u64, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st2, nc, d_cstart, c.g_n - chunk_bias, c.g_cstart + c.g_nc);
        W2_HIP(hipMemcpyAsync(c.g_ccnt + c.g_nc, d_ccnt, nc * 4, hipMemcpyDeviceToDevice, st2));
    }
    if (!c.g_copied) W2_HIP(hipEventCreateWithFlags(&c.g_copied, hipEventDisableTiming));
    W2_HIP(hipEventRecord(c.g_copied, st2));
    if (n) {
        LAUNCH_ON(c, st2, "k_table_insert", k_table_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.g_n, c.g_n + n,
                  c.g_hi, c.g_lo, c.d_table, c.tcap - 1);
        W2_HIP(hipGetLastError());
    }
    c.g_n += n; c.g_nc += nc;
    return 0;
}
// the gathered dictionary becomes the context's solid set; adjacency prune (count_table)
int dict_end(Ctx& c) {
    if (!c.g_open) { c.err = "dict_end before dict_begin"; return W2RAP_E_STATE; }
    if (c.cs_planned) { c.err = "dict_end while a sliced count is pending (count_records_end first)"; return W2RAP_E_STATE; }
    for (void* p : {(void*)c.d_shi, (void*)c.d_slo, (void*)c.d_scc, (void*)c.d_recs, (void*)c.d_sctx, (void*)c.d_nbr, (void*)c.d_chunk_start, (void*)c.d_chunk_cnt})
        if (p) c.release(p);
    c.d_recs = nullptr; c.d_sctx = nullptr; c.d_nbr = nullptr;
    c.d_shi = c.g_hi; c.d_slo = c.g_lo; c.d_scc = c.g_cc; c.S = c.g_n; c.solid_cap = c.g_cap;
    c.d_chunk_start = c.g_cstart; c.d_chunk_cnt = c.g_ccnt; c.nchunks = c.g_nc;
    c.g_hi = c.g_lo = nullptr; c.g_cc = nullptr; c.g_cstart = nullptr; c.g_ccnt = nullptr; c.g_open = false;
    bool wait_copies = true;
#ifdef W2RAP_TESTING
    if (test_hook("W2RAP_TEST_NO_APPEND_WAIT")) wait_copies = false;     // re-opens the race of commit b0ca512 (testing builds only)
#endif
    if (c.g_copied && wait_copies) W2_HIP(hipStreamWaitEvent(c.stream, c.g_copied, 0));       // the appended k-mers and chunks are in place (the last insert may still run)
    if (10 * c.tcap >= 13 * c.S) c.table_built = true;
    else {                                               // capacity guess far too small for the load factor: plain rebuild
        W2_HIP(hipStreamSynchronize(c.stream2));
        c.release(c.d_table);
        c.d_table = nullptr; c.table_built = false;
    }
    return count_table(c);
}

// ---- pieces of K4 + K5 for the sharded dictionary (step2_shard.hip): the table over this owner's k-mers, built in one go; the
// bucket-local step of the prune over its chunk list with 64-bit node numbers (LOCAL ones: 2 * index + orientation)
int table_build_plain(Ctx& c) {
    if (c.d_table) { c.release(c.d_table); c.d_table = nullptr; }
    W2_TRY(table_alloc(c, c.S, c.stream));
    if (c.S) {
        LAUNCH(c, "k_table_insert", k_table_insert, dim3((unsigned)((c.S + 255) / 256)), dim3(256), 0, (uint64_t)0, c.S, c.d_shi, c.d_slo, c.d_table, c.tcap - 1);
        W2_HIP(hipGetLastError());
    }
    c.table_built = false;
    return 0;
}
// The owner's OWN dictionary built under its counting (sharded dictionary, row e-3): after bucket slice k the solid k-mers [done, n_solid) of
// c.d_shi are inserted on the side stream while slice k+1 counts (what count_buckets does on one GPU).  The first call lays the table out
// for `expected_total` (the caller's extrapolation from the first slice); a table that turns out too small is rebuilt by shard_begin.
int local_dict_slice(Ctx& c, uint64_t n_solid, uint64_t expected_total) {
    if (!c.stream2) return 0;
    if (!c.table_built) {
        if (c.d_table) { c.release(c.d_table); c.d_table = nullptr; }
        W2_TRY(table_alloc(c, std::max<uint64_t>(expected_total, n_solid), c.stream2));
        c.table_built = true; c.ld_done = 0;
    }
    if (n_solid > c.solid_cap) n_solid = c.solid_cap;
    if (n_solid > c.ld_done && 10 * c.tcap >= 13 * n_solid) {          // (beyond its load limit the table is abandoned: nothing more goes in, shard_begin rebuilds)
        LAUNCH_ON(c, c.stream2, "k_table_insert", k_table_insert, dim3((unsigned)((n_solid - c.ld_done + 255) / 256)), dim3(256), 0, c.ld_done, n_solid,
                  c.d_shi, c.d_slo, c.d_table, c.tcap - 1);
        W2_HIP(hipGetLastError());
        c.ld_done = n_solid;
    }
    return 0;
}
int prune_local_chunks64(Ctx& c, uint8_t* sctx, uint64_t* nbr, uint8_t* unres) {
    hipStream_t st = c.stream;
    W2_HIP(hipMemsetAsync(unres, 0xFF, c.S, st));       // unvisited k-mers (oversized or unlisted chunks): every bit open
    W2_HIP(hipMemsetAsync(sctx, 0xFF, c.S, st));
    if (!c.nchunks || getenv("W2RAP_NO_LOCAL_PRUNE")) return 0;
    const unsigned gl = (unsigned)std::min<uint64_t>(c.nchunks, (uint64_t)c.sm_count * 64);
    PL_LAUNCH(c, uint64_t, gl, c.S, c.nchunks, c.d_chunk_start, c.d_chunk_cnt, c.d_shi, c.d_slo, c.d_scc, sctx, nbr, unres);
    W2_HIP(hipGetLastError());
    return 0;
}
int prune_local_chunks32(Ctx& c, uint8_t* sctx, uint32_t* nbr, uint8_t* unres) {
    hipStream_t st = c.stream;
    W2_HIP(hipMemsetAsync(unres, 0xFF, c.S, st));       // unvisited k-mers (oversized or unlisted chunks): every bit open
    W2_HIP(hipMemsetAsync(sctx, 0xFF, c.S, st));
    if (!c.nchunks || getenv("W2RAP_NO_LOCAL_PRUNE")) return 0;
    const unsigned gl = (unsigned)std::min<uint64_t>(c.nchunks, (uint64_t)c.sm_count * 64);
    PL_LAUNCH(c, uint32_t, gl, c.S, c.nchunks, c.d_chunk_start, c.d_chunk_cnt, c.d_shi, c.d_slo, c.d_scc, sctx, nbr, unres);
    W2_HIP(hipGetLastError());
    return 0;
}

// ---- K4+K5: lookup table over c.d_shi/d_slo/d_scc[0..S) and adjacency prune
template <class Id>
static int count_table_t(Ctx& c) {
    hipStream_t st = c.stream;
    if (!c.table_built) W2_TRY(table_alloc(c, c.S, st));
    const bool fused = c.fused_prune && sizeof(Id) == 4 && c.d_sctx && c.d_nbr && c.d_unres;      // most chunks are pruned already (inside k_count_fp's emit, or slice by slice on the side stream)
    Id* nbr = nullptr;
    if (fused) nbr = reinterpret_cast<Id*>(c.d_nbr);
    else {
        for (void* p : {(void*)c.d_sctx, (void*)c.d_nbr, (void*)c.d_unres}) if (p) c.release(p);
        c.d_unres = nullptr;
        W2_ALLOC(c.d_sctx, uint8_t, c.S);
        W2_ALLOC(nbr, Id, 2 * c.S);
        c.d_nbr = nbr;
    }
    c.fused_prune = false;
    if (c.S) {
        unsigned g = (unsigned)((c.S + 255) / 256);
        if (!c.table_built) {
            LAUNCH(c, "k_table_insert", k_table_insert, dim3(g), dim3(256), 0, (uint64_t)0, c.S, c.d_shi, c.d_slo, c.d_table, c.tcap - 1);
            W2_HIP(hipGetLastError());
        }
        uint8_t* d_unres = nullptr;
        if (fused) {
            d_unres = c.d_unres; c.d_unres = nullptr;
            for (auto& rg : c.unfused_chunks) {                  // the list kernel's chunks: the bucket-local prune as before
                const uint64_t nch = rg.second - rg.first;
                if (!nch) continue;
                const unsigned gl = (unsigned)std::min<uint64_t>(nch, (uint64_t)c.sm_count * 64);
                PL_LAUNCH(c, Id, gl, c.S, nch, c.d_chunk_start + rg.first, c.d_chunk_cnt + rg.first, c.d_shi, c.d_slo, c.d_scc, c.d_sctx, nbr, d_unres);
                W2_HIP(hipGetLastError());
            }
        } else if (c.nchunks) {
            W2_ALLOC(d_unres, uint8_t, c.S);
            W2_HIP(hipMemsetAsync(d_unres, 0xFF, c.S, st));       // unvisited k-mers (oversized or unlisted chunks): every bit open
            W2_HIP(hipMemsetAsync(c.d_sctx, 0xFF, c.S, st));
            // (the two steps as a pipeline over four groups of chunks on two streams -- bucket-local step of group j+1 beside the global probes of
            //  group j -- was measured in round 4: 12.8-13.8 ms against 11.9 ms back to back: both wait for the same random sectors.  Removed.)
            const unsigned gl = (unsigned)std::min<uint64_t>(c.nchunks, (uint64_t)c.sm_count * 64);
            PL_LAUNCH(c, Id, gl, c.S, c.nchunks, c.d_chunk_start, c.d_chunk_cnt, c.d_shi, c.d_slo, c.d_scc, c.d_sctx, nbr, d_unres);
            W2_HIP(hipGetLastError());
        }
        if (c.table_built && c.stream2) {            // dictionary built on the side stream: complete before the first probe
            hipEvent_t ev;
            W2_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            W2_HIP(hipEventRecord(ev, c.stream2));
            W2_HIP(hipStreamWaitEvent(st, ev, 0));
            (void)hipEventDestroy(ev);
        }
        LAUNCH(c, "k_prune", k_prune<Id>, dim3(g), dim3(256), 0, (uint64_t)0, c.S, c.d_shi, c.d_slo, c.d_scc, c.d_table, c.tcap - 1, c.d_sctx, nbr,
               (const uint8_t*)d_unres);
        W2_HIP(hipGetLastError());
        W2_HIP(hipStreamSynchronize(st));
        if (d_unres) c.release(d_unres);
    }
    W2_HIP(hipStreamSynchronize(st));
    if (c.stream2) W2_HIP(hipStreamSynchronize(c.stream2));
    c.table_built = false;
    c.counted = true;
    return 0;
}
// ---- K4+K5: lookup table over c.d_shi/d_slo/d_scc[0..S) and adjacency prune.  Node ids are 32-bit words while S < 2^31 and 64-bit
// words beyond (W2RAP_WIDE_IDS=1 forces the wide path on any input: the parity tests run both).
int count_table(Ctx& c) {
    if (c.S >= MAX_SOLID_KMERS) { c.err = "more than 2^32 solid k-mers on one GPU"; return W2RAP_E_LIMIT; }
    const char* wv = getenv("W2RAP_WIDE_IDS");
    c.wide_ids = c.S >= (1ull << 31) - 1 || (wv && atoi(wv) != 0);
    return c.wide_ids ? count_table_t<uint64_t>(c) : count_table_t<uint32_t>(c);
}

// number of hash-range passes of the counting phase when the caller leaves the choice to the library: the super-k-mer records (32 B
// per ~15..23 k-mer instances) and the two descriptor buffers should take no more than a quarter of the free HBM
static unsigned auto_passes(uint64_t M) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || !free_b) return 1;
    const double need = (double)M / 15.0 * REC_BYTES * 1.25;
    unsigned p = (unsigned)(need / ((double)free_b / 4.0)) + 1;
    return std::min(p, 64u);
}

int phase_count(Ctx& c, uint32_t min_qual, uint32_t min_freq) {
    const bool trace = getenv("W2RAP_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    W2_TRY(count_quality(c, min_qual));
    double t1 = now();
    const char* bv = getenv("W2RAP_BATCHES");
    const char* pv = getenv("W2RAP_PASSES");
    unsigned P = pv ? (unsigned)atoi(pv) : c.n_passes;
    if (!P) P = auto_passes(c.M);
    const uint32_t NB = default_buckets(c.M, 1);
    if (P > NB) P = NB;
    if (P < 1) P = 1;
    c.npass = P;
    double t_part = 0, t_cnt = 0;
    for (unsigned p = 0; p < P; ++p) {
        // pass p: the reads are cut again, only the records of buckets [NB p / P, NB (p+1) / P) are kept (MapReduceEngine.h:288-299)
        const uint32_t lo = (uint32_t)((uint64_t)NB * p / P), hi = (uint32_t)((uint64_t)NB * (p + 1) / P);
        unsigned nseg = 1;
        c.pass = p;
        double a = now();
        W2_TRY(count_partition_batched(c, NB, bv ? (unsigned)atoi(bv) : 4, &nseg, lo, hi));
        double b = now();
        W2_TRY(count_buckets(c, min_freq, hi - lo, nseg, c.d_recs, c.d_bcount, c.M, P == 1));
        c.release(c.d_recs); c.d_recs = nullptr;         // the records are no longer needed
        t_part += b - a; t_cnt += now() - b;
    }
    c.pass = 0; c.npass = 1;
    double t3 = now();
    W2_TRY(count_table(c));
    double t4 = now();
    if (trace) fprintf(stderr, "[w2rap] count: quality %.1f ms, partition %.1f ms, buckets %.1f ms, table %.1f ms (%u pass%s)\n",
                       (t1 - t0) * 1e3, t_part * 1e3, t_cnt * 1e3, (t4 - t3) * 1e3, P, P == 1 ? "" : "es");
    return 0;
}

}  // namespace w2
