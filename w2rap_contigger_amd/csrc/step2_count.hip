// step2_count.hip -- phases a1..a6 of Step 2 on gfx950:
//   K0  k_good_len      quality window per read            (BuildReadQGraph.cc:962-987)
//   K1  k_superkmers    canonical-minimizer super-k-mers   (replaces the leaf loop :1062-1080
//   K2                  + scatter into hash buckets         and std::sort's partitioning :1081)
//   K3  k_count_buckets per-bucket LDS hash count/merge     (collapse_entries :1002-1013,
//                        + min_freq filter + histogram       combine_Entries :943-949, filter :1094-1104)
//   K4  k_table_insert  lookup table over solid k-mers      (new BRQ_Dict + insertEntryNoLocking :1092-1099)
//   K5  k_prune         adjacency prune                     (kmers/ReadPather.h:317-346)
//
// Design (MI355X-first, integer/HBM work, no MFMA):
//  * reads are consumed wavefront-per-read, lane = k-mer position: no divergence, the
//    packed read block is loaded once (coalesced byte loads) into LDS;
//  * instead of shipping one 17-B record per k-mer instance to HBM and back, consecutive
//    k-mers that share a canonical minimizer bucket travel as ONE 36-B super-k-mer
//    record (<= 64 k-mers): ~1.6 B/k-mer of partition traffic instead of 34 B;
//  * every bucket is sized to fit an LDS hash table, so counting (count saturating add,
//    context OR) never touches HBM; only distinct solid k-mers are written back;
//  * bucket sizes are data dependent: a bucket whose distinct set overflows the LDS table
//    is re-processed in 2,4,.. hash sub-passes (same semantics as MapReduceEngine.h:288-291).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include "ctx.h"

namespace w2 {

// =============================================================================== K0
// One read per lane; scans the raw qualities backwards for the rightmost window of K
// consecutive q >= min_qual.  good_len is stored as uint16 like the reference (:1056).
__global__ void __launch_bounds__(256) k_good_len(uint64_t n, const uint8_t* __restrict__ quals,
                                                   const uint64_t* __restrict__ qoff, const uint32_t* __restrict__ len,
                                                   uint32_t min_qual, uint16_t* __restrict__ good,
                                                   unsigned long long* __restrict__ total_kmers,
                                                   uint32_t* __restrict__ max_len) {
    // The qualities of the block's 256 consecutive reads are contiguous: copy them to LDS with
    // 16-byte loads, then every lane scans its own read backwards out of LDS.
    constexpr unsigned BUF = 48 * 1024;
    __shared__ __attribute__((aligned(16))) uint8_t qbuf[BUF + 16];
    __shared__ unsigned long long s_sum[4];
    __shared__ uint32_t s_max[4];
    const uint64_t r0 = (uint64_t)blockIdx.x * blockDim.x;
    const uint64_t r = r0 + threadIdx.x;
    const uint64_t rend = r0 + blockDim.x < n ? r0 + blockDim.x : n;
    const uint64_t base0 = qoff[r0], endo = qoff[rend];
    const unsigned shift = (unsigned)((reinterpret_cast<uintptr_t>(quals) + base0) & 15);
    const bool in_lds = (endo - base0) + shift <= BUF;
    if (in_lds) {
        const uint4* src = reinterpret_cast<const uint4*>(quals + base0 - shift);
        const unsigned n16 = (unsigned)((endo - base0 + shift + 15) >> 4);
        for (unsigned i = threadIdx.x; i < n16; i += blockDim.x) reinterpret_cast<uint4*>(qbuf)[i] = src[i];
    }
    __syncthreads();
    unsigned long long mine = 0;
    uint32_t L = 0;
    if (r < n) {
        L = len[r];
        const uint64_t o = qoff[r];
        uint32_t g = 0, run = 0;
        if (in_lds) {
            const uint8_t* q = qbuf + shift + (unsigned)(o - base0);
            for (uint32_t i = L; i-- > 0;) {
                if (q[i] < min_qual) run = 0;
                else if (++run == K) { g = i + K; break; }
            }
        } else {
            const uint8_t* q = quals + o;
            for (uint32_t i = L; i-- > 0;) {
                if (q[i] < min_qual) run = 0;
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
        if (t) atomicAdd(total_kmers, t);
        if (m) atomicMax(max_len, m);
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
}

int decode_pq(Ctx& c, const uint8_t* d_pq, const uint64_t* d_pqoff, uint8_t* d_quals, const uint64_t* d_qoff) {
    if (!c.n) return 0;
    LAUNCH(c, "k_decode_pq", k_decode_pq, dim3((unsigned)((c.n + 255) / 256)), dim3(256), 0, c.n, d_pq, d_pqoff, d_quals, d_qoff);
    W2_HIP(hipGetLastError());
    return 0;
}

// =============================================================================== K1+K2
__device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
// bucket of a minimizer key (the key is a bijective image of the canonical m-mer; it is re-mixed
// because the minimum of 46 keys is biased towards small values)
__device__ inline uint32_t bucket_of(uint32_t key, uint32_t nb) {
    uint32_t h = mix32(key * 0x9E3779B1u + 0x85EBCA6Bu);
    return (uint32_t)(((uint64_t)h * nb) >> 32);
}

// One wavefront per read; a pass covers 128 k-mer positions (two per lane), i.e. a whole PE150 read.
// The canonical-minimizer of every k-mer is a sliding-window minimum over 46 m-mer keys; the window
// minimum is computed by doubling (2,4,8,16,32, then 32+16) entirely in registers with ds_bpermute
// lane shifts, so a pass is six cross-lane round trips deep and needs no barrier.  The m-mer key is
// mix32(canonical 15-mer): mix32 is a bijection, so distinct m-mers never tie and the choice depends
// only on the SET of canonical m-mers in the window (strand-symmetric, as it must be for a k-mer
// and its reverse complement to land in the same bucket).
// WRITE=false: count records per bucket.  WRITE=true: write the records.
__device__ inline uint32_t lane_shift(uint32_t lo, uint32_t hi, unsigned lane, unsigned d) {
    // value at position (lane + d) of the concatenation lo[0..63] ++ hi[0..63]
    const int idx = (int)(((lane + d) & 63u) << 2);
    const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)lo);
    const uint32_t b = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)hi);
    return lane + d < 64 ? a : b;
}

template <bool WRITE>
__global__ void __launch_bounds__(64) k_superkmers(uint64_t n, const uint8_t* __restrict__ bases,
                                                    const uint64_t* __restrict__ boff, const uint16_t* __restrict__ good,
                                                    uint32_t nb, uint32_t* __restrict__ bcount,
                                                    const uint64_t* __restrict__ bbase, uint32_t* __restrict__ cursor,
                                                    uint32_t* __restrict__ recs, uint32_t* __restrict__ bkmers) {
    __shared__ uint32_t rdw[24];          // rdw[0] = 0 pad, stream from rdw[1]
    const unsigned lane = threadIdx.x;
    const uint64_t nwaves = gridDim.x;
    for (uint64_t r = blockIdx.x; r < n; r += nwaves) {
        const unsigned gl = good[r];
        if (gl <= K) continue;                                   // strict, BuildReadQGraph.cc:1064
        const unsigned nk_total = gl - (K - 1);
        const uint8_t* rb = bases + boff[r];
        const unsigned nbytes_read = (gl + 3) >> 2;
        for (unsigned c0 = 0; c0 < nk_total; c0 += 128) {
            // ---- stage the window of the read this pass needs: bases [c0-1, c0+189) ----
            const unsigned first_base = c0 ? c0 - 1 : 0;
            const unsigned b0a = (first_base >> 2) & ~3u;        // window start byte, dword aligned in the read
            __syncthreads();
            if (lane < 24) rdw[lane] = 0;
            __syncthreads();
            {
                unsigned by = b0a + lane;                        // 56 bytes cover the window
                if (by < nbytes_read && lane < 60) reinterpret_cast<uint8_t*>(rdw + 1)[lane] = rb[by];
            }
            __syncthreads();
            const uint32_t* st = rdw + 1;                        // stream position s <-> read base 4*b0a + s
            const unsigned sbase = 4 * b0a;
            // ---- canonical m-mer keys at m-mer positions c0+lane, c0+64+lane, c0+128+lane ----
            uint32_t k0, k1, k2;
            {
                uint32_t kk[3];
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    const unsigned j = c0 + h * 64 + lane;
                    uint32_t key = 0xFFFFFFFFu;
                    if (j + MMER <= gl && (h < 2 || lane < WIN - 1)) {
                        const unsigned sp = j - sbase, o = 2 * sp, wi = o >> 5, sh = o & 31;
                        const uint64_t x = ((uint64_t)st[wi] | ((uint64_t)st[wi + 1] << 32)) >> sh;
                        const uint32_t f = (uint32_t)x & 0x3FFFFFFFu;
                        const uint32_t rc = (uint32_t)(rev2_64((uint64_t)(~f & 0x3FFFFFFFu)) >> 34);
                        key = mix32(f < rc ? f : rc);
                    }
                    kk[h] = key;
                }
                k0 = kk[0]; k1 = kk[1]; k2 = kk[2];
            }
            // ---- sliding-window minimum over WIN=46 keys: windows 2,4,8,16,32 by doubling, then 32+16 ----
            uint32_t a0 = k0, a1 = k1, a2 = k2, w16_0 = 0, w16_1 = 0, w16_2 = 0;
#pragma unroll
            for (unsigned d = 1; d <= 16; d <<= 1) {
                const uint32_t s0 = lane_shift(a0, a1, lane, d), s1 = lane_shift(a1, a2, lane, d), s2 = lane_shift(a2, 0xFFFFFFFFu, lane, d);
                a0 = min(a0, s0); a1 = min(a1, s1); a2 = min(a2, s2);
                if (d == 8) { w16_0 = a0; w16_1 = a1; w16_2 = a2; }
            }
            // a* = window 32; final window 46 = w32[p] min w16[p+30]
            const uint32_t mk0 = min(a0, lane_shift(w16_0, w16_1, lane, WIN - 16));
            const uint32_t mk1 = min(a1, lane_shift(w16_1, w16_2, lane, WIN - 16));
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
                const uint32_t prev = __shfl_up(bkt, 1);
                const bool start = valid && (lane == 0 || prev != bkt);
                const unsigned long long smask = __ballot(start);
                unsigned nvalid = cc0 < nk_total ? nk_total - cc0 : 0; if (nvalid > 64) nvalid = 64;
                unsigned long long rest = lane < 63 ? (smask >> (lane + 1)) : 0ull;
                unsigned nxt = rest ? lane + 1 + __builtin_ctzll(rest) : nvalid;
                sth[h] = start; bkh[h] = bkt; nkh[h] = start ? nxt - lane : 0;
            }
            if (!WRITE) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (sth[h]) { atomicAdd(&bcount[bkh[h]], 1u); if (bkmers) atomicAdd(&bkmers[bkh[h]], nkh[h]); }
            } else {
                uint32_t slot[2] = {0, 0}; uint64_t base[2] = {0, 0};
#pragma unroll
                for (int h = 0; h < 2; ++h) if (sth[h]) { slot[h] = atomicAdd(&cursor[bkh[h]], 1u); base[h] = bbase[bkh[h]]; }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (!sth[h]) continue;
                    const unsigned p = c0 + 64 * h + lane, nk = nkh[h];
                    bool hasL = p > 0, hasR = (p + nk - 1) < (nk_total - 1);
                    uint32_t out[9];
                    out[0] = (nk - 1) | (hasL ? 64u : 0u) | (hasR ? 128u : 0u);
                    // copy 2*(nk+61) stream bits starting at base p-1 (rdw[0] is the zero pad before base 0)
                    int sp = (int)p - 1 - (int)sbase;            // >= -1
                    unsigned bo = (unsigned)(32 + 2 * sp);
                    unsigned nbits = 2 * (nk + 61);
#pragma unroll
                    for (unsigned t = 0; t < 8; ++t) {
                        unsigned o = bo + 32 * t, wi = o >> 5, sh = o & 31;
                        uint32_t v = 0;
                        if (32 * t < nbits) {
                            uint64_t x = ((uint64_t)rdw[wi] | ((uint64_t)rdw[wi + 1] << 32)) >> sh;
                            v = (uint32_t)x;
                            unsigned remain = nbits - 32 * t;
                            if (remain < 32) v &= (1u << remain) - 1;
                        }
                        out[1 + t] = v;
                    }
                    uint32_t* dst = recs + (base[h] + slot[h]) * REC_DWORDS;
#pragma unroll
                    for (unsigned t = 0; t < 9; ++t) dst[t] = out[t];
                }
            }
        }
    }
}

// =============================================================================== K3
// Per-bucket LDS hash table.  state: 0 empty, 1 locked (key being written), >=2 ready
// (bits 31:2 = 30 hash bits for early reject).  cc: bits 23:0 occurrence count,
// bits 31:24 OR of contexts.
//
// A bucket's records are streamed through LDS in tiles of THREADS records (coalesced dword
// loads, next tile prefetched into registers while the current one is counted).  Inside a
// tile the k-mers of a wavefront's records are flattened: lane = k-mer index in the wave's
// share of the tile (record found by a 6-step search over the wave's prefix sums), so all
// 64 lanes insert into the hash table regardless of how long the individual records are.
template <unsigned CAP, unsigned THREADS, int ABLATE = 0>
__global__ void __launch_bounds__(THREADS) k_count_buckets(uint32_t nb, uint32_t nseg, const uint64_t* __restrict__ roff,
                                                            const uint32_t* __restrict__ recs, uint32_t min_freq,
                                                            uint32_t* __restrict__ queue,
                                                            uint64_t* __restrict__ shi, uint64_t* __restrict__ slo,
                                                            uint32_t* __restrict__ scc, uint64_t solid_cap,
                                                            unsigned long long* __restrict__ counters /*0 solid,1 distinct,2 overflow passes,3 error*/,
                                                            unsigned long long* __restrict__ ghist) {
    constexpr unsigned NW = THREADS / 64;
    constexpr unsigned LIMIT = CAP - THREADS - 8;
    constexpr unsigned TILE = THREADS;                       // records per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* khi = reinterpret_cast<uint64_t*>(smem);
    uint64_t* klo = khi + CAP;
    uint32_t* state = reinterpret_cast<uint32_t*>(klo + CAP);
    uint32_t* cc = state + CAP;
    uint32_t* tile = cc + CAP;                               // TILE * 9 dwords (+ 4 pad)
    uint32_t* wst = tile + TILE * REC_DWORDS + 4;            // NW * 64 prefix sums
    uint32_t* lhist = wst + NW * 64;                         // 104
    uint32_t* misc = lhist + 104;                            // 0 bucket, 1 fill, 2 overflow, 3 stack depth
    uint32_t* stk = misc + 8;                                // (class, P) pairs, depth <= 18
    auto ld = [](uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned long long my_distinct = 0;
    for (unsigned i = tid; i < 104; i += THREADS) lhist[i] = 0;
    if (tid < 4) tile[TILE * REC_DWORDS + tid] = 0;
    for (;;) {
        __syncthreads();
        if (tid == 0) misc[0] = atomicAdd(queue, 1u);
        __syncthreads();
        const uint32_t b = misc[0];
        if (b >= nb) break;
        // (class, P) work stack: a class is the k-mers with (hash>>40) & (P-1) == class.  A class whose
        // distinct set overflows the table is split into its two refinements at 2P; finished classes stay valid.
        if (tid == 0) { stk[0] = 0; stk[1] = 1; misc[3] = 1; }
        __syncthreads();
        while (ld(&misc[3])) {
            const unsigned sp = ld(&misc[3]) - 1;
            const uint32_t cls = stk[2 * sp], P = stk[2 * sp + 1];
            __syncthreads();
            for (unsigned i = tid; i < CAP; i += THREADS) { state[i] = 0; cc[i] = 0; }
            if (tid == 0) { misc[1] = 0; misc[2] = 0; misc[3] = sp; misc[4] = 0; misc[7] = 0; }
            __syncthreads();
            for (uint32_t seg = 0; seg < nseg; ++seg) {
                // bucket b's records in segment seg: [r0, r1)
                const uint64_t r0 = roff[(uint64_t)seg * nb + b], r1 = roff[(uint64_t)seg * nb + b + 1];
                if (r0 == r1) continue;
                const uint32_t* src = recs + r0 * REC_DWORDS;
                const uint64_t ndw = (r1 - r0) * REC_DWORDS;
                // prefetch tile 0
                uint32_t pf[REC_DWORDS];
#pragma unroll
                for (unsigned j = 0; j < REC_DWORDS; ++j) { uint64_t d = (uint64_t)j * THREADS + tid; pf[j] = d < ndw ? src[d] : 0u; }
                for (uint64_t t0 = 0; t0 < r1 - r0; t0 += TILE) {
                    __syncthreads();                         // previous tile fully consumed
#pragma unroll
                    for (unsigned j = 0; j < REC_DWORDS; ++j) tile[j * THREADS + tid] = pf[j];
                    {   // prefetch the next tile while this one is counted
                        const uint64_t base = (t0 + TILE) * REC_DWORDS;
#pragma unroll
                        for (unsigned j = 0; j < REC_DWORDS; ++j) { uint64_t d = base + (uint64_t)j * THREADS + tid; pf[j] = d < ndw ? src[d] : 0u; }
                    }
                    __syncthreads();
                    const unsigned nrec_tile = (unsigned)((r1 - r0 - t0) < TILE ? (r1 - r0 - t0) : TILE);
                    // wave wv owns records wv, wv+NW, ... of the tile; lane l <-> record l*NW + wv
                    const unsigned myrec = lane * NW + wv;
                    unsigned nk = myrec < nrec_tile ? (tile[myrec * REC_DWORDS] & 63u) + 1u : 0u;
                    unsigned incl = nk;                      // inclusive scan over the wave
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { unsigned v = __shfl_up(incl, o); if ((int)lane >= o) incl += v; }
                    uint32_t* ws = wst + wv * 64;
                    ws[lane] = incl - nk;                    // exclusive
                    const unsigned total = __shfl(incl, 63);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    for (unsigned g0 = 0; g0 < (ABLATE == 2 ? 0u : total); g0 += 64) {
                        const unsigned g = g0 + lane;
                        bool active = g < total && !ld(&misc[2]);
                        Kmer k{0, 0}; unsigned ctx = 0; uint64_t h = 0;
                        if (active) {
                            unsigned lo_i = 0;               // largest i with ws[i] <= g
#pragma unroll
                            for (unsigned step = 32; step > 0; step >>= 1) { unsigned c2 = lo_i + step; if (ws[c2] <= g) lo_i = c2; }
                            const unsigned rec = lo_i * NW + wv;
                            const uint32_t* w = tile + rec * REC_DWORDS;
                            const uint32_t hdr = w[0];
                            const unsigned rnk_ = (hdr & 63u) + 1u, idx = g - ws[lo_i];
                            const bool hasL = hdr & 64, hasR = hdr & 128;
                            const uint32_t* st = w + 1;
                            k = stream_kmer(st, idx + 1);
                            if (idx > 0 || hasL) ctx |= 1u << (4 + stream_base(st, idx));
                            if (idx + 1 < rnk_ || hasR) ctx |= 1u << stream_base(st, idx + 61);
                            if (kmer_canon(k)) ctx = brev8(ctx);
                            h = kmer_hash(k);
                            if (((uint32_t)(h >> 40) & (P - 1)) != cls) active = false;
                        }
                        if (ABLATE == 1) { if (active && h == 0x1234567ull) atomicAdd(&misc[1], 1u); active = false; }
                        if (active) {
                            unsigned s = (unsigned)h & (CAP - 1);
                            const uint32_t tag = ((uint32_t)(h >> 32) << 2) | 2u;
                            bool done = false;
                            while (!done) {
                                uint32_t stv = __hip_atomic_load(&state[s], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (stv == 0) {
                                    if (ld(&misc[2])) break;
                                    uint32_t old = atomicCAS(&state[s], 0u, 1u);
                                    if (old == 0) {
                                        khi[s] = k.hi; klo[s] = k.lo;
                                        __hip_atomic_store(&state[s], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        uint32_t f = atomicAdd(&misc[1], 1u);
                                        if (f >= LIMIT) atomicExch(&misc[2], 1u);
                                        done = true;
                                        break;
                                    }
                                    stv = old;
                                }
                                if (stv == 1) continue;                          // another lane is writing this slot's key
                                if (stv == tag && khi[s] == k.hi && klo[s] == k.lo) { done = true; break; }
                                s = (s + 1) & (CAP - 1);
                            }
                            if (done) {
                                uint32_t old = atomicAdd(&cc[s], 1u);
                                if ((old & 0xFFFFFFu) >= 0xFFFFF0u) atomicSub(&cc[s], 1u);
                                if (((old >> 24) & ctx) != ctx) atomicOr(&cc[s], ctx << 24);
                            }
                        }
                    }
                }
            }
            __syncthreads();
            if (ld(&misc[2])) {                // distinct set does not fit: refine this class and retry
                __syncthreads();
                if (tid == 0) {
                    atomicAdd(&counters[2], 1ull);
                    if (P >= (1u << 16)) { counters[3] = 1; }
                    else { stk[2 * sp] = cls + P; stk[2 * sp + 1] = 2 * P; stk[2 * sp + 2] = cls; stk[2 * sp + 3] = 2 * P; misc[3] = sp + 2; }
                }
                __syncthreads();
                continue;
            }
            // ---- emit: histogram over ALL distinct k-mers (:1097), solid ones to HBM (:1098-1100).
            // One global atomic per (bucket, class): the block sums its solid slots in LDS, reserves
            // the output range once and then places the entries with an LDS cursor.
            {
                constexpr unsigned PER = CAP / THREADS;
                uint32_t vals[PER];
                unsigned nsolid = 0;
#pragma unroll
                for (unsigned j = 0; j < PER; ++j) {
                    const unsigned i = j * THREADS + tid;
                    const uint32_t stv = state[i];
                    const bool occ = stv >= 2;
                    const uint32_t v = occ ? cc[i] : 0;
                    uint32_t cnt = v & 0xFFFFFFu; if (cnt > 255) cnt = 255;      // :943-949 saturating u8
                    {   // histogram: singletons (sequencing errors) dominate -> one LDS atomic per wave for bin 1
                        const unsigned long long m1 = __ballot(occ && cnt == 1);
                        if (m1 && lane == (unsigned)__builtin_ctzll(m1)) atomicAdd(&lhist[1], (uint32_t)__builtin_popcountll(m1));
                        if (occ && cnt != 1) atomicAdd(&lhist[cnt > 100 ? 100 : cnt], 1u);
                        if (occ) ++my_distinct;
                    }
                    const bool solid = occ && cnt >= min_freq;
                    vals[j] = solid ? (cnt | ((v >> 24) << 8) | 0x80000000u) : 0u;
                    nsolid += solid;
                }
                unsigned wsum = nsolid;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) wsum += __shfl_down(wsum, o);
                if (lane == 0 && wsum) atomicAdd(&misc[4], wsum);
                __syncthreads();
                if (tid == 0) {
                    const uint32_t tot = misc[4];
                    unsigned long long base = tot ? atomicAdd(&counters[0], (unsigned long long)tot) : 0ull;
                    misc[5] = (uint32_t)base; misc[6] = (uint32_t)(base >> 32); misc[4] = 0; misc[7] = 0;
                }
                __syncthreads();
                const unsigned long long gbase = (unsigned long long)misc[5] | ((unsigned long long)misc[6] << 32);
#pragma unroll
                for (unsigned j = 0; j < PER; ++j) {
                    const bool solid = vals[j] >> 31;
                    const unsigned long long m = __ballot(solid);
                    if (m) {
                        uint32_t wbase = 0;
                        const int leader = __builtin_ctzll(m);
                        if ((int)lane == leader) wbase = atomicAdd(&misc[7], (uint32_t)__builtin_popcountll(m));
                        wbase = __shfl(wbase, leader);
                        if (solid) {
                            const unsigned i = j * THREADS + tid;
                            const unsigned long long pos = gbase + wbase + __builtin_popcountll(m & ((1ull << lane) - 1));
                            if (pos < solid_cap) { shi[pos] = khi[i]; slo[pos] = klo[i]; scc[pos] = vals[j] & 0xFFFFu; }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    for (unsigned i = tid; i < 101; i += THREADS) if (lhist[i]) atomicAdd(&ghist[i], (unsigned long long)lhist[i]);
    for (int o = 32; o > 0; o >>= 1) my_distinct += __shfl_down(my_distinct, o);
    if (lane == 0 && my_distinct) atomicAdd(&counters[1], my_distinct);
}

// =============================================================================== K4
__global__ void __launch_bounds__(256) k_table_insert(uint64_t S, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo,
                                                       const uint32_t* __restrict__ scc, Slot* __restrict__ table, uint64_t mask,
                                                       uint32_t* __restrict__ sslot, uint32_t* __restrict__ filter, uint64_t fmask) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    Kmer k{shi[i], slo[i]};
    const uint64_t h = kmer_hash(k);
    if (filter) atomicOr(&filter[(h >> 34) & fmask], (1u << ((h >> 24) & 31)) | (1u << ((h >> 29) & 31)));
    uint64_t s = h & mask;
    for (;;) {
        unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&table[s].hi), (unsigned long long)EMPTY_HI, (unsigned long long)k.hi);
        if (old == EMPTY_HI) break;
        s = (s + 1) & mask;
    }
    table[s].lo = k.lo;
    table[s].val = make_val(scc[i] >> 8, NONE32, 0);
    table[s].idx = i;
    sslot[i] = (uint32_t)s;
}

// =============================================================================== K5
// KmerDict::recomputeAdjacencies (ReadPather.h:317-346): clear every context bit whose
// neighbour k-mer is not in the solid set.  Membership only, so it is order-free.
__global__ void __launch_bounds__(256) k_prune(uint64_t S, const uint64_t* __restrict__ shi, const uint64_t* __restrict__ slo,
                                                const uint32_t* __restrict__ scc, const Slot* __restrict__ table, uint64_t mask,
                                                uint8_t* __restrict__ sctx, uint32_t* __restrict__ nbr) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S) return;
    Kmer k{shi[i], slo[i]};
    unsigned c = (scc[i] >> 8) & 0xFF;
    // the neighbour found for each context bit is remembered (oriented node id 2*idx + reversed) so that the
    // unipath linking step does not have to probe the dictionary again
    uint32_t ns = NONE32, np = NONE32;
#pragma unroll
    for (unsigned b = 0; b < 4; ++b) {
        if (c & (1u << b)) {
            Kmer nk = kmer_succ(k, b); bool r = kmer_canon(nk);
            int64_t s = table_find(table, mask, nk);
            if (s < 0) c &= ~(1u << b);
            else ns = kmer_is_pal(nk) ? NONE32 - 1 : 2 * (uint32_t)table[s].idx + (r ? 1u : 0u);
        }
        if (c & (16u << b)) {
            Kmer pk = kmer_pred(k, b); bool r = kmer_canon(pk);
            int64_t s = table_find(table, mask, pk);
            if (s < 0) c &= ~(16u << b);
            else np = kmer_is_pal(pk) ? NONE32 - 1 : 2 * (uint32_t)table[s].idx + (r ? 1u : 0u);
        }
    }
    sctx[i] = (uint8_t)c;
    // only meaningful when exactly one successor / predecessor survives (then it is the last one found)
    nbr[2 * i] = popc4(c & 15) == 1 ? ns : NONE32;
    nbr[2 * i + 1] = popc4(c >> 4) == 1 ? np : NONE32;
}

// =============================================================================== driver
static constexpr unsigned COUNT_CAP = 4096, COUNT_THREADS = 1024;
static constexpr unsigned KMERS_PER_BUCKET = 5000;

// ---- K0: quality windows; sets c.M (k-mer instances of this rank's reads) and c.max_len
int count_quality(Ctx& c, uint32_t min_qual) {
    c.min_qual = min_qual;
    c.counted = false;
    hipStream_t st = c.stream;
    const uint64_t n = c.n;
    if (c.d_good) c.release(c.d_good);
    W2_ALLOC(c.d_good, uint16_t, n);
    unsigned long long* d_cnt = nullptr;                 // [0] M  [1] max_len (as u32)
    W2_ALLOC(d_cnt, unsigned long long, 2);
    W2_HIP(hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), st));
    if (n) {
        LAUNCH(c, "k_good_len", k_good_len, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, n, c.d_quals, c.d_qoff, c.d_len, min_qual,
               c.d_good, d_cnt, reinterpret_cast<uint32_t*>(d_cnt + 1));
        W2_HIP(hipGetLastError());
    }
    unsigned long long h_cnt[2];
    W2_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    c.M = h_cnt[0];
    c.max_len = (uint32_t)h_cnt[1];
    c.release(d_cnt);
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

// ---- K1/K2: super-k-mer records of this rank's reads, grouped by bucket (count pass, scan, write pass)
int count_partition(Ctx& c, uint32_t nb, bool want_bucket_kmers) {
    if (!c.quality_done) { c.err = "partition before quality_windows"; return W2RAP_E_STATE; }
    hipStream_t st = c.stream;
    const uint64_t n = c.n;
    c.NB = nb;
    if (c.d_bcount) c.release(c.d_bcount);
    if (c.d_bbase) c.release(c.d_bbase);
    if (c.d_recs) c.release(c.d_recs);
    W2_ALLOC(c.d_bcount, uint32_t, c.NB);
    W2_ALLOC(c.d_bbase, uint64_t, (uint64_t)c.NB + 1);
    uint32_t* d_cursor = nullptr;
    W2_ALLOC(d_cursor, uint32_t, c.NB);
    W2_HIP(hipMemsetAsync(c.d_bcount, 0, (size_t)c.NB * 4, st));
    W2_HIP(hipMemsetAsync(d_cursor, 0, (size_t)c.NB * 4, st));
    if (c.d_bkmers) { c.release(c.d_bkmers); c.d_bkmers = nullptr; }
    if (want_bucket_kmers) {
        W2_ALLOC(c.d_bkmers, uint32_t, c.NB);
        W2_HIP(hipMemsetAsync(c.d_bkmers, 0, (size_t)c.NB * 4, st));
    }
    unsigned ex_grid = (unsigned)std::min<uint64_t>(n ? n : 1, (uint64_t)c.sm_count * 32);
    if (n) {
        LAUNCH(c, "k_superkmers<false>", (k_superkmers<false>), dim3(ex_grid), dim3(64), 0, n, c.d_bases, c.d_boff, c.d_good, c.NB, c.d_bcount,
               (const uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, c.d_bkmers);
        W2_HIP(hipGetLastError());
    }
    W2_TRY(exclusive_scan_u32_to_u64(c, c.d_bcount, c.d_bbase, c.NB));
    W2_HIP(hipMemcpyAsync(&c.nrec, c.d_bbase + c.NB, 8, hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    W2_ALLOC(c.d_recs, uint32_t, c.nrec * REC_DWORDS);
    if (n) {
        LAUNCH(c, "k_superkmers<true>", (k_superkmers<true>), dim3(ex_grid), dim3(64), 0, n, c.d_bases, c.d_boff, c.d_good, c.NB, c.d_bcount,
               c.d_bbase, d_cursor, c.d_recs, (uint32_t*)nullptr);
        W2_HIP(hipGetLastError());
    }
    W2_HIP(hipStreamSynchronize(st));
    c.release(d_cursor);
    return 0;
}

// ---- K3: count `nbl` buckets whose records arrive in `nseg` segments (one per source rank), each
// segment grouped by bucket; d_counts[s*nbl + b] = records of bucket b in segment s, d_recs = the
// segments back to back.  total_kmers bounds the solid set (S <= kmers / min_freq).
int count_buckets(Ctx& c, uint32_t min_freq, uint32_t nbl, uint32_t nseg, const uint32_t* d_recs, const uint32_t* d_counts,
                  uint64_t total_kmers) {
    hipStream_t st = c.stream;
    c.min_freq = min_freq;
    const uint64_t nflat = (uint64_t)nbl * nseg;
    uint64_t* d_off = nullptr;
    W2_ALLOC(d_off, uint64_t, nflat + 1);
    W2_TRY(exclusive_scan_u32_to_u64(c, d_counts, d_off, nflat));
    unsigned long long* d_cnt = nullptr;                 // [2] queue  [4..7] counters  [8..108] hist
    W2_ALLOC(d_cnt, unsigned long long, 128);
    W2_HIP(hipMemsetAsync(d_cnt, 0, 128 * sizeof(unsigned long long), st));
    c.solid_cap = total_kmers / (min_freq ? min_freq : 1) + 1;
    for (void* p : {(void*)c.d_shi, (void*)c.d_slo, (void*)c.d_scc}) if (p) c.release(p);
    W2_ALLOC(c.d_shi, uint64_t, c.solid_cap);
    W2_ALLOC(c.d_slo, uint64_t, c.solid_cap);
    W2_ALLOC(c.d_scc, uint32_t, c.solid_cap);
    uint32_t* d_queue = reinterpret_cast<uint32_t*>(d_cnt + 2);
    {
        auto launch = [&](auto kern, unsigned cap, unsigned threads, unsigned blocks_per_cu) -> int {
            const unsigned lds = cap * 24 + (threads * REC_DWORDS + 4 + threads + 104 + 8 + 48) * 4;
            W2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            unsigned grid = (unsigned)std::min<uint64_t>(nbl, (uint64_t)c.sm_count * blocks_per_cu);
            LAUNCH(c, "k_count_buckets", kern, dim3(grid), dim3(threads), lds, nbl, nseg, d_off, d_recs, min_freq, d_queue, c.d_shi,
                   c.d_slo, c.d_scc, c.solid_cap, d_cnt + 4, d_cnt + 8);
            W2_HIP(hipGetLastError());
            return 0;
        };
        const char* v = getenv("W2RAP_K3");            // tuning knob: "cap,threads"
        int cfg = v ? atoi(v) : 0;
        if (cfg == 1) W2_TRY(launch(k_count_buckets<4096, 512>, 4096, 512, 1));
        else if (cfg == 2) W2_TRY(launch(k_count_buckets<2048, 256>, 2048, 256, 2));
        else if (cfg == 3) W2_TRY(launch(k_count_buckets<2048, 512>, 2048, 512, 2));
        else if (cfg == 11) W2_TRY(launch(k_count_buckets<COUNT_CAP, COUNT_THREADS, 1>, COUNT_CAP, COUNT_THREADS, 1));
        else if (cfg == 12) W2_TRY(launch(k_count_buckets<COUNT_CAP, COUNT_THREADS, 2>, COUNT_CAP, COUNT_THREADS, 1));
        else W2_TRY(launch(k_count_buckets<COUNT_CAP, COUNT_THREADS>, COUNT_CAP, COUNT_THREADS, 1));
    }
    unsigned long long h_all[128];
    W2_HIP(hipMemcpyAsync(h_all, d_cnt, sizeof(h_all), hipMemcpyDeviceToHost, st));
    W2_HIP(hipStreamSynchronize(st));
    c.release(d_cnt); c.release(d_off);
    if (h_all[7]) { c.err = "k_count_buckets: a bucket did not fit the LDS table after 2^16-way splitting"; return W2RAP_E_LIMIT; }
    c.S = h_all[4]; c.D = h_all[5];
    for (int i = 0; i < 101; ++i) c.hist[i] = h_all[8 + i];
    if (c.S > c.solid_cap) { c.err = "solid k-mer count exceeds its bound"; return W2RAP_E_LIMIT; }
    return 0;
}

// ---- K4+K5: lookup table over c.d_shi/d_slo/d_scc[0..S) and adjacency prune
int count_table(Ctx& c) {
    hipStream_t st = c.stream;
    if (c.S >= (1ull << 31)) { c.err = "more than 2^31 solid k-mers on one GPU (32-bit node ids)"; return W2RAP_E_LIMIT; }
    // slots per solid k-mer: 4 (load <= 0.25: ~1.3 probes per miss instead of ~2.3, pathing is probe-bound)
    // while the table stays under 64 GiB, else 2
    const char* lf = getenv("W2RAP_TABLE_X");
    const uint64_t mult = lf ? (uint64_t)atoll(lf) : (c.S * 4 * sizeof(Slot) <= (64ull << 30) ? 4 : 2);
    uint64_t tcap = 1024;
    while (tcap < mult * c.S) tcap <<= 1;
    c.tcap = tcap;
    W2_ALLOC(c.d_table, Slot, tcap);
    W2_HIP(hipMemsetAsync(c.d_table, 0xFF, tcap * sizeof(Slot), st));
    W2_ALLOC(c.d_sslot, uint32_t, c.S);
    W2_ALLOC(c.d_sctx, uint8_t, c.S);
    W2_ALLOC(c.d_nbr, uint32_t, 2 * c.S);
    // absence filter: >= 4 bits per key, at most 128 MiB (must stay Infinity-Cache resident); skipped beyond that
    c.d_filter = nullptr; c.fwords = 0;
    if (!getenv("W2RAP_NO_FILTER") && c.S && c.S * 4 <= (1ull << 30)) {
        uint64_t words = 1024;
        while (words * 32 < c.S * 4) words <<= 1;
        c.fwords = words;
        W2_ALLOC(c.d_filter, uint32_t, words);
        W2_HIP(hipMemsetAsync(c.d_filter, 0, words * 4, st));
    }
    if (c.S) {
        unsigned g = (unsigned)((c.S + 255) / 256);
        LAUNCH(c, "k_table_insert", k_table_insert, dim3(g), dim3(256), 0, c.S, c.d_shi, c.d_slo, c.d_scc, c.d_table, tcap - 1, c.d_sslot,
               c.d_filter, c.fwords ? c.fwords - 1 : 0);
        W2_HIP(hipGetLastError());
        LAUNCH(c, "k_prune", k_prune, dim3(g), dim3(256), 0, c.S, c.d_shi, c.d_slo, c.d_scc, c.d_table, tcap - 1, c.d_sctx, c.d_nbr);
        W2_HIP(hipGetLastError());
    }
    W2_HIP(hipStreamSynchronize(st));
    c.counted = true;
    return 0;
}

int phase_count(Ctx& c, uint32_t min_qual, uint32_t min_freq) {
    const bool trace = getenv("W2RAP_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    W2_TRY(count_quality(c, min_qual));
    double t1 = now();
    W2_TRY(count_partition(c, default_buckets(c.M, 1), false));
    double t2 = now();
    W2_TRY(count_buckets(c, min_freq, c.NB, 1, c.d_recs, c.d_bcount, c.M));
    double t3 = now();
    c.release(c.d_recs); c.d_recs = nullptr;             // the records are no longer needed
    W2_TRY(count_table(c));
    double t4 = now();
    if (trace) fprintf(stderr, "[w2rap] count: quality %.1f ms, partition %.1f ms, buckets %.1f ms, table %.1f ms\n",
                       (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
    return 0;
}

}  // namespace w2
