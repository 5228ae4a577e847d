// step2_prims.hip -- device-wide utility primitives of the graph phases (E- and S-sized arrays): exclusive / inclusive scans, a maximum,
// a stable radix sort of (u64, u32) pairs.  Hand-written for gfx950 since round 5 (rounds 1-4 wrapped rocPRIM here):
//   * scans: ONE kernel, one pass over the data -- a tile of 4096 elements per block (one ticket = one atomic on one address = ~24 ns: 2048-element tiles made a 150 M-element scan ticket-bound), tiles handed out in order by an atomic ticket, every
//     tile publishes its aggregate and then its inclusive prefix in a 64-bit status word (flag in the top two bits), a tile's exclusive
//     prefix comes from looking back over its predecessors' words (decoupled look-back): 8 B read + 8 B written per element, no
//     second pass, no temporary but the status words;
//   * sort: least-significant-digit radix sort, 8-bit digits, one kernel per pass ("onesweep": the scan of the block counts is a
//     decoupled look-back inside the scatter kernel, one chain per digit); a histogram launch up front gives every pass's digit totals.
//     HBVFromEdges.cc:88-125's sorts (vertices by hash, adjacency by (vertex, vertex)) and the unipath order are E-sized: launch-bound.
// rocPRIM stays only behind sort_pairs_u64 for n > 2^24 (Step 3's largest sorts), where a one-sweep library sort is the better tool.
#include <algorithm>
#include <cstring>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "ctx.h"

namespace w2 {

static void* tmp_alloc(Ctx& c, size_t bytes) { return c.alloc<uint8_t>(bytes ? bytes : 16, false); }

// ------------------------------------------------------------------------------------------------ scan
constexpr unsigned SCAN_THREADS = 256, SCAN_ITEMS = 16, SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;
constexpr unsigned long long ST_EMPTY = 0, ST_AGG = 1ull << 62, ST_PREFIX = 2ull << 62, ST_VAL = (1ull << 62) - 1;

struct OpPlus { __device__ static inline uint64_t id() { return 0; } __device__ static inline uint64_t f(uint64_t a, uint64_t b) { return a + b; } };
struct OpMax { __device__ static inline uint64_t id() { return 0; } __device__ static inline uint64_t f(uint64_t a, uint64_t b) { return a > b ? a : b; } };

// (a thread's SCAN_ITEMS consecutive 32-bit inputs as four 16-byte loads when they are aligned: one element at a time every load
// instruction of a wavefront touched 64 cache lines -- a 149 M-element scan ran at 1.35 TB/s)
__device__ inline bool load16_u32(const uint32_t* q, uint32_t (&t)[16]) {
    if ((uintptr_t)q & 15) return false;
    const uint4* q4 = reinterpret_cast<const uint4*>(q);
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) { const uint4 w = q4[k]; t[4 * k] = w.x; t[4 * k + 1] = w.y; t[4 * k + 2] = w.z; t[4 * k + 3] = w.w; }
    return true;
}
struct InU32 {
    const uint32_t* p;
    __device__ inline uint64_t operator()(uint64_t i) const { return p[i]; }
    __device__ inline bool vec(uint64_t i0, uint64_t (&v)[16]) const {
        uint32_t t[16];
        if (!load16_u32(p + i0, t)) return false;
#pragma unroll
        for (unsigned k = 0; k < 16; ++k) v[k] = t[k];
        return true;
    }
};
struct InU64 { const uint64_t* p; __device__ inline uint64_t operator()(uint64_t i) const { return p[i]; } __device__ inline bool vec(uint64_t, uint64_t (&)[16]) const { return false; } };
struct InIsSelf {
    const uint32_t* a;
    __device__ inline uint64_t operator()(uint64_t i) const { return a[i] == (uint32_t)i ? 1ull : 0ull; }
    __device__ inline bool vec(uint64_t i0, uint64_t (&v)[16]) const {
        uint32_t t[16];
        if (!load16_u32(a + i0, t)) return false;
#pragma unroll
        for (unsigned k = 0; k < 16; ++k) v[k] = t[k] == (uint32_t)(i0 + k) ? 1ull : 0ull;
        return true;
    }
};
static_assert(SCAN_ITEMS == 16, "the 16-element vector paths");

// out[i] = op over in[0 .. i) (EXCL) or in[0 .. i] (!EXCL); EXCL also writes out[n] = the total.  status: one zeroed word per tile + the ticket.
template <class In, class Op, class OutT, bool EXCL>
__global__ void __launch_bounds__(SCAN_THREADS) k_scan(In in, OutT* __restrict__ out, uint64_t n, unsigned long long* __restrict__ status,
                                                        unsigned long long* __restrict__ ticket) {
    __shared__ uint64_t s_wave[SCAN_THREADS / 64];
    __shared__ uint64_t s_excl;
    __shared__ unsigned long long s_tile;
    const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1ull);
    __syncthreads();
    const uint64_t tile = s_tile;
    const uint64_t i0 = tile * SCAN_TILE + (uint64_t)tid * SCAN_ITEMS;
    uint64_t v[SCAN_ITEMS];
    uint64_t mine = Op::id();
    const bool full = i0 + SCAN_ITEMS <= n;
    if (!(full && in.vec(i0, v))) {
#pragma unroll
        for (unsigned j = 0; j < SCAN_ITEMS; ++j) v[j] = i0 + j < n ? in(i0 + j) : Op::id();
    }
#pragma unroll
    for (unsigned j = 0; j < SCAN_ITEMS; ++j) mine = Op::f(mine, v[j]);
    // block scan of the threads' sums: wavefront scan by shuffles, then the four wave totals
    uint64_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t o = __shfl_up(incl, d); if ((int)lane >= d) incl = Op::f(o, incl); }
    if (lane == 63) s_wave[wv] = incl;
    __syncthreads();
    uint64_t wbase = Op::id(), agg = Op::id();
#pragma unroll
    for (unsigned w = 0; w < SCAN_THREADS / 64; ++w) { if (w < wv) wbase = Op::f(wbase, s_wave[w]); agg = Op::f(agg, s_wave[w]); }
    if (wv == 0) {
        // publish the aggregate, look back for the exclusive prefix, publish the inclusive prefix.  The status word CARRIES its value, so
        // nothing else has to be visible with it: relaxed agent-scope atomics (release / acquire at agent scope write back and invalidate
        // the XCD's L2 around every word on this multi-die part: 160 ns per tile, 4 ms for 50 M elements).  The look-back is the WAVEFRONT's: 64
        // predecessors per step (a single lane walking back met ~2000 aggregates -- every tile resident on the GPU -- before the first
        // finished prefix: 12 ms for 50 M elements)
        uint64_t excl = Op::id();
        if (tile == 0) { if (lane == 0) __hip_atomic_store(&status[0], ST_PREFIX | (agg & ST_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else {
            if (lane == 0) __hip_atomic_store(&status[tile], ST_AGG | (agg & ST_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t top = (int64_t)tile - 1;                                   // the nearest predecessor not yet taken in
            for (;;) {
                const int64_t p = top - (int64_t)lane;
                const unsigned long long sv = p >= 0 ? __hip_atomic_load(&status[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (ST_PREFIX | 0ull);   // (before tile 0: an empty prefix)
                const unsigned flag = (unsigned)(sv >> 62);
                const unsigned long long pm = __ballot(flag == 2), em = __ballot(flag == 0);
                const unsigned first_p = pm ? (unsigned)__builtin_ctzll(pm) : 64u;          // the nearest finished prefix in this window
                const unsigned long long need = first_p >= 63 ? ~0ull : ((2ull << first_p) - 1);       // lanes 0 .. first_p (all 64 if there is none)
                if (em & need) continue;                                        // one of them has not published yet: look again
                uint64_t v_ = ((1ull << lane) & need) ? (uint64_t)(sv & ST_VAL) : Op::id();
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) v_ = Op::f(v_, __shfl_xor(v_, d));
                excl = Op::f(v_, excl);
                if (pm) break;
                top -= 64;
            }
            if (lane == 0) __hip_atomic_store(&status[tile], ST_PREFIX | (Op::f(excl, agg) & ST_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_excl = excl;
    }
    __syncthreads();
    // exclusive prefix of this thread = tile prefix + earlier waves + earlier lanes of the wave
    uint64_t lanes_before = __shfl_up(incl, 1);
    if (lane == 0) lanes_before = Op::id();
    uint64_t run = Op::f(s_excl, Op::f(wbase, lanes_before));
    if (full && (((uintptr_t)(out + i0)) & 15) == 0) {                // the thread's outputs as 16-byte stores
        alignas(16) OutT o[SCAN_ITEMS];
#pragma unroll
        for (unsigned j = 0; j < SCAN_ITEMS; ++j) {
            if (EXCL) { o[j] = (OutT)run; run = Op::f(run, v[j]); }
            else { run = Op::f(run, v[j]); o[j] = (OutT)run; }
        }
        uint4* o4 = reinterpret_cast<uint4*>(out + i0); const uint4* s4 = reinterpret_cast<const uint4*>(o);
#pragma unroll
        for (unsigned k = 0; k < SCAN_ITEMS * sizeof(OutT) / 16; ++k) o4[k] = s4[k];
        if (EXCL && n - 1 >= i0 && n - 1 < i0 + SCAN_ITEMS) out[n] = (OutT)run;
        return;
    }
#pragma unroll
    for (unsigned j = 0; j < SCAN_ITEMS; ++j) {
        if (i0 + j < n) {
            if (EXCL) { out[i0 + j] = (OutT)run; run = Op::f(run, v[j]); }
            else { run = Op::f(run, v[j]); out[i0 + j] = (OutT)run; }
        } else run = Op::f(run, v[j]);
    }
    if (EXCL && n - 1 >= i0 && n - 1 < i0 + SCAN_ITEMS) out[n] = (OutT)run;           // the thread that holds the last element: the total
}

// queued on st, no host synchronisation: status = ntiles + 1 words the caller provides (zeroed here)
template <class In, class Op, class OutT, bool EXCL>
static int scan_async(Ctx& c, In in, OutT* out, uint64_t n, hipStream_t st, unsigned long long* status) {
    const uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    W2_HIP(hipMemsetAsync(status, 0, (ntiles + 1) * 8, st));
    hipLaunchKernelGGL((k_scan<In, Op, OutT, EXCL>), dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, st, in, out, n, status, status + ntiles);
    return 0;
}
template <class In, class Op, class OutT, bool EXCL>
static int run_scan(Ctx& c, In in, OutT* out, uint64_t n, hipStream_t st) {
    if (!n) {
        if (EXCL) W2_HIP(hipMemsetAsync(out, 0, sizeof(OutT), st));
        W2_HIP(hipStreamSynchronize(st));
        return 0;
    }
    const uint64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (ntiles >= (1ull << 31)) { c.err = "scan: too many elements"; return W2RAP_E_LIMIT; }
    unsigned long long* status = c.alloc<unsigned long long>(ntiles + 1, false);
    if (!status) return W2RAP_E_HIP;
    W2_HIP(hipMemsetAsync(status, 0, (ntiles + 1) * 8, st));
    hipLaunchKernelGGL((k_scan<In, Op, OutT, EXCL>), dim3((unsigned)ntiles), dim3(SCAN_THREADS), 0, st, in, out, n, status, status + ntiles);
    W2_HIP(hipGetLastError());
    W2_HIP(hipStreamSynchronize(st));
    c.release(status);
    return 0;
}

int exclusive_scan_u32_to_u64(Ctx& c, const uint32_t* in, uint64_t* out, uint64_t n) { return run_scan<InU32, OpPlus, uint64_t, true>(c, InU32{in}, out, n, c.stream); }
int exclusive_scan_u64(Ctx& c, const uint64_t* in, uint64_t* out, uint64_t n) { return run_scan<InU64, OpPlus, uint64_t, true>(c, InU64{in}, out, n, c.stream); }
// out[i] = number of j < i with a[j] == j (Step 3: occurrences that are their own representative), out[n] = their number
int exclusive_scan_is_self(Ctx& c, const uint32_t* a, uint64_t* out, uint64_t n) { return run_scan<InIsSelf, OpPlus, uint64_t, true>(c, InIsSelf{a}, out, n, c.stream); }
// byte offsets of .fastb-packed reads from their lengths: out[i] = sum over j < i of ceil(len[j] / 4)
struct InPackedBytes {
    const uint32_t* p;
    __device__ inline uint64_t operator()(uint64_t i) const { return ((uint64_t)p[i] + 3) >> 2; }
    __device__ inline bool vec(uint64_t i0, uint64_t (&v)[16]) const {
        uint32_t t[16];
        if (!load16_u32(p + i0, t)) return false;
#pragma unroll
        for (unsigned k = 0; k < 16; ++k) v[k] = ((uint64_t)t[k] + 3) >> 2;
        return true;
    }
};
int exclusive_scan_packed_bytes(Ctx& c, const uint32_t* len, uint64_t* out, uint64_t n) { return run_scan<InPackedBytes, OpPlus, uint64_t, true>(c, InPackedBytes{len}, out, n, c.stream); }
int inclusive_max_scan_u32(Ctx& c, const uint32_t* in, uint32_t* out, uint64_t n) { return run_scan<InU32, OpMax, uint32_t, false>(c, InU32{in}, out, n, c.stream); }

// ------------------------------------------------------------------------------------------------ maximum
__global__ void __launch_bounds__(256) k_max_u32(const uint32_t* __restrict__ in, uint64_t n, uint32_t* __restrict__ out) {
    uint32_t m = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) m = max(m, in[i]);
    for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, d));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
int max_u32(Ctx& c, const uint32_t* in, uint64_t n, uint32_t* result) {
    *result = 0;
    if (!n) return 0;
    uint32_t* d_out = c.alloc<uint32_t>(1, false);
    if (!d_out) return W2RAP_E_HIP;
    W2_HIP(hipMemsetAsync(d_out, 0, 4, c.stream));
    const unsigned grid = (unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)c.sm_count * 8);
    hipLaunchKernelGGL(k_max_u32, dim3(grid), dim3(256), 0, c.stream, in, n, d_out);
    W2_HIP(hipMemcpyAsync(result, d_out, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
    W2_HIP(hipStreamSynchronize(c.stream));
    c.release(d_out);
    return 0;
}

// ------------------------------------------------------------------------------------------------ radix sort of (u64 key, u32 value) pairs
// Least-significant-digit first, 8-bit digits, ONE kernel per pass ("onesweep"): the digit totals of ALL passes come from one histogram
// launch up front (they do not depend on the order of the elements); in a pass every block ranks its elements among the equal digits
// before them in the block with wavefront ballots (elements are dealt to lanes round by round, so ballot order is input order: stable),
// and thread d chains the block's count of digit d to the blocks before it by decoupled look-back (the scan's status words, one chain
// per digit): an element's place = digits below it (from the totals) + equal digits in earlier blocks + its rank in the block.  Blocks
// take their tiles in order from an atomic ticket, so a block's predecessors are always running.
constexpr unsigned RS_THREADS = 256, RS_ITEMS = 8, RS_TILE = RS_THREADS * RS_ITEMS, RS_BITS = 8, RS_DIGITS = 1u << RS_BITS, RS_MAXPASS = 8;
static_assert(RS_DIGITS == RS_THREADS, "thread = digit in the look-back");
// tot[pass * 256 + d] = elements whose digit of that pass is d
__global__ void __launch_bounds__(RS_THREADS) k_rs_totals(const uint64_t* __restrict__ keys, uint64_t n, int begin_bit, int end_bit, unsigned long long* __restrict__ tot) {
    __shared__ uint32_t s_h[RS_MAXPASS][RS_DIGITS];
    const unsigned tid = threadIdx.x;
    for (unsigned p = 0; p < RS_MAXPASS; ++p) s_h[p][tid] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + tid; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[i];
        unsigned p = 0;
        for (int sh = begin_bit; sh < end_bit; sh += RS_BITS, ++p) atomicAdd(&s_h[p][(k >> sh) & ((1u << min(RS_BITS, (unsigned)(end_bit - sh))) - 1u)], 1u);
    }
    __syncthreads();
    for (unsigned p = 0; p < RS_MAXPASS; ++p) if (s_h[p][tid]) atomicAdd(&tot[p * RS_DIGITS + tid], (unsigned long long)s_h[p][tid]);
}
__global__ void __launch_bounds__(RS_THREADS) k_rs_onesweep(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint64_t n, unsigned shift, unsigned dmask,
                                                             const unsigned long long* __restrict__ tot /* this pass's 256 totals */,
                                                             unsigned long long* __restrict__ status /* [nblocks][256] zeroed */, unsigned long long* __restrict__ ticket,
                                                             uint64_t* __restrict__ kout, uint32_t* __restrict__ vout) {
    __shared__ uint32_t s_run[RS_DIGITS];                          // elements of each digit in earlier (round, wave) steps of this block; at the end: the block's counts
    __shared__ uint32_t s_cnt[RS_THREADS / 64][RS_DIGITS];         // this round: per wave
    __shared__ unsigned long long s_base[RS_DIGITS];               // where the block's elements of digit d begin in the output
    __shared__ unsigned long long s_scan[RS_THREADS / 64];
    __shared__ unsigned long long s_tile;
    const unsigned tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(ticket, 1ull);
    s_run[tid] = 0;
    __syncthreads();
    const uint64_t blk = s_tile, base = blk * RS_TILE;
    uint64_t key[RS_ITEMS]; uint32_t rank[RS_ITEMS];               // rank: among the equal digits before it in the block
#pragma unroll
    for (unsigned j = 0; j < RS_ITEMS; ++j) {
#pragma unroll
        for (unsigned w = 0; w < RS_THREADS / 64; ++w) s_cnt[w][tid] = 0;
        __syncthreads();
        const uint64_t i = base + (uint64_t)j * RS_THREADS + tid;
        const bool live = i < n;
        key[j] = live ? keys[i] : 0;
        const unsigned d = (unsigned)(key[j] >> shift) & dmask;
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (unsigned b = 0; b < RS_BITS; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const unsigned before = (unsigned)__builtin_popcountll(peers & ((1ull << lane) - 1));
        if (live && before == 0) s_cnt[wv][d] = (unsigned)__builtin_popcountll(peers);
        __syncthreads();
        unsigned off = s_run[d];
        for (unsigned w = 0; w < wv; ++w) off += s_cnt[w][d];
        rank[j] = off + before;
        __syncthreads();
        {   // the round's digits join the running counts (thread = digit)
            uint32_t add = 0;
#pragma unroll
            for (unsigned w = 0; w < RS_THREADS / 64; ++w) add += s_cnt[w][tid];
            s_run[tid] += add;
        }
        __syncthreads();
    }
    // ---- thread d: digits below d in the whole input (exclusive scan of the totals), + digit d in the blocks before this one (look-back)
    {
        const unsigned long long t = tot[tid];
        unsigned long long incl = t;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) { const unsigned long long o = __shfl_up(incl, dd); if ((int)lane >= dd) incl += o; }
        if (lane == 63) s_scan[wv] = incl;
        __syncthreads();
        unsigned long long wbase = 0;
        for (unsigned w = 0; w < wv; ++w) wbase += s_scan[w];
        const unsigned long long below = wbase + incl - t;
        const unsigned long long mine = s_run[tid];
        unsigned long long* st = status + blk * RS_DIGITS;
        unsigned long long excl = 0;
        if (blk == 0) __hip_atomic_store(&st[tid], ST_PREFIX | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else {
            __hip_atomic_store(&st[tid], ST_AGG | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (uint64_t p = blk; p-- > 0;) {
                unsigned long long sv;
                do { sv = __hip_atomic_load(&status[p * RS_DIGITS + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((sv >> 62) == 0);
                excl += sv & ST_VAL;
                if ((sv >> 62) == 2) break;
            }
            __hip_atomic_store(&st[tid], ST_PREFIX | (excl + mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_base[tid] = below + excl;
    }
    __syncthreads();
#pragma unroll
    for (unsigned j = 0; j < RS_ITEMS; ++j) {
        const uint64_t i = base + (uint64_t)j * RS_THREADS + tid;
        if (i < n) {
            const unsigned d = (unsigned)(key[j] >> shift) & dmask;
            const unsigned long long pos = s_base[d] + rank[j];
            kout[pos] = key[j]; vout[pos] = vals[i];
        }
    }
}

int own_sort_pairs(Ctx& c, uint64_t* keys, uint32_t* vals, uint64_t n, int begin_bit, int end_bit) {
    hipStream_t st = c.stream;
    if (end_bit - begin_bit > (int)(RS_BITS * RS_MAXPASS)) { c.err = "sort: more than 64 key bits"; return W2RAP_E_ARG; }
    const unsigned nblocks = (unsigned)((n + RS_TILE - 1) / RS_TILE);
    const unsigned npass = (unsigned)((end_bit - begin_bit + RS_BITS - 1) / RS_BITS);
    uint64_t* k2 = c.alloc<uint64_t>(n, false);
    uint32_t* v2 = c.alloc<uint32_t>(n, false);
    const uint64_t nstat = (uint64_t)nblocks * RS_DIGITS + 1;        // one pass's status words + its ticket
    unsigned long long* work = c.alloc<unsigned long long>(RS_MAXPASS * RS_DIGITS + npass * nstat, false);
    if (!k2 || !v2 || !work) return W2RAP_E_HIP;
    unsigned long long* tot = work; unsigned long long* stat = work + RS_MAXPASS * RS_DIGITS;
    W2_HIP(hipMemsetAsync(work, 0, (RS_MAXPASS * RS_DIGITS + npass * nstat) * 8, st));
    uint64_t *ka = keys, *kb = k2; uint32_t *va = vals, *vb = v2;
    c.pbegin("k_radix_sort_pairs");                       // (the passes of one sort, timed as one entry of the per-kernel profile)
    hipLaunchKernelGGL(k_rs_totals, dim3(std::min<unsigned>(nblocks, (unsigned)c.sm_count * 4)), dim3(RS_THREADS), 0, st, (const uint64_t*)keys, n, begin_bit, end_bit, tot);
    unsigned p = 0;
    for (int shift = begin_bit; shift < end_bit; shift += RS_BITS, ++p) {
        const unsigned dmask = (1u << std::min<int>(RS_BITS, end_bit - shift)) - 1u;       // the last digit may be narrower: the sort is by the asked bits ONLY (ties keep their order)
        hipLaunchKernelGGL(k_rs_onesweep, dim3(nblocks), dim3(RS_THREADS), 0, st, (const uint64_t*)ka, (const uint32_t*)va, n, (unsigned)shift, dmask,
                           (const unsigned long long*)(tot + p * RS_DIGITS), stat + p * nstat, stat + p * nstat + (nstat - 1), kb, vb);
        std::swap(ka, kb); std::swap(va, vb);
    }
    if (ka != keys) {
        W2_HIP(hipMemcpyAsync(keys, ka, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
        W2_HIP(hipMemcpyAsync(vals, va, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    }
    c.pend();
    W2_HIP(hipGetLastError());
    W2_HIP(hipStreamSynchronize(st));
    c.release(k2); c.release(v2); c.release(work);
    return 0;
}

// stable, in place, by the key bits [begin_bit, end_bit)
int sort_pairs_u64(Ctx& c, uint64_t* keys, uint32_t* vals, uint64_t n, int begin_bit, int end_bit) {
    if (n < 2 || end_bit <= begin_bit) return 0;
    if (n <= (1ull << 24) && !getenv("W2RAP_ROCPRIM_SORT")) return own_sort_pairs(c, keys, vals, n, begin_bit, end_bit);
    uint64_t* k2 = c.alloc<uint64_t>(n, false);
    uint32_t* v2 = c.alloc<uint32_t>(n, false);
    if (!k2 || !v2) return W2RAP_E_HIP;
    size_t tmp_bytes = 0;
    W2_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys, k2, vals, v2, n, begin_bit, end_bit, c.stream));
    void* tmp = tmp_alloc(c, tmp_bytes);
    if (!tmp) return W2RAP_E_HIP;
    c.pbegin("rocprim_radix_sort_pairs");                 // (the library's kernels, timed as one entry of the per-kernel profile)
    W2_HIP(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, k2, vals, v2, n, begin_bit, end_bit, c.stream));
    W2_HIP(hipMemcpyAsync(keys, k2, n * sizeof(uint64_t), hipMemcpyDeviceToDevice, c.stream));
    W2_HIP(hipMemcpyAsync(vals, v2, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c.stream));
    c.pend();
    W2_HIP(hipStreamSynchronize(c.stream));
    c.release(tmp); c.release(k2); c.release(v2);
    return 0;
}

}  // namespace w2

// ------------------------------------------------------------------------------------------------ self test (tests/test_gpu_prims.py)
// n pseudo-random pairs: the hand-written sort against rocPRIM's, the scans against sums done on the host.  -> 0, or the number of the check
// that failed (1 sort keys, 2 sort values / stability, 3 exclusive scan u32, 4 its total, 5 inclusive max scan, 6 maximum, 7 scan u64, 8 is-self scan)
extern "C" int w2rap_step2_selftest_prims(w2rap_step2_ctx* h, uint64_t n, uint64_t seed, int key_bits) {
    if (!h) return -1;
    w2::Ctx& c = h->c;
    if (hipSetDevice(c.device) != hipSuccess) return -1;
    std::vector<uint64_t> hk(n); std::vector<uint32_t> hv(n), hu(n);
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    for (uint64_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        hk[i] = (key_bits >= 64 || key_bits < 0) ? x : (x & ((1ull << key_bits) - 1));       // key_bits < 0: 64 random bits, sorted by the low -key_bits only
        if (i % 5 == 0 && i) hk[i] = hk[i - 1];                        // ties: stability shows in the values
        hv[i] = (uint32_t)i;
        hu[i] = (uint32_t)(x >> 40) & 1023u;
    }
    uint64_t *dk = c.alloc<uint64_t>(n + 1, false), *dk2 = c.alloc<uint64_t>(n + 1, false), *ds = c.alloc<uint64_t>(n + 2, false);
    uint32_t *dv = c.alloc<uint32_t>(n + 1, false), *dv2 = c.alloc<uint32_t>(n + 1, false), *du = c.alloc<uint32_t>(n + 1, false), *dm = c.alloc<uint32_t>(n + 1, false);
    if (!dk || !dk2 || !ds || !dv || !dv2 || !du || !dm) return -1;
    auto up = [&](void* d, const void* s, size_t b) { return b ? hipMemcpy(d, s, b, hipMemcpyHostToDevice) : hipSuccess; };
    auto dn = [&](void* d, const void* s, size_t b) { return b ? hipMemcpy(d, s, b, hipMemcpyDeviceToHost) : hipSuccess; };
    int bad = 0;
    if (up(dk, hk.data(), n * 8) != hipSuccess || up(dv, hv.data(), n * 4) != hipSuccess || up(dk2, hk.data(), n * 8) != hipSuccess || up(dv2, hv.data(), n * 4) != hipSuccess ||
        up(du, hu.data(), n * 4) != hipSuccess) return -1;
    const int eb = key_bits < 0 ? -key_bits : key_bits >= 64 ? 64 : key_bits;
    const uint64_t smask = eb >= 64 ? ~0ull : ((1ull << eb) - 1);
    if (n >= 2) {
        if (w2::own_sort_pairs(c, dk, dv, n, 0, eb)) return -1;
        // reference: a stable sort on the host
        std::vector<uint32_t> perm(n);
        for (uint64_t i = 0; i < n; ++i) perm[i] = (uint32_t)i;
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return (hk[a] & smask) < (hk[b] & smask); });
        std::vector<uint64_t> gk(n); std::vector<uint32_t> gv(n);
        if (dn(gk.data(), dk, n * 8) != hipSuccess || dn(gv.data(), dv, n * 4) != hipSuccess) return -1;
        for (uint64_t i = 0; i < n && !bad; ++i) { if (gk[i] != hk[perm[i]]) bad = 1; else if (gv[i] != perm[i]) bad = 2; }
    }
    if (!bad) {
        if (w2::exclusive_scan_u32_to_u64(c, du, ds, n)) return -1;
        std::vector<uint64_t> gs(n + 1);
        if (dn(gs.data(), ds, (n + 1) * 8) != hipSuccess) return -1;
        uint64_t run = 0;
        for (uint64_t i = 0; i < n && !bad; ++i) { if (gs[i] != run) bad = 3; run += hu[i]; }
        if (!bad && gs[n] != run) bad = 4;
    }
    if (!bad && n) {
        if (w2::inclusive_max_scan_u32(c, du, dm, n)) return -1;
        std::vector<uint32_t> gm(n);
        if (dn(gm.data(), dm, n * 4) != hipSuccess) return -1;
        uint32_t mx = 0;
        for (uint64_t i = 0; i < n && !bad; ++i) { mx = std::max(mx, hu[i]); if (gm[i] != mx) bad = 5; }
        uint32_t r = 0;
        if (w2::max_u32(c, du, n, &r)) return -1;
        if (!bad && r != mx) bad = 6;
    }
    if (!bad) {
        if (w2::exclusive_scan_u64(c, dk2, ds, n)) return -1;          // (wraps modulo 2^62 in the status words only if the sum reaches 2^62: keys of <= 40 bits here)
        std::vector<uint64_t> gs(n + 1);
        if (dn(gs.data(), ds, (n + 1) * 8) != hipSuccess) return -1;
        uint64_t run = 0;
        if (key_bits > 0 && key_bits <= 40) for (uint64_t i = 0; i <= n && !bad; ++i) { if (gs[i] != run) bad = 7; if (i < n) run += hk[i]; }
    }
    if (!bad) {
        std::vector<uint32_t> a(n);
        for (uint64_t i = 0; i < n; ++i) a[i] = (hu[i] & 3) ? (uint32_t)i : (uint32_t)(i / 2);
        if (up(dv2, a.data(), n * 4) != hipSuccess) return -1;
        if (w2::exclusive_scan_is_self(c, dv2, ds, n)) return -1;
        std::vector<uint64_t> gs(n + 1);
        if (dn(gs.data(), ds, (n + 1) * 8) != hipSuccess) return -1;
        uint64_t run = 0;
        for (uint64_t i = 0; i <= n && !bad; ++i) { if (gs[i] != run) bad = 8; if (i < n) run += a[i] == (uint32_t)i; }
    }
    for (void* p : {(void*)dk, (void*)dk2, (void*)ds, (void*)dv, (void*)dv2, (void*)du, (void*)dm}) c.release(p);
    return bad;
}
