// step2_prims.hip -- device-wide utility primitives (stable radix sort of (u64,u32)
// pairs, exclusive scans, max) used by the graph phases on E- and S-sized arrays.
// These are not the hot path (SURVEY.md 8a a7/a8 are <5 % of Step 2); they wrap rocPRIM.
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "ctx.h"

namespace w2 {

// rocPRIM temp storage comes from the context's block pool (sizes repeat from run to run), never from hipMalloc/hipFree:
// hipFree synchronises the whole device and would stall the side stream's overlapped work.
static void* tmp_alloc(Ctx& c, size_t bytes) { return c.alloc<uint8_t>(bytes ? bytes : 16, false); }

int sort_pairs_u64(Ctx& c, uint64_t* keys, uint32_t* vals, uint64_t n, int begin_bit, int end_bit) {
    if (n < 2) return 0;
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

// ---- small inputs: ONE stable comparison sort of a permutation by up to three u64 words (most significant first) instead of 8..24
// radix passes with their launches, copies and host round trips -- the edge order, the edge ends and the adjacency lists of a graph
// with 10^4 .. 10^5 objects are sorted in a few tens of microseconds.  Ties keep the order of the object ids (perm starts as iota).
struct PermLess {
    const uint64_t *w0, *w1, *w2;
    __device__ bool operator()(uint32_t a, uint32_t b) const {
        const uint64_t x0 = w0[a], y0 = w0[b];
        if (x0 != y0) return x0 < y0;
        if (w1) { const uint64_t x1 = w1[a], y1 = w1[b]; if (x1 != y1) return x1 < y1; }
        if (w2) { const uint64_t x2 = w2[a], y2 = w2[b]; if (x2 != y2) return x2 < y2; }
        return a < b;
    }
};
__global__ void __launch_bounds__(256) k_iota_u32(uint64_t n, uint32_t* __restrict__ a) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = (uint32_t)i;
}
// perm <- the ids 0..n-1 ordered by (w0, w1, w2) [w1, w2 may be null], ties by id; true if it was done here (n small enough)
int sort_ids_by_words(Ctx& c, uint32_t* perm, uint64_t n, const uint64_t* w0, const uint64_t* w1, const uint64_t* w2, bool* done) {
    *done = false;
    if (n > SMALL_SORT_MAX || getenv("W2RAP_NO_SMALL_SORT")) return 0;
    *done = true;
    if (!n) return 0;
    uint32_t* in = c.alloc<uint32_t>(n, false);
    if (!in) return W2RAP_E_HIP;
    hipLaunchKernelGGL(k_iota_u32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, n, in);
    size_t tmp_bytes = 0;
    const PermLess less{w0, w1, w2};
    W2_HIP(rocprim::merge_sort(nullptr, tmp_bytes, in, perm, n, less, c.stream));
    void* tmp = tmp_alloc(c, tmp_bytes);
    if (!tmp) return W2RAP_E_HIP;
    c.pbegin("rocprim_merge_sort");
    W2_HIP(rocprim::merge_sort(tmp, tmp_bytes, in, perm, n, less, c.stream));
    c.pend();
    W2_HIP(hipStreamSynchronize(c.stream));
    c.release(tmp); c.release(in);
    return 0;
}

__global__ void k_store_total_u32(const uint32_t* in, uint64_t* out, uint64_t n) {
    if (n) out[n] = out[n - 1] + in[n - 1]; else out[0] = 0;
}
__global__ void k_store_total_u64(const uint64_t* in, uint64_t* out, uint64_t n) {
    if (n) out[n] = out[n - 1] + in[n - 1]; else out[0] = 0;
}

struct U32ToU64 { __device__ uint64_t operator()(uint32_t x) const { return x; } };

int exclusive_scan_u32_to_u64(Ctx& c, const uint32_t* in, uint64_t* out, uint64_t n) {
    if (n) {
        auto it = rocprim::make_transform_iterator(in, U32ToU64());
        size_t tmp_bytes = 0;
        W2_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c.stream));
        void* tmp = tmp_alloc(c, tmp_bytes);
        if (!tmp) return W2RAP_E_HIP;
        W2_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c.stream));
        hipLaunchKernelGGL(k_store_total_u32, 1, 1, 0, c.stream, in, out, n);
        W2_HIP(hipStreamSynchronize(c.stream));
        c.release(tmp);
    } else {
        hipLaunchKernelGGL(k_store_total_u32, 1, 1, 0, c.stream, in, out, n);
        W2_HIP(hipStreamSynchronize(c.stream));
    }
    return 0;
}

int exclusive_scan_u64(Ctx& c, const uint64_t* in, uint64_t* out, uint64_t n) {
    if (n) {
        size_t tmp_bytes = 0;
        W2_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c.stream));
        void* tmp = tmp_alloc(c, tmp_bytes);
        if (!tmp) return W2RAP_E_HIP;
        W2_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, in, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c.stream));
        hipLaunchKernelGGL(k_store_total_u64, 1, 1, 0, c.stream, in, out, n);
        W2_HIP(hipStreamSynchronize(c.stream));
        c.release(tmp);
    } else {
        hipLaunchKernelGGL(k_store_total_u64, 1, 1, 0, c.stream, in, out, n);
        W2_HIP(hipStreamSynchronize(c.stream));
    }
    return 0;
}

int inclusive_max_scan_u32(Ctx& c, const uint32_t* in, uint32_t* out, uint64_t n) {
    if (!n) return 0;
    size_t tmp_bytes = 0;
    W2_HIP(rocprim::inclusive_scan(nullptr, tmp_bytes, in, out, n, rocprim::maximum<uint32_t>(), c.stream));
    void* tmp = tmp_alloc(c, tmp_bytes);
    if (!tmp) return W2RAP_E_HIP;
    W2_HIP(rocprim::inclusive_scan(tmp, tmp_bytes, in, out, n, rocprim::maximum<uint32_t>(), c.stream));
    W2_HIP(hipStreamSynchronize(c.stream));
    c.release(tmp);
    return 0;
}

int max_u32(Ctx& c, const uint32_t* in, uint64_t n, uint32_t* result) {
    *result = 0;
    if (!n) return 0;
    uint32_t* d_out = c.alloc<uint32_t>(1, false);
    if (!d_out) return W2RAP_E_HIP;
    size_t tmp_bytes = 0;
    W2_HIP(rocprim::reduce(nullptr, tmp_bytes, in, d_out, (uint32_t)0, n, rocprim::maximum<uint32_t>(), c.stream));
    void* tmp = tmp_alloc(c, tmp_bytes);
    if (!tmp) return W2RAP_E_HIP;
    W2_HIP(rocprim::reduce(tmp, tmp_bytes, in, d_out, (uint32_t)0, n, rocprim::maximum<uint32_t>(), c.stream));
    W2_HIP(hipMemcpyAsync(result, d_out, sizeof(uint32_t), hipMemcpyDeviceToHost, c.stream));
    W2_HIP(hipStreamSynchronize(c.stream));
    c.release(tmp); c.release(d_out);
    return 0;
}

}  // namespace w2
