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

// out[i] = number of j < i with a[j] == j (Step 3: occurrences that are their own representative), out[n] = their number; the flags are
// never materialised
struct IsSelf {
    const uint32_t* a;
    __device__ uint64_t operator()(uint64_t i) const { return a[i] == (uint32_t)i ? 1ull : 0ull; }
};
__global__ void k_store_total_self(const uint32_t* a, uint64_t* out, uint64_t n) {
    if (n) out[n] = out[n - 1] + (a[n - 1] == (uint32_t)(n - 1) ? 1ull : 0ull); else out[0] = 0;
}
int exclusive_scan_is_self(Ctx& c, const uint32_t* a, uint64_t* out, uint64_t n) {
    if (n) {
        auto it = rocprim::make_transform_iterator(rocprim::counting_iterator<uint64_t>(0), IsSelf{a});
        size_t tmp_bytes = 0;
        W2_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c.stream));
        void* tmp = tmp_alloc(c, tmp_bytes);
        if (!tmp) return W2RAP_E_HIP;
        W2_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, it, out, (uint64_t)0, n, rocprim::plus<uint64_t>(), c.stream));
        hipLaunchKernelGGL(k_store_total_self, 1, 1, 0, c.stream, a, out, n);
        W2_HIP(hipStreamSynchronize(c.stream));
        c.release(tmp);
    } else {
        hipLaunchKernelGGL(k_store_total_self, 1, 1, 0, c.stream, a, out, n);
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
