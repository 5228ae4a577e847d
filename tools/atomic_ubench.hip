// atomic_ubench.hip -- device-scope atomic throughput on MI355X for the partition kernels' access pattern:
// N atomics to random dwords of a table of T dwords, fire-and-forget vs returning, 32-bit add vs 64-bit CAS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* tab, uint64_t mask, uint64_t n, uint32_t* sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        uint32_t* p = tab + (x & mask);
        if (MODE == 0) atomicAdd(p, 1u);
        if (MODE == 1) acc += atomicAdd(p, 1u);
        if (MODE == 2) acc += (uint32_t)atomicCAS((unsigned long long*)(tab + ((x & mask) & ~1ull)), 0ull, x);
        if (MODE == 3) *p = (uint32_t)x;                       // plain scattered 4-B stores
        if (MODE == 4) acc += *p;                              // plain scattered 4-B loads
        if (MODE == 5 || MODE == 6) {                          // a PRIVATE table per XCD (the wave asks the hardware which XCD it runs on): do atomics stay in that XCD's L2?
            uint32_t xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            uint32_t* q = tab + ((uint64_t)(xcc & 7u) * ((mask + 1) >> 3)) + (x & (mask >> 3));
            if (MODE == 5) acc += atomicAdd(q, 1u);
            else acc += __hip_atomic_fetch_add(q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    if (acc == 0x12345u) *sink = acc;
}
template <int MODE> void run(const char* what, uint32_t* tab, uint64_t tdw, uint64_t n, uint32_t* sink, int grid) {
    hipMemset(tab, 0, tdw * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, tab, tdw - 1, n / 8, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, tab, tdw - 1, n, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    printf("%-44s table %8.1f MB: %7.2f ms for %llu M ops = %6.2f G ops/s\n", what, tdw * 4 / 1e6, ms, (unsigned long long)(n / 1000000), n / ms / 1e6);
}
int main() {
    uint32_t *tab, *sink; const uint64_t big = 1ull << 30;   // 4 GiB
    hipMalloc(&tab, big * 4); hipMalloc(&sink, 4);
    const uint64_t n = 184000000ull;
    for (uint64_t tdw : {1ull << 20, 1ull << 23, 1ull << 26, 1ull << 30}) {
        run<0>("atomicAdd u32, no return", tab, tdw, n, sink, 256 * 8);
        run<1>("atomicAdd u32, returning", tab, tdw, n, sink, 256 * 8);
        run<2>("atomicCAS u64, returning", tab, tdw, n, sink, 256 * 8);
        run<3>("plain 4-B store", tab, tdw, n, sink, 256 * 8);
        run<4>("plain 4-B load", tab, tdw, n, sink, 256 * 8);
        run<5>("atomicAdd returning, table per XCD (agent scope)", tab, tdw, n, sink, 256 * 8);
        run<6>("atomicAdd returning, table per XCD (workgroup scope)", tab, tdw, n, sink, 256 * 8);
    }
    return 0;
}
