// lds_ubench.hip -- measured cost of the LDS access patterns k_count_buckets is built from (MI355X).
// 16 waves per CU (one 1024-thread block per CU), random addresses in a 64-KiB LDS region.
//   hipcc --offload-arch=gfx950 -O3 -o lds_ubench tools/lds_ubench.hip && ./lds_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int ITERS = 512;
template <int MODE>
__global__ void __launch_bounds__(1024) k(unsigned long long* out, uint32_t seed) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr unsigned N = 16384;                       // dwords (64 KiB)
    uint64_t* lds64 = reinterpret_cast<uint64_t*>(lds);
    for (unsigned i = threadIdx.x; i < N; i += 1024) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = (threadIdx.x + blockIdx.x * 1024u) * 2246822519u + seed, acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        x = x * 1664525u + 1013904223u;
        const unsigned r = (x >> 10);
        if (MODE == 0) { uint32_t v = __hip_atomic_load(&lds[r & (N - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); x += v; }
        if (MODE == 1) { uint64_t v = __hip_atomic_load(&lds64[r & (N / 2 - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); x += (uint32_t)v; }
        if (MODE == 2) { typedef uint32_t u4 __attribute__((ext_vector_type(4))); u4 v; const uint32_t a = (uint32_t)(uintptr_t)&lds[(r & (N / 4 - 1)) * 4];
                         asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory"); x += v.x ^ v.w; }
        if (MODE == 3) { atomicAdd(&lds[r & (N - 1)], 1u); }
        if (MODE == 4) { uint32_t v = atomicAdd(&lds[r & (N - 1)], 1u); x += v; }
        if (MODE == 5) { uint32_t v = atomicCAS(&lds[r & (N - 1)], x, 1u); x += v; }
        if (MODE == 6) { const uint64_t* g = &lds64[(r & (N / 8 - 1)) * 4];
                         uint64_t a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP),
                                  c = __hip_atomic_load(g + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), d = __hip_atomic_load(g + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                         x += (uint32_t)(a ^ b ^ c ^ d); }
        if (MODE == 7) { atomicAdd(&lds[r & (N - 1)], 1u); atomicOr(&lds[r & (N - 1)], 0x1000000u);
                         uint32_t v = __hip_atomic_load(&lds[(r >> 3) & (N - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); x += v; }
        if (MODE == 8) { uint32_t v = __hip_atomic_load(&lds[(r & 63) + 64 * (threadIdx.x >> 6)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); x += v; }   // conflict-free-ish
        if (MODE == 9) { unsigned long long v = atomicCAS((unsigned long long*)&lds64[r & (N / 2 - 1)], (unsigned long long)x, 1ull); x += (uint32_t)v; }
        if (MODE == 10) { // pure ALU chain of ~40 dependent integer ops, for the issue rate with 16 waves
#pragma unroll
            for (int j = 0; j < 20; ++j) { x = (x ^ (x >> 7)) + 0x9E3779B9u; x = __builtin_amdgcn_alignbit(x, x, 13); } }
        if (MODE == 11) { // the same with 64-bit shifts/compares
            uint64_t y = ((uint64_t)x << 32) | r;
#pragma unroll
            for (int j = 0; j < 20; ++j) { y = (y ^ (y >> 7)) + 0x9E3779B97F4A7C15ull; y = (y << 13) | (y >> 51); }
            x += (uint32_t)y ^ (uint32_t)(y >> 32); }
        acc ^= x;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);
    if (acc == 0x12345678u) out[1] = acc;
}

template <int MODE> void run(const char* what, unsigned long long* d) {
    hipMemset(d, 0, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 65536, 0, d, 1u);     // warm-up
    hipMemset(d, 0, 16);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 65536, 0, d, 7u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    const double per = (double)h[0] / (256.0 * 16) / ITERS;
    printf("%-58s %8.1f clocks per wave-iteration, %6.1f clocks per CU per wave-instruction-group (16 waves), kernel %.3f ms\n", what, per, per / 16, ms);
}

int main() {
    unsigned long long* d; hipMalloc(&d, 16);
    run<8>("ds_read_b32 wave-private 256 B window (dependent)", d);
    run<0>("ds_read_b32 random (dependent)", d);
    run<1>("ds_read_b64 random (dependent)", d);
    run<2>("ds_read_b128 random (dependent)", d);
    run<6>("4 x ds_read_b64 of one random 32-B group (dependent)", d);
    run<3>("ds_add_u32 random, no return (independent)", d);
    run<4>("ds_add_rtn_u32 random (dependent)", d);
    run<5>("ds_cmpst_rtn_b32 random (dependent)", d);
    run<9>("ds_cmpst_rtn_b64 random (dependent)", d);
    run<7>("ds_add + ds_or same random dword, then dependent read", d);
    run<10>("40 dependent 32-bit VALU ops", d);
    run<11>("40 dependent 64-bit VALU ops (shift/add/xor)", d);
    return 0;
}
