// issue_ubench.hip -- instruction issue rates on MI355X by unit and by waves per CU: is the scalar ALU shared by the CU's four
// SIMDs, how many cycles does one wave need per instruction, what does a divergent `if` cost against predication?
//   hipcc --offload-arch=gfx950 -O3 -o tools/issue_ubench tools/issue_ubench.hip && ./tools/issue_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ITERS = 2000;
// MODE 0: 32 independent SALU adds per iteration; 1: 32 independent VALU adds; 2: 16 SALU + 16 VALU interleaved;
// 3: 32 dependent SALU; 4: divergent-if shape (v_cmp, s_and_saveexec, v_add, s_or exec); 5: the same predicated
template <int MODE>
__global__ void k(unsigned long long* out, uint32_t seed) {
    uint32_t s0 = __builtin_amdgcn_readfirstlane(seed), s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
    uint32_t v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    const uint32_t sb = __builtin_amdgcn_readfirstlane(seed * 3 + 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("s_add_u32 %0, %0, %8\n s_add_u32 %1, %1, %8\n s_add_u32 %2, %2, %8\n s_add_u32 %3, %3, %8\n"
                             "s_add_u32 %4, %4, %8\n s_add_u32 %5, %5, %8\n s_add_u32 %6, %6, %8\n s_add_u32 %7, %7, %8\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(sb) : "scc");
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "s"(sb));
        } else if (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("v_add_u32 %0, %0, %8\n s_add_u32 %4, %4, %8\n v_add_u32 %1, %1, %8\n s_add_u32 %5, %5, %8\n"
                             "v_add_u32 %2, %2, %8\n s_add_u32 %6, %6, %8\n v_add_u32 %3, %3, %8\n s_add_u32 %7, %7, %8\n"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(sb) : "scc");
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                asm volatile("s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n"
                             "s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n"
                             : "+s"(s0) : "s"(sb) : "scc");
        } else if (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_u32 %0, %0, %2\n s_or_b64 exec, exec, s[20:21]\n"
                             : "+v"(v0) : "v"(v1), "s"(sb) : "vcc", "scc", "s20", "s21");
        } else if (MODE == 5) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_cmp_gt_u32 vcc, %0, %2\n v_add_u32 %1, %0, %3\n v_cndmask_b32 %0, %0, %1, vcc\n"
                             : "+v"(v0), "+v"(v2) : "v"(v1), "s"(sb) : "vcc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) atomicAdd(out, t1 - t0);
    if (s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 0x12345678u) out[1] = 1;
}

template <int MODE> void run(const char* what, unsigned long long* d, int per_iter) {
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 1u);
        hipMemset(d, 0, 16);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 7u);
        hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        const int waves = threads / 64;
        const double per_wave = (double)h[0] / (256.0 * waves) / ITERS;          // clocks per iteration as seen by one wave
        printf("%-44s %2d waves/CU: %7.1f clocks per wave-iteration (%d instr) = %5.2f clk/instr/wave, %5.2f instr/clk/CU\n", what, waves, per_wave, per_iter,
               per_wave / per_iter, per_iter * waves / per_wave);
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    unsigned long long* d; hipMalloc(&d, 16);
    run<0>("32 independent SALU", d, 32);
    run<3>("32 dependent SALU", d, 32);
    run<1>("32 independent VALU (v_add_u32)", d, 32);
    run<2>("16 VALU + 16 SALU interleaved", d, 32);
    run<4>("8 x (v_cmp, saveexec, v_add, s_or exec)", d, 32);
    run<5>("8 x (v_cmp, v_add, v_cndmask)", d, 24);
    return 0;
}
