// Issue-cost micro-benchmark with inline asm (nothing can be optimised away).
// Reports ns and cycles per wave-instruction per SIMD at 1, 2 and 4 resident waves per SIMD,
// cycles derived from the clock the chip actually held (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
constexpr int ITERS = 2048;

#define REP8(X) X X X X X X X X
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned long long *clk)
{
    unsigned a = threadIdx.x * 2654435761u, b = a ^ 0x9E3779B9u;
    unsigned r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    unsigned long long q0 = a, q1 = b, q2 = a + b, q3 = a * 3ull;
    unsigned long long src = ((unsigned long long)b << 32) | a;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; it++) {
        if (OP == 0) asm volatile(REP8("v_sad_u8 %0, %8, %9, %0\n v_sad_u8 %1, %8, %9, %1\n v_sad_u8 %2, %8, %9, %2\n v_sad_u8 %3, %8, %9, %3\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        if (OP == 1) asm volatile(REP8("v_qsad_pk_u16_u8 %0, %4, %5, %0\n v_qsad_pk_u16_u8 %1, %4, %5, %1\n v_qsad_pk_u16_u8 %2, %4, %5, %2\n v_qsad_pk_u16_u8 %3, %4, %5, %3\n") : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(src), "v"(b));
        if (OP == 2) asm volatile(REP8("v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %8\n v_mov_b32 %3, %9\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        if (OP == 3) asm volatile(REP8("v_min3_u32 %0, %0, %8, %9\n v_min3_u32 %1, %1, %8, %9\n v_min3_u32 %2, %2, %8, %9\n v_min3_u32 %3, %3, %8, %9\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        if (OP == 4) asm volatile(REP8("v_lshl_or_b32 %0, %0, 16, %8\n v_lshl_or_b32 %1, %1, 16, %9\n v_lshl_or_b32 %2, %2, 16, %8\n v_lshl_or_b32 %3, %3, 16, %9\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        if (OP == 5) asm volatile(REP8("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %9\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %9\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        if (OP == 6) asm volatile(REP8("v_sad_hi_u8 %0, %8, %9, %0\n v_sad_hi_u8 %1, %8, %9, %1\n v_sad_hi_u8 %2, %8, %9, %2\n v_sad_hi_u8 %3, %8, %9, %3\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        if (OP == 7) asm volatile(REP8("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        if (OP == 8) asm volatile(REP8("v_mqsad_u32_u8 %0, %1, %2, %0\n") : "+v"(*(__uint128_t *)&q0) : "v"(src), "v"(b));
        if (OP == 9) asm volatile(REP8("v_sad_u16 %0, %8, %9, %0\n v_sad_u16 %1, %8, %9, %1\n v_sad_u16 %2, %8, %9, %2\n v_sad_u16 %3, %8, %9, %3\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#define OP3(N, NAME) if (OP == N) asm volatile(REP8(NAME " %0, %8, %9, %0\n " NAME " %1, %8, %9, %1\n " NAME " %2, %8, %9, %2\n " NAME " %3, %8, %9, %3\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#define OP2(N, NAME) if (OP == N) asm volatile(REP8(NAME " %0, %8, %0\n " NAME " %1, %9, %1\n " NAME " %2, %8, %2\n " NAME " %3, %9, %3\n") : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        OP3(10, "v_lerp_u8")
        OP3(11, "v_alignbyte_b32")
        OP3(12, "v_perm_b32")
        OP2(13, "v_and_b32")
        OP2(14, "v_xor_b32")
        OP2(15, "v_lshrrev_b32")
        OP2(16, "v_min_u32")
        OP2(17, "v_pk_add_u16")
        OP3(18, "v_bfe_u32")
        OP3(19, "v_add3_u32")
        OP3(20, "v_xad_u32")
        OP2(21, "v_sub_u32")
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (unsigned)(q0 + q1 + q2 + q3);
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int OP> void run(const char *name, int n_per_iter, unsigned *d_out, unsigned long long *d_clk)
{
    for (int waves_per_simd : {1, 2, 4}) {
        const int blocks = 256 * waves_per_simd;  // 256 CUs x waves_per_simd workgroups of 4 waves (1 per SIMD)
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_clk);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 5; r++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_clk);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        unsigned long long c[2]; CHECK(hipMemcpy(c, d_clk, 16, hipMemcpyDeviceToHost));
        const double ghz = (double)c[0] / ((double)c[1] * 10.0);  // memrealtime ticks at 100 MHz
        const double instr = (double)waves_per_simd * ITERS * n_per_iter;  // per SIMD
        const double ns = best * 1e6 / instr;
        printf("%-20s waves/SIMD %d: %7.3f ms  %6.3f ns/instr/SIMD  clock %.2f GHz  => %5.2f cycles\n", name, waves_per_simd, best, ns, ghz, ns * ghz);
    }
}

int main()
{
    unsigned *d_out; unsigned long long *d_clk;
    CHECK(hipMalloc(&d_out, 1024 * 256 * 4)); CHECK(hipMalloc(&d_clk, 16));
    run<5>("v_add_u32", 32, d_out, d_clk);
    run<2>("v_mov_b32", 32, d_out, d_clk);
    run<4>("v_lshl_or_b32", 32, d_out, d_clk);
    run<7>("v_and_or_b32", 32, d_out, d_clk);
    run<3>("v_min3_u32", 32, d_out, d_clk);
    run<0>("v_sad_u8", 32, d_out, d_clk);
    run<6>("v_sad_hi_u8", 32, d_out, d_clk);
    run<9>("v_sad_u16", 32, d_out, d_clk);
    run<1>("v_qsad_pk_u16_u8", 32, d_out, d_clk);
    run<8>("v_mqsad_u32_u8", 8, d_out, d_clk);
    run<10>("v_lerp_u8", 32, d_out, d_clk);
    run<11>("v_alignbyte_b32", 32, d_out, d_clk);
    run<12>("v_perm_b32", 32, d_out, d_clk);
    run<13>("v_and_b32", 32, d_out, d_clk);
    run<14>("v_xor_b32", 32, d_out, d_clk);
    run<15>("v_lshrrev_b32", 32, d_out, d_clk);
    run<16>("v_min_u32", 32, d_out, d_clk);
    run<17>("v_pk_add_u16", 32, d_out, d_clk);
    run<18>("v_bfe_u32", 32, d_out, d_clk);
    run<19>("v_add3_u32", 32, d_out, d_clk);
    run<20>("v_xad_u32", 32, d_out, d_clk);
    run<21>("v_sub_u32", 32, d_out, d_clk);
    return 0;
}
