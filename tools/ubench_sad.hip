// Micro-benchmark + semantics self-check of the gfx950 byte-SAD instructions the
// search kernels are built on.  Prints wave-instruction issue cost per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_sad.hip -o tools/ubench_sad
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;

template <int OP>
__global__ __launch_bounds__(256) void k_rate(uint32_t *out, uint32_t seed)
{
    uint32_t a = seed * (threadIdx.x + 1), b = a ^ 0x9E3779B9u, c = a * 7u + 3u;
    u64 q[8]; uint32_t r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { q[i] = (u64)(a + i) << 20; r[i] = b + i; }
    const u64 src = ((u64)b << 32) | a;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) r[i] = __builtin_amdgcn_sad_u8(a, r[i], c);            // dependent through r
            if (OP == 1) r[i] = __builtin_amdgcn_sad_hi_u8(a, c, r[i]);
            if (OP == 2) q[i] = __builtin_amdgcn_qsad_pk_u16_u8(src, c, q[i]);
            if (OP == 3) q[i] = __builtin_amdgcn_mqsad_pk_u16_u8(src, c, q[i]);
            if (OP == 4) r[i] = __builtin_amdgcn_alignbyte(r[i], a, 3u);
            if (OP == 5) r[i] = min(min(r[i], a + i), c);
            if (OP == 6) r[i] = r[i] + a;
            if (OP == 7) r[i] = __builtin_amdgcn_perm(r[i], a, 0x07020500u);
            if (OP == 8) r[i] = (r[i] << 16) | c;
            if (OP == 9) r[i] = __builtin_amdgcn_msad_u8(a, r[i], c);
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += r[i] + (uint32_t)q[i] + (uint32_t)(q[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_semantics(const uint32_t *in, uint32_t *out)
{
    const uint32_t w0 = in[0], w1 = in[1], ref = in[2];
    const u64 q = __builtin_amdgcn_qsad_pk_u16_u8(((u64)w1 << 32) | w0, ref, 0x0004000300020001ull);
    out[0] = (uint32_t)q; out[1] = (uint32_t)(q >> 32);
    out[2] = __builtin_amdgcn_sad_u8(w0, ref, 5u);
    out[3] = __builtin_amdgcn_sad_hi_u8(w0, ref, 5u);
    out[4] = __builtin_amdgcn_alignbyte(w1, w0, 1u);
    out[5] = __builtin_amdgcn_perm(w1, w0, 0x07020500u);
}

static uint32_t sad4(const uint8_t *a, const uint8_t *b) { uint32_t s = 0; for (int i = 0; i < 4; i++) s += abs((int)a[i] - (int)b[i]); return s; }

template <int OP> double run(const char *name, uint32_t *d_out, int blocks)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 12345u + rep);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    // wave-instructions per SIMD: each block = 4 waves, one per SIMD of its CU
    const double wave_instr_per_simd = (double)blocks / 256.0 * ITERS * 8.0;  // blocks/256 waves per SIMD
    const double ns_per = best * 1e6 / wave_instr_per_simd;
    printf("%-22s %8.3f ms  %6.3f ns per wave-instr per SIMD  (= %5.2f cycles @2.4GHz)\n", name, best, ns_per, ns_per * 2.4);
    return ns_per;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  arch %s  CUs %d  clock %d kHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    // semantics
    uint32_t h_in[3] = {0x04030201u, 0x08070605u, 0x0A090807u}, h_out[6];
    uint32_t *d_in, *d_o; CHECK(hipMalloc(&d_in, 12)); CHECK(hipMalloc(&d_o, 24));
    CHECK(hipMemcpy(d_in, h_in, 12, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_semantics, dim3(1), dim3(1), 0, 0, d_in, d_o);
    CHECK(hipMemcpy(h_out, d_o, 24, hipMemcpyDeviceToHost));
    uint8_t bytes[8]; for (int i = 0; i < 4; i++) { bytes[i] = h_in[0] >> (8 * i); bytes[4 + i] = h_in[1] >> (8 * i); }
    uint8_t rb[4]; for (int i = 0; i < 4; i++) rb[i] = h_in[2] >> (8 * i);
    int ok = 1;
    for (int o = 0; o < 4; o++) {
        const uint32_t exp = sad4(bytes + o, rb) + (o + 1);
        const uint32_t got = (o < 2 ? h_out[0] >> (16 * o) : h_out[1] >> (16 * (o - 2))) & 0xFFFF;
        printf("qsad offset %d: got %u expected %u\n", o, got, exp); ok &= got == exp;
    }
    printf("sad_u8 %u (exp %u)  sad_hi_u8 0x%08x (exp 0x%08x)\n", h_out[2], sad4(bytes, rb) + 5, h_out[3], (sad4(bytes, rb) << 16) + 5);
    ok &= h_out[2] == sad4(bytes, rb) + 5 && h_out[3] == (sad4(bytes, rb) << 16) + 5;
    printf("alignbyte(w1,w0,1) 0x%08x (exp 0x05040302)  perm 0x%08x\n", h_out[4], h_out[5]); ok &= h_out[4] == 0x05040302u;
    printf("semantics %s\n", ok ? "OK" : "MISMATCH");

    const int blocks = 256 * 16;  // 16 workgroups per CU in total, 4 resident waves per SIMD at a time
    uint32_t *d_out; CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 4));
    run<6>("v_add_u32", d_out, blocks);
    run<0>("v_sad_u8", d_out, blocks);
    run<1>("v_sad_hi_u8", d_out, blocks);
    run<9>("v_msad_u8", d_out, blocks);
    run<2>("v_qsad_pk_u16_u8", d_out, blocks);
    run<3>("v_mqsad_pk_u16_u8", d_out, blocks);
    run<4>("v_alignbyte_b32", d_out, blocks);
    run<5>("v_min3_u32", d_out, blocks);
    run<7>("v_perm_b32", d_out, blocks);
    run<8>("v_lshl_or_b32", d_out, blocks);
    return ok ? 0 : 1;
}
