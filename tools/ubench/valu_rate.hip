// Micro-benchmark: VALU issue rate on gfx950 for the instruction kinds the stitching kernels use.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed)
{
    unsigned a = threadIdx.x + seed, b = blockIdx.x * 3 + 1, c = a ^ 0x55u, d = b + 7;
    float fa = a * 0.5f, fb = b * 0.25f, fc = 1.0001f, fd = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) { fa = fa * fc + fd; fb = fb * fc + fd; }                                   // v_fma_f32 (contracted)
            if (KIND == 1) { a = (a & 0xFFFFFFu) * 3u + c; b = (b & 0xFFFFFFu) * 5u + d; }              // and + mad_u32_u24
            if (KIND == 2) { a = (a >> 3) ^ c; b = (b << 1) + d; }                                     // shifts / logic / add
            if (KIND == 3) { a = a * c + 1u; b = b * d + 3u; }                                         // 32-bit mul (v_mul_lo_u32)
            if (KIND == 4) { fa = __builtin_amdgcn_rcpf(fa) + fd; fb = rintf(fb * fc); }               // rcp / rndne
            if (KIND == 5) { a = __builtin_amdgcn_alignbyte(a, c, b & 3u); b = __builtin_amdgcn_ubfe(b, 3, 9) + d; }  // alignbyte / bfe
            if (KIND == 6) { fa = (float)(int)a * fc; a = (unsigned)(int)fa + c; }                      // cvt i32<->f32 + mul
            if (KIND == 7) { a = __builtin_amdgcn_udot4(a, c, b, false); b = __builtin_amdgcn_udot4(b, d, a, false); }   // v_dot4_u32_u8
            if (KIND == 8) { a = __builtin_amdgcn_cvt_pk_u8_f32(fa, 1u, a); fa = fa + fd; b = __builtin_amdgcn_cvt_pk_u8_f32(fb, 2u, b); fb = fb + fd; }   // cvt_pk_u8 + add
            if (KIND == 9) { a = __umul24(a, c) + 512u; b = (__umul24(b, d) + a) >> 10; }               // mul24 / mad24 / shift
            if (KIND == 10) { a = min(max((int)a, -32768), 32767) + c; b = ((int)b >> 16) + d; }        // med3 / ashr / add
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (unsigned)fa + (unsigned)fb;
}

template <int KIND> double run(const char* name, int ops_per_iter_per_unroll)
{
    const int blocks = 256 * 8, iters = 2000;
    unsigned* d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * iters * 16 * ops_per_iter_per_unroll;
    double per_simd_cycle = wave_instr / (ms * 1e-3) / (1024.0 * 2.4e9);
    printf("%-28s %8.3f ms  %.3f wave-instr/cycle/SIMD (=> %.2f cycles per wave64 instr at 2.4 GHz)\n", name, ms,
           per_simd_cycle, 1.0 / per_simd_cycle);
    hipFree(d);
    return ms;
}

int main()
{
    run<0>("v_fma_f32", 2);
    run<1>("and + mad_u32_u24", 4);
    run<2>("shift/xor/add", 4);
    run<3>("v_mul_lo_u32 + add", 4);
    run<4>("rcp+add / mul+rndne", 4);
    run<5>("alignbyte / bfe+add", 4);
    run<6>("cvt+mul / cvt+add", 4);
    run<7>("v_dot4_u32_u8", 2);
    run<8>("cvt_pk_u8_f32 + add", 4);
    run<9>("mul24+add / mad24+shift", 4);
    run<10>("med3+add / ashr+add", 4);
    return 0;
}
