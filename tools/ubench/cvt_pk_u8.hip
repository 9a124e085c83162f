// Does v_cvt_pk_u8_f32 round and saturate like saturate_cast<uchar>(float) = clamp(cvRound(x), 0, 255)?
// Exhaustive over a grid of float inputs (all 2^32 bit patterns sampled by stride, plus every x.5 tie and the
// values around 0 and 255).  hipcc --offload-arch=gfx950 -O2 cvt_pk_u8.hip -o cvt_pk_u8 && ./cvt_pk_u8
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ uint32_t ref_u8(float x)
{
    const float r = rintf(x);
    return static_cast<uint32_t>(fminf(fmaxf(r, 0.f), 255.f));     // NaN -> 0
}

__global__ void k(uint32_t stride, unsigned long long* bad, uint32_t* first)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t bits = i * stride + (stride >> 1);
    float x;
    memcpy(&x, &bits, 4);
    uint32_t out = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %2" : "=v"(out) : "v"(x), "v"(0u));
    if ((out & 0xFFu) != ref_u8(x)) {
        if (atomicAdd(bad, 1ull) == 0) *first = bits;
    }
}

__global__ void ties(unsigned long long* bad, uint32_t* first)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // x = (i - 1024) / 4 covers -256..767 in quarter steps
    const float x = (static_cast<float>(i) - 1024.f) * 0.25f;
    for (int d = -2; d <= 2; ++d) {
        float y = x;
        uint32_t b;
        memcpy(&b, &y, 4);
        b += d;                                                 // neighbours in ulps
        memcpy(&y, &b, 4);
        uint32_t out = 0;
        asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %2" : "=v"(out) : "v"(y), "v"(0u));
        if ((out & 0xFFu) != ref_u8(y)) {
            if (atomicAdd(bad, 1ull) == 0) *first = b;
        }
    }
}

int main()
{
    unsigned long long* bad; uint32_t* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4);
    hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
    const uint32_t stride = 16;                                  // 2^28 samples across all bit patterns
    hipLaunchKernelGGL(k, dim3((1u << 28) / 256), dim3(256), 0, 0, stride, bad, first);
    hipLaunchKernelGGL(ties, dim3(16), dim3(256), 0, 0, bad, first);
    unsigned long long hb = 0; uint32_t hf = 0;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    float fx; memcpy(&fx, &hf, 4);
    std::printf("mismatches: %llu (first at bits 0x%08x = %g)\n", hb, hf, fx);
    return hb != 0;
}
