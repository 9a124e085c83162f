// Check on gfx950: float(n) / d by IEEE division (what the compiler emits under
// -fhip-fp32-correctly-rounded-divide-sqrt) against the same Newton / Markstein sequence with the reciprocal
// refinement shared by several numerators and no range scaling -- for the operands FeatherBlender's normalisation
// can produce: n an integer in [0, 4095] (a wrapped sum of shorts never exceeds 16 * 255), d = weight sum + 1e-5f in
// [1e-5, 17).  Every (n, d) pair with d on a stride of STRIDE float bit patterns (offset by the launch index) is compared
// bit for bit.  Build: hipcc -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float div_shared(float n, float d, float r1)
{
    const float q0 = n * r1;
    const float e1 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(e2, r1, q1);
}
__device__ __forceinline__ float refined_rcp(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}

__global__ __launch_bounds__(256) void k_check(uint32_t d_first, uint32_t d_last, uint32_t stride, unsigned long long* bad, uint32_t* example)
{
    const uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t bits = (uint64_t)d_first + idx * stride;
    if (bits > d_last) return;
    const float d = __uint_as_float((uint32_t)bits);
    const float r1 = refined_rcp(d);
    unsigned long long local = 0;
    for (int n = 0; n < 4096; ++n) {
        const float fn = (float)n;
        const float ref = fn / d;
        const float got = div_shared(fn, d, r1);
        if (__float_as_uint(ref) != __float_as_uint(got)) {
            if (!local) { example[0] = (uint32_t)bits; example[1] = n; example[2] = __float_as_uint(ref); example[3] = __float_as_uint(got); }
            ++local;
        }
    }
    if (local) atomicAdd(bad, local);
}

int main()
{
    unsigned long long* d_bad; uint32_t* d_ex;
    hipMalloc(&d_bad, 8); hipMalloc(&d_ex, 16);
    hipMemset(d_bad, 0, 8); hipMemset(d_ex, 0, 16);
    const float lo = 1e-5f, hi = 17.0f;
    uint32_t b0, b1; memcpy(&b0, &lo, 4); memcpy(&b1, &hi, 4);
    const uint32_t stride = 5;
    unsigned long long total = 0;
    for (uint32_t off = 0; off < stride; ++off) {          // all offsets: every float in the range
        const uint64_t count = ((uint64_t)b1 - (b0 + off)) / stride + 1;
        hipLaunchKernelGGL(k_check, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, b0 + off, b1, stride, d_bad, d_ex);
        total += count;
    }
    hipDeviceSynchronize();
    unsigned long long bad; uint32_t ex[4];
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(ex, d_ex, 16, hipMemcpyDeviceToHost);
    printf("divisors checked: %llu (every float in [1e-5, 17]) x 4096 numerators; mismatches: %llu\n", total, bad);
    if (bad) printf("example: d bits %08x n %u ref %08x got %08x\n", ex[0], ex[1], ex[2], ex[3]);
    return bad != 0;
}
