#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <bool SUM> __device__ __forceinline__ float j16(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return SUM ? a + b : fmaxf(a, b);
}
template <bool SUM> __device__ __forceinline__ float j32(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return SUM ? a + b : fmaxf(a, b);
}
__global__ void k(const float *in, float *out) {
    const int l = threadIdx.x;
    const float x = in[l];
    out[l] = j32<false>(j16<false>(x));
    const float m16 = fmaxf(x, __shfl_xor(x, 16));
    out[64 + l] = fmaxf(m16, __shfl_xor(m16, 32));
    out[128 + l] = j32<true>(j16<true>(x));
    float s = x + __shfl_xor(x, 16);
    s += __shfl_xor(s, 32);
    out[192 + l] = s;
}
int main() {
    float h[64], o[256];
    for (int i = 0; i < 64; ++i) h[i] = sinf(i * 1.7f) * 3.f;
    float *d, *e;
    (void)hipMalloc(&d, 256); (void)hipMalloc(&e, 1024);
    (void)hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    (void)hipMemcpy(o, e, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += (o[i] != o[64 + i]) + (__builtin_bit_cast(unsigned, o[128 + i]) != __builtin_bit_cast(unsigned, o[192 + i]));
    printf("mismatches %d\n", bad);
    return bad != 0;
}
