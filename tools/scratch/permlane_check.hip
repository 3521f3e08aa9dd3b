// stand-alone check of the v_permlane16_swap / v_permlane32_swap reductions against __shfl_xor (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <bool SUM> __device__ float j16(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    unsigned v = u;
    asm volatile("" : "+v"(v));
    const auto r = __builtin_amdgcn_permlane16_swap(u, v, false, false);
    const float a = __builtin_bit_cast(float, r[0]), b = __builtin_bit_cast(float, r[1]);
    return SUM ? a + b : fmaxf(a, b);
}
template <bool SUM> __device__ float j32(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    unsigned v = u;
    asm volatile("" : "+v"(v));
    const auto r = __builtin_amdgcn_permlane32_swap(u, v, false, false);
    const float a = __builtin_bit_cast(float, r[0]), b = __builtin_bit_cast(float, r[1]);
    return SUM ? a + b : fmaxf(a, b);
}
__global__ void k(const float *in, float *out, unsigned *raw) {
    const int l = threadIdx.x;
    const float x = in[l];
    out[l] = j32<false>(j16<false>(x));
    out[64 + l] = fmaxf(fmaxf(x, __shfl_xor(x, 16)), __shfl_xor(fmaxf(x, __shfl_xor(x, 16)), 32));
    out[128 + l] = j32<true>(j16<true>(x));
    float s = x + __shfl_xor(x, 16);
    s += __shfl_xor(s, 32);
    out[192 + l] = s;
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)l, (unsigned)(100 + l), false, false);
    raw[l] = r[0];
    raw[64 + l] = r[1];
    const auto r2 = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)(100 + l), false, false);
    raw[128 + l] = r2[0];
    raw[192 + l] = r2[1];
}
int main() {
    float h[64], o[256];
    unsigned raw[256];
    for (int i = 0; i < 64; ++i) h[i] = sinf(i * 1.7f) * 3.f;
    float *d, *e;
    unsigned *r;
    hipMalloc(&d, 256); hipMalloc(&e, 1024); hipMalloc(&r, 1024);
    hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e, r);
    hipMemcpy(o, e, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(raw, r, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += (o[i] != o[64 + i]) + (o[128 + i] != o[192 + i]);
    printf("mismatches %d\n", bad);
    printf("permlane16_swap(l, 100+l): vdst' ="); for (int i = 0; i < 64; i += 8) printf(" %u", raw[i]); printf("\n src' ="); for (int i = 0; i < 64; i += 8) printf(" %u", raw[64 + i]);
    printf("\npermlane32_swap: vdst' ="); for (int i = 0; i < 64; i += 8) printf(" %u", raw[128 + i]); printf("\n src' ="); for (int i = 0; i < 64; i += 8) printf(" %u", raw[192 + i]); printf("\n");
    return bad != 0;
}
