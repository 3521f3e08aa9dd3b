#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float *in, float *out) {
    const int l = threadIdx.x;
    const float x = in[l];
    unsigned u = __builtin_bit_cast(unsigned, x), v = u;
    asm volatile("" : "+v"(v));
    const auto r = __builtin_amdgcn_permlane16_swap(u, v, false, false);
    out[l] = __builtin_bit_cast(float, r[0]);
    out[64 + l] = __builtin_bit_cast(float, r[1]);
    out[128 + l] = __shfl_xor(x, 16);
    unsigned u2 = __builtin_bit_cast(unsigned, x), v2 = u2;
    asm volatile("s_nop 4" : "+v"(v2), "+v"(u2));
    const auto r2 = __builtin_amdgcn_permlane32_swap(u2, v2, false, false);
    out[192 + l] = __builtin_bit_cast(float, r2[0]);
    out[256 + l] = __builtin_bit_cast(float, r2[1]);
}
int main() {
    float h[64], o[320];
    for (int i = 0; i < 64; ++i) h[i] = (float)i;
    float *d, *e;
    (void)hipMalloc(&d, 256); (void)hipMalloc(&e, 1280);
    (void)hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    (void)hipMemcpy(o, e, 1280, hipMemcpyDeviceToHost);
    const char *names[5] = {"p16 r0", "p16 r1", "shfl16", "p32 r0", "p32 r1"};
    for (int a = 0; a < 5; ++a) { printf("%s:", names[a]); for (int i = 0; i < 64; i += 4) printf(" %g", o[a * 64 + i]); printf("\n"); }
    return 0;
}
