// crh_encoder.hip -- UniXcoder (RoBERTa-base geometry) encoder kernels for gfx950 / CDNA4.
//
// Replaces the torch/transformers forward behind UniXcoder.forward
// (src/lattice/providers/unixcoder_provider.py:137-155; HF modeling_roberta.py embeddings :56-121,
// self-attention :158-183, output+LN :329-340, FFN :372-398).  bf16 activations and weights,
// f32 accumulation / LayerNorm / softmax / pooling.  Oracle: oracle/encoder.py (pinned against HF).
//
// Kernels
//   k_embed_ln    gather word+type+position rows, LayerNorm, also emits the key-validity bitmask (ids != pad)
//   k_gemm_nt     C[M,N] = epi(A[M,K] . W[N,K]^T + bias), small-T kernel: 256x128x64 tiles, 3-stage LDS-DMA ring (the large-T
//                 kernel, 256x256x64 ping-pong, is g256::k_gemm_pp in crh_gemm256.hpp),
//                 epilogues: bias | bias+erf-GELU | bias+residual
//   k_layernorm   in-place row LayerNorm over 768 (one wave per row)
//   k_attn        per (batch row, head): K and V of the whole row staged once in LDS, S^T = K.Q^T and O^T = V^T.P^T on
//                 MFMA with the softmax statistics lane-local (query on the lane), V consumed through
//                 ds_read_b64_tr_b16 so it is never transposed in memory; keys masked by the validity bitmask
//   k_pool        masked mean over valid tokens (f32 out, no L2 normalisation)
#include <algorithm>
#include <cmath>
#include <cstdlib>

#ifndef CRH_ATTN_PERMLANE
#define CRH_ATTN_PERMLANE 1
#endif
#include "crh_common.h"

namespace crh {
namespace enc {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf2f(unsigned int b) { return __builtin_bit_cast(float, b << 16); }
__device__ __forceinline__ unsigned int f2bf(float x)
{
    unsigned int u = __builtin_bit_cast(unsigned int, x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u | 0x00400000u) >> 16;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// two floats -> packed bf16 pair, round-to-nearest-even: ONE v_cvt_pk_bf16_f32 on gfx950 (the integer sequence of f2bf is
// ~8 VALU instructions per element; a GEMM epilogue converts 512 elements per lane and spent ~10 us per tile on it)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned int pack2(float a, float b)
{
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}

// erf-GELU for the GEMM epilogues, two elements at a time on the packed-f32 VALU: x * Phi(x) with
// Phi(x) = 1 / (1 + 2^q(x)), q an odd degree-9 polynomial fitted (minimax, tools/fit_gelu.py) so that |x Phi(x) - gelu(x)| <= 4e-6
// for every x (f32 evaluation; relative error <= 2.2e-3 ~ half a bf16 ulp wherever |gelu| > 1e-3).  No clamp is needed: the
// polynomial q(x)/x is negative for every x (checked numerically, tools/fit_gelu.py), so q runs off to -/+inf with the right
// sign, 2^q saturates to 0 or inf and the result to x or -0, NaN-free up to 1e30.  Per pair: 7 packed mul/fma/add, 2 v_exp,
// 2 v_rcp -- about a third of the
// VALU time of the scalar Abramowitz-Stegun 7.1.26 form it replaced, which matters because the FFN1 epilogue evaluates 128 of these per lane per 256x256
// tile while the matrix pipe waits (measured: 12.7 us of a 20 us tile before).
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x)
{
    const f32x2_t x2 = x * x;
    const f32x2_t c9 = {-3.229079084121622e-06f, -3.229079084121622e-06f}, c7 = {8.823996904538944e-05f, 8.823996904538944e-05f},
                   c5 = {0.0003602632787078619f, 0.0003602632787078619f}, c3 = {-0.10522666573524475f, -0.10522666573524475f},
                   c1 = {-2.3020453453063965f, -2.3020453453063965f}, one = {1.f, 1.f};
    f32x2_t h = __builtin_elementwise_fma(x2, c9, c7);
    h = __builtin_elementwise_fma(h, x2, c5);
    h = __builtin_elementwise_fma(h, x2, c3);
    h = __builtin_elementwise_fma(h, x2, c1);
    const f32x2_t q = h * x;
    f32x2_t e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
    e = e + one;
    const f32x2_t r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
    return x * r;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ------------------------------------------------------------------ packed rows: offsets are device data, so they are clamped
// Row b of a packed batch = tokens [row_off[b], row_off[b+1]) of a T-token axis, at most Lmax of them.  The library cannot
// look at a device array when a call is made, so every kernel clamps what it reads: a bad offsets array gives wrong rows,
// never an access outside the buffers; k_check_row_off reports it (crh_encoder_finish).
__device__ __forceinline__ void packed_row(const int32_t *__restrict__ row_off, int b, int T, int Lmax, int64_t &r0, int &L)
{
    int a = row_off[b], e = row_off[b + 1];
    a = a < 0 ? 0 : (a > T ? T : a);
    e = e < a ? a : (e > T ? T : e);
    r0 = a;
    L = (e - a) < Lmax ? (e - a) : Lmax;
}
// status bits: 1 row_off[0] != 0, 2 decreasing, 4 a row longer than Lmax, 8 row_off[B] != T
__global__ __launch_bounds__(256) void k_check_row_off(const int32_t *__restrict__ row_off, int B, int T, int Lmax, unsigned int *status)
{
    unsigned int bad = 0u;
    for (int i = threadIdx.x; i <= B; i += 256) {
        const int v = row_off[i];
        if (i == 0 && v != 0) bad |= 1u;
        if (i == B && v != T) bad |= 8u;
        if (i < B) {
            const int n = row_off[i + 1] - v;
            if (n < 0) bad |= 2u;
            if (n > Lmax) bad |= 4u;
        }
    }
    if (bad) __hip_atomic_fetch_or(status, bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------ embeddings + LayerNorm

// grid = (B, L/16), block = 256: a workgroup embeds 16 tokens of one row (one row per workgroup left a single query on one
// CU walking its tokens in series: 20 us for 64 tokens); every workgroup of a row redoes the row's cheap mask / position scan,
// the first one writes the mask words.  ids int32 [B,L]; out bf16 [B,L,768]; kmask u64 [B][L/64] (bit = ids != pad).
// position id = cumsum(ids != pad) * (ids != pad) + pad   (modeling_roberta.py create_position_ids_from_input_ids)
__global__ __launch_bounds__(256) void k_embed_ln(const int32_t *__restrict__ ids, const bf16_t *__restrict__ word,
                                                  const bf16_t *__restrict__ pos, const bf16_t *__restrict__ type0,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                                  int pad_id, bf16_t *__restrict__ out, unsigned long long *__restrict__ kmask,
                                                  int Lpad, int D, const int32_t *__restrict__ row_off, int T)
{
    extern __shared__ int posid[];  // [L]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // packed rows (row_off != NULL): row b holds the tokens [row_off[b], row_off[b+1]) of a flat id array, no padding between
    // rows; Lpad (a multiple of 16, >= every row) only sizes the mask stride and the grid.  Padded rows: L = Lpad tokens at b*Lpad.
    // Whatever row_off holds, the row is clamped to the T tokens of the buffers and to Lpad tokens (see packed_row).
    int64_t r0 = (int64_t)b * Lpad;
    int L = Lpad;
    if (row_off) packed_row(row_off, b, T, Lpad, r0, L);
    if ((int)blockIdx.y * 16 >= L && blockIdx.y != 0) return;      // (block y = 0 of a row always writes the row's mask words)
    const int32_t *row = ids + r0;
    const int nw = (Lpad + 63) >> 6;                 // mask words per row (Lpad is a multiple of 16, not necessarily of 64)
    __shared__ int wcnt[16];                       // real tokens per 64-token word (L <= 1024)
    for (int t0 = wave * 64; t0 < nw * 64; t0 += 256) {  // validity bitmask, 64 tokens per wave step (words past the row: 0)
        const bool v = (t0 + lane < L) && row[t0 + lane] != pad_id;
        const unsigned long long m = __ballot(v);
        if (lane == 0) {
            if (blockIdx.y == 0) kmask[(size_t)b * nw + (t0 >> 6)] = m;
            wcnt[t0 >> 6] = __popcll(m);
        }
    }
    // position of token t = number of real tokens up to and including t: whole 64-token words before it (their popcounts
    // meet in LDS) + the bits of its own word up to its lane.  (Thread 0 walking the row was L dependent loads: 42 us at L = 128.)
    __syncthreads();
    for (int t0 = wave * 64; t0 < L; t0 += 256) {
        const bool v = (t0 + lane < L) && row[t0 + lane] != pad_id;
        const unsigned long long m = __ballot(v);
        int run = __popcll(m & ((2ull << lane) - 1ull));
        for (int w = 0; w < (t0 >> 6); ++w) run += wcnt[w];
        if (t0 + lane < L) posid[t0 + lane] = (v ? run : 0) + pad_id;
    }
    __syncthreads();
    constexpr int per = 3;  // D == 768: 3 groups of 4 elements per lane
    for (int t = blockIdx.y * 16 + wave; t < blockIdx.y * 16 + 16 && t < L; t += 4) {
        const bf16_t *w = word + (size_t)row[t] * D;
        const bf16_t *p = pos + (size_t)posid[t] * D;
        float x[12];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < per; ++i) {
            const int e = i * 256 + lane * 4;
            const u32x2 wv = *reinterpret_cast<const u32x2 *>(w + e);
            const u32x2 pv = *reinterpret_cast<const u32x2 *>(p + e);
            const u32x2 tv = *reinterpret_cast<const u32x2 *>(type0 + e);
            x[4 * i + 0] = (bf2f(wv.x & 0xffffu) + bf2f(tv.x & 0xffffu)) + bf2f(pv.x & 0xffffu);
            x[4 * i + 1] = (bf2f(wv.x >> 16) + bf2f(tv.x >> 16)) + bf2f(pv.x >> 16);
            x[4 * i + 2] = (bf2f(wv.y & 0xffffu) + bf2f(tv.y & 0xffffu)) + bf2f(pv.y & 0xffffu);
            x[4 * i + 3] = (bf2f(wv.y >> 16) + bf2f(tv.y >> 16)) + bf2f(pv.y >> 16);
            s += x[4 * i] + x[4 * i + 1] + x[4 * i + 2] + x[4 * i + 3];
        }
        const float mu = wave_sum(s) / D;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 4 * per; ++i) v += (x[i] - mu) * (x[i] - mu);
        const float rstd = rsqrtf(wave_sum(v) / D + eps);
        bf16_t *o = out + (size_t)(r0 + t) * D;
#pragma unroll
        for (int i = 0; i < per; ++i) {
            const int e = i * 256 + lane * 4;
            const float4 g = *reinterpret_cast<const float4 *>(gamma + e);
            const float4 bb = *reinterpret_cast<const float4 *>(beta + e);
            u32x2 r;
            r.x = pack2((x[4 * i] - mu) * rstd * g.x + bb.x, (x[4 * i + 1] - mu) * rstd * g.y + bb.y);
            r.y = pack2((x[4 * i + 2] - mu) * rstd * g.z + bb.z, (x[4 * i + 3] - mu) * rstd * g.w + bb.w);
            *reinterpret_cast<u32x2 *>(o + e) = r;
        }
    }
}

// ------------------------------------------------------------------ row LayerNorm (in place), D = 768
// (One wave per row, 8-byte accesses.  A form with two rows per wave and 16-byte accesses -- three dwordx4 per lane -- was measured
// in round 3: O-projection + LayerNorm 134 us against 126, FFN2 + LayerNorm 308 against 305 at 65 k tokens; half as many waves in
// flight cost more than the wider accesses returned.  Not shipped.)

__global__ __launch_bounds__(256) void k_layernorm768(bf16_t *__restrict__ x, const float *__restrict__ gamma,
                                                      const float *__restrict__ beta, float eps, int rows)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    bf16_t *p = x + (size_t)r * 768;
    float v[12];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const u32x2 w = *reinterpret_cast<const u32x2 *>(p + i * 256 + lane * 4);
        v[4 * i] = bf2f(w.x & 0xffffu);
        v[4 * i + 1] = bf2f(w.x >> 16);
        v[4 * i + 2] = bf2f(w.y & 0xffffu);
        v[4 * i + 3] = bf2f(w.y >> 16);
        s += v[4 * i] + v[4 * i + 1] + v[4 * i + 2] + v[4 * i + 3];
    }
    const float mu = wave_sum(s) * (1.f / 768.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) q += (v[i] - mu) * (v[i] - mu);
    const float rstd = rsqrtf(wave_sum(q) * (1.f / 768.f) + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = i * 256 + lane * 4;
        const float4 g = *reinterpret_cast<const float4 *>(gamma + e);
        const float4 b = *reinterpret_cast<const float4 *>(beta + e);
        u32x2 o;
        o.x = pack2((v[4 * i] - mu) * rstd * g.x + b.x, (v[4 * i + 1] - mu) * rstd * g.y + b.y);
        o.y = pack2((v[4 * i + 2] - mu) * rstd * g.z + b.z, (v[4 * i + 3] - mu) * rstd * g.w + b.w);
        *reinterpret_cast<u32x2 *>(p + e) = o;
    }
}

// LayerNorm(x + residual): the pre-LN sum is formed here in f32 (x = the GEMM output WITHOUT the residual).  Used where the
// GEMM's residual epilogue costs more than this kernel's extra read: the O-projection (K = 768), whose epilogue would pull
// 128 KB of cold residual per CU and tile with the matrix pipe idle (+31 us per call at 65 k tokens against +18 us here).
__global__ __launch_bounds__(256) void k_layernorm768_res(bf16_t *__restrict__ x, const bf16_t *__restrict__ res, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float eps, int rows)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    bf16_t *p = x + (size_t)r * 768;
    const bf16_t *pr = res + (size_t)r * 768;
    float v[12];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const u32x2 w = *reinterpret_cast<const u32x2 *>(p + i * 256 + lane * 4);
        const u32x2 z = *reinterpret_cast<const u32x2 *>(pr + i * 256 + lane * 4);
        v[4 * i] = bf2f(w.x & 0xffffu) + bf2f(z.x & 0xffffu);
        v[4 * i + 1] = bf2f(w.x >> 16) + bf2f(z.x >> 16);
        v[4 * i + 2] = bf2f(w.y & 0xffffu) + bf2f(z.y & 0xffffu);
        v[4 * i + 3] = bf2f(w.y >> 16) + bf2f(z.y >> 16);
        s += v[4 * i] + v[4 * i + 1] + v[4 * i + 2] + v[4 * i + 3];
    }
    const float mu = wave_sum(s) * (1.f / 768.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) q += (v[i] - mu) * (v[i] - mu);
    const float rstd = rsqrtf(wave_sum(q) * (1.f / 768.f) + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = i * 256 + lane * 4;
        const float4 g = *reinterpret_cast<const float4 *>(gamma + e);
        const float4 b = *reinterpret_cast<const float4 *>(beta + e);
        u32x2 o;
        o.x = pack2((v[4 * i] - mu) * rstd * g.x + b.x, (v[4 * i + 1] - mu) * rstd * g.y + b.y);
        o.y = pack2((v[4 * i + 2] - mu) * rstd * g.z + b.z, (v[4 * i + 3] - mu) * rstd * g.w + b.w);
        *reinterpret_cast<u32x2 *>(p + e) = o;
    }
}

// LayerNorm(x + residual) with the residual stream kept in F32 (the opt-in fidelity lever of round 5: the GEMMs still read bf16
// rows, but what a layer adds its output to -- and what twelve layers therefore accumulate their roundings in -- is not rounded to
// bf16 between layers).  x: the GEMM output (bf16, bias included), replaced by bf16(LayerNorm) for the next GEMM; res32: the f32
// residual, replaced by the f32 LayerNorm output, the next residual.  One wave per row; 12 bytes per element instead of 6.
__global__ __launch_bounds__(256) void k_layernorm768_res32(bf16_t *__restrict__ x, float *__restrict__ res32, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float eps, int rows)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    bf16_t *p = x + (size_t)r * 768;
    float *pr = res32 + (size_t)r * 768;
    float v[12];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const u32x2 w = *reinterpret_cast<const u32x2 *>(p + i * 256 + lane * 4);
        const float4 z = *reinterpret_cast<const float4 *>(pr + i * 256 + lane * 4);
        v[4 * i] = bf2f(w.x & 0xffffu) + z.x;
        v[4 * i + 1] = bf2f(w.x >> 16) + z.y;
        v[4 * i + 2] = bf2f(w.y & 0xffffu) + z.z;
        v[4 * i + 3] = bf2f(w.y >> 16) + z.w;
        s += v[4 * i] + v[4 * i + 1] + v[4 * i + 2] + v[4 * i + 3];
    }
    const float mu = wave_sum(s) * (1.f / 768.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) q += (v[i] - mu) * (v[i] - mu);
    const float rstd = rsqrtf(wave_sum(q) * (1.f / 768.f) + eps);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = i * 256 + lane * 4;
        const float4 g = *reinterpret_cast<const float4 *>(gamma + e);
        const float4 b = *reinterpret_cast<const float4 *>(beta + e);
        const float4 y = float4{(v[4 * i] - mu) * rstd * g.x + b.x, (v[4 * i + 1] - mu) * rstd * g.y + b.y, (v[4 * i + 2] - mu) * rstd * g.z + b.z,
                                (v[4 * i + 3] - mu) * rstd * g.w + b.w};
        u32x2 o;
        o.x = pack2(y.x, y.y);
        o.y = pack2(y.z, y.w);
        *reinterpret_cast<u32x2 *>(p + e) = o;
        *reinterpret_cast<float4 *>(pr + e) = y;
    }
}

// ------------------------------------------------------------------ LayerNorm folded into the GEMMs around it (round 5)
//
// The post-LN layer keeps its residual stream UN-normalised between kernels.  A producer GEMM (O-projection, FFN2; epilogue 5 of
// the three tiled kernels) writes r = bf16(acc + bias + h), h being the previous LayerNorm's output worked out on the fly from ITS
// un-normalised rows (h = fma(fma(r', rstd, nmr), gamma, beta'), beta' = beta + bias folded on the host), and per row and 32-column
// slot the slot's mean and sum of squared deviations of the ROUNDED outputs; k_ln_finalize combines a row's 24 pairs into
// (rstd, nmr = -mu rstd); the consumer GEMM (QKV, FFN1; epilogues 3 / 4) multiplies r by weights that carry the LayerNorm gain
// and finishes the normalisation per element: y = fma(acc, rstd, fma(nmr, c_n, b'_n)).  Nothing reads or writes a [T, 768]
// tensor just to normalise it (two such passes per layer before: 98 + 49 MB read, 2 x 98 MB written at 65 k tokens).
//
// The statistics are built the same way in all three GEMM kernels, so a token's bits do not depend on which kernel its batch
// size selects: a lane holds, per row and slot, 8 outputs (two C^T fragments of 4 consecutive columns); it sums their deviations
// from the first of them (one pass, nothing kept), and the four lanes that share the row are joined pairwise (xor 16, then xor
// 32) by the equal-count update mean = (a + b) / 2, M2 = M2a + M2b + (a - b)^2 n / 2 -- symmetric in its operands, so both lanes
// of a pair hold the same bits.
struct LnAcc {
    float s, sd, sdd;   // shift (the lane's first element), sum of (x - s), sum of (x - s)^2
};
__device__ __forceinline__ void ln_acc4(LnAcc &a, bool first, float x0, float x1, float x2, float x3)
{
    // (explicit fma chains: the file is compiled with -ffp-contract=off, and one instruction per element is what this costs)
    if (first) {
        a.s = x0;
        const float d1 = x1 - x0, d2 = x2 - x0, d3 = x3 - x0;
        a.sd = (d1 + d2) + d3;
        a.sdd = __builtin_fmaf(d3, d3, __builtin_fmaf(d2, d2, d1 * d1));
    } else {
        const float d0 = x0 - a.s, d1 = x1 - a.s, d2 = x2 - a.s, d3 = x3 - a.s;
        a.sd = a.sd + ((d0 + d1) + (d2 + d3));
        a.sdd = __builtin_fmaf(d3, d3, __builtin_fmaf(d2, d2, __builtin_fmaf(d1, d1, __builtin_fmaf(d0, d0, a.sdd))));
    }
}
__device__ __forceinline__ float2 ln_acc_done8(const LnAcc &a)   // 8 elements -> (mean, M2)
{
    const float md = a.sd * 0.125f;
    return float2{a.s + md, fmaxf(a.sdd - a.sd * md, 0.f)};
}
__device__ __forceinline__ float2 ln_join(float2 a, float2 b, float half_n)   // two partials of n elements each; half_n = n / 2
{
    const float d = a.x - b.x;
    return float2{0.5f * (a.x + b.x), (a.y + b.y) + (d * d) * half_n};
}
// own value and the value of lane l ^ 16 (l ^ 32), in an order that depends on the lane only: (a, b) -> the same pair in both lanes
// of a couple (see join16 / join32 further down for the instruction)
__device__ __forceinline__ void pair16(float x, float &a, float &b);
__device__ __forceinline__ void pair32(float x, float &a, float &b);
// (mean, M2) of 8 elements in each of the four lanes that share a row -> of the row's 32-column slot, in all four
__device__ __forceinline__ float2 ln_join_row(float2 p)
{
    float am, bm, a2, b2;
    pair16(p.x, am, bm);
    pair16(p.y, a2, b2);
    p = ln_join(float2{am, a2}, float2{bm, b2}, 4.0f);
    pair32(p.x, am, bm);
    pair32(p.y, a2, b2);
    return ln_join(float2{am, a2}, float2{bm, b2}, 8.0f);
}

// A row's 24 (mean, M2) pairs of 32 columns each -> (rstd, -mu * rstd), slots combined in index order (equal counts: mu = mean of
// the means, M2 = sum of the M2s + 32 * sum (mean_p - mu)^2).  One thread per row, its 192 bytes fetched as 12 independent
// 16-byte loads; 64 rows per workgroup so that 65 k rows fill the chip (one thread per row in 256-row workgroups, loads behind
// a loop of unknown length: 10.9 us per call).
constexpr int kLnSlots = 24;
__global__ __launch_bounds__(64) void k_ln_finalize(const float *__restrict__ partials, float *__restrict__ stats, int rows, float eps)
{
    const int m = blockIdx.x * 64 + threadIdx.x;
    if (m >= rows) return;
    const float4 *p4 = reinterpret_cast<const float4 *>(partials) + (size_t)m * (kLnSlots / 2);
    float4 v[kLnSlots / 2];
#pragma unroll
    for (int i = 0; i < kLnSlots / 2; ++i) v[i] = p4[i];
    float msum = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < kLnSlots / 2; ++i) {
        msum += v[i].x;
        msum += v[i].z;
    }
    const float mu = msum / (float)kLnSlots;
    float dev = 0.f;
#pragma unroll
    for (int i = 0; i < kLnSlots / 2; ++i) {
        const float d0 = v[i].x - mu, d1 = v[i].z - mu;
        m2 += v[i].y;
        dev += d0 * d0;
        m2 += v[i].w;
        dev += d1 * d1;
    }
    const float var = (m2 + 32.0f * dev) / (float)(32 * kLnSlots);
    const float rstd = rsqrtf(var + eps);
    *reinterpret_cast<float2 *>(stats + 2 * (size_t)m) = float2{rstd, -mu * rstd};
}

// y = bf16(fma(fma(x, rstd, nmr), gamma, beta)): the LayerNorm output itself, from statistics already known -- what the LAST layer
// of a folded forward needs (the pool reads normalised rows); in place when y == x.  One wave per row, 8-byte accesses.
__global__ __launch_bounds__(256) void k_ln_apply768(const bf16_t *__restrict__ x, const float *__restrict__ stats, const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, bf16_t *__restrict__ y, int rows)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float2 st = *reinterpret_cast<const float2 *>(stats + 2 * (size_t)r);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int e = i * 256 + lane * 4;
        const u32x2 w = *reinterpret_cast<const u32x2 *>(x + (size_t)r * 768 + e);
        const float4 g = *reinterpret_cast<const float4 *>(gamma + e);
        const float4 b = *reinterpret_cast<const float4 *>(beta + e);
        u32x2 o;
        o.x = pack2(__builtin_fmaf(__builtin_fmaf(bf2f(w.x & 0xffffu), st.x, st.y), g.x, b.x), __builtin_fmaf(__builtin_fmaf(bf2f(w.x >> 16), st.x, st.y), g.y, b.y));
        o.y = pack2(__builtin_fmaf(__builtin_fmaf(bf2f(w.y & 0xffffu), st.x, st.y), g.z, b.z), __builtin_fmaf(__builtin_fmaf(bf2f(w.y >> 16), st.x, st.y), g.w, b.w));
        *reinterpret_cast<u32x2 *>(y + (size_t)r * 768 + e) = o;
    }
}

// ------------------------------------------------------------------ GEMM  C = epi(A . W^T + bias)

constexpr int BM = 256, BN = 128, BK = 64, STAGES = 3, GEMM_WAVES = 8;

// byte offset of 16-byte chunk c (0..7) of row r inside a [rows][64 bf16] LDS tile (128-B rows), XOR-swizzled so that a
// wave's ds_read_b128 of 16 rows x 4 chunks spreads over all 16 slots of the 256-B bank row (conflict-free)
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

// EPI: 0 bias, 1 bias + erf-GELU, 2 bias + residual.  M arbitrary (guarded), N % 128 == 0, K % 64 == 0.
// 256x128x64 tiles, 8 waves (4 along M x 2 along N, 64x64 each), one workgroup per CU.
//  * Staging by LDS-DMA (global_load_lds_dwordx4): one wave-instruction moves 1 KiB = 8 tile rows straight into LDS
//    (destination = wave-uniform base + lane*16, a lane-linear image); the XOR swizzle the fragment reads expect is applied
//    to the per-lane SOURCE chunk.  No staging VGPRs, no ds_write.
//  * 3-stage ring, ONE raw s_barrier per k-iteration, counted vmcnt: the loads of tile kt+2 are issued right after the
//    barrier of iteration kt and stay in flight across the next barrier, so a tile has two full iterations (~1 us) to land.
//    (With 2 stages and __syncthreads' vmcnt(0) every iteration exposed the whole L2/HBM latency: 600 TFLOP/s.)
//  * The MFMA is issued as (W-fragment, A-fragment): the accumulator holds C^T tiles -- row = n (registers), col = m (lane)
//    -- so each lane owns 4 CONSECUTIVE n of one row m and the epilogue stores 8 bytes at a time.
// DBG bit mask (timing ablations only, wrong results): 1 = no LDS-DMA inside the loop, 2 = no MFMA, 4 = no fragment reads
// in the loop, 8 = no DMA of the W tile
//
// Persistent: the grid is 8k workgroups (<= one per CU); each walks a strided list of tiles with ONE software pipeline
// running across tile boundaries -- while a tile's epilogue runs, the first two k-tiles of the next tile are already in
// flight, so neither the DMA latency of a tile's first loads nor (most of) its epilogue is exposed any more
// (measured before: ~6 us of fixed cost per 256x128 tile against ~11 us of main loop at K = 768).
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(GEMM_WAVES * 64) void k_gemm_nt(const bf16_t *__restrict__ A, const bf16_t *__restrict__ W,
                                                            const float *__restrict__ bias, const bf16_t *__restrict__ R,
                                                            bf16_t *__restrict__ C, int M, int N, int K,
                                                            const float *__restrict__ aux0 = nullptr, const float *__restrict__ rstats = nullptr,
                                                            float *__restrict__ partials = nullptr)
{
    constexpr int kStage = (BM + BN) * BK * 2;  // bytes per stage: A tile (256 rows) then W tile (128 rows)
    constexpr int kPieces = (BM + BN) / 8;      // 1-KiB pieces per stage (8 rows each): 32 of A, 16 of W
    constexpr int kPer = kPieces / GEMM_WAVES;  // LDS-DMA instructions per wave per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // STAGES * kStage (the only LDS object)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, c16 = lane & 15;

    // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MB L2), so with the
    // plain order the nb column-tiles that share one 256-row A panel land on 8 different L2s and the panel is fetched from
    // HBM/MALL up to nb times (measured: the loop was bound by exactly that re-read traffic, ~6 TB/s).  Each XCD label
    // therefore owns a contiguous range of A panels and its workgroups stride through that range together, in SUPER-TILES
    // of 4 panels x 8 column tiles = 32 tiles = what the XCD's 32 CUs hold at once: ~1.5 MB of A + ~1.5 MB of W resident
    // in the 4 MB L2, each A panel shared by 8 CUs and each W tile by 4 (with N fastest over all of N, a W of 4.7 MB --
    // FFN1 -- cycled through the L2 once per panel).
    // (`b % 8` only labels workgroups that share an XCD; a different placement would change speed, never results.)
    constexpr int HM = 4, WN = 8;
    const int nb = N / BN, nk = K / BK;
    const int panels = (M + BM - 1) / BM;
    const int bpx = gridDim.x >> 3;             // workgroups per XCD label
    const int jx = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const bool by_panel = panels >= 16;         // enough panels to give every XCD label its own range
    int first, cnt;                             // this label's unit range: panels (by_panel) or tiles
    {
        const int units = by_panel ? panels : panels * nb;
        const int q = units >> 3, r = units & 7;
        first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        cnt = q + (xcd < r ? 1 : 0);
    }
    const int my_tiles = by_panel ? cnt * nb : cnt;
    const int ntl = jx < my_tiles ? (my_tiles - jx + bpx - 1) / bpx : 0;
    if (ntl == 0) return;                       // (before any barrier)
    const int total = ntl * nk;
    // ts-th tile of this workgroup -> origin of the tile
    auto tile_origin = [&](int ts, int &tm0, int &tn0) {
        const int u = jx + ts * bpx;            // index inside the label's range
        if (!by_panel) {
            const int tile = first + u;
            tm0 = (tile / nb) * BM;
            tn0 = (tile % nb) * BN;
            return;
        }
        const int gsz = HM * nb;                // tiles per full group of HM panels
        const int sr = u / gsz, ug = u - sr * gsz;
        const int hm = (cnt - sr * HM) < HM ? (cnt - sr * HM) : HM;
        const int bsz = hm * WN;
        const int sc = ug / bsz, ub = ug - sc * bsz;
        const int wn_ = (nb - sc * WN) < WN ? (nb - sc * WN) : WN;
        const int pm = ub / wn_, pn = ub - pm * wn_;
        tm0 = (first + sr * HM + pm) * BM;
        tn0 = (sc * WN + pn) * BN;
    };

    const int prow = lane >> 3, pslot = lane & 7;  // piece-local row / 16-byte slot of this lane
    // Issue cursor: the pipeline step whose LDS-DMA pieces are being dealt out (two steps ahead of the one computed).  Its
    // tile origin, k offset and stage are advanced incrementally -- an integer division per piece cost ~0.7 us per step.
    int c_m0, c_n0, c_k0 = 0, c_buf = 0, c_ts = 0;
    auto cursor_tile = [&]() { tile_origin(c_ts, c_m0, c_n0); };
    cursor_tile();
    auto cursor_advance = [&]() {
        c_buf = (c_buf == STAGES - 1) ? 0 : c_buf + 1;
        c_k0 += BK;
        if (c_k0 == K) {
            c_k0 = 0;
            ++c_ts;
            if (c_ts < ntl) cursor_tile();
        }
    };
    auto stage_one = [&](int i) {
        const int piece = wave * kPer + i;         // pieces 0..31 -> A rows, 32..47 -> W rows
        if ((DBG & 8) && piece >= BM / 8) return;
        const int r = piece * 8 + prow;            // row inside the stacked [A;W] stage image
        const int c = pslot ^ (r & 7);             // slot s of row r holds chunk s ^ (r & 7)
        const bf16_t *src;
        if (piece < BM / 8) {
            int m = c_m0 + r;
            m = m < M ? m : M - 1;                 // rows past M are loaded from a valid row and never stored
            src = A + (size_t)m * K + c_k0 + c * 8;
        } else {
            src = W + (size_t)(c_n0 + r - BM) * K + c_k0 + c * 8;
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(smem + c_buf * kStage + piece * 1024), 16, 0, 0);
    };

    // ---- pipeline.  Per step (one 64-deep k-tile, two MFMA k-steps ks = 0, 1):
    //   first half : MFMAs(it, 0) on fragments Fa (already in registers); the ks=1 fragments Fb are read underneath
    //   middle     : lgkmcnt(0) (stage `it` fully read), counted vmcnt (stage it+1 has landed), ONE barrier
    //   second half: MFMAs(it, 1) on Fb; underneath: the ks=0 fragments of step it+1 -> Fa, and the 6 LDS-DMA pieces of
    //                step it+3 into the stage step `it` just vacated
    // so every LDS read and every DMA issue sits under MFMAs of the same wave, and a stage has two steps to land.
    // (With the barrier at the top of the step, all 16 fragment reads of all 8 waves -- 128 KiB -- hit the LDS at once
    // behind it while the matrix pipes idled: 0.95 us per step with the DMA removed, against 0.45 of MFMA work.)
    f32x4 acc[4][4];  // [nt][mt]
    bf16x8 Fa[4], Wa[4], Fb[4], Wb[4];
    auto read_frags = [&](int buf_, int ks, bf16x8 (&fa_)[4], bf16x8 (&fw_)[4]) {
        const unsigned char *sa = smem + buf_ * kStage;
        const unsigned char *sw = sa + BM * BK * 2;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa_[t] = *reinterpret_cast<const bf16x8 *>(sa + lds_off(wm * 64 + t * 16 + c16, g + 4 * ks));
            fw_[t] = *reinterpret_cast<const bf16x8 *>(sw + lds_off(wn * 64 + t * 16 + c16, g + 4 * ks));
        }
    };
    auto mfma_group = [&](int nt, const bf16x8 (&fa_)[4], const bf16x8 (&fw_)[4]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (DBG & 2) {
                asm volatile("" ::"v"(fw_[nt]), "v"(fa_[mt]));
            } else {
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw_[nt], fa_[mt], acc[nt][mt], 0, 0, 0);
            }
        }
    };

    int issued = 0;
    for (; issued < STAGES && issued < total; ++issued) {
#pragma unroll
        for (int i = 0; i < kPer; ++i) stage_one(i);
        cursor_advance();
    }
    // step 0 must have landed; steps 1 and 2 (if any) may still be in flight
    if (total >= 3)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kPer) : "memory");
    else if (total == 2)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPer) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, Fa, Wa);

    int ts = 0, kt = -1, buf = STAGES - 1;   // compute cursor: tile-list index, k-tile and stage of step `it`
    int stores_before = 0;                    // stores the previous step's epilogue left in flight (8 on a full tile)
    for (int it = 0; it < total; ++it) {
        buf = (buf == STAGES - 1) ? 0 : buf + 1;
        if (++kt == nk) {
            kt = 0;
            ++ts;
        }
        if (kt == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const bool have_next = it + 1 < total;
        const int nbuf = (buf == STAGES - 1) ? 0 : buf + 1;
        const bool tile_end = kt == nk - 1;
        const bool dma = issued < total;       // step it+3 exists

        // ---- first half
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            mfma_group(nt, Fa, Wa);
            if (nt == 0 && !(DBG & 4)) read_frags(buf, 1, Fb, Wb);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- middle: stage `it` is fully read; stage it+1 has landed once only what was issued after it remains outstanding
        // (vmcnt retires in issue order): the pieces of step it+2, preceded -- right after a tile's epilogue -- by that
        // epilogue's stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (have_next) {
            const bool later = it + 2 < total;
            if (later && stores_before == 8)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPer + 8) : "memory");
            else if (later)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPer) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        stores_before = 0;
        __builtin_amdgcn_s_barrier();
        // ---- second half
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            mfma_group(nt, Fb, Wb);
            if (nt == 0 && have_next && !(DBG & 4)) read_frags(nbuf, 0, Fa, Wa);
            if (dma && !tile_end && !(DBG & 1)) {   // 6 pieces over the 4 groups: 2, 2, 1, 1
                stage_one(nt < 2 ? 2 * nt : nt + 2);
                if (nt < 2) stage_one(2 * nt + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (dma && !tile_end) {
            cursor_advance();
            ++issued;
        }
        if (!tile_end) continue;

        // ---- epilogue of this tile.  A lane holds, per (nt, mt), the 4 consecutive n = n0 + wn*64 + nt*16 + 4g + {0..3} of
        // row m = m0 + wm*64 + mt*16 + c16.  Everything it READS (bias, residual) is fetched in one batch up front (a load
        // inside a guarded per-element branch is waited for individually: 16 dependent L2 round trips per lane); the finished
        // bf16 tile goes through a wave-private, XOR-swizzled LDS image so that it leaves as full 128-byte rows (8 x dwordx4
        // per lane) instead of 16 scattered 8-byte pieces.  The image sits in the stage this step vacated (all reads of it
        // completed before the middle barrier), inside the 6 KiB that THIS wave's own DMA pieces of step it+3 will overwrite
        // -- which is why those pieces are issued after the epilogue on a tile's last step, not under its MFMAs.
        int m0, n0;
        tile_origin(ts, m0, n0);
        unsigned char *cimg = smem + buf * kStage + wave * (kPer * 1024);  // [32 rows][64 cols] bf16, both halves in turn
        float4 bv[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bv[nt] = *reinterpret_cast<const float4 *>(bias + n0 + wn * 64 + nt * 16 + 4 * g);
        u32x2 rv[4][4];
        if (EPI == 2 || EPI == 5) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    int m = m0 + wm * 64 + mt * 16 + c16;
                    m = m < M ? m : M - 1;
                    rv[nt][mt] = *reinterpret_cast<const u32x2 *>(R + (size_t)m * N + n0 + wn * 64 + nt * 16 + 4 * g);
                }
        }
        // EPI 3 / 4 ("LN in", crh_gemm256.hpp): per-row (rstd, -mu rstd) of the A rows and the column sums of the gain-scaled weights
        float2 st[4];
        float4 cv[4];
        const bool res_ln = EPI == 5 && rstats != nullptr;   // epilogue 5: the residual is normalised on the fly (cv = the gain)
        if (EPI == 3 || EPI == 4 || res_ln) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                int m = m0 + wm * 64 + mt * 16 + c16;
                m = m < M ? m : M - 1;
                st[mt] = *reinterpret_cast<const float2 *>(rstats + 2 * (size_t)m);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cv[nt] = *reinterpret_cast<const float4 *>(aux0 + n0 + wn * 64 + nt * 16 + 4 * g);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                const int mt = half * 2 + mh;
                LnAcc la[2];   // EPI 5: the row's two 32-column slots of this wave
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    float v0 = acc[nt][mt][0] + bv[nt].x, v1 = acc[nt][mt][1] + bv[nt].y, v2 = acc[nt][mt][2] + bv[nt].z,
                          v3 = acc[nt][mt][3] + bv[nt].w;
                    if (EPI == 3 || EPI == 4) {
                        v0 = __builtin_fmaf(acc[nt][mt][0], st[mt].x, __builtin_fmaf(st[mt].y, cv[nt].x, bv[nt].x));
                        v1 = __builtin_fmaf(acc[nt][mt][1], st[mt].x, __builtin_fmaf(st[mt].y, cv[nt].y, bv[nt].y));
                        v2 = __builtin_fmaf(acc[nt][mt][2], st[mt].x, __builtin_fmaf(st[mt].y, cv[nt].z, bv[nt].z));
                        v3 = __builtin_fmaf(acc[nt][mt][3], st[mt].x, __builtin_fmaf(st[mt].y, cv[nt].w, bv[nt].w));
                    }
                    if (EPI == 1 || EPI == 4) {
                        const f32x2_t ga = gelu_erf2(f32x2_t{v0, v1}), gb = gelu_erf2(f32x2_t{v2, v3});
                        v0 = ga.x;
                        v1 = ga.y;
                        v2 = gb.x;
                        v3 = gb.y;
                    }
                    if (EPI == 2 || EPI == 5) {
                        const float h0 = bf2f(rv[nt][mt].x & 0xffffu), h1 = bf2f(rv[nt][mt].x >> 16), h2 = bf2f(rv[nt][mt].y & 0xffffu),
                                    h3 = bf2f(rv[nt][mt].y >> 16);
                        if (res_ln) {
                            v0 = __builtin_fmaf(__builtin_fmaf(h0, st[mt].x, st[mt].y), cv[nt].x, v0);
                            v1 = __builtin_fmaf(__builtin_fmaf(h1, st[mt].x, st[mt].y), cv[nt].y, v1);
                            v2 = __builtin_fmaf(__builtin_fmaf(h2, st[mt].x, st[mt].y), cv[nt].z, v2);
                            v3 = __builtin_fmaf(__builtin_fmaf(h3, st[mt].x, st[mt].y), cv[nt].w, v3);
                        } else {
                            v0 += h0;
                            v1 += h1;
                            v2 += h2;
                            v3 += h3;
                        }
                    }
                    u32x2 o;
                    o.x = pack2(v0, v1);
                    o.y = pack2(v2, v3);
                    if (EPI == 5) ln_acc4(la[nt >> 1], (nt & 1) == 0, bf2f(o.x & 0xffffu), bf2f(o.x >> 16), bf2f(o.y & 0xffffu), bf2f(o.y >> 16));
                    const int row = mh * 16 + c16;           // row of the 32-row image
                    const int chunk = nt * 2 + (g >> 1);     // 16-byte chunk of the 128-byte row holding cols nt*16 + 4g ..
                    *reinterpret_cast<u32x2 *>(cimg + row * 128 + ((chunk ^ (row & 7)) << 4) + (g & 1) * 8) = o;
                }
                if (EPI == 5) {
                    const float2 p0 = ln_join_row(ln_acc_done8(la[0])), p1 = ln_join_row(ln_acc_done8(la[1]));
                    const int m = m0 + wm * 64 + mt * 16 + c16;
                    if (g == 0 && m < M)
                        *reinterpret_cast<float4 *>(partials + ((size_t)m * (N >> 5) + ((n0 >> 5) + wn * 2)) * 2) = float4{p0.x, p0.y, p1.x, p1.y};
                }
            }
            // (same wave wrote and reads: the compiler's lgkmcnt wait orders the LDS accesses; no barrier needed)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + (lane >> 3), chunk = lane & 7;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(cimg + row * 128 + ((chunk ^ (row & 7)) << 4));
                const int m = m0 + wm * 64 + half * 32 + row;
                if (m < M) *reinterpret_cast<u32x4 *>(C + (size_t)m * N + n0 + wn * 64 + chunk * 8) = v;
            }
        }
        stores_before = (m0 + BM <= M) ? 8 : 0;   // a ragged tile may have skipped stores: assume none are in flight
        if (dma) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image has been read back before its stage is refilled
            if (!(DBG & 1)) {
#pragma unroll
                for (int i = 0; i < kPer; ++i) stage_one(i);
            }
            cursor_advance();
            ++issued;
        }
    }
}

// ------------------------------------------------------------------ attention

// The value of lane (l ^ 16) / (l ^ 32), by ONE v_permlane16_swap / v_permlane32_swap (gfx950) instead of __shfl_xor's
// ds_bpermute_b32: no address arithmetic, no trip through the LDS crossbar and no lgkmcnt wait in a loop whose issue slots are
// what bounds it (the softmax's two row reductions cross the four 16-lane groups of a wave twice per key tile).
// permlane16_swap(a, b) exchanges a's odd 16-lane rows with b's even ones; with a = b = x the pair (a', b') holds, in every lane,
// x of its own row pair's even row and odd row: its own value and its xor-16 partner's, in some order -- which is all a sum or a
// maximum needs (a + b and b + a are the same bits).  Likewise for the two halves of the wave.
#if CRH_ATTN_PERMLANE
// (Written as inline asm: with ROCm 7.2's clang the builtin __builtin_amdgcn_permlane16_swap(u, u, ...) hands back ITS FIRST result for
// both elements when the two operands carry the same value -- found with tools/scratch/permlane_check*.hip; the asm form was checked
// against __shfl_xor there, bit for bit.  The s_nop covers the VALU-write -> permlane-read hazard the compiler cannot see inside asm.)
template <bool SUM>
__device__ __forceinline__ float join16(float x)     // x (+ or max) the value of lane l ^ 16: own and partner end up in a and b, in some order
{
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));     // a = {R0, R0, R2, R2}, b = {R1, R1, R3, R3}
    return SUM ? a + b : fmaxf(a, b);
}
template <bool SUM>
__device__ __forceinline__ float join32(float x)
{
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));     // a = {lo, lo}, b = {hi, hi}
    return SUM ? a + b : fmaxf(a, b);
}
__device__ __forceinline__ void pair16(float x, float &a, float &b)
{
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a = the even 16-lane row's value, b = the odd one's
}
__device__ __forceinline__ void pair32(float x, float &a, float &b)
{
    a = x;
    b = x;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // a = the lower half's value, b = the upper half's
}
#else
template <bool SUM>
__device__ __forceinline__ float join16(float x) { const float y = __shfl_xor(x, 16); return SUM ? x + y : fmaxf(x, y); }
template <bool SUM>
__device__ __forceinline__ float join32(float x) { const float y = __shfl_xor(x, 32); return SUM ? x + y : fmaxf(x, y); }
__device__ __forceinline__ void pair16(float x, float &a, float &b)
{
    const float y = __shfl_xor(x, 16);
    const bool odd = (threadIdx.x >> 4) & 1;
    a = odd ? y : x;
    b = odd ? x : y;
}
__device__ __forceinline__ void pair32(float x, float &a, float &b)
{
    const float y = __shfl_xor(x, 32);
    const bool hi = (threadIdx.x >> 5) & 1;
    a = hi ? y : x;
    b = hi ? x : y;
}
#endif

// grid = (H, B), block = NW*64.  qkv bf16 [B*L][3*H*64] (q | k | v thirds, head-major inside a third); out bf16 [B*L][H*64].
// Dynamic LDS: K image [L][64] then V image [L][64], both 128-B rows with the chunk XOR of lds_off().
template <int NW, int QT>
__global__ __launch_bounds__(NW * 64) void k_attn(const bf16_t *__restrict__ qkv, const unsigned long long *__restrict__ kmask,
                                                  bf16_t *__restrict__ out, int Lpad, int H, float scale_log2,
                                                  const int32_t *__restrict__ row_off, int T)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char kv[];
    unsigned char *Ks = kv, *Vs = kv + (size_t)((Lpad + 63) & ~63) * 128;
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int ld = 3 * H * 64;
    // packed rows (row_off != NULL): row b = tokens [row_off[b], row_off[b+1]) of the flat token axis; its 16-row query tiles and
    // 64-key tiles may reach into the next row's tokens -- those keys are masked, those query rows are computed and NOT stored
    int64_t r0s = (int64_t)b * Lpad;
    int L = Lpad;
    if (row_off) packed_row(row_off, b, T, Lpad, r0s, L);   // (clamped to the T tokens of the buffers, whatever row_off holds)
    const size_t r0 = (size_t)r0s;
    const int nkt = (Lpad + 63) >> 6;                // mask words per row; a 64-key tile may reach past L (those keys are masked)
    const unsigned long long *km = kmask + (size_t)b * nkt;
    const size_t last_row = (row_off ? (size_t)T : (size_t)gridDim.y * Lpad) - 1;   // never read past the buffer

    // last 64-key tile that holds a valid key: nothing beyond it is staged or visited
    int last = -1;
    for (int t = 0; t < nkt; ++t)
        if (km[t] != 0ull) last = t;
    const int Lk = (last + 1) * 64;

    const bf16_t *hbase = qkv + h * 64;
    for (int i = tid; i < Lk * 8; i += NW * 64) {
        const int r = i >> 3, c = i & 7;
        size_t gr = r0 + r;                           // keys >= L belong to the next row (or nothing): loaded, never used
        gr = gr < last_row ? gr : last_row;
        *reinterpret_cast<u32x4 *>(Ks + lds_off(r, c)) = *reinterpret_cast<const u32x4 *>(hbase + gr * ld + H * 64 + c * 8);
        *reinterpret_cast<u32x4 *>(Vs + lds_off(r, c)) = *reinterpret_cast<const u32x4 *>(hbase + gr * ld + 2 * H * 64 + c * 8);
    }
    __syncthreads();

    // Each wave works on QT 16-row query tiles AT ONCE so that every K fragment and every transposed V fragment it reads
    // from LDS feeds QT MFMAs: with one tile per wave the kernel moved 64 B/clk/wave through an LDS that delivers
    // 256 B/clk per CU -- LDS-bound at 8 waves by a factor of two (250 TFLOP/s at L = 512).
    const int nqt = (L + 15) >> 4;
    for (int qg = wave; qg * QT < nqt; qg += NW) {
        bf16x8 qf[QT][2];
        float mrun[QT], lrun[QT];
        f32x4 oacc[QT][4];
        bool live[QT];                                   // tile exists and starts before the last valid key tile's end
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
            const int q0 = (qg * QT + qi) * 16;
            live[qi] = q0 < L && q0 < Lk;
            size_t qr = r0 + (live[qi] ? q0 + c16 : 0);
            qr = qr < last_row ? qr : last_row;         // (a packed row's last tile may reach past the buffer's last token)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                qf[qi][ks] = *reinterpret_cast<const bf16x8 *>(qkv + qr * ld + h * 64 + 32 * ks + 8 * g);
            mrun[qi] = -INFINITY;
            lrun[qi] = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) oacc[qi][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        for (int kt = 0; kt <= last; ++kt) {
            const unsigned long long vm = km[kt];
            if (vm == 0ull) continue;
            f32x4 s[QT][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int qi = 0; qi < QT; ++qi) s[qi][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8 *>(Ks + lds_off(kt * 64 + t * 16 + c16, g + 4 * ks));
#pragma unroll
                    for (int qi = 0; qi < QT; ++qi)
                        s[qi][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qi][ks], s[qi][t], 0, 0, 0);
                }
            }
            // s[qi][t][r] = <K[kt*64 + 16t + 4g + r], Q[q0(qi) + c16]>
            // Softmax of the tile on the VALU, which is what bounds this kernel (16 scores per lane, tile and query tile):
            // the running maximum is kept on the RAW scores and the 1/sqrt(d)*log2(e) scale is folded into the exponent's fma;
            // the per-key validity select runs only on a tile that has a masked key (normally just a row's last tile);
            // fma / sum / rescale go through the packed-f32 pipe two elements at a time, exp2 is the bare v_exp_f32.
            const bool full = vm == ~0ull;               // wave-uniform
            const f32x2_t sc2 = {scale_log2, scale_log2};
            bf16x8 pb[QT][2];
#pragma unroll
            for (int qi = 0; qi < QT; ++qi) {
                float mloc = -INFINITY;
                if (full) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        mloc = fmaxf(fmaxf(mloc, s[qi][t][0]), s[qi][t][1]);
                        mloc = fmaxf(fmaxf(mloc, s[qi][t][2]), s[qi][t][3]);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool ok = (vm >> (16 * t + 4 * g + r)) & 1ull;
                            const float v = ok ? s[qi][t][r] : -INFINITY;
                            s[qi][t][r] = v;
                            mloc = fmaxf(mloc, v);
                        }
                }
                mloc = join32<false>(join16<false>(mloc));
                const float mnew = fmaxf(mrun[qi], mloc);  // raw-score maximum; finite: vm != 0 guarantees a valid key in this tile
                const float alpha = __builtin_amdgcn_exp2f((mrun[qi] - mnew) * scale_log2);   // first tile: 2^-inf = 0
                const float nb = -mnew * scale_log2;
                const f32x2_t nb2 = {nb, nb};
                f32x2_t ps2 = {0.f, 0.f};
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4 pk;
#pragma unroll
                    for (int j2 = 0; j2 < 4; ++j2) {         // elements 2*j2, 2*j2+1 of this half: accumulator t = 2*s2 + (j2 >> 1)
                        const f32x4 sv = s[qi][2 * s2 + (j2 >> 1)];
                        const f32x2_t raw = (j2 & 1) ? f32x2_t{sv[2], sv[3]} : f32x2_t{sv[0], sv[1]};
                        const f32x2_t e = __builtin_elementwise_fma(raw, sc2, nb2);   // masked: -inf * scale + nb = -inf -> p = 0
                        const f32x2_t pj = {__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
                        ps2 += pj;
                        pk[j2] = pack2(pj.x, pj.y);
                    }
                    pb[qi][s2] = __builtin_bit_cast(bf16x8, pk);
                }
                float psum = ps2.x + ps2.y;
                psum = join32<true>(join16<true>(psum));
                lrun[qi] = lrun[qi] * alpha + psum;
                mrun[qi] = mnew;
                const f32x2_t al2 = {alpha, alpha};
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const f32x2_t lo = f32x2_t{oacc[qi][dt][0], oacc[qi][dt][1]} * al2, hi = f32x2_t{oacc[qi][dt][2], oacc[qi][dt][3]} * al2;
                    oacc[qi][dt] = f32x4{lo.x, lo.y, hi.x, hi.y};
                }
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    // V^T fragment: keys {32*s2 + 4g + 0..3} and {32*s2 + 16 + 4g + 0..3} of d = 16*dt + c16, fetched by the
                    // transposing LDS read: lane 4q+p of each 16-lane group addresses row (key0 + q), 8 bytes at d-offset 4p
                    const int qq = c16 >> 2, pp = c16 & 3;
                    const int k0 = kt * 64 + 32 * s2 + 4 * g + qq;
                    const int off0 = lds_off(k0, 2 * dt + (pp >> 1)) + 8 * (pp & 1);
                    const int off1 = lds_off(k0 + 16, 2 * dt + (pp >> 1)) + 8 * (pp & 1);
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4 *)(Vs + off0));
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4 *)(Vs + off1));
                    typedef __attribute__((ext_vector_type(8))) short s16x8;
                    const s16x8 va = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                    for (int qi = 0; qi < QT; ++qi)
                        oacc[qi][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va), pb[qi][s2], oacc[qi][dt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
            const int q0 = (qg * QT + qi) * 16;
            if (q0 >= L || q0 + c16 >= L) continue;      // no such tile / a row of the NEXT packed row (never for padded rows: L % 16 == 0)
            bf16_t *orow = out + (r0 + q0 + c16) * (H * 64) + h * 64;
            // rows past the last valid token (not live): never read as keys nor pooled; written as zeros to stay finite
            const float inv = (live[qi] && lrun[qi] > 0.f) ? 1.f / lrun[qi] : 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                u32x2 o;
                o.x = pack2(oacc[qi][dt][0] * inv, oacc[qi][dt][1] * inv);
                o.y = pack2(oacc[qi][dt][2] * inv, oacc[qi][dt][3] * inv);
                *reinterpret_cast<u32x2 *>(orow + dt * 16 + 4 * g) = o;
            }
        }
    }
}

// ------------------------------------------------------------------ masked mean pool

// grid = (D/128, B), block 256: sent[b][d] = sum_{valid t} tok[b][t][d] / #valid    (unixcoder_provider.py:152-154)
// A lane owns two adjacent dims, wave w the tokens t = w (mod 4), eight of them in flight at a time (selected by the mask
// bit after the load: padded rows are real memory); the four partial sums are added in wave order.  (One thread per dim
// walking every token behind a branch on its mask bit was a chain of L dependent loads: 31 us at B = 16, L = 128.)
__global__ __launch_bounds__(256) void k_pool(const bf16_t *__restrict__ tok, const unsigned long long *__restrict__ kmask,
                                              float *__restrict__ sent, int Lpad, int D, const int32_t *__restrict__ row_off, int T)
{
    __shared__ float part[4][64][2];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = blockIdx.x * 128 + lane * 2;
    const int nw = (Lpad + 63) >> 6;
    const unsigned long long *km = kmask + (size_t)b * nw;
    int64_t r0s = (int64_t)b * Lpad;
    int L = Lpad;
    if (row_off) packed_row(row_off, b, T, Lpad, r0s, L);
    const size_t r0 = (size_t)r0s;
    const bf16_t *base = tok + r0 * D + d;
    float a0 = 0.f, a1 = 0.f;
    for (int t = wave; t < L; t += 32) {
        uint32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tt = t + 4 * i;
            v[i] = tt < L ? *reinterpret_cast<const uint32_t *>(base + (size_t)tt * D) : 0u;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tt = t + 4 * i;
            const bool on = tt < L && ((km[tt >> 6] >> (tt & 63)) & 1ull);
            a0 += on ? bf2f(v[i] & 0xffffu) : 0.f;
            a1 += on ? bf2f(v[i] >> 16) : 0.f;
        }
    }
    part[wave][lane][0] = a0;
    part[wave][lane][1] = a1;
    __syncthreads();
    if (wave) return;
    int cnt = 0;
    for (int w = 0; w < nw; ++w) cnt += __popcll(km[w]);
    float s0 = part[0][lane][0], s1 = part[0][lane][1];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        s0 += part[w][lane][0];
        s1 += part[w][lane][1];
    }
    // an all-pad row divides 0/0 exactly like the reference's mean
    *reinterpret_cast<float2 *>(sent + (size_t)b * D + d) = float2{s0 / (float)cnt, s1 / (float)cnt};
}

// ------------------------------------------------------------------ mid-size GEMM (512 < T <= a few thousand rows)
//
// From one query up to a few thousand rows the 256-row tiles run out of parallelism: at T = 2048 the N = 768
// GEMMs are 48 workgroups on 256 CUs, each walking its whole K in series with every LDS-DMA exposed to the full HBM/MALL
// latency (measured 1.1 us per k-step against 0.45 in steady state; 34 us per call, 54 % of the forward).  What is missing
// there is bytes in flight, not MFMA rate -- so: 64 x 64 tiles (8x the workgroups), two workgroups per CU, a 4-stage ring
// with three k-steps in flight per workgroup, and one raw barrier per step.  Staging, swizzle and the C^T fragment layout
// are k_gemm_nt's.  (A first version fetched MFMA operands straight from global memory: a wave's
// 16-byte pieces of 16 different rows cost the texture path one cache line each, and it stalled at ~260 TFLOP/s whatever
// the size -- LDS-DMA moves 8 full 128-byte rows per instruction.)  N % 64 == 0, K % 64 == 0.
template <int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_mid(const bf16_t *__restrict__ A, const bf16_t *__restrict__ W, const float *__restrict__ bias,
                                                     const bf16_t *__restrict__ R, bf16_t *__restrict__ C, int M, int N, int K,
                                                     const float *__restrict__ aux0 = nullptr, const float *__restrict__ rstats = nullptr,
                                                     float *__restrict__ partials = nullptr)
{
    constexpr int TM = 64, TN = 64, S = 4;
    constexpr int kStage = (TM + TN) * BK * 2;   // 16 KiB: A rows then W rows, 16 pieces of 1 KiB, 4 per wave
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * kStage];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, c16 = lane & 15;
    // workgroups are dealt round-robin over the 8 XCDs: give each XCD label a contiguous run of tiles (row panels), so the
    // column tiles sharing an A panel meet in one L2
    const int nb = N / TN, ntiles = ((M + TM - 1) / TM) * nb;
    const int run = (ntiles + 7) >> 3;
    const int t = (blockIdx.x & 7) * run + (blockIdx.x >> 3);
    if (t >= ntiles) return;                     // (before any barrier)
    const int m0 = (t / nb) * TM, n0 = (t % nb) * TN;
    const int nk = K / BK;
    const bf16_t *src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);   // row of the stacked [A;W] image: waves 0,1 stage A, waves 2,3 W
        const int c = (lane & 7) ^ (r & 7);               // slot s of row r holds chunk s ^ (r & 7)
        if (wave < 2) {
            int m = m0 + r;
            m = m < M ? m : M - 1;                        // rows past M: a valid row, result never stored
            src[i] = A + (size_t)m * K + c * 8;
        } else {
            src[i] = W + (size_t)(n0 + r - TM) * K + c * 8;
        }
    }
    auto issue = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void *)(smem + (kt & (S - 1)) * kStage + (wave * 4 + i) * 1024), 16, 0, 0);
    };
    for (int kt = 0; kt < 3 && kt < nk; ++kt) issue(kt);
    f32x4 acc[2][2];  // [nt][mt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < nk; ++it) {
        // stage `it` has landed once only the stages issued after it (at most two, 4 pieces each) remain outstanding
        const int later = nk - 1 - it;
        if (later >= 2)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (later == 1)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // every wave's pieces of stage `it` are in; all reads of stage it-1 are done
        if (it + 3 < nk) issue(it + 3);          // into the buffer stage it-1 just vacated
        const unsigned char *sa = smem + (it & (S - 1)) * kStage;
        const unsigned char *sw = sa + TM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[2], fw[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                fa[b] = *reinterpret_cast<const bf16x8 *>(sa + lds_off(wm * 32 + b * 16 + c16, g + 4 * ks));
                fw[b] = *reinterpret_cast<const bf16x8 *>(sw + lds_off(wn * 32 + b * 16 + c16, g + 4 * ks));
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
        }
    }
    // a lane holds, per (nt, mt), the 4 consecutive n = n0 + wn*32 + nt*16 + 4g + {0..3} of row m = m0 + wm*32 + mt*16 + c16
    const bool res_ln = EPI == 5 && rstats != nullptr;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + wm * 32 + mt * 16 + c16;
        const int mc = m < M ? m : M - 1;          // (rows past M compute on a valid row and store nothing: the row joins below need every lane)
        float2 st = float2{1.f, 0.f};
        if (EPI == 3 || EPI == 4 || res_ln) st = *reinterpret_cast<const float2 *>(rstats + 2 * (size_t)mc);
        LnAcc la;      // EPI 5: the row's 32-column slot of this wave
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + wn * 32 + nt * 16 + 4 * g;
            const float4 b4 = *reinterpret_cast<const float4 *>(bias + n);
            float o0 = acc[nt][mt][0] + b4.x, o1 = acc[nt][mt][1] + b4.y, o2 = acc[nt][mt][2] + b4.z, o3 = acc[nt][mt][3] + b4.w;
            if (EPI == 3 || EPI == 4) {      // "LN in" (crh_gemm256.hpp): the same two fma per element as the other tiled kernels
                const float4 c4 = *reinterpret_cast<const float4 *>(aux0 + n);
                o0 = __builtin_fmaf(acc[nt][mt][0], st.x, __builtin_fmaf(st.y, c4.x, b4.x));
                o1 = __builtin_fmaf(acc[nt][mt][1], st.x, __builtin_fmaf(st.y, c4.y, b4.y));
                o2 = __builtin_fmaf(acc[nt][mt][2], st.x, __builtin_fmaf(st.y, c4.z, b4.z));
                o3 = __builtin_fmaf(acc[nt][mt][3], st.x, __builtin_fmaf(st.y, c4.w, b4.w));
            }
            if (EPI == 1 || EPI == 4) {
                const f32x2_t ga = gelu_erf2(f32x2_t{o0, o1}), gb = gelu_erf2(f32x2_t{o2, o3});
                o0 = ga.x;
                o1 = ga.y;
                o2 = gb.x;
                o3 = gb.y;
            }
            if (EPI == 2 || EPI == 5) {
                const u32x2 r2 = *reinterpret_cast<const u32x2 *>(R + (size_t)mc * N + n);
                const float h0 = bf2f(r2.x & 0xffffu), h1 = bf2f(r2.x >> 16), h2 = bf2f(r2.y & 0xffffu), h3 = bf2f(r2.y >> 16);
                if (res_ln) {
                    const float4 g4 = *reinterpret_cast<const float4 *>(aux0 + n);
                    o0 = __builtin_fmaf(__builtin_fmaf(h0, st.x, st.y), g4.x, o0);
                    o1 = __builtin_fmaf(__builtin_fmaf(h1, st.x, st.y), g4.y, o1);
                    o2 = __builtin_fmaf(__builtin_fmaf(h2, st.x, st.y), g4.z, o2);
                    o3 = __builtin_fmaf(__builtin_fmaf(h3, st.x, st.y), g4.w, o3);
                } else {
                    o0 += h0;
                    o1 += h1;
                    o2 += h2;
                    o3 += h3;
                }
            }
            u32x2 o;
            o.x = pack2(o0, o1);
            o.y = pack2(o2, o3);
            if (EPI == 5) ln_acc4(la, nt == 0, bf2f(o.x & 0xffffu), bf2f(o.x >> 16), bf2f(o.y & 0xffffu), bf2f(o.y >> 16));
            if (m < M) *reinterpret_cast<u32x2 *>(C + (size_t)m * N + n) = o;
        }
        if (EPI == 5) {
            const float2 p = ln_join_row(ln_acc_done8(la));
            if (g == 0 && m < M) *reinterpret_cast<float2 *>(partials + ((size_t)m * (N >> 5) + ((n0 >> 5) + wn)) * 2) = p;
        }
    }
}

#include "crh_gemm256.hpp"

}  // namespace enc
}  // namespace crh

using namespace crh;
using namespace crh::enc;

namespace {
// Grid: a multiple of 8 workgroups (one XCD label each).  Default: persistent, capped at the CU count, each workgroup walking
// several tiles with one pipeline (whole forward 9.06 vs 9.29 ms against one workgroup per tile, same process, B=256 L=128;
// the two are within a few percent -- CODERAG_HIP_GEMM_PERSISTENT=0 selects one workgroup per tile).
// Both tiled kernels hand each of the 8 XCD labels a contiguous share of the row panels (>= 16 panels) or of the tiles, and a
// label's gridDim/8 workgroups stride through that share: size the grid by the BUSIEST label.  (Sizing it by the total --
// min(tiles, CUs) -- gave T = 5120, N = 768 a grid of 120: 15 workgroups per label for the 18 tiles of a 3-panel label,
// i.e. two rounds, 93 us against 50 at both T = 4096 and T = 6144.)
unsigned xcd_grid(int64_t panels, int64_t nb, int64_t max_per_label)
{
    const int64_t busiest = panels >= 16 ? crh::ceil_div(panels, 8) * nb : crh::ceil_div(panels * nb, 8);
    return (unsigned)(8 * std::max<int64_t>(1, std::min(busiest, max_per_label)));
}
unsigned gemm_grid(int T, int N)
{
    static int persistent = -1;
    if (persistent < 0) {
        const char *e = getenv("CODERAG_HIP_GEMM_PERSISTENT");
        persistent = (e && e[0] == '0') ? 0 : 1;
    }
    return xcd_grid(crh::ceil_div(T, BM), N / BN, persistent ? crh::current_device_cus() / 8 : INT32_MAX);
}

template <int EPI>
int launch_mid(const void *x, const void *w, const float *bias, const void *res, void *y, int T, int N, int K, hipStream_t st,
               const float *aux0 = nullptr, const float *rstats = nullptr, float *partials = nullptr)
{
    const int64_t tiles = crh::ceil_div(T, 64) * (N / 64);
    hipLaunchKernelGGL((k_gemm_mid<EPI>), dim3((unsigned)(crh::ceil_div(tiles, 8) * 8)), dim3(256), 0, st, (const bf16_t *)x, (const bf16_t *)w, bias,
                       (const bf16_t *)res, (bf16_t *)y, T, N, K, aux0, rstats, partials);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}
// ---- which tiled kernel?  Three kernels compete and the winner flips with the shape (measured per
// GEMM, tools/gemm_mid_sweep.py): a launch costs (rounds of its busiest XCD label) x (one tile's walk through K), so
// k_gemm_nt wins while its 256x128 tiles fill the chip in one round, the 256x256 ping-pong kernel wherever halving the
// tile count saves a round (and everywhere at scale), k_gemm_mid while both leave most CUs idle.  The constants are fits
// (us) to that sweep at K = 768 / 3072; later rounds of a persistent launch overlap with the previous one (x 0.85).
enum GemmKernel { GEMM_NT, GEMM_PP, GEMM_MID };
int env_mode(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
bool pp_allowed(int T, int N, int K)
{
    if (N % g256::BN || K % (2 * g256::BK) || K < 4 * g256::BK) return false;
    return (int64_t)T * K * 2 < (1LL << 32) && (int64_t)N * K * 2 < (1LL << 32) && (int64_t)T * N * 2 < (1LL << 32);   // 32-bit byte offsets
}
GemmKernel choose_gemm(int T, int N, int K, int act)
{
    static const int pp_mode = env_mode("CODERAG_HIP_GEMM256", 1);   // 0 never, 1 by cost, 2 whenever the shape allows
    static const int mid_mode = env_mode("CODERAG_HIP_MID", 1);      // 0 never, 1 by cost, 2 whenever the shape allows
    const bool pp_ok = pp_mode != 0 && pp_allowed(T, N, K), mid_ok = mid_mode != 0 && N % 64 == 0;
    if (mid_ok && mid_mode == 2) return GEMM_MID;
    if (pp_ok && pp_mode == 2) return GEMM_PP;
    const double kf = K / 768.0;
    auto tiled = [&](int bm, int bn, double per_round) {
        const int64_t panels = crh::ceil_div(T, bm), nb = N / bn;
        const int64_t busiest = panels >= 16 ? crh::ceil_div(panels, 8) * nb : crh::ceil_div(panels * nb, 8);
        const int64_t rounds = crh::ceil_div(busiest, 32);
        return per_round * (1.0 + 0.85 * (double)(rounds - 1));
    };
    const double nt = tiled(BM, BN, 11.0 * kf + 6.0 + (act ? 2.0 : 0.0));
    const double pp = pp_ok ? tiled(g256::BM, g256::BN, 13.0 * kf + 8.5 + (act ? 3.5 : 0.0)) : 1e30;
    const double mid = mid_ok ? 4.5 + 1.9 * kf + (double)(crh::ceil_div(T, 64) * (N / 64)) * kf * 0.0103 * (act ? 1.25 : 1.0) : 1e30;
    if (mid < nt && mid < pp) return GEMM_MID;
    return pp < nt ? GEMM_PP : GEMM_NT;
}
unsigned gemm256_grid(int T, int N)
{
    return xcd_grid(crh::ceil_div(T, g256::BM), N / g256::BN, crh::current_device_cus() / 8);
}
int launch_gemm256(int epi, const void *x, const void *w, const float *bias, const void *res, void *y, int T, int N, int K, hipStream_t st,
                   const float *aux0 = nullptr, const float *rstats = nullptr, float *partials = nullptr)
{
    static OncePerDevice once;
    if (once.need()) {
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g256::k_gemm_pp<0>), hipFuncAttributeMaxDynamicSharedMemorySize, g256::kLds));
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g256::k_gemm_pp<1>), hipFuncAttributeMaxDynamicSharedMemorySize, g256::kLds));
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g256::k_gemm_pp<2>), hipFuncAttributeMaxDynamicSharedMemorySize, g256::kLds));
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g256::k_gemm_pp<3>), hipFuncAttributeMaxDynamicSharedMemorySize, g256::kLds));
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g256::k_gemm_pp<4>), hipFuncAttributeMaxDynamicSharedMemorySize, g256::kLds));
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g256::k_gemm_pp<5>), hipFuncAttributeMaxDynamicSharedMemorySize, g256::kLds));
    }
    const dim3 grid(gemm256_grid(T, N)), block(g256::WAVES * 64);
    const bf16_t *xa = (const bf16_t *)x, *wa = (const bf16_t *)w, *ra = (const bf16_t *)res;
    bf16_t *ya = (bf16_t *)y;
    switch (epi) {
    case 0: hipLaunchKernelGGL((g256::k_gemm_pp<0>), grid, block, g256::kLds, st, xa, wa, bias, ra, ya, T, N, K, aux0, rstats, partials); break;
    case 1: hipLaunchKernelGGL((g256::k_gemm_pp<1>), grid, block, g256::kLds, st, xa, wa, bias, ra, ya, T, N, K, aux0, rstats, partials); break;
    case 2: hipLaunchKernelGGL((g256::k_gemm_pp<2>), grid, block, g256::kLds, st, xa, wa, bias, ra, ya, T, N, K, aux0, rstats, partials); break;
    case 3: hipLaunchKernelGGL((g256::k_gemm_pp<3>), grid, block, g256::kLds, st, xa, wa, bias, ra, ya, T, N, K, aux0, rstats, partials); break;
    case 4: hipLaunchKernelGGL((g256::k_gemm_pp<4>), grid, block, g256::kLds, st, xa, wa, bias, ra, ya, T, N, K, aux0, rstats, partials); break;
    case 5: hipLaunchKernelGGL((g256::k_gemm_pp<5>), grid, block, g256::kLds, st, xa, wa, bias, ra, ya, T, N, K, aux0, rstats, partials); break;
    default: return fail(CRH_E_INTERNAL, "gemm256: epilogue %d", epi);
    }
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

constexpr size_t kGemmLds = (size_t)STAGES * (BM + BN) * BK * 2;  // 144 KB of the CU's 160 KB

int gemm_lds_attr()
{
    static OncePerDevice once;
    if (once.need()) {
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt<2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt<3, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt<5, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
    }
    return CRH_OK;
}
}  // namespace

extern "C" {

int crh_gemm_bf16_bias(const void *x, const void *w, const float *bias, void *y, int T, int N, int K, int act, void *stream)
{
    if (!x || !w || !bias || !y) return fail(CRH_E_INVALID, "gemm: NULL pointer");
    if (T <= 0 || N <= 0 || K <= 0 || N % BN || K % BK) return fail(CRH_E_INVALID, "gemm: shape T=%d N=%d K=%d (need N%%128==0, K%%64==0)", T, N, K);
    if (act != 0 && act != 1) return fail(CRH_E_INVALID, "gemm: act=%d (0 none, 1 gelu)", act);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const GemmKernel which = choose_gemm(T, N, K, act);
    if (which == GEMM_MID) return act == 1 ? launch_mid<1>(x, w, bias, nullptr, y, T, N, K, st) : launch_mid<0>(x, w, bias, nullptr, y, T, N, K, st);
    if (which == GEMM_PP) return launch_gemm256(act, x, w, bias, nullptr, y, T, N, K, st);
    const dim3 grid(gemm_grid(T, N));
    CRH_TRY(gemm_lds_attr());
    if (act == 1)
        hipLaunchKernelGGL((k_gemm_nt<1, 0>), grid, dim3(GEMM_WAVES * 64), kGemmLds, st, (const bf16_t *)x, (const bf16_t *)w, bias, (const bf16_t *)nullptr, (bf16_t *)y, T, N, K);
    else
        hipLaunchKernelGGL((k_gemm_nt<0, 0>), grid, dim3(GEMM_WAVES * 64), kGemmLds, st, (const bf16_t *)x, (const bf16_t *)w, bias, (const bf16_t *)nullptr, (bf16_t *)y, T, N, K);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

#ifdef CRH_ENABLE_DEBUG   // libcoderag_hip_debug.so only (build.sh): never exported by the product library
// timing ablations of the GEMM main loop (tools/gemm_ablate.py); results are meaningless for variant != 0
int crh_debug_gemm_variant(const void *x, const void *w, const float *bias, void *y, int T, int N, int K, int variant, void *stream)
{
    if (!x || !w || !bias || !y || T <= 0 || N % BN || K % BK) return fail(CRH_E_INVALID, "debug gemm: bad arguments");
    const dim3 grid(gemm_grid(T, N));
    hipStream_t st = static_cast<hipStream_t>(stream);
    CRH_TRY(gemm_lds_attr());
#define CRH_DBG_LAUNCH(V)                                                                                                   \
    do {                                                                                                                    \
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_nt<0, V>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds)); \
        hipLaunchKernelGGL((k_gemm_nt<0, V>), grid, dim3(GEMM_WAVES * 64), kGemmLds, st, (const bf16_t *)x, (const bf16_t *)w, bias,  \
                           (const bf16_t *)nullptr, (bf16_t *)y, T, N, K);                                                 \
    } while (0)
    switch (variant) {
    case 16:   // the 256x256 ping-pong kernel regardless of the tile count (results are valid)
        if (N % g256::BN || K % (2 * g256::BK) || K < 4 * g256::BK || (int64_t)T * K * 2 >= (1LL << 32))
            return fail(CRH_E_INVALID, "debug gemm: variant 16 needs N%%256==0, K%%128==0, K>=256, T*K*2 < 4 GiB");
        return launch_gemm256(0, x, w, bias, nullptr, y, T, N, K, st);
    case 19:   // EPI 2 (bias + residual, residual = y's previous contents) without the LayerNorm: isolates the residual epilogue
        if (N % g256::BN || K % (2 * g256::BK) || K < 4 * g256::BK || (int64_t)T * K * 2 >= (1LL << 32)) return fail(CRH_E_INVALID, "debug gemm: bad shape");
        return launch_gemm256(2, x, w, bias, y, y, T, N, K, st);
    case 20: {   // no epilogue + 1.5x LDS-DMA + extra fragment reads (what a two-pass half-height schedule would load)
        if (N % g256::BN || K % (2 * g256::BK) || K < 4 * g256::BK || (int64_t)T * K * 2 >= (1LL << 32)) return fail(CRH_E_INVALID, "debug gemm: bad shape");
        const dim3 grid256(gemm256_grid(T, N)), block(g256::WAVES * 64);
        CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g256::k_gemm_pp<0, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, g256::kLds));
        hipLaunchKernelGGL((g256::k_gemm_pp<0, 6>), grid256, block, g256::kLds, st, (const bf16_t *)x, (const bf16_t *)w, bias, (const bf16_t *)nullptr, (bf16_t *)y, T, N, K);
        CRH_HIP(hipGetLastError());
        return CRH_OK;
    }
    case 17:
    case 18: {   // ablations of the ping-pong kernel's epilogue: 17 = no global stores, 18 = no epilogue
        if (N % g256::BN || K % (2 * g256::BK) || K < 4 * g256::BK || (int64_t)T * K * 2 >= (1LL << 32)) return fail(CRH_E_INVALID, "debug gemm: bad shape");
        const dim3 grid256(gemm256_grid(T, N)), block(g256::WAVES * 64);
        if (variant == 17) {
            CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g256::k_gemm_pp<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, g256::kLds));
            hipLaunchKernelGGL((g256::k_gemm_pp<0, 1>), grid256, block, g256::kLds, st, (const bf16_t *)x, (const bf16_t *)w, bias, (const bf16_t *)nullptr, (bf16_t *)y, T, N, K);
        } else {
            CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(g256::k_gemm_pp<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, g256::kLds));
            hipLaunchKernelGGL((g256::k_gemm_pp<0, 2>), grid256, block, g256::kLds, st, (const bf16_t *)x, (const bf16_t *)w, bias, (const bf16_t *)nullptr, (bf16_t *)y, T, N, K);
        }
        CRH_HIP(hipGetLastError());
        return CRH_OK;
    }
    case 0: CRH_DBG_LAUNCH(0); break;
    case 1: CRH_DBG_LAUNCH(1); break;
    case 2: CRH_DBG_LAUNCH(2); break;
    case 4: CRH_DBG_LAUNCH(4); break;
    case 5: CRH_DBG_LAUNCH(5); break;
    case 6: CRH_DBG_LAUNCH(6); break;
    case 8: CRH_DBG_LAUNCH(8); break;
    default: return fail(CRH_E_INVALID, "debug gemm: variant %d", variant);
    }
#undef CRH_DBG_LAUNCH
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}
#endif  // CRH_ENABLE_DEBUG

// One GEMM of any tiled kernel with epilogue `epi` (0 bias, 2 bias + residual)
static int launch_tiled(int epi, const void *x, const void *w, const float *bias, const void *res, void *y, int T, int N, int K, hipStream_t st)
{
    const GemmKernel which = choose_gemm(T, N, K, 0);
    if (which == GEMM_MID) return epi == 2 ? launch_mid<2>(x, w, bias, res, y, T, N, K, st) : launch_mid<0>(x, w, bias, nullptr, y, T, N, K, st);
    if (which == GEMM_PP) return launch_gemm256(epi, x, w, bias, res, y, T, N, K, st);
    CRH_TRY(gemm_lds_attr());
    if (epi == 2)
        hipLaunchKernelGGL((k_gemm_nt<2, 0>), dim3(gemm_grid(T, N)), dim3(GEMM_WAVES * 64), kGemmLds, st, (const bf16_t *)x, (const bf16_t *)w, bias,
                           (const bf16_t *)res, (bf16_t *)y, T, N, K);
    else
        hipLaunchKernelGGL((k_gemm_nt<0, 0>), dim3(gemm_grid(T, N)), dim3(GEMM_WAVES * 64), kGemmLds, st, (const bf16_t *)x, (const bf16_t *)w, bias,
                           (const bf16_t *)nullptr, (bf16_t *)y, T, N, K);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

int crh_gemm_bf16_bias_res_ln(const void *x, const void *w, const float *bias, const void *residual, const float *gamma,
                              const float *beta, float eps, void *y, int T, int N, int K, void *stream)
{
    if (!x || !w || !bias || !residual || !gamma || !beta || !y) return fail(CRH_E_INVALID, "gemm_res_ln: NULL pointer");
    if (N != 768) return fail(CRH_E_INVALID, "gemm_res_ln: N=%d (the fused LayerNorm is built for 768)", N);
    if (T <= 0 || K <= 0 || K % BK) return fail(CRH_E_INVALID, "gemm_res_ln: shape T=%d K=%d", T, K);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // WHERE the residual joins depends on the shape of the operation only, never on T or on which kernel the cost model picks --
    // so that a token's result has the same rounding points in a batch of 40 tokens and in one of 65 536 (the tiled kernels
    // add the same products in the same order; tests/test_encoder_gpu.py pins a chunk's embedding across batches to the bit):
    //   K <= 1024 (the O-projection): bias-only epilogue, the GEMM output rounded to bf16, residual added in f32 by the LayerNorm
    //     kernel (k_layernorm768_res).  The residual epilogue would pull 128 KB of cold residual per CU and tile with the
    //     matrix pipe idle: +31 us per call at 65 k tokens against +18 us for the LayerNorm's extra read.
    //   K  > 1024 (FFN2): residual added to the f32 accumulator in the GEMM epilogue, one rounding, plain LayerNorm after.
    // (y == residual with K <= 1024 cannot take the first form -- the GEMM would overwrite the residual -- and falls to the second.)
    const bool ln_side = K <= 1024 && y != residual;
    const int epi = ln_side ? 0 : 2;
    CRH_TRY(launch_tiled(epi, x, w, bias, residual, y, T, N, K, st));
    if (ln_side)
        hipLaunchKernelGGL(k_layernorm768_res, dim3((unsigned)ceil_div(T, 4)), dim3(256), 0, st, (bf16_t *)y, (const bf16_t *)residual, gamma, beta, eps, T);
    else
        hipLaunchKernelGGL(k_layernorm768, dim3((unsigned)ceil_div(T, 4)), dim3(256), 0, st, (bf16_t *)y, gamma, beta, eps, T);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

int crh_gemm_bf16_bias_res32_ln(const void *x, const void *w, const float *bias, float *residual_f32, const float *gamma, const float *beta,
                                float eps, void *y, int T, int N, int K, void *stream)
{
    if (!x || !w || !bias || !residual_f32 || !gamma || !beta || !y) return fail(CRH_E_INVALID, "gemm_res32_ln: NULL pointer");
    if (N != 768) return fail(CRH_E_INVALID, "gemm_res32_ln: N=%d (the LayerNorm is built for 768)", N);
    if (T <= 0 || K <= 0 || K % BK) return fail(CRH_E_INVALID, "gemm_res32_ln: shape T=%d K=%d", T, K);
    hipStream_t st = static_cast<hipStream_t>(stream);
    CRH_TRY(launch_tiled(0, x, w, bias, nullptr, y, T, N, K, st));      // bias-only epilogue at every K: the residual joins in f32, below
    hipLaunchKernelGGL(k_layernorm768_res32, dim3((unsigned)ceil_div(T, 4)), dim3(256), 0, st, (bf16_t *)y, residual_f32, gamma, beta, eps, T);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

// ---- LayerNorm folded into the GEMMs around it (ABI 4; kernels and arithmetic: the comment above struct LnAcc)
int crh_gemm_bf16_lnin(const void *x, const float *row_stats, const void *w_scaled, const float *colsum, const float *bias_folded, void *y,
                       int T, int N, int K, int act, void *stream)
{
    if (!x || !row_stats || !w_scaled || !colsum || !bias_folded || !y) return fail(CRH_E_INVALID, "gemm_lnin: NULL pointer");
    if (T <= 0 || N <= 0 || K <= 0 || N % BN || K % BK) return fail(CRH_E_INVALID, "gemm_lnin: shape T=%d N=%d K=%d (need N%%128==0, K%%64==0)", T, N, K);
    if (act != 0 && act != 1) return fail(CRH_E_INVALID, "gemm_lnin: act=%d (0 none, 1 gelu)", act);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const GemmKernel which = choose_gemm(T, N, K, act);
    if (which == GEMM_MID)
        return act == 1 ? launch_mid<4>(x, w_scaled, bias_folded, nullptr, y, T, N, K, st, colsum, row_stats)
                        : launch_mid<3>(x, w_scaled, bias_folded, nullptr, y, T, N, K, st, colsum, row_stats);
    if (which == GEMM_PP) return launch_gemm256(3 + act, x, w_scaled, bias_folded, nullptr, y, T, N, K, st, colsum, row_stats, nullptr);
    const dim3 grid(gemm_grid(T, N));
    CRH_TRY(gemm_lds_attr());
    if (act == 1)
        hipLaunchKernelGGL((k_gemm_nt<4, 0>), grid, dim3(GEMM_WAVES * 64), kGemmLds, st, (const bf16_t *)x, (const bf16_t *)w_scaled, bias_folded,
                           (const bf16_t *)nullptr, (bf16_t *)y, T, N, K, colsum, row_stats);
    else
        hipLaunchKernelGGL((k_gemm_nt<3, 0>), grid, dim3(GEMM_WAVES * 64), kGemmLds, st, (const bf16_t *)x, (const bf16_t *)w_scaled, bias_folded,
                           (const bf16_t *)nullptr, (bf16_t *)y, T, N, K, colsum, row_stats);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

int crh_gemm_bf16_res_lnstats(const void *x, const void *w, const float *bias, const void *residual, const float *res_stats,
                              const float *res_gamma, float eps, void *y, float *partials, float *stats_out, int T, int N, int K, void *stream)
{
    if (!x || !w || !bias || !residual || !y || !partials || !stats_out) return fail(CRH_E_INVALID, "gemm_res_lnstats: NULL pointer");
    if (res_stats && !res_gamma) return fail(CRH_E_INVALID, "gemm_res_lnstats: residual statistics without the gain");
    if (N != 768) return fail(CRH_E_INVALID, "gemm_res_lnstats: N=%d (built for 768)", N);
    if (T <= 0 || K <= 0 || K % BK) return fail(CRH_E_INVALID, "gemm_res_lnstats: shape T=%d K=%d", T, K);
    if (y == residual) return fail(CRH_E_INVALID, "gemm_res_lnstats: the output must not overwrite the residual");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const GemmKernel which = choose_gemm(T, N, K, 0);
    if (which == GEMM_MID) {
        CRH_TRY(launch_mid<5>(x, w, bias, residual, y, T, N, K, st, res_gamma, res_stats, partials));
    } else if (which == GEMM_PP) {
        CRH_TRY(launch_gemm256(5, x, w, bias, residual, y, T, N, K, st, res_gamma, res_stats, partials));
    } else {
        CRH_TRY(gemm_lds_attr());
        hipLaunchKernelGGL((k_gemm_nt<5, 0>), dim3(gemm_grid(T, N)), dim3(GEMM_WAVES * 64), kGemmLds, st, (const bf16_t *)x, (const bf16_t *)w, bias,
                           (const bf16_t *)residual, (bf16_t *)y, T, N, K, res_gamma, res_stats, partials);
        CRH_HIP(hipGetLastError());
    }
    static_assert(kLnSlots == 768 / 32, "k_ln_finalize is built for 768 columns");
    hipLaunchKernelGGL(k_ln_finalize, dim3((unsigned)ceil_div(T, 64)), dim3(64), 0, st, (const float *)partials, stats_out, T, eps);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

int crh_layernorm_apply(const void *x, const float *row_stats, const float *gamma, const float *beta, void *y, int T, int N, void *stream)
{
    if (!x || !row_stats || !gamma || !beta || !y) return fail(CRH_E_INVALID, "layernorm_apply: NULL pointer");
    if (N != 768 || T <= 0) return fail(CRH_E_INVALID, "layernorm_apply: shape T=%d N=%d (built for 768)", T, N);
    hipLaunchKernelGGL(k_ln_apply768, dim3((unsigned)ceil_div(T, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16_t *)x, row_stats, gamma, beta,
                       (bf16_t *)y, T);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

static int attn_launch(const void *qkv, const uint64_t *kmask, void *out, const int32_t *row_off, int B, int T, int L, int H, void *stream);

// ---- verdict of the device-side row_off checks: one word per device in pinned host memory (the kernel ORs into it at system
// scope; the host peeks at it without synchronising).  Bits: see k_check_row_off.
static unsigned int *packed_status_word()
{
    static unsigned int *words[64] = {};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return nullptr;
    d &= 63;
    if (!words[d]) {
        void *p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocDefault) != hipSuccess) return nullptr;
        *static_cast<volatile unsigned int *>(p) = 0u;
        words[d] = static_cast<unsigned int *>(p);
    }
    return words[d];
}
// CRH_E_INVALID once a check has found a bad offsets array (the word is cleared: one report per offence)
static int packed_verdict()
{
    unsigned int *w = packed_status_word();
    if (!w) return fail(CRH_E_HIP, "packed rows: no status word (hipHostMalloc failed)");
    const unsigned int bad = __atomic_exchange_n(w, 0u, __ATOMIC_ACQ_REL);
    if (!bad) return CRH_OK;
    return fail(CRH_E_INVALID, "packed rows: the device-side check rejected row_off:%s%s%s%s (what it described was clamped to the buffers; the results of that forward are meaningless)",
                (bad & 1u) ? " row_off[0] != 0;" : "", (bad & 2u) ? " offsets decrease;" : "", (bad & 4u) ? " a row is longer than Lmax;" : "",
                (bad & 8u) ? " row_off[B] != T;" : "");
}

int crh_encoder_finish(void *stream)
{
    CRH_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return packed_verdict();
}

int crh_attn_fwd_varlen(const void *qkv, const uint64_t *kmask, void *out, int B, int L, int H, void *stream)
{
    return attn_launch(qkv, kmask, out, nullptr, B, B * L, L, H, stream);
}

int crh_attn_fwd_packed(const void *qkv, const int32_t *row_off, const uint64_t *kmask, void *out, int B, int T, int Lmax, int H, void *stream)
{
    if (!row_off) return fail(CRH_E_INVALID, "attn_packed: row_off is NULL");
    if (T <= 0) return fail(CRH_E_INVALID, "attn_packed: T=%d tokens", T);
    CRH_TRY(packed_verdict());
    return attn_launch(qkv, kmask, out, row_off, B, T, Lmax, H, stream);
}

static int attn_launch(const void *qkv, const uint64_t *kmask, void *out, const int32_t *row_off, int B, int T, int L, int H, void *stream)
{
    if (!qkv || !kmask || !out) return fail(CRH_E_INVALID, "attn: NULL pointer");
    if (B <= 0 || H <= 0 || L <= 0 || L % 16 || L > 512) return fail(CRH_E_INVALID, "attn: B=%d L=%d H=%d (need L%%16==0, L<=512)", B, L, H);
    const float scale_log2 = 0.125f * 1.4426950408889634f;  // 64^-1/2 * log2(e)
    const size_t lds = (size_t)((L + 63) & ~63) * 256;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // One 16-row query tile per wave at a time (101 VGPRs: 4-5 waves per SIMD) and as many waves per CU as the K/V images
    // allow: up to 320 tokens two 8-wave workgroups share a CU's 160 KB, beyond that one 16-wave workgroup.  The kernel is
    // bound by the softmax VALU, so what pays is many resident waves whose VALU, LDS and MFMA phases interleave -- measured
    // (tools/attn_bench.py, us at ~65k tokens, L = 208 / 256 / 384 / 512): 4 tiles per wave x 8 waves 245 / 208 / 178 / 235,
    // 2 tiles x 8 waves 175 / 149 / 175 / 211, 1 tile x 8 waves 127 / 115 / 194 / 229, 1 tile x 16 waves 153 / 133 / 159 / 199.
    const int nqt = L / 16;
#define CRH_ATTN(NW_, QT_)                                                                                                      \
    do {                                                                                                                        \
        static OncePerDevice once;                                                                                              \
        if (once.need()) {                                                                                                      \
            CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_attn<NW_, QT_>), hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 256)); \
        }                                                                                                                       \
        hipLaunchKernelGGL((k_attn<NW_, QT_>), dim3(H, B), dim3(NW_ * 64), lds, st, (const bf16_t *)qkv,                        \
                           (const unsigned long long *)kmask, (bf16_t *)out, L, H, scale_log2, row_off, T);                     \
    } while (0)
    static int force = -1;   // CODERAG_HIP_ATTN_CFG=<waves> (4, 8, 16; tuning only)
    if (force < 0) {
        const char *e = getenv("CODERAG_HIP_ATTN_CFG");
        force = e ? atoi(e) : 0;
    }
    const int waves = force ? force : (nqt <= 4 ? 4 : nqt <= 20 ? 8 : 16);
    if (waves == 4)
        CRH_ATTN(4, 1);
    else if (waves == 8)
        CRH_ATTN(8, 1);
    else
        CRH_ATTN(16, 1);
#undef CRH_ATTN
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

int crh_embed_ln(const int32_t *ids, const void *word, const void *pos, const void *type0, const float *gamma, const float *beta,
                 float eps, int pad_id, void *out, uint64_t *kmask, int B, int L, int D, void *stream)
{
    if (!ids || !word || !pos || !type0 || !gamma || !beta || !out || !kmask) return fail(CRH_E_INVALID, "embed_ln: NULL pointer");
    if (B <= 0 || L <= 0 || L % 16 || L > 1024 || D != 768) return fail(CRH_E_INVALID, "embed_ln: B=%d L=%d D=%d (need L%%16==0, D==768)", B, L, D);
    hipLaunchKernelGGL(k_embed_ln, dim3(B, L / 16), dim3(256), (size_t)L * 4, static_cast<hipStream_t>(stream), ids, (const bf16_t *)word,
                       (const bf16_t *)pos, (const bf16_t *)type0, gamma, beta, eps, pad_id, (bf16_t *)out, (unsigned long long *)kmask, L, D,
                       (const int32_t *)nullptr, B * L);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}
int crh_embed_ln_packed(const int32_t *ids, const int32_t *row_off, const void *word, const void *pos, const void *type0, const float *gamma,
                        const float *beta, float eps, int pad_id, void *out, uint64_t *kmask, int B, int T, int Lmax, int D, void *stream)
{
    if (!ids || !row_off || !word || !pos || !type0 || !gamma || !beta || !out || !kmask) return fail(CRH_E_INVALID, "embed_ln_packed: NULL pointer");
    if (B <= 0 || T <= 0 || Lmax <= 0 || Lmax % 16 || Lmax > 1024 || D != 768)
        return fail(CRH_E_INVALID, "embed_ln_packed: B=%d T=%d Lmax=%d D=%d (need T>0, Lmax%%16==0, D==768)", B, T, Lmax, D);
    CRH_TRY(packed_verdict());
    // the first call of a forward: the whole offsets array is checked on the device (the kernels below only clamp)
    hipLaunchKernelGGL(k_check_row_off, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), row_off, B, T, Lmax, packed_status_word());
    CRH_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_embed_ln, dim3(B, Lmax / 16), dim3(256), (size_t)Lmax * 4, static_cast<hipStream_t>(stream), ids, (const bf16_t *)word,
                       (const bf16_t *)pos, (const bf16_t *)type0, gamma, beta, eps, pad_id, (bf16_t *)out, (unsigned long long *)kmask, Lmax, D, row_off, T);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

int crh_masked_mean_pool(const void *tok, const uint64_t *kmask, float *sent, int B, int L, int D, void *stream)
{
    if (!tok || !kmask || !sent) return fail(CRH_E_INVALID, "pool: NULL pointer");
    if (B <= 0 || L <= 0 || L % 16 || D % 128) return fail(CRH_E_INVALID, "pool: B=%d L=%d D=%d", B, L, D);
    hipLaunchKernelGGL(k_pool, dim3(D / 128, B), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16_t *)tok,
                       (const unsigned long long *)kmask, sent, L, D, (const int32_t *)nullptr, B * L);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}
int crh_masked_mean_pool_packed(const void *tok, const int32_t *row_off, const uint64_t *kmask, float *sent, int B, int T, int Lmax, int D, void *stream)
{
    if (!tok || !row_off || !kmask || !sent) return fail(CRH_E_INVALID, "pool_packed: NULL pointer");
    if (B <= 0 || T <= 0 || Lmax <= 0 || Lmax % 16 || D % 128) return fail(CRH_E_INVALID, "pool_packed: B=%d T=%d Lmax=%d D=%d", B, T, Lmax, D);
    CRH_TRY(packed_verdict());
    hipLaunchKernelGGL(k_pool, dim3(D / 128, B), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16_t *)tok,
                       (const unsigned long long *)kmask, sent, Lmax, D, row_off, T);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

}  // extern "C"
