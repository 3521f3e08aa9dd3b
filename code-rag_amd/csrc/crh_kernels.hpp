// crh_kernels.hpp -- device code of the HBM-resident cosine index (gfx950 / CDNA4 only).
//
// Replaces what runs inside the Qdrant server for the reference's
// QdrantManager.upsert / .search (src/lattice/embeddings/client.py:115-157):
// normalise on insert, score = dot of normalised vectors, top-k descending.
//
// HBM layout of the corpus ("tiled bf16"): rows are grouped in tiles of 32; one tile holds
// KSTEPS = dim/16 pieces of 1 KiB; piece s is exactly the A operand of one
// v_mfma_f32_32x32x16_bf16 for k = 16s..16s+15: lane l = 32h + r holds the 8 bf16
// X[32*tile + r][16s + 8h + 0..7], and finds them at slot 2r + h of the piece (the two halves of a row
// side by side: see piece_slot).  A wave therefore streams a tile with 1-KiB fully
// coalesced dwordx4 loads, straight into MFMA operand registers -- no LDS round trip,
// no transposition, every HBM byte fetched exactly once.  The (<= 64) queries sit in LDS
// in the matching B-operand layout for the whole kernel.
//
// Everything that decides a returned id or score is "canonical" f32 arithmetic: products
// and sums rounded separately, in index order, exactly as oracle/search_oracle.c does.
// The MFMA scan only nominates candidates; it never decides an order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace crh {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kTileRows = 32;
constexpr int kMaxQ = 64;   // queries per pass of the LDS-query scan k_scan (two 32-column MFMA B blocks)
constexpr int kWideQ = 256;  // queries per pass of the register-query scan k_scan_wide (eight waves x one 32-column block)

struct SearchStatus {
    unsigned int max_wave_cnt;  // largest per-wave candidate count (unclamped)
    unsigned int max_qcount;    // largest per-query candidate count (unclamped)
    unsigned int wave_overflow;
    unsigned int q_overflow;
    unsigned long long candidates;
    unsigned int bar_a, bar_b;   // arrival counters of the two grid-wide waits of k_scan_fused (zeroed with the slot)
    unsigned int bar_timeout;    // set when one of those waits gave up (a workgroup never became resident): results are void
    unsigned int next_chunk;     // k_scan_i8: tile chunks handed out so far beyond each workgroup's first (zeroed with the slot)
    unsigned int qcount[kWideQ];  // per-query candidate counters of the batch (the whole slot is zeroed by k_prep_queries)
    unsigned int qsurv[kMaxQ];    // k_select behind the int8 scan, several workgroups per query: fast-scored survivors published so far
    unsigned int qsurv2[kMaxQ];   // k_select_final, several workgroups per query: canonical keys published so far ...
    unsigned int qdone2[kMaxQ];   // ... and workgroups that have published (the last one ranks)
};

// ------------------------------------------------------------------ small helpers

__device__ __forceinline__ float bits_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f32_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// f32 -> bf16 bits, round to nearest even; NaN quieted (same integer recipe as the oracle's
// orc_bf16_round so that the stored corpus is bit-identical).
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float x)
{
    uint32_t u = f32_bits(x);
    if ((u & 0x7fffffffu) > 0x7f800000u) return ((u | 0x00400000u) >> 16);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_bits_f32(uint32_t b) { return bits_f32(b << 16); }

// order-preserving map f32 -> u32 (larger float <=> larger uint); no NaNs reach it.
__device__ __forceinline__ uint32_t ord_f32(float f)
{
    uint32_t u = f32_bits(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(uint32_t o)
{
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return bits_f32(u);
}

__device__ __forceinline__ int lane_id() { return __lane_id(); }

// address (in u32x4 units) of piece (tile, s) lane (h, r)
// Where the 16 bytes of MFMA lane (h, r) sit inside a 1-KiB piece: the two halves of a ROW are neighbours (32 contiguous bytes per
// row and piece).  A wave's load of a piece is one contiguous KiB either way; the canonical re-score of a single row, which
// fetches 64-byte sectors, gets 32 useful bytes per sector instead of 16 (its traffic is what bounds k_select behind the int8 scan).
// (Going further -- pieces 2j and 2j+1 interleaved per row, 64 contiguous bytes per row and piece pair, a full sector for the
// re-score -- was built and measured: a wave's load of one piece then covers half of each of 32 sectors, and the bf16 scan went
// from 2.33 to 2.69 ms, the wide scan from 8.4 to 9.1 ms, for 7 us less around the int8 pass.  Not kept.)
__device__ __forceinline__ constexpr int piece_off(int s) { return s * 64; }   // u32x4 units of piece s inside its tile
__device__ __forceinline__ int piece_slot(int h, int r) { return r * 2 + h; }
__device__ __forceinline__ int lane_slot(int lane) { return (lane & 31) * 2 + (lane >> 5); }
__device__ __forceinline__ size_t tiled_index(int64_t tile, int ksteps, int s, int h, int r)
{
    return (size_t)tile * ksteps * 64 + piece_off(s) + piece_slot(h, r);
}

// ------------------------------------------------------------------ append / preprocess

// Qdrant cosine_preprocess for one vector whose squared length is already summed.
// Returns the divisor (0 = leave unchanged).
__device__ __forceinline__ float cosine_divisor(float len2)
{
    if (len2 < 1.1920928955078125e-07f || fabsf(len2 - 1.0f) <= 1.0e-6f) return 0.0f;
    return __fsqrt_rn(len2);
}

// 32 rows per block, 256 threads.  src: raw f32 [n][dim]; rows land at first_row + i.
template <bool KEEP_F32>
__global__ __launch_bounds__(256) void k_append(const float *__restrict__ src, int64_t n, int64_t first_row,
                                                int dim, int ksteps, u32x4 *__restrict__ xt,
                                                float *__restrict__ xf32, int preprocessed)
{
    __shared__ float buf[32][65];
    __shared__ float divisor[32];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    const int nrows = (int)((n - row0) < 32 ? (n - row0) : 32);

    float acc = 0.0f;  // threads 0..31: running squared length of row tid, index order
    for (int kc = 0; kc < dim; kc += 64) {
        for (int e = tid; e < 32 * 64; e += 256) {
            int r = e >> 6, c = e & 63;
            buf[r][c] = (r < nrows) ? src[(row0 + r) * dim + kc + c] : 0.0f;
        }
        __syncthreads();
        if (tid < 32) {
#pragma unroll 8
            for (int c = 0; c < 64; ++c) {
                float v = buf[tid][c];
                float p = v * v;
                acc = acc + p;
            }
        }
        __syncthreads();
    }
    if (tid < 32) divisor[tid] = preprocessed ? 0.0f : cosine_divisor(acc);  // restored rows are stored verbatim
    __syncthreads();

    const int chunks = dim >> 3;  // 16-byte bf16 chunks per row
    for (int item = tid; item < chunks * 32; item += 256) {
        int r = item & 31, c8 = item >> 5;
        if (r >= nrows) continue;
        const float4 *sp = reinterpret_cast<const float4 *>(src + (row0 + r) * dim + c8 * 8);
        float4 a = sp[0], b = sp[1];
        float dv = divisor[r];
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if (dv != 0.0f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __fdiv_rn(v[j], dv);
        }
        int64_t gr = first_row + row0 + r;
        if (KEEP_F32) {
            float4 *dp = reinterpret_cast<float4 *>(xf32 + gr * dim + c8 * 8);
            dp[0] = make_float4(v[0], v[1], v[2], v[3]);
            dp[1] = make_float4(v[4], v[5], v[6], v[7]);
        }
        u32x4 pk;
        pk.x = f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16);
        pk.y = f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16);
        pk.z = f32_to_bf16_bits(v[4]) | (f32_to_bf16_bits(v[5]) << 16);
        pk.w = f32_to_bf16_bits(v[6]) | (f32_to_bf16_bits(v[7]) << 16);
        xt[tiled_index(gr >> 5, ksteps, c8 >> 1, c8 & 1, (int)(gr & 31))] = pk;
    }
}

// alive[tile] |= bits of rows [first, first+n)
__global__ void k_set_alive(uint32_t *alive, int64_t first, int64_t n)
{
    int64_t t0 = first >> 5, t1 = (first + n - 1) >> 5;
    int64_t t = t0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > t1) return;
    int lo = (t == t0) ? (int)(first & 31) : 0;
    int hi = (t == t1) ? (int)((first + n - 1) & 31) : 31;
    uint32_t m = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
    alive[t] |= m;
}

// codes_in [n][ncols] -> codes [ncols][cap_rows] at first_row
__global__ void k_store_codes(const int32_t *__restrict__ in, int64_t n, int ncols, int64_t first_row,
                              int64_t cap_rows, int32_t *__restrict__ codes)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * ncols) return;
    int64_t r = i / ncols;
    int c = (int)(i % ncols);
    codes[(int64_t)c * cap_rows + first_row + r] = in[i];
}

__global__ void k_tombstone(uint32_t *alive, const int64_t *__restrict__ rows, int64_t n, int64_t count,
                            unsigned int *cleared)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t r = rows[i];
    if (r < 0 || r >= count) return;
    uint32_t bit = 1u << (r & 31);
    uint32_t old = atomicAnd(&alive[r >> 5], ~bit);
    if (old & bit) atomicAdd(cleared, 1u);
}

// effective row mask of a filtered search: alive AND every (col == code).  One thread per row.
struct FilterSet {
    int n;
    int col[CRH_MAX_FILTERS];
    int code[CRH_MAX_FILTERS];
};
__global__ __launch_bounds__(256) void k_filter_mask(const uint32_t *__restrict__ alive,
                                                     const int32_t *__restrict__ codes, int64_t cap_rows,
                                                     int64_t count, FilterSet fs, uint32_t *__restrict__ out)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = r < count;
    if (ok) {
        for (int f = 0; f < fs.n; ++f) ok = ok && (codes[(int64_t)fs.col[f] * cap_rows + r] == fs.code[f]);
    }
    unsigned long long b = __ballot(ok);
    int lane = lane_id();
    int64_t tile = r >> 5;
    if ((lane & 31) == 0 && (tile << 5) < ((count + 31) & ~31LL)) {
        uint32_t m = (uint32_t)(lane ? (b >> 32) : b);
        out[tile] = m & alive[tile];
    }
}

// delete-by-filter: clear every alive bit the (already alive-ANDed) filter mask has set; counts the cleared rows
__global__ void k_tombstone_mask(uint32_t *__restrict__ alive, const uint32_t *__restrict__ mask, int64_t ntiles,
                                 unsigned int *cleared)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const uint32_t m = mask[t] & alive[t];
    if (m) {
        alive[t] &= ~m;
        atomicAdd(cleared, (unsigned int)__popc(m));
    }
}

// untile rows [first, first+n) into f32 [n][dim] (the bf16 copy; exact bf16 values)
__global__ void k_untile_rows(const u32x4 *__restrict__ xt, int ksteps, int64_t first, int64_t n, int dim,
                              float *__restrict__ out)
{
    int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int chunks = dim >> 3;
    if (item >= n * chunks) return;
    int64_t i = item / chunks;
    int c8 = (int)(item % chunks);
    int64_t gr = first + i;
    u32x4 pk = xt[tiled_index(gr >> 5, ksteps, c8 >> 1, c8 & 1, (int)(gr & 31))];
    float *o = out + i * dim + c8 * 8;
    o[0] = bf16_bits_f32(pk.x & 0xffffu);
    o[1] = bf16_bits_f32(pk.x >> 16);
    o[2] = bf16_bits_f32(pk.y & 0xffffu);
    o[3] = bf16_bits_f32(pk.y >> 16);
    o[4] = bf16_bits_f32(pk.z & 0xffffu);
    o[5] = bf16_bits_f32(pk.z >> 16);
    o[6] = bf16_bits_f32(pk.w & 0xffffu);
    o[7] = bf16_bits_f32(pk.w >> 16);
}

// ------------------------------------------------------------------ compaction (crh_index_compact)

// Stable compaction of the rows: tile_prefix[t] = alive rows in the tiles before t (host prefix sum of the popcounts).
// One thread per OLD row: its new number is tile_prefix + the alive bits below it in its own tile.
__global__ __launch_bounds__(256) void k_compact_map(const uint32_t *__restrict__ alive, const int64_t *__restrict__ tile_prefix, int64_t count,
                                                     int64_t *__restrict__ old_to_new, int32_t *__restrict__ new_to_old)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= count) return;
    const uint32_t w = alive[r >> 5];
    const int b = (int)(r & 31);
    int64_t n = -1;
    if ((w >> b) & 1u) {
        n = tile_prefix[r >> 5] + __popc(w & ((1u << b) - 1u));
        new_to_old[n] = (int32_t)r;
    }
    old_to_new[r] = n;
}

// New tiles [t0, t0 + nt) of the tiled bf16 image into a bounce buffer: one workgroup per new tile, 64 lanes x ksteps pieces;
// lane (h, r) of piece s copies the 16 bytes of old row new_to_old[32 t + r] (zeros past the new row count).
__global__ __launch_bounds__(256) void k_compact_tiles(const u32x4 *__restrict__ xt, int ksteps, const int32_t *__restrict__ new_to_old,
                                                       int64_t new_count, int64_t t0, u32x4 *__restrict__ bounce)
{
    const int64_t t = t0 + blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = lane >> 5, r = lane & 31;
    const int64_t nr = t * 32 + r;
    const int64_t old = nr < new_count ? (int64_t)new_to_old[nr] : -1;
    for (int s = wave; s < ksteps; s += 4) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (old >= 0) v = xt[tiled_index(old >> 5, ksteps, s, h, (int)(old & 31))];
        bounce[(size_t)blockIdx.x * ksteps * 64 + piece_off(s) + piece_slot(h, r)] = v;
    }
}

// rows [r0, r0 + n) of a row-major array of `row16` 16-byte pieces per row (the f32 master) into a bounce buffer
__global__ __launch_bounds__(256) void k_compact_rows16(const u32x4 *__restrict__ src, int row16, const int32_t *__restrict__ new_to_old,
                                                        int64_t new_count, int64_t r0, int64_t n, u32x4 *__restrict__ bounce)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * row16) return;
    const int64_t lr = i / row16;
    const int c = (int)(i - lr * row16);
    const int64_t nr = r0 + lr;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (nr < new_count) v = src[(size_t)new_to_old[nr] * row16 + c];
    bounce[i] = v;
}

// one int32 column, rows [r0, r0 + n): `fill` past the new row count
__global__ __launch_bounds__(256) void k_compact_i32(const int32_t *__restrict__ col, const int32_t *__restrict__ new_to_old, int64_t new_count,
                                                     int64_t r0, int64_t n, int32_t fill, int32_t *__restrict__ bounce)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t nr = r0 + i;
    bounce[i] = nr < new_count ? col[new_to_old[nr]] : fill;
}

// alive words after compaction: rows [0, new_count) alive, nothing beyond
__global__ void k_compact_alive(uint32_t *__restrict__ alive, int64_t new_count, int64_t ntiles)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const int64_t left = new_count - t * 32;
    alive[t] = left >= 32 ? 0xffffffffu : (left <= 0 ? 0u : ((1u << left) - 1u));
}

// ------------------------------------------------------------------ query preparation

// One block (64 threads) per query slot 0..63.  Slots >= nq become all-zero queries.
// qn:    [64][dim] f32 -- the canonical query the rescoring uses: cosine_preprocess(q), and for a
//        bf16 store additionally rounded to bf16 (what the oracle scores with).
// qfrag: [2][ksteps][64] u32x4 -- bf16 B-operand pieces for the scan.
__device__ __forceinline__ void prep_query_i8(const float *src, int dim, int qi, u32x4 *__restrict__ qfrag8, float *__restrict__ qpar);   // crh_i8.hpp

template <bool ROUND_BF16, bool I8 = false>
__global__ __launch_bounds__(64) void k_prep_queries(const float *__restrict__ q, int nq, int dim, int ksteps,
                                                     float *__restrict__ qn, u32x4 *__restrict__ qfrag, SearchStatus *__restrict__ status,
                                                     u32x4 *__restrict__ qfrag8 = nullptr, float *__restrict__ qpar = nullptr)
{
    __shared__ __attribute__((aligned(16))) float qs[2048];
    __shared__ float dv_s;
    const int qi = blockIdx.x, tid = threadIdx.x;
    if (qi == 0) {   // the batch's status slot starts from zero (first kernel of the batch: saves a memset node per search)
        uint32_t *w = reinterpret_cast<uint32_t *>(status);
        for (int i = tid; i < (int)(sizeof(SearchStatus) / 4); i += 64) w[i] = 0u;
    }
    const bool real = qi < nq;
    const float *src = q + (int64_t)qi * dim;
    for (int i = tid; i < dim; i += 64) qs[i] = real ? src[i] : 0.0f;
    __syncthreads();
    if (tid == 0) {
        float acc = 0.0f;  // index order, product and sum rounded separately (oracle: orc_cosine_preprocess)
        const float4 *q4 = reinterpret_cast<const float4 *>(qs);   // 16-byte LDS reads, four deep: the chain of additions is what is left
#pragma unroll 4
        for (int i = 0; i < (dim >> 2); ++i) {
            const float4 v = q4[i];
            float p;
            p = v.x * v.x;
            acc = acc + p;
            p = v.y * v.y;
            acc = acc + p;
            p = v.z * v.z;
            acc = acc + p;
            p = v.w * v.w;
            acc = acc + p;
        }
        dv_s = cosine_divisor(acc);
    }
    __syncthreads();
    const float dv = dv_s;
    const int qb = qi >> 5, c = qi & 31;
    for (int c8 = tid; c8 < (dim >> 3); c8 += 64) {
        float v[8];
        uint32_t hb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = qs[c8 * 8 + j];
            if (dv != 0.0f) x = __fdiv_rn(x, dv);
            hb[j] = f32_to_bf16_bits(x);
            v[j] = ROUND_BF16 ? bf16_bits_f32(hb[j]) : x;
        }
        float4 *dst = reinterpret_cast<float4 *>(qn + (int64_t)qi * dim + c8 * 8);
        dst[0] = make_float4(v[0], v[1], v[2], v[3]);
        dst[1] = make_float4(v[4], v[5], v[6], v[7]);
        u32x4 pk;
        pk.x = hb[0] | (hb[1] << 16);
        pk.y = hb[2] | (hb[3] << 16);
        pk.z = hb[4] | (hb[5] << 16);
        pk.w = hb[6] | (hb[7] << 16);
        qfrag[((size_t)qb * ksteps + (c8 >> 1)) * 64 + ((c8 & 1) * 32 + c)] = pk;
        if (I8) {   // (each thread rewrites the elements it alone has read: the canonical query replaces the raw one in LDS)
#pragma unroll
            for (int j = 0; j < 8; ++j) qs[c8 * 8 + j] = v[j];
        }
    }
    if (I8) {       // the 15-bit integer images of the canonical query for the scan over the int8 copy (crh_i8.hpp)
        __syncthreads();
        prep_query_i8(qs, dim, qi, qfrag8, qpar);
    }
}

// ------------------------------------------------------------------ the scan

// Streaming load of one 1-KiB corpus piece with the `nt` cache policy.  __builtin_nontemporal_load keeps the flag on only a
// fraction of the unrolled loads (those that get an immediate offset lose it), so the instruction is written out; the
// compiler does not know an asm load is asynchronous, hence nt_wait(): a counted vmcnt that takes the destination as an
// in/out operand, which makes every consumer (the MFMA) depend on the wait.
__device__ __forceinline__ void nt_load(u32x4 &dst, const u32x4 *p)
{
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void nt_wait(u32x4 &dst)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(dst) : "n"(N));
}

// MODE 0 (seed):  items are sample tiles; writes gmax[item][q] = max over the tile's valid rows.
// MODE 1 (main):  items are all tiles; rows with score >= tau[q] become candidates.
// MODE 2 (probe): the main scan's loads only (crh_debug_read_ceiling): the HBM read rate this access pattern can reach.
//
// One workgroup per CU (the 96 KB query image pins that), WAVES waves each streaming its own
// tiles: item i of wave g is i = g + j*total_waves.  Per k-step a wave issues one 1-KiB
// nontemporal load (RING of them in flight, running ahead across tile boundaries), reads two
// 1-KiB query pieces from LDS and issues two 32x32x16 MFMAs (queries 0-31 and 32-63 against the
// same 32 corpus rows).  C layout: column = lane&31 = query, rows in the 16 accumulators, so the
// per-query threshold is one VGPR per lane and the common case costs 32 v_max + 2 compares a tile.
// QB = number of 32-query MFMA column blocks per pass: 2 (64 queries) while the query image fits LDS beside the kernel's
// other needs (dim <= 1024), 1 (32 queries) for dim 1536.
template <int KSTEPS, int MODE, int WAVES, int RING, int QB = 2>
__global__ __launch_bounds__(WAVES * 64) void k_scan(
    const u32x4 *__restrict__ xt, const u32x4 *__restrict__ qfrag, const float *__restrict__ tau,
    const uint32_t *__restrict__ rowmask, int nitems, int tile_stride, float *__restrict__ gmax,
    u32x4 *__restrict__ wave_lists, int wave_cap, unsigned int *__restrict__ qcount,
    u32x2 *__restrict__ qlist, int qcap, SearchStatus *__restrict__ status)
{
    static_assert(KSTEPS % RING == 0, "ring must divide the k-steps of a tile");
    __shared__ u32x4 qs[QB * KSTEPS * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;

    // (Preparing the queries HERE, in the seed scan -- every workgroup building the image in its own LDS, workgroup 0 writing it
    // out for the later kernels -- was built and measured in round 3: it removes k_prep_queries' launch and adds the same time to
    // this kernel, 119.7 us against 119.3 us around the main scan.  Not shipped.)
    for (int i = tid; i < QB * KSTEPS * 64; i += WAVES * 64) qs[i] = qfrag[i];
    float t0 = 0.f, t1 = 0.f;
    if (MODE == 1) {
        t0 = tau[lane & 31];
        t1 = QB == 2 ? tau[32 + (lane & 31)] : INFINITY;
    }
    __syncthreads();

    const int total = gridDim.x * WAVES;
    const int gw = blockIdx.x * WAVES + wave;
    u32x4 *mylist = wave_lists + (size_t)gw * wave_cap;
    unsigned int wcnt = 0;

    int i = gw;
    const int lslot = lane_slot(lane);   // (the lane's 16 bytes inside a piece)
    const u32x4 *xp = xt + (size_t)(i < nitems ? (int64_t)i * tile_stride : 0) * (KSTEPS * 64) + lslot;
    u32x4 ring[RING];
    if (i < nitems) {
#pragma unroll
        for (int d = 0; d < RING; ++d) nt_load(ring[d], xp + piece_off(d));
    }
    while (i < nitems) {
        const int inext = i + total;
        const int64_t tile = (int64_t)i * tile_stride;
        const u32x4 *xn = (inext < nitems) ? xt + (size_t)((int64_t)inext * tile_stride) * (KSTEPS * 64) + lslot : xp;
        const uint32_t vmask = rowmask[tile];  // wave-uniform -> scalar load
        // the query image is loop-invariant: without this the compiler hoists all 96 LDS pieces (384 VGPRs)
        // out of the tile loop and spills; the clobber makes it re-read qs per tile, as intended
        asm volatile("" ::: "memory");

        f32x16 a0 = {0}, a1 = {0};
        u32x4 b0 = qs[lane], b1 = qs[(QB - 1) * KSTEPS * 64 + lane];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            // order pinned by the sched_barrier: next step's query pieces (LDS), this step's two MFMAs, then the
            // load that refills this ring slot RING steps ahead (it may belong to the wave's next tile).
            const int s1 = (s + 1 < KSTEPS) ? s + 1 : s;
            const u32x4 nb0 = qs[s1 * 64 + lane];
            const u32x4 nb1 = qs[((QB - 1) * KSTEPS + s1) * 64 + lane];
            nt_wait<RING - 1>(ring[s % RING]);   // the oldest of the RING loads in flight has landed (issue order)
            const bf16x8 xa = __builtin_bit_cast(bf16x8, ring[s % RING]);
            if (MODE == 2) {   // read-ceiling probe: the loads alone, kept alive by an empty asm
                asm volatile("" ::"v"(ring[s % RING]));
            } else {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, __builtin_bit_cast(bf16x8, b0), a0, 0, 0, 0);
                if (QB == 2) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, __builtin_bit_cast(bf16x8, b1), a1, 0, 0, 0);
            }
            const int sp = s + RING;
            const u32x4 *src = (sp < KSTEPS) ? xp + piece_off(sp) : xn + piece_off(sp - KSTEPS);
            nt_load(ring[s % RING], src);
            b0 = nb0;
            b1 = nb1;
            __builtin_amdgcn_sched_barrier(0);
        }

        if (MODE == 2) {
            // nothing: the probe measures what the same access pattern reads with no arithmetic and no candidate logic
        } else if (MODE == 0) {
            float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const bool ok = (vmask >> row) & 1u;
                m0 = fmaxf(m0, ok ? a0[r] : -INFINITY);
                m1 = fmaxf(m1, ok ? a1[r] : -INFINITY);
            }
            m0 = fmaxf(m0, __shfl_xor(m0, 32));
            m1 = fmaxf(m1, __shfl_xor(m1, 32));
            if (h == 0) {
                gmax[(size_t)i * 64 + lane] = m0;
                if (QB == 2) gmax[(size_t)i * 64 + 32 + lane] = m1;
            }
        } else {
            float m0 = a0[0], m1 = a1[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) {
                m0 = fmaxf(m0, a0[r]);
                m1 = fmaxf(m1, a1[r]);
            }
            const bool any = (m0 >= t0) || (m1 >= t1);
            if (__ballot(any) != 0ull && vmask != 0u) {
                const uint32_t rowbase = (uint32_t)(tile * 32);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const float tq = qb ? t1 : t0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float sc = qb ? a1[r] : a0[r];
                        const bool pass = ((vmask >> row) & 1u) && (sc >= tq);
                        const unsigned long long pm = __ballot(pass);
                        if (pm != 0ull) {
                            const unsigned int pre = __builtin_amdgcn_mbcnt_hi(
                                (unsigned int)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)pm, 0u));
                            const unsigned int pos = wcnt + pre;
                            if (pass && pos < (unsigned int)wave_cap) {
                                u32x4 e;
                                e.x = f32_bits(sc);
                                e.y = rowbase + row;
                                e.z = (uint32_t)(qb * 32 + (lane & 31));
                                e.w = 0u;
                                mylist[pos] = e;
                            }
                            wcnt += (unsigned int)__popcll(pm);
                        }
                    }
                }
            }
        }
        xp = xn;
        i = inext;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the run-ahead loads of the last tile are still in flight

    if (MODE == 1) {
        // Hand the workgroup's candidates over to the per-query lists: count per query in LDS, reserve
        // one contiguous range per (workgroup, query) with 64 global atomics, scatter.
        __syncthreads();  // every wave is done with qs (and its list stores have completed)
        unsigned int *wc = reinterpret_cast<unsigned int *>(qs);  // [WAVES] counts, [64] hist, [64] base, [64] off
        unsigned int *hist = wc + WAVES;
        unsigned int *base = hist + 64;
        unsigned int *off = base + 64;
        if (lane == 0) {
            wc[wave] = wcnt < (unsigned int)wave_cap ? wcnt : (unsigned int)wave_cap;
            atomicMax(&status->max_wave_cnt, wcnt);
            if (wcnt > (unsigned int)wave_cap) atomicAdd(&status->wave_overflow, 1u);
        }
        if (tid < 64) {
            hist[tid] = 0u;
            off[tid] = 0u;
        }
        __syncthreads();
        const u32x4 *wl = wave_lists + (size_t)blockIdx.x * WAVES * wave_cap;
        for (int w = 0; w < WAVES; ++w) {
            const unsigned int n = wc[w];
            for (unsigned int e = tid; e < n; e += WAVES * 64) atomicAdd(&hist[wl[(size_t)w * wave_cap + e].z & 63u], 1u);
        }
        __syncthreads();
        if (tid < 64) base[tid] = hist[tid] ? atomicAdd(&qcount[tid], hist[tid]) : 0u;
        __syncthreads();
        for (int w = 0; w < WAVES; ++w) {
            const unsigned int n = wc[w];
            for (unsigned int e = tid; e < n; e += WAVES * 64) {
                const u32x4 c = wl[(size_t)w * wave_cap + e];
                const unsigned int q = c.z & 63u;
                const unsigned int idx = base[q] + atomicAdd(&off[q], 1u);
                if (idx < (unsigned int)qcap) {
                    u32x2 o;
                    o.x = c.x;
                    o.y = c.y;
                    qlist[(size_t)q * qcap + idx] = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ the wide scan (65 .. 256 queries per corpus pass)

// k_scan keeps the QUERIES in LDS and streams the corpus through registers: 64 queries per pass is what 96 KB of LDS holds,
// and a call with more queries pays one corpus pass per 64.  k_scan_wide turns the operands around: each of a workgroup's
// 8 waves keeps ONE 32-query block as MFMA B fragments in its own registers for the whole kernel (KSTEPS x 4 VGPRs: 192 at
// dim 768 -- two waves per SIMD at 256 registers each), and the CORPUS tiles go through LDS: a ring of SLOTS tiles filled by
// LDS-DMA (global_load_lds_dwordx4, nt; every wave issues KSTEPS/8 of a tile's 1-KiB pieces), read back by all 8 waves with
// one conflict-free ds_read_b128 per MFMA.  One corpus pass then serves 256 queries: HBM bytes per query / 4.  The pass is
// MFMA-bound, not HBM-bound: 3.9 TFLOP per 10M rows run at ~0.96 PFLOP/s (4.0 ms; the ring's data path alone, with the waits
// and barriers but no arithmetic, streams 7.0 TB/s = 2.2 ms).  Variants measured and dropped (profiles/r02_wide_scan.md): 4
// waves holding two blocks each with a 6-deep LDS read-ahead (4.5 ms: one wave per SIMD leaves nobody to hide the DMA issue
// and the epilogue behind), the ring refilled in 8/12-KB chunks (112-120 KB in flight instead of 96: no gain once MFMA-bound),
// two accumulation chains per wave with the last query k-steps parked in LDS to make room (4.1 ms).
// Per tile: ONE raw s_barrier -- each wave first waits (counted vmcnt) for its own pieces of tile j, so after the barrier
// tile j has landed for everybody AND everybody is done reading tile j-1, whose slot the DMA of tile j+SLOTS-1 is then
// issued into.  Thresholds, candidate compaction and the hand-over to the per-query lists are k_scan's (one 32-query
// block per wave instead of two).  nblk = number of 32-query blocks in use: waves beyond it only move data.
template <int N>
__device__ __forceinline__ void vm_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int KSTEPS, int MODE>
__global__ __launch_bounds__(512) void k_scan_wide(
    const u32x4 *__restrict__ xt, const u32x4 *__restrict__ qfrag, const float *__restrict__ tau,
    const uint32_t *__restrict__ rowmask, int nitems, int tile_stride, int nblk, float *__restrict__ gmax, int qstride,
    u32x4 *__restrict__ wave_lists, int lists_per_block, int wave_cap, unsigned int *__restrict__ qcount,
    u32x2 *__restrict__ qlist, int qcap, SearchStatus *__restrict__ status)
{
    constexpr int WAVES = 8;
    constexpr int PPW = KSTEPS / WAVES;                   // 1-KiB pieces of a tile issued by each wave
    constexpr int SLOTS = (KSTEPS * 3 <= 144) ? 3 + (144 - KSTEPS * 3) / KSTEPS : 3;   // 144 KB of LDS: 3 tiles at dim 768, 6 at 384
    static_assert(KSTEPS % WAVES == 0 && SLOTS >= 3 && SLOTS * KSTEPS <= 144, "ring geometry");
    __shared__ u32x4 ring[SLOTS * KSTEPS * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const bool active = wave < nblk;

    u32x4 qreg[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) qreg[s] = active ? qfrag[((size_t)wave * KSTEPS + s) * 64 + lane] : u32x4{0u, 0u, 0u, 0u};
    float t0 = 0.f;
    if (MODE == 1) t0 = active ? tau[wave * 32 + (lane & 31)] : INFINITY;

    const int nmine = (int)blockIdx.x < nitems ? (nitems - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    u32x4 *mylist = wave_lists + ((size_t)blockIdx.x * lists_per_block + wave) * wave_cap;
    unsigned int wcnt = 0;

    auto issue = [&](int j) {
        const int64_t tile = (int64_t)((int)blockIdx.x + j * (int)gridDim.x) * tile_stride;
        const u32x4 *src = xt + (size_t)tile * KSTEPS * 64 + lane_slot(lane);   // (the LDS image stays lane-linear)
        u32x4 *dst = ring + ((size_t)(j % SLOTS) * KSTEPS + wave * PPW) * 64;
#pragma unroll
        for (int p = 0; p < PPW; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece_off(wave * PPW + p)),
                                             (__attribute__((address_space(3))) void *)(dst + p * 64), 16, 0, 2 /* nt */);
    };
    for (int j = 0; j < SLOTS - 1 && j < nmine; ++j) issue(j);

    for (int j = 0; j < nmine; ++j) {
        // my pieces of tile j have landed once at most min(SLOTS-2, tiles after j) later tiles of mine are still in flight
        const int ahead = (nmine - 1 - j) < (SLOTS - 2) ? (nmine - 1 - j) : (SLOTS - 2);
        if (ahead >= 4) vm_wait<4 * PPW>();
        else if (ahead == 3) vm_wait<3 * PPW>();
        else if (ahead == 2) vm_wait<2 * PPW>();
        else if (ahead == 1) vm_wait<PPW>();
        else vm_wait<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (j + SLOTS - 1 < nmine) issue(j + SLOTS - 1);   // into the slot tile j-1 was read from: free since the barrier

        if (!active) continue;
        const int item = (int)blockIdx.x + j * (int)gridDim.x;
        const int64_t tile = (int64_t)item * tile_stride;
        const uint32_t vmask = rowmask[tile];              // wave-uniform -> scalar load
        const u32x4 *lp = ring + (size_t)(j % SLOTS) * KSTEPS * 64 + lane;
        // 192 of the wave's 256 registers hold its queries: the corpus fragments get a 4-deep rotation (three ds_read_b128 in
        // flight ahead of the MFMA that consumes the fourth), enough to cover the LDS latency behind 32-cycle MFMAs
        f32x16 a0 = {0};
        u32x4 af[4];
        af[0] = lp[0];
        af[1] = lp[64];
        af[2] = lp[128];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            if (s + 3 < KSTEPS) af[(s + 3) & 3] = lp[(s + 3) * 64];
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[s & 3]), __builtin_bit_cast(bf16x8, qreg[s]), a0, 0, 0, 0);
        }

        if (MODE == 0) {
            float m0 = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                m0 = fmaxf(m0, ((vmask >> row) & 1u) ? a0[r] : -INFINITY);
            }
            m0 = fmaxf(m0, __shfl_xor(m0, 32));
            if (h == 0) gmax[(size_t)item * qstride + wave * 32 + lane] = m0;
        } else {
            float m0 = a0[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m0 = fmaxf(m0, a0[r]);
            if (__ballot(m0 >= t0) != 0ull && vmask != 0u) {
                uint32_t rowbase = (uint32_t)(tile * 32) + 4u * (uint32_t)h;
                asm volatile("" : "+v"(rowbase));   // keep the 16 row numbers out of the loop-invariant registers (rare path)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2);
                    const int rbit = row + 4 * h;
                    const float sc = a0[r];
                    const bool pass = ((vmask >> rbit) & 1u) && (sc >= t0);
                    const unsigned long long pm = __ballot(pass);
                    if (pm != 0ull) {
                        const unsigned int pre = __builtin_amdgcn_mbcnt_hi((unsigned int)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)pm, 0u));
                        const unsigned int pos = wcnt + pre;
                        if (pass && pos < (unsigned int)wave_cap) {
                            u32x4 e;
                            e.x = f32_bits(sc);
                            e.y = rowbase + row;
                            e.z = (uint32_t)(wave * 32 + (lane & 31));
                            e.w = 0u;
                            mylist[pos] = e;
                        }
                        wcnt += (unsigned int)__popcll(pm);
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if (MODE == 1) {
        // hand the workgroup's candidates over to the per-query lists (as k_scan does, 256 bins)
        __syncthreads();   // every wave is done with the ring, and its list stores have completed
        unsigned int *wc = reinterpret_cast<unsigned int *>(ring);   // [WAVES] counts, [256] hist, [256] base, [256] off
        unsigned int *hist = wc + WAVES;
        unsigned int *base = hist + kWideQ;
        unsigned int *off = base + kWideQ;
        if (lane == 0) {
            wc[wave] = wcnt < (unsigned int)wave_cap ? wcnt : (unsigned int)wave_cap;
            atomicMax(&status->max_wave_cnt, wcnt);
            if (wcnt > (unsigned int)wave_cap) atomicAdd(&status->wave_overflow, 1u);
        }
        if (tid < kWideQ) {
            hist[tid] = 0u;
            off[tid] = 0u;
        }
        __syncthreads();
        const u32x4 *wl = wave_lists + (size_t)blockIdx.x * lists_per_block * wave_cap;
        for (int w = 0; w < WAVES; ++w) {
            const unsigned int n = wc[w];
            for (unsigned int e = tid; e < n; e += WAVES * 64) atomicAdd(&hist[wl[(size_t)w * wave_cap + e].z & (kWideQ - 1)], 1u);
        }
        __syncthreads();
        if (tid < kWideQ) base[tid] = hist[tid] ? atomicAdd(&qcount[tid], hist[tid]) : 0u;
        __syncthreads();
        for (int w = 0; w < WAVES; ++w) {
            const unsigned int n = wc[w];
            for (unsigned int e = tid; e < n; e += WAVES * 64) {
                const u32x4 c = wl[(size_t)w * wave_cap + e];
                const unsigned int q = c.z & (kWideQ - 1);
                const unsigned int idx = base[q] + atomicAdd(&off[q], 1u);
                if (idx < (unsigned int)qcap) {
                    u32x2 o;
                    o.x = c.x;
                    o.y = c.y;
                    qlist[(size_t)q * qcap + idx] = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ workgroup-wide selection helpers

// k-th largest (1-based) of n keys fetched by get(i), by MSB-first 8-bit radix passes.
// All threads of the block call it; returns the key to every thread.  Requires 1 <= k <= n.
// hist: NT/64 sub-histograms of 256 bins (one per wave: scores of one query share their top bits, so a single
// histogram would serialise on one or two bins); equal consecutive digits are added as one atomic.
template <typename KeyT, int NT, typename Get>
__device__ KeyT wg_kth_largest(Get get, unsigned int n, unsigned int k, unsigned int *hist /*(NT/64)*256*/, unsigned int *bcast /*2*/)
{
    constexpr int BITS = sizeof(KeyT) * 8;
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x;
    unsigned int *myhist = hist + (tid >> 6) * 256;
    KeyT prefix = 0;
    unsigned int kk = k;
    for (int shift = BITS - 8; shift >= 0; shift -= 8) {
        for (int i = tid; i < NW * 256; i += NT) hist[i] = 0u;
        __syncthreads();
        unsigned int run_digit = 0xffffffffu, run_len = 0u;
        for (unsigned int i = tid; i < n; i += NT) {
            const KeyT key = get(i);
            const bool match = (shift == BITS - 8) ? true : ((key >> (shift + 8)) == prefix);
            if (match) {
                const unsigned int d = (unsigned int)(key >> shift) & 255u;
                if (d == run_digit) {
                    ++run_len;
                } else {
                    if (run_len) atomicAdd(&myhist[run_digit], run_len);
                    run_digit = d;
                    run_len = 1u;
                }
            }
        }
        if (run_len) atomicAdd(&myhist[run_digit], run_len);
        __syncthreads();
        unsigned int tot = 0u;  // fold the per-wave histograms into hist[0..255]
        if (tid < 256) {
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += hist[w * 256 + tid];
        }
        __syncthreads();
        if (tid < 256) hist[tid] = tot;
        __syncthreads();
        if (tid < 64) {
            // lane t owns bins 255-4t .. 252-4t (descending); suffix-scan across lanes
            const int b0 = 255 - 4 * tid;
            const unsigned int c0 = hist[b0], c1 = hist[b0 - 1], c2 = hist[b0 - 2], c3 = hist[b0 - 3];
            const unsigned int mine = c0 + c1 + c2 + c3;
            unsigned int incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned int o = __shfl_up(incl, d);
                if (tid >= d) incl += o;
            }
            const unsigned int excl = incl - mine;
            if (excl < kk && kk <= incl) {
                unsigned int rem = kk - excl;
                int digit;
                if (rem <= c0) {
                    digit = b0;
                } else if (rem <= c0 + c1) {
                    digit = b0 - 1;
                    rem -= c0;
                } else if (rem <= c0 + c1 + c2) {
                    digit = b0 - 2;
                    rem -= c0 + c1;
                } else {
                    digit = b0 - 3;
                    rem -= c0 + c1 + c2;
                }
                bcast[0] = (unsigned int)digit;
                bcast[1] = rem;
            }
        }
        __syncthreads();
        prefix = (prefix << 8) | (KeyT)bcast[0];
        kk = bcast[1];
        __syncthreads();
    }
    return prefix;
}

// f(key, i) for every i = tid, tid + NT, ... < n, the keys fetched EIGHT at a time: when get() reads global memory (tens of
// thousands of candidates per query behind the int8 scan) a sweep is otherwise a chain of dependent round trips, n / NT deep.
template <int NT, typename Get, typename F>
__device__ __forceinline__ void for_each_key(Get get, unsigned int n, F f)
{
    for (unsigned int base = threadIdx.x; base < n; base += NT * 8) {
        uint32_t key[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned int i = base + j * NT;
            key[j] = i < n ? get(i) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned int i = base + j * NT;
            if (i < n) f(key[j], i);
        }
    }
}

// k-th largest (1-based) of n u32 keys in TWO steps instead of four radix passes.  The keys of one query -- candidate scores
// above a common threshold, or tile maxima of one query -- sit in a narrow band of the u32 range, so an order-preserving LINEAR
// bucketing of [lo, hi] into NB buckets (lo / hi: smallest / largest key above `floor_key`, which takes the -inf padding out of
// the range) isolates the k-th key's bucket with ONE histogram pass; that bucket's members (n / NB on average) are then ranked
// by counting.  Exact: the bucket function is monotone in the key, and the same function files and retrieves.  When the bucket
// holds more than CAP keys (masses of equal scores) the radix passes of wg_kth_largest take over.
// scratch: NB + CAP + 2 * (NT / 64) + 8 words of LDS.  All threads call; the key is returned to every thread.
template <int NT, int NB, int CAP, typename Get>
__device__ uint32_t wg_kth_largest_fast(Get get, unsigned int n, unsigned int k, uint32_t floor_key, unsigned int *scratch,
                                        unsigned int *radix_hist /*(NT/64)*256*/, unsigned int *bcast /*2*/)
{
    constexpr int NW = NT / 64;
    static_assert(NB % 64 == 0 && NB <= NT, "one thread per bucket in the scan");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned int *hist = scratch, *list = scratch + NB, *wlo = list + CAP, *whi = wlo + NW, *ctl = whi + NW;   // ctl[0..5]
    // 1. range of the keys above the floor, and how many sit at or below it
    uint32_t lo = 0xffffffffu, hi = 0u;
    unsigned int below = 0u;
    for_each_key<NT>(get, n, [&](uint32_t key, unsigned int) {
        if (key > floor_key) {
            lo = key < lo ? key : lo;
            hi = key > hi ? key : hi;
        } else {
            ++below;
        }
    });
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t ol = __shfl_xor(lo, d), oh = __shfl_xor(hi, d);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
        below += __shfl_xor(below, d);
    }
    for (int i = tid; i < NB; i += NT) hist[i] = 0u;
    if (tid < 6) ctl[tid] = 0u;
    __syncthreads();
    if (lane == 0) {
        wlo[wave] = lo;
        whi[wave] = hi;
        atomicAdd(&ctl[0], below);
    }
    __syncthreads();
    lo = 0xffffffffu;
    hi = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        lo = wlo[w] < lo ? wlo[w] : lo;
        hi = whi[w] > hi ? whi[w] : hi;
    }
    const unsigned int n_above = n - ctl[0];
    if (k > n_above) {   // the k-th largest is one of the floor keys: every one of them compares equal for the callers (-inf)
        return floor_key;
    }
    if (lo == hi) return lo;
    // bucket: monotone non-decreasing in the key (u32 -> f32 conversion, a positive scale and truncation are all monotone)
    const float scale = (float)NB / ((float)(hi - lo) + 1.0f);
    auto bucket = [&](uint32_t key) {
        const unsigned int b = (unsigned int)((float)(key - lo) * scale);
        return b < (unsigned int)NB ? b : (unsigned int)NB - 1u;
    };
    // 2. one histogram pass
    for_each_key<NT>(get, n, [&](uint32_t key, unsigned int) {
        if (key > floor_key) atomicAdd(&hist[bucket(key)], 1u);
    });
    __syncthreads();
    // 3. which bucket holds the k-th: thread t owns bucket NB-1-t (descending), suffix sums by wave scan + wave totals
    unsigned int mine = tid < NB ? hist[NB - 1 - tid] : 0u, incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wlo[wave] = incl;      // (range words are dead: every thread holds lo / hi in registers)
    __syncthreads();
    unsigned int before = 0u;
    for (int w = 0; w < wave; ++w) before += wlo[w];
    const unsigned int excl = before + incl - mine;
    if (tid < NB && excl < k && k <= excl + mine) {
        ctl[1] = (unsigned int)(NB - 1 - tid);   // the bucket
        ctl[2] = k - excl;                       // rank of the wanted key inside it, from the top
        ctl[3] = mine;                           // its population
    }
    __syncthreads();
    const unsigned int bsel = ctl[1], rem = ctl[2], pop = ctl[3];
    if (pop > (unsigned int)CAP) {   // workgroup-uniform: masses of (nearly) equal keys -> the general path
        __syncthreads();
        return wg_kth_largest<uint32_t, NT>(get, n, k, radix_hist, bcast);
    }
    // 4. the bucket's members, then the rem-th largest of them by counting (equal keys allowed)
    for_each_key<NT>(get, n, [&](uint32_t key, unsigned int) {
        if (key > floor_key && bucket(key) == bsel) list[atomicAdd(&ctl[4], 1u)] = key;
    });
    __syncthreads();
    if ((unsigned int)tid < pop) {
        const uint32_t key = list[tid];
        unsigned int gt = 0u, ge = 0u;
        for (unsigned int j = 0; j < pop; ++j) {
            const uint32_t o = list[j];
            gt += o > key ? 1u : 0u;
            ge += o >= key ? 1u : 0u;
        }
        if (gt < rem && rem <= ge) ctl[5] = key;   // (every thread holding the wanted VALUE writes the same word)
    }
    __syncthreads();
    return ctl[5];
}

// descending bitonic sort of P (power of two, <= 2*NT) u64 keys in LDS
template <int NT>
__device__ void wg_bitonic_desc(unsigned long long *keys, int P)
{
    const int tid = threadIdx.x;
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (P >> 1); t += NT) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------ seed scan + threshold + main scan in ONE launch

// One of the two grid-wide waits of k_scan_fused: every storing wave drains its stores, the workgroup meets, ONE lane publishes
// (agent-scope release, then the arrival) and polls the counter with relaxed loads until `expected` arrivals are in, acquires
// (agent scope: this CU's L1 is invalidated) and the workgroup meets again -- the placement-independent protocol of the CDNA
// guide (Guideline 16, counter form).  The spin is bounded in TIME by the job: `ticks` (100 MHz) is what the host allows this
// launch -- eight times what its pass should take, at least 2 ms (wait_ticks_for in crh_index.hip; round 3 allowed a flat 0.5 s,
// 400 passes).  A workgroup that does not become resident in that time (another stream's kernels holding its CU) ends the wait
// with status->bar_timeout set; the host then voids the batch, runs it again in the three-launch form, which needs no
// co-residency, and keeps the index on that form for a while (finish_pending).
__device__ __forceinline__ void grid_wait(unsigned int *counter, unsigned int expected, bool arrive, SearchStatus *status, unsigned int ticks)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (arrive) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the fence's own wait can be dropped by the compiler: keep this one)
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long t0 = wall_clock64();          // constant 100 MHz
        unsigned int spins = 0u;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 255u) == 0u && wall_clock64() - t0 > (unsigned long long)ticks) {
                __hip_atomic_store(&status->bar_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            // (somebody else's wait has given up: the batch is void, nobody needs to sit out the rest of the time)
            if ((spins & 255u) == 128u && __hip_atomic_load(&status->bar_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// The <= 64-query batch in three launches instead of five: k_prep_queries, THIS, k_select.  The seed scan, the threshold and the
// main scan used to be separate kernels; the sample tiles (201 MB at 10M rows) were read twice and two kernel boundaries sat on
// the critical path.  Here every wave's FIRST tile is its sample tile -- wave g of the grid takes tile g * S, a strided sample
// of G tiles (G = waves in the grid, 4096 on 256 CUs) -- and its accumulators stay in registers while
//   1. the tile maxima go to gmax[q][g], grid-wide wait A (everybody's maxima are visible);
//   2. workgroup q computes tau[q] = the k-th largest of query q's G maxima minus the margin (wg_kth_largest_fast: 1024 threads,
//      ~2 us), grid-wide wait B (arrivals: the tau workgroups only; everybody waits);
//   3. every wave filters its retained sample tile against tau and walks on through the tiles that are not samples, in the
//      interleaved order of k_scan (remaining tile r = wave + j * waves), with the same ring of loads and the same per-tile work.
// The corpus is read exactly once per batch.  Requires the whole grid resident (one workgroup per CU: the LDS footprint pins
// that, the grid is never larger than the CU count); the waits are bounded (grid_wait).  Thresholds, candidate compaction and
// the hand-over to the per-query lists are k_scan's.
// Phase clocks of a MEASUREMENT build (-DCRH_FUSED_STAMPS, tools/fused_stamps.py); in the product CRH_STAMP is nothing.
#ifdef CRH_FUSED_STAMPS
__device__ unsigned long long g_fused_stamps[256 * 8 + 256 * 16];
__device__ unsigned long long g_select_stamps[64 * 16];   // per query: stamps 0..7, then survivors of a part, candidates, survivors of the query, rows given the canonical chain
#define CRH_SEL_STAMP(n)                                                                          \
    do {                                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < 64) g_select_stamps[blockIdx.x * 16 + (n)] = wall_clock64(); \
    } while (0)
#define CRH_STAMP(n)                                                                                      \
    do {                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.x < 256) g_fused_stamps[blockIdx.x * 8 + (n)] = wall_clock64();  \
    } while (0)
// (sub-stamps of k_scan_i8's threshold phase, workgroups 0..63: they go to the per-wave slots 8..15 that its eight waves leave unused)
#define CRH_TAU_STAMP(n)                                                                                                    \
    do {                                                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 64) g_fused_stamps[256 * 8 + blockIdx.x * 16 + (n)] = wall_clock64();          \
    } while (0)
#else
#define CRH_STAMP(n) ((void)0)
#define CRH_SEL_STAMP(n) ((void)0)
#define CRH_TAU_STAMP(n) ((void)0)
#endif

template <int KSTEPS, int WAVES, int RING, int QB = 2>
__global__ __launch_bounds__(WAVES * 64) void k_scan_fused(
    const u32x4 *__restrict__ xt, const u32x4 *__restrict__ qfrag, const uint32_t *__restrict__ rowmask, int ntiles, int G, int S,
    float *__restrict__ gmax, float *__restrict__ tau_g, int k, float margin, int nq, u32x4 *__restrict__ wave_lists, int wave_cap,
    unsigned int *__restrict__ qcount, u32x2 *__restrict__ qlist, int qcap, SearchStatus *__restrict__ status, int wait_extra,
    unsigned int wait_ticks)
{
    // (wait_extra: 0; a test passes 1 -- wait A then expects an arrival that never comes, which exercises the time-out path)
    static_assert(KSTEPS % RING == 0, "ring must divide the k-steps of a tile");
    constexpr int NT = WAVES * 64, NQS = QB * 32;
    __shared__ u32x4 qs[QB * KSTEPS * 64];
    __shared__ uint32_t col[4096];                       // one query's tile maxima as ordered keys (G <= 4096)
    __shared__ unsigned int hist[(NT / 64) * 256];
    __shared__ unsigned int bcast[2];
    __shared__ unsigned int fast[1024 + 256 + 2 * (NT / 64) + 8];
    __shared__ float tau_s[NQS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;

    CRH_STAMP(0);
    for (int i = tid; i < QB * KSTEPS * 64; i += NT) qs[i] = qfrag[i];
    __syncthreads();
    CRH_STAMP(1);

    const int total = gridDim.x * WAVES;
    const int gw = blockIdx.x * WAVES + wave;
    u32x4 *mylist = wave_lists + (size_t)gw * wave_cap;
    unsigned int wcnt = 0;

    // the tiles that are not samples, r = 0 .. nrem-1 in ascending tile order: below G * S every S-th tile is a sample
    const int nrem = ntiles - G, GS1 = G * (S - 1), Sm1 = S > 1 ? S - 1 : 1;
    auto tile_of = [&](int r) -> int64_t { return r < GS1 ? (int64_t)r + r / Sm1 + 1 : (int64_t)r + G; };
    const int lslot = lane_slot(lane);
    auto tile_ptr = [&](int64_t tile) { return xt + (size_t)tile * (KSTEPS * 64) + lslot; };

    // One tile: KSTEPS x (wait for the oldest of the RING loads in flight, two MFMAs, refill the slot RING steps ahead -- past the
    // tile's end from the NEXT tile `xn`, so the stream never stops at a tile boundary).  k_scan's inner loop.
    u32x4 ring[RING];
    auto scan_tile = [&](const u32x4 *xp, const u32x4 *xn, f32x16 &c0, f32x16 &c1) {
        asm volatile("" ::: "memory");         // (keeps the query image out of the loop-invariant registers: see k_scan)
        u32x4 b0 = qs[lane], b1 = qs[(QB - 1) * KSTEPS * 64 + lane];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int s1 = (s + 1 < KSTEPS) ? s + 1 : s;
            const u32x4 nb0 = qs[s1 * 64 + lane];
            const u32x4 nb1 = qs[((QB - 1) * KSTEPS + s1) * 64 + lane];
            nt_wait<RING - 1>(ring[s % RING]);
            const bf16x8 xa = __builtin_bit_cast(bf16x8, ring[s % RING]);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, __builtin_bit_cast(bf16x8, b0), c0, 0, 0, 0);
            if (QB == 2) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, __builtin_bit_cast(bf16x8, b1), c1, 0, 0, 0);
            const int sp = s + RING;
            const u32x4 *src = (sp < KSTEPS) ? xp + piece_off(sp) : xn + piece_off(sp - KSTEPS);
            nt_load(ring[s % RING], src);
            b0 = nb0;
            b1 = nb1;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- 1. the sample tile of this wave; its loads already run ahead into the wave's first ordinary tile, whose first pieces
    // then sit in registers while the thresholds are worked out
    f32x16 a0 = {0}, a1 = {0};
    const bool has_sample = gw < G;
    const int64_t stile = has_sample ? (int64_t)gw * S : 0;
    int i = gw;                                                  // index of this wave's next ordinary tile
    const u32x4 *xp = tile_ptr(i < nrem ? tile_of(i) : 0);       // ... and its address
    if (has_sample) {
        const u32x4 *xs = tile_ptr(stile);
#pragma unroll
        for (int d = 0; d < RING; ++d) nt_load(ring[d], xs + piece_off(d));
        scan_tile(xs, xp, a0, a1);
        const uint32_t vmask = rowmask[stile];
        float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool ok = (vmask >> row) & 1u;
            m0 = fmaxf(m0, ok ? a0[r] : -INFINITY);
            m1 = fmaxf(m1, ok ? a1[r] : -INFINITY);
        }
        m0 = fmaxf(m0, __shfl_xor(m0, 32));
        m1 = fmaxf(m1, __shfl_xor(m1, 32));
        if (h == 0) {
            gmax[(size_t)lane * G + gw] = m0;                       // [query][sample]: the threshold workgroup reads a row
            if (QB == 2) gmax[(size_t)(32 + lane) * G + gw] = m1;
        }
    } else if (i < nrem) {
#pragma unroll
        for (int d = 0; d < RING; ++d) nt_load(ring[d], xp + piece_off(d));
    }
    CRH_STAMP(2);
    grid_wait(&status->bar_a, gridDim.x + wait_extra, true, status, wait_ticks);
    CRH_STAMP(3);

    // ---- 2. thresholds: workgroup q (and q + gridDim.x, ...) owns query q
    for (int q = blockIdx.x; q < NQS; q += gridDim.x) {
        float t = -INFINITY;
        if (q >= nq) {                 // padding column of a short batch: nominates nothing
            t = INFINITY;
        } else if (G >= k) {
            for (int i = tid; i < G; i += NT) col[i] = ord_f32(gmax[(size_t)q * G + i]);
            __syncthreads();
            const uint32_t key = wg_kth_largest_fast<NT, 1024, 256>([&](unsigned int i) { return col[i]; }, (unsigned int)G, (unsigned int)k,
                                                                    ord_f32(-INFINITY), fast, hist, bcast);
            t = unord_f32(key);
            if (t > -INFINITY) t = t - margin;
            __syncthreads();
        }
        if (tid == 0) tau_g[q] = t;
    }
    const unsigned int producers = gridDim.x < (unsigned int)NQS ? gridDim.x : (unsigned int)NQS;
    CRH_STAMP(4);
    grid_wait(&status->bar_b, producers, blockIdx.x < producers, status, wait_ticks);
    CRH_STAMP(5);
    if (tid < NQS) tau_s[tid] = tau_g[tid];
    __syncthreads();
    const float t0 = tau_s[lane & 31], t1 = QB == 2 ? tau_s[32 + (lane & 31)] : INFINITY;

    // ---- 3. filter: rows of a tile whose score reaches the query's threshold become candidates (k_scan MODE 1)
    auto emit = [&](int64_t tile, const f32x16 &c0, const f32x16 &c1, uint32_t vmask) {
        float m0 = c0[0], m1 = c1[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) {
            m0 = fmaxf(m0, c0[r]);
            m1 = fmaxf(m1, c1[r]);
        }
        const bool any = (m0 >= t0) || (m1 >= t1);
        if (__ballot(any) != 0ull && vmask != 0u) {
            const uint32_t rowbase = (uint32_t)(tile * 32);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float tq = qb ? t1 : t0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float sc = qb ? c1[r] : c0[r];
                    const bool pass = ((vmask >> row) & 1u) && (sc >= tq);
                    const unsigned long long pm = __ballot(pass);
                    if (pm != 0ull) {
                        const unsigned int pre = __builtin_amdgcn_mbcnt_hi((unsigned int)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)pm, 0u));
                        const unsigned int pos = wcnt + pre;
                        if (pass && pos < (unsigned int)wave_cap) {
                            u32x4 e;
                            e.x = f32_bits(sc);
                            e.y = rowbase + row;
                            e.z = (uint32_t)(qb * 32 + (lane & 31));
                            e.w = 0u;
                            mylist[pos] = e;
                        }
                        wcnt += (unsigned int)__popcll(pm);
                    }
                }
            }
        }
    };
    if (has_sample) emit(stile, a0, a1, rowmask[stile]);

    while (i < nrem) {
        const int inext = i + total;
        const int64_t tile = tile_of(i);
        const u32x4 *xn = (inext < nrem) ? tile_ptr(tile_of(inext)) : xp;
        const uint32_t vmask = rowmask[tile];  // wave-uniform -> scalar load
        f32x16 c0 = {0}, c1 = {0};
        scan_tile(xp, xn, c0, c1);
        emit(tile, c0, c1, vmask);
        xp = xn;
        i = inext;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the run-ahead loads of the last tile are still in flight
    CRH_STAMP(6);
#ifdef CRH_FUSED_STAMPS
    if (lane == 0 && blockIdx.x < 256 && wave < 16) g_fused_stamps[256 * 8 + blockIdx.x * 16 + wave] = wall_clock64();
#endif

    // ---- hand the workgroup's candidates over to the per-query lists (as k_scan does)
    __syncthreads();
    unsigned int *wc = reinterpret_cast<unsigned int *>(qs);  // [WAVES] counts, [64] hist, [64] base, [64] off
    unsigned int *qh = wc + WAVES;
    unsigned int *base = qh + 64;
    unsigned int *off = base + 64;
    if (lane == 0) {
        wc[wave] = wcnt < (unsigned int)wave_cap ? wcnt : (unsigned int)wave_cap;
        atomicMax(&status->max_wave_cnt, wcnt);
        if (wcnt > (unsigned int)wave_cap) atomicAdd(&status->wave_overflow, 1u);
    }
    if (tid < 64) {
        qh[tid] = 0u;
        off[tid] = 0u;
    }
    __syncthreads();
    const u32x4 *wl = wave_lists + (size_t)blockIdx.x * WAVES * wave_cap;
    for (int w = 0; w < WAVES; ++w) {
        const unsigned int n = wc[w];
        for (unsigned int e = tid; e < n; e += NT) atomicAdd(&qh[wl[(size_t)w * wave_cap + e].z & 63u], 1u);
    }
    __syncthreads();
    if (tid < 64) base[tid] = qh[tid] ? atomicAdd(&qcount[tid], qh[tid]) : 0u;
    __syncthreads();
    for (int w = 0; w < WAVES; ++w) {
        const unsigned int n = wc[w];
        for (unsigned int e = tid; e < n; e += NT) {
            const u32x4 c = wl[(size_t)w * wave_cap + e];
            const unsigned int q = c.z & 63u;
            const unsigned int idx = base[q] + atomicAdd(&off[q], 1u);
            if (idx < (unsigned int)qcap) {
                u32x2 o;
                o.x = c.x;
                o.y = c.y;
                qlist[(size_t)q * qcap + idx] = o;
            }
        }
    }
}

// ------------------------------------------------------------------ threshold from the seed sample

// tau[q] = (k-th largest of the G sampled tile maxima of query q) - margin; -inf if G < k or fewer than k
// sampled tiles hold a valid row; +inf for the padding columns q >= nq of a short batch (they nominate nothing).  Any k tiles each contribute >= 1 row at or above their maximum, so
// at least k valid rows score >= the k-th largest maximum: it is a lower bound of the true k-th score.
__global__ __launch_bounds__(256) void k_tau(const float *__restrict__ gmax, int G, int k, float margin, int nq,
                                             float *__restrict__ tau, int qstride)
{
    extern __shared__ uint32_t col[];  // [G] ordered keys of this query's tile maxima
    __shared__ unsigned int hist[4 * 256];
    __shared__ unsigned int bcast[2];
    __shared__ unsigned int fast[256 + 256 + 8 + 8];
    const int q = blockIdx.x;
    float t = -INFINITY;
    if (q >= nq) {   // block-uniform
        t = INFINITY;
    } else if (G >= k) {
        for (int i = threadIdx.x; i < G; i += 256) col[i] = ord_f32(gmax[(size_t)i * qstride + q]);
        __syncthreads();
        // (tile maxima of a filtered or sparse shard may be -inf: they stay out of the bucket range)
        const uint32_t key = wg_kth_largest_fast<256, 256, 256>([&](unsigned int i) { return col[i]; }, (unsigned int)G, (unsigned int)k,
                                                                ord_f32(-INFINITY), fast, hist, bcast);
        t = unord_f32(key);
        if (t > -INFINITY) t = t - margin;
    }
    if (threadIdx.x == 0) tau[q] = t;
}

// ------------------------------------------------------------------ canonical scoring

// Sequential f32 dot of the canonical query (LDS) with one stored row: acc = acc + q_i*x_i, i ascending,
// product and sum rounded separately (file is compiled with -ffp-contract=off).  oracle: orc_dot().
__device__ __forceinline__ float canonical_dot_tiled(const u32x4 *__restrict__ xt, int ksteps, uint32_t row,
                                                     const float *qv)
{
    // A row is ksteps runs of 32 bytes, 1 KiB apart: 48 separate HBM sectors (96 of 16 useful bytes before the halves of a row
    // became neighbours).  Eight chunks are in flight all the time: chunk c is consumed (strictly in index order: the ordered
    // chain of 16 additions) and the chunk 8 further on is requested into its place, so the chain runs UNDER the fetches instead
    // of after them (a batch of 16 fetched, then consumed, then the next batch: 103 -> 90 us of the re-score came from the layout,
    // the rest of this loop's time was the two taking turns).  ksteps % 8 == 0.
    const size_t base = (size_t)(row >> 5) * ksteps * 64;
    auto chunk = [&](int c) { return xt[base + piece_off(c >> 1) + piece_slot(c & 1, (int)(row & 31))]; };   // chunk c = elements [8c, 8c + 8) of the row
    const int nchunks = 2 * ksteps;
    float acc = 0.0f;
    u32x4 pk[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pk[j] = chunk(j);
    for (int c0 = 0; c0 < nchunks; c0 += 8) {
        const bool more = c0 + 8 < nchunks;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float *qq = qv + (c0 + j) * 8;
            const uint32_t w[4] = {pk[j].x, pk[j].y, pk[j].z, pk[j].w};
            if (more) pk[j] = chunk(c0 + 8 + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float p0 = qq[2 * e] * bf16_bits_f32(w[e] & 0xffffu);
                acc = acc + p0;
                float p1 = qq[2 * e + 1] * bf16_bits_f32(w[e] >> 16);
                acc = acc + p1;
            }
        }
    }
    return acc;
}

// The same sum from the row-major side copy of the bf16 rows (crh_i8.hpp, ROW-MAJOR ROWS): the same values in the same order,
// so the same bits; a row is 12 full HBM lines instead of 48 quarter-used ones.
template <int DEPTH = 8>   // 16-byte chunks in flight (nchunks % DEPTH == 0): the chain is a chain of memory round trips, nchunks / DEPTH deep
__device__ __forceinline__ float canonical_dot_rows(const u32x4 *__restrict__ xrow, int dim, uint32_t row, const float *qv)
{
    const u32x4 *xr = xrow + (size_t)row * (size_t)(dim >> 3);
    const int nchunks = dim >> 3;                      // chunk c = elements [8c, 8c + 8)
    float acc = 0.0f;
    u32x4 pk[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) pk[j] = xr[j];
    for (int c0 = 0; c0 < nchunks; c0 += DEPTH) {
        const bool more = c0 + DEPTH < nchunks;
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const float *qq = qv + (c0 + j) * 8;
            const uint32_t w[4] = {pk[j].x, pk[j].y, pk[j].z, pk[j].w};
            if (more) pk[j] = xr[c0 + DEPTH + j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float p0 = qq[2 * e] * bf16_bits_f32(w[e] & 0xffffu);
                acc = acc + p0;
                float p1 = qq[2 * e + 1] * bf16_bits_f32(w[e] >> 16);
                acc = acc + p1;
            }
        }
    }
    return acc;
}

__device__ __forceinline__ float canonical_dot_f32(const float *__restrict__ xf32, int dim, uint32_t row, const float *qv)
{
    const float4 *xr = reinterpret_cast<const float4 *>(xf32 + (size_t)row * dim);
    float acc = 0.0f;
    for (int c0 = 0; c0 < (dim >> 2); c0 += 16) {   // dim % 64 == 0
        float4 x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = xr[c0 + j];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float *qq = qv + 4 * (c0 + j);
            float p;
            p = qq[0] * x[j].x;
            acc = acc + p;
            p = qq[1] * x[j].y;
            acc = acc + p;
            p = qq[2] * x[j].z;
            acc = acc + p;
            p = qq[3] * x[j].w;
            acc = acc + p;
        }
    }
    return acc;
}

// One lane's share of a row's dot with the query in LDS when EIGHT lanes split the row (lane `sub` takes every eighth 16-byte
// chunk, so all of a row's bytes are requested at once); the caller adds the eight shares (three shfl_xor).  ANY summation
// order: the result is within 2 * dim * 2^-24 * |q| |x| of the canonical score (both are f32 sums of the same dim products) --
// it nominates or bounds, it never becomes a returned score.  xf32 != nullptr: the row comes from the f32 master.
template <int DEPTH = 12>   // 16-byte loads a lane requests before it consumes the first (k_scan_i8's threshold phase has registers for 6)
__device__ __forceinline__ float partial_dot8(const u32x4 *__restrict__ xt, const float *__restrict__ xf32, const u32x4 *__restrict__ xrow, int dim,
                                              int ksteps, uint32_t row, const float *qv, int sub)
{
    float acc = 0.f;
    const int nch = xf32 != nullptr && xrow == nullptr ? (dim >> 2) : (dim >> 3);   // 16-byte chunks of the row: 4 f32 or 8 bf16 each
    const u32x4 *rr = xrow != nullptr ? xrow + (size_t)row * (size_t)nch                                  // row-major bf16 rows (crh_i8.hpp, ROW-MAJOR ROWS)
                      : xf32 != nullptr ? reinterpret_cast<const u32x4 *>(xf32 + (size_t)row * dim)       // the f32 master, row-major
                                        : xt + (size_t)(row >> 5) * ksteps * 64;                          // the tiled bf16 image
    const bool tiled = xrow == nullptr && xf32 == nullptr, f32 = xrow == nullptr && xf32 != nullptr;
    const int rin = (int)(row & 31u);
    for (int c0 = sub; c0 < nch; c0 += 8 * DEPTH) {
        u32x4 pk[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const int c = c0 + 8 * j;
            if (c < nch) pk[j] = tiled ? rr[piece_off(c >> 1) + piece_slot(c & 1, rin)] : rr[c];
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const int c = c0 + 8 * j;
            if (c < nch) {
                const uint32_t w[4] = {pk[j].x, pk[j].y, pk[j].z, pk[j].w};
                if (f32) {
                    const float *qq = qv + 4 * c;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = fmaf(qq[e], bits_f32(w[e]), acc);
                } else {
                    const float *qq = qv + 8 * c;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc = fmaf(qq[2 * e], bf16_bits_f32(w[e] & 0xffffu), acc);
                        acc = fmaf(qq[2 * e + 1], bf16_bits_f32(w[e] >> 16), acc);
                    }
                }
            }
        }
    }
    return acc;
}
__device__ __forceinline__ float sum8(float acc)
{
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    return (acc == acc) ? acc : -INFINITY;   // (a stored row holding a NaN bounds nothing)
}

// Step 4 of k_select: exact top-k of Ms unique keys (descending score, ascending row), written with row_base added.  The keys
// sit in LDS (`lkeys`, when in_lds) or in global memory (`sk`).
template <int NT>
__device__ void select_tail(const unsigned long long *sk, unsigned int Ms, bool in_lds, unsigned long long *lkeys, unsigned long long *sortbuf,
                            unsigned int *hist, unsigned int *bcast, unsigned int *scount_p, int k, int64_t row_base, float *os, int64_t *orow)
{
    const int tid = threadIdx.x;
    unsigned int &scount = *scount_p;
    const unsigned int k2 = (unsigned int)k < Ms ? (unsigned int)k : Ms;

    if (in_lds) {
        // keys are unique (distinct rows): the rank of a key is the number of larger keys -- counted from LDS
        // broadcast reads, no barriers, no sort
        for (unsigned int p = tid; p < Ms; p += NT) {
            const unsigned long long key = lkeys[p];
            unsigned int rank = 0u;
            for (unsigned int j = 0; j < Ms; ++j) rank += (lkeys[j] > key) ? 1u : 0u;
            if (rank < k2) {
                os[rank] = unord_f32((uint32_t)(key >> 32));
                orow[rank] = row_base + (int64_t)(uint32_t)(~(uint32_t)key);
            }
        }
        for (unsigned int i = k2 + tid; i < (unsigned int)k; i += NT) {
            os[i] = -INFINITY;
            orow[i] = -1;
        }
                return;
    }

    // general path (more survivors than LDS holds: massive ties / duplicates): 64-bit radix select + sort of the top k
    const unsigned long long kkey =
        wg_kth_largest<unsigned long long, NT>([&](unsigned int i) { return sk[i]; }, Ms, k2, hist, bcast);
    int P = 1;
    while (P < (int)k2) P <<= 1;
    for (int i = tid; i < P; i += NT) sortbuf[i] = 0ull;
    if (tid == 0) scount = 0u;
    __syncthreads();
    for (unsigned int p = tid; p < Ms; p += NT) {
        const unsigned long long key = sk[p];
        if (key >= kkey) {
            const unsigned int o = atomicAdd(&scount, 1u);
            if (o < (unsigned int)CRH_MAX_K) sortbuf[o] = key;
        }
    }
    wg_bitonic_desc<NT>(sortbuf, P);
    for (int i = tid; i < k; i += NT) {
        if (i < (int)k2) {
            const unsigned long long key = sortbuf[i];
            os[i] = unord_f32((uint32_t)(key >> 32));
            orow[i] = row_base + (int64_t)(uint32_t)(~(uint32_t)key);
        } else {
            os[i] = -INFINITY;
            orow[i] = -1;
        }
    }
}

// ------------------------------------------------------------------ final selection

// One workgroup per query.  From the query's candidate list (approximate MFMA scores):
//   1. a_k   = k-th largest approximate score;
//   2. keep every candidate with approx >= a_k - margin  (margin >= 2*max|approx - canonical|, so
//      the canonical top-k is inside);
//   3. re-score the survivors canonically, key = (ord(score) << 32) | ~row;
//   4. exact top-k of the keys (descending score, ascending row), written with row_base added.
// I8 (the candidates come from k_scan_i8, crh_i8.hpp): an entry's score is the UPPER end of the row's interval and qlo holds the
// lower ends; step 1 takes the k-th largest LOWER end, step 2 keeps every candidate whose UPPER end reaches it, no margin.
// With tens of thousands of candidates the ~800 survivors of a query are re-scored by gridDim.y workgroups (their rows are 96
// scattered 16-byte pieces each: one CU's L1 took 215 us for them, stamps of round 3): every workgroup finds the same cut, takes
// the survivors whose candidate index is its own modulo gridDim.y, publishes their keys in `fin`, and the LAST one to finish
// ranks them all (no workgroup waits for another).
template <bool F32, bool I8 = false>
__global__ __launch_bounds__(1024) void k_select(const u32x2 *__restrict__ qlist, const float *__restrict__ qlo,
                                                  const unsigned int *__restrict__ qcount,
                                                  int qcap, unsigned long long *__restrict__ skeys, unsigned long long *__restrict__ fin,
                                                  const float *__restrict__ qn, const u32x4 *__restrict__ xt,
                                                  const float *__restrict__ xf32, int dim, int ksteps, int k,
                                                  float margin, int64_t row_base, float *__restrict__ out_scores,
                                                  int64_t *__restrict__ out_rows, SearchStatus *__restrict__ status,
                                                  const u32x4 *__restrict__ xrow = nullptr)
{
    constexpr int NT = 1024;
    constexpr unsigned int LCAP = 2048;   // survivors kept in LDS (the normal case: ~k + a few)
    constexpr unsigned int SCAP = 16384;  // candidate scores kept in LDS for the selection passes (64 KB)
    __shared__ uint32_t lscore[SCAP];
    __shared__ float qv[2048];
    __shared__ unsigned long long lkeys[LCAP];
    __shared__ unsigned long long sortbuf[CRH_MAX_K];
    __shared__ unsigned int hist[(NT / 64) * 256];
    __shared__ unsigned int bcast[2];
    __shared__ unsigned int scount;
    __shared__ unsigned int fast[1024 + 256 + 2 * (NT / 64) + 8];
    const int q = blockIdx.x, tid = threadIdx.x;
    const unsigned int mtrue = qcount[q];
    const unsigned int M = mtrue < (unsigned int)qcap ? mtrue : (unsigned int)qcap;
    const bool split = I8 && gridDim.y > 1 && M > 4096u;
    if (!split && blockIdx.y != 0) return;
    const unsigned int parts = split ? gridDim.y : 1u, part = split ? blockIdx.y : 0u;
    if (tid == 0) {
        if (blockIdx.y == 0) {
            atomicMax(&status->max_qcount, mtrue);
            atomicAdd(&status->candidates, (unsigned long long)mtrue);
            if (mtrue > (unsigned int)qcap) atomicAdd(&status->q_overflow, 1u);
        }
        scount = 0u;
    }
    for (int i = tid; i < dim; i += NT) qv[i] = qn[(size_t)q * dim + i];
    float *os = out_scores + (size_t)q * k;
    int64_t *orow = out_rows + (size_t)q * k;
    if (M == 0u) {
        for (int i = tid; i < k; i += NT) {
            os[i] = -INFINITY;
            orow[i] = -1;
        }
        return;
    }
    CRH_SEL_STAMP(0);
    const u32x2 *ql = qlist + (size_t)q * qcap;
    const float *qlq = I8 ? qlo + (size_t)q * qcap : nullptr;
    unsigned long long *sk = skeys + (size_t)q * qcap + (size_t)part * ((unsigned int)qcap / parts);   // (a part keeps <= ceil(M / parts) rows)
    auto first_key = [&](unsigned int i) { return I8 ? ord_f32(qlq[i]) : ord_f32(bits_f32(ql[i].x)); };   // what step 1 ranks
    __syncthreads();

    // the four radix passes and the survivor cut all re-read the candidate scores: stage them in LDS once (one batch of
    // independent loads) instead of paying an L2 round trip per element per pass
    const unsigned int kk = (unsigned int)k < M ? (unsigned int)k : M;
    const bool staged = M <= SCAP;
    if (staged) {
#pragma unroll 4
        for (unsigned int i = tid; i < M; i += NT) lscore[i] = first_key(i);
        __syncthreads();
    }
    const uint32_t akey = staged ? wg_kth_largest_fast<NT, 1024, 256>([&](unsigned int i) { return lscore[i]; }, M, kk, 0u, fast, hist, bcast)
                          : I8   ? wg_kth_largest_fast<NT, 1024, 256>(first_key, M, kk, 0u, fast, hist, bcast)   // (tens of thousands of candidates)
                                 : wg_kth_largest<uint32_t, NT>(first_key, M, kk, hist, bcast);
    const uint32_t lower_key = I8 ? akey : ord_f32(unord_f32(akey) - margin);
    CRH_SEL_STAMP(1);

    // (whole waves take part in every round: the compaction uses ballots; the scores of eight rounds are fetched together)
    for (unsigned int i0 = 0; i0 < M; i0 += NT * 8) {
        uint32_t sc8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned int i = i0 + j * NT + tid;
            sc8[j] = i < M ? ((staged && !I8) ? lscore[i] : ord_f32(bits_f32(ql[i].x))) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (i0 + j * NT >= M) break;                   // (workgroup-uniform)
            const unsigned int i = i0 + j * NT + tid;
            const bool keep = i < M && sc8[j] >= lower_key && (!split || (i % parts) == part);
            const unsigned long long bm = __ballot(keep);   // one LDS atomic per wave, not one per survivor
            unsigned int base = 0u;
            if ((tid & 63) == 0 && bm != 0ull) base = atomicAdd(&scount, (unsigned int)__popcll(bm));
            base = __shfl(base, 0);
            if (keep) {
                const unsigned int p = base + __builtin_amdgcn_mbcnt_hi((unsigned int)(bm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)bm, 0u));
                const unsigned int row = ql[i].y;
                if (p < LCAP) lkeys[p] = (unsigned long long)row;
                sk[p] = (unsigned long long)row;
            }
        }
    }
    __syncthreads();
    CRH_SEL_STAMP(2);
    const unsigned int Ms = scount;
    if (tid == 0 && blockIdx.x < 64) {
#ifdef CRH_FUSED_STAMPS
        g_select_stamps[blockIdx.x * 16 + 8] = Ms;
        g_select_stamps[blockIdx.x * 16 + 9] = M;
#endif
    }
    const bool in_lds = Ms <= LCAP;
    if constexpr (I8) {
        // Behind the int8 scan the survivors of a query are many (its intervals are wide) and the ordered 768-term chain of
        // the canonical score is a chain of memory round trips per row.  So the survivors are first scored in ANY order, eight
        // lanes per row with every byte of the row requested at once (partial_dot8: within `margin` / 2 of the canonical
        // score); k_select_final then runs the canonical chain only for the rows whose fast score lies within `margin` of the
        // k-th largest fast score -- k + a few rows per query -- and decides every returned id and score as before.  (The true
        // top-k pass: at most k - 1 rows score canonically above the true k-th score s*, so the k-th largest fast score is
        // <= s* + margin / 2, and a row of the true top-k has fast >= s* - margin / 2.)
        const int sub = tid & 7;
        for (unsigned int p0 = 0; p0 < Ms; p0 += NT / 8) {
            const unsigned int p = p0 + (unsigned int)(tid >> 3);
            float acc = 0.f;
            uint32_t row = 0u;
            if (p < Ms) {
                row = (uint32_t)(in_lds ? lkeys[p] : sk[p]);
                acc = partial_dot8(xt, F32 ? xf32 : nullptr, F32 ? nullptr : xrow, dim, ksteps, row, qv, sub);
            }
            acc = sum8(acc);
            if (p < Ms && sub == 0) {
                const unsigned long long key = ((unsigned long long)ord_f32(acc) << 32) | (unsigned long long)(~row);
                if (in_lds)
                    lkeys[p] = key;
                else
                    sk[p] = key;
            }
        }
        __syncthreads();
        CRH_SEL_STAMP(3);
        // hand the fast keys over to k_select_final (the kernel boundary is the meeting point of a query's workgroups)
        if (tid == 0) bcast[0] = Ms ? atomicAdd(&status->qsurv[q], Ms) : 0u;
        __syncthreads();
        unsigned long long *fq = fin + (size_t)q * qcap;
        const unsigned int gbase = bcast[0];
        for (unsigned int p = tid; p < Ms; p += NT) fq[gbase + p] = in_lds ? lkeys[p] : sk[p];
        return;
    }
    // Canonical re-score: one thread per survivor walks its row (96 pieces of 16 bytes, fetched 16 at a time, consumed in index
    // order).  Splitting a row over 2 / 4 / 8 lanes (all lanes fetch at once, the running sum handed from lane to lane in index
    // order, products formed ahead of the ordered additions) returns the same bits with a quarter of the memory round trips --
    // and took 65 / 65 / 18 us LONGER per batch (tools: lib variants, 10M rows, 50 steps): every wave of the workgroup then
    // executes the whole ordered chain for a handful of rows, where here three waves do.
    for (unsigned int p = tid; p < Ms; p += NT) {
        const uint32_t row = (uint32_t)(in_lds ? lkeys[p] : sk[p]);
        const float c = F32 ? canonical_dot_f32(xf32, dim, row, qv) : canonical_dot_tiled(xt, ksteps, row, qv);
        const unsigned long long key = ((unsigned long long)ord_f32(c) << 32) | (unsigned long long)(~row);
        if (in_lds)
            lkeys[p] = key;
        else
            sk[p] = key;
    }
    __syncthreads();
    CRH_SEL_STAMP(3);
    select_tail<NT>(sk, Ms, in_lds, lkeys, sortbuf, hist, bcast, &scount, k, row_base, os, orow);
    CRH_SEL_STAMP(4);
}

// ------------------------------------------------------------------ final step behind the int8 scan

// k_select<.., I8> left, per query, the fast scores of its survivors (keys (ord(fast) << 32) | ~row in `fkeys`, status->qsurv[q]
// of them).  Here, gridDim.y workgroups per query:
//   1. the k-th largest fast score; rows within `margin` of it are the only ones that can be in the canonical top-k (k_select's
//      comment); every workgroup finds the same cut and takes every gridDim.y-th of those rows;
//   2. the canonical score of its rows: the rows are STAGED in LDS first (all threads fetch, one round trip for the lot), then
//      one thread per row walks the ordered chain out of LDS -- the chain used to sit behind a dozen dependent memory round
//      trips per row (32 us for ~100 rows per query, stamps of round 4);
//   3. keys (ord(score) << 32) | ~row published in `ckeys`; the workgroup that arrives last ranks them: exact top-k,
//      descending score, ascending row, row_base added.
template <bool F32>
__global__ __launch_bounds__(256) void k_select_final(const unsigned long long *__restrict__ fkeys, unsigned long long *__restrict__ ckeys, int qcap,
                                                      const float *__restrict__ qn, const u32x4 *__restrict__ xt, const float *__restrict__ xf32,
                                                      const u32x4 *__restrict__ xrow, int dim, int ksteps, int k, float margin, int64_t row_base,
                                                      float *__restrict__ out_scores, int64_t *__restrict__ out_rows, SearchStatus *__restrict__ status)
{
    constexpr int NT = 256;
    constexpr unsigned int LCAP = 2048;             // keys ranked out of LDS
    constexpr unsigned int MINE = 512;              // rows of one workgroup per round of canonical chains
    constexpr int STAGE_U4 = 4096;                  // 64 KB of staged rows
    __shared__ u32x4 stage[STAGE_U4];
    __shared__ unsigned long long lkeys[LCAP];
    __shared__ unsigned long long sortbuf[CRH_MAX_K];
    __shared__ float qv[2048];
    __shared__ unsigned int hist[(NT / 64) * 256];
    __shared__ unsigned int fast[256 + 256 + 2 * (NT / 64) + 8];
    __shared__ uint32_t mine[MINE];
    __shared__ unsigned int bcast[2];
    __shared__ unsigned int scount;
    const int q = blockIdx.x, tid = threadIdx.x;
    const unsigned int parts = gridDim.y, part = blockIdx.y;
    const unsigned int Mt = status->qsurv[q];
    if (Mt == 0u) return;                           // (no candidates: k_select has written the padding)
    const unsigned long long *fq = fkeys + (size_t)q * qcap;
    unsigned long long *cq = ckeys + (size_t)q * qcap;
    float *os = out_scores + (size_t)q * k;
    int64_t *orow = out_rows + (size_t)q * k;
    for (int i = tid; i < dim; i += NT) qv[i] = qn[(size_t)q * dim + i];
    uint32_t *fh = reinterpret_cast<uint32_t *>(lkeys);        // the fast scores' ord words, while they fit
    const bool staged = Mt <= 2 * LCAP;
    if (staged)
        for (unsigned int i = tid; i < Mt; i += NT) fh[i] = (uint32_t)(fq[i] >> 32);
    if (tid == 0) scount = 0u;
    __syncthreads();
    CRH_SEL_STAMP(4);
    const unsigned int k3 = (unsigned int)k < Mt ? (unsigned int)k : Mt;
    const uint32_t fkey = staged ? wg_kth_largest_fast<NT, 256, 256>([&](unsigned int i) { return fh[i]; }, Mt, k3, 0u, fast, hist, bcast)
                                 : wg_kth_largest_fast<NT, 256, 256>([&](unsigned int i) { return (uint32_t)(fq[i] >> 32); }, Mt, k3, 0u, fast, hist, bcast);
    const float fk = unord_f32(fkey);
    const uint32_t cut = fk > -INFINITY ? ord_f32(fk - margin) : 0u;
    __syncthreads();
    // this workgroup's rows: every parts-th position of the list, those at or above the cut.  Windows of MINE * parts positions
    // (so that `mine` holds a window's rows whatever the scores are -- masses of equal ones pass the cut together); a window's
    // rows are staged and chained in rounds of as many rows as the stage holds.
    const int cpr = F32 ? (dim >> 2) : (dim >> 3);  // 16-byte chunks per row
    const int pitch = cpr + 1;                      // (+1 chunk: the lanes of a wave read different rows; a pitch of whole KiB would put them on the same banks)
    const unsigned int rb = (unsigned int)(STAGE_U4 / pitch) < (unsigned int)NT ? (unsigned int)(STAGE_U4 / pitch) : (unsigned int)NT;
    for (unsigned int w0 = 0; w0 < Mt; w0 += MINE * parts) {
        if (tid == 0) scount = 0u;
        __syncthreads();
        const unsigned int w1 = w0 + MINE * parts < Mt ? w0 + MINE * parts : Mt;
        for (unsigned int p = w0 + part + (unsigned int)tid * parts; p < w1; p += NT * parts) {
            const unsigned long long key = fq[p];
            if ((uint32_t)(key >> 32) >= cut) mine[atomicAdd(&scount, 1u)] = ~(uint32_t)key;
        }
        __syncthreads();
        const unsigned int nw = scount;
        if (nw == 0u) continue;                     // (workgroup-uniform)
        if (tid == 0) bcast[0] = atomicAdd(&status->qsurv2[q], nw);
        __syncthreads();
        const unsigned int gbase = bcast[0];
        for (unsigned int r0 = 0; r0 < nw; r0 += rb) {
            const unsigned int nm = nw - r0 < rb ? nw - r0 : rb;
            // stage the rows: chunk c of row j at stage[j * pitch + c]; every thread's loads of a sweep are requested together
            const unsigned int total = nm * (unsigned int)cpr;
            for (unsigned int e0 = tid; e0 < total; e0 += NT * 8) {
                u32x4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned int e = e0 + j * NT;
                    if (e < total) {
                        const unsigned int rj = e / (unsigned int)cpr, c = e % (unsigned int)cpr;
                        const uint32_t row = mine[r0 + rj];
                        if (F32)
                            v[j] = reinterpret_cast<const u32x4 *>(xf32 + (size_t)row * dim)[c];
                        else if (xrow != nullptr)
                            v[j] = xrow[(size_t)row * (size_t)cpr + c];
                        else
                            v[j] = xt[(size_t)(row >> 5) * ksteps * 64 + piece_off(c >> 1) + piece_slot(c & 1, (int)(row & 31u))];
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned int e = e0 + j * NT;
                    if (e < total) stage[(e / (unsigned int)cpr) * pitch + (e % (unsigned int)cpr)] = v[j];
                }
            }
            __syncthreads();
            // the ordered chain out of LDS: acc = acc + q_i * x_i, i ascending, product and sum rounded separately (orc_dot)
            if ((unsigned int)tid < nm) {
                const u32x4 *xr = stage + (unsigned int)tid * pitch;
                float acc = 0.0f;
                if (F32) {
#pragma unroll 4
                    for (int c = 0; c < cpr; ++c) {
                        const u32x4 pk = xr[c];
                        const float *qq = qv + 4 * c;
                        float p;
                        p = qq[0] * bits_f32(pk.x);
                        acc = acc + p;
                        p = qq[1] * bits_f32(pk.y);
                        acc = acc + p;
                        p = qq[2] * bits_f32(pk.z);
                        acc = acc + p;
                        p = qq[3] * bits_f32(pk.w);
                        acc = acc + p;
                    }
                } else {
#pragma unroll 4
                    for (int c = 0; c < cpr; ++c) {
                        const u32x4 pk = xr[c];
                        const uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w};
                        const float *qq = qv + 8 * c;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float p0_ = qq[2 * e] * bf16_bits_f32(w[e] & 0xffffu);
                            acc = acc + p0_;
                            float p1_ = qq[2 * e + 1] * bf16_bits_f32(w[e] >> 16);
                            acc = acc + p1_;
                        }
                    }
                }
                const uint32_t row = mine[r0 + (unsigned int)tid];
                cq[gbase + r0 + (unsigned int)tid] = ((unsigned long long)ord_f32(acc) << 32) | (unsigned long long)(~row);
            }
            __syncthreads();
        }
    }
    CRH_SEL_STAMP(5);
    // publish; the workgroup that arrives last ranks
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int arrived = __hip_atomic_fetch_add(&status->qdone2[q], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int total = 0u;
        if (arrived == parts - 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            total = __hip_atomic_load(&status->qsurv2[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bcast[0] = arrived == parts - 1u ? 1u : 0u;
        bcast[1] = total;
    }
    __syncthreads();
    if (bcast[0] == 0u) return;
    const unsigned int Mc = bcast[1];
    __syncthreads();
    const bool all_lds = Mc <= LCAP;
    if (all_lds) {
        for (unsigned int p = tid; p < Mc; p += NT) lkeys[p] = cq[p];
        __syncthreads();
    }
#ifdef CRH_FUSED_STAMPS
    if (tid == 0 && blockIdx.x < 64) {
        g_select_stamps[blockIdx.x * 16 + 10] = Mt;
        g_select_stamps[blockIdx.x * 16 + 11] = Mc;
    }
#endif
    select_tail<NT>(cq, Mc, all_lds, lkeys, sortbuf, hist, bcast, &scount, k, row_base, os, orow);
    CRH_SEL_STAMP(6);
}

// ------------------------------------------------------------------ cross-shard merge

// One workgroup per query: nlists*k (score,row) pairs -> top-k by (score desc, row asc).
// Pairs with row < 0 are padding.  total = nlists*k <= 8192.
// List l starts at scores + l * s_stride / rows + l * r_stride (elements): nq * k for two plain [nlists, nq, k] arrays, more
// when both live in one all-gathered buffer of [scores | rows] records per rank.
__global__ __launch_bounds__(1024) void k_merge_topk(int nlists, int nq, int k, const float *__restrict__ scores,
                                                      const int64_t *__restrict__ rows, int64_t s_stride, int64_t r_stride,
                                                      float *__restrict__ out_scores, int64_t *__restrict__ out_rows)
{
    constexpr int NT = 1024;
    extern __shared__ unsigned char smem_raw[];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int total = nlists * k;
    int P = 1;
    while (P < total) P <<= 1;
    float *ss = reinterpret_cast<float *>(smem_raw);
    int64_t *rr = reinterpret_cast<int64_t *>(smem_raw + (size_t)P * sizeof(float) + ((P & 1) ? 4 : 0));
    for (int i = tid; i < P; i += NT) {
        if (i < total) {
            const int l = i / k, j = i % k;
            const int64_t r = rows[l * r_stride + (int64_t)q * k + j];
            ss[i] = r < 0 ? -INFINITY : scores[l * s_stride + (int64_t)q * k + j];
            rr[i] = r < 0 ? INT64_MAX : r;
        } else {
            ss[i] = -INFINITY;
            rr[i] = INT64_MAX;
        }
    }
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < (P >> 1); t += NT) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const float sa = ss[lo], sb = ss[hi];
                const int64_t ra = rr[lo], rb = rr[hi];
                // a ranks before b: higher score, then lower row
                const bool a_first = (sa > sb) || (sa == sb && ra < rb);
                if (a_first != desc && !(sa == sb && ra == rb)) {
                    ss[lo] = sb;
                    ss[hi] = sa;
                    rr[lo] = rb;
                    rr[hi] = ra;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < k; i += NT) {
        const bool pad = rr[i] == INT64_MAX;
        out_scores[(size_t)q * k + i] = pad ? -INFINITY : ss[i];
        out_rows[(size_t)q * k + i] = pad ? -1 : rr[i];
    }
}

}  // namespace crh
