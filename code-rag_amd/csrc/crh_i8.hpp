// crh_i8.hpp -- the int8 NOMINATION copy of the corpus and the scan over it (round 3).
//
// The 64-query scan is HBM-bound on the bf16 rows (2 bytes per element, 96 % of the measured read ceiling), and the scan only
// NOMINATES: every returned id and score is decided by k_select's canonical f32 arithmetic on the stored rows.  So the scan may
// read a coarser copy, as long as no row of the true top-k can be missed.  This file keeps a second, 1-byte copy of every row,
//     x_i = s_r * (X_i + d_i),   X_i = rint(x_i * 127 / max|x|) in [-127, 127],   s_r = max|x| / 127   (per row),
// and quantises each canonical query to 15 bits, q_i = s_q * (Q_i + g_i), Q = 128 * H + L with H, L int8.  Then, in exact
// integers, dot = 128 * sum X_i H_i + sum X_i L_i (two v_mfma_i32_32x32x32_i8 per 1-KiB piece and 32-query block) and
//     | x.q - s_r s_q dot |  =  s_r s_q | d.Q + X.g + d.g |  <=  s_r s_q * B_q,
//     B_q = dn * (|Q|_2 + gn) + 127 sqrt(D) * gn,        dn >= |d|_2 of EVERY row (kept by k_requant_i8), gn >= |g|_2
// by Cauchy-Schwarz -- no assumption on the data.  Each row therefore has an interval [lo, hi] = s_r s_q (dot -+ B_q) -+ c that
// contains its canonical score (c: the allowance for the canonical score's own f32 summation order, as in the bf16 scan's margin).
//   * threshold: tau_q = k-th largest over the sample tiles of max_rows(lo): k distinct rows score at least that, so the true
//     k-th score does too;
//   * nomination: every row with hi >= tau_q; the candidate carries (hi, lo);
//   * k_select<.., I8>: L = k-th largest lo of the candidates (again a lower bound of the true k-th score), survivors = hi >= L,
//     canonical re-score and exact top-k as before.
// Results are bit-identical to the bf16 scan's and the oracle's.  On N(0, I)/sqrt(D) rows at D = 768: s_r ~ 9.6e-4, |d| ~ 8.0
// (max 8.6), interval half-width ~ 8.3e-3 = 0.23 sigma of the score distribution; ~21 k candidates and ~800 survivors per query
// and 10M rows, against a pass that reads 7.68 + 0.04 GB instead of 15.36 GB.  Rows with outlier elements get a large s_r and
// are simply nominated more often; when the candidate buffers overflow the host re-runs the batch on the bf16 scan
// (finish_pending) and, after three overflows in a row, rests the int8 copy of that index for 4096 batches.
//
// ROW-MAJOR ROWS (round 4).  The selection behind this scan fetches SINGLE rows: ~800 survivors per query and 10M rows.  In the
// tiled image a row is 48 runs of 32 bytes, 1 KiB apart -- 48 HBM lines of 128 bytes for 1536 useful bytes, and the fetch rate of
// one CU (~24 GB/s of lines) is what bounded k_select (stamps: 50 us for 208 rows per workgroup, whatever the arithmetic).  An index
// that keeps the int8 copy therefore also keeps its bf16 rows ROW-MAJOR ([rows][dim] bf16, +2 bytes per element), written by
// k_requant_i8 from the tiles it reads anyway: a row is then 12 full lines.  Only single-row reads use it (partial_dot8,
// canonical_dot_rows); without it (no memory, an f32 store -- whose f32 master is row-major already) they read the tiles.
//
// Layout of the copy: tile = 32 rows; piece p of a tile = 1 KiB = elements [32p, 32p + 32) of its 32 rows as ONE MFMA operand
// (lane l: row l & 31, elements 32p + 16 (l >> 5) .. + 15, one byte each).  Pieces of a tile are consecutive: D / 32 KiB per tile.
// The copy is derived from the stored rows (k_requant_i8: the bf16 tiles, or the f32 master of an f32 store), never stored in
// snapshots, and brought up to date lazily before a scan.
#pragma once

namespace crh {

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kI8SampleTiles = 8192;   // most sample tiles behind the thresholds of k_scan_i8 (a workgroup keeps one query's maxima in LDS)
constexpr float kI8QueryLevels = 16256.0f;   // 127 * 128: Q = 128 H + L with H in [-127, 127], L in [-64, 63]

// ------------------------------------------------------------------ stored rows -> int8 tiles + per-row scale
// One wave per tile.  dn_bits: f32 bits of the running maximum of |d|_2 (+ allowance for the f32 evaluation) over all rows
// ever quantised; positive floats order like their bits, so atomicMax keeps it.
// F32: the rows come from the f32 master of an f32 store (row-major [rows][dim]) instead of the bf16 tiles, so the copy carries no
// bf16 rounding and the intervals of an f32 store are as narrow as a bf16 store's.
template <bool F32>
__global__ __launch_bounds__(64) void k_requant_i8(const u32x4 *__restrict__ xt, const float *__restrict__ xf32, u32x4 *__restrict__ x8,
                                                   float *__restrict__ srow, unsigned int *__restrict__ dn_bits, int64_t t0, int ksteps,
                                                   int64_t count, u32x4 *__restrict__ xrow)
{
    const int64_t tile = t0 + blockIdx.x;
    const int lane = threadIdx.x, row = lane & 31, hh = lane >> 5;
    const bool live = tile * 32 + row < count;          // (rows past the count: whatever the buffers hold there becomes zeros)
    const u32x4 *tp = xt + (size_t)tile * ksteps * 64;
    const int ks8 = ksteps >> 1, dim = ksteps * 16;
    const float4 *fr = reinterpret_cast<const float4 *>(xf32 + ((size_t)tile * 32 + row) * dim);   // (F32 only)
    // the lane's 16 elements of piece p: elements [32p + 16hh, + 16) of its row
    auto fetch16 = [&](int p, float (&v)[16], bool copy_row) {
        if (!live) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = 0.f;
        } else if (F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 f = fr[(32 * p + 16 * hh) / 4 + j];
                v[4 * j] = f.x;
                v[4 * j + 1] = f.y;
                v[4 * j + 2] = f.z;
                v[4 * j + 3] = f.w;
            }
        } else {
            const int s = 2 * p + hh;                    // bf16 piece holding those elements for the tile's 32 rows
            const u32x4 a = tp[piece_off(s) + piece_slot(0, row)], b = tp[piece_off(s) + piece_slot(1, row)];
            if (copy_row && xrow != nullptr) {           // the row-major side copy (ROW-MAJOR ROWS below): 32 bytes of this lane's row
                u32x4 *dst = xrow + ((size_t)tile * 32 + row) * (size_t)(ksteps * 2) + 2 * s;
                dst[0] = a;
                dst[1] = b;
            }
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[2 * e] = bf16_bits_f32(w[e] & 0xffffu);
                v[2 * e + 1] = bf16_bits_f32(w[e] >> 16);
            }
        }
    };
    float m = 0.f;
    for (int p = 0; p < ks8; ++p) {
        float v[16];
        fetch16(p, v, true);
#pragma unroll
        for (int j = 0; j < 16; ++j) m = fmaxf(m, fabsf(v[j]));
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    if (!(m < INFINITY)) m = 0.f;                       // (a row holding inf / nan: quantised to zeros, scale 0)
    const float s_r = m / 127.0f;
    const float inv = m > 0.f ? 127.0f / m : 0.f;
    float dsq = 0.f;
    for (int p = 0; p < ks8; ++p) {
        float v[16];
        fetch16(p, v, false);
        uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float y = v[j] * inv;
            y = (y == y) ? y : 0.f;
            float X = rintf(y);
            X = fminf(fmaxf(X, -127.f), 127.f);
            const float d = y - X;
            dsq = fmaf(d, d, dsq);
            o[j >> 2] |= ((uint32_t)(int)X & 0xffu) << (8 * (j & 3));   // byte j of the lane's 16
        }
        u32x4 pk;
        pk.x = o[0];
        pk.y = o[1];
        pk.z = o[2];
        pk.w = o[3];
        x8[((size_t)tile * ks8 + p) * 64 + lane] = pk;
    }
    dsq += __shfl_xor(dsq, 32);
    // |d| as evaluated + what the f32 evaluation can hide: x * inv is off by <= 127 * 2^-23 per element, and s_r * inv differs
    // from 1 by <= 2^-22, which moves every d_i by <= 127 * 2^-22: together < 1.3e-3 over 1536 elements
    float dn = sqrtf(dsq) * 1.0001f + 2e-3f;
    if (m == 0.f) dn = 0.f;
#pragma unroll
    for (int d = 16; d > 0; d >>= 1) dn = fmaxf(dn, __shfl_xor(dn, d));
    if (lane < 32) srow[(size_t)tile * 32 + row] = s_r;
    if (lane == 0 && dn > 0.f) atomicMax(dn_bits, f32_bits(dn));
}

// ------------------------------------------------------------------ canonical queries -> 15-bit integer images
// One wave per query slot (64 of them; slots >= nq hold zero queries).  src: the canonical query k_prep_queries has just worked out.
// qfrag8: [QB][2 (H, L)][KS8][64] u32x4 MFMA operand pieces (lane l: query 32 b + (l & 31), elements 32p + 16 (l >> 5) .. + 15).
// qpar:   [64][4] = s_q, |Q|_2, gn (bound of |g|_2), 0.
// Called by k_prep_queries<.., true> (crh_kernels.hpp) with the canonical query still in LDS: one launch prepares both images
// (round 3 launched a second kernel that read the canonical queries back from global memory).
__device__ __forceinline__ void prep_query_i8(const float *src /* the canonical query, LDS */, int dim, int qi, u32x4 *__restrict__ qfrag8,
                                              float *__restrict__ qpar)
{
    const int lane = threadIdx.x;
    const int ks8 = dim >> 5, qb = qi >> 5, c = qi & 31;
    float m = 0.f;
    for (int i = lane; i < dim; i += 64) m = fmaxf(m, fabsf(src[i]));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if (!(m < INFINITY)) m = 0.f;
    const float s_q = m / kI8QueryLevels;
    const float inv = m > 0.f ? kI8QueryLevels / m : 0.f;
    float qsq = 0.f, gsq = 0.f;
    uint32_t *out32 = reinterpret_cast<uint32_t *>(qfrag8);
    for (int c8 = lane; c8 < (dim >> 3); c8 += 64) {      // 8 consecutive elements
        uint32_t hb[2] = {0u, 0u}, lb[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float y = src[c8 * 8 + j] * inv;
            y = (y == y) ? y : 0.f;
            float Qf = rintf(y);
            Qf = fminf(fmaxf(Qf, -kI8QueryLevels), kI8QueryLevels);
            const float g = y - Qf;
            qsq = fmaf(Qf, Qf, qsq);
            gsq = fmaf(g, g, gsq);
            const int Q = (int)Qf;
            const int H = (Q + 64) >> 7;                  // floor((Q + 64) / 128): L = Q - 128 H in [-64, 63]
            const int L = Q - 128 * H;
            hb[j >> 2] |= ((uint32_t)H & 0xffu) << (8 * (j & 3));
            lb[j >> 2] |= ((uint32_t)L & 0xffu) << (8 * (j & 3));
        }
        const int p = c8 >> 2, hh = (c8 >> 1) & 1, half8 = c8 & 1;
        const size_t eh = (((size_t)(qb * 2 + 0) * ks8 + p) * 64 + hh * 32 + c) * 4 + half8 * 2;   // in 32-bit words
        const size_t el = (((size_t)(qb * 2 + 1) * ks8 + p) * 64 + hh * 32 + c) * 4 + half8 * 2;
        out32[eh] = hb[0];
        out32[eh + 1] = hb[1];
        out32[el] = lb[0];
        out32[el + 1] = lb[1];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        qsq += __shfl_xor(qsq, d);
        gsq += __shfl_xor(gsq, d);
    }
    if (lane == 0) {
        // gn: |g| as evaluated + the f32 evaluation's share (y off by <= 16256 * 2^-23, s_q * inv off 1 by 2^-22 -> each g_i moves
        // by < 6e-3; over 1536 elements < 0.24)
        qpar[qi * 4 + 0] = s_q;
        qpar[qi * 4 + 1] = sqrtf(qsq) * 1.0001f;
        qpar[qi * 4 + 2] = m > 0.f ? sqrtf(gsq) * 1.001f + 0.25f : 0.f;
        qpar[qi * 4 + 3] = 0.f;
    }
}

// ------------------------------------------------------------------ the scan over the int8 copy
// Phases as k_scan_fused (sample tiles -> thresholds -> all tiles), with two differences: the sample is G = min(8192, tiles)
// strided tiles walked by whoever's turn it is (the pass reads them again: 200 MB, 2.6 % of it), and the per-row interval
// arithmetic of the header comment replaces the single score.
// LDS: query image QB * 2 * KS8 KiB (96 KB at D = 768) in parts 1 and 3; the threshold step's 32 + 8 + 5 KB in part 2.  WAVES = 8:
// 64 accumulator registers + a 16-deep ring of loads per wave need more than the 128 registers a 16-wave workgroup leaves each wave.
// Three launches of one kernel body -- PART 1: the sample tiles, 2: the thresholds (one workgroup per query), 3: the pass over all
// tiles and the hand-over of the candidates.  Until late in round 4 the three were ONE launch with two grid-wide waits (the form
// k_scan_fused still has): the waits, during which 3/4 of the workgroups idle, cost more than the two launch boundaries that replace
// them (1.325 -> 1.312 ms per batch and 1.383 -> 1.377 on two boxes, interleaved pairs), and a kernel without a grid-wide wait
// needs nobody resident: no time-out, no recovery path, no resting window for this scan.
// bf16 bits of x rounded towards +inf (an upper end stays an upper end): positive values round their magnitude up, negative ones are cut
__device__ __forceinline__ uint32_t bf16_bits_up(float x)
{
    const uint32_t u = f32_bits(x);
    return (u >> 31) ? (u >> 16) : ((u + 0xffffu) >> 16);
}
// The pass's p-th tile when the sample tiles (0, S, 2S, ... (G - 1) S) are NOT scanned again: the tiles in between, in index order.
__device__ __forceinline__ int nonsample_tile(int p, int G, int S)
{
    const int per = S - 1, full = G * per;               // (S == 1: no tile is left for the pass and nobody calls this)
    if (p >= full) return G * S + (p - full);
    const int blk = p / per;
    return blk * S + 1 + (p - blk * per);
}

template <int KS8, int WAVES, int RING, int QB, int PART>
__global__ __launch_bounds__(WAVES * 64) void k_scan_i8(
    const u32x4 *__restrict__ x8, const float *__restrict__ srow, const unsigned int *__restrict__ dn_bits,
    const u32x4 *__restrict__ qfrag8, const float *__restrict__ qpar, const uint32_t *__restrict__ rowmask, int ntiles, int G, int S,
    uint32_t *__restrict__ gkey, float *__restrict__ tau_g, int k, float c_abs, float sqrt_dim, int nq, u32x4 *__restrict__ wave_lists,
    int wave_cap, unsigned int *__restrict__ qcount, u32x2 *__restrict__ qlist, float *__restrict__ qlo, int qcap,
    SearchStatus *__restrict__ status, const u32x4 *__restrict__ xt, const float *__restrict__ xf32,
    const float *__restrict__ qn, const u32x4 *__restrict__ xrow, u32x4 *__restrict__ shi = nullptr)
{
    // `shi` (round 5): the sample launch keeps the UPPER ends of every row of its tiles (bf16, rounded up: QB * 32 bytes per lane and
    // tile, 4 KB per tile at 64 queries, 33.5 MB for 8192 tiles), and the pass takes the sample tiles' candidates from there instead
    // of reading those tiles (201 MB) and multiplying them a second time; it then walks the other tiles only.
    static_assert(PART >= 1 && PART <= 3, "sample tiles / thresholds / pass");
    static_assert(KS8 % RING == 0, "the ring must divide the pieces of a tile (slot s % RING holds piece s of every tile)");
    constexpr int NT = WAVES * 64, NQS = QB * 32, NB = NT < 1024 ? NT : 1024;
    __shared__ u32x4 qs[QB * 2 * KS8 * 64];
    __shared__ uint32_t col[kI8SampleTiles];
    __shared__ unsigned int hist[(NT / 64) * 256];
    __shared__ unsigned int bcast[2];
    __shared__ unsigned int fast[NB + 256 + 2 * (NT / 64) + 8];
    __shared__ float tau_s[NQS];
    __shared__ unsigned int qcnt_l[64];   // candidates of this workgroup per query, counted as they are emitted
    __shared__ unsigned int wpre[WAVES + 1];
    __shared__ unsigned int wcnt_l[WAVES];   // candidates of each wave so far (unclamped): lanes reserve their list positions here
    __shared__ float hwf[64];                // s_q * B_q of each query: a row's interval is 2 (s_r * hwf[q] + c) wide
    __shared__ int next_m;                // tiles handed out so far to this workgroup's waves (main loop)
    __shared__ unsigned long long cslot[8];   // (slot number << 32) | chunk number of the workgroup's j-th chunk of tiles, in entry j % 8
    __shared__ int next_g;                // ... and sample tiles (the same hand-out: the older wave of a SIMD would otherwise wait at
                                          // the grid-wide wait for the younger one to get through its fixed share)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;

    CRH_STAMP(0);
    if (tid < 64) qcnt_l[tid] = 0u;
    if (tid < WAVES) wcnt_l[tid] = 0u;
    if (tid == 0) {
        next_m = WAVES;
        next_g = WAVES;
        if constexpr (PART == 3) {
            for (int j = 2; j < 8; ++j) cslot[j] = ~0ull;
            cslot[0] = (unsigned long long)blockIdx.x;
            cslot[1] = (1ull << 32) | (gridDim.x + __hip_atomic_fetch_add(&status->next_chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
    // per-lane constants of the lane's query in each block: s_q and B_q (integer-dot units)
    const float dn = bits_f32(*dn_bits);
    float sq[QB], Bq[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const int qi = b * 32 + (lane & 31);
        const float s_q = qpar[qi * 4 + 0], Qn = qpar[qi * 4 + 1], gn = qpar[qi * 4 + 2];
        sq[b] = s_q;
        // + kDotRound: what the f32 evaluation of f = 128 * dotH + dotL and of f + B_q can lose.  |dotH| <= 127 * 127 * D and
        // |dotL| <= 127 * 64 * D.  D <= 1024: both are exact in f32 (< 2^24), |f| < 2^31, the fma and the sum round once each by
        // <= 64.  D = 1536: |dotH| reaches 24.8M > 2^24, so its conversion to f32 may be off by 1 (x 128), and |f| reaches 3.2e9 in
        // [2^31, 2^32), where the two roundings are <= 128 each: 384 in all.  (Round 3 wrote 128 for every width; only saturated
        // rows against a saturated query -- every element at +-max -- come near these magnitudes: tests/test_search_gpu.py,
        // test_saturated_rows_and_queries.)
        constexpr float kDotRound = KS8 * 32 <= 1024 ? 128.0f : 512.0f;
        static_assert(KS8 * 32 <= 2048, "beyond D = 2048 |f| passes 2^32 and dotH 2^25: the allowance has to be derived again");
        Bq[b] = (dn * (Qn + gn) + 127.0f * sqrt_dim * gn) * 1.0001f + kDotRound;
        if (wave == 0 && lane < 32) hwf[qi] = s_q * Bq[b];
    }

    const int total = gridDim.x * WAVES;
    const int gw = blockIdx.x * WAVES + wave;
    u32x4 *mylist = wave_lists + (size_t)gw * wave_cap;
    // A tile's address is wave-uniform: loads take it as a SCALAR base plus the lane's byte offset (one VGPR for the whole kernel).
    // With 64-bit per-lane pointers every piece beyond the 4-KiB immediate range costs an address pair -- the compiler kept two
    // dozen of them live, which is what pushed a 12-deep ring over the 256 registers of a two-wave SIMD.
    auto tile_ptr = [&](int64_t tile) { return x8 + (size_t)tile * (KS8 * 64); };
    const uint32_t lane_off = (uint32_t)lane * 16u;
    auto ld = [&](u32x4 &dst, const u32x4 *piece) {
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(lane_off), "s"(piece) : "memory");
    };

    u32x4 ring[RING];
    const i32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // one tile: KS8 x (wait for the oldest load, four MFMAs, refill the slot RING pieces ahead -- past the tile's end from `xn`)
    auto scan_tile = [&](const u32x4 *xp, const u32x4 *xn, i32x16 (&acc)[QB][2]) {
        asm volatile("" ::: "memory");   // (keeps the query image out of the loop-invariant registers, as in k_scan)
        u32x4 bq[QB][2];
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            bq[b][0] = qs[((b * 2 + 0) * KS8) * 64 + lane];
            bq[b][1] = qs[((b * 2 + 1) * KS8) * 64 + lane];
        }
#pragma unroll
        for (int s = 0; s < KS8; ++s) {
            // order pinned by the sched_barrier: next piece's query operands (LDS), this piece's four MFMAs, the refill
            const int s1 = (s + 1 < KS8) ? s + 1 : s;
            u32x4 nb[QB][2];
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                nb[b][0] = qs[((b * 2 + 0) * KS8 + s1) * 64 + lane];
                nb[b][1] = qs[((b * 2 + 1) * KS8 + s1) * 64 + lane];
            }
            nt_wait<RING - 1>(ring[s % RING]);
            const i32x4 xa = __builtin_bit_cast(i32x4, ring[s % RING]);
#pragma unroll
            for (int b = 0; b < QB; ++b) {   // (the first piece accumulates onto the constant 0: no 64 register clears per tile)
                acc[b][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa, __builtin_bit_cast(i32x4, bq[b][0]), s == 0 ? zero16 : acc[b][0], 0, 0, 0);
#if !defined(CRH_I8_DBG) || CRH_I8_DBG != 1   // (timing ablation 1: no L MFMAs -- wrong results)
                acc[b][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa, __builtin_bit_cast(i32x4, bq[b][1]), s == 0 ? zero16 : acc[b][1], 0, 0, 0);
#else
                acc[b][1] = zero16;
#endif
            }
            const int sp = s + RING;
            const u32x4 *src = (sp < KS8) ? xp + sp * 64 : xn + (sp - KS8) * 64;
            ld(ring[s % RING], src);
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                bq[b][0] = nb[b][0];
                bq[b][1] = nb[b][1];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the intervals of a tile's 16 x QB accumulators of this lane, given the tile's 32 row scales (scalar registers: the tile is
    // wave-uniform).  The two candidates of a select are made opaque first: left alone, the compiler turns the select into ONE
    // load with a lane-dependent index, i.e. spills the scales to scratch and reads them back through the vector-memory counter.
    auto row_scale = [&](const float (&sr)[32], int r) {
        const int rowa = (r & 3) + 8 * (r >> 2);
        float sa = sr[rowa], sb = sr[rowa + 4];
        asm volatile("" : "+s"(sa), "+v"(sb));     // (one of the two may stay in its scalar register: v_cndmask takes it as it is)
        return h ? sb : sa;
    };
    // two values per instruction (v_pk_fma / v_pk_mul / v_pk_add): the two query blocks of one accumulator row, or two rows of one block
    auto intervals = [&](const i32x16 (&acc)[QB][2], const float (&sr)[32], float (&hi)[QB][16]) {
        const f32x2 k128 = {128.0f, 128.0f}, c2 = {c_abs, c_abs};
        if (QB == 2) {
            const f32x2 sq2 = {sq[0], sq[QB - 1]}, B2 = {Bq[0], Bq[QB - 1]};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float s_r = row_scale(sr, r);
                const f32x2 fh = {(float)acc[0][0][r], (float)acc[QB - 1][0][r]}, fl = {(float)acc[0][1][r], (float)acc[QB - 1][1][r]};
                const f32x2 f = __builtin_elementwise_fma(fh, k128, fl);
                const f32x2 w = sq2 * f32x2{s_r, s_r};
                const f32x2 u = __builtin_elementwise_fma(w, f + B2, c2);
                hi[0][r] = u.x;
                hi[QB - 1][r] = u.y;
            }
        } else {
            const f32x2 sq2 = {sq[0], sq[0]}, B2 = {Bq[0], Bq[0]};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 s2 = {row_scale(sr, r), row_scale(sr, r + 1)};
                const f32x2 fh = {(float)acc[0][0][r], (float)acc[0][0][r + 1]}, fl = {(float)acc[0][1][r], (float)acc[0][1][r + 1]};
                const f32x2 f = __builtin_elementwise_fma(fh, k128, fl);
                const f32x2 u = __builtin_elementwise_fma(sq2 * s2, f + B2, c2);
                hi[0][r] = u.x;
                hi[0][r + 1] = u.y;
            }
        }
    };
    // the lower end of an interval from its upper end: hi - 2 (s_r s_q B_q + c), one more allowance for the f32 evaluation
    auto lower_end = [&](float hi_v, const float (&sr)[32], int b, int r) {
        return hi_v - 2.0f * fmaf(row_scale(sr, r) * sq[b], Bq[b], c_abs) - 2e-6f;
    };
    auto prime = [&](const u32x4 *xp) {
#pragma unroll
        for (int d = 0; d < RING; ++d) ld(ring[d], xp + d * 64);
    };

    // ---- 1. sample tiles g = gw, gw + total, ...: per query the largest LOWER end among the tile's valid rows
    if constexpr (PART == 1) {
        int g = gw;
        if (g < G) prime(tile_ptr((int64_t)g * S));
        // the query image comes in behind the first corpus loads: 96 KB per workgroup that the first tile would otherwise wait for
        for (int i = tid; i < QB * 2 * KS8 * 64; i += NT) qs[i] = qfrag8[i];
        __syncthreads();
        while (g < G) {
            int mg = 0;
            if (lane == 0) mg = atomicAdd(&next_g, 1);
            mg = __builtin_amdgcn_readfirstlane(mg);
            const int gn_ = (mg / WAVES) * total + (int)blockIdx.x * WAVES + (mg % WAVES);
            const int64_t tile = (int64_t)g * S;
            const u32x4 *xp = tile_ptr(tile);
            const u32x4 *xn = gn_ < G ? tile_ptr((int64_t)gn_ * S) : xp;   // (past the last sample tile: its own pieces again, unused)
            const uint32_t vmask = rowmask[tile];
            float sr[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) sr[j] = srow[(size_t)tile * 32 + j];
            i32x16 acc[QB][2];
            scan_tile(xp, xn, acc);
            float hi[QB][16];
            intervals(acc, sr, hi);
            if (shi != nullptr) {
#pragma unroll
                for (int b = 0; b < QB; ++b)
#pragma unroll
                    for (int v = 0; v < 2; ++v) {
                        u32x4 pk;
                        pk.x = bf16_bits_up(hi[b][8 * v + 0]) | (bf16_bits_up(hi[b][8 * v + 1]) << 16);
                        pk.y = bf16_bits_up(hi[b][8 * v + 2]) | (bf16_bits_up(hi[b][8 * v + 3]) << 16);
                        pk.z = bf16_bits_up(hi[b][8 * v + 4]) | (bf16_bits_up(hi[b][8 * v + 5]) << 16);
                        pk.w = bf16_bits_up(hi[b][8 * v + 6]) | (bf16_bits_up(hi[b][8 * v + 7]) << 16);
                        shi[((size_t)g * (QB * 2) + b * 2 + v) * 64 + lane] = pk;
                    }
            }
#pragma unroll
            for (int b = 0; b < QB; ++b) {
                // (ord(lower end) with its five low bits given to the row's number inside the tile: rounded DOWN, still a lower
                // end, and the threshold phase learns WHICH row of the tile it belongs to; 0 = no valid row)
                uint32_t m = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const uint32_t key = (ord_f32(lower_end(hi[b][r], sr, b, r)) & ~31u) | (uint32_t)row;
                    m = max(m, ((vmask >> row) & 1u) ? key : 0u);
                }
                m = max(m, (uint32_t)__shfl_xor((int)m, 32));
                if (h == 0) gkey[(size_t)(b * 32 + lane) * G + g] = m;
            }
            g = gn_;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last sample tile's run-ahead loads)
        return;
    }

    // ---- 2. thresholds: workgroup q owns query q (no margin: the intervals carry it)
    // The k tiles with the largest lower ends name k distinct rows; their scores are then worked out from the STORED rows (any
    // summation order: within c_abs of the canonical score) and the k-th largest of those, minus c_abs, is the threshold -- a
    // lower bound of the true k-th score that no longer carries the width of an int8 interval (the k-th largest LOWER END, the
    // threshold of round 3, sits a whole half-width below it: 21 k -> ~9 k candidates per query at 10M Gaussian rows).
    constexpr unsigned int kFloorKey = 0x007fffffu;       // ord(-inf) | 31: keys at or below it belong to no valid row
    constexpr int SELCAP = 384;                           // k <= 256 (kI8MaxK) + room for equal keys
    static_assert(NB + 256 + 2 * (NT / 64) + 8 >= 2 * SELCAP, "sel / selv overlay the scratch of wg_kth_largest_fast");
    static_assert((NT / 64) * 256 >= KS8 * 32, "the canonical query overlays the radix histograms");
    if constexpr (PART == 2)
    for (int q = blockIdx.x; q < NQS; q += gridDim.x) {
        float t = -INFINITY;
        if (q >= nq) {
            t = INFINITY;
        } else if (G >= k) {
            constexpr int DIM = KS8 * 32;
            float qreg[(DIM + NT - 1) / NT];               // the canonical query: requested now, parked in LDS after the selection
#pragma unroll
            for (int j = 0; j < (DIM + NT - 1) / NT; ++j) qreg[j] = (tid + j * NT < DIM) ? qn[(size_t)q * DIM + tid + j * NT] : 0.f;
            {   // the query's G keys: every thread's loads requested together (one round trip, not G / NT of them)
                constexpr int PER = kI8SampleTiles / NT;
                uint32_t kreg[PER];
#pragma unroll
                for (int j = 0; j < PER; ++j) kreg[j] = (tid + j * NT < G) ? gkey[(size_t)q * G + tid + j * NT] : 0u;
#pragma unroll
                for (int j = 0; j < PER; ++j)
                    if (tid + j * NT < G) col[tid + j * NT] = kreg[j];
            }
            __syncthreads();
            CRH_TAU_STAMP(8);
            const uint32_t key = wg_kth_largest_fast<NT, NB, 256>([&](unsigned int i) { return col[i]; }, (unsigned int)G, (unsigned int)k,
                                                                  kFloorKey, fast, hist, bcast);
            __syncthreads();
            CRH_TAU_STAMP(9);
            if (key > kFloorKey) {
                t = unord_f32(key & ~31u);
                unsigned int *sel = fast;                                  // [SELCAP] sample tile << 5 | row inside it
                float *selv = reinterpret_cast<float *>(fast + SELCAP);    // [SELCAP] score of that row
                float *qv = reinterpret_cast<float *>(hist);               // [dim] the canonical query
                if (tid == 0) bcast[0] = 0u;
#pragma unroll
                for (int j = 0; j < (DIM + NT - 1) / NT; ++j)
                    if (tid + j * NT < DIM) qv[tid + j * NT] = qreg[j];
                __syncthreads();
                for (int i = tid; i < G; i += NT) {
                    const uint32_t c = col[i];
                    if (c >= key) {
                        const unsigned int p = atomicAdd(&bcast[0], 1u);
                        if (p < (unsigned int)SELCAP) sel[p] = ((uint32_t)i << 5) | (c & 31u);
                    }
                }
                __syncthreads();
                CRH_TAU_STAMP(10);
                const unsigned int nsel = bcast[0];
                if (nsel <= (unsigned int)SELCAP) {                        // (more: masses of equal keys -- the lower-end threshold stands)
                    // eight lanes per row (partial_dot8): all of a row's bytes are requested at once
                    const int sub = tid & 7;
                    for (unsigned int r0 = 0; r0 < nsel; r0 += NT / 8) {
                        const unsigned int r = r0 + (unsigned int)(tid >> 3);
                        float acc = 0.f;
                        if (r < nsel) {
                            const uint32_t e = sel[r];
                            acc = partial_dot8<6>(xt, xf32, xrow, DIM, KS8 * 2, (uint32_t)((e >> 5) * (uint32_t)S * 32u + (e & 31u)), qv, sub);
                        }
                        acc = sum8(acc);
                        if (r < nsel && sub == 0) selv[r] = acc;
                    }
                    __syncthreads();
                    CRH_TAU_STAMP(11);
                    // the k-th largest of the nsel >= k scores, by counting
                    if ((unsigned int)tid < nsel) {
                        const float v = selv[tid];
                        unsigned int gt = 0u, ge = 0u;
                        for (unsigned int j = 0; j < nsel; ++j) {
                            const float o = selv[j];
                            gt += o > v ? 1u : 0u;
                            ge += o >= v ? 1u : 0u;
                        }
                        if (gt < (unsigned int)k && (unsigned int)k <= ge) bcast[1] = f32_bits(v);
                    }
                    __syncthreads();
                    CRH_TAU_STAMP(12);
                    t = fmaxf(t, bits_f32(bcast[1]) - c_abs);
                }
            }
            __syncthreads();
        }
        if (tid == 0) tau_g[q] = t;
    }
    if constexpr (PART == 2) return;
    // ---- 3. the pass: the first corpus loads, then the query image behind them
    const bool stored = shi != nullptr;                     // the sample tiles' upper ends are on record: the pass walks the other tiles
    const int npass = stored ? ntiles - G : ntiles;        // positions of the pass
    auto order = [&](int p) { return stored ? nonsample_tile(p, G, S) : p; };   // position in the pass -> tile
    if (gw < npass) prime(tile_ptr(order(gw)));
    for (int i = tid; i < QB * 2 * KS8 * 64; i += NT) qs[i] = qfrag8[i];
    CRH_STAMP(5);
    if (tid < NQS) tau_s[tid] = tau_g[tid];
    __syncthreads();
    float tq[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) tq[b] = tau_s[b * 32 + (lane & 31)];

    // Rows whose UPPER end reaches the query's threshold become candidates (hi, row, query; the lower end follows in the hand-over).
    // `rounded`: the upper ends come from the sample launch's record (rounded up to bf16): the hand-over allows for that rounding
    // when it works out the lower end.
    auto emit_tile = [&](int64_t tile, uint32_t vmask, const float (&hi)[QB][16], uint32_t rounded) __attribute__((always_inline)) {
        // Which of the lane's 16 x QB values pass: one word per lane (bit 16 b + r), the validity of the lane's rows folded in.
        uint32_t pmask = 0u;
#pragma unroll
        for (int b = 0; b < QB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) pmask |= (hi[b][r] >= tq[b]) ? (1u << (b * 16 + r)) : 0u;
        {
            const uint32_t t = vmask >> (4 * h);          // the lane's rows are (r & 3) + 8 (r >> 2) + 4 h: nibbles 0, 2, 4, 6 of t
            const uint32_t v16 = (t & 0xfu) | ((t >> 4) & 0xf0u) | ((t >> 8) & 0xf00u) | ((t >> 12) & 0xf000u);
            pmask &= v16 | (v16 << 16);
        }
#if defined(CRH_I8_DBG) && CRH_I8_DBG == 2   // (timing ablation 2: no emit code -- no candidates)
        asm volatile("" ::"v"(pmask));
        pmask = 0u;
#endif
        if (pmask != 0u) {
            // Every lane emits its OWN candidates (the set bits of its word): it reserves list positions with one LDS atomic and
            // walks its bits -- one or two as a rule, a tile holds ~4 candidates over 64 lanes.  The form before this one entered a
            // ballot / prefix / store block per (block, accumulator) position with a candidate, ~230 cycles each, 70 us of the pass.
            // The candidate's lower end is worked out in the hand-over (it needs the row's scale: one gather there, nothing here).
            const uint32_t rowbase = (uint32_t)(tile * 32) + 4u * (uint32_t)h;
            unsigned int pos = atomicAdd(&wcnt_l[wave], (unsigned int)__popc(pmask));
            uint32_t m = pmask;
            do {
                const int bit = __builtin_ctz(m);
                m &= m - 1u;
                // hi[bit >> 4][bit & 15] without indexing registers by a lane-dependent number: compare-and-select over the 16 x QB values
                float hv = hi[0][0];
#pragma unroll
                for (int b = 0; b < QB; ++b)
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) hv = (bit == b * 16 + rr) ? hi[b][rr] : hv;
                const int r = bit & 15;
                if (pos < (unsigned int)wave_cap) {
                    u32x4 e;
                    e.x = f32_bits(hv);
                    e.y = rowbase + (uint32_t)((r & 3) + 8 * (r >> 2));
                    e.z = (uint32_t)((bit >> 4) * 32 + (lane & 31));
                    e.w = rounded;
                    mylist[pos] = e;
                    atomicAdd(&qcnt_l[e.z], 1u);
                }
                ++pos;
            } while (m != 0u);
        }
    };
    // the sample tiles, from the record of their upper ends: sample tile g of this wave's share, 32 bytes per lane and query block
    if (stored) {
        for (int g = gw; g < G; g += total) {
            float hi[QB][16];
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const u32x4 pk = shi[((size_t)g * (QB * 2) + b * 2 + v) * 64 + lane];
                    const uint32_t w4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        hi[b][8 * v + 2 * e] = bits_f32(w4[e] << 16);
                        hi[b][8 * v + 2 * e + 1] = bits_f32(w4[e] & 0xffff0000u);
                    }
                }
            const int64_t tile = (int64_t)g * S;
            emit_tile(tile, rowmask[tile], hi, 1u);
        }
    }
    int i = gw;
    const u32x4 *xp = tile_ptr(i < npass ? order(i) : 0);
    while (i < npass) {
        // Which tile a wave scans next is decided as the pass goes, at both levels.  The pass is cut into chunks of WAVES consecutive
        // tiles; a workgroup starts on chunk blockIdx.x and draws further chunks from ONE counter in device memory
        // (status->next_chunk), its waves draw the tiles of a chunk from a counter in LDS.  Inside a workgroup: the SIMD issues its
        // older wave first, and with a fixed list per wave the younger four finished 80-130 us after the older four on half the
        // loads in flight.  Across workgroups: the even XCDs stream faster than the odd ones (their workgroups were done ~100 us
        // before the last); this pass is not purely HBM-bound, so -- unlike the bf16 scan, where the same hand-out only cost its
        // bookkeeping -- what the early finishers leave does not speed the others up, and sharing the tail does: kernel 1.282 / 1.302 /
        // 1.285 / 1.283 -> 1.236 / 1.236 / 1.241 / 1.239 ms, interleaved runs on one box (profiles/r03_chunk_feed_ab.txt).  The wave that draws a chunk's first
        // tile fetches the NEXT chunk's number (one device atomic per WAVES tiles, a tile time ahead of its first use) and publishes
        // it in LDS with its slot number; a wave reads the entry right after its draw.  Chunks past the end make tiles past the
        // end: a wave stops at the first one (the counter only grows, so every later draw is past the end as well).
#if !defined(CRH_I8_STATIC)
        int inext_l = 0;
        if (lane == 0) {
            const int m = atomicAdd(&next_m, 1);
            const unsigned int j = (unsigned int)(m / WAVES), w = (unsigned int)(m % WAVES);
            if (w == 0) {
                const unsigned int c = gridDim.x + __hip_atomic_fetch_add(&status->next_chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&cslot[(j + 1) % 8], ((unsigned long long)(j + 1) << 32) | c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            unsigned long long e;
            while (((e = __hip_atomic_load(&cslot[j % 8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 32) != j) __builtin_amdgcn_s_sleep(1);
            const unsigned int c = (unsigned int)e;
            inext_l = c > 0x7fffffffu / WAVES ? 0x7fffffff : (int)(c * WAVES + w);
        }
        const int inext = __builtin_amdgcn_readfirstlane(inext_l);
#else
        const int inext = i + total;
#endif
        const int64_t tile = order(i);
        const u32x4 *xn = (inext < npass) ? tile_ptr(order(inext)) : xp;
        const uint32_t vmask = rowmask[tile];
        float sr[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) sr[j] = srow[(size_t)tile * 32 + j];
        i32x16 acc[QB][2];
        scan_tile(xp, xn, acc);
        float hi[QB][16];
        intervals(acc, sr, hi);
        emit_tile(tile, vmask, hi, 0u);
        xp = xn;
        i = inext;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the run-ahead loads of the last tile are still in flight
    CRH_STAMP(6);
#ifdef CRH_FUSED_STAMPS
    if (lane == 0 && blockIdx.x < 256 && wave < 16) g_fused_stamps[256 * 8 + blockIdx.x * 16 + wave] = wall_clock64();
#endif

    // ---- hand the workgroup's candidates over to the per-query lists: the per-query counts are known (counted at emit), so
    // one range per (workgroup, query) is reserved with 64 global atomics and ONE sweep copies the entries, eight loads deep
    if (lane == 0) {
        const unsigned int wcnt = wcnt_l[wave];
        wpre[wave + 1] = wcnt < (unsigned int)wave_cap ? wcnt : (unsigned int)wave_cap;
        atomicMax(&status->max_wave_cnt, wcnt);
        if (wcnt > (unsigned int)wave_cap) atomicAdd(&status->wave_overflow, 1u);
    }
    __syncthreads();
    unsigned int *base = reinterpret_cast<unsigned int *>(qs);   // [64] start of this workgroup's range in each query's list
    unsigned int *off = base + 64;                                // [64] entries placed so far
    // (hwf sits in its own LDS words: the query image that base / off overlay is dead, the factors are not)
    if (tid < 64) {
        const unsigned int n = qcnt_l[tid];
        base[tid] = n ? atomicAdd(&qcount[tid], n) : 0u;
        off[tid] = 0u;
    }
    if (tid == 0) {
        unsigned int run = 0u;
        for (int w = 0; w < WAVES; ++w) {
            const unsigned int n = wpre[w + 1];
            wpre[w] = run;
            run += n;
        }
        wpre[WAVES] = run;
    }
    __syncthreads();
    const unsigned int nall = wpre[WAVES];
    const u32x4 *wl = wave_lists + (size_t)blockIdx.x * WAVES * wave_cap;
    for (unsigned int e0 = tid; e0 < nall; e0 += NT * 8) {
        u32x4 cnd[8];
        float s_r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned int e = e0 + j * NT;
            if (e < nall) {
                int w = 0;
#pragma unroll
                for (int t = 1; t < WAVES; ++t) w += (e >= wpre[t]) ? 1 : 0;
                cnd[j] = wl[(size_t)w * wave_cap + (e - wpre[w])];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned int e = e0 + j * NT;
            s_r[j] = e < nall ? srow[cnd[j].y] : 0.f;      // the scale of the candidate's row (row number = position in srow)
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned int e = e0 + j * NT;
            if (e < nall) {
                const unsigned int q = cnd[j].z & 63u;
                const unsigned int idx = base[q] + atomicAdd(&off[q], 1u);
                if (idx < (unsigned int)qcap) {
                    u32x2 o;
                    o.x = cnd[j].x;
                    o.y = cnd[j].y;
                    qlist[(size_t)q * qcap + idx] = o;
                    // lower end = upper end - 2 (s_r s_q B_q + c), one more allowance for the f32 evaluation
                    // (an upper end from the sample launch's record was rounded UP to bf16, by at most 2^-7 of its magnitude: the lower end
                    // steps down by as much -- cnd.w says so)
                    const float up = bits_f32(cnd[j].x);
                    qlo[(size_t)q * qcap + idx] = up - 2.0f * fmaf(s_r[j], hwf[q], c_abs) - 2e-6f - (cnd[j].w ? fabsf(up) * (1.0f / 128.0f) : 0.f);
                }
            }
        }
    }
#ifdef CRH_FUSED_STAMPS
    __syncthreads();
    CRH_STAMP(7);
#endif
}

}  // namespace crh
