// crh_gemm256.hpp -- the large-T encoder GEMM: C[M,N] = epi(A[M,K] . W[N,K]^T + bias), 256x256x64 tiles, two wave
// groups per CU running half a phase apart ("ping-pong"): while one group's four waves issue MFMAs, the other group's
// four (their SIMD partners) read fragments from LDS and issue the LDS-DMA of a later k-tile, then the roles swap.
// Included by crh_encoder.hip (types, gelu_erf, pack2, bf2f come from there).
//
// Geometry.  8 waves; group grp = wave >> 2 owns tile rows [grp*128, +128), column wc = wave & 3 owns tile columns
// [wc*64, +64): 128 x 64 outputs per wave = acc[nt 0..3][mt 0..7] (C^T fragments: a lane holds 4 consecutive n of one m).
// A k-tile (64 deep) is computed in 4 PHASES of 16 MFMAs, one output quadrant each:
//     phase 0: rows m0 (mt 0-3) x cols n0 (nt 0-1)      needs fragments A(m0), W(n0)   (8 + 4 ds_read_b128)
//     phase 1: rows m0          x cols n1 (nt 2-3)      needs W(n1)                    (4)
//     phase 2: rows m1 (mt 4-7) x cols n1               needs A(m1)                    (8)
//     phase 3: rows m1          x cols n0               needs nothing new
// The staged k-tile is cut into four 16-KiB HALF-TILES by the phase that first needs them, not by position:
//     AH0 = rows {grp*128 + 0..63},  WH0 = cols {wc*64 + 0..31}   (phase 0)
//     WH1 = cols {wc*64 + 32..63}                                 (phase 1)
//     AH1 = rows {grp*128 + 64..127}                              (phase 2)
// LDS: 2 buffers x [AH0 | WH0 | WH1 | AH1] x 16 KiB = 128 KiB + 8 wave-private 4-KiB epilogue images = 160 KiB.
// Every half-tile image is [128 local rows][128 B], chunk c of local row r at slot c ^ (r & 7) (the XOR is applied to
// the per-lane SOURCE address of the LDS-DMA, whose destination is lane-linear); a half-tile = 16 one-KiB pieces, two per
// wave.
//
// Schedule.  Phases are numbered P = 4t + p over the workgroup's whole k-tile sequence t (it runs across tile boundaries).
// Phase P issues exactly one half-tile, the one 6 positions ahead in the order AH0(t), WH0(t), WH1(t), AH1(t), AH0(t+1)...:
//     p=0: WH1(t+1)   p=1: AH1(t+1)   p=2: AH0(t+2)   p=3: WH0(t+2)
// and ends its memory part with `s_waitcnt vmcnt(6)`: all but the three youngest half-tiles have landed, i.e. the one
// issued in phase P-3; it is read in phase P+1 or later (AH0(t+1): issued 4t-2, read 4t+4; WH0: 4t-1 / 4t+4; WH1: 4t / 4t+5;
// AH1: 4t+1 / 4t+6).  A slot is re-issued at least two phases after the phase that read it (AH0: read 4t, re-issued 4t+2;
// the others three phases later).  Barrier ticks: group 0 runs [mem P] b [mfma P] b, group 1 the same one tick later, so
// a wait placed before the tick that ends a memory part is separated from every later read by at least one barrier that
// both groups pass, and every re-issue from every read's completion likewise.  After a tile's epilogue the 16 output
// stores of a wave sit between the in-flight half-tiles in the vmcnt order: the next three waits allow 16 more.  When
// the issue cursor has run out (the workgroup's last k-tiles) the waits drain to zero.
#pragma once

namespace g256 {

constexpr int BM = 256, BN = 256, BK = 64, WAVES = 8;
constexpr int kHalf = 16384, kBuf = 4 * kHalf, kImg = 4096;
constexpr int kLds = 2 * kBuf + WAVES * kImg;   // 163840 = all of a CU's LDS
constexpr int R_AH0 = 0, R_WH0 = 1, R_WH1 = 2, R_AH1 = 3;

__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

// EPI: 0 bias, 1 bias + erf-GELU, 2 bias + residual; the LayerNorm-folded forms (round 5):
//   3 / 4  "LN in": A holds UN-normalised rows r, W the weights scaled by the LayerNorm gain (W' = gamma (.) W), and the epilogue
//          finishes the normalisation per row m: y = rstd_m (acc - mu_m c_n) + b'_n = fma(acc, rstd_m, fma(nmr_m, c_n, b'_n)) with
//          rstats[m] = (rstd_m, nmr_m = -mu_m rstd_m), aux0 = c_n = sum_k W'_nk, bias = b'_n = b_n + sum_k beta_k W_nk; 4 adds erf-GELU.
//   5      "residual + stats out": out = bf16(acc + bias_n + h), h the residual -- normalised on the fly when rstats is given:
//          out = bf16(fma(fma(r, rstd_m, nmr_m), aux0_n /*gamma*/, acc + bias_n)) with bias = this GEMM's bias + the LayerNorm's beta
//          (folded on the host), else out = bf16((acc + bias_n) + r), which is epilogue 2.  `out` is the UN-normalised input of
//          the next LayerNorm; per (row, 32-column slot) the mean and the sum of squared deviations of the rounded outputs go to
//          partials[m][n / 32] (crh_encoder.hip: LnAcc / ln_join_row; k_ln_finalize turns a row's N / 32 pairs into (rstd, nmr)).
// M arbitrary (guarded), N % 256 == 0, K % 128 == 0, K >= 256.
// DBG (timing ablations, wrong results): 1 = no global stores in the epilogue, 2 = no epilogue at all, 4 = 1.5x the LDS-DMA
// (one more half-tile every other phase, into the idle epilogue images) and 4 more fragment reads per phase -- the load a
// two-pass schedule with half-height wave tiles would put on the main loop
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(WAVES * 64) void k_gemm_pp(const bf16_t *__restrict__ A, const bf16_t *__restrict__ W,
                                                        const float *__restrict__ bias, const bf16_t *__restrict__ R,
                                                        bf16_t *__restrict__ C, int M, int N, int K,
                                                        const float *__restrict__ aux0 = nullptr,
                                                        const float *__restrict__ rstats = nullptr, float *__restrict__ partials = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int g = lane >> 4, c16 = lane & 15;
    const int lane_ = lane, g_ = g, c16_ = c16;   // (epilogue 5 shadows the three with opaque copies)

    // ---- tile list of this workgroup: same XCD-aware order as k_gemm_nt (contiguous panel range per XCD label,
    // super-tiles of 4 panels x 8 column tiles)
    constexpr int HM = 4, WN = 8;
    const int nb = N / BN, nk = K / BK;
    const int panels = (M + BM - 1) / BM;
    const int bpx = gridDim.x >> 3;
    const int jx = blockIdx.x >> 3, xcd = blockIdx.x & 7;
    const bool by_panel = panels >= 16;
    int first, cnt;
    {
        const int units = by_panel ? panels : panels * nb;
        const int q = units >> 3, r = units & 7;
        first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        cnt = q + (xcd < r ? 1 : 0);
    }
    const int my_tiles = by_panel ? cnt * nb : cnt;
    const int ntl = jx < my_tiles ? (my_tiles - jx + bpx - 1) / bpx : 0;
    if (ntl == 0) return;   // whole workgroup, before any barrier
    // Epilogues 2 / 5 (the FFN2 of a layer: A = the 400 MB the FFN1 before it has just written, in THIS order, each XCD label walking
    // its own range of panels) walk the row panels from the last to the first: every label starts on the rows it wrote last, which
    // are still in its L2 / the last-level cache, instead of on the ones written first and evicted since (+0.3-0.7 % of the encoder
    // leg on two of three devices, level on the third: profiles/r05_ffn2_panel_order.txt).  Order only: a tile's bits do not change.
    constexpr bool kLastFirst = EPI == 2 || EPI == 5;
    auto tile_origin = [&](int ts, int &tm0, int &tn0) {
        const int u = jx + ts * bpx;
        if (!by_panel) {
            const int tile = first + u;
            tm0 = (kLastFirst ? panels - 1 - tile / nb : tile / nb) * BM;
            tn0 = (tile % nb) * BN;
            return;
        }
        const int gsz = HM * nb;
        const int sr = u / gsz, ug = u - sr * gsz;
        const int hm = (cnt - sr * HM) < HM ? (cnt - sr * HM) : HM;
        const int bsz = hm * WN;
        const int sc = ug / bsz, ub = ug - sc * bsz;
        const int wn_ = (nb - sc * WN) < WN ? (nb - sc * WN) : WN;
        const int pm = ub / wn_, pn = ub - pm * wn_;
        tm0 = (kLastFirst ? panels - 1 - (first + sr * HM + pm) : first + sr * HM + pm) * BM;
        tn0 = (sc * WN + pn) * BN;
    };

    // ---- issue cursor (6 half-tiles ahead of the phase being computed)
    const int prow = lane >> 3, pslot = lane & 7;
    const int csrc = (pslot ^ prow) * 8;            // source chunk (elements) of this lane: slot s of row r holds chunk s ^ (r & 7)
    // 32-bit BYTE offsets from A / W (the host checks M*K*2 and N*K*2 < 4 GiB): uniform base + per-lane offset lets the
    // LDS-DMA use its scalar-base form and keeps the cursor at 5 VGPRs
    uint32_t pa[2][2];                              // [h][i]: A row of piece i of half-tile AHh, clamped to M-1, + chunk
    uint32_t pw;                                    // W row of piece 0 of WH0 + chunk (pieces: + (h*32 + i*8) * K)
    int c_ts = 0, c_k0 = 0;
    bool c_more = true;
    // origin of the tile the cursor enters next: the cursor runs 1.5 k-tiles ahead, so it crosses into tile ts+1 inside tile
    // ts's loop; that origin is computed ONCE at the top of tile ts (scalar registers), not inside the six inlined k-tile bodies
    int cm0, cn0;
    tile_origin(0, cm0, cn0);
    auto cursor_tile = [&]() {
        const int m0 = cm0, n0 = cn0;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int m = m0 + grp * 128 + h * 64 + wc * 16 + i * 8 + prow;   // piece = wave*2+i: local row wave*16 + i*8 + prow
                m = m < M ? m : M - 1;
                pa[h][i] = ((uint32_t)m * (uint32_t)K + (uint32_t)csrc) * 2u;
            }
        pw = ((uint32_t)(n0 + (wave >> 1) * 64 + (wave & 1) * 16 + prow) * (uint32_t)K + (uint32_t)csrc) * 2u;
    };
    cursor_tile();
    auto cursor_advance = [&]() {
        c_k0 += BK;
        if (c_k0 == K) {
            c_k0 = 0;
            ++c_ts;
            if (c_ts < ntl)
                cursor_tile();
            else
                c_more = false;
        }
    };
    // (no helper lambda returning an address_space(3) pointer here: clang's HOST pass then silently drops the kernel's
    // stub and the library fails to load with an undefined symbol)
    // one half-tile (2 pieces per wave) of the cursor's k-tile into buffer `buf`
    auto stage = [&](int region, int buf) {
        if (!c_more) return;
        const int dst = buf * kBuf + region * kHalf + wave * 2048;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned char *src;
            if (region == R_AH0)
                src = reinterpret_cast<const unsigned char *>(A) + (pa[0][i] + (uint32_t)c_k0 * 2u);
            else if (region == R_AH1)
                src = reinterpret_cast<const unsigned char *>(A) + (pa[1][i] + (uint32_t)c_k0 * 2u);
            else
                src = reinterpret_cast<const unsigned char *>(W) +
                      (pw + ((uint32_t)((region == R_WH1 ? 32 : 0) + i * 8) * (uint32_t)K + (uint32_t)c_k0) * 2u);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)(smem + dst + i * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][8];     // [nt][mt]
    bf16x8 FA[4][2];     // A fragments of the current m-half: [mt'][ks]
    bf16x8 FW0[2][2], FW1[2][2];   // W fragments of n-half 0 / 1: [nt'][ks]
    // Fragment read addresses.  Everything lane-dependent is in two registers (ra for A rows, rw for W rows); the k-half
    // ks = 1 is the same address with bit 6 flipped (chunk ^ 4), the rest are immediates.  The per-phase address is
    // re-derived from ra / rw (two VALU ops) and passed through an empty asm so the compiler cannot hoist a dozen
    // variants out of the six inlined k-tile bodies -- that cost ~16 VGPRs, which it spilled INSIDE the loop (each reload
    // a vmcnt(0) that drained the LDS-DMA queue).
    const uint32_t ra = (uint32_t)swz(grp * 64 + c16, g);   // byte offsets into smem
    const uint32_t rw = (uint32_t)swz(wc * 32 + c16, g);
    auto lds_read = [&](uint32_t off) { return *reinterpret_cast<const bf16x8 *>(smem + off); };
    auto read_a = [&](int buf, int h) __attribute__((always_inline)) {
        uint32_t a0 = ra + (uint32_t)(buf * kBuf + (h ? R_AH1 : R_AH0) * kHalf);
        asm volatile("" : "+v"(a0));
        const uint32_t a1 = a0 ^ 64u;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            FA[mt][0] = lds_read(a0 + mt * 2048);
            FA[mt][1] = lds_read(a1 + mt * 2048);
        }
    };
    auto read_w = [&](int buf, int h, bf16x8 (&fw)[2][2]) __attribute__((always_inline)) {
        uint32_t a0 = rw + (uint32_t)(buf * kBuf + (h ? R_WH1 : R_WH0) * kHalf);
        asm volatile("" : "+v"(a0));
        const uint32_t a1 = a0 ^ 64u;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            fw[nt][0] = lds_read(a0 + nt * 2048);
            fw[nt][1] = lds_read(a1 + nt * 2048);
        }
    };
    // `first`: the tile's first k-tile starts its accumulators from a zero C operand (no 128-register clear per tile)
    auto mfma16 = [&](int nh, int mh, const bf16x8 (&fw)[2][2], const bool first) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const f32x4 c = (first && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[nh * 2 + nt][mh * 4 + mt];
                    acc[nh * 2 + nt][mh * 4 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt][ks], FA[mt][ks], c, 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(0);
    };
    int after_epi = 0;   // phases left whose wait must also let the last epilogue's 16 stores stay in flight
    auto mem_end = [&]() {
        if (!c_more)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (after_epi > 0)
            asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        if (after_epi > 0) --after_epi;
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfma_end = [&]() {
        asm volatile("" ::: "memory");   // no LDS read of the next phase may be hoisted above this tick
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: the first six half-tiles; AH0(0) and WH0(0) must have landed before phase 0 reads them
    stage(R_AH0, 0);
    stage(R_WH0, 0);
    stage(R_WH1, 0);
    stage(R_AH1, 0);
    cursor_advance();
    stage(R_AH0, 1);
    stage(R_WH0, 1);
    if (c_more)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one tick behind group 0 from here on

    // one k-tile out of buffer B (compile-time); the cursor's half-tiles go to the other buffer in phases 0-1 and,
    // after the advance, to this one in phases 2-3
    int m0 = 0, n0 = 0;   // origin of the tile being computed
    float4 bv[4];         // its bias fragment, fetched under the tile's last 16 MFMAs (FW1 is dead by then)
    float b_lin = 0.f, g_lin = 1.f;   // EPI 5: bias and LayerNorm gain of column n0 + wc*64 + lane
    auto extra_load = [&](const bool dma) __attribute__((always_inline)) {
        if (!(DBG & 4)) return;
        if (dma && c_more) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned char *src = reinterpret_cast<const unsigned char *>(W) + (pw + ((uint32_t)(i * 8) * (uint32_t)K + (uint32_t)c_k0) * 2u);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(smem + 2 * kBuf + wave * 2048 + i * 1024), 16, 0, 0);
            }
        }
        uint32_t a0 = rw;
        asm volatile("" : "+v"(a0));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 v = lds_read((a0 ^ (i & 1 ? 64u : 0u)) + (i >> 1) * 2048);
            asm volatile("" ::"v"(v));
        }
    };
    auto ktile = [&](const int B, const bool first, const bool last) __attribute__((always_inline)) {
        read_a(B, 0);
        read_w(B, 0, FW0);
        stage(R_WH1, B ^ 1);
        extra_load(true);
        mem_end();
        mfma16(0, 0, FW0, first);
        mfma_end();

        read_w(B, 1, FW1);
        stage(R_AH1, B ^ 1);
        cursor_advance();
        extra_load(false);
        mem_end();
        mfma16(1, 0, FW1, first);
        mfma_end();

        read_a(B, 1);
        stage(R_AH0, B);
        extra_load(true);
        mem_end();
        mfma16(1, 1, FW1, first);
        mfma_end();

        stage(R_WH0, B);
        extra_load(false);
        if (last && !(DBG & 2)) {
            if (EPI == 5) {   // one column per lane (2 registers; the epilogue spreads them through a 512-byte LDS table)
                b_lin = bias[n0 + wc * 64 + lane];
                g_lin = rstats != nullptr ? aux0[n0 + wc * 64 + lane] : 1.0f;
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bv[nt] = *reinterpret_cast<const float4 *>(bias + n0 + wc * 64 + nt * 16 + 4 * g);
            }
        }
        mem_end();
        mfma16(0, 1, FW0, first);
        mfma_end();
    };

    unsigned char *cimg = smem + 2 * kBuf + wave * kImg;   // [32 rows][64 cols] bf16, XOR-swizzled, wave-private
    for (int ts = 0; ts < ntl; ++ts) {
        m0 = cm0;                 // cursor_tile() last ran for this tile (ts = 0: before the prologue)
        n0 = cn0;
        if (ts + 1 < ntl) tile_origin(ts + 1, cm0, cn0);
        ktile(0, true, false);    // (nk >= 4, even: checked by the host)
        ktile(1, false, false);
        for (int kt = 2; kt < nk - 2; kt += 2) {
            ktile(0, false, false);
            ktile(1, false, false);
        }
        ktile(0, false, false);
        ktile(1, false, true);

        // ---- epilogue.  acc[nt][mt][j] = C[m0 + grp*128 + mt*16 + c16][n0 + wc*64 + nt*16 + 4g + j].  Bias and residual are
        // fetched in one batch; the bf16 tile leaves through the wave's LDS image as full 128-byte rows (16 dwordx4 stores).
        if (DBG & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
            continue;
        }
        if constexpr (EPI == 5) {
            // ---- epilogue 5 (header comment): out = bf16(acc + bias + h), h = the residual, normalised on the fly when rstats is given;
            // statistics of the rounded outputs per (row, 32-column slot).  Registers are what bounds this epilogue (the 128
            // accumulators stay live while it runs): bias and gain live lane-linear in one register each and a 4-column fragment
            // is read from a small LDS table per (q, nt); the residual rows and their statistics come in two batches of four m-tiles.
            const bool res_ln = rstats != nullptr;
            // (lane-derived values made opaque HERE: otherwise every address this block derives from them is hoisted out of the
            // tile loop as an invariant, lives across the main loop and is spilled there -- the k_gemm_pp header's warning)
            int c16 = c16_, g = g_, lane = lane_;
            asm volatile("" : "+v"(c16), "+v"(g), "+v"(lane));
            // residual rows in two batches of 4 m-tiles (two q steps each), as epilogue 2 fetches them: the second batch is
            // issued once the first 32 rows have left, into the registers their accumulators vacated -- two exposed round trips
            // per tile instead of four (a batch per q step measured 9 us per tile over epilogue 2)
            u32x2 rq[2][4][4];   // [batch][nt][mt & 3]
            float2 sq[2][4];     // [batch][mt & 3]
            auto load_b = [&](int batch) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    int m = m0 + grp * 128 + (batch * 4 + mi) * 16 + c16;
                    m = m < M ? m : M - 1;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        rq[batch][nt][mi] = *reinterpret_cast<const u32x2 *>(R + (size_t)m * N + n0 + wc * 64 + nt * 16 + 4 * g);
                    sq[batch][mi] = res_ln ? *reinterpret_cast<const float2 *>(rstats + 2 * (size_t)m) : float2{1.f, 0.f};
                }
            };
            load_b(0);
            // bias and gain of the wave's 64 columns as a 512-byte table in LDS: the last k-tile (always buffer 1) has read its
            // WH1 region for the last time in phase 1, and the next tile's first k-tile refills it in phase 0 -- this wave's own two
            // pieces land on this wave's slice, after its epilogue in program order.  A fragment is then two ds_read_b128 per
            // (q, nt) (the first version fetched it with 8 ds_bpermute and waited for each group: 128 per tile, 3.7 us of waits)
            const uint32_t tbl = (uint32_t)(kBuf + R_WH1 * kHalf) + (uint32_t)wave * 2048u;      // byte offset into smem (stays an LDS address)
            *reinterpret_cast<float *>(smem + tbl + lane * 4) = b_lin;
            *reinterpret_cast<float *>(smem + tbl + 256 + lane * 4) = g_lin;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q == 1) load_b(1);
                uint32_t tq = tbl + (uint32_t)g * 16u;
                asm volatile("" : "+v"(tq));     // (per q: the fragments are not hoisted out of the loop into 32 registers)
                LnAcc la[2][2];   // [mh][slot of the wave's two]
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const float4 b4 = *reinterpret_cast<const float4 *>(smem + tq + nt * 64), g4 = *reinterpret_cast<const float4 *>(smem + tq + 256 + nt * 64);
                    const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh) {
                        const int mt = q * 2 + mh;
                        const u32x2 r2 = rq[mt >> 2][nt][mt & 3];
                        const float2 s2 = sq[mt >> 2][mt & 3];
                        const float h0 = bf2f(r2.x & 0xffffu), h1 = bf2f(r2.x >> 16), h2 = bf2f(r2.y & 0xffffu), h3 = bf2f(r2.y >> 16);
                        // (without statistics: rstd = 1, nmr = 0, gain = 1 -- fma(h, 1, 0) = h and fma(h, 1, v) = v + h: epilogue 2's bits)
                        const float v0 = __builtin_fmaf(__builtin_fmaf(h0, s2.x, s2.y), gg[0], acc[nt][mt][0] + bb[0]);
                        const float v1 = __builtin_fmaf(__builtin_fmaf(h1, s2.x, s2.y), gg[1], acc[nt][mt][1] + bb[1]);
                        const float v2 = __builtin_fmaf(__builtin_fmaf(h2, s2.x, s2.y), gg[2], acc[nt][mt][2] + bb[2]);
                        const float v3 = __builtin_fmaf(__builtin_fmaf(h3, s2.x, s2.y), gg[3], acc[nt][mt][3] + bb[3]);
                        u32x2 o;
                        o.x = pack2(v0, v1);
                        o.y = pack2(v2, v3);
                        ln_acc4(la[mh][nt >> 1], (nt & 1) == 0, bf2f(o.x & 0xffffu), bf2f(o.x >> 16), bf2f(o.y & 0xffffu), bf2f(o.y >> 16));
                        const int row = mh * 16 + c16;
                        const int chunk = nt * 2 + (g >> 1);
                        *reinterpret_cast<u32x2 *>(cimg + row * 128 + ((chunk ^ (row & 7)) << 4) + (g & 1) * 8) = o;
                    }
                }
#pragma unroll
                for (int mh = 0; mh < 2; ++mh) {
                    const float2 p0 = ln_join_row(ln_acc_done8(la[mh][0])), p1 = ln_join_row(ln_acc_done8(la[mh][1]));
                    const int m = m0 + grp * 128 + (q * 2 + mh) * 16 + c16;
                    if (g == 0 && m < M)
                        *reinterpret_cast<float4 *>(partials + ((size_t)m * (N >> 5) + ((n0 >> 5) + wc * 2)) * 2) = float4{p0.x, p0.y, p1.x, p1.y};
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 8 + (lane >> 3), chunk = lane & 7;
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(cimg + row * 128 + ((chunk ^ (row & 7)) << 4));
                    const int m = m0 + grp * 128 + q * 32 + row;
                    if (m < M) *reinterpret_cast<u32x4 *>(C + (size_t)m * N + n0 + wc * 64 + chunk * 8) = v;
                }
            }
            after_epi = (m0 + BM <= M) ? 3 : 0;
            continue;
        }
        // epilogues 3 / 4 ("LN in"): per-row (rstd, -mu rstd) of the A rows and the column sums of the gain-scaled weights
        constexpr bool kLnIn = EPI == 3 || EPI == 4;
        float2 st[8];
        float4 cv[4];
        if (kLnIn) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                int m = m0 + grp * 128 + mt * 16 + c16;
                m = m < M ? m : M - 1;
                st[mt] = *reinterpret_cast<const float2 *>(rstats + 2 * (size_t)m);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cv[nt] = *reinterpret_cast<const float4 *>(aux0 + n0 + wc * 64 + nt * 16 + 4 * g);
        }
        // residual rows in two batches of 4 m-tiles (the second is issued once the first 32 rows have left, into the
        // registers their accumulators vacated: all 8 at once would not fit beside the 128 accumulators)
        u32x2 rv[2][4][4];   // [batch][nt][mt & 3]
        auto load_res = [&](int batch) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    int m = m0 + grp * 128 + (batch * 4 + mi) * 16 + c16;
                    m = m < M ? m : M - 1;
                    rv[batch][nt][mi] = *reinterpret_cast<const u32x2 *>(R + (size_t)m * N + n0 + wc * 64 + nt * 16 + 4 * g);
                }
        };
        if (EPI == 2) load_res(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mh = 0; mh < 2; ++mh) {
                    const int mt = q * 2 + mh;
                    float v0 = acc[nt][mt][0] + bv[nt].x, v1 = acc[nt][mt][1] + bv[nt].y, v2 = acc[nt][mt][2] + bv[nt].z,
                          v3 = acc[nt][mt][3] + bv[nt].w;
                    if (kLnIn) {
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
                    if (EPI == 2) {
                        const u32x2 r2 = rv[mt >> 2][nt][mt & 3];
                        v0 += bf2f(r2.x & 0xffffu);
                        v1 += bf2f(r2.x >> 16);
                        v2 += bf2f(r2.y & 0xffffu);
                        v3 += bf2f(r2.y >> 16);
                    }
                    u32x2 o;
                    o.x = pack2(v0, v1);
                    o.y = pack2(v2, v3);
                    const int row = mh * 16 + c16;
                    const int chunk = nt * 2 + (g >> 1);
                    *reinterpret_cast<u32x2 *>(cimg + row * 128 + ((chunk ^ (row & 7)) << 4) + (g & 1) * 8) = o;
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + (lane >> 3), chunk = lane & 7;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(cimg + row * 128 + ((chunk ^ (row & 7)) << 4));
                const int m = m0 + grp * 128 + q * 32 + row;
                if (m < M && !((DBG & 1) && m >= 0)) {
                    // QKV / FFN1 outputs (150-400 MB per call, read once by the next kernel) leave with the nt policy: the
                    // stores are acknowledged sooner, and the next tile's counted waits stall less on them (-8 % on those
                    // GEMMs); the N = 768 outputs are re-read at once by the LayerNorm and keep the default policy
                    u32x4 *dst = reinterpret_cast<u32x4 *>(C + (size_t)m * N + n0 + wc * 64 + chunk * 8);
                    if (EPI == 2)
                        *dst = v;
                    else
                        __builtin_nontemporal_store(v, dst);
                }
                if (DBG & 1) asm volatile("" ::"v"(v));
            }
            if (EPI == 2 && q == 0) load_res(1);
        }
        after_epi = (m0 + BM <= M) ? 3 : 0;   // a ragged tile may have skipped stores: fall back to the tighter wait
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balances group 1's extra tick
}

}  // namespace g256
