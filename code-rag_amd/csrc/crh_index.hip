// crh_index.hip -- host side of the HBM-resident cosine index behind include/coderag_hip.h.
//
// Replaces the Qdrant collection the reference talks to through
// src/lattice/embeddings/client.py (QdrantManager): create / upsert / search / delete.
// One handle = one collection shard on one GPU.  gfx950 only; no fallback path exists:
// if HIP or the device is missing every entry point fails with CRH_E_HIP / CRH_E_NODEVICE.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "crh_common.h"
#include "crh_kernels.hpp"
#include "crh_i8.hpp"

namespace crh {
std::string &last_error_ref()
{
    thread_local std::string s;
    return s;
}
}  // namespace crh

using namespace crh;

namespace {

constexpr int kWaves = 16;      // waves per scan workgroup (one workgroup per CU)
constexpr int kRing = 8;        // 1-KiB loads in flight per wave
constexpr int kWideWaves = 8;   // waves per k_scan_wide workgroup: each sees every tile of its workgroup for 32 queries, so the
                                // workgroup's share of the candidate workspace is cut into 8 lists of 2 x wave_cap entries
constexpr int kI8Waves = 8;     // k_scan_i8: waves per workgroup, 1-KiB loads in flight per wave (crh_i8.hpp)
#ifndef CRH_I8_RING
#define CRH_I8_RING 8
#endif
constexpr int kI8Ring = CRH_I8_RING;
constexpr int kI8SelectParts = 4;   // workgroups per query in k_select behind the int8 scan (64 queries x 4 = the chip)
constexpr int kStatusSlots = 1024;
constexpr int64_t kWorkspaceBudget = 48LL << 30;

struct Pending {
    int nq, k, nfilt;
    crh_filter filt[CRH_MAX_FILTERS];
    int64_t row_base;
    const float *q_dev;
    float *out_s;
    int64_t *out_r;
    int slot;
    int path;       // how enqueue_batch nominated the batch's rows: CRH_NOMINATE_INT8 / _BF16 (one launch) / _BF16_3 (three launches, wide scan)
};

}  // namespace

struct crh_index {
    int dim = 0, ksteps = 0, dtype = 0, ncols = 0, device = 0, cu_count = 0;
    int batch_q = 64;  // queries per k_scan pass: 64, or 32 when the 64-query image would not fit LDS (dim 1536)
    bool wide_ok = false;  // k_scan_wide (up to 256 queries per corpus pass, query fragments in registers) exists for this dim
    bool fused_scan = true;   // <= batch_q queries: seed scan + threshold + main scan in one launch (CODERAG_HIP_FUSED_SCAN=0: three)
    // After a grid-wide wait timed out (another stream's kernels held CUs) the index runs the three-launch form for a WINDOW OF
    // TIME, not a number of batches: 0.2 s, doubled by every further time-out up to 5 s, back to 0.2 s once a one-launch batch
    // has gone through (round 3 rested 1024 batches: 2.5 s of queries at the bf16 rate for what is usually a transient cause).
    std::chrono::steady_clock::time_point fused_rest_until{};
    double fused_rest_s = 0.2;
    bool fused_resting() const { return std::chrono::steady_clock::now() < fused_rest_until; }
    // int8 nomination copy (crh_i8.hpp): derived from xt, brought up to date before a scan (i8_sync); tiles >= i8_dirty_from are stale
    bool i8 = false;          // <= batch_q queries are nominated from the copy (dim 384 / 768 / 1536; CODERAG_HIP_I8=0: never)
    u32x4 *x8 = nullptr;
    float *srow = nullptr;
    u32x4 *xrow = nullptr;    // row-major bf16 rows beside the copy (bf16 stores; crh_i8.hpp, ROW-MAJOR ROWS): what k_select's single-row reads use
    bool want_xrow = true;    // (CODERAG_HIP_ROWMAJOR=0: never -- the single-row reads then go to the tiles, as in round 3)
    unsigned int *i8stat = nullptr;
    int64_t x8_cap_tiles = 0, i8_dirty_from = 0;
    int i8_strikes = 0;       // consecutive int8-nominated batches whose candidate buffers overflowed (3: the copy is left unused ...
    int i8_cooldown = 0;      // ... for this many batches; then it gets ONE more try, and the next overflow rests it again)
    bool i8_suppress = false; // (while such a batch is run again on the bf16 scan)
    int nominate_max = CRH_NOMINATE_INT8;   // crh_index_set_nomination: the most advanced mode the caller allows
    bool i8_sample_record = true;           // the sample launch records its tiles' upper ends, the pass does not read those tiles again (CODERAG_HIP_I8_SAMPLE_RECORD=0: it does)
    bool i8_sample_auto = true;             // (CODERAG_HIP_I8_SAMPLE set: that many tiles at every size)
    int i8_sample = kI8SampleTiles;         // sample tiles behind the int8 scan's thresholds: 8192 halves the candidates of 4096 for 100 MB more
                                            // sample reads (-22 us per batch on one index, tools/sample_ab.py); CODERAG_HIP_I8_SAMPLE
    int64_t i8_min_rows = 1000000;          // below this the pass is too short for the copy to pay on every kind of data: Gaussian rows gain
                                            // from 100 k rows up (0.134 against 0.147 ms per batch; 0.285 / 0.360 at 1M), rows around one shared
                                            // mean with isotropic noise -- the widest candidate sets -- only from ~1.5M (1M: 0.397 against 0.358;
                                            // 2M: 0.554 / 0.597; profiles/r04_i8_crossover.txt); CODERAG_HIP_I8_MIN_ROWS
    int64_t cap_rows = 0, cap_tiles = 0, count = 0, alive_count = 0;
    // The validity mask of the last filter is kept while nothing it was built from has changed (rows, alive bits, codes: every
    // mutation bumps `mutations`): the reference's searchers send the same equality filter with query after query (project_name,
    // language: query/vector_search.py:83-93), and rebuilding the mask is a pass over the code columns per batch (~20 us at 10M rows).
    uint64_t mutations = 1, mask_built_at = 0;
    hipStream_t mask_stream = nullptr;   // (the stream the kept mask was built on: another stream rebuilds it -- nothing orders the two)
    int mask_nfilt = -1;
    crh_filter mask_filt[CRH_MAX_FILTERS];
    u32x4 *xt = nullptr;
    float *xf32 = nullptr;
    uint32_t *alive = nullptr;
    int32_t *codes = nullptr;
    unsigned int *scratch_u32 = nullptr;

    // tuning
    int seed_tiles = 4096, wave_cap = 2048, qcap = 65536, force_fallback = 0;

    // search workspace (lazily sized): everything a batch writes between its query preparation and its final selection
    struct Workspace {
        int ws_blocks = 0, ws_wave_cap = 0, ws_qcap = 0, ws_seed = 0;
        int64_t ws_mask_tiles = 0;
        float *qn = nullptr, *gmax = nullptr, *tau = nullptr, *qpar = nullptr, *qlo = nullptr;
        u32x4 *qfrag = nullptr, *qfrag8 = nullptr, *wave_lists = nullptr, *shi = nullptr;
        uint32_t *effmask = nullptr;
        u32x2 *qlist = nullptr;
        unsigned long long *skeys = nullptr, *skeys2 = nullptr;
    };
    Workspace ws;
    SearchStatus *status = nullptr;
    float *stage_q = nullptr;
    int64_t stage_q_elems = 0;
    float *stage_os = nullptr;
    int64_t *stage_or = nullptr;
    int64_t stage_out_elems = 0;
    void *stage_in = nullptr;
    int64_t stage_in_bytes = 0;

    std::vector<Pending> pending;
    int next_slot = 0;
    crh_search_stats stats{};

    // optional HIP-event timing of the dominant kernel (bench.py's roofline figure)
    bool profiling = false;
    bool profile_whole_scan = false;   // (crh_index_set_profiling(h, 2): the events bracket the int8 scan's three launches, not the pass alone)
    std::vector<hipEvent_t> ev;  // 2 per status slot: before / after the main scan launch
    double prof_scan_ms = 0.0;
    int64_t prof_scan_launches = 0;
};

namespace {

template <typename T>
int dev_alloc(T **p, int64_t elems)
{
    *p = nullptr;
    if (elems <= 0) return CRH_OK;
    CRH_HIP(hipMalloc(reinterpret_cast<void **>(p), (size_t)elems * sizeof(T)));
    return CRH_OK;
}
template <typename T>
void dev_free(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

float margin_for(const crh_index *h)
{
    // bound on |MFMA bf16 score - canonical score| for unit vectors, times 2 ("margin"):
    //  bf16 store: both sums run over the same exact products; f32 accumulation error <= 768*2^-24 each
    //  f32 store : + rounding q and x to bf16 for the scan, <= 2*2^-9 + 2^-18 by Cauchy-Schwarz
    //  (wider rows sum more terms: the bound scales with dim; 1.5e-4 = 1.6 x (2 * 768 * 2^-24) is kept up to dim 768)
    const float acc = 1.5e-4f * (h->dim > 768 ? (float)h->dim / 768.f : 1.f);
    return h->dtype == CRH_DTYPE_BF16 ? 2.f * acc : 2.f * (acc + 3.92e-3f);
}

int ensure_stage_in(crh_index *h, int64_t bytes)
{
    if (h->stage_in_bytes >= bytes) return CRH_OK;
    if (h->stage_in) (void)hipFree(h->stage_in);
    h->stage_in = nullptr;
    h->stage_in_bytes = 0;   // (sizes are cleared before every reallocation below too: a failed hipMalloc must not leave a
                             // recorded size next to a NULL pointer for the next call to trust)
    CRH_HIP(hipMalloc(&h->stage_in, (size_t)bytes));
    h->stage_in_bytes = bytes;
    return CRH_OK;
}

int scan_blocks(const crh_index *h, int64_t nitems)
{
    int64_t b = ceil_div(nitems, kWaves);
    if (b > h->cu_count) b = h->cu_count;
    if (b < 1) b = 1;
    return (int)b;
}

// (re)allocate the search workspace for the given candidate capacities
int ensure_workspace(crh_index *h, crh_index::Workspace &w, int wave_cap, int qcap)
{
    const int blocks = h->cu_count;
    if (!w.qn) CRH_TRY(dev_alloc(&w.qn, (int64_t)kWideQ * h->dim));
    if (!w.qfrag) CRH_TRY(dev_alloc(&w.qfrag, (int64_t)(kWideQ / 32) * h->ksteps * 64));
    if (!w.tau) CRH_TRY(dev_alloc(&w.tau, kWideQ));
    if (h->i8 && !w.qfrag8) CRH_TRY(dev_alloc(&w.qfrag8, (int64_t)(kMaxQ / 32) * 2 * (h->dim / 32) * 64));
    if (h->i8 && !w.qpar) CRH_TRY(dev_alloc(&w.qpar, kMaxQ * 4));
    if (!h->status) {
        CRH_TRY(dev_alloc(&h->status, kStatusSlots));
        CRH_HIP(hipMemset(h->status, 0, sizeof(SearchStatus) * kStatusSlots));
    }
    if (w.ws_seed < h->seed_tiles) {
        dev_free(w.gmax);
        w.ws_seed = 0;
        CRH_TRY(dev_alloc(&w.gmax, (int64_t)h->seed_tiles * kWideQ));
        w.ws_seed = h->seed_tiles;
    }
    if (w.ws_mask_tiles < h->cap_tiles) {
        dev_free(w.effmask);
        w.ws_mask_tiles = 0;
        h->mask_built_at = 0;
        CRH_TRY(dev_alloc(&w.effmask, h->cap_tiles));
        w.ws_mask_tiles = h->cap_tiles;
    }
    if (w.ws_blocks != blocks || w.ws_wave_cap != wave_cap) {
        const int64_t bytes = (int64_t)blocks * kWaves * wave_cap * 16;
        if (bytes > kWorkspaceBudget) return fail(CRH_E_CAPACITY, "candidate workspace of %lld bytes exceeds the budget", (long long)bytes);
        dev_free(w.wave_lists);
        w.ws_blocks = w.ws_wave_cap = 0;
        CRH_TRY(dev_alloc(&w.wave_lists, (int64_t)blocks * kWaves * wave_cap));
        w.ws_blocks = blocks;
        w.ws_wave_cap = wave_cap;
    }
    if (w.ws_qcap != qcap) {
        const int64_t bytes = (int64_t)kWideQ * qcap * 16;
        if (bytes > kWorkspaceBudget) return fail(CRH_E_CAPACITY, "per-query candidate lists of %lld bytes exceed the budget", (long long)bytes);
        dev_free(w.qlist);
        dev_free(w.skeys);
        dev_free(w.qlo);
        dev_free(w.skeys2);
        w.ws_qcap = 0;
        CRH_TRY(dev_alloc(&w.qlist, (int64_t)kWideQ * qcap));
        CRH_TRY(dev_alloc(&w.skeys, (int64_t)kWideQ * qcap));
        if (h->i8) CRH_TRY(dev_alloc(&w.qlo, (int64_t)kMaxQ * qcap));
        if (h->i8) CRH_TRY(dev_alloc(&w.skeys2, (int64_t)kMaxQ * qcap));
        w.ws_qcap = qcap;
    }
    return CRH_OK;
}
// the workspace at the current tuning: what the entry points outside crh_search work in
int ensure_workspace0(crh_index *h)
{
    crh_index::Workspace &w = h->ws;
    return ensure_workspace(h, w, std::max(h->wave_cap, w.ws_wave_cap), std::max(h->qcap, w.ws_qcap));
}

__global__ void k_fill_pad(float *s, int64_t *r, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        s[i] = -INFINITY;
        r[i] = -1;
    }
}

int build_mask(crh_index *h, crh_index::Workspace &w, const crh_filter *filters, int nfilt, const uint32_t **mask_out, hipStream_t st)
{
    if (nfilt == 0 || h->count == 0) {   // (an empty index: nothing to mask, and a zero-block launch is an error)
        *mask_out = h->alive;
        return CRH_OK;
    }
    FilterSet fs;
    fs.n = nfilt;
    for (int f = 0; f < nfilt; ++f) {
        if (filters[f].col < 0 || filters[f].col >= h->ncols)
            return fail(CRH_E_INVALID, "filter column %d out of range (index has %d code columns)", filters[f].col, h->ncols);
        fs.col[f] = filters[f].col;
        fs.code[f] = filters[f].code;
    }
    const bool same = h->mask_built_at == h->mutations && h->mask_stream == st && h->mask_nfilt == nfilt &&
                      memcmp(h->mask_filt, filters, sizeof(crh_filter) * (size_t)nfilt) == 0;
    if (!same) {
        const int64_t rows = (h->count + 63) & ~63LL;
        hipLaunchKernelGGL(k_filter_mask, dim3((unsigned)ceil_div(rows, 256)), dim3(256), 0, st, h->alive, h->codes, h->cap_rows,
                           h->count, fs, w.effmask);
        CRH_HIP(hipGetLastError());
        h->mask_built_at = h->mutations;
        h->mask_stream = st;
        h->mask_nfilt = nfilt;
        memcpy(h->mask_filt, filters, sizeof(crh_filter) * (size_t)nfilt);
    }
    *mask_out = w.effmask;
    return CRH_OK;
}

// scan kernel instantiations: k-steps = dim / 16; 64 queries per pass except for dim 1536 (32: LDS)
template <int MODE>
int launch_scan(crh_index *h, crh_index::Workspace &w, int blocks, hipStream_t st, const uint32_t *mask, int nitems, int stride, int wave_cap, int qcap,
                SearchStatus *stt)
{
#define CRH_SCAN(KS, QB)                                                                                                       \
    hipLaunchKernelGGL((k_scan<KS, MODE, kWaves, kRing, QB>), dim3(blocks), dim3(kWaves * 64), 0, st, h->xt, w.qfrag, w.tau, \
                       mask, nitems, stride, w.gmax, w.wave_lists, wave_cap, stt->qcount, w.qlist, qcap, stt)
    switch (h->ksteps) {
    case 24: CRH_SCAN(24, 2); break;
    case 48: CRH_SCAN(48, 2); break;
    case 64: CRH_SCAN(64, 2); break;
    case 96: CRH_SCAN(96, 1); break;
    default: return fail(CRH_E_INTERNAL, "no scan kernel for %d k-steps", h->ksteps);
    }
#undef CRH_SCAN
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

// What a one-launch scan may spend in ONE grid-wide wait, in 100 MHz ticks: eight times the time its pass over `bytes` should take
// (priced at 4 TB/s, two thirds of what the scans reach), at least 2 ms.  A wait ends when every workgroup has become resident and
// done its share of the phase before it; on an otherwise idle device that is microseconds, beside another stream's kernels it is
// that kernel's remaining time -- and beyond this bound the three-launch form, which needs nobody resident, is the better answer.
unsigned int wait_ticks_for(double bytes)
{
    const double ticks = 8.0 * bytes / 4.0e12 * 1.0e8;
    return (unsigned int)std::min(4.0e9, std::max(2.0e5, ticks));
}

// seed scan + threshold + main scan in one launch (k_scan_fused): <= batch_q queries, the default sample size
int launch_scan_fused(crh_index *h, crh_index::Workspace &w, int blocks, hipStream_t st, const uint32_t *mask, int ntiles, int G, int S, int k, float margin,
                      int nq, int wave_cap, int qcap, SearchStatus *stt)
{
#define CRH_FUSED(KS, QB)                                                                                                          \
    hipLaunchKernelGGL((k_scan_fused<KS, kWaves, kRing, QB>), dim3(blocks), dim3(kWaves * 64), 0, st, h->xt, w.qfrag, mask, ntiles, G, S, \
                       w.gmax, w.tau, k, margin, nq, w.wave_lists, wave_cap, stt->qcount, w.qlist, qcap, stt, h->force_fallback == 2 ? 1 : 0, \
                       wait_ticks_for((double)ntiles * h->ksteps * 1024.0))
    switch (h->ksteps) {
    case 24: CRH_FUSED(24, 2); break;
    case 48: CRH_FUSED(48, 2); break;
    case 96: CRH_FUSED(96, 1); break;
    default: return fail(CRH_E_INTERNAL, "no fused scan kernel for %d k-steps", h->ksteps);   // (dim 1024: its 128 KB query image leaves no LDS for the threshold step)
    }
#undef CRH_FUSED
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

// ---- int8 nomination (crh_i8.hpp)
constexpr int kI8MaxK = 256;   // beyond this k the threshold sits so low that the int8 intervals nominate several 100 k rows per query
bool i8_use(const crh_index *h, int nq, int k)
{
    // (three launches, no grid-wide wait: an index resting from a time-out of the one-launch bf16 scan still nominates from the copy)
    return k <= kI8MaxK && h->i8 && h->nominate_max >= CRH_NOMINATE_INT8 && !h->i8_suppress && h->i8_cooldown == 0 && nq <= h->batch_q && h->fused_scan &&
           (h->seed_tiles == 4096 || h->seed_tiles == kI8SampleTiles) && h->count >= h->i8_min_rows;
}

// the copy covers the index: (re)allocate with the capacity, requantise the tiles touched since the last scan.  Running out of
// memory for the copy is not an error: the index goes on with the bf16 scan.
int i8_alloc(crh_index *h)
{
    if (!h->i8 || h->x8_cap_tiles >= h->cap_tiles) return CRH_OK;
    const int ks8 = h->dim / 32;
    dev_free(h->x8);
    dev_free(h->srow);
    dev_free(h->xrow);
    h->x8_cap_tiles = 0;
    bool ok = hipMalloc(reinterpret_cast<void **>(&h->x8), (size_t)h->cap_tiles * ks8 * 1024) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&h->srow), (size_t)h->cap_tiles * 32 * sizeof(float)) == hipSuccess;
    if (ok && !h->i8stat)
        ok = hipMalloc(reinterpret_cast<void **>(&h->i8stat), 16) == hipSuccess && hipMemset(h->i8stat, 0, 16) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        dev_free(h->x8);
        dev_free(h->srow);
        h->i8 = false;
        return CRH_OK;
    }
    // the row-major rows are an accelerator of the selection step only: no memory for them -> the tiles are read instead
    if (h->want_xrow && h->dtype == CRH_DTYPE_BF16 &&
        hipMalloc(reinterpret_cast<void **>(&h->xrow), (size_t)h->cap_tiles * 32 * h->dim * 2) != hipSuccess) {
        (void)hipGetLastError();
        h->xrow = nullptr;
    }
    h->x8_cap_tiles = h->cap_tiles;
    h->i8_dirty_from = 0;
    return CRH_OK;
}

int i8_sync(crh_index *h, hipStream_t st)
{
    if (!h->i8) return CRH_OK;
    const int64_t ntiles = ceil_div(h->count, kTileRows);
    CRH_TRY(i8_alloc(h));
    if (!h->i8) return CRH_OK;
    if (h->i8_dirty_from < ntiles) {
        if (h->dtype == CRH_DTYPE_F32)
            hipLaunchKernelGGL(k_requant_i8<true>, dim3((unsigned)(ntiles - h->i8_dirty_from)), dim3(64), 0, st, h->xt, h->xf32, h->x8, h->srow,
                               h->i8stat, h->i8_dirty_from, h->ksteps, h->count, (u32x4 *)nullptr);
        else
            hipLaunchKernelGGL(k_requant_i8<false>, dim3((unsigned)(ntiles - h->i8_dirty_from)), dim3(64), 0, st, h->xt, h->xf32, h->x8, h->srow,
                               h->i8stat, h->i8_dirty_from, h->ksteps, h->count, h->xrow);
        CRH_HIP(hipGetLastError());
        h->i8_dirty_from = ntiles;
    }
    return CRH_OK;
}

// PART 1: the sample tiles, 2: the thresholds (one workgroup per query), 3: the pass -- three launches, stream order between them
template <int PART>
int launch_scan_i8(crh_index *h, crh_index::Workspace &w, int blocks, hipStream_t st, const uint32_t *mask, int ntiles, int G, int S, int k, float c_abs,
                   int nq, int wave_cap, int qcap, SearchStatus *stt, u32x4 *shi)
{
#define CRH_I8(KS8, RING, QB)                                                                                                          \
    hipLaunchKernelGGL((k_scan_i8<KS8, kI8Waves, RING, QB, PART>), dim3(PART == 2 ? QB * 32 : blocks), dim3(kI8Waves * 64), 0, st, h->x8, h->srow, \
                       h->i8stat, w.qfrag8, w.qpar, mask, ntiles, G, S, reinterpret_cast<uint32_t *>(w.gmax), w.tau, k, c_abs, sqrtf((float)h->dim), \
                       nq, w.wave_lists, wave_cap, stt->qcount, w.qlist, w.qlo, qcap, stt, h->xt,                                               \
                       h->dtype == CRH_DTYPE_F32 ? h->xf32 : (const float *)nullptr, w.qn, h->xrow, PART == 2 ? (u32x4 *)nullptr : shi)
    switch (h->dim) {
    case 384: CRH_I8(12, 12, 2); break;
    case 768: CRH_I8(24, kI8Ring, 2); break;
    case 1024: CRH_I8(32, kI8Ring, 2); break;   // (128 KB of query image: fits since the threshold step is a launch of its own)
    case 1536: CRH_I8(48, kI8Ring, 1); break;
    default: return fail(CRH_E_INTERNAL, "no int8 scan kernel for dim %d", h->dim);
    }
#undef CRH_I8
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

// the wide scan: dim 384 / 768 (the query block of a wave must fit its registers)
template <int MODE>
int launch_scan_wide(crh_index *h, crh_index::Workspace &w, hipStream_t st, const uint32_t *mask, int nitems, int stride, int nblk, int wave_cap, int qcap,
                     SearchStatus *stt)
{
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(nitems, h->cu_count));
#define CRH_WIDE(KS)                                                                                                              \
    hipLaunchKernelGGL((k_scan_wide<KS, MODE>), dim3(blocks), dim3(512), 0, st, h->xt, w.qfrag, w.tau, mask, nitems, stride, nblk, \
                       w.gmax, kWideQ, w.wave_lists, kWideWaves, wave_cap * (kWaves / kWideWaves), stt->qcount, w.qlist, qcap, stt)
    switch (h->ksteps) {
    case 24: CRH_WIDE(24); break;
    case 48: CRH_WIDE(48); break;
    default: return fail(CRH_E_INTERNAL, "no wide scan kernel for %d k-steps", h->ksteps);
    }
#undef CRH_WIDE
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

// one batch (<= batch_q queries through k_scan, or up to kWideQ through k_scan_wide), everything enqueued on `st`
int enqueue_batch(crh_index *h, crh_index::Workspace &w, const float *q_dev, int nq, int k, const uint32_t *mask, int64_t row_base, float *out_s,
                  int64_t *out_r, int slot, hipStream_t st, int *path_out)
{
    *path_out = CRH_NOMINATE_BF16_3;
    const int64_t ntiles = ceil_div(h->count, kTileRows);
    if (ntiles == 0) {
        CRH_HIP(hipMemsetAsync(h->status + slot, 0, sizeof(SearchStatus), st));
        const int64_t n = (int64_t)nq * k;
        hipLaunchKernelGGL(k_fill_pad, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, out_s, out_r, n);
        CRH_HIP(hipGetLastError());
        return CRH_OK;
    }
    const int wave_cap = w.ws_wave_cap, qcap = w.ws_qcap;
    const float margin = margin_for(h);
    SearchStatus *stt = h->status + slot;
    const bool wide = nq > h->batch_q;
    if (wide && (!h->wide_ok || nq > kWideQ)) return fail(CRH_E_INTERNAL, "batch of %d queries has no scan kernel", nq);
    const int nblk = (nq + 31) / 32;                       // 32-query blocks in use (wide scan)
    const int width = wide ? nblk * 32 : h->batch_q;       // query slots prepared (slots >= nq are zero queries, tau = +inf)
    const int qstride = wide ? kWideQ : 64;                // row pitch of the seed maxima

    // <= batch_q queries, nominated from the int8 copy (crh_i8.hpp): half the bytes of the pass, same results
    if (i8_use(h, nq, k)) {
        CRH_TRY(i8_sync(h, st));
    }
    const bool via_i8 = i8_use(h, nq, k);   // (i8_sync may have given the copy up for lack of memory)
    if (via_i8) {    // one launch prepares the bf16 image, the canonical queries and the integer images
        if (h->dtype == CRH_DTYPE_BF16)
            hipLaunchKernelGGL((k_prep_queries<true, true>), dim3(width), dim3(64), 0, st, q_dev, nq, h->dim, h->ksteps, w.qn, w.qfrag, stt, w.qfrag8, w.qpar);
        else
            hipLaunchKernelGGL((k_prep_queries<false, true>), dim3(width), dim3(64), 0, st, q_dev, nq, h->dim, h->ksteps, w.qn, w.qfrag, stt, w.qfrag8, w.qpar);
    } else if (h->dtype == CRH_DTYPE_BF16) {
        hipLaunchKernelGGL((k_prep_queries<true, false>), dim3(width), dim3(64), 0, st, q_dev, nq, h->dim, h->ksteps, w.qn, w.qfrag, stt, (u32x4 *)nullptr, (float *)nullptr);
    } else {
        hipLaunchKernelGGL((k_prep_queries<false, false>), dim3(width), dim3(64), 0, st, q_dev, nq, h->dim, h->ksteps, w.qn, w.qfrag, stt, (u32x4 *)nullptr, (float *)nullptr);
    }
    CRH_HIP(hipGetLastError());

    if (via_i8) {
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(ntiles, kI8Waves), h->cu_count));
        // The sample is read twice (its own launch, then the pass) and buys the thresholds: its cost grows with G, the candidates
        // it leaves with 1 / G -- the best G goes with the square root of the corpus.  8192 tiles were tuned at 10M rows (2.6 % of
        // them); the same 8192 are 26 % of a 1M-row corpus (profiles/r04_i8_crossover.txt).  An explicit sample size is kept.
        int64_t g_auto = h->i8_sample;
        if (h->seed_tiles == 4096 && h->i8_sample_auto)
            g_auto = std::max<int64_t>(1024, std::min<int64_t>(h->i8_sample, (int64_t)(h->i8_sample * std::sqrt((double)ntiles / 312500.0))));
        const int G8 = (int)std::min<int64_t>(h->seed_tiles == 4096 ? g_auto : h->seed_tiles, ntiles);
        const int S8 = (int)std::max<int64_t>(1, ntiles / G8);
        // what separates a row's canonical score from the exact dot of the QUANTISED-FROM rows and the canonical query: the two f32
        // summation orders only -- the copy of an f32 store is quantised from its f32 master, not from the bf16 tiles
        const float c_abs = 1.5e-4f * (h->dim > 768 ? (float)h->dim / 768.f : 1.f) + 1e-5f;
        if (h->profiling && h->profile_whole_scan) CRH_HIP(hipEventRecord(h->ev[2 * slot], st));   // (all three launches of the scan)
        // the sample launch records the upper ends of its tiles' rows and the pass takes those tiles' candidates from the record
        // instead of reading and multiplying the tiles again (crh_i8.hpp, `shi`; CODERAG_HIP_I8_SAMPLE_RECORD=0: the pass reads every tile)
        u32x4 *shi = nullptr;
        if (h->i8_sample_record) {
            if (!w.shi) CRH_TRY(dev_alloc(&w.shi, (int64_t)kI8SampleTiles * 64 * 4));
            shi = w.shi;
        }
        CRH_TRY(launch_scan_i8<1>(h, w, blocks, st, mask, (int)ntiles, G8, S8, k, c_abs, nq, wave_cap * (kWaves / kI8Waves), qcap, stt, shi));
        CRH_TRY(launch_scan_i8<2>(h, w, blocks, st, mask, (int)ntiles, G8, S8, k, c_abs, nq, wave_cap * (kWaves / kI8Waves), qcap, stt, shi));
        if (h->profiling && !h->profile_whole_scan) CRH_HIP(hipEventRecord(h->ev[2 * slot], st));   // (the dominant kernel: the pass)
        CRH_TRY(launch_scan_i8<3>(h, w, blocks, st, mask, (int)ntiles, G8, S8, k, c_abs, nq, wave_cap * (kWaves / kI8Waves), qcap, stt, shi));
        if (h->profiling) CRH_HIP(hipEventRecord(h->ev[2 * slot + 1], st));
        // (k_select's margin behind this scan: twice what separates a row's score summed in any order from its canonical score)
        if (h->dtype == CRH_DTYPE_F32)
            hipLaunchKernelGGL((k_select<true, true>), dim3(nq, kI8SelectParts), dim3(1024), 0, st, w.qlist, w.qlo, stt->qcount, qcap, w.skeys, w.skeys2, w.qn, h->xt,
                               h->xf32, h->dim, h->ksteps, k, 2.f * c_abs, row_base, out_s, out_r, stt, (const u32x4 *)nullptr);
        else
            hipLaunchKernelGGL((k_select<false, true>), dim3(nq, kI8SelectParts), dim3(1024), 0, st, w.qlist, w.qlo, stt->qcount, qcap, w.skeys, w.skeys2, w.qn, h->xt,
                               h->xf32, h->dim, h->ksteps, k, 2.f * c_abs, row_base, out_s, out_r, stt, h->xrow);
        CRH_HIP(hipGetLastError());
        // the canonical chains of the few rows the fast scores leave, and the ranking (the kernel boundary is where a query's workgroups meet)
        if (h->dtype == CRH_DTYPE_F32)
            hipLaunchKernelGGL(k_select_final<true>, dim3(nq, kI8SelectParts), dim3(256), 0, st, w.skeys2, w.skeys, qcap, w.qn, h->xt, h->xf32, (const u32x4 *)nullptr,
                               h->dim, h->ksteps, k, 2.f * c_abs, row_base, out_s, out_r, stt);
        else
            hipLaunchKernelGGL(k_select_final<false>, dim3(nq, kI8SelectParts), dim3(256), 0, st, w.skeys2, w.skeys, qcap, w.qn, h->xt, h->xf32, h->xrow,
                               h->dim, h->ksteps, k, 2.f * c_abs, row_base, out_s, out_r, stt);
        CRH_HIP(hipGetLastError());
        h->stats.rows += h->count;
        h->stats.tiles += ntiles;
        h->stats.seed_tiles += G8;
        h->stats.batches += 1;
        *path_out = CRH_NOMINATE_INT8;
        return CRH_OK;
    }

    // <= batch_q queries at the default sample size: seed scan, threshold and main scan are ONE launch (k_scan_fused: every wave's
    // first tile is its sample tile, two grid-wide waits, the corpus read once).  The whole grid must be resident for those
    // waits: it is never larger than the CU count and a workgroup's LDS footprint leaves room for one per CU.
    if (!wide && h->fused_scan && !h->fused_resting() && h->nominate_max >= CRH_NOMINATE_BF16 && h->seed_tiles == 4096 && h->ksteps != 64) {
        if (h->i8_cooldown > 0 && !h->i8_suppress) h->i8_cooldown -= 1;
        const int blocks = scan_blocks(h, ntiles);
        const int waves = blocks * kWaves;
        const int Gf = (int)std::min<int64_t>(std::min(waves, 4096), ntiles);
        const int Sf = Gf == waves ? (int)std::max<int64_t>(1, ntiles / Gf) : 1;
        if (h->profiling) CRH_HIP(hipEventRecord(h->ev[2 * slot], st));
        CRH_TRY(launch_scan_fused(h, w, blocks, st, mask, (int)ntiles, Gf, Sf, k, margin, nq, wave_cap, qcap, stt));
        if (h->profiling) CRH_HIP(hipEventRecord(h->ev[2 * slot + 1], st));
        if (h->dtype == CRH_DTYPE_F32)
            hipLaunchKernelGGL(k_select<true>, dim3(nq), dim3(1024), 0, st, w.qlist, (const float *)nullptr, stt->qcount, qcap, w.skeys, (unsigned long long *)nullptr, w.qn, h->xt,
                               h->xf32, h->dim, h->ksteps, k, margin, row_base, out_s, out_r, stt);
        else
            hipLaunchKernelGGL(k_select<false>, dim3(nq), dim3(1024), 0, st, w.qlist, (const float *)nullptr, stt->qcount, qcap, w.skeys, (unsigned long long *)nullptr, w.qn, h->xt,
                               h->xf32, h->dim, h->ksteps, k, margin, row_base, out_s, out_r, stt);
        CRH_HIP(hipGetLastError());
        h->stats.rows += h->count;
        h->stats.tiles += ntiles;
        h->stats.seed_tiles += Gf;
        h->stats.batches += 1;
        *path_out = CRH_NOMINATE_BF16;
        return CRH_OK;
    }
    if (!wide && h->i8_cooldown > 0 && !h->i8_suppress) h->i8_cooldown -= 1;
    const int G = (int)std::min<int64_t>(h->seed_tiles, ntiles);
    const int stride = (int)(ntiles / G);
    if (wide)
        CRH_TRY(launch_scan_wide<0>(h, w, st, mask, G, stride, nblk, wave_cap, qcap, stt));
    else
        CRH_TRY(launch_scan<0>(h, w, scan_blocks(h, G), st, mask, G, stride, wave_cap, qcap, stt));
    hipLaunchKernelGGL(k_tau, dim3(width), dim3(256), (size_t)G * 4, st, w.gmax, G, k, margin, nq, w.tau, qstride);
    CRH_HIP(hipGetLastError());
    if (h->profiling) CRH_HIP(hipEventRecord(h->ev[2 * slot], st));
    if (wide)
        CRH_TRY(launch_scan_wide<1>(h, w, st, mask, (int)ntiles, 1, nblk, wave_cap, qcap, stt));
    else
        CRH_TRY(launch_scan<1>(h, w, scan_blocks(h, ntiles), st, mask, (int)ntiles, 1, wave_cap, qcap, stt));
    if (h->profiling) CRH_HIP(hipEventRecord(h->ev[2 * slot + 1], st));
    if (h->dtype == CRH_DTYPE_F32)
        hipLaunchKernelGGL(k_select<true>, dim3(nq), dim3(1024), 0, st, w.qlist, (const float *)nullptr, stt->qcount, qcap, w.skeys, (unsigned long long *)nullptr, w.qn, h->xt,
                           h->xf32, h->dim, h->ksteps, k, margin, row_base, out_s, out_r, stt);
    else
        hipLaunchKernelGGL(k_select<false>, dim3(nq), dim3(1024), 0, st, w.qlist, (const float *)nullptr, stt->qcount, qcap, w.skeys, (unsigned long long *)nullptr, w.qn, h->xt,
                           h->xf32, h->dim, h->ksteps, k, margin, row_base, out_s, out_r, stt);
    CRH_HIP(hipGetLastError());
    h->stats.rows += h->count;
    h->stats.tiles += ntiles;
    h->stats.seed_tiles += G;
    h->stats.batches += 1;
    return CRH_OK;
}

int next_pow2(int64_t v)
{
    int64_t p = 1;
    while (p < v) p <<= 1;
    return (int)std::min<int64_t>(p, 1LL << 30);
}

int finish_pending(crh_index *h, hipStream_t st)
{
    if (h->pending.empty()) return CRH_OK;
    CRH_HIP(hipStreamSynchronize(st));
    const int used = std::max(1, std::min(h->next_slot, kStatusSlots));   // slots are handed out in order from 0
    std::vector<SearchStatus> host((size_t)used);
    // (copies of the search path go through `st`, not through hipMemcpy: that one runs on the NULL stream and would wait for everything
    // another thread has queued there -- an encoder forward under way on torch's default stream held every answer back ~13 ms)
    CRH_HIP(hipMemcpyAsync(host.data(), h->status, sizeof(SearchStatus) * (size_t)used, hipMemcpyDeviceToHost, st));
    CRH_HIP(hipStreamSynchronize(st));
    std::vector<Pending> todo;
    todo.swap(h->pending);
    h->next_slot = 0;
    crh_index::Workspace &w = h->ws;   // (everything is idle: a batch that overflowed is re-run alone on `st`)
    for (const Pending &p : todo) {
        SearchStatus s = host[p.slot];
        if (h->profiling && h->count > 0) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, h->ev[2 * p.slot], h->ev[2 * p.slot + 1]) == hipSuccess) {
                h->prof_scan_ms += ms;
                h->prof_scan_launches += 1;
            }
        }
        int attempts = 0;
        int path = p.path;
        bool i8_regrown = false, no_i8 = false;   // no_i8: this batch has been sent to the bf16 scan -- for ALL its remaining attempts
        if (!(s.bar_timeout || s.wave_overflow || s.q_overflow)) {
            if (path == CRH_NOMINATE_INT8) h->i8_strikes = 0;
            if (path != CRH_NOMINATE_BF16_3) h->fused_rest_s = 0.2;   // a one-launch batch went through: the next time-out rests the short window again
        }
        while (s.bar_timeout || s.wave_overflow || s.q_overflow) {
            if (++attempts > 6) return fail(CRH_E_INTERNAL, "candidate buffers still overflow after %d regrowths", attempts - 1);
            int wc = w.ws_wave_cap, qc = w.ws_qcap;
            if (s.bar_timeout) {
                // A grid-wide wait of the one-launch scan gave up: some workgroup was not resident within the launch's bound (other
                // streams' kernels holding CUs).  The batch's results are void and its counts mean nothing (the thresholds were never
                // agreed on): the buffers stay as they are, the batch is run again in the three-launch form, which needs no
                // co-residency, and the index stays on that form for a window of time (crh_index::fused_rest_until).
                h->fused_rest_until = std::chrono::steady_clock::now() + std::chrono::duration_cast<std::chrono::steady_clock::duration>(
                                                                            std::chrono::duration<double>(h->fused_rest_s));
                h->fused_rest_s = std::min(5.0, h->fused_rest_s * 2.0);
                h->stats.fallback_used |= 2;
            } else if (path == CRH_NOMINATE_INT8) {
                // The int8 intervals of this data / this k nominated more rows than the buffers hold.  The true counts are known:
                // when they are moderate (at most 2 % of the rows per query: beyond that the selection behind the scan costs more
                // than the half pass it saves) the buffers grow to them ONCE and the batch runs again from the copy; otherwise,
                // or when that was not enough, the batch goes to the bf16 scan (whose buffers regrow if they must), and three such
                // batches in a row rest the copy for 4096 batches.
                const int64_t need_q = next_pow2((int64_t)s.max_qcount), need_w = next_pow2((int64_t)s.max_wave_cnt) / (kWaves / kI8Waves);   // (an int8 wave's list is kWaves / kI8Waves x wave_cap entries)
                const bool moderate = h->force_fallback != 3 && (int64_t)s.max_qcount <= std::max<int64_t>(h->count / 50, 4096) &&
                                      (int64_t)kWideQ * need_q * 16 * 2 <= kWorkspaceBudget / 4;
                if (!i8_regrown && moderate) {
                    i8_regrown = true;
                    h->stats.fallback_used |= 1;
                    wc = std::max(wc, (int)std::min<int64_t>(need_w, 1 << 20));
                    qc = std::max(qc, (int)need_q);
                    h->qcap = std::max(h->qcap, qc);          // (the data will not change its mind: later batches start from here)
                    h->wave_cap = std::max(h->wave_cap, wc);
                } else {
                    h->i8_strikes += 1;
                    if (h->i8_strikes >= 3) {
                        h->i8_strikes = 2;
                        h->i8_cooldown = 4096;
                    }
                    no_i8 = true;
                    h->stats.fallback_used |= 4;
                }
            } else {
                h->stats.fallback_used |= 1;
                wc = std::max(wc, next_pow2((int64_t)s.max_wave_cnt));
                qc = std::max(qc, next_pow2((int64_t)s.max_qcount));
            }
            CRH_TRY(ensure_workspace(h, w, wc, qc));
            const uint32_t *mask = nullptr;
            CRH_TRY(build_mask(h, w, p.filt, p.nfilt, &mask, st));
            h->i8_suppress = no_i8;
            const int rc = enqueue_batch(h, w, p.q_dev, p.nq, p.k, mask, p.row_base, p.out_s, p.out_r, 0, st, &path);
            h->i8_suppress = false;
            CRH_TRY(rc);
            CRH_HIP(hipMemcpyAsync(&s, h->status, sizeof(SearchStatus), hipMemcpyDeviceToHost, st));
            CRH_HIP(hipStreamSynchronize(st));
        }
        h->stats.candidates += (int64_t)s.candidates;
        h->stats.max_query_cands = std::max<int64_t>(h->stats.max_query_cands, s.max_qcount);
    }
    return CRH_OK;
}

}  // namespace

extern "C" {

int crh_abi_version(void) { return CRH_ABI_VERSION; }
const char *crh_last_error(void) { return last_error_ref().c_str(); }

int crh_device_count(int *count)
{
    if (!count) return fail(CRH_E_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(CRH_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return CRH_OK;
}

int crh_device_info(int device, char *name_out, int name_cap, char *arch_out, int arch_cap, int64_t *hbm_bytes_out,
                    int *cu_count_out)
{
    hipDeviceProp_t prop;
    CRH_HIP(hipGetDeviceProperties(&prop, device));
    if (name_out && name_cap > 0) snprintf(name_out, (size_t)name_cap, "%s", prop.name);
    if (arch_out && arch_cap > 0) snprintf(arch_out, (size_t)arch_cap, "%s", prop.gcnArchName);
    if (hbm_bytes_out) *hbm_bytes_out = (int64_t)prop.totalGlobalMem;
    if (cu_count_out) *cu_count_out = prop.multiProcessorCount;
    return CRH_OK;
}

int crh_index_create(int dim, int dtype, int64_t capacity_rows, int n_code_cols, int device, crh_index **out)
{
    if (!out) return fail(CRH_E_INVALID, "out is NULL");
    *out = nullptr;
    if (dim != 384 && dim != 768 && dim != 1024 && dim != 1536)
        return fail(CRH_E_INVALID, "dim %d not supported: scan kernels are instantiated for 384, 768 (UniXcoder, the tuned case), 1024 and 1536", dim);
    if (dtype != CRH_DTYPE_F32 && dtype != CRH_DTYPE_BF16) return fail(CRH_E_INVALID, "unknown dtype %d", dtype);
    if (capacity_rows <= 0 || capacity_rows > (1LL << 31)) return fail(CRH_E_INVALID, "capacity_rows %lld out of range", (long long)capacity_rows);
    if (n_code_cols < 0 || n_code_cols > 64) return fail(CRH_E_INVALID, "n_code_cols %d out of range", n_code_cols);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(CRH_E_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(CRH_E_INVALID, "device %d out of range (%d visible)", device, ndev);
    hipDeviceProp_t prop;
    CRH_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CRH_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    DeviceGuard g(device);
    if (!g.ok) return fail(CRH_E_HIP, "hipSetDevice(%d) failed", device);

    crh_index *h = new crh_index();
    h->dim = dim;
    h->ksteps = dim / 16;
    h->batch_q = dim > 1024 ? 32 : 64;   // (k_scan's 64-query image does not fit LDS at dim 1536)
    h->wide_ok = dim <= 768 && getenv("CODERAG_HIP_NO_WIDE_SCAN") == nullptr;   // (the env switch exists for A/B timing only)
    {
        const char *e = getenv("CODERAG_HIP_FUSED_SCAN");
        h->fused_scan = !(e && e[0] == '0');
        const char *e8 = getenv("CODERAG_HIP_I8");
        h->i8 = h->fused_scan && !(e8 && e8[0] == '0');
        if (h->i8) h->qcap = 131072;   // ~21 k candidates per query and 10M Gaussian rows behind the int8 scan (37 k with 4096 sample tiles)
        if (const char *em = getenv("CODERAG_HIP_I8_MIN_ROWS")) h->i8_min_rows = atoll(em);
        if (const char *er = getenv("CODERAG_HIP_I8_SAMPLE_RECORD")) h->i8_sample_record = !(er[0] == '0');
        if (const char *es = getenv("CODERAG_HIP_I8_SAMPLE")) {
            h->i8_sample = std::max(1024, std::min(kI8SampleTiles, atoi(es)));
            h->i8_sample_auto = false;
        }
        if (const char *er = getenv("CODERAG_HIP_ROWMAJOR")) h->want_xrow = er[0] != '0';
    }
    h->dtype = dtype;
    h->ncols = n_code_cols;
    h->device = device;
    h->cu_count = prop.multiProcessorCount;
    h->cap_rows = (capacity_rows + 31) & ~31LL;
    h->cap_tiles = h->cap_rows / 32;
    int rc = dev_alloc(&h->xt, h->cap_tiles * h->ksteps * 64);
    if (rc == CRH_OK && dtype == CRH_DTYPE_F32) rc = dev_alloc(&h->xf32, h->cap_rows * dim);
    if (rc == CRH_OK) rc = dev_alloc(&h->alive, h->cap_tiles);
    if (rc == CRH_OK && n_code_cols) rc = dev_alloc(&h->codes, h->cap_rows * n_code_cols);
    if (rc == CRH_OK) rc = dev_alloc(&h->scratch_u32, 4);
    if (rc == CRH_OK && hipMemset(h->xt, 0, (size_t)h->cap_tiles * h->ksteps * 1024) != hipSuccess) rc = fail(CRH_E_HIP, "hipMemset(xt) failed");
    if (rc == CRH_OK && hipMemset(h->alive, 0, (size_t)h->cap_tiles * 4) != hipSuccess) rc = fail(CRH_E_HIP, "hipMemset(alive) failed");
    if (rc == CRH_OK && h->codes && hipMemset(h->codes, 0xff, (size_t)h->cap_rows * n_code_cols * 4) != hipSuccess)
        rc = fail(CRH_E_HIP, "hipMemset(codes) failed");
    // an index created for >= i8_min_rows rows takes the memory of its int8 copy now, next to the rows (one large, early
    // allocation: the pass over a copy allocated late, between a framework's cached blocks, measured ~5 % slower in some runs)
    if (rc == CRH_OK && h->i8 && h->cap_rows >= h->i8_min_rows) rc = i8_alloc(h);
    if (rc != CRH_OK) {
        crh_index_destroy(h);
        return rc;
    }
    *out = h;
    return CRH_OK;
}

int crh_index_destroy(crh_index *h)
{
    if (!h) return CRH_OK;
    DeviceGuard g(h->device);
    (void)hipDeviceSynchronize();
    dev_free(h->xt);
    dev_free(h->xf32);
    dev_free(h->alive);
    dev_free(h->codes);
    dev_free(h->scratch_u32);
    {
        crh_index::Workspace &w = h->ws;
        dev_free(w.qn);
        dev_free(w.gmax);
        dev_free(w.tau);
        dev_free(w.qfrag);
        dev_free(w.wave_lists);
        dev_free(w.effmask);
        dev_free(w.qlist);
        dev_free(w.skeys);
        dev_free(w.qfrag8);
        dev_free(w.shi);
        dev_free(w.qpar);
        dev_free(w.qlo);
        dev_free(w.skeys2);
    }
    dev_free(h->x8);
    dev_free(h->srow);
    dev_free(h->xrow);
    dev_free(h->i8stat);
    dev_free(h->status);
    dev_free(h->stage_q);
    dev_free(h->stage_os);
    dev_free(h->stage_or);
    if (h->stage_in) (void)hipFree(h->stage_in);
    for (auto &e : h->ev) (void)hipEventDestroy(e);
    delete h;
    return CRH_OK;
}

static int append_impl(crh_index *h, int64_t n, const float *vecs, int on_device, const int32_t *codes, int64_t *first_row_out,
                       void *stream, int preprocessed);

int crh_index_append(crh_index *h, int64_t n, const float *vecs, int on_device, const int32_t *codes, int64_t *first_row_out,
                     void *stream)
{
    return append_impl(h, n, vecs, on_device, codes, first_row_out, stream, 0);
}

int crh_index_append_preprocessed(crh_index *h, int64_t n, const float *vecs, int on_device, const int32_t *codes,
                                  int64_t *first_row_out, void *stream)
{
    return append_impl(h, n, vecs, on_device, codes, first_row_out, stream, 1);
}

static int append_impl(crh_index *h, int64_t n, const float *vecs, int on_device, const int32_t *codes, int64_t *first_row_out,
                       void *stream, int preprocessed)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    if (n < 0) return fail(CRH_E_INVALID, "n < 0");
    if (first_row_out) *first_row_out = h->count;
    if (n == 0) return CRH_OK;
    if (!vecs) return fail(CRH_E_INVALID, "vecs is NULL");
    if (h->ncols > 0 && !codes) return fail(CRH_E_INVALID, "index has %d code columns but codes is NULL", h->ncols);
    if (h->count + n > h->cap_rows)
        return fail(CRH_E_CAPACITY, "append of %lld rows exceeds capacity (%lld of %lld used)", (long long)n, (long long)h->count,
                    (long long)h->cap_rows);
    DeviceGuard g(h->device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t chunk = 65536;
    for (int64_t off = 0; off < n; off += chunk) {
        const int64_t m = std::min(chunk, n - off);
        const float *src = vecs + off * h->dim;
        const int32_t *csrc = codes ? codes + off * h->ncols : nullptr;
        if (!on_device) {
            const int64_t vb = m * h->dim * 4, cb = m * h->ncols * 4;
            CRH_TRY(ensure_stage_in(h, vb + cb));
            CRH_HIP(hipMemcpyAsync(h->stage_in, src, (size_t)vb, hipMemcpyHostToDevice, st));
            src = static_cast<const float *>(h->stage_in);
            if (csrc) {
                void *cd = static_cast<char *>(h->stage_in) + vb;
                CRH_HIP(hipMemcpyAsync(cd, csrc, (size_t)cb, hipMemcpyHostToDevice, st));
                csrc = static_cast<const int32_t *>(cd);
            }
        }
        const int64_t first = h->count + off;
        const unsigned blocks = (unsigned)ceil_div(m, 32);
        if (h->dtype == CRH_DTYPE_F32)
            hipLaunchKernelGGL(k_append<true>, dim3(blocks), dim3(256), 0, st, src, m, first, h->dim, h->ksteps, h->xt, h->xf32, preprocessed);
        else
            hipLaunchKernelGGL(k_append<false>, dim3(blocks), dim3(256), 0, st, src, m, first, h->dim, h->ksteps, h->xt, h->xf32, preprocessed);
        CRH_HIP(hipGetLastError());
        if (csrc) {
            hipLaunchKernelGGL(k_store_codes, dim3((unsigned)ceil_div(m * h->ncols, 256)), dim3(256), 0, st, csrc, m, h->ncols, first,
                               h->cap_rows, h->codes);
            CRH_HIP(hipGetLastError());
        }
        const int64_t ntl = ((first + m - 1) >> 5) - (first >> 5) + 1;
        hipLaunchKernelGGL(k_set_alive, dim3((unsigned)ceil_div(ntl, 256)), dim3(256), 0, st, h->alive, first, m);
        CRH_HIP(hipGetLastError());
        if (!on_device) CRH_HIP(hipStreamSynchronize(st));  // the staging buffer is reused by the next chunk
    }
    h->i8_dirty_from = std::min<int64_t>(h->i8_dirty_from, h->count / kTileRows);   // (the last tile may have been partly filled)
    h->count += n;
    h->alive_count += n;
    h->mutations += 1;
    return CRH_OK;
}

int crh_index_tombstone(crh_index *h, int64_t n, const int64_t *rows)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    if (n <= 0) return CRH_OK;
    if (!rows) return fail(CRH_E_INVALID, "rows is NULL");
    DeviceGuard g(h->device);
    CRH_TRY(ensure_stage_in(h, n * 8));
    CRH_HIP(hipMemcpy(h->stage_in, rows, (size_t)n * 8, hipMemcpyHostToDevice));
    CRH_HIP(hipMemset(h->scratch_u32, 0, 4));
    hipLaunchKernelGGL(k_tombstone, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, 0, h->alive,
                       static_cast<const int64_t *>(h->stage_in), n, h->count, h->scratch_u32);
    CRH_HIP(hipGetLastError());
    unsigned int cleared = 0;
    CRH_HIP(hipMemcpy(&cleared, h->scratch_u32, 4, hipMemcpyDeviceToHost));
    h->alive_count -= cleared;
    h->mutations += 1;
    return CRH_OK;
}

int crh_index_tombstone_filter(crh_index *h, const crh_filter *filters, int n_filters, int64_t *n_cleared_out)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    if (n_cleared_out) *n_cleared_out = 0;
    if (n_filters <= 0 || n_filters > CRH_MAX_FILTERS) return fail(CRH_E_INVALID, "n_filters=%d outside 1..%d (a delete needs a filter)", n_filters, CRH_MAX_FILTERS);
    if (!filters) return fail(CRH_E_INVALID, "filters is NULL");
    if (h->count == 0) return CRH_OK;
    DeviceGuard g(h->device);
    CRH_HIP(hipDeviceSynchronize());   // searches in flight on other streams read `alive` / the shared mask buffer
    CRH_TRY(ensure_workspace0(h));
    const uint32_t *mask = nullptr;
    CRH_TRY(build_mask(h, h->ws, filters, n_filters, &mask, nullptr));
    const int64_t ntiles = ceil_div(h->count, 32);
    CRH_HIP(hipMemset(h->scratch_u32, 0, 4));
    hipLaunchKernelGGL(k_tombstone_mask, dim3((unsigned)ceil_div(ntiles, 256)), dim3(256), 0, 0, h->alive, mask, ntiles, h->scratch_u32);
    CRH_HIP(hipGetLastError());
    unsigned int cleared = 0;
    CRH_HIP(hipMemcpy(&cleared, h->scratch_u32, 4, hipMemcpyDeviceToHost));
    h->alive_count -= cleared;
    h->mutations += 1;
    if (n_cleared_out) *n_cleared_out = cleared;
    return CRH_OK;
}

int crh_index_compact(crh_index *h, int64_t *old_to_new_host, int64_t *rows_after)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    if (rows_after) *rows_after = h->alive_count;
    DeviceGuard g(h->device);
    CRH_HIP(hipDeviceSynchronize());   // searches in flight on any stream read what is about to move
    if (!h->pending.empty()) return fail(CRH_E_INVALID, "compact: searches with device outputs are pending (call crh_search_finish first)");
    const int64_t count = h->count, ntiles = ceil_div(count, 32);
    if (count == 0) return CRH_OK;
    if (h->alive_count == count) {   // nothing to reclaim: the identity map
        if (old_to_new_host)
            for (int64_t r = 0; r < count; ++r) old_to_new_host[r] = r;
        return CRH_OK;
    }
    // per-tile prefix of the alive counts (host: ntiles words, 1.25 MB per 10M rows)
    std::vector<uint32_t> words((size_t)ntiles);
    CRH_HIP(hipMemcpy(words.data(), h->alive, (size_t)ntiles * 4, hipMemcpyDeviceToHost));
    std::vector<int64_t> prefix((size_t)ntiles);
    int64_t new_count = 0;
    for (int64_t t = 0; t < ntiles; ++t) {
        prefix[(size_t)t] = new_count;
        new_count += __builtin_popcount(words[(size_t)t]);
    }
    if (new_count != h->alive_count) return fail(CRH_E_INTERNAL, "compact: %lld alive bits, %lld alive rows on record", (long long)new_count, (long long)h->alive_count);
    const int64_t new_tiles = ceil_div(new_count, 32);
    // The rows only ever move DOWN (new <= old), so the image is compacted in place, chunk by chunk through a bounce buffer:
    // the new tiles [c, c + C) are gathered from old rows >= 32 c -- which no earlier chunk has overwritten -- and then copied
    // over the old tiles [c, c + C), which no later chunk reads.  The chunk bounds the extra memory (1M rows: 1.5 GB of tiles).
    const int64_t chunk_tiles = std::min<int64_t>(std::max<int64_t>(new_tiles, 1), 1 << 15);
    const int64_t chunk_rows = chunk_tiles * 32;
    const size_t per_row = std::max<size_t>((size_t)h->dim * 2, h->xf32 ? (size_t)h->dim * 4 : 0);
    int64_t *d_prefix = nullptr, *d_o2n = nullptr;
    int32_t *d_n2o = nullptr;
    void *bounce = nullptr;
    auto cleanup = [&]() {
        if (d_prefix) (void)hipFree(d_prefix);
        if (d_o2n) (void)hipFree(d_o2n);
        if (d_n2o) (void)hipFree(d_n2o);
        if (bounce) (void)hipFree(bounce);
    };
#define CRH_CPT(expr)                                                                                                        \
    do {                                                                                                                     \
        hipError_t e_ = (expr);                                                                                              \
        if (e_ != hipSuccess) {                                                                                              \
            cleanup();                                                                                                       \
            return fail(CRH_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);               \
        }                                                                                                                    \
    } while (0)
    CRH_CPT(hipMalloc(reinterpret_cast<void **>(&d_prefix), (size_t)ntiles * 8));
    CRH_CPT(hipMalloc(reinterpret_cast<void **>(&d_o2n), (size_t)count * 8));
    CRH_CPT(hipMalloc(reinterpret_cast<void **>(&d_n2o), (size_t)std::max<int64_t>(new_count, 1) * 4));
    CRH_CPT(hipMalloc(&bounce, (size_t)chunk_rows * per_row));
    CRH_CPT(hipMemcpy(d_prefix, prefix.data(), (size_t)ntiles * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_compact_map, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0, 0, h->alive, d_prefix, count, d_o2n, d_n2o);
    CRH_CPT(hipGetLastError());
    const size_t tile_bytes = (size_t)h->ksteps * 1024;
    const int row16 = h->dim / 4;   // 16-byte pieces of an f32 row
    for (int64_t t0 = 0; t0 < new_tiles; t0 += chunk_tiles) {
        const int64_t nt = std::min(chunk_tiles, new_tiles - t0), r0 = t0 * 32, nr = nt * 32;
        hipLaunchKernelGGL(k_compact_tiles, dim3((unsigned)nt), dim3(256), 0, 0, h->xt, h->ksteps, d_n2o, new_count, t0, static_cast<u32x4 *>(bounce));
        CRH_CPT(hipGetLastError());
        CRH_CPT(hipMemcpyAsync(reinterpret_cast<char *>(h->xt) + (size_t)t0 * tile_bytes, bounce, (size_t)nt * tile_bytes, hipMemcpyDeviceToDevice, 0));
        if (h->xf32) {
            hipLaunchKernelGGL(k_compact_rows16, dim3((unsigned)ceil_div(nr * row16, 256)), dim3(256), 0, 0, reinterpret_cast<const u32x4 *>(h->xf32), row16,
                               d_n2o, new_count, r0, nr, static_cast<u32x4 *>(bounce));
            CRH_CPT(hipGetLastError());
            CRH_CPT(hipMemcpyAsync(h->xf32 + r0 * h->dim, bounce, (size_t)nr * h->dim * 4, hipMemcpyDeviceToDevice, 0));
        }
        for (int c = 0; c < h->ncols; ++c) {
            int32_t *col = h->codes + (int64_t)c * h->cap_rows;
            hipLaunchKernelGGL(k_compact_i32, dim3((unsigned)ceil_div(nr, 256)), dim3(256), 0, 0, col, d_n2o, new_count, r0, nr, (int32_t)-1,
                               static_cast<int32_t *>(bounce));
            CRH_CPT(hipGetLastError());
            CRH_CPT(hipMemcpyAsync(col + r0, bounce, (size_t)nr * 4, hipMemcpyDeviceToDevice, 0));
        }
    }
    // what lies beyond the new end goes back to the state crh_index_create leaves: zero tiles, no alive bits, codes -1
    if (ntiles > new_tiles) {
        CRH_CPT(hipMemsetAsync(reinterpret_cast<char *>(h->xt) + (size_t)new_tiles * tile_bytes, 0, (size_t)(ntiles - new_tiles) * tile_bytes, 0));
        for (int c = 0; c < h->ncols; ++c)
            CRH_CPT(hipMemsetAsync(h->codes + (int64_t)c * h->cap_rows + new_tiles * 32, 0xff, (size_t)(ntiles - new_tiles) * 32 * 4, 0));
    }
    hipLaunchKernelGGL(k_compact_alive, dim3((unsigned)ceil_div(ntiles, 256)), dim3(256), 0, 0, h->alive, new_count, ntiles);
    CRH_CPT(hipGetLastError());
    if (old_to_new_host) CRH_CPT(hipMemcpy(old_to_new_host, d_o2n, (size_t)count * 8, hipMemcpyDeviceToHost));
    CRH_CPT(hipDeviceSynchronize());
#undef CRH_CPT
    cleanup();
    h->count = new_count;
    h->mutations += 1;
    h->i8_dirty_from = 0;
    h->alive_count = new_count;
    if (rows_after) *rows_after = new_count;
    return CRH_OK;
}

int crh_index_export(crh_index *h, int64_t first_tile, int64_t n_tiles, void *tiles_host, float *master_host, uint32_t *alive_host,
                     int32_t *codes_host)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    const int64_t used_tiles = ceil_div(h->count, 32);
    if (first_tile < 0 || n_tiles < 0 || first_tile + n_tiles > used_tiles)
        return fail(CRH_E_INVALID, "tile range [%lld,+%lld) outside the %lld tiles in use", (long long)first_tile, (long long)n_tiles, (long long)used_tiles);
    if (n_tiles == 0) return CRH_OK;
    DeviceGuard g(h->device);
    CRH_HIP(hipDeviceSynchronize());
    const size_t tile_bytes = (size_t)h->ksteps * 1024;
    if (tiles_host) CRH_HIP(hipMemcpy(tiles_host, reinterpret_cast<const char *>(h->xt) + (size_t)first_tile * tile_bytes, (size_t)n_tiles * tile_bytes, hipMemcpyDeviceToHost));
    const int64_t r0 = first_tile * 32, nr = std::min(n_tiles * 32, h->cap_rows - r0);
    if (master_host) {
        if (!h->xf32) return fail(CRH_E_INVALID, "export: this index keeps no f32 master copy (dtype bf16)");
        CRH_HIP(hipMemcpy(master_host, h->xf32 + r0 * h->dim, (size_t)nr * h->dim * 4, hipMemcpyDeviceToHost));
        if (r0 + nr > h->count)   // slots of the last tile past the row count were never written: the file gets zeros there
            memset(master_host + (h->count - r0) * h->dim, 0, (size_t)(r0 + nr - h->count) * h->dim * 4);
    }
    if (alive_host) CRH_HIP(hipMemcpy(alive_host, h->alive + first_tile, (size_t)n_tiles * 4, hipMemcpyDeviceToHost));
    if (codes_host)
        for (int c = 0; c < h->ncols; ++c)
            CRH_HIP(hipMemcpy(codes_host + (int64_t)c * n_tiles * 32, h->codes + (int64_t)c * h->cap_rows + r0, (size_t)nr * 4, hipMemcpyDeviceToHost));
    return CRH_OK;
}

int crh_index_import(crh_index *h, int64_t first_tile, int64_t n_tiles, int64_t rows_after, const void *tiles_host,
                     const float *master_host, const uint32_t *alive_host, const int32_t *codes_host)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    if (first_tile < 0 || n_tiles <= 0 || first_tile + n_tiles > h->cap_tiles)
        return fail(CRH_E_CAPACITY, "import of tiles [%lld,+%lld) exceeds the capacity of %lld tiles (reserve first)", (long long)first_tile,
                    (long long)n_tiles, (long long)h->cap_tiles);
    if (first_tile * 32 != ((h->count + 31) & ~31LL))
        return fail(CRH_E_INVALID, "import must continue at the end of the index (tile %lld, got %lld)", (long long)(((h->count + 31) & ~31LL) / 32), (long long)first_tile);
    if (rows_after <= first_tile * 32 || rows_after > (first_tile + n_tiles) * 32 || rows_after <= (first_tile + n_tiles - 1) * 32)
        return fail(CRH_E_INVALID, "rows_after=%lld does not end inside the last imported tile", (long long)rows_after);
    if (!tiles_host || !alive_host) return fail(CRH_E_INVALID, "import: tiles and alive words are required");
    if ((h->xf32 != nullptr) != (master_host != nullptr)) return fail(CRH_E_INVALID, "import: the f32 master copy goes with dtype f32, and only with it");
    if ((h->ncols > 0) != (codes_host != nullptr)) return fail(CRH_E_INVALID, "import: index has %d code columns", h->ncols);
    DeviceGuard g(h->device);
    CRH_HIP(hipDeviceSynchronize());
    const size_t tile_bytes = (size_t)h->ksteps * 1024;
    const int64_t r0 = first_tile * 32, nr = n_tiles * 32;
    CRH_HIP(hipMemcpy(reinterpret_cast<char *>(h->xt) + (size_t)first_tile * tile_bytes, tiles_host, (size_t)n_tiles * tile_bytes, hipMemcpyHostToDevice));
    if (master_host) CRH_HIP(hipMemcpy(h->xf32 + r0 * h->dim, master_host, (size_t)nr * h->dim * 4, hipMemcpyHostToDevice));
    // rows past rows_after in the last tile are not rows of the index: their alive bits must stay clear
    std::vector<uint32_t> al(alive_host, alive_host + n_tiles);
    const int tail = (int)(rows_after - (first_tile + n_tiles - 1) * 32);
    if (tail < 32) al[(size_t)n_tiles - 1] &= (1u << tail) - 1u;
    int64_t alive_rows = 0;
    for (uint32_t w : al) alive_rows += __builtin_popcount(w);
    CRH_HIP(hipMemcpy(h->alive + first_tile, al.data(), (size_t)n_tiles * 4, hipMemcpyHostToDevice));
    for (int c = 0; c < h->ncols; ++c)
        CRH_HIP(hipMemcpy(h->codes + (int64_t)c * h->cap_rows + r0, codes_host + (int64_t)c * nr, (size_t)nr * 4, hipMemcpyHostToDevice));
    h->count = rows_after;
    h->mutations += 1;
    h->i8_dirty_from = std::min<int64_t>(h->i8_dirty_from, first_tile);
    h->alive_count += alive_rows;
    return CRH_OK;
}

int crh_index_count(crh_index *h, int64_t *rows_out, int64_t *alive_out)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    if (rows_out) *rows_out = h->count;
    if (alive_out) *alive_out = h->alive_count;
    return CRH_OK;
}

int crh_index_clear(crh_index *h)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    DeviceGuard g(h->device);
    CRH_HIP(hipDeviceSynchronize());
    const int64_t used_tiles = ceil_div(h->count, 32);
    if (used_tiles) {
        CRH_HIP(hipMemset(h->xt, 0, (size_t)used_tiles * h->ksteps * 1024));
        CRH_HIP(hipMemset(h->alive, 0, (size_t)used_tiles * 4));
    }
    h->count = 0;
    h->mutations += 1;
    h->i8_dirty_from = 0;
    h->alive_count = 0;
    h->pending.clear();
    h->next_slot = 0;
    return CRH_OK;
}

int crh_index_reserve(crh_index *h, int64_t capacity_rows)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    if (capacity_rows <= h->cap_rows) return CRH_OK;
    if (capacity_rows > (1LL << 31)) return fail(CRH_E_INVALID, "capacity_rows %lld out of range", (long long)capacity_rows);
    DeviceGuard g(h->device);
    CRH_HIP(hipDeviceSynchronize());
    const int64_t new_rows = (capacity_rows + 31) & ~31LL, new_tiles = new_rows / 32;
    const int64_t used_tiles = ceil_div(h->count, 32);
    u32x4 *nxt = nullptr;
    float *nf32 = nullptr;
    uint32_t *nalive = nullptr;
    int32_t *ncodes = nullptr;
    int rc = dev_alloc(&nxt, new_tiles * h->ksteps * 64);
    if (rc == CRH_OK && h->xf32) rc = dev_alloc(&nf32, new_rows * h->dim);
    if (rc == CRH_OK) rc = dev_alloc(&nalive, new_tiles);
    if (rc == CRH_OK && h->ncols) rc = dev_alloc(&ncodes, new_rows * h->ncols);
    hipError_t e = hipSuccess;
    if (rc == CRH_OK) {
        const size_t used_b = (size_t)used_tiles * h->ksteps * 1024;
        if (e == hipSuccess) e = hipMemset(reinterpret_cast<char *>(nxt) + used_b, 0, (size_t)new_tiles * h->ksteps * 1024 - used_b);
        if (e == hipSuccess && used_b) e = hipMemcpy(nxt, h->xt, used_b, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemset(nalive, 0, (size_t)new_tiles * 4);
        if (e == hipSuccess && used_tiles) e = hipMemcpy(nalive, h->alive, (size_t)used_tiles * 4, hipMemcpyDeviceToDevice);
        if (e == hipSuccess && nf32 && h->count) e = hipMemcpy(nf32, h->xf32, (size_t)h->count * h->dim * 4, hipMemcpyDeviceToDevice);
        if (e == hipSuccess && ncodes) e = hipMemset(ncodes, 0xff, (size_t)new_rows * h->ncols * 4);
        for (int c = 0; c < h->ncols && e == hipSuccess && h->count; ++c)
            e = hipMemcpy(ncodes + (int64_t)c * new_rows, h->codes + (int64_t)c * h->cap_rows, (size_t)h->count * 4, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) rc = fail(CRH_E_HIP, "reserve copy failed: %s", hipGetErrorString(e));
    }
    if (rc != CRH_OK) {
        dev_free(nxt);
        dev_free(nf32);
        dev_free(nalive);
        dev_free(ncodes);
        return rc;
    }
    dev_free(h->xt);
    dev_free(h->xf32);
    dev_free(h->alive);
    dev_free(h->codes);
    h->xt = nxt;
    h->xf32 = nf32;
    h->alive = nalive;
    h->codes = ncodes;
    h->cap_rows = new_rows;
    h->mutations += 1;   // (the code columns moved to their new pitch)
    h->cap_tiles = new_tiles;
    return CRH_OK;
}

int crh_index_read_rows(crh_index *h, int64_t first, int64_t n, float *out_host)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    if (first < 0 || n < 0 || first + n > h->count) return fail(CRH_E_INVALID, "row range [%lld,+%lld) outside the index", (long long)first, (long long)n);
    if (n == 0) return CRH_OK;
    if (!out_host) return fail(CRH_E_INVALID, "out_host is NULL");
    DeviceGuard g(h->device);
    if (h->dtype == CRH_DTYPE_F32) {
        CRH_HIP(hipMemcpy(out_host, h->xf32 + first * h->dim, (size_t)n * h->dim * 4, hipMemcpyDeviceToHost));
        return CRH_OK;
    }
    const int64_t chunk = 65536;
    CRH_TRY(ensure_stage_in(h, std::min(chunk, n) * h->dim * 4));
    for (int64_t off = 0; off < n; off += chunk) {
        const int64_t m = std::min(chunk, n - off);
        hipLaunchKernelGGL(k_untile_rows, dim3((unsigned)ceil_div(m * (h->dim / 8), 256)), dim3(256), 0, 0, h->xt, h->ksteps,
                           first + off, m, h->dim, static_cast<float *>(h->stage_in));
        CRH_HIP(hipGetLastError());
        CRH_HIP(hipMemcpy(out_host + off * h->dim, h->stage_in, (size_t)m * h->dim * 4, hipMemcpyDeviceToHost));
    }
    return CRH_OK;
}

#ifdef CRH_ENABLE_DEBUG   // libcoderag_hip_debug.so only
int crh_debug_read_ceiling(crh_index *h, void *stream)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    DeviceGuard g(h->device);
    const int64_t ntiles = ceil_div(h->count, kTileRows);
    if (ntiles == 0) return CRH_OK;
    CRH_TRY(ensure_workspace0(h));
    crh_index::Workspace &w = h->ws;
    const uint32_t *mask = nullptr;
    CRH_TRY(build_mask(h, w, nullptr, 0, &mask, static_cast<hipStream_t>(stream)));
    return launch_scan<2>(h, w, scan_blocks(h, ntiles), static_cast<hipStream_t>(stream), mask, (int)ntiles, 1, w.ws_wave_cap, w.ws_qcap, h->status);
}

// Moves the int8 copy to a fresh allocation (the new one is taken BEFORE the old one is released, so it lands elsewhere): the
// speed of a pass depends on where the copy sits (tools/alloc_modes.py), this lets one index try several places.
int crh_debug_i8_move(crh_index *h)
{
    if (!h || !h->i8 || !h->x8) return fail(CRH_E_INVALID, "no int8 copy to move");
    DeviceGuard g(h->device);
    CRH_HIP(hipDeviceSynchronize());
    const int ks8 = h->dim / 32;
    u32x4 *nx = nullptr;
    float *ns = nullptr;
    CRH_HIP(hipMalloc(reinterpret_cast<void **>(&nx), (size_t)h->x8_cap_tiles * ks8 * 1024));
    CRH_HIP(hipMalloc(reinterpret_cast<void **>(&ns), (size_t)h->x8_cap_tiles * 32 * sizeof(float)));
    dev_free(h->x8);
    dev_free(h->srow);
    h->x8 = nx;
    h->srow = ns;
    h->i8_dirty_from = 0;
    return CRH_OK;
}
#endif  // CRH_ENABLE_DEBUG

#ifdef CRH_FUSED_STAMPS   // a measurement build only (tools/fused_stamps.py): per-workgroup phase clocks of the last k_scan_fused
int crh_debug_fused_stamps(unsigned long long *out)
{
    CRH_HIP(hipDeviceSynchronize());
    CRH_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_stamps), sizeof(unsigned long long) * 256 * 24));
    return CRH_OK;
}
int crh_debug_select_stamps(unsigned long long *out)
{
    CRH_HIP(hipDeviceSynchronize());
    CRH_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_select_stamps), sizeof(unsigned long long) * 64 * 16));
    return CRH_OK;
}
#endif

int crh_index_set_nomination(crh_index *h, int mode)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    if (mode < CRH_NOMINATE_BF16_3 || mode > CRH_NOMINATE_INT8) return fail(CRH_E_INVALID, "unknown nomination mode %d", mode);
    h->nominate_max = mode;
    if (mode == CRH_NOMINATE_INT8) h->i8_strikes = h->i8_cooldown = 0;   // asking for it again gives the copy another chance at once
    return CRH_OK;
}

int crh_index_get_nomination(crh_index *h, int *mode_out)
{
    if (!h || !mode_out) return fail(CRH_E_INVALID, "NULL argument");
    const bool one_launch = h->fused_scan && !h->fused_resting() && h->nominate_max >= CRH_NOMINATE_BF16 && h->seed_tiles == 4096 && h->ksteps != 64;
    *mode_out = i8_use(h, 1, 100) ? CRH_NOMINATE_INT8 : (one_launch ? CRH_NOMINATE_BF16 : CRH_NOMINATE_BF16_3);
    return CRH_OK;
}

int crh_index_set_tuning(crh_index *h, int seed_tiles, int wave_cand_cap, int query_cand_cap, int force_fallback)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    if (seed_tiles > 0) h->seed_tiles = seed_tiles > 12288 ? 12288 : seed_tiles;  // k_tau keeps the sample column in LDS
    if (wave_cand_cap > 0) h->wave_cap = wave_cand_cap;
    if (query_cand_cap > 0) h->qcap = query_cand_cap;
    if (force_fallback >= 0) h->force_fallback = force_fallback;
    return CRH_OK;
}

int crh_search(crh_index *h, int nq, const float *queries, int queries_on_device, int k, const crh_filter *filters, int n_filters,
               int64_t row_base, float *out_scores, int64_t *out_rows, int out_on_device, void *stream)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    if (nq < 0) return fail(CRH_E_INVALID, "nq < 0");
    if (nq == 0) return CRH_OK;
    if (!queries || !out_scores || !out_rows) return fail(CRH_E_INVALID, "NULL query or output pointer");
    if (k <= 0 || k > CRH_MAX_K) return fail(CRH_E_CAPACITY, "k=%d outside 1..%d", k, CRH_MAX_K);
    if (n_filters < 0 || n_filters > CRH_MAX_FILTERS) return fail(CRH_E_INVALID, "n_filters=%d outside 0..%d", n_filters, CRH_MAX_FILTERS);
    if (n_filters > 0 && !filters) return fail(CRH_E_INVALID, "filters is NULL");
    DeviceGuard g(h->device);
    hipStream_t st = static_cast<hipStream_t>(stream);

    // force_fallback (testing): 1 = start from absurdly small candidate buffers so the regrow-and-rerun path runs (behind the int8
    // scan: its one regrowth); 3 = the same, and the int8 scan may NOT regrow (its batches go to the bf16 scan: the strike path);
    // 2 = the one-launch scan's first grid-wide wait expects an arrival too many, so its time-out path runs
    const bool tiny = h->force_fallback == 1 || h->force_fallback == 3;
    int wc = tiny ? 4 : std::max(h->wave_cap, h->ws.ws_wave_cap);
    int qc = tiny ? 8 : std::max(h->qcap, h->ws.ws_qcap);
    if (!tiny && i8_use(h, 1, k)) {
        // the int8 intervals nominate a fixed FRACTION of the rows (the thresholds come from a fixed number of sample tiles):
        // ~21 k per query and ~660 per wave of the int8 scan at 10M rows.  The defaults hold that twice over up to 10M rows; beyond,
        // the buffers grow with the index (the int8 scan cannot regrow them after the fact: an overflow sends the batch to bf16)
        const int64_t scale = next_pow2(ceil_div(h->count, 10000000));
        wc = (int)std::max<int64_t>(wc, std::min<int64_t>(2048 * scale, 1 << 18));
        qc = (int)std::max<int64_t>(qc, std::min<int64_t>(131072 * scale, 1 << 24));
    }
    CRH_TRY(ensure_workspace(h, h->ws, wc, qc));

    const float *q_dev = queries;
    if (!queries_on_device) {
        const int64_t elems = (int64_t)nq * h->dim;
        if (h->stage_q_elems < elems) {
            dev_free(h->stage_q);
            h->stage_q_elems = 0;
            CRH_TRY(dev_alloc(&h->stage_q, elems));
            h->stage_q_elems = elems;
        }
        CRH_HIP(hipMemcpyAsync(h->stage_q, queries, (size_t)elems * 4, hipMemcpyHostToDevice, st));
        q_dev = h->stage_q;
    }
    float *os = out_scores;
    int64_t *orow = out_rows;
    if (!out_on_device) {
        const int64_t elems = (int64_t)nq * k;
        if (h->stage_out_elems < elems) {
            dev_free(h->stage_os);
            dev_free(h->stage_or);
            h->stage_out_elems = 0;
            CRH_TRY(dev_alloc(&h->stage_os, elems));
            CRH_TRY(dev_alloc(&h->stage_or, elems));
            h->stage_out_elems = elems;
        }
        os = h->stage_os;
        orow = h->stage_or;
    }
    if (h->pending.empty()) h->stats = crh_search_stats{};

    // more than one k_scan pass worth of queries: up to kWideQ of them share ONE corpus pass through k_scan_wide
    for (int q0 = 0, b = 0; q0 < nq; q0 += b) {
        const int left = nq - q0;
        b = (h->wide_ok && left > h->batch_q) ? std::min(kWideQ, left) : std::min(h->batch_q, left);
        // up to two passes over the int8 copy (2 x ~1.55 ms at 10M rows) beat one wide pass over the bf16 tiles (~4.2 ms)
        if (left > h->batch_q && left <= 2 * h->batch_q && h->count > 0 && i8_use(h, h->batch_q, k)) b = h->batch_q;
        if (h->next_slot >= kStatusSlots) CRH_TRY(finish_pending(h, st));
        crh_index::Workspace &w = h->ws;
        Pending p{};
        p.nq = b;
        p.k = k;
        p.nfilt = n_filters;
        for (int f = 0; f < n_filters; ++f) p.filt[f] = filters[f];
        p.row_base = row_base;
        p.q_dev = q_dev + (int64_t)q0 * h->dim;
        p.out_s = os + (int64_t)q0 * k;
        p.out_r = orow + (int64_t)q0 * k;
        p.slot = h->next_slot++;
        const uint32_t *mask = nullptr;
        // (the filter mask lives in the workspace: stream order puts its rebuild behind the previous batch's scan)
        CRH_TRY(build_mask(h, w, filters, n_filters, &mask, st));
        CRH_TRY(enqueue_batch(h, w, p.q_dev, b, k, mask, row_base, p.out_s, p.out_r, p.slot, st, &p.path));
        h->pending.push_back(p);
    }
    if (!out_on_device || !queries_on_device) {
        CRH_TRY(finish_pending(h, st));
        if (!out_on_device) {
            CRH_HIP(hipMemcpyAsync(out_scores, h->stage_os, (size_t)nq * k * 4, hipMemcpyDeviceToHost, st));
            CRH_HIP(hipMemcpyAsync(out_rows, h->stage_or, (size_t)nq * k * 8, hipMemcpyDeviceToHost, st));
            CRH_HIP(hipStreamSynchronize(st));
        }
    }
    return CRH_OK;
}

int crh_search_finish(crh_index *h, void *stream)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    DeviceGuard g(h->device);
    return finish_pending(h, static_cast<hipStream_t>(stream));
}

int crh_index_set_profiling(crh_index *h, int enable)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    DeviceGuard g(h->device);
    if (enable && h->ev.empty()) {
        h->ev.resize(2 * kStatusSlots);
        for (auto &e : h->ev) CRH_HIP(hipEventCreate(&e));
    }
    h->profiling = enable != 0;
    h->profile_whole_scan = enable == 2;
    h->prof_scan_ms = 0.0;
    h->prof_scan_launches = 0;
    return CRH_OK;
}

int crh_index_get_profile(crh_index *h, double *scan_ms_total, int64_t *scan_launches)
{
    if (!h) return fail(CRH_E_INVALID, "index is NULL");
    if (scan_ms_total) *scan_ms_total = h->prof_scan_ms;
    if (scan_launches) *scan_launches = h->prof_scan_launches;
    return CRH_OK;
}

int crh_search_get_stats(crh_index *h, crh_search_stats *out)
{
    if (!h || !out) return fail(CRH_E_INVALID, "NULL argument");
    *out = h->stats;
    return CRH_OK;
}

int crh_merge_topk(int nlists, int nq, int k, const float *scores_dev, const int64_t *rows_dev, float *out_scores_dev,
                   int64_t *out_rows_dev, void *stream)
{
    return crh_merge_topk_strided(nlists, nq, k, scores_dev, rows_dev, (int64_t)nq * k, (int64_t)nq * k, out_scores_dev, out_rows_dev, stream);
}

int crh_merge_topk_strided(int nlists, int nq, int k, const float *scores_dev, const int64_t *rows_dev, int64_t score_list_stride,
                           int64_t row_list_stride, float *out_scores_dev, int64_t *out_rows_dev, void *stream)
{
    if (nlists <= 0 || nq < 0 || k <= 0) return fail(CRH_E_INVALID, "bad merge shape nlists=%d nq=%d k=%d", nlists, nq, k);
    if (score_list_stride < (int64_t)nq * k || row_list_stride < (int64_t)nq * k)
        return fail(CRH_E_INVALID, "merge list strides %lld / %lld are shorter than a list of %d x %d", (long long)score_list_stride,
                    (long long)row_list_stride, nq, k);
    if (nq == 0) return CRH_OK;
    if (!scores_dev || !rows_dev || !out_scores_dev || !out_rows_dev) return fail(CRH_E_INVALID, "NULL pointer");
    const int total = nlists * k;
    if (total > 8192) return fail(CRH_E_CAPACITY, "nlists*k=%d exceeds 8192", total);
    int P = 1;
    while (P < total) P <<= 1;
    const size_t lds = (size_t)P * 4 + 8 + (size_t)P * 8;   // 96 KiB at P = 8192: above the 64 KiB a launch gets by default
    static OncePerDevice once;
    if (once.need()) CRH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_merge_topk), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 12 + 8));
    hipLaunchKernelGGL(k_merge_topk, dim3(nq), dim3(1024), lds, static_cast<hipStream_t>(stream), nlists, nq, k, scores_dev, rows_dev,
                       score_list_stride, row_list_stride, out_scores_dev, out_rows_dev);
    CRH_HIP(hipGetLastError());
    return CRH_OK;
}

int crh_index_match_rows(crh_index *h, const crh_filter *filters, int n_filters, int64_t limit, int64_t *rows_out_host, int64_t *n_out)
{
    if (!h || !n_out) return fail(CRH_E_INVALID, "NULL argument");
    *n_out = 0;
    if (limit <= 0 || h->count == 0) return CRH_OK;   // (rows_out_host == NULL: count only, up to `limit`)
    if (n_filters < 0 || n_filters > CRH_MAX_FILTERS) return fail(CRH_E_INVALID, "n_filters=%d outside 0..%d", n_filters, CRH_MAX_FILTERS);
    DeviceGuard g(h->device);
    CRH_TRY(ensure_workspace0(h));
    const uint32_t *mask = nullptr;
    CRH_TRY(build_mask(h, h->ws, filters, n_filters, &mask, nullptr));
    const int64_t ntiles = ceil_div(h->count, 32);
    std::vector<uint32_t> hm((size_t)ntiles);
    CRH_HIP(hipMemcpy(hm.data(), mask, (size_t)ntiles * 4, hipMemcpyDeviceToHost));
    int64_t found = 0;
    for (int64_t t = 0; t < ntiles && found < limit; ++t) {
        uint32_t m = hm[(size_t)t];
        while (m && found < limit) {
            const int b = __builtin_ctz(m);
            m &= m - 1;
            if (rows_out_host) rows_out_host[found] = t * 32 + b;
            ++found;
        }
    }
    *n_out = found;
    return CRH_OK;
}

}  // extern "C"
