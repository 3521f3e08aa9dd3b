/*
 * coderag_hip.h -- C ABI of libcoderag_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary of this repo.  The reference (iAmLakshya/code-rag, Python
 * package `lattice`) reaches its vector store through qdrant-client gRPC calls and
 * its encoder through torch/transformers; it has no FFI of its own.  These entry
 * points are what a ctypes binding for that path binds instead (INTEGRATION.md shows
 * the stub).  Each entry cites the reference interface it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error (CRH_E_*);
 *    crh_last_error() returns a thread-local message for the last failure;
 *  - no C++ types, no torch types, no exceptions across this boundary;
 *  - the caller owns every buffer it passes in; the library copies what it keeps;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *  - a handle is used from one thread at a time (the store holds a lock; the
 *    provider uses a 1-thread executor like providers/unixcoder_provider.py:260);
 *  - "dev" pointers are device memory on the handle's device, "host" pointers are
 *    ordinary process memory; `*_on_device` flags say which one a dual-mode
 *    argument is.
 */
#ifndef CODERAG_HIP_H
#define CODERAG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRH_ABI_VERSION 4

/* status codes */
#define CRH_OK 0
#define CRH_E_INVALID (-1)  /* bad argument */
#define CRH_E_HIP (-2)      /* a HIP runtime call failed */
#define CRH_E_CAPACITY (-3) /* index full / k too large */
#define CRH_E_NODEVICE (-4) /* no usable gfx950 device */
#define CRH_E_INTERNAL (-5)

/* stored precision of the corpus */
#define CRH_DTYPE_F32 0  /* bf16 scan copy + f32 master copy; ids exact vs f32 oracle */
#define CRH_DTYPE_BF16 1 /* bf16 only; ids exact vs oracle on the bf16-rounded corpus */

#define CRH_MAX_FILTERS 8
#define CRH_MAX_K 1024

typedef struct crh_index crh_index; /* opaque */

/* One payload equality predicate: column `col` of the dictionary-coded payload must equal
 * `code`.  Replaces one models.FieldCondition(key, MatchValue(value)) of
 * embeddings/client.py:171-176; the string<->code dictionaries stay in host Python. */
typedef struct crh_filter {
    int32_t col;
    int32_t code;
} crh_filter;

/* Counters of the most recent crh_search* call on a handle (diagnostics / bench). */
typedef struct crh_search_stats {
    int64_t rows;            /* rows scanned */
    int64_t tiles;           /* 32-row tiles scanned */
    int64_t seed_tiles;      /* tiles in the threshold-seeding sample */
    int64_t candidates;      /* (score,row) pairs that passed the scan threshold, all queries */
    int64_t max_query_cands; /* largest per-query candidate count */
    int32_t fallback_used;   /* bit 0: a batch was re-run with larger candidate buffers; bit 1: a grid-wide wait of the
                                one-launch scan timed out and the index went back to the three-launch form */
    int32_t batches;         /* 64-query batches processed */
} crh_search_stats;

int crh_abi_version(void);
const char *crh_last_error(void);

/* Number of HIP devices / name+arch of one ("gfx950..." expected).  */
int crh_device_count(int *count);
int crh_device_info(int device, char *name_out, int name_cap, char *arch_out, int arch_cap,
                    int64_t *hbm_bytes_out, int *cu_count_out);

/* ------------------------------------------------------------------ index ------ */

/* Create an empty HBM-resident cosine index.  Replaces
 * QdrantManager._create_collection_with_indexes (embeddings/client.py:93-113):
 * VectorParams(size=dim, distance=COSINE) + `n_code_cols` keyword payload indexes.
 * dim: 384, 768 (UniXcoder; the tuned case), 1024 or 1536 (the reference's EMBEDDING_DIMENSIONS default,
 * config/settings.py:53). */
int crh_index_create(int dim, int dtype, int64_t capacity_rows, int n_code_cols, int device,
                     crh_index **out);
int crh_index_destroy(crh_index *h);

/* Append n vectors (raw, un-normalised f32 [n, dim]; host or device).  Each is passed
 * through Qdrant's cosine_preprocess and stored; rows are numbered consecutively from
 * the current count, the first new row is returned.  codes: [n, n_code_cols] int32
 * (same memory space as vecs) or NULL when n_code_cols == 0.
 * Replaces the vector half of QdrantManager.upsert (embeddings/client.py:115-130). */
int crh_index_append(crh_index *h, int64_t n, const float *vecs, int on_device,
                     const int32_t *codes, int64_t *first_row_out, void *stream);

/* Same, for vectors that already went through cosine_preprocess (rows read back with crh_index_read_rows, and only those:
 * the scan's exactness margin assumes unit or zero rows): stored verbatim.  (Whole-index snapshots use crh_index_export /
 * crh_index_import, which move the stored image itself.) */
int crh_index_append_preprocessed(crh_index *h, int64_t n, const float *vecs, int on_device,
                                  const int32_t *codes, int64_t *first_row_out, void *stream);

/* Mark rows deleted (they stop matching; the space comes back with crh_index_compact).  rows: host int64[n].
 * Replaces the point-removal half of QdrantManager.delete (embeddings/client.py:159-169);
 * which rows a payload filter selects is resolved by the host-side payload table. */
int crh_index_tombstone(crh_index *h, int64_t n, const int64_t *rows);

/* Delete by payload filter on the device: every alive row matching ALL the (column == code) predicates stops matching;
 * n_cleared_out receives how many.  Replaces QdrantManager.delete -> client.delete(FilterSelector(filter))
 * (embeddings/client.py:159-169) without resolving the filter to row numbers on the host. */
int crh_index_tombstone_filter(crh_index *h, const crh_filter *filters, int n_filters, int64_t *n_cleared_out);

/* Reclaim the rows of deleted points -- what Qdrant's optimizer does when it vacuums a segment.  The reference's indexing
 * flow deletes and re-inserts every chunk of a file on every run (embeddings/indexer.py:61-64, called with force=True from
 * pipeline/orchestrator.py:630-650), so without this the scan streams one more corpus of dead rows per re-index.
 * Device-side stable compaction of everything stored per row (the tiled bf16 image, the f32 master, the code columns): the
 * alive rows move down in their old order, so row numbers stay ascending in insertion order and ties keep "lower row first";
 * afterwards count == alive and the scan reads only live tiles.  old_to_new_host (may be NULL): int64 per OLD row, its new
 * row number or -1 for a deleted one -- what the caller's host tables (ids, payloads, side columns) are remapped with.
 * rows_after (may be NULL) receives the new row count.  Capacity is unchanged. */
int crh_index_compact(crh_index *h, int64_t *old_to_new_host, int64_t *rows_after);

/* Snapshot support -- what Qdrant's on-disk volume does for the reference (docker-compose.yml:42-43): the stored image moves
 * VERBATIM between HBM and host buffers (typically an mmap of a file) in chunks of 32-row tiles, so a restored index answers
 * with identical bits.  Per chunk of n_tiles tiles starting at first_tile:
 *   tiles   n_tiles * (dim/16) KiB  the tiled bf16 image the scan streams (ABI 3: inside a 1-KiB piece the 16-byte chunk of
 *                                   row r, half h sits at slot 2r + h; ABI 2 had h * 32 + r -- ffi.Index.load reorders such files)
 *   master  n_tiles * 32 * dim f32  the normalised f32 rows (dtype F32 only; NULL otherwise)
 *   alive   n_tiles u32             one validity word per tile (tombstones included)
 *   codes   [n_code_cols][n_tiles*32] int32, column after column (NULL when the index has no code columns)
 * export: any pointer may be NULL to skip that part.  import: appends at the end of the index (first_tile must be the first
 * unused tile; capacity must have been reserved), `rows_after` = the row count once this chunk is in (it ends inside the
 * chunk's last tile). */
int crh_index_export(crh_index *h, int64_t first_tile, int64_t n_tiles, void *tiles_host, float *master_host,
                     uint32_t *alive_host, int32_t *codes_host);
int crh_index_import(crh_index *h, int64_t first_tile, int64_t n_tiles, int64_t rows_after, const void *tiles_host,
                     const float *master_host, const uint32_t *alive_host, const int32_t *codes_host);

/* rows appended so far / rows still alive (CollectionInfo.points_count, query/engine.py:299-302). */
int crh_index_count(crh_index *h, int64_t *rows_out, int64_t *alive_out);

/* Drop every row (QdrantManager.clear_collections, embeddings/client.py:213-222). */
int crh_index_clear(crh_index *h);

/* Grow the row capacity (no-op if already that large); contents and row numbers are kept.
 * Qdrant collections grow without bound; the HBM arrays are re-allocated and copied device-side. */
int crh_index_reserve(crh_index *h, int64_t capacity_rows);

/* Copy stored (preprocessed) vectors back as f32 [n, dim] to host: rows first..first+n.
 * For dtype BF16 these are the bf16-rounded values.  Test / persistence support. */
int crh_index_read_rows(crh_index *h, int64_t first, int64_t n, float *out_host);

/* -------------------------------------------------------------- search --------- */

/* Cosine top-k of nq raw f32 queries [nq, dim] (host or device) against the alive rows
 * that satisfy every filter.  Writes scores [nq, k] f32 and rows [nq, k] int64 (host or
 * device per out_on_device), descending score, ties by lower row, padded with
 * (-inf, -1) when fewer than k rows qualify.  `row_base` is added to every returned
 * row (shard offset of a row-sharded corpus).
 * Replaces QdrantManager.search -> client.query_points (embeddings/client.py:132-157),
 * batched: the reference issues one query per RPC.
 * With host outputs the call returns after the results have landed.  With device
 * outputs it only enqueues on `stream`; call crh_search_finish() before trusting the
 * buffers (it re-runs the exact fallback if a candidate buffer overflowed). */
int crh_search(crh_index *h, int nq, const float *queries, int queries_on_device, int k,
               const crh_filter *filters, int n_filters, int64_t row_base, float *out_scores,
               int64_t *out_rows, int out_on_device, void *stream);

/* Completes every crh_search enqueued with device outputs since the last finish. */
int crh_search_finish(crh_index *h, void *stream);

int crh_search_get_stats(crh_index *h, crh_search_stats *out);

/* HIP-event timing of the dominant kernel (the corpus scan; behind the int8 copy: the pass, the third of that scan's three
 * launches), recorded on the search stream around each launch while enabled; totals since the last enable.  enable = 2: behind
 * the int8 copy the two events bracket all three launches of the scan (sample tiles, thresholds, pass) -- the span one launch
 * covered in rounds 3-4.  Measurement support for bench.py. */
int crh_index_set_profiling(crh_index *h, int enable);
int crh_index_get_profile(crh_index *h, double *scan_ms_total, int64_t *scan_launches);

/* Tuning knobs (0 / negative = keep default): seed sample tiles, per-wave candidate
 * capacity, per-query candidate capacity, force_fallback (testing): 1 = start from tiny candidate buffers so the
 * regrow-and-rerun path runs (behind the int8 scan: its one regrowth), 3 = the same with that regrowth refused (its batches go to
 * the bf16 scan: the strike path), 2 = make the one-launch bf16 scan's grid-wide wait time out so its recovery path runs, 0 = off. */
int crh_index_set_tuning(crh_index *h, int seed_tiles, int wave_cand_cap, int query_cand_cap,
                         int force_fallback);

/* (No counterpart in the reference: it leaves the search strategy to the Qdrant server, embeddings/client.py:96-102,132-157.)
 * How a batch of up to 64 queries is NOMINATED (every returned id and score is decided by the canonical f32 arithmetic on the
 * stored rows whatever the mode; results are bit-identical across modes):
 *   CRH_NOMINATE_BF16_3  seed scan, threshold, main scan over the bf16 tiles as three launches
 *   CRH_NOMINATE_BF16    the same in one launch (grid-wide waits; needs the whole grid resident)
 *   CRH_NOMINATE_INT8    sample tiles, thresholds and the pass over the int8 copy of the rows, three launches, no grid-wide
 *                        wait (half the bytes of the pass; +1 byte per element and 4 per row of device memory, and for a bf16
 *                        store +2 bytes per element for the row-major rows its selection step reads -- all derived from the
 *                        stored rows, never part of a snapshot) -- the default at every supported dim, from 1M rows up, for
 *                        k <= 256.
 * set: the most advanced mode the index may use.  It still falls back by itself: no memory for the copy; a grid-wide wait of
 * the one-launch bf16 scan that timed out (the wait is bounded by eight times the pass's own time, at least 2 ms; the batch is
 * run again in the three-launch form and the index stays off the one-launch form for 0.2 s, doubling with every further time-out
 * up to 5 s -- the int8 nomination, which waits for nobody, stays in use); int8 candidate buffers that
 * overflowed (they grow ONCE to the observed need when that is at most 2 % of the rows per query; otherwise the batch goes to the
 * bf16 scan, and three such batches in a row rest the copy for 4096 batches, then one more try).  get: the mode the next batch
 * would use. */
#define CRH_NOMINATE_BF16_3 0
#define CRH_NOMINATE_BF16 1
#define CRH_NOMINATE_INT8 2
int crh_index_set_nomination(crh_index *h, int mode);
int crh_index_get_nomination(crh_index *h, int *mode_out);


/* Merge nlists sorted per-shard lists ([nlists, nq, k] device f32 / int64, padded with
 * (-inf,-1)) into [nq, k]: the step after the all-gather of a row-sharded search
 * (does not exist in the reference). */
int crh_merge_topk(int nlists, int nq, int k, const float *scores_dev, const int64_t *rows_dev,
                   float *out_scores_dev, int64_t *out_rows_dev, void *stream);
/* Same, with list l at scores_dev + l * score_list_stride and rows_dev + l * row_list_stride (elements, each >= nq * k): lets
 * every rank send ONE [scores | rows] record through ONE all-gather and merge straight out of the gathered buffer. */
int crh_merge_topk_strided(int nlists, int nq, int k, const float *scores_dev, const int64_t *rows_dev,
                           int64_t score_list_stride, int64_t row_list_stride, float *out_scores_dev,
                           int64_t *out_rows_dev, void *stream);

/* ---- hybrid re-rank of the vector branch (BASELINE config 5; replaces, for vector-only candidate lists, the per-query
 * host loop of HybridRanker.rank_results, src/lattice/query/ranking/ranker.py:24-54 + scorer.py:80-126 + ranker.py:171-229).
 * Scores are computed in f64 with the reference's operand order: bit-identical to the Python result. */
#define CRH_RR_NAME_BYTES 64    /* bytes of the lower-cased entity name kept per row */
#define CRH_RR_MAX_ENTITIES 8   /* query entities per query handled on the device */
#define CRH_RR_ENTITY_BYTES 48

/* What score_vector_result reads of one query (scorer.py:80-126): RankingConfig.weights_for(plan.primary_intent) and the
 * lower-cased names of plan.entities.  n_entities outside 0..CRH_RR_MAX_ENTITIES sends the query back to the host. */
typedef struct crh_rerank_query {
    double vector_weight;
    double centrality_weight;
    int32_t n_entities;
    int32_t entity_len[CRH_RR_MAX_ENTITIES];
    uint8_t entity[CRH_RR_MAX_ENTITIES][CRH_RR_ENTITY_BYTES];
    int32_t pad_;
} crh_rerank_query;

/* Per-candidate side data, device arrays of nq*k entries in candidate order (crh_gather_rows_* fills them from per-row
 * columns): len(content) in characters (0 = empty); total_degree of the row's graph node (-1 = the graph has no answer);
 * dictionary codes of file_path, of the merge key "file:entity_name:start_line" (models.py:55-56) and of the centrality
 * key (graph_node_id or entity_name, scorer.py:48-54); the lower-cased UTF-8 entity name, zero padded to
 * CRH_RR_NAME_BYTES, with its true byte length (a longer name sends its query back to the host). */
typedef struct crh_rerank_columns {
    const int32_t *content_len;
    const int32_t *degree;
    const int32_t *file_code;
    const int32_t *key_code;
    const int32_t *node_code;
    const int32_t *name_len;
    const uint8_t *name;
} crh_rerank_columns;

/* out[i] = col[rows[i] - row_base] for rows owned by this shard (row_base <= row < row_base + n_local), `fill` (bytes: 0)
 * otherwise -- so that the columns of a merged multi-shard candidate list are the sum over the shards' gathers. */
int crh_gather_rows_i32(int64_t n, const int64_t *rows_dev, int64_t row_base, int64_t n_local, const int32_t *col_dev, int32_t fill,
                        int32_t *out_dev, void *stream);
int crh_gather_rows_bytes(int64_t n, const int64_t *rows_dev, int64_t row_base, int64_t n_local, const uint8_t *col_dev, int width,
                          uint8_t *out_dev, void *stream);
/* The seven of them in ONE launch: `cols` holds this shard's PER-ROW columns (device pointers; `name` 4-byte aligned), the output
 * is one int32 buffer -- the six integer columns in the order of the struct, n entries each, then n x CRH_RR_NAME_BYTES name
 * bytes -- i.e. the per-candidate arrays crh_rerank_vector takes are views of it, and a row-sharded caller completes the
 * table of a merged candidate list with one all-reduce(sum) over it (rows of other shards give zeros). */
int crh_gather_rerank_columns(int64_t n, const int64_t *rows_dev, int64_t row_base, int64_t n_local, const crh_rerank_columns *cols,
                              int32_t *out_packed_dev, void *stream);

/* Re-rank nq candidate lists of k vector hits (scores/rows as crh_search or crh_merge_topk return them; rows < 0 = padding).
 * The centrality table of a query holds the first `centrality_top` named hits, as QueryEngine._get_centrality_scores builds
 * it for a query without graph results (query/engine.py:348-377).  Outputs, per query, in final order: index into the
 * candidate list, final score, the four signals (vector_similarity, query_entity_match, centrality, code_quality), flag
 * bit 0 = "hybrid" (entries sharing a merge key were fused); out_count[q] = survivors (<= max_total), or -1 when the host
 * must rank this query (too many entities, a truncated name); the slots behind a query's survivors are written as (-1, 0, ...):
 * the outputs need no clearing before a call.  `queries_dev` is a device copy of nq crh_rerank_query. */
int crh_rerank_vector(int nq, int k, const float *scores_dev, const int64_t *rows_dev, const crh_rerank_columns *cols,
                      const crh_rerank_query *queries_dev, double entity_match_bonus, int max_per_file, int max_total,
                      int centrality_top, int32_t *out_index_dev, double *out_score_dev, double *out_signals_dev,
                      int32_t *out_count_dev, int32_t *out_flags_dev, void *stream);

/* Filter-only fetch: first `limit` alive rows (ascending) matching the filters, host int64 out;
 * n_out receives how many (rows_out_host may be NULL to count only).  Replaces QdrantManager.search(query_vector=None, ...) as used by
 * query/context/builder.py:111-119 and the scroll of embeddings/client.py:178-202. */
int crh_index_match_rows(crh_index *h, const crh_filter *filters, int n_filters, int64_t limit,
                         int64_t *rows_out_host, int64_t *n_out);

/* ------------------------------------------------------------- encoder --------- */
/* UniXcoder = RoBERTa-base geometry encoder (providers/unixcoder_provider.py:137-155 and
 * the HF RobertaModel it wraps).  All pointers are device pointers; activations bf16,
 * LayerNorm / softmax / accumulation f32.  T = total tokens (sum of padded rows). */

/* y[T, N] = act(x[T, K] @ w[N, K]^T + bias[N]); act: 0 none, 1 erf-GELU.  out bf16. */
int crh_gemm_bf16_bias(const void *x, const void *w, const float *bias, void *y, int T, int N, int K,
                       int act, void *stream);

/* y[T, N] = LayerNorm(x @ w^T + bias + residual) * gamma + beta  (post-LN block end). N == 768. */
int crh_gemm_bf16_bias_res_ln(const void *x, const void *w, const float *bias, const void *residual,
                              const float *gamma, const float *beta, float eps, void *y, int T, int N,
                              int K, void *stream);

/* The same with the RESIDUAL STREAM IN F32 (ABI 4; opt-in fidelity lever: the reference's forward is fp32 throughout,
 * unixcoder_provider.py:137-155): y[T, 768] = bf16(LayerNorm(bf16(x @ w^T + bias) + residual_f32)) for the next GEMM, and
 * residual_f32[T, 768] is REPLACED by the same LayerNorm output in f32 -- the next residual.  N == 768. */
int crh_gemm_bf16_bias_res32_ln(const void *x, const void *w, const float *bias, float *residual_f32, const float *gamma, const float *beta,
                                float eps, void *y, int T, int N, int K, void *stream);

/* The same post-LN block with the LayerNorm FOLDED into the GEMMs around it (ABI 4; modeling_roberta.py:329-340,387-398 --
 * RobertaSelfOutput / RobertaOutput: dense, dropout, LayerNorm(hidden + input): the arithmetic these two entry points split
 * differently).  The residual stream stays UN-normalised between kernels; no [T, 768] tensor is read or written just to be
 * normalised (crh_encoder.hip, "LayerNorm folded into the GEMMs around it").
 *
 * crh_gemm_bf16_res_lnstats -- the producer (O-projection, FFN2): y[T, 768] = bf16(x @ w^T + bias + h), h = the previous
 *   LayerNorm's output worked out from ITS un-normalised rows: h = (residual * rstd + nmr) * res_gamma (+ its beta, which the
 *   caller folds into `bias`), with res_stats[T][2] = (rstd, nmr = -mu * rstd); res_stats == NULL: h = residual as it is.
 *   Also writes stats_out[T][2] = (rstd, nmr) of the rows of y as stored (what the next LayerNorm needs); `partials` is
 *   caller-provided scratch of T * (N / 32) * 2 floats.  y must not alias residual.  N == 768.
 * crh_gemm_bf16_lnin -- the consumer (QKV, FFN1): y[T, N] = act(LayerNorm(x) @ w^T + bias) computed as
 *   rstd * (x @ w_scaled^T - mu * colsum) + bias_folded, with x the un-normalised rows, row_stats[T][2] = (rstd, nmr) from the
 *   call above, w_scaled[N, K] = w * gamma (per k, rounded to bf16), colsum[N] = sum_k w_scaled[n][k] (of the ROUNDED values),
 *   bias_folded[N] = bias + w @ beta. */
int crh_gemm_bf16_res_lnstats(const void *x, const void *w, const float *bias, const void *residual, const float *res_stats,
                              const float *res_gamma, float eps, void *y, float *partials, float *stats_out, int T, int N, int K,
                              void *stream);
int crh_gemm_bf16_lnin(const void *x, const float *row_stats, const void *w_scaled, const float *colsum, const float *bias_folded,
                       void *y, int T, int N, int K, int act, void *stream);
/* y[T, 768] = bf16((x * rstd + nmr) * gamma + beta): the LayerNorm output itself from statistics already known (the last layer of a
 * folded forward: the masked mean pool, unixcoder_provider.py:152-154, reads normalised rows).  y may be x. */
int crh_layernorm_apply(const void *x, const float *row_stats, const float *gamma, const float *beta, void *y, int T, int N, void *stream);

/* Bidirectional self-attention with key masking.  qkv [B*L, 3*H*64] bf16 exactly as the QKV GEMM writes it
 * (q | k | v thirds, head-major inside each), out [B*L, H*64] bf16.  kmask: uint64 [B, ceil(L/64)], bit j of word t
 * set when token 64t+j of the row is a real token (ids != pad) -- the reference's `mask` (unixcoder_provider.py:148).
 * L % 16 == 0 (padding granularity of a length bucket), L <= 512. */
int crh_attn_fwd_varlen(const void *qkv, const uint64_t *kmask, void *out, int B, int L, int H, void *stream);

/* out[b, t, :] = LN((word[ids[b,t]] + type[0]) + pos[pos_id]); pos_id = cumsum(ids != pad) * (ids != pad) + pad.
 * ids int32 [B, L]; tables bf16; out bf16 [B, L, 768].  Also writes kmask (see above).
 * The library is not told the table heights: every id must index a row of `word`, and `pos` must hold
 * pad_id + L + 1 rows -- the caller checks (the Python driver does, encoder.py `_check_ids`). */
int crh_embed_ln(const int32_t *ids, const void *word, const void *pos, const void *type0,
                 const float *gamma, const float *beta, float eps, int pad_id, void *out,
                 uint64_t *kmask, int B, int L, int D, void *stream);

/* sent[b, :] = sum over real tokens of tok[b, t, :] / #real tokens  (f32 out, no L2 normalisation). */
int crh_masked_mean_pool(const void *tok, const uint64_t *kmask, float *sent, int B, int L, int D,
                         void *stream);

/* Packed rows: the same three kernels on a batch WITHOUT padding.  Row b of the batch is the tokens
 * [row_off[b], row_off[b+1]) of one flat token axis of T tokens (ids int32 [T], activations [T, ...]); Lmax (a multiple of
 * 16, >= every row's length, <= 512) sizes the key-mask stride (ceil(Lmax/64) words per row) and the launch.  The GEMM /
 * LayerNorm entry points above take T tokens as they are.  What it buys: the padded form rounds every row up to its bucket's
 * length (a multiple of 16) -- ~4 % of the tokens of a mean-200 mix -- and every kernel of the forward pays for them.
 * row_off lives on the device, so the library cannot look at it when a call is made: every kernel CLAMPS what it reads from
 * it to the T tokens the buffers hold (a bad offsets array can produce wrong rows, never an access outside the buffers), and
 * crh_embed_ln_packed -- the first call of a forward -- also launches a check of the whole array (row_off[0] == 0,
 * non-decreasing, every row <= Lmax, row_off[B] == T) whose verdict is reported as CRH_E_INVALID by the next packed entry
 * point called after the check has run, and in any case by crh_encoder_finish. */
int crh_embed_ln_packed(const int32_t *ids, const int32_t *row_off, const void *word, const void *pos, const void *type0,
                        const float *gamma, const float *beta, float eps, int pad_id, void *out, uint64_t *kmask, int B,
                        int T, int Lmax, int D, void *stream);
int crh_attn_fwd_packed(const void *qkv, const int32_t *row_off, const uint64_t *kmask, void *out, int B, int T, int Lmax,
                        int H, void *stream);
int crh_masked_mean_pool_packed(const void *tok, const int32_t *row_off, const uint64_t *kmask, float *sent, int B, int T,
                                int Lmax, int D, void *stream);
/* Waits for `stream` and returns the verdict of the device-side checks of the packed entry points launched on this device
 * since the last call (CRH_OK, or CRH_E_INVALID with the reason in crh_last_error) -- the sync point of a forward. */
int crh_encoder_finish(void *stream);

/* ------------------------------------------------------------- debug build only --------- */
/* Exported ONLY by libcoderag_hip_debug.so (code-rag_amd/build.sh compiles the same sources a second time with
 * -DCRH_ENABLE_DEBUG for tools/ and the kernel-selection tests); the product library has no crh_debug_* symbol. */
#ifdef CRH_ENABLE_DEBUG
/* Measurement support: launches the main scan's loads alone (same grid, same tile walk, same nt loads; no MFMA, no
 * candidates) over the whole corpus -- the HBM read rate this access pattern reaches on the device at hand, the ceiling
 * the scan's achieved GB/s is to be read against (tools/read_ceiling.py). */
int crh_debug_read_ceiling(crh_index *h, void *stream);

/* Measurement support: moves the int8 nomination copy of an index to a fresh device allocation (taken before the old one is
 * released) and marks it for requantisation -- lets one index try several places in HBM (tools/i8_places.py). */
int crh_debug_i8_move(crh_index *h);

/* Timing ablations of the GEMM main loop (variant 0 = the real kernel; others skip a pipeline stage and return
 * garbage).  Development aid used by tools/gemm_ablate.py; not part of the product path. */
int crh_debug_gemm_variant(const void *x, const void *w, const float *bias, void *y, int T, int N, int K,
                           int variant, void *stream);

#endif /* CRH_ENABLE_DEBUG */

#ifdef __cplusplus
}
#endif
#endif /* CODERAG_HIP_H */
