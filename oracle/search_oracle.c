/*
 * oracle/search_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Scalar restatement of what the reference's vector store computes for one
 * cosine top-k query.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path (code-rag_amd/) never does.
 *
 * What it follows
 * ---------------
 * The reference never does this arithmetic itself; it delegates to a Qdrant
 * server through qdrant-client (pyproject.toml:10 `qdrant-client>=1.12.0`, no
 * lock file; server image `qdrant/qdrant:latest`, docker-compose.yml:37 --
 * both UNPINNED and absent from /root/reference and from this image).  Call
 * sites that fix the semantics:
 *   src/lattice/embeddings/client.py:96-102   collections use Distance.COSINE
 *   src/lattice/embeddings/client.py:127      upsert(points) -> normalise on insert
 *   src/lattice/embeddings/client.py:142-148  query_points(query, limit, filter)
 *   src/lattice/embeddings/client.py:171-176  filter = AND of MatchValue equalities
 * Qdrant's published scalar algorithm (lib/segment/src/spaces/simple.rs,
 * restated from the public source, not checkable offline):
 *   cosine_preprocess(v): len2 = sum_i v_i*v_i (f32, in index order);
 *       if len2 < f32::EPSILON or |len2 - 1| <= 1e-6 -> v unchanged
 *       else v_i / sqrt(len2)
 *   score(q, x) = sum_i q_i*x_i (f32, in index order) on the preprocessed
 *       vectors; results in descending score.
 * Tie order is unspecified upstream; this repo defines it: lower row first.
 *
 * PARITY UNPINNED at the Qdrant boundary: the reference holds no golden
 * vector for any score or order (tests/test_database.py:88-124 needs a live
 * server and only asserts ">= 1 hit").  The fixtures under tests/golden/ pin
 * THIS restatement (fp64 cross-check in oracle/search.py), nothing more.
 *
 * Build with -ffp-contract=off: every product and every sum below is a
 * separately rounded IEEE f32 operation, the same sequence the HIP rescoring
 * kernel executes (crh_search.hip: canonical_dot_*).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* f32 -> bf16 (round to nearest even) -> f32.  NaN stays NaN. */
float orc_bf16_round(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) {
        u |= 0x00400000u; /* quiet */
        u &= 0xffff0000u;
    } else {
        u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    }
    float r;
    memcpy(&r, &u, 4);
    return r;
}

/* Qdrant cosine_preprocess, scalar form (see header). */
void orc_cosine_preprocess(const float *v, float *out, int d)
{
    float len2 = 0.0f;
    for (int i = 0; i < d; ++i) {
        float p = v[i] * v[i];
        len2 = len2 + p;
    }
    if (len2 < FLT_EPSILON || fabsf(len2 - 1.0f) <= 1.0e-6f) {
        for (int i = 0; i < d; ++i) out[i] = v[i];
        return;
    }
    float len = sqrtf(len2);
    for (int i = 0; i < d; ++i) out[i] = v[i] / len;
}

void orc_preprocess_rows(const float *v, float *out, int64_t n, int d, int to_bf16)
{
    for (int64_t r = 0; r < n; ++r) {
        orc_cosine_preprocess(v + r * d, out + r * d, d);
        if (to_bf16)
            for (int i = 0; i < d; ++i) out[r * d + i] = orc_bf16_round(out[r * d + i]);
    }
}

/* Sequential f32 dot product: acc = acc + a_i*b_i, i ascending. */
float orc_dot(const float *a, const float *b, int d)
{
    float acc = 0.0f;
    for (int i = 0; i < d; ++i) {
        float p = a[i] * b[i];
        acc = acc + p;
    }
    return acc;
}

static int row_passes(int64_t r, const uint8_t *alive, const int32_t *codes, int ncols,
                      const int32_t *fcols, const int32_t *fvals, int nfilt)
{
    if (alive && !alive[r]) return 0;
    for (int f = 0; f < nfilt; ++f)
        if (codes[r * ncols + fcols[f]] != fvals[f]) return 0;
    return 1;
}

/* better(a,b): a ranks strictly before b -- higher score, then lower row. */
static int better(float sa, int64_t ra, float sb, int64_t rb)
{
    if (sa > sb) return 1;
    if (sa < sb) return 0;
    return ra < rb;
}

/*
 * Exact top-k of `nq` preprocessed queries over `n` preprocessed rows.
 * corpus: n x d f32 (already cosine_preprocess'ed; bf16-rounded when the
 * store keeps bf16).  alive: n bytes or NULL.  codes: n x ncols int32
 * dictionary codes or NULL; (fcols[f], fvals[f]) are AND-ed equalities
 * (client.py:171-176).  Output rows beyond the number of passing rows are
 * (-inf, -1).  Eight rows are scored together so the add chains pipeline;
 * each dot product keeps its own strictly sequential order.
 */
int orc_search(const float *corpus, int64_t n, int d, const uint8_t *alive,
               const int32_t *codes, int ncols, const int32_t *fcols, const int32_t *fvals,
               int nfilt, const float *queries, int nq, int k, float *out_scores,
               int64_t *out_rows)
{
    if (k <= 0 || d <= 0) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int q = 0; q < nq; ++q) {
        const float *qv = queries + (int64_t)q * d;
        float *bs = out_scores + (int64_t)q * k;
        int64_t *br = out_rows + (int64_t)q * k;
        int cnt = 0;
        for (int i = 0; i < k; ++i) {
            bs[i] = -INFINITY;
            br[i] = -1;
        }
        for (int64_t r0 = 0; r0 < n; r0 += 8) {
            int m = (n - r0) < 8 ? (int)(n - r0) : 8;
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const float *x = corpus + r0 * d;
            if (m == 8) {
                for (int i = 0; i < d; ++i) {
                    float qi = qv[i];
                    for (int j = 0; j < 8; ++j) {
                        float p = qi * x[(int64_t)j * d + i];
                        acc[j] = acc[j] + p;
                    }
                }
            } else {
                for (int j = 0; j < m; ++j) acc[j] = orc_dot(qv, x + (int64_t)j * d, d);
            }
            for (int j = 0; j < m; ++j) {
                int64_t r = r0 + j;
                if (!row_passes(r, alive, codes, ncols, fcols, fvals, nfilt)) continue;
                float s = acc[j];
                if (cnt == k && !better(s, r, bs[k - 1], br[k - 1])) continue;
                int pos = cnt < k ? cnt : k - 1;
                while (pos > 0 && better(s, r, bs[pos - 1], br[pos - 1])) {
                    bs[pos] = bs[pos - 1];
                    br[pos] = br[pos - 1];
                    --pos;
                }
                bs[pos] = s;
                br[pos] = r;
                if (cnt < k) ++cnt;
            }
        }
    }
    return 0;
}

/*
 * Merge `nlists` per-shard top-k lists (each nq x k, sorted, padded with
 * (-inf,-1)) into one nq x k list: same total order as orc_search.  This is the
 * step an 8-GPU row-sharded search performs after its all-gather.
 */
int orc_merge_topk(int nlists, int nq, int k, const float *scores, const int64_t *rows,
                   float *out_scores, int64_t *out_rows)
{
    for (int q = 0; q < nq; ++q) {
        float *bs = out_scores + (int64_t)q * k;
        int64_t *br = out_rows + (int64_t)q * k;
        int cnt = 0;
        for (int i = 0; i < k; ++i) {
            bs[i] = -INFINITY;
            br[i] = -1;
        }
        for (int l = 0; l < nlists; ++l) {
            const float *ls = scores + ((int64_t)l * nq + q) * k;
            const int64_t *lr = rows + ((int64_t)l * nq + q) * k;
            for (int i = 0; i < k; ++i) {
                if (lr[i] < 0) continue;
                float s = ls[i];
                int64_t r = lr[i];
                if (cnt == k && !better(s, r, bs[k - 1], br[k - 1])) continue;
                int pos = cnt < k ? cnt : k - 1;
                while (pos > 0 && better(s, r, bs[pos - 1], br[pos - 1])) {
                    bs[pos] = bs[pos - 1];
                    br[pos] = br[pos - 1];
                    --pos;
                }
                bs[pos] = s;
                br[pos] = r;
                if (cnt < k) ++cnt;
            }
        }
    }
    return 0;
}
