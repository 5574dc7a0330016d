/*
 * match_oracle.c -- TEST INFRASTRUCTURE (see oracle.h).  CPU restatement of
 *   cv2.BFMatcher(norm, crossCheck=True).match(desc1, desc2)
 *   (reference src/core/pose_estimator.py:131,144) followed by the reference's
 *   own Python post-processing: stable sort by distance (:147) and
 *   truncation to max_matches (:150-151).
 *
 * crossCheck semantics follow OpenCV core/batch_distance.cpp
 * (batchDistance(..., crosscheck=true)) as of the 4.5.x fix that every
 * opencv-python >= 4.8 (requirements.txt:1) carries -- two passes:
 *   batchDistance(src2, src1, tdist, tidx)   tidx[j] = nearest query of train j
 *   batchDistance(src1, src2, sdist, sidx)   sidx[i] = nearest train of query i
 * (K = 1, strict '<' while scanning ascending => lowest index on ties); then
 *   for j ascending: i = tidx[j]; if (tdist[j] < dist[i]) { dist[i] = tdist[j]; nidx[i] = j; }
 *   for i:           if (tidx[sidx[i]] != i) nidx[i] = -1;
 * i.e. a query keeps its best elector, and the match survives only if the
 * query's own nearest train elected it: strict mutual nearest neighbours with
 * lowest-index tie-breaks.  (If sidx[i] elected i, its distance is the row
 * minimum and it is the lowest such train, so the surviving match is always
 * (i, sidx[i]).)  Rounds 1-2 restated the older one-pass rule (electors only,
 * a superset); experiment knob 2 = 1 brings it back.  Matches are emitted in
 * ascending query index.  Python's sorted() is stable, so the final order is
 * (distance, queryIdx).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <limits.h>

int orc_debug_get_variant(int key);    /* orb_oracle.c */

static int hamming32(const uint8_t *a, const uint8_t *b)
{
    int d = 0;
    for (int k = 0; k < 32; ++k) d += __builtin_popcount((unsigned)(a[k] ^ b[k]));
    return d;
}

int orc_match_hamming(const uint8_t *d1, int n1, const uint8_t *d2, int n2,
                      int max_matches, int32_t *qidx, int32_t *tidx, int32_t *dist)
{
    if (n1 <= 0 || n2 <= 0) return 0;
    int *best_d = (int *)malloc(sizeof(int) * (size_t)n1);
    int *best_t = (int *)malloc(sizeof(int) * (size_t)n1);
    int *elect = (int *)malloc(sizeof(int) * (size_t)n2);       /* tidx */
    for (int i = 0; i < n1; ++i) { best_d[i] = INT_MAX; best_t[i] = -1; }
    for (int j = 0; j < n2; ++j) {
        int bi = -1, bd = INT_MAX;
        for (int i = 0; i < n1; ++i) {
            int d = hamming32(d1 + 32 * (size_t)i, d2 + 32 * (size_t)j);
            if (d < bd) { bd = d; bi = i; }
        }
        elect[j] = bi;
        if (bd < best_d[bi]) { best_d[bi] = bd; best_t[bi] = j; }
    }
    if (orc_debug_get_variant(2) == 0)
        for (int i = 0; i < n1; ++i) {                          /* sidx[i], then tidx[sidx[i]] == i */
            int bj = -1, bd = INT_MAX;
            for (int j = 0; j < n2; ++j) {
                int d = hamming32(d1 + 32 * (size_t)i, d2 + 32 * (size_t)j);
                if (d < bd) { bd = d; bj = j; }
            }
            if (elect[bj] != i) best_t[i] = -1;
        }
    free(elect);
    /* stable sort by distance == counting sort over 0..256 in ascending query index */
    int n = 0;
    int lim = max_matches >= 0 ? max_matches : INT_MAX;
    for (int d = 0; d <= 256 && n < lim; ++d)
        for (int i = 0; i < n1 && n < lim; ++i)
            if (best_t[i] >= 0 && best_d[i] == d) { qidx[n] = i; tidx[n] = best_t[i]; dist[n] = d; ++n; }
    free(best_d); free(best_t);
    return n;
}

typedef struct { float d; int q, t; } l2m;

static int cmp_l2(const void *a, const void *b)
{
    const l2m *x = (const l2m *)a, *y = (const l2m *)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return x->q - y->q; /* stability: ascending query index */
}

/* NORM_L2 on SIFT-style descriptors (integer-valued f32 in 0..255): the squared
 * distance is exact in int64; the reported distance is the f32 sqrt, and the
 * sort key is that f32 value (distinct integers may collide after sqrt). */
int orc_match_l2(const float *d1, int n1, const float *d2, int n2, int dim,
                 int max_matches, int32_t *qidx, int32_t *tidx, float *dist)
{
    if (n1 <= 0 || n2 <= 0) return 0;
    float *best_d = (float *)malloc(sizeof(float) * (size_t)n1);
    int *best_t = (int *)malloc(sizeof(int) * (size_t)n1);
    int *elect = (int *)malloc(sizeof(int) * (size_t)n2);       /* tidx */
    for (int i = 0; i < n1; ++i) { best_d[i] = INFINITY; best_t[i] = -1; }
    for (int j = 0; j < n2; ++j) {
        int bi = -1; float bd = INFINITY;
        for (int i = 0; i < n1; ++i) {
            const float *a = d1 + (size_t)dim * i, *b = d2 + (size_t)dim * j;
            float s = 0.f;
            for (int k = 0; k < dim; ++k) { float df = a[k] - b[k]; s += df * df; }
            float d = sqrtf(s);
            if (d < bd) { bd = d; bi = i; }
        }
        elect[j] = bi;
        if (bi >= 0 && bd < best_d[bi]) { best_d[bi] = bd; best_t[bi] = j; }
    }
    if (orc_debug_get_variant(2) == 0)
        for (int i = 0; i < n1; ++i) {                          /* sidx[i], then tidx[sidx[i]] == i */
            int bj = -1; float bd = INFINITY;
            for (int j = 0; j < n2; ++j) {
                const float *a = d1 + (size_t)dim * i, *b = d2 + (size_t)dim * j;
                float s = 0.f;
                for (int k = 0; k < dim; ++k) { float df = a[k] - b[k]; s += df * df; }
                float d = sqrtf(s);
                if (d < bd) { bd = d; bj = j; }
            }
            if (bj < 0 || elect[bj] != i) best_t[i] = -1;
        }
    free(elect);
    l2m *m = (l2m *)malloc(sizeof(l2m) * (size_t)n1);
    int n = 0;
    for (int i = 0; i < n1; ++i) if (best_t[i] >= 0) { m[n].d = best_d[i]; m[n].q = i; m[n].t = best_t[i]; ++n; }
    qsort(m, (size_t)n, sizeof(l2m), cmp_l2);
    if (max_matches >= 0 && n > max_matches) n = max_matches;
    for (int i = 0; i < n; ++i) { qidx[i] = m[i].q; tidx[i] = m[i].t; dist[i] = m[i].d; }
    free(m); free(best_d); free(best_t);
    return n;
}

/* ------------------------------------------------------------------ Lowe ratio (extension)
 * NOT in the reference (it uses crossCheck, pose_estimator.py:131); named by the project brief.  Restates
 * the usual cv2 idiom
 *     good = [m for m, n in BFMatcher(norm).knnMatch(desc1, desc2, k=2) if m.distance < ratio * n.distance]
 * followed by the reference's own sorted(key=distance)[:max_matches].  batchDistance keeps the K best trains
 * per query with a strict '<' insertion while scanning trains in ascending order: ties keep the lower train
 * index.  The ratio comparison is Python's: both distances as doubles.  Queries with fewer than two candidates
 * emit nothing. */
int orc_match_hamming_ratio(const uint8_t *d1, int n1, const uint8_t *d2, int n2, double ratio,
                            int max_matches, int32_t *qidx, int32_t *tidx, int32_t *dist)
{
    if (n1 <= 0 || n2 < 2) return 0;
    int *best_d = (int *)malloc(sizeof(int) * (size_t)n1);
    int *best_t = (int *)malloc(sizeof(int) * (size_t)n1);
    for (int i = 0; i < n1; ++i) {
        int b = INT_MAX, s = INT_MAX, bj = -1;
        for (int j = 0; j < n2; ++j) {
            int d = hamming32(d1 + 32 * (size_t)i, d2 + 32 * (size_t)j);
            if (d < b) { s = b; b = d; bj = j; }
            else if (d < s) s = d;
        }
        best_t[i] = ((double)b < ratio * (double)s) ? bj : -1;
        best_d[i] = b;
    }
    int n = 0;
    int lim = max_matches >= 0 ? max_matches : INT_MAX;
    for (int d = 0; d <= 256 && n < lim; ++d)
        for (int i = 0; i < n1 && n < lim; ++i)
            if (best_t[i] >= 0 && best_d[i] == d) { qidx[n] = i; tidx[n] = best_t[i]; dist[n] = d; ++n; }
    free(best_d); free(best_t);
    return n;
}

int orc_match_l2_ratio(const float *d1, int n1, const float *d2, int n2, int dim, double ratio,
                       int max_matches, int32_t *qidx, int32_t *tidx, float *dist)
{
    if (n1 <= 0 || n2 < 2) return 0;
    l2m *m = (l2m *)malloc(sizeof(l2m) * (size_t)n1);
    int n = 0;
    for (int i = 0; i < n1; ++i) {
        float b = INFINITY, s = INFINITY; int bj = -1;
        for (int j = 0; j < n2; ++j) {
            const float *a = d1 + (size_t)dim * i, *c = d2 + (size_t)dim * j;
            float acc = 0.f;
            for (int k = 0; k < dim; ++k) { float df = a[k] - c[k]; acc += df * df; }
            float d = sqrtf(acc);
            if (d < b) { s = b; b = d; bj = j; }
            else if (d < s) s = d;
        }
        if ((double)b < ratio * (double)s) { m[n].d = b; m[n].q = i; m[n].t = bj; ++n; }
    }
    qsort(m, (size_t)n, sizeof(l2m), cmp_l2);
    if (max_matches >= 0 && n > max_matches) n = max_matches;
    for (int i = 0; i < n; ++i) { qidx[i] = m[i].q; tidx[i] = m[i].t; dist[i] = m[i].d; }
    free(m);
    return n;
}
