/*
 * oracle/match_ref.c -- CPU restatement of the reference's pairwise matcher.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (easysfm_amd/, the
 * C-ABI library, bin/) may include, link, call or execute this file.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * (SURVEY.md section 4), and it cannot be built here (needs OpenCV 3.x +
 * contrib, Ceres, PCL).  The arithmetic lives in un-vendored OpenCV (author ran
 * 3.4.2, python_code/notebook/test_feature.ipynb cell 0).  This file restates
 *   - the control flow of cpp_code/src/feature_matching.cpp:71-97 (ORB) and
 *     :115-142 (SURF): knnMatch(query, train, nn2, 2), then keep nn2[i][0] iff
 *     nn2[i][0].distance < ratio_thre * nn2[i][1].distance  (:88, :133; the
 *     right-hand side is double*float, so the compare happens in double),
 *     survivors appended in query order (:90, :135);
 *   - the exact brute-force variant python_code/feature_match.py:31-40 uses
 *     for SURF (cv2.BFMatcher(NORM_L2).knnMatch(k=2)); the C++ SURF path uses
 *     the approximate FlannBasedMatcher (feature_matching.cpp:120), whose
 *     output is not a function of the inputs alone (randomised kd-trees), so
 *     "bit-exact" is only definable against exact brute force;
 *   - [upstream, from memory of OpenCV 3.4 modules/core/src/{stat,batch_distance}.cpp,
 *     NOT verifiable in this container]
 *       * L2 distance of one pair = sqrt(normL2Sqr_(a,b,n)) in float, where the
 *         SSE2-baseline normL2Sqr_ keeps two 4-lane accumulators over blocks
 *         of 8 elements (t=a-b; acc += t*t, separate multiply and add), adds
 *         the two accumulators lane-wise, sums the 4 lanes left to right and
 *         then adds the scalar tail sequentially;
 *       * Hamming distance = popcount(a xor b) over nbytes, as int, converted
 *         to float in DMatch.distance;
 *       * K-best selection per query row: scan train rows in ascending index,
 *         insert d iff d < current K-th best (strict), shifting entries that
 *         are > d.  Equal distances therefore keep the LOWER train index
 *         first, and a NaN distance is never inserted.
 *   - defined behaviour where the reference has undefined behaviour: with
 *     fewer than 2 train rows the reference indexes nn2[i][1] out of bounds
 *     (feature_matching.cpp:88,133); here such a query emits no match and
 *     knn2 reports index -1 / distance FLT_MAX for the missing neighbours.
 *
 * Build: see oracle/Makefile (gcc -O3 -ffp-contract=off; NO -ffast-math: the
 * summation order above is part of the contract).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* Canonical squared L2 distance (see header): 8 partial sums over blocks of 8,
 * (acc[c] + acc[c+4]) for c = 0..3, summed left to right, then the tail. */
float esfm_ref_l2sqr(const float *a, const float *b, int n)
{
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j <= n - 8; j += 8) {
        for (int c = 0; c < 8; ++c) {
            float t = a[j + c] - b[j + c];
            acc[c] = acc[c] + t * t;
        }
    }
    float s0 = acc[0] + acc[4];
    float s1 = acc[1] + acc[5];
    float s2 = acc[2] + acc[6];
    float s3 = acc[3] + acc[7];
    float d = s0 + s1;
    d = d + s2;
    d = d + s3;
    for (; j < n; ++j) {
        float t = a[j] - b[j];
        d = d + t * t;
    }
    return d;
}

static inline int popcount8(uint8_t x) { return __builtin_popcount((unsigned)x); }

int esfm_ref_hamming(const uint8_t *a, const uint8_t *b, int nbytes)
{
    int d = 0;
    for (int j = 0; j < nbytes; ++j) d += popcount8((uint8_t)(a[j] ^ b[j]));
    return d;
}

/* K = 2 insertion of (d, j) into (d0,i0),(d1,i1), OpenCV batchDistance rule. */
#define KNN2_INSERT(d, j, d0, i0, d1, i1) \
    do {                                  \
        if ((d) < (d1)) {                 \
            if ((d0) > (d)) {             \
                d1 = d0; i1 = i0;         \
                d0 = (d); i0 = (j);       \
            } else {                      \
                d1 = (d); i1 = (j);       \
            }                             \
        }                                 \
    } while (0)

/* knnMatch(query, train, out, 2) for NORM_L2 on float descriptors
 * (feature_matching.cpp:125 with exact BF; feature_match.py:33-34).
 * idx[2*i+k], dist[2*i+k] = k-th neighbour of query i; -1 / FLT_MAX if absent. */
void esfm_ref_knn2_l2_f32(const float *q, int nq, const float *t, int nt, int dim,
                          int32_t *idx, float *dist)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < nq; ++i) {
        float d0 = FLT_MAX, d1 = FLT_MAX;
        int32_t i0 = -1, i1 = -1;
        const float *a = q + (size_t)i * dim;
        for (int j = 0; j < nt; ++j) {
            float d = sqrtf(esfm_ref_l2sqr(a, t + (size_t)j * dim, dim));
            KNN2_INSERT(d, j, d0, i0, d1, i1);
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1;
        dist[2 * i] = d0; dist[2 * i + 1] = d1;
    }
}

/* knnMatch(query, train, out, 2) for "BruteForce-Hamming" (feature_matching.cpp:74,80).
 * Distances are exact ints, reported as float like cv::DMatch::distance. */
void esfm_ref_knn2_hamming(const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes,
                           int32_t *idx, float *dist)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < nq; ++i) {
        int d0 = INT32_MAX, d1 = INT32_MAX;
        int32_t i0 = -1, i1 = -1;
        const uint8_t *a = q + (size_t)i * nbytes;
        for (int j = 0; j < nt; ++j) {
            int d = esfm_ref_hamming(a, t + (size_t)j * nbytes, nbytes);
            KNN2_INSERT(d, j, d0, i0, d1, i1);
        }
        idx[2 * i] = i0; idx[2 * i + 1] = i1;
        dist[2 * i] = (i0 >= 0) ? (float)d0 : FLT_MAX;
        dist[2 * i + 1] = (i1 >= 0) ? (float)d1 : FLT_MAX;
    }
}

/* Lowe ratio filter, feature_matching.cpp:84-92 / :129-137.  Returns the number
 * of survivors, written in query order (DMatch.queryIdx, .trainIdx, .distance). */
int esfm_ref_ratio_filter(const int32_t *idx, const float *dist, int nq, double ratio,
                          int32_t *query_idx, int32_t *train_idx, float *distance)
{
    int n = 0;
    for (int i = 0; i < nq; ++i) {
        if (idx[2 * i] < 0 || idx[2 * i + 1] < 0) continue; /* reference: UB */
        if ((double)dist[2 * i] < ratio * (double)dist[2 * i + 1]) {
            query_idx[n] = i;
            train_idx[n] = idx[2 * i];
            distance[n] = dist[2 * i];
            ++n;
        }
    }
    return n;
}

/* matchFeaturesSURF with exact brute force (feature_matching.cpp:115-142). */
int esfm_ref_match_l2_f32(const float *q, int nq, const float *t, int nt, int dim, double ratio,
                          int32_t *query_idx, int32_t *train_idx, float *distance,
                          int32_t *scratch_idx /*2*nq*/, float *scratch_dist /*2*nq*/)
{
    esfm_ref_knn2_l2_f32(q, nq, t, nt, dim, scratch_idx, scratch_dist);
    return esfm_ref_ratio_filter(scratch_idx, scratch_dist, nq, ratio, query_idx, train_idx, distance);
}

/* matchFeaturesORB (feature_matching.cpp:71-97). */
int esfm_ref_match_hamming(const uint8_t *q, int nq, const uint8_t *t, int nt, int nbytes, double ratio,
                           int32_t *query_idx, int32_t *train_idx, float *distance,
                           int32_t *scratch_idx, float *scratch_dist)
{
    esfm_ref_knn2_hamming(q, nq, t, nt, nbytes, scratch_idx, scratch_dist);
    return esfm_ref_ratio_filter(scratch_idx, scratch_dist, nq, ratio, query_idx, train_idx, distance);
}

void esfm_ref_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int esfm_ref_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
