/*
 * oracle/cloud_ref.c -- CPU restatement of the reference's statistical outlier removal on the sparse cloud
 * (SURVEY.md section 8 row f-3).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include, link, call or execute this file
 * (see oracle/match_ref.c header).
 *
 * PARITY UNPINNED: the reference calls pcl::StatisticalOutlierRemoval<PointXYZRGB> with MeanK = 50 and
 * StddevMulThresh = 2.0 (cpp_code/include/cloudprocessing.hpp:24-36, called at cpp_code/test/sfm.cpp:333); PCL
 * (>= 1.7, unpinned, cpp_code/CMakeLists.txt:30) is absent here and the reference holds no fixture for it.  This file
 * restates the published algorithm [upstream pcl/filters/impl/statistical_outlier_removal.hpp applyFilterIndices,
 * pcl/kdtree/impl/kdtree_flann.hpp, flann L2_Simple], from memory:
 *   - the search structure holds the FINITE points only; search is exact (eps = 0);
 *   - per point i with finite coordinates: the mean_k + 1 nearest neighbours (the point itself is entry 0) by the
 *     float squared distance  ((dx*dx + dy*dy) + dz*dz)  (float accumulator, x then y then z, no fused multiply-add);
 *     dist_sum (double) += sqrt(nn_dists[k]) for k = 1..mean_k in ascending order, where the float overload of sqrt
 *     is taken (libstdc++ >= 6 <math.h> brings std::sqrt(float) into the global namespace);
 *     distances[i] = (float)(dist_sum / mean_k).  Non-finite points: distances[i] = 0, not counted;
 *   - sum (double) += distances[i]; sq_sum += distances[i] * distances[i] (float product), i ascending;
 *     mean = sum / valid; variance = (sq_sum - sum*sum/valid) / (valid - 1); threshold = mean + std_mul * sqrt(variance);
 *   - a point is removed iff distances[i] > threshold (float against double).
 * Fewer than mean_k + 1 finite points is undefined behaviour in PCL (reads past the result vector); here the sum
 * runs over the neighbours that exist and is still divided by mean_k.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static inline int finite3(const float *p) { return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]); }

/* max-heap of the m smallest values seen */
static inline void heap_sift_down(float *h, int m, int i)
{
    for (;;) {
        int l = 2 * i + 1, r = l + 1, b = i;
        if (l < m && h[l] > h[b]) b = l;
        if (r < m && h[r] > h[b]) b = r;
        if (b == i) return;
        float t = h[i]; h[i] = h[b]; h[b] = t; i = b;
    }
}

static int cmp_float(const void *a, const void *b)
{
    float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

/* points: n rows of `stride` floats, x y z first (stride 3 = packed, 8 = pcl::PointXYZRGB).
 * mean_dist[n] out.  Returns the number of valid (finite) points. */
int esfm_ref_sor_mean_distances(const float *points, int n, int stride, int mean_k, float *mean_dist)
{
    int valid = 0;
    for (int i = 0; i < n; ++i) valid += finite3(points + (size_t)i * stride);
    const int m = mean_k + 1;
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        float *heap = (float *)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
        for (int i = 0; i < n; ++i) {
            const float *q = points + (size_t)i * stride;
            if (!finite3(q)) { mean_dist[i] = 0.0f; continue; }
            int cnt = 0;
            for (int j = 0; j < n; ++j) {
                const float *p = points + (size_t)j * stride;
                if (!finite3(p)) continue;
                const float dx = q[0] - p[0], dy = q[1] - p[1], dz = q[2] - p[2];
                float d = dx * dx;
                d = d + dy * dy;
                d = d + dz * dz;
                if (cnt < m) {
                    heap[cnt++] = d;
                    if (cnt == m) for (int t = m / 2 - 1; t >= 0; --t) heap_sift_down(heap, m, t);
                } else if (d < heap[0]) { heap[0] = d; heap_sift_down(heap, m, 0); }
            }
            qsort(heap, (size_t)cnt, sizeof(float), cmp_float);
            double dist_sum = 0.0;
            for (int k = 1; k < cnt; ++k) dist_sum += (double)sqrtf(heap[k]);
            mean_dist[i] = (float)(dist_sum / (double)mean_k);
        }
        free(heap);
    }
    return valid;
}

/* keep[n] (1 = survives), *threshold; returns the number kept. */
int esfm_ref_sor_filter(const float *points, int n, int stride, int mean_k, double std_mul, float *mean_dist /*n, out*/,
                        uint8_t *keep, double *threshold)
{
    const int valid = esfm_ref_sor_mean_distances(points, n, stride, mean_k, mean_dist);
    double sum = 0.0, sq_sum = 0.0;
    for (int i = 0; i < n; ++i) { sum += (double)mean_dist[i]; sq_sum += (double)(mean_dist[i] * mean_dist[i]); }
    const double mean = sum / (double)valid;
    const double variance = (sq_sum - sum * sum / (double)valid) / ((double)valid - 1.0);
    const double thr = mean + std_mul * sqrt(variance);
    if (threshold) *threshold = thr;
    int kept = 0;
    for (int i = 0; i < n; ++i) { keep[i] = !((double)mean_dist[i] > thr); kept += keep[i]; }
    return kept;
}
