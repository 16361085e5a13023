// Launch interface between surf_api.cpp and surf_kernels.hip.
#pragma once

#include "common.hpp"

namespace esfm {

constexpr int kSurfOctaves = 4, kSurfOctaveLayers = 3, kSurfLayers = (kSurfOctaveLayers + 2) * kSurfOctaves;
constexpr int kSurfOriRadius = 6, kSurfPatch = 20, kSurfOriSamples = 113;   // grid points with i^2 + j^2 <= 36

struct SurfHF { int32_t p0, p1, p2, p3; float w; };

struct SurfLayer {          // one layer of the determinant pyramid
    int32_t size, step, rows, cols;          // filter size, sample step, layer dimensions
    int32_t samples_i, samples_j, margin;    // written region (calcLayerDetAndTrace)
    int32_t offset;                          // first float of this layer in the det / trace buffers
    int32_t valid, pad0, pad1, pad2;         // filter fits the image
    SurfHF dx[3], dy[3], dxy[4];
};

struct SurfParams {
    int32_t rows, cols;                      // image
    float hessian_threshold;
    int32_t max_candidates;
    SurfLayer layer[kSurfLayers];
};

struct SurfKeypoint { float x, y, size, angle, response; int32_t octave, class_id, valid; };

struct SurfDescTables {     // host-computed with the host's exp(), uploaded once per call
    int32_t n_ori;
    int32_t aptx[kSurfOriSamples], apty[kSurfOriSamples];
    float aptw[kSurfOriSamples];
    float DW[kSurfPatch * kSurfPatch];
};

int launch_surf_gray(hipStream_t st, const uint8_t *bgr, int n_pixels, uint8_t *gray);
int launch_surf_integral(hipStream_t st, const uint8_t *gray, int rows, int cols, int32_t *sum);
int launch_surf_det_trace(hipStream_t st, const SurfParams *params_dev, const SurfParams &params_host, const int32_t *sum, float *det, float *trace,
                          esfm_ctx *timing_ctx);
int launch_surf_maxima(hipStream_t st, const SurfParams *params_dev, const SurfParams &params_host, const float *det, const float *trace,
                       SurfKeypoint *cand, int32_t *n_cand);
// per-keypoint descriptor scratch: window bytes | 21 row sums per window row | (x, y) start of every window row
struct SurfBlk { int32_t k, first; };          // a 256-thread block of a keypoint's item list: the keypoint, the block's first item
constexpr int kSurfWinChunk = 48;              // samples of a window row per work item (about)
__host__ __device__ inline size_t surf_align16(size_t n) { return (n + 15) / 16 * 16; }
__host__ __device__ inline int surf_window_chunks(int win_size) { return win_size <= 0 ? 1 : (win_size + kSurfWinChunk - 1) / kSurfWinChunk; }
__host__ __device__ inline size_t surf_scratch_bytes(int win_size)
{
    return surf_align16((size_t)win_size * win_size) + surf_align16(sizeof(float) * (kSurfPatch + 1) * (size_t)win_size) + surf_align16(sizeof(float) * 2 * (size_t)win_size);
}
int launch_surf_describe(hipStream_t st, const SurfParams *params_dev, const SurfDescTables *tables_dev, const uint8_t *gray, const int32_t *sum,
                         SurfKeypoint *kps, int n_kp, const int64_t *win_offset, const SurfBlk *win_blocks, int n_win_blocks,
                         const SurfBlk *row_blocks, int n_row_blocks, uint8_t *win_scratch, float *desc, esfm_ctx *timing_ctx);

}  // namespace esfm
