// Launch interface between orb_api.cpp and orb_kernels.hip (ORB detection + description).
#pragma once

#include "common.hpp"

namespace esfm {

constexpr int kOrbLevels = 8, kOrbEdge = 31, kOrbHalfPatch = 15, kOrbFastThreshold = 20;

struct OrbLevels {           // the pyramid: level l is rows[l] x cols[l] bytes at offset[l] of the level buffers
    int32_t rows[kOrbLevels], cols[kOrbLevels];
    int64_t offset[kOrbLevels];
    int64_t total;
};
struct OrbCand { int32_t x, y, level; float resp; };                  // level coordinates
struct OrbKp { int32_t cx, cy, level, pad; float a, b; };              // descriptor centre (level coordinates), cos / sin of the angle
struct OrbTables { int32_t umax[kOrbHalfPatch + 2]; int32_t gauss[7]; int8_t pattern[1024]; };

int launch_orb_resize(hipStream_t st, const uint8_t *src, int sr, int sc, uint8_t *dst, int dr, int dc);
int launch_orb_blur(hipStream_t st, const OrbTables *tab, const uint8_t *src, int rows, int cols, uint8_t *dst);
int launch_orb_fast(hipStream_t st, const OrbLevels &L, const uint8_t *pyr, uint8_t *score, esfm_ctx *timing_ctx);
int launch_orb_nms(hipStream_t st, const OrbLevels &L, const uint8_t *score, OrbCand *cand, int32_t *n_cand, int cap);
int launch_orb_harris(hipStream_t st, const OrbLevels &L, const uint8_t *pyr, OrbCand *cand, int n);
int launch_orb_angles(hipStream_t st, const OrbLevels &L, const OrbTables *tab, const uint8_t *pyr, const OrbCand *cand, int n, float *angles);
int launch_orb_describe(hipStream_t st, const OrbLevels &L, const OrbTables *tab, const uint8_t *blurred, const OrbKp *kps, int n, uint8_t *desc);

}  // namespace esfm
