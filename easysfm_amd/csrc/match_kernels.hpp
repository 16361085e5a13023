// Launch interface between match_api.cpp (host logic) and match_kernels.hip (gfx950 kernels).
#pragma once

#include "common.hpp"

namespace esfm {

// One image pair of the pair loop (cpp_code/test/sfm.cpp:140-161), rows counted in the
// concatenated descriptor buffer.
struct PairDesc {
    int32_t q_row0, nq;   // query set: first row, row count
    int32_t t_row0, nt;   // train set
    int64_t out_off;      // first output slot of this pair (exclusive prefix sum of nq)
    int32_t blk_off;      // first workgroup of this pair in the knn launch
    int32_t blk_off2;     // ... in the launch of the one-product front pass (l2_x1_query_block() queries per workgroup)
};

int launch_l2_norms(hipStream_t st, const float *desc, int dim, long long n_rows, float *norms);
bool l2_mfma_supported(int dim);
int launch_l2_knn_mfma(hipStream_t st, int dim, const float *desc, const float *norms, const PairDesc *pairs, int n_pairs,
                       int n_blocks, int32_t *knn_idx, float *knn_dist, int32_t *flagged, int32_t *counters, int flag_cap);
// 64-float descriptors: split-bf16 distance pass (needs l2_split_bytes(total_rows) of scratch; 256 queries per workgroup)
bool l2_bf16_pass(int dim);
int l2_query_block(int dim);
size_t l2_split_bytes(int dim, long long total_rows);
// writes the hi/lo image AND the row norms
// also zeroes counters[0..16), pair_cnt[0..n_pairs) and (if given) pair_cnt2[0..n_pairs).
// hi != NULL: also the one-product pass's images and residual norms (l2_hi_bytes(total_rows) of scratch, layout l2_hi_*() below)
int launch_l2_split_bf16(hipStream_t st, const float *desc, long long total_rows, void *split, float *norms, int32_t *counters,
                         int32_t *pair_cnt, int n_pairs, void *hi, int32_t *pair_cnt2);
// three-product pass (round 2's kernel: the fallback for train sets beyond the one-product pass's position code, ESFM_L2_PASS=bf16x3)
int launch_l2_knn_bf16(hipStream_t st, const float *desc, const void *split, long long total_rows, const float *norms, const PairDesc *pairs,
                       int n_pairs, int n_blocks, int32_t *knn_idx, float *knn_dist, int32_t *flagged, int32_t *counters, int flag_cap,
                       int32_t *pair_cnt, int32_t *pair_list);
// one-product pass (64-float rows): distance pass + ratio screen.  Queries that provably fail d0 < ratio d1 get train index -2 if
// `markers` is set -- otherwise nothing: l2_finish_kernel's ratio stage walks the survivors only -- (+inf:
// screen off); every other query leaves a survivor entry (l2_survivor_entry_bytes() each, pair p's at surv_list + out_off[p]
// entries, surv_cnt[p] of them; counters[2] counts them).  rejected != NULL (audit): the screen's rejections on that list, counters[0].
// zero_a / zero_b [0, zero_n), zero_counters[0, 16): the other phase's per-pair and global counters, zeroed for the next call.
size_t l2_hi_bytes(long long total_rows);
size_t l2_survivor_entry_bytes();
// blk_pair[b]: the pair of the launch's b-th 512-query block (pair p owns blocks [blk_off2[p], blk_off2[p + 1])); the grid is
// one workgroup per block (ESFM_X1_GRID: that many persistent workgroups instead -- measurements).
int launch_l2_knn_bf16x1(hipStream_t st, int num_cu, const float *desc, const void *hi, long long total_rows, const float *norms, const PairDesc *pairs,
                         const int32_t *blk_pair, int n_blocks, int32_t *knn_idx, float *knn_dist, int32_t *counters, int flag_cap,
                         int32_t *surv_cnt, void *surv_list, double ratio, bool markers, int32_t *rejected,
                         int32_t *zero_a, int32_t *zero_b, int zero_n, int32_t *zero_counters);
// everything behind it in one launch (l2_finish_kernel): exact re-rank of the survivors (uncertified / undecided ones on unc_cnt /
// unc_list with their thresholds in knn_d2; counters[1] counts them), threshold-filter second pass, brute force of overflowed chunks
// (counters[0], listed in `flagged`), and -- do_ratio -- the ratio test + ordered compaction of every pair.  `done` zero on entry and
// on exit.  audit: 0 product path, 1 no brute force, 3 / 4 stop after the re-rank (its uncertified queries / the ratio verdicts'
// rejections appended to `flagged`)
int launch_l2_finish(hipStream_t st, const float *desc, const void *hi, long long total_rows, const float *norms, const PairDesc *pairs,
                     const int32_t *pair_order /* the pair indices sorted by train set */, int n_pairs, const int32_t *surv_cnt, const void *surv_list, int32_t *unc_cnt, int32_t *unc_list, float *knn_d2,
                     int32_t *knn_idx, float *knn_dist, int32_t *counters, int32_t *flagged, int flag_cap, int32_t *done,
                     int audit, bool do_ratio, double ratio, int32_t *query_idx, int32_t *train_idx, float *distance, int32_t *n_out);
int l2_x1_query_block();
bool l2_x1_supported(int max_nt);          // train sets the front pass's position code covers
bool l2_one_product_pass();   // ESFM_L2_PASS=bf16x3 in the environment switches the one-product front pass off (measurement)
// exact re-scan of the queries launch_l2_knn_bf16 binned per pair (pair_cnt[p] entries at pair_list[out_off[p]...])
int launch_l2_rescan64_pairs(hipStream_t st, const float *desc, const PairDesc *pairs, int n_pairs, const int32_t *pair_cnt,
                             const int32_t *pair_list, int32_t *knn_idx, float *knn_dist);
int launch_l2_exact_scan(hipStream_t st, int dim, const float *desc, const PairDesc *pairs, int n_pairs,
                         const int32_t *flagged, const int32_t *counters, long long total_queries, int grid,
                         int32_t *knn_idx, float *knn_dist);
bool hamming_supported(int nbytes);
int hamming_query_block(int nbytes);
// 256-bit descriptors take the i8-MFMA path, which needs hamming_expanded_bytes(total_rows) of scratch (exp_scratch);
// other widths run the XOR/popcount kernel and ignore it.
size_t hamming_expanded_bytes(int nbytes, long long total_rows);
int launch_hamming_expand(hipStream_t st, int nbytes, const void *desc, long long total_rows, void *exp_scratch);
// expanded: exp_scratch already holds this buffer's expansion (esfm_match_prepare_dev)
// 256-bit descriptors, train sets the position code can number (262 144 rows): the FP4-MFMA form (hamming_fp4_kernel: nibble-per-bit operands, the
// one-product L2 pass's main loop, exact tail, exact ratio screen -- queries that fail d0 < ratio d1 whatever d1 is get train index -2;
// +inf: none).  blk_pair / n_blocks: the 512-query block numbering (PairDesc::blk_off2).  ESFM_HM_PASS=i8 keeps the byte-per-bit kernel.
bool hamming_fp4_supported(int nbytes, int max_nt);
int launch_hamming_expand_fp4(hipStream_t st, const void *desc, long long total_rows, void *exp_scratch);
int launch_hamming_fp4(hipStream_t st, const void *desc, long long total_rows, void *exp_scratch, const PairDesc *pairs, const int32_t *blk_pair,
                       int n_blocks, int32_t *knn_idx, float *knn_dist, double ratio, bool expanded, int32_t *done, int n_pairs, int32_t *query_idx,
                       int32_t *train_idx, float *distance, int32_t *n_out);
// done != NULL (n_pairs zeroed counters, left zero): the launch also runs the ratio test + ordered compaction, pair by pair, in the
// workgroup that finishes a pair's last block (query_idx / train_idx / distance / n_out as for launch_ratio_compact)
int launch_hamming_knn(hipStream_t st, int nbytes, const void *desc, long long total_rows, void *exp_scratch,
                       const PairDesc *pairs, int n_pairs, int n_blocks,
                       int32_t *knn_idx, float *knn_dist, bool expanded);
int launch_buffer_checksum(hipStream_t st, const void *buf, size_t bytes, unsigned long long *out);
int launch_pack_match_lists(hipStream_t st, const long long *tab, const int32_t *n_out, int n_pairs, const int32_t *sq, const int32_t *stn, const float *sd,
                            int32_t *dq, int32_t *dtn, float *dd);
int launch_ratio_compact(hipStream_t st, const PairDesc *pairs, int n_pairs, const int32_t *knn_idx, const float *knn_dist,
                         double ratio, int32_t *query_idx, int32_t *train_idx, float *distance, int32_t *n_out);

}  // namespace esfm
