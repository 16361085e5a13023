// The one library call on a product path: rocPRIM's device radix sort (through hipCUB) of the cloud's (key, index) pairs for the
// sorted sweep of sor_knn_mean_kernel (cloud_kernels.hip, clouds from 4096 points on).  In a translation unit of its own: the sort's
// many kernel instantiations make a code object that takes 3.5 ms to load, and a code object is loaded with the first launch of ANY
// of its kernels -- the few-thousand-point clouds of a small reconstruction never sort, and no longer pay for it.
#include <hipcub/hipcub.hpp>

#include "common.hpp"

namespace esfm {

int sor_sort_scratch_bytes(int n, size_t *bytes, hipStream_t st)
{
    *bytes = 0;
    ESFM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, *bytes, (const float *)nullptr, (float *)nullptr, (const int32_t *)nullptr, (int32_t *)nullptr, n, 0, 32, st));
    return ESFM_OK;
}

int sor_sort_pairs(void *tmp, size_t tmp_bytes, const float *keys_in, float *keys_out, const int32_t *idx_in, int32_t *idx_out, int n, hipStream_t st)
{
    ESFM_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys_in, keys_out, idx_in, idx_out, n, 0, 32, st));
    return ESFM_OK;
}

}  // namespace esfm
