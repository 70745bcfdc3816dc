/*
 * sort.hip - key/value radix sort of the packet ids by direction key.
 * Separate translation unit: the rocPRIM templates are slow to compile and do
 * not depend on the engine's kernels.
 */
#include "sort.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

hipError_t cmi_sort_pairs_temp_bytes(size_t n, int end_bit, size_t *bytes) {
  return rocprim::radix_sort_pairs(nullptr, *bytes, (uint32_t *)nullptr,
                                   (uint32_t *)nullptr, (uint32_t *)nullptr,
                                   (uint32_t *)nullptr, n, 0, end_bit);
}

hipError_t cmi_sort_pairs(void *temp, size_t temp_bytes,
                          const uint32_t *keys_in, uint32_t *keys_out,
                          const uint32_t *values_in, uint32_t *values_out,
                          size_t n, int end_bit, hipStream_t stream) {
  return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out,
                                   values_in, values_out, n, 0, end_bit,
                                   stream);
}
