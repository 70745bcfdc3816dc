/*
 * sort.h - device radix sort of (direction key, packet id) pairs.
 */
#ifndef CMI_SORT_H
#define CMI_SORT_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

/* bytes of temporary device storage cmi_sort_pairs needs for n pairs */
hipError_t cmi_sort_pairs_temp_bytes(size_t n, int end_bit, size_t *bytes);

/* sort n (key, value) pairs by the low `end_bit` bits of the key */
hipError_t cmi_sort_pairs(void *temp, size_t temp_bytes,
                          const uint32_t *keys_in, uint32_t *keys_out,
                          const uint32_t *values_in, uint32_t *values_out,
                          size_t n, int end_bit, hipStream_t stream);

#endif
