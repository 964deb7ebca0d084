#ifndef PB_SCAN_H
#define PB_SCAN_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
// Exclusive prefix sum of n 32-bit counts; d_tmp holds pb_scan_tmp_words(n) words,
// *d_total receives the sum.  d_in and d_out may be the same array.
size_t pb_scan_tmp_words(int n);
int pb_exclusive_scan_u32(const uint32_t* d_in, uint32_t* d_out, int n, uint32_t* d_total, uint32_t* d_tmp,
                          hipStream_t s);
#endif
