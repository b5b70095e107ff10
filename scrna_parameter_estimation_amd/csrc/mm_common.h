// mm_common.h -- shared helpers for libmemento_hip.so (gfx950 only; no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/memento_hip.h"

#define MM_WAVE 64
#define MM_OK 0
#define MM_ERR_HIP (-1)
#define MM_ERR_ARG (-2)

extern thread_local char mm_err_buf[512];

static inline int mm_fail(int code, const char *fmt, const char *a, const char *b, int line) {
  snprintf(mm_err_buf, sizeof(mm_err_buf), fmt, a, b, line);
  return code;
}

#define MM_HIP(call)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) return mm_fail(MM_ERR_HIP, "HIP error '%s' in %s (line %d)", hipGetErrorString(e_), #call, __LINE__); \
  } while (0)

#define MM_ARG(cond)                                                                                   \
  do {                                                                                                 \
    if (!(cond)) return mm_fail(MM_ERR_ARG, "bad argument: %s%s (line %d)", #cond, "", __LINE__);       \
  } while (0)

#define MM_LAUNCH_CHECK() MM_HIP(hipGetLastError())

// SELL layout constants: 64 genes per slice, 4 consecutive j's packed per lane (one dwordx4 per lane)
#define MM_SLICE 64
#define MM_JVEC 4
// a work item is at most this many dwordx4 rows of one slice (64 rows * 1 KiB = 64 KiB of entries)
#ifndef MM_ITEM_ROWS
#define MM_ITEM_ROWS 64
#endif

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int mm_lane() { return threadIdx.x & 63; }
