// runtime.hip -- error text, raw device-memory helpers and HIP-event timers of the C-ABI.
#include "mm_common.h"

thread_local char mm_err_buf[512] = "";

struct MmTimer {
  hipEvent_t a, b;
};

// Streaming-read probe: every wave reads 64 KiB chunks (one dwordx4 per lane per KiB row, 8 rows in flight -- the access
// pattern of k_moments1d_sell) and folds them into one word.  MODE 0: loads only -> the read bandwidth this device reaches for
// that pattern, i.e. the practical ceiling of the K1 kernel.  MODE 1: + K1's per-entry LDS gather (8-byte read at the entry's
// cell index).  MODE 2: + K1's fp64 arithmetic.  MODE 3: + K1's five per-(chunk, lane) result stores (sink must then hold
// 32 bytes per lane per chunk).  (tools/hbm_read_peak.py)
template <int MODE>
__global__ __launch_bounds__(1024) void k_read_probe(const u32x4 *__restrict__ src, int64_t n_chunks, uint32_t *__restrict__ sink) {
  __shared__ double w_lds[MODE ? 8192 : 1];
  if (MODE) {
    for (int i = threadIdx.x; i < 8192; i += 1024) w_lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
  }
  int lane = threadIdx.x & 63;
  int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  uint32_t acc = 0;
  double a1 = 0.0, a2 = 0.0, a3 = 0.0;
  for (int64_t c = wave; c < n_chunks; c += n_waves) {
    const u32x4 *p = src + c * 4096 + lane;
    if (MODE == 3) {
      a1 = a2 = a3 = 0.0;
      acc = 0;
    }
    for (int r = 0; r < 64; r += 8) {
      u32x4 e[8];
#pragma unroll
      for (int u = 0; u < 8; u++) e[u] = __builtin_nontemporal_load(p + (int64_t)(r + u) * 64);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        uint32_t q[4] = {e[u].x, e[u].y, e[u].z, e[u].w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if (MODE == 0) {
            acc ^= q[j];
          } else {
            double w = w_lds[q[j] & 8191];
            if (MODE == 1) {
              a1 += w;
            } else {  // MODE 2, 3
              double xd = (double)(q[j] >> 13), xw = xd * w, xw2 = xw * w;
              a1 += xw;
              a3 += xw2;
              a2 += xd * xw2;
              acc += q[j] >> 13;
            }
          }
        }
      }
    }
    if (MODE == 3) {  // K1's five result streams, 32 bytes per (chunk, lane)
      double *S1 = (double *)sink, *S2 = S1 + n_chunks * 64, *S3 = S2 + n_chunks * 64;
      uint32_t *SX = (uint32_t *)(S3 + n_chunks * 64), *MX = SX + n_chunks * 64;
      int64_t o = c * 64 + lane;
      S1[o] = a1;
      S2[o] = a2;
      S3[o] = a3;
      SX[o] = acc;
      MX[o] = acc >> 3;
    }
  }
  if (MODE != 3 && (acc == 0x9e3779b9u || a1 + a2 + a3 == 0.123)) sink[0] = acc;   // keeps the work alive, practically never taken
}

extern "C" {

int mm_debug_read_probe(const void *d_src, int64_t n_bytes, int32_t n_workgroups, int32_t mode, uint32_t *d_sink, void *stream) {
  MM_ARG(d_src && d_sink && n_bytes >= 65536 && n_workgroups > 0 && mode >= 0 && mode <= 3);
  auto kern = mode == 0 ? k_read_probe<0> : (mode == 1 ? k_read_probe<1> : (mode == 2 ? k_read_probe<2> : k_read_probe<3>));
  hipLaunchKernelGGL(kern, dim3((unsigned)n_workgroups), dim3(1024), 0, (hipStream_t)stream, (const u32x4 *)d_src, n_bytes / 65536,
                     d_sink);
  MM_LAUNCH_CHECK();
  return MM_OK;
}

const char *mm_last_error(void) { return mm_err_buf; }
int mm_version(void) { return 100; }

int mm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mm_set_device(int dev) {
  MM_HIP(hipSetDevice(dev));
  return MM_OK;
}

int mm_malloc(void **d_ptr, size_t bytes) {
  MM_ARG(d_ptr);
  MM_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
  return MM_OK;
}

int mm_free(void *d_ptr) {
  if (d_ptr) MM_HIP(hipFree(d_ptr));
  return MM_OK;
}

int mm_memset(void *d_ptr, int value, size_t bytes, void *stream) {
  if (bytes) MM_HIP(hipMemsetAsync(d_ptr, value, bytes, (hipStream_t)stream));
  return MM_OK;
}

int mm_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream) {
  if (bytes) MM_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  return MM_OK;
}

int mm_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream) {
  if (bytes) {
    MM_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    MM_HIP(hipStreamSynchronize((hipStream_t)stream));
  }
  return MM_OK;
}

int mm_sync(void *stream) {
  MM_HIP(hipStreamSynchronize((hipStream_t)stream));
  return MM_OK;
}

int mm_timer_create(void **timer) {
  MM_ARG(timer);
  MmTimer *t = new MmTimer;
  if (hipEventCreate(&t->a) != hipSuccess || hipEventCreate(&t->b) != hipSuccess) {
    delete t;
    return mm_fail(MM_ERR_HIP, "HIP error '%s' in %s (line %d)", "hipEventCreate", "mm_timer_create", __LINE__);
  }
  *timer = t;
  return MM_OK;
}

int mm_timer_begin(void *timer, void *stream) {
  MM_ARG(timer);
  MM_HIP(hipEventRecord(((MmTimer *)timer)->a, (hipStream_t)stream));
  return MM_OK;
}

int mm_timer_end(void *timer, void *stream) {
  MM_ARG(timer);
  MM_HIP(hipEventRecord(((MmTimer *)timer)->b, (hipStream_t)stream));
  return MM_OK;
}

int mm_timer_elapsed_ms(void *timer, float *ms) {
  MM_ARG(timer && ms);
  MmTimer *t = (MmTimer *)timer;
  MM_HIP(hipEventSynchronize(t->b));
  MM_HIP(hipEventElapsedTime(ms, t->a, t->b));
  return MM_OK;
}

int mm_timer_destroy(void *timer) {
  if (timer) {
    MmTimer *t = (MmTimer *)timer;
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
  }
  return MM_OK;
}

}  // extern "C"
