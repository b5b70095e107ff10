// runtime.hip -- error text, raw device-memory helpers and HIP-event timers of the C-ABI.
#include "mm_common.h"

thread_local char mm_err_buf[512] = "";

struct MmTimer {
  hipEvent_t a, b;
};

extern "C" {

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
