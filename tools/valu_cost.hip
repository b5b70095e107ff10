// Issue cost of the vector instructions the replay bootstrap is made of (gfx950): cycles per wave-instruction with one, two
// and three waves per SIMD, independent instructions (8 accumulators), measured with s_memtime around 4096 instructions.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_cost tools/valu_cost.hip && /tmp/valu_cost
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(64) void k_cost(uint64_t *out, uint32_t seed, int iters) {
  uint32_t a[8], b[8];
  uint64_t w[8];
  double d[8];
  float f[8];
  for (int i = 0; i < 8; i++) {
    a[i] = seed * (i + 3) + threadIdx.x;
    b[i] = seed ^ (i * 77 + 5);
    w[i] = ((uint64_t)a[i] << 32) | b[i];
    d[i] = 1.0 + 1e-3 * (double)(a[i] & 1023);
    f[i] = 1.0f + 1e-3f * (float)(b[i] & 1023);
  }
  uint32_t k = seed | 1;
  double dk = 1.0000001;
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(k));
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "s"(k));
#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(b[i]), "s"(k) : "vcc");
#define MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "s"(k));
#define MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "s"(k));
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "s"(k));
#define ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "s"(k));
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "s"(k));
#define LSHLADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
#define LSHL64(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(w[i]));
#define LSHR64(i) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(w[i]) : "v"(b[i]));
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(dk));
#define MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dk));
#define ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dk));
#define RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
#define CVT64U(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
#define CVT32_64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
#define LDEXP64(i) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(d[i]));
#define FMA32(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i]));
#define EXP32(i) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i]));
#define RCP32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
#define CMP64(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(d[i]), "v"(dk) : "vcc");
#define CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
#define FLOOR64(i) asm volatile("v_floor_f64 %0, %0" : "+v"(d[i]));
#define CVTI64(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
#define PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[i]));
#define XOR32(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b[i]));
#define DO(M) REP8(M) REP8(M) REP8(M) REP8(M)
    if (OP == 0) { DO(MULLO) }
    if (OP == 1) { DO(MULHI) }
    if (OP == 2) { DO(MAD64) }
    if (OP == 3) { DO(MUL24) }
    if (OP == 4) { DO(MULHI24) }
    if (OP == 5) { DO(MAD24) }
    if (OP == 6) { DO(ADD32) }
    if (OP == 7) { DO(ADD3) }
    if (OP == 8) { DO(LSHLADD64) }
    if (OP == 9) { DO(LSHL64) }
    if (OP == 10) { DO(LSHR64) }
    if (OP == 11) { DO(FMA64) }
    if (OP == 12) { DO(MUL64) }
    if (OP == 13) { DO(ADD64) }
    if (OP == 14) { DO(RCP64) }
    if (OP == 15) { DO(CVT64U) }
    if (OP == 16) { DO(CVT32_64) }
    if (OP == 17) { DO(LDEXP64) }
    if (OP == 18) { DO(FMA32) }
    if (OP == 19) { DO(EXP32) }
    if (OP == 20) { DO(RCP32) }
    if (OP == 21) { DO(CMP64) }
    if (OP == 22) { DO(CNDMASK) }
    if (OP == 23) { DO(FLOOR64) }
    if (OP == 24) { DO(CVTI64) }
    if (OP == 25) { DO(PKFMA32) }
    if (OP == 26) { DO(XOR32) }
    if (OP == 27) { DO(ALIGNBIT) }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint64_t acc = 0;
  for (int i = 0; i < 8; i++) acc += a[i] + w[i] + (uint64_t)d[i] + (uint64_t)f[i];
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) == 0) {
    out[wave * 2] = t1 - t0;
    out[wave * 2 + 1] = acc;
  }
}

static const char *NAMES[] = {"v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24", "v_mul_hi_u32_u24", "v_mad_u32_u24",
                              "v_add_u32", "v_add3_u32", "v_lshl_add_u64", "v_lshlrev_b64", "v_lshrrev_b64", "v_fma_f64", "v_mul_f64",
                              "v_add_f64", "v_rcp_f64", "v_cvt_f64_u32", "v_cvt_f32_f64", "v_ldexp_f64", "v_fma_f32", "v_exp_f32",
                              "v_rcp_f32", "v_cmp_lt_f64", "v_cndmask_b32", "v_floor_f64", "v_cvt_i32_f64", "v_pk_fma_f32", "v_xor_b32",
                              "v_alignbit_b32"};

template <int OP>
void run_one(uint64_t *d_out, std::vector<uint64_t> &h, double ghz_guess) {
  const int iters = 128;   // x 32 instructions
  double res[3];
  int cfg = 0;
  for (int waves_per_simd : {1, 2, 3}) {
    int blocks = 256 * 4 * waves_per_simd;   // one 64-thread block per wave slot, every SIMD of the chip
    hipLaunchKernelGGL(k_cost<OP>, dim3(blocks), dim3(64), 0, 0, d_out, 12345u, iters);
    hipLaunchKernelGGL(k_cost<OP>, dim3(blocks), dim3(64), 0, 0, d_out, 12345u, iters);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, blocks * 16, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; i++) s += (double)h[i * 2];
    res[cfg++] = s / blocks / (iters * 32.0);
  }
  printf("%-18s  cycles per wave-instruction: 1 wave/SIMD %.2f, 2: %.2f, 3: %.2f\n", NAMES[OP], res[0], res[1], res[2]);
}

template <int OP>
struct Runner {
  static void go(uint64_t *d, std::vector<uint64_t> &h) {
    run_one<OP>(d, h, 0);
    Runner<OP + 1>::go(d, h);
  }
};
template <>
struct Runner<28> {
  static void go(uint64_t *, std::vector<uint64_t> &) {}
};

int main() {
  uint64_t *d_out;
  hipMalloc(&d_out, 1 << 20);
  std::vector<uint64_t> h(1 << 17);
  printf("(cycles = s_memtime ticks, the shader clock counter)\n");
  Runner<0>::go(d_out, h);
  return 0;
}
