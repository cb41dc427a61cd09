// Probe: issue rate of VALU instruction kinds on gfx950 as a function of waves per SIMD, with the
// placement (SIMD id from HW_ID) and the cycle span of every wave of workgroup 0.
// 256 workgroups (one per CU) x T threads; each wave runs REPS x 16 independent instructions of one kind.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int T>
__global__ __launch_bounds__(T) void probe(float *out, long long *rec, int reps, float s) {
  const long long t0 = clock64();
  float a[16];
  f32x2 p[16];
  for (int i = 0; i < 16; ++i) {
    a[i] = threadIdx.x * 0.001f + i;
    p[i] = f32x2{a[i], a[i] + 0.5f};
  }
  const f32x2 ss = {s, s};
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(ss));
      if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(ss));
      if (KIND == 4) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 5) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s));
      if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 7) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 8) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 9) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 10) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 11) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 12) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 13) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : "vcc");
      if (KIND == 14) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 15) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
      if (KIND == 16) asm volatile("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 17) asm volatile("v_max_i32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (KIND == 18) asm volatile("v_add_f32_e64 %0, %0, |%0|" : "+v"(a[i]));               // x + |x| = 2 relu(x)
      if (KIND == 19) asm volatile("v_max_i32 %0, 0, %0" : "+v"(a[i]));                        // relu_bits
      if (KIND == 20) asm volatile("v_max_f32 %0, 0, %0" : "+v"(a[i]));
      if (KIND == 21) asm volatile("v_mul_f32_e64 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(s));
      if (KIND == 22) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    }
  }
  float acc = 0.f;
  for (int i = 0; i < 16; ++i) acc += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * T + threadIdx.x] = acc;
  if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    rec[2 * (threadIdx.x >> 6)] = clock64() - t0;
    rec[2 * (threadIdx.x >> 6) + 1] = hw;
  }
}

static float *dout = nullptr;
static long long *drec = nullptr;

template <int KIND, int T>
double run1(int grid, bool verbose) {
  if (!dout) { (void)hipMalloc(&dout, 512 * 1024 * 4); (void)hipMalloc(&drec, 16 * 16); }
  const int reps = 16384;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float ms = 0;
  for (int it = 0; it < 2; ++it) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((probe<KIND, T>), dim3(grid), dim3(T), 0, 0, dout, drec, reps, 1.0001f);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b);
  }
  if (verbose) {
    long long rec[32];
    (void)hipMemcpy(rec, drec, 16 * 16, hipMemcpyDeviceToHost);
    printf("    T=%d grid=%d kernel %.0f us; workgroup 0 waves (simd: Mcycles):", T, grid, ms * 1e3);
    for (int w = 0; w < T / 64; ++w) printf(" %lld:%.1f", (rec[2 * w + 1] >> 4) & 3, rec[2 * w] * 1e-6);
    printf("\n");
  }
  const double inst = (double)(grid / 256) * (T / 256) * reps * 16;   // per SIMD
  return ms * 1e-3 * 2.4e9 / inst;
}

template <int KIND>
void run(const char *name, bool verbose = false) {
  const double c1 = run1<KIND, 256>(256, verbose), c2 = run1<KIND, 512>(256, verbose);
  const double c4 = run1<KIND, 1024>(256, verbose), c8 = run1<KIND, 1024>(512, false);
  printf("%-28s waves/SIMD 1: %5.2f  2: %5.2f  4: %5.2f  8: %5.2f  cycles/instruction/SIMD\n", name, c1, c2, c4, c8);
}

int main() {
  run<0>("v_mul_f32", true);
  run<4>("v_min_u32", true);
  run<2>("v_add_f32");
  run<15>("v_sub_f32");
  run<5>("v_sub_f32 (sgpr src0, e32)");
  run<9>("v_max_f32");
  run<10>("v_fmac_f32");
  run<6>("v_fma_f32");
  run<1>("v_pk_mul_f32");
  run<3>("v_pk_fma_f32");
  run<7>("v_max_i32");
  run<14>("v_max_u32");
  run<8>("v_med3_i32");
  run<11>("v_and_b32");
  run<12>("v_mov_b32");
  run<13>("v_cmp + v_cndmask (pair)");
  run<16>("v_mov_b32_dpp row_ror");
  run<17>("v_max_i32_dpp row_ror");
  run<18>("v_add_f32 x, x, |x| (e64)");
  run<19>("v_max_i32 x, 0, x");
  run<20>("v_max_f32 x, 0, x");
  run<21>("v_mul_f32 ... clamp (e64)");
  run<22>("v_exp_f32");
  return 0;
}
