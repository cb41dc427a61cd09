// Probe: the shader clock the chip actually sustains while every SIMD runs the fp32 MFMA layer loop of the stack kernels
// (128 -> 128 -> 64, P = 2, 8 waves per workgroup, 160 KB of LDS: one workgroup per CU, two waves per SIMD), against the same
// grid running a light vector loop.  Wave 0 of every workgroup reads s_memtime (clock64: counts shader cycles) and
// s_memrealtime (wall_clock64: constant 100 MHz) around its loop; cycles / wall time = the clock it ran at.  The roofline's
// 157.3 TFLOP/s assumes 2.4 GHz.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/clock_probe.hip -o /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#include "../pwclonet_pylidarslam_amd/csrc/mlp_core.hpp"
using namespace pwclo;

template <bool MFMA>
__global__ __launch_bounds__(512) void probe(const float *w, float *out, long long *stamps, int tiles) {
  constexpr int NBI = 8, B1 = 8, B2 = 4, P = 2;
  constexpr int W1 = layer_floats(NBI, B1), W2 = layer_floats(B1, B2);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, w, W1 + W2);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 in[NBI][P];
  for (int m = 0; m < NBI; ++m)
    for (int p = 0; p < P; ++p) in[m][p] = f32x4{0.001f * lane, 0.002f * m, 0.003f * p, 1.0f};
  f32x4 accum = {0.f, 0.f, 0.f, 0.f};
  const long long c0 = clock64(), r0 = wall_clock64();
  if (MFMA) {
    for (int t = 0; t < tiles; ++t) {
      f32x4 h1[B1][P], h2[B2][P];
      mlp_layer<NBI, B1, P, true>(h1, in, lds_w, lane);
      mlp_layer<B1, B2, P, true>(h2, h1, lds_w + W1, lane);
      for (int o = 0; o < B2; ++o)
        for (int p = 0; p < P; ++p) accum += h2[o][p];
      in[0][0] = accum * 1e-6f;
    }
  } else {
    for (int t = 0; t < tiles * 768; ++t) {   // one dependent vector add where the other loop issues one MFMA
      accum.x = accum.x * 1.0000001f + 1e-7f;
      __builtin_amdgcn_s_sleep(7);
    }
  }
  const long long c1 = clock64(), r1 = wall_clock64();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
  out[blockIdx.x * 512 + threadIdx.x] = accum.x + accum.y + accum.z + accum.w;
}

template <bool MFMA>
void run(const char *name, int tiles, int reps) {
  constexpr int nw = layer_floats(8, 8) + layer_floats(8, 4);
  std::vector<float> hw(nw, 0.01f);
  float *dw, *dout;
  long long *dst;
  hipMalloc(&dw, nw * 4);
  hipMalloc(&dout, 256 * 512 * 4);
  hipMalloc(&dst, 256 * 2 * 8);
  hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
  auto k = probe<MFMA>;
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < reps; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), nw * 4, 0, dw, dout, dst, tiles);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<long long> st(512);
    hipMemcpy(st.data(), dst, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> mhz;
    for (int i = 0; i < 256; ++i) mhz.push_back(st[2 * i] / (st[2 * i + 1] / 100.0));
    std::sort(mhz.begin(), mhz.end());
    const double mfma = 256.0 * 8 * tiles * (8 * 8 + 8 * 4) * 4.0 * 2;
    const double tf = MFMA ? mfma * 2048.0 / ms / 1e9 : 0.0;
    printf("%-12s launch %d: %8.2f ms   clock64 / wall_clock64: min %7.1f  median %7.1f  max %7.1f MHz", name, it, ms, mhz[0],
           mhz[128], mhz[255]);
    if (MFMA) printf("   %6.1f TFLOP/s = %4.1f %% of 157.3, %4.1f %% of 256 CUs x 4 SIMDs x 64 FLOP/clk at the median clock", tf,
                     tf / 157.3 * 100, tf * 1e12 / (256.0 * 4 * 64 * mhz[128] * 1e6) * 100);
    printf("\n");
  }
}

int main() {
  run<false>("light loop", 400, 2);
  run<true>("fp32 MFMA", 2000, 6);
  run<false>("light loop", 400, 2);
  return 0;
}
