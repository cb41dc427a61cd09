// Probe: pure mlp_layer throughput (no gathers / epilogue) for the two wave/tile geometries.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../pwclonet_pylidarslam_amd/csrc/mlp_core.hpp"
using namespace pwclo;

template <int NBI, int B1, int B2, int P, int W>
__global__ __launch_bounds__(W * 64) void probe(const float *w, float *out, int tiles) {
  constexpr int W1 = layer_floats(NBI, B1), W2 = layer_floats(B1, B2);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, w, W1 + W2);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 in[NBI][P];
  for (int m = 0; m < NBI; ++m)
    for (int p = 0; p < P; ++p) in[m][p] = f32x4{0.001f * lane, 0.002f * m, 0.003f * p, 1.0f};
  f32x4 accum = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < tiles; ++t) {
    f32x4 h1[B1][P], h2[B2][P];
    mlp_layer<NBI, B1, P, true>(h1, in, lds_w, lane);
    mlp_layer<B1, B2, P, true>(h2, h1, lds_w + W1, lane);
    for (int o = 0; o < B2; ++o)
      for (int p = 0; p < P; ++p) accum += h2[o][p];
    in[0][0] = accum * 1e-6f;   // carry a dependence so nothing is hoisted out of the loop
  }
  out[(blockIdx.x * W * 64 + threadIdx.x)] = accum.x + accum.y + accum.z + accum.w;
}

template <int NBI, int B1, int B2, int P, int W>
void run(const char *name) {
  constexpr int nw = layer_floats(NBI, B1) + layer_floats(B1, B2);
  std::vector<float> hw(nw, 0.01f);
  float *dw, *dout;
  hipMalloc(&dw, nw * 4);
  hipMalloc(&dout, 256 * W * 64 * 4);
  hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
  auto k = probe<NBI, B1, B2, P, W>;
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int tiles = 64 / P * 2;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 2; ++it) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(W * 64), nw * 4, 0, dw, dout, tiles);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double mfma = 256.0 * W * tiles * (NBI * B1 + B1 * B2) * 4.0 * P;
  const double flops = mfma * 2.0 * 1024;
  printf("%-28s W=%2d P=%d: %8.1f us  %6.1f TFLOP/s  (%4.1f %% of 157.3)\n", name, W, P, ms * 1e3,
         flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
  run<1, 8, 4, 1, 16>("upconv_h 16->128->64");
  run<1, 8, 4, 2, 8>("upconv_h 16->128->64");
  run<1, 8, 4, 3, 8>("upconv_h 16->128->64");
  run<5, 8, 4, 2, 8>("upconv 80->128->64");
  run<5, 8, 4, 1, 16>("upconv 80->128->64");
  run<5, 8, 4, 4, 4>("upconv 80->128->64");
  run<8, 8, 4, 1, 16>("128->128->64 (cv_a2)");
  run<8, 8, 4, 2, 8>("128->128->64 (cv_a2)");
  run<8, 8, 4, 3, 8>("128->128->64 (cv_a2)");
  run<8, 8, 4, 4, 4>("128->128->64");
  run<8, 8, 4, 4, 8>("128->128->64");
  return 0;
}
