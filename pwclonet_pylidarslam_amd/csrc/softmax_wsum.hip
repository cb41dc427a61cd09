// Softmax over the K neighbours of a query and the weighted sum of the gathered values, as ONE pass each way (gfx950).
//
// The attentive cost volume ends both of its aggregates with (PW/costvolume.py:139-141, 181-183)
//     w = softmax(logits, dim=3);   out = sum(w * values, dim=3)          logits, values: (B, C, S, K)
// In the module path (training, gradient checks) torch runs that as softmax (read + write), a multiply (2 reads + write)
// and a reduction (read + write) forward, and about ten passes backward, over tensors of up to 200 MB.  Here: a row of K
// logits and K values per thread, forward = 2 reads + a (B,C,S) write, backward = 3 reads + 2 writes (the probabilities
// are recomputed from the logits, nothing but the inputs is saved).  HBM-bound; K in [1, 32].
//     p_k = exp(x_k - max x) / sum_j exp(x_j - max x);   out = sum_k p_k v_k
//     dv_k = dout p_k;   dx_k = p_k dout (v_k - out)
// Plain expf / IEEE division (this is the training path: values agree with torch's softmax to fp32 rounding).
#include <math.h>
#include <stdint.h>

#include "common.hpp"

namespace pwclo {

constexpr int SW_THREADS = 256;

template <int K>
__device__ __forceinline__ void sw_load(const float *__restrict__ p, float (&r)[K]) {
  if constexpr (K % 4 == 0) {
#pragma unroll
    for (int i = 0; i < K / 4; ++i) {
      const float4 t = reinterpret_cast<const float4 *>(p)[i];
      r[4 * i] = t.x; r[4 * i + 1] = t.y; r[4 * i + 2] = t.z; r[4 * i + 3] = t.w;
    }
  } else if constexpr (K % 2 == 0) {
#pragma unroll
    for (int i = 0; i < K / 2; ++i) {
      const float2 t = reinterpret_cast<const float2 *>(p)[i];
      r[2 * i] = t.x; r[2 * i + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < K; ++i) r[i] = p[i];
  }
}

template <int K>
__device__ __forceinline__ void sw_store(float *__restrict__ p, const float (&r)[K]) {
  if constexpr (K % 4 == 0) {
#pragma unroll
    for (int i = 0; i < K / 4; ++i) reinterpret_cast<float4 *>(p)[i] = make_float4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
  } else if constexpr (K % 2 == 0) {
#pragma unroll
    for (int i = 0; i < K / 2; ++i) reinterpret_cast<float2 *>(p)[i] = make_float2(r[2 * i], r[2 * i + 1]);
  } else {
#pragma unroll
    for (int i = 0; i < K; ++i) p[i] = r[i];
  }
}

// probabilities of one row; returns the weighted sum
template <int K>
__device__ __forceinline__ float sw_row(const float (&x)[K], const float (&v)[K], float (&p)[K]) {
  float m = x[0];
#pragma unroll
  for (int i = 1; i < K; ++i) m = max_nan(m, x[i]);
  float den = 0.f;
#pragma unroll
  for (int i = 0; i < K; ++i) { p[i] = expf(x[i] - m); den += p[i]; }
  float out = 0.f;
#pragma unroll
  for (int i = 0; i < K; ++i) { p[i] = p[i] / den; out += p[i] * v[i]; }
  return out;
}

template <int K>
__global__ __launch_bounds__(SW_THREADS) void softmax_wsum_fwd_kernel(long long rows, const float *__restrict__ x,
                                                                      const float *__restrict__ v, float *__restrict__ out) {
  const long long r = (long long)blockIdx.x * SW_THREADS + threadIdx.x;
  if (r >= rows) return;
  float xv[K], vv[K], p[K];
  sw_load<K>(x + r * K, xv);
  sw_load<K>(v + r * K, vv);
  out[r] = sw_row<K>(xv, vv, p);
}

template <int K>
__global__ __launch_bounds__(SW_THREADS) void softmax_wsum_bwd_kernel(long long rows, const float *__restrict__ x,
                                                                      const float *__restrict__ v,
                                                                      const float *__restrict__ dout, float *__restrict__ dx,
                                                                      float *__restrict__ dv) {
  const long long r = (long long)blockIdx.x * SW_THREADS + threadIdx.x;
  if (r >= rows) return;
  float xv[K], vv[K], p[K];
  sw_load<K>(x + r * K, xv);
  sw_load<K>(v + r * K, vv);
  const float out = sw_row<K>(xv, vv, p);
  const float g = dout[r];
  float gx[K], gv[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    gv[i] = g * p[i];
    gx[i] = gv[i] * (vv[i] - out);
  }
  sw_store<K>(dx + r * K, gx);
  sw_store<K>(dv + r * K, gv);
}

}  // namespace pwclo

using namespace pwclo;

#define SW_DISPATCH(KERN, ...)                                                                                  \
  switch (k) {                                                                                                  \
    case 1: hipLaunchKernelGGL((KERN<1>), grid, dim3(SW_THREADS), 0, current_stream(), __VA_ARGS__); break;     \
    case 2: hipLaunchKernelGGL((KERN<2>), grid, dim3(SW_THREADS), 0, current_stream(), __VA_ARGS__); break;     \
    case 4: hipLaunchKernelGGL((KERN<4>), grid, dim3(SW_THREADS), 0, current_stream(), __VA_ARGS__); break;     \
    case 6: hipLaunchKernelGGL((KERN<6>), grid, dim3(SW_THREADS), 0, current_stream(), __VA_ARGS__); break;     \
    case 8: hipLaunchKernelGGL((KERN<8>), grid, dim3(SW_THREADS), 0, current_stream(), __VA_ARGS__); break;     \
    case 16: hipLaunchKernelGGL((KERN<16>), grid, dim3(SW_THREADS), 0, current_stream(), __VA_ARGS__); break;   \
    case 32: hipLaunchKernelGGL((KERN<32>), grid, dim3(SW_THREADS), 0, current_stream(), __VA_ARGS__); break;   \
    default: set_error(PWCLO_EINVAL, "softmax_wsum: K=%d is not one of 1, 2, 4, 6, 8, 16, 32", k); return;      \
  }

extern "C" int softmax_wsum_supported_k(int k) { return k == 1 || k == 2 || k == 4 || k == 6 || k == 8 || k == 16 || k == 32; }

extern "C" void softmax_wsum_forward_kernel_wrapper(long long rows, int k, const float *x, const float *v, float *out) {
  if (rows <= 0) return;
  PWCLO_REQUIRE(rows < (1ll << 31) * 256ll, "softmax_wsum: %lld rows exceed the grid", rows);
  PWCLO_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(v)) & 15) == 0,
                "softmax_wsum: logits and values must be 16-byte aligned%s", "");
  const dim3 grid((unsigned)((rows + SW_THREADS - 1) / SW_THREADS));
  SW_DISPATCH(softmax_wsum_fwd_kernel, rows, x, v, out)
  check_launch("softmax_wsum_forward");
}

extern "C" void softmax_wsum_backward_kernel_wrapper(long long rows, int k, const float *x, const float *v, const float *dout,
                                                     float *dx, float *dv) {
  if (rows <= 0) return;
  PWCLO_REQUIRE(rows < (1ll << 31) * 256ll, "softmax_wsum: %lld rows exceed the grid", rows);
  PWCLO_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(dx) |
                  reinterpret_cast<uintptr_t>(dv)) & 15) == 0, "softmax_wsum: tensors must be 16-byte aligned%s", "");
  const dim3 grid((unsigned)((rows + SW_THREADS - 1) / SW_THREADS));
  SW_DISPATCH(softmax_wsum_bwd_kernel, rows, x, v, dout, dx, dv)
  check_launch("softmax_wsum_backward");
}
