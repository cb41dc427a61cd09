// three_nn / three_interpolate (+grad) for gfx950.  Replaces P2/_ext-src/src/interpolate_gpu.cu.
//
// three_nn: the reference scans all m known points serially per thread.  Here a wave owns one
// unknown point; each lane keeps the three best of its strided share (strict <, so the earlier
// index survives ties), then three wave-wide min-reductions on the packed key
// (distance bits << 32 | index) pop the global three best in (distance, index) order -- the same
// order the reference's serial strict-< insertion produces.  Empty slots carry +inf / 0 like the
// reference's (float)1e40 / 0.
#include "common.hpp"

namespace pwclo {

constexpr int NN_WAVES = 4;
constexpr int TI_THREADS = 256;
constexpr int TI_CH_PER_BLOCK = 8;

__global__ __launch_bounds__(NN_WAVES * 64) void three_nn_kernel(int n, int m,
                                                                 const float *__restrict__ unknown,
                                                                 const float *__restrict__ known,
                                                                 float *__restrict__ dist2,
                                                                 int *__restrict__ idx) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * NN_WAVES + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (j >= n) return;  // wave-uniform
  const float *u = unknown + ((size_t)b * n + j) * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  const float *kn = known + (size_t)b * m * 3;

  const float INF = __int_as_float(0x7f800000);
  float b1 = INF, b2 = INF, b3 = INF;
  int i1 = 0, i2 = 0, i3 = 0;
  for (int k = lane; k < m; k += 64) {
    const float dx = ux - kn[k * 3 + 0], dy = uy - kn[k * 3 + 1], dz = uz - kn[k * 3 + 2];
    const float d = dx * dx + dy * dy + dz * dz;
    if (d < b1) {
      b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k;
    } else if (d < b2) {
      b3 = b2; i3 = i2; b2 = d; i2 = k;
    } else if (d < b3) {
      b3 = d; i3 = k;
    }
  }
  float *dj = dist2 + ((size_t)b * n + j) * 3;
  int *ij = idx + ((size_t)b * n + j) * 3;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const unsigned long long key =
        ((unsigned long long)__float_as_uint(b1) << 32) | (unsigned long long)(unsigned)i1;
    const unsigned long long best = wave_allreduce_min_u64(key);
    if (key == best) {  // pop this lane's head (all-empty lanes pop harmlessly)
      b1 = b2; i1 = i2; b2 = b3; i2 = i3; b3 = INF; i3 = 0;
    }
    if (lane == 0) {
      dj[r] = __uint_as_float((unsigned)(best >> 32));
      ij[r] = (int)(unsigned)(best & 0xFFFFFFFFull);
    }
  }
}

// out[b,c,j] = (p[i1]*w1 + p[i2]*w2) + p[i3]*w3, uncontracted (interpolate_gpu.cu:98-99).
__global__ __launch_bounds__(TI_THREADS) void three_interpolate_kernel(
    int c, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * TI_THREADS + threadIdx.x;
  if (j >= n) return;
  const float *w = weight + ((size_t)b * n + j) * 3;
  const int *ix = idx + ((size_t)b * n + j) * 3;
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int i1 = ix[0], i2 = ix[1], i3 = ix[2];
  const int c0 = blockIdx.y * TI_CH_PER_BLOCK, c1 = min(c0 + TI_CH_PER_BLOCK, c);
  for (int l = c0; l < c1; ++l) {
    const float *p = points + ((size_t)b * c + l) * m;
    const float a = p[i1] * w1, bb = p[i2] * w2, cc = p[i3] * w3;
    out[((size_t)b * c + l) * n + j] = (a + bb) + cc;
  }
}

__global__ __launch_bounds__(TI_THREADS) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * TI_THREADS + threadIdx.x;
  if (j >= n) return;
  const float *w = weight + ((size_t)b * n + j) * 3;
  const int *ix = idx + ((size_t)b * n + j) * 3;
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  const int i1 = ix[0], i2 = ix[1], i3 = ix[2];
  const int c0 = blockIdx.y * TI_CH_PER_BLOCK, c1 = min(c0 + TI_CH_PER_BLOCK, c);
  for (int l = c0; l < c1; ++l) {
    const float g = grad_out[((size_t)b * c + l) * n + j];
    float *gp = grad_points + ((size_t)b * c + l) * m;
    atomicAdd(gp + i1, g * w1);
    atomicAdd(gp + i2, g * w2);
    atomicAdd(gp + i3, g * w3);
  }
}

// The same sums through LDS (see group_points_grad_lds_kernel): a workgroup owns (cloud, CT channels, a range
// of the n fine points), accumulates w_j * grad_out into CT x m LDS accumulators and adds its non-zero totals
// to grad_points once.
constexpr int TG_THREADS = 512;
constexpr int TG_LDS_BYTES = 128 * 1024;

__global__ __launch_bounds__(TG_THREADS) void three_interpolate_grad_lds_kernel(
    int c, int n, int m, int ct, int per_split, const float *__restrict__ grad_out,
    const int *__restrict__ idx, const float *__restrict__ weight, float *__restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float tg_acc[];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * ct;
  const int nch = min(ct, c - c0);
  const int total = nch * m;
  for (int i = threadIdx.x; i < total; i += TG_THREADS) tg_acc[i] = 0.f;
  __syncthreads();
  const int j0 = blockIdx.x * per_split, j1 = min(n, j0 + per_split);
  const float *g0 = grad_out + ((size_t)b * c + c0) * n;
  for (int j = j0 + threadIdx.x; j < j1; j += TG_THREADS) {
    const float *w = weight + ((size_t)b * n + j) * 3;
    const int *ix = idx + ((size_t)b * n + j) * 3;
    const float w1 = w[0], w2 = w[1], w3 = w[2];
    const int i1 = ix[0], i2 = ix[1], i3 = ix[2];
    float g[8];      // ct <= 8: all channel loads of a point in flight before the first LDS atomic
#pragma unroll
    for (int l = 0; l < 8; ++l)
      if (l < nch) g[l] = g0[(size_t)l * n + j];
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      if (l < nch) {
        float *a = tg_acc + l * m;
        __hip_atomic_fetch_add(a + i1, g[l] * w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(a + i2, g[l] * w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(a + i3, g[l] * w3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  float *o = grad_points + ((size_t)b * c + c0) * m;
  if (gridDim.x == 1) {   // sole owner of these rows: plain read-modify-write (global fp32 atomics run at ~24 G/s)
    for (int i = threadIdx.x; i < total; i += TG_THREADS) o[i] += tg_acc[i];
    return;
  }
  for (int i = threadIdx.x; i < total; i += TG_THREADS) {
    const float v = tg_acc[i];
    if (v != 0.f) atomicAdd(o + i, v);
  }
}

}  // namespace pwclo

using namespace pwclo;

extern "C" void three_nn_kernel_wrapper(int b, int n, int m, const float *unknown,
                                        const float *known, float *dist2, int *idx) {
  if (b <= 0 || n <= 0) return;
  PWCLO_REQUIRE(b <= 65535, "three_nn: b=%d exceeds the grid limit", b);
  hipLaunchKernelGGL(three_nn_kernel, dim3(ceil_div(n, NN_WAVES), b), dim3(NN_WAVES * 64), 0,
                     current_stream(), n, m, unknown, known, dist2, idx);
  check_launch("three_nn");
}

extern "C" void three_interpolate_kernel_wrapper(int b, int c, int m, int n, const float *points,
                                                 const int *idx, const float *weight, float *out) {
  if (b <= 0 || c <= 0 || n <= 0) return;
  PWCLO_REQUIRE(b <= 65535, "three_interpolate: b=%d exceeds the grid limit", b);
  hipLaunchKernelGGL(three_interpolate_kernel,
                     dim3(ceil_div(n, TI_THREADS), ceil_div(c, TI_CH_PER_BLOCK), b), dim3(TI_THREADS),
                     0, current_stream(), c, m, n, points, idx, weight, out);
  check_launch("three_interpolate");
}

extern "C" void three_interpolate_grad_kernel_wrapper(int b, int c, int n, int m,
                                                      const float *grad_out, const int *idx,
                                                      const float *weight, float *grad_points) {
  if (b <= 0 || c <= 0 || n <= 0) return;
  PWCLO_REQUIRE(b <= 65535, "three_interpolate_grad: b=%d exceeds the grid limit", b);
  static int use_lds = -1;
  if (use_lds < 0) { const char *e = getenv("PWCLO_GRAD_LDS"); use_lds = e ? atoi(e) : 1; }
  if (use_lds && m > 0 && (long long)m * 4 <= TG_LDS_BYTES) {
    int ct = 8;
    while ((long long)ct * m * 4 > TG_LDS_BYTES) ct >>= 1;
    if (ct > c) ct = c;
    while (ct > 1 && b * ceil_div(c, ct) < 256) ct >>= 1;
    const int slices = ceil_div(c, ct);
    int splits = b * slices >= 64 ? 1 : ceil_div(256, b * slices);
    splits = max(1, min(splits, ceil_div(n, 2 * TG_THREADS)));
    const int per_split = ceil_div(n, splits);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void *)three_interpolate_grad_lds_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, TG_LDS_BYTES);
      attr_set = true;
    }
    hipLaunchKernelGGL(three_interpolate_grad_lds_kernel, dim3(ceil_div(n, per_split), slices, b),
                       dim3(TG_THREADS), (size_t)ct * m * 4, current_stream(), c, n, m, ct, per_split, grad_out, idx,
                       weight, grad_points);
    check_launch("three_interpolate_grad");
    return;
  }
  hipLaunchKernelGGL(three_interpolate_grad_kernel,
                     dim3(ceil_div(n, TI_THREADS), ceil_div(c, TI_CH_PER_BLOCK), b), dim3(TI_THREADS),
                     0, current_stream(), c, n, m, grad_out, idx, weight, grad_points);
  check_launch("three_interpolate_grad");
}
