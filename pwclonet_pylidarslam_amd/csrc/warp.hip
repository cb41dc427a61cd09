// Quaternion pose warp for gfx950.  Replaces PW/PWCLO_utils.py:31-63 (warp = two Hamilton
// products built from ~40 tiny torch ops, plus two host-side tensor constructions and H2D copies
// per call).  One thread per point, streaming (B,3,N) -> (B,3,N); the products follow the
// reference's term order (PWCLO_utils.py:83-95,117-129) without contraction so the warped
// coordinates -- which feed the neighbour searches of the refinement levels -- agree with the
// reference to the last bit wherever torch's own evaluation order is defined.
#include <stdint.h>

#include "common.hpp"

PWCLO_TRACE_TU(warp)

namespace pwclo {

struct quat { float w, x, y, z; };

// Hamilton product a (x) b with the reference's left-to-right evaluation of each component.
__device__ __forceinline__ quat hamilton(const quat a, const quat b) {
  quat r;
  r.w = ((a.w * b.w - a.x * b.x) - a.y * b.y) - a.z * b.z;
  r.x = ((a.w * b.x + a.x * b.w) + a.y * b.z) - a.z * b.y;
  r.y = ((a.w * b.y - a.x * b.z) + a.y * b.w) + a.z * b.x;
  r.z = ((a.w * b.z + a.x * b.y) - a.y * b.x) + a.z * b.w;
  return r;
}

template <bool POINT_MAJOR>
__global__ __launch_bounds__(256) void quat_warp_kernel(int n, const float *__restrict__ xyz,
                                                        const float *__restrict__ q,
                                                        const float *__restrict__ t,
                                                        float *__restrict__ out) {
  TraceScope trace_scope_(TK_WARP);
  const int b = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const quat qq = {q[b * 4 + 0], q[b * 4 + 1], q[b * 4 + 2], q[b * 4 + 3]};
  // inv_q (PWCLO_utils.py:31-39): conj(q) / (sum(q*q) + 1e-10)
  const float q2 = (((qq.w * qq.w + qq.x * qq.x) + qq.y * qq.y) + qq.z * qq.z) + 1e-10f;
  const quat qi = {qq.w / q2, (qq.x * -1.0f) / q2, (qq.y * -1.0f) / q2, (qq.z * -1.0f) / q2};
  const float *src = xyz + (size_t)b * 3 * n;
  const int sx = POINT_MAJOR ? 3 * j : j, st = POINT_MAJOR ? 1 : n;  // (b,n,3) or (b,3,n)
  const quat p = {0.0f, src[sx], src[sx + st], src[sx + 2 * st]};
  const quat r = hamilton(hamilton(qq, p), qi);
  float *dst = out + (size_t)b * 3 * n;
  dst[sx] = r.x + t[b * 3 + 0];
  dst[sx + st] = r.y + t[b * 3 + 1];
  dst[sx + 2 * st] = r.z + t[b * 3 + 2];
}

// Input adapter of the fused pipeline: two (B,3,N) channel-major clouds (what PWCLONet.forward
// receives, pwclo_net.py:125-126) -> one point-major (2B,N,3) batch (frame 1 first).  Replaces
// torch.cat + permute + contiguous (three passes) by one.
__global__ __launch_bounds__(256) void ingest_pairs_kernel(int bsz, int n, const float *__restrict__ f1,
                                                           const float *__restrict__ f2,
                                                           float *__restrict__ out) {
  TraceScope trace_scope_(TK_INGEST, 15u);
  const int b = blockIdx.y;               // 0 .. 2*bsz-1
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float *src = (b < bsz ? f1 + (size_t)b * 3 * n : f2 + (size_t)(b - bsz) * 3 * n);
  float *dst = out + ((size_t)b * n + j) * 3;
  dst[0] = src[j];
  dst[1] = src[n + j];
  dst[2] = src[2 * n + j];
}

// Input adapter one level up (prediction_modules.py:144-160): two point-major frames (B, n_total, c),
// c >= 3 floats per point, of which the first n points and the first 3 channels are used
// (`pcd[:, :num_points, :3]`) -> the same (2B,n,3) batch.  Replaces slice + permute + contiguous in
// the reference's adapter and the permute back in the fused forward.
__global__ __launch_bounds__(256) void ingest_frames_kernel(int bsz, int n, int n_total, int c,
                                                            const float *__restrict__ f1,
                                                            const float *__restrict__ f2,
                                                            float *__restrict__ out) {
  const int b = blockIdx.y;               // 0 .. 2*bsz-1
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const float *src = (b < bsz ? f1 + (size_t)b * n_total * c : f2 + (size_t)(b - bsz) * n_total * c) + (size_t)j * c;
  float *dst = out + ((size_t)b * n + j) * 3;
  dst[0] = src[0];
  dst[1] = src[1];
  dst[2] = src[2];
}

// Raw KITTI velodyne frame -> camera-frame cloud + keep mask (kitti_odometry_dataset.py:375-397 and
// filter_pcd :149-160): p' = Tr[:3,:4] . (x, y, z, 1) in fp64 like the reference's numpy matmul on the
// float64-promoted points, keep = not ground (y' <= 1.1) and |x'| < 30 and |z'| < 30 (strict, as
// the reference's `<` / `>`), coordinates stored as fp32.  One thread per point; HBM-trivial
// (16 bytes in, 16 bytes out per point).
__global__ __launch_bounds__(256) void kitti_transform_filter_kernel(int n, const double *__restrict__ tr,
                                                                     const float *__restrict__ points,
                                                                     float *__restrict__ xyz,
                                                                     int *__restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = *reinterpret_cast<const float4 *>(points + (size_t)i * 4);   // (x, y, z, intensity)
  const double x = p.x, y = p.y, z = p.z;
  double o[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
    o[r] = ((tr[r * 4 + 0] * x + tr[r * 4 + 1] * y) + tr[r * 4 + 2] * z) + tr[r * 4 + 3];
  const bool ground = o[1] > 1.1;
  const bool near = (o[0] < 30.0 && o[0] > -30.0) && (o[2] < 30.0 && o[2] > -30.0);
  xyz[(size_t)i * 3 + 0] = (float)o[0];
  xyz[(size_t)i * 3 + 1] = (float)o[1];
  xyz[(size_t)i * 3 + 2] = (float)o[2];
  keep[i] = (!ground && near) ? 1 : 0;
}

// KITTI-360 front end (slam/dataset/kitti_360_dataset_2.py:113-123): raw velodyne rows stay in the sensor
// frame; keep = not ground (z >= ground_z) and |x| < near and |y| < near, compared in fp32 as NumPy
// compares a float32 column with a Python scalar.  xyz = the first three columns.
__global__ __launch_bounds__(256) void kitti360_filter_kernel(int n, float ground_z, float near,
                                                              const float *__restrict__ points,
                                                              float *__restrict__ xyz, int *__restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 p = *reinterpret_cast<const float4 *>(points + (size_t)i * 4);
  const bool ground = p.z < ground_z;
  const bool close = (p.x < near && p.x > -near) && (p.y < near && p.y > -near);
  xyz[(size_t)i * 3 + 0] = p.x;
  xyz[(size_t)i * 3 + 1] = p.y;
  xyz[(size_t)i * 3 + 2] = p.z;
  keep[i] = (!ground && close) ? 1 : 0;
}

// Stable per-frame compaction of the kept rows: pos = inclusive scan of keep along the frame, row i of
// frame f goes to out[f, pos-1]; rows >= count stay as the caller zero-filled them (the sampler never
// selects zero rows).  Rows beyond `cap` are dropped (the caller sizes cap >= max count).
__global__ __launch_bounds__(256) void compact_frames_kernel(int n, int cap, const int *__restrict__ keep,
                                                             const int *__restrict__ pos,
                                                             const float *__restrict__ xyz,
                                                             float *__restrict__ out, int *__restrict__ counts) {
  const int f = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const size_t row = (size_t)f * n + i;
  const int p = pos[row];
  if (keep[row] != 0 && p <= cap) {
    float *d = out + ((size_t)f * cap + (p - 1)) * 3;
    d[0] = xyz[row * 3 + 0]; d[1] = xyz[row * 3 + 1]; d[2] = xyz[row * 3 + 2];
  }
  if (i == n - 1) counts[f] = p < cap ? p : cap;
}

// The same compaction WITHOUT a precomputed scan (round 3: the scan used to be torch.cumsum, a library kernel on the product
// path of BASELINE configs[4]): one 1024-thread workgroup per frame; wave w owns a contiguous range of rows and walks it
// 64 rows at a time (coalesced), a ballot + popcount gives every kept row its slot inside the step, the running sum of
// popcounts its slot inside the wave's range; the 16 waves' totals are scanned through LDS between a counting pass and
// the writing pass.  Stable (frame order), same output as the scan + scatter pair.
__global__ __launch_bounds__(1024) void compact_frames_scan_kernel(int n, int cap, const int *__restrict__ keep,
                                                                   const float *__restrict__ xyz,
                                                                   float *__restrict__ out, int *__restrict__ counts) {
  __shared__ int wave_total[16];
  const int f = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per_wave = ((n + 15) / 16 + 63) / 64 * 64;          // multiple of 64: steps never straddle two waves' ranges
  const int i0 = wave * per_wave, i1 = min(n, i0 + per_wave);
  const int *kf = keep + (size_t)f * n;
  int total = 0;
  for (int i = i0 + lane; i - lane < i1; i += 64)
    total += __popcll(__ballot(i < i1 && kf[i] != 0));
  if (lane == 0) wave_total[wave] = total;
  __syncthreads();
  int base = 0, all = 0;
  for (int w = 0; w < 16; ++w) {
    const int t = wave_total[w];
    if (w < wave) base += t;
    all += t;
  }
  const float *xf = xyz + (size_t)f * n * 3;
  float *of = out + (size_t)f * cap * 3;
  for (int i = i0 + lane; i - lane < i1; i += 64) {
    const bool k = i < i1 && kf[i] != 0;
    const unsigned long long m = __ballot(k);
    const int p = base + mbcnt64(m);
    if (k && p < cap) {
      of[(size_t)p * 3 + 0] = xf[(size_t)i * 3 + 0];
      of[(size_t)p * 3 + 1] = xf[(size_t)i * 3 + 1];
      of[(size_t)p * 3 + 2] = xf[(size_t)i * 3 + 2];
    }
    base += __popcll(m);
  }
  if (threadIdx.x == 0) counts[f] = all < cap ? all : cap;
}

}  // namespace pwclo

using namespace pwclo;

extern "C" void kitti_transform_filter_kernel_wrapper(int n, const double *tr, const float *points, float *xyz,
                                                      int *keep) {
  if (n <= 0) return;
  PWCLO_REQUIRE((reinterpret_cast<uintptr_t>(points) & 15) == 0, "kitti_transform_filter: points must be 16-byte aligned%s", "");
  hipLaunchKernelGGL(kitti_transform_filter_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, current_stream(), n, tr,
                     points, xyz, keep);
  check_launch("kitti_transform_filter");
}

extern "C" void kitti360_filter_kernel_wrapper(int n, float ground_z, float near, const float *points, float *xyz,
                                               int *keep) {
  if (n <= 0) return;
  PWCLO_REQUIRE((reinterpret_cast<uintptr_t>(points) & 15) == 0, "kitti360_filter: points must be 16-byte aligned%s", "");
  hipLaunchKernelGGL(kitti360_filter_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, current_stream(), n, ground_z,
                     near, points, xyz, keep);
  check_launch("kitti360_filter");
}

extern "C" void compact_frames_kernel_wrapper(int b, int n, int cap, const int *keep, const int *pos,
                                              const float *xyz, float *out, int *counts) {
  if (b <= 0 || n <= 0) return;
  PWCLO_REQUIRE(b <= 65535 && cap > 0, "compact_frames: b=%d cap=%d out of range", b, cap);
  hipLaunchKernelGGL(compact_frames_kernel, dim3(ceil_div(n, 256), b), dim3(256), 0, current_stream(), n, cap,
                     keep, pos, xyz, out, counts);
  check_launch("compact_frames");
}

extern "C" void compact_frames_scan_kernel_wrapper(int b, int n, int cap, const int *keep, const float *xyz, float *out,
                                                   int *counts) {
  if (b <= 0 || n <= 0) return;
  PWCLO_REQUIRE(cap > 0 && (long long)n * 3 < (1ll << 31), "compact_frames_scan: n=%d cap=%d out of range", n, cap);
  hipLaunchKernelGGL(compact_frames_scan_kernel, dim3(b), dim3(1024), 0, current_stream(), n, cap, keep, xyz, out, counts);
  check_launch("compact_frames_scan");
}

extern "C" void ingest_frames_kernel_wrapper(int b, int n, int n_total, int c, const float *frame1,
                                             const float *frame2, float *out) {
  if (b <= 0 || n <= 0) return;
  PWCLO_REQUIRE(2 * b <= 65535, "ingest_frames: b=%d exceeds the grid limit", b);
  PWCLO_REQUIRE(c >= 3 && n_total >= n, "ingest_frames: need c >= 3 and n_total >= n (c=%d n_total=%d n=%d)", c,
                n_total, n);
  hipLaunchKernelGGL(ingest_frames_kernel, dim3(ceil_div(n, 256), 2 * b), dim3(256), 0, current_stream(), b, n,
                     n_total, c, frame1, frame2, out);
  check_launch("ingest_frames");
}

extern "C" void ingest_pairs_kernel_wrapper(int b, int n, const float *xyz_f1, const float *xyz_f2,
                                            float *out) {
  if (b <= 0 || n <= 0) return;
  PWCLO_REQUIRE(2 * b <= 65535, "ingest_pairs: b=%d exceeds the grid limit", b);
  hipLaunchKernelGGL(ingest_pairs_kernel, dim3(ceil_div(n, 256), 2 * b), dim3(256), 0, current_stream(), b, n,
                     xyz_f1, xyz_f2, out);
  check_launch("ingest_pairs");
}

// Hamilton product of two batches of quaternion rows, c[b][:, n] = a[b][:, n or 0] (x) b[b][:, n or 0], rows (B, 4, N)
// (PW/PWCLO_utils.py:83-95 mul_q_point and 117-129 mul_point_q: both are this product with the left operand's factors
// first).  Every product is rounded before the sums, left to right as the reference's torch expression, so the result
// is the expression's bit for bit.  conj_a / conj_b: use the conjugate (a0, -a1, -a2, -a3) of that operand -- the
// product's gradients are products with conjugates (dA = dC (x) conj(B), dB = conj(A) (x) dC), so the same kernel
// serves the backward.
namespace pwclo {
__global__ void hamilton_kernel(int n, int na, int nb, int conj_a, int conj_b, const float *__restrict__ a,
                                const float *__restrict__ b, float *__restrict__ c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, bi = blockIdx.y;
  if (i >= n) return;
  const float *pa = a + (size_t)bi * 4 * na + (na == 1 ? 0 : i);
  const float *pb = b + (size_t)bi * 4 * nb + (nb == 1 ? 0 : i);
  const float sa = conj_a ? -1.f : 1.f, sb = conj_b ? -1.f : 1.f;
  const float a0 = pa[0], a1 = sa * pa[na], a2 = sa * pa[2 * na], a3 = sa * pa[3 * na];
  const float b0 = pb[0], b1 = sb * pb[nb], b2 = sb * pb[2 * nb], b3 = sb * pb[3 * nb];
  float *pc = c + (size_t)bi * 4 * n + i;
  auto m = [](float u, float v) { return __fmul_rn(u, v); };
  pc[0] = __fsub_rn(__fsub_rn(__fsub_rn(m(a0, b0), m(a1, b1)), m(a2, b2)), m(a3, b3));
  pc[n] = __fsub_rn(__fadd_rn(__fadd_rn(m(a0, b1), m(a1, b0)), m(a2, b3)), m(a3, b2));
  pc[2 * n] = __fadd_rn(__fadd_rn(__fsub_rn(m(a0, b2), m(a1, b3)), m(a2, b0)), m(a3, b1));
  pc[3 * n] = __fadd_rn(__fsub_rn(__fadd_rn(m(a0, b3), m(a1, b2)), m(a2, b1)), m(a3, b0));
}
}  // namespace pwclo

extern "C" void hamilton_product_kernel_wrapper(int b, int n, int na, int nb, int conj_a, int conj_b, const float *a,
                                                const float *q, float *out) {
  if (b <= 0 || n <= 0) return;
  PWCLO_REQUIRE(b <= 65535, "hamilton_product: b=%d exceeds the grid limit", b);
  PWCLO_REQUIRE((na == 1 || na == n) && (nb == 1 || nb == n), "hamilton_product: operand lengths %d, %d must be 1 or n=%d",
                na, nb, n);
  hipLaunchKernelGGL(pwclo::hamilton_kernel, dim3(ceil_div(n, 256), b), dim3(256), 0, current_stream(), n, na, nb, conj_a,
                     conj_b, a, q, out);
  check_launch("hamilton_product");
}

extern "C" void quat_warp_kernel_wrapper(int b, int n, const float *xyz, const float *q,
                                         const float *t, float *out) {
  if (b <= 0 || n <= 0) return;
  PWCLO_REQUIRE(b <= 65535, "quat_warp: b=%d exceeds the grid limit", b);
  hipLaunchKernelGGL(quat_warp_kernel<false>, dim3(ceil_div(n, 256), b), dim3(256), 0, current_stream(),
                     n, xyz, q, t, out);
  check_launch("quat_warp");
}

// Same transform on point-major (b,n,3) clouds (layout of the fused eval-mode pipeline).
extern "C" void quat_warp_pm_kernel_wrapper(int b, int n, const float *xyz, const float *q,
                                            const float *t, float *out) {
  if (b <= 0 || n <= 0) return;
  PWCLO_REQUIRE(b <= 65535, "quat_warp_pm: b=%d exceeds the grid limit", b);
  hipLaunchKernelGGL(quat_warp_kernel<true>, dim3(ceil_div(n, 256), b), dim3(256), 0, current_stream(),
                     n, xyz, q, t, out);
  check_launch("quat_warp_pm");
}
