// KITTI odometry evaluation of predicted poses on the device (gfx950).
//
// The reference turns every batch's pose rows into 4x4 matrices on the HOST, one sample at a time
// (train.py:866-893: D2H copy, quat2mat, np.linalg.inv per sample), chains them with a Python loop
// (kitti360_utils.py:406-431), writes text files and re-reads them to compute the KITTI segment errors with
// three nested Python loops (evaluation.py:236-271 / slam/eval/eval_odometry.py:316-361).  Here the pose rows
// stay in HBM as the network wrote them; four small fp64 kernels do the rest for ALL sequences at once:
//   rows -> transforms        one thread per frame                      (train.py:762-795 quat2mat, :873-878)
//   transforms -> trajectory  one wave per sequence, chunked scan       (kitti360_utils.py:422-426)
//   trajectory -> distances   one wave per sequence, chunked scan       (evaluation.py:198-215)
//   segment errors            one thread per (first frame, length)      (evaluation.py:236-271)
// The work is a few thousand 4x4 fp64 products: latency-bound, nowhere near any roofline; the point is that
// nothing leaves the device and no host loop runs per frame.  SE(3) composition is associative, so the scans
// differ from the reference's sequential products only by fp64 rounding (tests: 1e-9).
#include "common.hpp"

namespace pwclo {

struct Se3 {           // rows 0..2 of a homogeneous transform, row-major: r[4*i + j], j = 3 is the translation
  double m[12];
};

__device__ __forceinline__ Se3 se3_identity() {
  Se3 a;
#pragma unroll
  for (int i = 0; i < 12; ++i) a.m[i] = (i % 5 == 0) ? 1.0 : 0.0;
  return a;
}
// c = a . b
__device__ __forceinline__ Se3 se3_mul(const Se3 &a, const Se3 &b) {
  Se3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double s = a.m[4 * i + 0] * b.m[j] + a.m[4 * i + 1] * b.m[4 + j] + a.m[4 * i + 2] * b.m[8 + j];
      if (j == 3) s += a.m[4 * i + 3];
      c.m[4 * i + j] = s;
    }
  }
  return c;
}
// General inverse of [A t; 0 1] (A need not be orthonormal: quat2mat of a non-unit quaternion is still a
// rotation, but ground-truth files may hold anything): A^-1 by cofactors, -A^-1 t.
__device__ __forceinline__ Se3 se3_inv(const Se3 &a) {
  const double a00 = a.m[0], a01 = a.m[1], a02 = a.m[2], a10 = a.m[4], a11 = a.m[5], a12 = a.m[6], a20 = a.m[8],
               a21 = a.m[9], a22 = a.m[10];
  const double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const double det = a00 * c00 + a01 * c01 + a02 * c02;
  const double id = 1.0 / det;
  Se3 r;
  r.m[0] = c00 * id; r.m[1] = (a02 * a21 - a01 * a22) * id; r.m[2] = (a01 * a12 - a02 * a11) * id;
  r.m[4] = c01 * id; r.m[5] = (a00 * a22 - a02 * a20) * id; r.m[6] = (a02 * a10 - a00 * a12) * id;
  r.m[8] = c02 * id; r.m[9] = (a01 * a20 - a00 * a21) * id; r.m[10] = (a00 * a11 - a01 * a10) * id;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    r.m[4 * i + 3] = -(r.m[4 * i] * a.m[3] + r.m[4 * i + 1] * a.m[7] + r.m[4 * i + 2] * a.m[11]);
  return r;
}
__device__ __forceinline__ Se3 se3_load(const double *p) {   // from a 4x4 row-major matrix
  Se3 a;
#pragma unroll
  for (int i = 0; i < 12; ++i) a.m[i] = p[i];
  return a;
}
__device__ __forceinline__ void se3_store(double *p, const Se3 &a) {
#pragma unroll
  for (int i = 0; i < 12; ++i) p[i] = a.m[i];
  p[12] = 0.0; p[13] = 0.0; p[14] = 0.0; p[15] = 1.0;
}
__device__ __forceinline__ Se3 se3_shfl_up(const Se3 &a, int delta) {
  Se3 r;
#pragma unroll
  for (int i = 0; i < 12; ++i) r.m[i] = __shfl_up(a.m[i], delta, 64);
  return r;
}
__device__ __forceinline__ double f64_shfl_up(double v, int delta) { return __shfl_up(v, delta, 64); }

// train.py:762-795 (quat2mat: the nibabel form, valid for non-unit quaternions, identity below 1e-8) and
// :873-878: T = [[R t], [0 0 0 1]] from a pose row [tx ty tz qw qx qy qz].  fp32 inputs, fp64 arithmetic.
__global__ __launch_bounds__(256) void odom_rows_to_transforms_kernel(int n, int row_stride,
                                                                      const float *__restrict__ rows,
                                                                      double *__restrict__ T, int invert) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *r = rows + (size_t)i * row_stride;
  const double w = r[3], x = r[4], y = r[5], z = r[6];
  const double nq = w * w + x * x + y * y + z * z;
  Se3 a = se3_identity();
  if (!(nq < 1e-8)) {
    const double s = 2.0 / nq;
    const double X = x * s, Y = y * s, Z = z * s;
    const double wX = w * X, wY = w * Y, wZ = w * Z, xX = x * X, xY = x * Y, xZ = x * Z, yY = y * Y, yZ = y * Z,
                 zZ = z * Z;
    a.m[0] = 1.0 - (yY + zZ); a.m[1] = xY - wZ; a.m[2] = xZ + wY;
    a.m[4] = xY + wZ; a.m[5] = 1.0 - (xX + zZ); a.m[6] = yZ - wX;
    a.m[8] = xZ - wY; a.m[9] = yZ + wX; a.m[10] = 1.0 - (xX + yY);
  }
  a.m[3] = r[0]; a.m[7] = r[1]; a.m[11] = r[2];
  if (invert) a = se3_inv(a);
  se3_store(T + (size_t)i * 16, a);
}

// abs[f] = T[0] . T[1] ... T[f] within each sequence (kitti360_utils.py:422-426 with rel = T^-1:
// abs[f] = inv(rel[f] @ inv(abs[f-1])) = abs[f-1] . T[f], abs[-1] = I).  One wave per sequence: every lane
// multiplies a contiguous chunk, the 64 chunk products are scanned with 6 shuffle steps, then each lane
// replays its chunk from its exclusive prefix.
__global__ __launch_bounds__(64) void odom_accumulate_kernel(const int *__restrict__ seq_start,
                                                             const double *__restrict__ T,
                                                             double *__restrict__ abs_out) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const int lo = seq_start[s], n = seq_start[s + 1] - lo;
  if (n <= 0) return;
  const int chunk = (n + 63) / 64;
  const int c0 = min(lane * chunk, n), c1 = min(c0 + chunk, n);
  Se3 prod = se3_identity();
  for (int i = c0; i < c1; ++i) prod = se3_mul(prod, se3_load(T + (size_t)(lo + i) * 16));
  Se3 incl = prod;                                   // inclusive scan over lanes (left operand = earlier lanes)
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const Se3 up = se3_shfl_up(incl, d);
    if (lane >= d) incl = se3_mul(up, incl);
  }
  Se3 run = se3_shfl_up(incl, 1);                    // exclusive prefix
  if (lane == 0) run = se3_identity();
  for (int i = c0; i < c1; ++i) {
    run = se3_mul(run, se3_load(T + (size_t)(lo + i) * 16));
    se3_store(abs_out + (size_t)(lo + i) * 16, run);
  }
}

// evaluation.py:198-215 / eval_odometry.py:268-276: dist[0] = 0, dist[i] = dist[i-1] + |p[i] - p[i-1]|.
__global__ __launch_bounds__(64) void odom_cumdist_kernel(const int *__restrict__ seq_start,
                                                          const double *__restrict__ poses,
                                                          double *__restrict__ dist) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const int lo = seq_start[s], n = seq_start[s + 1] - lo;
  if (n <= 0) return;
  const int chunk = (n + 63) / 64;
  const int c0 = min(lane * chunk, n), c1 = min(c0 + chunk, n);
  auto step = [&](int i) -> double {                 // length of the move INTO frame i (0 for the first frame)
    if (i == 0) return 0.0;
    const double *a = poses + (size_t)(lo + i - 1) * 16, *b = poses + (size_t)(lo + i) * 16;
    const double dx = a[3] - b[3], dy = a[7] - b[7], dz = a[11] - b[11];
    return sqrt(dx * dx + dy * dy + dz * dz);
  };
  double sum = 0.0;
  for (int i = c0; i < c1; ++i) sum += step(i);
  double incl = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const double up = f64_shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  double run = f64_shfl_up(incl, 1);
  if (lane == 0) run = 0.0;
  for (int i = c0; i < c1; ++i) {
    run += step(i);
    dist[lo + i] = run;
  }
}

// evaluation.py:236-271 (= eval_odometry.py:316-361): for every first frame f = 0, step, 2*step, ... and every
// segment length L: last = first i >= f with dist[i] > dist[f] + L (dist is non-decreasing, so the reference's
// linear search is this upper-bound search); pose_error = inv(inv(R[f]) R[last]) . (inv(G[f]) G[last]);
// row = [f, r_err / L, t_err / L, L, speed], speed = L / (0.1 * (last - f + 1)).
__global__ __launch_bounds__(256) void odom_sequence_errors_kernel(int nseq, const int *__restrict__ seq_start,
                                                                   const int *__restrict__ slot_start,
                                                                   const double *__restrict__ gt,
                                                                   const double *__restrict__ res,
                                                                   const double *__restrict__ dist, int step,
                                                                   int nlen, const double *__restrict__ lengths,
                                                                   double *__restrict__ err,
                                                                   int *__restrict__ valid) {
  const int slot = blockIdx.x * 256 + threadIdx.x;
  if (slot >= slot_start[nseq]) return;
  int s = 0;                                          // few sequences: linear search for the owner
  while (s + 1 < nseq && slot >= slot_start[s + 1]) ++s;
  const int local = slot - slot_start[s];
  const int lo = seq_start[s], n = seq_start[s + 1] - lo;
  const int first = (local / nlen) * step, li = local % nlen;
  const double L = lengths[li];
  const double *d = dist + lo;
  const double target = d[first] + L;
  int a = first, b = n;                               // first index in [first, n) with d[i] > target
  while (a < b) {
    const int mid = (a + b) >> 1;
    if (d[mid] > target) b = mid; else a = mid + 1;
  }
  double *row = err + (size_t)slot * 5;
  if (a >= n) {
    valid[slot] = 0;
    row[0] = first; row[1] = 0.0; row[2] = 0.0; row[3] = L; row[4] = 0.0;
    return;
  }
  const int last = a;
  const Se3 dg = se3_mul(se3_inv(se3_load(gt + (size_t)(lo + first) * 16)), se3_load(gt + (size_t)(lo + last) * 16));
  const Se3 dr = se3_mul(se3_inv(se3_load(res + (size_t)(lo + first) * 16)), se3_load(res + (size_t)(lo + last) * 16));
  const Se3 e = se3_mul(se3_inv(dr), dg);
  const double tr = 0.5 * (e.m[0] + e.m[5] + e.m[10] - 1.0);
  const double r_err = acos(fmax(fmin(tr, 1.0), -1.0));
  const double t_err = sqrt(e.m[3] * e.m[3] + e.m[7] * e.m[7] + e.m[11] * e.m[11]);
  const double frames = (double)(last - first) + 1.0;
  valid[slot] = 1;
  row[0] = first; row[1] = r_err / L; row[2] = t_err / L; row[3] = L; row[4] = L / (0.1 * frames);
}

}  // namespace pwclo

using namespace pwclo;

extern "C" void odom_rows_to_transforms_kernel_wrapper(int n, int row_stride, const float *rows, double *T,
                                                       int invert) {
  if (n <= 0) return;
  PWCLO_REQUIRE(row_stride >= 7, "odom_rows_to_transforms: row_stride=%d must be >= 7", row_stride);
  hipLaunchKernelGGL(odom_rows_to_transforms_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, current_stream(), n,
                     row_stride, rows, T, invert);
  check_launch("odom_rows_to_transforms");
}

extern "C" void odom_accumulate_kernel_wrapper(int nseq, const int *seq_start, const double *T, double *abs_out) {
  if (nseq <= 0) return;
  hipLaunchKernelGGL(odom_accumulate_kernel, dim3(nseq), dim3(64), 0, current_stream(), seq_start, T, abs_out);
  check_launch("odom_accumulate");
}

extern "C" void odom_cumulative_distance_kernel_wrapper(int nseq, const int *seq_start, const double *poses,
                                                        double *dist) {
  if (nseq <= 0) return;
  hipLaunchKernelGGL(odom_cumdist_kernel, dim3(nseq), dim3(64), 0, current_stream(), seq_start, poses, dist);
  check_launch("odom_cumulative_distance");
}

extern "C" void odom_sequence_errors_kernel_wrapper(int nseq, int total_slots, const int *seq_start,
                                                    const int *slot_start, const double *poses_gt,
                                                    const double *poses_result, const double *dist, int step,
                                                    int nlen, const double *lengths, double *err, int *valid) {
  if (nseq <= 0 || total_slots <= 0) return;
  PWCLO_REQUIRE(step >= 1 && nlen >= 1, "odom_sequence_errors: step=%d nlen=%d must be >= 1", step, nlen);
  hipLaunchKernelGGL(odom_sequence_errors_kernel, dim3(ceil_div(total_slots, 256)), dim3(256), 0, current_stream(),
                     nseq, seq_start, slot_start, poses_gt, poses_result, dist, step, nlen, lengths, err, valid);
  check_launch("odom_sequence_errors");
}
