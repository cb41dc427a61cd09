// group_points (+grad) for gfx950.  Replaces P2/_ext-src/src/group_points_gpu.cu.
//
// HBM-bound copy: out[b,c,j,k] = points[b,c,idx[b,j,k]].  The reference launches ONE block per
// cloud and lets each thread write `nsample` consecutive floats (strided across the wave).  Here
// the flattened (j,k) axis is spread over the grid: one thread owns 4 consecutive output
// elements, loads their 4 indices with one 16-byte load (read once, reused for every channel of
// the block's channel slice), gathers from the channel row (C*N*4 bytes per cloud: L2 resident)
// and stores 16 bytes, so every wave-store is one contiguous 1 KiB line run.
// Algorithmic bytes per call: 4*(S*K + C*N + C*S*K).
#include <stdint.h>
#include <stdlib.h>

#include "common.hpp"

namespace pwclo {

constexpr int GP_THREADS = 256;
constexpr int GP_CH_PER_BLOCK = 8;  // channels handled by one block (grid.y = ceil(C / 8))

// P = npoints*nsample.  VEC4 requires P % 4 == 0 (then every row start stays 16-byte aligned).
template <bool VEC4>
__global__ __launch_bounds__(GP_THREADS) void group_points_kernel(int c, int n, int P,
                                                                  const float *__restrict__ points,
                                                                  const int *__restrict__ idx,
                                                                  float *__restrict__ out, long long out_bstride) {
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * GP_CH_PER_BLOCK;
  const int c1 = min(c0 + GP_CH_PER_BLOCK, c);
  const int *ib = idx + (size_t)b * P;
  if (VEC4) {
    const int p = (blockIdx.x * GP_THREADS + threadIdx.x) * 4;
    if (p >= P) return;
    const int4 ii = *reinterpret_cast<const int4 *>(ib + p);
    const float *row0 = points + ((size_t)b * c + c0) * n;
    float *out0 = out + (size_t)b * out_bstride + (size_t)c0 * P + p;
    if (c1 - c0 == GP_CH_PER_BLOCK) {
      // full channel slice: all 32 gathers are in flight before the first store; the output is a
      // pure stream (never re-read by this kernel) and goes out non-temporal so that it does not
      // evict the channel rows being gathered from L2
      float4 v[GP_CH_PER_BLOCK];
#pragma unroll
      for (int l = 0; l < GP_CH_PER_BLOCK; ++l) {
        const float *row = row0 + (size_t)l * n;
        v[l].x = row[ii.x]; v[l].y = row[ii.y]; v[l].z = row[ii.z]; v[l].w = row[ii.w];
      }
#pragma unroll
      for (int l = 0; l < GP_CH_PER_BLOCK; ++l) {
        float *o = out0 + (size_t)l * P;
        __builtin_nontemporal_store(v[l].x, o + 0);
        __builtin_nontemporal_store(v[l].y, o + 1);
        __builtin_nontemporal_store(v[l].z, o + 2);
        __builtin_nontemporal_store(v[l].w, o + 3);
      }
    } else {
      for (int l = 0; l < c1 - c0; ++l) {
        const float *row = row0 + (size_t)l * n;
        float4 v;
        v.x = row[ii.x]; v.y = row[ii.y]; v.z = row[ii.z]; v.w = row[ii.w];
        *reinterpret_cast<float4 *>(out0 + (size_t)l * P) = v;
      }
    }
  } else {
    const int p = blockIdx.x * GP_THREADS + threadIdx.x;
    if (p >= P) return;
    const int ii = ib[p];
    for (int l = c0; l < c1; ++l)
      out[(size_t)b * out_bstride + (size_t)l * P + p] = points[((size_t)b * c + l) * n + ii];
  }
}

// LDS-staged variant for n <= GP_LDS_MAXN: the 4-byte gathers of the direct kernel are served by the
// vector L1 at roughly one 64-byte line per clock and lane group -- ~40 clocks per wave-gather when 64
// lanes hit a 4 KiB row at random -- which caps it near 3.7-3.9 TB/s of output.  Here a workgroup first
// copies its channel slice's rows (8 x n floats, coalesced 16-byte loads, L2-resident source) into
// LDS and gathers from there (a random 64-lane ds_read_b32 costs a few clocks of bank conflicts), then
// streams GP_CHUNK outputs per channel with non-temporal 16-byte stores: the store stream becomes
// the only HBM-rate traffic.
constexpr int GP_LDS_MAXN = 4096;     // 8 channels x 4096 floats = 128 KiB of LDS

// T threads per workgroup; the workgroup streams positions [blockIdx.x * per_wg, +per_wg) of its channel slice.
// NT: non-temporal stores (the output is never re-read by this kernel).
template <int T, bool NT>
__global__ __launch_bounds__(T) void group_points_lds_kernel(int c, int n, int P, int per_wg,
                                                             const float *__restrict__ points,
                                                             const int *__restrict__ idx,
                                                             float *__restrict__ out, long long out_bstride) {
  extern __shared__ __attribute__((aligned(16))) float rows[];   // [GP_CH_PER_BLOCK][n]
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * GP_CH_PER_BLOCK;
  const int nch = min(GP_CH_PER_BLOCK, c - c0);
  const float *src = points + ((size_t)b * c + c0) * n;
  const int total = nch * n;
  const int *ib = idx + (size_t)b * P;
  const int p_begin = blockIdx.x * per_wg;
  const int p_end = min(P, p_begin + per_wg);
  // the first indices are requested before the rows are staged: their latency hides under the copy
  int p = p_begin + threadIdx.x * 4;
  int4 ii = make_int4(0, 0, 0, 0);
  if (p < p_end) ii = *reinterpret_cast<const int4 *>(ib + p);
  if ((n & 3) == 0) {
    for (int i = threadIdx.x * 4; i < total; i += T * 4)
      *reinterpret_cast<float4 *>(rows + i) = *reinterpret_cast<const float4 *>(src + i);
  } else {
    for (int i = threadIdx.x; i < total; i += T) rows[i] = src[i];
  }
  __syncthreads();
  float *ob = out + (size_t)b * out_bstride + (size_t)c0 * P;     // out_bstride = c * P for a dense (B, C, S, K) result
  for (; p < p_end; p += T * 4) {
    const int pn = p + T * 4;
    int4 inext = ii;
    if (pn < p_end) inext = *reinterpret_cast<const int4 *>(ib + pn);
#pragma unroll
    for (int l = 0; l < GP_CH_PER_BLOCK; ++l) {
      if (l < nch) {
        const float *row = rows + l * n;
        typedef float gp_f32x4 __attribute__((ext_vector_type(4)));
        gp_f32x4 v;
        v.x = row[ii.x]; v.y = row[ii.y]; v.z = row[ii.z]; v.w = row[ii.w];
        gp_f32x4 *o = reinterpret_cast<gp_f32x4 *>(ob + (size_t)l * P + p);
        if (NT) __builtin_nontemporal_store(v, o);
        else *o = v;
      }
    }
    ii = inext;
  }
}

// grad_points[b,c,idx[b,p]] += grad_out[b,c,p]  (fp32 atomics into a zero-filled buffer).
__global__ __launch_bounds__(GP_THREADS) void group_points_grad_kernel(
    int c, int n, int P, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points, long long go_bstride) {
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * GP_CH_PER_BLOCK;
  const int c1 = min(c0 + GP_CH_PER_BLOCK, c);
  const int p = blockIdx.x * GP_THREADS + threadIdx.x;
  if (p >= P) return;
  const int ii = idx[(size_t)b * P + p];
  for (int l = c0; l < c1; ++l)
    atomicAdd(grad_points + ((size_t)b * c + l) * n + ii, grad_out[(size_t)b * go_bstride + (size_t)l * P + p]);
}

// Same sums through LDS: a workgroup owns (cloud b, a slice of CT channels, a contiguous range of positions
// p), keeps the slice's CT x n accumulators in LDS (ds_add_f32, no return), and adds its non-zero totals to
// grad_points once at the end -- S*K*C global fp32 atomics (two dozen G/s on contended rows) become LDS
// atomics plus one plain read-modify-write of the slice (or, when the positions of a slice are split over
// several workgroups, at most n*CT global atomics per workgroup).  idx is read once per position and reused for the CT
// channels; grad_out rows are read coalesced.  CT*n*4 <= 128 KiB (host picks CT), `splits` ranges of p per
// (b, slice) keep > 256 workgroups in flight when b*C/CT is small.
constexpr int GG_THREADS = 512;
constexpr int GG_LDS_BYTES = 128 * 1024;

__global__ __launch_bounds__(GG_THREADS) void group_points_grad_lds_kernel(
    int c, int n, int P, int ct, int per_split, int vec4, const float *__restrict__ grad_out,
    const int *__restrict__ idx, float *__restrict__ grad_points, long long go_bstride) {
  extern __shared__ __attribute__((aligned(16))) float gg_acc[];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * ct;
  const int nch = min(ct, c - c0);
  const int total = nch * n;
  for (int i = threadIdx.x; i < total; i += GG_THREADS) gg_acc[i] = 0.f;
  __syncthreads();
  const int p0 = blockIdx.x * per_split, p1 = min(P, p0 + per_split);
  const int *ib = idx + (size_t)b * P;
  const float *g0 = grad_out + (size_t)b * go_bstride + (size_t)c0 * P;     // go_bstride = c * P for a dense gradient
  auto add = [&](int l, int ii, float v) {
    __hip_atomic_fetch_add(gg_acc + l * n + ii, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  if (vec4) {     // P % 4 == 0, per_split % 4 == 0, 16-byte aligned rows: all loads of a step in flight together
    for (int p = p0 + threadIdx.x * 4; p < p1; p += GG_THREADS * 4) {
      const int4 ii = *reinterpret_cast<const int4 *>(ib + p);
      float4 g[GP_CH_PER_BLOCK];
#pragma unroll
      for (int l = 0; l < GP_CH_PER_BLOCK; ++l)
        if (l < nch) g[l] = *reinterpret_cast<const float4 *>(g0 + (size_t)l * P + p);
#pragma unroll
      for (int l = 0; l < GP_CH_PER_BLOCK; ++l)
        if (l < nch) { add(l, ii.x, g[l].x); add(l, ii.y, g[l].y); add(l, ii.z, g[l].z); add(l, ii.w, g[l].w); }
    }
  } else {
    for (int p = p0 + threadIdx.x; p < p1; p += GG_THREADS) {
      const int ii = ib[p];
      float g[GP_CH_PER_BLOCK];
#pragma unroll
      for (int l = 0; l < GP_CH_PER_BLOCK; ++l)
        if (l < nch) g[l] = g0[(size_t)l * P + p];
#pragma unroll
      for (int l = 0; l < GP_CH_PER_BLOCK; ++l)
        if (l < nch) add(l, ii, g[l]);
    }
  }
  __syncthreads();
  float *o = grad_points + ((size_t)b * c + c0) * n;
  if (gridDim.x == 1) {   // sole owner of these rows: plain read-modify-write (global fp32 atomics run at ~24 G/s)
    for (int i = threadIdx.x; i < total; i += GG_THREADS) o[i] += gg_acc[i];
    return;
  }
  for (int i = threadIdx.x; i < total; i += GG_THREADS) {
    const float v = gg_acc[i];
    if (v != 0.f) atomicAdd(o + i, v);     // += like the reference (caller zero-fills)
  }
}

// Atomics-free, run-to-run deterministic scatter-add: the (b, p) -> n map is
// inverted on the host side of the C ABI (a stable sort of idx per cloud: `perm` lists the p of every
// source point n contiguously and in ascending p, `seg` holds the n+1 segment boundaries); one thread
// per (cloud, source point) then adds its segment in that fixed order for the block's channel slice.
// grad_points needs no zero fill (every element is written).  Same values as the atomic kernel up to
// fp32 summation order.
__global__ __launch_bounds__(GP_THREADS) void group_points_grad_sorted_kernel(
    int c, int n, int P, const float *__restrict__ grad_out, const int *__restrict__ perm,
    const int *__restrict__ seg, float *__restrict__ grad_points, long long go_bstride) {
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * GP_CH_PER_BLOCK;
  const int nch = min(GP_CH_PER_BLOCK, c - c0);
  const int i = blockIdx.x * GP_THREADS + threadIdx.x;
  if (i >= n) return;
  const int *pb = perm + (size_t)b * P;
  const int s0 = seg[(size_t)b * (n + 1) + i], s1 = seg[(size_t)b * (n + 1) + i + 1];
  float acc[GP_CH_PER_BLOCK];
#pragma unroll
  for (int l = 0; l < GP_CH_PER_BLOCK; ++l) acc[l] = 0.f;
  const float *g = grad_out + (size_t)b * go_bstride + (size_t)c0 * P;
  for (int j = s0; j < s1; ++j) {
    const int p = pb[j];
#pragma unroll
    for (int l = 0; l < GP_CH_PER_BLOCK; ++l)
      if (l < nch) acc[l] += g[(size_t)l * P + p];
  }
#pragma unroll
  for (int l = 0; l < GP_CH_PER_BLOCK; ++l)
    if (l < nch) grad_points[((size_t)b * c + c0 + l) * n + i] = acc[l];
}


// ---- cost-volume inputs (PW/costvolume.py:92-107, 155-166) -----------------------------------------------------------
// The reference builds the 10-channel geometry encoding [p, q, q - p, |q - p|] of every (centre p, neighbour q) pair with
// tile / grouping_operation / sub / square / sum / sqrt / cat, and tiles the centre's features over the K neighbours before a
// second cat: a dozen element-wise launches forward, two dozen backward, each a full pass over a (B, C, S, K) tensor.  These
// four kernels write / differentiate those channels directly in their slice of the concatenated MLP input (`out` points
// at the slice's first channel of cloud 0, rows of consecutive clouds `bstride` floats apart).  Forward values are the
// reference's bit for bit: the same fp32 subtraction, products and left-to-right sum, no fused multiply-add.
constexpr int GEO_THREADS = 256;

__global__ __launch_bounds__(GEO_THREADS) void geometry_encode_kernel(int n, int s, int k, const float *__restrict__ centre,
                                                                      const float *__restrict__ src,
                                                                      const int *__restrict__ idx, float *__restrict__ out,
                                                                      long long bstride) {
  const int b = blockIdx.y;
  const int P = s * k;
  const int p = blockIdx.x * GEO_THREADS + threadIdx.x;
  if (p >= P) return;
  const int j = p / k;
  const int i = idx[(size_t)b * P + p];
  const float *c = centre + (size_t)b * 3 * s, *q = src + (size_t)b * 3 * n;
  const float px = c[j], py = c[s + j], pz = c[2 * s + j];
  const float qx = q[i], qy = q[n + i], qz = q[2 * n + i];
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  const float e = sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)), 1e-20f));
  float *o = out + (size_t)b * bstride + p;
  o[0] = px; o[(size_t)P] = py; o[(size_t)2 * P] = pz;
  o[(size_t)3 * P] = qx; o[(size_t)4 * P] = qy; o[(size_t)5 * P] = qz;
  o[(size_t)6 * P] = dx; o[(size_t)7 * P] = dy; o[(size_t)8 * P] = dz;
  o[(size_t)9 * P] = e;
}

// out[b, 0:3, j, t] = src[b, :, idx[b, j, t]] - centre[b, :, j]: the grouped coordinates relative to their centre
// (P2/pointnet2_modules.py:215-218, 485-488: grouping_operation, then the subtraction of the tiled centres).
__global__ __launch_bounds__(GEO_THREADS) void xyz_diff_kernel(int n, int s, int k, const float *__restrict__ centre,
                                                               const float *__restrict__ src, const int *__restrict__ idx,
                                                               float *__restrict__ out, long long bstride) {
  const int b = blockIdx.y;
  const int P = s * k;
  const int p = blockIdx.x * GEO_THREADS + threadIdx.x;
  if (p >= P) return;
  const int j = p / k;
  const int i = idx[(size_t)b * P + p];
  const float *c = centre + (size_t)b * 3 * s, *q = src + (size_t)b * 3 * n;
  float *o = out + (size_t)b * bstride + p;
  o[0] = q[i] - c[j];
  o[(size_t)P] = q[n + i] - c[s + j];
  o[(size_t)2 * P] = q[2 * n + i] - c[2 * s + j];
}

// One thread per centre: the K neighbour gradients of a centre are consecutive.  d_centre (B, 3, S) is written (the sum over
// the centre's K pairs, in neighbour order: deterministic); d_src (B, 3, N), when wanted, is zero-filled by the caller and
// receives atomic adds (or none at all: the pyramid's coordinates need no gradient); d_pair (B, 3, S, K) instead takes the
// neighbours' gradients pair by pair, for the caller's atomics-free sorted scatter (group_points_grad_sorted).
__global__ __launch_bounds__(GEO_THREADS) void geometry_encode_grad_kernel(int n, int s, int k, const float *__restrict__ centre,
                                                                           const float *__restrict__ src,
                                                                           const int *__restrict__ idx,
                                                                           const float *__restrict__ g, long long bstride,
                                                                           float *__restrict__ d_centre,
                                                                           float *__restrict__ d_src,
                                                                           float *__restrict__ d_pair) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * GEO_THREADS + threadIdx.x;
  if (j >= s) return;
  const int P = s * k;
  const float *c = centre + (size_t)b * 3 * s, *q = src + (size_t)b * 3 * n;
  const float px = c[j], py = c[s + j], pz = c[2 * s + j];
  const float *gb = g + (size_t)b * bstride + (size_t)j * k;
  const int *ib = idx + (size_t)b * P + (size_t)j * k;
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int t = 0; t < k; ++t) {
    const int i = ib[t];
    const float dx = q[i] - px, dy = q[n + i] - py, dz = q[2 * n + i] - pz;
    const float e = sqrtf(__fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz)), 1e-20f));
    const float ge = gb[(size_t)9 * P + t] / e;                      // d|q - p| / d(q - p) = (q - p) / |q - p|
    const float ddx = gb[(size_t)6 * P + t] + ge * dx, ddy = gb[(size_t)7 * P + t] + ge * dy,
                ddz = gb[(size_t)8 * P + t] + ge * dz;              // gradient w.r.t. the difference q - p
    ax += gb[t] - ddx;
    ay += gb[(size_t)P + t] - ddy;
    az += gb[(size_t)2 * P + t] - ddz;
    if (d_src != nullptr) {
      float *ds = d_src + (size_t)b * 3 * n;
      atomicAdd(ds + i, gb[(size_t)3 * P + t] + ddx);
      atomicAdd(ds + n + i, gb[(size_t)4 * P + t] + ddy);
      atomicAdd(ds + 2 * n + i, gb[(size_t)5 * P + t] + ddz);
    } else if (d_pair != nullptr) {          // per-pair neighbour gradients (b, 3, s, k) for the atomics-free scatter
      float *dp = d_pair + (size_t)b * 3 * P + (size_t)j * k + t;
      dp[0] = gb[(size_t)3 * P + t] + ddx;
      dp[(size_t)P] = gb[(size_t)4 * P + t] + ddy;
      dp[(size_t)2 * P] = gb[(size_t)5 * P + t] + ddz;
    }
  }
  if (d_centre != nullptr) {
    float *dc = d_centre + (size_t)b * 3 * s;
    dc[j] = ax; dc[s + j] = ay; dc[2 * s + j] = az;
  }
}

// out[b, c, j, t] = feats[b, c, j] for t < k (the reference's torch.tile of the centre features); grid (P / 4 / T, c, b).
__global__ __launch_bounds__(GEO_THREADS) void broadcast_centre_kernel(int c, int s, int k, const float *__restrict__ feats,
                                                                       float *__restrict__ out, long long bstride) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int P = s * k;
  const int p = (blockIdx.x * GEO_THREADS + threadIdx.x) * 4;
  if (p >= P) return;
  const float *f = feats + ((size_t)b * c + ch) * s;
  float *o = out + (size_t)b * bstride + (size_t)ch * P + p;
  if (p + 3 < P && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v = {f[p / k], f[(p + 1) / k], f[(p + 2) / k], f[(p + 3) / k]};
    *reinterpret_cast<f4 *>(o) = v;
  } else {
    for (int u = 0; u < 4 && p + u < P; ++u) o[u] = f[(p + u) / k];
  }
}

// d_feats[b, c, j] = sum_t grad[b, c, j, t] in neighbour order; one thread per (c, j).
__global__ __launch_bounds__(GEO_THREADS) void broadcast_centre_grad_kernel(int c, int s, int k, const float *__restrict__ g,
                                                                            long long bstride, float *__restrict__ d_feats) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int j = blockIdx.x * GEO_THREADS + threadIdx.x;
  if (j >= s) return;
  const float *gb = g + (size_t)b * bstride + ((size_t)ch * s + j) * k;
  float a = 0.f;
  if ((k & 3) == 0 && ((reinterpret_cast<uintptr_t>(gb) & 15) == 0)) {
    for (int t = 0; t < k; t += 4) {
      const float4 v = *reinterpret_cast<const float4 *>(gb + t);
      a += v.x; a += v.y; a += v.z; a += v.w;
    }
  } else {
    for (int t = 0; t < k; ++t) a += gb[t];
  }
  d_feats[((size_t)b * c + ch) * s + j] = a;
}

}  // namespace pwclo

using namespace pwclo;

// `*_strided` forms: the grouped tensor / its gradient is a channel slice of a larger (B, Ctot, S, K) tensor -- the concatenated
// input of a shared MLP (P2/pointnet2_modules.py:222,490: cat of the grouped features with the coordinate differences) --
// so rows of consecutive clouds are `batch_stride` floats apart instead of c * npoints * nsample.  The module path groups
// straight into / differentiates straight out of that tensor: no torch.cat, no contiguous() copy of the slice.
static void gp_grad_sorted(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *perm,
                           const int *seg, float *grad_points, long long go_bstride);
static void gp_forward(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out,
                       long long out_bstride);
static void gp_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                    float *grad_points, long long go_bstride);

extern "C" void group_points_grad_sorted_kernel_wrapper(int b, int c, int n, int npoints, int nsample,
                                                        const float *grad_out, const int *perm, const int *seg,
                                                        float *grad_points) {
  gp_grad_sorted(b, c, n, npoints, nsample, grad_out, perm, seg, grad_points, (long long)c * npoints * nsample);
}
extern "C" void group_points_grad_sorted_strided_kernel_wrapper(int b, int c, int n, int npoints, int nsample,
                                                                const float *grad_out, long long batch_stride,
                                                                const int *perm, const int *seg, float *grad_points) {
  gp_grad_sorted(b, c, n, npoints, nsample, grad_out, perm, seg, grad_points, batch_stride);
}
extern "C" void group_points_strided_kernel_wrapper(int b, int c, int n, int npoints, int nsample, const float *points,
                                                    const int *idx, float *out, long long batch_stride) {
  gp_forward(b, c, n, npoints, nsample, points, idx, out, batch_stride);
}
extern "C" void group_points_grad_strided_kernel_wrapper(int b, int c, int n, int npoints, int nsample,
                                                         const float *grad_out, long long batch_stride, const int *idx,
                                                         float *grad_points) {
  gp_grad(b, c, n, npoints, nsample, grad_out, idx, grad_points, batch_stride);
}

static void gp_grad_sorted(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *perm,
                           const int *seg, float *grad_points, long long go_bstride) {
  if (b <= 0 || c <= 0 || n <= 0) return;
  const long long P64 = (long long)npoints * nsample;
  PWCLO_REQUIRE(P64 < (1ll << 31) && b <= 65535, "group_points_grad_sorted: npoints*nsample=%lld or b=%d too large",
                P64, b);
  hipLaunchKernelGGL(group_points_grad_sorted_kernel, dim3(ceil_div(n, GP_THREADS), ceil_div(c, GP_CH_PER_BLOCK), b),
                     dim3(GP_THREADS), 0, current_stream(), c, n, (int)P64, grad_out, perm, seg, grad_points, go_bstride);
  check_launch("group_points_grad_sorted");
}

extern "C" void group_points_kernel_wrapper(int b, int c, int n, int npoints, int nsample,
                                            const float *points, const int *idx, float *out) {
  gp_forward(b, c, n, npoints, nsample, points, idx, out, (long long)c * npoints * nsample);
}

static void gp_forward(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx, float *out,
                       long long out_bstride) {
  if (b <= 0 || c <= 0 || npoints <= 0 || nsample <= 0) return;
  const long long P64 = (long long)npoints * nsample;
  PWCLO_REQUIRE(P64 < (1ll << 31) && b <= 65535, "group_points: npoints*nsample=%lld or b=%d too large",
                P64, b);
  const int P = (int)P64;
  const int gy = ceil_div(c, GP_CH_PER_BLOCK);
  const bool vec = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                   ((reinterpret_cast<uintptr_t>(idx) & 15) == 0);
  static int use_lds = -1;
  if (use_lds < 0) { const char *e = getenv("PWCLO_GROUP_LDS"); use_lds = e ? atoi(e) : 1; }
  // rows of one channel slice in LDS: min(C, 8) x n floats, at most 128 KiB (8 x 4096, or e.g. the 3 coordinate rows of an
  // 8192-point cloud)
  const long long lds_need = (long long)min(c, GP_CH_PER_BLOCK) * n * 4;
  if (vec && use_lds && lds_need <= (long long)GP_CH_PER_BLOCK * GP_LDS_MAXN * 4 && (reinterpret_cast<uintptr_t>(points) & 15) == 0) {
    const int lds_bytes = (int)lds_need;
    // Workgroup shape: 1024 threads (16 waves: one workgroup saturates a CU's store path) when the slices alone give every
    // CU a workgroup, 256 threads otherwise; positions are split over just enough workgroups to put ~2 per CU (LDS
    // permitting), so a channel slice's rows are staged once or twice, not once per 4096 positions (round 3: the
    // 4096-position chunks re-staged the rows 4x at the largest shape and left the kernel at 0.60 of the HBM roof).
    static int t_env = -1, tgt_env = -1, nt_env = -1;
    if (t_env < 0) { const char *e = getenv("PWCLO_GP_THREADS"); t_env = e ? atoi(e) : 0; }
    if (tgt_env < 0) { const char *e = getenv("PWCLO_GP_TARGET"); tgt_env = e ? atoi(e) : 0; }
    if (nt_env < 0) { const char *e = getenv("PWCLO_GP_NT"); nt_env = e ? atoi(e) : 1; }
    const long long slices = (long long)b * gy;
    int T = t_env ? t_env : 1024;
    // measured (profiles/r03: tools/gp_sweep.sh): two workgroups per CU pay while a slice's rows are <= 32 KiB; with 64 KiB
    // rows a second staging per slice costs more than the overlap gives
    const int target = tgt_env ? tgt_env : (lds_bytes < 64 * 1024 ? 512 : 256);
    int gx = (int)max(1LL, (target + slices - 1) / slices);
    const int max_gx = ceil_div(P, T * 4);
    if (gx > max_gx) gx = max_gx;
    const int per_wg = ceil_div(ceil_div(P, gx), T * 4) * T * 4;
    gx = ceil_div(P, per_wg);
#define GP_LAUNCH(TT, NTT)                                                                                      \
    {                                                                                                           \
      static bool attr_set = false;                                                                             \
      if (lds_bytes > 64 * 1024 && !attr_set) {                                                                 \
        (void)hipFuncSetAttribute((const void *)group_points_lds_kernel<TT, NTT>,                               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, GP_CH_PER_BLOCK * GP_LDS_MAXN * 4); \
        attr_set = true;                                                                                        \
      }                                                                                                         \
      hipLaunchKernelGGL((group_points_lds_kernel<TT, NTT>), dim3(gx, gy, b), dim3(TT), lds_bytes,              \
                         current_stream(), c, n, P, per_wg, points, idx, out, out_bstride);                     \
    }
    if (T == 1024 && nt_env) GP_LAUNCH(1024, true)
    else if (T == 1024) GP_LAUNCH(1024, false)
    else if (T == 512 && nt_env) GP_LAUNCH(512, true)
    else if (T == 512) GP_LAUNCH(512, false)
    else if (nt_env) GP_LAUNCH(256, true)
    else GP_LAUNCH(256, false)
#undef GP_LAUNCH
    check_launch("group_points");
    return;
  }
  if (vec)
    hipLaunchKernelGGL(group_points_kernel<true>, dim3(ceil_div(P / 4, GP_THREADS), gy, b),
                       dim3(GP_THREADS), 0, current_stream(), c, n, P, points, idx, out, out_bstride);
  else
    hipLaunchKernelGGL(group_points_kernel<false>, dim3(ceil_div(P, GP_THREADS), gy, b),
                       dim3(GP_THREADS), 0, current_stream(), c, n, P, points, idx, out, out_bstride);
  check_launch("group_points");
}

extern "C" void group_points_grad_kernel_wrapper(int b, int c, int n, int npoints, int nsample,
                                                 const float *grad_out, const int *idx,
                                                 float *grad_points) {
  gp_grad(b, c, n, npoints, nsample, grad_out, idx, grad_points, (long long)c * npoints * nsample);
}

static void gp_grad(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                    float *grad_points, long long go_bstride) {
  if (b <= 0 || c <= 0 || npoints <= 0 || nsample <= 0) return;
  const long long P64 = (long long)npoints * nsample;
  PWCLO_REQUIRE(P64 < (1ll << 31) && b <= 65535,
                "group_points_grad: npoints*nsample=%lld or b=%d too large", P64, b);
  const int P = (int)P64;
  static int use_lds = -1;
  if (use_lds < 0) { const char *e = getenv("PWCLO_GRAD_LDS"); use_lds = e ? atoi(e) : 1; }
  if (use_lds && (long long)n * 4 <= GG_LDS_BYTES) {
    int ct = 8;
    while ((long long)ct * n * 4 > GG_LDS_BYTES) ct >>= 1;
    if (ct > c) ct = c;
    while (ct > 1 && b * ceil_div(c, ct) < 256) ct >>= 1;        // narrower slices before splitting positions
    const int slices = ceil_div(c, ct);
    int splits = b * slices >= 64 ? 1 : ceil_div(256, b * slices);   // split ranges flush with global atomics
    splits = max(1, min(splits, ceil_div(P, 4 * GG_THREADS)));
    const int per_split = ceil_div(ceil_div(P, splits), 4) * 4;
    const int vec4 = (P % 4 == 0) && ((reinterpret_cast<uintptr_t>(grad_out) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(idx) & 15) == 0);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void *)group_points_grad_lds_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, GG_LDS_BYTES);
      attr_set = true;
    }
    hipLaunchKernelGGL(group_points_grad_lds_kernel, dim3(ceil_div(P, per_split), slices, b), dim3(GG_THREADS),
                       (size_t)ct * n * 4, current_stream(), c, n, P, ct, per_split, vec4, grad_out, idx, grad_points, go_bstride);
    check_launch("group_points_grad");
    return;
  }
  hipLaunchKernelGGL(group_points_grad_kernel, dim3(ceil_div(P, GP_THREADS), ceil_div(c, GP_CH_PER_BLOCK), b),
                     dim3(GP_THREADS), 0, current_stream(), c, n, P, grad_out, idx, grad_points, go_bstride);
  check_launch("group_points_grad");
}

extern "C" void geometry_encode_kernel_wrapper(int b, int n, int s, int k, const float *centre_xyz, const float *src_xyz,
                                               const int *idx, float *out, long long batch_stride) {
  if (b <= 0 || s <= 0 || k <= 0) return;
  PWCLO_REQUIRE(n > 0 && b <= 65535 && (long long)s * k < (1ll << 31), "geometry_encode: b=%d n=%d s*k=%lld out of range", b, n,
                (long long)s * k);
  hipLaunchKernelGGL(geometry_encode_kernel, dim3(ceil_div(s * k, GEO_THREADS), b), dim3(GEO_THREADS), 0, current_stream(), n, s,
                     k, centre_xyz, src_xyz, idx, out, batch_stride);
  check_launch("geometry_encode");
}

extern "C" void geometry_encode_grad_kernel_wrapper(int b, int n, int s, int k, const float *centre_xyz, const float *src_xyz,
                                                    const int *idx, const float *grad_out, long long batch_stride,
                                                    float *d_centre_xyz, float *d_src_xyz, float *d_pair) {
  if (b <= 0 || s <= 0 || k <= 0 || (d_centre_xyz == nullptr && d_src_xyz == nullptr && d_pair == nullptr)) return;
  PWCLO_REQUIRE(d_src_xyz == nullptr || d_pair == nullptr, "geometry_encode_grad: d_src_xyz and d_pair are alternatives%s", "");
  PWCLO_REQUIRE(n > 0 && b <= 65535 && (long long)s * k < (1ll << 31), "geometry_encode_grad: b=%d n=%d s*k=%lld out of range", b,
                n, (long long)s * k);
  hipLaunchKernelGGL(geometry_encode_grad_kernel, dim3(ceil_div(s, GEO_THREADS), b), dim3(GEO_THREADS), 0, current_stream(), n, s,
                     k, centre_xyz, src_xyz, idx, grad_out, batch_stride, d_centre_xyz, d_src_xyz, d_pair);
  check_launch("geometry_encode_grad");
}

extern "C" void broadcast_centre_kernel_wrapper(int b, int c, int s, int k, const float *feats, float *out,
                                                long long batch_stride) {
  if (b <= 0 || c <= 0 || s <= 0 || k <= 0) return;
  PWCLO_REQUIRE(b <= 65535 && c <= 65535 && (long long)s * k < (1ll << 31), "broadcast_centre: b=%d c=%d s*k=%lld out of range", b,
                c, (long long)s * k);
  hipLaunchKernelGGL(broadcast_centre_kernel, dim3(ceil_div(ceil_div(s * k, 4), GEO_THREADS), c, b), dim3(GEO_THREADS), 0,
                     current_stream(), c, s, k, feats, out, batch_stride);
  check_launch("broadcast_centre");
}

extern "C" void broadcast_centre_grad_kernel_wrapper(int b, int c, int s, int k, const float *grad_out, long long batch_stride,
                                                     float *d_feats) {
  if (b <= 0 || c <= 0 || s <= 0 || k <= 0) return;
  PWCLO_REQUIRE(b <= 65535 && c <= 65535, "broadcast_centre_grad: b=%d c=%d exceed the grid limits", b, c);
  hipLaunchKernelGGL(broadcast_centre_grad_kernel, dim3(ceil_div(s, GEO_THREADS), c, b), dim3(GEO_THREADS), 0, current_stream(), c,
                     s, k, grad_out, batch_stride, d_feats);
  check_launch("broadcast_centre_grad");
}

extern "C" void xyz_diff_kernel_wrapper(int b, int n, int s, int k, const float *centre_xyz, const float *src_xyz, const int *idx,
                                        float *out, long long batch_stride) {
  if (b <= 0 || s <= 0 || k <= 0) return;
  PWCLO_REQUIRE(n > 0 && b <= 65535 && (long long)s * k < (1ll << 31), "xyz_diff: b=%d n=%d s*k=%lld out of range", b, n,
                (long long)s * k);
  hipLaunchKernelGGL(xyz_diff_kernel, dim3(ceil_div(s * k, GEO_THREADS), b), dim3(GEO_THREADS), 0, current_stream(), n, s, k,
                     centre_xyz, src_xyz, idx, out, batch_stride);
  check_launch("xyz_diff");
}
