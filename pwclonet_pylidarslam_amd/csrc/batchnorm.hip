// Training-mode BatchNorm over (B, C, L) / (B, C, S, K) activations for gfx950.
//
// The module path of PWCLO-Net normalises (B, C, S, K) tensors with FEW channels and very long rows
// (set abstraction level 1: C = 8..16, S*K = 65 536 per cloud).  The stock kernels assign their
// parallelism per channel and run those shapes an order of magnitude below the HBM rate (measured: 2.5 ms
// forward for a 268 MB tensor with C = 16).  Here every pass is a grid over (position range, channel):
//   forward   stats:  per-(channel, range) partial sum / sum of squares in fp64  -> workspace
//             finish: mean, biased variance, 1/sqrt(var+eps), running statistics (momentum, unbiased variance)
//             apply:  y = (x - mean) * invstd * gamma + beta
//   backward  reduce: partial sum(dy), sum(dy * xhat) in fp64                      -> workspace
//             finish: dgamma, dbeta
//             apply:  dx = gamma * invstd * (dy - dbeta/M - xhat * dgamma/M)
// HBM-bound: 3 passes forward (2 reads + 1 write), 5 backward; fp64 accumulation keeps the statistics
// independent of the split and within fp32 rounding of a two-pass evaluation.
#include <math.h>
#include <stdint.h>

#include "common.hpp"
#include "train_internal.hpp"

namespace pwclo {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_SPLITS = 256;

__device__ __forceinline__ double bn_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum of (a, b) over the workgroup; valid in thread 0.
__device__ __forceinline__ void bn_block_sum2(double &a, double &b) {
  __shared__ double red[2][BN_THREADS / 64];
  a = bn_wave_sum(a);
  b = bn_wave_sum(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = 0.0; b = 0.0;
#pragma unroll
    for (int w = 0; w < BN_THREADS / 64; ++w) { a += red[0][w]; b += red[1][w]; }
  }
}

// One workgroup: channel blockIdx.y, flat positions [blockIdx.x*per_split, +per_split) of the B*L elements of
// that channel (position f = b*L + l).  MODE 0: (sum x, sum x^2).  MODE 1: (sum dy, sum dy*xhat).
// RELU (MODE 1): the forward applied max(., 0) after the affine map; dy is masked where that output was 0,
// the mask recomputed from x with the forward's own expression (no saved activation, no extra pass).
template <int MODE, bool RELU>
__global__ __launch_bounds__(BN_THREADS) void bn_partial_kernel(int C, int L, long long M, long long per_split,
                                                                const float *__restrict__ x,
                                                                const float *__restrict__ dy,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ invstd,
                                                                const float *__restrict__ gamma,
                                                                const float *__restrict__ beta,
                                                                double *__restrict__ partial) {
  const int ch = blockIdx.y;
  const long long f0 = (long long)blockIdx.x * per_split;
  const long long f1 = f0 + per_split < M ? f0 + per_split : M;
  const float mu = MODE == 1 ? mean[ch] : 0.f, is = MODE == 1 ? invstd[ch] : 0.f;
  const float ga = (RELU && gamma != nullptr) ? gamma[ch] : 1.f, be = (RELU && beta != nullptr) ? beta[ch] : 0.f;
  auto masked = [&](float g, float xh) -> float { return (!RELU || xh * ga + be > 0.f) ? g : 0.f; };
  double s0 = 0.0, s1 = 0.0;
  const bool vec = (L % 4 == 0);           // per_split is a multiple of 4: a float4 never straddles a row
  const int step = vec ? BN_THREADS * 4 : BN_THREADS;
  long long f = f0 + (long long)threadIdx.x * (vec ? 4 : 1);
  int b = (int)(f / L);                    // one division per thread; (b, l) advance incrementally below
  int l = (int)(f - (long long)b * L);
  for (; f < f1; f += step) {
    const size_t off = ((size_t)b * C + ch) * (size_t)L + l;
    if (vec) {
      const float4 v = *reinterpret_cast<const float4 *>(x + off);
      if (MODE == 0) {
        s0 += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        s1 += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
      } else {
        float4 g = *reinterpret_cast<const float4 *>(dy + off);
        const float hx = (v.x - mu) * is, hy = (v.y - mu) * is, hz = (v.z - mu) * is, hw = (v.w - mu) * is;
        g.x = masked(g.x, hx); g.y = masked(g.y, hy); g.z = masked(g.z, hz); g.w = masked(g.w, hw);
        s0 += ((double)g.x + (double)g.y) + ((double)g.z + (double)g.w);
        s1 += ((double)g.x * hx + (double)g.y * hy) + ((double)g.z * hz + (double)g.w * hw);
      }
    } else {
      const float v = x[off];
      if (MODE == 0) {
        s0 += (double)v;
        s1 += (double)v * v;
      } else {
        const float xh = (v - mu) * is;
        const float g = masked(dy[off], xh);
        s0 += (double)g;
        s1 += (double)g * xh;
      }
    }
    l += step;
    while (l >= L) { l -= L; ++b; }
  }
  bn_block_sum2(s0, s1);
  if (threadIdx.x == 0) {
    partial[((size_t)ch * gridDim.x + blockIdx.x) * 2 + 0] = s0;
    partial[((size_t)ch * gridDim.x + blockIdx.x) * 2 + 1] = s1;
  }
}

// One WAVE per channel folds the partials: lane q sums splits q, q + 64, ... in order, the 64 lane sums are combined
// by a fixed butterfly -- a fixed summation order (deterministic), a sixth of the latency of one thread walking up to
// 256 partials (these two kernels run once per BatchNorm call: ~190 times per training step).
__device__ __forceinline__ void bn_fold(const double *__restrict__ partial, int ch, int nsplit, double &s0, double &s1) {
  const int lane = threadIdx.x & 63;
  s0 = 0.0; s1 = 0.0;
  for (int s = lane; s < nsplit; s += 64) {
    s0 += partial[((size_t)ch * nsplit + s) * 2 + 0];
    s1 += partial[((size_t)ch * nsplit + s) * 2 + 1];
  }
  s0 = bn_wave_sum(s0);
  s1 = bn_wave_sum(s1);
}

__global__ __launch_bounds__(256) void bn_forward_finish_kernel(int C, int nsplit, long long M, float eps, float momentum,
                                                                const double *__restrict__ partial,
                                                                float *__restrict__ running_mean,
                                                                float *__restrict__ running_var,
                                                                float *__restrict__ save_mean,
                                                                float *__restrict__ save_invstd) {
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);        // 4 waves = 4 channels per workgroup
  if (ch >= C) return;
  double s0, s1;
  bn_fold(partial, ch, nsplit, s0, s1);
  if ((threadIdx.x & 63) != 0) return;
  const double mean = s0 / (double)M;
  double var = s1 / (double)M - mean * mean;     // fp64: no visible cancellation for fp32 data
  if (var < 0.0) var = 0.0;
  save_mean[ch] = (float)mean;
  save_invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean != nullptr) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[ch] = (float)((1.0 - (double)momentum) * (double)running_mean[ch] + (double)momentum * mean);
    running_var[ch] = (float)((1.0 - (double)momentum) * (double)running_var[ch] + (double)momentum * unbiased);
  }
}

__global__ __launch_bounds__(256) void bn_backward_finish_kernel(int C, int nsplit, const double *__restrict__ partial,
                                                                 float *__restrict__ dgamma, float *__restrict__ dbeta) {
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= C) return;
  double s0, s1;
  bn_fold(partial, ch, nsplit, s0, s1);
  if ((threadIdx.x & 63) != 0) return;
  dbeta[ch] = (float)s0;
  dgamma[ch] = (float)s1;
}

// Element-wise passes: grid (ceil(L / (4*BN_THREADS)) or ceil(L / BN_THREADS), C, B).
// MODE 0: y = (x - mean) * invstd * gamma + beta.   MODE 1: dx (see header).
template <int MODE, bool VEC, bool RELU>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(int C, int L, float inv_m,
                                                              const float *__restrict__ x,
                                                              const float *__restrict__ dy,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta,
                                                              const float *__restrict__ mean,
                                                              const float *__restrict__ invstd,
                                                              const float *__restrict__ dgamma,
                                                              const float *__restrict__ dbeta,
                                                              float *__restrict__ out) {
  const int ch = blockIdx.y, b = blockIdx.z;
  const size_t row = ((size_t)b * C + ch) * (size_t)L;
  const float mu = mean[ch], is = invstd[ch];
  const float g = gamma != nullptr ? gamma[ch] : 1.f;
  const float be = beta != nullptr ? beta[ch] : 0.f;
  float c0 = 0.f, c1 = 0.f;
  if (MODE == 1) {
    c0 = dbeta[ch] * inv_m;
    c1 = dgamma[ch] * inv_m;
  }
  auto f = [&](float xv, float gv) -> float {
    const float xh = (xv - mu) * is;
    const float yv = xh * g + be;
    if (MODE == 0) return RELU ? relu_nan(yv) : yv;
    if (RELU && !(yv > 0.f)) gv = 0.f;
    return ((gv - c0) - xh * c1) * (g * is);
  };
  if (VEC) {
    const int l = (blockIdx.x * BN_THREADS + threadIdx.x) * 4;
    if (l >= L) return;
    const float4 xv = *reinterpret_cast<const float4 *>(x + row + l);
    float4 gv = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) gv = *reinterpret_cast<const float4 *>(dy + row + l);
    float4 o;
    o.x = f(xv.x, gv.x); o.y = f(xv.y, gv.y); o.z = f(xv.z, gv.z); o.w = f(xv.w, gv.w);
    *reinterpret_cast<float4 *>(out + row + l) = o;
  } else {
    const int l = blockIdx.x * BN_THREADS + threadIdx.x;
    if (l >= L) return;
    out[row + l] = f(x[row + l], MODE == 1 ? dy[row + l] : 0.f);
  }
}

// Tail of a grouped stack: BatchNorm -> ReLU -> max over the K neighbours, without writing the (B, C, S, K)
// activation.  K in {4, 8, 16, 32}: K / 4 neighbouring lanes hold one row (a float4 each) and combine (value, first
// index) pairs by shuffles.  Per row: pooled = max_k relu(bn(x_k)), arg = the first k reaching it, xsel = x at arg
// (all the backward's reductions need, see batchnorm_train_relu_maxk_backward).
template <int K>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_relu_maxk_kernel(int C, int S, const float *__restrict__ x,
                                                                       const float *__restrict__ gamma,
                                                                       const float *__restrict__ beta,
                                                                       const float *__restrict__ mean,
                                                                       const float *__restrict__ invstd,
                                                                       float *__restrict__ pooled,
                                                                       unsigned char *__restrict__ arg,
                                                                       float *__restrict__ xsel) {
  constexpr int LPR = K / 4;
  const int ch = blockIdx.y, b = blockIdx.z;
  const size_t plane = (size_t)b * C + ch;
  const long long q = (long long)blockIdx.x * BN_THREADS + threadIdx.x;     // float4 index inside the (S, K) plane
  if (q >= (long long)S * LPR) return;                                       // whole rows leave together (LPR | 64)
  const int srow = (int)(q / LPR), part = (int)(q - (long long)srow * LPR);
  const float mu = mean[ch], is = invstd[ch];
  const float g = gamma != nullptr ? gamma[ch] : 1.f;
  const float be = beta != nullptr ? beta[ch] : 0.f;
  const float4 xv = *reinterpret_cast<const float4 *>(x + plane * (size_t)S * K + (size_t)q * 4);
  const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
  float best = -1.f, bx = 0.f;
  int bi = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float yv = relu_nan((xs[e] - mu) * is * g + be);
    if (yv > best || yv != yv) { best = yv; bi = part * 4 + e; bx = xs[e]; }
  }
#pragma unroll
  for (int off = 1; off < LPR; off <<= 1) {
    const float ob = __shfl_xor(best, off, 64), ox = __shfl_xor(bx, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ob > best || ob != ob || (ob == best && oi < bi)) { best = ob; bi = oi; bx = ox; }   // a NaN wins, like torch.max
  }
  if (part == 0) {
    const size_t o = plane * (size_t)S + srow;
    pooled[o] = best;
    arg[o] = (unsigned char)bi;
    xsel[o] = bx;
  }
}

// dx of the tail above from the pooled gradient: dy is dpool at (row, arg) where the pooled value was positive and 0
// elsewhere, never materialised; dx = gamma * invstd * (dy - dbeta / M - xhat * dgamma / M), M = B * S * K.
template <int K>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_bwd_maxk_kernel(int C, int S, float inv_m,
                                                                      const float *__restrict__ x,
                                                                      const float *__restrict__ dpool,
                                                                      const unsigned char *__restrict__ arg,
                                                                      const float *__restrict__ gamma,
                                                                      const float *__restrict__ beta,
                                                                      const float *__restrict__ mean,
                                                                      const float *__restrict__ invstd,
                                                                      const float *__restrict__ dgamma,
                                                                      const float *__restrict__ dbeta,
                                                                      float *__restrict__ dx) {
  constexpr int LPR = K / 4;
  const int ch = blockIdx.y, b = blockIdx.z;
  const size_t plane = (size_t)b * C + ch;
  const long long q = (long long)blockIdx.x * BN_THREADS + threadIdx.x;
  if (q >= (long long)S * LPR) return;
  const int srow = (int)(q / LPR), part = (int)(q - (long long)srow * LPR);
  const float mu = mean[ch], is = invstd[ch];
  const float g = gamma != nullptr ? gamma[ch] : 1.f;
  const float be = beta != nullptr ? beta[ch] : 0.f;
  const float c0 = dbeta[ch] * inv_m, c1 = dgamma[ch] * inv_m;
  const size_t o = plane * (size_t)S + srow;
  const float gp = dpool[o];
  const int sel = (int)arg[o] - part * 4;                                    // 0..3 when the selected neighbour is here
  const size_t at = plane * (size_t)S * K + (size_t)q * 4;
  const float4 xv = *reinterpret_cast<const float4 *>(x + at);
  const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
  float out[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float xh = (xs[e] - mu) * is;
    const float gv = (e == sel && xh * g + be > 0.f) ? gp : 0.f;
    out[e] = ((gv - c0) - xh * c1) * (g * is);
  }
  *reinterpret_cast<float4 *>(dx + at) = make_float4(out[0], out[1], out[2], out[3]);
}

static int bn_splits(int c, long long M, long long *per_split) {
  long long want = 2048 / (c > 0 ? c : 1);                       // ~2048 workgroups in the reduction passes
  if (want < 1) want = 1;
  if (want > BN_MAX_SPLITS) want = BN_MAX_SPLITS;
  long long ps = (M + want - 1) / want;
  const long long min_ps = 4ll * BN_THREADS * 4;                 // at least 4 float4 per thread
  if (ps < min_ps) ps = min_ps;
  ps = (ps + 3) / 4 * 4;
  *per_split = ps;
  return (int)((M + ps - 1) / ps);
}

// for conv1x1.hip: the statistics of a convolution's output from the partial sums its epilogue left
void bn_forward_finish_launch(int c, int nsplit, long long M, float eps, float momentum, const double *partial,
                              float *running_mean, float *running_var, float *save_mean, float *save_invstd) {
  hipLaunchKernelGGL(bn_forward_finish_kernel, dim3(ceil_div(c, 4)), dim3(256), 0, current_stream(), c, nsplit, M, eps,
                     momentum, partial, running_mean, running_var, save_mean, save_invstd);
  check_launch("bn_forward_finish");
}

void bn_backward_finish_launch(int c, int nsplit, const double *partial, float *dgamma, float *dbeta) {
  hipLaunchKernelGGL(bn_backward_finish_kernel, dim3(ceil_div(c, 4)), dim3(256), 0, current_stream(), c, nsplit, partial, dgamma,
                     dbeta);
  check_launch("bn_backward_finish");
}

}  // namespace pwclo

using namespace pwclo;

extern "C" long long batchnorm_train_workspace_bytes(int c) {
  return (long long)(c > 0 ? c : 1) * BN_MAX_SPLITS * 2 * (long long)sizeof(double);
}

template <int MODE, bool RELU>
static void bn_launch_apply(bool vec, int b, int c, int l, float inv_m, const float *x, const float *dy,
                            const float *gamma, const float *beta, const float *mean, const float *invstd,
                            const float *dgamma, const float *dbeta, float *out, hipStream_t st) {
  if (vec)
    hipLaunchKernelGGL((bn_apply_kernel<MODE, true, RELU>), dim3(ceil_div(l, 4 * BN_THREADS), c, b), dim3(BN_THREADS),
                       0, st, c, l, inv_m, x, dy, gamma, beta, mean, invstd, dgamma, dbeta, out);
  else
    hipLaunchKernelGGL((bn_apply_kernel<MODE, false, RELU>), dim3(ceil_div(l, BN_THREADS), c, b), dim3(BN_THREADS), 0,
                       st, c, l, inv_m, x, dy, gamma, beta, mean, invstd, dgamma, dbeta, out);
}

extern "C" void batchnorm_train_forward_kernel_wrapper(int b, int c, int l, const float *x, const float *gamma,
                                                       const float *beta, float eps, float momentum,
                                                       float *running_mean, float *running_var, float *y,
                                                       float *save_mean, float *save_invstd, void *workspace,
                                                       int relu) {
  if (b <= 0 || c <= 0 || l <= 0) return;
  PWCLO_REQUIRE(b <= 65535 && c <= 65535, "batchnorm_train_forward: b=%d c=%d exceed the grid limits", b, c);
  PWCLO_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
                "batchnorm_train_forward: running_mean and running_var must be given together%s", "");
  const bool vec = (l % 4 == 0);
  PWCLO_REQUIRE(!vec || ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0),
                "batchnorm_train_forward: x and y must be 16-byte aligned%s", "");
  const bool stats_only = (y == nullptr);     // the next layer's convolution applies the normalisation on its loads
  const long long M = (long long)b * l;
  long long per_split;
  const int nsplit = bn_splits(c, M, &per_split);
  double *partial = reinterpret_cast<double *>(workspace);
  hipStream_t st = current_stream();
  const float *none = nullptr;
  hipLaunchKernelGGL((bn_partial_kernel<0, false>), dim3(nsplit, c), dim3(BN_THREADS), 0, st, c, l, M, per_split, x,
                     none, none, none, none, none, partial);
  hipLaunchKernelGGL(bn_forward_finish_kernel, dim3(ceil_div(c, 4)), dim3(256), 0, st, c, nsplit, M, eps, momentum,
                     partial, running_mean, running_var, save_mean, save_invstd);
  if (stats_only) {
  } else if (relu)
    bn_launch_apply<0, true>(vec, b, c, l, 0.f, x, none, gamma, beta, save_mean, save_invstd, none, none, y, st);
  else
    bn_launch_apply<0, false>(vec, b, c, l, 0.f, x, none, gamma, beta, save_mean, save_invstd, none, none, y, st);
  check_launch("batchnorm_train_forward");
}

extern "C" void batchnorm_train_backward_kernel_wrapper(int b, int c, int l, const float *x, const float *dy,
                                                        const float *gamma, const float *beta,
                                                        const float *save_mean, const float *save_invstd, float *dx,
                                                        float *dgamma, float *dbeta, void *workspace, int relu) {
  if (b <= 0 || c <= 0 || l <= 0) return;
  PWCLO_REQUIRE(b <= 65535 && c <= 65535, "batchnorm_train_backward: b=%d c=%d exceed the grid limits", b, c);
  const bool vec = (l % 4 == 0);
  PWCLO_REQUIRE(!vec || ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(dx) & 15) == 0),
                "batchnorm_train_backward: x, dy and dx must be 16-byte aligned%s", "");
  const long long M = (long long)b * l;
  long long per_split;
  const int nsplit = bn_splits(c, M, &per_split);
  double *partial = reinterpret_cast<double *>(workspace);
  hipStream_t st = current_stream();
  if (relu)
    hipLaunchKernelGGL((bn_partial_kernel<1, true>), dim3(nsplit, c), dim3(BN_THREADS), 0, st, c, l, M, per_split, x, dy,
                       save_mean, save_invstd, gamma, beta, partial);
  else
    hipLaunchKernelGGL((bn_partial_kernel<1, false>), dim3(nsplit, c), dim3(BN_THREADS), 0, st, c, l, M, per_split, x,
                       dy, save_mean, save_invstd, gamma, beta, partial);
  hipLaunchKernelGGL(bn_backward_finish_kernel, dim3(ceil_div(c, 4)), dim3(256), 0, st, c, nsplit, partial, dgamma,
                     dbeta);
  const float inv_m = (float)(1.0 / (double)M);
  if (relu)
    bn_launch_apply<1, true>(vec, b, c, l, inv_m, x, dy, gamma, beta, save_mean, save_invstd, dgamma, dbeta, dx, st);
  else
    bn_launch_apply<1, false>(vec, b, c, l, inv_m, x, dy, gamma, beta, save_mean, save_invstd, dgamma, dbeta, dx, st);
  check_launch("batchnorm_train_backward");
}

// y = [relu]((x - mean) * invstd * gamma + beta) with GIVEN batch statistics (those conv1x1_forward_bnstats_kernel_wrapper
// left): the apply pass of batchnorm_train_forward_kernel_wrapper alone.
extern "C" void batchnorm_train_apply_kernel_wrapper(int b, int c, int l, const float *x, const float *gamma,
                                                     const float *beta, const float *mean, const float *invstd, float *y,
                                                     int relu) {
  if (b <= 0 || c <= 0 || l <= 0) return;
  PWCLO_REQUIRE(b <= 65535 && c <= 65535, "batchnorm_train_apply: b=%d c=%d exceed the grid limits", b, c);
  const bool vec = (l % 4 == 0);
  PWCLO_REQUIRE(!vec || ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0),
                "batchnorm_train_apply: x and y must be 16-byte aligned%s", "");
  const float *none = nullptr;
  hipStream_t st = current_stream();
  if (relu)
    bn_launch_apply<0, true>(vec, b, c, l, 0.f, x, none, gamma, beta, mean, invstd, none, none, y, st);
  else
    bn_launch_apply<0, false>(vec, b, c, l, 0.f, x, none, gamma, beta, mean, invstd, none, none, y, st);
  check_launch("batchnorm_train_apply");
}

// dx = gamma * invstd * (g - dbeta / M - xhat * dgamma / M) with GIVEN dgamma / dbeta (conv1x1_dgrad_bnstats left them): the
// apply pass of batchnorm_train_backward_kernel_wrapper alone.
extern "C" void batchnorm_train_backward_apply_kernel_wrapper(int b, int c, int l, const float *x, const float *dy,
                                                              const float *gamma, const float *beta, const float *save_mean,
                                                              const float *save_invstd, const float *dgamma,
                                                              const float *dbeta, float *dx, int relu) {
  if (b <= 0 || c <= 0 || l <= 0) return;
  PWCLO_REQUIRE(b <= 65535 && c <= 65535, "batchnorm_train_backward_apply: b=%d c=%d exceed the grid limits", b, c);
  const bool vec = (l % 4 == 0);
  PWCLO_REQUIRE(!vec || ((reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
                         (reinterpret_cast<uintptr_t>(dx) & 15) == 0),
                "batchnorm_train_backward_apply: x, dy and dx must be 16-byte aligned%s", "");
  const float inv_m = (float)(1.0 / ((double)b * l));
  hipStream_t st = current_stream();
  if (relu)
    bn_launch_apply<1, true>(vec, b, c, l, inv_m, x, dy, gamma, beta, save_mean, save_invstd, dgamma, dbeta, dx, st);
  else
    bn_launch_apply<1, false>(vec, b, c, l, inv_m, x, dy, gamma, beta, save_mean, save_invstd, dgamma, dbeta, dx, st);
  check_launch("batchnorm_train_backward_apply");
}

// ---- BatchNorm -> ReLU -> max over K (tail of the grouped stacks) ------------------------------------------------
#define PWCLO_BN_MAXK_DISPATCH(k, CALL) \
  switch (k) {                          \
    case 4: CALL(4); break;             \
    case 8: CALL(8); break;             \
    case 16: CALL(16); break;           \
    default: CALL(32); break;           \
  }

extern "C" void batchnorm_train_relu_maxk_forward_kernel_wrapper(int b, int c, int s, int k, const float *x,
                                                                 const float *gamma, const float *beta, float eps,
                                                                 float momentum, float *running_mean,
                                                                 float *running_var, float *pooled,
                                                                 unsigned char *arg, float *xsel, float *save_mean,
                                                                 float *save_invstd, void *workspace) {
  if (b <= 0 || c <= 0 || s <= 0) return;
  PWCLO_REQUIRE(k == 4 || k == 8 || k == 16 || k == 32, "batchnorm_train_relu_maxk_forward: k=%d not in {4,8,16,32}", k);
  PWCLO_REQUIRE(b <= 65535 && c <= 65535, "batchnorm_train_relu_maxk_forward: b=%d c=%d exceed the grid limits", b, c);
  PWCLO_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
                "batchnorm_train_relu_maxk_forward: running_mean and running_var must be given together%s", "");
  PWCLO_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "batchnorm_train_relu_maxk_forward: x must be 16-byte aligned%s", "");
  const int l = s * k;
  const long long M = (long long)b * l;
  long long per_split;
  const int nsplit = bn_splits(c, M, &per_split);
  double *partial = reinterpret_cast<double *>(workspace);
  hipStream_t st = current_stream();
  const float *none = nullptr;
  hipLaunchKernelGGL((bn_partial_kernel<0, false>), dim3(nsplit, c), dim3(BN_THREADS), 0, st, c, l, M, per_split, x,
                     none, none, none, none, none, partial);
  hipLaunchKernelGGL(bn_forward_finish_kernel, dim3(ceil_div(c, 4)), dim3(256), 0, st, c, nsplit, M, eps, momentum,
                     partial, running_mean, running_var, save_mean, save_invstd);
  const dim3 grid(ceil_div(l / 4, BN_THREADS), c, b);
#define PWCLO_CALL(KK)                                                                                              \
  hipLaunchKernelGGL((bn_apply_relu_maxk_kernel<KK>), grid, dim3(BN_THREADS), 0, st, c, s, x, gamma, beta, save_mean, \
                     save_invstd, pooled, arg, xsel)
  PWCLO_BN_MAXK_DISPATCH(k, PWCLO_CALL)
#undef PWCLO_CALL
  check_launch("batchnorm_train_relu_maxk_forward");
}

// The pooled pass alone, with GIVEN batch statistics (see batchnorm_train_apply_kernel_wrapper).
extern "C" void batchnorm_train_relu_maxk_apply_kernel_wrapper(int b, int c, int s, int k, const float *x,
                                                               const float *gamma, const float *beta, const float *mean,
                                                               const float *invstd, float *pooled, unsigned char *arg,
                                                               float *xsel) {
  if (b <= 0 || c <= 0 || s <= 0) return;
  PWCLO_REQUIRE(k == 4 || k == 8 || k == 16 || k == 32, "batchnorm_train_relu_maxk_apply: k=%d not in {4,8,16,32}", k);
  PWCLO_REQUIRE(b <= 65535 && c <= 65535, "batchnorm_train_relu_maxk_apply: b=%d c=%d exceed the grid limits", b, c);
  PWCLO_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "batchnorm_train_relu_maxk_apply: x must be 16-byte aligned%s", "");
  const int l = s * k;
  hipStream_t st = current_stream();
  const dim3 grid(ceil_div(l / 4, BN_THREADS), c, b);
#define PWCLO_CALL(KK)                                                                                              \
  hipLaunchKernelGGL((bn_apply_relu_maxk_kernel<KK>), grid, dim3(BN_THREADS), 0, st, c, s, x, gamma, beta, mean, invstd, \
                     pooled, arg, xsel)
  PWCLO_BN_MAXK_DISPATCH(k, PWCLO_CALL)
#undef PWCLO_CALL
  check_launch("batchnorm_train_relu_maxk_apply");
}

extern "C" void batchnorm_train_relu_maxk_backward_kernel_wrapper(int b, int c, int s, int k, const float *x,
                                                                  const float *dpool, const unsigned char *arg,
                                                                  const float *xsel, const float *gamma,
                                                                  const float *beta, const float *save_mean,
                                                                  const float *save_invstd, float *dx, float *dgamma,
                                                                  float *dbeta, void *workspace) {
  if (b <= 0 || c <= 0 || s <= 0) return;
  PWCLO_REQUIRE(k == 4 || k == 8 || k == 16 || k == 32, "batchnorm_train_relu_maxk_backward: k=%d not in {4,8,16,32}", k);
  PWCLO_REQUIRE(b <= 65535 && c <= 65535, "batchnorm_train_relu_maxk_backward: b=%d c=%d exceed the grid limits", b, c);
  PWCLO_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0,
                "batchnorm_train_relu_maxk_backward: x and dx must be 16-byte aligned%s", "");
  // the reductions run over the selected elements only: (xsel, dpool) as a (b, c, s) problem of the dense kernel --
  // every other element has dy = 0 and adds nothing to sum(dy) or sum(dy * xhat)
  const long long Ms = (long long)b * s;
  long long per_split;
  const int nsplit = bn_splits(c, Ms, &per_split);
  double *partial = reinterpret_cast<double *>(workspace);
  hipStream_t st = current_stream();
  hipLaunchKernelGGL((bn_partial_kernel<1, true>), dim3(nsplit, c), dim3(BN_THREADS), 0, st, c, s, Ms, per_split, xsel,
                     dpool, save_mean, save_invstd, gamma, beta, partial);
  hipLaunchKernelGGL(bn_backward_finish_kernel, dim3(ceil_div(c, 4)), dim3(256), 0, st, c, nsplit, partial, dgamma,
                     dbeta);
  const float inv_m = (float)(1.0 / ((double)b * s * k));
  const dim3 grid(ceil_div(s * k / 4, BN_THREADS), c, b);
#define PWCLO_CALL(KK)                                                                                               \
  hipLaunchKernelGGL((bn_apply_bwd_maxk_kernel<KK>), grid, dim3(BN_THREADS), 0, st, c, s, inv_m, x, dpool, arg, gamma, \
                     beta, save_mean, save_invstd, dgamma, dbeta, dx)
  PWCLO_BN_MAXK_DISPATCH(k, PWCLO_CALL)
#undef PWCLO_CALL
  check_launch("batchnorm_train_relu_maxk_backward");
}
