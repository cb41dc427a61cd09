// Hoisted variants of the fused layers (gfx950): the per-point parts of every first layer are
// precomputed once per point by linear_jobs_kernel; the pixel kernels gather those partial
// products as accumulator seeds and run only the 16-channel geometry block of layer 1 (plus the
// deeper layers) on the matrix cores.  See mlp_core.hpp "Hoisting".
//
//   set abstraction   [diff | feat[n]]              -> pre[n] = W1_feat feat[n] + b1
//   set-upconv        [feat1[n] | diff]             -> pre[n] = W1_feat feat1[n] + b1
//   cost volume a1    [geo | feat1[s] | feat2[n]]   -> u[s] = W1_p feat1[s] + b1, v[n] = W1_q feat2[n]
//   cost volume b     [enc2 | feat1[s] | first[n]]  -> u2[s] = W_p feat1[s] + b,  v2[n] = W_f first[n]
// MACs per pixel: set-upconv 18 432 -> 10 240, cv_a1 (C=64) 30 208 -> 14 592, cv_b (C=64) 33 408 -> 17 024.
#include <math.h>
#include <stdlib.h>

#include "mlp_core.hpp"

PWCLO_TRACE_TU(fused_hoisted)

namespace pwclo {

static int fh_tuning(const char *name, int dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}

// Geometry inputs are laid out "k-step major": logical channel c sits at lane group c % 4, component
// c / 4, i.e. MFMA k-step c / 4 (a 16x16x4 instruction consumes component r of all four lane
// groups), so n real channels need only ceil(n / 4) of the block's four k-steps (KS below;
// fused.py: kstep_major_map packs the weights to match).
// cost volume: [p(3), q(3), q-p(3), |q-p|] (costvolume.py:92-105) -> 3 k-steps.
__device__ __forceinline__ f32x4 geometry_block_h(const float *p, const float *q, int g) {
  const float px = p[0], py = p[1], pz = p[2], qx = q[0], qy = q[1], qz = q[2];
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  const float euc = sqrtf(((dx * dx + dy * dy) + dz * dz) + 1e-20f);
  f32x4 v = {px, qy, dz, 0.f};                   // channels 0, 4, 8
  if (g == 1) v = f32x4{py, qz, euc, 0.f};       // 1, 5, 9
  if (g == 2) v = f32x4{pz, dx, 0.f, 0.f};       // 2, 6
  if (g == 3) v = f32x4{qx, dy, 0.f, 0.f};       // 3, 7
  return v;
}
// [diff(3)] (one k-step) or, at level 0, [diff(3), neighbour xyz(3)] (two k-steps).
__device__ __forceinline__ f32x4 diff_block_h(float dx, float dy, float dz, float qx, float qy, float qz,
                                              bool with_q, int g) {
  f32x4 v = {dx, with_q ? qy : 0.f, 0.f, 0.f};
  if (g == 1) v = f32x4{dy, with_q ? qz : 0.f, 0.f, 0.f};
  if (g == 2) v = f32x4{dz, 0.f, 0.f, 0.f};
  if (g == 3) v = f32x4{with_q ? qx : 0.f, 0.f, 0.f, 0.f};
  return v;
}

__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }

#define PWCLO_H_TILE_LOOP(KP_, P_, S_, B_)                                                          \
  constexpr int TILE = 16 * (P_);                                                                   \
  const int pix_per_cloud = (S_) * (KP_);                                                           \
  const int tiles_per_cloud = (pix_per_cloud + TILE - 1) / TILE;                                    \
  const int ntiles = (B_) * tiles_per_cloud;                                                        \
  for (int t = blockIdx.x * W + wave_index(); t < ntiles; t += gridDim.x * W)

// ---- per-point linear maps (the hoisted partial products) -----------------------------------------
struct LinJob {
  const float *src;   // (npts, 16*nbi) point-major
  const float *w;     // packed single layer (bias included, no activation)
  float *out;         // (npts, 16*nbo); bf16 rows when out_bf16
  int npts, nbi, nbo;
  int out_bf16;
};
constexpr int LIN_MAX_JOBS = 8;
struct LinArgs { LinJob job[LIN_MAX_JOBS]; };
constexpr int LIN_WAVES = 8;

template <int NBI, int NBO>
__device__ __forceinline__ void linear_tiles(const LinJob &jb, const float *lds_w, int lane) {
  constexpr int P = 2;
  const int g = lane >> 4, j = lane & 15;
  const int ntiles = (jb.npts + 16 * P - 1) / (16 * P);
  for (int t = blockIdx.x * LIN_WAVES + wave_index(); t < ntiles; t += gridDim.x * LIN_WAVES) {
    f32x4 in[NBI][P];
    int pt[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int q = t * 16 * P + 16 * p + j;
      pt[p] = q < jb.npts ? q : -1;
      const float *row = at32(jb.src, (unsigned)(q < jb.npts ? q : jb.npts - 1) * (unsigned)(64 * NBI) + 16u * (unsigned)g);
#pragma unroll
      for (int m = 0; m < NBI; ++m) in[m][p] = ld4(row + 16 * m);
    }
    f32x4 o1[NBO][P];
    mlp_layer<NBI, NBO, P, false>(o1, in, lds_w, lane);
    if (jb.out_bf16) {        // job-uniform
#pragma unroll
      for (int o = 0; o < NBO; ++o)
#pragma unroll
        for (int p = 0; p < P; ++p)
          if (pt[p] >= 0) st_group<true>(jb.out, (unsigned)pt[p], 16u * NBO, o, g, o1[o][p]);
    } else {
#pragma unroll
      for (int o = 0; o < NBO; ++o)
#pragma unroll
        for (int p = 0; p < P; ++p)
          if (pt[p] >= 0) st_group<false>(jb.out, (unsigned)pt[p], 16u * NBO, o, g, o1[o][p]);
    }
  }
}

__global__ __launch_bounds__(LIN_WAVES * 64) void linear_jobs_kernel(LinArgs a) {
  TraceScope trace_scope_(TK_LINEAR);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  const LinJob jb = a.job[blockIdx.y];
  stage_weights(lds_w, jb.w, layer_floats(jb.nbi, jb.nbo));
  __syncthreads();
  const int lane = threadIdx.x & 63;
#define LJ(I, O) if (jb.nbi == I && jb.nbo == O) { linear_tiles<I, O>(jb, lds_w, lane); return; }
  LJ(1, 1) LJ(1, 2) LJ(1, 4) LJ(1, 8) LJ(2, 1) LJ(2, 2) LJ(2, 4) LJ(2, 8) LJ(4, 1) LJ(4, 2) LJ(4, 4) LJ(4, 8)
#undef LJ
}

// ---- set abstraction, hoisted ------------------------------------------------------------------------
struct SAHArgs {
  const float *xyz, *new_xyz;   // (B,N,3), (B,S,3)
  const float *pre;             // (B,N,16*B1) = W1_feat feat + b1, or nullptr (level 0: plain bias)
  const int *idx;               // (B,S,K)
  const float *w;               // packed: layer 1 on the geometry block, layers 2, 3
  float *out;                   // (B,S,16*B3)
  int B, N, S, K;
};

// KMAJ (level 0 only): layers 1 and 2 produce their <= 8 real channels "k-step major" (fused.py: pack_layer
// kmajor_out), so layers 2 and 3 run only the two k-steps that carry data.
template <int B1, int B2, int B3, int KP, int P, int W, bool XYZ_ONLY, int FMT = 0, bool KMAJ = false>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void sa_h_kernel(SAHArgs a) {
  TraceScope trace_scope_(TK_SA_H);
  constexpr int W1 = layer_floats(1, B1), W2 = layer_floats_any<FMT>(B1, B2), W3 = layer_floats_any<FMT>(B2, B3);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, W1 + W2 + W3);
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  constexpr int C1 = 16 * B1, C3 = 16 * B3;
  PWCLO_H_TILE_LOOP(KP, P, a.S, a.B) {
    const int b = t / tiles_per_cloud;
    const int pix0 = (t - b * tiles_per_cloud) * TILE;
    const unsigned bS = (unsigned)b * (unsigned)a.S, bN = (unsigned)b * (unsigned)a.N;   // scalar
    f32x4 in[1][P];
    int sq[P];
    unsigned psrc[P];
    constexpr bool H16 = FMT == 2;                 // hoisted rows stored as bf16 (mlp_core.hpp)
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const PixelMap<KP> pm(pix0 + 16 * p + j);
      const bool valid = pm.s < a.S;
      const int s = valid ? pm.s : a.S - 1;
      const int k = pm.k < a.K ? pm.k : 0;
      sq[p] = valid ? s : -1;
      const unsigned row = bS + (unsigned)s;
      const int nbr = *at32(a.idx, (mul24(row, (unsigned)a.K) + (unsigned)k) * 4u);
      const float *c = at32(a.new_xyz, mul24(row, 12u));
      const unsigned src = bN + (unsigned)nbr;
      const float *q = at32(a.xyz, mul24(src, 12u));
      const float qx = q[0], qy = q[1], qz = q[2];
      in[0][p] = diff_block_h(qx - c[0], qy - c[1], qz - c[2], qx, qy, qz, XYZ_ONLY, g);
      psrc[p] = src;
    }
    f32x4 h1[B1][P], h2[B2][P], h3[B3][P];
    if (XYZ_ONLY) {
      mlp_layer<1, B1, P, true, 2>(h1, in, lds_w, lane);          // 6 real inputs
    } else {
      // all seed rows are requested before the first MFMA: one exposed memory latency per tile
#pragma unroll
      for (int o = 0; o < B1; ++o)
#pragma unroll
        for (int p = 0; p < P; ++p) h1[o][p] = ld_group<H16>(a.pre, psrc[p], (unsigned)C1, o, g);
      mlp_layer_init<1, B1, P, true, 1>(h1, in, lds_w, lane, [&](int o, int p) { return h1[o][p]; });   // diff(3)
    }
    if constexpr (KMAJ) {
      mlp_layer<B1, B2, P, true, 2>(h2, h1, lds_w + W1, lane);
      mlp_layer<B2, B3, P, false, 2>(h3, h2, lds_w + W1 + W2, lane);
    } else {
      mlp_layer_any<FMT, B1, B2, P, true>(h2, h1, lds_w + W1, lane);
      mlp_layer_any<FMT, B2, B3, P, false>(h3, h2, lds_w + W1 + W2, lane);   // its ReLU is applied after the pool
    }
    constexpr int GROUP = KP < 16 ? KP : 16;
    constexpr int BPQ = KP > 16 ? KP / 16 : 1;
#pragma unroll
    for (int o = 0; o < B3; ++o) {
#pragma unroll
      for (int p = 0; p < P; p += BPQ) {
        f32x4 v = h3[o][p];
#pragma unroll
        for (int e = 1; e < BPQ; ++e) {
          const f32x4 u = h3[o][p + e];
          v.x = max_bits(v.x, u.x); v.y = max_bits(v.y, u.y); v.z = max_bits(v.z, u.z); v.w = max_bits(v.w, u.w);
        }
        v.x = relu_bits(group_max_nonneg<GROUP>(v.x)); v.y = relu_bits(group_max_nonneg<GROUP>(v.y));
        v.z = relu_bits(group_max_nonneg<GROUP>(v.z)); v.w = relu_bits(group_max_nonneg<GROUP>(v.w));
        if ((j & (GROUP - 1)) == 0 && sq[p] >= 0)
          *reinterpret_cast<f32x4 *>(at32(a.out, (bS + (unsigned)sq[p]) * (unsigned)(C3 * 4) + 64u * o +
                                                      16u * (unsigned)g)) = v;
      }
    }
  }
}

// ---- set-upconv, hoisted --------------------------------------------------------------------------------
struct UpHArgs {
  const float *xyz2, *xyz1;   // (B,S,3) fine queries, (B,N,3) coarse points
  const float *pre;           // (B,N,128) = W1_feat feat1 + b1
  const int *idx;             // (B,S,K)
  const float *w;             // packed: layer 1 on the diff block (-> 128), layer 2 (128 -> 64)
  float *out;                 // (B,S,64)
  int B, N, S, K;
};

template <int KP, int P, int W, int FMT = 0>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void upconv_h_kernel(UpHArgs a) {
  TraceScope trace_scope_(TK_UPCONV_H);
  constexpr int B1 = 8, B2 = 4;
  constexpr int W1 = layer_floats(1, B1), W2 = layer_floats_any<FMT>(B1, B2);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, W1 + W2);
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  PWCLO_H_TILE_LOOP(KP, P, a.S, a.B) {
    const int b = t / tiles_per_cloud;
    const int pix0 = (t - b * tiles_per_cloud) * TILE;
    const unsigned bS = (unsigned)b * (unsigned)a.S, bN = (unsigned)b * (unsigned)a.N;   // scalar
    f32x4 in[1][P];
    int sq[P];
    unsigned psrc[P];
    constexpr bool H16 = FMT == 2;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const PixelMap<KP> pm(pix0 + 16 * p + j);
      const bool valid = pm.s < a.S;
      const int s = valid ? pm.s : a.S - 1;
      const int k = pm.k < a.K ? pm.k : 0;
      sq[p] = valid ? s : -1;
      const unsigned row = bS + (unsigned)s;
      const int nbr = *at32(a.idx, (mul24(row, (unsigned)a.K) + (unsigned)k) * 4u);
      const float *c = at32(a.xyz2, mul24(row, 12u));
      const unsigned src = bN + (unsigned)nbr;
      const float *q = at32(a.xyz1, mul24(src, 12u));
      in[0][p] = diff_block_h(q[0] - c[0], q[1] - c[1], q[2] - c[2], 0.f, 0.f, 0.f, false, g);
      psrc[p] = src;                                                  // 128 channels per row
    }
    f32x4 h1[B1][P], h2[B2][P];
#pragma unroll
    for (int o = 0; o < B1; ++o)
#pragma unroll
      for (int p = 0; p < P; ++p) h1[o][p] = ld_group<H16>(a.pre, psrc[p], 128u, o, g);
    mlp_layer_init<1, B1, P, true, 1>(h1, in, lds_w, lane, [&](int o, int p) { return h1[o][p]; });   // diff(3)
    mlp_layer_any<FMT, B1, B2, P, false>(h2, h1, lds_w + W1, lane);   // its ReLU is applied after the pool
    constexpr int GROUP = KP < 16 ? KP : 16;
#pragma unroll
    for (int o = 0; o < B2; ++o)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        f32x4 v = h2[o][p];
        v.x = relu_bits(group_max_nonneg<GROUP>(v.x)); v.y = relu_bits(group_max_nonneg<GROUP>(v.y));
        v.z = relu_bits(group_max_nonneg<GROUP>(v.z)); v.w = relu_bits(group_max_nonneg<GROUP>(v.w));
        if ((j & (GROUP - 1)) == 0 && sq[p] >= 0)
          *reinterpret_cast<f32x4 *>(at32(a.out, ((bS + (unsigned)sq[p]) << 8) + 64u * o + 16u * (unsigned)g)) = v;
      }
  }
}

// Set-upconv with the max over the K neighbours IN-LANE (cf. cv_a2_lane6_kernel): a wave tile = 16 consecutive
// queries, pass k runs neighbour k of those queries as one 16-pixel block (lane j <-> query j), the pooled value is a
// running maximum in registers -- no DPP reduction, and every lane stores its own query's row.
template <int W>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void upconv_lane_kernel(UpHArgs a) {
  TraceScope trace_scope_(TK_UPCONV_LANE);
  constexpr int B1 = 8, B2 = 4;
  constexpr int W1 = layer_floats(1, B1);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, W1 + layer_floats(B1, B2));
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int tiles_per_cloud = (a.S + 15) / 16;
  const int ntiles = a.B * tiles_per_cloud;
  for (int t = blockIdx.x * W + wave_index(); t < ntiles; t += gridDim.x * W) {
    const int b = t / tiles_per_cloud;
    const int q = (t - b * tiles_per_cloud) * 16 + j;
    const bool valid = q < a.S;
    const unsigned bN = (unsigned)b * (unsigned)a.N;
    const unsigned row = (unsigned)b * (unsigned)a.S + (unsigned)(valid ? q : a.S - 1);
    const float *c = at32(a.xyz2, mul24(row, 12u));
    const float cx = c[0], cy = c[1], cz = c[2];
    const unsigned slot0 = mul24(row, (unsigned)a.K);
    f32x4 mx[B2];
    auto run_pass = [&](auto first_tag, int k) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const int nbr = *at32(a.idx, (slot0 + (unsigned)k) * 4u);
      const unsigned src = bN + (unsigned)nbr;
      const float *qp = at32(a.xyz1, mul24(src, 12u));
      f32x4 in[1][1], h1[B1][1], h2[B2][1];
      in[0][0] = diff_block_h(qp[0] - cx, qp[1] - cy, qp[2] - cz, 0.f, 0.f, 0.f, false, g);
      const float *prow = at32(a.pre, (src << 9) + 16u * (unsigned)g);
#pragma unroll
      for (int o = 0; o < B1; ++o) h1[o][0] = ld4(prow + 16 * o);
      mlp_layer_init<1, B1, 1, true, 1>(h1, in, lds_w, lane, [&](int o, int) { return h1[o][0]; });
      mlp_layer<B1, B2, 1, false>(h2, h1, lds_w + W1, lane);       // ReLU after the pool
#pragma unroll
      for (int o = 0; o < B2; ++o) {
        if (FIRST) {
          mx[o] = h2[o][0];
        } else {
          mx[o].x = max_bits(mx[o].x, h2[o][0].x); mx[o].y = max_bits(mx[o].y, h2[o][0].y);
          mx[o].z = max_bits(mx[o].z, h2[o][0].z); mx[o].w = max_bits(mx[o].w, h2[o][0].w);
        }
      }
    };
    run_pass(std::true_type{}, 0);
#pragma unroll 1
    for (int k = 1; k < a.K; ++k) run_pass(std::false_type{}, k);
    if (valid) {
#pragma unroll
      for (int o = 0; o < B2; ++o) {
        f32x4 v = mx[o];
        v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
        *reinterpret_cast<f32x4 *>(at32(a.out, (row << 8) + 64u * o + 16u * (unsigned)g)) = v;
      }
    }
  }
}

// Set-upconv of a refinement level as ONE launch (round 3): the in-lane kernel above for up to TWO jobs that share
// queries, coarse points and the neighbour lists (the features and the mask branch of PW/pose_warp_refinement.py:95-103:
// same xyz, same knn, same fine features, different coarse rows and weights; blockIdx.y = job), with the module's
// post-MLP (P2/pointnet2_modules.py:508-515: cat(pooled 64, fine features C) -> 64, ReLU) applied to the pooled rows
// while they are still in registers -- lane j holds query j's pooled channels in the accumulator layout, which IS the
// next layer's B operand.  Replaces 2 x (upconv + pointwise) launches; values are bit-identical to those (same layer
// routines, same k order; the max over K is exact in any order).
struct UpPostArgs {
  const float *xyz2, *xyz1;   // (B,S,3) fine queries, (B,N,3) coarse points
  const int *idx;             // (B,S,K)
  const float *feat2;         // (B,S,16*NB2) fine features, second source of the post-MLP
  const float *pre[2];        // (B,N,128) = W1_feat feat1 + b1 of each job
  const float *w[2];          // packed: layer 1 on the diff block (-> 128), layer 2 (128 -> 64)
  const float *wpost[2];      // packed post layer (64 + 16*NB2 -> 64)
  float *out[2];              // (B,S,64)
  int B, N, S, K;
};

template <int NB2, int W>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void upconv_lane_post_kernel(UpPostArgs a) {
  TraceScope trace_scope_(TK_UPCONV_LANE);
  constexpr int B1 = 8, B2 = 4;
  constexpr int W1 = layer_floats(1, B1), W2 = layer_floats(B1, B2), WP = layer_floats(B2 + NB2, 4);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  const int job = blockIdx.y;
  stage_weights(lds_w, a.w[job], W1 + W2);
  stage_weights(lds_w + W1 + W2, a.wpost[job], WP);
  __syncthreads();
  const float *pre = a.pre[job];
  float *out = a.out[job];
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int tiles_per_cloud = (a.S + 15) / 16;
  const int ntiles = a.B * tiles_per_cloud;
  for (int t = blockIdx.x * W + wave_index(); t < ntiles; t += gridDim.x * W) {
    const int b = t / tiles_per_cloud;
    const int q = (t - b * tiles_per_cloud) * 16 + j;
    const bool valid = q < a.S;
    const unsigned bN = (unsigned)b * (unsigned)a.N;
    const unsigned row = (unsigned)b * (unsigned)a.S + (unsigned)(valid ? q : a.S - 1);
    const float *c = at32(a.xyz2, mul24(row, 12u));
    const float cx = c[0], cy = c[1], cz = c[2];
    const unsigned slot0 = mul24(row, (unsigned)a.K);
    f32x4 cat[B2 + NB2][1];
    auto run_pass = [&](auto first_tag, int k) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const int nbr = *at32(a.idx, (slot0 + (unsigned)k) * 4u);
      const unsigned src = bN + (unsigned)nbr;
      const float *qp = at32(a.xyz1, mul24(src, 12u));
      f32x4 in[1][1], h1[B1][1], h2[B2][1];
      in[0][0] = diff_block_h(qp[0] - cx, qp[1] - cy, qp[2] - cz, 0.f, 0.f, 0.f, false, g);
      const float *prow = at32(pre, (src << 9) + 16u * (unsigned)g);
#pragma unroll
      for (int o = 0; o < B1; ++o) h1[o][0] = ld4(prow + 16 * o);
      mlp_layer_init<1, B1, 1, true, 1>(h1, in, lds_w, lane, [&](int o, int) { return h1[o][0]; });
      mlp_layer<B1, B2, 1, false>(h2, h1, lds_w + W1, lane);       // ReLU after the pool
#pragma unroll
      for (int o = 0; o < B2; ++o) {
        if (FIRST) {
          cat[o][0] = h2[o][0];
        } else {
          cat[o][0].x = max_bits(cat[o][0].x, h2[o][0].x); cat[o][0].y = max_bits(cat[o][0].y, h2[o][0].y);
          cat[o][0].z = max_bits(cat[o][0].z, h2[o][0].z); cat[o][0].w = max_bits(cat[o][0].w, h2[o][0].w);
        }
      }
    };
    run_pass(std::true_type{}, 0);
#pragma unroll 1
    for (int k = 1; k < a.K; ++k) run_pass(std::false_type{}, k);
    // post-MLP on [pooled (64) | fine features (16*NB2)] of the wave's 16 queries
    const float *frow = at32(a.feat2, row * (unsigned)(64 * NB2) + 16u * (unsigned)g);
#pragma unroll
    for (int m = 0; m < NB2; ++m) cat[B2 + m][0] = ld4(frow + 16 * m);
#pragma unroll
    for (int o = 0; o < B2; ++o) {
      f32x4 v = cat[o][0];
      v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
      cat[o][0] = v;
    }
    f32x4 res[4][1];
    mlp_layer<B2 + NB2, 4, 1, true>(res, cat, lds_w + W1 + W2, lane);
    if (valid) {
#pragma unroll
      for (int o = 0; o < 4; ++o)
        *reinterpret_cast<f32x4 *>(at32(out, (row << 8) + 64u * o + 16u * (unsigned)g)) = res[o][0];
    }
  }
}

// ---- cost volume a1 / b, hoisted ------------------------------------------------------------------------
struct CVHArgs {
  const float *xyz1;    // (B,S,3) queries
  const float *u;       // (B,S,128) centre partial product (bias included)
  const float *xyz2;    // (B,N,3) candidates
  const float *v;       // (B,N,128) neighbour partial product
  const float *val;     // b only: (B,N,64) first-aggregate result (the softmax-weighted values)
  const int *idx;       // (B,S,K)
  const float *w;       // packed weights
  float *out;           // a1: pix (B,S*KP,64); b: (B,S,64)
  int B, N, S, K;
};

template <int KP, int P, int W, int FMT = 0>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void cv_a1_h_kernel(CVHArgs a) {
  TraceScope trace_scope_(TK_CV_A1_H);
  constexpr int B1 = 8, B2 = 4, B3 = 4;
  constexpr int W1 = layer_floats(1, B1), W2 = layer_floats_any<FMT>(B1, B2), W3 = layer_floats_any<FMT>(B2, B3);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, W1 + W2 + W3);
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  PWCLO_H_TILE_LOOP(KP, P, a.S, a.B) {
    const int b = t / tiles_per_cloud;
    const int pix0 = (t - b * tiles_per_cloud) * TILE;
    const unsigned bS = (unsigned)b * (unsigned)a.S, bN = (unsigned)b * (unsigned)a.N;   // scalar
    const unsigned bPix = (unsigned)b * (unsigned)pix_per_cloud;
    f32x4 in[1][P];
    int pixv[P];
    unsigned urow[P], vrow[P];
    constexpr bool H16 = FMT == 2;                 // u / v rows in, per-pixel features out: bf16
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int pix = pix0 + 16 * p + j;
      const PixelMap<KP> pm(pix);
      const bool valid = pm.s < a.S;
      const int s = valid ? pm.s : a.S - 1;
      const int k = pm.k < a.K ? pm.k : 0;
      pixv[p] = (valid && pm.k < a.K) ? pix : -1;      // padded slots are never read back (cv_a2)
      const unsigned row = bS + (unsigned)s;
      const int nbr = *at32(a.idx, (mul24(row, (unsigned)a.K) + (unsigned)k) * 4u);
      const unsigned src = bN + (unsigned)nbr;
      in[0][p] = geometry_block_h(at32(a.xyz1, mul24(row, 12u)), at32(a.xyz2, mul24(src, 12u)), g);
      urow[p] = row;
      vrow[p] = src;
    }
    f32x4 h1[B1][P], h2[B2][P], h3[B3][P];
#pragma unroll
    for (int o = 0; o < B1; ++o)
#pragma unroll
      for (int p = 0; p < P; ++p)
        h1[o][p] = ld_group<H16>(a.u, urow[p], 128u, o, g) + ld_group<H16>(a.v, vrow[p], 128u, o, g);
    mlp_layer_init<1, B1, P, true, 3>(h1, in, lds_w, lane, [&](int o, int p) { return h1[o][p]; });   // geometry(10)
    mlp_layer_any<FMT, B1, B2, P, true>(h2, h1, lds_w + W1, lane);
    mlp_layer_any<FMT, B2, B3, P, true>(h3, h2, lds_w + W1 + W2, lane);
#pragma unroll
    for (int o = 0; o < B3; ++o)
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (pixv[p] >= 0) st_group<H16>(a.out, bPix + (unsigned)pixv[p], 64u, o, g, h3[o][p]);
  }
}

template <int KP, int P, int W, int FMT = 0>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void cv_b_h_kernel(CVHArgs a) {
  TraceScope trace_scope_(TK_CV_B_H);
  constexpr int WX = layer_floats(1, 4), W1 = layer_floats_any<FMT>(4, 8), W2 = layer_floats_any<FMT>(8, 4);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, WX + W1 + W2);
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  PWCLO_H_TILE_LOOP(KP, P, a.S, a.B) {
    const int b = t / tiles_per_cloud;
    const int pix0 = (t - b * tiles_per_cloud) * TILE;
    const unsigned bS = (unsigned)b * (unsigned)a.S, bN = (unsigned)b * (unsigned)a.N;   // scalar
    f32x4 geo[1][P], val[4][P];
    int sq[P];
    bool padded[P];
    unsigned urow[P], vrow[P];
    constexpr bool H16 = FMT == 2;                 // u2 / v2 rows: bf16 (the gathered values `first` stay fp32)
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const PixelMap<KP> pm(pix0 + 16 * p + j);
      const bool valid = pm.s < a.S;
      const int s = valid ? pm.s : a.S - 1;
      padded[p] = pm.k >= a.K;
      const int k = padded[p] ? 0 : pm.k;
      sq[p] = valid ? s : -1;
      const unsigned row = bS + (unsigned)s;
      const int nbr = *at32(a.idx, (mul24(row, (unsigned)a.K) + (unsigned)k) * 4u);
      const unsigned src = bN + (unsigned)nbr;
      geo[0][p] = geometry_block_h(at32(a.xyz1, mul24(row, 12u)), at32(a.xyz2, mul24(src, 12u)), g);
      urow[p] = row;
      vrow[p] = src;
      const float *fr = at32(a.val, (src << 8) + 16u * (unsigned)g);
#pragma unroll
      for (int m = 0; m < 4; ++m) val[m][p] = ld4(fr + 16 * m);
    }
    f32x4 enc[4][P], h1[8][P], h2[4][P];
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int p = 0; p < P; ++p)
        h1[o][p] = ld_group<H16>(a.u, urow[p], 128u, o, g) + ld_group<H16>(a.v, vrow[p], 128u, o, g);
    mlp_layer<1, 4, P, true, 3>(enc, geo, lds_w, lane);
    mlp_layer_any_init<FMT, 4, 8, P, true>(h1, enc, lds_w + WX, lane, [&](int o, int p) { return h1[o][p]; });
    mlp_layer_any<FMT, 8, 4, P, true>(h2, h1, lds_w + WX + W1, lane);
    // softmax over the neighbours, weighted sum of the gathered first-aggregate rows
    constexpr int GROUP = KP < 16 ? KP : 16;
    const float NEG_INF = __int_as_float(0xff800000);
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        f32x4 res;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float xv = padded[p] ? NEG_INF : h2[o][p][c];
          const float mx = group_max_nonneg<GROUP>(xv);
          const float ex = padded[p] ? 0.f : exp_nonpos(xv - mx);
          const float den = group_sum<GROUP>(ex);
          const float num = group_sum<GROUP>(ex * val[o][p][c]);
          res[c] = div_ge1(num, den);
        }
        if ((j & (GROUP - 1)) == 0 && sq[p] >= 0)
          *reinterpret_cast<f32x4 *>(at32(a.out, ((bS + (unsigned)sq[p]) << 8) + 64u * o + 16u * (unsigned)g)) = res;
      }
  }
}

// ---- first aggregate of the cost volume for K = 6 as ONE kernel (round 3) ----------------------------------------------
// cv_a1_h (per-pixel feature stack [geo | u[s] + v[n]] -> 128 -> 64 -> 64) used to write its 64-channel result per
// pixel, (B,S,6,64), for cv_a2_lane6 to read back and run [enc(geo) | feat] -> 128 -> 64 -> softmax over the six
// neighbours.  Both stages work on the SAME pixel, so here a wave tile = 16 consecutive queries (lane j <-> query j),
// pass `grp` runs neighbours 2*grp, 2*grp+1 of those queries through BOTH stacks in registers and merges the softmax with
// the running-maximum form (cv_a2_lane6_kernel): no per-pixel buffer (level 1: 100 MB written + 100 MB read per call), one
// launch instead of two, the centre rows u[s] read once per pass instead of once per pixel.  All 161.8 KB of both stages'
// packed weights are LDS resident (the device's limit is 160 KiB = 163.8 KB).  Every layer runs the same routine on the
// same operands in the same k order as the two kernels it replaces: results are bit-identical to them.
// V2: the partial product v2 = W_f . first that cv_b gathers per neighbour (linear_jobs' job_v2) is formed in the epilogue
// while `first` is in registers -- lane j holds query j's 64 channels in the accumulator layout, i.e. as a B operand; its
// 32 weight tiles do not fit LDS any more and are read from global memory (L2 resident: 33 KB shared by every wave).
struct CVLaneArgs {
  const float *xyz1;    // (B,S,3) queries (warped frame-1 points)
  const float *u;       // (B,S,128) centre partial product of a1's first layer (bias included)
  const float *xyz2;    // (B,N,3) candidates
  const float *v;       // (B,N,128) neighbour partial product
  const int *idx;       // (B,S,6)
  const float *w_a1;    // packed: geometry block (-> 128), 128 -> 64, 64 -> 64
  const float *w_a2;    // packed: mlp_conv_xyz_1 (geo -> 64), 128 -> 128, 128 -> 64
  const float *w_v2;    // packed single layer 64 -> 128 (no activation), or nullptr
  float *first;         // (B,S,64)
  float *v2;            // (B,S,128) when w_v2
  int B, N, S;
};

__device__ __forceinline__ f32x4 geometry_block_std(const float *p, const float *q, int g) {   // fused_layers.hip: geometry_block
  const float px = p[0], py = p[1], pz = p[2], qx = q[0], qy = q[1], qz = q[2];
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  const float euc = sqrtf(((dx * dx + dy * dy) + dz * dz) + 1e-20f);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (g == 0) v = f32x4{px, py, pz, qx};
  if (g == 1) v = f32x4{qy, qz, dx, dy};
  if (g == 2) v = f32x4{dz, euc, 0.f, 0.f};
  return v;
}

template <int W, bool V2>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void cv_a_lane6_kernel(CVLaneArgs a) {
  TraceScope trace_scope_(TK_CV_A2_LANE6);
  constexpr int P = 2, PASSES = 3;
  constexpr int A1 = layer_floats(1, 8), A2 = layer_floats(8, 4), A3 = layer_floats(4, 4);
  constexpr int WX = layer_floats(1, 4), W1 = layer_floats(8, 8), W2 = layer_floats(8, 4);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  float *lds_a2 = lds_w + A1 + A2 + A3;
  stage_weights(lds_w, a.w_a1, A1 + A2 + A3);
  stage_weights(lds_a2, a.w_a2, WX + W1 + W2);
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int tiles_per_cloud = (a.S + 15) / 16;
  const int ntiles = a.B * tiles_per_cloud;
  for (int t = blockIdx.x * W + wave_index(); t < ntiles; t += gridDim.x * W) {
    const int b = t / tiles_per_cloud;
    const int q = (t - b * tiles_per_cloud) * 16 + j;
    const bool valid = q < a.S;
    const unsigned row = (unsigned)b * (unsigned)a.S + (unsigned)(valid ? q : a.S - 1);
    const unsigned bN = (unsigned)b * (unsigned)a.N;
    const float *centre = at32(a.xyz1, mul24(row, 12u));
    f32x4 mrun[4], drun[4], nrun[4];
    auto run_pass = [&](auto first_tag, int pass) {
      constexpr bool FIRST = decltype(first_tag)::value;
      f32x4 geo_h[1][P], geo[1][P], h1[8][P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const unsigned slot = mul24(row, 6u) + (unsigned)(P * pass + p);
        const int nbr = *at32(a.idx, slot * 4u);
        const unsigned src = bN + (unsigned)nbr;
        const float *qp = at32(a.xyz2, mul24(src, 12u));
        geo_h[0][p] = geometry_block_h(centre, qp, g);
        geo[0][p] = geometry_block_std(centre, qp, g);
#pragma unroll
        for (int o = 0; o < 8; ++o) h1[o][p] = ld_group<false>(a.v, src, 128u, o, g);
      }
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        const f32x4 uc = ld_group<false>(a.u, row, 128u, o, g);
#pragma unroll
        for (int p = 0; p < P; ++p) h1[o][p] = uc + h1[o][p];
      }
      f32x4 h2[4][P], cat[8][P];
      {   // stage a1: the per-pixel feature (what cv_a1_h stored)
        mlp_layer_init<1, 8, P, true, 3>(h1, geo_h, lds_w, lane, [&](int o, int p) { return h1[o][p]; });
        mlp_layer<8, 4, P, true>(h2, h1, lds_w + A1, lane);
        f32x4 h3[4][P];
        mlp_layer<4, 4, P, true>(h3, h2, lds_w + A1 + A2, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int p = 0; p < P; ++p) cat[4 + m][p] = h3[m][p];
      }
      {   // stage a2: position encoding, attention logits
        f32x4 enc[4][P];
        mlp_layer<1, 4, P, true>(enc, geo, lds_a2, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int p = 0; p < P; ++p) cat[m][p] = enc[m][p];
      }
      f32x4 g1[8][P], g2[4][P];
      mlp_layer<8, 8, P, true>(g1, cat, lds_a2 + WX, lane);
      mlp_layer<8, 4, P, true>(g2, g1, lds_a2 + WX + W1, lane);
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float x0 = g2[o][0][c], x1 = g2[o][1][c];                     // post-ReLU: >= 0
          const float m = max_bits(x0, x1);
          const float e0 = exp_nonpos(x0 - m), e1 = exp_nonpos(x1 - m);
          const float d = e0 + e1;
          const float n = e0 * cat[4 + o][0][c] + e1 * cat[4 + o][1][c];
          if (FIRST) {
            mrun[o][c] = m; drun[o][c] = d; nrun[o][c] = n;
          } else {
            const float mm = max_bits(mrun[o][c], m);
            const float fa = exp_nonpos(mrun[o][c] - mm), fb = exp_nonpos(m - mm);
            mrun[o][c] = mm;
            drun[o][c] = drun[o][c] * fa + d * fb;
            nrun[o][c] = nrun[o][c] * fa + n * fb;
          }
        }
    };
    run_pass(std::true_type{}, 0);
#pragma unroll 1
    for (int pass = 1; pass < PASSES; ++pass) run_pass(std::false_type{}, pass);
    f32x4 res[4][1];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int c = 0; c < 4; ++c) res[o][0][c] = div_ge1(nrun[o][c], drun[o][c]);
    if (valid) {
#pragma unroll
      for (int o = 0; o < 4; ++o)
        *reinterpret_cast<f32x4 *>(at32(a.first, (row << 8) + 64u * o + 16u * (unsigned)g)) = res[o][0];
    }
    if constexpr (V2) {
      f32x4 pv[8][1];
      mlp_layer<4, 8, 1, false>(pv, res, a.w_v2, lane);      // weights from global memory (see the header comment)
      if (valid) {
#pragma unroll
        for (int o = 0; o < 8; ++o) st_group<false>(a.v2, row, 128u, o, g, pv[o][0]);
      }
    }
  }
}

// ---- launch helpers ----------------------------------------------------------------------------------------
template <int W, typename Kern, typename Args>
static void launch_h(Kern kern, bool &attr_set, int lds_bytes, long long ntiles, const Args &a) {
  if (lds_bytes > 64 * 1024 && !attr_set) {
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  // once per kernel: the largest any configuration can ask for
    attr_set = true;
  }
  static const int rounds = fh_tuning("PWCLO_FL_ROUNDS", 1);
  const int per_cu = (lds_bytes > 80 * 1024 || W > 8) ? 1 : 2;
  long long grid = (ntiles + W - 1) / W;
  if (grid > 256LL * per_cu * rounds) grid = 256LL * per_cu * rounds;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(W * 64), lds_bytes, current_stream(), a);
}

static int coarse_tiles_h() {   // fused_layers.hip: coarse_tiles()
  static const int v = fh_tuning("PWCLO_COARSE_W4_TILES", 2047);
  return v;
}

static long long tiles_h(int b, int s, int kp, int p) {
  return (long long)b * (((long long)s * kp + 16 * p - 1) / (16 * p));
}

}  // namespace pwclo

using namespace pwclo;

extern "C" void linear_jobs_kernel_wrapper(int njobs, const int *npts, const int *cin, const int *cout,
                                           const float *const *src, const float *const *w,
                                           float *const *out, const int *out_bf16) {
  if (njobs <= 0) return;
  PWCLO_REQUIRE(njobs <= LIN_MAX_JOBS, "linear_jobs: at most %d jobs per launch (got %d)", LIN_MAX_JOBS, njobs);
  LinArgs a;
  int max_tiles = 1, max_lds = 0;
  for (int i = 0; i < njobs; ++i) {
    const bool ok = (cin[i] == 16 || cin[i] == 32 || cin[i] == 64) &&
                    (cout[i] == 16 || cout[i] == 32 || cout[i] == 64 || cout[i] == 128);
    PWCLO_REQUIRE(ok, "linear_jobs: job %d has unsupported channels %d -> %d", i, cin[i], cout[i]);
    PWCLO_REQUIRE(rows_fit_32bit(npts[i]), "linear_jobs: job %d has too many rows for 32-bit offsets (%d)", i, npts[i]);
    a.job[i] = LinJob{src[i], w[i], out[i], npts[i], cin[i] / 16, cout[i] / 16, out_bf16 ? out_bf16[i] : 0};
    max_tiles = max(max_tiles, ceil_div(npts[i], 32));
    max_lds = max(max_lds, 4 * layer_floats(cin[i] / 16, cout[i] / 16));
  }
  int gx = ceil_div(max_tiles, LIN_WAVES);
  const int cap = max(1, 512 / njobs);
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL(linear_jobs_kernel, dim3(gx, njobs), dim3(LIN_WAVES * 64), max_lds, current_stream(), a);
  check_launch("linear_jobs");
}

extern "C" void sa_fused_h_kernel_wrapper(int b, int n, int s, int k, int c1, int c2, int c3, const float *xyz,
                                          const float *new_xyz, const float *pre, const int *idx,
                                          const float *packed_w, float *out, int wfmt, int packed_floats,
                                          int kmajor) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(k >= 1 && k <= 32, "sa_fused_h: nsample=%d outside [1,32]", k);
  PWCLO_REQUIRE(kmajor == 0 || (pre == nullptr && c1 == 16 && c2 == 16 && c3 == 16),
                "sa_fused_h: the k-step-major layout exists for the level-0 stack (16,16,16 without features) only");
  PWCLO_REQUIRE(rows_fit_32bit((long long)b * max(n, s * 32)), "sa_fused_h: batch too large for 32-bit offsets (b=%d)", b);
  SAHArgs a{xyz, new_xyz, pre, idx, packed_w, out, b, n, s, k};
  const int kp = k > 16 ? 32 : 16;
  const bool lvl0 = pre == nullptr;
#define SAH_CASE(A1, A2, A3, KP, XYZ, PP, WW)                                                         \
  if (c1 == A1 && c2 == A2 && c3 == A3 && kp == KP && lvl0 == XYZ) {                                  \
    static bool attr = false, attr3 = false;                                                          \
    constexpr int lds = 4 * (layer_floats(1, A1 / 16) + layer_floats(A1 / 16, A2 / 16) +               \
                             layer_floats(A2 / 16, A3 / 16));                                          \
    constexpr int lds3 = 4 * (layer_floats(1, A1 / 16) + layer_floats_any<1>(A1 / 16, A2 / 16) +       \
                              layer_floats_any<1>(A2 / 16, A3 / 16));                                  \
    constexpr int lds2 = 4 * (layer_floats(1, A1 / 16) + layer_floats_any<2>(A1 / 16, A2 / 16) +       \
                              layer_floats_any<2>(A2 / 16, A3 / 16));                                  \
    static bool attr2 = false;                                                                        \
    PWCLO_REQUIRE_PACKED("sa_fused_h", wfmt, packed_floats, lds / 4, lds3 / 4, lds2 / 4);              \
    if (wfmt == PWCLO_WFMT_BF16X3)                                                                    \
      launch_h<WW>(sa_h_kernel<A1 / 16, A2 / 16, A3 / 16, KP, PP, WW, XYZ, 1>, attr3, lds3,             \
                   tiles_h(b, s, KP, PP), a);                                                         \
    else if (wfmt == PWCLO_WFMT_BF16)                                                                 \
      launch_h<WW>(sa_h_kernel<A1 / 16, A2 / 16, A3 / 16, KP, PP, WW, XYZ, 2>, attr2, lds2,             \
                   tiles_h(b, s, KP, PP), a);                                                         \
    else                                                                                              \
      launch_h<WW>(sa_h_kernel<A1 / 16, A2 / 16, A3 / 16, KP, PP, WW, XYZ>, attr, lds, tiles_h(b, s, KP, PP), a); \
    check_launch("sa_fused_h");                                                                       \
    return;                                                                                           \
  }
  if (kmajor) {                               // psa_1 with 8-channel layers packed k-step major (fp32 tiles only)
    static bool attrk = false;
    constexpr int ldsk = 4 * (layer_floats(1, 1) + layer_floats(1, 1) + layer_floats(1, 1));
    PWCLO_REQUIRE(packed_floats == ldsk / 4, "sa_fused_h: packed weights hold %d floats, the level-0 stack needs %d",
                  packed_floats, ldsk / 4);
    PWCLO_REQUIRE(kp == 32, "sa_fused_h: the level-0 stack is built for nsample in (16, 32] (got %d)", k);
    launch_h<8>(sa_h_kernel<1, 1, 1, 32, 2, 8, true, 0, true>, attrk, ldsk, tiles_h(b, s, 32, 2), a);
    check_launch("sa_fused_h");
    return;
  }
  SAH_CASE(16, 16, 16, 32, true, 2, 8)      // psa_1
  SAH_CASE(16, 16, 32, 32, false, 2, 8)     // psa_2
  SAH_CASE(32, 32, 64, 16, false, 1, 16)    // psa_3
  SAH_CASE(64, 64, 128, 16, false, 1, 16)   // psa_4
  static const int coarse_w4 = fh_tuning("PWCLO_COARSE_W4", 1);
  if (coarse_w4 && tiles_h(b, s, 16, 1) <= coarse_tiles_h()) { SAH_CASE(128, 64, 64, 16, false, 1, 4) }   // flow_feature_encoding, coarse
  SAH_CASE(128, 64, 64, 16, false, 1, 16)   // flow_feature_encoding
#undef SAH_CASE
  set_error(PWCLO_EINVAL, "sa_fused_h: no kernel for mlp=(%d,%d,%d) nsample=%d level0=%d", c1, c2, c3, k, (int)lvl0);
}

extern "C" void upconv_fused_h_kernel_wrapper(int b, int n, int s, int k, const float *xyz2, const float *xyz1,
                                              const float *pre, const int *idx, const float *packed_w,
                                              float *out, int wfmt, int packed_floats) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(k >= 1 && k <= 8, "upconv_fused_h: nsample=%d outside [1,8]", k);
  PWCLO_REQUIRE(rows_fit_32bit((long long)b * max(n, s * 8)), "upconv_fused_h: batch too large for 32-bit offsets (b=%d)", b);
  UpHArgs a{xyz2, xyz1, pre, idx, packed_w, out, b, n, s, k};
  static bool attr = false, attr3 = false;
  constexpr int lds = 4 * (layer_floats(1, 8) + layer_floats(8, 4));
  constexpr int lds3 = 4 * (layer_floats(1, 8) + layer_floats_bf3(8, 4));
  static const int lane_up = fh_tuning("PWCLO_LANE_UP", 1);
  static bool attrl = false;
  const long long t16 = (long long)b * ((s + 15) / 16);
  constexpr int lds2 = 4 * (layer_floats(1, 8) + layer_floats_bf16(8, 4));
  static bool attr2 = false;
  PWCLO_REQUIRE_PACKED("upconv_fused_h", wfmt, packed_floats, lds / 4, lds3 / 4, lds2 / 4);
  if (wfmt == PWCLO_WFMT_BF16X3) launch_h<16>(upconv_h_kernel<8, 1, 16, 1>, attr3, lds3, tiles_h(b, s, 8, 1), a);
  else if (wfmt == PWCLO_WFMT_BF16) launch_h<16>(upconv_h_kernel<8, 1, 16, 2>, attr2, lds2, tiles_h(b, s, 8, 1), a);
  else if (lane_up && t16 > 2048) launch_h<16>(upconv_lane_kernel<16>, attrl, lds, t16, a);   // in-lane max over K
  else launch_h<16>(upconv_h_kernel<8, 1, 16>, attr, lds, tiles_h(b, s, 8, 1), a);
  check_launch("upconv_fused_h");
}

extern "C" void upconv_post_fused_h_kernel_wrapper(int njobs, int b, int n, int s, int k, int c2, const float *xyz2,
                                                   const float *xyz1, const int *idx, const float *feat2,
                                                   const float *const *pre, const float *const *packed_w,
                                                   const float *const *packed_post, float *const *out,
                                                   int packed_floats, int post_floats) {
  if (b <= 0 || s <= 0 || njobs <= 0) return;
  PWCLO_REQUIRE(njobs <= 2, "upconv_post_fused_h: at most 2 jobs per launch (got %d)", njobs);
  PWCLO_REQUIRE(k >= 1 && k <= 8, "upconv_post_fused_h: nsample=%d outside [1,8]", k);
  PWCLO_REQUIRE(c2 == 16 || c2 == 32 || c2 == 64, "upconv_post_fused_h: %d fine feature channels (16, 32 or 64)", c2);
  PWCLO_REQUIRE(rows_fit_32bit((long long)b * max(n, s * 8)), "upconv_post_fused_h: batch too large for 32-bit offsets (b=%d)", b);
  const int nb2 = c2 / 16;
  PWCLO_REQUIRE(packed_floats == layer_floats(1, 8) + layer_floats(8, 4),
                "upconv_post_fused_h: packed stack holds %d floats, needs %d (fp32 tiles)", packed_floats,
                layer_floats(1, 8) + layer_floats(8, 4));
  PWCLO_REQUIRE(post_floats == layer_floats(4 + nb2, 4), "upconv_post_fused_h: packed post layer holds %d floats, needs %d",
                post_floats, layer_floats(4 + nb2, 4));
  UpPostArgs a{xyz2, xyz1, idx, feat2, {pre[0], pre[njobs - 1]}, {packed_w[0], packed_w[njobs - 1]},
               {packed_post[0], packed_post[njobs - 1]}, {out[0], out[njobs - 1]}, b, n, s, k};
  const long long t16 = (long long)b * ((s + 15) / 16);
  const int lds = 4 * (layer_floats(1, 8) + layer_floats(8, 4) + layer_floats(4 + nb2, 4));
  // 16-wave workgroups when every wave slot of the chip gets a tile, 8-wave ones (twice the workgroups) below that, 4-wave
  // ones at a coarse level (a few hundred tiles: spread them over as many CUs as possible)
  const long long tiles = t16 * njobs;
  const int W = tiles >= 4096 ? 16 : tiles >= 2048 ? 8 : 4;
  long long gx = (t16 + W - 1) / W;
  const long long cap = (W == 16 ? 512 : W == 8 ? 1024 : 2048) / njobs;
  if (gx > cap) gx = cap;
#define UPP_CASE(NB2)                                                                                         \
  if (nb2 == NB2) {                                                                                           \
    static bool attr16 = false, attr8 = false, attr4 = false;                                                 \
    if (W == 4) {                                                                                             \
      if (!attr4) { (void)hipFuncSetAttribute((const void *)upconv_lane_post_kernel<NB2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr4 = true; } \
      hipLaunchKernelGGL((upconv_lane_post_kernel<NB2, 4>), dim3((unsigned)gx, njobs), dim3(4 * 64), lds, current_stream(), a); \
    } else if (W == 16) {                                                                                     \
      if (!attr16) { (void)hipFuncSetAttribute((const void *)upconv_lane_post_kernel<NB2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr16 = true; } \
      hipLaunchKernelGGL((upconv_lane_post_kernel<NB2, 16>), dim3((unsigned)gx, njobs), dim3(16 * 64), lds, current_stream(), a); \
    } else {                                                                                                  \
      if (!attr8) { (void)hipFuncSetAttribute((const void *)upconv_lane_post_kernel<NB2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr8 = true; } \
      hipLaunchKernelGGL((upconv_lane_post_kernel<NB2, 8>), dim3((unsigned)gx, njobs), dim3(8 * 64), lds, current_stream(), a); \
    }                                                                                                         \
    check_launch("upconv_post_fused_h");                                                                      \
    return;                                                                                                   \
  }
  UPP_CASE(1) UPP_CASE(2) UPP_CASE(4)
#undef UPP_CASE
}

extern "C" void cv_fused_a1_h_kernel_wrapper(int b, int n, int s, int k, const float *xyz1, const float *u,
                                             const float *xyz2, const float *v, const int *idx,
                                             const float *packed_w, float *pix, int pix_slots, int wfmt,
                                             int packed_floats) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(k >= 1 && k <= 32, "cv_fused_a1_h: nsample_q=%d outside [1,32]", k);
  PWCLO_REQUIRE(cv_pix_slots_valid(k, pix_slots), "cv_fused_a1_h: pix_slots=%d invalid for nsample_q=%d", pix_slots, k);
  PWCLO_REQUIRE(rows_fit_32bit((long long)b * max(n, s * 32)), "cv_fused_a1_h: batch too large for 32-bit offsets (b=%d)", b);
  CVHArgs a{xyz1, u, xyz2, v, nullptr, idx, packed_w, pix, b, n, s, k};
  const int kp = pix_slots;
  constexpr int lds = 4 * (layer_floats(1, 8) + layer_floats(8, 4) + layer_floats(4, 4));
  constexpr int lds3 = 4 * (layer_floats(1, 8) + layer_floats_bf3(8, 4) + layer_floats_bf3(4, 4));
  static bool a32 = false, a16 = false, a8 = false, a6 = false, b32 = false, b16 = false, b8 = false, b6 = false;
  constexpr int lds2 = 4 * (layer_floats(1, 8) + layer_floats_bf16(8, 4) + layer_floats_bf16(4, 4));
  PWCLO_REQUIRE_PACKED("cv_fused_a1_h", wfmt, packed_floats, lds / 4, lds3 / 4, lds2 / 4);
  if (wfmt == PWCLO_WFMT_BF16X3) {
    if (kp == 6) launch_h<16>(cv_a1_h_kernel<6, 1, 16, 1>, b6, lds3, tiles_h(b, s, 6, 1), a);
    else if (kp == 32) launch_h<16>(cv_a1_h_kernel<32, 1, 16, 1>, b32, lds3, tiles_h(b, s, 32, 1), a);
    else if (kp == 16) launch_h<16>(cv_a1_h_kernel<16, 1, 16, 1>, b16, lds3, tiles_h(b, s, 16, 1), a);
    else launch_h<16>(cv_a1_h_kernel<8, 1, 16, 1>, b8, lds3, tiles_h(b, s, 8, 1), a);
    check_launch("cv_fused_a1_h");
    return;
  }
  if (wfmt == PWCLO_WFMT_BF16) {
    static bool c32 = false, c16 = false, c8 = false, c6 = false;
    if (kp == 6) launch_h<16>(cv_a1_h_kernel<6, 1, 16, 2>, c6, lds2, tiles_h(b, s, 6, 1), a);
    else if (kp == 32) launch_h<16>(cv_a1_h_kernel<32, 1, 16, 2>, c32, lds2, tiles_h(b, s, 32, 1), a);
    else if (kp == 16) launch_h<16>(cv_a1_h_kernel<16, 1, 16, 2>, c16, lds2, tiles_h(b, s, 16, 1), a);
    else launch_h<16>(cv_a1_h_kernel<8, 1, 16, 2>, c8, lds2, tiles_h(b, s, 8, 1), a);
    check_launch("cv_fused_a1_h");
    return;
  }
  if (kp == 6) launch_h<16>(cv_a1_h_kernel<6, 1, 16>, a6, lds, tiles_h(b, s, 6, 1), a);
  else if (kp == 32) launch_h<16>(cv_a1_h_kernel<32, 1, 16>, a32, lds, tiles_h(b, s, 32, 1), a);
  else if (kp == 16) launch_h<16>(cv_a1_h_kernel<16, 1, 16>, a16, lds, tiles_h(b, s, 16, 1), a);
  else launch_h<16>(cv_a1_h_kernel<8, 1, 16>, a8, lds, tiles_h(b, s, 8, 1), a);
  check_launch("cv_fused_a1_h");
}

extern "C" void cv_fused_a_lane6_kernel_wrapper(int b, int n, int s, const float *xyz1, const float *u, const float *xyz2,
                                                const float *v, const int *idx, const float *packed_a1,
                                                const float *packed_a2, const float *packed_v2, float *first,
                                                float *v2, int a1_floats, int a2_floats, int v2_floats) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(rows_fit_32bit((long long)b * max(n, s * 6)), "cv_fused_a_lane6: batch too large for 32-bit offsets (b=%d)", b);
  constexpr int fa1 = layer_floats(1, 8) + layer_floats(8, 4) + layer_floats(4, 4);
  constexpr int fa2 = layer_floats(1, 4) + layer_floats(8, 8) + layer_floats(8, 4);
  PWCLO_REQUIRE(a1_floats == fa1 && a2_floats == fa2, "cv_fused_a_lane6: packed stacks hold %d / %d floats, need %d / %d (fp32 tiles)",
                a1_floats, a2_floats, fa1, fa2);
  PWCLO_REQUIRE((packed_v2 == nullptr) == (v2 == nullptr) && (packed_v2 == nullptr || v2_floats == layer_floats(4, 8)),
                "cv_fused_a_lane6: the v2 layer needs both its packed weights (%d floats, got %d) and an output", layer_floats(4, 8), v2_floats);
  CVLaneArgs a{xyz1, u, xyz2, v, idx, packed_a1, packed_a2, packed_v2, first, v2, b, n, s};
  constexpr int lds = 4 * (fa1 + fa2);
  static_assert(lds <= 160 * 1024, "both stages' weights must fit the 160 KiB of LDS");
  const long long t16 = (long long)b * ((s + 15) / 16);
  static bool attr = false, attr_v = false, attr_v4 = false;
  // a coarse level's few hundred tiles: 4-wave workgroups spread them over twice the CUs
  if (packed_v2 != nullptr && t16 <= 1024) launch_h<4>(cv_a_lane6_kernel<4, true>, attr_v4, lds, t16, a);
  else if (packed_v2 != nullptr) launch_h<8>(cv_a_lane6_kernel<8, true>, attr_v, lds, t16, a);
  else launch_h<8>(cv_a_lane6_kernel<8, false>, attr, lds, t16, a);
  check_launch("cv_fused_a_lane6");
}

extern "C" void cv_fused_b_h_kernel_wrapper(int b, int s, int k, const float *xyz1, const float *u2,
                                            const float *v2, const float *first, const int *idx,
                                            const float *packed_w, float *out, int wfmt, int packed_floats) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(k >= 1 && k <= 4, "cv_fused_b_h: nsample=%d outside [1,4]", k);
  PWCLO_REQUIRE(rows_fit_32bit((long long)b * s * 4), "cv_fused_b_h: batch too large for 32-bit offsets (b=%d)", b);
  CVHArgs a{xyz1, u2, xyz1, v2, first, idx, packed_w, out, b, s, s, k};
  static bool attr = false;
  constexpr int lds = 4 * (layer_floats(1, 4) + layer_floats(4, 8) + layer_floats(8, 4));
  static bool attr_s = false, attr3 = false, attr3_s = false;
  constexpr int lds3 = 4 * (layer_floats(1, 4) + layer_floats_bf3(4, 8) + layer_floats_bf3(8, 4));
  static const int coarse_w4 = fh_tuning("PWCLO_COARSE_W4", 1);
  const bool small = coarse_w4 && tiles_h(b, s, 4, 1) <= coarse_tiles_h();
  constexpr int lds2 = 4 * (layer_floats(1, 4) + layer_floats_bf16(4, 8) + layer_floats_bf16(8, 4));
  static bool attr2 = false, attr2_s = false;
  PWCLO_REQUIRE_PACKED("cv_fused_b_h", wfmt, packed_floats, lds / 4, lds3 / 4, lds2 / 4);
  if (wfmt == PWCLO_WFMT_BF16X3 && small) launch_h<4>(cv_b_h_kernel<4, 1, 4, 1>, attr3_s, lds3, tiles_h(b, s, 4, 1), a);
  else if (wfmt == PWCLO_WFMT_BF16X3) launch_h<16>(cv_b_h_kernel<4, 1, 16, 1>, attr3, lds3, tiles_h(b, s, 4, 1), a);
  else if (wfmt == PWCLO_WFMT_BF16 && small) launch_h<4>(cv_b_h_kernel<4, 1, 4, 2>, attr2_s, lds2, tiles_h(b, s, 4, 1), a);
  else if (wfmt == PWCLO_WFMT_BF16) launch_h<16>(cv_b_h_kernel<4, 1, 16, 2>, attr2, lds2, tiles_h(b, s, 4, 1), a);
  else if (small) launch_h<4>(cv_b_h_kernel<4, 1, 4>, attr_s, lds, tiles_h(b, s, 4, 1), a);
  else launch_h<16>(cv_b_h_kernel<4, 1, 16>, attr, lds, tiles_h(b, s, 4, 1), a);
  check_launch("cv_fused_b_h");
}
