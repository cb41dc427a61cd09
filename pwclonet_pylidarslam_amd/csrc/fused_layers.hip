// Fused eval-mode layers of the pose warp-refinement path (gfx950): set-upconv, point-wise MLPs
// (post-MLP, flow / mask predictors), the attentive cost volume and the mask-softmax pooling of
// the pose head.  All built on mlp_core.hpp: activations stay in registers across a stack's
// layers, packed weights stay resident in LDS, persistent workgroups stream 16-pixel blocks.
//
// The cost volume (PW/costvolume.py:63-190) is cut where an activation has to outlive a weight
// set that no longer fits in the 160 KiB LDS next to the following one:
//   cv_a1: [geometry10 | centre feat | gathered feat] -> mlp_convs (128,64,64) -> feat (per pixel)
//   cv_a2: geometry10 -> mlp_conv_xyz_1 ; [enc | feat] -> mlp2_convs (128,64) -> softmax over the
//          neighbours -> sum_k w * feat                                   ("first aggregate", :86-144)
//   cv_b : geometry10' -> mlp_conv_xyz_2 ; [enc2 | centre feat | gathered first] -> mlp3_convs
//          -> softmax over the neighbours -> sum_k w * gathered first     ("second aggregate", :146-188)
// The only materialised intermediate is cv_a1's 64-channel per-pixel feature (written once, read
// once); the reference materialises > 10 (B,C,S,K) tensors for the same result.
#include <math.h>
#include <stdlib.h>

#include "mlp_core.hpp"

PWCLO_TRACE_TU(fused_layers)

namespace pwclo {


// ---- shared prologue pieces ----------------------------------------------------------------------

// 16-channel geometry block [p(3), q(3), q-p(3), |q-p|, 0 x 6] (costvolume.py:92-105) for lane
// group g: g0 = (px,py,pz,qx), g1 = (qy,qz,dx,dy), g2 = (dz,euc,0,0), g3 = 0.
__device__ __forceinline__ f32x4 geometry_block(const float *p, const float *q, int g) {
  const float px = p[0], py = p[1], pz = p[2], qx = q[0], qy = q[1], qz = q[2];
  const float dx = qx - px, dy = qy - py, dz = qz - pz;
  const float euc = sqrtf(((dx * dx + dy * dy) + dz * dz) + 1e-20f);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (g == 0) v = f32x4{px, py, pz, qx};
  if (g == 1) v = f32x4{qy, qz, dx, dy};
  if (g == 2) v = f32x4{dz, euc, 0.f, 0.f};
  return v;
}

template <int NB>
__device__ __forceinline__ void load_row_blocks(f32x4 *dst, int stride_p, const float *row, int g) {
#pragma unroll
  for (int m = 0; m < NB; ++m) dst[m * stride_p] = *reinterpret_cast<const f32x4 *>(row + 16 * m + 4 * g);
}
// the cost volume's per-pixel features (64 channels): bf16 rows when the stack format is bf16 (mlp_core.hpp)
template <bool H16>
__device__ __forceinline__ void load_pix_blocks(f32x4 *dst, int stride_p, const float *pix, unsigned slot, int g) {
#pragma unroll
  for (int m = 0; m < 4; ++m) dst[m * stride_p] = ld_group<H16>(pix, slot, 64u, m, g);
}

// Softmax over the K neighbours of each query (dim=3 of the reference's (B,C,S,K) tensor) and
// weighted sum of `val`; logits are post-ReLU (>= 0), padded neighbour slots carry weight 0.
// Returns, in every lane of a neighbour group, sum_k softmax(x)_k * val_k per component.
template <int KP, int P>
__device__ __forceinline__ void softmax_weighted_sum(f32x4 (&res)[P], const f32x4 (&x)[P],
                                                     const f32x4 (&val)[P], const bool (&padded)[P]) {
  constexpr int GROUP = KP < 16 ? KP : 16;
  constexpr int BPQ = KP > 16 ? KP / 16 : 1;
  const float NEG_INF = __int_as_float(0xff800000);
#pragma unroll
  for (int p = 0; p < P; p += BPQ) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float xv[BPQ];
      float mx = NEG_INF;
#pragma unroll
      for (int e = 0; e < BPQ; ++e) {
        xv[e] = padded[p + e] ? NEG_INF : x[p + e][c];
        mx = (__float_as_int(xv[e]) > __float_as_int(mx)) ? xv[e] : mx;  // values >= 0 or -inf
      }
      mx = group_max_nonneg<GROUP>(mx);
      float den = 0.f, num = 0.f;
#pragma unroll
      for (int e = 0; e < BPQ; ++e) {
        const float ex = padded[p + e] ? 0.f : exp_nonpos(xv[e] - mx);
        den += ex;
        num += ex * val[p + e][c];
      }
      den = group_sum<GROUP>(den);
      num = group_sum<GROUP>(num);
      const float r = div_ge1(num, den);
#pragma unroll
      for (int e = 0; e < BPQ; ++e) res[p + e][c] = r;
    }
  }
}

#define PWCLO_TILE_LOOP(KP_, P_, S_, B_)                                                           \
  constexpr int TILE = 16 * (P_);                                                                   \
  const int pix_per_cloud = (S_) * (KP_);                                                           \
  const int tiles_per_cloud = (pix_per_cloud + TILE - 1) / TILE;                                    \
  const int ntiles = (B_) * tiles_per_cloud;                                                        \
  for (int t = blockIdx.x * W + wave_index(); t < ntiles; t += gridDim.x * W)

// ---- set-upconv: PointnetFPModulePWCLONet, knn branch up to the max (pointnet2_modules.py:479-506) --
struct UpconvArgs {
  const float *xyz2;    // (B,S,3) fine points (queries)
  const float *xyz1;    // (B,N,3) coarse points
  const float *feat1;   // (B,N,64) coarse features, point-major
  const int *idx;       // (B,S,K)
  const float *w;       // packed [64 feat | diff block] -> 128 -> 64
  float *out;           // (B,S,64)
  int B, N, S, K;
  int stagger;
};

template <int KP, int P, int W>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void upconv_kernel(UpconvArgs a) {
  constexpr int NBI = 5, B1 = 8, B2 = 4;
  constexpr int W1 = layer_floats(NBI, B1), W2 = layer_floats(B1, B2);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, W1 + W2);
  __syncthreads();
  stagger_start(threadIdx.x >> 6, (NBI * B1 + B1 * B2) * 4 * P, a.stagger);
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  PWCLO_TILE_LOOP(KP, P, a.S, a.B) {
    const int b = t / tiles_per_cloud;
    const int pix0 = (t - b * tiles_per_cloud) * TILE;
    f32x4 in[NBI][P];
    int sq[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const PixelMap<KP> pm(pix0 + 16 * p + j);
      const bool valid = pm.s < a.S;
      const int s = valid ? pm.s : a.S - 1;
      const int k = pm.k < a.K ? pm.k : 0;
      sq[p] = valid ? s : -1;
      const int nbr = a.idx[((size_t)b * a.S + s) * a.K + k];
      load_row_blocks<4>(&in[0][p], P, a.feat1 + ((size_t)b * a.N + nbr) * 64, g);
      const float *c = a.xyz2 + ((size_t)b * a.S + s) * 3;
      const float *q = a.xyz1 + ((size_t)b * a.N + nbr) * 3;
      f32x4 d = {0.f, 0.f, 0.f, 0.f};
      if (g == 0) d = f32x4{q[0] - c[0], q[1] - c[1], q[2] - c[2], 0.f};
      in[4][p] = d;
    }
    f32x4 h1[B1][P], h2[B2][P];
    mlp_layer<NBI, B1, P, true>(h1, in, lds_w, lane);
    mlp_layer<B1, B2, P, false>(h2, h1, lds_w + W1, lane);        // its ReLU is applied after the pool
    constexpr int GROUP = KP < 16 ? KP : 16;
#pragma unroll
    for (int o = 0; o < B2; ++o) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        f32x4 v = h2[o][p];
        v.x = relu_bits(group_max_nonneg<GROUP>(v.x));
        v.y = relu_bits(group_max_nonneg<GROUP>(v.y));
        v.z = relu_bits(group_max_nonneg<GROUP>(v.z));
        v.w = relu_bits(group_max_nonneg<GROUP>(v.w));
        if ((j & (GROUP - 1)) == 0 && sq[p] >= 0)
          *reinterpret_cast<f32x4 *>(a.out + ((size_t)b * a.S + sq[p]) * 64 + 16 * o + 4 * g) = v;
      }
    }
  }
}

// ---- point-wise MLP over concatenated per-point sources (K = 1) ------------------------------------
// post_mlp (pointnet2_modules.py:508-515), FlowPredictor (PW/flowpredictor.py:53-83).
struct PointwiseArgs {
  const float *src[3];  // (B,S,16*NB*) point-major, concatenated in this order
  const float *w;       // packed layers
  float *out;           // (B,S,16*BOUT)
  int B, S;
  int stagger;
  const float *w_tail;  // BT > 0: packed single layer 16*B2 -> 16*BT, no activation (a consumer's hoisted partial product)
  float *out_tail;      // BT > 0: (B,S,16*BT)
};

// BT > 0: the linear map a consumer of this stack's output would otherwise get from linear_jobs (the next refinement
// level's set-upconv seeds W1_feat . out + b1) runs as a third layer while the output is in registers -- the accumulators
// of layer 2 are its B operands -- with its tiles LDS resident beside the stack's; same routine, operands and k order as
// linear_jobs_kernel: bit-identical rows.
template <int NB0, int NB1, int NB2, int B1, int B2 /*0 = single layer*/, int P, int W, int BT = 0>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void pointwise_kernel(PointwiseArgs a) {
  TraceScope trace_scope_(TK_POINTWISE);
  constexpr int NBI = NB0 + NB1 + NB2;
  constexpr int W1 = layer_floats(NBI, B1);
  constexpr int W2 = B2 > 0 ? layer_floats(B1, B2) : 0;
  constexpr int WT = BT > 0 ? layer_floats(B2, BT) : 0;
  constexpr int BOUT = B2 > 0 ? B2 : B1;
  static_assert(BT == 0 || B2 > 0, "the tail follows a two-layer stack");
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, W1 + W2);
  if constexpr (BT > 0) stage_weights(lds_w + W1 + W2, a.w_tail, WT);
  __syncthreads();
  stagger_start(threadIdx.x >> 6, (NBI * B1 + B1 * B2) * 4 * P, a.stagger);
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  PWCLO_TILE_LOOP(1, P, a.S, a.B) {
    const int b = t / tiles_per_cloud;
    const int pix0 = (t - b * tiles_per_cloud) * TILE;
    f32x4 in[NBI][P];
    int sq[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int s0 = pix0 + 16 * p + j;
      const bool valid = s0 < a.S;
      const int s = valid ? s0 : a.S - 1;
      sq[p] = valid ? s : -1;
      const unsigned row = (unsigned)b * (unsigned)a.S + (unsigned)s;
      load_row_blocks<NB0>(&in[0][p], P, at32(a.src[0], row * (unsigned)(64 * NB0)), g);
      if (NB1 > 0) load_row_blocks<NB1>(&in[NB0][p], P, at32(a.src[1], row * (unsigned)(64 * NB1)), g);
      if (NB2 > 0) load_row_blocks<NB2>(&in[NB0 + NB1][p], P, at32(a.src[2], row * (unsigned)(64 * NB2)), g);
    }
    f32x4 h1[B1][P];
    mlp_layer<NBI, B1, P, true>(h1, in, lds_w, lane);
    if constexpr (B2 > 0) {
      f32x4 h2[B2][P];
      mlp_layer<B1, B2, P, true>(h2, h1, lds_w + W1, lane);
#pragma unroll
      for (int o = 0; o < B2; ++o)
#pragma unroll
        for (int p = 0; p < P; ++p)
          if (sq[p] >= 0)
            *reinterpret_cast<f32x4 *>(at32(a.out, ((unsigned)b * (unsigned)a.S + (unsigned)sq[p]) * (unsigned)(64 * BOUT) +
                                                        64u * o + 16u * (unsigned)g)) = h2[o][p];
      if constexpr (BT > 0) {
        f32x4 ht[BT][P];
        mlp_layer<B2, BT, P, false>(ht, h2, lds_w + W1 + W2, lane);
#pragma unroll
        for (int o = 0; o < BT; ++o)
#pragma unroll
          for (int p = 0; p < P; ++p)
            if (sq[p] >= 0)
              st_group<false>(a.out_tail, (unsigned)b * (unsigned)a.S + (unsigned)sq[p], 16u * BT, o, g, ht[o][p]);
      }
    } else {
#pragma unroll
      for (int o = 0; o < B1; ++o)
#pragma unroll
        for (int p = 0; p < P; ++p)
          if (sq[p] >= 0)
            *reinterpret_cast<f32x4 *>(at32(a.out, ((unsigned)b * (unsigned)a.S + (unsigned)sq[p]) * (unsigned)(64 * BOUT) +
                                                        64u * o + 16u * (unsigned)g)) = h1[o][p];
    }
  }
}

// ---- cost volume ------------------------------------------------------------------------------------
struct CVArgs {
  const float *xyz1;    // (B,S,3) (warped) frame-1 points = queries
  const float *feat1;   // (B,S,C) frame-1 features, point-major
  const float *xyz2;    // (B,N,3) candidate points (frame 2 for a1/a2, frame 1 for b)
  const float *feat2;   // a1: (B,N,C) frame-2 features; b: (B,S,64) first-aggregate result
  const int *idx;       // (B,S,K)
  const float *w;       // packed weights of this stage
  float *pix;           // a1: out (B,S*KP,64); a2: in (same)
  float *out;           // a2 / b: (B,S,64)
  int B, N, S, K;
  int stagger;
};

// a1: [geo | centre feat (CB blocks) | gathered feat (CB blocks)] -> 128 -> 64 -> 64, stored per pixel.
template <int CB, int KP, int P, int W>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void cv_a1_kernel(CVArgs a) {
  constexpr int NBI = 1 + 2 * CB, B1 = 8, B2 = 4, B3 = 4, C = 16 * CB;
  constexpr int W1 = layer_floats(NBI, B1), W2 = layer_floats(B1, B2), W3 = layer_floats(B2, B3);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, W1 + W2 + W3);
  __syncthreads();
  stagger_start(threadIdx.x >> 6, (NBI * B1 + B1 * B2 + B2 * B3) * 4 * P, a.stagger);
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  PWCLO_TILE_LOOP(KP, P, a.S, a.B) {
    const int b = t / tiles_per_cloud;
    const int pix0 = (t - b * tiles_per_cloud) * TILE;
    f32x4 in[NBI][P];
    int pixv[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int pix = pix0 + 16 * p + j;
      const PixelMap<KP> pm(pix);
      const bool valid = pm.s < a.S;
      const int s = valid ? pm.s : a.S - 1;
      const int k = pm.k < a.K ? pm.k : 0;
      pixv[p] = (valid && pm.k < a.K) ? pix : -1;      // padded slots are never read back (cv_a2)
      const int nbr = a.idx[((size_t)b * a.S + s) * a.K + k];
      in[0][p] = geometry_block(a.xyz1 + ((size_t)b * a.S + s) * 3, a.xyz2 + ((size_t)b * a.N + nbr) * 3, g);
      load_row_blocks<CB>(&in[1][p], P, a.feat1 + ((size_t)b * a.S + s) * C, g);
      load_row_blocks<CB>(&in[1 + CB][p], P, a.feat2 + ((size_t)b * a.N + nbr) * C, g);
    }
    f32x4 h1[B1][P], h2[B2][P], h3[B3][P];
    mlp_layer<NBI, B1, P, true>(h1, in, lds_w, lane);
    mlp_layer<B1, B2, P, true>(h2, h1, lds_w + W1, lane);
    mlp_layer<B2, B3, P, true>(h3, h2, lds_w + W1 + W2, lane);
#pragma unroll
    for (int o = 0; o < B3; ++o)
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (pixv[p] >= 0)
          *reinterpret_cast<f32x4 *>(a.pix + ((size_t)b * pix_per_cloud + pixv[p]) * 64 + 16 * o + 4 * g) = h3[o][p];
  }
}

// a2: enc = mlp_conv_xyz_1(geo); w = softmax_k(mlp2_convs([enc | feat])); out = sum_k w * feat.
template <int KP, int P, int W, int FMT = 0>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void cv_a2_kernel(CVArgs a) {
  TraceScope trace_scope_(TK_CV_A2);
  constexpr int WX = layer_floats(1, 4), W1 = layer_floats_any<FMT>(8, 8), W2 = layer_floats_any<FMT>(8, 4);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, WX + W1 + W2);
  __syncthreads();
  stagger_start(threadIdx.x >> 6, (4 + 64 + 32) * 4 * P, a.stagger);
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  PWCLO_TILE_LOOP(KP, P, a.S, a.B) {
    const int b = t / tiles_per_cloud;
    const int pix0 = (t - b * tiles_per_cloud) * TILE;
    f32x4 geo[1][P], cat[8][P];
    int sq[P];
    bool padded[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int pix = pix0 + 16 * p + j;
      const PixelMap<KP> pm(pix);
      const bool valid = pm.s < a.S;
      const int s = valid ? pm.s : a.S - 1;
      padded[p] = pm.k >= a.K;
      const int k = padded[p] ? 0 : pm.k;
      sq[p] = valid ? s : -1;
      const unsigned row = (unsigned)b * (unsigned)a.S + (unsigned)s;
      const int nbr = *at32(a.idx, (mul24(row, (unsigned)a.K) + (unsigned)k) * 4u);
      const unsigned src = (unsigned)b * (unsigned)a.N + (unsigned)nbr;
      geo[0][p] = geometry_block(at32(a.xyz1, mul24(row, 12u)), at32(a.xyz2, mul24(src, 12u)), g);
      // padded slots (masked below) re-read slot 0's row: a1 does not write them, no extra traffic
      const int pixc = valid ? pix - (padded[p] ? pm.k : 0) : pix_per_cloud - KP;
      load_pix_blocks<FMT == 2>(&cat[4][p], P, a.pix, (unsigned)b * (unsigned)pix_per_cloud + (unsigned)pixc, g);
    }
    f32x4 enc[4][P];
    mlp_layer<1, 4, P, true>(enc, geo, lds_w, lane);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int p = 0; p < P; ++p) cat[m][p] = enc[m][p];
    f32x4 h1[8][P], h2[4][P];
    mlp_layer_any<FMT, 8, 8, P, true>(h1, cat, lds_w + WX, lane);
    mlp_layer_any<FMT, 8, 4, P, true>(h2, h1, lds_w + WX + W1, lane);
    constexpr int GROUP = KP < 16 ? KP : 16;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      f32x4 res[P];
      softmax_weighted_sum<KP, P>(res, h2[o], cat[4 + o], padded);
      constexpr int BPQ = KP > 16 ? KP / 16 : 1;
#pragma unroll
      for (int p = 0; p < P; p += BPQ)
        if ((j & (GROUP - 1)) == 0 && sq[p] >= 0)
          *reinterpret_cast<f32x4 *>(at32(a.out, (((unsigned)b * (unsigned)a.S + (unsigned)sq[p]) << 8) + 64u * o +
                                                      16u * (unsigned)g)) = res[p];
    }
  }
}

// a2 for K == 6 with no padded slots (the refinement levels: nsample_q = 6 would waste a quarter of
// the matrix work in 8-wide groups).  A wave tile = 8 consecutive queries = 48 pixels = 3 blocks.
// Lane j of block p (h = j >> 3, i = j & 7):
//   i <  6: query 2p + h, neighbour i          -- "main" segment, 6 lanes of an 8-lane half row:
//           reduced with the group-of-8 DPP ops, lanes 6,7 contributing the neutral element;
//   i >= 6: query 6 + h,  neighbour 2p + (i-6) -- "split" segment, lanes 6,7 / 14,15 of the three
//           blocks: reduced in-lane across the blocks, then across the lane pair (one DPP step).
// cv_a1 writes the per-pixel features densely as (B, S, 6, 64) for this kernel.
template <int W, int FMT = 0>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void cv_a2_dense6_kernel(CVArgs a) {
  TraceScope trace_scope_(TK_CV_A2_DENSE6);
  constexpr int P = 3;
  constexpr int WX = layer_floats(1, 4), W1 = layer_floats_any<FMT>(8, 8), W2 = layer_floats_any<FMT>(8, 4);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, WX + W1 + W2);
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int i = j & 7, h = j >> 3;
  const bool split = i >= 6;
  const int tiles_per_cloud = (a.S + 7) / 8;
  const int ntiles = a.B * tiles_per_cloud;
  for (int t = blockIdx.x * W + wave_index(); t < ntiles; t += gridDim.x * W) {
    const int b = t / tiles_per_cloud;
    const int s0 = (t - b * tiles_per_cloud) * 8;
    f32x4 geo[1][P], cat[8][P];
    int sq[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int q = s0 + (split ? 6 + h : 2 * p + h);
      const int k = split ? 2 * p + (i - 6) : i;
      const bool valid = q < a.S;
      const int s = valid ? q : a.S - 1;
      sq[p] = valid ? s : -1;
      const unsigned row = (unsigned)b * (unsigned)a.S + (unsigned)s;
      const unsigned slot = mul24(row, 6u) + (unsigned)k;
      const int nbr = *at32(a.idx, slot * 4u);
      const unsigned src = (unsigned)b * (unsigned)a.N + (unsigned)nbr;
      geo[0][p] = geometry_block(at32(a.xyz1, mul24(row, 12u)), at32(a.xyz2, mul24(src, 12u)), g);
      load_pix_blocks<FMT == 2>(&cat[4][p], P, a.pix, slot, g);
    }
    f32x4 h2[4][P];
    if constexpr (FMT != 0) {
      // the split operands of three blocks at once do not fit the register file: one 16-pixel block at a time
#pragma unroll
      for (int p = 0; p < P; ++p) {
        f32x4 g1[1][1] = {{geo[0][p]}}, enc1[4][1], c1[8][1], h1[8][1], o1[4][1];
        mlp_layer<1, 4, 1, true>(enc1, g1, lds_w, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m) { c1[m][0] = enc1[m][0]; c1[4 + m][0] = cat[4 + m][p]; }
        mlp_layer_any<FMT, 8, 8, 1, true>(h1, c1, lds_w + WX, lane);
        mlp_layer_any<FMT, 8, 4, 1, true>(o1, h1, lds_w + WX + W1, lane);
#pragma unroll
        for (int o = 0; o < 4; ++o) h2[o][p] = o1[o][0];
      }
    } else {
      f32x4 enc[4][P];
      mlp_layer<1, 4, P, true>(enc, geo, lds_w, lane);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int p = 0; p < P; ++p) cat[m][p] = enc[m][p];
      f32x4 h1[8][P];
      mlp_layer<8, 8, P, true>(h1, cat, lds_w + WX, lane);
      mlp_layer<8, 4, P, true>(h2, h1, lds_w + WX + W1, lane);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      f32x4 res[P];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // logits are post-ReLU (>= 0): 0 is neutral for the max, and every segment has 6 real entries
        float ms = split ? max_bits(max_bits(h2[o][0][c], h2[o][1][c]), h2[o][2][c]) : 0.f;
        ms = max_bits(ms, __uint_as_float(dpp_u32<0xB1>(__float_as_uint(ms))));
        float e[P];
        float ds = 0.f, ns = 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float mm = group_max_nonneg<8>(split ? 0.f : h2[o][p][c]);
          e[p] = exp_nonpos(h2[o][p][c] - (split ? ms : mm));
          ds += split ? e[p] : 0.f;
          ns += split ? e[p] * cat[4 + o][p][c] : 0.f;
        }
        ds += __uint_as_float(dpp_u32<0xB1>(__float_as_uint(ds)));
        ns += __uint_as_float(dpp_u32<0xB1>(__float_as_uint(ns)));
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float dm = group_sum<8>(split ? 0.f : e[p]);
          const float nm = group_sum<8>(split ? 0.f : e[p] * cat[4 + o][p][c]);
          res[p][c] = split ? div_ge1(ns, ds) : div_ge1(nm, dm);
        }
      }
#pragma unroll
      for (int p = 0; p < P; ++p)
        if ((i == 0 || (i == 6 && p == 0)) && sq[p] >= 0)
          *reinterpret_cast<f32x4 *>(at32(a.out, (((unsigned)b * (unsigned)a.S + (unsigned)sq[p]) << 8) + 64u * o +
                                                      16u * (unsigned)g)) = res[p];
    }
  }
}

// a2 for K == 6 with the softmax IN-LANE: a wave tile = 16 consecutive queries; pixel block p of pass `grp` holds
// neighbour 3*grp + p of those 16 queries (lane j <-> query j), so the six logits of a query and channel sit in the
// same lane of six blocks and the softmax-weighted sum needs no DPP step, no select and no padded lane.  The six
// blocks run as three passes of two (256 VGPRs without spills); the passes are merged with the
// running-maximum form of the softmax: (m, d, n) -> m' = max(m_a, m_b), d' = d_a e^(m_a-m') + d_b e^(m_b-m'), same
// for n; out = n / d.  Same values as the direct form up to two extra roundings (1e-7 relative).
template <int W>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void cv_a2_lane6_kernel(CVArgs a) {
  TraceScope trace_scope_(TK_CV_A2_LANE6);
  constexpr int P = 2, PASSES = 3;          // three passes of two neighbour blocks: fits 256 VGPRs without spills
  constexpr int WX = layer_floats(1, 4), W1 = layer_floats(8, 8), W2 = layer_floats(8, 4);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, WX + W1 + W2);
  __syncthreads();
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int tiles_per_cloud = (a.S + 15) / 16;
  const int ntiles = a.B * tiles_per_cloud;
  for (int t = blockIdx.x * W + wave_index(); t < ntiles; t += gridDim.x * W) {
    const int b = t / tiles_per_cloud;
    const int q = (t - b * tiles_per_cloud) * 16 + j;
    const bool valid = q < a.S;
    const unsigned row = (unsigned)b * (unsigned)a.S + (unsigned)(valid ? q : a.S - 1);
    const float *centre = at32(a.xyz1, mul24(row, 12u));
    f32x4 mrun[4], drun[4], nrun[4];
    // a real loop over the passes (not unrolled): the scheduler must not pull a later pass's gathers forward
    auto run_pass = [&](auto first_tag, int pass) {
      constexpr bool FIRST = decltype(first_tag)::value;
      f32x4 geo[1][P], cat[8][P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const unsigned slot = mul24(row, 6u) + (unsigned)(P * pass + p);
        const int nbr = *at32(a.idx, slot * 4u);
        const unsigned src = (unsigned)b * (unsigned)a.N + (unsigned)nbr;
        geo[0][p] = geometry_block(centre, at32(a.xyz2, mul24(src, 12u)), g);
        load_row_blocks<4>(&cat[4][p], P, at32(a.pix, slot << 8), g);
      }
      f32x4 enc[4][P], h1[8][P], h2[4][P];
      mlp_layer<1, 4, P, true>(enc, geo, lds_w, lane);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int p = 0; p < P; ++p) cat[m][p] = enc[m][p];
      mlp_layer<8, 8, P, true>(h1, cat, lds_w + WX, lane);
      mlp_layer<8, 4, P, true>(h2, h1, lds_w + WX + W1, lane);
#pragma unroll
      for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float x0 = h2[o][0][c], x1 = h2[o][1][c];                     // post-ReLU: >= 0
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
    if (valid) {
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        f32x4 res;
#pragma unroll
        for (int c = 0; c < 4; ++c) res[c] = div_ge1(nrun[o][c], drun[o][c]);
        *reinterpret_cast<f32x4 *>(at32(a.out, (row << 8) + 64u * o + 16u * (unsigned)g)) = res;
      }
    }
  }
}

// b: enc2 = mlp_conv_xyz_2(geo'); w = softmax_k(mlp3_convs([enc2 | centre feat | gathered first]));
//    out = sum_k w * gathered first.  Candidates = the frame-1 points themselves (N == S).
template <int CB, int KP, int P, int W>
__global__ __launch_bounds__(W * 64, (W <= 4 ? 2 : 1)) void cv_b_kernel(CVArgs a) {
  constexpr int NBI = 4 + CB + 4, C = 16 * CB;
  constexpr int WX = layer_floats(1, 4), W1 = layer_floats(NBI, 8), W2 = layer_floats(8, 4);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, WX + W1 + W2);
  __syncthreads();
  stagger_start(threadIdx.x >> 6, (4 + NBI * 8 + 32) * 4 * P, a.stagger);
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  PWCLO_TILE_LOOP(KP, P, a.S, a.B) {
    const int b = t / tiles_per_cloud;
    const int pix0 = (t - b * tiles_per_cloud) * TILE;
    f32x4 geo[1][P], cat[NBI][P];
    int sq[P];
    bool padded[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const PixelMap<KP> pm(pix0 + 16 * p + j);
      const bool valid = pm.s < a.S;
      const int s = valid ? pm.s : a.S - 1;
      padded[p] = pm.k >= a.K;
      const int k = padded[p] ? 0 : pm.k;
      sq[p] = valid ? s : -1;
      const int nbr = a.idx[((size_t)b * a.S + s) * a.K + k];
      geo[0][p] = geometry_block(a.xyz1 + ((size_t)b * a.S + s) * 3, a.xyz2 + ((size_t)b * a.N + nbr) * 3, g);
      load_row_blocks<CB>(&cat[4][p], P, a.feat1 + ((size_t)b * a.S + s) * C, g);
      load_row_blocks<4>(&cat[4 + CB][p], P, a.feat2 + ((size_t)b * a.N + nbr) * 64, g);
    }
    f32x4 enc[4][P];
    mlp_layer<1, 4, P, true>(enc, geo, lds_w, lane);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int p = 0; p < P; ++p) cat[m][p] = enc[m][p];
    f32x4 h1[8][P], h2[4][P];
    mlp_layer<NBI, 8, P, true>(h1, cat, lds_w + WX, lane);
    mlp_layer<8, 4, P, true>(h2, h1, lds_w + WX + W1, lane);
    constexpr int GROUP = KP < 16 ? KP : 16;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      f32x4 res[P];
      softmax_weighted_sum<KP, P>(res, h2[o], cat[4 + CB + o], padded);
#pragma unroll
      for (int p = 0; p < P; ++p)
        if ((j & (GROUP - 1)) == 0 && sq[p] >= 0)
          *reinterpret_cast<f32x4 *>(a.out + ((size_t)b * a.S + sq[p]) * 64 + 16 * o + 4 * g) = res[p];
    }
  }
}

// ---- pose head pooling: out[b,c] = sum_n emb[b,n,c] * softmax_n(mask[b,n,c]) ----------------------
// (F.softmax(mask, dim=2) + PoseCalculator's masked sum, PW/pose_calculator.py:58; pwclo_net.py:172)
constexpr int MP_PARTS = 16;  // 1024 threads: 64 channels x 16 point slices
__global__ __launch_bounds__(64 * MP_PARTS) void masked_pool_kernel(int n, const float *__restrict__ emb,
                                                                    const float *__restrict__ mask,
                                                                    float *__restrict__ out) {
  __shared__ float red[MP_PARTS][64];
  __shared__ float red2[MP_PARTS][64];
  const int c = threadIdx.x & 63, part = threadIdx.x >> 6, b = blockIdx.x;
  const float *e = emb + (size_t)b * n * 64, *m = mask + (size_t)b * n * 64;
  float mx = -INFINITY;
  // one coalesced 256-byte row per wave and trip: unrolled so that 8 loads are in flight per thread
#pragma unroll 8
  for (int i = part; i < n; i += MP_PARTS) mx = fmaxf(mx, m[(size_t)i * 64 + c]);
  red[part][c] = mx;
  __syncthreads();
  mx = red[0][c];
#pragma unroll
  for (int q = 1; q < MP_PARTS; ++q) mx = fmaxf(mx, red[q][c]);
  __syncthreads();
  float den = 0.f, num = 0.f;
#pragma unroll 8
  for (int i = part; i < n; i += MP_PARTS) {
    const float ex = expf(m[(size_t)i * 64 + c] - mx);
    den += ex;
    num += ex * e[(size_t)i * 64 + c];
  }
  red[part][c] = den;
  red2[part][c] = num;
  __syncthreads();
  if (part == 0) {
    float d = 0.f, s = 0.f;
#pragma unroll
    for (int q = 0; q < MP_PARTS; ++q) { d += red[q][c]; s += red2[q][c]; }
    out[b * 64 + c] = s / d;
  }
}

// ---- fused pose head ------------------------------------------------------------------------------------
// One workgroup per cloud does everything between the last per-point tensor and the pose:
//   pooled = sum_n emb * softmax_n(mask)                                  (pose_calculator.py:58)
//   big = W_qt pooled + b_qt (64 -> 256); q_det = normalise(W_q big + b_q); t_det = W_t big + b_t
//   level 4 (q_prev == nullptr): q = q_det, t = t_det                      (pwclo_net.py:174)
//   refinement levels: q = q_det (x) q_coarse, t = q_det (x) (0,t_coarse) (x) q_det^-1 + t_det
//                                                                          (pose_warp_refinement.py:139,148)
//   pose_row = [t, q / (sqrt(|q|^2 + 1e-10) + 1e-10)]                       (pwclo_net.py:195-205)
// Replaces ~25 tiny torch launches per level.  Dropout is the identity in eval mode.
struct PoseHeadArgs {
  const float *emb, *mask;          // (B,N,64) point-major
  const float *w_qt, *b_qt;         // (256,64), (256)
  const float *w_q, *b_q;           // (4,256), (4)
  const float *w_t, *b_t;           // (3,256), (3)
  const float *q_prev, *t_prev;     // (B,4), (B,3) or nullptr
  float *q_out, *t_out;             // (B,4), (B,3)
  float *pose_row;                  // (B, row_stride) base of this level's row; 7 floats written
  int n, row_stride;
  // optional (round 3): the NEXT level's first step, warp(xyz, q, t) of PW/pose_warp_refinement.py:104-106 on the
  // finer cloud with the pose this launch has just composed -- the same expressions as csrc/warp.hip quat_warp_kernel
  // (bit-identical output), one launch fewer per level
  const float *warp_src;            // (B, warp_n, 3) point-major, or nullptr
  float *warp_out;                  // (B, warp_n, 3)
  int warp_n;
};

__device__ __forceinline__ void quat_mul(const float *a, const float *b, float *r) {
  r[0] = ((a[0] * b[0] - a[1] * b[1]) - a[2] * b[2]) - a[3] * b[3];
  r[1] = ((a[0] * b[1] + a[1] * b[0]) + a[2] * b[3]) - a[3] * b[2];
  r[2] = ((a[0] * b[2] - a[1] * b[3]) + a[2] * b[0]) + a[3] * b[1];
  r[3] = ((a[0] * b[3] + a[1] * b[2]) - a[2] * b[1]) + a[3] * b[0];
}

__global__ __launch_bounds__(64 * MP_PARTS) void pose_head_kernel(PoseHeadArgs a) {
  TraceScope trace_scope_(TK_POSE_HEAD);
  // Pooling layout: a lane owns 4 channels (16-byte loads) of every 4th point of its wave's slice, so a
  // wave-load covers 4 whole 256-byte rows; 64 (wave, point-slot) partials per channel meet in LDS.
  __shared__ float red[4 * MP_PARTS][64];
  __shared__ float red2[4 * MP_PARTS][64];
  __shared__ float chmax[64];
  __shared__ float pooled[64];
  __shared__ float big[256];
  __shared__ float qt[8];
  __shared__ float pose_s[8];
  const int c = threadIdx.x & 63, part = threadIdx.x >> 6, b = blockIdx.x, n = a.n;
  const int c4 = 4 * (c & 15), sub = c >> 4, slot = 4 * part + sub;
  const float *e = a.emb + (size_t)b * n * 64 + c4, *m = a.mask + (size_t)b * n * 64 + c4;
  f32x4 mx = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll 4
  for (int i = slot; i < n; i += 4 * MP_PARTS) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(m + (size_t)i * 64);
    mx.x = fmaxf(mx.x, v.x); mx.y = fmaxf(mx.y, v.y); mx.z = fmaxf(mx.z, v.z); mx.w = fmaxf(mx.w, v.w);
  }
  *reinterpret_cast<f32x4 *>(&red[slot][c4]) = mx;
  __syncthreads();
  if (part == 0) {
    float v = red[0][c];
    for (int q = 1; q < 4 * MP_PARTS; ++q) v = fmaxf(v, red[q][c]);
    chmax[c] = v;
  }
  __syncthreads();
  const f32x4 cm = *reinterpret_cast<const f32x4 *>(&chmax[c4]);
  f32x4 den = {0.f, 0.f, 0.f, 0.f}, num = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int i = slot; i < n; i += 4 * MP_PARTS) {
    const f32x4 mv = *reinterpret_cast<const f32x4 *>(m + (size_t)i * 64);
    const f32x4 ev = *reinterpret_cast<const f32x4 *>(e + (size_t)i * 64);
    const float e0 = expf(mv.x - cm.x), e1 = expf(mv.y - cm.y), e2 = expf(mv.z - cm.z), e3 = expf(mv.w - cm.w);
    den.x += e0; den.y += e1; den.z += e2; den.w += e3;
    num.x += e0 * ev.x; num.y += e1 * ev.y; num.z += e2 * ev.z; num.w += e3 * ev.w;
  }
  *reinterpret_cast<f32x4 *>(&red[slot][c4]) = den;
  *reinterpret_cast<f32x4 *>(&red2[slot][c4]) = num;
  __syncthreads();
  if (part == 0) {
    float d = 0.f, s_ = 0.f;
    for (int q = 0; q < 4 * MP_PARTS; ++q) { d += red[q][c]; s_ += red2[q][c]; }
    pooled[c] = s_ / d;
  }
  __syncthreads();
  if (threadIdx.x < 256) {                       // conv1d_q_t: 64 -> 256
    const float *w = a.w_qt + threadIdx.x * 64;
    float acc = a.b_qt[threadIdx.x];
#pragma unroll 16
    for (int k = 0; k < 64; ++k) acc += w[k] * pooled[k];
    big[threadIdx.x] = acc;
  }
  __syncthreads();
  if (part < 7) {                                // conv1d_q (4 rows) and conv1d_t (3 rows): 256 -> 1 each
    const float *w = part < 4 ? a.w_q + part * 256 : a.w_t + (part - 4) * 256;
    float acc = 0.f;
    for (int k = c; k < 256; k += 64) acc += w[k] * big[k];
    acc = wave_allreduce_f32(acc, [](float x, float y) { return x + y; });
    if (c == 0) qt[part] = acc + (part < 4 ? a.b_q[part] : a.b_t[part - 4]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float qd[4] = {qt[0], qt[1], qt[2], qt[3]}, td[3] = {qt[4], qt[5], qt[6]};
    const float nq = sqrtf((((qd[0] * qd[0] + qd[1] * qd[1]) + qd[2] * qd[2]) + qd[3] * qd[3]) + 1e-10f) + 1e-10f;
    for (int i = 0; i < 4; ++i) qd[i] = qd[i] / nq;
    float q[4], t[3];
    if (a.q_prev) {
      const float *qc = a.q_prev + b * 4, *tc = a.t_prev + b * 3;
      quat_mul(qd, qc, q);
      const float q2 = (((qd[0] * qd[0] + qd[1] * qd[1]) + qd[2] * qd[2]) + qd[3] * qd[3]) + 1e-10f;
      const float qi[4] = {qd[0] / q2, (qd[1] * -1.0f) / q2, (qd[2] * -1.0f) / q2, (qd[3] * -1.0f) / q2};
      const float p[4] = {0.f, tc[0], tc[1], tc[2]};
      float r1[4], r2[4];
      quat_mul(qd, p, r1);
      quat_mul(r1, qi, r2);
      t[0] = r2[1] + td[0]; t[1] = r2[2] + td[1]; t[2] = r2[3] + td[2];
    } else {
      for (int i = 0; i < 4; ++i) q[i] = qd[i];
      for (int i = 0; i < 3; ++i) t[i] = td[i];
    }
    for (int i = 0; i < 4; ++i) { a.q_out[b * 4 + i] = q[i]; pose_s[i] = q[i]; }
    for (int i = 0; i < 3; ++i) { a.t_out[b * 3 + i] = t[i]; pose_s[4 + i] = t[i]; }
    const float nn = sqrtf((((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]) + 1e-10f) + 1e-10f;
    float *row = a.pose_row + (size_t)b * a.row_stride;
    row[0] = t[0]; row[1] = t[1]; row[2] = t[2];
    for (int i = 0; i < 4; ++i) row[3 + i] = q[i] / nn;
  }
  if (a.warp_src != nullptr) {                   // workgroup-uniform
    __syncthreads();
    const float qw = pose_s[0], qx = pose_s[1], qy = pose_s[2], qz = pose_s[3];
    const float tx = pose_s[4], ty = pose_s[5], tz = pose_s[6];
    // inv_q (PWCLO_utils.py:31-39): conj(q) / (sum(q*q) + 1e-10)
    const float q2 = (((qw * qw + qx * qx) + qy * qy) + qz * qz) + 1e-10f;
    const float iw = qw / q2, ix = (qx * -1.0f) / q2, iy = (qy * -1.0f) / q2, iz = (qz * -1.0f) / q2;
    const float *src = a.warp_src + (size_t)b * 3 * a.warp_n;
    float *dst = a.warp_out + (size_t)b * 3 * a.warp_n;
    for (int j = threadIdx.x; j < a.warp_n; j += 64 * MP_PARTS) {
      const float pq[4] = {qw, qx, qy, qz}, pp[4] = {0.0f, src[3 * j], src[3 * j + 1], src[3 * j + 2]};
      const float pi[4] = {iw, ix, iy, iz};
      float r1[4], r2[4];
      quat_mul(pq, pp, r1);                     // same left-to-right component sums as warp.hip: hamilton()
      quat_mul(r1, pi, r2);
      dst[3 * j] = r2[1] + tx;
      dst[3 * j + 1] = r2[2] + ty;
      dst[3 * j + 2] = r2[3] + tz;
    }
  }
}

// ---- launch helpers ------------------------------------------------------------------------------------
static int fl_tuning(const char *name, int dflt) {   // PWCLO_FL_<NAME> overrides (experiments)
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}

// Largest launch (in wave tiles) that still takes the 4-wave workgroup variants of the point-wise / coarse-level kernels.  Narrow
// workgroups reach more CUs -- lower latency of a lone forward (batch 1: +7 %, batch 4: +5 %) -- but every workgroup stages the
// stack's 100-160 KB of weights again, which costs CU-time: in the pipelined batch-32 run the launches of exactly 2048 tiles
// (level-2 flow predictors) are better off wide (+0.3 %, profiles/r03/r03_v7_ab_coarse_w4.txt); smaller ones stay narrow.
static int coarse_tiles() {
  static const int v = fl_tuning("PWCLO_COARSE_W4_TILES", 2047);
  return v;
}

template <int W, typename Kern, typename Args>
static void launch_persistent(Kern kern, bool &attr_set, int lds_bytes, long long ntiles, const Args &a) {
  if (lds_bytes > 64 * 1024 && !attr_set) {
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  // once per kernel: the largest any configuration can ask for
    attr_set = true;
  }
  // Workgroups beyond one resident set queue behind it; >1 "rounds" keeps the kernel balanced when
  // part of the chip is held by another stream's kernels (e.g. the other in-flight batch's FPS).
  static const int rounds = fl_tuning("PWCLO_FL_ROUNDS", 1);
  const int per_cu = (lds_bytes > 80 * 1024 || W > 8) ? 1 : 2;
  long long grid = (ntiles + W - 1) / W;
  if (grid > 256LL * per_cu * rounds) grid = 256LL * per_cu * rounds;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(W * 64), lds_bytes, current_stream(), a);
}

static long long tiles_of(int b, int s, int kp, int p) {
  return (long long)b * (((long long)s * kp + 16 * p - 1) / (16 * p));
}

}  // namespace pwclo

using namespace pwclo;

extern "C" void upconv_fused_kernel_wrapper(int b, int n, int s, int k, const float *xyz2,
                                            const float *xyz1, const float *feat1, const int *idx,
                                            const float *packed_w, float *out) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(k >= 1 && k <= 8, "upconv_fused: nsample=%d outside [1,8]", k);
  UpconvArgs a{xyz2, xyz1, feat1, idx, packed_w, out, b, n, s, k, fl_tuning("PWCLO_FL_STAGGER", 0)};
  static bool attr = false, attr1 = false;
  static const int wide = fl_tuning("PWCLO_FL_WIDE", 1);
  constexpr int lds = 4 * (layer_floats(5, 8) + layer_floats(8, 4));
  if (wide) launch_persistent<16>(upconv_kernel<8, 1, 16>, attr1, lds, tiles_of(b, s, 8, 1), a);
  else launch_persistent<8>(upconv_kernel<8, 2, 8>, attr, lds, tiles_of(b, s, 8, 2), a);
  check_launch("upconv_fused");
}

extern "C" void pointwise_fused_kernel_wrapper(int b, int s, int c0, int c1, int c2, int w1, int w2,
                                               const float *src0, const float *src1, const float *src2,
                                               const float *packed_w, float *out) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(rows_fit_32bit((long long)b * s), "pointwise_fused: batch too large for 32-bit offsets (b=%d)", b);
  PointwiseArgs a{{src0, src1, src2}, packed_w, out, b, s, fl_tuning("PWCLO_FL_STAGGER", 0), nullptr, nullptr};
#define PW_CASE(C0, C1, C2, A1, A2)                                                                 \
  if (c0 == C0 && c1 == C1 && c2 == C2 && w1 == A1 && w2 == A2) {                                   \
    static bool attr = false, attr1 = false, attr4 = false;                                         \
    static const int wide = fl_tuning("PWCLO_FL_WIDE", 1);                                          \
    constexpr int NBI = (C0 + C1 + C2) / 16;                                                        \
    constexpr int lds = 4 * (layer_floats(NBI, A1 / 16) + (A2 > 0 ? layer_floats(A1 / 16, A2 / 16) : 0)); \
    /* few tiles (coarse levels): 4-wave workgroups spread them over 4x more CUs; a 16-wave        \
       workgroup would run 4 tiles back to back on each SIMD while most of the chip idles */        \
    if (wide && fl_tuning("PWCLO_COARSE_W4", 1) && tiles_of(b, s, 1, 1) <= coarse_tiles())                  \
      launch_persistent<4>(pointwise_kernel<C0 / 16, C1 / 16, C2 / 16, A1 / 16, A2 / 16, 1, 4>,      \
                           attr4, lds, tiles_of(b, s, 1, 1), a);                                    \
    else if (wide) launch_persistent<16>(pointwise_kernel<C0 / 16, C1 / 16, C2 / 16, A1 / 16, A2 / 16, 1, 16>, \
                                    attr1, lds, tiles_of(b, s, 1, 1), a);                           \
    else launch_persistent<8>(pointwise_kernel<C0 / 16, C1 / 16, C2 / 16, A1 / 16, A2 / 16, 2, 8>, attr, lds, \
                              tiles_of(b, s, 1, 2), a);                                             \
    check_launch("pointwise_fused");                                                                \
    return;                                                                                         \
  }
  PW_CASE(64, 64, 0, 64, 0)      // set-upconv post_mlp, level 3: [64 | 64] -> 64
  PW_CASE(64, 32, 0, 64, 0)      // level 2
  PW_CASE(64, 16, 0, 64, 0)      // level 1
  PW_CASE(64, 64, 64, 128, 64)   // flow predictors, level 3: [64|64|64] -> 128 -> 64
  PW_CASE(32, 64, 64, 128, 64)   // features predictor, level 2: [C|64|64]
  PW_CASE(64, 64, 32, 128, 64)   // mask predictor, level 2: [64|64|C]
  PW_CASE(16, 64, 64, 128, 64)   // features predictor, level 1
  PW_CASE(128, 64, 0, 128, 64)   // l4_flow_predictor: [128|64] -> 128 -> 64
#undef PW_CASE
  set_error(PWCLO_EINVAL, "pointwise_fused: no kernel for sources (%d,%d,%d) widths (%d,%d)", c0, c1, c2,
            w1, w2);
}

// The same stack followed by one linear layer (w2 -> wt channels, no activation) written to out_tail (b,s,wt): the
// hoisted partial product a consumer would otherwise request from linear_jobs_kernel_wrapper (see pointwise_kernel).
extern "C" void pointwise_tail_fused_kernel_wrapper(int b, int s, int c0, int c1, int c2, int w1, int w2, int wt,
                                                    const float *src0, const float *src1, const float *src2,
                                                    const float *packed_w, const float *packed_tail, float *out,
                                                    float *out_tail, int tail_floats) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(rows_fit_32bit((long long)b * s), "pointwise_tail_fused: batch too large for 32-bit offsets (b=%d)", b);
  PWCLO_REQUIRE(packed_tail != nullptr && out_tail != nullptr, "pointwise_tail_fused: the tail needs its packed layer and an output");
  PointwiseArgs a{{src0, src1, src2}, packed_w, out, b, s, 0, packed_tail, out_tail};
#define PWT_CASE(C0, C1, C2, A1, A2, AT)                                                            \
  if (c0 == C0 && c1 == C1 && c2 == C2 && w1 == A1 && w2 == A2 && wt == AT) {                       \
    static bool attr16 = false, attr4 = false;                                                      \
    constexpr int NBI = (C0 + C1 + C2) / 16;                                                        \
    constexpr int lds = 4 * (layer_floats(NBI, A1 / 16) + layer_floats(A1 / 16, A2 / 16) + layer_floats(A2 / 16, AT / 16)); \
    static_assert(lds <= 160 * 1024, "stack and tail must fit the 160 KiB of LDS");                 \
    PWCLO_REQUIRE(tail_floats == layer_floats(A2 / 16, AT / 16), "pointwise_tail_fused: packed tail holds %d floats, needs %d", \
                  tail_floats, layer_floats(A2 / 16, AT / 16));                                     \
    if (fl_tuning("PWCLO_COARSE_W4", 1) && tiles_of(b, s, 1, 1) <= coarse_tiles())                        \
      launch_persistent<4>(pointwise_kernel<C0 / 16, C1 / 16, C2 / 16, A1 / 16, A2 / 16, 1, 4, AT / 16>, \
                           attr4, lds, tiles_of(b, s, 1, 1), a);                                    \
    else launch_persistent<16>(pointwise_kernel<C0 / 16, C1 / 16, C2 / 16, A1 / 16, A2 / 16, 1, 16, AT / 16>, \
                               attr16, lds, tiles_of(b, s, 1, 1), a);                               \
    check_launch("pointwise_tail_fused");                                                           \
    return;                                                                                         \
  }
  PWT_CASE(32, 64, 64, 128, 64, 128)   // features predictor, level 2 (+ level 1's set-upconv seeds)
  PWT_CASE(64, 64, 32, 128, 64, 128)   // mask predictor, level 2
#undef PWT_CASE
  set_error(PWCLO_EINVAL, "pointwise_tail_fused: no kernel for sources (%d,%d,%d) widths (%d,%d) tail %d", c0, c1, c2,
            w1, w2, wt);
}

extern "C" void cv_fused_a1_kernel_wrapper(int b, int n, int s, int k, int c, const float *xyz1,
                                           const float *feat1, const float *xyz2, const float *feat2,
                                           const int *idx, const float *packed_w, float *pix, int pix_slots) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(k >= 1 && k <= 32, "cv_fused_a1: nsample_q=%d outside [1,32]", k);
  PWCLO_REQUIRE(cv_pix_slots_valid(k, pix_slots), "cv_fused_a1: pix_slots=%d invalid for nsample_q=%d", pix_slots, k);
  CVArgs a{xyz1, feat1, xyz2, feat2, idx, packed_w, pix, nullptr, b, n, s, k, fl_tuning("PWCLO_FL_STAGGER", 0)};
  const int kp = pix_slots;
#define A1_CASE(C, KP)                                                                              \
  if (c == C && kp == KP) {                                                                         \
    static bool attr = false, attr1 = false;                                                        \
    static const int wide = fl_tuning("PWCLO_FL_WIDE", 1);                                          \
    constexpr int lds = 4 * (layer_floats(1 + 2 * (C / 16), 8) + layer_floats(8, 4) + layer_floats(4, 4)); \
    if (wide && KP <= 16) launch_persistent<16>(cv_a1_kernel<C / 16, KP, 1, 16>, attr1, lds, tiles_of(b, s, KP, 1), a); \
    else launch_persistent<8>(cv_a1_kernel<C / 16, KP, 2, 8>, attr, lds, tiles_of(b, s, KP, 2), a);  \
    check_launch("cv_fused_a1");                                                                    \
    return;                                                                                         \
  }
  A1_CASE(64, 32) A1_CASE(64, 8) A1_CASE(32, 8) A1_CASE(16, 8) A1_CASE(64, 6) A1_CASE(32, 6) A1_CASE(16, 6)
#undef A1_CASE
  set_error(PWCLO_EINVAL, "cv_fused_a1: no kernel for c=%d nsample_q=%d", c, k);
}

extern "C" void cv_fused_a2_kernel_wrapper(int b, int n, int s, int k, const float *xyz1,
                                           const float *xyz2, const int *idx, const float *packed_w,
                                           const float *pix, float *out, int pix_slots, int wfmt,
                                           int packed_floats) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(k >= 1 && k <= 32, "cv_fused_a2: nsample_q=%d outside [1,32]", k);
  PWCLO_REQUIRE(cv_pix_slots_valid(k, pix_slots), "cv_fused_a2: pix_slots=%d invalid for nsample_q=%d", pix_slots, k);
  PWCLO_REQUIRE(rows_fit_32bit((long long)b * max(n, s * 32)), "cv_fused_a2: batch too large for 32-bit offsets (b=%d)", b);
  CVArgs a{xyz1, nullptr, xyz2, nullptr, idx, packed_w, const_cast<float *>(pix), out, b, n, s, k,
           fl_tuning("PWCLO_FL_STAGGER", 0)};
  const int kp = pix_slots;
  constexpr int lds = 4 * (layer_floats(1, 4) + layer_floats(8, 8) + layer_floats(8, 4));
  {
    constexpr int lds3c = 4 * (layer_floats(1, 4) + layer_floats_bf3(8, 8) + layer_floats_bf3(8, 4));
    constexpr int lds2c = 4 * (layer_floats(1, 4) + layer_floats_bf16(8, 8) + layer_floats_bf16(8, 4));
    PWCLO_REQUIRE(wfmt == PWCLO_WFMT_F32 || kp == 6 || kp == 32,
                  "cv_fused_a2: the reduced formats exist for 6 / 32 pixel slots only (got %d)", kp);
    PWCLO_REQUIRE_PACKED("cv_fused_a2", wfmt, packed_floats, lds / 4, lds3c / 4, lds2c / 4);
  }
  static bool attr32 = false, attr16 = false, attr8 = false, attr16w = false, attr8w = false, attr6 = false;
  static const int wide = fl_tuning("PWCLO_FL_WIDE", 1);
  static bool attr6s = false;
  const long long t6 = (long long)b * ((s + 7) / 8);
  if (wfmt == PWCLO_WFMT_BF16X3) {     // opt-in split path (mlp_core.hpp)
    constexpr int lds3 = 4 * (layer_floats(1, 4) + layer_floats_bf3(8, 8) + layer_floats_bf3(8, 4));
    static bool b6 = false, b6s = false, b32 = false;
    if (kp == 6 && t6 <= 2048) launch_persistent<4>(cv_a2_dense6_kernel<4, 1>, b6s, lds3, t6, a);
    else if (kp == 6) launch_persistent<8>(cv_a2_dense6_kernel<8, 1>, b6, lds3, t6, a);
    else launch_persistent<8>(cv_a2_kernel<32, 2, 8, 1>, b32, lds3, tiles_of(b, s, 32, 2), a);
    check_launch("cv_fused_a2");
    return;
  }
  if (wfmt == PWCLO_WFMT_BF16) {
    constexpr int lds2 = 4 * (layer_floats(1, 4) + layer_floats_bf16(8, 8) + layer_floats_bf16(8, 4));
    static bool c6 = false, c6s = false, c32 = false;
    if (kp == 6 && t6 <= 2048) launch_persistent<4>(cv_a2_dense6_kernel<4, 2>, c6s, lds2, t6, a);
    else if (kp == 6) launch_persistent<8>(cv_a2_dense6_kernel<8, 2>, c6, lds2, t6, a);
    else launch_persistent<8>(cv_a2_kernel<32, 2, 8, 2>, c32, lds2, tiles_of(b, s, 32, 2), a);
    check_launch("cv_fused_a2");
    return;
  }
  static const int lane6 = fl_tuning("PWCLO_LANE6", 1);
  static bool attrl = false;
  const long long t16 = (long long)b * ((s + 15) / 16);
  // in-lane softmax for the large levels; a coarse level has too few 16-query tiles to fill the chip (measured:
  // 48 us against 31 us for the dense-6 kernel on 4-wave workgroups at S = 256)
  if (kp == 6 && lane6 && t16 > 1024) launch_persistent<8>(cv_a2_lane6_kernel<8>, attrl, lds, t16, a);
  else if (kp == 6 && t6 <= 2048 && fl_tuning("PWCLO_COARSE_W4", 1))   // coarse level: 4-wave workgroups reach twice as many CUs
    launch_persistent<4>(cv_a2_dense6_kernel<4>, attr6s, lds, t6, a);
  else if (kp == 6) launch_persistent<8>(cv_a2_dense6_kernel<8>, attr6, lds, t6, a);
  else if (kp == 32) launch_persistent<8>(cv_a2_kernel<32, 2, 8>, attr32, lds, tiles_of(b, s, 32, 2), a);
  else if (kp == 16 && wide) launch_persistent<16>(cv_a2_kernel<16, 1, 16>, attr16w, lds, tiles_of(b, s, 16, 1), a);
  else if (kp == 16) launch_persistent<8>(cv_a2_kernel<16, 2, 8>, attr16, lds, tiles_of(b, s, 16, 2), a);
  else if (wide) launch_persistent<16>(cv_a2_kernel<8, 1, 16>, attr8w, lds, tiles_of(b, s, 8, 1), a);
  else launch_persistent<8>(cv_a2_kernel<8, 2, 8>, attr8, lds, tiles_of(b, s, 8, 2), a);
  check_launch("cv_fused_a2");
}

extern "C" void cv_fused_b_kernel_wrapper(int b, int s, int k, int c, const float *xyz1,
                                          const float *feat1, const float *first, const int *idx,
                                          const float *packed_w, float *out) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(k >= 1 && k <= 4, "cv_fused_b: nsample=%d outside [1,4]", k);
  CVArgs a{xyz1, feat1, xyz1, first, idx, packed_w, nullptr, out, b, s, s, k, fl_tuning("PWCLO_FL_STAGGER", 0)};
#define B_CASE(C)                                                                                   \
  if (c == C) {                                                                                     \
    static bool attr = false, attr1 = false;                                                        \
    static const int wide = fl_tuning("PWCLO_FL_WIDE", 1);                                          \
    constexpr int lds = 4 * (layer_floats(1, 4) + layer_floats(8 + C / 16, 8) + layer_floats(8, 4)); \
    if (wide) launch_persistent<16>(cv_b_kernel<C / 16, 4, 1, 16>, attr1, lds, tiles_of(b, s, 4, 1), a); \
    else launch_persistent<8>(cv_b_kernel<C / 16, 4, 2, 8>, attr, lds, tiles_of(b, s, 4, 2), a);     \
    check_launch("cv_fused_b");                                                                     \
    return;                                                                                         \
  }
  B_CASE(64) B_CASE(32) B_CASE(16)
#undef B_CASE
  set_error(PWCLO_EINVAL, "cv_fused_b: no kernel for c=%d", c);
}

extern "C" void masked_pool_kernel_wrapper(int b, int n, const float *emb, const float *mask, float *out) {
  if (b <= 0 || n <= 0) return;
  hipLaunchKernelGGL(masked_pool_kernel, dim3(b), dim3(64 * MP_PARTS), 0, current_stream(), n, emb, mask, out);
  check_launch("masked_pool");
}

extern "C" void pose_head_fused_kernel_wrapper(int b, int n, const float *emb, const float *mask,
                                               const float *w_qt, const float *b_qt, const float *w_q,
                                               const float *b_q, const float *w_t, const float *b_t,
                                               const float *q_prev, const float *t_prev, float *q_out,
                                               float *t_out, float *pose_row, int row_stride) {
  if (b <= 0 || n <= 0) return;
  PoseHeadArgs a{emb, mask, w_qt, b_qt, w_q, b_q, w_t, b_t, q_prev, t_prev, q_out, t_out, pose_row, n, row_stride,
                 nullptr, nullptr, 0};
  hipLaunchKernelGGL(pose_head_kernel, dim3(b), dim3(64 * MP_PARTS), 0, current_stream(), a);
  check_launch("pose_head_fused");
}

extern "C" void pose_head_warp_fused_kernel_wrapper(int b, int n, const float *emb, const float *mask,
                                                    const float *w_qt, const float *b_qt, const float *w_q,
                                                    const float *b_q, const float *w_t, const float *b_t,
                                                    const float *q_prev, const float *t_prev, float *q_out,
                                                    float *t_out, float *pose_row, int row_stride, int warp_n,
                                                    const float *warp_src, float *warp_out) {
  if (b <= 0 || n <= 0) return;
  PWCLO_REQUIRE(warp_n > 0 && warp_src != nullptr && warp_out != nullptr, "pose_head_warp_fused: needs a cloud to warp (n=%d)", warp_n);
  PoseHeadArgs a{emb, mask, w_qt, b_qt, w_q, b_q, w_t, b_t, q_prev, t_prev, q_out, t_out, pose_row, n, row_stride,
                 warp_src, warp_out, warp_n};
  hipLaunchKernelGGL(pose_head_kernel, dim3(b), dim3(64 * MP_PARTS), 0, current_stream(), a);
  check_launch("pose_head_warp_fused");
}
