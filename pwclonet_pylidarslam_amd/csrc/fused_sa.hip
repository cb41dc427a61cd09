// Fused set-abstraction layer (eval mode): neighbour gather -> centre subtraction -> 3-layer
// shared MLP on the fp32 matrix cores -> max over the K neighbours.
//
// Replaces, for PointnetSAModulePWCLONet.forward (P2/pointnet2_modules.py:205-243), the chain
// grouping_operation x2 -> permute/tile/sub -> cat -> 3x(Conv2d, BatchNorm2d, ReLU) -> max_pool2d
// (>= 11 launches and >= 90 MB of materialised (B,C,S,K) activations per pair)
// by ONE kernel whose HBM traffic is the algorithmic minimum: indices + gathered rows in,
// (B,S,Cout) out.  Feature tensors are point-major (B,N,C) so a neighbour's channels are one
// contiguous row (16-byte gathers).  See mlp_core.hpp for the register/MFMA layout.
#include <stdlib.h>

#include "mlp_core.hpp"

namespace pwclo {

struct SAArgs {
  const float *xyz;      // (B,N,3) source points
  const float *new_xyz;  // (B,S,3) query points (FPS samples)
  const float *feat;     // (B,N,16*CFB) point-major features, or nullptr when CFB == 0
  const int *idx;        // (B,S,K) neighbour lists
  const float *w;        // packed layers 1..3 (fused.py: pack_layer), consecutive
  float *out;            // (B,S,16*B3)
  int B, N, S, K;
  int stagger;
};

constexpr int SA_WAVES = 8;

// CFB: gathered feature blocks (16 channels each); B1..B3: output blocks of the three layers;
// KP: K padded to a power of two (pixels per query); P: 16-pixel blocks per wave and tile.
// XYZ_ONLY (level 0, features None): input = [xyz_diff(3), grouped_xyz(3)] (:224-233), otherwise
// [xyz_diff(3) | gathered features] (:209-222).  Physical channel order: block 0 = geometry
// (dx,dy,dz[,qx,qy,qz], zero padded to 16), blocks 1.. = features.
template <int CFB, int B1, int B2, int B3, int KP, int P, bool XYZ_ONLY>
__global__ __launch_bounds__(SA_WAVES * 64) void sa_kernel(SAArgs a) {
  constexpr int NBI = 1 + CFB;
  constexpr int W1 = layer_floats(NBI, B1), W2 = layer_floats(B1, B2), W3 = layer_floats(B2, B3);
  extern __shared__ __attribute__((aligned(16))) float lds_w[];
  stage_weights(lds_w, a.w, W1 + W2 + W3);
  __syncthreads();
  stagger_start(threadIdx.x >> 6, (NBI * B1 + B1 * B2 + B2 * B3) * 4 * P, a.stagger);

  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int wave = threadIdx.x >> 6;
  constexpr int TILE = 16 * P;                       // pixels per wave-tile
  const int pix_per_cloud = a.S * KP;
  const int tiles_per_cloud = (pix_per_cloud + TILE - 1) / TILE;
  const int ntiles = a.B * tiles_per_cloud;
  constexpr int CF = 16 * CFB, C3 = 16 * B3;

  for (int t = blockIdx.x * SA_WAVES + wave; t < ntiles; t += gridDim.x * SA_WAVES) {
    const int b = t / tiles_per_cloud;
    const int pix0 = (t - b * tiles_per_cloud) * TILE;
    f32x4 in[NBI][P];
    int sq[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const PixelMap<KP> pm(pix0 + 16 * p + j);
      const bool valid = pm.s < a.S;
      const int s = valid ? pm.s : a.S - 1;
      const int k = pm.k < a.K ? pm.k : 0;           // padded slots replicate neighbour 0 (max-neutral)
      sq[p] = valid ? s : -1;
      const int nbr = a.idx[((size_t)b * a.S + s) * a.K + k];
      const float *c = a.new_xyz + ((size_t)b * a.S + s) * 3;
      const float *q = a.xyz + ((size_t)b * a.N + nbr) * 3;
      const float qx = q[0], qy = q[1], qz = q[2];
      const float dx = qx - c[0], dy = qy - c[1], dz = qz - c[2];
      f32x4 geo = {0.f, 0.f, 0.f, 0.f};
      if (g == 0) geo = f32x4{dx, dy, dz, XYZ_ONLY ? qx : 0.f};
      if (XYZ_ONLY && g == 1) geo = f32x4{qy, qz, 0.f, 0.f};
      in[0][p] = geo;
#pragma unroll
      for (int m = 0; m < CFB; ++m)
        in[1 + m][p] = *reinterpret_cast<const f32x4 *>(a.feat + ((size_t)b * a.N + nbr) * CF + 16 * m + 4 * g);
    }
    f32x4 h1[B1][P], h2[B2][P], h3[B3][P];
    mlp_layer<NBI, B1, P, true>(h1, in, lds_w, lane);
    mlp_layer<B1, B2, P, true>(h2, h1, lds_w + W1, lane);
    mlp_layer<B2, B3, P, false>(h3, h2, lds_w + W1 + W2, lane);   // its ReLU is applied after the pool

    // max over the K neighbours (F.max_pool2d(kernel=[1,K]), :239-243), then one 16-byte store
    // per (query, lane group, output block)
    constexpr int GROUP = KP < 16 ? KP : 16;
    constexpr int BPQ = KP > 16 ? KP / 16 : 1;        // pixel blocks per query
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
        v.x = relu_bits(group_max_nonneg<GROUP>(v.x));
        v.y = relu_bits(group_max_nonneg<GROUP>(v.y));
        v.z = relu_bits(group_max_nonneg<GROUP>(v.z));
        v.w = relu_bits(group_max_nonneg<GROUP>(v.w));
        if ((j & (GROUP - 1)) == 0 && sq[p] >= 0)
          *reinterpret_cast<f32x4 *>(a.out + ((size_t)b * a.S + sq[p]) * C3 + 16 * o + 4 * g) = v;
      }
    }
  }
}

template <int CFB, int B1, int B2, int B3, int KP, int P, bool XYZ_ONLY>
static void launch_sa(const SAArgs &a) {
  constexpr int NBI = 1 + CFB;
  constexpr int lds_bytes = 4 * (layer_floats(NBI, B1) + layer_floats(B1, B2) + layer_floats(B2, B3));
  static_assert(lds_bytes <= 160 * 1024, "packed weights must fit in LDS");
  auto kern = sa_kernel<CFB, B1, B2, B3, KP, P, XYZ_ONLY>;
  static bool attr_set = false;
  if (lds_bytes > 64 * 1024 && !attr_set) {
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);  // once per kernel: the largest any configuration can ask for
    attr_set = true;
  }
  const long long pix = (long long)a.S * KP;
  const long long ntiles = (long long)a.B * ((pix + 16 * P - 1) / (16 * P));
  static int rounds = -1;
  if (rounds < 0) { const char *e = getenv("PWCLO_FL_ROUNDS"); rounds = e ? atoi(e) : 2; }
  const int per_cu = lds_bytes > 80 * 1024 ? 1 : 2;
  long long grid = (ntiles + SA_WAVES - 1) / SA_WAVES;
  if (grid > 256LL * per_cu * rounds) grid = 256LL * per_cu * rounds;   // persistent workgroups
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(SA_WAVES * 64), lds_bytes, current_stream(), a);
}

}  // namespace pwclo

using namespace pwclo;

// See include/pwclo_ops.h section 3.
extern "C" void sa_fused_kernel_wrapper(int b, int n, int s, int k, int c_feat, int c1, int c2, int c3,
                                        const float *xyz, const float *new_xyz, const float *feat,
                                        const int *idx, const float *packed_w, float *out) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(k >= 1 && k <= 32, "sa_fused: nsample=%d outside [1,32]", k);
  static int stagger = -1;
  if (stagger < 0) { const char *e = getenv("PWCLO_FL_STAGGER"); stagger = e ? atoi(e) : 0; }
  SAArgs a{xyz, new_xyz, feat, idx, packed_w, out, b, n, s, k, stagger};
  const int kp = k > 16 ? 32 : 16;
#define SA_CASE(CF, A1, A2, A3, KP, XYZ)                                                          \
  if (c_feat == CF && c1 == A1 && c2 == A2 && c3 == A3 && kp == KP) {                             \
    launch_sa<CF / 16, A1 / 16, A2 / 16, A3 / 16, KP, 2, XYZ>(a);                                 \
    check_launch("sa_fused");                                                                     \
    return;                                                                                       \
  }
  SA_CASE(0, 16, 16, 16, 32, true)     // psa_1: 6 -> 8 -> 8 -> 16 (8s padded to 16)
  SA_CASE(16, 16, 16, 32, 32, false)   // psa_2: 19 -> 16 -> 16 -> 32
  SA_CASE(32, 32, 32, 64, 16, false)   // psa_3: 35 -> 32 -> 32 -> 64
  SA_CASE(64, 64, 64, 128, 16, false)  // psa_4: 67 -> 64 -> 64 -> 128
  SA_CASE(64, 128, 64, 64, 16, false)  // flow_feature_encoding: 67 -> 128 -> 64 -> 64
#undef SA_CASE
  set_error(PWCLO_EINVAL, "sa_fused: no kernel for c_feat=%d mlp=(%d,%d,%d) nsample=%d", c_feat, c1, c2,
            c3, k);
}
