// Register-resident per-point MLP on the fp32 matrix cores (gfx950, v_mfma_f32_16x16x4_f32).
//
// The grouped "shared MLPs" of PWCLO-Net (P2/pytorch_utils.py:52-83: Conv2d 1x1 -> BN -> ReLU
// over (B,C,S,K)) are dense contractions over channels at every (query, neighbour) "pixel":
//     D[cout][pixel] = sum_cin W[cout][cin] * X[cin][pixel] + bias[cout]
// One wave owns P blocks of 16 pixels and walks them through ALL layers of a stack without
// leaving registers:
//   * MFMA roles: A = W (16 couts x 4 cins), B = X (4 cins x 16 pixels), so a lane (g = lane>>4,
//     j = lane&15) holds pixel j and, in every block of 16 channels, the four channels
//     16m+4g+r (r = 0..3) -- the accumulator layout of the 16x16 MFMA (row = 4g+r, col = j).
//   * The k-steps of the next layer are enumerated as (m, r) and use channel 16m+4g+r from lane
//     group g, so a layer's accumulators ARE the next layer's B operands: no LDS round trip and
//     no cross-lane movement between layers.  The matching permutation lives in the packed
//     weights (host side, fused.py: pack_layer).
//   * BatchNorm (eval) is folded into W and bias on the host; bias seeds the accumulator.
//   * Packed weights are staged once per workgroup into LDS ([o][m][lane][4] floats: one
//     conflict-free ds_read_b128 per (o, m) and lane feeds 4*P MFMAs) and stay resident while
//     the (persistent) workgroup streams pixel tiles; there is no barrier after the fill.
// fp32 MFMA is bit-for-bit an fmaf chain in k order, so results differ from a CPU GEMM only by
// summation order (parity bound 1e-5, tests/test_gpu_fused.py).
#pragma once
#include <type_traits>
#include <stdlib.h>

#include "common.hpp"

namespace pwclo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Floats occupied in LDS by one packed layer: NBO*NBI 1-KiB operand tiles + NBO*16 biases.
constexpr int layer_floats(int nbi, int nbo) { return nbo * nbi * 256 + nbo * 16; }

// Cooperative copy of `nfloats` (multiple of 4) packed floats from global memory into LDS, with up
// to 8 independent 16-byte loads in flight per thread (a one-load-per-trip loop pays the full memory
// latency per trip: 5-6 us for the 100-130 KB stacks, twice per CU at two workgroup rounds).
template <int U>
__device__ __forceinline__ int stage_batch(f32x4 *dst, const f32x4 *__restrict__ src, int i, int n4, int step) {
  for (; i + (U - 1) * step < n4; i += U * step) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i + u * step];
#pragma unroll
    for (int u = 0; u < U; ++u) dst[i + u * step] = v[u];
  }
  return i;
}
__device__ __forceinline__ void stage_weights(float *lds, const float *__restrict__ g, int nfloats) {
  const f32x4 *src = reinterpret_cast<const f32x4 *>(g);
  f32x4 *dst = reinterpret_cast<f32x4 *>(lds);
  const int n4 = nfloats / 4, step = blockDim.x;
  int i = stage_batch<8>(dst, src, threadIdx.x, n4, step);
  i = stage_batch<2>(dst, src, i, n4, step);
  stage_batch<1>(dst, src, i, n4, step);
}

// ReLU as ONE integer max on the bit pattern (negative floats are negative ints; -0.0 -> +0.0).
// fmaxf(x, 0.f) costs two VALU instructions (the compiler canonicalises x first), and on gfx950 the
// fp32 MFMA does not co-issue with other vector instructions of the SIMD (measured:
// SQ_VALU_MFMA_BUSY + SQ_ACTIVE_INST_VALU = SQ_BUSY, co-execution 0), so every VALU instruction in a
// stack kernel is time taken from the matrix pipe.
__device__ __forceinline__ float relu_bits(float x) {
  const int b = __float_as_int(x);
  return __int_as_float(b < 0 ? 0 : b);
}
// max of two floats through their bit patterns as signed ints: exact when at least one is >= +0;
// when both are negative the result is still negative (all a following relu_bits needs).
__device__ __forceinline__ float max_bits(float a, float b) {
  const int x = __float_as_int(a), y = __float_as_int(b);
  return __int_as_float(x > y ? x : y);
}

// One layer for P pixel blocks.  `w` points at the layer's packed weights in LDS.
// The (output block o, input block m) operand tiles are walked as ONE flat sequence with the next
// tile's ds_read_b128 issued before the current tile's 4*P MFMAs (register double buffer, 8 VGPRs),
// so the LDS latency sits under matrix work; a sched_barrier per tile keeps the scheduler from
// hoisting every later read as well (which spilled).
// `init(o, p)` seeds the accumulator of output block o / pixel block p: the layer's bias, or --
// for hoisted first layers -- the per-point partial products gathered from memory (see below).
// Softmax pieces for the K-neighbour epilogues, priced in VALU instructions (see relu_bits):
// exp of a non-positive argument as v_mul + v_exp_f32 (expf is a 12-instruction sequence; the
// argument is x - max <= 0, the result feeds a ratio of sums, error ~1e-6 relative for |x| < 20),
// and a quotient whose denominator is >= 1 (it contains exp(0)) as v_rcp_f32 + v_mul (IEEE division
// is ~10 instructions).  Both stay inside the 1e-5 parity bound of the fused layers.
__device__ __forceinline__ float exp_nonpos(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float div_ge1(float num, float den) { return num * __builtin_amdgcn_rcpf(den); }

// KS < 4 runs only the first KS 4-channel k-steps of every input block: for a lone geometry block
// whose channels 4*KS.. are zero padding (3 or 6 or 10 real inputs) the other MFMAs multiply zeros.
template <int NBI, int NBO, int P, bool RELU, int KS = 4, typename Init>
__device__ __forceinline__ void mlp_layer_init(f32x4 (&out)[NBO][P], const f32x4 (&in)[NBI][P],
                                               const float *w, int lane, Init init) {
  const float *wl = w + lane * 4;
  f32x4 wv = *reinterpret_cast<const f32x4 *>(wl);
  f32x4 acc[P];
#pragma unroll
  for (int t = 0; t < NBO * NBI; ++t) {
    const int o = t / NBI, m = t % NBI;
    f32x4 wnext = wv;
    if (t + 1 < NBO * NBI) wnext = *reinterpret_cast<const f32x4 *>(wl + (t + 1) * 256);
    if (m == 0) {
#pragma unroll
      for (int p = 0; p < P; ++p) acc[p] = init(o, p);
    }
#pragma unroll
    for (int r = 0; r < KS; ++r) {
#pragma unroll
      for (int p = 0; p < P; ++p)
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[r], in[m][p][r], acc[p], 0, 0, 0);
    }
    if (m == NBI - 1) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (RELU) {
          f32x4 v = acc[p];
          v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
          out[o][p] = v;
        } else {
          out[o][p] = acc[p];
        }
      }
    }
    wv = wnext;
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NBI, int NBO, int P, bool RELU, int KS = 4>
__device__ __forceinline__ void mlp_layer(f32x4 (&out)[NBO][P], const f32x4 (&in)[NBI][P],
                                          const float *w, int lane) {
  const float *bias = w + NBO * NBI * 256 + 4 * (lane >> 4);
  mlp_layer_init<NBI, NBO, P, RELU, KS>(out, in, w, lane, [&](int o, int) {
    return *reinterpret_cast<const f32x4 *>(bias + 16 * o);
  });
}

// ---- opt-in: fp32-accurate layer on the bf16 matrix pipe (PWCLO_BF16X3=1, default off) -------------
// Every operand is split into three bf16 terms (x = hi + mid + lo, each rounded to nearest) and six of the
// nine cross products are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 (the dropped mid*lo, lo*mid,
// lo*lo terms are below 2^-24 relative).  A K = 32 instruction consumes TWO 16-channel blocks: lane group g
// supplies k = 8g..8g+7, taken as its four registers of block 2*mp and its four of block 2*mp+1, so the
// accumulator-is-the-next-operand property of the fp32 path carries over; the packed weights hold, per
// (o, mp) tile, [split][lane][8 bf16] in the matching order (fused.py: pack_layer_bf3).  96 matrix cycles
// per 32 input channels instead of 256, paid for with ~5.5 VALU instructions per split activation.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct Split3 { bf16x8 hi, mid, lo; };

__device__ __forceinline__ Split3 split3(const f32x4 a, const f32x4 b) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  Split3 s;
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const bf16x2 h = {(__bf16)v[i], (__bf16)v[i + 1]};
    const float r0 = v[i] - (float)h[0], r1 = v[i + 1] - (float)h[1];
    const bf16x2 m = {(__bf16)r0, (__bf16)r1};
    const float q0 = r0 - (float)m[0], q1 = r1 - (float)m[1];
    const bf16x2 l = {(__bf16)q0, (__bf16)q1};
    s.hi[i] = h[0]; s.hi[i + 1] = h[1];
    s.mid[i] = m[0]; s.mid[i + 1] = m[1];
    s.lo[i] = l[0]; s.lo[i + 1] = l[1];
  }
  return s;
}

constexpr int BF3_TILE_FLOATS = 768;   // 3 splits x 64 lanes x 8 bf16 = 3 KiB per (o, mp) tile
constexpr int layer_floats_bf3(int nbi, int nbo) { return nbo * (nbi / 2) * BF3_TILE_FLOATS + nbo * 16; }

template <int NBI, int NBO, int P, bool RELU, typename Init>
__device__ __forceinline__ void mlp_layer_bf3_init(f32x4 (&out)[NBO][P], const f32x4 (&in)[NBI][P],
                                                   const float *w, int lane, Init init) {
  static_assert(NBI % 2 == 0, "a K = 32 step consumes two 16-channel blocks");
  constexpr int NP = NBI / 2;
  Split3 x[NP][P];
#pragma unroll
  for (int mp = 0; mp < NP; ++mp)
#pragma unroll
    for (int p = 0; p < P; ++p) x[mp][p] = split3(in[2 * mp][p], in[2 * mp + 1][p]);
  const float *wl = w + lane * 4;                      // 16 bytes per lane and split
#pragma unroll
  for (int o = 0; o < NBO; ++o) {
    f32x4 acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = init(o, p);
#pragma unroll
    for (int mp = 0; mp < NP; ++mp) {
      const float *t = wl + (o * NP + mp) * BF3_TILE_FLOATS;
      const bf16x8 whi = *reinterpret_cast<const bf16x8 *>(t);
      const bf16x8 wmid = *reinterpret_cast<const bf16x8 *>(t + 256);
      const bf16x8 wlo = *reinterpret_cast<const bf16x8 *>(t + 512);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        f32x4 c = acc[p];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, x[mp][p].hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, x[mp][p].lo, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wmid, x[mp][p].mid, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wmid, x[mp][p].hi, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, x[mp][p].mid, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, x[mp][p].hi, c, 0, 0, 0);
        acc[p] = c;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (RELU) {
        f32x4 v = acc[p];
        v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
        out[o][p] = v;
      } else {
        out[o][p] = acc[p];
      }
    }
  }
}

template <int NBI, int NBO, int P, bool RELU>
__device__ __forceinline__ void mlp_layer_bf3(f32x4 (&out)[NBO][P], const f32x4 (&in)[NBI][P],
                                              const float *w, int lane) {
  const float *bias = w + NBO * (NBI / 2) * BF3_TILE_FLOATS + 4 * (lane >> 4);
  mlp_layer_bf3_init<NBI, NBO, P, RELU>(out, in, w, lane, [&](int o, int) {
    return *reinterpret_cast<const f32x4 *>(bias + 16 * o);
  });
}

// ---- bf16 layer (dtype = "bf16": BASELINE configs[4]) ----------------------------------------------------------
// Weights and activations rounded ONCE to bf16 (round to nearest even), products accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16: 16 matrix cycles per 32 input channels (fp32 path: 256), 1 KiB of LDS per (o, mp) tile
// (fp32 path: 2 KiB).  Same register layout as the split path above with only its `hi` term.  Bias / hoisted seeds,
// accumulation, ReLU, pooling and softmax stay fp32; coordinates, distances and indices never pass through here.
constexpr int BF16_TILE_FLOATS = 256;   // 64 lanes x 8 bf16 = 1 KiB per (o, mp) tile
constexpr int layer_floats_bf16(int nbi, int nbo) { return nbo * (nbi / 2) * BF16_TILE_FLOATS + nbo * 16; }

__device__ __forceinline__ bf16x8 to_bf16x8(const f32x4 a, const f32x4 b) {
  bf16x8 r;
  r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
  r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
  return r;
}

template <int NBI, int NBO, int P, bool RELU, typename Init>
__device__ __forceinline__ void mlp_layer_bf16_init(f32x4 (&out)[NBO][P], const f32x4 (&in)[NBI][P],
                                                    const float *w, int lane, Init init) {
  static_assert(NBI % 2 == 0, "a K = 32 step consumes two 16-channel blocks");
  constexpr int NP = NBI / 2;
  bf16x8 x[NP][P];
#pragma unroll
  for (int mp = 0; mp < NP; ++mp)
#pragma unroll
    for (int p = 0; p < P; ++p) x[mp][p] = to_bf16x8(in[2 * mp][p], in[2 * mp + 1][p]);
  const float *wl = w + lane * 4;                      // 16 bytes per lane
#pragma unroll
  for (int o = 0; o < NBO; ++o) {
    f32x4 acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = init(o, p);
#pragma unroll
    for (int mp = 0; mp < NP; ++mp) {
      const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(wl + (o * NP + mp) * BF16_TILE_FLOATS);
#pragma unroll
      for (int p = 0; p < P; ++p) acc[p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, x[mp][p], acc[p], 0, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      if (RELU) {
        f32x4 v = acc[p];
        v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
        out[o][p] = v;
      } else {
        out[o][p] = acc[p];
      }
    }
  }
}

template <int NBI, int NBO, int P, bool RELU>
__device__ __forceinline__ void mlp_layer_bf16(f32x4 (&out)[NBO][P], const f32x4 (&in)[NBI][P],
                                               const float *w, int lane) {
  const float *bias = w + NBO * (NBI / 2) * BF16_TILE_FLOATS + 4 * (lane >> 4);
  mlp_layer_bf16_init<NBI, NBO, P, RELU>(out, in, w, lane, [&](int o, int) {
    return *reinterpret_cast<const f32x4 *>(bias + 16 * o);
  });
}

// Layer dispatch used by the stack kernels.  FMT: 0 = fp32 MFMA, 1 = three-term bf16 split, 2 = bf16 (the reduced
// formats need an even NBI; odd layers -- lone geometry blocks -- always run the fp32 MFMA).
template <int FMT, int NBI, int NBO, int P, bool RELU>
__device__ __forceinline__ void mlp_layer_any(f32x4 (&out)[NBO][P], const f32x4 (&in)[NBI][P],
                                              const float *w, int lane) {
  if constexpr (FMT == 1 && NBI % 2 == 0) mlp_layer_bf3<NBI, NBO, P, RELU>(out, in, w, lane);
  else if constexpr (FMT == 2 && NBI % 2 == 0) mlp_layer_bf16<NBI, NBO, P, RELU>(out, in, w, lane);
  else mlp_layer<NBI, NBO, P, RELU>(out, in, w, lane);
}
template <int FMT, int NBI, int NBO, int P, bool RELU, typename Init>
__device__ __forceinline__ void mlp_layer_any_init(f32x4 (&out)[NBO][P], const f32x4 (&in)[NBI][P],
                                                   const float *w, int lane, Init init) {
  if constexpr (FMT == 1 && NBI % 2 == 0) mlp_layer_bf3_init<NBI, NBO, P, RELU>(out, in, w, lane, init);
  else if constexpr (FMT == 2 && NBI % 2 == 0) mlp_layer_bf16_init<NBI, NBO, P, RELU>(out, in, w, lane, init);
  else mlp_layer_init<NBI, NBO, P, RELU>(out, in, w, lane, init);
}
template <int FMT>
constexpr int layer_floats_any(int nbi, int nbo) {
  return (FMT == 1 && nbi % 2 == 0) ? layer_floats_bf3(nbi, nbo)
         : (FMT == 2 && nbi % 2 == 0) ? layer_floats_bf16(nbi, nbo) : layer_floats(nbi, nbo);
}

// The format of a packed weight buffer is a property of the BUFFER, fixed when it was packed (fused.py records it
// on the packed object): the stack launchers take it as an explicit argument `wfmt` together with the buffer's
// length in floats, and refuse a length that does not match the layout the selected kernel will index
// (a buffer packed in one format and launched as the other would otherwise be read out of bounds, silently).
enum : int { PWCLO_WFMT_F32 = 0, PWCLO_WFMT_BF16X3 = 1, PWCLO_WFMT_BF16 = 2 };
#define PWCLO_REQUIRE_PACKED(what, wfmt, packed_floats, floats_f32, floats_bf3, floats_bf16)                     \
  do {                                                                                                           \
    PWCLO_REQUIRE((wfmt) >= PWCLO_WFMT_F32 && (wfmt) <= PWCLO_WFMT_BF16, what ": unknown weight format %d",        \
                  (int)(wfmt));                                                                                  \
    const int expect_ = (wfmt) == PWCLO_WFMT_BF16X3 ? (int)(floats_bf3)                                           \
                        : (wfmt) == PWCLO_WFMT_BF16 ? (int)(floats_bf16) : (int)(floats_f32);                      \
    PWCLO_REQUIRE((packed_floats) == expect_, what ": packed weights hold %d floats, format %d needs %d",          \
                  (int)(packed_floats), (int)(wfmt), expect_);                                                   \
  } while (0)

// Hoisting.  The first layer of a grouped MLP is linear in its concatenated input
// [geometry(q,p) | feat_centre[s] | feat_nbr[n]], and the feature parts depend on ONE point, not on
// the (query, neighbour) pixel.  W_feat . feat[point] (+ bias) is therefore computed once per
// point by linear_jobs_kernel (K x fewer MACs than per pixel) and the pixel kernels seed their
// first-layer accumulators with the gathered rows, leaving only the 16-channel geometry block
// for the per-pixel MFMAs.  Same mathematics, different fp32 summation order (parity bound 1e-5).

// ---- start-up stagger ------------------------------------------------------------------------------
// Experiment kept for reference (PWCLO_FL_STAGGER, default off, no measurable effect): delay the j-th
// wave of a SIMD once at kernel start so that one wave's gather prologue / epilogue would sit under
// the other waves' MFMAs.  The premise was wrong for fp32: SQ_VALU_MFMA_BUSY + SQ_ACTIVE_INST_VALU
// add up to the busy time with zero co-execution in every stack kernel, i.e. the fp32 MFMA and the
// other vector instructions of a SIMD execute one after the other however the waves are phased
// (and software-prefetching the next tile's gathers did not help either).  What does help is
// issuing fewer VALU instructions per MFMA (relu_bits, ReLU after the pool, k-step trimming).
__device__ __forceinline__ void stagger_start(int wave, int mfma_per_tile, int enable) {
  const int j = wave >> 2;
  if (enable && j > 0) {
    const int naps = (j * mfma_per_tile * 32 + 4095) / 4096;
    for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(64);   // 64 x 64 clocks
  }
}

// ---- reductions over the K neighbours of a query ------------------------------------------------
// Pixels are ordered (query, k) with k fastest and K padded to KP in {1, 4, 8, 16, 32}; inside a
// 16-pixel block the neighbours of one query are KP (<= 16) consecutive lanes of a DPP row.
// GROUP = min(KP, 16).  After the call every lane of a group holds the group's result.
template <int GROUP, typename Op>
__device__ __forceinline__ unsigned group_allreduce_u32(unsigned v, Op op) {
  if (GROUP >= 2) v = op(v, dpp_u32<0xB1>(v));    // xor 1
  if (GROUP >= 4) v = op(v, dpp_u32<0x4E>(v));    // xor 2
  if (GROUP >= 8) v = op(v, dpp_u32<0x141>(v));   // mirror within 8
  if (GROUP >= 16) v = op(v, dpp_u32<0x140>(v));  // mirror within 16
  return v;
}

struct OpAddF32 {
  __device__ __forceinline__ unsigned operator()(unsigned a, unsigned b) const {
    return __float_as_uint(__uint_as_float(a) + __uint_as_float(b));
  }
};
struct OpMaxF32Bits {  // max of floats through their bit patterns: valid for non-negative values,
  __device__ __forceinline__ unsigned operator()(unsigned a, unsigned b) const {  // -inf sorts low
    return (int)a > (int)b ? a : b;
  }
};

template <int GROUP>
__device__ __forceinline__ float group_sum(float v) {
  return __uint_as_float(group_allreduce_u32<GROUP>(__float_as_uint(v), OpAddF32()));
}

// max for values that are >= +0 (post-ReLU) or the sentinel -inf (bit pattern 0xff800000 < 0); for
// raw pre-activation values the result is exact whenever the true max is >= +0 and negative
// otherwise, so relu(max_k x_k) == relu_bits(group_max_nonneg(x)) (ReLU after the pool: 1/K of the work).
template <int GROUP>
__device__ __forceinline__ float group_max_nonneg(float v) {
  return __uint_as_float(group_allreduce_u32<GROUP>(__float_as_uint(v), OpMaxF32Bits()));
}

// ---- pixel bookkeeping --------------------------------------------------------------------------
// Neighbour slots per query of the cost volume's per-pixel feature buffer (cv_a1 -> cv_a2): K
// rounded up to 8 / 16 / 32, or 6 for K == 6 (the refinement levels: stored densely, consumed by the
// dense-6 / in-lane kernels).  The CALLER chooses (it allocates the buffer: fused.py cv_pix_slots) and passes
// the choice to both kernels as `pix_slots`; the launchers only validate it.
static inline bool cv_pix_slots_valid(int k, int slots) {
  if (slots == 6) return k == 6;
  return (slots == 8 || slots == 16 || slots == 32) && k <= slots && (slots == 8 || k > slots / 2);
}

// Index of the wave inside its workgroup as a SCALAR: the compiler cannot know that threadIdx.x >> 6 is
// wave-uniform, and without this the tile index, the cloud index (an integer division) and every per-cloud
// base address of the tile loops are computed on the vector ALU -- which the matrix pipe waits for.
__device__ __forceinline__ int wave_index() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// base + 32-bit BYTE offset: lets the compiler address global memory as (scalar base, 32-bit vector offset,
// immediate) instead of building 64-bit addresses per lane with quarter-rate multiplies and 64-bit adds on the
// vector ALU.  The launchers check that every such offset stays below 4 GiB.
template <typename T>
__device__ __forceinline__ T *at32(T *base, unsigned byte_off) {
  using C = typename std::conditional<std::is_const<T>::value, const char, char>::type;
  return reinterpret_cast<T *>(reinterpret_cast<C *>(base) + (size_t)byte_off);
}
__device__ __forceinline__ unsigned mul24(unsigned a, unsigned b) { return __umul24(a, b); }   // both < 2^24: full rate
// Host-side guard of the above: `rows` rows (points or pixels over the whole batch) of at most 512 bytes each
// must be addressable with 32 bits (and rows < 2^24 for mul24): 8.3 M rows, e.g. batch 1024 of 8192-point clouds.
static inline bool rows_fit_32bit(long long rows) { return rows >= 0 && rows < (1ll << 23); }

// ---- bf16 STORAGE of intermediate feature rows (dtype = "bf16") ---------------------------------------------------
// With FMT == 2 the hoisted partial products (linear_jobs outputs gathered K times per pixel: most of the gathered
// bytes of a forward) and the cost volume's per-pixel feature buffer live in HBM as bf16: a 4-channel group is 8 bytes
// instead of 16.  Values are rounded once (round to nearest even) when written and widened exactly when read; point
// features between modules, coordinates, distances and indices stay fp32.
__device__ __forceinline__ f32x4 ld4_bf16(const void *p) {          // 4 consecutive bf16 -> fp32 (exact)
  const uint2 v = *reinterpret_cast<const uint2 *>(p);
  f32x4 r;
  r.x = __uint_as_float(v.x << 16); r.y = __uint_as_float(v.x & 0xFFFF0000u);
  r.z = __uint_as_float(v.y << 16); r.w = __uint_as_float(v.y & 0xFFFF0000u);
  return r;
}
__device__ __forceinline__ void st4_bf16(void *p, const f32x4 v) {  // fp32 -> 4 bf16, round to nearest even
  const bf16x2 a = {(__bf16)v.x, (__bf16)v.y}, b = {(__bf16)v.z, (__bf16)v.w};
  uint2 o;
  o.x = *reinterpret_cast<const unsigned *>(&a);
  o.y = *reinterpret_cast<const unsigned *>(&b);
  *reinterpret_cast<uint2 *>(p) = o;
}
// Row of C channels, 4-channel group (block o, lane group g): fp32 rows are 4*C bytes, bf16 rows 2*C.
template <bool H16>
__device__ __forceinline__ f32x4 ld_group(const float *base, unsigned row, unsigned c, int o, int g) {
  if constexpr (H16) return ld4_bf16(at32(base, row * (2u * c) + 32u * (unsigned)o + 8u * (unsigned)g));
  else return *reinterpret_cast<const f32x4 *>(at32(base, row * (4u * c) + 64u * (unsigned)o + 16u * (unsigned)g));
}
template <bool H16>
__device__ __forceinline__ void st_group(float *base, unsigned row, unsigned c, int o, int g, const f32x4 v) {
  if constexpr (H16) st4_bf16(at32(base, row * (2u * c) + 32u * (unsigned)o + 8u * (unsigned)g), v);
  else *reinterpret_cast<f32x4 *>(at32(base, row * (4u * c) + 64u * (unsigned)o + 16u * (unsigned)g)) = v;
}

// pixel index -> (query s, neighbour slot k).
template <int KP>
struct PixelMap {
  int s, k;
  __device__ __forceinline__ PixelMap(int pix) : s(pix / KP), k(pix % KP) {}
};

}  // namespace pwclo
