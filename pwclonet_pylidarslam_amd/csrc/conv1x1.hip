// Pointwise (1x1) convolution of the module path's shared MLPs on the fp32 matrix cores: forward, input gradient and
// weight gradient, directly on the (B, C, S, K) channel-major activations the reference's modules use
// (the layers are pytorch_utils.py:114-167 `Conv2d(kernel_size=(1,1), bias=False)` inside
// SharedMLP, pytorch_utils.py:12-37).  The library convolutions need NHWC transposes around their GEMMs for these
// shapes (6..192 input channels, 8..128 output channels, up to 2M pixels); these kernels read and write the
// channel-major rows as they are:
//   * forward / input gradient: every lane loads / stores 2 consecutive pixels of one channel row (16 lanes cover one
//     128-byte line), the B operands of 2 independent 16-pixel MFMA columns; a load cursor runs CONV_AHEAD channel
//     blocks ahead of the multiply cursor, across tile boundaries;
//   * the weights sit in LDS in MFMA A-operand order, one ds_read_b128 feeds 8 MFMAs;
//   * an optional epilogue folds an eval-mode BatchNorm (+ReLU) and the stack's max over K into the forward;
//   * the weight gradient stages (dY, X) pixel chunks through LDS once per workgroup and every wave accumulates its
//     own rectangle of 16x16 (co, ci) tiles over the chunk; partial sums per workgroup are reduced in a fixed order by
//     a second kernel (deterministic, no atomics).
// v_mfma_f32_16x16x4_f32 everywhere: products and sums are fp32 FMAs, so the results differ from any other fp32
// convolution by summation order only.
#include <mutex>

#include "common.hpp"
#include "train_internal.hpp"

namespace pwclo {

typedef float cv_f32x4 __attribute__((ext_vector_type(4)));

constexpr int CONV_THREADS = 512;            // 8 waves: 2 per SIMD
constexpr int CONV_WAVES = CONV_THREADS / WAVE;
constexpr int CONV_MAX_NBO = 8;              // 16-channel output blocks per workgroup (128 accumulator VGPRs)
constexpr int CONV_AHEAD = 4;               // channel blocks in flight per wave (forward / input gradient)
constexpr int CONV_STAT_TILES = 32;        // tiles a lane adds in fp32 before its sums go to the fp64 slots
constexpr int WGRAD_MAXV = 8;                // float4 per thread of one staged weight-gradient chunk

// y[b][co][p] = sum_ci W(co, ci) x[b][ci][p], W(co, ci) = w[co * w_ld_o + ci * w_ld_i]  (strides: the same kernel
// computes the input gradient with the transposed view).  P % 4 == 0, rows 16-byte aligned.  Epilogue (forward only):
// ep_scale / ep_shift (per output channel, nullable) and ep_relu; ep_pool = K > 0 writes max over each row of K pixels.
// grid (x = persistent tile workers, y = groups of NBO output blocks); dynamic LDS = NBO * nbi KiB.
// STATS (training forward): the kernel also leaves, per output channel, the partial sums (sum y, sum y^2) of everything
// THIS workgroup wrote, in fp64, at stats_partial[(co * gridDim.x + blockIdx.x) * 2 + {0, 1}] -- the layout
// bn_forward_finish_kernel folds -- so the BatchNorm that follows the convolution (pytorch_utils.py:114-167: conv -> bn ->
// relu) needs no statistics pass over y.  Per lane the two pixels of a tile are added in fp32 for at most CONV_STAT_TILES
// tiles (64 values), then reduced over the 16 lanes of a row (DPP) and added to the wave's own fp64 slot in LDS by one
// lane; the 8 slots are summed in wave order at the end: deterministic, and within ~1e-6 of the fp64 pass it replaces.
// LEAN: no folded BatchNorm, no pooling in the epilogue and y below 4 GiB (training forward / input gradient): the short
// epilogue with descriptor stores; STATS implies LEAN.
// STATS == 2 (input gradient of a layer that follows a training-mode BatchNorm + ReLU): the output is da, the gradient
// w.r.t. the rectified, normalised activation; the epilogue reads the BatchNorm's INPUT bx at the tile's own positions
// (same shape as the output), rebuilds xhat and the ReLU mask with the forward's expression and leaves the partial sums
// (sum g, sum g * xhat), g = masked da, in the same workspace layout -- the reduction pass of the BatchNorm backward
// (bn_partial_kernel<1, true>, the largest kernel of the training step: a read of bx AND of da) without reading da again.
// The per-OUTPUT-channel (mean, invstd, gamma, beta) arrive through the in_* pointers (an input gradient has no input
// transform).
template <int NBO, int STATS, bool LEAN>
__global__ __launch_bounds__(CONV_THREADS, (STATS == 2 ? 2 : NBO <= 2 || (NBO == 3 && !STATS) ? 4 : (STATS && NBO > 4) ? 1 : 2)) void conv1x1_kernel(int B, int Cin, int Cout, int P, int nbi,
                                                              long long w_ld_o, long long w_ld_i,
                                                              const float *__restrict__ x,
                                                              const float *__restrict__ w, float *__restrict__ y,
                                                              const float *__restrict__ ep_scale,
                                                              const float *__restrict__ ep_shift, int ep_relu, int ep_pool,
                                                              const float *__restrict__ in_mean,
                                                              const float *__restrict__ in_invstd,
                                                              const float *__restrict__ in_gamma,
                                                              const float *__restrict__ in_beta,
                                                              double *__restrict__ stats_partial, unsigned x_bytes,
                                                              unsigned y_bytes, const float *__restrict__ bx) {
  extern __shared__ float4 conv_w[];         // [NBO][nbi][64 lanes] : the 4 k-steps of one (o, m) tile per lane
  const int ob0 = blockIdx.y * NBO;
  {
    // pack: zero the padding, then walk the stored matrix in memory order (coalesced) and scatter into operand order
    float *wf = reinterpret_cast<float *>(conv_w);
    for (int e = threadIdx.x; e < NBO * nbi * WAVE; e += CONV_THREADS) conv_w[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const bool tr = (w_ld_o == 1);           // stored (Cin, Cout) row-major: the input-gradient view
    // only this workgroup's NBO output blocks: rows [16 ob0, +16 NBO) of the stored (Cout, Cin) matrix, or those COLUMNS of
    // the (Cin, Cout) one -- a workgroup of a channel-split launch does not walk the other groups' weights
    const int gco = min(NBO * 16, Cout - 16 * ob0);
    const int cols = tr ? Cout : Cin;
    const int grows = tr ? Cin : gco, gcols = tr ? gco : Cin;
    const int total = grows * gcols;
    constexpr int PK = 8;                    // independent loads in flight per thread, then the scatter (16 measured the same:
                                             // the pack is not what the small layers wait for)
    for (int e0 = threadIdx.x; e0 < total; e0 += PK * CONV_THREADS) {
      float v[PK];
      int rr[PK], cc[PK];
#pragma unroll
      for (int u = 0; u < PK; ++u) {
        const int e = e0 + u * CONV_THREADS;
        const int r = e / gcols, c = e - r * gcols;
        rr[u] = r; cc[u] = c;
        v[u] = e < total ? w[tr ? (size_t)r * cols + 16 * ob0 + c : ((size_t)16 * ob0 + r) * cols + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < PK; ++u) {
        const int e = e0 + u * CONV_THREADS;
        if (e < total) {
          const int co = 16 * ob0 + (tr ? cc[u] : rr[u]), ci = tr ? rr[u] : cc[u];
          const int o = (co >> 4) - ob0;
          const int m = ci >> 4, s = (ci >> 2) & 3, gg = ci & 3;
          wf[(((o * nbi + m) * WAVE) + gg * 16 + (co & 15)) * 4 + s] = v[u];
        }
      }
    }
    // in_mean != nullptr: the input is the PREVIOUS layer's convolution output and its training-mode BatchNorm + ReLU
    // is applied on the fly, a = max(((x - mean) * invstd) * gamma + beta, 0) -- the expression of bn_apply_kernel, so
    // the normalised activation is never written; per input channel (mean, invstd, gamma, beta), zeros for padding
    float4 *conv_in = conv_w + NBO * nbi * WAVE;
    if (STATS) {
      double *slots = reinterpret_cast<double *>(conv_in + nbi * 16);       // [8 waves][NBO * 16 channels][2]
      for (int e = threadIdx.x; e < CONV_WAVES * NBO * 16 * 2; e += CONV_THREADS) slots[e] = 0.0;
    }
    if (STATS == 2) {                        // (mean, invstd, gamma, beta) of this workgroup's OUTPUT channels, behind the slots
      float4 *outp = reinterpret_cast<float4 *>(reinterpret_cast<double *>(conv_in + nbi * 16) + CONV_WAVES * NBO * 16 * 2);
      for (int c = threadIdx.x; c < NBO * 16; c += CONV_THREADS) {
        const int co = 16 * ob0 + c;
        outp[c] = co < Cout ? make_float4(in_mean[co], in_invstd[co], in_gamma ? in_gamma[co] : 1.f, in_beta ? in_beta[co] : 0.f)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else if (in_mean != nullptr)
      for (int c = threadIdx.x; c < nbi * 16; c += CONV_THREADS)
        conv_in[c] = c < Cin ? make_float4(in_mean[c], in_invstd[c], in_gamma ? in_gamma[c] : 1.f, in_beta ? in_beta[c] : 0.f)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
  }
  const float4 *conv_in = conv_w + NBO * nbi * WAVE;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // provably uniform
  const int g = lane >> 4, j = lane & 15;
  const int tpb = (P + 31) >> 5;             // 32-pixel tiles per batch element: lane (g, j) owns pixels 2j, 2j + 1
  const long long tiles = (long long)B * tpb;
  const long long tstep = (long long)gridDim.x * CONV_WAVES;
  const long long t0 = (long long)blockIdx.x * CONV_WAVES + wave;
  if (!STATS && t0 >= tiles) return;          // (STATS: every wave reaches the barrier below; its loops run zero times)
  // The wave's work is one stream of (tile, 16-channel block) steps.  A load cursor runs CONV_AHEAD steps ahead of
  // the multiply cursor through a ring of register buffers, so a block has CONV_AHEAD multiply steps (of this wave
  // and of the waves sharing its SIMD) to arrive from HBM -- tile boundaries included.
  // x is read through a buffer descriptor (the launcher guarantees < 4 GiB): a lane whose pixel or channel does not exist
  // asks for an offset beyond the records and gets zeros back, so the loads are UNCONDITIONAL instructions.  (With
  // conditional global loads every ring slot sat under control flow and the compiler waited for ALL outstanding loads --
  // s_waitcnt vmcnt(0) -- before each multiply step: the 4-deep ring hid one step of latency instead of four.)
  // The loads and their waits are inline assembly: hipcc's own wait insertion, even for unconditional buffer loads in this
  // loop, waits at every step until the step's OWN four loads are the newest outstanding ones (s_waitcnt vmcnt(3..0)), i.e.
  // until the three younger ring slots have arrived as well -- a prefetch distance of one step instead of four (profile:
  // the 128-channel layers ran at 2.0x their MFMA bound).  Here a step waits for `vmcnt(4 * (CONV_AHEAD - 1))`: loads return
  // in order, every step issues exactly four (out-of-range lanes included), so at most the 12 younger loads may still be
  // in flight when ring[u] is read; stores and the compiler's own loads only make that wait more conservative.
  typedef int cv_i32x4 __attribute__((ext_vector_type(4)));
  typedef float cv_f32x2 __attribute__((ext_vector_type(2)));
  const unsigned long long xaddr = reinterpret_cast<unsigned long long>(x);
  const cv_i32x4 xr = {__builtin_amdgcn_readfirstlane((int)(unsigned)xaddr),
                       __builtin_amdgcn_readfirstlane((int)(unsigned)((xaddr >> 32) & 0xFFFFull)),   // stride 0
                       __builtin_amdgcn_readfirstlane((int)x_bytes), 0x00020000};
  static_assert(LEAN || !STATS, "the statistics epilogue is part of the short one");
  constexpr bool lean = LEAN;
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(y, 0, (int)y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t bxr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(STATS == 2 ? bx : y), 0,
                                                                         (int)y_bytes, 0x00020000);
  const float4 *outp = reinterpret_cast<const float4 *>(
      reinterpret_cast<const double *>(conv_in + nbi * 16) + CONV_WAVES * NBO * 16 * 2);
  // Cursor arithmetic is scalar: the tile number, its cloud b and its position r inside the cloud are wave-uniform (the wave
  // index above comes through readfirstlane), and moving to the wave's next tile adds the precomputed quotient / remainder
  // of the stride instead of dividing again (a 64-bit division per tile on the VECTOR pipe, twice, in the first version).
  // Per lane a tile costs one offset: xoff = byte offset of (b, channel g, pixel pair), or out of range.
  const int qstep = (int)(tstep / tpb), rstep = (int)(tstep - (long long)qstep * tpb);
  const int row_bytes_x = P * 4;
  struct Cursor { long long t; int m, b, r; unsigned xoff; int px; bool pv; };
  auto locate = [&](Cursor &c) {
    const int px = (c.r << 5) + 2 * j;
    c.pv = c.t < tiles && px < P;
    c.xoff = c.pv ? (unsigned)((((long long)c.b * Cin + g) * P + px) * 4) : 0xFFFFFFF0u;
    c.px = px;
  };
  auto advance = [&](Cursor &c) {
    if (++c.m == nbi) {
      c.m = 0;
      c.t += tstep;
      c.b += qstep;
      c.r += rstep;
      if (c.r >= tpb) { c.r -= tpb; ++c.b; }
      locate(c);
    }
  };
  // channel tail (Cin not a multiple of 16): in the LAST 16-channel block the lanes of row group g read channel
  // 16 (nbi - 1) + 4 s + g; those beyond Cin ask out of range
  unsigned tail[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) tail[s4] = (16 * (nbi - 1) + 4 * s4 + g < Cin) ? 0u : 0xFFFFFFF0u;
  cv_f32x2 ring[CONV_AHEAD][4];
  auto load = [&](const Cursor &c, cv_f32x2(&d)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // lane part in the vector offset (| the tail mask in the last block), the block / row part (16 m + 4 s) rows as the
      // scalar offset: one vector instruction per load
      const unsigned off = c.xoff | (c.m == nbi - 1 ? tail[s] : 0u);
      const int soff = (16 * c.m + 4 * s) * row_bytes_x;
      asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(d[s]) : "v"(off), "s"(xr), "s"(soff));
    }
  };
  const int b0 = (int)(t0 / tpb), r0 = (int)(t0 - (long long)b0 * tpb);
  Cursor lc{t0, 0, b0, r0, 0u, 0, false}, mc{t0, 0, b0, r0, 0u, 0, false};
  locate(lc);
  locate(mc);
#pragma unroll
  for (int u = 0; u < CONV_AHEAD; ++u) {
    load(lc, ring[u]);
    advance(lc);
  }
  cv_f32x4 acc[NBO][2];
#pragma unroll
  for (int o = 0; o < NBO; ++o) acc[o][0] = acc[o][1] = cv_f32x4{0.f, 0.f, 0.f, 0.f};
  float st_s[STATS ? NBO : 1][4], st_q[STATS ? NBO : 1][4];
  int st_tiles = 0;
  if (STATS) {
#pragma unroll
    for (int o = 0; o < NBO; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) st_s[o][r] = st_q[o][r] = 0.f;
  }
  double *st_slot = reinterpret_cast<double *>(const_cast<float4 *>(conv_in) + nbi * 16) + (size_t)wave * NBO * 16 * 2;
  auto st_flush = [&]() {
    struct AddF { __device__ __forceinline__ unsigned operator()(unsigned a, unsigned b) const {
      return __float_as_uint(__uint_as_float(a) + __uint_as_float(b)); } };
#pragma unroll
    for (int o = 0; o < (STATS ? NBO : 0); ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ss = __uint_as_float(row16_allreduce_u32(__float_as_uint(st_s[o][r]), AddF()));
        const float qq = __uint_as_float(row16_allreduce_u32(__float_as_uint(st_q[o][r]), AddF()));
        if (j == 0) {                        // the row's channel 16 o + 4 g + r: this lane alone owns the slot
          double *d = st_slot + (o * 16 + 4 * g + r) * 2;
          d[0] += (double)ss;
          d[1] += (double)qq;
        }
        st_s[o][r] = st_q[o][r] = 0.f;
      }
    st_tiles = 0;
  };
  // (STATS: one extra trip after the last tile, so that the flush below exists ONCE in the code -- five inlined copies of
  // it pushed the ring buffer out of registers)
  for (bool more = true; more;) {
    more = mc.t < tiles;
    // past the last tile: the ring's final (unused, out-of-range) loads must land before their registers are reused
    if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < CONV_AHEAD; ++u) {
      if (mc.t < tiles) {                    // wave-uniform
        static_assert(CONV_AHEAD == 4, "the wait below counts 4 * (CONV_AHEAD - 1) younger loads");
        asm volatile("s_waitcnt vmcnt(12)" : "+v"(ring[u][0]), "+v"(ring[u][1]), "+v"(ring[u][2]), "+v"(ring[u][3]));
        if (mc.m == nbi - 1 && (Cin & 15) != 0) {      // wave-uniform.  Channel tail: exact zeros whatever the descriptor's
#pragma unroll                                        // range check makes of vector + scalar offset (0 x NaN would be NaN)
          for (int s = 0; s < 4; ++s)
            if (tail[s] != 0u) ring[u][s] = cv_f32x2{0.f, 0.f};
        }
        if (STATS != 2 && in_mean != nullptr) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float4 t = conv_in[16 * mc.m + 4 * s + g];
            ring[u][s].x = relu_nan(((ring[u][s].x - t.x) * t.y) * t.z + t.w);
            ring[u][s].y = relu_nan(((ring[u][s].y - t.x) * t.y) * t.z + t.w);
          }
        }
#pragma unroll
        for (int o = 0; o < NBO; ++o) {
          const float4 a = conv_w[(o * nbi + mc.m) * WAVE + lane];
          const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            acc[o][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], ring[u][s].x, acc[o][0], 0, 0, 0);
            acc[o][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], ring[u][s].y, acc[o][1], 0, 0, 0);
          }
        }
        load(lc, ring[u]);                   // the slot just consumed takes the step CONV_AHEAD ahead
        advance(lc);
        if (mc.m == nbi - 1) {
          if (lean) {
            // Training forward / input gradient (no folded BatchNorm, no pooling): y through its descriptor.  The lane's
            // part of the address (cloud, pixel pair, row 4 g of the block) is ONE 32-bit offset per tile; the (block, row)
            // part is wave-uniform and travels as the store's scalar offset -- 2 moves + 1 store per row instead of the
            // ~37 vector instructions (64-bit address, guards, selects) of the general epilogue below, which was a third
            // of this kernel's vector-pipe time at 128 channels (the fp32 MFMA does not share the pipe).
            if (mc.pv) {
              const unsigned yoff = (unsigned)((((long long)mc.b * Cout + 16 * ob0 + 4 * g) * P + mc.px) * 4);
              const int row_bytes = P * 4;
              if (STATS == 2) {
                // sums of the BatchNorm backward, two output blocks at a time: all their bx loads first, then the arithmetic
                typedef unsigned cv_u32x2 __attribute__((ext_vector_type(2)));
                constexpr int HB = NBO < 2 ? NBO : 2;
#pragma unroll
                for (int o0 = 0; o0 < NBO; o0 += HB) {
                  cv_u32x2 xv[HB][4];
#pragma unroll
                  for (int oo = 0; oo < HB; ++oo)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                      if (o0 + oo < NBO)       // rows past Cout: in range of the tensor or not, their da is exactly 0
                        xv[oo][r] = __builtin_amdgcn_raw_buffer_load_b64(bxr, (int)yoff, (16 * (o0 + oo) + r) * row_bytes, 0);
#pragma unroll
                  for (int oo = 0; oo < HB; ++oo)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                      if (o0 + oo < NBO) {
                        const int o = o0 + oo;
                        const float4 t = outp[o * 16 + 4 * g + r];
                        const float h0 = (__uint_as_float(xv[oo][r].x) - t.x) * t.y, h1 = (__uint_as_float(xv[oo][r].y) - t.x) * t.y;
                        const float g0 = (h0 * t.z + t.w > 0.f) ? acc[o][0][r] : 0.f;      // bn_partial_kernel's mask
                        const float g1 = (h1 * t.z + t.w > 0.f) ? acc[o][1][r] : 0.f;
                        st_s[o][r] += g0 + g1;
                        st_q[o][r] = fmaf(g1, h1, fmaf(g0, h0, st_q[o][r]));
                      }
                }
              }
#pragma unroll
              for (int o = 0; o < NBO; ++o) {
                const bool whole = 16 * (ob0 + o) + 16 <= Cout;            // wave-uniform: only the last block has a tail
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const float v0 = acc[o][0][r], v1 = acc[o][1][r];
                  if (STATS == 1) {           // padding channels hold exact zeros: they add nothing and are never read
                    st_s[o][r] += v0 + v1;
                    st_q[o][r] = fmaf(v1, v1, fmaf(v0, v0, st_q[o][r]));
                  }
                  if (whole || 16 * (ob0 + o) + 4 * g + r < Cout) {
                    typedef unsigned cv_u32x2 __attribute__((ext_vector_type(2)));
                    const cv_u32x2 bits = {__float_as_uint(v0), __float_as_uint(v1)};
                    __builtin_amdgcn_raw_buffer_store_b64(bits, yr, (int)yoff, (16 * o + r) * row_bytes, 0);
                  }
                }
              }
            }
          } else if (mc.pv) {
            // ep_pool = K > 0: the row of K consecutive pixels (K / 2 neighbouring lanes, rows never straddle a 32-pixel
            // tile) is reduced to its maximum and only that is written: y is (B, Cout, P / K)
            float *yb = ep_pool > 0 ? y + ((long long)mc.b * Cout * (P / ep_pool) + mc.px / ep_pool)
                                    : y + ((long long)mc.b * Cout * P + mc.px);
            const long long cstride = ep_pool > 0 ? P / ep_pool : P;
            const bool writer = ep_pool > 0 && (mc.px % ep_pool) == 0;
#pragma unroll
            for (int o = 0; o < NBO; ++o)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int co = 16 * (ob0 + o) + 4 * g + r;
                float v0 = acc[o][0][r], v1 = acc[o][1][r];
                if (ep_scale != nullptr && co < Cout) {   // eval-mode BatchNorm folded to y * scale + shift, optional ReLU
                  const float sc = ep_scale[co], sh = ep_shift[co];
                  v0 = v0 * sc + sh;
                  v1 = v1 * sc + sh;
                  if (ep_relu) { v0 = relu_nan(v0); v1 = relu_nan(v1); }
                }
                if (STATS) {                  // padding channels hold exact zeros: they add nothing and are never read
                  st_s[o][r] += v0 + v1;
                  st_q[o][r] = fmaf(v1, v1, fmaf(v0, v0, st_q[o][r]));
                }
                if (ep_pool > 0) {
                  float mx = max_nan(v0, v1);
                  for (int off = 1; off < (ep_pool >> 1); off <<= 1) mx = max_nan(mx, __shfl_xor(mx, off, 64));
                  if (writer && co < Cout) yb[(long long)co * cstride] = mx;
                } else if (co < Cout) {
                  *reinterpret_cast<float2 *>(yb + (long long)co * cstride) = make_float2(v0, v1);
                }
              }
          }
#pragma unroll
          for (int o = 0; o < NBO; ++o) acc[o][0] = acc[o][1] = cv_f32x4{0.f, 0.f, 0.f, 0.f};
          if (STATS) ++st_tiles;
        }
        advance(mc);
      }
    }
    if (STATS && (st_tiles >= CONV_STAT_TILES || !more)) st_flush();     // at most CONV_AHEAD - 1 tiles over the limit
  }

  if (STATS) {
    __syncthreads();
    const double *slots = reinterpret_cast<const double *>(conv_in + nbi * 16);
    for (int e = threadIdx.x; e < NBO * 16 * 2; e += CONV_THREADS) {
      const int co = 16 * ob0 + (e >> 1);
      if (co >= Cout) continue;
      double t = 0.0;
#pragma unroll
      for (int wv = 0; wv < CONV_WAVES; ++wv) t += slots[(size_t)wv * NBO * 16 * 2 + e];
      stats_partial[((size_t)co * gridDim.x + blockIdx.x) * 2 + (e & 1)] = t;
    }
  }
}

// dW(co, ci) partial sums of one workgroup over its pixel chunks.  Chunk = CP pixels of one batch element (CP a
// multiple of 32, runtime): rows of dY (Cout) and X (Cin) staged in LDS with a row stride of CP + 4 floats (the
// 16 rows of an operand read then fall into 16 different bank groups).  Waves: `ph` pixel phases x (wo x wm) tile
// workers (ph * wo * wm = 8); worker (a, c) owns the RO x RM rectangle of 16x16 tiles {(a + wo * r, c + wm * q)}
// -- RO + RM operand reads feed RO * RM * 4 MFMAs per 16 pixels -- over the 16-pixel sub-chunks phase, phase + ph,
// ...; the phases are summed in LDS and the workgroup writes partial row blockIdx.x.
template <int RO, int RM, bool XF>     // XF: X = max(bn(x), 0) applied while staging (in_mean ... in_beta)
__global__ __launch_bounds__(CONV_THREADS, ((RO * RM <= 2 && !XF) ? 4 : 2)) void conv1x1_wgrad_kernel(
    int B, int Cin, int Cout, int P, int CP, int ph, int wo, int wm, const float *__restrict__ dy,
    const float *__restrict__ x, float *__restrict__ partial, const float *__restrict__ in_mean,
    const float *__restrict__ in_invstd, const float *__restrict__ in_gamma, const float *__restrict__ in_beta) {
  extern __shared__ float conv_s[];          // [2][(Cout + Cin) rows][CP + 4]
  const int rows = Cout + Cin, ld = CP + 4;
  const int nbo = (Cout + 15) >> 4, nbi = (Cin + 15) >> 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, i = lane & 15;
  const int tgw = wo * wm, phase = wave / tgw, tw = wave - phase * tgw;
  const int wa = tw / wm, wc = tw - wa * wm;
  const int cpb = (P + CP - 1) / CP;         // chunks per batch element
  const long long chunks = (long long)B * cpb;
  const int vec = CP >> 2;                   // float4 per staged row
  cv_f32x4 acc[RO][RM];
  int aoff[RO], boff[RM];                    // LDS offsets of this lane's operand rows, -1 = padding row / no tile
#pragma unroll
  for (int r = 0; r < RO; ++r) {
    const int ro = 16 * (wa + wo * r) + i;
    aoff[r] = (wa + wo * r < nbo && ro < Cout) ? ro * ld + 4 * g : -1;
#pragma unroll
    for (int q = 0; q < RM; ++q) acc[r][q] = cv_f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int q = 0; q < RM; ++q) {
    const int rm = 16 * (wc + wm * q) + i;
    boff[q] = (wc + wm * q < nbi && rm < Cin) ? (Cout + rm) * ld + 4 * g : -1;
  }

  // Staging plan of this thread, fixed for the whole kernel: element e = threadIdx.x + u * 512 of a chunk is float4 q of
  // row r (dY rows first, then X rows).  Its source address without the chunk's (cloud, first pixel) part and its place in
  // the stage are computed ONCE instead of dividing e by the (runtime) row length for every element of every chunk.
  // (Measured, profiles/r03: -1.5 % only; so was a second chunk of register prefetch behind hand-placed vmcnt waits, which
  // cost the narrow layers their second workgroup per CU -- the kernel's 0.55 matrix-pipe busy is not a staging problem.)
  float4 pre[WGRAD_MAXV];                   // the next chunk on its way from HBM while this one is multiplied
  const float *src0[WGRAD_MAXV];            // row start + 4 q of cloud 0 (nullptr: no element)
  int dst0[WGRAD_MAXV], qpx[WGRAD_MAXV], xrow[WGRAD_MAXV];   // stage offset, first pixel inside the chunk, X channel or -1
#pragma unroll
  for (int u = 0; u < WGRAD_MAXV; ++u) {
    const int e = threadIdx.x + u * CONV_THREADS;
    const int r = e / vec, q = e - r * vec;
    const bool any = e < rows * vec;
    const bool is_dy = r < Cout;
    src0[u] = !any ? nullptr : is_dy ? dy + (long long)r * P + 4 * q : x + (long long)(r - Cout) * P + 4 * q;
    dst0[u] = r * ld + 4 * q;
    qpx[u] = 4 * q;
    xrow[u] = (any && !is_dy) ? r - Cout : -1;
  }
  const long long cloud_dy = (long long)Cout * P, cloud_x = (long long)Cin * P;
  auto fetch = [&](long long c) {
    const int b = (int)(c / cpb);           // (wave-uniform: scalar division)
    const int px0 = (int)(c - (long long)b * cpb) * CP;
    const long long add_dy = b * cloud_dy + px0, add_x = b * cloud_x + px0;
#pragma unroll
    for (int u = 0; u < WGRAD_MAXV; ++u) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (src0[u] != nullptr && px0 + qpx[u] < P)
        v = *reinterpret_cast<const float4 *>(src0[u] + (xrow[u] < 0 ? add_dy : add_x));
      pre[u] = v;
    }
  };
  // XF: per-input-channel (mean, invstd, gamma, beta) of the previous layer's BatchNorm, in LDS behind the two stages;
  // X = max(bn(x), 0) is applied when a fetched chunk is written to its stage (pixels beyond the row stay 0)
  float *zero_row = conv_s + (size_t)2 * rows * ld;                        // 16 zeros: operand of padding rows / tiles
  float4 *xf = reinterpret_cast<float4 *>(zero_row + 16);
  if (threadIdx.x < 16) zero_row[threadIdx.x] = 0.f;
  if (XF) {
    for (int ci = threadIdx.x; ci < Cin; ci += CONV_THREADS)
      xf[ci] = make_float4(in_mean[ci], in_invstd[ci], in_gamma ? in_gamma[ci] : 1.f, in_beta ? in_beta[ci] : 0.f);
  }
  __syncthreads();
  auto put = [&](int buf, long long c) {
    float *dst = conv_s + (size_t)buf * rows * ld;
    const int px0 = (int)(c - (c / cpb) * cpb) * CP;
#pragma unroll
    for (int u = 0; u < WGRAD_MAXV; ++u) {
      if (src0[u] != nullptr) {
        float4 v = pre[u];
        if (XF && xrow[u] >= 0 && px0 + qpx[u] < P) {
          const float4 t = xf[xrow[u]];
          v.x = relu_nan(((v.x - t.x) * t.y) * t.z + t.w);
          v.y = relu_nan(((v.y - t.x) * t.y) * t.z + t.w);
          v.z = relu_nan(((v.z - t.x) * t.y) * t.z + t.w);
          v.w = relu_nan(((v.w - t.x) * t.y) * t.z + t.w);
        }
        *reinterpret_cast<float4 *>(dst + dst0[u]) = v;
      }
    }
  };

  long long c = blockIdx.x;
  int buf = 0;
  if (c < chunks) {
    fetch(c);
    put(0, c);
  }
  __syncthreads();
  for (; c < chunks; c += gridDim.x) {
    const long long cn = c + gridDim.x;
    if (cn < chunks) fetch(cn);
    const float *sd = conv_s + (size_t)buf * rows * ld;
    // operand reads are unconditional (a padding row / absent tile reads the zero row) and one sub-chunk ahead of the
    // multiplies: the 16-pixel step used to start with its reads and an s_waitcnt lgkmcnt(0)
    float4 av[RO], bv[RM];
    auto read_ops = [&](int sc, float4(&a)[RO], float4(&bb)[RM]) {
      const float *sp = sd + 16 * sc;
#pragma unroll
      for (int r = 0; r < RO; ++r) a[r] = *reinterpret_cast<const float4 *>(aoff[r] >= 0 ? sp + aoff[r] : zero_row + 4 * g);
#pragma unroll
      for (int q = 0; q < RM; ++q) bb[q] = *reinterpret_cast<const float4 *>(boff[q] >= 0 ? sp + boff[q] : zero_row + 4 * g);
    };
    const int nsc = CP >> 4;
    if (phase < nsc) read_ops(phase, av, bv);
    for (int sc = phase; sc < nsc; sc += ph) {
      float4 an[RO], bn[RM];
      const int scn = sc + ph < nsc ? sc + ph : sc;          // (the last step re-reads its own operands: no branch)
      read_ops(scn, an, bn);
#pragma unroll
      for (int r = 0; r < RO; ++r)
#pragma unroll
        for (int q = 0; q < RM; ++q) {
          acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].x, bv[q].x, acc[r][q], 0, 0, 0);
          acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].y, bv[q].y, acc[r][q], 0, 0, 0);
          acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].z, bv[q].z, acc[r][q], 0, 0, 0);
          acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r].w, bv[q].w, acc[r][q], 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < RO; ++r) av[r] = an[r];
#pragma unroll
      for (int q = 0; q < RM; ++q) bv[q] = bn[q];
    }
    if (cn < chunks) put(buf ^ 1, cn);      // the other buffer was last read before the previous barrier
    __syncthreads();
    buf ^= 1;
  }
  if (ph > 1) {
    // the pixel phases of a tile are summed here, in phase order (ph > 1 only for <= 4 tiles of one per worker)
    cv_f32x4 *red = reinterpret_cast<cv_f32x4 *>(conv_s);   // the staging buffers are free after the loop's last barrier
    red[(phase * tgw + tw) * WAVE + lane] = acc[0][0];
    __syncthreads();
    if (phase != 0) return;
    for (int q = 1; q < ph; ++q) acc[0][0] += red[(q * tgw + tw) * WAVE + lane];
  }
  float *out = partial + (size_t)blockIdx.x * Cout * Cin;
#pragma unroll
  for (int r = 0; r < RO; ++r)
#pragma unroll
    for (int q = 0; q < RM; ++q) {
      const int o = wa + wo * r, m = wc + wm * q;
      if (o < nbo && m < nbi) {
        const int ci = 16 * m + i;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int co = 16 * o + 4 * g + k;
          if (co < Cout && ci < Cin) out[(size_t)co * Cin + ci] = acc[r][q][k];
        }
      }
    }
}

// dw[e] = sum over the partial rows in a fixed order: 16 row groups per element (rows q = rg, rg + 16, ...), each
// summed in row order, then combined by a fixed binary tree.  Block = 32 elements x 16 row groups.
__global__ __launch_bounds__(512) void conv1x1_wgrad_reduce_kernel(int n, int nparts, const float *__restrict__ partial,
                                                                  float *__restrict__ dw) {
  __shared__ float part[16][33];
  const int el = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  float s = 0.f;
  if (e < n)
    for (int q = rg; q < nparts; q += 16) s += partial[(size_t)q * n + e];
  part[rg][el] = s;
  __syncthreads();
  for (int half = 8; half >= 1; half >>= 1) {
    if (rg < half) part[rg][el] += part[rg + half][el];
    __syncthreads();
  }
  if (rg == 0 && e < n) dw[e] = part[0][el];
}

static int conv_grid_x() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

struct WgradPlan { int cp, ph, wo, wm, ro, rm, grid; size_t lds; };

// Split of the 8 waves: `ph` pixel phases when there are fewer than 5 tiles (one tile per worker then), else
// wo x wm = 8 workers over the (nbo, nbi) tile grid with the fewest operand reads per MFMA; ro in {1,2,4},
// rm in {1,2,3,4} (the instantiated rectangles).  ro = 0: not covered.
static WgradPlan wgrad_plan(int b, int cin, int cout, int p) {
  WgradPlan pl;
  const int rows = cin + cout;
  const int nbo = ceil_div(cout, 16), nbi = ceil_div(cin, 16), ntiles = nbo * nbi;
  pl.ph = ntiles >= 5 ? 1 : ntiles >= 3 ? 2 : ntiles == 2 ? 4 : 8;
  const int tgw = CONV_WAVES / pl.ph;
  pl.wo = pl.wm = pl.ro = pl.rm = 0;
  int best = 1 << 30;
  for (int wo = 1; wo <= tgw; wo *= 2) {
    const int wm = tgw / wo;
    int ro = ceil_div(nbo, wo), rm = ceil_div(nbi, wm);
    ro = ro <= 2 ? ro : ro <= 4 ? 4 : 99;
    if (ro > 4 || rm > 4 || (pl.ph > 1 && ro * rm != 1)) continue;
    const int cost = ro * rm * 16 + ro + rm;      // MFMA slots per wave (padding tiles cost time) first, then operand reads
    if (cost < best) {
      best = cost;
      pl.wo = wo; pl.wm = wm; pl.ro = ro; pl.rm = rm;
    }
  }
  // chunk: as many pixels as keep the double-buffered stage under ~96 KiB, 32..512, no longer than a row
  int cp = 512;
  auto fits = [&](int c, size_t kib) {
    return (size_t)2 * rows * (c + 4) * 4 <= kib * 1024 && (long long)rows * (c / 4) <= (long long)WGRAD_MAXV * CONV_THREADS;
  };
  while (cp > 32 && !fits(cp, 96)) cp >>= 1;
  // many channels leave short chunks (256 rows: 32 pixels = one barrier per 128 MFMAs of a wave): there the stage may take
  // 144 KiB -- measured -4 ... -10 % on the 128-channel layers (174 -> 157 us at 128 x 128 x 12 288 pixels), nothing or a
  // lost second workgroup per CU on the narrow ones, which keep the 96 KiB rule
  if (cp <= 64 && fits(2 * cp, 144)) cp *= 2;
  while (cp > 32 && cp / 2 >= p) cp >>= 1;
  if (cp < 16 * pl.ph) cp = 16 * pl.ph;      // every phase has at least one 16-pixel sub-chunk
  pl.cp = cp;
  pl.lds = (size_t)2 * rows * (cp + 4) * 4;
  // with pixel phases the staging area is reused for the phase sums: ph * tgw * 64 float4 (3 -> 4 channels
  // on <= 128 pixels reserved 7392 bytes for an 8192-byte reduction)
  if (pl.ph > 1 && pl.lds < (size_t)pl.ph * tgw * 64 * 16) pl.lds = (size_t)pl.ph * tgw * 64 * 16;
  const long long chunks = (long long)b * ceil_div(p, cp);
  const int per_cu = ((size_t)2 * pl.lds <= (size_t)150 * 1024 && pl.ro * pl.rm <= 2) ? 2 : 1;
  const long long want = (long long)conv_grid_x() * per_cu;
  pl.grid = (int)(chunks < want ? chunks : want);
  return pl;
}

// Kernels that need more than the default 64 KiB of dynamic LDS are allowed their maximum once per process (the
// attribute call costs tens of microseconds of host time: not something to pay per launch).
template <typename K>
static bool allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return true;
  static std::mutex mu;
  static const void *seen[64];
  static int nseen = 0;
  const void *fn = reinterpret_cast<const void *>(kernel);
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < nseen; ++i)
    if (seen[i] == fn) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
  if (nseen < 64) seen[nseen++] = fn;
  return true;
}

static bool conv_args_ok(const char *what, int b, int cin, int cout, int p, const void *p0, const void *p1,
                         const void *p2) {
  if (p % 4 != 0) {
    set_error(PWCLO_EINVAL, "%s: pixels per row p=%d must be a multiple of 4", what, p);
    return false;
  }
  if (cin > 512 || cout > 512) {
    set_error(PWCLO_EINVAL, "%s: cin=%d cout=%d exceed 512", what, cin, cout);
    return false;
  }
  if (((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15) != 0) {
    set_error(PWCLO_EINVAL, "%s: tensors must be 16-byte aligned", what);
    return false;
  }
  (void)b;
  return true;
}

}  // namespace pwclo

using namespace pwclo;

// Launch shape of the forward / input-gradient kernel: gy groups of nbo output blocks, gx persistent tile workers.
struct ConvGrid { int nbi, nbo, gy; long long gx; size_t lds; };
static ConvGrid conv_grid(int b, int cin, int cout, int p, bool stats, bool lean = false, bool bwd_sums = false) {
  ConvGrid cg;
  cg.nbi = ceil_div(cin, 16);
  const int nbo_all = ceil_div(cout, 16);
  cg.gy = ceil_div(nbo_all, CONV_MAX_NBO);                              // groups of output blocks (input re-read per group)
  // Few pixels (the coarse pyramid levels: 64 ... 2048 pixels per cloud): one 32-pixel tile per wave does not fill the chip,
  // and a wave that owns all 8 output blocks of a 192-channel layer issues 768 dependent-rate MFMAs (10 us) after its
  // workgroup has packed 96 KiB of weights.  Split the output channels over more workgroups instead -- narrower weight
  // slices, the (L2-resident) input tile re-read per group -- until the launch has at least half a workgroup per CU.  Same k order per
  // output element: bit-identical results.  (tools/conv_table.py: 40 -> ~15 us for 192 -> 128 channels at 64 pixels.)
  {
    const long long one_tile_per_wave = (( (long long)b * ceil_div(p, 32)) + CONV_WAVES - 1) / CONV_WAVES;
    static int split_env = -1;
    if (split_env < 0) { const char *e = getenv("PWCLO_CONV_SPLIT"); split_env = e ? atoi(e) : 1; }
    // (measured: worth it while the launch is under HALF a workgroup per CU; at 192 of 256 the split only adds input re-reads)
    while (split_env && cg.gy < nbo_all && 2 * one_tile_per_wave * cg.gy <= conv_grid_x()) cg.gy = min(nbo_all, cg.gy * 2);
  }
  cg.nbo = ceil_div(nbo_all, cg.gy);
  cg.gy = ceil_div(nbo_all, cg.nbo);
  cg.lds = (size_t)cg.nbo * cg.nbi * WAVE * sizeof(float4) + (size_t)cg.nbi * 16 * sizeof(float4);   // weights + input transform
  if (stats) cg.lds += (size_t)CONV_WAVES * cg.nbo * 16 * 2 * sizeof(double);                        // + the waves' fp64 slots
  if (bwd_sums) cg.lds += (size_t)cg.nbo * 16 * sizeof(float4);                                      // + the output channels' BatchNorm
  const long long tiles = (long long)b * ceil_div(p, 32);
  // workgroups a CU can hold (LDS, registers): the short-epilogue kernels without statistics stay under 128 VGPRs at every
  // width (4 waves per SIMD), with statistics up to 4 output blocks, the general epilogue up to 3
  const int per_cu = (cg.lds <= 72 * 1024 && (cg.nbo <= 3 || (lean && !stats) || (stats && cg.nbo <= 4))) ? 2 : 1;
  cg.gx = (long long)conv_grid_x() * per_cu / cg.gy;
  const long long need = (tiles + CONV_WAVES - 1) / CONV_WAVES;
  if (cg.gx > need) cg.gx = need;
  if (cg.gx < 1) cg.gx = 1;
  return cg;
}

static void conv1x1_launch(int b, int cin, int cout, int p, const float *x, const float *w, int transposed, float *y,
                           const float *scale, const float *shift, int relu, int pool = 0,
                           const float *in_mean = nullptr, const float *in_invstd = nullptr,
                           const float *in_gamma = nullptr, const float *in_beta = nullptr, double *stats = nullptr,
                           const float *bx = nullptr) {
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0) return;
  if (!conv_args_ok("conv1x1_forward", b, cin, cout, p, x, y, x)) return;
  PWCLO_REQUIRE((scale == nullptr) == (shift == nullptr), "conv1x1_forward: scale and shift must be given together%s", "");
  PWCLO_REQUIRE(pool == 0 || ((pool == 4 || pool == 8 || pool == 16 || pool == 32) && p % pool == 0),
                "conv1x1_forward: pooled rows of k=%d pixels need k in {4,8,16,32} dividing p=%d", pool, p);
  // transposed = 1: w is stored (cin, cout) row-major -- the input-gradient pass of a layer whose weight it is.
  const long long ld_o = transposed ? 1 : cin, ld_i = transposed ? cout : 1;
  // y through 32-bit offsets where it fits (0: the general epilogue with 64-bit addresses)
  const long long yb64 = (long long)b * cout * p * 4;
  const unsigned y_bytes = (pool == 0 && yb64 < (1ll << 32) - 16) ? (unsigned)yb64 : 0u;
  const bool lean = scale == nullptr && pool == 0 && y_bytes != 0u;
  PWCLO_REQUIRE(stats == nullptr || lean, "conv1x1_forward_bnstats: output of %lld bytes (the statistics epilogue addresses y with "
                "32-bit byte offsets)", yb64);
  const ConvGrid cg = conv_grid(b, cin, cout, p, stats != nullptr, lean, bx != nullptr);
  const int nbi = cg.nbi, nbo = cg.nbo, gy = cg.gy;
  const size_t lds = cg.lds;
  const long long gx = cg.gx;
  PWCLO_REQUIRE((in_mean == nullptr) == (in_invstd == nullptr), "conv1x1_forward: in_mean and in_invstd must be given together%s", "");
  PWCLO_REQUIRE(lds <= 154 * 1024, "conv1x1_forward: cin=%d cout=%d need %zu bytes of LDS for the weights", cin, cout, lds);
  PWCLO_REQUIRE((long long)b * cin * p * 4 < (1ll << 32) - 16, "conv1x1_forward: input of %lld bytes: the kernel addresses x with "
                "32-bit byte offsets (< 4 GiB)", (long long)b * cin * p * 4);
  const unsigned x_bytes = (unsigned)((long long)b * cin * p * 4);
  hipStream_t st = current_stream();
  dim3 grid((unsigned)gx, (unsigned)gy), block(CONV_THREADS);
#define PWCLO_CONV_LAUNCH_S(N, S, L)                                                                             \
    PWCLO_REQUIRE(allow_lds(conv1x1_kernel<N, S, L>, lds), "conv1x1_forward: cannot reserve %zu bytes of LDS", lds); \
    hipLaunchKernelGGL((conv1x1_kernel<N, S, L>), grid, block, lds, st, b, cin, cout, p, nbi, ld_o, ld_i, x, w, y, \
                       scale, shift, relu, pool, in_mean, in_invstd, in_gamma, in_beta, stats, x_bytes, y_bytes, bx);
#define PWCLO_CONV_LAUNCH(N)                                                                                     \
  case N:                                                                                                        \
    if (stats != nullptr && bx != nullptr) {                                                                     \
      if (N <= 4) { PWCLO_CONV_LAUNCH_S((N <= 4 ? N : 1), 2, true) }                                             \
    }                                                                                                            \
    else if (stats != nullptr) { PWCLO_CONV_LAUNCH_S(N, 1, true) }                                               \
    else if (lean) { PWCLO_CONV_LAUNCH_S(N, 0, true) }                                                           \
    else { PWCLO_CONV_LAUNCH_S(N, 0, false) }                                                                    \
    break
  switch (nbo) {
    PWCLO_CONV_LAUNCH(1);
    PWCLO_CONV_LAUNCH(2);
    PWCLO_CONV_LAUNCH(3);
    PWCLO_CONV_LAUNCH(4);
    PWCLO_CONV_LAUNCH(5);
    PWCLO_CONV_LAUNCH(6);
    PWCLO_CONV_LAUNCH(7);
    default:
      PWCLO_CONV_LAUNCH(8);
  }
#undef PWCLO_CONV_LAUNCH
#undef PWCLO_CONV_LAUNCH_S
  check_launch("conv1x1_forward");
}

extern "C" void conv1x1_forward_kernel_wrapper(int b, int cin, int cout, int p, const float *x, const float *w,
                                               int transposed, float *y) {
  conv1x1_launch(b, cin, cout, p, x, w, transposed, y, nullptr, nullptr, 0);
}

extern "C" void conv1x1_affine_forward_kernel_wrapper(int b, int cin, int cout, int p, const float *x, const float *w,
                                                      const float *scale, const float *shift, int relu, float *y) {
  conv1x1_launch(b, cin, cout, p, x, w, 0, y, scale, shift, relu);
}

extern "C" void conv1x1_bnrelu_forward_kernel_wrapper(int b, int cin, int cout, int p, const float *x, const float *w,
                                                      const float *in_mean, const float *in_invstd,
                                                      const float *in_gamma, const float *in_beta, float *y) {
  conv1x1_launch(b, cin, cout, p, x, w, 0, y, nullptr, nullptr, 0, 0, in_mean, in_invstd, in_gamma, in_beta);
}

// Training forward of conv -> BatchNorm: y = W x (x optionally max(bn_prev(x), 0) applied on load, as above), and the
// batch statistics of y -- save_mean, save_invstd, the momentum update of running_mean / running_var (nullable) -- from
// the convolution's own epilogue sums: what batchnorm_train_forward_kernel_wrapper(y = nullptr) computes with a pass over y.
// workspace: conv1x1_stats_workspace_bytes(b, cin, cout, p) bytes.
extern "C" long long conv1x1_stats_workspace_bytes(int b, int cin, int cout, int p) {
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0) return 0;
  const ConvGrid cg = conv_grid(b, cin, cout, p, true);
  return (long long)cout * cg.gx * 2 * (long long)sizeof(double);
}

extern "C" void conv1x1_forward_bnstats_kernel_wrapper(int b, int cin, int cout, int p, const float *x, const float *w,
                                                       const float *in_mean, const float *in_invstd,
                                                       const float *in_gamma, const float *in_beta, float *y, float eps,
                                                       float momentum, float *running_mean, float *running_var,
                                                       float *save_mean, float *save_invstd, void *workspace) {
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0) return;
  PWCLO_REQUIRE(workspace != nullptr && save_mean != nullptr && save_invstd != nullptr,
                "conv1x1_forward_bnstats: workspace, save_mean and save_invstd are required%s", "");
  PWCLO_REQUIRE((running_mean == nullptr) == (running_var == nullptr),
                "conv1x1_forward_bnstats: running_mean and running_var must be given together%s", "");
  const ConvGrid cg = conv_grid(b, cin, cout, p, true);
  PWCLO_REQUIRE(cg.lds <= 154 * 1024, "conv1x1_forward_bnstats: cin=%d cout=%d need %zu bytes of LDS", cin, cout, cg.lds);
  double *partial = reinterpret_cast<double *>(workspace);
  conv1x1_launch(b, cin, cout, p, x, w, 0, y, nullptr, nullptr, 0, 0, in_mean, in_invstd, in_gamma, in_beta, partial);
  bn_forward_finish_launch(cout, (int)cg.gx, (long long)b * p, eps, momentum, partial, running_mean, running_var, save_mean,
                           save_invstd);
}

// Input gradient of a layer that follows a training-mode BatchNorm (+ ReLU): da = W^T dy (w stored (cout, cin) like the
// forward's, dy (b, cout, p)) AND dgamma / dbeta (cin) of that BatchNorm -- sum over (b, p) of g * xhat and of g, g = da where
// the forward's ReLU let the activation through -- from the convolution's own epilogue; bn_x (b, cin, p) is the BatchNorm's
// input, mean / invstd its saved statistics, gamma / beta nullable.  What batchnorm_train_backward computes with a pass over
// bn_x and da; the caller finishes with batchnorm_train_backward_apply.  workspace: conv1x1_stats_workspace_bytes(b, cout, cin, p).
extern "C" void conv1x1_dgrad_bnstats_kernel_wrapper(int b, int cin, int cout, int p, const float *dy, const float *w,
                                                     const float *bn_x, const float *mean, const float *invstd,
                                                     const float *gamma, const float *beta, float *da, float *dgamma,
                                                     float *dbeta, void *workspace) {
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0) return;
  PWCLO_REQUIRE(workspace != nullptr && bn_x != nullptr && mean != nullptr && invstd != nullptr && dgamma != nullptr &&
                    dbeta != nullptr,
                "conv1x1_dgrad_bnstats: workspace, bn_x, mean, invstd, dgamma and dbeta are required%s", "");
  PWCLO_REQUIRE((reinterpret_cast<uintptr_t>(bn_x) & 15) == 0, "conv1x1_dgrad_bnstats: bn_x must be 16-byte aligned%s", "");
  // the kernel's "input" is dy (cout rows), its "output" da (cin rows): the transposed view of the forward
  const ConvGrid cg = conv_grid(b, cout, cin, p, true, true, true);
  // up to 64 BatchNorm channels: beyond that the epilogue's registers (sums + accumulators + the bx rows) do not fit, and those
  // layers' tensors are small (the reduction pass costs where channels are few and rows are long)
  PWCLO_REQUIRE(cg.nbo <= 4, "conv1x1_dgrad_bnstats: cin=%d: at most 64 channels per workgroup (the caller keeps the separate "
                "reduction pass beyond)", cin);
  PWCLO_REQUIRE(cg.lds <= 154 * 1024, "conv1x1_dgrad_bnstats: cin=%d cout=%d need %zu bytes of LDS", cin, cout, cg.lds);
  double *partial = reinterpret_cast<double *>(workspace);
  conv1x1_launch(b, cout, cin, p, dy, w, 1, da, nullptr, nullptr, 0, 0, mean, invstd, gamma, beta, partial, bn_x);
  bn_backward_finish_launch(cin, (int)cg.gx, partial, dgamma, dbeta);
}

extern "C" void conv1x1_affine_maxk_forward_kernel_wrapper(int b, int cin, int cout, int s, int k, const float *x,
                                                           const float *w, const float *scale, const float *shift,
                                                           int relu, float *pooled) {
  conv1x1_launch(b, cin, cout, s * k, x, w, 0, pooled, scale, shift, relu, k);
}

extern "C" long long conv1x1_wgrad_workspace_bytes(int b, int cin, int cout, int p) {
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0) return 0;
  const WgradPlan pl = wgrad_plan(b, cin, cout, p);
  return (long long)pl.grid * cin * cout * (long long)sizeof(float);
}

static void conv1x1_wgrad_launch(int b, int cin, int cout, int p, const float *dy, const float *x, float *dw,
                                 void *workspace, const float *in_mean, const float *in_invstd,
                                 const float *in_gamma, const float *in_beta) {
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0) return;
  if (!conv_args_ok("conv1x1_wgrad", b, cin, cout, p, dy, x, workspace)) return;
  const WgradPlan pl = wgrad_plan(b, cin, cout, p);
  PWCLO_REQUIRE(pl.lds <= 150 * 1024, "conv1x1_wgrad: cin=%d cout=%d need %zu bytes of LDS", cin, cout, pl.lds);
  PWCLO_REQUIRE((long long)(cin + cout) * (pl.cp / 4) <= (long long)WGRAD_MAXV * CONV_THREADS,
                "conv1x1_wgrad: cin=%d cout=%d: chunk of %d pixels exceeds the staging registers", cin, cout, pl.cp);
  hipStream_t st = current_stream();
  float *partial = reinterpret_cast<float *>(workspace);
  PWCLO_REQUIRE(pl.ro > 0, "conv1x1_wgrad: cin=%d cout=%d: more than 4 x 4 tiles of 16 x 16 per wave", cin, cout);
  const size_t xf_lds = 64 + (in_mean != nullptr ? (size_t)cin * sizeof(float4) : 0);   // the zero row + the input transform's parameters
#define PWCLO_WGRAD_LAUNCH_X(R, M, X)                                                                                  \
  {                                                                                                                    \
    PWCLO_REQUIRE(allow_lds(conv1x1_wgrad_kernel<R, M, X>, pl.lds + xf_lds), "conv1x1_wgrad: cannot reserve %zu bytes of LDS",  \
                  pl.lds);                                                                                             \
    hipLaunchKernelGGL((conv1x1_wgrad_kernel<R, M, X>), dim3(pl.grid), dim3(CONV_THREADS), pl.lds + xf_lds, st, b, cin, cout, p, \
                       pl.cp, pl.ph, pl.wo, pl.wm, dy, x, partial, in_mean, in_invstd, in_gamma, in_beta);             \
  }
#define PWCLO_WGRAD_LAUNCH(R, M)                                                                                       \
  if (pl.ro == R && pl.rm == M) {                                                                                      \
    if (in_mean != nullptr) PWCLO_WGRAD_LAUNCH_X(R, M, true) else PWCLO_WGRAD_LAUNCH_X(R, M, false)                    \
  }
  PWCLO_WGRAD_LAUNCH(1, 1) PWCLO_WGRAD_LAUNCH(1, 2) PWCLO_WGRAD_LAUNCH(1, 3) PWCLO_WGRAD_LAUNCH(1, 4)
  PWCLO_WGRAD_LAUNCH(2, 1) PWCLO_WGRAD_LAUNCH(2, 2) PWCLO_WGRAD_LAUNCH(2, 3) PWCLO_WGRAD_LAUNCH(2, 4)
  PWCLO_WGRAD_LAUNCH(4, 1) PWCLO_WGRAD_LAUNCH(4, 2) PWCLO_WGRAD_LAUNCH(4, 3) PWCLO_WGRAD_LAUNCH(4, 4)
#undef PWCLO_WGRAD_LAUNCH_X
#undef PWCLO_WGRAD_LAUNCH
  const int n = cin * cout;
  hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel, dim3(ceil_div(n, 32)), dim3(512), 0, st, n, pl.grid, partial, dw);
  check_launch("conv1x1_wgrad");
}

extern "C" void conv1x1_wgrad_kernel_wrapper(int b, int cin, int cout, int p, const float *dy, const float *x,
                                             float *dw, void *workspace) {
  conv1x1_wgrad_launch(b, cin, cout, p, dy, x, dw, workspace, nullptr, nullptr, nullptr, nullptr);
}

extern "C" void conv1x1_bnrelu_wgrad_kernel_wrapper(int b, int cin, int cout, int p, const float *dy, const float *x,
                                                    const float *in_mean, const float *in_invstd,
                                                    const float *in_gamma, const float *in_beta, float *dw,
                                                    void *workspace) {
  PWCLO_REQUIRE(in_mean != nullptr && in_invstd != nullptr, "conv1x1_bnrelu_wgrad: in_mean and in_invstd are required%s", "");
  conv1x1_wgrad_launch(b, cin, cout, p, dy, x, dw, workspace, in_mean, in_invstd, in_gamma, in_beta);
}
