// Pointwise (1x1) convolution of the module path's shared MLPs on the fp32 matrix cores: forward, input gradient and
// weight gradient, directly on the (B, C, S, K) channel-major activations the reference's modules use
// (the layers are pytorch_utils.py:114-167 `Conv2d(kernel_size=(1,1), bias=False)` inside
// SharedMLP, pytorch_utils.py:12-37).  The library convolutions need NHWC transposes around their GEMMs for these
// shapes (6..192 input channels, 8..128 output channels, up to 2M pixels); these kernels read and write the
// channel-major rows as they are:
//   * every lane loads / stores 4 consecutive pixels of one channel row (16 bytes; 16 lanes cover 256 contiguous
//     bytes), which are the B operands (forward) of 4 independent 16-pixel MFMA columns;
//   * the weights sit in LDS in MFMA A-operand order, one ds_read_b128 feeds 16 MFMAs;
//   * the weight gradient stages (dY, X) pixel chunks through LDS once per workgroup and every wave accumulates its
//     own 16x16 (co, ci) tiles over the chunk; partial sums per workgroup are reduced in a fixed order by a second
//     kernel (deterministic, no atomics).
// v_mfma_f32_16x16x4_f32 everywhere: products and sums are fp32 FMAs, so the results differ from any other fp32
// convolution by summation order only.
#include "common.hpp"

namespace pwclo {

typedef float cv_f32x4 __attribute__((ext_vector_type(4)));

constexpr int CONV_THREADS = 512;            // 8 waves: 2 per SIMD
constexpr int CONV_WAVES = CONV_THREADS / WAVE;
constexpr int CONV_MAX_NBO = 8;              // 16-channel output blocks per workgroup (128 accumulator VGPRs)
constexpr int WGRAD_MAXV = 7;                // float4 per thread of one staged weight-gradient chunk

// y[b][co][p] = sum_ci W(co, ci) x[b][ci][p], W(co, ci) = w[co * w_ld_o + ci * w_ld_i]  (strides: the same kernel
// computes the input gradient with the transposed view).  P % 4 == 0, rows 16-byte aligned.
// grid (x = persistent tile workers, y = groups of NBO output blocks); dynamic LDS = NBO * nbi KiB.
template <int NBO>
__global__ __launch_bounds__(CONV_THREADS) void conv1x1_kernel(int B, int Cin, int Cout, int P, int nbi,
                                                              long long w_ld_o, long long w_ld_i,
                                                              const float *__restrict__ x,
                                                              const float *__restrict__ w, float *__restrict__ y) {
  extern __shared__ float4 conv_w[];         // [NBO][nbi][64 lanes] : the 4 k-steps of one (o, m) tile per lane
  const int ob0 = blockIdx.y * NBO;
  for (int e = threadIdx.x; e < NBO * nbi * WAVE; e += CONV_THREADS) {
    const int lane = e & 63, om = e >> 6;
    const int o = om / nbi, m = om - o * nbi;
    const int co = 16 * (ob0 + o) + (lane & 15), g = lane >> 4;
    float v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ci = 16 * m + 4 * s + g;
      v[s] = (co < Cout && ci < Cin) ? w[co * w_ld_o + ci * w_ld_i] : 0.f;
    }
    conv_w[e] = make_float4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int tpb = (P + 63) >> 6;             // 64-pixel tiles per batch element
  const long long tiles = (long long)B * tpb;
  for (long long t = (long long)blockIdx.x * CONV_WAVES + wave; t < tiles; t += (long long)gridDim.x * CONV_WAVES) {
    const int b = (int)(t / tpb);
    const int px = ((int)(t - (long long)b * tpb) << 6) + 4 * j;
    const bool pv = px < P;
    const float *xb = x + (long long)b * Cin * P + px;
    cv_f32x4 acc[NBO][4];
#pragma unroll
    for (int o = 0; o < NBO; ++o)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[o][p] = cv_f32x4{0.f, 0.f, 0.f, 0.f};
    float4 xa[4], xn[4];
    auto load = [&](int m, float4(&d)[4]) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int c = 16 * m + 4 * s + g;
        d[s] = (pv && c < Cin) ? *reinterpret_cast<const float4 *>(xb + (long long)c * P) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto mac = [&](int m, const float4(&d)[4]) {
#pragma unroll
      for (int o = 0; o < NBO; ++o) {
        const float4 a = conv_w[(o * nbi + m) * WAVE + lane];
        const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc[o][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], d[s].x, acc[o][0], 0, 0, 0);
          acc[o][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], d[s].y, acc[o][1], 0, 0, 0);
          acc[o][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], d[s].z, acc[o][2], 0, 0, 0);
          acc[o][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], d[s].w, acc[o][3], 0, 0, 0);
        }
      }
    };
    load(0, xa);
    for (int m = 0; m < nbi; m += 2) {
      if (m + 1 < nbi) load(m + 1, xn);
      mac(m, xa);
      if (m + 1 < nbi) {
        if (m + 2 < nbi) load(m + 2, xa);
        mac(m + 1, xn);
      }
    }
    if (pv) {
      float *yb = y + (long long)b * Cout * P + px;
#pragma unroll
      for (int o = 0; o < NBO; ++o)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = 16 * (ob0 + o) + 4 * g + r;
          if (co < Cout)
            *reinterpret_cast<float4 *>(yb + (long long)co * P) =
                make_float4(acc[o][0][r], acc[o][1][r], acc[o][2][r], acc[o][3][r]);
        }
    }
  }
}

// dW(co, ci) partial sums of one workgroup over its pixel chunks.  Chunk = CP pixels of one batch element (CP a
// multiple of 32, runtime): rows of dY (Cout) and X (Cin) staged in LDS with a row stride of CP + 4 floats (the
// 16 rows of an operand read then fall into 16 different bank groups).  Waves: `ph` pixel phases x `tgw` tile
// workers (ph * tgw = 8); wave (phase, tw) accumulates tiles tw, tw + tgw, ... over the 16-pixel sub-chunks
// phase, phase + ph, ...; it writes its sums as partial row (blockIdx.x * ph + phase).
template <int MAXT>
__global__ __launch_bounds__(CONV_THREADS) void conv1x1_wgrad_kernel(int B, int Cin, int Cout, int P, int CP, int ph,
                                                                    const float *__restrict__ dy,
                                                                    const float *__restrict__ x,
                                                                    float *__restrict__ partial) {
  extern __shared__ float conv_s[];          // [2][(Cout + Cin) rows][CP + 4]
  const int rows = Cout + Cin, ld = CP + 4;
  const int nbo = (Cout + 15) >> 4, nbi = (Cin + 15) >> 4, ntiles = nbo * nbi;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, i = lane & 15;
  const int tgw = CONV_WAVES / ph, phase = wave / tgw, tw = wave - phase * tgw;
  const int cpb = (P + CP - 1) / CP;         // chunks per batch element
  const long long chunks = (long long)B * cpb;
  const int vec = CP >> 2;                   // float4 per staged row
  cv_f32x4 acc[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) acc[t] = cv_f32x4{0.f, 0.f, 0.f, 0.f};

  float4 pre[WGRAD_MAXV];                   // the next chunk on its way from HBM while this one is multiplied
  auto fetch = [&](long long c) {
    const int b = (int)(c / cpb);
    const int px0 = (int)(c - (long long)b * cpb) * CP;
#pragma unroll
    for (int u = 0; u < WGRAD_MAXV; ++u) {
      const int e = threadIdx.x + u * CONV_THREADS;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < rows * vec) {
        const int r = e / vec, q = e - r * vec;
        const int px = px0 + 4 * q;
        if (px < P) {
          const float *src = (r < Cout) ? dy + ((long long)b * Cout + r) * P : x + ((long long)b * Cin + (r - Cout)) * P;
          v = *reinterpret_cast<const float4 *>(src + px);
        }
      }
      pre[u] = v;
    }
  };
  auto put = [&](int buf) {
    float *dst = conv_s + (size_t)buf * rows * ld;
#pragma unroll
    for (int u = 0; u < WGRAD_MAXV; ++u) {
      const int e = threadIdx.x + u * CONV_THREADS;
      if (e < rows * vec) {
        const int r = e / vec, q = e - r * vec;
        *reinterpret_cast<float4 *>(dst + (size_t)r * ld + 4 * q) = pre[u];
      }
    }
  };

  long long c = blockIdx.x;
  int buf = 0;
  if (c < chunks) {
    fetch(c);
    put(0);
  }
  __syncthreads();
  for (; c < chunks; c += gridDim.x) {
    const long long cn = c + gridDim.x;
    if (cn < chunks) fetch(cn);
    const float *sd = conv_s + (size_t)buf * rows * ld;
    const float *sx = sd + (size_t)Cout * ld;
    for (int sc = phase; sc < (CP >> 4); sc += ph) {
      const int off = 16 * sc + 4 * g;
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const int tile = tw + t * tgw;
        if (tile < ntiles) {
          const int o = tile / nbi, m = tile - o * nbi;
          const int ro = 16 * o + i, rm = 16 * m + i;
          const float4 a = (ro < Cout) ? *reinterpret_cast<const float4 *>(sd + (size_t)ro * ld + off)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 bq = (rm < Cin) ? *reinterpret_cast<const float4 *>(sx + (size_t)rm * ld + off)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq.x, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq.y, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq.z, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq.w, acc[t], 0, 0, 0);
        }
      }
    }
    if (cn < chunks) put(buf ^ 1);          // the other buffer was last read before the previous barrier
    __syncthreads();
    buf ^= 1;
  }
  float *out = partial + ((size_t)blockIdx.x * ph + phase) * Cout * Cin;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    const int tile = tw + t * tgw;
    if (tile < ntiles) {
      const int o = tile / nbi, m = tile - o * nbi;
      const int ci = 16 * m + i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = 16 * o + 4 * g + r;
        if (co < Cout && ci < Cin) out[(size_t)co * Cin + ci] = acc[t][r];
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

struct WgradPlan { int cp, ph, grid; size_t lds; };

static WgradPlan wgrad_plan(int b, int cin, int cout, int p) {
  WgradPlan pl;
  const int rows = cin + cout;
  const int ntiles = ceil_div(cout, 16) * ceil_div(cin, 16);
  pl.ph = ntiles >= 5 ? 1 : ntiles >= 3 ? 2 : ntiles == 2 ? 4 : 8;
  // chunk: as many pixels as keep the double-buffered stage under ~96 KiB, 32..512, no longer than a row
  int cp = 512;
  while (cp > 32 && (size_t)2 * rows * (cp + 4) * 4 > (size_t)96 * 1024) cp >>= 1;
  while (cp > 32 && cp / 2 >= p) cp >>= 1;
  if (cp < 16 * pl.ph) cp = 16 * pl.ph;      // every phase has at least one 16-pixel sub-chunk
  pl.cp = cp;
  pl.lds = (size_t)2 * rows * (cp + 4) * 4;
  const long long chunks = (long long)b * ceil_div(p, cp);
  const int cus = conv_grid_x();
  pl.grid = (int)(chunks < cus ? chunks : cus);
  return pl;
}

template <typename K>
static bool allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return true;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes) == hipSuccess;
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

extern "C" void conv1x1_forward_kernel_wrapper(int b, int cin, int cout, int p, const float *x, const float *w,
                                               int transposed, float *y) {
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0) return;
  if (!conv_args_ok("conv1x1_forward", b, cin, cout, p, x, y, x)) return;
  // transposed = 1: w is stored (cin, cout) row-major -- the input-gradient pass of a layer whose weight it is.
  const long long ld_o = transposed ? 1 : cin, ld_i = transposed ? cout : 1;
  const int nbi = ceil_div(cin, 16), nbo_all = ceil_div(cout, 16);
  const int gy = ceil_div(nbo_all, CONV_MAX_NBO);                       // groups of output blocks (input re-read per group)
  const int nbo = ceil_div(nbo_all, gy);
  const size_t lds = (size_t)nbo * nbi * WAVE * sizeof(float4);
  PWCLO_REQUIRE(lds <= 150 * 1024, "conv1x1_forward: cin=%d cout=%d need %zu bytes of LDS for the weights", cin, cout, lds);
  const long long tiles = (long long)b * ceil_div(p, 64);
  const int per_cu = lds <= 72 * 1024 ? 2 : 1;                          // workgroups a CU can hold
  long long gx = (long long)conv_grid_x() * per_cu / gy;
  const long long need = (tiles + CONV_WAVES - 1) / CONV_WAVES;
  if (gx > need) gx = need;
  if (gx < 1) gx = 1;
  hipStream_t st = current_stream();
  dim3 grid((unsigned)gx, (unsigned)gy), block(CONV_THREADS);
#define PWCLO_CONV_LAUNCH(N)                                                                                     \
  case N:                                                                                                        \
    PWCLO_REQUIRE(allow_lds(conv1x1_kernel<N>, lds), "conv1x1_forward: cannot reserve %zu bytes of LDS", lds);    \
    hipLaunchKernelGGL((conv1x1_kernel<N>), grid, block, lds, st, b, cin, cout, p, nbi, ld_o, ld_i, x, w, y);     \
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
  check_launch("conv1x1_forward");
}

extern "C" long long conv1x1_wgrad_workspace_bytes(int b, int cin, int cout, int p) {
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0) return 0;
  const WgradPlan pl = wgrad_plan(b, cin, cout, p);
  return (long long)pl.grid * pl.ph * cin * cout * (long long)sizeof(float);
}

extern "C" void conv1x1_wgrad_kernel_wrapper(int b, int cin, int cout, int p, const float *dy, const float *x,
                                             float *dw, void *workspace) {
  if (b <= 0 || cin <= 0 || cout <= 0 || p <= 0) return;
  if (!conv_args_ok("conv1x1_wgrad", b, cin, cout, p, dy, x, workspace)) return;
  const WgradPlan pl = wgrad_plan(b, cin, cout, p);
  PWCLO_REQUIRE(pl.lds <= 150 * 1024, "conv1x1_wgrad: cin=%d cout=%d need %zu bytes of LDS", cin, cout, pl.lds);
  PWCLO_REQUIRE((long long)(cin + cout) * (pl.cp / 4) <= (long long)WGRAD_MAXV * CONV_THREADS,
                "conv1x1_wgrad: cin=%d cout=%d: chunk of %d pixels exceeds the staging registers", cin, cout, pl.cp);
  hipStream_t st = current_stream();
  float *partial = reinterpret_cast<float *>(workspace);
  const int per = ceil_div(ceil_div(cout, 16) * ceil_div(cin, 16), CONV_WAVES / pl.ph);
  PWCLO_REQUIRE(per <= 12, "conv1x1_wgrad: cin=%d cout=%d: %d 16x16 tiles per wave, at most 12", cin, cout, per);
#define PWCLO_WGRAD_LAUNCH(T)                                                                                         \
  do {                                                                                                                \
    PWCLO_REQUIRE(allow_lds(conv1x1_wgrad_kernel<T>, pl.lds), "conv1x1_wgrad: cannot reserve %zu bytes of LDS", pl.lds); \
    hipLaunchKernelGGL((conv1x1_wgrad_kernel<T>), dim3(pl.grid), dim3(CONV_THREADS), pl.lds, st, b, cin, cout, p,      \
                       pl.cp, pl.ph, dy, x, partial);                                                                 \
  } while (0)
  if (per <= 1) PWCLO_WGRAD_LAUNCH(1);
  else if (per <= 2) PWCLO_WGRAD_LAUNCH(2);
  else if (per <= 4) PWCLO_WGRAD_LAUNCH(4);
  else if (per <= 8) PWCLO_WGRAD_LAUNCH(8);
  else PWCLO_WGRAD_LAUNCH(12);
#undef PWCLO_WGRAD_LAUNCH
  const int n = cin * cout;
  hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel, dim3(ceil_div(n, 32)), dim3(512), 0, st, n, pl.grid * pl.ph, partial, dw);
  check_launch("conv1x1_wgrad");
}
