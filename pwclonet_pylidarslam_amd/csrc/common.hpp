// Shared host/device helpers of libpwclo_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pwclo_ops.h"

namespace pwclo {

constexpr int WAVE = 64;

// ---- per-thread library state (stream + sticky error), defined in state.hip ----------------
hipStream_t current_stream();
void set_error(int code, const char *fmt, ...);
bool check_launch(const char *what);  // hipGetLastError() -> sticky error; true when OK
// batchnorm.hip: save_mean / save_invstd / running statistics of c channels from fp64 partial sums
// partial[(ch * nsplit + s) * 2 + {sum, sum of squares}] over M elements per channel (conv1x1.hip's statistics epilogue)
void bn_forward_finish_launch(int c, int nsplit, long long M, float eps, float momentum, const double *partial,
                              float *running_mean, float *running_var, float *save_mean, float *save_invstd);
// Device view of the library's pinned error word (state.hip): a kernel stores a PWCLO_E* code there with a
// system-scope atomic; pwclo_last_error() picks it up.  nullptr (and a sticky error) if it cannot be allocated.
unsigned *device_error_word();

// Argument guard used by the launchers: records PWCLO_EINVAL and makes the launcher return.
#define PWCLO_REQUIRE(cond, ...)                      \
  do {                                                \
    if (!(cond)) {                                    \
      ::pwclo::set_error(PWCLO_EINVAL, __VA_ARGS__);  \
      return;                                         \
    }                                                 \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// The reference's block-size rule (cuda_utils.h:15-19): 2^floor(log2(work)) clamped to [1,512].
// Integer form; agrees with the reference's double log() evaluation for every work_size >= 1
// that this library accepts (checked against the oracle in tests/test_host_logic.py).
static inline int ref_opt_n_threads(int work_size) {
  int p = 1;
  while (p * 2 <= work_size && p < 512) p *= 2;
  return p;
}

// ---- workgroup trace (developer tool, off unless pwclo_trace_enable() was called) ---------------------------
// rocprofv3's kernel trace serialises the pipelined run (3.7 ms per step instead of 2.26), so it cannot show how
// kernels of the four in-flight batches share the chip.  With the trace enabled every workgroup of the kernels on
// the fused forward path appends one record (kernel id, block, start / end of the constant 100 MHz clock,
// hardware id) to a buffer the caller owns; tools/wgtrace.py turns that into per-kernel CU-time and a concurrency
// timeline of the REAL pipelined run.  Cost when off: one pointer test per workgroup.
struct TraceRec { unsigned long long t0, t1; unsigned kernel, block, nblocks, hw; };   // 32 bytes
struct TraceBuf { TraceRec *rec; unsigned *count; unsigned cap; };
enum TraceKernel : unsigned {
  TK_INGEST = 1, TK_FPS = 2, TK_KNN = 3, TK_KNN_BUILD = 4, TK_KNN_PRUNED = 5, TK_LINEAR = 6, TK_SA_H = 7, TK_UPCONV_H = 8,
  TK_UPCONV_LANE = 9, TK_CV_A1_H = 10, TK_CV_A2 = 11, TK_CV_A2_DENSE6 = 12, TK_CV_A2_LANE6 = 13, TK_CV_B_H = 14,
  TK_POINTWISE = 15, TK_POSE_HEAD = 16, TK_WARP = 17, TK_OTHER = 18
};
void trace_set_fused_layers(const TraceBuf &b);
void trace_set_fused_hoisted(const TraceBuf &b);
void trace_set_knn(const TraceBuf &b);
void trace_set_sampling(const TraceBuf &b);
void trace_set_warp(const TraceBuf &b);

// ---- device helpers ------------------------------------------------------------------------
#if defined(__HIPCC__)

// One copy per translation unit (no relocatable device code in this build); the TU's trace_set_*() fills it.
static __device__ __attribute__((unused)) TraceBuf g_trace = {nullptr, nullptr, 0u};
#define PWCLO_TRACE_TU(name)                                                                   \
  namespace pwclo {                                                                            \
  void trace_set_##name(const TraceBuf &b) {                                                   \
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &b, sizeof(TraceBuf), 0, hipMemcpyHostToDevice); \
  }                                                                                            \
  }

#if !defined(PWCLO_TRACE)
// The product library is built WITHOUT the hooks: even a dormant hook changes register allocation (the level-1
// sampler went from 79 to 120 VGPRs and the step from 2.26 to 2.29 ms).  `python -m pwclonet_pylidarslam_amd.build
// --trace` builds lib/libpwclo_hip_trace.so with -DPWCLO_TRACE for tools/wgtrace.py.
struct TraceScope {
  __device__ __forceinline__ explicit TraceScope(unsigned, unsigned = 0u) {}
};
#else
struct TraceScope {
  unsigned long long t0;
  unsigned kid;
  bool on;
  // sample_mask: record only workgroups with (blockIdx.x & sample_mask) == 0 (kernels with ~1e5 tiny workgroups
  // per step would otherwise perturb the run they are tracing); the tool scales by the sampling rate.
  __device__ __forceinline__ explicit TraceScope(unsigned k, unsigned sample_mask = 0u) : t0(0), kid(k) {
    on = threadIdx.x == 0 && threadIdx.y == 0 && g_trace.rec != nullptr && (blockIdx.x & sample_mask) == 0u;
    if (on) t0 = wall_clock64();
  }
  __device__ __forceinline__ ~TraceScope() {
    if (on) {
      const unsigned long long t1 = wall_clock64();
      const unsigned slot = atomicAdd(g_trace.count, 1u);
      if (slot < g_trace.cap) {
        TraceRec r;
        r.t0 = t0; r.t1 = t1; r.kernel = kid;
        r.block = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        r.nblocks = gridDim.x * gridDim.y * gridDim.z;
        r.hw = (unsigned)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 8 << 6 | 4) |            /* HW_ID: cu_id[11:8] */
               ((unsigned)__builtin_amdgcn_s_getreg((3 - 1) << 11 | 13 << 6 | 4) << 4) |     /* HW_ID: se_id[15:13] */
               ((unsigned)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 8);      /* XCC_ID[3:0] */
        g_trace.rec[slot] = r;
      }
    }
  }
};
#endif

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// ---- DPP helpers (gfx9 encodings: quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror 0x140).
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// Reduce over each row of 16 lanes (result in every lane of the row): xor-1, xor-2 inside quads,
// then mirror within 8 and within 16.  4 VALU-rate steps, no LDS crossbar round trips.
template <typename Op>
__device__ __forceinline__ unsigned row16_allreduce_u32(unsigned v, Op op) {
  v = op(v, dpp_u32<0xB1>(v));   // quad_perm [1,0,3,2]
  v = op(v, dpp_u32<0x4E>(v));   // quad_perm [2,3,0,1]
  v = op(v, dpp_u32<0x141>(v));  // row_half_mirror
  v = op(v, dpp_u32<0x140>(v));  // row_mirror
  return v;
}

// Full-wave reduction; the result is wave-uniform (lives in SGPRs after the readlanes).
template <typename Op>
__device__ __forceinline__ unsigned wave_reduce_u32(unsigned v, Op op) {
  v = row16_allreduce_u32(v, op);
  const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0);
  const unsigned r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32);
  const unsigned r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return op(op(r0, r1), op(r2, r3));
}

struct OpMaxU32 { __device__ __forceinline__ unsigned operator()(unsigned a, unsigned b) const { return a > b ? a : b; } };
struct OpMinU32 { __device__ __forceinline__ unsigned operator()(unsigned a, unsigned b) const { return a < b ? a : b; } };

// Shuffle-based all-reduce (ds_bpermute); only for cold paths.
template <typename Op>
__device__ __forceinline__ float wave_allreduce_f32(float v, Op op) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = op(v, __shfl_xor(v, off, 64));
  return v;
}

template <typename Op>
__device__ __forceinline__ unsigned wave_allreduce_u32(unsigned v, Op op) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = op(v, (unsigned)__shfl_xor((int)v, off, 64));
  return v;
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int off) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
  lo = (unsigned)__shfl_xor((int)lo, off, 64);
  hi = (unsigned)__shfl_xor((int)hi, off, 64);
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long wave_allreduce_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    unsigned long long o = shfl_xor_u64(v, off);
    v = o < v ? o : v;
  }
  return v;
}

// ReLU / max that PROPAGATE NaN like torch.relu / torch.max (fmaxf returns the non-NaN operand, so a diverged training
// run would continue with finite activations and NaN statistics: ADVICE r2).  Used by the module path's training and
// eval kernels (conv1x1.hip, batchnorm.hip); same instruction count as fmaxf (compare + select vs canonicalise + max).
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float max_nan(float a, float b) { return (a > b || a != a) ? a : b; }

// Number of set bits of `mask` below this lane (wave64 prefix count).
__device__ __forceinline__ int mbcnt64(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

#endif  // __HIPCC__

}  // namespace pwclo
