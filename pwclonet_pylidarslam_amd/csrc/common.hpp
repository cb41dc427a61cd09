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

// ---- device helpers ------------------------------------------------------------------------
#if defined(__HIPCC__)

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

// Number of set bits of `mask` below this lane (wave64 prefix count).
__device__ __forceinline__ int mbcnt64(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

#endif  // __HIPCC__

}  // namespace pwclo
