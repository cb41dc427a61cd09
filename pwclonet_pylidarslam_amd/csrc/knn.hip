// Native k-nearest-neighbour search for gfx950 -- the kernel behind pt_utils.knn_point.
//
// The reference has no kernel for this (P2/pytorch_utils.py:12-49): it materialises two
// (B,S,N,3) tensors, a (B,S,N) distance matrix and runs torch.topk -- >= 3.2 GB of intermediates
// per 2x8192 pair for 1.86 MB of algorithmic traffic.  Here nothing is
// materialised: a wave owns KNN_QPW queries (coordinates wave-uniform), its 64 lanes sweep the
// candidates 64 at a time (coalesced reads, each candidate loaded once for all of the wave's
// queries), and selection is a filtered append + bitonic sort:
//   * per query a wave-uniform bound on t = (dx*dx+dy*dy)+dz*dz; candidates with t >= bound
//     cannot enter the current top-K and cost 9 VALU ops + one ballot;
//   * survivors get the exact key sqrtf(t + 1e-8f) (IEEE, same operation order as the oracle)
//     and are appended, packed as (key bits << 32 | index), to the query's LDS pool;
//   * when a pool holds more than 64 entries the wave sorts it (128-wide bitonic network in
//     registers, two entries per lane), keeps the K smallest and tightens the bound.
// Packed 64-bit keys order by (key, index), which is the documented tie rule (lower index first).
// bound = key_K^2 * (1 + 2^-20): strictly above every t whose rounded key can still be <= key_K
// (sqrt and the +1e-8 add each move t by < 2^-23 relative), so the filter never drops a
// candidate the exact comparison would keep; false positives only cost a pool slot.
#include "common.hpp"

namespace pwclo {

constexpr int KNN_WAVES = 4;  // waves per workgroup
constexpr int KNN_QPW = 4;    // queries handled together by one wave
constexpr int KNN_POOL = 128; // pool entries per query (<= 64 carried + <= 64 appended per step)
typedef unsigned long long u64;
constexpr u64 KNN_EMPTY = ~0ull;

__device__ __forceinline__ void cmpx(u64 &a, u64 other, bool keep_min) {
  const bool other_less = other < a;
  a = (other_less == keep_min) ? other : a;
}

// Ascending bitonic sort of 128 keys: element `lane` in e0, element `lane + 64` in e1.
__device__ __forceinline__ void bitonic_sort_128(u64 &e0, u64 &e1, int lane) {
#pragma unroll
  for (int size = 2; size <= 128; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride >= 1; stride >>= 1) {
      if (stride == 64) {  // size == 128: partners live in the same lane, ascending
        const u64 lo = e0 < e1 ? e0 : e1;
        const u64 hi = e0 < e1 ? e1 : e0;
        e0 = lo;
        e1 = hi;
      } else {
        const bool lower = (lane & stride) == 0;
        const bool up0 = size == 128 ? true : ((lane & size) == 0);
        const bool up1 = size == 128 ? true : (size == 64 ? false : ((lane & size) == 0));
        const u64 p0 = shfl_xor_u64(e0, stride);
        const u64 p1 = shfl_xor_u64(e1, stride);
        cmpx(e0, p0, lower == up0);
        cmpx(e1, p1, lower == up1);
      }
    }
  }
}

__global__ __launch_bounds__(KNN_WAVES * 64) void knn_kernel(int n, int s, int K,
                                                             const float *__restrict__ xyz,
                                                             const float *__restrict__ new_xyz,
                                                             int *__restrict__ idx,
                                                             float *__restrict__ dist) {
  __shared__ u64 pools[KNN_WAVES][KNN_QPW][KNN_POOL];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q0 = (blockIdx.x * KNN_WAVES + wave) * KNN_QPW;
  if (q0 >= s) return;  // wave-uniform; the kernel uses no workgroup barrier

  const float *cand = xyz + (size_t)b * n * 3;
  const float *qry = new_xyz + (size_t)b * s * 3;
  float qx[KNN_QPW], qy[KNN_QPW], qz[KNN_QPW], bound[KNN_QPW];
  int cnt[KNN_QPW];
#pragma unroll
  for (int i = 0; i < KNN_QPW; ++i) {
    const int q = min(q0 + i, s - 1);
    qx[i] = qry[q * 3 + 0];
    qy[i] = qry[q * 3 + 1];
    qz[i] = qry[q * 3 + 2];
    bound[i] = __int_as_float(0x7f800000);  // +inf until K candidates have been seen
    cnt[i] = 0;
  }

  // Sort one query's pool, keep the K smallest, tighten its bound.  Wave-uniform control flow.
  auto flush = [&](u64 *pool, int &count, float &bnd) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int c = count;
    u64 e0 = lane < c ? pool[lane] : KNN_EMPTY;
    u64 e1 = lane + 64 < c ? pool[lane + 64] : KNN_EMPTY;
    bitonic_sort_128(e0, e1, lane);
    __builtin_amdgcn_wave_barrier();
    if (lane < K) pool[lane] = e0;
    const int kept = c < K ? c : K;
    count = kept;
    if (kept == K) {
      const unsigned kbits = (unsigned)__shfl((int)(unsigned)(e0 >> 32), K - 1, 64);
      const float key_k = __uint_as_float(kbits);
      bnd = (key_k * key_k) * 1.000001f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  for (int k0 = 0; k0 < n; k0 += 64) {
    const int k = k0 + lane;
    const bool valid = k < n;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (valid) {
      cx = cand[k * 3 + 0];
      cy = cand[k * 3 + 1];
      cz = cand[k * 3 + 2];
    }
#pragma unroll
    for (int i = 0; i < KNN_QPW; ++i) {
      const float dx = qx[i] - cx, dy = qy[i] - cy, dz = qz[i] - cz;
      const float t = (dx * dx + dy * dy) + dz * dz;
      const bool pass = valid && (t < bound[i]);
      const u64 mask = __ballot(pass);
      if (mask != 0ull) {
        if (pass) {
          const float key = sqrtf(t + 1e-8f);
          pools[wave][i][cnt[i] + mbcnt64(mask)] = ((u64)__float_as_uint(key) << 32) | (u64)(unsigned)k;
        }
        cnt[i] = __builtin_amdgcn_readfirstlane(cnt[i] + (int)__popcll(mask));
        if (cnt[i] > 64) flush(pools[wave][i], cnt[i], bound[i]);
      }
    }
  }

#pragma unroll
  for (int i = 0; i < KNN_QPW; ++i) {
    flush(pools[wave][i], cnt[i], bound[i]);
    const int q = q0 + i;
    if (q < s && lane < K) {
      const u64 e = pools[wave][i][lane];
      idx[((size_t)b * s + q) * K + lane] = (int)(unsigned)(e & 0xFFFFFFFFull);
      if (dist) dist[((size_t)b * s + q) * K + lane] = __uint_as_float((unsigned)(e >> 32));
    }
  }
}

}  // namespace pwclo

using namespace pwclo;

extern "C" void knn_point_kernel_wrapper(int b, int n, int s, int nsample, const float *xyz,
                                         const float *new_xyz, int *idx, float *dist) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(nsample >= 1 && nsample <= 64, "knn_point: nsample=%d outside [1,64]", nsample);
  PWCLO_REQUIRE(nsample <= n, "knn_point: nsample=%d exceeds the number of points n=%d", nsample, n);
  PWCLO_REQUIRE(b <= 65535, "knn_point: b=%d exceeds the grid limit", b);
  hipLaunchKernelGGL(knn_kernel, dim3(ceil_div(s, KNN_WAVES * KNN_QPW), b), dim3(KNN_WAVES * 64), 0,
                     current_stream(), n, s, nsample, xyz, new_xyz, idx, dist);
  check_launch("knn_point");
}
