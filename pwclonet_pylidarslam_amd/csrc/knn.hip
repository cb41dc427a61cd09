// Native k-nearest-neighbour search for gfx950 -- the kernel behind pt_utils.knn_point.
//
// The reference has no kernel for this (P2/pytorch_utils.py:12-49): it materialises two
// (B,S,N,3) tensors, a (B,S,N) distance matrix and runs torch.topk -- >= 3.2 GB of intermediates
// per 2x8192 pair for 1.86 MB of algorithmic traffic.  Here nothing is
// materialised: a wave owns QPW queries (coordinates wave-uniform), its 64 lanes sweep the
// candidates 64 at a time (coalesced reads, each candidate loaded once for all of the wave's
// queries), and selection is a filtered append + merge:
//   * per query a wave-uniform bound on t = (dx*dx+dy*dy)+dz*dz; candidates with t >= bound
//     cannot enter the current top-K and cost 9 VALU ops + one ballot;
//   * survivors are appended, packed as (t bits << 32 | index), to the query's LDS pool; the
//     exact key sqrtf(t + 1e-8f) (IEEE, same operation order as the oracle) is taken when the
//     pool is folded in, one root per lane;
//   * the query's best list lives in REGISTERS, sorted, one packed key per lane.  When 64
//     survivors have accumulated the wave sorts them descending with a 21-step bitonic network,
//     takes the lane-wise minimum with the ascending best list (the 64 smallest of the union, as
//     a bitonic sequence) and re-sorts with a 6-step merge; then it tightens the bound.
//     Every compare-exchange step runs at VALU rate: partners come from DPP quad_perm / row_ror
//     (lane xor 1,2,4,8) and from gfx950's v_permlane16_swap / v_permlane32_swap (xor 16, 32),
//     not from the LDS crossbar (the first version's 108 serial ds_bpermute round trips per
//     flush were 85 % of its time).
// Packed 64-bit keys order by (key, index), which is the documented tie rule (lower index first).
// bound = key_K^2 * (1 + 2^-20): strictly above every t whose rounded key can still be <= key_K
// (sqrt and the +1e-8 add each move t by < 2^-23 relative), so the filter never drops a
// candidate the exact comparison would keep; false positives only cost a pool slot.
#include "common.hpp"

namespace pwclo {

constexpr int KNN_WAVES = 4;   // waves per workgroup
constexpr int KNN_POOL = 128;  // pool entries per query: < 64 carried + <= 64 appended per step
typedef unsigned long long u64;
constexpr u64 KNN_EMPTY = ~0ull;

// ---- lane xor S for 32-bit values (S in {1,2,4,8}) through DPP ------------------------------------
template <int S>
__device__ __forceinline__ unsigned xor_dpp(unsigned v) {
  if (S == 1) return dpp_u32<0xB1>(v);   // quad_perm [1,0,3,2]
  if (S == 2) return dpp_u32<0x4E>(v);   // quad_perm [2,3,0,1]
  if (S == 8) return dpp_u32<0x128>(v);  // row_ror:8
  // S == 4: row_ror:n reads lane i-n, so banks {0,2} (bit 2 clear) take row_ror:12 (= i+4) and
  // banks {1,3} take row_ror:4 (= i-4)
  unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x12C, 0xF, 0x5, false);
  return (unsigned)__builtin_amdgcn_update_dpp((int)t, (int)v, 0x124, 0xF, 0xA, false);
}

// Lane mask (bit l = lane l) of the lanes that keep the minimum in the compare-exchange with
// partner l ^ S inside bitonic blocks of SIZE; DESC flips the sort direction.
constexpr u64 keepmin_mask(int size, int stride, bool desc) {
  u64 m = 0;
  for (int l = 0; l < 64; ++l) {
    const bool lower = (l & stride) == 0;
    bool up = size >= 64 ? true : ((l & size) == 0);
    if (desc) up = !up;
    if (lower == up) m |= 1ull << l;
  }
  return m;
}

// One compare-exchange step on a packed key held as (lo, hi).
template <int SIZE, int S, bool DESC>
__device__ __forceinline__ void cmpx_step(unsigned &lo, unsigned &hi) {
  constexpr u64 KEEPMIN = keepmin_mask(SIZE, S, DESC);
  if (S <= 8) {
    const unsigned olo = xor_dpp<S>(lo), ohi = xor_dpp<S>(hi);
    const u64 mine = ((u64)hi << 32) | lo, other = ((u64)ohi << 32) | olo;
    const u64 lt = __builtin_amdgcn_ballot_w64(other < mine);
    const bool take = __builtin_amdgcn_inverse_ballot_w64(~(lt ^ KEEPMIN));  // (other<mine) == keepmin
    lo = take ? olo : lo;
    hi = take ? ohi : hi;
  } else {
    // permlaneN_swap(v, v) = ({lower-lane values twice}, {upper-lane values twice}): every lane
    // of a pair sees (L, U) = (value of the lane with bit S clear, value of the lane with bit S set)
    unsigned Llo, Ulo, Lhi, Uhi;
    if (S == 16) {
      auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
      auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
      Llo = a[0]; Ulo = a[1]; Lhi = b[0]; Uhi = b[1];
    } else {
      auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
      auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
      Llo = a[0]; Ulo = a[1]; Lhi = b[0]; Uhi = b[1];
    }
    const u64 L = ((u64)Lhi << 32) | Llo, U = ((u64)Uhi << 32) | Ulo;
    // a lane wants min(L,U) if it keeps the minimum, else max(L,U):  take U iff (U<L) == keepmin
    const u64 lt = __builtin_amdgcn_ballot_w64(U < L);
    const bool takeU = __builtin_amdgcn_inverse_ballot_w64(~(lt ^ KEEPMIN));
    lo = takeU ? Ulo : Llo;
    hi = takeU ? Uhi : Lhi;
  }
}

template <int SIZE, bool DESC>
__device__ __forceinline__ void merge_stage(unsigned &lo, unsigned &hi) {  // strides SIZE/2 .. 1
  if (SIZE >= 64) cmpx_step<SIZE, 32, DESC>(lo, hi);
  if (SIZE >= 32) cmpx_step<SIZE, 16, DESC>(lo, hi);
  if (SIZE >= 16) cmpx_step<SIZE, 8, DESC>(lo, hi);
  if (SIZE >= 8) cmpx_step<SIZE, 4, DESC>(lo, hi);
  if (SIZE >= 4) cmpx_step<SIZE, 2, DESC>(lo, hi);
  cmpx_step<SIZE, 1, DESC>(lo, hi);
}

// Full bitonic sort of 64 packed keys, one per lane (21 steps).
template <bool DESC>
__device__ __forceinline__ void sort64(unsigned &lo, unsigned &hi) {
  merge_stage<2, DESC>(lo, hi);
  merge_stage<4, DESC>(lo, hi);
  merge_stage<8, DESC>(lo, hi);
  merge_stage<16, DESC>(lo, hi);
  merge_stage<32, DESC>(lo, hi);
  merge_stage<64, DESC>(lo, hi);
}

template <int QPW>
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_kernel(int n, int s, int K,
                                                             const float *__restrict__ xyz,
                                                             const float *__restrict__ new_xyz,
                                                             int *__restrict__ idx,
                                                             float *__restrict__ dist) {
  __shared__ u64 pools[KNN_WAVES][QPW][KNN_POOL];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int q0 = (blockIdx.x * KNN_WAVES + wave) * QPW;
  if (q0 >= s) return;  // wave-uniform; the kernel uses no workgroup barrier

  const float *cand = xyz + (size_t)b * n * 3;
  const float *qry = new_xyz + (size_t)b * s * 3;
  const float INF = __int_as_float(0x7f800000);
  float qx[QPW], qy[QPW], qz[QPW], bound[QPW];
  unsigned blo[QPW], bhi[QPW];  // sorted best list: lane i = i-th smallest packed key
  int cnt[QPW];
#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    const int q = min(q0 + i, s - 1);
    qx[i] = qry[q * 3 + 0];
    qy[i] = qry[q * 3 + 1];
    qz[i] = qry[q * 3 + 2];
    bound[i] = INF;  // until K candidates have been seen
    blo[i] = 0xFFFFFFFFu;
    bhi[i] = 0xFFFFFFFFu;
    cnt[i] = 0;
  }

  // Fold min(count, 64) pooled survivors into the best list; keep the remainder in the pool.
  auto flush = [&](u64 *pool, int &count, float &bnd, unsigned &bl, unsigned &bh) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int c = count;
    const u64 e = lane < c ? pool[lane] : KNN_EMPTY;
    const u64 rest = (lane + 64 < c) ? pool[lane + 64] : KNN_EMPTY;
    unsigned lo = (unsigned)e, hi = (unsigned)(e >> 32);
    if (lane < c) hi = __float_as_uint(sqrtf(__uint_as_float(hi) + 1e-8f));  // exact key (oracle order)
    sort64<true>(lo, hi);                                  // survivors, descending
    const u64 nv = ((u64)hi << 32) | lo, bv = ((u64)bh << 32) | bl;
    const bool tk = nv < bv;                               // lane-wise min: 64 smallest of the union,
    bl = tk ? lo : bl;                                     // a bitonic sequence
    bh = tk ? hi : bh;
    merge_stage<64, false>(bl, bh);                        // ... sorted ascending again
    __builtin_amdgcn_wave_barrier();
    const int left = c > 64 ? c - 64 : 0;
    if (lane < left) pool[lane] = rest;
    count = left;
    const unsigned kbits = (unsigned)__builtin_amdgcn_readlane((int)bh, K - 1);  // K-th best key
    if (kbits != 0xFFFFFFFFu) {                            // K candidates seen: tighten the filter
      const float key_k = __uint_as_float(kbits);
      bnd = (key_k * key_k) * 1.000001f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  for (int k0 = 0; k0 < n; k0 += 64) {
    const int k = k0 + lane;
    float cx = INF, cy = 0.f, cz = 0.f;  // out-of-range lanes: t = inf never passes `t < bound`
    if (k < n) {
      cx = cand[k * 3 + 0];
      cy = cand[k * 3 + 1];
      cz = cand[k * 3 + 2];
    }
#pragma unroll
    for (int i = 0; i < QPW; ++i) {
      const float dx = qx[i] - cx, dy = qy[i] - cy, dz = qz[i] - cz;
      const float t = (dx * dx + dy * dy) + dz * dz;
      const bool pass = t < bound[i];
      const u64 mask = __ballot(pass);
      if (mask != 0ull) {
        if (pass)  // pool keeps t; the exact key is taken once per flush, 64 lanes at a time
          pools[wave][i][cnt[i] + mbcnt64(mask)] = ((u64)__float_as_uint(t) << 32) | (u64)(unsigned)k;
        cnt[i] = __builtin_amdgcn_readfirstlane(cnt[i] + (int)__popcll(mask));
        if (cnt[i] >= 64) flush(pools[wave][i], cnt[i], bound[i], blo[i], bhi[i]);
      }
    }
  }

#pragma unroll
  for (int i = 0; i < QPW; ++i) {
    if (cnt[i] > 0) flush(pools[wave][i], cnt[i], bound[i], blo[i], bhi[i]);
    const int q = q0 + i;
    if (q < s && lane < K) {
      idx[((size_t)b * s + q) * K + lane] = (int)blo[i];
      if (dist) dist[((size_t)b * s + q) * K + lane] = __uint_as_float(bhi[i]);
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
  // 8 queries per wave amortise the candidate loads when there are enough queries to fill the
  // chip; small problems keep 4 (or 2) to expose more waves.
  const long long queries = (long long)b * s;
  if (queries >= 65536) {
    hipLaunchKernelGGL(knn_kernel<8>, dim3(ceil_div(s, KNN_WAVES * 8), b), dim3(KNN_WAVES * 64), 0,
                       current_stream(), n, s, nsample, xyz, new_xyz, idx, dist);
  } else if (queries >= 8192) {
    hipLaunchKernelGGL(knn_kernel<4>, dim3(ceil_div(s, KNN_WAVES * 4), b), dim3(KNN_WAVES * 64), 0,
                       current_stream(), n, s, nsample, xyz, new_xyz, idx, dist);
  } else {
    hipLaunchKernelGGL(knn_kernel<2>, dim3(ceil_div(s, KNN_WAVES * 2), b), dim3(KNN_WAVES * 64), 0,
                       current_stream(), n, s, nsample, xyz, new_xyz, idx, dist);
  }
  check_launch("knn_point");
}
