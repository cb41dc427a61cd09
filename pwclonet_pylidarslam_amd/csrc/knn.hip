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
#include <stdlib.h>

#include "common.hpp"

PWCLO_TRACE_TU(knn)

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

// One compare-exchange step on a packed key held as (lo, hi); KEEPMIN = lanes that keep the minimum.
template <u64 KEEPMIN, int S>
__device__ __forceinline__ void cmpx_mask(unsigned &lo, unsigned &hi) {
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

template <int SIZE, int S, bool DESC>
__device__ __forceinline__ void cmpx_step(unsigned &lo, unsigned &hi) {
  cmpx_mask<keepmin_mask(SIZE, S, DESC), S>(lo, hi);
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
  TraceScope trace_scope_(TK_KNN, 3u);
  __shared__ u64 pools[KNN_WAVES][QPW][KNN_POOL];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: query index and addresses on the SALU
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

// =====================================================================================================
// Spatially pruned exact search (n >= 256, PWCLO_KNN_MIN_N).
//
// knn_build_kernel (one workgroup per cloud): counting sort into (x-slab, z-bin) order, rows
// (x, y, z, bits(original index)) in that order + one axis-aligned box per block of 64 consecutive rows.
// knn_pruned_kernel (one wave per query): lower bound of t to every block box (64 boxes per
// instruction), nearest block first to get a finite bound, then only blocks whose lower bound is
// below the current bound.  Candidates carry their ORIGINAL index in the packed key, and a block is
// skipped only when lb * (1 - 2^-18) >= bound (lb and t are evaluated with the same expression on
// points inside the box, so that margin absorbs every rounding difference), hence the K smallest
// (key, index) pairs -- and therefore the output -- are exactly those of the exhaustive scan.
// =====================================================================================================
namespace pwclo {

constexpr int KB_THREADS = 1024;
constexpr int KNN_MAX_SORT = 16384;  // u64 keys in LDS: 128 KiB

__device__ __forceinline__ unsigned sortable(float f) {   // float -> unsigned with the same order
  const unsigned u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float unsortable(unsigned u) {
  return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu));
}

// Block-wide exclusive scan over KB_THREADS values (one per thread); returns the exclusive prefix.
__device__ __forceinline__ int block_exclusive_scan(int v, int *wsum, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  __syncthreads();
  return base + incl - v;
}

// Build (one workgroup per cloud, points in registers): a counting sort into (x-slab, z-bin) order.
//   1. 1024-bin histogram of x -> slabs of ~n/nslab points (whole bins; exact balance is not needed);
//   2. per slab, 1024/nslab z-bins; every slab is padded to a multiple of 64 rows so that no block of
//      64 rows straddles two slabs;
//   3. scatter rows (x, y, z, bits(original index)) with LDS cursors (order inside a bin is arbitrary:
//      the query's result does not depend on it), inf rows in the padding;
//   4. one bounding box per block.
// 64 consecutive rows are then narrow in x (slab) and z (a few bins) and, lidar sweeps being 2.5-D,
// in y: a query visits ~4 of 128 blocks at n = 8192 (Morton order: ~80).
constexpr int KB_BINS = 1024;

template <int R>
__global__ __launch_bounds__(KB_THREADS) void knn_build_kernel(int n, int nslab, int nblk_max,
                                                               const float *__restrict__ xyz,
                                                               float4 *__restrict__ rows,
                                                               float4 *__restrict__ boxes,
                                                               int *__restrict__ slab_tab) {
  TraceScope trace_scope_(TK_KNN_BUILD);
  __shared__ int hist[KB_BINS];      // x histogram, then slab of every x-bin
  __shared__ int hist2[KB_BINS];     // (slab, z-bin) histogram, then row offset of every bin
  __shared__ int cursor[KB_BINS];
  __shared__ int wsum[KB_THREADS / 64];
  __shared__ float red[4][KB_THREADS / 64];
  __shared__ int slab_first[17], slab_count[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *pts = xyz + (size_t)blockIdx.x * n * 3;
  const float INF = __int_as_float(0x7f800000);
  const int zb = KB_BINS / nslab;                            // z-bins per slab

  float px[R], py[R], pz[R];
  float lox = INF, hix = -INF, loz = INF, hiz = -INF;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int k = r * KB_THREADS + tid;
    px[r] = INF; py[r] = 0.f; pz[r] = 0.f;
    if (k < n) {
      px[r] = pts[k * 3 + 0]; py[r] = pts[k * 3 + 1]; pz[r] = pts[k * 3 + 2];
      lox = fminf(lox, px[r]); hix = fmaxf(hix, px[r]);
      loz = fminf(loz, pz[r]); hiz = fmaxf(hiz, pz[r]);
    }
  }
  lox = wave_allreduce_f32(lox, [](float a, float b) { return fminf(a, b); });
  hix = wave_allreduce_f32(hix, [](float a, float b) { return fmaxf(a, b); });
  loz = wave_allreduce_f32(loz, [](float a, float b) { return fminf(a, b); });
  hiz = wave_allreduce_f32(hiz, [](float a, float b) { return fmaxf(a, b); });
  if (lane == 0) { red[0][wave] = lox; red[1][wave] = hix; red[2][wave] = loz; red[3][wave] = hiz; }
  hist[tid] = 0; hist2[tid] = 0; cursor[tid] = 0;
  __syncthreads();
  for (int w = 0; w < KB_THREADS / 64; ++w) {
    lox = fminf(lox, red[0][w]); hix = fmaxf(hix, red[1][w]);
    loz = fminf(loz, red[2][w]); hiz = fmaxf(hiz, red[3][w]);
  }
  const float sx = hix > lox ? (float)KB_BINS / (hix - lox) : 0.f;
  const float sz = hiz > loz ? (float)zb / (hiz - loz) : 0.f;

  int bx[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    bx[r] = -1;
    if (r * KB_THREADS + tid < n) {
      bx[r] = min(max((int)((px[r] - lox) * sx), 0), KB_BINS - 1);
      atomicAdd(&hist[bx[r]], 1);
    }
  }
  __syncthreads();
  {  // slab of every x-bin from the cumulative count at the start of the bin
    const int before = block_exclusive_scan(hist[tid], wsum, tid);
    const int target = (n + nslab - 1) / nslab;
    hist[tid] = min(before / target, nslab - 1);
  }
  __syncthreads();
  int b2[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    b2[r] = -1;
    if (bx[r] >= 0) {
      const int bz = min(max((int)((pz[r] - loz) * sz), 0), zb - 1);
      b2[r] = hist[bx[r]] * zb + bz;
      atomicAdd(&hist2[b2[r]], 1);
    }
  }
  __syncthreads();
  const int ex2 = block_exclusive_scan(hist2[tid], wsum, tid);   // rows before bin `tid`, unpadded
  if ((tid % zb) == 0) slab_first[tid / zb] = ex2;               // unpadded start of each slab
  if (tid == 0) slab_first[nslab] = n;
  __syncthreads();
  if (tid == 0) {                                               // padded slab starts (multiples of 64)
    int start = 0;
    for (int sl = 0; sl < nslab; ++sl) {
      const int cnt = slab_first[sl + 1] - slab_first[sl];
      slab_count[sl] = cnt;
      wsum[sl] = start;                                          // reuse wsum[] as padded starts
      start += (cnt + 63) / 64 * 64;
    }
    red[0][0] = __int_as_float(start);                           // rows in use (end of the last slab)
    if (slab_tab != nullptr) {                                   // (padded first row, rows) of every slab, for the
      int *tab = slab_tab + (size_t)blockIdx.x * 32;             // slab-pruned sampler (sampling.hip: fps_slab_kernel)
      for (int sl = 0; sl < 16; ++sl) {
        tab[2 * sl] = sl < nslab ? wsum[sl] : 0;
        tab[2 * sl + 1] = sl < nslab ? slab_count[sl] : 0;
      }
    }
  }
  __syncthreads();
  const int my_slab = tid / zb;
  hist2[tid] = wsum[my_slab] + (ex2 - slab_first[my_slab]);      // row offset of bin `tid`
  const int rows_used = __float_as_int(red[0][0]);
  __syncthreads();
  float4 *orow = rows + (size_t)blockIdx.x * nblk_max * 64;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (b2[r] >= 0) {
      const int pos = hist2[b2[r]] + atomicAdd(&cursor[b2[r]], 1);
      orow[pos] = make_float4(px[r], py[r], pz[r], __uint_as_float((unsigned)(r * KB_THREADS + tid)));
    }
  }
  // padding rows: the tail of every slab and everything after the last slab
  if (tid < 64) {
    for (int sl = 0; sl < nslab; ++sl) {
      const int beg = wsum[sl] + slab_count[sl];
      const int end = wsum[sl] + (slab_count[sl] + 63) / 64 * 64;
      if (beg + tid < end) orow[beg + tid] = make_float4(INF, 0.f, 0.f, 0.f);
    }
  }
  for (int p = rows_used + tid; p < nblk_max * 64; p += KB_THREADS) orow[p] = make_float4(INF, 0.f, 0.f, 0.f);
  __syncthreads();   // rows written by this workgroup are read back below (never read before: no stale L1 line)
  float4 *obox = boxes + (size_t)blockIdx.x * nblk_max * 2;
  for (int blk = wave; blk < nblk_max; blk += KB_THREADS / 64) {
    const float4 c = orow[blk * 64 + lane];
    const bool valid = c.x != INF;
    const unsigned big = 0xFFFFFFFFu;
    const unsigned lx = wave_reduce_u32(valid ? sortable(c.x) : big, OpMinU32());
    const unsigned ly = wave_reduce_u32(valid ? sortable(c.y) : big, OpMinU32());
    const unsigned lz = wave_reduce_u32(valid ? sortable(c.z) : big, OpMinU32());
    const unsigned hx = wave_reduce_u32(valid ? sortable(c.x) : 0u, OpMaxU32());
    const unsigned hy = wave_reduce_u32(valid ? sortable(c.y) : 0u, OpMaxU32());
    const unsigned hz = wave_reduce_u32(valid ? sortable(c.z) : 0u, OpMaxU32());
    if (lane == 0) {
      const bool any = lx != big;
      obox[blk * 2 + 0] = any ? make_float4(unsortable(lx), unsortable(ly), unsortable(lz), 0.f)
                              : make_float4(INF, INF, INF, 0.f);
      obox[blk * 2 + 1] = any ? make_float4(unsortable(hx), unsortable(hy), unsortable(hz), 0.f)
                              : make_float4(-INF, -INF, -INF, 0.f);
    }
  }
}

constexpr int KP_MAXR = 5;  // block boxes per lane: up to 320 blocks (16384 points + slab padding)

__global__ __launch_bounds__(KNN_WAVES * 64) void knn_pruned_kernel(int nblk, int s, int K,
                                                                    const float4 *__restrict__ rows,
                                                                    const float4 *__restrict__ boxes,
                                                                    const float *__restrict__ new_xyz,
                                                                    int *__restrict__ idx,
                                                                    float *__restrict__ dist) {
  TraceScope trace_scope_(TK_KNN_PRUNED, 15u);
  __shared__ u64 pools[KNN_WAVES][KNN_POOL];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: query index and addresses on the SALU
  const int q = blockIdx.x * KNN_WAVES + wave;
  if (q >= s) return;  // wave-uniform; no workgroup barrier below
  const float INF = __int_as_float(0x7f800000);
  const float *qp = new_xyz + ((size_t)b * s + q) * 3;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const float4 *crow = rows + (size_t)b * nblk * 64;
  const float4 *cbox = boxes + (size_t)b * nblk * 2;
  u64 *pool = pools[wave];

  float lb[KP_MAXR];
#pragma unroll
  for (int r = 0; r < KP_MAXR; ++r) {
    const int blk = r * 64 + lane;
    float v = INF;
    if (blk < nblk) {
      const float4 l = cbox[blk * 2], h = cbox[blk * 2 + 1];
      const float dx = fmaxf(fmaxf(l.x - qx, qx - h.x), 0.f);
      const float dy = fmaxf(fmaxf(l.y - qy, qy - h.y), 0.f);
      const float dz = fmaxf(fmaxf(l.z - qz, qz - h.z), 0.f);
      v = ((dx * dx + dy * dy) + dz * dz) * 0.999996f;   // 1 - 2^-18: strictly conservative
    }
    lb[r] = v;                                            // empty boxes give inf (or NaN): never visited
  }

  float bound = INF;
  unsigned bl = 0xFFFFFFFFu, bh = 0xFFFFFFFFu;
  int cnt = 0;

  auto fold = [&]() {  // fold min(cnt,64) pooled survivors into the sorted best list (see knn_kernel)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int c = cnt;
    const u64 e = lane < c ? pool[lane] : KNN_EMPTY;
    const u64 rest = (lane + 64 < c) ? pool[lane + 64] : KNN_EMPTY;
    unsigned lo = (unsigned)e, hi = (unsigned)(e >> 32);
    if (lane < c) hi = __float_as_uint(sqrtf(__uint_as_float(hi) + 1e-8f));
    sort64<true>(lo, hi);
    const u64 nv = ((u64)hi << 32) | lo, bv = ((u64)bh << 32) | bl;
    const bool tk = nv < bv;
    bl = tk ? lo : bl;
    bh = tk ? hi : bh;
    merge_stage<64, false>(bl, bh);
    __builtin_amdgcn_wave_barrier();
    const int left = c > 64 ? c - 64 : 0;
    if (lane < left) pool[lane] = rest;
    cnt = left;
    const unsigned kbits = (unsigned)__builtin_amdgcn_readlane((int)bh, K - 1);
    if (kbits != 0xFFFFFFFFu) {
      const float key_k = __uint_as_float(kbits);
      bound = (key_k * key_k) * 1.000001f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  auto process = [&](const float4 c) {                     // padding rows have x = inf -> t = inf
    const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
    const float t = (dx * dx + dy * dy) + dz * dz;
    const bool pass = t < bound;
    const u64 mask = __ballot(pass);
    if (mask != 0ull) {
      if (pass) pool[cnt + mbcnt64(mask)] = ((u64)__float_as_uint(t) << 32) | (u64)__float_as_uint(c.w);
      cnt = __builtin_amdgcn_readfirstlane(cnt + (int)__popcll(mask));
      if (cnt >= 64) fold();
    }
  };

  // nearest block first: a finite bound as early as possible
  float best_lb = lb[0];
  int best_blk = lane;
#pragma unroll
  for (int r = 1; r < KP_MAXR; ++r)
    if (lb[r] < best_lb) { best_lb = lb[r]; best_blk = r * 64 + lane; }
  const unsigned mlb = wave_reduce_u32(__float_as_uint(best_lb), OpMinU32());   // lb >= 0: bits order
  const unsigned cand = __float_as_uint(best_lb) == mlb ? (unsigned)best_blk : 0xFFFFFFFFu;
  const int first = (int)wave_reduce_u32(cand, OpMinU32());
  process(crow[first * 64 + lane]);
  if (cnt > 0 && bound == INF) fold();

  // remaining blocks whose box can still hold a neighbour, 4 row loads in flight at a time (a wave
  // is a chain of dependent steps; batching the loads is what hides the L2 latency)
  constexpr int VB = 4;
#pragma unroll
  for (int r = 0; r < KP_MAXR; ++r) {
    if (r * 64 >= nblk) break;
    u64 todo = __ballot(lb[r] < bound && (r * 64 + lane) != first);
    while (todo != 0ull) {
      int blk[VB];
      float lbv[VB];
      float4 c[VB];
#pragma unroll
      for (int u = 0; u < VB; ++u) {
        blk[u] = -1;
        lbv[u] = INF;
        if (todo != 0ull) {
          const int i = __builtin_ctzll(todo);
          todo &= todo - 1;
          blk[u] = r * 64 + i;
          lbv[u] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(lb[r]), i));
        }
      }
#pragma unroll
      for (int u = 0; u < VB; ++u)
        if (blk[u] >= 0) c[u] = crow[blk[u] * 64 + lane];
#pragma unroll
      for (int u = 0; u < VB; ++u)
        if (blk[u] >= 0 && lbv[u] < bound) process(c[u]);   // the bound may have tightened meanwhile
    }
  }
  if (cnt > 0) fold();
  if (lane < K) {
    idx[((size_t)b * s + q) * K + lane] = (int)bl;
    if (dist) dist[((size_t)b * s + q) * K + lane] = __uint_as_float(bh);
  }
}


// =====================================================================================================
// Several queries per wave ("rows"): the pruned search for K <= 32.
//
// A bitonic network costs the same number of steps whatever the number of independent sequences packed into the
// wave, and knn_pruned_kernel spends most of its instructions in 64-wide sorts (21 + 6 steps per fold) to select
// 4..32 neighbours.  Here a wave serves Q = 64 / L queries, L = 16 (K <= 16) or 32 (K <= 32): query r owns lanes
// [r*L, (r+1)*L) -- its sorted best list (one packed (key, index) per lane), its block lower bounds (block
// p, p+L, ... on lane p), its LDS pool.  All rows run the same instruction stream on their own data:
//   * every row walks ITS blocks in order of increasing lower bound (nearest first), L candidates per sub-step,
//     until the next block's bound cannot beat its K-th key;
//   * survivors (t < row bound) are appended to the row's pool; a pool holding >= L entries (or the first K while
//     the bound is still infinite) is folded in with an L-wide sort (10 / 15 steps) + L-wide merge (4 / 5 steps);
//   * at the end of a block the few survivors a tight bound lets through are INSERTED one at a time (position =
//     number of smaller list entries, lanes above it shift up by one: ~12 instructions for all rows at once)
//     instead of sorted.
// Keys, tie rule (lower index first), the conservative filter bound and the block-skip margin are those of
// knn_pruned_kernel, so the output is the same list, bit for bit (tests/test_gpu_ops.py::test_knn_*).
// =====================================================================================================
template <int L>
constexpr u64 keepmin_rows(int size, int stride, bool desc) {   // like keepmin_mask, every row sorted the same way
  u64 m = 0;
  for (int l = 0; l < 64; ++l) {
    const bool lower = (l & stride) == 0;
    bool up = size >= L ? true : ((l & size) == 0);
    if (desc) up = !up;
    if (lower == up) m |= 1ull << l;
  }
  return m;
}
template <int L, int SIZE, bool DESC>
__device__ __forceinline__ void merge_stage_rows(unsigned &lo, unsigned &hi) {   // strides SIZE/2 .. 1, SIZE <= L
  if (SIZE >= 32) cmpx_mask<keepmin_rows<L>(SIZE, 16, DESC), 16>(lo, hi);
  if (SIZE >= 16) cmpx_mask<keepmin_rows<L>(SIZE, 8, DESC), 8>(lo, hi);
  if (SIZE >= 8) cmpx_mask<keepmin_rows<L>(SIZE, 4, DESC), 4>(lo, hi);
  if (SIZE >= 4) cmpx_mask<keepmin_rows<L>(SIZE, 2, DESC), 2>(lo, hi);
  cmpx_mask<keepmin_rows<L>(SIZE, 1, DESC), 1>(lo, hi);
}
template <int L, bool DESC>
__device__ __forceinline__ void sort_rows(unsigned &lo, unsigned &hi) {          // every row of L lanes, 10 / 15 steps
  merge_stage_rows<L, 2, DESC>(lo, hi);
  merge_stage_rows<L, 4, DESC>(lo, hi);
  merge_stage_rows<L, 8, DESC>(lo, hi);
  merge_stage_rows<L, 16, DESC>(lo, hi);
  if (L >= 32) merge_stage_rows<L, 32, DESC>(lo, hi);
}
template <int L>
__device__ __forceinline__ unsigned row_min_u32(unsigned v) {                    // all lanes of a row get the row's min
  v = row16_allreduce_u32(v, OpMinU32());
  if (L == 32) {
    const unsigned o = (unsigned)__shfl_xor((int)v, 16, 64);
    v = o < v ? o : v;
  }
  return v;
}

template <int L, int R>
__global__ __launch_bounds__(KNN_WAVES * 64) void knn_rows_kernel(int nblk, int s, int K,
                                                                  const float4 *__restrict__ rows,
                                                                  const float4 *__restrict__ boxes,
                                                                  const float *__restrict__ new_xyz,
                                                                  int *__restrict__ idx,
                                                                  float *__restrict__ dist, int settle_min) {
  TraceScope trace_scope_(TK_KNN_PRUNED, 15u);
  constexpr int Q = 64 / L, SUB = 64 / L;
  constexpr unsigned LMASK = L == 32 ? 0xFFFFFFFFu : 0xFFFFu;
  constexpr int INS_MAX = L / 4;            // up to this many pooled survivors per row: insert; more: sort
  __shared__ u64 pools[KNN_WAVES][Q][2 * L];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int q0 = (blockIdx.x * KNN_WAVES + wave) * Q;
  if (q0 >= s) return;                      // wave-uniform; no workgroup barrier below
  const int row = lane / L, p = lane % L;
  const int rowbase = lane - p;             // first lane of this row
  const int q = q0 + row;
  const bool qvalid = q < s;
  const float INF = __int_as_float(0x7f800000);
  const float *qp = new_xyz + ((size_t)b * s + (qvalid ? q : s - 1)) * 3;
  const float qx = qp[0], qy = qp[1], qz = qp[2];
  const float4 *crow = rows + (size_t)b * nblk * 64;
  const float4 *cbox = boxes + (size_t)b * nblk * 2;
  u64 *pool = pools[wave][row];

  float lb[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int blk = r * L + p;
    float v = INF;
    if (blk < nblk) {
      const float4 l = cbox[blk * 2], h = cbox[blk * 2 + 1];
      const float dx = fmaxf(fmaxf(l.x - qx, qx - h.x), 0.f);
      const float dy = fmaxf(fmaxf(l.y - qy, qy - h.y), 0.f);
      const float dz = fmaxf(fmaxf(l.z - qz, qz - h.z), 0.f);
      v = ((dx * dx + dy * dy) + dz * dz) * 0.999996f;   // 1 - 2^-18: strictly conservative (knn_pruned_kernel)
      if (!(v < INF)) v = INF;                           // empty boxes give inf or NaN: never visited
    }
    lb[r] = v;
  }

  float bound = INF;
  unsigned bl = 0xFFFFFFFFu, bh = 0xFFFFFFFFu;
  int cnt = 0;

  auto row_bits = [&](u64 mask) -> unsigned { return (unsigned)(mask >> rowbase) & LMASK; };
  auto refresh_bound = [&]() {
    const unsigned kbits = (unsigned)__builtin_amdgcn_ds_bpermute((rowbase + K - 1) << 2, (int)bh);   // row's K-th key
    if (kbits != 0xFFFFFFFFu) {
      const float key_k = __uint_as_float(kbits);
      bound = (key_k * key_k) * 1.000001f;
    }
  };
  // fold min(cnt, L) pooled survivors of every row into its best list with an L-wide sort
  auto fold_sort = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int c = cnt;
    const u64 e = p < c ? pool[p] : KNN_EMPTY;
    const u64 rest = (p + L < c) ? pool[p + L] : KNN_EMPTY;
    unsigned lo = (unsigned)e, hi = (unsigned)(e >> 32);
    if (p < c) hi = __float_as_uint(sqrtf(__uint_as_float(hi) + 1e-8f));   // exact key (oracle order)
    sort_rows<L, true>(lo, hi);                            // survivors, descending
    const u64 nv = ((u64)hi << 32) | lo, bv = ((u64)bh << 32) | bl;
    const bool tk = nv < bv;                               // lane-wise min with the ascending list: bitonic
    bl = tk ? lo : bl;
    bh = tk ? hi : bh;
    merge_stage_rows<L, L, false>(bl, bh);                 // ... ascending again
    __builtin_amdgcn_wave_barrier();
    const int left = c > L ? c - L : 0;
    if (p < left) pool[p] = rest;
    cnt = left;
    refresh_bound();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // insert every row's pooled survivors one at a time (few of them: the bound is already tight)
  auto insert_all = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = 0; __ballot(i < cnt) != 0ull; ++i) {
      const bool have = i < cnt;
      const u64 e = have ? pool[i] : KNN_EMPTY;            // same address for the whole row: broadcast read
      const unsigned elo = (unsigned)e;
      const unsigned ehi = have ? __float_as_uint(sqrtf(__uint_as_float((unsigned)(e >> 32)) + 1e-8f)) : 0xFFFFFFFFu;
      const u64 E = ((u64)ehi << 32) | elo, mine = ((u64)bh << 32) | bl;
      const int pos = (int)__popc(row_bits(__ballot(mine < E)));   // the list is ascending: a prefix of the row
      const unsigned plo = dpp_u32<0x138>(bl), phi = dpp_u32<0x138>(bh);    // wave_shr:1 -- value of lane - 1
      if (p == pos) { bl = elo; bh = ehi; }
      else if (p > pos) { bl = plo; bh = phi; }
    }
    cnt = 0;
    refresh_bound();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // unvisited block of this row with the smallest lower bound, if it can still hold a neighbour (-1 otherwise)
  auto next_block = [&]() -> int {
    float m = lb[0];
    int mb = p;
#pragma unroll
    for (int r = 1; r < R; ++r)
      if (lb[r] < m) { m = lb[r]; mb = r * L + p; }
    const unsigned mv = row_min_u32<L>(__float_as_uint(m));            // lb >= 0: the bit patterns order
    const unsigned cb = __float_as_uint(m) == mv ? (unsigned)mb : 0xFFFFFFFFu;
    const unsigned blk = row_min_u32<L>(cb);
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (blk == (unsigned)(r * L + p)) lb[r] = INF;                   // visited
    return __uint_as_float(mv) < bound ? (int)blk : -1;
  };

  int cur = next_block();
  if (!qvalid) cur = -1;
  while (true) {
    const bool active = cur >= 0;
    if (__ballot(active) == 0ull) break;
    const float4 *base = crow + (size_t)(active ? cur : 0) * 64 + p;
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      float4 c = make_float4(INF, 0.f, 0.f, 0.f);
      if (active) c = base[sub * L];
      const float dx = qx - c.x, dy = qy - c.y, dz = qz - c.z;
      const float t = (dx * dx + dy * dy) + dz * dz;               // padding rows have x = inf -> t = inf
      const bool pass = active && t < bound;
      const u64 mask = __ballot(pass);
      if (mask != 0ull) {
        const unsigned rb = row_bits(mask);
        if (pass) pool[cnt + (int)__popc(rb & ((1u << p) - 1u))] = ((u64)__float_as_uint(t) << 32) | (u64)__float_as_uint(c.w);
        cnt += (int)__popc(rb);
        if (__ballot(cnt >= L || (bound == INF && cnt >= K)) != 0ull) fold_sort();
      }
    }
    // end of the block: settle the survivors so that the bound is current (a row with fewer than settle_min pooled
    // survivors keeps collecting: its bound stays a little loose, which is still exact), then move on
    if (__ballot(cnt >= settle_min) != 0ull) {
      while (__ballot(cnt > 0) != 0ull) {
        if (__ballot(cnt > INS_MAX) != 0ull) fold_sort();
        else insert_all();
      }
    }
    const int nb = next_block();
    cur = active ? nb : -1;
  }
  while (__ballot(cnt > 0) != 0ull) {         // whatever is still pooled when the last row finishes
    if (__ballot(cnt > INS_MAX) != 0ull) fold_sort();
    else insert_all();
  }
  if (qvalid && p < K) {
    idx[((size_t)b * s + q) * K + p] = (int)bl;
    if (dist) dist[((size_t)b * s + q) * K + p] = __uint_as_float(bh);
  }
}

}  // namespace pwclo

static int knn_tune_early(const char *name, int dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}

static int knn_slabs(int n) {            // ~ sqrt(#blocks), power of two in [2,16]
  int nslab = 2;
  while (nslab < 16 && nslab * nslab * 4 <= (n + 63) / 64) nslab <<= 1;
  return nslab;
}

static int knn_tune(const char *name, int dflt) {
  const char *e = getenv(name);
  return e ? atoi(e) : dflt;
}
static int knn_min_n() { static const int v = knn_tune("PWCLO_KNN_MIN_N", 256); return v; }
static int knn_min_s() { static const int v = knn_tune("PWCLO_KNN_MIN_S", 256); return v; }

extern "C" long long knn_point_workspace_bytes(int b, int n) {
  if (n < knn_min_n() || n > pwclo::KNN_MAX_SORT) return 0;       // exhaustive kernel: no workspace
  const long long nblk = (n + 63) / 64 + knn_slabs(n);
  return (long long)b * nblk * (64 * 16 + 32);
}

// Build pass alone: sorted rows + block boxes into `workspace` (knn_point_workspace_bytes(b, n) bytes) and, when
// slab_tab != NULL, 32 ints per cloud: (padded first row, row count) of up to 16 x-slabs.
static bool knn_build_launch(int b, int n, const float *xyz, void *workspace, int *slab_tab) {
  const int nslab = knn_slabs(n);
  const int nblk = (n + 63) / 64 + nslab;
  float4 *rows = reinterpret_cast<float4 *>(workspace);
  float4 *boxes = rows + (size_t)b * nblk * 64;
  const int regs = ceil_div(n, KB_THREADS);
#define KB_CASE(RR)                                                                                   \
  hipLaunchKernelGGL(knn_build_kernel<RR>, dim3(b), dim3(KB_THREADS), 0, current_stream(), n, nslab, nblk, \
                     xyz, rows, boxes, slab_tab);
  if (regs <= 1) { KB_CASE(1) }
  else if (regs <= 2) { KB_CASE(2) }
  else if (regs <= 4) { KB_CASE(4) }
  else if (regs <= 8) { KB_CASE(8) }
  else { KB_CASE(16) }
#undef KB_CASE
  return check_launch("knn_point(build)");
}

// `first_cloud` / `built_b`: the structure in `workspace` was built for `built_b` clouds and this search covers the `b`
// clouds starting at `first_cloud` of them (the siamese pyramid builds both frames' clouds in one batch; the refinement
// levels search one frame's half -- round 3).  The defaults describe a structure built for exactly these b clouds.
static void knn_search_launch(int b, int n, int s, int nsample, const float *new_xyz, int *idx, float *dist,
                              void *workspace, int first_cloud = 0, int built_b = -1) {
  const int nslab = knn_slabs(n);
  const int nblk = (n + 63) / 64 + nslab;
  if (built_b < 0) built_b = b;
  float4 *rows = reinterpret_cast<float4 *>(workspace) + (size_t)first_cloud * nblk * 64;
  float4 *boxes = reinterpret_cast<float4 *>(workspace) + (size_t)built_b * nblk * 64 + (size_t)first_cloud * nblk * 2;
  // K <= 32: several queries per wave (knn_rows_kernel); PWCLO_KNN_ROWS=0 keeps one query per wave (A/B switch)
  static int use_rows = -1;
  if (use_rows < 0) { const char *e = getenv("PWCLO_KNN_ROWS"); use_rows = e ? atoi(e) : 1; }
  const int L = nsample <= 16 ? 16 : 32;
  const int need = ceil_div(nblk, L);
  static const int settle16 = knn_tune_early("PWCLO_KNN_SETTLE16", 4), settle32 = knn_tune_early("PWCLO_KNN_SETTLE32", 16);
  const int settle_min = L == 16 ? settle16 : settle32;
#define KR_CASE(LL, RR)                                                                                        \
  if (L == LL && need <= RR) {                                                                                 \
    hipLaunchKernelGGL((knn_rows_kernel<LL, RR>), dim3(ceil_div(s, KNN_WAVES * (64 / LL)), b), dim3(KNN_WAVES * 64), \
                       0, current_stream(), nblk, s, nsample, rows, boxes, new_xyz, idx, dist, settle_min);    \
    check_launch("knn_point(rows)");                                                                           \
    return;                                                                                                    \
  }
  if (use_rows && nsample <= 32) {
    KR_CASE(16, 2) KR_CASE(16, 3) KR_CASE(16, 5) KR_CASE(16, 9) KR_CASE(16, 17)
    KR_CASE(32, 1) KR_CASE(32, 2) KR_CASE(32, 3) KR_CASE(32, 5) KR_CASE(32, 9)
  }
#undef KR_CASE
  hipLaunchKernelGGL(knn_pruned_kernel, dim3(ceil_div(s, KNN_WAVES), b), dim3(KNN_WAVES * 64), 0,
                     current_stream(), nblk, s, nsample, rows, boxes, new_xyz, idx, dist);
  check_launch("knn_point(pruned)");
}

extern "C" void knn_point_ws_kernel_wrapper(int b, int n, int s, int nsample, const float *xyz,
                                            const float *new_xyz, int *idx, float *dist,
                                            void *workspace) {
  if (b <= 0 || s <= 0) return;
  if (workspace == nullptr || knn_point_workspace_bytes(b, n) == 0 || s < knn_min_s()) {
    knn_point_kernel_wrapper(b, n, s, nsample, xyz, new_xyz, idx, dist);   // too few queries to amortise the build
    return;
  }
  PWCLO_REQUIRE(nsample >= 1 && nsample <= 64, "knn_point: nsample=%d outside [1,64]", nsample);
  PWCLO_REQUIRE(nsample <= n, "knn_point: nsample=%d exceeds the number of points n=%d", nsample, n);
  PWCLO_REQUIRE(b <= 65535, "knn_point: b=%d exceeds the grid limit", b);
  if (!knn_build_launch(b, n, xyz, workspace, nullptr)) return;
  knn_search_launch(b, n, s, nsample, new_xyz, idx, dist, workspace);
}

// The two passes separately: one build of a cloud's search structure can serve several searches AND the
// slab-pruned furthest point sampling of the same cloud (furthest_point_sampling_slab_kernel_wrapper).
extern "C" void knn_build_kernel_wrapper(int b, int n, const float *xyz, void *workspace, int *slab_tab) {
  if (b <= 0) return;
  PWCLO_REQUIRE(workspace != nullptr && n >= 64 && n <= pwclo::KNN_MAX_SORT,
                "knn_build: n=%d outside [64,%d] or no workspace", n, pwclo::KNN_MAX_SORT);
  PWCLO_REQUIRE(b <= 65535, "knn_build: b=%d exceeds the grid limit", b);
  knn_build_launch(b, n, xyz, workspace, slab_tab);
}

extern "C" void knn_point_prebuilt_kernel_wrapper(int b, int n, int s, int nsample, const float *new_xyz, int *idx,
                                                  float *dist, void *workspace) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(workspace != nullptr && n >= 64 && n <= pwclo::KNN_MAX_SORT,
                "knn_point(prebuilt): n=%d outside [64,%d] or no workspace", n, pwclo::KNN_MAX_SORT);
  PWCLO_REQUIRE(nsample >= 1 && nsample <= 64 && nsample <= n, "knn_point(prebuilt): nsample=%d invalid for n=%d", nsample, n);
  PWCLO_REQUIRE(b <= 65535, "knn_point(prebuilt): b=%d exceeds the grid limit", b);
  knn_search_launch(b, n, s, nsample, new_xyz, idx, dist, workspace);
}

extern "C" void knn_point_prebuilt_slice_kernel_wrapper(int b, int n, int s, int nsample, const float *new_xyz, int *idx,
                                                        float *dist, void *workspace, int first_cloud, int built_b) {
  if (b <= 0 || s <= 0) return;
  PWCLO_REQUIRE(workspace != nullptr && n >= 64 && n <= pwclo::KNN_MAX_SORT,
                "knn_point(prebuilt slice): n=%d outside [64,%d] or no workspace", n, pwclo::KNN_MAX_SORT);
  PWCLO_REQUIRE(nsample >= 1 && nsample <= 64 && nsample <= n, "knn_point(prebuilt slice): nsample=%d invalid for n=%d", nsample, n);
  PWCLO_REQUIRE(first_cloud >= 0 && built_b >= 1 && first_cloud + b <= built_b && built_b <= 65535,
                "knn_point(prebuilt slice): clouds [%d, %d) are not inside the %d the structure was built for", first_cloud,
                first_cloud + b, built_b);
  knn_search_launch(b, n, s, nsample, new_xyz, idx, dist, workspace, first_cloud, built_b);
}

extern "C" int knn_point_slabs(int n) { return knn_slabs(n); }
extern "C" long long knn_point_build_bytes(int b, int n) {        // workspace of the build for ANY 64 <= n <= 16384
  if (n < 64 || n > pwclo::KNN_MAX_SORT) return 0;
  const long long nblk = (n + 63) / 64 + knn_slabs(n);
  return (long long)b * nblk * (64 * 16 + 32);
}
