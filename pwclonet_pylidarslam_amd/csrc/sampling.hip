// Furthest point sampling + gather_points(+grad) for gfx950.
//
// Replaces P2/_ext-src/src/sampling_gpu.cu (reference).  Not a translation: the reference keeps
// the running distances in global memory, re-reads every point each iteration and reduces
// through a 9-level LDS tree with 10 barriers per iteration.  Here one workgroup owns one
// cloud, every point and its running distance live in VGPRs for the whole call, the
// per-iteration arg-max is a wave64 shuffle reduction followed by ONE barrier (double-buffered
// LDS slots), and the winner's coordinates come from an LDS-resident copy of the cloud, so the
// loop touches HBM only to store one index.
//
// FPS tie rule (must match the reference bit for bit): the reference runs
// bs = opt_n_threads(n) threads; thread t = k mod bs keeps the FIRST maximum of its strided
// points (strict >), then a tree that keeps slot t over slot t+s on ties.  Net effect: among
// equal maxima the winner minimises (bitrev_{log2 bs}(k mod bs), k div bs) lexicographically.
// That pair is packed into a 32-bit priority `pri = bitrev << 23 | (k div bs)`; the kernel
// maximises the distance and, among equal distances, minimises pri.  A thread whose points are
// all skipped (|p|^2 <= 1e-3) or out of range contributes nothing; if no thread has a
// candidate the result is index 0, as in the reference (best = -1, besti = 0 everywhere).
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include <atomic>

#include "common.hpp"

PWCLO_TRACE_TU(sampling)

namespace pwclo {

constexpr int PRI_SHIFT = 23;  // k div bs < 2^23
constexpr int FPS_SLOT_BYTES = 64;  // 3 rotating u64 arg-max slots in front of the LDS point table

__device__ __forceinline__ unsigned fps_bitrev(unsigned v, int bits) {
  return bits == 0 ? 0u : (__brev(v) >> (32 - bits));
}

// T threads; a thread owns PPT = I << E points.  The reference partitions the cloud over
// bs = opt_n_threads(n) threads by k mod bs.  Here
//   * T >= bs (E = 0): thread tid owns residue tid mod bs, points k = tid + T*i, i < I;
//   * T <  bs (E = log2(bs/T)): thread tid owns the 2^E residues tid + T*u.  Their bit-reversed
//     ranks are bitrev(tid) + bitrev_E(u), so visiting u in bit-reversed order (u' = 0..2^E-1,
//     u = bitrev_E(u')) visits residues by ascending rank; k = tid + T*u + bs*i.
// Either way the thread's local order j = u'*I + i has strictly increasing priority
// pri = (rank << 23) | (k div bs), so a strict > over j keeps exactly the candidate the
// reference's per-thread scan + tie-keeping tree would keep.
//
// Distances are >= +0, so their IEEE bit patterns order like signed integers; the running
// distance of a never-eligible point is -1.0f (a negative int).  min / compare / max therefore
// run as 1-instruction integer ops with no NaN-canonicalisation, and the wave arg-max is two
// DPP row reductions (value, then priority among the lanes holding it) finished by readlanes.
// Cross-wave: lane 0 of each wave does ONE ds_max_u64 on a rotating LDS slot, one barrier, one
// broadcast read.  Slot (it-1)%3 is cleared right after barrier `it` (every wave has read it
// before arriving there, nobody adds to it before barrier it+1).
//
// Sampling chains.  When a cloud is the sample list of a previous FPS call, in sampling order (the
// pyramid: level l samples level l-1's samples), its own FPS is a PREFIX of that list as long as
// every arg-max of the previous call was unique: sample i+1 maximised the distance to samples 0..i
// over the whole parent cloud, so it also maximises it over the subset, and the distances are the
// same fp32 expressions on the same coordinates.  Only a tie -- two points at exactly the same
// maximal distance V, which the two calls break by different index-based priorities -- can make the
// sequences differ, and the common tie is simple: points a, b tie at iteration i, a is taken, b keeps
// its distance V (it is far from a) and is taken, alone at V, at iteration i+1.  The child call then
// sees the same two points tie at list positions i and i+1, takes the one whose POSITION has the
// better priority, then the other: the prefix with an adjacent pair possibly swapped (and the state
// after both picks is the same again).  ~5 % of 8192-point lidar clouds have one such tie in their
// first 1023 decisions (exact fp32 equality among the top candidates).
//   `tie_out` (FPS_CHAIN_INTS ints per cloud) / `tie_iters`: this call records, for its decisions
//   1..tie_iters-1, the iterations at which a simple tie happened, or raises the fallback flag for
//   anything else (tie not resolved alone at V on the next iteration, two ties in a row, no candidate,
//   more than 8 events).  `prefix_in`: a later call on this call's samples (m <= tie_iters) writes the
//   prefix with those swaps directly when the flag is clear and runs the full algorithm otherwise.
// Results are identical to always running the full algorithm (tests/test_gpu_fused.py).
constexpr int FPS_CHAIN_INTS = 12;    // [0] fallback flag, [1] number of events, [2..9] event iterations
constexpr int FPS_CHAIN_MAXEV = 8;

// XCHG (round 3, multi-wave kernels with the LDS table): the winner's coordinates travel WITH the exchange instead of
// being looked up after it.  Every lane requests the coordinates of its own best candidate from the LDS table right after
// the update loop -- the read's latency hides under the two wave reductions -- and the lane that wins its wave stores them
// in the wave's slot of a double-buffered 8-entry table before the barrier; after the barrier the slot key and the eight
// coordinate slots are read in ONE round trip and the winner's are picked with v_readlane (its wave follows from the key).
// Removes the dependent table read (key -> index -> coordinates) from the head of the next iteration.  Same arithmetic,
// same winner: bit-identical output (tests/test_gpu_ops.py FPS cases run both forms).
// CPW = 2 (round 3, tools/fps_pair_probe.py): TWO clouds per workgroup, T threads each, every cloud's eight waves meeting
// at their OWN barrier -- an LDS counter the waves add to and poll -- so that the two dependent chains run side by side on
// one CU and fill each other's exchange stalls (one chain leaves a third of the CU's issue slots idle: co-resident chains
// measured -17 % CU-time per cloud at two, -28 % at four per CU).  Needs the global winner lookup (LDS_TABLE = false: two
// 131 KB tables do not fit) and neither `prefix_in` nor `done` (their early exits are per cloud).  Same arithmetic, same
// winner: bit-identical output.
template <int T, int E, int I, bool LDS_TABLE, bool XCHG = false, int CPW = 1>
__global__ __launch_bounds__(T * CPW) void fps_reg_kernel(int n, int m, int bs, int log2bs,
                                                    const float *__restrict__ dataset,
                                                    int *__restrict__ idxs,
                                                    float *__restrict__ new_xyz,
                                                    int *__restrict__ tie_out, int tie_iters,
                                                    const int *__restrict__ prefix_in,
                                                    const int *__restrict__ done = nullptr) {
  TraceScope trace_scope_(TK_FPS);
  constexpr int PPT = I << E;
  constexpr int NW_ = T / 64;
  static_assert(CPW == 1 || (CPW == 2 && !LDS_TABLE && !XCHG), "two clouds per workgroup: global winner lookup only");
  const int half = CPW == 1 ? 0 : (int)(threadIdx.x / T);          // which of the workgroup's clouds (wave-uniform)
  const int cloud = (int)blockIdx.x * CPW + half;
  if (CPW == 1 && done != nullptr && done[blockIdx.x] != 0) return;   // this cloud was sampled by fps_slab_kernel (workgroup-uniform)
  if (CPW == 1 && prefix_in != nullptr && prefix_in[blockIdx.x * FPS_CHAIN_INTS] == 0) {       // workgroup-uniform
    const int *rec = prefix_in + blockIdx.x * FPS_CHAIN_INTS;
    const int nev = rec[1];
    const float *src = dataset + (size_t)blockIdx.x * n * 3;
    for (int t = threadIdx.x; t < m; t += T) {
      int pick = t;
      for (int e = 0; e < nev; ++e) {
        const int it = rec[2 + e];                 // tie between list positions it and it+1
        if ((t == it || t == it + 1) && it + 1 < n) {
          const unsigned p0 = (fps_bitrev((unsigned)(it & (bs - 1)), log2bs) << PRI_SHIFT) | (unsigned)(it >> log2bs);
          const unsigned p1 = (fps_bitrev((unsigned)((it + 1) & (bs - 1)), log2bs) << PRI_SHIFT) |
                              (unsigned)((it + 1) >> log2bs);
          const int first = p0 < p1 ? it : it + 1;
          pick = t == it ? first : (first == it ? it + 1 : it);
        }
      }
      idxs[(size_t)blockIdx.x * m + t] = pick;
      if (new_xyz) {
        float *d = new_xyz + ((size_t)blockIdx.x * m + t) * 3;
        d[0] = src[pick * 3 + 0]; d[1] = src[pick * 3 + 1]; d[2] = src[pick * 3 + 2];
      }
    }
    return;
  }
  // a sampler wave is one link of a long dependent chain: whenever it can issue, it should, ahead of
  // the throughput kernels of other in-flight batches that may share its SIMD
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  // CPW == 2: each cloud has its own slot header + chain log, one after the other
  unsigned char *smem = smem_all + (CPW == 1 ? 0 : (size_t)half * ((FPS_SLOT_BYTES + (size_t)(tie_out ? (tie_iters + 2) * 8 : 0) + 15) & ~(size_t)15));
  unsigned *arrive = reinterpret_cast<unsigned *>(smem + 48);                // CPW == 2: this cloud's barrier counter
  unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem);  // [3] rotating
  float4 *table = reinterpret_cast<float4 *>(smem + FPS_SLOT_BYTES);         // [n] when LDS_TABLE
  // chain log (only when tie_out): winning value and tie status of iterations 0..tie_iters+1, judged
  // once after the loop instead of inside it
  unsigned *cvals = reinterpret_cast<unsigned *>(smem + FPS_SLOT_BYTES + (LDS_TABLE ? (size_t)n * 16 : 0));
  int *cstat = reinterpret_cast<int *>(cvals + (tie_iters + 2));
  int *cmeta = reinterpret_cast<int *>(smem + 32);                           // [0] events, [1] fallback (slot header)

  const int tid = CPW == 1 ? (int)threadIdx.x : (int)(threadIdx.x % T);
  const int lane = tid & 63;
  constexpr int NW = T / 64;
  const float *pts = dataset + (size_t)cloud * n * 3;
  int *out = idxs + (size_t)cloud * m;
  float *oxyz = new_xyz ? new_xyz + (size_t)cloud * m * 3 : nullptr;  // optional (b,m,3) samples
  const int kstride = E == 0 ? T : bs;  // distance between consecutive points of one residue

  float x[PPT], y[PPT], z[PPT];
  int td[PPT];  // running min distance as IEEE bits; bits(-1.0f) < 0 marks "never a candidate"
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int up = j / I, i = j % I;
    const int u = E == 0 ? 0 : (int)(__brev((unsigned)up) >> (32 - (E == 0 ? 1 : E)));
    const int k = tid + T * u + kstride * i;
    float px = 0.f, py = 0.f, pz = 0.f, t0 = -1.0f;
    if (k < n) {
      px = pts[k * 3 + 0];
      py = pts[k * 3 + 1];
      pz = pts[k * 3 + 2];
      const float mag = (px * px) + (py * py) + (pz * pz);
      if (!((double)mag <= 1e-3)) t0 = 1e10f;  // sampling.cpp:74-76 initial temp
      if (LDS_TABLE) table[k] = make_float4(px, py, pz, 0.f);
    }
    x[j] = px; y[j] = py; z[j] = pz; td[j] = __float_as_int(t0);
  }
  // priority of local point j: pri_base + (u' << 23) + qstep * i
  const unsigned pri_base = (fps_bitrev((unsigned)(tid & (bs - 1)), log2bs) << PRI_SHIFT) |
                            (E == 0 ? (unsigned)(tid / bs) : 0u);
  const unsigned qstep = E == 0 ? (unsigned)(T / bs) : 1u;
  if (tid == 0) {
    out[0] = 0;
    slots[0] = 0ull; slots[1] = 0ull; slots[2] = 0ull;
    cmeta[0] = 0; cmeta[1] = 0;
    if (CPW > 1) *arrive = 0u;
  }
  if (tie_out != nullptr)
    for (int i = tid; i < tie_iters + 2; i += T) { cstat[i] = 0; cvals[i] = 0u; }
  __syncthreads();

  constexpr bool XC = XCHG && LDS_TABLE && NW_ > 1;
  __shared__ float4 wxyz[2][XC ? NW_ : 1];      // [iteration parity][wave]: coordinates of each wave's local winner
  float nx1 = 0.f, ny1 = 0.f, nz1 = 0.f;        // XC: coordinates of the current sample, carried over from the exchange
  if (XC) {
    const float4 p0 = table[0];
    nx1 = p0.x; ny1 = p0.y; nz1 = p0.z;
  }
  int old = 0;
  // One iteration; TRACK additionally records whether the arg-max was unique.  Two instantiations run
  // back to back (iterations < tie_iters, then the rest) rather than one loop with a branch inside:
  // merging the two variants' register arrays at a join cost 50 VGPRs and a third of the speed.
  auto iteration = [&](auto track_tag, int it) {
    constexpr bool TRACK = decltype(track_tag)::value;
    float x1, y1, z1;
    if (XC) {
      x1 = nx1; y1 = ny1; z1 = nz1;
    } else if (LDS_TABLE) {
      const float4 p = table[old];
      x1 = p.x; y1 = p.y; z1 = p.z;
    } else {
      x1 = pts[old * 3 + 0]; y1 = pts[old * 3 + 1]; z1 = pts[old * 3 + 2];
    }
    if (oxyz && tid == 0) {  // coordinates of sample it-1 (gather_operation fused into the sampler)
      oxyz[(it - 1) * 3 + 0] = x1; oxyz[(it - 1) * 3 + 1] = y1; oxyz[(it - 1) * 3 + 2] = z1;
    }
    int best = __float_as_int(-1.0f);
    int best2 = __float_as_int(-1.0f);   // TRACK: the lane's second-largest value (1 more integer op per point)
    int bestj = 0;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const float dx = x[j] - x1, dy = y[j] - y1, dz = z[j] - z1;
      const float d = dx * dx + dy * dy + dz * dz;  // -ffp-contract=off: (a+b)+c, no FMA
      const int d2 = min(__float_as_int(d), td[j]); // == fminf for d >= +0, td >= +0 or td == -1.0f
      td[j] = d2;
      if (TRACK)   // best >= best2 always, so the new runner-up is the median of (best, best2, d2): one instruction
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(best2) : "v"(best), "v"(best2), "v"(d2));
      const bool better = d2 > best;
      bestj = better ? j : bestj;
      best = better ? d2 : best;
    }
    float4 mycand = make_float4(0.f, 0.f, 0.f, 0.f);
    if (XC) {   // coordinates of this lane's own candidate: requested now, needed after the wave reductions
      const int u = E == 0 ? 0 : (int)(__brev((unsigned)(bestj / I)) >> (32 - (E == 0 ? 1 : E)));
      const int kc = tid + T * u + kstride * (bestj % I);
      mycand = table[kc < n ? kc : 0];
    }
    const bool lane_tie = TRACK && best >= 0 && best2 == best;
    // 0 = no candidate; otherwise bits+1 so that a legitimate distance of +0.0 stays distinct
    const unsigned mine = best < 0 ? 0u : (unsigned)best + 1u;
    const unsigned wmax = wave_reduce_u32(mine, OpMaxU32());
    const unsigned mypri = pri_base + ((unsigned)(bestj / I) << PRI_SHIFT) + qstep * (unsigned)(bestj % I);
    const bool holds = mine == wmax && mine != 0u;
    const unsigned cand = holds ? mypri : 0xFFFFFFFFu;
    const unsigned wpri = wave_reduce_u32(cand, OpMinU32());
    const unsigned long long wkey =
        wmax == 0u ? 0ull : (((unsigned long long)wmax << 32) | (unsigned long long)(0xFFFFFFFFu - wpri));
    unsigned long long key = wkey;
    bool tie = false;
    if (TRACK) {     // more than one lane at the wave's maximum, or the holder has it twice
      const unsigned long long hm = __ballot(holds);
      tie = (hm & (hm - 1ull)) != 0ull || __ballot(holds && lane_tie) != 0ull;
    }
    if (NW > 1) {
      unsigned long long *slot = slots + (it % 3);
      if (lane == 0) atomicMax(slot, key);
      if (XC && holds && mypri == wpri) wxyz[it & 1][tid >> 6] = mycand;     // exactly one lane per wave with a candidate
      if (CPW == 1) {
        __syncthreads();
      } else {
        // this cloud's own barrier: a wave's LDS operations execute in order, so once the counter shows 8 arrivals for
        // this iteration all eight arg-max updates above are in the slot
        if (lane == 0) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned want = (unsigned)NW * (unsigned)it;
        while (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) <
               (int)want)
          __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
      }
      key = *slot;
      float4 all = make_float4(0.f, 0.f, 0.f, 0.f);
      if (XC) all = wxyz[it & 1][lane & (NW - 1)];                           // same round trip as the key
      if (tid == 0) slots[(it + 2) % 3] = 0ull;
      if (XC) {
        if (key == 0ull) {                       // no candidate anywhere: the reference falls back to index 0
          const float4 p0 = table[0];
          nx1 = p0.x; ny1 = p0.y; nz1 = p0.z;
        } else {
          const unsigned pw = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
          const int kw = (int)fps_bitrev(pw >> PRI_SHIFT, log2bs) + bs * (int)(pw & ((1u << PRI_SHIFT) - 1u));
          const int wsel = __builtin_amdgcn_readfirstlane((kw & (T - 1)) >> 6);   // the wave that owns point kw
          nx1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(all.x), wsel));
          ny1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(all.y), wsel));
          nz1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(all.z), wsel));
        }
      }
      // another wave reached the same maximal distance with a different point
      if (TRACK) tie = tie || (wmax != 0u && wmax == (unsigned)(key >> 32) && wkey != key);
    }
    if (TRACK) {   // log only; judged after the loop
      if (lane == 0 && (tie || key == 0ull)) cstat[it] = key == 0ull ? 2 : 1;
      if (tid == 0) cvals[it] = (unsigned)(key >> 32);
    }
    if (key == 0ull) {
      old = 0;
    } else {
      const unsigned p = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
      old = (int)fps_bitrev(p >> PRI_SHIFT, log2bs) + bs * (int)(p & ((1u << PRI_SHIFT) - 1u));
    }
    if (tid == 0) out[it] = old;
  };
  int it = 1;
  // decisions 1..tie_iters-1 matter; the one at tie_iters-1 is only judged two iterations later
  const int tracked_end = tie_out != nullptr ? (tie_iters + 2 < m ? tie_iters + 2 : m) : 1;
  for (; it < tracked_end; ++it) iteration(std::true_type{}, it);
  for (; it < m; ++it) iteration(std::false_type{}, it);
  if (oxyz && tid == 0) {
    oxyz[(m - 1) * 3 + 0] = pts[old * 3 + 0];
    oxyz[(m - 1) * 3 + 1] = pts[old * 3 + 1];
    oxyz[(m - 1) * 3 + 2] = pts[old * 3 + 2];
  }
  (void)nx1; (void)ny1; (void)nz1;
  if (tie_out != nullptr) {
    // Judge the log: a tie at iteration i < tie_iters is a simple event iff iteration i+1 had no tie
    // and won with the same value (the other tied point, alone at V); anything else -> fallback.
    int *rec = tie_out + cloud * FPS_CHAIN_INTS;
    const bool enough = m >= tie_iters + 2;                 // decisions tie_iters-1 and tie_iters both made
    __syncthreads();
    if (enough) {
      for (int i = 1 + tid; i < tie_iters; i += T) {
        const int st = cstat[i];
        if (st == 0) continue;
        if (st == 2 || cstat[i + 1] != 0 || cvals[i + 1] != cvals[i] || (i > 1 && cstat[i - 1] != 0)) {
          cmeta[1] = 1;
        } else {
          const int e = atomicAdd(&cmeta[0], 1);
          if (e < FPS_CHAIN_MAXEV) rec[2 + e] = i; else cmeta[1] = 1;
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      rec[0] = (!enough || cmeta[1] != 0) ? 1 : 0;
      rec[1] = cmeta[0] < FPS_CHAIN_MAXEV ? cmeta[0] : FPS_CHAIN_MAXEV;
    }
  }
}

// ---- bucket-pruned sampler (level 1 of the pyramid: n ~ 8192) ---------------------------------------------------
// The distance update touches every point every iteration although, once a few dozen samples exist, a new sample
// can only lower the running distance of points near it.  Exact pruning, per BUCKET of 64 points: the neighbour
// search of the same cloud has already sorted it by (x-slab, z-bin) (knn_build_kernel), so 64 consecutive valid
// rows form a compact cell.  Bucket b = rows [64b, 64b + 64) lives in slot b / 8 of wave b % 8 (neighbouring cells
// on different waves and SIMDs), one point per lane.  With (lo, hi) the bucket's bounding box and V its largest
// running distance, the bucket's update can be skipped whenever
//     bd = gx*gx + gy*gy + gz*gz >= V,   g = max(lo - s, s - hi, 0) per axis (s = the new sample),
// because every point p of the bucket has d(p, s) >= bd in fp32 as well -- subtraction, squaring and the two
// additions are monotone under round-to-nearest and the expression is the one the update evaluates -- so
// min(d, running) = running for all of them and the bucket's cached summary (max, best priority at the max,
// "attained twice") is still what a full update would produce.  The bucket holding the sample itself is never
// skipped (bd = 0).  Lane j < I of a wave holds bucket j's box and summary; per iteration the 16 bounds are one
// vector expression, the wave updates only the buckets of the resulting mask (measured on the benchmark clouds:
// ~4 of 128 per iteration, ~9 during the first 200) and re-reduces their summaries; a wave with an empty mask reuses
// its cached arg-max.  Priorities derive from the ORIGINAL index each sorted row carries, hence the same winner as
// fps_reg_kernel for every input (tests: lattice clouds, zero padding, chain records, benchmark clouds bit for bit).
// LDS_TABLE = false: the winner's coordinates come from the original cloud in global memory (L2-resident) instead of a
// 16-byte-per-point LDS copy: the workgroup then needs ~8 KiB of LDS instead of 139 KiB, so that SEVERAL pruned chains
// can share a CU (each uses well under half of its issue slots; round 3 experiment, tools/fps_pair_probe.py).
template <int T, int I, bool LDS_TABLE = true>
__global__ __launch_bounds__(T) void fps_slab_kernel(int n, int m, int bs, int log2bs, int nblk,
                                                     const float4 *__restrict__ rows_all,
                                                     const int *__restrict__ slab_tab,
                                                     int *__restrict__ idxs, float *__restrict__ new_xyz,
                                                     int *__restrict__ tie_out, int tie_iters,
                                                     int *__restrict__ status, int dbg,
                                                     const float *__restrict__ dataset = nullptr) {
  TraceScope trace_scope_(TK_FPS);
  constexpr int NW = T / 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int *tab = slab_tab + (size_t)blockIdx.x * 32;
  int sstart[NW], scum[NW + 1];                             // slab s: rows [sstart[s], +count), valid ranks [scum[s], scum[s+1])
  scum[0] = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) { sstart[w] = tab[2 * w]; scum[w + 1] = scum[w] + tab[2 * w + 1]; }
  if (tid == 0) status[blockIdx.x] = 1;
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long *slots = reinterpret_cast<unsigned long long *>(smem);  // [3] rotating
  float4 *table = reinterpret_cast<float4 *>(smem + FPS_SLOT_BYTES);         // [n], by ORIGINAL index (LDS_TABLE)
  unsigned *cvals = reinterpret_cast<unsigned *>(smem + FPS_SLOT_BYTES + (LDS_TABLE ? (size_t)n * 16 : 0));
  int *cstat = reinterpret_cast<int *>(cvals + (tie_iters + 2));
  int *cmeta = reinterpret_cast<int *>(smem + 32);
  const float *opts = LDS_TABLE ? nullptr : dataset + (size_t)blockIdx.x * n * 3;

  const float4 *rows = rows_all + (size_t)blockIdx.x * nblk * 64;
  int *out = idxs + (size_t)blockIdx.x * m;
  float *oxyz = new_xyz ? new_xyz + (size_t)blockIdx.x * m * 3 : nullptr;

  float x[I], y[I], z[I];
  int td[I];
  unsigned pri[I];
  const float INF = __int_as_float(0x7f800000);
  // bucket j of this wave: box and summary live in lane j
  float blox = INF, bloy = INF, bloz = INF, bhix = -INF, bhiy = -INF, bhiz = -INF;
#pragma unroll
  for (int j = 0; j < I; ++j) {
    const int r = (j * NW + wave) * 64 + lane;               // rank among the cloud's valid rows
    float px = 0.f, py = 0.f, pz = 0.f, t0 = -1.0f;
    unsigned pr = 0xFFFFFFFFu;
    float lx = INF, ly = INF, lz = INF, hx = -INF, hy = -INF, hz = -INF;
    if (r < n) {
      int row = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w)
        if (r >= scum[w] && r < scum[w + 1]) row = sstart[w] + (r - scum[w]);
      const float4 c = rows[row];
      px = c.x; py = c.y; pz = c.z;
      const unsigned k = __float_as_uint(c.w);              // original index
      pr = (fps_bitrev(k & (unsigned)(bs - 1), log2bs) << PRI_SHIFT) | (k >> log2bs);
      const float mag = (px * px) + (py * py) + (pz * pz);
      if (!((double)mag <= 1e-3)) {                         // never-eligible points stay outside the box
        t0 = 1e10f;
        lx = hx = px; ly = hy = py; lz = hz = pz;
      }
      if (LDS_TABLE) table[k] = make_float4(px, py, pz, 0.f);
    }
    x[j] = px; y[j] = py; z[j] = pz; td[j] = __float_as_int(t0); pri[j] = pr;
    lx = wave_allreduce_f32(lx, [](float a, float b) { return fminf(a, b); });
    ly = wave_allreduce_f32(ly, [](float a, float b) { return fminf(a, b); });
    lz = wave_allreduce_f32(lz, [](float a, float b) { return fminf(a, b); });
    hx = wave_allreduce_f32(hx, [](float a, float b) { return fmaxf(a, b); });
    hy = wave_allreduce_f32(hy, [](float a, float b) { return fmaxf(a, b); });
    hz = wave_allreduce_f32(hz, [](float a, float b) { return fmaxf(a, b); });
    if (lane == j) { blox = lx; bloy = ly; bloz = lz; bhix = hx; bhiy = hy; bhiz = hz; }
  }
  if (tid == 0) {
    out[0] = 0;
    slots[0] = 0ull; slots[1] = 0ull; slots[2] = 0ull;
    cmeta[0] = 0; cmeta[1] = 0;
  }
  if (tie_out != nullptr)
    for (int i = tid; i < tie_iters + 2; i += T) { cstat[i] = 0; cvals[i] = 0u; }
  __syncthreads();

  // Running distances are kept as bits + 1 (0 = never a candidate), so that (distance, ~priority) is ONE unsigned
  // 64-bit key per point and the in-lane arg-max is a 64-bit compare + two selects per slot.
  unsigned tdm[I], npri[I];
#pragma unroll
  for (int j = 0; j < I; ++j) { tdm[j] = td[j] < 0 ? 0u : (unsigned)td[j] + 1u; npri[j] = ~pri[j]; }
  // the wave's cached arg-max (valid while none of its buckets is updated)
  unsigned wmax = 0u, wpri = 0xFFFFFFFFu;
  bool wtie = false;
  // Upper bound of EVERY running distance: the previous winner's value (it was the global maximum).  A bucket whose
  // box is at least that far from the new sample cannot change; no per-bucket maximum has to be maintained.
  unsigned gmax = 0x7FFFFFFFu;
  int old = 0;
  auto iteration = [&](auto track_tag, int it) {
    constexpr bool TRACK = decltype(track_tag)::value;
    float x1, y1, z1;
    if (LDS_TABLE) {
      const float4 p1 = table[old];
      x1 = p1.x; y1 = p1.y; z1 = p1.z;
    } else {
      x1 = opts[old * 3 + 0]; y1 = opts[old * 3 + 1]; z1 = opts[old * 3 + 2];
    }
    if (oxyz && tid == 0) {
      oxyz[(it - 1) * 3 + 0] = x1; oxyz[(it - 1) * 3 + 1] = y1; oxyz[(it - 1) * 3 + 2] = z1;
    }
    // lane j: lower bound of d(p, sample) over bucket j's box, in the update's own arithmetic
    const float gx = fmaxf(fmaxf(blox - x1, x1 - bhix), 0.f);
    const float gy = fmaxf(fmaxf(bloy - y1, y1 - bhiy), 0.f);
    const float gz = fmaxf(fmaxf(bloz - z1, z1 - bhiz), 0.f);
    const float bd = gx * gx + gy * gy + gz * gz;
    unsigned long long todo = __ballot(lane < I && (unsigned)__float_as_int(bd) + 1u < gmax);
    if (dbg == 1 && it > 1) todo = 0ull;                    // timing probe: overhead floor (wrong results)
    if (todo != 0ull) {
      auto update = [&](auto jtag) {
        constexpr int j = decltype(jtag)::value;
        // opaque copies of the sample: without them the compiler hoists the distance arithmetic of ALL sixteen
        // cases above the switch (speculation), i.e. it undoes the pruning
        float sx = x1, sy = y1, sz = z1;
        asm volatile("" : "+v"(sx), "+v"(sy), "+v"(sz));
        const float dx = x[j] - sx, dy = y[j] - sy, dz = z[j] - sz;
        const float d = dx * dx + dy * dy + dz * dz;       // -ffp-contract=off: (a+b)+c, no FMA
        tdm[j] = min((unsigned)__float_as_int(d) + 1u, tdm[j]);          // 0 (ineligible) stays 0
      };
      while (todo != 0ull) {
        const int j = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        switch (j) {
#define FSL_CASE(J) case J: if constexpr (J < I) update(std::integral_constant<int, J>{}); break;
          FSL_CASE(0) FSL_CASE(1) FSL_CASE(2) FSL_CASE(3) FSL_CASE(4) FSL_CASE(5) FSL_CASE(6) FSL_CASE(7)
          FSL_CASE(8) FSL_CASE(9) FSL_CASE(10) FSL_CASE(11) FSL_CASE(12) FSL_CASE(13) FSL_CASE(14) FSL_CASE(15)
          FSL_CASE(16) FSL_CASE(17) FSL_CASE(18) FSL_CASE(19)
#undef FSL_CASE
          default: break;
        }
      }
      // the lane's arg-max over its slots (independent 64-bit compares), then the wave's
      unsigned bhi = 0u, blo = 0u, b2 = 0u;
#pragma unroll
      for (int j = 0; j < I; ++j) {
        if (TRACK) asm("v_med3_u32 %0, %1, %2, %3" : "=v"(b2) : "v"(bhi), "v"(b2), "v"(tdm[j]));   // runner-up value
        const unsigned long long kj = ((unsigned long long)tdm[j] << 32) | npri[j];
        const bool better = kj > (((unsigned long long)bhi << 32) | blo);
        bhi = better ? tdm[j] : bhi;
        blo = better ? npri[j] : blo;
      }
      const bool lane_tie = TRACK && bhi != 0u && b2 == bhi;
      wmax = wave_reduce_u32(bhi, OpMaxU32());
      const bool holds = bhi == wmax && bhi != 0u;
      wpri = ~wave_reduce_u32(holds ? blo : 0u, OpMaxU32());
      if (TRACK) {
        const unsigned long long hm = __ballot(holds);
        wtie = (hm & (hm - 1ull)) != 0ull || __ballot(holds && lane_tie) != 0ull;
      }
    }
    const unsigned long long wkey =
        wmax == 0u ? 0ull : (((unsigned long long)wmax << 32) | (unsigned long long)(0xFFFFFFFFu - wpri));
    unsigned long long key = wkey;
    {
      unsigned long long *slot = slots + (it % 3);
      if (lane == 0) atomicMax(slot, key);
      __syncthreads();
      key = *slot;
      if (tid == 0) slots[(it + 2) % 3] = 0ull;
    }
    gmax = (unsigned)(key >> 32);                          // bits + 1 of the winner's distance: bounds every point
    if (TRACK) {
      // a tie = the GLOBAL maximum attained by two points: inside this wave (wtie) or by another wave (wkey != key)
      const bool at_max = wmax != 0u && wmax == (unsigned)(key >> 32);
      const bool tie = at_max && (wtie || wkey != key);
      if (lane == 0 && (tie || key == 0ull)) cstat[it] = key == 0ull ? 2 : 1;
      if (tid == 0) cvals[it] = (unsigned)(key >> 32);
    }
    if (key == 0ull) {
      old = 0;
    } else {
      const unsigned p = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
      old = (int)fps_bitrev(p >> PRI_SHIFT, log2bs) + bs * (int)(p & ((1u << PRI_SHIFT) - 1u));
    }
    if (tid == 0) out[it] = old;
  };
  int it = 1;
  const int tracked_end = tie_out != nullptr ? (tie_iters + 2 < m ? tie_iters + 2 : m) : 1;
  for (; it < tracked_end; ++it) iteration(std::true_type{}, it);
  for (; it < m; ++it) iteration(std::false_type{}, it);
  if (oxyz && tid == 0) {
    if (LDS_TABLE) {
      const float4 p = table[old];
      oxyz[(m - 1) * 3 + 0] = p.x; oxyz[(m - 1) * 3 + 1] = p.y; oxyz[(m - 1) * 3 + 2] = p.z;
    } else {
      oxyz[(m - 1) * 3 + 0] = opts[old * 3 + 0]; oxyz[(m - 1) * 3 + 1] = opts[old * 3 + 1]; oxyz[(m - 1) * 3 + 2] = opts[old * 3 + 2];
    }
  }
  if (tie_out != nullptr) {      // judge the log exactly as fps_reg_kernel does
    int *rec = tie_out + blockIdx.x * FPS_CHAIN_INTS;
    const bool enough = m >= tie_iters + 2;
    __syncthreads();
    if (enough) {
      for (int i = 1 + tid; i < tie_iters; i += T) {
        const int st = cstat[i];
        if (st == 0) continue;
        if (st == 2 || cstat[i + 1] != 0 || cvals[i + 1] != cvals[i] || (i > 1 && cstat[i - 1] != 0)) {
          cmeta[1] = 1;
        } else {
          const int e = atomicAdd(&cmeta[0], 1);
          if (e < FPS_CHAIN_MAXEV) rec[2 + e] = i; else cmeta[1] = 1;
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      rec[0] = (!enough || cmeta[1] != 0) ? 1 : 0;
      rec[1] = cmeta[0] < FPS_CHAIN_MAXEV ? cmeta[0] : FPS_CHAIN_MAXEV;
    }
  }
}

__global__ void fps_fill_flag_kernel(int *rec, int n, int v) {   // fallback flag of n chain records
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { rec[i * FPS_CHAIN_INTS] = v; rec[i * FPS_CHAIN_INTS + 1] = 0; }
}

// Fallback for clouds too large for the register file: same selection rule, running distances in
// the caller's temp buffer (pre-filled with 1e10), points re-read from global memory (L2).
template <int T>
__global__ __launch_bounds__(T) void fps_stream_kernel(int n, int m, int bs, int log2bs,
                                                       const float *__restrict__ dataset,
                                                       float *__restrict__ temp,
                                                       int *__restrict__ idxs,
                                                       float *__restrict__ new_xyz) {
  __shared__ unsigned long long slots[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = T / 64;
  const float *pts = dataset + (size_t)blockIdx.x * n * 3;
  float *tmp = temp + (size_t)blockIdx.x * n;
  int *out = idxs + (size_t)blockIdx.x * m;
  float *oxyz = new_xyz ? new_xyz + (size_t)blockIdx.x * m * 3 : nullptr;
  const unsigned rev = fps_bitrev((unsigned)(tid & (bs - 1)), log2bs) << PRI_SHIFT;
  if (tid == 0) out[0] = 0;
  int old = 0;
  for (int it = 1; it < m; ++it) {
    const float x1 = pts[old * 3 + 0], y1 = pts[old * 3 + 1], z1 = pts[old * 3 + 2];
    if (oxyz && tid == 0) {
      oxyz[(it - 1) * 3 + 0] = x1; oxyz[(it - 1) * 3 + 1] = y1; oxyz[(it - 1) * 3 + 2] = z1;
    }
    float best = -1.0f;
    unsigned bestpri = 0xFFFFFFFFu;
    for (int k = tid; k < n; k += T) {
      const float px = pts[k * 3 + 0], py = pts[k * 3 + 1], pz = pts[k * 3 + 2];
      const float mag = (px * px) + (py * py) + (pz * pz);
      if ((double)mag <= 1e-3) continue;
      const float dx = px - x1, dy = py - y1, dz = pz - z1;
      const float d = dx * dx + dy * dy + dz * dz;
      const float d2 = fminf(d, tmp[k]);
      tmp[k] = d2;
      if (d2 > best) {  // k ascending within a thread => priorities ascending
        best = d2;
        bestpri = rev | (unsigned)(k >> log2bs);
      }
    }
    const float wmax = wave_allreduce_f32(best, [](float a, float b) { return fmaxf(a, b); });
    unsigned pri = (best == wmax && best >= 0.0f) ? bestpri : 0xFFFFFFFFu;
    pri = wave_allreduce_u32(pri, [](unsigned a, unsigned b) { return a < b ? a : b; });
    unsigned long long key = 0ull;
    if (wmax >= 0.0f)
      key = ((unsigned long long)__float_as_uint(wmax) << 32) | (unsigned long long)(0xFFFFFFFFu - pri);
    if (lane == 0) slots[it & 1][wave] = key;
    __syncthreads();
    unsigned long long kmax = slots[it & 1][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const unsigned long long o = slots[it & 1][w];
      kmax = o > kmax ? o : kmax;
    }
    if (kmax == 0ull) {
      old = 0;
    } else {
      const unsigned p = 0xFFFFFFFFu - (unsigned)(kmax & 0xFFFFFFFFull);
      old = (int)fps_bitrev(p >> PRI_SHIFT, log2bs) + bs * (int)(p & ((1u << PRI_SHIFT) - 1u));
    }
    if (tid == 0) out[it] = old;
  }
  if (oxyz && tid == 0) {
    oxyz[(m - 1) * 3 + 0] = pts[old * 3 + 0];
    oxyz[(m - 1) * 3 + 1] = pts[old * 3 + 1];
    oxyz[(m - 1) * 3 + 2] = pts[old * 3 + 2];
  }
}

// Clouds too large for one workgroup's registers (n > 24576, e.g. raw 120k-point KITTI-360 frames
// sampled to 8192): G workgroups of 1024 threads share one cloud, every point still lives in a VGPR
// for the whole call, and each iteration ends in one cross-workgroup exchange through global memory:
//   workgroup-local arg-max (as fps_reg_kernel) -> the lane that OWNS the local winner posts
//   (value, priority, x, y, z) into its workgroup's slot, five 64-bit words each tagged with the iteration
//   number -> wave 0 of every workgroup polls the G slots (lane g reads workgroup g's five words) until
//   all carry the current tag, reduces them to the global winner and broadcasts key AND coordinates
//   through LDS.  One store burst and (typically) one or two poll round trips per iteration; no atomic
//   read-modify-write, no arrival counter, no re-zeroing, and the next iteration needs no coordinate load.
//   Two slot sets alternate: a workgroup can only post iteration it+2 after every workgroup has posted it+1,
//   i.e. after every workgroup has finished polling iteration it.
// Virtual thread id vtid = g*1024 + tid over T_total = G*1024 >= bs threads reproduces exactly the
// E == 0 indexing of fps_reg_kernel (thread owns residue vtid mod bs, points k = vtid + T_total*i),
// hence the same tie rule.  All G*clouds workgroups of a launch must be resident together (the host
// launches at most 224 at a time); every spin is bounded, a timeout raises the error word and ends
// the kernel.  Workspace (the reference's `temp` scratch): COOP_WS_WORDS u64 per cloud, zeroed by
// fps_coop_init (tag 0 never matches an iteration >= 1).
constexpr int COOP_T = 1024;
constexpr int COOP_MAX_G = 32;
constexpr int COOP_SLOT_WORDS = 5;                                       // value, priority, x, y, z
constexpr int COOP_WS_WORDS = 2 * COOP_MAX_G * COOP_SLOT_WORDS + 8;       // two sets + error word (+ padding)

__global__ void fps_coop_init_kernel(unsigned long long *ws, int nwords) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nwords) ws[i] = 0ull;
}

// `local` = every workgroup of the cloud sits on ONE XCD (verified at kernel start): a plain store stays in that XCD's L2,
// where the peers' L1-bypassing polls find it (an agent-scope store writes through to the fabric and drops the line, so
// the poll pays the memory round trip).  Never used when the workgroups' XCC ids differ: other XCDs would not see it.
__device__ __forceinline__ void coop_post(unsigned long long *w, unsigned payload, unsigned tag, bool local = false) {
  const unsigned long long v = ((unsigned long long)payload << 32) | (unsigned long long)tag;
  if (local)
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(w), "v"(v) : "memory");
  else
    __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// `spin_limit` bounds every poll loop; on expiry the workgroup raises the workspace's error word (so its peers
// stop too), posts PWCLO_ECOOP_TIMEOUT into the library's pinned error word `host_err` (state.hip) and ends.
// `holdback` (debug, PWCLO_FPS_COOP_DEBUG_TIMEOUT=1): the last workgroup of every cloud exits at once, which
// is what a non-resident peer looks like to the others -- used by the test of the failure path.
// SORTED (furthest_point_sampling_sorted_kernel_wrapper): `dataset` is the cloud in a spatially coherent order and
// `perm` gives every position's ORIGINAL index.  A wave then owns I * 64 consecutive positions = a compact cell, and
// its whole distance update is skipped -- exactly -- whenever the new sample is at least as far from the cell's
// bounding box as the largest running distance anywhere (the previous winner's value): every point of the cell has
// d >= that box distance in fp32 as well (monotone rounding, same expression), so min(d, running) = running and the
// lanes' cached candidates stay valid.  Within a cell the host orders positions by priority, which keeps the
// strict `>` of the per-lane scan equivalent to the reference's tie rule; priorities come from the original index
// (LDS table), so the output is the unsorted kernel's, bit for bit.  At n ~ 1e5 a sample touches a handful of the
// ~100 cells: the update leaves the critical path of most SIMDs.
template <int I, bool SORTED>
__global__ __launch_bounds__(COOP_T) void fps_coop_kernel(int n, int m, int bs, int log2bs, int G, int nclouds, int xcd_local,
                                                          const float *__restrict__ dataset,
                                                          unsigned long long *__restrict__ ws,
                                                          int *__restrict__ idxs,
                                                          float *__restrict__ new_xyz, int spin_limit,
                                                          int holdback, int poll_delay, unsigned *host_err,
                                                          const int *__restrict__ perm,
                                                          const float *__restrict__ orig_dataset) {
  extern __shared__ __attribute__((aligned(16))) unsigned pri_lds[];   // SORTED: [I][COOP_T] priorities
  __shared__ unsigned long long slots[3];
  __shared__ unsigned long long bcast;      // winning key, ~0 = timed out
  __shared__ float bxyz[3];                 // its coordinates
  // 1-D grid of 8 * G * ceil(clouds / 8) workgroups.  Blocks are dealt round-robin over the 8 XCDs (observed, not a
  // contract), so blocks with equal blockIdx.x % 8 share one: the G workgroups of a cloud are taken from one residue
  // class.  Whether that really put them on one XCD is CHECKED below (XCC ids exchanged once); only then the fast
  // exchange is used.
  const int rclass = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int cloud = rclass + 8 * (q / G), g = q % G;
  if (cloud >= nclouds) return;
  if (holdback == 1 && g == G - 1) return;            // 2, 3: timing probes (results meaningless), see coop_launch
  const int tid = threadIdx.x, lane = tid & 63;
  const int vtid = g * COOP_T + tid, ttotal = G * COOP_T;
  const float *pts = dataset + (size_t)cloud * n * 3;
  int *out = idxs + (size_t)cloud * m;
  float *oxyz = new_xyz ? new_xyz + (size_t)cloud * m * 3 : nullptr;
  unsigned long long *gws = ws + (size_t)cloud * COOP_WS_WORDS;
  unsigned long long *errw = gws + 2 * COOP_MAX_G * COOP_SLOT_WORDS;

  float x[I], y[I], z[I];
  int td[I];
  const float INF_ = __int_as_float(0x7f800000);
  float lox = INF_, loy = INF_, loz = INF_, hix = -INF_, hiy = -INF_, hiz = -INF_;   // SORTED: this wave's box
  const int *permc = SORTED ? perm + (size_t)cloud * n : nullptr;
#pragma unroll
  for (int i = 0; i < I; ++i) {
    // unsorted: residue classes of the reference's thread partition; sorted: I * 64 consecutive positions per wave
    // sorted: cell c (I * 64 consecutive positions of the spatial order) goes to workgroup c % G, wave c / G -- a new
    // sample wakes a handful of NEIGHBOURING cells, which this spreads over all workgroups and SIMDs instead of
    // piling them onto the 16 waves of one workgroup (the iteration waits for the slowest workgroup)
    const int k = SORTED ? (((tid >> 6) * G + g) * I + i) * 64 + lane : vtid + ttotal * i;
    float px = 0.f, py = 0.f, pz = 0.f, t0 = -1.0f;
    unsigned pr = 0xFFFFFFFFu;
    if (k < n) {
      px = pts[(size_t)k * 3 + 0]; py = pts[(size_t)k * 3 + 1]; pz = pts[(size_t)k * 3 + 2];
      const float mag = (px * px) + (py * py) + (pz * pz);
      if (!((double)mag <= 1e-3)) {
        t0 = 1e10f;
        if (SORTED) {
          lox = fminf(lox, px); hix = fmaxf(hix, px);
          loy = fminf(loy, py); hiy = fmaxf(hiy, py);
          loz = fminf(loz, pz); hiz = fmaxf(hiz, pz);
        }
      }
      if (SORTED) {
        const unsigned o = (unsigned)permc[k];
        pr = (fps_bitrev(o & (unsigned)(bs - 1), log2bs) << PRI_SHIFT) | (o >> log2bs);
      }
    }
    if (SORTED) pri_lds[i * COOP_T + tid] = pr;
    x[i] = px; y[i] = py; z[i] = pz; td[i] = __float_as_int(t0);
  }
  if (SORTED) {
    lox = wave_allreduce_f32(lox, [](float a, float b) { return fminf(a, b); });
    loy = wave_allreduce_f32(loy, [](float a, float b) { return fminf(a, b); });
    loz = wave_allreduce_f32(loz, [](float a, float b) { return fminf(a, b); });
    hix = wave_allreduce_f32(hix, [](float a, float b) { return fmaxf(a, b); });
    hiy = wave_allreduce_f32(hiy, [](float a, float b) { return fmaxf(a, b); });
    hiz = wave_allreduce_f32(hiz, [](float a, float b) { return fmaxf(a, b); });
  }
  const unsigned pri_base = (fps_bitrev((unsigned)(vtid & (bs - 1)), log2bs) << PRI_SHIFT) | (unsigned)(vtid / bs);
  const unsigned qstep = (unsigned)(ttotal / bs);
  if (tid == 0) {
    slots[0] = 0ull; slots[1] = 0ull; slots[2] = 0ull;
    if (g == 0) out[0] = 0;
  }
  // sample 0 is ORIGINAL point 0 (in sorted mode pts[0] is some other point)
  const float *first = SORTED ? orig_dataset + (size_t)cloud * n * 3 : pts;
  float x1 = first[0], y1 = first[1], z1 = first[2];
  if (oxyz && g == 0 && tid == 0) { oxyz[0] = x1; oxyz[1] = y1; oxyz[2] = z1; }
  __syncthreads();

  bool failed = false;
  // Placement check: every workgroup posts its XCC id (slot set 0, tag 0xFFFFFFFF: no iteration uses it, and set 0 is
  // first written at iteration 2, which no peer reaches before all have left this check); all equal -> `local`.
  bool local = false;
  if (xcd_local) {
    __shared__ int same_xcd;
    if (tid < 64) {
      const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 15u;      // HW_REG_XCC_ID[3:0]
      if (lane == 0) coop_post(gws + (size_t)g * COOP_SLOT_WORDS, xcc, 0xFFFFFFFFu);
      const unsigned long long *qs = gws + (size_t)lane * COOP_SLOT_WORDS;
      unsigned long long w0 = 0ull;
      bool done = false;
      for (int spin = 0; spin < spin_limit; ++spin) {
        bool ready = true;
        if (lane < G) {
          w0 = __hip_atomic_load(qs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ready = (unsigned)w0 == 0xFFFFFFFFu;
        }
        if (__ballot(ready) == ~0ull || holdback == 3) { done = true; break; }   // probe 3: one poll, never wait
        if ((spin & 63) == 63 && __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
        __builtin_amdgcn_s_sleep(1);
      }
      const unsigned mine_x = (unsigned)(w0 >> 32);
      const unsigned first_x = (unsigned)__builtin_amdgcn_readfirstlane((int)mine_x);
      const bool all_same = __ballot(lane < G && mine_x != first_x) == 0ull;
      if (lane == 0) same_xcd = (done && all_same) ? 1 : 0;   // a time-out here is caught by the first iteration's poll
    }
    __syncthreads();
    local = same_xcd != 0;
  }
  // SORTED: the lane's candidate and the wave's arg-max persist across iterations whose update is skipped
  int bestj = 0;
  unsigned mine = 0u, wmax = 0u, wpri = 0xFFFFFFFFu, mypri = 0xFFFFFFFFu;
  unsigned gmax = 0x7FFFFFFFu;            // bits + 1 of the previous winner's distance: bounds every running distance
  for (int it = 1; it < m && !failed; ++it) {
    bool active = true;
    if (SORTED) {
      const float gx = fmaxf(fmaxf(lox - x1, x1 - hix), 0.f);
      const float gy = fmaxf(fmaxf(loy - y1, y1 - hiy), 0.f);
      const float gz = fmaxf(fmaxf(loz - z1, z1 - hiz), 0.f);
      const float bd = gx * gx + gy * gy + gz * gz;            // the update's own expression on the box gap
      active = __builtin_amdgcn_readfirstlane((int)((unsigned)__float_as_int(bd) + 1u < gmax)) != 0;
    }
    if (holdback == 2) active = false;                  // probe: exchange + barriers only
    if (active) {
      int best = __float_as_int(-1.0f);
      bestj = 0;
#pragma unroll
      for (int j = 0; j < I; ++j) {
        const float dx = x[j] - x1, dy = y[j] - y1, dz = z[j] - z1;
        const float d = dx * dx + dy * dy + dz * dz;
        const int d2 = min(__float_as_int(d), td[j]);
        td[j] = d2;
        const bool better = d2 > best;
        bestj = better ? j : bestj;
        best = better ? d2 : best;
      }
      mine = best < 0 ? 0u : (unsigned)best + 1u;
      wmax = wave_reduce_u32(mine, OpMaxU32());
      mypri = SORTED ? pri_lds[bestj * COOP_T + tid] : pri_base + qstep * (unsigned)bestj;
      const unsigned cand = (mine == wmax && mine != 0u) ? mypri : 0xFFFFFFFFu;
      wpri = wave_reduce_u32(cand, OpMinU32());
    }
    const unsigned long long key =
        wmax == 0u ? 0ull : (((unsigned long long)wmax << 32) | (unsigned long long)(0xFFFFFFFFu - wpri));
    unsigned long long *slot = slots + (it % 3);
    if (lane == 0) atomicMax(slot, key);
    __syncthreads();
    const unsigned long long lkey = *slot;
    unsigned long long *mys = gws + ((size_t)(it & 1) * COOP_MAX_G + g) * COOP_SLOT_WORDS;
    const unsigned tag = (unsigned)it;
    if (lkey != 0ull && key == lkey && mine == wmax && mypri == wpri) {
      // exactly one lane of the workgroup: it owns the local winner and has its coordinates in registers
      float px = x[0], py = y[0], pz = z[0];
#pragma unroll
      for (int j = 1; j < I; ++j)
        if (bestj == j) { px = x[j]; py = y[j]; pz = z[j]; }
      coop_post(mys + 2, __float_as_uint(px), tag, local);
      coop_post(mys + 3, __float_as_uint(py), tag, local);
      coop_post(mys + 4, __float_as_uint(pz), tag, local);
      coop_post(mys + 1, (unsigned)(lkey & 0xFFFFFFFFull), tag, local);
      coop_post(mys + 0, (unsigned)(lkey >> 32), tag, local);
    }
    if (tid == 0) {
      slots[(it + 2) % 3] = 0ull;
      if (lkey == 0ull)                                  // no candidate in this workgroup: value 0
        for (int w = 0; w < COOP_SLOT_WORDS; ++w) coop_post(mys + w, 0u, tag, local);
    }
    if (tid < 64) {                                      // wave 0 polls: lane q reads workgroup q's slot
      const unsigned long long *qs = gws + ((size_t)(it & 1) * COOP_MAX_G + lane) * COOP_SLOT_WORDS;
      unsigned long long w0 = 0ull, w1 = 0ull, w2 = 0ull, w3 = 0ull, w4 = 0ull;
      bool done = false;
      // the peers' posts need a fabric hop to land: a first poll issued at once misses them and costs a second round
      // trip; poll_delay x 64 cycles of sleep first (tuned: PWCLO_FPS_COOP_POLL_DELAY)
      for (int d = 0; d < poll_delay; ++d) __builtin_amdgcn_s_sleep(1);
      for (int spin = 0; spin < spin_limit; ++spin) {
        bool ready = true;
        if (lane < G) {
          w0 = __hip_atomic_load(qs + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w1 = __hip_atomic_load(qs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w2 = __hip_atomic_load(qs + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w3 = __hip_atomic_load(qs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w4 = __hip_atomic_load(qs + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ready = (unsigned)w0 == tag && (unsigned)w1 == tag && (unsigned)w2 == tag && (unsigned)w3 == tag &&
                  (unsigned)w4 == tag;
        }
        if (__ballot(ready) == ~0ull || holdback == 3) { done = true; break; }   // probe 3: one poll, never wait
        if ((spin & 63) == 63 &&
            __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) break;
        __builtin_amdgcn_s_sleep(1);
      }
      const unsigned val = lane < G ? (unsigned)(w0 >> 32) : 0u;
      const unsigned low = lane < G ? (unsigned)(w1 >> 32) : 0u;      // 0xFFFFFFFF - priority
      const unsigned vmax = wave_reduce_u32(val, OpMaxU32());
      const unsigned lmax = wave_reduce_u32(val == vmax ? low : 0u, OpMaxU32());
      const unsigned long long win = __ballot(lane < G && val == vmax && low == lmax);
      const int wl = win != 0ull ? __builtin_ctzll(win) : 0;
      const float wx = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)(unsigned)(w2 >> 32), wl));
      const float wy = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)(unsigned)(w3 >> 32), wl));
      const float wz = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)(unsigned)(w4 >> 32), wl));
      if (lane == 0) {
        if (!done) {
          __hip_atomic_store(errw, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (host_err)
            __hip_atomic_store(host_err, (unsigned)PWCLO_ECOOP_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          bcast = ~0ull;
        } else {
          bcast = vmax == 0u ? 0ull : (((unsigned long long)vmax << 32) | (unsigned long long)lmax);
          bxyz[0] = wx; bxyz[1] = wy; bxyz[2] = wz;
        }
      }
    }
    __syncthreads();
    const unsigned long long kmax = bcast;
    int old = 0;
    if (kmax == ~0ull) {
      failed = true;
    } else if (kmax == 0ull) {       // nothing left to sample anywhere: index 0 again, like the reference
      x1 = first[0]; y1 = first[1]; z1 = first[2];
    } else {
      const unsigned p = 0xFFFFFFFFu - (unsigned)(kmax & 0xFFFFFFFFull);
      old = (int)fps_bitrev(p >> PRI_SHIFT, log2bs) + bs * (int)(p & ((1u << PRI_SHIFT) - 1u));
      x1 = bxyz[0]; y1 = bxyz[1]; z1 = bxyz[2];
    }
    gmax = failed ? 0u : (unsigned)(kmax >> 32);
    if (g == 0 && tid == 0 && !failed) {
      out[it] = old;
      if (oxyz) { oxyz[it * 3 + 0] = x1; oxyz[it * 3 + 1] = y1; oxyz[it * 3 + 2] = z1; }
    }
  }
}

// Per-cloud "already sampled" flags of the launch being dispatched (set by the slab wrapper around fps_dispatch).
static thread_local const int *t_done_flags = nullptr;

template <int T, int E, int I>
static void launch_fps_reg(int b, int n, int m, int bs, int log2bs, const float *dataset, int *idxs,
                           float *new_xyz, int *tie_out, int tie_iters, const int *prefix_in) {
  const size_t chain_bytes = tie_out ? (size_t)(tie_iters + 2) * 8 : 0;
  const size_t table_bytes = FPS_SLOT_BYTES + (size_t)n * sizeof(float4) + chain_bytes;
  hipStream_t st = current_stream();
  static int use_table = -1;
  if (use_table < 0) { const char *e = getenv("PWCLO_FPS_TABLE"); use_table = e ? atoi(e) : 1; }
  static int use_pair = -1;
  if (use_pair < 0) { const char *e = getenv("PWCLO_FPS_PAIR"); use_pair = e ? atoi(e) : 0; }   // opt-in: -15 % CU-time per cloud but 1.7x the chain latency (profiles/r03)
  if constexpr (T == 512 && E == 0 && I == 16) {
    // two clouds per 1024-thread workgroup, each with its own LDS barrier (see the kernel): the pyramid's first level
    if (use_pair && (b % 2) == 0 && prefix_in == nullptr && t_done_flags == nullptr) {
      auto kern = fps_reg_kernel<512, 0, 16, false, false, 2>;
      // a dynamic-LDS request of 72 KiB (the kernel uses ~17 of them): two such workgroups may share a CU, the MFMA
      // workgroups of other in-flight batches (>= 40 KiB of weights each) may not move in beside the chains
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); attr = true; }
      hipLaunchKernelGGL(kern, dim3(b / 2), dim3(1024), 72 * 1024, st, n, m, bs, log2bs, dataset, idxs, new_xyz, tie_out,
                         tie_iters, prefix_in, t_done_flags);
      return;
    }
  }
  static int use_xchg = -1;
  if (use_xchg < 0) { const char *e = getenv("PWCLO_FPS_XCHG"); use_xchg = e ? atoi(e) : 0; }   // measured SLOWER (profiles/r03/r03_fps_exchange_variant.txt): opt-in
  if (table_bytes + 512 <= 160 * 1024 && T > 64 && use_xchg && (use_table || table_bytes <= 64 * 1024)) {
    auto kern = fps_reg_kernel<T, E, I, true, true>;       // coordinates travel with the exchange
    static bool big_lds_enabled = false;
    if (table_bytes > 60 * 1024 && !big_lds_enabled) {
      (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
      big_lds_enabled = true;
    }
    hipLaunchKernelGGL(kern, dim3(b), dim3(T), table_bytes, st, n, m, bs, log2bs, dataset, idxs, new_xyz, tie_out,
                       tie_iters, prefix_in, t_done_flags);
  } else if (table_bytes <= 160 * 1024 && (use_table || table_bytes <= 64 * 1024)) {
    auto kern = fps_reg_kernel<T, E, I, true>;
    static bool big_lds_enabled = false;  // per instantiation; raises the 64 KiB dynamic-LDS default
    if (table_bytes > 64 * 1024 && !big_lds_enabled) {
      (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024);
      big_lds_enabled = true;
    }
    hipLaunchKernelGGL(kern, dim3(b), dim3(T), table_bytes, st, n, m, bs, log2bs, dataset, idxs, new_xyz, tie_out,
                       tie_iters, prefix_in, t_done_flags);
  } else {
    hipLaunchKernelGGL((fps_reg_kernel<T, E, I, false>), dim3(b), dim3(T), FPS_SLOT_BYTES + chain_bytes, st, n, m,
                       bs, log2bs, dataset, idxs, new_xyz, tie_out, tie_iters, prefix_in, t_done_flags);
  }
}

// Threads per cloud as a function of n (measured on MI355X, tools/microbench.py).
// PWCLO_FPS_THREADS=<64|128|256|512|1024> overrides it for experiments.
static int fps_pick_threads(int n, int bs) {
  static int forced = -1;
  if (forced < 0) {
    const char *e = getenv("PWCLO_FPS_THREADS");
    forced = e ? atoi(e) : 0;
  }
  int T;
  if (forced == 64 || forced == 128 || forced == 256 || forced == 512 || forced == 1024) T = forced;
  else if (n <= 256) T = 64;     // single wave: no LDS exchange, no barrier (342 ns/iter at n=256)
  else if (n <= 4096) T = 256;   // 505 ns/iter at n=2048 (512 threads: 573, 128: 671)
  else T = 512;                  // 916 ns/iter at n=8192 (1024 threads: 988, 256: 1010)
  if (T > 512 && n <= 512) T = 512;
  (void)bs;
  return T;
}

// out[b,c,j] = points[b,c,idx[b,j]]; grid (ceil(m/256), c, b) so that small m still fills CUs.
__global__ __launch_bounds__(256) void gather_points_kernel(int c, int n, int m,
                                                            const float *__restrict__ points,
                                                            const int *__restrict__ idx,
                                                            float *__restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const size_t row = (size_t)blockIdx.z * c + blockIdx.y;
  const int a = idx[(size_t)blockIdx.z * m + j];
  out[row * m + j] = points[row * n + a];
}

__global__ __launch_bounds__(256) void gather_points_grad_kernel(int c, int n, int m,
                                                                 const float *__restrict__ grad_out,
                                                                 const int *__restrict__ idx,
                                                                 float *__restrict__ grad_points) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const size_t row = (size_t)blockIdx.z * c + blockIdx.y;
  const int a = idx[(size_t)blockIdx.z * m + j];
  atomicAdd(grad_points + row * n + a, grad_out[row * m + j]);
}

}  // namespace pwclo

using namespace pwclo;

extern "C" void group_points_grad_kernel_wrapper(int b, int c, int n, int npoints, int nsample,
                                                 const float *grad_out, const int *idx, float *grad_points);

// Launch of the cooperative sampler (both orders).  All G workgroups of a cloud must be resident together:
// hipLaunchCooperativeKernel makes the runtime guarantee exactly that (the launch is REJECTED if the grid cannot be
// co-resident, and the grid is dispatched as a whole even when other streams hold CUs); a stream under graph capture
// cannot take a cooperative launch, there the plain launch with the 224-workgroup cap is used and the bounded
// spins + error word are the safety net.  The cooperative API runs on the device's one cooperative queue, so two such
// launches on different streams SERIALISE; a caller that keeps two large-cloud batches in flight (each 128 of the 256
// CUs: bench.py --config 5) selects the plain launch with pwclo_fps_large_cloud_launch(0) (or PWCLO_FPS_COOP_LAUNCH=0):
// co-residency then holds by construction as long as at most 256 workgroups of this kernel are in flight and the other
// kernels on the device are short -- and a violation still ends in PWCLO_ECOOP_TIMEOUT, never in wrong indices.
static std::atomic<int> g_xcd_local{-1};    // -1: not decided (PWCLO_FPS_COOP_XCD_LOCAL, default 1); set by pwclo_fps_large_cloud_exchange
static std::atomic<int> g_coop_api{-1};     // -1: not decided (PWCLO_FPS_COOP_LAUNCH, default 1); set by pwclo_fps_large_cloud_launch

static void coop_launch(int b, int n, int m, int bs, int log2bs, int G, const float *dataset,
                        unsigned long long *ws, int *idxs, float *new_xyz, const int *perm,
                        const float *orig_dataset) {
  hipStream_t st = current_stream();
  unsigned *host_err = device_error_word();            // a timeout inside the kernel reaches pwclo_last_error()
  if (host_err == nullptr) return;
  const char *dbg = getenv("PWCLO_FPS_COOP_DEBUG_TIMEOUT");   // test hook of the failure path (read per call)
  int holdback = dbg ? atoi(dbg) : 0;
  int spin_limit = holdback == 1 ? 256 : (1 << 21);
  int xcd_local = g_xcd_local.load();
  if (xcd_local < 0) {
    const char *e = getenv("PWCLO_FPS_COOP_XCD_LOCAL");
    xcd_local = e ? atoi(e) : 1;
    g_xcd_local.store(xcd_local);
  }
  static int poll_delay = -1;
  if (poll_delay < 0) { const char *e = getenv("PWCLO_FPS_COOP_POLL_DELAY"); poll_delay = e ? atoi(e) : 12; }   // 12 x 64 cycles: swept 0..32 on configs[4] (tools/scratch/poll_delay.sh)
  hipLaunchKernelGGL(fps_coop_init_kernel, dim3(ceil_div(b * COOP_WS_WORDS, 256)), dim3(256), 0, st, ws,
                     b * COOP_WS_WORDS);
  int coop_api = g_coop_api.load();
  if (coop_api < 0) {
    const char *e = getenv("PWCLO_FPS_COOP_LAUNCH");
    coop_api = e ? atoi(e) : 1;
    g_coop_api.store(coop_api);
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap);
  const bool use_coop_api = coop_api && cap == hipStreamCaptureStatusNone;
  const bool sorted = perm != nullptr;
  const size_t lds = sorted ? (size_t)16 * COOP_T * sizeof(unsigned) : 0;
  const void *kern = sorted ? reinterpret_cast<const void *>(fps_coop_kernel<16, true>)
                            : reinterpret_cast<const void *>(fps_coop_kernel<16, false>);
  static bool lds_attr = false;
  if (sorted && !lds_attr) {
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_attr = true;
  }
  int per_launch = (224 * (1024 / COOP_T)) / G;         // plain launch: workgroups that are certainly co-resident on an idle device
  if (per_launch >= 8) per_launch &= ~7;                // whole groups of 8 clouds: the grid is padded to 8 * G * ceil(clouds / 8)
  for (int c0 = 0; c0 < b; c0 += per_launch) {
    const int nb = min(per_launch, b - c0);
    const float *d0 = dataset + (size_t)c0 * n * 3;
    unsigned long long *w0 = ws + (size_t)c0 * COOP_WS_WORDS;
    int *i0 = idxs + (size_t)c0 * m;
    float *x0 = new_xyz ? new_xyz + (size_t)c0 * m * 3 : nullptr;
    const int *p0 = sorted ? perm + (size_t)c0 * n : nullptr;
    const float *o0 = sorted ? orig_dataset + (size_t)c0 * n * 3 : nullptr;
    const dim3 grid(8 * G * ceil_div(nb, 8));            // workgroups of a cloud share blockIdx.x % 8 (see the kernel)
    if (use_coop_api) {
      int nv = n, mv = m, bsv = bs, lbv = log2bs, Gv = G, nbv = nb, xlv = xcd_local;
      void *args[] = {&nv, &mv, &bsv, &lbv, &Gv, &nbv, &xlv, &d0, &w0, &i0, &x0, &spin_limit, &holdback, &poll_delay, &host_err, &p0, &o0};
      hipError_t e = hipLaunchCooperativeKernel(kern, grid, dim3(COOP_T), args, (unsigned)lds, st);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error((int)e, "furthest_point_sampling(coop): cooperative launch of %d x %d workgroups rejected: %s",
                  G, nb, hipGetErrorString(e));
        return;
      }
    } else if (sorted) {
      hipLaunchKernelGGL((fps_coop_kernel<16, true>), grid, dim3(COOP_T), lds, st, n, m, bs, log2bs, G, nb, xcd_local, d0, w0,
                         i0, x0, spin_limit, holdback, poll_delay, host_err, p0, o0);
    } else {
      hipLaunchKernelGGL((fps_coop_kernel<16, false>), grid, dim3(COOP_T), 0, st, n, m, bs, log2bs, G, nb, xcd_local, d0, w0,
                         i0, x0, spin_limit, holdback, poll_delay, host_err, p0, o0);
    }
  }
  check_launch("furthest_point_sampling(coop)");
}

static void fps_dispatch(int b, int n, int m, const float *dataset, float *temp, int *idxs,
                         float *new_xyz, int *tie_out = nullptr, int tie_iters = 0,
                         const int *prefix_in = nullptr) {
  if (b <= 0 || m <= 0) return;
  PWCLO_REQUIRE(n >= 1, "furthest_point_sampling: n=%d must be >= 1", n);
  PWCLO_REQUIRE((long long)n < (1ll << PRI_SHIFT) * 1ll, "furthest_point_sampling: n=%d too large", n);
  const int bs = ref_opt_n_threads(n);
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  int T = fps_pick_threads(n, bs);
  int E = 0;
  while ((T << E) < bs) ++E;                  // T < bs: a thread owns 2^E residues
  int I = ceil_div(n, E == 0 ? T : bs);       // points per residue per thread
  if (T == 1024 && I > 16) { T = 512; E = 0; I = ceil_div(n, 512); }   // VGPR budget at 16 waves
#define FPS_CASE(TT, EE, II)                                               \
  if (T == TT && E == EE && I <= II) {                                     \
    launch_fps_reg<TT, EE, II>(b, n, m, bs, log2bs, dataset, idxs, new_xyz, tie_out, tie_iters, prefix_in); \
    check_launch("furthest_point_sampling");                              \
    return;                                                                \
  }
  if (T == 1024) { FPS_CASE(1024, 0, 8) FPS_CASE(1024, 0, 16) }
  if (T == 512) { FPS_CASE(512, 0, 1) FPS_CASE(512, 0, 2) FPS_CASE(512, 0, 4) FPS_CASE(512, 0, 8)
                  FPS_CASE(512, 0, 16) FPS_CASE(512, 0, 48) }
  if (T == 256) { FPS_CASE(256, 0, 1) FPS_CASE(256, 0, 2)
                  FPS_CASE(256, 1, 1) FPS_CASE(256, 1, 2) FPS_CASE(256, 1, 4) FPS_CASE(256, 1, 8)
                  FPS_CASE(256, 1, 16) FPS_CASE(256, 1, 32) }
  if (T == 128) { FPS_CASE(128, 0, 1) FPS_CASE(128, 0, 2)
                  FPS_CASE(128, 1, 1) FPS_CASE(128, 1, 2)
                  FPS_CASE(128, 2, 1) FPS_CASE(128, 2, 2) FPS_CASE(128, 2, 4) FPS_CASE(128, 2, 8)
                  FPS_CASE(128, 2, 16) }
  if (T == 64) { FPS_CASE(64, 0, 1) FPS_CASE(64, 0, 2)
                 FPS_CASE(64, 1, 1) FPS_CASE(64, 1, 2)
                 FPS_CASE(64, 2, 1) FPS_CASE(64, 2, 2)
                 FPS_CASE(64, 3, 1) FPS_CASE(64, 3, 2) FPS_CASE(64, 3, 4) FPS_CASE(64, 3, 8) }
#undef FPS_CASE
  if (n <= 24576) {  // register-resident fallback (large clouds, or a forced T without a case)
    launch_fps_reg<512, 0, 48>(b, n, m, bs, log2bs, dataset, idxs, new_xyz, tie_out, tie_iters, prefix_in);
    check_launch("furthest_point_sampling");
    return;
  }
  PWCLO_REQUIRE(temp != nullptr,
                "furthest_point_sampling: n=%d needs the (b,n) temp buffer pre-filled with 1e10", n);
  if (tie_out != nullptr)   // the large-cloud samplers keep no tie record: later levels run in full
    hipLaunchKernelGGL(fps_fill_flag_kernel, dim3(ceil_div(b, 256)), dim3(256), 0, current_stream(), tie_out, b, 1);
  static int coop = -1;
  if (coop < 0) { const char *e = getenv("PWCLO_FPS_COOP"); coop = e ? atoi(e) : 1; }
  const int G = ceil_div(n, COOP_T * 16);                // workgroups per cloud, <= 16 points per thread
  // (8 points per thread on twice the workgroups was measured slower: 3.2 vs 2.9 us per iteration at n = 120k)
  if (coop && G <= COOP_MAX_G && (reinterpret_cast<uintptr_t>(temp) & 7) == 0 && (size_t)COOP_WS_WORDS * 2 <= (size_t)n) {
    // cooperative multi-workgroup sampler; `temp` doubles as its (re-zeroed) exchange workspace
    coop_launch(b, n, m, bs, log2bs, G, dataset, reinterpret_cast<unsigned long long *>(temp), idxs, new_xyz,
                nullptr, nullptr);
    return;
  }
  hipLaunchKernelGGL((fps_stream_kernel<1024>), dim3(b), dim3(1024), 0, current_stream(), n, m, bs,
                     log2bs, dataset, temp, idxs, new_xyz);
  check_launch("furthest_point_sampling(stream)");
}

extern "C" void furthest_point_sampling_kernel_wrapper(int b, int n, int m, const float *dataset,
                                                       float *temp, int *idxs) {
  fps_dispatch(b, n, m, dataset, temp, idxs, nullptr);
}

extern "C" void furthest_point_sampling_xyz_kernel_wrapper(int b, int n, int m, const float *dataset,
                                                           float *temp, int *idxs, float *new_xyz) {
  fps_dispatch(b, n, m, dataset, temp, idxs, new_xyz);
}

extern "C" void furthest_point_sampling_chain_kernel_wrapper(int b, int n, int m, const float *dataset,
                                                             float *temp, int *idxs, float *new_xyz,
                                                             int *tie_out, int tie_iters, const int *prefix_in) {
  fps_dispatch(b, n, m, dataset, temp, idxs, new_xyz, tie_out, tie_iters, prefix_in);
}

// Large clouds (n > 24576) in a spatially coherent order: `sorted` (b,n,3) = `dataset` gathered by `perm` (b,n), perm[p] =
// original index of sorted position p; inside every block of 1024 consecutive positions the positions must be
// ordered by ascending sampling priority (pointnet2_ops/_ext.py builds both with torch sorts).  Same indices
// (ORIGINAL numbering) and coordinates as furthest_point_sampling_xyz_kernel_wrapper, bit for bit; the distance
// update of a wave is skipped whenever the new sample cannot lower any running distance in its cell.
extern "C" void furthest_point_sampling_sorted_kernel_wrapper(int b, int n, int m, const float *dataset,
                                                              const float *sorted, const int *perm, float *temp,
                                                              int *idxs, float *new_xyz) {
  if (b <= 0 || m <= 0) return;
  PWCLO_REQUIRE(n > 24576 && (long long)n < (1ll << PRI_SHIFT), "furthest_point_sampling(sorted): n=%d outside (24576, 2^23)", n);
  PWCLO_REQUIRE(dataset != nullptr && sorted != nullptr && perm != nullptr && temp != nullptr,
                "furthest_point_sampling(sorted): dataset, sorted copy, permutation and temp are required");
  const int bs = ref_opt_n_threads(n);
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  const int G = ceil_div(n, COOP_T * 16);
  PWCLO_REQUIRE(G <= COOP_MAX_G && (reinterpret_cast<uintptr_t>(temp) & 7) == 0 && (size_t)COOP_WS_WORDS * 2 <= (size_t)n,
                "furthest_point_sampling(sorted): n=%d needs %d workgroups per cloud (max %d) / an 8-byte aligned temp", n, G,
                COOP_MAX_G);
  coop_launch(b, n, m, bs, log2bs, G, sorted, reinterpret_cast<unsigned long long *>(temp), idxs, new_xyz, perm, dataset);
}

// ---- spatial order for the large-cloud sampler, on the device (round 3) -----------------------------------------------
// The order furthest_point_sampling_sorted_kernel_wrapper wants -- spatially compact cells of 1024 consecutive positions,
// each ordered by ascending sampling priority -- used to be built with torch sorts (two argsorts + a scatter: library
// kernels on the product path of BASELINE configs[4]).  Hand-written replacement, five small kernels:
//   bounding box per cloud -> Morton code of every point's 32 x 32 x 32 cell + histogram of the 32768 cells (global atomics)
//   -> exclusive scan of the histogram (one workgroup per cloud) -> counting-sort scatter (atomic cursor per cell: the order
//   INSIDE a cell is whatever the atomics give, which is fine -- any order is exact, see below) -> per block of 1024
//   consecutive positions a bitonic sort by priority in LDS, writing `perm` and the gathered coordinates.
// Exactness does not depend on WHICH order comes out: the sampler's pruning test is exact for any partition into cells and
// the per-cell priority order keeps the reference's tie rule; a better order only prunes more.
namespace pwclo {
constexpr int SO_BITS = 18;                 // cell-code bits (handed out to the axes by the cloud's aspect ratio)
constexpr int SO_CELLS = 1 << SO_BITS;
constexpr int SO_BINS = SO_CELLS + 1024;     // + one bin (and padding to a multiple of 1024) for the never-eligible rows

// order-preserving map float -> unsigned (so that atomicMin / atomicMax on the code give the float min / max)
__device__ __forceinline__ unsigned so_code(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float so_decode(unsigned c) {
  return __uint_as_float((c & 0x80000000u) ? (c & 0x7FFFFFFFu) : ~c);
}

// box (b,8) as CODES: [0..2] min per axis (initialised 0xFFFFFFFF), [4..6] max per axis (initialised 0)
__global__ __launch_bounds__(256) void so_bbox_kernel(int n, const float *__restrict__ pts, unsigned *__restrict__ box) {
  const int b = blockIdx.y;
  const float *p = pts + (size_t)b * n * 3;
  unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float px = p[(size_t)i * 3], py = p[(size_t)i * 3 + 1], pz = p[(size_t)i * 3 + 2];
    if ((double)((px * px) + (py * py) + (pz * pz)) <= 1e-3) continue;     // never-eligible rows stay outside the box
    const float v[3] = {px, py, pz};
#pragma unroll
    for (int a = 0; a < 3; ++a) { const unsigned c = so_code(v[a]); lo[a] = min(lo[a], c); hi[a] = max(hi[a], c); }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    lo[a] = wave_allreduce_u32(lo[a], [](unsigned x, unsigned y) { return x < y ? x : y; });
    hi[a] = wave_allreduce_u32(hi[a], [](unsigned x, unsigned y) { return x > y ? x : y; });
  }
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int a = 0; a < 3; ++a) { atomicMin(box + b * 8 + a, lo[a]); atomicMax(box + b * 8 + 4 + a, hi[a]); }
}

// boxes to (min = +max code, max = 0) and the histograms to 0 in one launch (no hipMemsetAsync: the runtime's fill path
// serialised the two sampler streams of configs[4] against each other)
__global__ void so_init_kernel(unsigned *box, int nbox, int *hist, int nhist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nbox) box[i] = (i & 4) ? 0u : 0xFFFFFFFFu;
  for (int k = i; k < nhist; k += gridDim.x * blockDim.x) hist[k] = 0;
}


__global__ __launch_bounds__(256) void so_hist_kernel(int n, const float *__restrict__ pts, const unsigned *__restrict__ box,
                                                      int *__restrict__ cell, int *__restrict__ hist) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float *p = pts + ((size_t)b * n + i) * 3;
  const unsigned *bx = box + b * 8;
  // SO_BITS bits of cell code, handed out coarse to fine to whichever axis currently has the LARGEST cell (lidar clouds are
  // 160 m x 160 m x 4 m slabs: equal bits per axis would spend a third of them slicing the slab into 15 cm layers); the
  // order of the hand-outs is the bit order of the code, i.e. a Morton curve on the cloud's own aspect ratio
  float lo[3], size[3];
  int bits[3] = {0, 0, 0};
#pragma unroll
  for (int a = 0; a < 3; ++a) { lo[a] = so_decode(bx[a]); size[a] = fmaxf(so_decode(bx[4 + a]) - lo[a], 1e-20f); }
  const float span[3] = {size[0], size[1], size[2]};
  unsigned long long seq = 0ull;                 // 2 bits per hand-out, first hand-out in the top bits
#pragma unroll
  for (int t = 0; t < SO_BITS; ++t) {
    const int a = (size[0] >= size[1] && size[0] >= size[2]) ? 0 : (size[1] >= size[2] ? 1 : 2);
    seq = (seq << 2) | (unsigned long long)a;
    size[a] *= 0.5f;
    bits[a] += 1;
  }
  int q[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int cells = 1 << bits[a];
    int v = (int)((p[a] - lo[a]) * ((float)cells / span[a]));
    q[a] = v < 0 ? 0 : (v >= cells ? cells - 1 : v);
  }
  unsigned c = 0u;
  int left[3] = {bits[0], bits[1], bits[2]};
#pragma unroll
  for (int t = SO_BITS - 1; t >= 0; --t) {
    const int a = (int)((seq >> (2 * t)) & 3ull);
    left[a] -= 1;
    c = (c << 1) | (unsigned)((q[a] >> left[a]) & 1);
  }
  // rows the sampler never selects (|p|^2 <= 1e-3: the zero padding behind a frame's survivors, sampling.cpp:74-76) go
  // behind every real cell: mixed into the origin's cell they would smear its few real points over dozens of blocks
  const float mag = (p[0] * p[0]) + (p[1] * p[1]) + (p[2] * p[2]);
  if ((double)mag <= 1e-3) c = (unsigned)SO_CELLS;
  cell[(size_t)b * n + i] = (int)c;
  atomicAdd(hist + (size_t)b * SO_BINS + c, 1);
}

// hist (b, SO_CELLS) -> exclusive scan in place (the counting sort's cursors); one 1024-thread workgroup per cloud
__global__ __launch_bounds__(1024) void so_scan_kernel(int *__restrict__ hist) {
  __shared__ int wsum[16];
  int *h = hist + (size_t)blockIdx.x * SO_BINS;
  constexpr int PER = SO_BINS / 1024;
  int v[PER], tot = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) { v[k] = h[threadIdx.x * PER + k]; tot += v[k]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = inc - tot;
  for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
  for (int k = 0; k < PER; ++k) { h[threadIdx.x * PER + k] = base; base += v[k]; }
}

__global__ __launch_bounds__(256) void so_scatter_kernel(int n, const int *__restrict__ cell, int *__restrict__ cursor,
                                                         int *__restrict__ order) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = cell[(size_t)b * n + i];
  const int pos = atomicAdd(cursor + (size_t)b * SO_BINS + c, 1);
  order[(size_t)b * n + pos] = i;
}

// positions [1024 * blockIdx.x, +1024) of cloud blockIdx.y: sorted by ascending sampling priority of their original index
__global__ __launch_bounds__(1024) void so_block_sort_kernel(int n, int bs, int log2bs, const int *__restrict__ order,
                                                             const float *__restrict__ pts, float *__restrict__ sorted,
                                                             int *__restrict__ perm) {
  __shared__ unsigned long long key[1024];
  const int b = blockIdx.y, t = threadIdx.x, pos = blockIdx.x * 1024 + t;
  unsigned long long k = ~0ull;
  if (pos < n) {
    const unsigned o = (unsigned)order[(size_t)b * n + pos];
    const unsigned pri = (fps_bitrev(o & (unsigned)(bs - 1), log2bs) << PRI_SHIFT) | (o >> log2bs);
    k = ((unsigned long long)pri << 32) | o;
  }
  key[t] = k;
  __syncthreads();
  for (int size = 2; size <= 1024; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int partner = t ^ stride;
      if (partner > t) {
        const unsigned long long a = key[t], c = key[partner];
        const bool up = (t & size) == 0;
        if ((a > c) == up) { key[t] = c; key[partner] = a; }
      }
      __syncthreads();
    }
  if (pos < n) {
    const unsigned o = (unsigned)(key[t] & 0xFFFFFFFFull);
    perm[(size_t)b * n + pos] = (int)o;
    const float *src = pts + ((size_t)b * n + o) * 3;
    float *dst = sorted + ((size_t)b * n + pos) * 3;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
  }
}
}  // namespace pwclo

extern "C" long long fps_spatial_order_workspace_bytes(int b, int n) {
  return ((long long)b * 8 + (long long)b * n * 2 + (long long)b * pwclo::SO_BINS) * 4;     // box, cell, order, histogram
}

// points (b,n,3) -> sorted (b,n,3), perm (b,n) i32 as furthest_point_sampling_sorted_kernel_wrapper wants them; `workspace`
// of fps_spatial_order_workspace_bytes(b, n) bytes.  Five launches on the current stream, no host synchronisation.
extern "C" void fps_spatial_order_kernel_wrapper(int b, int n, const float *points, float *sorted, int *perm,
                                                 void *workspace) {
  if (b <= 0 || n <= 0) return;
  PWCLO_REQUIRE(workspace != nullptr && points != nullptr && sorted != nullptr && perm != nullptr,
                "fps_spatial_order: points, outputs and workspace are required%s", "");
  PWCLO_REQUIRE(b <= 65535 && (long long)n < (1ll << PRI_SHIFT), "fps_spatial_order: b=%d n=%d out of range", b, n);
  const int bs = ref_opt_n_threads(n);
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  unsigned *box = reinterpret_cast<unsigned *>(workspace);
  int *cell = reinterpret_cast<int *>(box + (size_t)b * 8);
  int *order = cell + (size_t)b * n;
  int *hist = order + (size_t)b * n;
  hipStream_t st = current_stream();
  hipLaunchKernelGGL(so_init_kernel, dim3(max(ceil_div(b * 8, 256), min(1024, ceil_div(b * SO_BINS, 1024)))), dim3(256), 0, st, box,
                     b * 8, hist, b * SO_BINS);
  hipLaunchKernelGGL(so_bbox_kernel, dim3(32, b), dim3(256), 0, st, n, points, box);
  hipLaunchKernelGGL(so_hist_kernel, dim3(ceil_div(n, 256), b), dim3(256), 0, st, n, points, box, cell, hist);
  hipLaunchKernelGGL(so_scan_kernel, dim3(b), dim3(1024), 0, st, hist);
  hipLaunchKernelGGL(so_scatter_kernel, dim3(ceil_div(n, 256), b), dim3(256), 0, st, n, cell, hist, order);
  hipLaunchKernelGGL(so_block_sort_kernel, dim3(ceil_div(n, 1024), b), dim3(1024), 0, st, n, bs, log2bs, order, points, sorted,
                     perm);
  check_launch("fps_spatial_order");
}

extern "C" int knn_point_slabs(int n);
extern "C" long long knn_point_build_bytes(int b, int n);

// Level-1 sampler of the fused pipeline: `knn_workspace` / `slab_tab` come from knn_build_kernel_wrapper on the
// same cloud (the neighbour search of that level needs the build anyway).  Clouds whose slabs fit are sampled by
// fps_slab_kernel (exact pruning of the distance update), the others by the register-resident kernel, which
// runs second and skips the clouds flagged in `status` (b ints, device).  Same outputs as the chain wrapper.
extern "C" void furthest_point_sampling_slab_kernel_wrapper(int b, int n, int m, const float *dataset, int *idxs,
                                                            float *new_xyz, int *tie_out, int tie_iters,
                                                            const void *knn_workspace, const int *slab_tab,
                                                            int *status) {
  if (b <= 0 || m <= 0) return;
  PWCLO_REQUIRE(knn_workspace != nullptr && slab_tab != nullptr && status != nullptr,
                "furthest_point_sampling(slab): workspace, slab table and status buffer are required");
  const int bs = ref_opt_n_threads(n);
  int log2bs = 0;
  while ((1 << log2bs) < bs) ++log2bs;
  const size_t chain_bytes = tie_out ? (size_t)(tie_iters + 2) * 8 : 0;
  const size_t lds = FPS_SLOT_BYTES + (size_t)n * sizeof(float4) + chain_bytes;
  PWCLO_REQUIRE(knn_point_slabs(n) == 8 && n >= 4096 && n <= 8 * 18 * 64 && lds <= 160 * 1024,
                "furthest_point_sampling(slab): n=%d is outside the slab sampler's range (8 slabs, LDS table)", n);
  const int nblk = (n + 63) / 64 + 8;
  const int per = (n + 7) / 8;
  const char *dbg_e = getenv("PWCLO_FPS_SLAB_DBG");
  const int dbg = dbg_e ? atoi(dbg_e) : 0;
  static int lds_table = -1;
  if (lds_table < 0) { const char *e = getenv("PWCLO_FPS_SLAB_TABLE"); lds_table = e ? atoi(e) : 1; }
  if (!lds_table) {        // winner coordinates from the original cloud in global memory: a few KiB of LDS per workgroup
    const size_t small = FPS_SLOT_BYTES + chain_bytes;
    if (per <= 16 * 64)
      hipLaunchKernelGGL((fps_slab_kernel<512, 16, false>), dim3(b), dim3(512), small, current_stream(), n, m, bs, log2bs,
                         nblk, reinterpret_cast<const float4 *>(knn_workspace), slab_tab, idxs, new_xyz, tie_out, tie_iters,
                         status, dbg, dataset);
    else
      hipLaunchKernelGGL((fps_slab_kernel<512, 18, false>), dim3(b), dim3(512), small, current_stream(), n, m, bs, log2bs,
                         nblk, reinterpret_cast<const float4 *>(knn_workspace), slab_tab, idxs, new_xyz, tie_out, tie_iters,
                         status, dbg, dataset);
    check_launch("furthest_point_sampling(slab)");
    return;
  }
  static bool big16 = false, big18 = false;
#define SLAB_LAUNCH(II, FLAG)                                                                                   \
  {                                                                                                             \
    auto kern = fps_slab_kernel<512, II>;                                                                       \
    if (!FLAG) {                                                                                                \
      (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);    \
      FLAG = true;                                                                                              \
    }                                                                                                           \
    hipLaunchKernelGGL(kern, dim3(b), dim3(512), lds, current_stream(), n, m, bs, log2bs, nblk,                 \
                       reinterpret_cast<const float4 *>(knn_workspace), slab_tab, idxs, new_xyz, tie_out,       \
                       tie_iters, status, dbg, dataset);                                                        \
  }
  if (per <= 16 * 64) SLAB_LAUNCH(16, big16) else SLAB_LAUNCH(18, big18)
#undef SLAB_LAUNCH
  check_launch("furthest_point_sampling(slab)");
}

extern "C" void gather_points_kernel_wrapper(int b, int c, int n, int npoints, const float *points,
                                             const int *idx, float *out) {
  if (b <= 0 || c <= 0 || npoints <= 0) return;
  PWCLO_REQUIRE(c <= 65535 && b <= 65535, "gather_points: b=%d c=%d exceed the grid limits", b, c);
  hipLaunchKernelGGL(gather_points_kernel, dim3(ceil_div(npoints, 256), c, b), dim3(256), 0,
                     current_stream(), c, n, npoints, points, idx, out);
  check_launch("gather_points");
}

extern "C" void gather_points_grad_kernel_wrapper(int b, int c, int n, int npoints,
                                                  const float *grad_out, const int *idx,
                                                  float *grad_points) {
  if (b <= 0 || c <= 0 || npoints <= 0) return;
  PWCLO_REQUIRE(c <= 65535 && b <= 65535, "gather_points_grad: b=%d c=%d exceed the grid limits", b, c);
  if ((long long)n * 4 <= 128 * 1024) {   // same scatter-add as group_points_grad with one sample per centre
    group_points_grad_kernel_wrapper(b, c, n, npoints, 1, grad_out, idx, grad_points);
    return;
  }
  hipLaunchKernelGGL(gather_points_grad_kernel, dim3(ceil_div(npoints, 256), c, b), dim3(256), 0,
                     current_stream(), c, n, npoints, grad_out, idx, grad_points);
  check_launch("gather_points_grad");
}

extern "C" void pwclo_fps_large_cloud_launch(int cooperative) { g_coop_api.store(cooperative ? 1 : 0); }

extern "C" void pwclo_fps_large_cloud_exchange(int xcd_local) { g_xcd_local.store(xcd_local ? 1 : 0); }
