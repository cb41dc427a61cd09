// ball_query for gfx950.  Replaces P2/_ext-src/src/ball_query_gpu.cu.
//
// The reference gives every centre to one thread that scans all n candidates serially.  Here a
// wave owns a centre and tests 64 candidates per step (coalesced 768-byte reads of xyz); the
// ordered "first nsample hits" semantics come from a wave ballot + prefix count, and the scan
// stops as soon as the list is full.  Distances use the reference's expression and operation
// order without contraction, so the strict `d2 < r*r` test selects the same candidates.
#include "common.hpp"

namespace pwclo {

constexpr int BQ_WAVES = 4;

__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(
    int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * BQ_WAVES + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (j >= m) return;  // wave-uniform
  const float *q = new_xyz + ((size_t)b * m + j) * 3;
  const float nx = q[0], ny = q[1], nz = q[2];
  const float *p = xyz + (size_t)b * n * 3;
  int *o = idx + ((size_t)b * m + j) * nsample;

  int cnt = 0, first = -1;
  for (int k0 = 0; k0 < n && cnt < nsample; k0 += 64) {
    const int k = k0 + lane;
    bool hit = false;
    if (k < n) {
      const float dx = nx - p[k * 3 + 0], dy = ny - p[k * 3 + 1], dz = nz - p[k * 3 + 2];
      const float d2 = dx * dx + dy * dy + dz * dz;
      hit = d2 < radius2;
    }
    const unsigned long long mask = __ballot(hit);
    if (mask != 0ull) {
      const int pos = cnt + mbcnt64(mask);
      if (hit && pos < nsample) o[pos] = k;
      if (first < 0) first = k0 + __builtin_ctzll(mask);
      cnt += __popcll(mask);
    }
  }
  if (first >= 0) {
    cnt = cnt < nsample ? cnt : nsample;
    for (int l = cnt + lane; l < nsample; l += 64) o[l] = first;  // pad with the first hit
  }
}

}  // namespace pwclo

using namespace pwclo;

extern "C" void query_ball_point_kernel_wrapper(int b, int n, int m, float radius, int nsample,
                                                const float *new_xyz, const float *xyz, int *idx) {
  if (b <= 0 || m <= 0 || nsample <= 0) return;
  PWCLO_REQUIRE(b <= 65535, "ball_query: b=%d exceeds the grid limit", b);
  const float radius2 = radius * radius;
  hipLaunchKernelGGL(ball_query_kernel, dim3(ceil_div(m, BQ_WAVES), b), dim3(BQ_WAVES * 64), 0,
                     current_stream(), n, m, radius2, nsample, new_xyz, xyz, idx);
  check_launch("ball_query");
}
