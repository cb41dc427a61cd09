/*
 * oracle/pointnet2_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's point-cloud
 * operators, used only as the parity checker (tests/, __graft_entry__.smoke(),
 * bench.py's cpu_baseline leg).  The product path (pwclonet_pylidarslam_amd/)
 * never imports, links or executes anything in this directory.
 *
 * Arithmetic contract: IEEE-754 binary32, every product and sum rounded
 * individually, evaluated in the source order of the reference expressions
 * (build with -ffp-contract=off, no -ffast-math).  Citations are into
 * /root/reference/slam/models/Pointnet2_PyTorch/pointnet2_ops_lib/pointnet2_ops/
 * ("P2/").
 *
 * Pinning: the reference has no tests / golden vectors for this path
 * (SURVEY.md section 4).  The functions that restate Python layers (knn_point)
 * are pinned by fixtures generated from the imported reference
 * (oracle/gen_golden.py -> tests/golden/).  The nine extension kernels are CUDA
 * and cannot run here: for them this file follows the .cu text and is pinned
 * by hand-derived known-answer cases (tests/test_oracle_cpu.py) only --
 * "parity unpinned" by any reference execution.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* P2/_ext-src/include/cuda_utils.h:13-19 : block size = 2^floor(log2(work)),
 * clamped to [1, 512]; the reference evaluates log(x)/log(2) in double and
 * truncates, which is what is done here. */
int oracle_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int t = 1 << pow_2;
  if (t > 512) t = 512;
  if (t < 1) t = 1;
  return t;
}

/* P2/_ext-src/src/sampling_gpu.cu:69-173 (kernel), sampling.cpp:66-87 (host:
 * temp pre-filled with 1e10, idxs zero-initialised).
 *
 * The CUDA kernel runs `bs = opt_n_threads(n)` threads per cloud.  Thread t
 * walks points k = t, t+bs, ... keeping (best, besti) with a strict `>`;
 * then a shared-memory tree keeps slot t over slot t+s on ties.  Both stages
 * are emulated literally because together they define which index wins among
 * exactly-equal maxima.  Points with x*x+y*y+z*z <= 1e-3 are skipped (the
 * comparison is against the double constant 1e-3, :100-101).
 */
void oracle_furthest_point_sampling(int b, int n, int m, const float *dataset,
                                    float *temp, int *idxs) {
  if (m <= 0) return;
  const int bs = oracle_opt_n_threads(n);
  /* clouds are independent: one host thread per cloud (same results as the serial loop) */
#pragma omp parallel for schedule(dynamic, 1)
  for (int bi = 0; bi < b; ++bi) {
    float *dists = (float *)malloc(sizeof(float) * (size_t)bs);
    int *dists_i = (int *)malloc(sizeof(int) * (size_t)bs);
    const float *pts = dataset + (size_t)bi * n * 3;
    float *tmp = temp + (size_t)bi * n;
    int *out = idxs + (size_t)bi * m;
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = pts[old * 3 + 0];
      const float y1 = pts[old * 3 + 1];
      const float z1 = pts[old * 3 + 2];
      for (int t = 0; t < bs; ++t) {
        int besti = 0;
        float best = -1.0f;
        for (int k = t; k < n; k += bs) {
          const float x2 = pts[k * 3 + 0];
          const float y2 = pts[k * 3 + 1];
          const float z2 = pts[k * 3 + 2];
          const float mag = (x2 * x2) + (y2 * y2) + (z2 * z2);
          if ((double)mag <= 1e-3) continue;
          const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
          const float d = dx * dx + dy * dy + dz * dz;
          const float d2 = d < tmp[k] ? d : tmp[k]; /* min(d, temp[k]) */
          tmp[k] = d2;
          if (d2 > best) {
            besti = k;
            best = d2;
          }
        }
        dists[t] = best;
        dists_i[t] = besti;
      }
      for (int s = bs / 2; s >= 1; s >>= 1) {
        for (int t = 0; t < s; ++t) {
          const float v1 = dists[t], v2 = dists[t + s];
          const int i1 = dists_i[t], i2 = dists_i[t + s];
          dists[t] = v1 > v2 ? v1 : v2;
          dists_i[t] = v2 > v1 ? i2 : i1;
        }
      }
      old = dists_i[0];
      out[j] = old;
    }
    free(dists);
    free(dists_i);
  }
}

/* P2/_ext-src/src/sampling_gpu.cu:8-20 : out[b,c,j] = points[b,c,idx[b,j]] */
void oracle_gather_points(int b, int c, int n, int m, const float *points,
                          const int *idx, float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j)
        out[((size_t)i * c + l) * m + j] =
            points[((size_t)i * c + l) * n + idx[(size_t)i * m + j]];
}

/* P2/_ext-src/src/sampling_gpu.cu:34-47 : scatter-add into zero-initialised
 * grad_points (sampling.cpp:51-53).  The GPU uses fp32 atomics in an
 * unspecified order; the oracle accumulates in double and rounds once, which
 * every fp32 ordering agrees with to within summation error. */
void oracle_gather_points_grad(int b, int c, int n, int m,
                               const float *grad_out, const int *idx,
                               float *grad_points) {
  double *acc = (double *)calloc((size_t)n, sizeof(double));
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l) {
      memset(acc, 0, sizeof(double) * (size_t)n);
      for (int j = 0; j < m; ++j)
        acc[idx[(size_t)i * m + j]] += grad_out[((size_t)i * c + l) * m + j];
      for (int k = 0; k < n; ++k)
        grad_points[((size_t)i * c + l) * n + k] = (float)acc[k];
    }
  free(acc);
}

/* P2/_ext-src/src/group_points_gpu.cu:8-28 :
 * out[b,c,j,k] = points[b,c,idx[b,j,k]] */
void oracle_group_points(int b, int c, int n, int npoints, int nsample,
                         const float *points, const int *idx, float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k) {
          const int ii = idx[((size_t)i * npoints + j) * nsample + k];
          out[(((size_t)i * c + l) * npoints + j) * nsample + k] =
              points[((size_t)i * c + l) * n + ii];
        }
}

/* P2/_ext-src/src/group_points_gpu.cu:43-64 (atomics; see gather grad). */
void oracle_group_points_grad(int b, int c, int n, int npoints, int nsample,
                              const float *grad_out, const int *idx,
                              float *grad_points) {
  double *acc = (double *)calloc((size_t)n, sizeof(double));
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l) {
      memset(acc, 0, sizeof(double) * (size_t)n);
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          acc[idx[((size_t)i * npoints + j) * nsample + k]] +=
              grad_out[(((size_t)i * c + l) * npoints + j) * nsample + k];
      for (int k = 0; k < n; ++k)
        grad_points[((size_t)i * c + l) * n + k] = (float)acc[k];
    }
  free(acc);
}

/* P2/_ext-src/src/ball_query_gpu.cu:9-44, host ball_query.cpp:19-21 (idx is
 * zero-initialised, so a centre with no hit keeps zeros).  Scan candidates in
 * index order, keep the first nsample with d2 < r*r (strict); the first hit
 * pre-fills every slot. */
void oracle_ball_query(int b, int n, int m, float radius, int nsample,
                       const float *new_xyz, const float *xyz, int *idx) {
  const float radius2 = radius * radius;
  for (int i = 0; i < b; ++i) {
    const float *q = new_xyz + (size_t)i * m * 3;
    const float *p = xyz + (size_t)i * n * 3;
    int *o = idx + (size_t)i * m * nsample;
    for (int j = 0; j < m; ++j) {
      const float nx = q[j * 3 + 0], ny = q[j * 3 + 1], nz = q[j * 3 + 2];
      int cnt = 0;
      for (int k = 0; k < n && cnt < nsample; ++k) {
        const float dx = nx - p[k * 3 + 0];
        const float dy = ny - p[k * 3 + 1];
        const float dz = nz - p[k * 3 + 2];
        const float d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) o[j * nsample + l] = k;
          o[j * nsample + cnt] = k;
          ++cnt;
        }
      }
    }
  }
}

/* P2/_ext-src/src/interpolate_gpu.cu:9-59 : three smallest squared distances,
 * strict `<` insertion (earlier k wins ties); running bests are doubles
 * initialised to 1e40 and compared against the float distance (:27). */
void oracle_three_nn(int b, int n, int m, const float *unknown,
                     const float *known, float *dist2, int *idx) {
  for (int i = 0; i < b; ++i) {
    const float *u = unknown + (size_t)i * n * 3;
    const float *kn = known + (size_t)i * m * 3;
    for (int j = 0; j < n; ++j) {
      const float ux = u[j * 3 + 0], uy = u[j * 3 + 1], uz = u[j * 3 + 2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float dx = ux - kn[k * 3 + 0];
        const float dy = uy - kn[k * 3 + 1];
        const float dz = uz - kn[k * 3 + 2];
        const float d = dx * dx + dy * dy + dz * dz;
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      float *dj = dist2 + ((size_t)i * n + j) * 3;
      int *ij = idx + ((size_t)i * n + j) * 3;
      dj[0] = (float)best1; dj[1] = (float)best2; dj[2] = (float)best3;
      ij[0] = besti1; ij[1] = besti2; ij[2] = besti3;
    }
  }
}

/* P2/_ext-src/src/interpolate_gpu.cu:72-101 :
 * out[b,c,j] = p[i1]*w1 + p[i2]*w2 + p[i3]*w3 (left to right). */
void oracle_three_interpolate(int b, int c, int m, int n, const float *points,
                              const int *idx, const float *weight, float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l) {
      const float *p = points + ((size_t)i * c + l) * m;
      for (int j = 0; j < n; ++j) {
        const float *w = weight + ((size_t)i * n + j) * 3;
        const int *ix = idx + ((size_t)i * n + j) * 3;
        const float a = p[ix[0]] * w[0];
        const float bb = p[ix[1]] * w[1];
        const float cc = p[ix[2]] * w[2];
        out[((size_t)i * c + l) * n + j] = (a + bb) + cc;
      }
    }
}

/* P2/_ext-src/src/interpolate_gpu.cu:116-143 (atomics; see gather grad). */
void oracle_three_interpolate_grad(int b, int c, int n, int m,
                                   const float *grad_out, const int *idx,
                                   const float *weight, float *grad_points) {
  double *acc = (double *)calloc((size_t)m, sizeof(double));
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l) {
      memset(acc, 0, sizeof(double) * (size_t)m);
      for (int j = 0; j < n; ++j) {
        const float g = grad_out[((size_t)i * c + l) * n + j];
        const float *w = weight + ((size_t)i * n + j) * 3;
        const int *ix = idx + ((size_t)i * n + j) * 3;
        acc[ix[0]] += (double)(g * w[0]);
        acc[ix[1]] += (double)(g * w[1]);
        acc[ix[2]] += (double)(g * w[2]);
      }
      for (int k = 0; k < m; ++k)
        grad_points[((size_t)i * c + l) * m + k] = (float)acc[k];
    }
  free(acc);
}

/* P2/pytorch_utils.py:12-49 (knn_point over _nn_distance + torch.topk).
 * Key of candidate k for query s (verified bitwise against
 * the imported reference by oracle/gen_golden.py):
 *     d = sqrtf(((dx*dx + dy*dy) + dz*dz) + 1e-8f),  dx = query - candidate.
 * Result: the nsample smallest keys in ascending order.  torch.topk leaves the
 * order inside a group of equal keys unspecified; the oracle (and the HIP
 * kernel) break ties towards the lower candidate index.  Also returns the keys
 * so tests can do tie-aware comparisons.  Requires nsample <= n.
 */
typedef struct { float d; int i; } knn_ent;

static int knn_less(const knn_ent *a, const knn_ent *b) {
  return (a->d < b->d) || (a->d == b->d && a->i < b->i);
}

void oracle_knn_point(int b, int n, int s, int nsample, const float *xyz,
                      const float *new_xyz, int *idx, float *dist) {
  /* queries are independent: host threads share the (cloud, query) range (same results as the serial loop) */
#pragma omp parallel
  {
  knn_ent *heap = (knn_ent *)malloc(sizeof(knn_ent) * (size_t)nsample);
#pragma omp for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi) {
    for (int j = 0; j < s; ++j) {
      const float *p = xyz + (size_t)bi * n * 3;
      const float *q = new_xyz + (size_t)bi * s * 3;
      const float qx = q[j * 3 + 0], qy = q[j * 3 + 1], qz = q[j * 3 + 2];
      int cnt = 0; /* heap[0..cnt) kept sorted ascending by (d, i) */
      for (int k = 0; k < n; ++k) {
        const float dx = qx - p[k * 3 + 0];
        const float dy = qy - p[k * 3 + 1];
        const float dz = qz - p[k * 3 + 2];
        const float t = ((dx * dx + dy * dy) + dz * dz) + 1e-8f;
        knn_ent e; e.d = sqrtf(t); e.i = k;
        if (cnt == nsample && !knn_less(&e, &heap[cnt - 1])) continue;
        int pos = cnt < nsample ? cnt : nsample - 1;
        while (pos > 0 && knn_less(&e, &heap[pos - 1])) {
          heap[pos] = heap[pos - 1];
          --pos;
        }
        heap[pos] = e;
        if (cnt < nsample) ++cnt;
      }
      for (int k = 0; k < nsample; ++k) {
        idx[((size_t)bi * s + j) * nsample + k] = heap[k].i;
        if (dist) dist[((size_t)bi * s + j) * nsample + k] = heap[k].d;
      }
    }
  }
  free(heap);
  }
}

/* Host threads the parallel loops above use (1 when built without OpenMP). */
#ifdef _OPENMP
#include <omp.h>
int oracle_num_threads(void) { return omp_get_max_threads(); }
void oracle_set_num_threads(int n) { if (n >= 1) omp_set_num_threads(n); }
#else
int oracle_num_threads(void) { return 1; }
void oracle_set_num_threads(int n) { (void)n; }
#endif
