/*
 * pzn_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Single-threaded CPU restatement of the reference algorithms on the hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (libpzn.so) never does.
 *
 * Pinning status
 *   - point ops (fps / knn / ball query / group / gather / square_distance):
 *     pinned bit-exactly against outputs of the reference's own
 *     pointnet_util.py executed on CPU (tests/golden/point_ops.npz, generated
 *     by tests/golden/make_golden.py).
 *   - EMD: the reference implementation is CUDA-only (THC headers, no CPU
 *     path, PyTorchEMD/emd.py:10) and cannot be built or run here.  This
 *     restatement follows PyTorchEMD/cuda/emd_kernel.cu line by line and is
 *     pinned ONLY by the reference's commented 2-point known-answer test
 *     (PyTorchEMD/test_emd_loss.py:8-25: cost 0.71 per item).  Beyond that
 *     EMD parity is UNPINNED by the reference.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no fast-math, no FMA,
 * so fp32 results follow the written operation order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* pointnet_util.py:22-36 — sum((src[:, :, None] - dst[:, None]) ** 2, -1).
 * For a 3-vector torch's CPU reduction evaluates ((dx*dx + dy*dy) + dz*dz). */
static inline float sqdist3(const float* a, const float* b) {
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  float s = dx * dx + dy * dy;
  return s + dz * dz;
}

ORC_API void orc_square_distance_f32(const float* src, const float* dst, int B,
                                     int S, int N, float* out) {
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < S; ++s) {
      const float* q = src + ((size_t)b * S + s) * 3;
      float* o = out + ((size_t)b * S + s) * N;
      for (int j = 0; j < N; ++j) o[j] = sqdist3(q, dst + ((size_t)b * N + j) * 3);
    }
}

/* pointnet_util.py:53-73 — farthest_point_sample.  start_idx stands in for the
 * torch.randint draw at :65. */
ORC_API void orc_fps_f32(const float* xyz, int B, int N, int npoint,
                         const int64_t* start_idx, int64_t* out) {
  float* distance = (float*)malloc(sizeof(float) * (size_t)N);
  for (int b = 0; b < B; ++b) {
    const float* p = xyz + (size_t)b * N * 3;
    for (int j = 0; j < N; ++j) distance[j] = 1e10f; /* :64 */
    int64_t farthest = start_idx[b];                 /* :65 */
    for (int i = 0; i < npoint; ++i) {
      out[(size_t)b * npoint + i] = farthest;        /* :68 */
      const float* c = p + farthest * 3;             /* :69 */
      float best = -INFINITY;
      int64_t besti = 0;
      for (int j = 0; j < N; ++j) {
        float d = sqdist3(p + (size_t)j * 3, c);     /* :70 */
        if (d < distance[j]) distance[j] = d;        /* :71 torch.min */
        if (distance[j] > best) {                    /* :72 first max wins */
          best = distance[j];
          besti = j;
        }
      }
      farthest = besti;
    }
  }
  free(distance);
}

typedef struct {
  float d;
  int32_t i;
} orc_di;

static int cmp_di(const void* a, const void* b) {
  const orc_di* x = (const orc_di*)a;
  const orc_di* y = (const orc_di*)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return (x->i > y->i) - (x->i < y->i); /* stable: ties by ascending index */
}

/* pointnet_util.py:118-119 — square_distance(new_xyz, xyz).argsort()[:, :, :K] */
ORC_API void orc_knn_f32(const float* xyz, const float* new_xyz, int B, int N,
                         int S, int K, int64_t* idx) {
  orc_di* row = (orc_di*)malloc(sizeof(orc_di) * (size_t)N);
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < S; ++s) {
      const float* q = new_xyz + ((size_t)b * S + s) * 3;
      for (int j = 0; j < N; ++j) {
        row[j].d = sqdist3(q, xyz + ((size_t)b * N + j) * 3);
        row[j].i = j;
      }
      qsort(row, (size_t)N, sizeof(orc_di), cmp_di);
      for (int k = 0; k < K; ++k) idx[((size_t)b * S + s) * K + k] = row[k].i;
    }
  free(row);
}

/* pointnet_util.py:76-96 — query_ball_point. radius2 = (float)(radius**2):
 * the comparison `sqrdists > radius ** 2` is evaluated in fp32. */
ORC_API void orc_ball_query_f32(float radius2, int nsample, const float* xyz,
                                const float* new_xyz, int B, int N, int S,
                                int64_t* idx) {
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < S; ++s) {
      const float* q = new_xyz + ((size_t)b * S + s) * 3;
      int64_t* o = idx + ((size_t)b * S + s) * nsample;
      int cnt = 0;
      for (int j = 0; j < N && cnt < nsample; ++j) {
        float d = sqdist3(q, xyz + ((size_t)b * N + j) * 3);
        if (!(d > radius2)) o[cnt++] = j; /* :91 keeps d <= r^2 */
      }
      /* :92-95 — the sorted row is [hits..., N, N, ...]; slots equal to N are
       * replaced by slot 0, which is N itself when there was no hit at all. */
      int64_t first = cnt > 0 ? o[0] : (int64_t)N;
      for (int k = cnt; k < nsample; ++k) o[k] = first;
    }
}

/* pointnet_util.py:39-50 — index_points on a flattened idx[B,M]. */
ORC_API void orc_gather_fwd_f32(const float* points, const int64_t* idx, int B,
                                int N, int M, int C, float* out) {
  for (int b = 0; b < B; ++b)
    for (int m = 0; m < M; ++m) {
      int64_t j = idx[(size_t)b * M + m];
      memcpy(out + ((size_t)b * M + m) * C, points + ((size_t)b * N + j) * C,
             sizeof(float) * (size_t)C);
    }
}

ORC_API void orc_gather_bwd_f32(const float* grad_out, const int64_t* idx,
                                int B, int N, int M, int C, float* grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)B * N * C);
  for (int b = 0; b < B; ++b)
    for (int m = 0; m < M; ++m) {
      int64_t j = idx[(size_t)b * M + m];
      const float* g = grad_out + ((size_t)b * M + m) * C;
      float* o = grad_points + ((size_t)b * N + j) * C;
      for (int c = 0; c < C; ++c) o[c] += g[c];
    }
}

/* pointnet_util.py:123-132 — grouped_xyz - centre, concatenated with feats. */
ORC_API void orc_group_fwd_f32(const float* xyz, const float* feat,
                               const float* new_xyz, const int64_t* idx, int B,
                               int N, int S, int K, int D, float* out,
                               float* grouped_xyz) {
  const int W = 3 + D;
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < S; ++s) {
      const float* c = new_xyz + ((size_t)b * S + s) * 3;
      for (int k = 0; k < K; ++k) {
        size_t r = ((size_t)b * S + s) * K + k;
        int64_t j = idx[r];
        const float* p = xyz + ((size_t)b * N + j) * 3;
        float* o = out + r * W;
        o[0] = p[0] - c[0];
        o[1] = p[1] - c[1];
        o[2] = p[2] - c[2];
        if (grouped_xyz) memcpy(grouped_xyz + r * 3, p, 3 * sizeof(float));
        if (D) memcpy(o + 3, feat + ((size_t)b * N + j) * D, sizeof(float) * (size_t)D);
      }
    }
}

ORC_API void orc_group_bwd_f32(const float* grad_out, const int64_t* idx, int B,
                               int N, int S, int K, int D, float* grad_xyz,
                               float* grad_feat, float* grad_new_xyz) {
  const int W = 3 + D;
  if (grad_xyz) memset(grad_xyz, 0, sizeof(float) * (size_t)B * N * 3);
  if (grad_feat) memset(grad_feat, 0, sizeof(float) * (size_t)B * N * D);
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < S; ++s) {
      double acc[3] = {0, 0, 0};
      for (int k = 0; k < K; ++k) {
        size_t r = ((size_t)b * S + s) * K + k;
        int64_t j = idx[r];
        const float* g = grad_out + r * W;
        if (grad_xyz)
          for (int c = 0; c < 3; ++c) grad_xyz[((size_t)b * N + j) * 3 + c] += g[c];
        for (int c = 0; c < 3; ++c) acc[c] += g[c];
        if (grad_feat)
          for (int c = 0; c < D; ++c) grad_feat[((size_t)b * N + j) * D + c] += g[3 + c];
      }
      if (grad_new_xyz)
        for (int c = 0; c < 3; ++c)
          grad_new_xyz[((size_t)b * S + s) * 3 + c] = (float)(-acc[c]);
    }
}

/* ------------------------------------------------------------------------ */
/* EMD — PyTorchEMD/cuda/emd_kernel.cu                                      */
/* ------------------------------------------------------------------------ */

#define ORC_EMD_IMPL(T, SUF, EXPF, POWF, FMIN, FMAX, EPS)                        \
  /* emd_kernel.cu:25-158 approxmatch; match is (b, m, n): [i*n*m + l*n + k] */ \
  ORC_API void orc_emd_approxmatch_##SUF(const T* xyz1, const T* xyz2, int b,   \
                                         int n, int m, T* match) {              \
    T* remainL = (T*)malloc(sizeof(T) * (size_t)(n + m) * 2);                   \
    T* remainR = remainL + n;                                                   \
    T* ratioL = remainR + m;                                                    \
    T* ratioR = ratioL + n;                                                     \
    T multiL, multiR;                                                           \
    if (n >= m) { /* :29-35, integer division */                                \
      multiL = 1;                                                               \
      multiR = (T)(n / m);                                                      \
    } else {                                                                    \
      multiL = (T)(m / n);                                                      \
      multiR = 1;                                                               \
    }                                                                           \
    for (int i = 0; i < b; ++i) {                                               \
      const T* p1 = xyz1 + (size_t)i * n * 3;                                   \
      const T* p2 = xyz2 + (size_t)i * m * 3;                                   \
      T* mt = match + (size_t)i * n * m;                                        \
      for (size_t j = 0; j < (size_t)n * m; ++j) mt[j] = 0; /* :39-40 */        \
      for (int j = 0; j < n; ++j) remainL[j] = multiL;      /* :41-42 */        \
      for (int j = 0; j < m; ++j) remainR[j] = multiR;      /* :43-44 */        \
      for (int j = 7; j >= -2; --j) {                       /* :46 */           \
        T level = -(T)POWF(4.0f, (float)j);                 /* :47 */           \
        if (j == -2) level = 0;                             /* :48-50 */        \
        for (int k = 0; k < n; ++k) {                       /* :51-84 */        \
          T x1 = p1[k * 3], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];             \
          T suml = (T)EPS;                                  /* :59 */           \
          for (int l = 0; l < m; ++l) {                                         \
            T x2 = p2[l * 3], y2 = p2[l * 3 + 1], z2 = p2[l * 3 + 2];           \
            T d = level * ((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) +      \
                           (z2 - z1) * (z2 - z1));          /* :76 */           \
            suml += EXPF(d) * remainR[l];                   /* :77-78 */        \
          }                                                                     \
          ratioL[k] = remainL[k] / suml;                    /* :83 */           \
        }                                                                       \
        for (int l = 0; l < m; ++l) {                       /* :86-119 */       \
          T x2 = p2[l * 3], y2 = p2[l * 3 + 1], z2 = p2[l * 3 + 2];             \
          T sumr = 0;                                                           \
          for (int k = 0; k < n; ++k) {                                         \
            T x1 = p1[k * 3], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];           \
            sumr += EXPF(level * ((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + \
                                  (z2 - z1) * (z2 - z1))) *                     \
                    ratioL[k];                              /* :108-109 */      \
          }                                                                     \
          sumr *= remainR[l];                               /* :114 */          \
          T consumption = FMIN(remainR[l] / (sumr + (T)EPS), (T)1); /* :115 */  \
          ratioR[l] = consumption * remainR[l];             /* :116 */          \
          remainR[l] = FMAX((T)0, remainR[l] - sumr);       /* :117 */          \
        }                                                                       \
        for (int k = 0; k < n; ++k) {                       /* :121-154 */      \
          T x1 = p1[k * 3], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];             \
          T suml = 0;                                                           \
          T rl = ratioL[k];                                 /* :139 */          \
          for (int l = 0; l < m; ++l) {                                         \
            T x2 = p2[l * 3], y2 = p2[l * 3 + 1], z2 = p2[l * 3 + 2];           \
            T w = EXPF(level * ((x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + \
                                (z2 - z1) * (z2 - z1))) *                       \
                  rl * ratioR[l];                           /* :145 */          \
            mt[(size_t)l * n + k] += w;                     /* :146 */          \
            suml += w;                                      /* :147 */          \
          }                                                                     \
          remainL[k] = FMAX((T)0, remainL[k] - suml);       /* :153 */          \
        }                                                                       \
      }                                                                         \
    }                                                                           \
    free(remainL);                                                              \
  }                                                                             \
  /* emd_kernel.cu:200-243 matchcost: squared distance, no sqrt (:225-226). */  \
  /* The kernel's 512-thread strided partial sums + tree are order-only; a */   \
  /* double accumulator is used here as the rounding-free statement. */         \
  ORC_API void orc_emd_matchcost_##SUF(const T* xyz1, const T* xyz2,            \
                                       const T* match, int b, int n, int m,     \
                                       T* cost) {                               \
    for (int i = 0; i < b; ++i) {                                               \
      const T* p1 = xyz1 + (size_t)i * n * 3;                                   \
      const T* p2 = xyz2 + (size_t)i * m * 3;                                   \
      const T* mt = match + (size_t)i * n * m;                                  \
      double acc = 0;                                                           \
      for (int k = 0; k < n; ++k) {                                             \
        T x1 = p1[k * 3], y1 = p1[k * 3 + 1], z1 = p1[k * 3 + 2];               \
        for (int l = 0; l < m; ++l) {                                           \
          T x2 = p2[l * 3], y2 = p2[l * 3 + 1], z2 = p2[l * 3 + 2];             \
          T d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) +                 \
                (z2 - z1) * (z2 - z1);                      /* :225 */          \
          acc += (double)(d * mt[(size_t)l * n + k]);       /* :226 */          \
        }                                                                       \
      }                                                                         \
      cost[i] = (T)acc;                                                         \
    }                                                                           \
  }                                                                             \
  /* emd_kernel.cu:333-355 (grad1) and :286-327 (grad2). */                     \
  ORC_API void orc_emd_matchcost_grad_##SUF(                                    \
      const T* grad_cost, const T* xyz1, const T* xyz2, const T* match, int b,  \
      int n, int m, T* grad1, T* grad2) {                                       \
    for (int i = 0; i < b; ++i) {                                               \
      const T* p1 = xyz1 + (size_t)i * n * 3;                                   \
      const T* p2 = xyz2 + (size_t)i * m * 3;                                   \
      const T* mt = match + (size_t)i * n * m;                                  \
      for (int l = 0; l < n; ++l) {                         /* :336-353 */      \
        T x1 = p1[l * 3], y1 = p1[l * 3 + 1], z1 = p1[l * 3 + 2];               \
        T dx = 0, dy = 0, dz = 0;                                               \
        for (int k = 0; k < m; ++k) {                                           \
          T d = mt[(size_t)k * n + l] * 2;                  /* :345 */          \
          dx += (x1 - p2[k * 3]) * d;                                           \
          dy += (y1 - p2[k * 3 + 1]) * d;                                       \
          dz += (z1 - p2[k * 3 + 2]) * d;                                       \
        }                                                                       \
        grad1[((size_t)i * n + l) * 3 + 0] = dx * grad_cost[i];                 \
        grad1[((size_t)i * n + l) * 3 + 1] = dy * grad_cost[i];                 \
        grad1[((size_t)i * n + l) * 3 + 2] = dz * grad_cost[i];                 \
      }                                                                         \
      for (int k = 0; k < m; ++k) {                         /* :292-324 */      \
        T x2 = p2[k * 3], y2 = p2[k * 3 + 1], z2 = p2[k * 3 + 2];               \
        T sx = 0, sy = 0, sz = 0;                                               \
        for (int j = 0; j < n; ++j) {                                           \
          T d = mt[(size_t)k * n + j] * 2;                  /* :301 */          \
          sx += (x2 - p1[j * 3]) * d;                                           \
          sy += (y2 - p1[j * 3 + 1]) * d;                                       \
          sz += (z2 - p1[j * 3 + 2]) * d;                                       \
        }                                                                       \
        grad2[((size_t)i * m + k) * 3 + 0] = sx * grad_cost[i];                 \
        grad2[((size_t)i * m + k) * 3 + 1] = sy * grad_cost[i];                 \
        grad2[((size_t)i * m + k) * 3 + 2] = sz * grad_cost[i];                 \
      }                                                                         \
    }                                                                           \
  }

ORC_EMD_IMPL(float, f32, expf, powf, fminf, fmaxf, 1e-9f)
ORC_EMD_IMPL(double, f64, exp, powf, fmin, fmax, 1e-9f)

/* model5_b.py:1495-1505 chamfer_loss: P = |a_i|^2 + |b_j|^2 - 2 a_i.b_j
 * (expansion form, as the reference's three bmm calls compute it).
 * min_over_a[j] = min_i P[i][j]  (torch.min(P, 1)),
 * min_over_b[i] = min_j P[i][j]  (torch.min(P, 2)). */
ORC_API void orc_chamfer_fwd_f32(const float* a, const float* b, int B, int n,
                                 int m, float* min_over_a, int32_t* arg_over_a,
                                 float* min_over_b, int32_t* arg_over_b) {
  for (int bb = 0; bb < B; ++bb) {
    const float* pa = a + (size_t)bb * n * 3;
    const float* pb = b + (size_t)bb * m * 3;
    for (int j = 0; j < m; ++j) {
      min_over_a[(size_t)bb * m + j] = INFINITY;
      arg_over_a[(size_t)bb * m + j] = 0;
    }
    for (int i = 0; i < n; ++i) {
      float rx = pa[i * 3] * pa[i * 3] + pa[i * 3 + 1] * pa[i * 3 + 1] +
                 pa[i * 3 + 2] * pa[i * 3 + 2];
      float best = INFINITY;
      int32_t bj = 0;
      for (int j = 0; j < m; ++j) {
        float ry = pb[j * 3] * pb[j * 3] + pb[j * 3 + 1] * pb[j * 3 + 1] +
                   pb[j * 3 + 2] * pb[j * 3 + 2];
        float zz = pa[i * 3] * pb[j * 3] + pa[i * 3 + 1] * pb[j * 3 + 1] +
                   pa[i * 3 + 2] * pb[j * 3 + 2];
        float P = (rx + ry) - 2 * zz;
        if (P < best) {
          best = P;
          bj = j;
        }
        if (P < min_over_a[(size_t)bb * m + j]) {
          min_over_a[(size_t)bb * m + j] = P;
          arg_over_a[(size_t)bb * m + j] = i;
        }
      }
      min_over_b[(size_t)bb * n + i] = best;
      arg_over_b[(size_t)bb * n + i] = bj;
    }
  }
}
