// Arithmetic of the batched primitive fits (fitbatch.hip): moment table, dual numbers, 3x3
// eigen / least-squares helpers, the four fits and the residual distances.  Plain C++ behind the
// FB_HD qualifier so that the SAME source is compiled into the gfx950 kernels and, by the test
// suite only (tests/native/), into a host harness that checks it against the oracle.
#pragma once
#include <math.h>
#include <stdint.h>
#if defined(__HIPCC__)
#define FB_HD __device__ static inline
#define FB_TABLE __device__ const
#else
#define FB_HD static inline
#define FB_TABLE static const
#endif

#define FB_NMOM 64      // moment slots per segment (60 used; = lanes of a wave)
#define FB_NPAR 16      // parameter slots per segment
#define FB_CH 4         // point chunks per segment in the moment pass
#define FB_TILE 256     // points staged per LDS tile
#define FB_EPS 1.1920928955078125e-07   // np.finfo(np.float32).eps

enum { FB_PLANE = 0, FB_SPHERE = 1, FB_CYLINDER = 2, FB_CONE = 3 };

// ---- moment table ------------------------------------------------------------------
// z = [1, px, py, pz, nx, ny, nz]; moment m = sum w^e * z[i1] * z[i2] * z[i3]
struct FbMono {
  signed char e, i1, i2, i3;
};
enum {
  M_S1 = 0, M_M1 = 1, M_N1 = 4, M_C1 = 7,          // e = 1: 1, p, n, p p^T
  M_S2 = 13, M_M2 = 14, M_N2 = 17, M_C2 = 20,      // e = 2: 1, p, n, p p^T,
  M_NN2 = 26, M_NNP = 32,                          //        n n^T, n_a n_b p_b
  M_C3 = 41, M_T3 = 47,                            // e = 3: p p^T, p p p
  M_NS = 57,                                       // e = 0: n
  M_USED = 60
};
#define PX 1
#define PY 2
#define PZ 3
#define NX 4
#define NY 5
#define NZ 6
FB_TABLE FbMono fb_table[FB_NMOM] = {
    {1, 0, 0, 0},
    {1, PX, 0, 0}, {1, PY, 0, 0}, {1, PZ, 0, 0},
    {1, NX, 0, 0}, {1, NY, 0, 0}, {1, NZ, 0, 0},
    {1, PX, PX, 0}, {1, PX, PY, 0}, {1, PX, PZ, 0}, {1, PY, PY, 0}, {1, PY, PZ, 0}, {1, PZ, PZ, 0},
    {2, 0, 0, 0},
    {2, PX, 0, 0}, {2, PY, 0, 0}, {2, PZ, 0, 0},
    {2, NX, 0, 0}, {2, NY, 0, 0}, {2, NZ, 0, 0},
    {2, PX, PX, 0}, {2, PX, PY, 0}, {2, PX, PZ, 0}, {2, PY, PY, 0}, {2, PY, PZ, 0}, {2, PZ, PZ, 0},
    {2, NX, NX, 0}, {2, NX, NY, 0}, {2, NX, NZ, 0}, {2, NY, NY, 0}, {2, NY, NZ, 0}, {2, NZ, NZ, 0},
    {2, NX, NX, PX}, {2, NX, NY, PY}, {2, NX, NZ, PZ},
    {2, NY, NX, PX}, {2, NY, NY, PY}, {2, NY, NZ, PZ},
    {2, NZ, NX, PX}, {2, NZ, NY, PY}, {2, NZ, NZ, PZ},
    {3, PX, PX, 0}, {3, PX, PY, 0}, {3, PX, PZ, 0}, {3, PY, PY, 0}, {3, PY, PZ, 0}, {3, PZ, PZ, 0},
    {3, PX, PX, PX}, {3, PX, PX, PY}, {3, PX, PX, PZ}, {3, PX, PY, PY}, {3, PX, PY, PZ},
    {3, PX, PZ, PZ}, {3, PY, PY, PY}, {3, PY, PY, PZ}, {3, PY, PZ, PZ}, {3, PZ, PZ, PZ},
    {0, NX, 0, 0}, {0, NY, 0, 0}, {0, NZ, 0, 0},
    {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // 60..63 unused (weight power 0 of "1" is
                                                                 // never read: lanes >= M_USED are masked)

// ---- dual numbers: value + one tangent (the lane's moment) -----------------------------
struct Dd {
  double v, d;
};
FB_HD Dd mk(double v, double d = 0.0) { return Dd{v, d}; }
FB_HD Dd operator+(Dd a, Dd b) { return Dd{a.v + b.v, a.d + b.d}; }
FB_HD Dd operator-(Dd a, Dd b) { return Dd{a.v - b.v, a.d - b.d}; }
FB_HD Dd operator-(Dd a) { return Dd{-a.v, -a.d}; }
FB_HD Dd operator*(Dd a, Dd b) { return Dd{a.v * b.v, a.d * b.v + a.v * b.d}; }
FB_HD Dd operator*(double a, Dd b) { return Dd{a * b.v, a * b.d}; }
FB_HD Dd operator/(Dd a, Dd b) {
  const double q = a.v / b.v;
  return Dd{q, (a.d - q * b.d) / b.v};
}
FB_HD Dd dsqrt(Dd a) {
  const double r = sqrt(a.v);
  return Dd{r, a.d / (2.0 * r)};
}
// torch.clamp passes the gradient where min <= x <= max (inclusive)
FB_HD Dd dclamp_min(Dd a, double lo) { return a.v >= lo ? a : Dd{lo, 0.0}; }
FB_HD Dd dclamp(Dd a, double lo, double hi) {
  if (a.v < lo) return Dd{lo, 0.0};
  if (a.v > hi) return Dd{hi, 0.0};
  return a;
}

FB_HD int sym6(int a, int b) {   // xx xy xz yy yz zz
  if (a > b) { const int t = a; a = b; b = t; }
  return a == 0 ? b : (a == 1 ? 2 + b : 5);
}
FB_HD int sym10(int a, int b, int c) {   // sorted triple -> xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz
  int t;
  if (a > b) { t = a; a = b; b = t; }
  if (b > c) { t = b; b = c; c = t; }
  if (a > b) { t = a; a = b; b = t; }
  if (a == 0) return b == 0 ? c : (b == 1 ? 2 + c : 5);
  if (a == 1) return b == 1 ? 5 + c : 8;
  return 9;
}

// Eigen-decomposition of a symmetric 3x3 (values only): cyclic Jacobi, eigenvalues descending,
// eigenvector columns with their largest-magnitude component positive — the arithmetic and the
// conventions of pn_sym3_eig_kernel (fit.hip), which the per-segment API path uses.
FB_HD void jacobi3(const double G[3][3], double w[3], double V[3][3]) {
  double a[3][3], v[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      a[i][j] = 0.5 * (G[i][j] + G[j][i]);
      v[i][j] = i == j ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 32; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    if (off <= 1e-300 || off <= 1e-18 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
        for (int k = 0; k < 3; ++k) {
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int ord[3] = {0, 1, 2};
  const double ww[3] = {a[0][0], a[1][1], a[2][2]};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (ww[ord[j]] < ww[ord[j + 1]]) {
        const int tmp = ord[j];
        ord[j] = ord[j + 1];
        ord[j + 1] = tmp;
      }
  for (int c = 0; c < 3; ++c) {
    const int o = ord[c];
    w[c] = ww[o];
    int big = 0;
    for (int k = 1; k < 3; ++k)
      if (fabs(v[k][o]) > fabs(v[big][o])) big = k;
    const double sg = v[big][o] < 0 ? -1.0 : 1.0;
    for (int k = 0; k < 3; ++k) V[k][c] = sg * v[k][o];
  }
}

// Right singular vector of the SMALLEST singular value of a tall matrix A given its Gram
// matrix G = A^T A (dual), with the tangent the reference's CustomSVD backward implies:
//   backward (fitting_utils.py:385-417):  gA = 2 U S sym(K^T o (V^T gV)) V^T,
//   K_ij = 1 / (guard(s_i - s_j) (s_i + s_j)),  guard(x) = sign(x) max(|x|, 1e-6), K_ii = 0,
// which is gG = V sym(K^T o (V^T gV)) V^T in terms of G, whose transpose is the forward rule
//   dV = V (K^T o (V^T dG V)).
// sv (out): singular values sqrt(max(eigenvalue, 0)), descending.
FB_HD void min_singular_vector(const Dd G[3][3], Dd a[3], double sv[3]) {
  double Gv[3][3], w[3], V[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Gv[i][j] = G[i][j].v;
  jacobi3(Gv, w, V);
  for (int i = 0; i < 3; ++i) sv[i] = sqrt(fmax(w[i], 0.0));
  // the reference's K is built from fp32 singular values
  double s32[3];
  for (int i = 0; i < 3; ++i) s32[i] = (double)(float)sv[i];
  double da[3] = {0.0, 0.0, 0.0};
  for (int i = 0; i < 2; ++i) {
    // D_i2 = v_i^T dG v_2
    double Di2 = 0.0;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Di2 += V[r][i] * 0.5 * (G[r][c].d + G[c][r].d) * V[c][2];
    const double diff = s32[2] - s32[i];
    const double sgn = diff > 0 ? 1.0 : (diff < 0 ? -1.0 : 0.0);
    const double kneg = sgn * fmax(fabs(diff), 1e-6);
    const double K2i = (1.0 / kneg) * (1.0 / (s32[2] + s32[i]));
    for (int r = 0; r < 3; ++r) da[r] += V[r][i] * K2i * Di2;
  }
  for (int r = 0; r < 3; ++r) a[r] = Dd{V[r][2], da[r]};
}

FB_HD void eigvals3(const Dd G[3][3], double w[3]) {
  double Gv[3][3], V[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Gv[i][j] = G[i][j].v;
  jacobi3(Gv, w, V);
}

// 3x3 solve, Gaussian elimination with partial pivoting on the values
FB_HD void solve3(const Dd A_[3][3], const Dd b_[3], Dd x[3]) {
  Dd A[3][3], b[3];
  for (int i = 0; i < 3; ++i) {
    b[i] = b_[i];
    for (int j = 0; j < 3; ++j) A[i][j] = A_[i][j];
  }
  for (int c = 0; c < 3; ++c) {
    int piv = c;
    for (int r = c + 1; r < 3; ++r)
      if (fabs(A[r][c].v) > fabs(A[piv][c].v)) piv = r;
    if (piv != c) {
      for (int j = 0; j < 3; ++j) { const Dd t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
      const Dd t = b[c]; b[c] = b[piv]; b[piv] = t;
    }
    for (int r = c + 1; r < 3; ++r) {
      const Dd f = A[r][c] / A[c][c];
      for (int j = c; j < 3; ++j) A[r][j] = A[r][j] - f * A[c][j];
      b[r] = b[r] - f * b[c];
    }
  }
  for (int r = 2; r >= 0; --r) {
    Dd acc = b[r];
    for (int j = r + 1; j < 3; ++j) acc = acc - A[r][j] * x[j];
    x[r] = acc / A[r][r];
  }
}

// LeastSquares.lstsq on normal-equation blocks (src/fitting_utils.py:32-85): G = A^T A,
// rhs = A^T Y of a tall n_rows x 3 system.  Full numerical rank (torch.matrix_rank convention:
// singular values above max * max(shape) * eps32): x = G^-1 rhs.  Otherwise ridge regression on
// the normal equations with the smallest lambda in {1e-6 * 10^i, i < 7} that restores full rank
// (best_lambda), the ridge system itself going through the same test (the recursion of the
// reference).  Returns 0, or 1 for non-finite input / no full-rank system within 4 levels.
FB_HD int lstsq3(const Dd G_[3][3], const Dd rhs_[3], int n_rows, Dd x[3], double* lambda_used) {
  Dd M[3][3], r[3];
  for (int i = 0; i < 3; ++i) {
    r[i] = rhs_[i];
    for (int j = 0; j < 3; ++j) M[i][j] = G_[i][j];
  }
  double w[3];
  eigvals3(M, w);
  for (int i = 0; i < 3; ++i)
    if (!isfinite(w[i])) return 1;
  // level 0: singular values of the tall A are sqrt(eig(G))
  double sv[3];
  for (int i = 0; i < 3; ++i) sv[i] = sqrt(fmax(w[i], 0.0));
  double tol = sv[0] * (double)(n_rows > 3 ? n_rows : 3) * FB_EPS;
  int rank = (sv[0] > tol) + (sv[1] > tol) + (sv[2] > tol);
  if (rank == 3) {
    solve3(M, r, x);
    *lambda_used = 0.0;
    return 0;
  }
  for (int level = 0; level < 4; ++level) {
    // best_lambda on the symmetric Gram matrix M: its singular values are |eig + lambda|
    double lamb = 1e-6;
    for (int it = 0; it < 7; ++it) {
      const double s0 = fabs(w[0] + lamb), s1 = fabs(w[1] + lamb), s2 = fabs(w[2] + lamb);
      const double mx = fmax(s0, fmax(s1, s2));
      const double t = mx * 3.0 * FB_EPS;
      if ((s0 > t) + (s1 > t) + (s2 > t) == 3) break;
      lamb *= 10.0;
    }
    *lambda_used = lamb;
    for (int i = 0; i < 3; ++i) M[i][i] = M[i][i] + mk(lamb);
    // the recursion's rank test of the square ridge system (max(shape) = 3)
    eigvals3(M, w);
    for (int i = 0; i < 3; ++i)
      if (!isfinite(w[i])) return 1;
    const double a0 = fabs(w[0]), a1 = fabs(w[1]), a2 = fabs(w[2]);
    const double mx = fmax(a0, fmax(a1, a2));
    tol = mx * 3.0 * FB_EPS;
    rank = (a0 > tol) + (a1 > tol) + (a2 > tol);
    if (rank == 3) {
      solve3(M, r, x);
      return 0;
    }
    // still deficient: normal equations of the square system, M <- M^T M, r <- M^T r
    Dd M2[3][3], r2[3];
    for (int i = 0; i < 3; ++i) {
      r2[i] = mk(0.0);
      for (int k = 0; k < 3; ++k) r2[i] = r2[i] + M[k][i] * r[k];
      for (int j = 0; j < 3; ++j) {
        M2[i][j] = mk(0.0);
        for (int k = 0; k < 3; ++k) M2[i][j] = M2[i][j] + M[k][i] * M[k][j];
      }
    }
    for (int i = 0; i < 3; ++i) {
      r[i] = r2[i];
      for (int j = 0; j < 3; ++j) M[i][j] = M2[i][j];
    }
    eigvals3(M, w);
  }
  return 1;
}

// Gram matrix of the weighted, centred cloud  A_i = w_i (x_i - c),  c = sum(w x) / (sum(w) + eps):
//   G = S2xx - c m2^T - m2 c^T + s2 c c^T
FB_HD void centred_gram(Dd s1, Dd s2, const Dd m1[3], const Dd m2[3], const Dd C2[6], Dd c[3],
                                    Dd G[3][3]) {
  const Dd ws = s1 + mk(FB_EPS);
  for (int i = 0; i < 3; ++i) c[i] = m1[i] / ws;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) G[i][j] = C2[sym6(i, j)] - c[i] * m2[j] - m2[i] * c[j] + s2 * c[i] * c[j];
}

// Fit.fit_plane_torch (primitive_forward.py:708-729): a = smallest right singular vector of
// w (x - c), d = sum w (a . x) / (sum w + eps) = a . c
FB_HD void fit_plane(Dd s1, Dd s2, const Dd m1[3], const Dd m2[3], const Dd C2[6], Dd a[3], Dd* d) {
  Dd c[3], G[3][3];
  double sv[3];
  centred_gram(s1, s2, m1, m2, C2, c, G);
  min_singular_vector(G, a, sv);
  *d = a[0] * c[0] + a[1] * c[1] + a[2] * c[2];
}

// Fit.fit_sphere_torch (primitive_forward.py:746-769) from moments of the (possibly projected)
// points: A_i = 2 w_i (c - x_i), Y_i = w_i (w_i |x_i|^2 - q1 / ws);  centre = -lstsq(A, Y),
// r^2 = sum w |x - centre|^2 / ws clamped at 1e-3, r = sqrt(clamp(r^2, 1e-5)).
FB_HD int fit_sphere(Dd s1, Dd s2, const Dd m1[3], const Dd m2[3], const Dd C2[6], Dd q1, Dd q3,
                                 const Dd t3[3], int n_rows, Dd centre[3], Dd* radius, double* lamb) {
  Dd c[3], G[3][3], rhs[3], x[3];
  centred_gram(s1, s2, m1, m2, C2, c, G);
  const Dd ws = s1 + mk(FB_EPS);
  const Dd nrm = q1 / ws;
  for (int i = 0; i < 3; ++i) {
    rhs[i] = 2.0 * (c[i] * q3 - t3[i] - nrm * (c[i] * s2 - m2[i]));
    for (int j = 0; j < 3; ++j) G[i][j] = 4.0 * G[i][j];
  }
  const int bad = lstsq3(G, rhs, n_rows, x, lamb);
  for (int i = 0; i < 3; ++i) centre[i] = -x[i];
  Dd r2 = q1;
  for (int i = 0; i < 3; ++i) r2 = r2 - 2.0 * (centre[i] * m1[i]) + centre[i] * centre[i] * s1;
  r2 = dclamp_min(r2 / ws, 1e-3);
  *radius = dsqrt(dclamp_min(r2, 1e-5));
  return bad;
}


// ---- stage 2 body: moments of one segment -> parameters (dual: tangent w.r.t. moment `lane`) ----
//   plane    : a(3), d            sphere : centre(3), r
//   cylinder : axis(3), centre(3), r      cone : apex(3), axis(3), [theta: stage 3]
// status bit 0: lstsq failed (non-finite / no full-rank ridge system); bit 1: null cone
// (condition number of w n above 1e5: the reference returns a zero cone without gradient).
FB_HD void fit_segment(const double* msum, int lane, int type, int n_rows, Dd out[FB_NPAR], int* st_out,
                       double* lamb_out) {
#define MOM(i) (Dd{msum[(i)], lane == (i) ? 1.0 : 0.0})
  int st = 0;
  double lamb = 0.0;
  for (int k = 0; k < FB_NPAR; ++k) out[k] = mk(0.0);
  const Dd s1 = MOM(M_S1), s2 = MOM(M_S2);
  Dd m1[3], m2[3], C1[6], C2[6], C3[6];
  for (int i = 0; i < 3; ++i) {
    m1[i] = MOM(M_M1 + i);
    m2[i] = MOM(M_M2 + i);
  }
  for (int i = 0; i < 6; ++i) {
    C1[i] = MOM(M_C1 + i);
    C2[i] = MOM(M_C2 + i);
    C3[i] = MOM(M_C3 + i);
  }
  if (type == FB_PLANE) {
    Dd a[3], d;
    fit_plane(s1, s2, m1, m2, C2, a, &d);
    out[0] = a[0]; out[1] = a[1]; out[2] = a[2]; out[3] = d;
  } else if (type == FB_SPHERE) {
    const Dd q1 = C1[0] + C1[3] + C1[5];
    const Dd q3 = C3[0] + C3[3] + C3[5];
    Dd t3[3];
    for (int a = 0; a < 3; ++a)
      t3[a] = MOM(M_T3 + sym10(a, 0, 0)) + MOM(M_T3 + sym10(a, 1, 1)) + MOM(M_T3 + sym10(a, 2, 2));
    Dd ctr[3], r;
    st |= fit_sphere(s1, s2, m1, m2, C2, q1, q3, t3, n_rows, ctr, &r, &lamb);
    out[0] = ctr[0]; out[1] = ctr[1]; out[2] = ctr[2]; out[3] = r;
  } else if (type == FB_CYLINDER) {
    // axis: smallest right singular vector of w n (primitive_forward.py:784-806), normalised
    // with the reference's "+ EPS"
    Dd Gn[3][3], a[3];
    double sv[3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Gn[i][j] = MOM(M_NN2 + sym6(i, j));
    min_singular_vector(Gn, a, sv);
    const Dd nrm = dsqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) + mk(FB_EPS);
    for (int i = 0; i < 3; ++i) a[i] = a[i] / nrm;
    // projection x' = Q x, Q = I - a a^T; |x'|^2 = x^T Q^T Q x
    Dd Q[3][3], QQ[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Q[i][j] = mk(i == j ? 1.0 : 0.0) - a[i] * a[j];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        QQ[i][j] = mk(0.0);
        for (int k = 0; k < 3; ++k) QQ[i][j] = QQ[i][j] + Q[k][i] * Q[k][j];
      }
    Dd pm1[3], pm2[3], pC2[6], pt3[3];
    for (int i = 0; i < 3; ++i) {
      pm1[i] = mk(0.0);
      pm2[i] = mk(0.0);
      for (int k = 0; k < 3; ++k) {
        pm1[i] = pm1[i] + Q[i][k] * m1[k];
        pm2[i] = pm2[i] + Q[i][k] * m2[k];
      }
    }
    for (int i = 0; i < 3; ++i)
      for (int j = i; j < 3; ++j) {
        Dd acc = mk(0.0);
        for (int k = 0; k < 3; ++k)
          for (int l = 0; l < 3; ++l) acc = acc + Q[i][k] * C2[sym6(k, l)] * Q[j][l];
        pC2[sym6(i, j)] = acc;
      }
    Dd pq1 = mk(0.0), pq3 = mk(0.0);
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 3; ++l) {
        pq1 = pq1 + QQ[k][l] * C1[sym6(k, l)];
        pq3 = pq3 + QQ[k][l] * C3[sym6(k, l)];
      }
    // sum w^3 |x'|^2 x'_a = Q_ab T3_bcd (Q^T Q)_cd
    for (int i = 0; i < 3; ++i) {
      Dd acc = mk(0.0);
      for (int bq = 0; bq < 3; ++bq) {
        Dd inner = mk(0.0);
        for (int c = 0; c < 3; ++c)
          for (int d = 0; d < 3; ++d) inner = inner + MOM(M_T3 + sym10(bq, c, d)) * QQ[c][d];
        acc = acc + Q[i][bq] * inner;
      }
      pt3[i] = acc;
    }
    Dd ctr[3], r;
    st |= fit_sphere(s1, s2, pm1, pm2, pC2, pq1, pq3, pt3, n_rows, ctr, &r, &lamb);
    out[0] = a[0]; out[1] = a[1]; out[2] = a[2];
    out[3] = ctr[0]; out[4] = ctr[1]; out[5] = ctr[2];
    out[6] = r;
  } else if (type == FB_CONE) {
    // apex: least squares n_i . c = n_i . p_i, rows weighted by w (primitive_forward.py:808-843)
    Dd Gn[3][3], rhs[3];
    for (int i = 0; i < 3; ++i) {
      rhs[i] = MOM(M_NNP + 3 * i) + MOM(M_NNP + 3 * i + 1) + MOM(M_NNP + 3 * i + 2);
      for (int j = 0; j < 3; ++j) Gn[i][j] = MOM(M_NN2 + sym6(i, j));
    }
    double w[3];
    eigvals3(Gn, w);
    const double sv0 = sqrt(fmax(w[0], 0.0)), sv2 = sqrt(fmax(w[2], 0.0));
    if (!(sv0 / fmax(sv2, 1e-300) <= 1e5)) {
      st |= 2;   // null cone: apex 0, axis (1,0,0), theta 0, no gradient
      out[3] = mk(1.0);
    } else {
      Dd c[3];
      st |= lstsq3(Gn, rhs, n_rows, c, &lamb);
      // axis: plane fit of the normals, oriented against the mean normal
      Dd n1[3], n2[3], NN[6], a[3], dd;
      for (int i = 0; i < 3; ++i) {
        n1[i] = MOM(M_N1 + i);
        n2[i] = MOM(M_N2 + i);
      }
      for (int i = 0; i < 6; ++i) NN[i] = MOM(M_NN2 + i);
      fit_plane(s1, s2, n1, n2, NN, a, &dd);
      const double dotn = msum[M_NS] * a[0].v + msum[M_NS + 1] * a[1].v + msum[M_NS + 2] * a[2].v;
      if (dotn > 0)
        for (int i = 0; i < 3; ++i) a[i] = -a[i];
      out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
      out[3] = a[0]; out[4] = a[1]; out[5] = a[2];
    }
  }
#undef MOM
  *st_out = st;
  *lamb_out = lamb;
}

FB_HD float cone_acos_term(float px, float py, float pz, const float c[3], const float a[3],
                                              float* ux, float* uy, float* uz, float* nu_, float* t_) {
  *ux = px - c[0];
  *uy = py - c[1];
  *uz = pz - c[2];
  const float nrm = fmaxf(sqrtf(*ux * *ux + *uy * *uy + *uz * *uz), 1e-12f);   // F.normalize eps
  const float t = (*ux * a[0] + *uy * a[1] + *uz * a[2]) / nrm;
  *nu_ = nrm;
  *t_ = t;
  return acosf(fminf(fabsf(t), 0.999f));
}

// one point of the cone's second pass: acc = [sum w acos, d/d apex (3), d/d axis (3), sum w]
FB_HD void cone_point(float px, float py, float pz, float w, const float c[3], const float a[3], double acc[8]) {
    float ux, uy, uz, nu, t;
    const float f = cone_acos_term(px, py, pz, c, a, &ux, &uy, &uz, &nu, &t);
    acc[0] += (double)w * f;
    acc[7] += (double)w;
    const float at = fabsf(t);
    if (at <= 0.999f) {   // clamp(max) passes the gradient inclusively
      // d acos(|t|) = -sign(t) / sqrt(1 - t^2) dt;  t = (u . a) / |u|,  u = p - c
      const float g = -(t > 0 ? 1.f : (t < 0 ? -1.f : 0.f)) / sqrtf(1.f - at * at);
      const float u[3] = {ux, uy, uz};
      const bool free_norm = nu > 1e-12f;   // below the eps the normalisation is a constant scale
      for (int k = 0; k < 3; ++k) {
        // dt/du_k = a_k / |u| - (u.a) u_k / |u|^3 ;  du/dc = -I
        const float dtdu = a[k] / nu - (free_norm ? t * u[k] / (nu * nu) : 0.f);
        acc[1 + k] += (double)(w * g * (-dtdu));
        acc[4 + k] += (double)(w * g * (u[k] / nu));
      }
    }
}

// adjoint of the moment pass for one point: sum_m gM_m e_m w^(e_m - 1) phi_m(z)
FB_HD double wmom_bwd_point(const double* gM, double w, const double z[7]) {
  const double dw[4] = {0.0, 1.0, 2.0 * w, 3.0 * w * w};
  double g = 0.0;
  for (int m = 0; m < M_NS; ++m) {   // the e = 0 moments do not depend on w
    const FbMono mono = fb_table[m];
    g += gM[m] * dw[mono.e] * z[mono.i1] * z[mono.i2] * z[mono.i3];
  }
  return g;
}

// fp32 dual numbers with up to 7 tangents (the parameters of the primitive)
#define FB_NT 7
struct Df {
  float v;
  float d[FB_NT];
};
FB_HD Df fc(float v) {
  Df r;
  r.v = v;
  for (int k = 0; k < FB_NT; ++k) r.d[k] = 0.f;
  return r;
}
FB_HD Df fvar(float v, int k) {
  Df r = fc(v);
  r.d[k] = 1.f;
  return r;
}
FB_HD Df operator+(Df a, Df b) {
  Df r;
  r.v = a.v + b.v;
  for (int k = 0; k < FB_NT; ++k) r.d[k] = a.d[k] + b.d[k];
  return r;
}
FB_HD Df operator-(Df a, Df b) {
  Df r;
  r.v = a.v - b.v;
  for (int k = 0; k < FB_NT; ++k) r.d[k] = a.d[k] - b.d[k];
  return r;
}
FB_HD Df operator*(Df a, Df b) {
  Df r;
  r.v = a.v * b.v;
  for (int k = 0; k < FB_NT; ++k) r.d[k] = a.d[k] * b.v + a.v * b.d[k];
  return r;
}
FB_HD Df operator/(Df a, Df b) {
  Df r;
  r.v = a.v / b.v;
  for (int k = 0; k < FB_NT; ++k) r.d[k] = (a.d[k] - r.v * b.d[k]) / b.v;
  return r;
}
FB_HD Df fscale(Df a, float g) {   // chain rule: value already set by the caller
  for (int k = 0; k < FB_NT; ++k) a.d[k] *= g;
  return a;
}
FB_HD Df fsqrt(Df a) {
  const float r = sqrtf(a.v);
  Df o = fscale(a, 0.5f / r);
  o.v = r;
  return o;
}
// torch.norm's gradient at 0 is 0
FB_HD Df fnorm3(Df x, Df y, Df z) {
  const float r = sqrtf(x.v * x.v + y.v * y.v + z.v * z.v);
  Df o;
  o.v = r;
  const float ir = r > 0.f ? 1.f / r : 0.f;
  for (int k = 0; k < FB_NT; ++k) o.d[k] = (x.v * x.d[k] + y.v * y.d[k] + z.v * z.d[k]) * ir;
  return o;
}
FB_HD Df fclamp(Df a, float lo, float hi) {
  if (a.v < lo) return fc(lo);
  if (a.v > hi) return fc(hi);
  return a;
}

FB_HD Df residual_point(int type, float px, float py, float pz, const float* th, int sqrt_flag) {
  Df d;
  if (type == FB_PLANE) {
    // (p . a - d)^2
    const Df e = fc(px) * fvar(th[0], 0) + fc(py) * fvar(th[1], 1) + fc(pz) * fvar(th[2], 2) - fvar(th[3], 3);
    d = e * e;
  } else if (type == FB_SPHERE) {
    // (|p - c| - r)^2
    const Df e = fnorm3(fc(px) - fvar(th[0], 0), fc(py) - fvar(th[1], 1), fc(pz) - fvar(th[2], 2)) - fvar(th[3], 3);
    d = e * e;
  } else if (type == FB_CYLINDER) {
    // v = p - c; (sqrt(clamp(|v|^2 - (v . a)^2, 1e-5)) - r)^2
    const Df vx = fc(px) - fvar(th[3], 3), vy = fc(py) - fvar(th[4], 4), vz = fc(pz) - fvar(th[5], 5);
    const Df prj = vx * fvar(th[0], 0) + vy * fvar(th[1], 1) + vz * fvar(th[2], 2);
    const Df q = fclamp(vx * vx + vy * vy + vz * vz - prj * prj, 1e-5f, __builtin_inff());
    const Df e = fsqrt(q) - fvar(th[6], 6);
    d = e * e;
  } else {
    // cone: v = p - apex + 1e-8; alpha = acos(clamp(v . axis / (|v| + 1e-7), +-0.999));
    //       (|v| sin(clamp(|alpha - theta|, max 3.142/2)))^2
    const Df vx = fc(px) - fvar(th[0], 0) + fc(1e-8f), vy = fc(py) - fvar(th[1], 1) + fc(1e-8f),
             vz = fc(pz) - fvar(th[2], 2) + fc(1e-8f);
    const Df mod = fnorm3(vx, vy, vz);
    const Df ax = fclamp((vx * fvar(th[3], 3) + vy * fvar(th[4], 4) + vz * fvar(th[5], 5)) / (mod + fc(1e-7f)),
                         -0.999f, 0.999f);
    Df alpha = fscale(ax, -1.f / sqrtf(1.f - ax.v * ax.v));
    alpha.v = acosf(ax.v);
    Df da = alpha - fvar(th[6], 6);
    if (da.v < 0.f) da = fc(0.f) - da;
    else if (da.v == 0.f) da = fc(0.f);          // |x|' = sign(x) = 0 at 0
    da = fclamp(da, -__builtin_inff(), 3.142f / 2.0f);
    Df sn = fscale(da, cosf(da.v));
    sn.v = sinf(da.v);
    const Df e = mod * sn;
    d = e * e;
  }
  if (sqrt_flag) d = fsqrt(fclamp(d, 1e-5f, __builtin_inff()));   // guard_sqrt
  return d;
}

