/* neptune_oracle.c — CPU restatement (plain C99, fp64) of the NEPTUNE back-end path.
 *
 * TEST INFRASTRUCTURE ONLY (see neptune_oracle.h).  "parity unpinned" against a live
 * Gurobi/GLPK/CGAL run; pinned by tests/golden/ instead.
 *
 * The QP is built in the reference's own variable space — 12K polynomial coefficients with the
 * 9K+6 equality rows written out — following PolySolverGurobi::addObjective/addConstraints row
 * for row, and solved by a dense primal-dual interior point.  The GPU product solves a reduced
 * (null-space) form of the same problem, so agreement between the two checks the reduction.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  -ffp-contract=off is
 * required: hull and separator outputs are compared bit-for-bit with the HIP kernels.
 */
#include "neptune_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* MINVO basis                                                                                 */
/* ------------------------------------------------------------------------------------------ */
/* Inverses of mt::basisConverter::A_pos_mv_rest (mader_types.hpp:152-157) and A_vel_mv_rest
 * (:159-163), t in [0,1]: exact rational inverse of the double-rounded literals, rounded to
 * double (tests/golden/make_golden.py regenerates and checks them).  The reference inverts
 * A_mv*diag(T^-3,T^-2,T^-1,1) with Eigen (solver_gurobi_poly.cpp:61,93); (A*C)^-1 = C^-1*A^-1. */
static const double A_POS_INV[4][4] = {
    {-0.03203276669713047, -0.09273093424558249, 0.3420572455666699, 1.1023313949144335},
    {-0.05111494245568798, -0.046272612998418894, 0.5458234872124772, 1.0979806946005568},
    {-0.07454781852812224, 0.203951949894552, 0.796048050105448, 1.0745478185281223},
    {1.0, 1.0, 0.9999999999999996, 0.9999999999999993}};
static const double A_VEL_INV[3][3] = {
    {-0.07735026918962577, 0.16666666666666635, 1.077350269189625},
    {-0.07735026918962577, 0.49999999999999967, 1.077350269189625},
    {1.0000000000000002, 1.0000000000000009, 1.0000000000000016}};

/* A_rest_pos_basis_inverse_ for interval length T (solver_gurobi_poly.cpp:39-45,61,93). */
static void pos_inv_T(double T, double M[4][4]) {
  double tp[4] = {T * T * T, T * T, T, 1.0};
  for (int j = 0; j < 4; j++)
    for (int k = 0; k < 4; k++) M[j][k] = tp[j] * A_POS_INV[j][k];
}
/* A_rest_vel_basis_inverse321_ (solver_gurobi_poly.cpp:47-51,62,94-97). */
static void vel_inv321_T(double T, double M[3][3]) {
  double tv[3] = {T * T, T, 1.0};
  double m321[3] = {3.0, 2.0, 1.0};
  for (int j = 0; j < 3; j++)
    for (int k = 0; k < 3; k++) M[j][k] = m321[j] * (tv[j] * A_VEL_INV[j][k]);
}

void orc_pos_ctrl_pts(const double P[4], double T, double Q[4]) {
  double M[4][4];
  pos_inv_T(T, M);
  for (int k = 0; k < 4; k++) Q[k] = ((P[0] * M[0][k] + P[1] * M[1][k]) + P[2] * M[2][k]) + P[3] * M[3][k];
}
void orc_vel_ctrl_pts(const double P[4], double T, double Qv[3]) {
  double M[3][3];
  vel_inv321_T(T, M);
  for (int k = 0; k < 3; k++) Qv[k] = (P[0] * M[0][k] + P[1] * M[1][k]) + P[2] * M[2][k];
}

/* ------------------------------------------------------------------------------------------ */
/* 2-D convex hull (cu::convexHullOfPoints2d, cgal_utils.cpp:157-174)                          */
/* ------------------------------------------------------------------------------------------ */
static double cross3(const double o[2], const double a[2], const double b[2]) {
  return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0]);
}
static int lex_less(const double a[2], const double b[2]) {
  return a[0] < b[0] || (a[0] == b[0] && a[1] < b[1]);
}
/* Andrew monotone chain; CGAL::convex_hull_2 returns the extreme points counter-clockwise
 * starting at the lexicographically smallest one, collinear points dropped. */
int orc_convex_hull_2d(int n, const double (*pts)[2], double (*out)[2]) {
  if (n <= 0) return 0;
  double p[64][2];
  if (n > 64) n = 64;
  for (int i = 0; i < n; i++) { p[i][0] = pts[i][0]; p[i][1] = pts[i][1]; }
  /* insertion sort, lexicographic (stable order is irrelevant: equal points are identical) */
  for (int i = 1; i < n; i++) {
    double kx = p[i][0], ky = p[i][1];
    int j = i - 1;
    double key[2] = {kx, ky};
    while (j >= 0 && lex_less(key, p[j])) { p[j + 1][0] = p[j][0]; p[j + 1][1] = p[j][1]; j--; }
    p[j + 1][0] = kx; p[j + 1][1] = ky;
  }
  /* unique */
  int m = 0;
  for (int i = 0; i < n; i++)
    if (m == 0 || p[i][0] != p[m - 1][0] || p[i][1] != p[m - 1][1]) { p[m][0] = p[i][0]; p[m][1] = p[i][1]; m++; }
  if (m == 1) { out[0][0] = p[0][0]; out[0][1] = p[0][1]; return 1; }
  double h[130][2];
  int k = 0;
  for (int i = 0; i < m; i++) { /* lower hull */
    while (k >= 2 && cross3(h[k - 2], h[k - 1], p[i]) <= 0.0) k--;
    h[k][0] = p[i][0]; h[k][1] = p[i][1]; k++;
  }
  int lo = k + 1;
  for (int i = m - 2; i >= 0; i--) { /* upper hull */
    while (k >= lo && cross3(h[k - 2], h[k - 1], p[i]) <= 0.0) k--;
    h[k][0] = p[i][0]; h[k][1] = p[i][1]; k++;
  }
  k--; /* last == first */
  for (int i = 0; i < k; i++) { out[i][0] = h[i][0]; out[i][1] = h[i][1]; }
  return k;
}

/* ------------------------------------------------------------------------------------------ */
/* Hulls of a committed trajectory over one interval (neptune.cpp:349-452, 288-309)            */
/* ------------------------------------------------------------------------------------------ */
static int lower_bound_d(const double* a, int n, double v) { /* first i with a[i] >= v */
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) / 2; if (a[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}
static int upper_bound_d(const double* a, int n, double v) { /* first i with a[i] > v */
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) / 2; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
  return lo;
}

/* Returns 1 when a capacity of the fixed-size records was exceeded (more than NEP_HULL_MAX_CP / 4 committed segments
 * overlap the interval, or a hull has more than NEP_HULL_MAX_V vertices): the product reports NEP_E_CAP there, and the
 * output is then NOT the reference's (which has no such limit), else 0. */
int orc_hull_of_interval(const nep_pwp* pwp, double t0, double t1, double T_span,
                         const double delta[2], double (*hull)[2], int* nv, double (*hull0)[2],
                         int* nv0) {
  int n = pwp->n_seg;
  int overflow = 0;
  double pts[4 * NEP_HULL_MAX_CP][2], pts0[NEP_HULL_MAX_CP][2];
  int np = 0, np0 = 0;
  /* neptune.cpp:379-389 */
  int first = lower_bound_d(pwp->times, n + 1, t0) - 1;
  int last = upper_bound_d(pwp->times, n + 1, t1) - 1;
  if (first < 0) first = 0; if (first > n - 1) first = n - 1;
  if (last < 0) last = 0; if (last > n - 1) last = n - 1;
  if (last - first + 1 > NEP_HULL_MAX_CP / 4) overflow = 1;
  for (int i = first; i <= last && np0 + 4 <= NEP_HULL_MAX_CP; i++) {
    double _t; /* neptune.cpp:399-424 */
    if (i != last) _t = pwp->times[i + 1] - pwp->times[i];
    else if (t1 > pwp->times[i + 1]) _t = pwp->times[i + 1] - pwp->times[i];
    else _t = t1 - pwp->times[i];
    if (_t > T_span) _t = T_span; else if (_t < 0) _t = 0;
    double c[4] = {_t * _t * _t, _t * _t, _t, 1.0};
    for (int k = 0; k < 4; k++) { /* V = (P * C) * A^-1, neptune.cpp:426-429 */
      double v[2];
      for (int ax = 0; ax < 2; ax++) {
        const double* P = pwp->coeff[ax][i];
        v[ax] = (((P[0] * c[0]) * A_POS_INV[0][k] + (P[1] * c[1]) * A_POS_INV[1][k]) +
                 (P[2] * c[2]) * A_POS_INV[2][k]) + (P[3] * c[3]) * A_POS_INV[3][k];
      }
      /* neptune.cpp:436-446: delta.norm()<1e-6 -> no inflation */
      if (sqrt(delta[0] * delta[0] + delta[1] * delta[1]) < 1e-6) {
        pts[np][0] = v[0]; pts[np][1] = v[1]; np++;
      } else {
        pts[np][0] = v[0] + delta[0]; pts[np][1] = v[1] + delta[1]; np++;
        pts[np][0] = v[0] + delta[0]; pts[np][1] = v[1] - delta[1]; np++;
        pts[np][0] = v[0] - delta[0]; pts[np][1] = v[1] - delta[1]; np++;
        pts[np][0] = v[0] - delta[0]; pts[np][1] = v[1] + delta[1]; np++;
      }
      pts0[np0][0] = v[0]; pts0[np0][1] = v[1]; np0++;
    }
  }
  double tmp[64][2];
  int k = orc_convex_hull_2d(np, pts, tmp);
  if (k > NEP_HULL_MAX_V) { k = NEP_HULL_MAX_V; overflow = 1; }
  for (int i = 0; i < k; i++) { hull[i][0] = tmp[i][0]; hull[i][1] = tmp[i][1]; }
  *nv = k;
  k = orc_convex_hull_2d(np0, pts0, tmp);
  if (k > NEP_HULL_MAX_V) k = NEP_HULL_MAX_V;
  for (int i = 0; i < k; i++) { hull0[i][0] = tmp[i][0]; hull0[i][1] = tmp[i][1]; }
  *nv0 = k;
  return overflow;
}

int orc_inflate_static(int nv, const double (*v)[2], double sd, double (*out)[2]) {
  double pts[64][2];
  int np = 0;
  for (int j = 0; j < nv && np + 4 <= 64; j++) { /* neptune.cpp:648-657 */
    pts[np][0] = v[j][0] + sd; pts[np][1] = v[j][1] + sd; np++;
    pts[np][0] = v[j][0] + sd; pts[np][1] = v[j][1] - sd; np++;
    pts[np][0] = v[j][0] - sd; pts[np][1] = v[j][1] - sd; np++;
    pts[np][0] = v[j][0] - sd; pts[np][1] = v[j][1] + sd; np++;
  }
  return orc_convex_hull_2d(np, pts, out);
}

/* ------------------------------------------------------------------------------------------ */
/* Separator (separator_glpk.cpp:248-373; 3-set :375-498 == A := A u A+)                       */
/* ------------------------------------------------------------------------------------------ */
#define SEP_MIN_GAP 1e-7

/* The LP  find (n,d): n.a+d >= 1 (a in A), n.b+d <= -1 (b in B), zero objective, has as
 * vertices exactly the lines through two points of one set (both rows tight) that touch the
 * nearest point of the other set (third tight row); GLPK returns whichever vertex its pivoting
 * reaches.  This restatement returns the vertex with the largest geometric gap; ties go to the
 * first candidate in the order: pairs of A, then pairs of B.  Pairs of A are (p<q) in
 * lexicographic order; when A is known to be a counter-clockwise convex polygon (hulls, inflated
 * statics) only its edges can be tight and only they are tried — (0,1),(V-1,0),(1,2),... — with the
 * A rows taken as satisfied by convexity.
 * Projections are taken relative to p so that both points of the pair project to exactly 0. */
/* Candidate bookkeeping: gaps are compared as num^2/len2 by cross-multiplication, so that only
 * the winning candidate needs a square root and divisions. */
typedef struct { int have; double num, len2, sg, tA, nx, ny, px, py; } sep_best;

/* ---- which LP vertex is returned: a study knob (scripts/separator_sensitivity.py), NOT part of the restated path ----
 * GLPK hands back whichever vertex its simplex reaches; the product's rule (policy 0) is the largest gap.  The other
 * policies return other ADMISSIBLE vertices of the same LP, so that the dependence of the QP optimum on this
 * solver-defined choice can be measured:
 *   1 a pseudo-random admissible vertex (seeded)      2 the admissible vertex with the smallest gap
 *   3 the admissible vertex that leaves the reference control points of the current segment the least room
 *   4 the vertex a textbook two-phase Bland simplex reaches (orc_separator_simplex)
 *   5 the vertex a primal simplex of GLPK's default class reaches (orc_separator_glpk_class): standard start basis, projected
 *     steepest-edge pricing, Harris ratio test — also selectable in the product (nep_batch_set_separator_rule)
 * Thread-local: concurrent callers (bench.py's cpu_baseline) keep the default. */
#define SEP_MAX_CAND 512
static __thread int g_policy = 0;
static __thread int g_sep_rule = 0;       /* 0: the largest-gap vertex (default), 1: the GLPK-class simplex's (nep_batch_set_separator_rule) */
static __thread unsigned long long g_rng = 1;
static __thread int g_cur_seg = 0;
static __thread double g_ref_ctrl[NEP_MAX_POL][4][2];
static __thread int g_ncand = 0;
static __thread sep_best g_cand[SEP_MAX_CAND];
static __thread long g_stat_lps = 0, g_stat_vertices = 0;
void orc_set_separator_rule(int rule) { g_sep_rule = rule; }
static __thread double g_tol_res = 1e-10, g_tol_gap = 1e-11, g_tol_res_inv = 1e10;     /* (round 6: a tenth of rounds 1-5's 1e-9 / 1e-10 — strictly converged solves of two row sets or two hosts then agree to 3e-8 in the coefficients instead of 2e-6, for 0.3 iterations more) */     /* nep_batch_set_tolerances */
void orc_set_qp_tolerances(double residual_tol, double gap_tol) { g_tol_res = residual_tol; g_tol_gap = gap_tol; g_tol_res_inv = residual_tol == 1e-10 ? 1e10 : (residual_tol == 1e-9 ? 1e9 : 1.0 / residual_tol); }
void orc_set_vertex_policy(int policy, unsigned long long seed, const double* ref_ctrl /* [NEP_MAX_POL][4][2] or NULL */) {
  g_sep_rule = policy == 5 ? 1 : 0;        /* policy 5 IS the product's second rule */
  if (policy == 5) policy = 0;
  g_policy = policy; g_rng = seed * 2862933555777941757ULL + 3037000493ULL; g_stat_lps = 0; g_stat_vertices = 0;
  if (ref_ctrl) memcpy(g_ref_ctrl, ref_ctrl, sizeof(g_ref_ctrl)); else memset(g_ref_ctrl, 0, sizeof(g_ref_ctrl));
}
void orc_vertex_policy_stats(long* n_lps, long* n_vertices) { *n_lps = g_stat_lps; *n_vertices = g_stat_vertices; }

static void sep_consider(sep_best* b, double num, double len2, double sg, double tA, double nx, double ny, double px, double py) {
  if (!(num > 0.0)) return;
  if (g_policy != 0 && (num * num) > (SEP_MIN_GAP * SEP_MIN_GAP) * len2 && g_ncand < SEP_MAX_CAND) {
    sep_best* c = &g_cand[g_ncand++]; c->have = 1; c->num = num; c->len2 = len2; c->sg = sg; c->tA = tA; c->nx = nx; c->ny = ny; c->px = px; c->py = py;
  }
  int better;
  if (!b->have) better = (num * num) > (SEP_MIN_GAP * SEP_MIN_GAP) * len2;
  else better = (num * num) * b->len2 > (b->num * b->num) * len2;
  if (better) { b->have = 1; b->num = num; b->len2 = len2; b->sg = sg; b->tA = tA; b->nx = nx; b->ny = ny; b->px = px; b->py = py; }
}

static void sep_pair(const double p[2], const double q[2], int from_A, int nA, const double (*A)[2],
                     int nB, const double (*B)[2], sep_best* best) {
  double ex = q[0] - p[0], ey = q[1] - p[1];
  double nx = -ey, ny = ex;
  double len2 = nx * nx + ny * ny;
  if (!(len2 > 0.0)) return;
  double minA = INFINITY, maxA = -INFINITY, minB = INFINITY, maxB = -INFINITY;
  for (int i = 0; i < nA; i++) { double t = nx * (A[i][0] - p[0]) + ny * (A[i][1] - p[1]); if (t < minA) minA = t; if (t > maxA) maxA = t; }
  for (int i = 0; i < nB; i++) { double t = nx * (B[i][0] - p[0]) + ny * (B[i][1] - p[1]); if (t < minB) minB = t; if (t > maxB) maxB = t; }
  double np_ = -INFINITY, nm = -INFINITY, tAp = 0.0, tAm = 0.0; /* gap numerators for +n / -n pointing toward A */
  if (from_A) {
    if (minA >= 0.0) { np_ = 0.0 - maxB; tAp = 0.0; }
    if (maxA <= 0.0) { nm = minB - 0.0; tAm = 0.0; }
  } else {
    if (maxB <= 0.0) { np_ = minA - 0.0; tAp = minA; }
    if (minB >= 0.0) { nm = 0.0 - maxA; tAm = maxA; }
  }
  if (np_ >= nm) sep_consider(best, np_, len2, 1.0, tAp, nx, ny, p[0], p[1]);
  else sep_consider(best, nm, len2, -1.0, tAm, nx, ny, p[0], p[1]);
}

/* Edge p->q of a counter-clockwise convex polygon A (CGAL's hull output order, and ours): A lies on
 * the left of its own edges, so the edge is a tight pair by construction and only B decides. */
static void sep_edge_ccw(const double p[2], const double q[2], int nB, const double (*B)[2], sep_best* best) {
  double ex = q[0] - p[0], ey = q[1] - p[1];
  double nx = -ey, ny = ex;
  double len2 = nx * nx + ny * ny;
  if (!(len2 > 0.0)) return;
  double maxB = -INFINITY;
  for (int i = 0; i < nB; i++) { double t = nx * (B[i][0] - p[0]) + ny * (B[i][1] - p[1]); if (t > maxB) maxB = t; }
  sep_consider(best, 0.0 - maxB, len2, 1.0, 0.0, nx, ny, p[0], p[1]);
}

int orc_separator_simplex(int nA, const double (*A)[2], int nB, const double (*B)[2], double nd[3]);
int orc_separator_glpk_class(int nA, const double (*A)[2], int nB, const double (*B)[2], double nd[3], int* n_pivots);
static int separator_impl(int nA, const double (*A)[2], int a_ordered, int nB, const double (*B)[2], double nd[3]) {
  if (g_sep_rule == 1) return orc_separator_glpk_class(nA, A, nB, B, nd, NULL);      /* (the point sets' order plays no role for it) */
  sep_best best; best.have = 0; best.num = 0; best.len2 = 1; best.sg = 1; best.tA = 0; best.nx = best.ny = best.px = best.py = 0;
  g_ncand = 0;
  if (a_ordered && nA >= 3) {
    for (int p = 0; p < nA - 1; p++) {
      sep_edge_ccw(A[p], A[p + 1], nB, B, &best);
      if (p == 0) sep_edge_ccw(A[nA - 1], A[0], nB, B, &best); /* closing edge, same orientation */
    }
  } else {
    for (int p = 0; p < nA; p++) for (int q = p + 1; q < nA; q++) sep_pair(A[p], A[q], 1, nA, A, nB, B, &best);
  }
  for (int p = 0; p < nB; p++) for (int q = p + 1; q < nB; q++) sep_pair(B[p], B[q], 0, nA, A, nB, B, &best);
  if (!best.have && nA > 0 && nB > 0) {
    /* degenerate sets (every pair coincident, e.g. a hovering agent against a point):
     * direction between the centroids, supported at the extreme points */
    double ca[2] = {0, 0}, cb[2] = {0, 0};
    for (int i = 0; i < nA; i++) { ca[0] += A[i][0]; ca[1] += A[i][1]; }
    for (int i = 0; i < nB; i++) { cb[0] += B[i][0]; cb[1] += B[i][1]; }
    ca[0] /= nA; ca[1] /= nA; cb[0] /= nB; cb[1] /= nB;
    double nx = ca[0] - cb[0], ny = ca[1] - cb[1];
    double len2 = nx * nx + ny * ny;
    if (len2 > 0.0) {
      double minA = INFINITY, maxB = -INFINITY;
      for (int i = 0; i < nA; i++) { double t = nx * (A[i][0] - cb[0]) + ny * (A[i][1] - cb[1]); if (t < minA) minA = t; }
      for (int i = 0; i < nB; i++) { double t = nx * (B[i][0] - cb[0]) + ny * (B[i][1] - cb[1]); if (t > maxB) maxB = t; }
      sep_consider(&best, minA - maxB, len2, 1.0, minA, nx, ny, cb[0], cb[1]);
    }
  }
  if (best.have && g_policy != 0) {   /* study knob: another admissible vertex of the same LP (see orc_set_vertex_policy) */
    g_stat_lps++; g_stat_vertices += g_ncand;
    if (g_policy == 4) { double t[3]; if (orc_separator_simplex(nA, A, nB, B, t)) { nd[0] = t[0]; nd[1] = t[1]; nd[2] = t[2]; return 1; } }

    else if (g_ncand > 0) {
      int pick = 0;
      if (g_policy == 1) { g_rng = g_rng * 6364136223846793005ULL + 1442695040888963407ULL; pick = (int)((g_rng >> 33) % (unsigned long long)g_ncand); }
      else {
        double best_v = INFINITY;
        for (int c = 0; c < g_ncand; c++) {
          const sep_best* k = &g_cand[c]; double len = sqrt(k->len2), v;
          if (g_policy == 2) v = k->num / len;                      /* the gap */
          else {                                                    /* room left to the reference control points (metres) */
            v = INFINITY;
            for (int q = 0; q < 4; q++) {
              double room = k->sg * k->tA / len - k->sg * (k->nx * (g_ref_ctrl[g_cur_seg][q][0] - k->px) + k->ny * (g_ref_ctrl[g_cur_seg][q][1] - k->py)) / len;
              if (room < v) v = room;
            }
          }
          if (v < best_v) { best_v = v; pick = c; }
        }
      }
      best = g_cand[pick];
    }
  }
  if (best.have) { /* the winning LP vertex in the reference's epsilon = 1 scaling */
    double len = sqrt(best.len2);
    double g = best.num / len;
    double s = 2.0 / g;
    double n1 = s * (best.sg * best.nx / len), n2 = s * (best.sg * best.ny / len);
    nd[0] = n1; nd[1] = n2;
    nd[2] = (1.0 - s * (best.sg * best.tA / len)) - (n1 * best.px + n2 * best.py);
    return 1;
  }
  nd[0] = nd[1] = nd[2] = 0.0;
  return 0;
}

int orc_separator(int nA, const double (*A)[2], int nB, const double (*B)[2], double nd[3]) {
  return separator_impl(nA, A, 0, nB, B, nd);
}
int orc_separator_ordered(int nA, const double (*A)[2], int nB, const double (*B)[2], double nd[3]) {
  return separator_impl(nA, A, 1, nB, B, nd);
}

/* Two-phase primal simplex, Bland's rule, dense tableau.  Rows: A: a.x+d - u = 1 ; B: -(b.x+d)
 * - u = 1 with x = x+ - x- (6 columns), surplus u >= 0, one artificial per row. */
int orc_separator_simplex(int nA, const double (*A)[2], int nB, const double (*B)[2],
                          double nd[3]) {
  int m = nA + nB;
  if (m <= 0 || m > 40) return 0;
  int nv = 6 + m + m; /* x+-(6), surplus(m), artificial(m) */
  double* T = (double*)calloc((size_t)(m + 1) * (nv + 1), sizeof(double));
  int* basis = (int*)malloc(sizeof(int) * m);
#define TB(r, c) T[(r) * (nv + 1) + (c)]
  for (int r = 0; r < m; r++) {
    double px, py, sg;
    if (r < nA) { px = A[r][0]; py = A[r][1]; sg = 1.0; } else { px = B[r - nA][0]; py = B[r - nA][1]; sg = -1.0; }
    double c3[3] = {sg * px, sg * py, sg};
    for (int j = 0; j < 3; j++) { TB(r, 2 * j) = c3[j]; TB(r, 2 * j + 1) = -c3[j]; }
    TB(r, 6 + r) = -1.0;
    TB(r, 6 + m + r) = 1.0;
    TB(r, nv) = 1.0;
    basis[r] = 6 + m + r;
  }
  /* phase-1 cost row: minimise sum of artificials -> reduced costs = -(sum of rows) */
  for (int c = 0; c <= nv; c++) { double s = 0; for (int r = 0; r < m; r++) s += TB(r, c); TB(m, c) = -s; }
  for (int r = 0; r < m; r++) TB(m, 6 + m + r) = 0.0;
  int ok = 0;
  for (int it = 0; it < 2000; it++) {
    int enter = -1;
    for (int c = 0; c < 6 + m; c++) if (TB(m, c) < -1e-9) { enter = c; break; } /* Bland */
    if (enter < 0) break;
    int leave = -1; double best = 0;
    for (int r = 0; r < m; r++) if (TB(r, enter) > 1e-9) {
      double ratio = TB(r, nv) / TB(r, enter);
      if (leave < 0 || ratio < best - 1e-12 || (fabs(ratio - best) <= 1e-12 && basis[r] < basis[leave])) { leave = r; best = ratio; }
    }
    if (leave < 0) break; /* unbounded phase 1 cannot happen */
    double pv = TB(leave, enter);
    for (int c = 0; c <= nv; c++) TB(leave, c) /= pv;
    for (int r = 0; r <= m; r++) if (r != leave) {
      double f = TB(r, enter);
      if (f != 0.0) for (int c = 0; c <= nv; c++) TB(r, c) -= f * TB(leave, c);
    }
    basis[leave] = enter;
  }
  if (-TB(m, nv) < 1e-7) { /* sum of artificials == 0 -> feasible */
    double x[6] = {0, 0, 0, 0, 0, 0};
    for (int r = 0; r < m; r++) if (basis[r] < 6) x[basis[r]] = TB(r, nv);
    nd[0] = x[0] - x[1]; nd[1] = x[2] - x[3]; nd[2] = x[4] - x[5];
    ok = 1;
  }
#undef TB
  free(T); free(basis);
  return ok;
}

/* ---- vertex policy 5 / separator rule 1: a primal simplex of the class GLPK's glp_simplex runs by default -------------
 * The reference calls glp_simplex with glp_init_smcp's defaults (separator_glpk.cpp:39-41, 336): primal simplex, projected
 * steepest-edge pricing (GLP_PT_PSE), Harris' two-pass ratio test (GLP_RT_HAR), tol_bnd = tol_dj = 1e-7, tol_piv = 1e-9, no
 * presolve, no scaling, started from the standard basis glp_create_prob leaves behind: every auxiliary (row) variable
 * basic, the three free structurals (n1, n2, d) non-basic at zero.  GLPK 4.65's source is not in the tree (downloaded at
 * build time, submodules/separator/cmake/glpk.cmake.in:6), so this is that documented algorithm CLASS, not a clone of its
 * pivot sequence: ties and loop orders are this file's (stated below), and GLPK's internals that the documentation does not
 * pin (its phase-1 bookkeeping, periodic refactorisation, reference-space resets) are not imitated.
 *
 * LP in GLPK's standard form: r = [A 1; B 1] x, rows of A bounded r >= 1, rows of B bounded r <= -1, x free, objective 0.
 * A basis leaves exactly three variables non-basic; the structurals are expressed through them, x = W v with W = G^-1, G's
 * rows the gradients of the non-basic variables (e_k for a structural, (a_x, a_y, 1) for a row) and v their values (0 for a
 * structural, the bound for a row): the whole tableau is A W, recomputed from G every iteration (3 x 3 by cofactors).
 * Phase 1 minimises the sum of infeasibilities of the basic rows (cost -1 on a row below its lower bound, +1 above its upper
 * bound; an infeasible row blocks where it becomes feasible); with the zero objective the first feasible vertex is optimal.
 * Pricing: largest d_j^2 / gamma_j among the eligible non-basics (free structural: |d_j| > tol_dj; row at its lower bound:
 * d_j < -tol_dj; at its upper bound: d_j > tol_dj), gamma_j the projected steepest-edge weight for the reference space of
 * the initial non-basics, gamma_j = [j in R] + sum over basic structurals k of W[k][j]^2 (evaluated from its definition).
 * Ratio test: Harris — pass 1 the largest step with every bound relaxed by delta = tol_bnd (1 + 1e-3 |bound|) — pass 2 among
 * the rows that block within it the one with the largest |pivot|; ties to the lowest row index, non-basic slots in order
 * 0, 1, 2; the leaving row takes the entering variable's slot.  Returns 1 with the vertex (n1, n2, d), 0 when phase 1 ends
 * with an infeasible row (no separating line) or after SPX_MAX_IT pivots. */
#define SPX_MAX_IT 60
#define SPX_TOL_BND 1e-7
#define SPX_TOL_DJ 1e-7
#define SPX_TOL_PIV 1e-9
static int spx_inv3(const double G[3][3], double W[3][3]) {
  const double c00 = G[1][1] * G[2][2] - G[1][2] * G[2][1], c01 = G[1][2] * G[2][0] - G[1][0] * G[2][2], c02 = G[1][0] * G[2][1] - G[1][1] * G[2][0];
  const double det = (G[0][0] * c00 + G[0][1] * c01) + G[0][2] * c02;
  if (!(fabs(det) > 1e-300)) return 0;
  const double id = 1.0 / det;
  W[0][0] = c00 * id; W[1][0] = c01 * id; W[2][0] = c02 * id;
  W[0][1] = (G[0][2] * G[2][1] - G[0][1] * G[2][2]) * id; W[1][1] = (G[0][0] * G[2][2] - G[0][2] * G[2][0]) * id; W[2][1] = (G[0][1] * G[2][0] - G[0][0] * G[2][1]) * id;
  W[0][2] = (G[0][1] * G[1][2] - G[0][2] * G[1][1]) * id; W[1][2] = (G[0][2] * G[1][0] - G[0][0] * G[1][2]) * id; W[2][2] = (G[0][0] * G[1][1] - G[0][1] * G[1][0]) * id;
  return 1;
}
int orc_separator_glpk_class(int nA, const double (*A)[2], int nB, const double (*B)[2], double nd[3], int* n_pivots) {
  const int m = nA + nB;
  nd[0] = nd[1] = nd[2] = 0.0;
  if (n_pivots) *n_pivots = 0;
  if (nA <= 0 || nB <= 0 || m > 64) return 0;
  int slot[3] = {-1, -2, -3};                 /* non-basic slots: -(k+1) structural k, i >= 0 row i */
  unsigned long long nb_rows = 0;             /* rows that are non-basic (sitting at their bound) */
  for (int it = 0; it <= SPX_MAX_IT; it++) {
    double G[3][3], W[3][3], v[3], x[3];
    for (int j = 0; j < 3; j++) {
      if (slot[j] < 0) { const int k = -slot[j] - 1; G[j][0] = k == 0; G[j][1] = k == 1; G[j][2] = k == 2; v[j] = 0.0; }
      else { const int i = slot[j]; const double* pt = i < nA ? A[i] : B[i - nA]; G[j][0] = pt[0]; G[j][1] = pt[1]; G[j][2] = 1.0; v[j] = i < nA ? 1.0 : -1.0; }
    }
    if (!spx_inv3(G, W)) return 0;
    for (int k = 0; k < 3; k++) x[k] = (W[k][0] * v[0] + W[k][1] * v[1]) + W[k][2] * v[2];
    /* reduced costs of the phase-1 objective over the basic rows, and whether any is infeasible */
    double d[3] = {0, 0, 0}; int n_inf = 0;
    for (int i = 0; i < m; i++) {
      if ((nb_rows >> i) & 1ULL) continue;
      const double* pt = i < nA ? A[i] : B[i - nA];
      const double r = (pt[0] * x[0] + pt[1] * x[1]) + x[2];
      const double bnd = i < nA ? 1.0 : -1.0, delta = SPX_TOL_BND * (1.0 + 1e-3 * fabs(bnd));
      double c = 0.0;
      if (i < nA) { if (r < bnd - delta) c = -1.0; } else { if (r > bnd + delta) c = 1.0; }
      if (c != 0.0) { n_inf++; for (int j = 0; j < 3; j++) d[j] += c * ((pt[0] * W[0][j] + pt[1] * W[1][j]) + W[2][j]); }
    }
    if (n_inf == 0) { nd[0] = x[0]; nd[1] = x[1]; nd[2] = x[2]; if (n_pivots) *n_pivots = it; return 1; }
    if (it == SPX_MAX_IT) break;
    /* pricing */
    int q = -1; double best = 0.0, sdir = 0.0;
    for (int j = 0; j < 3; j++) {
      double s_;
      if (slot[j] < 0) { if (d[j] < -SPX_TOL_DJ) s_ = 1.0; else if (d[j] > SPX_TOL_DJ) s_ = -1.0; else continue; }
      else if (slot[j] < nA) { if (d[j] < -SPX_TOL_DJ) s_ = 1.0; else continue; }       /* at its lower bound: may only increase */
      else { if (d[j] > SPX_TOL_DJ) s_ = -1.0; else continue; }                          /* at its upper bound: may only decrease */
      double gamma = slot[j] < 0 ? 1.0 : 0.0;
      for (int k = 0; k < 3; k++) { int basic = 1; for (int jj = 0; jj < 3; jj++) if (slot[jj] == -(k + 1)) basic = 0; if (basic) gamma += W[k][j] * W[k][j]; }
      if (!(gamma > 1e-300)) gamma = 1e-300;
      const double score = d[j] * d[j] / gamma;
      if (score > best) { best = score; q = j; sdir = s_; }
    }
    if (q < 0) return 0;                        /* no improving direction with infeasible rows left: the LP has no solution */
    /* Harris ratio test over the basic rows */
    double tmax = INFINITY;
    for (int pass = 0; pass < 2; pass++) {
      int p = -1; double piv = 0.0, step = 0.0;
      for (int i = 0; i < m; i++) {
        if ((nb_rows >> i) & 1ULL) continue;
        const double* pt = i < nA ? A[i] : B[i - nA];
        const double r = (pt[0] * x[0] + pt[1] * x[1]) + x[2];
        const double rho = sdir * ((pt[0] * W[0][q] + pt[1] * W[1][q]) + W[2][q]);
        if (!(fabs(rho) > SPX_TOL_PIV)) continue;
        const double bnd = i < nA ? 1.0 : -1.0, delta = SPX_TOL_BND * (1.0 + 1e-3 * fabs(bnd));
        /* the bound the row meets in this direction: an infeasible row where it becomes feasible, a feasible one where it
         * would stop being so */
        double dist;
        if (i < nA) { const int inf = r < bnd - delta; if (inf ? rho > 0 : rho < 0) dist = inf ? bnd - r : r - bnd; else continue; }
        else { const int inf = r > bnd + delta; if (inf ? rho < 0 : rho > 0) dist = inf ? r - bnd : bnd - r; else continue; }
        const double arho = fabs(rho);
        if (pass == 0) { const double t = (dist + delta) / arho; if (t < tmax) tmax = t; }
        else { const double t = dist / arho; if (t <= tmax && arho > piv) { piv = arho; p = i; step = t > 0.0 ? t : 0.0; } }
      }
      if (pass == 1) {
        if (p < 0) return 0;                    /* nothing blocks: cannot happen while the infeasibility decreases */
        (void)step;
        nb_rows |= 1ULL << p;
        if (slot[q] >= 0) nb_rows &= ~(1ULL << slot[q]);
        slot[q] = p;
      }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Dense helpers                                                                               */
/* ------------------------------------------------------------------------------------------ */
static int chol(int n, double* A) { /* in place lower, row-major; returns 0 ok */
  for (int j = 0; j < n; j++) {
    double d = A[j * n + j];
    for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0)) return -1;
    d = sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double v = A[i * n + j];
      for (int k = 0; k < j; k++) v -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = v / d;
    }
  }
  return 0;
}
static void chol_solve(int n, const double* L, double* b) {
  for (int i = 0; i < n; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= L[i * n + k] * b[k]; b[i] = v / L[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double v = b[i]; for (int k = i + 1; k < n; k++) v -= L[k * n + i] * b[k]; b[i] = v / L[i * n + i]; }
}

/* ------------------------------------------------------------------------------------------ */
/* QP in the reference's variable space                                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int nnz; int idx[8]; double val[8]; double rhs; } qrow;

typedef struct {
  int n, p, m;
  double* P; double* q; double c0;     /* 1/2 th'P th + q'th + c0 */
  double* E; double* e;                /* E th = e */
  qrow* rows;                          /* row . th <= rhs */
  int has_qc; double* C; double* cq; double cc; /* th'C th + 2 cq'th + cc <= 0 */
} qp_t;

static double qp_obj(const qp_t* Q, const double* th) {
  int n = Q->n; double v = Q->c0;
  for (int i = 0; i < n; i++) { double r = 0; for (int j = 0; j < n; j++) r += Q->P[i * n + j] * th[j]; v += 0.5 * th[i] * r + Q->q[i] * th[i]; }
  return v;
}
static double qc_val(const qp_t* Q, const double* th, double* grad) {
  int n = Q->n; double v = Q->cc;
  for (int i = 0; i < n; i++) { double r = 0; for (int j = 0; j < n; j++) r += Q->C[i * n + j] * th[j]; if (grad) grad[i] = 2.0 * (r + Q->cq[i]); v += th[i] * r + 2.0 * Q->cq[i] * th[i]; }
  return v;
}

/* Householder QR with column pivoting of A (rows x cols, row-major, overwritten): on return
 * the Householder vectors are in A/beta, perm holds the column order; returns the rank. */
static int qr_pivot(int rows, int cols, double* A, double* beta, int* perm, double tol) {
  int kmax = rows < cols ? rows : cols, rank = 0;
  double r00 = 0;
  for (int j = 0; j < cols; j++) perm[j] = j;
  for (int k = 0; k < kmax; k++) {
    int pj = k; double best = -1;
    for (int j = k; j < cols; j++) { double nn = 0; for (int i = k; i < rows; i++) nn += A[i * cols + j] * A[i * cols + j]; if (nn > best) { best = nn; pj = j; } }
    if (pj != k) { for (int i = 0; i < rows; i++) { double t = A[i * cols + k]; A[i * cols + k] = A[i * cols + pj]; A[i * cols + pj] = t; } int t = perm[k]; perm[k] = perm[pj]; perm[pj] = t; }
    double nrm = sqrt(best);
    if (k == 0) r00 = nrm;
    if (!(nrm > tol * (r00 > 0 ? r00 : 1.0))) break;
    double x0 = A[k * cols + k];
    double alpha = x0 >= 0 ? -nrm : nrm;
    double v0 = x0 - alpha;
    /* v = [v0, A[k+1..][k]]; beta = 2/(v'v) */
    double vv = v0 * v0; for (int i = k + 1; i < rows; i++) vv += A[i * cols + k] * A[i * cols + k];
    beta[k] = vv > 0 ? 2.0 / vv : 0.0;
    for (int j = k + 1; j < cols; j++) {
      double d = v0 * A[k * cols + j]; for (int i = k + 1; i < rows; i++) d += A[i * cols + k] * A[i * cols + j];
      d *= beta[k];
      A[k * cols + j] -= d * v0; for (int i = k + 1; i < rows; i++) A[i * cols + j] -= d * A[i * cols + k];
    }
    A[k * cols + k] = alpha; /* R diagonal; v0 kept separately */
    beta[kmax + k] = v0;
    rank++;
  }
  return rank;
}
/* y := Q y or Q' y for the reflectors stored by qr_pivot (A rows x cols). */
static void qr_apply(int rows, int cols, const double* A, const double* beta, int rank, int kmax, int transpose, double* y) {
  for (int t = 0; t < rank; t++) {
    int k = transpose ? t : rank - 1 - t;
    double v0 = beta[kmax + k];
    double d = v0 * y[k]; for (int i = k + 1; i < rows; i++) d += A[i * cols + k] * y[i];
    d *= beta[k];
    y[k] -= d * v0; for (int i = k + 1; i < rows; i++) y[i] -= d * A[i * cols + k];
  }
}

static _Thread_local int g_polish = 1, g_last_polished = 0;      /* (per calling thread, like the tolerance and rule settings: the multi-threaded CPU baseline and parallel tests do not mix their solves) orc_set_polish: the active-set polish of solves that end without the strict tests (on by default) */
void orc_set_polish(int on) { g_polish = on; }
static _Thread_local long g_stat_iters = 0, g_stat_trig = 0;      /* interior-point iterations and discarded predictors since the last orc_pass_stats (a device solve's passes = their sum) */
void orc_pass_stats(long* iters, long* trig) { if (iters) *iters = g_stat_iters; if (trig) *trig = g_stat_trig; g_stat_iters = 0; g_stat_trig = 0; }
int orc_last_polished(void) { const int v = g_last_polished; g_last_polished = 0; return v; }      /* (test hook: did a solve since the last call end on the polish?) */
static double hy_abs_slack(const double* g, const double* y, double h, int ny) { double a = h; for (int c = 0; c < ny; c++) a -= g[c] * y[c]; return a; }
/* Mehrotra predictor-corrector primal-dual interior point on
 *   min 1/2 th'P th + q'th  s.t.  E th = e,  G th <= h,  c(th) <= 0.
 * The equality rows are removed numerically first (Householder QR of E': th = th_p + Z y), then
 * the interior point runs on y with the dense rows G Z.  Returns 0 converged, 1 no solution
 * (cf. the status whitelist solver_gurobi_poly.cpp:832-836). */
static int qp_solve(const qp_t* Q, double* th, int* iters_out) {
  const int n = Q->n, p = Q->p, m = Q->m, qc = Q->has_qc;
  const int mt = m + qc;
  *iters_out = 0;
  /* ---- null space of E ---- */
  double* Et = (double*)malloc(sizeof(double) * ((size_t)n * (p + 1) + 4 * (size_t)(n + p + 2)));
  double* beta = Et + (size_t)n * (p + 1); int* perm = (int*)malloc(sizeof(int) * (p + 1));
  for (int i = 0; i < p; i++) for (int j = 0; j < n; j++) Et[j * p + i] = Q->E[i * n + j]; /* n x p */
  int kmax = n < p ? n : p;
  int rank = p > 0 ? qr_pivot(n, p, Et, beta, perm, 1e-11) : 0;
  /* particular solution: E' = Q R Pi'  ->  E th = e  <=>  R' (Q' th)[:rank] = (Pi' e)[:rank] */
  double* w1 = beta + 2 * (kmax + 1); double* thp = w1 + (n + p + 2);
  for (int i = 0; i < n; i++) w1[i] = 0.0;
  for (int i = 0; i < rank; i++) { double v = Q->e[perm[i]]; for (int k = 0; k < i; k++) v -= Et[k * p + i] * w1[k]; w1[i] = v / Et[i * p + i]; }
  for (int i = 0; i < n; i++) thp[i] = w1[i];
  qr_apply(n, p, Et, beta, rank, kmax, 0, thp); /* thp = Q [w;0] */
  double escale = 1.0, eres = 0.0;
  for (int i = 0; i < p; i++) { if (fabs(Q->e[i]) > escale) escale = fabs(Q->e[i]); double a = -Q->e[i]; for (int j = 0; j < n; j++) a += Q->E[i * n + j] * thp[j]; if (fabs(a) > eres) eres = fabs(a); }
  if (eres > 1e-6 * escale) { free(Et); free(perm); return 1; } /* inconsistent equalities */
  const int ny = n - rank;
  /* Z columns: Q e_{rank+c} */
  double* Z = (double*)malloc(sizeof(double) * ((size_t)n * (ny + 1)));
  for (int c = 0; c < ny; c++) { double* col = w1; for (int i = 0; i < n; i++) col[i] = 0.0; col[rank + c] = 1.0; qr_apply(n, p, Et, beta, rank, kmax, 0, col); for (int i = 0; i < n; i++) Z[i * ny + c] = col[i]; }
  /* reduced data */
  double* Gy = (double*)malloc(sizeof(double) * ((size_t)(m + 1) * (ny + 1) + (size_t)(m + 2)));
  double* hy = Gy + (size_t)(m + 1) * (ny + 1);
  for (int r = 0; r < m; r++) {
    const qrow* R = &Q->rows[r]; double a = 0;
    for (int c = 0; c < ny; c++) { double v = 0; for (int k = 0; k < R->nnz; k++) v += R->val[k] * Z[R->idx[k] * ny + c]; Gy[(size_t)r * ny + c] = v; }
    for (int k = 0; k < R->nnz; k++) a += R->val[k] * thp[R->idx[k]];
    hy[r] = R->rhs - a;
  }
  if (ny == 0) { /* a single point: feasible or not (Gurobi FeasibilityTol 1e-6) */
    int ok = 1;
    for (int r = 0; r < m; r++) if (hy[r] < -1e-6) ok = 0;
    if (qc && qc_val(Q, thp, NULL) > 1e-6) ok = 0;
    if (ok) memcpy(th, thp, sizeof(double) * n);
    free(Gy); free(Z); free(Et); free(perm);
    return ok ? 0 : 1;
  }
  double* Py = (double*)malloc(sizeof(double) * ((size_t)ny * ny * 3 + 12 * (size_t)(ny + n + 1)));
  double* Cy = Py + (size_t)ny * ny, *M = Cy + (size_t)ny * ny;
  double* qy = M + (size_t)ny * ny, *y = qy + (ny + n + 1), *rd = y + (ny + n + 1), *rhs = rd + (ny + n + 1), *dy = rhs + (ny + n + 1),
         *gq = dy + (ny + n + 1), *tn = gq + (ny + n + 1), *cqy = tn + (ny + n + 1), *yl = cqy + (ny + n + 1), *gth = yl + (ny + n + 1);
  /* Py = Z'PZ, qy = Z'(P thp + q), Cy = Z'CZ, cqy = Z'(C thp + cq), ccy */
  for (int i = 0; i < n; i++) { double v = Q->q[i]; for (int j = 0; j < n; j++) v += Q->P[i * n + j] * thp[j]; tn[i] = v; }
  for (int c = 0; c < ny; c++) { double v = 0; for (int i = 0; i < n; i++) v += Z[i * ny + c] * tn[i]; qy[c] = v; }
  for (int a = 0; a < ny; a++) for (int b = 0; b < ny; b++) { double v = 0, vc = 0; for (int i = 0; i < n; i++) { double ri = 0, rc = 0; for (int j = 0; j < n; j++) { ri += Q->P[i * n + j] * Z[j * ny + b]; rc += Q->C[i * n + j] * Z[j * ny + b]; } v += Z[i * ny + a] * ri; vc += Z[i * ny + a] * rc; } Py[a * ny + b] = v; Cy[a * ny + b] = vc; }
  double ccy = qc_val(Q, thp, gth); /* c(thp), grad at thp */
  for (int c = 0; c < ny; c++) { double v = 0; for (int i = 0; i < n; i++) v += Z[i * ny + c] * 0.5 * gth[i]; cqy[c] = v; }
  double obj0 = qp_obj(Q, thp);
  /* start at the projection of the guess */
  for (int c = 0; c < ny; c++) { double v = 0; for (int i = 0; i < n; i++) v += Z[i * ny + c] * (th[i] - thp[i]); y[c] = v; }
  /* start point of the rows (slack floor, mu0): development knobs, defaults = what the kernel uses */
  /* the solver's constants, overridable for experiments (read per solve, not per iteration) */
  const double EXP_FLOOR = getenv("ORC_EXP_FLOOR") ? atof(getenv("ORC_EXP_FLOOR")) : 0.1;
  const double EXP_MU0 = getenv("ORC_EXP_MU0") ? atof(getenv("ORC_EXP_MU0")) : 2.0;
  const double EXP_TAU = getenv("ORC_EXP_TAU") ? atof(getenv("ORC_EXP_TAU")) : 0.99999;
  const double EXP_GAPTOL = getenv("ORC_EXP_GAPTOL") ? atof(getenv("ORC_EXP_GAPTOL")) : g_tol_gap;
  const double EXP_TAU2 = getenv("ORC_EXP_TAU2") ? atof(getenv("ORC_EXP_TAU2")) : 1.0;
  const int EXP_TAU2_AFTER = getenv("ORC_EXP_TAU2_AFTER") ? atoi(getenv("ORC_EXP_TAU2_AFTER")) : 1;
  const int trace = getenv("ORC_QP_TRACE") != NULL;
  const int EXP_REFINE = getenv("ORC_EXP_REFINE") ? atoi(getenv("ORC_EXP_REFINE")) : 0;   /* experiment: refinement steps per Newton solve (0 = what the kernel does) */
  double* M0 = (double*)malloc(sizeof(double) * ((size_t)ny * ny + ny)); double* res = M0 + (size_t)ny * ny;
  /* Mehrotra's second-order term dsa*dla extrapolates the affine step to its full length.  From the tenth iteration on (the
     usual solve has ended by then), whenever less than a tenth of that step is admissible, the predictor is discarded: the
     corrector is formed as if the affine direction were zero (dsa = -rp, dla = -lam + w rp: what is left of the term vanishes
     with the primal residual), sigma as computed.  With the full term marginally feasible problems cycle (gap down 10x, then
     back up over three short steps, for ever) and infeasible ones blow up to 1e18 and idle to the iteration cap; without it the
     former converge and the latter stall within a few iterations.  Overridable for experiments. */
  const double EXP_CORR = getenv("ORC_EXP_CORR") ? atof(getenv("ORC_EXP_CORR")) : 0.1;
  const int EXP_CORR_IT = getenv("ORC_EXP_CORR_IT") ? atoi(getenv("ORC_EXP_CORR_IT")) : 10;      /* (the kernels' kCorrFromIt / kCorrMaxCount; round 5 tried 6 / 5 and took it back: scripts/giveup_rule_sweep.py, qp_common.h) */
  double alpha_aff = 1.0;
  int ntrig = 0, give_up = 0;
  const int EXP_CORR_MAX = getenv("ORC_EXP_CORR_MAX") ? atoi(getenv("ORC_EXP_CORR_MAX")) : 8;
  double* s = (double*)malloc(sizeof(double) * (mt + 1) * 10);
  double* lam = s + (mt + 1), *ds = lam + (mt + 1), *dl = ds + (mt + 1), *rp = dl + (mt + 1), *rc = rp + (mt + 1),
         *w = rc + (mt + 1), *gdx = w + (mt + 1), *dsa = gdx + (mt + 1), *dla = dsa + (mt + 1);
#define QCY(yv, grad, out) do { double v_ = ccy; for (int a_ = 0; a_ < ny; a_++) { double r_ = 0; for (int b_ = 0; b_ < ny; b_++) r_ += Cy[a_ * ny + b_] * (yv)[b_]; if (grad) (grad)[a_] = 2.0 * (r_ + cqy[a_]); v_ += (yv)[a_] * r_ + 2.0 * cqy[a_] * (yv)[a_]; } out = v_; } while (0)
  for (int r = 0; r < m; r++) { double a = 0; for (int c = 0; c < ny; c++) a += Gy[(size_t)r * ny + c] * y[c]; double sl = hy[r] - a; s[r] = sl > EXP_FLOOR ? sl : EXP_FLOOR; lam[r] = EXP_MU0 / s[r]; }
  if (qc) { double c; QCY(y, (double*)NULL, c); s[m] = (-c > 1e-3) ? -c : 1e-3; lam[m] = 1.0 / s[m]; }
  double qscale = 1.0; for (int c = 0; c < ny; c++) if (fabs(qy[c]) > qscale) qscale = fabs(qy[c]);
  int ret = 1, it = 0, loose_ok = 0, stall = 0, first_loose = -1;
  double best_merit = 0.0;
  for (it = 0; it < 100; it++) {
    g_stat_iters++;      /* (test hook: orc_pass_stats) */
    for (int a = 0; a < ny; a++) { double v = qy[a]; for (int b = 0; b < ny; b++) v += Py[a * ny + b] * y[b]; rd[a] = v; }
    for (int r = 0; r < m; r++) { double a = 0; const double* g = Gy + (size_t)r * ny; for (int c = 0; c < ny; c++) { a += g[c] * y[c]; rd[c] += g[c] * lam[r]; } rp[r] = a + s[r] - hy[r]; }
    if (qc) { double c; QCY(y, gq, c); rp[m] = c + s[m]; for (int a = 0; a < ny; a++) rd[a] += lam[m] * gq[a]; }
    double mu = 0; for (int r = 0; r < mt; r++) mu += s[r] * lam[r]; mu /= (mt > 0 ? mt : 1);
    double nrp = 0, nrd = 0;
    for (int r = 0; r < mt; r++) if (fabs(rp[r]) > nrp) nrp = fabs(rp[r]);
    for (int a = 0; a < ny; a++) if (fabs(rd[a]) > nrd) nrd = fabs(rd[a]);
    double obj = obj0; for (int a = 0; a < ny; a++) { double v = 0; for (int b = 0; b < ny; b++) v += Py[a * ny + b] * y[b]; obj += 0.5 * y[a] * v + qy[a] * y[a]; }
    double gap = mu * mt;
    if (nrp <= g_tol_res && nrd <= g_tol_res * qscale && gap <= EXP_GAPTOL * (1.0 + fabs(obj))) { ret = 0; break; }
    /* Loosely converged iterates: keep the one closest to the strict tolerances (merit <= 1 is the strict test) and stop three
       iterations after the first of them: with mu that small the weights lam/s amplify the rounding of the row activities into
       rd, so an iteration that has not passed the strict test by then never will */
    {
      const int is_loose = nrp <= 1e-6 && nrd <= 1e-6 * qscale && gap <= 1e-7 * (1.0 + fabs(obj));
      if (is_loose || first_loose >= 0) {
        const double merit = fmax(fmax(nrp * g_tol_res_inv, nrd / qscale * g_tol_res_inv), gap / (1.0 + fabs(obj)) / EXP_GAPTOL);
        const int better = is_loose && (!loose_ok || merit < best_merit);
        const int last = first_loose >= 0 && it - first_loose >= 3;
        if (first_loose < 0) first_loose = it;
        if (better) { loose_ok = 1; best_merit = merit; memcpy(yl, y, sizeof(double) * ny); }
        if (last) break;          /* the snapshot (possibly this very iterate) is the answer */
      }
    }
    for (int i = 0; i < ny * ny; i++) M[i] = Py[i];
    for (int r = 0; r < m; r++) { const double* g = Gy + (size_t)r * ny; w[r] = lam[r] / s[r]; for (int a = 0; a < ny; a++) { double wa = w[r] * g[a]; for (int b = 0; b <= a; b++) M[a * ny + b] += wa * g[b]; } }
    for (int a = 0; a < ny; a++) for (int b = a + 1; b < ny; b++) M[a * ny + b] = M[b * ny + a];
    if (qc) { w[m] = lam[m] / s[m]; for (int a = 0; a < ny; a++) for (int b = 0; b < ny; b++) M[a * ny + b] += lam[m] * 2.0 * Cy[a * ny + b] + w[m] * gq[a] * gq[b]; }
    if (EXP_REFINE) memcpy(M0, M, sizeof(double) * ny * ny);
    if (chol(ny, M)) break;
    double alpha = 1.0, sigma = 0.0;
    for (int pass = 0; pass < 2; pass++) {
      /* centring target: never below a tenth of the gap the strict test asks for (long steps would otherwise collapse mu to
         ~1e-15 in the last iteration and the weights lam/s with it to ~1e17) */
      double smu = sigma * mu; { const double fl = 0.1 * EXP_GAPTOL * (1.0 + fabs(obj)) / (mt > 0 ? mt : 1); if (smu < fl) smu = fl; }
      if (pass == 1 && it >= EXP_CORR_IT && alpha_aff < EXP_CORR) {
        g_stat_trig++;
        if (++ntrig > EXP_CORR_MAX) { give_up = 1; break; }     /* a solve that needs this more than eight times is not going to end (converging ones: at most five in 16 000) */
        for (int r = 0; r < mt; r++) { dsa[r] = -rp[r]; dla[r] = -lam[r] + w[r] * rp[r]; }
      }
      for (int r = 0; r < mt; r++) rc[r] = (pass == 0) ? s[r] * lam[r] : s[r] * lam[r] - smu + dsa[r] * dla[r];
      for (int a = 0; a < ny; a++) rhs[a] = -rd[a];
      for (int r = 0; r < m; r++) { const double* g = Gy + (size_t)r * ny; double v = rc[r] / s[r] - w[r] * rp[r]; for (int c = 0; c < ny; c++) rhs[c] += g[c] * v; }
      if (qc) { double v = rc[m] / s[m] - w[m] * rp[m]; for (int a = 0; a < ny; a++) rhs[a] += gq[a] * v; }
      for (int a = 0; a < ny; a++) dy[a] = rhs[a];
      chol_solve(ny, M, dy);
      for (int k = 0; k < EXP_REFINE; k++) {      /* experiment only (scripts/parity_floor.py): iterative refinement of the Newton solve */
        for (int a = 0; a < ny; a++) { double v = rhs[a]; for (int b = 0; b < ny; b++) v -= M0[a * ny + b] * dy[b]; res[a] = v; }
        chol_solve(ny, M, res); for (int a = 0; a < ny; a++) dy[a] += res[a];
      }
      for (int r = 0; r < m; r++) { const double* g = Gy + (size_t)r * ny; double a = 0; for (int c = 0; c < ny; c++) a += g[c] * dy[c]; gdx[r] = a; }
      if (qc) { double a = 0; for (int c = 0; c < ny; c++) a += gq[c] * dy[c]; gdx[m] = a; }
      for (int r = 0; r < mt; r++) { ds[r] = -rp[r] - gdx[r]; dl[r] = -rc[r] / s[r] + w[r] * (rp[r] + gdx[r]); }
      alpha = 1.0;
      for (int r = 0; r < mt; r++) {
        if (ds[r] < 0) { double a = -s[r] / ds[r]; if (a < alpha) alpha = a; }
        if (dl[r] < 0) { double a = -lam[r] / dl[r]; if (a < alpha) alpha = a; }
      }
      if (pass == 0) {
        alpha_aff = alpha;
        double mua = 0; for (int r = 0; r < mt; r++) mua += (s[r] + alpha * ds[r]) * (lam[r] + alpha * dl[r]);
        mua /= (mt > 0 ? mt : 1);
        double rr = mua / mu; sigma = rr * rr * rr;
        for (int r = 0; r < mt; r++) { dsa[r] = ds[r]; dla[r] = dl[r]; }
      }
    }
    if (give_up) break;
    { double tau = 1.0 - mu;      /* fraction of the step to the boundary: 1 - mu clamped to [0.999, EXP_TAU = 0.99999] */
      if (tau < 0.999) tau = 0.999;
      if (tau > EXP_TAU) tau = EXP_TAU;
      if (ntrig >= EXP_TAU2_AFTER && tau > EXP_TAU2) tau = EXP_TAU2;
      alpha *= tau; if (alpha > 1.0) alpha = 1.0; }
    if (trace) {
      /* (trace only) how close lam / |lam|_1 is to a Farkas ray: G'lam -> 0 with h'lam < 0 proves the rows infeasible */
      double l1 = 0, hl = 0, gl[64]; for (int c = 0; c < ny && c < 64; c++) gl[c] = 0;
      for (int r = 0; r < m; r++) { l1 += lam[r]; hl += hy[r] * lam[r]; const double* g = Gy + (size_t)r * ny; for (int c = 0; c < ny && c < 64; c++) gl[c] += g[c] * lam[r]; }
      double gmax = 0, g1 = 0; for (int c = 0; c < ny && c < 64; c++) { if (fabs(gl[c]) > gmax) gmax = fabs(gl[c]); g1 += fabs(gl[c]); }
      { double cmin = 1e300, cmax = 0; int nsm = 0; for (int r = 0; r < mt; r++) { const double c_ = s[r] * lam[r] / mu; if (c_ < cmin) cmin = c_; if (c_ > cmax) cmax = c_; if (c_ < 1e-3) nsm++; }
        fprintf(stderr, "   alpha_aff %.3e  centrality min %.3e max %.3e  pairs below 1e-3 mu: %d  ntrig %d\n", alpha_aff, cmin, cmax, nsm, ntrig); }
      fprintf(stderr, "it %2d nrp %.3e nrd %.3e (qs %.3e) gap %.3e obj %.9g sigma %.3e alpha %.3e loose %d | farkas: |G'l|_1/|l|_1 %.3e  h'l/|l|_1 %.3e  ratio %.3e\n", it, nrp, nrd, qscale, gap, obj, sigma, alpha, loose_ok, g1 / l1, hl / l1, g1 / fmax(-hl, 1e-300));
    }
    if (alpha < 1e-8) { if (++stall >= 3) break; } else stall = 0;
    for (int a = 0; a < ny; a++) y[a] += alpha * dy[a];
    for (int r = 0; r < mt; r++) { s[r] += alpha * ds[r]; lam[r] += alpha * dl[r]; }
  }
  /* Active-set polish (round 5; the product runs the same rule in qp_polish_kernel).  A solve that never passed the strict tests
     ends on the loose snapshot, or gives up (discarded predictors, stall, iteration cap, lost pivot) — the first on iterates whose
     dual residual sits on its rounding floor, the second also on FEASIBLE problems whose optimum is degenerate (no strict
     complementarity: the gap stalls at 1e-2, tests/golden/moving_hard_cases.npz).  Both are finished exactly: the rows whose slack
     at the last iterate is below 1e-6 (1 + |h|) are taken as the active set, the equality-constrained QP on a maximal independent
     subset of them is solved directly (KKT system), rows with a negative multiplier are dropped and violated rows added, at most
     six times (a certificate takes one to three; a row that comes back after leaving ends the attempt); a point that satisfies every row to 1e-9 (1 + |h|) with multipliers >= -1e-9 (1 + max|nu|) is the optimum of a
     strictly convex QP whatever iterate it was found from, and the solve counts as converged.  Not for problems with the terminal
     ball row (a quadratic constraint); an infeasible problem can never be certified (every row is checked). */
  if (ret != 0 && !qc && g_polish) {
    const double* ysrc = loose_ok ? yl : y;
    int finite = 1; for (int c = 0; c < ny; c++) if (!(fabs(ysrc[c]) < 1e100)) finite = 0;
    int* act = (int*)malloc(sizeof(int) * (m + 1)); int na = 0;
    /* (a start point that violates a row by more than 1e-4 (1 + |rhs|) is not polished: an interior point that gives up on a feasible
       problem has long driven the primal residual down; what is left are infeasible problems, which cannot be certified) */
    if (finite) for (int r = 0; r < m; r++) { const double* g = Gy + (size_t)r * ny; double a = hy_abs_slack(g, ysrc, hy[r], ny); if (-a > 1e-4 * (1.0 + fabs(Q->rows[r].rhs))) finite = 0; }
    if (finite) for (int r = 0; r < m; r++) { const double* g = Gy + (size_t)r * ny; double a = hy_abs_slack(g, ysrc, hy[r], ny); if (a < 1e-6 * (1.0 + fabs(Q->rows[r].rhs))) act[na++] = r; }
    int ok = 0, rounds = 0, n_dropped = 0, dropped[12];
    double* ys = (double*)malloc(sizeof(double) * (ny + 1));
    double* nu = (double*)malloc(sizeof(double) * (m + 1));
    double* Qb = (double*)malloc(sizeof(double) * (size_t)(ny + 1) * ny);
    for (rounds = 0; finite && rounds < 6; rounds++) {
      { int nq = 0, keep = 0;      /* a maximal independent subset, in row order (modified Gram-Schmidt) */
        for (int i = 0; i < na; i++) {
          const double* g = Gy + (size_t)act[i] * ny; double v[64]; double n0 = 0, n1 = 0;
          for (int c = 0; c < ny; c++) { v[c] = g[c]; n0 += g[c] * g[c]; }
          for (int k = 0; k < nq; k++) { double d = 0; for (int c = 0; c < ny; c++) d += Qb[k * ny + c] * v[c]; for (int c = 0; c < ny; c++) v[c] -= d * Qb[k * ny + c]; }
          for (int c = 0; c < ny; c++) n1 += v[c] * v[c];
          if (nq < ny && n1 > 1e-16 * n0 && n0 > 0) { const double inv = 1.0 / sqrt(n1); for (int c = 0; c < ny; c++) Qb[nq * ny + c] = v[c] * inv; nq++; act[keep++] = act[i]; }
        }
        na = keep; }
      const int nk = ny + na;
      double* K = (double*)calloc((size_t)nk * (nk + 1), sizeof(double));
      for (int a_ = 0; a_ < ny; a_++) { for (int b_ = 0; b_ < ny; b_++) K[a_ * (nk + 1) + b_] = Py[a_ * ny + b_]; K[a_ * (nk + 1) + nk] = -qy[a_]; }
      for (int i = 0; i < na; i++) { const double* g = Gy + (size_t)act[i] * ny; for (int c = 0; c < ny; c++) { K[(ny + i) * (nk + 1) + c] = g[c]; K[c * (nk + 1) + ny + i] = g[c]; } K[(ny + i) * (nk + 1) + nk] = hy[act[i]]; }
      int sing = 0;
      for (int c = 0; c < nk && !sing; c++) {
        int pv = c; double best = fabs(K[c * (nk + 1) + c]);
        for (int r = c + 1; r < nk; r++) if (fabs(K[r * (nk + 1) + c]) > best) { best = fabs(K[r * (nk + 1) + c]); pv = r; }
        if (!(best > 1e-10)) { sing = 1; break; }
        if (pv != c) for (int k = 0; k <= nk; k++) { double t = K[c * (nk + 1) + k]; K[c * (nk + 1) + k] = K[pv * (nk + 1) + k]; K[pv * (nk + 1) + k] = t; }
        for (int r = c + 1; r < nk; r++) { const double f = K[r * (nk + 1) + c] / K[c * (nk + 1) + c]; if (f != 0.0) for (int k = c; k <= nk; k++) K[r * (nk + 1) + k] -= f * K[c * (nk + 1) + k]; }
      }
      if (sing) { free(K); break; }
      double* x = (double*)malloc(sizeof(double) * nk);
      for (int r = nk - 1; r >= 0; r--) { double v = K[r * (nk + 1) + nk]; for (int k = r + 1; k < nk; k++) v -= K[r * (nk + 1) + k] * x[k]; x[r] = v / K[r * (nk + 1) + r]; }
      for (int a_ = 0; a_ < ny; a_++) ys[a_] = x[a_];
      for (int i = 0; i < na; i++) nu[i] = x[ny + i];
      free(x); free(K);
      double numax = 0; for (int i = 0; i < na; i++) if (fabs(nu[i]) > numax) numax = fabs(nu[i]);
      { /* the rows with a negative multiplier leave, all at once (a row that is missed comes back through the violation test) */
        const double wv = -1e-9 * (1.0 + numax); int keep = 0, n_neg = 0;
        for (int i = 0; i < na; i++) { if (nu[i] < wv) { n_neg++; if (n_dropped < 12) dropped[n_dropped++] = act[i]; } else act[keep++] = act[i]; }
        if (n_neg > 0) { na = keep; continue; } }
      int viol = -1; double vv = 0.0;
      for (int r = 0; r < m; r++) { const double* g = Gy + (size_t)r * ny; const double a = -hy_abs_slack(g, ys, hy[r], ny) / (1.0 + fabs(Q->rows[r].rhs)); if (a > 1e-9 && a > vv) { vv = a; viol = r; } }
      if (viol >= 0) { int have = 0; for (int i = 0; i < na; i++) have |= act[i] == viol; for (int i = 0; i < n_dropped; i++) have |= dropped[i] == viol; if (have) break;      /* (a row of the set still violated, or one that left with a negative multiplier comes back: the iteration would go round in circles — no certificate) */ int pos = na; while (pos > 0 && act[pos - 1] > viol) { act[pos] = act[pos - 1]; pos--; } act[pos] = viol; na++; continue; }
      ok = 1; break;
    }
    if (trace) fprintf(stderr, "polish: %s after %d rounds, %d active rows\n", ok ? "certified" : "no", rounds, na);
    if (ok) { memcpy(y, ys, sizeof(double) * ny); ret = 0; loose_ok = 0; g_last_polished = 1; }
    free(act); free(ys); free(nu); free(Qb);
  }
#undef QCY
  if (ret != 0 && loose_ok) { memcpy(y, yl, sizeof(double) * ny); ret = 0; }
  if (ret == 0) for (int i = 0; i < n; i++) { double v = thp[i]; for (int c = 0; c < ny; c++) v += Z[i * ny + c] * y[c]; th[i] = v; }
  *iters_out = it;
  free(s); free(Py); free(Gy); free(Z); free(Et); free(perm); free(M0);
  return ret;
}

/* ------------------------------------------------------------------------------------------ */
/* PolySolverGurobi::optimize                                                                  */
/* ------------------------------------------------------------------------------------------ */
#define VAR(ax, seg, j) ((ax) * 4 * K + (seg) * 4 + (j))

static void add_line_rows(qrow* rows, int* m, int K, int seg, const double M4[4][4], const double nd[3]) {
  /* solver_gurobi_poly.cpp:485-489: ctrl_pt_x[k]*n1 + ctrl_pt_y[k]*n2 + d - 1 <= 0 */
  for (int k = 0; k < 4; k++) {
    qrow* R = &rows[(*m)++]; R->nnz = 8;
    for (int j = 0; j < 4; j++) { R->idx[j] = VAR(0, seg, j); R->val[j] = nd[0] * M4[j][k]; R->idx[4 + j] = VAR(1, seg, j); R->val[4 + j] = nd[1] * M4[j][k]; }
    R->rhs = 1.0 - nd[2];
  }
}

int orc_optimize(const orc_params* par, int K, const double coeff_init[3][NEP_MAX_POL][4],
                 int n_obst, const orc_polys* hulls, const orc_polys* statics,
                 const orc_ent* ent, int override_n, const int* override_seg,
                 const double (*override_nd)[3], orc_result* out) {
  const double T = par->T_span, wgt = par->weight;
  const int n = 12 * K;
  double M4[4][4], V3[3][3];
  pos_inv_T(T, M4); vel_inv321_T(T, V3);
  const double tp[4] = {T * T * T, T * T, T, 1.0};       /* q_p_term :126 */
  const double qv[4] = {3 * (T * T), 2 * T, 1.0, 0.0};   /* q_v_term :128 */
  const double qa[4] = {6 * T, 2.0, 0.0, 0.0};           /* q_a_term :129 */
  memset(out, 0, sizeof(*out));
  /* setInitTrajectory :226-243 */
  double final_pos[3], ctrl[NEP_MAX_POL][4][2];
  for (int ax = 0; ax < 3; ax++) { const double* c = coeff_init[ax][K - 1]; final_pos[ax] = ((tp[0] * c[0] + tp[1] * c[1]) + tp[2] * c[2]) + tp[3] * c[3]; }
  for (int i = 0; i < K; i++) for (int ax = 0; ax < 2; ax++) { double Q[4]; orc_pos_ctrl_pts(coeff_init[ax][i], T, Q); for (int k = 0; k < 4; k++) ctrl[i][k][ax] = Q[k]; }
  double long_length = sqrt((par->maxs[0] - par->mins[0]) * (par->maxs[0] - par->mins[0]) + (par->maxs[1] - par->mins[1]) * (par->maxs[1] - par->mins[1])); /* :173 */

  /* ---- rows (addConstraints :385-710) ---- */
  int n_st = statics ? statics->n : 0;
  int cap = 48 * K + 4 * (K * (n_obst + par->num_agents + n_st + par->num_agents * (NEP_MAX_BEND + 1)) + (override_n > 0 ? override_n : 0)) + 16;
  qrow* rows = (qrow*)calloc((size_t)cap, sizeof(qrow));
  int m = 0, nl = 0;
  for (int i = 0; i < K; i++) {
    for (int ax = 0; ax < 3; ax++) { /* :437-471 */
      for (int k = 0; k < 4; k++) {
        qrow* R = &rows[m++]; R->nnz = 4; for (int j = 0; j < 4; j++) { R->idx[j] = VAR(ax, i, j); R->val[j] = M4[j][k]; } R->rhs = par->maxs[ax];
        R = &rows[m++]; R->nnz = 4; for (int j = 0; j < 4; j++) { R->idx[j] = VAR(ax, i, j); R->val[j] = -M4[j][k]; } R->rhs = -par->mins[ax];
      }
      for (int k = 0; k < 3; k++) {
        qrow* R = &rows[m++]; R->nnz = 3; for (int j = 0; j < 3; j++) { R->idx[j] = VAR(ax, i, j); R->val[j] = V3[j][k]; } R->rhs = par->v_max;
        R = &rows[m++]; R->nnz = 3; for (int j = 0; j < 3; j++) { R->idx[j] = VAR(ax, i, j); R->val[j] = -V3[j][k]; } R->rhs = par->v_max;
      }
      qrow* R = &rows[m++]; R->nnz = 2; R->idx[0] = VAR(ax, i, 0); R->val[0] = T * 6; R->idx[1] = VAR(ax, i, 1); R->val[1] = 2; R->rhs = par->a_max;
      R = &rows[m++]; R->nnz = 2; R->idx[0] = VAR(ax, i, 0); R->val[0] = -(T * 6); R->idx[1] = VAR(ax, i, 1); R->val[1] = -2; R->rhs = par->a_max;
    }
    if (override_n >= 0) {
      for (int l = 0; l < override_n; l++) if (override_seg[l] == i) {
        out->line_seg[nl] = i; memcpy(out->line_nd[nl], override_nd[l], sizeof(double) * 3); nl++;
        add_line_rows(rows, &m, K, i, M4, override_nd[l]);
      }
      continue;
    }
    double B4[4][2]; for (int k = 0; k < 4; k++) { B4[k][0] = ctrl[i][k][0]; B4[k][1] = ctrl[i][k][1]; }
    g_cur_seg = i;
    /* inter-agent :477-495 */
    for (int j = 0; j < n_obst; j++) {
      int pi = j * par->num_pol + i; int o = hulls->off[pi], nv = hulls->off[pi + 1] - o; double nd[3];
      out->n_lp++;
      if (nv > 0 && orc_separator_ordered(nv, (const double(*)[2])(hulls->xy + 2 * o), 4, (const double(*)[2])B4, nd)) {
        out->line_seg[nl] = i; memcpy(out->line_nd[nl], nd, sizeof(nd)); nl++; add_line_rows(rows, &m, K, i, M4, nd);
      } else out->n_lp_failed++;
    }
    /* bases :521-553 */
    const double base_radius = 0.7;
    for (int j = 0; j < par->num_agents; j++) {
      int close_to_base = 0;
      for (int k = 0; k < 4; k++) { double dx = B4[k][0] - par->pb[2 * j], dy = B4[k][1] - par->pb[2 * j + 1]; if (sqrt(dx * dx + dy * dy) < base_radius * 3) { close_to_base = 1; break; } }
      if (!close_to_base) continue;
      double bx = par->pb[2 * j], by = par->pb[2 * j + 1];
      double bh[4][2] = {{bx + base_radius, by + base_radius}, {bx + base_radius, by - base_radius}, {bx - base_radius, by + base_radius}, {bx - base_radius, by - base_radius}};
      double nd[3]; out->n_lp++;
      if (orc_separator(4, (const double(*)[2])bh, 4, (const double(*)[2])B4, nd)) { out->line_seg[nl] = i; memcpy(out->line_nd[nl], nd, sizeof(nd)); nl++; add_line_rows(rows, &m, K, i, M4, nd); }
      else out->n_lp_failed++;
    }
    /* static obstacles :556-615 */
    for (int j = 0; j < n_st; j++) {
      int o = statics->off[j], nv = statics->off[j + 1] - o; const double(*S)[2] = (const double(*)[2])(statics->xy + 2 * o);
      if (nv <= 0) continue;
      int close_s = 0;
      double dx = B4[0][0] - S[0][0], dy = B4[0][1] - S[0][1]; double dist = sqrt(dx * dx + dy * dy);
      for (int k = 0; k < 3; k++) { double ex = B4[k + 1][0] - B4[k][0], ey = B4[k + 1][1] - B4[k][1]; dist -= sqrt(ex * ex + ey * ey); if (dist < 0) { close_s = 1; break; } }
      for (int k = 0; k < nv - 1; k++) { double ex = S[k + 1][0] - S[k][0], ey = S[k + 1][1] - S[k][1]; dist -= sqrt(ex * ex + ey * ey); if (dist < 0) { close_s = 1; break; } }
      if (!close_s) continue;
      double nd[3]; out->n_lp++;
      if (orc_separator_ordered(nv, S, 4, (const double(*)[2])B4, nd)) { out->line_seg[nl] = i; memcpy(out->line_nd[nl], nd, sizeof(nd)); nl++; add_line_rows(rows, &m, K, i, M4, nd); }
      else out->n_lp_failed++;
    }
    /* entanglement :620-642, 715-764 */
    if (ent && ent->enabled) {
      double hulldist = 0;
      for (int k = 0; k < 3; k++) { double ex = B4[k + 1][0] - B4[k][0], ey = B4[k + 1][1] - B4[k][1]; hulldist += sqrt(ex * ex + ey * ey); }
      for (int j = 0; j < par->num_agents; j++) {
        if (j == par->id - 1) continue;
        int case_id = ent->case_id[i * par->num_agents + j];
        if (case_id == 0) continue;
        int nb = ent->bend_off[j + 1] - ent->bend_off[j]; const double(*bp)[2] = (const double(*)[2])(ent->bend_xy + 2 * ent->bend_off[j]);
        int hp = j * par->num_pol + i; int ho = ent->hulls_noinfl->off[hp], hn = ent->hulls_noinfl->off[hp + 1] - ho;
        for (int k = 1; k < nb + 1; k++) {
          if (k == case_id) continue;
          double pA[2], pB[2];
          if (k == 1) {
            if (hn <= 0) continue; const double* h0 = ent->hulls_noinfl->xy + 2 * ho;
            pA[0] = (1 - long_length) * bp[nb - 1][0] + long_length * h0[0]; pA[1] = (1 - long_length) * bp[nb - 1][1] + long_length * h0[1];
            pB[0] = h0[0]; pB[1] = h0[1];
          } else if (k > 1 && k <= nb) { pA[0] = bp[k - 2][0]; pA[1] = bp[k - 2][1]; pB[0] = bp[k - 1][0]; pB[1] = bp[k - 1][1]; }
          else continue;
          double ax_ = pA[0] - B4[0][0], ay_ = pA[1] - B4[0][1], bx_ = pB[0] - B4[0][0], by_ = pB[1] - B4[0][1];
          if (sqrt(ax_ * ax_ + ay_ * ay_) - hulldist > 0 && sqrt(bx_ * bx_ + by_ * by_) - hulldist > 0) continue; /* :743-745 */
          double A2[2][2] = {{pA[0], pA[1]}, {pB[0], pB[1]}}; double nd[3]; out->n_lp++;
          if (orc_separator(2, (const double(*)[2])A2, 4, (const double(*)[2])B4, nd)) { out->line_seg[nl] = i; memcpy(out->line_nd[nl], nd, sizeof(nd)); nl++; add_line_rows(rows, &m, K, i, M4, nd); }
          else out->n_lp_failed++;
        }
      }
    }
  }
  out->n_lines = nl; out->n_rows = m;

  /* ---- equalities ---- */
  int p_full = 9 * K + 6;
  double* E = (double*)calloc((size_t)p_full * n, sizeof(double)); double* e = (double*)calloc((size_t)p_full, sizeof(double));
  int p = 0;
  for (int k1 = 1; k1 < 4; k1++) for (int ax = 0; ax < 3; ax++) { E[p * n + VAR(ax, 0, k1)] = 1.0; e[p] = coeff_init[ax][0][k1]; p++; } /* :390-396 */
  for (int i = 0; i < K - 1; i++) for (int ax = 0; ax < 3; ax++) { /* :400-425 */
    for (int k = 0; k < 4; k++) E[p * n + VAR(ax, i, k)] = tp[k]; E[p * n + VAR(ax, i + 1, 3)] = -1.0; p++;
    for (int k = 0; k < 3; k++) E[p * n + VAR(ax, i, k)] = qv[k]; E[p * n + VAR(ax, i + 1, 2)] = -1.0; p++;
    for (int k = 0; k < 2; k++) E[p * n + VAR(ax, i, k)] = qa[k]; E[p * n + VAR(ax, i + 1, 1)] = -2.0; p++;
  }
  int p_noterm = p;
  for (int ax = 0; ax < 3; ax++) { /* :659-678 */
    for (int k = 0; k < 3; k++) E[p * n + VAR(ax, K - 1, k)] = qv[k]; p++;
    for (int k = 0; k < 2; k++) E[p * n + VAR(ax, K - 1, k)] = qa[k]; p++;
  }
  /* ---- objective :322-383 ---- */
  double* P = (double*)calloc((size_t)n * n, sizeof(double)); double* q = (double*)calloc((size_t)n, sizeof(double)); double c0 = 0;
  double q_jerk = 36 * T;
  for (int i = 0; i < K; i++) for (int ax = 0; ax < 3; ax++) P[VAR(ax, i, 0) * n + VAR(ax, i, 0)] += 2.0 * q_jerk;
  for (int ax = 0; ax < 3; ax++) {
    for (int k1 = 0; k1 < 4; k1++) { for (int k2 = 0; k2 < 4; k2++) P[VAR(ax, K - 1, k1) * n + VAR(ax, K - 1, k2)] += 2.0 * wgt * (tp[k1] * tp[k2]); q[VAR(ax, K - 1, k1)] += -2.0 * wgt * tp[k1] * final_pos[ax]; }
    c0 += wgt * final_pos[ax] * final_pos[ax];
  }
  /* ---- terminal ball :680-702 ---- */
  double* C = (double*)calloc((size_t)n * n, sizeof(double)); double* cq = (double*)calloc((size_t)n, sizeof(double)); double cc = -0.10 * 0.10;
  double dinit = 0; for (int ax = 0; ax < 3; ax++) { double d = coeff_init[ax][0][3] - final_pos[ax]; dinit += d * d; }
  int has_qc = sqrt(dinit) < 1.0;
  for (int ax = 0; ax < 3; ax++) { for (int k1 = 0; k1 < 4; k1++) { for (int k2 = 0; k2 < 4; k2++) C[VAR(ax, K - 1, k1) * n + VAR(ax, K - 1, k2)] += tp[k1] * tp[k2]; cq[VAR(ax, K - 1, k1)] += -tp[k1] * final_pos[ax]; } cc += final_pos[ax] * final_pos[ax]; }
  out->qc_active = has_qc;

  double* th = (double*)malloc(sizeof(double) * n);
  for (int ax = 0; ax < 3; ax++) for (int i = 0; i < K; i++) for (int j = 0; j < 4; j++) th[VAR(ax, i, j)] = coeff_init[ax][i][j];
  qp_t Q; Q.n = n; Q.p = p; Q.m = m; Q.P = P; Q.q = q; Q.c0 = c0; Q.E = E; Q.e = e; Q.rows = rows; Q.has_qc = has_qc; Q.C = C; Q.cq = cq; Q.cc = cc;
  int it1 = 0, it2 = 0;
  int fail = qp_solve(&Q, th, &it1);
  out->iters_first = it1; out->iters = it1; out->status = NEP_OK;
  if (fail) { /* :838-861 remove terminal v/a rows, add them to the cost */
    for (int ax = 0; ax < 3; ax++) for (int k1 = 0; k1 < 3; k1++) for (int k2 = 0; k2 < 3; k2++) P[VAR(ax, K - 1, k1) * n + VAR(ax, K - 1, k2)] += 2.0 * wgt * (qv[k1] * qv[k2]);
    for (int ax = 0; ax < 3; ax++) for (int k1 = 0; k1 < 2; k1++) for (int k2 = 0; k2 < 2; k2++) P[VAR(ax, K - 1, k1) * n + VAR(ax, K - 1, k2)] += 2.0 * wgt * (qa[k1] * qa[k2]);
    Q.p = p_noterm;
    for (int ax = 0; ax < 3; ax++) for (int i = 0; i < K; i++) for (int j = 0; j < 4; j++) th[VAR(ax, i, j)] = coeff_init[ax][i][j];
    fail = qp_solve(&Q, th, &it2);
    out->iters = it2; out->status = fail ? NEP_FAILED : NEP_RELAXED;
  }
  if (fail) { /* :856-859 */
    for (int ax = 0; ax < 3; ax++) for (int i = 0; i < K; i++) for (int j = 0; j < 4; j++) out->coeff[ax][i][j] = coeff_init[ax][i][j];
  } else {
    for (int ax = 0; ax < 3; ax++) for (int i = 0; i < K; i++) for (int j = 0; j < 4; j++) out->coeff[ax][i][j] = th[VAR(ax, i, j)];
    double dx = coeff_init[0][0][3] - final_pos[0], dy = coeff_init[1][0][3] - final_pos[1];
    if (sqrt(dx * dx + dy * dy) < 1.0) for (int i = 0; i < K; i++) for (int j = 0; j < 4; j++) out->coeff[2][i][j] = coeff_init[2][i][j]; /* :879-880 */
    out->objective = qp_obj(&Q, th); /* :882 */
  }
  free(th); free(C); free(cq); free(P); free(q); free(E); free(e); free(rows);
  return out->status;
}

/* ------------------------------------------------------------------------------------------ */
/* generatePwpOut sampling (:911-934)                                                          */
/* ------------------------------------------------------------------------------------------ */
int orc_sample(int K, const double coeff[3][NEP_MAX_POL][4], double T_span, double dc,
               double* states, int cap) {
  double _t = 0; int i = 0, ns = 0;
  while (i < K && ns < cap) {
    double dt = _t - i * T_span;
    double* st = states + (size_t)ns * NEP_STATE_DOUBLES;
    for (int ax = 0; ax < 3; ax++) {
      const double* c = coeff[ax][i];
      st[ax] = ((c[0] * (dt * dt * dt) + c[1] * (dt * dt)) + c[2] * dt) + c[3];
      st[3 + ax] = (c[0] * (3 * dt * dt) + c[1] * (2 * dt)) + c[2];
      st[6 + ax] = c[0] * (6 * dt) + c[1] * 2;
      st[9 + ax] = c[0] * 6;
    }
    ns++;
    _t += dc;
    if (_t > (i + 1) * T_span) i++;
  }
  return ns;
}

/* ------------------------------------------------------------------------------------------ */
/* Whole replan from committed-trajectory records (neptune.cpp:1422-1435, 1514-1519)           */
/* ------------------------------------------------------------------------------------------ */
int orc_replan(const orc_params* par, double drone_radius, int n_rec, const nep_traj_rec* recs,
               const nep_guess* guess, const orc_polys* statics, const int* case_id,
               orc_result* out, double* hull_xy_out, int* hull_nv_out) {
  const int np = par->num_pol;
  int* off = (int*)calloc((size_t)n_rec * np + 1, sizeof(int));
  double* xy = (double*)calloc((size_t)n_rec * np * NEP_HULL_MAX_V * 2, sizeof(double));
  int* off0 = (int*)calloc((size_t)par->num_agents * np + 1, sizeof(int));
  double* xy0 = (double*)calloc((size_t)par->num_agents * np * NEP_HULL_MAX_V * 2, sizeof(double));
  int* boff = (int*)calloc((size_t)par->num_agents + 1, sizeof(int));
  double* bxy = (double*)calloc((size_t)par->num_agents * NEP_MAX_BEND * 2, sizeof(double));
  int n_obst = 0, nvtot = 0, nv0tot = 0;
  /* neptune.cpp:235-263: ids 1..num_agents in order, own id and unknown ids skipped */
  for (int id = 1; id <= par->num_agents; id++) {
    const nep_traj_rec* r = NULL;
    for (int k = 0; k < n_rec; k++) if (recs[k].id == id && recs[k].valid && recs[k].is_agent) { r = &recs[k]; break; }
    int present = (id != par->id) && r != NULL;
    for (int i = 0; i < np; i++) {
      int nv = 0, nv0 = 0;
      if (present) {
        double delta[2] = {r->bbox[0] / 2.0 + drone_radius, r->bbox[1] / 2.0 + drone_radius}; /* neptune.cpp:340 */
        double t0 = guess->t_start + i * par->T_span, t1 = guess->t_start + (i + 1) * par->T_span; /* :273-280 with deltaT = T_span */
        double h[NEP_HULL_MAX_V][2], h0[NEP_HULL_MAX_V][2];
        orc_hull_of_interval(&r->pwp, t0, t1, par->T_span, delta, h, &nv, h0, &nv0);
        memcpy(xy + 2 * nvtot, h, sizeof(double) * 2 * nv); memcpy(xy0 + 2 * nv0tot, h0, sizeof(double) * 2 * nv0);
        if (hull_xy_out) { memcpy(hull_xy_out + ((size_t)(n_obst * np + i) * NEP_HULL_MAX_V) * 2, h, sizeof(double) * 2 * nv); hull_nv_out[n_obst * np + i] = nv; }
        nvtot += nv; off[n_obst * np + i + 1] = nvtot;
      }
      nv0tot += nv0; off0[(id - 1) * np + i + 1] = nv0tot;
    }
    if (present) n_obst++;
    int nb = (r && id != par->id) ? r->n_bend : 0; if (nb > NEP_MAX_BEND) nb = NEP_MAX_BEND;
    for (int b = 0; b < nb; b++) { bxy[2 * (boff[id - 1] + b)] = r->bend[b][0]; bxy[2 * (boff[id - 1] + b) + 1] = r->bend[b][1]; }
    boff[id] = boff[id - 1] + nb;
  }
  orc_polys hulls = {n_obst * np, off, xy};
  orc_polys hulls0 = {par->num_agents * np, off0, xy0};
  orc_ent ent = {case_id != NULL, case_id, boff, bxy, &hulls0};
  int st = orc_optimize(par, guess->K, guess->coeff, n_obst, &hulls, statics, case_id ? &ent : NULL, -1, NULL, NULL, out);
  free(off); free(xy); free(off0); free(xy0); free(boff); free(bxy);
  return st;
}


/* ------------------------------------------------------------------------------------------ */
/* post-solve safety check                                                */
/* ------------------------------------------------------------------------------------------ */
/* gjk::collision (neptune/src/gjk.cpp:76-149) with its helpers (:17-68), vertices1 = V1, vertices2
 * = V2.  The reference loops without a bound; a cap of 64 simplex updates returns "no collision"
 * (never reached on the fixtures). */
static int gjk_furthest(int n, const double (*V)[2], double dx, double dy) {
  double mx = dx * V[0][0] + dy * V[0][1]; int idx = 0;
  for (int i = 1; i < n; i++) { double p = dx * V[i][0] + dy * V[i][1]; if (p > mx) { mx = p; idx = i; } }
  return idx;
}
static void gjk_support(int n1, const double (*V1)[2], int n2, const double (*V2)[2], double dx, double dy, double out[2]) {
  int i = gjk_furthest(n1, V1, dx, dy), j = gjk_furthest(n2, V2, -dx, -dy);
  out[0] = V1[i][0] - V2[j][0]; out[1] = V1[i][1] - V2[j][1];
}
static void gjk_triple(const double a[2], const double b[2], const double c[2], double out[2]) { /* b*(a.c) - a*(b.c) */
  double ac = a[0] * c[0] + a[1] * c[1], bc = b[0] * c[0] + b[1] * c[1];
  out[0] = b[0] * ac - a[0] * bc; out[1] = b[1] * ac - a[1] * bc;
}
int orc_gjk_collision(int n1, const double (*V1)[2], int n2, const double (*V2)[2]) {
  if (n1 <= 0 || n2 <= 0) return 0;
  double p1[2] = {0, 0}, p2[2] = {0, 0};
  for (int i = 0; i < n1; i++) { p1[0] += V1[i][0]; p1[1] += V1[i][1]; }
  for (int i = 0; i < n2; i++) { p2[0] += V2[i][0]; p2[1] += V2[i][1]; }
  p1[0] /= n1; p1[1] /= n1; p2[0] /= n2; p2[1] /= n2;
  double d[2] = {p1[0] - p2[0], p1[1] - p2[1]};
  if (d[0] == 0 && d[1] == 0) d[0] = 1.0;
  double simplex[3][2], a[2], b[2], c[2], ao[2], ab[2], ac[2], abperp[2], acperp[2];
  int index = 0;
  gjk_support(n1, V1, n2, V2, d[0], d[1], simplex[0]);
  a[0] = simplex[0][0]; a[1] = simplex[0][1];
  if (a[0] * d[0] + a[1] * d[1] <= 0) return 0;
  d[0] = -a[0]; d[1] = -a[1];
  for (int iter = 0; iter < 64; iter++) {
    ++index;
    gjk_support(n1, V1, n2, V2, d[0], d[1], simplex[index]);
    a[0] = simplex[index][0]; a[1] = simplex[index][1];
    if (a[0] * d[0] + a[1] * d[1] <= 0) return 0;
    ao[0] = -a[0]; ao[1] = -a[1];
    if (index < 2) {
      b[0] = simplex[0][0]; b[1] = simplex[0][1];
      ab[0] = b[0] - a[0]; ab[1] = b[1] - a[1];
      gjk_triple(ab, ao, ab, d);
      if (sqrt(d[0] * d[0] + d[1] * d[1]) == 0) { d[0] = ab[1]; d[1] = -ab[0]; }
      continue;
    }
    b[0] = simplex[1][0]; b[1] = simplex[1][1]; c[0] = simplex[0][0]; c[1] = simplex[0][1];
    ab[0] = b[0] - a[0]; ab[1] = b[1] - a[1]; ac[0] = c[0] - a[0]; ac[1] = c[1] - a[1];
    gjk_triple(ab, ac, ac, acperp);
    if (acperp[0] * ao[0] + acperp[1] * ao[1] >= 0) { d[0] = acperp[0]; d[1] = acperp[1]; }
    else {
      gjk_triple(ac, ab, ab, abperp);
      if (abperp[0] * ao[0] + abperp[1] * ao[1] < 0) return 1;
      simplex[0][0] = simplex[1][0]; simplex[0][1] = simplex[1][1];
      d[0] = abperp[0]; d[1] = abperp[1];
    }
    simplex[1][0] = simplex[2][0]; simplex[1][1] = simplex[2][1];
    --index;
  }
  return 0;
}

/* Neptune::trajsAndPwpAreInCollision2d (neptune.cpp:767-806): my new trajectory `mine` against the
 * inflated interval hulls of `other`. */
int orc_trajs_and_pwp_in_collision(const nep_traj_rec* other, const nep_pwp* mine, double T_span, double drone_radius) {
  int n = mine->n_seg;
  if (n <= 0) return 0;
  double t_start = mine->times[0], t_end = mine->times[n];
  double deltaT = (t_end - t_start) / n;
  if (fabs(deltaT - T_span) > 0.1) return 1; /* :773-781 */
  double delta[2] = {other->bbox[0] / 2.0 + drone_radius, other->bbox[1] / 2.0 + drone_radius};
  for (int i = 0; i < n; i++) {
    double A4[4][2];
    for (int ax = 0; ax < 2; ax++) { double Q[4]; orc_pos_ctrl_pts(mine->coeff[ax][i], T_span, Q); for (int k = 0; k < 4; k++) A4[k][ax] = Q[k]; } /* :789 */
    double hull[NEP_HULL_MAX_V][2], hull0[NEP_HULL_MAX_V][2]; int nv, nv0;
    orc_hull_of_interval(&other->pwp, t_start + deltaT * i, t_start + deltaT * (i + 1), T_span, delta, hull, &nv, hull0, &nv0); /* :792 */
    if (orc_gjk_collision(nv, (const double(*)[2])hull, 4, (const double(*)[2])A4)) return 1; /* :794 */
  }
  return 0;
}

/* Same test on the bulk-synchronous round's interval grid t_start + i*T_span (the grid the replan
 * itself used; (t_end - t_start)/n of :770 equals T_span up to rounding, but rounding would move
 * interval ends across knots and change which segments a hull covers). */
static int trajs_collide_on_grid(const nep_traj_rec* other, const nep_pwp* mine, double t_start, double T_span, double drone_radius) {
  int n = mine->n_seg;
  double delta[2] = {other->bbox[0] / 2.0 + drone_radius, other->bbox[1] / 2.0 + drone_radius};
  for (int i = 0; i < n; i++) {
    double A4[4][2];
    for (int ax = 0; ax < 2; ax++) { double Q[4]; orc_pos_ctrl_pts(mine->coeff[ax][i], T_span, Q); for (int k = 0; k < 4; k++) A4[k][ax] = Q[k]; }
    double hull[NEP_HULL_MAX_V][2], hull0[NEP_HULL_MAX_V][2]; int nv, nv0;
    orc_hull_of_interval(&other->pwp, t_start + i * T_span, t_start + (i + 1) * T_span, T_span, delta, hull, &nv, hull0, &nv0);
    if (orc_gjk_collision(nv, (const double(*)[2])hull, 4, (const double(*)[2])A4)) return 1;
  }
  return 0;
}

/* Bulk-synchronous form of safetyCheckAfterReplan (neptune.cpp:719-765): every other agent's new
 * trajectory counts as "received while optimizing".  conflict[a][j] = agent a's new trajectory
 * collides with the hulls of j's new trajectory.  Deterministic resolution (ours; the reference is
 * asynchronous and simply fails the replan of whoever checks second): agents are visited by id; an
 * agent keeps its new trajectory unless it conflicts (either direction) with an already accepted
 * lower id, in which case it keeps executing its previous plan (neptune_ros.cpp:651-663). */
void orc_safety_resolve(int n, const nep_traj_rec* fresh, double t_start, double T_span, double drone_radius, unsigned char* conflict, int* accept) {
  for (int a = 0; a < n; a++) for (int j = 0; j < n; j++) {
    int c = 0;
    if (a != j && fresh[a].valid && fresh[j].valid && fresh[j].is_agent) c = trajs_collide_on_grid(&fresh[j], &fresh[a].pwp, t_start, T_span, drone_radius);
    conflict[a * n + j] = (unsigned char)c;
  }
  for (int a = 0; a < n; a++) {
    int ok = 1;
    for (int j = 0; j < a; j++) if (accept[j] && (conflict[a * n + j] || conflict[j * n + a])) { ok = 0; break; }
    accept[a] = ok;
  }
}

/* The same with nep_batch_set_safety_check_prev: a new trajectory that collides with the previous record
 * of any other agent (on the round's grid) is turned down too. */
void orc_safety_resolve_prev(int n, const nep_traj_rec* prev, const nep_traj_rec* fresh, double t_start, double T_span, double drone_radius,
                             unsigned char* conflict, int* accept) {
  orc_safety_resolve(n, fresh, t_start, T_span, drone_radius, conflict, accept);
  for (int a = 0; a < n; a++) {
    int bad = 0;
    for (int j = 0; j < n && !bad; j++)
      if (j != a && fresh[a].valid && prev[j].valid && prev[j].is_agent) bad = trajs_collide_on_grid(&prev[j], &fresh[a].pwp, t_start, T_span, drone_radius);
    if (bad) accept[a] = -1;
  }
  /* redo the id-ordered pass with the turned-down agents out of it */
  for (int a = 0; a < n; a++) {
    if (accept[a] == -1) { accept[a] = 0; continue; }
    int ok = 1;
    for (int j = 0; j < a; j++) if (accept[j] && (conflict[a * n + j] || conflict[j * n + a])) { ok = 0; break; }
    accept[a] = ok;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* front-end initial guess — the deterministic beam rule of               */
/* include/neptune_frontend.h over KinodynamicSearch's lattice, pruning and cost rules          */
/* (kinodynamic_search.cpp:190-227 setUp, :1045-1228 / :1240-1385 expansion, :1514-1553          */
/* collision, :1629-1827 run, :521-553 recoverPwpOut).  The HIP kernel must match bit for bit.   */
/* ------------------------------------------------------------------------------------------ */
typedef struct fe_node {
  double end[6];            /* px py vx vy ax ay (:1079-1086) */
  double cx[4], cy[4];      /* the segment that led here */
  double g, f, dist;
  long vx, vy;              /* voxel */
  int parent;               /* rank in the previous depth's beam, -1 = root */
} fe_node;

static int fe_aabb_overlap(int n, const double (*V)[2], const double Qx[4], const double Qy[4]) {
  double ax0 = V[0][0], ax1 = V[0][0], ay0 = V[0][1], ay1 = V[0][1];
  for (int i = 1; i < n; i++) {
    if (V[i][0] < ax0) ax0 = V[i][0]; if (V[i][0] > ax1) ax1 = V[i][0];
    if (V[i][1] < ay0) ay0 = V[i][1]; if (V[i][1] > ay1) ay1 = V[i][1];
  }
  double bx0 = Qx[0], bx1 = Qx[0], by0 = Qy[0], by1 = Qy[0];
  for (int i = 1; i < 4; i++) {
    if (Qx[i] < bx0) bx0 = Qx[i]; if (Qx[i] > bx1) bx1 = Qx[i];
    if (Qy[i] < by0) by0 = Qy[i]; if (Qy[i] > by1) by1 = Qy[i];
  }
  return !(ax1 < bx0 || bx1 < ax0 || ay1 < by0 || by1 < ay0);
}

/* one lattice child of `par` (end state, g); returns 0 when a kinodynamic test prunes it */
static int fe_child(const orc_fe_cfg* c, const double init_end[6], double par_g, int first, int jx, int jy, const double goal[2], fe_node* out) {
  const double tau = c->T_span, j_min = -c->j_max, j_max = c->j_max, v_max = c->v_max, v_min = -c->v_max, a_max = c->a_max, a_min = -c->a_max;
  const double delta = (j_max - j_min) / (c->num_samples - 1);
  const double ji[2] = {j_min + jx * delta, j_min + jy * delta};
  double e[6];
  for (int ax = 0; ax < 2; ax++) {
    const double p = init_end[ax], v = init_end[2 + ax], a = init_end[4 + ax], j = ji[ax];
    e[ax] = ((p + v * tau) + ((a * tau) * tau) / 2) + (((j * tau) * tau) * tau) / 6;
    e[2 + ax] = (v + a * tau) + ((j * tau) * tau) / 2;
    e[4 + ax] = a + j * tau;
  }
  double n2 = 0;
  for (int i = 0; i < 6; i++) n2 += (e[i] - init_end[i]) * (e[i] - init_end[i]);
  if (n2 < 0.00001 * 0.00001) return 0;   /* |end - start| < 1e-5, on the squares */
  if (e[5] > a_max || e[5] < a_min || e[4] > a_max || e[4] < a_min) return 0;
  for (int ax = 0; ax < 2; ax++) {
    double* co = ax == 0 ? out->cx : out->cy;
    co[0] = ji[ax] / 6; co[1] = init_end[4 + ax] / 2; co[2] = init_end[2 + ax]; co[3] = init_end[ax];
  }
  double Qx[4], Qy[4];
  orc_pos_ctrl_pts(out->cx, tau, Qx); orc_pos_ctrl_pts(out->cy, tau, Qy);
  const double bx = c->pb[2 * (c->id - 1)], by = c->pb[2 * (c->id - 1) + 1];
  for (int i = 0; i < 4; i++) {
    if (Qx[i] < c->mins[0] || Qx[i] > c->maxs[0] || Qy[i] < c->mins[1] || Qy[i] > c->maxs[1]) return 0;
    if ((Qx[i] - bx) * (Qx[i] - bx) + (Qy[i] - by) * (Qy[i] - by) > c->cable_length * c->cable_length) return 0;   /* on the squares */
  }
  if (!first) {
    double Vx[3], Vy[3];
    orc_vel_ctrl_pts(out->cx, tau, Vx); orc_vel_ctrl_pts(out->cy, tau, Vy);
    for (int i = 0; i < 3; i++) if (Vx[i] < v_min || Vx[i] > v_max || Vy[i] < v_min || Vy[i] > v_max) return 0;
  }
  for (int ax = 0; ax < 2; ax++) {
    const double a = e[4 + ax], v = e[2 + ax];
    if (a > 0 && v - ((0.5 * a) * a) / j_min > v_max) return 0;
    else if (a < 0 && v - ((0.5 * a) * a) / j_max < v_min) return 0;
  }
  for (int i = 0; i < 6; i++) out->end[i] = e[i];
  const double arc = sqrt((e[0] - init_end[0]) * (e[0] - init_end[0]) + (e[1] - init_end[1]) * (e[1] - init_end[1]));
  out->g = par_g + arc;
  out->dist = sqrt((e[0] - goal[0]) * (e[0] - goal[0]) + (e[1] - goal[1]) * (e[1] - goal[1]));
  out->f = out->g + c->bias * out->dist;
  out->vx = (long)round(e[0] / c->voxel_size); out->vy = (long)round(e[1] / c->voxel_size);
  return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* Entanglement state of a lattice node (enable_entangle_check): KinodynamicSearch::entanglesWithOtherAgents  */
/* (kinodynamic_search.cpp:707-895) with eu::entangleHSigToAddAgentInd (entangle_utils.cpp:1129-1228),        */
/* entangleHSigToAddStatic (:1231-1277), addAlphaBetaToList + breakcondition (:1402-1534, :1608-1647),        */
/* updateBendPts (:1536-1604), getBendPt2d / calculateBetaForCase / getTetherLength (:1649-1743).             */
/* No capacity of its own: like the reference's vectors, a node's list is bounded by the reference's rule only   */
/* (a child is pruned when its list plus the new crossings of a step exceed cap_mult * (num_agents + statics),  */
/* kinodynamic_search.cpp:850-854, :944-948), its bend points by its list, a step's new crossings by what the   */
/* agents' tethers can give.  nep_fe_result.ent_overflow stays 0.  active_cases[i] is the number of list entries */
/* of agent i (every append adds one, every cancellation removes one): it is derived, not stored.               */
/* ------------------------------------------------------------------------------------------ */
typedef struct orc_ent_node { int n_alpha, n_bend; short* id; signed char* cs; double* beta; short* bend; } orc_ent_node;
/* `count` nodes of `cap` entries each in one allocation (free(nodes) releases everything) */
static orc_ent_node* ent_nodes_alloc(size_t count, int cap) {
  const size_t per = ((size_t)cap * (sizeof(double) + 2 * sizeof(short) + 1) + 7) & ~(size_t)7;
  char* m = (char*)calloc(1, count * (sizeof(orc_ent_node) + per) + 16);
  orc_ent_node* nd = (orc_ent_node*)m;
  char* slab = m + ((count * sizeof(orc_ent_node) + 7) & ~(size_t)7);
  for (size_t i = 0; i < count; i++) {
    char* q = slab + i * per;
    nd[i].beta = (double*)q; nd[i].id = (short*)(q + (size_t)cap * 8); nd[i].bend = (short*)(q + (size_t)cap * 10); nd[i].cs = (signed char*)(q + (size_t)cap * 12);
  }
  return nd;
}
static void ent_node_copy(orc_ent_node* d, const orc_ent_node* s) {
  d->n_alpha = s->n_alpha; d->n_bend = s->n_bend;
  for (int i = 0; i < s->n_alpha; i++) { d->id[i] = s->id[i]; d->cs[i] = s->cs[i]; d->beta[i] = s->beta[i]; }
  for (int i = 0; i < s->n_bend; i++) d->bend[i] = s->bend[i];
}
static void ent_node_from_fixed(orc_ent_node* d, const nep_fe_ent_state* s) {      /* the state at point A, as the C ABI carries it */
  d->n_alpha = s->n_alpha; d->n_bend = s->n_bend;
  for (int i = 0; i < s->n_alpha && i < NEP_FE_ENT_CAP; i++) { d->id[i] = s->id[i]; d->cs[i] = s->cs[i]; d->beta[i] = s->beta[i]; }
  for (int i = 0; i < s->n_bend && i < NEP_MAX_BEND; i++) d->bend[i] = s->bend[i];
}

typedef struct ent_ctx {
  int N, S, own, num_pol, ns; double T_span, cable;
  const double* pb; const double* srep; const double* slong; const double* sampled; const int* present; const int* bend_n; const double* bend_xy;
} ent_ctx;
typedef struct { double x, y; } ev2;
typedef struct { short* id; signed char* cs; int n, cap, overflow; } ent_add;      /* (overflow: cannot happen, ent_add_cap) */
/* entries a node of this problem can ever hold (cap_mult = 3 for entangleCheckGivenPwp), and new crossings one step can add */
static int ent_node_cap(const ent_ctx* c, int cap_mult) { int k = (c->N + c->S) * cap_mult; if (k < NEP_FE_ENT_CAP) k = NEP_FE_ENT_CAP; return k + 4; }
static int ent_add_cap(const ent_ctx* c) { return c->N * (2 * NEP_MAX_BEND + 2) + c->S + 4; }      /* every tether segment of every agent twice (crossing + base sweep), every static once */

static ev2 ent_pb(const ent_ctx* c, int j) { ev2 r = {c->pb[2 * j], c->pb[2 * j + 1]}; return r; }
static ev2 ent_srep(const ent_ctx* c, int s, int col) { ev2 r = {c->srep[(s * 2 + col) * 2], c->srep[(s * 2 + col) * 2 + 1]}; return r; }
static ev2 ent_bendpt(const ent_ctx* c, int j, int k) { ev2 r = {c->bend_xy[((size_t)j * NEP_MAX_BEND + k) * 2], c->bend_xy[((size_t)j * NEP_MAX_BEND + k) * 2 + 1]}; return r; }
static ev2 ent_sampled(const ent_ctx* c, int i, int interval, int col) {
  const double* q = c->sampled + (((size_t)i * c->num_pol + interval) * (c->ns + 1) + col) * 2; ev2 r = {q[0], q[1]}; return r;
}
static double ent_wedge(ev2 a, ev2 b, ev2 cc) { return (b.x - a.x) * (cc.y - a.y) - (cc.x - a.x) * (b.y - a.y); }
static double ent_wedge2(ev2 a, ev2 b, ev2 cc, ev2* ab, ev2* ac) { ab->x = b.x - a.x; ab->y = b.y - a.y; ac->x = cc.x - a.x; ac->y = cc.y - a.y; return ab->x * ac->y - ac->x * ab->y; }
static double ent_ratio(ev2 u, ev2 v) { return fabs(u.y * v.y) > fabs(u.x * v.x) ? u.y / v.y : u.x / v.x; }
static double ent_dist(ev2 a, ev2 b) { return sqrt((a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y)); }
static void ent_push(ent_add* a, int id, int cs) { if (a->n < a->cap) { a->id[a->n] = (short)id; a->cs[a->n] = (signed char)cs; a->n++; } else a->overflow = 1; }

static void ent_cross_agent(ent_add* add, ev2 pk, ev2 pk1, ev2 pik, ev2 pik1, ev2 pb_self, const ent_ctx* c, int i, int agent_id) {
  const int nb = c->bend_n[i];
  int base_addition = 0;
  for (int k = 0; k < nb; k++) {
    const int last = k == nb - 1;
    const ev2 bk = ent_bendpt(c, i, k);
    ev2 u, v; double c1, c2;
    if (!last) { const ev2 bn = ent_bendpt(c, i, k + 1); c1 = ent_wedge2(pk, bn, bk, &u, &v); c2 = ent_wedge(pk1, bn, bk); }
    else { c1 = ent_wedge2(pk, pik, bk, &u, &v); c2 = ent_wedge(pk1, pik1, bk); }
    if (last) {   /* the other agent's last tether piece sweeping over OUR base */
      ev2 ub, vb;
      const double f1 = ent_wedge2(pb_self, pik, bk, &ub, &vb), f2 = ent_wedge(pb_self, pik1, bk);
      if (f1 * f2 < 0) {
        const double a = ent_ratio(ub, vb);
        if (a < 0) { }
        else if (a < 1) ent_push(add, agent_id, 1);
        else if (k == 0) ent_push(add, agent_id, 0);
        base_addition = 1;
      }
    }
    if (c1 * c2 < 0) {
      const double a = ent_ratio(u, v);
      if (a < 0) ent_push(add, agent_id, k + 2);
      else if (a < 1 && last) ent_push(add, agent_id, 1);
      else if (a >= 1 && k == 0) ent_push(add, agent_id, 0);
    }
  }
  if (base_addition && add->n >= 2 && add->id[add->n - 1] == add->id[add->n - 2] && add->cs[add->n - 1] == add->cs[add->n - 2]) add->n -= 2;
}
static void ent_cross_static(ent_add* add, ev2 pk, ev2 pk1, const ent_ctx* c) {
  for (int s = 0; s < c->S; s++) {
    const ev2 pik = ent_srep(c, s, 1), pbi = ent_srep(c, s, 0);
    ev2 u, v;
    const double c1 = ent_wedge2(pk, pik, pbi, &u, &v), c2 = ent_wedge(pk1, pik, pbi);
    if (c1 * c2 < 0) {
      const double a = ent_ratio(u, v);
      if (a < 0) { }
      else if (a < 1) ent_push(add, c->N + s + 1, 1);
      else ent_push(add, c->N + s + 1, 0);
    }
  }
}
static ev2 ent_anchor(int id, int cs, const ent_ctx* c) { return id <= c->N ? ent_pb(c, id - 1) : ent_srep(c, id - c->N - 1, cs); }
static ev2 ent_cur_bend(const orc_ent_node* st, ev2 pb_self, const ent_ctx* c) {
  if (st->n_bend == 0) return pb_self;
  const int b = st->bend[st->n_bend - 1];
  const int id = st->id[b], cs = st->cs[b];
  if (id <= c->N && id >= 1) return ent_pb(c, id - 1);
  if (id > c->N) return ent_srep(c, id - c->N - 1, cs);
  { ev2 z = {0, 0}; return z; }
}
static double ent_beta(int id, int cs, ev2 pk, ev2 bp, const ent_ctx* c) { return id <= c->N ? 0.0 : ent_wedge(pk, ent_srep(c, id - c->N - 1, cs), bp); }
static int ent_scan_stops(int t_id, int t_cs, int l_id, int j, int last_bend, const ent_ctx* c) {
  if (t_id <= c->N && t_cs >= 2) return j <= last_bend;
  if (t_id <= c->N) return 0;
  return l_id > c->N || j <= last_bend;
}
static void ent_erase(orc_ent_node* st, int j) {
  for (int k = j; k + 1 < st->n_alpha; k++) { st->id[k] = st->id[k + 1]; st->cs[k] = st->cs[k + 1]; st->beta[k] = st->beta[k + 1]; }
  st->n_alpha--;
}
static int ent_merge(ent_add* add, orc_ent_node* st, ev2 pk, ev2 pb_self, const ent_ctx* c) {
  int again = 1;
  while (again) {
    again = 0;
    const int b = st->n_bend ? st->bend[st->n_bend - 1] : -1;
    for (int i = 0; i < add->n && !again; i++) {
      const int t_id = add->id[i], t_cs = add->cs[i];
      for (int j = st->n_alpha - 1; j >= 0; j--) {
        const int l_id = st->id[j], l_cs = st->cs[j];
        const int agent = t_id <= c->N;
        const int match = (l_id == t_id && l_cs == t_cs) ||
                          (agent && l_id == t_id && t_cs >= c->bend_n[t_id - 1] + 1 && t_cs < l_cs) ||
                          (agent && l_id == t_id && l_cs >= 2 && t_cs >= 2 && abs(t_cs - l_cs) == 1 && j > b);
        if (match) {
          for (int k = i; k + 1 < add->n; k++) { add->id[k] = add->id[k + 1]; add->cs[k] = add->cs[k + 1]; }
          add->n--;
          ent_erase(st, j);
          if (j == b) {
            st->n_bend--;
            const ev2 bp = ent_cur_bend(st, pb_self, c);
            for (int k = j; k < st->n_alpha; k++) st->beta[k] = ent_beta(st->id[k], st->cs[k], pk, bp, c);
          } else if (j < b) {
            st->bend[st->n_bend - 1] = (short)(b - 1);
            for (int k = st->n_bend - 2; k >= 0; k--) { if (st->bend[k] > j) st->bend[k] -= 1; else break; }
          }
          again = 1;
          break;
        }
        if (ent_scan_stops(t_id, t_cs, l_id, j, b, c)) break;
      }
    }
  }
  if (add->n == 0) return 0;
  const ev2 bp = ent_cur_bend(st, pb_self, c);
  for (int i = 0; i < add->n; i++) {
    st->id[st->n_alpha] = add->id[i]; st->cs[st->n_alpha] = add->cs[i];
    st->beta[st->n_alpha] = ent_beta(add->id[i], add->cs[i], pk, bp, c);
    st->n_alpha++;
  }
  return 0;
}
static int ent_update_bends(orc_ent_node* st, ev2 pk1, ev2 pb_self, const ent_ctx* c) {
  const ev2 bp = ent_cur_bend(st, pb_self, c);
  int idx_new = -1;
  const int start = st->n_bend ? st->bend[st->n_bend - 1] : -1;
  for (int i = start + 1; i < st->n_alpha; i++) { const double beta = ent_beta(st->id[i], st->cs[i], pk1, bp, c); if (beta * st->beta[i] < -1e-7) idx_new = i; }
  if (idx_new > -1) {
    st->bend[st->n_bend++] = (short)idx_new;
    const ev2 nb = ent_anchor(st->id[idx_new], st->cs[idx_new], c);
    for (int i = idx_new + 1; i < st->n_alpha; i++) st->beta[i] = ent_beta(st->id[i], st->cs[i], pk1, nb, c);
    return 0;
  }
  while (st->n_bend) {
    const ev2 prev = st->n_bend == 1 ? pb_self : ent_anchor(st->id[st->bend[st->n_bend - 2]], st->cs[st->bend[st->n_bend - 2]], c);
    const int bi = st->bend[st->n_bend - 1];
    const double beta = ent_beta(st->id[bi], st->cs[bi], pk1, prev, c);
    if (beta * st->beta[bi] > 1e-7) {
      for (int k = bi + 1; k < st->n_alpha; k++) st->beta[k] = ent_beta(st->id[k], st->cs[k], pk1, prev, c);
      st->n_bend--;
    } else break;
  }
  return 0;
}
static double ent_tether(const orc_ent_node* st, ev2 from, ev2 pk1, const ent_ctx* c) {
  double len = 0.0;
  for (int q = 0; q < st->n_bend; q++) {
    const int b = st->bend[q]; const int id = st->id[b], cs = st->cs[b];
    ev2 bp; double comp;
    if (id <= c->N) { bp = ent_pb(c, id - 1); comp = 0.0; } else { bp = ent_srep(c, id - c->N - 1, cs); comp = c->slong[(id - c->N - 1) * 2 + cs]; }
    len += ent_dist(bp, from) + 2 * comp;
    from = bp;
  }
  return len + ent_dist(pk1, from);
}
static int ent_count(const short* ids, int n, int id) { int k = 0; for (int i = 0; i < n; i++) k += ids[i] == id; return k; }
/* 0: fine; 1: the reference's function returns true (prune); 2: capacity exceeded (pruned, flagged).  check_tether / cap_mult:
 * entanglesWithOtherAgents (1, 1) and entangleCheckGivenPwp (0, 3: kinodynamic_search.cpp:944-948, 977-983). */
static int ent_propagate(const ent_ctx* c, orc_ent_node* st, const double cxo[4], const double cyo[4], ev2 end, int index, double* arc, int check_tether, int cap_mult) {
  const int ns = c->ns;
  const ev2 pb_self = ent_pb(c, c->own);
  ev2 pk = {cxo[3], cyo[3]}, pk1 = pk;
  const int acap = ent_add_cap(c), ncap = ent_node_cap(c, cap_mult);
  short* abuf = (short*)malloc((size_t)acap * 3 + (size_t)ncap * 2 + 16);
  short* old_id = abuf + acap + ((acap + 1) >> 1) + 2;
  int verdict = 0;
  for (int j = 1; j <= ns && !verdict; j++) {
    ent_add add; add.id = abuf; add.cs = (signed char*)(abuf + acap); add.cap = acap; add.n = 0; add.overflow = 0;
    if (j < ns) {
      const double t = c->T_span * j / ns;
      const double t3 = t * t * t, t2 = t * t;
      pk1.x = ((cxo[0] * t3 + cxo[1] * t2) + cxo[2] * t) + cxo[3] * 1.0; pk1.y = ((cyo[0] * t3 + cyo[1] * t2) + cyo[2] * t) + cyo[3] * 1.0;
    } else pk1 = end;
    *arc += ent_dist(pk1, pk);
    for (int i = 0; i < c->N; i++) {
      if (i == c->own || !c->present[i]) continue;
      ev2 pik, pik1;
      if (index > c->num_pol) { pik = ent_sampled(c, i, c->num_pol - 1, ns); pik1 = pik; }
      else { pik = ent_sampled(c, i, index - 1, j - 1); pik1 = ent_sampled(c, i, index - 1, j); }
      ent_cross_agent(&add, pk, pk1, pik, pik1, pb_self, c, i, i + 1);
    }
    ent_cross_static(&add, pk, pk1, c);
    if (add.overflow) { verdict = 2; break; }      /* (not reachable: ent_add_cap) */
    if (st->n_alpha + add.n > (c->N + c->S) * cap_mult) { verdict = 1; break; }
    const int old_n = st->n_alpha;
    for (int i = 0; i < old_n; i++) old_id[i] = st->id[i];
    ent_merge(&add, st, pk, pb_self, c);
    for (int i = 0; i < st->n_alpha && !verdict; i++) {
      const int id = st->id[i];
      if (id > c->N) continue;
      const int nw = ent_count(st->id, st->n_alpha, id), od = ent_count(old_id, old_n, id);
      if (od < 2 && nw >= 2) verdict = 1;
      if (od >= 2 && nw > od) verdict = 1;
    }
    if (verdict) break;
    ent_update_bends(st, pk1, pb_self, c);
    pk = pk1;
  }
  free(abuf);
  if (verdict) return verdict;
  if (check_tether && ent_tether(st, pb_self, pk1, c) > c->cable) return 1;
  return 0;
}
/* (knot, agent) -> case id of solver_gurobi_poly.cpp:624-631: the last list entry of an agent with exactly one */
static void ent_case_row(const orc_ent_node* st, int N, int* row) {
  for (int j = 0; j < N; j++) row[j] = 0;
  for (int j = 0; j < N; j++) {
    if (ent_count(st->id, st->n_alpha, j + 1) != 1) continue;
    for (int a = 0; a < st->n_alpha; a++) if (st->id[a] == j + 1) row[j] = st->cs[a];
  }
}
/* KinodynamicSearch::getIz (:2006-2014) with power_int (:2016-2031), unsigned arithmetic */
static unsigned ent_iz(const orc_ent_node* st) {
  unsigned iz = 0;
  for (int i = 0; i < st->n_alpha; i++) {
    unsigned base = (unsigned)st->id[i], ex = (unsigned)st->cs[i], r;
    if (ex == 0) r = 1; else if (base < 2) r = base;
    else { r = 1; for (unsigned term = base;; term = term * term) { if (ex % 2 != 0) r *= term; ex /= 2; if (ex == 0) break; } }
    iz += (unsigned)(i + 1) * r;
  }
  return iz;
}

/* test hook: every feasible child's control polygon and collision verdict, [n][10] = depth, id, Q[4][2] ... */
static double* g_fe_dump = 0; static int g_fe_dump_cap = 0, g_fe_dump_n = 0;
void orc_fe_set_dump(double* buf, int cap) { g_fe_dump = buf; g_fe_dump_cap = cap; g_fe_dump_n = 0; }
int orc_fe_dump_count(void) { return g_fe_dump_n; }

static int fe_collides(const orc_fe_cfg* c, const fe_node* nd, int depth, const double* hull_xy, const int* hull_nv, const orc_polys* statics) {
  double Qx[4], Qy[4], Q[4][2];
  orc_pos_ctrl_pts(nd->cx, c->T_span, Qx); orc_pos_ctrl_pts(nd->cy, c->T_span, Qy);
  for (int i = 0; i < 4; i++) { Q[i][0] = Qx[i]; Q[i][1] = Qy[i]; }
  int idx = depth > c->num_pol ? c->num_pol : depth;
  for (int j = 0; j < c->num_agents; j++) {
    if (j == c->id - 1) continue;
    const int nv = hull_nv[j * c->num_pol + (idx - 1)];
    if (nv <= 0) continue;
    const double(*V)[2] = (const double(*)[2])(hull_xy + ((size_t)(j * c->num_pol + (idx - 1)) * NEP_HULL_MAX_V) * 2);
    if (!fe_aabb_overlap(nv, V, Qx, Qy)) continue;
    if (orc_gjk_collision(nv, V, 4, (const double(*)[2])Q)) return 1;
  }
  for (int s = 0; statics && s < statics->n; s++) {
    const int nv = statics->off[s + 1] - statics->off[s];
    if (nv <= 0) continue;
    const double(*V)[2] = (const double(*)[2])(statics->xy + 2 * (size_t)statics->off[s]);
    if (!fe_aabb_overlap(nv, V, Qx, Qy)) continue;
    if (orc_gjk_collision(nv, V, 4, (const double(*)[2])Q)) return 1;
  }
  return 0;
}

/* Neptune::getInitialZPwp (neptune.cpp:1727-1810): height profile of the guess as a clamped-ramp uniform B-spline,
 * bspline_basis_[i] = diag(1, 1/T, 1/T^2, 1/T^3) * M/6 (neptune.cpp:67-82), coefficients reversed to [a b c d]. */
static void fe_initial_z(double p0, double v0, double a0, double z_final, double T, int np, double v_max_z, double a_max_z, double co[NEP_MAX_POL][4]) {
  double q[NEP_MAX_POL + 3], v[NEP_MAX_POL + 2];
  if (np < 3) { for (int i = 0; i < np; i++) { co[i][0] = co[i][1] = co[i][2] = 0; co[i][3] = p0; } return; }
  if (v0 < -v_max_z) v0 = -v_max_z; else if (v0 > v_max_z) v0 = v_max_z;
  if (a0 < -a_max_z) a0 = -a_max_z; else if (a0 > a_max_z) a0 = a_max_z;
  for (int i = 0; i < np + 2; i++) v[i] = 0;
  q[0] = p0;
  q[1] = p0 + T * v0 / 3;
  q[2] = (3 * 3 * q[1] - 2 * T * (-a0 * T + v0) - 3 * (q[1] + (-2 * T) * v0)) / 6;
  q[np] = z_final;
  const double increment = (z_final - q[2]) / (np - 2);
  for (int i = 3; i <= np - 1; i++) q[i] = q[i - 1] + increment;
  for (int i = 3; i <= np; i++) {
    v[i - 1] = (q[i] - q[i - 1]) / T;
    if (v[i - 1] > v_max_z) { q[i] = q[i - 1] + v_max_z * T; v[i - 1] = v_max_z; }
    else if (v[i - 1] < -v_max_z) { q[i] = q[i - 1] - v_max_z * T; v[i - 1] = -v_max_z; }
  }
  for (int i = 2; i <= np - 1; i++) {
    const double a_i = (v[i] - v[i - 1]) / T;
    if (a_i > a_max_z) v[i] = v[i - 1] + a_max_z * T;
    else if (a_i < -a_max_z) v[i] = v[i - 1] - a_max_z * T;
    q[i + 1] = q[i] + v[i] * T;
  }
  q[np + 1] = q[np]; q[np + 2] = q[np];
  for (int i = 0; i < np; i++) {
    const double* s = q + i;
    /* M/6 rows, then the interval scaling, lowest power first */
    const double c0 = (((1.0 / 6.0) * s[0] + (4.0 / 6.0) * s[1]) + (1.0 / 6.0) * s[2]) + (0.0 / 6.0) * s[3];
    const double c1 = (((-3.0 / 6.0) * s[0] + (0.0 / 6.0) * s[1]) + (3.0 / 6.0) * s[2]) + (0.0 / 6.0) * s[3];
    const double c2 = (((3.0 / 6.0) * s[0] + (-6.0 / 6.0) * s[1]) + (3.0 / 6.0) * s[2]) + (0.0 / 6.0) * s[3];
    const double c3 = (((-1.0 / 6.0) * s[0] + (3.0 / 6.0) * s[1]) + (-3.0 / 6.0) * s[2]) + (1.0 / 6.0) * s[3];
    co[i][3] = 1.0 * c0; co[i][2] = (1 / T) * c1; co[i][1] = (1 / (T * T)) * c2; co[i][0] = (1 / (T * T * T)) * c3;
  }
}

static int fe_before(const fe_node* a, int ia, const fe_node* b, int ib) { return a->f < b->f || (a->f == b->f && ia < ib); }

/* collidesWithBases2d (kinodynamic_search.cpp:1583-1628): the control polygon against the 0.7 m squares of the other agents' bases */
static int fe_collides_bases(const orc_fe_cfg* c, const fe_node* nd) {
  double Qx[4], Qy[4], Q[4][2];
  orc_pos_ctrl_pts(nd->cx, c->T_span, Qx); orc_pos_ctrl_pts(nd->cy, c->T_span, Qy);
  for (int i = 0; i < 4; i++) { Q[i][0] = Qx[i]; Q[i][1] = Qy[i]; }
  const double radius = 0.7, safe_dist = (c->T_span * c->v_max) * 2;
  for (int j = 0; j < c->num_agents; j++) {
    if (j == c->id - 1) continue;
    const double bx = c->pb[2 * j], by = c->pb[2 * j + 1];
    const double d1 = sqrt((Q[0][0] - bx) * (Q[0][0] - bx) + (Q[0][1] - by) * (Q[0][1] - by));
    if (d1 > safe_dist) continue;
    const double B[4][2] = {{bx + radius, by + radius}, {bx + radius, by - radius}, {bx - radius, by - radius}, {bx - radius, by + radius}};
    if (orc_gjk_collision(4, B, 4, (const double(*)[2])Q)) return 1;
  }
  return 0;
}

int orc_frontend_beam_ent(const orc_fe_cfg* c, const nep_fe_start* st, const double* hull_xy, const int* hull_nv,
                          const orc_polys* statics, const orc_fe_ent* E, nep_guess* guess, nep_fe_result* res, int* case_out);
int orc_frontend_beam(const orc_fe_cfg* c, const nep_fe_start* st, const double* hull_xy, const int* hull_nv,
                      const orc_polys* statics, nep_guess* guess, nep_fe_result* res) {
  return orc_frontend_beam_ent(c, st, hull_xy, hull_nv, statics, 0, guess, res, 0);
}
/* E != NULL: enable_entangle_check — every child carries its parent's entangle state through entanglesWithOtherAgents
 * (pruned when that returns true), g is the sampled arc length, h gains 0.3 per crossing and 1.0 per bend point
 * (:1177-1182), a voxel is (ix, iy, getIz(state)) (:1170-1173), base squares are obstacles (:1675), and only nodes whose
 * active_cases are all <= 1 may end the plan (:1693-1700).  case_out [NEP_MAX_POL][N]: the case id per (knot, agent) the
 * back end consumes (solver_gurobi_poly.cpp:624-631), state at the START of segment i. */
int orc_frontend_beam_ent(const orc_fe_cfg* c, const nep_fe_start* st, const double* hull_xy, const int* hull_nv,
                          const orc_polys* statics, const orc_fe_ent* E, nep_guess* guess, nep_fe_result* res, int* case_out) {
  const int W = c->beam_width, ns = c->num_samples, NC = ns * ns, D = c->num_pol;
  if (W < 1 || W > NEP_FE_MAX_BEAM || ns < 2 || ns > NEP_FE_MAX_SAMPLES || D < 1 || D > NEP_MAX_POL) return -1;
  static const int CAP = NEP_FE_MAX_BEAM * NEP_FE_MAX_SAMPLES * NEP_FE_MAX_SAMPLES;
  ent_ctx ec; memset(&ec, 0, sizeof(ec));
  orc_ent_node* ent_root_p = 0;
  orc_ent_node* cand_ent = 0; orc_ent_node (*beam_ent)[NEP_FE_MAX_BEAM] = 0; unsigned* cand_iz = 0; unsigned (*beam_iz)[NEP_FE_MAX_BEAM] = 0;
  unsigned* vis_iz = 0;
  if (E) {
    if (E->num_samples < 1 || E->num_samples > 8) return -1;
    ec.N = c->num_agents; ec.S = E->n_static; ec.own = c->id - 1; ec.num_pol = c->num_pol; ec.ns = E->num_samples; ec.T_span = c->T_span; ec.cable = c->cable_length;
    ec.pb = c->pb; ec.srep = E->static_rep; ec.slong = E->static_longest; ec.sampled = E->sampled; ec.present = E->present; ec.bend_n = E->bend_n; ec.bend_xy = E->bend_xy;
    const int ncap = ent_node_cap(&ec, 1);
    ent_root_p = ent_nodes_alloc(1, ncap);
    if (E->init) ent_node_from_fixed(ent_root_p, E->init);
    cand_ent = ent_nodes_alloc((size_t)W * NC, ncap);
    beam_ent = (orc_ent_node(*)[NEP_FE_MAX_BEAM])ent_nodes_alloc((size_t)NEP_FE_MAX_BEAM * (NEP_MAX_POL + 1), ncap);
    cand_iz = (unsigned*)calloc(CAP, sizeof(unsigned));
    beam_iz = (unsigned(*)[NEP_FE_MAX_BEAM])calloc((size_t)NEP_FE_MAX_BEAM * (NEP_MAX_POL + 1), sizeof(unsigned));
    vis_iz = (unsigned*)calloc((size_t)NEP_FE_MAX_BEAM * (NEP_MAX_POL + 1), sizeof(unsigned));
  }
  fe_node* cand = (fe_node*)malloc(sizeof(fe_node) * CAP);
  int* keep = (int*)malloc(sizeof(int) * CAP);
  fe_node (*beam)[NEP_FE_MAX_BEAM] = (fe_node(*)[NEP_FE_MAX_BEAM])malloc(sizeof(fe_node) * NEP_FE_MAX_BEAM * (NEP_MAX_POL + 1));
  int beam_n[NEP_MAX_POL + 2];
  long (*visited)[2] = (long(*)[2])malloc(sizeof(long) * 2 * NEP_FE_MAX_BEAM * (NEP_MAX_POL + 1));
  int n_vis = 0;
  memset(res, 0, sizeof(*res));
  memset(guess, 0, sizeof(*guess));
  guess->t_start = st->t_start;
  const double goal[2] = {st->goal[0], st->goal[1]};
  /* goal_occupied_ (setUp :210-226); recorded, the beam's ranking does not use it */
  {
    const double r = 0.5;
    const double G[4][2] = {{goal[0] + r, goal[1] + r}, {goal[0] + r, goal[1] - r}, {goal[0] - r, goal[1] + r}, {goal[0] - r, goal[1] - r}};
    for (int j = 0; j < c->num_agents && !res->goal_occupied; j++) {
      if (j == c->id - 1) continue;
      const int nv = hull_nv[j * c->num_pol + (c->num_pol - 1)];
      if (nv <= 0) continue;
      if (orc_gjk_collision(nv, (const double(*)[2])(hull_xy + ((size_t)(j * c->num_pol + c->num_pol - 1) * NEP_HULL_MAX_V) * 2), 4, G)) res->goal_occupied = 1;
    }
  }
  const double root[6] = {st->pos[0], st->pos[1], st->vel[0], st->vel[1], st->accel[0], st->accel[1]};
  int status = NEP_FE_NO_SOLUTION, best_depth = 0, best_rank = -1;
  beam_n[0] = 1;
  int depth;
  for (depth = 1; depth <= D; depth++) {
    const int n_par = depth == 1 ? 1 : beam_n[depth - 1];
    int n_c = 0;
    for (int pr = 0; pr < n_par; pr++)
      for (int cc = 0; cc < NC; cc++) {
        const int id = pr * NC + cc;
        fe_node* nd = &cand[id];
        keep[id] = 0;
        res->n_children++;
        const double* pe = depth == 1 ? root : beam[depth - 1][pr].end;
        const double pg = depth == 1 ? 0.0 : beam[depth - 1][pr].g;
        if (!fe_child(c, pe, pg, depth == 1, cc / ns, cc % ns, goal, nd)) continue;
        nd->parent = depth == 1 ? -1 : pr;
        res->n_feasible++;
        {
          const int col = fe_collides(c, nd, depth, hull_xy, hull_nv, statics);
          if (g_fe_dump && g_fe_dump_n < g_fe_dump_cap) {
            double* o = g_fe_dump + 11 * (size_t)g_fe_dump_n++;
            double Qx[4], Qy[4]; orc_pos_ctrl_pts(nd->cx, c->T_span, Qx); orc_pos_ctrl_pts(nd->cy, c->T_span, Qy);
            o[0] = depth; o[1] = id; o[2] = col; for (int k = 0; k < 4; k++) { o[3 + 2 * k] = Qx[k]; o[4 + 2 * k] = Qy[k]; }
          }
          if (col) continue;
        }
        if (E && fe_collides_bases(c, nd)) continue;
        res->n_collision_free++;
        if (E) {
          ent_node_copy(&cand_ent[id], depth == 1 ? ent_root_p : &beam_ent[depth - 1][pr]);
          double arc = 0.0;
          const ev2 end = {nd->end[0], nd->end[1]};
          const int rc = ent_propagate(&ec, &cand_ent[id], nd->cx, nd->cy, end, depth, &arc, 1, 1);
          if (rc) { res->n_entangled++; if (rc == 2) res->ent_overflow = 1; continue; }
          nd->g = pg + arc;
          nd->f = nd->g + c->bias * ((nd->dist + 0.3 * (double)cand_ent[id].n_alpha) + 1.0 * (double)cand_ent[id].n_bend);
          cand_iz[id] = ent_iz(&cand_ent[id]);
        }
        int seen = 0;
        for (int v = 0; v < n_vis && !seen; v++) seen = visited[v][0] == nd->vx && visited[v][1] == nd->vy && (!E || vis_iz[v] == cand_iz[id]);
        if (seen) continue;
        keep[id] = 1;
      }
    n_c = n_par * NC;
    /* one node per voxel: the best (f, id) of the depth */
    for (int i = 0; i < n_c; i++) {
      if (!keep[i]) continue;
      for (int k = 0; k < n_c; k++) {
        if (k == i || !keep[k] || cand[k].vx != cand[i].vx || cand[k].vy != cand[i].vy || (E && cand_iz[k] != cand_iz[i])) continue;
        if (fe_before(&cand[k], k, &cand[i], i)) { keep[i] = 2; break; }   /* 2: loses its voxel (still counts for the others' comparisons) */
      }
    }
    /* the beam: the W best survivors in (f, id) order */
    int nb = 0;
    for (;;) {
      int bi = -1;
      for (int i = 0; i < n_c; i++) if (keep[i] == 1 && (bi < 0 || fe_before(&cand[i], i, &cand[bi], bi))) bi = i;
      if (bi < 0 || nb == W) break;
      if (E) { ent_node_copy(&beam_ent[depth][nb], &cand_ent[bi]); beam_iz[depth][nb] = cand_iz[bi]; }
      beam[depth][nb++] = cand[bi];
      keep[bi] = 3;
    }
    beam_n[depth] = nb;
    if (nb == 0) { status = depth == 1 ? NEP_FE_NO_SOLUTION : NEP_FE_EMPTY; break; }
    for (int r = 0; r < nb; r++) { visited[n_vis][0] = beam[depth][r].vx; visited[n_vis][1] = beam[depth][r].vy; if (E) vis_iz[n_vis] = beam_iz[depth][r]; n_vis++; }
    /* a plan may only end at a node none of whose agents has two active cases (:1693-1700): always true without the check */
    int valid[NEP_FE_MAX_BEAM];
    for (int r = 0; r < nb; r++) {
      valid[r] = 1;
      if (E) for (int a = 0; a < beam_ent[depth][r].n_alpha && valid[r]; a++)
        if (beam_ent[depth][r].id[a] <= c->num_agents && ent_count(beam_ent[depth][r].id, beam_ent[depth][r].n_alpha, beam_ent[depth][r].id[a]) > 1) valid[r] = 0;
    }
    { int fv = -1; for (int r = 0; r < nb && fv < 0; r++) if (valid[r]) fv = r; if (fv >= 0) { best_depth = depth; best_rank = fv; } }
    int reached = -1;
    for (int r = 0; r < nb && reached < 0; r++) if (beam[depth][r].dist < c->goal_size && valid[r]) reached = r;
    if (reached >= 0) { status = NEP_FE_GOAL_REACHED; best_depth = depth; best_rank = reached; break; }
    if (depth == D) { status = NEP_FE_DEPTH_REACHED; break; }
  }
  res->status = status;
  res->depth = depth > D ? D : depth;
  if (best_rank >= 0) {
    const fe_node* nd = &beam[best_depth][best_rank];
    res->K = best_depth; res->cost = nd->f; res->dist_to_goal = nd->dist;
    guess->K = best_depth;
    int r = best_rank;
    for (int d = best_depth; d >= 1; d--) {
      const fe_node* q = &beam[d][r];
      for (int k = 0; k < 4; k++) { guess->coeff[0][d - 1][k] = q->cx[k]; guess->coeff[1][d - 1][k] = q->cy[k]; }
      r = q->parent;
    }
    double cz[NEP_MAX_POL][4];
    fe_initial_z(st->pos[2], st->vel[2], st->accel[2], st->goal[2], c->T_span, D, c->v_max, c->a_max, cz);   /* coeffs_z_ (:540) */
    for (int d = 1; d <= best_depth; d++) for (int k = 0; k < 4; k++) guess->coeff[2][d - 1][k] = cz[d - 1][k];
    if (c->pad_hold && best_depth < D) {   /* hold the end point for the rest of the horizon */
      for (int d = best_depth + 1; d <= D; d++) {
        for (int ax = 0; ax < 2; ax++) for (int k = 0; k < 3; k++) guess->coeff[ax][d - 1][k] = 0;
        guess->coeff[0][d - 1][3] = nd->end[0]; guess->coeff[1][d - 1][3] = nd->end[1];
        for (int k = 0; k < 4; k++) guess->coeff[2][d - 1][k] = cz[d - 1][k];      /* the height keeps following its profile */
      }
      guess->K = D;
    }
    if (E && case_out) {   /* state at the start of segment i = the path's node of depth i (depth 0: the initial state) */
      int path[NEP_MAX_POL + 1]; int rr = best_rank;
      for (int d = best_depth; d >= 1; d--) { path[d] = rr; rr = beam[d][rr].parent; }
      for (int i = 0; i < NEP_MAX_POL; i++) {
        const int dd = i < best_depth ? i : best_depth;      /* held segments keep the last node's state */
        const orc_ent_node* sn = dd == 0 ? ent_root_p : &beam_ent[dd][path[dd]];
        if (i < guess->K) ent_case_row(sn, c->num_agents, case_out + (size_t)i * c->num_agents);
        else for (int j = 0; j < c->num_agents; j++) case_out[(size_t)i * c->num_agents + j] = 0;
      }
    }
  } else if (E && case_out) memset(case_out, 0, sizeof(int) * NEP_MAX_POL * (size_t)c->num_agents);
  free(cand); free(keep); free(beam); free(visited);
  if (E) { free(cand_ent); free(beam_ent); free(ent_root_p); free(cand_iz); free(beam_iz); free(vis_iz); }
  return 0;
}

/* Test hook: the states KinodynamicSearch::recoverEntStateVector returns for a GIVEN K-segment path (:582-603), reduced to
 * the (knot, agent) case ids; *hit_at = first 1-based segment whose update returned true (states repeat from there), 0 if
 * none.  Lets the C restatement above be compared with oracle/entangle_oracle.py on the same guesses. */
int orc_ent_propagate_guess(const orc_fe_cfg* c, const orc_fe_ent* E, const nep_guess* g, int* case_out, int* hit_at, int* n_alpha_final) {
  ent_ctx ec; memset(&ec, 0, sizeof(ec));
  ec.N = c->num_agents; ec.S = E->n_static; ec.own = c->id - 1; ec.num_pol = c->num_pol; ec.ns = E->num_samples; ec.T_span = c->T_span; ec.cable = c->cable_length;
  ec.pb = c->pb; ec.srep = E->static_rep; ec.slong = E->static_longest; ec.sampled = E->sampled; ec.present = E->present; ec.bend_n = E->bend_n; ec.bend_xy = E->bend_xy;
  orc_ent_node* two = ent_nodes_alloc(2, ent_node_cap(&ec, 1));
  orc_ent_node* stt = &two[0]; orc_ent_node* nx = &two[1];
  if (E->init) ent_node_from_fixed(stt, E->init);
  const double T = c->T_span;
  const int K = g->K;
  *hit_at = 0;
  for (int s = 1; s <= K; s++) {
    ent_case_row(stt, c->num_agents, case_out + (size_t)(s - 1) * c->num_agents);
    if (*hit_at) continue;
    const double* cxo = g->coeff[0][s - 1]; const double* cyo = g->coeff[1][s - 1];
    ev2 end;
    if (s < K) { end.x = g->coeff[0][s][3]; end.y = g->coeff[1][s][3]; }
    else { end.x = ((cxo[0] * (T * T * T) + cxo[1] * (T * T)) + cxo[2] * T) + cxo[3]; end.y = ((cyo[0] * (T * T * T) + cyo[1] * (T * T)) + cyo[2] * T) + cyo[3]; }
    double arc = 0.0;
    ent_node_copy(nx, stt);
    if (ent_propagate(&ec, nx, cxo, cyo, end, s, &arc, 1, 1)) *hit_at = s; else ent_node_copy(stt, nx);
  }
  for (int s = K; s < NEP_MAX_POL; s++) for (int j = 0; j < c->num_agents; j++) case_out[(size_t)s * c->num_agents + j] = 0;
  if (n_alpha_final) *n_alpha_final = stt->n_alpha;
  free(two);
  return 0;
}

/* KinodynamicSearch::entangleCheckGivenPwp (kinodynamic_search.cpp:897-985) as called by safetyCheckAfterReplan
 * (neptune.cpp:746-754): the new trajectory against the entangle state at its start.  As in the reference only the FIRST
 * interval is examined (the loop returns false at the end of its first pass, :983), the capacity test is three times the
 * search's (:944-948) and the tether length is not tested (:977-982 commented out).  1 = entangles. */
int orc_entangle_check_pwp(const orc_fe_cfg* c, const orc_fe_ent* E, const double cx0[4], const double cy0[4]) {
  ent_ctx ec; memset(&ec, 0, sizeof(ec));
  ec.N = c->num_agents; ec.S = E->n_static; ec.own = c->id - 1; ec.num_pol = c->num_pol; ec.ns = E->num_samples; ec.T_span = c->T_span; ec.cable = c->cable_length;
  ec.pb = c->pb; ec.srep = E->static_rep; ec.slong = E->static_longest; ec.sampled = E->sampled; ec.present = E->present; ec.bend_n = E->bend_n; ec.bend_xy = E->bend_xy;
  orc_ent_node* stt = ent_nodes_alloc(1, ent_node_cap(&ec, 3));
  if (E->init) ent_node_from_fixed(stt, E->init);
  const double T = c->T_span;
  /* pkplus1 = P * sampled_time_vector_[j] for every j, including the last (:914): the polynomial at T */
  const ev2 end = {((cx0[0] * (T * T * T) + cx0[1] * (T * T)) + cx0[2] * T) + cx0[3] * 1.0, ((cy0[0] * (T * T * T) + cy0[1] * (T * T)) + cy0[2] * T) + cy0[3] * 1.0};
  double arc = 0.0;
  const int rc = ent_propagate(&ec, stt, cx0, cy0, end, 1, &arc, 0, 3) != 0;
  free(stt);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* SURVEY §8(f) rank 2, first half: KinodynamicSearch::run itself (kinodynamic_search.cpp:1629-1827) */
/* with its two expansion routines (:1045-1228, :1240-1385), CompareCost (kinodynamic_search.hpp:  */
/* 163-179) on a binary heap with libstdc++'s push_heap / pop_heap mechanics, the (ix, iy) -> first  */
/* node hash (NodeHashTable::insert does not overwrite), the in-place update of an open node of the  */
/* same index on every second such event (ran_trigger, :1190-1203), the closest-so-far bookkeeping   */
/* (:1683-1690) and recoverPwpOut (:521-553).  Entangle check off.  The two things that make the     */
/* reference irreproducible are parameters here: `order` = all_combinations_ (shuffled with a time   */
/* seed there, :321-322, :1462-1463) and `max_pops` in place of the wall-clock budget (:1644-1651).   */
/* Reference material for the beam variant above (quality comparisons in tests); not on any product   */
/* path.                                                                                             */
/* ------------------------------------------------------------------------------------------ */
#define AS_MAX_NODES 20000          /* node_num_max_ (kinodynamic_search.hpp:345) */
typedef struct as_node { int prev, state, index; double g, h, end[6], cx[4], cy[4], Q[4][2]; long vx, vy; } as_node;
typedef struct as_ctx {
  const orc_fe_cfg* c; const double* hull_xy; const int* hull_nv; const orc_polys* statics; const double* goal; const int* order;
  as_node* pool; int used; int* heap; int heap_n; long (*hkey)[2]; int* hval; int hcap; int ran_trigger;
} as_ctx;
static double as_cost(const as_ctx* A, int n) { return A->pool[n].g + A->c->bias * A->pool[n].h; }
static int as_comp(const as_ctx* A, int l, int r) {     /* CompareCost(left, right) */
  const double cl = as_cost(A, l), cr = as_cost(A, r);
  if (fabs(cl - cr) < 1e-5) return A->pool[l].h > A->pool[r].h;
  return cl > cr;
}
static void as_push_heap(as_ctx* A, int hole, int top, int value) {      /* std::__push_heap */
  int parent = (hole - 1) / 2;
  while (hole > top && as_comp(A, A->heap[parent], value)) { A->heap[hole] = A->heap[parent]; hole = parent; parent = (hole - 1) / 2; }
  A->heap[hole] = value;
}
static void as_push(as_ctx* A, int n) { A->heap[A->heap_n++] = n; as_push_heap(A, A->heap_n - 1, 0, n); }
static int as_pop(as_ctx* A) {                                           /* top(); pop(): std::pop_heap + pop_back */
  const int top = A->heap[0];
  const int len = A->heap_n - 1;          /* __pop_heap(first, last-1, last-1): value = *(last-1), *(last-1) = *first */
  const int value = A->heap[len];
  A->heap[len] = top;
  int hole = 0, second = 0;
  while (second < (len - 1) / 2) {        /* std::__adjust_heap(first, 0, len, value) */
    second = 2 * (second + 1);
    if (as_comp(A, A->heap[second], A->heap[second - 1])) second--;
    A->heap[hole] = A->heap[second]; hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) { second = 2 * (second + 1); A->heap[hole] = A->heap[second - 1]; hole = second - 1; }
  if (len > 0) as_push_heap(A, hole, 0, value);
  A->heap_n = len;
  return top;
}
static int as_find(const as_ctx* A, long vx, long vy) {
  unsigned long h = ((unsigned long)vx * 0x9E3779B97F4A7C15ul) ^ ((unsigned long)vy * 0xC2B2AE3D27D4EB4Ful);
  for (int k = (int)(h % (unsigned long)A->hcap);; k = (k + 1) % A->hcap) {
    if (A->hval[k] < 0) return -1 - k;                                   /* empty slot: where it would go */
    if (A->hkey[k][0] == vx && A->hkey[k][1] == vy) return A->hval[k];
  }
}
static void as_insert(as_ctx* A, long vx, long vy, int n) {               /* unordered_map::insert: keeps the first */
  const int f = as_find(A, vx, vy);
  if (f >= 0) return;
  const int k = -1 - f; A->hkey[k][0] = vx; A->hkey[k][1] = vy; A->hval[k] = n;
}
static int as_collides(const as_ctx* A, const as_node* nd) {              /* collidesWithObstacles2dSolve (:1514-1553) */
  const orc_fe_cfg* c = A->c;
  int idx = nd->index > c->num_pol ? c->num_pol : nd->index;
  for (int j = 0; j < c->num_agents; j++) {
    if (j == c->id - 1) continue;
    const int nv = A->hull_nv[j * c->num_pol + (idx - 1)];
    if (nv <= 0) continue;
    if (orc_gjk_collision(nv, (const double(*)[2])(A->hull_xy + ((size_t)(j * c->num_pol + (idx - 1)) * NEP_HULL_MAX_V) * 2), 4, nd->Q)) return 1;
  }
  for (int s = 0; A->statics && s < A->statics->n; s++) {
    const int nv = A->statics->off[s + 1] - A->statics->off[s];
    if (nv > 0 && orc_gjk_collision(nv, (const double(*)[2])(A->statics->xy + 2 * (size_t)A->statics->off[s]), 4, nd->Q)) return 1;
  }
  return 0;
}
/* both expandAndAddToQueue overloads: cur < 0 = from the initial state */
static void as_expand(as_ctx* A, int cur, const double init[6]) {
  const orc_fe_cfg* c = A->c;
  const int ns = c->num_samples;
  const double* st = cur < 0 ? init : A->pool[cur].end;
  for (int q = 0; q < ns * ns; q++) {
    if (cur >= 0 && A->used == AS_MAX_NODES - 1) return;                  /* "run out of memory" (:1061-1065) */
    if (A->used >= AS_MAX_NODES - 1) return;
    const int comb = A->order ? A->order[q] : q;
    as_node* nb = &A->pool[A->used];
    fe_node tmp;
    /* the child tests (fe_child above states the two norm tests on the squares) */
    if (!fe_child(c, st, 0.0, cur < 0, comb / ns, comb % ns, A->goal, &tmp)) continue;
    nb->index = cur < 0 ? 1 : A->pool[cur].index + 1; nb->prev = cur;
    for (int i = 0; i < 6; i++) nb->end[i] = tmp.end[i];
    for (int i = 0; i < 4; i++) { nb->cx[i] = tmp.cx[i]; nb->cy[i] = tmp.cy[i]; }
    double Qx[4], Qy[4];
    orc_pos_ctrl_pts(nb->cx, c->T_span, Qx); orc_pos_ctrl_pts(nb->cy, c->T_span, Qy);
    for (int i = 0; i < 4; i++) { nb->Q[i][0] = Qx[i]; nb->Q[i][1] = Qy[i]; }
    const double arc = sqrt((nb->end[0] - st[0]) * (nb->end[0] - st[0]) + (nb->end[1] - st[1]) * (nb->end[1] - st[1]));
    nb->g = (cur < 0 ? 0.0 : A->pool[cur].g) + arc;
    nb->h = tmp.dist;                                                     /* getH (:401-405) */
    nb->vx = tmp.vx; nb->vy = tmp.vy;
    if (cur >= 0) {
      const int f = as_find(A, nb->vx, nb->vy);
      if (f >= 0) {
        as_node* o = &A->pool[f];
        if (o->state == 1 && o->index == nb->index) {
          if (nb->g + c->bias * nb->h < o->g + c->bias * o->h && A->ran_trigger % 2 == 0) {   /* update the open node in place */
            o->prev = cur; o->g = nb->g; o->h = nb->h;
            for (int i = 0; i < 6; i++) o->end[i] = nb->end[i];
            for (int i = 0; i < 4; i++) { o->cx[i] = nb->cx[i]; o->cy[i] = nb->cy[i]; o->Q[i][0] = nb->Q[i][0]; o->Q[i][1] = nb->Q[i][1]; }
          }
          A->ran_trigger++;
        }
        continue;
      }
    }
    nb->state = 1;
    as_push(A, A->used);
    as_insert(A, nb->vx, nb->vy, A->used);
    A->used++;
  }
}

int orc_frontend_astar(const orc_fe_cfg* c, const nep_fe_start* st, const double* hull_xy, const int* hull_nv, const orc_polys* statics,
                       const int* order, int max_pops, nep_guess* guess, nep_fe_result* res) {
  const int ns = c->num_samples;
  if (ns < 2 || ns > NEP_FE_MAX_SAMPLES || c->num_pol < 1 || c->num_pol > NEP_MAX_POL) return -1;
  as_ctx A; memset(&A, 0, sizeof(A));
  const double goal[2] = {st->goal[0], st->goal[1]};
  A.c = c; A.hull_xy = hull_xy; A.hull_nv = hull_nv; A.statics = statics; A.goal = goal; A.order = order;
  A.pool = (as_node*)calloc(AS_MAX_NODES, sizeof(as_node)); A.heap = (int*)malloc(sizeof(int) * AS_MAX_NODES);
  A.hcap = 4 * AS_MAX_NODES + 7; A.hkey = (long(*)[2])malloc(sizeof(long) * 2 * A.hcap); A.hval = (int*)malloc(sizeof(int) * A.hcap);
  for (int k = 0; k < A.hcap; k++) A.hval[k] = -1;
  memset(res, 0, sizeof(*res)); memset(guess, 0, sizeof(*guess)); guess->t_start = st->t_start;
  {   /* goal_occupied_ (setUp :210-226) */
    const double r = 0.5;
    const double G[4][2] = {{goal[0] + r, goal[1] + r}, {goal[0] + r, goal[1] - r}, {goal[0] - r, goal[1] + r}, {goal[0] - r, goal[1] - r}};
    for (int j = 0; j < c->num_agents && !res->goal_occupied; j++) {
      if (j == c->id - 1) continue;
      const int nv = hull_nv[j * c->num_pol + (c->num_pol - 1)];
      if (nv > 0 && orc_gjk_collision(nv, (const double(*)[2])(hull_xy + ((size_t)(j * c->num_pol + c->num_pol - 1) * NEP_HULL_MAX_V) * 2), 4, G)) res->goal_occupied = 1;
    }
  }
  const double init[6] = {st->pos[0], st->pos[1], st->vel[0], st->vel[1], st->accel[0], st->accel[1]};
  as_expand(&A, -1, init);
  int status = NEP_FE_EMPTY, cur = -1, closest = -1, pops = 0;
  double smallest = 1.7976931348623157e308;
  while (A.heap_n > 0) {
    if (pops >= max_pops) { status = NEP_FE_DEPTH_REACHED; break; }       /* RUNTIME_REACHED */
    cur = as_pop(&A); pops++;
    as_node* nd = &A.pool[cur];
    nd->state = -1;
    const double dist = sqrt((nd->end[0] - goal[0]) * (nd->end[0] - goal[0]) + (nd->end[1] - goal[1]) * (nd->end[1] - goal[1]));
    const double dfi = sqrt((nd->end[0] - init[0]) * (nd->end[0] - init[0]) + (nd->end[1] - init[1]) * (nd->end[1] - init[1]));
    if (as_collides(&A, nd)) continue;                                    /* ignore_first_col_ is always false (:1449-1455) */
    const double dtc = res->goal_occupied ? dist * dist : dfi;
    const double dti = dtc * (double)nd->index;
    if (dti < smallest) { smallest = dti; closest = cur; }
    if (dist < c->goal_size) { status = NEP_FE_GOAL_REACHED; break; }
    as_expand(&A, cur, init);
  }
  int best = -1;
  if (status == NEP_FE_GOAL_REACHED) best = cur;
  else best = closest;                                                    /* use_not_reaching_soln */
  res->n_children = pops; res->n_feasible = A.used; res->depth = best >= 0 ? A.pool[best].index : 0;
  if (best < 0) status = NEP_FE_NO_SOLUTION;
  res->status = status;
  if (best >= 0) {
    /* recoverPwpOut: the nodes of the path with index <= num_pol */
    int chain[4096], n = 0;
    for (int k = best; k >= 0 && n < 4096; k = A.pool[k].prev) chain[n++] = k;
    int K = 0;
    for (int i = n - 1; i >= 0; i--) {
      const as_node* q = &A.pool[chain[i]];
      if (q->index > c->num_pol) break;
      for (int k = 0; k < 4; k++) { guess->coeff[0][K][k] = q->cx[k]; guess->coeff[1][K][k] = q->cy[k]; }
      K++;
    }
    double cz[NEP_MAX_POL][4];
    fe_initial_z(st->pos[2], st->vel[2], st->accel[2], st->goal[2], c->T_span, c->num_pol, c->v_max, c->a_max, cz);
    for (int d = 0; d < K; d++) for (int k = 0; k < 4; k++) guess->coeff[2][d][k] = cz[d][k];
    guess->K = K; res->K = K;
    res->cost = A.pool[best].g + c->bias * A.pool[best].h;
    res->dist_to_goal = sqrt((A.pool[best].end[0] - goal[0]) * (A.pool[best].end[0] - goal[0]) + (A.pool[best].end[1] - goal[1]) * (A.pool[best].end[1] - goal[1]));
  }
  free(A.pool); free(A.heap); free(A.hkey); free(A.hval);
  return 0;
}
