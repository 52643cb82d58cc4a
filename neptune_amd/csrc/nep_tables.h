// nep_tables.h — constants and per-(K, mode) tables of the reduced QP (host builds, device reads).
//
// The back end's QP (reference neptune/src/solver_gurobi_poly.cpp:322-710) has 12K coefficients
// tied by 9K+6 equalities.  Per axis only the K leading coefficients a_i (jerk/6) are free; all
// other coefficients follow from the initial (b0,c0,d0) by forward integration of the C2
// continuity rows (:400-425), and the terminal v=a=0 rows (:659-678) remove two more degrees of
// freedom.  Tables below express every quantity the solver needs as an affine map of the reduced
// variable z (nz = K-2, or K in the relaxed re-solve :838-861) and the initial state.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/neptune_backend.h"

namespace nep {

// Inverses of mt::basisConverter::A_pos_mv_rest / A_vel_mv_rest on t in [0,1]
// (reference neptune/include/mader_types.hpp:152-163): exact rational inverse of the
// double-rounded literals, rounded to double.  (A*diag(T^-3,T^-2,T^-1,1))^-1 = diag(T^3,..)*A^-1
// gives A_rest_pos_basis_inverse_ of solver_gurobi_poly.cpp:93.
static const double kAPosInv[4][4] = {
    {-0.03203276669713047, -0.09273093424558249, 0.3420572455666699, 1.1023313949144335},
    {-0.05111494245568798, -0.046272612998418894, 0.5458234872124772, 1.0979806946005568},
    {-0.07454781852812224, 0.203951949894552, 0.796048050105448, 1.0745478185281223},
    {1.0, 1.0, 0.9999999999999996, 0.9999999999999993}};
static const double kAVelInv[3][3] = {
    {-0.07735026918962577, 0.16666666666666635, 1.077350269189625},
    {-0.07735026918962577, 0.49999999999999967, 1.077350269189625},
    {1.0000000000000002, 1.0000000000000009, 1.0000000000000016}};

constexpr int kMaxK = NEP_MAX_POL;   // 8
constexpr int kMaxR = 8 * kMaxK;     // base rows per axis: 4K position CPs, 3K velocity CPs, K accelerations
constexpr int kNZ = 8;               // padded reduced dimension per axis

// One table per (K, mode).  mode 0: terminal v=a=0 eliminated; mode 1: relaxed (kept in the cost).
struct QpTable {
  int K, nz, R, mode;
  double B[kMaxR][kNZ];      // base row rho: value = B[rho].z + U[rho].(b0,c0,d0)
  double U[kMaxR][3];
  double Th[4 * kMaxK][kNZ]; // coefficient recovery: theta[4i+j] = Th.z + ThU.init
  double ThU[4 * kMaxK][3];
  double Hax[kNZ][kNZ];      // per-axis Hessian of the cost in z
  // (round 6, qp_presolve_kernel) the minimiser of the cost without inequality rows is LINEAR in v = (b0, c0, d0, f) of an axis —
  // z* = -HaxInv (Gi init - 2 w ep f) — and so is everything evaluated there: base row rho = RowMap[rho] . v, coefficient r = ThMap[r] . v,
  // cost of the axis = v' ObjQ v.  One 32-byte table row per lane instead of the chain g -> z* -> B z* + U init.
  double RowMap[kMaxR][4];
  double ThMap[4 * kMaxK][4];
  double ObjQ[4][4];
  double Gi[kNZ][3];         // gradient map of the init state; g = Gi.init - 2 w ep f
  double ep[kNZ], ev[kNZ], ea[kNZ]; // end position / velocity / acceleration maps (z part)
  double up[3], uv[3], ua[3];       // (init part)
  double Nt[kNZ][kMaxK];     // z0 = Nt (a_guess - Pp init)
  double Pp[kMaxK][3];
  double res_u[2][3];        // terminal (b_K, c_K) at a = Pp.init: consistency check for K <= 2
  double Zp[kNZ][4 * kMaxK]; // start point: z0 = Zp (theta_guess - ThU init), the orthogonal projection of the guess's
                             // coefficients onto the feasible affine set {Th z + ThU init}  (Zp = (Th'Th)^-1 Th')
  double HaxInv[kNZ][kNZ];   // inverse of the per-axis Hessian: the minimiser without inequality rows is -HaxInv g (presolve)
};

struct SampleTable {         // generatePwpOut's time walk (solver_gurobi_poly.cpp:911-934)
  int n;
  int seg[1];                // followed by n ints and n doubles (allocated flat)
};

inline void build_qp_table(int K, double T, double weight, int mode, QpTable* t) {
  std::memset(t, 0, sizeof(*t));
  double M4[4][4], V3[3][3];
  const double tp[4] = {T * T * T, T * T, T, 1.0};
  const double tv[3] = {T * T, T, 1.0}, m321[3] = {3.0, 2.0, 1.0};
  for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) M4[j][k] = tp[j] * kAPosInv[j][k];
  for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) V3[j][k] = m321[j] * (tv[j] * kAVelInv[j][k]);
  // theta = Phi a + PhiU init by the continuity recurrences
  std::vector<double> Phi(4 * K * K, 0.0), PhiU(4 * K * 3, 0.0);
  double ba[3][kMaxK] = {{0}}, bu[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};  // rows b,c,d
  for (int i = 0; i < K; i++) {
    Phi[(4 * i + 0) * K + i] = 1.0;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < K; c++) Phi[(4 * i + 1 + r) * K + c] = ba[r][c];
      for (int c = 0; c < 3; c++) PhiU[(4 * i + 1 + r) * 3 + c] = bu[r][c];
    }
    double na[3][kMaxK], nu[3][3];
    for (int c = 0; c < K; c++) {
      double e = (c == i) ? 1.0 : 0.0;
      na[0][c] = ba[0][c] + 3 * T * e;                                        // 2b' = 6T a + 2b   (:418-423)
      na[1][c] = ba[1][c] + 2 * T * ba[0][c] + 3 * T * T * e;                 // c' = 3T^2 a + 2T b + c (:411-416)
      na[2][c] = ba[2][c] + T * ba[1][c] + T * T * ba[0][c] + T * T * T * e;  // d' = p(T)          (:404-409)
    }
    for (int c = 0; c < 3; c++) {
      nu[0][c] = bu[0][c];
      nu[1][c] = bu[1][c] + 2 * T * bu[0][c];
      nu[2][c] = bu[2][c] + T * bu[1][c] + T * T * bu[0][c];
    }
    std::memcpy(ba, na, sizeof(ba)); std::memcpy(bu, nu, sizeof(bu));
  }
  // ba/bu now hold (b_K, c_K, d_K)
  int nz;
  std::vector<double> N(K * kNZ, 0.0), Pp(K * 3, 0.0);
  if (mode == 1) {
    nz = K;
    for (int i = 0; i < K; i++) N[i * kNZ + i] = 1.0;
  } else {
    nz = K > 2 ? K - 2 : 0;
    // Et (2 x K) = rows (b_K, c_K); Pp = -Et^+ Ft with Et^+ = Et'(Et Et')^-1 (rank 2 when K>=2)
    const double* e0 = ba[0]; const double* e1 = ba[1];
    double g00 = 0, g01 = 0, g11 = 0;
    for (int c = 0; c < K; c++) { g00 += e0[c] * e0[c]; g01 += e0[c] * e1[c]; g11 += e1[c] * e1[c]; }
    if (K >= 2) {
      double det = g00 * g11 - g01 * g01;
      double i00 = g11 / det, i01 = -g01 / det, i11 = g00 / det;
      for (int c = 0; c < K; c++) {
        double p0 = e0[c] * i00 + e1[c] * i01, p1 = e0[c] * i01 + e1[c] * i11;  // column c of Et^+ (K x 2)
        for (int u = 0; u < 3; u++) Pp[c * 3 + u] = -(p0 * bu[0][u] + p1 * bu[1][u]);
      }
    } else {  // K == 1: least squares on a single unknown
      double den = e0[0] * e0[0] + e1[0] * e1[0];
      for (int u = 0; u < 3; u++) Pp[u] = -(e0[0] * bu[0][u] + e1[0] * bu[1][u]) / den;
    }
    if (nz > 0) {  // null space of Et: Householder QR of Et' (K x 2), N = Q[:, 2:]
      std::vector<double> A(K * 2), Q(K * K, 0.0);
      for (int c = 0; c < K; c++) { A[c * 2 + 0] = e0[c]; A[c * 2 + 1] = e1[c]; }
      for (int i = 0; i < K; i++) Q[i * K + i] = 1.0;
      for (int k = 0; k < 2; k++) {
        double nrm = 0; for (int i = k; i < K; i++) nrm += A[i * 2 + k] * A[i * 2 + k];
        nrm = std::sqrt(nrm);
        double alpha = A[k * 2 + k] >= 0 ? -nrm : nrm;
        std::vector<double> v(K, 0.0);
        v[k] = A[k * 2 + k] - alpha; for (int i = k + 1; i < K; i++) v[i] = A[i * 2 + k];
        double vv = 0; for (int i = k; i < K; i++) vv += v[i] * v[i];
        if (vv <= 0) continue;
        double beta = 2.0 / vv;
        for (int j = 0; j < 2; j++) { double d = 0; for (int i = k; i < K; i++) d += v[i] * A[i * 2 + j]; d *= beta; for (int i = k; i < K; i++) A[i * 2 + j] -= d * v[i]; }
        for (int r = 0; r < K; r++) { double d = 0; for (int i = k; i < K; i++) d += Q[r * K + i] * v[i]; d *= beta; for (int i = k; i < K; i++) Q[r * K + i] -= d * v[i]; }  // Q := Q H_k
      }
      for (int r = 0; r < K; r++) for (int c = 0; c < nz; c++) N[r * kNZ + c] = Q[r * K + 2 + c];
    }
  }
  t->K = K; t->nz = nz; t->R = 8 * K; t->mode = mode;
  for (int r = 0; r < K; r++) { for (int c = 0; c < nz; c++) t->Nt[c][r] = N[r * kNZ + c]; for (int u = 0; u < 3; u++) t->Pp[r][u] = Pp[r * 3 + u]; }
  // Th = Phi N, ThU = Phi Pp + PhiU
  for (int r = 0; r < 4 * K; r++) {
    for (int c = 0; c < nz; c++) { double v = 0; for (int a = 0; a < K; a++) v += Phi[r * K + a] * N[a * kNZ + c]; t->Th[r][c] = v; }
    for (int u = 0; u < 3; u++) { double v = PhiU[r * 3 + u]; for (int a = 0; a < K; a++) v += Phi[r * K + a] * Pp[a * 3 + u]; t->ThU[r][u] = v; }
  }
  for (int i = 0; i < K; i++) {
    for (int k = 0; k < 4; k++) {  // position control points (:441-447)
      for (int c = 0; c < nz; c++) { double v = 0; for (int j = 0; j < 4; j++) v += M4[j][k] * t->Th[4 * i + j][c]; t->B[4 * i + k][c] = v; }
      for (int u = 0; u < 3; u++) { double v = 0; for (int j = 0; j < 4; j++) v += M4[j][k] * t->ThU[4 * i + j][u]; t->U[4 * i + k][u] = v; }
    }
    for (int k = 0; k < 3; k++) {  // velocity control points (:456-461)
      for (int c = 0; c < nz; c++) { double v = 0; for (int j = 0; j < 3; j++) v += V3[j][k] * t->Th[4 * i + j][c]; t->B[4 * K + 3 * i + k][c] = v; }
      for (int u = 0; u < 3; u++) { double v = 0; for (int j = 0; j < 3; j++) v += V3[j][k] * t->ThU[4 * i + j][u]; t->U[4 * K + 3 * i + k][u] = v; }
    }
    for (int c = 0; c < nz; c++) t->B[7 * K + i][c] = 6 * T * t->Th[4 * i][c] + 2 * t->Th[4 * i + 1][c];  // :467-470
    for (int u = 0; u < 3; u++) t->U[7 * K + i][u] = 6 * T * t->ThU[4 * i][u] + 2 * t->ThU[4 * i + 1][u];
  }
  // end state maps: p_end = d_K, v_end = c_K, a_end = 2 b_K
  for (int c = 0; c < nz; c++) {
    double p = 0, v = 0, a = 0;
    for (int r = 0; r < K; r++) { p += ba[2][r] * N[r * kNZ + c]; v += ba[1][r] * N[r * kNZ + c]; a += 2 * ba[0][r] * N[r * kNZ + c]; }
    t->ep[c] = p; t->ev[c] = v; t->ea[c] = a;
  }
  for (int u = 0; u < 3; u++) {
    double p = bu[2][u], v = bu[1][u], a = 2 * bu[0][u];
    for (int r = 0; r < K; r++) { p += ba[2][r] * Pp[r * 3 + u]; v += ba[1][r] * Pp[r * 3 + u]; a += 2 * ba[0][r] * Pp[r * 3 + u]; }
    t->up[u] = p; t->uv[u] = v; t->ua[u] = a;
    t->res_u[0][u] = a / 2; t->res_u[1][u] = v;
  }
  if (nz > 0) {  // Zp = (Th'Th)^-1 Th' by Gauss-Jordan with partial pivoting on the nz x nz Gram matrix
    double G[kNZ][kNZ + 4 * kMaxK];
    const int nr = 4 * K, w = nz + nr;
    for (int a = 0; a < nz; a++) {
      for (int b = 0; b < nz; b++) { double v = 0; for (int r = 0; r < nr; r++) v += t->Th[r][a] * t->Th[r][b]; G[a][b] = v; }
      for (int r = 0; r < nr; r++) G[a][nz + r] = t->Th[r][a];
    }
    for (int c = 0; c < nz; c++) {
      int piv = c; for (int r = c + 1; r < nz; r++) if (std::fabs(G[r][c]) > std::fabs(G[piv][c])) piv = r;
      if (piv != c) for (int j = 0; j < w; j++) std::swap(G[c][j], G[piv][j]);
      const double d = G[c][c];
      for (int j = 0; j < w; j++) G[c][j] /= d;
      for (int r = 0; r < nz; r++) if (r != c) { const double f = G[r][c]; if (f != 0.0) for (int j = 0; j < w; j++) G[r][j] -= f * G[c][j]; }
    }
    for (int a = 0; a < nz; a++) for (int r = 0; r < nr; r++) t->Zp[a][r] = G[a][nz + r];
  }
  // cost: 36T |a|^2 + w (p_end - f)^2 [+ w v_end^2 + w a_end^2]   (:322-383)
  for (int a = 0; a < nz; a++) {
    for (int b = 0; b < nz; b++) {
      double nn = 0; for (int r = 0; r < K; r++) nn += N[r * kNZ + a] * N[r * kNZ + b];
      double h = 72 * T * nn + 2 * weight * t->ep[a] * t->ep[b];
      if (mode == 1) h += 2 * weight * (t->ev[a] * t->ev[b] + t->ea[a] * t->ea[b]);
      t->Hax[a][b] = h;
    }
    for (int u = 0; u < 3; u++) {
      double np_ = 0; for (int r = 0; r < K; r++) np_ += N[r * kNZ + a] * Pp[r * 3 + u];
      double g = 72 * T * np_ + 2 * weight * t->ep[a] * t->up[u];
      if (mode == 1) g += 2 * weight * (t->ev[a] * t->uv[u] + t->ea[a] * t->ua[u]);
      t->Gi[a][u] = g;
    }
  }
  if (nz > 0) {  // HaxInv by Gauss-Jordan with partial pivoting (Hax is symmetric positive definite)
    double G[kNZ][2 * kNZ];
    for (int a = 0; a < nz; a++) for (int b = 0; b < nz; b++) { G[a][b] = t->Hax[a][b]; G[a][nz + b] = (a == b) ? 1.0 : 0.0; }
    for (int c = 0; c < nz; c++) {
      int piv = c; for (int r = c + 1; r < nz; r++) if (std::fabs(G[r][c]) > std::fabs(G[piv][c])) piv = r;
      if (piv != c) for (int j = 0; j < 2 * nz; j++) std::swap(G[c][j], G[piv][j]);
      const double d = G[c][c];
      for (int j = 0; j < 2 * nz; j++) G[c][j] /= d;
      for (int r = 0; r < nz; r++) if (r != c) { const double f = G[r][c]; if (f != 0.0) for (int j = 0; j < 2 * nz; j++) G[r][j] -= f * G[c][j]; }
    }
    for (int a = 0; a < nz; a++) for (int b = 0; b < nz; b++) t->HaxInv[a][b] = G[a][nz + b];
  }
  // the unconstrained minimiser as a linear map of v = (b0, c0, d0, f): z* = Zm v (see RowMap / ThMap / ObjQ)
  {
    double Zm[kNZ][4] = {{0}}, Gm[kNZ][4] = {{0}};
    for (int c = 0; c < nz; c++) {
      for (int u = 0; u < 3; u++) Gm[c][u] = t->Gi[c][u];
      Gm[c][3] = -2 * weight * t->ep[c];
    }
    for (int c = 0; c < nz; c++) for (int m = 0; m < 4; m++) { double v = 0; for (int e = 0; e < nz; e++) v -= t->HaxInv[c][e] * Gm[e][m]; Zm[c][m] = v; }
    for (int rho = 0; rho < 8 * K; rho++) for (int m = 0; m < 4; m++) { double v = m < 3 ? t->U[rho][m] : 0.0; for (int c = 0; c < nz; c++) v += t->B[rho][c] * Zm[c][m]; t->RowMap[rho][m] = v; }
    for (int r = 0; r < 4 * K; r++) for (int m = 0; m < 4; m++) { double v = m < 3 ? t->ThU[r][m] : 0.0; for (int c = 0; c < nz; c++) v += t->Th[r][c] * Zm[c][m]; t->ThMap[r][m] = v; }
    // cost of an axis at z*: sum_r 36 T (Pp_r . init)^2 + w (up . init - f)^2 + z*'(0.5 Hax z* + g), g = Gm v   (the first problem's cost, :322-383)
    for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) {
      double q = 0;
      for (int r = 0; r < K; r++) q += 36 * T * (a < 3 ? Pp[r * 3 + a] : 0.0) * (b < 3 ? Pp[r * 3 + b] : 0.0);
      q += weight * (a < 3 ? t->up[a] : -1.0) * (b < 3 ? t->up[b] : -1.0);
      for (int c = 0; c < nz; c++) { double hz = 0; for (int e = 0; e < nz; e++) hz += t->Hax[c][e] * Zm[e][b]; q += Zm[c][a] * (0.5 * hz + Gm[c][b]); }
      t->ObjQ[a][b] = q;
    }
  }
}

// generatePwpOut's walk over time: sample s uses segment seg[s] at local time dt[s].
inline int build_sample_schedule(int K, double T, double dc, int cap, int* seg, double* dt) {
  double _t = 0; int i = 0, n = 0;
  while (i < K && n < cap) {
    seg[n] = i; dt[n] = _t - i * T; n++;
    _t += dc;
    if (_t > (i + 1) * T) i++;
  }
  return n;
}

}  // namespace nep
