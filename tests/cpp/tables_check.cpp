// Host-only check of the reduced-QP tables (neptune_amd/csrc/nep_tables.h): for every (K, mode)
//   * Zp Th = I            (Zp is the left inverse the start point uses)
//   * theta = Th z + ThU init is C2-continuous, starts at init and, in mode 0, ends with v = a = 0
//   * projecting a feasible theta gives its z back
//   * Hax HaxInv = I
// Prints "ok" or the first violation; tests/test_abi.py compiles and runs it (no GPU, no HIP).
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../neptune_amd/csrc/nep_tables.h"

int main() {
  const double T = 0.5, w = 1000.0;
  double worst = 0;
  for (int mode = 0; mode < 2; mode++) for (int K = 1; K <= nep::kMaxK; K++) {
    nep::QpTable t; nep::build_qp_table(K, T, w, mode, &t);
    const int nz = t.nz;
    for (int a = 0; a < nz; a++) for (int b = 0; b < nz; b++) {
      double v = 0; for (int r = 0; r < 4 * K; r++) v += t.Zp[a][r] * t.Th[r][b];
      const double e = std::fabs(v - (a == b ? 1.0 : 0.0));
      if (e > worst) worst = e;
      if (e > 1e-9) { std::printf("Zp Th != I at K=%d mode=%d (%d,%d): %g\n", K, mode, a, b, v); return 1; }
    }
    for (int a = 0; a < nz; a++) for (int b = 0; b < nz; b++) {   // HaxInv is the inverse of Hax (the presolve's unconstrained minimiser)
      double v = 0; for (int c = 0; c < nz; c++) v += t.Hax[a][c] * t.HaxInv[c][b];
      if (std::fabs(v - (a == b ? 1.0 : 0.0)) > 1e-9) { std::printf("Hax HaxInv != I at K=%d mode=%d (%d,%d): %g\n", K, mode, a, b, v); return 1; }
    }
    // a feasible theta from a pseudo-random z and init
    double z[nep::kNZ], init[3] = {0.3, -1.1, 2.0}, th[4 * nep::kMaxK];
    unsigned s = 12345u + 17u * K + mode;
    for (int c = 0; c < nz; c++) { s = s * 1664525u + 1013904223u; z[c] = ((s >> 8) % 2001) / 1000.0 - 1.0; }
    for (int r = 0; r < 4 * K; r++) { double v = t.ThU[r][0] * init[0] + t.ThU[r][1] * init[1] + t.ThU[r][2] * init[2]; for (int c = 0; c < nz; c++) v += t.Th[r][c] * z[c]; th[r] = v; }
    if (std::fabs(th[1] - init[0]) > 1e-12 || std::fabs(th[2] - init[1]) > 1e-12 || std::fabs(th[3] - init[2]) > 1e-12) { std::printf("init state not reproduced K=%d mode=%d\n", K, mode); return 1; }
    for (int i = 0; i + 1 < K; i++) {   // continuity of position, velocity, acceleration at the knots (solver_gurobi_poly.cpp:400-425)
      const double* c0 = th + 4 * i; const double* c1 = th + 4 * (i + 1);
      const double p = ((c0[0] * T + c0[1]) * T + c0[2]) * T + c0[3], v = (3 * c0[0] * T + 2 * c0[1]) * T + c0[2], a = 6 * c0[0] * T + 2 * c0[1];
      if (std::fabs(p - c1[3]) > 1e-9 || std::fabs(v - c1[2]) > 1e-9 || std::fabs(a - 2 * c1[1]) > 1e-9) { std::printf("continuity broken K=%d mode=%d seg=%d\n", K, mode, i); return 1; }
    }
    if (mode == 0 && nz > 0) {
      const double* c0 = th + 4 * (K - 1);
      const double v = (3 * c0[0] * T + 2 * c0[1]) * T + c0[2], a = 6 * c0[0] * T + 2 * c0[1];
      if (std::fabs(v) > 1e-9 || std::fabs(a) > 1e-9) { std::printf("terminal v/a not zero K=%d: %g %g\n", K, v, a); return 1; }
    }
    for (int c = 0; c < nz; c++) {   // projection gives z back
      double v = 0;
      for (int r = 0; r < 4 * K; r++) v += t.Zp[c][r] * (th[r] - (t.ThU[r][0] * init[0] + t.ThU[r][1] * init[1] + t.ThU[r][2] * init[2]));
      if (std::fabs(v - z[c]) > 1e-8) { std::printf("projection does not return z K=%d mode=%d c=%d: %g vs %g\n", K, mode, c, v, z[c]); return 1; }
    }
  }
  std::printf("ok %g\n", worst);
  return 0;
}
