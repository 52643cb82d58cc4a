// setup_host.cpp — one-time, host-only set-up steps of the path behind the C ABI (no HIP call; the reference runs them on the
// host in its constructors as well).
//
// nep_inflate_static: Neptune::setStaticObst (reference neptune/src/neptune.cpp:639-664).  Every vertex of a static
// obstacle's footprint is pushed out to the four corners (+-safe_dist, +-safe_dist), safe_dist = 2 drone_radius + 0.2, and
// the convex hull of those points (cu::convexHullOfPoints2d -> CGAL::convex_hull_2, cgal_utils.cpp:157-174: the extreme points
// in counter-clockwise order, starting from the lexicographically smallest, collinear points dropped) is what
// setStaticObstVert hands to the front end and the back end.
#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "../../include/neptune_backend.h"

namespace nep { void set_last_error(const std::string& msg); }

namespace {
typedef std::pair<double, double> P2;
double orient(const P2& o, const P2& a, const P2& b) { return (a.first - o.first) * (b.second - o.second) - (a.second - o.second) * (b.first - o.first); }

// Andrew's monotone chain over the lexicographically sorted, de-duplicated points: lower chain left to right, upper chain
// right to left; a point is kept only on a strict left turn
void hull_ccw(std::vector<P2> pts, std::vector<P2>& out) {
  std::sort(pts.begin(), pts.end());
  pts.erase(std::unique(pts.begin(), pts.end()), pts.end());
  out.clear();
  if (pts.size() <= 2) { out = pts; return; }
  std::vector<P2> ch(2 * pts.size());
  size_t k = 0;
  for (size_t i = 0; i < pts.size(); i++) { while (k >= 2 && orient(ch[k - 2], ch[k - 1], pts[i]) <= 0) k--; ch[k++] = pts[i]; }
  for (size_t i = pts.size() - 1, lo = k + 1; i-- > 0;) { while (k >= lo && orient(ch[k - 2], ch[k - 1], pts[i]) <= 0) k--; ch[k++] = pts[i]; }
  out.assign(ch.begin(), ch.begin() + (k - 1));
}
}  // namespace

extern "C" int nep_inflate_static(int32_t n_obst, const int32_t* vert_off, const double* xy, double drone_radius, int32_t* out_off,
                                  double* out_xy, int32_t cap) {
  if (n_obst < 0 || !out_off || (n_obst > 0 && (!vert_off || !xy)) || cap < 0 || (cap > 0 && !out_xy)) { nep::set_last_error("bad arguments"); return NEP_E_ARG; }
  const double sd = 2 * drone_radius + 0.2;      // neptune.cpp:642
  std::vector<P2> pts, hull;
  int total = 0;
  out_off[0] = 0;
  for (int j = 0; j < n_obst; j++) {
    const int nv = vert_off[j + 1] - vert_off[j];
    if (nv < 0) { nep::set_last_error("static obstacle offsets must not decrease"); return NEP_E_ARG; }
    pts.clear();
    for (int v = 0; v < nv; v++) {
      const double x = xy[2 * (vert_off[j] + v)], y = xy[2 * (vert_off[j] + v) + 1];
      pts.emplace_back(x + sd, y + sd); pts.emplace_back(x + sd, y - sd); pts.emplace_back(x - sd, y - sd); pts.emplace_back(x - sd, y + sd);   // :648-655
    }
    hull_ccw(pts, hull);
    if ((int)hull.size() > NEP_HULL_MAX_V) { nep::set_last_error("inflated static obstacle with more than NEP_HULL_MAX_V vertices"); return NEP_E_CAP; }
    if (total + (int)hull.size() > cap) { nep::set_last_error("output capacity too small for the inflated polygons"); return NEP_E_CAP; }
    for (const P2& q : hull) { out_xy[2 * total] = q.first; out_xy[2 * total + 1] = q.second; total++; }
    out_off[j + 1] = total;
  }
  return total;
}
