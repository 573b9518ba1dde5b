// bezier_kernels.hip — closed Bézier track through a level's control points (BASELINE config 5: CarRacing-Bezier adversary).
//
// Replaces, for a batch of levels, envs/box2d/bezier.py:get_bezier_curve (:56-82) as CarRacingBezier._create_track_bezier
// calls it (car_racing_bezier.py:297-303: rad = 0.2, edgy = 0.2, numpoints = 40): counter-clockwise sort of the control
// points around their centroid (:50-53), tangent angle at every point from the two adjacent chord directions (:69-77),
// one cubic Bézier segment per consecutive pair with inner control points at distance rad * |chord| along the tangents
// (:24-42), `numpoints` samples per segment (Bernstein form, :10-19).  All fp64 like the numpy statement.
//
// Mapping: one wavefront per level (a level has <= 64 control points; the adversary places <= 12): lane <-> control point
// for the sort and the angles, then the lanes stride over the n * numpoints samples.  The work is tiny and launch-bound;
// what matters is that 4096 levels are ONE launch instead of 4096 Python calls.  No Box2D here: the dynamics of the
// car have no oracle in this environment (parity unpinned).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "../../include/dcd_hip.h"
#include "dcd_internal.h"

namespace {

constexpr int MAXP = 64;

// np.mean(p, axis=0) of an [n,2] array: the reduction runs over the OUTER axis, i.e. the rows are added one after the
// other (numpy's pairwise scheme only applies along a contiguous reduced axis)
__device__ double column_sum(const double* a, int n) {
  double r = 0.;
  for (int i = 0; i < n; ++i) r += a[2 * i];
  return r;
}

__global__ void __launch_bounds__(64) bezier_track_kernel(const double* __restrict__ pts, const int32_t* __restrict__ npts, int nmax,
                                                          double rad, double edgy, int numpoints, double* __restrict__ xy,
                                                          double* __restrict__ ctrl, int32_t* __restrict__ n_out, int B) {
  __shared__ double px[MAXP + 1], py[MAXP + 1], ang[MAXP + 1], raw[2 * MAXP], key[MAXP], a1[MAXP];
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) return;
  const int n = min(max(npts[b], 0), min(nmax, MAXP));
  if (lane == 0 && n_out) n_out[b] = n >= 2 ? n * numpoints : 0;
  if (n < 2) return;
  const double* P = pts + (size_t)b * nmax * 2;
  if (lane < n) { raw[2 * lane] = P[2 * lane]; raw[2 * lane + 1] = P[2 * lane + 1]; }
  __syncthreads();
  // ccw_sort (:50-53): s = arctan2(dx, dy) about the centroid, ascending
  const double mx = column_sum(raw, n) / (double)n, my = column_sum(raw + 1, n) / (double)n;
  if (lane < n) key[lane] = atan2(raw[2 * lane] - mx, raw[2 * lane + 1] - my);
  __syncthreads();
  if (lane < n) {
    int r = 0;
    // keys within 1e-12 rad are a tie (points on one ray from the centroid: equal angles up to libm noise) -> input order
    for (int j = 0; j < n; ++j) r += (key[j] < key[lane] - 1e-12) || (!(key[j] > key[lane] + 1e-12) && j < lane);
    px[r] = raw[2 * lane]; py[r] = raw[2 * lane + 1];
  }
  __syncthreads();
  if (lane == 0) { px[n] = px[0]; py[n] = py[0]; }                  // closed: a = append(a, a[0])
  __syncthreads();
  // chord directions in [0, 2 pi) (:70-72)
  if (lane < n) {
    double a = atan2(py[lane + 1] - py[lane], px[lane + 1] - px[lane]);
    a1[lane] = a >= 0 ? a : a + 2 * M_PI;                           // (ang>=0)*ang + (ang<0)*(ang+2*pi)
  }
  __syncthreads();
  const double p = atan(edgy) / M_PI + .5;
  if (lane < n) {
    const double x1 = a1[lane], x2 = a1[(lane + n - 1) % n];         // ang2 = np.roll(ang, 1)
    ang[lane] = p * x1 + (1 - p) * x2 + ((fabs(x2 - x1) > M_PI) ? M_PI : 0.0);
  }
  __syncthreads();
  if (lane == 0) ang[n] = ang[0];
  __syncthreads();
  if (ctrl && lane <= n) {                                           // the third return value: [n+1, 3] = (x, y, angle)
    double* c = ctrl + ((size_t)b * (nmax + 1) + lane) * 3;
    c[0] = px[lane]; c[1] = py[lane]; c[2] = ang[lane];
  }
  // segments (:24-42) and their samples (:10-19)
  const double step = 1.0 / (double)(numpoints - 1);                 // np.linspace(0, 1, num): arange * step, last := 1
  double* out = xy + (size_t)b * nmax * numpoints * 2;
  for (int q = lane; q < n * numpoints; q += 64) {
    const int s = q / numpoints, j = q - s * numpoints;
    const double x0 = px[s], y0 = py[s], x3 = px[s + 1], y3 = py[s + 1];
    const double dx = x3 - x0, dy = y3 - y0;
    const double r = rad * sqrt(dx * dx + dy * dy);
    const double x1 = x0 + r * cos(ang[s]), y1 = y0 + r * sin(ang[s]);
    const double x2 = x3 + r * cos(ang[s + 1] + M_PI), y2 = y3 + r * sin(ang[s + 1] + M_PI);
    const double t = (j == numpoints - 1) ? 1.0 : (double)j * step, u = 1. - t;
    // binom(3,k) * t**k * (1-t)**(3-k), k = 0..3, accumulated in that order onto zeros
    const double b0 = 1.0 * 1.0 * pow(u, 3.0), b1 = 3.0 * t * (u * u), b2 = 3.0 * (t * t) * u, b3 = 1.0 * pow(t, 3.0) * 1.0;
    out[2 * q] = ((0.0 + b0 * x0) + b1 * x1 + b2 * x2) + b3 * x3;
    out[2 * q + 1] = ((0.0 + b0 * y0) + b1 * y1 + b2 * y2) + b3 * y3;
  }
}

}  // namespace

extern "C" int dcd_bezier_track(const double* points, const int32_t* n_points, int32_t max_points, int32_t B, double rad, double edgy,
                                int32_t numpoints, double* xy, double* control, int32_t* n_out, void* stream) {
  if (!points || !n_points || !xy || B <= 0 || max_points < 2 || max_points > MAXP || numpoints < 2)
    return dcd::fail("dcd_bezier_track: bad arguments (2 <= max_points <= %d, numpoints >= 2)", MAXP);
  hipLaunchKernelGGL(bezier_track_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, points, n_points, max_points, rad, edgy, numpoints,
                     xy, control, n_out, B);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return dcd::fail("dcd_bezier_track: %s", hipGetErrorString(e));
  return 0;
}
