// host_geodesy.hpp -- downstream of the hot path, host side: WGS-84 conversions and
// the reference's 3-station TDOA solver, kept call-compatible (processor.go:125-163,
// 932-1045) so the peaks this library returns feed the same least-squares step.
#pragma once

#include <cmath>

namespace geo {

constexpr double kA = 6378137.0;                 // WGS-84 semi-major axis
constexpr double kF = 1.0 / 298.257223563;       // flattening
constexpr double kE2 = 2 * kF - kF * kF;         // first eccentricity squared
constexpr double kPi = 3.14159265358979323846;
constexpr double kC = 299792458.0;               // processor.go:873

inline double prime_vertical(double sin_lat) { return kA / std::sqrt(1 - kE2 * sin_lat * sin_lat); }

inline void latlon_to_ecef(double lat, double lon, double elev, double xyz[3])
{
    const double phi = lat * kPi / 180, lam = lon * kPi / 180;
    const double sp = std::sin(phi), cp = std::cos(phi);
    const double nu = prime_vertical(sp);
    xyz[0] = (nu + elev) * cp * std::cos(lam);
    xyz[1] = (nu + elev) * cp * std::sin(lam);
    xyz[2] = (nu * (1 - kE2) + elev) * sp;
}

inline void ecef_to_latlon(double x, double y, double z, double lle[3])
{
    const double p = std::sqrt(x * x + y * y);
    double phi = std::atan2(z, p * (1 - kE2));
    double h = 0;
    for (int it = 0; it < 6; it++) {             // 5 refinements + the final evaluation
        const double nu = prime_vertical(std::sin(phi));
        h = p / std::cos(phi) - nu;
        if (it < 5) phi = std::atan2(z, p * (1 - kE2 * nu / (nu + h)));
    }
    lle[0] = phi * 180.0 / kPi;
    lle[1] = std::atan2(y, x) * 180.0 / kPi;
    lle[2] = h;
}

inline double range(const double a[3], const double b[3])
{
    const double d0 = a[0] - b[0], d1 = a[1] - b[1], d2 = a[2] - b[2];
    return std::sqrt(d0 * d0 + d1 * d1 + d2 * d2);
}

// Damped (0.5) Newton on the two range-difference equations (0,1) and (0,2) in ECEF
// X,Y with Z held at the centroid's value; 10 iterations; stops below 1 m.
// returns 0 ok, -1 singular Jacobian.
inline int solve_3station(const double st[9], const double *rd, double out[3], int *iters)
{
    double s[3][3], x[3];
    for (int i = 0; i < 3; i++) latlon_to_ecef(st[3 * i], st[3 * i + 1], st[3 * i + 2], s[i]);
    latlon_to_ecef((st[0] + st[3] + st[6]) / 3.0, (st[1] + st[4] + st[7]) / 3.0, (st[2] + st[5] + st[8]) / 3.0, x);
    int it = 0;
    for (; it < 10; it++) {
        double r[3], ux[3], uy[3];
        for (int i = 0; i < 3; i++) {
            r[i] = range(x, s[i]);
            ux[i] = (x[0] - s[i][0]) / r[i];
            uy[i] = (x[1] - s[i][1]) / r[i];
        }
        const double f1 = (r[1] - r[0]) - rd[0], f2 = (r[2] - r[0]) - rd[1];
        if (std::fabs(f1) < 1.0 && std::fabs(f2) < 1.0) break;
        const double a = ux[1] - ux[0], b = uy[1] - uy[0], c = ux[2] - ux[0], d = uy[2] - uy[0];
        const double det = a * d - b * c;
        if (std::fabs(det) < 1e-10) {
            if (iters) *iters = it;
            return -1;
        }
        x[0] += 0.5 * ((-f1 * d + f2 * b) / det);
        x[1] += 0.5 * ((f1 * c - f2 * a) / det);
    }
    if (iters) *iters = it;
    ecef_to_latlon(x[0], x[1], x[2], out);
    return 0;
}

// N-station generalisation: Gauss-Newton over ALL pair range differences
// rd[p], pairs ordered i<j like processor.go:816-817, optional weights (e.g. |corr|), unknowns
// ECEF X,Y (Z frozen at the centroid like the reference, processor.go:1004) or X,Y,Z when
// solve_z is set.  Same start (centroid), damping (0.5), iteration cap (10) and 1 m stop rule as
// the reference; with n = 3 and weights {1,1,0} it solves the reference's own 2x2 system.
// A pair with weight 0 is skipped; a negative or non-finite weight (or range difference of a used pair) is an
// error, and so is a weight set that leaves fewer usable pairs than unknowns -- with nothing to fit the
// iteration would "converge" at the centroid after 0 steps.
// returns 0 ok, -1 singular normal matrix, -2 unsupported station count, -3 too few usable pairs / bad weights.
inline int solve_nstation(const double *st_lle, int n, const double *rd, const double *wt, int solve_z,
                          int max_iter, double damping, double tol_m, double out[3], int *iters)
{
    if (n < 3 || n > 64) return -2;
    double s[64][3], x[3], c[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        latlon_to_ecef(st_lle[3 * i], st_lle[3 * i + 1], st_lle[3 * i + 2], s[i]);
        for (int k = 0; k < 3; k++) c[k] += st_lle[3 * i + k] / n;
    }
    latlon_to_ecef(c[0], c[1], c[2], x);
    const int nu = solve_z ? 3 : 2;
    {
        int used = 0, p = 0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++, p++) {
                const double w = wt ? wt[p] : 1.0;
                if (!(w >= 0) || !std::isfinite(w)) return -3;        // negative or NaN / inf
                if (w == 0) continue;
                if (!std::isfinite(rd[p])) return -3;
                used++;
            }
        if (used < nu) return -3;
    }
    int it = 0;
    for (; it < max_iter; it++) {
        double r[64], u[64][3];
        for (int i = 0; i < n; i++) {
            r[i] = range(x, s[i]);
            for (int k = 0; k < 3; k++) u[i][k] = (x[k] - s[i][k]) / r[i];
        }
        double A[3][3] = {{0}}, g[3] = {0, 0, 0}, worst = 0;
        int p = 0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++, p++) {
                const double w = wt ? wt[p] : 1.0;
                if (w == 0) continue;
                const double f = (r[j] - r[i]) - rd[p];
                worst = std::fmax(worst, std::fabs(f));
                double jr[3];
                for (int k = 0; k < 3; k++) jr[k] = u[j][k] - u[i][k];
                for (int a = 0; a < nu; a++) {
                    g[a] += w * jr[a] * f;
                    for (int b = 0; b < nu; b++) A[a][b] += w * jr[a] * jr[b];
                }
            }
        if (worst < tol_m) break;
        double d[3] = {0, 0, 0};
        if (nu == 2) {
            const double det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
            if (std::fabs(det) < 1e-20) { if (iters) *iters = it; return -1; }
            d[0] = (-g[0] * A[1][1] + g[1] * A[0][1]) / det;
            d[1] = (g[0] * A[1][0] - g[1] * A[0][0]) / det;
        } else {
            const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                               A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
            if (std::fabs(det) < 1e-30) { if (iters) *iters = it; return -1; }
            const double b0 = -g[0], b1 = -g[1], b2 = -g[2];
            d[0] = (b0 * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (b1 * A[2][2] - A[1][2] * b2) + A[0][2] * (b1 * A[2][1] - A[1][1] * b2)) / det;
            d[1] = (A[0][0] * (b1 * A[2][2] - A[1][2] * b2) - b0 * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) + A[0][2] * (A[1][0] * b2 - b1 * A[2][0])) / det;
            d[2] = (A[0][0] * (A[1][1] * b2 - b1 * A[2][1]) - A[0][1] * (A[1][0] * b2 - b1 * A[2][0]) + b0 * (A[1][0] * A[2][1] - A[1][1] * A[2][0])) / det;
        }
        for (int k = 0; k < 3; k++) x[k] += damping * d[k];
    }
    if (iters) *iters = it;
    ecef_to_latlon(x[0], x[1], x[2], out);
    return 0;
}

// Ground transmitter: the position constrained to the ellipsoid surface at height h0 (unknowns latitude, longitude).
// The reference freezes ECEF Z at the stations' centroid (processor.go:1004) -- a plane that does not contain a transmitter
// a few kilometres north or south of that centroid, so its fix is off by hundreds of metres to kilometres however good the
// delays are.  Same residuals and weights as solve_nstation, undamped Gauss-Newton from the centroid, stops below tol_m.
// returns 0 ok, -1 singular normal matrix, -2 unsupported station count, -3 too few usable pairs / bad weights.
inline int solve_surface(const double *st_lle, int n, const double *rd, const double *wt, double h0, int max_iter, double tol_m,
                         double out[3], int *iters)
{
    if (n < 3 || n > 64) return -2;
    double s[64][3], c[2] = {0, 0};
    for (int i = 0; i < n; i++) {
        latlon_to_ecef(st_lle[3 * i], st_lle[3 * i + 1], st_lle[3 * i + 2], s[i]);
        c[0] += st_lle[3 * i] / n;
        c[1] += st_lle[3 * i + 1] / n;
    }
    {
        int used = 0, p = 0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++, p++) {
                const double w = wt ? wt[p] : 1.0;
                if (!(w >= 0) || !std::isfinite(w)) return -3;
                if (w == 0) continue;
                if (!std::isfinite(rd[p])) return -3;
                used++;
            }
        if (used < 2) return -3;
    }
    double phi = c[0] * kPi / 180, lam = c[1] * kPi / 180;
    int it = 0;
    for (; it < max_iter; it++) {
        const double sp = std::sin(phi), cp = std::cos(phi), sl = std::sin(lam), cl = std::cos(lam);
        const double nu = prime_vertical(sp), den = 1 - kE2 * sp * sp;
        const double mer = kA * (1 - kE2) / (den * std::sqrt(den));      // meridional radius of curvature
        const double x[3] = {(nu + h0) * cp * cl, (nu + h0) * cp * sl, (nu * (1 - kE2) + h0) * sp};
        const double dphi[3] = {-(mer + h0) * sp * cl, -(mer + h0) * sp * sl, (mer + h0) * cp};
        const double dlam[3] = {-(nu + h0) * cp * sl, (nu + h0) * cp * cl, 0.0};
        double r[64], jp[64], jl[64];
        for (int i = 0; i < n; i++) {
            r[i] = range(x, s[i]);
            jp[i] = jl[i] = 0;
            for (int k = 0; k < 3; k++) {
                const double u = (x[k] - s[i][k]) / r[i];
                jp[i] += u * dphi[k];
                jl[i] += u * dlam[k];
            }
        }
        double A[2][2] = {{0, 0}, {0, 0}}, g[2] = {0, 0}, worst = 0;
        int p = 0;
        for (int i = 0; i < n; i++)
            for (int j = i + 1; j < n; j++, p++) {
                const double w = wt ? wt[p] : 1.0;
                if (w == 0) continue;
                const double f = (r[j] - r[i]) - rd[p];
                worst = std::fmax(worst, std::fabs(f));
                const double a = jp[j] - jp[i], b = jl[j] - jl[i];
                g[0] += w * a * f;
                g[1] += w * b * f;
                A[0][0] += w * a * a;
                A[0][1] += w * a * b;
                A[1][1] += w * b * b;
            }
        A[1][0] = A[0][1];
        const double det = A[0][0] * A[1][1] - A[0][1] * A[1][0];
        if (!(std::fabs(det) > 1e-6 * A[0][0] * A[1][1]) ) { if (iters) *iters = it; return -1; }
        const double d0 = (-g[0] * A[1][1] + g[1] * A[0][1]) / det, d1 = (g[0] * A[1][0] - g[1] * A[0][0]) / det;
        phi += d0;
        lam += d1;
        // (a least-squares fit does not drive the residuals of inconsistent delays to zero: stop on the step as well)
        if (worst < tol_m || std::hypot(d0 * (mer + h0), d1 * (nu + h0) * cp) < 1e-3) { it++; break; }
    }
    if (iters) *iters = it;
    out[0] = phi * 180.0 / kPi;
    out[1] = lam * 180.0 / kPi;
    out[2] = h0;
    return 0;
}

}  // namespace geo
