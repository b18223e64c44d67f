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

}  // namespace geo
