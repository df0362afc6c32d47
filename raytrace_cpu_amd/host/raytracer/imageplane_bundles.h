// raytracer/imageplane_bundles.h -- image plane that traces a 5-ray bundle per pixel (centre, east, west, north, south)
// so that callers can form the local Jacobian of the lens map by central differences over a sub-pixel offset.
//
// API contract of the reference's src/raytracer/imageplane_bundles.h:45-199 (header-only there as well): public Nx, Ny,
// eps_x, eps_y, RAYS_PER_BUNDLE, the constructor with its defaults, the five *_ray(ix, iy) index helpers,
// redshift_start() and the using-declaration for redshift().  Ray initialisation is ImagePlane's (same dy-for-x
// quirk) except that the on-axis pixel takes beta = 0 instead of asin(0/0).  All 5 * Nx * Ny rays are integrated in one
// GPU launch by the base class; the caustic applications of the reference (src/caustic/*.cpp) build against this
// header unchanged (dropin/build_apps.sh).
#ifndef IMAGEPLANE_BUNDLES_H_
#define IMAGEPLANE_BUNDLES_H_

#include <cmath>
#include <iostream>

#include "ray_destination.h"
#include "raytracer.h"

template <typename T>
class ImagePlaneBundles : public Raytracer<T> {
public:
    static constexpr int RAYS_PER_BUNDLE = 5;

    int Nx, Ny;        // bundle centres along each axis (fencepost counts)
    T eps_x, eps_y;    // satellite offsets in image-plane units

    ImagePlaneBundles(T dist, T inc_deg, T x0, T xmax, T dx, T y0, T ymax, T dy, T spin, T phi, T precision = PRECISION, T eps_frac = 0.01)
        : Raytracer<T>(int((((xmax - x0) / dx) + 1) * (((ymax - y0) / dy) + 1)) * RAYS_PER_BUNDLE, -1 * spin, precision),
          Nx(int(((xmax - x0) / dx) + 1)),
          Ny(int(((ymax - y0) / dy) + 1)),
          eps_x(eps_frac * dx),
          eps_y(eps_frac * dy),
          plane_dist(dist),
          plane_incl(inc_deg * M_PI / 180.0),
          plane_phi0(phi),
          origin_x(x0),
          origin_y(y0),
          pitch(dy)
    {
        static const int off_x[RAYS_PER_BUNDLE] = {0, +1, -1, 0, 0};
        static const int off_y[RAYS_PER_BUNDLE] = {0, 0, 0, +1, -1};
        for (int i = 0; i < Nx; i++) {
            const T x = origin_x + i * pitch;          // dy along x as well: ImagePlane's convention (imageplane.cpp:43)
            for (int j = 0; j < Ny; j++) {
                const T y = origin_y + j * pitch;
                const int base = (i * Ny + j) * RAYS_PER_BUNDLE;
                for (int m = 0; m < RAYS_PER_BUNDLE; m++) {
                    const T xs = (off_x[m] == 0) ? x : (off_x[m] > 0 ? x + eps_x : x - eps_x);
                    const T ys = (off_y[m] == 0) ? y : (off_y[m] > 0 ? y + eps_y : y - eps_y);
                    place_ray(base + m, xs, ys);
                }
            }
        }
    }

    inline int centre_ray(int ix, int iy) const { return (ix * Ny + iy) * RAYS_PER_BUNDLE + 0; }
    inline int east_ray(int ix, int iy) const { return (ix * Ny + iy) * RAYS_PER_BUNDLE + 1; }
    inline int west_ray(int ix, int iy) const { return (ix * Ny + iy) * RAYS_PER_BUNDLE + 2; }
    inline int north_ray(int ix, int iy) const { return (ix * Ny + iy) * RAYS_PER_BUNDLE + 3; }
    inline int south_ray(int ix, int iy) const { return (ix * Ny + iy) * RAYS_PER_BUNDLE + 4; }

    void redshift_start() { Raytracer<T>::redshift_start(0, true); }
    using Raytracer<T>::redshift;

private:
    T plane_dist, plane_incl, plane_phi0;
    T origin_x, origin_y, pitch;

    // one ray through image-plane point (x, y): position, line-of-sight momentum, null condition, impact-parameter constants
    void place_ray(int ix, T x, T y)
    {
        const T a = Raytracer<T>::spin;
        const T D = plane_dist, incl = plane_incl;
        Ray<T>& R = Raytracer<T>::rays[ix];

        const T r = sqrt(D * D + x * x + y * y);
        const T theta = acos((D * cos(incl) + y * sin(incl)) / r);
        const T phi = plane_phi0 + atan2(x, D * sin(incl) - y * cos(incl));
        const T pr = D / r;
        const T ptheta = sin(acos(D / r)) / r;
        const T pphi = x * sin(incl) / (x * x + (D * sin(incl) - y * cos(incl)) * (D * sin(incl) - y * cos(incl)));

        const krhost::BLCoefficients<T> m(r, theta, a);
        const T g00 = m.e2nu - m.omega * m.omega * m.e2psi;
        const T g03 = m.omega * m.e2psi;
        const T g11 = -m.rhosq / m.delta;
        const T g22 = -m.rhosq;
        const T g33 = -m.e2psi;
        const T A = g00;
        const T B = 2 * g03 * pphi;
        const T C = g11 * pr * pr + g22 * ptheta * ptheta + g33 * pphi * pphi;
        T pt = (-B + sqrt(B * B - 4 * A * C)) / (2 * A);
        if (pt < 0) pt = (-B - sqrt(B * B - 4 * A * C)) / (2 * A);

        R.t = 0;
        R.r = r;
        R.theta = theta;
        R.phi = phi;
        R.pt = pt;
        R.pr = pr;
        R.ptheta = ptheta;
        R.pphi = pphi;
        Raytracer<T>::calculate_constants_from_p(ix, pt, pr, ptheta, pphi);
        R.rdot_sign = -1;
        R.k = 1;

        const T b = sqrt(x * x + y * y);
        T beta_ang = (b > 0) ? asin(y / b) : 0;
        if (x < 0) beta_ang = M_PI - beta_ang;
        const T h = -b * sin(incl) * cos(beta_ang);
        const T ltheta = b * sin(beta_ang);
        R.h = h;
        R.Q = ltheta * ltheta - (a * cos(theta)) * (a * cos(theta)) + (h / tan(theta)) * (h / tan(theta));
        R.thetadot_sign = (ltheta >= 0) ? 1 : -1;
        R.steps = 0;
        R.alpha = x;
        R.beta = y;
    }
};

#endif /* IMAGEPLANE_BUNDLES_H_ */
