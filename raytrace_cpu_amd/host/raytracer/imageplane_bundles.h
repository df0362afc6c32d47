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
#include "image_ray.h"

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

    // one ray through image-plane point (x, y): krhost::camera_ray with this class's conventions (spin in T, the centre point guarded)
    void place_ray(int ix, T x, T y)
    {
        const T a = Raytracer<T>::spin;
        const krhost::CameraRay<T> c = krhost::camera_ray<T, T, false, true>(plane_dist, plane_incl, plane_phi0, x, y, a);
        Ray<T>& R = Raytracer<T>::rays[ix];
        R.t = 0;
        R.r = c.r; R.theta = c.theta; R.phi = c.phi;
        R.pt = c.pt; R.pr = c.pr; R.ptheta = c.ptheta; R.pphi = c.pphi;
        Raytracer<T>::calculate_constants_from_p(ix, c.pt, c.pr, c.ptheta, c.pphi);
        R.rdot_sign = -1;
        R.k = 1;
        R.h = c.h;
        R.Q = c.Q;
        R.thetadot_sign = c.thetadot_sign;
        R.steps = 0;
        R.alpha = x;
        R.beta = y;
    }
};

#endif /* IMAGEPLANE_BUNDLES_H_ */
