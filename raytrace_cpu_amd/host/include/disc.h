// include/disc.h -- proper area of accretion-disc annuli in the frame of the orbiting material
// (API of the reference's src/include/disc.h: rel_vector_disc_area :12, rel_vector_disc_area_plunge :37,
// ..._varradius :84, integrate_disc_area :135, integrate_disc_area_varplungeradius :154).
//
// The emissivity applications divide their per-annulus sums by integrate_disc_area(r_i, r_i * dr, spin)
// (emissivity.cpp:76-79, :131-132), so the second column of their output tables comes from here.
//
// A patch (dr, dphi) of the equatorial plane at radius r is projected on the spatial legs of the local observer's
// orthonormal frame and the area of the parallelogram is taken there.  Outside the ISCO the observer is on a circular
// orbit (tetrad(), kerr.h); inside, on the plunging geodesic that left the ISCO (or `r_plunge`) with its energy and
// angular momentum, with a Gram-Schmidt frame.  The reference evaluates the plunge constants with sqrtf (:48-56);
// so does this.
#ifndef CUDAKERR_DISC_H
#define CUDAKERR_DISC_H

#include <cmath>

#include "gramschmidt_basis.h"
#include "kerr.h"

namespace krhost {

// area of the patch spanned by (0, dr, 0, 0) and (0, 0, 0, dphi) in the frame (et, e1, e2, e3) under metric g
template <typename T>
T patch_area_in_frame(T (*g)[4], T* et, T* e1, T* e2, T* e3, T dr, T dphi)
{
    T side_r[] = {0, dr, 0, 0};
    T side_phi[] = {0, 0, 0, dphi};
    T* legs[4] = {et, e1, e2, e3};
    T a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = dot_product(g, side_r, legs[i]);
        b[i] = dot_product(g, side_phi, legs[i]);
    }
    const T cx = a[2] * b[3] - a[3] * b[2];
    const T cy = a[3] * b[1] - a[1] * b[3];
    const T cz = a[1] * b[2] - a[2] * b[1];
    return std::sqrt(cx * cx + cy * cy + cz * cz);
}

// 4-velocity of the geodesic plunging from a circular orbit at r_from (energy k, angular momentum h of that orbit)
template <typename T>
void plunge_velocity(T* vel, T r, T a, T r_from)
{
    const T delta = r * r - 2 * r + a * a;
    const T u = 1 / r_from;
    const T k = (1 - 2 * u + a * u * sqrtf(u)) / sqrtf(1 - 3 * u + 2 * a * u * sqrtf(u));
    const T h = (1 + a * a * u * u - 2 * a * u * sqrtf(u)) / sqrtf(u * (1 - 3 * u + 2 * a * u * sqrtf(u)));
    vel[0] = (1 / delta) * ((r * r + a * a + 2 * a * a / r) * k - 2 * a * h / r);
    vel[1] = -1 * sqrtf(k * k - 1 + 2 / r + (a * a * (k * k - 1) - h * h) / (r * r) + 2 * (h - a * k) * (h - a * k) / (r * r * r));
    vel[2] = 0;
    vel[3] = (1 / delta) * (2 * a * k / r + (1 - 2 / r) * h);
    if (!(std::fabs(vel[1]) > 0)) vel[1] = 0;   // rounding pushed the radicand below zero: still on the circular orbit
}

template <typename T>
T plunging_patch_area(T r, T dr, T dphi, T a, T r_from)
{
    T g[4][4], vel[4];
    T pos[] = {0, r, M_PI / 2, 0};
    plunge_velocity(vel, r, a, r_from);
    double posd[4] = {0, static_cast<double>(r), M_PI / 2, 0}, veld[4] = {(double) vel[0], (double) vel[1], (double) vel[2], (double) vel[3]};
    GramSchmidt_Basis<double> frame(posd, veld, static_cast<double>(a));
    T legs[4][4];
    for (int i = 0; i < 4; ++i)
        for (int c = 0; c < 4; ++c) legs[i][c] = static_cast<T>(frame.vectors[i][c]);
    kerr_metric(g, pos, a);
    return patch_area_in_frame(g, legs[0], legs[1], legs[2], legs[3], dr, dphi);
}

}   // namespace krhost

// material on circular orbits
template <typename T>
T rel_vector_disc_area(T r, T dr, T dphi, T a)
{
    T et[4], e1[4], e2[4], e3[4], g[4][4];
    T pos[] = {0, r, M_PI / 2, 0};
    const T V = disc_velocity(r, a, +1);
    tetrad(et, e1, e2, e3, pos, V, a);
    kerr_metric(g, pos, a);
    return krhost::patch_area_in_frame(g, et, e1, e2, e3, dr, dphi);
}

// material plunging from the ISCO
template <typename T>
T rel_vector_disc_area_plunge(T r, T dr, T dphi, T a)
{
    return krhost::plunging_patch_area(r, dr, dphi, a, kerr_isco(a, +1));
}

// ... or from r_plunge (< 0: the ISCO)
template <typename T>
T rel_vector_disc_area_plunge_varradius(T r, T dr, T dphi, T a, T r_plunge = -1)
{
    return krhost::plunging_patch_area(r, dr, dphi, a, r_plunge < 0 ? kerr_isco(a, +1) : r_plunge);
}

// area of the annulus [rmin, rmax) per dphi: Nr - 1 sub-annuli (log- or linearly spaced), non-positive / NaN ones skipped
template <typename T>
T integrate_disc_area_varplungeradius(T rmin, T rmax, T a, T r_plunge = -1, int Nr = 50, T dphi = 0.1, bool logbin_r = true)
{
    const T step = logbin_r ? std::exp(std::log(rmax / rmin) / (Nr - 1)) : (rmax - rmin) / (Nr - 1);
    if (r_plunge < 0) r_plunge = kerr_isco(a, +1);
    T total = 0;
    for (T r = rmin; r < rmax; r = logbin_r ? r * step : r + step) {
        const T width = logbin_r ? r * (step - 1) : step;
        const T piece = (r >= r_plunge) ? rel_vector_disc_area(r, width, dphi, a) : rel_vector_disc_area_plunge_varradius(r, width, dphi, a, r_plunge);
        if (piece > 0) total += piece;
    }
    return total;
}

template <typename T>
T integrate_disc_area(T rmin, T rmax, T a, bool force_keplerian = false, int Nr = 50, T dphi = 0.1, bool logbin_r = true)
{
    // force_keplerian: circular orbits everywhere, i.e. a plunge radius below any r
    return integrate_disc_area_varplungeradius(rmin, rmax, a, force_keplerian ? T(0) : kerr_isco(a, +1), Nr, dphi, logbin_r);
}

#endif /* CUDAKERR_DISC_H */
