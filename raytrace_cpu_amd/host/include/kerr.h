// include/kerr.h -- Kerr-spacetime helper templates of the host-side API mirror.
//
// Same names, argument meaning and arithmetic as the reference header src/include/kerr.h (function
// by function, cited below), so that application code written against the reference (`kerr_isco`,
// `cartesian`, `disc_velocity`, `momentum_from_consts`, and the metric/tetrad helpers that
// src/include/disc.h builds on) compiles unchanged against this tree.  The include guard is the
// reference's on purpose: whichever of the two headers a translation unit meets first provides the
// (identical) interface.  Written as a coefficient struct + thin wrappers rather than as free
// functions that each rebuild the coefficients.
#ifndef KERR_H_
#define KERR_H_

#include <cmath>

namespace krhost {

// Boyer-Lindquist metric functions at (r, theta) for spin a, in the (e2nu, e2psi, omega) form
template <typename T>
struct BLCoefficients {
    T rhosq, delta, sigmasq, e2nu, e2psi, omega;

    BLCoefficients() = default;       // (filled in by a caller whose spin has another type than T: raytracer/image_ray.h)
    BLCoefficients(T r, T theta, T a)
    {
        using std::cos;
        using std::sin;
        rhosq = r * r + (a * cos(theta)) * (a * cos(theta));
        delta = r * r - 2 * r + a * a;
        sigmasq = (r * r + a * a) * (r * r + a * a) - a * a * delta * sin(theta) * sin(theta);
        e2nu = rhosq * delta / sigmasq;
        e2psi = sigmasq * sin(theta) * sin(theta) / rhosq;
        omega = 2 * a * r / sigmasq;
    }
};

}  // namespace krhost

// outer (+1) / inner (-1) horizon radius.  Reference: kerr.h:14-20
template <typename T>
T kerr_horizon(T a, int sign = 1)
{
    using std::sqrt;
    return 1 + sign * sqrt((1 - a) * (1 + a));
}

// ISCO radius, prograde (sign = +1) or retrograde (-1).  Reference: kerr.h:23-32.
// The two intermediates are single precision there, and the closing square root is taken in single
// precision as well; applications use the result as a histogram edge, so that rounding is part of the
// interface and is kept.
template <typename T>
T kerr_isco(T a, int sign)
{
    const float z1 = 1. + std::pow(1. - a * a, 1. / 3.) * (std::pow(1. + a, 1. / 3.) + std::pow(1. - a, 1. / 3.));
    const float z2 = std::sqrt(3. * a * a + z1 * z1);
    const float under_root = (3 - z1) * (3 + z1 + 2 * z2);
    return 3 + z2 - sign * std::sqrt(under_root);
}

// angular velocity dphi/dt of a circular equatorial orbit.  Reference: kerr.h:35-38
template <typename T>
T disc_velocity(T r, T a, int sign)
{
    return 1 / (a + sign * std::pow(r, 3. / 2.));
}

// Boyer-Lindquist -> "Cartesian" (x, y, z).  Reference: kerr.h:41-56
template <typename T>
void cartesian(T& x, T& y, T& z, T r, T theta, T phi, T a)
{
    using std::cos;
    using std::sin;
    using std::sqrt;
    x = sqrt(r * r + a * a) * sin(theta) * cos(phi);
    y = sqrt(r * r + a * a) * sin(theta) * sin(phi);
    z = r * cos(theta);
}

// g_ij u^i v^j.  Reference: kerr.h:59-72
template <typename T>
T dot_product(T (*g)[4], T* u, T* v)
{
    T sum = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) sum += g[i][j] * u[i] * v[j];
    return sum;
}

// eta = diag(1, -1, -1, -1).  Reference: kerr.h:75-91
template <typename T>
void minkowski(T (*g)[4])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) g[i][j] = (i == j) ? ((i == 0) ? 1 : -1) : 0;
}

// Kerr metric at position x = (t, r, theta, phi).  Reference: kerr.h:94-124
template <typename T>
void kerr_metric(T (*g)[4], T* x, T a)
{
    const krhost::BLCoefficients<T> m(x[1], x[2], a);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) g[i][j] = 0;
    g[0][0] = m.e2nu - m.omega * m.omega * m.e2psi;
    g[0][3] = m.omega * m.e2psi;
    g[3][0] = g[0][3];
    g[3][3] = -m.e2psi;
    g[1][1] = -m.rhosq / m.delta;
    g[2][2] = -m.rhosq;
}

// orthonormal tetrad of an observer orbiting at angular velocity V.  Reference: kerr.h:127-170
template <typename T>
void tetrad(T* et, T* e1, T* e2, T* e3, T* x, T V, T a)
{
    using std::sqrt;
    const krhost::BLCoefficients<T> m(x[1], x[2], a);
    const T e2nu = m.e2nu, e2psi = m.e2psi, omega = m.omega;

    et[0] = (1 / sqrt(e2nu)) / sqrt(1 - (V - omega) * (V - omega) * e2psi / e2nu);
    et[1] = 0;
    et[2] = 0;
    et[3] = (1 / sqrt(e2nu)) * V / sqrt(1 - (V - omega) * (V - omega) * e2psi / e2nu);

    e1[0] = (V - omega) * sqrt(e2psi / e2nu) / sqrt(e2nu - (V - omega) * (V - omega) * e2psi);
    e1[1] = 0;
    e1[2] = 0;
    e1[3] = (1 / sqrt(e2nu * e2psi)) * (e2nu + V * omega * e2psi - omega * omega * e2psi) / sqrt(e2nu - (V - omega) * (V - omega) * e2psi);

    e2[0] = 0;
    e2[1] = 0;
    e2[2] = 1 / sqrt(m.rhosq);
    e2[3] = 0;

    e3[0] = 0;
    e3[1] = sqrt(m.delta / m.rhosq);
    e3[2] = 0;
    e3[3] = 0;
}

// Lorentz factor of 4-velocity v seen by the zero-angular-momentum observer at x; vel[1..3] receive the
// 3-velocity.  Reference: kerr.h:173-213
template <typename T>
T lorentz(T* vel, T* v, T* x, T a)
{
    T g[4][4], et[4], e1[4], e2[4], e3[4], gv[4];
    const krhost::BLCoefficients<T> m(x[1], x[2], a);
    kerr_metric(g, x, a);
    tetrad(et, e1, e2, e3, x, m.omega, a);

    gv[0] = g[0][0] * v[0] * et[0] + g[0][3] * v[0] * et[3] + g[3][0] * v[3] * et[0] + g[3][3] * v[3] * et[3];
    gv[1] = g[0][0] * v[0] * e1[0] + g[0][3] * v[0] * e1[3] + g[3][0] * v[3] * e1[0] + g[3][3] * v[3] * e1[3];
    gv[2] = g[2][2] * v[2] * e2[2];
    gv[3] = g[1][1] * v[1] * e3[1];
    for (int i = 1; i < 4; i++) vel[i] = gv[i] / gv[0];
    return gv[0];
}

// 4-velocity of a stable circular equatorial orbit; returns dphi/dt.  Reference: kerr.h:216-247
template <typename T>
T disc_velocity_vector(T* v, T r, T a, int sign)
{
    using std::sqrt;
    const T u = 1 / r;
    const T k = (1 - 2 * u + sign * a * sqrt(u * u * u)) / sqrt(1 - 3 * u + sign * 2 * a * sqrt(u * u * u));
    const T h = sign * (1 + a * a * u * u - sign * 2 * a * sqrt(u * u * u)) / (sqrt(u) * sqrt(1 - 3 * u + sign * 2 * a * sqrt(u * u * u)));

    v[0] = (r * r * (r * r + a * a) + 2 * a * a * r) * k - 2 * a * r * h;
    v[0] = v[0] / (r * r * (1 - (2 / r)) * (r * r + a * a) + 2 * a * a * r);
    v[1] = 0;
    v[2] = 0;
    v[3] = 2 * a * r * k + (r * r - 2 * r) * h;
    v[3] = v[3] / (r * r * (1 - (2 / r)) * (r * r + a * a) + 2 * a * a * r);
    return v[3] / v[0];
}

// proper area of an equatorial annulus.  Reference: kerr.h:250-265
template <typename T>
T disc_area(T r, T dr, T a)
{
    using std::sqrt;
    const T rhosq = r * r;
    const T delta = r * r - 2 * r + a * a;
    return sqrt(r * r + a * a + (2 * a * a * r) / rhosq) * sqrt(rhosq / delta) * dr;
}

// ... divided by the Lorentz factor of the orbiting material.  Reference: kerr.h:268-297
template <typename T>
T rel_disc_area(T r, T dr, T a)
{
    T g[4][4], et[4], e1[4], e2[4], e3[4], v[4], vel[3];
    T pos[] = {0, 0, 0, 0};
    pos[1] = r;
    kerr_metric<T>(g, pos, a);
    const T V = disc_velocity_vector<T>(v, r, a, +1);
    tetrad(et, e1, e2, e3, pos, V, a);
    const T gr_area = disc_area<T>(r, dr, a);
    const T gamma = lorentz<T>(vel, v, pos, a);
    return gr_area / gamma;
}

// photon momentum (tdot, rdot, thetadot, phidot) from the constants of motion (k, h, Q) and the two
// direction signs at (r, theta).  Reference: kerr.h:300-335.  This is the derivative evaluation that the HIP
// trace kernel performs on the device (raytrace_cpu_amd/csrc/kr_device.hpp); the host copy serves the O(N)
// member functions of the float instantiation and user-defined RayDestination velocity fields.
template <typename T>
inline void momentum_from_consts(T& pt, T& pr, T& ptheta, T& pphi, T k, T h, T Q, int rdot_sign, int thetadot_sign, T r, T theta,
                                 T phi, const T a)
{
    using std::abs;
    using std::cos;
    using std::sin;
    using std::sqrt;
    (void) phi;
    const T s = sin(theta);
    const T c = cos(theta);
    const T s2 = s * s;
    const T rhosq = r * r + (a * c) * (a * c);
    const T delta = r * r - 2 * r + a * a;
    const T rhosq_delta = rhosq * delta;

    pt = (rhosq * (r * r + a * a) + 2 * a * a * r * s2) * k - 2 * a * r * h;
    pt /= rhosq_delta;

    pphi = 2 * a * r * s2 * k + (rhosq - 2 * r) * h;
    pphi /= s2 * rhosq_delta;

    T thetadotsq = Q + (k * a * c + h * c / s) * (k * a * c - h * c / s);
    thetadotsq = thetadotsq / (rhosq * rhosq);
    ptheta = sqrt(abs(thetadotsq)) * thetadot_sign;

    T rdotsq = k * pt - h * pphi - rhosq * ptheta * ptheta;
    rdotsq = rdotsq * delta / rhosq;
    pr = sqrt(abs(rdotsq)) * rdot_sign;
}

#endif /* KERR_H_ */
