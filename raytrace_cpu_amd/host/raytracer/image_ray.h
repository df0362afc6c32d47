// raytracer/image_ray.h -- the camera ray through one point (x, y) of a distant image plane, shared by ImagePlane<T> and ImagePlaneBundles<T>
// (the reference writes the same construction out in both: src/raytracer/imageplane.cpp:45-113 and imageplane_bundles.h:123-196).
//
//   position    r = sqrt(D^2 + x^2 + y^2), cos(theta) = (D cos i + y sin i) / r, phi = phi0 + atan2(x, D sin i - y cos i)
//   direction   along the line of sight: pr = D / r, ptheta = sin(acos(D / r)) / r, pphi = x sin i / (x^2 + (D sin i - y cos i)^2);
//               pt from the null condition g_ab p^a p^b = 0 (positive root)
//   constants   k = 1; h and Q from the impact parameters of the point: b = sqrt(x^2 + y^2), sin(beta) = y / b (mirrored for x < 0),
//               h = -b sin i cos(beta), l_theta = b sin(beta), Q = l_theta^2 - a^2 cos^2(theta) + h^2 cot^2(theta)
//
// Every expression keeps the reference's operand order and intermediate types, because the bits of k, h, Q decide which pixel a ray ends
// in.  Two places differ between the two classes in the float build and are therefore parameters: the type the spin enters the metric
// with (ImagePlane copies it into a double, imageplane.cpp:38) and whether h is formed with a double factor (its "-1." there, :103).
#ifndef KR_HOST_IMAGE_RAY_H_
#define KR_HOST_IMAGE_RAY_H_

#include <cmath>

#include "../include/kerr.h"

namespace krhost {

template <typename T>
struct CameraRay {
    T r, theta, phi;
    T pt, pr, ptheta, pphi;
    T h, Q;
    int thetadot_sign;
};

// covariant metric components the null condition needs, from the (e2nu, e2psi, omega) functions
template <typename T>
struct CovariantBL {
    T g00, g03, g11, g22, g33;
    explicit CovariantBL(const BLCoefficients<T>& m)
        : g00(m.e2nu - m.omega * m.omega * m.e2psi), g03(m.omega * m.e2psi), g11(-m.rhosq / m.delta), g22(-m.rhosq), g33(-m.e2psi) {}

    // pt with g_ab p^a p^b = 0: positive root of g00 pt^2 + 2 g03 pphi pt + (g11 pr^2 + g22 ptheta^2 + g33 pphi^2) = 0, the other one if that is negative
    T null_pt(T pr, T ptheta, T pphi) const
    {
        using std::sqrt;
        const T A = g00;
        const T B = 2 * g03 * pphi;
        const T C = g11 * pr * pr + g22 * ptheta * ptheta + g33 * pphi * pphi;
        T pt = (-B + sqrt(B * B - 4 * A * C)) / (2 * A);
        if (pt < 0) pt = (-B - sqrt(B * B - 4 * A * C)) / (2 * A);
        return pt;
    }
};

// SPIN: type of the spin inside the metric functions and Q; H_IN_DOUBLE: h = -1. * b sin i cos(beta) (double factor) instead of -b sin i cos(beta);
// GUARD_CENTRE: beta = 0 at the point x = y = 0 instead of asin(0 / 0)
template <typename T, typename SPIN, bool H_IN_DOUBLE, bool GUARD_CENTRE>
CameraRay<T> camera_ray(T D, T incl, T phi0, T x, T y, SPIN a)
{
    using std::acos; using std::asin; using std::atan2; using std::cos; using std::sin; using std::sqrt; using std::tan;
    CameraRay<T> c;
    c.r = sqrt(D * D + x * x + y * y);
    c.theta = acos((D * cos(incl) + y * sin(incl)) / c.r);
    c.phi = phi0 + atan2(x, D * sin(incl) - y * cos(incl));
    c.pr = D / c.r;
    c.ptheta = sin(acos(D / c.r)) / c.r;
    c.pphi = x * sin(incl) / (x * x + (D * sin(incl) - y * cos(incl)) * (D * sin(incl) - y * cos(incl)));

    const T r = c.r, theta = c.theta;
    BLCoefficients<T> m;
    m.rhosq = r * r + (a * cos(theta)) * (a * cos(theta));
    m.delta = r * r - 2 * r + a * a;
    m.sigmasq = (r * r + a * a) * (r * r + a * a) - a * a * m.delta * sin(theta) * sin(theta);
    m.e2nu = m.rhosq * m.delta / m.sigmasq;
    m.e2psi = m.sigmasq * sin(theta) * sin(theta) / m.rhosq;
    m.omega = 2 * a * r / m.sigmasq;
    c.pt = CovariantBL<T>(m).null_pt(c.pr, c.ptheta, c.pphi);

    const T b = sqrt(x * x + y * y);
    T beta = (GUARD_CENTRE && !(b > 0)) ? T(0) : asin(y / b);
    if (x < 0) beta = M_PI - beta;
    const T h = H_IN_DOUBLE ? T(-1. * b * sin(incl) * cos(beta)) : T(-b * sin(incl) * cos(beta));
    const T ltheta = b * sin(beta);
    c.h = h;
    c.Q = (ltheta * ltheta) - (a * cos(theta)) * (a * cos(theta)) + ((h / tan(theta))) * ((h / tan(theta)));
    c.thetadot_sign = (ltheta >= 0) ? 1 : -1;
    return c;
}

}  // namespace krhost

#endif /* KR_HOST_IMAGE_RAY_H_ */
