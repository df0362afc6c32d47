// kr_post_device.hpp -- the per-ray device functions of the O(N) passes either side of the trace (ray sources, redshift_start, redshift,
// range_phi, the emissivity reducer's accumulation), applied by the streaming kernels of kr_post.hip to 144-byte records in HBM; the fused and the
// separate passes share one definition each, so both routes produce the same per-ray bits.  Reference lines cited per function.
#pragma once

#include <hip/hip_runtime.h>

#include <cstring>

#include "kr_device.hpp"

namespace kr {

// Kerr metric in the (e2nu, e2psi, omega) form used throughout the reference
// (raytracer.cpp:370-388, :491-509, :564-582, :632-639).  Everything up to calculate_momentum below is a template over
// the ray record R (kr_ray_f64 / kr_ray_f32) and computes in the record's scalar type, like the reference's
// Raytracer<double> / Raytracer<float> instantiations of the same source lines.
template <typename T>
struct Metric {
    T rhosq, delta, sigmasq, e2nu, e2psi, omega;
    T g00, g03, g11, g22, g33;
};

template <typename R> struct ScalarOf;
template <> struct ScalarOf<kr_ray_f64> { using type = double; };
template <> struct ScalarOf<kr_ray_f32> { using type = float; };

template <typename T>
KR_DEV Metric<T> kerr_metric(T r, T theta, T a)
{
    Metric<T> m;
    const T st = kr_sin(theta), ct = kr_cos(theta);
    m.rhosq = r * r + (a * ct) * (a * ct);
    m.delta = r * r - 2 * r + a * a;
    m.sigmasq = (r * r + a * a) * (r * r + a * a) - a * a * m.delta * st * st;
    m.e2nu = m.rhosq * m.delta / m.sigmasq;
    m.e2psi = m.sigmasq * st * st / m.rhosq;
    m.omega = 2 * a * r / m.sigmasq;
    m.g00 = m.e2nu - m.omega * m.omega * m.e2psi;
    m.g03 = m.omega * m.e2psi;
    m.g11 = -m.rhosq / m.delta;
    m.g22 = -m.rhosq;
    m.g33 = -m.e2psi;
    return m;
}

// sum_ij g[i][j] * et[i] * p[j] over all 16 entries in row-major order, zeros included, exactly like the
// reference loops (raytracer.cpp:412-415, :547-550): a 0 * inf or 0 * NaN term must poison the sum the same way.
template <typename T>
KR_DEV T energy_dot(const Metric<T>& m, const T* et, const T* p)
{
    const T g[16] = {m.g00, 0, 0, m.g03, 0, m.g11, 0, 0, 0, 0, m.g22, 0, m.g03, 0, 0, m.g33};
    T e = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) e += g[i * 4 + j] * et[i] * p[j];
    return e;
}

template <typename T>
KR_DEV T keplerian_V(T a, T r, T theta, bool projradius)
{
    if (projradius) return 1 / (a + r * kr_sin(theta) * kr_sqrt(r * kr_sin(theta)));
    return 1 / (a + r * kr_sqrt(r));
}

// ---- redshift_start (raytracer.cpp:342-417) ------------------------------------------------------------
// V is a by-value parameter that the reference's loop overwrites when it equals -1, so the orbital velocity
// computed at the FIRST record (index 0, valid or not) is used for every ray.
template <typename R, typename T = typename ScalarOf<R>::type>
KR_DEV T emit_value(const R& ray, T spin, T a, T V, int reverse)
{
    const T r = ray.r, theta = ray.theta;
    const Metric<T> m = kerr_metric<T>(r, theta, a);
    const T et[4] = {(1 / kr_sqrt(m.e2nu)) / kr_sqrt(1 - (V - m.omega) * (V - m.omega) * m.e2psi / m.e2nu), 0, 0,
                     (1 / kr_sqrt(m.e2nu)) * V / kr_sqrt(1 - (V - m.omega) * (V - m.omega) * m.e2psi / m.e2nu)};
    T p[4];
    momentum<T>(p[0], p[1], p[2], p[3], ray.k, ray.h, ray.Q, ray.rdot_sign, ray.thetadot_sign, r, theta, spin);
    if (reverse) { p[1] *= -1; p[2] *= -1; p[3] *= -1; }
    return energy_dot<T>(m, et, p);
}

// ---- redshift(V, ...) (raytracer.cpp:420-447, :480-553) ----------------------------------------------
template <typename R, typename T = typename ScalarOf<R>::type>
KR_DEV T redshift_value(const R& ray, T spin, T V_in, int reverse, int projradius, int motion)
{
    const T a = reverse ? -1 * spin : spin;
    const T r = ray.r, theta = ray.theta;
    const Metric<T> m = kerr_metric<T>(r, theta, a);
    T V = V_in;
    T et[4] = {0, 0, 0, 0};
    if (motion == 0) {
        if (V == -1) V = keplerian_V<T>(a, r, theta, projradius != 0);
        et[0] = (1 / kr_sqrt(m.e2nu)) / kr_sqrt(1 - (V - m.omega) * (V - m.omega) * m.e2psi / m.e2nu);
        et[3] = (1 / kr_sqrt(m.e2nu)) * V / kr_sqrt(1 - (V - m.omega) * (V - m.omega) * m.e2psi / m.e2nu);
    } else if (motion == 1) {
        if (V < 0) V = kr_abs(V) * (r * r - 2 * r + spin + spin) / (r * r + spin * spin);   // sic, :531
        et[0] = (T) (1. / kr_sqrt(m.g00 + m.g11 * V * V));                                  // (a double division in the float build too, :533)
        et[1] = V * et[0];
    }
    T p[4];
    momentum<T>(p[0], p[1], p[2], p[3], ray.k, ray.h, ray.Q, ray.rdot_sign, ray.thetadot_sign, r, theta, spin);
    if (reverse) { p[1] *= -1; p[2] *= -1; p[3] *= -1; }
    const T recv = energy_dot<T>(m, et, p);
    return reverse ? recv / ray.emit : ray.emit / recv;
}

// ---- range_phi (raytracer.cpp:603-622): repeated +-2pi like the reference, so the result is bit-identical.  2 * M_PI is a double:
//      in the float build each `phi -= 2 * M_PI` is a double subtraction rounded back to float, here as there. ----
template <typename T>
KR_DEV T range_phi_value(T phi, int steps, T lo, T hi)
{
    if (kr_abs(phi) > 1000 || phi != phi || !(steps > 0)) return phi;
    while (phi >= hi) phi -= 2 * kPi;
    while (phi < lo) phi += 2 * kPi;
    return phi;
}

// ---- PointSource ctor: Raytracer ctor (steps=-1, status=0, raytracer.cpp:45-49) + init_pointsource
//      (pointsource.cpp:30-64) + calculate_constants (raytracer.cpp:625-676).  Fields the reference leaves
//      indeterminate are zeroed. ------------------------------------------------------------------------
// The constructor's transcendental values, tabulated ON THE HOST with the C library the reference itself calls: alpha = acos(cos alpha) takes
// only n_cosalpha distinct values over the whole grid, beta n_beta, and the source position is one point -- so sin(alpha_i), cos(alpha_i), sin(beta_j),
// cos(beta_j) (two small device arrays: 100 KB at 1e7 rays) and sin / cos / tan of the source's polar angle (three scalars) carry glibc's bits
// and everything that is left for the device is + - x / sqrt, which is IEEE on both sides: k, h, Q of EVERY device-built ray are the reference
// constructor's, bit for bit (with the device library's acos 5 % of the rays differed in the last bit of h and Q, and on chaotic rays that is
// another bin).  kr_post.hip::source_tables builds and caches the arrays per (device, grid).
struct SourceTables {
    const double2* alpha_sc;     // [n_cosalpha]  (sin, cos) of acos(cosalpha0 + i dcosalpha)     pointsource.cpp:38,46; raytracer.cpp:653
    const double2* beta_sc;      // [n_beta]      (sin, cos) of beta0 + j dbeta                   pointsource.cpp:39;    raytracer.cpp:653
    double sin_th, cos_th, tan_th;   // of pos[2]                                                 raytracer.cpp:631-672
};

KR_DEV kr_ray_f64 pointsource_ray(const kr_pointsource& s, const SourceTables& tb, long long n_grid, int n_beta, long long ix)
{
    kr_ray_f64 ray;
        memset(&ray, 0, sizeof(ray));
        ray.steps = -1;
        if (ix < n_grid) {
            const int i = (int) (ix / n_beta), j = (int) (ix % n_beta);
            const double cosalpha = s.cosalpha0 + i * s.dcosalpha;
            const double beta = s.beta0 + j * s.dbeta;
            if (!(cosalpha >= s.cosalphamax || beta >= s.betamax)) {
                const double2 a_sc = tb.alpha_sc[i], b_sc = tb.beta_sc[j];
                const double sin_alpha = a_sc.x, cos_alpha = a_sc.y, sin_beta = b_sc.x, cos_beta = b_sc.y;
                ray.alpha = cosalpha;      // sic: cos(alpha), pointsource.cpp:48
                ray.beta = beta;
                ray.t = s.pos[0]; ray.r = s.pos[1]; ray.theta = s.pos[2]; ray.phi = s.pos[3];
                ray.steps = 0;

                const double spin = s.spin, V = s.V, E = s.E;
                const double r = ray.r;
                const double st = tb.sin_th, ct = tb.cos_th;
                const double rhosq = r * r + (spin * ct) * (spin * ct);
                const double delta = r * r - 2 * r + spin * spin;
                const double sigmasq = (r * r + spin * spin) * (r * r + spin * spin) - spin * spin * delta * st * st;
                const double e2nu = rhosq * delta / sigmasq;
                const double e2psi = sigmasq * st * st / rhosq;
                const double omega = 2 * spin * r / sigmasq;

                const double et0 = (1 / kr_sqrt(e2nu)) / kr_sqrt(1 - (V - omega) * (V - omega) * e2psi / e2nu);
                const double et3 = (1 / kr_sqrt(e2nu)) * V / kr_sqrt(1 - (V - omega) * (V - omega) * e2psi / e2nu);
                const double e10 = (V - omega) * kr_sqrt(e2psi / e2nu) / kr_sqrt(e2nu - (V - omega) * (V - omega) * e2psi);
                const double e13 = (1 / kr_sqrt(e2nu * e2psi)) * (e2nu + V * omega * e2psi - omega * omega * e2psi) /
                                   kr_sqrt(e2nu - (V - omega) * (V - omega) * e2psi);
                const double e22 = -1 / kr_sqrt(rhosq);
                const double e31 = kr_sqrt(delta / rhosq);

                const double rp0 = E, rp1 = E * sin_alpha * cos_beta, rp2 = E * sin_alpha * sin_beta, rp3 = E * cos_alpha;
                const double tdot = rp0 * et0 + rp1 * e10;
                const double phidot = rp0 * et3 + rp1 * e13;
                const double rdot = rp3 * e31;
                const double thetadot = rp2 * e22;

                ray.k = (1 - 2 * r / rhosq) * tdot + (2 * spin * r * st * st / rhosq) * phidot;
                double h = phidot * ((r * r + spin * spin) * (r * r + spin * spin * ct * ct - 2 * r) * st * st + 2 * spin * spin * r * st * st * st * st);
                h = h - 2 * spin * r * ray.k * st * st;
                h = h / (r * r + spin * spin * ct * ct - 2 * r);
                ray.h = h;
                const double tt = tb.tan_th;
                ray.Q = rhosq * rhosq * thetadot * thetadot - (spin * ray.k * ct + h / tt) * (spin * ray.k * ct - h / tt);
                ray.rdot_sign = (rdot >= 0) ? 1 : -1;
                ray.thetadot_sign = (thetadot > 0) ? 1 : -1;
            }
        }
    return ray;
}

// d_hist layout: [count(nr) | flux(nr) | emis(nr) | sum_redshift(nr) | sum_time(nr) | disc_count(1)], doubles.
// LDS holds one private copy per workgroup when it fits (5*nr+1 doubles); flushed with global f64 atomics.
constexpr int kMaxLdsBins = 1024;

// one ray's contribution to the radial histogram (emissivity.cpp:96-126); acc: LDS copy or the global histogram
KR_DEV void emissivity_accumulate(double* acc, const kr_emis_bins& b, double log_dr, int steps, double r, double theta, double g, double t)
{
    if (!(steps > 0)) return;
    const int nr = b.nr;
    const double z = r * kr_cos(theta);         // cartesian(), kerr.h:55
    if (z < 1E-2 && g > 0 && r >= b.r_isco) {
        const int ir = b.logbin ? (int) (kr_log(r / b.r_min) / log_dr) : (int) ((r - b.r_min) / b.dr);
        if (ir >= 0 && ir < nr) {
            atomicAdd(&acc[ir], 1.0);
            atomicAdd(&acc[nr + ir], 1 / (b.num_primary_rays * kr_pow(g, 1.0)));
            atomicAdd(&acc[2 * nr + ir], 1 / kr_pow(g, b.gamma));
            atomicAdd(&acc[3 * nr + ir], g);
            atomicAdd(&acc[4 * nr + ir], t);
        }
        atomicAdd(&acc[5 * nr], 1.0);
    }
}

}  // namespace kr
