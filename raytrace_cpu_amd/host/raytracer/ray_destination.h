// raytracer/ray_destination.h -- stop surfaces for Raytracer<T>::run_raytrace(RayDestination<T>*, ...).
//
// Mirrors the interface of the reference's src/raytracer/ray_destination.h (abstract base :43-79 and the
// three concrete classes :86-102, :116-152, :173-204): same class names, constructors, virtuals and
// semantics, so application code that instantiates them compiles unchanged.
//
// How the HIP path uses them: the per-step virtual calls of the reference (raytracer.cpp:1203, :1753,
// :1844) cannot run on the device, so every built-in destination also describes itself as a POD record
// (`describe()`), which Raytracer<T> hands to the trace kernel as kr_params.stop_kind / stop_params.
// A user-defined subclass has no such description (describe() returns false): run_raytrace() then
// throws instead of silently falling back to a CPU loop.
#ifndef RAY_DESTINATION_H_
#define RAY_DESTINATION_H_

#include <cmath>
#include <limits>

#include "../../../include/kr_trace.h"
#include "../include/kerr.h"

template <typename T>
class RayDestination {
public:
    virtual ~RayDestination() = default;

    // true -> the ray stops at (r, theta, phi)
    virtual bool reached(T r, T theta, T phi) const = 0;
    // crossing-aware form; prev_theta is theta before the step just taken
    virtual bool reached(T r, T theta, T phi, T prev_theta) const { return reached(r, theta, phi); }
    // largest step (linear extrapolation with the stage-1 momenta) that does not overshoot the surface
    virtual T step_limit(T r, T theta, T phi, T pr, T ptheta, T pphi) const { return std::numeric_limits<T>::max(); }
    // angular velocity of the material at the surface; -1 = equatorial Keplerian
    virtual T velocity(T r, T theta, T phi) const { return -1; }

    // contravariant Boyer-Lindquist 4-velocity {ut, ur, utheta, uphi} of the material (circular motion at velocity())
    virtual void four_velocity(T r, T theta, T phi, T spin, T et[4]) const
    {
        using std::sqrt;
        T V = velocity(r, theta, phi);
        if (V == -1) V = 1 / (spin + r * sqrt(r));                 // equatorial Keplerian
        // u = (ut, 0, 0, V ut) normalised in the frame of the zero-angular-momentum observer: relative angular velocity V - omega
        const krhost::BLCoefficients<T> m(r, theta, spin);
        const T lorentz = 1 / sqrt(1 - (V - m.omega) * (V - m.omega) * m.e2psi / m.e2nu);
        et[0] = lorentz / sqrt(m.e2nu);
        et[1] = 0;
        et[2] = 0;
        et[3] = lorentz * V / sqrt(m.e2nu);
    }

    // POD description for the device path: fills stop_kind (KR_STOP_*) and stop_params[4]; false if this
    // object cannot be evaluated on the device.  `default_velocity` tells whether four_velocity() is the
    // base-class circular-orbit field with velocity() == -1 (then redshift(dest) runs on the device too).
    virtual bool describe(int& stop_kind, double stop_params[4], bool& default_velocity) const { return false; }
};

// theta_lim > 0: stop when theta >= theta_lim; < 0: when theta <= |theta_lim|; 0: never
template <typename T>
class FlatDiscDestination : public RayDestination<T> {
    T theta_lim;

public:
    explicit FlatDiscDestination(T theta_lim = M_PI_2) : theta_lim(theta_lim) {}

    bool reached(T r, T theta, T phi) const override
    {
        if (theta_lim > 0) return theta >= theta_lim;
        if (theta_lim < 0) return theta <= -theta_lim;
        return false;
    }
    T step_limit(T r, T theta, T phi, T pr, T ptheta, T pphi) const override
    {
        if (theta_lim > 0 && ptheta > 0 && theta < theta_lim) return (theta_lim - theta) / ptheta;
        if (theta_lim < 0 && ptheta < 0 && theta > -theta_lim) return (-theta_lim - theta) / ptheta;
        return std::numeric_limits<T>::max();
    }
    bool describe(int& stop_kind, double sp[4], bool& default_velocity) const override
    {
        stop_kind = KR_STOP_FLATDISC;
        sp[0] = theta_lim; sp[1] = sp[2] = sp[3] = 0;
        default_velocity = true;
        return true;
    }
};

// a flat disc that only exists for r_isco <= r <= r_out (r_out <= 0: unbounded); stops on an actual crossing of
// theta_lim between two steps, from either side
template <typename T>
class DiscWithISCODestination : public RayDestination<T> {
    T theta_lim;
    T r_isco;
    T r_out;

    bool outside(T r) const { return r < r_isco || (r_out > 0 && r > r_out); }

public:
    explicit DiscWithISCODestination(T r_isco, T r_out = -1, T theta_lim = M_PI_2) : theta_lim(theta_lim), r_isco(r_isco), r_out(r_out) {}

    bool reached(T r, T theta, T phi) const override
    {
        if (outside(r)) return false;
        if (theta_lim > 0) return theta >= theta_lim;
        if (theta_lim < 0) return theta <= -theta_lim;
        return false;
    }
    bool reached(T r, T theta, T phi, T prev_theta) const override
    {
        if (outside(r)) return false;
        if (theta_lim > 0) return (prev_theta < theta_lim && theta >= theta_lim) || (prev_theta > theta_lim && theta <= theta_lim);
        if (theta_lim < 0) {
            const T tl = -theta_lim;
            return (prev_theta > tl && theta <= tl) || (prev_theta < tl && theta >= tl);
        }
        return false;
    }
    T step_limit(T r, T theta, T phi, T pr, T ptheta, T pphi) const override
    {
        if (outside(r)) return std::numeric_limits<T>::max();
        if (theta_lim > 0 && ptheta > 0 && theta < theta_lim) return (theta_lim - theta) / ptheta;
        if (theta_lim < 0 && ptheta < 0 && theta > -theta_lim) return (-theta_lim - theta) / ptheta;
        return std::numeric_limits<T>::max();
    }
    bool describe(int& stop_kind, double sp[4], bool& default_velocity) const override
    {
        stop_kind = KR_STOP_DISC_ISCO;
        sp[0] = r_isco; sp[1] = r_out; sp[2] = theta_lim; sp[3] = 0;
        default_velocity = true;
        return true;
    }
};

// plane perpendicular to the line of sight (incl, phi0), a distance z_s behind the black hole:
// reached when r (sin(theta) sin(incl) cos(phi - phi0) + cos(theta) cos(incl)) <= -z_s
template <typename T>
class FlatPlaneDestination : public RayDestination<T> {
public:
    T incl;
    T phi0;
    T z_s;

    FlatPlaneDestination(T incl, T phi0, T z_s) : incl(incl), phi0(phi0), z_s(z_s) {}

    T projection(T r, T theta, T phi) const
    {
        using std::cos;
        using std::sin;
        return r * (sin(theta) * sin(incl) * cos(phi - phi0) + cos(theta) * cos(incl));
    }
    bool reached(T r, T theta, T phi) const override { return projection(r, theta, phi) <= -z_s; }

    // (East, North) coordinates on the plane, same orientation as the image plane
    void source_coords(T r, T theta, T phi, T& x_s, T& y_s) const
    {
        using std::cos;
        using std::sin;
        const T X = r * sin(theta) * cos(phi);
        const T Y = r * sin(theta) * sin(phi);
        const T Z = r * cos(theta);
        x_s = -X * sin(phi0) + Y * cos(phi0);
        y_s = -X * cos(incl) * cos(phi0) - Y * cos(incl) * sin(phi0) + Z * sin(incl);
    }
    bool describe(int& stop_kind, double sp[4], bool& default_velocity) const override
    {
        stop_kind = KR_STOP_FLATPLANE;
        sp[0] = incl; sp[1] = phi0; sp[2] = z_s; sp[3] = 0;
        default_velocity = true;
        return true;
    }
};

#endif /* RAY_DESTINATION_H_ */
