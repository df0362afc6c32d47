// raytracer/raytracer.h -- host-side mirror of the reference's Raytracer<T> class API over the HIP path.
//
// Drop-in boundary: same class name, template parameter, public members (`rays`), member
// function names, overloads, default arguments, macros, status flags and Ray<T> layout as the reference's
// src/raytracer/raytracer.h (:19-46 macros, :58-63 flags, :65-78 Ray, :83 Integrator, :85-198 class), so the
// reference's applications (src/emissivity, src/imageplane, src/tests) compile and link against this tree
// unchanged.  What differs is behind the interface: run_raytrace() hands rays[] to libkrtrace.so
// (include/kr_trace.h) and the integration runs on an MI355X; there is no CPU integration loop here.
//
// Deliberate differences (all documented in INTEGRATION.md):
//   * outfile != nullptr (per-step trajectory dumps, reference raytracer.cpp:86-100) -> std::runtime_error;
//   * a RayDestination subclass other than the three built-ins -> std::runtime_error from run_raytrace();
//   * rays whose RK45 error norm is NaN end with KR_STATUS_NAN instead of hanging the process;
//   * `new Ray<T>[n]` is value-initialised (the reference leaves most fields indeterminate);
//   * show_progress is accepted and ignored (a GPU launch has no per-ray progress).
#ifndef RAYTRACER_H_
#define RAYTRACER_H_

// default step-size controls (reference raytracer.h:19-46)
#define PRECISION 100
#define TOL 100
#define THETA_PRECISION 50
#define MAXDT 1
#define MAXDT_RLIM 100
#define MAXDPHI 0.1
#define RLIM 1000
#define STEPLIM 10000000
#define RK45_STEPLIM 100000
#define THREAD_STEPLIM 10000000
#define MIN_STEP 1E-3
#define COUNT_MIN 100

#include <cmath>
#include <iomanip>
#include <iostream>
using namespace std;   // the reference header does this and its applications rely on it

#include "../include/kerr.h"

class TextOutput;      // only ever passed as a (null) pointer here; applications include text_output.h themselves

constexpr int RAY_STATUS_DEST = (1 << 0);        // reached the destination / polar-angle limit
constexpr int RAY_STATUS_HORIZON = (1 << 1);     // fell through the event horizon
constexpr int RAY_STATUS_RLIM = (1 << 2);        // reached the outer radial limit
constexpr int RAY_STATUS_STEPLIM = (1 << 3);     // exceeded the step limit
constexpr int RAY_STATUS_ERGO = (1 << 4);        // pt <= 0 at some step
constexpr int RAY_STATUS_NEG_ENERGY = (1 << 5);  // negative Killing energy at some step
constexpr int RAY_STATUS_NAN = (1 << 6);         // extension: NaN error norm in RK45 (the reference never returns)

template <typename T>
struct Ray {
    T t, r, theta, phi;
    T pt, pr, ptheta, pphi;
    T k, h, Q;
    T emit, redshift;
    int steps;
    int status;
    int rdot_sign, thetadot_sign;
    int rdot_flips;
    int equatorial_crossings;
    T alpha, beta;
};

template <typename T>
class RayDestination;

enum class Integrator { Euler, RK4, RK45 };

template <typename T>
class Raytracer {
private:
    T precision;
    T theta_precision;
    T max_tstep;
    T max_phistep;
    T maxtstep_rlim;
    T rk45_tol;

    struct kr_params_box;   // kr_params without exposing the C header to applications
    void fill_params(void* kr_params_out, Integrator method, T r_max, int steplim) const;
    void trace(const void* kr_params_in, Ray<T>* first, long n);

protected:
    int nRays;
    T spin;
    T horizon;

    void calculate_constants(int ray, T alpha, T beta, T V, T E);
    void calculate_constants_from_p(int ray, T pt, T pr, T ptheta, T pphi);

public:
    Ray<T>* rays;

    Raytracer(int num_rays, T spin, T precision = PRECISION, T init_max_phistep = MAXDPHI, T init_max_tstep = MAXDT);
    ~Raytracer();

    void run_raytrace(Integrator method = Integrator::Euler, T theta_max = M_PI / 2, T r_max = 1000, int show_progress = 1,
                      TextOutput* outfile = 0, int write_step = 1, T write_rmax = -1, T write_rmin = -1, bool write_cartesian = true,
                      int steplim = -1);
    void run_raytrace(RayDestination<T>* dest, Integrator method = Integrator::Euler, T r_max = 1000, int show_progress = 1,
                      TextOutput* outfile = 0, int write_step = 1, T write_rmax = -1, T write_rmin = -1, bool write_cartesian = true,
                      int steplim = -1);

    // single-ray forms of the reference API; each is a one-ray launch of the same kernels
    int propagate(int ray, const T rlim, const T thetalim, const int steplim, TextOutput* outfile = 0, int write_step = 1,
                  T write_rmax = -1, T write_rmin = -1, bool write_cartesian = true);
    int propagate_rk4(int ray, const T rlim, const T thetalim, const int steplim, TextOutput* outfile = 0, int write_step = 1,
                      T write_rmax = -1, T write_rmin = -1, bool write_cartesian = true);
    int propagate_rk4(int ray, const T rlim, RayDestination<T>* dest, const int steplim, TextOutput* outfile = 0, int write_step = 1,
                      T write_rmax = -1, T write_rmin = -1, bool write_cartesian = true);
    int propagate_rk45(int ray, const T rlim, const T thetalim, const int steplim, TextOutput* outfile = 0, int write_step = 1,
                       T write_rmax = -1, T write_rmin = -1, bool write_cartesian = true);
    int propagate_rk45(int ray, const T rlim, RayDestination<T>* dest, const int steplim, TextOutput* outfile = 0, int write_step = 1,
                       T write_rmax = -1, T write_rmin = -1, bool write_cartesian = true);

    void redshift_start(T V, bool reverse = false, bool projradius = false);
    void redshift(T V, bool reverse = false, bool projradius = false, int motion = 0);
    void redshift(RayDestination<T>* dest, bool reverse = false, bool projradius = false, int motion = 0);
    T ray_redshift(T V, bool reverse, bool projradius, T r, T theta, T phi, T k, T h, T Q, int rdot_sign, int thetadot_sign, T emit,
                   int motion = 0);
    T ray_redshift(const T et[4], bool reverse, T r, T theta, T phi, T k, T h, T Q, int rdot_sign, int thetadot_sign, T emit);

    void range_phi(T min = -1 * M_PI, T max = M_PI);
    void calculate_momentum();

    int get_count() { return nRays; }

    void set_boundary(T r = -1)
    {
        if (r > 0)
            horizon = r;
        else
            horizon = kerr_horizon<T>(spin);
    }
    T calculate_horizon() { return kerr_horizon<T>(spin); }

    // The reference's setters assign the parameter to itself (raytracer.h:169-178), i.e. they do nothing; an
    // application that calls them must keep getting the constructor's precision, so they do nothing here too.
    void set_precision(T precision) { (void) precision; }
    void set_precision(T precision, T theta_precision)
    {
        (void) precision;
        (void) theta_precision;
    }

    void set_rk45_tol(T tol) { rk45_tol = tol; }
    T get_rk45_tol() const { return rk45_tol; }
    void set_max_tstep(T max, T rlim = MAXDT_RLIM)
    {
        max_tstep = max;
        maxtstep_rlim = rlim;
    }
    void set_max_phistep(T max) { max_phistep = max; }
};

#endif /* RAYTRACER_H_ */
