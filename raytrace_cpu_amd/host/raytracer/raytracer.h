// raytracer/raytracer.h -- host-side mirror of the reference's Raytracer<T> class API over the HIP path.
//
// Drop-in boundary.  Everything an application can name is kept: the macros of the reference's
// src/raytracer/raytracer.h:19-46, the status flags :58-63, the layout of Ray<T> :65-78, enum Integrator :83 and the
// members of class Raytracer<T> :85-198 (names, overloads, default arguments, public `rays`).  The reference's
// applications (src/emissivity, src/imageplane, src/tests) therefore compile and link against this tree unchanged.
// What differs is behind the interface: run_raytrace() hands rays[] to libkrtrace.so (include/kr_trace.h) and the
// integration runs on an MI355X; there is no CPU integration loop in this tree.
//
// Deliberate differences:
//   * outfile != nullptr (per-step trajectory dumps, reference raytracer.cpp:86-100)  -> std::runtime_error
//   * a RayDestination subclass other than the three built-ins in run_raytrace()       -> std::runtime_error
//   * an RK45 ray whose error norm is NaN ends with RAY_STATUS_NAN instead of hanging the process
//   * `new Ray<T>[n]` is value-initialised (the reference leaves most fields indeterminate)
//   * show_progress reports the rays the kernels have taken off their work queue, polled every 20 ms (the reference: one report per
//     `show_progress` rays from the OpenMP loop): the same line, fewer updates
#ifndef RAYTRACER_H_
#define RAYTRACER_H_

// ---- compile-time defaults --------------------------------------------------------------------------------------
#define PRECISION 100           // step = distance-to-horizon / |rdot| / PRECISION
#define TOL 100                 // (alias the applications pass as the `tol` constructor argument)
#define THETA_PRECISION 50      // ... and theta / |thetadot| / THETA_PRECISION when that is the tighter one
#define MAXDT 1                 // cap on the coordinate-time advance per step ...
#define MAXDT_RLIM 100          // ... applied inside this radius only
#define MAXDPHI 0.1             // cap on the azimuth advance per step
#define RLIM 1000               // default outer radius
#define STEPLIM 10000000        // steps per ray before it is abandoned (Euler, RK4)
#define RK45_STEPLIM 100000     // the same for the adaptive integrator
#define THREAD_STEPLIM 10000000 // unused (kept for source compatibility)
#define MIN_STEP 1E-3           // floor on the affine step
#define COUNT_MIN 100           // unused (kept for source compatibility)

#include <cmath>
#include <cstdlib>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <iomanip>
#include <iostream>
using namespace std;   // the reference header does this and its applications rely on it

#include "../include/kerr.h"

// Team size of the mirror's own host loops (first touch of rays[], the source constructors).  At most 32 threads, whatever the machine has:
// these loops are memory-bound and done in 5-20 ms on 16-32 threads, while on a 256-thread box libgomp's idle workers keep spinning after each
// region and starve the HIP runtime's own threads -- measured on the reference's emissivity main() at 1e7 rays: 1487 ms with 256 threads,
// 1134 ms with 32 (scripts/app_threads.sh).  KRTRACE_HOST_THREADS overrides.
inline int kr_host_threads()
{
#ifdef _OPENMP
    static const int n = [] {
        const char* e = std::getenv("KRTRACE_HOST_THREADS");
        const int want = e ? std::atoi(e) : 32;
        const int have = omp_get_max_threads();
        return want > 0 ? (want < have ? want : have) : have;
    }();
    return n;
#else
    return 1;
#endif
}

class TextOutput;      // only ever passed as a (null) pointer here; applications include text_output.h themselves

// ---- per-ray status bits (OR-ed, never cleared) --------------------------------------------------------------------
constexpr int RAY_STATUS_DEST = (1 << 0);        // reached the destination / polar-angle limit
constexpr int RAY_STATUS_HORIZON = (1 << 1);     // fell through the event horizon
constexpr int RAY_STATUS_RLIM = (1 << 2);        // reached the outer radial limit
constexpr int RAY_STATUS_STEPLIM = (1 << 3);     // exceeded the step limit (steps is then negated)
constexpr int RAY_STATUS_ERGO = (1 << 4);        // dt/dlambda <= 0 at some step
constexpr int RAY_STATUS_NEG_ENERGY = (1 << 5);  // negative Killing energy at some step
constexpr int RAY_STATUS_NAN = (1 << 6);         // extension: NaN error norm in RK45 (the reference never returns)

// One photon.  Layout-identical to kr_ray_f64 / kr_ray_f32 of the C ABI (144 / 84 bytes): rays[] goes to the GPU as is.
template <typename T>
struct Ray {
    T t, r, theta, phi;             // Boyer-Lindquist position
    T pt, pr, ptheta, pphi;         // momenta of the last derivative evaluation
    T k, h, Q;                      // constants of motion: energy, axial angular momentum, Carter constant
    T emit, redshift;               // emitted energy (redshift_start) and energy ratio (redshift)
    int steps;                      // -1: unused slot; < -1: hit the step limit; accumulates across calls
    int status;                     // RAY_STATUS_* bits
    int rdot_sign, thetadot_sign;   // current branch of the two square roots
    int rdot_flips;                 // radial turning points met
    int equatorial_crossings;       // crossings of theta = pi/2
    T alpha, beta;                  // source-specific labels (PointSource: cos(alpha), beta; ImagePlane: x, y)
};

template <typename T>
class RayDestination;

enum class Integrator { Euler, RK4, RK45 };

template <typename T>
class Raytracer {
public:
    Ray<T>* rays;   // nRays records, owned by this object

    Raytracer(int num_rays, T spin, T precision = PRECISION, T init_max_phistep = MAXDPHI, T init_max_tstep = MAXDT);
    ~Raytracer();

    // ---- the hot path: integrate every ray until it stops (GPU) -----------------------------------------------
    // stop at theta_max (> 0: theta >= theta_max; < 0: theta <= |theta_max|; 0: never), r_max, the horizon or steplim
    void run_raytrace(Integrator method = Integrator::Euler,
                      T theta_max = M_PI / 2,
                      T r_max = 1000,
                      int show_progress = 1,
                      TextOutput* outfile = 0,
                      int write_step = 1,
                      T write_rmax = -1,
                      T write_rmin = -1,
                      bool write_cartesian = true,
                      int steplim = -1);
    // stop on a RayDestination surface instead (RK4 / RK45 only)
    void run_raytrace(RayDestination<T>* dest,
                      Integrator method = Integrator::Euler,
                      T r_max = 1000,
                      int show_progress = 1,
                      TextOutput* outfile = 0,
                      int write_step = 1,
                      T write_rmax = -1,
                      T write_rmin = -1,
                      bool write_cartesian = true,
                      int steplim = -1);

    // single-ray forms of the reference API; each is a one-ray launch of the same kernels and returns the steps taken
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

    // ---- O(N) passes either side of the trace --------------------------------------------------------------------
    void redshift_start(T V, bool reverse = false, bool projradius = false);                    // before: emitted energy
    void redshift(T V, bool reverse = false, bool projradius = false, int motion = 0);          // after: energy ratio
    void redshift(RayDestination<T>* dest, bool reverse = false, bool projradius = false, int motion = 0);
    T ray_redshift(T V, bool reverse, bool projradius, T r, T theta, T phi, T k, T h, T Q, int rdot_sign, int thetadot_sign, T emit,
                   int motion = 0);
    T ray_redshift(const T et[4], bool reverse, T r, T theta, T phi, T k, T h, T Q, int rdot_sign, int thetadot_sign, T emit);
    void range_phi(T min = -1 * M_PI, T max = M_PI);                                            // wrap phi into [min, max)
    void calculate_momentum();                                                                    // momenta from (k, h, Q) in place

    // ---- small accessors -------------------------------------------------------------------------------------------
    int get_count() { return nRays; }   // slots in rays[]; some may be unused (steps == -1)

    void set_boundary(T r = -1)         // inner radius rays cannot pass; default: the event horizon
    {
        if (r > 0)
            horizon = r;
        else
            horizon = kerr_horizon<T>(spin);
    }
    T calculate_horizon() { return kerr_horizon<T>(spin); }

    // The reference's two setters assign their parameters to themselves (raytracer.h:169-178) and so change nothing;
    // an application calling them must keep getting the constructor's precision, hence they change nothing here.
    void set_precision(T precision) { (void) precision; }
    void set_precision(T precision, T theta_precision)
    {
        (void) precision;
        (void) theta_precision;
    }

    void set_rk45_tol(T tol) { rk45_tol = tol; }       // DOPRI5 mixed abs/rel tolerance per step (default 1e-8)
    T get_rk45_tol() const { return rk45_tol; }
    void set_max_tstep(T max, T rlim = MAXDT_RLIM)
    {
        max_tstep = max;
        maxtstep_rlim = rlim;
    }
    void set_max_phistep(T max) { max_phistep = max; }

protected:   // ray sources derive from this class and fill rays[]
    int nRays;
    T spin;
    T horizon;

    void calculate_constants(int ray, T alpha, T beta, T V, T E);                 // (k, h, Q) of a ray leaving an orbiting source
    void calculate_constants_from_p(int ray, T pt, T pr, T ptheta, T pphi);      // (k, h, Q) from a 4-momentum

private:
    T precision;
    T theta_precision;
    T max_tstep;
    T max_phistep;
    T maxtstep_rlim;
    T rk45_tol;

    void fill_params(void* kr_params_out, Integrator method, T r_max, int steplim) const;   // -> kr_params of the C ABI
    void trace(const void* kr_params_in, Ray<T>* first, long n, int show_progress);         // kr_trace_f64 / kr_trace_f32 (kr_trace_progress_* while reporting)
};

#endif /* RAYTRACER_H_ */
