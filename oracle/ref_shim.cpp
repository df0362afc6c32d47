// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A thin extern "C" handle API around the *unmodified* reference classes
// (PointSource<double>, ImagePlane<double>, Raytracer<double>, the three
// RayDestination kinds).  It is compiled together with the reference's own
// raytracer.cpp / pointsource.cpp / imageplane.cpp, which are read where they
// lie under $(REF)/src (never copied), into oracle/_ref/libkr_ref.so by
// oracle/Makefile.  Python (tests, tests/golden/make_golden.py, bench.py's
// cpu_baseline leg) drives it through ctypes.
//
// Nothing in raytrace_cpu_amd/ may link or load this file's output.
//
// Reference interfaces wrapped (paths relative to the reference tree):
//   src/raytracer/pointsource.h:24      PointSource ctor
//   src/raytracer/imageplane.h:26       ImagePlane ctor
//   src/raytracer/raytracer.h:112,119   run_raytrace (theta-limit / RayDestination)
//   src/raytracer/raytracer.h:132-138   redshift_start / redshift / range_phi
//   src/raytracer/ray_destination.h:86,116,173  the three concrete destinations

#include <cstdint>
#include <cstring>
#include <iostream>
#include <sstream>

#include "raytracer/pointsource.h"
#include "raytracer/imageplane.h"
#include "raytracer/ray_destination.h"

namespace {

// the reference chats on cout from ctors and every phase; silence it while inside the shim
struct Quiet {
    std::streambuf* old;
    std::ostringstream sink;
    Quiet() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~Quiet() { std::cout.rdbuf(old); }
};

struct Handle {
    Raytracer<double>* base = nullptr;
    PointSource<double>* ps = nullptr;
    ImagePlane<double>* ip = nullptr;
};

RayDestination<double>* make_dest(int kind, const double* p)
{
    switch (kind) {
        case 1: return new FlatDiscDestination<double>(p[0]);
        case 2: return new DiscWithISCODestination<double>(p[0], p[1], p[2]);   // (r_isco, r_out, theta_lim)
        case 3: return new FlatPlaneDestination<double>(p[0], p[1], p[2]);      // (incl, phi0, z_s)
        default: return nullptr;
    }
}

Integrator method_of(int m) { return m == 0 ? Integrator::Euler : m == 1 ? Integrator::RK4 : Integrator::RK45; }

} // namespace

extern "C" {

int ref_sizeof_ray() { return (int) sizeof(Ray<double>); }

void* ref_pointsource_new(const double* pos, double V, double spin, double tol, double dcosalpha, double dbeta,
                          double cosalpha0, double cosalphamax, double beta0, double betamax, double E)
{
    Quiet q;
    double p[4] = {pos[0], pos[1], pos[2], pos[3]};
    Handle* h = new Handle;
    h->ps = new PointSource<double>(p, V, spin, tol, dcosalpha, dbeta, cosalpha0, cosalphamax, beta0, betamax, E);
    h->base = h->ps;
    return h;
}

void* ref_imageplane_new(double dist, double inc_deg, double x0, double xmax, double dx, double y0, double ymax,
                         double dy, double spin, double phi, double precision)
{
    Quiet q;
    Handle* h = new Handle;
    h->ip = new ImagePlane<double>(dist, inc_deg, x0, xmax, dx, y0, ymax, dy, spin, phi, precision);
    h->base = h->ip;
    return h;
}

void ref_free(void* hv)
{
    Quiet q;
    Handle* h = (Handle*) hv;
    if (h->ps) delete h->ps;
    if (h->ip) delete h->ip;
    delete h;
}

int ref_count(void* hv) { return ((Handle*) hv)->base->get_count(); }
void* ref_rays(void* hv) { return ((Handle*) hv)->base->rays; }

void ref_set_rk45_tol(void* hv, double tol) { ((Handle*) hv)->base->set_rk45_tol(tol); }
void ref_set_max_tstep(void* hv, double m, double rlim) { ((Handle*) hv)->base->set_max_tstep(m, rlim); }
void ref_set_max_phistep(void* hv, double m) { ((Handle*) hv)->base->set_max_phistep(m); }
void ref_set_boundary(void* hv, double r) { ((Handle*) hv)->base->set_boundary(r); }

// Raytracer<T>::redshift_start(V, reverse, projradius)  raytracer.cpp:342
void ref_redshift_start(void* hv, double V, int reverse, int projradius)
{
    Quiet q;
    ((Handle*) hv)->base->redshift_start(V, reverse != 0, projradius != 0);
}

// the source-specific no-argument forms (pointsource.cpp:66, imageplane.cpp:123)
void ref_redshift_start_source(void* hv)
{
    Quiet q;
    Handle* h = (Handle*) hv;
    if (h->ps) h->ps->redshift_start();
    else h->ip->redshift_start();
}

void ref_run_thetalim(void* hv, int method, double theta_max, double r_max, int steplim)
{
    Quiet q;
    ((Handle*) hv)->base->run_raytrace(method_of(method), theta_max, r_max, 0, nullptr, 1, -1, -1, true, steplim);
}

int ref_run_dest(void* hv, int method, int dest_kind, const double* dest_params, double r_max, int steplim)
{
    Quiet q;
    RayDestination<double>* d = make_dest(dest_kind, dest_params);
    if (!d) return -1;
    ((Handle*) hv)->base->run_raytrace(d, method_of(method), r_max, 0, nullptr, 1, -1, -1, true, steplim);
    delete d;
    return 0;
}

void ref_redshift(void* hv, double V, int reverse, int projradius, int motion)
{
    Quiet q;
    ((Handle*) hv)->base->redshift(V, reverse != 0, projradius != 0, motion);
}

int ref_redshift_dest(void* hv, int dest_kind, const double* dest_params, int reverse)
{
    Quiet q;
    RayDestination<double>* d = make_dest(dest_kind, dest_params);
    if (!d) return -1;
    ((Handle*) hv)->base->redshift(d, reverse != 0);
    delete d;
    return 0;
}

void ref_range_phi(void* hv, double lo, double hi)
{
    Quiet q;
    ((Handle*) hv)->base->range_phi(lo, hi);
}

void ref_calculate_momentum(void* hv)
{
    Quiet q;
    ((Handle*) hv)->base->calculate_momentum();
}

// small header-only helpers from src/include/kerr.h used to pin the restatement's constants
double ref_kerr_horizon(double a) { return kerr_horizon<double>(a); }
double ref_kerr_isco(double a, int sign) { return kerr_isco<double>(a, sign); }
double ref_disc_velocity(double r, double a, int sign) { return disc_velocity<double>(r, a, sign); }

} // extern "C"

// ---- Raytracer<float> (the reference's second explicit instantiation, raytracer.cpp:1897) ---------------------
namespace {
struct HandleF {
    Raytracer<float>* base = nullptr;
    PointSource<float>* ps = nullptr;
    ImagePlane<float>* ip = nullptr;
};
RayDestination<float>* make_dest_f(int kind, const double* p)
{
    switch (kind) {
        case 1: return new FlatDiscDestination<float>((float) p[0]);
        case 2: return new DiscWithISCODestination<float>((float) p[0], (float) p[1], (float) p[2]);
        case 3: return new FlatPlaneDestination<float>((float) p[0], (float) p[1], (float) p[2]);
        default: return nullptr;
    }
}
} // namespace

extern "C" {

int ref_sizeof_ray_f32() { return (int) sizeof(Ray<float>); }

void* ref_pointsource_new_f32(const double* pos, double V, double spin, double tol, double dcosalpha, double dbeta,
                              double cosalpha0, double cosalphamax, double beta0, double betamax, double E)
{
    Quiet q;
    float p[4] = {(float) pos[0], (float) pos[1], (float) pos[2], (float) pos[3]};
    HandleF* h = new HandleF;
    h->ps = new PointSource<float>(p, (float) V, (float) spin, (float) tol, (float) dcosalpha, (float) dbeta, (float) cosalpha0,
                                   (float) cosalphamax, (float) beta0, (float) betamax, (float) E);
    h->base = h->ps;
    return h;
}

void* ref_imageplane_new_f32(double dist, double inc_deg, double x0, double xmax, double dx, double y0, double ymax, double dy,
                             double spin, double phi, double precision)
{
    Quiet q;
    HandleF* h = new HandleF;
    h->ip = new ImagePlane<float>((float) dist, (float) inc_deg, (float) x0, (float) xmax, (float) dx, (float) y0, (float) ymax,
                                  (float) dy, (float) spin, (float) phi, (float) precision);
    h->base = h->ip;
    return h;
}

void ref_free_f32(void* hv)
{
    Quiet q;
    HandleF* h = (HandleF*) hv;
    if (h->ps) delete h->ps;
    if (h->ip) delete h->ip;
    delete h;
}

int ref_count_f32(void* hv) { return ((HandleF*) hv)->base->get_count(); }
void* ref_rays_f32(void* hv) { return ((HandleF*) hv)->base->rays; }
void ref_set_rk45_tol_f32(void* hv, double tol) { ((HandleF*) hv)->base->set_rk45_tol((float) tol); }

void ref_redshift_start_f32(void* hv, double V, int reverse, int projradius)
{
    Quiet q;
    ((HandleF*) hv)->base->redshift_start((float) V, reverse != 0, projradius != 0);
}

// the O(N) passes of the float instantiation (raytracer.cpp:420-477, :603-622, :704-753)
void ref_redshift_f32(void* hv, double V, int reverse, int projradius, int motion)
{
    Quiet q;
    ((HandleF*) hv)->base->redshift((float) V, reverse != 0, projradius != 0, motion);
}

int ref_redshift_dest_f32(void* hv, int dest_kind, const double* dest_params, int reverse)
{
    Quiet q;
    RayDestination<float>* d = make_dest_f(dest_kind, dest_params);
    if (!d) return -1;
    ((HandleF*) hv)->base->redshift(d, reverse != 0);
    delete d;
    return 0;
}

void ref_range_phi_f32(void* hv, double lo, double hi)
{
    Quiet q;
    ((HandleF*) hv)->base->range_phi((float) lo, (float) hi);
}

void ref_calculate_momentum_f32(void* hv)
{
    Quiet q;
    ((HandleF*) hv)->base->calculate_momentum();
}

void ref_run_thetalim_f32(void* hv, int method, double theta_max, double r_max, int steplim)
{
    Quiet q;
    ((HandleF*) hv)->base->run_raytrace(method_of(method), (float) theta_max, (float) r_max, 0, nullptr, 1, -1, -1, true, steplim);
}

int ref_run_dest_f32(void* hv, int method, int dest_kind, const double* dest_params, double r_max, int steplim)
{
    Quiet q;
    RayDestination<float>* d = make_dest_f(dest_kind, dest_params);
    if (!d) return -1;
    ((HandleF*) hv)->base->run_raytrace(d, method_of(method), (float) r_max, 0, nullptr, 1, -1, -1, true, steplim);
    delete d;
    return 0;
}

} // extern "C"

