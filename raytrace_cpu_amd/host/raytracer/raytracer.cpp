// raytracer/raytracer.cpp -- Raytracer<T> members of the host-side API mirror.
//
// run_raytrace (both overloads; reference raytracer.cpp:63-127, :972-1034) and the O(N) passes (redshift_start :342-417,
// redshift :420-477, range_phi :603-622, calculate_momentum :704-753) are calls into libkrtrace.so (include/kr_trace.h)
// and execute on the GPU, for Raytracer<double> (kr_*_f64) and for the float instantiation (reference raytracer.cpp:1897;
// kr_*_f32) alike.  Host code is left only where the reference calls back into per-ray virtual functions: the ray_redshift()
// helpers and redshift() with a user-defined velocity field.

#include "raytracer.h"

#include <sys/ioctl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <map>
#include <mutex>
#include <thread>
#include <type_traits>
#include <typeinfo>

#include "../../../include/kr_trace.h"
#include "ray_destination.h"

static_assert(sizeof(Ray<double>) == sizeof(kr_ray_f64), "Ray<double> must be layout-identical to kr_ray_f64");
static_assert(sizeof(Ray<float>) == sizeof(kr_ray_f32), "Ray<float> must be layout-identical to kr_ray_f32");

namespace {

// KR_TIMING=1: wall-clock marks on stderr, relative to the first mark, around what the mirror does (scripts/app_wall.sh)
void mark(const char* what)
{
    static const bool on = std::getenv("KR_TIMING") != nullptr;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "kr_timing: t = %8.1f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
}

[[noreturn]] void fail(const char* what, int rc)
{
    const char* msg = kr_last_error();
    throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + (msg ? msg : ""));
}

inline void check(int rc, const char* what)
{
    if (rc != KR_OK) fail(what, rc);
}

// The line run_raytrace draws while it works (reference src/include/progress_bar.h:25-75 as used by raytracer.cpp:84-85, :107-115, :126: label "Ray",
// the count right-aligned to the width of the total, a bar as wide as the terminal allows; show_progress < 0: one plain line per report).  The count
// is the number of rays whose integration has STARTED, as in the reference; here it is read from the trace's work queue while the kernels run.
class ProgressLine {
public:
    ProgressLine(long end, bool draw_bar) : end_(end), draw_bar_(draw_bar)
    {
        for (long e = end; e; e /= 10) ++digits_;
        struct winsize w;
        const int cols = ioctl(STDOUT_FILENO, TIOCGWINSZ, &w) == 0 ? w.ws_col : 0;       // (not a terminal: no bar cells)
        length_ = cols - 3 - 2 * digits_ - 5;
    }
    void show(long at)
    {
        std::ostringstream line;
        if (draw_bar_) {
            line << "\033[?25l" << "\r" << "Ray" << ' ' << std::setw(digits_) << at << '/' << end_ << " [";
            for (int cell = 1; cell <= length_; cell++) line << (((float) cell / length_) <= ((float) at / end_) ? "\u2588" : "-");
            line << ']' << "\033[?25h";
        } else {
            line << "Ray" << ' ' << std::setw(digits_) << at << '/' << end_ << std::endl;
        }
        std::cout << line.str() << std::flush;
    }
    void done() { std::cout << std::endl; }

private:
    long end_;
    int length_ = 0, digits_ = 0;
    bool draw_bar_;
};
void progress_cb(int64_t at, int64_t, void* user) { static_cast<ProgressLine*>(user)->show((long) at); }

int trace_call(const kr_params* p, Ray<double>* rays, long n, long every, ProgressLine* line)
{
    if (every > 0 && line) return kr_trace_progress_f64(p, reinterpret_cast<kr_ray_f64*>(rays), n, nullptr, every, progress_cb, line);
    return kr_trace_f64(p, reinterpret_cast<kr_ray_f64*>(rays), n, nullptr);
}
int trace_call(const kr_params* p, Ray<float>* rays, long n, long every, ProgressLine* line)
{
    if (every > 0 && line) return kr_trace_progress_f32(p, reinterpret_cast<kr_ray_f32*>(rays), n, nullptr, every, progress_cb, line);
    return kr_trace_f32(p, reinterpret_cast<kr_ray_f32*>(rays), n, nullptr);
}

template <typename T>
bool builtin_destination(const RayDestination<T>* dest, int& kind, double sp[4], bool& default_velocity)
{
    if (!dest) return false;
    const std::type_info& ti = typeid(*dest);
    const bool exact = ti == typeid(FlatDiscDestination<T>) || ti == typeid(DiscWithISCODestination<T>) || ti == typeid(FlatPlaneDestination<T>);
    return exact && dest->describe(kind, sp, default_velocity);
}

void no_outfile(const TextOutput* outfile)
{
    if (outfile != nullptr)
        throw std::runtime_error("Raytracer: per-step trajectory output (outfile != nullptr) is not supported by the HIP path; "
                                 "it is an ordered per-step file write that the reference runs serially on the CPU");
}

// The HIP runtime takes ~0.13 s to come up in a fresh process.  It does so on a thread of its own, started by the first constructor, while
// the constructing thread touches rays[] and the ray source fills it; whoever needs the device first (the first pass over an array) waits for
// it and gives the array its device residency (kr_host_attach) then.
struct DeviceWarmUp {
    std::thread th;
    std::mutex mu;
    bool started = false;
    std::map<const void*, std::pair<long, int>> pending;      // host arrays not attached yet: base -> (records, record bytes)
    void start()
    {
        std::lock_guard<std::mutex> lk(mu);
        // (an unchanged reference program owns its process and this is its first HIP user: more hardware queues for overlapping launches,
        // include/kr_trace.h::kr_configure_process -- a no-op when the user chose GPU_MAX_HW_QUEUES or the runtime is already up)
        if (!started) { started = true; (void) kr_configure_process(); th = std::thread([] { (void) kr_device_count(); }); }
    }
    void announce(const void* rays, long n, int bytes)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (n > 0) pending[rays] = {n, bytes};
    }
    // before the first library call on `rays`.  whole_array: the call is a pass over rays[] -- only then is the device residency worth
    // n x 176 B on the device; a program that only ever calls propagate*() on single rays stages just those records.
    void ready(const void* rays, bool whole_array = true)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (th.joinable()) th.join();
        if (!whole_array) return;
        auto it = pending.find(rays);
        if (it != pending.end()) {
            const int rc = kr_host_attach(const_cast<void*>(rays), it->second.first, (int32_t) it->second.second);
            pending.erase(it);
            if (rc != KR_OK)      // not fatal: every pass then stages through a private buffer (slower); say so once
                std::fprintf(stderr, "raytrace_cpu_amd: no device residency for rays[] (%s); passes will stage through temporary buffers\n", kr_last_error());
            else
                mark("attached (device buffer)");
        }
    }
    // destructor of an array's owner: true if the array never reached the device
    bool forget(const void* rays)
    {
        std::lock_guard<std::mutex> lk(mu);
        return pending.erase(rays) > 0;
    }
    ~DeviceWarmUp() { if (th.joinable()) th.join(); }
};
DeviceWarmUp g_warm_up;

}  // namespace

template <typename T>
Raytracer<T>::Raytracer(int num_rays, T spin_par, T init_precision, T init_max_phistep, T init_max_tstep)
    : precision(init_precision),
      theta_precision(THETA_PRECISION),
      max_tstep(init_max_tstep),
      max_phistep(init_max_phistep),
      maxtstep_rlim(MAXDT_RLIM),
      rk45_tol(T(1e-8)),
      nRays(num_rays),
      spin(spin_par)
{
    mark("Raytracer ctor: begin");
    horizon = kerr_horizon<T>(spin);
    // Every record zeroed, steps = -1 (the reference leaves most fields indeterminate).  At 1e7 rays the array is 1.44 GB: `new
    // Ray<T>[n]()` + a second serial pass cost ~0.35 s of page faults and stores on one core -- more than the whole GPU trace --
    // so the memory is taken raw and first touched by all host threads at once (Ray<T> is trivially constructible).  (Any array the
    // reference's delete[] would have released is released here by the matching free(): applications never free `rays` themselves.)
    static_assert(std::is_trivially_copyable<Ray<T>>::value && std::is_trivially_destructible<Ray<T>>::value, "Ray<T> must be a plain record");
    const size_t bytes = sizeof(Ray<T>) * static_cast<size_t>(nRays > 0 ? nRays : 0);
    // 2-MB aligned and advised for transparent huge pages where the kernel allows: 512 x fewer page faults on the first touch
    const size_t huge = size_t(2) << 20;
    void* mem = nullptr;
    if (posix_memalign(&mem, bytes >= huge ? huge : 64, bytes ? ((bytes + huge - 1) / huge) * huge : 64) != 0) throw std::bad_alloc();
#ifdef MADV_HUGEPAGE
    if (bytes >= huge) (void) madvise(mem, ((bytes + huge - 1) / huge) * huge, MADV_HUGEPAGE);
#endif
    rays = static_cast<Ray<T>*>(mem);
    g_warm_up.start();                  // the HIP runtime comes up beside the first touch and the source constructor
#pragma omp parallel for schedule(static) num_threads(kr_host_threads())
    for (int ray = 0; ray < nRays; ray++) {
        std::memset(static_cast<void*>(&rays[ray]), 0, sizeof(Ray<T>));
        rays[ray].steps = -1;
    }
    mark("Raytracer ctor: rays[] allocated and first touched");
    g_warm_up.announce(rays, nRays, (int) sizeof(Ray<T>));      // device residency is set up by the first pass (DeviceWarmUp::ready)
}

template <typename T>
Raytracer<T>::~Raytracer()
{
    mark("Raytracer dtor: begin");
    if (!g_warm_up.forget(rays)) (void) kr_host_detach(rays);
    std::free(static_cast<void*>(rays));
    mark("Raytracer dtor: end");
}

// Arithmetic of the double-precision trace (include/kr_trace.h), chosen by the environment so that
// the reference's applications need no new option: KRTRACE_ARITHMETIC = hybrid | strict | fast.  Unset:
//   * theta-limit overloads, Euler / RK4: hybrid;  RK45: strict -- the adaptive step control amplifies the few-ulp differences of the
//     fast arithmetic into +-1 differences of the step counts, and an RK45 launch is bounded by its longest ray either way;
//   * run_raytrace(RayDestination*), every integrator: STRICT.  A destination that does not stop rays at their first equatorial crossing
//     (DiscWithISCO inside the ISCO / beyond r_out, FlatPlane) lets them whirl near the photon sphere and come back; at a <= 0.5 that is
//     0.1-0.35 % of a lamp post's rays, and the reference's own result for them is decided at the 1-ulp level (a 1-ulp change of Q on the
//     CPU: 946 of 1e6 rays with another integer outcome, tests/tool_oracle_isco_noise.py).  Only the arithmetic that carries the
//     reference's bits reproduces them (strict: 0 integer differences on 13 geometries x 1e6 rays, profiles/r03_hybrid_sweep_rk4_isco.jsonl);
//     no in-flight criterion separates those rays from the 7-18 % that merely pass inside the ISCO.  Cost: 1.3-2.8 x the hybrid launch.
static int arithmetic_flags(int integrator, bool destination)
{
    const char* e = std::getenv("KRTRACE_ARITHMETIC");
    if (!e || !*e) return (destination || integrator == KR_RK45) ? 0 : KR_FLAG_HYBRID;
    if (!std::strcmp(e, "hybrid")) return KR_FLAG_HYBRID;
    if (!std::strcmp(e, "strict")) return 0;
    if (!std::strcmp(e, "fast")) return KR_FLAG_FAST_MATH;
    throw std::invalid_argument(std::string("KRTRACE_ARITHMETIC: expected hybrid, strict or fast, got '") + e + "'");
}

template <typename T>
void Raytracer<T>::fill_params(void* out, Integrator method, T r_max, int steplim) const
{
    kr_params* p = static_cast<kr_params*>(out);
    std::memset(p, 0, sizeof(*p));
    p->spin = spin;
    p->horizon = horizon;
    p->precision = precision;
    p->theta_precision = theta_precision;
    p->max_tstep = max_tstep;
    p->maxtstep_rlim = maxtstep_rlim;
    p->max_phistep = max_phistep;
    p->rk45_tol = rk45_tol;
    p->r_max = r_max;
    p->theta_max = 0;
    p->integrator = static_cast<int>(method);   // Euler, RK4, RK45 == KR_EULER, KR_RK4, KR_RK45
    p->stop_kind = KR_STOP_THETA;
    p->steplim = steplim;                        // <= 0 selects STEPLIM / RK45_STEPLIM inside the library
    // KRTRACE_STEPLIM=<n>: the limit used where the application passes none (the reference's STEPLIM is 1e7 for Euler / RK4).  A ray that runs
    // to that limit -- a photon trapped between radial turning points inside the ISCO, which run_raytrace(RayDestination*) does not stop: 1 ray in
    // 1e6 on two of thirteen test geometries -- is 1e7 SEQUENTIAL steps: ~1 s on a CPU core, ~15 s on a GPU wave, and the launch cannot end
    // before it does.  Every consumer drops such rays (steps < 0); a lower limit only marks them STEPLIM sooner.
    if (steplim <= 0) {
        static const int env_steplim = [] { const char* e = std::getenv("KRTRACE_STEPLIM"); return e ? std::atoi(e) : 0; }();
        if (env_steplim > 0 && (method != Integrator::RK45 || env_steplim < RK45_STEPLIM)) p->steplim = env_steplim;
    }
    p->flags = arithmetic_flags(p->integrator, false);      // (the RayDestination overloads choose again once the destination is known)
}

template <typename T>
void Raytracer<T>::trace(const void* params, Ray<T>* first, long n, int show_progress)
{
    if (n > 1) mark("run_raytrace: begin");
    g_warm_up.ready(rays, n > 1);
    if (n > 1) {
        // what run_raytrace writes on stdout (raytracer.cpp:76-85, :126): the integrator's name, the progress line, a newline
        static const char* names[] = {"Running raytracer...", "Running raytracer (RK4)...", "Running raytracer (RK45/DOPRI5)..."};
        std::cout << names[static_cast<const kr_params*>(params)->integrator] << std::endl;
        ProgressLine line(n, show_progress > 0);
        check(trace_call(static_cast<const kr_params*>(params), first, n, std::abs((long) show_progress), &line), "kr_trace");
        line.done();
    } else {
        check(trace_call(static_cast<const kr_params*>(params), first, n, 0, nullptr), "kr_trace");
    }
    if (n > 1) mark("run_raytrace: end");
}

template <typename T>
void Raytracer<T>::run_raytrace(Integrator method, T theta_max, T r_max, int show_progress, TextOutput* outfile, int write_step,
                                T write_rmax, T write_rmin, bool write_cartesian, int steplim)
{
    (void) write_step; (void) write_rmax; (void) write_rmin; (void) write_cartesian;
    no_outfile(outfile);
    kr_params p;
    fill_params(&p, method, r_max, steplim);
    p.theta_max = theta_max;
    trace(&p, rays, nRays, show_progress);
}

template <typename T>
void Raytracer<T>::run_raytrace(RayDestination<T>* dest, Integrator method, T r_max, int show_progress, TextOutput* outfile,
                                int write_step, T write_rmax, T write_rmin, bool write_cartesian, int steplim)
{
    (void) write_step; (void) write_rmax; (void) write_rmin; (void) write_cartesian;
    no_outfile(outfile);
    if (method == Integrator::Euler)   // assert in the reference, raytracer.cpp:983
        throw std::invalid_argument("Integrator::Euler does not support RayDestination stopping conditions");
    kr_params p;
    fill_params(&p, method, r_max, steplim);
    bool default_velocity = false;
    if (!builtin_destination(dest, p.stop_kind, p.stop_params, default_velocity))
        throw std::runtime_error("Raytracer::run_raytrace: only FlatDiscDestination, DiscWithISCODestination and FlatPlaneDestination "
                                 "can be evaluated by the HIP kernel; a user-defined RayDestination would need per-step host callbacks");
    p.flags = arithmetic_flags(p.integrator, true);
    trace(&p, rays, nRays, show_progress);
}

// ---- single-ray forms --------------------------------------------------------------------------------------
// The reference's propagate*() take the ray as it is: no skip rule (a ray with steps == -1, or one already past the step limit, is
// traced all the same; that rule lives in run_raytrace, raytracer.cpp:116-117) and `steplim` verbatim (<= 0: zero iterations and
// RAY_STATUS_STEPLIM, :172 / :315-316).  The kernel implements run_raytrace, so the ray is staged with a zero step count, which
// always passes the skip rule, and the epilogue's bookkeeping (:335-337) is redone here on the caller's own count.
#define KR_SINGLE(method_, theta_expr, dest_expr)                                                       \
    no_outfile(outfile);                                                                                \
    (void) write_step; (void) write_rmax; (void) write_rmin; (void) write_cartesian;                   \
    kr_params p;                                                                                        \
    fill_params(&p, method_, rlim, steplim);                                                            \
    theta_expr;                                                                                         \
    dest_expr;                                                                                          \
    const int before = rays[ray].steps;                                                                 \
    int taken = 0;                                                                                      \
    if (steplim <= 0) {                                                                                 \
        rays[ray].status |= KR_STATUS_STEPLIM;                                                          \
    } else {                                                                                            \
        rays[ray].steps = 0;                                                                            \
        trace(&p, &rays[ray], 1, 0);                                                                     \
        taken = std::abs(rays[ray].steps);                                                              \
    }                                                                                                   \
    rays[ray].steps = before + taken;                                                                   \
    if (rays[ray].status & KR_STATUS_STEPLIM) rays[ray].steps = -rays[ray].steps;                       \
    return taken;

template <typename T>
int Raytracer<T>::propagate(int ray, const T rlim, const T thetalim, const int steplim, TextOutput* outfile, int write_step, T write_rmax,
                            T write_rmin, bool write_cartesian)
{
    KR_SINGLE(Integrator::Euler, p.theta_max = thetalim, (void) 0)
}

template <typename T>
int Raytracer<T>::propagate_rk4(int ray, const T rlim, const T thetalim, const int steplim, TextOutput* outfile, int write_step,
                                T write_rmax, T write_rmin, bool write_cartesian)
{
    KR_SINGLE(Integrator::RK4, p.theta_max = thetalim, (void) 0)
}

template <typename T>
int Raytracer<T>::propagate_rk45(int ray, const T rlim, const T thetalim, const int steplim, TextOutput* outfile, int write_step,
                                 T write_rmax, T write_rmin, bool write_cartesian)
{
    KR_SINGLE(Integrator::RK45, p.theta_max = thetalim, (void) 0)
}

#define KR_DEST_OR_THROW                                                                                 \
    bool dv = false;                                                                                     \
    if (!builtin_destination(dest, p.stop_kind, p.stop_params, dv)) throw std::runtime_error("unsupported RayDestination subclass"); \
    p.flags = arithmetic_flags(p.integrator, true)

template <typename T>
int Raytracer<T>::propagate_rk4(int ray, const T rlim, RayDestination<T>* dest, const int steplim, TextOutput* outfile, int write_step,
                                T write_rmax, T write_rmin, bool write_cartesian)
{
    KR_SINGLE(Integrator::RK4, (void) 0, KR_DEST_OR_THROW)
}

template <typename T>
int Raytracer<T>::propagate_rk45(int ray, const T rlim, RayDestination<T>* dest, const int steplim, TextOutput* outfile, int write_step,
                                 T write_rmax, T write_rmin, bool write_cartesian)
{
    KR_SINGLE(Integrator::RK45, (void) 0, KR_DEST_OR_THROW)
}

// ---- per-ray helpers (host), reference raytracer.cpp:480-600 ---------------------------------------------------
namespace {

template <typename T>
struct HostMetric {
    T g[16];
    krhost::BLCoefficients<T> m;
    HostMetric(T r, T theta, T a) : m(r, theta, a)
    {
        for (int i = 0; i < 16; i++) g[i] = 0;
        g[0] = m.e2nu - m.omega * m.omega * m.e2psi;
        g[3] = m.omega * m.e2psi;
        g[12] = g[3];
        g[5] = -m.rhosq / m.delta;
        g[10] = -m.rhosq;
        g[15] = -m.e2psi;
    }
    T contract(const T* et, const T* p) const
    {
        T e = 0;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) e += g[i * 4 + j] * et[i] * p[j];
        return e;
    }
};

}  // namespace

template <typename T>
T Raytracer<T>::ray_redshift(T V, bool reverse, bool projradius, T r, T theta, T phi, T k, T h, T Q, int rdot_sign, int thetadot_sign,
                             T emit, int motion)
{
    const T a = (reverse) ? -1 * spin : spin;
    const HostMetric<T> gm(r, theta, a);
    T et[] = {0, 0, 0, 0};
    if (motion == 0) {
        if (V == -1 && projradius)
            V = 1 / (a + r * sin(theta) * sqrt(r * sin(theta)));
        else if (V == -1)
            V = 1 / (a + r * sqrt(r));
        et[0] = (1 / sqrt(gm.m.e2nu)) / sqrt(1 - (V - gm.m.omega) * (V - gm.m.omega) * gm.m.e2psi / gm.m.e2nu);
        et[3] = (1 / sqrt(gm.m.e2nu)) * V / sqrt(1 - (V - gm.m.omega) * (V - gm.m.omega) * gm.m.e2psi / gm.m.e2nu);
    } else if (motion == 1) {
        if (V < 0) V = abs(V) * (r * r - 2 * r + spin + spin) / (r * r + spin * spin);   // as in the reference (:531)
        et[0] = 1. / sqrt(gm.g[0] + gm.g[5] * V * V);
        et[1] = V * et[0];
    }
    T p[4];
    momentum_from_consts<T>(p[0], p[1], p[2], p[3], k, h, Q, rdot_sign, thetadot_sign, r, theta, phi, spin);
    if (reverse) { p[1] *= -1; p[2] *= -1; p[3] *= -1; }
    const T recv = gm.contract(et, p);
    return (reverse) ? recv / emit : emit / recv;
}

template <typename T>
T Raytracer<T>::ray_redshift(const T et[4], bool reverse, T r, T theta, T phi, T k, T h, T Q, int rdot_sign, int thetadot_sign, T emit)
{
    const HostMetric<T> gm(r, theta, spin);
    T p[4];
    momentum_from_consts<T>(p[0], p[1], p[2], p[3], k, h, Q, rdot_sign, thetadot_sign, r, theta, phi, spin);
    if (reverse) { p[1] *= -1; p[2] *= -1; p[3] *= -1; }
    const T recv = gm.contract(et, p);
    return (reverse) ? recv / emit : emit / recv;
}

// ---- O(N) passes: streaming kernels over rays[] for both instantiations (kr_*_f64 / kr_*_f32) -----------------------
// Only a user-defined RayDestination::four_velocity() keeps a host loop (a per-ray virtual call, as in the reference).
namespace {

struct PassF64 {
    using ray = kr_ray_f64;
    static int redshift_start(double a, double V, int rev, int pr, ray* r, int64_t n) { return kr_redshift_start_f64(a, V, rev, pr, r, n); }
    static int redshift(double a, double V, int rev, int pr, int motion, ray* r, int64_t n) { return kr_redshift_f64(a, V, rev, pr, motion, r, n); }
    static int redshift_dest(double a, int rev, ray* r, int64_t n) { return kr_redshift_dest_f64(a, rev, r, n); }
    static int range_phi(double lo, double hi, ray* r, int64_t n) { return kr_range_phi_f64(lo, hi, r, n); }
    static int calculate_momentum(double a, ray* r, int64_t n) { return kr_calculate_momentum_f64(a, r, n); }
};
struct PassF32 {
    using ray = kr_ray_f32;
    static int redshift_start(double a, double V, int rev, int pr, ray* r, int64_t n) { return kr_redshift_start_f32(a, V, rev, pr, r, n); }
    static int redshift(double a, double V, int rev, int pr, int motion, ray* r, int64_t n) { return kr_redshift_f32(a, V, rev, pr, motion, r, n); }
    static int redshift_dest(double a, int rev, ray* r, int64_t n) { return kr_redshift_dest_f32(a, rev, r, n); }
    static int range_phi(double lo, double hi, ray* r, int64_t n) { return kr_range_phi_f32(lo, hi, r, n); }
    static int calculate_momentum(double a, ray* r, int64_t n) { return kr_calculate_momentum_f32(a, r, n); }
};
template <typename T> struct PassOf;
template <> struct PassOf<double> { using type = PassF64; };
template <> struct PassOf<float> { using type = PassF32; };

template <typename T>
inline typename PassOf<T>::type::ray* as_kr(Ray<T>* r) { return reinterpret_cast<typename PassOf<T>::type::ray*>(r); }

}  // namespace

template <typename T>
void Raytracer<T>::redshift_start(T V, bool reverse, bool projradius)
{
    mark("redshift_start: begin");
    g_warm_up.ready(rays);
    check(PassOf<T>::type::redshift_start(spin, V, reverse, projradius, as_kr<T>(rays), nRays), "kr_redshift_start");
    mark("redshift_start: end");
}

template <typename T>
void Raytracer<T>::redshift(T V, bool reverse, bool projradius, int motion)
{
    mark("redshift: begin");
    g_warm_up.ready(rays);
    check(PassOf<T>::type::redshift(spin, V, reverse, projradius, motion, as_kr<T>(rays), nRays), "kr_redshift");
    mark("redshift: end");
}

template <typename T>
void Raytracer<T>::redshift(RayDestination<T>* dest, bool reverse, bool projradius, int motion)
{
    (void) projradius; (void) motion;
    int kind = 0;
    double sp[4];
    bool default_velocity = false;
    if (builtin_destination(dest, kind, sp, default_velocity) && default_velocity) {
        g_warm_up.ready(rays);
        check(PassOf<T>::type::redshift_dest(spin, reverse, as_kr<T>(rays), nRays), "kr_redshift_dest");
        return;
    }
    // user-defined velocity field: per-ray virtual call, host loop as in the reference (:467-476)
    for (int ray = 0; ray < nRays; ray++) {
        Ray<T>& R = rays[ray];
        T et[4];
        dest->four_velocity(R.r, R.theta, R.phi, spin, et);
        R.redshift = ray_redshift(et, reverse, R.r, R.theta, R.phi, R.k, R.h, R.Q, R.rdot_sign, R.thetadot_sign, R.emit);
    }
}

template <typename T>
void Raytracer<T>::range_phi(T min, T max)
{
    mark("range_phi: begin");
    g_warm_up.ready(rays);
    check(PassOf<T>::type::range_phi(min, max, as_kr<T>(rays), nRays), "kr_range_phi");
    mark("range_phi: end");
}

template <typename T>
void Raytracer<T>::calculate_momentum()
{
    g_warm_up.ready(rays);
    check(PassOf<T>::type::calculate_momentum(spin, as_kr<T>(rays), nRays), "kr_calculate_momentum");
}

// ---- constants of motion for the ray sources (host, O(N)); reference raytracer.cpp:625-701 ------------------------
template <typename T>
void Raytracer<T>::calculate_constants(int ray, T alpha, T beta, T V, T E)
{
    Ray<T>& R = rays[ray];
    const T r = R.r, th = R.theta;
    const krhost::BLCoefficients<T> m(r, th, spin);
    const T e2nu = m.e2nu, e2psi = m.e2psi, omega = m.omega, rhosq = m.rhosq, delta = m.delta;

    // orbiting source's tetrad
    const T et0 = (1 / sqrt(e2nu)) / sqrt(1 - (V - omega) * (V - omega) * e2psi / e2nu);
    const T et3 = (1 / sqrt(e2nu)) * V / sqrt(1 - (V - omega) * (V - omega) * e2psi / e2nu);
    const T e10 = (V - omega) * sqrt(e2psi / e2nu) / sqrt(e2nu - (V - omega) * (V - omega) * e2psi);
    const T e13 = (1 / sqrt(e2nu * e2psi)) * (e2nu + V * omega * e2psi - omega * omega * e2psi) / sqrt(e2nu - (V - omega) * (V - omega) * e2psi);
    const T e22 = -1 / sqrt(rhosq);
    const T e31 = sqrt(delta / rhosq);

    // photon 4-momentum in the source frame and its Boyer-Lindquist components
    const T q[] = {E, E * sin(alpha) * cos(beta), E * sin(alpha) * sin(beta), E * cos(alpha)};
    const T tdot = q[0] * et0 + q[1] * e10;
    const T phidot = q[0] * et3 + q[1] * e13;
    const T rdot = q[3] * e31;
    const T thetadot = q[2] * e22;

    R.k = (1 - 2 * r / rhosq) * tdot + (2 * spin * r * sin(th) * sin(th) / rhosq) * phidot;
    R.h = phidot * ((r * r + spin * spin) * (r * r + spin * spin * cos(th) * cos(th) - 2 * r) * sin(th) * sin(th) +
                    2 * spin * spin * r * sin(th) * sin(th) * sin(th) * sin(th));
    R.h = R.h - 2 * spin * r * R.k * sin(th) * sin(th);
    R.h = R.h / (r * r + spin * spin * cos(th) * cos(th) - 2 * r);
    R.Q = rhosq * rhosq * thetadot * thetadot - (spin * R.k * cos(th) + R.h / tan(th)) * (spin * R.k * cos(th) - R.h / tan(th));

    R.rdot_sign = (rdot >= 0) ? 1 : -1;
    R.thetadot_sign = (thetadot > 0) ? 1 : -1;
    R.rdot_flips = 0;
    R.equatorial_crossings = 0;
}

template <typename T>
void Raytracer<T>::calculate_constants_from_p(int ray, T pt, T pr, T ptheta, T pphi)
{
    (void) pt;
    Ray<T>& R = rays[ray];
    const T a = spin, r = R.r, theta = R.theta;
    const T rhosq = r * r + (a * cos(theta)) * (a * cos(theta));
    T k = (1 - 2 * r / rhosq) * pr + (2 * a * r * sin(theta) * sin(theta) / rhosq) * pphi;   // pr, as in the reference (:690)
    T h = pphi * ((r * r + a * a) * (r * r + a * a * cos(theta) * cos(theta) - 2 * r) * sin(theta) * sin(theta) +
                  2 * a * a * r * sin(theta) * sin(theta) * sin(theta) * sin(theta));
    h = h - 2 * a * r * k * sin(theta) * sin(theta);
    h = h / (r * r + a * a * cos(theta) * cos(theta) - 2 * r);
    const T Q = rhosq * rhosq * ptheta * ptheta - (a * k * cos(theta) + h / tan(theta)) * (a * k * cos(theta) - h / tan(theta));
    R.k = k;
    R.h = h;
    R.Q = Q;
}

template class Raytracer<double>;
template class Raytracer<float>;
