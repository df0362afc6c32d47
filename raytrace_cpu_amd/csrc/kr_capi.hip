// kr_capi.hip -- the extern "C" surface declared in include/kr_trace.h.
// Host-pointer entry points stage rays through a private device buffer (H2D, kernels, D2H) so that the
// caller's rays[] is up to date on return, which is the contract of the reference's member functions.
// There is no CPU implementation behind any of these: without a HIP device they return KR_ENODEVICE.

#include <hip/hip_runtime.h>

#include <atomic>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kr_common.hpp"
#include "kr_post_device.hpp"

namespace kr {

// implemented in kr_post.hip
int redshift_start_dev(double spin, double V, int reverse, int projradius, void* d, int64_t n, hipStream_t st, bool f32 = false);
int redshift_dev(double spin, double V, int reverse, int projradius, int motion, void* d, int64_t n, hipStream_t st, bool f32 = false);
int redshift_dest_dev(double spin, int reverse, void* d, int64_t n, hipStream_t st, bool f32 = false);
int range_phi_dev(double lo, double hi, void* d, int64_t n, hipStream_t st, bool f32 = false);
int calculate_momentum_dev(double spin, void* d, int64_t n, hipStream_t st, bool f32 = false);
int pointsource_init_dev(const kr_pointsource* s, void* d, int64_t n, int64_t first, int64_t stride, hipStream_t st);
int imageplane_init_dev(const kr_imageplane* s, void* d, int64_t n, int64_t first, int64_t stride, hipStream_t st);
int reduce_emissivity_dev(const kr_emis_bins* b, const void* d, int64_t n, void* d_hist, hipStream_t st);
int post_emissivity_dev(double spin, double V, int reverse, int projradius, int motion, double lo, double hi, const kr_emis_bins* b, void* d, int64_t n,
                        void* d_hist, hipStream_t st);
int imageplane_init_emit_dev(const kr_imageplane* s, void* d, int64_t n, int64_t first, int64_t stride, int64_t run, double spin, double V, int reverse,
                             int projradius, hipStream_t st);
int post_image_dev(double spin, double V, int reverse, int projradius, int motion, double lo, double hi, const kr_image_bins* b, void* d, int64_t n,
                   void* d_planes, hipStream_t st);
int pointsource_init_emit_dev(const kr_pointsource* s, void* d, int64_t n, int64_t first, int64_t stride, double V, int reverse, int projradius, hipStream_t st);
int reduce_image_dev(const kr_image_bins* b, const void* d, int64_t n, void* d_planes, hipStream_t st);
int reduce_return_dev(const kr_return_bins* b, const void* d, int64_t n, void* d_out4, hipStream_t st);
int post_return_dev(double lo, double hi, const kr_return_bins* b, void* d, int64_t n, void* d_out4, hipStream_t st);
int post_return_batch_dev(int count, double lo, double hi, const kr_return_bins* b, void* const* d, const int64_t* n, void* const* d_out4, hipStream_t st);
int pointsource_init_emit_batch_dev(int count, const kr_pointsource* s, const double* V, int reverse, int projradius, void* const* d, const int64_t* n, hipStream_t st);
int arith_probe_dev(int op, const double* a, const double* b, double* out, int64_t n);

static thread_local std::string g_error;

// (hardware queues: include/kr_trace.h, kr_configure_process -- the library no longer edits the environment when it is loaded)
static std::atomic<bool> g_runtime_touched{false};     // any entry point that reaches the HIP runtime sets it
// (Nothing is released from a library destructor: at process exit the HIP runtime's own exit handlers may already have run -- they are
// registered when the runtime starts, i.e. after this library's -- and calling into a torn-down runtime can hang.  kr_shutdown() is explicit.)

void set_error(const std::string& msg) { g_error = msg; }

int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
    char buf[512];
    std::snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    g_error = buf;
    (void) hipGetLastError();
    return (e == hipErrorOutOfMemory) ? KR_ENOMEM : (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? KR_ENODEVICE : KR_EHIP;
}

int require_device()
{
    g_runtime_touched = true;
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void) hipGetLastError();
        g_error = "no HIP device available (libkrtrace has no CPU fallback)";
        return KR_ENODEVICE;
    }
    return KR_OK;
}

int DeviceBuffer::alloc(size_t bytes)
{
    KR_HIP(hipMalloc(&p, bytes ? bytes : 1));
    return KR_OK;
}

namespace {

using clk = std::chrono::steady_clock;
double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

// ---- attached host arrays ----------------------------------------------------------------------------------------------
// A caller that keeps ONE host ray array through several passes (the class API: Raytracer<T>::rays lives as long as the object)
// attaches it once: a device buffer of the same size is kept for it, and every host-pointer entry point called on it (a) allocates
// nothing, (b) copies back only the bytes its pass modifies -- one 8-byte field per record for redshift_start / range_phi /
// redshift -- through a compact buffer: field-gather kernel, 8 n bytes over PCIe, scatter into rays[] by host threads.
// (Page-locking the array in place with hipHostRegister was measured and dropped: transfers do reach 57 GB/s, but registering
// 1.44 GB costs ~60 ms, the first transfer out of it 140 ms, unregistering ~100 ms -- more than four passes save;
// the runtime's own path for pageable memory already runs at 30-55 GB/s after the first touch: profiles/r02_app_wall.txt.)
// The host array is still the input of every call (it is uploaded each time: an application may have written to it) and is
// complete when the call returns, which is the reference's contract.
struct Attached {
    void* host = nullptr;
    int64_t n = 0;
    size_t ray_bytes = 0;
    void* dev = nullptr;          // n * ray_bytes
    void* d_field = nullptr;      // n * 32: the widest partial write-back (four momenta)
    void* h_field = nullptr;      // host side of it
};
std::mutex g_att_mu;
std::map<const void*, Attached> g_attached;

__global__ void __launch_bounds__(256) gather_field_kernel(const char* __restrict__ rays, long long n, int ray_bytes, int off, int words, double* __restrict__ out)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long) gridDim.x * 256) {
        const double* src = reinterpret_cast<const double*>(rays + i * ray_bytes + off);
        for (int w = 0; w < words; w++) out[i * words + w] = src[w];
    }
}

void scatter_field_host(char* rays, int64_t n, size_t ray_bytes, int off, int words, const double* field)
{
    const int nthreads = (int) std::max<int64_t>(1, std::min<int64_t>(8, n / 65536));
    auto work = [&](int t) {
        const int64_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
        for (int64_t i = lo; i < hi; i++) std::memcpy(rays + i * ray_bytes + off, field + i * words, (size_t) words * 8);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
}

// what a pass writes into the records it was given
struct WriteBack {
    int off = 0;         // byte offset of the modified field(s) inside a record
    int words = 0;       // 8-byte words; 0 = the whole record
};
constexpr WriteBack kWholeRecord = {0, 0};

// run `body(d_rays)` on a device copy of a host ray array and copy the result back
template <typename Body>
int with_staged_rays(void* rays, int64_t n, size_t ray_bytes, bool copy_in, bool copy_out, kr_stats* stats, Body body, WriteBack wb = kWholeRecord)
{
    if (n < 0 || (n > 0 && !rays)) { set_error("null rays pointer or negative n"); return KR_EINVAL; }
    int rc = require_device();
    if (rc != KR_OK) return rc;
    if (n == 0) return body(nullptr);
    static const bool timing = getenv("KR_TIMING") != nullptr;      // per-call breakdown on stderr (scripts/app_wall.sh)
    Attached att;
    {
        std::lock_guard<std::mutex> lk(g_att_mu);
        // an attached array, or a sub-range of one (the single-ray propagate() forms pass &rays[i])
        auto it = g_attached.upper_bound(rays);
        if (it != g_attached.begin()) {
            --it;
            const char* base = (const char*) it->second.host;
            const char* p = (const char*) rays;
            if (it->second.ray_bytes == ray_bytes && p >= base && p + (size_t) n * ray_bytes <= base + (size_t) it->second.n * ray_bytes &&
                (size_t) (p - base) % ray_bytes == 0)
                att = it->second;
        }
    }
    auto t0 = clk::now();
    DeviceBuffer buf;
    char* d = nullptr;
    if (att.dev) {
        d = (char*) att.dev + ((const char*) rays - (const char*) att.host);
    } else {
        rc = buf.alloc((size_t) n * ray_bytes);
        if (rc != KR_OK) return rc;
        d = (char*) buf.p;
    }
    const double t_alloc = ms_since(t0);
    t0 = clk::now();
    if (copy_in) KR_HIP(hipMemcpy(d, rays, (size_t) n * ray_bytes, hipMemcpyHostToDevice));
    const double h2d = ms_since(t0);
    t0 = clk::now();
    rc = body((void*) d);
    if (rc != KR_OK) return rc;
    KR_HIP(hipDeviceSynchronize());
    const double t_body = ms_since(t0);
    t0 = clk::now();
    if (copy_out) {
        if (att.dev && wb.words > 0 && wb.words <= 4 && n >= 4096) {
            hipLaunchKernelGGL(gather_field_kernel, dim3((unsigned) std::min<int64_t>((n + 255) / 256, 65536)), dim3(256), 0, nullptr, (const char*) d, (long long) n,
                               (int) ray_bytes, wb.off, wb.words, (double*) att.d_field);
            KR_HIP(hipGetLastError());
            KR_HIP(hipMemcpy(att.h_field, att.d_field, (size_t) n * wb.words * 8, hipMemcpyDeviceToHost));
            scatter_field_host((char*) rays, n, ray_bytes, wb.off, wb.words, (const double*) att.h_field);
        } else {
            KR_HIP(hipMemcpy(rays, d, (size_t) n * ray_bytes, hipMemcpyDeviceToHost));
        }
    }
    const double d2h = ms_since(t0);
    if (stats) { stats->h2d_ms = h2d; stats->d2h_ms = d2h; }
    if (timing) std::fprintf(stderr, "kr_timing: staged call n=%lld%s alloc %.1f ms h2d %.1f ms kernels %.1f ms d2h %.1f ms%s\n", (long long) n, att.dev ? " (attached)" : "",
                             t_alloc, h2d, t_body, d2h, (att.dev && wb.words > 0) ? " (one field)" : "");
    return KR_OK;
}

}  // namespace
}  // namespace kr

using namespace kr;

extern "C" {

int kr_abi_version(void) { return KR_ABI_VERSION; }
const char* kr_last_error(void) { return g_error.c_str(); }

int kr_device_count(void)
{
    int n = 0;
    g_runtime_touched = true;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void) hipGetLastError();
        set_error("no HIP device available");
        return KR_ENODEVICE;
    }
    return n;
}

int kr_set_device(int device)
{
    KR_HIP(hipSetDevice(device));
    return KR_OK;
}

int kr_device_info(int* cus, int* clock_khz, int64_t* hbm_bytes, char* name, int name_len)
{
    int rc = require_device();
    if (rc != KR_OK) return rc;
    int dev = 0;
    KR_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    KR_HIP(hipGetDeviceProperties(&prop, dev));
    if (cus) *cus = prop.multiProcessorCount;
    if (clock_khz) *clock_khz = prop.clockRate;
    if (hbm_bytes) *hbm_bytes = (int64_t) prop.totalGlobalMem;
    if (name && name_len > 0) {
        std::snprintf(name, (size_t) name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    return KR_OK;
}

// Raytracer<T> ctor defaults, raytracer.cpp:12-22 + raytracer.h:19-44; horizon = kerr_horizon(spin), kerr.h:14-20
void kr_params_default(kr_params* p, double spin)
{
    std::memset(p, 0, sizeof(*p));
    p->spin = spin;
    p->horizon = 1 + std::sqrt((1 - spin) * (1 + spin));
    p->precision = KR_PRECISION;
    p->theta_precision = KR_THETA_PRECISION;
    p->max_tstep = KR_MAXDT;
    p->maxtstep_rlim = KR_MAXDT_RLIM;
    p->max_phistep = KR_MAXDPHI;
    p->rk45_tol = 1e-8;
    p->r_max = 1000;
    p->theta_max = 1.57079632679489661923;
    p->integrator = KR_EULER;
    p->stop_kind = KR_STOP_THETA;
    p->steplim = -1;
}

// src/include/kerr.h:14-20
double kr_kerr_horizon(double a) { return 1 + std::sqrt((1 - a) * (1 + a)); }

// src/include/kerr.h:23-32: A and B are `const float`, and with `using namespace std` in force the last sqrt
// is the float overload, so the ISCO radius the apps bin against is a float-precision number
// (r_isco(0.998) = 1.2369706630706787).  Reproduced, not fixed: it is a bin edge.
double kr_kerr_isco(double a, int sign)
{
    const float A = (float) (1. + std::pow(1. - a * a, 1. / 3.) * (std::pow(1. + a, 1. / 3.) + std::pow(1. - a, 1. / 3.)));
    const float AA = A * A;
    const float B = (float) std::sqrt(3. * a * a + AA);
    const float inner = (3 - A) * (3 + A + 2 * B);
    const float res = 3 + B - sign * std::sqrt(inner);
    return res;
}

// src/include/kerr.h:35-38
double kr_disc_velocity(double r, double a, int sign) { return 1 / (a + sign * std::pow(r, 3. / 2.)); }

// nRays is the int-truncated PRODUCT of doubles (pointsource.cpp:12), n_cosalpha / n_beta the truncated factors (:16-17)
int64_t kr_pointsource_count(const kr_pointsource* s, int32_t* n_cosalpha, int32_t* n_beta)
{
    const int nrays = (int) ((((s->cosalphamax - s->cosalpha0) / s->dcosalpha) + 1) * (((s->betamax - s->beta0) / s->dbeta) + 1));
    if (n_cosalpha) *n_cosalpha = (int) (((s->cosalphamax - s->cosalpha0) / s->dcosalpha) + 1);
    if (n_beta) *n_beta = (int) (((s->betamax - s->beta0) / s->dbeta) + 1);
    return nrays;
}

// What the device PointSource constructor reads instead of calling acos / sin / cos / tan itself (kr_post_device.hpp::SourceTables): host values.
int kr_pointsource_tables(const kr_pointsource* s, double* alpha_sincos, double* beta_sincos, double* pos_sin_cos_tan)
{
    if (!s) { set_error("kr_pointsource_tables: null spec"); return KR_EINVAL; }
    int32_t nc = 0, nb = 0;
    kr_pointsource_count(s, &nc, &nb);
    if (alpha_sincos) angle_values(0, s->cosalpha0, s->dcosalpha, nc, alpha_sincos);
    if (beta_sincos) angle_values(1, s->beta0, s->dbeta, nb, beta_sincos);
    if (pos_sin_cos_tan) { ::sincos(s->pos[2], &pos_sin_cos_tan[0], &pos_sin_cos_tan[1]); pos_sin_cos_tan[2] = std::tan(s->pos[2]); }
    return KR_OK;
}

// imageplane.cpp:12-14
int64_t kr_imageplane_count(const kr_imageplane* s, int32_t* nx, int32_t* ny)
{
    const int nrays = (int) ((((s->xmax - s->x0) / s->dx) + 1) * (((s->ymax - s->y0) / s->dy) + 1));
    if (nx) *nx = (int) (((s->xmax - s->x0) / s->dx) + 1);
    if (ny) *ny = (int) (((s->ymax - s->y0) / s->dy) + 1);
    return nrays;
}

// ---- trace ---------------------------------------------------------------------------------------------
int kr_trace_dev_f64(const kr_params* p, void* d_rays, int64_t n, void* stream, kr_stats* stats)
{
    return trace_dev(p, d_rays, n, (hipStream_t) stream, stats, false);
}

int kr_trace_dev_f32(const kr_params* p, void* d_rays, int64_t n, void* stream, kr_stats* stats)
{
    return trace_dev(p, d_rays, n, (hipStream_t) stream, stats, true);
}

int kr_trace_async_f64(const kr_params* p, void* d_rays, int64_t n, void* stream, void** ticket)
{
    if (!ticket) { set_error("kr_trace_async: null ticket pointer"); return KR_EINVAL; }
    return trace_async(p, d_rays, n, (hipStream_t) stream, false, ticket);
}

int kr_trace_async_f32(const kr_params* p, void* d_rays, int64_t n, void* stream, void** ticket)
{
    if (!ticket) { set_error("kr_trace_async: null ticket pointer"); return KR_EINVAL; }
    return trace_async(p, d_rays, n, (hipStream_t) stream, true, ticket);
}

int kr_trace_batch_async_f64(int32_t count, const kr_params* const* p, void* const* d_rays, const int64_t* n, void* const* streams, void** tickets)
{
    return trace_batch_async(count, p, d_rays, n, streams, tickets);
}

int kr_trace_wait(void* ticket, kr_stats* stats) { return trace_wait(ticket, stats); }

int kr_trace_wait_many(int32_t count, void* const* tickets, kr_stats* per_ticket, kr_stats* total)
{
    if (count < 0 || (count > 0 && !tickets)) { set_error("kr_trace_wait_many: null argument"); return KR_EINVAL; }
    if (total) std::memset(total, 0, sizeof(*total));
    int first_rc = KR_OK;
    for (int32_t i = 0; i < count; i++) {
        kr_stats st;
        const int rc = trace_wait(tickets[i], (per_ticket || total) ? &st : nullptr);
        if (rc != KR_OK) { if (first_rc == KR_OK) first_rc = rc; continue; }
        if (per_ticket) per_ticket[i] = st;
        if (total) {
            total->rays_total += st.rays_total; total->rays_traced += st.rays_traced; total->steps_total += st.steps_total;
            total->rk45_attempts += st.rk45_attempts; total->rk45_rejects += st.rk45_rejects; total->rays_strict_side += st.rays_strict_side;
            total->rk45_stationary_steps += st.rk45_stationary_steps; total->rk45_extrapolated_steps += st.rk45_extrapolated_steps;
            total->steps_strict_side += st.steps_strict_side; total->rk45_evaluated_strict_side += st.rk45_evaluated_strict_side;
            total->kernel_ms = std::max(total->kernel_ms, st.kernel_ms);
            total->strict_side_ms = std::max(total->strict_side_ms, st.strict_side_ms);
            total->main_ms = std::max(total->main_ms, st.main_ms);
            total->longest_ray_steps = std::max(total->longest_ray_steps, st.longest_ray_steps);
            total->longest_ray_steps_strict_side = std::max(total->longest_ray_steps_strict_side, st.longest_ray_steps_strict_side);
        }
    }
    return first_rc;
}

int kr_trace_release(void* ticket)
{
    trace_release(ticket);
    return KR_OK;
}

int kr_trace_f64(const kr_params* p, kr_ray_f64* rays, int64_t n, kr_stats* stats)
{
    if (!p) { set_error("kr_trace: null params"); return KR_EINVAL; }
    if (stats) std::memset(stats, 0, sizeof(*stats));
    return with_staged_rays(rays, n, sizeof(kr_ray_f64), true, true, stats,
                            [&](void* d) { return trace_dev(p, d, n, nullptr, stats, false); });
}

int kr_trace_f32(const kr_params* p, kr_ray_f32* rays, int64_t n, kr_stats* stats)
{
    if (!p) { set_error("kr_trace: null params"); return KR_EINVAL; }
    if (stats) std::memset(stats, 0, sizeof(*stats));
    return with_staged_rays(rays, n, sizeof(kr_ray_f32), true, true, stats,
                            [&](void* d) { return trace_dev(p, d, n, nullptr, stats, true); });
}

// run_raytrace with show_progress != 0 (raytracer.cpp:84-85, :107-115): the same trace, and while it runs the calling thread polls the work
// queue and reports every multiple of `every` rays it sees passed
namespace {
int trace_with_progress(const kr_params* p, void* d, int64_t n, bool f32, kr_stats* stats, int64_t every, kr_progress_fn fn, void* user)
{
    void* ticket = nullptr;
    int rc = trace_async(p, d, n, nullptr, f32, &ticket);
    if (rc != KR_OK) return rc;
    int64_t shown = 0;
    for (;;) {
        int64_t started = 0;
        int32_t fin = 1;
        rc = trace_poll(ticket, &started, &fin);
        if (rc != KR_OK) break;                       // (the trace itself is still waited for below)
        if (fn && every > 0) {
            const int64_t at = started / every * every;
            if (at > shown) { shown = at; fn(at, n, user); }
        }
        if (fin) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
    const int rc2 = trace_wait(ticket, stats);
    if (stats) stats->rays_total = n;
    return rc != KR_OK ? rc : rc2;
}
}  // namespace

int kr_trace_progress_f64(const kr_params* p, kr_ray_f64* rays, int64_t n, kr_stats* stats, int64_t every, kr_progress_fn fn, void* user)
{
    if (!p) { set_error("kr_trace: null params"); return KR_EINVAL; }
    if (stats) std::memset(stats, 0, sizeof(*stats));
    return with_staged_rays(rays, n, sizeof(kr_ray_f64), true, true, stats, [&](void* d) { return trace_with_progress(p, d, n, false, stats, every, fn, user); });
}
int kr_trace_progress_f32(const kr_params* p, kr_ray_f32* rays, int64_t n, kr_stats* stats, int64_t every, kr_progress_fn fn, void* user)
{
    if (!p) { set_error("kr_trace: null params"); return KR_EINVAL; }
    if (stats) std::memset(stats, 0, sizeof(*stats));
    return with_staged_rays(rays, n, sizeof(kr_ray_f32), true, true, stats, [&](void* d) { return trace_with_progress(p, d, n, true, stats, every, fn, user); });
}
int kr_trace_poll(void* ticket, int64_t* rays_started, int32_t* finished)
{
    return trace_poll(ticket, rays_started, finished);
}

// ---- O(N) passes -------------------------------------------------------------------------------------------
int kr_redshift_start_dev_f64(double spin, double V, int reverse, int projradius, void* d, int64_t n, void* st)
{
    int rc = require_device();
    return rc != KR_OK ? rc : redshift_start_dev(spin, V, reverse, projradius, d, n, (hipStream_t) st);
}
int kr_redshift_start_f64(double spin, double V, int reverse, int projradius, kr_ray_f64* rays, int64_t n)
{
    return with_staged_rays(rays, n, sizeof(kr_ray_f64), true, true, nullptr,
                            [&](void* d) { return redshift_start_dev(spin, V, reverse, projradius, d, n, nullptr); }, WriteBack{offsetof(kr_ray_f64, emit), 1});
}

int kr_redshift_dev_f64(double spin, double V, int reverse, int projradius, int motion, void* d, int64_t n, void* st)
{
    int rc = require_device();
    return rc != KR_OK ? rc : redshift_dev(spin, V, reverse, projradius, motion, d, n, (hipStream_t) st);
}
int kr_redshift_f64(double spin, double V, int reverse, int projradius, int motion, kr_ray_f64* rays, int64_t n)
{
    return with_staged_rays(rays, n, sizeof(kr_ray_f64), true, true, nullptr,
                            [&](void* d) { return redshift_dev(spin, V, reverse, projradius, motion, d, n, nullptr); }, WriteBack{offsetof(kr_ray_f64, redshift), 1});
}

int kr_redshift_dest_dev_f64(double spin, int reverse, void* d, int64_t n, void* st)
{
    int rc = require_device();
    return rc != KR_OK ? rc : redshift_dest_dev(spin, reverse, d, n, (hipStream_t) st);
}
int kr_redshift_dest_f64(double spin, int reverse, kr_ray_f64* rays, int64_t n)
{
    return with_staged_rays(rays, n, sizeof(kr_ray_f64), true, true, nullptr,
                            [&](void* d) { return redshift_dest_dev(spin, reverse, d, n, nullptr); }, WriteBack{offsetof(kr_ray_f64, redshift), 1});
}

int kr_range_phi_dev_f64(double lo, double hi, void* d, int64_t n, void* st)
{
    int rc = require_device();
    return rc != KR_OK ? rc : range_phi_dev(lo, hi, d, n, (hipStream_t) st);
}
int kr_range_phi_f64(double lo, double hi, kr_ray_f64* rays, int64_t n)
{
    return with_staged_rays(rays, n, sizeof(kr_ray_f64), true, true, nullptr,
                            [&](void* d) { return range_phi_dev(lo, hi, d, n, nullptr); }, WriteBack{offsetof(kr_ray_f64, phi), 1});
}

int kr_calculate_momentum_dev_f64(double spin, void* d, int64_t n, void* st)
{
    int rc = require_device();
    return rc != KR_OK ? rc : calculate_momentum_dev(spin, d, n, (hipStream_t) st);
}
int kr_calculate_momentum_f64(double spin, kr_ray_f64* rays, int64_t n)
{
    return with_staged_rays(rays, n, sizeof(kr_ray_f64), true, true, nullptr,
                            [&](void* d) { return calculate_momentum_dev(spin, d, n, nullptr); }, WriteBack{offsetof(kr_ray_f64, pt), 4});
}

// The same passes for Raytracer<float> (kr_ray_f32 records, float arithmetic; the scalars are float values carried in doubles).
// Host-pointer forms copy the whole 84-byte record back.
int kr_redshift_start_dev_f32(double spin, double V, int reverse, int projradius, void* d, int64_t n, void* st)
{
    int rc = require_device();
    return rc != KR_OK ? rc : redshift_start_dev(spin, V, reverse, projradius, d, n, (hipStream_t) st, true);
}
int kr_redshift_start_f32(double spin, double V, int reverse, int projradius, kr_ray_f32* rays, int64_t n)
{
    return with_staged_rays(rays, n, sizeof(kr_ray_f32), true, true, nullptr,
                            [&](void* d) { return redshift_start_dev(spin, V, reverse, projradius, d, n, nullptr, true); });
}
int kr_redshift_dev_f32(double spin, double V, int reverse, int projradius, int motion, void* d, int64_t n, void* st)
{
    int rc = require_device();
    return rc != KR_OK ? rc : redshift_dev(spin, V, reverse, projradius, motion, d, n, (hipStream_t) st, true);
}
int kr_redshift_f32(double spin, double V, int reverse, int projradius, int motion, kr_ray_f32* rays, int64_t n)
{
    return with_staged_rays(rays, n, sizeof(kr_ray_f32), true, true, nullptr,
                            [&](void* d) { return redshift_dev(spin, V, reverse, projradius, motion, d, n, nullptr, true); });
}
int kr_redshift_dest_dev_f32(double spin, int reverse, void* d, int64_t n, void* st)
{
    int rc = require_device();
    return rc != KR_OK ? rc : redshift_dest_dev(spin, reverse, d, n, (hipStream_t) st, true);
}
int kr_redshift_dest_f32(double spin, int reverse, kr_ray_f32* rays, int64_t n)
{
    return with_staged_rays(rays, n, sizeof(kr_ray_f32), true, true, nullptr,
                            [&](void* d) { return redshift_dest_dev(spin, reverse, d, n, nullptr, true); });
}
int kr_range_phi_dev_f32(double lo, double hi, void* d, int64_t n, void* st)
{
    int rc = require_device();
    return rc != KR_OK ? rc : range_phi_dev(lo, hi, d, n, (hipStream_t) st, true);
}
int kr_range_phi_f32(double lo, double hi, kr_ray_f32* rays, int64_t n)
{
    return with_staged_rays(rays, n, sizeof(kr_ray_f32), true, true, nullptr,
                            [&](void* d) { return range_phi_dev(lo, hi, d, n, nullptr, true); });
}
int kr_calculate_momentum_dev_f32(double spin, void* d, int64_t n, void* st)
{
    int rc = require_device();
    return rc != KR_OK ? rc : calculate_momentum_dev(spin, d, n, (hipStream_t) st, true);
}
int kr_calculate_momentum_f32(double spin, kr_ray_f32* rays, int64_t n)
{
    return with_staged_rays(rays, n, sizeof(kr_ray_f32), true, true, nullptr,
                            [&](void* d) { return calculate_momentum_dev(spin, d, n, nullptr, true); });
}

// ---- sources -----------------------------------------------------------------------------------------------
int kr_pointsource_init_dev_f64(const kr_pointsource* s, void* d, int64_t n, void* st)
{
    if (!s) { set_error("kr_pointsource_init: null spec"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : pointsource_init_dev(s, d, n, 0, 1, (hipStream_t) st);
}
int kr_pointsource_init_f64(const kr_pointsource* s, kr_ray_f64* rays, int64_t n)
{
    if (!s) { set_error("kr_pointsource_init: null spec"); return KR_EINVAL; }
    return with_staged_rays(rays, n, sizeof(kr_ray_f64), false, true, nullptr,
                            [&](void* d) { return pointsource_init_dev(s, d, n, 0, 1, nullptr); });
}

int kr_imageplane_init_dev_f64(const kr_imageplane* s, void* d, int64_t n, void* st)
{
    if (!s) { set_error("kr_imageplane_init: null spec"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : imageplane_init_dev(s, d, n, 0, 1, (hipStream_t) st);
}
int kr_imageplane_init_f64(const kr_imageplane* s, kr_ray_f64* rays, int64_t n)
{
    if (!s) { set_error("kr_imageplane_init: null spec"); return KR_EINVAL; }
    return with_staged_rays(rays, n, sizeof(kr_ray_f64), false, true, nullptr,
                            [&](void* d) { return imageplane_init_dev(s, d, n, 0, 1, nullptr); });
}

// strided forms: slot k of d_rays receives ray (first + k*stride) of the source's array -- the multi-GPU shard of rank r
// of R is (first = r, stride = R, count = ceil((total - r) / R)); no rank ever materialises another rank's rays.
int kr_pointsource_init_strided_dev_f64(const kr_pointsource* s, int64_t first, int64_t stride, void* d, int64_t count, void* st)
{
    if (!s) { set_error("kr_pointsource_init: null spec"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : pointsource_init_dev(s, d, count, first, stride, (hipStream_t) st);
}
// fused pipeline ends (device-resident callers): source constructor + redshift_start() in one pass ...
int kr_pointsource_init_emit_dev_f64(const kr_pointsource* s, int64_t first, int64_t stride, double V, int reverse, int projradius, void* d, int64_t count, void* st)
{
    if (!s) { set_error("kr_pointsource_init_emit: null spec"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : pointsource_init_emit_dev(s, d, count, first, stride, V, reverse, projradius, (hipStream_t) st);
}
int kr_pointsource_init_emit_batch_dev_f64(int32_t count, const kr_pointsource* s, const double* V, int reverse, int projradius, void* const* d, const int64_t* n, void* st)
{
    if (count < 0 || (count > 0 && (!s || !d || !n))) { set_error("kr_pointsource_init_emit_batch: null argument"); return KR_EINVAL; }
    for (int32_t i = 0; i < count; i++)
        if (n[i] > 0 && !d[i]) { set_error("kr_pointsource_init_emit_batch: null ray buffer"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : pointsource_init_emit_batch_dev(count, s, V, reverse, projradius, d, n, (hipStream_t) st);
}
// ... and range_phi() + redshift() + the emissivity histogram in one pass
int kr_post_emissivity_dev_f64(double spin, double V, int reverse, int projradius, int motion, double lo, double hi, const kr_emis_bins* b, void* d, int64_t n,
                               void* d_hist, void* st)
{
    if (!b || !d_hist) { set_error("kr_post_emissivity: null argument"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : post_emissivity_dev(spin, V, reverse, projradius, motion, lo, hi, b, d, n, d_hist, (hipStream_t) st);
}
int kr_imageplane_init_emit_dev_f64(const kr_imageplane* s, int64_t first, int64_t stride, double V, int reverse, int projradius, void* d, int64_t count, void* st)
{
    if (!s) { set_error("kr_imageplane_init_emit: null spec"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : imageplane_init_emit_dev(s, d, count, first, stride, 1, -1 * s->spin, V, reverse, projradius, (hipStream_t) st);
}
int kr_imageplane_init_emit_runs_dev_f64(const kr_imageplane* s, int64_t first, int64_t stride, int64_t run, double V, int reverse, int projradius, void* d,
                                         int64_t count, void* st)
{
    if (!s) { set_error("kr_imageplane_init_emit_runs: null spec"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : imageplane_init_emit_dev(s, d, count, first, stride, run, -1 * s->spin, V, reverse, projradius, (hipStream_t) st);
}
int kr_post_image_dev_f64(double spin, double V, int reverse, int projradius, int motion, double lo, double hi, const kr_image_bins* b, void* d, int64_t n,
                          void* d_planes, void* st)
{
    if (!b || !d_planes) { set_error("kr_post_image: null argument"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : post_image_dev(spin, V, reverse, projradius, motion, lo, hi, b, d, n, d_planes, (hipStream_t) st);
}
int kr_imageplane_init_strided_dev_f64(const kr_imageplane* s, int64_t first, int64_t stride, void* d, int64_t count, void* st)
{
    if (!s) { set_error("kr_imageplane_init: null spec"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : imageplane_init_dev(s, d, count, first, stride, (hipStream_t) st);
}

// ---- reducers ------------------------------------------------------------------------------------------------
int kr_reduce_emissivity_dev_f64(const kr_emis_bins* b, const void* d, int64_t n, void* d_hist, void* st)
{
    if (!b || !d_hist) { set_error("kr_reduce_emissivity: null argument"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : reduce_emissivity_dev(b, d, n, d_hist, (hipStream_t) st);
}

int kr_reduce_emissivity_f64(const kr_emis_bins* b, const kr_ray_f64* rays, int64_t n, int64_t* count, double* flux,
                             double* emis, double* sum_redshift, double* sum_time, int64_t* disc_count)
{
    if (!b || !count || !flux || !emis || !sum_redshift || !sum_time) { set_error("kr_reduce_emissivity: null argument"); return KR_EINVAL; }
    if (b->nr <= 0) { set_error("kr_reduce_emissivity: nr must be positive"); return KR_EINVAL; }
    const size_t words = (size_t) 5 * b->nr + 1;
    std::vector<double> h(words, 0.0);
    int rc = with_staged_rays((void*) rays, n, sizeof(kr_ray_f64), true, false, nullptr, [&](void* d) {
        DeviceBuffer hist;
        int r2 = hist.alloc(words * sizeof(double));
        if (r2 != KR_OK) return r2;
        KR_HIP(hipMemset(hist.p, 0, words * sizeof(double)));
        r2 = reduce_emissivity_dev(b, d, n, hist.p, nullptr);
        if (r2 != KR_OK) return r2;
        KR_HIP(hipMemcpy(h.data(), hist.p, words * sizeof(double), hipMemcpyDeviceToHost));
        return (int) KR_OK;
    });
    if (rc != KR_OK) return rc;
    const int nr = b->nr;
    for (int i = 0; i < nr; i++) {
        count[i] = (int64_t) h[i];
        flux[i] = h[nr + i];
        emis[i] = h[2 * nr + i];
        sum_redshift[i] = h[3 * nr + i];
        sum_time[i] = h[4 * nr + i];
    }
    if (disc_count) *disc_count = (int64_t) h[5 * nr];
    return KR_OK;
}

int kr_reduce_image_dev_f64(const kr_image_bins* b, const void* d, int64_t n, void* d_planes, void* st)
{
    if (!b || !d_planes) { set_error("kr_reduce_image: null argument"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : reduce_image_dev(b, d, n, d_planes, (hipStream_t) st);
}

int kr_reduce_image_f64(const kr_image_bins* b, const kr_ray_f64* rays, int64_t n, int32_t* nrays, double* flux, double* r,
                        double* phi, double* enshift, double* time, double* emis, int64_t* disc_count)
{
    if (!b || !nrays || !flux || !r || !phi || !enshift || !time || !emis) { set_error("kr_reduce_image: null argument"); return KR_EINVAL; }
    if (b->img_nx <= 0 || b->img_ny <= 0) { set_error("kr_reduce_image: image size must be positive"); return KR_EINVAL; }
    const size_t npix = (size_t) b->img_nx * b->img_ny;
    const size_t words = 7 * npix + 1;
    std::vector<double> h(words, 0.0);
    int rc = with_staged_rays((void*) rays, n, sizeof(kr_ray_f64), true, false, nullptr, [&](void* d) {
        DeviceBuffer planes;
        int r2 = planes.alloc(words * sizeof(double));
        if (r2 != KR_OK) return r2;
        KR_HIP(hipMemset(planes.p, 0, words * sizeof(double)));
        r2 = reduce_image_dev(b, d, n, planes.p, nullptr);
        if (r2 != KR_OK) return r2;
        KR_HIP(hipMemcpy(h.data(), planes.p, words * sizeof(double), hipMemcpyDeviceToHost));
        return (int) KR_OK;
    });
    if (rc != KR_OK) return rc;
    for (size_t i = 0; i < npix; i++) nrays[i] = (int32_t) h[i];
    std::memcpy(flux, &h[npix], npix * sizeof(double));
    std::memcpy(r, &h[2 * npix], npix * sizeof(double));
    std::memcpy(phi, &h[3 * npix], npix * sizeof(double));
    std::memcpy(enshift, &h[4 * npix], npix * sizeof(double));
    std::memcpy(time, &h[5 * npix], npix * sizeof(double));
    std::memcpy(emis, &h[6 * npix], npix * sizeof(double));
    if (disc_count) *disc_count = (int64_t) h[7 * npix];
    return KR_OK;
}

int kr_reduce_return_dev_f64(const kr_return_bins* b, const void* d, int64_t n, void* d_out4, void* st)
{
    if (!b || !d_out4) { set_error("kr_reduce_return: null argument"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : reduce_return_dev(b, d, n, d_out4, (hipStream_t) st);
}

int kr_post_return_dev_f64(double lo, double hi, const kr_return_bins* b, void* d, int64_t n, void* d_out4, void* st)
{
    if (!b || !d_out4) { set_error("kr_post_return: null argument"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : post_return_dev(lo, hi, b, d, n, d_out4, (hipStream_t) st);
}

int kr_post_return_batch_dev_f64(int32_t count, double lo, double hi, const kr_return_bins* b, void* const* d, const int64_t* n, void* const* d_out4, void* st)
{
    if (count < 0 || (count > 0 && (!b || !d || !n || !d_out4))) { set_error("kr_post_return_batch: null argument"); return KR_EINVAL; }
    for (int32_t i = 0; i < count; i++)
        if (n[i] > 0 && (!d[i] || !d_out4[i])) { set_error("kr_post_return_batch: null buffer"); return KR_EINVAL; }
    int rc = require_device();
    return rc != KR_OK ? rc : post_return_batch_dev(count, lo, hi, b, d, n, d_out4, (hipStream_t) st);
}

int kr_reduce_return_f64(const kr_return_bins* b, const kr_ray_f64* rays, int64_t n, double out[4])
{
    if (!b || !out) { set_error("kr_reduce_return: null argument"); return KR_EINVAL; }
    return with_staged_rays((void*) rays, n, sizeof(kr_ray_f64), true, false, nullptr, [&](void* d) {
        DeviceBuffer acc;
        int r2 = acc.alloc(4 * sizeof(double));
        if (r2 != KR_OK) return r2;
        KR_HIP(hipMemset(acc.p, 0, 4 * sizeof(double)));
        r2 = reduce_return_dev(b, d, n, acc.p, nullptr);
        if (r2 != KR_OK) return r2;
        KR_HIP(hipMemcpy(out, acc.p, 4 * sizeof(double), hipMemcpyDeviceToHost));
        return (int) KR_OK;
    });
}

// ---- diagnostics ---------------------------------------------------------------------------------------------
int kr_debug_arith_f64(int op, const double* a, const double* b, double* out, int64_t n)
{
    if (!a || !b || !out || n < 0) { set_error("kr_debug_arith: bad argument"); return KR_EINVAL; }
    int rc = require_device();
    if (rc != KR_OK) return rc;
    if (n == 0) return KR_OK;
    DeviceBuffer da, db, dout;
    const size_t bytes = (size_t) n * sizeof(double);
    if ((rc = da.alloc(bytes)) != KR_OK || (rc = db.alloc(bytes)) != KR_OK || (rc = dout.alloc(bytes)) != KR_OK) return rc;
    KR_HIP(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
    KR_HIP(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
    rc = arith_probe_dev(op, (const double*) da.p, (const double*) db.p, (double*) dout.p, n);
    if (rc != KR_OK) return rc;
    KR_HIP(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    return KR_OK;
}

// ---- attached host arrays (see with_staged_rays) --------------------------------------------------------------------------
int kr_host_attach(void* rays, int64_t n, int32_t ray_bytes)
{
    if (!rays || n <= 0 || (ray_bytes != (int32_t) sizeof(kr_ray_f64) && ray_bytes != (int32_t) sizeof(kr_ray_f32))) { set_error("kr_host_attach: bad argument"); return KR_EINVAL; }
    int rc = require_device();
    if (rc != KR_OK) return rc;
    Attached a;
    a.host = rays; a.n = n; a.ray_bytes = (size_t) ray_bytes;
    {
        std::lock_guard<std::mutex> lk(g_att_mu);
        if (g_attached.count(rays)) { set_error("kr_host_attach: array is already attached"); return KR_EINVAL; }
    }
    auto undo = [&]() { if (a.dev) (void) hipFree(a.dev); if (a.d_field) (void) hipFree(a.d_field); std::free(a.h_field); };
    hipError_t e = hipMalloc(&a.dev, (size_t) n * ray_bytes);
    if (e == hipSuccess) e = hipMalloc(&a.d_field, (size_t) n * 32);
    if (e == hipSuccess && !(a.h_field = std::malloc((size_t) n * 32))) e = hipErrorOutOfMemory;
    if (e != hipSuccess) { undo(); return hip_fail(e, "kr_host_attach allocation", __FILE__, __LINE__); }
    std::lock_guard<std::mutex> lk(g_att_mu);
    g_attached[rays] = a;
    return KR_OK;
}

int kr_host_detach(void* rays)
{
    Attached a;
    {
        std::lock_guard<std::mutex> lk(g_att_mu);
        auto it = g_attached.find(rays);
        if (it == g_attached.end()) return KR_OK;
        a = it->second;
        g_attached.erase(it);
    }
    (void) hipFree(a.dev);
    (void) hipFree(a.d_field);
    std::free(a.h_field);
    (void) hipGetLastError();
    return KR_OK;
}

// ---- memory helpers ------------------------------------------------------------------------------------------
int kr_malloc(void** d_ptr, int64_t bytes)
{
    if (!d_ptr || bytes < 0) { set_error("kr_malloc: bad argument"); return KR_EINVAL; }
    int rc = require_device();
    if (rc != KR_OK) return rc;
    KR_HIP(hipMalloc(d_ptr, (size_t) (bytes ? bytes : 1)));
    return KR_OK;
}
int kr_free(void* d_ptr)
{
    KR_HIP(hipFree(d_ptr));
    return KR_OK;
}
int kr_host_alloc(void** h_ptr, int64_t bytes)
{
    if (!h_ptr || bytes < 0) { set_error("kr_host_alloc: bad argument"); return KR_EINVAL; }
    int rc = require_device();
    if (rc != KR_OK) return rc;
    KR_HIP(hipHostMalloc(h_ptr, (size_t) (bytes ? bytes : 1), hipHostMallocDefault));
    return KR_OK;
}
int kr_host_free(void* h_ptr)
{
    KR_HIP(hipHostFree(h_ptr));
    return KR_OK;
}
int kr_memcpy_h2d(void* d_dst, const void* h_src, int64_t bytes)
{
    KR_HIP(hipMemcpy(d_dst, h_src, (size_t) bytes, hipMemcpyHostToDevice));
    return KR_OK;
}
int kr_memcpy_d2h(void* h_dst, const void* d_src, int64_t bytes)
{
    KR_HIP(hipMemcpy(h_dst, d_src, (size_t) bytes, hipMemcpyDeviceToHost));
    return KR_OK;
}
int kr_memset(void* d_ptr, int value, int64_t bytes)
{
    KR_HIP(hipMemset(d_ptr, value, (size_t) bytes));
    return KR_OK;
}
int kr_synchronize(void* stream)
{
    KR_HIP(hipStreamSynchronize((hipStream_t) stream));
    return KR_OK;
}
int kr_stream_create(void** stream)
{
    if (!stream) { set_error("kr_stream_create: null argument"); return KR_EINVAL; }
    int rc = require_device();
    if (rc != KR_OK) return rc;
    hipStream_t s = nullptr;
    KR_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void*) s;
    return KR_OK;
}
int kr_stream_destroy(void* stream)
{
    if (!stream) return KR_OK;
    KR_HIP(hipStreamSynchronize((hipStream_t) stream));
    side_stream_forget((hipStream_t) stream);
    KR_HIP(hipStreamDestroy((hipStream_t) stream));
    return KR_OK;
}
int kr_configure_process(void)
{
    if (g_runtime_touched) return 0;
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    return 1;
}
int kr_shutdown(void)
{
    if (!g_runtime_touched) return KR_OK;
    const int rc = trace_shutdown();       // (drains every device this library has used)
    source_tables_shutdown();
    return rc;
}

}  // extern "C"
