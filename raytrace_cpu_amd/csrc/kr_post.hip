// kr_post.hip -- the O(N) passes either side of the trace kernel, one ray per work-item:
//   ray sources     PointSource / ImagePlane ctors            (pointsource.cpp:11-64, imageplane.cpp:11-121)
//   redshift_start  emitted energy before the trace            (raytracer.cpp:342-417)
//   redshift        received/emitted energy ratio after it     (raytracer.cpp:420-600)
//   range_phi, calculate_momentum                               (raytracer.cpp:603-622, :704-753)
//   reducers        emissivity radial histogram, disc image    (emissivity.cpp:96-126, imageplane_disc_image.cpp:122-161)
// The reference runs these as serial host loops over rays[]; here they are streaming kernels over the same
// 144-B records (HBM-bound: 144 B read + <=16 B written per ray), so that at 1e7..1e8 rays the whole
// source -> trace -> redshift -> histogram pipeline can stay resident in HBM and only the histogram leaves.
// Histograms are accumulated per workgroup in LDS (ds_add_f64) and flushed with one global atomic per bin.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "kr_common.hpp"
#include "kr_device.hpp"
#include "kr_post_device.hpp"
#include "kr_crmath.hpp"

namespace kr {

namespace {

constexpr int kBlock = 256;

inline int grid_for(int64_t n, int cap_blocks = 256 * 16)
{
    const int64_t b = (n + kBlock - 1) / kBlock;
    return (int) std::max<int64_t>(1, std::min<int64_t>(b, cap_blocks));
}

template <typename R, typename T = typename ScalarOf<R>::type>
__global__ void __launch_bounds__(kBlock)
redshift_start_kernel(R* __restrict__ rays, long long n, T spin, T V, int reverse, int projradius)
{
    const T a = reverse ? -1 * spin : spin;
    if (V == -1) V = keplerian_V<T>(a, rays[0].r, rays[0].theta, projradius != 0);
    for (long long i = blockIdx.x * (long long) kBlock + threadIdx.x; i < n; i += (long long) gridDim.x * kBlock) {
        R* ray = &rays[i];
        R v;
        v.r = ray->r; v.theta = ray->theta; v.k = ray->k; v.h = ray->h; v.Q = ray->Q; v.rdot_sign = ray->rdot_sign; v.thetadot_sign = ray->thetadot_sign;
        ray->emit = emit_value(v, spin, a, V, reverse);
    }
}

template <typename R, typename T = typename ScalarOf<R>::type>
__global__ void __launch_bounds__(kBlock)
redshift_kernel(R* __restrict__ rays, long long n, T spin, T V_in, int reverse, int projradius, int motion)
{
    for (long long i = blockIdx.x * (long long) kBlock + threadIdx.x; i < n; i += (long long) gridDim.x * kBlock) {
        R* ray = &rays[i];
        R v;
        v.r = ray->r; v.theta = ray->theta; v.k = ray->k; v.h = ray->h; v.Q = ray->Q; v.rdot_sign = ray->rdot_sign; v.thetadot_sign = ray->thetadot_sign;
        v.emit = ray->emit;
        ray->redshift = redshift_value(v, spin, V_in, reverse, projradius, motion);
    }
}

// ---- redshift(RayDestination*, ...) with the default four_velocity (raytracer.cpp:450-477, :556-600;
//      ray_destination.h:59-78) -----------------------------------------------------------------------
template <typename R, typename T = typename ScalarOf<R>::type>
__global__ void __launch_bounds__(kBlock)
redshift_dest_kernel(R* __restrict__ rays, long long n, T spin, int reverse)
{
    for (long long i = blockIdx.x * (long long) kBlock + threadIdx.x; i < n; i += (long long) gridDim.x * kBlock) {
        R* ray = &rays[i];
        const T r = ray->r, theta = ray->theta;
        const Metric<T> m = kerr_metric<T>(r, theta, spin);
        const T V = 1 / (spin + r * kr_sqrt(r));
        const T gamma_factor = 1 / kr_sqrt(1 - (V - m.omega) * (V - m.omega) * m.e2psi / m.e2nu);
        const T et[4] = {gamma_factor / kr_sqrt(m.e2nu), 0, 0, gamma_factor * V / kr_sqrt(m.e2nu)};
        T p[4];
        momentum<T>(p[0], p[1], p[2], p[3], ray->k, ray->h, ray->Q, ray->rdot_sign, ray->thetadot_sign, r, theta, spin);
        if (reverse) { p[1] *= -1; p[2] *= -1; p[3] *= -1; }
        const T recv = energy_dot<T>(m, et, p);
        ray->redshift = reverse ? recv / ray->emit : ray->emit / recv;
    }
}

template <typename R, typename T = typename ScalarOf<R>::type>
__global__ void __launch_bounds__(kBlock) range_phi_kernel(R* __restrict__ rays, long long n, T lo, T hi)
{
    for (long long i = blockIdx.x * (long long) kBlock + threadIdx.x; i < n; i += (long long) gridDim.x * kBlock) {
        const T phi = rays[i].phi;
        const T wrapped = range_phi_value<T>(phi, rays[i].steps, lo, hi);
        if (!(wrapped == phi) && wrapped == wrapped) rays[i].phi = wrapped;
    }
}

// ---- calculate_momentum (raytracer.cpp:704-753) ---------------------------------------------------------
template <typename R, typename T = typename ScalarOf<R>::type>
__global__ void __launch_bounds__(kBlock) calculate_momentum_kernel(R* __restrict__ rays, long long n, T spin)
{
    for (long long i = blockIdx.x * (long long) kBlock + threadIdx.x; i < n; i += (long long) gridDim.x * kBlock) {
        R* ray = &rays[i];
        T pt, pr, ptheta, pphi;
        momentum<T>(pt, pr, ptheta, pphi, ray->k, ray->h, ray->Q, ray->rdot_sign, ray->thetadot_sign, ray->r, ray->theta, spin);
        ray->pt = pt; ray->pr = pr; ray->ptheta = ptheta; ray->pphi = pphi;
    }
}

__global__ void __launch_bounds__(kBlock)
pointsource_init_kernel(kr_ray_f64* __restrict__ rays, long long n, kr_pointsource s, SourceTables tb, int n_cosalpha, int n_beta, long long first, long long stride)
{
    const long long n_grid = (long long) n_cosalpha * n_beta;
    for (long long slot = blockIdx.x * (long long) kBlock + threadIdx.x; slot < n; slot += (long long) gridDim.x * kBlock)
        rays[slot] = pointsource_ray(s, tb, n_grid, n_beta, first + slot * stride);
}

// ---- fused prologue of the emissivity pipeline: PointSource ctor + redshift_start() in ONE pass (the record is written once,
//      with its `emit`); same per-ray functions as the two separate kernels, V == -1 resolved from record 0 as there ----------
__global__ void __launch_bounds__(kBlock)
pointsource_init_emit_kernel(kr_ray_f64* __restrict__ rays, long long n, kr_pointsource s, SourceTables tb, int n_cosalpha, int n_beta, long long first, long long stride,
                             double V, int reverse, int projradius)
{
    const long long n_grid = (long long) n_cosalpha * n_beta;
    const double a = reverse ? -1 * s.spin : s.spin;
    if (V == -1) {
        // the reference's loop overwrites V with the orbital velocity at rays[0] of the WHOLE source and keeps it for every ray
        // (raytracer.cpp:389-393): source ray 0, not this shard's first ray
        const kr_ray_f64 r0 = pointsource_ray(s, tb, n_grid, n_beta, 0);
        V = keplerian_V<double>(a, r0.r, r0.theta, projradius != 0);
    }
    for (long long slot = blockIdx.x * (long long) kBlock + threadIdx.x; slot < n; slot += (long long) gridDim.x * kBlock) {
        kr_ray_f64 ray = pointsource_ray(s, tb, n_grid, n_beta, first + slot * stride);
        ray.emit = emit_value(ray, s.spin, a, V, reverse);
        rays[slot] = ray;
    }
}

// The same for SEVERAL sources in one launch (kr_pointsource_init_emit_batch_dev_f64: the multi-radius drivers).  A hundred launches of 1e6 rays
// each cannot fill the GPU one at a time (0.093 ms per launch = 1.5 TB/s of stores against ~4 TB/s for one large launch, rocprofv3 r03); the
// items of a chunk ride in the kernel arguments (no staging buffer whose lifetime a later call would have to track), blockIdx.y picks the item.
struct SourceItem {
    kr_pointsource s;
    SourceTables tb;
    double V;
    kr_ray_f64* rays;
    long long n;
    int n_cosalpha, n_beta;
};
constexpr int kSourceChunk = 19;                       // 19 x 200 B of the 4 KB a kernel's arguments may take
struct SourceChunk { SourceItem item[kSourceChunk]; };

__global__ void __launch_bounds__(kBlock)
pointsource_init_emit_multi_kernel(SourceChunk c, int reverse, int projradius)
{
    const SourceItem& it = c.item[blockIdx.y];
    const kr_pointsource s = it.s;
    const SourceTables tb = it.tb;
    const long long n = it.n, n_grid = (long long) it.n_cosalpha * it.n_beta;
    const int n_beta = it.n_beta;
    kr_ray_f64* __restrict__ rays = it.rays;
    const double a = reverse ? -1 * s.spin : s.spin;
    double V = it.V;
    if (V == -1) {
        const kr_ray_f64 r0 = pointsource_ray(s, tb, n_grid, n_beta, 0);
        V = keplerian_V<double>(a, r0.r, r0.theta, projradius != 0);
    }
    for (long long slot = blockIdx.x * (long long) kBlock + threadIdx.x; slot < n; slot += (long long) gridDim.x * kBlock) {
        kr_ray_f64 ray = pointsource_ray(s, tb, n_grid, n_beta, slot);
        ray.emit = emit_value(ray, s.spin, a, V, reverse);
        rays[slot] = ray;
    }
}

// ---- ImagePlane ctor + init_image_plane (imageplane.cpp:11-121) ---------------------------------------------
// sin / cos of the inclination come from the host's C library (one angle per plane: PlaneTrig); the per-ray acos, atan2, asin and tan -- N^2 distinct
// arguments, nothing to tabulate -- are kr_crmath.hpp's correctly rounded routines, sin / cos kr_sincos.hpp's: a device-built ray then differs from
// the reference constructor's only where the host library itself is not correctly rounded (~1e-3 of the rays in some last bit; the device library's
// 1-2 ulp routines left 21-25 % of the rays with another phi and 4-7 % with another theta or Q).
struct PlaneTrig { double sin_incl, cos_incl; };

KR_DEV kr_ray_f64 imageplane_ray(const kr_imageplane& s, const PlaneTrig& pt_, long long n_grid, int Ny, double a, double D, double phi0, long long ix)
{
    kr_ray_f64 ray;
        memset(&ray, 0, sizeof(ray));
        ray.steps = -1;
        if (ix < n_grid) {
            const int i = (int) (ix / Ny), j = (int) (ix % Ny);
            const double x = s.x0 + i * s.dy;            // sic: dy, imageplane.cpp:43
            const double y = s.y0 + j * s.dy;
            const double si = pt_.sin_incl, ci = pt_.cos_incl;

            const double r = kr_sqrt(D * D + x * x + y * y);
            const double theta = krcr::kr_acos_cr((D * ci + y * si) / r);
            const double phi = phi0 + krcr::kr_atan2_cr(x, D * si - y * ci);

            const double pr = D / r;
            const double ptheta = kr_sin(krcr::kr_acos_cr(D / r)) / r;
            const double pphi = x * si / (x * x + (D * si - y * ci) * (D * si - y * ci));

            const double st = kr_sin(theta), ct = kr_cos(theta);
            const double rhosq = r * r + (a * ct) * (a * ct);
            const double delta = r * r - 2 * r + a * a;
            const double sigmasq = (r * r + a * a) * (r * r + a * a) - a * a * delta * st * st;
            const double e2nu = rhosq * delta / sigmasq;
            const double e2psi = sigmasq * st * st / rhosq;
            const double omega = 2 * a * r / sigmasq;
            const double g00 = e2nu - omega * omega * e2psi, g03 = omega * e2psi, g11 = -rhosq / delta, g22 = -rhosq, g33 = -e2psi;

            const double A = g00, B = 2 * g03 * pphi;
            const double Cq = g11 * pr * pr + g22 * ptheta * ptheta + g33 * pphi * pphi;
            double pt = (-B + kr_sqrt(B * B - 4 * A * Cq)) / (2 * A);
            if (pt < 0) pt = (-B - kr_sqrt(B * B - 4 * A * Cq)) / (2 * A);

            ray.t = 0; ray.r = r; ray.theta = theta; ray.phi = phi;
            ray.pt = pt; ray.pr = pr; ray.ptheta = ptheta; ray.pphi = pphi;
            ray.rdot_sign = -1;
            ray.k = 1;                                   // calculate_constants_from_p's k/h/Q are overwritten, :100-113

            const double b = kr_sqrt(x * x + y * y);
            double beta = krcr::kr_asin_cr(y / b);
            if (x < 0) beta = kPi - beta;
            const double h = -1. * b * si * kr_cos(beta);
            const double ltheta = b * kr_sin(beta);
            const double tt = krcr::kr_tan_cr(theta);
            ray.h = h;
            ray.Q = (ltheta * ltheta) - (a * ct) * (a * ct) + ((h / tt)) * ((h / tt));
            ray.thetadot_sign = (ltheta >= 0) ? 1 : -1;
            ray.steps = 0;
            ray.alpha = x;
            ray.beta = y;
        }
    return ray;
}

__global__ void __launch_bounds__(kBlock)
imageplane_init_kernel(kr_ray_f64* __restrict__ rays, long long n, kr_imageplane s, PlaneTrig tr, int Nx, int Ny, long long first, long long stride)
{
    const long long n_grid = (long long) Nx * Ny;
    const double a = -1 * s.spin;                       // imageplane.cpp:12
    const double D = s.dist, phi0 = s.phi0;
    for (long long slot = blockIdx.x * (long long) kBlock + threadIdx.x; slot < n; slot += (long long) gridDim.x * kBlock)
        rays[slot] = imageplane_ray(s, tr, n_grid, Ny, a, D, phi0, first + slot * stride);
}

// ---- fused prologue of the image pipeline: ImagePlane ctor + redshift_start(V, reverse, projradius) in one pass; `spin` is the
//      Raytracer member (the ImagePlane has negated it), as for kr_redshift_start_dev_f64 -----------------------------------
__global__ void __launch_bounds__(kBlock)
imageplane_init_emit_kernel(kr_ray_f64* __restrict__ rays, long long n, kr_imageplane s, PlaneTrig tr, int Nx, int Ny, long long first, long long stride, long long run,
                            double spin, double V, int reverse, int projradius)
{
    const long long n_grid = (long long) Nx * Ny;
    const double a = -1 * s.spin;
    const double D = s.dist, phi0 = s.phi0;
    const double am = reverse ? -1 * spin : spin;
    if (V == -1) {
        const kr_ray_f64 r0 = imageplane_ray(s, tr, n_grid, Ny, a, D, phi0, 0);      // source ray 0 (raytracer.cpp:389-393), not the shard's first
        V = keplerian_V<double>(am, r0.r, r0.theta, projradius != 0);
    }
    for (long long slot = blockIdx.x * (long long) kBlock + threadIdx.x; slot < n; slot += (long long) gridDim.x * kBlock) {
        // slot -> source ray: runs of `run` consecutive rays, `stride` apart (run = 1: plain ray-cyclic)
        const long long src = (run == 1) ? first + slot * stride : first + (slot / run) * stride + (slot % run);
        kr_ray_f64 ray = imageplane_ray(s, tr, n_grid, Ny, a, D, phi0, src);
        ray.emit = emit_value(ray, spin, am, V, reverse);
        rays[slot] = ray;
    }
}


// ---- emissivity reducer (emissivity.cpp:96-126) -----------------------------------------------------------
template <bool USE_LDS>
__global__ void __launch_bounds__(kBlock)
reduce_emissivity_kernel(const kr_ray_f64* __restrict__ rays, long long n, kr_emis_bins b, double* __restrict__ hist)
{
    __shared__ double lds[USE_LDS ? (5 * kMaxLdsBins + 1) : 1];
    const int nr = b.nr;
    const int words = 5 * nr + 1;
    if (USE_LDS) {
        for (int w = threadIdx.x; w < words; w += kBlock) lds[w] = 0;
        __syncthreads();
    }
    double* acc = USE_LDS ? lds : hist;
    const double log_dr = kr_log(b.dr);
    for (long long i = blockIdx.x * (long long) kBlock + threadIdx.x; i < n; i += (long long) gridDim.x * kBlock) {
        const kr_ray_f64* ray = &rays[i];
        emissivity_accumulate(acc, b, log_dr, ray->steps, ray->r, ray->theta, ray->redshift, ray->t);
    }
    if (USE_LDS) {
        __syncthreads();
        for (int w = threadIdx.x; w < words; w += kBlock)
            if (lds[w] != 0) atomicAdd(&hist[w], lds[w]);
    }
}

// ---- fused epilogue of the emissivity pipeline: range_phi -> redshift -> histogram in ONE pass over the records
//      (emissivity.cpp:93-126: the three O(N) steps after run_raytrace).  Same per-ray functions as the separate
//      kernels, so rays[] and the integer bin counts are bit-identical to running them one after another. -------------
template <bool USE_LDS>
__global__ void __launch_bounds__(kBlock)
post_emissivity_kernel(kr_ray_f64* __restrict__ rays, long long n, double spin, double V, int reverse, int projradius, int motion, double lo, double hi,
                       kr_emis_bins b, double* __restrict__ hist)
{
    __shared__ double lds[USE_LDS ? (5 * kMaxLdsBins + 1) : 1];
    const int nr = b.nr;
    const int words = 5 * nr + 1;
    if (USE_LDS) {
        for (int w = threadIdx.x; w < words; w += kBlock) lds[w] = 0;
        __syncthreads();
    }
    double* acc = USE_LDS ? lds : hist;
    const double log_dr = kr_log(b.dr);
    for (long long i = blockIdx.x * (long long) kBlock + threadIdx.x; i < n; i += (long long) gridDim.x * kBlock) {
        kr_ray_f64* ray = &rays[i];
        kr_ray_f64 v;
        v.r = ray->r; v.theta = ray->theta; v.k = ray->k; v.h = ray->h; v.Q = ray->Q; v.rdot_sign = ray->rdot_sign; v.thetadot_sign = ray->thetadot_sign;
        v.emit = ray->emit;
        const int steps = ray->steps;
        const double phi = ray->phi;
        const double wrapped = range_phi_value<double>(phi, steps, lo, hi);
        if (!(wrapped == phi) && wrapped == wrapped) ray->phi = wrapped;
        const double g = redshift_value(v, spin, V, reverse, projradius, motion);
        ray->redshift = g;
        emissivity_accumulate(acc, b, log_dr, steps, v.r, v.theta, g, ray->t);
    }
    if (USE_LDS) {
        __syncthreads();
        for (int w = threadIdx.x; w < words; w += kBlock)
            if (lds[w] != 0) atomicAdd(&hist[w], lds[w]);
    }
}

// ---- image reducer (imageplane_disc_image.cpp:20-28, :122-161) -------------------------------------------------
KR_DEV double powerlaw3(double r, double q1, double rb1, double q2, double rb2, double q3)
{
    if (r < rb1) return kr_pow(r, -1 * q1);
    else if (r < rb2) return kr_pow(rb1, q2 - q1) * kr_pow(r, -1 * q2);
    else return kr_pow(rb1, q2 - q1) * kr_pow(rb2, q3 - q2) * kr_pow(r, -1 * q3);
}

// d_planes layout: [nrays | flux | r | phi | enshift | time | emis](npix each) + disc_count(1), doubles.
// one ray's contribution to the seven image planes (imageplane_disc_image.cpp:122-161); returns 1 if it was counted
KR_DEV unsigned image_accumulate(double* planes, long long npix, const kr_image_bins& b, int steps, double r, double theta, double phi, double t, double g,
                                 double alpha, double beta)
{
    if (!(steps > 0)) return 0;
    const double z = r * kr_cos(theta);
    if (!(z < 1E-2 && r >= b.r_isco && r < b.r_disc && g > 0)) return 0;
    int ix = (int) ((alpha - b.x0) / b.img_dx);
    int iy = (int) ((beta - b.y0) / b.img_dy);
    if (b.flip_image) iy = b.img_ny - iy - 1;
    if (!(ix >= 0 && ix < b.img_nx && iy >= 0 && iy < b.img_ny)) return 0;
    const long long px = (long long) ix * b.img_ny + iy;
    const double e = powerlaw3(r, b.q1, b.rb1, b.q2, b.rb2, b.q3);
    atomicAdd(&planes[px], 1.0);
    atomicAdd(&planes[npix + px], e / kr_pow(g, 3.0));
    atomicAdd(&planes[2 * npix + px], r);
    atomicAdd(&planes[3 * npix + px], phi);
    atomicAdd(&planes[4 * npix + px], 1. / g);
    atomicAdd(&planes[5 * npix + px], t);
    atomicAdd(&planes[6 * npix + px], e);
    return 1;
}

// A ray lands in one pixel and the ray grid is about the pixel grid, so contention is low: global f64 atomics.
__global__ void __launch_bounds__(kBlock)
reduce_image_kernel(const kr_ray_f64* __restrict__ rays, long long n, kr_image_bins b, double* __restrict__ planes)
{
    const long long npix = (long long) b.img_nx * b.img_ny;
    unsigned long long hits = 0;
    for (long long i = blockIdx.x * (long long) kBlock + threadIdx.x; i < n; i += (long long) gridDim.x * kBlock) {
        const kr_ray_f64* ray = &rays[i];
        hits += image_accumulate(planes, npix, b, ray->steps, ray->r, ray->theta, ray->phi, ray->t, ray->redshift, ray->alpha, ray->beta);
    }
    if (hits) atomicAdd(&planes[7 * npix], (double) hits);
}

// ---- fused epilogue of the image pipeline: redshift(V, reverse, projradius, motion) + range_phi + the seven planes in one pass
//      (imageplane_disc_image.cpp:117-161) -----------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
post_image_kernel(kr_ray_f64* __restrict__ rays, long long n, double spin, double V, int reverse, int projradius, int motion, double lo, double hi,
                  kr_image_bins b, double* __restrict__ planes)
{
    const long long npix = (long long) b.img_nx * b.img_ny;
    unsigned long long hits = 0;
    for (long long i = blockIdx.x * (long long) kBlock + threadIdx.x; i < n; i += (long long) gridDim.x * kBlock) {
        kr_ray_f64* ray = &rays[i];
        kr_ray_f64 v;
        v.r = ray->r; v.theta = ray->theta; v.k = ray->k; v.h = ray->h; v.Q = ray->Q; v.rdot_sign = ray->rdot_sign; v.thetadot_sign = ray->thetadot_sign;
        v.emit = ray->emit;
        const int steps = ray->steps;
        const double g = redshift_value(v, spin, V, reverse, projradius, motion);
        ray->redshift = g;
        const double phi = ray->phi;
        const double wrapped = range_phi_value<double>(phi, steps, lo, hi);
        if (!(wrapped == phi) && wrapped == wrapped) ray->phi = wrapped;
        hits += image_accumulate(planes, npix, b, steps, v.r, v.theta, wrapped, ray->t, g, ray->alpha, ray->beta);
    }
    if (hits) atomicAdd(&planes[7 * npix], (double) hits);
}

}  // namespace

// ---- returning-radiation classification (disc_source_photonfrac_r.cpp:97-126) ------------------------------------
// out4 = {ray_count, return, escape, lost}; per-wave shuffle reduction, one atomic per wave and word
// FUSED: range_phi(lo, hi) first, in the same pass over the records (kr_post_return_dev_f64)
template <bool FUSED>
KR_DEV void reduce_return_body(kr_ray_f64* __restrict__ rays, long long n, const kr_return_bins& b, double* __restrict__ out4, double lo, double hi)
{
    double acc[4] = {0, 0, 0, 0};
    for (long long i = blockIdx.x * (long long) kBlock + threadIdx.x; i < n; i += (long long) gridDim.x * kBlock) {
        kr_ray_f64* ray = &rays[i];
        double phi = ray->phi;
        if (FUSED) {
            const double wrapped = range_phi_value<double>(phi, ray->steps, lo, hi);
            if (!(wrapped == phi) && wrapped == wrapped) { ray->phi = wrapped; phi = wrapped; }
        }
        if (!(ray->steps > 0)) continue;
        const double alpha = krcr::kr_acos_cr(ray->alpha);  // rays[].alpha holds cos(alpha)
        const double sasb = kr_abs(kr_sin(alpha) * kr_sin(ray->beta));
        double w = b.plane_iso ? sasb : 1;
        if (b.limb) w *= 1 + 2.06 * sasb;
        acc[0] += b.weight_norm ? w : 1;
        const double r = ray->r;
        if (ray->theta >= kPi2 && r >= b.r_isco && r < b.r_disc) {
            if (kr_abs(r - b.source_r) > 0.1 * b.source_r || kr_abs(phi - b.source_phi) > 0.1) acc[1] += w;
        } else if (r > b.r_esc) {
            acc[2] += w;
        } else if (r < b.r_isco) {
            acc[3] += w;
        }
    }
    // wave shuffle -> workgroup (LDS) -> ONE atomic per word and workgroup.  (One per wave was 15 600 atomics on the same four words for 1e6
    // rays: 0.16 ms of serialised L2 atomics around 0.03 ms of streaming -- 16 ms of the 253-ms returning-radiation pass, rocprofv3 r03.)
    __shared__ double part[kBlock / 64][4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double v = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; w++) v += part[w][threadIdx.x];
        if (v != 0) atomicAdd(&out4[threadIdx.x], v);
    }
}

template <bool FUSED>
__global__ void __launch_bounds__(kBlock)
reduce_return_kernel(kr_ray_f64* __restrict__ rays, long long n, kr_return_bins b, double* __restrict__ out4, double lo, double hi)
{
    reduce_return_body<FUSED>(rays, n, b, out4, lo, hi);
}

// several launches' reductions in one (kr_post_return_batch_dev_f64); items in the kernel arguments, blockIdx.y picks the item
struct ReturnItem {
    kr_return_bins b;
    kr_ray_f64* rays;
    long long n;
    double* out4;
};
constexpr int kReturnChunk = 32;
struct ReturnChunk { ReturnItem item[kReturnChunk]; };

__global__ void __launch_bounds__(kBlock)
reduce_return_multi_kernel(ReturnChunk c, double lo, double hi)
{
    const ReturnItem& it = c.item[blockIdx.y];
    const kr_return_bins b = it.b;
    reduce_return_body<true>(it.rays, it.n, b, it.out4, lo, hi);
}

// ---- diagnostics: the device arithmetic primitives the trace kernel is built from, exposed one at a time so that a
//      test can compare them with the host's IEEE results (tests/test_gpu_primitives.py) ------------------------------
__global__ void __launch_bounds__(kBlock) arith_probe_kernel(int op, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, long long n)
{
    for (long long i = blockIdx.x * (long long) kBlock + threadIdx.x; i < n; i += (long long) gridDim.x * kBlock) {
        const double x = a[i], y = b[i];
        double r = 0, s, c;
        switch (op) {
            case 0: r = x / y; break;                       // the compiler's IEEE division
            case 1: r = lean_div(x, y); break;                // the lean chain used on the strict path
            case 2: r = __builtin_sqrt(x); break;           // the compiler's IEEE sqrt
            case 3: r = lean_sqrt(x); break;
            case 4: kr_sincos_f64(x, s, c); r = s; break;
            case 5: kr_sincos_f64(x, s, c); r = c; break;
            case 6: r = x * fast_rcp(y); break;             // fast-math path
            case 7: r = fast_sqrt(x); break;
            case 8: ::sincos(x, &s, &c); r = s; break;      // device libm
            case 9: ::sincos(x, &s, &c); r = c; break;
            case 10: r = ::pow(x, y); break;
            case 11: { double v = __builtin_amdgcn_rcp(y); r = x * v; } break;                                             // raw v_rcp_f64
            case 12: { double v = __builtin_amdgcn_rcp(y); v = __builtin_fma(__builtin_fma(-y, v, 1.0), v, v); r = x * v; } break;   // + one Newton step
            case 13: { double v = __builtin_amdgcn_rsq(x); r = x * v; } break;                                             // raw v_rsq_f64
            case 14: { double v = __builtin_amdgcn_rsq(x); double g = x * v, h = 0.5 * v; const double e = __builtin_fma(-h, g, 0.5);
                       r = __builtin_fma(g, e, g); } break;                                                                  // + one coupled step
            case 18: r = fifth_root_for_controller(x); break;                        // x^(1/5) of the RK45 step controller
            case 19: r = kr_replay_additions(x, y, 20000); break;                    // 20 000 additions of y to x in closed form (kr_replay.hpp)
            case 17: r = div_by_uniform(x, y, 1.0 / y, true); break;                 // quotient by a launch-uniform divisor (IEEE reciprocal)
            case 20: { const StageRecips<double> q(y, x, (x * x) * y, 2.0, 1.0); r = q.over_sin2_rhosq_delta(1.2345678901234567); } break;   // x = sin, y = rho^2 Delta: the reciprocals
            case 21: { const StageRecips<double> q(1.0, 1.0, 1.0, x, 1.0); r = q.over_rho4(y); } break;                                       // x = rho^2:               put together
            case 15: kr_sincos_fast_f64(x, s, c); r = s; break;
            case 16: kr_sincos_fast_f64(x, s, c); r = c; break;
        }
        out[i] = r;
    }
}

// ---- launchers (device pointers) ---------------------------------------------------------------------------
#define KR_LAUNCH_CHECK() KR_HIP(hipGetLastError())

int arith_probe_dev(int op, const double* a, const double* b, double* out, int64_t n)
{
    if (n <= 0) return KR_OK;
    hipLaunchKernelGGL(arith_probe_kernel, dim3(grid_for(n)), dim3(kBlock), 0, nullptr, op, a, b, out, (long long) n);
    KR_LAUNCH_CHECK();
    return KR_OK;
}

// f32 != 0: the records are kr_ray_f32 and the pass computes in float (the scalars are float values carried in doubles)
#define KR_POST_LAUNCH(kernel, ...)                                                                                      \
    do {                                                                                                                 \
        if (n <= 0) return KR_OK;                                                                                        \
        if (f32) hipLaunchKernelGGL((kernel<kr_ray_f32, float>), dim3(grid_for(n)), dim3(kBlock), 0, st, (kr_ray_f32*) d, (long long) n, __VA_ARGS__); \
        else hipLaunchKernelGGL((kernel<kr_ray_f64, double>), dim3(grid_for(n)), dim3(kBlock), 0, st, (kr_ray_f64*) d, (long long) n, __VA_ARGS__);   \
        KR_LAUNCH_CHECK();                                                                                               \
        return KR_OK;                                                                                                    \
    } while (0)

int redshift_start_dev(double spin, double V, int reverse, int projradius, void* d, int64_t n, hipStream_t st, bool f32)
{
    KR_POST_LAUNCH(redshift_start_kernel, spin, V, reverse, projradius);
}

int redshift_dev(double spin, double V, int reverse, int projradius, int motion, void* d, int64_t n, hipStream_t st, bool f32)
{
    KR_POST_LAUNCH(redshift_kernel, spin, V, reverse, projradius, motion);
}

int redshift_dest_dev(double spin, int reverse, void* d, int64_t n, hipStream_t st, bool f32)
{
    KR_POST_LAUNCH(redshift_dest_kernel, spin, reverse);
}

int range_phi_dev(double lo, double hi, void* d, int64_t n, hipStream_t st, bool f32)
{
    KR_POST_LAUNCH(range_phi_kernel, lo, hi);
}

int calculate_momentum_dev(double spin, void* d, int64_t n, hipStream_t st, bool f32)
{
    KR_POST_LAUNCH(calculate_momentum_kernel, spin);
}

// ---- the PointSource constructor's transcendental values, from the HOST's C library (SourceTables, kr_post_device.hpp) -------------------------
// One device array of (sin, cos) pairs per distinct (device, kind, first angle / cosine, spacing, count); built on the first call that needs
// it (a hipMalloc and a blocking 50-KB copy: that one call waits for the copy, no later one does), kept until kr_shutdown.  The hundred sources
// of a multi-radius driver share one pair of arrays.  A process that keeps inventing new grids fills the cache (kMaxAngleTables): the device is
// then drained and the arrays are released together, since a kernel in flight may still be reading one.
namespace {
struct AngleKey {
    int dev, kind;           // kind 0: x is cos(alpha) -> (sin, cos) of acos(x);  1: x is beta -> (sin, cos) of x
    double x0, dx;
    int n;
    bool operator<(const AngleKey& o) const { return std::tie(dev, kind, x0, dx, n) < std::tie(o.dev, o.kind, o.x0, o.dx, o.n); }
};
std::mutex g_tables_mu;
std::map<AngleKey, double2*> g_tables;
constexpr size_t kMaxAngleTables = 256;

}  // namespace

// (sin, cos) pairs of the n angles of one grid axis, with the HOST's C library -- the one the reference's constructor calls.  sincos(), not sin()
// and cos(): an optimising build of the reference (g++ -O2, the build the oracle is pinned to: 40 calls of sincos in oracle/_ref/libkr_ref.so) merges
// sin(x) and cos(x) of one argument into ONE sincos(x) call, and glibc's sincos differs from its sin / cos in the last bit of one result on 0.13 % of
// arguments -- 6 of the 6324 table entries of the BASELINE grid, one of which moved k of one ray in 1e7 by an ulp.
void angle_values(int kind, double x0, double dx, int n, double* sincos_pairs)
{
    for (int i = 0; i < n; i++) {
        const double x = x0 + i * dx;                               // the reference's own expression (pointsource.cpp:38-39): int -> double, one product, one sum
        const double ang = kind == 0 ? std::acos(x) : x;            // pointsource.cpp:46
        ::sincos(ang, &sincos_pairs[2 * i], &sincos_pairs[2 * i + 1]);     // raytracer.cpp:653
    }
}

namespace {
int angle_table(int kind, double x0, double dx, int n, const double2** out)
{
    int dev = 0;
    KR_HIP(hipGetDevice(&dev));
    const AngleKey key{dev, kind, x0, dx, n};
    std::lock_guard<std::mutex> lk(g_tables_mu);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) { *out = it->second; return KR_OK; }
    if (g_tables.size() >= kMaxAngleTables) {
        KR_HIP(hipDeviceSynchronize());
        for (auto& kv : g_tables) (void) hipFree(kv.second);
        g_tables.clear();
    }
    std::vector<double2> h((size_t) std::max(n, 1));
    angle_values(kind, x0, dx, n, reinterpret_cast<double*>(h.data()));
    double2* d = nullptr;
    KR_HIP(hipMalloc((void**) &d, h.size() * sizeof(double2)));
    const hipError_t e = hipMemcpy(d, h.data(), h.size() * sizeof(double2), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void) hipFree(d); return kr::hip_fail(e, "hipMemcpy(angle table)", __FILE__, __LINE__); }
    g_tables.emplace(key, d);
    *out = d;
    return KR_OK;
}

int source_tables(const kr_pointsource* s, int n_cosalpha, int n_beta, SourceTables* tb)
{
    int rc = angle_table(0, s->cosalpha0, s->dcosalpha, n_cosalpha, &tb->alpha_sc);
    if (rc == KR_OK) rc = angle_table(1, s->beta0, s->dbeta, n_beta, &tb->beta_sc);
    ::sincos(s->pos[2], &tb->sin_th, &tb->cos_th);                  // raytracer.cpp:631-672 (calculate_constants; see angle_values)
    tb->tan_th = std::tan(s->pos[2]);
    return rc;
}
}  // namespace

void source_tables_shutdown()
{
    std::lock_guard<std::mutex> lk(g_tables_mu);
    for (auto& kv : g_tables) {
        if (hipSetDevice(kv.first.dev) == hipSuccess) (void) hipFree(kv.second);
        else (void) hipGetLastError();
    }
    g_tables.clear();
}

int pointsource_init_dev(const kr_pointsource* s, void* d, int64_t n, int64_t first, int64_t stride, hipStream_t st)
{
    int32_t nc = 0, nb = 0;
    const int64_t total = kr_pointsource_count(s, &nc, &nb);
    if (first < 0 || stride < 1) { set_error("kr_pointsource_init: bad first/stride"); return KR_EINVAL; }
    if (first == 0 && stride == 1 && n < total) { set_error("kr_pointsource_init: n smaller than kr_pointsource_count()"); return KR_EINVAL; }
    if (n <= 0) return KR_OK;
    SourceTables tb;
    const int rc = source_tables(s, nc, nb, &tb);
    if (rc != KR_OK) return rc;
    hipLaunchKernelGGL(pointsource_init_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, (kr_ray_f64*) d, (long long) n, *s, tb, nc, nb, (long long) first, (long long) stride);
    KR_LAUNCH_CHECK();
    return KR_OK;
}

int pointsource_init_emit_dev(const kr_pointsource* s, void* d, int64_t n, int64_t first, int64_t stride, double V, int reverse, int projradius, hipStream_t st)
{
    int32_t nc = 0, nb = 0;
    kr_pointsource_count(s, &nc, &nb);
    if (first < 0 || stride < 1) { set_error("kr_pointsource_init_emit: bad first/stride"); return KR_EINVAL; }
    if (n <= 0) return KR_OK;
    SourceTables tb;
    const int rc = source_tables(s, nc, nb, &tb);
    if (rc != KR_OK) return rc;
    hipLaunchKernelGGL(pointsource_init_emit_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, (kr_ray_f64*) d, (long long) n, *s, tb, nc, nb, (long long) first,
                       (long long) stride, V, reverse, projradius);
    KR_LAUNCH_CHECK();
    return KR_OK;
}

// sin / cos of the plane's inclination with the host's C library (sincos(): see angle_values) at the reference's own argument, incl * M_PI / 180
// (imageplane.cpp:23)
static PlaneTrig plane_trig(const kr_imageplane* s)
{
    PlaneTrig t;
    ::sincos(s->inc_deg * M_PI / 180, &t.sin_incl, &t.cos_incl);
    return t;
}

int imageplane_init_dev(const kr_imageplane* s, void* d, int64_t n, int64_t first, int64_t stride, hipStream_t st)
{
    int32_t nx = 0, ny = 0;
    const int64_t total = kr_imageplane_count(s, &nx, &ny);
    if (first < 0 || stride < 1) { set_error("kr_imageplane_init: bad first/stride"); return KR_EINVAL; }
    if (first == 0 && stride == 1 && n < total) { set_error("kr_imageplane_init: n smaller than kr_imageplane_count()"); return KR_EINVAL; }
    if (n <= 0) return KR_OK;
    hipLaunchKernelGGL(imageplane_init_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, (kr_ray_f64*) d, (long long) n, *s, plane_trig(s), nx, ny, (long long) first, (long long) stride);
    KR_LAUNCH_CHECK();
    return KR_OK;
}

int imageplane_init_emit_dev(const kr_imageplane* s, void* d, int64_t n, int64_t first, int64_t stride, int64_t run, double spin, double V, int reverse,
                             int projradius, hipStream_t st)
{
    int32_t nx = 0, ny = 0;
    kr_imageplane_count(s, &nx, &ny);
    if (first < 0 || stride < 1 || run < 1 || run > stride) { set_error("kr_imageplane_init_emit: bad first/stride/run"); return KR_EINVAL; }
    if (n <= 0) return KR_OK;
    hipLaunchKernelGGL(imageplane_init_emit_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, (kr_ray_f64*) d, (long long) n, *s, plane_trig(s), nx, ny, (long long) first,
                       (long long) stride, (long long) run, spin, V, reverse, projradius);
    KR_LAUNCH_CHECK();
    return KR_OK;
}

int post_image_dev(double spin, double V, int reverse, int projradius, int motion, double lo, double hi, const kr_image_bins* b, void* d, int64_t n,
                   void* d_planes, hipStream_t st)
{
    if (b->img_nx <= 0 || b->img_ny <= 0) { set_error("kr_post_image: image size must be positive"); return KR_EINVAL; }
    if (n <= 0) return KR_OK;
    hipLaunchKernelGGL(post_image_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, (kr_ray_f64*) d, (long long) n, spin, V, reverse, projradius, motion, lo, hi, *b,
                       (double*) d_planes);
    KR_LAUNCH_CHECK();
    return KR_OK;
}

int reduce_emissivity_dev(const kr_emis_bins* b, const void* d, int64_t n, void* d_hist, hipStream_t st)
{
    if (b->nr <= 0) { set_error("kr_reduce_emissivity: nr must be positive"); return KR_EINVAL; }
    if (n <= 0) return KR_OK;
    // few, fat workgroups: every workgroup flushes 5*nr+1 atomics, so keep the flush traffic below the ray traffic
    const int grid = grid_for(n, 256 * 4);
    if (b->nr <= kMaxLdsBins)
        hipLaunchKernelGGL(reduce_emissivity_kernel<true>, dim3(grid), dim3(kBlock), 0, st, (const kr_ray_f64*) d, (long long) n, *b, (double*) d_hist);
    else
        hipLaunchKernelGGL(reduce_emissivity_kernel<false>, dim3(grid), dim3(kBlock), 0, st, (const kr_ray_f64*) d, (long long) n, *b, (double*) d_hist);
    KR_LAUNCH_CHECK();
    return KR_OK;
}

int post_emissivity_dev(double spin, double V, int reverse, int projradius, int motion, double lo, double hi, const kr_emis_bins* b, void* d, int64_t n,
                        void* d_hist, hipStream_t st)
{
    if (b->nr <= 0) { set_error("kr_post_emissivity: nr must be positive"); return KR_EINVAL; }
    if (n <= 0) return KR_OK;
    const int grid = grid_for(n, 256 * 4);
    if (b->nr <= kMaxLdsBins)
        hipLaunchKernelGGL(post_emissivity_kernel<true>, dim3(grid), dim3(kBlock), 0, st, (kr_ray_f64*) d, (long long) n, spin, V, reverse, projradius, motion, lo, hi, *b,
                           (double*) d_hist);
    else
        hipLaunchKernelGGL(post_emissivity_kernel<false>, dim3(grid), dim3(kBlock), 0, st, (kr_ray_f64*) d, (long long) n, spin, V, reverse, projradius, motion, lo, hi, *b,
                           (double*) d_hist);
    KR_LAUNCH_CHECK();
    return KR_OK;
}

int reduce_return_dev(const kr_return_bins* b, const void* d, int64_t n, void* d_out4, hipStream_t st)
{
    if (n <= 0) return KR_OK;
    hipLaunchKernelGGL(reduce_return_kernel<false>, dim3(grid_for(n, 512)), dim3(kBlock), 0, st, (kr_ray_f64*) d, (long long) n, *b, (double*) d_out4, 0.0, 0.0);
    KR_LAUNCH_CHECK();
    return KR_OK;
}

int post_return_dev(double lo, double hi, const kr_return_bins* b, void* d, int64_t n, void* d_out4, hipStream_t st)
{
    if (n <= 0) return KR_OK;
    hipLaunchKernelGGL(reduce_return_kernel<true>, dim3(grid_for(n, 512)), dim3(kBlock), 0, st, (kr_ray_f64*) d, (long long) n, *b, (double*) d_out4, lo, hi);
    KR_LAUNCH_CHECK();
    return KR_OK;
}

int post_return_batch_dev(int count, double lo, double hi, const kr_return_bins* b, void* const* d, const int64_t* n, void* const* d_out4, hipStream_t st)
{
    static_assert(sizeof(ReturnChunk) <= 3840, "kernel arguments are limited to 4 KB");
    for (int base = 0; base < count; base += kReturnChunk) {
        ReturnChunk c;
        std::memset(&c, 0, sizeof c);
        int m = 0;
        int64_t n_max = 0;
        for (int i = base; i < count && i < base + kReturnChunk; i++) {
            if (n[i] <= 0) continue;
            c.item[m++] = ReturnItem{b[i], (kr_ray_f64*) d[i], (long long) n[i], (double*) d_out4[i]};
            n_max = std::max(n_max, n[i]);
        }
        if (m == 0) continue;
        // blocks per item: enough to keep ~16 waves per SIMD in flight over the whole chunk, never more than an item's rays need
        const int per_item = (int) std::max<int64_t>(1, std::min<int64_t>((n_max + kBlock - 1) / kBlock, std::max(64, 8192 / m)));
        hipLaunchKernelGGL(reduce_return_multi_kernel, dim3(per_item, m), dim3(kBlock), 0, st, c, lo, hi);
        KR_LAUNCH_CHECK();
    }
    return KR_OK;
}

int pointsource_init_emit_batch_dev(int count, const kr_pointsource* s, const double* V, int reverse, int projradius, void* const* d, const int64_t* n, hipStream_t st)
{
    static_assert(sizeof(SourceChunk) <= 3968, "kernel arguments are limited to 4 KB");
    for (int base = 0; base < count; base += kSourceChunk) {
        SourceChunk c;
        std::memset(&c, 0, sizeof c);
        int m = 0;
        int64_t n_max = 0;
        for (int i = base; i < count && i < base + kSourceChunk; i++) {
            if (n[i] <= 0) continue;
            int32_t nc = 0, nb = 0;
            kr_pointsource_count(&s[i], &nc, &nb);
            SourceTables tb;
            const int rc = source_tables(&s[i], nc, nb, &tb);
            if (rc != KR_OK) return rc;
            c.item[m++] = SourceItem{s[i], tb, V ? V[i] : s[i].V, (kr_ray_f64*) d[i], (long long) n[i], nc, nb};
            n_max = std::max(n_max, n[i]);
        }
        if (m == 0) continue;
        const int per_item = (int) std::max<int64_t>(1, std::min<int64_t>((n_max + kBlock - 1) / kBlock, std::max(64, 16384 / m)));
        hipLaunchKernelGGL(pointsource_init_emit_multi_kernel, dim3(per_item, m), dim3(kBlock), 0, st, c, reverse, projradius);
        KR_LAUNCH_CHECK();
    }
    return KR_OK;
}

int reduce_image_dev(const kr_image_bins* b, const void* d, int64_t n, void* d_planes, hipStream_t st)
{
    if (b->img_nx <= 0 || b->img_ny <= 0) { set_error("kr_reduce_image: image size must be positive"); return KR_EINVAL; }
    if (n <= 0) return KR_OK;
    hipLaunchKernelGGL(reduce_image_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, (const kr_ray_f64*) d, (long long) n, *b, (double*) d_planes);
    KR_LAUNCH_CHECK();
    return KR_OK;
}

}  // namespace kr
