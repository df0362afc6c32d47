/*
 * kr_trace.h -- C ABI of libkrtrace.so, the MI355X (gfx950) Kerr null-geodesic hot path.
 *
 * The reference (wilkinsdr/raytrace_cpu) has no FFI/plugin layer: its boundary is the C++ class API
 * Raytracer<T> / PointSource<T> / ImagePlane<T> (src/raytracer/raytracer.h:85-198).  This header is
 * the thin C ABI that a replacement Raytracer<T> calls from inside those member functions; each entry
 * point below names the reference function (file:line, relative to the reference tree) it replaces.
 * The host-side mirror of the class API that does exactly that lives in raytrace_cpu_amd/host/.
 *
 * Conventions
 *   - plain C, no C++/torch types; all sizes are int64_t; all entry points return 0 on success and a
 *     negative KR_E* code on failure (kr_last_error() gives the message).  There is NO CPU fallback:
 *     without a usable HIP device every compute entry point returns KR_ENODEVICE.
 *   - "_dev" entry points take DEVICE pointers (hipMalloc / torch data_ptr) and a hipStream_t passed
 *     as void* (NULL = default stream); they enqueue work and return without synchronising unless a
 *     kr_stats* is requested (stats need the kernel's counters -> the call synchronises the stream).
 *   - entry points without "_dev" take HOST pointers, stage through private device buffers and return
 *     with the host arrays updated (the contract of Raytracer<T>::run_raytrace: results are in
 *     rays[] when it returns, raytracer.cpp:63-127).
 *   - kr_ray_f64 / kr_ray_f32 are layout-identical to Ray<double> / Ray<float>
 *     (raytracer.h:65-78; 144 B / 84 B), so `rays` can be handed over without conversion.
 */
#ifndef KR_TRACE_H_
#define KR_TRACE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KR_ABI_VERSION 15

/* error codes */
#define KR_OK          0
#define KR_EINVAL     -1   /* bad argument (NULL pointer, unknown integrator/stop kind, Euler+destination) */
#define KR_ENODEVICE  -2   /* no HIP device / HIP runtime unusable */
#define KR_EHIP       -3   /* a HIP call failed; see kr_last_error() */
#define KR_ENOMEM     -4

/* enum class Integrator { Euler, RK4, RK45 }  (raytracer.h:83) */
#define KR_EULER 0
#define KR_RK4   1
#define KR_RK45  2

/* stop surfaces: the theta-limit overloads (raytracer.cpp:129,755,1260) and the three concrete
 * RayDestination classes (ray_destination.h:86,116,173) as POD descriptors */
#define KR_STOP_THETA      0   /* theta_max field; stop_params unused */
#define KR_STOP_FLATDISC   1   /* stop_params = {theta_lim} */
#define KR_STOP_DISC_ISCO  2   /* stop_params = {r_isco, r_out, theta_lim} */
#define KR_STOP_FLATPLANE  3   /* stop_params = {incl, phi0, z_s} */

/* ray status bit flags (raytracer.h:58-63) + one extension */
#define KR_STATUS_DEST        (1 << 0)
#define KR_STATUS_HORIZON     (1 << 1)
#define KR_STATUS_RLIM        (1 << 2)
#define KR_STATUS_STEPLIM     (1 << 3)
#define KR_STATUS_ERGO        (1 << 4)
#define KR_STATUS_NEG_ENERGY  (1 << 5)
/* extension: the RK45 retry loop met a NaN error norm.  The reference never leaves that loop
 * (raytracer.cpp:1438-1541); the device path ends the ray and sets this bit instead. */
#define KR_STATUS_NAN         (1 << 6)

/* kr_params.flags */
#define KR_FLAG_FAST_MATH     (1 << 0)  /* f64 trace only: same formulas with shared reciprocals, Newton-refined rcp/rsq, FMA
                                           contraction and stage sin/cos by angle addition instead of IEEE division/sqrt and one
                                           sincos per evaluation (a few ulp per operation; ~1.6x faster).  Rays whose outcome is
                                           rounding-decided in the reference may end differently: prefer KR_FLAG_HYBRID.
                                           0 = strict: the reference's association with IEEE + - * / sqrt, no contraction.  (A strict
                                           launch of >= 2^18 rays also puts its ill-conditioned rays on a side launch -- see HYBRID --
                                           which changes where they run, never their bits; env KR_NO_ISOLATE=1 disables that.) */

#define KR_FLAG_HYBRID        (1 << 1)  /* f64 trace only: rays whose polar motion / axial angular momentum is a cancellation residue
                                           (their outcome in the reference is decided by rounding) and NaN rays are integrated on
                                           the strict path, in a side launch whose waves own their SIMDs; all other rays take the
                                           fast-math path.  Reproduces the reference on every ray class at ~1.4x the strict speed.
                                           Nothing in the call waits for the device: the number of flagged rays stays in device
                                           memory, where the launches read it.  The concurrent launch runs on a second, internal
                                           stream of its own priority level that belongs to `stream` (released by kr_stream_destroy /
                                           kr_shutdown).  Growing the per-ray selector of a pooled workspace allocates (hipMalloc,
                                           which does not synchronise); nothing is freed in the launch path.
                                           Ignored by the f32 entry points and when KR_FLAG_FAST_MATH is set. */
#define KR_FLAG_RK45_ITERATE_ALL (1 << 2) /* RK45: iterate creeping captured rays to the step limit one step at a time, as the reference does,
                                           instead of extrapolating them (kr_stats.rk45_extrapolated_steps) */
#define KR_FLAG_BLOCKS_PER_CU(n)      (((n) & 0xF) << 8)   /* resident 256-thread workgroups per CU for the trace kernel; 0 = default: 3 for the main
                                                              launch of a split trace, for n >= 2e7 and for fast-math with n >= 5e6, else 2 */
#define KR_FLAG_GET_BLOCKS_PER_CU(f)  (((f) >> 8) & 0xF)

/* defaults, raytracer.h:19-44 */
#define KR_PRECISION        100.0
#define KR_THETA_PRECISION  50.0
#define KR_MAXDT            1.0
#define KR_MAXDT_RLIM       100.0
#define KR_MAXDPHI          0.1
#define KR_STEPLIM          10000000
#define KR_RK45_STEPLIM     100000
#define KR_MIN_STEP         1e-3

typedef struct kr_ray_f64 {
    double t, r, theta, phi;
    double pt, pr, ptheta, pphi;
    double k, h, Q;
    double emit, redshift;
    int32_t steps, status, rdot_sign, thetadot_sign, rdot_flips, equatorial_crossings;
    double alpha, beta;
} kr_ray_f64;

typedef struct kr_ray_f32 {
    float t, r, theta, phi;
    float pt, pr, ptheta, pphi;
    float k, h, Q;
    float emit, redshift;
    int32_t steps, status, rdot_sign, thetadot_sign, rdot_flips, equatorial_crossings;
    float alpha, beta;
} kr_ray_f32;

/* Everything Raytracer<T>::run_raytrace reads besides rays[] (members raytracer.h:89-99 + call
 * arguments raytracer.h:112-122).  Doubles carry float values exactly for the f32 entry points. */
typedef struct kr_params {
    double spin;             /* as stored by Raytracer (ImagePlane has already negated it, imageplane.cpp:12) */
    double horizon;          /* kerr_horizon(spin) or set_boundary() value */
    double precision;        /* PRECISION */
    double theta_precision;  /* THETA_PRECISION */
    double max_tstep;        /* MAXDT */
    double maxtstep_rlim;    /* MAXDT_RLIM */
    double max_phistep;      /* MAXDPHI */
    double rk45_tol;         /* 1e-8 */
    double r_max;            /* rlim */
    double theta_max;        /* thetalim (KR_STOP_THETA only) */
    double stop_params[4];
    int32_t integrator;      /* KR_EULER / KR_RK4 / KR_RK45 */
    int32_t stop_kind;       /* KR_STOP_* */
    int32_t steplim;         /* <=0: STEPLIM for Euler/RK4, RK45_STEPLIM for RK45 (raytracer.cpp:80) */
    int32_t flags;           /* KR_FLAG_* */
} kr_params;

/* counters gathered by the trace kernel (what integrator_perf_test.cpp:82-93 derives on the host) */
typedef struct kr_stats {
    int64_t rays_total;      /* n */
    int64_t rays_traced;     /* rays that entered a propagate loop (steps >= 0 and < steplim on entry) */
    int64_t steps_total;     /* sum of per-call `steps` over traced rays (every ++steps, incl. theta-flip iterations) */
    int64_t rk45_attempts;   /* RK45: trial steps evaluated (accepted + rejected) */
    int64_t rk45_rejects;    /* RK45: trial steps rejected */
    double  kernel_ms;       /* trace kernel duration, HIP events on the launch stream */
    double  h2d_ms, d2h_ms;  /* host-buffer entry points only */
    int64_t rays_strict_side;       /* rays classified ill-conditioned and traced by the strict side launch (KR_FLAG_HYBRID, and
                                       strict launches of >= 2^18 rays); 0 when the trace was a single launch */
    int64_t rk45_stationary_steps;  /* RK45: steps (included in steps_total and rk45_attempts) that were replayed as bare t/phi
                                       additions after a captured ray reached an exact fp64 fixed point in (r, theta, step);
                                       bit-identical to iterating them (kr_device.hpp::step_rk45) */
    int64_t rk45_extrapolated_steps; /* RK45: steps (included in steps_total and rk45_attempts) of captured rays whose r was stationary
                                       and whose theta advanced by a constant number of ulps per step: extrapolated to the step
                                       limit (r, theta, every integer output exact; t, phi, momenta to ~1e-11) */
    double  strict_side_ms;         /* split traces: duration of the strict side launch (caller's stream) ... */
    double  main_ms;                /* ... and of the main launch beside it (internal stream); kernel_ms spans both.  0 otherwise */
    int64_t longest_ray_steps;      /* most steps one ray took in this call: a ray is one sequential chain on a wave, so the launch that
                                       carries it lasts at least this many wave steps whatever else the GPU does (DESIGN.md "Known limit") */
    int64_t longest_ray_steps_strict_side; /* the same over the rays of the strict side launch of a split trace (0 otherwise):
                                       strict_side_ms / this = that launch's time per step on a wave of its own */
    int64_t steps_strict_side;      /* steps_total of the strict side launch alone (split traces; 0 otherwise): with the profiler's per-kernel counters,
                                       instructions per step of EACH of the two launches */
    int64_t rk45_evaluated_strict_side; /* RK45: trial steps the strict side launch evaluated in full (its attempts minus replayed / extrapolated steps) */
} kr_stats;

/* PointSource<T> ctor arguments (pointsource.h:24, pointsource.cpp:11-64) */
typedef struct kr_pointsource {
    double pos[4];
    double V, spin, tol;
    double dcosalpha, dbeta;
    double cosalpha0, cosalphamax, beta0, betamax;
    double E;
} kr_pointsource;

/* ImagePlane<T> ctor arguments (imageplane.h:26, imageplane.cpp:11-121); `spin` is the PHYSICAL spin,
 * the ctor's negation is applied inside */
typedef struct kr_imageplane {
    double dist, inc_deg;
    double x0, xmax, dx;
    double y0, ymax, dy;
    double spin, phi0, precision;
} kr_imageplane;

/* radial histogram of src/emissivity/emissivity.cpp:96-126 */
typedef struct kr_emis_bins {
    double r_min;            /* first bin edge */
    double dr;               /* logbin: ratio between edges; linear: width */
    double r_isco;           /* rays with r < r_isco are dropped */
    double gamma;            /* emis += redshift^-gamma */
    double spin;             /* unused by the filter (z = r cos(theta) only) but kept for symmetry with cartesian() */
    double num_primary_rays; /* flux += 1/(num_primary_rays * redshift) */
    int32_t nr;
    int32_t logbin;
} kr_emis_bins;

/* image accumulation of src/imageplane/imageplane_disc_image.cpp:122-161 */
typedef struct kr_image_bins {
    double x0, y0, img_dx, img_dy;
    double r_isco, r_disc;
    double q1, rb1, q2, rb2, q3;   /* powerlaw3, imageplane_disc_image.cpp:20-28 */
    int32_t img_nx, img_ny;
    int32_t flip_image;
    int32_t pad;
} kr_image_bins;

/* disc -> disc returning-radiation classification, src/return_radiation/disc_source_photonfrac_r.cpp:97-126
 * (that app is stale in the reference -- it calls accessors that no longer exist -- so this follows its loop body
 * against the live Ray<T> fields: cos(alpha) is rays[].alpha, beta is rays[].beta) */
typedef struct kr_return_bins {
    double r_isco, r_disc, r_esc;
    double source_r, source_phi;
    int32_t plane_iso;       /* weight = |sin(alpha) sin(beta)| instead of 1 */
    int32_t limb;            /* weight *= 1 + 2.06 |sin(alpha) sin(beta)| */
    int32_t weight_norm;     /* ray_count accumulates the weight instead of 1 */
    int32_t pad;
} kr_return_bins;

/* ---- runtime ---------------------------------------------------------------------------------- */
int         kr_abi_version(void);
const char* kr_last_error(void);
int         kr_device_count(void);                 /* <0: KR_ENODEVICE */
int         kr_set_device(int device);
int         kr_device_info(int* cu_count, int* clock_khz, int64_t* hbm_bytes, char* name, int name_len);
void        kr_params_default(kr_params* p, double spin);   /* Raytracer ctor defaults, raytracer.cpp:12-22 */
double      kr_kerr_horizon(double a);                      /* kerr_horizon(), src/include/kerr.h:14-20 */
double      kr_kerr_isco(double a, int sign);               /* kerr_isco(), kerr.h:23-32 (float-rounded A, B: sic) */
double      kr_disc_velocity(double r, double a, int sign); /* disc_velocity(), kerr.h:35-38 */
int64_t     kr_pointsource_count(const kr_pointsource* s, int32_t* n_cosalpha, int32_t* n_beta);   /* pointsource.cpp:12,16-17 */
int64_t     kr_imageplane_count(const kr_imageplane* s, int32_t* nx, int32_t* ny);                 /* imageplane.cpp:12-14 */
/* The transcendental values of the PointSource constructor (pointsource.cpp:38-46: alpha = acos(cosalpha0 + i dcosalpha), beta = beta0 + j dbeta;
 * raytracer.cpp:631-672: sin / cos of alpha and beta, sin / cos / tan of the source's polar angle pos[2]) computed on the HOST with the C library
 * the reference calls.  The device constructors (kr_pointsource_init*_dev_f64) read exactly these -- they upload them once per (device, grid) and
 * keep them until kr_shutdown -- so k, h, Q of a device-built ray carry the reference constructor's bits.  Needs no GPU.
 * alpha_sincos: 2 * n_cosalpha doubles (sin, cos interleaved); beta_sincos: 2 * n_beta; pos_sin_cos_tan: 3.  Any of them may be NULL. */
int         kr_pointsource_tables(const kr_pointsource* s, double* alpha_sincos, double* beta_sincos, double* pos_sin_cos_tan);

/* ---- the hot path: Raytracer<T>::run_raytrace, both overloads (raytracer.cpp:63-127, 972-1034) --
 * Host-pointer forms stage through a private device buffer and return when rays[] is final.  *_dev forms take a device
 * pointer and a hipStream_t (NULL: the default stream): work is enqueued on that stream and, with stats == NULL, the call
 * returns before it has finished; nothing in the launch path waits for the device.  Every call draws its queue counters,
 * index list, events and second stream from a per-device pool of workspaces and gives them back when its last kernel
 * has finished, so any number of traces may be in flight on one device, from several streams or host threads
 * (multi-launch drivers -- one launch per source radius, one per tolerance -- overlap their long-ray tails that way). */
int kr_trace_f64(const kr_params* p, kr_ray_f64* rays, int64_t n, kr_stats* stats);
int kr_trace_f32(const kr_params* p, kr_ray_f32* rays, int64_t n, kr_stats* stats);
int kr_trace_dev_f64(const kr_params* p, void* d_rays, int64_t n, void* stream, kr_stats* stats);
int kr_trace_dev_f32(const kr_params* p, void* d_rays, int64_t n, void* stream, kr_stats* stats);
/* The same trace in two halves, for callers that want the counters of overlapping launches: kr_trace_async_* enqueues and
 * returns a ticket at once; kr_trace_wait blocks until THAT trace has finished, fills *stats (may be NULL) and retires the
 * ticket; kr_trace_release retires it without waiting.  Every ticket must go to exactly one of the two (at most 512 may be
 * outstanding per device).  kr_trace_dev_*(.., stats) == async + wait;  (.., NULL) == async + release. */
int kr_trace_async_f64(const kr_params* p, void* d_rays, int64_t n, void* stream, void** ticket);
int kr_trace_async_f32(const kr_params* p, void* d_rays, int64_t n, void* stream, void** ticket);
/* `count` traces at once (one per tolerance of a sweep, per source radius, ...), each with its own ray buffer.  When all of them use the
 * same kernel instances (same integrator, same kind of stop surface, same arithmetic flags; count <= 256, every n >= 4096) the batch is
 * MERGED: one classification per trace, then ONE strict side launch and ONE main launch over all traces (every wave serves one of the
 * traces' queues), on streams[0]; other streams wait for the batch at both ends, so whatever the caller enqueued on streams[i] before
 * the call is seen and whatever it enqueues afterwards sees trace i's result -- but a merged batch wants ONE stream for all its traces
 * (18 streams waiting on it cost the 18-point RK45 sweep 2.5-6 s instead of 0.41 s).  Otherwise the strict side
 * launches of all traces are enqueued before any main launch, each trace on its own stream.  Strict traces are split whatever their
 * size.  Same results and counters as `count` kr_trace_async_f64 calls, bit for bit (kernel_ms etc. then time the whole batch).
 * streams may be NULL (all on the default stream). */
int kr_trace_batch_async_f64(int32_t count, const kr_params* const* p, void* const* d_rays, const int64_t* n, void* const* streams, void** tickets);
int kr_trace_wait(void* ticket, kr_stats* stats);
/* kr_trace_wait on `count` tickets (a batch's, in any order).  per_ticket (optional): count records; total (optional): the counters summed
 * (rays_total, rays_traced, steps_total, rk45_*, rays_strict_side) and the LARGEST kernel_ms / strict_side_ms / main_ms / longest_ray_steps*.  Every ticket is
 * released, also when one of them fails (the first error code is returned). */
int kr_trace_wait_many(int32_t count, void* const* tickets, kr_stats* per_ticket, kr_stats* total);
int kr_trace_release(void* ticket);
/* Progress of a trace in flight (run_raytrace's show_progress, raytracer.cpp:84-85, :107-115: a counter of rays whose loop iteration has STARTED).
 * kr_trace_poll: how many rays the trace behind `ticket` has taken off its work queue so far and whether it has finished; it neither waits nor
 * retires the ticket (an 8-byte DMA read of the queue head on a stream of the library's own: it completes while the trace kernels hold every SIMD).
 * kr_trace_progress_*: kr_trace_* (host pointers) which, while the trace runs, polls every 20 ms on the calling thread and calls
 * fn(rays_started rounded down to a multiple of `every`, n, user) each time a new multiple has been passed.  every <= 0 or fn == NULL: no calls. */
typedef void (*kr_progress_fn)(int64_t rays_started, int64_t rays_total, void* user);
int kr_trace_poll(void* ticket, int64_t* rays_started, int32_t* finished);
int kr_trace_progress_f64(const kr_params* p, kr_ray_f64* rays, int64_t n, kr_stats* stats, int64_t every, kr_progress_fn fn, void* user);
int kr_trace_progress_f32(const kr_params* p, kr_ray_f32* rays, int64_t n, kr_stats* stats, int64_t every, kr_progress_fn fn, void* user);

/* ---- O(N) passes either side of it ----------------------------------------------------------- */
/* Raytracer<T>::redshift_start(V, reverse, projradius)  raytracer.cpp:342-417 */
int kr_redshift_start_f64(double spin, double V, int reverse, int projradius, kr_ray_f64* rays, int64_t n);
int kr_redshift_start_dev_f64(double spin, double V, int reverse, int projradius, void* d_rays, int64_t n, void* stream);
/* Raytracer<T>::redshift(V, reverse, projradius, motion)  raytracer.cpp:420-447, 480-553 */
int kr_redshift_f64(double spin, double V, int reverse, int projradius, int motion, kr_ray_f64* rays, int64_t n);
int kr_redshift_dev_f64(double spin, double V, int reverse, int projradius, int motion, void* d_rays, int64_t n, void* stream);
/* Raytracer<T>::redshift(RayDestination*, reverse, ...) with the default four_velocity()
 * (raytracer.cpp:450-477, 556-600; ray_destination.h:59-78: every concrete class keeps velocity() = -1) */
int kr_redshift_dest_f64(double spin, int reverse, kr_ray_f64* rays, int64_t n);
int kr_redshift_dest_dev_f64(double spin, int reverse, void* d_rays, int64_t n, void* stream);
/* Raytracer<T>::range_phi(min, max)  raytracer.cpp:603-622 */
int kr_range_phi_f64(double lo, double hi, kr_ray_f64* rays, int64_t n);
int kr_range_phi_dev_f64(double lo, double hi, void* d_rays, int64_t n, void* stream);
/* Raytracer<T>::calculate_momentum()  raytracer.cpp:704-753 */
int kr_calculate_momentum_f64(double spin, kr_ray_f64* rays, int64_t n);
int kr_calculate_momentum_dev_f64(double spin, void* d_rays, int64_t n, void* stream);
/* The same five passes for Raytracer<float>: kr_ray_f32 records, float arithmetic throughout (the reference's float instantiation of the
 * same source lines; spin, V, lo, hi are float values carried in doubles).  Host-pointer forms copy the whole 84-byte record back. */
int kr_redshift_start_f32(double spin, double V, int reverse, int projradius, kr_ray_f32* rays, int64_t n);
int kr_redshift_start_dev_f32(double spin, double V, int reverse, int projradius, void* d_rays, int64_t n, void* stream);
int kr_redshift_f32(double spin, double V, int reverse, int projradius, int motion, kr_ray_f32* rays, int64_t n);
int kr_redshift_dev_f32(double spin, double V, int reverse, int projradius, int motion, void* d_rays, int64_t n, void* stream);
int kr_redshift_dest_f32(double spin, int reverse, kr_ray_f32* rays, int64_t n);
int kr_redshift_dest_dev_f32(double spin, int reverse, void* d_rays, int64_t n, void* stream);
int kr_range_phi_f32(double lo, double hi, kr_ray_f32* rays, int64_t n);
int kr_range_phi_dev_f32(double lo, double hi, void* d_rays, int64_t n, void* stream);
int kr_calculate_momentum_f32(double spin, kr_ray_f32* rays, int64_t n);
int kr_calculate_momentum_dev_f32(double spin, void* d_rays, int64_t n, void* stream);

/* ---- ray sources: PointSource / ImagePlane ctors (pointsource.cpp:11-64, imageplane.cpp:11-121) -- */
int kr_pointsource_init_f64(const kr_pointsource* s, kr_ray_f64* rays, int64_t n);
int kr_pointsource_init_dev_f64(const kr_pointsource* s, void* d_rays, int64_t n, void* stream);
int kr_imageplane_init_f64(const kr_imageplane* s, kr_ray_f64* rays, int64_t n);
int kr_imageplane_init_dev_f64(const kr_imageplane* s, void* d_rays, int64_t n, void* stream);
/* shard forms of the two ctors: slot k of d_rays receives ray (first + k*stride) of the source's own array, k < count.
 * Rank r of R uses first = r, stride = R: ray-cyclic sharding, nothing else has to be exchanged before the reducers. */
int kr_pointsource_init_strided_dev_f64(const kr_pointsource* s, int64_t first, int64_t stride, void* d_rays, int64_t count, void* stream);
int kr_imageplane_init_strided_dev_f64(const kr_imageplane* s, int64_t first, int64_t stride, void* d_rays, int64_t count, void* stream);

/* ---- fused ends of the emissivity pipeline (device-resident callers): the same per-ray arithmetic as the separate passes, one
 * pass over the records instead of two / three.  kr_pointsource_init_emit = PointSource ctor + redshift_start(V, reverse,
 * projradius) (pointsource.cpp:11-64 + raytracer.cpp:342-417), strided like kr_pointsource_init_strided_dev_f64;
 * kr_post_emissivity = range_phi(lo, hi) + redshift(V, reverse, projradius, motion) + the histogram of kr_reduce_emissivity_dev_f64
 * (raytracer.cpp:603-622, :420-553, emissivity.cpp:96-126); rays[] ends up exactly as after the separate calls. */
int kr_pointsource_init_emit_dev_f64(const kr_pointsource* s, int64_t first, int64_t stride, double V, int reverse, int projradius, void* d_rays, int64_t count,
                                     void* stream);
/* `count` sources at once -- the multi-radius drivers (disc_source_photonfrac_r.cpp:74-92: one PointSource per radius): what `count` calls of
 * kr_pointsource_init_emit_dev_f64(&s[i], 0, 1, V ? V[i] : s[i].V, reverse, projradius, d_rays[i], n[i], stream) write, bit for bit, in
 * ceil(count / 24) kernel launches instead of `count` (a hundred launches of 1e6 rays each reach a third of the store bandwidth of one large one).
 * s, V (may be NULL), d_rays and n are HOST arrays of `count` entries, read before the call returns. */
int kr_pointsource_init_emit_batch_dev_f64(int32_t count, const kr_pointsource* s, const double* V, int reverse, int projradius, void* const* d_rays, const int64_t* n,
                                           void* stream);
/* the same for the image pipeline: ImagePlane ctor + redshift_start(V, reverse, projradius) (imageplane.cpp:11-121; the negated spin
 * of the ImagePlane is applied inside), and redshift(V, reverse, projradius, motion) + range_phi(lo, hi) + the seven planes of
 * kr_reduce_image_dev_f64 (imageplane_disc_image.cpp:117-161; `spin` as stored by the Raytracer, i.e. negated) */
int kr_imageplane_init_emit_dev_f64(const kr_imageplane* s, int64_t first, int64_t stride, double V, int reverse, int projradius, void* d_rays, int64_t count,
                                    void* stream);
/* the same ctor for shards made of RUNS of rays: slot k receives source ray first + (k / run) * stride + k % run.  With run = (ray columns
 * per pixel column) * ny, stride = R * run and first = r * run, rank r of R owns whole pixel columns r, r + R, ... of the image: the ranks'
 * image planes are then disjoint and are GATHERED, not summed (bench.py --image-exchange gather).  run = 1 is the plain strided form. */
int kr_imageplane_init_emit_runs_dev_f64(const kr_imageplane* s, int64_t first, int64_t stride, int64_t run, double V, int reverse, int projradius, void* d_rays,
                                         int64_t count, void* stream);
int kr_post_image_dev_f64(double spin, double V, int reverse, int projradius, int motion, double lo, double hi, const kr_image_bins* b, void* d_rays, int64_t n,
                          void* d_planes, void* stream);
int kr_post_emissivity_dev_f64(double spin, double V, int reverse, int projradius, int motion, double lo, double hi, const kr_emis_bins* b, void* d_rays, int64_t n,
                               void* d_hist, void* stream);

/* ---- reducers of the two target apps --------------------------------------------------------- */
/* emissivity.cpp:96-126.  Outputs (length nr each): count, flux, emis, sum_redshift, sum_time -- the raw
 * accumulators BEFORE the divisions of emissivity.cpp:128-134; *disc_count = rays passing the filter. */
int kr_reduce_emissivity_f64(const kr_emis_bins* b, const kr_ray_f64* rays, int64_t n,
                             int64_t* count, double* flux, double* emis, double* sum_redshift, double* sum_time,
                             int64_t* disc_count);
/* d_hist: device buffer of nr*5+1 doubles laid out [count | flux | emis | sum_redshift | sum_time | disc_count];
 * counts are held as doubles (exact below 2^53).  The call ADDS into d_hist (zero it first). */
int kr_reduce_emissivity_dev_f64(const kr_emis_bins* b, const void* d_rays, int64_t n, void* d_hist, void* stream);
/* imageplane_disc_image.cpp:122-161.  Seven planes of img_nx*img_ny, [ix*img_ny + iy] like Array2D:
 * nrays(int32), flux, r, phi, enshift, time, emis -- raw sums BEFORE the divisions of :165-174. */
int kr_reduce_image_f64(const kr_image_bins* b, const kr_ray_f64* rays, int64_t n,
                        int32_t* nrays, double* flux, double* r, double* phi, double* enshift, double* time,
                        double* emis, int64_t* disc_count);
/* d_planes: device buffer of 7*img_nx*img_ny+1 doubles [nrays | flux | r | phi | enshift | time | emis | disc_count],
 * nrays held as doubles.  ADDS into d_planes. */
int kr_reduce_image_dev_f64(const kr_image_bins* b, const void* d_rays, int64_t n, void* d_planes, void* stream);

/* disc_source_photonfrac_r.cpp:97-126: out[4] = {ray_count, return_count, escape_count, lost_count} (weighted sums);
 * the app's three fractions are out[1..3] / out[0].  _dev ADDS into d_out4 (4 doubles, zero first). */
int kr_reduce_return_f64(const kr_return_bins* b, const kr_ray_f64* rays, int64_t n, double out[4]);
int kr_reduce_return_dev_f64(const kr_return_bins* b, const void* d_rays, int64_t n, void* d_out4, void* stream);
/* range_phi(lo, hi) + kr_reduce_return_dev_f64 in one pass over the records (the returning-radiation driver's two passes after a trace) */
int kr_post_return_dev_f64(double lo, double hi, const kr_return_bins* b, void* d_rays, int64_t n, void* d_out4, void* stream);
/* kr_post_return_dev_f64 for `count` launches' rays at once: b, d_rays, n, d_out4 are HOST arrays of `count` entries (d_out4[i]: 4 doubles on the
 * device, ADDED into); the same rays[] and, up to the order of the additions, the same sums as `count` single calls, in ceil(count / 32) launches. */
int kr_post_return_batch_dev_f64(int32_t count, double lo, double hi, const kr_return_bins* b, void* const* d_rays, const int64_t* n, void* const* d_out4,
                                 void* stream);

/* ---- diagnostics ------------------------------------------------------------------------------- */
/* out[i] = op(a[i], b[i]) evaluated ON THE DEVICE with the exact primitive the trace kernel uses (host pointers):
 * 0 a/b (compiler IEEE)  1 a/b (lean IEEE chain of the strict path)  2 sqrt(a) (compiler)  3 sqrt(a) (lean)
 * 4 sin(a)  5 cos(a) (compact polar-angle sincos)  6 a*rcp(b)  7 sqrt(a) (fast-math path)  8 sin  9 cos  10 pow(a,b) (device libm)
 * 11-18 further primitives of the two arithmetic paths (kr_post.hip::arith_probe_kernel)  19 a after 20 000 additions of b (kr_replay.hpp)
 * 20 1.2345678901234567 / ((a a) b), 21 b / (a a) through the assembled reciprocals of one derivative evaluation (kr_device.hpp::StageRecips) */
int kr_debug_arith_f64(int op, const double* a, const double* b, double* out, int64_t n);

/* ---- a long-lived host ray array (what Raytracer<T> holds as `rays`) ----------------------------- */
/* kr_host_attach keeps a device buffer for the array until kr_host_detach.  Host-pointer entry points called on an attached array
 * (or on a sub-range of it) allocate nothing and copy back only what their pass modifies: `emit` (redshift_start), `phi` (range_phi), `redshift` (redshift*), the four momenta (calculate_momentum),
 * the whole record (trace, the source constructors).  Semantics are unchanged: the host array is the input of every call and is
 * complete when the call returns.  ray_bytes = sizeof(kr_ray_f64) or sizeof(kr_ray_f32).  Detach before freeing the array. */
int kr_host_attach(void* rays, int64_t n, int32_t ray_bytes);
int kr_host_detach(void* rays);

/* ---- device memory helpers for callers without a HIP runtime of their own --------------------- */
int kr_malloc(void** d_ptr, int64_t bytes);
int kr_free(void* d_ptr);
/* page-locked host memory: kr_memcpy_* to / from it run at full PCIe rate without a staging copy or first-touch faults */
int kr_host_alloc(void** h_ptr, int64_t bytes);
int kr_host_free(void* h_ptr);
int kr_memcpy_h2d(void* d_dst, const void* h_src, int64_t bytes);
int kr_memcpy_d2h(void* h_dst, const void* d_src, int64_t bytes);
int kr_memset(void* d_ptr, int value, int64_t bytes);
int kr_synchronize(void* stream);
/* streams for such callers (hipStreamCreateWithFlags(hipStreamNonBlocking) / hipStreamDestroy); the handle is what the *_dev
 * entry points take as `stream` */
int kr_stream_create(void** stream);
int kr_stream_destroy(void* stream);                /* also releases the internal side stream split traces on `stream` used */
/* Hardware queues.  Traces are meant to overlap (a split trace uses two streams, a multi-launch driver keeps many in flight) and the
 * HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues per device -- 4 by default: 18 concurrent RK45 sweep points take
 * 1.65 s on 4 queues, 0.76 s on 16 (profiles/r02_hw_queues.txt).  The variable is read when the runtime initialises, and it is
 * process-global: the library does NOT touch it when it is loaded.  An application that owns its process calls kr_configure_process() FIRST (before
 * any other HIP user -- this library, PyTorch, RCCL -- starts the runtime); it sets GPU_MAX_HW_QUEUES=16 unless the user chose a value.
 * Returns 1 if it set (or found) the variable before THIS LIBRARY touched the runtime, 0 if this library already had: it cannot see whether another
 * HIP user in the process (PyTorch, RCCL) started the runtime earlier -- then the setting comes too late and 1 is returned all the same.  The Python
 * package sets the default in raytrace_cpu_amd/__init__.py, i.e. at import, before `import torch` can start the runtime; bench.py, the kr_* apps
 * and the class mirror call this function first thing. */
int kr_configure_process(void);
/* Waits for the devices the library has used, then releases every pooled trace workspace, internal stream and PointSource table.  Refused with
 * KR_EINVAL (nothing released) while a ticket of kr_trace_async_* / kr_trace_batch_async_f64 is outstanding: wait for or release the tickets first.
 * Optional, and never done implicitly: at process exit the HIP runtime may already be gone when this library is unloaded. */
int kr_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif /* KR_TRACE_H_ */
