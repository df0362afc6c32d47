// kr_trace.hip -- the hot path: Raytracer<T>::run_raytrace (reference raytracer.cpp:63-127, :972-1034)
// as a persistent gfx950 kernel.
//
// Mapping onto the hardware
//   * one ray per lane, all ray state in VGPRs, metric terms recomputed every evaluation; no LDS, no MFMA
//     (a latency-bound scalar fp64 ODE, SURVEY.md 8d: bound = fp64 VALU, ~0.5 B of HBM traffic per step);
//   * rays live in HBM as the reference's own 144-B (84-B) AoS records; a lane touches its record exactly
//     twice (load on entry, store on exit) -> 288 B per ray, irrelevant next to ~450 steps x ~1200 instructions;
//   * divergence: rays need 60 ... 100 000 steps.  Waves are persistent: a lane whose ray has ended writes it
//     back and, through one wave-aggregated atomicAdd on a global queue head (ballot + popcount), pulls the
//     next unprocessed ray, so a wave never idles on its slowest ray while work is left.  One loop iteration =
//     one integration step (RK45: one trial step) for every lane that holds a ray;
//   * the grid is sized to the device (CUs x resident waves), not to n;
//   * a launch cannot end before its longest ray does.  Large launches are therefore split in two concurrent ones
//     (dispatch_split): the few ill-conditioned rays -- in the lamp-post workloads also the longest -- run the strict
//     arithmetic on waves that own their SIMDs (HOG instances), everything else fills the rest of the chip, with the
//     fast arithmetic (KR_FLAG_HYBRID) or the strict one (flags = 0; same bits as a single launch).
//
// Work-queue exit: every wave leaves the loop once the queue head has passed n AND none of its lanes holds a
// ray; every ray ends after at most steplim iterations (steps is incremented on every path through a step,
// and RK45 retries either shrink the step to MIN_STEP and force-accept, or end the ray on a NaN error norm).

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <limits>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "kr_common.hpp"
#include "kr_device.hpp"
#include "kr_post_device.hpp"

namespace kr {

namespace {

constexpr int kBlock = 256;          // classification kernel
// Trace kernels: ONE wave per workgroup -- a wave gives its registers back the moment IT has finished, not when the slowest of four has
// (main launch 93.1 -> 87.2 ms at 1e7 rays against 256-thread workgroups).  The HOG instances (the strict side launch) too: their first claims
// are static (wave g takes list slots 64 g ..), so the listed rays go to the lowest-numbered workgroups -- which the dispatcher spreads over as
// many compute units -- and every other workgroup leaves at once.  (Four HOG waves per workgroup, i.e. a workgroup that owns its compute unit,
// measured worse: four lone waves on one CU slow one another down more than nine waves of the main launch do; profiles/r03_ab_experiments.txt.)
constexpr int kTraceBlock = 64;
// A wave goes back to the queue when at least this many of its lanes are free (or none holds a ray).  The refill / finish / store code runs with
// only the free lanes active, ~500 vector instructions per visit -- as much as an RK4 step: visiting for every single finished lane cost the image
// plane (1.25 lanes per visit) 11 % and the Euler launches 26 %; waiting for 4 leaves ~1.5 lanes of 64 idle on average.
// Measured 1 -> 4 (8 is the same): image plane 170.4 -> 151.7 ms, Euler 1e7 rays 55.2 -> 40.8, returning radiation 318 -> 307, headline 81.8 -> 80.7,
// RK45 412 -> 406 (profiles/r02_ab_experiments.txt).
#ifndef KR_REFILL_MIN
#define KR_REFILL_MIN 4
#endif
constexpr int kLongRaySteps = 2048;  // a wave whose oldest ray is older than 1 x / 3 x / 8 x this raises its issue priority to 1 / 2 / 3 (trace_body)
#ifndef KR_OCC_STATS
#define KR_OCC_STATS 0               // 1: lane-occupancy bookkeeping of the step loop (diagnostic builds: scripts/gpu_occ_stats.sh), printed by trace_wait
#endif
constexpr int kCounters = KR_OCC_STATS ? 13 : 8;         // [0] queue head, [1] rays traced, [2] steps, [3] rk45 attempts, [4] rk45 rejects, [5] rk45 stationary steps, [6] rk45 extrapolated steps,
                                                         // [7] steps of the launch's longest ray (atomicMax); KR_OCC_STATS builds: [8..12] occupancy sums
constexpr int kCounterBlocks = 4;    // main launch, strict side launch, strict overflow launch, split bookkeeping ([1] = number of ill-conditioned rays)
constexpr int kListCap = 32768;      // index-list entries of the strict side launch

template <typename T> struct RayOf;
template <> struct RayOf<double> { using type = kr_ray_f64; };
template <> struct RayOf<float> { using type = kr_ray_f32; };

// ---- AoS record <-> registers -----------------------------------------------------------------
KR_DEV void load_ray(const kr_ray_f64* p, Lane<double>& s)
{
    const double2* d = reinterpret_cast<const double2*>(p);      // 144-B records, 16-B aligned
    const double2 a0 = d[0], a1 = d[1], a2 = d[2], a3 = d[3], a4 = d[4];
    s.t = a0.x; s.r = a0.y; s.theta = a1.x; s.phi = a1.y;
    s.pt = a2.x; s.pr = a2.y; s.ptheta = a3.x; s.pphi = a3.y;
    s.k = a4.x; s.h = a4.y;
    s.Q = p->Q;
    const int2* iv = reinterpret_cast<const int2*>(&p->steps);
    const int2 i0 = iv[0], i1 = iv[1], i2 = iv[2];
    s.steps0 = i0.x; s.status = i0.y; s.rdot_sign = i1.x; s.thetadot_sign = i1.y; s.rdot_flips = i2.x; s.eq_cross = i2.y;
}

KR_DEV void store_ray(kr_ray_f64* p, const Lane<double>& s, int32_t out_steps)
{
    double2* d = reinterpret_cast<double2*>(p);
    d[0] = make_double2(s.t, s.r);
    d[1] = make_double2(s.theta, s.phi);
    d[2] = make_double2(s.pt, s.pr);
    d[3] = make_double2(s.ptheta, s.pphi);
    int2* iv = reinterpret_cast<int2*>(&p->steps);
    iv[0] = make_int2(out_steps, s.status);
    iv[1] = make_int2(s.rdot_sign, s.thetadot_sign);
    iv[2] = make_int2(s.rdot_flips, s.eq_cross);
}

KR_DEV void load_ray(const kr_ray_f32* p, Lane<float>& s)
{
    s.t = p->t; s.r = p->r; s.theta = p->theta; s.phi = p->phi;
    s.pt = p->pt; s.pr = p->pr; s.ptheta = p->ptheta; s.pphi = p->pphi;
    s.k = p->k; s.h = p->h; s.Q = p->Q;
    s.steps0 = p->steps; s.status = p->status; s.rdot_sign = p->rdot_sign; s.thetadot_sign = p->thetadot_sign;
    s.rdot_flips = p->rdot_flips; s.eq_cross = p->equatorial_crossings;
}

KR_DEV void store_ray(kr_ray_f32* p, const Lane<float>& s, int32_t out_steps)
{
    p->t = s.t; p->r = s.r; p->theta = s.theta; p->phi = s.phi;
    p->pt = s.pt; p->pr = s.pr; p->ptheta = s.ptheta; p->pphi = s.pphi;
    p->steps = out_steps; p->status = s.status; p->rdot_sign = s.rdot_sign; p->thetadot_sign = s.thetadot_sign;
    p->rdot_flips = s.rdot_flips; p->equatorial_crossings = s.eq_cross;
}

KR_DEV unsigned long long wave_max(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_down(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

template <typename T> KR_DEV unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---- the persistent kernel ------------------------------------------------------------------------
// METHOD: KR_EULER / KR_RK4 / KR_RK45.  REFILL_MIN: a wave goes back to the queue when at least this many of its
// lanes are free (or when none holds a ray).
// `list` (optional): the launch works on rays list[0 .. n) instead of rays 0 .. n); `n_ptr` (optional): the item count is read
// from device memory (the classification kernel of the split path produced it; see n_mode below); `mask` (optional): only
// rays with mask[i] == mask_want are traced (the others belong to another launch of the split).  HOG: the kernel claims the whole register
// file (512 VGPR+AGPR per lane), so each of its waves owns its SIMD and no other kernel's wave can be co-resident on
// the CUs it occupies -- used for the few ill-conditioned / long rays that define the critical path.
// Resident waves per SIMD the register allocation must allow: HOG 1 (the scheduler may trade registers for ILP; capping it at (1, 1) measured 2 %
// slower), RK4 3 (<= 168 VGPRs), RK45 2, Euler 4 (its step is short and branchy: at 3 waves per SIMD the vector unit is 78 % busy; 1e7 rays
// 38.5 / 31.4 / 28.3 ms at 2 / 3 / 4 resident waves, profiles/r03_ab_experiments.txt).
// (Euler at 5: the fast kernel fits -- 96 registers, 32 B of spills -- and an all-fast 1e7-ray launch gains 2.5 %, but the hybrid pass and the
// returning-radiation batches, whose side launches share the chip with it, lose 1-3 %: profiles/r04_ab_experiments.txt.)
#define KR_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(HOG ? 1 : METHOD == KR_RK4 ? 3 : METHOD == KR_RK45 ? 2 : 4, 8)))
// everything one trace launch works on; a batch of traces hands the kernel an array of these (trace_multi_kernel)
template <typename T> struct TraceDesc {
    typename RayOf<T>::type* rays;
    long long n;
    TraceConsts<T> c;
    unsigned long long* counters;
    const int* list;
    const unsigned long long* n_ptr;
    const unsigned char* mask;
    int n_mode, mask_want;
};

template <typename T, int METHOD, bool USE_DEST, bool FAST, bool HOG, int REFILL_MIN>
KR_DEV void trace_body(typename RayOf<T>::type* __restrict__ rays, long long n, const TraceConsts<T>& c, unsigned long long* __restrict__ counters,
                       const int* __restrict__ list, const unsigned long long* __restrict__ n_ptr, int n_mode, const unsigned char* __restrict__ mask, int mask_want,
                       int& has_prio, long long first_slot = -1, unsigned long long head_offset = 0)
{
    if (n_ptr) {
        // the item count was produced on the device (classify_kernel) and never visits the host:
        // n_mode 1: the first min(n, *n_ptr) list entries;  n_mode 2: all n slots, but only if the list overflowed (else nothing)
        const long long m = (long long) *n_ptr;
        n = (n_mode == 1) ? (m < n ? m : n) : (m > (long long) kListCap ? n : 0);
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long lane_bit = 1ull << lane;

    Lane<T> s;
    long long idx = -1;
    bool have = false;          // this lane holds a ray
    bool pend = false;          // this lane's ray has ended and is still in its registers: written out at the wave's next visit to the queue (or on exit)
    bool exhausted = false;     // wave-uniform: the queue head has passed n
    unsigned long long my_steps = 0, my_traced = 0;
    int32_t my_longest = 0;     // most steps any of this lane's rays took in this call
    uint32_t my_attempts = 0, my_rejects = 0, my_stationary = 0, my_creep = 0;

#if KR_OCC_STATS
    unsigned long long occ_iters = 0, occ_tail_iters = 0, occ_tail_steps = 0, occ_refills = 0, occ_refill_lanes = 0;
#endif
    for (;;) {
        const unsigned long long need = __builtin_amdgcn_ballot_w64(!have);      // (not __ballot: that one takes its predicate through a vector register and a compare)
        const int n_need = __popcll(need);
        const bool any_have = (need != ~0ull);

        // A wave visits the queue when enough of its lanes are free, and once more when it leaves: rays that have ended since the last visit are
        // written out there -- the ONE place in the kernel where a ray is stored.
        const bool visit = !exhausted && n_need > 0 && (n_need >= REFILL_MIN || !any_have);
        const bool leaving = !visit && !any_have;          // nothing held and nothing left to take
        if (visit || leaving) {
            if (pend) {
                // (not under a divergent branch of its own in the step loop: that branch ran in one wave iteration out of nine for a single
                // lane's ~40 instructions)
                my_steps += (unsigned long long) s.steps;
                my_longest = s.steps > my_longest ? s.steps : my_longest;
                store_ray(&rays[idx], s, finish_status<T, USE_DEST>(s, c));
                pend = false;
            }
            if (leaving) break;
            // The launch cannot end before its longest ray does, and a ray advances one step per iteration of ITS wave: a wave that carries a long
            // ray (orbiting / polar-axis rays: 2e4..1e7 steps against a median of ~450) is given issue priority over its SIMD neighbours so that the
            // critical path runs at single-wave speed instead of at 1/(waves per SIMD) of it -- graded: the longer the wave's oldest ray, the higher
            // its priority (0..3), so that the rays that define the critical path do not share their level with the many merely "longish" ones.
            // Re-evaluated here, at the queue visits (a wave that carries a long ray keeps visiting for its other 63 lanes until the queue is empty;
            // a counter in the step loop cost three vector instructions per step).  A wave that owns its SIMD (HOG) has nobody to overtake.
            if constexpr (!HOG) {
                const int32_t st = have ? s.steps : 0;
                auto any = [](bool x) { return __builtin_amdgcn_ballot_w64(x) != 0; };
                const int want_prio = any(st > 8 * kLongRaySteps) ? 3 : any(st > 3 * kLongRaySteps) ? 2 : any(st > kLongRaySteps) ? 1 : 0;
                if (want_prio != has_prio) {
                    has_prio = want_prio;
                    switch (want_prio) {
                        case 3: __builtin_amdgcn_s_setprio(3); break;
                        case 2: __builtin_amdgcn_s_setprio(2); break;
                        case 1: __builtin_amdgcn_s_setprio(1); break;
                        default: __builtin_amdgcn_s_setprio(0); break;
                    }
                }
            }
#if KR_OCC_STATS
            ++occ_refills; occ_refill_lanes += n_need;
#endif
            // wave-aggregated dequeue: one atomic for all free lanes
            const int leader = __ffsll((long long) need) - 1;
            unsigned long long base = 0;
            if (first_slot >= 0) {
                // (HOG instances: this wave's first 64 slots are its own by position; the shared queue head counts from head_offset on)
                base = (unsigned long long) first_slot;
                first_slot = -1;
            } else {
                if (lane == leader) base = atomicAdd(&counters[0], (unsigned long long) n_need);
                base = __shfl(base, leader, 64) + head_offset;
            }
            if (base + (unsigned long long) n_need >= (unsigned long long) n) exhausted = true;
            if (!have) {
                const long long slot = (long long) base + __popcll(need & (lane_bit - 1));
                if (slot < n) {
                    // (the record is loaded beside its mask byte, not after it -- one memory round trip per visit instead of two; a ray that belongs to
                    // the other launch of a split, 0.03 % of them, is dropped again)
                    long long mine = slot;
                    if constexpr (HOG) { if (list) mine = (long long) list[slot]; }          // (only side launches work from a list)
                    const unsigned char* launch = mask ? mask + slot : (const unsigned char*) &rays[mine];      // one straight line of loads, mask or not
                    const unsigned char launch_of_ray = *launch;
                    load_ray(&rays[mine], s);
                    // skip rule of run_raytrace (raytracer.cpp:116-117)
                    if ((!mask || launch_of_ray == (unsigned char) mask_want) && s.steps0 >= 0 && s.steps0 < c.steplim) {
                        idx = mine;
                        have = true;
                        ++my_traced;
                        s.steps = 0;
                        s.r_was_positive = false;
                        s.theta_was_positive = true;
                        s.in_retry = false;
                        s.creep_m = 0;
                        s.creep_run = 0;
                        s.creep_mode = false;
                        s.fsal_valid = false;
                        energy_guard_set(s);
                        if (METHOD == KR_RK45) rk45_seed(s, c);
                        if (!loop_cond<T, USE_DEST>(s, c)) {
                            // zero-iteration call: only the epilogue runs (at the next visit)
                            have = false;
                            pend = true;
                        }
                    }
                }
            }
            continue;   // re-evaluate the ballots (skipped / zero-iteration rays leave lanes free)
        }

        int replay_batch = 1;
        if constexpr (METHOD == KR_RK45 && sizeof(T) == 8) {
            // the tail of an RK45 launch is waves that hold nothing but creeping captured rays: they take 16 cheap steps per iteration
            if (!__any(have && !s.creep_mode)) replay_batch = 16;      // (__any, not the ballot builtin: with the builtin this kernel's allocation came out 19 % slower)
        }
#if KR_OCC_STATS
        ++occ_iters;
        if (exhausted) { ++occ_tail_iters; occ_tail_steps += have ? 1 : 0; }
#endif
        if (have) {
            bool fin;
            if (METHOD == KR_EULER) fin = step_fixed<T, false, USE_DEST, FAST, HOG>(s, c);
            else if (METHOD == KR_RK4) fin = step_fixed<T, true, USE_DEST, FAST, HOG>(s, c);
            else fin = step_rk45<T, USE_DEST, FAST, HOG>(s, c, my_attempts, my_rejects, my_stationary, my_creep, replay_batch);
            if (fin) {
                have = false;
                pend = true;
            }
        }
    }
    // per-wave totals -> global counters (4 atomics per wave, once)
    const unsigned long long w_traced = wave_sum<T>(my_traced);
    const unsigned long long w_steps = wave_sum<T>(my_steps);
    const unsigned long long w_att = wave_sum<T>((unsigned long long) my_attempts);
    const unsigned long long w_rej = wave_sum<T>((unsigned long long) my_rejects);
    const unsigned long long w_sta = wave_sum<T>((unsigned long long) my_stationary);
    const unsigned long long w_creep = wave_sum<T>((unsigned long long) my_creep);
    const unsigned long long w_longest = wave_max((unsigned long long) my_longest);
#if KR_OCC_STATS
    {
        const unsigned long long w_tail_steps = wave_sum<T>(occ_tail_steps);
        if (lane == 0) {
            atomicAdd(&counters[8], occ_iters); atomicAdd(&counters[9], occ_tail_iters); atomicAdd(&counters[10], w_tail_steps);
            atomicAdd(&counters[11], occ_refills); atomicAdd(&counters[12], occ_refill_lanes);
        }
    }
#endif
    if (lane == 0) {
        if (w_longest) atomicMax(&counters[7], w_longest);
        if (w_sta) atomicAdd(&counters[5], w_sta);
        if (w_creep) atomicAdd(&counters[6], w_creep);
        if (w_traced) atomicAdd(&counters[1], w_traced);
        if (w_steps) atomicAdd(&counters[2], w_steps);
        if (w_att) atomicAdd(&counters[3], w_att);
        if (w_rej) atomicAdd(&counters[4], w_rej);
    }
}

// A wave that owns its SIMD has 512 vector registers to itself and the same ~100 scalar ones as any other wave: in the side launch's kernels the launch
// constants live in vector registers (the empty asm makes them opaque to the compiler's uniformity analysis) instead of being spilled to lanes and
// read back in the step loop, where such a wave pays four cycles for every instruction of any kind.  Euler and RK45 side launches -2 ... -3 %; the RK4
// kernel's allocation comes out 1 % slower with it and stays as it was (profiles/r04_ab_experiments.txt).
template <typename T> KR_DEV void consts_into_vector_registers(TraceConsts<T>& c)
{
    if constexpr (sizeof(T) == 8) {
        auto hold = [](T& x) { asm("" : "+v"(x)); };
        hold(c.a); hold(c.horizon); hold(c.rlim); hold(c.precision); hold(c.theta_precision); hold(c.max_tstep); hold(c.tol);
        hold(c.inv_precision); hold(c.inv_theta_precision); hold(c.tstep_rlim_eff); hold(c.phistep_eff); hold(c.theta_lo); hold(c.theta_hi);
    }
}

template <typename T, int METHOD, bool USE_DEST, bool FAST, bool HOG, int REFILL_MIN>
__global__ void __attribute__((amdgpu_flat_work_group_size(kTraceBlock, kTraceBlock))) KR_WAVES_ATTR
trace_kernel(typename RayOf<T>::type* __restrict__ rays, long long n, TraceConsts<T> c, unsigned long long* __restrict__ counters,
             const int* __restrict__ list, const unsigned long long* __restrict__ n_ptr, int n_mode, const unsigned char* __restrict__ mask, int mask_want)
{
    if (HOG) asm volatile("; claim the whole register file" ::: "v255", "a255");
    int has_prio = 0;
    if constexpr (HOG) {
        if constexpr (METHOD != KR_RK4) consts_into_vector_registers(c);
        trace_body<T, METHOD, USE_DEST, FAST, HOG, REFILL_MIN>(rays, n, c, counters, list, n_ptr, n_mode, mask, mask_want, has_prio, (long long) blockIdx.x * 64,
                                                               (unsigned long long) gridDim.x * 64);
    } else {
        trace_body<T, METHOD, USE_DEST, FAST, HOG, REFILL_MIN>(rays, n, c, counters, list, n_ptr, n_mode, mask, mask_want, has_prio);
    }
}

// ONE grid over MANY traces (kr_trace_batch_async_f64 when all traces of the batch use the same kernel instances).  Every wave serves
// ONE trace -- wave_trace[workgroup index], or workgroup index mod n_desc when no table is given -- with the persistent loop above,
// and leaves when that trace's queue is exhausted and its own lanes have drained.  The table interleaves the traces in proportion to
// their ray counts and the grid is larger than what is resident, so that waves of later workgroups move in wherever earlier ones have
// left: traces balance at wave granularity, a trace's tail is covered by the other traces' work, and the whole batch is two or three
// launches on two streams whatever the number of traces (beyond ~16 streams per process side launches slow down,
// profiles/r02_hw_queues.txt).  (A first version let each wave walk through all traces in turn: every wave then paid the tail of one
// long ray PER TRACE, 2.9 s for the 18-point sweep instead of 0.6 s.)  Per-ray arithmetic is the single-trace kernel's: same bits.
template <typename T, int METHOD, bool USE_DEST, bool FAST, bool HOG, int REFILL_MIN>
__global__ void __attribute__((amdgpu_flat_work_group_size(kTraceBlock, kTraceBlock))) KR_WAVES_ATTR
trace_multi_kernel(const TraceDesc<T>* __restrict__ descs, int n_desc, const int* __restrict__ wave_trace)
{
    if (HOG) asm volatile("; claim the whole register file" ::: "v255", "a255");
    int has_prio = 0;
    const int ti = wave_trace ? wave_trace[blockIdx.x] : (int) (blockIdx.x % (unsigned) n_desc);
    const TraceDesc<T>* d = &descs[ti];                           // wave-uniform: scalar loads
    if constexpr (HOG) {
        // (HOG batches map workgroup -> trace by modulo: workgroup b is the (b / n_desc)-th of its trace's gridDim.x / n_desc workgroups)
        const long long g = (long long) (blockIdx.x / (unsigned) n_desc);
        if constexpr (METHOD != KR_RK4) {
            TraceConsts<T> c = d->c;
            consts_into_vector_registers(c);
            trace_body<T, METHOD, USE_DEST, FAST, HOG, REFILL_MIN>(d->rays, d->n, c, d->counters, d->list, d->n_ptr, d->n_mode, d->mask, d->mask_want, has_prio, g * 64,
                                                                   (unsigned long long) (gridDim.x / (unsigned) n_desc) * 64);
        } else {
            trace_body<T, METHOD, USE_DEST, FAST, HOG, REFILL_MIN>(d->rays, d->n, d->c, d->counters, d->list, d->n_ptr, d->n_mode, d->mask, d->mask_want, has_prio, g * 64,
                                                                   (unsigned long long) (gridDim.x / (unsigned) n_desc) * 64);
        }
    } else {
        trace_body<T, METHOD, USE_DEST, FAST, HOG, REFILL_MIN>(d->rays, d->n, d->c, d->counters, d->list, d->n_ptr, d->n_mode, d->mask, d->mask_want, has_prio);
    }
}

// ---- hybrid path: which rays must be integrated with the reference's exact arithmetic? ----------------------------
// A ray is ILL-CONDITIONED when its polar motion or its axial angular momentum is a cancellation residue:
//   thetadot^2 rho^4 = Q + (k a cos + h cos/sin)(k a cos - h cos/sin)  with |sum| <= 1e-9 (|Q| + |product|), or |h| < 1e-13
// (PointSource rays emitted at beta = -pi: sin(beta) = -1.2e-16; ImagePlane rays on x = 0 / y = 0; NaN rays).  The
// reference's outcome for such a ray is decided by the rounding of exactly its own operation sequence, so only the
// strict path reproduces it; every other ray is insensitive to a few ulp per operation (tests/parity.py) and may
// take the fast path.  In the lamp-post workloads the ill-conditioned rays are also the longest ones (they ride the
// polar axis in MIN_STEP steps), which is why they get SIMDs of their own (HOG launch).
KR_DEV bool ill_conditioned(double k, double h, double Q, double theta, double a)
{
    double sn, cs;
    kr_sincos_f64(theta, sn, cs);
    const double kac = k * a * cs;
    const double hcs = h * cs / sn;
    const double prod = (kac + hcs) * (kac - hcs);
    const double sum = Q + prod;
    return !(__builtin_fabs(sum) > 1e-9 * (__builtin_fabs(Q) + __builtin_fabs(prod))) || !(__builtin_fabs(h) >= 1e-13);
}

// wave-aggregated append; ill-conditioned rays are rare (a column / a row of the source grid), so are the atomics.
// mask: 0 = main launch, 1 = listed (strict side launch), 2 = ill-conditioned but the list is full (strict overflow launch)
KR_DEV unsigned char classify_append(bool strict, long long i, int* __restrict__ list_strict, unsigned long long* __restrict__ n_strict)
{
    const unsigned long long m = __ballot(strict);
    unsigned char mine = 0;
    if (m != 0) {
        const int lane = threadIdx.x & 63;
        const int leader = __ffsll((long long) m) - 1;
        unsigned long long base = 0;
        if (lane == leader) base = atomicAdd(n_strict, (unsigned long long) __popcll(m));
        base = __shfl(base, leader, 64);
        if (strict) {
            const unsigned long long slot = base + __popcll(m & ((1ull << lane) - 1));
            if (slot < (unsigned long long) kListCap) { list_strict[slot] = (int) i; mine = 1; }
            else mine = 2;
        }
    }
    return mine;
}

__global__ void __launch_bounds__(kBlock)
classify_kernel(const kr_ray_f64* __restrict__ rays, long long n, double a, unsigned char* __restrict__ strict_mask, int* __restrict__ list_strict,
                unsigned long long* __restrict__ n_strict)
{
    // one ray per work-item: the pass is a 4-field gather over 144-byte records, so it wants every load in flight at once
    const long long i = blockIdx.x * (long long) kBlock + threadIdx.x;
    if (i >= n) return;
    const kr_ray_f64* ray = &rays[i];
    bool strict = false;
    if (ray->steps >= 0) strict = ill_conditioned(ray->k, ray->h, ray->Q, ray->theta, a);       // unused slots (steps == -1) are skipped by either launch
    strict_mask[i] = classify_append(strict, i, list_strict, n_strict);
}

// ---- host side ---------------------------------------------------------------------------------------
// Everything one trace call needs besides the rays -- queue heads and counters, the strict list and mask, timing events, the
// second stream of a split launch -- lives in a Workspace taken from a per-device pool for the duration of the call
// (until the call's last kernel has finished, which the pool learns from an event, not from the host).  Two traces on two
// streams, or from two host threads, therefore never share mutable state: any number may be in flight on one device.
struct Workspace {
    int device = 0;
    int cus = 0;
    unsigned long long* counters = nullptr;      // device: kCounterBlocks x kCounters
    unsigned long long* h_counters = nullptr;    // pinned host copy, filled by an async copy at the end of the call
    int* list = nullptr;                         // device: kListCap ray indices
    unsigned char* mask = nullptr;               // device: one byte per ray
    int64_t mask_capacity = 0;
    std::vector<void*> retired;                  // outgrown masks (workspace_mask_reserve)
    hipStream_t side_stream = nullptr;                   // the second stream of a split trace: belongs to the caller's stream (side_stream_for)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;             // the whole trace, caller's stream
    hipEvent_t ev_strict0 = nullptr, ev_strict1 = nullptr; // strict side (+ overflow) launch, caller's stream
    hipEvent_t ev_main0 = nullptr, ev_main1 = nullptr;     // main launch of a split, side stream
    hipEvent_t ev_classified = nullptr, done = nullptr;
    hipEvent_t ev_in = nullptr;                          // merged batch: the caller's stream of this trace has reached the batch call
    void* d_descs = nullptr;                             // merged batch (held by its first trace): 3 x kMaxBatch descriptors on the device ...
    void* h_descs = nullptr;                             // ... and their pinned staging copy
    bool leased = false;       // a caller holds it (between trace_async and trace_wait / trace_release)
    bool pending = false;      // `done` has been recorded and not yet seen complete
    bool split = false;        // the last call used the split path (ev_strict*/ev_main* are valid)
    int64_t n = 0;
};

std::mutex g_mu;
std::vector<Workspace*> g_pool[64];
constexpr size_t kMaxPool = 512;
constexpr int kMaxBatch = 256;                           // traces one merged batch can hold
constexpr int kMaxMultiGrid = 32768;                     // single-wave workgroups of a merged main launch (wave -> trace table entries)

// The second stream of a split trace is a property of the CALLER's stream, not of the call: traces issued on one stream run one after
// the other anyway, so they can share it, and a driver with 8 streams and 30 tickets outstanding then holds 16 streams, not 38 --
// beyond ~16 streams in a process the strict side launches slow down (profiles/r02_hw_queues.txt).
// It must not share a hardware queue with the caller's stream, or the two launches of a split serialise (seen once a process also
// holds RCCL's streams: HIP multiplexes streams onto a few queues per priority level).  A different priority level has queues of
// its own; the main launch it carries is also the one that may wait.
// One table per device: every caller stream that has run a split trace has an entry; at most kMaxSideStreams DISTINCT side streams exist per
// device -- a process that keeps making streams shares them (chosen by a hash of the caller's stream), each sharer with an entry of its own, so that
// a side stream is counted by everyone who may launch on it.  kr_stream_destroy / side_stream_forget drops the caller's entry and destroys the side
// stream when its last user has gone; kr_shutdown destroys the rest.
constexpr size_t kMaxSideStreams = 16;
std::map<hipStream_t, hipStream_t> g_side_streams[64];      // caller's stream -> side stream
std::map<hipStream_t, int> g_side_users[64];                // side stream -> number of entries above that point at it

int side_stream_for(int dev, hipStream_t user, hipStream_t* out)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto& table = g_side_streams[dev];
    auto& users = g_side_users[dev];
    auto it = table.find(user);
    if (it == table.end()) {
        hipStream_t s = nullptr;
        if (users.size() >= kMaxSideStreams) {             // share one of this device's
            auto pick = users.begin();
            std::advance(pick, (size_t) (((uintptr_t) user) >> 8) % users.size());
            s = pick->first;
        } else {
            int least = 0, greatest = 0;
            KR_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            KR_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least));
        }
        ++users[s];
        it = table.emplace(user, s).first;
    }
    *out = it->second;
    return KR_OK;
}

void workspace_destroy(Workspace* w)
{
    if (!w) return;
    if (w->counters) (void) hipFree(w->counters);
    if (w->h_counters) (void) hipHostFree(w->h_counters);
    if (w->list) (void) hipFree(w->list);
    if (w->mask) (void) hipFree(w->mask);
    if (w->d_descs) (void) hipFree(w->d_descs);
    if (w->h_descs) (void) hipHostFree(w->h_descs);
    for (void* old : w->retired) (void) hipFree(old);
    for (hipEvent_t e : {w->ev0, w->ev1, w->ev_strict0, w->ev_strict1, w->ev_main0, w->ev_main1, w->ev_classified, w->done, w->ev_in})
        if (e) (void) hipEventDestroy(e);
    delete w;
}

int workspace_create(int dev, Workspace** out)
{
    Workspace* w = new Workspace();
    w->device = dev;
    auto fail = [&](int rc) { workspace_destroy(w); return rc; };
#define KR_WS(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(kr::hip_fail(e__, #call, __FILE__, __LINE__)); } while (0)
    KR_WS(hipMalloc((void**) &w->counters, kCounterBlocks * kCounters * sizeof(unsigned long long)));
    KR_WS(hipHostMalloc((void**) &w->h_counters, kCounterBlocks * kCounters * sizeof(unsigned long long), hipHostMallocDefault));
    KR_WS(hipMalloc((void**) &w->list, kListCap * sizeof(int)));
    KR_WS(hipEventCreate(&w->ev0));
    KR_WS(hipEventCreate(&w->ev1));
    KR_WS(hipEventCreate(&w->ev_strict0));
    KR_WS(hipEventCreate(&w->ev_strict1));
    KR_WS(hipEventCreate(&w->ev_main0));
    KR_WS(hipEventCreate(&w->ev_main1));
    KR_WS(hipEventCreateWithFlags(&w->ev_classified, hipEventDisableTiming));
    KR_WS(hipEventCreateWithFlags(&w->done, hipEventDisableTiming));
    KR_WS(hipEventCreateWithFlags(&w->ev_in, hipEventDisableTiming));
    hipDeviceProp_t prop;
    KR_WS(hipGetDeviceProperties(&prop, dev));
#undef KR_WS
    w->cus = prop.multiProcessorCount;
    *out = w;
    return KR_OK;
}

// The per-ray launch selector of a split trace grows geometrically and never frees in the launch path: hipFree synchronises the whole
// device, i.e. every trace in flight on every stream would stall whenever a pooled workspace met a larger n (the returning-radiation
// radii differ in ray count).  The outgrown buffer is parked on the workspace and released by kr_shutdown; the parked
// bytes of a workspace sum to less than its current capacity.
int workspace_mask_reserve(Workspace* ws, int64_t n)
{
    if (ws->mask_capacity >= n) return KR_OK;
    const int64_t want = std::max<int64_t>(n, ws->mask_capacity + ws->mask_capacity / 2);
    unsigned char* grown = nullptr;
    KR_HIP(hipMalloc((void**) &grown, (size_t) want));
    if (ws->mask) ws->retired.push_back(ws->mask);
    ws->mask = grown;
    ws->mask_capacity = want;
    return KR_OK;
}

// takes an idle workspace of the current device out of the pool (or makes one); the caller owns it until workspace_release
int workspace_acquire(Workspace** out)
{
    int dev = 0;
    KR_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("device ordinal out of range"); return KR_EINVAL; }
    Workspace* wait_for = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (Workspace* w : g_pool[dev]) {
            if (w->leased) continue;
            if (w->pending) {
                if (hipEventQuery(w->done) != hipSuccess) { (void) hipGetLastError(); if (!wait_for) wait_for = w; continue; }
                w->pending = false;
            }
            w->leased = true;
            *out = w;
            return KR_OK;
        }
        if (g_pool[dev].size() < kMaxPool) {
            Workspace* w = nullptr;
            const int rc = workspace_create(dev, &w);
            if (rc != KR_OK) return rc;
            w->leased = true;
            g_pool[dev].push_back(w);
            *out = w;
            return KR_OK;
        }
        if (!wait_for) { set_error("kr_trace: too many trace tickets outstanding on this device (kr_trace_wait releases them)"); return KR_EINVAL; }
        wait_for->leased = true;          // ours from here on; its last call is still running
    }
    KR_HIP(hipEventSynchronize(wait_for->done));
    wait_for->pending = false;
    *out = wait_for;
    return KR_OK;
}

void workspace_release(Workspace* w)
{
    std::lock_guard<std::mutex> lk(g_mu);
    w->leased = false;
}

template <typename T>
TraceConsts<T> make_consts(const kr_params* p, int steplim)
{
    TraceConsts<T> c;
    c.a = (T) p->spin; c.horizon = (T) p->horizon; c.rlim = (T) p->r_max; c.thetalim = (T) p->theta_max;
    c.precision = (T) p->precision; c.theta_precision = (T) p->theta_precision;
    c.max_tstep = (T) p->max_tstep; c.maxtstep_rlim = (T) p->maxtstep_rlim; c.max_phistep = (T) p->max_phistep;
    c.tol = (T) p->rk45_tol;
    c.sp0 = (T) p->stop_params[0]; c.sp1 = (T) p->stop_params[1]; c.sp2 = (T) p->stop_params[2];
    c.inv_precision = (T) (1.0 / p->precision);
    c.inv_theta_precision = (T) (1.0 / p->theta_precision);
    c.rk45_extrapolate = !(p->flags & KR_FLAG_RK45_ITERATE_ALL);
    {
        // div_by_uniform (kr_device.hpp) needs a finite, normal divisor with a normal reciprocal and a significand that is not all ones
        auto qualifies = [](double b) {
            if (!(std::fabs(b) >= 1e-300 && std::fabs(b) <= 1e300)) return false;
            int e;
            const double m = std::frexp(std::fabs(b), &e);          // m in [0.5, 1)
            return m != 1.0 - std::ldexp(1.0, -53);
        };
        c.inv_ok = qualifies(p->precision) && qualifies(p->theta_precision);
    }
    c.steplim = steplim;
    c.stop_kind = p->stop_kind;
    const T inf = std::numeric_limits<T>::infinity();
    c.theta_lo = c.thetalim < 0 ? std::fabs(c.thetalim) : -inf;
    c.theta_hi = c.thetalim > 0 ? c.thetalim : (c.thetalim <= 0 ? inf : -inf);
    c.tstep_rlim_eff = c.max_tstep > 0 ? c.maxtstep_rlim : -inf;
    {
        unsigned long long bits;
        const double mt = (double) p->max_tstep;
        std::memcpy(&bits, &mt, sizeof bits);
        c.tstep_lo = (uint32_t) bits; c.tstep_on_hi = (uint32_t) (bits >> 32); c.tstep_off_hi = 0x7FE00000u;
    }
    c.phistep_eff = c.max_phistep > 0 ? c.max_phistep : inf;
    return c;
}

constexpr int64_t kIsolateMinRays = 1 << 18;      // below this a strict launch is too short for the split to pay

struct ListArgs {
    const int* list = nullptr;                    // ray indices, or null for 0 .. n
    const unsigned long long* n_ptr = nullptr;    // item count in device memory, or null (use n)
    int n_mode = 0;                               // how n_ptr is applied (trace_kernel)
    const unsigned char* mask = nullptr;          // per-ray launch selector, or null
    int mask_want = 0;
    int fixed_grid = 0;                           // > 0: launch exactly this many workgroups
};

template <typename T, int METHOD, bool USE_DEST, bool FAST, bool HOG = false>
int launch(typename RayOf<T>::type* rays, int64_t n, const TraceConsts<T>& c, unsigned long long* counters, int cus,
           hipStream_t stream, int max_blocks_per_cu, ListArgs la = ListArgs())
{
    constexpr int kRefill = KR_REFILL_MIN;
    auto kern = trace_kernel<T, METHOD, USE_DEST, FAST, HOG, kRefill>;
    // occupancy of each instance is a property of the code object: asked once per process
    static std::atomic<int> occ{0};                 // (host threads may race here: both would store the same value)
    int blocks_per_cu = occ.load(std::memory_order_relaxed);
    if (blocks_per_cu == 0) {
        int v = 0;
        KR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, kern, kTraceBlock, 0));
        blocks_per_cu = v < 1 ? 1 : v;
        occ.store(blocks_per_cu, std::memory_order_relaxed);
    }
    // Resident workgroups per CU (= waves per SIMD).  The launch ends with its longest ray, which advances one step
    // per turn of its wave: with w waves per SIMD that turn comes round ~w times slower, while throughput keeps
    // improving up to ~3 waves.  Measured on MI355X, RK4 f64 strict, kernel ms at 1 / 2 / 3 workgroups per CU:
    //   PointSource 1e7 rays (longest ray 39 280 steps)  211 / 165 / 183      PointSource 3e7 rays   - / 455 / 432
    //   ImagePlane 4097^2 rays (longest ~2 000 steps)     480 / 344 / 318      fast-math 1e7 rays   176 / 121 / 110
    // Default: 3 when the launch is long enough for throughput to dominate (n >= 2e7, or fast-math with n >= 5e6),
    // else 2.  kr_params.flags bits 8..11 (KR_FLAG_BLOCKS_PER_CU) or the KR_BLOCKS_PER_CU environment variable override.
    int want = max_blocks_per_cu > 0 ? max_blocks_per_cu : (METHOD == KR_EULER && FAST) ? 4 : ((n >= 20000000 || (FAST && n >= 5000000)) ? 3 : 2);
    if (const char* e = getenv("KR_BLOCKS_PER_CU")) {
        const int v = atoi(e);
        if (v >= 1) want = v;
    }
    want *= 4;                                  // `want` counts waves per SIMD; a workgroup is one wave, a CU has four SIMDs
    if (want < blocks_per_cu) blocks_per_cu = want;
    const int64_t resident = (int64_t) cus * blocks_per_cu;
    const int64_t wanted = (n + kTraceBlock - 1) / kTraceBlock;
    int grid = (int) std::max<int64_t>(1, std::min(resident, wanted));
    if (la.fixed_grid > 0) grid = la.fixed_grid;
    if (!HOG && la.list) { set_error("kr_trace: only side launches work from a list"); return KR_EINVAL; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kTraceBlock), 0, stream, rays, (long long) n, c, counters, la.list, la.n_ptr, la.n_mode, la.mask, la.mask_want);
    KR_HIP(hipGetLastError());
    return KR_OK;
}

// one trace launch of the requested flavour (double only): FAST over `la` on `stream`, or strict HOG over `la`
template <bool FAST, bool HOG>
int launch_f64(const kr_params* p, kr_ray_f64* rays, int64_t n, const TraceConsts<double>& c, unsigned long long* counters, int cus,
               hipStream_t stream, int mb, const ListArgs& la)
{
    const bool dest = (p->stop_kind != KR_STOP_THETA);
    switch (p->integrator) {
        case KR_EULER: return launch<double, KR_EULER, false, FAST, HOG>(rays, n, c, counters, cus, stream, mb, la);
        case KR_RK4:
            return dest ? launch<double, KR_RK4, true, FAST, HOG>(rays, n, c, counters, cus, stream, mb, la)
                        : launch<double, KR_RK4, false, FAST, HOG>(rays, n, c, counters, cus, stream, mb, la);
        default:
            return dest ? launch<double, KR_RK45, true, FAST, HOG>(rays, n, c, counters, cus, stream, mb, la)
                        : launch<double, KR_RK45, false, FAST, HOG>(rays, n, c, counters, cus, stream, mb, la);
    }
}

// the same two for a whole batch of traces in ONE launch (trace_multi_kernel): `grid` single-wave workgroups, wave -> trace by table or modulo
template <typename T, int METHOD, bool USE_DEST, bool FAST, bool HOG>
int launch_multi(const TraceDesc<T>* d_descs, int n_desc, const int* d_wave_trace, int grid, hipStream_t stream)
{
    constexpr int kRefill = KR_REFILL_MIN;
    auto kern = trace_multi_kernel<T, METHOD, USE_DEST, FAST, HOG, kRefill>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kTraceBlock), 0, stream, d_descs, n_desc, d_wave_trace);
    KR_HIP(hipGetLastError());
    return KR_OK;
}

template <bool FAST, bool HOG>
int launch_multi_f64(int integrator, bool dest, const TraceDesc<double>* d, int n_desc, const int* wt, int grid, hipStream_t stream)
{
    switch (integrator) {
        case KR_EULER: return launch_multi<double, KR_EULER, false, FAST, HOG>(d, n_desc, wt, grid, stream);
        case KR_RK4:
            return dest ? launch_multi<double, KR_RK4, true, FAST, HOG>(d, n_desc, wt, grid, stream)
                        : launch_multi<double, KR_RK4, false, FAST, HOG>(d, n_desc, wt, grid, stream);
        default:
            return dest ? launch_multi<double, KR_RK45, true, FAST, HOG>(d, n_desc, wt, grid, stream)
                        : launch_multi<double, KR_RK45, false, FAST, HOG>(d, n_desc, wt, grid, stream);
    }
}

// The split trace: classify -> strict HOG launch over the (listed) ill-conditioned rays, on the caller's stream, first, so that
// its workgroups are placed while the chip is still empty  ||  main launch over all other rays on the workspace's side
// stream, filling what is left (the other way round the main launch takes every SIMD's registers and the strict one waits).
// fast_main: the main launch uses the fast arithmetic (KR_FLAG_HYBRID); otherwise it is the strict kernel too, i.e. the
// results are those of one strict launch, bit for bit, and only the placement of the long rays differs.
// Nothing here waits for the device: how many rays were flagged stays in device memory (counters block 3, word 1) and the
// launches read it there.  The side launch is sized for the worst case the list can hold (workgroups that find the queue
// empty leave at once); a source made mostly of ill-conditioned rays (all rays in one meridional plane, say) overflows the
// list, and the overflow -- mask value 2 -- is traced by a third, ordinary-occupancy strict launch that is a no-op otherwise
// (its workgroups read the count and leave).
int split_front(const kr_params* p, kr_ray_f64* rays, int64_t n, int steplim, Workspace* ws, hipStream_t stream)
{
    if (n > 0x7fffffff) { set_error("kr_trace: the split path indexes rays with 32 bits"); return KR_EINVAL; }
    {
        const int rc = workspace_mask_reserve(ws, n);
        if (rc != KR_OK) return rc;
    }
    unsigned long long* split_words = ws->counters + 3 * kCounters;     // [1] n_strict (zeroed by the caller's memset)
    const TraceConsts<double> c = make_consts<double>(p, steplim);
    const int cgrid = (int) ((n + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(classify_kernel, dim3(cgrid), dim3(kBlock), 0, stream, rays, (long long) n, p->spin, ws->mask, ws->list, split_words + 1);
    KR_HIP(hipGetLastError());
    KR_HIP(hipEventRecord(ws->ev_classified, stream));
    KR_HIP(hipEventRecord(ws->ev_strict0, stream));
    // strict side launch: one wave, alone on its SIMD, per 64 listed rays, on at most half of the chip
    ListArgs strict_la;
    strict_la.list = ws->list;
    strict_la.n_ptr = split_words + 1;
    strict_la.n_mode = 1;
    const int64_t list_max = std::min<int64_t>(n, kListCap);
    strict_la.fixed_grid = (int) std::max<int64_t>(1, std::min<int64_t>((list_max + kTraceBlock - 1) / kTraceBlock, (int64_t) (ws->cus / 2) * 4));
    const int rc = launch_f64<false, true>(p, rays, list_max, c, ws->counters + kCounters, ws->cus, stream, 1, strict_la);
    if (rc != KR_OK) return rc;
    KR_HIP(hipEventRecord(ws->ev_strict1, stream));
    ws->split = true;
    return KR_OK;
}

int split_back(const kr_params* p, kr_ray_f64* rays, int64_t n, int steplim, Workspace* ws, hipStream_t stream, bool fast_main)
{
    unsigned long long* split_words = ws->counters + 3 * kCounters;
    const TraceConsts<double> c = make_consts<double>(p, steplim);
    const int mb = KR_FLAG_GET_BLOCKS_PER_CU(p->flags);
    // main launch
    KR_HIP(hipStreamWaitEvent(ws->side_stream, ws->ev_classified, 0));
    KR_HIP(hipEventRecord(ws->ev_main0, ws->side_stream));
    ListArgs main_la;
    main_la.mask = ws->mask;
    main_la.mask_want = 0;
    const int main_waves = mb ? mb : (fast_main && p->integrator == KR_EULER) ? 4 : 3;      // resident waves per SIMD of the main launch
    int rc = fast_main ? launch_f64<true, false>(p, rays, n, c, ws->counters, ws->cus, ws->side_stream, main_waves, main_la)
                       : launch_f64<false, false>(p, rays, n, c, ws->counters, ws->cus, ws->side_stream, main_waves, main_la);
    if (rc != KR_OK) return rc;
    // strict overflow launch (mask == 2): only has work when more than kListCap rays were flagged, and then the main launch has
    // next to none.  It follows the main launch on the side stream: behind the side launch on the caller's stream its idle
    // workgroups would sit waiting for the main launch's registers (measured: 10 ms of "kernel time" doing nothing).
    if (n > kListCap) {
        ListArgs rest_la;
        rest_la.n_ptr = split_words + 1;
        rest_la.n_mode = 2;
        rest_la.mask = ws->mask;
        rest_la.mask_want = 2;
        rc = launch_f64<false, false>(p, rays, n, c, ws->counters + 2 * kCounters, ws->cus, ws->side_stream, mb, rest_la);
        if (rc != KR_OK) return rc;
    }
    KR_HIP(hipEventRecord(ws->ev_main1, ws->side_stream));
    KR_HIP(hipStreamWaitEvent(stream, ws->ev_main1, 0));
    return KR_OK;
}

template <typename T>
int dispatch(const kr_params* p, void* d_rays, int64_t n, int steplim, unsigned long long* counters, int cus, hipStream_t stream)
{
    ListArgs la;
    using R = typename RayOf<T>::type;
    R* rays = (R*) d_rays;
    const TraceConsts<T> c = make_consts<T>(p, steplim);
    const bool dest = (p->stop_kind != KR_STOP_THETA);
    const int mb = KR_FLAG_GET_BLOCKS_PER_CU(p->flags);
    if constexpr (std::is_same<T, double>::value) {
        if (p->flags & KR_FLAG_FAST_MATH) {
            switch (p->integrator) {
                case KR_EULER: return launch<T, KR_EULER, false, true>(rays, n, c, counters, cus, stream, mb, la);
                case KR_RK4:
                    return dest ? launch<T, KR_RK4, true, true>(rays, n, c, counters, cus, stream, mb, la)
                                : launch<T, KR_RK4, false, true>(rays, n, c, counters, cus, stream, mb, la);
                default:
                    return dest ? launch<T, KR_RK45, true, true>(rays, n, c, counters, cus, stream, mb, la)
                                : launch<T, KR_RK45, false, true>(rays, n, c, counters, cus, stream, mb, la);
            }
        }
    }
    switch (p->integrator) {
        case KR_EULER: return launch<T, KR_EULER, false, false>(rays, n, c, counters, cus, stream, mb, la);
        case KR_RK4:
            return dest ? launch<T, KR_RK4, true, false>(rays, n, c, counters, cus, stream, mb, la)
                        : launch<T, KR_RK4, false, false>(rays, n, c, counters, cus, stream, mb, la);
        default:
            return dest ? launch<T, KR_RK45, true, false>(rays, n, c, counters, cus, stream, mb, la)
                        : launch<T, KR_RK45, false, false>(rays, n, c, counters, cus, stream, mb, la);
    }
}

int validate(const kr_params* p, void* d_rays, int64_t n)
{
    if (p && n > 0 && !d_rays) { set_error("kr_trace: null argument or negative n"); return KR_EINVAL; }
    if (!p || n < 0) { set_error("kr_trace: null argument or negative n"); return KR_EINVAL; }
    if (p->integrator < KR_EULER || p->integrator > KR_RK45) { set_error("kr_trace: unknown integrator"); return KR_EINVAL; }
    if (p->stop_kind < KR_STOP_THETA || p->stop_kind > KR_STOP_FLATPLANE) { set_error("kr_trace: unknown stop_kind"); return KR_EINVAL; }
    if (p->stop_kind != KR_STOP_THETA && p->integrator == KR_EULER) {
        // assert(method != Integrator::Euler), raytracer.cpp:983
        set_error("kr_trace: Integrator::Euler does not support RayDestination stopping conditions");
        return KR_EINVAL;
    }
    return require_device();
}

// One trace is enqueued in two halves, so that a batch of traces can put ALL its front halves on the device before any back half:
//   front: counters zeroed, [classification + strict side launch] -- a handful of waves that want SIMDs of their own and carry the
//          launch's longest rays;        back: the main launch (and the overflow launch), counters copied out, `done` recorded.
// A lone kr_trace_async runs both at once.  In a batch (kr_trace_batch_async: one launch per tolerance, per source radius, ...)
// every side launch is placed while the chip is still empty; enqueued trace by trace, the main launch of trace k would own every
// SIMD's registers by the time the side launch of trace k+1 asks for them (measured: 18 RK45 sweep points in 1.6 s instead of 0.6 s).
struct Pending {
    const kr_params* p = nullptr;
    void* d_rays = nullptr;
    int64_t n = 0;
    hipStream_t stream = nullptr;
    bool f32 = false, hybrid = false, split = false;
    int steplim = 0;
    Workspace* ws = nullptr;
};

int trace_front(Pending& t, bool batch)
{
    int rc = validate(t.p, t.d_rays, t.n);
    if (rc != KR_OK) return rc;
    if (t.n == 0) return KR_OK;
    // effective_steplim, raytracer.cpp:80
    t.steplim = (t.p->steplim > 0) ? t.p->steplim : (t.p->integrator == KR_RK45) ? KR_RK45_STEPLIM : KR_STEPLIM;
    rc = workspace_acquire(&t.ws);
    if (rc != KR_OK) return rc;
    Workspace* ws = t.ws;
    ws->split = false;
    ws->n = t.n;
    t.hybrid = !t.f32 && (t.p->flags & KR_FLAG_HYBRID) && !(t.p->flags & KR_FLAG_FAST_MATH);
    // all-strict launches of some size isolate their ill-conditioned (in the lamp-post workloads: longest) rays the same way:
    // identical results, no tail; in a batch every strict launch does, whatever its size (its tail is what the batch overlaps).
    // KR_NO_ISOLATE=1 keeps the single launch (A/B and bit-identity tests).
    const bool isolate = !t.f32 && !t.hybrid && !(t.p->flags & KR_FLAG_FAST_MATH) && (t.n >= kIsolateMinRays || (batch && t.n >= 4096)) && !getenv("KR_NO_ISOLATE");
    t.split = t.hybrid || isolate;
    KR_HIP(hipMemsetAsync(ws->counters, 0, kCounterBlocks * kCounters * sizeof(unsigned long long), t.stream));
    KR_HIP(hipEventRecord(ws->ev0, t.stream));
    if (t.split) {
        rc = side_stream_for(ws->device, t.stream, &ws->side_stream);
        if (rc != KR_OK) return rc;
        return split_front(t.p, (kr_ray_f64*) t.d_rays, t.n, t.steplim, ws, t.stream);
    }
    return KR_OK;
}

int trace_back(Pending& t)
{
    if (t.n == 0 || !t.ws) return KR_OK;
    Workspace* ws = t.ws;
    int r = t.f32 ? dispatch<float>(t.p, t.d_rays, t.n, t.steplim, ws->counters, ws->cus, t.stream)
                  : t.split ? split_back(t.p, (kr_ray_f64*) t.d_rays, t.n, t.steplim, ws, t.stream, t.hybrid)
                            : dispatch<double>(t.p, t.d_rays, t.n, t.steplim, ws->counters, ws->cus, t.stream);
    if (r != KR_OK) return r;
    KR_HIP(hipEventRecord(ws->ev1, t.stream));
    KR_HIP(hipMemcpyAsync(ws->h_counters, ws->counters, kCounterBlocks * kCounters * sizeof(unsigned long long), hipMemcpyDeviceToHost, t.stream));
    KR_HIP(hipEventRecord(ws->done, t.stream));
    return KR_OK;
}

// After a failure part-way: whatever was enqueued must drain before the workspace is reused.  The kernels of a split trace sit on the
// caller's stream AND its side stream, those of a merged batch on the batch's primary stream whatever t.stream is -- an event on
// t.stream alone would not cover them -- so the (rare) error path simply waits for the device before it lets the workspace go.
void abandon(Pending& t)
{
    if (!t.ws) return;
    (void) hipDeviceSynchronize();
    (void) hipGetLastError();
    {
        std::lock_guard<std::mutex> lk(g_mu);
        t.ws->pending = false;
    }
    workspace_release(t.ws);
    t.ws = nullptr;
}

void hand_over(Pending& t, void** ticket)
{
    if (t.ws) {
        std::lock_guard<std::mutex> lk(g_mu);
        t.ws->pending = true;
    }
    *ticket = t.ws;
}

}  // namespace

// Enqueues one trace on `stream` and returns at once; *ticket (never null on success, unless n == 0) must go to trace_wait or
// trace_release.  Nothing in here synchronises with the device.
int trace_async(const kr_params* p, void* d_rays, int64_t n, hipStream_t stream, bool f32, void** ticket)
{
    *ticket = nullptr;
    Pending t;
    t.p = p; t.d_rays = d_rays; t.n = n; t.stream = stream; t.f32 = f32;
    int rc = trace_front(t, false);
    if (rc == KR_OK) rc = trace_back(t);
    if (rc != KR_OK) { abandon(t); return rc; }
    hand_over(t, ticket);
    return KR_OK;
}

// A batch whose traces all run the same kernel instances (same integrator, same kind of stop surface, same arithmetic mode) is
// MERGED: one classification per trace, then ONE side launch, ONE main launch (and one overflow launch) over all of them
// (trace_multi_kernel), on the first trace's stream and its second stream; every other trace's stream waits for the batch at both ends.
int merged_batch(std::vector<Pending>& ts, bool hybrid)
{
    const int count = (int) ts.size();
    const kr_params* p0 = ts[0].p;
    const bool dest = p0->stop_kind != KR_STOP_THETA;
    hipStream_t primary = ts[0].stream;
    for (auto& t : ts) {
        t.steplim = (t.p->steplim > 0) ? t.p->steplim : (t.p->integrator == KR_RK45) ? KR_RK45_STEPLIM : KR_STEPLIM;
        int rc = workspace_acquire(&t.ws);
        if (rc != KR_OK) return rc;
        t.ws->split = true;
        t.ws->n = t.n;
        t.hybrid = hybrid;
        t.split = true;
    }
    Workspace* w0 = ts[0].ws;
    hipStream_t side = nullptr;
    int rc = side_stream_for(w0->device, primary, &side);
    if (rc != KR_OK) return rc;
    const size_t desc_bytes = sizeof(TraceDesc<double>);
    const size_t table_off = 3 * kMaxBatch * desc_bytes;                   // [side descs | main descs | overflow descs | wave -> trace table of the main launch]
    const size_t staging_bytes = table_off + kMaxMultiGrid * sizeof(int);
    if (!w0->d_descs) {
        KR_HIP(hipMalloc(&w0->d_descs, staging_bytes));
        KR_HIP(hipHostMalloc(&w0->h_descs, staging_bytes, hipHostMallocDefault));
    }
    // inputs: whatever the callers enqueued on the traces' own streams (their ray sources) comes first
    for (auto& t : ts)
        if (t.stream != primary) {
            KR_HIP(hipEventRecord(t.ws->ev_in, t.stream));
            KR_HIP(hipStreamWaitEvent(primary, t.ws->ev_in, 0));
        }
    TraceDesc<double>* h = (TraceDesc<double>*) w0->h_descs;
    TraceDesc<double>* hog = h, * mainv = h + kMaxBatch, * rest = h + 2 * kMaxBatch;
    int n_rest = 0;
    int64_t hog_max = 1, main_waves = 0, n_total = 0;
    for (int i = 0; i < count; i++) {
        Pending& t = ts[i];
        Workspace* ws = t.ws;
        ws->side_stream = side;
        if (t.n > 0x7fffffff) { set_error("kr_trace: the split path indexes rays with 32 bits"); return KR_EINVAL; }
        rc = workspace_mask_reserve(ws, t.n);
        if (rc != KR_OK) return rc;
        KR_HIP(hipMemsetAsync(ws->counters, 0, kCounterBlocks * kCounters * sizeof(unsigned long long), primary));
        KR_HIP(hipEventRecord(ws->ev0, primary));
        unsigned long long* split_words = ws->counters + 3 * kCounters;
        const int cgrid = (int) ((t.n + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(classify_kernel, dim3(cgrid), dim3(kBlock), 0, primary, (const kr_ray_f64*) t.d_rays, (long long) t.n, t.p->spin, ws->mask, ws->list, split_words + 1);
        KR_HIP(hipGetLastError());
        const TraceConsts<double> c = make_consts<double>(t.p, t.steplim);
        const int64_t list_max = std::min<int64_t>(t.n, kListCap);
        hog[i] = TraceDesc<double>{(kr_ray_f64*) t.d_rays, (long long) list_max, c, ws->counters + kCounters, ws->list, split_words + 1, nullptr, 1, 0};
        mainv[i] = TraceDesc<double>{(kr_ray_f64*) t.d_rays, (long long) t.n, c, ws->counters, nullptr, nullptr, ws->mask, 0, 0};
        if (t.n > kListCap) rest[n_rest++] = TraceDesc<double>{(kr_ray_f64*) t.d_rays, (long long) t.n, c, ws->counters + 2 * kCounters, nullptr, split_words + 1, ws->mask, 2, 2};
        hog_max = std::max<int64_t>(hog_max, (list_max + kTraceBlock - 1) / kTraceBlock);
        main_waves += (t.n + kTraceBlock - 1) / kTraceBlock;
        n_total += t.n;
    }
    // main launch: twice what is resident (3 waves per SIMD), never more waves than 64-ray loads; wave -> trace in proportion to the
    // traces' ray counts, interleaved so that the first waves to be placed cover every trace
    const int mb = KR_FLAG_GET_BLOCKS_PER_CU(p0->flags);
    const int64_t resident = (int64_t) w0->cus * 4 * (mb ? mb : (hybrid && p0->integrator == KR_EULER) ? 4 : 3);
    const int main_grid = (int) std::max<int64_t>(count, std::min<int64_t>({main_waves, 2 * resident, (int64_t) kMaxMultiGrid}));
    int* wave_trace = (int*) ((char*) w0->h_descs + table_off);
    {
        std::vector<std::pair<double, int>> order;
        order.reserve((size_t) main_grid + (size_t) count);
        for (int i = 0; i < count; i++) {
            const int64_t q = std::max<int64_t>(1, (int64_t) ((double) (main_grid - count) * (double) ts[i].n / (double) n_total) + 1);
            for (int64_t k = 0; k < q; k++) order.emplace_back(((double) k + 0.5) / (double) q + 1e-9 * i, i);
        }
        std::sort(order.begin(), order.end());
        for (int b = 0; b < main_grid; b++) wave_trace[b] = order[(size_t) b % order.size()].second;     // (quotas sum to main_grid up to rounding)
    }
    KR_HIP(hipMemcpyAsync(w0->d_descs, w0->h_descs, staging_bytes, hipMemcpyHostToDevice, primary));
    const TraceDesc<double>* d = (const TraceDesc<double>*) w0->d_descs;
    const int* d_wave_trace = (const int*) ((const char*) w0->d_descs + table_off);
    KR_HIP(hipEventRecord(w0->ev_classified, primary));
    for (auto& t : ts) KR_HIP(hipEventRecord(t.ws->ev_strict0, primary));
    // side launch (wave -> trace by modulo): as many waves per trace as its list could need, but no more than fit on the chip at once
    // in total (1024 SIMDs; at least 4 per trace) -- a wave refills from its trace's list, so fewer waves only mean more rays per wave,
    // whereas thousands of exclusive single-wave workgroups that find nothing to do still have to be placed one by one
    const int64_t hog_per_trace = std::max<int64_t>(4, std::min<int64_t>(hog_max, (4 * (int64_t) w0->cus) / count));            // waves
    rc = launch_multi_f64<false, true>(p0->integrator, dest, d, count, nullptr, (int) (hog_per_trace * count), primary);
    if (rc != KR_OK) return rc;
    for (auto& t : ts) KR_HIP(hipEventRecord(t.ws->ev_strict1, primary));
    KR_HIP(hipStreamWaitEvent(side, w0->ev_classified, 0));
    for (auto& t : ts) KR_HIP(hipEventRecord(t.ws->ev_main0, side));
    rc = hybrid ? launch_multi_f64<true, false>(p0->integrator, dest, d + kMaxBatch, count, d_wave_trace, main_grid, side)
                : launch_multi_f64<false, false>(p0->integrator, dest, d + kMaxBatch, count, d_wave_trace, main_grid, side);
    if (rc != KR_OK) return rc;
    if (n_rest > 0) {
        // overflow launch: a no-op unless some trace flagged more rays than its list holds (then: few persistent waves per such trace)
        rc = launch_multi_f64<false, false>(p0->integrator, dest, d + 2 * kMaxBatch, n_rest, nullptr, (int) std::min<int64_t>((int64_t) n_rest * 64, resident), side);
        if (rc != KR_OK) return rc;
    }
    for (auto& t : ts) KR_HIP(hipEventRecord(t.ws->ev_main1, side));
    KR_HIP(hipStreamWaitEvent(primary, ts.back().ws->ev_main1, 0));      // the last one recorded: every ev_main1 has completed by then
    for (auto& t : ts) {
        Workspace* ws = t.ws;
        KR_HIP(hipEventRecord(ws->ev1, primary));
        KR_HIP(hipMemcpyAsync(ws->h_counters, ws->counters, kCounterBlocks * kCounters * sizeof(unsigned long long), hipMemcpyDeviceToHost, primary));
        KR_HIP(hipEventRecord(ws->done, primary));
        if (t.stream != primary) KR_HIP(hipStreamWaitEvent(t.stream, ws->done, 0));       // the caller's next kernels on that stream see the traced rays
    }
    return KR_OK;
}

// The same for `count` traces at once (double precision).  On failure nothing is left outstanding.
int trace_batch_async(int count, const kr_params* const* p, void* const* d_rays, const int64_t* n, void* const* streams, void** tickets)
{
    if (count < 0 || (count > 0 && (!p || !d_rays || !n || !tickets))) { set_error("kr_trace_batch_async: null argument"); return KR_EINVAL; }
    std::vector<Pending> ts((size_t) count);
    for (int i = 0; i < count; i++) {
        tickets[i] = nullptr;
        ts[i].p = p[i]; ts[i].d_rays = d_rays[i]; ts[i].n = n[i]; ts[i].stream = streams ? (hipStream_t) streams[i] : nullptr;
    }
    int rc = KR_OK;
    // mergeable: every trace valid, non-empty, split-capable, and on the same kernel instances
    bool merge = count >= 2 && count <= kMaxBatch && !getenv("KR_NO_MERGED_BATCH") && !getenv("KR_NO_ISOLATE");
    for (int i = 0; i < count && merge; i++) {
        if (validate(p[i], d_rays[i], n[i]) != KR_OK || n[i] < 4096 || (p[i]->flags & KR_FLAG_FAST_MATH)) merge = false;
        else if (p[i]->integrator != p[0]->integrator || (p[i]->stop_kind != KR_STOP_THETA) != (p[0]->stop_kind != KR_STOP_THETA) ||
                 (p[i]->flags & KR_FLAG_HYBRID) != (p[0]->flags & KR_FLAG_HYBRID) || KR_FLAG_GET_BLOCKS_PER_CU(p[i]->flags) != KR_FLAG_GET_BLOCKS_PER_CU(p[0]->flags))
            merge = false;
    }
    if (merge) {
        rc = merged_batch(ts, (p[0]->flags & KR_FLAG_HYBRID) != 0);
        if (rc != KR_OK) {
            for (auto& t : ts) abandon(t);
            return rc;
        }
        for (int i = 0; i < count; i++) hand_over(ts[i], &tickets[i]);
        return KR_OK;
    }
    for (int i = 0; i < count && rc == KR_OK; i++) rc = trace_front(ts[i], true);
    for (int i = 0; i < count && rc == KR_OK; i++) rc = trace_back(ts[i]);
    if (rc != KR_OK) {
        for (auto& t : ts) abandon(t);
        return rc;
    }
    for (int i = 0; i < count; i++) hand_over(ts[i], &tickets[i]);
    return KR_OK;
}

// Waits for the trace behind `ticket`, fills *stats (may be null) and returns the workspace to the pool.
int trace_wait(void* ticket, kr_stats* stats)
{
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (!ticket) return KR_OK;                       // the n == 0 call
    Workspace* ws = (Workspace*) ticket;
    auto body = [&]() -> int {
        KR_HIP(hipEventSynchronize(ws->done));
        if (!stats) return KR_OK;
        const unsigned long long* h2 = ws->h_counters;
        unsigned long long h[kCounters];
        for (int i = 0; i < kCounters; i++) h[i] = h2[i] + h2[kCounters + i] + h2[2 * kCounters + i];
        h[7] = std::max(h2[7], std::max(h2[kCounters + 7], h2[2 * kCounters + 7]));      // a maximum, not a sum
#if KR_OCC_STATS
        for (int b = 0; b < 3; b++) {
            const unsigned long long* q = h2 + b * kCounters;
            if (q[8]) std::fprintf(stderr, "kr_occ: launch %d (0 main, 1 strict side, 2 overflow): steps %llu wave_iters %llu step-loop lane occupancy %.4f | after queue exhaustion: "
                                   "wave_iters %llu (%.2f %%) lane occupancy %.4f | refills %llu lanes/refill %.2f | longest ray %llu steps\n", b, q[2], q[8], (double) q[2] / (64.0 * q[8]), q[9],
                                   100.0 * q[9] / q[8], q[9] ? (double) q[10] / (64.0 * q[9]) : 0.0, q[11], q[11] ? (double) q[12] / q[11] : 0.0, q[7]);
        }
#endif
        stats->rays_total = ws->n;
        stats->rays_strict_side = (int64_t) h2[3 * kCounters + 1];
        stats->rays_traced = (int64_t) h[1];
        stats->steps_total = (int64_t) h[2];
        stats->rk45_attempts = (int64_t) h[3];
        stats->rk45_rejects = (int64_t) h[4];
        stats->rk45_stationary_steps = (int64_t) h[5];
        stats->rk45_extrapolated_steps = (int64_t) h[6];
        stats->longest_ray_steps = (int64_t) h[7];
        stats->longest_ray_steps_strict_side = ws->split ? (int64_t) h2[kCounters + 7] : 0;
        stats->steps_strict_side = ws->split ? (int64_t) h2[kCounters + 2] : 0;
        stats->rk45_evaluated_strict_side = ws->split ? (int64_t) (h2[kCounters + 3] - h2[kCounters + 5] - h2[kCounters + 6]) : 0;
        float ms = 0;
        KR_HIP(hipEventElapsedTime(&ms, ws->ev0, ws->ev1));
        stats->kernel_ms = ms;
        if (ws->split) {
            KR_HIP(hipEventElapsedTime(&ms, ws->ev_strict0, ws->ev_strict1));
            stats->strict_side_ms = ms;
            KR_HIP(hipEventElapsedTime(&ms, ws->ev_main0, ws->ev_main1));
            stats->main_ms = ms;
        }
        return KR_OK;
    };
    const int rc = body();
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (rc == KR_OK) ws->pending = false;
        ws->leased = false;
    }
    return rc;
}

// How far the trace behind `ticket` has come: the number of rays its waves have taken off the work queue so far (what the reference's progress
// counter counts, raytracer.cpp:107-112: a ray is counted when its loop iteration STARTS), and whether the trace has finished.  The queue head lives
// in device memory; it is read with an 8-byte copy on a stream of the library's own (a DMA transfer: it needs no compute unit, so it completes while
// the persistent kernels hold every SIMD), a few microseconds per call.  Does not wait for the trace and does not retire the ticket.
namespace {
std::mutex g_poll_mu;
hipStream_t g_poll_stream[64];
unsigned long long* g_poll_word[64];
}  // namespace

int trace_poll(void* ticket, int64_t* rays_started, int32_t* finished)
{
    if (rays_started) *rays_started = 0;
    if (finished) *finished = 1;
    if (!ticket) return KR_OK;                       // the n == 0 call
    Workspace* ws = (Workspace*) ticket;
    const hipError_t q = hipEventQuery(ws->done);
    if (q == hipSuccess) {
        if (rays_started) *rays_started = ws->n;
        return KR_OK;
    }
    (void) hipGetLastError();
    if (q != hipErrorNotReady) return kr::hip_fail(q, "hipEventQuery(trace)", __FILE__, __LINE__);
    if (finished) *finished = 0;
    int dev = 0;
    KR_HIP(hipGetDevice(&dev));
    if (dev != ws->device) { set_error("kr_trace_poll: the ticket belongs to another device than the current one"); return KR_EINVAL; }
    std::lock_guard<std::mutex> lk(g_poll_mu);
    if (!g_poll_stream[dev]) {
        KR_HIP(hipStreamCreateWithFlags(&g_poll_stream[dev], hipStreamNonBlocking));
        KR_HIP(hipHostMalloc((void**) &g_poll_word[dev], sizeof(unsigned long long), hipHostMallocDefault));
    }
    // block 0 = the main launch (or the only one): its head runs over all n slots, also those whose rays belong to the side launch
    KR_HIP(hipMemcpyAsync(g_poll_word[dev], ws->counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, g_poll_stream[dev]));
    KR_HIP(hipStreamSynchronize(g_poll_stream[dev]));
    if (rays_started) *rays_started = (int64_t) std::min<unsigned long long>(*g_poll_word[dev], (unsigned long long) ws->n);
    return KR_OK;
}

// Gives the ticket back without waiting: the workspace is reused once its trace has finished.
void trace_release(void* ticket)
{
    if (ticket) workspace_release((Workspace*) ticket);
}

// kr_stream_destroy: the side stream that belonged to `user` goes with it (its work has drained: the caller's stream waits for it at the
// end of every split trace, and is synchronised here before it is destroyed).
void side_stream_forget(hipStream_t user)
{
    // (the entry is looked for on every device: the device that is current when a stream is destroyed need not be the one it was used on)
    hipStream_t side = nullptr;
    int side_dev = -1;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (int dev = 0; dev < 64 && side_dev < 0; dev++) {
            auto it = g_side_streams[dev].find(user);
            if (it == g_side_streams[dev].end()) continue;
            hipStream_t s = it->second;
            g_side_streams[dev].erase(it);
            auto u = g_side_users[dev].find(s);
            if (u != g_side_users[dev].end() && --u->second <= 0) {
                g_side_users[dev].erase(u);
                side = s;                                    // its last user: nobody can be handed it any more
                side_dev = dev;
            } else {
                return;                                      // other caller streams still launch on it
            }
        }
    }
    if (!side) return;
    int keep = 0;
    const bool have_dev = hipGetDevice(&keep) == hipSuccess;
    if (hipSetDevice(side_dev) == hipSuccess) {
        (void) hipStreamSynchronize(side);
        (void) hipStreamDestroy(side);
    }
    if (have_dev) (void) hipSetDevice(keep);
    (void) hipGetLastError();
}

// kr_shutdown: waits for the devices this library has used, then gives back every pooled workspace and side stream.  Refused (KR_EINVAL, nothing
// released) while a trace ticket is outstanding: its kr_trace_wait / kr_trace_release would touch a freed workspace.
int trace_shutdown()
{
    int keep = 0;
    const bool have_dev = hipGetDevice(&keep) == hipSuccess;
    (void) hipGetLastError();
    std::lock_guard<std::mutex> lk(g_mu);
    {
        size_t leased = 0;
        for (int dev = 0; dev < 64; dev++)
            for (Workspace* w : g_pool[dev]) leased += w->leased ? 1 : 0;
        if (leased) {
            set_error("kr_shutdown: trace tickets are outstanding (kr_trace_wait / kr_trace_release them first)");
            return KR_EINVAL;
        }
    }
    for (int dev = 0; dev < 64; dev++) {
        if (g_pool[dev].empty() && g_side_users[dev].empty() && !g_poll_stream[dev]) continue;
        if (hipSetDevice(dev) != hipSuccess) { (void) hipGetLastError(); continue; }
        (void) hipDeviceSynchronize();
        for (Workspace* w : g_pool[dev]) workspace_destroy(w);
        g_pool[dev].clear();
        {
            std::lock_guard<std::mutex> pl(g_poll_mu);
            if (g_poll_stream[dev]) { (void) hipStreamDestroy(g_poll_stream[dev]); g_poll_stream[dev] = nullptr; }
            if (g_poll_word[dev]) { (void) hipHostFree(g_poll_word[dev]); g_poll_word[dev] = nullptr; }
        }
        for (auto& kv : g_side_users[dev]) (void) hipStreamDestroy(kv.first);
        g_side_users[dev].clear();
        g_side_streams[dev].clear();
    }
    if (have_dev) (void) hipSetDevice(keep);
    (void) hipGetLastError();
    return KR_OK;
}

int trace_dev(const kr_params* p, void* d_rays, int64_t n, hipStream_t stream, kr_stats* stats, bool f32)
{
    if (stats) { std::memset(stats, 0, sizeof(*stats)); stats->rays_total = n; }
    void* ticket = nullptr;
    const int rc = trace_async(p, d_rays, n, stream, f32, &ticket);
    if (rc != KR_OK) return rc;
    if (!stats) { trace_release(ticket); return KR_OK; }
    const int rc2 = trace_wait(ticket, stats);
    stats->rays_total = n;
    return rc2;
}

}  // namespace kr
