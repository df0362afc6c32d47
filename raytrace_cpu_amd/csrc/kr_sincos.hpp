// kr_sincos.hpp -- compact sin/cos pair for polar angles (see the comment on kr_sincos_f64).
// Compiles both as device code (hipcc) and as plain host C++ (g++), the latter only so that
// tests/test_sincos_accuracy.py can measure its error against long-double libm on the CPU.
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#define KR_SC_FN __device__ __forceinline__
#else
#define KR_SC_FN static inline
#endif

// Horner step p <- fma(z, p, C) as ONE vector instruction.  The compiler keeps the coefficients C resident in VGPR pairs across the
// step loop (good) but then emits the destructive two-operand form, v_mov_b64 tmp, C + v_fmac_f64 tmp, z, p -- two vector
// instructions per polynomial step, ~40 per RK4 step of the fast kernel.  Spelling the three-operand v_fma_f64 out keeps C where it is.
// Same operation, same rounding.
// (Measured and rejected, profiles/r02_ab_experiments.txt: pinning every coefficient to an SGPR pair, or building without v_fmac_f64 -- the SGPR
// spills that follow are reloaded by v_readlane inside the step loop: 94.1 / 93.2 ms against 92.7 / 90.8.)
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ double kr_fma3(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
#else
#define kr_fma3(a, b, c) __builtin_fma((a), (b), (c))
#endif
// the same with the addend in a SCALAR register pair (one SGPR operand is allowed per VALU instruction): for constants that are only live
// for part of a step, so that they need neither a VGPR pair nor a v_mov_b64 to materialise them (kr_device.hpp::sincos_near)
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __forceinline__ double kr_fma3s(double a, double b, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
}
#else
#define kr_fma3s(a, b, c) __builtin_fma((a), (b), (c))
#endif

// sin and cos of a polar angle.  theta stays within a few multiples of pi (it is reflected back into [0, pi]
// after every step and the RK stages move it by a fraction of that), so the general-purpose device sincos
// (Payne-Hanek capable, ~90 VALU instructions, 4 calls per RK4 step = a third of the step) is replaced by:
//   n = rint(x * 2/pi);  r = x - n*pi/2 in two FMA steps (pi/2 = P1 + P2, the first one exact for |n| < 2^10:
//   x and n*P1 are multiples of ulp(P1) and |r| < 1), with the rounding tail of the second step kept;
//   sin/cos of r in [-pi/4, pi/4] from kr_sincos_cr_core_f64 below (strict callers) or plain Horner kernels (the fast
//   arithmetic, kr_sincos_fast_f64); quadrant fix-up by n mod 4.
// |x| >= 1024 (never reached by a healthy ray) and non-finite inputs take the library path.
//
// Small polar angles (|x| < 2^-7) take a branch of their own, kr_sincos_small_f64: no reduction (n = 0, r = x exactly), no
// quadrant logic, Taylor kernels x + x z (S1 + z (S2 + z S3)) and 1 + (z^2 (C2 + z C3) - z/2) with z = x^2 < 6.2e-5: the first
// dropped terms are 3.8e-23 x and 3.4e-22, every intermediate rounding is scaled by <= z, so both results are within
// 0.5001 ulp (tests/test_sincos_accuracy.py).  10 fp64 instructions instead of ~34 fp64 + ~22 integer / select ones.  Why it
// matters: rays that ride the polar axis (the beta = -pi column of a lamp-post PointSource: theta stays at the source's 1e-3
// for 3e4 steps, 1e5 under RK45) are the longest rays of a launch, and a launch cannot end before its longest ray does.
// The choice is a pure function of x, so a ray gets the same bits in whichever kernel / wave it is traced; on the device the
// branch is taken wave-uniformly (all lanes small -> only the short kernel; mixed wave -> both and a select).
#define KR_SMALL_ANGLE_LIMIT 0.0078125      /* 2^-7 */
KR_SC_FN void kr_sincos_small_f64(double x, double& s, double& c)
{
    const double z = x * x;
    const double ps = kr_fma3(z, kr_fma3(z, -1.98412698412698412698e-04, 8.33333333333333333333e-03), -1.66666666666666666667e-01);
    s = __builtin_fma(x * z, ps, x);
    const double pc = kr_fma3(z, -1.38888888888888888889e-03, 4.16666666666666666667e-02);
    c = 1.0 + __builtin_fma(z * z, pc, -0.5 * z);
}

KR_SC_FN void kr_sincos_general_f64(double x, double& s, double& c);

// |x| >= 1024 and non-finite arguments (never reached by a healthy ray): the library routine, OUT OF LINE.  Inlined it is ~1 KB of
// Payne-Hanek code per call site -- 9 KB of the RK45 kernels, 4 KB of the RK4 ones -- and the instruction cache (64 KB per two CUs)
// is what a lone strict wave and the main launch's waves on the same CUs compete for.
#if defined(__HIPCC__)
static __device__ __attribute__((noinline)) void kr_sincos_libm_f64(double x, double* s, double* c) { sincos(x, s, c); }
#else
KR_SC_FN void kr_sincos_libm_f64(double x, double* s, double* c) { sincos(x, s, c); }
#endif

// Correctly rounded in practice: sin and cos of the reduced argument r + y, |r| <= pi/4, with the leading
// terms carried in double-double -- r^3 (S1 + z (S2 + ...)) and z^2 (C2 + z (C3 + ...)) with z = r^2 as an exact product, S1, S2, C2, C3
// as (hi, lo) pairs, Taylor tails (through r^21 / z^10) in plain double -- and ONE final rounding.  The accumulated error before that
// rounding is ~2^-66 of the result: 12e6 random arguments in (-40, 40) all come out correctly rounded (against __float128), which makes
// the strict path differ from glibc only where glibc itself is not correctly rounded (0.01-0.26 % of arguments; the fdlibm kernels
// this replaces: 3-5 %).  ~95 fp64 instructions for the pair instead of ~68: the TwoSum / TwoProd sequences below rely on
// -ffp-contract=off (every fma is spelled out).
// (the pairs sh + sl and ch + cl BEFORE the final rounding: kr_crmath.hpp's tan divides them)
KR_SC_FN void kr_sincos_cr_core_pairs(double r, double y, double& sh_o, double& sl_o, double& ch_o, double& cl_o)
{
    const double zh = r * r, zl = __builtin_fma(r, r, -zh);
    // sin(r + y) = r + y cos r + r^3 (S1 + z (S2 + z T(z)))
    double T = kr_fma3(zh, 1.9572941063391263e-20, -8.22063524662433e-18);
    T = kr_fma3(zh, T, 2.8114572543455206e-15);
    T = kr_fma3(zh, T, -7.647163731819816e-13);
    T = kr_fma3(zh, T, 1.6059043836821613e-10);
    T = kr_fma3(zh, T, -2.505210838544172e-08);
    T = kr_fma3(zh, T, 2.7557319223985893e-06);
    T = kr_fma3(zh, T, -0.0001984126984126984);
    double s_ph, s_pl;
    {
        const double u = zh * T;
        const double S2h = 0x1.1111111111111p-7, S2l = 0x1.1111111111111p-63, S1h = -0x1.5555555555555p-3, S1l = -0x1.5555555555555p-57;
        const double ah = S2h + u, al = ((S2h - ah) + u) + S2l;                                   // Fast2Sum: |S2h| > |u|
        const double bh = zh * ah, bl = __builtin_fma(zh, ah, -bh) + (zh * al + zl * ah);
        const double qh = S1h + bh, ql = (((S1h - qh) + bh) + bl) + S1l;
        const double ch = r * zh, cl = __builtin_fma(r, zh, -ch) + r * zl;                        // r^3
        s_ph = ch * qh;
        s_pl = __builtin_fma(ch, qh, -s_ph) + (ch * ql + cl * qh);
    }
    // cos(r + y) = 1 - z/2 - y sin r + z^2 (C2 + z (C3 + z TC(z)))
    double TC = kr_fma3(zh, 4.110317623312165e-19, -1.5619206968586225e-16);
    TC = kr_fma3(zh, TC, 4.779477332387385e-14);
    TC = kr_fma3(zh, TC, -1.1470745597729725e-11);
    TC = kr_fma3(zh, TC, 2.08767569878681e-09);
    TC = kr_fma3(zh, TC, -2.755731922398589e-07);
    TC = kr_fma3(zh, TC, 2.48015873015873e-05);
    {
        const double u = zh * TC;
        const double C3h = -0.001388888888888889, C3l = 5.300543954373577e-20, C2h = 0.041666666666666664, C2l = 2.3129646346357427e-18;
        const double ah = C3h + u, al = ((C3h - ah) + u) + C3l;
        const double bh = zh * ah, bl = __builtin_fma(zh, ah, -bh) + (zh * al + zl * ah);
        const double qh = C2h + bh, ql = (((C2h - qh) + bh) + bl) + C2l;
        const double wh = zh * zh, wl = __builtin_fma(zh, zh, -wh) + 2.0 * (zh * zl);            // z^2
        const double ph = wh * qh, pl = __builtin_fma(wh, qh, -ph) + (wh * ql + wl * qh);
        const double hz = 0.5 * zh;
        const double eh = 1.0 - hz, el = (1.0 - eh) - hz;                                         // 1 - z/2, exactly
        const double c1 = eh + ph;                                                                // cos r and sin r to double precision:
        const double sh = r + s_ph;                                                               // enough for the y terms
        const double c2 = ((eh - c1) + ph) + (((el - 0.5 * zl) + pl) - sh * y);
        const double sl = ((r - sh) + s_ph) + (s_pl + c1 * y);
        sh_o = sh; sl_o = sl; ch_o = c1; cl_o = c2;
    }
}

KR_SC_FN void kr_sincos_cr_core_f64(double r, double y, double& sr, double& cr)
{
    double sh, sl, ch, cl;
    kr_sincos_cr_core_pairs(r, y, sh, sl, ch, cl);
    cr = ch + cl;
    sr = sh + sl;
}

// sin and cos of a polar angle, correctly rounded in practice (what every strict-arithmetic caller uses: the integrators' stages, the O(N) passes,
// the ray sources, the stop tests).  A pure function of x, so a ray gets the same bits in whichever kernel / wave it is traced.
// (The shorter fdlibm minimax kernels this replaced -- max 0.84 ulp, equal to glibc on 96.5 % of arguments against 99.7-99.99 % -- are in the history
// of this file; profiles/r02_ab_experiments.txt has their timings.)
// LONE: the caller is a wave that owns its SIMD (the strict side launch: polar-axis rays, whose angles are small for most of their tens of thousands
// of steps).  Nothing covers such a wave's branch bubbles, so the small-angle path is laid out as the fall-through and the general one out of line:
// side launches 1.6-1.8 % shorter; the same layout in the ordinary strict kernels, where the general path is the usual one, costs them 3-5 %
// (profiles/r04_ab_experiments.txt).  Same values either way.
template <bool LONE = false>
KR_SC_FN void kr_sincos_f64(double x, double& s, double& c)
{
    const bool small = __builtin_fabs(x) < KR_SMALL_ANGLE_LIMIT;
#if defined(__HIP_DEVICE_COMPILE__)
    const bool all_small = __builtin_amdgcn_ballot_w64(!small) == 0;      // every active lane: the usual case on a wave of polar-axis rays
    if (LONE ? __builtin_expect(all_small, 1) : all_small) {
        kr_sincos_small_f64(x, s, c);
        return;
    }
    kr_sincos_general_f64(x, s, c);
    if (__builtin_amdgcn_ballot_w64(small) != 0) {       // mixed wave
        double s1, c1;
        kr_sincos_small_f64(x, s1, c1);
        s = small ? s1 : s;
        c = small ? c1 : c;
    }
#else
    if (small) kr_sincos_small_f64(x, s, c);
    else kr_sincos_general_f64(x, s, c);
#endif
}

KR_SC_FN void kr_sincos_general_f64(double x, double& s, double& c)
{
    const double ax = __builtin_fabs(x);
    if (__builtin_expect(!(ax < 1024.0), 0)) {
        double ls, lc;                       // (locals: the caller's variables must not have their address handed to an out-of-line call,
        kr_sincos_libm_f64(x, &ls, &lc);     // or the compiler keeps them in scratch memory on the hot path as well)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(ls), "+v"(lc));   // values from here on, not loads the optimiser may merge with other memory
#endif
        s = ls;
        c = lc;
        return;
    }
    const double t = __builtin_rint(x * 6.36619772367581382433e-01);        // 2/pi
    const int n = (int) t;
    const double r0 = __builtin_fma(-t, 1.57079632679489655800e+00, x);     // P1 = fl(pi/2); exact
    const double r = __builtin_fma(-t, 6.12323399573676603587e-17, r0);     // P2 = pi/2 - P1
    const double y = __builtin_fma(-t, 6.12323399573676603587e-17, r0 - r); // what rounding r dropped
    double sr, cr;
    kr_sincos_cr_core_f64(r, y - t * -1.4973849048591698e-33, sr, cr);                            // (third piece of pi/2)
    // quadrant: sin -> {s, c, -s, -c}[n & 3], cos -> {c, -s, -c, s}[n & 3]
    const bool odd = (n & 1) != 0;
    const double ss = odd ? cr : sr;
    const double cc = odd ? sr : cr;
    const unsigned long long sgn_s = ((unsigned long long) (unsigned) (n & 2)) << 62;
    const unsigned long long sgn_c = ((unsigned long long) (unsigned) ((n + 1) & 2)) << 62;
    s = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, ss) ^ sgn_s);
    c = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, cc) ^ sgn_c);
}

// The same pair for the fast arithmetic path (kr_device.hpp: momentum_fast / k1_with_flips_fast), which tolerates a few
// ulp per operation: identical two-step Cody-Waite reduction, plain Horner kernels without the tail corrections
// (sin: r + r z S(z), cos: 1 + z C(z); degree 13 / 14 as above), quadrant fix-up.  <= ~1.5 ulp for |x| < 1024, about 35
// instructions against ~85 for the <1-ulp routine above.  Arguments outside that range take the routine above.
KR_SC_FN void kr_sincos_fast_core_f64(double x, double& s, double& c);
KR_SC_FN void kr_sincos_fast_f64(double x, double& s, double& c)
{
    if (__builtin_expect(!(__builtin_fabs(x) < 1024.0), 0)) {
        // what kr_sincos_f64 does with such an argument -- the out-of-line library routine -- called directly: going through kr_sincos_f64 inlined
        // the correctly rounded kernel, which this path never reaches, into every fast kernel, and its ~25 coefficients (hoisted into vector
        // registers with everything else) shared high dwords with this routine's: 5 v_mov_b32 per evaluation to piece the pairs together
        double ls, lc;
        kr_sincos_libm_f64(x, &ls, &lc);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(ls), "+v"(lc));
#endif
        s = ls;
        c = lc;
        return;
    }
    kr_sincos_fast_core_f64(x, s, c);
}

// the routine proper, |x| < 1024 (the first reduction step is exact up to there; a polar angle after reflect_poles is in [0, pi]); NaN / inf in, NaN out
KR_SC_FN void kr_sincos_fast_core_f64(double x, double& s, double& c)
{
    const double t = __builtin_rint(x * 6.36619772367581382433e-01);
    const int n = (int) t;
    double r = __builtin_fma(-t, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-t, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = kr_fma3(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = kr_fma3(z, ps, 2.75573137070700676789e-06);
    ps = kr_fma3(z, ps, -1.98412698298579493134e-04);
    ps = kr_fma3(z, ps, 8.33333333332248946124e-03);
    ps = kr_fma3(z, ps, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(r * z, ps, r);
    double pc = kr_fma3(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = kr_fma3(z, pc, -2.75573143513906633035e-07);
    pc = kr_fma3(z, pc, 2.48015872894767294178e-05);
    pc = kr_fma3(z, pc, -1.38888888888741095749e-03);
    pc = kr_fma3(z, pc, 4.16666666666666019037e-02);
    pc = __builtin_fma(z, pc, -0.5);
    const double cr = __builtin_fma(z, pc, 1.0);
    const bool odd = (n & 1) != 0;
    const double ss = odd ? cr : sr;
    const double cc = odd ? sr : cr;
    const unsigned long long sgn_s = ((unsigned long long) (unsigned) (n & 2)) << 62;
    const unsigned long long sgn_c = ((unsigned long long) (unsigned) ((n + 1) & 2)) << 62;
    s = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, ss) ^ sgn_s);
    c = __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, cc) ^ sgn_c);
}
