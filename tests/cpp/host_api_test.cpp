// GPU test program for the host-side C++ mirror (raytrace_cpu_amd/host): exercises parts of the reference class API
// that the reference's own apps do not reach.  Built by tests/cpp/Makefile, run by tests/test_gpu_host_mirror.py.
// Exit code 0 = every check passed; each failed check prints a line.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "raytracer/imageplane.h"
#include "raytracer/pointsource.h"
#include "raytracer/ray_destination.h"

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } while (0)

// a user-defined destination: the HIP path must refuse it in run_raytrace, but redshift(dest) must still work (host loop)
template <typename T>
class SlowDisc : public RayDestination<T> {
public:
    bool reached(T r, T theta, T phi) const override { return theta >= T(M_PI_2); }
    T velocity(T r, T theta, T phi) const override { return T(0.5) / (T(0.998) + r * std::sqrt(r)); }   // half Keplerian
};

int main(int argc, char** argv)
{
    const double spin = 0.998;
    double pos[4] = {0.0, 10.0, 1e-3, 1.5707};

    // `host_api_test steplim-env` with KRTRACE_STEPLIM=40 in the environment: the limit applies where the application passes none
    // (Euler / RK4 default STEPLIM = 1e7), marks the unfinished rays RAY_STATUS_STEPLIM with a negated count, and leaves an explicit steplim alone
    if (argc > 1 && std::string(argv[1]) == "steplim-env") {
        PointSource<double> a(pos, 0.0, spin, TOL, 0.2, 0.2, -0.995, 0.995, -M_PI, M_PI);
        PointSource<double> b(pos, 0.0, spin, TOL, 0.2, 0.2, -0.995, 0.995, -M_PI, M_PI);
        a.run_raytrace(Integrator::RK4, M_PI_2, 1000.0, 0);                                   // no steplim passed -> the environment's 40
        b.run_raytrace(Integrator::RK4, M_PI_2, 1000.0, 0, 0, 1, -1, -1, true, 100000);       // explicit -> untouched
        int capped = 0, live = 0;
        for (int i = 0; i < a.get_count(); i++) {
            if (b.rays[i].steps == -1) continue;
            ++live;
            if (b.rays[i].steps > 40) {
                ++capped;
                CHECK(a.rays[i].steps == -40 && (a.rays[i].status & RAY_STATUS_STEPLIM), "ray %d: steps %d status %d under KRTRACE_STEPLIM=40", i, a.rays[i].steps, a.rays[i].status);
            } else {
                CHECK(a.rays[i].steps == b.rays[i].steps, "ray %d finished early but differs: %d vs %d", i, a.rays[i].steps, b.rays[i].steps);
            }
        }
        CHECK(live > 100 && capped > 50, "too few rays: live %d capped %d", live, capped);
        std::printf(failures ? "FAIL\n" : "PASS\n");
        return failures ? 1 : 0;
    }

    if (argc > 1 && !std::strcmp(argv[1], "progress")) {
        // run_raytrace(show_progress = -250000) on 2e6 rays: plain report lines "Ray <done>/<total>" (negative: no bar), the integrator's banner first
        const double d = 1.99 / (std::sqrt(2.0e6) - 1.0);
        PointSource<double> a(pos, 0.0, spin, TOL, d, d * M_PI / 0.995, -0.995, 0.995, -M_PI, M_PI);
        a.run_raytrace(Integrator::RK4, M_PI_2, 1000.0, -250000);
        int traced = 0;
        for (int i = 0; i < a.get_count(); i++) traced += a.rays[i].steps > 0;
        CHECK(traced > 1900000, "traced %d", traced);
        std::printf(failures ? "FAIL\n" : "PASS\n");
        return failures ? 1 : 0;
    }

    // 1. whole-array run_raytrace == per-ray propagate_rk4 (single-ray launches of the same kernel)
    {
        PointSource<double> a(pos, 0.0, spin, TOL, 0.2, 0.2, -0.995, 0.995, -M_PI, M_PI);
        PointSource<double> b(pos, 0.0, spin, TOL, 0.2, 0.2, -0.995, 0.995, -M_PI, M_PI);
        a.run_raytrace(Integrator::RK4, M_PI_2, 1000.0, 0);
        int moved = 0;
        for (int i = 0; i < b.get_count(); i += 7) {
            if (b.rays[i].steps < 0) continue;
            const int st = b.propagate_rk4(i, 1000.0, M_PI_2, STEPLIM);
            CHECK(st == a.rays[i].steps, "ray %d: propagate_rk4 returned %d steps, run_raytrace %d", i, st, a.rays[i].steps);
            CHECK(b.rays[i].r == a.rays[i].r && b.rays[i].t == a.rays[i].t && b.rays[i].status == a.rays[i].status, "ray %d state differs", i);
            ++moved;
        }
        CHECK(moved > 10, "too few rays compared");
        // the single-ray forms have no skip rule and take steplim verbatim (reference raytracer.cpp:129-340): a ray marked
        // unused (steps == -1) is traced all the same, and steplim <= 0 means zero iterations + RAY_STATUS_STEPLIM
        PointSource<double> c(pos, 0.0, spin, TOL, 0.2, 0.2, -0.995, 0.995, -M_PI, M_PI);
        int live = -1;
        for (int i = 0; i < c.get_count(); i++) if (c.rays[i].steps == 0 && a.rays[i].steps > 5) { live = i; break; }
        CHECK(live >= 0, "no live ray found");
        c.rays[live].steps = -1;
        const int st1 = c.propagate_rk4(live, 1000.0, M_PI_2, STEPLIM);
        CHECK(st1 == a.rays[live].steps && c.rays[live].steps == st1 - 1 && c.rays[live].r == a.rays[live].r, "steps == -1 ray: returned %d, record %d (run_raytrace took %d)", st1,
              c.rays[live].steps, a.rays[live].steps);
        PointSource<double> z(pos, 0.0, spin, TOL, 0.2, 0.2, -0.995, 0.995, -M_PI, M_PI);
        const double r_before = z.rays[live].r;
        z.rays[live].steps = 7;
        const int st0 = z.propagate_rk4(live, 1000.0, M_PI_2, 0);
        CHECK(st0 == 0 && z.rays[live].r == r_before && (z.rays[live].status & RAY_STATUS_STEPLIM) && z.rays[live].steps == -7, "steplim = 0: returned %d, steps %d status %d", st0,
              z.rays[live].steps, z.rays[live].status);
        // a bounded call followed by the rest equals the one-shot result
        PointSource<double> h2(pos, 0.0, spin, TOL, 0.2, 0.2, -0.995, 0.995, -M_PI, M_PI);
        const int part = h2.propagate_rk4(live, 1000.0, M_PI_2, 3);
        CHECK(part == 3 && h2.rays[live].steps == -3, "bounded call: returned %d, steps %d", part, h2.rays[live].steps);
    }

    // 2. the float instantiation traces on the GPU too and lands near the double result
    {
        float posf[4] = {0.f, 10.f, 1e-3f, 1.5707f};
        PointSource<float> f(posf, 0.f, 0.998f, TOL, 0.2f, 0.2f, -0.995f, 0.995f, (float) -M_PI, (float) M_PI);
        PointSource<double> d(pos, 0.0, spin, TOL, 0.2, 0.2, -0.995, 0.995, -M_PI, M_PI);
        f.redshift_start();
        d.redshift_start();
        f.run_raytrace(Integrator::RK4, (float) M_PI_2, 1000.f, 0, 0, 1, -1, -1, true, 20000);
        d.run_raytrace(Integrator::RK4, M_PI_2, 1000.0, 0, 0, 1, -1, -1, true, 20000);
        f.range_phi();
        f.redshift(-1.0f);
        d.range_phi();
        d.redshift(-1.0);
        int close = 0, live = 0;
        const int n = std::min(f.get_count(), d.get_count());
        for (int i = 0; i < n; i++) {
            if (d.rays[i].steps <= 0 || f.rays[i].steps <= 0 || d.rays[i].status != 1) continue;
            ++live;
            if (f.rays[i].status == 1 && std::fabs(f.rays[i].r - d.rays[i].r) < 2e-2 * d.rays[i].r && std::fabs(f.rays[i].redshift - d.rays[i].redshift) < 1e-2) ++close;
        }
        CHECK(live > 50 && close > 0.9 * live, "float vs double: %d of %d disc rays close", close, live);
    }

    // 3. RayDestination overloads: built-ins go to the kernel; Euler and user subclasses are refused loudly
    {
        PointSource<double> s(pos, 0.0, spin, TOL, 0.2, 0.2, -0.995, 0.995, -M_PI, M_PI);
        PointSource<double> ref(pos, 0.0, spin, TOL, 0.2, 0.2, -0.995, 0.995, -M_PI, M_PI);
        FlatDiscDestination<double> flat(M_PI_2);
        s.redshift_start();
        ref.redshift_start();
        s.run_raytrace(&flat, Integrator::RK4, 1000.0, 0);
        ref.run_raytrace(Integrator::RK4, M_PI_2, 1000.0, 0);
        int same = 0, live = 0;
        for (int i = 0; i < s.get_count(); i++) {
            if (ref.rays[i].steps <= 0 || ref.rays[i].status != 1) continue;
            ++live;
            if (s.rays[i].status == 1 && s.rays[i].theta >= M_PI_2 && std::fabs(s.rays[i].r - ref.rays[i].r) < 0.1 * ref.rays[i].r) ++same;   // no theta clip in the dest overload: it lands up to one step past the plane
        }
        CHECK(live > 50 && same > 0.9 * live, "FlatDiscDestination vs theta limit: %d of %d", same, live);

        bool threw = false;
        try { s.run_raytrace(&flat, Integrator::Euler, 1000.0, 0); } catch (const std::invalid_argument&) { threw = true; }
        CHECK(threw, "Euler + RayDestination must throw (reference asserts)");
        SlowDisc<double> custom;
        threw = false;
        try { s.run_raytrace(&custom, Integrator::RK4, 1000.0, 0); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw, "user-defined RayDestination must be refused by run_raytrace");
        threw = false;
        try { s.run_raytrace(Integrator::RK4, M_PI_2, 1000.0, 0, (TextOutput*) 0x1); } catch (const std::runtime_error&) { threw = true; }
        CHECK(threw, "outfile != nullptr must be refused");

        // redshift(dest): built-in -> device kernel; custom velocity field -> host loop through the virtual call
        s.redshift(&flat);
        std::vector<double> g_kep(s.get_count());
        for (int i = 0; i < s.get_count(); i++) g_kep[i] = s.rays[i].redshift;
        s.redshift(-1.0);
        int agree = 0, n_disc = 0;
        for (int i = 0; i < s.get_count(); i++) {
            if (s.rays[i].steps <= 0 || s.rays[i].status != 1) continue;
            ++n_disc;
            if (std::fabs(s.rays[i].redshift - g_kep[i]) < 1e-12 * std::fabs(g_kep[i])) ++agree;   // Keplerian either way
        }
        CHECK(n_disc > 50 && agree == n_disc, "redshift(dest) vs redshift(-1): %d of %d", agree, n_disc);
        s.redshift(&custom);
        int differ = 0;
        for (int i = 0; i < s.get_count(); i++)
            if (s.rays[i].steps > 0 && s.rays[i].status == 1 && std::fabs(s.rays[i].redshift - g_kep[i]) > 1e-6) ++differ;
        CHECK(differ > n_disc / 2, "custom velocity field had no effect (%d)", differ);
    }

    // 4. ImagePlane helpers and the reverse redshift pair
    {
        ImagePlane<double> ip(10000.0, 80.0, -30.0, 30.0, 4.0, -30.0, 30.0, 4.0, spin, 0.0);
        CHECK(ip.get_count() == 256 && ip.get_x_index(17) == 1 && ip.get_y_index(17) == 1, "ImagePlane indexing");
        CHECK(ip.ray_x(16) == -26.0 && ip.ray_y(3) == -18.0, "ray_x/ray_y");
        ip.redshift_start();
        ip.run_raytrace(Integrator::RK4, M_PI_2, 11000.0, 0);
        ip.redshift(false);
        ip.range_phi();
        int hits = 0;
        for (int i = 0; i < ip.get_count(); i++)
            if (ip.rays[i].steps > 0 && ip.rays[i].status == 1 && ip.rays[i].redshift > 0.3 && ip.rays[i].redshift < 3) ++hits;
        CHECK(hits > 100, "image plane disc hits with sane redshift: %d", hits);
    }

    std::printf(failures ? "FAILED (%d)\n" : "PASS\n", failures);
    return failures ? 1 : 0;
}
