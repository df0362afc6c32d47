// apps/kr_emissivity.cpp -- the reference's `emissivity` program (src/emissivity/emissivity.cpp) with the whole ray
// pipeline resident on the MI355X: rays are generated, traced, redshifted and binned in HBM; only the Nr-bin
// histogram comes back.  Same parameter file, same command-line overrides, same 7-column output table
// (r, area, count, flux/area, emis/area, <g>, <t>; TextOutput format).
//
// Reads (emissivity.cpp:17-55): --parfile (default ../par/emissivity.par), --outfile | outfile, source[4], V = 0,
// --spin | spin, cosalpha0 = -0.995, cosalphamax = 0.995, dcosalpha, beta0 = -pi, betamax = pi, dbeta, r_esc = 1000
// (outer radius of the trace), --rmin | rmin = -1 (-> ISCO), --Nr | Nr = 100, r_disc := r_esc key, default 500 (sic, :51),
// logbin_r = true, gamma = 2, --source_h.  Extensions: --integrator | integrator = rk45 (the reference hard-codes RK45,
// :91; BASELINE configs[1] is the same run with rk4), --arithmetic | KRTRACE_ARITHMETIC = hybrid|strict|fast,
// --device = 0, --timing.
//
// What is NOT taken from the reference: with logbin_r = 0 its bin areas do not compile (`disc_r + dr`, :79); the
// intended upper edge r + dr (as in emissivity_rd.cpp:88 ... which passes `dr` alone) is used here.
#include <cmath>
#include <iostream>
#include <string>
#include <vector>
using namespace std;

#include "../host/include/disc.h"
#include "../host/include/kerr.h"
#include "../host/include/par_args.h"
#include "../host/include/par_file.h"
#include "app_common.h"
#include "emissivity_table.h"

int main(int argc, char** argv)
try {
    (void) kr_configure_process();       // first HIP user of this process: hardware queues for overlapping launches (include/kr_trace.h)
    ParameterArgs args(argc, argv);
    const string par_name = args.key_exists("--parfile") ? args.get_string_parameter("--parfile") : string("../par/emissivity.par");
    ParameterFile par(par_name);

    const string out_name = args.key_exists("--outfile") ? args.get_parameter<string>("--outfile") : par.get_parameter<string>("outfile");
    double source[4];
    par.get_parameter_array("source", source, 4);
    if (args.key_exists("--source_h")) source[1] = args.get_parameter<double>("--source_h");
    const double V = par.get_parameter<double>("V", 0);
    const double spin = args.key_exists("--spin") ? args.get_parameter<double>("--spin") : par.get_parameter<double>("spin");
    const double r_max = par.get_parameter<double>("r_esc", 1000);
    double r_min = args.key_exists("--rmin") ? args.get_parameter<double>("--rmin") : par.get_parameter<double>("rmin", -1);
    const int Nr = args.key_exists("--Nr") ? args.get_parameter<int>("--Nr") : par.get_parameter<int>("Nr", 100);
    const double r_disc = par.get_parameter<double>("r_esc", 500);
    const bool logbin_r = par.get_parameter<bool>("logbin_r", true);
    const double gamma = par.get_parameter<double>("gamma", 2);
    const string integ = args.key_exists("--integrator") ? args.get_parameter<string>("--integrator") : par.get_parameter<string>("integrator", "rk45");
    const string arith = args.key_exists("--arithmetic") ? args.get_parameter<string>("--arithmetic") : krapp::arithmetic_from_env();
    const bool timing = args.key_exists("--timing");

    kr_pointsource src;
    memset(&src, 0, sizeof src);
    for (int i = 0; i < 4; ++i) src.pos[i] = source[i];
    src.V = V;
    src.spin = spin;
    src.tol = 100;   // TOL, raytracer.h
    src.E = 1;
    src.cosalpha0 = par.get_parameter<double>("cosalpha0", -0.995);
    src.cosalphamax = par.get_parameter<double>("cosalphamax", 0.995);
    src.dcosalpha = par.get_parameter<double>("dcosalpha");
    src.beta0 = par.get_parameter<double>("beta0", -1 * M_PI);
    src.betamax = par.get_parameter<double>("betamax", M_PI);
    src.dbeta = par.get_parameter<double>("dbeta");

    // radial bins and their proper areas (host: Nr x 49 small tetrad evaluations)
    const double r_isco = kerr_isco<double>(spin, +1);
    if (r_min < 0) r_min = r_isco;
    const double dr = logbin_r ? exp(log(r_disc / r_min) / Nr) : (r_disc - r_min) / Nr;
    vector<double> bin_r(Nr), bin_area(Nr);
    for (int ir = 0; ir < Nr; ++ir) {
        bin_r[ir] = logbin_r ? r_min * pow(dr, ir) : r_min + ir * dr;
        bin_area[ir] = integrate_disc_area(bin_r[ir], logbin_r ? bin_r[ir] * dr : bin_r[ir] + dr, spin);
    }
    const long num_primary_rays = (((src.cosalphamax - src.cosalpha0) / src.dcosalpha) * ((src.betamax - src.beta0) / src.dbeta));

    kr_emis_bins bins;
    memset(&bins, 0, sizeof bins);
    bins.r_min = r_min;
    bins.dr = dr;
    bins.r_isco = r_isco;
    bins.gamma = gamma;
    bins.spin = spin;
    bins.num_primary_rays = static_cast<double>(num_primary_rays);
    bins.nr = Nr;
    bins.logbin = logbin_r ? 1 : 0;

    kr_params p;
    kr_params_default(&p, spin);
    p.integrator = krapp::integrator_code(integ, KR_RK45);
    p.theta_max = M_PI_2;
    p.r_max = r_max;
    p.stop_kind = KR_STOP_THETA;
    p.flags = krapp::arithmetic_flags(arith, p.integrator);

    // ---- device pipeline ------------------------------------------------------------------------------------------
    krapp::check(kr_set_device(args.get_parameter<int>("--device", 0)), "kr_set_device");
    krapp::Stopwatch clock;
    const int64_t n = kr_pointsource_count(&src, nullptr, nullptr);
    if (n <= 0) throw runtime_error("empty ray grid");
    krapp::DeviceBuffer rays(n * (int64_t) sizeof(kr_ray_f64));
    krapp::DeviceBuffer hist((5 * (int64_t) Nr + 1) * (int64_t) sizeof(double));
    hist.zero();
    krapp::check(kr_pointsource_init_emit_dev_f64(&src, 0, 1, V, 0, 0, rays.get(), n, nullptr), "pointsource_init + redshift_start");
    krapp::check(kr_synchronize(nullptr), "sync");
    const double ms_init = clock.lap_ms();
    kr_stats st;
    krapp::check(kr_trace_dev_f64(&p, rays.get(), n, nullptr, &st), "trace");
    const double ms_trace = clock.lap_ms();
    krapp::check(kr_post_emissivity_dev_f64(spin, -1.0, 0, 0, 0, -1 * M_PI, M_PI, &bins, rays.get(), n, hist.get(), nullptr), "range_phi + redshift + histogram");
    vector<double> h(5 * (size_t) Nr + 1);
    krapp::check(kr_memcpy_d2h(h.data(), hist.get(), (int64_t) (h.size() * sizeof(double))), "d2h");
    const double ms_post = clock.lap_ms();

    // ---- the divisions of emissivity.cpp:128-134 and the table ---------------------------------------------------------
    krapp::write_emissivity_table(out_name, Nr, bin_r.data(), bin_area.data(), &h[0], &h[Nr], &h[2 * (size_t) Nr], &h[3 * (size_t) Nr], &h[4 * (size_t) Nr]);

    if (timing)
        cout << "timing: rays " << st.rays_traced << " steps " << st.steps_total << " | init+redshift_start " << ms_init << " ms | trace " << ms_trace
             << " ms (kernel " << st.kernel_ms << ") | range_phi+redshift+histogram+readback " << ms_post << " ms | disc rays " << static_cast<long>(h[5 * Nr]) << endl;
    cout << "Done" << endl;
    return 0;
} catch (const exception& e) {
    cerr << e.what() << endl;
    return 1;
}
