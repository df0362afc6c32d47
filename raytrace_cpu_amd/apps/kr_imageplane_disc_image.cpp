// apps/kr_imageplane_disc_image.cpp -- the reference's `imageplane_disc_image` program
// (src/imageplane/imageplane_disc_image.cpp) with the ray pipeline resident on the MI355X: the image-plane rays are
// generated, traced to the disc, redshifted and accumulated into the seven image planes in HBM; only the planes come
// back.  Same parameter file and overrides, same 7-HDU FITS file (primary + FLUX, RADIUS, PHI, ENSHIFT, TIME, EMIS),
// written by include/fits_output.h without cfitsio.
//
// Reads (imageplane_disc_image.cpp:30-66): --parfile (default ../par/imageplane_disc_image.par), --outfile | outfile, dist,
// --incl | incl, plane_phi0 = 0, --spin | spin, r_disc, x0 = -r_disc, xmax = r_disc, Nx, y0 = x0, ymax = xmax, Ny = Nx,
// img_Nx = Nx, img_Ny = img_Nx, q1 = 3, rb1 = 4, q2 = 3, rb2 = 10, q3 = 3, precision = 100, flip_image = true,
// integrator = rk45, rk45_tol = 1e-8.  (max_tstep is read and ignored by the reference, :63, :109.)
// Extensions: --integrator, --arithmetic | KRTRACE_ARITHMETIC, --device, --timing.
// Difference: under RK45 a pixel exactly at (0, 0) never returns in the reference; here it ends with KR_STATUS_NAN.
#include <cmath>
#include <iostream>
#include <string>
#include <vector>
using namespace std;

#include "../host/include/kerr.h"
#include "../host/include/par_args.h"
#include "../host/include/par_file.h"
#include "app_common.h"
#include "disc_image_fits.h"

using krapp::AxisInfo;

int main(int argc, char** argv)
try {
    (void) kr_configure_process();       // first HIP user of this process: hardware queues for overlapping launches (include/kr_trace.h)
    ParameterArgs args(argc, argv);
    const string par_name = args.key_exists("--parfile") ? args.get_string_parameter("--parfile") : string("../par/imageplane_disc_image.par");
    ParameterFile par(par_name);

    const string out_name = args.key_exists("--outfile") ? args.get_parameter<string>("--outfile") : par.get_parameter<string>("outfile");
    const double dist = par.get_parameter<double>("dist");
    const double incl = args.key_exists("--incl") ? args.get_parameter<double>("--incl") : par.get_parameter<double>("incl");
    const double plane_phi0 = par.get_parameter<double>("plane_phi0", 0);
    const double spin = args.key_exists("--spin") ? args.get_parameter<double>("--spin") : par.get_parameter<double>("spin");
    const double r_disc = par.get_parameter<double>("r_disc");
    AxisInfo ax;
    ax.x0 = par.get_parameter<double>("x0", -1 * r_disc);
    ax.xmax = par.get_parameter<double>("xmax", r_disc);
    const int Nx = par.get_parameter<int>("Nx");
    ax.y0 = par.get_parameter<double>("y0", ax.x0);
    ax.ymax = par.get_parameter<double>("ymax", ax.xmax);
    const int Ny = par.get_parameter<int>("Ny", Nx);
    ax.img_nx = par.get_parameter<int>("img_Nx", Nx);
    ax.img_ny = par.get_parameter<int>("img_Ny", ax.img_nx);
    const double q1 = par.get_parameter<double>("q1", 3), rb1 = par.get_parameter<double>("rb1", 4), q2 = par.get_parameter<double>("q2", 3),
                 rb2 = par.get_parameter<double>("rb2", 10), q3 = par.get_parameter<double>("q3", 3);
    const double precision = par.get_parameter<double>("precision", 100);
    const bool flip_image = par.get_parameter<bool>("flip_image", true);
    const string integ = args.key_exists("--integrator") ? args.get_parameter<string>("--integrator") : par.get_parameter<string>("integrator", "rk45");
    const double rk45_tol = par.get_parameter<double>("rk45_tol", 1e-8);
    const string arith = args.key_exists("--arithmetic") ? args.get_parameter<string>("--arithmetic") : krapp::arithmetic_from_env();
    const bool timing = args.key_exists("--timing");

    ax.dx = (ax.xmax - ax.x0) / Nx;
    ax.dy = (ax.ymax - ax.y0) / Ny;
    const double r_isco = kerr_isco<double>(spin, +1);
    cout << "ISCO at " << r_isco << endl;

    kr_imageplane plane;
    memset(&plane, 0, sizeof plane);
    plane.dist = dist;
    plane.inc_deg = incl;
    plane.x0 = ax.x0; plane.xmax = ax.xmax; plane.dx = ax.dx;
    plane.y0 = ax.y0; plane.ymax = ax.ymax; plane.dy = ax.dy;
    plane.spin = spin;
    plane.phi0 = plane_phi0;
    plane.precision = precision;

    kr_image_bins bins;
    memset(&bins, 0, sizeof bins);
    bins.x0 = ax.x0; bins.y0 = ax.y0;
    bins.img_dx = (ax.xmax - ax.x0) / ax.img_nx;
    bins.img_dy = (ax.ymax - ax.y0) / ax.img_ny;
    bins.r_isco = r_isco; bins.r_disc = r_disc;
    bins.q1 = q1; bins.rb1 = rb1; bins.q2 = q2; bins.rb2 = rb2; bins.q3 = q3;
    bins.img_nx = ax.img_nx; bins.img_ny = ax.img_ny;
    bins.flip_image = flip_image ? 1 : 0;

    kr_params p;
    kr_params_default(&p, -spin);            // the image plane traces backwards in time: spin enters negated (imageplane.cpp:12)
    p.precision = precision;
    p.integrator = krapp::integrator_code(integ, KR_RK45);
    if (p.integrator == KR_RK45) p.rk45_tol = rk45_tol;
    p.theta_max = M_PI_2;
    p.r_max = 1.1 * dist;
    p.stop_kind = KR_STOP_THETA;
    p.flags = krapp::arithmetic_flags(arith, p.integrator);

    // ---- device pipeline ------------------------------------------------------------------------------------------
    krapp::check(kr_set_device(args.get_parameter<int>("--device", 0)), "kr_set_device");
    krapp::Stopwatch clock;
    const int64_t n = kr_imageplane_count(&plane, nullptr, nullptr);
    if (n <= 0) throw runtime_error("empty ray grid");
    const int64_t npix = (int64_t) ax.img_nx * ax.img_ny;
    krapp::DeviceBuffer rays(n * (int64_t) sizeof(kr_ray_f64));
    krapp::DeviceBuffer planes((7 * npix + 1) * (int64_t) sizeof(double));
    planes.zero();
    krapp::check(kr_imageplane_init_emit_dev_f64(&plane, 0, 1, 0.0, 1, 0, rays.get(), n, nullptr), "imageplane_init + redshift_start");
    krapp::check(kr_synchronize(nullptr), "sync");
    const double ms_init = clock.lap_ms();
    kr_stats st;
    krapp::check(kr_trace_dev_f64(&p, rays.get(), n, nullptr, &st), "trace");
    const double ms_trace = clock.lap_ms();
    krapp::check(kr_post_image_dev_f64(-spin, -1.0, 1, 0, 0, -1 * M_PI, M_PI, &bins, rays.get(), n, planes.get(), nullptr), "redshift + range_phi + image planes");
    krapp::PinnedDoubles h(7 * npix + 1);     // 0.94 GB at 4096^2: page-locked, or the read-back runs at a tenth of the PCIe rate
    krapp::check(kr_memcpy_d2h(h.data(), planes.get(), h.size() * (int64_t) sizeof(double)), "d2h");
    const double ms_post = clock.lap_ms();

    // ---- per-pixel means (imageplane_disc_image.cpp:165-174): flux only where rays arrived, the rest 0/0 -> NaN ---------
    // In place in the page-locked read-back buffer, on a team of threads: seven 134-MB Array2D copies, their zero-fill and six serial
    // division passes were 0.3 s of this program at 4096 x 4096.
    const long disc_count = static_cast<long>(h[7 * npix]);
    cout << disc_count << " rays hit the disc" << endl;
    double* hp = h.data();
    krapp::parallel_for(npix, [=](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; ++i) {
            const int hits = static_cast<int>(hp[i]);
            if (hits > 0) hp[npix + i] /= hits;
            for (int k = 2; k <= 6; ++k) hp[k * npix + i] /= hits;
        }
    });
    vector<vector<double*>> rows(6, vector<double*>(static_cast<size_t>(ax.img_nx)));
    double** sums[6];
    for (int k = 0; k < 6; ++k) {
        for (int ix = 0; ix < ax.img_nx; ++ix) rows[k][ix] = hp + (k + 1) * npix + static_cast<int64_t>(ix) * ax.img_ny;
        sums[k] = rows[k].data();
    }
    const double ms_means = clock.lap_ms();

    // ---- FITS (imageplane_disc_image.cpp:176-304) ----------------------------------------------------------------------
    krapp::DiscImageInfo info = {dist, incl, spin, r_isco, r_disc, q1, rb1, q2, rb2, q3, Nx * Ny, disc_count, ax};
    krapp::write_disc_image_fits(out_name, info, sums);
    const double ms_fits = clock.lap_ms();

    if (timing)
        cout << "timing: rays " << st.rays_traced << " steps " << st.steps_total << " | init+redshift_start " << ms_init << " ms | trace " << ms_trace
             << " ms (kernel " << st.kernel_ms << ") | redshift+range_phi+planes+readback " << ms_post << " ms | per-pixel means " << ms_means
             << " ms | FITS file " << ms_fits << " ms" << endl;
    cout << "Done" << endl;
    return 0;
} catch (const exception& e) {
    cerr << e.what() << endl;
    return 1;
}
