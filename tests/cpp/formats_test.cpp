// tests/cpp/formats_test.cpp -- exercises the host-side file formats (raytrace_cpu_amd/host/include/*.h, apps/*.h)
// for tests/test_host_formats.py.  Pure host code: no GPU, no libkrtrace.
//   formats_test dat  <in.dat> <out.dat>                 re-emit a 7-column emissivity table through TextOutput
//   formats_test fits <parfile> <planes.bin> <disc_count> <out.fits>
//                                                        rebuild an imageplane_disc_image FITS file from its parameter
//                                                        file and 6 planes of img_Nx*img_Ny doubles ([ix][iy] order)
//   formats_test cards <spec.txt> <out.fits>             replay a list of FITSOutput calls (P | I nx ny | E name | C text |
//                                                        K{i,l,d,s,b} key|comment|value), images filled with zeros
//   formats_test par  <parfile> [--key=value ...]        print what ParameterFile / ParameterArgs parse
//   formats_test fullimage <file> <limit>                one 512 x 512 image under RLIMIT_FSIZE = limit bytes (0: none); exit code 3 when the
//                                                        writer reports an I/O error
//   formats_test area <spin> <r0> <r1>                   integrate_disc_area(r0, r1, spin), 17 digits
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <sys/resource.h>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
using namespace std;

#include "../../raytrace_cpu_amd/apps/disc_image_fits.h"
#include "../../raytrace_cpu_amd/apps/emissivity_table.h"
#include "../../raytrace_cpu_amd/host/include/disc.h"
#include "../../raytrace_cpu_amd/host/include/kerr.h"
#include "../../raytrace_cpu_amd/host/include/par_args.h"
#include "../../raytrace_cpu_amd/host/include/par_file.h"

static int redo_dat(const char* in_name, const char* out_name)
{
    ifstream in(in_name);
    vector<double> col[7];
    string line;
    while (getline(in, line)) {
        if (line.find_first_not_of(" \t") == string::npos) continue;
        istringstream row(line);
        string tok;
        for (int c = 0; c < 7; ++c) {
            row >> tok;
            col[c].push_back(strtod(tok.c_str(), nullptr));   // strtod keeps the sign of "-nan"
        }
    }
    const int n = (int) col[0].size();
    TextOutput out(out_name);
    for (int i = 0; i < n; ++i) out << col[0][i] << col[1][i] << static_cast<long>(col[2][i]) << col[3][i] << col[4][i] << col[5][i] << col[6][i] << endl;
    out.close();
    return 0;
}

static int redo_fits(const char* par_name, const char* planes_name, long disc_count, const char* out_name)
{
    ParameterFile par(par_name);
    krapp::DiscImageInfo info;
    info.dist = par.get_parameter<double>("dist");
    info.incl = par.get_parameter<double>("incl");
    info.spin = par.get_parameter<double>("spin");
    info.r_isco = kerr_isco<double>(info.spin, +1);
    info.r_disc = par.get_parameter<double>("r_disc");
    info.q1 = par.get_parameter<double>("q1", 3);
    info.rb1 = par.get_parameter<double>("rb1", 4);
    info.q2 = par.get_parameter<double>("q2", 3);
    info.rb2 = par.get_parameter<double>("rb2", 10);
    info.q3 = par.get_parameter<double>("q3", 3);
    const int Nx = par.get_parameter<int>("Nx"), Ny = par.get_parameter<int>("Ny", Nx);
    info.nrays = Nx * Ny;
    info.disc_count = disc_count;
    krapp::AxisInfo& ax = info.ax;
    ax.x0 = par.get_parameter<double>("x0", -info.r_disc);
    ax.xmax = par.get_parameter<double>("xmax", info.r_disc);
    ax.y0 = par.get_parameter<double>("y0", ax.x0);
    ax.ymax = par.get_parameter<double>("ymax", ax.xmax);
    ax.dx = (ax.xmax - ax.x0) / Nx;
    ax.dy = (ax.ymax - ax.y0) / Ny;
    ax.img_nx = par.get_parameter<int>("img_Nx", Nx);
    ax.img_ny = par.get_parameter<int>("img_Ny", ax.img_nx);
    Array2D<double> a0(ax.img_nx, ax.img_ny), a1(ax.img_nx, ax.img_ny), a2(ax.img_nx, ax.img_ny), a3(ax.img_nx, ax.img_ny), a4(ax.img_nx, ax.img_ny),
        a5(ax.img_nx, ax.img_ny);
    Array2D<double>* arrays[6] = {&a0, &a1, &a2, &a3, &a4, &a5};
    ifstream in(planes_name, ios::binary);
    for (auto* p : arrays) p->read(&in);
    if (!in) { cerr << "short planes file" << endl; return 2; }
    double** planes[6] = {a0.ptr, a1.ptr, a2.ptr, a3.ptr, a4.ptr, a5.ptr};
    krapp::write_disc_image_fits(out_name, info, planes);
    return 0;
}

static int replay_cards(const char* spec_name, const char* out_name)
{
    ifstream spec(spec_name);
    FITSOutput<double> fits(out_name);
    string line;
    while (getline(spec, line)) {
        if (line.empty()) continue;
        const char op = line[0];
        const string rest = line.size() > 2 ? line.substr(2) : string();
        if (op == 'P') {
            fits.create_primary();
        } else if (op == 'I') {
            int nx = 0, ny = 0;
            sscanf(rest.c_str(), "%d %d", &nx, &ny);
            vector<double> zeros((size_t) nx * ny, 0.0);
            fits.write_image_array(zeros.data(), nx, ny);
        } else if (op == 'E') {
            fits.set_ext_name(rest.c_str());
        } else if (op == 'C') {
            fits.write_comment(rest.c_str());
        } else if (op == 'K') {
            const char type = line[1];
            const string body = line.substr(3);
            const size_t a = body.find('|'), b = body.find('|', a + 1);
            const string key = body.substr(0, a), comment = body.substr(a + 1, b - a - 1), value = body.substr(b + 1);
            switch (type) {
                case 'i': fits.write_keyword(key.c_str(), comment.c_str(), atoi(value.c_str())); break;
                case 'l': fits.write_keyword(key.c_str(), comment.c_str(), atol(value.c_str())); break;
                case 'd': fits.write_keyword(key.c_str(), comment.c_str(), strtod(value.c_str(), nullptr)); break;
                case 'b': fits.write_keyword(key.c_str(), comment.c_str(), value == "T"); break;
                default: fits.write_keyword(key.c_str(), comment.c_str(), value.c_str()); break;
            }
        }
    }
    fits.close();
    return 0;
}

static int show_par(int argc, char** argv)
{
    ParameterFile par(argv[2]);
    ParameterArgs args(argc - 2, argv + 2);
    cout.precision(17);
    cout << "source:";
    double src[4];
    par.get_parameter_array("source", src, 4);
    for (double v : src) cout << " " << v;
    cout << "\nspin: " << (args.key_exists("--spin") ? args.get_parameter<double>("--spin") : par.get_parameter<double>("spin"));
    cout << "\nNr: " << par.get_parameter<int>("Nr", 100);
    cout << "\nlogbin_r: " << par.get_parameter<bool>("logbin_r", true);
    cout << "\noutfile: " << par.get_parameter<string>("outfile");
    cout << "\ngamma(default): " << par.get_parameter<double>("gamma", 2);
    cout << "\npositional: " << args.num_positional();
    try {
        par.get_parameter<double>("no_such_key");
    } catch (const exception& e) {
        cout << "\nmissing: " << e.what();
    }
    try {
        par.get_parameter<int>("outfile");
    } catch (const exception& e) {
        cout << "\nunparsable: " << e.what();
    }
    cout << endl;
    return 0;
}

int main(int argc, char** argv)
{
    const string mode = argc > 1 ? argv[1] : "";
    if (mode == "dat" && argc == 4) return redo_dat(argv[2], argv[3]);
    if (mode == "fits" && argc == 6) return redo_fits(argv[2], argv[3], atol(argv[4]), argv[5]);
    if (mode == "cards" && argc == 4) return replay_cards(argv[2], argv[3]);
    if (mode == "par" && argc >= 3) return show_par(argc, argv);
    if (mode == "fullimage" && argc == 4) {
        // a 512 x 512 image (2 MB) into <file> under a file-size limit of <limit> bytes (0: none): writes beyond it fail (EFBIG) and close() has to say so
        if (atol(argv[3]) > 0) {
            signal(SIGXFSZ, SIG_IGN);
            struct rlimit lim = {(rlim_t) atol(argv[3]), (rlim_t) atol(argv[3])};
            setrlimit(RLIMIT_FSIZE, &lim);
        }
        try {
            FITSOutput<double> fits(argv[2]);
            fits.create_primary();
            vector<double> img(512 * 512, 1.0);
            vector<double*> rows(512);
            for (int i = 0; i < 512; ++i) rows[i] = &img[(size_t) i * 512];
            fits.write_image(rows.data(), 512, 512);
            fits.close();
        } catch (const FITSOutputException& e) {
            cerr << e.what() << endl;
            return 3;
        }
        return 0;
    }
    if (mode == "area" && argc == 5) {
        printf("%.17g\n", integrate_disc_area(atof(argv[3]), atof(argv[4]), atof(argv[2])));
        return 0;
    }
    cerr << "usage: formats_test dat|fits|par|area ..." << endl;
    return 64;
}
