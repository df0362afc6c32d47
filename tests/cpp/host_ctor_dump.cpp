// CPU test program: the host mirror's PointSource<double> / ImagePlane<double> constructors write their rays[] to a file, for comparison with the
// oracle's constructors (tests/test_host_constructors.py).  No GPU call is made: nothing is traced.
// usage: host_ctor_dump ps|ip <out> <ctor arguments ...>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "raytracer/imageplane.h"
#include "raytracer/pointsource.h"

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    FILE* f = std::fopen(argv[2], "wb");
    if (!f) return 3;
    auto arg = [&](int i) { return std::atof(argv[3 + i]); };
    if (!std::strcmp(argv[1], "ps") && argc == 3 + 13) {
        double pos[4] = {arg(0), arg(1), arg(2), arg(3)};
        PointSource<double> s(pos, arg(4), arg(5), TOL, arg(6), arg(7), arg(8), arg(9), arg(10), arg(11), arg(12));
        const int n = s.get_count();
        std::fwrite(&n, 4, 1, f);
        std::fwrite(s.rays, sizeof(Ray<double>), n, f);
    } else if (!std::strcmp(argv[1], "ip") && argc == 3 + 10) {
        ImagePlane<double> s(arg(0), arg(1), arg(2), arg(3), arg(4), arg(5), arg(6), arg(7), arg(8), arg(9));
        const int n = s.get_count();
        std::fwrite(&n, 4, 1, f);
        std::fwrite(s.rays, sizeof(Ray<double>), n, f);
    } else {
        std::fclose(f);
        return 2;
    }
    std::fclose(f);
    return 0;
}
