// Host microbenchmark (GPU box): fused first touch + constructor stores, by thread count.  g++ -O2 -fopenmp first_touch2.cpp
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <omp.h>
#include <initializer_list>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = 1440000000ull;
    { volatile int warm = 0;
#pragma omp parallel
      { warm = 1; } }
    for (int nt : {8, 16, 32, 64, 128, 256}) {
        if (nt > omp_get_max_threads()) break;
        char* p = nullptr;
        double t0 = now();
        if (posix_memalign((void**) &p, 1 << 21, bytes)) return 1;
        madvise(p, bytes, MADV_HUGEPAGE);
        const long n = (long) (bytes / 144);
#pragma omp parallel for schedule(static) num_threads(nt)
        for (long i = 0; i < n; i++) { int* r = (int*) (p + (size_t) i * 144 + 104); r[0] = -1; r[1] = 0; }
        double t1 = now();
#pragma omp parallel for schedule(static) num_threads(nt)
        for (long i = 0; i < n; i++) { double* r = (double*) (p + (size_t) i * 144); for (int k = 0; k < 13; k++) r[k] = 1.0 + i; }      // a source-constructor-like full write
        double t2 = now();
        printf("threads %3d: fused touch + ctor stores %.1f ms, full-record write pass %.1f ms\n", nt, t1 - t0, t2 - t1);
        free(p);
    }
}
