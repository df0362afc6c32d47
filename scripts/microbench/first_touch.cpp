// Host microbenchmark (GPU box): how fast can 1.44 GB of ray records be made resident?  g++ -O2 -fopenmp first_touch.cpp
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <omp.h>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = 1440000000ull;
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char buf[128] = {0};
    if (f) { fgets(buf, sizeof buf, f); fclose(f); }
    printf("THP: %s threads %d\n", buf, omp_get_max_threads());
    for (int variant = 0; variant < 4; variant++) {
        double t0 = now();
        char* p = nullptr;
        if (variant == 0) { posix_memalign((void**) &p, 1 << 21, bytes); madvise(p, bytes, MADV_HUGEPAGE); }
        if (variant == 1) { posix_memalign((void**) &p, 1 << 21, bytes); }
        if (variant == 2) { p = (char*) mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0); }
        if (variant == 3) { p = (char*) mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); madvise(p, bytes, MADV_HUGEPAGE); }
        double t1 = now();
        if (variant != 2) {
#pragma omp parallel for schedule(static)
            for (long i = 0; i < (long) (bytes >> 12); i++) p[(size_t) i << 12] = 0;
        }
        double t2 = now();
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long) (bytes / 144); i++) { int* r = (int*) (p + (size_t) i * 144 + 104); r[0] = -1; r[1] = 0; }
        double t3 = now();
        const char* names[] = {"posix_memalign 2MB + MADV_HUGEPAGE + parallel touch", "posix_memalign + parallel touch", "mmap MAP_POPULATE", "mmap + MADV_HUGEPAGE + parallel touch"};
        printf("%-52s alloc %.1f ms  touch %.1f ms  ctor-style store pass %.1f ms\n", names[variant], t1 - t0, t2 - t1, t3 - t2);
        if (variant >= 2) munmap(p, bytes); else free(p);
    }
}
