// GPU box: what the host-array (class API) boundary costs -- pinned allocation, registration, H2D / D2H rates, 1.44 GB (1e7 Ray<double>)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const size_t bytes = (size_t) 10000000 * 144;
    void* d = nullptr;
    auto t0 = clk::now();
    CK(hipFree(0));
    printf("hip init                 %8.1f ms\n", ms(t0));
    t0 = clk::now(); CK(hipMalloc(&d, bytes)); printf("hipMalloc 1.44 GB        %8.1f ms\n", ms(t0));
    char* pageable = (char*) malloc(bytes);
    t0 = clk::now(); memset(pageable, 1, bytes); printf("first touch (memset)     %8.1f ms\n", ms(t0));
    t0 = clk::now(); memset(pageable, 2, bytes); printf("memset again             %8.1f ms\n", ms(t0));
    for (int r = 0; r < 2; r++) {
        t0 = clk::now(); CK(hipMemcpy(d, pageable, bytes, hipMemcpyHostToDevice)); printf("H2D pageable             %8.1f ms  %.1f GB/s\n", ms(t0), bytes / ms(t0) / 1e6);
        t0 = clk::now(); CK(hipMemcpy(pageable, d, bytes, hipMemcpyDeviceToHost)); printf("D2H pageable             %8.1f ms  %.1f GB/s\n", ms(t0), bytes / ms(t0) / 1e6);
    }
    void* pinned = nullptr;
    t0 = clk::now(); CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault)); printf("hipHostMalloc 1.44 GB    %8.1f ms\n", ms(t0));
    t0 = clk::now(); memset(pinned, 1, bytes); printf("touch pinned             %8.1f ms\n", ms(t0));
    for (int r = 0; r < 2; r++) {
        t0 = clk::now(); CK(hipMemcpy(d, pinned, bytes, hipMemcpyHostToDevice)); printf("H2D pinned               %8.1f ms  %.1f GB/s\n", ms(t0), bytes / ms(t0) / 1e6);
        t0 = clk::now(); CK(hipMemcpy(pinned, d, bytes, hipMemcpyDeviceToHost)); printf("D2H pinned               %8.1f ms  %.1f GB/s\n", ms(t0), bytes / ms(t0) / 1e6);
    }
    t0 = clk::now(); CK(hipHostFree(pinned)); printf("hipHostFree              %8.1f ms\n", ms(t0));
    t0 = clk::now(); CK(hipHostRegister(pageable, bytes, hipHostRegisterDefault)); printf("hipHostRegister 1.44 GB  %8.1f ms\n", ms(t0));
    for (int r = 0; r < 2; r++) {
        t0 = clk::now(); CK(hipMemcpy(d, pageable, bytes, hipMemcpyHostToDevice)); printf("H2D registered           %8.1f ms  %.1f GB/s\n", ms(t0), bytes / ms(t0) / 1e6);
        t0 = clk::now(); CK(hipMemcpy(pageable, d, bytes, hipMemcpyDeviceToHost)); printf("D2H registered           %8.1f ms  %.1f GB/s\n", ms(t0), bytes / ms(t0) / 1e6);
    }
    t0 = clk::now(); CK(hipHostUnregister(pageable)); printf("hipHostUnregister        %8.1f ms\n", ms(t0));
    // strided 8-byte field scatter on the host (what a one-field write-back costs after an 80 MB D2H)
    double* field = (double*) malloc(10000000 * 8);
    memset(field, 0, 10000000 * 8);
    t0 = clk::now();
    for (size_t i = 0; i < 10000000; i++) memcpy(pageable + i * 144 + 96, &field[i], 8);
    printf("host scatter 1 field     %8.1f ms (1 thread)\n", ms(t0));
    return 0;
}
