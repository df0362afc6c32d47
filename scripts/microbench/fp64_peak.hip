// Measures the sustained vector-fp64 rate of the device (independent FMA / MUL / ADD chains, operands in registers),
// to calibrate the roofline peak the trace kernel is priced against.  Build: hipcc --offload-arch=gfx950 -O3 -o fp64_peak fp64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int OP, int CHAINS>
__global__ void __launch_bounds__(256) k(double* out, double a, double b, int iters)
{
    double x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) x[i] = a + i * 1e-3 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < CHAINS; i++) {
            if (OP == 0) x[i] = __builtin_fma(x[i], a, b);
            else if (OP == 1) x[i] = x[i] * a;
            else if (OP == 2) x[i] = x[i] + b;
            else if (OP == 3) x[i] = 1.0 / x[i];
            else x[i] = __builtin_sqrt(x[i]);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) s += x[i];
    if (s == 12345.678) out[0] = s;
}

template <int OP, int CHAINS>
void run(const char* name, int blocks_per_cu, double flop_per_op)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    double* d;
    hipMalloc(&d, 8);
    const int iters = (OP >= 3) ? 20000 : 200000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = cus * blocks_per_cu;
    k<OP, CHAINS><<<grid, 256>>>(d, 0.999999, 1e-7, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP, CHAINS><<<grid, 256>>>(d, 0.999999, 1e-7, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double) grid * 256 * iters * CHAINS;
    const double waves_per_simd = blocks_per_cu;   // 256 threads = 4 waves = 1 per SIMD
    const double cyc_per_wave_instr = (ms * 1e-3 * prop.clockRate * 1e3) / ((double) iters * CHAINS * waves_per_simd);
    printf("%-10s chains=%d waves/SIMD=%d : %.3f ms  %.2f Tops/s  %.2f TFLOP/s  ~%.2f cycles per wave-instruction per SIMD (at %d MHz nominal)\n", name, CHAINS,
           blocks_per_cu, ms, ops / ms / 1e9, ops * flop_per_op / ms / 1e9, cyc_per_wave_instr, prop.clockRate / 1000);
    hipFree(d);
}

int main()
{
    for (int occ : {1, 2, 4}) {
        if (occ == 1) { run<0, 8>("fma_f64", 1, 2); run<1, 8>("mul_f64", 1, 1); run<2, 8>("add_f64", 1, 1); run<3, 4>("div_f64", 1, 1); run<4, 4>("sqrt_f64", 1, 1); }
        if (occ == 2) { run<0, 8>("fma_f64", 2, 2); run<1, 8>("mul_f64", 2, 1); run<3, 4>("div_f64", 2, 1); }
        if (occ == 4) { run<0, 8>("fma_f64", 4, 2); run<2, 8>("add_f64", 4, 1); run<3, 4>("div_f64", 4, 1); run<4, 4>("sqrt_f64", 4, 1); }
    }
    run<0, 8>("fma_f64", 8, 2);
    run<0, 16>("fma_f64", 8, 2);
    run<3, 4>("div_f64", 8, 1);
    run<4, 4>("sqrt_f64", 8, 1);
    run<0, 1>("fma_dep1", 1, 2);
    run<0, 2>("fma_dep2", 1, 2);
    return 0;
}
