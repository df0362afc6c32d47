// How fast does ONE wave, alone on its SIMD (the strict side launch of the hybrid trace), issue fp64 work -- and how much does that depend on
// the instruction-level parallelism of the stream?  A single 64-thread workgroup (plus, optionally, one such workgroup per CU) runs CHAINS
// independent dependency chains of v_fma_f64 / v_mul_f64 / v_rcp_f64; cycles per instruction from s_memrealtime-free wall clock (wall_clock64,
// 100 MHz) against the shader clock reported by the device.
// Build: hipcc --offload-arch=gfx950 -O3 -o lone_wave_ilp lone_wave_ilp.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS, int OP>
__global__ void __launch_bounds__(64) k(double* out, double a, double b, int iters, unsigned long long* ticks)
{
    double x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; i++) x[i] = a + i * 1e-3 + threadIdx.x * 1e-6;
    const unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 16 / CHAINS; rep++)
#pragma unroll
            for (int i = 0; i < CHAINS; i++) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                if (OP == 2) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[i]));
                if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
                if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(reinterpret_cast<unsigned*>(&x[i])[0]) : "v"(threadIdx.x) : "vcc");
            }
    }
    const unsigned long long t1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; i++) s += x[i];
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

template <int CHAINS, int OP>
void run(const char* name, int grid, double mhz)
{
    double* d; unsigned long long* t;
    hipMalloc(&d, 8); hipMalloc(&t, 8);
    const int iters = 200000;
    k<CHAINS, OP><<<grid, 64>>>(d, 0.999999, 1e-7, 1000, t);
    hipDeviceSynchronize();
    k<CHAINS, OP><<<grid, 64>>>(d, 0.999999, 1e-7, iters, t);
    hipDeviceSynchronize();
    unsigned long long ticks = 0;
    hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost);
    const double ns = ticks * 10.0;                       // wall_clock64 counts at 100 MHz
    printf("%-5s chains %2d grid %4d : %.2f ns per instruction = %.2f cycles at %.0f MHz\n", name, CHAINS, grid, ns / (iters * 16.0), ns / (iters * 16.0) * mhz * 1e-3, mhz);
    hipFree(d); hipFree(t);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double mhz = prop.clockRate / 1000.0;
    for (int grid : {1, 50}) {
        run<1, 0>("fma", grid, mhz); run<2, 0>("fma", grid, mhz); run<4, 0>("fma", grid, mhz); run<8, 0>("fma", grid, mhz); run<16, 0>("fma", grid, mhz);
        run<1, 1>("mul", grid, mhz); run<4, 1>("mul", grid, mhz); run<16, 1>("mul", grid, mhz);
        run<1, 3>("add", grid, mhz); run<4, 3>("add", grid, mhz);
        run<1, 2>("rcp", grid, mhz); run<4, 2>("rcp", grid, mhz); run<16, 2>("rcp", grid, mhz);
        run<1, 4>("cnd32", grid, mhz); run<4, 4>("cnd32", grid, mhz); run<16, 4>("cnd32", grid, mhz);
    }
    return 0;
}
