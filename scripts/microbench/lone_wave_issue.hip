// What does an extra non-fp64 vector instruction cost a wave that has its SIMD to itself (the strict side launch of the
// hybrid trace)?  One wave per SIMD over the whole chip; independent chains; cycles per wave-instruction at nominal clock.
//   f64   : 8 fp64 FMA chains                      i32 : 8 u32 add chains            f32 : 8 fp32 FMA chains
//   mix   : 8 fp64 FMA chains + 8 u32 add chains interleaved (cycles per PAIR)   sel : 8 fp64 FMA + 8 v_cndmask pairs
// Build: hipcc --offload-arch=gfx950 -O3 -o lone_wave_issue lone_wave_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, double a, double b, unsigned ua, int iters)
{
    double x[8];
    unsigned u[8];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = a + i * 1e-3 + threadIdx.x * 1e-6; u[i] = ua + i + threadIdx.x; f[i] = (float) x[i]; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0 || MODE == 3 || MODE == 4) x[i] = __builtin_fma(x[i], a, b);
            if (MODE == 1 || MODE == 3) u[i] = u[i] * 3u + ua;          // v_mad_u32_u24 / v_mul_lo + add: integer VALU work
            if (MODE == 2) f[i] = __builtin_fmaf(f[i], (float) a, (float) b);
            if (MODE == 4) x[i] = (u[i] & (1u << (it & 31))) ? x[i] : -x[i];   // compare + 2 x v_cndmask on a double
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i] + u[i] + f[i];
    if (s == 12345.678) out[0] = s;
}

template <int MODE>
void run(const char* name)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    double* d;
    hipMalloc(&d, 8);
    const int iters = 100000, grid = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(d, 0.999999, 1e-7, 7u, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(d, 0.999999, 1e-7, 7u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-4s : %.3f ms  ~%.2f nominal cycles per loop slot (8 slots per iteration, %d MHz)\n", name, ms, ms * 1e-3 * prop.clockRate * 1e3 / ((double) iters * 8),
           prop.clockRate / 1000);
    hipFree(d);
}

int main()
{
    run<0>("f64");
    run<1>("i32");
    run<2>("f32");
    run<3>("mix");
    run<4>("sel");
    return 0;
}
