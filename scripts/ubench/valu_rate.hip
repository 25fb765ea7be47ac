// Micro-benchmark: wave-instruction issue cost of the f64 VALU ops the traversal uses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 4096
template <int OP>
__global__ void __launch_bounds__(256) k(double* out, double a0, double b0) {
    double x0 = a0 + threadIdx.x, x1 = a0 * 1.1, x2 = a0 * 1.2, x3 = a0 * 1.3, x4 = a0 * 1.4, x5 = a0 * 1.5, x6 = a0*1.6, x7 = a0*1.7;
    const double b = b0;
    float f0 = (float)a0, f1 = f0 * 1.1f, f2 = f0*1.2f, f3 = f0*1.3f;
    for (int i = 0; i < N_ITER; i++) {
        if (OP == 0) { x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b; }
        if (OP == 1) { x0 = x0 * b; x1 = x1 * b; x2 = x2 * b; x3 = x3 * b; x4 = x4 * b; x5 = x5 * b; x6 = x6 * b; x7 = x7 * b; }
        if (OP == 2) { x0 = __builtin_fma(x0, b, b); x1 = __builtin_fma(x1, b, b); x2 = __builtin_fma(x2, b, b); x3 = __builtin_fma(x3, b, b); x4 = __builtin_fma(x4, b, b); x5 = __builtin_fma(x5, b, b); x6 = __builtin_fma(x6, b, b); x7 = __builtin_fma(x7, b, b); }
        if (OP == 3) { x0 = __builtin_fmax(x0, b); x1 = __builtin_fmin(x1, b); x2 = __builtin_fmax(x2, b); x3 = __builtin_fmin(x3, b); x4 = __builtin_fmax(x4, b); x5 = __builtin_fmin(x5, b); x6 = __builtin_fmax(x6, b); x7 = __builtin_fmin(x7, b);
                       asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)); }
        if (OP == 4) { x0 = x0 / b; x1 = x1 / b; x2 = x2 / b; x3 = x3 / b; x4 = x4 / b; x5 = x5 / b; x6 = x6 / b; x7 = x7 / b; }
        if (OP == 5) { x0 = __builtin_sqrt(x0); x1 = __builtin_sqrt(x1); x2 = __builtin_sqrt(x2); x3 = __builtin_sqrt(x3); x4 = __builtin_sqrt(x4); x5 = __builtin_sqrt(x5); x6 = __builtin_sqrt(x6); x7 = __builtin_sqrt(x7); }
        if (OP == 6) { x0 = (double)f0; x1 = (double)f1; x2 = (double)f2; x3 = (double)f3; f0 += 1.f; f1 += 1.f; f2 += 1.f; f3 += 1.f; x4 += x0; x5 += x1; x6 += x2; x7 += x3; }
        if (OP == 7) { f0 = f0 * f1 + f2; f1 = f1 * f2 + f3; f2 = f2 * f3 + f0; f3 = f3 * f0 + f1; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + f0 + f1 + f2 + f3;
}
template <int OP> void run(const char* name, int ops_per_iter, double* d, int waves_per_simd) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.000001, 0.999999);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.000001, 0.999999); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instrs_per_simd = (double)N_ITER * ops_per_iter * waves_per_simd;
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles @2.4GHz)\n", name, waves_per_simd, ms, ms * 1e6 / wave_instrs_per_simd, ms * 1e6 / wave_instrs_per_simd * 2.4);
}
int main() {
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f64", 8, d, w); run<1>("v_mul_f64", 8, d, w); run<2>("v_fma_f64", 8, d, w); run<3>("v_max/min_f64", 8, d, w);
        run<4>("f64 divide (IEEE)", 8, d, w); run<5>("f64 sqrt (IEEE)", 8, d, w); run<6>("cvt_f64_f32+add", 12, d, w); run<7>("v_fma_f32", 4, d, w);
    }
    return 0;
}
