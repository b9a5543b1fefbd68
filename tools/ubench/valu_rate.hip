// Microbenchmark: issue rate of plain vs packed f32 VALU on gfx950 (informs the
// kernel design: DESIGN.md "VALU budget").  hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    float2v sv = {s, s};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // v_fma_f32 x8
            a0 = __builtin_fmaf(a0, s, a0); a1 = __builtin_fmaf(a1, s, a1); a2 = __builtin_fmaf(a2, s, a2); a3 = __builtin_fmaf(a3, s, a3);
            a4 = __builtin_fmaf(a4, s, a4); a5 = __builtin_fmaf(a5, s, a5); a6 = __builtin_fmaf(a6, s, a6); a7 = __builtin_fmaf(a7, s, a7);
        } else if (MODE == 1) {  // v_mul_f32 + v_add_f32 x4
            a0 = a0 * s; a1 = a1 + s; a2 = a2 * s; a3 = a3 + s; a4 = a4 * s; a5 = a5 + s; a6 = a6 * s; a7 = a7 + s;
        } else if (MODE == 2) {  // v_pk_fma_f32 x8
            p0 = __builtin_elementwise_fma(p0, sv, p0); p1 = __builtin_elementwise_fma(p1, sv, p1);
            p2 = __builtin_elementwise_fma(p2, sv, p2); p3 = __builtin_elementwise_fma(p3, sv, p3);
            p4 = __builtin_elementwise_fma(p4, sv, p4); p5 = __builtin_elementwise_fma(p5, sv, p5);
            p6 = __builtin_elementwise_fma(p6, sv, p6); p7 = __builtin_elementwise_fma(p7, sv, p7);
        } else {  // v_pk_mul_f32 / v_pk_add_f32 x8
            p0 = p0 * sv; p1 = p1 + sv; p2 = p2 * sv; p3 = p3 + sv; p4 = p4 * sv; p5 = p5 + sv; p6 = p6 * sv; p7 = p7 + sv;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y +
                                                 p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}

template <int MODE>
void run(const char *name, float *d, int wpb)
{
    const int blocks = 256 * 8, iters = 4096;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs * iters * 8
    const double winstr = (double)blocks * 4 / 1024.0 * iters * 8;
    printf("%-28s %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / winstr, ms * 1e6 / winstr * 2.4);
}

int main()
{
    float *d;
    (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", d, 4);
    run<1>("v_mul_f32/v_add_f32", d, 4);
    run<2>("v_pk_fma_f32", d, 4);
    run<3>("v_pk_mul_f32/v_pk_add_f32", d, 4);
    return 0;
}
