// Exhaustive check of short sqrt sequences against the compiler's IEEE-correct sqrtf on the GPU, and
// a look at v_cvt_pk_u8_f32's rounding.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o exact_sqrt exact_sqrt.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

struct Result { unsigned long long bad[4]; uint32_t first[4]; unsigned long long bad_by_exp[4][256]; };
__device__ void note(Result *res, int k, uint32_t bits, bool ok)
{
    if (ok) return;
    atomicAdd(&res->bad[k], 1ull);
    atomicMin(&res->first[k], bits);
    atomicAdd(&res->bad_by_exp[k][(bits >> 23) & 0xff], 1ull);
}
__device__ bool same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

__device__ __forceinline__ float sqrt_a(float x)  // rsq-based, one Markstein correction
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}
__device__ __forceinline__ float sqrt_b(float x)  // v_sqrt-based, correction with rsq as 1/(2s)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
__device__ __forceinline__ float sqrt_c(float x)  // two corrections
{
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    const float h = 0.5f * y;
    g = __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
    return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}

__global__ void k_check(Result *res)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 31); i += stride) {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float ref = sqrtf(x);
        note(res, 0, bits, same(sqrt_a(x), ref));
        note(res, 1, bits, same(sqrt_b(x), ref));
        note(res, 2, bits, same(sqrt_c(x), ref));
    }
}
__global__ void k_cvt(const float *in, uint32_t *out, int n)
{
    const int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xAABBCCDDu);
}

int main()
{
    Result *d, h;
    (void)hipMalloc(&d, sizeof(Result));
    std::memset(&h, 0, sizeof(h));
    for (int k = 0; k < 4; k++) h.first[k] = 0xffffffffu;
    (void)hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"sqrt: rsq, 2 mul, 2 fma", "sqrt: v_sqrt + rsq correction", "sqrt: rsq + two corrections"};
    for (int k = 0; k < 3; k++) {
        std::printf("%-32s mismatches %llu first 0x%08x\n", names[k], h.bad[k], h.first[k]);
        if (h.bad[k]) {
            std::printf("   by biased exponent:");
            int shown = 0;
            for (int e = 0; e < 256 && shown < 40; e++) if (h.bad_by_exp[k][e]) { std::printf(" %d:%llu", e, h.bad_by_exp[k][e]); shown++; }
            std::printf("\n");
        }
    }
    const float vals[] = {-1.0f, 0.0f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.5f, 254.6f, 255.4f, 255.5f, 255.6f, 300.0f, 1e9f, __builtin_nanf("")};
    const int n = sizeof(vals) / sizeof(vals[0]);
    float *din; uint32_t *dout, hout[32];
    (void)hipMalloc(&din, sizeof(vals)); (void)hipMalloc(&dout, n * 4);
    (void)hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, din, dout, n);
    (void)hipMemcpy(hout, dout, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) std::printf("cvt_pk_u8_f32(%g, byte 1, 0xAABBCCDD) = 0x%08x\n", vals[i], hout[i]);
    return 0;
}
