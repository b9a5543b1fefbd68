// Exhaustive check (all 2^32 f32 bit patterns) of short instruction sequences against the
// compiler's IEEE-correct f32 division, on the GPU itself (v_rcp_f32's own roundings matter).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o exact_div exact_div.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

constexpr float kNear = 0.01f, kFar = 100.0f;

__device__ __forceinline__ float fixup(float q, float den, float num) { return __builtin_amdgcn_div_fixupf(q, den, num); }

// candidate A: x / c for the constant c, rc = RN(1/c) computed by the IEEE division
template <int VARIANT>
__device__ __forceinline__ float div_const(float x, float c, float rc)
{
    const float q0 = x * rc;
    const float r0 = __builtin_fmaf(-c, q0, x);
    float q1 = __builtin_fmaf(r0, rc, q0);
    if (VARIANT >= 1) {
        const float r1 = __builtin_fmaf(-c, q1, x);
        q1 = __builtin_fmaf(r1, rc, q1);
    }
    return fixup(q1, c, x);
}
// candidate B: 1 / t
template <int ITERS>
__device__ __forceinline__ float recip(float t)
{
    float r = __builtin_amdgcn_rcpf(t);
#pragma unroll
    for (int i = 0; i < ITERS; i++) {
        const float e = __builtin_fmaf(-t, r, 1.0f);
        r = __builtin_fmaf(e, r, r);
    }
    return fixup(r, t, 1.0f);
}

struct Result { unsigned long long bad[8]; uint32_t first[8]; unsigned long long bad_by_exp[8][256]; };

__device__ void note(Result *res, int k, uint32_t bits, bool ok)
{
    if (ok) return;
    atomicAdd(&res->bad[k], 1ull);
    atomicMin(&res->first[k], bits);
    atomicAdd(&res->bad_by_exp[k][(bits >> 23) & 0xff], 1ull);
}
__device__ bool same(float a, float b)
{
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}

__global__ void k_check(Result *res, float c, float rc, float width, float rwidth)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t bits = (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float ref_c = x / c;
        note(res, 0, bits, same(div_const<0>(x, c, rc), ref_c));
        note(res, 1, bits, same(div_const<1>(x, c, rc), ref_c));
        const float ref_r = 1.0f / x;
        note(res, 2, bits, same(recip<1>(x), ref_r));
        note(res, 3, bits, same(recip<2>(x), ref_r));
        const float ref_w = x / width;
        note(res, 4, bits, same(div_const<0>(x, width, rwidth), ref_w));
    }
}

int main()
{
    Result *d, h;
    (void)hipMalloc(&d, sizeof(Result));
    std::memset(&h, 0, sizeof(h));
    for (int k = 0; k < 8; k++) h.first[k] = 0xffffffffu;
    (void)hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice);
    const float c = (1.0f / kFar) - (1.0f / kNear);
    const float rc = 1.0f / c;
    const float width = 1920.0f, rwidth = 1.0f / width;
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d, c, rc, width, rwidth);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"x/C2 (mul,fma,fma,fixup)", "x/C2 two corrections", "1/x rcp+1 Newton+fixup", "1/x rcp+2 Newton+fixup", "x/1920 (mul,fma,fma,fixup)"};
    std::printf("c = %.9g rc = %.9g\n", c, rc);
    for (int k = 0; k < 5; k++) {
        std::printf("%-30s mismatches %llu first 0x%08x\n", names[k], h.bad[k], h.first[k]);
        if (h.bad[k]) {
            std::printf("   by biased exponent:");
            for (int e = 0; e < 256; e++) if (h.bad_by_exp[k][e]) std::printf(" %d:%llu", e, h.bad_by_exp[k][e]);
            std::printf("\n");
        }
    }
    return 0;
}
