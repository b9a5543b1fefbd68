// Self-test of the short exact forms (rwr_device.h): they must return the very bits of the
// expressions they stand in for, on this GPU, for every input of their stated domain.
//   * to_non_linear_depth_fast vs to_non_linear_depth, sqrt_fast vs sqrtf: ALL 2^32 float bit patterns
//     are visited and every one inside the form's domain is compared (scalar and two-wide forms);
//   * normalize3_fast vs normalize3: 2^30 pseudo-random vectors spread over the whole domain of
//     normalize_fast_domain and a little beyond it (exponents 2^-44 .. 2^43 per component, all sign combinations;
//     vectors outside the domain are skipped, as the kernel skips them), plus the
//     two-wide form.
// Run through rwr_selftest_exact_math() by tests/test_gpu_exact_math.py.
#include "rwr_device_p2.h"

namespace rwr {

RWR_DEV bool same_bits(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

__global__ void __launch_bounds__(256)
k_selftest_depth(unsigned long long *out)  // out[0] compared, out[1] mismatches
{
    unsigned long long n = 0, bad = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float t = __uint_as_float((uint32_t)i);
        if (t >= 0x1p-100f && t <= 0x1p100f) {  // sqrt_fast's range
            const f2 sq = sqrt_fast(f2{t, t});
            const float want_sq = sqrtf(t);
            bad += !(same_bits(sqrt_fast(t), want_sq) && same_bits(sq.x, want_sq) && same_bits(sq.y, want_sq));
        }
        if (!depth_fast_domain(t)) continue;
        const float want = to_non_linear_depth(t);
        const f2 pair = to_non_linear_depth_fast(f2{t, t});
        n++;
        bad += !(same_bits(to_non_linear_depth_fast(t), want) && same_bits(pair.x, want) && same_bits(pair.y, want));
    }
    atomicAdd(&out[0], n);
    atomicAdd(&out[1], bad);
}

// component k of pseudo-random vector i: sign, exponent in [-44, 43], 23 random mantissa bits
RWR_DEV float selftest_component(uint32_t i, uint32_t k, uint32_t seed)
{
    const uint32_t h = rng_hash(i, k, 0u, seed), g = rng_hash(i, k, 1u, seed);
    const uint32_t expo = 127u - 44u + g % 88u;
    return __uint_as_float((h & 0x80000000u) | (expo << 23) | (h & 0x007fffffu));
}

__global__ void __launch_bounds__(256)
k_selftest_normalize(unsigned long long *out, uint32_t count, uint32_t seed)  // out[2] compared, out[3] mismatches
{
    unsigned long long n = 0, bad = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        f3 a = mk3(selftest_component(i, 0u, seed), selftest_component(i, 1u, seed), selftest_component(i, 2u, seed));
        if ((i & 7u) == 0u) {  // every eighth vector: comparable magnitudes (what a camera produces)
            const float s = __builtin_fabsf(a.x);
            a.y = __builtin_copysignf(s * (1.0f + (float)(i >> 3 & 1023u) * 0x1p-10f), a.y);
            a.z = __builtin_copysignf(s * (0.5f + (float)(i >> 13 & 1023u) * 0x1p-11f), a.z);
        }
        if (!normalize_fast_domain(a)) continue;
        const f3 want = normalize3(a), got = normalize3_fast(a);
        const v3 a2 = v3{f2{a.x, a.y}, f2{a.y, a.z}, f2{a.z, a.x}};  // two different vectors side by side
        const bool ok2 = normalize_fast_domain(a2);
        const v3 got2 = normalize3_fast(a2);
        const f3 want2 = normalize3(mk3(a.y, a.z, a.x));
        n++;
        bool ok = same_bits(got.x, want.x) && same_bits(got.y, want.y) && same_bits(got.z, want.z);
        ok = ok && ok2 && same_bits(got2.x.x, want.x) && same_bits(got2.y.x, want.y) && same_bits(got2.z.x, want.z);
        ok = ok && same_bits(got2.x.y, want2.x) && same_bits(got2.y.y, want2.y) && same_bits(got2.z.y, want2.z);
        bad += !ok;
    }
    atomicAdd(&out[2], n);
    atomicAdd(&out[3], bad);
}

hipError_t launch_selftest_exact_math(hipStream_t s, unsigned long long *d_out4, uint32_t normalize_count, uint32_t seed)
{
    hipLaunchKernelGGL(k_selftest_depth, dim3(8192), dim3(256), 0, s, d_out4);
    hipLaunchKernelGGL(k_selftest_normalize, dim3(8192), dim3(256), 0, s, d_out4, normalize_count, seed);
    return hipGetLastError();
}

}  // namespace rwr
