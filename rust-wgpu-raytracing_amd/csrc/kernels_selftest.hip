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

// Shader clock and f32 VALU issue rate under load, for the roofline accounting of bench.py: every wave runs
// `iters` rounds of eight independent v_fma_f32 (MODE 0) or v_pk_fma_f32 (MODE 1) chains and stamps the shader
// cycle counter (s_memtime, one tick per shader cycle) and the constant 100 MHz counter (s_memrealtime) around
// them (MI355X_MICROARCH.md: in-kernel clock = d s_memtime / d s_memrealtime * 100 MHz).  out[wave] =
// {cycles, realtime ticks}.  The launch puts `waves_per_simd` waves on every SIMD at once.
template <int MODE>
__global__ void __launch_bounds__(256)
k_measure_valu(ulonglong2 *out, uint32_t iters, float s)
{
    float a0 = (float)threadIdx.x, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const f2 sv = {s, s};
    __builtin_amdgcn_sched_barrier(0);
    const uint64_t c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    for (uint32_t i = 0; i < iters; i++) {
        if (MODE == 0) {   // (inline assembly: the compiler would pair adjacent scalar FMAs into v_pk_fma_f32)
            asm volatile("v_fma_f32 %0, %0, %8, %0\n\tv_fma_f32 %1, %1, %8, %1\n\tv_fma_f32 %2, %2, %8, %2\n\tv_fma_f32 %3, %3, %8, %3\n\t"
                         "v_fma_f32 %4, %4, %8, %4\n\tv_fma_f32 %5, %5, %8, %5\n\tv_fma_f32 %6, %6, %8, %6\n\tv_fma_f32 %7, %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        } else {
            p0 = fma2(p0, sv, p0); p1 = fma2(p1, sv, p1); p2 = fma2(p2, sv, p2); p3 = fma2(p3, sv, p3);
            p4 = fma2(p4, sv, p4); p5 = fma2(p5, sv, p5); p6 = fma2(p6, sv, p6); p7 = fma2(p7, sv, p7);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const uint64_t c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    const float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y +
                       p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if ((threadIdx.x & 63u) == 0u || sink == 12345.678f)  // the sink keeps the chains alive
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = make_ulonglong2(c1 - c0, r1 - r0);
}

// One wave that does nothing but watch the two counters for `ticks` ticks of the 100 MHz counter: launched on a
// side stream while frames render, it reports the shader clock the chip runs at UNDER THAT WORKLOAD (a pure FMA
// loop pulls the clock lower than the frame kernel does).
__global__ void __launch_bounds__(64)
k_clock_probe(ulonglong2 *out, uint32_t ticks)
{
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    uint64_t r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) out[0] = make_ulonglong2(c1 - c0, r1 - r0);
}

hipError_t launch_clock_probe(hipStream_t s, ulonglong2 *d_out, uint32_t ticks)
{
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, d_out, ticks);
    return hipGetLastError();
}

hipError_t launch_measure_valu(hipStream_t s, int mode, ulonglong2 *d_out, uint32_t n_workgroups, uint32_t iters)
{
    if (mode == 0) hipLaunchKernelGGL((k_measure_valu<0>), dim3(n_workgroups), dim3(256), 0, s, d_out, iters, 1.0001f);
    else hipLaunchKernelGGL((k_measure_valu<1>), dim3(n_workgroups), dim3(256), 0, s, d_out, iters, 1.0001f);
    return hipGetLastError();
}

hipError_t launch_selftest_exact_math(hipStream_t s, unsigned long long *d_out4, uint32_t normalize_count, uint32_t seed)
{
    hipLaunchKernelGGL(k_selftest_depth, dim3(8192), dim3(256), 0, s, d_out4);
    hipLaunchKernelGGL(k_selftest_normalize, dim3(8192), dim3(256), 0, s, d_out4, normalize_count, seed);
    return hipGetLastError();
}

}  // namespace rwr
