// Microbenchmark: what do FETCH_SIZE and the clock say about reading 32-byte records at RANDOM 32-byte-aligned places of a
// buffer far larger than the caches (8 GiB; MALL 256 MiB) — the access pattern of the packet trace kernel's sorted gather
// (two 16-byte loads per lane per ray)?  MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streaming reads
// ("double it"); "other access widths are uncalibrated".  Three kernels over the same number of records:
//   stream   every lane reads consecutive 32-byte records (coalesced: the calibrated case)
//   gather   every lane reads the record a hash of its index points at
//   gather64 the same with 64-byte records (a whole 64-byte piece per lane)
// Run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  for the counter; the program prints GB/s of RECORD bytes.
// hipcc --offload-arch=gfx950 -O3 -o gather32 gather32.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ buf, uint32_t n_records_log2, uint32_t per_thread, float *out)
{
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    float acc = 0.0f;
    for (uint32_t i = 0; i < per_thread; i++) {
        const uint32_t idx = t * per_thread + i;
        if (MODE == 0) {
            const uint32_t r = (blockIdx.x * per_thread + i) * 256u + threadIdx.x;   // consecutive lanes, consecutive records
            const float4 a = buf[2u * r], b = buf[2u * r + 1u];
            acc += a.x + b.w;
        } else if (MODE == 1) {
            const uint32_t r = mix(idx) & ((1u << n_records_log2) - 1u);
            const float4 a = buf[2u * (size_t)r], b = buf[2u * (size_t)r + 1u];
            acc += a.x + b.w;
        } else {
            const uint32_t r = mix(idx) & ((1u << (n_records_log2 - 1u)) - 1u);   // 64-byte records
            const float4 a = buf[4u * (size_t)r], b = buf[4u * (size_t)r + 1u], c = buf[4u * (size_t)r + 2u], d = buf[4u * (size_t)r + 3u];
            acc += a.x + b.w + c.y + d.z;
        }
    }
    out[t] = acc;
}

template <int MODE>
void run(const char *name, const float4 *buf, uint32_t log2n, float *out, uint32_t bytes_per_record)
{
    const uint32_t blocks = 8192, per_thread = 32;   // 67 M records
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, log2n, per_thread, out);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, log2n, per_thread, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double records = (double)blocks * 256 * per_thread;
    printf("%-10s %8.3f ms  %6.1f M records  %7.1f GB/s of record bytes (%u B each)\n", name, ms, records / 1e6, records * bytes_per_record / ms / 1e6,
           bytes_per_record);
}

int main()
{
    const uint32_t log2n = 28;   // 2^28 records of 32 B = 8 GiB
    float4 *buf; float *out;
    if (hipMalloc(&buf, (size_t)32 << log2n) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMemset(buf, 0, (size_t)32 << log2n);
    (void)hipMalloc(&out, 8192u * 256u * sizeof(float));
    run<0>("stream", buf, log2n, out, 32);
    run<1>("gather", buf, log2n, out, 32);
    run<2>("gather64", buf, log2n, out, 64);
    return 0;
}
