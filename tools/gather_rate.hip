// Micro-benchmark behind DESIGN.md 5.2: how fast can one CU issue per-lane 16-byte gathers out of an
// L2/MALL-resident array, as a function of (a) how many adjacent lanes share an address ("group"),
// (b) how many lanes of the wave are active, (c) whether the next address depends on the loaded data.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_rate.hip -o build/exp/gather_rate && build/exp/gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int DEP>
__global__ void __launch_bounds__(256, 5) k_gather(const float4 *__restrict__ a, uint32_t n4, int iters, int group, int active_mod,
                                                   float *out) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    if (lane % active_mod != 0) return;
    const uint32_t stream = gid / group;             // lanes of one group walk the same addresses
    uint32_t idx = mix(stream) % n4;
    float best = 1e30f;
    for (int it = 0; it < iters; ++it) {
        const float4 *b = a + (size_t)idx * 4;       // one aligned 64-byte line: 4 records
        const float4 p0 = b[0], p1 = b[1], p2 = b[2], p3 = b[3];
        const float d0 = p0.x * p0.x + p0.y, d1 = p1.x * p1.x + p1.y, d2 = p2.x * p2.x + p2.y, d3 = p3.x * p3.x + p3.y;
        best = fminf(best, fminf(fminf(d0, d1), fminf(d2, d3)));
        uint32_t nx = mix(idx + stream * 2654435761u + it);
        if (DEP) nx += __float_as_uint(p0.w) & 1u;   // the next address needs the data: one round trip per batch
        idx = nx % n4;
    }
    if (best == 123.456f) out[gid] = best;
}


// Same walk, W dwords per record (W = 1..4), 4 consecutive records per batch.
template <int W> struct Rec { float v[W]; };
template <int W>
__global__ void __launch_bounds__(256, 5) k_gather_w(const float *__restrict__ a, uint32_t nb, int iters, int group, float *out) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t stream = gid / group;
    uint32_t idx = mix(stream) % nb;
    float best = 1e30f;
    typedef float vec __attribute__((ext_vector_type(W == 3 ? 3 : W)));
    for (int it = 0; it < iters; ++it) {
        const float *b = a + (size_t)idx * 16;       // batches start on 64-byte lines
        vec p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            __builtin_memcpy(&p[u], b + 4 * u, sizeof(float) * W);   // 16-byte stride: the compiler cannot merge them
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) best = fminf(best, p[u][0] * p[u][0] + p[u][W - 1]);
        idx = (mix(idx + stream * 2654435761u + it) + (__float_as_uint(p[0][0]) & 1u)) % nb;
    }
    if (best == 123.456f) out[gid] = best;
}

// Address path: the same dependent walk with (0) 64-bit per-lane addresses (global_load, vaddr pair), (1) a
// uniform base + 32-bit per-lane offset (global_load ... saddr), (2) buffer_load ... offen (descriptor + 32-bit offset).
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256, 5) k_gather_addr(const float4 *__restrict__ a, uint32_t nb, int iters, int group, float *out) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t stream = gid / group;
    uint32_t idx = mix(stream) % nb;
    float best = 1e30f;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a, 0, nb * 64u, 0x00027000);
    for (int it = 0; it < iters; ++it) {
        float4 p[4];
        if (MODE == 0) {
            const float4 *b = a + (size_t)idx * 4 + (size_t)(it & 0) * (1ull << 33);     // keeps the address 64-bit
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = b[u];
        } else if (MODE == 1) {
            const uint32_t off = idx * 64u;
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u] = *(const float4 *)((const char *)a + (off + 16u * u));
        } else {
            const uint32_t off = idx * 64u;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u4v v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u * u, 0, 0);
                p[u] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) best = fminf(best, p[u].x * p[u].x + p[u].y);
        idx = (mix(idx + stream * 2654435761u + it) + (__float_as_uint(p[0].w) & 1u)) % nb;
    }
    if (best == 123.456f) out[gid] = best;
}

int main() {
    const uint32_t n = 1u << 20;                     // 16 MiB of float4: L2 (4 MiB per XCD) misses, MALL hits
    std::vector<float> h((size_t)n * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 977) * 0.25f;
    float4 *d; float *o;
    CHECK(hipMalloc(&d, (size_t)n * 16)); CHECK(hipMalloc(&o, 4u << 20));
    CHECK(hipMemcpy(d, h.data(), (size_t)n * 16, hipMemcpyHostToDevice));
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount; const double ghz = pr.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz; array 16 MiB; 5 blocks of 256 per CU resident, 20 waves per CU\n", pr.name, cus, ghz);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 256, blocks = cus * 5 * 4;
    printf("%4s %6s %7s | %10s %14s %16s %14s\n", "dep", "group", "active", "us", "cyc/VMEMinstr", "lane-loads/clk/CU", "GB/s (lines)");
    for (int dep = 0; dep < 2; ++dep)
        for (int group : {1, 4, 8, 16, 64})
            for (int am : {1, 2, 4}) {
                float ms = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(e0));
                    if (dep) hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, d, n / 4, iters, group, am, o);
                    else hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(256), 0, 0, d, n / 4, iters, group, am, o);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                }
                const double us = ms * 1e3, cyc = us * 1e-6 * ghz * 1e9;
                const double winstr_per_cu = (double)blocks * 4 * iters * 4 / cus;          // wave-level dwordx4 loads per CU
                const double lane_loads = (double)blocks * 256 / am * iters * 4;
                const double lines = (double)blocks * 256 / am / (group >= am ? group / am : 1) * iters;   // distinct 64-byte lines requested
                printf("%4d %6d %7s | %10.1f %14.1f %16.2f %14.0f\n", dep, group, am == 1 ? "64/64" : am == 2 ? "32/64" : "16/64", us,
                       cyc / winstr_per_cu, lane_loads / cyc / cus, lines * 64 / (us * 1e-6) * 1e-9);
            }
    printf("\nrecord width sweep (dependent walk, all lanes active): cycles per wave-level load instruction per CU\n%6s %8s %8s %8s %8s\n", "group", "dword", "dwordx2", "dwordx3", "dwordx4");
    for (int group : {1, 4, 8, 16, 64}) {
        printf("%6d", group);
        for (int w = 1; w <= 4; ++w) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                const float *df = (const float *)d;
                if (w == 1) hipLaunchKernelGGL(k_gather_w<1>, dim3(blocks), dim3(256), 0, 0, df, n / 4, iters, group, o);
                if (w == 2) hipLaunchKernelGGL(k_gather_w<2>, dim3(blocks), dim3(256), 0, 0, df, n / 4, iters, group, o);
                if (w == 3) hipLaunchKernelGGL(k_gather_w<3>, dim3(blocks), dim3(256), 0, 0, df, n / 4, iters, group, o);
                if (w == 4) hipLaunchKernelGGL(k_gather_w<4>, dim3(blocks), dim3(256), 0, 0, df, n / 4, iters, group, o);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double cyc = ms * 1e-3 * ghz * 1e9;
            printf(" %8.1f", cyc / ((double)blocks * 4 * iters * 4 / cus));
        }
        printf("\n");
    }
    printf("\naddress path (dwordx4, dependent walk, all lanes active): cycles per wave-level load instruction per CU\n%6s %12s %12s %12s\n", "group", "global vaddr", "global saddr", "buffer offen");
    for (int group : {1, 4, 8, 16, 64}) {
        printf("%6d", group);
        for (int m = 0; m < 3; ++m) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                if (m == 0) hipLaunchKernelGGL(k_gather_addr<0>, dim3(blocks), dim3(256), 0, 0, d, n / 4, iters, group, o);
                if (m == 1) hipLaunchKernelGGL(k_gather_addr<1>, dim3(blocks), dim3(256), 0, 0, d, n / 4, iters, group, o);
                if (m == 2) hipLaunchKernelGGL(k_gather_addr<2>, dim3(blocks), dim3(256), 0, 0, d, n / 4, iters, group, o);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double cyc = ms * 1e-3 * ghz * 1e9;
            printf(" %12.1f", cyc / ((double)blocks * 4 * iters * 4 / cus));
        }
        printf("\n");
    }
    return 0;
}
