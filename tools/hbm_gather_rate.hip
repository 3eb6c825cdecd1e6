// Micro-benchmark behind DESIGN.md section 5 (round 5): what rate of independent 32-byte gathers does an MI355X sustain out of
// an array that fits no cache (3.2 GB = the PtN records of a 1e8-point target), against the same records read as a stream?
// k_reduce_finalize<PLANE> at 1e8 points gathers one such record per scan point; this is the ceiling it can be held against.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_gather_rate.hip -o build/exp/hbm_gather_rate && build/exp/hbm_gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// MODE 0: record i (a stream); 1: a random record per lane; 2: random 64-record windows, lanes of a wave side by side in a window
// (what a Morton-sorted scan does to a cell-sorted target: neighbours in the scan are neighbours in the array, waves are not)
template <int MODE, int W>
__global__ void __launch_bounds__(256) k_gather32(const float4 *__restrict__ rec, uint32_t nrec, uint64_t n, float *out) {
    float acc = 0.f;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += W * stride) {
        float4 a[W], b[W];
#pragma unroll
        for (int u = 0; u < W; ++u) {
            const uint64_t iu = i + u * stride;
            uint32_t j;
            if (MODE == 0) j = (uint32_t)(iu % nrec);
            else if (MODE == 1) j = mix((uint32_t)iu * 2654435761u + 12345u) % nrec;
            else j = (mix((uint32_t)(iu >> 6) * 2654435761u + 777u) % (nrec >> 6)) * 64u + (uint32_t)(iu & 63);
            const float4 *r = rec + (size_t)j * 2;
            a[u] = r[0]; b[u] = r[1];
        }
#pragma unroll
        for (int u = 0; u < W; ++u) acc += a[u].x * b[u].x + a[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int W>
static void run(const char *name, const float4 *rec, uint32_t nrec, uint64_t n, float *out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 grid(256 * 8), block(256);
    hipLaunchKernelGGL((k_gather32<MODE, W>), grid, block, 0, 0, rec, nrec, n, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_gather32<MODE, W>), grid, block, 0, 0, rec, nrec, n, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    printf("%-44s W=%d  %8.3f ms  %7.2f G gathers/s  %6.2f TB/s of 32-byte records (%5.2f TB/s if every gather moves a 64-byte sector)\n",
           name, W, ms, (double)n / ms / 1e6, (double)n * 32 / ms / 1e9, (double)n * 64 / ms / 1e9);
}

int main() {
    const uint32_t nrec = 100000000u;                 // 3.2 GB
    const uint64_t n = 12500000ull * 4;               // gathers per launch (4 x the 12.5 M-point shard)
    float4 *rec; float *out;
    CHECK(hipMalloc(&rec, (size_t)nrec * 32));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(rec, 0, (size_t)nrec * 32));
    run<0, 2>("stream (record i)", rec, nrec, n, out);
    run<1, 1>("random record per lane", rec, nrec, n, out);
    run<1, 2>("random record per lane", rec, nrec, n, out);
    run<1, 4>("random record per lane", rec, nrec, n, out);
    run<1, 8>("random record per lane", rec, nrec, n, out);
    run<2, 2>("random 64-record window per wave", rec, nrec, n, out);
    run<2, 4>("random 64-record window per wave", rec, nrec, n, out);
    return 0;
}
