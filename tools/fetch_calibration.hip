// FETCH_SIZE / WRITE_SIZE against known byte counts, for the access patterns of the traversal kernels (VERDICT r04 item 3;
// /opt/skills/guides/MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other
// access widths ... uncalibrated: calibrate on a known byte count in your own access pattern").
//   hipcc --offload-arch=gfx950 -O2 tools/fetch_calibration.hip -o ignis_amd/lib/fetch_calibration
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o pmc -- ignis_amd/lib/fetch_calibration      (tools/run_fetch_calibration.sh)
// Kernels (each prints the bytes it must have fetched from beyond the L2 and its time; the buffer is 4 GiB, 16x the Infinity Cache):
//   k_stream      : every lane reads consecutive 16-byte words, the whole buffer once                     truth = buffer bytes
//   k_gather16    : every lane reads ONE random 16-byte word                                               truth >= 64 or 128 B per word (the line / sector the memory side moves)
//   k_node8       : every lane picks a random 256-byte record and reads 14 of its 16 rows (a Node8 visit: 12 plane rows + 2 id rows)   truth = 256 B per visit (two lines)
//   k_node128     : every lane picks a random 128-byte record and reads 6 of its 8 rows (a quantised-node visit)                       truth = 128 B per visit (one line)
//   k_node8_warm  : k_node8 over a 16 MB window (L2-resident after the first touch)                        truth ~ 0: what the counter shows for hits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ inline uint32_t hash32(uint32_t x)
{
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) k_stream(const float4* __restrict__ p, size_t n, float* sink)
{
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f)
        *sink = acc;
}

__global__ void __launch_bounds__(256) k_gather16(const float4* __restrict__ p, uint32_t words_mask, uint32_t per_lane, float* sink)
{
    float acc          = 0;
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    for (uint32_t k = 0; k < per_lane; ++k) {
        const float4 v = p[hash32(gid * 977u + k * 0x9e3779b9u) & words_mask];
        acc += v.x + v.w;
    }
    if (acc == 12345.678f)
        *sink = acc;
}

template <int RECORD_ROWS, int READ_ROWS>
__global__ void __launch_bounds__(256) k_node(const float4* __restrict__ p, uint32_t records_mask, uint32_t per_lane, float* sink)
{
    float acc          = 0;
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    for (uint32_t k = 0; k < per_lane; ++k) {
        const float4* r = p + (size_t)(hash32(gid * 977u + k * 0x9e3779b9u) & records_mask) * RECORD_ROWS;
        float4 v[READ_ROWS];
#pragma unroll
        for (int j = 0; j < READ_ROWS; ++j)
            v[j] = r[j];
#pragma unroll
        for (int j = 0; j < READ_ROWS; ++j)
            acc += v[j].x + v[j].w;
    }
    if (acc == 12345.678f)
        *sink = acc;
}

template <class F>
static double timed(F&& launch)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main()
{
    const size_t bytes = (size_t)4 << 30;
    float4* buf;
    float* sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    const size_t words    = bytes / 16;
    const unsigned grid   = 256 * 16, lanes = grid * 256;
    const uint32_t per    = 64;
    const double visits   = (double)lanes * per;
    double ms;
    ms = timed([&] { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, 0, buf, words, sink); });
    std::printf("k_stream      truth_bytes %.0f  ms %.3f  GB/s %.0f\n", (double)bytes, ms, bytes / ms / 1e6);
    ms = timed([&] { hipLaunchKernelGGL(k_gather16, dim3(grid), dim3(256), 0, 0, buf, (uint32_t)(words - 1), per, sink); });
    std::printf("k_gather16    gathers %.0f  bytes_if_64B_sectors %.0f  bytes_if_128B_lines %.0f  ms %.3f  Ggathers/s %.2f\n", visits, visits * 64, visits * 128, ms, visits / ms / 1e6);
    ms = timed([&] { hipLaunchKernelGGL((k_node<16, 14>), dim3(grid), dim3(256), 0, 0, buf, (uint32_t)(bytes / 256 - 1), per, sink); });
    std::printf("k_node8       visits %.0f  truth_bytes %.0f  ms %.3f  GB/s %.0f  Gvisits/s %.2f\n", visits, visits * 256, ms, visits * 256 / ms / 1e6, visits / ms / 1e6);
    ms = timed([&] { hipLaunchKernelGGL((k_node<8, 6>), dim3(grid), dim3(256), 0, 0, buf, (uint32_t)(bytes / 128 - 1), per, sink); });
    std::printf("k_node128     visits %.0f  truth_bytes %.0f  ms %.3f  GB/s %.0f  Gvisits/s %.2f\n", visits, visits * 128, ms, visits * 128 / ms / 1e6, visits / ms / 1e6);
    ms = timed([&] { hipLaunchKernelGGL((k_node<16, 14>), dim3(grid), dim3(256), 0, 0, buf, (uint32_t)(((size_t)16 << 20) / 256 - 1), per, sink); });
    std::printf("k_node8_warm  visits %.0f  truth_bytes ~%.0f (a 16 MB window, fetched once)  ms %.3f  Gvisits/s %.2f\n", visits, (double)((size_t)16 << 20), ms, visits / ms / 1e6);
    CHECK(hipDeviceSynchronize());
    return 0;
}
