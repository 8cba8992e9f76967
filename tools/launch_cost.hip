// launch_cost.hip — what does launching a wave that finds nothing to do cost on gfx950, and what does it depend on?
//
// The tail kernel's passes (ignis_amd/csrc/device/tail.hip) take 0.19 ms for 3 072 one-wave workgroups even when no path is left:
// 62 ns per launched wave. This program launches kernels that read one word and return, in the shapes that could
// matter -- plain; 10 KiB of LDS per wave; private (scratch) memory; a 168-register budget; all three like k_tail; the same with
// four waves per workgroup -- at 256 / 1 024 / 3 072 / 12 288 waves, and reports microseconds per launch and nanoseconds per wave
// (HIP events around 200 back-to-back launches on one stream).
//
// build: hipcc --offload-arch=gfx950 -O2 tools/launch_cost.hip -o ignis_amd/lib/launch_cost
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                           \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

// every kernel: n = *count (0 here); the body runs only for n != 0, so the features are allocated but never touched
__global__ void __launch_bounds__(256) k_plain(const unsigned* count, unsigned* out)
{
    if (*count)
        out[threadIdx.x] = 1;
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_lds(const unsigned* count, unsigned* out)
{
    __shared__ uint2 s[20][THREADS]; // 10 KiB per wave, like StackOf<64>
    if (*count) {
        s[threadIdx.x % 20][threadIdx.x] = make_uint2(threadIdx.x, 0);
        __syncthreads();
        out[threadIdx.x] = s[(threadIdx.x + 1) % 20][threadIdx.x].x;
    }
}

__global__ void __launch_bounds__(64) k_scratch(const unsigned* count, unsigned* out)
{
    if (*count) {
        volatile unsigned priv[24]; // indexed at run time: lives in scratch
        for (int i = 0; i < 24; ++i)
            priv[i] = i * threadIdx.x;
        out[threadIdx.x] = priv[(out[0] + threadIdx.x) % 24];
    }
}

// the same, but every wave really stores to and loads from its scratch (an idle k_tail wave walks past spill code)
__global__ void __launch_bounds__(64) k_scratch_touch(const unsigned* count, unsigned* out)
{
    volatile unsigned priv[24];
    const unsigned n = *count;
    for (int i = 0; i < 24; ++i)
        priv[i] = i + n;
    if (priv[(n + threadIdx.x) % 24] == 0xFFFFFFFFu)
        out[threadIdx.x] = 1;
}

__global__ void __launch_bounds__(64, 3) k_regs(const unsigned* count, unsigned* out) // 3 waves per SIMD -> a 168-register allocation
{
    if (*count) {
        float r[150];
        for (int i = 0; i < 150; ++i)
            r[i] = (float)out[i] * 1.0001f;
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < 150; ++i)
                r[i] = r[i] * r[(i + 7) % 150] + 0.5f;
        float s = 0;
        for (int i = 0; i < 150; ++i)
            s += r[i];
        out[threadIdx.x] = (unsigned)s;
    }
}

template <int THREADS>
__global__ void __launch_bounds__(THREADS, 3) k_all(const unsigned* count, unsigned* out)
{
    __shared__ uint2 s[20][THREADS];
    if (*count) {
        volatile unsigned priv[24];
        float r[150];
        for (int i = 0; i < 24; ++i)
            priv[i] = i * threadIdx.x;
        for (int i = 0; i < 150; ++i)
            r[i] = (float)out[i] * 1.0001f;
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < 150; ++i)
                r[i] = r[i] * r[(i + 7) % 150] + 0.5f;
        float acc = 0;
        for (int i = 0; i < 150; ++i)
            acc += r[i];
        s[threadIdx.x % 20][threadIdx.x] = make_uint2((unsigned)acc, priv[(out[0] + threadIdx.x) % 24]);
        __syncthreads();
        out[threadIdx.x] = s[(threadIdx.x + 1) % 20][threadIdx.x].x;
    }
}

template <typename F>
static void measure(const char* name, int threads, F launch)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::printf("%-34s", name);
    for (int waves : { 256, 1024, 3072, 12288 }) {
        const int blocks = waves * 64 / threads;
        for (int i = 0; i < 20; ++i)
            launch(blocks);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        for (int i = 0; i < 200; ++i)
            launch(blocks);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        std::printf("  %6d waves %7.1f us %6.1f ns/wave", waves, ms * 1000 / 200, ms * 1e6 / 200 / waves);
    }
    std::printf("\n");
}

int main()
{
    unsigned *count, *out;
    CHECK(hipMalloc(&count, 4));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(count, 0, 4));
    CHECK(hipMemset(out, 0, 4096));
    measure("plain, 1 wave / workgroup", 64, [&](int g) { hipLaunchKernelGGL(k_plain, dim3(g), dim3(64), 0, 0, count, out); });
    measure("10 KiB LDS per wave", 64, [&](int g) { hipLaunchKernelGGL(k_lds<64>, dim3(g), dim3(64), 0, 0, count, out); });
    measure("scratch", 64, [&](int g) { hipLaunchKernelGGL(k_scratch, dim3(g), dim3(64), 0, 0, count, out); });
    measure("scratch, touched by every wave", 64, [&](int g) { hipLaunchKernelGGL(k_scratch_touch, dim3(g), dim3(64), 0, 0, count, out); });
    measure("168 registers", 64, [&](int g) { hipLaunchKernelGGL(k_regs, dim3(g), dim3(64), 0, 0, count, out); });
    measure("LDS + scratch + registers (k_tail)", 64, [&](int g) { hipLaunchKernelGGL(k_all<64>, dim3(g), dim3(64), 0, 0, count, out); });
    measure("the same, 4 waves / workgroup", 256, [&](int g) { hipLaunchKernelGGL(k_all<256>, dim3(g), dim3(256), 0, 0, count, out); });
    measure("plain, 4 waves / workgroup", 256, [&](int g) { hipLaunchKernelGGL(k_plain, dim3(g), dim3(256), 0, 0, count, out); });
    return 0;
}
