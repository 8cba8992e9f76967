// atomic_denormal.hip — does gfx950's no-return global_atomic_add_f32 (performed in L2) give the bits of a VALU v_add_f32
// read-modify-write, denormals included? k_shade / k_traverse<any hit> sum into per-sample accumulators with it.
// build: hipcc --offload-arch=gfx950 -O2 tools/atomic_denormal.hip -o ignis_amd/lib/atomic_denormal
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_both(float* atomic_dst, float* plain_dst, const float* add, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    unsafeAtomicAdd(&atomic_dst[i], add[i]);
    plain_dst[i] = plain_dst[i] + add[i];
}

int main()
{
    const int n = 1 << 20;
    std::vector<float> start(n), add(n);
    uint32_t s = 12345u;
    auto next  = [&] { s = s * 1664525u + 1013904223u; return s; };
    for (int i = 0; i < n; ++i) {
        uint32_t a = next(), b = next();
        if (i % 4 == 0) // both denormal
            a &= 0x807FFFFFu, b &= 0x807FFFFFu;
        else if (i % 4 == 1) // a tiny normal and its near negation: denormal result
            a = (a & 0x807FFFFFu) | 0x00800000u, b = (a ^ 0x80000000u) ^ (b & 0xFFu);
        else if (i % 4 == 2) // ordinary magnitudes (exponents 2^-20 .. 2^20)
            a = (a & 0x807FFFFFu) | ((107u + (a >> 23) % 40u) << 23), b = (b & 0x807FFFFFu) | ((107u + (b >> 23) % 40u) << 23);
        // else: arbitrary bit patterns (infinities / NaNs included)
        std::memcpy(&start[i], &a, 4);
        std::memcpy(&add[i], &b, 4);
    }
    float *da, *dp, *dd;
    hipMalloc(&da, n * 4), hipMalloc(&dp, n * 4), hipMalloc(&dd, n * 4);
    hipMemcpy(da, start.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dp, start.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dd, add.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_both, dim3(n / 256), dim3(256), 0, 0, da, dp, dd, n);
    std::vector<float> ra(n), rp(n);
    hipMemcpy(ra.data(), da, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(rp.data(), dp, n * 4, hipMemcpyDeviceToHost);
    long diff[4] = { 0, 0, 0, 0 }, nan_only[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < n; ++i) {
        uint32_t x, y;
        std::memcpy(&x, &ra[i], 4), std::memcpy(&y, &rp[i], 4);
        if (x != y) {
            if (ra[i] != ra[i] && rp[i] != rp[i])
                ++nan_only[i % 4];
            else {
                if (diff[i % 4]++ < 3) {
                    uint32_t a, b;
                    std::memcpy(&a, &start[i], 4), std::memcpy(&b, &add[i], 4);
                    std::printf("  class %d: %08x + %08x -> atomic %08x, v_add_f32 %08x\n", i % 4, a, b, x, y);
                }
            }
        }
    }
    const char* names[4] = { "denormal + denormal", "normal - normal -> denormal", "ordinary magnitudes", "arbitrary bit patterns" };
    for (int c = 0; c < 4; ++c)
        std::printf("%-30s %d cases: %ld differ (+ %ld where both are NaNs with different payloads)\n", names[c], n / 4, diff[c], nan_only[c]);
    return 0;
}
