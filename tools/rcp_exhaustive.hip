// rcp_exhaustive.hip — is v_rcp_f32 plus Newton steps the IEEE quotient 1 / x? Compared for EVERY float32 bit pattern against the
// compiler's correctly rounded division (v_div_scale / v_rcp / fma chain / v_div_fmas / v_div_fixup). Counts the inputs where a
// candidate differs, split by whether 2^-100 <= |x| <= 2^100. Result on MI355X (profiles/r03_rcp_exhaustive.txt): one Newton step
// is exact inside that range. Not used by the kernels: the range guard and its branch cost what the shorter sequence saves.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/rcp_exhaustive.hip -o ignis_amd/lib/rcp_exhaustive
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__device__ __forceinline__ float rcp1(float x)
{
    const float r = __builtin_amdgcn_rcpf(x);
    return __builtin_fmaf(r, __builtin_fmaf(-x, r, 1.0f), r);
}
__device__ __forceinline__ float rcp2(float x)
{
    const float r = rcp1(x);
    return __builtin_fmaf(r, __builtin_fmaf(-x, r, 1.0f), r);
}

__global__ void k_check(unsigned long long* bad /* [4] */)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long b1_in = 0, b1_out = 0, b2_in = 0, b2_out = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        if (x != x)
            continue;
        const float q  = 1.0f / x;
        const float ax = fabsf(x);
        const bool in  = ax >= 0x1p-100f && ax <= 0x1p100f;
        const bool da = __float_as_uint(rcp1(x)) != __float_as_uint(q), db = __float_as_uint(rcp2(x)) != __float_as_uint(q);
        b1_in += da && in, b1_out += da && !in, b2_in += db && in, b2_out += db && !in;
    }
    atomicAdd(&bad[0], b1_in), atomicAdd(&bad[1], b1_out), atomicAdd(&bad[2], b2_in), atomicAdd(&bad[3], b2_out);
}

int main()
{
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 32) != hipSuccess || hipMemset(d, 0, 32) != hipSuccess)
        return 1;
    hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, d);
    unsigned long long h[4] = {};
    if (hipMemcpy(h, d, 32, hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    std::printf("rcp + 1 Newton step : %llu mismatches for 2^-100 <= |x| <= 2^100, %llu outside\n", h[0], h[1]);
    std::printf("rcp + 2 Newton steps: %llu mismatches for 2^-100 <= |x| <= 2^100, %llu outside\n", h[2], h[3]);
    return 0;
}
