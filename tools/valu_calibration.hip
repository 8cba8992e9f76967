// valu_calibration.hip — how many cycles does one wave64 VALU instruction occupy a gfx950 SIMD?
//
// DESIGN.md prices k_traverse's VALU instructions at 4 cycles per wave64 instruction ("VALU busy 80 %"), the microarchitecture
// guide at 2 (`v_fma_f32`, SIMD-32); the conclusion "only fewer instructions help" depends on which one holds for the
// instructions the traversal kernel actually issues (selects, compares, min / max, integer address arithmetic).
// This program settles it: dependency-free streams of ONE opcode (32 independent destination registers, so neither the
// 4-cycle dependent latency nor the register file ports of a single accumulator limit the issue), at 1, 2, 3 and 4 waves per
// SIMD (one workgroup of 256 x w threads per CU, grid = one workgroup per CU), timed inside the kernel with s_memtime (shader
// cycles) and outside with HIP events. Reported: SIMD cycles per wave-instruction = cycles elapsed / (instructions per wave x
// waves per SIMD). Run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE`
// the same launches give the counter side (tools/run_valu_calibration.sh).
//
// build: hipcc --offload-arch=gfx950 -O2 tools/valu_calibration.hip -o ignis_amd/lib/valu_calibration
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                 \
    do {                                                                                         \
        hipError_t e_ = (x);                                                                     \
        if (e_ != hipSuccess) {                                                                  \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
            std::exit(1);                                                                        \
        }                                                                                        \
    } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kUnroll = 32;   // independent destinations per loop iteration
constexpr int kIters  = 4096; // loop iterations -> 131 072 instructions of the opcode per wave

enum Op { FMA, MUL, ADD, MAXF, CNDMASK, CMP, MOV, LSHL_ADD, AND, MAD_U24, PK_FMA, RCP, FMA_DEP, N_OPS };
static const char* kNames[N_OPS] = { "v_fma_f32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_cndmask_b32", "v_cmp_lt_f32", "v_mov_b32", "v_lshl_add_u32",
                                     "v_and_b32", "v_mad_u32_u24", "v_pk_fma_f32", "v_rcp_f32", "v_fma_f32 (one dependent chain)" };

template <int OP>
__global__ void k_stream(float* out, unsigned long long* cycles, float x, float y)
{
    __shared__ float s_pad[24576]; // 96 KiB: at most one workgroup per CU, so `w` really is the number of waves per SIMD
    s_pad[threadIdx.x] = x;
    float a[kUnroll];
    v2f p[kUnroll / 2];
#pragma unroll
    for (int i = 0; i < kUnroll; ++i)
        a[i] = x * (float)(i + 1) + (float)threadIdx.x;
#pragma unroll
    for (int i = 0; i < kUnroll / 2; ++i)
        p[i] = v2f{ a[2 * i], a[2 * i + 1] };
    const v2f xx = { x, x }, yy = { y, y };
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) {
            if (OP == FMA)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
            else if (OP == MUL)
                asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
            else if (OP == ADD)
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
            else if (OP == MAXF)
                asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
            else if (OP == CNDMASK)
                asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(y) : "vcc");
            else if (OP == CMP)
                asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(y) : "vcc");
            else if (OP == MOV)
                asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(y));
            else if (OP == LSHL_ADD)
                asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(y));
            else if (OP == AND)
                asm volatile("v_and_b32 %0, %1, %0" : "+v"(a[i]) : "v"(y));
            else if (OP == MAD_U24)
                asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
            else if (OP == PK_FMA) {
                if (i < kUnroll / 2) // 16 packed instructions = 32 lane-fmas per iteration
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(xx), "v"(yy));
            } else if (OP == RCP)
                asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            else if (OP == FMA_DEP)
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(x), "v"(y));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i)
        s += a[i];
#pragma unroll
    for (int i = 0; i < kUnroll / 2; ++i)
        s += p[i].x + p[i].y;
    if (s == 12345.678f)
        out[0] = s + s_pad[(threadIdx.x * 7) & 1023]; // keeps the streams alive
    if ((threadIdx.x & 63) == 0)
        atomicMax(&cycles[blockIdx.x], t1 - t0);
}

using Kernel = void (*)(float*, unsigned long long*, float, float);
static Kernel kKernels[N_OPS] = { k_stream<FMA>, k_stream<MUL>, k_stream<ADD>, k_stream<MAXF>, k_stream<CNDMASK>, k_stream<CMP>, k_stream<MOV>,
                                  k_stream<LSHL_ADD>, k_stream<AND>, k_stream<MAD_U24>, k_stream<PK_FMA>, k_stream<RCP>, k_stream<FMA_DEP> };

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("# %s, %d CUs, clockRate %.0f MHz; %d x %d instructions of one opcode per wave; grid = one workgroup per CU (96 KiB of LDS each)\n", prop.gcnArchName, cus, prop.clockRate / 1e3,
                kIters, kUnroll);
    std::printf("# cyc/inst = s_memtime cycles (max over the waves of a workgroup, mean over workgroups) / (instructions per wave x waves per SIMD)\n");
    std::printf("%-34s %6s %12s %10s %10s %10s %12s\n", "opcode", "w/SIMD", "cycles", "cyc/inst", "min", "event_ms", "implied_GHz");
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&out, 4));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * cus));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<unsigned long long> h(cus);
    for (int op = 0; op < N_OPS; ++op) {
        for (int w = 1; w <= 4; ++w) {
            float ms = 0;
            double mean = 0, lo = 0;
            for (int rep = 0; rep < 3; ++rep) { // the last repetition counts (clocks ramped up)
                CHECK(hipMemset(cyc, 0, sizeof(unsigned long long) * cus));
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(kKernels[op], dim3(cus), dim3(256 * w), 0, 0, out, cyc, 1.0000001f, 1e-9f);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * cus, hipMemcpyDeviceToHost));
                mean = 0, lo = (double)h[0];
                for (int i = 0; i < cus; ++i)
                    mean += (double)h[i], lo = (double)h[i] < lo ? (double)h[i] : lo;
                mean /= cus;
            }
            const double insts = (double)kIters * (op == PK_FMA ? kUnroll / 2 : kUnroll);
            std::printf("%-34s %6d %12.0f %10.3f %10.3f %10.4f %12.3f\n", kNames[op], w, mean, mean / (insts * w), lo / (insts * w), ms, mean / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
