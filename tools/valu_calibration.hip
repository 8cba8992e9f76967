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

enum Op { FMA, MUL, ADD, MAXF, CNDMASK, CMP, MOV, LSHL_ADD, AND, MAD_U24, PK_FMA, RCP, FMA_DEP, CNDMASK_SGPR, CMP_CNDMASK, MAX3, CMP_SGPR, ADD_U32, FMAC, S_AND, N_OPS };
static const char* kNames[N_OPS] = { "v_fma_f32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_cndmask_b32", "v_cmp_lt_f32", "v_mov_b32", "v_lshl_add_u32",
                                     "v_and_b32", "v_mad_u32_u24", "v_pk_fma_f32", "v_rcp_f32", "v_fma_f32 (one dependent chain)", "v_cndmask_b32_e64 (mask in SGPRs)",
                                     "v_cmp_lt_f32 + v_cndmask_b32 (per pair)", "v_max3_f32", "v_cmp_lt_f32_e64 (to SGPRs)", "v_add_u32", "v_fmac_f32", "s_and_b64 (scalar unit)" };

// One loop iteration = ONE asm statement of 32 instructions on registers named in the text (v64 - v95 as destinations, s40 - s47 for
// the scalar rows, all declared clobbered). Between separate asm statements the compiler's hazard recogniser pads with s_nop
// whatever might read vcc or an SGPR a VALU instruction wrote (round 3's rows for v_cmp, v_mov and the "19.5-cycle" v_cndmask had
// such a nop per instruction); inside one statement the stream is exactly what is written.
#define R32(F) F(64) F(65) F(66) F(67) F(68) F(69) F(70) F(71) F(72) F(73) F(74) F(75) F(76) F(77) F(78) F(79) F(80) F(81) F(82) F(83) F(84) F(85) F(86) F(87) F(88) F(89) F(90) F(91) F(92) F(93) F(94) F(95)
#define R16(F) F(64) F(66) F(68) F(70) F(72) F(74) F(76) F(78) F(80) F(82) F(84) F(86) F(88) F(90) F(92) F(94)
#define CLOB "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47"
#define I_FMA(d) "v_fma_f32 v" #d ", %0, %1, v" #d "\n"
#define I_MUL(d) "v_mul_f32 v" #d ", %0, v" #d "\n"
#define I_ADD(d) "v_add_f32 v" #d ", %1, v" #d "\n"
#define I_MAX(d) "v_max_f32 v" #d ", %1, v" #d "\n"
#define I_CND(d) "v_cndmask_b32 v" #d ", v" #d ", %1, vcc\n"
#define I_CMP(d) "v_cmp_lt_f32 vcc, v" #d ", %1\n"
#define I_MOV(d) "v_mov_b32 v" #d ", %1\n"
#define I_LSHL(d) "v_lshl_add_u32 v" #d ", v" #d ", 1, %1\n"
#define I_AND(d) "v_and_b32 v" #d ", %1, v" #d "\n"
#define I_MAD(d) "v_mad_u32_u24 v" #d ", %0, %1, v" #d "\n"
#define I_PK(d) "v_pk_fma_f32 v[" #d ":" #d "+1], %2, %3, v[" #d ":" #d "+1]\n"
#define I_RCP(d) "v_rcp_f32 v" #d ", v" #d "\n"
#define I_DEP(d) "v_fma_f32 v64, %0, %1, v64\n"
#define I_CNDS(d) "v_cndmask_b32_e64 v" #d ", v" #d ", %1, %4\n"
#define I_CMPCND(d) "v_cmp_lt_f32 vcc, v" #d ", %1\nv_cndmask_b32 v" #d ", v" #d ", %0, vcc\n"
#define I_MAX3(d) "v_max3_f32 v" #d ", v" #d ", %0, %1\n"
#define I_CMPS(d) "v_cmp_lt_f32_e64 s[40:41], v" #d ", %1\n"
#define I_ADDU(d) "v_add_u32 v" #d ", %1, v" #d "\n"
#define I_FMAC(d) "v_fmac_f32 v" #d ", %0, %1\n"
#define I_SAND(d) "s_and_b64 s[42:43], s[42:43], %4\n"
#define STREAM(F) asm volatile(R32(F) : : "v"(x), "v"(y), "v"(xx), "v"(yy), "s"(mask) : CLOB)

template <int OP>
__global__ void k_stream(float* out, unsigned long long* cycles, float x, float y)
{
    __shared__ float s_pad[24576]; // 96 KiB: at most one workgroup per CU, so `w` really is the number of waves per SIMD
    s_pad[threadIdx.x] = x;
    const v2f xx = { x, x }, yy = { y, y };
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(threadIdx.x & 1);
    asm volatile("v_cmp_lt_f32 vcc, %0, %1\ns_mov_b64 s[42:43], vcc" : : "v"(x), "v"(y) : CLOB);
    asm volatile(R32(I_MOV) : : "v"(x), "v"(y), "v"(xx), "v"(yy), "s"(mask) : CLOB); // defined values in the stream's registers
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
        if (OP == FMA) STREAM(I_FMA);
        else if (OP == MUL) STREAM(I_MUL);
        else if (OP == ADD) STREAM(I_ADD);
        else if (OP == MAXF) STREAM(I_MAX);
        else if (OP == CNDMASK) STREAM(I_CND);
        else if (OP == CMP) STREAM(I_CMP);
        else if (OP == MOV) STREAM(I_MOV);
        else if (OP == LSHL_ADD) STREAM(I_LSHL);
        else if (OP == AND) STREAM(I_AND);
        else if (OP == MAD_U24) STREAM(I_MAD);
        else if (OP == PK_FMA) asm volatile(R16(I_PK) : : "v"(x), "v"(y), "v"(xx), "v"(yy), "s"(mask) : CLOB); // 16 packed instructions = 32 lane-fmas per iteration
        else if (OP == RCP) STREAM(I_RCP);
        else if (OP == FMA_DEP) STREAM(I_DEP);
        else if (OP == CNDMASK_SGPR) STREAM(I_CNDS);
        else if (OP == CMP_CNDMASK) STREAM(I_CMPCND);
        else if (OP == MAX3) STREAM(I_MAX3);
        else if (OP == CMP_SGPR) STREAM(I_CMPS);
        else if (OP == ADD_U32) STREAM(I_ADDU);
        else if (OP == FMAC) STREAM(I_FMAC);
        else if (OP == S_AND) STREAM(I_SAND);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s;
    asm volatile("v_add_f32 %0, v64, v95" : "=v"(s) : : CLOB);
    if (s == 12345.678f)
        out[0] = s + s_pad[(threadIdx.x * 7) & 1023]; // keeps the streams alive
    if ((threadIdx.x & 63) == 0)
        atomicMax(&cycles[blockIdx.x], t1 - t0);
}

using Kernel = void (*)(float*, unsigned long long*, float, float);
static Kernel kKernels[N_OPS] = { k_stream<FMA>, k_stream<MUL>, k_stream<ADD>, k_stream<MAXF>, k_stream<CNDMASK>, k_stream<CMP>, k_stream<MOV>,
                                  k_stream<LSHL_ADD>, k_stream<AND>, k_stream<MAD_U24>, k_stream<PK_FMA>, k_stream<RCP>, k_stream<FMA_DEP>, k_stream<CNDMASK_SGPR>,
                                  k_stream<CMP_CNDMASK>, k_stream<MAX3>, k_stream<CMP_SGPR>, k_stream<ADD_U32>, k_stream<FMAC>, k_stream<S_AND> };

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
            std::fflush(stdout);
        }
    }
    return 0;
}
