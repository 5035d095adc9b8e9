// Developer micro-benchmark (gfx950): issue cost of the VALU instruction classes the step kernel uses, per SIMD.
// Every wavefront runs ITER iterations of 16 independent copies of one instruction (16 destination registers, so the
// dependent-issue latency of a single wavefront does not bound the rate once several wavefronts share a SIMD).
// Reported: SIMD cycles per wave-instruction = elapsed * clock / (ITER * 16 * wavefronts per SIMD), clock from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 2048
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, DECL, BODY, SINK)                                                       \
    __global__ void __launch_bounds__(256) NAME(float* out, unsigned long long* cyc, float seed) { \
        DECL;                                                                                \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                          \
        for (int it = 0; it < ITER; ++it) { BODY; }                                          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                          \
        SINK;                                                                                \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                           \
    }

#define DECL_F float r[16]; for (int i = 0; i < 16; ++i) r[i] = seed + threadIdx.x * 1e-3f + i; float s = seed * 0.5f
#define SINK_F float acc = 0; for (int i = 0; i < 16; ++i) acc += r[i]; if (acc == 12345.0f) out[threadIdx.x] = acc

#define OP_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(s));
KERNEL(k_fma, DECL_F, R16(OP_FMA), SINK_F)
#define OP_MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
KERNEL(k_mul, DECL_F, R16(OP_MUL), SINK_F)
#define OP_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
KERNEL(k_rcp, DECL_F, R16(OP_RCP), SINK_F)
#define OP_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
KERNEL(k_exp, DECL_F, R16(OP_EXP), SINK_F)
#define OP_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[i]));
KERNEL(k_sqrt, DECL_F, R16(OP_SQRT), SINK_F)
#define OP_DPP(i) asm volatile("v_subrev_f32_dpp %0, %0, %1 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(s));
KERNEL(k_dpp, DECL_F, R16(OP_DPP), SINK_F)
#define OP_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(s) : "vcc");
KERNEL(k_cmp, DECL_F, R16(OP_CMP), SINK_F)
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(s) : "s20", "s21");
KERNEL(k_cndmask, DECL_F, R16(OP_CNDMASK), SINK_F)
#define OP_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(s));
KERNEL(k_med3, DECL_F, R16(OP_MED3), SINK_F)
#define OP_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
KERNEL(k_mul_lo_u32, DECL_F, R16(OP_MULLO), SINK_F)
#define OP_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r[i]) : "v"(s));
KERNEL(k_mul_u24, DECL_F, R16(OP_MUL24), SINK_F)
#define OP_CVTI(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[i]));
KERNEL(k_cvt_i32_f32, DECL_F, R16(OP_CVTI), SINK_F)
#define OP_RNDNE(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(r[i]));
KERNEL(k_rndne, DECL_F, R16(OP_RNDNE), SINK_F)
#define OP_READLANE(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(r[i]) : "s20");
KERNEL(k_readlane, DECL_F, R16(OP_READLANE), SINK_F)
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(r[i]) : "v"(s));
KERNEL(k_mov, DECL_F, R16(OP_MOV), SINK_F)
#define OP_SALU(i) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
KERNEL(k_salu, DECL_F, R16(OP_SALU), SINK_F)
// one SALU between every two VALU: do the two units overlap within ONE wavefront?
#define OP_MIX(i) asm volatile("v_fma_f32 %0, %0, %1, %0\n s_add_u32 s20, s20, 1" : "+v"(r[i]) : "v"(s) : "s20", "scc");
KERNEL(k_fma_salu_mix, DECL_F, R16(OP_MIX), SINK_F)

// dependent chains: the same instruction on 1 / 2 / 4 registers (a wavefront alone on its SIMD: what ILP is worth)
#define OP_FMA1(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[0]) : "v"(s));
KERNEL(k_fma_chain1, DECL_F, R16(OP_FMA1), SINK_F)
#define OP_FMA2(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[(i) & 1]) : "v"(s));
KERNEL(k_fma_chain2, DECL_F, R16(OP_FMA2), SINK_F)
#define OP_FMA4(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[(i) & 3]) : "v"(s));
KERNEL(k_fma_chain4, DECL_F, R16(OP_FMA4), SINK_F)
#define OP_RCP1(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[0]));
KERNEL(k_rcp_chain1, DECL_F, R16(OP_RCP1), SINK_F)
// a compare into a scalar register pair consumed by the next instruction's select (the override chain's shape)
#define OP_CMPSEL(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(s) : "vcc");
KERNEL(k_cmp_select_pair, DECL_F, R16(OP_CMPSEL), SINK_F)
// a compare, a scalar test of its mask and a branch never taken (the wave-uniform fast-path tests' shape)
#define OP_CMPBR(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_cmp_eq_u64 vcc, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:" : : "v"(r[i]), "v"(s) : "vcc", "scc");
KERNEL(k_cmp_test_branch, DECL_F, R16(OP_CMPBR), SINK_F)
// the pieces of such a test: scalar compare + branch taken / not taken; branch on vcc directly; exec-mask region
#define OP_SBR_T(i) asm volatile("s_cmp_eq_u32 s20, s20\n s_cbranch_scc1 1f\n s_nop 0\n1:" : : : "s20", "scc");
KERNEL(k_scmp_branch_taken, DECL_F, R16(OP_SBR_T), SINK_F)
#define OP_SBR_N(i) asm volatile("s_cmp_lg_u32 s20, s20\n s_cbranch_scc1 1f\n s_nop 0\n1:" : : : "s20", "scc");
KERNEL(k_scmp_branch_not, DECL_F, R16(OP_SBR_N), SINK_F)
#define OP_VCCBR_T(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_cbranch_vccz 1f\n s_nop 0\n1:" : : "v"(r[i]), "v"(s) : "vcc");
KERNEL(k_vcmp_vccz_taken, DECL_F, R16(OP_VCCBR_T), SINK_F)
#define OP_VCCBR_N(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_cbranch_vccnz 1f\n s_nop 0\n1:" : : "v"(r[i]), "v"(s) : "vcc");
KERNEL(k_vcmp_vccnz_not, DECL_F, R16(OP_VCCBR_N), SINK_F)
#define OP_VCMP_SCMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_cmp_eq_u64 vcc, 0" : : "v"(r[i]), "v"(s) : "vcc", "scc");
KERNEL(k_vcmp_scmp, DECL_F, R16(OP_VCMP_SCMP), SINK_F)
#define OP_EXECZ(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n s_nop 0\n1:\n s_mov_b64 exec, s[20:21]" : : "v"(r[i]), "v"(s) : "vcc", "s20", "s21");
KERNEL(k_saveexec_execz_taken, DECL_F, R16(OP_EXECZ), SINK_F)
// the same region without the branch: the skipped instruction runs with an empty exec mask
#define OP_EXEC_NOBR(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_mov_b32 %0, %0\n s_mov_b64 exec, s[20:21]" : "+v"(r[i]) : "v"(s) : "vcc", "s20", "s21");
KERNEL(k_saveexec_nobranch, DECL_F, R16(OP_EXEC_NOBR), SINK_F)
// VALU result consumed by a scalar instruction (v_readfirstlane -> s_add)
#define OP_RFL_SALU(i) asm volatile("v_readfirstlane_b32 s20, %0\n s_add_u32 s21, s20, 1" : : "v"(r[i]) : "s20", "s21", "scc");
KERNEL(k_readfirstlane_salu, DECL_F, R16(OP_RFL_SALU), SINK_F)
// scalar loads with an immediate wait (kernarg re-reads)
#define OP_SLOAD(i) asm volatile("s_load_dword s20, %0, 0x0\n s_waitcnt lgkmcnt(0)" : : "s"(out) : "s20", "memory");
KERNEL(k_sload_wait, DECL_F, R16(OP_SLOAD), SINK_F)

typedef float v2f __attribute__((ext_vector_type(2)));
#define DECL_P v2f r[16]; for (int i = 0; i < 16; ++i) r[i] = v2f{seed + threadIdx.x * 1e-3f + i, seed + i}; v2f s = {seed * 0.5f, seed}
#define SINK_P float acc = 0; for (int i = 0; i < 16; ++i) acc += r[i][0] + r[i][1]; if (acc == 12345.0f) out[threadIdx.x] = acc
#define OP_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(s));
KERNEL(k_pk_fma, DECL_P, R16(OP_PKFMA), SINK_P)
#define OP_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
KERNEL(k_pk_mul, DECL_P, R16(OP_PKMUL), SINK_P)
#define OP_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
KERNEL(k_pk_add, DECL_P, R16(OP_PKADD), SINK_P)

#define DECL_D double r[16]; for (int i = 0; i < 16; ++i) r[i] = seed + threadIdx.x * 1e-3 + i; double s = seed * 0.5; int k = (int)seed
#define SINK_D double acc = 0; for (int i = 0; i < 16; ++i) acc += r[i]; if (acc == 12345.0) out[threadIdx.x] = (float)acc
#define OP_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(s));
KERNEL(k_add_f64, DECL_D, R16(OP_ADD64), SINK_D)
#define OP_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(r[i]) : "v"(s));
KERNEL(k_fma_f64, DECL_D, R16(OP_FMA64), SINK_D)
#define OP_LDEXP64(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(r[i]) : "v"(k));
KERNEL(k_ldexp_f64, DECL_D, R16(OP_LDEXP64), SINK_D)
#define OP_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[i]) : "v"(s));
KERNEL(k_mul_f64, DECL_D, R16(OP_MUL64), SINK_D)
#define OP_RNDNE64(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(r[i]));
KERNEL(k_rndne_f64, DECL_D, R16(OP_RNDNE64), SINK_D)
#define OP_FMA64S(i) asm volatile("v_fma_f64 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(s) : "s20", "s21");
KERNEL(k_fma_f64_sgpr, DECL_D, R16(OP_FMA64S), SINK_D)
#define OP_FMA64C1(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(r[0]) : "v"(s));
KERNEL(k_fma_f64_chain1, DECL_D, R16(OP_FMA64C1), SINK_D)
#define OP_FMA64C2(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(r[(i) & 1]) : "v"(s));
KERNEL(k_fma_f64_chain2, DECL_D, R16(OP_FMA64C2), SINK_D)
#define DECL_C float r[16]; double q[16]; for (int i = 0; i < 16; ++i) { r[i] = seed + threadIdx.x + i; q[i] = i; } float s = seed
#define SINK_C double acc = 0; for (int i = 0; i < 16; ++i) acc += q[i] + r[i]; if (acc == 12345.0) out[threadIdx.x] = (float)acc
#define OP_CVT64(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(q[i]) : "v"(r[i]));
KERNEL(k_cvt_f64_i32, DECL_C, R16(OP_CVT64), SINK_C)
#define OP_CVT32(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[i]) : "v"(q[i]));
KERNEL(k_cvt_f32_f64, DECL_C, R16(OP_CVT32), SINK_C)
#define OP_CVT64U(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(q[i]) : "v"(r[i]));
KERNEL(k_cvt_f64_u32, DECL_C, R16(OP_CVT64U), SINK_C)
#define OP_CVT64F(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(q[i]) : "v"(r[i]));
KERNEL(k_cvt_f64_f32, DECL_C, R16(OP_CVT64F), SINK_C)
#define OP_CVTI64(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(r[i]) : "v"(q[i]));
KERNEL(k_cvt_i32_f64, DECL_C, R16(OP_CVTI64), SINK_C)
#define OP_CVTU64(i) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(r[i]) : "v"(q[i]));
KERNEL(k_cvt_u32_f64, DECL_C, R16(OP_CVTU64), SINK_C)
#define OP_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, 11" : "+v"(r[i]) : "v"(s));
KERNEL(k_alignbit, DECL_F, R16(OP_ALIGNBIT), SINK_F)
#define OP_BFREV(i) asm volatile("v_bfrev_b32 %0, %0" : "+v"(r[i]));
KERNEL(k_bfrev, DECL_F, R16(OP_BFREV), SINK_F)
#define OP_ADDSAT(i) asm volatile("v_add_i32 %0, %0, %1 clamp" : "+v"(r[i]) : "v"(s));
KERNEL(k_add_i32_clamp, DECL_F, R16(OP_ADDSAT), SINK_F)

template <typename F>
static void run(const char* name, F kernel, int waves_per_simd, float* out, unsigned long long* cyc, int n_cu) {
    // 256 threads = 4 wavefronts = one per SIMD of a CU; `waves_per_simd` workgroups per CU
    const int blocks = n_cu * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // per-SIMD cycles per wave-instruction from the wavefront's own clock (includes the loop's SALU)
    // wave 0 is the oldest wavefront of its SIMD (issue arbitration favours it): its own clock gives the cost of an instruction
    // to ONE wavefront; the elapsed time gives the SIMD's throughput with `waves_per_simd` wavefronts sharing it
    printf("%-18s waves/SIMD %d: %6.2f cycles per instruction in wave 0 | %7.2f ns per wave-instruction per SIMD (%.1f us)\n", name,
           waves_per_simd, (double)c / ((double)ITER * 16), ms * 1e6 / ((double)ITER * 16 * waves_per_simd), ms * 1e3);
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    printf("%s, %d CUs\n", prop.name, n_cu);
    const bool only64 = getenv("ATC_UBENCH_F64") != nullptr;
    for (int w : {1, 2, 4, 8}) {
#define RUN(k) run(#k, k, w, out, cyc, n_cu)
        RUN(k_fma); RUN(k_fma_f64); RUN(k_fma_f64_sgpr); RUN(k_fma_f64_chain1); RUN(k_fma_f64_chain2); RUN(k_mul_f64); RUN(k_rndne_f64); RUN(k_add_f64);
        RUN(k_cvt_f64_i32); RUN(k_cvt_f64_u32); RUN(k_cvt_f64_f32); RUN(k_cvt_i32_f64); RUN(k_cvt_u32_f64); RUN(k_cvt_f32_f64);
        RUN(k_alignbit); RUN(k_bfrev); RUN(k_add_i32_clamp);
        printf("\n");
        if (only64) continue;
        RUN(k_fma); RUN(k_mul); RUN(k_mov); RUN(k_pk_fma); RUN(k_pk_mul); RUN(k_pk_add); RUN(k_rcp); RUN(k_exp); RUN(k_sqrt);
        RUN(k_dpp); RUN(k_cmp); RUN(k_cndmask); RUN(k_med3); RUN(k_mul_lo_u32); RUN(k_mul_u24); RUN(k_cvt_i32_f32); RUN(k_rndne);
        RUN(k_readlane); RUN(k_add_f64); RUN(k_fma_f64); RUN(k_ldexp_f64); RUN(k_cvt_f64_i32); RUN(k_cvt_f32_f64);
        RUN(k_salu); RUN(k_fma_salu_mix);
        RUN(k_fma_chain1); RUN(k_fma_chain2); RUN(k_fma_chain4); RUN(k_rcp_chain1); RUN(k_cmp_select_pair); RUN(k_cmp_test_branch); RUN(k_sload_wait);
        RUN(k_scmp_branch_taken); RUN(k_scmp_branch_not); RUN(k_vcmp_vccz_taken); RUN(k_vcmp_vccnz_not); RUN(k_vcmp_scmp);
        // (k_saveexec_execz_taken / k_saveexec_nobranch / k_readfirstlane_salu are not run: the first did not return on the GPU box)
        printf("\n");
    }
    return 0;
}
