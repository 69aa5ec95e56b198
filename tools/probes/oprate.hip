// Probe: issue cost of single vector instructions on gfx950 (2 and 4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP6(S) S(r0) S(r1) S(r2) S(r3) S(r4) S(r5)
#define REP48(S) REP6(S) REP6(S) REP6(S) REP6(S) REP6(S) REP6(S) REP6(S) REP6(S)
#define OP_FMA(r) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(x), "v"(y));
#define OP_ADD(r) asm volatile("v_add_f32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_SUB(r) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_MUL(r) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_MAX(r) asm volatile("v_max_f32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_MAX3(r) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(r) : "v"(x), "v"(y));
#define OP_AND(r) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_ANDLIT(r) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(r));
#define OP_ANDSGPR(r) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r) : "s"(sm));
#define OP_SUBSGPR(r) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r) : "s"(sm));
#define OP_ADDLIT(r) asm volatile("v_add_f32 %0, 0x3f99999a, %0" : "+v"(r));
#define OP_ADDINL(r) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(r));
#define OP_FMA3(r) asm volatile("v_fma_f32 %0, %1, %1, -%0" : "+v"(r) : "v"(x));
#define OP_PERMSGPR(r) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(r) : "v"(x), "s"(sm));
#define OP_SUBE64(r) asm volatile("v_sub_f32_e64 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_PERM(r) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(r) : "v"(x), "v"(y));
#define OP_CVTPK(r) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_DOT2C(r) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
#define OP_MOV(r) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(x));
#define OP_CNDMASK(r) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r) : "v"(x));
#define OP_EXP(r) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
#define OP_LOG(r) asm volatile("v_log_f32 %0, %0" : "+v"(r));
#define OP_PKMUL(r) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p##r) : "v"(px));
#define OP_PKFMA(r) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p##r) : "v"(px), "v"(px));
#define OP_PKADD(r) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p##r) : "v"(px));
#define OP_SWAP(r) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r), "+v"(x));
#define OP_LSHL(r) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(r));
#define OP_BFI(r) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(r) : "v"(x), "v"(y));
#define OP_ADDU(r) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_FMAC(r) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
#define OP_MULLEG(r) asm volatile("v_mul_legacy_f32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_CVTF16(r) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_PKADDF16(r) asm volatile("v_pk_add_f16 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_PKMAXF16(r) asm volatile("v_pk_max_f16 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_CNDE64(r) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(r) : "v"(x), "s"(sm64));
#define OP_CNDE64Z(r) asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(r) : "s"(sm64));
#define OP_MULLO(r) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_BCNT(r) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_LSHLOR(r) asm volatile("v_lshl_or_b32 %0, %1, 6, %0" : "+v"(r) : "v"(x));
#define OP_MINI(r) asm volatile("v_min_i32 %0, %1, %0" : "+v"(r) : "v"(x));
#define OP_CVTF64(r) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d##r) : "v"(r));
#define OP_ADDF64(r) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d##r) : "v"(dx));
#define OP_SWAP32(r) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r), "+v"(x));
#define OP_RCP(r) asm volatile("v_rcp_f32 %0, %0" : "+v"(r));
#define OP_DPP(r) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(x));

typedef float f2 __attribute__((ext_vector_type(2)));
#define KERNEL(NAME, OP)                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(float* out, int iters) {          \
        float r0 = threadIdx.x, r1 = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, x = 0.999f, y = 1e-3f; \
        unsigned sm = 0xffff0000u; asm volatile("" : "+s"(sm));                       \
        unsigned long long sm64 = 0x0f0f0f0f0f0f0f0full; asm volatile("" : "+s"(sm64));   \
        double dr0 = r0, dr1 = 1, dr2 = 2, dr3 = 3, dr4 = 4, dr5 = 5, dx = 1.5;      \
        f2 pr0 = {r0, 1}, pr1 = pr0, pr2 = pr0, pr3 = pr0, pr4 = pr0, pr5 = pr0, px = {x, y};  \
        for (int it = 0; it < iters; ++it) { REP48(OP) }                              \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + x + pr0[0] + pr1[1] + pr2[0] + pr3[1] + pr4[0] + pr5[1] + (float)(dr0 + dr1 + dr2 + dr3 + dr4 + dr5); \
    }
#define LIST(X) X(FMA) X(FMAC) X(ADD) X(SUB) X(MUL) X(MULLEG) X(MAX) X(MAX3) X(AND) X(ANDLIT) X(ANDSGPR) X(SUBSGPR) X(SUBE64) X(ADDLIT) X(ADDINL) X(FMA3) X(PERMSGPR) X(LSHL) X(BFI) X(ADDU) X(PERM) X(CVTPK) X(CVTF16) \
    X(DOT2C) X(MOV) X(CNDMASK) X(EXP) X(LOG) X(PKMUL) X(PKFMA) X(PKADD) X(PKADDF16) X(PKMAXF16) X(SWAP) X(DPP) X(CNDE64) X(CNDE64Z) X(MULLO) X(BCNT) X(LSHLOR) X(MINI) X(CVTF64) X(ADDF64) X(SWAP32) X(RCP)
#define DEF(N) KERNEL(N, OP_##N)
LIST(DEF)

void run(const char* name, void (*kern)(float*, int), float* out) {
    const int iters = 5000;
    double res[2];
    for (int w = 0; w < 2; ++w) {
        const int blocks = 256 * (w ? 4 : 2);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 16);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        res[w] = ms * 1e6 / iters / 48 / (w ? 4 : 2);
    }
    printf("%-10s %6.2f ns per wave instruction at 2 waves/SIMD, %6.2f at 4\n", name, res[0], res[1]);
}
int main() {
    float* out; (void)hipMalloc(&out, 4 * 256 * 1024 * 4);
#define RUN(N) run(#N, k_##N, out);
    LIST(RUN)
    return 0;
}
