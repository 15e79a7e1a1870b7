// Dev microbenchmark (gfx950): issue cost of VALU instruction FORMS (encoding, operand kinds), cycles per wave-instruction
// per SIMD at 4 waves per SIMD, with the shader clock measured in the kernel.  Every form is a hand-written asm block of 64
// independent instructions on 8 accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(fmt)                                                                                                  \
    asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
                 fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
                 fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
                 fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),   \
                   "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])    \
                 : "v"(b), "v"(c), "s"(sc)                                                                            \
                 : "vcc")

typedef float v2f __attribute__((ext_vector_type(2)));

#define F_ADD32(i) "v_add_f32_e32 %" #i ", %16, %" #i "\n"
#define F_ADD64(i) "v_add_f32_e64 %" #i ", %16, %" #i "\n"
#define F_ADDS(i) "v_add_f32_e32 %" #i ", %18, %" #i "\n"
#define F_MUL32(i) "v_mul_f32_e32 %" #i ", %16, %" #i "\n"
#define F_FMAC32(i) "v_fmac_f32_e32 %" #i ", %16, %17\n"
#define F_FMA(i) "v_fma_f32 %" #i ", %16, %17, %" #i "\n"
#define F_FMAS(i) "v_fma_f32 %" #i ", %" #i ", %18, 0.5\n"
#define F_FMAMK(i) "v_fmamk_f32 %" #i ", %16, 0x3f8003a3, %" #i "\n"
#define F_MOVDPP(i) "v_mov_b32_dpp %" #i ", %16 row_mirror row_mask:0xf bank_mask:0xf\n"
#define F_ADDDPP(i) "v_add_f32_dpp %" #i ", %16, %" #i " row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define F_CND32(i) "v_cndmask_b32_e32 %" #i ", %16, %" #i ", vcc\n"
#define F_CND64(i) "v_cndmask_b32_e64 %" #i ", %16, %" #i ", vcc\n"
#define F_CVTSDWA(i) "v_cvt_f32_i32_sdwa %" #i ", sext(%16) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
#define F_SQRT(i) "v_sqrt_f32_e32 %" #i ", %16\n"
#define F_LOG(i) "v_log_f32_e32 %" #i ", %16\n"
#define F_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %19, %20\n"
#define F_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %19\n"
#define F_PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %19\n"
#define F_MAX3(i) "v_max3_f32 %" #i ", %16, %17, %" #i "\n"
#define F_LSHLADD(i) "v_lshl_add_u32 %" #i ", %16, 2, %" #i "\n"
#define F_ADDU32(i) "v_add_u32_e32 %" #i ", %16, %" #i "\n"

template <int FORM>
__global__ void __launch_bounds__(1024) k(float *out, int iters, unsigned long long *clk)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a[8], q[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x + i;
        q[i] = i;
    }
    v2f p[8], pb = {1.0001f, 0.9999f}, pc = {0.5f, 0.25f};
    for (int i = 0; i < 8; ++i) p[i] = v2f{(float)threadIdx.x, (float)i};
    float b = 1.0001f, c = 0.5f, sc = 1.0002f;
    for (int it = 0; it < iters; ++it) {
        if (FORM == 0) BODY(F_ADD32);
        if (FORM == 1) BODY(F_ADD64);
        if (FORM == 2) BODY(F_ADDS);
        if (FORM == 3) BODY(F_MUL32);
        if (FORM == 4) BODY(F_FMAC32);
        if (FORM == 5) BODY(F_FMA);
        if (FORM == 6) BODY(F_FMAS);
        if (FORM == 7) BODY(F_FMAMK);
        if (FORM == 8) BODY(F_MOVDPP);
        if (FORM == 9) BODY(F_ADDDPP);
        if (FORM == 10) BODY(F_CND32);
        if (FORM == 11) BODY(F_CND64);
        if (FORM == 12) BODY(F_CVTSDWA);
        if (FORM == 13) BODY(F_SQRT);
        if (FORM == 14) BODY(F_LOG);
        if (FORM == 18) BODY(F_MAX3);
        if (FORM == 19) BODY(F_LSHLADD);
        if (FORM == 20) BODY(F_ADDU32);
        if (FORM >= 15 && FORM <= 17) {
#define PK(fmt)                                                                                                    \
    asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
                 fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
                 fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
                 fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
                 : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]),   \
                   "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])    \
                 : "v"(b), "v"(c), "s"(sc), "v"(pb), "v"(pc))
            if (FORM == 15) PK(F_PKFMA);
            if (FORM == 16) PK(F_PKADD);
            if (FORM == 17) PK(F_PKMUL);
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + q[i] + p[i].x + p[i].y;
    out[blockIdx.x * 1024 + threadIdx.x] = r;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int FORM>
void run(const char *name)
{
    const int iters = 4000;
    float *out;
    unsigned long long *clk, h[512];
    (void)hipMalloc(&out, (size_t)256 * 1024 * 4);
    (void)hipMalloc(&clk, sizeof(h));
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<FORM>, dim3(256), dim3(1024), 0, 0, out, iters, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<FORM>, dim3(256), dim3(1024), 0, 0, out, iters, clk);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double ghz = 0, cyc = 0;
    for (int i = 0; i < 256; ++i) {
        ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
        cyc += (double)h[2 * i];
    }
    ghz /= 256;
    cyc /= 256;
    // 16 waves per block = 4 per SIMD, 64 instructions per iteration per wave
    printf("%-34s %.3f ms  clock %.2f GHz  -> %.2f cycles per wave-instruction per SIMD\n", name, ms, ghz,
           cyc / ((double)iters * 64 * 4));
    (void)hipFree(out);
    (void)hipFree(clk);
}

int main()
{
    run<0>("v_add_f32_e32 (VOP2, 4 B)");
    run<1>("v_add_f32_e64 (VOP3, 8 B)");
    run<2>("v_add_f32_e32 sgpr src0");
    run<3>("v_mul_f32_e32");
    run<4>("v_fmac_f32_e32 (VOP2)");
    run<5>("v_fma_f32 3 vgpr (VOP3)");
    run<6>("v_fma_f32 vgpr, sgpr, const");
    run<7>("v_fmamk_f32 (VOP2 + literal)");
    run<8>("v_mov_b32_dpp row_mirror");
    run<9>("v_add_f32_dpp row_ror:1");
    run<10>("v_cndmask_b32_e32");
    run<11>("v_cndmask_b32_e64");
    run<12>("v_cvt_f32_i32_sdwa");
    run<13>("v_sqrt_f32");
    run<14>("v_log_f32");
    run<15>("v_pk_fma_f32");
    run<16>("v_pk_add_f32");
    run<17>("v_pk_mul_f32");
    run<18>("v_max3_f32 (VOP3, 3 src)");
    run<19>("v_lshl_add_u32 (VOP3)");
    run<20>("v_add_u32_e32");
    return 0;
}
