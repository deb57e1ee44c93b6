// Issue cost of single instructions on one wavefront per SIMD (gfx950): 4096 dependent repetitions of
// one instruction (16 per asm statement, so that the compiler puts no s_nop between them) between two s_memtime reads; prints shader-clock ticks per instruction
// next to v_fma_f64.  build: hipcc --offload-arch=gfx950 -O3 -o op_cost op_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(t) t "\n" t "\n" t "\n" t "\n" t "\n" t "\n" t "\n" t "\n" t "\n" t "\n" t "\n" t "\n" t "\n" t "\n" t "\n" t "\n"
#define BODY(name, asmtext, ...)                                                              \
    __global__ void __launch_bounds__(256) name(double* out, long long* cyc, double a, int ia) \
    {                                                                                         \
        double x = (double)threadIdx.x * 1e-3 + 1.0, y = a;                                   \
        int n = ia + (int)threadIdx.x % 3; unsigned u = threadIdx.x, w = threadIdx.x * 7u;    \
        long long t0 = clock64();                                                             \
        for (int it = 0; it < 256; ++it) { asm volatile(REP16(asmtext) : __VA_ARGS__); }       \
        long long t1 = clock64();                                                             \
        out[blockIdx.x * 256 + threadIdx.x] = x + y + n + u + w;                              \
        if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                            \
    }
BODY(k_fma, "v_fma_f64 %0, %0, %1, %1", "+v"(x) : "v"(y))
BODY(k_mul, "v_mul_f64 %0, %0, %1", "+v"(x) : "v"(y))
BODY(k_add, "v_add_f64 %0, %0, %1", "+v"(x) : "v"(y))
BODY(k_ldexp, "v_ldexp_f64 %0, %0, %1", "+v"(x) : "v"(n))
BODY(k_rndne, "v_rndne_f64 %0, %0", "+v"(x) : "v"(y))
BODY(k_cvt, "v_cvt_i32_f64 %0, %1", "+v"(n) : "v"(x))
BODY(k_rcp, "v_rcp_f64 %0, %0", "+v"(x) : "v"(y))
BODY(k_cnd, "v_cndmask_b32 %0, %0, %1, vcc", "+v"(u) : "v"(w))
BODY(k_addu, "v_add_u32 %0, %0, %1", "+v"(u) : "v"(w))
BODY(k_accw, "v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0", "+v"(u) : "v"(w) : "a0")
BODY(k_swap, "v_permlane32_swap_b32 %0, %1", "+v"(u), "+v"(w) : )
BODY(k_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "+v"(u) : "v"(w))
BODY(k_smov, "s_mov_b32 s20, 0x12345678", "+v"(u) : "v"(w) : "s20")
BODY(k_smov_fma, "s_mov_b32 s20, 0x12345678\n s_mov_b32 s21, 0x3ff00000\n v_fma_f64 %0, %0, s[20:21], %1", "+v"(x) : "v"(y) : "s20", "s21")
BODY(k_rdlane, "v_readlane_b32 s20, %0, 3", "+v"(u) : "v"(w) : "s20")
BODY(k_wrlane, "v_writelane_b32 %0, s4, 3", "+v"(u) : "v"(w))
BODY(k_movb64, "v_mov_b64 %0, %1", "+v"(x) : "v"(y))
template <class K> void run(const char* name, K k, double* d_out, long long* d_cyc, int per)
{
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d_out, d_cyc, 1.0000001, 1);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d_out, d_cyc, 1.0000001, 1);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s %6.2f ticks per instruction (%d per repetition)\n", name, (double)c / 4096.0 / per, per);
}
int main()
{
    double* d_out; long long* d_cyc;
    (void)hipMalloc(&d_out, 8 * 256 * 256); (void)hipMalloc(&d_cyc, 8);
    run("v_fma_f64", k_fma, d_out, d_cyc, 1); run("v_mul_f64", k_mul, d_out, d_cyc, 1); run("v_add_f64", k_add, d_out, d_cyc, 1);
    run("v_ldexp_f64", k_ldexp, d_out, d_cyc, 1); run("v_rndne_f64", k_rndne, d_out, d_cyc, 1); run("v_cvt_i32_f64", k_cvt, d_out, d_cyc, 1);
    run("v_rcp_f64", k_rcp, d_out, d_cyc, 1); run("v_cndmask_b32", k_cnd, d_out, d_cyc, 1); run("v_add_u32", k_addu, d_out, d_cyc, 1);
    run("accvgpr write+read", k_accw, d_out, d_cyc, 2); run("v_permlane32_swap", k_swap, d_out, d_cyc, 1); run("v_mov_b32_dpp", k_dpp, d_out, d_cyc, 1);
    run("s_mov_b32", k_smov, d_out, d_cyc, 1); run("2 s_mov + v_fma(sgpr)", k_smov_fma, d_out, d_cyc, 3);
    run("v_readlane_b32", k_rdlane, d_out, d_cyc, 1); run("v_writelane_b32", k_wrlane, d_out, d_cyc, 1); run("v_mov_b64", k_movb64, d_out, d_cyc, 1);
    return 0;
}
