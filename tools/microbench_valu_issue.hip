// VALU issue-rate microbenchmark on gfx950 (profiles/README.md quotes its output): 256 workgroups of 4 or 8
// waves (1 or 2 per SIMD), long unrolled streams of one instruction kind; wall time per instruction per SIMD,
// and wave 0's own shader-clock (clock64) and 100 MHz (wall_clock64) tick counts, which give the clock.
// Build and run on the GPU box: hipcc -O2 --offload-arch=gfx950 -w -o /tmp/valu tools/microbench_valu_issue.hip && /tmp/valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int n, long long* cyc)
{
    f2 a0 = {1.0f + threadIdx.x, 2.0f}, a1 = {3.0f, 4.0f}, a2 = {5.0f, 6.0f}, a3 = {7.0f, 8.0f};
    f2 a4 = {1.5f, 2.5f}, a5 = {3.5f, 4.5f}, a6 = {5.5f, 6.5f}, a7 = {7.5f, 8.5f};
    const f2 m = {0.999f, 1.001f}, c = {1e-3f, -1e-3f};
    long long w0 = wall_clock64();
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {  // pk_fma
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (MODE == 1) {  // pk_mul
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (MODE == 2) {  // scalar fma (v_fma_f32) on .x
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));)
        } else if (MODE == 3) {  // rsq
            REP16(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                         "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));)
        } else if (MODE == 4) {  // pk_add
            REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (MODE == 5) {  // med3
            REP16(asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
                         "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));)
        } else if (MODE == 6) {  // mov dpp wave_shr
            REP16(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         "v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));)
        } else if (MODE == 8) {  // v_add_f32
            REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));)
        } else if (MODE == 9) {  // v_mul_f32
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));)
        } else if (MODE == 10) {  // v_sub_f32 two different sources (like x-differences)
            REP16(asm volatile("v_sub_f32 %0, %1, %0\n v_sub_f32 %1, %2, %1\n v_sub_f32 %2, %3, %2\n v_sub_f32 %3, %4, %3\n"
                         "v_sub_f32 %4, %5, %4\n v_sub_f32 %5, %6, %5\n v_sub_f32 %6, %7, %6\n v_sub_f32 %7, %8, %7\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));)
        } else if (MODE == 11) {  // v_max_f32 + v_min_f32 (clamp in 2 plain ops)
            REP16(asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                         "v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x), "v"(c.x));)
        } else if (MODE == 12) {  // mixed: pk_fma and rsq alternating (co-issue?)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_rsq_f32 %10, %10\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_rsq_f32 %11, %11\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "v"(a1.x), "v"(a5.x));)
        } else if (MODE == 7) {  // pk_mov
            REP16(asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]\n v_pk_mov_b32 %1, %2, %3 op_sel:[1,0]\n v_pk_mov_b32 %2, %3, %4 op_sel:[1,0]\n v_pk_mov_b32 %3, %4, %5 op_sel:[1,0]\n"
                         "v_pk_mov_b32 %4, %5, %6 op_sel:[1,0]\n v_pk_mov_b32 %5, %6, %7 op_sel:[1,0]\n v_pk_mov_b32 %6, %7, %0 op_sel:[1,0]\n v_pk_mov_b32 %7, %0, %1 op_sel:[1,0]\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        }
    }
    long long t1 = clock64();
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
template <int MODE>
void run(const char* name, int nblk, int nthr)
{
    float* out; long long* cyc;
    hipMalloc(&out, 4 * 1024 * 1024); hipMalloc(&cyc, 16);
    const int n = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<nblk, nthr>>>(out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<nblk, nthr>>>(out, n, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cc[2]; hipMemcpy(cc, cyc, 16, hipMemcpyDeviceToHost); long long c = cc[0];
    const double instr = (double)n * 16 * 8;                 // per wave
    const int wps = nthr / 64 / 4 > 0 ? nthr / 64 / 4 : 1;   // waves per SIMD
    printf("[wall_clock64 ticks %lld = %.3f ms @100MHz] ", cc[1], cc[1] / 1e5);
    printf("%-8s blocks %4d thr %4d: %.3f ms; clock64 ticks %lld; ns per instr per SIMD (wall) %.3f; ticks/instr/SIMD %.2f\n", name, nblk, nthr, ms, c,
           ms * 1e6 / (instr * wps), (double)c / (instr * wps));
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int thr : {256, 512}) {
        run<0>("pk_fma", 256, thr); run<1>("pk_mul", 256, thr); run<4>("pk_add", 256, thr); run<2>("fma", 256, thr);
        run<5>("med3", 256, thr); run<3>("rsq", 256, thr); run<6>("mov_dpp", 256, thr); run<7>("pk_mov", 256, thr);
    }
    run<8>("add", 256, 512); run<9>("mul", 256, 512); run<10>("sub2", 256, 512); run<11>("maxmin", 256, 512); run<12>("6pk+2rsq", 256, 512);
    run<0>("pk_fma", 1, 64); run<2>("fma", 1, 64);
    return 0;
}
