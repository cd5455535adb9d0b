// What clock does the chip hold under a dense bf16 / fp32 MFMA stream?  (round 3: the 14x14 conv kernel's MFMAs-only timing build ran
// its 1 152 MFMAs per SIMD in ~24 us = 21 ns each, although v_mfma_f32_32x32x16_bf16 issues every 32 cycles.)
// One wave per SIMD on every CU, N back-to-back MFMAs on random operands in registers (four independent accumulators), for a few
// milliseconds; wall time per MFMA per SIMD, shader cycles per MFMA (clock64), and the clock = shader cycles / wall time.
//   hipcc -O3 --offload-arch=gfx950 -w -o /tmp/mfmaclk tools/microbench_mfma_clock.hip && /tmp/mfmaclk
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>  // 0: bf16 32x32x16, 1: fp32 32x32x2, 2: bf16 16x16x32
__global__ void __launch_bounds__(256) k(const float* in, float* out, int n, long long* cyc)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = in[(t + 64 * i + r) & 65535];
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) a[e] = (__bf16)in[(t * 8 + e) & 65535], b[e] = (__bf16)in[(t * 8 + e + 4096) & 65535];
    const float fa = in[t & 65535], fb = in[(t + 777) & 65535];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc4[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 4; ++r) acc4[i][r] = in[(t + 4 * i + r) & 65535];
    const long long t0 = clock64();
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                else if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
                else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[i], 0, 0, 0);
            }
    }
    const long long t1 = clock64();
    float s = 0.0f;
    for (int i = 0; i < 4; ++i) {
        for (int r = 0; r < 16; ++r) s += acc[i][r];
        for (int r = 0; r < 4; ++r) s += acc4[i][r];
    }
    out[t] = s;
    if (t == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, const float* in, float* out, long long* cyc, int n, double flop_per_mfma)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE><<<256, 256>>>(in, out, 1000, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, 256>>>(in, out, n, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double mf = 16.0 * n;  // MFMAs per wave = per SIMD
    printf("%-28s %8.3f ms for %.0f MFMAs per SIMD: %6.2f ns and %6.2f shader cycles per MFMA, clock %.2f GHz, %.0f TFLOP/s chip-wide\n", name, ms,
           mf, ms * 1e6 / mf, (double)c / mf, (double)c / (ms * 1e6), flop_per_mfma * mf * 1024 / (ms * 1e-3) / 1e12);
}

int main()
{
    float *in, *out;
    long long* cyc;
    hipMalloc(&in, 65536 * 4);
    hipMalloc(&out, 65536 * 4 * 4);
    hipMalloc(&cyc, 16);
    float* h = (float*)malloc(65536 * 4);
    unsigned x = 99;
    for (int i = 0; i < 65536; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = ((x >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.1f;
    }
    hipMemcpy(in, h, 65536 * 4, hipMemcpyHostToDevice);
    for (int n : {2000, 20000, 200000}) {
        run<0>("bf16 32x32x16", in, out, cyc, n, 32768.0);
        run<2>("bf16 16x16x32", in, out, cyc, n, 16384.0);
        run<1>("fp32 32x32x2", in, out, cyc, n / 2, 4096.0);
    }
    return 0;
}
