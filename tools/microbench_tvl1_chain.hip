// What does the S6 arithmetic of the row pipeline cost when NOTHING but the arithmetic is there?
// (round 3, stall accounting of k_iter_stream<2,8,2>: profiles/README.md quotes the output.)
//
// One wave = one strip of 128 columns (2 pixels per lane, packed math), KH time levels in registers, exactly the
// `level` arithmetic of stream_job in csrc/tvl1.hip (same operations, same order), but
//   MODE 0  chain:      level t + 1 consumes what level t emits in the SAME step (the real kernel's dependency);
//                       per-warp constants in registers, no LDS, no global memory inside the loop
//   MODE 1  decoupled:  level t + 1 consumes what level t emitted in the PREVIOUS step (6 more row registers per level):
//                       the KH levels of a step are independent of each other
//   MODE 2  chain + the constants of every level read from an LDS ring (4 x ds_read_b64 per level, one level ahead)
//   MODE 3  chain, 1-ulp v_sqrt / v_rcp instead of the exact sequences (the fast_math arithmetic)
//   MODE 4  two independent strips per wave, KH/2 levels each, interleaved level by level (same registers as MODE 0)
// Launched as 256 x 4 x WPS one-wave workgroups (WPS = 1, 2 waves per SIMD; register budget pinned with
// amdgpu_waves_per_eu) for `steps` steps.  Prints wall ns and shader cycles per level-row (128 pixel-iterations) per SIMD.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o /tmp/chain tools/microbench_tvl1_chain.hip && /tmp/chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ float sub_s(float a, float b)
{
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float dpp_from_left(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
constexpr float kSqrtReg = 7.888609052210118e-31f;
template <bool FAST>
__device__ __forceinline__ f2 sqrt_pk(f2 s)
{
    if constexpr (FAST) return f2{__builtin_amdgcn_sqrtf(s.x), __builtin_amdgcn_sqrtf(s.y)};
    const f2 y = f2{__builtin_amdgcn_rsqf(s.x), __builtin_amdgcn_rsqf(s.y)};
    const f2 g = s * y, h = y * 0.5f;
    const f2 d = pk_fma(-g, g, s);
    return pk_fma(d, h, g);
}
template <bool FAST>
__device__ __forceinline__ f2 rcp_pk(f2 d)
{
    const f2 r = f2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    if constexpr (FAST) return r;
    const f2 e = pk_fma(-d, r, splat(1.0f));
    return pk_fma(e, r, r);
}
__device__ __forceinline__ f2 diff_back(f2 a)
{
    const float prev = dpp_from_left(a.y);
    return f2{a.x - prev, sub_s(a.y, a.x)};
}
__device__ __forceinline__ f2 diff_fwd(f2 a)
{
    const float right = dpp_from_right(a.x);
    return f2{sub_s(a.y, a.x), right - a.y};
}

struct Lv {  // what a level keeps: the p of its row above (time t) and the new u of that row
    f2 P11, P12, P21, P22, U1, U2;
};
struct Rw {  // a row travelling through the levels
    f2 u1, u2, p11, p12, p21, p22;
};

template <bool FAST>
__device__ __forceinline__ void level(Lv& L, Rw& c, f2 wx, f2 wy, f2 rc, f2 ig, f2 mx, float l_t, float taut, float theta)
{
    const f2 dx11 = diff_back(c.p11), dx21 = diff_back(c.p21);
    const f2 div1 = dx11 + (c.p12 - L.P12);
    const f2 div2 = dx21 + (c.p22 - L.P22);
    const f2 rho = pk_fma(wy, c.u2, pk_fma(wx, c.u1, rc));
    const f2 tt = -rho * ig;
    const f2 fi = f2{__builtin_amdgcn_fmed3f(tt.x, -l_t, l_t), __builtin_amdgcn_fmed3f(tt.y, -l_t, l_t)};
    const f2 v1 = pk_fma(fi, wx, c.u1);
    const f2 v2 = pk_fma(fi, wy, c.u2);
    const f2 n1 = pk_fma(splat(theta), div1, v1);
    const f2 n2 = pk_fma(splat(theta), div2, v2);
    const f2 d1x = diff_fwd(L.U1), d2x = diff_fwd(L.U2);
    const f2 u1x = d1x * mx, u1y = n1 - L.U1;
    const f2 u2x = d2x * mx, u2y = n2 - L.U2;
    const f2 s1 = pk_fma(u1y, u1y, pk_fma(u1x, u1x, splat(kSqrtReg)));
    const f2 s2 = pk_fma(u2y, u2y, pk_fma(u2x, u2x, splat(kSqrtReg)));
    const f2 d1 = pk_fma(splat(taut), sqrt_pk<FAST>(s1), splat(1.0f));
    const f2 d2 = pk_fma(splat(taut), sqrt_pk<FAST>(s2), splat(1.0f));
    f2 q1, q2;
    if constexpr (FAST) {
        q1 = rcp_pk<true>(d1);
        q2 = rcp_pk<true>(d2);
    } else {
        const f2 rinv = rcp_pk<false>(d1 * d2);
        q1 = d2 * rinv;
        q2 = d1 * rinv;
    }
    const f2 o11 = pk_fma(splat(taut), u1x, L.P11) * q1;
    const f2 o12 = pk_fma(splat(taut), u1y, L.P12) * q1;
    const f2 o21 = pk_fma(splat(taut), u2x, L.P21) * q2;
    const f2 o22 = pk_fma(splat(taut), u2y, L.P22) * q2;
    const f2 ou1 = L.U1, ou2 = L.U2;
    L.P11 = c.p11, L.P12 = c.p12, L.P21 = c.p21, L.P22 = c.p22, L.U1 = n1, L.U2 = n2;
    c.u1 = ou1, c.u2 = ou2, c.p11 = o11, c.p12 = o12, c.p21 = o21, c.p22 = o22;
}

template <int MODE, int KH, int WPS>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPS, WPS)))
k_chain(const float* __restrict__ in, float* __restrict__ out, int steps, long long* cyc)
{
    constexpr bool FAST = MODE == 3;
    constexpr int NSTRIP = MODE == 4 ? 2 : 1, KL = KH / NSTRIP;
    __shared__ f2 ring[KH][4][64];
    const int lane = threadIdx.x;
    const float* src = in + (size_t)((blockIdx.x % 2048) * 64 + lane) * 16;  // (the input holds 2048 waves' worth)
    const f2 wx = {src[0], src[1]}, wy = {src[2], src[3]}, rc = {src[4], src[5]}, ig = {src[6], src[7]};
    const f2 mx = {1.0f, lane == 63 ? 0.0f : 1.0f};
    const float l_t = 0.045f, taut = 0.8333333f, theta = 0.3f;
    Lv L[NSTRIP][KL];
    Rw hold[NSTRIP][KL];  // MODE 1: the row level t emitted in the previous step
    Rw c[NSTRIP];
#pragma unroll
    for (int j = 0; j < NSTRIP; ++j) {
        c[j] = Rw{f2{src[8], src[9]}, f2{src[10], src[11]}, f2{src[12], src[13]} * 0.01f, f2{src[14], src[15]} * 0.01f,
                  f2{src[13], src[12]} * 0.01f, f2{src[15], src[14]} * 0.01f};
#pragma unroll
        for (int t = 0; t < KL; ++t) {
            L[j][t] = Lv{c[j].p11, c[j].p12, c[j].p21, c[j].p22, c[j].u1, c[j].u2};
            hold[j][t] = c[j];
        }
    }
    if (MODE == 2) {
#pragma unroll
        for (int t = 0; t < KH; ++t) ring[t][0][lane] = wx, ring[t][1][lane] = wy, ring[t][2][lane] = rc, ring[t][3][lane] = ig;
    }
    const long long t0 = clock64();
    for (int s = 0; s < steps; ++s) {
        if constexpr (MODE == 1) {
            // all KL levels work on rows that were complete at the end of the previous step
            Rw nxt[KL];
#pragma unroll
            for (int t = 0; t < KL; ++t) {
                nxt[t] = hold[0][t];
                level<FAST>(L[0][t], nxt[t], wx, wy, rc, ig, mx, l_t, taut, theta);
            }
#pragma unroll
            for (int t = KL - 1; t > 0; --t) hold[0][t] = nxt[t - 1];
            hold[0][0] = nxt[KL - 1];  // (closes the loop so that nothing is dead code)
        } else if constexpr (MODE == 2) {
            const int s0 = s % KH;
            f2 q[4] = {ring[s0][0][lane], ring[s0][1][lane], ring[s0][2][lane], ring[s0][3][lane]};
#pragma unroll
            for (int t = 0; t < KL; ++t) {
                f2 nq[4] = {q[0], q[1], q[2], q[3]};
                if (t + 1 < KL) {
                    const int sl = (s0 + KH - t - 1) % KH;
#pragma unroll
                    for (int f = 0; f < 4; ++f) nq[f] = ring[sl][f][lane];
                    __builtin_amdgcn_sched_barrier(0);
                }
                level<FAST>(L[0][t], c[0], q[0], q[1], q[2], q[3], mx, l_t, taut, theta);
#pragma unroll
                for (int f = 0; f < 4; ++f) q[f] = nq[f];
            }
        } else {
#pragma unroll
            for (int t = 0; t < KL; ++t)
#pragma unroll
                for (int j = 0; j < NSTRIP; ++j) level<FAST>(L[j][t], c[j], wx, wy, rc, ig, mx, l_t, taut, theta);
        }
    }
    const long long t1 = clock64();
    f2 acc = splat(0.0f);
#pragma unroll
    for (int j = 0; j < NSTRIP; ++j) {
        acc += c[j].u1 + c[j].u2 + c[j].p11 + c[j].p12 + c[j].p21 + c[j].p22;
#pragma unroll
        for (int t = 0; t < KL; ++t) {
            acc += L[j][t].U1 + L[j][t].P11;
            if (MODE == 1) acc += hold[j][t].u1 + hold[j][t].p22;
        }
    }
    out[(blockIdx.x % 2048) * 64 + lane] = acc.x + acc.y;
    if (blockIdx.x == 0 && lane == 0) cyc[0] = t1 - t0;
}

template <int MODE, int KH, int WPS>
void run(const char* name, const float* in, float* out, long long* cyc)
{
    const int steps = 4000, nwaves = 256 * 4 * WPS;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_chain<MODE, KH, WPS><<<nwaves, 64>>>(in, out, 50, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_chain<MODE, KH, WPS><<<nwaves, 64>>>(in, out, steps, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double rows_per_simd = (double)steps * KH * WPS;  // level-rows one SIMD works through
    printf("%-28s KH %2d  waves/SIMD %d: %8.3f ms | per level-row per SIMD: %7.1f ns, %6.1f shader cycles (clock %.2f GHz) | "
           "%6.0f Gpx-it/s chip-wide\n",
           name, KH, WPS, ms, ms * 1e6 / rows_per_simd, (double)c / (steps * KH) / WPS, (double)c / (ms * 1e6),
           1024.0 * 128.0 * rows_per_simd / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main()
{
    float* in;
    float* out;
    long long* cyc;
    const size_t n = (size_t)2048 * 64 * 16;
    hipMalloc(&in, n * 4);
    hipMalloc(&out, 2048 * 64 * 4);
    hipMalloc(&cyc, 16);
    float* h = (float*)malloc(n * 4);
    unsigned x = 12345;
    for (size_t i = 0; i < n; ++i) {
        x = x * 1664525u + 1013904223u;
        h[i] = 0.25f + (x >> 8) * (1.0f / 16777216.0f);
    }
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    run<0, 8, 1>("chain (exact)", in, out, cyc);
    run<0, 8, 2>("chain (exact)", in, out, cyc);
    run<3, 8, 1>("chain (fast sqrt/rcp)", in, out, cyc);
    run<3, 8, 2>("chain (fast sqrt/rcp)", in, out, cyc);
    run<2, 8, 1>("chain + LDS ring reads", in, out, cyc);
    run<2, 8, 2>("chain + LDS ring reads", in, out, cyc);
    run<4, 8, 1>("2 strips x 4 levels", in, out, cyc);
    run<4, 8, 2>("2 strips x 4 levels", in, out, cyc);
    run<1, 5, 1>("decoupled levels", in, out, cyc);
    run<1, 5, 2>("decoupled levels", in, out, cyc);
    run<0, 5, 2>("chain (exact)", in, out, cyc);
    run<0, 4, 2>("chain (exact)", in, out, cyc);
    run<0, 4, 3>("chain (exact)", in, out, cyc);
    run<0, 4, 4>("chain (exact)", in, out, cyc);
    run<0, 10, 2>("chain (exact)", in, out, cyc);
    return 0;
}
