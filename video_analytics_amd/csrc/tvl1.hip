// TV-L1 dense optical flow for gfx950 (MI355X): hand-written HIP, no CPU fallback.
//
// Replaces the unknown upstream tool that wrote the flow_x_/flow_y_ images the reference reads
// (Sheet03/temporalModel.py:76-81, Sheet03/parameters.py:27,38-39).  The algorithm is the
// published TV-L1 (Zach/Pock/Bischof 2007; Sanchez/Meinhardt-Llopis/Facciolo, IPOL 2013) with
// every open choice fixed in DESIGN.md "TV-L1 specification" (steps S0..S9).  This file must be
// compiled with -ffp-contract=off: the arithmetic contract is IEEE binary32 in the written
// operation order with explicit fmaf only, so that results are independent of the tiling and of
// block_iters, and comparable bit for bit with an independent implementation.
//
// Data layout in HBM (all fp32, row pitch = width rounded up to 4 floats so that a lane's
// 4-pixel run is one aligned 16-byte access):
//   pyramid  [level][frame][3 = I, Ix, Iy][h][pitch]      built once per FRAME, shared by the
//                                                          two pairs that frame belongs to
//   state    2 x [pair][6 = u1,u2,p11,p12,p21,p22][h][pitch]   ping-pong (halo reads vs writes)
//   ro       [pair][4 = I1wx, I1wy, rho_c, 1/|grad|^2][h][pitch]  per-warp constants
//
// Dominant kernels: the inner iterations (S6), in two forms that share their arithmetic operation
// for operation (and so their results, bit for bit):
//   k_iter_tile    One workgroup owns a (64*R) x (NW*C) pixel tile; every thread keeps an R x C
//                  patch of all 10 fields in REGISTERS for `K` consecutive inner iterations
//                  (temporal blocking with a K-pixel overlapped halo).  Horizontal neighbours come
//                  from the adjacent lane by DPP wave shifts, vertical neighbours of the patch's
//                  first/last row from the adjacent wave through a 2-row LDS exchange.
//   k_iter_stream  A pipeline of FOUR waves (one or two in the fallback forms) owns a strip of 128
//                  columns and streams its rows top to bottom ONCE per launch, carrying every row
//                  through K iterations on the way (time-skewed: level t works one row behind level
//                  t-1; a wave hands its rows on to the next through LDS); no y halo.  Used where
//                  the strips are well filled (level_streams()).
// Either way the HBM traffic per pixel-iteration falls from the algorithmic 64 B to about
// 64 B / K / efficiency.
#include "va_internal.h"
#include <cmath>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <type_traits>

namespace {

constexpr int kMaxScales = 16;
constexpr int kMaxRadius = 8;
constexpr int kNF_STATE = 6;
constexpr int kNF_RO = 4;

struct Taps {
    float g[2 * kMaxRadius + 1];
    int R;
};

__device__ __forceinline__ int d_min(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int d_max(int a, int b) { return a > b ? a : b; }

// State / per-warp-constant planes store every aligned run of four pixels as [x0, x0+2, x0+1, x0+3]: a thread of
// k_iter_tile then gets its four pixels as the register pairs E = (x0, x0+2), O = (x0+1, x0+3) straight from one
// 16-byte load, and two of its four x differences per row are a single packed subtract (O - E).
// (`perm` = 0: plain row order, the layout of the levels that k_iter_stream iterates.)
__device__ __forceinline__ int pslot(int x, int perm) { return perm ? ((x & ~3) | ((x & 1) << 1) | ((x >> 1) & 1)) : x; }

// S3 (perm: the image uses the interleaved state layout above)
__device__ __forceinline__ float bilinear(const float* __restrict__ img, int w, int h, int pitch, float x, float y, int perm = 0)
{
    x = fminf(fmaxf(x, 0.0f), (float)(w - 1));
    y = fminf(fmaxf(y, 0.0f), (float)(h - 1));
    const int x0 = (int)x, y0 = (int)y;
    const int x1 = d_min(x0 + 1, w - 1), y1 = d_min(y0 + 1, h - 1);
    const float ax = x - (float)x0, ay = y - (float)y0;
    const int s0 = pslot(x0, perm), s1 = pslot(x1, perm);
    const float a = img[y0 * pitch + s0], b = img[y0 * pitch + s1];
    const float c = img[y1 * pitch + s0], d = img[y1 * pitch + s1];
    const float top = fmaf(ax, b - a, a);
    const float bot = fmaf(ax, d - c, c);
    return fmaf(ay, bot - top, top);
}

// ---------------------------------------------------------------- pyramid kernels ------------

template <typename T>
__global__ void k_frames_to_level0(const T* __restrict__ frames, float* __restrict__ pyr0, int w, int h, int pitch,
                                   size_t plane)
{
    const int f = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= w * h) return;
    const int y = idx / w, x = idx - y * w;
    pyr0[(size_t)f * 3 * plane + (size_t)y * pitch + x] = (float)frames[(size_t)f * w * h + idx];
}

// S1: horizontal / vertical Gaussian; src and dst are [frame][stride_f floats] images of pitch `pitch`.
template <bool VERT>
__global__ void k_gauss(const float* __restrict__ src, size_t src_fstride, float* __restrict__ dst, size_t dst_fstride,
                        int w, int h, int pitch, Taps t)
{
    const int f = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= w * h) return;
    const int y = idx / w, x = idx - y * w;
    const float* s = src + (size_t)f * src_fstride;
    const int R = t.R;
    float acc;
    if (VERT) {
        acc = t.g[0] * s[d_max(y - R, 0) * pitch + x];
        for (int k = -R + 1; k <= R; ++k) acc = fmaf(t.g[k + R], s[d_min(d_max(y + k, 0), h - 1) * pitch + x], acc);
    } else {
        acc = t.g[0] * s[y * pitch + d_max(x - R, 0)];
        for (int k = -R + 1; k <= R; ++k) acc = fmaf(t.g[k + R], s[y * pitch + d_min(d_max(x + k, 0), w - 1)], acc);
    }
    dst[(size_t)f * dst_fstride + (size_t)y * pitch + x] = acc;
}

// S1: bilinear resample of the smoothed level s-1 into level s.
__global__ void k_resample(const float* __restrict__ src, size_t src_fstride, int w, int h, int pitch,
                           float* __restrict__ dst, size_t dst_fstride, int ow, int oh, int opitch)
{
    const int f = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ow * oh) return;
    const int y = idx / ow, x = idx - y * ow;
    const float fx = (float)w / (float)ow, fy = (float)h / (float)oh;
    dst[(size_t)f * dst_fstride + (size_t)y * opitch + x] =
        bilinear(src + (size_t)f * src_fstride, w, h, pitch, (float)x * fx, (float)y * fy);
}

// S2: centred gradient of every frame's level (planes 1,2 of the frame's [3][plane] block).
__global__ void k_grad(float* __restrict__ pyr, int w, int h, int pitch, size_t plane)
{
    const int f = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= w * h) return;
    const int y = idx / w, x = idx - y * w;
    float* base = pyr + (size_t)f * 3 * plane;
    const float* I = base;
    base[plane + (size_t)y * pitch + x] = 0.5f * (I[y * pitch + d_min(x + 1, w - 1)] - I[y * pitch + d_max(x - 1, 0)]);
    base[2 * plane + (size_t)y * pitch + x] = 0.5f * (I[d_min(y + 1, h - 1) * pitch + x] - I[d_max(y - 1, 0) * pitch + x]);
}

// ---------------------------------------------------------------- per-warp kernel ------------

// S3 for the three warped images at once, with the wave supplying half of the taps: consecutive lanes
// are consecutive pixels of a row, and wherever the flow is smooth lane i+1's left taps (x0+1, y0),
// (x0+1, y1) ARE lane i's right taps, so they arrive by a wave shuffle instead of a second gather
// (same values, same arithmetic: the result is bit-identical to bilinear()).
__device__ __forceinline__ void bilinear3_shfl(const float* __restrict__ I1, size_t plane, int w, int h, int pitch, float x,
                                               float y, bool active, float& v0, float& v1, float& v2)
{
    x = fminf(fmaxf(x, 0.0f), (float)(w - 1));
    y = fminf(fmaxf(y, 0.0f), (float)(h - 1));
    const int x0 = (int)x, y0 = (int)y;
    const int x1 = d_min(x0 + 1, w - 1), y1 = d_min(y0 + 1, h - 1);
    const float ax = x - (float)x0, ay = y - (float)y0;
    const int o00 = y0 * pitch + x0, o10 = y1 * pitch + x0;
    float a[3], c[3], b[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a[k] = active ? I1[k * plane + o00] : 0.0f;
        c[k] = active ? I1[k * plane + o10] : 0.0f;
    }
    // the neighbouring lane's left column is my right column iff it starts one pixel to the right on the same rows
    const int nx0 = __shfl_down(x0, 1), ny0 = __shfl_down(y0, 1), ny1 = __shfl_down(y1, 1);
    const bool nact = __shfl_down((int)active, 1) != 0;
    const bool from_lane = (threadIdx.x & 63) != 63 && nact && nx0 == x1 && ny0 == y0 && ny1 == y1;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float na = __shfl_down(a[k], 1), nc = __shfl_down(c[k], 1);
        if (x1 == x0) {  // clamped at the right border: the right taps are the left taps
            b[k] = a[k];
            d[k] = c[k];
        } else if (from_lane) {
            b[k] = na;
            d[k] = nc;
        } else {
            b[k] = active ? I1[k * plane + y0 * pitch + x1] : 0.0f;
            d[k] = active ? I1[k * plane + y1 * pitch + x1] : 0.0f;
        }
    }
    float r[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float top = fmaf(ax, b[k] - a[k], a[k]);
        const float bot = fmaf(ax, d[k] - c[k], c[k]);
        r[k] = fmaf(ay, bot - top, top);
    }
    v0 = r[0];
    v1 = r[1];
    v2 = r[2];
}

// S5.  One thread per pixel of one pair; the 12 bilinear taps of a pixel are 6 L1/L2-served gathers plus
// 6 wave shuffles where the flow is smooth (25 launches per pair against 7500 inner iterations:
// lower-order).
__global__ void k_warp(const float* __restrict__ pyr, size_t plane, int w, int h, int pitch, int fps,
                       const float* __restrict__ stA, const float* __restrict__ stB, const int* __restrict__ sel, int cur,
                       int* __restrict__ base, float* __restrict__ ro, int pair0, int perm)
{
    // XCD-aware workgroup order (as in k_iter_tile): the pixel blocks of a pair gather from one L2
    unsigned lid = blockIdx.y * gridDim.x + blockIdx.x;
    {
        const unsigned nb = gridDim.x * gridDim.y, q = nb / 8, r = nb % 8, xcd = lid % 8, kk = lid / 8;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
    }
    const int pair = pair0 + (int)(lid / gridDim.x);
    const int idx = (int)(lid % gridDim.x) * blockDim.x + threadIdx.x;
    const int which = sel ? sel[pair] : cur;
    if (base && idx == 0) base[pair] = which;
    const bool active = idx < w * h;  // inactive lanes still take part in the shuffles
    const int y = active ? idx / w : 0, x = active ? idx - y * w : 0;
    const int seq = pair / (fps - 1), k = pair - seq * (fps - 1);
    const int f0 = seq * fps + k;
    const float* I0 = pyr + (size_t)f0 * 3 * plane;
    const float* I1 = pyr + (size_t)(f0 + 1) * 3 * plane;
    const float* st = (which ? stB : stA) + (size_t)pair * kNF_STATE * plane;
    const size_t o = (size_t)y * pitch + x, op = (size_t)y * pitch + pslot(x, perm);
    const float u1 = active ? st[op] : 0.0f, u2 = active ? st[plane + op] : 0.0f;
    float Iw, Iwx, Iwy;
    bilinear3_shfl(I1, plane, w, h, pitch, (float)x + u1, (float)y + u2, active, Iw, Iwx, Iwy);
    if (!active) return;
    const float grad = fmaf(Iwy, Iwy, Iwx * Iwx);
    float* r = ro + (size_t)pair * kNF_RO * plane;
    r[op] = Iwx;
    r[plane + op] = Iwy;
    r[2 * plane + op] = fmaf(-Iwy, u2, fmaf(-Iwx, u1, Iw - I0[o]));
    r[3 * plane + op] = grad < 1e-10f ? 0.0f : 1.0f / grad;
}

// S8.  Upsample the coarse flow (cw,ch) of buffer `which` into tmp [pair][2][fplane].  A separate
// target is needed because with the stopping rule different pairs can hold their newest state in
// different ping-pong buffers, and the coarse and fine layouts of different pairs overlap.
__global__ void k_upsample(const float* __restrict__ stA, const float* __restrict__ stB, const int* __restrict__ sel,
                           int cur, int cw, int ch, int cpitch, size_t cplane, float* __restrict__ tmp, int fw, int fh,
                           int fpitch, size_t fplane, float inv_step, int cperm)
{
    const int pair = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= fw * fh) return;
    const int which = sel ? sel[pair] : cur;
    const float* src = (which ? stB : stA) + (size_t)pair * kNF_STATE * cplane;
    float* dst = tmp + (size_t)pair * 2 * fplane;
    const int y = idx / fw, x = idx - y * fw;
    const float rx = (float)cw / (float)fw, ry = (float)ch / (float)fh;
    const size_t o = (size_t)y * fpitch + x;
    dst[o] = bilinear(src, cw, ch, cpitch, (float)x * rx, (float)y * ry, cperm) * inv_step;
    dst[fplane + o] = bilinear(src + cplane, cw, ch, cpitch, (float)x * rx, (float)y * ry, cperm) * inv_step;
}

// S4: start of a level: u from tmp, p = 0, everything in ping-pong buffer 0.
// Also zeroes the pitch padding (columns w .. pitch-1) of the state: k_iter_tile loads whole 4-pixel
// runs, and although no padding value can reach a valid pixel (the forward differences at x = w-1
// are multiplied by 0), a NaN/Inf bit pattern left there by an earlier owner of the workspace would
// (NaN * 0 = NaN).  Everything the kernels themselves write is finite.
__global__ void k_level_init(const float* __restrict__ tmp, float* __restrict__ st0, int w, int h, int pitch, size_t plane, int perm)
{
    const int pair = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pitch * h) return;
    const int y = idx / pitch, x = idx - y * pitch;
    const size_t o = (size_t)y * pitch + x, op = (size_t)y * pitch + pslot(x, perm);  // (pitch is a multiple of 4: pslot stays inside the row)
    const float* t = tmp + (size_t)pair * 2 * plane;
    float* dst = st0 + (size_t)pair * kNF_STATE * plane;
    dst[op] = x < w ? t[o] : 0.0f;
    dst[plane + op] = x < w ? t[plane + o] : 0.0f;
    dst[2 * plane + op] = 0.0f;
    dst[3 * plane + op] = 0.0f;
    dst[4 * plane + op] = 0.0f;
    dst[5 * plane + op] = 0.0f;
}

// Zero the pitch padding of the nf planes of every pair (the per-warp constants: k_warp writes x < w only).
__global__ void k_zero_pad(float* __restrict__ buf, int nf, int w, int h, int pitch, size_t plane, int perm)
{
    const int pair = blockIdx.y, pw = pitch - w;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pw * h * nf) return;
    const int f = idx / (pw * h), r = idx - f * pw * h, y = r / pw, x = w + r - y * pw;
    buf[((size_t)pair * nf + f) * plane + (size_t)y * pitch + pslot(x, perm)] = 0.0f;
}

__global__ void k_flow_out(const float* __restrict__ stA, const float* __restrict__ stB, const int* __restrict__ sel,
                           int cur, int w, int h, int pitch, size_t plane, float* __restrict__ flow, int perm)
{
    const int pair = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= w * h) return;
    const int which = sel ? sel[pair] : cur;
    const float* st = (which ? stB : stA) + (size_t)pair * kNF_STATE * plane;
    const int y = idx / w, x = idx - y * w;
    flow[((size_t)pair * 2) * w * h + idx] = st[(size_t)y * pitch + pslot(x, perm)];
    flow[((size_t)pair * 2 + 1) * w * h + idx] = st[plane + (size_t)y * pitch + pslot(x, perm)];
}

// S9
__global__ void k_flow_to_stack(const float* __restrict__ flow, float* __restrict__ stack, size_t n, float bound,
                                float mean, float stdv)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float t = (255.0f * (flow[i] + bound)) / (2.0f * bound);
    const float q = rintf(fminf(fmaxf(t, 0.0f), 255.0f));
    stack[i] = (q / 255.0f - mean) / stdv;
}

// ---------------------------------------------------------------- inner iterations -----------

struct IterArgs {
    const float* ro;
    const float* stA;
    const float* stB;
    float* outA;
    float* outB;
    const int* base;  // EPS mode: per-pair buffer index at the start of this warp's loop
    int* sel;         // EPS mode: per-pair index of the buffer holding the newest state
    unsigned long long* err;  // EPS mode: [pair][iters] exact integer error sums (S7)
    unsigned long long qthr;
    size_t plane;
    int cur;  // fixed mode: input buffer index
    int w, h, pitch;
    int K;    // iterations in this launch
    int ntx, nty, HX;
    int it, iters;
    int rev;  // odd launches walk the workgroup order backwards: what the previous launch wrote last is read first
    int pair0;  // first pair of this launch (a level's pairs are processed in chunks)
    float l_t, taut, theta;
};

__device__ __forceinline__ float dpp_from_left(float v)  // lane i <- lane i-1, lane 0 <- 0
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_right(float v)  // lane i <- lane i+1, lane 63 <- 0
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

typedef float f2 __attribute__((ext_vector_type(2)));
#ifndef VA_REV
#define VA_REV 1
#endif
#ifndef VA_STREAM_QUEUE_DEFAULT
#define VA_STREAM_QUEUE_DEFAULT 0  // 1: k_iter_stream_q (all passes of a warp step in one launch) wherever it applies
#endif
#ifndef VA_STREAM4_DEFAULT
#define VA_STREAM4_DEFAULT 0  // 1: four jobs per 512-thread workgroup (k_iter_stream4) wherever the two-wave pipeline runs
#endif
#ifndef VA_STREAM_UNROLL2
#define VA_STREAM_UNROLL2 1  // measured on the 224^2 level: 9.8 -> 9.3 ms per warp step of 320 pairs
#endif
#ifndef VA_CHUNK
#define VA_CHUNK 1
#endif
// TIMING BUILDS ONLY (wrong results; never set in a product build): bit set of parts of k_iter_stream's steady steps to
// leave out, for the stall accounting of profiles/README.md (round 3): 1 = the global stores of the last wave, 2 = the
// global loads of the first wave (the rows loaded before the loop are reused), 4 = the hand-over rows (no if_put / if_get),
// 8 = the ring reads (the first row's constants are reused), 16 = the sched_barriers, 32 = the ring writes
#ifndef VA_TIMING_SKIP
#define VA_TIMING_SKIP 0
#endif
#ifndef VA_CHUNK_MB
#define VA_CHUNK_MB 150.0
#endif

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }  // v_pk_fma_f32
// a - b as one v_sub_f32 the vectoriser cannot re-pack (see the x differences in k_iter_tile)
__device__ __forceinline__ float sub_s(float a, float b)
{
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }

// Correctly rounded square root and reciprocal from the hardware seeds with packed FMAs only (one
// Newton step on the exact FMA residual; no denormal scaling, no special-case selects).  Both are
// bit-identical to the IEEE sqrtf() / 1.0f/x of the arithmetic contract on their whole domain:
// va_selftest_exact_math() compares them EXHAUSTIVELY (every float in [2^-100, 1e30] resp. [1, 1e30])
// against the compiler's correctly rounded expansions; tests/test_tvl1_gpu.py runs that check.  (The
// raw seeds fail it on 31 % resp. 11 % of the inputs; gfx950's v_rsq_f32 / v_rcp_f32 are accurate
// enough that the second refinement step of the textbook sequences is never needed.)
constexpr float kSqrtReg = 7.888609052210118e-31f;  // 2^-100
__device__ __forceinline__ f2 sqrt_exact_pk(f2 s)  // s in [2^-100, 1e30]
{
    const f2 y = f2{__builtin_amdgcn_rsqf(s.x), __builtin_amdgcn_rsqf(s.y)};
    const f2 g = s * y, h = y * 0.5f;
    const f2 d = pk_fma(-g, g, s);  // exact residual s - g^2
    return pk_fma(d, h, g);
}
__device__ __forceinline__ f2 rcp_exact_pk(f2 d)  // d in [1, 1e30]
{
    const f2 r = f2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f2 e = pk_fma(-d, r, splat(1.0f));  // exact residual 1 - d r
    return pk_fma(e, r, r);
}

__global__ void k_selftest_exact_math(unsigned lo, unsigned long long n, unsigned long long* bad)
{
    unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long b0 = 0, b1 = 0;
    for (; i < n; i += stride) {
        const float x = __uint_as_float(lo + (unsigned)i);
        const f2 v = f2{x, x};
        b0 += __float_as_uint(sqrt_exact_pk(v).y) != __float_as_uint(sqrtf(x));
        if (x >= 1.0f) b1 += __float_as_uint(rcp_exact_pk(v).x) != __float_as_uint(1.0f / x);
    }
    if (b0) atomicAdd(&bad[0], b0);
    if (b1) atomicAdd(&bad[1], b1);
}

// RP float2 per row: RP == 1 -> one 8-byte access, RP == 2 -> one 16-byte access.
template <int RP>
__device__ __forceinline__ void load_row(const float* __restrict__ row, int x0, int pitch, bool rowok, f2 (&v)[RP])
{
#pragma unroll
    for (int j = 0; j < RP; ++j) v[j] = f2{0.0f, 0.0f};
    if (!rowok || x0 >= pitch) return;
    if constexpr (RP == 2) {
        const float4 t = *reinterpret_cast<const float4*>(row + x0);
        v[0] = f2{t.x, t.y};
        v[1] = f2{t.z, t.w};
    } else {
        const float2 t = *reinterpret_cast<const float2*>(row + x0);
        v[0] = f2{t.x, t.y};
    }
}

template <int RP>
__device__ __forceinline__ void store_row(float* __restrict__ row, int x0, int pitch, const f2 (&v)[RP])
{
    if (x0 >= pitch) return;
    if constexpr (RP == 2) *reinterpret_cast<float4*>(row + x0) = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
    else *reinterpret_cast<float2*>(row + x0) = make_float2(v[0].x, v[0].y);
}

// S6 (+S7 when EPS).  Tile = (128*RP) x (NW*C) pixels; a thread keeps C rows of 2*RP pixels of all ten
// fields in registers as float2 pairs, so that the bulk of the arithmetic issues as v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 (two pixels per VALU instruction: a wave64 VALU instruction occupies
// its SIMD for 4 cycles on gfx950 whether it is packed or not -- measured, profiles/README.md).
//
// LX = lanes of a wave along x.  A wave is folded into GPW = 64/LX row groups of LX lanes (LX = 64:
// 256x32 tile; 32: 128x64; 21: 84x96, lane 63 idle; 16: 64x128), so that narrow pyramid levels get
// tall tiles; NG = NW*GPW row groups of C rows each.
template <int RP, int C, int NW, int LX, bool EPS, bool FAST>
__global__ void __launch_bounds__(NW * 64) k_iter_tile(IterArgs a)
{
    constexpr int R = 2 * RP, TW = LX * R, GPW = 64 / LX, NG = NW * GPW, TH = NG * C;
    static_assert(RP == 2, "the interleaved pixel layout (pslot) is written for four pixels per thread row");
    // row NG of the exchange arrays is a dummy for idle lanes (64 % LX != 0); row NG + 1 stays zero: it is
    // the "row above" of the first and the "row below" of the last row group
    __shared__ __attribute__((aligned(16))) float sP12[NG + 2][TW], sP22[NG + 2][TW], sU1[NG + 2][TW], sU2[NG + 2][TW];
    __shared__ unsigned long long sErr;

    // XCD-aware workgroup order: the tiles of a pair (which share their halos) on one XCD's L2
    unsigned lid = blockIdx.y * gridDim.x + blockIdx.x;
    {
        const unsigned nb = gridDim.x * gridDim.y, q = nb / 8, r = nb % 8, xcd = lid % 8, kk = lid / 8;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
        if (a.rev) lid = nb - 1 - lid;
    }
    const int pair = a.pair0 + (int)(lid / gridDim.x), bx = (int)(lid % gridDim.x);
    int inbuf = a.cur;
    if constexpr (EPS) {
        // S7: a pair that met the stopping rule at iteration it-1 (or earlier: its later
        // counters stay 0 < qthr) does nothing more in this warp.
        if (a.it > 0 && a.err[(size_t)pair * a.iters + a.it - 1] < a.qthr) return;
        inbuf = a.base[pair] ^ (a.it & 1);
        if (bx == 0 && threadIdx.x == 0) a.sel[pair] = inbuf ^ 1;
        if (threadIdx.x == 0) sErr = 0ull;
    }
    const int l64 = threadIdx.x & 63;
    const bool idle = l64 / LX >= GPW;                                     // lanes beyond the last whole row group
    const int lane = idle ? 0 : l64 % LX;                                  // position along x
    const int wave = idle ? NG : (int)(threadIdx.x >> 6) * GPW + l64 / LX;  // row group
    const int tx = bx % a.ntx, ty = bx / a.ntx;
    const int w = a.w, h = a.h, pitch = a.pitch, K = a.K;
    const int ox = tx * (TW - 2 * a.HX), oy = ty * (TH - 2 * K);
    const int vx0 = ox + (tx > 0 ? a.HX : 0), vx1 = (tx == a.ntx - 1) ? w : ox + TW - a.HX;
    const int vy0 = oy + (ty > 0 ? K : 0), vy1 = (ty == a.nty - 1) ? h : oy + TH - K;
    const int x0 = idle ? (1 << 28) : ox + lane * R, y0 = oy + wave * C;  // idle lanes: outside every image
    // folded waves: the DPP "left neighbour" of a row group's first lane belongs to another row group
    const float lfix = (LX == 64 || lane > 0) ? 1.0f : 0.0f;

    const float* __restrict__ ro = a.ro + (size_t)pair * kNF_RO * a.plane;
    const float* __restrict__ sin = (inbuf ? a.stB : a.stA) + (size_t)pair * kNF_STATE * a.plane;
    float* __restrict__ sout = (inbuf ? a.outA : a.outB) + (size_t)pair * kNF_STATE * a.plane;

    // columns a run may touch: up to the last 4-pixel run holding a valid pixel.  The row pitch can be larger (a level
    // laid out for k_iter_rows): those columns are never read or written here (what they hold is arbitrary).
    const int xlim = (w + 3) & ~3;
    f2 u1[C][RP], u2[C][RP], p11[C][RP], p12[C][RP], p21[C][RP], p22[C][RP];
    f2 wx[C][RP], wy[C][RP], rc[C][RP], ig[C][RP];
#pragma clang loop unroll(full)
    for (int c = 0; c < C; ++c) {
        const int y = y0 + c;
        const bool rowok = y < h;
        const size_t ro_ = (size_t)y * pitch;
        load_row<RP>(sin + ro_, x0, xlim, rowok, u1[c]);
        load_row<RP>(sin + a.plane + ro_, x0, xlim, rowok, u2[c]);
        load_row<RP>(sin + 2 * a.plane + ro_, x0, xlim, rowok, p11[c]);
        load_row<RP>(sin + 3 * a.plane + ro_, x0, xlim, rowok, p12[c]);
        load_row<RP>(sin + 4 * a.plane + ro_, x0, xlim, rowok, p21[c]);
        load_row<RP>(sin + 5 * a.plane + ro_, x0, xlim, rowok, p22[c]);
        load_row<RP>(ro + ro_, x0, xlim, rowok, wx[c]);
        load_row<RP>(ro + a.plane + ro_, x0, xlim, rowok, wy[c]);
        load_row<RP>(ro + 2 * a.plane + ro_, x0, xlim, rowok, rc[c]);
        load_row<RP>(ro + 3 * a.plane + ro_, x0, xlim, rowok, ig[c]);
    }

    // Forward-difference border rule as 0/1 multipliers (x < w-1, y < h-1): exact (d*1 = d, d*0 = 0).
    f2 mx[RP];
    float my[C];
#pragma clang loop unroll(full)
    for (int j = 0; j < RP; ++j) mx[j] = f2{x0 + j < w - 1 ? 1.0f : 0.0f, x0 + j + 2 < w - 1 ? 1.0f : 0.0f};  // pack j = pixels x0+j, x0+j+2
#pragma clang loop unroll(full)
    for (int c = 0; c < C; ++c) my[c] = y0 + c < h - 1 ? 1.0f : 0.0f;

    const float l_t = a.l_t;
    const f2 taut = splat(a.taut), theta = splat(a.theta), one = splat(1.0f);
    unsigned long long qsum = 0ull;

    f2* const rowP12 = reinterpret_cast<f2*>(&sP12[wave][lane * R]);
    f2* const rowP22 = reinterpret_cast<f2*>(&sP22[wave][lane * R]);
    f2* const rowU1 = reinterpret_cast<f2*>(&sU1[wave][lane * R]);
    f2* const rowU2 = reinterpret_cast<f2*>(&sU2[wave][lane * R]);
    const int upRow = wave > 0 ? wave - 1 : NG + 1, dnRow = wave < NG - 1 ? wave + 1 : NG + 1;
    const f2* const upP12 = reinterpret_cast<const f2*>(&sP12[upRow][lane * R]);
    const f2* const upP22 = reinterpret_cast<const f2*>(&sP22[upRow][lane * R]);
    const f2* const dnU1 = reinterpret_cast<const f2*>(&sU1[dnRow][lane * R]);
    const f2* const dnU2 = reinterpret_cast<const f2*>(&sU2[dnRow][lane * R]);

    for (int i = threadIdx.x; i < TW; i += NW * 64) sP12[NG + 1][i] = sP22[NG + 1][i] = sU1[NG + 1][i] = sU2[NG + 1][i] = 0.0f;
#pragma clang loop unroll(full)
    for (int j = 0; j < RP; ++j) {
        rowP12[j] = p12[C - 1][j];
        rowP22[j] = p22[C - 1][j];
    }
    __syncthreads();

    for (int k = 0; k < K; ++k) {
        // ---- phase A: u <- TH(u) + theta * div p   (p of the left and upper neighbours)
        // The IPOL border rules of the divergence need no selects: the left neighbour of x = 0 and
        // the upper neighbour of y = 0 arrive as 0 (DPP bound_ctrl / wave 0), and p11 at x = w-1,
        // p12 at y = h-1 are identically 0 (their forward differences are masked in phase B), so
        // p - neighbour reproduces {p, -neighbour, p - neighbour} exactly.
        f2 A12[RP], A22[RP];
#pragma clang loop unroll(full)
        for (int j = 0; j < RP; ++j) {
            A12[j] = upP12[j];
            A22[j] = upP22[j];
        }
        // Row order: row 0 needs the neighbour row fetched from LDS just above (its latency hides behind
        // row 1) and produces the row the group above waits for (stored right after it, so that the
        // store's latency hides behind rows 2.. before the barrier).
#pragma clang loop unroll(full)
        for (int cc = 0; cc < C; ++cc) {
            const int c = (C > 1 && cc < 2) ? 1 - cc : cc;
            // backward difference across the lane boundary: a plain subtraction, so that the compiler folds the lane shift
            // into it (one v_subrev_f32_dpp); in folded waves the first lane of a row group has no left neighbour in its
            // row: it keeps p itself (= p - 0, what the 0/1 multiplier lfix used to produce)
            float e11 = p11[c][0].x - dpp_from_left(p11[c][RP - 1].y), e21 = p21[c][0].x - dpp_from_left(p21[c][RP - 1].y);
            if constexpr (LX != 64) {
                e11 = lfix != 0.0f ? e11 : p11[c][0].x;
                e21 = lfix != 0.0f ? e21 : p21[c][0].x;
            }
#pragma clang loop unroll(full)
            for (int j = 0; j < RP; ++j) {
                // backward x differences in the interleaved layout (pack 0 = E = pixels x0, x0+2; pack 1 = O = x0+1,
                // x0+3): O - E is one packed subtract; E needs the left lane's last pixel and O.x: two v_sub_f32
                const f2 dx11 = j == 0 ? f2{e11, sub_s(p11[c][0].y, p11[c][1].x)} : p11[c][1] - p11[c][0];
                const f2 dx21 = j == 0 ? f2{e21, sub_s(p21[c][0].y, p21[c][1].x)} : p21[c][1] - p21[c][0];
                const f2 a12 = c > 0 ? p12[c > 0 ? c - 1 : 0][j] : A12[j];
                const f2 a22 = c > 0 ? p22[c > 0 ? c - 1 : 0][j] : A22[j];
                const f2 div1 = dx11 + (p12[c][j] - a12);
                const f2 div2 = dx21 + (p22[c][j] - a22);
                const f2 rho = pk_fma(wy[c][j], u2[c][j], pk_fma(wx[c][j], u1[c][j], rc[c][j]));
                const f2 t = -rho * ig[c][j];
                const f2 fi = f2{__builtin_amdgcn_fmed3f(t.x, -l_t, l_t), __builtin_amdgcn_fmed3f(t.y, -l_t, l_t)};
                const f2 v1 = pk_fma(fi, wx[c][j], u1[c][j]);
                const f2 v2 = pk_fma(fi, wy[c][j], u2[c][j]);
                const f2 n1 = pk_fma(theta, div1, v1);
                const f2 n2 = pk_fma(theta, div2, v2);
                if constexpr (EPS) {
                    const int y = y0 + c, x = x0 + j;  // pack j holds pixels x and x + 2
                    const f2 e1 = n1 - u1[c][j], e2 = n2 - u2[c][j];
                    const f2 e = pk_fma(e2, e2, e1 * e1);
                    if (y >= vy0 && y < vy1) {
                        if (x >= vx0 && x < vx1) qsum += (unsigned long long)(fminf(e.x, 1024.0f) * 4294967296.0f);
                        if (x + 2 >= vx0 && x + 2 < vx1) qsum += (unsigned long long)(fminf(e.y, 1024.0f) * 4294967296.0f);
                    }
                }
                u1[c][j] = n1;
                u2[c][j] = n2;
            }
            if (c == 0) {
#pragma clang loop unroll(full)
                for (int j = 0; j < RP; ++j) {
                    rowU1[j] = u1[0][j];
                    rowU2[j] = u2[0][j];
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep rows in program order: bounds register pressure
        }
        __syncthreads();

        // ---- phase B: p <- (p + taut * grad u) / (1 + taut * |grad u|)   (right and lower neighbours)
        f2 B1[RP], B2[RP];
#pragma clang loop unroll(full)
        for (int j = 0; j < RP; ++j) {
            B1[j] = dnU1[j];
            B2[j] = dnU2[j];
        }
        // Row order 0 .. C-3, C-1, C-2: the last row needs the LDS fetch and produces the exchanged row
#pragma clang loop unroll(full)
        for (int cc = 0; cc < C; ++cc) {
            const int c = (C > 1 && cc >= C - 2) ? 2 * C - 3 - cc : cc;
            const float r1 = dpp_from_right(u1[c][0].x), r2 = dpp_from_right(u2[c][0].x);
#pragma clang loop unroll(full)
            for (int j = 0; j < RP; ++j) {
                // forward x differences: E's are O - E (packed); O's need E.y and the right lane's first pixel
                const f2 d1x = j == 0 ? u1[c][1] - u1[c][0] : f2{sub_s(u1[c][0].y, u1[c][1].x), r1 - u1[c][1].y};
                const f2 d2x = j == 0 ? u2[c][1] - u2[c][0] : f2{sub_s(u2[c][0].y, u2[c][1].x), r2 - u2[c][1].y};
                const f2 b1 = c < C - 1 ? u1[c < C - 1 ? c + 1 : 0][j] : B1[j];
                const f2 b2 = c < C - 1 ? u2[c < C - 1 ? c + 1 : 0][j] : B2[j];
                const f2 u1x = d1x * mx[j], u1y = (b1 - u1[c][j]) * my[c];
                const f2 u2x = d2x * mx[j], u2y = (b2 - u2[c][j]) * my[c];
                // |grad u|^2 + 2^-100 (S6): the regulariser rides in the first FMA
                const f2 s1 = pk_fma(u1y, u1y, pk_fma(u1x, u1x, splat(kSqrtReg)));
                const f2 s2 = pk_fma(u2y, u2y, pk_fma(u2x, u2x, splat(kSqrtReg)));
                f2 q1, q2;
                if constexpr (FAST) {  // 1-ulp hardware sqrt / rcp (va_tvl1_params.fast_math)
                    const f2 g1 = f2{__builtin_amdgcn_sqrtf(s1.x), __builtin_amdgcn_sqrtf(s1.y)};
                    const f2 g2 = f2{__builtin_amdgcn_sqrtf(s2.x), __builtin_amdgcn_sqrtf(s2.y)};
                    const f2 d1 = pk_fma(taut, g1, one), d2 = pk_fma(taut, g2, one);
                    q1 = f2{__builtin_amdgcn_rcpf(d1.x), __builtin_amdgcn_rcpf(d1.y)};
                    q2 = f2{__builtin_amdgcn_rcpf(d2.x), __builtin_amdgcn_rcpf(d2.y)};
                } else {
                    // exact contract: correctly rounded sqrt, ONE correctly rounded division per pixel
                    // (s >= 2^-100: inside sqrt_exact_pk's domain).
                    const f2 g1 = sqrt_exact_pk(s1), g2 = sqrt_exact_pk(s2);
                    const f2 d1 = pk_fma(taut, g1, one), d2 = pk_fma(taut, g2, one);
                    const f2 rinv = rcp_exact_pk(d1 * d2);
                    q1 = d2 * rinv;
                    q2 = d1 * rinv;
                }
                p11[c][j] = pk_fma(taut, u1x, p11[c][j]) * q1;
                p12[c][j] = pk_fma(taut, u1y, p12[c][j]) * q1;
                p21[c][j] = pk_fma(taut, u2x, p21[c][j]) * q2;
                p22[c][j] = pk_fma(taut, u2y, p22[c][j]) * q2;
            }
            if (c == C - 1) {
#pragma clang loop unroll(full)
                for (int j = 0; j < RP; ++j) {
                    rowP12[j] = p12[C - 1][j];
                    rowP22[j] = p22[C - 1][j];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- write back the tile's valid interior (the valid regions partition the image)
    const bool runok = x0 >= vx0 && x0 < vx1;
#pragma clang loop unroll(full)
    for (int c = 0; c < C; ++c) {
        const int y = y0 + c;
        if (runok && y >= vy0 && y < vy1 && y < h) {
            const size_t ro_ = (size_t)y * pitch;
            store_row<RP>(sout + ro_, x0, xlim, u1[c]);
            store_row<RP>(sout + a.plane + ro_, x0, xlim, u2[c]);
            store_row<RP>(sout + 2 * a.plane + ro_, x0, xlim, p11[c]);
            store_row<RP>(sout + 3 * a.plane + ro_, x0, xlim, p12[c]);
            store_row<RP>(sout + 4 * a.plane + ro_, x0, xlim, p21[c]);
            store_row<RP>(sout + 5 * a.plane + ro_, x0, xlim, p22[c]);
        }
    }
    if constexpr (EPS) {
        if (qsum) atomicAdd(&sErr, qsum);
        __syncthreads();
        if (threadIdx.x == 0 && sErr) atomicAdd(&a.err[(size_t)pair * a.iters + a.it], sErr);
    }
}

// ---------------------------------------------------------------- rows of N pixels per lane ---
// A lane's N consecutive pixels of one field (N = 2, 3, 4) as N/2 packed pairs + a scalar tail: the arithmetic of the
// row pipelines (k_iter_stream, k_iter_rows) is written once over these; pairs issue as v_pk_*_f32, the tail as plain
// VALU -- element by element the same IEEE operations, so results do not depend on N.
template <int N>
struct Row {
    static constexpr int NP = N / 2, NT = N & 1;
    f2 p[NP > 0 ? NP : 1];
    float t;
};
template <int N>
__device__ __forceinline__ Row<N> row_splat(float v)
{
    Row<N> r;
#pragma clang loop unroll(full)
    for (int j = 0; j < Row<N>::NP; ++j) r.p[j] = splat(v);
    r.t = v;
    return r;
}
template <int N>
__device__ __forceinline__ float row_get(const Row<N>& r, int i)
{
    return i < 2 * Row<N>::NP ? ((i & 1) ? r.p[i >> 1].y : r.p[i >> 1].x) : r.t;
}
template <int N>
__device__ __forceinline__ void row_set(Row<N>& r, int i, float v)
{
    if (i < 2 * Row<N>::NP) {
        if (i & 1) r.p[i >> 1].y = v;
        else r.p[i >> 1].x = v;
    } else {
        r.t = v;
    }
}
#define VA_ROW_OP(NAME, ARGS, PK, SC)                                   \
    template <int N>                                                    \
    __device__ __forceinline__ Row<N> NAME ARGS                         \
    {                                                                   \
        Row<N> r;                                                       \
        _Pragma("clang loop unroll(full)") for (int j = 0; j < Row<N>::NP; ++j) r.p[j] = PK; \
        r.t = 0.0f;                                                     \
        if constexpr (Row<N>::NT) r.t = SC;                             \
        return r;                                                       \
    }
VA_ROW_OP(r_fma, (const Row<N>& a, const Row<N>& b, const Row<N>& c), pk_fma(a.p[j], b.p[j], c.p[j]), fmaf(a.t, b.t, c.t))
VA_ROW_OP(r_fma_s, (float s, const Row<N>& b, const Row<N>& c), pk_fma(splat(s), b.p[j], c.p[j]), fmaf(s, b.t, c.t))
VA_ROW_OP(r_fma_c, (const Row<N>& a, const Row<N>& b, float c), pk_fma(a.p[j], b.p[j], splat(c)), fmaf(a.t, b.t, c))
VA_ROW_OP(r_mul, (const Row<N>& a, const Row<N>& b), a.p[j] * b.p[j], a.t * b.t)
VA_ROW_OP(r_mul_s, (const Row<N>& a, float s), a.p[j] * splat(s), a.t * s)
VA_ROW_OP(r_negmul, (const Row<N>& a, const Row<N>& b), -a.p[j] * b.p[j], -a.t * b.t)
VA_ROW_OP(r_add, (const Row<N>& a, const Row<N>& b), a.p[j] + b.p[j], a.t + b.t)
VA_ROW_OP(r_sub, (const Row<N>& a, const Row<N>& b), a.p[j] - b.p[j], a.t - b.t)
VA_ROW_OP(r_med3, (const Row<N>& a, float lo, float hi),
          (f2{__builtin_amdgcn_fmed3f(a.p[j].x, lo, hi), __builtin_amdgcn_fmed3f(a.p[j].y, lo, hi)}), __builtin_amdgcn_fmed3f(a.t, lo, hi))
#undef VA_ROW_OP
// scalar twins of sqrt_exact_pk / rcp_exact_pk (the same operations on one element: bit-identical)
__device__ __forceinline__ float sqrt_exact_s(float s)
{
    const float y = __builtin_amdgcn_rsqf(s);
    const float g = s * y, h = y * 0.5f;
    const float d = fmaf(-g, g, s);
    return fmaf(d, h, g);
}
__device__ __forceinline__ float rcp_exact_s(float d)
{
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = fmaf(-d, r, 1.0f);
    return fmaf(e, r, r);
}
template <int N, bool FAST>
__device__ __forceinline__ Row<N> r_sqrt(const Row<N>& s)
{
    Row<N> r;
#pragma clang loop unroll(full)
    for (int j = 0; j < Row<N>::NP; ++j)
        r.p[j] = FAST ? f2{__builtin_amdgcn_sqrtf(s.p[j].x), __builtin_amdgcn_sqrtf(s.p[j].y)} : sqrt_exact_pk(s.p[j]);
    r.t = 0.0f;
    if constexpr (Row<N>::NT) r.t = FAST ? __builtin_amdgcn_sqrtf(s.t) : sqrt_exact_s(s.t);
    return r;
}
template <int N, bool FAST>
__device__ __forceinline__ Row<N> r_rcp(const Row<N>& d)
{
    Row<N> r;
#pragma clang loop unroll(full)
    for (int j = 0; j < Row<N>::NP; ++j)
        r.p[j] = FAST ? f2{__builtin_amdgcn_rcpf(d.p[j].x), __builtin_amdgcn_rcpf(d.p[j].y)} : rcp_exact_pk(d.p[j]);
    r.t = 0.0f;
    if constexpr (Row<N>::NT) r.t = FAST ? __builtin_amdgcn_rcpf(d.t) : rcp_exact_s(d.t);
    return r;
}
// backward / forward x differences of a lane's N consecutive pixels; the neighbour lane supplies the pixel across the
// lane boundary (0 beyond the wave's ends: the divergence's border rule on the left, masked by mx on the right)
template <int N>
__device__ __forceinline__ Row<N> r_diff_back(const Row<N>& a)
{
    Row<N> d;
    d.t = 0.0f;
    float prev = dpp_from_left(row_get(a, N - 1));
#pragma clang loop unroll(full)
    for (int i = 0; i < N; ++i) {
        const float cur = row_get(a, i);
        row_set(d, i, i == 0 ? cur - prev : sub_s(cur, prev));  // plain across the lane boundary: folds into v_subrev_f32_dpp
        prev = cur;
    }
    return d;
}
template <int N>
__device__ __forceinline__ Row<N> r_diff_fwd(const Row<N>& a)
{
    Row<N> d;
    d.t = 0.0f;
    const float right = dpp_from_right(row_get(a, 0));
#pragma clang loop unroll(full)
    for (int i = 0; i < N; ++i)
        row_set(d, i, i < N - 1 ? sub_s(row_get(a, i + 1 < N ? i + 1 : 0), row_get(a, i)) : right - row_get(a, i));
    return d;
}

template <int PPL> struct RowIo;
template <> struct RowIo<2> {
    typedef unsigned V __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ V ld(__amdgpu_buffer_rsrc_t r, int vo, int so) { return __builtin_amdgcn_raw_buffer_load_b64(r, vo, so, 0); }
    static __device__ __forceinline__ void st(V v, __amdgpu_buffer_rsrc_t r, int vo, int so) { __builtin_amdgcn_raw_buffer_store_b64(v, r, vo, so, 0); }
};
template <> struct RowIo<3> {
    typedef unsigned V __attribute__((ext_vector_type(3)));
    static __device__ __forceinline__ V ld(__amdgpu_buffer_rsrc_t r, int vo, int so) { return __builtin_amdgcn_raw_buffer_load_b96(r, vo, so, 0); }
    static __device__ __forceinline__ void st(V v, __amdgpu_buffer_rsrc_t r, int vo, int so) { __builtin_amdgcn_raw_buffer_store_b96(v, r, vo, so, 0); }
};
template <> struct RowIo<4> {
    typedef unsigned V __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ V ld(__amdgpu_buffer_rsrc_t r, int vo, int so) { return __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0); }
    static __device__ __forceinline__ void st(V v, __amdgpu_buffer_rsrc_t r, int vo, int so) { __builtin_amdgcn_raw_buffer_store_b128(v, r, vo, so, 0); }
};
template <int N>
__device__ __forceinline__ Row<N> row_load(__amdgpu_buffer_rsrc_t r, int vo, int so)
{
    const typename RowIo<N>::V v = RowIo<N>::ld(r, vo, so);
    Row<N> o;
    o.t = 0.0f;
#pragma clang loop unroll(full)
    for (int i = 0; i < N; ++i) row_set(o, i, __uint_as_float(v[i]));
    return o;
}
template <int N>
__device__ __forceinline__ void row_store(const Row<N>& o, __amdgpu_buffer_rsrc_t r, int vo, int so)
{
    typename RowIo<N>::V v;
#pragma clang loop unroll(full)
    for (int i = 0; i < N; ++i) v[i] = __float_as_uint(row_get(o, i));
    RowIo<N>::st(v, r, vo, so);
}


// ---------------------------------------------------------------- streaming inner iterations ---
//
// k_iter_stream: the same S6 arithmetic organised as a time-skewed pipeline along y ("3.5-D blocking").  One WAVE owns a
// strip of 128 columns (2 pixels per lane, plain row order) and a chunk of rows [a0, b0); it streams the rows of the
// strip top to bottom ONCE per launch and carries every row through K iterations on the way: level t (t = 0 .. K-1)
// turns the time-t values of its incoming row into time-(t+1) values one row behind, so level t works on row
// (s - t) at step s.  A level keeps only ONE row of p (its row above) and one row of the new u (waiting for its row
// below) in registers: 12 VGPRs per level, 120 for K = 10.  The per-warp constants of the K rows in flight sit in a
// private LDS ring (each lane reads back only what it wrote: no barrier anywhere in the kernel).
// Against k_iter_tile's overlapped tiles (halo K on four sides: 2.0x redundant work on the 224^2 level) a strip pays
// the halo only in x where the level is wider than 128 columns and a triangular K(K+1) level-rows per chunk in y.
struct StreamArgs {
    const float* ro;
    const float* sin;
    float* sout;
    size_t plane;
    int w, h, pitch;
    int nsx, nch, R, HX;  // strips per row, chunks per column, rows per chunk, x halo of interior strip edges
    int K;                // iterations in this launch (<= KS)
    int pair0, rev;
    float l_t, taut, theta;
    // narrow = 1: the last strip of a row needs at most 64 columns (32 lanes), so that ONE wave carries the last strips of TWO
    // consecutive pairs, pair 2c in its lanes 0..31 and pair 2c + 1 in its lanes 32..63 (the 143^2 level: 128 + 64 lane-columns
    // per row instead of 2 x 128).  What crosses the lane 31 | 32 boundary is the left halo of the one and columns beyond the
    // image of the other.  The grid then is ((2 nsx - 1) x nch, couples of pairs); npairs: pairs of this launch.
    int narrow, npairs;
};

// KH levels per wave, NWV waves per workgroup.  NWV = 1: the wave is the whole pipeline (K <= KH iterations per pass).
// NWV = 2: the second wave continues where the first stops -- it takes the rows that leave the first wave's level
// KH-1 out of a double-buffered LDS row instead of HBM and carries them through levels KH .. 2KH-1, so that one pass
// over the strip is worth K <= 2 KH iterations of HBM traffic; both waves share the ring of per-warp constants; one
// workgroup barrier per step keeps them a step apart.
// NCH chains per wave (round 3): the KH levels of a wave are cut into NCH chains of KC = KH / NCH consecutive levels.
// Inside a chain level t + 1 consumes what level t emits in the SAME step (one long dependent sequence of ~16 operations
// per level); from one chain to the next the row waits in registers for one step (a latch, 6 x PPL registers), exactly
// like the hand-over between the two waves but inside one.  The chains of a step are therefore independent of each
// other and are issued interleaved, level by level: the exact square-root / reciprocal sequences and the s_nop hazard
// slots of one chain are covered by the other's arithmetic (tools/microbench_tvl1_chain.hip: 125 instead of 147 ns per
// level-row on the bare arithmetic).  Every chain end costs one step of pipeline depth: global level g works at
// pipeline position pos(g) = g + (chain ends before g), i.e. on row s - pos(g) of the strip at step s.
// PPL pixels per lane: a strip is 64 * PPL columns wide (2: 128, the default; 3: 192, so that a 129..192-column level is
// ONE well-filled strip without any x halo -- a tested option that measured no faster, see stream_ppl()).
// SUBS = 4 (k_iter_stream4): a 512-thread workgroup runs FOUR jobs, job `sub` on the waves sub (first wave) and sub + 4
// (second wave) -- the dispatcher puts waves w and w + 4 of a workgroup on the same SIMD, so the two waves of a pipeline
// share one: while one of them waits for the other, the other has the SIMD to itself.  All eight waves meet at every
// barrier, so every job of the workgroup runs `steps_pad` steps (a job with fewer only joins the barrier); `active` =
// false: a padding job.
template <int KH, int NCH>
struct StreamShape {
    static_assert(KH % NCH == 0, "the levels of a wave are cut into chains of equal length");
    static constexpr int KC = KH / NCH;      // levels per chain
    static constexpr int SPAN = KH + NCH;    // pipeline positions a wave occupies (its levels + its chain ends)
    static constexpr int lpos(int t) { return t + t / KC; }  // position of the wave's level t behind the wave's first level
};
// rows of per-warp constants the LDS ring holds: the oldest row a level reads at step s is s - (its pipeline position)
// (three and four waves: the LAST level of the last wave takes the constants its neighbour level read one step earlier --
// the same row -- out of registers, so that the ring's oldest row is never read: one row less, which is what lets three
// 4 x 4 workgroups share a CU's 160 KB)
template <int KH, int NWV, int NCH>
constexpr bool stream_keep_last() { return NWV >= 3 && NCH == 1 && KH >= 2; }
template <int KH, int NWV, int NCH>
constexpr int stream_ring_rows() { return NWV == 1 ? KH + NCH - 2 : NWV * (KH + NCH) - 2 - (stream_keep_last<KH, NWV, NCH>() ? 1 : 0); }

template <int PPL>
struct StreamRow {  // the six state fields of one row of a strip
    Row<PPL> u1, u2, p11, p12, p21, p22;
};

template <int PPL, int KH, int NWV, bool FAST, int SUBS = 1, int NCH = 1>
__device__ __forceinline__ void stream_job(const StreamArgs& a, const int pair, const int job, const int K,
                                           const float* __restrict__ sin_all, float* __restrict__ sout_all, const int sub = 0,
                                           const int role = -1, const bool active = true, const int steps_pad = 0, const int half = 0)
{   // half: 0 = a whole-wave strip of `pair`; 1 = the shared last strip of `pair` (lanes 0..31) and `pair + 1` (lanes 32..63);
    // 2 = the same without a second pair (lanes 32..63 idle)
    typedef Row<PPL> R;
    typedef StreamRow<PPL> SR;
    typedef StreamShape<KH, NCH> SH;
    constexpr int NP = R::NP, NT = R::NT, SW = 64 * PPL, KC = SH::KC;
    // ring rows: NWV = 1: level t >= 1 reads row s - lpos(t), row s is written at the end of step s.  NWV = 2: the first
    // wave writes row s - 1 at the start of step s (after the barrier), the oldest row read in step s is
    // s - (NWV * SPAN - 2), whose slot is that of row s (written in step s + 1).
    constexpr int NRING = stream_ring_rows<KH, NWV, NCH>() > 0 ? stream_ring_rows<KH, NWV, NCH>() : 1;
    __shared__ f2 ringP_[SUBS][NRING][kNF_RO][NP][64];
    __shared__ float ringT_[SUBS][NRING][kNF_RO][NT ? 64 : 1];
    __shared__ f2 ifaceP_[SUBS][NWV > 1 ? NWV - 1 : 1][2][kNF_STATE][NP][64];
    __shared__ float ifaceT_[SUBS][NWV > 1 ? NWV - 1 : 1][2][kNF_STATE][NT ? 64 : 1];
    auto& ringP = ringP_[sub];
    auto& ringT = ringT_[sub];
    auto& ifaceP = ifaceP_[sub];
    auto& ifaceT = ifaceT_[sub];

    const int sx = job % a.nsx, ch = job / a.nsx;
    const int w = a.w, h = a.h, pitch = a.pitch;
    const int lane = threadIdx.x & 63;
    const int wv = NWV == 1 ? 0 : role >= 0 ? role : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ox = sx * (SW - 2 * a.HX);
    const int vx0 = ox + (sx > 0 ? a.HX : 0), vx1 = (sx == a.nsx - 1) ? pitch : ox + SW - a.HX;
    const int x0 = ox + PPL * (half ? (lane & 31) : lane);  // (the pitch, the halo and so every strip origin are multiples of PPL)
    const bool colok = x0 < pitch && !(half == 2 && lane >= 32);
    const bool stok = colok && x0 >= vx0 && x0 < vx1;
    const int a0 = ch * a.R, b0 = d_min(h, a0 + a.R);
    const int ys = d_max(0, a0 - K), ye = d_min(h, b0 + K);

    // buffer addressing: resource (scalar) + per-lane byte offset (one VGPR) + scalar plane/row offset.  Lanes beyond the
    // pitch read column 0 (finite values that no valid pixel can see: their differences are multiplied by mx = 0) and
    // never store.
    const int planeb = (int)(a.plane * sizeof(float)), pitchb = pitch * (int)sizeof(float);
    const int npr = half == 1 ? 2 : 1;  // pairs the resources span (the planes of consecutive pairs follow each other)
    const __amdgpu_buffer_rsrc_t rs_ro = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.ro + (size_t)pair * kNF_RO * a.plane), 0, npr * kNF_RO * planeb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(sin_all + (size_t)pair * kNF_STATE * a.plane), 0, npr * kNF_STATE * planeb, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(sout_all + (size_t)pair * kNF_STATE * a.plane, 0,
                                                                           npr * kNF_STATE * planeb, 0x00020000);
    const bool second = half == 1 && lane >= 32;  // this lane works on pair + 1
    const int loff = colok ? x0 * (int)sizeof(float) + (second ? kNF_STATE * planeb : 0) : 0;     // state planes
    const int loff_ro = colok ? x0 * (int)sizeof(float) + (second ? kNF_RO * planeb : 0) : 0;     // per-warp constants

    R mx;
    mx.t = 0.0f;
#pragma clang loop unroll(full)
    for (int i = 0; i < PPL; ++i) row_set(mx, i, x0 + i < w - 1 ? 1.0f : 0.0f);
    const float l_t = a.l_t, taut = a.taut, theta = a.theta;
    const R zero = row_splat<PPL>(0.0f), one = row_splat<PPL>(1.0f);

    auto ring_put = [&](int slot, int f, const R& v) __attribute__((always_inline)) {
#pragma clang loop unroll(full)
        for (int j = 0; j < NP; ++j) ringP[slot][f][j][lane] = v.p[j];
        if constexpr (NT) ringT[slot][f][lane] = v.t;
    };
    auto ring_get = [&](int slot, int f) __attribute__((always_inline)) -> R {
        R v;
#pragma clang loop unroll(full)
        for (int j = 0; j < NP; ++j) v.p[j] = ringP[slot][f][j][lane];
        v.t = 0.0f;
        if constexpr (NT) v.t = ringT[slot][f][lane];
        return v;
    };
    auto if_put = [&](int wb, int b, int f, const R& v) __attribute__((always_inline)) {
#pragma clang loop unroll(full)
        for (int j = 0; j < NP; ++j) ifaceP[wb][b][f][j][lane] = v.p[j];
        if constexpr (NT) ifaceT[wb][b][f][lane] = v.t;
    };
    auto if_get = [&](int wb, int b, int f) __attribute__((always_inline)) -> R {
        R v;
#pragma clang loop unroll(full)
        for (int j = 0; j < NP; ++j) v.p[j] = ifaceP[wb][b][f][j][lane];
        v.t = 0.0f;
        if constexpr (NT) v.t = ifaceT[wb][b][f][lane];
        return v;
    };

    for (int r = wv; r < NRING; r += NWV)
#pragma clang loop unroll(full)
        for (int f = 0; f < kNF_RO; ++f) ring_put(r, f, zero);
    if constexpr (NWV > 1) {
        if (wv < NWV - 1)  // wave w zeroes the hand-over rows it will write
#pragma clang loop unroll(full)
            for (int f = 0; f < kNF_STATE; ++f) {
                if_put(wv, 0, f, zero);
                if_put(wv, 1, f, zero);
            }
        __syncthreads();
    }

    auto ld = [&](const __amdgpu_buffer_rsrc_t& r, int f, int y) -> R { return row_load<PPL>(r, loff, f * planeb + y * pitchb); };
    auto ld_ro = [&](int f, int y) -> R { return row_load<PPL>(rs_ro, loff_ro, f * planeb + y * pitchb); };
    auto st = [&](const R& v, int f, int y) { row_store<PPL>(v, rs_out, loff, f * planeb + y * pitchb); };

    // the last valid row b0 - 1 leaves the last ACTIVE level (K - 1) when row b0 (or the dummy row h) comes in, and then
    // waits one step at every chain end on its way to the last wave's output: NWV * NCH - 1 of them
    const int nsteps = active ? b0 - ys + K + (NWV * NCH - 1) : 0;

    auto run = [&](auto wave_tag) __attribute__((always_inline)) {
        constexpr int W = decltype(wave_tag)::value;
        constexpr bool FIRST = W == 0, LAST = W == NWV - 1;
        constexpr int G0 = W * KH, LAG = W * SH::SPAN;  // first level of this wave; steps it runs behind the first wave

        R P11[KH], P12[KH], P21[KH], P22[KH], U1[KH], U2[KH];
#pragma clang loop unroll(full)
        for (int t = 0; t < KH; ++t) P11[t] = P12[t] = P21[t] = P22[t] = U1[t] = U2[t] = zero;
        SR latch[NCH > 1 ? NCH - 1 : 1];  // what chain c emitted in the previous step: chain c + 1's input of this one
#pragma clang loop unroll(full)
        for (int c = 0; c < (NCH > 1 ? NCH - 1 : 1); ++c) latch[c] = SR{zero, zero, zero, zero, zero, zero};
        R nst[kNF_STATE], nro[kNF_RO], rprev[kNF_RO] = {zero, zero, zero, zero};
        constexpr bool KEEP = LAST && stream_keep_last<KH, NWV, NCH>();
        R keep[kNF_RO] = {zero, zero, zero, zero};  // KEEP: the constants level KH - 2 worked with in the previous step = level KH - 1's of this one
#pragma clang loop unroll(full)
        for (int f = 0; f < kNF_STATE; ++f) nst[f] = FIRST ? ld(rs_in, f, ys) : zero;
#pragma clang loop unroll(full)
        for (int f = 0; f < kNF_RO; ++f) nro[f] = FIRST ? ld_ro(f, ys) : zero;

        // One level: phase A on the incoming row C (k_iter_tile's arithmetic, operation for operation), phase B on the
        // row above it (whose lower neighbour is the row just computed); the level keeps the incoming p and the new u,
        // and passes on (its old u, the new p) = the time-(t+1) values of the row above.
        // MY1: the row the level holds is not the image's last one (my = 1): its y differences need no multiplier
        // (x * 1 = x exactly; the straight-line steady-state steps use this form)
        auto level = [&](int t, SR& c, const R& wx, const R& wy, const R& rc, const R& ig, const float my, auto my1_tag) __attribute__((always_inline)) {
            constexpr bool MY1 = decltype(my1_tag)::value;
            // (the difference across the lane boundary is a plain subtraction inside r_diff_*: the compiler folds the
            // lane shift into it -- one v_subrev_f32_dpp instead of v_mov_b32_dpp + v_sub_f32)
            const R dx11 = r_diff_back(c.p11), dx21 = r_diff_back(c.p21);
            const R div1 = r_add(dx11, r_sub(c.p12, P12[t]));
            const R div2 = r_add(dx21, r_sub(c.p22, P22[t]));
            const R rho = r_fma(wy, c.u2, r_fma(wx, c.u1, rc));
            const R tt = r_negmul(rho, ig);
            const R fi = r_med3(tt, -l_t, l_t);
            const R v1 = r_fma(fi, wx, c.u1);
            const R v2 = r_fma(fi, wy, c.u2);
            const R n1 = r_fma_s(theta, div1, v1);
            const R n2 = r_fma_s(theta, div2, v2);
            const R d1x = r_diff_fwd(U1[t]), d2x = r_diff_fwd(U2[t]);
            const R u1x = r_mul(d1x, mx), u1y = MY1 ? r_sub(n1, U1[t]) : r_mul_s(r_sub(n1, U1[t]), my);
            const R u2x = r_mul(d2x, mx), u2y = MY1 ? r_sub(n2, U2[t]) : r_mul_s(r_sub(n2, U2[t]), my);
            const R s1 = r_fma(u1y, u1y, r_fma_c(u1x, u1x, kSqrtReg));
            const R s2 = r_fma(u2y, u2y, r_fma_c(u2x, u2x, kSqrtReg));
            const R d1 = r_fma_s(taut, r_sqrt<PPL, FAST>(s1), one);
            const R d2 = r_fma_s(taut, r_sqrt<PPL, FAST>(s2), one);
            R q1, q2;
            if constexpr (FAST) {
                q1 = r_rcp<PPL, true>(d1);
                q2 = r_rcp<PPL, true>(d2);
            } else {
                const R rinv = r_rcp<PPL, false>(r_mul(d1, d2));
                q1 = r_mul(d2, rinv);
                q2 = r_mul(d1, rinv);
            }
            const R o11 = r_mul(r_fma_s(taut, u1x, P11[t]), q1);
            const R o12 = r_mul(r_fma_s(taut, u1y, P12[t]), q1);
            const R o21 = r_mul(r_fma_s(taut, u2x, P21[t]), q2);
            const R o22 = r_mul(r_fma_s(taut, u2y, P22[t]), q2);
            const R ou1 = U1[t], ou2 = U2[t];
            P11[t] = c.p11;
            P12[t] = c.p12;
            P21[t] = c.p21;
            P22[t] = c.p22;
            U1[t] = n1;
            U2[t] = n2;
            c.u1 = ou1;
            c.u2 = ou2;
            c.p11 = o11;
            c.p12 = o12;
            c.p21 = o21;
            c.p22 = o22;
        };

        auto step = [&](const int s, auto steady_tag) __attribute__((always_inline)) {
            constexpr bool STEADY = decltype(steady_tag)::value;
            const int s0 = s % NRING;  // row s of the strip lives in ring slot s % NRING
            SR cc[NCH];                // the row travelling through chain c in this step
            R r0[kNF_RO] = {zero, zero, zero, zero};
            if constexpr (FIRST) {
                cc[0] = SR{nst[0], nst[1], nst[2], nst[3], nst[4], nst[5]};
#pragma clang loop unroll(full)
                for (int f = 0; f < kNF_RO; ++f) r0[f] = nro[f];
                // next row (beyond the last row of the strip: row ye - 1 again -- finite values nobody uses, or, at the
                // image bottom, the dummy row h whose only consumer multiplies its difference by my = 0)
                const int rn = d_min(ys + s + 1, ye - 1);
                if (!(STEADY && (VA_TIMING_SKIP & 2))) {
#pragma clang loop unroll(full)
                    for (int f = 0; f < kNF_STATE; ++f) nst[f] = ld(rs_in, f, rn);
#pragma clang loop unroll(full)
                    for (int f = 0; f < kNF_RO; ++f) nro[f] = ld_ro(f, rn);
                }
                if constexpr (NWV > 1) {  // the constants of row s - 1 go into the ring now that the barrier has passed
                    const int sl = s0 == 0 ? NRING - 1 : s0 - 1;
                    if (!(STEADY && (VA_TIMING_SKIP & 32)))
#pragma clang loop unroll(full)
                        for (int f = 0; f < kNF_RO; ++f) ring_put(sl, f, rprev[f]);
                }
                if (!(VA_TIMING_SKIP & 16)) __builtin_amdgcn_sched_barrier(0);  // the loads stay at the top of the step: a whole step hides their latency
            } else {
                const int bsel = (s + 1) & 1;  // what the wave before wrote in step s - 1
                if (STEADY && (VA_TIMING_SKIP & 4)) {
                    cc[0] = SR{U1[0], U2[0], P11[0], P12[0], P21[0], P22[0]};
                } else {
                    cc[0].u1 = if_get(W - 1, bsel, 0);
                    cc[0].u2 = if_get(W - 1, bsel, 1);
                    cc[0].p11 = if_get(W - 1, bsel, 2);
                    cc[0].p12 = if_get(W - 1, bsel, 3);
                    cc[0].p21 = if_get(W - 1, bsel, 4);
                    cc[0].p22 = if_get(W - 1, bsel, 5);
                }
            }
#pragma clang loop unroll(full)
            for (int c = 1; c < NCH; ++c) cc[c] = latch[c - 1];
            // ring slot of the constants of level t's incoming row, s - LAG - lpos(t)
            auto slot_of = [&](int t) {
                const int c = (LAG + SH::lpos(t)) % NRING;
                return s0 - c < 0 ? s0 - c + NRING : s0 - c;
            };
            if constexpr (STEADY) {
                // every level of the wave works and no row is the image's last: straight-line code.  The levels are
                // issued round by round -- round i = level i of every chain, independent of each other -- and the
                // constants of round i + 1 are fetched from the ring while round i computes
                R q[NCH][kNF_RO];
#pragma clang loop unroll(full)
                for (int c = 0; c < NCH; ++c) {
                    if (FIRST && c == 0) {
#pragma clang loop unroll(full)
                        for (int f = 0; f < kNF_RO; ++f) q[c][f] = r0[f];
                    } else if (VA_TIMING_SKIP & 8) {
#pragma clang loop unroll(full)
                        for (int f = 0; f < kNF_RO; ++f) q[c][f] = rprev[f];
                    } else {
                        const int slot = slot_of(c * KC);
#pragma clang loop unroll(full)
                        for (int f = 0; f < kNF_RO; ++f) q[c][f] = ring_get(slot, f);
                    }
                }
#pragma clang loop unroll(full)
                for (int i = 0; i < KC; ++i) {
                    R nq[NCH][kNF_RO];
#pragma clang loop unroll(full)
                    for (int c = 0; c < NCH; ++c)
#pragma clang loop unroll(full)
                        for (int f = 0; f < kNF_RO; ++f) nq[c][f] = zero;
                    if (i + 1 < KC) {
#pragma clang loop unroll(full)
                        for (int c = 0; c < NCH; ++c) {
                            if (KEEP && i + 1 == KH - 1) {  // (NCH == 1) the last level: the row level KH - 2 had one step ago
#pragma clang loop unroll(full)
                                for (int f = 0; f < kNF_RO; ++f) nq[c][f] = keep[f];
                            } else {
                                const int slot = slot_of(c * KC + i + 1);
#pragma clang loop unroll(full)
                                for (int f = 0; f < kNF_RO; ++f) nq[c][f] = (VA_TIMING_SKIP & 8) ? q[c][f] : ring_get(slot, f);
                            }
                        }
                        if (!(VA_TIMING_SKIP & 16)) __builtin_amdgcn_sched_barrier(0);
                    }
                    if (KEEP && i == KH - 2) {
#pragma clang loop unroll(full)
                        for (int f = 0; f < kNF_RO; ++f) keep[f] = q[0][f];
                    }
#pragma clang loop unroll(full)
                    for (int c = 0; c < NCH; ++c) level(c * KC + i, cc[c], q[c][0], q[c][1], q[c][2], q[c][3], 1.0f, std::true_type{});
#pragma clang loop unroll(full)
                    for (int c = 0; c < NCH; ++c)
#pragma clang loop unroll(full)
                        for (int f = 0; f < kNF_RO; ++f) q[c][f] = nq[c][f];
                }
            } else {
#pragma clang loop unroll(full)
                for (int t = 0; t < KH; ++t) {
                    // global level g takes the rows [a0 - (K - g), b0 + (K - g)) of the image and, at the image bottom,
                    // the dummy row h
                    const int g = G0 + t, rin = ys + s - LAG - SH::lpos(t);
                    const int lo = d_max(0, a0 - (K - g)), hi = d_min(h, b0 + (K - g) - 1);
                    const bool act = g < K && rin >= lo && rin <= hi;
                    if (act) {
                        const float my = rin - 1 < h - 1 ? 1.0f : 0.0f;
                        if (FIRST && t == 0) {
                            level(0, cc[0], r0[0], r0[1], r0[2], r0[3], my, std::false_type{});
                        } else if (KEEP && t == KH - 1) {
                            level(t, cc[t / KC], keep[0], keep[1], keep[2], keep[3], my, std::false_type{});
                        } else {
                            const int slot = slot_of(t);
                            level(t, cc[t / KC], ring_get(slot, 0), ring_get(slot, 1), ring_get(slot, 2), ring_get(slot, 3), my, std::false_type{});
                        }
                    }
                }
                if constexpr (KEEP) {  // whether or not level KH - 2 worked in this step: the row it stands at is level KH - 1's next one
                    const int slot = slot_of(KH - 2);
#pragma clang loop unroll(full)
                    for (int f = 0; f < kNF_RO; ++f) keep[f] = ring_get(slot, f);
                }
            }
#pragma clang loop unroll(full)
            for (int c = 0; c + 1 < NCH; ++c) latch[c] = cc[c];
            const SR& co = cc[NCH - 1];
            if constexpr (LAST) {
                // the row that leaves the pipeline in this step: K iterations on, one step later per chain end on its way
                const int rout = ys + s - K - (NWV * NCH - 1);
                if (stok && rout >= a0 && rout < b0 && !(STEADY && (VA_TIMING_SKIP & 1))) {
                    st(co.u1, 0, rout);
                    st(co.u2, 1, rout);
                    st(co.p11, 2, rout);
                    st(co.p12, 3, rout);
                    st(co.p21, 4, rout);
                    st(co.p22, 5, rout);
                }
            } else {
                const int bsel = s & 1;
                if (!(STEADY && (VA_TIMING_SKIP & 4))) {
                    if_put(W, bsel, 0, co.u1);
                    if_put(W, bsel, 1, co.u2);
                    if_put(W, bsel, 2, co.p11);
                    if_put(W, bsel, 3, co.p12);
                    if_put(W, bsel, 4, co.p21);
                    if_put(W, bsel, 5, co.p22);
                }
            }
            if constexpr (FIRST) {
                if constexpr (NWV == 1) {
#pragma clang loop unroll(full)
                    for (int f = 0; f < kNF_RO; ++f) ring_put(s0, f, r0[f]);
                } else {
#pragma clang loop unroll(full)
                    for (int f = 0; f < kNF_RO; ++f) rprev[f] = r0[f];
                }
            }
            // hand-over barrier: LDS traffic only (no wait for the global loads in flight or the stores just issued)
            if constexpr (NWV > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        };
        // steady state of this wave: the steps in which all its KH levels are inside their row windows and below the
        // image's last row -- level t (global g) takes row ys + s - LAG - lpos(t), its window is [lo_g, min(hi_g, h - 1)]
        int s_a = 0, s_b = nsteps;
#pragma clang loop unroll(full)
        for (int t = 0; t < KH; ++t) {
            const int g = G0 + t, p = LAG + SH::lpos(t) - ys;
            s_a = d_max(s_a, d_max(0, a0 - (K - g)) + p);
            s_b = d_min(s_b, d_min(h - 1, d_min(h, b0 + (K - g) - 1)) + p + 1);
        }
        if (K < G0 + KH || s_a > s_b || !active) s_a = s_b = 0;
        int s = 0;
        for (; s < s_a; ++s) step(s, std::false_type{});
#if VA_STREAM_UNROLL2
        // two steady steps per loop trip: the rows a level hands on and keeps (c_* -> P[t], n -> U[t]) change registers
        // by renaming between the two copies instead of by v_mov (5 per level and step otherwise).  Not in the one-wave
        // form at two waves per SIMD: with ten levels in one wave's 256 registers the second copy spills (19-21 registers;
        // 1280x720: 62 instead of 114 pairs/s)
        if constexpr (NWV > 1 || KH > 10)
            for (; s + 1 < s_b; s += 2) {
                step(s, std::true_type{});
                step(s + 1, std::true_type{});
            }
#endif
        for (; s < s_b; ++s) step(s, std::true_type{});
        for (; s < nsteps; ++s) step(s, std::false_type{});
        if constexpr (SUBS > 1)  // the other jobs of the workgroup may have more steps: join their barriers
            for (; s < steps_pad; ++s) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    if constexpr (NWV == 1) {
        run(std::integral_constant<int, 0>{});
    } else if constexpr (NWV == 2) {
        if (wv == 0) run(std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, NWV - 1>{});
    } else if constexpr (NWV == 3) {
        if (wv == 0) run(std::integral_constant<int, 0>{});
        else if (wv == 1) run(std::integral_constant<int, 1>{});
        else run(std::integral_constant<int, 2>{});
    } else {
        static_assert(NWV == 4, "pipelines of one to four waves");
        if (wv == 0) run(std::integral_constant<int, 0>{});
        else if (wv == 1) run(std::integral_constant<int, 1>{});
        else if (wv == 2) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 3>{});
    }
}

// One launch = one pass (a.K iterations) of every pair of the call: the grid is (strips x chunks, pairs).
template <int PPL, int KH, int NWV, bool FAST, int NCH = 1>
__global__ void __launch_bounds__(NWV * 64)
    __attribute__((amdgpu_waves_per_eu((NWV == 1 && KH > 10) ? 1 : 2, (NWV == 1 && KH > 10) ? 1 : NWV >= 3 ? 4 : 2)))
    k_iter_stream(StreamArgs a)
{
    unsigned lid = blockIdx.y * gridDim.x + blockIdx.x;
    {
        const unsigned nb = gridDim.x * gridDim.y, q = nb / 8, r = nb % 8, xcd = lid % 8, kk = lid / 8;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kk;
        if (a.rev) lid = nb - 1 - lid;
    }
    if (!a.narrow) {
        stream_job<PPL, KH, NWV, FAST, 1, NCH>(a, a.pair0 + (int)(lid / gridDim.x), (int)(lid % gridDim.x), a.K, a.sin, a.sout);
        return;
    }
    // shared last strips: per couple of pairs and chunk 2 (nsx - 1) whole-wave strips and ONE wave for both last strips
    const int couple = (int)(lid / gridDim.x), jx = (int)(lid % gridDim.x), per = 2 * a.nsx - 1;
    const int ch = jx / per, r = jx % per, first = 2 * couple;
    int which = 0, sx = a.nsx - 1, half = first + 1 < a.npairs ? 1 : 2;
    if (r < 2 * (a.nsx - 1)) {
        which = r / (a.nsx - 1);
        sx = r % (a.nsx - 1);
        half = 0;
        if (first + which >= a.npairs) return;  // an odd number of pairs: the last couple has no second one
    }
    stream_job<PPL, KH, NWV, FAST, 1, NCH>(a, a.pair0 + first + which, ch * a.nsx + sx, a.K, a.sin, a.sout, 0, -1, true, 0, half);
}

#ifdef VA_EXPERIMENTS  // measured-slower kernel families: k_iter_stream4, k_iter_stream_q, k_iter_rows
#include "tvl1_experiments.inc"
#endif  // VA_EXPERIMENTS

// ---------------------------------------------------------------- host side -------------------

struct TileCfg {
    int R, C, NW, LX;
};
// Candidate tilings (R, C, NW, LX): 256x32, 128x64, 84x96 and 64x128 tiles, 512 threads, a 4x4 pixel
// patch per thread (the packed-math kernel needs more than the 128 VGPRs a 1024-thread workgroup
// would leave it).  Every candidate costs the same time per workgroup-iteration.
constexpr TileCfg kCfgs[] = {{4, 4, 8, 64}, {4, 4, 8, 32}, {4, 4, 8, 21}, {4, 4, 8, 16},
                             {4, 4, 4, 64}, {4, 4, 4, 32}, {4, 4, 4, 21}, {4, 4, 4, 16}};
constexpr int kNumCfgs = (int)(sizeof(kCfgs) / sizeof(kCfgs[0]));

struct TilePick {
    int cfg, ntx, nty, HX, K;
};

// Cost of a workgroup in CU-microseconds, measured on MI355X (tools/bench_tvl1_levels.py,
// tools/microbench_tvl1_tile.py): exposed HBM round trip per launch + time per inner iteration.
// The 4-wave candidates (half-size tiles) run two workgroups per CU: one's load/store phase hides
// behind the other's arithmetic (a few us exposed per 4096-pixel tile instead of ~15 us per 8192) and
// the 4-wave barriers cost less (1.2 us per iteration instead of ~2.7 / 2).  The constants are fitted to the
// per-level optimum found by brute force with the cache-aware workgroup order in place (tools/sweep_tvl1_tiles.py).
double tile_launch_us(int cfg) { return kCfgs[cfg].NW == 8 ? 17.0 : 4.5; }
double tile_iter_us(int cfg)
{
    const double t = kCfgs[cfg].NW == 8 ? 2.75 : 1.2;
    return kCfgs[cfg].LX >= 32 ? t : 1.03 * t;  // 3- and 4-fold waves: a little slower
}
// A launch cannot beat the HBM stream of its tiles: 4.5 TB/s = 17.6 kB per us per CU (5.2 TB/s is the best this access pattern reaches).
double tile_hbm_us(int TW, int TH, int HX, int K) { return ((double)TW * TH * 40.0 + (double)(TW - 2 * HX) * (TH - 2 * K) * 24.0) / 17600.0; }
double tile_cost_us(int cfg, int HX, int K)
{
    const int TW = kCfgs[cfg].LX * kCfgs[cfg].R, TH = kCfgs[cfg].NW * (64 / kCfgs[cfg].LX) * kCfgs[cfg].C;
    const double c = tile_launch_us(cfg) + tile_iter_us(cfg) * K, m = tile_hbm_us(TW, TH, HX, K < TH / 2 ? K : 0);
    return c > m ? c : m;
}

int tiles_1d(int n, int T, int halo)
{
    if (n <= T) return 1;
    return va_cdiv(n - 2 * halo, T - 2 * halo);
}

// Pick the candidate with the lowest modelled cost per iteration for this level and K.
TilePick pick_tiles(int w, int h, int K, unsigned mask)
{
    if (!mask) mask = (1u << kNumCfgs) - 1;
    TilePick best{};
    double best_cost = 1e300;
    for (int i = 0; i < kNumCfgs; ++i) {
        if (!((mask >> i) & 1)) continue;
        const int R = kCfgs[i].R, TW = kCfgs[i].LX * R, TH = kCfgs[i].NW * (64 / kCfgs[i].LX) * kCfgs[i].C;
        int k = K;
        if (h > TH && 2 * k >= TH) k = (TH - 1) / 2 > 0 ? (TH - 1) / 2 : 1;
        const int align = 4;  // tile origins stay 16-byte aligned
        int HX = 0;
        if (w > TW) {
            HX = va_cdiv(k, align) * align;
            if (2 * HX >= TW) continue;
        }
        const int ntx = tiles_1d(w, TW, HX), nty = tiles_1d(h, TH, k);
        const double cost = (double)ntx * nty * tile_cost_us(i, HX, k) / k;
        if (cost < best_cost) {
            best_cost = cost;
            best = TilePick{i, ntx, nty, HX, k};
        }
    }
    if (best_cost == 1e300) return pick_tiles(w, h, K, 0u);  // no allowed candidate fits this halo: allow all
    return best;
}

// block_iters = 0: the depth K that minimises the modelled time of one warp's `iters` iterations
// (full launches of K plus one shorter launch for the remainder).
TilePick pick_tiles_auto(int w, int h, int iters, unsigned mask)
{
    TilePick best{};
    double best_cost = 1e300;
    for (int K = 1; K <= 31 && K <= iters; ++K) {
        const TilePick tp = pick_tiles(w, h, K, mask);
        if (tp.K != K) continue;
        const int full = iters / K, rem = iters - full * K;
        double cost = (double)full * tp.ntx * tp.nty * tile_cost_us(tp.cfg, tp.HX, K);
        if (rem) {
            const TilePick tr = pick_tiles(w, h, rem, mask);
            cost += (double)tr.ntx * tr.nty * tile_cost_us(tr.cfg, tr.HX, tr.K) * ((double)rem / tr.K);
        }
        if (cost < best_cost) {
            best_cost = cost;
            best = tp;
        }
    }
    return best;
}

template <bool EPS, bool FAST>
void launch_iter(const TilePick& tp, const IterArgs& a, int npairs, hipStream_t st)
{
    const dim3 grid(tp.ntx * tp.nty, npairs);
    switch (tp.cfg) {
        case 0: k_iter_tile<2, 4, 8, 64, EPS, FAST><<<grid, 512, 0, st>>>(a); break;  // 256 x 32 tile
        case 1: k_iter_tile<2, 4, 8, 32, EPS, FAST><<<grid, 512, 0, st>>>(a); break;  // 128 x 64 (2 row groups per wave)
        case 2: k_iter_tile<2, 4, 8, 21, EPS, FAST><<<grid, 512, 0, st>>>(a); break;  // 84 x 96 (3 row groups, lane 63 idle)
        case 3: k_iter_tile<2, 4, 8, 16, EPS, FAST><<<grid, 512, 0, st>>>(a); break;  // 64 x 128 (4 row groups)
        case 4: k_iter_tile<2, 4, 4, 64, EPS, FAST><<<grid, 256, 0, st>>>(a); break;  // 256 x 16, two workgroups per CU
        case 5: k_iter_tile<2, 4, 4, 32, EPS, FAST><<<grid, 256, 0, st>>>(a); break;  // 128 x 32
        case 6: k_iter_tile<2, 4, 4, 21, EPS, FAST><<<grid, 256, 0, st>>>(a); break;  // 84 x 48
        default: k_iter_tile<2, 4, 4, 16, EPS, FAST><<<grid, 256, 0, st>>>(a); break; // 64 x 64
    }
}

// ---- k_iter_stream: strips x chunks of a level
constexpr int kStreamK1 = 10;       // one-wave pipeline: iterations per pass (2 pixels per lane)
constexpr int kStreamKH2 = 8;       // two-wave pipeline: levels per wave (16 iterations per pass; 2 pixels per lane)
static_assert(kStreamKH2 == 8, "k_iter_stream4 instantiates stream_job<2, 8, 2>");
// A level keeps 6 x PPL registers per lane: with 3 pixels per lane the pipelines are shallower (5 levels per wave), so
// that two waves per SIMD still fit the register file (8 and 6 levels spill, seen at compile time and in the timings)
constexpr int stream_k1(int ppl) { return ppl == 2 ? kStreamK1 : 5; }
constexpr int stream_kh2(int ppl) { return ppl == 2 ? kStreamKH2 : 5; }
constexpr int kStreamBit = 1 << 8;  // va_tvl1_params.tile_mask bit: iterate with k_iter_stream
constexpr int kNoNarrowBit = 1 << 10;  // va_tvl1_params.tile_mask bit: no shared last strips (StreamArgs.narrow) -- A/B and tests
struct StreamPick {
    int nsx, nch, R, HX, two, ppl, deep1;
    int chains;  // chains of levels per wave (stream_job, NCH): 1, or 2 with stream_waves = 5 (one deep wave) / 6 (two waves)
    int mw_nwv, mw_kh;  // > 0: the pipeline of mw_nwv waves x mw_kh levels (3 or 4 waves; stream_waves >= 7)
    int narrow;         // the last strip of a row fits 32 lanes: two pairs' last strips share a wave (StreamArgs.narrow)
};
// stream_waves >= 7: pipelines of three or four waves (waves x levels per wave)
constexpr int kMwShapes[][2] = {{4, 4}, {4, 5}, {4, 3}, {3, 5}, {3, 6}, {4, 6}};  // 7, 8, 9: compiled always; 10 ... 12: VA_EXPERIMENTS
constexpr int kNumMwShapes = (int)(sizeof(kMwShapes) / sizeof(kMwShapes[0]));
constexpr int kMwFirst = 7;
// Pixels per lane of k_iter_stream: 2 (128-column strips) unless va_tvl1_params.stream_ppl asks for 3 (192-column
// strips: a 129..192-column level then is ONE strip without x halo -- 179^2 fills 93 % of the lanes instead of 70 % of
// two 128-column strips).  Measured on the benchmark's 179^2 / 143^2 levels (320 pairs, two streams): 33.6 / 25.6 ms
// with 3 per lane (best of 5 or 4 levels per wave and 2..4 chunks of rows) against 32.9 / 25.5 ms with 2: the better
// fill is paid back by the shallower pipeline (6 x 3 registers per level: 10 instead of 16 iterations per pass) and by
// half as many strip x chunk jobs.  So 2 stays the default; 3 is kept as a tested option.
int stream_ppl(const va_tvl1_params* p, int w)
{
    (void)w;
    return p->tuning[VA_TUNE_STREAM_PPL] == 3 ? 3 : 2;
}
template <bool TWO, bool FAST>
void launch_stream(int ppl, dim3 grid, hipStream_t st, const StreamArgs& sa)
{
    constexpr int NWV = TWO ? 2 : 1;
#ifdef VA_EXPERIMENTS
    if (ppl == 3) {
        k_iter_stream<3, TWO ? stream_kh2(3) : stream_k1(3), NWV, FAST><<<grid, NWV * 64, 0, st>>>(sa);
        return;
    }
#endif
    (void)ppl;
    k_iter_stream<2, TWO ? stream_kh2(2) : stream_k1(2), NWV, FAST><<<grid, NWV * 64, 0, st>>>(sa);
}
template <int NWV, int KH>
void launch_stream_mw1(bool fast, dim3 grid, hipStream_t st, const StreamArgs& sa)
{
    if (fast) k_iter_stream<2, KH, NWV, true><<<grid, NWV * 64, 0, st>>>(sa);
    else k_iter_stream<2, KH, NWV, false><<<grid, NWV * 64, 0, st>>>(sa);
}
void launch_stream_mw(int nwv, int kh, bool fast, dim3 grid, hipStream_t st, const StreamArgs& sa)
{
    if (nwv == 4 && kh == 5) return launch_stream_mw1<4, 5>(fast, grid, st, sa);
    if (nwv == 4 && kh == 3) return launch_stream_mw1<4, 3>(fast, grid, st, sa);
#ifdef VA_EXPERIMENTS
    if (nwv == 3 && kh == 5) return launch_stream_mw1<3, 5>(fast, grid, st, sa);
    if (nwv == 3 && kh == 6) return launch_stream_mw1<3, 6>(fast, grid, st, sa);
    if (nwv == 4 && kh == 6) return launch_stream_mw1<4, 6>(fast, grid, st, sa);
#endif
    (void)nwv;
    (void)kh;
    launch_stream_mw1<4, 4>(fast, grid, st, sa);
}
// Strips of a level and the shape of the pipeline that carries them (see the comment on the four-wave forms below).  The
// forms of rounds 1-2 remain as explicit choices and as fallbacks for passes too short for four waves: two waves x 8
// levels (16 iterations per pass) where the 16-column x halo costs no third strip (w <= 224), one wave x 10 levels
// (halo 10) on wider levels.  va_tvl1_params.stream_waves = 1: one wave everywhere; 2: these two.
void stream_strips(const va_tvl1_params* p, int w, StreamPick& sp)
{
    sp.ppl = stream_ppl(p, w);
    const int SW = 64 * sp.ppl, hq = sp.ppl == 3 ? 3 : 2;  // strip origins stay multiples of the pixels per lane
    sp.two = p->tuning[VA_TUNE_STREAM_WAVES] != 1 && tiles_1d(w, SW, va_cdiv(2 * stream_kh2(sp.ppl), hq) * hq) <= 2;
    // stream_waves == 3 (experiment): where the two-wave pipeline would run, ONE wave with all 16 levels and the whole
    // register file of its SIMD (no hand-over, no barrier)
    sp.deep1 = sp.two && (p->tuning[VA_TUNE_STREAM_WAVES] == 3 || p->tuning[VA_TUNE_STREAM_WAVES] == 5) && sp.ppl == 2;
    sp.chains = sp.two && sp.ppl == 2 && (p->tuning[VA_TUNE_STREAM_WAVES] == 5 || p->tuning[VA_TUNE_STREAM_WAVES] == 6) ? 2 : 1;
    // (Also measured, round 2, and removed again: THREE waves of 4 / 5 levels each -- 12 / 15 iterations per pass at 144 /
    // 168 registers, i.e. three resident waves per SIMD instead of two: 45.2 / 43.9 ms on the 224^2 level and 35.3 / 34.8
    // on 179^2 against 41.6 / 32.9 for the two-wave form: more resident waves do not fill the idle issue slots.)
    sp.HX = va_cdiv(sp.two ? 2 * stream_kh2(sp.ppl) : stream_k1(sp.ppl), hq) * hq;
    sp.nsx = tiles_1d(w, SW, sp.HX);
    // Pipelines of FOUR waves (round 3): 4 x 4 levels (16 iterations per pass, 143 registers: three waves per SIMD) or, where
    // a 20-column halo still costs no third strip, 4 x 5 levels (20 per pass, 167 registers).  Measured per level on the
    // benchmark's pyramid (320 pairs on two streams, ms per 5 x 300 iterations, against the two-wave form): 224^2 38.7 / 41.7
    // (4 x 4, whole columns per job), 179^2 28.4 / 33.5 (4 x 5), 114^2 12.4 / 14.4 (4 x 4, two chunks): four shallow waves
    // need a quarter of the strip x chunk jobs to fill the GPU, so that the levels are cut into fewer chunks of rows (each
    // chunk costs its halo rows and the pipeline's fill), and leave room for a third wave per SIMD.
    // Levels with more than two strips (the 1280x720 pyramid): 4 x 3 levels (12 per pass, halo 12, 119 registers: four waves
    // per SIMD, three workgroups per CU) -- 138 against 118 pairs/s for the one-wave form with 10 per pass (16 pairs).
    // stream_waves: 0 = this choice, 2 = two waves where they fit (rounds 1-2), 1 = one wave everywhere, 7 / 8 / 9 = 4 x 4 /
    // 4 x 5 / 4 x 3 wherever a strip keeps valid columns, (VA_EXPERIMENTS) 10 ... 12 = 3 x 5, 3 x 6, 4 x 6.
    const int sw = p->tuning[VA_TUNE_STREAM_WAVES];
    int shape = -1;
    if (sw >= kMwFirst && sw < kMwFirst + kNumMwShapes) shape = sw - kMwFirst;
    else if (sw == 0) shape = !sp.two ? 2 : (w > SW && tiles_1d(w, SW, 20) <= 2) ? 1 : 0;
    if (shape >= 0 && sp.ppl == 2) {
        const int nwv = kMwShapes[shape][0], kh = kMwShapes[shape][1], hx = va_cdiv(nwv * kh, 2) * 2;
        // the default choice: like the two-wave form only where the deeper x halo costs no third strip; an explicit
        // stream_waves >= 7 takes the shape wherever a strip keeps valid columns
        if (2 * hx < SW && (sw >= kMwFirst || !sp.two || tiles_1d(w, SW, hx) <= 2)) {
            sp.mw_nwv = nwv;
            sp.mw_kh = kh;
            if (hx > sp.HX) sp.HX = hx;  // (a pass that falls back to the two-wave form runs with this halo too)
            sp.nsx = tiles_1d(w, SW, sp.HX);
            // the last strip starts at (nsx - 1) (SW - 2 HX); if the level (with its pitch padding) ends within 64 columns of
            // that, the strip is narrow (tile_mask bit 10: never)
            const int pitch4 = (w + 3) & ~3;
            sp.narrow = sp.nsx >= 2 && !(p->tile_mask & kNoNarrowBit) && (sp.nsx - 1) * (SW - 2 * sp.HX) + 64 >= pitch4;
        }
    }
}
StreamPick pick_stream(const va_tvl1_params* p, int w, int h, int npairs)
{
    StreamPick sp{};
    stream_strips(p, w, sp);
    // chunks of rows: the number of jobs (strip x chunk x pair) that keeps the GPU busiest was measured with one and
    // with two concurrent calls on different HIP streams: ~1024 one-wave jobs, ~640 two-wave jobs per call (256 CUs x
    // 8 waves); rows per chunk not below 32
    // (four-wave jobs: 320 per call on the levels of at most two strips -- 256 ... 400 measured equal, 480 and more 5 % slower
    // in the whole benchmark --, 1024 on the wide levels: 512 ... 1280 within 2 %, 2048 8 % slower)
    const int slots = p->tuning[VA_TUNE_STREAM_SLOTS] > 0 ? p->tuning[VA_TUNE_STREAM_SLOTS]
                      : sp.mw_nwv ? (sp.nsx <= 2 ? 320 : 1024) : (sp.two ? 640 : 1024);
    int nch = p->tuning[VA_TUNE_STREAM_CHUNKS];
    if (nch <= 0) nch = (int)((double)slots / ((double)npairs * sp.nsx) + 0.5);
    if (nch > h / 32) nch = h / 32;
    if (nch < 1) nch = 1;
    sp.R = va_cdiv(h, nch);
    sp.nch = va_cdiv(h, sp.R);
    return sp;
}
// Level (w, h) of the pyramid iterates with k_iter_stream: wherever its strips are well filled and the level is large
// enough for a GPU-full of strip x chunk jobs (measured, tools/bench_tvl1_levels.py).  tile_mask bit 8 forces it (tests);
// va_tvl1_params.stream_levels >= 0 is the explicit per-level choice (bit s = level s, 0 = never).
bool level_streams(const va_tvl1_params* p, bool eps, int s, int w, int h, size_t plane)
{
    if (eps || (double)plane * kNF_STATE * sizeof(float) >= 2147483648.0) return false;  // 32-bit buffer offsets
    if (p->tile_mask & kStreamBit) return true;
    if ((p->tile_mask & ~kNoNarrowBit) != 0) return false;
    if (p->tuning[VA_TUNE_STREAM_LEVELS] >= 0) return ((p->tuning[VA_TUNE_STREAM_LEVELS] >> s) & 1) != 0;
    StreamPick sp{};
    stream_strips(p, w, sp);
    // measured per level of the 224x224 pyramid (320 pairs, two streams; profiles/README.md): the row pipeline wins on 224^2
    // (two strips 88 % full), 179^2 (70 %) and 114^2 (one strip, 89 %), the register tiles on 143^2 (56 %) and 91^2 (71 %
    // of one strip, too few jobs); every level of the 1280x720 pyramid (80..85 %) streams
    // (round 3: with the last strips of two pairs sharing a wave 143^2 fills 74 % of 1.5 strips: 18.9 ms against 21.6 on tiles)
    const double fill = (double)w / (64.0 * sp.ppl * (sp.nsx - (sp.narrow ? 0.5 : 0.0))), px = (double)w * h;
    return (fill >= 0.85 && px >= 10000.0) || (fill >= 0.69 && px >= 20000.0);
}

// ---- k_iter_rows: which levels, which pipeline shape
constexpr int kRowsBit = 1 << 9;  // va_tvl1_params.tile_mask bit: iterate with k_iter_rows wherever it applies
struct RowsPick {
    int ppl, nwv, kh, K, K0, N;
};
#ifdef VA_EXPERIMENTS
// pipeline shapes compiled in (waves x levels per wave); rows_cfg = waves * 16 + levels per wave, 0 = the default
constexpr int kRowsShapes[][2] = {{4, 4}, {2, 8}, {3, 5}, {4, 3}, {8, 2}, {2, 6}};
constexpr int kNumRowsShapes = (int)(sizeof(kRowsShapes) / sizeof(kRowsShapes[0]));
// LDS of one k_iter_rows workgroup: the ring of per-warp constants (K + NWV rows) + the double-buffered hand-over rows
constexpr int rows_lds_bytes(int ppl, int nwv, int kh)
{
    return (nwv * kh + nwv) * kNF_RO * ppl * 256 + (nwv > 1 ? nwv - 1 : 1) * 2 * (kNF_STATE * ppl * 256 + 4);
}
constexpr int kRowsLdsMax = 160 * 1024 - 512;
#endif
bool pick_rows(const va_tvl1_params* p, int h, int pitch, RowsPick& rp)
{
#ifndef VA_EXPERIMENTS
    (void)p, (void)h, (void)pitch, (void)rp;
    return false;  // k_iter_rows is not compiled in
#else
    int nwv = kRowsShapes[0][0], kh = kRowsShapes[0][1];
    if (p->tuning[VA_TUNE_ROWS_CFG] > 0) {
        nwv = p->tuning[VA_TUNE_ROWS_CFG] >> 4;
        kh = p->tuning[VA_TUNE_ROWS_CFG] & 15;
    }
    bool known = false;
    for (int i = 0; i < kNumRowsShapes; ++i) known = known || (kRowsShapes[i][0] == nwv && kRowsShapes[i][1] == kh);
    if (!known) return false;
    rp.ppl = pitch <= 128 ? 2 : pitch <= 192 ? 3 : 4;
    if (pitch > 256) return false;
    rp.nwv = nwv;
    rp.kh = kh;
    if (rows_lds_bytes(rp.ppl, nwv, kh) > kRowsLdsMax) return false;
    const int Kfull = nwv * kh;
    rp.K = p->iters < Kfull ? p->iters : Kfull;
    rp.N = va_cdiv(p->iters, rp.K);
    rp.K0 = p->iters - (rp.N - 1) * rp.K;
    if (rp.N >= 2047 || pitch % rp.ppl != 0) return false;  // the pass index must fit the row identity; whole lanes per row
    // pass n + 1 reads a row back at least two steps after pass n stored it
    return h >= Kfull + nwv + 4 && h < (1 << kRowIdRowBits);
#endif
}

enum { LK_TILE = 0, LK_STREAM = 1, LK_ROWS = 2 };
bool level_streams(const va_tvl1_params* p, bool eps, int s, int w, int h, size_t plane);
// Which kernel iterates level s.  Explicit choices first (tests, experiments), then the measured default.
int level_kernel(const va_tvl1_params* p, bool eps, int s, int w, int h, int pitch, size_t plane, RowsPick* rp)
{
    RowsPick tmp{};
    if (!rp) rp = &tmp;
    if (eps || (double)plane * kNF_STATE * sizeof(float) >= 2147483648.0) return LK_TILE;
    const bool rows_ok = pick_rows(p, h, pitch, *rp);
    if (p->tile_mask & kRowsBit) return rows_ok ? LK_ROWS : LK_STREAM;
    if ((p->tile_mask & ~kNoNarrowBit) != 0) return level_streams(p, eps, s, w, h, plane) ? LK_STREAM : LK_TILE;
    if (p->tuning[VA_TUNE_ROWS_LEVELS] >= 0 || p->tuning[VA_TUNE_STREAM_LEVELS] >= 0) {
        if (p->tuning[VA_TUNE_ROWS_LEVELS] >= 0 && ((p->tuning[VA_TUNE_ROWS_LEVELS] >> s) & 1) && rows_ok) return LK_ROWS;
        return p->tuning[VA_TUNE_STREAM_LEVELS] >= 0 && ((p->tuning[VA_TUNE_STREAM_LEVELS] >> s) & 1) ? LK_STREAM : LK_TILE;
    }
    return level_streams(p, eps, s, w, h, plane) ? LK_STREAM : LK_TILE;
}

#ifdef VA_EXPERIMENTS
template <int PPL, bool FAST>
int launch_rows(const RowsPick& rp, const RowsArgs& a, int npairs, hipStream_t st)
{
#define VA_ROWS_CASE(NWV_, KH_)                                                           \
    if constexpr (rows_lds_bytes(PPL, NWV_, KH_) <= kRowsLdsMax) {                        \
        if (rp.nwv == NWV_ && rp.kh == KH_) {                                             \
            k_iter_rows<PPL, KH_, NWV_, FAST><<<npairs, NWV_ * 64, 0, st>>>(a);           \
            return VA_OK;                                                                 \
        }                                                                                 \
    }
    VA_ROWS_CASE(4, 4)
    VA_ROWS_CASE(2, 8)
    VA_ROWS_CASE(3, 5)
    VA_ROWS_CASE(4, 3)
    VA_ROWS_CASE(8, 2)
    VA_ROWS_CASE(2, 6)
#undef VA_ROWS_CASE
    va_set_error("va_tvl1_flow: row pipeline shape %dx%d is not compiled in", rp.nwv, rp.kh);
    return VA_ERR_INVALID;
}
#endif

// Pairs per chunk of a level: the iteration launches of a chunk re-read what the previous launch wrote, so a chunk
// whose state (64 B per pixel) stays within ~150 MB (70, 110, 200 MB measured slower) is served largely by the Infinity Cache (measured on the 179^2
// and 143^2 levels of the benchmark: -4 % each).  No chunking where a chunk could not fill the GPU.
int chunk_pairs(int lw, int lh, const TilePick& tp, int NP)
{
    const int tiles = tp.ntx * tp.nty;
    const bool w8 = kCfgs[tp.cfg].NW == 8;
    const int cp_min = (int)std::ceil((w8 ? 2.0 * 256 : 1.25 * 512) / tiles);
    const int cp_mem = (int)(VA_CHUNK_MB * 1.0e6 / ((double)lw * lh * 64.0));
    if (cp_mem < cp_min || cp_mem >= NP) return NP;
    const int n = va_cdiv(NP, cp_mem);
    return va_cdiv(NP, n);
}

// Row pitch in floats: the width rounded up to 4 (rows start 16-byte aligned: k_iter_tile's 4-pixel runs); a level that
// k_iter_rows iterates gets a multiple of 12 instead, so that a lane's 2, 3 or 4 consecutive pixels never straddle the
// end of a row.
int level_pitch(int w, bool rows) { return rows ? (w + 11) / 12 * 12 : (w + 3) / 4 * 4; }
// kernel and layout of level s: decided once, on the k_iter_rows layout (if the level does not qualify it keeps the
// plain one)
int plan_level(const va_tvl1_params* p, int s, int w, int h, int* pitch, size_t* plane, RowsPick* rp)
{
    const bool eps = p->epsilon > 0.0f;
    const int p12 = level_pitch(w, true);
    int lk = level_kernel(p, eps, s, w, h, p12, va_align_up((size_t)p12 * h, 64), rp);
    *pitch = level_pitch(w, lk == LK_ROWS);
    *plane = va_align_up((size_t)*pitch * h, 64);
    if (lk != LK_ROWS) {
        lk = level_kernel(p, eps, s, w, h, *pitch, *plane, nullptr) == LK_STREAM ? LK_STREAM : LK_TILE;
        if (lk == LK_STREAM && stream_ppl(p, w) == 3) {  // three pixels per lane: whole lanes per row need a pitch % 3 == 0
            *pitch = p12;
            *plane = va_align_up((size_t)p12 * h, 64);
        }
    }
    return lk;
}

struct Plan {
    int ns, ws[kMaxScales], hs[kMaxScales], pitch[kMaxScales];
    int lk[kMaxScales];        // LK_TILE / LK_STREAM / LK_ROWS per level
    RowsPick rows[kMaxScales];
    size_t plane[kMaxScales], plane_max;
    int NF, NP, F;
    size_t off_pyr[kMaxScales], off_tmp, off_state[2], off_ro, off_err, off_sel, off_ctl, total;
    int ctl_words;  // k_iter_stream_q: head, abort flag and one completion counter per (pair, pass)
};

int zoom_taps(float step, Taps* t)
{
    const float sigma = 0.6f * sqrtf(1.0f / (step * step) - 1.0f);
    int R = (int)(3.0f * sigma) + 1;
    if (R > kMaxRadius) R = kMaxRadius;
    double g[2 * kMaxRadius + 1], sum = 0.0;
    for (int k = -R; k <= R; ++k) {
        g[k + R] = std::exp(-(double)(k * k) / (2.0 * (double)sigma * (double)sigma));
        sum += g[k + R];
    }
    for (int k = 0; k <= 2 * R; ++k) t->g[k] = (float)(g[k] / sum);
    t->R = R;
    return R;
}

int check_params(const va_tvl1_params* p, int w, int h, int n_seq, int fps)
{
    VA_CHECK_ARG(p != nullptr, "va_tvl1: params is NULL");
    VA_CHECK_ARG(w >= 16 && h >= 16 && w <= 16384 && h <= 16384, "va_tvl1: frame size %dx%d out of range [16,16384]", w, h);
    VA_CHECK_ARG(n_seq >= 1 && fps >= 2, "va_tvl1: need n_seq >= 1 and frames_per_seq >= 2 (got %d, %d)", n_seq, fps);
    VA_CHECK_ARG((long)n_seq * (fps - 1) <= 65535, "va_tvl1: more than 65535 pairs in one call");
    VA_CHECK_ARG(p->nscales >= 1 && p->warps >= 1 && p->iters >= 1, "va_tvl1: nscales, warps, iters must be >= 1");
    VA_CHECK_ARG(p->scale_step > 0.0f && p->scale_step < 1.0f, "va_tvl1: scale_step must be in (0,1)");
    VA_CHECK_ARG(p->tau > 0.0f && p->lambda > 0.0f && p->theta > 0.0f, "va_tvl1: tau, lambda, theta must be > 0");
    VA_CHECK_ARG(p->block_iters >= 0 && p->block_iters <= 64, "va_tvl1: block_iters must be in [0,64]");
    VA_CHECK_ARG(p->fast_math == 0 || p->fast_math == 1, "va_tvl1: fast_math must be 0 or 1");
    VA_CHECK_ARG(p->tile_mask >= 0 && p->tile_mask < (1 << (kNumCfgs + 3)), "va_tvl1: tile_mask must be in [0, %d]", (1 << (kNumCfgs + 3)) - 1);
    VA_CHECK_ARG(p->tuning[VA_TUNE_ROWS_LEVELS] >= -1 && p->tuning[VA_TUNE_ROWS_LEVELS] < (1 << kMaxScales) && p->tuning[VA_TUNE_ROWS_CFG] >= 0 && p->tuning[VA_TUNE_ROWS_CFG] < 256,
                 "va_tvl1: rows_levels must be -1 or a level bit set, rows_cfg in [0,255]");
    VA_CHECK_ARG(p->tuning[VA_TUNE_STREAM_PPL] == 0 || p->tuning[VA_TUNE_STREAM_PPL] == 2 || p->tuning[VA_TUNE_STREAM_PPL] == 3, "va_tvl1: stream_ppl must be 0 (default), 2 or 3");
    VA_CHECK_ARG(p->tuning[VA_TUNE_STREAM_QUEUE] >= 0 && p->tuning[VA_TUNE_STREAM_QUEUE] <= 2, "va_tvl1: stream_queue must be 0 (default), 1 (queued) or 2 (a launch per pass)");
    VA_CHECK_ARG(p->tuning[VA_TUNE_STREAM_LEVELS] >= -1 && p->tuning[VA_TUNE_STREAM_LEVELS] < (1 << kMaxScales) && (p->tuning[VA_TUNE_STREAM_WAVES] >= 0 && p->tuning[VA_TUNE_STREAM_WAVES] < kMwFirst + kNumMwShapes) &&
                     p->tuning[VA_TUNE_STREAM_CHUNKS] >= 0 && p->tuning[VA_TUNE_STREAM_SLOTS] >= 0,
                 "va_tvl1: stream_levels must be -1 or a level bit set, stream_waves in [0, 12], stream_chunks and stream_slots >= 0");
    VA_CHECK_ARG(p->tau / p->theta <= 1000.0f && p->lambda * p->theta <= 1000.0f, "va_tvl1: tau/theta and lambda*theta must be <= 1000");
    VA_CHECK_ARG(!(p->tuning[VA_TUNE_STREAM_WAVES] == 3 && p->fast_math), "va_tvl1: stream_waves = 3 (one deep wave) is compiled for the exact arithmetic only");
    if (!kVaExperiments) {
        const int sw = p->tuning[VA_TUNE_STREAM_WAVES];
        VA_CHECK_ARG(sw != 3 && sw != 4 && sw != 5 && sw != 6 && sw <= kMwFirst + 2 && p->tuning[VA_TUNE_STREAM_PPL] != 3 && p->tuning[VA_TUNE_STREAM_QUEUE] != 1 &&
                         p->tuning[VA_TUNE_ROWS_LEVELS] <= 0 && p->tuning[VA_TUNE_ROWS_CFG] == 0 && !(p->tile_mask & kRowsBit),
                     "va_tvl1: this tuning value selects an experiment kernel (k_iter_rows, k_iter_stream_q, k_iter_stream4, one deep "
                     "wave, two chains per wave, four waves x four levels, 3 pixels per lane); build the library with -DVA_EXPERIMENTS (make EXPERIMENTS=1) to get them");
    }
    return VA_OK;
}

int pyramid_sizes(int w, int h, const va_tvl1_params* p, int* ws, int* hs)
{
    int n = 1;
    ws[0] = w;
    hs[0] = h;
    const int want = p->nscales > kMaxScales ? kMaxScales : p->nscales;
    while (n < want) {
        const int nw = (int)((float)ws[n - 1] * p->scale_step + 0.5f);
        const int nh = (int)((float)hs[n - 1] * p->scale_step + 0.5f);
        if ((nw < nh ? nw : nh) < 16) break;
        ws[n] = nw;
        hs[n] = nh;
        ++n;
    }
    return n;
}

void make_plan(Plan& P, int w, int h, int n_seq, int fps, const va_tvl1_params* p)
{
    P.ns = pyramid_sizes(w, h, p, P.ws, P.hs);
    P.F = fps;
    P.NF = n_seq * fps;
    P.NP = n_seq * (fps - 1);
    size_t off = 0;
    for (int s = 0; s < P.ns; ++s) {
        P.lk[s] = plan_level(p, s, P.ws[s], P.hs[s], &P.pitch[s], &P.plane[s], &P.rows[s]);
        P.off_pyr[s] = off;
        off += va_align_up((size_t)P.NF * 3 * P.plane[s] * sizeof(float), 256);
    }
    // the buffers every level shares are sized for the LARGEST plane: a coarser level can have the larger one when the
    // levels' pitches are padded differently (a 12-float pitch on a narrow, tall level: 16 x 100 -> pitch 16, plane 1600
    // at level 0, but pitch 24, plane 1920 at level 1 when only that level is iterated by k_iter_rows)
    P.plane_max = 0;
    for (int s = 0; s < P.ns; ++s) P.plane_max = P.plane[s] > P.plane_max ? P.plane[s] : P.plane_max;
    P.off_tmp = off;
    off += va_align_up((size_t)P.NF * 2 * P.plane_max * sizeof(float), 256);
    for (int b = 0; b < 2; ++b) {
        P.off_state[b] = off;
        off += va_align_up((size_t)P.NP * kNF_STATE * P.plane_max * sizeof(float), 256);
    }
    // k_iter_rows reaches both state buffers through one 32-bit buffer resource; a batch too large for that streams instead
    // (same plain row order, any even pitch)
    for (int s = 0; s < P.ns; ++s)
        if (P.lk[s] == LK_ROWS && (P.off_state[1] - P.off_state[0]) + (size_t)kNF_STATE * P.plane[s] * sizeof(float) >= 2147483648ull)
            P.lk[s] = LK_STREAM;
    P.off_ro = off;
    off += va_align_up((size_t)P.NP * kNF_RO * P.plane_max * sizeof(float), 256);
    P.off_err = off;
    if (p->epsilon > 0.0f) off += va_align_up((size_t)P.NP * p->iters * sizeof(unsigned long long), 256);
    P.off_sel = off;
    off += va_align_up((size_t)P.NP * 2 * sizeof(int), 256);
    P.off_ctl = off;
    P.ctl_words = 2 + P.NP * (p->iters / (2 * kStreamKH2) + 2);
    off += va_align_up((size_t)P.ctl_words * sizeof(unsigned), 256);
    P.total = off;
}

va_prof_span prof_get(va_ctx* ctx)
{
    va_prof_span s{};
    if (!ctx->prof_pool.empty()) {
        s = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
    } else {
        (void)hipEventCreate(&s.beg);
        (void)hipEventCreate(&s.end);
    }
    return s;
}

}  // namespace

extern "C" void va_tvl1_default_params(va_tvl1_params* p)
{
    if (!p) return;
    p->tau = 0.25f;
    p->lambda = 0.15f;
    p->theta = 0.3f;
    p->nscales = 5;
    p->warps = 5;
    p->epsilon = 0.01f;
    p->iters = 300;
    p->scale_step = 0.8f;
    p->block_iters = 0;
    p->fast_math = 0;
    p->tile_mask = 0;
    p->tuning[VA_TUNE_STREAM_LEVELS] = -1;
    p->tuning[VA_TUNE_STREAM_WAVES] = 0;
    p->tuning[VA_TUNE_STREAM_CHUNKS] = 0;
    p->tuning[VA_TUNE_STREAM_SLOTS] = 0;
    p->tuning[VA_TUNE_ROWS_LEVELS] = -1;
    p->tuning[VA_TUNE_ROWS_CFG] = 0;
    p->tuning[VA_TUNE_STREAM_PPL] = 0;
    p->tuning[VA_TUNE_STREAM_QUEUE] = 0;
}

extern "C" int va_tvl1_pyramid_sizes(int w, int h, const va_tvl1_params* p, int* ws, int* hs)
{
    if (!p || !ws || !hs || w < 1 || h < 1 || !(p->scale_step > 0.0f && p->scale_step < 1.0f)) return 0;
    return pyramid_sizes(w, h, p, ws, hs);
}

// The register tiling va_tvl1_flow will use (host logic only; no device needed): for every pyramid level s,
// out[6*s .. 6*s+5] = { tile width, tile height, waves per workgroup, block depth K, tiles in x, tiles in y }.
extern "C" int va_tvl1_tile_plan(int w, int h, const va_tvl1_params* p, int* out)
{
    if (!p || !out || w < 16 || h < 16 || !(p->scale_step > 0.0f && p->scale_step < 1.0f) || p->iters < 1) return 0;
    int ws[kMaxScales], hs[kMaxScales];
    const int ns = pyramid_sizes(w, h, p, ws, hs);
    int K0 = p->block_iters;
    if (p->epsilon > 0.0f) K0 = 1;
    for (int s = 0; s < ns; ++s) {
        const unsigned tmask = (unsigned)p->tile_mask & (unsigned)(kStreamBit - 1);
        const TilePick tp = K0 > 0 ? pick_tiles(ws[s], hs[s], K0, tmask) : pick_tiles_auto(ws[s], hs[s], p->iters, tmask);
        const TileCfg& c = kCfgs[tp.cfg];
        int lp;
        size_t lplane;
        RowsPick rp{};
        const int lk = plan_level(p, s, ws[s], hs[s], &lp, &lplane, &rp);
        if (lk == LK_ROWS) {
            const int plan[6] = {64 * rp.ppl, 0, rp.nwv, rp.K, 1, 0};
            memcpy(out + 6 * s, plan, sizeof(plan));
            continue;
        }
        if (lk == LK_STREAM) {
            StreamPick sp{};
            stream_strips(p, ws[s], sp);
            const int plan[6] = {64 * sp.ppl, 0, sp.mw_nwv ? sp.mw_nwv : sp.two ? 2 : 1,
                                 sp.mw_nwv ? sp.mw_nwv * sp.mw_kh : sp.two ? 2 * stream_kh2(sp.ppl) : stream_k1(sp.ppl), sp.nsx, 0};
            memcpy(out + 6 * s, plan, sizeof(plan));
            continue;
        }
        out[6 * s + 0] = c.LX * c.R;
        out[6 * s + 1] = c.NW * (64 / c.LX) * c.C;
        out[6 * s + 2] = c.NW;
        out[6 * s + 3] = tp.K;
        out[6 * s + 4] = tp.ntx;
        out[6 * s + 5] = tp.nty;
    }
    return ns;
}

extern "C" size_t va_tvl1_workspace_bytes(int w, int h, int n_seq, int frames_per_seq, const va_tvl1_params* p)
{
    if (check_params(p, w, h, n_seq, frames_per_seq) != VA_OK) return 0;
    Plan P;
    make_plan(P, w, h, n_seq, frames_per_seq, p);
    return P.total;
}

extern "C" int va_tvl1_flow(va_ctx* ctx, const void* frames, int frames_are_u8, int n_seq, int frames_per_seq, int w,
                            int h, const va_tvl1_params* p, void* flow, void* workspace, size_t workspace_bytes,
                            void* stream)
{
    VA_CHECK_ARG(ctx != nullptr, "va_tvl1_flow: ctx is NULL");
    VA_USE_DEVICE(ctx);
    VA_CHECK_ARG(frames != nullptr && flow != nullptr && workspace != nullptr, "va_tvl1_flow: NULL buffer");
    if (int rc = check_params(p, w, h, n_seq, frames_per_seq)) return rc;
    VA_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "va_tvl1_flow: workspace must be 256-byte aligned");
    Plan P;
    make_plan(P, w, h, n_seq, frames_per_seq, p);
    if (workspace_bytes < P.total) {
        va_set_error("va_tvl1_flow: workspace too small (%zu < %zu bytes)", workspace_bytes, P.total);
        return VA_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* pyr[kMaxScales];
    for (int s = 0; s < P.ns; ++s) pyr[s] = (float*)(ws + P.off_pyr[s]);
    float* tmp1 = (float*)(ws + P.off_tmp);
    float* tmp2 = tmp1 + (size_t)P.NF * P.plane_max;
    float* state[2] = {(float*)(ws + P.off_state[0]), (float*)(ws + P.off_state[1])};
    float* ro = (float*)(ws + P.off_ro);
    unsigned long long* err = (unsigned long long*)(ws + P.off_err);
    int* sel = (int*)(ws + P.off_sel);
    int* base = sel + P.NP;
    const bool eps = p->epsilon > 0.0f;
    const int TPB = 256;

    // S0/S1: level 0, then the pyramid of every FRAME (shared by both pairs it belongs to).
    {
        const dim3 g(va_cdiv(w * h, TPB), P.NF);
        if (frames_are_u8)
            k_frames_to_level0<unsigned char><<<g, TPB, 0, st>>>((const unsigned char*)frames, pyr[0], w, h, P.pitch[0], P.plane[0]);
        else
            k_frames_to_level0<float><<<g, TPB, 0, st>>>((const float*)frames, pyr[0], w, h, P.pitch[0], P.plane[0]);
        VA_LAUNCH_CHECK();
    }
    Taps taps;
    zoom_taps(p->scale_step, &taps);
    for (int s = 1; s < P.ns; ++s) {
        const int pw = P.ws[s - 1], ph = P.hs[s - 1], pp = P.pitch[s - 1];
        const dim3 g(va_cdiv(pw * ph, TPB), P.NF);
        k_gauss<false><<<g, TPB, 0, st>>>(pyr[s - 1], 3 * P.plane[s - 1], tmp1, P.plane_max, pw, ph, pp, taps);
        k_gauss<true><<<g, TPB, 0, st>>>(tmp1, P.plane_max, tmp2, P.plane_max, pw, ph, pp, taps);
        const dim3 g2(va_cdiv(P.ws[s] * P.hs[s], TPB), P.NF);
        k_resample<<<g2, TPB, 0, st>>>(tmp2, P.plane_max, pw, ph, pp, pyr[s], 3 * P.plane[s], P.ws[s], P.hs[s], P.pitch[s]);
        VA_LAUNCH_CHECK();
    }
    for (int s = 0; s < P.ns; ++s) {
        const dim3 g(va_cdiv(P.ws[s] * P.hs[s], TPB), P.NF);
        k_grad<<<g, TPB, 0, st>>>(pyr[s], P.ws[s], P.hs[s], P.pitch[s], P.plane[s]);
        VA_LAUNCH_CHECK();
    }

    // S4: u = 0, p = 0 at the coarsest level.
    const int sc = P.ns - 1;
    VA_HIP(hipMemsetAsync(state[0], 0, (size_t)P.NP * kNF_STATE * P.plane[sc] * sizeof(float), st));
    VA_HIP(hipMemsetAsync(sel, 0, (size_t)P.NP * 2 * sizeof(int), st));
    int cur = 0;
    int K0 = p->block_iters;  // 0 = per-level automatic choice (12 on every level of the 224x224 pyramid's top three)
    if (eps) K0 = 1;

    for (int s = sc; s >= 0; --s) {
        const int lw = P.ws[s], lh = P.hs[s], lp = P.pitch[s];
        const size_t plane = P.plane[s];
        const unsigned tmask = (unsigned)p->tile_mask & (unsigned)(kStreamBit - 1);
        const TilePick tp = K0 > 0 ? pick_tiles(lw, lh, K0, tmask) : pick_tiles_auto(lw, lh, p->iters, tmask);
        const RowsPick rp = P.rows[s];
        const int lk = P.lk[s];
        // k_iter_rows addresses both state buffers through one 32-bit buffer resource
        const size_t st_lo = state[0] < state[1] ? 0 : 1;
        const size_t st_span = (size_t)((char*)state[st_lo ^ 1] - (char*)state[st_lo]) + (size_t)kNF_STATE * plane * sizeof(float);
        if (lk == LK_ROWS && st_span >= 2147483648ull) {
            va_set_error("va_tvl1_flow: internal error: state buffers too far apart for k_iter_rows");
            return VA_ERR_INVALID;
        }
        const bool strm = lk == LK_STREAM, rows = lk == LK_ROWS;
        const int perm = (strm || rows) ? 0 : 1;
        if (lp != lw) {
            k_zero_pad<<<dim3(va_cdiv((lp - lw) * lh * kNF_RO, TPB), P.NP), TPB, 0, st>>>(ro, kNF_RO, lw, lh, lp, plane, perm);
            VA_LAUNCH_CHECK();
        }
        // the pairs of a level go through its warps x iterations in chunks (cache residency: chunk_pairs); every chunk
        // starts from the level's entry buffer index and ends on the same one
        const int cp = (eps || !VA_CHUNK || strm || rows) ? P.NP : chunk_pairs(lw, lh, tp, P.NP);
        const int cur_in = cur;
        for (int c0 = 0; c0 < P.NP; c0 += cp) {
        const int nc = P.NP - c0 < cp ? P.NP - c0 : cp;
        const dim3 gpc(va_cdiv(lw * lh, TPB), nc);
        cur = cur_in;
        for (int wp = 0; wp < p->warps; ++wp) {
            k_warp<<<gpc, TPB, 0, st>>>(pyr[s], plane, lw, lh, lp, P.F, state[0], state[1], eps ? sel : nullptr, cur,
                                         eps ? base : nullptr, ro, c0, perm);
            VA_LAUNCH_CHECK();
            if (eps) VA_HIP(hipMemsetAsync(err, 0, (size_t)P.NP * p->iters * sizeof(unsigned long long), st));
            va_prof_span span{};
            if (ctx->prof_on) {
                span = prof_get(ctx);
                span.level = s;
                VA_HIP(hipEventRecord(span.beg, st));
            }
            IterArgs a{};
            a.ro = ro;
            a.stA = state[0];
            a.stB = state[1];
            a.outA = state[0];
            a.outB = state[1];
            a.base = base;
            a.sel = sel;
            a.err = err;
            a.qthr = eps ? (unsigned long long)((double)p->epsilon * (double)p->epsilon * (double)lw * (double)lh * 4294967296.0) : 0ull;
            a.plane = plane;
            a.w = lw;
            a.h = lh;
            a.pitch = lp;
            a.ntx = tp.ntx;
            a.nty = tp.nty;
            a.HX = tp.HX;
            a.iters = p->iters;
            a.l_t = p->lambda * p->theta;
            a.taut = p->tau / p->theta;
            a.theta = p->theta;
            a.pair0 = c0;
            int launches = 0;
#ifdef VA_EXPERIMENTS
            if (rows) {
                RowsArgs ra{};
                ra.ro = ro;
                ra.st = state[st_lo];
                ra.delta[st_lo] = 0u;
                ra.delta[st_lo ^ 1] = (unsigned)((char*)state[st_lo ^ 1] - (char*)state[st_lo]);
                ra.plane = plane;
                ra.w = lw;
                ra.h = lh;
                ra.pitch = lp;
                ra.K = rp.K;
                ra.K0 = rp.K0;
                ra.N = rp.N;
                ra.cur = cur;
                ra.pair0 = c0;
                ra.l_t = a.l_t;
                ra.taut = a.taut;
                ra.theta = a.theta;
                int rc;
                if (rp.ppl == 2) rc = p->fast_math ? launch_rows<2, true>(rp, ra, nc, st) : launch_rows<2, false>(rp, ra, nc, st);
                else if (rp.ppl == 3) rc = p->fast_math ? launch_rows<3, true>(rp, ra, nc, st) : launch_rows<3, false>(rp, ra, nc, st);
                else rc = p->fast_math ? launch_rows<4, true>(rp, ra, nc, st) : launch_rows<4, false>(rp, ra, nc, st);
                if (rc) return rc;
                cur ^= rp.N & 1;
                launches = 1;
            }
#else
            (void)rp, (void)st_lo;
#endif
            if (strm) {
                const StreamPick sp = pick_stream(p, lw, lh, nc);
                StreamArgs sa{};
                sa.ro = ro;
                sa.plane = plane;
                sa.w = lw;
                sa.h = lh;
                sa.pitch = lp;
                sa.nsx = sp.nsx;
                sa.nch = sp.nch;
                sa.R = sp.R;
                sa.HX = sp.HX;
                sa.pair0 = c0;
                sa.l_t = a.l_t;
                sa.taut = a.taut;
                sa.theta = a.theta;
                const dim3 grid(sp.nsx * sp.nch, nc);
                // the four-wave passes of a level whose last strip fits 32 lanes: one wave for the last strips of two pairs
                const bool narrow = sp.narrow != 0;
                const dim3 grid_mw = narrow ? dim3((2 * sp.nsx - 1) * sp.nch, va_cdiv(nc, 2)) : grid;
                sa.npairs = nc;
                const bool two = sp.two != 0;
                bool queued = false;
#ifdef VA_EXPERIMENTS
                // the queued form (all passes in one launch): two-wave pipeline, 2 pixels per lane, and a last pass deep
                // enough to end in the second wave
                const int qK = 2 * kStreamKH2, qn = va_cdiv(p->iters, qK), qlast = p->iters - (qn - 1) * qK;
                queued = two && sp.ppl == 2 && !sp.deep1 && qlast > kStreamKH2 &&
                         (p->tuning[VA_TUNE_STREAM_QUEUE] == 1 || (p->tuning[VA_TUNE_STREAM_QUEUE] == 0 && VA_STREAM_QUEUE_DEFAULT));
                if (queued) {
                    unsigned* ctl = (unsigned*)(ws + P.off_ctl);
                    VA_HIP(hipMemsetAsync(ctl, 0, (size_t)P.ctl_words * sizeof(unsigned), st));
                    StreamQArgs qa{};
                    qa.base = sa;
                    qa.st[0] = state[0];
                    qa.st[1] = state[1];
                    qa.ctl = ctl;
                    qa.cur = cur;
                    qa.npass = qn;
                    qa.K = qK;
                    qa.Klast = qlast;
                    qa.npairs = nc;
                    qa.tpp = sp.nsx * sp.nch;
                    const int want = p->tuning[VA_TUNE_STREAM_SLOTS] > 0 ? p->tuning[VA_TUNE_STREAM_SLOTS] : 512;  // persistent workgroups: half the GPU's 1024 slots per call
                    const int nwg = qa.npairs * qa.tpp < want ? qa.npairs * qa.tpp : want;
                    if (p->fast_math) k_iter_stream_q<2, kStreamKH2, 2, true><<<nwg, 128, 0, st>>>(qa);
                    else k_iter_stream_q<2, kStreamKH2, 2, false><<<nwg, 128, 0, st>>>(qa);
                    VA_LAUNCH_CHECK();
                    k_poison_if_aborted<<<nc, 64, 0, st>>>(ctl, state[0] + (size_t)c0 * kNF_STATE * plane, state[1] + (size_t)c0 * kNF_STATE * plane,
                                                            (size_t)kNF_STATE * plane, lw);
                    cur ^= qn & 1;
                    launches = 1;
                }
                const bool four = sp.ppl == 2 && (p->tuning[VA_TUNE_STREAM_WAVES] == 4 || (p->tuning[VA_TUNE_STREAM_WAVES] == 0 && VA_STREAM4_DEFAULT));
#endif
                // pipelines of three or four waves: the passes share the iterations as evenly as possible, so that every pass
                // is deep enough to end in the last wave (K > (waves - 1) x levels per wave); otherwise the forms below run
                int mw_n = 0, mw_base = 0, mw_extra = 0;
                if (sp.mw_nwv && !queued) {
                    const int kmax = sp.mw_nwv * sp.mw_kh, kmin = (sp.mw_nwv - 1) * sp.mw_kh + 1;
                    mw_n = va_cdiv(p->iters, kmax);
                    mw_base = p->iters / mw_n;
                    mw_extra = p->iters % mw_n;
                    if (mw_base < kmin) mw_n = 0;
                }
                for (int i = 0; i < mw_n; ++i) {
                    sa.K = mw_base + (i < mw_extra ? 1 : 0);
                    sa.sin = state[cur];
                    sa.sout = state[cur ^ 1];
                    sa.rev = VA_REV ? (launches & 1) : 0;
                    sa.narrow = narrow ? 1 : 0;
                    launch_stream_mw(sp.mw_nwv, sp.mw_kh, p->fast_math != 0, grid_mw, st, sa);
                    sa.narrow = 0;
                    cur ^= 1;
                    ++launches;
                }
                for (int it = (queued || mw_n) ? p->iters : 0; it < p->iters;) {
                    const int rem = p->iters - it;
                    const int kh2 = stream_kh2(sp.ppl), k1 = stream_k1(sp.ppl);
                    const bool w2 = two && rem > kh2 && !sp.deep1;  // the two-wave kernel needs its last level in the second wave
                    sa.K = sp.deep1 ? (rem < 16 ? rem : 16) : w2 ? (rem < 2 * kh2 ? rem : 2 * kh2) : (rem < k1 ? rem : k1);
                    sa.sin = state[cur];
                    sa.sout = state[cur ^ 1];
                    sa.rev = VA_REV ? (launches & 1) : 0;
#ifdef VA_EXPERIMENTS
                    if (sp.deep1 && sp.chains == 2) {  // round 3: two interleaved chains of levels per wave (measured slower)
                        if (p->fast_math) k_iter_stream<2, 16, 1, true, 2><<<grid, 64, 0, st>>>(sa);
                        else k_iter_stream<2, 16, 1, false, 2><<<grid, 64, 0, st>>>(sa);
                    } else if (w2 && sp.chains == 2) {
                        if (p->fast_math) k_iter_stream<2, kStreamKH2, 2, true, 2><<<grid, 128, 0, st>>>(sa);
                        else k_iter_stream<2, kStreamKH2, 2, false, 2><<<grid, 128, 0, st>>>(sa);
                    } else if (sp.deep1) {
                        k_iter_stream<2, 16, 1, false><<<grid, 64, 0, st>>>(sa);
                    } else if (w2 && four) {
                        const int njobs = (int)(grid.x * grid.y), g4 = va_cdiv(njobs, 4);
                        if (p->fast_math) k_iter_stream4<true><<<g4, 512, 0, st>>>(sa, (int)grid.x, njobs);
                        else k_iter_stream4<false><<<g4, 512, 0, st>>>(sa, (int)grid.x, njobs);
                    } else
#endif
                    if (w2) {
                        if (p->fast_math) launch_stream<true, true>(sp.ppl, grid, st, sa);
                        else launch_stream<true, false>(sp.ppl, grid, st, sa);
                    } else {
                        if (p->fast_math) launch_stream<false, true>(sp.ppl, grid, st, sa);
                        else launch_stream<false, false>(sp.ppl, grid, st, sa);
                    }
                    cur ^= 1;
                    it += sa.K;
                    ++launches;
                }
            }
            for (int it = (strm || rows) ? p->iters : 0; it < p->iters;) {
                // the tile grid depends on the halo depth: a shorter last launch gets its own grid
                const int k = (p->iters - it) < tp.K ? (p->iters - it) : tp.K;
                const TilePick tk = (k == tp.K) ? tp : pick_tiles(lw, lh, k, tmask);
                a.ntx = tk.ntx;
                a.nty = tk.nty;
                a.HX = tk.HX;
                a.cur = cur;
                a.K = tk.K;
                a.it = it;
                a.rev = VA_REV ? (launches & 1) : 0;
                if (eps) {
                    if (p->fast_math) launch_iter<true, true>(tk, a, nc, st);
                    else launch_iter<true, false>(tk, a, nc, st);
                } else {
                    if (p->fast_math) launch_iter<false, true>(tk, a, nc, st);
                    else launch_iter<false, false>(tk, a, nc, st);
                }
                cur ^= 1;
                it += tk.K;
                ++launches;
            }
            VA_LAUNCH_CHECK();
            if (ctx->prof_on) {
                VA_HIP(hipEventRecord(span.end, st));
                ctx->prof_spans.push_back(span);
                ctx->prof_launches += launches;
                ctx->prof_pxiters += (double)nc * lw * lh * p->iters;
                ctx->prof_pxwarps += (double)nc * lw * lh;
                if (s < kVaProfLevels) {
                    ctx->prof_level_pxiters[s] += (double)nc * lw * lh * p->iters;
                    ctx->prof_level_launches[s] += launches;
                }
            }
        }
        }
        if (s > 0) {
            // ro is free between levels: use it as the upsampling target (2 of its 4 planes per pair)
            const dim3 g(va_cdiv(P.ws[s - 1] * P.hs[s - 1], TPB), P.NP);
            k_upsample<<<g, TPB, 0, st>>>(state[0], state[1], eps ? sel : nullptr, cur, lw, lh, lp, plane, ro, P.ws[s - 1],
                                           P.hs[s - 1], P.pitch[s - 1], P.plane[s - 1], 1.0f / p->scale_step, perm);
            const int fperm = P.lk[s - 1] != LK_TILE ? 0 : 1;
            const dim3 gi(va_cdiv(P.pitch[s - 1] * P.hs[s - 1], TPB), P.NP);
            k_level_init<<<gi, TPB, 0, st>>>(ro, state[0], P.ws[s - 1], P.hs[s - 1], P.pitch[s - 1], P.plane[s - 1], fperm);
            VA_LAUNCH_CHECK();
            cur = 0;
            if (eps) VA_HIP(hipMemsetAsync(sel, 0, (size_t)P.NP * sizeof(int), st));
        }
    }
    {
        const dim3 g(va_cdiv(w * h, TPB), P.NP);
        k_flow_out<<<g, TPB, 0, st>>>(state[0], state[1], eps ? sel : nullptr, cur, w, h, P.pitch[0], P.plane[0], (float*)flow,
                                       P.lk[0] != LK_TILE ? 0 : 1);
        VA_LAUNCH_CHECK();
    }
    return VA_OK;
}

extern "C" int va_flow_to_stack(va_ctx* ctx, const void* flow, int n_pairs, int w, int h, float bound, float mean,
                                float stdv, void* stack, void* stream)
{
    VA_CHECK_ARG(ctx != nullptr, "va_flow_to_stack: ctx is NULL");
    VA_USE_DEVICE(ctx);
    VA_CHECK_ARG(flow != nullptr && stack != nullptr, "va_flow_to_stack: NULL buffer");
    VA_CHECK_ARG(n_pairs >= 1 && w >= 1 && h >= 1, "va_flow_to_stack: bad shape");
    VA_CHECK_ARG(bound > 0.0f && stdv > 0.0f, "va_flow_to_stack: bound and std must be > 0");
    const size_t n = (size_t)n_pairs * 2 * w * h;
    k_flow_to_stack<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>((const float*)flow, (float*)stack, n, bound, mean, stdv);
    VA_LAUNCH_CHECK();
    return VA_OK;
}

extern "C" int va_selftest_exact_math(va_ctx* ctx, float lo, float hi, unsigned long long* mismatches, void* stream)
{
    VA_CHECK_ARG(ctx != nullptr && mismatches != nullptr, "va_selftest_exact_math: NULL argument");
    VA_USE_DEVICE(ctx);
    VA_CHECK_ARG(lo >= kSqrtReg && hi <= 1e30f && lo <= hi, "va_selftest_exact_math: range must lie in [2^-100, 1e30]");
    unsigned ulo, uhi;
    memcpy(&ulo, &lo, 4);
    memcpy(&uhi, &hi, 4);
    VA_HIP(hipMemsetAsync(mismatches, 0, 2 * sizeof(unsigned long long), (hipStream_t)stream));
    k_selftest_exact_math<<<4096, 256, 0, (hipStream_t)stream>>>(ulo, (unsigned long long)uhi - ulo + 1, mismatches);
    VA_LAUNCH_CHECK();
    return VA_OK;
}

extern "C" int va_tvl1_profile_enable(va_ctx* ctx, int on)
{
    VA_CHECK_ARG(ctx != nullptr, "va_tvl1_profile_enable: ctx is NULL");
    VA_USE_DEVICE(ctx);
    ctx->prof_on = on != 0;
    if (on) {
        // time origin: everything recorded later (on any stream) is measured against it
        if (!ctx->prof_ref) VA_HIP(hipEventCreate(&ctx->prof_ref));
        VA_HIP(hipDeviceSynchronize());
        VA_HIP(hipEventRecord(ctx->prof_ref, nullptr));
        VA_HIP(hipEventSynchronize(ctx->prof_ref));
    }
    return VA_OK;
}

extern "C" int va_tvl1_profile_read(va_ctx* ctx, double* out, int reset)
{
    VA_CHECK_ARG(ctx != nullptr && out != nullptr, "va_tvl1_profile_read: NULL argument");
    VA_USE_DEVICE(ctx);
    std::vector<std::pair<float, float>> iv;
    for (va_prof_span& s : ctx->prof_spans) {
        VA_HIP(hipEventSynchronize(s.end));
        float ms = 0.0f, t0 = 0.0f, t1 = 0.0f;
        VA_HIP(hipEventElapsedTime(&ms, s.beg, s.end));
        ctx->prof_ms += ms;
        if (s.level >= 0 && s.level < kVaProfLevels) ctx->prof_level_ms[s.level] += ms;
        if (ctx->prof_ref) {
            VA_HIP(hipEventElapsedTime(&t0, ctx->prof_ref, s.beg));
            VA_HIP(hipEventElapsedTime(&t1, ctx->prof_ref, s.end));
            iv.emplace_back(t0, t1);
        }
        ctx->prof_pool.push_back(s);
    }
    ctx->prof_spans.clear();
    // union of the spans: with several streams the launches of different calls overlap in time
    std::sort(iv.begin(), iv.end());
    float cur0 = 0.0f, cur1 = -1.0f;
    for (auto& p : iv) {
        if (cur1 < cur0 || p.first > cur1) {
            if (cur1 >= cur0) ctx->prof_union_ms += cur1 - cur0;
            cur0 = p.first;
            cur1 = p.second;
        } else if (p.second > cur1) {
            cur1 = p.second;
        }
    }
    if (cur1 >= cur0) ctx->prof_union_ms += cur1 - cur0;
    out[0] = ctx->prof_ms;
    out[1] = ctx->prof_launches;
    out[2] = ctx->prof_pxiters;
    out[3] = ctx->prof_pxwarps;
    out[4] = ctx->prof_union_ms;
    if (reset) ctx->prof_ms = ctx->prof_union_ms = ctx->prof_launches = ctx->prof_pxiters = ctx->prof_pxwarps = 0.0;
    return VA_OK;
}

// Per pyramid level (0 = full resolution), since the last reset: out[3*s + 0] = summed per-call milliseconds of the
// level's inner-iteration launches, out[3*s + 1] = pixel-iterations, out[3*s + 2] = launches (HOST array of 3*n
// doubles, n <= 16).  Call after va_tvl1_profile_read (which synchronises the events) and before its reset.
extern "C" int va_tvl1_profile_levels(va_ctx* ctx, double* out, int n, int reset)
{
    VA_CHECK_ARG(ctx != nullptr && out != nullptr && n >= 1 && n <= kVaProfLevels, "va_tvl1_profile_levels: bad argument");
    for (int s = 0; s < n; ++s) {
        out[3 * s + 0] = ctx->prof_level_ms[s];
        out[3 * s + 1] = ctx->prof_level_pxiters[s];
        out[3 * s + 2] = ctx->prof_level_launches[s];
    }
    if (reset)
        for (int s = 0; s < kVaProfLevels; ++s) ctx->prof_level_ms[s] = ctx->prof_level_pxiters[s] = ctx->prof_level_launches[s] = 0.0;
    return VA_OK;
}
