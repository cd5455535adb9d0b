// VGG-16 'D' feature stack + the reference's 4-layer classifier for gfx950 (MI355X).
//
// Stands behind self.features(ip) and the classifierList traversal of validate()
// (Sheet03/spatialModel.py:110-113,127-129,136-152,212-218; temporal twin
// Sheet03/temporalModel.py:122-126,149-181,241-247).  Hand-written HIP, no MIOpen / hipBLASLt,
// no CPU fallback.
//
// Design
//   * activations live in HBM as NHWC fp32 (channels innermost) so that the implicit-GEMM A
//     operand (one output pixel's 3x3xCin patch) is 9 contiguous channel runs and every global
//     access is a 16-byte lane access;
//   * conv3x3+bias+ReLU(+maxpool2x2) is ONE kernel: implicit GEMM  D[m][n] = sum_k A[m][k] W[n][k],
//     m = output pixel, n = output channel, k = (ky,kx,ci), on v_mfma_f32_32x32x2_f32 (fp32 in,
//     fp32 accumulate: exact fp32 products, one rounding per FMA -- DESIGN.md "numerics");
//   * the 128 pixels of an M tile are a (TB images) x (TH rows) x (TW cols) brick whose two lowest
//     index bits are (x&1, y&1): the four pixels of a 2x2 pooling window are then the four
//     accumulator registers (reg&3) of ONE lane, so the fused max-pool is register-only;
//   * weights are repacked once into [Cout][9*Cin_pad] (K-contiguous, same shape as the A tile);
//     the first layer's input channels are zero-padded to 16 (spatial, 3) / 32 (temporal, 20);
//   * FC layers: the same MFMA tile as a split-K "NT" GEMM over the batch (M = 32 rows per tile),
//     partial slabs reduced by a second kernel that fuses bias+ReLU (deterministic, no atomics).
//     FC1's weight is repacked so that the NHWC feature map can be used without a transpose while
//     keeping the reference's CHW-major flatten order c*49+h*7+w (Sheet03/spatialModel.py:213).
#include "vgg_internal.h"
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));       // native vectors for the staging registers:
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));    // arrays of HIP's float4/uint4 structs were kept in scratch

// Compile-time loop: the body sees its index as a constant expression, so register arrays indexed by it can
// never be demoted to scratch (a `#pragma unroll` loop over rb[] in the conv kernels was: hipcc kept the
// prefetched weight tile in private memory and waited for every global load right after issuing it).
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

constexpr int kBK = 16;       // K chunk per LDS tile (floats)
constexpr int kLdsStride = 20;  // floats per LDS row: 80 B keeps ds_read_b128 conflict-free

// ---------------------------------------------------------------- layout / packing kernels ----

// x NCHW (f32, or u8 with ToTensor+Normalize: Sheet03/utils.py:148-150) -> NHWC (f32 or bf16) with C padded
// to cpad <= 64.  One workgroup transposes 64 pixels x cpad channels through LDS: reads are coalesced along
// the pixels of a channel plane, writes are one contiguous run of 64*cpad elements.
template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256) k_nchw_to_nhwc_pad(const TIN* __restrict__ x, TOUT* __restrict__ out, int B, int C,
                                                           int HW, int cpad, const float* __restrict__ mean,
                                                           const float* __restrict__ stdv)
{
    __shared__ float tile[64][65];
    const int tiles_per_img = (HW + 63) / 64;
    const int b = blockIdx.x / tiles_per_img, p0 = (blockIdx.x - b * tiles_per_img) * 64;
    const int t = threadIdx.x, p = t & 63;
    for (int c = t >> 6; c < cpad; c += 4) {
        float v = 0.0f;
        if (c < C && p0 + p < HW) {
            const TIN raw = x[((size_t)b * C + c) * HW + p0 + p];
            if constexpr (sizeof(TIN) == 1) v = ((float)raw / 255.0f - mean[c]) / stdv[c];
            else v = (float)raw;
        }
        tile[p][c] = v;
    }
    __syncthreads();
    const int npx = HW - p0 < 64 ? HW - p0 : 64;
    TOUT* o = out + ((size_t)b * HW + p0) * cpad;
    for (int i = t; i < npx * cpad; i += 256) o[i] = (TOUT)tile[i / cpad][i % cpad];
}

// bf16 first layer (3*C <= 64): NCHW -> NHWC bf16 [B][HW][64] holding, for every pixel, the three horizontal
// taps x-1, x, x+1 of all C channels (channel j = kx*C + c; zeros outside the image and for j >= 3*C).  The
// 3x3 convolution of the first layer then needs 3 K steps (one per kernel row) instead of 9, on an input of
// the same size as the plain 64-channel padding.
template <typename TIN>
__global__ void __launch_bounds__(256) k_nchw_to_nhwc_xcol(const TIN* __restrict__ x, __bf16* __restrict__ out, int B, int C, int W,
                                                            int HW, const float* __restrict__ mean, const float* __restrict__ stdv)
{
    __shared__ float tile[66][21];  // pixels p0-1 .. p0+64, C <= 21
    const int tiles_per_img = (HW + 63) / 64;
    const int b = blockIdx.x / tiles_per_img, p0 = (blockIdx.x - b * tiles_per_img) * 64;
    const int t = threadIdx.x;
    for (int i = t; i < 66 * C; i += 256) {
        const int c = i / 66, pl = i - c * 66, p = p0 + pl - 1;
        float v = 0.0f;
        if (p >= 0 && p < HW) {
            const TIN raw = x[((size_t)b * C + c) * HW + p];
            if constexpr (sizeof(TIN) == 1) v = ((float)raw / 255.0f - mean[c]) / stdv[c];
            else v = (float)raw;
        }
        tile[pl][c] = v;
    }
    __syncthreads();
    const int npx = HW - p0 < 64 ? HW - p0 : 64;
    __bf16* o = out + ((size_t)b * HW + p0) * 64;
    for (int i = t; i < npx * 64; i += 256) {
        const int pl = i >> 6, j = i & 63, kx = j / C, c = j - kx * C;
        const int xx = (p0 + pl) % W + kx - 1;
        o[i] = (__bf16)((kx < 3 && xx >= 0 && xx < W) ? tile[pl + kx][c] : 0.0f);
    }
}

// OIHW f32 [Cout][Cin][3][3] -> bf16 [Cout][3 (ky)][64]: element kx*Cin + ci of row ky (the layout above)
__global__ void k_pack_conv_w_bf16_xcol(const float* __restrict__ w, __bf16* __restrict__ wp, int Cout, int Cin)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * 3 * 64) return;
    const int j = idx & 63, ky = (idx >> 6) % 3, n = idx / 192;
    const int kx = j / Cin, ci = j - kx * Cin;
    wp[idx] = (__bf16)(kx < 3 ? w[((size_t)n * Cin + ci) * 9 + ky * 3 + kx] : 0.0f);
}

// OIHW [Cout][Cin][3][3] -> [Cout][9][cpad]
__global__ void k_pack_conv_w(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int cpad)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)Cout * 9 * cpad;
    if (idx >= total) return;
    const int ci = (int)(idx % cpad);
    const int kp = (int)((idx / cpad) % 9);
    const int n = (int)(idx / ((size_t)cpad * 9));
    wp[idx] = ci < Cin ? w[((size_t)n * Cin + ci) * 9 + kp] : 0.0f;
}

// FC1 [out][c*HW + p] -> [out][p*C + c]
__global__ void k_pack_fc1(const float* __restrict__ w, float* __restrict__ wp, int O, int C, int HW)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)O * C * HW;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    const int p = (int)((idx / C) % HW);
    const size_t o = idx / ((size_t)C * HW);
    wp[idx] = w[(o * C + c) * HW + p];
}

// NCHW [B][C][HW] -> NHWC [B][HW][C]  (small tensors: the classifier's feature-map input)
__global__ void k_nchw_to_nhwc(const float* __restrict__ in, float* __restrict__ out, int B, int C, int HW)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * C * HW) return;
    const int c = (int)(idx % C);
    const int p = (int)((idx / C) % HW);
    const size_t b = idx / ((size_t)HW * C);
    out[idx] = in[(b * C + c) * HW + p];
}

// NHWC [B][HW][C] -> NCHW [B][C][HW]  (optional `feat` output)
__global__ void k_nhwc_to_nchw(const float* __restrict__ in, float* __restrict__ out, int B, int C, int HW)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * C * HW) return;
    const int p = (int)(idx % HW);
    const int c = (int)((idx / HW) % C);
    const size_t b = idx / ((size_t)HW * C);
    out[idx] = in[(b * HW + p) * C + c];
}

// Sheet03/temporalModel.py:155-161
__global__ void k_copy_first_layer(const float* __restrict__ w, float* __restrict__ out, int cout, int n_in)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cout * n_in * 9) return;
    const int kp = idx % 9, o = idx / (9 * n_in);
    float avg = 0.0f;
    avg = avg + w[(o * 3 + 0) * 9 + kp];
    avg = avg + w[(o * 3 + 1) * 9 + kp];
    avg = avg + w[(o * 3 + 2) * 9 + kp];
    out[idx] = avg / 3.0f;
}

// ---------------------------------------------------------------- conv3x3 implicit GEMM --------

struct ConvArgs {
    const float* in;    // NHWC [B][H][W][Cin], Cin % 16 == 0
    const float* wp;    // [Cout][9*Cin]
    const float* bias;  // [Cout]
    float* out;         // NHWC [B][H][W][Cout] or pooled [B][H/2][W/2][Cout]
    int B, H, W, Cin, Cout;
    int lgTW, lgTH;     // log2 of the brick's width/height (>= 1)
    int TB;
    int tiles_x, tiles_y, tiles_n;
    const float* mask;  // training (dgrad): zero the output where mask[same index] <= 0 (NULL: no mask; not with POOL)
    int linear;         // 1: no ReLU (training: dgrad)
    const float* zeros; // DMA kernel: >= 16 zero bytes, the source of every out-of-image tap
};

#ifndef VA_XCD_REMAP
#define VA_XCD_REMAP 1
#endif
// XCD-aware workgroup order: consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), so
// the channel tiles of one pixel tile and its neighbours (which share the A operand and its halo) would land on
// different L2s.  This bijective remap gives every XCD a contiguous run of tiles instead.
__device__ __forceinline__ int xcd_remap(unsigned bid, unsigned nb)
{
#if VA_XCD_REMAP
    const unsigned q = nb / 8, r = nb % 8, xcd = bid % 8, k = bid / 8;
    return (int)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k);
#else
    return (int)bid;
#endif
}

__device__ __forceinline__ void brick_coords(int m, int lgTW, int lgTH, int& xl, int& yl, int& bl)
{
    const int x0 = m & 1, y0 = (m >> 1) & 1;
    int r = m >> 2;
    const int xh = r & ((1 << (lgTW - 1)) - 1);
    r >>= (lgTW - 1);
    const int yh = r & ((1 << (lgTH - 1)) - 1);
    r >>= (lgTH - 1);
    xl = 2 * xh + x0;
    yl = 2 * yh + y0;
    bl = r;
}

// BM = WM*MT*32 output pixels x BN = WN*NT*32 output channels per workgroup of WM*WN waves.
// BK = channels per K step (16 or 32): a pixel's BK-channel run is BK*4 contiguous bytes, so BK = 32
// reads whole 128-byte lines and halves the barriers per FLOP.
template <int WM, int WN, int MT, int NT, bool POOL, int BK>
__global__ void __launch_bounds__(WM * WN * 64) k_conv3x3_mfma(ConvArgs a)
{
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32, NTHR = WM * WN * 64;
    constexpr int QPR = BK / 4, LDS = BK + 4;  // float4 per row; padded LDS row (conflict-free ds_read_b128)
    constexpr int A_ROWS_PER_PASS = NTHR / QPR, A_PASSES = BM / A_ROWS_PER_PASS, B_PASSES = BN / A_ROWS_PER_PASS;
    static_assert(BM % A_ROWS_PER_PASS == 0 && BN % A_ROWS_PER_PASS == 0, "tile/thread mismatch");
    __shared__ __attribute__((aligned(16))) float sA[2][BM * LDS];
    __shared__ __attribute__((aligned(16))) float sB[2][BN * LDS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n_tile = bid % a.tiles_n;
    bid /= a.tiles_n;
    const int tile_x = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int tile_y = bid % a.tiles_y;
    const int tile_b = bid / a.tiles_y;
    const int n0 = n_tile * BN;
    const int H = a.H, W = a.W, Cin = a.Cin;
    const int X0 = tile_x << a.lgTW, Y0 = tile_y << a.lgTH, B0 = tile_b * a.TB;

    // loader role: float4 column q of rows tid/QPR + A_ROWS_PER_PASS*i
    const int q = tid % QPR, rowbase = tid / QPR;
    int ax[A_PASSES], ay[A_PASSES];
    long apix[A_PASSES];
    bool aok[A_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        int xl, yl, bl;
        brick_coords(rowbase + A_ROWS_PER_PASS * i, a.lgTW, a.lgTH, xl, yl, bl);
        ax[i] = X0 + xl;
        ay[i] = Y0 + yl;
        const int b = B0 + bl;
        aok[i] = b < a.B && ax[i] < W && ay[i] < H;
        apix[i] = (((long)b * H + ay[i]) * W + ax[i]) * Cin + 4 * q;
    }
    const float* wrow[B_PASSES];
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) wrow[i] = a.wp + (size_t)(n0 + rowbase + A_ROWS_PER_PASS * i) * 9 * Cin + 4 * q;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const int cchunks = Cin / BK;
    const int T = 9 * cchunks;
    // Staging registers of the next K tile.  (Written as macros, not lambdas: with the arrays captured by
    // reference hipcc kept rb[] in scratch and waited for every prefetch right after issuing it.)
    f32x4 ra[A_PASSES], rb[B_PASSES];
#define VA_CONV_GLOAD(t_)                                                                                       \
    {                                                                                                           \
        const int kp_ = (t_) / cchunks, c0_ = ((t_)-kp_ * cchunks) * BK;                                        \
        const int ky_ = kp_ / 3 - 1, kx_ = kp_ % 3 - 1;                                                         \
        static_for<A_PASSES>([&](auto I) {                                                                      \
            constexpr int i = decltype(I)::value;                                                               \
            const int yy_ = ay[i] + ky_, xx_ = ax[i] + kx_;                                                     \
            const bool ok_ = aok[i] && yy_ >= 0 && yy_ < H && xx_ >= 0 && xx_ < W;                              \
            ra[i] = ok_ ? *reinterpret_cast<const f32x4*>(a.in + apix[i] + ((long)ky_ * W + kx_) * Cin + c0_)   \
                        : f32x4{0.f, 0.f, 0.f, 0.f};                                                            \
        });                                                                                                     \
        static_for<B_PASSES>([&](auto I) {                                                                      \
            constexpr int i = decltype(I)::value;                                                               \
            rb[i] = *reinterpret_cast<const f32x4*>(wrow[i] + (size_t)kp_ * Cin + c0_);                         \
        });                                                                                                     \
    }
#define VA_CONV_LSTORE(buf_)                                                                                    \
    {                                                                                                           \
        static_for<A_PASSES>([&](auto I) {                                                                      \
            constexpr int i = decltype(I)::value;                                                               \
            *reinterpret_cast<f32x4*>(&sA[buf_][(rowbase + A_ROWS_PER_PASS * i) * LDS + 4 * q]) = ra[i];         \
        });                                                                                                     \
        static_for<B_PASSES>([&](auto I) {                                                                      \
            constexpr int i = decltype(I)::value;                                                               \
            *reinterpret_cast<f32x4*>(&sB[buf_][(rowbase + A_ROWS_PER_PASS * i) * LDS + 4 * q]) = rb[i];         \
        });                                                                                                     \
    }

    VA_CONV_GLOAD(0)
    VA_CONV_LSTORE(0)
    __syncthreads();
    const int r31 = lane & 31, hh = lane >> 5;
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        if (t + 1 < T) VA_CONV_GLOAD(t + 1)
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float4 fa[MT], fb[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                fa[mt] = *reinterpret_cast<const float4*>(&sA[buf][((wm * MT + mt) * 32 + r31) * LDS + 8 * g + 4 * hh]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                fb[nt] = *reinterpret_cast<const float4*>(&sB[buf][((wn * NT + nt) * 32 + r31) * LDS + 8 * g + 4 * hh]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt].x, fb[nt].x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt].y, fb[nt].y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt].z, fb[nt].z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt].w, fb[nt].w, acc[mt][nt], 0, 0, 0);
                }
        }
        if (t + 1 < T) VA_CONV_LSTORE(buf ^ 1)
        __syncthreads();
    }
#undef VA_CONV_GLOAD
#undef VA_CONV_LSTORE

    // epilogue: bias + ReLU (+ 2x2 max-pool over the 4 registers reg&3 of a lane)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + (wn * NT + nt) * 32 + r31;
        const float bias = a.bias[n];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                // rows m = (wm*MT+mt)*32 + 8*g4 + 4*hh + (0..3)
                const int mbase = (wm * MT + mt) * 32 + 8 * g4 + 4 * hh;
                int xl, yl, bl;
                brick_coords(mbase, a.lgTW, a.lgTH, xl, yl, bl);
                const int x = X0 + xl, y = Y0 + yl, b = B0 + bl;  // (x,y) even: the 2x2 window's origin
                if (b >= a.B) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = acc[mt][nt][4 * g4 + j] + bias;
                    if (!a.linear) v[j] = fmaxf(v[j], 0.0f);
                }
                if constexpr (POOL) {
                    if (x < W && y < H) {
                        const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                        a.out[((((size_t)b * (H >> 1)) + (y >> 1)) * (W >> 1) + (x >> 1)) * a.Cout + n] = mx;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int xx = x + (j & 1), yy = y + (j >> 1);
                        if (xx < W && yy < H) {
                            const size_t o = (((size_t)b * H + yy) * W + xx) * a.Cout + n;
                            a.out[o] = (a.mask && !(a.mask[o] > 0.0f)) ? 0.0f : v[j];
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------- bf16 conv3x3 (BASELINE config 5) ------
//
// Same implicit GEMM and the same pooling-friendly pixel brick, with bf16 activations/weights and fp32
// accumulation on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).  K step = 64 channels: a pixel's
// chunk is one 128-byte line; LDS rows are padded to 144 B so that the 16-byte fragment reads of the 32
// rows of a half-wave fall on distinct bank groups.  The first layer's input channels are zero-padded
// to 64.  Results deviate from the fp32 path at the bf16 level (~1e-2 relative on class scores): this is
// the throughput configuration, not the parity configuration (DESIGN.md "Numerics of the CNN").

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kBfBK = 64;      // channels per K step

// OIHW f32 [Cout][Cin][3][3] -> bf16 [Cout][9][cpad]
__global__ void k_pack_conv_w_bf16(const float* __restrict__ w, __bf16* __restrict__ wp, int Cout, int Cin, int cpad)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)Cout * 9 * cpad;
    if (idx >= total) return;
    const int ci = (int)(idx % cpad);
    const int kp = (int)((idx / cpad) % 9);
    const int n = (int)(idx / ((size_t)cpad * 9));
    wp[idx] = (__bf16)(ci < Cin ? w[((size_t)n * Cin + ci) * 9 + kp] : 0.0f);
}

struct ConvArgsBf {
    const __bf16* in;     // NHWC [B][H][W][Cin], Cin % 64 == 0
    const __bf16* wp;     // [Cout][9*Cin]
    const float* bias;    // [Cout]
    void* out;            // NHWC bf16 (or f32 when OUT_F32), pooled when POOL
    const __bf16* zeros;  // >= 16 zero bytes: the source of every out-of-image tap
    int B, H, W, Cin, Cout;
    int lgTW, lgTH, TB;
    int tiles_x, tiles_y, tiles_n;
    int taps_x;           // 3: 3x3 taps; 1: the x taps are folded into the channels (k_nchw_to_nhwc_xcol), 3 row taps
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// Epilogue of the bf16 kernels: bias + ReLU (+ 2x2 max-pool) in fp32, then the wave's tile goes through a wave-private
// LDS tile so that whole runs of NT * 64 bytes per pixel reach memory, 16 bytes per lane (instead of one 2-byte store
// per value).  The MFMA operand order follows the layer:
//   * no pooling: the WEIGHT fragment is the first operand -- accumulator (mt, nt), register 4 g4 + j of lane (r31, hh) is
//     channel 32 nt + 8 g4 + 4 hh + j of pixel 32 mt + r31: four consecutive channels of one pixel = one 8-byte LDS write;
//   * pooling: the PIXEL fragment is the first operand -- register 4 g4 + j is pixel 32 mt + 8 g4 + 4 hh + j of channel
//     32 nt + r31: the four registers are one pixel quad of the brick order, their maximum one 2-byte LDS write.
// The tile's rows are NT * 64 bytes, unpadded; the 16-byte chunk c of row r sits at chunk c ^ ((r / RPL) & (CH8 - 1)),
// RPL = rows per 256-byte bank line: the 16 lanes of a write or read group then touch 16 different bank positions.
// `stage`: 64 * NT * 64 bytes of LDS that no other wave touches; m0: the tile's first pixel in the workgroup's brick
// order; nb: its first channel.
template <bool POOL>
constexpr bool kWeightsFirst = !POOL;
template <int NT, bool POOL>
__device__ __forceinline__ void epilogue_lines_bf16(const f32x16 (&acc)[2][NT], __bf16* stage, const ConvArgsBf& a, int m0, int nb,
                                                    int X0, int Y0, int B0, int lane)
{
    constexpr int ROW = NT * 32, CH8 = NT * 4, RPL = NT >= 4 ? 1 : 4 / NT;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const int r31 = lane & 31, hh = lane >> 5;
    const int H = a.H, W = a.W;
    if constexpr (POOL) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = 32 * nt + r31;
            const float bias = a.bias[nb + n];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[mt][nt][4 * g4 + j] + bias, 0.0f);
                    const int row = 8 * mt + 2 * g4 + hh;  // pooled pixel (32 mt + 8 g4 + 4 hh) / 4
                    const int col = (((n >> 3) ^ ((row / RPL) & (CH8 - 1))) << 3) | (n & 7);
                    stage[row * ROW + col] = (__bf16)fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                }
        }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = 32 * nt + 8 * g4 + 4 * hh;
                const float4 bs = *(const float4*)(a.bias + nb + n);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    bf16x4 v;
                    v.x = (__bf16)fmaxf(acc[mt][nt][4 * g4 + 0] + bs.x, 0.0f);
                    v.y = (__bf16)fmaxf(acc[mt][nt][4 * g4 + 1] + bs.y, 0.0f);
                    v.z = (__bf16)fmaxf(acc[mt][nt][4 * g4 + 2] + bs.z, 0.0f);
                    v.w = (__bf16)fmaxf(acc[mt][nt][4 * g4 + 3] + bs.w, 0.0f);
                    const int row = 32 * mt + r31;
                    const int col = (((n >> 3) ^ ((row / RPL) & (CH8 - 1))) << 3) | (n & 4);
                    *(bf16x4*)(stage + row * ROW + col) = v;
                }
            }
    }
    __builtin_amdgcn_wave_barrier();  // (wave-private tile: the wave's LDS writes are ordered before its reads)
    constexpr int ROWS = POOL ? 16 : 64;
#pragma unroll
    for (int it = 0; it < ROWS * CH8 / 64; ++it) {
        const int ch = it * 64 + lane, pl = ch / CH8, c8 = ch - pl * CH8;
        const uint4 v = *(const uint4*)(stage + pl * ROW + 8 * (c8 ^ ((pl / RPL) & (CH8 - 1))));
        int xl, yl, bl;
        brick_coords(m0 + (POOL ? 4 * pl : pl), a.lgTW, a.lgTH, xl, yl, bl);
        const int x = X0 + xl, y = Y0 + yl, b = B0 + bl;
        if (b < a.B && x < W && y < H) {
            const size_t o = POOL ? ((((size_t)b * (H >> 1)) + (y >> 1)) * (W >> 1) + (x >> 1)) * a.Cout
                                  : (((size_t)b * H + y) * W + x) * a.Cout;
            *(uint4*)((__bf16*)a.out + o + nb + 8 * c8) = v;
        }
    }
}

// 128 pixels x (NT*64) channels per workgroup of 4 waves (2 x 2), wave tile 64 x (NT*32).
//
// Staging: LDS-DMA (global_load_lds_dwordx4), no staging registers and no ds_write pass.  One wave
// instruction fills 1 KB = 8 rows x 128 B of the tile; the LDS image is lane-linear, so the bank swizzle
// sits on the SOURCE side: lane (row r, slot p) fetches 16-byte chunk c = p ^ ((r >> 1) & 7) of its row's
// 128-byte line (still one whole line per 8 lanes), and a fragment read of chunk c of row m goes to slot
// c ^ ((m >> 1) & 7): the 16 lanes of a ds_read_b128 group (16 consecutive rows, same c) then cover the
// 16 slots of the 256-byte bank row exactly once.  Out-of-image taps read a zero line.  One LDS buffer
// (32 KB at NT = 2), two barriers per K step, <= 128 VGPRs: four workgroups per CU hide each other's
// load latency (measured: a second LDS buffer with the next step's DMA in flight, at two workgroups per
// CU, is 8 % slower).
//
// NBUF = 1: the scheme above.  NBUF >= 3: a ring of NBUF tiles with NBUF-1 K steps of DMA in flight (counted
// s_waitcnt vmcnt, raw s_barrier, one barrier per step) for the layers that cannot put several workgroups on
// every CU (14x14: 392 workgroups of 72 latency-bound steps).
template <int NT, bool POOL, bool OUT_F32, int NBUF>
__device__ __forceinline__ void conv3x3_mfma_bf16_body(const ConvArgsBf& a)
{
    constexpr int BM = 128, BN = NT * 64, MT = 2;
    constexpr int A_INSTR = BM / 32, B_INSTR = BN / 32;  // LDS-DMA instructions per wave and K step (8 rows each)
    constexpr int TILE = (BM + BN) * kBfBK;  // bf16 elements of one K step's A and B tiles
    __shared__ __attribute__((aligned(1024))) __bf16 smem[NBUF * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n_tile = bid % a.tiles_n;
    bid /= a.tiles_n;
    const int tile_x = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int tile_y = bid % a.tiles_y;
    const int tile_b = bid / a.tiles_y;
    const int n0 = n_tile * BN;
    const int H = a.H, W = a.W, Cin = a.Cin;
    const int X0 = tile_x << a.lgTW, Y0 = tile_y << a.lgTH, B0 = tile_b * a.TB;

    // loader role: row (lane >> 3) of the 8-row group, slot (lane & 7).
    // Addressing: buffer loads to LDS.  The activation resource starts (W + 1) pixels BEFORE the tensor, so that the tap
    // offset ((ky + 1) W + kx + 1) Cin + c0 is a non-negative scalar (soffset) and a lane's own offset carries the same
    // bias; a lane whose tap falls outside the image gets an out-of-range offset, for which the buffer load writes zeros
    // (no zero line, no 64-bit pointer select per piece and step); a lane's nine in-image decisions are nine bits.
    const int lrow = lane >> 3, lslot = lane & 7;
    const long bias_el = (long)(W + 1) * Cin;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(a.in) - bias_el, 0, (int)(((long)a.B * H * W * Cin + 2 * bias_el) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(a.wp), 0, (int)((long)a.Cout * 3 * a.taps_x * Cin * 2), 0x00020000);
    int aoff[A_INSTR];
    unsigned amask[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int row = (wave * A_INSTR + i) * 8 + lrow;
        const int c = lslot ^ ((row >> 1) & 7);
        int xl, yl, bl;
        brick_coords(row, a.lgTW, a.lgTH, xl, yl, bl);
        const int x = X0 + xl, y = Y0 + yl, b = B0 + bl;
        const bool ok = b < a.B && x < W && y < H;
        aoff[i] = (int)(((((long)b * H + y) * W + x) * Cin + 8 * c) * 2);
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int ky = a.taps_x == 3 ? k / 3 - 1 : k - 1, kx = a.taps_x == 3 ? k % 3 - 1 : 0;
            const int yy = y + ky, xx = x + kx;
            m |= (ok && k < 3 * a.taps_x && yy >= 0 && yy < H && xx >= 0 && xx < W) ? 1u << k : 0u;
        }
        amask[i] = m;
    }
    int boff[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int row = (wave * B_INSTR + i) * 8 + lrow;
        boff[i] = (int)(((long)(n0 + row) * 3 * a.taps_x * Cin + 8 * (lslot ^ ((row >> 1) & 7))) * 2);
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const int cchunks = Cin / kBfBK;
    const int T = 3 * a.taps_x * cchunks;
    const int r31 = lane & 31, hh = lane >> 5;
    const int fsw = (r31 >> 1) & 7;  // fragment rows are (multiple of 32) + r31
    int kp = 0, kx = 0, c0 = 0;
    // scalar byte offsets of the step: ((ky + 1) W + kx + 1) Cin + c0 (activations, biased), kp Cin + c0 (weights)
    int so_a = a.taps_x == 3 ? 0 : Cin * 2, so_b = 0;
    // LDS-DMA of K step (kp, c0) into ring slot `buf`; advances (kp, c0)
    auto stage = [&](int buf) {
        __bf16* const sA = smem + buf * TILE;
        __bf16* const sB = sA + BM * kBfBK;
        static_for<A_INSTR>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const int vo = (amask[i] >> kp) & 1 ? aoff[i] : 0x7fffffff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(sA + (wave * A_INSTR + i) * 8 * kBfBK), 16, vo, so_a, 0, 0);
        });
        static_for<B_INSTR>([&](auto I) {
            constexpr int i = decltype(I)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(sB + (wave * B_INSTR + i) * 8 * kBfBK), 16, boff[i], so_b, 0, 0);
        });
        c0 += kBfBK;
        so_a += kBfBK * 2;
        so_b += kBfBK * 2;
        if (c0 == Cin) {  // next tap: one pixel to the right, or (3 x 3 taps: two pixels short of) a row down
            c0 = 0;
            ++kp;
            if (a.taps_x == 3) {
                ++kx;
                so_a += kx == 3 ? (W - 3) * Cin * 2 : 0;
                kx = kx == 3 ? 0 : kx;
            } else {
                so_a += (W - 1) * Cin * 2;
            }
        }
    };
    auto compute = [&](int buf) {
        const __bf16* const sA = smem + buf * TILE;
        const __bf16* const sB = sA + BM * kBfBK;
#pragma unroll
        for (int ks = 0; ks < kBfBK / 16; ++ks) {
            bf16x8 fa[MT], fb[NT];
            const int slot = ((2 * ks + hh) ^ fsw) * 8;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                fa[mt] = *reinterpret_cast<const bf16x8*>(&sA[((wm * MT + mt) * 32 + r31) * kBfBK + slot]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                fb[nt] = *reinterpret_cast<const bf16x8*>(&sB[((wn * NT + nt) * 32 + r31) * kBfBK + slot]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = (OUT_F32 || !kWeightsFirst<POOL>) ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0)
                                                                    : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
        }
    };
    if constexpr (NBUF == 1) {
        for (int t = 0; t < T; ++t) {
            stage(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();  // every wave has read the tile before the next DMA overwrites it
        }
    } else {
        constexpr int LOADS = A_INSTR + B_INSTR;  // this wave's DMA instructions per K step
        // prologue: NBUF-1 steps in flight (T >= 3 >= NBUF-1 is not guaranteed for NBUF > 4: launch only NBUF <= 4)
        for (int j = 0; j < NBUF - 1 && j < T; ++j) stage(j);
        int slot = 0;
        for (int t = 0; t < T; ++t) {
            // wait until this wave's DMAs of step t have landed: steps t+1 .. t+NBUF-2 may stay in flight
            if (t + NBUF - 2 < T) {
                if constexpr ((NBUF - 2) * LOADS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if constexpr ((NBUF - 2) * LOADS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if constexpr ((NBUF - 2) * LOADS == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if constexpr ((NBUF - 2) * LOADS == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // all waves: step t's tile is complete, and step t-1's reads are done, so its slot can be refilled
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + NBUF - 1 < T) stage(slot == 0 ? NBUF - 1 : slot - 1);
            compute(slot);
            slot = slot + 1 == NBUF ? 0 : slot + 1;
        }
    }

    if constexpr (!OUT_F32) {
        // (weights as the MFMA's first operand: a lane owns channel quads of one pixel) whole lines through LDS
        if constexpr (NBUF != 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // every wave has read the last tile: the ring becomes the staging tiles
        }
        epilogue_lines_bf16<NT, POOL>(acc, smem + wave * 64 * NT * 32, a, wm * MT * 32, n0 + wn * NT * 32, X0, Y0, B0, lane);
        return;
    }
    // epilogue: bias + ReLU (+ 2x2 max-pool over the 4 registers reg&3 of a lane), fp32 math, bf16/f32 store
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + (wn * NT + nt) * 32 + r31;
        const float bias = a.bias[n];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int mbase = (wm * MT + mt) * 32 + 8 * g4 + 4 * hh;
                int xl, yl, bl;
                brick_coords(mbase, a.lgTW, a.lgTH, xl, yl, bl);
                const int x = X0 + xl, y = Y0 + yl, b = B0 + bl;
                if (b >= a.B) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[mt][nt][4 * g4 + j] + bias, 0.0f);
                if constexpr (POOL) {
                    if (x < W && y < H) {
                        const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                        const size_t o = ((((size_t)b * (H >> 1)) + (y >> 1)) * (W >> 1) + (x >> 1)) * a.Cout + n;
                        if constexpr (OUT_F32) ((float*)a.out)[o] = mx;
                        else ((__bf16*)a.out)[o] = (__bf16)mx;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int xx = x + (j & 1), yy = y + (j >> 1);
                        if (xx < W && yy < H) {
                            const size_t o = (((size_t)b * H + yy) * W + xx) * a.Cout + n;
                            if constexpr (OUT_F32) ((float*)a.out)[o] = v[j];
                            else ((__bf16*)a.out)[o] = (__bf16)v[j];
                        }
                    }
                }
            }
        }
    }
}

// (the body is a __device__ function: buffer-resource builtins in a __global__ template body keep the host pass from
// emitting the kernel's stub)
template <int NT, bool POOL, bool OUT_F32, int NBUF>
__global__ void __launch_bounds__(256, NBUF == 1 ? 4 : 2) k_conv3x3_mfma_bf16(ConvArgsBf a)
{
    conv3x3_mfma_bf16_body<NT, POOL, OUT_F32, NBUF>(a);
}

// ---------------------------------------------------------------- bf16 first layer, fused with the input conversion ----
//
// The first layer has 3 (RGB) or 20 (flow stack) input channels: its K is 27 / 180, its cost is the 205 MB of bf16
// activations it writes per 32 images.  Staging its input as a 64-channel NHWC tensor first (k_nchw_to_nhwc_xcol) writes
// and re-reads another 205 MB.  k_conv1_fused_bf16 reads the NCHW input directly: a workgroup owns a 16 x 16 pixel brick
// of one image and all 64 output channels,
//   * converts the brick's 18 x 18 halo patch (u8: ToTensor + Normalize, the expression of k_nchw_to_nhwc_pad; zeros
//     outside the image) to bf16 in LDS, pixel-major with Cp = C rounded up to 4 channels per pixel: the 3 Cp values of
//     the pixels x-1, x, x+1 of one patch row are then CONTIGUOUS -- they are row ky of the pixel's im2col line
//     (element kx Cp + c), and an MFMA fragment is 8 consecutive elements of that run (two ds_read_b64; a run is rounded
//     up to KROW = 16-element blocks whose tail reads the next pixels' finite values against zero weights);
//   * holds the packed weights [64][3][KROW] in LDS (k_pack_conv_w_bf16_f1);
//   * multiplies with the WEIGHTS as the MFMA's first operand: a lane then owns four consecutive output channels of one
//     pixel per accumulator quad, packs them to 8 bytes, and the wave transposes its 64 pixels x 64 channels through LDS
//     so that every pixel's 128-byte line goes to memory whole (16 bytes per lane, 8 full lines per store instruction).
// K = 3 KROW: 48 for RGB (3 MFMA k-blocks per accumulator), 192 for the flow stack (12).
struct Conv1Args {
    const void* x;       // NCHW [B][C][224][224], f32 or u8
    const __bf16* wp;    // [64][3][KROW]
    const float* bias;   // [64]
    __bf16* out;         // NHWC [B][224][224][64]
    const float* mean;   // u8 input: ToTensor + Normalize
    const float* stdv;
    int B, C, Cp, KROW;
};

constexpr int kF1MaxCp = 24, kF1MaxKrow = 80;

__global__ void k_pack_conv_w_bf16_f1(const float* __restrict__ w, __bf16* __restrict__ wp, int Cout, int Cin, int Cp, int KROW)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout * 3 * KROW) return;
    const int j = idx % KROW, ky = (idx / KROW) % 3, n = idx / (3 * KROW);
    const int kx = j / Cp, ci = j - kx * Cp;
    wp[idx] = (__bf16)((kx < 3 && ci < Cin) ? w[((size_t)n * Cin + ci) * 9 + ky * 3 + kx] : 0.0f);
}

template <typename TIN>
__global__ void __launch_bounds__(256) k_conv1_fused_bf16(Conv1Args a)
{
    constexpr int H = 224, W = 224, TW = 16, PW = 18, NPIX = 18 * 18, NPAD = NPIX + 12;
    constexpr int PATCH_EL = NPAD * kF1MaxCp;             // bf16 elements
    constexpr int WS_MAX = 3 * kF1MaxKrow + 8;            // weight row stride (elements): + 16 B against bank conflicts
    constexpr int OST = 64 + 8;                           // staging row stride (elements): 144 B
    constexpr int SMEM_A = (PATCH_EL + 64 * WS_MAX) * 2, SMEM_B = 4 * 64 * OST * 2;
    __shared__ __attribute__((aligned(16))) char smem[SMEM_A > SMEM_B ? SMEM_A : SMEM_B];
    __bf16* const patch = (__bf16*)smem;
    __bf16* const wl = patch + PATCH_EL;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int C = a.C, Cp = a.Cp, KROW = a.KROW, WS = 3 * KROW + 8, NBK = KROW / 16;
    int bid = blockIdx.x;
    const int tx = bid % (W / TW);
    bid /= (W / TW);
    const int ty = bid % (H / TW);
    const int b = bid / (H / TW);
    const int X0 = tx * TW, Y0 = ty * TW;

    // zero the patch (pad channels, pad pixels), then fill it; copy the weights
    for (int i = tid; i < NPAD * Cp / 8; i += 256) ((uint4*)patch)[i] = uint4{0, 0, 0, 0};
    for (int i = tid; i < 64 * 3 * KROW / 8; i += 256) {
        const int n = i / (3 * KROW / 8), r = i - n * (3 * KROW / 8);
        *(uint4*)(wl + n * WS + 8 * r) = ((const uint4*)a.wp)[i];
    }
    __syncthreads();
    // the patch: rows of 18 pixels starting at x = X0 - 1, fetched as the six ALIGNED groups of four pixels that cover
    // [X0 - 4, X0 + 20) (a group is wholly inside or wholly outside the image); U groups in flight per thread
    const TIN* const xin = (const TIN*)a.x + (size_t)b * C * H * W;
    typedef TIN tin4 __attribute__((ext_vector_type(4)));
    constexpr int U = 3;
    const int items = C * PW * 6;
    for (int i0 = 0; i0 < items; i0 += 256 * U) {
        tin4 raw[U];
        int dst[U], cc[U], q4[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256 + tid;
            const int c = i / (PW * 6), r = i - c * (PW * 6), py = r / 6, q = r - py * 6;
            const int gy = Y0 - 1 + py, gx = X0 - 4 + 4 * q;
            ok[u] = i < items && gy >= 0 && gy < H && gx >= 0 && gx < W;  // (outside the image the patch stays zero)
            raw[u] = tin4{0, 0, 0, 0};
            if (ok[u]) raw[u] = *(const tin4*)(xin + ((size_t)c * H + gy) * W + gx);
            cc[u] = c;
            q4[u] = 4 * q - 3;                       // patch column of the group's first pixel
            dst[u] = (py * PW + q4[u]) * Cp + c;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int px = q4[u] + e;
                if (px < 0 || px >= PW) continue;
                float v;
                if constexpr (sizeof(TIN) == 1) v = ((float)raw[u][e] / 255.0f - a.mean[cc[u]]) / a.stdv[cc[u]];
                else v = (float)raw[u][e];
                patch[dst[u] + e * Cp] = (__bf16)v;
            }
        }
    }
    __syncthreads();

    const int r31 = lane & 31, hh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
    // this lane's pixels: m = 64 wave + 32 mt + r31, row-major in the 16 x 16 brick
    int pbase[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = 64 * wave + 32 * mt + r31;
        pbase[mt] = ((m >> 4) * PW + (m & 15)) * Cp;  // patch pixel (y - 1, x - 1): tap (ky, kx) = (0, 0) of pixel m
    }
    typedef short s4 __attribute__((ext_vector_type(4)));
    typedef short s8 __attribute__((ext_vector_type(8)));
    for (int ky = 0; ky < 3; ++ky)
        for (int kb = 0; kb < NBK; ++kb) {
            const int j0 = kb * 16 + hh * 8;
            bf16x8 fx[2], fw[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const s4* p = (const s4*)(patch + pbase[mt] + ky * PW * Cp + j0);
                const s4 lo = p[0], hi = p[1];
                const s8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                fx[mt] = __builtin_bit_cast(bf16x8, v);
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) fw[nt] = *(const bf16x8*)(wl + (32 * nt + r31) * WS + ky * KROW + j0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[nt], fx[mt], acc[mt][nt], 0, 0, 0);
        }
    __syncthreads();  // every wave is done with the patch and the weights: the staging tiles take their place

    // accumulator (mt, nt), register 4 g4 + j: channel 32 nt + 8 g4 + 4 hh + j of pixel 32 mt + r31 of this wave
    __bf16* const ost = (__bf16*)smem + wave * 64 * OST;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int n = 32 * nt + 8 * g4 + 4 * hh;
            const float4 bs = *(const float4*)(a.bias + n);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                bf16x4 v;
                v.x = (__bf16)fmaxf(acc[mt][nt][4 * g4 + 0] + bs.x, 0.0f);
                v.y = (__bf16)fmaxf(acc[mt][nt][4 * g4 + 1] + bs.y, 0.0f);
                v.z = (__bf16)fmaxf(acc[mt][nt][4 * g4 + 2] + bs.z, 0.0f);
                v.w = (__bf16)fmaxf(acc[mt][nt][4 * g4 + 3] + bs.w, 0.0f);
                *(bf16x4*)(ost + (32 * mt + r31) * OST + n) = v;
            }
        }
    // (the tile is private to the wave: its own LDS writes are ordered before its reads)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int ch = it * 64 + lane, pl = ch >> 3, c8 = ch & 7;
        const uint4 v = *(const uint4*)(ost + pl * OST + 8 * c8);
        const int m = 64 * wave + pl, y = Y0 + (m >> 4), x = X0 + (m & 15);
        *(uint4*)(a.out + (((size_t)b * H + y) * W + x) * 64 + 8 * c8) = v;
    }
}

// ---------------------------------------------------------------- bf16 conv3x3, two wave groups in turns ------
//
// k_conv3x3_pp_bf16: 512 threads = two GROUPS of four waves (one wave of each group on every SIMD).  The workgroup tile is
// 256 pixels x (NT * 64) channels; group g owns the pixels [128 g, 128 g + 128), its four waves (2 x 2) a 64 x (NT * 32)
// tile each.  The K loop runs in steps of 32 channels of one tap, and every step of a group has two phases:
//   L(t): issue this wave's share of the LDS-DMA of step t + 3, then read ALL the step's fragments from LDS into registers;
//   C(t): the step's 4 * NT MFMAs, operands in registers, no memory instruction.
// The groups run half a step apart -- while group 0 is in C(t), group 1 is in L(t), and the other way round -- and one
// workgroup barrier separates the phases: the matrix pipe of a SIMD always has one wave in C while its partner does the
// address arithmetic, the DMA issue and the LDS reads (MI355X_MICROARCH.md, "Two waves per SIMD").  A ring of four
// 32-channel tiles (16 KB of pixels + NT * 4 KB of weights each) keeps three steps of DMA in flight; waits are counted
// (s_waitcnt vmcnt) so that nothing is drained that is not needed in the next phase.
//   * LDS rows are 64 B (32 channels); the bank swizzle sits on the source side as in the kernels above: lane (row r,
//     slot p) of a DMA piece (16 rows x 64 B) fetches chunk p ^ ((r >> 2) & 3); a fragment read of chunk c of row m goes
//     to slot c ^ ((m >> 2) & 3): the 16 lanes of a ds_read_b128 group then cover a 256-byte bank row exactly once.
//   * K order: tap-major, 32-channel chunks inside a tap -- the same fp32 summation order as k_conv3x3_mfma_bf16 up to
//     the split of each 64-channel step into two (sums of the same products; bf16-level agreement, tested).
template <int NT, bool POOL>
__device__ __forceinline__ void conv3x3_pp_body(const ConvArgsBf& a)
{
    constexpr int BM = 256, BN = NT * 64, MT = 2, KB = 32, NB = 4, D = 3;
    constexpr int ROWB = KB * 2;                              // bytes of an LDS row
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int APW = 2;                                    // A pieces (16 rows) per wave and step: 128 rows / 4 waves
    constexpr int BPW = (BN / 16 + 7) / 8;                    // B pieces per wave and step
    constexpr int PW = APW + BPW;
    static_assert(BN % 128 == 0, "every wave stages whole pieces of the weight tile");
    __shared__ __attribute__((aligned(1024))) char smem[NB * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, q = wave & 3, wm = q >> 1, wn = q & 1;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n_tile = bid % a.tiles_n;
    bid /= a.tiles_n;
    const int tile_x = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int tile_y = bid % a.tiles_y;
    const int tile_b = bid / a.tiles_y;
    const int n0 = n_tile * BN;
    const int H = a.H, W = a.W, Cin = a.Cin;
    const int X0 = tile_x << a.lgTW, Y0 = tile_y << a.lgTH, B0 = tile_b * a.TB;

    // loader role: row (lane >> 2) of a 16-row piece, slot (lane & 3); the chunk this lane fetches.
    // Addressing: buffer loads to LDS.  The activation resource starts (W + 1) pixels BEFORE the tensor, so that the tap
    // offset ((ky + 1) W + kx + 1) Cin + c0 is a non-negative scalar (soffset); a lane's own offset carries the same
    // bias.  A lane whose tap falls outside the image gets an out-of-range offset: the buffer load then writes zeros --
    // no zero line, no pointer select; the nine in-image decisions of a lane are nine bits computed once.
    const int lrow = lane >> 2, lchunk = (lane & 3) ^ ((lrow >> 2) & 3);
    const long bias_el = (long)(W + 1) * Cin;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(a.in) - bias_el, 0, (int)(((long)a.B * H * W * Cin + 2 * bias_el) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.wp), 0, (int)((long)a.Cout * 9 * Cin * 2), 0x00020000);
    int aoff[APW];
    unsigned amask[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int row = 128 * g + 32 * q + 16 * i + lrow;
        int xl, yl, bl;
        brick_coords(row, a.lgTW, a.lgTH, xl, yl, bl);
        const int x = X0 + xl, y = Y0 + yl, b = B0 + bl;
        const bool ok = b < a.B && x < W && y < H;
        aoff[i] = (int)(((((long)b * H + y) * W + x) * Cin + 8 * lchunk) * 2);
        unsigned m = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            m |= (ok && yy >= 0 && yy < H && xx >= 0 && xx < W) ? 1u << k : 0u;
        }
        amask[i] = m;
    }
    int boff[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) boff[i] = (int)(((long)(n0 + (wave * BPW + i) * 16 + lrow) * 9 * Cin + 8 * lchunk) * 2);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const int T = 9 * (Cin / KB);
    const int r31 = lane & 31, hh = lane >> 5, fsw = (r31 >> 2) & 3;
    // the next step to stage: tap kp = 3 ky + kx, channel offset c0, the two scalar byte offsets, ring slot
    int kp = 0, kx = 0, c0 = 0, sbuf = 0;
    int so_a = 0, so_b = 0;  // ((ky W + kx) Cin + c0) * 2 (biased: tap (-1,-1) is 0);  (kp Cin + c0) * 2
    auto stage = [&]() {
        char* const sA = smem + sbuf * STAGE;
        char* const sB = sA + A_BYTES;
        static_for<APW>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const int vo = (amask[i] >> kp) & 1 ? aoff[i] : 0x7fffffff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(sA + (128 * g + 32 * q + 16 * i) * ROWB), 16, vo, so_a, 0, 0);
        });
        static_for<BPW>([&](auto I) {
            constexpr int i = decltype(I)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(sB + (wave * BPW + i) * 16 * ROWB), 16, boff[i], so_b, 0, 0);
        });
        c0 += KB;
        so_a += KB * 2;
        so_b += KB * 2;
        if (c0 == Cin) {  // next tap: one pixel to the right, or two pixels short of a row down
            c0 = 0;
            ++kp;
            ++kx;
            so_a += kx == 3 ? (W - 3) * Cin * 2 : 0;
            kx = kx == 3 ? 0 : kx;
        }
        sbuf = sbuf + 1 == NB ? 0 : sbuf + 1;
    };
    // wait until at most `steps` staged steps of this wave are still in flight, and its LDS reads are done; then the barrier
    auto sync = [&](int steps) {
        if (steps >= 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (steps == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * PW) : "memory");
        else if (steps == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    for (int j = 0; j < D; ++j) stage();  // (T >= 18 > D)
    sync(D - 1);                          // step 0 has landed
    if (g == 1) sync(3);                  // group 1 starts half a step later

    bf16x8 fa[MT][KB / 16] = {}, fb[NT][KB / 16] = {};
    int rbuf = 0;
    // one step of this wave's group: L(t) (the step's fragments into registers, then the DMA of step t + D), barrier, C(t),
    // barrier.  MAIN: step t + D exists (the straight-line body of all but the last D steps)
    auto step = [&](const int t, auto main_tag) __attribute__((always_inline)) {
        constexpr bool MAIN = decltype(main_tag)::value;
        {
            const char* const sA = smem + rbuf * STAGE;
            const char* const sB = sA + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < KB / 16; ++ks) {
                const int slot = ((2 * ks + hh) ^ fsw) * 16;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    fa[mt][ks] = *reinterpret_cast<const bf16x8*>(sA + (128 * g + (wm * MT + mt) * 32 + r31) * ROWB + slot);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    fb[nt][ks] = *reinterpret_cast<const bf16x8*>(sB + ((wn * NT + nt) * 32 + r31) * ROWB + slot);
            }
            rbuf = rbuf + 1 == NB ? 0 : rbuf + 1;
        }
        if (MAIN || t + D < T) stage();
        // steps still in flight that the NEXT reader of the ring does not need: t + 2 .. min(t + D, T - 1).  Group 0's
        // L(t + 1) is two phases away (nothing to wait for yet); group 1's L(t) is followed at once by group 0's L(t + 1)
        const int inflight = MAIN ? D - 1 : (t + D < T ? t + D : T - 1) - (t + 1);
        if (g == 0) sync(3);
        else sync(inflight);
#pragma unroll
        for (int ks = 0; ks < KB / 16; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = kWeightsFirst<POOL> ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[nt][ks], fa[mt][ks], acc[mt][nt], 0, 0, 0)
                                                      : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][ks], fb[nt][ks], acc[mt][nt], 0, 0, 0);
        if (g == 0) sync(inflight);       // before group 0's own L(t + 1) (and group 1's C(t))
        else if (MAIN || t + 1 < T) sync(3);  // group 1 has no partner phase after its last C
    };
    int t = 0;
    for (; t < T - D; ++t) step(t, std::true_type{});
    for (; t < T; ++t) step(t, std::false_type{});

    // epilogue (after barrier 2T nobody reads the ring any more: group 1's last L phase lies before it): bias, ReLU, pool,
    // whole lines through this wave's 64 x (NT * 64)-byte share of the ring
    epilogue_lines_bf16<NT, POOL>(acc, (__bf16*)smem + wave * 64 * NT * 32, a, 128 * g + wm * MT * 32, n0 + wn * NT * 32, X0, Y0, B0, lane);
}

// (the body is a __device__ function: buffer-resource builtins in a __global__ template body keep the host pass from
// emitting the kernel's stub)
template <int NT, bool POOL>
__global__ void __launch_bounds__(512) k_conv3x3_pp_bf16(ConvArgsBf a)
{
    conv3x3_pp_body<NT, POOL>(a);
}

#ifdef VA_EXPERIMENTS
// ---------------------------------------------------------------- bf16 conv3x3, two wave groups + halo brick -------
//
// An experiment kept as a tested option (VA_OPT_BF16_VARIANT 6), not a default.  Tap-major staging moves (pixels +
// channels) x 64 B per 32-channel step: 32 KB per step of the 256 x 256 tile, and a CU took in 27..29 B/clk when every
// CU staged and did nothing else (k_conv3x3_pp_bf16 with its reads and MFMAs removed; LDS-DMA and register loads alike).
// k_conv3x3_bpp_bf16 keeps the two-group schedule and removes most of the activation traffic: per 32-channel CHUNK the
// workgroup stages the halo brick of its 256 output pixels ONCE -- TB images x (TH + 2) x (TW + 2) pixels, one 64-byte
// LDS row each, double-buffered -- and runs the nine taps on it by shifting the fragment's row index; only the weights
// (NT * 4 KB per step, ring of four) are staged per step: 18 KB instead of 32 KB per step (16 x 16 bricks).
// Measured (B = 32, us per layer, this kernel / the defaults): 28 x 28: 94 / 85 and 120 / 109 (pooled); 56 x 56: 102 / 95
// and 126 / 117; 112 x 112: 80 / 80 and 128 / 118 -- 43 % fewer staged bytes and no gain: the steps are not bound by
// the bytes taken in but by the serial parts of a phase (fragment reads -> barrier, the phase's skew), DESIGN.md.
//   * K order: chunk-major (32 channels), the nine taps inside -- not the summation order of the tap-major kernels:
//     agreement at the bf16 level (tested), not bit for bit;
//   * waits: the brick of chunk c + 1 is issued at tap 0 of chunk c, BEFORE that step's weight tile; loads complete in
//     issue order, so the counted waits of the weight ring cover it (it must have landed by tap 2; the two steps in
//     between allow NA more operations in flight);
//   * the bank swizzle is keyed on the LDS row as above (chunk c of row r at slot c ^ ((r >> 2) & 3)); a fragment's 32
//     rows are brick-ordered pixels shifted by the tap, not 32 consecutive rows: two of the sixteen lanes of a read group
//     can share a bank position (2-way on those; measured, not modelled).
template <int LGTW, int LGTH, int NT, bool POOL>
__device__ __forceinline__ void conv3x3_bpp_body(const ConvArgsBf& a)
{
    constexpr int TW = 1 << LGTW, TH = 1 << LGTH, BM = 256, TB = BM / (TW * TH), PW = TW + 2, PH = TH + 2;
    static_assert(TW * TH * TB == BM && TB >= 1, "the brick must hold 256 output pixels");
    constexpr int BN = NT * 64, MT = 2, KB = 32, NB = 4, D = 3, ROWB = KB * 2;
    constexpr int NR = TB * PH * PW;                    // halo brick rows (one pixel each)
    constexpr int NA = (NR + 127) / 128;                // brick pieces (16 rows) per wave and chunk
    constexpr int A_ROWS = 8 * NA * 16;                 // rows of one brick buffer (rows >= NR: zero filler)
    constexpr int A_BYTES = A_ROWS * ROWB, B_BYTES = BN * ROWB;
    constexpr int BPW = BN / 128;                       // weight pieces per wave and step
    static_assert(BN % 128 == 0, "every wave stages whole pieces of the weight tile");
    constexpr int RING_BYTES = 2 * A_BYTES + NB * B_BYTES, EPI_BYTES = 8 * 64 * NT * 64;  // (the epilogue's staging tiles reuse the ring)
    __shared__ __attribute__((aligned(1024))) char smem[RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES];
    char* const sAbuf = smem;
    char* const sBbuf = smem + 2 * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave >> 2, q = wave & 3, wm = q >> 1, wn = q & 1;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n_tile = bid % a.tiles_n;
    bid /= a.tiles_n;
    const int tile_x = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int tile_y = bid % a.tiles_y;
    const int tile_b = bid / a.tiles_y;
    const int n0 = n_tile * BN;
    const int H = a.H, W = a.W, Cin = a.Cin;
    const int X0 = tile_x << LGTW, Y0 = tile_y << LGTH, B0 = tile_b * TB;

    // loader role: row (lane >> 2) of a 16-row piece, slot (lane & 3).  Buffer loads to LDS; a brick row outside the image
    // (or beyond the brick) carries an out-of-range offset, for which the load writes zeros
    const int lrow = lane >> 2, lchunk = (lane & 3) ^ ((lrow >> 2) & 3);
    const __amdgpu_buffer_rsrc_t rs_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.in), 0, (int)((long)a.B * H * W * Cin * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.wp), 0, (int)((long)a.Cout * 9 * Cin * 2), 0x00020000);
    int aoff[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int R = (wave * NA + i) * 16 + lrow;
        const int b = R / (PH * PW), rem = R - b * (PH * PW);
        const int yy = rem / PW, xx = rem - yy * PW;
        const int y = Y0 + yy - 1, x = X0 + xx - 1, bb = B0 + b;
        const bool ok = R < NR && bb < a.B && y >= 0 && y < H && x >= 0 && x < W;
        aoff[i] = ok ? (int)(((((long)bb * H + y) * W + x) * Cin + 8 * lchunk) * 2) : 0x7fffffff;
    }
    int boff[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) boff[i] = (int)(((long)(n0 + (wave * BPW + i) * 16 + lrow) * 9 * Cin + 8 * lchunk) * 2);

    // fragment role: brick row of this lane's pixel at tap (0, 0)
    const int r31 = lane & 31, hh = lane >> 5, fswb = (r31 >> 2) & 3;
    int idx0[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int xl, yl, bl;
        brick_coords(128 * g + (wm * MT + mt) * 32 + r31, LGTW, LGTH, xl, yl, bl);
        idx0[mt] = (bl * PH + yl + 1) * PW + xl + 1;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const int nchunks = Cin / KB, T = 9 * nchunks;
    // the next weight tile to stage: tap, chunk, ring slot
    int stap = 0, schunk = 0, sbuf = 0;
    auto stage_b = [&]() {
        char* const sB = sBbuf + sbuf * B_BYTES;
        const int so = (stap * Cin + schunk * KB) * 2;
        static_for<BPW>([&](auto I) {
            constexpr int i = decltype(I)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(sB + (wave * BPW + i) * 16 * ROWB), 16, boff[i], so, 0, 0);
        });
        ++stap;
        if (stap == 9) {
            stap = 0;
            ++schunk;
        }
        sbuf = sbuf + 1 == NB ? 0 : sbuf + 1;
    };
    auto stage_a = [&](int chunk) {
        char* const sA = sAbuf + (chunk & 1) * A_BYTES;
        static_for<NA>([&](auto I) {
            constexpr int i = decltype(I)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(sA + (wave * NA + i) * 16 * ROWB), 16, aoff[i], chunk * KB * 2, 0, 0);
        });
    };
    // wait until at most `ops` of this wave's loads are still in flight, and its LDS reads are done; then the barrier
    auto sync = [&](int ops) {
        if (ops < 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (ops == 2 * BPW + NA) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * BPW + NA) : "memory");
        else if (ops == 2 * BPW) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * BPW) : "memory");
        else if (ops == BPW) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(BPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    stage_a(0);
    for (int j = 0; j < D; ++j) stage_b();  // (T >= 18 > D)
    sync(2 * BPW);                          // the first brick and step 0's weights have landed
    if (g == 1) sync(-1);                   // group 1 starts half a step later

    bf16x8 fa[MT][KB / 16] = {}, fb[NT][KB / 16] = {};
    int rbuf = 0, tap = 0, chunk = 0;
    auto step = [&](const int t, auto main_tag) __attribute__((always_inline)) {
        constexpr bool MAIN = decltype(main_tag)::value;
        {
            const char* const sA = sAbuf + (chunk & 1) * A_BYTES;
            const char* const sB = sBbuf + rbuf * B_BYTES;
            const int ky = tap / 3, toff = (ky - 1) * PW + (tap - 3 * ky) - 1;
            int arow[MT], asw[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                arow[mt] = idx0[mt] + toff;
                asw[mt] = (arow[mt] >> 2) & 3;
            }
#pragma unroll
            for (int ks = 0; ks < KB / 16; ++ks) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    fa[mt][ks] = *reinterpret_cast<const bf16x8*>(sA + arow[mt] * ROWB + ((2 * ks + hh) ^ asw[mt]) * 16);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    fb[nt][ks] = *reinterpret_cast<const bf16x8*>(sB + ((wn * NT + nt) * 32 + r31) * ROWB + ((2 * ks + hh) ^ fswb) * 16);
            }
            rbuf = rbuf + 1 == NB ? 0 : rbuf + 1;
        }
        const bool brick = tap == 0 && chunk + 1 < nchunks;
        if (brick) stage_a(chunk + 1);  // (before this step's weight tile: loads complete in issue order)
        if (MAIN || t + D < T) stage_b();
        // loads that may stay in flight at the barrier before the ring's next reader: the weight tiles t + 2 .. min(t + D,
        // T - 1), plus the brick while it is younger than the tile the reader needs (taps 0 and 1)
        const int steps = MAIN ? D - 1 : (t + D < T ? D - 1 : T - 2 - t < 0 ? 0 : T - 2 - t);
        const int inflight = steps * BPW + ((tap < 2 && chunk + 1 < nchunks) ? NA : 0);
        if (g == 0) sync(-1);
        else sync(inflight);
#pragma unroll
        for (int ks = 0; ks < KB / 16; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = kWeightsFirst<POOL> ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[nt][ks], fa[mt][ks], acc[mt][nt], 0, 0, 0)
                                                      : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt][ks], fb[nt][ks], acc[mt][nt], 0, 0, 0);
        if (g == 0) sync(inflight);
        else if (MAIN || t + 1 < T) sync(-1);
        ++tap;
        if (tap == 9) {
            tap = 0;
            ++chunk;
        }
    };
    int t = 0;
    for (; t < T - D; ++t) step(t, std::true_type{});
    for (; t < T; ++t) step(t, std::false_type{});

    // epilogue (after barrier 2T nobody reads the LDS tiles any more): whole lines through this wave's share of them
    epilogue_lines_bf16<NT, POOL>(acc, (__bf16*)smem + wave * 64 * NT * 32, a, 128 * g + wm * MT * 32, n0 + wn * NT * 32, X0, Y0, B0, lane);
}

template <int LGTW, int LGTH, int NT, bool POOL>
__global__ void __launch_bounds__(512) k_conv3x3_bpp_bf16(ConvArgsBf a)
{
    conv3x3_bpp_body<LGTW, LGTH, NT, POOL>(a);
}
#endif  // VA_EXPERIMENTS

// ---------------------------------------------------------------- bf16 conv3x3 with 64 input channels: weights resident ----
//
// conv1_2 (64 -> 64 at 224 x 224) and conv2_1 (64 -> 128 at 112 x 112) have K = 576: nine tap-major steps of the kernels
// above, each staging (128 pixels + 64 or 128 channels) x 128 B for 8 or 16 MFMAs per wave -- 2.7 GB through the CUs'
// vector-memory path for conv1_2 at batch 32, which is what its 162 us are (the measured 27-29 B/clk per CU).
// k_conv3x3_ws_bf16 stages the WEIGHTS once per workgroup -- 64 output channels x 576 x 2 B = 72 KB stay in LDS -- and
// the workgroup walks over 16 x 16-pixel bricks: per brick it stages the 18 x 18 halo patch (41 KB, double-buffered: the
// next brick's patch lands during this brick's 576 MFMAs) and forms the nine taps by shifting the fragment's row index.
// No barrier inside a brick: four waves, one per SIMD, each 64 pixels x 64 channels, run their 144 MFMAs on operands that
// are all in LDS; two barriers per brick (patch swap; the epilogue's staging tiles reuse the patch just consumed).
//   * 72 B staged per MFMA instead of 768;
//   * one 64-channel chunk, so chunk-major = tap-major: the same products in the same order per accumulator as
//     k_conv3x3_mfma_bf16 -- bit-equal results (tested);
//   * weight rows are padded to 1168 B (conflict-free 16-lane fragment reads); patch rows are 128 B with a source-side
//     swizzle keyed on the patch COORDINATES: chunk c of patch pixel (yy, xx) sits at slot c ^ (((xx >> 1) + 4 yy) & 7).
//     A fragment's 16-lane read group holds the pixels x in {0,1,6,7,10,11,12,13} + const, y in {0,1} + const of the
//     brick order; their bank positions 2 slot + (xx & 1) = (xx + 8 yy) mod 16 are then all different, for every tap
//     (keyed on the row index (R >> 1) & 7, two of the sixteen collided: a third of the LDS cycles were conflict cycles);
//   * workgroup w serves channel half w % (Cout / 64) of the bricks w / NH, w / NH + G / NH, ...: G = the CU count
//     (158 KB of LDS: one workgroup per CU).
struct ConvArgsWs {
    ConvArgsBf c;      // lgTW = lgTH = 4, TB = 1, tiles_x = W / 16, tiles_y = H / 16; Cin = 64
    int n_bricks;      // B * tiles_y * tiles_x
};

template <bool POOL>
__device__ __forceinline__ void conv3x3_ws_body(const ConvArgsWs& aw)
{
    const ConvArgsBf& a = aw.c;
    constexpr int PW = 18, NR = 18 * 18, A_ROWS = 328, A_BYTES = A_ROWS * 128;  // 41 pieces of 8 rows
    constexpr int WSTRIDE = 1168, W_BYTES = 64 * WSTRIDE;
    constexpr int NPIECE = A_ROWS / 8, PPW = (NPIECE + 3) / 4;                  // pieces per wave (11: the last wave has 8)
    __shared__ __attribute__((aligned(1024))) char smem[2 * A_BYTES + W_BYTES];
    char* const sA = smem;
    char* const sW = smem + 2 * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    const int NH = a.Cout / 64, nh = blockIdx.x % NH, nb = nh * 64;
    const int first = blockIdx.x / NH, stride = gridDim.x / NH;

    // the weights of this channel half: [64][576] bf16 -> LDS rows of WSTRIDE bytes (once)
    {
        const uint4* src = (const uint4*)(a.wp + (size_t)nb * 576);
        for (int i = tid; i < 64 * 72; i += 256) {
            const int n = i / 72, c = i - n * 72;
            *(uint4*)(sW + n * WSTRIDE + 16 * c) = src[i];
        }
    }

    // loader role: row (lane >> 3) of an 8-row piece, slot (lane & 7); per piece the patch pixel (yy, xx) and its offset
    // relative to the brick's first pixel
    const __amdgpu_buffer_rsrc_t rs_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.in), 0, (int)((long)a.B * H * W * 64 * 2), 0x00020000);
    const int lrow = lane >> 3, lslot = lane & 7;
    int rel[PPW], pyx[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int R = (wave * PPW + i) * 8 + lrow;
        const int yy = R / PW, xx = R - yy * PW;
        rel[i] = (((yy - 1) * W + (xx - 1)) * 64 + 8 * (lslot ^ (((xx >> 1) + 4 * yy) & 7))) * 2;
        pyx[i] = R < NR ? (yy << 8) | xx : -1;
    }
    auto stage_a = [&](int brick, int buf) {
        const int tx = brick % a.tiles_x, r1 = brick / a.tiles_x, ty = r1 % a.tiles_y, b = r1 / a.tiles_y;
        const int X0 = tx << 4, Y0 = ty << 4;
        const int base = (((b * H + Y0) * W + X0) * 64) * 2;
        static_for<PPW>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if ((wave * PPW + i) < NPIECE) {  // (wave-uniform)
                const int y = Y0 + (pyx[i] >> 8) - 1, x = X0 + (pyx[i] & 255) - 1;
                const bool ok = pyx[i] >= 0 && y >= 0 && y < H && x >= 0 && x < W;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(sA + buf * A_BYTES + (wave * PPW + i) * 1024), 16,
                                                         ok ? base + rel[i] : 0x7fffffff, 0, 0, 0);
            }
        });
    };

    // fragment role
    const int r31 = lane & 31, hh = lane >> 5;
    int idx0[2], px0[2], py0[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int xl, yl, bl;
        brick_coords(64 * wave + 32 * mt + r31, 4, 4, xl, yl, bl);
        px0[mt] = xl + 1;
        py0[mt] = yl + 1;
        idx0[mt] = py0[mt] * PW + px0[mt];
    }
    const char* const wfrag = sW + r31 * WSTRIDE + hh * 16;

    if (first < aw.n_bricks) stage_a(first, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    int buf = 0;
    for (int brick = first; brick < aw.n_bricks; brick += stride) {
        if (brick + stride < aw.n_bricks) stage_a(brick + stride, buf ^ 1);
        f32x16 acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;
        const char* const pA = sA + buf * A_BYTES;
        // 36 k-steps (9 taps x 4 blocks of 16 channels), the fragments of step i + 2 fetched before the MFMAs of step i
        // are issued (one wave per SIMD: nothing else hides the LDS latency; one step ahead measured 134 us, none 148)
        bf16x8 fa[3][2], fb[3][2];
        auto fetch = [&](int i, int slot) __attribute__((always_inline)) {
            const int tap = i >> 2, ks = i & 3;
            const int toff = (tap / 3 - 1) * PW + (tap % 3) - 1;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = idx0[mt] + toff;
                const int sw = (((px0[mt] + (tap % 3) - 1) >> 1) + 4 * (py0[mt] + tap / 3 - 1)) & 7;
                fa[slot][mt] = *reinterpret_cast<const bf16x8*>(pA + row * 128 + (((2 * ks + hh) ^ sw) << 4));
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                fb[slot][nt] = *reinterpret_cast<const bf16x8*>(wfrag + nt * 32 * WSTRIDE + (tap * 64 + ks * 16) * 2);
        };
        fetch(0, 0);
        fetch(1, 1);
#pragma unroll
        for (int i = 0; i < 36; ++i) {
            if (i + 2 < 36) fetch(i + 2, (i + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = kWeightsFirst<POOL> ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[i % 3][nt], fa[i % 3][mt], acc[mt][nt], 0, 0, 0)
                                                      : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i % 3][mt], fb[i % 3][nt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave has read this brick's patch: it becomes the staging tiles
        {
            const int tx = brick % a.tiles_x, r1 = brick / a.tiles_x, ty = r1 % a.tiles_y, b = r1 / a.tiles_y;
            epilogue_lines_bf16<2, POOL>(acc, (__bf16*)(sA + buf * A_BYTES) + wave * 64 * 64, a, 64 * wave, nb, tx << 4, ty << 4, b, lane);
        }
        // the next patch has landed (and this wave's staging reads are done) before anyone stages into this buffer again
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf ^= 1;
    }
}

template <bool POOL>
__global__ void __launch_bounds__(256) k_conv3x3_ws_bf16(ConvArgsWs a)
{
    conv3x3_ws_body<POOL>(a);
}

// ---------------------------------------------------------------- bf16 conv3x3 on 14 x 14 images: one image per workgroup ----
//
// conv5_1 .. conv5_3 (512 -> 512 channels at 14 x 14): M = 196 pixels per image, K = 9 x 512.  The tap-major kernel above
// runs them as 392 workgroups of 128 pixels x 64 channels that stage, per 64-channel K step, a fresh copy of the activations
// for every one of the nine taps and every one of the eight channel tiles: 462 MB of activation staging + 231 MB of weights
// per layer at batch 32, which at the measured 27-29 B/clk a CU takes in is its 51-54 us (23 % of the matrix pipe).
// k_conv3x3_img14 gives every workgroup ONE WHOLE IMAGE and 64 output channels (batch 32: 32 x 8 = 256 workgroups, one
// per CU, one round):
//   * the image's 64-channel chunk is staged ONCE as a zero-bordered 16 x 16 halo brick (32 KB, double-buffered); the nine
//     taps are formed by shifting the brick pixel a fragment row reads -- 1/9 of the activation staging, and 1/8 of it
//     again because the brick serves all nine taps of the 64 output channels: 0.26 MB of activations + 0.59 MB of weights
//     per workgroup instead of 2.7 MB;
//   * weights stream through a ring of three 24 KB tiles (64 channels x one kernel ROW of three taps x 64 input channels),
//     two steps ahead, counted s_waitcnt vmcnt; one barrier per step of 48 MFMAs per wave (24 steps per layer);
//   * eight waves: four COMPUTING waves (one per SIMD: fragment reads one sub-step ahead + MFMAs, nothing else) and four
//     LOADER waves (one per SIMD) that issue all LDS-DMA pieces, wait for them and certify them at the step's barrier;
//   * the rows of the GEMM are the image's pixels in quad order: row block qy (32 rows) = the seven 2 x 2 quads of quad row qy
//     (row 4 qx + 2 dy + dx = pixel (2 qy + dy, 2 qx + dx)) + one dummy quad, so that the four registers of a lane's
//     accumulator quad are one 2 x 2 pooling window AND the 16-lane groups of a fragment read (four quads: slots {0,3,5,6} or
//     {1,2,4,7} of the block) hit 16 different bank positions under the brick's swizzle; waves 0..2 own two row blocks,
//     wave 3 one (+ a dummy block of zero rows); both 32-channel column blocks;
//   * K order is chunk-major (64-channel chunk outside, tap inside): another fp32 summation order than the tap-major
//     kernels, so its results agree with theirs at the bf16 noise level, not bit for bit (like the first-layer paths).
// POOLF32 = false: bf16 NHWC output (conv5_1, conv5_2); true: bias + ReLU + 2 x 2 max-pool, fp32 NHWC [B][7][7][Cout] (conv5_3).
// TIMING BUILDS ONLY (wrong results): parts of k_conv3x3_img14's steps to leave out -- 1 = the MFMAs, 2 = the fragment
// reads, 4 = the LDS-DMA refills inside the loop
#ifndef VA_I14_SKIP
#define VA_I14_SKIP 0
#endif
struct Img14Args {
    const void* in;     // NHWC [B][14][14][Cin] of T, Cin % (128 / sizeof(T)) == 0
    const void* wp;     // [Cout][9][Cin] of T
    const float* bias;  // [Cout]
    void* out;          // NHWC [B][14][14][Cout] of T, or pooled f32 [B][7][7][Cout] when POOL
    int B, Cin, Cout;
};

// T = __bf16 (chunk = 64 channels, v_mfma_f32_32x32x16_bf16) or float (chunk = 32 channels, v_mfma_f32_32x32x2_f32: the fp32
// parity path, round 3): a pixel's chunk is one 128-byte line either way, so bricks, weight tiles, DMA pieces, swizzles and
// fragment addresses are the same; a 16-byte fragment is 8 bf16 for one MFMA or 4 floats for four.
template <typename T, bool POOL>
__device__ __forceinline__ void conv3x3_img14_body(const Img14Args& a)
{
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
    constexpr int BK = 8 * EPC;               // channels per chunk: one 128-byte line per pixel
    // one step = one kernel ROW (three taps) of one chunk: 48 (bf16) / 192 (fp32) MFMAs per computing wave between two barriers
    constexpr int BRICK = 256 * BK, WT = 3 * 64 * BK, NWR = 3;  // elements: halo brick chunk, weight tile (3 taps); ring depth
    constexpr int WPIECES = 6, BPIECES = 8;                      // a loader wave's 1 KB DMA pieces per weight tile / per brick chunk
    __shared__ __attribute__((aligned(1024))) T smem[2 * BRICK + NWR * WT];
    T* const sBrick = smem;
    T* const sW = smem + 2 * BRICK;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..3: computing waves, 4..7: loader waves (one of each per SIMD)
    const int wave = wave8 & 3;
    const int tiles_n = a.Cout / 64;
    // Workgroup id -> (channel tile, image): consecutive ids go round-robin over the 8 XCDs, so with id % tiles_n as the
    // channel tile (8 tiles at 512 channels) every XCD works on ONE 64-channel slice of the weights (0.59 / 1.18 MB: resident in
    // its L2, read by the XCD's 32 workgroups) -- the weight tile a step waits for then comes from L2, while the image
    // bricks, which are fetched a whole chunk ahead, may come from further away.
    const int n_tile = blockIdx.x % tiles_n, b = blockIdx.x / tiles_n;
    const int n0 = n_tile * 64, Cin = a.Cin;
    const int r31 = lane & 31, hh = lane >> 5, lrow = lane >> 3, lslot = lane & 7;
    const int nchunks = Cin / BK, NT_STEPS = 3 * nchunks;
    // Brick pixel bp = 16 by + bx holds image pixel (by - 1, bx - 1), zeros on the border; its 16-byte chunk c sits at slot
    // c ^ swz(bp) of its 128-byte line (swizzle on the source side, as above).  A fragment read's 16-lane groups hold four
    // 2 x 2 quads: the pixels of a quad's two rows are 16 brick pixels = 2 KB apart, i.e. on the same banks, so the swizzle
    // takes the row's parity as well: swz = ((bx >> 1) & 3) | ((by & 1) << 2).
    auto brick_swz = [](int bp) { return ((bp >> 1) & 3) | (((bp >> 4) & 1) << 2); };

    if (wave8 >= 4) {
        // ------------------------------------------------------------------ loader waves ----
        // An LDS-DMA piece costs the issuing wave 100-185 cycles of its instruction stream: issued by the computing waves
        // (6-14 pieces per step in front of 48 MFMAs) they cost as much as the MFMAs themselves (46 us per layer, MFMA
        // pipe 33 % busy).  Here they belong to a wave of their own on every SIMD, which does nothing else: it refills the
        // ring behind the barrier, waits for its pieces of the NEXT step to land, and meets the computing waves at the
        // next barrier -- its arrival is what tells them the tile is complete.
        const T* const in = (const T*)a.in;
        const T* const wp = (const T*)a.wp;
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(in) + (size_t)b * 196 * Cin, 0, 196 * Cin * (int)sizeof(T), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(wp) + (size_t)n0 * 9 * Cin, 0, 64 * 9 * Cin * (int)sizeof(T), 0x00020000);
        int aoff[BPIECES];  // brick pixels (8 wave + i) * 8 + lrow
#pragma unroll
        for (int i = 0; i < BPIECES; ++i) {
            const int bp = ((wave * 8 + i) << 3) + lrow, by = bp >> 4, bx = bp & 15;
            const bool ok = by >= 1 && by <= 14 && bx >= 1 && bx <= 14;
            aoff[i] = ok ? (((by - 1) * 14 + (bx - 1)) * Cin + EPC * (lslot ^ brick_swz(bp))) * (int)sizeof(T) : 0x7fffffff;  // out of range: zeros
        }
        int boff[2];  // output channels (2 wave + i) * 8 + lrow of a tap's 64 x BK sub-tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (wave * 2 + i) * 8 + lrow;
            boff[i] = (row * 9 * Cin + EPC * (lslot ^ ((row >> 1) & 7))) * (int)sizeof(T);
        }
        auto stage_brick = [&](int chunk) {
            T* const dst = sBrick + (chunk & 1) * BRICK;
            static_for<BPIECES>([&](auto I) {
                constexpr int i = decltype(I)::value;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(dst + (wave * 8 + i) * 8 * BK), 16, aoff[i], chunk * BK * (int)sizeof(T), 0, 0);
            });
        };
        auto stage_w = [&](int t) {  // step t = 3 chunk + ky: weights [n][3 ky + kx][chunk * BK ..], kx = 0, 1, 2
            const int chunk = t / 3, ky = t - 3 * chunk;
            T* const dst = sW + (t % NWR) * WT;
            static_for<WPIECES>([&](auto I) {
                constexpr int i = decltype(I)::value, kx = i >> 1, h = i & 1;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(dst + kx * 64 * BK + (wave * 2 + h) * 8 * BK), 16, boff[h],
                                                         ((3 * ky + kx) * Cin + chunk * BK) * (int)sizeof(T), 0, 0);
            });
        };
        // prologue: brick 0, weight tiles of steps 0 and 1
        stage_brick(0);
        stage_w(0);
        if (NT_STEPS > 1) stage_w(1);
        bool brick_prev = false;  // a brick was issued in the previous step (it sits in the queue BEFORE that step's weight tile)
        for (int t = 0; t < NT_STEPS; ++t) {
            const int chunk = t / 3, ky = t - 3 * chunk;
            // this wave's pieces of step t must have landed; younger ones may fly: the weight tile of step t + 1 (6 pieces) and,
            // when a brick was issued in the step before (after tile t, before tile t + 1), its 8 pieces
            // (queue at this point, oldest first: tile t | [brick, 8 pieces] | tile t + 1, 6 pieces)
            if (t + 1 >= NT_STEPS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (brick_prev) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // step t is complete in LDS; the computing waves have read step t - 1
            // refills: the NEXT chunk's brick at the first row of this one (its buffer was last read in step t - 1), then the
            // weight tile of step t + 2 (its slot held step t - 1)
            brick_prev = ky == 0 && chunk + 1 < nchunks && !(VA_I14_SKIP & 4);
            if (brick_prev) stage_brick(chunk + 1);
            if (t + 2 < NT_STEPS && !(VA_I14_SKIP & 4)) stage_w(t + 2);
        }
        return;
    }

    // ---------------------------------------------------------------------- computing waves ----
    // fragment rows: row blocks wave and wave + 4 (wave 3's second block is the dummy block 7: all rows read the brick's
    // zero corner; its MFMAs cost nothing the other waves do not spend anyway)
    constexpr int MT = 2, NT = 2;
    typedef typename std::conditional<F32, f32x4, bf16x8>::type frag_t;
    const int nrb = wave < 3 ? 2 : 1;
    int bp0[MT];  // centre-tap brick pixel of this lane's row in each block (0 = rows beyond the image: the zero corner pixel)
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int qy = wave + 4 * i, qx = r31 >> 2;  // row block = quad row of the image; 8 quad slots, the last one a dummy
        const int y = 2 * qy + ((r31 >> 1) & 1), x = 2 * qx + (r31 & 1);
        bp0[i] = (qx < 7 && qy < 7) ? (y + 1) * 16 + (x + 1) : 0;
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    const int fsw = (r31 >> 1) & 7;
    // pixel fragment first (rows of the accumulator = pixels: a register quad is a pooling window) when pooling; weight
    // fragment first (rows = channels: four consecutive channels of one pixel per register quad) otherwise
    auto mma = [&](const frag_t& fpix, const frag_t& fwgt, f32x16& c) __attribute__((always_inline)) {
        if constexpr (F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                c = POOL ? __builtin_amdgcn_mfma_f32_32x32x2f32(fpix[e], fwgt[e], c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(fwgt[e], fpix[e], c, 0, 0, 0);
        } else {
            c = POOL ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fpix, fwgt, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fwgt, fpix, c, 0, 0, 0);
        }
    };

    for (int t = 0; t < NT_STEPS; ++t) {
        const int chunk = t / 3, ky = t - 3 * chunk;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the loader waves have seen step t land; every computing wave has read step t - 1
        const T* const sA = sBrick + (chunk & 1) * BRICK;
        const T* const sB = sW + (t % NWR) * WT;
        // Six sub-steps (3 taps x 2 halves of the chunk) of 8 fragment pairs, branch-free and fully unrolled; the 8 fragment reads
        // of sub-step s + 1 are ISSUED BEFORE the MFMAs of sub-step s (sched_barrier pins that order) and waited for only at
        // the start of s + 1: with one computing wave per SIMD nothing else covers the LDS latency.
        int arow[3][MT], asw[3][MT];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int bp = bp0[i] ? bp0[i] + (ky - 1) * 16 + (kx - 1) : 0;
                arow[kx][i] = bp * BK;
                asw[kx][i] = brick_swz(bp);
            }
        frag_t fa[2][2][MT], fb[2][2][NT];  // [buffer][half of the sub-step][block]
        auto fetch = [&](auto SS, auto BUF) {
            constexpr int ss = decltype(SS)::value, kx = ss >> 1, bf = decltype(BUF)::value;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ks = 2 * (ss & 1) + h;
#pragma unroll
                for (int i = 0; i < MT; ++i) fa[bf][h][i] = *reinterpret_cast<const frag_t*>(&sA[arow[kx][i] + (((2 * ks + hh) ^ asw[kx][i]) * EPC)]);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    fb[bf][h][j] = *reinterpret_cast<const frag_t*>(&sB[kx * 64 * BK + (j * 32 + r31) * BK + (((2 * ks + hh) ^ fsw) * EPC)]);
            }
        };
        fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        static_for<6>([&](auto SS) {
            constexpr int ss = decltype(SS)::value, cur = ss & 1;
            if constexpr (ss + 1 < 6 && !(VA_I14_SKIP & 2)) fetch(std::integral_constant<int, ss + 1>{}, std::integral_constant<int, cur ^ 1>{});
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(VA_I14_SKIP & 1))
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) mma(fa[cur][h][i], fb[cur][h][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    // ---- epilogue
    if constexpr (POOL) {
        // pixel fragment first: register 4 g4 + j of lane (r31, hh) is row 8 g4 + 4 hh + j of the block, channel r31 of the
        // column block: the four registers are one quad = one pooling window
        float* const out = (float*)a.out + (size_t)b * 49 * a.Cout;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (i >= nrb) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n0 + 32 * j + r31;
                const float bias = a.bias[n];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int qy = wave + 4 * i, qx = 2 * g4 + hh;
                    if (qx >= 7 || qy >= 7) continue;
                    const int quad = 7 * qy + qx;
                    const float m01 = fmaxf(acc[i][j][4 * g4 + 0], acc[i][j][4 * g4 + 1]), m23 = fmaxf(acc[i][j][4 * g4 + 2], acc[i][j][4 * g4 + 3]);
                    out[(size_t)quad * a.Cout + n] = fmaxf(fmaxf(m01, m23) + bias, 0.0f);  // max(relu(v + b)) = relu(max(v) + b)
                }
            }
        }
    } else {
        // weight fragment first: register 4 g4 + j of lane (r31, hh) is channel 8 g4 + 4 hh + j of the column block, row r31
        // of the row block: four consecutive channels of one pixel = one 8-/16-byte LDS write into a wave-private 32-row x 64-channel
        // tile, read back as whole pixel lines, 16 bytes per lane (direct 8-byte stores cost 4.6 us per layer).
        // The tile lives in brick buffer 0, whose last reader (step T - 4) every wave has left behind at barrier T - 1;
        // 16-byte chunk c of row r sits at slot c ^ (r & (chunks per row - 1)).
        constexpr int NCHK = 64 / EPC;  // 16-byte chunks per 64-channel row: 8 (bf16) / 16 (fp32)
        T* const stage = sBrick + wave * 32 * 64;  // 4 / 8 KB per wave
        T* const out = (T*)a.out + (size_t)b * 196 * a.Cout + n0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (i >= nrb) continue;  // (wave-uniform)
            const int qy = wave + 4 * i;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int n = 32 * j + 8 * g4 + 4 * hh;
                    const float4 bs = *(const float4*)(a.bias + n0 + n);
                    const float v0 = fmaxf(acc[i][j][4 * g4 + 0] + bs.x, 0.0f), v1 = fmaxf(acc[i][j][4 * g4 + 1] + bs.y, 0.0f);
                    const float v2 = fmaxf(acc[i][j][4 * g4 + 2] + bs.z, 0.0f), v3 = fmaxf(acc[i][j][4 * g4 + 3] + bs.w, 0.0f);
                    T* const dst = stage + r31 * 64 + (((n / EPC) ^ (r31 & (NCHK - 1))) * EPC) + (n % EPC);
                    if constexpr (F32) {
                        *(float4*)dst = make_float4(v0, v1, v2, v3);
                    } else {
                        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                        bf16x4 v;
                        v.x = (__bf16)v0, v.y = (__bf16)v1, v.z = (__bf16)v2, v.w = (__bf16)v3;
                        *(bf16x4*)dst = v;
                    }
                }
            __builtin_amdgcn_wave_barrier();  // (wave-private tile: the wave's LDS writes are ordered before its reads)
#pragma unroll
            for (int it = 0; it < 32 * NCHK / 64; ++it) {  // 32 rows x NCHK chunks of 16 bytes
                const int ch = it * 64 + lane, r = ch / NCHK, c8 = ch % NCHK, qx = r >> 2;
                const uint4 v = *(const uint4*)(stage + r * 64 + ((c8 ^ (r & (NCHK - 1))) * EPC));
                const int y = 2 * qy + ((r >> 1) & 1), x = 2 * qx + (r & 1);
                if (qx < 7) *(uint4*)(out + (size_t)(y * 14 + x) * a.Cout + EPC * c8) = v;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <typename T, bool POOL>
__global__ void __launch_bounds__(512) k_conv3x3_img14(Img14Args a)
{
    conv3x3_img14_body<T, POOL>(a);
}

// ---------------------------------------------------------------- fp32 conv3x3, LDS-DMA staging ---------
//
// The bf16 kernel's structure (LDS-DMA with source-side swizzle, one 32 KB buffer, four workgroups per CU, or
// the DMA ring for small grids) with fp32 data: K step = 32 channels (a 128-byte line per pixel), 16-byte
// fragment reads feed four v_mfma_f32_32x32x2_f32 each (lane half hh holds k = 8*ks + 4*hh + j for both
// operands).  Needs Cin % 32 == 0.
constexpr int kDmaBK = 32;

template <int NT, bool POOL, int NBUF>
__global__ void __launch_bounds__(256, NBUF == 1 ? 4 : 2) k_conv3x3_dma_f32(ConvArgs a)
{
    constexpr int BM = 128, BN = NT * 64, MT = 2;
    constexpr int A_INSTR = BM / 32, B_INSTR = BN / 32;  // LDS-DMA instructions per wave and K step (8 rows each)
    constexpr int TILE = (BM + BN) * kDmaBK;  // bf16 elements of one K step's A and B tiles
    __shared__ __attribute__((aligned(1024))) float smem[NBUF * TILE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int n_tile = bid % a.tiles_n;
    bid /= a.tiles_n;
    const int tile_x = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int tile_y = bid % a.tiles_y;
    const int tile_b = bid / a.tiles_y;
    const int n0 = n_tile * BN;
    const int H = a.H, W = a.W, Cin = a.Cin;
    const int X0 = tile_x << a.lgTW, Y0 = tile_y << a.lgTH, B0 = tile_b * a.TB;

    // loader role: row (lane >> 3) of the 8-row group, slot (lane & 7)
    const int lrow = lane >> 3, lslot = lane & 7;
    int ax[A_INSTR], ay[A_INSTR];
    long apix[A_INSTR];
    bool aok[A_INSTR];
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
        const int row = (wave * A_INSTR + i) * 8 + lrow;
        const int c = lslot ^ ((row >> 1) & 7);
        int xl, yl, bl;
        brick_coords(row, a.lgTW, a.lgTH, xl, yl, bl);
        ax[i] = X0 + xl;
        ay[i] = Y0 + yl;
        const int b = B0 + bl;
        aok[i] = b < a.B && ax[i] < W && ay[i] < H;
        apix[i] = (((long)b * H + ay[i]) * W + ax[i]) * Cin + 4 * c;
    }
    const float* wrow[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int row = (wave * B_INSTR + i) * 8 + lrow;
        wrow[i] = a.wp + (size_t)(n0 + row) * 9 * Cin + 4 * (lslot ^ ((row >> 1) & 7));
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    const int cchunks = Cin / kDmaBK;
    const int T = 9 * cchunks;
    const int r31 = lane & 31, hh = lane >> 5;
    const int fsw = (r31 >> 1) & 7;  // fragment rows are (multiple of 32) + r31
    int kp = 0, c0 = 0;
    // LDS-DMA of K step (kp, c0) into ring slot `buf`; advances (kp, c0)
    auto stage = [&](int buf) {
        float* const sA = smem + buf * TILE;
        float* const sB = sA + BM * kDmaBK;
        const int ky = kp / 3 - 1, kx = kp % 3 - 1;
        const long tap = ((long)ky * W + kx) * Cin + c0;
        static_for<A_INSTR>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const int yy = ay[i] + ky, xx = ax[i] + kx;
            const bool ok = aok[i] && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const float* src = ok ? a.in + apix[i] + tap : a.zeros;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sA + (wave * A_INSTR + i) * 8 * kDmaBK), 16, 0, 0);
        });
        static_for<B_INSTR>([&](auto I) {
            constexpr int i = decltype(I)::value;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wrow[i] + (size_t)kp * Cin + c0),
                                             (lds_ptr_t)(sB + (wave * B_INSTR + i) * 8 * kDmaBK), 16, 0, 0);
        });
        c0 += kDmaBK;
        if (c0 == Cin) {
            c0 = 0;
            ++kp;
        }
    };
    auto compute = [&](int buf) {
        const float* const sA = smem + buf * TILE;
        const float* const sB = sA + BM * kDmaBK;
#pragma unroll
        for (int ks = 0; ks < kDmaBK / 8; ++ks) {
            f32x4 fa[MT], fb[NT];
            const int slot = ((2 * ks + hh) ^ fsw) * 4;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                fa[mt] = *reinterpret_cast<const f32x4*>(&sA[((wm * MT + mt) * 32 + r31) * kDmaBK + slot]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                fb[nt] = *reinterpret_cast<const f32x4*>(&sB[((wn * NT + nt) * 32 + r31) * kDmaBK + slot]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mt][j], fb[nt][j], acc[mt][nt], 0, 0, 0);
        }
    };
    if constexpr (NBUF == 1) {
        for (int t = 0; t < T; ++t) {
            stage(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();  // every wave has read the tile before the next DMA overwrites it
        }
    } else {
        constexpr int LOADS = A_INSTR + B_INSTR;  // this wave's DMA instructions per K step
        // prologue: NBUF-1 steps in flight (T >= 3 >= NBUF-1 is not guaranteed for NBUF > 4: launch only NBUF <= 4)
        for (int j = 0; j < NBUF - 1 && j < T; ++j) stage(j);
        int slot = 0;
        for (int t = 0; t < T; ++t) {
            // wait until this wave's DMAs of step t have landed: steps t+1 .. t+NBUF-2 may stay in flight
            if (t + NBUF - 2 < T) {
                if constexpr ((NBUF - 2) * LOADS == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if constexpr ((NBUF - 2) * LOADS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if constexpr ((NBUF - 2) * LOADS == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if constexpr ((NBUF - 2) * LOADS == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // all waves: step t's tile is complete, and step t-1's reads are done, so its slot can be refilled
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + NBUF - 1 < T) stage(slot == 0 ? NBUF - 1 : slot - 1);
            compute(slot);
            slot = slot + 1 == NBUF ? 0 : slot + 1;
        }
    }

    // epilogue: bias + ReLU (+ 2x2 max-pool over the 4 registers reg&3 of a lane)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + (wn * NT + nt) * 32 + r31;
        const float bias = a.bias[n];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                // rows m = (wm*MT+mt)*32 + 8*g4 + 4*hh + (0..3)
                const int mbase = (wm * MT + mt) * 32 + 8 * g4 + 4 * hh;
                int xl, yl, bl;
                brick_coords(mbase, a.lgTW, a.lgTH, xl, yl, bl);
                const int x = X0 + xl, y = Y0 + yl, b = B0 + bl;  // (x,y) even: the 2x2 window's origin
                if (b >= a.B) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = acc[mt][nt][4 * g4 + j] + bias;
                    if (!a.linear) v[j] = fmaxf(v[j], 0.0f);
                }
                if constexpr (POOL) {
                    if (x < W && y < H) {
                        const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                        a.out[((((size_t)b * (H >> 1)) + (y >> 1)) * (W >> 1) + (x >> 1)) * a.Cout + n] = mx;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int xx = x + (j & 1), yy = y + (j >> 1);
                        if (xx < W && yy < H) {
                            const size_t o = (((size_t)b * H + yy) * W + xx) * a.Cout + n;
                            a.out[o] = (a.mask && !(a.mask[o] > 0.0f)) ? 0.0f : v[j];
                        }
                    }
                }
            }
        }
    }
}


// ---------------------------------------------------------------- FC split-K GEMM --------------

struct FcArgs {
    const float* A;   // [M][K] row-major (lda = K)
    const float* Wt;  // [N][K]
    float* slab;      // [S][Mtiles*32][Npad]
    int M, N, K, S, kchunk, Npad;
};

// One workgroup: 32 rows (batch) x 128 columns x one K slice.  4 waves, each 32x32.
// The weights are read once (FC1: 411 MB per model and batch) and every byte is used for only 32 rows: the kernel is a
// stream of HBM reads, and what it needs is BYTES IN FLIGHT -- about 2 us of latency x 8 TB/s = 16 MB over the chip, 64 KB
// per CU.  Round 2's loop kept ONE 16-float step per thread in registers beside the one in LDS (12 KB per workgroup,
// two workgroups per CU: 2.3 TB/s, latency-bound).  Now a ring of kFcDepth register sets holds the steps t + 1 ..
// t + kFcDepth while step t is multiplied out of LDS: 48 B x kFcDepth per thread = 48 KB per workgroup in flight; the loads
// return in order, so the set that goes to LDS next is always the oldest one outstanding (counted vmcnt, no full drain).
// Same products, same order of accumulation per slab as before (K ascending inside a slice, slabs reduced in fixed order).
constexpr int kFcDepth = 4;
__global__ void __launch_bounds__(256) k_fc_splitk(FcArgs a)
{
    __shared__ __attribute__((aligned(16))) float sA[2][32 * kLdsStride];
    __shared__ __attribute__((aligned(16))) float sB[2][128 * kLdsStride];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntiles = a.Npad / 128;
    const int n_tile = blockIdx.x % ntiles, s = blockIdx.x / ntiles;
    const int m0 = blockIdx.y * 32, n0 = n_tile * 128;
    const int k0 = s * a.kchunk;
    const int kend = (k0 + a.kchunk < a.K) ? k0 + a.kchunk : a.K;
    const int T = (kend - k0) / kBK;
    const int q = tid & 3, rowbase = tid >> 2;

    // Every load is UNCONDITIONAL (rows beyond M / columns beyond N read row 0 instead: their products land in slab rows /
    // columns k_fc_reduce never reads; steps beyond the slice re-read its last step): a branch around a load makes the
    // compiler's wait-count pass fall back to vmcnt(0) at every use, which would drain the whole ring at every step.
    const bool a_loader = rowbase < 32;
    const int am = m0 + (rowbase & 31);
    const float* arow = a.A + (size_t)(am < a.M ? am : 0) * a.K + 4 * q;
    const float* brow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + rowbase + 64 * i;
        brow[i] = a.Wt + (size_t)(n < a.N ? n : 0) * a.K + 4 * q;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    f32x4 ra[kFcDepth], rb[kFcDepth][2];
    const int tlast = T > 0 ? T - 1 : 0;
    auto gload = [&](auto jc, int t) {  // step min(t, T - 1) of the slice into register set j
        constexpr int j = decltype(jc)::value;
        const int k = k0 + (t < tlast ? t : tlast) * kBK;
        ra[j] = *reinterpret_cast<const f32x4*>(arow + k);
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[j][i] = *reinterpret_cast<const f32x4*>(brow[i] + k);
    };
    auto lstore = [&](auto jc, int buf) {
        constexpr int j = decltype(jc)::value;
        if (a_loader) *reinterpret_cast<f32x4*>(&sA[buf][rowbase * kLdsStride + 4 * q]) = ra[j];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&sB[buf][(rowbase + 64 * i) * kLdsStride + 4 * q]) = rb[j][i];
    };
    const int r31 = lane & 31, hh = lane >> 5;
    if (T <= 0) return;  // (uniform; a slice always has at least one step)
    // prologue: steps 0 .. kFcDepth - 1 on their way, step 0 into LDS, then its set refilled with step kFcDepth
    static_for<kFcDepth>([&](auto jc) { gload(jc, jc.value); });
    lstore(std::integral_constant<int, 0>{}, 0);
    gload(std::integral_constant<int, 0>{}, kFcDepth);
    __syncthreads();
    // step t sits in LDS buffer t & 1; register set (t + 1) % kFcDepth holds step t + 1 (the oldest load outstanding)
    auto body = [&](auto jc, int t) {
        constexpr int jn = (decltype(jc)::value + 1) % kFcDepth;  // set of step t + 1, where t % kFcDepth == jc
        const int buf = t & 1;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float4 fa = *reinterpret_cast<const float4*>(&sA[buf][r31 * kLdsStride + 8 * g + 4 * hh]);
            const float4 fb = *reinterpret_cast<const float4*>(&sB[buf][(wave * 32 + r31) * kLdsStride + 8 * g + 4 * hh]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb.w, acc, 0, 0, 0);
        }
        lstore(std::integral_constant<int, jn>{}, buf ^ 1);  // (past the last step: a copy of it that nobody multiplies)
        gload(std::integral_constant<int, jn>{}, t + 1 + kFcDepth);
        __syncthreads();
    };
    int t = 0;
    for (; t + kFcDepth <= T; t += kFcDepth)
        static_for<kFcDepth>([&](auto jc) { body(jc, t + jc.value); });
    static_for<kFcDepth>([&](auto jc) {
        if (t + jc.value < T) body(jc, t + jc.value);
    });
    // slab[s][m][n]
    const int n = n0 + wave * 32 + r31;
    const size_t Mp = (size_t)gridDim.y * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        a.slab[((size_t)s * Mp + m) * a.Npad + n] = acc[r];
    }
}

// out[m][n] = act(bias[n] + sum_s slab[s][m][n]); fixed summation order s = 0..S-1.
__global__ void k_fc_reduce(const float* __restrict__ slab, const float* __restrict__ bias, float* __restrict__ out,
                            int M, int N, int Npad, int Mp, int S, int relu)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * N) return;
    const int m = idx / N, n = idx - m * N;
    float acc = bias[n];
    for (int s = 0; s < S; ++s) acc += slab[((size_t)s * Mp + m) * Npad + n];
    out[idx] = relu ? fmaxf(acc, 0.0f) : acc;
}

// Sheet03/spatialModel.py:219-221: mean CE over the batch, first-max argmax, correct count.
__global__ void k_validate_batch(const float* __restrict__ logits, const long long* __restrict__ labels, int B, int C,
                                 float* __restrict__ out)
{
    __shared__ float sloss[256];
    __shared__ int scorr[256];
    float loss = 0.0f;
    int corr = 0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* l = logits + (size_t)b * C;
        float mx = l[0];
        int am = 0;
        for (int c = 1; c < C; ++c)
            if (l[c] > mx) { mx = l[c]; am = c; }
        float se = 0.0f;
        for (int c = 0; c < C; ++c) se += expf(l[c] - mx);
        // A label outside [0, C) (nn.CrossEntropyLoss refuses it: "Target y is out of bounds") reads nothing and makes
        // the batch loss NaN; the Python wrapper raises ValueError before the call when the labels are on the host.
        const long long y = labels[b];
        const bool yok = y >= 0 && y < (long long)C;
        loss += yok ? (logf(se) + mx) - l[yok ? y : 0] : __builtin_nanf("");
        corr += (yok && am == (int)y);
    }
    sloss[threadIdx.x] = loss;
    scorr[threadIdx.x] = corr;
    __syncthreads();
    if (threadIdx.x == 0) {
        float L = 0.0f;
        int Cc = 0;
        for (int i = 0; i < (int)blockDim.x; ++i) { L += sloss[i]; Cc += scorr[i]; }
        out[0] = L / (float)B;
        out[1] = (float)Cc;
    }
}

// ---------------------------------------------------------------- host side --------------------

constexpr int kConvCout[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
constexpr bool kConvPool[13] = {false, true, false, true, false, false, true, false, false, true, false, false, true};

}  // namespace

namespace {

// Tile / staging policy of the conv launches (measured on MI355X)
constexpr long VA_WIDE_MIN = 512;           // register-staged fp32 kernel: 128-channel tiles from this many workgroups
constexpr int VA_RING = 3;                  // depth of the LDS-DMA ring (two workgroups per CU)
constexpr long VA_RING_MAXGRID = 1024;      // bf16: ring + 64-channel tiles below this many workgroups (the 14x14 layers)
constexpr long VA_RING_MAXGRID_F32 = 1024;  // fp32: the same threshold (0 and 4096 measured 2-4 % slower)
#ifndef VA_WS_DEFAULT
#define VA_WS_DEFAULT 1     // bf16: the weights-resident kernel on the layers with 64 input channels
#endif
#ifndef VA_IMG14_F32_DEFAULT
#define VA_IMG14_F32_DEFAULT 1  // fp32: the same kernel (T = float) on the 14 x 14 layers of the inference path
#endif
#ifndef VA_IMG14_DEFAULT
#define VA_IMG14_DEFAULT 1  // bf16: the one-image-per-workgroup kernel on the 14 x 14 layers (k_conv3x3_img14)
#endif
#ifndef VA_BPP_DEFAULT
#define VA_BPP_DEFAULT 0    // bf16: 1 = the two-group halo-brick kernel wherever it applies (set after measurement)
#endif
#ifndef VA_PP_DEFAULT
#define VA_PP_DEFAULT 1     // bf16: the two-group kernel where launch_conv_bf16 measured it faster (0: never by default)
#endif
constexpr int VA_F32_CONV_DEFAULT = 1;      // 1: LDS-DMA fp32 kernel where Cin % 32 == 0; va_vgg16_set_option(VA_OPT_F32_CONV_KERNEL, 0) selects the register-staged one (A/B)
constexpr int VA_CIN_ALIGN = 16;            // fp32 first-layer channel padding (3 -> 16: register-staged kernel; 20 -> 32: DMA kernel)

void pick_brick(int W, int H, int B, int& lgTW, int& lgTH, int& TB, int lgpx = 7)
{
    // 128 (2^lgpx) pixels = TW x TH x TB, powers of two, TW,TH >= 2: maximise the fraction of real pixels.
    double best = -1.0;
    for (int lw = 1; lw <= 5; ++lw)
        for (int lh = 1; lw + lh <= lgpx; ++lh) {
            const int tw = 1 << lw, th = 1 << lh, tb = (1 << lgpx) / (tw * th);
            const double cover = (double)va_cdiv(W, tw) * tw * va_cdiv(H, th) * th * (double)va_cdiv(B, tb) * tb;
            double util = (double)W * H * B / cover;
            util += 1e-3 * lw;  // tie-break: wider bricks (longer contiguous runs in x)
            if (lgpx != 7) util += 5e-4 * lh;  // (256-pixel bricks: then taller ones -- the taps of a brick overlap in cache)
            if (util > best) {
                best = util;
                lgTW = lw;
                lgTH = lh;
                TB = tb;
            }
        }
}

int launch_conv_ex(int hw, int cin_pad, int cout, const float* wp, const float* bias, const float* in, float* out,
                   const float* mask, int linear, bool pool, int B, const float* zeros, int f32_conv, hipStream_t st)
{
    ConvArgs a{};
    a.in = in;
    a.wp = wp;
    a.bias = bias;
    a.out = out;
    a.mask = mask;
    a.linear = linear;
    a.zeros = zeros;
    a.B = B;
    a.H = a.W = hw;
    a.Cin = cin_pad;
    a.Cout = cout;
    pick_brick(hw, hw, B, a.lgTW, a.lgTH, a.TB);
    a.tiles_x = va_cdiv(hw, 1 << a.lgTW);
    a.tiles_y = va_cdiv(hw, 1 << a.lgTH);
    const int tiles_b = va_cdiv(B, a.TB);
    // 14 x 14 layers (conv5_x) of the inference path: one image x 64 output channels per workgroup (k_conv3x3_img14<float>,
    // round 3): 256 workgroups at batch 32 = one round of the CUs instead of 392 tap-major tiles (53 % MFMA utilisation)
    if (f32_conv == 1 && VA_IMG14_F32_DEFAULT && hw == 14 && mask == nullptr && !linear && cin_pad % 32 == 0 && cout % 64 == 0) {
        const Img14Args ia{in, wp, bias, out, B, cin_pad, cout};
        const unsigned gridi = (unsigned)(B * (cout / 64));
        if (pool) k_conv3x3_img14<float, true><<<gridi, 512, 0, st>>>(ia);
        else k_conv3x3_img14<float, false><<<gridi, 512, 0, st>>>(ia);
        VA_LAUNCH_CHECK();
        return VA_OK;
    }
    if (f32_conv == 1 && zeros != nullptr && cin_pad % kDmaBK == 0) {
        const long grid64 = (long)(cout / 64) * a.tiles_x * a.tiles_y * tiles_b;
        const int ksteps = 9 * (cin_pad / kDmaBK);
        const bool ring = grid64 < VA_RING_MAXGRID_F32 && ksteps >= VA_RING;
        const bool wide = !ring && cout % 128 == 0;
        a.tiles_n = cout / (wide ? 128 : 64);
        const unsigned grid = (unsigned)(a.tiles_n * a.tiles_x * a.tiles_y * tiles_b);
#define VA_LAUNCH_F32(NT_, NB_)                                                      \
    {                                                                                \
        if (pool) k_conv3x3_dma_f32<NT_, true, NB_><<<grid, 256, 0, st>>>(a);        \
        else k_conv3x3_dma_f32<NT_, false, NB_><<<grid, 256, 0, st>>>(a);            \
    }
        if (ring) VA_LAUNCH_F32(1, VA_RING)
        else if (wide) VA_LAUNCH_F32(2, 1)
        else VA_LAUNCH_F32(1, 1)
#undef VA_LAUNCH_F32
        VA_LAUNCH_CHECK();
        return VA_OK;
    }
    // BK = 32 (whole 128-byte lines per pixel, half the barriers) was measured 5 % SLOWER than BK = 16 with register
    // staging: its 72 KB of LDS per workgroup drops the occupancy from 3 to 2 workgroups per CU.
    // 128-channel tiles only where they still give every CU two workgroups (not the 14x14 layers at B = 32)
    const bool wide = cout % 128 == 0 && (long)(cout / 128) * a.tiles_x * a.tiles_y * tiles_b >= VA_WIDE_MIN;
    if (wide) {
        a.tiles_n = cout / 128;
        const unsigned grid = (unsigned)(a.tiles_n * a.tiles_x * a.tiles_y * tiles_b);
        if (pool) k_conv3x3_mfma<2, 2, 2, 2, true, 16><<<grid, 256, 0, st>>>(a);
        else k_conv3x3_mfma<2, 2, 2, 2, false, 16><<<grid, 256, 0, st>>>(a);
    } else {
        a.tiles_n = cout / 64;
        const unsigned grid = (unsigned)(a.tiles_n * a.tiles_x * a.tiles_y * tiles_b);
        if (pool) k_conv3x3_mfma<2, 2, 2, 1, true, 16><<<grid, 256, 0, st>>>(a);
        else k_conv3x3_mfma<2, 2, 2, 1, false, 16><<<grid, 256, 0, st>>>(a);
    }
    VA_LAUNCH_CHECK();
    return VA_OK;
}

int launch_conv(const ConvLayer& L, const float* zeros, const float* in, float* out, int B, int f32_conv, hipStream_t st)
{
    return launch_conv_ex(L.hw, L.cin_pad, L.cout, L.wp, L.bias, in, out, nullptr, 0, L.pool, B, zeros, f32_conv, st);
}

int launch_conv_bf16(const ConvLayer& L, const __bf16* zeros, int variant, const __bf16* in, void* out, bool out_f32, int B, int n_cu,
                     hipStream_t st)
{
    ConvArgsBf a{};
    a.in = in;
    a.zeros = zeros;
    a.taps_x = L.xcol ? 1 : 3;
    a.wp = L.wp_bf;
    a.bias = L.bias;
    a.out = out;
    a.B = B;
    a.H = a.W = L.hw;
    a.Cin = L.cin_pad;
    a.Cout = L.cout;
    // the staging addresses a layer's activations with 32-bit buffer offsets (bytes)
    VA_CHECK_ARG(((long)B * L.hw * L.hw + 2L * (L.hw + 1)) * L.cin_pad * 2 < 2147483647L,
                 "va_vgg16_forward (bf16): batch %d x %dx%dx%d activations exceed 2 GiB of 32-bit buffer offsets: split the batch", B,
                 L.hw, L.hw, L.cin_pad);
    pick_brick(L.hw, L.hw, B, a.lgTW, a.lgTH, a.TB);
    a.tiles_x = va_cdiv(L.hw, 1 << a.lgTW);
    a.tiles_y = va_cdiv(L.hw, 1 << a.lgTH);
    const int tiles_b = va_cdiv(B, a.TB);
    // Tile width and staging scheme by the number of 128-pixel x 64-channel workgroups the layer has:
    //  - few (14x14 and, with VA_RING_MAXGRID, 28x28 layers): 64-channel tiles and the 3-deep DMA ring at
    //    two workgroups per CU (all workgroups resident at once);
    //  - many: 128-channel tiles, single buffer, four workgroups per CU.
    // The two-group kernel (k_conv3x3_pp_bf16).  variant 5: on every layer with >= 128 output channels and at least
    // 28 x 28 pixels.  Default: where it measured faster (B = 32, profiles/README.md): the 256-channel tiles whose grid of
    // 256-pixel x 256-channel workgroups is ONE round of the 256 CUs (the 28 x 28 layers: 196 workgroups, 105 / 107 us
    // against 117 / 139 us), or many rounds; with 392 workgroups (56 x 56) the second round is half empty and the
    // 128 x 128 tiles at four workgroups per CU are as fast
    bool pp = !L.xcol && !out_f32 && a.Cin % 32 == 0 && L.cout % 128 == 0 && L.hw >= 28 && (variant == 5 || variant == 0);
    if (pp) {
        int lw, lh, tb;
        pick_brick(L.hw, L.hw, B, lw, lh, tb, 8);
        const bool nt4 = L.cout % 256 == 0;
        const long gridp = (long)(L.cout / (nt4 ? 256 : 128)) * va_cdiv(L.hw, 1 << lw) * va_cdiv(L.hw, 1 << lh) * va_cdiv(B, tb);
        if (variant == 0) pp = VA_PP_DEFAULT && nt4 && ((gridp >= 160 && gridp <= 256) || gridp >= 1024);
    }
    if (pp) {
        pick_brick(L.hw, L.hw, B, a.lgTW, a.lgTH, a.TB, 8);
        a.tiles_x = va_cdiv(L.hw, 1 << a.lgTW);
        a.tiles_y = va_cdiv(L.hw, 1 << a.lgTH);
        const bool nt4 = L.cout % 256 == 0;
        a.tiles_n = L.cout / (nt4 ? 256 : 128);
        const unsigned gridp = (unsigned)(a.tiles_n * a.tiles_x * a.tiles_y * va_cdiv(B, a.TB));
        if (nt4) {
            if (L.pool) k_conv3x3_pp_bf16<4, true><<<gridp, 512, 0, st>>>(a);
            else k_conv3x3_pp_bf16<4, false><<<gridp, 512, 0, st>>>(a);
        } else {
            if (L.pool) k_conv3x3_pp_bf16<2, true><<<gridp, 512, 0, st>>>(a);
            else k_conv3x3_pp_bf16<2, false><<<gridp, 512, 0, st>>>(a);
        }
        VA_LAUNCH_CHECK();
        return VA_OK;
    }
    // 14 x 14 layers (conv5_x): one image x 64 output channels per workgroup, the image's halo brick staged once per
    // 64-channel chunk (k_conv3x3_img14): the default there; variants 1, 2 keep the tap-major kernels for A/B and tests
    if ((variant == 0 || variant >= 5) && VA_IMG14_DEFAULT && !L.xcol && L.hw == 14 && a.Cin % 64 == 0 && L.cout % 64 == 0 &&
        (long)B * (L.cout / 64) <= 65535L * 16 && ((out_f32 && L.pool) || (!out_f32 && !L.pool))) {
        const unsigned gridi = (unsigned)(B * (L.cout / 64));
        const Img14Args ia{a.in, a.wp, a.bias, a.out, B, a.Cin, a.Cout};
        if (out_f32) k_conv3x3_img14<__bf16, true><<<gridi, 512, 0, st>>>(ia);
        else k_conv3x3_img14<__bf16, false><<<gridi, 512, 0, st>>>(ia);
        VA_LAUNCH_CHECK();
        return VA_OK;
    }
    // The weights-resident kernel (k_conv3x3_ws_bf16) on the layers with 64 input channels: variant 7 forces it (conv1_2
    // and conv2_1), the default uses it where it measured faster: 64 output channels (conv1_2: 135 against 162 us; conv2_1
    // with its two channel halves: 84 against 80).  Same products in the same order as the tap-major kernel: bit-equal
    if ((variant == 7 || (variant == 0 && VA_WS_DEFAULT && L.cout == 64)) && !L.xcol && !out_f32 && a.Cin == 64 && L.cout % 64 == 0 &&
        L.cout <= 256 && L.hw % 16 == 0 && n_cu >= L.cout / 64) {
        ConvArgsWs w{};
        w.c = a;
        w.c.lgTW = w.c.lgTH = 4;
        w.c.TB = 1;
        w.c.tiles_x = w.c.tiles_y = L.hw / 16;
        w.n_bricks = B * w.c.tiles_x * w.c.tiles_y;
        const int nh = L.cout / 64;
        const unsigned gridw = (unsigned)((n_cu / nh) * nh);  // one workgroup per CU (158 KB of LDS), whole channel-half groups
        if (L.pool) k_conv3x3_ws_bf16<true><<<gridw, 256, 0, st>>>(w);
        else k_conv3x3_ws_bf16<false><<<gridw, 256, 0, st>>>(w);
        VA_LAUNCH_CHECK();
        return VA_OK;
    }
#ifdef VA_EXPERIMENTS
    // variant 6: the two-group kernel on halo bricks (k_conv3x3_bpp_bf16) on the layers with >= 128 output channels and
    // 28 x 28 pixels or more (16 x 16 bricks; 8 x 8 of four images; 4 x 4 of sixteen)
    if ((variant == 6 || (variant == 0 && VA_BPP_DEFAULT)) && !L.xcol && !out_f32 && a.Cin % 32 == 0 && L.cout % 128 == 0 &&
        (L.hw == 224 || L.hw == 112 || L.hw == 56 || L.hw == 28)) {
        const int lg = L.hw >= 112 ? 4 : L.hw == 56 ? 3 : 2;
        a.lgTW = a.lgTH = lg;
        a.TB = 256 >> (2 * lg);
        a.tiles_x = a.tiles_y = L.hw >> lg;
        const bool nt4 = L.cout % 256 == 0;
        a.tiles_n = L.cout / (nt4 ? 256 : 128);
        const unsigned gridb = (unsigned)(a.tiles_n * a.tiles_x * a.tiles_y * va_cdiv(B, a.TB));
#define VA_LAUNCH_BPP(LG_, NT_)                                                              \
    {                                                                                        \
        if (L.pool) k_conv3x3_bpp_bf16<LG_, LG_, NT_, true><<<gridb, 512, 0, st>>>(a);       \
        else k_conv3x3_bpp_bf16<LG_, LG_, NT_, false><<<gridb, 512, 0, st>>>(a);             \
    }
        if (lg == 4 && nt4) VA_LAUNCH_BPP(4, 4)
        else if (lg == 4) VA_LAUNCH_BPP(4, 2)
        else if (lg == 3 && nt4) VA_LAUNCH_BPP(3, 4)
        else if (lg == 3) VA_LAUNCH_BPP(3, 2)
        else if (nt4) VA_LAUNCH_BPP(2, 4)
        else VA_LAUNCH_BPP(2, 2)
#undef VA_LAUNCH_BPP
        VA_LAUNCH_CHECK();
        return VA_OK;
    }
#endif
    const long grid64 = (long)(L.cout / 64) * a.tiles_x * a.tiles_y * tiles_b;
    const int ksteps = 3 * a.taps_x * (a.Cin / 64);
    const bool autosel = variant == 0 || variant >= 5;  // (variants 5 .. 7 fall through to the automatic choice where they do not apply)
    const bool ring = autosel ? (grid64 < VA_RING_MAXGRID && ksteps >= VA_RING) : (variant == 2 && ksteps >= VA_RING);
    const bool wide = autosel && !ring && L.cout % 128 == 0;
    a.tiles_n = L.cout / (wide ? 128 : 64);
    const unsigned grid = (unsigned)(a.tiles_n * a.tiles_x * a.tiles_y * tiles_b);
#define VA_LAUNCH_BF(NT_, NB_)                                                                   \
    {                                                                                            \
        if (out_f32) k_conv3x3_mfma_bf16<NT_, true, true, NB_><<<grid, 256, 0, st>>>(a);         \
        else if (L.pool) k_conv3x3_mfma_bf16<NT_, true, false, NB_><<<grid, 256, 0, st>>>(a);    \
        else k_conv3x3_mfma_bf16<NT_, false, false, NB_><<<grid, 256, 0, st>>>(a);               \
    }
    if (ring) VA_LAUNCH_BF(1, VA_RING)
    else if (wide) VA_LAUNCH_BF(2, 1)
    else VA_LAUNCH_BF(1, 1)
#undef VA_LAUNCH_BF
    VA_LAUNCH_CHECK();
    return VA_OK;
}

struct FcPlan {
    int S, kchunk, Npad, Mtiles;
    size_t slab_floats;
};

FcPlan plan_fc(int M, int N, int K)
{
    FcPlan p;
    p.Npad = va_cdiv(N, 128) * 128;
    p.Mtiles = va_cdiv(M, 32);
    const int ntiles = p.Npad / 128;
    // aim for >= ~512 workgroups, K slices multiples of 16, at least 256 deep
    int S = va_cdiv(512, ntiles * p.Mtiles);
    int maxS = K / 256;
    if (maxS < 1) maxS = 1;
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    p.kchunk = va_cdiv(va_cdiv(K, S), kBK) * kBK;
    p.S = va_cdiv(K, p.kchunk);
    p.slab_floats = (size_t)p.S * p.Mtiles * 32 * p.Npad;
    return p;
}

int launch_fc(const float* A, const float* Wt, const float* bias, float* out, float* slab, int M, int N, int K,
              bool relu, hipStream_t st)
{
    const FcPlan p = plan_fc(M, N, K);
    FcArgs a{};
    a.A = A;
    a.Wt = Wt;
    a.slab = slab;
    a.M = M;
    a.N = N;
    a.K = K;
    a.S = p.S;
    a.kchunk = p.kchunk;
    a.Npad = p.Npad;
    const dim3 grid((unsigned)((p.Npad / 128) * p.S), (unsigned)p.Mtiles);
    k_fc_splitk<<<grid, 256, 0, st>>>(a);
    k_fc_reduce<<<va_cdiv(M * N, 256), 256, 0, st>>>(slab, bias, out, M, N, p.Npad, p.Mtiles * 32, p.S, relu ? 1 : 0);
    VA_LAUNCH_CHECK();
    return VA_OK;
}

struct WsPlan {
    size_t off_act[2], off_slab, off_fc[2], total;
};

WsPlan plan_ws(const va_vgg16* m, int B)
{
    WsPlan w;
    size_t off = 0;
    const size_t act = va_align_up((size_t)B * 224 * 224 * 64 * sizeof(float), 256);
    w.off_act[0] = off; off += act;
    w.off_act[1] = off; off += act;
    size_t slab = 0;
    for (int i = 0; i < 4; ++i) {
        const FcPlan p = plan_fc(B, m->fc_out[i], m->fc_in[i]);
        if (p.slab_floats > slab) slab = p.slab_floats;
    }
    w.off_slab = off; off += va_align_up(slab * sizeof(float), 256);
    const size_t fc = va_align_up((size_t)B * 4096 * sizeof(float), 256);
    w.off_fc[0] = off; off += fc;
    w.off_fc[1] = off; off += fc;
    w.total = off;
    return w;
}

}  // namespace

int va_conv3x3_f32(int hw, int cin_pad, int cout, const float* wp, const float* bias, const float* in, float* out,
                   const float* mask, int linear, int pool, int B, const float* zeros, int f32_conv, hipStream_t st)
{
    return launch_conv_ex(hw, cin_pad, cout, wp, bias, in, out, mask, linear, pool != 0, B, zeros, f32_conv, st);
}

int va_fc_f32(const float* A, const float* Wt, const float* bias, float* out, float* slab, int M, int N, int K, int relu, hipStream_t st)
{
    return launch_fc(A, Wt, bias, out, slab, M, N, K, relu != 0, st);
}

size_t va_fc_slab_floats(int M, int N, int K) { return plan_fc(M, N, K).slab_floats; }

int va_input_to_nhwc_f32(const va_vgg16* m, const void* x, int x_is_u8, int B, float* out, hipStream_t st)
{
    const int HW0 = 224 * 224;
    const unsigned pgrid = (unsigned)(B * ((HW0 + 63) / 64));
    if (x_is_u8)
        k_nchw_to_nhwc_pad<unsigned char, float><<<pgrid, 256, 0, st>>>((const unsigned char*)x, out, B, m->c_in, HW0, m->c_in_pad, m->in_mean, m->in_std);
    else
        k_nchw_to_nhwc_pad<float, float><<<pgrid, 256, 0, st>>>((const float*)x, out, B, m->c_in, HW0, m->c_in_pad, nullptr, nullptr);
    VA_LAUNCH_CHECK();
    return VA_OK;
}

// FC1..FC4 on an NHWC feature map f [B][7][7][512] (Sheet03/spatialModel.py:213-218).
static int run_classifier(va_vgg16* m, const float* f, int B, void* desc, void* logits, float* slab, float* const* fcbuf,
                          hipStream_t st)
{
    if (int rc = launch_fc(f, m->fcw[0], m->fcb[0], fcbuf[0], slab, B, 4096, 512 * 49, true, st)) return rc;
    if (int rc = launch_fc(fcbuf[0], m->fcw[1], m->fcb[1], fcbuf[1], slab, B, 4096, 4096, true, st)) return rc;
    float* d = desc ? (float*)desc : fcbuf[0];
    if (int rc = launch_fc(fcbuf[1], m->fcw[2], m->fcb[2], d, slab, B, m->desc_dim, 4096, true, st)) return rc;
    if (logits)
        if (int rc = launch_fc(d, m->fcw[3], m->fcb[3], (float*)logits, slab, B, m->n_classes, m->desc_dim, false, st)) return rc;
    return VA_OK;
}

extern "C" int va_vgg16_create(va_ctx* ctx, int c_in, int n_classes, int desc_dim, int dtype,
                               const void* const* conv_w, const void* const* conv_b, const void* const* fc_w,
                               const void* const* fc_b, const float* in_mean, const float* in_std, void* stream,
                               va_vgg16** out)
{
    VA_CHECK_ARG(ctx != nullptr && out != nullptr, "va_vgg16_create: NULL ctx/out");
    VA_USE_DEVICE(ctx);
    *out = nullptr;
    VA_CHECK_ARG(c_in >= 1 && c_in <= 64, "va_vgg16_create: c_in %d out of range [1,64]", c_in);
    VA_CHECK_ARG(n_classes >= 1 && n_classes <= 4096 && desc_dim >= 16 && desc_dim <= 4096 && desc_dim % 16 == 0,
                 "va_vgg16_create: n_classes %d / desc_dim %d unsupported (desc_dim must be a multiple of 16)", n_classes, desc_dim);
    VA_CHECK_ARG(dtype == VA_DTYPE_F32 || dtype == VA_DTYPE_BF16, "va_vgg16_create: dtype must be VA_DTYPE_F32 or VA_DTYPE_BF16");
    VA_CHECK_ARG(conv_w && conv_b && fc_w && fc_b, "va_vgg16_create: NULL weight tables");
    for (int i = 0; i < 13; ++i) VA_CHECK_ARG(conv_w[i] && conv_b[i], "va_vgg16_create: conv layer %d weight/bias is NULL", i);
    for (int i = 0; i < 4; ++i) VA_CHECK_ARG(fc_w[i] && fc_b[i], "va_vgg16_create: fc layer %d weight/bias is NULL", i);
    hipStream_t st = (hipStream_t)stream;
    va_vgg16* m = new va_vgg16();
    memset(m, 0, sizeof(*m));
    m->ctx = ctx;
    m->c_in = c_in;
    m->dtype = dtype;
    m->c_in_pad = dtype == VA_DTYPE_BF16 ? 64 : va_cdiv(c_in, VA_CIN_ALIGN) * VA_CIN_ALIGN;
    m->n_classes = n_classes;
    m->desc_dim = desc_dim;
    int hw = 224, cin = c_in, cin_pad = m->c_in_pad;
    int rc = VA_OK;
    auto fail = [&](int code) { va_vgg16_destroy(m); return code; };
    if (hipMalloc(&m->zeros_f32, 512 * sizeof(float)) != hipSuccess || hipMemsetAsync(m->zeros_f32, 0, 512 * sizeof(float), st) != hipSuccess) {
        va_set_error("va_vgg16_create: hipMalloc failed for the zero line");
        return fail(VA_ERR_HIP);
    }
    m->bf16_variant = 0;
    m->bf16_first = 1;
    m->f32_conv = VA_F32_CONV_DEFAULT;
    m->train_stop_at = -1;
    if (dtype == VA_DTYPE_BF16) {
        if (hipMalloc(&m->zeros, 256) != hipSuccess || hipMemsetAsync(m->zeros, 0, 256, st) != hipSuccess) {
            va_set_error("va_vgg16_create: hipMalloc failed for the zero line");
            return fail(VA_ERR_HIP);
        }
    }
    for (int i = 0; i < 13; ++i) {
        ConvLayer& L = m->conv[i];
        L.cin = cin;
        L.cin_pad = cin_pad;
        L.cout = kConvCout[i];
        L.hw = hw;
        L.pool = kConvPool[i];
        const size_t nw = (size_t)L.cout * 9 * L.cin_pad;
        const bool bf = dtype == VA_DTYPE_BF16;
        if ((bf ? hipMalloc(&L.wp_bf, nw * sizeof(__bf16)) : hipMalloc(&L.wp, nw * sizeof(float))) != hipSuccess ||
            hipMalloc(&L.bias, L.cout * sizeof(float)) != hipSuccess) {
            va_set_error("va_vgg16_create: hipMalloc failed for conv layer %d", i);
            return fail(VA_ERR_HIP);
        }
        L.xcol = bf && i == 0 && 3 * L.cin <= 64 && L.cin <= 21;
        if (L.xcol) {
            k_pack_conv_w_bf16_xcol<<<va_cdiv(L.cout * 192, 256), 256, 0, st>>>((const float*)conv_w[i], L.wp_bf, L.cout, L.cin);
            // the fused first layer (k_conv1_fused_bf16): its own packing of the same weights
            m->f1_cp = (L.cin + 3) & ~3;
            m->f1_krow = (3 * m->f1_cp + 15) & ~15;
            if (hipMalloc(&m->wp_f1, (size_t)64 * 3 * m->f1_krow * sizeof(__bf16)) != hipSuccess) {
                va_set_error("va_vgg16_create: hipMalloc failed for the first layer's weights");
                return fail(VA_ERR_HIP);
            }
            k_pack_conv_w_bf16_f1<<<va_cdiv(64 * 3 * m->f1_krow, 256), 256, 0, st>>>((const float*)conv_w[i], m->wp_f1, 64, L.cin, m->f1_cp, m->f1_krow);
        }
        else if (bf) k_pack_conv_w_bf16<<<(unsigned)((nw + 255) / 256), 256, 0, st>>>((const float*)conv_w[i], L.wp_bf, L.cout, L.cin, L.cin_pad);
        else k_pack_conv_w<<<(unsigned)((nw + 255) / 256), 256, 0, st>>>((const float*)conv_w[i], L.wp, L.cout, L.cin, L.cin_pad);
        if (hipMemcpyAsync(L.bias, conv_b[i], L.cout * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) rc = VA_ERR_HIP;
        cin = cin_pad = L.cout;
        if (L.pool) hw /= 2;
    }
    const int fin[4] = {512 * 7 * 7, 4096, 4096, desc_dim};
    const int fout[4] = {4096, 4096, desc_dim, n_classes};
    for (int i = 0; i < 4; ++i) {
        m->fc_in[i] = fin[i];
        m->fc_out[i] = fout[i];
        const size_t nw = (size_t)fin[i] * fout[i];
        if (hipMalloc(&m->fcw[i], nw * sizeof(float)) != hipSuccess || hipMalloc(&m->fcb[i], fout[i] * sizeof(float)) != hipSuccess) {
            va_set_error("va_vgg16_create: hipMalloc failed for fc layer %d", i);
            return fail(VA_ERR_HIP);
        }
        if (i == 0) k_pack_fc1<<<(unsigned)((nw + 255) / 256), 256, 0, st>>>((const float*)fc_w[0], m->fcw[0], fout[0], 512, 49);
        else if (hipMemcpyAsync(m->fcw[i], fc_w[i], nw * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) rc = VA_ERR_HIP;
        if (hipMemcpyAsync(m->fcb[i], fc_b[i], fout[i] * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) rc = VA_ERR_HIP;
    }
    if (in_mean && in_std) {
        if (hipMalloc(&m->in_mean, c_in * sizeof(float)) != hipSuccess || hipMalloc(&m->in_std, c_in * sizeof(float)) != hipSuccess) {
            va_set_error("va_vgg16_create: hipMalloc failed for input normalisation");
            return fail(VA_ERR_HIP);
        }
        if (hipMemcpyAsync(m->in_mean, in_mean, c_in * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) rc = VA_ERR_HIP;
        if (hipMemcpyAsync(m->in_std, in_std, c_in * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess) rc = VA_ERR_HIP;
    }
    if (hipGetLastError() != hipSuccess) rc = VA_ERR_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) rc = VA_ERR_HIP;
    if (rc != VA_OK) {
        va_set_error("va_vgg16_create: HIP failure while packing weights");
        return fail(rc);
    }
    *out = m;
    return VA_OK;
}

extern "C" void va_vgg16_destroy(va_vgg16* m)
{
    if (!m) return;
    for (int i = 0; i < 13; ++i) {
        if (m->conv[i].wp) (void)hipFree(m->conv[i].wp);
        if (m->conv[i].wp_bf) (void)hipFree(m->conv[i].wp_bf);
        if (m->conv[i].bias) (void)hipFree(m->conv[i].bias);
        if (m->conv[i].mom_w) (void)hipFree(m->conv[i].mom_w);
        if (m->conv[i].mom_b) (void)hipFree(m->conv[i].mom_b);
    }
    if (m->wp_f1) (void)hipFree(m->wp_f1);
    for (int i = 0; i < 4; ++i) {
        if (m->fcw[i]) (void)hipFree(m->fcw[i]);
        if (m->fcb[i]) (void)hipFree(m->fcb[i]);
        if (m->fc_mom_w[i]) (void)hipFree(m->fc_mom_w[i]);
        if (m->fc_mom_b[i]) (void)hipFree(m->fc_mom_b[i]);
    }
    if (m->zeros_f32) (void)hipFree(m->zeros_f32);
    if (m->zeros) (void)hipFree(m->zeros);
    if (m->in_mean) (void)hipFree(m->in_mean);
    if (m->in_std) (void)hipFree(m->in_std);
    delete m;
}

extern "C" size_t va_vgg16_workspace_bytes(const va_vgg16* m, int batch)
{
    if (!m || batch < 1 || batch > 4096) return 0;
    return plan_ws(m, batch).total;
}

extern "C" int va_vgg16_forward(va_vgg16* m, const void* x, int x_is_u8, int batch, void* feat, void* desc, void* logits,
                                void* workspace, size_t workspace_bytes, void* stream)
{
    VA_CHECK_ARG(m != nullptr, "va_vgg16_forward: model is NULL");
    VA_USE_DEVICE(m->ctx);
    VA_CHECK_ARG(x != nullptr && workspace != nullptr, "va_vgg16_forward: NULL input/workspace");
    VA_CHECK_ARG(batch >= 1 && batch <= 4096, "va_vgg16_forward: batch %d out of range [1,4096]", batch);
    VA_CHECK_ARG(!x_is_u8 || (m->in_mean && m->in_std), "va_vgg16_forward: u8 input needs in_mean/in_std at create time");
    VA_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "va_vgg16_forward: workspace must be 256-byte aligned");
    const WsPlan wp = plan_ws(m, batch);
    if (workspace_bytes < wp.total) {
        va_set_error("va_vgg16_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, wp.total);
        return VA_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* act[2] = {(float*)(ws + wp.off_act[0]), (float*)(ws + wp.off_act[1])};
    float* slab = (float*)(ws + wp.off_slab);
    float* fcbuf[2] = {(float*)(ws + wp.off_fc[0]), (float*)(ws + wp.off_fc[1])};
    const int B = batch, HW0 = 224 * 224;
    const unsigned pgrid = (unsigned)(B * ((HW0 + 63) / 64));  // one workgroup per 64 pixels
    int cur = 1;
    if (m->dtype == VA_DTYPE_BF16) {
        // bf16 activations live in the same two ping-pong buffers (half their size is used)
        int first = 0;
        if (m->conv[0].xcol && m->bf16_first == 1 && m->wp_f1 != nullptr && ((uintptr_t)x & 15) == 0) {  // (it loads four pixels at a time)
            // the first layer straight from the NCHW input (no staged 64-channel copy of the input)
            Conv1Args c1{x, m->wp_f1, m->conv[0].bias, (__bf16*)act[0], m->in_mean, m->in_std, B, m->c_in, m->f1_cp, m->f1_krow};
            const unsigned g1 = (unsigned)(B * (224 / 16) * (224 / 16));
            if (x_is_u8) k_conv1_fused_bf16<unsigned char><<<g1, 256, 0, st>>>(c1);
            else k_conv1_fused_bf16<float><<<g1, 256, 0, st>>>(c1);
            first = 1;
            cur = 0;
        } else if (m->conv[0].xcol) {
            if (x_is_u8)
                k_nchw_to_nhwc_xcol<unsigned char><<<pgrid, 256, 0, st>>>((const unsigned char*)x, (__bf16*)act[1], B, m->c_in, 224, HW0, m->in_mean, m->in_std);
            else
                k_nchw_to_nhwc_xcol<float><<<pgrid, 256, 0, st>>>((const float*)x, (__bf16*)act[1], B, m->c_in, 224, HW0, nullptr, nullptr);
        } else if (x_is_u8)
            k_nchw_to_nhwc_pad<unsigned char, __bf16><<<pgrid, 256, 0, st>>>((const unsigned char*)x, (__bf16*)act[1], B, m->c_in, HW0, m->c_in_pad, m->in_mean, m->in_std);
        else
            k_nchw_to_nhwc_pad<float, __bf16><<<pgrid, 256, 0, st>>>((const float*)x, (__bf16*)act[1], B, m->c_in, HW0, m->c_in_pad, nullptr, nullptr);
        VA_LAUNCH_CHECK();
        for (int i = first; i < 13; ++i) {
            if (int rc = launch_conv_bf16(m->conv[i], m->zeros, m->bf16_variant, (const __bf16*)act[cur], act[cur ^ 1], i == 12, B, m->ctx->n_cu, st)) return rc;
            cur ^= 1;
        }
    } else {
        if (x_is_u8)
            k_nchw_to_nhwc_pad<unsigned char, float><<<pgrid, 256, 0, st>>>((const unsigned char*)x, act[1], B, m->c_in, HW0, m->c_in_pad, m->in_mean, m->in_std);
        else
            k_nchw_to_nhwc_pad<float, float><<<pgrid, 256, 0, st>>>((const float*)x, act[1], B, m->c_in, HW0, m->c_in_pad, nullptr, nullptr);
        VA_LAUNCH_CHECK();
        for (int i = 0; i < 13; ++i) {
            if (int rc = launch_conv(m->conv[i], m->zeros_f32, act[cur], act[cur ^ 1], B, m->f32_conv, st)) return rc;
            cur ^= 1;
        }
    }
    const float* f = act[cur];  // NHWC f32 [B][7][7][512] (the last bf16 layer stores f32)
    if (feat) {
        const size_t n = (size_t)B * 512 * 49;
        k_nhwc_to_nchw<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(f, (float*)feat, B, 512, 49);
        VA_LAUNCH_CHECK();
    }
    if (desc || logits) return run_classifier(m, f, B, desc, logits, slab, fcbuf, st);
    return VA_OK;
}

extern "C" int va_vgg16_set_option(va_vgg16* m, int option, int value)
{
    VA_CHECK_ARG(m != nullptr, "va_vgg16_set_option: model is NULL");
    switch (option) {
        case VA_OPT_BF16_VARIANT:
            VA_CHECK_ARG((value >= 0 && value <= 2) || (value >= 5 && value <= 7), "va_vgg16_set_option: VA_OPT_BF16_VARIANT must be 0, 1, 2, 5, 6 or 7");
            VA_CHECK_ARG(value != 6 || kVaExperiments, "va_vgg16_set_option: VA_OPT_BF16_VARIANT 6 (k_conv3x3_bpp_bf16) needs a library built with -DVA_EXPERIMENTS");
            m->bf16_variant = value;
            return VA_OK;
        case VA_OPT_BF16_FIRST_LAYER:
            VA_CHECK_ARG(value == 0 || value == 1, "va_vgg16_set_option: VA_OPT_BF16_FIRST_LAYER must be 0 or 1");
            m->bf16_first = value;
            return VA_OK;
        case VA_OPT_F32_CONV_KERNEL:
            VA_CHECK_ARG(value == 0 || value == 1, "va_vgg16_set_option: VA_OPT_F32_CONV_KERNEL must be 0 or 1");
            m->f32_conv = value;
            return VA_OK;
        case VA_OPT_TRAIN_STOP_AT:
            VA_CHECK_ARG(value >= -1 && value <= 12, "va_vgg16_set_option: VA_OPT_TRAIN_STOP_AT must be in [-1,12]");
            m->train_stop_at = value;
            return VA_OK;
        default:
            va_set_error("va_vgg16_set_option: unknown option %d", option);
            return VA_ERR_INVALID;
    }
}

extern "C" int va_vgg16_classify(va_vgg16* m, const void* feat, int batch, void* desc, void* logits, void* workspace,
                                 size_t workspace_bytes, void* stream)
{
    VA_CHECK_ARG(m != nullptr, "va_vgg16_classify: model is NULL");
    VA_USE_DEVICE(m->ctx);
    VA_CHECK_ARG(feat != nullptr && workspace != nullptr, "va_vgg16_classify: NULL input/workspace");
    VA_CHECK_ARG(batch >= 1 && batch <= 4096, "va_vgg16_classify: batch %d out of range [1,4096]", batch);
    VA_CHECK_ARG(((uintptr_t)workspace & 255) == 0, "va_vgg16_classify: workspace must be 256-byte aligned");
    const WsPlan wp = plan_ws(m, batch);
    if (workspace_bytes < wp.total) {
        va_set_error("va_vgg16_classify: workspace too small (%zu < %zu bytes)", workspace_bytes, wp.total);
        return VA_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    float* nhwc = (float*)(ws + wp.off_act[0]);
    float* slab = (float*)(ws + wp.off_slab);
    float* fcbuf[2] = {(float*)(ws + wp.off_fc[0]), (float*)(ws + wp.off_fc[1])};
    const size_t n = (size_t)batch * 49 * 512;
    k_nchw_to_nhwc<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((const float*)feat, nhwc, batch, 512, 49);
    VA_LAUNCH_CHECK();
    if (desc || logits) return run_classifier(m, nhwc, batch, desc, logits, slab, fcbuf, st);
    return VA_OK;
}

extern "C" int va_copy_first_layer(va_ctx* ctx, const void* w_rgb, int cout, int n_in, void* w_out, void* stream)
{
    VA_CHECK_ARG(ctx != nullptr && w_rgb != nullptr && w_out != nullptr, "va_copy_first_layer: NULL argument");
    VA_USE_DEVICE(ctx);
    VA_CHECK_ARG(cout >= 1 && n_in >= 1 && (long)cout * n_in * 9 < (1L << 30), "va_copy_first_layer: bad shape");
    const int n = cout * n_in * 9;
    k_copy_first_layer<<<va_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>((const float*)w_rgb, (float*)w_out, cout, n_in);
    VA_LAUNCH_CHECK();
    return VA_OK;
}

extern "C" int va_validate_batch(va_ctx* ctx, const void* logits, const void* labels, int batch, int n_classes, void* out,
                                 void* stream)
{
    VA_CHECK_ARG(ctx != nullptr && logits != nullptr && labels != nullptr && out != nullptr, "va_validate_batch: NULL argument");
    VA_USE_DEVICE(ctx);
    VA_CHECK_ARG(batch >= 1 && n_classes >= 1, "va_validate_batch: bad shape");
    k_validate_batch<<<1, 256, 0, (hipStream_t)stream>>>((const float*)logits, (const long long*)labels, batch, n_classes, (float*)out);
    VA_LAUNCH_CHECK();
    return VA_OK;
}
