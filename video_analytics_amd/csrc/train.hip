// Training step of the two-stream VGG-16 on gfx950: the body of the batch loop of
// SpatialNetwork.train() / TemporalNetwork.train() (Sheet03/spatialModel.py:165-182, Sheet03/temporalModel.py:194-211):
// forward in train mode (Dropout p = 0.5 after the three hidden classifier ReLUs), mean cross-entropy, backward
// through the whole network, momentum-SGD update (torch.optim.SGD: buf = momentum*buf + grad; w -= lr*buf; no
// weight decay, no Nesterov -- Sheet03/spatialModel.py:116) of every parameter.  fp32 throughout.
//
// Convolution backward reuses the forward implicit-GEMM kernel for the data gradient (a 3x3 convolution of the
// output gradient with the flipped, transposed weights) and adds one MFMA kernel for the weight gradient
// (dW[co][tap][ci] = sum over pixels of dY[p][co] * X[p + tap][ci]: a GEMM whose K dimension is the pixels,
// split over workgroups and reduced in a fixed order together with the SGD update).
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "vgg_internal.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


// ---------------------------------------------------------------- small kernels ----------------

// NHWC 2x2/2 max-pool, 4 channels per thread
__global__ void k_maxpool(const float* __restrict__ y, float* __restrict__ p, int B, int H, int C)
{
    const int Ho = H >> 1, C4 = C >> 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * Ho * Ho * C4) return;
    const int c4 = (int)(idx % C4);
    size_t r = idx / C4;
    const int xo = (int)(r % Ho);
    r /= Ho;
    const int yo = (int)(r % Ho), b = (int)(r / Ho);
    const f32x4* src = reinterpret_cast<const f32x4*>(y) + (((size_t)b * H + 2 * yo) * H + 2 * xo) * C4 + c4;
    const f32x4 a0 = src[0], a1 = src[C4], a2 = src[(size_t)H * C4], a3 = src[(size_t)H * C4 + C4];
    f32x4 m;
#pragma unroll
    for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(a0[j], a1[j]), fmaxf(a2[j], a3[j]));
    reinterpret_cast<f32x4*>(p)[idx] = m;
}

// Max-pool backward (torch semantics: the gradient goes to the FIRST maximum of the window in row-major order)
// fused with the ReLU mask of the pooled value: dY = (Y == P at the first such position && P > 0) ? dP : 0.
__global__ void k_unpool(const float* __restrict__ dp, const float* __restrict__ y, const float* __restrict__ p,
                         float* __restrict__ dy, int B, int H, int C)
{
    const int Ho = H >> 1, C4 = C >> 2;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * Ho * Ho * C4) return;
    const int c4 = (int)(idx % C4);
    size_t r = idx / C4;
    const int xo = (int)(r % Ho);
    r /= Ho;
    const int yo = (int)(r % Ho), b = (int)(r / Ho);
    const size_t o00 = (((size_t)b * H + 2 * yo) * H + 2 * xo) * C4 + c4;
    const size_t off[4] = {o00, o00 + C4, o00 + (size_t)H * C4, o00 + (size_t)H * C4 + C4};
    const f32x4 pv = reinterpret_cast<const f32x4*>(p)[idx], g = reinterpret_cast<const f32x4*>(dp)[idx];
    f32x4 a[4], out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = reinterpret_cast<const f32x4*>(y)[off[q]];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bool taken = !(pv[j] > 0.0f);  // ReLU mask: nothing flows where the pooled activation is 0
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool hit = !taken && a[q][j] == pv[j];
            out[q][j] = hit ? g[j] : 0.0f;
            taken = taken || hit;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) reinterpret_cast<f32x4*>(dy)[off[q]] = out[q];
}

__device__ __forceinline__ unsigned mix32(unsigned x)
{
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

// Dropout(p = 0.5) in place: element i is kept (and doubled) iff the top bit of mix(mix(i ^ key) + key2) is set,
// i.e. iff synth.hash_uniform(seed, stream)[i] >= 0.5 (the host derives key/key2 from (seed, stream)).
__global__ void k_dropout(float* __restrict__ x, size_t n, unsigned key, unsigned key2)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned h = mix32(mix32((unsigned)i ^ key) + key2);
    x[i] = (h & 0x80000000u) ? 2.0f * x[i] : 0.0f;
}

// logits [B][C], labels i64 -> dlogits = (softmax - onehot) / B; out[0] = mean CE, out[1] = #(argmax == label)
__global__ void k_ce_fwd_bwd(const float* __restrict__ logits, const long long* __restrict__ labels, int B, int C,
                             float* __restrict__ dlogits, float* __restrict__ out)
{
    __shared__ float sloss[256];
    __shared__ int scorr[256];
    float loss = 0.0f;
    int corr = 0;
    const float invB = 1.0f / (float)B;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* l = logits + (size_t)b * C;
        float mx = l[0];
        int am = 0;
        for (int c = 1; c < C; ++c)
            if (l[c] > mx) { mx = l[c]; am = c; }
        float se = 0.0f;
        for (int c = 0; c < C; ++c) se += expf(l[c] - mx);
        // label outside [0, C): no out-of-bounds read; loss and this row's gradient become NaN (nn.CrossEntropyLoss
        // refuses such a target; the Python wrapper raises ValueError first when the labels are on the host)
        const long long y = labels[b];
        const bool yok = y >= 0 && y < (long long)C;
        loss += yok ? (logf(se) + mx) - l[yok ? y : 0] : __builtin_nanf("");
        corr += (yok && am == (int)y);
        const float inv = yok ? 1.0f / se : __builtin_nanf("");
        for (int c = 0; c < C; ++c) dlogits[(size_t)b * C + c] = (expf(l[c] - mx) * inv - (c == (int)y ? 1.0f : 0.0f)) * invB;
    }
    sloss[threadIdx.x] = loss;
    scorr[threadIdx.x] = corr;
    __syncthreads();
    if (threadIdx.x == 0) {
        float L = 0.0f;
        int Cc = 0;
        for (int i = 0; i < (int)blockDim.x; ++i) { L += sloss[i]; Cc += scorr[i]; }
        out[0] = L * invB;
        out[1] = (float)Cc;
    }
}

// ---------------------------------------------------------------- classifier backward ----------

// dX[b][i] = sum_o dZ[b][o] * W[o][i], then * scale where mask[b][i] > 0, else 0 (mask may be NULL).
// A workgroup owns 64 columns i; its four 64-thread groups take the output rows o = 4q + g of every 64-row chunk
// and keep all BMAX batch rows in registers (dZ staged in LDS as [o][b]); the four partial sums are added in
// the fixed order g = 0..3 (deterministic).  25088 / 64 = 392 workgroups for FC1.
template <int BMAX>
__global__ void __launch_bounds__(256) k_fc_dx(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ dx,
                                               int B, int O, int I, const float* __restrict__ mask, float scale)
{
    constexpr int OC = 64;
    __shared__ __attribute__((aligned(16))) float sdz[OC][BMAX];
    __shared__ float spart[3][BMAX][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    float acc[BMAX];
#pragma unroll
    for (int b = 0; b < BMAX; ++b) acc[b] = 0.0f;
    for (int o0 = 0; o0 < O; o0 += OC) {
        __syncthreads();
        for (int t = threadIdx.x; t < OC * BMAX; t += 256) {
            const int oo = t / BMAX, b = t - oo * BMAX;
            sdz[oo][b] = (b < B && o0 + oo < O) ? dz[(size_t)b * O + o0 + oo] : 0.0f;
        }
        __syncthreads();
        if (i < I) {
            const int on = O - o0 < OC ? O - o0 : OC;
#pragma unroll 4
            for (int oo = grp; oo < on; oo += 4) {
                const float wv = w[(size_t)(o0 + oo) * I + i];
#pragma unroll
                for (int b4 = 0; b4 < BMAX / 4; ++b4) {
                    const f32x4 d = *reinterpret_cast<const f32x4*>(&sdz[oo][4 * b4]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 * b4 + j] = fmaf(d[j], wv, acc[4 * b4 + j]);
                }
            }
        }
    }
    if (grp > 0) {
#pragma unroll
        for (int b = 0; b < BMAX; ++b) spart[grp - 1][b][col] = acc[b];
    }
    __syncthreads();
    if (grp != 0 || i >= I) return;
#pragma unroll
    for (int b = 0; b < BMAX; ++b)
        if (b < B) {
            float v = ((acc[b] + spart[0][b][col]) + spart[1][b][col]) + spart[2][b][col];
            if (mask) v = mask[(size_t)b * I + i] > 0.0f ? v * scale : 0.0f;
            dx[(size_t)b * I + i] = v;
        }
}

// g[o][i] = sum_b dZ[b][o] * X[b][i] (b ascending); V = mu*V + g; W -= lr*V.  One thread per column i keeps X[:, i]
// in registers and walks a slice of the output rows; blockIdx.y selects the slice.
template <int BMAX>
__global__ void __launch_bounds__(256) k_fc_wgrad_sgd(const float* __restrict__ dz, const float* __restrict__ x, float* __restrict__ w,
                                                      float* __restrict__ v, int B, int O, int I, int orows, float lr, float mu)
{
    extern __shared__ __attribute__((aligned(16))) float sdz[];  // [orows][BMAX]
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int o0 = blockIdx.y * orows;
    const int on = O - o0 < orows ? O - o0 : orows;
    for (int t = threadIdx.x; t < orows * BMAX; t += 256) {
        const int oo = t / BMAX, b = t - oo * BMAX;
        sdz[t] = (b < B && oo < on) ? dz[(size_t)b * O + o0 + oo] : 0.0f;
    }
    __syncthreads();
    if (i >= I) return;
    float xr[BMAX];
#pragma unroll
    for (int b = 0; b < BMAX; ++b) xr[b] = b < B ? x[(size_t)b * I + i] : 0.0f;
    for (int oo = 0; oo < on; ++oo) {
        float g = 0.0f;
#pragma unroll
        for (int b4 = 0; b4 < BMAX / 4; ++b4) {
            const f32x4 d = *reinterpret_cast<const f32x4*>(&sdz[oo * BMAX + 4 * b4]);
#pragma unroll
            for (int j = 0; j < 4; ++j) g = fmaf(d[j], xr[4 * b4 + j], g);
        }
        const size_t idx = (size_t)(o0 + oo) * I + i;
        const float nv = fmaf(mu, v[idx], g);
        v[idx] = nv;
        w[idx] = fmaf(-lr, nv, w[idx]);
    }
}

// bias: g[o] = sum_b dZ[b][o]; momentum-SGD
__global__ void k_fc_bgrad_sgd(const float* __restrict__ dz, float* __restrict__ bias, float* __restrict__ vb, int B, int O, float lr, float mu)
{
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= O) return;
    float g = 0.0f;
    for (int b = 0; b < B; ++b) g += dz[(size_t)b * O + o];
    const float nv = fmaf(mu, vb[o], g);
    vb[o] = nv;
    bias[o] = fmaf(-lr, nv, bias[o]);
}

// ---------------------------------------------------------------- convolution backward ---------

// data-gradient weights: wt[ci][kp][co] = wp[co][8 - kp][ci]   (ci < cin: the real input channels)
__global__ void k_pack_dgrad_w(const float* __restrict__ wp, float* __restrict__ wt, int Cout, int cin_pad, int Cin)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)Cin * 9 * Cout) return;
    const int co = (int)(idx % Cout);
    const int kp = (int)((idx / Cout) % 9);
    const int ci = (int)(idx / ((size_t)Cout * 9));
    wt[idx] = wp[((size_t)co * 9 + (8 - kp)) * cin_pad + ci];
}

struct WgradArgs {
    const float* dy;  // NHWC [B][H][W][Cout] (the gradient at the layer's pre-pool, post-ReLU output, ReLU mask applied)
    const float* x;   // NHWC [B][H][W][cin_pad] (the layer's input)
    float* slab;      // [S][Mpad][Npad]
    int B, H, Cout, cin_pad;
    int N, Npad, Mpad;  // N = 9*cin_pad
    long P;             // pixels = B*H*H
    int chunk;          // pixels per split (multiple of 16)
};

// (WM*64) output channels x (WN*64) (tap, input channel) columns per workgroup of WM*WN = 4 waves (64 x 64 per wave:
// 128 x 128, or 64 x 256 for the 64-channel layers), K = a run of `chunk` pixels, 16 per step.  LDS tiles are
// pixel-major ([k][m], [k][n], row stride = width + 32 floats: the two k rows of an MFMA fragment read fall in
// different bank halves), operands one float per lane for v_mfma_f32_32x32x2_f32.
template <int WM, int WN>
__global__ void __launch_bounds__(WM * WN * 64) k_conv_wgrad(WgradArgs a)
{
    constexpr int BK = 16, BM = WM * 64, BN = WN * 64, LSA = BM + 32, LSB = BN + 32, NT = WM * WN * 64;
    static_assert(NT % (BM / 4) == 0 && NT % (BN / 4) == 0, "a loader thread keeps one float4 column over its passes");
    constexpr int RA = NT / (BM / 4), RB = NT / (BN / 4);        // pixel rows one pass of the loader threads covers
    constexpr int A_PER = (BK + RA - 1) / RA, B_PER = (BK + RB - 1) / RB;  // float4 loads per thread and step (the last pass may be partial)
    __shared__ __attribute__((aligned(16))) float sA[2][BK * LSA], sB[2][BK * LSB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const long p0 = (long)blockIdx.z * a.chunk;
    const long p1 = p0 + a.chunk < a.P ? p0 + a.chunk : a.P;
    const int H = a.H, Cout = a.Cout, cpad = a.cin_pad;

    // loader roles: one float4 column per thread (fixed over the passes), pixel rows krow + R * i
    const int ca4 = (tid % (BM / 4)) * 4, krowa = tid / (BM / 4);
    const int cb4 = (tid % (BN / 4)) * 4, krowb = tid / (BN / 4);
    const bool mok = m0 + ca4 < Cout;
    const int n = n0 + cb4;
    const bool nok = n < a.N;
    const int kp = nok ? n / cpad : 0, ci = nok ? n - kp * cpad : 0;
    const int ky = kp / 3 - 1, kx = kp % 3 - 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f32x4 ra[A_PER], rb[B_PER];
    auto gload = [&](long pbase) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const long p = pbase + krowa + RA * i;
            ra[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (BK % RA != 0 && krowa + RA * i >= BK) continue;
            if (p < p1 && mok) ra[i] = *reinterpret_cast<const f32x4*>(a.dy + p * Cout + m0 + ca4);
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const long p = pbase + krowb + RB * i;
            rb[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (BK % RB != 0 && krowb + RB * i >= BK) continue;
            if (p < p1 && nok) {
                const int x = (int)(p % H);
                const long r = p / H;
                const int y = (int)(r % H);
                const int yy = y + ky, xx = x + kx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < H) rb[i] = *reinterpret_cast<const f32x4*>(a.x + (p + (long)ky * H + kx) * cpad + ci);
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
            if (BK % RA == 0 || krowa + RA * i < BK) *reinterpret_cast<f32x4*>(&sA[buf][(krowa + RA * i) * LSA + ca4]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PER; ++i)
            if (BK % RB == 0 || krowb + RB * i < BK) *reinterpret_cast<f32x4*>(&sB[buf][(krowb + RB * i) * LSB + cb4]) = rb[i];
    };

    const int r31 = lane & 31, hh = lane >> 5;
    const long steps = (p1 - p0 + BK - 1) / BK;
    if (steps > 0) {
        gload(p0);
        lstore(0);
    }
    __syncthreads();
    for (long t = 0; t < steps; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < steps) gload(p0 + (t + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            float fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = sA[buf][(2 * ks + hh) * LSA + (wm * 2 + i) * 32 + r31];
                fb[i] = sB[buf][(2 * ks + hh) * LSB + (wn * 2 + i) * 32 + r31];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < steps) lstore(buf ^ 1);
        __syncthreads();
    }

    // slab[s][m][n]: accumulator register r of lane (r31, hh) is row 8*(r/4) + 4*hh + (r%4), column r31
    float* slab = a.slab + (size_t)blockIdx.z * a.Mpad * a.Npad;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nn = n0 + (wn * 2 + j) * 32 + r31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = m0 + (wm * 2 + i) * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
                slab[(size_t)mm * a.Npad + nn] = acc[i][j][r];
            }
        }
}

// g = sum over the S slabs in a FIXED order; V = mu*V + g; W -= lr*V   (W, V: [M][N] = the packed conv weight layout).
// A block = 32 weights x 8 groups of consecutive slabs: a thread adds its group's slabs (four interleaved partial sums, so
// that four loads are in flight; combined as (a0 + a1) + (a2 + a3)), the eight group sums are added ascending.  One thread
// per weight walking all S slabs (round 1) was a chain of S dependent-latency loads on 9 216 ... 36 864 threads: 804 us
// for the first layer's 256 slabs, 267 us for conv1_2's.
__global__ void __launch_bounds__(256) k_wgrad_reduce_sgd(const float* __restrict__ slab, float* __restrict__ w, float* __restrict__ v,
                                                          int M, int N, int Mpad, int Npad, int S, float lr, float mu)
{
    __shared__ float part[8][32];
    const int o = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const size_t idx = (size_t)blockIdx.x * 32 + o;
    const bool ok = idx < (size_t)M * N;
    float acc = 0.0f;
    if (ok) {
        const int n = (int)(idx % N), m = (int)(idx / N);
        const int sg = (S + 7) / 8, s0 = grp * sg, s1 = s0 + sg < S ? s0 + sg : S;
        const size_t stride = (size_t)Mpad * Npad;
        const float* p = slab + ((size_t)s0 * Mpad + m) * Npad + n;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        int t = s0;
        for (; t + 3 < s1; t += 4) {
            a0 += p[0];
            a1 += p[stride];
            a2 += p[2 * stride];
            a3 += p[3 * stride];
            p += 4 * stride;
        }
        for (; t < s1; ++t) {
            a0 += p[0];
            p += stride;
        }
        acc = (a0 + a1) + (a2 + a3);
    }
    part[grp][o] = acc;
    __syncthreads();
    if (grp == 0 && ok) {
        float g = part[0][o];
#pragma unroll
        for (int k = 1; k < 8; ++k) g += part[k][o];
        const float nv = fmaf(mu, v[idx], g);
        v[idx] = nv;
        w[idx] = fmaf(-lr, nv, w[idx]);
    }
}

// bias gradient, pass 1: part[blk][co] = sum of dy[p][co] over the block's pixel range.  256 threads = (256 / cw)
// pixel rows x cw channels per pass over the channels; four independent partial sums per thread keep four loads
// in flight; the combination order is fixed (deterministic).
__global__ void __launch_bounds__(256) k_conv_bgrad_partial(const float* __restrict__ dy, float* __restrict__ part, long P, int Cout, long chunk)
{
    __shared__ float red[256];
    const long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
    for (int c0 = 0; c0 < Cout; c0 += 256) {
        const int cw = Cout - c0 < 256 ? Cout - c0 : 256;
        const int rows = 256 / cw;
        const int c = threadIdx.x % cw, r = threadIdx.x / cw;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        if (r < rows) {
            long p = p0 + r;
            for (; p + 3L * rows < p1; p += 4L * rows) {
                s0 += dy[p * Cout + c0 + c];
                s1 += dy[(p + rows) * Cout + c0 + c];
                s2 += dy[(p + 2L * rows) * Cout + c0 + c];
                s3 += dy[(p + 3L * rows) * Cout + c0 + c];
            }
            for (; p < p1; p += rows) s0 += dy[p * Cout + c0 + c];
        }
        float s = (s0 + s1) + (s2 + s3);
        red[threadIdx.x] = s;
        __syncthreads();
        if (r == 0) {
            for (int q = 1; q < rows; ++q) s += red[q * cw + c];
            part[(size_t)blockIdx.x * Cout + c0 + c] = s;
        }
        __syncthreads();
    }
}

// pass 2 + momentum SGD: 64 channels per workgroup, four 64-thread groups each add a quarter of the partials
__global__ void __launch_bounds__(256) k_conv_bgrad_sgd(const float* __restrict__ part, int nblk, float* __restrict__ bias, float* __restrict__ vb,
                                                        int Cout, float lr, float mu)
{
    __shared__ float red[3][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + col;
    float g = 0.0f;
    if (c < Cout)
        for (int i = grp; i < nblk; i += 4) g += part[(size_t)i * Cout + c];
    if (grp > 0) red[grp - 1][col] = g;
    __syncthreads();
    if (grp != 0 || c >= Cout) return;
    g = ((g + red[0][col]) + red[1][col]) + red[2][col];
    const float nv = fmaf(mu, vb[c], g);
    vb[c] = nv;
    bias[c] = fmaf(-lr, nv, bias[c]);
}

// ---------------------------------------------------------------- export / import --------------

// packed conv [Cout][9][cpad] -> OIHW [Cout][Cin][3][3]
__global__ void k_unpack_conv_w(const float* __restrict__ wp, float* __restrict__ w, int Cout, int Cin, int cpad)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)Cout * Cin * 9) return;
    const int kp = (int)(idx % 9);
    const int ci = (int)((idx / 9) % Cin);
    const int co = (int)(idx / ((size_t)Cin * 9));
    w[idx] = wp[((size_t)co * 9 + kp) * cpad + ci];
}
__global__ void k_repack_conv_w(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int cpad)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)Cout * 9 * cpad) return;
    const int ci = (int)(idx % cpad);
    const int kp = (int)((idx / cpad) % 9);
    const int co = (int)(idx / ((size_t)cpad * 9));
    wp[idx] = ci < Cin ? w[((size_t)co * Cin + ci) * 9 + kp] : 0.0f;
}
// FC1: [out][p*C + c] <-> [out][c*HW + p]
__global__ void k_unpack_fc1(const float* __restrict__ wp, float* __restrict__ w, int O, int C, int HW)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)O * C * HW) return;
    const int p = (int)(idx % HW);
    const int c = (int)((idx / HW) % C);
    const size_t o = idx / ((size_t)C * HW);
    w[idx] = wp[(o * HW + p) * C + c];
}
__global__ void k_repack_fc1(const float* __restrict__ w, float* __restrict__ wp, int O, int C, int HW)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)O * C * HW) return;
    const int c = (int)(idx % C);
    const int p = (int)((idx / C) % HW);
    const size_t o = idx / ((size_t)C * HW);
    wp[idx] = w[(o * C + c) * HW + p];
}

// ---------------------------------------------------------------- host side --------------------

struct WgradPlan {
    int BM, BN, Mpad, Npad, S, chunk;
    size_t slab_floats;
};

WgradPlan plan_wgrad(int B, int hw, int cout, int cin_pad)
{
    WgradPlan p;
    const int N = 9 * cin_pad;
    p.BM = cout <= 64 ? 64 : 128;  // 64 x 192 tiles (three waves) for the 64-channel layers, 128 x 128 otherwise:
    p.BN = cout <= 64 ? 192 : 128;  // conv1_2's 576 columns are three tiles (256-column tiles: 2.25), the first layer's 144 one
    p.Mpad = va_cdiv(cout, p.BM) * p.BM;
    p.Npad = va_cdiv(N, p.BN) * p.BN;
    const long P = (long)B * hw * hw;
    const int tiles = (p.Mpad / p.BM) * (p.Npad / p.BN);
    long S = va_cdiv(2048, tiles);  // aim for ~2048 workgroups
    const long maxS = P / 256 > 0 ? P / 256 : 1;
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    long chunk = (P + S - 1) / S;
    chunk = (chunk + 15) / 16 * 16;
    p.chunk = (int)chunk;
    p.S = (int)((P + chunk - 1) / chunk);
    p.slab_floats = (size_t)p.S * p.Mpad * p.Npad;
    return p;
}

struct TrainPlan {
    size_t x0, y[13], p[13], a_d[3], logits, dlogits, dd[3], da0, g[2], wt, slab, bpart, total;
};

constexpr int kBgradBlocks = 1024;

TrainPlan plan_train(const va_vgg16* m, int B)
{
    TrainPlan t{};
    size_t off = 0;
    auto take = [&](size_t floats) {
        const size_t o = off;
        off += va_align_up(floats * sizeof(float), 256);
        return o;
    };
    t.x0 = take((size_t)B * 224 * 224 * m->c_in_pad);
    size_t slab = 0;
    for (int i = 0; i < 13; ++i) {
        const ConvLayer& L = m->conv[i];
        t.y[i] = take((size_t)B * L.hw * L.hw * L.cout);
        t.p[i] = L.pool ? take((size_t)B * (L.hw / 2) * (L.hw / 2) * L.cout) : 0;
        const WgradPlan w = plan_wgrad(B, L.hw, L.cout, L.cin_pad);
        if (w.slab_floats > slab) slab = w.slab_floats;
    }
    for (int i = 0; i < 4; ++i) {
        const size_t s = va_fc_slab_floats(B, m->fc_out[i], m->fc_in[i]);
        if (s > slab) slab = s;
    }
    t.a_d[0] = take((size_t)B * 4096);
    t.a_d[1] = take((size_t)B * 4096);
    t.a_d[2] = take((size_t)B * m->desc_dim);
    t.logits = take((size_t)B * m->n_classes);
    t.dlogits = take((size_t)B * m->n_classes);
    t.dd[0] = take((size_t)B * 4096);
    t.dd[1] = take((size_t)B * 4096);
    t.dd[2] = take((size_t)B * m->desc_dim);
    t.da0 = take((size_t)B * 512 * 49);
    t.g[0] = take((size_t)B * 224 * 224 * 64);
    t.g[1] = take((size_t)B * 224 * 224 * 64);
    t.wt = take((size_t)512 * 9 * 512);
    t.slab = take(slab);
    t.bpart = take((size_t)kBgradBlocks * 512);
    t.total = off;
    return t;
}

void keys_for(unsigned long long seed, unsigned stream, unsigned& key, unsigned& key2)
{
    auto mix = [](unsigned x) {
        x ^= x >> 16;
        x *= 0x7FEB352Du;
        x ^= x >> 15;
        x *= 0x846CA68Bu;
        x ^= x >> 16;
        return x;
    };
    key = mix((unsigned)(seed * 0x9E3779B1ull + (unsigned long long)stream * 0x85EBCA77ull + 0x1234567ull));
    key2 = mix(key + 0x68E31DA4u);
}

template <typename F>
int fc_backward_dispatch(int B, F&& f)
{
    if (B <= 32) return f(std::integral_constant<int, 32>());
    return f(std::integral_constant<int, 64>());
}

}  // namespace

extern "C" int va_vgg16_train_init(va_vgg16* m, void* stream)
{
    VA_CHECK_ARG(m != nullptr, "va_vgg16_train_init: model is NULL");
    VA_USE_DEVICE(m->ctx);
    VA_CHECK_ARG(m->dtype == VA_DTYPE_F32, "va_vgg16_train_init: training is fp32 only");
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < 13; ++i) {
        ConvLayer& L = m->conv[i];
        const size_t nw = (size_t)L.cout * 9 * L.cin_pad;
        if (!L.mom_w) VA_HIP(hipMalloc(&L.mom_w, nw * sizeof(float)));
        if (!L.mom_b) VA_HIP(hipMalloc(&L.mom_b, L.cout * sizeof(float)));
        VA_HIP(hipMemsetAsync(L.mom_w, 0, nw * sizeof(float), st));
        VA_HIP(hipMemsetAsync(L.mom_b, 0, L.cout * sizeof(float), st));
    }
    for (int i = 0; i < 4; ++i) {
        const size_t nw = (size_t)m->fc_out[i] * m->fc_in[i];
        if (!m->fc_mom_w[i]) VA_HIP(hipMalloc(&m->fc_mom_w[i], nw * sizeof(float)));
        if (!m->fc_mom_b[i]) VA_HIP(hipMalloc(&m->fc_mom_b[i], m->fc_out[i] * sizeof(float)));
        VA_HIP(hipMemsetAsync(m->fc_mom_w[i], 0, nw * sizeof(float), st));
        VA_HIP(hipMemsetAsync(m->fc_mom_b[i], 0, m->fc_out[i] * sizeof(float), st));
    }
    if (!m->zeros_f32) VA_HIP(hipMalloc(&m->zeros_f32, 512 * sizeof(float)));
    VA_HIP(hipMemsetAsync(m->zeros_f32, 0, 512 * sizeof(float), st));
    return VA_OK;
}

extern "C" size_t va_vgg16_train_workspace_bytes(const va_vgg16* m, int batch)
{
    if (!m || batch < 1 || batch > 64 || m->dtype != VA_DTYPE_F32) return 0;
    return plan_train(m, batch).total;
}

extern "C" int va_vgg16_train_step(va_vgg16* m, const void* x, int x_is_u8, const void* labels, int batch, float lr, float momentum,
                                   unsigned long long dropout_seed, void* desc, void* loss_out, void* workspace, size_t workspace_bytes,
                                   void* stream)
{
    VA_CHECK_ARG(m != nullptr && x != nullptr && labels != nullptr && loss_out != nullptr, "va_vgg16_train_step: NULL argument");
    VA_USE_DEVICE(m->ctx);
    VA_CHECK_ARG(m->dtype == VA_DTYPE_F32, "va_vgg16_train_step: training is fp32 only");
    VA_CHECK_ARG(batch >= 1 && batch <= 64, "va_vgg16_train_step: batch %d out of range [1,64]", batch);
    VA_CHECK_ARG(m->conv[0].mom_w != nullptr && m->zeros_f32 != nullptr, "va_vgg16_train_step: call va_vgg16_train_init first");
    VA_CHECK_ARG(!x_is_u8 || (m->in_mean && m->in_std), "va_vgg16_train_step: u8 input needs the model's mean/std");
    const TrainPlan T = plan_train(m, batch);
    if (workspace == nullptr || workspace_bytes < T.total) {
        va_set_error("va_vgg16_train_step: workspace of %zu bytes needed, %zu given", T.total, workspace_bytes);
        return VA_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int B = batch;
    char* ws = (char*)workspace;
    auto F = [&](size_t off) { return (float*)(ws + off); };
    float* slab = F(T.slab);

    // ---------------- forward (train mode) ----------------
    if (int rc = va_input_to_nhwc_f32(m, x, x_is_u8, B, F(T.x0), st)) return rc;
    const float* in = F(T.x0);
    for (int i = 0; i < 13; ++i) {
        const ConvLayer& L = m->conv[i];
        if (int rc = va_conv3x3_f32(L.hw, L.cin_pad, L.cout, L.wp, L.bias, in, F(T.y[i]), nullptr, 0, 0, B, m->zeros_f32, m->f32_conv, st)) return rc;
        in = F(T.y[i]);
        if (L.pool) {
            const size_t n = (size_t)B * (L.hw / 2) * (L.hw / 2) * (L.cout / 4);
            k_maxpool<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(F(T.y[i]), F(T.p[i]), B, L.hw, L.cout);
            in = F(T.p[i]);
        }
    }
    VA_LAUNCH_CHECK();
    const float* a0 = F(T.p[12]);  // NHWC flatten [B][7*7*512]; FC1's weights are stored in that order
    const float* fin[4] = {a0, F(T.a_d[0]), F(T.a_d[1]), F(T.a_d[2])};
    float* fout[4] = {F(T.a_d[0]), F(T.a_d[1]), F(T.a_d[2]), F(T.logits)};
    for (int l = 0; l < 4; ++l) {
        if (int rc = va_fc_f32(fin[l], m->fcw[l], m->fcb[l], fout[l], slab, B, m->fc_out[l], m->fc_in[l], l < 3, st)) return rc;
        if (l < 3) {
            unsigned key, key2;
            keys_for(dropout_seed, 100u + (unsigned)l, key, key2);
            const size_t n = (size_t)B * m->fc_out[l];
            k_dropout<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(fout[l], n, key, key2);
        }
    }
    if (desc) VA_HIP(hipMemcpyAsync(desc, F(T.a_d[2]), (size_t)B * m->desc_dim * sizeof(float), hipMemcpyDeviceToDevice, st));
    k_ce_fwd_bwd<<<1, 256, 0, st>>>(F(T.logits), (const long long*)labels, B, m->n_classes, F(T.dlogits), (float*)loss_out);
    VA_LAUNCH_CHECK();

    // ---------------- classifier backward + update ----------------
    // dz_l = gradient at FC l's pre-activation; for l < 3 it arrives through Dropout and ReLU: x2 where the
    // (post-dropout) activation is positive, 0 elsewhere -- applied by the k_fc_dx that produced it.
    const float* dz[4] = {F(T.dd[0]), F(T.dd[1]), F(T.dd[2]), F(T.dlogits)};
    float* dxo[4] = {F(T.da0), F(T.dd[0]), F(T.dd[1]), F(T.dd[2])};
    for (int l = 3; l >= 0; --l) {
        const int O = m->fc_out[l], I = m->fc_in[l];
        const float* mask = l > 0 ? fin[l] : nullptr;  // fin[l] = post-dropout activation of layer l-1
        int rc = fc_backward_dispatch(B, [&](auto BM) {
            constexpr int bm = decltype(BM)::value;
            k_fc_dx<bm><<<va_cdiv(I, 64), 256, 0, st>>>(dz[l], m->fcw[l], dxo[l], B, O, I, mask, 2.0f);
            const int orows = 16;
            k_fc_wgrad_sgd<bm><<<dim3(va_cdiv(I, 256), va_cdiv(O, orows)), 256, (size_t)orows * bm * sizeof(float), st>>>(
                dz[l], fin[l], m->fcw[l], m->fc_mom_w[l], B, O, I, orows, lr, momentum);
            return VA_OK;
        });
        if (rc) return rc;
        k_fc_bgrad_sgd<<<va_cdiv(O, 256), 256, 0, st>>>(dz[l], m->fcb[l], m->fc_mom_b[l], B, O, lr, momentum);
    }
    VA_LAUNCH_CHECK();

    // ---------------- feature stack backward + update ----------------
    const int stop_at = m->train_stop_at;  // VA_OPT_TRAIN_STOP_AT (tests): -1 = the whole step
    const float* dout = F(T.da0);  // gradient at layer 12's pooled output
    int cur = -1;                  // which of the two gradient buffers holds `dout` (-1: neither)
    for (int i = 12; i >= 0; --i) {
        ConvLayer& L = m->conv[i];
        const float* lin = i == 0 ? F(T.x0) : (m->conv[i - 1].pool ? F(T.p[i - 1]) : F(T.y[i - 1]));
        const float* dyr = dout;
        if (L.pool) {
            const int dst = cur == 0 ? 1 : 0;
            const size_t n = (size_t)B * (L.hw / 2) * (L.hw / 2) * (L.cout / 4);
            k_unpool<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(dout, F(T.y[i]), F(T.p[i]), F(T.g[dst]), B, L.hw, L.cout);
            dyr = F(T.g[dst]);
            cur = dst;
        }
        // data gradient first (it needs this step's weights), into the other gradient buffer
        if (i > 0) {
            const int cin = L.cin;  // = previous layer's cout
            const size_t nwt = (size_t)cin * 9 * L.cout;
            k_pack_dgrad_w<<<(unsigned)((nwt + 255) / 256), 256, 0, st>>>(L.wp, F(T.wt), L.cout, L.cin_pad, cin);
            float* g = F(T.g[1 - cur]);
            const float* mask = m->conv[i - 1].pool ? nullptr : F(T.y[i - 1]);
            if (int rc = va_conv3x3_f32(L.hw, L.cout, cin, F(T.wt), m->zeros_f32, dyr, g, mask, 1, 0, B, m->zeros_f32, m->f32_conv, st)) return rc;
            dout = g;
        }
        // weight and bias gradients + update (dyr stays intact: the data gradient went to the other buffer)
        const WgradPlan wp = plan_wgrad(B, L.hw, L.cout, L.cin_pad);
        WgradArgs a{};
        a.dy = dyr;
        a.x = lin;
        a.slab = slab;
        a.B = B;
        a.H = L.hw;
        a.Cout = L.cout;
        a.cin_pad = L.cin_pad;
        a.N = 9 * L.cin_pad;
        a.Npad = wp.Npad;
        a.Mpad = wp.Mpad;
        a.P = (long)B * L.hw * L.hw;
        a.chunk = wp.chunk;
        if (wp.BM == 64) k_conv_wgrad<1, 3><<<dim3(wp.Npad / 192, wp.Mpad / 64, wp.S), 192, 0, st>>>(a);
        else k_conv_wgrad<2, 2><<<dim3(wp.Npad / 128, wp.Mpad / 128, wp.S), 256, 0, st>>>(a);
        const size_t nw = (size_t)L.cout * a.N;
        k_wgrad_reduce_sgd<<<(unsigned)((nw + 31) / 32), 256, 0, st>>>(slab, L.wp, L.mom_w, L.cout, a.N, wp.Mpad, wp.Npad, wp.S, lr, momentum);
        const long bchunk = (a.P + kBgradBlocks - 1) / kBgradBlocks;
        const int nblk = (int)((a.P + bchunk - 1) / bchunk);
        k_conv_bgrad_partial<<<nblk, 256, 0, st>>>(dyr, F(T.bpart), a.P, L.cout, bchunk);
        k_conv_bgrad_sgd<<<va_cdiv(L.cout, 64), 256, 0, st>>>(F(T.bpart), nblk, L.bias, L.mom_b, L.cout, lr, momentum);
        if (i > 0) cur = 1 - cur;
        VA_LAUNCH_CHECK();
        if (stop_at == i) {  // debugging aid of the tests: leave the gradient buffers as layer i left them; NOT a completed step
            va_set_error("va_vgg16_train_step: stopped after the backward pass of conv layer %d (VA_OPT_TRAIN_STOP_AT); layers below were not updated", i);
            return VA_ERR_STOPPED;
        }
    }
    return VA_OK;
}

// which = 0: parameters, 1: momentum buffers.  Destination tensors in the reference's layouts (conv OIHW
// [cout][cin][3][3], fc [out][in] with FC1's input CHW-major): what model.state_dict() / optimizer.state_dict() hold.
extern "C" int va_vgg16_export_state(va_vgg16* m, int which, void* const* conv_w, void* const* conv_b, void* const* fc_w,
                                     void* const* fc_b, void* stream)
{
    VA_CHECK_ARG(m != nullptr && conv_w && conv_b && fc_w && fc_b, "va_vgg16_export_state: NULL argument");
    VA_USE_DEVICE(m->ctx);
    VA_CHECK_ARG(m->dtype == VA_DTYPE_F32, "va_vgg16_export_state: fp32 models only");
    VA_CHECK_ARG(which == 0 || (which == 1 && m->conv[0].mom_w), "va_vgg16_export_state: which must be 0, or 1 after va_vgg16_train_init");
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < 13; ++i) {
        const ConvLayer& L = m->conv[i];
        const size_t n = (size_t)L.cout * L.cin * 9;
        k_unpack_conv_w<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(which ? L.mom_w : L.wp, (float*)conv_w[i], L.cout, L.cin, L.cin_pad);
        VA_HIP(hipMemcpyAsync(conv_b[i], which ? L.mom_b : L.bias, L.cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    for (int i = 0; i < 4; ++i) {
        const size_t n = (size_t)m->fc_out[i] * m->fc_in[i];
        const float* src = which ? m->fc_mom_w[i] : m->fcw[i];
        if (i == 0) k_unpack_fc1<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(src, (float*)fc_w[0], m->fc_out[0], 512, 49);
        else VA_HIP(hipMemcpyAsync(fc_w[i], src, n * sizeof(float), hipMemcpyDeviceToDevice, st));
        VA_HIP(hipMemcpyAsync(fc_b[i], which ? m->fc_mom_b[i] : m->fcb[i], m->fc_out[i] * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    VA_LAUNCH_CHECK();
    return VA_OK;
}

// The inverse: load parameters (which = 0) or momentum buffers (which = 1) from tensors in the reference's layouts.
extern "C" int va_vgg16_import_state(va_vgg16* m, int which, const void* const* conv_w, const void* const* conv_b,
                                     const void* const* fc_w, const void* const* fc_b, void* stream)
{
    VA_CHECK_ARG(m != nullptr && conv_w && conv_b && fc_w && fc_b, "va_vgg16_import_state: NULL argument");
    VA_USE_DEVICE(m->ctx);
    VA_CHECK_ARG(m->dtype == VA_DTYPE_F32, "va_vgg16_import_state: fp32 models only");
    VA_CHECK_ARG(which == 0 || (which == 1 && m->conv[0].mom_w), "va_vgg16_import_state: which must be 0, or 1 after va_vgg16_train_init");
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < 13; ++i) {
        ConvLayer& L = m->conv[i];
        const size_t n = (size_t)L.cout * 9 * L.cin_pad;
        k_repack_conv_w<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((const float*)conv_w[i], which ? L.mom_w : L.wp, L.cout, L.cin, L.cin_pad);
        VA_HIP(hipMemcpyAsync(which ? L.mom_b : L.bias, conv_b[i], L.cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    for (int i = 0; i < 4; ++i) {
        const size_t n = (size_t)m->fc_out[i] * m->fc_in[i];
        float* dst = which ? m->fc_mom_w[i] : m->fcw[i];
        if (i == 0) k_repack_fc1<<<(unsigned)((n + 255) / 256), 256, 0, st>>>((const float*)fc_w[0], dst, m->fc_out[0], 512, 49);
        else VA_HIP(hipMemcpyAsync(dst, fc_w[i], n * sizeof(float), hipMemcpyDeviceToDevice, st));
        VA_HIP(hipMemcpyAsync(which ? m->fc_mom_b[i] : m->fcb[i], fc_b[i], m->fc_out[i] * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    VA_LAUNCH_CHECK();
    return VA_OK;
}

// Debugging aid (tests): byte offsets inside the training workspace: out[0..12] = y[i], out[13..25] = p[i] (0 when
// the layer has no pool), out[26] = g[0], out[27] = g[1], out[28] = da0, out[29] = x0.
extern "C" int va_vgg16_train_plan(const va_vgg16* m, int batch, unsigned long long* out)
{
    VA_CHECK_ARG(m != nullptr && out != nullptr && batch >= 1 && batch <= 64, "va_vgg16_train_plan: bad argument");
    VA_USE_DEVICE(m->ctx);
    const TrainPlan T = plan_train(m, batch);
    for (int i = 0; i < 13; ++i) {
        out[i] = T.y[i];
        out[13 + i] = T.p[i];
    }
    out[26] = T.g[0];
    out[27] = T.g[1];
    out[28] = T.da0;
    out[29] = T.x0;
    return VA_OK;
}
