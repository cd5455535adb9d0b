// Internal definitions shared by vgg.hip (inference) and train.hip (training step).
#pragma once
#include "va_internal.h"

struct ConvLayer {
    int cin, cin_pad, cout, hw;  // hw = input height = width
    bool pool;
    bool xcol;      // bf16 first layer: x taps folded into the channels (3 K steps)
    float* wp;      // f32 [Cout][9*cin_pad]  (VA_DTYPE_F32)
    __bf16* wp_bf;  // bf16 [Cout][9*cin_pad] (VA_DTYPE_BF16)
    float* bias;
    float* mom_w;   // momentum buffers of the training step (same layouts; NULL until the first step)
    float* mom_b;
};


struct va_vgg16 {
    va_ctx* ctx;
    int c_in, c_in_pad, n_classes, desc_dim, dtype;
    ConvLayer conv[13];
    float* fcw[4];
    float* fcb[4];
    int fc_in[4], fc_out[4];
    float* in_mean;  // device [c_in] or NULL
    float* in_std;
    __bf16* zeros;   // 256 zero bytes (VA_DTYPE_BF16: source of the out-of-image taps)
    float* fc_mom_w[4];  // momentum buffers of the classifier (training)
    float* fc_mom_b[4];
    float* zeros_f32;    // >= 512 zero floats (training: bias of the linear dgrad convolutions)
    // va_vgg16_set_option (explicit A/B and test switches; nothing is read from the environment)
    __bf16* wp_f1;      // bf16 first layer, fused kernel: [64][3][f1_krow] (NULL unless the first layer qualifies)
    int f1_cp, f1_krow; // channels per pixel of its LDS patch (c_in rounded up to 4), elements per kernel row (16-element blocks)
    int bf16_first;     // VA_OPT_BF16_FIRST_LAYER: 1 = fused first layer (default), 0 = input conversion + 3-step convolution
    int bf16_variant;   // VA_OPT_BF16_VARIANT: 0 = automatic tile/staging choice, 1 = 64-channel tiles + single buffer, 2 = DMA ring everywhere
    int f32_conv;       // VA_OPT_F32_CONV_KERNEL: 1 = LDS-DMA kernel where Cin % 32 == 0 (default), 0 = register-staged kernel
    int train_stop_at;  // VA_OPT_TRAIN_STOP_AT: -1 = full step; i = va_vgg16_train_step returns VA_ERR_STOPPED after conv layer i's backward
};


// fp32 3x3 convolution (implicit GEMM on MFMA) of an NHWC tensor with explicit packed weights [cout][9*cin_pad]:
// out = conv(in) + bias, then ReLU unless `linear`, then zeroed where mask[same index] <= 0 (mask may be NULL;
// only without pooling), then 2x2 max-pooled when `pool`.  zeros: the model's zeros_f32 (>= 512 zero floats).
int va_conv3x3_f32(int hw, int cin_pad, int cout, const float* wp, const float* bias, const float* in, float* out,
                   const float* mask, int linear, int pool, int B, const float* zeros, int f32_conv, hipStream_t st);
// out[M][N] = A[M][K] . Wt[N][K]^T + bias (+ReLU); slab: >= va_fc_slab_floats(M, N, K) floats of scratch
int va_fc_f32(const float* A, const float* Wt, const float* bias, float* out, float* slab, int M, int N, int K, int relu, hipStream_t st);
size_t va_fc_slab_floats(int M, int N, int K);
// x NCHW (f32 or u8 + ToTensor/Normalize) -> NHWC f32 with c_in_pad channels
int va_input_to_nhwc_f32(const va_vgg16* m, const void* x, int x_is_u8, int B, float* out, hipStream_t st);
