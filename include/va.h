/*
 * va.h -- C ABI of the MI355X-native two-stream inference hot path
 *         (libva_hip.so, built from video_analytics_amd/csrc/ with hipcc for gfx950).
 *
 * The reference (arindamrc/video_analytics, Sheet03) has no FFI of its own: its hot path
 * is reached through three Python call surfaces.  Each entry point
 * below names the reference interface it stands behind:
 *
 *   va_vgg16_*        self.features(ip) (forward with only `feat` requested) + the classifierList
 *                     traversal (va_vgg16_classify) inside validate():
 *                     Sheet03/spatialModel.py:110-113,127-129,136-152,212-218 and the
 *                     temporal twin Sheet03/temporalModel.py:122-126,140-142,165-181,241-247.
 *   va_copy_first_layer   TemporalNetwork.__copyFirstLayer__: Sheet03/temporalModel.py:149-162.
 *   va_tvl1_flow      the upstream tool that wrote the flow_x_%04d.jpg / flow_y_%04d.jpg files
 *                     TemporalDataset reads: Sheet03/temporalModel.py:76-81,
 *                     Sheet03/parameters.py:27,38-39 (no reference function exists; the
 *                     algorithm is the published TV-L1).
 *   va_flow_to_stack  the 8-bit flow image + getTransforms' ToTensor/Normalize + the x/y
 *                     interleave of TemporalDataset.__getitem__: Sheet03/temporalModel.py:83-90,
 *                     Sheet03/utils.py:148-150.
 *   va_validate_batch the loss / argmax / correct-count lines of validate():
 *                     Sheet03/spatialModel.py:219-221.
 *   va_meter_*        the per-video AverageMeter collation of validate():
 *                     Sheet03/utils.py:154-171, Sheet03/spatialModel.py:223-228.
 *   va_linear_svm_predict   LinearSVC.predict on the joined descriptors:
 *                     Sheet03/combinedModel.py:38.
 *   va_vgg16_train_*  the batch-loop body of train(): Sheet03/spatialModel.py:165-182;
 *   va_vgg16_export/import_state   checkpoint contents: Sheet03/spatialModel.py:234-260.
 *
 * Conventions
 *   - return 0 (VA_OK) or an error code; va_last_error() returns a thread-local message.
 *   - every data pointer is a DEVICE pointer (hipMalloc'ed; torch tensor.data_ptr()) unless
 *     a parameter comment says "host".  The caller owns all inputs, outputs and workspaces;
 *     the library owns only va_ctx and the packed-weight handle.  No hidden allocation and
 *     no host synchronisation inside va_vgg16_forward / va_tvl1_flow / va_flow_to_stack:
 *     all work is enqueued on `stream` (a hipStream_t; NULL = the null stream).
 *   - a handle may be used by one thread at a time; one process per GPU.  Every entry point selects its
 *     context's device itself (hipSetDevice); `stream` and all pointers must belong to that device.
 *   - the library reads NO environment variable and keeps no process-global switch: every tuning or test
 *     knob is a field of va_tvl1_params or a va_vgg16_set_option value of one handle.
 *   - there is NO CPU fallback in this library.
 */
#ifndef VA_H
#define VA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VA_OK 0
#define VA_ERR_INVALID 1   /* bad argument / shape: Python wrapper raises ValueError */
#define VA_ERR_HIP 2       /* HIP runtime failure: RuntimeError */
#define VA_ERR_WORKSPACE 3 /* workspace too small: ValueError */
#define VA_ERR_STOPPED 4   /* a test switch cut the call short (VA_OPT_TRAIN_STOP_AT): RuntimeError in production code */

#define VA_DTYPE_F32 0
#define VA_DTYPE_BF16 1

typedef struct va_ctx va_ctx;
typedef struct va_vgg16 va_vgg16;

/* version number (currently 3) + VA_VERSION_EXPERIMENTS when the library was built with -DVA_EXPERIMENTS, i.e. when it
 * also holds the measured-slower kernel families behind va_tvl1_params.tuning / VA_OPT_BF16_VARIANT 6 */
#define VA_VERSION_EXPERIMENTS 0x10000
int va_version(void);
const char* va_last_error(void);
int va_ctx_create(int device, va_ctx** out);
void va_ctx_destroy(va_ctx* ctx);

/* ------------------------------------------------------------------ VGG-16 streams --- */

/*
 * Pack weights once.  conv_w[13]: f32 OIHW [Cout][Cin][3][3]; conv_b[13]: f32 [Cout];
 * fc_w[4]: f32 [out][in] with FC1's input index in the reference's flatten order
 * c*49 + h*7 + w (Sheet03/spatialModel.py:213); fc_b[4]: f32 [out].
 * c_in is 3 (spatial) or 2L (temporal, 20).  in_mean/in_std: HOST arrays of c_in floats used
 * only when va_vgg16_forward is given u8 input (ToTensor+Normalize, Sheet03/utils.py:148-150);
 * may be NULL.  dtype: VA_DTYPE_F32 (fp32 in, fp32 MFMA accumulate: the parity configuration) or
 * VA_DTYPE_BF16 (bf16 activations and conv weights on the bf16 MFMA, fp32 accumulate, fp32 classifier:
 * the throughput configuration of BASELINE config 5; class scores deviate at the 1e-2 level).
 * The call synchronises `stream` before returning (the source tensors may be freed).
 */
int va_vgg16_create(va_ctx* ctx, int c_in, int n_classes, int desc_dim, int dtype,
                    const void* const* conv_w, const void* const* conv_b,
                    const void* const* fc_w, const void* const* fc_b,
                    const float* in_mean, const float* in_std,
                    void* stream, va_vgg16** out);
void va_vgg16_destroy(va_vgg16* model);
size_t va_vgg16_workspace_bytes(const va_vgg16* model, int batch);

/*
 * A/B and test switches of ONE model handle (defaults are the measured choices of DESIGN.md; results of the
 * fp32 path do not depend on VA_OPT_F32_CONV_KERNEL):
 *   VA_OPT_BF16_VARIANT     0 (default) kernel, tile and staging scheme chosen per layer (measured); 1 = 64-channel
 *                           tiles with one LDS buffer on every layer; 2 = the LDS-DMA ring on every layer; 5 = the
 *                           two-group kernel (k_conv3x3_pp_bf16) on every layer with >= 128 output channels and >= 28x28
 *                           pixels (the default uses it on the 28x28 layers); 0, 1, 2, 5 and 7 add the same products in the
 *                           same order and agree bit for bit.  6 = the two-group kernel on halo bricks
 *                           (k_conv3x3_bpp_bf16; chunk-major K order: another fp32 summation order, so within bf16 noise
 *                           of the others; measured no faster).  7 = the weights-resident kernel (k_conv3x3_ws_bf16) on
 *                           both layers with 64 input channels (the default uses it on conv1_2 only); bit-equal to 0, 1, 2, 5.
 *                           (3 and 4, round 2's first halo-brick kernel, are gone.)
 *   VA_OPT_BF16_FIRST_LAYER 1 (default) the first layer reads the NCHW input itself (k_conv1_fused_bf16); 0 = the input is
 *                           first staged as a 64-channel NHWC tensor and convolved in three K steps (the round-1 path;
 *                           another fp32 summation order: bf16-level agreement).  Independent of VA_OPT_BF16_VARIANT
 *   VA_OPT_F32_CONV_KERNEL  1 (default) LDS-DMA staged fp32 kernel where Cin % 32 == 0; 0 = register-staged kernel
 *   VA_OPT_TRAIN_STOP_AT    -1 (default) full training step; i in [0,12]: va_vgg16_train_step returns
 *                           VA_ERR_STOPPED after the backward pass of conv layer i, leaving the gradient buffers
 *                           as that layer left them and the layers below WITHOUT their update (tests only)
 */
#define VA_OPT_BF16_VARIANT 1
#define VA_OPT_F32_CONV_KERNEL 2
#define VA_OPT_TRAIN_STOP_AT 3
#define VA_OPT_BF16_FIRST_LAYER 4
int va_vgg16_set_option(va_vgg16* model, int option, int value);

/*
 * x: f32 (x_is_u8 = 0) or u8 (x_is_u8 = 1) [batch][c_in][224][224] NCHW.
 * feat: f32 [batch][512][7][7] NCHW or NULL; desc: f32 [batch][desc_dim] (post-ReLU output of
 * classifier index 8) or NULL; logits: f32 [batch][n_classes] (no softmax) or NULL.
 * VA_DTYPE_BF16 models: batch <= 334 (a layer's activations are addressed with 32-bit byte offsets; VA_ERR_INVALID
 * beyond that: split the batch).
 */
int va_vgg16_forward(va_vgg16* model, const void* x, int x_is_u8, int batch,
                     void* feat, void* desc, void* logits,
                     void* workspace, size_t workspace_bytes, void* stream);

/*
 * The classifier traversal alone (Sheet03/spatialModel.py:213-218): feat f32 [batch][512][7][7]
 * NCHW (as returned through `feat` above; flattened CHW-major) -> desc / logits (either may be
 * NULL).  va_vgg16_forward(x) == va_vgg16_classify(features(x)).
 */
int va_vgg16_classify(va_vgg16* model, const void* feat, int batch, void* desc, void* logits,
                      void* workspace, size_t workspace_bytes, void* stream);

/* w_rgb f32 [cout][3][3][3] -> w_out f32 [cout][n_in][3][3]: mean over the 3 input channels,
 * accumulated in channel order then divided by 3, replicated n_in times. */
int va_copy_first_layer(va_ctx* ctx, const void* w_rgb, int cout, int n_in, void* w_out, void* stream);

/* logits f32 [batch][n_classes], labels i64 [batch] -> out f32[2] (device):
 * out[0] = mean cross-entropy over the batch, out[1] = number of argmax(first max)==label. */
int va_validate_batch(va_ctx* ctx, const void* logits, const void* labels, int batch, int n_classes,
                      void* out, void* stream);

/* ------------------------------------------------------------------ TV-L1 flow ------- */

typedef struct va_tvl1_params {
    float tau;        /* 0.25 */
    float lambda;     /* 0.15 */
    float theta;      /* 0.3  */
    int nscales;      /* 5    */
    int warps;        /* 5    */
    float epsilon;    /* 0.01; <= 0: run exactly `iters` inner iterations per warp */
    int iters;        /* 300  */
    float scale_step; /* 0.8  */
    int block_iters;  /* inner iterations fused per launch (register-resident temporal
                         blocking); 0 = library default.  Results do not depend on it.
                         Forced to 1 when epsilon > 0. */
    int fast_math;    /* 0 (default): the exact arithmetic contract of DESIGN.md (correctly rounded
                         sqrt and division): results bit-identical to the CPU oracle.
                         1: the two special functions of the dual update use the 1-ulp hardware
                         v_sqrt_f32 / v_rcp_f32 (measured: 11 % more clips/s; flow within 7e-6 px on average
                         of the exact mode, 99.9 % of the pixels within 1e-3 px). */
    int tile_mask;    /* tuning/testing: bit i allows register-tile candidate i of the inner-iteration
                         kernel (bits 0-3: 256x32, 128x64, 84x96, 64x128 pixels, one 8-wave workgroup
                         per CU; bits 4-7: 256x16, 128x32, 84x48, 64x64, two 4-wave workgroups per
                         CU).  Bit 8 (256): iterate every level with the streaming kernel (a wave carries a
                         128-column strip row by row through 10 iterations per pass) instead of the
                         register tiles; fixed-iteration mode only.  Bit 9 (512): iterate every level that fits one
                         strip (<= 256 columns) with the persistent row pipeline k_iter_rows, the others with the
                         streaming kernel.  Bit 10 (1024): the streaming kernel never lets the narrow last strips of two
                         pairs share a wave (A/B switch).  0 (default) = library choice per level.  Results do not depend on it. */
    int tuning[8];    /* the library's own tuning / experiment switches (which kernel iterates which pyramid level, chunking of
                         rows, pipeline shapes: named in csrc/va_internal.h, VA_TUNE_*); va_tvl1_default_params fills in the
                         defaults (-1, 0, 0, 0, -1, 0, 0, 0) and callers leave them alone.  Results do not depend on any
                         of them; the library reads no environment variable.  Several values select kernels that are
                         only compiled with -DVA_EXPERIMENTS (measured slower, kept reproducible: DESIGN.md section 7);
                         a default build rejects those with VA_ERR_INVALID. */
} va_tvl1_params;

void va_tvl1_default_params(va_tvl1_params* p);

/* Number of pyramid levels actually used and their sizes (ws/hs: HOST arrays of >= 16 ints). */
int va_tvl1_pyramid_sizes(int w, int h, const va_tvl1_params* p, int* ws, int* hs);

/* The register tiling va_tvl1_flow will use, level by level (host logic; HOST array of >= 6*16 ints): per level
 * { tile width, tile height, waves per workgroup, block depth K, tiles in x, tiles in y }; a level that streams
 * reports { strip width (64 x pixels per lane: 128 or 192), 0 (rows stream through), waves of the row pipeline (2 or 1), iterations per pass
 * (16 with two waves, 10 with one), strips in x, 0 (chunks of rows: chosen per call from the number of pairs) }.
 * Returns the number of levels (0 on bad arguments). */
int va_tvl1_tile_plan(int w, int h, const va_tvl1_params* p, int* out);

size_t va_tvl1_workspace_bytes(int w, int h, int n_seq, int frames_per_seq, const va_tvl1_params* p);

/*
 * frames: u8 (frames_are_u8 = 1) or f32 in [0,255] [n_seq][frames_per_seq][h][w] gray.
 * flow:   f32 [n_seq*(frames_per_seq-1)][2][h][w]; plane 0 = x flow, plane 1 = y flow of the
 *         pair (frame k, frame k+1) of each sequence.
 */
int va_tvl1_flow(va_ctx* ctx, const void* frames, int frames_are_u8, int n_seq, int frames_per_seq,
                 int w, int h, const va_tvl1_params* p, void* flow,
                 void* workspace, size_t workspace_bytes, void* stream);

/*
 * flow f32 [n_pairs][2][h][w] -> stack f32 [2*n_pairs][h][w]: channel 2k = x flow of pair k,
 * 2k+1 = y flow (Sheet03/temporalModel.py:83).  Each value is quantised to the 8-bit flow
 * image convention q = rint(clamp(255*(v+bound)/(2*bound), 0, 255)) and then normalised as
 * (q/255 - mean)/std (Sheet03/utils.py:148-150; single-channel rule: mean 0.485, std 0.229).
 */
int va_flow_to_stack(va_ctx* ctx, const void* flow, int n_pairs, int w, int h,
                     float bound, float mean, float stdv, void* stack, void* stream);

/*
 * Self-test of the arithmetic contract: compares the kernel's packed correctly-rounded sqrt and
 * reciprocal sequences with IEEE sqrtf / division on EVERY float in [lo, hi] (within [2^-100, 1e30];
 * the reciprocal on the part >= 1).  mismatches: DEVICE u64[2] = {sqrt, reciprocal} counts.
 */
int va_selftest_exact_math(va_ctx* ctx, float lo, float hi, unsigned long long* mismatches, void* stream);

/*
 * Measurement hooks (bench.py): when enabled, va_tvl1_flow brackets every run of
 * inner-iteration launches with HIP events on `stream`.  va_tvl1_profile_read synchronises
 * those events and returns, summed over all va_tvl1_flow calls since the last reset:
 * out[0] = milliseconds inside the inner-iteration kernel, out[1] = its launches,
 * out[2] = pixel-iterations it performed (valid pixels x iterations; algorithmic bytes =
 * 64 B x out[2]), out[3] = pixel-warps (44 B each), out[4] = milliseconds during which at least one
 * run of inner-iteration launches was in flight (the union of the bracketed intervals: equals
 * out[0] on one stream, smaller when calls on several streams overlap) -- all as doubles (HOST
 * array of 5).  va_tvl1_profile_enable(1) synchronises the device and sets the time origin.
 */
int va_tvl1_profile_enable(va_ctx* ctx, int on);
int va_tvl1_profile_read(va_ctx* ctx, double* out, int reset);
/* The same measurement per pyramid level (0 = full resolution): out[3*s + 0] = summed per-call milliseconds of the
 * level's inner-iteration launches, out[3*s + 1] = its pixel-iterations, out[3*s + 2] = its launches (HOST array of
 * 3*n doubles, n <= 16).  Call va_tvl1_profile_read(reset = 0) first (it synchronises the events), then this. */
int va_tvl1_profile_levels(va_ctx* ctx, double* out, int n, int reset);

/* ------------------------------------------------ video-level aggregation and fusion --- */

/*
 * The AverageMeter bank of validate() (Sheet03/utils.py:154-171, Sheet03/spatialModel.py:223-228) kept
 * on the device: desc f32 [batch][dim] (the descriptors of one batch), slot i32 [batch] = index of each
 * clip's video in the caller's video list (a negative slot skips the row: padding);
 * sums f32 [n_slots][dim] += desc row, counts i32 [n_slots] += 1, rows applied in batch order (a video
 * occurring twice in a batch gets both adds in that order: bit-identical to update() row by row).
 * Replaces the per-batch device-to-host copy of the reference loop.
 */
int va_meter_update(va_ctx* ctx, const void* desc, const void* slot, int batch, int dim,
                    void* sums, void* counts, int n_slots, void* stream);
/* avg f32 [n_slots][dim] = sums / counts (AverageMeter.avg; 0 where counts == 0). */
int va_meter_average(va_ctx* ctx, const void* sums, const void* counts, int n_slots, int dim,
                     void* avg, void* stream);

/*
 * LinearSVC.predict of the fusion step (Sheet03/combinedModel.py:38): x f64 [n][dim] (the joined
 * descriptors of combineDescriptors, Sheet03/combinedModel.py:9-26), coef f64 [n_class_rows][dim],
 * intercept f64 [n_class_rows] (sklearn's coef_ / intercept_; n_class_rows == 1 for a binary problem)
 * -> scores f64 [n][n_class_rows] = x coef^T + intercept (sum over dim in ascending order, double
 * multiply then add), pred i32 [n] = index into classes_: arg-max (first maximum), or score > 0 for
 * the binary case.  Fitting the SVM stays on the CPU (liblinear).
 */
int va_linear_svm_predict(va_ctx* ctx, const void* x, int n, int dim, const void* coef,
                          const void* intercept, int n_class_rows, void* scores, void* pred,
                          void* stream);

/* ------------------------------------------------------------------ training step --- */

/*
 * The body of the batch loop of SpatialNetwork.train() / TemporalNetwork.train()
 * (Sheet03/spatialModel.py:165-182, Sheet03/temporalModel.py:194-211), fp32 models only:
 * forward in train mode (Dropout(p=0.5) after the three hidden classifier ReLUs; element i of dropout
 * layer d is kept and doubled iff video_analytics_amd.synth.hash_uniform(dropout_seed, 100 + d)[i] >= 0.5),
 * mean cross-entropy (nn.CrossEntropyLoss, Sheet03/spatialModel.py:114), backward through the whole network
 * (max-pool gradient to the first maximum of each window, like torch), then torch.optim.SGD's update of every
 * parameter (Sheet03/spatialModel.py:116: buf = momentum*buf + grad; p -= lr*buf; no weight decay).
 *   va_vgg16_train_init   allocates and zeroes the momentum buffers (the only allocation of the training path)
 *   x, labels (i64 [batch]): device; batch <= 64
 *   desc: device f32 [batch][desc_dim] or NULL: the train-mode descriptor tap (after the third Dropout:
 *         Sheet03/spatialModel.py:171-173, quirk 7 of SURVEY.md)
 *   loss_out: device f32[2] = { mean cross-entropy, number of arg-max hits }, both of the forward pass
 *             that preceded the update
 */
int va_vgg16_train_init(va_vgg16* model, void* stream);
size_t va_vgg16_train_workspace_bytes(const va_vgg16* model, int batch);
int va_vgg16_train_step(va_vgg16* model, const void* x, int x_is_u8, const void* labels, int batch,
                        float lr, float momentum, unsigned long long dropout_seed, void* desc,
                        void* loss_out, void* workspace, size_t workspace_bytes, void* stream);
/*
 * Checkpoints (Sheet03/spatialModel.py:234-260, Sheet03/utils.py:29-35): copy the parameters (which = 0) or the
 * momentum buffers (which = 1) out to / in from device tensors in the reference's layouts -- conv OIHW
 * [cout][cin][3][3], fc [out][in] (FC1's input CHW-major), biases [out] -- i.e. what model.state_dict() and
 * optimizer.state_dict()['state'][..]['momentum_buffer'] hold.
 */
/* Debugging aid of the tests: byte offsets inside the training workspace -- out[0..12] the 13 conv outputs,
 * out[13..25] the pooled maps (0 where a layer has no pool), out[26], out[27] the two gradient buffers,
 * out[28] the gradient at the classifier input, out[29] the NHWC input (HOST array of 30).  With the
 * va_vgg16_set_option(model, VA_OPT_TRAIN_STOP_AT, i), va_vgg16_train_step stops after the backward pass of conv
 * layer i (return code VA_ERR_STOPPED), leaving the gradient buffers as that layer left them. */
int va_vgg16_train_plan(const va_vgg16* model, int batch, unsigned long long* out);
int va_vgg16_export_state(va_vgg16* model, int which, void* const* conv_w, void* const* conv_b,
                          void* const* fc_w, void* const* fc_b, void* stream);
int va_vgg16_import_state(va_vgg16* model, int which, const void* const* conv_w, const void* const* conv_b,
                          const void* const* fc_w, const void* const* fc_b, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VA_H */
