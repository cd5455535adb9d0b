// Video-level aggregation and two-stream fusion on the device:
// the AverageMeter bank of validate() (Sheet03/utils.py:154-171, Sheet03/spatialModel.py:223-228)
// and LinearSVC.predict of the fusion step (Sheet03/combinedModel.py:38).  Both are small,
// HBM/latency-bound byte-and-index work: plain coalesced kernels, no MFMA.
#include "va_internal.h"

namespace {

// One thread per descriptor column walks the batch in order: a video that occurs several times in
// one batch receives its adds in batch order, exactly like the reference's Python loop, so the f32
// sums are bit-identical to AverageMeter.update() called row by row.
__global__ void k_meter_update(const float* __restrict__ desc, const int* __restrict__ slot, int B, int D, float* __restrict__ sums,
                               int* __restrict__ counts, int n_slots)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    for (int b = 0; b < B; ++b) {
        const int s = slot[b];
        if (s < 0 || s >= n_slots) continue;  // padding rows
        sums[(size_t)s * D + d] += desc[(size_t)b * D + d];
        if (d == 0) counts[s] += 1;
    }
}

__global__ void k_meter_average(const float* __restrict__ sums, const int* __restrict__ counts, int n_slots, int D, float* __restrict__ avg)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_slots * D) return;
    const int c = counts[i / D];
    avg[i] = c > 0 ? sums[i] / (float)c : 0.0f;
}

// scores[n][c] = sum_k x[n][k] * coef[c][k] (k ascending, IEEE double multiply then add: -ffp-contract=off)
// + intercept[c].  One workgroup per row n: the row is staged in LDS, thread c walks coef row c.
__global__ void k_svm_scores(const double* __restrict__ x, int dim, const double* __restrict__ coef, const double* __restrict__ intercept,
                             int C, double* __restrict__ scores)
{
    extern __shared__ double sx[];
    const int n = blockIdx.x;
    for (int k = threadIdx.x; k < dim; k += blockDim.x) sx[k] = x[(size_t)n * dim + k];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double* w = coef + (size_t)c * dim;
        double acc = 0.0;
        for (int k = 0; k < dim; ++k) acc = acc + sx[k] * w[k];
        scores[(size_t)n * C + c] = acc + intercept[c];
    }
}

// LinearSVC.predict: arg-max over classes (first maximum, like numpy.argmax); one class row
// (binary problem): index of (score > 0).
__global__ void k_svm_argmax(const double* __restrict__ scores, int N, int C, int* __restrict__ pred)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double* s = scores + (size_t)n * C;
    if (C == 1) {
        pred[n] = s[0] > 0.0 ? 1 : 0;
        return;
    }
    double mx = s[0];
    int am = 0;
    for (int c = 1; c < C; ++c)
        if (s[c] > mx) { mx = s[c]; am = c; }
    pred[n] = am;
}

}  // namespace

extern "C" int va_meter_update(va_ctx* ctx, const void* desc, const void* slot, int batch, int dim, void* sums, void* counts, int n_slots,
                               void* stream)
{
    VA_CHECK_ARG(ctx != nullptr, "va_meter_update: ctx is NULL");
    VA_USE_DEVICE(ctx);
    VA_CHECK_ARG(desc && slot && sums && counts, "va_meter_update: NULL pointer");
    VA_CHECK_ARG(batch >= 1 && dim >= 1 && n_slots >= 1, "va_meter_update: batch, dim, n_slots must be >= 1 (got %d, %d, %d)", batch, dim, n_slots);
    k_meter_update<<<va_cdiv(dim, 256), 256, 0, (hipStream_t)stream>>>((const float*)desc, (const int*)slot, batch, dim, (float*)sums,
                                                                       (int*)counts, n_slots);
    VA_LAUNCH_CHECK();
    return VA_OK;
}

extern "C" int va_meter_average(va_ctx* ctx, const void* sums, const void* counts, int n_slots, int dim, void* avg, void* stream)
{
    VA_CHECK_ARG(ctx != nullptr, "va_meter_average: ctx is NULL");
    VA_USE_DEVICE(ctx);
    VA_CHECK_ARG(sums && counts && avg, "va_meter_average: NULL pointer");
    VA_CHECK_ARG(dim >= 1 && n_slots >= 1, "va_meter_average: dim, n_slots must be >= 1 (got %d, %d)", dim, n_slots);
    const size_t n = (size_t)n_slots * dim;
    k_meter_average<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>((const float*)sums, (const int*)counts, n_slots, dim, (float*)avg);
    VA_LAUNCH_CHECK();
    return VA_OK;
}

extern "C" int va_linear_svm_predict(va_ctx* ctx, const void* x, int n, int dim, const void* coef, const void* intercept, int n_class_rows,
                                     void* scores, void* pred, void* stream)
{
    VA_CHECK_ARG(ctx != nullptr, "va_linear_svm_predict: ctx is NULL");
    VA_USE_DEVICE(ctx);
    VA_CHECK_ARG(x && coef && intercept && scores && pred, "va_linear_svm_predict: NULL pointer");
    VA_CHECK_ARG(n >= 1 && dim >= 1 && dim <= 8192 && n_class_rows >= 1,
                 "va_linear_svm_predict: need n >= 1, 1 <= dim <= 8192, n_class_rows >= 1 (got %d, %d, %d)", n, dim, n_class_rows);
    k_svm_scores<<<n, 128, (size_t)dim * sizeof(double), (hipStream_t)stream>>>((const double*)x, dim, (const double*)coef, (const double*)intercept,
                                                                               n_class_rows, (double*)scores);
    VA_LAUNCH_CHECK();
    k_svm_argmax<<<va_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>((const double*)scores, n, n_class_rows, (int*)pred);
    VA_LAUNCH_CHECK();
    return VA_OK;
}
