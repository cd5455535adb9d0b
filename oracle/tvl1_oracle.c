/*
 * oracle/tvl1_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar per pixel, OpenMP across frame pairs only) of
 * the dense TV-L1 optical flow that produces the flow_x / flow_y images the
 * reference's temporal stream reads (Sheet03/temporalModel.py:76-81,
 * Sheet03/parameters.py:27,38-39).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this file's shared object.
 *
 * PARITY UNPINNED: the reference repository contains no TV-L1 code, no flow
 * fixtures and no golden vectors: it only reads JPEGs
 * written by an unknown upstream tool.  This file therefore restates the
 * PUBLISHED algorithm
 *   - C. Zach, T. Pock, H. Bischof, "A Duality Based Approach for Realtime
 *     TV-L1 Optical Flow", DAGM 2007;
 *   - J. Sanchez Perez, E. Meinhardt-Llopis, G. Facciolo, "TV-L1 Optical Flow
 *     Estimation", IPOL 2013 (Algorithm 1 + multiscale driver),
 * with the parameter set and the bilinear (rather than bicubic) warp of the
 * OpenCV cuda::OpticalFlowDual_TVL1 generation that the directory name
 * "..._tvl1_gpu" points at, and it is pinned only by analytic known-answer
 * tests (zero motion => zero flow, constant translation => constant flow) and
 * by its own committed outputs (tests/golden/).  Every choice the papers leave
 * open is fixed below and mirrored in DESIGN.md section "TV-L1 specification".
 *
 * Arithmetic contract (so that an independent implementation can be compared
 * bit for bit): IEEE-754 binary32 everywhere, no contraction except where
 * fmaf() is written out, correctly rounded sqrtf and division, the operation
 * order exactly as written.  Build with -ffp-contract=off.
 *
 *   S0  Frames are float in [0,255] (u8 promoted exactly).  No normalisation,
 *       no pre-smoothing of level 0.
 *   S1  Pyramid: w[s] = (int)((float)w[s-1]*step + 0.5f) (same for h); levels
 *       stop before min(w,h) < 16.  Level s = separable Gaussian of level s-1
 *       (sigma = 0.6*sqrt(1/step^2 - 1), radius = (int)(3 sigma) + 1, taps
 *       normalised in double then rounded to float, replicate border,
 *       accumulation k = -R..R as acc = g*v then fmaf) followed by bilinear
 *       sampling at (j*fx, i*fy), fx = (float)w[s-1]/(float)w[s].
 *   S2  Centred gradient of every level of the second frame:
 *       Ix = 0.5f*(I[min(x+1,w-1)] - I[max(x-1,0)]).
 *   S3  bilinear(img,x,y): clamp x to [0,w-1], y to [0,h-1]; x0 = (int)x,
 *       x1 = min(x0+1,w-1), ax = x - x0; top = fmaf(ax, b-a, a),
 *       bot = fmaf(ax, d-c, c), value = fmaf(ay, bot-top, top).
 *   S4  Per level, coarse to fine: u = 0 at the coarsest level, p = 0 at the
 *       start of every level; `warps` times { warp; up to `iters` inner
 *       iterations }.
 *   S5  Warp at pixel (x,y): sample I1, I1x, I1y at (x+u1, y+u2) with S3;
 *       grad = fmaf(I1wy,I1wy, I1wx*I1wx); ig = grad < 1e-10f ? 0 : 1/grad;
 *       rho_c = fmaf(-I1wy,u2, fmaf(-I1wx,u1, I1w - I0)).
 *   S6  Inner iteration (Zach et al. eq. 12-15; IPOL Algorithm 1 lines 8-17):
 *         rho = fmaf(I1wy,u2, fmaf(I1wx,u1, rho_c));
 *         fi  = fminf(fmaxf(-rho*ig, -l_t), l_t);              (TH operator,
 *               written as a clamp: the three IPOL cases coincide with it)
 *         v1  = fmaf(fi,I1wx,u1);  v2 = fmaf(fi,I1wy,u2);
 *         u1' = fmaf(theta, div(p11,p12), v1);  (same for u2)
 *         d1  = fmaf(taut, sqrtf(fmaf(u1y,u1y, fmaf(u1x,u1x, 2^-100))), 1); d2 likewise from u2
 *               (|grad u| regularised by 2^-50: never 0 under the square root);
 *         rinv = 1/(d1*d2); r1 = d2*rinv; r2 = d1*rinv;   (1/d1 and 1/d2 from ONE division)
 *         p11 = fmaf(taut,u1x,p11)*r1; p12 = fmaf(taut,u1y,p12)*r1; p21, p22 with r2
 *       with l_t = lambda*theta, taut = tau/theta, backward-difference
 *       divergence and forward-difference gradient with the IPOL boundary
 *       rules.
 *   S7  Stopping rule (epsilon > 0 only): per iteration
 *         e = (u1'-u1)^2 + (u2'-u2)^2 (float: d1*d1 then fmaf(d2,d2,.)),
 *         q = (uint64)(fminf(e,1024.f) * 4294967296.f), summed EXACTLY in
 *       integers (order independent), stop when
 *         sum q < (uint64)((double)eps*eps * npix * 4294967296.0).
 *       epsilon <= 0 means "run exactly `iters` iterations".
 *   S8  Upsampling to the next finer level: bilinear at (x*rx, y*ry),
 *       rx = (float)w[s]/(float)w[s-1], times inv_step = 1.0f/step.
 *   S9  flow_to_stack (quantisation convention of the public dense_flow /
 *       TSN tool chain, to which the reference's 8-bit flow JPEGs belong;
 *       normalisation of Sheet03/utils.py:148-150 with the single-channel
 *       rule of SURVEY.md a5):
 *         t = (255.0f*(v + bound)) / (2.0f*bound); q = rintf(clamp(t,0,255));
 *         out = (q/255.0f - mean)/std; channel 2k = x flow of pair k,
 *         2k+1 = y flow (Sheet03/temporalModel.py:83).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct ora_tvl1_params {
    float tau, lambda, theta;
    int nscales, warps;
    float epsilon;
    int iters;
    float scale_step;
} ora_tvl1_params;

#define ORA_MAX_SCALES 16
#define ORA_MAX_RADIUS 8
#define ORA_NORM_REG 0x1p-100f /* S6: regulariser of |grad u|^2 */

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* S1: level sizes.  Returns the number of levels actually used. */
int ora_tvl1_pyramid_sizes(int w, int h, int nscales, float step, int* ws, int* hs)
{
    int n = 1;
    ws[0] = w; hs[0] = h;
    if (nscales > ORA_MAX_SCALES) nscales = ORA_MAX_SCALES;
    while (n < nscales) {
        int nw = (int)((float)ws[n - 1] * step + 0.5f);
        int nh = (int)((float)hs[n - 1] * step + 0.5f);
        if (imin(nw, nh) < 16) break;
        ws[n] = nw; hs[n] = nh; n++;
    }
    return n;
}

/* S1: Gaussian taps for the zoom-out filter.  Returns the radius. */
int ora_tvl1_zoom_taps(float step, float* taps /* [2*R+1] */)
{
    float sigma = 0.6f * sqrtf(1.0f / (step * step) - 1.0f);
    int R = (int)(3.0f * sigma) + 1;
    double g[2 * ORA_MAX_RADIUS + 1], sum = 0.0;
    int k;
    if (R > ORA_MAX_RADIUS) R = ORA_MAX_RADIUS;
    for (k = -R; k <= R; k++) {
        g[k + R] = exp(-(double)(k * k) / (2.0 * (double)sigma * (double)sigma));
        sum += g[k + R];
    }
    for (k = 0; k <= 2 * R; k++) taps[k] = (float)(g[k] / sum);
    return R;
}

/* S3 */
static float bilinear(const float* img, int w, int h, float x, float y)
{
    int x0, x1, y0, y1;
    float ax, ay, a, b, c, d, top, bot;
    x = fminf(fmaxf(x, 0.0f), (float)(w - 1));
    y = fminf(fmaxf(y, 0.0f), (float)(h - 1));
    x0 = (int)x; y0 = (int)y;
    x1 = imin(x0 + 1, w - 1); y1 = imin(y0 + 1, h - 1);
    ax = x - (float)x0; ay = y - (float)y0;
    a = img[y0 * w + x0]; b = img[y0 * w + x1];
    c = img[y1 * w + x0]; d = img[y1 * w + x1];
    top = fmaf(ax, b - a, a);
    bot = fmaf(ax, d - c, c);
    return fmaf(ay, bot - top, top);
}

/* S1: one zoom-out step. */
void ora_tvl1_zoom_out(const float* in, int w, int h, float* out, int ow, int oh, float step)
{
    float taps[2 * ORA_MAX_RADIUS + 1];
    int R = ora_tvl1_zoom_taps(step, taps);
    float* t1 = (float*)malloc(sizeof(float) * (size_t)w * h);
    float* t2 = (float*)malloc(sizeof(float) * (size_t)w * h);
    float fx = (float)w / (float)ow, fy = (float)h / (float)oh;
    int x, y, k;
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) {
            float acc = taps[0] * in[y * w + imax(x - R, 0)];
            for (k = -R + 1; k <= R; k++)
                acc = fmaf(taps[k + R], in[y * w + imin(imax(x + k, 0), w - 1)], acc);
            t1[y * w + x] = acc;
        }
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) {
            float acc = taps[0] * t1[imax(y - R, 0) * w + x];
            for (k = -R + 1; k <= R; k++)
                acc = fmaf(taps[k + R], t1[imin(imax(y + k, 0), h - 1) * w + x], acc);
            t2[y * w + x] = acc;
        }
    for (y = 0; y < oh; y++)
        for (x = 0; x < ow; x++)
            out[y * ow + x] = bilinear(t2, w, h, (float)x * fx, (float)y * fy);
    free(t1); free(t2);
}

/* S2 */
void ora_tvl1_centered_gradient(const float* I, int w, int h, float* Ix, float* Iy)
{
    int x, y;
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) {
            Ix[y * w + x] = 0.5f * (I[y * w + imin(x + 1, w - 1)] - I[y * w + imax(x - 1, 0)]);
            Iy[y * w + x] = 0.5f * (I[imin(y + 1, h - 1) * w + x] - I[imax(y - 1, 0) * w + x]);
        }
}

/* IPOL divergence (backward differences) at one pixel. */
static float divergence_at(const float* v1, const float* v2, int w, int h, int x, int y)
{
    float v1x, v2y;
    int i = y * w + x;
    if (w == 1) v1x = 0.0f;
    else if (x == 0) v1x = v1[i];
    else if (x == w - 1) v1x = -v1[i - 1];
    else v1x = v1[i] - v1[i - 1];
    if (h == 1) v2y = 0.0f;
    else if (y == 0) v2y = v2[i];
    else if (y == h - 1) v2y = -v2[i - w];
    else v2y = v2[i] - v2[i - w];
    return v1x + v2y;
}

/* S4-S7: one pyramid level.  u1,u2 in/out; returns inner iterations executed. */
long ora_tvl1_level(const float* I0, const float* I1, const float* I1x, const float* I1y,
                    int w, int h, const ora_tvl1_params* P, float* u1, float* u2)
{
    size_t n = (size_t)w * h;
    float* buf = (float*)calloc(n * 10, sizeof(float));
    float *p11 = buf, *p12 = buf + n, *p21 = buf + 2 * n, *p22 = buf + 3 * n;
    float *wx = buf + 4 * n, *wy = buf + 5 * n, *rc = buf + 6 * n, *ig = buf + 7 * n;
    float *n1 = buf + 8 * n, *n2 = buf + 9 * n;
    const float l_t = P->lambda * P->theta;
    const float taut = P->tau / P->theta;
    const float theta = P->theta;
    const int fixed = !(P->epsilon > 0.0f);
    const uint64_t qthr = fixed ? 0 : (uint64_t)((double)P->epsilon * (double)P->epsilon * (double)n * 4294967296.0);
    long total = 0;
    int wp, it;
    /* Large frames called one pair at a time (the 1280x720 tests) spread the ROWS of a pass over the threads; inside the
     * pair-parallel region of ora_tvl1_flow these inner regions are nested and run on one thread.  Every pixel of a pass
     * is computed from the previous pass's arrays only and the error sum is an exact integer: results do not depend on it. */
    const int par_rows = n >= 262144;

    for (wp = 0; wp < P->warps; wp++) {
        /* S5 */
#pragma omp parallel for schedule(static) if (par_rows)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                float fxp, fyp, Iw, Iwx, Iwy, grad;
                const size_t i = (size_t)y * w + x;
                fxp = (float)x + u1[i]; fyp = (float)y + u2[i];
                Iw = bilinear(I1, w, h, fxp, fyp);
                Iwx = bilinear(I1x, w, h, fxp, fyp);
                Iwy = bilinear(I1y, w, h, fxp, fyp);
                grad = fmaf(Iwy, Iwy, Iwx * Iwx);
                wx[i] = Iwx; wy[i] = Iwy;
                ig[i] = grad < 1e-10f ? 0.0f : 1.0f / grad;
                rc[i] = fmaf(-Iwy, u2[i], fmaf(-Iwx, u1[i], Iw - I0[i]));
            }
        /* S6, S7 */
        for (it = 0; it < P->iters; it++) {
            uint64_t qsum = 0;
#pragma omp parallel for schedule(static) reduction(+ : qsum) if (par_rows)
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    float rho, fi, v1, v2, a, b;
                    const size_t i = (size_t)y * w + x;
                    rho = fmaf(wy[i], u2[i], fmaf(wx[i], u1[i], rc[i]));
                    fi = fminf(fmaxf(-rho * ig[i], -l_t), l_t);
                    v1 = fmaf(fi, wx[i], u1[i]);
                    v2 = fmaf(fi, wy[i], u2[i]);
                    a = fmaf(theta, divergence_at(p11, p12, w, h, x, y), v1);
                    b = fmaf(theta, divergence_at(p21, p22, w, h, x, y), v2);
                    if (!fixed) {
                        float d1 = a - u1[i], d2 = b - u2[i];
                        float e = fmaf(d2, d2, d1 * d1);
                        qsum += (uint64_t)(fminf(e, 1024.0f) * 4294967296.0f);
                    }
                    n1[i] = a; n2[i] = b;
                }
            memcpy(u1, n1, n * sizeof(float));
            memcpy(u2, n2, n * sizeof(float));
#pragma omp parallel for schedule(static) if (par_rows)
            for (int y = 0; y < h; y++)
                for (int x = 0; x < w; x++) {
                    float u1x, u1y, u2x, u2y, d1, d2, rinv, r1, r2;
                    const size_t i = (size_t)y * w + x;
                    u1x = x < w - 1 ? u1[i + 1] - u1[i] : 0.0f;
                    u1y = y < h - 1 ? u1[i + w] - u1[i] : 0.0f;
                    u2x = x < w - 1 ? u2[i + 1] - u2[i] : 0.0f;
                    u2y = y < h - 1 ? u2[i + w] - u2[i] : 0.0f;
                    d1 = fmaf(taut, sqrtf(fmaf(u1y, u1y, fmaf(u1x, u1x, ORA_NORM_REG))), 1.0f);
                    d2 = fmaf(taut, sqrtf(fmaf(u2y, u2y, fmaf(u2x, u2x, ORA_NORM_REG))), 1.0f);
                    rinv = 1.0f / (d1 * d2); /* one division per pixel: 1/d1 = d2/(d1 d2) */
                    r1 = d2 * rinv;
                    r2 = d1 * rinv;
                    p11[i] = fmaf(taut, u1x, p11[i]) * r1;
                    p12[i] = fmaf(taut, u1y, p12[i]) * r1;
                    p21[i] = fmaf(taut, u2x, p21[i]) * r2;
                    p22[i] = fmaf(taut, u2y, p22[i]) * r2;
                }
            total++;
            if (!fixed && qsum < qthr) break;
        }
    }
    free(buf);
    return total;
}

/* S8 */
void ora_tvl1_upsample_flow(const float* uc, int cw, int ch, float* uf, int fw, int fh, float step)
{
    float rx = (float)cw / (float)fw, ry = (float)ch / (float)fh;
    float inv = 1.0f / step;
    int x, y;
    for (y = 0; y < fh; y++)
        for (x = 0; x < fw; x++)
            uf[y * fw + x] = bilinear(uc, cw, ch, (float)x * rx, (float)y * ry) * inv;
}

/* One frame's pyramid (levels + gradients). */
typedef struct { float* I[ORA_MAX_SCALES]; float* Ix[ORA_MAX_SCALES]; float* Iy[ORA_MAX_SCALES]; } ora_pyr;

static void pyr_build(ora_pyr* p, const float* frame, int ns, const int* ws, const int* hs, float step)
{
    int s;
    for (s = 0; s < ns; s++) {
        size_t n = (size_t)ws[s] * hs[s];
        p->I[s] = (float*)malloc(n * sizeof(float));
        p->Ix[s] = (float*)malloc(n * sizeof(float));
        p->Iy[s] = (float*)malloc(n * sizeof(float));
        if (s == 0) memcpy(p->I[0], frame, n * sizeof(float));
        else ora_tvl1_zoom_out(p->I[s - 1], ws[s - 1], hs[s - 1], p->I[s], ws[s], hs[s], step);
        ora_tvl1_centered_gradient(p->I[s], ws[s], hs[s], p->Ix[s], p->Iy[s]);
    }
}
static void pyr_free(ora_pyr* p, int ns)
{
    int s;
    for (s = 0; s < ns; s++) { free(p->I[s]); free(p->Ix[s]); free(p->Iy[s]); }
}

/*
 * Flow for every consecutive pair of every sequence.
 * frames [n_seq][frames_per_seq][h][w] float, flow [n_seq*(frames_per_seq-1)][2][h][w].
 * iters_run (may be NULL): inner iterations executed per pair (all levels, all warps).
 * Returns 0, or -1 on bad arguments.
 */
int ora_tvl1_flow(const float* frames, int n_seq, int frames_per_seq, int w, int h,
                  const ora_tvl1_params* P, float* flow, long* iters_run, int nthreads)
{
    int ws[ORA_MAX_SCALES], hs[ORA_MAX_SCALES];
    int ns, npairs, pi;
    if (!frames || !flow || !P || n_seq < 1 || frames_per_seq < 2 || w < 1 || h < 1) return -1;
    if (P->nscales < 1 || P->warps < 1 || P->iters < 1 || !(P->scale_step > 0.0f && P->scale_step < 1.0f) ||
        !(P->theta > 0.0f) || !(P->tau > 0.0f) || !(P->lambda > 0.0f)) return -1;
    ns = ora_tvl1_pyramid_sizes(w, h, P->nscales, P->scale_step, ws, hs);
    npairs = n_seq * (frames_per_seq - 1);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) if (npairs > 1) /* one pair: its rows are spread over the threads instead (ora_tvl1_level) */
    for (pi = 0; pi < npairs; pi++) {
        int sq = pi / (frames_per_seq - 1), k = pi % (frames_per_seq - 1), s;
        const float* f0 = frames + ((size_t)sq * frames_per_seq + k) * (size_t)w * h;
        const float* f1 = f0 + (size_t)w * h;
        ora_pyr A, B;
        float *u1, *u2;
        long total = 0;
        pyr_build(&A, f0, ns, ws, hs, P->scale_step);
        pyr_build(&B, f1, ns, ws, hs, P->scale_step);
        u1 = (float*)calloc((size_t)ws[ns - 1] * hs[ns - 1], sizeof(float));
        u2 = (float*)calloc((size_t)ws[ns - 1] * hs[ns - 1], sizeof(float));
        for (s = ns - 1; s >= 0; s--) {
            total += ora_tvl1_level(A.I[s], B.I[s], B.Ix[s], B.Iy[s], ws[s], hs[s], P, u1, u2);
            if (s > 0) {
                size_t nf = (size_t)ws[s - 1] * hs[s - 1];
                float* f1u = (float*)malloc(nf * sizeof(float));
                float* f2u = (float*)malloc(nf * sizeof(float));
                ora_tvl1_upsample_flow(u1, ws[s], hs[s], f1u, ws[s - 1], hs[s - 1], P->scale_step);
                ora_tvl1_upsample_flow(u2, ws[s], hs[s], f2u, ws[s - 1], hs[s - 1], P->scale_step);
                free(u1); free(u2); u1 = f1u; u2 = f2u;
            }
        }
        memcpy(flow + (size_t)pi * 2 * w * h, u1, (size_t)w * h * sizeof(float));
        memcpy(flow + ((size_t)pi * 2 + 1) * w * h, u2, (size_t)w * h * sizeof(float));
        if (iters_run) iters_run[pi] = total;
        free(u1); free(u2);
        pyr_free(&A, ns); pyr_free(&B, ns);
    }
    return 0;
}

/* S9.  flow [n_pairs][2][h][w] -> stack [2*n_pairs][h][w] (same memory order; values quantised+normalised). */
int ora_flow_to_stack(const float* flow, int n_pairs, int w, int h, float bound, float mean, float stdv, float* stack)
{
    size_t n = (size_t)n_pairs * 2 * w * h, i;
    if (!flow || !stack || n_pairs < 1 || !(bound > 0.0f) || !(stdv > 0.0f)) return -1;
    for (i = 0; i < n; i++) {
        float t = (255.0f * (flow[i] + bound)) / (2.0f * bound);
        float q = rintf(fminf(fmaxf(t, 0.0f), 255.0f));
        stack[i] = (q / 255.0f - mean) / stdv;
    }
    return 0;
}
