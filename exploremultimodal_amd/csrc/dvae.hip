// Glue kernels of the dall_e dVAE encoder (dall_e/encoder.py:74-121) around the
// implicit-GEMM convolutions of gemm.hip.  Activations are NHWC fp16 matrices
// [B*H*W, C] (the reference runs this encoder in fp16 on GPU: dall_e/utils.py:37-42).
#include "common.h"
#include "vlmo_hip.h"
#include <stdlib.h>

namespace {

// stem: x fp32 NCHW [B,C,H,W] -> fp16 [B*H*W, Kpad], column = c*kw*kw + ky*kw + kx (zero beyond C*kw*kw
// and outside the image).  One thread = 8 consecutive columns (16-byte store); the kernel size is a template constant
// so that column -> (c, ky, kx) is multiplications by constants, and the pixel decomposition is done once per thread
// (the generic form spent 275 us in integer divisions for 308 MB of output).
template <int KW>
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, f16* __restrict__ out, int B, int C,
                                                     int H, int W, int kw_rt, int Kpad, long total8) {
    const int kw = KW ? KW : kw_rt;
    const int pad = (kw - 1) / 2, kk = kw * kw, kreal = C * kk, k8 = Kpad / 8;
    const int HW = H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
        const long m = i / k8;
        const int c0 = (int)(i - m * k8) * 8;
        const int b = (int)(m / HW), pix = (int)(m - (long)b * HW);
        const int y = pix / W, xx = pix - y * W;
        const float* xb = x + (size_t)b * C * HW;
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = c0 + j;
            float val = 0.f;
            if (col < kreal) {
                const int c = col / kk, r = col - c * kk, ky = r / kw, kx = r - ky * kw;
                const int sy = y + ky - pad, sx = xx + kx - pad;
                if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) val = xb[(c * H + sy) * W + sx];
            }
            v[j] = (f16)val;
        }
        __builtin_nontemporal_store(v, (f16x8*)(out + m * Kpad + c0));
    }
}

// Same matrix, one workgroup per image row: the KW input rows x C channels under the row's patches are read once,
// coalesced, into LDS with a zero halo (plus one all-zero row for the padding columns).  A thread owns ONE 8-column
// piece of the patch vector, so column -> (c, ky, kx) is decomposed once into eight LDS offsets, and walks the row's
// pixels: per piece eight LDS reads + one 16-byte store, the row's W * Kpad halves leave as one contiguous stream.
// (The gather form above pays ~25 integer instructions and one scattered global load per ELEMENT: 204 us for 308 MB.)
template <int KW>
__global__ __launch_bounds__(256) void im2col_row_kernel(const float* __restrict__ x, f16* __restrict__ out, int C,
                                                         int H, int W, int Kpad) {
    extern __shared__ float rows[];                 // [C * KW + 1][W + KW - 1]
    constexpr int pad = (KW - 1) / 2, kk = KW * KW;
    const int Wp = W + KW - 1, nrow = C * KW, k8 = Kpad / 8, kreal = C * kk;
    const int b = blockIdx.x / H, y = blockIdx.x - b * H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xb = x + (size_t)b * C * H * W;
    for (int r = wave; r <= nrow; r += 4) {
        const int c = r / KW, sy = y + (r - c * KW) - pad;
        const bool live = r < nrow && (unsigned)sy < (unsigned)H;
        const float* src = xb + ((size_t)c * H + (live ? sy : 0)) * W;
        for (int px = lane; px < Wp; px += 64) {
            const int sx = px - pad;
            rows[r * Wp + px] = (live && (unsigned)sx < (unsigned)W) ? src[sx] : 0.f;
        }
    }
    const int ppp = 256 / k8;                       // pixels per pass
    const int p = threadIdx.x / k8, v = threadIdx.x - p * k8;
    int off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int col = v * 8 + j;
        const int c = col / kk, r = col - c * kk, ky = r / KW, kx = r - ky * KW;
        off[j] = col < kreal ? (c * KW + ky) * Wp + kx : nrow * Wp;
    }
    __syncthreads();
    if (p >= ppp) return;
    f16* orow = out + (size_t)blockIdx.x * W * Kpad + v * 8;
    for (int xx = p; xx < W; xx += ppp) {
        f16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)rows[off[j] + xx];
        __builtin_nontemporal_store(o, (f16x8*)(orow + (size_t)xx * Kpad));
    }
}

// MaxPool2d(2) on NHWC (encoder.py:85,95,105) producing the raw pooled map and relu of it
// (relu(maxpool(x)) == maxpool(relu(x))).  One thread = 8 channels.
__global__ __launch_bounds__(256) void maxpool2_kernel(const f16* __restrict__ x, f16* __restrict__ raw,
                                                       f16* __restrict__ rel, int B, int H, int W, int C, long total8) {
    const int Ho = H / 2, Wo = W / 2, c8n = C / 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % c8n) * 8;
        const long m = i / c8n;
        const int b = (int)(m / (Ho * Wo)), pix = (int)(m % (Ho * Wo));
        const int y = pix / Wo, xx = pix % Wo;
        const f16* p = x + (((size_t)b * H + 2 * y) * W + 2 * xx) * C + c0;
        const f16x8 a = *(const f16x8*)p, bb = *(const f16x8*)(p + C);
        const f16x8 c = *(const f16x8*)(p + (size_t)W * C), d = *(const f16x8*)(p + (size_t)W * C + C);
        f16x8 r, rr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = fmaxf(fmaxf((float)a[j], (float)bb[j]), fmaxf((float)c[j], (float)d[j]));
            r[j] = (f16)v;
            rr[j] = (f16)fmaxf(v, 0.f);
        }
        *(f16x8*)(raw + m * C + c0) = r;
        if (rel) *(f16x8*)(rel + m * C + c0) = rr;
    }
}

// partial [M, nchunk, {value f32, index i32}] -> ids int64 [M]; ties -> lowest index (torch.argmax)
__global__ __launch_bounds__(256) void argmax_reduce_kernel(const float* __restrict__ part, int nchunk,
                                                            int64_t* __restrict__ ids, int M) {
    // one wavefront per row, lanes stride over the (value, index) pairs of the row's chunks
    typedef __attribute__((ext_vector_type(2))) int i32x2;
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const i32x2* p = (const i32x2*)(part + (size_t)m * nchunk * 2);
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < nchunk; c += 64) {
        const i32x2 q = p[c];
        const int q0 = q[0], idx = q[1];
        const float v = __builtin_bit_cast(float, q0);
        if (v > best || (v == best && idx < bi)) {
            best = v;
            bi = idx;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    if (lane == 0) ids[m] = bi;
}

// Finish of the fused cross-entropy forward (VLMO_EPI_CE): partial [M, nchunk, {max, sumexp, argmax, label logit}]
// -> lse [M], per-row loss (0 where label == ignore), prediction [M]
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ part, int nchunk,
                                                        const int32_t* __restrict__ labels, int ignore_index,
                                                        float* __restrict__ lse, float* __restrict__ loss,
                                                        int32_t* __restrict__ pred, int M) {
    // one wavefront per row: the lanes stride over the row's chunks with 16-byte loads (a thread per row walked its
    // 477 chunks serially with a 7.6 KB stride between neighbouring threads: 211 us for a few thousand rows)
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    const i32x4* p = (const i32x4*)(part + (size_t)m * nchunk * 4);
    float best = -INFINITY, labv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < nchunk; c += 64) {
        // (scalar copies first: hipcc 7.2 folds __builtin_bit_cast of an ext-vector ELEMENT to element 0)
        const i32x4 q = p[c];
        const int q0 = q[0], idx = q[2], q3 = q[3];
        const float v = __builtin_bit_cast(float, q0);
        if (v > best || (v == best && idx < bi)) {
            best = v;
            bi = idx;
        }
        labv = fmaxf(labv, __builtin_bit_cast(float, q3));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
        labv = fmaxf(labv, __shfl_xor(labv, o, 64));
    }
    float s = 0.f;
    for (int c = lane; c < nchunk; c += 64) {
        const i32x4 q = p[c];
        const int q0 = q[0], q1 = q[1];
        s += __builtin_bit_cast(float, q1) * __expf(__builtin_bit_cast(float, q0) - best);
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float l = best + __logf(s);
        lse[m] = l;
        if (pred) pred[m] = bi;
        if (loss) loss[m] = (labels[m] == ignore_index) ? 0.f : l - labv;
    }
}

}  // namespace

extern "C" int vlmo_ce_reduce(const float* partial, int nchunk, const int32_t* labels, int ignore_index, float* lse,
                              float* loss, int32_t* pred, int M, hipStream_t stream) {
    VLMO_CHECK_ARG(partial && labels && lse && M > 0 && nchunk > 0, "vlmo_ce_reduce: bad arguments");
    hipLaunchKernelGGL(ce_reduce_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, partial, nchunk, labels,
                       ignore_index, lse, loss, pred, M);
    VLMO_CHECK_LAUNCH("vlmo_ce_reduce");
    return 0;
}

extern "C" int vlmo_dvae_im2col(const float* x, void* out, int B, int C, int H, int W, int kw, int Kpad,
                                hipStream_t stream) {
    VLMO_CHECK_ARG(x && out, "vlmo_dvae_im2col: null pointer");
    VLMO_CHECK_ARG(B > 0 && C > 0 && kw % 2 == 1 && Kpad % 64 == 0 && Kpad >= C * kw * kw, "vlmo_dvae_im2col: bad shape");
    const long total8 = (long)B * H * W * (Kpad / 8);
    const int grid = (int)((total8 + 255) / 256 < 65536 ? (total8 + 255) / 256 : 65536);
    const size_t row_lds = (size_t)(C * kw + 1) * (W + kw - 1) * sizeof(float);
    static const bool gather = getenv("VLMO_IM2COL_GATHER") != nullptr;      // measurement aid: the element-gather form
    if (kw == 7 && Kpad <= 2048 && row_lds <= 48 * 1024 && (long)B * H < 0x7fffffffL && !gather)
        hipLaunchKernelGGL(im2col_row_kernel<7>, dim3(B * H), dim3(256), row_lds, stream, x, (f16*)out, C, H, W, Kpad);
    else if (kw == 7)
        hipLaunchKernelGGL(im2col_kernel<7>, dim3(grid), dim3(256), 0, stream, x, (f16*)out, B, C, H, W, kw, Kpad, total8);
    else
        hipLaunchKernelGGL(im2col_kernel<0>, dim3(grid), dim3(256), 0, stream, x, (f16*)out, B, C, H, W, kw, Kpad, total8);
    VLMO_CHECK_LAUNCH("vlmo_dvae_im2col");
    return 0;
}

extern "C" int vlmo_maxpool2_nhwc(const void* x, void* raw, void* relu, int B, int H, int W, int C,
                                  hipStream_t stream) {
    VLMO_CHECK_ARG(x && raw, "vlmo_maxpool2_nhwc: null pointer");
    VLMO_CHECK_ARG(B > 0 && H % 2 == 0 && W % 2 == 0 && C % 8 == 0, "vlmo_maxpool2_nhwc: bad shape");
    const long total8 = (long)B * (H / 2) * (W / 2) * (C / 8);
    const int grid = (int)((total8 + 255) / 256 < 65536 ? (total8 + 255) / 256 : 65536);
    hipLaunchKernelGGL(maxpool2_kernel, dim3(grid), dim3(256), 0, stream, (const f16*)x, (f16*)raw, (f16*)relu, B, H, W,
                       C, total8);
    VLMO_CHECK_LAUNCH("vlmo_maxpool2_nhwc");
    return 0;
}

extern "C" int vlmo_argmax_reduce(const float* partial, int nchunk, int64_t* ids, int M, hipStream_t stream) {
    VLMO_CHECK_ARG(partial && ids && M > 0 && nchunk > 0, "vlmo_argmax_reduce: bad arguments");
    hipLaunchKernelGGL(argmax_reduce_kernel, dim3((M + 3) / 4), dim3(256), 0, stream, partial, nchunk, ids, M);
    VLMO_CHECK_LAUNCH("vlmo_argmax_reduce");
    return 0;
}
