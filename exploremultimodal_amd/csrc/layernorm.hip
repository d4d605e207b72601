// LayerNorm forward/backward, one 64-lane wavefront per token row, fp32 math.
// Reference: LayerNorm factory vlmo.py:26-36 bound with eps=1e-12 at
// vlmo_module.py:21-23; used at vlmo.py:188,192 (norm1/norm2) and :413 (norm).
// HBM-bound: each row is read once (16 B/lane vector loads) and kept in
// registers for both the statistics and the normalisation.
#include "common.h"
#include "vlmo_hip.h"

namespace {

template <int VPL, bool OUT_F32>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, void* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     const int32_t* __restrict__ rowmap, int M, int d, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d >> 2;
    for (int m = blockIdx.x * 4 + wave; m < M; m += gridDim.x * 4) {
        const f32x4* xr = (const f32x4*)(x + (size_t)m * d);
        f32x4 v[VPL];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int i = lane + 64 * j;
            v[j] = (i < nv) ? xr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
        }
        const float mu = wave_sum(s) / d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = v[j][k] - mu;
                    q += t * t;
                }
            }
        }
        const float rs = rsqrtf(wave_sum(q) / d + eps);
        if (lane == 0) {
            if (mean) mean[m] = mu;
            if (rstd) rstd[m] = rs;
        }
        const int om = rowmap ? rowmap[m] : m;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                const f32x4 ww = ((const f32x4*)w)[i], bb = ((const f32x4*)b)[i];
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (v[j][k] - mu) * rs * ww[k] + bb[k];
                if constexpr (OUT_F32) {
                    ((f32x4*)((float*)y + (size_t)om * d))[i] = o;
                } else {
                    bf16x4 ob = {(bf16)o[0], (bf16)o[1], (bf16)o[2], (bf16)o[3]};
                    ((bf16x4*)((bf16*)y + (size_t)om * d))[i] = ob;
                }
            }
        }
    }
}

template <int VPL, bool DY_F32>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, const int32_t* __restrict__ rowmap,
                                                     const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ dres, float* __restrict__ dx,
                                                     float* __restrict__ ws, int M, int d, int rows_per_block) {
    __shared__ float red[4][2][VPL * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d >> 2;
    f32x4 aw[VPL], ab[VPL], ww[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        aw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[j] = aw[j];
        const int i = lane + 64 * j;
        ww[j] = (i < nv) ? ((const f32x4*)w)[i] : aw[j];
    }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    // Everything a row needs from HBM (dy, x, the incoming residual gradient, mean/rstd) is fetched one row
    // AHEAD of the arithmetic: beside the weight-gradient GEMMs of the side stream this kernel gets one or two
    // waves per SIMD, so its speed is (bytes in flight per wave) / latency, not occupancy.
    struct Row {
        f32x4 dy[VPL], x[VPL], dr[VPL];
        float mu, rs;
    };
    auto fetch = [&](int m, Row& r) {
        const int sm = rowmap ? rowmap[m] : m;
        r.mu = mean[m];
        r.rs = rstd[m];
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                if constexpr (DY_F32) {
                    r.dy[j] = ((const f32x4*)((const float*)dy + (size_t)sm * d))[i];
                } else {
                    const bf16x4 t = ((const bf16x4*)((const bf16*)dy + (size_t)sm * d))[i];
                    r.dy[j] = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
                }
                r.x[j] = ((const f32x4*)(x + (size_t)m * d))[i];
                r.dr[j] = dres ? ((const f32x4*)(dres + (size_t)m * d))[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    Row cur, nxt;
    int m = r0 + wave;
    if (m < r1) fetch(m, cur);
    for (; m < r1; m += 4) {
        if (m + 4 < r1) fetch(m + 4, nxt);
        const float mu = cur.mu, rs = cur.rs;
        f32x4 g[VPL], xh[VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                const f32x4 dyv = cur.dy[j], xv = cur.x[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    xh[j][k] = (xv[k] - mu) * rs;
                    g[j][k] = dyv[k] * ww[j][k];
                    s1 += g[j][k];
                    s2 += g[j][k] * xh[j][k];
                    aw[j][k] += dyv[k] * xh[j][k];
                    ab[j][k] += dyv[k];
                }
            } else {
                g[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                xh[j] = g[j];
            }
        }
        const float c1 = wave_sum(s1) / d, c2 = wave_sum(s2) / d;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = rs * (g[j][k] - c1 - xh[j][k] * c2);
                if (dres) o += cur.dr[j];
                ((f32x4*)(dx + (size_t)m * d))[i] = o;
            }
        }
        cur = nxt;
    }
    // cross-wave reduction of the column sums, then one atomic per column per block
#pragma unroll
    for (int j = 0; j < VPL; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[wave][0][(j * 64 + lane) * 4 + k] = aw[j][k];
            red[wave][1][(j * 64 + lane) * 4 + k] = ab[j][k];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        // column c lives at vector i = c/4 -> (j = i/64, lane = i%64), slot k = c%4
        const int i = c >> 2, idx = ((i >> 6) * 64 + (i & 63)) * 4 + (c & 3);
        const float sw = red[0][0][idx] + red[1][0][idx] + red[2][0][idx] + red[3][0][idx];
        const float sb = red[0][1][idx] + red[1][1][idx] + red[2][1][idx] + red[3][1][idx];
        if (ws) {
            ws[(size_t)blockIdx.x * 2 * d + c] = sw;
            ws[(size_t)blockIdx.x * 2 * d + d + c] = sb;
        }
    }
}

}  // namespace

extern "C" int vlmo_ln_fwd(const float* x, const float* w, const float* b, void* y, int out_f32, float* mean,
                           float* rstd, const int32_t* rowmap, int M, int d, float eps, hipStream_t stream) {
    VLMO_CHECK_ARG(x && w && b && y, "vlmo_ln_fwd: null pointer");
    VLMO_CHECK_ARG(M > 0 && d > 0 && d % 4 == 0 && d <= 1024, "vlmo_ln_fwd: need 0 < d <= 1024, d %% 4 == 0 (d=%d, M=%d)", d, M);
    const int vpl = (d / 4 + 63) / 64;
    const int grid = min((M + 3) / 4, 8192);
#define LNF(V)                                                                                        \
    if (out_f32)                                                                                      \
        hipLaunchKernelGGL((ln_fwd_kernel<V, true>), dim3(grid), dim3(256), 0, stream, x, w, b, y, mean, rstd, rowmap, M, d, eps); \
    else                                                                                              \
        hipLaunchKernelGGL((ln_fwd_kernel<V, false>), dim3(grid), dim3(256), 0, stream, x, w, b, y, mean, rstd, rowmap, M, d, eps);
    switch (vpl) {
        case 1: LNF(1) break;
        case 2: LNF(2) break;
        case 3: LNF(3) break;
        default: LNF(4) break;
    }
#undef LNF
    VLMO_CHECK_LAUNCH("vlmo_ln_fwd");
    return 0;
}

extern "C" int vlmo_ln_bwd(const void* dy, int dy_f32, const int32_t* rowmap, const float* x, const float* w,
                           const float* mean, const float* rstd, const float* dres, float* dx, float* dw, float* db,
                           int M, int d, float* ws, int64_t ws_bytes, hipStream_t stream) {
    VLMO_CHECK_ARG(dy && x && w && mean && rstd && dx, "vlmo_ln_bwd: null pointer");
    VLMO_CHECK_ARG(M > 0 && d > 0 && d % 4 == 0 && d <= 1024, "vlmo_ln_bwd: need 0 < d <= 1024, d %% 4 == 0 (d=%d, M=%d)", d, M);
    const bool need_w = dw || db;
    VLMO_CHECK_ARG(!need_w || (ws && ws_bytes >= reduce_ws_need(2 * d)),
                   "vlmo_ln_bwd: workspace too small (need %lld bytes)", (long long)reduce_ws_need(2 * d));
    if (!need_w) ws = nullptr;
    const int vpl = (d / 4 + 63) / 64;
    int rpb = (M + VLMO_MAX_PARTIAL_BLOCKS - 1) / VLMO_MAX_PARTIAL_BLOCKS;
    rpb = ((rpb + 3) / 4) * 4;
    if (rpb < 8) rpb = 8;
    const int grid = (M + rpb - 1) / rpb;
#define LNB(V)                                                                                         \
    if (dy_f32)                                                                                        \
        hipLaunchKernelGGL((ln_bwd_kernel<V, true>), dim3(grid), dim3(256), 0, stream, dy, rowmap, x, w, mean, rstd, dres, dx, ws, M, d, rpb); \
    else                                                                                               \
        hipLaunchKernelGGL((ln_bwd_kernel<V, false>), dim3(grid), dim3(256), 0, stream, dy, rowmap, x, w, mean, rstd, dres, dx, ws, M, d, rpb);
    switch (vpl) {
        case 1: LNB(1) break;
        case 2: LNB(2) break;
        case 3: LNB(3) break;
        default: LNB(4) break;
    }
#undef LNB
    VLMO_CHECK_LAUNCH("vlmo_ln_bwd");
    if (need_w) return reduce_partials(ws, grid, 2 * d, dw, d, db, stream);
    return 0;
}
