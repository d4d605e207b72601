// LayerNorm forward/backward, one 64-lane wavefront per token row, fp32 math.
// Reference: LayerNorm factory vlmo.py:26-36 bound with eps=1e-12 at
// vlmo_module.py:21-23; used at vlmo.py:188,192 (norm1/norm2) and :413 (norm).
// HBM-bound: each row is read once (16 B/lane vector loads) and kept in
// registers for both the statistics and the normalisation.
#include "common.h"
#include "vlmo_hip.h"
#include <stdlib.h>

namespace {

template <int VPL, bool OUT_F32>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ b, void* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                     const int32_t* __restrict__ rowmap, int M, int d, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d >> 2;
    // weight and bias stay in registers for all rows of the wave; TWO rows are in flight per wave (both rows' loads are
    // issued before any arithmetic): the kernel is HBM-bound and a wave's bytes in flight are what it has to offer
    f32x4 ww[VPL], bb[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int i = min(lane + 64 * j, nv - 1);
        ww[j] = ((const f32x4*)w)[i];
        bb[j] = ((const f32x4*)b)[i];
    }
    const int stride = gridDim.x * 4;
    for (int m0 = blockIdx.x * 4 + wave; m0 < M; m0 += 2 * stride) {
        f32x4 v[2][VPL];
        int mr[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            mr[r] = m0 + r * stride;
            const f32x4* xr = (const f32x4*)(x + (size_t)min(mr[r], M - 1) * d);
#pragma unroll
            for (int j = 0; j < VPL; ++j) v[r][j] = xr[min(lane + 64 * j, nv - 1)];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int m = mr[r];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < VPL; ++j)
                if (lane + 64 * j < nv) s += v[r][j][0] + v[r][j][1] + v[r][j][2] + v[r][j][3];
            const float mu = wave_sum(s) / d;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < VPL; ++j)
                if (lane + 64 * j < nv) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t = v[r][j][k] - mu;
                        q += t * t;
                    }
                }
            const float rs = rsqrtf(wave_sum(q) / d + eps);
            if (m < M) {
                if (lane == 0) {
                    if (mean) mean[m] = mu;
                    if (rstd) rstd[m] = rs;
                }
                const int om = rowmap ? rowmap[m] : m;
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const int i = lane + 64 * j;
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = (v[r][j][k] - mu) * rs * ww[j][k] + bb[j][k];
                    if (i < nv) {
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
    }
}

// residual branch fed by the LayerNorm's input gradient (fused form): z = drop(branch) is what forward added as
// x + gamma * rs * z, so dz = mask * dx * gamma * rs, dgamma = sum dx * rs * z, dbias = sum dz  (resid_bwd_kernel)
struct ResidArgs {
    const bf16* zd;
    const float* gamma;
    const float* row_scale;
    const int32_t* row_index;
    bf16* dz;
    uint32_t thresh;
    float inv_keep;
    uint64_t seed;
    // rows [seg, M) form a second segment with its own dropout stream (seed1) whose counters restart at row seg:
    // the per-modality experts of a block below the fusion layer (their FFN residual branches were produced by
    // separate problems of a grouped GEMM).  seg = M: one segment.  No workgroup straddles the boundary, so the
    // column partials of the two segments are separate rows of the workspace.
    int seg;
    uint64_t seed1;
};

template <int VPL, bool DY_F32, bool RESID, bool NTL = false>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, const int32_t* __restrict__ rowmap,
                                                     const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ dres, float* __restrict__ dx,
                                                     float* __restrict__ ws, int M, int d, int rows_per_block,
                                                     const ResidArgs ra) {
    constexpr int NCOL = RESID ? 4 : 2;      // column partials per block: dw, db (, dgamma, dbias)
    __shared__ float red[4][NCOL][VPL * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = d >> 2;
    f32x4 aw[VPL], ab[VPL], ww[VPL], ag[VPL], az[VPL], gg[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        aw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[j] = ag[j] = az[j] = aw[j];
        const int i = lane + 64 * j;
        ww[j] = (i < nv) ? ((const f32x4*)w)[i] : aw[j];
        gg[j] = f32x4{1.f, 1.f, 1.f, 1.f};
        if (RESID && ra.gamma && i < nv) gg[j] = ((const f32x4*)ra.gamma)[i];
    }
    int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    uint64_t drop_seed = 0;
    int drop_row0 = 0;
    if constexpr (RESID) {
        const int nb0 = (ra.seg + rows_per_block - 1) / rows_per_block;
        drop_seed = ra.seed;
        if ((int)blockIdx.x < nb0) {
            r1 = min(ra.seg, r0 + rows_per_block);
        } else {
            r0 = ra.seg + ((int)blockIdx.x - nb0) * rows_per_block;
            r1 = min(M, r0 + rows_per_block);
            drop_seed = ra.seed1;
            drop_row0 = ra.seg;
        }
    }
    // Everything a row needs from HBM (dy, x, the incoming residual gradient, mean/rstd) is fetched one row
    // AHEAD of the arithmetic: beside the weight-gradient GEMMs of the side stream this kernel gets one or two
    // waves per SIMD, so its speed is (bytes in flight per wave) / latency, not occupancy.
    struct Row {
        f32x4 dy[VPL], x[VPL], dr[VPL];
        bf16x4 z[VPL];
        float mu, rs, scale;
    };
    auto fetch = [&](int m, Row& r) {
        const int sm = rowmap ? rowmap[m] : m;
        r.mu = mean[m];
        r.rs = rstd[m];
        if constexpr (RESID) r.scale = ra.row_scale ? ra.row_scale[ra.row_index ? ra.row_index[m] : m] : 1.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                // NTL: every operand row is read exactly once by this kernel and by nobody after it -- streamed past the caches
                auto ld = [](const auto* p) { return NTL ? __builtin_nontemporal_load(p) : *p; };
                if constexpr (DY_F32) {
                    r.dy[j] = ld((const f32x4*)((const float*)dy + (size_t)sm * d) + i);
                } else {
                    const bf16x4 t = ld((const bf16x4*)((const bf16*)dy + (size_t)sm * d) + i);
                    r.dy[j] = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
                }
                r.x[j] = ld((const f32x4*)(x + (size_t)m * d) + i);
                r.dr[j] = dres ? ld((const f32x4*)(dres + (size_t)m * d) + i) : f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (RESID) r.z[j] = ld((const bf16x4*)(ra.zd + (size_t)m * d) + i);
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
                if constexpr (RESID) {
                    const f32x4 orr = o * cur.scale;
                    f32x4 v = o * gg[j] * cur.scale;      // same association as resid_bwd_kernel: (dx * gamma) * rs
                    if (ra.thresh) {
                        const uint64_t bits = drop_bits4(drop_seed, ((uint64_t)(m - drop_row0) * d + 4 * i) >> 2);
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = drop_keep(bits, k, ra.thresh) ? v[k] * ra.inv_keep : 0.f;
                    }
                    const bf16x4 vb = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                    ((bf16x4*)(ra.dz + (size_t)m * d))[i] = vb;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        ag[j][k] += orr[k] * (float)cur.z[j][k];
                        az[j][k] += v[k];
                    }
                }
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
            if constexpr (RESID) {
                red[wave][2][(j * 64 + lane) * 4 + k] = ag[j][k];
                red[wave][3][(j * 64 + lane) * 4 + k] = az[j][k];
            }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) {
        // column c lives at vector i = c/4 -> (j = i/64, lane = i%64), slot k = c%4
        const int i = c >> 2, idx = ((i >> 6) * 64 + (i & 63)) * 4 + (c & 3);
        if (ws) {
#pragma unroll
            for (int q = 0; q < NCOL; ++q)
                ws[((size_t)blockIdx.x * NCOL + q) * d + c] = red[0][q][idx] + red[1][q][idx] + red[2][q][idx] + red[3][q][idx];
        }
    }
}

}  // namespace

extern "C" int vlmo_ln_fwd(const float* x, const float* w, const float* b, void* y, int out_f32, float* mean,
                           float* rstd, const int32_t* rowmap, int M, int d, float eps, hipStream_t stream) {
    VLMO_CHECK_ARG(x && w && b && y, "vlmo_ln_fwd: null pointer");
    VLMO_CHECK_ARG(M > 0 && d > 0 && d % 4 == 0 && d <= 1024, "vlmo_ln_fwd: need 0 < d <= 1024, d %% 4 == 0 (d=%d, M=%d)", d, M);
    const int vpl = (d / 4 + 63) / 64;
    const int grid = min((M + 7) / 8, 8192);
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

namespace {
int ln_bwd_launch(const void* dy, int dy_f32, const int32_t* rowmap, const float* x, const float* w, const float* mean,
                  const float* rstd, const float* dres, float* dx, float* dw, float* db, int M, int d, float* ws,
                  int64_t ws_bytes, const ResidArgs* ra, float* dgamma, float* dbias, hipStream_t stream) {
    VLMO_CHECK_ARG(dy && x && w && mean && rstd && dx, "vlmo_ln_bwd: null pointer");
    VLMO_CHECK_ARG(M > 0 && d > 0 && d % 4 == 0 && d <= 1024, "vlmo_ln_bwd: need 0 < d <= 1024, d %% 4 == 0 (d=%d, M=%d)", d, M);
    const int ncol = ra ? 4 : 2;
    const bool need_w = dw || db || ra;
    VLMO_CHECK_ARG(!need_w || (ws && ws_bytes >= reduce_ws_need(ncol * d)),
                   "vlmo_ln_bwd: workspace too small (need %lld bytes)", (long long)reduce_ws_need(ncol * d));
    if (!need_w) ws = nullptr;
    const int vpl = (d / 4 + 63) / 64;
    int rpb = (M + VLMO_MAX_PARTIAL_BLOCKS - 1) / VLMO_MAX_PARTIAL_BLOCKS;
    rpb = ((rpb + 3) / 4) * 4;
    if (rpb < 8) rpb = 8;
    int grid = (M + rpb - 1) / rpb;
    if (ra && ra->seg < M) {        // two segments: workgroups do not straddle the boundary
        for (;; rpb += 4) {
            grid = (ra->seg + rpb - 1) / rpb + (M - ra->seg + rpb - 1) / rpb;
            if (grid <= VLMO_MAX_PARTIAL_BLOCKS) break;
        }
    }
    const ResidArgs none{};
    static const bool ntl = getenv("VLMO_LN_BWD_NT") && atoi(getenv("VLMO_LN_BWD_NT")) != 0;     // measurement aid
#define LNB(V)                                                                                         \
    if (ra && ntl)                                                                                     \
        hipLaunchKernelGGL((ln_bwd_kernel<V, false, true, true>), dim3(grid), dim3(256), 0, stream, dy, rowmap, x, w, mean, rstd, dres, dx, ws, M, d, rpb, *ra); \
    else if (ra)                                                                                       \
        hipLaunchKernelGGL((ln_bwd_kernel<V, false, true>), dim3(grid), dim3(256), 0, stream, dy, rowmap, x, w, mean, rstd, dres, dx, ws, M, d, rpb, *ra); \
    else if (dy_f32)                                                                                   \
        hipLaunchKernelGGL((ln_bwd_kernel<V, true, false>), dim3(grid), dim3(256), 0, stream, dy, rowmap, x, w, mean, rstd, dres, dx, ws, M, d, rpb, none); \
    else                                                                                               \
        hipLaunchKernelGGL((ln_bwd_kernel<V, false, false>), dim3(grid), dim3(256), 0, stream, dy, rowmap, x, w, mean, rstd, dres, dx, ws, M, d, rpb, none);
    switch (vpl) {
        case 1: LNB(1) break;
        case 2: LNB(2) break;
        case 3: LNB(3) break;
        default: LNB(4) break;
    }
#undef LNB
    VLMO_CHECK_LAUNCH("vlmo_ln_bwd");
    if (need_w) return reduce_partials(ws, grid, ncol * d, dw, d, db, stream, dgamma, dbias);
    return 0;
}
}  // namespace

extern "C" int vlmo_ln_bwd(const void* dy, int dy_f32, const int32_t* rowmap, const float* x, const float* w,
                           const float* mean, const float* rstd, const float* dres, float* dx, float* dw, float* db,
                           int M, int d, float* ws, int64_t ws_bytes, hipStream_t stream) {
    return ln_bwd_launch(dy, dy_f32, rowmap, x, w, mean, rstd, dres, dx, dw, db, M, d, ws, ws_bytes, nullptr, nullptr,
                         nullptr, stream);
}

extern "C" int vlmo_ln_resid_bwd(const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
                                 const float* dres, float* dx, float* dw, float* db, const void* zd, const float* gamma,
                                 const float* row_scale, const int32_t* row_index, void* dz, float* dgamma, float* dbias,
                                 uint32_t drop_thresh, float inv_keep, uint64_t seed, int M, int d, float* ws,
                                 int64_t ws_bytes, hipStream_t stream) {
    VLMO_CHECK_ARG(zd && dz, "vlmo_ln_resid_bwd: null pointer");
    const ResidArgs ra{(const bf16*)zd, gamma, row_scale, row_index, (bf16*)dz, drop_thresh, inv_keep, seed, M, 0};
    return ln_bwd_launch(dy, 0, nullptr, x, w, mean, rstd, dres, dx, dw, db, M, d, ws, ws_bytes, &ra, dgamma, dbias, stream);
}

// LayerNorm backward of one block fused with the residual-branch backward of the block BELOW it (block.hip: norm1
// of block i + the FFN branch of block i-1, whose incoming gradient is exactly the dx this kernel writes).  The
// branch may consist of two row segments [0, seg) and [seg, M) with their own dropout streams (per-modality experts);
// the caller folds the bias-gradient partials of the segments itself: workspace rows [0, *nb0) belong to segment 0,
// [*nb0, *nblk) to segment 1, columns [3d, 4d).
int ln_resid_seg_bwd(const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
                     const float* dres, float* dx, float* dw, float* db, const void* zd, const float* gamma,
                     const float* row_scale, const int32_t* row_index, void* dz, float* dgamma, uint32_t drop_thresh,
                     float inv_keep, uint64_t seed0, uint64_t seed1, int seg, int M, int d, float* ws, int64_t ws_bytes,
                     int* nb0, int* nblk, hipStream_t stream) {
    VLMO_CHECK_ARG(zd && dz && seg >= 1 && seg <= M, "ln_resid_seg_bwd: bad arguments");
    const ResidArgs ra{(const bf16*)zd, gamma, row_scale, row_index, (bf16*)dz, drop_thresh, inv_keep, seed0, seg, seed1};
    int rpb = (M + VLMO_MAX_PARTIAL_BLOCKS - 1) / VLMO_MAX_PARTIAL_BLOCKS;
    rpb = ((rpb + 3) / 4) * 4;
    if (rpb < 8) rpb = 8;
    int grid = (M + rpb - 1) / rpb;
    if (seg < M)
        for (;; rpb += 4) {
            grid = (seg + rpb - 1) / rpb + (M - seg + rpb - 1) / rpb;
            if (grid <= VLMO_MAX_PARTIAL_BLOCKS) break;
        }
    *nb0 = (seg + rpb - 1) / rpb;
    *nblk = grid;
    return ln_bwd_launch(dy, 0, nullptr, x, w, mean, rstd, dres, dx, dw, db, M, d, ws, ws_bytes, &ra, dgamma, nullptr, stream);
}
