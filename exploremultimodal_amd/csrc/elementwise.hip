// HBM-bound glue kernels of the VLMo hot path: residual-branch backward,
// bias-gradient column sums, fp32 -> bf16/f16 weight shadows (+ transposes),
// image patchify / embedding finish (vlmo.py:298-319) and the text embedding
// (vlmo.py:321-324 + BertEmbeddings).  All vectorised 8-16 B per lane.
#include "common.h"
#include "vlmo_hip.h"
#include <stdarg.h>
#include <stdio.h>

// ------------------------------------------------------------- error plumbing
static thread_local char g_err[512] = "";
void vlmo_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* vlmo_last_error(void) { return g_err; }
extern "C" int vlmo_abi_version(void) { return 5; }

namespace {

// out_k[c - k*n0] += sum_b ws[b][c] for k = c / n0 (up to 4 outputs of n0 columns each).  block (64, 4), grid (ncols/64, S)
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ ws, int nblk, int ncols,
                                                              float* __restrict__ out0, int n0,
                                                              float* __restrict__ out1, int rows_per_slice,
                                                              float* __restrict__ out2, float* __restrict__ out3) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int r0 = blockIdx.y * rows_per_slice, r1 = min(nblk, r0 + rows_per_slice);
    float a = 0.f;
    if (c < ncols)
        for (int r = r0 + threadIdx.y; r < r1; r += 4) a += ws[(size_t)r * ncols + c];
    red[threadIdx.y][threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.y == 0 && c < ncols) {
        const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        const int k = c / n0;
        float* base = k == 0 ? out0 : (k == 1 ? out1 : (k == 2 ? out2 : out3));
        if (base) atomicAdd(base + (c - k * n0), t);
    }
}

template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<bf16>(const bf16* p) {
    const bf16x4 v = *(const bf16x4*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <> __device__ __forceinline__ f32x4 ld4<f16>(const f16* p) {
    const f16x4 v = *(const f16x4*)p;
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<bf16>(bf16* p, f32x4 v) {
    *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
}
template <> __device__ __forceinline__ void st4<f16>(f16* p, f32x4 v) {
    *(f16x4*)p = f16x4{(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
}

// block = (d/4, RY); thread owns 4 columns and every RY-th row of its block's chunk,
// two rows in flight per thread (independent loads) for memory-level parallelism
__global__ void resid_bwd_kernel(const float* __restrict__ dx, const bf16* __restrict__ zd,
                                 const float* __restrict__ gamma, const float* __restrict__ row_scale,
                                 const int32_t* __restrict__ row_index, bf16* __restrict__ dz,
                                 float* __restrict__ ws, bool need_gamma, int M,
                                 int d, int rows_per_block, uint32_t thresh, float inv_keep, uint64_t seed) {
    extern __shared__ float red[];   // [RY][2][d]
    const int c4 = threadIdx.x * 4, RY = blockDim.y;
    const f32x4 g = gamma ? *(const f32x4*)(gamma + c4) : f32x4{1.f, 1.f, 1.f, 1.f};
    f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = ag;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int mb = r0 + threadIdx.y; mb < r1; mb += 2 * RY) {
        f32x4 gx[2], z[2];
        float rs[2];
        bool ok[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = mb + k * RY;
            ok[k] = m < r1;
            const size_t o = (size_t)(ok[k] ? m : mb) * d + c4;
            gx[k] = *(const f32x4*)(dx + o);
            z[k] = need_gamma ? ld4<bf16>(zd + o) : f32x4{0.f, 0.f, 0.f, 0.f};
            const int mr = ok[k] ? m : mb;
            rs[k] = row_scale ? row_scale[row_index ? row_index[mr] : mr] : 1.f;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (!ok[k]) continue;
            const int m = mb + k * RY;
            const size_t o = (size_t)m * d + c4;
            f32x4 v = gx[k] * g * rs[k];
            if (thresh) {
                const uint64_t bits = drop_bits4(seed, ((uint64_t)m * d + c4) >> 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = drop_keep(bits, j, thresh) ? v[j] * inv_keep : 0.f;
            }
            st4<bf16>(dz + o, v);
            ag += gx[k] * rs[k] * z[k];
            ab += v;
        }
    }
    *(f32x4*)(red + (threadIdx.y * 2 + 0) * d + c4) = ag;
    *(f32x4*)(red + (threadIdx.y * 2 + 1) * d + c4) = ab;
    __syncthreads();
    if (threadIdx.y == 0) {
        for (int y = 1; y < RY; ++y) {
            ag += *(const f32x4*)(red + (y * 2 + 0) * d + c4);
            ab += *(const f32x4*)(red + (y * 2 + 1) * d + c4);
        }
        *(f32x4*)(ws + (size_t)blockIdx.x * 2 * d + c4) = ag;
        *(f32x4*)(ws + (size_t)blockIdx.x * 2 * d + d + c4) = ab;
    }
}

// block (64, 4): 64 column groups of 8 x 4 row lanes, 4 rows in flight per thread
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int ld, float* __restrict__ ws, int M,
                                                     int N, int rows_per_block) {
    typedef typename Elem<T>::v8 v8;
    __shared__ float red[4][64][8];
    const int c8 = (blockIdx.x * 64 + threadIdx.x) * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
    if (c8 < N) {
        for (int mb = r0 + threadIdx.y; mb < r1; mb += 16) {
            v8 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = mb + 4 * k;
                v[k] = *(const v8*)(x + (size_t)(m < r1 ? m : mb) * ld + c8);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (mb + 4 * k < r1)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += (float)v[k][j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.y][threadIdx.x][j] = acc[j];
    __syncthreads();
    if (threadIdx.y == 0 && c8 < N) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            ws[(size_t)blockIdx.y * N + c8 + j] = red[0][threadIdx.x][j] + red[1][threadIdx.x][j] +
                                                  red[2][threadIdx.x][j] + red[3][threadIdx.x][j];
    }
}

// Batched column work (vlmo_colwork_multi): job q owns the workgroups [b0[q], b0[q+1]), laid out gx[q] column
// groups x row slices of rps[q] rows.  block (64, 4).
constexpr int MAX_COL_JOBS = 32;
struct ColMulti {
    int n;
    int b0[MAX_COL_JOBS + 1];
    int gx[MAX_COL_JOBS], rps[MAX_COL_JOBS];
    VlmoColJob j[MAX_COL_JOBS];
};
template <typename T>
__global__ __launch_bounds__(256) void colwork_multi_kernel(const ColMulti cm) {
    typedef typename Elem<T>::v8 v8;
    __shared__ float red[4][64][8];
    int ji = 0;
#pragma unroll
    for (int q = 1; q < MAX_COL_JOBS; ++q)
        if (q < cm.n && (int)blockIdx.x >= cm.b0[q]) ji = q;
    ji = __builtin_amdgcn_readfirstlane(ji);
    const VlmoColJob& J = cm.j[ji];
    const int local = blockIdx.x - cm.b0[ji], gx = cm.gx[ji], rps = cm.rps[ji];
    const int bx = local % gx, by = local / gx;
    const int r0 = by * rps, r1 = min(J.rows, r0 + rps);
    const int tx = threadIdx.x, ty = threadIdx.y;
    if (J.kind == 0) {
        const float* ws = (const float*)J.src;
        const int c = bx * 64 + tx;
        float a = 0.f;
        if (c < J.ncols)
            for (int r = r0 + ty; r < r1; r += 4) a += ws[(size_t)r * J.ld + c];
        red[ty][tx][0] = a;
        __syncthreads();
        if (ty == 0 && c < J.ncols) {
            const float t = red[0][tx][0] + red[1][tx][0] + red[2][tx][0] + red[3][tx][0];
            const int k = c / J.n0;
            float* base = J.out[k];
            if (base) atomicAdd(base + (c - k * J.n0), t);
        }
    } else {
        const T* x = (const T*)J.src;
        const int c8 = (bx * 64 + tx) * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (c8 < J.ncols) {
            for (int mb = r0 + ty; mb < r1; mb += 16) {
                v8 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int m = mb + 4 * k;
                    v[k] = *(const v8*)(x + (size_t)(m < r1 ? m : mb) * J.ld + c8);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (mb + 4 * k < r1)
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) acc[jj] += (float)v[k][jj];
            }
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) red[ty][tx][jj] = acc[jj];
        __syncthreads();
        if (ty == 0 && c8 < J.ncols) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int c = c8 + jj;
                const int k = c / J.n0;
                float* base = J.out[k];
                if (base) atomicAdd(base + (c - k * J.n0), red[0][tx][jj] + red[1][tx][jj] + red[2][tx][jj] + red[3][tx][jj]);
            }
        }
    }
}

// 32x32 tile transpose through LDS; block (32, 8)
template <typename T>
__global__ __launch_bounds__(256) void cast_weight_kernel(const float* __restrict__ src, int rows, int cols,
                                                          T* __restrict__ dst, T* __restrict__ dstT) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + threadIdx.y + 8 * k, c = c0 + threadIdx.x;
        float v = 0.f;
        if (r < rows && c < cols) {
            v = src[(size_t)r * cols + c];
            if (dst) dst[(size_t)r * cols + c] = (T)v;
        }
        tile[threadIdx.y + 8 * k][threadIdx.x] = v;
    }
    if (!dstT) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + threadIdx.y + 8 * k, r = r0 + threadIdx.x;
        if (r < rows && c < cols) dstT[(size_t)c * rows + r] = (T)tile[threadIdx.x][threadIdx.y + 8 * k];
    }
}

// All stale weight shadows of a step in ONE launch (after an optimizer step every weight of the model is stale: ~100
// launches of the kernel above at ~5 us of launch floor each were 0.65 ms per step).  64 x 64 tiles, fp32 rows read as
// 16-byte pieces, both bf16 copies written as 8-byte pieces of full rows (the transposed one through LDS).
struct CastJobs {
    static constexpr int MAXJ = 72;
    int n;
    int tile0[MAXJ + 1];            // first tile of job j (prefix sums)
    const float* src[MAXJ];
    void* dst[MAXJ];
    void* dstT[MAXJ];
    int rows[MAXJ], cols[MAXJ];
};
template <typename T>
__global__ __launch_bounds__(256) void cast_weight_multi_kernel(const CastJobs J) {
    __shared__ float tile[64][65];
    int j = 0;
    for (int q = 1; q < J.n; ++q)
        if ((int)blockIdx.x >= J.tile0[q]) j = q;
    j = __builtin_amdgcn_readfirstlane(j);
    const int rows = J.rows[j], cols = J.cols[j];
    const float* src = J.src[j];
    T* dst = (T*)J.dst[j];
    T* dstT = (T*)J.dstT[j];
    const int t = blockIdx.x - J.tile0[j], tc = (cols + 63) / 64;
    const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;        // 16 x 4-column pieces, 16 rows per pass
    typedef typename Elem<T>::v4 v4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 16 * k, c = c0 + 4 * tx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
            if (c + 3 < cols && (cols & 3) == 0) {
                v = *(const f32x4*)(src + (size_t)r * cols + c);
                if (dst) *(v4*)(dst + (size_t)r * cols + c) = v4{(T)v[0], (T)v[1], (T)v[2], (T)v[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < cols) {
                        v[e] = src[(size_t)r * cols + c + e];
                        if (dst) dst[(size_t)r * cols + c + e] = (T)v[e];
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[ty + 16 * k][4 * tx + e] = v[e];
    }
    if (!dstT) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 16 * k, r = r0 + 4 * tx;           // output row = source column
        if (c < cols) {
            if (r + 3 < rows && (rows & 3) == 0) {
                *(v4*)(dstT + (size_t)c * rows + r) = v4{(T)tile[4 * tx][ty + 16 * k], (T)tile[4 * tx + 1][ty + 16 * k],
                                                         (T)tile[4 * tx + 2][ty + 16 * k], (T)tile[4 * tx + 3][ty + 16 * k]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (r + e < rows) dstT[(size_t)c * rows + r + e] = (T)tile[4 * tx + e][ty + 16 * k];
            }
        }
    }
}

__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, bf16* __restrict__ out, int B, int C,
                                                       int H, int W, int p, long total4) {
    const int gw = W / p, gh = H / p, kcols = C * p * p;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const long o = i * 4;
        const long row = o / kcols;
        const int col = (int)(o % kcols);
        const int c = col / (p * p), ky = (col / p) % p, kx = col % p;
        const int b = (int)(row / (gh * gw)), pr = (int)(row % (gh * gw));
        const int py = pr / gw, px = pr % gw;
        const f32x4 v = *(const f32x4*)(img + (((size_t)b * C + c) * H + py * p + ky) * W + px * p + kx);
        st4<bf16>(out + o, v);
    }
}

// grid (npatch+1, B), block d/4 threads
__global__ void embed_img_finish_kernel(const bf16* __restrict__ proj, const float* __restrict__ cls_tok,
                                        const float* __restrict__ mask_tok, const float* __restrict__ pos,
                                        const float* __restrict__ type_row, const uint8_t* __restrict__ masked,
                                        float* __restrict__ x, int npatch, int d, uint32_t thresh, float inv_keep,
                                        uint64_t seed) {
    const int t = blockIdx.x, b = blockIdx.y, c4 = threadIdx.x * 4;
    f32x4 v;
    if (t == 0)
        v = *(const f32x4*)(cls_tok + c4);
    else if (masked && masked[(size_t)b * npatch + t - 1])
        v = *(const f32x4*)(mask_tok + c4);
    else
        v = ld4<bf16>(proj + ((size_t)b * npatch + t - 1) * d + c4);
    v += *(const f32x4*)(pos + (size_t)t * d + c4);
    const size_t o = ((size_t)b * (npatch + 1) + t) * d + c4;
    if (thresh) {
        const uint64_t bits = drop_bits4(seed, o >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = drop_keep(bits, j, thresh) ? v[j] * inv_keep : 0.f;
    }
    v += *(const f32x4*)(type_row + c4);
    *(f32x4*)(x + o) = v;
}

// grid (npatch+1), block d/4: one workgroup per token position sums over the batch
__global__ void embed_img_bwd_kernel(const float* __restrict__ dx, const uint8_t* __restrict__ masked,
                                     bf16* __restrict__ dproj, float* __restrict__ dcls, float* __restrict__ dmask,
                                     float* __restrict__ dpos, float* __restrict__ dtype_row, int B, int npatch, int d,
                                     uint32_t thresh, float inv_keep, uint64_t seed) {
    const int t = blockIdx.x, c4 = threadIdx.x * 4;
    f32x4 apos = {0.f, 0.f, 0.f, 0.f}, atype = apos, amask = apos;
    for (int b = 0; b < B; ++b) {
        const size_t o = ((size_t)b * (npatch + 1) + t) * d + c4;
        f32x4 g = *(const f32x4*)(dx + o);
        atype += g;
        if (thresh) {
            const uint64_t bits = drop_bits4(seed, o >> 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = drop_keep(bits, j, thresh) ? g[j] * inv_keep : 0.f;
        }
        apos += g;
        if (t > 0) {
            const bool mk = masked && masked[(size_t)b * npatch + t - 1];
            if (mk) amask += g;
            st4<bf16>(dproj + ((size_t)b * npatch + t - 1) * d + c4, mk ? f32x4{0.f, 0.f, 0.f, 0.f} : g);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (dpos) dpos[(size_t)t * d + c4 + j] += apos[j];   // this workgroup owns position t
        if (dtype_row) atomicAdd(dtype_row + c4 + j, atype[j]);
        if (t == 0 && dcls) dcls[c4 + j] += apos[j];
        if (t > 0 && dmask && masked) atomicAdd(dmask + c4 + j, amask[j]);
    }
}

template <int VPL>
__global__ __launch_bounds__(256) void embed_txt_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ word,
                                                            const float* __restrict__ pos, const float* __restrict__ btype0,
                                                            const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                            const float* __restrict__ type0, float* __restrict__ x,
                                                            float* __restrict__ xhat, float* __restrict__ rstd, int ntok,
                                                            int T, int d, float eps, uint32_t thresh, float inv_keep,
                                                            uint64_t seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nv = d >> 2;
    for (int m = blockIdx.x * 4 + wave; m < ntok; m += gridDim.x * 4) {
        const int64_t id = ids[m];
        const int t = m % T;
        f32x4 v[VPL];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                v[j] = ((const f32x4*)(word + (size_t)id * d))[i] + ((const f32x4*)btype0)[i] +
                       ((const f32x4*)(pos + (size_t)t * d))[i];
                s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
            } else {
                v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const float mu = wave_sum(s) / d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j)
            if (lane + 64 * j < nv)
#pragma unroll
                for (int k = 0; k < 4; ++k) q += (v[j][k] - mu) * (v[j][k] - mu);
        const float rs = rsqrtf(wave_sum(q) / d + eps);
        if (lane == 0 && rstd) rstd[m] = rs;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                const f32x4 xh = (v[j] - mu) * rs;
                if (xhat) ((f32x4*)(xhat + (size_t)m * d))[i] = xh;
                f32x4 o = xh * ((const f32x4*)ln_w)[i] + ((const f32x4*)ln_b)[i];
                if (thresh) {
                    const uint64_t bits = drop_bits4(seed, ((uint64_t)m * d >> 2) + i);
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = drop_keep(bits, k, thresh) ? o[k] * inv_keep : 0.f;
                }
                ((f32x4*)(x + (size_t)m * d))[i] = o + ((const f32x4*)type0)[i];
            }
        }
    }
}

template <int VPL>
__global__ __launch_bounds__(256) void embed_txt_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ ids,
                                                            const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                            const float* __restrict__ ln_w, float* __restrict__ dword,
                                                            float* __restrict__ dpos, float* __restrict__ dbtype0,
                                                            float* __restrict__ dln_w, float* __restrict__ dln_b,
                                                            float* __restrict__ dtype0, int ntok, int T, int d,
                                                            int rows_per_block, uint32_t thresh, float inv_keep,
                                                            uint64_t seed) {
    __shared__ float red[4][4][VPL * 256];   // [wave][lnw, lnb, type0, btype0 (= position row of this workgroup)]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nv = d >> 2;
    f32x4 aw[VPL], ab[VPL], at[VPL], ae[VPL], ww[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        aw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        ab[j] = aw[j];
        at[j] = aw[j];
        ae[j] = aw[j];
        ww[j] = (lane + 64 * j < nv) ? ((const f32x4*)ln_w)[lane + 64 * j] : aw[j];
    }
    // a workgroup owns ONE position t and `rows_per_block` sequences: the position-embedding gradient of its rows is one
    // register accumulator (ae) folded once per workgroup, not an atomic per row (64 sequences on every position row:
    // 106 -> ~40 us); the word-embedding rows are scattered per token as before
    const int t = blockIdx.x % T;
    const int b0 = (blockIdx.x / T) * rows_per_block, b1 = min(ntok / T, b0 + rows_per_block);
    for (int bq = b0 + wave; bq < b1; bq += 4) {
        const int m = bq * T + t;
        const int64_t id = ids[m];
        const float rs = rstd[m];
        f32x4 g[VPL], xh[VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int i = lane + 64 * j;
            g[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            xh[j] = g[j];
            if (i < nv) {
                f32x4 gy = ((const f32x4*)(dx + (size_t)m * d))[i];
                at[j] += gy;
                if (thresh) {
                    const uint64_t bits = drop_bits4(seed, ((uint64_t)m * d >> 2) + i);
#pragma unroll
                    for (int k = 0; k < 4; ++k) gy[k] = drop_keep(bits, k, thresh) ? gy[k] * inv_keep : 0.f;
                }
                xh[j] = ((const f32x4*)(xhat + (size_t)m * d))[i];
                aw[j] += gy * xh[j];
                ab[j] += gy;
                g[j] = gy * ww[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s1 += g[j][k];
                    s2 += g[j][k] * xh[j][k];
                }
            }
        }
        const float c1 = wave_sum(s1) / d, c2 = wave_sum(s2) / d;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int i = lane + 64 * j;
            if (i < nv) {
                const f32x4 de = rs * (g[j] - c1 - xh[j] * c2);
                ae[j] += de;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (dword && id != 0) atomicAdd(dword + (size_t)id * d + 4 * i + k, de[k]);   // padding_idx = 0
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VPL; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = (j * 64 + lane) * 4 + k;
            red[wave][0][idx] = aw[j][k];
            red[wave][1][idx] = ab[j][k];
            red[wave][2][idx] = at[j][k];
            red[wave][3][idx] = ae[j][k];
        }
    __syncthreads();
    float* outs[4] = {dln_w, dln_b, dtype0, dbtype0};
    for (int c = threadIdx.x; c < d; c += 256) {
        const int i = c >> 2, idx = ((i >> 6) * 64 + (i & 63)) * 4 + (c & 3);
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (outs[w]) atomicAdd(outs[w] + c, red[0][w][idx] + red[1][w][idx] + red[2][w][idx] + red[3][w][idx]);
        // dbtype0 and the position row both sum `de` (red[..][3]): the position row of THIS workgroup's t
        if (dpos) atomicAdd(dpos + (size_t)t * d + c, red[0][3][idx] + red[1][3][idx] + red[2][3][idx] + red[3][3][idx]);
    }
}

}  // namespace

thread_local PartialReduce* vlmo_defer_reduce = nullptr;

int reduce_partials(const float* ws, int nblk, int ncols, float* out0, int n0, float* out1, hipStream_t stream,
                    float* out2, float* out3) {
    if (vlmo_defer_reduce) {
        PartialReduce* r = vlmo_defer_reduce;
        vlmo_defer_reduce = nullptr;
        r->ws = ws, r->nblk = nblk, r->ncols = ncols, r->out0 = out0, r->n0 = n0, r->out1 = out1;
        r->out2 = out2, r->out3 = out3;
        return 0;
    }
    int slices = nblk >= 64 ? 8 : 1;
    const int rps = (nblk + slices - 1) / slices;
    slices = (nblk + rps - 1) / rps;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((ncols + 63) / 64, slices), dim3(64, 4), 0, stream, ws, nblk, ncols,
                       out0, n0, out1, rps, out2, out3);
    VLMO_CHECK_LAUNCH("reduce_partials");
    return 0;
}

extern "C" int64_t vlmo_reduce_ws_bytes(int ncols) { return reduce_ws_need(ncols); }

namespace {

int rows_per_block_for(int M, int target_blocks, int mult) {
    int r = (M + target_blocks - 1) / target_blocks;
    r = ((r + mult - 1) / mult) * mult;
    return r < mult ? mult : r;
}

}  // namespace

extern "C" int vlmo_resid_bwd(const float* dx, const void* zd, const float* gamma, const float* row_scale,
                              const int32_t* row_index, void* dz, float* dgamma, float* dbias, int M, int d,
                              uint32_t drop_thresh, float inv_keep,
                              uint64_t seed, float* ws, int64_t ws_bytes, hipStream_t stream) {
    VLMO_CHECK_ARG(dx && dz, "vlmo_resid_bwd: null pointer");
    VLMO_CHECK_ARG(!dgamma || zd, "vlmo_resid_bwd: dgamma needs zd");
    VLMO_CHECK_ARG(M > 0 && d % 4 == 0 && d >= 4 && d <= 4096, "vlmo_resid_bwd: bad shape M=%d d=%d", M, d);
    VLMO_CHECK_ARG(ws && ws_bytes >= reduce_ws_need(2 * d), "vlmo_resid_bwd: workspace too small (need %lld bytes)",
                   (long long)reduce_ws_need(2 * d));
    const int tx = d / 4;
    int ry = 1024 / tx;
    if (ry < 1) ry = 1;
    if (ry > 4) ry = 4;
    const int rpb = rows_per_block_for(M, VLMO_MAX_PARTIAL_BLOCKS, ry * 2);
    const int grid = (M + rpb - 1) / rpb;
    hipLaunchKernelGGL(resid_bwd_kernel, dim3(grid), dim3(tx, ry), ry * 2 * d * sizeof(float), stream, dx,
                       (const bf16*)zd, gamma, row_scale, row_index, (bf16*)dz, ws, dgamma != nullptr, M, d, rpb, drop_thresh,
                       inv_keep, seed);
    VLMO_CHECK_LAUNCH("vlmo_resid_bwd");
    if (dgamma || dbias) return reduce_partials(ws, grid, 2 * d, dgamma, d, dbias, stream);
    return 0;
}

extern "C" int vlmo_colsum(int dtype, const void* x, int ld, float* out, int M, int N, float* ws, int64_t ws_bytes,
                           hipStream_t stream) {
    VLMO_CHECK_ARG(x && out, "vlmo_colsum: null pointer");
    VLMO_CHECK_ARG(M > 0 && N > 0 && N % 8 == 0 && ld % 8 == 0 && ld >= N, "vlmo_colsum: bad shape M=%d N=%d ld=%d", M, N, ld);
    VLMO_CHECK_ARG(ws && ws_bytes >= reduce_ws_need(N), "vlmo_colsum: workspace too small (need %lld bytes)",
                   (long long)reduce_ws_need(N));
    const int gx = (N / 8 + 63) / 64;
    const int rpb = rows_per_block_for(M, VLMO_MAX_PARTIAL_BLOCKS, 16);
    dim3 grid(gx, (M + rpb - 1) / rpb), block(64, 4);
    if (dtype == VLMO_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16>, grid, block, 0, stream, (const bf16*)x, ld, ws, M, N, rpb);
    else if (dtype == VLMO_F16)
        hipLaunchKernelGGL(colsum_kernel<f16>, grid, block, 0, stream, (const f16*)x, ld, ws, M, N, rpb);
    else {
        vlmo_set_error("vlmo_colsum: dtype must be bf16 or f16");
        return -1;
    }
    VLMO_CHECK_LAUNCH("vlmo_colsum");
    return reduce_partials(ws, grid.y, N, out, N, nullptr, stream);
}

extern "C" int vlmo_colwork_multi(int dtype, const VlmoColJob* jobs, int n, hipStream_t stream) {
    VLMO_CHECK_ARG(jobs && n >= 1, "vlmo_colwork_multi: no jobs");
    VLMO_CHECK_ARG(dtype == VLMO_BF16 || dtype == VLMO_F16, "vlmo_colwork_multi: dtype must be bf16 or f16");
    for (int q0 = 0; q0 < n; q0 += MAX_COL_JOBS) {
        const int nq = n - q0 < MAX_COL_JOBS ? n - q0 : MAX_COL_JOBS;
        ColMulti cm{};
        cm.n = nq;
        int b = 0;
        for (int q = 0; q < nq; ++q) {
            const VlmoColJob& J = jobs[q0 + q];
            VLMO_CHECK_ARG(J.src && J.rows > 0 && J.ncols > 0 && J.n0 > 0 && J.ncols <= 4 * J.n0 && J.ld >= J.ncols,
                           "vlmo_colwork_multi: bad job %d", q0 + q);
            VLMO_CHECK_ARG(J.kind == 0 || (J.kind == 1 && J.ncols % 8 == 0 && J.ld % 8 == 0),
                           "vlmo_colwork_multi: job %d: matrix column sums need ncols, ld multiples of 8", q0 + q);
            cm.j[q] = J;
            cm.gx[q] = J.kind == 0 ? (J.ncols + 63) / 64 : (J.ncols / 8 + 63) / 64;
            int slices = J.kind == 0 ? (J.rows >= 64 ? 8 : 1) : (J.rows >= 2048 ? 32 : (J.rows >= 256 ? 8 : 1));
            int rps = (J.rows + slices - 1) / slices;
            if (J.kind == 1) rps = ((rps + 15) / 16) * 16;
            slices = (J.rows + rps - 1) / rps;
            cm.rps[q] = rps;
            cm.b0[q] = b;
            b += cm.gx[q] * slices;
        }
        for (int q = nq; q <= MAX_COL_JOBS; ++q) cm.b0[q] = b;
        if (dtype == VLMO_F16)
            hipLaunchKernelGGL(colwork_multi_kernel<f16>, dim3(b), dim3(64, 4), 0, stream, cm);
        else
            hipLaunchKernelGGL(colwork_multi_kernel<bf16>, dim3(b), dim3(64, 4), 0, stream, cm);
        VLMO_CHECK_LAUNCH("vlmo_colwork_multi");
    }
    return 0;
}

extern "C" int vlmo_cast_weight(int dtype, const float* src, int rows, int cols, void* dst, void* dstT,
                                hipStream_t stream) {
    VLMO_CHECK_ARG(src && (dst || dstT), "vlmo_cast_weight: null pointer");
    VLMO_CHECK_ARG(rows > 0 && cols > 0, "vlmo_cast_weight: empty matrix");
    dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(32, 8);
    if (dtype == VLMO_BF16)
        hipLaunchKernelGGL(cast_weight_kernel<bf16>, grid, block, 0, stream, src, rows, cols, (bf16*)dst, (bf16*)dstT);
    else if (dtype == VLMO_F16)
        hipLaunchKernelGGL(cast_weight_kernel<f16>, grid, block, 0, stream, src, rows, cols, (f16*)dst, (f16*)dstT);
    else {
        vlmo_set_error("vlmo_cast_weight: dtype must be bf16 or f16");
        return -1;
    }
    VLMO_CHECK_LAUNCH("vlmo_cast_weight");
    return 0;
}

extern "C" int vlmo_cast_weight_multi(int dtype, int n, const float* const* src, const int32_t* rows, const int32_t* cols,
                                      void* const* dst, void* const* dstT, hipStream_t stream) {
    VLMO_CHECK_ARG(n >= 0 && (n == 0 || (src && rows && cols && dst && dstT)), "vlmo_cast_weight_multi: null table");
    VLMO_CHECK_ARG(dtype == VLMO_BF16 || dtype == VLMO_F16, "vlmo_cast_weight_multi: dtype must be bf16 or f16");
    for (int j0 = 0; j0 < n; j0 += CastJobs::MAXJ) {
        CastJobs J{};
        J.n = n - j0 < CastJobs::MAXJ ? n - j0 : CastJobs::MAXJ;
        int tiles = 0;
        for (int j = 0; j < J.n; ++j) {
            const int q = j0 + j;
            VLMO_CHECK_ARG(src[q] && (dst[q] || dstT[q]) && rows[q] > 0 && cols[q] > 0, "vlmo_cast_weight_multi: bad job %d", q);
            J.tile0[j] = tiles;
            J.src[j] = src[q], J.dst[j] = dst[q], J.dstT[j] = dstT[q], J.rows[j] = rows[q], J.cols[j] = cols[q];
            tiles += ((rows[q] + 63) / 64) * ((cols[q] + 63) / 64);
        }
        J.tile0[J.n] = tiles;
        if (dtype == VLMO_BF16)
            hipLaunchKernelGGL(cast_weight_multi_kernel<bf16>, dim3(tiles), dim3(256), 0, stream, J);
        else
            hipLaunchKernelGGL(cast_weight_multi_kernel<f16>, dim3(tiles), dim3(256), 0, stream, J);
        VLMO_CHECK_LAUNCH("vlmo_cast_weight_multi");
    }
    return 0;
}

extern "C" int vlmo_patchify(const float* img, void* out, int B, int C, int H, int W, int patch, hipStream_t stream) {
    VLMO_CHECK_ARG(img && out, "vlmo_patchify: null pointer");
    VLMO_CHECK_ARG(B > 0 && C > 0 && patch >= 4 && patch % 4 == 0 && H % patch == 0 && W % patch == 0,
                   "vlmo_patchify: bad shape B=%d C=%d H=%d W=%d patch=%d", B, C, H, W, patch);
    const long total4 = (long)B * C * H * W / 4;
    const int grid = (int)((total4 + 255) / 256 < 16384 ? (total4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(patchify_kernel, dim3(grid), dim3(256), 0, stream, img, (bf16*)out, B, C, H, W, patch, total4);
    VLMO_CHECK_LAUNCH("vlmo_patchify");
    return 0;
}

extern "C" int vlmo_embed_img_finish(const void* proj, const float* cls_tok, const float* mask_tok, const float* pos,
                                     const float* type_row, const uint8_t* masked_pos, float* x, int B, int npatch,
                                     int d, uint32_t drop_thresh, float inv_keep, uint64_t seed, hipStream_t stream) {
    VLMO_CHECK_ARG(proj && cls_tok && mask_tok && pos && type_row && x, "vlmo_embed_img_finish: null pointer");
    VLMO_CHECK_ARG(B > 0 && npatch > 0 && d % 4 == 0 && d <= 4096, "vlmo_embed_img_finish: bad shape");
    hipLaunchKernelGGL(embed_img_finish_kernel, dim3(npatch + 1, B), dim3(d / 4), 0, stream, (const bf16*)proj, cls_tok,
                       mask_tok, pos, type_row, masked_pos, x, npatch, d, drop_thresh, inv_keep, seed);
    VLMO_CHECK_LAUNCH("vlmo_embed_img_finish");
    return 0;
}

extern "C" int vlmo_embed_img_bwd(const float* dx, const uint8_t* masked_pos, void* dproj, float* dcls, float* dmask,
                                  float* dpos, float* dtype_row, int B, int npatch, int d, uint32_t drop_thresh,
                                  float inv_keep, uint64_t seed, hipStream_t stream) {
    VLMO_CHECK_ARG(dx && dproj, "vlmo_embed_img_bwd: null pointer");
    VLMO_CHECK_ARG(B > 0 && npatch > 0 && d % 4 == 0 && d <= 4096, "vlmo_embed_img_bwd: bad shape");
    hipLaunchKernelGGL(embed_img_bwd_kernel, dim3(npatch + 1), dim3(d / 4), 0, stream, dx, masked_pos, (bf16*)dproj,
                       dcls, dmask, dpos, dtype_row, B, npatch, d, drop_thresh, inv_keep, seed);
    VLMO_CHECK_LAUNCH("vlmo_embed_img_bwd");
    return 0;
}

extern "C" int vlmo_embed_txt_fwd(const int64_t* ids, const float* word, const float* pos, const float* btype0,
                                  const float* ln_w, const float* ln_b, const float* type0, float* x, float* xhat,
                                  float* rstd, int B, int T, int d, float eps, uint32_t drop_thresh, float inv_keep,
                                  uint64_t seed, hipStream_t stream) {
    VLMO_CHECK_ARG(ids && word && pos && btype0 && ln_w && ln_b && type0 && x, "vlmo_embed_txt_fwd: null pointer");
    VLMO_CHECK_ARG(B > 0 && T > 0 && d % 4 == 0 && d <= 1024, "vlmo_embed_txt_fwd: bad shape");
    const int ntok = B * T, vpl = (d / 4 + 63) / 64;
    const int grid = (ntok + 3) / 4 < 8192 ? (ntok + 3) / 4 : 8192;
#define ETF(V)                                                                                                         \
    hipLaunchKernelGGL(embed_txt_fwd_kernel<V>, dim3(grid), dim3(256), 0, stream, ids, word, pos, btype0, ln_w, ln_b,  \
                       type0, x, xhat, rstd, ntok, T, d, eps, drop_thresh, inv_keep, seed);
    switch (vpl) {
        case 1: ETF(1) break;
        case 2: ETF(2) break;
        case 3: ETF(3) break;
        default: ETF(4) break;
    }
#undef ETF
    VLMO_CHECK_LAUNCH("vlmo_embed_txt_fwd");
    return 0;
}

extern "C" int vlmo_embed_txt_bwd(const float* dx, const int64_t* ids, const float* xhat, const float* rstd,
                                  const float* ln_w, float* dword, float* dpos, float* dbtype0, float* dln_w,
                                  float* dln_b, float* dtype0, int B, int T, int d, uint32_t drop_thresh,
                                  float inv_keep, uint64_t seed, hipStream_t stream) {
    VLMO_CHECK_ARG(dx && ids && xhat && rstd && ln_w, "vlmo_embed_txt_bwd: null pointer");
    VLMO_CHECK_ARG(B > 0 && T > 0 && d % 4 == 0 && d <= 1024, "vlmo_embed_txt_bwd: bad shape");
    const int ntok = B * T, vpl = (d / 4 + 63) / 64;
    const int rpb = B >= 64 ? 16 : (B >= 16 ? 8 : 4);         // sequences per workgroup (one position each)
    const int grid = T * ((B + rpb - 1) / rpb);
#define ETB(V)                                                                                                          \
    hipLaunchKernelGGL(embed_txt_bwd_kernel<V>, dim3(grid), dim3(256), 0, stream, dx, ids, xhat, rstd, ln_w, dword,    \
                       dpos, dbtype0, dln_w, dln_b, dtype0, ntok, T, d, rpb, drop_thresh, inv_keep, seed);
    switch (vpl) {
        case 1: ETB(1) break;
        case 2: ETB(2) break;
        case 3: ETB(3) break;
        default: ETB(4) break;
    }
#undef ETB
    VLMO_CHECK_LAUNCH("vlmo_embed_txt_bwd");
    return 0;
}
