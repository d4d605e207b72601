// MFMA GEMMs for the VLMo hot path on gfx950.
//
//  gemm_nt : C[M,N] = A[M,K] . B[N,K]^T   (both operands K-contiguous)
//            forward linears (x . W^T) and, with pre-transposed weights, dgrad.
//            Reference call sites: vlmo.py:70-80 (qkv), :96 (proj), timm Mlp
//            fc1/fc2 (vlmo.py:141-157), patch-embed conv as GEMM (vlmo.py:304).
//  gemm_tn : C[N1,N2] += A[M,N1]^T . B[M,N2] (reduction over rows, split over
//            the grid's z dimension, fp32 atomics) = wgrad.
//
// Structure (per workgroup): BMxBNx64 tile, operands staged global->LDS by
// LDS-DMA (global_load_lds, 16 B/lane) into a 2-deep ring, XOR-swizzled on the
// SOURCE address so ds_read_b128 fragment reads are bank-conflict free, one
// barrier per K-tile, v_mfma_f32_32x32x16 accumulating in fp32, epilogue
// transposed through wave-private LDS so every global access is a full
// 128/256-byte row segment with the bias/GELU/dropout/layer-scale/residual
// math fused in.
#include "common.h"
#include "vlmo_hip.h"
#include <stdlib.h>
#include <mutex>
#include <string>
#include <vector>
#include <utility>
#include <type_traits>

namespace {

enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_RESID = 2, EPI_DGELU = 3, EPI_F32 = 4, EPI_DUAL = 5, EPI_ARGMAX = 6, EPI_CE = 7,
       EPI_CE_BWD = 8 };

struct GemmNT {
    const void* A;
    const void* B;
    int M, N, K, lda, ldb;
    VlmoEpilogue e;
    // implicit-GEMM convolution over an NHWC activation matrix [B*H*W, Cin] (dVAE encoder):
    // K = kw*kw*Cin, k-tile -> (tap, 64-channel chunk); taps outside the image read `zero`
    int cH, cW, cCin, ckw;
    const void* zero;
    int group_m;     // L2 tile swizzle: row-tiles per group
    // two-segment A (vlmo_gemm_nt_2src): columns [0, k1) of the reduction come from A, [k1, K) from A2 (own leading
    // dimension); the partial sum of the first segment is multiplied by seg_scale before the second one is added.
    // k1 == 0: single source.
    const void* A2;
    int lda2, k1;
    float seg_scale;
};

// Up to 4 problems with the same N, K, leading dimensions and epilogue kind in ONE launch (the per-modality
// expert FFNs below the fusion layer: different row ranges, weights, biases): group g owns the logical tiles
// [t0[g], t0[g+1]).  A launch never takes less than one tile time, so two half-empty launches cost twice one.
constexpr int MAX_GROUPS = 4;
struct GemmNTGroups {
    int ngroups;
    int t0[MAX_GROUPS + 1];
    GemmNT g[MAX_GROUPS];
};

// element address split into a wave-uniform 64-bit part and a per-lane 32-bit part
struct RowAddr {
    size_t base;
    uint32_t off;
    template <typename U> __device__ __forceinline__ U* at(const void* p) const {
        return (U*)((char*)((U*)p + base) + off * (uint32_t)sizeof(U));
    }
};
// NT = true: streaming output, a non-temporal store does not push the operand panels the neighbouring tiles re-read
// out of the XCD's L2 (measured per output kind -- bias, u, h, GELU-derivative: non-temporal is equal or better for
// each, -0.3 ms per step together; the same hint on the fp32 residual output costs +0.23 ms and on the outputs of
// the LayerNorm / attention kernels +1.0 ms: their consumers find them in the Infinity Cache)
template <typename T, bool NT = true> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d) {
    typedef typename Elem<T>::v4 v4;
    const v4 v = {(T)a, (T)b, (T)c, (T)d};
    if constexpr (NT)
        __builtin_nontemporal_store(v, (v4*)p);
    else
        *(v4*)p = v;
}
// epilogue inputs read exactly once (fp32 residual, pre-activation): non-temporal, -0.08 ms per step.  The fp32 residual
// OUTPUT stays a normal store: the next kernel (LayerNorm) reads it back at once, non-temporal was +0.23 ms.
#define NT_LD(p) __builtin_nontemporal_load(p)
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
    bf16x4 v = NT_LD((const bf16x4*)p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <> __device__ __forceinline__ f32x4 load4<f16>(const f16* p) {
    f16x4 v = NT_LD((const f16x4*)p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// Epilogue math for 4 consecutive output columns (gn..gn+3) of row gm.  Everything that has to come
// from global memory is passed in (bias/gamma: loaded once per tile; `ext` = residual / pre-activation
// row segment and `rs` = drop-path scale: loaded for a whole pass BEFORE any math so the ~1-2 us
// global latencies overlap instead of serialising load -> math -> store per row group).
// keeps the four values live in registers HERE: hipcc otherwise sinks the arithmetic that produced them into the
// guarded block of their only user (the store) -- see epilogue4
__device__ __forceinline__ void pin4(f32x4& v) {
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
}

// GD: VlmoEpilogue.relu bit 2 (saved GELU derivative, see EPI_BIAS_GELU below) known at compile time (0 / 1) or read from the
// descriptor (-1).  The 16x16x32 kernels instantiate both: with the choice at run time each unrolled epilogue pass carried
// both bodies and the fc1 kernel grew to 110 KB of instructions.
template <typename T, int EPI, int GD = -1>
__device__ __forceinline__ f32x4 epilogue4(const GemmNT& p, int gmb, int row, int gn, f32x4 v, f32x4 bias4,
                                           f32x4 gamma4, f32x4 ext, float rs, bool ok) {
    // ALL arithmetic runs unconditionally (rows past M compute on clamped inputs) and only the stores sit under
    // `ok`: with the math inside the guard every guarded block was the first user of a pending load (bias, residual
    // row) on SOME path, so hipcc put `s_waitcnt vmcnt(0)` in front of each of them -- which also waits for the
    // previous block's STORE: 32 serialised HBM round trips per wave, ~10 us of a 256x256 tile's epilogue.
    const VlmoEpilogue& e = p.e;
    v += bias4;
    // wave-uniform 64-bit row base (scalar unit) + 32-bit in-tile offset: a per-lane 64-bit
    // multiply-add per store costs 4x a plain VALU op
    // (global_* saddr form: SGPR base + zero-extended 32-bit VGPR byte offset)
    const int gm = gmb + row;
    const RowAddr o{(size_t)gmb * e.ldo, (uint32_t)(row * e.ldo + gn)};
    const RowAddr o2{(size_t)gmb * e.ld2, (uint32_t)(row * e.ld2 + gn)};
    if constexpr (EPI == EPI_BIAS) {
        const float lo = (e.relu & 1) ? 0.f : -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], lo);
        pin4(v);
        if (ok) store4<T>(o.at<T>(e.out), v[0], v[1], v[2], v[3]);
    } else if constexpr (EPI == EPI_F32) {
        if (e.beta != 0.f) v += e.beta * ext;
        pin4(v);
        if (ok) *(f32x4*)o.at<float>(e.out) = v;
    } else if constexpr (EPI == EPI_BIAS_GELU) {
        f32x4 h;
        if (GD >= 0 ? GD != 0 : (e.relu & 4) != 0) {
            // `out` receives d h / d u = GELU'(u) * dropout mask / (1 - p) instead of the pre-activation u: the backward's
            // GELU-derivative epilogue (EPI_DGELU with the same bit) is then ONE multiply per element -- no erf, no
            // exponential, no dropout hash (27.8 -> ~6 vector instructions per element there for ~5 more here: the
            // Gaussian and the normal CDF are shared with GELU itself)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float u = v[j], ee = gauss_from(u), cdf = norm_cdf_from(u, ee);
                h[j] = u * cdf;
                v[j] = fmaf(u * 0.39894228040143268f, ee, cdf);
            }
            if (e.drop_thresh) {
                const uint64_t bits = drop_bits4(e.seed, ((uint64_t)gm * p.N + gn) >> 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float m = drop_keep(bits, j, e.drop_thresh) ? e.inv_keep : 0.f;
                    h[j] *= m;
                    v[j] *= m;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = gelu_erf(v[j]);
            if (e.drop_thresh) {
                const uint64_t bits = drop_bits4(e.seed, ((uint64_t)gm * p.N + gn) >> 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = drop_keep(bits, j, e.drop_thresh) ? h[j] * e.inv_keep : 0.f;
            }
        }
        pin4(v);
        pin4(h);
        if (ok) {
            store4<T>(o.at<T>(e.out), v[0], v[1], v[2], v[3]);   // u (pre-activation)
            store4<T>(o2.at<T>(e.out2), h[0], h[1], h[2], h[3]);
        }
    } else if constexpr (EPI == EPI_RESID) {
        if (e.drop_thresh) {
            const uint64_t bits = drop_bits4(e.seed, ((uint64_t)gm * p.N + gn) >> 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = drop_keep(bits, j, e.drop_thresh) ? v[j] * e.inv_keep : 0.f;
        }
        f32x4 x2 = ext + gamma4 * v * rs;
        pin4(v);
        pin4(x2);
        if (ok) {
            if (e.out2) store4<T>(o2.at<T>(e.out2), v[0], v[1], v[2], v[3]);
            *(f32x4*)o.at<float>(e.out) = x2;
        }
    } else if constexpr (EPI == EPI_DUAL) {
        // dVAE EncoderBlock tail (dall_e/encoder.py:45-46): out = id + post_gain * res ; out2 = relu(out)
        v = v * e.beta + ext;
        pin4(v);
        if (ok) {
            store4<T>(o.at<T>(e.out), v[0], v[1], v[2], v[3]);
            if (e.out2)
                store4<T>((T*)e.out2 + (size_t)gm * e.ld2 + gn, fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f),
                          fmaxf(v[3], 0.f));
        }
    } else if constexpr (EPI == EPI_CE_BWD) {
        // d(cross-entropy)/d(logits) of row gm, recomputed from the logits instead of read back: (softmax - onehot) *
        // row scale (heads.py:86-112 + objectives.py:57-68,571-582).  resid = lse [M], row_scale = dloss / n_valid per
        // row (0 on ignored rows), row_index = labels [M]
        const int gmc = min(gm, p.M - 1);
        const float lse = e.resid[gmc], sc = e.row_scale[gmc];
        const int lab = e.row_index[gmc];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = (__builtin_amdgcn_exp2f((v[j] - lse) * 1.4426950408889634f) - (gn + j == lab ? 1.f : 0.f)) * sc;
        pin4(v);
        if (ok) store4<T>(o.at<T>(e.out), v[0], v[1], v[2], v[3]);
    } else if constexpr (EPI == EPI_DGELU) {
        if (GD >= 0 ? GD != 0 : (e.relu & 4) != 0) {
            v *= ext;       // aux holds GELU'(u) * mask / (1 - p), written by the forward's EPI_BIAS_GELU with the same bit
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_grad(ext[j]);
            if (e.drop_thresh) {
                const uint64_t bits = drop_bits4(e.seed, ((uint64_t)gm * p.N + gn) >> 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = drop_keep(bits, j, e.drop_thresh) ? v[j] * e.inv_keep : 0.f;
            }
        }
        pin4(v);
        if (ok) store4<T>(o.at<T>(e.out), v[0], v[1], v[2], v[3]);
    }
    return v;
}

// the row segment an epilogue needs from global memory besides the accumulators (clamped row: always valid)
template <typename T, int EPI>
__device__ __forceinline__ f32x4 epilogue_ext(const GemmNT& p, int gmb, int rowc, int gnc) {
    const VlmoEpilogue& e = p.e;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const RowAddr o{(size_t)gmb * e.ldo, (uint32_t)(rowc * e.ldo + gnc)};
    const RowAddr o2{(size_t)gmb * e.ld2, (uint32_t)(rowc * e.ld2 + gnc)};
    if constexpr (EPI == EPI_RESID) {
        return NT_LD((const f32x4*)o.at<float>(e.resid));
    } else if constexpr (EPI == EPI_DGELU) {
        return load4<T>(o2.at<T>(e.aux));
    } else if constexpr (EPI == EPI_DUAL) {
        return e.resid ? load4<T>(o.at<T>(e.resid)) : z;
    } else if constexpr (EPI == EPI_F32) {
        return e.beta != 0.f ? *(const f32x4*)o.at<float>(e.out) : z;
    } else {
        return z;
    }
}

// The same row segment as it comes from memory (no conversion: a conversion would be the load's first user and pin an
// s_waitcnt right behind it), for epilogues that request the NEXT pass's segments before working on the current one.
template <typename T, int EPI> struct ExtRaw { typedef f32x4 type; };
template <typename T> struct ExtRaw<T, EPI_DGELU> { typedef typename Elem<T>::v4 type; };
template <typename T, int EPI>
__device__ __forceinline__ typename ExtRaw<T, EPI>::type epilogue_ext_raw(const GemmNT& p, int gmb, int rowc, int gnc) {
    if constexpr (EPI == EPI_DGELU) {
        const RowAddr o2{(size_t)gmb * p.e.ld2, (uint32_t)(rowc * p.e.ld2 + gnc)};
        return NT_LD((const typename Elem<T>::v4*)o2.at<T>(p.e.aux));
    } else {
        return epilogue_ext<T, EPI>(p, gmb, rowc, gnc);
    }
}
template <typename T> __device__ __forceinline__ f32x4 ext_f32(f32x4 v) { return v; }
template <typename T> __device__ __forceinline__ f32x4 ext_f32(typename Elem<T>::v4 v) {
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}

// The rebuilt pointer is typed GLOBAL before it decays to a generic one: an integer -> generic pointer would make every
// access through it a FLAT instruction (which counts on vmcnt AND lgkmcnt, so each epilogue store was followed by
// `s_waitcnt vmcnt(0) lgkmcnt(0)` before the next LDS access: ~10 us of serialised store round trips per 256x256 tile).
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const void*)(const __attribute__((address_space(1))) void*)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniform_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// bijective XCD-chunked remap: workgroups that share an XCD (bid % 8 equal)
// get a contiguous range of logical tile ids, so the A row-panel re-reads of
// neighbouring column tiles hit that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// K-tile depth BK (64 or 32): BK=32 halves the LDS ring (32 KB for 128x128) so four workgroups
// fit a CU instead of two (more latency hiding, phases of co-resident workgroups decorrelate).
template <int BK> __device__ __forceinline__ int nt_swz(int row) {
    return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

template <typename T, int BM, int BN, int WM, int WN, int EPI, bool CONV, int BK, int NSTG, bool PP = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_nt_kernel(const GemmNTGroups gp) {
    typedef typename Elem<T>::v8 v8;
    // two-segment reduction (vlmo_gemm_nt_2src): instantiated for the f16 (dVAE) kernels only -- the extra branch in the
    // staging step cost the bf16 256x128x32 kernels of the transformer (fc1, qkv) 8-10 %
    constexpr bool SEG2 = __is_same(T, f16) && !CONV;
    constexpr int NW = WM * WN;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int ROWB = BK * 2, CPR = ROWB / 16, SRPI = 1024 / ROWB, KS = BK / 16;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int NA = BM / SRPI / NW, NB = BN / SRPI / NW;
    static_assert(BM % (SRPI * NW) == 0 && BN % (SRPI * NW) == 0, "tile/wave mismatch");
    static_assert(NW * 32 * TN * 32 * 4 <= NSTG * STAGE, "epilogue LDS must fit in the ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int lid_all = xcd_remap(blockIdx.x, gridDim.x);
    // (group, row origin, column origin) of a logical tile
    auto locate = [&](int la, int& gi_, int& m0_, int& n0_) {
        int g = 0;
#pragma unroll
        for (int q = 1; q < MAX_GROUPS; ++q)
            if (q < gp.ngroups && la >= gp.t0[q]) g = q;
        g = __builtin_amdgcn_readfirstlane(g);
        const GemmNT& r = gp.g[g];
        const int M_ = uniform_i(r.M), N_ = uniform_i(r.N);
        const int tiles_n = (N_ + BN - 1) / BN, tiles_m = (M_ + BM - 1) / BM;
        const int lid = la - uniform_i(gp.t0[g]);
        // grouped order: the ~64 tiles an XCD works on at once form a compact group_m x (64/group_m)
        // block, so their A row-panels AND B column-panels together fit the XCD's 4 MiB L2
        const int gmr = uniform_i(r.group_m);
        const int gm_ = gmr > 0 ? gmr : 1;
        const int per_group = gm_ * tiles_n;
        const int first_m = (lid / per_group) * gm_;
        const int gsz = min(tiles_m - first_m, gm_);
        const int in_g = lid % per_group;
        gi_ = g;
        m0_ = (first_m + in_g % gsz) * BM;
        n0_ = (in_g / gsz) * BN;
    };
    const T* a_src[NA];
    const T* b_src[NB];
    int a_yx[NA];            // CONV: (y << 16) | x of the staged output pixel
    // per-lane source addresses of the tile's operand rows (LDS-DMA: one 16-byte chunk per lane)
    auto point = [&](int gi_, int m0_, int n0_) {
        const GemmNT& r = gp.g[gi_];
        const T* A_ = (const T*)uniform_ptr(r.A);
        const T* B_ = (const T*)uniform_ptr(r.B);
        const int M_ = uniform_i(r.M), N_ = uniform_i(r.N), lda_ = uniform_i(r.lda), ldb_ = uniform_i(r.ldb);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int rr = (i * NW + wave) * SRPI + lane / CPR;
            const int c = (lane % CPR) ^ nt_swz<BK>(rr);
            const int gr = min(m0_ + rr, M_ - 1);
            a_src[i] = A_ + (size_t)gr * lda_ + c * 8;
            if constexpr (CONV) {
                const int cH = uniform_i(r.cH), cW = uniform_i(r.cW);
                const int pix = gr % (cH * cW);
                a_yx[i] = ((pix / cW) << 16) | (pix % cW);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int rr = (i * NW + wave) * SRPI + lane / CPR;
            const int c = (lane % CPR) ^ nt_swz<BK>(rr);
            const int gr = min(n0_ + rr, N_ - 1);
            b_src[i] = B_ + (size_t)gr * ldb_ + c * 8;
        }
    };
    int gi, m0, n0;
    locate(lid_all, gi, m0, n0);
    point(gi, m0, n0);
    GemmNT pl;
    const GemmNT* pp = &gp.g[gi];
    // the chosen problem, copied into SGPRs ONCE per group (see uniform_i): a dynamically indexed kernarg struct is
    // otherwise re-read with s_load + s_waitcnt at every use (115 scalar loads in the fc1 epilogue before this)
    auto hoist = [&](int gi_) {
        const GemmNT& gq = gp.g[gi_];
        {
            pl.A = uniform_ptr(gq.A), pl.B = uniform_ptr(gq.B), pl.zero = uniform_ptr(gq.zero);
            pl.M = uniform_i(gq.M), pl.N = uniform_i(gq.N), pl.K = uniform_i(gq.K), pl.lda = uniform_i(gq.lda), pl.ldb = uniform_i(gq.ldb);
            pl.cH = uniform_i(gq.cH), pl.cW = uniform_i(gq.cW), pl.cCin = uniform_i(gq.cCin), pl.ckw = uniform_i(gq.ckw);
            pl.group_m = uniform_i(gq.group_m);
            pl.A2 = uniform_ptr(gq.A2), pl.lda2 = uniform_i(gq.lda2), pl.k1 = uniform_i(gq.k1), pl.seg_scale = uniform_f(gq.seg_scale);
            pl.e.out = (void*)uniform_ptr(gq.e.out), pl.e.out2 = (void*)uniform_ptr(gq.e.out2);
            pl.e.bias = (const float*)uniform_ptr(gq.e.bias), pl.e.gamma = (const float*)uniform_ptr(gq.e.gamma);
            pl.e.resid = (const float*)uniform_ptr(gq.e.resid), pl.e.row_scale = (const float*)uniform_ptr(gq.e.row_scale);
            pl.e.row_index = (const int32_t*)uniform_ptr(gq.e.row_index), pl.e.aux = uniform_ptr(gq.e.aux);
            pl.e.ldo = uniform_i(gq.e.ldo), pl.e.ld2 = uniform_i(gq.e.ld2), pl.e.relu = uniform_i(gq.e.relu);
            pl.e.drop_thresh = (uint32_t)uniform_i((int)gq.e.drop_thresh);
            pl.e.inv_keep = uniform_f(gq.e.inv_keep), pl.e.beta = uniform_f(gq.e.beta);
            pl.e.seed = (uint64_t)uniform_ptr((const void*)gq.e.seed);
            pl.e.colpart = (float*)uniform_ptr(gq.e.colpart);
            pp = &pl;
        }
    };
    hoist(gi);

    f32x16 acc[TM][TN];
    const int l31 = lane & 31, h = lane >> 5;
    const int swz = nt_swz<BK>(l31);
    const int a_row_off = (wm * (BM / WM) + l31) * ROWB;
    const int b_row_off = A_BYTES + (wn * (BN / WN) + l31) * ROWB;

    const int nk = pp->K / BK;       // the groups of a launch share N and K
    const int cpt = CONV ? (pp->cCin / BK) : 1, cpad = CONV ? (pp->ckw - 1) / 2 : 0;
    auto stage = [&](int buf, int kt) {
        char* s = smem + buf * STAGE;
        if constexpr (CONV) {
            const GemmNT& p = *pp;
            const int tap = kt / cpt, cc = kt - tap * cpt;
            const int dy = tap / p.ckw - cpad, dx = tap % p.ckw - cpad;
            const int delta = (dy * p.cW + dx) * p.cCin + cc * BK;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int y = (a_yx[i] >> 16) + dy, x = (a_yx[i] & 0xFFFF) + dx;
                const bool in = (unsigned)y < (unsigned)p.cH && (unsigned)x < (unsigned)p.cW;
                const T* src = in ? a_src[i] + delta : (const T*)p.zero;
                glds16(src, s + (i * NW + wave) * 1024);
            }
        } else {
            if (SEG2 && pp->k1 && kt * BK == pp->k1) {
                // second A segment: the same rows of A2, rebased so that `+ kt * BK` keeps addressing the reduction index
                const T* A2_ = (const T*)pp->A2;
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int rr = (i * NW + wave) * SRPI + lane / CPR;
                    const int c = (lane % CPR) ^ nt_swz<BK>(rr);
                    const int gr = min(m0 + rr, pp->M - 1);
                    a_src[i] = A2_ + (size_t)gr * pp->lda2 + c * 8 - pp->k1;
                }
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) glds16(a_src[i] + kt * BK, s + (i * NW + wave) * 1024);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) glds16(b_src[i] + kt * BK, s + A_BYTES + (i * NW + wave) * 1024);
    };
    auto compute = [&](const char* s) {
        v8 af[KS][TM], bf[KS][TN];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int coff = ((2 * ks + h) ^ swz) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[ks][i] = *(const v8*)(s + a_row_off + i * 32 * ROWB + coff);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[ks][j] = *(const v8*)(s + b_row_off + j * 32 * ROWB + coff);
        }
        if constexpr (CONV && sizeof(T) == 2 && __is_same(T, f16)) {
            // ReLU on the INPUT (VlmoEpilogue.relu bit 1): the dVAE's residual path convolves relu(x) (encoder.py:21-29)
            // while the identity path and the max-pool take x itself, so the producer would have to write both; four
            // v_pk_max_f16 per fragment beside 4-8 MFMAs are cheaper than a second [B*H*W, C] tensor through HBM
            if (pp->e.relu & 2) {
                const v8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[ks][i] = __builtin_elementwise_max(af[ks][i], z);
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Elem<T>::mfma(af[ks][i], bf[ks][j], acc[i][j]);
    };
    const GemmNT& p = *pp;
    // first K-tile of the second A segment: the first segment's partial sum takes its scale (EncoderBlock tail:
    // post_gain * res_path + id_path as ONE reduction over [conv_3 output | block input])
    auto seg_boundary = [&](int kt) {
        if constexpr (SEG2) {
            if (pp->k1 && kt * BK == pp->k1 && pp->seg_scale != 1.f) {
                const float sc = pp->seg_scale;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int k = 0; k < 16; ++k) acc[i][j][k] *= sc;
            }
        }
    };
    // EncoderBlock tail (EPI_DUAL, K = n_hid = 64 .. 512): the kernel is a read of the identity map and a write of the
    // output around a one-tile product; with the identity rows fetched in the epilogue passes (8 bytes per lane, 4 KB
    // per wave in flight) a CU pulled 11 GB/s.  All of a wave's identity segments are requested BEFORE the K loop
    // instead: they fly under the operand staging.
    constexpr bool PRE = (EPI == EPI_DUAL) && !PP && !CONV;
    constexpr int PRE_LPR = TN * 8, PRE_RPI = 64 / PRE_LPR, PRE_NIT = 32 / PRE_RPI;
    typename Elem<T>::v4 pre[PRE ? TM : 1][PRE ? PRE_NIT : 1];
    if constexpr (PRE) {
        if (p.e.resid) {
            const int gn_ = n0 + wn * (BN / WN) + (lane % PRE_LPR) * 4;
            const int gnc_ = gn_ < p.N ? gn_ : 0;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int gmbc_ = min(m0 + wm * (BM / WM) + i * 32, p.M - 1);
#pragma unroll
                for (int it = 0; it < PRE_NIT; ++it) {
                    const int rowc_ = min(it * PRE_RPI + lane / PRE_LPR, p.M - 1 - gmbc_);
                    pre[i][it] = NT_LD((const typename Elem<T>::v4*)((const T*)p.e.resid + (size_t)(gmbc_ + rowc_) * p.e.ldo + gnc_));
                }
            }
        }
    }
    stage(0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    if constexpr (PP) {
        // Ping-pong schedule (8 waves, WM == 2): every K-tile is four segments separated by raw
        // s_barriers -- read fragments of k-half 0 | 16 MFMAs | read k-half 1 | 16 MFMAs -- and the
        // wm == 1 waves run ONE segment behind the wm == 0 waves (one extra barrier up front, one
        // at the end for wm == 0).  A SIMD hosts one wave of each group, so while one multiplies the
        // other reads LDS: the MFMA pipe and the LDS port are both busy all the time instead of
        // taking turns.  The LDS-DMA of tile t+1 is issued at the start of the first MFMA segment of
        // tile t (all reads of that buffer retired >= 1 barrier earlier) and waited for, by the
        // issuing wave, in its segment before the barrier that opens tile t+1 for the leading group.
        static_assert(NSTG == 2 && WM == 2 && KS == 4, "ping-pong schedule: 2 buffers, 2 row groups, BK = 64");
        v8 af[2][TM], bf[2][TN];
        auto read_half = [&](const char* s_, int hf) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int coff = ((2 * (2 * hf + q) + h) ^ swz) << 4;
#pragma unroll
                for (int i = 0; i < TM; ++i) af[q][i] = *(const v8*)(s_ + a_row_off + i * 32 * ROWB + coff);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[q][j] = *(const v8*)(s_ + b_row_off + j * 32 * ROWB + coff);
            }
            if constexpr (CONV && sizeof(T) == 2 && __is_same(T, f16)) {
                if (pp->e.relu & 2) {       // ReLU on the convolution's input (see compute())
                    const v8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int i = 0; i < TM; ++i) af[q][i] = __builtin_elementwise_max(af[q][i], z);
                }
            }
        };
        auto mfma_half = [&]() {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = Elem<T>::mfma(af[q][i], bf[q][j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        };
        auto bar = [&]() {      // raw barrier: no vmcnt drain; nothing may be scheduled across it
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wm == 1) bar();
        for (int kt = 0; kt < nk; ++kt) {
            const char* cur = smem + (kt & 1) * STAGE;
            const bool more = kt + 1 < nk;
            seg_boundary(kt);
            read_half(cur, 0);
            // the LDS-DMA of the next K-tile goes out in the READ segment, behind the fragment reads (round 4: in-kernel
            // stamps showed the eight DMA instructions' issue time -- ~400 cycles -- in front of the wave's own MFMAs when
            // they were issued at the head of the MFMA segment; here it runs beside the partner wave's MFMA segment.  The
            // other buffer was last read two segments ago by this group and one segment ago by the other, each behind
            // lgkmcnt(0) + barrier.)  3 273 -> 2 776 cycles per K-tile in the stamped build, +4..9 % per GEMM.
            if (more) stage((kt + 1) & 1, kt + 1);
            bar();
            mfma_half();
            bar();
            read_half(cur, 1);
            if (more && wm == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bar();
            mfma_half();
            if (more && wm == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bar();
        }
        if (wm == 0) bar();
    } else {
        static_assert(NSTG == 2, "two LDS buffers");
        // 2-deep ring: one K-tile in flight behind the one being multiplied
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
            seg_boundary(kt);
            compute(smem + (kt & 1) * STAGE);
        }
    }

    // ---- epilogue: accumulators -> wave-private LDS -> full-row segments ----
    __syncthreads();
    constexpr int ROWF = TN * 32;                 // floats per LDS row
    constexpr int LPR = TN * 8, RPI = 64 / LPR;   // lanes per row, rows per read instr
    float* ep = (float*)(smem + wave * (32 * ROWF * 4));
    const int rrow = lane / LPR, rcol = (lane % LPR) * 4;
    const int gn = n0 + wn * (BN / WN) + rcol;
    const bool col_ok = gn < p.N;
    const int gnc = col_ok ? gn : 0;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, gamma4 = {1.f, 1.f, 1.f, 1.f};
    if (p.e.bias) bias4 = *(const f32x4*)(p.e.bias + gnc);
    if (EPI == EPI_RESID && p.e.gamma) gamma4 = *(const f32x4*)(p.e.gamma + gnc);
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};      // EPI_DGELU: column sums of a 32-row block (fc1 bias gradient partials)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        // issue this pass's global loads first: they fly while the accumulators go through LDS
        constexpr int NIT = 32 / RPI;
        const int gmb = __builtin_amdgcn_readfirstlane(m0 + wm * (BM / WM) + i * 32);
        const int gmbc = min(gmb, p.M - 1);      // edge tiles: a wave's rows may all lie past M
        f32x4 ext[NIT];
        float rs[NIT];
        int ridx[NIT];
        // the drop-path scale is a two-step lookup (token -> scale group -> scale): all index loads of the pass go out
        // first, then all dependent loads, instead of eight index -> wait -> scale round trips in a row
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int gmc = gmbc + min(it * RPI + rrow, p.M - 1 - gmbc);
            ridx[it] = (EPI == EPI_RESID && p.e.row_scale && p.e.row_index) ? p.e.row_index[gmc] : gmc;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int rowc = min(it * RPI + rrow, p.M - 1 - gmbc);
            if constexpr (PRE) {
                const typename Elem<T>::v4 q = pre[i][it];
                ext[it] = p.e.resid ? f32x4{(float)q[0], (float)q[1], (float)q[2], (float)q[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                ext[it] = epilogue_ext<T, EPI>(p, gmbc, rowc, gnc);
            }
            rs[it] = (EPI == EPI_RESID && p.e.row_scale) ? p.e.row_scale[ridx[it]] : 1.f;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ep[((r & 3) + 8 * (r >> 2) + 4 * h) * ROWF + j * 32 + l31] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        f32x4 v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) v[it] = *(const f32x4*)(ep + (it * RPI + rrow) * ROWF + rcol);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = it * RPI + rrow;
            const int gm = gmb + row;
            if constexpr (EPI == EPI_ARGMAX) {
                // fused arg-max over the vocabulary (modeling_discrete_vae.py:246-248): per row, the best
                // (value, index) of this wave's ROWF columns -> partial[gm][chunk]; logits never reach HBM
                f32x4 vv = v[it] + bias4;
                float best = -INFINITY;
                int bi = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (gn + j < p.N && vv[j] > best) {
                        best = vv[j];
                        bi = gn + j;
                    }
#pragma unroll
                for (int o = 1; o < LPR; o <<= 1) {
                    const float ob = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (ob > best || (ob == best && oi < bi)) {
                        best = ob;
                        bi = oi;
                    }
                }
                if ((lane % LPR) == 0 && gm < p.M) {
                    const int chunk = (n0 + wn * (BN / WN)) / ROWF;
                    float* pv = (float*)p.e.out + ((size_t)gm * p.e.ldo + chunk) * 2;
                    pv[0] = best;
                    ((int*)pv)[1] = bi;
                }
            } else if constexpr (EPI == EPI_CE) {
                // fused cross-entropy forward: per row and 64-column chunk {max, sum exp(x - max), arg-max, logit of
                // the row's label or -inf}; the [n, vocabulary] logits never reach HBM.  row_index = labels
                f32x4 vv = v[it] + bias4;
                const int lab = (gm < p.M) ? p.e.row_index[gm] : -1;
                float best = -INFINITY, labv = -INFINITY;
                int bi = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (gn + j < p.N) {
                        if (vv[j] > best) {
                            best = vv[j];
                            bi = gn + j;
                        }
                        if (gn + j == lab) labv = vv[j];
                    }
#pragma unroll
                for (int o = 1; o < LPR; o <<= 1) {
                    const float ob = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(bi, o, 64);
                    if (ob > best || (ob == best && oi < bi)) {
                        best = ob;
                        bi = oi;
                    }
                    labv = fmaxf(labv, __shfl_xor(labv, o, 64));
                }
                float se = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (gn + j < p.N) se += __builtin_amdgcn_exp2f((vv[j] - best) * 1.4426950408889634f);
#pragma unroll
                for (int o = 1; o < LPR; o <<= 1) se += __shfl_xor(se, o, 64);
                if ((lane % LPR) == 0 && gm < p.M) {
                    const int chunk = (n0 + wn * (BN / WN)) / ROWF;
                    float* pv = (float*)p.e.out + ((size_t)gm * p.e.ldo + chunk) * 4;
                    pv[0] = best;
                    pv[1] = se;
                    ((int*)pv)[2] = bi;
                    pv[3] = labv;
                }
            } else {
                const bool ok = gm < p.M && col_ok;
                const f32x4 w = epilogue4<T, EPI>(p, gmb, row, gn, v[it], bias4, gamma4, ext[it], rs[it], ok);
                if constexpr (EPI == EPI_DGELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) csum[j] += ok ? w[j] : 0.f;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if constexpr (EPI == EPI_DGELU) {
            // column sums of du per 32-row block (one epilogue pass of one wave) -> colpart[block][N]: the fc1 bias gradient
            // is their fold, done with the other column folds of the block instead of a second pass over [M, hidden]
            if (p.e.colpart) {
#pragma unroll
                for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
                    for (int j = 0; j < 4; ++j) csum[j] += __shfl_xor(csum[j], o, 64);
                // colpart has one row per 16 output rows (the 16x16x32 kernels place their passes at multiples of 16):
                // this 32-row pass owns two of them, the sums go to the first, zeros to the second
                const int blk = gmb >> 4;
                if (lane < LPR && col_ok && gmb < p.M) {
                    *(f32x4*)(p.e.colpart + (size_t)blk * p.N + gn) = csum;
                    if (gmb + 16 < p.M) *(f32x4*)(p.e.colpart + (size_t)(blk + 1) * p.N + gn) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                csum = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

// ------------------------------------------------ NT, 16x16x32 MFMA, tile height as a parameter ---
// Same LDS image, staging, ping-pong schedule and epilogue arithmetic as gemm_nt_kernel<..., PP>, with two differences:
//  * v_mfma_f32_16x16x32_bf16: at equal cycles per flop the chip holds a higher clock on this shape than on 32x32x16
//    (MI355X guide, DVFS give-back item 7: 1.12-1.15x the FLOP/s of the 32x32x16 loop on random data).  Fragment
//    reads are ds_read_b128 of 16 rows x 4 k-chunks out of the unchanged swizzled [rows][64] image
//    (tests/test_lds_layouts.py::test_gemm_nt16_fragments: right elements, conflict-free).
//  * the workgroup tile is (32 * TM) x 256: eight waves as 2 x 4, a wave owns TM x 4 accumulator tiles of 16 x 16, so the
//    tile HEIGHT moves in 32-row steps (TM = 6 .. 10: 192 .. 320 rows).  M = 16 704 = 65.25 x 256 puts every N = 3 072
//    GEMM of VLMo-Base at 64 pairs a few tiles into a fourth dispatch round of 256-row tiles; 288 rows = 58 x 12 tiles =
//    2.7 rounds, 320 rows for N = 2 304 = 53 x 9 = 1.9 rounds, 224 rows for N = 768 = 75 x 3 = 225 of 256 CUs.
template <int... Is, typename F> __device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}

// (A two-per-CU variant of this kernel -- four waves, (32 * TM) x 128 tile on 32-deep slices, the epilogue of one workgroup
// under the K loop of the other -- was built and measured in round 4: 256 rows = the 256x128x32 tile of gemm_nt_kernel
// (121 vs 122 us for fc1), 288 rows 125 us against 116 us for this kernel at 288 rows: the 256-row build fits THREE
// workgroups per CU (168 registers), the 288-row one two.  Removed.)
// H16 = tile height in 16-row units (12 .. 20): the wm == 0 waves own ceil(H16 / 2) accumulator tile rows, the wm == 1 waves
// floor(H16 / 2) -- the two wave groups take turns on the matrix pipe, so a K-tile costs TMA + TMB MFMA segments whatever
// the split, and the tile height moves in 16-row steps (208 rows: 243 tiles for N = 768 at M = 16 704; 272 rows: 744 tiles
// = three rounds for N = 3 072; 304 rows: 495 tiles = two rounds for N = 2 304).  Everything from the accumulators on is
// written once and instantiated per wave group (`body`); both copies execute the same barrier sequence.
template <typename T, int H16, int EPI, int SCHED = 1, int GD = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt16_kernel(const GemmNTGroups gp) {
    typedef typename Elem<T>::v8 v8;
    constexpr int TMA = (H16 + 1) / 2, TMB = H16 / 2;
    constexpr int BM = 16 * H16, WN = 4, BN = WN * 64, BK = 64, NW = 2 * WN;
    constexpr int ROWB = BK * 2, SRPI = 1024 / ROWB, CPR = ROWB / 16;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int NAI = BM / SRPI;                                  // A staging instructions per K-tile, whole workgroup
    constexpr int NA = (NAI + NW - 1) / NW, NB = BN / SRPI / NW;    // ... per wave (the last A one only on waves < NAI % NW)
    static_assert(NW * 32 * 64 * 4 <= 2 * STAGE, "epilogue LDS must fit in the ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const uint64_t t_kernel = (SCHED & 8) ? __builtin_amdgcn_s_memtime() : 0;      // diagnostic build only
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int la = xcd_remap(blockIdx.x, gridDim.x);
    int gi = 0;
#pragma unroll
    for (int q = 1; q < MAX_GROUPS; ++q)
        if (q < gp.ngroups && la >= gp.t0[q]) gi = q;
    gi = __builtin_amdgcn_readfirstlane(gi);
    // the chosen problem in SGPRs (see gemm_nt_kernel)
    GemmNT p;
    {
        const GemmNT& gq = gp.g[gi];
        p.A = uniform_ptr(gq.A), p.B = uniform_ptr(gq.B);
        p.M = uniform_i(gq.M), p.N = uniform_i(gq.N), p.K = uniform_i(gq.K), p.lda = uniform_i(gq.lda), p.ldb = uniform_i(gq.ldb);
        p.group_m = uniform_i(gq.group_m);
        p.e.out = (void*)uniform_ptr(gq.e.out), p.e.out2 = (void*)uniform_ptr(gq.e.out2);
        p.e.bias = (const float*)uniform_ptr(gq.e.bias), p.e.gamma = (const float*)uniform_ptr(gq.e.gamma);
        p.e.resid = (const float*)uniform_ptr(gq.e.resid), p.e.row_scale = (const float*)uniform_ptr(gq.e.row_scale);
        p.e.row_index = (const int32_t*)uniform_ptr(gq.e.row_index), p.e.aux = uniform_ptr(gq.e.aux);
        p.e.ldo = uniform_i(gq.e.ldo), p.e.ld2 = uniform_i(gq.e.ld2), p.e.relu = uniform_i(gq.e.relu);
        p.e.drop_thresh = (uint32_t)uniform_i((int)gq.e.drop_thresh);
        p.e.inv_keep = uniform_f(gq.e.inv_keep), p.e.beta = uniform_f(gq.e.beta);
        p.e.seed = (uint64_t)uniform_ptr((const void*)gq.e.seed);
        p.e.colpart = (float*)uniform_ptr(gq.e.colpart);
    }
    int m0, n0;
    {
        const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
        const int lid = la - uniform_i(gp.t0[gi]);
        const int gm_ = p.group_m > 0 ? p.group_m : 1;
        const int per_group = gm_ * tiles_n;
        const int first_m = (lid / per_group) * gm_;
        const int gsz = min(tiles_m - first_m, gm_);
        const int in_g = lid % per_group;
        m0 = (first_m + in_g % gsz) * BM;
        n0 = (in_g / gsz) * BN;
    }
    // staging sources: a wave-uniform 64-bit base (the tile's first row, advanced by 128 bytes per K-tile on the scalar
    // unit) + one 32-bit byte offset per lane and instruction -- half the address registers of per-lane pointers,
    // which the 320-row tile (160 accumulator registers) needs
    const char* a_base = (const char*)p.A + (size_t)m0 * p.lda * 2;
    const char* b_base = (const char*)p.B + (size_t)n0 * p.ldb * 2;
    uint32_t a_off[NA], b_off[NB];
    auto chunk_of = [&](int rr) { return (lane % CPR) ^ nt_swz<64>(rr); };
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int rr = (i * NW + wave) * SRPI + lane / CPR;
        const int gr = min(m0 + rr, p.M - 1) - m0;
        a_off[i] = (uint32_t)((gr * p.lda + chunk_of(rr) * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int rr = (i * NW + wave) * SRPI + lane / CPR;
        const int gr = min(n0 + rr, p.N - 1) - n0;
        b_off[i] = (uint32_t)((gr * p.ldb + chunk_of(rr) * 8) * 2);
    }
    // buffer form of the LDS-DMA (buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds): the K-tile advance rides in the
    // scalar offset, the per-lane part is one 32-bit register and no vector instruction precedes the load (the global_
    // form cost a 64-bit v_lshl_add per instruction: the address pairs and ~30 cycles of issue per DMA instruction)
    BufSrc a_rs, b_rs;
    a_rs.init(a_base);
    b_rs.init(b_base);
    auto stage = [&](int buf, int kt) {
        char* s = smem + buf * STAGE;
        const int koff = kt * ROWB;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if ((i + 1) * NW <= NAI || i * NW + wave < NAI)
                a_rs.load16(s + (i * NW + wave) * 1024, a_off[i], koff);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            b_rs.load16(s + A_BYTES + (i * NW + wave) * 1024, b_off[i], koff);
    };

    auto body = [&](auto tm_c, auto row0_c) __attribute__((always_inline)) {
        constexpr int TM = decltype(tm_c)::value, ROW0 = decltype(row0_c)::value;
        f32x4 acc[TM][4];
    #pragma unroll
        for (int i = 0; i < TM; ++i)
    #pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int l15 = lane & 15, g4 = lane >> 4;
        const int swz = (l15 >> 1) & 7;                 // nt_swz of the lane's row: row origins are multiples of 16
        const int a_row_off = (ROW0 + l15) * ROWB;
        const int b_row_off = A_BYTES + (wn * 64 + l15) * ROWB;
        const int nk = p.K / BK;

        stage(0, 0);
        {
            // ping-pong schedule of gemm_nt_kernel<PP>: per K-tile  read k-half 0 | MFMAs | read k-half 1 | MFMAs,  the wm == 1
            // waves one segment behind the wm == 0 waves; a k-half is ONE 32-deep MFMA step here
            v8 af[TM], bf[4];
            auto read_half = [&](const char* s_, int hf) {
                const int coff = ((4 * hf + g4) ^ swz) << 4;
    #pragma unroll
                for (int j = 0; j < 4; ++j) bf[j] = *(const v8*)(s_ + b_row_off + j * 16 * ROWB + coff);
    #pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *(const v8*)(s_ + a_row_off + i * 16 * ROWB + coff);
            };
            auto mfma_half = [&]() {
                __builtin_amdgcn_s_setprio(1);
    #pragma unroll
                for (int i = 0; i < TM; ++i)
    #pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = Elem<T>::mfma16(af[i], bf[j], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
            };
            auto bar = [&]() {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            };
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (wm == 1) bar();
            // diagnostic build (SCHED & 8): s_memtime at every segment boundary of one K loop, sums per segment kind and the
            // in-kernel clock (cycles per 100 MHz tick of s_memrealtime) -> e.colpart[workgroup][wave][8] (tools/nt16_probe.py)
            constexpr bool PROBE = (SCHED & 8) != 0;
            constexpr int SCH = SCHED & 7;
            uint64_t seg_sum[4] = {0, 0, 0, 0}, t_prev = 0, t_begin = 0, r_begin = 0;
            auto stamp = [&](int k) {
                if constexpr (PROBE) {
                    const uint64_t t = __builtin_amdgcn_s_memtime();
                    seg_sum[k] += t - t_prev;
                    t_prev = t;
                }
            };
            if constexpr (PROBE) {
                t_begin = t_prev = __builtin_amdgcn_s_memtime();
                r_begin = __builtin_amdgcn_s_memrealtime();
                // o[7] of the record: cycles from the kernel's first instruction to here (set-up, first DMA issued and landed)
                if (p.e.colpart && lane == 0)
                    ((uint64_t*)p.e.colpart)[((size_t)blockIdx.x * NW + wave) * 8 + 7] = t_begin - t_kernel;
            }
            for (int kt = 0; kt < nk; ++kt) {
                const char* cur = smem + (kt & 1) * STAGE;
                const bool more = kt + 1 < nk;
                read_half(cur, 0);
                // SCHED 1: the LDS-DMA of the next K-tile is issued in the READ segment, behind the fragment reads (the other
                // buffer was last read two segments ago by this group, one segment ago by the other, each behind lgkmcnt(0) +
                // barrier): the issue cost of the eight DMA instructions (~60-180 cycles each) then runs beside the partner
                // wave's MFMA segment instead of in front of this wave's own
                if (SCH == 1 && more) stage((kt + 1) & 1, kt + 1);
                bar();
                stamp(0);
                if (SCH == 0 && more) stage((kt + 1) & 1, kt + 1);
                mfma_half();
                bar();
                stamp(1);
                read_half(cur, 1);
                if (more && wm == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bar();
                stamp(2);
                mfma_half();
                if (more && wm == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bar();
                stamp(3);
            }
            if (wm == 0) bar();
            if constexpr (PROBE) {
                const uint64_t t_end = __builtin_amdgcn_s_memtime(), r_end = __builtin_amdgcn_s_memrealtime();
                if (p.e.colpart && lane == 0) {
                    uint64_t* o = (uint64_t*)p.e.colpart + ((size_t)blockIdx.x * NW + wave) * 8;
                    o[0] = seg_sum[0], o[1] = seg_sum[1], o[2] = seg_sum[2], o[3] = seg_sum[3];
                    o[4] = t_end - t_begin, o[5] = r_end - r_begin, o[6] = (uint64_t)nk;
                }
            }
        }

        // ---- epilogue: accumulators -> wave-private LDS -> full-row segments, in passes of two 16-row tiles (one for the last
        // tile of an odd TM); the arithmetic is gemm_nt_kernel's (epilogue4 / epilogue_ext)
        __syncthreads();
        constexpr int ROWF = 64, LPR = 16, RPI = 4;
        float* ep = (float*)(smem + wave * (32 * ROWF * 4));
        const int rrow = lane / LPR, rcol = (lane % LPR) * 4;
        const int gn = n0 + wn * 64 + rcol;
        const bool col_ok = gn < p.N;
        const int gnc = col_ok ? gn : 0;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, gamma4 = {1.f, 1.f, 1.f, 1.f};
        if (p.e.bias) bias4 = *(const f32x4*)(p.e.bias + gnc);
        if (EPI == EPI_RESID && p.e.gamma) gamma4 = *(const f32x4*)(p.e.gamma + gnc);
        // What a pass needs from global memory besides the accumulators (GELU-derivative factor / fp32 residual rows, drop-path
        // scales) is requested ONE PASS AHEAD where the registers allow it: a wave's passes were otherwise 4 - 5 serialised
        // load round trips (request -> LDS transpose -> wait -> arithmetic -> stores).  (Deeper does not pay: see DESIGN.md.)
        typedef typename ExtRaw<T, EPI>::type XR;
        constexpr int NP = (TM + 1) / 2;
        constexpr bool AHEAD = (EPI == EPI_DGELU) || (EPI == EPI_RESID && TM <= 7);      // 8 tile rows + two passes of fp32 rows: spills
        XR xr[2][8];
        float rsv[2][8];
        auto epi_load = [&](auto pi_c) __attribute__((always_inline)) {
            constexpr int P = decltype(pi_c)::value, I0 = 2 * P, NR = (I0 + 2 <= TM) ? 2 : 1;
            constexpr int NIT = NR * 16 / RPI;
            const int gmb = __builtin_amdgcn_readfirstlane(m0 + ROW0 + I0 * 16);
            const int gmbc = min(gmb, p.M - 1);
            int ridx[NIT];
    #pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int gmc = gmbc + min(it * RPI + rrow, p.M - 1 - gmbc);
                ridx[it] = (EPI == EPI_RESID && p.e.row_scale && p.e.row_index) ? p.e.row_index[gmc] : gmc;
            }
    #pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int rowc = min(it * RPI + rrow, p.M - 1 - gmbc);
                xr[P & 1][it] = epilogue_ext_raw<T, EPI>(p, gmbc, rowc, gnc);
                rsv[P & 1][it] = (EPI == EPI_RESID && p.e.row_scale) ? p.e.row_scale[ridx[it]] : 1.f;
            }
        };
        auto epi_pass = [&](auto pi_c) __attribute__((always_inline)) {
            constexpr int P = decltype(pi_c)::value, I0 = 2 * P, NR = (I0 + 2 <= TM) ? 2 : 1;
            constexpr int NIT = NR * 16 / RPI;
            const int gmb = __builtin_amdgcn_readfirstlane(m0 + ROW0 + I0 * 16);
            if constexpr (!AHEAD) epi_load(pi_c);
            // C/D map of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + register
    #pragma unroll
            for (int ii = 0; ii < NR; ++ii)
    #pragma unroll
                for (int j = 0; j < 4; ++j)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) ep[(ii * 16 + 4 * g4 + r) * ROWF + j * 16 + l15] = acc[I0 + ii][j][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            f32x4 v[NIT];
    #pragma unroll
            for (int it = 0; it < NIT; ++it) v[it] = *(const f32x4*)(ep + (it * RPI + rrow) * ROWF + rcol);
            if constexpr (AHEAD && P + 1 < NP) epi_load(std::integral_constant<int, P + 1>{});
            f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = it * RPI + rrow;
                const bool ok = gmb + row < p.M && col_ok;
                const f32x4 w = epilogue4<T, EPI, GD>(p, gmb, row, gn, v[it], bias4, gamma4, ext_f32<T>(xr[P & 1][it]), rsv[P & 1][it], ok);
                if constexpr (EPI == EPI_DGELU) {
    #pragma unroll
                    for (int j = 0; j < 4; ++j) csum[j] += ok ? w[j] : 0.f;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if constexpr (EPI == EPI_DGELU) {
                // column sums of this pass -> colpart row of its FIRST 16-row block, zeros to the second one's (every row of
                // colpart[ceil(M/16)] has exactly one writer, whatever the tile height)
                if (p.e.colpart) {
    #pragma unroll
                    for (int o = LPR; o < 64; o <<= 1)
    #pragma unroll
                        for (int j = 0; j < 4; ++j) csum[j] += __shfl_xor(csum[j], o, 64);
                    const int blk = gmb >> 4;
                    if (lane < LPR && col_ok && gmb < p.M) {
                        *(f32x4*)(p.e.colpart + (size_t)blk * p.N + gn) = csum;
                        if (NR == 2 && gmb + 16 < p.M) *(f32x4*)(p.e.colpart + (size_t)(blk + 1) * p.N + gn) = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        };
        if constexpr (AHEAD) epi_load(std::integral_constant<int, 0>{});
        static_for(std::make_integer_sequence<int, NP>{}, [&](auto pi) __attribute__((always_inline)) {
            epi_pass(std::integral_constant<int, decltype(pi)::value>{});
        });
    };
    if (wm == 0)
        body(std::integral_constant<int, TMA>{}, std::integral_constant<int, 0>{});
    else
        body(std::integral_constant<int, TMB>{}, std::integral_constant<int, 16 * TMA>{});
}

// ------------------------------------------------------------------ wgrad ---
struct GemmTN {
    const void* A;   // [M, lda], uses columns [0, N1)
    const void* B;   // [M, ldb], uses columns [0, N2)
    float* C;        // [N1, ldc] fp32, atomically accumulated
    int M, N1, N2, lda, ldb, ldc;
    int kt_per_split, tiles;
    float alpha;
    float* slab;     // [splits, N1, N2] fp32 partial products (plain stores) or NULL (see mode)
    int mode;        // without a slab: 0 = fp32 atomics into C, 1 = C += alpha * acc (this workgroup owns the
                     // tile: no K split), 2 = C = alpha * acc
};
enum { TN_ATOMIC = 0, TN_ACCUM = 1, TN_STORE = 2 };

#ifndef VLMO_TN_DMA_IN_READ
#define VLMO_TN_DMA_IN_READ 1
#endif
constexpr bool TN_DMA_IN_READ = VLMO_TN_DMA_IN_READ != 0;      // build-time A/B (make EXTRA=-DVLMO_TN_DMA_IN_READ=0)

// 256 zero bytes: the staging source of token rows past the end of the reduction dimension
__device__ __attribute__((aligned(256))) char tn_zero_page[256];

// dual-use 256-byte-row image: chunk swizzle serving the transposed reads
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// Output tile BM x BN (multiples of 128) per workgroup of WM x WN waves; each operand's K-tile (64 token rows)
// is staged as BM/128 resp. BN/128 side-by-side sub-images of [64 rows][128 columns] in the dual-use swizzle.
// R4 (ping-pong kernel only): the reduction is staged in 32-token SLICES through a ring of four 32 KB slots instead of
// 64-token tiles through two 64 KB buffers.  The reduction index of this kernel is the ROW of both operands, so a slice
// is still made of whole 256-byte row pieces (the NT kernel cannot do this: its K runs along the rows, half a K-tile is
// half of every cache line).  Slice h + 3 is issued in the read segment of slice h -- four DMA instructions per wave
// and segment instead of eight in every other one -- and waited for with a counted vmcnt that leaves two slices in flight.
template <typename T, int BM, int BN, int WM, int WN, bool PP = false, bool PROBE = false, bool R4 = false, bool M16 = false>
__device__ __forceinline__ void gemm_tn_body(const GemmTN& p, const int lid) {
    typedef typename Elem<T>::v8 v8;
    typedef typename Elem<T>::v4 v4;
    constexpr int NW = WM * WN;
    constexpr int SROWS = R4 ? 32 : 64;                 // token rows per staged unit
    constexpr int SUB = SROWS * 256;                    // one sub-image
    constexpr int NSA = BM / 128, NSB = BN / 128;
    constexpr int A_BYTES = NSA * SUB, STAGE = (NSA + NSB) * SUB;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int IPS = SROWS / 4;                      // LDS-DMA instructions per sub-image and staged unit
    constexpr int IA = NSA * IPS / NW, IB = NSB * IPS / NW;   // ... per wave
    static_assert((NSA * IPS) % NW == 0 && (NSB * IPS) % NW == 0, "tile/wave mismatch");
    static_assert(!R4 || (PP && IA + IB == 4), "slice ring: ping-pong schedule, four DMA instructions per wave and slice");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // 1-D grid of splits x tiles, remapped so that the workgroups one XCD runs are consecutive
    // (split-major): the ~32 tiles of one K-split share that split's token rows through the XCD's L2
    const int tiles_n = (p.N2 + BN - 1) / BN;
    const int split = lid / p.tiles, tile = lid - split * p.tiles;
    const int n1_0 = (tile / tiles_n) * BM, n2_0 = (tile % tiles_n) * BN;
    const int nk_total = (p.M + 63) >> 6;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(nk_total, kt0 + p.kt_per_split);
    if (kt0 >= kt1) return;

    // staging: one wave-instruction = 4 rows x 256 B of one sub-image; 16 instructions per sub-image
    int a_off[IA], b_off[IB], a_row[IA], b_row[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int ii = i * NW + wave, sub = ii / IPS, row = (ii % IPS) * 4 + (lane >> 4);
        const int ch = (lane & 15) ^ tn_swz(row);
        a_row[i] = row;
        a_off[i] = min(n1_0 + sub * 128 + ch * 8, p.N1 - 8);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int ii = i * NW + wave, sub = ii / IPS, row = (ii % IPS) * 4 + (lane >> 4);
        const int ch = (lane & 15) ^ tn_swz(row);
        b_row[i] = row;
        b_off[i] = min(n2_0 + sub * 128 + ch * 8, p.N2 - 8);
    }
    // Full units go out in the buffer form of the LDS-DMA (common.h BufSrc): descriptor + scalar unit offset + one fixed 32-bit
    // lane offset, no vector instruction and no scalar load in front of the DMA.  (The pointer form below computed a 64-bit
    // multiply-add per instruction, selected the zero page under a divergent exec mask and fetched that page's address
    // through the GOT with s_load + s_waitcnt lgkmcnt(0) -- which also waited for the fragment reads just issued: ~450
    // cycles per staging call in the stamped build.)  The ragged last unit keeps the pointer form.
    const bool fits32 = (uint64_t)p.M * (uint64_t)max(p.lda, p.ldb) * 2 < 0x7FFFFFFFull;
    BufSrc a_rs, b_rs;
    a_rs.init(p.A);
    b_rs.init(p.B);
    uint32_t a_vo[IA], b_vo[IB];
#pragma unroll
    for (int i = 0; i < IA; ++i) a_vo[i] = (uint32_t)(a_row[i] * p.lda + a_off[i]) * 2u;
#pragma unroll
    for (int i = 0; i < IB; ++i) b_vo[i] = (uint32_t)(b_row[i] * p.ldb + b_off[i]) * 2u;
    // kt: index of the staged unit (64-token K-tile, or 32-token slice with R4)
    auto stage = [&](int buf, int kt) {
        char* s = smem + buf * STAGE;
        if (PP && fits32 && (kt + 1) * SROWS <= p.M) {
            const int sa = kt * SROWS * p.lda * 2, sb = kt * SROWS * p.ldb * 2;
#pragma unroll
            for (int i = 0; i < IA; ++i) a_rs.load16(s + (i * NW + wave) * 1024, a_vo[i], sa);
#pragma unroll
            for (int i = 0; i < IB; ++i) b_rs.load16(s + A_BYTES + (i * NW + wave) * 1024, b_vo[i], sb);
            return;
        }
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            // token rows past M (ragged last K-tile) are staged from a zero page: operand A is then exactly zero
            // there, so the K loop needs no masking (B keeps the clamped last row; 0 * finite = 0)
            const int gr = kt * SROWS + a_row[i];
            const T* src = gr < p.M ? (const T*)p.A + (size_t)gr * p.lda + a_off[i] : (const T*)tn_zero_page;
            glds16(src, s + (i * NW + wave) * 1024);
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int gr = min(kt * SROWS + b_row[i], p.M - 1);
            glds16((const T*)p.B + (size_t)gr * p.ldb + b_off[i], s + A_BYTES + (i * NW + wave) * 1024);
        }
    };

    if constexpr (M16) {
        // ---- v_mfma_f32_16x16x32 variant of the slice ring (round 4; the NT kernels' port measured +8.6 % at the MFMA-paced
        // probe shape): a 32-token slice is ONE k-step.  Operand lane l holds column l & 15, tokens 8 (l >> 4) + j: two
        // transposed reads of 4 tokens x 16 columns each (tests/test_lds_layouts.py::test_gemm_tn16_fragments: right elements,
        // conflict-free on the unchanged image).  C/D: column = l & 15, row = 4 (l >> 4) + register.
        static_assert(PP && R4 && WM == 2, "16x16x32: slice ring, ping-pong schedule");
        constexpr int T16M = BM / WM / 16, T16N = BN / WN / 16;
        f32x4 acc[T16M][T16N];
#pragma unroll
        for (int i = 0; i < T16M; ++i)
#pragma unroll
            for (int j = 0; j < T16N; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, l15 = lane & 15;
        const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
        uint32_t a_rd[2][T16M], b_rd[2][T16N];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int m0 = 8 * g + 4 * half + q;
            const int sw = tn_swz(m0);
#pragma unroll
            for (int t = 0; t < T16M; ++t) {
                const int na = wm * (BM / WM) + t * 16 + 4 * pp;
                a_rd[half][t] = lds0 + (na >> 7) * SUB + m0 * 256 + ((((na & 127) >> 3) ^ sw) << 4) + (na & 7) * 2;
            }
#pragma unroll
            for (int t = 0; t < T16N; ++t) {
                const int nb = wn * (BN / WN) + t * 16 + 4 * pp;
                b_rd[half][t] = lds0 + A_BYTES + (nb >> 7) * SUB + m0 * 256 + ((((nb & 127) >> 3) ^ sw) << 4) + (nb & 7) * 2;
            }
        }
        v8 af[T16M], bf[T16N];
        typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
        auto tr = [&](uint32_t addr) -> v4 {        // inline asm: see the 32x32x16 path below (hipcc waits vmcnt(0) behind the builtin)
            u32x2 r;
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
            return __builtin_bit_cast(v4, r);
        };
        auto read_slice = [&](int slot) {
            const uint32_t so = (uint32_t)slot * STAGE;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int t = 0; t < T16N; ++t) {
                    const v4 vb = tr(b_rd[half][t] + so);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bf[t][4 * half + e] = vb[e];
                }
#pragma unroll
                for (int t = 0; t < T16M; ++t) {
                    const v4 va = tr(a_rd[half][t] + so);
#pragma unroll
                    for (int e = 0; e < 4; ++e) af[t][4 * half + e] = va[e];
                }
            }
        };
        auto mfma_slice = [&]() {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < T16M; ++i)
#pragma unroll
                for (int j = 0; j < T16N; ++j) acc[i][j] = Elem<T>::mfma16(af[i], bf[j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        };
        auto bar = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        const int ns = 2 * (kt1 - kt0), s0 = 2 * kt0;
        auto wait_next = [&](int h) {
            if (h + 3 < ns) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (h + 2 < ns) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        stage(0, s0);
        stage(1, s0 + 1);
        if (ns > 2) stage(2, s0 + 2);
        wait_next(-1);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) bar();
        for (int h_ = 0; h_ < ns; ++h_) {
            read_slice(h_ & 3);
            if (h_ + 3 < ns) stage((h_ + 3) & 3, s0 + h_ + 3);
            if (wm == 1 && h_ + 1 < ns) wait_next(h_);
            bar();
            mfma_slice();
            if (wm == 0 && h_ + 1 < ns) wait_next(h_);
            bar();
        }
        if (wm == 0) bar();
        // epilogue: lane = output column (16 per tile), registers = four consecutive output rows
#pragma unroll
        for (int i = 0; i < T16M; ++i)
#pragma unroll
            for (int j = 0; j < T16N; ++j) {
                const int gn = n2_0 + wn * (BN / WN) + j * 16 + l15;
                const int gm0 = n1_0 + wm * (BM / WM) + i * 16 + 4 * g;
                const bool okn = gn < p.N2;
                if (!p.slab && p.mode == TN_ACCUM) {
                    float old[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) old[r] = (okn && gm0 + r < p.N1) ? p.C[(size_t)(gm0 + r) * p.ldc + gn] : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (okn && gm0 + r < p.N1) p.C[(size_t)(gm0 + r) * p.ldc + gn] = old[r] + p.alpha * acc[i][j][r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gm = gm0 + r;
                        if (gm < p.N1 && okn) {
                            float* c = p.C + (size_t)gm * p.ldc + gn;
                            if (p.slab)
                                p.slab[((size_t)split * p.N1 + gm) * p.N2 + gn] = acc[i][j][r];
                            else if (p.mode == TN_STORE)
                                *c = p.alpha * acc[i][j][r];
                            else
                                atomicAdd(c, p.alpha * acc[i][j][r]);
                        }
                    }
                }
            }
        return;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    // transposed-read addressing (ds_read_b64_tr_b16): lane 4q+p of a 16-lane
    // group supplies row q, columns 4p..4p+3 of a 4x16 block
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, h = lane >> 5;
    const int ncol_a = wm * (BM / WM) + 16 * (g & 1) + 4 * pp;   // + tile*32
    const int ncol_b = wn * (BN / WN) + 16 * (g & 1) + 4 * pp;

    // LDS byte offsets of this lane's transposed reads inside a K-tile image, loop-invariant: the k-step enters
    // additively (16 rows = 4096 B: an immediate offset of the ds_read), only (half, t) need their own register
    // because the chunk swizzle is an XOR.  tn_swz(16 ks + m0) == tn_swz(m0).
    int a_rd[2][TM], b_rd[2][TN];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int m0 = 8 * h + 4 * half + q;
        const int sw = tn_swz(m0);
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int na = ncol_a + t * 32;
            a_rd[half][t] = (na >> 7) * SUB + m0 * 256 + ((((na & 127) >> 3) ^ sw) << 4) + (na & 7) * 2;
        }
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int nb = ncol_b + t * 32;
            b_rd[half][t] = A_BYTES + (nb >> 7) * SUB + m0 * 256 + ((((nb & 127) >> 3) ^ sw) << 4) + (nb & 7) * 2;
        }
    }
    // fragments of one 16-row k-step ks of the K-tile in image s_.
    // In the ping-pong schedule the transposed reads are INLINE ASM: behind the ds_read_tr builtin hipcc (ROCm 7.2)
    // waits `vmcnt(0)` for every LDS-DMA in flight (the builtin carries no memory operand, so the waitcnt pass
    // assumes it reads what the DMA writes), which put the whole HBM latency of the next K-tile's staging in front
    // of the second read segment of every tile (2.3-3.0 us per K-tile where the plain-load NT kernel takes 1.7).
    // The schedule orders DMA and reads itself (counted vmcnt + barriers); the reads' results are consumed only
    // behind bar() = lgkmcnt(0) + s_barrier + sched_barrier.
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
    auto tr_read = [&](const char* s_, int off, int imm) -> v4 {
        if constexpr (PP) {
            typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
            u32x2 r;
            const uint32_t addr = lds0 + (uint32_t)(s_ - smem) + (uint32_t)off;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(imm));
            return __builtin_bit_cast(v4, r);
        } else {
            return lds_tr4<T>(s_ + off + imm);
        }
    };
    auto read_step = [&](const char* s_, int kt, int ks, v8* af, v8* bf) {
        (void)kt;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const v4 va = ks == 0 ? tr_read(s_, a_rd[half][t], 0) : ks == 1 ? tr_read(s_, a_rd[half][t], 4096)
                            : ks == 2 ? tr_read(s_, a_rd[half][t], 8192) : tr_read(s_, a_rd[half][t], 12288);
#pragma unroll
                for (int e = 0; e < 4; ++e) af[t][4 * half + e] = va[e];
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const v4 vb = ks == 0 ? tr_read(s_, b_rd[half][t], 0) : ks == 1 ? tr_read(s_, b_rd[half][t], 4096)
                            : ks == 2 ? tr_read(s_, b_rd[half][t], 8192) : tr_read(s_, b_rd[half][t], 12288);
#pragma unroll
                for (int e = 0; e < 4; ++e) bf[t][4 * half + e] = vb[e];
            }
        }
    };
    if constexpr (PP && R4) {
        static_assert(WM == 2, "ping-pong schedule needs two row groups");
        v8 af[2][TM], bf[2][TN];
        auto mfma_half = [&]() {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = Elem<T>::mfma(af[u][i], bf[u][j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        };
        auto bar = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        const int ns = 2 * (kt1 - kt0), s0 = 2 * kt0;       // slices of this workgroup
        // this wave's DMA of slice h + 1 has landed; slices h + 2 and h + 3 (four instructions each) may stay in flight
        auto wait_next = [&](int h) {
            if (h + 3 < ns) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (h + 2 < ns) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        stage(0, s0);
        stage(1, s0 + 1);
        if (ns > 2) stage(2, s0 + 2);
        wait_next(-1);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) bar();
        uint64_t seg_sum[4] = {0, 0, 0, 0}, t_prev = 0, t_begin = 0, r_begin = 0;
        auto stamp = [&](int k) {
            if constexpr (PROBE) {
                const uint64_t t = __builtin_amdgcn_s_memtime();
                seg_sum[k] += t - t_prev;
                t_prev = t;
            }
        };
        if constexpr (PROBE) {
            t_begin = t_prev = __builtin_amdgcn_s_memtime();
            r_begin = __builtin_amdgcn_s_memrealtime();
        }
        for (int h_ = 0; h_ < ns; ++h_) {
            const char* cur = smem + (h_ & 3) * STAGE;
            read_step(cur, 0, 0, af[0], bf[0]);
            read_step(cur, 0, 1, af[1], bf[1]);
            // slot (h + 3) & 3 held slice h - 1: read by this group two segments ago, by the other one segment ago
            if (h_ + 3 < ns) stage((h_ + 3) & 3, s0 + h_ + 3);
            if (wm == 1 && h_ + 1 < ns) wait_next(h_);
            bar();
            stamp((h_ & 1) * 2);
            mfma_half();
            if (wm == 0 && h_ + 1 < ns) wait_next(h_);
            bar();
            stamp((h_ & 1) * 2 + 1);
        }
        if (wm == 0) bar();
        if constexpr (PROBE) {
            const uint64_t t_end = __builtin_amdgcn_s_memtime(), r_end = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) {
                uint64_t* o = (uint64_t*)p.slab + ((size_t)blockIdx.x * NW + wave) * 8;
                o[0] = seg_sum[0], o[1] = seg_sum[1], o[2] = seg_sum[2], o[3] = seg_sum[3];
                o[4] = t_end - t_begin, o[5] = r_end - r_begin, o[6] = (uint64_t)(kt1 - kt0), o[7] = 1;
            }
            return;
        }
    } else if constexpr (PP) {
        // ping-pong schedule: see gemm_nt_kernel (wm == 1 waves run one segment behind wm == 0)
        static_assert(WM == 2, "ping-pong schedule needs two row groups");
        v8 af[2][TM], bf[2][TN];
        auto mfma_half = [&]() {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = Elem<T>::mfma(af[u][i], bf[u][j], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        };
        auto bar = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        // One whole K-tile is staged at the start of the first MFMA segment of the previous tile and waited for by the
        // issuing wave before the barrier that opens it.  Measured against a half-tile variant with 1.5 K-tiles in
        // flight and counted waits (DMA issued in the read segments): 610 vs 650 us for the two-block launch, 527 vs
        // 585 us with L2-resident operands -- what bounds this loop is the CU's L1 fill rate, not exposed latency, and
        // DMA issue beside the partner group's LDS reads costs more than the extra half tile in flight buys.
        stage(0, kt0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wm == 1) bar();
        // diagnostic build (PROBE): cycles per segment kind and the in-kernel clock -> p.slab[workgroup][wave][8] (uint64)
        uint64_t seg_sum[4] = {0, 0, 0, 0}, t_prev = 0, t_begin = 0, r_begin = 0;
        auto stamp = [&](int k) {
            if constexpr (PROBE) {
                const uint64_t t = __builtin_amdgcn_s_memtime();
                seg_sum[k] += t - t_prev;
                t_prev = t;
            }
        };
        if constexpr (PROBE) {
            t_begin = t_prev = __builtin_amdgcn_s_memtime();
            r_begin = __builtin_amdgcn_s_memrealtime();
        }
        for (int kt = kt0; kt < kt1; ++kt) {
            const char* cur = smem + ((kt - kt0) & 1) * STAGE;
            const bool more = kt + 1 < kt1;
            read_step(cur, kt, 0, af[0], bf[0]);
            read_step(cur, kt, 1, af[1], bf[1]);
            if (TN_DMA_IN_READ && more) stage((kt - kt0 + 1) & 1, kt + 1);      // see gemm_nt_kernel: DMA issue beside the partner's MFMAs
            bar();
            stamp(0);
            if (!TN_DMA_IN_READ && more) stage((kt - kt0 + 1) & 1, kt + 1);
            mfma_half();
            bar();
            stamp(1);
            read_step(cur, kt, 2, af[0], bf[0]);
            read_step(cur, kt, 3, af[1], bf[1]);
            if (more && wm == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bar();
            stamp(2);
            mfma_half();
            if (more && wm == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bar();
            stamp(3);
        }
        if (wm == 0) bar();
        if constexpr (PROBE) {
            const uint64_t t_end = __builtin_amdgcn_s_memtime(), r_end = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) {
                uint64_t* o = (uint64_t*)p.slab + ((size_t)blockIdx.x * NW + wave) * 8;
                o[0] = seg_sum[0], o[1] = seg_sum[1], o[2] = seg_sum[2], o[3] = seg_sum[3];
                o[4] = t_end - t_begin, o[5] = r_end - r_begin, o[6] = (uint64_t)(kt1 - kt0), o[7] = 0;
            }
            return;     // the probe build leaves C alone
        }
    } else {
        stage(0, kt0);
        for (int kt = kt0; kt < kt1; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < kt1) stage((kt - kt0 + 1) & 1, kt + 1);
            const char* s = smem + ((kt - kt0) & 1) * STAGE;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                v8 af[TM], bf[TN];
                read_step(s, kt, ks, af, bf);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = Elem<T>::mfma(af[i], bf[j], acc[i][j]);
            }
        }
    }
    // epilogue: lane = output column, register = output row: each half-wave
    // writes / adds 128 contiguous bytes
    const int l31 = lane & 31;
    if (!p.slab && p.mode == TN_ACCUM) {
        // in-place accumulate by the tile's only owner: all 16 loads of a 32x32 sub-tile are issued before the
        // first add (one HBM round trip per sub-tile; a load -> add -> store chain per element costs 128 of them,
        // ~200 us per workgroup in the first build of this kernel)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int gn = n2_0 + wn * (BN / WN) + j * 32 + l31;
                const int gm0 = n1_0 + wm * (BM / WM) + i * 32 + 4 * h;
                const bool okn = gn < p.N2;
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = gm0 + (r & 3) + 8 * (r >> 2);
                    old[r] = (okn && gm < p.N1) ? p.C[(size_t)gm * p.ldc + gn] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gm = gm0 + (r & 3) + 8 * (r >> 2);
                    if (okn && gm < p.N1) p.C[(size_t)gm * p.ldc + gn] = old[r] + p.alpha * acc[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = n2_0 + wn * (BN / WN) + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = n1_0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (gm < p.N1 && gn < p.N2) {
                    float* c = p.C + (size_t)gm * p.ldc + gn;
                    if (p.slab)
                        p.slab[((size_t)split * p.N1 + gm) * p.N2 + gn] = acc[i][j][r];
                    else if (p.mode == TN_STORE)
                        *c = p.alpha * acc[i][j][r];
                    else
                        atomicAdd(c, p.alpha * acc[i][j][r]);
                }
            }
        }
}

template <typename T, int BM, int BN, int WM, int WN, bool PP = false, bool PROBE = false, bool R4 = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_tn_kernel(const GemmTN p) {
    gemm_tn_body<T, BM, BN, WM, WN, PP, PROBE, R4>(p, xcd_remap(blockIdx.x, gridDim.x));
}

// Up to MAX_TN_PROBS weight-gradient problems in ONE launch of 256x256 tiles (the four to six linears of one or
// two transformer blocks): problem q owns the logical workgroups [t0[q], t0[q+1]).  A single weight gradient of
// VLMo-Base has 9-36 output tiles, so on its own it needs a 7-way split of the token dimension (slabs + a reduction
// pass, or atomics) to occupy 256 CUs; the gradients of two blocks together have 216 tiles and need no split at all.
constexpr int MAX_TN_PROBS = 16;
// Workgroup b of the launch runs order[b] = (problem << 12) | (workgroup index inside the problem), or nothing
// (TN_NOP).  The host fills the table so that the workgroups one XCD receives (b % 8 equal, dealt round-robin) are
// a BALANCED mix: with problems of different reduction lengths in one launch (below the fusion layer: 261, 197 and
// 64 K-tiles) contiguous XCD chunks gave one XCD 45 long tiles for its 32 CUs -- two rounds, 1 035 us instead of
// 525 -- while another finished its 45 short ones in a quarter of the time.
constexpr int MAX_TN_ORDER = 1024;
constexpr uint16_t TN_NOP = 0xFFFF;
struct GemmTNMulti {
    int n;
    GemmTN p[MAX_TN_PROBS];
    uint16_t order[MAX_TN_ORDER];
};
template <typename T, bool R4 = false, bool M16 = false>
__global__ __launch_bounds__(512, 2) void gemm_tn_multi_kernel(const GemmTNMulti mp) {
    const uint32_t code = mp.order[blockIdx.x];
    if (code == TN_NOP) return;
    int gi = (int)(code >> 12);
    const int lid_in = __builtin_amdgcn_readfirstlane((int)(code & 0xFFFu));
    gi = __builtin_amdgcn_readfirstlane(gi);
    // copy the chosen problem into SGPRs ONCE: a dynamically indexed kernarg struct is otherwise re-read with
    // s_load + s_waitcnt at every use inside the K loop (908 scalar loads in the first build of this kernel)
    const GemmTN& q = mp.p[gi];
    GemmTN p;
    p.A = uniform_ptr(q.A), p.B = uniform_ptr(q.B), p.C = (float*)uniform_ptr(q.C);
    p.M = __builtin_amdgcn_readfirstlane(q.M), p.N1 = __builtin_amdgcn_readfirstlane(q.N1);
    p.N2 = __builtin_amdgcn_readfirstlane(q.N2), p.lda = __builtin_amdgcn_readfirstlane(q.lda);
    p.ldb = __builtin_amdgcn_readfirstlane(q.ldb), p.ldc = __builtin_amdgcn_readfirstlane(q.ldc);
    p.kt_per_split = __builtin_amdgcn_readfirstlane(q.kt_per_split), p.tiles = __builtin_amdgcn_readfirstlane(q.tiles);
    p.alpha = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, q.alpha)));
    p.slab = nullptr;
    p.mode = __builtin_amdgcn_readfirstlane(q.mode);
    gemm_tn_body<T, 256, 256, 2, 4, true, false, R4, M16>(p, lid_in);
}

// C[r, c] += alpha * sum_s slab[s, r, c]   (one float4 per thread)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ slab, int splits, int N1, int N2,
                                                        float* __restrict__ C, int ldc, float alpha) {
    const int n4 = N2 >> 2;
    const long total = (long)N1 * n4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / n4), c = (int)(i % n4) * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < splits; ++s2) a += *(const f32x4*)(slab + ((size_t)s2 * N1 + r) * N2 + c);
        f32x4* o = (f32x4*)(C + (size_t)r * ldc + c);
        *o = *o + alpha * a;
    }
}

// EMASK: bit e set = epilogue e is instantiated for this tile shape (every instantiation costs build time and code size)
template <typename T, int BM, int BN, int WM, int WN, bool CONV = false, int BK = 64, int NSTG = 2, bool PP = false,
          unsigned EMASK = 0xFFFFFFFFu>
int launch_nt(int epi, GemmNTGroups& p, hipStream_t st) {
    int tiles = 0;
    for (int q = 0; q < p.ngroups; ++q) {
        p.t0[q] = tiles;
        tiles += ((p.g[q].M + BM - 1) / BM) * ((p.g[q].N + BN - 1) / BN);
    }
    for (int q = p.ngroups; q <= MAX_GROUPS; ++q) p.t0[q] = tiles;
    constexpr int LDS = NSTG * (BM + BN) * BK * 2;
    dim3 grid(tiles), block(WM * WN * 64);
#define VLMO_LAUNCH_EPI(E)                                                                     \
    case E:                                                                                    \
    if constexpr (((EMASK >> E) & 1u) == 0) {                                                  \
        known = false;                                                                         \
    } else {                                                                                   \
        auto k = gemm_nt_kernel<T, BM, BN, WM, WN, E, CONV, BK, NSTG, PP>;                                        \
        if (LDS > 65536) {                                                                     \
            static DeviceOnce attr_set;        /* per kernel instantiation AND per device */      \
            if (attr_set.first())                                                              \
                (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
        }                                                                                      \
        hipLaunchKernelGGL(k, grid, block, LDS, st, p);                                        \
    } break;
    bool known = true;
    switch (epi) {
        VLMO_LAUNCH_EPI(EPI_BIAS)
        VLMO_LAUNCH_EPI(EPI_F32)
        VLMO_LAUNCH_EPI(EPI_DUAL)
        default:
            if constexpr (!CONV) {
                switch (epi) {
                    VLMO_LAUNCH_EPI(EPI_BIAS_GELU)
                    VLMO_LAUNCH_EPI(EPI_RESID)
                    VLMO_LAUNCH_EPI(EPI_DGELU)
                    VLMO_LAUNCH_EPI(EPI_ARGMAX)
                    VLMO_LAUNCH_EPI(EPI_CE)
                    VLMO_LAUNCH_EPI(EPI_CE_BWD)
                    default:
                        known = false;
                }
            } else {
                known = false;
            }
    }
    if (!known) {
        vlmo_set_error("vlmo_gemm_nt/conv: unsupported epilogue %d", epi);
        return -1;
    }
#undef VLMO_LAUNCH_EPI
    VLMO_CHECK_LAUNCH("vlmo_gemm_nt");
    return 0;
}

// 16x16x32 kernels: (16 * H16) x 256 tiles, one workgroup per CU
template <typename T, int H16, int SCHED = 1, unsigned EMASK = 0xFu>
int launch_nt16(int epi, GemmNTGroups& p, hipStream_t st) {
    constexpr int BM = 16 * H16, BN = 256, BK = 64;
    int tiles = 0;
    for (int q = 0; q < p.ngroups; ++q) {
        p.t0[q] = tiles;
        tiles += ((p.g[q].M + BM - 1) / BM) * ((p.g[q].N + BN - 1) / BN);
    }
    for (int q = p.ngroups; q <= MAX_GROUPS; ++q) p.t0[q] = tiles;
    constexpr int LDS = 2 * (BM + BN) * BK * 2;
    dim3 grid(tiles), block(512);
    bool known = true;
    // saved-GELU-derivative variant (VlmoEpilogue.relu bit 2): one choice per launch, every group must agree
    const bool gd = (p.g[0].e.relu & 4) != 0;
    for (int q = 1; q < p.ngroups; ++q)
        if (((p.g[q].e.relu & 4) != 0) != gd) {
            vlmo_set_error("vlmo_gemm_nt_grouped: the groups of a launch must agree on VlmoEpilogue.relu bit 2");
            return -1;
        }
#define VLMO_LAUNCH16(E)                                                                        \
    case E:                                                                                     \
    if constexpr (((EMASK >> E) & 1u) == 0) {                                                   \
        known = false;                                                                          \
    } else {                                                                                    \
        constexpr bool HAS_GD = (E == EPI_BIAS_GELU || E == EPI_DGELU);                         \
        if (HAS_GD && gd) {                                                                     \
            auto k = gemm_nt16_kernel<T, H16, E, SCHED, HAS_GD ? 1 : 0>;                        \
            static DeviceOnce attr_set;                                                         \
            if (LDS > 65536 && attr_set.first())                                                \
                (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
            hipLaunchKernelGGL(k, grid, block, LDS, st, p);                                     \
        } else {                                                                                \
            auto k = gemm_nt16_kernel<T, H16, E, SCHED, 0>;                                     \
            static DeviceOnce attr_set;                                                         \
            if (LDS > 65536 && attr_set.first())                                                \
                (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
            hipLaunchKernelGGL(k, grid, block, LDS, st, p);                                     \
        }                                                                                       \
    } break;
    switch (epi) {
        VLMO_LAUNCH16(EPI_BIAS)
        VLMO_LAUNCH16(EPI_BIAS_GELU)
        VLMO_LAUNCH16(EPI_RESID)
        VLMO_LAUNCH16(EPI_DGELU)
        default:
            known = false;
    }
#undef VLMO_LAUNCH16
    if (!known) {
        vlmo_set_error("vlmo_gemm_nt: this 16x16x32 tile is not built with epilogue %d", epi);
        return -1;
    }
    VLMO_CHECK_LAUNCH("vlmo_gemm_nt");
    return 0;
}

}  // namespace

// ---- optional in-library timing of GEMM launches (bench.py's roofline): HIP event pairs recorded on the
// launch stream around every vlmo_gemm_nt / vlmo_gemm_tn / vlmo_conv2d_nhwc while profiling is on.
namespace {
struct ProfRec {
    hipEvent_t a, b;
    int tag;
    double flops;
};
struct Prof {
    std::vector<ProfRec> recs;
    size_t used = 0;
    bool on = false;
    std::mutex mu;
} g_prof;

struct ProfScope {
    ProfRec* r = nullptr;
    hipStream_t st;
    ProfScope(int tag, double flops, hipStream_t s) : st(s) {
        if (!g_prof.on) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (g_prof.used < g_prof.recs.size()) {
            r = &g_prof.recs[g_prof.used++];
            r->tag = tag;
            r->flops = flops;
            (void)hipEventRecord(r->a, st);
        }
    }
    ~ProfScope() {
        if (r) (void)hipEventRecord(r->b, st);
    }
};
}  // namespace

extern "C" int vlmo_profile_start(int max_records) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    while ((int)g_prof.recs.size() < max_records) {
        ProfRec r{};
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
            vlmo_set_error("vlmo_profile_start: hipEventCreate failed");
            return -1;
        }
        g_prof.recs.push_back(r);
    }
    g_prof.used = 0;
    g_prof.on = true;
    return 0;
}

// Stops recording and sums per tag (tag = epilogue id for gemm_nt, +16 when the 256x256 tile ran; 32 + epilogue for
// conv; 48 + epilogue for the 256x128 tile; 64 / 72 for gemm_tn 128x128 / 256x256, 73 for gemm_tn_multi; 80 + epilogue for
// the 16x16x32 kernels of any tile height).  Call after the stream(s) have been synchronised.
extern "C" int vlmo_profile_stop(int ntags, double* ms, double* flops, int64_t* launches) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = false;
    for (int i = 0; i < ntags; ++i) {
        ms[i] = 0;
        flops[i] = 0;
        launches[i] = 0;
    }
    for (size_t i = 0; i < g_prof.used; ++i) {
        const ProfRec& r = g_prof.recs[i];
        float t = 0.f;
        if (r.tag < 0 || r.tag >= ntags || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
        ms[r.tag] += t;
        flops[r.tag] += r.flops;
        launches[r.tag] += 1;
    }
    return (int)g_prof.used;
}

namespace {
int check_nt(int epi, const void* A, int lda, const void* B, int ldb, int M, int N, int K, const VlmoEpilogue* e) {
    VLMO_CHECK_ARG(A && B && e, "vlmo_gemm_nt: null operand");
    VLMO_CHECK_ARG(M > 0 && N > 0 && K > 0, "vlmo_gemm_nt: empty problem M=%d N=%d K=%d", M, N, K);
    VLMO_CHECK_ARG(K % 64 == 0, "vlmo_gemm_nt: K=%d must be a multiple of 64", K);
    VLMO_CHECK_ARG(N % 4 == 0, "vlmo_gemm_nt: N=%d must be a multiple of 4", N);
    VLMO_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K, "vlmo_gemm_nt: bad lda/ldb %d/%d", lda, ldb);
    VLMO_CHECK_ARG(e->out && (epi == EPI_ARGMAX || epi == EPI_CE || (e->ldo >= N && e->ldo % 4 == 0)), "vlmo_gemm_nt: bad output / ldo");
    VLMO_CHECK_ARG(epi != EPI_CE || (e->ldo >= (N + 63) / 64 && e->row_index), "vlmo_gemm_nt: cross-entropy epilogue needs labels and ldo >= chunks");
    VLMO_CHECK_ARG(epi != EPI_CE_BWD || (e->resid && e->row_scale && e->row_index), "vlmo_gemm_nt: cross-entropy backward needs lse, row scale, labels");
    VLMO_CHECK_ARG(epi != EPI_BIAS_GELU || (e->out2 && e->ld2 >= N), "vlmo_gemm_nt: gelu epilogue needs out2");
    VLMO_CHECK_ARG(epi != EPI_ARGMAX || e->ldo >= (N + 63) / 64, "vlmo_gemm_nt: argmax partial buffer too narrow");
    VLMO_CHECK_ARG(epi != EPI_RESID || e->resid, "vlmo_gemm_nt: residual epilogue needs resid");
    VLMO_CHECK_ARG(epi != EPI_DGELU || (e->aux && e->ld2 >= N), "vlmo_gemm_nt: dgelu epilogue needs aux");
    return 0;
}

int run_nt(int epi, int dtype, int tile, GemmNTGroups& gp, hipStream_t stream) {
    VLMO_CHECK_ARG(dtype == VLMO_BF16 || dtype == VLMO_F16, "vlmo_gemm_nt: dtype must be bf16 or f16");
    const int tile_in = tile;
    long Mtot = 0;
    for (int q = 0; q < gp.ngroups; ++q) Mtot += gp.g[q].M;
    const int N = gp.g[0].N, K = gp.g[0].K;
    if (tile < 0) {
        // measured on MI355X (tools/gemm_bench.py): deep reductions want the 256x256 ping-pong kernel (half
        // the staged bytes per flop, MFMA pipe and LDS port busy at the same time, one workgroup/CU);
        // shallow ones (K = d) are epilogue bound and want two 128x128 workgroups per CU so that one's
        // stores overlap the other's MFMAs
        tile = (K >= 1536 && Mtot >= 2048 && N >= 512) ? 3 : 0;
        // shallow reductions whose 256x256 tiles fit ONE dispatch round (proj, dgrad_proj at N = d: 198 tiles) also do
        // better with the big tile: 1 round instead of 1.53 -> 2 rounds of 128x128 (42 vs 47 us, 30 vs 34 us)
        if (tile == 0 && K >= 512 && N >= 512 && Mtot >= 2048) {
            long t256 = 0;
            for (int q = 0; q < gp.ngroups; ++q) t256 += (long)((gp.g[q].M + 255) / 256) * ((N + 255) / 256);
            if (t256 <= 256) tile = 3;
        }
        // fewer than 128 tiles of 256x256 (the text-only pass of the four-loss objective: 2 048 rows) leave most CUs without
        // work whatever the reduction depth: 128x128 tiles, one or two per CU (tools/nt16_bench.py --M 2048, N = 768, K = 3 072:
        // 76 us with 24 tiles of 256x256, 62 with 192x256, 36.5 with 96 of 128x128)
        long t256_all = 0;
        for (int q = 0; q < gp.ngroups; ++q) t256_all += (long)((gp.g[q].M + 255) / 256) * ((N + 255) / 256);
        const bool few_tiles = t256_all <= 128;
        if (few_tiles) tile = 0;
        static const bool tile4 = !getenv("VLMO_NT_TILE4") || atoi(getenv("VLMO_NT_TILE4")) != 0;     // measurement aid
        if (tile4 && tile == 0 && dtype == VLMO_BF16 && (epi == EPI_BIAS || epi == EPI_BIAS_GELU) && K <= 1024 && N >= 2048 && Mtot >= 4096)
            tile = 4;
        // 192-row ping-pong tiles when they cut the dispatch rounds (VLMo-Large at 32 pairs: M = 8 352 = 32.6 x 256, so
        // N = 1 024 is 132 tiles of 256x256 on 256 CUs but 176 tiles of 192x256, each 3/4 of the work)
        static const int t192 = getenv("VLMO_NT_TILE192") ? atoi(getenv("VLMO_NT_TILE192")) : 2;       // measurement aid: 0 off, 1 only in place of 256x256, 2 in place of any
        if (t192 && !few_tiles && dtype == VLMO_BF16 && (epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_RESID || epi == EPI_DGELU) && K >= 1024 &&
            (tile == 3 || t192 == 2)) {
            long t256 = 0, t192n = 0;
            for (int q = 0; q < gp.ngroups; ++q) {
                t256 += (long)((gp.g[q].M + 255) / 256) * ((N + 255) / 256);
                t192n += (long)((gp.g[q].M + 191) / 192) * ((N + 255) / 256);
            }
            const long e256 = ((t256 + 255) / 256) * 256, e192 = ((t192n + 255) / 256) * 192;
            if (e192 * 10 <= e256 * 9) tile = 8;
        }
    }
    // 16x16x32 tiles of (16 * H16) x 256, H16 = 9 .. 20: the height is chosen so that the tiles fill whole dispatch rounds of
    // the 256 CUs.  Cost model (tools/nt16_bench.py at M = 2 048 ... 33 408, profiles/r04_nt16_*.txt): a launch costs
    // rounds x (H16 + 26), rounds = ceil(tiles / 256) -- a K-tile of a (16 H16) x 256 tile stages 16 (H16 + 16) rows through the
    // CU's fill path, and a tile-round carries a fixed part worth ~10 more (prologue, epilogue, launch): 144 against 192 rows
    // measured 0.92 - 0.94 x (model 0.92), 160 against 208 0.93 - 0.95 (0.92), three rounds of 272 against four of 256 0.87
    // (0.77).  The tiles picked above cost, in the same units: 256x256 rounds x 42, 192x256 rounds x 38, 256x128x32 (two per
    // CU, the epilogue of one under the K loop of the other) rounds-of-512 x 38, 128x128 rounds-of-512 x 21 (16 when every
    // tile has a CU to itself).
    static const int nt16 = getenv("VLMO_NT16") ? atoi(getenv("VLMO_NT16")) : 1;       // measurement aid: 0 = off
    static const int nt16_epis = getenv("VLMO_NT16_EPIS") ? atoi(getenv("VLMO_NT16_EPIS")) : 0xF;     // bit e: epilogue e may take these tiles
    // at EQUAL tile height the 16x16x32 kernel is 4 - 8 % faster than the 32x32x16 ones (DMA issued in the read segment;
    // tools/nt16_bench.py at M = 12 608 / 33 408, profiles/r04_nt16_bigM.txt), so a tie in the model goes to it; problems from
    // 40 output tiles of 256 x 256 up (M = 12 608 at N = 768: 42 -> 36 us).  In-session A/Bs of the full four-loss objective
    // (B = 32, merged passes): (97 %, 150 tiles) -> (102, 100) -1.0 ms, -> (106, 40) another -1.3 ms, (110, 16) no further
    // change; VLMo-Large -0.3 ms (its N = 1 024 GEMMs at 8 352 rows were below the old threshold), VLMo-Base unchanged.
    static const int nt16_tie = getenv("VLMO_NT16_TIE") ? atoi(getenv("VLMO_NT16_TIE")) : 106;        // measurement aid: percent of the old tile's cost
    static const long nt16_min = getenv("VLMO_NT16_MIN") ? atol(getenv("VLMO_NT16_MIN")) : 40;
    if (nt16 && ((nt16_epis >> epi) & 1) && (tile == 0 || tile == 3 || tile == 4 || tile == 8) && dtype == VLMO_BF16 && !gp.g[0].k1 && !gp.g[0].ckw &&
        (epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_RESID || epi == EPI_DGELU) && N >= 512 && K >= 512 &&
        Mtot * (long)N >= nt16_min * 65536l && tile_in < 0) {
        auto count = [&](int bm, int bn) {
            long t = 0;
            for (int q = 0; q < gp.ngroups; ++q) t += (long)((gp.g[q].M + bm - 1) / bm) * ((N + bn - 1) / bn);
            return t;
        };
        double cur;
        if (tile == 3) cur = (double)((count(256, 256) + 255) / 256) * 42;
        else if (tile == 8) cur = (double)((count(192, 256) + 255) / 256) * 38;
        else if (tile == 4) cur = (double)((count(256, 128) + 511) / 512) * 38;
        else cur = count(128, 128) <= 256 ? 16.0 : (double)((count(128, 128) + 511) / 512) * 21;
        int best = 0;
        double bc = 1e30;
        for (int h16 = 9; h16 <= 20; ++h16) {
            const double c = (double)((count(16 * h16, 256) + 255) / 256) * (h16 + 26);
            if (c < bc) bc = c, best = h16;
        }
        if (bc * 100 <= cur * nt16_tie) tile = 300 + best;
    }
    if (tile >= 106 && tile <= 110) tile = 300 + 2 * (tile - 100);      // (32 * (tile - 100)) rows = an even H16
    static const bool trace = getenv("VLMO_NT_TRACE") != nullptr;       // measurement aid: every distinct (epilogue, shape, tile) once
    if (trace) {
        static std::mutex mu;
        static std::vector<std::string> seen;
        char buf[256];
        int o = snprintf(buf, sizeof buf, "vlmo_gemm_nt: epi %d N %d K %d tile %d (asked %d) M", epi, N, K, tile, tile_in);
        for (int q = 0; q < gp.ngroups && o < 230; ++q) o += snprintf(buf + o, sizeof buf - o, " %d", gp.g[q].M);
        std::lock_guard<std::mutex> lk(mu);
        bool have = false;
        for (auto& t : seen) have |= (t == buf);
        if (!have) {
            seen.emplace_back(buf);
            fprintf(stderr, "%s\n", buf);
        }
    }
    if (tile >= 309 && tile <= 320) {
        // 16x16x32 MFMA, (16 * (tile - 300)) x 256 tile: bf16, plain GEMM (no convolution, no second segment)
        VLMO_CHECK_ARG(dtype == VLMO_BF16 && !gp.g[0].k1, "vlmo_gemm_nt: tiles 106..110 / 309..320 are bf16, single-source");
        ProfScope prof(80 + epi, 2.0 * Mtot * N * K, stream);
        switch (tile) {
            case 309: return launch_nt16<bf16, 9>(epi, gp, stream);
            case 310: return launch_nt16<bf16, 10>(epi, gp, stream);
            case 311: return launch_nt16<bf16, 11>(epi, gp, stream);
            case 312: return launch_nt16<bf16, 12>(epi, gp, stream);
            case 313: return launch_nt16<bf16, 13>(epi, gp, stream);
            case 314: return launch_nt16<bf16, 14>(epi, gp, stream);
            case 315: return launch_nt16<bf16, 15>(epi, gp, stream);
            case 316: return launch_nt16<bf16, 16>(epi, gp, stream);
            case 317: return launch_nt16<bf16, 17>(epi, gp, stream);
            case 318: return launch_nt16<bf16, 18>(epi, gp, stream);
            case 319: return launch_nt16<bf16, 19>(epi, gp, stream);
            default: return launch_nt16<bf16, 20>(epi, gp, stream);
        }
    }
    if (tile >= 908 && tile <= 909) {       // diagnostic: segment stamps of the 256-row kernel, schedule 0 / 1 (e.colpart = stamp buffer)
        VLMO_CHECK_ARG(epi == EPI_BIAS, "vlmo_gemm_nt: the probe build has the bias epilogue only");
        return tile == 908 ? launch_nt16<bf16, 16, 8, 1u>(epi, gp, stream) : launch_nt16<bf16, 16, 9, 1u>(epi, gp, stream);
    }
    if (tile == 208) return launch_nt16<bf16, 16, 0, 1u>(epi, gp, stream);     // measurement aid: schedule 0 (DMA issued in the MFMA segment), bias epilogue
    VLMO_CHECK_ARG(tile == 0 || tile == 3 || tile == 4 || tile == 8, "vlmo_gemm_nt: tile must be -1, 0, 3, 4, 8, 106..110 or 309..320 (got %d)", tile);
    if (tile == 4 && !(dtype == VLMO_BF16 && (epi == EPI_BIAS || epi == EPI_BIAS_GELU))) tile = 0;
    if (tile == 8 && !(dtype == VLMO_BF16 && (epi == EPI_BIAS || epi == EPI_BIAS_GELU || epi == EPI_RESID || epi == EPI_DGELU))) tile = 3;
    ProfScope prof(epi + (tile == 3 || tile == 8 ? 16 : (tile == 4 ? 48 : 0)), 2.0 * Mtot * N * K, stream);
    // tile 4 = 256x128x32, four waves, two workgroups per CU (bf16; bias and bias+GELU epilogues only): the wide shallow
    // GEMMs (qkv, fc1: K = d, N >= 3d).  1.5x the staged bytes per flop of 256x256 instead of the 2x of 128x128, still two
    // desynchronised workgroups per CU, finer tile quantisation: fc1 119 -> 113 us, qkv 80 -> 75 us.
    if (tile == 4)
        return launch_nt<bf16, 256, 128, 2, 2, false, 32, 2, false, (1u << EPI_BIAS) | (1u << EPI_BIAS_GELU)>(epi, gp, stream);
    if (tile == 8)
        return launch_nt<bf16, 192, 256, 2, 4, false, 64, 2, true,
                         (1u << EPI_BIAS) | (1u << EPI_BIAS_GELU) | (1u << EPI_RESID) | (1u << EPI_DGELU)>(epi, gp, stream);
    if (dtype == VLMO_F16) {
        if (tile == 3) return launch_nt<f16, 256, 256, 2, 4, false, 64, 2, true>(epi, gp, stream);
        return launch_nt<f16, 128, 128, 2, 2>(epi, gp, stream);
    }
    if (tile == 3) return launch_nt<bf16, 256, 256, 2, 4, false, 64, 2, true>(epi, gp, stream);
    return launch_nt<bf16, 128, 128, 2, 2>(epi, gp, stream);
}

// Row-tiles per group of the L2 tile order.  VLMO_GROUP_M fixes it (measurement aid).  Else by the bytes of the weight
// matrix: a group is group_m row panels against ALL column tiles, so group_m = 1 streams the whole weight once per row
// panel -- cheap while the weight is of the size of an XCD's L2 (4 MB), and then the activation panel is fetched once;
// larger weights want their column tiles reused across several row panels.  In-step sweeps (tools/ab_multi.sh, weight
// gradients on the main stream): VLMo-Base (weights <= 4.7 MB) group_m 1 / 2 / 3 / 4 = 14.34 / 14.36 / 14.38 / 14.43 ms;
// VLMo-Large (2 - 8.4 MB) 24.88 against 24.79 at 4; the dVAE encoder (output convolution: 67 MB) 6.03 against 5.92 ms at 4.
// (Under the side stream round 3 had measured 2 - 6 equal, 8 +0.1 ms, 16 +0.35 ms.)
int group_m_for(int N, int K) {
    static const int forced = [] {
        const char* sv = getenv("VLMO_GROUP_M");
        return sv ? atoi(sv) : 0;
    }();
    if (forced > 0) return forced;
    return (long)N * K * 2 <= 5l << 20 ? 1 : 4;
}
}  // namespace

extern "C" int vlmo_gemm_nt(int epi, int dtype, int tile, const void* A, int lda, const void* B, int ldb,
                            int M, int N, int K, const VlmoEpilogue* e, hipStream_t stream) {
    if (int rc = check_nt(epi, A, lda, B, ldb, M, N, K, e)) return rc;
    GemmNTGroups gp{};
    gp.ngroups = 1;
    gp.g[0] = GemmNT{A, B, M, N, K, lda, ldb, *e, 0, 0, 0, 0, nullptr, group_m_for(N, K), nullptr, 0, 0, 1.f};
    return run_nt(epi, dtype, tile, gp, stream);
}

extern "C" int vlmo_gemm_nt_2src(int epi, int dtype, int tile, const void* A, int lda, int k1, float seg_scale,
                                 const void* A2, int lda2, const void* B, int ldb, int M, int N, int K,
                                 const VlmoEpilogue* e, hipStream_t stream) {
    VLMO_CHECK_ARG(A2 && k1 > 0 && k1 < K && k1 % 64 == 0, "vlmo_gemm_nt_2src: need 0 < k1 < K, k1 %% 64 == 0 (k1=%d, K=%d)", k1, K);
    VLMO_CHECK_ARG(dtype == VLMO_F16, "vlmo_gemm_nt_2src: instantiated for f16 (the dVAE encoder) only");
    VLMO_CHECK_ARG(lda % 8 == 0 && lda >= k1 && lda2 % 8 == 0 && lda2 >= K - k1, "vlmo_gemm_nt_2src: bad lda/lda2 %d/%d", lda, lda2);
    if (int rc = check_nt(epi, A, K > lda ? K : lda, B, ldb, M, N, K, e)) return rc;
    GemmNTGroups gp{};
    gp.ngroups = 1;
    gp.g[0] = GemmNT{A, B, M, N, K, lda, ldb, *e, 0, 0, 0, 0, nullptr, group_m_for(N, K), A2, lda2, k1, seg_scale};
    return run_nt(epi, dtype, tile, gp, stream);
}

extern "C" int vlmo_gemm_nt_grouped(int epi, int dtype, int tile, int ngroups, const void* const* A, int lda,
                                    const void* const* B, int ldb, const int32_t* M, int N, int K,
                                    const VlmoEpilogue* e, hipStream_t stream) {
    VLMO_CHECK_ARG(ngroups >= 1 && ngroups <= MAX_GROUPS && A && B && M && e, "vlmo_gemm_nt_grouped: 1..%d groups", MAX_GROUPS);
    VLMO_CHECK_ARG(epi != EPI_ARGMAX && epi != EPI_CE, "vlmo_gemm_nt_grouped: the arg-max / cross-entropy epilogues are single-problem");
#ifdef VLMO_NO_GROUPING      // measurement aid: one launch per group
    for (int q = 0; q < ngroups; ++q)
        if (int rc = vlmo_gemm_nt(epi, dtype, tile, A[q], lda, B[q], ldb, M[q], N, K, &e[q], stream)) return rc;
    return 0;
#endif
    GemmNTGroups gp{};
    gp.ngroups = ngroups;
    for (int q = 0; q < ngroups; ++q) {
        if (int rc = check_nt(epi, A[q], lda, B[q], ldb, M[q], N, K, &e[q])) return rc;
        gp.g[q] = GemmNT{A[q], B[q], M[q], N, K, lda, ldb, e[q], 0, 0, 0, 0, nullptr, group_m_for(N, K), nullptr, 0, 0, 1.f};
    }
    return run_nt(epi, dtype, tile, gp, stream);
}

namespace {
// tile / split plan of the weight-gradient GEMM: 256x256 tiles (half the staged bytes per flop, one
// workgroup per CU) when they fill the chip in ONE dispatch round with <= 16 splits, else 128x128 tiles
// (two workgroups per CU) with the fewest splits that fill whole rounds of 512 workgroup slots.
struct TnPlan {
    int big, tiles, splits, per;
};
TnPlan tn_plan(int M, int N1, int N2, int splits_req, int force_tile) {
    const int nk = (M + 63) / 64;
    TnPlan pl{};
    const int t256 = ((N1 + 255) / 256) * ((N2 + 255) / 256);
    const int t128 = ((N1 + 127) / 128) * ((N2 + 127) / 128);
    const bool big_ok = N1 >= 256 && N2 >= 256 && nk >= 16 && t256 <= 256;
    pl.big = force_tile == 256 ? 1 : (force_tile == 128 ? 0 : (big_ok && (256 / t256) <= 16 && (256 / t256) >= 1 && nk / (256 / t256) >= 8));
    pl.tiles = pl.big ? t256 : t128;
    int splits = splits_req;
    if (splits <= 0) {
        if (pl.big) {
            splits = 256 / pl.tiles;
        } else {
            splits = 512 / pl.tiles;
            if (splits < 4) splits = 1024 / pl.tiles;
        }
        if (splits < 1) splits = 1;
    }
    if (splits > nk) splits = nk;
    pl.per = (nk + splits - 1) / splits;
    pl.splits = (nk + pl.per - 1) / pl.per;
    return pl;
}
}  // namespace

namespace {
// measurement aid: VLMO_TN_RING4=0 selects the two-buffer 64-token staging of the bf16 ping-pong weight-gradient kernels
// measurement aid: VLMO_TN_MFMA16=1 runs the batched weight-gradient launch on v_mfma_f32_16x16x32 (default: see DESIGN.md)
bool tn_mfma16() {
    static const bool v = [] {
        const char* e = getenv("VLMO_TN_MFMA16");
        return e && e[0] == '1';
    }();
    return v;
}
bool tn_ring4() {
    static const bool v = [] {
        const char* e = getenv("VLMO_TN_RING4");
        return !(e && e[0] == '0');
    }();
    return v;
}
}  // namespace

extern "C" int64_t vlmo_gemm_tn_ws_bytes(int M, int N1, int N2) {
    const TnPlan a = tn_plan(M, N1, N2, 0, 0);
    return (int64_t)a.splits * N1 * N2 * 4;
}

extern "C" int vlmo_gemm_tn(int dtype, const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                            int M, int N1, int N2, float alpha, int splits, float* ws, int64_t ws_bytes,
                            hipStream_t stream) {
    VLMO_CHECK_ARG(A && B && C, "vlmo_gemm_tn: null operand");
    VLMO_CHECK_ARG(M > 0 && N1 >= 8 && N2 >= 8, "vlmo_gemm_tn: bad problem M=%d N1=%d N2=%d", M, N1, N2);
    VLMO_CHECK_ARG(N1 % 8 == 0 && N2 % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "vlmo_gemm_tn: N1,N2,lda,ldb must be multiples of 8");
    VLMO_CHECK_ARG(lda >= N1 && ldb >= N2 && ldc >= N2, "vlmo_gemm_tn: leading dimension too small");
    VLMO_CHECK_ARG(dtype == VLMO_BF16 || dtype == VLMO_F16, "vlmo_gemm_tn: dtype must be bf16 or f16");
    int force = 0;
    const bool probe = splits >= 3000;      // diagnostic: 3000 + s = the 256x256 kernel with segment stamps -> ws (C untouched)
    if (splits >= 1000) {      // test hook: 1000 + s forces 128x128 tiles, 2000 + s forces 256x256
        force = splits >= 2000 ? 256 : 128;
        splits %= 1000;
    }
    const TnPlan pl = tn_plan(M, N1, N2, splits, force);
    // partial products go to a caller-owned slab (plain stores, then one reduction pass) when the workspace is
    // big enough and there is more than one split; else straight into C with fp32 atomics.  Measured on MI355X:
    // 7 splits of a 3072x768 gradient as atomics cost ~30 us of a 135 us launch (memory-side atomic rate).
    const bool use_slab = ws && pl.splits > 1 && N2 % 4 == 0 && ldc % 4 == 0 && ws_bytes >= (int64_t)pl.splits * N1 * N2 * 4;
    GemmTN p{A, B, C, M, N1, N2, lda, ldb, ldc, pl.per, pl.tiles, alpha, use_slab ? ws : nullptr, TN_ATOMIC};
    dim3 grid(pl.tiles * pl.splits);
    ProfScope prof(64 + (pl.big ? 8 : 0), 2.0 * M * N1 * N2, stream);
    const bool r4 = tn_ring4();
    if (probe) {
        VLMO_CHECK_ARG(pl.big && ws && ws_bytes >= (int64_t)pl.tiles * pl.splits * 8 * 64 && dtype == VLMO_BF16, "vlmo_gemm_tn: probe needs the 256x256 plan, bf16 and a stamp buffer");
        constexpr int LDS = 2 * 4 * 64 * 256;
        static DeviceOnce pattr;
        if (pattr.first()) {
            (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16, 256, 256, 2, 4, true, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16, 256, 256, 2, 4, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        }
        GemmTN pp_ = p;
        pp_.slab = ws;
        if (r4)
            hipLaunchKernelGGL((gemm_tn_kernel<bf16, 256, 256, 2, 4, true, true, true>), grid, dim3(512), LDS, stream, pp_);
        else
            hipLaunchKernelGGL((gemm_tn_kernel<bf16, 256, 256, 2, 4, true, true, false>), grid, dim3(512), LDS, stream, pp_);
        VLMO_CHECK_LAUNCH("vlmo_gemm_tn(probe)");
        return 0;
    }
    if (pl.big) {
        constexpr int LDS = 2 * 4 * 64 * 256;
        static DeviceOnce attr;
        if (attr.first()) {
            (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16, 256, 256, 2, 4, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16, 256, 256, 2, 4, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<f16, 256, 256, 2, 4, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        }
        if (dtype == VLMO_F16)
            hipLaunchKernelGGL((gemm_tn_kernel<f16, 256, 256, 2, 4, true, false, true>), grid, dim3(512), LDS, stream, p);
        else if (r4)
            hipLaunchKernelGGL((gemm_tn_kernel<bf16, 256, 256, 2, 4, true, false, true>), grid, dim3(512), LDS, stream, p);
        else
            hipLaunchKernelGGL((gemm_tn_kernel<bf16, 256, 256, 2, 4, true, false, false>), grid, dim3(512), LDS, stream, p);
    } else {
        if (dtype == VLMO_F16)
            hipLaunchKernelGGL((gemm_tn_kernel<f16, 128, 128, 2, 2>), grid, dim3(256), 65536, stream, p);
        else
            hipLaunchKernelGGL((gemm_tn_kernel<bf16, 128, 128, 2, 2>), grid, dim3(256), 65536, stream, p);
    }
    VLMO_CHECK_LAUNCH("vlmo_gemm_tn");
    if (use_slab) {
        const long total = (long)N1 * (N2 / 4);
        const int rg = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(rg), dim3(256), 0, stream, ws, pl.splits, N1, N2, C, ldc, alpha);
        VLMO_CHECK_LAUNCH("vlmo_gemm_tn(reduce)");
    }
    return 0;
}

// Weight gradients of several linears in one launch (see gemm_tn_multi_kernel).  Tiles are 256x256; when the
// problems together have fewer than ~3/4 of the CUs' worth of tiles every problem's token dimension is split
// (fp32 atomics), else each tile is owned by one workgroup and written / accumulated in place.
extern "C" int vlmo_gemm_tn_multi(int dtype, const VlmoTnProblem* probs, int n, hipStream_t stream) {
    VLMO_CHECK_ARG(probs && n >= 1, "vlmo_gemm_tn_multi: no problems");
    VLMO_CHECK_ARG(dtype == VLMO_BF16 || dtype == VLMO_F16, "vlmo_gemm_tn_multi: dtype must be bf16 or f16");
    static DeviceOnce attr;
    if (attr.first()) {
        constexpr int LDS = 2 * 4 * 64 * 256;
        (void)hipFuncSetAttribute((const void*)gemm_tn_multi_kernel<bf16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_tn_multi_kernel<bf16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_tn_multi_kernel<bf16, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)gemm_tn_multi_kernel<f16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }
    for (int q0 = 0, nq = 0; q0 < n; q0 += nq) {
        // one launch = as many of the remaining problems as fit the problem table and the placement table
        nq = 0;
        for (long tl = 0; q0 + nq < n && nq < MAX_TN_PROBS; ++nq) {
            const VlmoTnProblem& r = probs[q0 + nq];
            const long t = (long)((r.N1 + 255) / 256) * ((r.N2 + 255) / 256);
            if (nq > 0 && tl + t > MAX_TN_ORDER - 64) break;
            tl += t;
        }
        long tiles_all = 0;
        int min_nk = 1 << 30;
        double flops = 0;
        for (int q = 0; q < nq; ++q) {
            const VlmoTnProblem& r = probs[q0 + q];
            VLMO_CHECK_ARG(r.A && r.B && r.C, "vlmo_gemm_tn_multi: null operand in problem %d", q0 + q);
            VLMO_CHECK_ARG(r.M > 0 && r.N1 >= 8 && r.N2 >= 8 && r.N1 % 8 == 0 && r.N2 % 8 == 0 && r.lda % 8 == 0 &&
                               r.ldb % 8 == 0 && r.lda >= r.N1 && r.ldb >= r.N2 && r.ldc >= r.N2,
                           "vlmo_gemm_tn_multi: bad shape in problem %d (M=%d N1=%d N2=%d)", q0 + q, r.M, r.N1, r.N2);
            tiles_all += (long)((r.N1 + 255) / 256) * ((r.N2 + 255) / 256);
            const int nk = (r.M + 63) / 64;
            if (nk < min_nk) min_nk = nk;
            flops += 2.0 * r.M * r.N1 * r.N2;
        }
        int splits = 1;
        if (tiles_all < 192) {
            splits = (int)(256 / tiles_all);
            if (splits > min_nk / 8) splits = min_nk / 8;
            if (splits < 1) splits = 1;
        }
        static const int force_splits = [] {
            const char* sv = getenv("VLMO_TN_SPLITS");      // measurement aid
            return sv ? atoi(sv) : 0;
        }();
        if (force_splits > 0) splits = force_splits;
        GemmTNMulti mp{};
        mp.n = nq;
        struct Job {
            uint16_t code;
            int cost;
        };
        std::vector<Job> jobs;
        for (int q = 0; q < nq; ++q) {
            const VlmoTnProblem& r = probs[q0 + q];
            const int tiles = ((r.N1 + 255) / 256) * ((r.N2 + 255) / 256);
            const int nk = (r.M + 63) / 64;
            int sp = splits > nk ? nk : splits;
            const int per = (nk + sp - 1) / sp;
            sp = (nk + per - 1) / per;
            int mode = r.accumulate ? TN_ACCUM : TN_STORE;
            if (sp > 1) {
                mode = TN_ATOMIC;
                if (!r.accumulate) {
                    hipError_t rc = hipMemset2DAsync(r.C, (size_t)r.ldc * 4, 0, (size_t)r.N2 * 4, r.N1, stream);
                    if (rc != hipSuccess) {
                        vlmo_set_error("vlmo_gemm_tn_multi: memset failed: %s", hipGetErrorString(rc));
                        return (int)rc;
                    }
                }
            }
            VLMO_CHECK_ARG(tiles * sp <= 4096, "vlmo_gemm_tn_multi: problem %d has too many tiles", q0 + q);
            mp.p[q] = GemmTN{r.A, r.B, r.C, r.M, r.N1, r.N2, r.lda, r.ldb, r.ldc, per, tiles, r.alpha, nullptr, mode};
            for (int l = 0; l < tiles * sp; ++l) jobs.push_back(Job{(uint16_t)((q << 12) | l), per});
        }
        // placement: classes of equal reduction length, longest first; every class is cut into 8 contiguous runs (a
        // run = neighbouring tiles of one problem: they share operand panels through the XCD's L2) and XCD x takes run
        // x of every class, so all XCDs get the same mix and, inside an XCD, long tiles are dispatched before short ones
        std::stable_sort(jobs.begin(), jobs.end(), [](const Job& a, const Job& b) { return a.cost > b.cost; });
        std::vector<uint16_t> bins[8];
        for (size_t i = 0; i < jobs.size();) {
            size_t j = i;
            while (j < jobs.size() && jobs[j].cost == jobs[i].cost) ++j;
            const size_t cnt = j - i;
            for (int x = 0; x < 8; ++x)
                for (size_t k = i + cnt * x / 8; k < i + cnt * (x + 1) / 8; ++k) bins[x].push_back(jobs[k].code);
            i = j;
        }
        size_t deepest = 0;
        for (int x = 0; x < 8; ++x) deepest = bins[x].size() > deepest ? bins[x].size() : deepest;
        VLMO_CHECK_ARG(deepest * 8 <= (size_t)MAX_TN_ORDER, "vlmo_gemm_tn_multi: %zu workgroups exceed one launch (pass fewer problems per call)",
                       jobs.size());
        const int t = (int)deepest * 8;
        for (int b = 0; b < t; ++b) {
            const std::vector<uint16_t>& bin = bins[b & 7];
            mp.order[b] = (size_t)(b >> 3) < bin.size() ? bin[b >> 3] : TN_NOP;
        }
        ProfScope prof(73, flops, stream);
        constexpr int LDS = 2 * 4 * 64 * 256;
        if (dtype == VLMO_F16)
            hipLaunchKernelGGL((gemm_tn_multi_kernel<f16, true>), dim3(t), dim3(512), LDS, stream, mp);
        else if (tn_ring4() && tn_mfma16())
            hipLaunchKernelGGL((gemm_tn_multi_kernel<bf16, true, true>), dim3(t), dim3(512), LDS, stream, mp);
        else if (tn_ring4())
            hipLaunchKernelGGL((gemm_tn_multi_kernel<bf16, true>), dim3(t), dim3(512), LDS, stream, mp);
        else
            hipLaunchKernelGGL((gemm_tn_multi_kernel<bf16, false>), dim3(t), dim3(512), LDS, stream, mp);
        VLMO_CHECK_LAUNCH("vlmo_gemm_tn_multi");
    }
    return 0;
}

namespace {
// ---- 3x3 convolution with <= 64 output channels (dall_e EncoderBlock bottleneck of the first group, encoder.py:21-29:
// 112 x 112 x {256 -> 64, 64 -> 64}) ------------------------------------------------------------------------------
// With 64 output channels the implicit GEMM is bound by the per-CU fill rate of its A operand: the generic kernel
// stages the 256 input rows of a tile once per TAP (nine times per channel chunk: 40 KB per 16 MFMAs of a wave).  Here a
// K-step is (dy, 32-channel half chunk): the rows [m0 - 8, m0 + 264) of image row y + dy are staged ONCE and the three
// dx taps read them at row offsets -1 / 0 / +1 (the fragment of a lane whose pixel has no left / right neighbour is
// zeroed in registers); the three taps' weights ride along: 29 KB per 24 MFMAs of a wave, 2.1x fewer staged bytes per
// flop.  256 x 64 tile, 4 waves (64 pixels x 64 channels each), 2-deep LDS ring of 32-deep slices (59 KB: two
// workgroups per CU, as the generic kernel -- a 64-deep ring with one workgroup per CU filled at 20 GB/s per CU and
// lost on the K = 576 convolutions, whose three steps never fill the pipeline), f16.
struct Conv3Args {
    const f16* x;        // [B*H*W, Cin]
    const f16* w;        // [Cout <= 64, 9 * Cin] tap-major, channel-minor
    const f16* zero;     // >= 128 zero bytes
    const float* bias;
    f16* out;            // [B*H*W, ldo]
    int M, H, W, Cin, Cout, ldo, relu;
};

// WM x WN waves of 64 pixels x 64 channels: <4, 1> = 256 x 64 tile (<= 64 output channels), <2, 2> = 128 x 128 tile
template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv3_dx_kernel(const Conv3Args a) {
    static_assert(WM * WN == 4, "four waves");
    constexpr int BM = WM * 64, BN = WN * 64, HALO = 8, AROWS = BM + 2 * HALO;
    constexpr int A_BYTES = AROWS * 64, B_BYTES = 3 * BN * 64, STAGE = A_BYTES + B_BYTES;
    constexpr int NAI = AROWS / 16, NBI = 3 * BN / 16;      // one-KiB staging pieces per step (17 + 12 or 9 + 24)
    constexpr int NAS = (NAI + 3) / 4, NBS = NBI / 4;       // ... per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid / tiles_n) * BM, n0 = (lid % tiles_n) * BN;
    const int HW = a.H * a.W, Cin = a.Cin, K = 9 * Cin;
    const int cpt = Cin >> 5, nsteps = 3 * cpt;

    // staging sources: piece ii = i * 4 + wave covers LDS rows ii * 16 .. + 15, lane -> (row, 16-byte chunk of 4)
    const f16* a_src[NAS];
    int a_y[NAS];
#pragma unroll
    for (int i = 0; i < NAS; ++i) {
        const int rr = (i * 4 + wave) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ nt_swz<32>(rr);
        const int pix = min(max(m0 - HALO + rr, 0), a.M - 1);
        a_src[i] = a.x + (size_t)pix * Cin + c * 8;
        a_y[i] = (pix % HW) / a.W;
    }
    const f16* b_src[NBS];
#pragma unroll
    for (int i = 0; i < NBS; ++i) {
        const int rr = (i * 4 + wave) * 16 + (lane >> 2);      // tap dx * BN + output channel of the tile
        const int c = (lane & 3) ^ nt_swz<32>(rr);
        const int n = min(n0 + rr % BN, a.Cout - 1);
        b_src[i] = a.w + (size_t)n * K + (rr / BN) * Cin + c * 8;
    }
    auto stage = [&](int buf, int s_) {
        char* s = smem + buf * STAGE;
        const int dyi = s_ / cpt, hc = s_ - dyi * cpt, dy = dyi - 1;
        const int delta = dy * a.W * Cin + hc * 32;
#pragma unroll
        for (int i = 0; i < NAS; ++i) {
            if (i * 4 + wave < NAI) {
                const bool in = (unsigned)(a_y[i] + dy) < (unsigned)a.H;
                glds16(in ? a_src[i] + delta : a.zero, s + (i * 4 + wave) * 1024);
            }
        }
        const int wofs = dyi * 3 * Cin + hc * 32;
#pragma unroll
        for (int i = 0; i < NBS; ++i) glds16(b_src[i] + wofs, s + A_BYTES + (i * 4 + wave) * 1024);
    };

    const int l31 = lane & 31, h = lane >> 5;
    // A fragment of tap dx, row block i: LDS row HALO + wm * 64 + i * 32 + l31 + dx (the swizzle key of a row does not
    // change with + 32); B fragment: row dxi * BN + wn * 64 + j * 32 + l31
    int a_off[3], a_swz[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int row = HALO + wm * 64 + l31 + (t - 1);
        a_off[t] = row * 64;
        a_swz[t] = nt_swz<32>(row);
    }
    const int b_off = A_BYTES + (wn * 64 + l31) * 64, b_swz = nt_swz<32>(l31);
    bool edge_l[2], edge_r[2];      // the lane's pixel has no left / right neighbour in its image row
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int xx = (m0 + wm * 64 + i * 32 + l31) % a.W;
        edge_l[i] = xx == 0;
        edge_r[i] = xx == a.W - 1;
    }
    const bool relu_in = (a.relu & 2) != 0;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    stage(0, 0);
    for (int s_ = 0; s_ < nsteps; ++s_) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s_ + 1 < nsteps) stage((s_ + 1) & 1, s_ + 1);
        const char* s = smem + (s_ & 1) * STAGE;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 af[2], bf[2];
                const int ca = ((2 * ks + h) ^ a_swz[t]) << 4, cb = ((2 * ks + h) ^ b_swz) << 4;
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i] = *(const f16x8*)(s + a_off[t] + i * 2048 + ca);
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = *(const f16x8*)(s + b_off + (t * BN + j * 32) * 64 + cb);
                const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (relu_in) af[i] = __builtin_elementwise_max(af[i], z);
                    if (t == 0) af[i] = edge_l[i] ? z : af[i];
                    if (t == 2) af[i] = edge_r[i] ? z : af[i];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = Elem<f16>::mfma(af[i], bf[j], acc[i][j]);
            }
        }
    }
    // epilogue: bias (+ ReLU) -> f16, through a wave-private [64 pixels][64 channels] LDS image so that every global store
    // is a 16-byte piece of a 128-byte run of an output row
    __syncthreads();
    f16* ep = (f16*)(smem + wave * 8192);
    const bool relu_out = (a.relu & 1) != 0;
    const int nw0 = n0 + wn * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nw0 + j * 32 + l31;
        const float bv = (a.bias && n < a.Cout) ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + bv;
                if (relu_out) v = fmaxf(v, 0.f);
                ep[(i * 32 + 8 * (r >> 2) + 4 * h + (r & 3)) * 64 + j * 32 + l31] = (f16)v;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ch = lane + 64 * q, row = ch >> 3, c8 = (ch & 7) * 8;
        const int m = m0 + wm * 64 + row;
        if (m < a.M && nw0 + c8 < a.Cout) {
            const f16x8 v = *(const f16x8*)(ep + row * 64 + c8);
            __builtin_nontemporal_store(v, (f16x8*)(a.out + (size_t)m * a.ldo + nw0 + c8));
        }
    }
}

template <int WM, int WN>
int launch_conv3_dx(const void* x, int B, int H, int W, int Cin, const void* w, int Cout, const void* zero_page,
                    const VlmoEpilogue* e, hipStream_t stream) {
    Conv3Args a{(const f16*)x, (const f16*)w, (const f16*)zero_page, e->bias, (f16*)e->out, B * H * W, H, W, Cin, Cout,
                e->ldo, e->relu};
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int LDS = 2 * ((BM + 16) * 64 + 3 * BN * 64);
    static DeviceOnce once;
    if (once.first())
        (void)hipFuncSetAttribute((const void*)conv3_dx_kernel<WM, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const int grid = ((a.M + BM - 1) / BM) * ((Cout + BN - 1) / BN);
    hipLaunchKernelGGL((conv3_dx_kernel<WM, WN>), dim3(grid), dim3(256), LDS, stream, a);
    VLMO_CHECK_LAUNCH("vlmo_conv2d_nhwc");
    return 0;
}
}  // namespace

// 2-D convolution, stride 1, "same" zero padding (kw-1)/2, over an NHWC activation matrix
// x [B*H*W, Cin] with weights w [Cout, kw*kw*Cin] (tap-major, channel-minor): dall_e/utils.py:37-48.
extern "C" int vlmo_conv2d_nhwc(int epi, int dtype, const void* x, int B, int H, int W, int Cin, int kw,
                                const void* w, int Cout, const void* zero_page, const VlmoEpilogue* e,
                                hipStream_t stream) {
    VLMO_CHECK_ARG(x && w && e && zero_page, "vlmo_conv2d_nhwc: null pointer");
    VLMO_CHECK_ARG(B > 0 && H > 0 && W > 0 && H < 32768 && W < 32768, "vlmo_conv2d_nhwc: bad geometry");
    VLMO_CHECK_ARG(Cin % 64 == 0 && Cout % 4 == 0, "vlmo_conv2d_nhwc: Cin must be a multiple of 64 (got %d), Cout of 4", Cin);
    VLMO_CHECK_ARG(kw >= 1 && kw % 2 == 1, "vlmo_conv2d_nhwc: kernel width must be odd (dall_e/utils.py:14)");
    VLMO_CHECK_ARG(e->out && e->ldo >= Cout, "vlmo_conv2d_nhwc: bad output");
    VLMO_CHECK_ARG(dtype == VLMO_BF16 || dtype == VLMO_F16, "vlmo_conv2d_nhwc: dtype must be bf16 or f16");
    const int K = kw * kw * Cin;
    GemmNTGroups p{};
    p.ngroups = 1;
    p.g[0] = GemmNT{x, w, B * H * W, Cout, K, Cin, K, *e, H, W, Cin, kw, zero_page, 8, nullptr, 0, 0, 1.f};
    ProfScope prof(32 + epi, 2.0 * B * H * W * Cout * K, stream);
    static const bool shared_dx = [] {
        const char* v = getenv("VLMO_CONV3_DX");        // A/B: 0 = the generic per-tap kernels
        return !(v && v[0] == '0');
    }();
    // wide bottlenecks whose 256 x 256 tiles fill most of one dispatch round (group 3 of the dVAE at 64 images: 196 tiles):
    // the ping-pong kernel with per-tap staging -- half the staged bytes per flop of the 128 x 128 tile
    static const bool pp_conv = [] {
        const char* v = getenv("VLMO_CONV_PP");
        return !(v && v[0] == '0');
    }();
    if (dtype == VLMO_F16 && epi == EPI_BIAS && pp_conv && Cout % 256 == 0) {
        const long t256 = (long)((B * H * W + 255) / 256) * (Cout / 256);
        if (t256 >= 160 && (t256 <= 256 || t256 >= 640))
            return launch_nt<f16, 256, 256, 2, 4, true, 64, 2, true, (1u << EPI_BIAS)>(epi, p, stream);
    }
    if (dtype == VLMO_F16 && epi == EPI_BIAS && kw == 3 && shared_dx && Cin % 32 == 0 && Cout % 8 == 0 && e->ldo % 8 == 0) {
        if (Cout <= 64) return launch_conv3_dx<4, 1>(x, B, H, W, Cin, w, Cout, zero_page, e, stream);
        return launch_conv3_dx<2, 2>(x, B, H, W, Cin, w, Cout, zero_page, e, stream);
    }
    // <= 64 output channels: a 256 x 64 tile -- with the 128-wide tile half of every MFMA and half of the weight staging
    // multiplied padding
    if (dtype == VLMO_F16 && Cout <= 64 && epi == EPI_BIAS)
        return launch_nt<f16, 256, 64, 4, 1, true, 64, 2, false, (1u << EPI_BIAS)>(epi, p, stream);
    if (dtype == VLMO_F16) return launch_nt<f16, 128, 128, 2, 2, true>(epi, p, stream);
    return launch_nt<bf16, 128, 128, 2, 2, true>(epi, p, stream);
}
