// Shared device/host helpers for the gfx950 (MI355X, CDNA4) VLMo kernels.
// gfx950 only: 64-wide wavefronts, MFMA 32x32x16, LDS-DMA (global_load_lds 16 B),
// ds_read_b64_tr_b16.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- error plumbing (C-ABI: 0 ok, <0 argument error, >0 hipError_t) -------
void vlmo_set_error(const char* fmt, ...);
#define VLMO_CHECK_ARG(cond, ...)                         \
    do {                                                  \
        if (!(cond)) {                                    \
            vlmo_set_error(__VA_ARGS__);                  \
            return -1;                                    \
        }                                                 \
    } while (0)
#define VLMO_CHECK_LAUNCH(name)                                             \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            vlmo_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                \
        }                                                                   \
    } while (0)

// One-time setup per DEVICE (hipFuncSetAttribute for a kernel's dynamic-LDS limit): a process may drive several GPUs,
// and a function attribute set on one device is not set on another.
struct DeviceOnce {
    uint64_t done = 0;      // bit d: already done on device d (< 64 devices per process)
    bool first() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = 1ull << (dev & 63);
        const uint64_t old = __atomic_fetch_or(&done, bit, __ATOMIC_RELAXED);
        return (old & bit) == 0;
    }
};

// ---- column reductions -------------------------------------------------------
// Column sums over the token dimension (bias / gamma / LayerNorm-weight gradients)
// are done in two stages: each workgroup writes its partial row to a caller-owned
// workspace [nblk, ncols], then reduce_partials() folds the partials into the
// outputs with ~8 atomics per address.  (One atomic per column per workgroup straight
// into the output serialises ~1000 adders on each address: measured 7x slower.)
#define VLMO_MAX_PARTIAL_BLOCKS 512
int reduce_partials(const float* ws, int nblk, int ncols, float* out0, int n0, float* out1, hipStream_t stream,
                    float* out2 = nullptr, float* out3 = nullptr);
// vlmo_block_bwd runs these tiny folds on its side stream (they only produce parameter gradients; on the
// caller's stream each one is a launch + dependency bubble on the critical path): while `vlmo_defer_reduce`
// points at a record, the NEXT reduce_partials() of this thread fills it instead of launching.
struct PartialReduce {
    const float* ws = nullptr;
    int nblk = 0, ncols = 0;
    float* out0 = nullptr;
    int n0 = 0;
    float* out1 = nullptr;
    float* out2 = nullptr;
    float* out3 = nullptr;
};
extern thread_local PartialReduce* vlmo_defer_reduce;
inline int reduce_partials(const PartialReduce& r, hipStream_t stream) {
    return r.ws ? reduce_partials(r.ws, r.nblk, r.ncols, r.out0, r.n0, r.out1, stream, r.out2, r.out3) : 0;
}
// layernorm.hip: LayerNorm backward fused with the residual-branch backward of the block below (see there)
int ln_resid_seg_bwd(const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
                     const float* dres, float* dx, float* dw, float* db, const void* zd, const float* gamma,
                     const float* row_scale, const int32_t* row_index, void* dz, float* dgamma, uint32_t drop_thresh,
                     float inv_keep, uint64_t seed0, uint64_t seed1, int seg, int M, int d, float* ws, int64_t ws_bytes,
                     int* nb0, int* nblk, hipStream_t stream);
inline int64_t reduce_ws_need(int ncols) { return (int64_t)VLMO_MAX_PARTIAL_BLOCKS * ncols * 4; }

// ---- element traits: the transformer runs bf16, the dVAE runs fp16 --------
template <typename T> struct Elem;
template <> struct Elem<bf16> {
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Elem<f16> {
    typedef f16x8 v8;
    typedef f16x4 v4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

// ---- LDS-DMA: 16 B per lane, LDS destination = wave-uniform base + lane*16 -
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

// buffer form of the LDS-DMA: buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds -- a wave-uniform descriptor, a wave-uniform
// scalar byte offset (the K-tile advance) and ONE 32-bit per-lane byte offset; no vector instruction precedes the load.
// (The descriptor type exists in the device pass only; the host pass needs the kernels' signatures, not their bodies.)
struct BufSrc {
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t r;
#endif
    __device__ __forceinline__ void init(const void* base) {
#if defined(__HIP_DEVICE_COMPILE__)
        r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFFF, 0x00020000);
#endif
    }
    __device__ __forceinline__ void load16(void* lds_wave_base, uint32_t lane_off, int scalar_off) const {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds_wave_base), 16, lane_off, scalar_off, 0, 0);
#endif
    }
};

// transposed LDS read: 4x16 block of 16-bit elements, column-major into lanes
template <typename T>
__device__ __forceinline__ typename Elem<T>::v4 lds_tr4(const void* p);
template <>
__device__ __forceinline__ bf16x4 lds_tr4<bf16>(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4*)(p));
}
template <>
__device__ __forceinline__ f16x4 lds_tr4<f16>(const void* p) {
    typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 hv4;
    const hv4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hv4*)(p));
    return __builtin_bit_cast(f16x4, r);
}

// ---- counter-based RNG for dropout -----------------------------------------------------------
// keep(e) <=> u16(e) >= thresh, thresh = round(p * 65536).  Forward and backward regenerate the
// same mask from (seed, element-group index): group g covers 4 consecutive elements, elements
// 2h and 2h+1 of it take the low / high 16 bits of hash32(seed, 2g + h).  One 32-bit hash
// (2 multiplies, 3 xor-shifts) serves two elements; a kernel whose lanes hold single elements of a
// group (attention backward, keys on lanes) evaluates only the half it needs.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t drop_half(uint64_t seed, uint64_t group_idx, int half) {
    const uint32_t i = (uint32_t)(group_idx * 2 + half);
    return hash32((i ^ (uint32_t)seed) * 0x9E3779B9u + (uint32_t)(seed >> 32));
}
__device__ __forceinline__ uint64_t drop_bits4(uint64_t seed, uint64_t group_idx) {
    return (uint64_t)drop_half(seed, group_idx, 0) | ((uint64_t)drop_half(seed, group_idx, 1) << 32);
}
__device__ __forceinline__ bool drop_keep(uint64_t bits, int j, uint32_t thresh) {
    return ((uint32_t)(bits >> (16 * j)) & 0xFFFFu) >= thresh;
}
// element j (0..3) of a group without computing the other half
__device__ __forceinline__ bool drop_keep1(uint64_t seed, uint64_t group_idx, int j, uint32_t thresh) {
    return ((drop_half(seed, group_idx, j >> 1) >> (16 * (j & 1))) & 0xFFFFu) >= thresh;
}

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16/f16 output
// resolution): one v_exp + one v_rcp + 5 FMAs instead of libm erff's ~30 VALU
// instructions -- the GELU epilogues run once per GEMM output element.
// e = exp(-x^2/2) is shared between the erf tail and the Gaussian pdf.
__device__ __forceinline__ float norm_cdf_from(float x, float e) {
    // v_rcp_f32 (1 ulp) rather than an IEEE divide: the latter expands to ~11 VALU ops
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, fabsf(x), 1.0f));
    float poly = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    poly = fmaf(poly, t, 0.5f * 1.421413741f);
    poly = fmaf(poly, t, 0.5f * -0.284496736f);
    poly = fmaf(poly, t, 0.5f * 0.254829592f);
    const float tail = poly * t * e;                 // 0.5 * erfc(|x|/sqrt2)
    return x >= 0.f ? 1.0f - tail : tail;
}
// exp(-x^2/2) as one mul + one raw v_exp_f32 (results below 2^-126 flush to 0)
__device__ __forceinline__ float gauss_from(float x) {
    return __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);
}
__device__ __forceinline__ float gelu_erf(float x) {
    const float e = gauss_from(x);
    return x * norm_cdf_from(x, e);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float e = gauss_from(x);
    return norm_cdf_from(x, e) + x * 0.39894228040143268f * e;
}

// v of the lane CTRL selects, 0 where that lane does not exist or the row / bank mask excludes the destination
template <int CTRL, int ROW_MASK, int BANK_MASK> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, false));
}
// Sum over the 64 lanes, the same value in every lane.  All lanes must be active.  A DPP scan (gfx9 row_shr / row_bcast:
// seven dependent vector adds, ~60 cycles) + one v_readlane; the butterfly through __shfl_xor is six ds_bpermute round
// trips (~700 cycles), which is what a LayerNorm row waited for twice when the kernel has one or two waves per SIMD.
__device__ __forceinline__ float wave_sum(float v) {
    float s = v + dpp_mov<0x111, 0xf, 0xf>(v);          // row_shr:1
    s += dpp_mov<0x112, 0xf, 0xf>(v);                   // row_shr:2
    s += dpp_mov<0x113, 0xf, 0xf>(v);                   // row_shr:3  -> sums of 4
    s += dpp_mov<0x114, 0xf, 0xe>(s);                   // row_shr:4, banks 1-3 -> sums of 8
    s += dpp_mov<0x118, 0xf, 0xc>(s);                   // row_shr:8, banks 2-3 -> lane 15 of a row = the row's total
    s += dpp_mov<0x142, 0xa, 0xf>(s);                   // row_bcast15 into rows 1 and 3
    s += dpp_mov<0x143, 0xc, 0xf>(s);                   // row_bcast31 into rows 2 and 3 -> lane 63 = the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 63));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
