// Fused multi-head softmax attention for VLMo on gfx950 (head_dim 64).
// Reference: Attention.forward, vlmo.py:79-95:
//   attn = softmax((q k^T) * dh^-0.5 + keymask(-inf)) ; dropout ; ctx = attn v
// VLMo sequences are short (64 / 197 / 261 tokens), so a whole head's K and V
// sit in LDS (<= 36 KB each) and the scores of a 32-query tile never leave
// registers: no N x N tensor in HBM (the reference materialises [B,h,N,N]
// several times).  Forward: attn_fwd1_kernel (one wave per query tile, key
// tiles one at a time with a lazily rescaled running maximum; sequences of up
// to 288 tokens) and attn_fwd_kernel (128-key chunks with an exact running
// maximum; longer sequences).  Backward: attn_bwd1_kernel (single pass, up to
// 256 tokens) and attn_bwd_kernel (two phases, up to 288).
//
// Orientation: scores are computed TRANSPOSED, S^T[key][query] = K . Q^T, so a
// lane owns one query column (softmax reductions are in-register + one
// cross-half shuffle) and the fp32 accumulator tile is directly the B operand
// of the next MFMA (O^T = V^T . P^T) after a bf16 pack; V^T fragments come from
// ds_read_b64_tr_b16.  All four operand images use one dual-use swizzle that
// is bank-conflict free for row reads and transposed reads (tests/ldssim.py).
//
// Packed rows: sequence s = rows [rowA,rowA+lenA) ++ [rowB,rowB+lenB) of the
// [M, 3d] qkv matrix, so text+image fusion needs no concatenated copy.
#include "common.h"
#include "vlmo_hip.h"
#include <stdlib.h>
#include <string.h>

namespace {

struct AttnArgs {
    const bf16* qkv;
    const bf16* ctx;      // bwd: forward output
    const bf16* dctx;     // bwd: grad of ctx
    bf16* out;            // fwd: ctx ; bwd: dqkv
    float* lse;
    float* qvsum;        // bwd, optional: [num_seq][2 d] column sums of dq | dv over each sequence's tokens
    const int32_t* seg;
    const int32_t* keymask;
    int lse_stride, heads, d;
    float scale, scale_log2e;
    uint32_t drop_thresh, drop_cmp;     // drop_cmp = thresh << 16: keep <=> att_mix(...) >= drop_cmp
    float inv_keep;
    uint64_t seed;
    int bh0;             // dropout-mask index of the launch's first (sequence, head): a backward launch over a SUFFIX of the
                         // forward launch's sequences regenerates the forward's mask (vlmo.py:93 has one mask per call)
    int warm;            // bwd: every workgroup touches the operand rows of the (sequence, head) `warm` workgroups further on (the
                         // one its XCD sees a dispatch round later), one 4-byte load per 128-byte row piece; 0 = off
    int split_tok;       // bwd, 0 = off: sequences longer than this many tokens (a multiple of 32) are differentiated by TWO
                         // launches -- attn_bwd_kernel takes every tile pair that touches the last (fringe) tile and writes
                         // partial dq / dk / dv rows, attn_bwd1_kernel then takes the pairs among the first split_tok tokens
                         // and adds those partials (see vlmo_attn_bwd)
};

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

__device__ __forceinline__ int att_off(int row, int ch) {
    return 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

// stage rows [0, NPAD) x 64 columns starting at column `col0` of qkv-like matrix
// into a dual-use image by LDS-DMA (one wave-instruction = 8 rows = 1 KiB)
template <int NWAVES>
__device__ __forceinline__ void stage_image(const bf16* base, int ld, int col0, const int* rowidx, char* img, int ninstr,
                                            int wave, int lane) {
    const int chhi = lane >> 5, rowlo = (lane >> 2) & 7, pc = lane & 3;
    for (int ii = wave; ii < ninstr; ii += NWAVES) {
        const int row = ii * 8 + rowlo;
        const int ch = chhi * 4 + (pc ^ ((row >> 2) & 3));
        glds16(base + (size_t)rowidx[row] * ld + col0 + ch * 8, img + ii * 1024);
    }
}

// ---- attention dropout (vlmo.py:93): counter-based, regenerated in the backward -------------------------------
// keep(seq*heads + head, q, key) <=> top 16 bits of att_mix(c * G + att_key) >= thresh, c = q * 512 + key (sequences
// are < 512 tokens).  The soft-max arithmetic of these kernels is bound by VALU ISSUE (one wave alone on a SIMD issues
// a vector instruction every 4 cycles; the single-pass backward spends ~2/3 of a step in it), so the hash is as short
// as its use allows: the counter is affine in q and in key (either orientation advances it with ONE add of a
// compile-time constant), one xor-shift + one multiply mix it, and only the TOP half of the product is used -- the
// best-mixed bits, compared as a whole word against thresh << 16 (no field extraction).  5 instructions per element.
#define ATT_G 0x9E3779B1u
#define ATT_G512 ((uint32_t)(512ull * ATT_G))
__device__ __forceinline__ uint32_t att_key(uint64_t seed, int bh) {
    return hash32((uint32_t)seed ^ ((uint32_t)bh * 0x9E3779B9u)) + (uint32_t)(seed >> 32);
}
__device__ __forceinline__ uint32_t att_mix(uint32_t x) {
    x ^= x >> 15;
    x *= 0x7feb352du;
    return x;
}

// B/A operand by row read: rows row_base + (lane&31), features 16*ks + 8*(lane>>5) ..+7
__device__ __forceinline__ bf16x8 row_frag(const char* img, int row_base, int ks, int lane) {
    return *(const bf16x8*)(img + att_off(row_base + (lane & 31), 2 * ks + (lane >> 5)));
}
// A operand by transposed read: operand rows = features cb + (lane&31), k = image rows in
// accumulator order rb + 8*(j>>2) + 4*(lane>>5) + (j&3)
__device__ __forceinline__ bf16x8 tr_frag(const char* img, int rb, int cb, int lane) {
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, h = lane >> 5;
    const int col = cb + 16 * (g & 1) + 4 * pp;
    const int r0 = rb + 4 * h + q;
    const bf16x4 lo = lds_tr4<bf16>(img + att_off(r0, col >> 3) + (col & 7) * 2);
    const bf16x4 hi = lds_tr4<bf16>(img + att_off(r0 + 8, col >> 3) + (col & 7) * 2);
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

__device__ __forceinline__ void setup_rows(const int32_t* seg, int sidx, const int32_t* keymask, int npad, int* rowidx,
                                           float* kbias, int& N) {
    const int rowA = seg[4 * sidx + 0], lenA = seg[4 * sidx + 1], rowB = seg[4 * sidx + 2], lenB = seg[4 * sidx + 3];
    N = lenA + lenB;
    for (int i = threadIdx.x; i < npad; i += blockDim.x) {
        const int tok = min(i, N - 1);
        const int row = tok < lenA ? rowA + tok : rowB + (tok - lenA);
        rowidx[i] = row;
        const bool ok = (i < N) && (!keymask || keymask[row] != 0);
        kbias[i] = ok ? 0.f : -INFINITY;
    }
}

// Keys are swept in chunks of CK*32 = 128 with an exact running max (the chunk's
// scores live in 64 accumulator registers; O is rescaled at most once per chunk).
#define ATT_CK 4
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const AttnArgs a, const int NPAD) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kimg = smem;
    char* Vimg = smem + NPAD * 128;
    float* kbias = (float*)(smem + 2 * NPAD * 128);
    int* rowidx = (int*)(kbias + NPAD);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.x, sidx = bh / a.heads, hd = bh % a.heads;
    const int ld = 3 * a.d;
    int N;
    setup_rows(a.seg, sidx, a.keymask, NPAD, rowidx, kbias, N);
    __syncthreads();
    const int nq = (N + 31) >> 5;
    stage_image<4>(a.qkv, ld, a.d + hd * 64, rowidx, Kimg, nq * 4, wave, lane);
    stage_image<4>(a.qkv, ld, 2 * a.d + hd * 64, rowidx, Vimg, nq * 4, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int l31 = lane & 31, h = lane >> 5;
    const uint32_t akey = att_key(a.seed, bh + a.bh0);
    for (int qt = wave; qt < nq; qt += 4) {
        const int qi = qt * 32 + l31;
        const int qrow = rowidx[qi];
        const uint32_t rq = ((uint32_t)qi * 512u + 4u * h) * ATT_G + akey;     // dropout counter base of this lane
        const bf16* qp = a.qkv + (size_t)qrow * ld + hd * 64 + 8 * h;
        bf16x8 qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(qp + 16 * s);

        float m_run = -INFINITY, l_run = 0.f;
        f32x16 O[2] = {zero16(), zero16()};
        for (int c0 = 0; c0 < nq; c0 += ATT_CK) {
            f32x16 S[ATT_CK];
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < ATT_CK; ++c) {
                S[c] = zero16();
                const int kt = c0 + c;
                if (kt < nq) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) S[c] = Elem<bf16>::mfma(row_frag(Kimg, kt * 32, s, lane), qf[s], S[c]);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 kb = *(const f32x4*)(kbias + kt * 32 + 8 * g4 + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = S[c][4 * g4 + e] * a.scale_log2e + kb[e];
                            S[c][4 * g4 + e] = t;
                            mx = fmaxf(mx, t);
                        }
                    }
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const bool dead = (m_new == -INFINITY);            // every key so far is masked
            const float alpha = dead ? 1.f : __builtin_amdgcn_exp2f(m_run - m_new);
            float lsum = 0.f;
#pragma unroll
            for (int c = 0; c < ATT_CK; ++c) {
                if (c0 + c < nq) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float p = dead ? 0.f : __builtin_amdgcn_exp2f(S[c][i] - m_new);
                        S[c][i] = p;
                        lsum += p;
                    }
                }
            }
            lsum += __shfl_xor(lsum, 32, 64);
            l_run = l_run * alpha + lsum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
#pragma unroll
            for (int c = 0; c < ATT_CK; ++c) {
                const int kt = c0 + c;
                if (kt < nq) {
                    if (a.drop_thresh) {
                        const uint32_t rk = rq + (uint32_t)(kt * 32) * ATT_G;
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (att_mix(rk + (uint32_t)(8 * g4 + e) * ATT_G) < a.drop_cmp) S[c][4 * g4 + e] = 0.f;
                    }
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        bf16x8 pf;
#pragma unroll
                        for (int j = 0; j < 8; ++j) pf[j] = (bf16)S[c][8 * s2 + j];
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt)
                            O[dt] = Elem<bf16>::mfma(tr_frag(Vimg, kt * 32 + 16 * s2, dt * 32, lane), pf, O[dt]);
                    }
                }
            }
        }
        if (qi < N) {
            const float inv = a.inv_keep / l_run;
            bf16* op = a.out + (size_t)qrow * a.d + hd * 64 + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    bf16x4 o = {(bf16)(O[dt][4 * g4 + 0] * inv), (bf16)(O[dt][4 * g4 + 1] * inv),
                                (bf16)(O[dt][4 * g4 + 2] * inv), (bf16)(O[dt][4 * g4 + 3] * inv)};
                    *(bf16x4*)(op + dt * 32 + 8 * g4) = o;
                }
            if (h == 0 && a.lse) a.lse[(size_t)bh * a.lse_stride + qi] = (m_run + log2f(l_run)) * LN2;
        }
    }
}

// ---- forward, one wave per query tile (sequences of <= 9 tiles = 288 tokens) ------------------------------------
// The chunked kernel above gives a wave several query tiles in turn (7 tiles over 4 waves at 197 tokens: the last round
// runs one wave short, each tile starts with an exposed global load of its Q rows) and carries 64 score registers per
// chunk, i.e. two waves per SIMD.  Here a workgroup has one wave per query tile (right-sized: 7 waves at 197 tokens,
// 2 at 64; waves past the sequence's last tile of a mixed launch leave after the staging barrier), the Q fragments are
// loaded from global memory before the K / V images are staged, and the key tiles are walked one at a time:
//   * the accumulator of the S^T MFMAs starts at the key bias (0 / -inf), so masking costs no vector instruction;
//   * the running maximum is lazy: the accumulators are rescaled only when some row's maximum grew by more than 2^8
//     since the last rescale (a wave-uniform branch, taken on the first tile and almost never again) -- exp2 arguments
//     stay <= 8, the final division by the row sum cancels the stale reference exactly as in the eager form;
//   * row sums are per half-wave partials, combined once at the end;
//   * the S^T product of tile t + 1 is issued before the soft-max arithmetic of tile t, whose P^T V products run under the
//     next tile's arithmetic: the matrix pipe works while the wave issues vector instructions.
// ~110 registers: four waves per SIMD, two workgroups per CU at 197 tokens.
__global__ __launch_bounds__(576) void attn_fwd1_kernel(const AttnArgs a, const int NPAD) {
    const int IMG = NPAD * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kimg = smem;
    char* Vimg = smem + IMG;
    float* kbias = (float*)(smem + 2 * IMG);

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int bh = blockIdx.x, sidx = bh / a.heads, hd = bh % a.heads;
    const int ld = 3 * a.d;
    const int rowA = a.seg[4 * sidx + 0], lenA = a.seg[4 * sidx + 1], rowB = a.seg[4 * sidx + 2], lenB = a.seg[4 * sidx + 3];
    const int N = lenA + lenB;
    auto rowof = [&](int tok) {
        tok = min(tok, N - 1);
        return tok < lenA ? rowA + tok : rowB + (tok - lenA);
    };
    const int nq = (N + 31) >> 5;
    const int l31 = lane & 31, h = lane >> 5;
    const bool active = w < nq;
    const int qi = w * 32 + l31;
    const int qrow = rowof(qi);
    bf16x8 qf[4];
    if (active) {
        const bf16* qp = a.qkv + (size_t)qrow * ld + hd * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(qp + 16 * s);
    }
    for (int ii = w; ii < nq * 4; ii += nw) {
        const int chhi = lane >> 5, rowlo = (lane >> 2) & 7, pc = lane & 3;
        const int row = ii * 8 + rowlo;
        const int ch = chhi * 4 + (pc ^ ((row >> 2) & 3));
        const size_t r = (size_t)rowof(row);
        glds16(a.qkv + r * ld + a.d + hd * 64 + ch * 8, Kimg + ii * 1024);
        glds16(a.qkv + r * ld + 2 * a.d + hd * 64 + ch * 8, Vimg + ii * 1024);
    }
    for (int i = threadIdx.x; i < NPAD; i += blockDim.x) {
        const bool ok = (i < N) && (!a.keymask || a.keymask[rowof(i)] != 0);
        kbias[i] = ok ? 0.f : -INFINITY;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!active) return;

    // LDS addressing as in attn_bwd1_kernel: per-lane offsets once, tile bases as scalars / immediates
    const int xq = (l31 >> 2) & 3;
    const int rf0 = 1024 * (l31 >> 3) + 64 * (l31 & 7) + 16 * (h ^ xq);
    const int rf1 = 1024 * (l31 >> 3) + 64 * (l31 & 7) + 16 * ((2 + h) ^ xq);
    const int tg = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;
    const int trl = 64 * (4 * h + tq) + 16 * ((2 * tg + (tp >> 1)) ^ h) + 8 * (tp & 1);
    const int trh = 1024 + 64 * (4 * h + tq) + 16 * ((2 * tg + (tp >> 1)) ^ ((2 + h) & 3)) + 8 * (tp & 1);
    auto rfrag = [&](const char* img_rb, int s) -> bf16x8 {
        return *(const bf16x8*)(img_rb + 512 * (s >> 1) + ((s & 1) ? rf1 : rf0));
    };
    auto tfrag = [&](const char* img_rb, int dt) -> bf16x8 {
        const bf16x4 lo = lds_tr4<bf16>(img_rb + 512 * dt + trl);
        const bf16x4 hi = lds_tr4<bf16>(img_rb + 512 * dt + trh);
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    auto other_half = [&](float v) -> float {       // the value lane ^ 32 holds (v_permlane32_swap: no LDS round trip)
        const uint32_t u = __builtin_bit_cast(uint32_t, v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        return __builtin_bit_cast(float, h ? r[0] : r[1]);
    };
    // S^T tile of key tile kt: accumulator = key bias of the rows this lane holds (8 (i >> 2) + 4 h + (i & 3))
    auto s_tile = [&](int kt) -> f32x16 {
        f32x16 acc;
        const float* kb = kbias + kt * 32 + 4 * h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v = *(const f32x4*)(kb + 8 * g4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * g4 + e] = v[e];
        }
        const char* kimg = Kimg + 4096 * kt;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = Elem<bf16>::mfma(rfrag(kimg, s), qf[s], acc);
        return acc;
    };

    const uint32_t akey = att_key(a.seed, bh + a.bh0);
    const uint32_t rq = ((uint32_t)qi * 512u + 4u * h) * ATT_G + akey;
    const float c2 = a.scale_log2e;
    const float thr = 8.f / c2;             // lazy-rescale threshold in raw score units
    float m_run = -INFINITY;                // reference maximum in raw score units
    f32x2 l2 = {0.f, 0.f};                  // this half-wave's partial row sum, two interleaved accumulators (v_pk_add_f32)
    f32x16 O[2] = {zero16(), zero16()};
    // soft-max + dropout of one S^T tile in place, then its P^T V products
    auto consume = [&](f32x16& S, int kt) {
        float mx = fmaxf(fmaxf(S[0], S[1]), S[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, S[i]), S[i + 1]);
        mx = fmaxf(mx, S[15]);
        mx = fmaxf(mx, other_half(mx));
        if (__builtin_amdgcn_ballot_w64(mx - m_run > thr)) {
            asm volatile("" ::: "memory");      // a real branch: hipcc otherwise if-converts it into 17 packed multiplies per tile
            const float m_new = fmaxf(m_run, mx);
            const float alpha = (m_run == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f((m_run - m_new) * c2);
            l2 *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int i = 0; i < 16; ++i) O[dt][i] *= alpha;
            m_run = m_new;
        }
        const float msub = (m_run == -INFINITY) ? 0.f : -m_run * c2;
        const f32x2 c2v = {c2, c2}, msv = {msub, msub};
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const f32x2 t = f32x2{S[i], S[i + 1]} * c2v + msv;      // v_pk_fma_f32
            const f32x2 pr = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            l2 += pr;
            S[i] = pr[0];
            S[i + 1] = pr[1];
        }
        if (a.drop_thresh) {
            const uint32_t rk = rq + (uint32_t)(kt * 32) * ATT_G;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (att_mix(rk + (uint32_t)(8 * g4 + e) * ATT_G) < a.drop_cmp) S[4 * g4 + e] = 0.f;
        }
        const char* vimg = Vimg + 4096 * kt;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (bf16)S[8 * s2 + j];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) O[dt] = Elem<bf16>::mfma(tfrag(vimg + 2048 * s2, dt), pf, O[dt]);
        }
    };
    // two tiles per trip, the S^T buffers alternate (no register copies): the product of the next tile is in flight
    // while the current one is consumed
    f32x16 S0 = s_tile(0), S1;
    for (int kt = 0; kt < nq; kt += 2) {
        if (kt + 1 < nq) S1 = s_tile(kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        consume(S0, kt);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nq) {
            if (kt + 2 < nq) S0 = s_tile(kt + 2);
            __builtin_amdgcn_sched_barrier(0);
            consume(S1, kt + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float l_run = l2[0] + l2[1];
    const float l_tot = l_run + other_half(l_run);
    if (qi < N) {
        const float inv = a.inv_keep / l_tot;
        bf16* op = a.out + (size_t)qrow * a.d + hd * 64 + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 o = {(bf16)(O[dt][4 * g4 + 0] * inv), (bf16)(O[dt][4 * g4 + 1] * inv),
                            (bf16)(O[dt][4 * g4 + 2] * inv), (bf16)(O[dt][4 * g4 + 3] * inv)};
                *(bf16x4*)(op + dt * 32 + 8 * g4) = o;
            }
        if (h == 0 && a.lse) a.lse[(size_t)bh * a.lse_stride + qi] = (m_run * c2 + log2f(l_tot)) * LN2;
    }
}

// ------------------------------------------------------------------ backward
// Column sums for the qkv-bias gradient (vlmo.py:71-75: q_bias / v_bias; the k third is a constant zero): t[dt][r] holds
// feature dt * 32 + (r & 3) + 8 (r >> 2) + 4 h of the token on lane & 31.  Sum over the wave's 32 tokens (butterfly inside
// each half-wave), scale, and add into acc[64] in LDS; tokens past the sequence carry weight 0.
// inclusive DPP scan over each 32-lane half (gfx9 row_shr / row_bcast15 sequence): lanes 31 and 63 end up with their
// half's total.  Six v_add_f32_dpp; the same sum through __shfl_xor is five ds_bpermute round trips per register and
// made the two reductions of a wave cost more than the 51 MB re-read of dqkv they replace.
__device__ __forceinline__ float half_wave_total(float v) {
    float s_ = v + dpp_mov<0x111, 0xf, 0xf>(v);         // row_shr:1 (lanes without a source read 0)
    s_ += dpp_mov<0x112, 0xf, 0xf>(v);                  // row_shr:2
    s_ += dpp_mov<0x113, 0xf, 0xf>(v);                  // row_shr:3  -> sums of 4
    s_ += dpp_mov<0x114, 0xf, 0xe>(s_);                 // row_shr:4, banks 1-3 -> sums of 8
    s_ += dpp_mov<0x118, 0xf, 0xc>(s_);                 // row_shr:8, banks 2-3 -> lane 15 of a row = the row's total
    s_ += dpp_mov<0x142, 0xa, 0xf>(s_);                 // row_bcast15 into rows 1 and 3 -> lanes 31 / 63 = half totals
    return s_;
}
__device__ __forceinline__ void colsum_tiles(const f32x16* t, float w, float scale, float* acc, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = half_wave_total(t[dt][r] * w);
            if ((lane & 31) == 31) atomicAdd(acc + dt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, v * scale);
        }
}

// The backward kernels run one workgroup per CU and stage ~165 KB per (sequence, head) before their first product; in a
// training step those rows (the forward's qkv and ctx) come from HBM.  Their main loops issue no global loads at all, so
// a workgroup uses the idle vector-memory path to pull the rows of a LATER workgroup of its own XCD (blockIdx % 8 picks
// the XCD: + 256 = one dispatch round on) into that XCD's L2: token i of that item = thread i, one dword out of each
// 128-byte row piece of q, k, v, dctx and ctx.  The values are folded into a word that is consumed after the loop (the
// workgroup's last act is to retire them).
// The loads are inline assembly into ONE register that stays live until warm_retire: written as C++ loads the compiler
// folds the five values right behind the loads (one live register instead of five) and parks a full HBM round trip in
// front of the main loop.  vmcnt retires in order, so the compiler's own counted waits only become stricter.
__device__ __forceinline__ void warm_issue(const AttnArgs& a, int bh2, uint32_t& w) {
    w = 0;
    if (bh2 < (int)gridDim.x) {
        const int sidx2 = bh2 / a.heads, hd2 = bh2 % a.heads;
        const int rowA = a.seg[4 * sidx2 + 0], lenA = a.seg[4 * sidx2 + 1], rowB = a.seg[4 * sidx2 + 2], lenB = a.seg[4 * sidx2 + 3];
        const int i = threadIdx.x;
        if (i < lenA + lenB) {
            const size_t row = (size_t)(i < lenA ? rowA + i : rowB + (i - lenA));
            const bf16* q = a.qkv + row * 3 * a.d + hd2 * 64;
            const bf16* k = q + a.d;
            const bf16* v = k + a.d;
            const bf16* g = a.dctx + row * a.d + hd2 * 64;
            const bf16* o = a.ctx + row * a.d + hd2 * 64;
            asm volatile("global_load_dword %0, %1, off" : "+v"(w) : "v"(q) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "+v"(w) : "v"(k) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "+v"(w) : "v"(v) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "+v"(w) : "v"(g) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "+v"(w) : "v"(o) : "memory");
        }
    }
}
__device__ __forceinline__ void warm_retire(uint32_t& w) {
    asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %0" : "+v"(w)::"memory");
}

__global__ __launch_bounds__(512, 2) void attn_bwd_kernel(const AttnArgs a, const int NPAD) {
    const int IMG = NPAD * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qimg = smem;
    char* Kimg = smem + IMG;
    char* Vimg = smem + 2 * IMG;
    char* Dimg = smem + 3 * IMG;
    float* kbias = (float*)(smem + 4 * IMG);
    float* lseq = kbias + NPAD;    // log2-domain LSE per query (+inf on padded queries)
    float* delta = lseq + NPAD;
    int* rowidx = (int*)(delta + NPAD);
    int* next_item = rowidx + NPAD;
    float* csum = (float*)(next_item + 4);      // [2][64]: column sums of dq | dv of this (sequence, head)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.x, sidx = bh / a.heads, hd = bh % a.heads;
    const int ld = 3 * a.d;
    int N;
    setup_rows(a.seg, sidx, a.keymask, NPAD, rowidx, kbias, N);
    if (threadIdx.x == 0) *next_item = 0;
    if (threadIdx.x < 128) csum[threadIdx.x] = 0.f;
    __syncthreads();
    const int nq = (N + 31) >> 5;   // query tiles == key tiles (self-attention)
    // fringe mode (a.split_tok): only the tile pairs with the LAST tile on either side; the rest is attn_bwd1_kernel's
    const int ft = a.split_tok ? nq - 1 : -1;
    if (a.split_tok && N <= a.split_tok) return;        // nothing beyond the single-pass kernel's reach (uniform exit)
    stage_image<8>(a.qkv, ld, hd * 64, rowidx, Qimg, nq * 4, wave, lane);
    stage_image<8>(a.qkv, ld, a.d + hd * 64, rowidx, Kimg, nq * 4, wave, lane);
    stage_image<8>(a.qkv, ld, 2 * a.d + hd * 64, rowidx, Vimg, nq * 4, wave, lane);
    stage_image<8>(a.dctx, a.d, hd * 64, rowidx, Dimg, nq * 4, wave, lane);
    const float keep_prob = 1.f / a.inv_keep;
    for (int i = threadIdx.x; i < NPAD; i += 512) {
        float dl = 0.f, lq = INFINITY;
        if (i < N) {
            const size_t o = (size_t)rowidx[i] * a.d + hd * 64;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const bf16x8 x = *(const bf16x8*)(a.ctx + o + 8 * c);
                const bf16x8 y = *(const bf16x8*)(a.dctx + o + 8 * c);
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += (float)x[j] * (float)y[j];
            }
            lq = a.lse[(size_t)bh * a.lse_stride + i] * LOG2E;
        }
        delta[i] = dl * keep_prob;      // dS = scale * inv_keep * P * (keep * dP - delta / inv_keep): see below
        lseq[i] = lq;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t warm_word = 0;
    if (a.warm) warm_issue(a, bh + a.warm, warm_word);

    const int l31 = lane & 31, h = lane >> 5;
    const uint32_t akey = att_key(a.seed, bh + a.bh0);
    // the constant factors of dS = P * (keep * dP * inv_keep - delta) * scale and Pd = keep * P * inv_keep are
    // applied ONCE to the 32 accumulator values a lane owns, not to every score element
    const float out_scale = a.scale * a.inv_keep;

    // running column sums of the tiles this wave finishes (a.qvsum): one kind at a time -- items are handed out dK/dV
    // first, then dQ, so a wave switches at most once -- reduced over the lanes when the kind changes and at the end
    f32x16 cs[2] = {zero16(), zero16()};
    int cs_kind = -1;
    for (;;) {
    int item = 0;
    if (lane == 0) item = atomicAdd(next_item, 1);
    item = __builtin_amdgcn_readfirstlane(item);
    if (a.qvsum) {
        const int kind = item >= 2 * nq ? -1 : (item >= nq ? 0 : 1);        // 0: dq sums, 1: dv sums
        if (cs_kind >= 0 && kind != cs_kind) {
            colsum_tiles(cs, 1.f, cs_kind == 0 ? out_scale : a.inv_keep, csum + 64 * cs_kind, lane);
            cs[0] = zero16(), cs[1] = zero16();
        }
        cs_kind = kind;
    }
    if (item >= 2 * nq) break;
    // ---- dQ item: dQ^T[d][q] = sum_k K^T[d][k] dS^T[k][q]
    if (item >= nq) {
        // fringe mode: the long item (the fringe query tile against every key tile) is handed out first
        const int qt = ft < 0 ? item - nq : (item == nq ? ft : item - nq - 1);
        const int kt0 = (ft < 0 || qt == ft) ? 0 : ft;
        const int qi = qt * 32 + l31;
        bf16x8 qf[4], df[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = row_frag(Qimg, qt * 32, s, lane);
            df[s] = row_frag(Dimg, qt * 32, s, lane);
        }
        const float lq = lseq[qi], dl = delta[qi];
        const uint32_t rq = ((uint32_t)qi * 512u + 4u * h) * ATT_G + akey;
        f32x16 dQ[2] = {zero16(), zero16()};
        for (int kt = kt0; kt < nq; ++kt) {
            {
                f32x16 S = zero16(), dP = zero16();
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    S = Elem<bf16>::mfma(row_frag(Kimg, kt * 32, s, lane), qf[s], S);
                    dP = Elem<bf16>::mfma(row_frag(Vimg, kt * 32, s, lane), df[s], dP);
                }
                const uint32_t rk = rq + (uint32_t)(kt * 32) * ATT_G;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 kb = *(const f32x4*)(kbias + kt * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * g4 + e;
                        const float p = __builtin_amdgcn_exp2f(S[i] * a.scale_log2e + (kb[e] - lq));
                        bool keep = true;
                        if (a.drop_thresh) keep = att_mix(rk + (uint32_t)(8 * g4 + e) * ATT_G) >= a.drop_cmp;
                        const float dp = keep ? dP[i] : 0.f;
                        S[i] = p * (dp - dl);
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 sf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) sf[j] = (bf16)S[8 * s2 + j];
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        dQ[dt] = Elem<bf16>::mfma(tr_frag(Kimg, kt * 32 + 16 * s2, dt * 32, lane), sf, dQ[dt]);
                }
            }
        }
        if (a.qvsum) {
            // (fringe mode: the partial rows enter the sums through the single-pass launch, which adds them to its tiles)
            const float wq = (qi < N && (ft < 0 || qt == ft)) ? 1.f : 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) cs[dt][r] = fmaf(dQ[dt][r], wq, cs[dt][r]);
        }
        if (qi < N) {
            bf16* op = a.out + (size_t)rowidx[qi] * ld + hd * 64 + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    bf16x4 o = {(bf16)(dQ[dt][4 * g4 + 0] * out_scale), (bf16)(dQ[dt][4 * g4 + 1] * out_scale),
                                (bf16)(dQ[dt][4 * g4 + 2] * out_scale), (bf16)(dQ[dt][4 * g4 + 3] * out_scale)};
                    *(bf16x4*)(op + dt * 32 + 8 * g4) = o;
                }
        }
    }

    // ---- dK/dV item: dV^T[d][k] = sum_q dO^T[d][q] Pd[q][k] ; dK^T[d][k] = sum_q Q^T[d][q] dS[q][k]
    else {
        const int kt = ft < 0 ? item : (item == 0 ? ft : item - 1);
        const int qt0 = (ft < 0 || kt == ft) ? 0 : ft;
        const int ki = kt * 32 + l31;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = row_frag(Kimg, kt * 32, s, lane);
            vf[s] = row_frag(Vimg, kt * 32, s, lane);
        }
        const float kb = kbias[ki];
        // dropout counter of (q, ki) = q * 512 + ki: affine in q, so a lane walks its 16 queries of a tile with
        // compile-time constant adds
        const uint32_t rl = (uint32_t)ki * ATT_G + akey + (uint32_t)(4 * h) * ATT_G512;
        f32x16 dK[2] = {zero16(), zero16()}, dV[2] = {zero16(), zero16()};
        for (int qt = qt0; qt < nq; ++qt) {
            {
                f32x16 S = zero16(), dP = zero16();
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    S = Elem<bf16>::mfma(row_frag(Qimg, qt * 32, s, lane), kf[s], S);
                    dP = Elem<bf16>::mfma(row_frag(Dimg, qt * 32, s, lane), vf[s], dP);
                }
                f32x16 Pd;
                const uint32_t rqt = rl + (uint32_t)(qt * 32) * ATT_G512;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int q0 = qt * 32 + 8 * g4 + 4 * h;
                    const f32x4 lq = *(const f32x4*)(lseq + q0);
                    const f32x4 dl = *(const f32x4*)(delta + q0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * g4 + e;
                        const float p = __builtin_amdgcn_exp2f(S[i] * a.scale_log2e + (kb - lq[e]));
                        bool keep = true;
                        if (a.drop_thresh) keep = att_mix(rqt + (uint32_t)(8 * g4 + e) * ATT_G512) >= a.drop_cmp;
                        Pd[i] = keep ? p : 0.f;
                        S[i] = p * ((keep ? dP[i] : 0.f) - dl[e]);
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf, sf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        pf[j] = (bf16)Pd[8 * s2 + j];
                        sf[j] = (bf16)S[8 * s2 + j];
                    }
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        dV[dt] = Elem<bf16>::mfma(tr_frag(Dimg, qt * 32 + 16 * s2, dt * 32, lane), pf, dV[dt]);
                        dK[dt] = Elem<bf16>::mfma(tr_frag(Qimg, qt * 32 + 16 * s2, dt * 32, lane), sf, dK[dt]);
                    }
                }
            }
        }
        if (a.qvsum) {
            const float wk = (ki < N && (ft < 0 || kt == ft)) ? 1.f : 0.f;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) cs[dt][r] = fmaf(dV[dt][r], wk, cs[dt][r]);
        }
        if (ki < N) {
            bf16* op = a.out + (size_t)rowidx[ki] * ld + hd * 64 + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    bf16x4 ok = {(bf16)(dK[dt][4 * g4 + 0] * out_scale), (bf16)(dK[dt][4 * g4 + 1] * out_scale),
                                 (bf16)(dK[dt][4 * g4 + 2] * out_scale), (bf16)(dK[dt][4 * g4 + 3] * out_scale)};
                    bf16x4 ov = {(bf16)(dV[dt][4 * g4 + 0] * a.inv_keep), (bf16)(dV[dt][4 * g4 + 1] * a.inv_keep),
                                 (bf16)(dV[dt][4 * g4 + 2] * a.inv_keep), (bf16)(dV[dt][4 * g4 + 3] * a.inv_keep)};
                    *(bf16x4*)(op + a.d + dt * 32 + 8 * g4) = ok;
                    *(bf16x4*)(op + 2 * a.d + dt * 32 + 8 * g4) = ov;
                }
        }
    }
    }
    if (a.qvsum) {
        __syncthreads();
        if (threadIdx.x < 128)
            a.qvsum[(size_t)sidx * 2 * a.d + (threadIdx.x >> 6) * a.d + hd * 64 + (threadIdx.x & 63)] = csum[threadIdx.x];
    }
    if (a.warm) warm_retire(warm_word);
}

// ------------------------------------------------------------------ backward, single pass
// One workgroup = one (sequence, head); wave w OWNS key tile w (32 keys: its K / V row fragments, dK^T and dV^T
// accumulators) AND the dQ^T accumulators of query tile w.  The nq x nq tile pairs are swept along the diagonals of a
// rotation: in step t wave w works on (query tile (w + t) mod nq, key tile w), so S, dP, P and dS of every pair are
// computed exactly ONCE (the two-phase kernel above computes them in the dK/dV items and again in the dQ items: 28
// MFMAs and two passes of soft-max / dropout arithmetic per pair instead of 20 and one).  With the key on the lane
// (S = Q . K^T, rows = queries) the accumulator tiles P and dS are directly the B operands of dV^T += dO^T . P and
// dK^T += Q^T . dS; only dS crosses LDS, once: the wave stores its bf16 tile [key][query] into its slot and, after the
// step's barrier, the owner of that query tile reads it back transposed (ds_read_b64_tr_b16) as the B operand of
// dQ^T += K^T . dS^T.  Slots are double buffered: one barrier per step.
// Row constants ride in as INITIAL accumulators: S starts at -lse / scale and dP at -delta, so p = exp2(c * S' + keybias)
// and dS = p * dP' need no per-element subtraction (cdna guide, attention backward).
// LDS at 256 padded tokens (8 key tiles): Q, dO, K images 3 x 32 KB + 2 x 8 slots of 2 KB + row constants = 132 KB.
__device__ __forceinline__ int slot_off(int k, int q) {
    // [key][query] bf16 tile, 64-byte rows, 8-byte units XOR-swizzled by the key pair: conflict-free for the producer's
    // ds_write_b64 (16 consecutive keys, one unit) and for the consumer's transposed reads (4 keys x 8 units)
    return k * 64 + ((((q >> 2) ^ (k >> 1)) & 7) << 3) + (q & 3) * 2;
}
// One wave per key tile, at most 8 (two per SIMD, 251 registers): a ninth wave would put three on one SIMD, i.e. a
// 168-register budget against 96 accumulator registers + V fragments + the S / dP tiles -- hipcc spills 160-330
// registers there and the N = 261 backward takes 200-450 us instead of 120.  Sequences of 257-288 tokens (the fused
// layers at T = 64) therefore stay on the two-phase kernel above.
// Measured (MI355X, B = 64, 12 heads, dropout 0.1; tools/attn_bench.py): N = 197 78 us (two-phase 94), N = 64 16 us (25).
// Per workgroup at N = 197 (s_memrealtime stamps): 7 us before the first product (165 KB through one CU's vector
// memory path at ~25 GB/s: images + V fragments + the ctx / dctx rows for delta), 13 us in the loop (1.9 us per step:
// the two waves of a SIMD spend it in ~180 + ~180 soft-max / dropout instructions, SQ_ACTIVE_INST_VALU is what bounds
// it, the 40 MFMAs of both hide under it), 2-3 us of stores; one workgroup per CU, so the three phases add.
__global__ __launch_bounds__(512) void attn_bwd1_kernel(const AttnArgs a, const int NPAD) {
    constexpr bool KF_REG = false;      // K row fragments re-read from the LDS image each step (16 registers: no spill)
    const int IMG = NPAD * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qimg = smem;
    char* Dimg = smem + IMG;
    char* Kimg = smem + 2 * IMG;
    const int nt = NPAD >> 5;
    char* slots = smem + 3 * IMG;                           // [2][nt][2048]
    float* kbias = (float*)(slots + 2 * nt * 2048);
    float* nlq = kbias + NPAD;      // -lse / scale per query (-inf on padded queries)
    float* ndl = nlq + NPAD;        // -delta * keep_prob
    float* csum = ndl + NPAD;       // [2][64]: column sums of dq | dv of this (sequence, head)


    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int bh = blockIdx.x, sidx = bh / a.heads, hd = bh % a.heads;
    const int ld = 3 * a.d;
    // Prologue with ONE level of dependent loads: the packed row of a token is arithmetic on the sequence's four
    // descriptor words (scalar loads), so the LDS-DMA of the three images, the key mask, the rows of ctx / dctx for
    // delta, the log-sum-exp and this wave's V fragments are all issued together (a row table in LDS first costs two more
    // global round trips and a barrier: 6.4 of the 7.2 us a workgroup spent before its first product).
    const int rowA = a.seg[4 * sidx + 0], lenA = a.seg[4 * sidx + 1], rowB = a.seg[4 * sidx + 2], lenB = a.seg[4 * sidx + 3];
    // a.split_tok: the tokens beyond it (and their partial sums into the rows below it) were attn_bwd_kernel's
    const bool add_partials = a.split_tok && lenA + lenB > a.split_tok;
    const int N = add_partials ? a.split_tok : lenA + lenB;
    auto rowof = [&](int tok) {
        tok = min(tok, N - 1);
        return tok < lenA ? rowA + tok : rowB + (tok - lenA);
    };
    const int nq = (N + 31) >> 5;
    for (int ii = w; ii < nq * 4; ii += nw) {
        const int chhi = lane >> 5, rowlo = (lane >> 2) & 7, pc = lane & 3;
        const int row = ii * 8 + rowlo;
        const int ch = chhi * 4 + (pc ^ ((row >> 2) & 3));
        const size_t r = (size_t)rowof(row);
        glds16(a.qkv + r * ld + hd * 64 + ch * 8, Qimg + ii * 1024);
        glds16(a.qkv + r * ld + a.d + hd * 64 + ch * 8, Kimg + ii * 1024);
        glds16(a.dctx + r * a.d + hd * 64 + ch * 8, Dimg + ii * 1024);
    }
    const int l31 = lane & 31, h = lane >> 5;
    const bool active = w < nq;
    const int ki = w * 32 + l31;
    const int krow = rowof(ki);
    // row fragments (B operands) of this wave's keys straight from global memory: V has no LDS image at all
    bf16x8 kf[4], vf[4];
    if (active) {
        const bf16* kp = a.qkv + (size_t)krow * ld + a.d + hd * 64 + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (KF_REG) kf[s] = *(const bf16x8*)(kp + 16 * s);
            vf[s] = *(const bf16x8*)(kp + a.d + 16 * s);
        }
    }
    for (int i = threadIdx.x; i < 128; i += blockDim.x) csum[i] = 0.f;      // a one-tile launch has 64 threads
    const float keep_prob = 1.f / a.inv_keep;
    const float inv_scale = 1.f / a.scale;
    for (int i = threadIdx.x; i < NPAD; i += blockDim.x) {
        float dl = 0.f, lq = INFINITY;
        const int row = rowof(i);
        const bool ok = (i < N) && (!a.keymask || a.keymask[row] != 0);
        if (i < N) {
            const size_t o = (size_t)row * a.d + hd * 64;
            bf16x8 x[8], y[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                x[c] = *(const bf16x8*)(a.ctx + o + 8 * c);
                y[c] = *(const bf16x8*)(a.dctx + o + 8 * c);
            }
            lq = a.lse[(size_t)bh * a.lse_stride + i];
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) dl += (float)x[c][j] * (float)y[c][j];
        }
        kbias[i] = ok ? 0.f : -INFINITY;
        ndl[i] = -dl * keep_prob;       // dS = scale * inv_keep * P * (keep * dP - delta * keep_prob)
        nlq[i] = -lq * inv_scale;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t warm_word = 0;
    if (a.warm) warm_issue(a, bh + a.warm, warm_word);

    const uint32_t akey = att_key(a.seed, bh + a.bh0);
    const float out_scale = a.scale * a.inv_keep;
    const float c_l2 = a.scale_log2e;
    const float kb = kbias[min(ki, NPAD - 1)];
    // dropout counter of (q, ki) = q * 512 + ki: affine in q (see attn_bwd_kernel)
    const uint32_t rl = (uint32_t)ki * ATT_G + akey + (uint32_t)(4 * h) * ATT_G512;
    f32x16 dK[2] = {zero16(), zero16()}, dV[2] = {zero16(), zero16()}, dQ[2] = {zero16(), zero16()};

    // LDS addressing with FOUR per-lane offsets for all image accesses (att_off spelled out for tile-aligned bases: the
    // row base enters as 128 * row (a scalar), the k-step / feature half as immediates), so that the loop does not
    // carry ~50 address registers:
    //   row fragment (rows rb + l31, 16-byte chunk 2 s + h):   img + 128 rb + 512 (s >> 1) + rf[s & 1]
    //   transposed fragment (rows rb + ..., features cb + ...): img + 128 rb + 512 (cb >> 5) + {trl | trh}
    const int xq = (l31 >> 2) & 3;
    const int rf0 = 1024 * (l31 >> 3) + 64 * (l31 & 7) + 16 * (h ^ xq);
    const int rf1 = 1024 * (l31 >> 3) + 64 * (l31 & 7) + 16 * ((2 + h) ^ xq);
    const int tg = (lane >> 4) & 1, tq = (lane >> 2) & 3, tp = lane & 3;
    const int trl = 64 * (4 * h + tq) + 16 * ((2 * tg + (tp >> 1)) ^ h) + 8 * (tp & 1);
    const int trh = 1024 + 64 * (4 * h + tq) + 16 * ((2 * tg + (tp >> 1)) ^ ((2 + h) & 3)) + 8 * (tp & 1);
    auto rfrag = [&](const char* img_rb, int s) -> bf16x8 {     // img_rb = image + 128 * row base (wave-uniform)
        return *(const bf16x8*)(img_rb + 512 * (s >> 1) + ((s & 1) ? rf1 : rf0));
    };
    auto tfrag = [&](const char* img_rb, int dt) -> bf16x8 {    // img_rb = image + 128 * (row base of the 16-row k-step)
        const bf16x4 lo = lds_tr4<bf16>(img_rb + 512 * dt + trl);
        const bf16x4 hi = lds_tr4<bf16>(img_rb + 512 * dt + trh);
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    // slot tile [key][query] (slot_off): the producer's two 8-byte stores per k-step, the consumer's transposed reads
    const int sw0 = slot_off(l31, 4 * h), sw1 = slot_off(l31, 8 + 4 * h);           // k-step 1: + 32 bytes XOR-wise
    const int srl = slot_off(4 * h + tq, 16 * tg + 4 * tp), srh = slot_off(8 + 4 * h + tq, 16 * tg + 4 * tp);
    auto sfrag = [&](const char* slot, int s2) -> bf16x8 {
        const bf16x4 lo = lds_tr4<bf16>(slot + 1024 * s2 + srl);
        const bf16x4 hi = lds_tr4<bf16>(slot + 1024 * s2 + srh);
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };

    // One step of a wave = four stages:
    //   sdp   S = Q K^T - lse / scale, dP = dO V^T                      8 MFMAs   (row fragments of Q, dO)
    //   valu  p = exp2(c S + keybias), dropout, dS = p (dP - delta)      ~15 VALU instructions per element; dS -> slot
    //   dvdk  dV^T += dO^T P, dK^T += Q^T dS                             8 MFMAs   (transposed fragments of dO, Q)
    //   dq    dQ^T += K^T dS^T for the query tile this wave owns         4 MFMAs   (behind the step's barrier)
    // Left alone, the compiler issues every LDS read right in front of the MFMA that uses it (the LDS latency is then
    // exposed 20 times per step): operands are fetched a stage ahead and __builtin_amdgcn_sched_barrier(0) keeps the
    // fetch groups where they are written.
    // The two waves of a SIMD (w and w + 4) run the stages in DIFFERENT orders between two barriers, so that one is in
    // its MFMA stages while the other does arithmetic (in lockstep every SIMD alternates between idle matrix pipe and
    // idle vector issue, and all waves queue on the LDS port together):
    //   w <  4:  | dq(t-1) sdp(t) valu(t) dvdk(t)          | barrier t
    //   w >= 4:  | valu(t) dvdk(t) dq(t-1) sdp(t+1)        | barrier t        (sdp(0) in front of the loop)
    // Legal because dq(t-1) only needs barrier t-1 behind it and the slot buffer it reads is rewritten after barrier t.
    bf16x8 qa[4], da[4], pf[2], sf[2];
    f32x16 S, dP;
    auto fetch_a = [&](int qt_) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qa[s] = rfrag(Qimg + qt_ * 4096, s);
            da[s] = rfrag(Dimg + qt_ * 4096, s);
        }
    };
    auto qtile = [&](int t_) {
        int q_ = w + t_;
        return q_ >= nq ? q_ - nq : q_;
    };
    auto stage_sdp = [&](int qt) {      // qa / da hold the fragments of query tile qt
        // the row constant -lse / scale as the initial accumulator of S (dP starts at zero: its row constant -delta
        // is needed as a value by the dropout select anyway, and a second set of initial registers spills)
        dP = zero16();
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 l4 = *(const f32x4*)(nlq + qt * 32 + 8 * g4 + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) S[4 * g4 + e] = l4[e];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 kfs = KF_REG ? kf[s] : rfrag(Kimg + w * 4096, s);
            S = Elem<bf16>::mfma(qa[s], kfs, S);
            dP = Elem<bf16>::mfma(da[s], vf[s], dP);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stage_valu_dvdk = [&](int qt, int t) {
        bf16x8 dtr[2], qtr[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            dtr[dt] = tfrag(Dimg + qt * 4096, dt);
            qtr[dt] = tfrag(Qimg + qt * 4096, dt);
        }
        __builtin_amdgcn_sched_barrier(0);
        // P (after dropout) and dS are packed to bf16 as they are produced: registers 8 s2 .. 8 s2 + 7 of a tile are
        // the fragment of k-step s2 of the products that follow
        if (a.drop_thresh) {
            const uint32_t rqt = rl + (uint32_t)(qt * 32) * ATT_G512;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 nd = *(const f32x4*)(ndl + qt * 32 + 8 * g4 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g4 + e;
                    const float p = __builtin_amdgcn_exp2f(fmaf(S[i], c_l2, kb));
                    const bool keep = att_mix(rqt + (uint32_t)(8 * g4 + e) * ATT_G512) >= a.drop_cmp;
                    const float pd = keep ? p : 0.f;
                    pf[i >> 3][i & 7] = (bf16)pd;
                    sf[i >> 3][i & 7] = (bf16)fmaf(pd, dP[i], p * nd[e]);      // p * (keep * dP - delta)
                }
            }
        } else {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 nd = *(const f32x4*)(ndl + qt * 32 + 8 * g4 + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * g4 + e;
                    const float p = __builtin_amdgcn_exp2f(fmaf(S[i], c_l2, kb));
                    pf[i >> 3][i & 7] = (bf16)p;
                    sf[i >> 3][i & 7] = (bf16)(p * (dP[i] + nd[e]));
                }
            }
        }
        char* myslot = slots + ((t & 1) * nt + w) * 2048;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            // dS^T tile for the owner of this query tile: registers 4g .. 4g+3 = queries 8g + 4h .. +3 of key l31
            *(bf16x4*)(myslot + (sw0 ^ (32 * s2))) = bf16x4{sf[s2][0], sf[s2][1], sf[s2][2], sf[s2][3]};
            *(bf16x4*)(myslot + (sw1 ^ (32 * s2))) = bf16x4{sf[s2][4], sf[s2][5], sf[s2][6], sf[s2][7]};
        }
        __builtin_amdgcn_sched_barrier(0);
        // the second k-step's fragments are fetched under the first one's products
        bf16x8 dtr1[2], qtr1[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            dtr1[dt] = tfrag(Dimg + qt * 4096 + 2048, dt);
            qtr1[dt] = tfrag(Qimg + qt * 4096 + 2048, dt);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            dV[dt] = Elem<bf16>::mfma(dtr[dt], pf[0], dV[dt]);
            dK[dt] = Elem<bf16>::mfma(qtr[dt], sf[0], dK[dt]);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            dV[dt] = Elem<bf16>::mfma(dtr1[dt], pf[1], dV[dt]);
            dK[dt] = Elem<bf16>::mfma(qtr1[dt], sf[1], dK[dt]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // dq(t): the wave that worked on query tile w in step t owns key tile wp = (w - t) mod nq.  next_q >= 0: also fetch
    // the row fragments of that query tile (the following sdp stage) under these products
    auto stage_dq = [&](int t, int next_q) {
        int wp = w - t;
        if (wp < 0) wp += nq;
        const char* slot = slots + ((t & 1) * nt + wp) * 2048;
        bf16x8 ktr[2][2], sb[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            sb[s2] = sfrag(slot, s2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) ktr[s2][dt] = tfrag(Kimg + wp * 4096 + 2048 * s2, dt);
        }
        if (next_q >= 0) fetch_a(next_q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) dQ[dt] = Elem<bf16>::mfma(ktr[s2][dt], sb[s2], dQ[dt]);
        __builtin_amdgcn_sched_barrier(0);
    };
    // VALU issue is arbitrated by priority, then age: left alone the younger wave of a SIMD (w >= 4) gets the leftovers of
    // the older one's arithmetic stage and every barrier waits for it (interval 3 900 -> 3 540 cycles with this)
    if (w >= 4) __builtin_amdgcn_s_setprio(1);
    if (!active) {
        for (int t = 0; t < nq; ++t) __syncthreads();
    } else if (w < 4) {
        fetch_a(w);
        for (int t = 0; t < nq; ++t) {
            const int qt = qtile(t);
            stage_sdp(qt);
            stage_valu_dvdk(qt, t);
            __syncthreads();
            stage_dq(t, t + 1 < nq ? qtile(t + 1) : -1);
        }
    } else {
        fetch_a(w);
        stage_sdp(w);
        for (int t = 0; t < nq; ++t) {
            stage_valu_dvdk(qtile(t), t);
            if (t > 0) stage_dq(t - 1, -1);
            if (t + 1 < nq) {
                fetch_a(qtile(t + 1));
                stage_sdp(qtile(t + 1));
            }
            __syncthreads();
        }
        stage_dq(nq - 1, -1);
    }
    if (active && ki < N) {
        // ki doubles as the query index of the dQ tile this wave owns
        bf16* op = a.out + (size_t)krow * ld + hd * 64 + 4 * h;
        if (add_partials) {
            // the fringe launch left bf16 partial sums in these rows: added in fp32 in front of the one final rounding
            // (in units of the accumulators, so that the stores below stay as they are)
            const float iq = 1.f / out_scale, iv = keep_prob;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const bf16x4 pq = *(const bf16x4*)(op + dt * 32 + 8 * g4);
                    const bf16x4 pk = *(const bf16x4*)(op + a.d + dt * 32 + 8 * g4);
                    const bf16x4 pv = *(const bf16x4*)(op + 2 * a.d + dt * 32 + 8 * g4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        dQ[dt][4 * g4 + e] = fmaf((float)pq[e], iq, dQ[dt][4 * g4 + e]);
                        dK[dt][4 * g4 + e] = fmaf((float)pk[e], iq, dK[dt][4 * g4 + e]);
                        dV[dt][4 * g4 + e] = fmaf((float)pv[e], iv, dV[dt][4 * g4 + e]);
                    }
                }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 oq = {(bf16)(dQ[dt][4 * g4 + 0] * out_scale), (bf16)(dQ[dt][4 * g4 + 1] * out_scale),
                             (bf16)(dQ[dt][4 * g4 + 2] * out_scale), (bf16)(dQ[dt][4 * g4 + 3] * out_scale)};
                bf16x4 ok = {(bf16)(dK[dt][4 * g4 + 0] * out_scale), (bf16)(dK[dt][4 * g4 + 1] * out_scale),
                             (bf16)(dK[dt][4 * g4 + 2] * out_scale), (bf16)(dK[dt][4 * g4 + 3] * out_scale)};
                bf16x4 ov = {(bf16)(dV[dt][4 * g4 + 0] * a.inv_keep), (bf16)(dV[dt][4 * g4 + 1] * a.inv_keep),
                             (bf16)(dV[dt][4 * g4 + 2] * a.inv_keep), (bf16)(dV[dt][4 * g4 + 3] * a.inv_keep)};
                *(bf16x4*)(op + dt * 32 + 8 * g4) = oq;
                *(bf16x4*)(op + a.d + dt * 32 + 8 * g4) = ok;
                *(bf16x4*)(op + 2 * a.d + dt * 32 + 8 * g4) = ov;
            }
    }
    if (a.qvsum) {
        if (active) {
            const float wt = ki < N ? 1.f : 0.f;
            colsum_tiles(dQ, wt, out_scale, csum, lane);
            colsum_tiles(dV, wt, a.inv_keep, csum + 64, lane);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 128; i += blockDim.x) {
            float* dst = a.qvsum + (size_t)sidx * 2 * a.d + (i >> 6) * a.d + hd * 64 + (i & 63);
            *dst = add_partials ? *dst + csum[i] : csum[i];     // the fringe launch left the fringe rows' sums there
        }
    }
    if (a.warm) warm_retire(warm_word);
}

// dynamic-LDS limit already raised for a kernel on a device (a process may drive several GPUs)
int& lds_limit_set(int which) {
    static int lim[4][64];
    static bool init = false;
    if (!init) {
        for (auto& r : lim)
            for (int& v : r) v = 65536;
        init = true;
    }
    int dev = 0;
    (void)hipGetDevice(&dev);
    return lim[which][dev & 63];
}

int launch_fwd(const AttnArgs& a, int nt, int nblocks, hipStream_t st) {
    const int LDS = nt * 32 * 256 + nt * 32 * 8;
    int& max_set = lds_limit_set(0);
    if (LDS > max_set) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        max_set = LDS;
    }
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(nblocks), dim3(256), LDS, st, a, nt * 32);
    return 0;
}
int launch_fwd1(const AttnArgs& a, int nt, int nblocks, hipStream_t st) {
    const int LDS = nt * 32 * 256 + nt * 32 * 4;
    int& max_set = lds_limit_set(3);
    if (LDS > max_set) {
        (void)hipFuncSetAttribute((const void*)attn_fwd1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        max_set = LDS;
    }
    hipLaunchKernelGGL(attn_fwd1_kernel, dim3(nblocks), dim3(nt * 64), LDS, st, a, nt * 32);
    return 0;
}
int launch_bwd(const AttnArgs& a, int nt, int nblocks, hipStream_t st) {
    const int LDS = nt * 32 * 512 + nt * 32 * 16 + 32 + 512;
    int& max_set = lds_limit_set(1);
    if (LDS > max_set) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        max_set = LDS;
    }
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(nblocks), dim3(512), LDS, st, a, nt * 32);
    return 0;
}

int launch_bwd1(const AttnArgs& a, int nt, int nblocks, hipStream_t st) {
    const int LDS = nt * 32 * 384 + 2 * nt * 2048 + nt * 32 * 12 + 512;
    int& max_set = lds_limit_set(2);
    if (LDS > max_set) {
        (void)hipFuncSetAttribute((const void*)attn_bwd1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        max_set = LDS;
    }
    hipLaunchKernelGGL(attn_bwd1_kernel, dim3(nblocks), dim3(nt * 64), LDS, st, a, nt * 32);
    return 0;
}

int check_common(const char* fn, const void* qkv, const int32_t* seg, int num_seq, int heads, int d, int max_len,
                 int cap) {
    VLMO_CHECK_ARG(qkv && seg, "%s: null pointer", fn);
    VLMO_CHECK_ARG(num_seq > 0 && heads > 0, "%s: empty problem", fn);
    VLMO_CHECK_ARG(d == heads * 64, "%s: head_dim must be 64 (d=%d, heads=%d)", fn, d, heads);
    VLMO_CHECK_ARG(max_len > 0 && max_len <= cap, "%s: sequence length %d exceeds this build's limit %d", fn, max_len, cap);
    return 0;
}

}  // namespace

extern "C" int vlmo_attn_fwd(const void* qkv, const int32_t* seg, int num_seq, const int32_t* keymask, void* ctx,
                             float* lse, int lse_stride, int heads, int d, int max_len, float scale,
                             uint32_t drop_thresh, float inv_keep, uint64_t seed, int mask_seq0, hipStream_t stream) {
    if (int rc = check_common("vlmo_attn_fwd", qkv, seg, num_seq, heads, d, max_len, 576)) return rc;
    VLMO_CHECK_ARG(ctx, "vlmo_attn_fwd: null ctx");
    VLMO_CHECK_ARG(!lse || lse_stride >= max_len, "vlmo_attn_fwd: lse_stride too small");
    AttnArgs a{};
    a.qkv = (const bf16*)qkv;
    a.out = (bf16*)ctx;
    a.lse = lse;
    a.seg = seg;
    a.keymask = keymask;
    a.lse_stride = lse_stride;
    a.heads = heads;
    a.d = d;
    a.scale = scale;
    a.scale_log2e = scale * LOG2E;
    a.drop_thresh = drop_thresh;
    a.drop_cmp = drop_thresh >= 65536u ? 0xFFFFFFFFu : drop_thresh << 16;
    a.inv_keep = drop_thresh ? inv_keep : 1.f;
    a.seed = seed;
    a.bh0 = mask_seq0 * heads;
    const int nt = (max_len + 31) / 32, nb = num_seq * heads;
    static const bool chunked = [] {
        const char* e = getenv("VLMO_ATTN_FWD");
        return e && !strcmp(e, "chunked");
    }();
    if (nt <= 9 && !chunked) launch_fwd1(a, nt, nb, stream);
    else launch_fwd(a, nt, nb, stream);
    VLMO_CHECK_LAUNCH("vlmo_attn_fwd");
    return 0;
}

extern "C" int vlmo_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse, int lse_stride,
                             const int32_t* seg, int num_seq, const int32_t* keymask, void* dqkv, float* qv_colsum,
                             int heads, int d, int max_len, float scale, uint32_t drop_thresh, float inv_keep,
                             uint64_t seed, int mask_seq0, hipStream_t stream) {
    if (int rc = check_common("vlmo_attn_bwd", qkv, seg, num_seq, heads, d, max_len, 288)) return rc;
    VLMO_CHECK_ARG(ctx && dctx && lse && dqkv, "vlmo_attn_bwd: null pointer");
    VLMO_CHECK_ARG(lse_stride >= max_len, "vlmo_attn_bwd: lse_stride too small");
    AttnArgs a{};
    a.qkv = (const bf16*)qkv;
    a.ctx = (const bf16*)ctx;
    a.dctx = (const bf16*)dctx;
    a.out = (bf16*)dqkv;
    a.qvsum = qv_colsum;
    a.lse = (float*)lse;
    a.seg = seg;
    a.keymask = keymask;
    a.lse_stride = lse_stride;
    a.heads = heads;
    a.d = d;
    a.scale = scale;
    a.scale_log2e = scale * LOG2E;
    a.drop_thresh = drop_thresh;
    a.drop_cmp = drop_thresh >= 65536u ? 0xFFFFFFFFu : drop_thresh << 16;
    a.inv_keep = drop_thresh ? inv_keep : 1.f;
    a.seed = seed;
    a.bh0 = mask_seq0 * heads;
    static const int warm = getenv("VLMO_ATTN_WARM") ? atoi(getenv("VLMO_ATTN_WARM")) : 0;      // measurement aid
    a.warm = warm;
    const int nt = (max_len + 31) / 32, nb = num_seq * heads;
    static const bool two_phase = getenv("VLMO_ATTN_BWD") && !strcmp(getenv("VLMO_ATTN_BWD"), "two_phase");   // measurement aid
    static const bool split = getenv("VLMO_ATTN_BWD_SPLIT") && atoi(getenv("VLMO_ATTN_BWD_SPLIT")) != 0;       // measurement aid, off
    if (!two_phase && split && nt == 9) {
        // 257 .. 288 tokens (the fused layers at 64 text tokens: 261): one tile too many for the single-pass kernel's
        // eight tile owners, and the two-phase kernel computes every S / dP tile twice.  The backward is a sum over
        // (query tile, key tile) pairs whose terms only share the row constants lse and delta, so the 17 pairs that touch
        // the ninth tile go to the two-phase kernel (two long items + sixteen one-step items; it writes the fringe rows
        // and bf16 partial sums into the rows below) and the 64 pairs among the first 256 tokens to the single-pass
        // kernel, which adds the partials in fp32 in front of its one rounding.  Exact (tests/test_kernels_gpu.py ran
        // green on this path) and NOT the default: 91 + 36 us against 131 us back to back, but +0.34 ms per step
        // (14.33 -> 14.67 ms, two in-session pairs) -- both launches stage every operand of a (sequence, head) through one
        // CU, and in the step those operands are cold (tools/attn_cold_probe.py: +16 us per launch), so the prologue
        // is paid twice.
        a.split_tok = 256;
        launch_bwd(a, nt, nb, stream);
        VLMO_CHECK_LAUNCH("vlmo_attn_bwd(fringe)");
        launch_bwd1(a, 8, nb, stream);
    } else if (two_phase || nt > 8)
        launch_bwd(a, nt, nb, stream);
    else
        launch_bwd1(a, nt, nb, stream);
    VLMO_CHECK_LAUNCH("vlmo_attn_bwd");
    return 0;
}
