// Multi-tensor optimizer step for the VLMo training loop: global gradient L2 norm + clip coefficient and a
// fused Adam / AdamW update over a LIST of parameter tensors in one launch each (the reference's default
// optimizer is apex FusedAdam(adam_w_mode=True), utils/optim_factory.py:185-186; the clip is
// torch.nn.utils.clip_grad_norm_ inside NativeScalerWithGradNormCount, utils/utils.py:343-364).
// HBM-bound: 16 B read + 12 B written per parameter; work is cut into fixed-size chunks so that ~300 tensors of
// very different sizes fill the chip evenly, one workgroup per chunk.
#include "common.h"
#include "vlmo_hip.h"

namespace {

constexpr int OPT_THREADS = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x < OPT_THREADS / 64) t = red[threadIdx.x];
    if (w == 0) t = wave_sum(t);
    return t;      // valid in wave 0
}

// partial[c] = sum of g^2 over chunk c
__global__ __launch_bounds__(OPT_THREADS) void mt_sqnorm_kernel(const VlmoTensorList tl, float* __restrict__ partial) {
    __shared__ float red[OPT_THREADS / 64];
    const int c = blockIdx.x;
    const int t = tl.chunk_tensor[c];
    const int64_t off = tl.chunk_start[c];
    const int64_t n = min((int64_t)tl.chunk, tl.numel[t] - off);
    const float* g = (const float*)tl.g[t] + off;
    float a = 0.f;
    if ((((uintptr_t)g) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += OPT_THREADS) {
            const f32x4 v = ((const f32x4*)g)[i];
            a += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += OPT_THREADS) a += g[i] * g[i];
    } else {
        for (int64_t i = threadIdx.x; i < n; i += OPT_THREADS) a += g[i] * g[i];
    }
    const float s = block_sum(a, red);
    if (threadIdx.x == 0) partial[c] = s;
}

// out[0] = ||g||_2, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)) (1 when max_norm <= 0),
// out[2] = 1 if the norm is not finite (the update is then skipped, as torch.cuda.amp.GradScaler.step does)
__global__ __launch_bounds__(OPT_THREADS) void mt_norm_finish_kernel(const float* __restrict__ partial, int n, float inv_scale,
                                                                    float max_norm, float* __restrict__ out) {
    __shared__ float red[OPT_THREADS / 64];
    double a = 0.0;     // fixed summation order: deterministic
    for (int i = threadIdx.x; i < n; i += OPT_THREADS) a += (double)partial[i];
    const float s = block_sum((float)a, red);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(s) * inv_scale;
        const bool bad = !(norm == norm) || norm > 3.0e38f;
        out[0] = norm;
        float coef = inv_scale;
        if (max_norm > 0.f) coef *= fminf(1.f, max_norm / (norm + 1e-6f));
        out[1] = bad ? 0.f : coef;
        out[2] = bad ? 1.f : 0.f;
    }
}

__global__ __launch_bounds__(OPT_THREADS) void mt_adam_kernel(const VlmoTensorList tl, const VlmoAdamArgs a,
                                                             const float* __restrict__ ctl) {
    const float gscale = ctl ? ctl[1] : 1.f;
    if (ctl && ctl[2] != 0.f) return;       // non-finite gradients: skip the step
    const int c = blockIdx.x;
    const int t = tl.chunk_tensor[c];
    const int64_t off = tl.chunk_start[c];
    const int64_t n = min((int64_t)tl.chunk, tl.numel[t] - off);
    float* p = (float*)tl.p[t] + off;
    const float* g = (const float*)tl.g[t] + off;
    float* m = (float*)tl.m[t] + off;
    float* v = (float*)tl.v[t] + off;
    const float lr = tl.lr[t], wd = tl.wd[t];
    const float b1 = a.beta1, b2 = a.beta2, ob1 = 1.f - a.beta1, ob2 = 1.f - a.beta2;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= gscale;
        if (!a.adam_w_mode) gg += wd * pp;                 // L2 mode: decay enters the moments
        mm = b1 * mm + ob1 * gg;
        vv = b2 * vv + ob2 * gg * gg;
        const float denom = sqrtf(vv * a.inv_bc2) + a.eps;
        float u = (mm * a.inv_bc1) / denom;
        if (a.adam_w_mode) u += wd * pp;                   // decoupled decay
        pp -= lr * u;
    };
    const bool al = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    if (al) {
        const int64_t n4 = n >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += OPT_THREADS) {
            f32x4 pp = ((f32x4*)p)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
            const f32x4 gg = ((const f32x4*)g)[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x = pp[j], y = mm[j], z = vv[j];
                upd(x, gg[j], y, z);
                pp[j] = x, mm[j] = y, vv[j] = z;
            }
            ((f32x4*)p)[i] = pp, ((f32x4*)m)[i] = mm, ((f32x4*)v)[i] = vv;
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += OPT_THREADS) upd(p[i], g[i], m[i], v[i]);
    } else {
        for (int64_t i = threadIdx.x; i < n; i += OPT_THREADS) upd(p[i], g[i], m[i], v[i]);
    }
}

int check_list(const VlmoTensorList* tl, const char* who) {
    VLMO_CHECK_ARG(tl && tl->n_chunks >= 0 && tl->chunk > 0 && tl->chunk % 4 == 0, "%s: bad tensor list", who);
    VLMO_CHECK_ARG(tl->n_chunks == 0 || (tl->g && tl->numel && tl->chunk_tensor && tl->chunk_start), "%s: null table", who);
    return 0;
}

}  // namespace

extern "C" int vlmo_mt_grad_norm(const VlmoTensorList* tl, float inv_scale, float max_norm, float* partial, float* out,
                                 hipStream_t stream) {
    if (int rc = check_list(tl, "vlmo_mt_grad_norm")) return rc;
    VLMO_CHECK_ARG(partial && out, "vlmo_mt_grad_norm: null output");
    if (tl->n_chunks > 0) {
        hipLaunchKernelGGL(mt_sqnorm_kernel, dim3(tl->n_chunks), dim3(OPT_THREADS), 0, stream, *tl, partial);
        VLMO_CHECK_LAUNCH("vlmo_mt_grad_norm");
    }
    hipLaunchKernelGGL(mt_norm_finish_kernel, dim3(1), dim3(OPT_THREADS), 0, stream, partial, tl->n_chunks, inv_scale, max_norm, out);
    VLMO_CHECK_LAUNCH("vlmo_mt_grad_norm(finish)");
    return 0;
}

extern "C" int vlmo_mt_adam(const VlmoTensorList* tl, const VlmoAdamArgs* a, const float* ctl, hipStream_t stream) {
    if (int rc = check_list(tl, "vlmo_mt_adam")) return rc;
    VLMO_CHECK_ARG(a && a->beta1 >= 0.f && a->beta1 < 1.f && a->beta2 >= 0.f && a->beta2 < 1.f && a->eps >= 0.f,
                   "vlmo_mt_adam: bad hyper-parameters");
    if (tl->n_chunks == 0) return 0;
    VLMO_CHECK_ARG(tl->p && tl->m && tl->v && tl->lr && tl->wd, "vlmo_mt_adam: null table");
    hipLaunchKernelGGL(mt_adam_kernel, dim3(tl->n_chunks), dim3(OPT_THREADS), 0, stream, *tl, *a, ctl);
    VLMO_CHECK_LAUNCH("vlmo_mt_adam");
    return 0;
}
