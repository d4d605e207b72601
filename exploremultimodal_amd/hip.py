"""ctypes binding of libvlmo_hip.so (include/vlmo_hip.h).

PyTorch is used only for device memory and streams: every wrapper takes torch
tensors, passes raw device pointers + the caller's CURRENT stream to the C-ABI
and returns immediately (the library only enqueues work).  There is no CPU or
eager-PyTorch fallback: if the library is missing, importing this module works
but the first call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VLMO_HIP_LIB: load another build of the same ABI (A/B timing of kernel variants in one GPU session)
LIB_PATH = os.environ.get('VLMO_HIP_LIB') or os.path.join(_HERE, 'lib', 'libvlmo_hip.so')

BF16, F16, F32 = 0, 1, 2
EPI_BIAS, EPI_BIAS_GELU, EPI_RESID, EPI_DGELU, EPI_F32, EPI_DUAL, EPI_ARGMAX, EPI_CE, EPI_CE_BWD = 0, 1, 2, 3, 4, 5, 6, 7, 8

_vp, _i32, _u32, _u64, _f32, _i64 = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32,
                                     ctypes.c_uint64, ctypes.c_float, ctypes.c_int64)


class Epilogue(ctypes.Structure):
    _fields_ = [('out', _vp), ('out2', _vp), ('bias', _vp), ('gamma', _vp),
                ('resid', _vp), ('row_scale', _vp), ('row_index', _vp), ('aux', _vp),
                ('ldo', _i32), ('ld2', _i32), ('relu', _i32),
                ('drop_thresh', _u32), ('inv_keep', _f32), ('beta', _f32),
                ('seed', _u64), ('colpart', _vp)]


_fp = _vp      # device pointers are passed as integers (tensor.data_ptr())


class BlockDesc(ctypes.Structure):
    """Mirror of VlmoBlockDesc (include/vlmo_hip.h); field order and types must match."""
    _fields_ = (
        [('M', _i32), ('d', _i32), ('hidden', _i32), ('heads', _i32),
         ('n_experts', _i32), ('exp_row0', _i32 * 2), ('exp_rows', _i32 * 2),
         ('n_attn', _i32), ('nseq', _i32 * 2), ('maxlen', _i32 * 2), ('lse_stride', _i32 * 2),
         ('attn_seed_idx', _i32 * 2), ('attn_seq0', _i32 * 2),
         ('seg', _fp * 2), ('keymask', _fp), ('eps', _f32),
         ('drop_thresh', _u32), ('attn_drop_thresh', _u32), ('inv_keep', _f32), ('attn_inv_keep', _f32),
         ('seed', _u64), ('rs1', _fp), ('rs2', _fp), ('row_index', _fp), ('tile', _i32), ('need_bwd', _i32)]
        + [(n, _fp) for n in ('g1', 'g2', 'n1w', 'n1b', 'n2w', 'n2b', 'qkv_bias', 'proj_b',
                              'qkv_w', 'qkv_wT', 'proj_w', 'proj_wT')]
        + [(n, _fp * 2) for n in ('b1', 'b2', 'w1', 'w1T', 'w2', 'w2T')]
        + [(n, _fp) for n in ('x', 'x1', 'x2', 'y1', 'qkv', 'ctx', 'zd1', 'y2', 'u', 'h', 'zd2',
                              'mean1', 'rstd1', 'mean2', 'rstd2')]
        + [('lse', _fp * 2)]
        + [(n, _fp) for n in ('dx2', 'dx1', 'dx0', 'dz2', 'du', 'dy2', 'dz1', 'dctx', 'dqkv', 'dy1',
                              'dg1', 'dg2', 'dn1w', 'dn1b', 'dn2w', 'dn2b', 'dqkv_w', 'dqkv_b',
                              'dproj_w', 'dproj_b')]
        + [(n, _fp * 2) for n in ('dw1', 'db1', 'dw2', 'db2')]
        + [('ws_main', _fp), ('ws_side', _fp), ('ws_bytes', _i64), ('ws_tn', _fp), ('ws_tn_bytes', _i64),
           ('side_stream', _fp)])


class StackDesc(ctypes.Structure):
    """Mirror of VlmoStackDesc (include/vlmo_hip.h)."""
    _fields_ = [('n_blocks', _i32), ('wgrad_batch', _i32), ('n_tmp_sets', _i32), ('wgrad_store', _i32),
                ('blocks', ctypes.POINTER(BlockDesc)), ('side_stream', _vp), ('grad_ready', ctypes.POINTER(_vp))]


class TnProblem(ctypes.Structure):
    """Mirror of VlmoTnProblem."""
    _fields_ = [('A', _vp), ('B', _vp), ('C', _vp), ('lda', _i32), ('ldb', _i32), ('ldc', _i32),
                ('M', _i32), ('N1', _i32), ('N2', _i32), ('alpha', _f32), ('accumulate', _i32)]


class ColJob(ctypes.Structure):
    """Mirror of VlmoColJob."""
    _fields_ = [('kind', _i32), ('ld', _i32), ('src', _vp), ('rows', _i32), ('ncols', _i32),
                ('out', _vp * 4), ('n0', _i32), ('pad_', _i32)]


_SIGS = {
    'vlmo_gemm_nt': [_i32, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32,
                     ctypes.POINTER(Epilogue), _vp],
    'vlmo_gemm_tn': [_i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _i64, _vp],
    'vlmo_ln_fwd': [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _f32, _vp],
    'vlmo_ln_bwd': [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp],
    'vlmo_ln_resid_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _f32, _u64,
                          _i32, _i32, _vp, _i64, _vp],
    'vlmo_attn_fwd': [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _u32, _f32,
                      _u64, _i32, _vp],
    'vlmo_attn_bwd': [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _f32,
                      _u32, _f32, _u64, _i32, _vp],
    'vlmo_resid_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _u32, _f32, _u64, _vp, _i64, _vp],
    'vlmo_colsum': [_i32, _vp, _i32, _vp, _i32, _i32, _vp, _i64, _vp],
    'vlmo_cast_weight': [_i32, _vp, _i32, _i32, _vp, _vp, _vp],
    'vlmo_cast_weight_multi': [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    'vlmo_patchify': [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    'vlmo_embed_img_finish': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _u32, _f32,
                              _u64, _vp],
    'vlmo_embed_img_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _u32, _f32,
                           _u64, _vp],
    'vlmo_embed_txt_fwd': [_vp] * 10 + [_i32, _i32, _i32, _f32, _u32, _f32, _u64, _vp],
    'vlmo_embed_txt_bwd': [_vp] * 11 + [_i32, _i32, _i32, _u32, _f32, _u64, _vp],
    'vlmo_conv2d_nhwc': [_i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp,
                         ctypes.POINTER(Epilogue), _vp],
    'vlmo_dvae_im2col': [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    'vlmo_maxpool2_nhwc': [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    'vlmo_argmax_reduce': [_vp, _i32, _vp, _i32, _vp],
    'vlmo_ce_reduce': [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp],
    'vlmo_block_fwd': [ctypes.POINTER(BlockDesc), _vp],
    'vlmo_block_bwd': [ctypes.POINTER(BlockDesc), _vp],
    'vlmo_stack_fwd': [ctypes.POINTER(StackDesc), _vp],
    'vlmo_stack_bwd': [ctypes.POINTER(StackDesc), _vp],
    'vlmo_gemm_tn_multi': [_i32, ctypes.POINTER(TnProblem), _i32, _vp],
    'vlmo_colwork_multi': [_i32, ctypes.POINTER(ColJob), _i32, _vp],
    'vlmo_event_create': [ctypes.POINTER(ctypes.c_void_p)],
    'vlmo_event_destroy': [_vp],
    'vlmo_stream_wait_event': [_vp, _vp],
    'vlmo_profile_start': [_i32],
    'vlmo_gemm_nt_grouped': [_i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp],
    'vlmo_mt_grad_norm': [ctypes.c_void_p, _f32, _f32, _vp, _vp, _vp],
    'vlmo_mt_adam': [ctypes.c_void_p, ctypes.c_void_p, _vp, _vp],
    'vlmo_side_stream_create': [_i32, ctypes.POINTER(ctypes.c_uint32), _i32, ctypes.POINTER(ctypes.c_void_p)],
    'vlmo_profile_stop': [_i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                          ctypes.POINTER(ctypes.c_int64)],
    'vlmo_gemm_nt_2src': [_i32, _i32, _i32, _vp, _i32, _i32, _f32, _vp, _i32, _vp, _i32, _i32, _i32, _i32,
                          ctypes.POINTER(Epilogue), _vp],
    'vlmo_comm_available': [],
    'vlmo_comm_unique_id': [_vp],
    'vlmo_comm_init': [ctypes.POINTER(ctypes.c_void_p), _vp, _i32, _i32],
    'vlmo_comm_destroy': [_vp],
    'vlmo_comm_count': [_vp, _vp, _vp],
    'vlmo_comm_all_reduce': [_vp, _vp, _vp, _i64, _i32, _vp],
    'vlmo_comm_reduce_scatter': [_vp, _vp, _vp, _i64, _i32, _vp],
    'vlmo_comm_all_gather': [_vp, _vp, _vp, _i64, _i32, _vp],
    'vlmo_grad_pack': [_vp, _vp, _i64, _f32, _vp],
    'vlmo_grad_unpack': [_vp, _vp, _i64, _vp],
}

_lib = None
ABI_VERSION = 5      # vlmo_abi_version(): struct layouts of include/vlmo_hip.h mirrored above

def lib():
    """Load (once) and return the C-ABI library; raise loudly if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: build the HIP extension first '
                '(python -c "import __graft_entry__ as g; g.build()" or '
                'make -C exploremultimodal_amd/csrc). There is no CPU fallback.')
        L = ctypes.CDLL(LIB_PATH)
        L.vlmo_last_error.restype = ctypes.c_char_p
        L.vlmo_abi_version.restype = ctypes.c_int
        if L.vlmo_abi_version() != ABI_VERSION:
            raise RuntimeError(f'{LIB_PATH} has ABI version {L.vlmo_abi_version()}, this package needs {ABI_VERSION}: '
                               'rebuild it (make -C exploremultimodal_amd/csrc)')
        L.vlmo_reduce_ws_bytes.restype = ctypes.c_int64
        L.vlmo_reduce_ws_bytes.argtypes = [_i32]
        L.vlmo_gemm_tn_ws_bytes.restype = ctypes.c_int64
        L.vlmo_gemm_tn_ws_bytes.argtypes = [_i32, _i32, _i32]
        for name, sig in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = sig
            fn.restype = ctypes.c_int
        _lib = L
    return _lib


def exported_symbols():
    return ['vlmo_last_error', 'vlmo_abi_version', 'vlmo_reduce_ws_bytes', 'vlmo_gemm_tn_ws_bytes'] + list(_SIGS)


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f'{name} failed (rc={rc}): {lib().vlmo_last_error().decode()}')


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dt(t):
    return {torch.bfloat16: BF16, torch.float16: F16, torch.float32: F32}[t.dtype]


_WS = {}


def workspace(device, ncols):
    """Persistent scratch for the column-reducing kernels, one per (device, stream): reuse is ordered by
    the stream, and kernels on different streams never share a buffer."""
    need = lib().vlmo_reduce_ws_bytes(ncols)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty(max(need, 1 << 23) // 4, dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


def drop_params(p, training):
    """(thresh, inv_keep) of the counter-based dropout; thresh 0 disables it."""
    if not training or p <= 0.0:
        return 0, 1.0
    thresh = int(round(p * 65536))
    return thresh, 65536.0 / (65536 - thresh)


# ------------------------------------------------------------------ wrappers

def gemm_nt(epi, A, B, M, N, K, out, *, out2=None, bias=None, gamma=None, resid=None,
            row_scale=None, row_index=None, aux=None, ldo=None, ld2=None, relu=False, drop=(0, 1.0), seed=0,
            beta=0.0, tile=-1, lda=None, ldb=None, colpart=None, A2=None, k1=0, seg_scale=1.0):
    """A2 / k1 / seg_scale: two-segment reduction (vlmo_gemm_nt_2src): columns [0, k1) of B meet A, [k1, K) meet A2, the
    first segment's partial sum is multiplied by seg_scale."""
    e = Epilogue(_p(out), _p(out2), _p(bias), _p(gamma), _p(resid), _p(row_scale), _p(row_index), _p(aux),
                 ldo if ldo is not None else out.stride(0),
                 ld2 if ld2 is not None else (out2.stride(0) if out2 is not None else
                                              (aux.stride(0) if aux is not None else 0)),
                 int(relu), drop[0], drop[1], beta, seed & 0xFFFFFFFFFFFFFFFF, _p(colpart))
    if A2 is not None:
        rc = lib().vlmo_gemm_nt_2src(epi, _dt(A), tile, _p(A), lda if lda is not None else A.stride(0), k1,
                                     float(seg_scale), _p(A2), A2.stride(0), _p(B),
                                     ldb if ldb is not None else B.stride(0), M, N, K, ctypes.byref(e), _stream())
        _check(rc, 'vlmo_gemm_nt_2src')
        return
    rc = lib().vlmo_gemm_nt(epi, _dt(A), tile, _p(A), lda if lda is not None else A.stride(0),
                            _p(B), ldb if ldb is not None else B.stride(0), M, N, K,
                            ctypes.byref(e), _stream())
    _check(rc, 'vlmo_gemm_nt')


def gemm_nt_grouped(epi, As, Bs, Ms, N, K, outs, *, per_group=None, tile=-1, **common):
    """vlmo_gemm_nt_grouped: problems g = (As[g] [Ms[g], K], Bs[g] [N, K]) -> outs[g]; ``per_group[g]`` holds the
    epilogue keywords of gemm_nt that differ per group (bias, out2, seed, ...), ``common`` the shared ones."""
    n = len(As)
    es = (Epilogue * n)()
    for g in range(n):
        kw = dict(common)
        kw.update((per_group or [{}] * n)[g])
        out, out2, aux = outs[g], kw.get('out2'), kw.get('aux')
        drop = kw.get('drop', (0, 1.0))
        es[g] = Epilogue(_p(out), _p(out2), _p(kw.get('bias')), _p(kw.get('gamma')), _p(kw.get('resid')),
                         _p(kw.get('row_scale')), _p(kw.get('row_index')), _p(aux), out.stride(0),
                         out2.stride(0) if out2 is not None else (aux.stride(0) if aux is not None else 0),
                         int(kw.get('relu', False)), drop[0], drop[1], kw.get('beta', 0.0),
                         kw.get('seed', 0) & 0xFFFFFFFFFFFFFFFF, _p(kw.get('colpart')))
    pa = (ctypes.c_void_p * n)(*[_p(a) for a in As])
    pb = (ctypes.c_void_p * n)(*[_p(b) for b in Bs])
    ms = (ctypes.c_int32 * n)(*Ms)
    rc = lib().vlmo_gemm_nt_grouped(epi, _dt(As[0]), tile, n, pa, As[0].stride(0), pb, Bs[0].stride(0), ms, N, K, es, _stream())
    _check(rc, 'vlmo_gemm_nt_grouped')


_WS_TN = {}


def tn_workspace(device, nbytes):
    """Persistent slab scratch of the weight-gradient GEMM, one per (device, stream)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS_TN.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(max(nbytes, 1 << 24) // 4, dtype=torch.float32, device=device)
        _WS_TN[key] = ws
    return ws


def gemm_tn(A, B, C, M, N1, N2, alpha=1.0, splits=0, slab=True):
    ws = tn_workspace(A.device, max(splits % 1000, 16) * N1 * N2 * 4 if splits > 0 else
                      lib().vlmo_gemm_tn_ws_bytes(M, N1, N2)) if slab else None
    rc = lib().vlmo_gemm_tn(_dt(A), _p(A), A.stride(0), _p(B), B.stride(0), _p(C), C.stride(0),
                            M, N1, N2, alpha, splits, _p(ws), ws.numel() * 4 if ws is not None else 0, _stream())
    _check(rc, 'vlmo_gemm_tn')


def ln_fwd(x, w, b, y, mean, rstd, rowmap, M, d, eps):
    rc = lib().vlmo_ln_fwd(_p(x), _p(w), _p(b), _p(y), int(y.dtype == torch.float32), _p(mean),
                           _p(rstd), _p(rowmap), M, d, eps, _stream())
    _check(rc, 'vlmo_ln_fwd')


def ln_bwd(dy, rowmap, x, w, mean, rstd, dres, dx, dw, db, M, d):
    rc = lib().vlmo_ln_bwd(_p(dy), int(dy.dtype == torch.float32), _p(rowmap), _p(x), _p(w),
                           _p(mean), _p(rstd), _p(dres), _p(dx), _p(dw), _p(db), M, d, _p(ws := workspace(x.device, 2 * d)),
                           ws.numel() * 4, _stream())
    _check(rc, 'vlmo_ln_bwd')


def ln_resid_bwd(dy, x, w, mean, rstd, dres, dx, dw, db, zd, gamma, row_scale, row_index, dz, dgamma, dbias, M, d,
                 drop=(0, 1.0), seed=0):
    ws = workspace(x.device, 4 * d)
    rc = lib().vlmo_ln_resid_bwd(_p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dres), _p(dx), _p(dw), _p(db), _p(zd),
                                 _p(gamma), _p(row_scale), _p(row_index), _p(dz), _p(dgamma), _p(dbias), drop[0], drop[1],
                                 seed & 0xFFFFFFFFFFFFFFFF, M, d, _p(ws), ws.numel() * 4, _stream())
    _check(rc, 'vlmo_ln_resid_bwd')


def attn_fwd(qkv, seg, nseq, keymask, ctx, lse, heads, d, max_len, scale, drop=(0, 1.0), seed=0, mask_seq0=0):
    rc = lib().vlmo_attn_fwd(_p(qkv), _p(seg), nseq, _p(keymask), _p(ctx), _p(lse),
                             lse.stride(0) if lse is not None else 0, heads, d, max_len, scale,
                             drop[0], drop[1], seed & 0xFFFFFFFFFFFFFFFF, mask_seq0, _stream())
    _check(rc, 'vlmo_attn_fwd')


def attn_bwd(qkv, ctx, dctx, lse, seg, nseq, keymask, dqkv, heads, d, max_len, scale,
             drop=(0, 1.0), seed=0, qv_colsum=None, mask_seq0=0):
    rc = lib().vlmo_attn_bwd(_p(qkv), _p(ctx), _p(dctx), _p(lse), lse.stride(0), _p(seg), nseq,
                             _p(keymask), _p(dqkv), _p(qv_colsum), heads, d, max_len, scale, drop[0], drop[1],
                             seed & 0xFFFFFFFFFFFFFFFF, mask_seq0, _stream())
    _check(rc, 'vlmo_attn_bwd')


def resid_bwd(dx, zd, gamma, row_scale, dz, dgamma, dbias, M, d, drop=(0, 1.0), seed=0, row_index=None):
    ws = workspace(dx.device, 2 * d)
    rc = lib().vlmo_resid_bwd(_p(dx), _p(zd), _p(gamma), _p(row_scale), _p(row_index), _p(dz), _p(dgamma),
                              _p(dbias), M, d, drop[0], drop[1], seed & 0xFFFFFFFFFFFFFFFF,
                              _p(ws), ws.numel() * 4, _stream())
    _check(rc, 'vlmo_resid_bwd')


def colsum(x, out, M, N):
    ws = workspace(x.device, N)
    rc = lib().vlmo_colsum(_dt(x), _p(x), x.stride(0), _p(out), M, N, _p(ws), ws.numel() * 4, _stream())
    _check(rc, 'vlmo_colsum')


def cast_weight(src, dst, dstT):
    rows, cols = src.shape
    ref = dst if dst is not None else dstT
    rc = lib().vlmo_cast_weight(_dt(ref), _p(src), rows, cols, _p(dst), _p(dstT), _stream())
    _check(rc, 'vlmo_cast_weight')


def cast_weight_multi(jobs):
    """jobs: [(src fp32 [rows, cols], dst or None, dstT or None)] -> one launch (per 72 jobs)."""
    n = len(jobs)
    if n == 0:
        return
    ref = jobs[0][1] if jobs[0][1] is not None else jobs[0][2]
    src = (ctypes.c_void_p * n)(*[_p(a) for a, _, _ in jobs])
    dst = (ctypes.c_void_p * n)(*[_p(b) for _, b, _ in jobs])
    dstT = (ctypes.c_void_p * n)(*[_p(c) for _, _, c in jobs])
    rows = (ctypes.c_int32 * n)(*[a.shape[0] for a, _, _ in jobs])
    cols = (ctypes.c_int32 * n)(*[a.shape[1] for a, _, _ in jobs])
    _check(lib().vlmo_cast_weight_multi(_dt(ref), n, src, rows, cols, dst, dstT, _stream()), 'vlmo_cast_weight_multi')


def patchify(img, out, patch):
    B, C, H, W = img.shape
    rc = lib().vlmo_patchify(_p(img), _p(out), B, C, H, W, patch, _stream())
    _check(rc, 'vlmo_patchify')


def embed_img_finish(proj, cls_tok, mask_tok, pos, type_row, masked, x, B, npatch, d,
                     drop=(0, 1.0), seed=0):
    rc = lib().vlmo_embed_img_finish(_p(proj), _p(cls_tok), _p(mask_tok), _p(pos), _p(type_row),
                                     _p(masked), _p(x), B, npatch, d, drop[0], drop[1],
                                     seed & 0xFFFFFFFFFFFFFFFF, _stream())
    _check(rc, 'vlmo_embed_img_finish')


def embed_img_bwd(dx, masked, dproj, dcls, dmask, dpos, dtype_row, B, npatch, d, drop=(0, 1.0),
                  seed=0):
    rc = lib().vlmo_embed_img_bwd(_p(dx), _p(masked), _p(dproj), _p(dcls), _p(dmask), _p(dpos),
                                  _p(dtype_row), B, npatch, d, drop[0], drop[1],
                                  seed & 0xFFFFFFFFFFFFFFFF, _stream())
    _check(rc, 'vlmo_embed_img_bwd')


def embed_txt_fwd(ids, word, pos, btype0, ln_w, ln_b, type0, x, xhat, rstd, B, T, d, eps,
                  drop=(0, 1.0), seed=0):
    rc = lib().vlmo_embed_txt_fwd(_p(ids), _p(word), _p(pos), _p(btype0), _p(ln_w), _p(ln_b),
                                  _p(type0), _p(x), _p(xhat), _p(rstd), B, T, d, eps, drop[0],
                                  drop[1], seed & 0xFFFFFFFFFFFFFFFF, _stream())
    _check(rc, 'vlmo_embed_txt_fwd')


def embed_txt_bwd(dx, ids, xhat, rstd, ln_w, dword, dpos, dbtype0, dln_w, dln_b, dtype0, B, T, d,
                  drop=(0, 1.0), seed=0):
    rc = lib().vlmo_embed_txt_bwd(_p(dx), _p(ids), _p(xhat), _p(rstd), _p(ln_w), _p(dword),
                                  _p(dpos), _p(dbtype0), _p(dln_w), _p(dln_b), _p(dtype0), B, T, d,
                                  drop[0], drop[1], seed & 0xFFFFFFFFFFFFFFFF, _stream())
    _check(rc, 'vlmo_embed_txt_bwd')


# ------------------------------------------------------------------ dVAE encoder
_ZERO = {}


def zero_page(device):
    z = _ZERO.get(device)
    if z is None:
        z = torch.zeros(256, dtype=torch.float32, device=device)
        _ZERO[device] = z
    return z


def conv2d_nhwc(epi, x, B, H, W, Cin, kw, w, Cout, out, *, out2=None, bias=None, resid=None, relu=False,
                relu_in=False, beta=0.0, ldo=None):
    """relu: ReLU on the output; relu_in: the convolution reads relu(x) (applied to the fragments, x stays as it is)."""
    e = Epilogue(_p(out), _p(out2), _p(bias), None, _p(resid), None, None, None,
                 ldo if ldo is not None else out.stride(0), out2.stride(0) if out2 is not None else 0,
                 int(relu) | (2 if relu_in else 0), 0, 1.0, beta, 0)
    rc = lib().vlmo_conv2d_nhwc(epi, _dt(x), _p(x), B, H, W, Cin, kw, _p(w), Cout, _p(zero_page(x.device)),
                                ctypes.byref(e), _stream())
    _check(rc, 'vlmo_conv2d_nhwc')


def dvae_im2col(x, out, kw, Kpad):
    B, C, H, W = x.shape
    _check(lib().vlmo_dvae_im2col(_p(x), _p(out), B, C, H, W, kw, Kpad, _stream()), 'vlmo_dvae_im2col')


def maxpool2_nhwc(x, raw, relu, B, H, W, C):
    _check(lib().vlmo_maxpool2_nhwc(_p(x), _p(raw), _p(relu), B, H, W, C, _stream()), 'vlmo_maxpool2_nhwc')


def argmax_reduce(partial, nchunk, ids, M):
    _check(lib().vlmo_argmax_reduce(_p(partial), nchunk, _p(ids), M, _stream()), 'vlmo_argmax_reduce')


def gemm_tn_multi(problems, dtype=BF16):
    """problems: [(A, B, C, M, N1, N2, accumulate)] with A [M, >=N1], B [M, >=N2] bf16/f16 and C [N1, >=N2] fp32."""
    n = len(problems)
    arr = (TnProblem * n)()
    for i, (A, B, C, M, N1, N2, acc) in enumerate(problems):
        arr[i] = TnProblem(_p(A), _p(B), _p(C), A.stride(0), B.stride(0), C.stride(0), M, N1, N2, 1.0, int(acc))
    _check(lib().vlmo_gemm_tn_multi(dtype, arr, n, _stream()), 'vlmo_gemm_tn_multi')


def colwork_multi(jobs, dtype=BF16):
    """jobs: [(kind, src, rows, ncols, outs, n0)]; kind 0 = fold fp32 partial rows src [rows, ncols],
    kind 1 = column sums of the bf16/f16 matrix src [rows, ld]."""
    n = len(jobs)
    arr = (ColJob * n)()
    for i, (kind, src, rows, ncols, outs, n0) in enumerate(jobs):
        o = (ctypes.c_void_p * 4)(*[_p(t) for t in (list(outs) + [None] * 4)[:4]])
        arr[i] = ColJob(kind, src.stride(0), _p(src), rows, ncols, o, n0, 0)
    _check(lib().vlmo_colwork_multi(dtype, arr, n, _stream()), 'vlmo_colwork_multi')


def event_create():
    out = ctypes.c_void_p()
    _check(lib().vlmo_event_create(ctypes.byref(out)), 'vlmo_event_create')
    return out.value


def stream_wait_event(stream, ev):
    """stream: torch stream (or raw handle); ev: handle from event_create()."""
    raw = stream.cuda_stream if hasattr(stream, 'cuda_stream') else stream
    _check(lib().vlmo_stream_wait_event(raw, ev), 'vlmo_stream_wait_event')


# ---- gradient exchange (include/vlmo_hip.h: vlmo_comm_*) -------------------------------------------------------------
COMM_ID_BYTES = 128


def comm_available():
    return bool(lib().vlmo_comm_available())


def comm_unique_id():
    """128 opaque bytes; rank 0 creates them, every rank passes the same ones to comm_init."""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    _check(lib().vlmo_comm_unique_id(buf), 'vlmo_comm_unique_id')
    return buf.raw


def comm_init(uid, rank, world):
    if len(uid) != COMM_ID_BYTES:
        raise ValueError(f'communicator id must be {COMM_ID_BYTES} bytes')
    out = ctypes.c_void_p()
    _check(lib().vlmo_comm_init(ctypes.byref(out), ctypes.c_char_p(bytes(uid)), rank, world), 'vlmo_comm_init')
    return out.value


def comm_destroy(comm):
    _check(lib().vlmo_comm_destroy(comm), 'vlmo_comm_destroy')


def comm_count(comm):
    """ranks in the communicator, as RCCL reports it."""
    n = ctypes.c_int(0)
    _check(lib().vlmo_comm_count(comm, ctypes.byref(n), None), 'vlmo_comm_count')
    return n.value


def comm_all_reduce(comm, t, stream=None):
    """in-place sum over the ranks, enqueued on `stream` (default: the current stream)."""
    raw = _stream() if stream is None else stream.cuda_stream
    _check(lib().vlmo_comm_all_reduce(comm, _p(t), _p(t), t.numel(), _dt(t), raw), 'vlmo_comm_all_reduce')


def comm_reduce_scatter(comm, out, src, stream=None):
    assert src.numel() % out.numel() == 0 and src.dtype == out.dtype
    raw = _stream() if stream is None else stream.cuda_stream
    _check(lib().vlmo_comm_reduce_scatter(comm, _p(src), _p(out), out.numel(), _dt(out), raw),
           'vlmo_comm_reduce_scatter')


def comm_all_gather(comm, out, src, stream=None):
    assert out.numel() % src.numel() == 0 and src.dtype == out.dtype
    raw = _stream() if stream is None else stream.cuda_stream
    _check(lib().vlmo_comm_all_gather(comm, _p(src), _p(out), src.numel(), _dt(src), raw), 'vlmo_comm_all_gather')


def grad_pack(src, dst, scale, stream=None):
    """dst (bf16) = src (fp32) * scale, one pass."""
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
    raw = _stream() if stream is None else stream.cuda_stream
    _check(lib().vlmo_grad_pack(_p(src), _p(dst), src.numel(), float(scale), raw), 'vlmo_grad_pack')


def grad_unpack(src, dst, stream=None):
    assert src.dtype == torch.bfloat16 and dst.dtype == torch.float32 and src.numel() == dst.numel()
    raw = _stream() if stream is None else stream.cuda_stream
    _check(lib().vlmo_grad_unpack(_p(src), _p(dst), src.numel(), raw), 'vlmo_grad_unpack')


def stack_fwd(sdesc):
    _check(lib().vlmo_stack_fwd(ctypes.byref(sdesc), _stream()), 'vlmo_stack_fwd')


def stack_bwd(sdesc):
    _check(lib().vlmo_stack_bwd(ctypes.byref(sdesc), _stream()), 'vlmo_stack_bwd')


def ce_reduce(partial, nchunk, labels, ignore_index, lse, loss, pred, M):
    _check(lib().vlmo_ce_reduce(_p(partial), nchunk, _p(labels), ignore_index, _p(lse), _p(loss), _p(pred), M, _stream()),
           'vlmo_ce_reduce')


def block_fwd(desc):
    _check(lib().vlmo_block_fwd(ctypes.byref(desc), _stream()), 'vlmo_block_fwd')


def block_bwd(desc):
    _check(lib().vlmo_block_bwd(ctypes.byref(desc), _stream()), 'vlmo_block_bwd')


PROFILE_TAGS = 96


class TensorList(ctypes.Structure):
    """VlmoTensorList (include/vlmo_hip.h): device tables of one multi-tensor optimizer launch."""
    _fields_ = [('p', _vp), ('g', _vp), ('m', _vp), ('v', _vp), ('numel', _vp), ('lr', _vp), ('wd', _vp),
                ('chunk_tensor', _vp), ('chunk_start', _vp), ('n_chunks', _i32), ('chunk', _i32)]


class AdamArgs(ctypes.Structure):
    _fields_ = [('beta1', _f32), ('beta2', _f32), ('eps', _f32), ('inv_bc1', _f32), ('inv_bc2', _f32),
                ('adam_w_mode', _i32)]


def mt_grad_norm(tl, inv_scale, max_norm, partial, out):
    """out[0] = grad norm, out[1] = factor for the raw gradients, out[2] = non-finite flag (device floats)."""
    _check(lib().vlmo_mt_grad_norm(ctypes.byref(tl), float(inv_scale), float(max_norm), _p(partial), _p(out), _stream()),
           'vlmo_mt_grad_norm')


def mt_adam(tl, args, ctl=None):
    _check(lib().vlmo_mt_adam(ctypes.byref(tl), ctypes.byref(args), _p(ctl), _stream()), 'vlmo_mt_adam')


def side_stream_create(low_priority=True, cu_mask=None):
    """Native stream for the weight-gradient work (include/vlmo_hip.h: vlmo_side_stream_create) -> raw
    hipStream_t value.  Create it with the target device current."""
    out = ctypes.c_void_p()
    if cu_mask:
        words = (ctypes.c_uint32 * len(cu_mask))(*cu_mask)
        rc = lib().vlmo_side_stream_create(0, words, len(cu_mask), ctypes.byref(out))
    else:
        rc = lib().vlmo_side_stream_create(1 if low_priority else 0, None, 0, ctypes.byref(out))
    _check(rc, 'vlmo_side_stream_create')
    return out.value


def profile_start(max_records=1 << 15):
    _check(lib().vlmo_profile_start(max_records), 'vlmo_profile_start')


def profile_stop():
    """-> {tag: (seconds, flops, launches)} for the launches recorded since profile_start()."""
    n = PROFILE_TAGS
    ms, fl, ln = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_int64 * n)()
    lib().vlmo_profile_stop(n, ms, fl, ln)
    names = {0: 'bias', 1: 'bias_gelu', 2: 'resid', 3: 'dgelu', 4: 'f32', 5: 'dual', 6: 'argmax', 7: 'ce', 8: 'ce_bwd'}
    out = {}
    for t in range(n):
        if ln[t]:
            if t in (64, 72):
                name = 'gemm_tn_kernel<%s>' % ('256x256' if t == 72 else '128x128')
            elif t == 73:
                name = 'gemm_tn_multi_kernel<256x256>'
            elif 80 <= t < 96:
                name = f'gemm_nt16_kernel<{names.get(t - 80, t - 80)},Hx256>'
            elif 48 <= t < 64:
                name = f'gemm_nt_kernel<{names.get(t - 48, t - 48)},256x128>'
            elif t >= 32:
                name = f'conv_nt_kernel<{names.get(t - 32, t - 32)}>'
            else:
                name = f'gemm_nt_kernel<{names.get(t & 15, t & 15)},{"256x256" if t & 16 else "128x128"}>'
            out[name] = (ms[t] * 1e-3, fl[t], ln[t])
    return out
