"""Host-side checks of the HIP kernels' LDS images: every MFMA operand fragment
receives the intended matrix elements and every read pattern is bank-conflict
free under the gfx950 rules (tests/ldssim.py).  No GPU needed."""
import numpy as np

from tests import ldssim as S


def _ids(rows, cols):
    return np.arange(rows * cols, dtype=np.int64).reshape(rows, cols)


def test_gemm_nt_image():
    ids = _ids(128, 64)
    lds = S.Lds(128 * 128)
    for instr in range(16):
        src = []
        for lane in range(64):
            r, c = S.nt_stage_src(instr, lane)
            src.append(ids[r, c * 8:c * 8 + 8])
        lds.dma16(instr * 1024, src)
    assert (lds.ids >= 0).all()
    for row_base in (0, 32, 64, 96):
        for ks in range(4):
            addrs = [S.nt_frag_addr(row_base + (l & 31), ks, l) for l in range(64)]
            got = lds.read_b128(addrs)
            for l in range(64):
                k0 = 16 * ks + 8 * (l >> 5)
                np.testing.assert_array_equal(got[l], ids[row_base + (l & 31), k0:k0 + 8])
            assert S.b128_conflicts(addrs) == 1


def test_gemm_tn_image():
    ids = _ids(64, 128)
    lds = S.Lds(64 * 256)
    for instr in range(16):
        src = []
        for lane in range(64):
            r, ch = S.tn_stage_src(instr, lane)
            src.append(ids[r, ch * 8:ch * 8 + 8])
        lds.dma16(instr * 1024, src)
    assert (lds.ids >= 0).all()
    worst = 1
    for ncol_base in (0, 32, 64, 96):
        for ks in range(4):
            for half in range(2):
                addrs = [S.tn_tr_addr(ncol_base, ks, half, l) for l in range(64)]
                got = lds.read_tr(addrs)
                for l in range(64):
                    n = ncol_base + (l & 31)
                    m0 = 16 * ks + 8 * (l >> 5) + 4 * half
                    np.testing.assert_array_equal(got[l], ids[m0:m0 + 4, n])
                worst = max(worst, S.tr_conflicts(addrs))
    assert worst == 1


def test_attention_dual_use_image():
    R = 96
    ids = _ids(R, 64)
    lds = S.Lds(R * 128)
    for instr in range(R // 8):
        src = []
        for lane in range(64):
            r, ch = S.att_stage_src(instr, lane)
            src.append(ids[r, ch * 8:ch * 8 + 8])
        lds.dma16(instr * 1024, src)
    assert (lds.ids >= 0).all()
    # row reads: A/B operand rows = image rows (keys or queries), k = feature
    for row_base in (0, 32, 64):
        for ks in range(4):
            addrs = [S.att_row_addr(row_base, ks, l) for l in range(64)]
            got = lds.read_b128(addrs)
            for l in range(64):
                k0 = 16 * ks + 8 * (l >> 5)
                np.testing.assert_array_equal(got[l], ids[row_base + (l & 31), k0:k0 + 8])
            assert S.b128_conflicts(addrs) == 1
    # transposed reads: operand rows = features, k = image rows in the
    # accumulator-as-operand order 16*s2 + 8*(j>>2) + 4*h + (j&3)
    worst = 1
    for kt in range(R // 32):
        for s2 in range(2):
            for half in range(2):
                for dt in range(2):
                    rb = kt * 32 + 16 * s2 + 8 * half
                    addrs = [S.att_tr_addr(rb, dt * 32, l) for l in range(64)]
                    got = lds.read_tr(addrs)
                    for l in range(64):
                        dcol = dt * 32 + (l & 31)
                        r0 = rb + 4 * (l >> 5)
                        np.testing.assert_array_equal(got[l], ids[r0:r0 + 4, dcol])
                    worst = max(worst, S.tr_conflicts(addrs))
    assert worst == 1


def test_gemm_nt_bk32_image():
    ids = _ids(128, 32)
    lds = S.Lds(128 * 64)
    for instr in range(8):
        src = []
        for lane in range(64):
            r, c = S.nt32_stage_src(instr, lane)
            src.append(ids[r, c * 8:c * 8 + 8])
        lds.dma16(instr * 1024, src)
    assert (lds.ids >= 0).all()
    for row_base in (0, 32, 64, 96):
        for ks in range(2):
            addrs = [S.nt32_frag_addr(row_base + (l & 31), ks, l) for l in range(64)]
            got = lds.read_b128(addrs)
            for l in range(64):
                k0 = 16 * ks + 8 * (l >> 5)
                np.testing.assert_array_equal(got[l], ids[row_base + (l & 31), k0:k0 + 8])
            assert S.b128_conflicts(addrs) == 1


def test_gemm_nt16_fragments():
    """16x16x32 operand fragments out of the SAME staged image as the 32x32x16 kernel (no new swizzle needed)."""
    ids = _ids(128, 64)
    lds = S.Lds(128 * 128)
    for instr in range(16):
        src = []
        for lane in range(64):
            r, c = S.nt_stage_src(instr, lane)
            src.append(ids[r, c * 8:c * 8 + 8])
        lds.dma16(instr * 1024, src)
    for row_base in range(0, 128, 16):
        for kh in range(2):
            addrs = [S.nt16_frag_addr(row_base, kh, l) for l in range(64)]
            got = lds.read_b128(addrs)
            for l in range(64):
                k0 = 32 * kh + 8 * (l >> 4)
                np.testing.assert_array_equal(got[l], ids[row_base + (l & 15), k0:k0 + 8])
            assert S.b128_conflicts(addrs) == 1


def test_gemm_tn16_fragments():
    ids = _ids(64, 128)
    lds = S.Lds(64 * 256)
    for instr in range(16):
        src = []
        for lane in range(64):
            r, ch = S.tn_stage_src(instr, lane)
            src.append(ids[r, ch * 8:ch * 8 + 8])
        lds.dma16(instr * 1024, src)
    worst = 1
    for ncol_base in range(0, 128, 16):
        for ks in range(2):
            for half in range(2):
                addrs = [S.tn16_tr_addr(ncol_base, ks, half, l) for l in range(64)]
                got = lds.read_tr(addrs)
                for l in range(64):
                    n = ncol_base + (l & 15)
                    m0 = 32 * ks + 8 * (l >> 4) + 4 * half
                    np.testing.assert_array_equal(got[l], ids[m0:m0 + 4, n])
                worst = max(worst, S.tr_conflicts(addrs))
    assert worst == 1

