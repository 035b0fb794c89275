#!/usr/bin/env python3
"""Known-answer vectors for the 10-bit (uint16_t, `_hbd`) kernels from the REAL reference functions in
oracle/_ref/libthorref.so (build container only; `make -C oracle reflib`): sad_calc_simd_hbd (enc/enc_kernels.c:36 compiled
through enc_kernels_hbd.c), get_inter_prediction_luma_hbd (common/inter_prediction.c:117), transform / quantize /
dequantize_hbd / inverse_transform / reconstruct_block_hbd (common/transform.c:245,411, enc/encode_block.c:84,
common/common_block.c:45,75) and deblock_frame_y_hbd / deblock_frame_uv_hbd (common/common_frame.c:47,354), all with
bitdepth 10 (or the bitdepth given as argument).  Stores inputs + outputs in tests/golden/kat4.npz / kat4_<bitdepth>.npz (travels to the GPU box; nothing there reads /root/reference)."""
import ctypes as C, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.path.join(ROOT, 'oracle/_ref/libthorref.so'))
P = lambda a: a.ctypes.data_as(C.c_void_p)
BD = int(sys.argv[1]) if len(sys.argv) > 1 else 10   # round 6: `gen_kat4.py 12` -> kat4_12.npz (the same families at bitdepth 12)
MAXV = (1 << BD) - 1
SC = 1 << (BD - 10)          # content amplitudes were chosen for 10 bits


class MV(C.Structure):
    _fields_ = [('x', C.c_int16), ('y', C.c_int16)]


class InterPred(C.Structure):  # common/types.h:138-145
    _fields_ = [('mv0', MV), ('mv1', MV), ('ref_idx0', C.c_uint32), ('ref_idx1', C.c_uint32), ('bipred_flag', C.c_uint32)]


class DeblockData(C.Structure):  # common/types.h:178-187
    _fields_ = [('mode', C.c_int), ('cbp_y', C.c_int), ('cbp_u', C.c_int), ('cbp_v', C.c_int), ('size', C.c_uint8), ('tb_split', C.c_uint8),
                ('pb_part', C.c_int), ('inter_pred', InterPred), ('inter_pred_arr', InterPred * 16)]


class YuvFrame(C.Structure):  # common/types.h:58-80
    _fields_ = [('y', C.c_void_p), ('u', C.c_void_p), ('v', C.c_void_p)] + [(n, C.c_int) for n in (
        'width', 'height', 'stride_y', 'stride_c', 'offset_y', 'offset_c', 'pad_hor_y', 'pad_hor_c', 'pad_ver_y', 'pad_ver_c',
        'area_y', 'area_c', 'sub', 'subsample', 'frame_num', 'bitdepth', 'input_bitdepth')]


assert C.sizeof(DeblockData) == 364 and C.sizeof(YuvFrame) == 96


def main():
    rng = np.random.default_rng(4321)
    out = {}
    L.ref_init(1)
    L.sad_calc_simd_hbd.restype = C.c_int
    # --- SAD
    plane = rng.integers(0, MAXV + 1, size=(96, 96), dtype=np.uint16)
    out['sad_plane'] = plane
    # the SIMD kernel serves widths >= 8 (sad_calc falls back to scalar code for 4, enc/encode_block.c:417-428) and loads the
    # original block with aligned vector loads
    for i, (w, h) in enumerate(((8, 8), (16, 8), (8, 16), (16, 16), (32, 32), (64, 64))):
        raw = np.zeros(h * w + 32, dtype=np.uint16)
        o = (-raw.ctypes.data % 64) // 2
        org = raw[o:o + h * w].reshape(h, w)
        org[:] = rng.integers(0, MAXV + 1, size=(h, w), dtype=np.uint16)
        cand = rng.integers(-8, 9, size=(24, 2)).astype(np.int32)
        res = np.array([L.sad_calc_simd_hbd(P(org), C.c_void_p(int(plane.ctypes.data) + 2 * ((12 + int(dy)) * 96 + 12 + int(dx))), w, 96, w, h)
                        for dx, dy in cand], dtype=np.uint32)
        out[f'sad_org{i}'] = org.copy(); out[f'sad_cand{i}'] = cand; out[f'sad_out{i}'] = res
    # --- quarter-pel luma
    pad, pw, ph = 16, 64, 48
    ref = rng.integers(0, MAXV + 1, size=(ph + 2 * pad, pw + 2 * pad), dtype=np.uint16)
    out['ip_ref'] = ref
    k = 0
    for (w, h, bx, by) in ((8, 8, 8, 8), (16, 16, 24, 16), (4, 4, 0, 0), (32, 16, 32, 24)):
        mvs = rng.integers(-23, 24, size=(20, 2)).astype(np.int16)
        for bip in (0, 1):
            res = np.zeros((len(mvs), h, w), dtype=np.uint16)
            for i, (mx, my) in enumerate(mvs):
                mv = MV(int(mx), int(my))
                base = int(ref.ctypes.data) + 2 * ((pad + by) * int(ref.shape[1]) + pad + bx)
                L.get_inter_prediction_luma_hbd(P(res[i]), C.c_void_p(base), w, h, ref.shape[1], w, C.byref(mv), 0, bip, pw, ph, bx, by, BD)
            out[f'ip_geo{k}'] = np.array([w, h, bx, by, bip], dtype=np.int32); out[f'ip_mv{k}'] = mvs; out[f'ip_out{k}'] = res
            k += 1
    # --- TU pipeline
    k = 0
    for size in (4, 8, 16, 32, 64, 128):
        for (qp, ctype, fast) in ((22, 0, 0), (32, 2, 0), (38, 1, 0), (30, 0, 1 if size in (32, 64) else 0)):
            n = (6 if size <= 32 else 2) if BD == 10 else (3 if size <= 32 else 1)
            q = min(size, 16)
            org = rng.integers(0, MAXV + 1, size=(n, size, size), dtype=np.uint16)
            smooth = (org.astype(np.int32) + rng.integers(-80 * SC, 80 * SC + 1, size=org.shape)).clip(0, MAXV)
            pred = np.where(rng.random(org.shape) < 0.5, smooth, np.roll(smooth, 1, axis=2)).astype(np.uint16)
            if k % 3 == 0: pred = rng.integers(0, MAXV + 1, size=org.shape, dtype=np.uint16)   # large residuals
            coefq = np.zeros((n, q, q), dtype=np.int16); rec = np.zeros_like(org); cbp = np.zeros(n, dtype=np.int32)
            for i in range(n):
                res16 = (org[i].astype(np.int32) - pred[i].astype(np.int32)).astype(np.int16)
                coeff = np.zeros((size, size), dtype=np.int16)
                L.transform(P(res16), P(coeff), size, fast, BD)
                cq = np.zeros((q, q), dtype=np.int16)
                cbp[i] = L.ref_quantize(P(coeff), P(cq), qp, size, ctype)
                coefq[i] = cq
                if cbp[i]:
                    rco = np.zeros((size, size), dtype=np.int16); rb = np.zeros((size, size), dtype=np.int16)
                    L.dequantize_hbd(P(cq), P(rco), qp, size, None)
                    L.inverse_transform(P(rco), P(rb), size, BD)
                    L.reconstruct_block_hbd(P(rb), P(pred[i]), P(rec[i]), size, size, size, BD)
                else:
                    rec[i] = pred[i]
            out[f'tu_par{k}'] = np.array([size, qp, ctype, fast], dtype=np.int32)
            out[f'tu_org{k}'] = org; out[f'tu_pred{k}'] = pred
            out[f'tu_coefq{k}'] = coefq; out[f'tu_rec{k}'] = rec; out[f'tu_cbp{k}'] = cbp
            k += 1
    # --- deblocking
    chroma_qp = list((C.c_int * 52).in_dll(L, 'chroma_qp'))  # common/common_tables.c:68-72
    for k, (w, h, qp) in enumerate(((64, 48, 32), (128, 80, 38), (96, 64, 22))):
        tiles = rng.integers(240 * SC, 800 * SC, size=(h // 8, w // 8))
        Y = np.clip(np.kron(tiles, np.ones((8, 8))) + rng.normal(0, 12 * SC, size=(h, w)) + rng.integers(-24 * SC, 24 * SC + 1, size=(h, 1)), 0, MAXV).astype(np.uint16)
        tc = rng.integers(360 * SC, 680 * SC, size=(h // 16, w // 16))
        U = np.clip(np.kron(tc, np.ones((8, 8))) + rng.normal(0, 8 * SC, size=(h // 2, w // 2)), 0, MAXV).astype(np.uint16)
        V = np.clip(np.kron(tc[::-1], np.ones((8, 8))) + rng.normal(0, 8 * SC, size=(h // 2, w // 2)), 0, MAXV).astype(np.uint16)
        ch, cw = h // 4, w // 4
        dd = (DeblockData * (ch * cw))()
        cells = np.zeros((ch, cw, 16), dtype=np.uint8)
        cv = cells.view(np.int16)
        for i in range(ch):
            for j in range(cw):
                d = dd[i * cw + j]
                d.mode = int(rng.integers(0, 5))
                d.cbp_y, d.cbp_u, d.cbp_v = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
                d.size = int(rng.choice([8, 8, 16, 16, 32, 64]))
                d.tb_split = int(rng.integers(0, 2))
                d.pb_part = int(rng.integers(0, 4))
                mvs = rng.integers(-6, 7, size=4) * int(rng.integers(0, 2))
                d.inter_pred.mv0.x, d.inter_pred.mv0.y, d.inter_pred.mv1.x, d.inter_pred.mv1.y = (int(v) for v in mvs)
                d.inter_pred.ref_idx0, d.inter_pred.ref_idx1, d.inter_pred.bipred_flag = int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(0, 3))
                cv[i, j, 0:4] = mvs
                cells[i, j, 8] = d.mode; cells[i, j, 9] = d.size; cells[i, j, 10] = d.tb_split | (d.pb_part << 1)
                cells[i, j, 11] = d.cbp_y | (d.cbp_u << 1) | (d.cbp_v << 2)
                cells[i, j, 12] = d.inter_pred.ref_idx0; cells[i, j, 13] = d.inter_pred.ref_idx1; cells[i, j, 14] = d.inter_pred.bipred_flag
        yo, uo, vo = Y.copy(), U.copy(), V.copy()
        fr = YuvFrame()
        fr.y, fr.u, fr.v = yo.ctypes.data, uo.ctypes.data, vo.ctypes.data
        fr.width, fr.height, fr.stride_y, fr.stride_c = w, h, w, w // 2
        fr.sub, fr.subsample, fr.bitdepth, fr.input_bitdepth = 1, 420, BD, BD
        L.deblock_frame_y_hbd(C.byref(fr), dd, w, h, C.c_uint8(qp), BD)
        L.deblock_frame_uv_hbd(C.byref(fr), dd, w, h, C.c_uint8(chroma_qp[qp]), BD)
        out[f'db_par{k}'] = np.array([w, h, qp], dtype=np.int32)
        out[f'db_in{k}'] = np.concatenate([Y.ravel(), U.ravel(), V.ravel()])
        out[f'db_cells{k}'] = cells
        out[f'db_out{k}'] = np.concatenate([yo.ravel(), uo.ravel(), vo.ravel()])
        print('deblock', k, 'changed samples:', int((yo != Y).sum()), int((uo != U).sum()), int((vo != V).sum()))
    name = 'kat4.npz' if BD == 10 else f'kat4_{BD}.npz'
    out['bitdepth'] = np.array([BD], dtype=np.int32)
    np.savez_compressed(os.path.join(ROOT, 'tests/golden', name), **out)
    print('wrote', name, 'with', len(out), 'arrays')


if __name__ == '__main__':
    main()
