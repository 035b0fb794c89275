#!/usr/bin/env python3
"""Known-answer vectors for the in-loop deblocking filter from the REAL reference functions deblock_frame_y_lbd /
deblock_frame_uv_lbd (common/common_frame.c:47,354) in oracle/_ref/libthorref.so (build container only).  Random frames
with block structure, random per-4x4 block data (modes, cbp, sizes, tb/pb splits, motion vectors); stores the inputs in
the library's compact cell form plus the filtered frames in tests/golden/kat3.npz."""
import ctypes as C, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.path.join(ROOT, 'oracle/_ref/libthorref.so'))


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
    rng = np.random.default_rng(77)
    out = {}
    L.ref_init(1)
    for k, (w, h, qp) in enumerate(((64, 48, 32), (128, 80, 38), (96, 64, 22), (64, 64, 51), (80, 48, 27))):
        # blocky content: piecewise-constant 8x8 tiles + noise, so that edges are in the range where the filter switches
        tiles = rng.integers(60, 200, size=(h // 8, w // 8))
        Y = np.clip(np.kron(tiles, np.ones((8, 8))) + rng.normal(0, 3, size=(h, w)) + rng.integers(-6, 7, size=(h, 1)), 0, 255).astype(np.uint8)
        tc = rng.integers(90, 170, size=(h // 16, w // 16))
        U = np.clip(np.kron(tc, np.ones((8, 8))) + rng.normal(0, 2, size=(h // 2, w // 2)), 0, 255).astype(np.uint8)
        V = np.clip(np.kron(tc[::-1], np.ones((8, 8))) + rng.normal(0, 2, size=(h // 2, w // 2)), 0, 255).astype(np.uint8)
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
        fr.sub, fr.subsample, fr.bitdepth, fr.input_bitdepth = 1, 420, 8, 8
        chroma_qp = list((C.c_int * 52).in_dll(L, 'chroma_qp'))  # common/common_tables.c:68-72
        L.deblock_frame_y_lbd(C.byref(fr), dd, w, h, C.c_uint8(qp), 8)
        L.deblock_frame_uv_lbd(C.byref(fr), dd, w, h, C.c_uint8(chroma_qp[qp]), 8)   # enc/encode_frame.c:760-764
        out[f'db_par{k}'] = np.array([w, h, qp], dtype=np.int32)
        out[f'db_in{k}'] = np.concatenate([Y.ravel(), U.ravel(), V.ravel()])
        out[f'db_cells{k}'] = cells
        out[f'db_out{k}'] = np.concatenate([yo.ravel(), uo.ravel(), vo.ravel()])
        print(k, 'changed samples:', int((yo != Y).sum()), int((uo != U).sum()), int((vo != V).sum()))
    np.savez_compressed(os.path.join(ROOT, 'tests/golden/kat3.npz'), **out)


if __name__ == '__main__':
    main()
