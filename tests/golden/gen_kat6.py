#!/usr/bin/env python3
"""Known answers of widesad_calc (enc/encode_block.c:430-453, file-static: reached through oracle/refshim.c) and of its SIMD kernels widesad_calc_simd_lbd /
_hbd (enc/enc_kernels.c:84-113): the SAD at the five horizontal offsets -3 -1 0 1 3 of a candidate position, the smallest one and its offset (the LEFTMOST among
equal ones).  The motion search evaluates the candidate list of 16x16 coding blocks with it (enc/encode_block.c:545, :570); the device restates it inside
motion_estimate (thor_amd/csrc/tk_me.h: eval_wide), covered by the stream goldens; this file pins the ORACLE's restatement (oracle/thor_oracle.c: orc_widesad,
orc_widesad16).  Build container only (`make -C oracle reflib`); writes tests/golden/kat6.npz.  Content: random, flat (every offset ties), period-2 and period-4
columns (ties between some offsets), a ramp."""
import ctypes as C, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.path.join(ROOT, 'oracle/_ref/libthorref.so'))


def aligned(shape, dtype, align=64):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.zeros(n + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].view(dtype).reshape(shape)


def planes(rng, maxv, dtype):
    H, W = 64, 96
    out = []
    x = np.arange(W)[None, :] + np.zeros((H, 1), dtype=np.int64)
    y = np.arange(H)[:, None] + np.zeros((1, W), dtype=np.int64)
    out.append(rng.integers(0, maxv + 1, size=(H, W)))
    out.append(np.full((H, W), maxv // 3))
    out.append(np.where(x % 2 == 0, maxv // 4, 3 * maxv // 4))
    out.append(np.where(x % 4 < 2, 0, maxv))
    out.append((x * 5 + y * 3) % (maxv + 1))
    return [np.ascontiguousarray(p.astype(dtype)) for p in out]


def main():
    rng = np.random.default_rng(60606)
    out = {}
    # 8-bit: the file-static dispatcher with use_simd = 0 (scalar loop, every size) and 1 (16x16 -> SIMD kernel): must agree
    pl = planes(rng, 255, np.uint8)
    out['ws8_planes'] = np.stack(pl)
    xv = C.c_int()
    k = 0
    for pi, plane in enumerate(pl):
        for (w, h) in ((16, 16), (8, 8), (16, 8), (8, 16), (32, 32), (4, 4)):
            org = aligned((h, max(w, 16)), np.uint8)
            src = pl[(pi + 1) % len(pl)] if pi % 2 else plane   # original = another plane, or a shifted copy of the same one (exact matches at some offset)
            cands = rng.integers(0, 24, size=(6, 2))
            for (cx, cy) in cands:
                oy, ox = int(rng.integers(0, 64 - h)), int(rng.integers(4, 96 - w - 4))
                org[:, :w] = src[oy:oy + h, ox:ox + w]
                by, bx = int(cy) % (64 - h), 4 + int(cx) % (96 - w - 8)
                res = []
                for simd in (0, 1):
                    L.ref_init(simd)
                    s = L.ref_widesad_calc(org.ctypes.data_as(C.c_void_p), C.c_void_p(plane.ctypes.data + by * 96 + bx), org.shape[1], 96, w, h, C.byref(xv))
                    res.append((s & 0xffffffff, xv.value))
                assert res[0] == res[1], ('scalar and SIMD widesad differ', res, w, h)
                out[f'ws8_org{k}'] = org[:, :w].copy()
                out[f'ws8_arg{k}'] = np.array([pi, by, bx, w, h], dtype=np.int32)
                out[f'ws8_out{k}'] = np.array([res[1][0], res[1][1]], dtype=np.int64)
                k += 1
    out['ws8_n'] = np.array(k)
    L.ref_init(1)
    # 16-bit samples (bitdepth 10 and 12): the SIMD kernel (16x16 only; the scalar loop is the same code as above compiled for 16-bit samples)
    L.widesad_calc_simd_hbd.restype = C.c_uint
    k = 0
    for bd in (10, 12):
        pl = planes(rng, (1 << bd) - 1, np.uint16)
        out[f'ws16_planes_bd{bd}'] = np.stack(pl)
        for pi, plane in enumerate(pl):
            org = aligned((16, 16), np.uint16)
            src = pl[(pi + 1) % len(pl)] if pi % 2 else plane
            for _ in range(8):
                oy, ox = int(rng.integers(0, 48)), int(rng.integers(4, 72))
                org[:, :] = src[oy:oy + 16, ox:ox + 16]
                by, bx = int(rng.integers(0, 48)), int(rng.integers(4, 72))
                s = L.widesad_calc_simd_hbd(org.ctypes.data_as(C.c_void_p), C.c_void_p(plane.ctypes.data + 2 * (by * 96 + bx)), 16, 96, 16, 16, C.byref(xv))
                out[f'ws16_org{k}'] = org.copy()
                out[f'ws16_arg{k}'] = np.array([bd, pi, by, bx], dtype=np.int32)
                out[f'ws16_out{k}'] = np.array([s, xv.value], dtype=np.int64)
                k += 1
    out['ws16_n'] = np.array(k)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'kat6.npz'), **out)
    offs8 = [int(out[f'ws8_out{i}'][1]) for i in range(int(out['ws8_n']))]
    print('kat6.npz:', int(out['ws8_n']), '8-bit and', int(out['ws16_n']), '16-bit vectors; offsets chosen (8-bit):', {o: offs8.count(o) for o in sorted(set(offs8))})


if __name__ == '__main__':
    main()
