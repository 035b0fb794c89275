#!/usr/bin/env python3
"""Known-answer vectors for the low/medium-complexity kernels (encoder_speed > 0 sub-pel approximations, CLPF) from
the REAL reference code, via oracle/_ref/libthorref.so (oracle/refshim.c).  Both the scalar C and the SIMD variant
of every function are run and must agree.  Output: tests/golden/kat2.npz (committed; the GPU box never reads
/root/reference)."""
import ctypes as C, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.path.join(ROOT, 'oracle/_ref/libthorref.so'))
P = lambda a: a.ctypes.data_as(C.c_void_p)

def main():
    rng = np.random.default_rng(4321)
    out = {}
    L.ref_init(1)
    # --- sad_calc_fasthalf / sad_calc_fastquarter (encode_block.c:174, :286)
    plane = rng.integers(0, 256, size=(112, 112), dtype=np.uint8)
    plane[40:80, 30:90] = (plane[40:80, 30:90] // 8 + 100).astype(np.uint8)   # a smooth patch: close SADs, exercises ties
    out['fs_plane'] = plane
    k = 0
    for (w, h) in ((8, 8), (16, 16), (16, 8), (8, 16), (32, 32), (64, 64)):
        for rep in range(6):
            bx, by = int(rng.integers(8, 112 - w - 8)), int(rng.integers(8, 112 - h - 8))
            org = (plane[by:by + h, bx:bx + w].astype(np.int32) + rng.integers(-6, 7, size=(h, w))).clip(0, 255).astype(np.uint8)
            org = np.ascontiguousarray(org)
            base = C.c_void_p(int(plane.ctypes.data) + by * 112 + bx)
            res = []
            for simd in (0, 1):
                x, y = C.c_int(0), C.c_int(0)
                s = L.ref_fasthalf(P(org), base, w, 112, w, h, C.byref(x), C.byref(y), simd)
                res.append((s, x.value, y.value))
            assert res[0] == res[1], ('fasthalf C vs SIMD', res)
            qres = []
            for (hx, hy) in ((0, 0), (2, 0), (0, -2), (-2, 2)):
                two = []
                for simd in (0, 1):
                    x, y = C.c_int(hx), C.c_int(hy)
                    s = L.ref_fastquarter(P(org), base, w, 112, w, h, C.byref(x), C.byref(y), simd)
                    two.append((s, x.value, y.value))
                assert two[0] == two[1], ('fastquarter C vs SIMD', two)
                qres.append([hx, hy, *two[1]])
            out[f'fs_geo{k}'] = np.array([w, h, bx, by], dtype=np.int32); out[f'fs_org{k}'] = org
            out[f'fs_half{k}'] = np.array(res[1], dtype=np.int64); out[f'fs_quarter{k}'] = np.array(qres, dtype=np.int64)
            k += 1
    # --- clpf_sample (common_block.c:315) on random neighbourhoods
    v = rng.integers(0, 256, size=(400, 9)).astype(np.int32)
    v[:200, 1:] = (v[:200, :1] + rng.integers(-12, 13, size=(200, 8))).clip(0, 255)   # small differences: the constrained range
    par = np.stack([rng.choice([1, 2, 4], size=400), rng.integers(4, 8, size=400)], axis=1).astype(np.int32)
    out['cs_in'] = v; out['cs_par'] = par
    out['cs_out'] = np.array([L.ref_clpf_sample(*[int(t) for t in row], int(s), int(d)) for row, (s, d) in zip(v, par)], dtype=np.int32)
    # --- detect_multi_clpf (encode_block.c:2584) incl. frame-edge blocks
    W, H = 64, 48
    org = rng.integers(0, 256, size=(H, W), dtype=np.uint8)
    rec = (org.astype(np.int32) + rng.integers(-9, 10, size=(H, W))).clip(0, 255).astype(np.uint8)
    out['dm_org'] = org; out['dm_rec'] = rec
    blocks, sums = [], []
    for (x0, y0) in ((0, 0), (56, 0), (0, 40), (56, 40), (24, 16), (8, 32)):
        for dmp in (4, 6):
            two = []
            for simd in (0, 1):
                s = (C.c_int * 4)(0, 0, 0, 0)
                L.ref_detect_multi_clpf(P(rec), P(org), x0, y0, W, H, W, W, s, 0, 8, dmp, simd)
                two.append(list(s))
            assert two[0] == two[1], ('detect_multi_clpf C vs SIMD', two)
            blocks.append([x0, y0, dmp]); sums.append(two[1])
    out['dm_blocks'] = np.array(blocks, dtype=np.int32); out['dm_sums'] = np.array(sums, dtype=np.int64)
    np.savez_compressed(os.path.join(ROOT, 'tests/golden/kat2.npz'), **out)
    print('wrote kat2.npz with', len(out), 'arrays')

if __name__ == '__main__':
    main()
