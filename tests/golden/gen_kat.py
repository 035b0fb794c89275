#!/usr/bin/env python3
"""Generate known-answer vectors for the kernel-level arithmetic from the REAL reference code.

Runs only in the build container (needs /root/reference): `make -C oracle reflib` builds
oracle/_ref/libthorref.so from the reference sources where they lie; this script calls the reference
functions through ctypes on seeded random inputs and stores inputs + outputs in tests/golden/kat.npz.
The committed .npz travels to the GPU box; nothing there reads /root/reference."""
import ctypes as C, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.path.join(ROOT, 'oracle/_ref/libthorref.so'))
P = lambda a: a.ctypes.data_as(C.c_void_p)

class MV(C.Structure):
    _fields_ = [('x', C.c_int16), ('y', C.c_int16)]

def main():
    rng = np.random.default_rng(1234)
    out = {}
    # --- SAD (sad_calc, encode_block.c:417); scalar C and SIMD paths must agree
    plane = rng.integers(0, 256, size=(96, 96), dtype=np.uint8)
    out['sad_plane'] = plane
    for i, (w, h) in enumerate(((4, 4), (8, 8), (16, 8), (8, 16), (16, 16), (32, 32), (64, 64))):
        org = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
        cand = rng.integers(-8, 9, size=(24, 2)).astype(np.int32)
        both = []
        for simd in (0, 1):
            L.ref_init(simd)
            both.append(np.array([L.ref_sad_calc(P(org), C.c_void_p(int(plane.ctypes.data) + (12 + int(dy)) * 96 + 12 + int(dx)), w, 96, w, h)
                                  for dx, dy in cand], dtype=np.uint32))
        assert (both[0] == both[1]).all(), 'C and SIMD SAD differ'
        out[f'sad_org{i}'] = org; out[f'sad_cand{i}'] = cand; out[f'sad_out{i}'] = both[1]
    L.ref_init(1)
    # --- mv bit estimate (quote_mv_bits, encode_block.c:467)
    d = rng.integers(-300, 301, size=(200, 2)).astype(np.int32)
    out['mvb_in'] = d
    out['mvb_out'] = np.array([L.ref_quote_mv_bits(int(a), int(b)) for a, b in d], dtype=np.int32)
    # --- quarter-pel luma (get_inter_prediction_luma_lbd, inter_prediction.c:117)
    pad = 16
    pw, ph = 64, 48
    ref = rng.integers(0, 256, size=(ph + 2 * pad, pw + 2 * pad), dtype=np.uint8)
    out['ip_ref'] = ref
    k = 0
    for (w, h, bx, by) in ((8, 8, 8, 8), (16, 16, 24, 16), (4, 4, 0, 0), (32, 16, 32, 24)):
        mvs = rng.integers(-23, 24, size=(20, 2)).astype(np.int16)
        for bip in (0, 1):
            res = np.zeros((len(mvs), h, w), dtype=np.uint8)
            for i, (mx, my) in enumerate(mvs):
                mv = MV(int(mx), int(my))
                base = int(ref.ctypes.data) + (pad + by) * int(ref.shape[1]) + pad + bx
                L.get_inter_prediction_luma_lbd(P(res[i]), C.c_void_p(base), w, h, ref.shape[1], w, C.byref(mv), 0, bip, pw, ph, bx, by, 8)
            out[f'ip_geo{k}'] = np.array([w, h, bx, by, bip], dtype=np.int32); out[f'ip_mv{k}'] = mvs; out[f'ip_out{k}'] = res
            k += 1
    # --- TU pipeline: transform / quantize / dequantize / inverse_transform / reconstruct_block
    k = 0
    for size in (4, 8, 16, 32, 64, 128):
        for (qp, ctype, fast) in ((22, 0, 0), (32, 2, 0), (38, 1, 0), (30, 0, 1 if size in (32, 64) else 0)):
            n = 6 if size <= 32 else 2
            q = min(size, 16)
            org = rng.integers(0, 256, size=(n, size, size), dtype=np.uint8)
            smooth = (org.astype(np.int32) + rng.integers(-20, 21, size=org.shape)).clip(0, 255)
            pred = np.where(rng.random(org.shape) < 0.5, smooth, np.roll(smooth, 1, axis=2)).astype(np.uint8)
            if k % 3 == 0: pred = rng.integers(0, 256, size=org.shape, dtype=np.uint8)   # large residuals
            coefq = np.zeros((n, q, q), dtype=np.int16); rec = np.zeros_like(org); cbp = np.zeros(n, dtype=np.int32)
            fwd = np.zeros((n, q, q), dtype=np.int16); bits = np.zeros(n, dtype=np.int32)
            for i in range(n):
                res16 = (org[i].astype(np.int16) - pred[i].astype(np.int16))
                coeff = np.zeros((size, size), dtype=np.int16)
                L.transform(P(res16), P(coeff), size, fast, 8)
                fwd[i] = coeff[:q, :q]
                cq = np.zeros((q, q), dtype=np.int16)
                cbp[i] = L.ref_quantize(P(coeff), P(cq), qp, size, ctype)
                coefq[i] = cq
                if cbp[i]:
                    rco = np.zeros((size, size), dtype=np.int16); rb = np.zeros((size, size), dtype=np.int16)
                    L.dequantize_lbd(P(cq), P(rco), qp, size, None)
                    L.inverse_transform(P(rco), P(rb), size, 8)
                    L.reconstruct_block_lbd(P(rb), P(pred[i]), P(rec[i]), size, size, size, 8)
                    if size <= 16 * 8:   # write_coeff wants qsize-compact input
                        bits[i] = L.ref_coeff_bits(P(cq), size, ctype | 0)
                else:
                    rec[i] = pred[i]
            out[f'tu_par{k}'] = np.array([size, qp, ctype, fast], dtype=np.int32)
            out[f'tu_org{k}'] = org; out[f'tu_pred{k}'] = pred; out[f'tu_fwd{k}'] = fwd
            out[f'tu_coefq{k}'] = coefq; out[f'tu_rec{k}'] = rec; out[f'tu_cbp{k}'] = cbp; out[f'tu_bits{k}'] = bits
            k += 1
    np.savez_compressed(os.path.join(ROOT, 'tests/golden/kat.npz'), **out)
    print('wrote kat.npz with', len(out), 'arrays')

if __name__ == '__main__':
    main()
