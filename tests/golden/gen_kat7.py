#!/usr/bin/env python3
"""Known answers of the two sub-block tests of the early-skip check (enc/encode_block.c:2146-2229, file-static: reached through oracle/refshim.c): luma
check_early_skip_sub_block (2x2 average + half-size transform against half the threshold) and chroma check_early_skip_sub_blockC with use_simd = 1, i.e.
calc_cbp_simd (enc/enc_kernels.c:828-907) - the form the encoder executes, whose 4-wide case differs from the scalar calc_cbp.  8-bit samples (the shim is the
reference's 8-bit build).  Sizes as the block path calls them: luma 8 / 16 / 32, chroma 4 / 8 / 16; qp 22 / 32 / 42, early_skip_thr 0.3 / 1.0 (the operating
points' values); residuals around the decision boundary: flat offsets, single-column offsets, low-amplitude noise, sign-alternating columns.
Build container only (`make -C oracle reflib`); writes tests/golden/kat7.npz.  Pins oracle/thor_oracle.c (orc_early_skip_sub / _subC, tests/test_oracle_c.py)
and the host build of the device functions early_skip_sub / early_skip_subC (tests/hostsim/kat_host_es.cpp, tests/test_kat_host.py)."""
import ctypes as C, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.path.join(ROOT, 'oracle/_ref/libthorref.so'))
L.ref_early_skip_sub.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int, C.c_int]


def aligned(shape, dtype, align=64):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.zeros(n + align, dtype=np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].view(dtype).reshape(shape)


def main():
    rng = np.random.default_rng(70707)
    orgs, preds, args, outs = [], [], [], []
    for chroma, sizes in ((0, (8, 16, 32)), (1, (4, 8, 16))):
        for size in sizes:
            for qp in (22, 32, 42):
                for thr in (0.3, 1.0):
                    for kind in range(8):
                        base = (40 + 3 * np.arange(size)[None, :] + 2 * np.arange(size)[:, None] + rng.integers(0, 4, size=(size, size))).astype(np.int32)
                        amp = int(rng.integers(1, 12))
                        if kind == 0: d = np.full((size, size), int(rng.integers(-amp, amp + 1)))                     # flat offset
                        elif kind == 1: d = rng.integers(-amp, amp + 1, size=(size, size))                             # noise
                        elif kind == 2: d = np.zeros((size, size), np.int64); d[:, int(rng.integers(0, size))] = int(rng.integers(-2 * amp, 2 * amp + 1))   # one column
                        elif kind == 3: d = np.where(np.arange(size)[None, :] % 2 == 0, amp, -amp) + np.zeros((size, 1), np.int64)                        # alternating columns
                        elif kind == 4: d = np.where(np.arange(size)[None, :] % 2 == 0, -amp, amp - 1) + np.zeros((size, 1), np.int64)                    # pair (negative, positive)
                        elif kind == 5: d = np.zeros((size, size), np.int64); d[int(rng.integers(0, size)), :] = int(rng.integers(-3 * amp, 3 * amp + 1)) # one row
                        elif kind == 6: d = (np.arange(size)[None, :] + np.arange(size)[:, None]) % (amp + 1) - amp // 2                                  # ramp
                        else: d = np.zeros((size, size), np.int64)                                                                                      # identical
                        org = aligned((size, size), np.uint8); pred = aligned((size, size), np.uint8)
                        org[:] = np.clip(base + d, 0, 255).astype(np.uint8)
                        pred[:] = np.clip(base, 0, 255).astype(np.uint8)
                        r = L.ref_early_skip_sub(chroma, org.ctypes.data_as(C.c_void_p), size, size, qp, pred.ctypes.data_as(C.c_void_p), thr, 8, 1)
                        o = np.zeros((32, 32), np.uint8); p = np.zeros((32, 32), np.uint8)
                        o[:size, :size] = org; p[:size, :size] = pred
                        orgs.append(o); preds.append(p); args.append((chroma, size, qp, int(round(thr * 10)))); outs.append(r)
    out = {'es_org': np.stack(orgs), 'es_pred': np.stack(preds), 'es_arg': np.array(args, dtype=np.int32), 'es_out': np.array(outs, dtype=np.int32)}
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'kat7.npz'), **out)
    a, o = out['es_arg'], out['es_out']
    print('kat7.npz:', len(o), 'vectors;', 'significant:', {('chroma' if c else 'luma', s): f'{int(o[(a[:, 0] == c) & (a[:, 1] == s)].sum())}/{int(((a[:, 0] == c) & (a[:, 1] == s)).sum())}'
                                                        for c, ss in ((0, (8, 16, 32)), (1, (4, 8, 16))) for s in ss})


if __name__ == '__main__':
    main()
