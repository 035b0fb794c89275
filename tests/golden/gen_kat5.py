#!/usr/bin/env python3
"""Round 6: known-answer vectors for the sample kernels that were only covered by whole-stream hashes, recorded from the REAL reference
functions in oracle/_ref/libthorref.so (build container only; `make -C oracle reflib`), at bitdepth 8 (the _lbd instances), 10 and 12 (_hbd):
  intra   make_top_and_left + get_intra_prediction (common/intra_prediction.c:57-183, :403-428), whole blocks and transform-split units
  inter   get_inter_prediction_yuv (common/inter_prediction.c:185-226): clip_mv, luma, eighth-pel chroma, one PU and four quadrant PUs
  avg     average_blocks_all (common/inter_prediction.c:228-247)
  cfl     improve_uv_prediction (common/common_block.c:347-428)
  cdef    cdef_find_dir (common/common_block.c:94-162) and cdef_filter_block[_simd] (:224-279; the SIMD kernel is what the binary executes)
  clpf    detect_multi_clpf (enc/encode_block.c, through oracle/refshim.c) and clpf_block (common/common_block.c:325-345)
  interp  interpolate_frames (common/temporal_interp.c:909)
Stores inputs + outputs in tests/golden/kat5.npz (travels to the GPU box; nothing there reads /root/reference)."""
import ctypes as C, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = C.CDLL(os.path.join(ROOT, 'oracle/_ref/libthorref.so'))
P = lambda a: a.ctypes.data_as(C.c_void_p)


class MV(C.Structure):
    _fields_ = [('x', C.c_int16), ('y', C.c_int16)]


class BlockPos(C.Structure):  # common/types.h:224-232
    _fields_ = [('ypos', C.c_uint16), ('xpos', C.c_uint16), ('size', C.c_uint8), ('bwidth', C.c_uint8), ('bheight', C.c_uint8), ('sb_size', C.c_uint8)]


class YuvFrame(C.Structure):  # common/types.h:58-80
    _fields_ = [('y', C.c_void_p), ('u', C.c_void_p), ('v', C.c_void_p)] + [(n, C.c_int) for n in (
        'width', 'height', 'stride_y', 'stride_c', 'offset_y', 'offset_c', 'pad_hor_y', 'pad_hor_c', 'pad_ver_y', 'pad_ver_c',
        'area_y', 'area_c', 'sub', 'subsample', 'frame_num', 'bitdepth', 'input_bitdepth')]


assert C.sizeof(YuvFrame) == 96 and C.sizeof(BlockPos) == 8


def sfx(bd):
    return '_lbd' if bd == 8 else '_hbd'


def dt(bd):
    return np.uint8 if bd == 8 else np.uint16


def smooth(rng, h, w, bd, noise=6.0):
    """Textured content: low-pass random field + edges + noise, full range of the bitdepth."""
    sc = 1 << (bd - 8)
    base = rng.integers(20, 236, size=(h // 8 + 2, w // 8 + 2)).astype(np.float64)
    img = np.kron(base, np.ones((8, 8)))[:h, :w]
    k = np.ones(5) / 5
    img = np.apply_along_axis(lambda r: np.convolve(r, k, mode='same'), 1, img)
    img = np.apply_along_axis(lambda r: np.convolve(r, k, mode='same'), 0, img)
    img += rng.normal(0, noise, size=(h, w))
    img[:, w // 3] += 40
    return np.ascontiguousarray(np.clip(np.rint(img * sc), 0, (1 << bd) - 1).astype(dt(bd)))   # C order: the reference functions get raw pointers


def frame_planes(fr, bd):
    T = C.c_uint8 if bd == 8 else C.c_uint16
    h, w = fr.height, fr.width
    ya = np.ctypeslib.as_array(C.cast(fr.y, C.POINTER(T)), shape=(h * fr.stride_y,))
    ua = np.ctypeslib.as_array(C.cast(fr.u, C.POINTER(T)), shape=(h // 2 * fr.stride_c,))
    va = np.ctypeslib.as_array(C.cast(fr.v, C.POINTER(T)), shape=(h // 2 * fr.stride_c,))
    Y = ya.reshape(h, fr.stride_y)[:, :w]; U = ua.reshape(h // 2, fr.stride_c)[:, :w // 2]; V = va.reshape(h // 2, fr.stride_c)[:, :w // 2]
    return Y, U, V


def make_frame(bd, w, h, pad, planes=None):
    fr = YuvFrame()
    getattr(L, 'create_yuv_frame' + sfx(bd))(C.byref(fr), w, h, 420, pad, pad, bd, bd)
    if planes is not None:
        Y, U, V = frame_planes(fr, bd)
        Y[:] = planes[0]; U[:] = planes[1]; V[:] = planes[2]
        getattr(L, 'pad_yuv_frame' + sfx(bd))(C.byref(fr))
    return fr


def flat(planes):
    return np.concatenate([p.ravel() for p in planes])


def rec_intra(out, rng, bd):
    T = dt(bd)
    W = H = 96
    plane = smooth(rng, H, W, bd)
    out[f'in{bd}_plane'] = plane
    mk = getattr(L, 'make_top_and_left' + sfx(bd)); gp = getattr(L, 'get_intra_prediction' + sfx(bd))
    k = 0
    for size, tb in ((4, 0), (8, 0), (16, 0), (32, 0), (4, 1), (8, 1), (16, 1)):
        cbs = 2 * size if tb else size
        pars, outs, rbs = [], [], []
        poss = [(0, 0), (0, cbs), (cbs, 0), (cbs, cbs), (2 * cbs if 3 * cbs <= H else cbs, cbs)]
        for (ypos, xpos) in poss:
            for mode in range(11):
                ur = int(rng.integers(0, 2)) if xpos + 2 * cbs <= W and ypos > 0 else 0
                dl = int(rng.integers(0, 2)) if ypos + 2 * cbs <= H and xpos > 0 else 0
                for (i, j) in ([(0, 0), (0, size), (size, 0), (size, size)] if tb else [(0, 0)]):
                    left = np.zeros(2 * 128 + 2, dtype=T); top = np.zeros(2 * 128 + 2, dtype=T)
                    tl = (C.c_uint8 if bd == 8 else C.c_uint16)(0)
                    rb = rng.integers(0, 1 << bd, size=(cbs, cbs)).astype(T)
                    esz = plane.itemsize
                    rec_ptr = C.c_void_p(plane.ctypes.data + esz * (ypos * W + xpos))
                    rb_ptr = C.c_void_p(rb.ctypes.data + esz * (i * cbs + j)) if tb else None
                    mk(C.c_void_p(left.ctypes.data + esz), C.c_void_p(top.ctypes.data + esz), C.byref(tl), rec_ptr, W, rb_ptr, cbs if tb else 0, i, j, ypos, xpos,
                       size, ur, dl, tb, bd)
                    pb = np.zeros((size, size), dtype=T)
                    gp(C.c_void_p(left.ctypes.data + esz), C.c_void_p(top.ctypes.data + esz), tl, ypos + i, xpos + j, size, P(pb), size, mode, bd)
                    pars.append([ypos, xpos, ur, dl, mode, i, j]); outs.append(pb); rbs.append(rb)
        out[f'in{bd}_geo{k}'] = np.array([size, tb], dtype=np.int32)
        out[f'in{bd}_par{k}'] = np.array(pars, dtype=np.int32); out[f'in{bd}_out{k}'] = np.array(outs)
        if tb:
            out[f'in{bd}_rb{k}'] = np.array(rbs)
        k += 1


def rec_inter(out, rng, bd):
    W, H, PAD = 128, 96, 160
    planes = (smooth(rng, H, W, bd), smooth(rng, H // 2, W // 2, bd, 3.0), smooth(rng, H // 2, W // 2, bd, 3.0))
    fr = make_frame(bd, W, H, PAD, planes)
    out[f'ip{bd}_yuv'] = flat(planes)
    fn = getattr(L, 'get_inter_prediction_yuv' + sfx(bd))
    T = dt(bd)
    k = 0
    for size in (8, 16, 32, 64):
        pars, mvs, outs = [], [], []
        for n in range(14):
            ypos = int(rng.integers(0, (H - size) // 8 + 1)) * 8; xpos = int(rng.integers(0, (W - size) // 8 + 1)) * 8
            if n == 0: ypos, xpos = 0, 0
            if n == 1: ypos, xpos = H - size, W - size
            sign, bip = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            split = int(rng.integers(0, 2)) if size >= 16 else 0
            big = n % 4 == 3   # vectors that leave the frame: clip_mv and the padded area
            mv4 = rng.integers(-700 if big else -40, 701 if big else 41, size=(4, 2)).astype(np.int16)
            arr = (MV * 4)(*[MV(int(x), int(y)) for x, y in mv4])
            bp = BlockPos(ypos, xpos, size, size, size, 128)
            py = np.zeros(size * size, dtype=T); pu = np.zeros(size * size // 4, dtype=T); pv = np.zeros(size * size // 4, dtype=T)
            fn(C.byref(fr), P(py), P(pu), P(pv), C.byref(bp), arr, sign, W, H, bip, split, bd)
            pars.append([ypos, xpos, sign, bip, split]); mvs.append(mv4); outs.append(np.concatenate([py, pu, pv]))
        out[f'ip{bd}_size{k}'] = np.array([size], dtype=np.int32)
        out[f'ip{bd}_par{k}'] = np.array(pars, dtype=np.int32); out[f'ip{bd}_mv{k}'] = np.array(mvs); out[f'ip{bd}_out{k}'] = np.array(outs)
        k += 1
    out[f'ip{bd}_geo'] = np.array([W, H], dtype=np.int32)
    # average_blocks_all
    fa = getattr(L, 'average_blocks_all' + sfx(bd))
    for k, size in enumerate((8, 32)):
        n = 4
        a = rng.integers(0, 1 << bd, size=(n, size * size * 3 // 2)).astype(T); b = rng.integers(0, 1 << bd, size=(n, size * size * 3 // 2)).astype(T)
        o = np.zeros_like(a)
        n2, c2 = size * size, size * size // 4
        for i in range(n):
            bp = BlockPos(0, 0, size, size, size, 128)
            esz = a.itemsize
            pp = lambda arr, off: C.c_void_p(arr[i].ctypes.data + esz * off)
            fa(pp(o, 0), pp(o, n2), pp(o, n2 + c2), pp(a, 0), pp(a, n2), pp(a, n2 + c2), pp(b, 0), pp(b, n2), pp(b, n2 + c2), C.byref(bp), 1)
        out[f'av{bd}_a{k}'] = a; out[f'av{bd}_b{k}'] = b; out[f'av{bd}_out{k}'] = o


def rec_cfl(out, rng, bd):
    fn = getattr(L, 'improve_uv_prediction' + sfx(bd))
    T = dt(bd)
    sc = 1 << (bd - 8)
    for k, n in enumerate((8, 16, 32, 64)):
        cnt = 6
        c = n // 2
        ys, uvs, rys, outs = [], [], [], []
        for it in range(cnt):
            y = smooth(rng, n, n, bd, 10.0)
            # reconstructed luma: prediction + a residual that is large for most items (CfL acts only when the luma prediction was poor)
            ry = np.clip(y.astype(np.int64) + rng.normal(0, (2 if it == 0 else 14) * sc, size=(n, n)), 0, (1 << bd) - 1).astype(T)
            ya = (y.reshape(c, 2, c, 2).astype(np.int64).sum(axis=(1, 3)) + 2) >> 2
            u = np.clip(ya * (0.6 if it % 2 else -0.4) + 90 * sc + rng.normal(0, 3 * sc, size=(c, c)), 0, (1 << bd) - 1).astype(T)
            v = np.clip(ya * 0.3 + 40 * sc + rng.normal(0, (30 if it == 2 else 2) * sc, size=(c, c)), 0, (1 << bd) - 1).astype(T)
            uv = np.concatenate([u.ravel(), v.ravel()])
            uo = np.ascontiguousarray(u.copy()); vo = np.ascontiguousarray(v.copy())
            fn(P(np.ascontiguousarray(y)), P(uo), P(vo), P(np.ascontiguousarray(ry)), n, n, n, 1, bd)
            ys.append(y); uvs.append(uv); rys.append(ry); outs.append(np.concatenate([uo.ravel(), vo.ravel()]))
        out[f'cf{bd}_n{k}'] = np.array([n], dtype=np.int32)
        out[f'cf{bd}_y{k}'] = np.array(ys); out[f'cf{bd}_uv{k}'] = np.array(uvs); out[f'cf{bd}_ry{k}'] = np.array(rys); out[f'cf{bd}_out{k}'] = np.array(outs)
        print('cfl', bd, n, 'items changed:', sum(int((a != b).any()) for a, b in zip(uvs, outs)), 'of', cnt)


def rec_cdef(out, rng, bd):
    T = dt(bd)
    cs = bd - 8
    W, H = 64, 48
    plane = smooth(rng, H, W, bd, 8.0)
    out[f'cd{bd}_plane'] = plane
    # directions
    fd = getattr(L, 'cdef_find_dir' + sfx(bd)); fd.restype = C.c_int
    blocks = []
    for by in range(0, H, 8):
        for bx in range(0, W, 8):
            blocks.append(np.ascontiguousarray(plane[by:by + 8, bx:bx + 8]))
    for a in range(8):   # synthetic oriented gratings: every direction is hit
        yy, xx = np.mgrid[0:8, 0:8]
        ang = np.pi * a / 8
        g = 128 + 90 * np.sin((xx * np.cos(ang) + yy * np.sin(ang)) * 1.3)
        blocks.append(np.clip(np.rint(g * (1 << cs)), 0, (1 << bd) - 1).astype(T))
    blocks = np.array(blocks)
    dirs, vars_ = [], []
    for b in blocks:
        v = C.c_int32(0)
        bb = np.ascontiguousarray(b)
        dirs.append(fd(P(bb), 8, C.byref(v), cs)); vars_.append(v.value)
    out[f'cd{bd}_blocks'] = blocks.reshape(len(blocks), 64); out[f'cd{bd}_dir'] = np.array(dirs, dtype=np.int32); out[f'cd{bd}_var'] = np.array(vars_, dtype=np.int32)
    print('cdef dirs', bd, sorted(set(dirs)))
    # filter: the reference reads a uint16 copy with CDEF_VERY_LARGE outside the frame (cdef_prepare_input)
    B = 8   # the SIMD kernels load rows of the block with aligned 16- (8x8) / 8-byte (4x4) loads: 16-byte aligned rows, as cdef_frame's src16
    s16 = W + 2 * B
    raw16 = np.zeros((H + 2 * B) * s16 + 64, dtype=np.uint16)
    o16 = (-raw16.ctypes.data % 64) // 2
    in16 = raw16[o16:o16 + (H + 2 * B) * s16].reshape(H + 2 * B, s16)
    in16[:] = 30000
    in16[B:B + H, B:B + W] = plane
    dx = (C.c_int * 16).in_dll(L, 'cdef_directions_x'); dy = (C.c_int * 16).in_dll(L, 'cdef_directions_y')
    dirs_tab = (C.c_int * 16)(*[dy[i] * s16 + dx[i] for i in range(16)])
    for k, bsize in enumerate((8, 4)):
        pars, outs = [], []
        for n in range(40):
            x0 = int(rng.integers(0, W // bsize)) * bsize; y0 = int(rng.integers(0, H // bsize)) * bsize
            if n < 4: x0, y0 = [(0, 0), (W - bsize, 0), (0, H - bsize), (W - bsize, H - bsize)][n]
            pri = int(rng.integers(0, 16)) << cs; sec = int(rng.choice([0, 1, 2, 4])) << cs
            d = int(rng.integers(0, 8))
            pd = int(rng.integers(3, 7)) + cs; sd = int(rng.integers(3, 7)) + cs
            if pri: pd = max(pd, int(np.log2(pri)))
            if sec: sd = max(sd, int(np.log2(sec)))
            o_simd = np.zeros((bsize, bsize), dtype=T); o_c = np.zeros((bsize, bsize), dtype=T)
            src = C.c_void_p(in16.ctypes.data + 2 * ((B + y0) * s16 + B + x0))
            for fn, o in ((L.cdef_filter_block_simd, o_simd), (L.cdef_filter_block, o_c)):
                if bd == 8: fn(P(o), None, bsize, src, s16, pri, sec, d, pd, sd, bsize, dirs_tab, cs)
                else: fn(None, P(o), bsize, src, s16, pri, sec, d, pd, sd, bsize, dirs_tab, cs)
            assert (o_simd == o_c).all(), 'SIMD and C CDEF kernels differ'
            pars.append([x0, y0, pri, sec, d, pd, sd]); outs.append(o_simd)
        out[f'cd{bd}_fpar{k}'] = np.array(pars, dtype=np.int32); out[f'cd{bd}_fout{k}'] = np.array(outs)
        print('cdef filter', bd, bsize, 'blocks changed:', sum(int((o != plane[p[1]:p[1] + bsize, p[0]:p[0] + bsize]).any()) for p, o in zip(pars, outs)), 'of 40')


def rec_clpf(out, rng, bd):
    """Frame-level expectation built from the reference's block functions: detect_multi_clpf per non-skip 8x8 block (statistics), clpf_block per
    non-skip filter unit with the frame-boundary flags of clpf_frame (common_frame.c:1030-1140: a unit at the frame edge clamps its taps there)."""
    T = dt(bd)
    W, H, qp = 64, 48, 34
    shift = bd - 8
    rec = (smooth(rng, H, W, bd, 9.0), smooth(rng, H // 2, W // 2, bd, 5.0), smooth(rng, H // 2, W // 2, bd, 5.0))
    org = tuple(np.clip(p.astype(np.int64) + rng.normal(0, 5 * (1 << shift), size=p.shape), 0, (1 << bd) - 1).astype(T) for p in rec)
    cw, ch = W // 4, H // 4
    cells = np.zeros((ch, cw, 16), dtype=np.uint8)
    modes = rng.integers(0, 5, size=(ch // 2, cw // 2))
    modes[rng.random(modes.shape) < 0.3] = 0   # M_SKIP = 0 (8x8 granularity)
    cells[:, :, 8] = np.kron(modes, np.ones((2, 2))).astype(np.uint8)
    cells[:, :, 9] = 8
    strength = [2, 1, 4]
    fb_log2 = 5
    nfb_h, nfb_v = (W + 31) // 32, (H + 31) // 32
    fb_on = rng.integers(0, 2, size=nfb_h * nfb_v).astype(np.uint8); fb_on[0] = 1
    det = L.ref_detect_multi_clpf
    blk = getattr(L, 'clpf_block' + sfx(bd))
    stats = []
    outp = [np.ascontiguousarray(p).copy() for p in rec]
    for pl in range(3):
        sub = 1 if pl else 0
        w, h = W >> sub, H >> sub
        r = np.ascontiguousarray(rec[pl]); o = np.ascontiguousarray(org[pl])
        dmp = bd - 4 - sub + (qp >> 4)
        for y0 in range(0, h, 8):
            for x0 in range(0, w, 8):
                if cells[(y0 << sub) // 4, (x0 << sub) // 4, 8] == 0:
                    stats.append([0, 0xffffffff, 0, 0]); continue
                s = (C.c_int * 4)(0, 0, 0, 0)
                if bd == 8:
                    det(P(r), P(o), x0, y0, w, h, w, w, s, 0, 8, dmp, 1)
                else:
                    getattr(L, 'detect_multi_clpf_simd_hbd')(P(r), P(o), x0, y0, w, h, w, w, s, shift, 8, dmp)
                stats.append([s[0], s[1], s[2], s[3]])
        bs = 4 if pl else 8
        for y0 in range(0, h, bs):
            for x0 in range(0, w, bs):
                lin = ((y0 << sub) // 4) * (w // 4) + ((x0 << sub) // 4)   # sic: plane width as row pitch (common_frame.c:1048,1075)
                if cells.reshape(-1, 16)[lin, 8] == 0: continue
                if pl == 0 and not fb_on[(y0 >> fb_log2) * nfb_h + (x0 >> fb_log2)]: continue
                bt = (1 if x0 == 0 else 0) | (2 if x0 + bs == w else 0) | (4 if y0 == 0 else 0) | (8 if y0 + bs == h else 0)   # boundary_type, common/types.h:41-44
                blk(P(r), P(outp[pl]), w, w, x0, y0, bs, bs, bt, strength[pl] << shift, dmp)
    out[f'cl{bd}_par'] = np.array([W, H, qp, fb_log2] + strength, dtype=np.int32)
    out[f'cl{bd}_rec'] = flat(rec); out[f'cl{bd}_org'] = flat(org); out[f'cl{bd}_cells'] = cells; out[f'cl{bd}_fb_on'] = fb_on
    out[f'cl{bd}_stats'] = np.array(stats, dtype=np.uint64).astype(np.uint32); out[f'cl{bd}_out'] = flat(outp)
    print('clpf', bd, 'samples changed:', int((flat(outp) != flat(rec)).sum()))


def rec_interp(out, rng, bd):
    W, H, PAD = 192, 128, 160
    T = dt(bd)
    big = (smooth(rng, H + 16, W + 16, bd, 3.0), smooth(rng, H // 2 + 8, W // 2 + 8, bd, 2.0), smooth(rng, H // 2 + 8, W // 2 + 8, bd, 2.0))
    def cut(dx, dy):
        return (big[0][8 + dy:8 + dy + H, 8 + dx:8 + dx + W], big[1][4 + dy // 2:4 + dy // 2 + H // 2, 4 + dx // 2:4 + dx // 2 + W // 2],
                big[2][4 + dy // 2:4 + dy // 2 + H // 2, 4 + dx // 2:4 + dx // 2 + W // 2])
    p0, p1 = cut(-4, -2), cut(4, 2)   # a pan of (8, 4) samples between the two references
    f0, f1 = make_frame(bd, W, H, PAD, p0), make_frame(bd, W, H, PAD, p1)
    fo = make_frame(bd, W, H, PAD)
    getattr(L, 'interpolate_frames' + sfx(bd))(C.byref(fo), C.byref(f0), C.byref(f1), 2, 1)
    res = [np.ascontiguousarray(p).copy() for p in frame_planes(fo, bd)]
    out[f'it{bd}_geo'] = np.array([W, H], dtype=np.int32)
    out[f'it{bd}_a'] = flat(p0); out[f'it{bd}_b'] = flat(p1); out[f'it{bd}_out'] = flat(res)
    mid = cut(0, 0)
    print('interp', bd, 'mean |out - true middle frame| luma:', float(np.abs(res[0].astype(np.int64) - mid[0]).mean()), ' |avg of refs - middle|:',
          float(np.abs(((p0[0].astype(np.int64) + p1[0]) >> 1) - mid[0]).mean()))


def main():
    L.ref_init(1)
    out = {}
    for bd in (8, 10, 12):
        rng = np.random.default_rng(600 + bd)
        rec_intra(out, rng, bd); rec_inter(out, rng, bd); rec_cfl(out, rng, bd); rec_cdef(out, rng, bd); rec_clpf(out, rng, bd)
        if bd != 12:
            rec_interp(out, rng, bd)
    np.savez_compressed(os.path.join(ROOT, 'tests/golden/kat5.npz'), **out)
    print('wrote kat5.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
