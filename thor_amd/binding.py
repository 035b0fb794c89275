"""ctypes binding of libthor_hip.so (include/thor_hip.h)."""
import ctypes as C
import os
import subprocess
import numpy as np

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


def lib_path():
    """In-tree library; THOR_HIP_LIB selects another build of it (A/B measurements of kernel variants)."""
    return os.environ.get('THOR_HIP_LIB') or os.path.join(REPO_ROOT, 'thor_amd', 'libthor_hip.so')


HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-strict-aliasing', '-fPIC', '-pthread']
NATIVE_SOURCES = ['thor_hip.cpp', 'thor_hip_lat.cpp', 'thor_hip_wide.cpp']   # the throughput build of the engine + its two builds for the few-stream operating points


def build_native(force=False):
    """Compile libthor_hip.so (gfx950) and the C front end in-tree with hipcc/gcc: the two translation units are compiled side by side, then linked."""
    csrc = os.path.join(REPO_ROOT, 'thor_amd', 'csrc')
    out = os.path.join(REPO_ROOT, 'thor_amd', 'libthor_hip.so')
    hdrs = [os.path.join(csrc, f) for f in os.listdir(csrc)]
    hdrs += [os.path.join(REPO_ROOT, 'include', f) for f in os.listdir(os.path.join(REPO_ROOT, 'include'))]
    newest = max(os.path.getmtime(h) for h in hdrs)
    if force or not os.path.exists(out) or os.path.getmtime(out) < newest:
        hipcc = '/opt/rocm/bin/hipcc' if os.path.exists('/opt/rocm/bin/hipcc') else 'hipcc'
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            objs = [os.path.join(d, f[:-4] + '.o') for f in NATIVE_SOURCES]
            procs = [subprocess.Popen([hipcc] + HIPCC_FLAGS + ['-c', '-o', o, os.path.join(csrc, f)]) for f, o in zip(NATIVE_SOURCES, objs)]
            rcs = [p.wait() for p in procs]
            if any(rcs):
                raise subprocess.CalledProcessError(max(rcs), 'hipcc -c ' + ' '.join(NATIVE_SOURCES))
            subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-pthread', '-o', out] + objs)
    tool = os.path.join(REPO_ROOT, 'tools', 'thorenc_hip')
    tsrc = tool + '.c'
    if force or not os.path.exists(tool) or os.path.getmtime(tool) < os.path.getmtime(tsrc):
        subprocess.check_call(['gcc', '-O2', '-std=c99', '-D_POSIX_C_SOURCE=200809L', '-o', tool, tsrc,
                               '-L' + os.path.join(REPO_ROOT, 'thor_amd'), '-lthor_hip', '-Wl,-rpath,$ORIGIN/../thor_amd'])
    return out


class ThorParams(C.Structure):
    """thor_hip_params (include/thor_hip.h) == the enc_params fields this path honours."""
    _fields_ = [('width', C.c_int), ('height', C.c_int), ('qp', C.c_int), ('bitdepth', C.c_int), ('input_bitdepth', C.c_int),
                ('frame_rate', C.c_float), ('lambda_coeffI', C.c_float), ('lambda_coeffP', C.c_float),
                ('early_skip_thr', C.c_float), ('enable_tb_split', C.c_int), ('enable_pb_split', C.c_int),
                ('max_num_ref', C.c_int), ('HQperiod', C.c_int), ('num_reorder_pics', C.c_int), ('interp_ref', C.c_int),
                ('dqpP', C.c_int), ('dqpI', C.c_int), ('mqpP', C.c_float), ('intra_period', C.c_int), ('intra_rdo', C.c_int),
                ('encoder_speed', C.c_int), ('deblocking', C.c_int), ('cdef', C.c_int), ('clpf', C.c_int),
                ('use_block_contexts', C.c_int), ('enable_bipred', C.c_int), ('cfl_intra', C.c_int), ('cfl_inter', C.c_int),
                ('dyadic_coding', C.c_int), ('lambda_coeffB', C.c_float), ('lambda_coeffB0', C.c_float), ('lambda_coeffB1', C.c_float),
                ('lambda_coeffB2', C.c_float), ('lambda_coeffB3', C.c_float), ('dqpB', C.c_int), ('dqpB0', C.c_int), ('dqpB1', C.c_int),
                ('dqpB2', C.c_int), ('dqpB3', C.c_int), ('mqpB', C.c_float), ('mqpB0', C.c_float), ('mqpB1', C.c_float),
                ('mqpB2', C.c_float), ('mqpB3', C.c_float), ('max_clpf_strength', C.c_int)]


FRAMES_DONE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int)   # thor_hip_frames_done_fn


def lib():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError(f'{p} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no CPU path')
        L = C.CDLL(p)
        L.thor_hip_open.restype = C.c_void_p
        L.thor_hip_open.argtypes = [C.POINTER(ThorParams), C.c_int, C.c_int]
        L.thor_hip_close.argtypes = [C.c_void_p]
        L.thor_hip_begin_sequence.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.thor_hip_next_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.thor_hip_stage_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.thor_hip_stage_frame_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.thor_hip_encode_staged.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.thor_hip_encode_frame.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.thor_hip_encode_staged_run.argtypes = [C.c_void_p, C.c_int, FRAMES_DONE_FN, C.c_void_p]
        L.thor_hip_last_display_index.argtypes = [C.c_void_p, C.c_int]
        L.thor_hip_stream_bytes.restype = C.c_size_t
        L.thor_hip_stream_bytes.argtypes = [C.c_void_p, C.c_int]
        L.thor_hip_stream_data.restype = C.c_void_p
        L.thor_hip_stream_data.argtypes = [C.c_void_p, C.c_int]
        L.thor_hip_get_recon.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.thor_hip_kernel_time.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_double)]
        L.thor_hip_kernel_time_reset.argtypes = [C.c_void_p]
        L.thor_hip_read_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
        L.thor_hip_deblock_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.thor_hip_deblock_frame_hbd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.thor_hip_params_from_config.argtypes = [C.POINTER(ThorParams), C.c_char_p]
        L.thor_hip_params_set.argtypes = [C.POINTER(ThorParams), C.c_char_p, C.c_char_p]
        _LIB = L
    return _LIB


def load_config(cfg_path=None, **overrides):
    """ThorParams from a Thorenc-style config file plus "-name value" overrides (width=..., qp=...)."""
    p = ThorParams()
    rc = lib().thor_hip_params_from_config(C.byref(p), cfg_path.encode() if cfg_path else None)
    if rc:
        raise ValueError(f'{cfg_path}: unknown or unsupported option (rc={rc}); this path rejects what it cannot encode bit-exactly')
    for k, v in overrides.items():
        rc = lib().thor_hip_params_set(C.byref(p), ('-' + k).encode(), str(v).encode())
        if rc:
            raise ValueError(f'option -{k} {v}: ' + {1: 'unknown', 2: 'not implemented by this path', 3: 'front-end option, not an encoder parameter'}.get(rc, f'rc={rc}'))
    return p


class Encoder:
    """N independent closed streams encoded in lock step on one GPU (thor_hip_open ... thor_hip_close)."""

    def __init__(self, params, num_streams=1, device=0):
        self.p = params
        self.S = num_streams
        self.h = lib().thor_hip_open(C.byref(params), num_streams, device)
        if not self.h:
            raise RuntimeError('thor_hip_open failed (unsupported parameters?)')
        self.sample_bytes = 2 if params.bitdepth > 8 else 1
        self.frame_bytes = params.width * params.height * 3 // 2 * self.sample_bytes

    def close(self):
        if self.h:
            lib().thor_hip_close(self.h)
            self.h = None

    def stage(self, stream, slot, frame):
        frame = np.ascontiguousarray(frame).view(np.uint8)
        assert frame.size == self.frame_bytes
        rc = lib().thor_hip_stage_frame(self.h, stream, slot, frame.ctypes.data_as(C.c_void_p))
        if rc:
            raise RuntimeError(f'thor_hip_stage_frame rc={rc}')

    def stage_device(self, stream, slot, dev_ptr):
        """Stage a frame that already lives in HBM (device pointer to a contiguous planar 4:2:0 frame)."""
        rc = lib().thor_hip_stage_frame_device(self.h, stream, slot, C.c_void_p(dev_ptr))
        if rc:
            raise RuntimeError(f'thor_hip_stage_frame_device rc={rc}')

    def begin_sequence(self, stream, skip, num_frames, file_frames):
        """Fix the chunk [skip, skip+num_frames) of a file_frames-long input (thor_hip_begin_sequence)."""
        rc = lib().thor_hip_begin_sequence(self.h, stream, skip, num_frames, file_frames)
        if rc:
            raise RuntimeError(f'thor_hip_begin_sequence rc={rc}')

    def next_frame(self, stream):
        """Chunk-relative display index of the next frame to code, or None when the chunk is finished."""
        d = C.c_int()
        return d.value if lib().thor_hip_next_frame(self.h, stream, C.byref(d)) else None

    def encode_run(self, nframes, on_done=None):
        """thor_hip_encode_staged_run: the next nframes frames of every stream (inputs staged at their display index), the streams in two
        groups half a frame apart.  on_done(first_stream, num_streams) is called when the frames of those streams are complete (recon(s) and
        last_display_index(s) describe that frame inside the callback)."""
        err = []

        def _cb(_user, first, count):
            if on_done is not None and not err:
                try:
                    on_done(first, count)
                except BaseException as e:   # an exception must not unwind through the C frames
                    err.append(e)
        cb = FRAMES_DONE_FN(_cb)
        rc = lib().thor_hip_encode_staged_run(self.h, nframes, cb, None)
        if err:
            raise err[0]
        if rc:
            raise RuntimeError(f'thor_hip_encode_staged_run rc={rc}')

    def last_display_index(self, stream):
        return lib().thor_hip_last_display_index(self.h, stream)

    def encode_clips(self, clips, want_recon=True, skips=None, file_frames=None, staggered=False):
        """Encode clips[s] (equal-length lists of frames in display order) as closed streams in lock step - or, staggered=True, in two groups
        half a frame apart (encode_run) - following the coding-order schedule.  Stream s stands for frames [skips[s], skips[s]+n) of an input
        file holding file_frames frames (defaults: 0 and n).  Returns (bitstreams, recon[s][display index])."""
        n = len(clips[0])
        for s in range(self.S):
            for f in range(n):
                self.stage(s, f, clips[s][f])
            sk = skips[s] if skips else 0
            self.begin_sequence(s, sk, n, file_frames if file_frames else sk + n)
        recs = [[None] * n for _ in range(self.S)]
        if staggered:
            def done(first, count):
                if want_recon:
                    for s in range(first, first + count):
                        recs[s][self.last_display_index(s)] = self.recon(s)
            self.encode_run(n, done)
            return [self.bitstream(s) for s in range(self.S)], recs
        while True:
            idx = [self.next_frame(s) for s in range(self.S)]
            if idx[0] is None:
                break
            self.encode_staged(idx)
            if want_recon:
                for s in range(self.S):
                    recs[s][idx[s]] = self.recon(s)
        return [self.bitstream(s) for s in range(self.S)], recs

    def encode_staged(self, slots):
        arr = (C.c_int * self.S)(*slots)
        rc = lib().thor_hip_encode_staged(self.h, arr)
        if rc:
            raise RuntimeError(f'thor_hip_encode_staged rc={rc}')

    def bitstream(self, stream):
        n = lib().thor_hip_stream_bytes(self.h, stream)
        return C.string_at(lib().thor_hip_stream_data(self.h, stream), n)

    def recon(self, stream):
        out = np.empty(self.frame_bytes, dtype=np.uint8)
        rc = lib().thor_hip_get_recon(self.h, stream, out.ctypes.data_as(C.c_void_p))
        if rc:
            raise RuntimeError(f'thor_hip_get_recon rc={rc}')
        return out

    def recon_into(self, stream, ptr):
        """Reconstruction of the stream's last frame into a caller-owned host buffer of frame_bytes bytes (e.g. pinned memory that is reused)."""
        rc = lib().thor_hip_get_recon(self.h, stream, C.c_void_p(ptr))
        if rc:
            raise RuntimeError(f'thor_hip_get_recon rc={rc}')

    def kernel_time(self):
        a, b, c = C.c_double(), C.c_long(), C.c_double()
        lib().thor_hip_kernel_time(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def kernel_time_reset(self):
        lib().thor_hip_kernel_time_reset(self.h)
        self.stats(reset=True)

    def stats(self, reset=False):
        """Early-skip statistics of the inter frames coded since the last reset (thor_hip_read_stats)."""
        a = (C.c_ulonglong * 4)()
        lib().thor_hip_read_stats(self.h, a, 1 if reset else 0)
        px, sb = max(int(a[3]), 1), max(int(a[2]), 1)
        return {'early_skip_pixel_fraction': round(int(a[0]) / px, 4), 'early_skip_sb128_fraction': round(int(a[1]) / sb, 4),
                'inter_superblocks': int(a[2]), 'inter_luma_pixels': int(a[3])}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _pix(bitdepth):
    return np.uint16 if bitdepth > 8 else np.uint8


def sad_batch(org, ref_plane, bx, by, cand, bitdepth=8):
    """SADs of the candidates (thor_hip_sad_batch; bitdepth > 8: thor_hip_sad_batch_hbd on uint16 samples)."""
    T = _pix(bitdepth)
    org = np.ascontiguousarray(org, dtype=T); ref_plane = np.ascontiguousarray(ref_plane, dtype=T)
    cand = np.ascontiguousarray(cand, dtype=np.int32)
    out = np.zeros(len(cand), dtype=np.uint32)
    fn = lib().thor_hip_sad_batch_hbd if bitdepth > 8 else lib().thor_hip_sad_batch
    rc = fn(_vp(org), org.shape[1], org.shape[0], _vp(ref_plane), ref_plane.shape[1], ref_plane.shape[0], ref_plane.shape[1], bx, by,
            _vp(cand), len(cand), _vp(out))
    if rc:
        raise RuntimeError(f'thor_hip_sad_batch rc={rc}')
    return out


def interp_luma(ref_padded, pad, pic_w, pic_h, bx, by, w, h, mvs, bipred, bitdepth=8):
    T = _pix(bitdepth)
    ref_padded = np.ascontiguousarray(ref_padded, dtype=T); mvs = np.ascontiguousarray(mvs, dtype=np.int16)
    out = np.zeros((len(mvs), h, w), dtype=T)
    if bitdepth > 8:
        rc = lib().thor_hip_interp_luma_hbd(_vp(ref_padded), pic_w, pic_h, ref_padded.shape[1], pad, bx, by, w, h, _vp(mvs), len(mvs),
                                            bipred, bitdepth, _vp(out))
    else:
        rc = lib().thor_hip_interp_luma(_vp(ref_padded), pic_w, pic_h, ref_padded.shape[1], pad, bx, by, w, h, _vp(mvs), len(mvs),
                                        bipred, _vp(out))
    if rc:
        raise RuntimeError(f'thor_hip_interp_luma rc={rc}')
    return out


def code_tu_batch(org, pred, qp, coeff_type, fast, bitdepth=8):
    T = _pix(bitdepth)
    org = np.ascontiguousarray(org, dtype=T); pred = np.ascontiguousarray(pred, dtype=T)
    n, size = org.shape[0], org.shape[1]
    q = min(size, 16)
    coefq = np.zeros((n, q, q), dtype=np.int16); rec = np.zeros_like(org); cbp = np.zeros(n, dtype=np.int32)
    if bitdepth > 8:
        rc = lib().thor_hip_code_tu_batch_hbd(_vp(org), _vp(pred), size, qp, coeff_type, fast, n, bitdepth, _vp(coefq), _vp(rec), _vp(cbp))
    else:
        rc = lib().thor_hip_code_tu_batch(_vp(org), _vp(pred), size, qp, coeff_type, fast, n, _vp(coefq), _vp(rec), _vp(cbp))
    if rc:
        raise RuntimeError(f'thor_hip_code_tu_batch rc={rc}')
    return coefq, rec, cbp


def deblock_frame(yuv, width, height, qp, cells, bitdepth=8):
    """In-loop deblocking of one planar 4:2:0 frame (thor_hip_deblock_frame / _hbd); cells: (h/4, w/4, 16) uint8 records."""
    yuv = np.ascontiguousarray(yuv, dtype=_pix(bitdepth)).copy()
    cells = np.ascontiguousarray(cells, dtype=np.uint8)
    assert yuv.size == width * height * 3 // 2 and cells.size == (width // 4) * (height // 4) * 16
    if bitdepth > 8:
        rc = lib().thor_hip_deblock_frame_hbd(_vp(yuv), width, height, qp, bitdepth, _vp(cells))
    else:
        rc = lib().thor_hip_deblock_frame(_vp(yuv), width, height, qp, _vp(cells))
    if rc:
        raise RuntimeError(f'thor_hip_deblock_frame rc={rc}')
    return yuv


# ---- round 6: known-answer entry points (include/thor_hip.h, section 3b) ----------------------------------------------------------
def _kat(name, *args):
    fn = getattr(lib(), name)
    fn.restype = C.c_int
    rc = fn(*args)
    if rc:
        raise RuntimeError(f'{name} rc={rc}')


def kat_intra(plane, size, par, bitdepth=8, rblocks=None):
    """Intra prediction of len(par) transform units (thor_hip_kat_intra); par rows: ypos, xpos, upright, downleft, mode, i, j."""
    T = _pix(bitdepth)
    plane = np.ascontiguousarray(plane, dtype=T); par = np.ascontiguousarray(par, dtype=np.int32)
    n = len(par)
    out = np.zeros((n, size, size), dtype=T)
    rb = np.ascontiguousarray(rblocks, dtype=T) if rblocks is not None else None
    _kat('thor_hip_kat_intra', _vp(plane), plane.shape[1], plane.shape[0], plane.shape[1], bitdepth, size, 1 if rb is not None else 0, n, _vp(par),
         _vp(rb) if rb is not None else None, _vp(out))
    return out


def kat_inter_yuv(yuv, width, height, size, par, mvs, bitdepth=8):
    """thor_hip_kat_inter_yuv: par rows ypos, xpos, sign, enable_bipred, split; mvs [n][4][2]."""
    T = _pix(bitdepth)
    yuv = np.ascontiguousarray(yuv, dtype=T); par = np.ascontiguousarray(par, dtype=np.int32); mvs = np.ascontiguousarray(mvs, dtype=np.int16)
    n = len(par)
    out = np.zeros((n, size * size * 3 // 2), dtype=T)
    _kat('thor_hip_kat_inter_yuv', _vp(yuv), width, height, bitdepth, size, n, _vp(par), _vp(mvs), _vp(out))
    return out


def kat_average(a, b, size, bitdepth=8):
    T = _pix(bitdepth)
    a = np.ascontiguousarray(a, dtype=T); b = np.ascontiguousarray(b, dtype=T)
    out = np.zeros_like(a)
    _kat('thor_hip_kat_average', _vp(a), _vp(b), size, bitdepth, len(a), _vp(out))
    return out


def kat_cfl(y, uv, ry, n_luma, bitdepth=8):
    T = _pix(bitdepth)
    y = np.ascontiguousarray(y, dtype=T); ry = np.ascontiguousarray(ry, dtype=T); uv = np.array(uv, dtype=T, order='C')
    _kat('thor_hip_kat_cfl', _vp(y), _vp(uv), _vp(ry), n_luma, bitdepth, len(y))
    return uv


def kat_cdef_dir(blocks, bitdepth=8):
    T = _pix(bitdepth)
    blocks = np.ascontiguousarray(blocks, dtype=T)
    n = len(blocks)
    d = np.zeros(n, dtype=np.int32); v = np.zeros(n, dtype=np.int32)
    _kat('thor_hip_kat_cdef_dir', _vp(blocks), bitdepth, n, _vp(d), _vp(v))
    return d, v


def kat_cdef_filter(plane, bsize, par, bitdepth=8):
    T = _pix(bitdepth)
    plane = np.ascontiguousarray(plane, dtype=T); par = np.ascontiguousarray(par, dtype=np.int32)
    out = np.zeros((len(par), bsize, bsize), dtype=T)
    _kat('thor_hip_kat_cdef_filter', _vp(plane), plane.shape[1], plane.shape[0], plane.shape[1], bitdepth, bsize, len(par), _vp(par), _vp(out))
    return out


def kat_clpf(rec, org, width, height, qp, cells, strength, fb_log2, fb_on, bitdepth=8):
    T = _pix(bitdepth)
    rec = np.ascontiguousarray(rec, dtype=T); org = np.ascontiguousarray(org, dtype=T)
    cells = np.ascontiguousarray(cells, dtype=np.uint8); fb_on = np.ascontiguousarray(fb_on, dtype=np.uint8)
    st = np.ascontiguousarray(strength, dtype=np.int32)
    nblk = (width // 8) * (height // 8) + 2 * (width // 16) * (height // 16)
    stats = np.zeros((nblk, 4), dtype=np.uint32)
    out = np.zeros_like(rec)
    _kat('thor_hip_kat_clpf', _vp(rec), _vp(org), width, height, bitdepth, qp, _vp(cells), _vp(st), fb_log2, _vp(fb_on), _vp(stats), _vp(out))
    return stats, out


def kat_interpolate(yuv0, yuv1, width, height, bitdepth=8):
    T = _pix(bitdepth)
    a = np.ascontiguousarray(yuv0, dtype=T); b = np.ascontiguousarray(yuv1, dtype=T)
    out = np.zeros_like(a)
    _kat('thor_hip_kat_interpolate', _vp(a), _vp(b), width, height, bitdepth, _vp(out))
    return out
