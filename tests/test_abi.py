"""The C-ABI library loads and exports every function include/thor_hip.h declares (no compute calls:
this runs without a GPU), and thor_abi.h matches the reference structure layouts."""
import ctypes as C
import os
import re
import subprocess
import tempfile
import pytest
from util import ROOT, HAVE_REFERENCE_TREE


def declared_functions():
    h = open(os.path.join(ROOT, 'include', 'thor_hip.h')).read()
    h = re.sub(r'/\*.*?\*/', '', h, flags=re.S)
    return sorted(set(re.findall(r'\b(thor_hip_\w+|encode_frame_\w+)\s*\(', h)))


def test_library_exports_every_declared_symbol():
    import thor_amd
    lib = thor_amd.lib()
    names = declared_functions()
    assert 'encode_frame_lbd' in names and 'thor_hip_encode_staged' in names and len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/thor_hip.h but not exported'


def test_params_from_config_matches_config_file():
    import thor_amd
    p = thor_amd.load_config(os.path.join(ROOT, 'configs', 'ldb_high_efficiency.cfg'), width=1920, height=1080, qp=32)
    assert (p.max_num_ref, p.HQperiod, p.enable_bipred, p.enable_tb_split, p.enable_pb_split, p.intra_rdo) == (4, 12, 1, 1, 1, 1)
    assert abs(p.mqpP - 1.2) < 1e-6 and p.dqpI == -2 and p.cdef == 2 and p.encoder_speed == 0


@pytest.mark.skipif(not HAVE_REFERENCE_TREE, reason='reference headers only exist in the build container')
def test_abi_layout_matches_reference_headers():
    probe = r'''
#include <stdio.h>
#include <stddef.h>
#include "mainenc.h"
#include "%s/include/thor_abi.h"
#define CHK(rt, m, mt) if (offsetof(rt, m) != offsetof(mt, m)) { printf("offset %%s.%%s\n", #rt, #m); bad++; }
#define SZ(rt, mt) if (sizeof(rt) != sizeof(mt)) { printf("size %%s\n", #rt); bad++; }
int main(void) { int bad = 0;
  SZ(yuv_frame_t, thor_yuv_frame) SZ(stream_t, thor_stream) SZ(enc_params, thor_enc_params) SZ(frame_info_t, thor_frame_info)
  SZ(encoder_info_t, thor_encoder_info) SZ(stream_pos_t, thor_stream_pos) SZ(deblock_data_t, thor_deblock_data) SZ(inter_pred_t, thor_inter_pred)
  CHK(yuv_frame_t, stride_c, thor_yuv_frame) CHK(yuv_frame_t, frame_num, thor_yuv_frame) CHK(yuv_frame_t, pad_ver_c, thor_yuv_frame)
  CHK(enc_params, early_skip_thr, thor_enc_params) CHK(enc_params, mqpP, thor_enc_params) CHK(enc_params, cdef, thor_enc_params)
  CHK(enc_params, enable_bipred, thor_enc_params) CHK(enc_params, cfl_inter, thor_enc_params) CHK(enc_params, input_bitdepth, thor_enc_params)
  CHK(frame_info_t, qp, thor_frame_info) CHK(frame_info_t, ref_array, thor_frame_info) CHK(frame_info_t, lambda, thor_frame_info)
  CHK(frame_info_t, num_intra_modes, thor_frame_info) CHK(frame_info_t, frame_num, thor_frame_info) CHK(frame_info_t, prev_qp, thor_frame_info)
  CHK(encoder_info_t, params, thor_encoder_info) CHK(encoder_info_t, orig, thor_encoder_info) CHK(encoder_info_t, ref, thor_encoder_info)
  CHK(encoder_info_t, stream, thor_encoder_info) CHK(encoder_info_t, width, thor_encoder_info) CHK(encoder_info_t, cdef_bits, thor_encoder_info)
  CHK(encoder_info_t, deblock_data, thor_encoder_info) CHK(deblock_data_t, mode, thor_deblock_data) CHK(deblock_data_t, size, thor_deblock_data)
  CHK(deblock_data_t, tb_split, thor_deblock_data) CHK(deblock_data_t, pb_part, thor_deblock_data) CHK(deblock_data_t, inter_pred, thor_deblock_data)
  CHK(deblock_data_t, inter_pred_arr, thor_deblock_data) CHK(inter_pred_t, ref_idx0, thor_inter_pred) CHK(inter_pred_t, bipred_flag, thor_inter_pred)
  if (offsetof(deblock_data_t, cbp) != offsetof(thor_deblock_data, cbp_y) || offsetof(deblock_data_t, cbp) + offsetof(cbp_t, v) != offsetof(thor_deblock_data, cbp_v)) { printf("offset deblock_data_t.cbp\n"); bad++; }
  return bad; }''' % ROOT
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 'p.c'), 'w').write(probe)
        subprocess.check_call(['gcc', '-fcommon', '-std=c99', '-I', '/root/reference/common', '-I', '/root/reference/enc', '-o',
                               os.path.join(d, 'p'), os.path.join(d, 'p.c')])
        r = subprocess.run([os.path.join(d, 'p')], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
