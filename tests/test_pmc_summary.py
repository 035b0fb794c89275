"""scripts/pmc_summary.py on synthetic rocprofv3 CSVs: counters are summed over the launches of k_superblocks only, the leading SKIP launches (the
warm-up frames of a bench.py run) are left out by dispatch order, FETCH_SIZE gets the gfx950 x2 correction, and the JSON carries the digest of the engine
sources bench.py compares with before it attaches roofline.traffic."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(d, tag, counters, n_launch):
    p = os.path.join(d, f'pmc_{tag}', 'host')
    os.makedirs(p)
    with open(os.path.join(p, '1_counter_collection.csv'), 'w', newline='') as f:
        w = csv.DictWriter(f, ['Dispatch_Id', 'Kernel_Name', 'Counter_Name', 'Counter_Value'])
        w.writeheader()
        for i in range(n_launch):
            for c, v in counters.items():
                w.writerow({'Dispatch_Id': 10 * i + 3, 'Kernel_Name': 'void tk::k_superblocks<unsigned char>(...)', 'Counter_Name': c, 'Counter_Value': v * (100 if i < 2 else 1)})
                w.writerow({'Dispatch_Id': 10 * i + 4, 'Kernel_Name': 'void tk::k_cdef<unsigned char>(...)', 'Counter_Name': c, 'Counter_Value': 12345})
    with open(os.path.join(p, '1_kernel_trace.csv'), 'w', newline='') as f:
        w = csv.DictWriter(f, ['Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
        w.writeheader()
        for i in range(n_launch):
            w.writerow({'Kernel_Name': 'void tk::k_superblocks<unsigned char>(...)', 'Start_Timestamp': 10**10 * i, 'End_Timestamp': 10**10 * i + (5 * 10**9 if i < 2 else 10**9)})


def test_summary_skips_the_warm_up_launches_and_applies_the_gfx950_corrections(tmp_path):
    d = str(tmp_path)
    _write(d, 'sq1', {'SQ_INSTS_VALU': 1000.0, 'SQ_WAVE_CYCLES': 8000.0, 'SQ_WAIT_ANY': 4000.0, 'SQ_ACTIVE_INST_ANY': 2000.0, 'SQ_ACTIVE_INST_VALU': 1000.0,
                      'SQ_WAIT_INST_ANY': 500.0, 'SQ_INSTS_SALU': 400.0, 'SQ_INSTS_LDS': 100.0}, 6)
    _write(d, 'fetch', {'FETCH_SIZE': 10.0}, 6)
    _write(d, 'write', {'WRITE_SIZE': 30.0}, 6)
    out = os.path.join(d, 'sum')
    # 64x64 frames, 2 streams, 4 timed frames; the first 2 launches are warm-up (100x larger counters: they must not show up)
    subprocess.check_call([sys.executable, os.path.join(ROOT, 'scripts', 'pmc_summary.py'), os.path.join(d, 'pmc'), '64', '64', '2', '4', out, 'synthetic', '2'],
                          stdout=subprocess.DEVNULL)
    r = json.load(open(out + '.json'))
    px = 64 * 64 * 2 * 4
    assert r['launches'] == 4 and abs(r['kernel_seconds_total'] - 4.0) < 1e-9
    assert abs(r['valu_insts_per_px'] - 4 * 1000.0 / px) < 1e-9
    assert abs(r['fetch_bytes_per_px'] - 4 * 10.0 * 1024 * 2 / px) < 1e-9      # KiB -> bytes, x2 on gfx950
    assert abs(r['write_bytes_per_px'] - 4 * 30.0 * 1024 / px) < 1e-9
    assert abs(r['wave_cycles_wait_any_frac'] - 0.5) < 1e-9
    sys.path.insert(0, ROOT)
    import bench
    assert r['csrc_digest'] == bench.csrc_digest() and r['streams'] == 2 and r['frames'] == 4
    assert os.path.exists(out + '.md')
