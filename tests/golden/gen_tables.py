#!/usr/bin/env python3
"""Record the reference's constant tables (parsed from its C sources, build container only) into
tests/golden/tables.json; tests/test_tables.py checks the engine's generated tables against it."""
import json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = '/root/reference'
def arr(path, name):
    s = open(os.path.join(REF, path)).read()
    m = re.search(name + r'\s*(\[[^\]]*\])+\s*=\s*\{(.*?)\};', s, re.S)
    return [float(x) if '.' in x else int(x) for x in re.findall(r'-?\d+\.?\d*', re.sub(r'//.*', '', m.group(2)))]
out = {
    'zigzag16': arr('common/common_tables.c', 'zigzag16'), 'zigzag64': arr('common/common_tables.c', 'zigzag64'),
    'zigzag256': arr('common/common_tables.c', 'zigzag256'), 'chroma_qp': arr('common/common_tables.c', 'chroma_qp'),
    'gquant': arr('common/common_tables.c', 'gquant_table'), 'gdequant': arr('common/common_tables.c', 'gdequant_table'),
    'lambda': arr('enc/encode_tables.c', 'squared_lambda_QP'),
    'dct4': arr('common/transform.c', 'g1mat_hevc'), 'dct8': arr('common/transform.c', 'g2mat_hevc'),
    'dct16': arr('common/transform.c', 'g3mat_hevc'), 'dct32': arr('common/transform.c', 'g4mat_hevc'),
    'beta': arr('common/common_frame.c', 'beta_table'), 'tc': arr('common/common_frame.c', 'tc_table'),
}
json.dump(out, open(os.path.join(ROOT, 'tests/golden/tables.json'), 'w'))
print({k: len(v) for k, v in out.items()})
