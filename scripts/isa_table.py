#!/usr/bin/env python3
"""isa_table.py [OUT.md] [extra hipcc flags...] - static resource table of the gfx950 code object of thor_amd/csrc/thor_hip.cpp:
per kernel VGPRs / SGPRs / spill slots / private (scratch) bytes per lane / LDS bytes / occupancy, and per function the static
counts of scratch_load / scratch_store / flat_* / global_* / ds_* / v_sad / v_dot instructions (where the spill and generic-pointer
traffic of k_superblocks sits).  Runs in the build container (hipcc cross-compiles; no GPU needed):
  python scripts/isa_table.py profiles/r04_isa_resources.md
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(names):
    try:
        out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True, check=True).stdout.split('\n')
        return dict(zip(names, out))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


def short(n):
    n = re.sub(r'\(.*$', '', n)           # drop the argument list
    n = re.sub(r'^(void|int|unsigned int|unsigned long) ', '', n)
    n = n.replace('unsigned char', 'u8').replace('unsigned short', 'u16').replace('tk::', '')
    return n[:110]


def main():
    out_md = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else None
    flags = [a for a in sys.argv[1:] if a.startswith('-') and not a.startswith('--src=')]
    src = ([a[6:] for a in sys.argv[1:] if a.startswith('--src=')] or ['thor_hip.cpp'])[0]   # --src=thor_hip_lat.cpp: the second build of the engine (round 6)
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, 'dev.s')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-strict-aliasing', '--cuda-device-only', '-S', '-w'] + flags +
                              ['-o', asm, os.path.join(ROOT, 'thor_amd', 'csrc', src)])
        text = open(asm).read()
    # ---- per function instruction counts
    cur = None
    cnt = collections.OrderedDict()
    pats = {'scratch_load': r'\tscratch_load', 'scratch_store': r'\tscratch_store', 'flat': r'\tflat_(load|store|atomic)', 'global': r'\tglobal_(load|store|atomic)',
            'ds': r'\tds_', 'v_sad': r'\tv_(m|q)?sad', 'v_dot': r'\tv_dot', 's_barrier': r'\ts_barrier', 'call': r'\ts_swappc'}
    pats = {k: re.compile(v) for k, v in pats.items()}
    info = {}
    for line in text.split('\n'):
        m = re.match(r'^([A-Za-z_][\w.$]*):\s*(;.*)?$', line)
        if m and not m.group(1).startswith('.L'):
            cur = m.group(1)
            cnt.setdefault(cur, collections.Counter())
            continue
        if cur is None:
            continue
        if line.startswith('\t.size\t') or line.startswith('.Lfunc_end'):
            pass
        for k, p in pats.items():
            if p.search(line):
                cnt[cur][k] += 1
        m = re.match(r'^; (codeLenInByte|NumVgprs|TotalNumSgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize)\s*[:=]\s*(\d+)', line)
        if m:
            info.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    # ---- kernel metadata (the .amdgpu_metadata yaml at the end)
    kern = []
    for blk in re.split(r'\n  - \.agpr_count:', text)[1:]:
        g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
        kern.append({k: g(k) for k in ('name', 'vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size', 'group_segment_fixed_size', 'max_flat_workgroup_size')})
    dm = demangle([k['name'] for k in kern] + list(cnt))
    L = ['# Static resources of the gfx950 code object (`thor_amd/csrc/' + src + '`, hipcc -O3' + (' ' + ' '.join(flags) if flags else '') + ')', '',
         'From `hipcc --cuda-device-only -S`: the kernel descriptors (`.amdgpu_metadata`) and instruction counts of the assembly; `scripts/isa_table.py`.', '',
         '## Kernels', '', '| kernel | VGPRs | SGPRs | VGPR spill slots | SGPR spill slots | private bytes / lane | LDS bytes / workgroup | waves / SIMD |', '|---|---|---|---|---|---|---|---|']
    for k in kern:
        occ = info.get(k['name'], {}).get('Occupancy', '?')
        L.append(f"| `{short(dm[k['name']])}` | {k['vgpr_count']} | {k['sgpr_count']} | {k['vgpr_spill_count']} | {k['sgpr_spill_count']} | {k['private_segment_fixed_size']} | {k['group_segment_fixed_size']} | {occ} |")
    tot = collections.Counter()
    for c in cnt.values():
        tot.update(c)
    L += ['', '## Instruction counts (static, whole code object)', '', '| ' + ' | '.join(pats) + ' |', '|' + '---|' * len(pats), '| ' + ' | '.join(str(tot[k]) for k in pats) + ' |', '',
          '## Functions with scratch instructions (spills, callee-saved registers, private arrays), largest first', '',
          '| function | code bytes | VGPRs | scratch bytes | scratch_load | scratch_store | flat | global | ds |', '|---|---|---|---|---|---|---|---|---|']
    rows = sorted(cnt.items(), key=lambda kv: -(kv[1]['scratch_load'] + kv[1]['scratch_store']))
    for name, c in rows[:40]:
        if c['scratch_load'] + c['scratch_store'] == 0:
            break
        i = info.get(name, {})
        L.append(f"| `{short(dm[name])}` | {i.get('codeLenInByte', '?')} | {i.get('NumVgprs', '?')} | {i.get('ScratchSize', '?')} | {c['scratch_load']} | {c['scratch_store']} | {c['flat']} | {c['global']} | {c['ds']} |")
    s = '\n'.join(L) + '\n'
    if out_md:
        open(out_md, 'w').write(s)
    print(s)


if __name__ == '__main__':
    main()
