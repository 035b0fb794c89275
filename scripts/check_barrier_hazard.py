#!/usr/bin/env python3
"""check_barrier_hazard.py FILE.s [function-substring ...] - static check of the gfx950 assembly (hipcc --cuda-device-only -S) for the code-generation
hazard bisected in round 4 (profiles/r04_hang_bisect_extra.diff): a workgroup barrier that sits inside an exec-masked region
or inside a lane-divergent loop.  In the block-decision functions (md_worker_sp, bipred_par, wg_helper_loop / mode_decision_par and the kernel
itself) every `s_barrier` must be reached with the wave's full EXEC mask: a lane-divergent region at the tail of md_worker_sp's item loop once
made the compiler thread the loop's back edge through an exec-masked branch, and the lock-step barrier behind it never completed.

The check walks each function in layout order (the structuriser emits properly nested regions):
  * `s_and_saveexec_b64 / s_or_saveexec_b64 / s_andn2_saveexec_b64 sX, ...` opens a masked region, `s_or_b64 exec, exec, sX` closes it;
    `s_xor_b64 exec, exec, sX` (the else of a region) keeps it open;
  * `s_andn2_b64 exec, exec, sX` followed by `s_cbranch_execnz LABEL` closes a lane-divergent loop whose body starts at LABEL.
An `s_barrier` inside either is reported (exit code 1 with --strict).

LIMITS (round 5, measured on both builds): this is a layout-order heuristic, not a control-flow analysis.  It found one real site - the branch on
`sh->do_bipred == 2` in md_worker_sp was a vector compare + s_and_saveexec around the lock-step barrier (uniform in practice); it is a scalar
branch now (tk_uniform) - but it does NOT separate the plain build from the hanging -DTHOR_PROF_MD -DTHOR_PROF_MD_PARTS=1 build: both keep the
trap exit of the intra-workgroup waits (`s_trap 2 ; divergent unreachable` + s_andn2_saveexec) in front of that barrier, and regions whose closing
s_or_b64 is laid out before their opening are misread.  The protection against the hazard therefore stays the coding rule of DESIGN.md 8
(per-item bookkeeping in md_worker_sp's loop is wave-uniform or accumulated in registers); this script is a reading aid for the assembly."""
import re
import sys


def functions(lines):
    """(name, start, end) of every function body in the assembly."""
    out, cur = [], None
    for i, l in enumerate(lines):
        m = re.match(r'^([A-Za-z_][\w$.]*):\s*(;.*)?$', l)
        if m and not l.startswith('.L'):
            cur = (m.group(1), i)
        elif l.startswith('.Lfunc_end') and cur:
            out.append((cur[0], cur[1], i))
            cur = None
    return out


def check(lines, name, a, b):
    bad = []
    stack = []          # (saved-exec register, line) of the open masked regions
    labels = {}
    body = lines[a:b]
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    loops = []          # (start, end) of lane-divergent loops
    for i, l in enumerate(body):
        t = l.strip()
        m = re.match(r's_cbranch_execnz\s+(\.LBB\d+_\d+)', t)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            # a backward branch on "some lanes still active" = a loop with a per-lane exit
            prev = next((body[j].strip() for j in range(i - 1, max(i - 6, -1), -1) if body[j].strip() and not body[j].strip().startswith(';')), '')
            if re.match(r's_andn2_b64\s+exec,\s*exec', prev) or re.match(r's_and_b64\s+exec,\s*exec', prev) or 'exec' in prev:
                loops.append((labels[m.group(1)], i))
    for i, l in enumerate(body):
        t = l.strip()
        if re.match(r's_or_saveexec_b64\s+s\[\d+:\d+\],\s*-1', t):
            continue    # whole-wave mode around a register spill: widens EXEC, restored by the s_mov_b64 exec that follows
        m = re.match(r's_(?:and|or|andn2|orn2|xor|nand|nor|xnor)_saveexec_b64\s+(s\[\d+:\d+\]|vcc)', t)
        if m:
            stack.append((m.group(1), a + i + 1))
            continue
        m = re.match(r's_or_b64\s+exec,\s*exec,\s*(s\[\d+:\d+\]|vcc)', t)
        if m:
            for k in range(len(stack) - 1, -1, -1):
                if stack[k][0] == m.group(1):
                    del stack[k:]
                    break
            else:
                if stack:
                    stack.pop()
            continue
        if re.match(r's_mov_b64\s+exec,\s*-1', t):
            stack.clear()
            continue
        if t.startswith('s_barrier'):
            if stack:
                bad.append((a + i + 1, 'inside the exec-masked region opened at line %d (%s)' % (stack[-1][1], stack[-1][0])))
            for (s0, s1) in loops:
                if s0 <= i <= s1:
                    bad.append((a + i + 1, 'inside the lane-divergent loop of lines %d..%d' % (a + s0 + 1, a + s1 + 1)))
    return bad


def main():
    path = sys.argv[1]
    want = [a for a in sys.argv[2:] if not a.startswith('--')] or ['md_worker_sp', 'bipred_par', 'mode_decision_par', 'wg_helper_loop', 'k_superblocks', 'process_sb']
    lines = open(path).read().split('\n')
    nbar, nfun, total = 0, 0, 0
    for name, a, b in functions(lines):
        if not any(w in name for w in want):
            continue
        n = sum(1 for l in lines[a:b] if l.strip().startswith('s_barrier'))
        if not n:
            continue
        nfun += 1
        nbar += n
        for line, why in check(lines, name, a, b):
            total += 1
            print(f'{path}:{line}: s_barrier in {name[:60]}... {why}')
    print(f'check_barrier_hazard: {nbar} workgroup barriers in {nfun} functions checked, {total} behind a narrowed EXEC mask')
    return 1 if (total and '--strict' in sys.argv) else 0


if __name__ == '__main__':
    sys.exit(main())
