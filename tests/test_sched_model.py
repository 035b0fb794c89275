"""Model check of the superblock dependency scheduler (thor_amd/csrc/tk_sched.h:df_finish, used by thor_hip.cpp:k_superblocks).

The kernel releases a task when its dependency counter reaches `need`; a finishing task (k,l) bumps the counters
of  (k,l+1),  (k+1,l-1)  and, in the last column,  (k+1,l).  This test restates that successor rule and checks,
for every grid up to 40x24 superblocks (4K is 30x17, 1080p 15x9; single-row and single-column frames included),
that it is exactly the inverse of the dependency relation the codec imposes - SB(k,l) needs its left neighbour
(k,l-1) and the up-right neighbour (k-1,l+1), or (k-1,l) in the last column - and that a
FIFO execution releases every task exactly once, never before its dependencies."""
import itertools


def deps(k, l, rows, cols):
    d = []
    if l > 0:
        d.append((k, l - 1))
    if k > 0:
        d.append((k - 1, min(l + 1, cols - 1)))
    return d


def successors(k, l, rows, cols):  # the rule coded in tk_sched.h:df_finish
    s = []
    if l + 1 < cols:
        s.append((k, l + 1))
    if k + 1 < rows:
        if l >= 1:
            s.append((k + 1, l - 1))
        if l == cols - 1:
            s.append((k + 1, l))
    return s


def test_successor_rule_inverts_the_dependency_relation():
    for rows, cols in itertools.product(range(1, 25), range(1, 41)):
        want = {}
        for k in range(rows):
            for l in range(cols):
                for d in deps(k, l, rows, cols):
                    want.setdefault(d, set()).add((k, l))
        for k in range(rows):
            for l in range(cols):
                got = successors(k, l, rows, cols)
                assert len(got) == len(set(got)), (rows, cols, k, l)
                assert set(got) == want.get((k, l), set()), (rows, cols, k, l)


def test_fifo_execution_releases_every_task_once_and_in_dependency_order():
    for rows, cols in ((1, 1), (1, 7), (9, 1), (2, 2), (9, 15), (17, 30), (3, 3), (2, 4)):
        need = {(k, l): (l > 0) + (k > 0) for k in range(rows) for l in range(cols)}
        cnt = dict.fromkeys(need, 0)
        queue, done, pushed = [(0, 0)], set(), {(0, 0)}
        while queue:
            k, l = queue.pop(0)
            assert all(d in done for d in deps(k, l, rows, cols)), (rows, cols, k, l)
            done.add((k, l))
            for s in successors(k, l, rows, cols):
                cnt[s] += 1
                if cnt[s] == need[s]:
                    assert s not in pushed
                    pushed.add(s)
                    queue.append(s)
        assert len(done) == rows * cols and pushed == set(need), (rows, cols)
