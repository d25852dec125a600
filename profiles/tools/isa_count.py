#!/usr/bin/env python3
"""Count instruction classes of a gfx950 .s listing between two line numbers (or labels).

usage: isa_count.py file.s [first_line last_line]
Classes: valu (v_*), of which f64 arithmetic / dpp / cndmask / readlane / cmp; salu (s_* except
waitcnt/nop/branch), branch, lds (ds_*), vmem (global_/buffer_/flat_/scratch_), smem (s_load*),
waitcnt, nop.
"""
import re, sys, collections
def classify(op):
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    if op.startswith(('s_load', 's_buffer_load')):
        return 'smem'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_'):
        return 'salu'
    return 'other'
def main():
    path = sys.argv[1]
    lines = open(path).read().split('\n')
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(lines)
    cls = collections.Counter(); ops = collections.Counter()
    for ln in lines[lo - 1:hi]:
        s = ln.strip()
        if not s or s.startswith((';', '.', '//')) or s.endswith(':'):
            continue
        op = s.split()[0]
        if not re.match(r'^[a-z_0-9]+$', op):
            continue
        cls[classify(op)] += 1
        ops[op] += 1
    print(dict(cls))
    for op, n in ops.most_common(60):
        print(f'{n:6d} {op}')
main()
