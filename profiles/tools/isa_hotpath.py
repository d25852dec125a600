#!/usr/bin/env python3
"""Walk the HOT path of a loop in a gfx950 listing: from the header of the N-th depth-2 loop,
follow fall-throughs and unconditional branches, take a conditional branch only when it is the
loop's back edge; count instruction classes per stretch between labels.  Blocks the compiler
moved out of line (rare paths) are not visited.
usage: isa_hotpath.py file.s [N=last]"""
import re, sys, collections
path = sys.argv[1]
L = open(path).read().split('\n')
hdrs = [i for i, l in enumerate(L) if 'This Loop Header: Depth=2' in l or 'This Inner Loop Header: Depth=2' in l]
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(hdrs)
h = hdrs[n - 1]
# the header label is a few lines above the comment
start = max(i for i in range(h - 3, h + 1) if re.match(r'^\.LBB\d+_\d+:', L[i]))
hdr_label = L[start].split(':')[0]
labels = {l.split(':')[0]: i for i, l in enumerate(L) if re.match(r'^\.LBB\d+_\d+:', l)}
def cls(op):
  if op.startswith('v_'): return 'valu'
  if op.startswith('ds_'): return 'lds'
  if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
  if op.startswith('s_load'): return 'smem'
  if op.startswith('s_waitcnt'): return 'waitcnt'
  if op.startswith('s_nop'): return 'nop'
  if op.startswith(('s_cbranch', 's_branch')): return 'branch'
  if op.startswith('s_'): return 'salu'
  return 'other'
tot = collections.Counter(); ops = collections.Counter()
i = start + 1; visited = set(); trace = []
while True:
  if i in visited or i >= len(L): break
  visited.add(i)
  s = L[i].strip()
  if not s or s.startswith((';', '//')) or s.endswith(':') or s.startswith('.'):
    i += 1; continue
  op = s.split()[0]
  if not re.match(r'^[a-z_0-9]+$', op): i += 1; continue
  tot[cls(op)] += 1; ops[op] += 1
  if op == 's_branch':
    tgt = s.split()[1]
    if tgt == hdr_label: break
    i = labels[tgt] + 1; continue
  if op.startswith('s_cbranch'):
    tgt = s.split()[1]
    if tgt == hdr_label: break
    # heuristics: a forward branch to a block placed far below (out of line) is a rare path
  if op in ('s_endpgm',): break
  i += 1
print(dict(tot))
for op, c in ops.most_common(40): print('%5d %s' % (c, op))
