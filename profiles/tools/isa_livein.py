#!/usr/bin/env python3
"""Rough live-in set of a straight-line region of a gfx950 listing: VGPRs read before they are
written between two line numbers (control flow ignored).  usage: isa_livein.py file.s lo hi"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
lo, hi = int(sys.argv[2]), int(sys.argv[3])
written, livein = set(), set()
def regs(tok):
  out = []
  for m in re.finditer(r'v\[(\d+):(\d+)\]|v(\d+)', tok):
    if m.group(1):
      out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    else:
      out.append(int(m.group(3)))
  return out
for ln in lines[lo - 1:hi]:
  s = ln.split(';')[0].strip()
  if not s or s.startswith('.') or s.endswith(':'):
    continue
  parts = s.split(None, 1)
  if len(parts) < 2:
    continue
  op, args = parts
  ops = [a.strip() for a in args.split(',')]
  if op.startswith(('ds_write', 'scratch_store', 'global_store', 's_', 'buffer_store')) or op.startswith('v_cmp') and not op.endswith('_e64'):
    dst, src = [], ops
  else:
    dst, src = ops[:1], ops[1:]
  if op in ('v_fmac_f64_e32', 'v_fmac_f32_e32', 'v_mov_b32_dpp', 'v_writelane_b32'):
    src = ops  # destination is also read
  for t in src:
    for r in regs(t):
      if r not in written:
        livein.add(r)
  for t in dst:
    for r in regs(t):
      written.add(r)
print("live-in VGPRs: %d" % len(livein), sorted(livein))
