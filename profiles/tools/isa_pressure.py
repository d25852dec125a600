#!/usr/bin/env python3
"""Rough VGPR pressure profile of a straight-line region of a gfx950 listing (control flow
ignored; the region is treated as a loop body: registers read before written are live at both
ends).  usage: isa_pressure.py file.s lo hi [step]   -- prints live count every `step` lines and
the maximum."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
lo, hi = int(sys.argv[2]), int(sys.argv[3])
step = int(sys.argv[4]) if len(sys.argv) > 4 else 25
def regs(tok):
  out = []
  for m in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
    if m.group(1):
      out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    else:
      out.append(int(m.group(3)))
  return out
ins = []
for n in range(lo - 1, hi):
  s = lines[n].split(';')[0].strip()
  if not s or s.startswith('.') or s.endswith(':'):
    continue
  parts = s.split(None, 1)
  if len(parts) < 2:
    continue
  op, args = parts
  ops = [a.strip() for a in args.split(',')]
  if op.startswith(('ds_write', 'scratch_store', 'global_store', 's_', 'buffer_store')) or (op.startswith('v_cmp') and not op.endswith('_e64')):
    dst, src = [], ops
  else:
    dst, src = ops[:1], ops[1:]
  if op in ('v_fmac_f64_e32', 'v_fmac_f32_e32', 'v_mov_b32_dpp', 'v_writelane_b32'):
    src = ops
  d = set(r for t in dst for r in regs(t))
  u = set(r for t in src for r in regs(t))
  ins.append((n + 1, d, u))
# loop: live-out = live-in; iterate twice
live = set()
for _ in range(2):
  prof = []
  for n, d, u in reversed(ins):
    live = (live - d) | u
    prof.append((n, len(live)))
prof.reverse()
mx = max(prof, key=lambda t: t[1])
print("max live VGPRs %d at line %d" % (mx[1], mx[0]))
for i in range(0, len(prof), step):
  print(prof[i][0], prof[i][1])
