#!/usr/bin/env python3
"""Average rocprofv3 counters per kernel: pmc_summary.py <dir> [kernel-substring ...]

Reads every *_counter_collection.csv and *_kernel_stats.csv below <dir>; prints, per kernel
whose name contains one of the substrings (default: all), the mean of every counter over its
dispatches (the first quarter dropped as warm-up) and the derived per-wave figures.
SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md)."""
import collections, csv, glob, os, sys
def main():
  d = sys.argv[1]
  subs = sys.argv[2:]
  acc = collections.defaultdict(lambda: collections.defaultdict(list))
  for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
      k = r["Kernel_Name"].split("(")[0]
      if subs and not any(s in k for s in subs):
        continue
      acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
  for k, cnt in sorted(acc.items()):
    print("==", k)
    m = {}
    for name, v in sorted(cnt.items()):
      v = v[len(v) // 4:]
      m[name] = sum(v) / len(v)
      print("  %-24s %.5g  (n=%d)" % (name, m[name], len(v)))
    w = m.get("SQ_WAVES")
    if w:
      for a, b in (("SQ_INSTS_VALU", "valu/wave"), ("SQ_INSTS_SALU", "salu/wave"), ("SQ_INSTS_LDS", "lds/wave"),
                   ("SQ_INSTS_BRANCH", "branch/wave"), ("SQ_INSTS_VMEM_RD", "vmem_rd/wave"), ("SQ_INSTS_SMEM", "smem/wave")):
        if a in m:
          print("  -> %-14s %.1f" % (b, m[a] / w))
      if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        for a in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS"):
          if a in m:
            print("  -> %-28s %.3f of wave cycles" % (a, m[a] / wc))
        print("  -> wave life (cycles, x4)      %.0f" % (4 * wc / w))
  for f in glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True):
    print("==", f)
    for r in csv.DictReader(open(f)):
      n = r["Name"].split("(")[0]
      if subs and not any(s in n for s in subs):
        continue
      print("  %-60s calls %s avg %.1f us min %.1f max %.1f  %s%%" % (
          n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
main()
