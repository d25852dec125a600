#!/usr/bin/env python3
"""gpurun_out/pmc_coupled/ (profiles/collect_coupled_pmc_r02.sh) -> profiles/coupled_counters.json
and profiles/r02/coupled_counters.md.

Per kernel and config, averaged over the launches of the timed region:
  valu / salu / lds / branch instructions per wave;
  valu_busy = SQ_ACTIVE_INST_VALU x 4 / (kernel cycles x 1024 SIMDs): the fraction of the
  machine's vector-issue cycles the kernel fills (kernel cycles = GRBM_GUI_ACTIVE summed over the
  8 XCDs / 8).  SQ_*_CYCLES / SQ_ACTIVE_* count quad-cycles (MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "pmc_coupled")
KERNELS = ("k_thermwind", "k_psi_so", "k_jn2018_steps", "k_column_steps")
out, lines = {}, ["# SQ counters of the coupled configs' kernels (round 2, MI355X)", "",
                  "| config | kernel | launches | waves | VALU / wave | SALU / wave | LDS / wave | "
                  "branches / wave | VALU-busy fraction of the machine |", "|---|---|---|---|---|---|---|---|---|"]
for c in (3, 4, 5):
  acc = collections.defaultdict(lambda: collections.defaultdict(list))
  for sub in ("sq", "grbm"):
    for f in glob.glob(os.path.join(P, "c%d_%s" % (c, sub), "*", "*_counter_collection.csv")):
      for r in csv.DictReader(open(f)):
        for k in KERNELS:
          if k in r["Kernel_Name"]:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
  for k, cnt in acc.items():
    def mean(name):
      v = cnt.get(name, [])
      v = v[len(v) // 4:]  # drop warm-up launches
      return sum(v) / len(v) if v else float("nan")
    waves = mean("SQ_WAVES")
    cycles = mean("GRBM_GUI_ACTIVE") / 8.0
    busy = mean("SQ_ACTIVE_INST_VALU") * 4.0 / (cycles * 1024.0)
    rec = {"launches": len(cnt.get("SQ_WAVES", [])), "waves": waves,
           "valu_per_wave": mean("SQ_INSTS_VALU") / waves, "salu_per_wave": mean("SQ_INSTS_SALU") / waves,
           "lds_per_wave": mean("SQ_INSTS_LDS") / waves, "branch_per_wave": mean("SQ_INSTS_BRANCH") / waves,
           "kernel_cycles": cycles, "valu_busy": busy,
           "source": "profiles/collect_coupled_pmc_r02.sh on MI355X, summarised by "
                     "profiles/summarize_coupled_r02.py"}
    out["config%d/%s" % (c, k)] = rec
    lines.append("| %d | %s | %d | %.0f | %.0f | %.0f | %.0f | %.0f | **%.2f** |" % (
        c, k, rec["launches"], waves, rec["valu_per_wave"], rec["salu_per_wave"],
        rec["lds_per_wave"], rec["branch_per_wave"], busy))
json.dump(out, open(os.path.join(ROOT, "profiles", "coupled_counters.json"), "w"), indent=1)
os.makedirs(os.path.join(ROOT, "profiles", "r02"), exist_ok=True)
open(os.path.join(ROOT, "profiles", "r02", "coupled_counters.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
