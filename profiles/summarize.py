#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (written by profiles/collect.sh) into
profiles/<tag>_summary.md and profiles/traffic_<tag>.json (read by bench.py).

Counter handling follows MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in
KiB, collected in separate --pmc passes; on gfx950 FETCH_SIZE under-reports wide streaming
reads, so it is calibrated on a run of the SAME kernel whose true traffic is known (one
step per launch on 262144 columns = 1.26 GB per launch, far beyond L2 + Infinity Cache).
"""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof_" + tag)


def rows(pattern):
  out = []
  for f in glob.glob(os.path.join(P, pattern)):
    out += list(csv.DictReader(open(f)))
  return out


def counter_mean(sub, name, kernel="k_column_steps", skip=2):
  vals = [float(r["Counter_Value"]) for r in rows(sub + "/*/*_counter_collection.csv")
          if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
  vals = vals[skip:]  # drop warm-up launches
  return sum(vals) / len(vals), len(vals)


def kstats(sub):
  return [r for r in rows(sub + "/*/*_kernel_stats.csv")]


lines = ["# rocprofv3 summary `%s` (MI355X, gfx950)" % tag, ""]
for sub, title in (("trace", "bench.py default (1024 columns x nz=100, 1000 steps fused per launch)"),
                   ("cal_trace", "calibration: 262144 columns, 1 step per launch")):
  lines += ["## %s" % title, "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
  for r in kstats(sub):
    lines.append("| `%s` | %s | %.1f | %.1f | %.1f | %s |" % (
        r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]) / 1e3,
        float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
  lines.append("")

fetch, n = counter_mean("pmc_fetch", "FETCH_SIZE")
write, _ = counter_mean("pmc_write", "WRITE_SIZE")
cfetch, cn = counter_mean("cal_fetch", "FETCH_SIZE")
cwrite, _ = counter_mean("cal_write", "WRITE_SIZE")
C, nz = 262144, 100
true_read = 5 * 8 * nz * C + 8 * nz  # b, wA, kappa, area, dAkappa (+ the shared grid)
true_write = 8 * nz * C
fcorr = true_read / (cfetch * 1024)
wcorr = true_write / (cwrite * 1024)
traffic = fetch * 1024 * fcorr + write * 1024 * wcorr
alg = 24.0 * 100 * 1024 * 1000
lines += ["## HBM-side traffic of `k_column_steps` (PMC, separate passes)", "",
          "| run | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | launches averaged |",
          "|---|---|---|---|",
          "| bench default (F=1000) | %.1f | %.1f | %d |" % (fetch, write, n),
          "| calibration (F=1, 262144 cols) | %.1f | %.1f | %d |" % (cfetch, cwrite, cn), "",
          "Calibration: true bytes per launch = %.4g read / %.4g written; counters x1024 give "
          "%.4g / %.4g, so the correction factors for this kernel's access pattern are "
          "**x%.3f (FETCH_SIZE)** and **x%.3f (WRITE_SIZE)**." % (
              true_read, true_write, cfetch * 1024, cwrite * 1024, fcorr, wcorr), "",
          "Corrected HBM traffic of the benchmarked launch: **%.3f MB per launch** against "
          "%.1f MB of algorithmic bytes (24 B x nz x columns x 1000 fused steps): the state "
          "lives in registers for the whole launch, so only the compulsory first read and "
          "last write reach the memory side." % (traffic / 1e6, alg / 1e6), ""]
if os.path.exists(os.path.join(ROOT, "profiles", tag, "test_suite_kernel_stats.csv")):
  lines += ["## Kernel inventory of the GPU test suite", "",
            "`rocprofv3 --kernel-trace --stats -- python3 -m pytest tests -q -m gpu` -> "
            "`profiles/%s/test_suite_kernel_stats.csv`: every kernel the parity tests exercise "
            "(all `pm::k_*` of `libpymoc_hip.so`; the only foreign entries are the runtime's "
            "`__amd_rocclr_copyBuffer` / `fillBufferAligned` for memcpy / memset)." % tag, ""]
open(os.path.join(ROOT, "profiles", tag + "_summary.md"), "w").write("\n".join(lines))
json.dump({"column_steps_F1000_C1024_nz100": traffic,
           "_fetch_KiB": fetch, "_write_KiB": write, "_fetch_corr": fcorr, "_write_corr": wcorr},
          open(os.path.join(ROOT, "profiles", "traffic_%s.json" % tag), "w"), indent=1)
print("\n".join(lines))
