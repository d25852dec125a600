#!/usr/bin/env python3
"""Summarise gpurun_out/prof_r02/ (written by profiles/collect_r02.sh) into profiles/r02/ and
profiles/k1_counters.json (replayed, labelled as such, by bench.py).

Counter handling follows MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and
WRITE_SIZE are in KiB and come from separate --pmc passes; on gfx950 FETCH_SIZE under-reports
wide streaming reads, so both are calibrated on a run of the same kernel family whose true
traffic is known (one step per launch on 262144 columns = 1.26 GB per launch, far beyond L2 +
Infinity Cache).  SQ_*_CYCLES / SQ_ACTIVE_* count quad-cycles per wave."""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "gpurun_out", "prof_r02")
OUT = os.path.join(ROOT, "profiles", "r02")
os.makedirs(OUT, exist_ok=True)


def rows(pattern):
  out = []
  for f in glob.glob(os.path.join(P, pattern)):
    out += list(csv.DictReader(open(f)))
  return out


def counter_mean(sub, name, kernel, skip_frac=0.3):
  vals = [float(r["Counter_Value"]) for r in rows(sub + "/*/*_counter_collection.csv")
          if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
  vals = vals[int(len(vals) * skip_frac):]  # drop warm-up launches
  return (sum(vals) / len(vals), len(vals)) if vals else (float("nan"), 0)


for sub, name in (("trace", "bench_kernel_stats.csv"), ("cal_trace", "streaming_kernel_stats.csv")):
  fs = glob.glob(os.path.join(P, sub, "*", "*_kernel_stats.csv"))
  if fs:
    shutil.copy(fs[0], os.path.join(OUT, name))
for f, name in (("bench_traced.json", "bench_under_rocprof.json"),):
  if os.path.exists(os.path.join(P, f)):
    shutil.copy(os.path.join(P, f), os.path.join(OUT, name))

K = "k_column_steps"
C, nz, F = 1024, 100, 1000
sq = {n: counter_mean("pmc_sq", n, K)[0] for n in
      ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU",
       "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")}
sq2 = {n: counter_mean("pmc_sq2", n, K)[0] for n in
       ("SQ_INSTS_BRANCH", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY")}
fetch, nf = counter_mean("pmc_fetch", "FETCH_SIZE", K)
write, _ = counter_mean("pmc_write", "WRITE_SIZE", K)
SK = "k_column_stream"
cfetch, cn = counter_mean("cal_fetch", "FETCH_SIZE", SK)
cwrite, _ = counter_mean("cal_write", "WRITE_SIZE", SK)
Cb = 262144
true_read = 5 * 8 * nz * Cb + 8 * nz
true_write = 8 * nz * Cb
fcorr = true_read / (cfetch * 1024)
wcorr = true_write / (cwrite * 1024)
traffic = fetch * 1024 * fcorr + write * 1024 * wcorr
per = sq["SQ_WAVES"] * F
issue_frac = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]
grbm, _ = counter_mean("pmc_grbm", "GRBM_GUI_ACTIVE", K)
js = {"column_steps_F%d_C%d_nz%d" % (F, C, nz): {
    "issue_frac": issue_frac, "hbm_bytes_per_launch": traffic,
    "valu_insts_per_wave_step": sq["SQ_INSTS_VALU"] / per,
    "salu_insts_per_wave_step": sq["SQ_INSTS_SALU"] / per,
    "wave_cycles_per_step": 4 * sq["SQ_WAVE_CYCLES"] / per,
    "fetch_KiB": fetch, "write_KiB": write, "fetch_corr": fcorr, "write_corr": wcorr,
    "source": "profiles/collect_r02.sh on MI355X, summarised by profiles/summarize_r02.py"}}
json.dump(js, open(os.path.join(ROOT, "profiles", "k1_counters.json"), "w"), indent=1)

lines = ["# rocprofv3 summary, round 2 (MI355X, gfx950)", "",
         "## Kernel times of the default bench line (`rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline`)",
         "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
for r in rows("trace/*/*_kernel_stats.csv"):
  lines.append("| `%s` | %s | %.1f | %.1f | %.1f | %s |" % (
      r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
      float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
lines += ["", "## SQ counters of `k_column_steps<64,2,2,true>` (config 2: 1024 columns x nz=100, 1000 fused steps per launch)",
          "", "| counter | per launch | per wave and step |", "|---|---|---|"]
for n, v in list(sq.items()) + list(sq2.items()):
  lines.append("| %s | %.4g | %.2f |" % (n, v, v / per))
lines += ["", "VALU issue fraction = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = **%.3f**; %.1f vector + %.1f scalar "
          "instructions and %.0f cycles per wave and step; GRBM_GUI_ACTIVE %.3g cycles per launch summed over the 8 XCDs."
          % (issue_frac, sq["SQ_INSTS_VALU"] / per, sq["SQ_INSTS_SALU"] / per,
             4 * sq["SQ_WAVE_CYCLES"] / per, grbm), "",
          "## HBM-side traffic (PMC, separate passes)", "",
          "| run | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | launches averaged |", "|---|---|---|---|",
          "| config 2 fused (F=1000, k_column_steps) | %.1f | %.1f | %d |" % (fetch, write, nf),
          "| memory-bound regime (F=1, 262144 cols, k_column_stream) | %.1f | %.1f | %d |" % (cfetch, cwrite, cn), "",
          "Calibration on the memory-bound run (true bytes per launch %.4g read / %.4g written): correction "
          "**x%.3f (FETCH_SIZE)**, **x%.3f (WRITE_SIZE)**.  Corrected traffic of the fused config-2 launch: "
          "**%.3f MB per launch** (compulsory: 5 arrays in, 1 out = 4.9 MB) against %.1f MB of algorithmic bytes."
          % (true_read, true_write, fcorr, wcorr, traffic / 1e6, 24.0 * nz * C * F / 1e6), ""]
open(os.path.join(OUT, "summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
