#!/usr/bin/env python3
"""gpurun_out/r05/ (profiles/collect_r05.sh) -> profiles/r05/: per config the rocprofv3
kernel statistics (copied), a counters table (summary.md) and counters.json, which bench.py
replays next to the numbers it measures live (labelled as replayed).

Per kernel and config, averaged over the dispatches of the run (first quarter dropped):
  avg_us            average duration, rocprofv3 --kernel-trace --stats (AverageNs)
  valu/salu/lds/branch per wave   SQ_INSTS_* / SQ_WAVES
  fp64 flop per launch            64 lanes x (SQ_INSTS_VALU_ADD_F64 + MUL_F64 + TRANS_F64 +
                                  2 SQ_INSTS_VALU_FMA_F64): flop the kernel ISSUED (padding lanes
                                  and speculative work included), not an algorithmic count
  valu_busy         SQ_ACTIVE_INST_VALU x 4 / (kernel cycles x 1024 SIMDs); SQ_ACTIVE_* count
                    quad-cycles (MI355X_MICROARCH.md); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
  issue_frac        SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: share of a resident wave's life spent
                    issuing vector instructions (both counters in quad-cycles: no factor)
  hbm_bytes         FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 per launch: the guide's gfx950
                    corrections (FETCH_SIZE counts 128-B requests at 64 B; WRITE_SIZE exact)"""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r05")
DST = os.path.join(ROOT, "profiles", "r05")
KERNELS = ("k_column_steps", "k_column_stream", "k_thermwind", "k_psi_so", "k_jn2018_fast",
           "k_jn2018_steps", "k_so_ml_step", "k_column_weff", "k_so_tw_update", "k_jn2018_ieee",
           "k_twocol_run", "k_jn2018_run", "k_jn2018_split", "k_rows_pack")
def short(name):
  n = name.split("(")[0].replace("void ", "").replace("pm::", "")
  return n.replace(" ", "")
out = {}
lines = ["# Round 5: rocprofv3 counters per config and kernel (MI355X)", "",
         "Collected by `profiles/collect_r05.sh`, summarised by `profiles/summarize_r05.py`; "
         "definitions in that script's docstring.  Kernel statistics of each run: "
         "`profiles/r05/<config>_kernel_stats.csv`.", "",
         "| config | kernel | launches | avg us | waves | VALU/wave | SALU/wave | LDS/wave | branch/wave | "
         "fp64 Gflop issued / launch | VALU busy | issue frac | HBM MB / launch |", "|" + "---|" * 13]
for cdir in sorted(glob.glob(os.path.join(SRC, "c*"))):
  cname = os.path.basename(cdir)
  stats = glob.glob(os.path.join(cdir, "trace", "**", "*_kernel_stats.csv"), recursive=True)
  avg = {}
  if stats:
    shutil.copy(stats[0], os.path.join(DST, "%s_kernel_stats.csv" % cname))
    for r in csv.DictReader(open(stats[0])):
      avg[short(r["Name"])] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["MinNs"]) / 1e3)
  bj = os.path.join(cdir, "bench.json")
  if os.path.exists(bj):
    txt = [l for l in open(bj).read().splitlines() if l.startswith("{")]
    if txt:
      open(os.path.join(DST, "%s_bench_under_rocprof.json" % cname), "w").write(txt[-1] + "\n")
  acc = collections.defaultdict(lambda: collections.defaultdict(list))
  for f in glob.glob(os.path.join(cdir, "pmc*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
      k = short(r["Kernel_Name"])
      if any(s in k for s in KERNELS):
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
  for k, cnt in sorted(acc.items()):
    def mean(name):
      v = cnt.get(name, [])
      v = v[len(v) // 4:]
      return sum(v) / len(v) if v else float("nan")
    waves = mean("SQ_WAVES")
    cycles = mean("GRBM_GUI_ACTIVE") / 8.0
    flop = 64.0 * (mean("SQ_INSTS_VALU_ADD_F64") + mean("SQ_INSTS_VALU_MUL_F64") +
                   mean("SQ_INSTS_VALU_TRANS_F64") + 2.0 * mean("SQ_INSTS_VALU_FMA_F64"))
    rec = {"launches": len(cnt.get("SQ_WAVES", [])), "waves": waves,
           "avg_us": avg.get(k, (float("nan"),))[0], "min_us": avg.get(k, (0, 0, float("nan")))[2],
           "valu_per_wave": mean("SQ_INSTS_VALU") / waves, "salu_per_wave": mean("SQ_INSTS_SALU") / waves,
           "lds_per_wave": mean("SQ_INSTS_LDS") / waves, "branch_per_wave": mean("SQ_INSTS_BRANCH") / waves,
           "fp64_add": mean("SQ_INSTS_VALU_ADD_F64"), "fp64_mul": mean("SQ_INSTS_VALU_MUL_F64"),
           "fp64_fma": mean("SQ_INSTS_VALU_FMA_F64"), "fp64_trans": mean("SQ_INSTS_VALU_TRANS_F64"),
           "fp64_flop_issued_per_launch": flop,
           "kernel_cycles": cycles, "valu_busy": mean("SQ_ACTIVE_INST_VALU") * 4.0 / (cycles * 1024.0),
           "issue_frac": mean("SQ_ACTIVE_INST_VALU") / mean("SQ_WAVE_CYCLES"),
           "wait_any_frac": mean("SQ_WAIT_ANY") / mean("SQ_WAVE_CYCLES"),
           "hbm_bytes_per_launch": 1024.0 * (2.0 * mean("FETCH_SIZE") + mean("WRITE_SIZE")),
           "fetch_size_kb": mean("FETCH_SIZE"), "write_size_kb": mean("WRITE_SIZE"),
           "source": "profiles/collect_r05.sh run '%s' on MI355X; profiles/summarize_r05.py" % cname}
    out["%s/%s" % (cname, k)] = rec
    lines.append("| %s | %s | %d | %.1f | %.0f | %.0f | %.0f | %.0f | %.0f | %.2f | %.2f | %.3f | %.2f |" % (
        cname, k, rec["launches"], rec["avg_us"], waves, rec["valu_per_wave"], rec["salu_per_wave"],
        rec["lds_per_wave"], rec["branch_per_wave"], flop / 1e9, rec["valu_busy"], rec["issue_frac"],
        rec["hbm_bytes_per_launch"] / 1e6))
os.makedirs(DST, exist_ok=True)
json.dump(out, open(os.path.join(DST, "counters.json"), "w"), indent=1)
open(os.path.join(DST, "summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
