#!/bin/bash
# usage: profiles/tools/isa_resources.sh > profiles/r03/isa_resources.md
# Compiles every translation unit of libpymoc_hip.so with -save-temps (the product's flags) and
# lists each kernel's registers, scratch, LDS and occupancy as the assembler reports them.
REPO=$(cd "$(dirname "$0")/../.." && pwd)
T=/tmp/isa_res; [ -z "$REUSE" ] && rm -rf $T; mkdir -p $T; cd $T
echo "# ISA resource usage of every kernel (hipcc $(/opt/rocm/bin/hipcc --version | grep -o 'HIP version: [0-9.]*'), gfx950, product flags)"
echo
echo "| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | static LDS B | waves/SIMD |"
echo "|---|---|---|---|---|---|---|"
for f in pymoc_hip column_g16 column_g32 column_g64 equi equi_column jn2018_fast; do
  if [ -z "$REUSE" ] || [ ! -f $T/$f-hip-amdgcn-amd-amdhsa-gfx950.s ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-value \
      -save-temps -c -o $T/$f.o $REPO/pymoc_amd/csrc/$f.hip > /dev/null 2>&1
  fi
  S=$T/$f-hip-amdgcn-amd-amdhsa-gfx950.s
  # (the info block follows the kernel body: pair it with the most recent .amdhsa_kernel above it)
  python3 - "$S" <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read().split("\n")
cur = None; info = {}
rows = []
for ln in txt:
    m = re.match(r"\s*\.globl\s+(\S+)", ln)
    if m: cur = m.group(1)
    for key in ("NumVgprs", "NumAgprs", "TotalNumSgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
        m = re.match(r"; %s: (\d+)" % key, ln)
        if m: info[key] = int(m.group(1))
    if ln.startswith("; Occupancy") and cur:
        rows.append((cur, dict(info))); info = {}
seen = set()
for name, i in rows:
    if name in seen or "NumVgprs" not in i: continue
    seen.add(name)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.split("(")[0].replace("void ", "").replace("pm::", "")
    print("| `%s` | %d | %d | %d | %d | %d | %d |" % (dem, i["NumVgprs"], i.get("NumAgprs", 0), i["TotalNumSgprs"],
          i["ScratchSize"], i.get("LDSByteSize", 0), i["Occupancy"]))
PY
done
