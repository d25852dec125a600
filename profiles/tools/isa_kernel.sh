#!/bin/bash
# usage: isa_kernel.sh file.hip mangled-prefix  -> compiles with -save-temps into /tmp/isa, prints
# the kernel's resource usage and extracts its listing to /tmp/isa/<prefix>.s
set -e
src=$1; pref=$2
mkdir -p /tmp/isa; cd /tmp/isa
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Wno-unused-value $EXTRA -save-temps -c -o /tmp/isa/$base.o $src 2>&1 | grep -v "^$" | grep -v "warning: argument unused" || true
S=$base-hip-amdgcn-amd-amdhsa-gfx950.s
awk -v k="^$pref" '$0 ~ k {f=1} f{print} /^\.Lfunc_end/{if(f){exit}}' $S > $pref.s
awk -v k="$pref" '/^; Kernel info/{getline l1; } /\.protected/{cur=$2} /NumVgprs|TotalNumSgprs|ScratchSize|Occupancy:/{ if (cur ~ k) print cur, $0 }' $S | sort -u
wc -l $pref.s
