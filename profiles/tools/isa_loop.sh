#!/bin/bash
# usage: isa_loop.sh file.s N  -> prints "first last" line numbers of the N-th depth-2 loop:
# its header label .. the last block that names it as "in Loop: Header=" (rare-path blocks the
# compiler placed behind the latch are included; the hot path ends at the latch's branch)
F=$1; N=${2:-1}
H=$(grep -n "Loop Header: Depth=2" $F | sed -n ${N}p | cut -d: -f1)
LBL=$(sed -n "$((H-2)),$((H))p" $F | grep -o "^\.LBB[0-9_]*" | tail -1)
NAME=${LBL#.L}
# latch: first branch at or below the header that targets a label <= header (back edge)
LAST=$(awk -v h=$H -v lbl="$LBL" 'NR>h && ($1 ~ /^s_c?branch/) && $2==lbl {print NR}' $F | head -1)
if [ -z "$LAST" ]; then
  # back edge goes through a latch block placed above the header: take the branch to that block
  PRE=$(awk -v h=$H 'NR<h && /^\.LBB/ {l=$1} END{}; NR<h && /^\.LBB/ {last=$1} NR==h {sub(":","",last); print last}' $F)
  LAST=$(awk -v h=$H -v lbl="$PRE" 'NR>h && ($1 ~ /^s_c?branch/) && $2==lbl {print NR}' $F | head -1)
fi
echo $H $LAST
