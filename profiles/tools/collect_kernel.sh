#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:
#   profiles/tools/collect_kernel.sh <outdir-under-gpurun_out> <python script + args ...>
# One --kernel-trace --stats pass and three --pmc passes (never combined with tracing).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$1; shift
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $REPO/$*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/run_trace.log 2>> $OUT/err.txt
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -- $CMD > $OUT/run_pmc1.log 2>> $OUT/err.txt
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc2 -- $CMD > $OUT/run_pmc2.log 2>> $OUT/err.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_IFETCH --output-format csv -d $OUT/pmc3 -- $CMD > $OUT/run_pmc3.log 2>> $OUT/err.txt
find $OUT -name "*agent_info*" -delete
du -sh $OUT; tail -3 $OUT/err.txt
