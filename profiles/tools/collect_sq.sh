#!/bin/bash
# SQ / GRBM counter passes for the fp32 kernels:  bash profiles/tools/collect_sq.sh <tag> [workload ...]
# One rocprofv3 --pmc pass per counter group, --kernel-trace only (no sys / hip / hsa trace domains), the program directly
# after `--`.  Output: gpurun_out/sq_<tag>/<workload>/{sq1,sq2,grbm}/...counter_collection.csv + a condensed table
# (profiles/tools/summarise_sq.py).
set -o pipefail
tag=${1:-r03}; shift
wls=${@:-dccrn_cl dccrn_cl_train}
out=gpurun_out/sq_$tag
export TMPDIR=/tmp
mkdir -p "$out"
P="--kernel-trace --output-format csv"
for wl in $wls; do
  steps=2; [ "$wl" = dccrn_cl ] || steps=1
  args="bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline --no-alt"
  for grp in "sq1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES" \
             "sq2:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
             "grbm:GRBM_GUI_ACTIVE GRBM_COUNT"; do
    name=${grp%%:*}; ctrs=${grp#*:}
    d="$out/$wl/$name"
    rm -rf "$d"; mkdir -p "$d"
    timeout -k 10 300 rocprofv3 $P --pmc $ctrs -d "$d" -- python3 $args > "$d/bench.json" 2> "$d/bench.err" \
      || { echo "FAILED $wl $name"; tail -5 "$d/bench.err"; exit 1; }
    echo "$wl $name done"
  done
done
python3 profiles/tools/summarise_sq.py "$out"
