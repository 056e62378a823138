#!/bin/bash
# Usage (on the GPU box via gpurun): scripts/profile_modes.sh <tag> <ndt|tsdf> <steps>
# Kernel trace + stats of scripts/profile_modes.py, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes (never mixed
# with trace domains).  Summary in gpurun_out/prof_<tag>_<mode>/summary.txt.
set -u
TAG=${1:-r02}
MODE=${2:-ndt}
STEPS=${3:-4}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}_${MODE}
rm -rf $OUT; mkdir -p $OUT
CMD="python scripts/profile_modes.py $MODE $STEPS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/pmc4.log 2>&1
echo "command: $CMD" > $OUT/summary.txt
grep -h "rays_in" $OUT/trace.log | cut -c1-400 >> $OUT/summary.txt
python scripts/summarise_prof.py $OUT >> $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -40
