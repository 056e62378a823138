#!/bin/bash
# Kernel trace + timeline of the C2 (ndt) / C3 (tsdf) batches: gpurun_out/prof_<tag>_<mode>/{summary,timeline}.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r05}
for MODE in ndt tsdf; do
  OUT=gpurun_out/prof_${TAG}_${MODE}
  rm -rf $OUT; mkdir -p $OUT
  STEPS=6; [ "$MODE" = "tsdf" ] && STEPS=4
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python scripts/profile_modes.py $MODE $STEPS > $OUT/trace.log 2>&1
  python scripts/summarise_prof.py $OUT > $OUT/summary.txt 2>&1
  python scripts/timeline.py $OUT 1 > $OUT/timeline.txt 2>&1
  head -22 $OUT/summary.txt
  cat $OUT/timeline.txt | head -70
done
