#!/bin/bash
# Usage (on the GPU box via gpurun): scripts/profile_bench.sh <tag> [pmc]
# Kernel trace + stats of the default bench command; with "pmc": counter passes in separate runs (never mixed with
# trace domains).  Summaries land in gpurun_out/prof_<tag>/summary.txt; copy what should be judged into profiles/.
set -u
TAG=${1:-r01}
MODE=${2:-trace}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="python bench.py --steps 20 --warmup 5 --no-cpu --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/bench_trace.log 2>&1
if [ "$MODE" = "pmc" ]; then
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -o pmc1 -- $CMD > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc2 -o pmc2 -- $CMD > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o pmc3 -- $CMD > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o pmc4 -- $CMD > $OUT/bench_pmc4.log 2>&1
fi
python scripts/summarise_prof.py $OUT > $OUT/summary.txt 2>&1
tail -2 $OUT/bench_trace.log | head -1 | cut -c1-400
cat $OUT/summary.txt
