#!/bin/bash
# rocprofv3 passes of the bench command on the GPU box (run through gpurun). Outputs under gpurun_out/prof_$1/.
# Pass 1: --kernel-trace --stats (per-kernel durations). Passes 2,3: PMC FETCH_SIZE / WRITE_SIZE, each in its own
# run with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2: not in one pass).
TAG=${1:-r01}; shift
ARGS=${@:-"--steps 5 --warmup 2"}
ARGS="$ARGS --no-secondary --cpu-pairs 0 --parity-pairs 0 --no-pmc --no-sequences"   # one workload per profile, nothing but the timed steps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python bench.py $ARGS > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python bench.py $ARGS > $OUT/bench_pmc_write.log 2>&1
# SQ pass (own run): VALU instructions issued / busy quad-cycles / wave residency of the LM-stage kernels
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/pmc_sq -o bench -- python bench.py $ARGS > $OUT/bench_pmc_sq.log 2>&1
tail -1 $OUT/bench_trace.log | cut -c1-400
find $OUT -name "*.csv" | head -20
# keep the merged payload small: drop the big per-dispatch traces except the ones we summarise
python tools/summarize_prof.py $OUT $TAG
# keep the merged payload small: the per-dispatch CSVs are summarised above
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
