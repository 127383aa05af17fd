#!/bin/bash
# round 3, run 12: SQ counters of the bf16x3 engine (MFMA-pipe utilisation, effective clock, wait buckets)
set -x
export TMPDIR=/tmp
O=gpurun_out/r03l
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $O/pmc_sq -o sq --output-format csv -- python3 bench.py --precision bf16x3 --legs none --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1
tail -3 $O/pmc_sq.log
ls $O/pmc_sq
python scripts/pmc_sq_summary.py $O/pmc_sq/sq_counter_collection.csv $O/pmc_sq/sq_kernel_trace.csv $O/r03_pmc_sq_bf16x3.json "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -- python3 bench.py --precision bf16x3 --legs none --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline"
head -3 $O/pmc_sq/sq_counter_collection.csv
rm -f $O/pmc_sq/sq_kernel_trace.csv $O/pmc_sq/sq_counter_collection.csv
