#!/bin/bash
# usage: bash tools/profile_round.sh <tag> [extra bench args]   (on the GPU box, from the repo root)
# Runs the default bench, a rocprofv3 kernel-trace of the SAME command (minus the CPU leg) and PMC passes
# (FETCH_SIZE, WRITE_SIZE separately: they do not fit one pass on gfx950; short runs, counters do not depend
# on clocks), all into gpurun_out/<tag>/, then writes gpurun_out/<tag>/summary.json (kernel trace + PMC averages,
# tools/rocprof_summary.py) and gpurun_out/<tag>/pmc.json (per-kernel figures keyed to the kernel sources' SHA-256,
# the file bench.py quotes roofline.traffic / fp32_tflops from once it is copied to profiles/).
tag=${1:-prof}; shift
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
python bench.py "$@" > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
cd $R
S="--steps 5 --warmup 1 --settle-steps 0 --no-cpu-baseline --no-extras $@"
# (--no-extras: only the C2 launches, so that the per-kernel averages are those of the bench line's workload)
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --no-cpu-baseline --no-extras "$@" > $O/kt.log 2>&1
HSSFSST_NO_TEAM=1 rocprofv3 --kernel-trace --stats -d $O/ktt -o ktt -- python bench.py --no-cpu-baseline --no-extras "$@" > $O/ktt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch -- python bench.py $S > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o write -- python bench.py $S > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM -d $O/p1 -o p1 -- python bench.py $S > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $O/p2 -o p2 -- python bench.py $S > $O/p2.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/p3 -o p3 -- python bench.py $S > $O/p3.log 2>&1
python tools/rocprof_summary.py $O/summary_onecu.json $O/ktt/ktt_results.db > $O/summary_onecu.log 2>&1
python tools/rocprof_summary.py $O/summary.json $O/kt/kt_results.db $O/fetch/fetch_results.db $O/write/write_results.db $O/p1/p1_results.db $O/p2/p2_results.db $O/p3/p3_results.db > $O/summary.log 2>&1
python tools/pmc_traffic.py $O/summary.json $O/pmc.json > $O/pmc.log 2>&1
rm -rf $O/kt $O/ktt $O/fetch $O/write $O/p1 $O/p2 $O/p3     # the sqlite files are large; the summaries are what travels back
cat $O/bench.json
