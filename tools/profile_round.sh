#!/bin/bash
# usage: bash tools/profile_round.sh <tag>   (on the GPU box, from the repo root)
# Runs the default bench, a rocprofv3 kernel-trace of the same command and two PMC passes
# (FETCH_SIZE, WRITE_SIZE separately: they do not fit one pass on gfx950), all into gpurun_out/<tag>/.
tag=${1:-prof}
R=$(pwd); O=$R/gpurun_out/$tag; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o write -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM -d $O/p1 -o p1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -d $O/p2 -o p2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/p2.log 2>&1
cat $O/bench.json
