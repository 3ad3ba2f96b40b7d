#!/bin/bash
# round 6, first GPU session: where the time of the team kernel goes (development builds in devlibs/; results of the ablated builds are invalid)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s1
L="devlibs/base.so devlibs/abl3.so devlibs/wait3.so devlibs/abl3wait3.so devlibs/ilv.so devlibs/oneplane.so"
for inp in pcg noise zeros; do
  echo "== input $inp" >> gpurun_out/s1/ab.txt
  AB_INPUT=$inp AB_ROUNDS=7 AB_STEPS=20 timeout 600 python tools/ab_bench.py $L >> gpurun_out/s1/ab.txt 2>&1
done
timeout 900 python bench.py --steps 2000 --warmup 20 > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err
tail -c 3000 gpurun_out/s1/ab.txt
