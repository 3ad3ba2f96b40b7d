#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s3
for k in pcg zeros; do
echo "== xccwait3 (only resolvers wait) $k" >> gpurun_out/s3/xcc.txt
timeout 300 python tools/xcc_speed.py devlibs/xccwait3.so $k >> gpurun_out/s3/xcc.txt 2>&1
echo "== xcc (full waits) $k" >> gpurun_out/s3/xcc.txt
timeout 300 python tools/xcc_speed.py devlibs/xcc.so $k >> gpurun_out/s3/xcc.txt 2>&1
done
cat gpurun_out/s3/xcc.txt
