#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/canon_check.py devlibs/tq.so quick > gpurun_out/s15_check.txt 2>&1; echo "check rc=$?" >> gpurun_out/s15_check.txt
grep -c "bit-identical" gpurun_out/s15_check.txt; grep -i "differ\|rc=\|worst\|error\|Traceback\|assert" gpurun_out/s15_check.txt | tail -8; tail -3 gpurun_out/s15_check.txt
bash tools/r06_ab.sh s15 "pcg noise" devlibs/base.so devlibs/push3.so devlibs/tq.so devlibs/push3nw.so
