#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/canon_check.py devlibs/push2.so quick > gpurun_out/s8_check.txt 2>&1; echo "check rc=$?" >> gpurun_out/s8_check.txt
grep -c "bit-identical" gpurun_out/s8_check.txt; grep -i "differ\|rc=\|worst" gpurun_out/s8_check.txt | tail -6
bash tools/r06_ab.sh s8 "pcg noise" devlibs/base.so devlibs/push.so devlibs/push2.so devlibs/push2nw.so
timeout 200 python tools/xcc_speed.py devlibs/push2xcc.so pcg 2 2>&1 | grep "statistics not"
