#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/canon_check.py devlibs/push.so quick > gpurun_out/s6_check.txt 2>&1; echo "check rc=$?" >> gpurun_out/s6_check.txt
grep -c "bit-identical" gpurun_out/s6_check.txt; grep -i "differ\|rc=\|err\|C2" gpurun_out/s6_check.txt | tail -8
bash tools/r06_ab.sh s6 "pcg noise" devlibs/base.so devlibs/push.so devlibs/pushp1.so devlibs/pushat3.so devlibs/wait3.so
timeout 200 python tools/xcc_speed.py devlibs/pushxcc.so pcg 3 2>&1 | grep "statistics not"
