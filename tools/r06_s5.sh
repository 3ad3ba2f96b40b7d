#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/canon_check.py devlibs/push.so quick > gpurun_out/s5_check.txt 2>&1; echo "check rc=$?" >> gpurun_out/s5_check.txt
tail -25 gpurun_out/s5_check.txt
bash tools/r06_ab.sh s5 "pcg" devlibs/base.so devlibs/push.so devlibs/pushp1.so devlibs/pushat0.so devlibs/pushat3.so devlibs/wait3.so
timeout 200 python tools/xcc_speed.py devlibs/pushxcc.so pcg 3 2>&1 | grep -v amdgpu | tail -8
