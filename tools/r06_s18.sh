#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python tools/canon_check.py devlibs/tq.so quick > gpurun_out/s18_check.txt 2>&1; echo "check rc=$?" >> gpurun_out/s18_check.txt
grep -c "bit-identical" gpurun_out/s18_check.txt; grep -i "differ\|rc=\|worst\|error\|Traceback\|assert" gpurun_out/s18_check.txt | tail -8
python tools/blk_probe.py devlibs/tqblk.so pcg 2>&1 | grep -v amdgpu | tail -2
bash tools/r06_ab.sh s18 "pcg noise" devlibs/base.so devlibs/tq.so devlibs/tqnw.so devlibs/push3nw.so
