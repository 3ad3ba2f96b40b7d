#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python tools/canon_check.py devlibs/push4.so quick > gpurun_out/s14_check.txt 2>&1; echo "check rc=$?" >> gpurun_out/s14_check.txt
grep -c "bit-identical" gpurun_out/s14_check.txt; grep -i "differ\|rc=\|worst" gpurun_out/s14_check.txt | tail -6
python tools/blk_probe.py devlibs/push4blk.so pcg 2>&1 | grep -v amdgpu | tail -2
bash tools/r06_ab.sh s14 "pcg noise" devlibs/base.so devlibs/push3.so devlibs/push4.so devlibs/push3nw.so
