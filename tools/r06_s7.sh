#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python tools/xcc_speed.py devlibs/pushxcc.so pcg 3 2>&1 | grep "statistics not\|rep"
timeout 200 python tools/xcc_speed.py devlibs/pushxcc.so zeros 3 2>&1 | grep "statistics not"
