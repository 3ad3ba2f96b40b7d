#!/bin/bash
# usage (GPU box): bash tools/r06_ab.sh <tag> "<inputs>" lib1 lib2 ...   -> gpurun_out/<tag>.txt
cd $GRAFT_REPO_ROOT; tag=$1; inputs=$2; shift; shift
for inp in $inputs; do
  echo "== input $inp" >> gpurun_out/$tag.txt
  AB_INPUT=$inp timeout 900 python tools/ab2.py "$@" 2>&1 | grep -v amdgpu.ids >> gpurun_out/$tag.txt
done
cat gpurun_out/$tag.txt
