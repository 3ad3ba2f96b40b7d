#!/bin/bash
# usage: tools/mk.sh <output.so> [extra hipcc flags]  -- builds libhssfsst from anywhere
out=$1; shift
cd /root/repo/heart_sounds_segmentation_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" -o "$out" hssfsst.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|core128" -A9 | grep -E "error|VGPRs:|Scratch|Occupancy"
