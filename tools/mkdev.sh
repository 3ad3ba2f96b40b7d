#!/bin/bash
# usage: tools/mkdev.sh <output.so> [kernel-substring] [extra hipcc flags]  -- quick development build (nwin 128 kernels only)
out=$1; pat=${2:-canon_kernel}; shift; shift
cd /root/repo/heart_sounds_segmentation_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DHSS_DEV -DHSS_DEV_ONLY128 "$@" -o "$out" hssfsst.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|$pat" -A12 | grep -E "error|Function Name|VGPRs:|SGPRs:|Scratch|Spill" | sed 's/.*remark: *//'
