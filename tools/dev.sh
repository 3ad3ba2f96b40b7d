#!/bin/bash
# usage: tools/dev.sh <name> [extra hipcc flags]  -- development build (nwin-128 kernels only) -> devlibs/<name>.so (git-ignored; travels to the GPU box), prints VGPRs / scratch / SGPR spills of the canonical-band kernels
name=$1; shift
mkdir -p /root/repo/devlibs
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DHSS_DEV -DHSS_DEV_ONLY128 "$@" -o /root/repo/devlibs/$name.so hssfsst.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs:|ScratchSize|SGPRs Spill" | grep -v "AGPRs\|VGPRs Spill" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | awk '/error/{print} /Function Name/{n=$3} /^ *VGPRs:/{v=$2} /ScratchSize/{sc=$3} /SGPRs Spill/{printf "%-90s vgpr %3s scratch %3s sgpr-spill %s\n", substr(n,1,90), v, sc, $3}' | sed 's/_ZN7hssfsst//' | grep "team1\|teamq\|canon"
