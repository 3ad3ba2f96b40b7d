// fsst_gather.hpp -- frame list -> dense batch, the first stage of hssfsst_exec_list.
//
// The batched form of the reference's dataset loop (hss/datasets/heart_sounds.py:155-169 with the frames of
// hss/utils/preprocess.py:40-52) over MANY recordings: the recordings sit back to back in one device buffer, a list
// gives the first sample of every frame.  The frames are gathered into a dense [batch][n] buffer (8 kB read + 8 kB
// written per 2000-sample frame, against 360 kB of features) and then take the same transform kernels as a dense
// batch.  (A second addressing mode inside the transform kernels was built first: they sit at the 128-VGPR limit of
// their occupancy and it cost the 16-wave nwin = 128 kernels spilled registers.)
#pragma once
#include <hip/hip_runtime.h>

namespace hssfsst {

// out[b][i] = x[starts[b] + i], i < n.  One thread per four samples; the source is read with scalar loads (a frame
// may start at any sample), the destination is written as float4 when the row pitch allows.  Grid-stride.
__global__ __launch_bounds__(256) void fsst_gather_frames_kernel(const float* x, const long long* starts, float* out,
                                                                 long long batch, int n)
{
    const long long quads = (static_cast<long long>(n) + 3) / 4;
    const long long total = batch * quads;
    const bool wide = (n & 3) == 0;
    for (long long u = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; u < total; u += static_cast<long long>(gridDim.x) * 256) {
        const long long b = u / quads;
        const int i = static_cast<int>(u - b * quads) * 4;
        const float* src = x + starts[b] + i;
        float* dst = out + b * n + i;
        if (wide) {
            *reinterpret_cast<float4*>(dst) = make_float4(src[0], src[1], src[2], src[3]);
        } else {
            for (int e = 0; e < 4 && i + e < n; ++e) dst[e] = src[e];
        }
    }
}

}  // namespace hssfsst
