// Interface between the C entry points (proj_render.hip) and the tuned 21-tap kernels (proj_render21.hip).
#pragma once
#include "common.h"

namespace m355 {

struct Render21Args {
    const int32_t *tile_start;  // [B, ntiles+1]   (m355_proj_bin_fwd)
    const float *tile_pts;      // [B, 4N] float4 records (c0,c1,c2,n)
    const float *scale;  // nullable
    const float *taps;   // NT taps, explicit
    float *proj;         // fwd
    const float *dproj;  // bwd
    float gmul;
    float *dcam_slots;   // bwd [B,N,4,3]
    float *dscale_part;  // bwd [B,nparts]
    int N, S, tiles_x, tiles_y;
    int fixed_weights;
    float empty_val;  // silhouette value of an untouched ray (render_empty_value(S))
    // M355_DET_SPLAT (non-zero): the occupancy splat accumulates round(w * 2^k) in 64-bit integer LDS cells (order-independent; k
    // chosen per tile from its record count, k_render21) and converts to fp32 once; 0 = LDS float atomics (the sum of >= 3 weights
    // in one voxel depends on their arrival order)
    float det_scale;
};

template <bool BWD>
int launch_render21(Render21Args a, int B, hipStream_t st);
float render_empty_value(int S);

}  // namespace m355
