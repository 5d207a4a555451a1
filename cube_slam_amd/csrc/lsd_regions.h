// lsd_regions.h -- device stage of the line detector between the level-line maps and the segments (lsd_regions.hip), called by lsd.hip
#pragma once
#include <vector>

#include <hip/hip_runtime.h>
struct cs_ctx;
struct LsdSeq;
int lsd_seq_run(cs_ctx *ctx, LsdSeq **handle, int F, int w, int h, const float *d_ang, const double *d_mod, const int *d_caddr, const float *d_cdeg, const float2 *d_ccs, const int *frame_base,
                std::vector<std::vector<float>> &lines, long *stats, void (*before_seq)(void *), void (*after_seq)(void *), void *gate_arg, int grp_p, int waves_per_workgroup = 16, void *scratch = nullptr,
                size_t scratch_bytes = 0, bool pix_ready = false, bool walk_bg = false);
void lsd_seq_destroy(LsdSeq *r);
