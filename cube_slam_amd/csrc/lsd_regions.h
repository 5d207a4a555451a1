// lsd_regions.h -- device stage of the line detector between the level-line maps and the segments (lsd_regions.hip), called by lsd.hip
#pragma once
#include <vector>
struct cs_ctx;
struct LsdRegions;
int lsd_regions_run(cs_ctx *ctx, LsdRegions **handle, int F, int w, int h, const double *d_ang, const double *d_mod, const int *d_caddr, const int *frame_base,
                    std::vector<std::vector<float>> &lines, long *stats);
void lsd_regions_destroy(LsdRegions *r);
