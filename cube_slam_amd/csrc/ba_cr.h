// ba_cr.h -- nested-dissection (block cyclic reduction) solver of the banded reduced camera system (ba_cr.hip), called by ba.hip
#pragma once
#include "common.h"
struct BaCr;
bool ba_cr_supported(int C, int Bc);
int ba_cr_solve(cs_ctx *ctx, BaCr **handle, int C, int Bc, const double *d_bandA, const double *d_brhs, double *d_x, int *d_status);
void ba_cr_destroy(BaCr *w);
