// Launch arguments shared by the two builds of the flat-pixel-space 3x3 convolution (conv3x3_flat.hip: one 8-wave work-group
// per CU; conv3x3_flat2.hip: two co-resident 4-wave work-groups per CU).
#pragma once
#include "common.h"

struct FlatArgs {
  const char* x; const char* w; char* y; float* stats;
  long long P;                 // B*(H+2)*(W+2) padded pixels
  int B, H, W;
  int Cx, cx_off, K;
  int Cy, cy_off, N;
  int kc_count, flip;
  int persist;
  int carry;                   // persistent + one N tile: BatchNorm sums carried over the tiles, slab row = blockIdx.x
  int num_mtiles, num_ntiles;
  int total_tiles;             // num_mtiles * num_ntiles
  int lw;                      // row tiles: log2(W)
  int dil;                     // dilated row tiles (GEO = 2): dilation of the 3x3 taps; taps beyond the one-pixel halo read zeros
  const char* by; const float* bscale; const float* bshift;   // BatchNorm-backward sums in the stats slab (InsarBstat)
};

// conv3x3_flat2.hip: bf16; flat geometry, or (rows) row tiles (W = 2^k in 16 .. 256). `a` as flat_impl fills it;
// bn = 128 / 64 columns per tile.
int insar_flat2_launch(FlatArgs& a, int bn, bool bstat, bool rows, hipStream_t s);
bool insar_flat2_rows_geometry(const InsarAct& x);
// grid of a persistent flat2 launch (two work-groups per CU; three on 64-column tiles)
int insar_flat2_persistent_grid(int bn);
