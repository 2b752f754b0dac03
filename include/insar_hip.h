/*
 * insar_hip.h — flat C ABI of libinsar_hip.so (gfx950 / MI355X).
 *
 * The reference (Createroner/InSAR-Unet-CA) has no FFI layer: its boundary is the
 * torch nn.Module API of Unet-ChannalAttention.py. This header is the thin C ABI the
 * Python host (insar_unet_ca_amd/) binds with ctypes; every entry point names the
 * reference arithmetic it replaces (file:line into /root/reference).
 *
 * Conventions
 *  - Every function returns 0 on success, a negative INSAR_E_* code or -hipError_t on
 *    failure; insar_last_error() returns a thread-local message. No C++ exception
 *    crosses the ABI.
 *  - The library never allocates, frees or synchronises: every buffer is caller-owned
 *    device memory (torch tensors passed as raw pointers); kernels are enqueued on the
 *    caller's hipStream_t (passed as void*) and the call returns immediately.
 *  - Activations are NHWC with a one-pixel zero halo: [B][H+2][W+2][C] elements of
 *    `dtype` (INSAR_F32 / INSAR_BF16). Kernels write interiors only, so the halo of a
 *    buffer allocated zeroed stays zero (that is what pads the 3x3 convolutions).
 *    An InsarAct describes a channel slice [c_off, c_off+c_len) of such a buffer, which
 *    is how the skip-concat (Unet-ChannalAttention.py:140,146,152,158) is zero-copy.
 */
#ifndef INSAR_HIP_H
#define INSAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define INSAR_ABI_VERSION 6

enum { INSAR_F32 = 0, INSAR_BF16 = 1 };

enum {
  INSAR_OK = 0,
  INSAR_E_SHAPE = -1001, /* unsupported / inconsistent shape */
  INSAR_E_DTYPE = -1002,
  INSAR_E_ALIGN = -1003, /* pointer not 16-byte aligned */
  INSAR_E_WS = -1004,    /* workspace too small */
  INSAR_E_ARG = -1005
};

typedef struct InsarAct {
  void* ptr;     /* base of the padded buffer (pixel (-1,-1) of image 0, channel 0) */
  int32_t B, H, W; /* interior extent */
  int32_t C;     /* channels of the whole buffer (pixel pitch, elements) */
  int32_t c_off; /* first channel of the slice */
  int32_t c_len; /* channels in the slice */
  int32_t dtype; /* INSAR_F32 | INSAR_BF16 */
  int32_t _pad;
} InsarAct;

int insar_version(void);
const char* insar_last_error(void);

/* Kernel-variant selectors for same-process A/B measurements (tools/, bench.py --tune): named integer knobs read by
 * the launchers at launch time. Every value of every knob selects between kernels that the parity tests hold to
 * the same tolerances; unknown names return INSAR_E_ARG. Not part of the reference's surface (it has no kernels). */
int insar_tune_set(const char* name, int32_t value);
int insar_tune_get(const char* name);   /* value, or INSAR_E_ARG */

/* ---- layout conversion at the nn.Module boundary (forward(x:[B,C,H,W]), :127) ------------ */
/* NCHW contiguous fp32 -> padded NHWC slice (cast to dst.dtype). */
int insar_pack_nchw(const float* src, const InsarAct* dst, void* stream);
/* padded NHWC slice -> NCHW contiguous fp32. */
int insar_unpack_nchw(const InsarAct* src, float* dst, void* stream);

/* ---- weight re-layout: torch fp32 parameter -> GEMM operand [t][n][k] of `dtype` ----------
 * out[(t*N + n)*K + k] = cast(in[t*st + n*sn + k*sk]).  Used for Conv2d (Co,Ci,3,3) weights
 * (:81,84), ConvTranspose2d (Ci,Co,2,2) weights (:112-121) in their forward and dgrad forms. */
int insar_weight_prep(const float* in, void* out, int32_t dtype, int32_t T, int32_t N, int32_t K,
                      int64_t st, int64_t sn, int64_t sk, void* stream);

/* batched form, one launch for every weight of the network. jobs (device): int64[njobs][10] =
 * {in*, out*, T, N, K, st, sn, sk, first_tile, dtype}; a job owns T*ceil(N/32)*ceil(K/32) consecutive tiles
 * starting at first_tile; total_tiles = sum over jobs. */
int insar_weight_prep_batch(const int64_t* jobs, int32_t njobs, int64_t total_tiles, void* stream);
/* Paired form: both GEMM layouts from one read of the fp32 master in[a][b][T] (T <= 9 taps contiguous):
 * out_ab[t][a][b] and out_ba[t][b][a]. jobs: int64[njobs][8] = {in*, out_ab*, out_ba*, A, B, T, first_tile,
 * dtype}; one work-group per 32x32 (a x b) tile, tiles numbered job after job. */
int insar_weight_prep_pair_batch(const int64_t* jobs, int32_t njobs, int64_t total_tiles, void* stream);

/* ---- implicit-GEMM convolution family (MFMA) ------------------------------------------------
 * y[pix(m), n] = sum_{tap, k} x[in_pix(m, tap), k] * w[tap][n][k]  (+ bias[n])
 *   m enumerates the out.B x Ho x Wo grid row-major; in_pix = (ho*stride + dy[tap], wo*stride + dx[tap]).
 * mode 0: Conv2d 3x3 pad 1 forward (:81,84) / its dgrad (flipped taps, transposed weights) /
 *         ConvTranspose2d dgrad (stride 2, taps {0,1}^2): out pixel (ho,wo), channel n.
 * mode 1: ConvTranspose2d(k=2,s=2) forward (:112,115,118,121): N = 4*Cout, n=(a*2+b)*Cout+co
 *         is scattered to out pixel (2ho+a, 2wo+b), channel co.
 * stats (nullable): per-M-tile partial column sums of the stored output,
 *         float[insar_igemm_num_mtiles(M, N)][2][N] (sum, sum of squares) for BatchNorm (:82,85). */
/* BatchNorm-backward sums taken in the epilogue of the GEMM that PRODUCES a unit's incoming gradient ("bstat"), instead of
 * a pass of its own over (dout, y) (insar_bnrelu_bwd_reduce; :82-83,85-86 inside loss.backward(), :345): with g = the
 * stored output element and yv = y at the same (pixel, channel),
 *   stats[row][0][n] = sum_pixels g * [scale[n]*yv + shift[n] > 0],   stats[row][1][n] = sum_pixels g * [...] * yv
 * in the slab the forward statistics would use (same rows). y = the consumer unit's raw conv output in a buffer with the
 * OUTPUT's layout (same B, H, W, C, c_off, dtype); y == NULL switches the mode off. */
typedef struct InsarBstat {
  const void* y;
  const float* scale;
  const float* shift;
} InsarBstat;

typedef struct InsarIgemm {
  InsarAct x;          /* input slice, c_len = K per tap */
  InsarAct y;          /* output slice */
  const void* w;       /* [ntaps][N][K] of x.dtype */
  const float* bias;   /* nullable, per output channel */
  float* stats;        /* nullable */
  int32_t N;           /* GEMM N (mode 1: 4*Cout) */
  int32_t Ho, Wo;      /* GEMM row grid (per image) */
  int32_t stride;      /* 1 or 2 */
  int32_t ntaps;       /* 1, 4 or 9 */
  int32_t mode;        /* 0 | 1 */
  int8_t dy[12], dx[12];
  int32_t flags;       /* INSAR_IGEMM_* */
  int32_t out_stride;  /* mode 0: 0 / 1 = dense output; s > 1: GEMM row (ho, wo) is stored at output pixel (ho*s + out_oy, */
  int32_t out_oy, out_ox; /*   wo*s + out_ox) — one parity class of the input gradient of a stride-s convolution */
  int32_t _pad;
  const void* add;     /* nullable, mode 0: a tensor with y's buffer layout (same C, c_off, dtype), added to the result */
  InsarBstat bstat;    /* mode 0, dense output, with stats: BatchNorm-backward sums instead of (sum, sum of squares) */
  const void* gate;    /* nullable, mode 0, no stats (ABI 4): a tensor with y's buffer layout; the result (after bias / add) is stored as
                        * zero where it is <= 0 — the ReLU mask of a residual block's input applied to the block's input gradient by
                        * the GEMM that writes it (torchvision Bottleneck `out = relu(out + identity)`, DeepLabV3-ChannelAttention.py:95) */
} InsarIgemm;
/* flags. OOB_ZERO: taps may leave the padded input and read zeros there (dilated 3x3 convolutions of DeepLabV3's
 * layer3 / layer4 / ASPP, torchvision resnet.py / deeplabv3.py; the 1-pixel halo covers only |dy|,|dx| <= 1). */
enum { INSAR_IGEMM_OOB_ZERO = 1,
       INSAR_IGEMM_PINGPONG = 2 /* 256 x 256 bf16 tiles: the ping-pong K loop (csrc/igemm.hip); same results, bit for bit */ };
/* rows of the stats slab = number of M tiles the library will use for a GEMM with M rows and N columns
 * (the tile height, 128 or 256 pixels, is chosen from M and N so that the grid fills the 256 CUs). */
int insar_igemm_num_mtiles(int64_t M, int32_t N);
int insar_igemm_tile_rows(int64_t M, int32_t N);
int insar_igemm_tile_cols(int64_t M, int32_t N);   /* 128 or 64 output channels per tile (any dtype) */
int insar_igemm_tile_cols_dt(int64_t M, int32_t N, int32_t dtype);   /* + 256 (bf16, N % 256 == 0, >= 256 such tiles) */
int insar_igemm(const InsarIgemm* d, void* stream);

/* ---- 3x3 / stride-1 convolution over the flat padded pixel space (MFMA) -------------------------------
 * Same arithmetic as insar_igemm mode 0 with the 9 forward taps (Conv2d 3x3 pad 1 forward, :81,84, and its
 * input gradient with flip = 1), for large grids: the three dx taps of a dy share one LDS tile of input
 * rows, so A rows cross the L2->LDS path 3 times instead of 9. x and y cover the same B x H x W grid.
 * flip: bit 0 = walk the taps backwards (input gradient); bit 1 = the ping-pong K loop (bf16; same results bit for bit);
 * bit 2 = persistent work-groups (one per CU, each walking its tiles in the same XCD-aware order; same outputs bit for bit,
 * the statistics summed per work-group instead of per tile).
 * w: [9][N][K] in the raster order produced by insar_weight_prep; stats (nullable):
 * float[insar_conv3x3_flat_stat_rows(x, N, flip)][2][N]. insar_conv3x3_flat_ok() is the library's own
 * eligibility heuristic (enough tiles to fill the chip, W,H >= 30). */
int insar_conv3x3_flat_ok(const InsarAct* x, int32_t N);
int insar_conv3x3_flat_num_mtiles(const InsarAct* x);
/* Row tiles (flip bit 3 = 8; bf16): a tile is 256 REAL output pixels = 256 / W whole image rows, staged with their halo
 * pixels, so that the three dx taps share one staged tile as in the flat geometry while the tile count is M / 256 (the deep
 * levels of the U-Net, where the flat geometry's 254-pixel step breaks the one-round-of-work-groups grid). Needs W a power
 * of two in 16..256 and H a multiple of 256 / W: insar_conv3x3_flat_rows_ok. flip bit 4 = 16: 64-column tiles whatever N.
 * The statistics slab has insar_conv3x3_flat_stat_rows(x, N, flip) rows, with the same bits. Row tiles need K and N multiples
 * of 64 and have no persistent form: flip bit 2 together with bit 3 is INSAR_E_ARG. */
int insar_conv3x3_flat_rows_ok(const InsarAct* x, int32_t N);
/* Row tiles of a DILATED 3x3 convolution (padding = dilation; flip bits 8-11 carry the dilation, 0 = 1): taps that reach beyond
 * the one-pixel halo read zeros, as insar_igemm's INSAR_IGEMM_OOB_ZERO. Needs 256 / W * (W + 2 * dil) <= 320 besides the above. */
int insar_conv3x3_flat_rows_dil_ok(const InsarAct* x, int32_t N, int32_t dil);
/* flip bit 5 = 32 (bf16; K a multiple of 32, N of 64): the kernel built as TWO co-resident 4-wave work-groups per CU
 * (csrc/conv3x3_flat2.hip: 32-channel K slabs, 128 x 64 wave tiles; one group's prologue / epilogue under the other's K loop).
 * Flat geometry, or with bit 3 row tiles (W a power of two in 16..256, H a multiple of 256 / W: insar_conv3x3_flat2_rows_ok;
 * the halo pixels are not staged); bit 2 = persistent work-groups (two per CU) in both geometries. Results equal the 8-wave kernel's up to fp32
 * summation order (channels are added 32 by 32 instead of 64 by 64 per tap). Dilation bits with bit 5: INSAR_E_ARG. */
int insar_conv3x3_flat2_rows_ok(const InsarAct* x, int32_t N);
/* rows of the statistics slab for a launch with these flags: one per M tile, or one per work-group for persistent
 * work-groups (flip bit 2) with a single N tile, which carry the sums over their tiles */
int insar_conv3x3_flat_stat_rows(const InsarAct* x, int32_t N, int32_t flip);
int insar_conv3x3_flat(const InsarAct* x, const InsarAct* y, const void* w, int32_t flip, float* stats,
                       void* stream);
/* the same with BatchNorm-backward sums in the statistics slab (InsarBstat above; bstat->y has y's layout) */
int insar_conv3x3_flat_bstat(const InsarAct* x, const InsarAct* y, const void* w, int32_t flip, float* stats,
                             const InsarBstat* bstat, void* stream);

/* ---- 3x3 conv with 64 -> 64 channels, bf16 (the full-resolution level; :81,84 and their dgrad) -------
 * Persistent work-groups, weights in registers, activations through a rolling LDS window over the flat
 * padded pixel space (csrc/conv3x3_c64.hip). Same operands and tap order as insar_conv3x3_flat.
 * stats: [insar_conv3x3_c64_rows(x)][2][64] BatchNorm partial sums (one row per work-group) or null. */
int insar_conv3x3_c64_ok(const InsarAct* x, int32_t N);
int insar_conv3x3_c64_rows(const InsarAct* x);
/* host-side query of the rolling-window geometry: out[6] = {A, Af, o, R, ntiles, TILE}; 1 if the shape is accepted */
int insar_conv3x3_c64_geometry(const InsarAct* x, int32_t* out);
int insar_conv3x3_c64(const InsarAct* x, const InsarAct* y, const void* w, int32_t flip, float* stats,
                      void* stream);
/* the same with BatchNorm-backward sums in the statistics slab (InsarBstat; scale / shift 16-byte aligned) */
int insar_conv3x3_c64_bstat(const InsarAct* x, const InsarAct* y, const void* w, int32_t flip, float* stats,
                            const InsarBstat* bstat, void* stream);

/* ---- weight-gradient GEMM (MFMA, split-K over pixels, no atomics) ------------------------------
 * part[split][tap][co][ci] = sum_{p in split} dy[pixB(p,tap), co] * x[pixA(p,tap), ci]
 * with pixA/pixB given as int32 pixel-index tables (padded-buffer pixel index of tap (0,0) for
 * every flattened p, length Mpad = multiple of 64, tail entries 0 = a halo pixel) plus a per-tap
 * pixel offset. Covers Conv2d wgrad (x taps move) and ConvTranspose2d wgrad (dy taps move). */
typedef struct InsarWgrad {
  InsarAct x;             /* c_len = Cin */
  InsarAct dy;            /* c_len = Cout */
  const int32_t* tabx;    /* [Mpad] */
  const int32_t* tabdy;   /* [Mpad] */
  float* part;            /* [nsplit][ntaps][Cout][Cin] fp32 */
  int64_t Mpad;           /* padded pixel count (multiple of 64) */
  int32_t nsplit;
  int32_t ntaps;
  int32_t offx[12];       /* per-tap pixel offset added to tabx entries */
  int32_t offdy[12];      /* per-tap pixel offset added to tabdy entries */
  int64_t tabx_tap_stride; /* 0: tabx serves every tap (offx moves it); else tap t reads tabx + t*tabx_tap_stride (per-tap
                            * tables from insar_pixel_table_taps: strided / dilated convolutions, out-of-bounds -> pixel 0) */
} InsarWgrad;
/* tile extent the kernel uses along a channel dimension of C channels: 64 | 128 | 256 (256 only when both
 * Cin and Cout allow it, otherwise capped at 128); needed by callers that size the split-K factor. */
int insar_wgrad_tile(int32_t C, int32_t dtype);
/* the (Cin, Cout) tile of a layer as (tile(Cin) << 16) | tile(Cout): 256x256 (8 waves), 128x128, 128x64, 64x128,
 * 64x64 in bf16; 128x128 (8 waves) or 64x64 in fp32. */
int insar_wgrad_tile_pair(int32_t Cin, int32_t Cout, int32_t dtype);
int insar_wgrad(const InsarWgrad* d, void* stream);
/* The same slabs for a 3x3 / stride-1 / pad-1 convolution (Unet-ChannalAttention.py:81,84; tap = 3*ty + tx) from a
 * kernel whose work-groups compute the three taps of a kernel row together: dy staged once per 64-pixel K step, x
 * once with a one-pixel lead and tail, a third of the LDS-DMA pieces of insar_wgrad (which these layers are bound by).
 * x (B,H,W,Cin), dy (B,H,W,Cout) on the same grid, part[nsplit][9][Cout][Cin]. insar_wgrad_conv3_tile returns
 * (tile(Cin) << 16) | tile(Cout) (tiles <= 128), or 0 when the layer needs insar_wgrad (W neither a multiple of 64
 * nor 16 / 32). */
int insar_wgrad_conv3_tile(const InsarAct* x, int32_t Cout);
int insar_wgrad_conv3(const InsarAct* x, const InsarAct* dy, float* part, int32_t nsplit, void* stream);
/* The same slabs again (bit for bit insar_wgrad_conv3's at the same nsplit) for the bf16 layers with >= 256 channels on one
 * side and >= 128 on the other, from 256 x 128 (Cin x Cout; 128 x 256 where only Cout has 256) tiles: 8 waves of 64 x 64 x
 * three taps, a K step as six [fragment reads + LDS-DMA issue | 16 MFMAs] phases with the two waves of a SIMD in opposite
 * roles, a three-slot LDS ring whose only vector-memory wait is a counted one per K step (csrc/wgrad3x.hip).
 * insar_wgrad_conv3x_tile: (tile(Cin) << 16) | tile(Cout), or 0 where the layer keeps insar_wgrad_conv3. */
int insar_wgrad_conv3x_tile(const InsarAct* x, int32_t Cout);
int insar_wgrad_conv3x(const InsarAct* x, const InsarAct* dy, float* part, int32_t nsplit, void* stream);
/* ... and from 128 x 128 tiles by 4-wave work-groups built to run TWO PER CU (csrc/wgrad3y.hip: the same 64 x 64 x three-tap wave
 * tiles and LDS image, a two-slot LDS ring of 34 KB stages, one barrier per K step, the two work-groups of a CU not coupled at all):
 * bf16 layers with both channel counts multiples of 128. insar_wgrad_conv3y_tile: (128 << 16) | 128, or 0. */
int insar_wgrad_conv3y_tile(const InsarAct* x, int32_t Cout);
int insar_wgrad_conv3y(const InsarAct* x, const InsarAct* dy, float* part, int32_t nsplit, void* stream);
/* The same decomposition for the bf16 layers with 64 or 128 channels on BOTH sides (the 256^2 / 128^2 levels): the tile is
 * too small to give eight waves a 64 x 64 x three-tap tile each, so the waves of a work-group split the PIXELS of a K step
 * (KS = 8 / wave tiles slices of 32 pixels) and a work-group writes KS slabs: part[nsplit * KS][9][Cout][Cin], folded by
 * insar_wgrad_reduce over nsplit * KS slabs (csrc/wgrad3k.hip). A K step (KS * 32 pixels) must lie inside one image row:
 * insar_wgrad_conv3k_tile returns (tile(Cin) << 16) | tile(Cout) or 0 (insar_wgrad_conv3 then), _slices returns KS.
 * The sums are partitioned differently from insar_wgrad_conv3's: equal to fp32 summation order, not bit for bit. */
int insar_wgrad_conv3k_tile(const InsarAct* x, int32_t Cout);
int insar_wgrad_conv3k_slices(const InsarAct* x, int32_t Cout);
int insar_wgrad_conv3k(const InsarAct* x, const InsarAct* dy, float* part, int32_t nsplit, void* stream);
/* grad = sum_split part[...] re-laid out to the torch parameter layout.
 * layout 0: Conv2d (Co,Ci,kh,kw): grad[(co*Ci+ci)*ntaps + tap]
 * layout 1: ConvTranspose2d (Ci,Co,2,2): grad[(ci*Co+co)*ntaps + tap]
 * accumulate != 0 adds into grad instead of overwriting. */
int insar_wgrad_reduce(const float* part, float* grad, int32_t nsplit, int32_t ntaps, int32_t Co,
                       int32_t Ci, int32_t layout, int32_t accumulate, void* stream);
/* first stage of the fold when nsplit is large: part_out[g] = sum of `group` consecutive slabs of
 * part_in (slab_floats = ntaps*Co*Ci); the result feeds insar_wgrad_reduce with nsplit = ceil(nsplit/group). */
int insar_wgrad_fold(const float* part_in, float* part_out, int64_t slab_floats, int32_t nsplit,
                     int32_t group, void* stream);
/* pixel-index table for a B x H x W grid mapped with stride s into a padded buffer of interior
 * (Hb, Wb): tab[p] = (n*(Hb+2) + h*s + 1)*(Wb+2) + w*s + 1 ; entries p >= B*H*W are `tail`
 * (0 = a zero halo pixel for the operand whose taps do not move; Wb+3 = the first interior pixel,
 * in bounds under every 3x3 tap, for the operand whose taps move). */
int insar_pixel_table(int32_t* tab, int64_t Mpad, int32_t B, int32_t H, int32_t W, int32_t s,
                      int32_t Hb, int32_t Wb, int32_t tail, void* stream);

/* ---- first layer: direct 3x3 conv for tiny Cin (inc.double_conv.0, Cin<=4; :81 with :464) ---- */
/* rows of the statistics slab insar_conv3x3_small_fwd writes for this (x, y): stats[rows][2][Cout] */
int insar_conv3x3_small_fwd_rows(const InsarAct* x, const InsarAct* y);
int insar_conv3x3_small_fwd(const InsarAct* x, const float* w /*torch (Co,Ci,3,3) fp32*/,
                            const InsarAct* y, float* stats /*[B*H][2][Co]*/, void* stream);
/* part: [insar_conv3x3_small_wgrad_blocks(B,H)][Co*Ci*9] partial rows (torch (Co,Ci,3,3) order). */
int insar_conv3x3_small_wgrad_blocks(int32_t B, int32_t H);
int insar_conv3x3_small_wgrad(const InsarAct* x, const InsarAct* dy, float* part, void* stream);
/* The same with the unit's BatchNorm / ReLU backward apply pass (insar_bnrelu_bwd_apply without an SE gate) evaluated on the
 * way in: dy — which only this launch would read, the network's first layer having no input gradient (:81 under :345) — is never
 * written. g: the unit's incoming gradient, y: its conv output; bf16, Cin = 2, Cout = 64, W % 64 == 0 (insar_..._fused_ok).
 * Bit for bit the two launches it replaces. */
int insar_conv3x3_small_wgrad_fused_ok(const InsarAct* x, const InsarAct* y);
int insar_conv3x3_small_wgrad_fused(const InsarAct* x, const InsarAct* g, const InsarAct* y, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, const float* k1,
                                    const float* k2, int32_t relu, float* part, void* stream);

/* ---- segmented column sum of partial slabs: out[s][c] (+)= sum_r part[s][r][c] ---------------
 * Two-stage (deterministic, no atomics) when rows > 256: stage 1 writes into `tmp`
 * (>= ceil(rows/128)*segments*cols floats), stage 2 folds it. */
int insar_colsum(const float* part, float* out, int32_t segments, int64_t rows, int32_t cols,
                 int32_t accumulate, float* tmp, int64_t tmp_floats, void* stream);
/* The same over the first `cols` columns of rows that are `ld` >= cols floats apart: a bias / 1x1-weight gradient summed
 * straight into its place in the flat gradient buffer (no staging tensor and no device copy behind it). */
int insar_colsum_ld(const float* part, float* out, int32_t segments, int64_t rows, int32_t cols, int64_t ld,
                    int32_t accumulate, float* tmp, int64_t tmp_floats, void* stream);

/* first stage only: out[ceil(rows/rps)][cols]; insar_bn_finalize folds the remaining rows itself. */
int insar_colsum_partial(const float* part, float* out, int64_t rows, int32_t cols, int32_t rps,
                         void* stream);

/* ---- BatchNorm2d (+ReLU) (:82-83, :85-86) -------------------------------------------------------
 * finalize: from `rows` (<= 4096) remaining partial rows part[rows][2][C] (sum, sum of squares) of
 * the raw (bias-free) conv output over `count` pixels (larger slabs go through insar_colsum first): batch mean/var -> scale = gamma*invstd, shift = beta - mean*scale; running stats
 * (momentum, unbiased var; the conv bias is added to the mean) and num_batches_tracked.
 * training == 0: scale/shift from the running stats (+ conv bias), slabs ignored. */
typedef struct InsarBnFinalize {
  const float* part; int64_t rows; int64_t count; int32_t C; int32_t training;
  const float* conv_bias; const float* gamma; const float* beta;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float momentum; float eps;
  float* scale; float* shift; float* mean; float* invstd; /* outputs, [C] each */
} InsarBnFinalize;
int insar_bn_finalize(const InsarBnFinalize* d, void* stream);
/* z = relu(y*scale + shift) * gate[n][c] (gate nullable) -> dst slice. `relu` = 0 drops the ReLU
 * (and the mask in the reductions below): that is the stand-alone SELayer applied to a raw tensor. */
int insar_bn_relu_apply(const InsarAct* y, const float* scale, const float* shift,
                        const float* gate /*[B][C] or null*/, const InsarAct* dst, int32_t relu,
                        void* stream);

/* the same pass for an encoder block, writing the 2x2 max-pool (:106-109) of dst as well: pooled is the
 * (B, H/2, W/2, C) slice; identical to insar_bn_relu_apply followed by insar_maxpool2_fwd, one read less. */
int insar_bn_relu_apply_pool(const InsarAct* y, const float* scale, const float* shift, const float* gate,
                             const InsarAct* dst, const InsarAct* pooled, int32_t relu, void* stream);

/* ---- SELayer (:45-72) ----------------------------------------------------------------------------
 * squeeze: partial sums of mask and mask*y, mask = (y*scale+shift > 0), over `rows_per_part` consecutive
 *   image rows: part[B][P][2][C], P = ceil(H / rows_per_part). */
int insar_se_squeeze(const InsarAct* y, const float* scale, const float* shift, float* part,
                     int32_t relu, int32_t rows_per_part, void* stream);
/* excitation: mean -> Linear(C,C/r) -> ReLU -> Linear(C/r,C) -> Sigmoid (two bias-free Linears,
 * :54-59). Saves sq[B][C] (the squeezed mean), hid[B][Cr] (post-ReLU), gate[B][C]. */
typedef struct InsarSeFwd {
  const float* part;   /* squeeze slabs [B][rows][2][C] (insar_se_squeeze: rows = P), folded in-kernel */
  int32_t B, H, W, C, Cr; int32_t rows;
  const float* scale; const float* shift;
  const float* w1; /* (Cr, C) */ const float* w2; /* (C, Cr) */
  float* pooled; /* out [B][2][C]: per image sum of mask, sum of mask*y (kept for backward) */
  float* sq; float* hid; float* gate;
} InsarSeFwd;
int insar_se_excite(const InsarSeFwd* d, void* stream);

/* ---- backward of [BN -> ReLU -> (SE gate)] ------------------------------------------------------
 * reduce: part[B][P][2][C] = sums over `rows_per_part` image rows of  g*mask  and  g*mask*y , g = dout (T),
 *   P = ceil(H / rows_per_part).  */
int insar_bnrelu_bwd_reduce(const InsarAct* dout, const InsarAct* y, const float* scale,
                            const float* shift, float* part, int32_t relu, int32_t rows_per_part, void* stream);
/* coefficient kernels: turn the reduce slabs red[B][rows][2][C] (insar_bnrelu_bwd_reduce: rows = P;
 * folded per image in-kernel) into everything the apply pass needs.
 *  with SE:  ds -> MLP backward (dW1, dW2, dsq);  g_eff = (dout*gate + dsq/HW) on the ReLU mask
 *  dgamma/dbeta; coefB[B][C] = dsq/HW and per-channel k1[C] = dbeta/N, k2[C] = dgamma/N (0 in eval):
 *    dy = scale_c * ( (dout*gate + coefB)*mask - k1 - xhat*k2 ).
 *  dconv_bias (nullable): gradient of the bias of the conv feeding this BN: exactly 0 in training
 *  (the batch mean removes it), scale*dbeta in eval.
 *  ws: float[B*(3C + Cr)] scratch. */
typedef struct InsarBnSeBwd {
  int32_t B, H, W, C, Cr; int32_t use_se;
  const float* mean; const float* invstd;
  const float* pooled; const float* sq; const float* hid; const float* gate;
  const float* w1; const float* w2;
  float* dw1; float* dw2; float* dgamma; float* dbeta;
  float* coefB; float* k1; float* k2;
  int32_t accumulate; int32_t _pad;
} InsarBnSeBwd;
int insar_bnse_bwd_coef(const InsarBnSeBwd* d, const float* red, int32_t rows, const float* scale,
                        const float* shift, float* ws, float* dconv_bias, int32_t training, void* stream);
/* The same in ONE launch: the work-group that finishes stage 1 last (a ticket counter; the hand-off data written through
 * to memory and read past the caches) runs stage 2. ticket: a zero-initialised 32-bit device word per call site (the
 * kernel resets it); bitwise the results of insar_bnse_bwd_coef. Pays for units without an SE gate, whose stage 2 is small. */
int insar_bnse_bwd_coef_fused(const InsarBnSeBwd* d, const float* red, int32_t rows, const float* scale,
                              const float* shift, float* ws, float* dconv_bias, int32_t training, uint32_t* ticket,
                              void* stream);
/* Units WITHOUT an SE gate (d->use_se == 0): the coefficients (k1, k2, dgamma, dbeta, conv-bias gradient) from ALL
 * `rows_total` rows of the reduction slab red[rows_total][2][C] in one channel-parallel launch; any partition of the pixels
 * into rows will do (BatchNorm backward, :82-83,85-86 inside loss.backward(), :345). */
int insar_bn_bwd_coef(const InsarBnSeBwd* d, const float* red, int64_t rows_total, const float* scale,
                      float* dconv_bias, int32_t training, void* stream);
int insar_bnrelu_bwd_apply(const InsarAct* dout, const InsarAct* y, const float* scale,
                           const float* shift, const float* mean, const float* invstd,
                           const float* gate, const float* coefB, const float* k1,
                           const float* k2, const InsarAct* dy, int32_t relu, void* stream);
/* The two stages of insar_bnse_bwd_coef on their own (stage = 1: per-image stage — SE backward, coefB and the
 * per-image partial sums tb/tg in ws; stage = 2: batch fold — k1, k2, dgamma, dbeta, dW1, dW2, conv-bias gradient),
 * and the apply pass that folds k1[c] = sum_n tb[n][c] / (B*H*W), k2[c] = sum_n tg[n][c] / (B*H*W) itself
 * (training mode; tb = ws + B*(C+Cr), tg = tb + B*C; C <= 1024). With these the input-gradient chain
 * (loss.backward(), Unet-ChannalAttention.py:345) needs only stage 1; stage 2 runs beside it on another stream.
 * Bitwise the same dy as insar_bnse_bwd_coef + insar_bnrelu_bwd_apply. */
int insar_bnse_bwd_coef_stage(const InsarBnSeBwd* d, const float* red, int32_t rows, const float* scale,
                              const float* shift, float* ws, float* dconv_bias, int32_t training, int32_t stage,
                              void* stream);
int insar_bnrelu_bwd_apply_part(const InsarAct* dout, const InsarAct* y, const float* scale, const float* shift,
                                const float* mean, const float* invstd, const float* gate, const float* coefB,
                                const float* tb, const float* tg, const InsarAct* dy, int32_t relu, void* stream);

/* ---- MaxPool2d(2) backward inside the encoder block's BatchNorm-backward passes -----------------------------------
 * insar_bn_relu_apply_pool_arg is insar_bn_relu_apply_pool that also records which element of each 2x2 window is
 * the maximum (arg[B][H/2][W/2][C] bytes 0..3 = 2*(row parity) + (column parity); insar_maxpool2_bwd's rule on the
 * stored values). The _pool variants of the reduce and apply passes then take
 *   dout = round_T(dskip + (arg == position ? dpooled : 0))
 * on the fly — the expression insar_maxpool2_bwd evaluates when it accumulates into dskip — so that launch (a read of
 * the full-resolution activation and a read-modify-write of its gradient) is not needed. Bitwise the same results. */
int insar_bn_relu_apply_pool_arg(const InsarAct* y, const float* scale, const float* shift, const float* gate,
                                 const InsarAct* dst, const InsarAct* pooled, uint8_t* arg, int32_t relu, void* stream);
int insar_bnrelu_bwd_reduce_pool(const InsarAct* dskip, const InsarAct* dpooled, const uint8_t* arg, const InsarAct* y,
                                 const float* scale, const float* shift, float* part, int32_t relu,
                                 int32_t rows_per_part, void* stream);
int insar_bnrelu_bwd_apply_pool(const InsarAct* dskip, const InsarAct* dpooled, const uint8_t* arg, const InsarAct* y,
                                const float* scale, const float* shift, const float* mean, const float* invstd,
                                const float* gate, const float* coefB, const float* k1, const float* k2,
                                const InsarAct* dy, int32_t relu, void* stream);

/* ---- the unit that feeds the 1x1 output conv (outc, Unet-ChannalAttention.py:125,162) ----------------------------
 * Its incoming gradient is g[n,h,w,c] = round_T(sum_k dlogits[n,k,h,w] * W[k][c]): 2K multiply-adds per element, so the
 * reduce and apply passes recompute it from dlogits (fp32 [B][K][H][W], K <= 4) and outc's weight ([K][C] fp32)
 * instead of reading a materialised 64-channel tensor, and insar_conv1x1_out_wgrad produces outc's parameter
 * gradients (part as insar_conv1x1_out_bwd) without writing that tensor. Bitwise the same results as
 * insar_conv1x1_out_bwd + insar_bnrelu_bwd_reduce + insar_bnrelu_bwd_apply, three activation-sized HBM passes less. */
int insar_conv1x1_out_wgrad(const InsarAct* x, const float* w, const float* dlogits, int32_t K, float* part, void* stream);
/* Forward counterpart: the last unit's BN/ReLU/gate pass hands z straight to outc and writes only the logits
 * (fp32 [B][K][H][W], K <= 4; bitwise insar_bn_relu_apply + insar_conv1x1_out_fwd); outc's parameter gradients then
 * come from insar_conv1x1_out_wgrad_y, which recomputes z = round(relu(y*scale+shift) * gate[n]) from y. */
int insar_bn_relu_apply_outc(const InsarAct* y, const float* scale, const float* shift, const float* gate /*nullable*/,
                             const float* wout, const float* bias /*nullable*/, float* logits, int32_t K, int32_t relu,
                             void* stream);
int insar_conv1x1_out_wgrad_y(const InsarAct* y, const float* scale, const float* shift, const float* gate /*nullable*/,
                              const float* w, const float* dlogits, int32_t K, float* part, void* stream);
/* wpart (nullable; gate [B][C] nullable): the same pass also writes the output conv's parameter-gradient partials,
 * wpart[B * ceil(H / rows_per_part)][K*C + K] in the layout of insar_conv1x1_out_wgrad_y's `part` (fold with insar_colsum):
 * the weight gradient of the output conv then needs no pass of its own over y. */
int insar_bnrelu_bwd_reduce_outc(const float* dlogits, const float* wout, int32_t K, const InsarAct* y,
                                 const float* scale, const float* shift, float* part, int32_t relu,
                                 int32_t rows_per_part, const float* gate, float* wpart, void* stream);
int insar_bnrelu_bwd_apply_outc(const float* dlogits, const float* wout, int32_t K, const InsarAct* y,
                                const float* scale, const float* shift, const float* mean, const float* invstd,
                                const float* gate, const float* coefB, const float* k1, const float* k2,
                                const InsarAct* dy, int32_t relu, void* stream);

/* ---- ChannelAttentionModule (DeepLabV3-ChannelAttention.py:49-79; config 5) ---------------------------
 * out = x * sigmoid(W2 relu(W1 avg_hw(x)) + W2 relu(W1 max_hw(x))), W1 (Cr,C), W2 (C,Cr), no biases.
 * forward : insar_cam_pool -> insar_cam_excite -> insar_bn_relu_apply(x, ones, zeros, gate, out, relu=0)
 * backward: insar_bnrelu_bwd_reduce(dout, x, ones, zeros, red, relu=0) -> insar_cam_bwd_coef ->
 *           insar_bnrelu_bwd_apply(dout, x, ones, zeros, zeros, ones, gate, coefB, zeros, zeros, dx, relu=0)
 *           -> insar_cam_scatter_max(dx, dmax, arg)
 * pool: per part (image, rows_per_part consecutive rows) and channel the sum, the maximum and the flat
 * index h*W+w of its first occurrence: psum/pmax/parg[B][P][C], P = ceil(H / rows_per_part). */
typedef struct InsarCam {
  int32_t B, H, W, C, Cr; int32_t rows;            /* rows = P of the pool slabs */
  const float* psum; const float* pmax; const int32_t* parg;
  const float* w1; const float* w2;
  float* avg; float* mx; int32_t* arg;             /* [B][C] saved for backward */
  float* ha; float* hm;                            /* [B][Cr] post-ReLU hidden activations of the two branches */
  float* gate;                                     /* [B][C] */
  float* coefB; float* dmax;                       /* backward: [B][C] mean-branch term /HW, max-branch term */
  float* ws;                                       /* backward scratch: float[B*(C + 2*Cr)] */
  float* dw1; float* dw2;
  int32_t accumulate; int32_t _pad;
} InsarCam;
int insar_cam_pool(const InsarAct* x, float* psum, float* pmax, int32_t* parg, int32_t rows_per_part,
                   void* stream);
int insar_cam_excite(const InsarCam* d, void* stream);
int insar_cam_bwd_coef(const InsarCam* d, const float* red /*[B][rows][2][C]*/, int32_t rows, void* stream);
int insar_cam_scatter_max(const InsarAct* dx, const float* dmax, const int32_t* arg, void* stream);

/* ---- MaxPool2d(2) (:106-109): y is the (H/2, W/2) grid rounded down (an odd last row / column belongs to no window) ---- */
int insar_maxpool2_fwd(const InsarAct* x, const InsarAct* y, void* stream);
/* dx (+)= route(dy) to the first maximum in scan order (torch semantics). */
int insar_maxpool2_bwd(const InsarAct* x, const InsarAct* dy, const InsarAct* dx, int32_t accumulate,
                       void* stream);

/* ---- F_T.resize(x, size, BILINEAR) of an NHWC slice (:138-139,144-145,150-151,156-157: the decoder's fallback for tile
 * sizes that are not multiples of 16), bilinear with align_corners = False, and its adjoint. src / dst: any two grids with
 * the same batch, channel slice width and dtype. */
int insar_resize_bilinear_fwd(const InsarAct* src, const InsarAct* dst, void* stream);
int insar_resize_bilinear_bwd(const InsarAct* ddst, const InsarAct* dsrc, void* stream);

/* ---- outc: Conv2d(64, num_classes, 1) (:125,162) ------------------------------------------------ */
int insar_conv1x1_out_fwd(const InsarAct* x, const float* w /*(K,C)*/, const float* bias,
                          float* logits /*NCHW fp32*/, int32_t K, void* stream);
/* dx slice <- dlogits * W ; part[insar_conv1x1_out_bwd_blocks(B,H)][K*C + K] partial (dW, dbias) sums. */
int insar_conv1x1_out_bwd_blocks(int32_t B, int32_t H);
int insar_conv1x1_out_bwd(const InsarAct* x, const float* w, const float* dlogits, int32_t K,
                          const InsarAct* dx, float* part, void* stream);

/* ---- loss entry: CrossEntropyLoss(ignore_index) (:465,344), fused forward + gradient ----------- */
/* ws: float[2 + 2*blocks] scratch (blocks = insar_ce_blocks(npix)); loss_out[0] = mean loss,
 * dlogits = (softmax - onehot)/n_valid on valid pixels, 0 elsewhere. */
int insar_ce_blocks(int64_t npix);
int insar_cross_entropy(const float* logits, const int64_t* target, int32_t B, int32_t K, int64_t HW,
                        int64_t ignore_index, float* dlogits, float* loss_out, float* ws, void* stream);
/* soft-Dice on softmax probabilities (build-side addition; the reference has no Dice). */
int insar_dice(const float* logits, const int64_t* target, int32_t B, int32_t K, int64_t HW,
               int64_t ignore_index, float smooth, float* dlogits, float* loss_out, float* ws,
               void* stream);

/* ce_weight*CE + dice_weight*Dice in one statistics pass + one gradient pass. ws: float[3 + 3K + blocks*(2 + 3K)];
 * loss_out[0] = combined, [1] = CE, [2] = Dice; dlogits = gradient of the combined loss. */
int insar_dice_ce(const float* logits, const int64_t* target, int32_t B, int32_t K, int64_t HW,
                  int64_t ignore_index, float smooth, float ce_weight, float dice_weight, float* dlogits,
                  float* loss_out, float* ws, void* stream);

/* ---- metrics (compute_metrics, :215-269): argmax (ties -> lower class) + TP/FP/FN counts -------- */
int insar_confusion(const float* logits, const int64_t* target, int32_t B, int32_t K, int64_t HW,
                    int64_t ignore_index, int64_t* counts /*[3][K], zeroed by the call*/, void* stream);

/* ---- optimizer: optim.Adam(lr=1e-4) (:466,346), multi-tensor ------------------------------------
 * table: int64[ntensors][5] = {param*, grad*, exp_avg*, exp_avg_sq*, numel}; chunks: int32[nchunks][2]
 * = {tensor index, chunk index}; each chunk covers `chunk_elems` elements. */
int insar_adam_step(const int64_t* table, const int32_t* chunks, int32_t nchunks, int32_t chunk_elems,
                    float lr, float beta1, float beta2, float eps, float bias_correction1,
                    float bias_correction2_sqrt, float grad_scale, void* stream);

/* ---- config 5: DeepLabV3-CA (DeepLabV3-ChannelAttention.py:83-162; backbone / ASPP arithmetic = torchvision's
 * resnet50(replace_stride_with_dilation=[False,True,True]) + DeepLabHead, restated: csrc/deeplab.hip) -----------------
 * Convolutions other than the stem run on insar_igemm (1x1 / 3x3, stride 1|2, dilation through the tap offsets with
 * INSAR_IGEMM_OOB_ZERO) and insar_wgrad with per-tap tables: */
int insar_pixel_table_taps(int32_t* tab /*[ntaps][Mpad]*/, int64_t Mpad, int32_t B, int32_t H, int32_t W, int32_t s,
                           int32_t Hb, int32_t Wb, int32_t ntaps, const int8_t* dy, const int8_t* dx, void* stream);
/* stem: Conv2d(1, 64, 7, stride 2, padding 3, bias=False) on the module input x (NCHW fp32 [B][1][H][W], :105-118).
 * stats: [insar_conv7x7s2_fwd_rows(B, H)][2][64] BatchNorm partial sums (nullable).
 * wgrad: part[insar_conv7x7s2_wgrad_blocks(B, H/2)][64*49] partial rows in torch (64,1,7,7) order. */
int insar_conv7x7s2_fwd_rows(int32_t B, int32_t H);
int insar_conv7x7s2_fwd(const float* x, int32_t H, int32_t W, const float* w, const InsarAct* y, float* stats, void* stream);
int insar_conv7x7s2_wgrad_blocks(int32_t B, int32_t Ho);
int insar_conv7x7s2_wgrad(const float* x, int32_t H, int32_t W, const InsarAct* dy, float* part, void* stream);
/* MaxPool2d(3, stride 2, padding 1) of the stem: arg[B][Ho][Wo][C] = window position ky*3+kx of the first maximum. */
int insar_maxpool3s2_fwd(const InsarAct* x, const InsarAct* y, uint8_t* arg, void* stream);
int insar_maxpool3s2_bwd(const InsarAct* dy, const uint8_t* arg, const InsarAct* dx, void* stream);
/* Bottleneck tail: dst = relu(y*scale + shift + res); and g = dout * (out > 0) (g may alias dout). */
int insar_bn_add_relu(const InsarAct* y, const float* scale, const float* shift, const InsarAct* res, const InsarAct* dst,
                      int32_t relu, void* stream);
int insar_relu_gate_bwd(const InsarAct* dout, const InsarAct* out, const InsarAct* g, void* stream);
/* ASPP pooling branch: out (B,1,1,C) = factor * sum_hw x;  dst (B,H,W,C) (+)= factor * src (B,1,1,C). */
int insar_sum_hw(const InsarAct* x, const InsarAct* out, float factor, void* stream);
int insar_broadcast_hw(const InsarAct* src, const InsarAct* dst, float factor, int32_t accumulate, void* stream);
/* the same, storing zero where `gate` (dst's grid, channels and dtype) is <= 0: the ReLU mask of the residual block whose incoming
 * gradient dst is, applied by its last writer (ABI 4; InsarIgemm.gate is the GEMM writers' form) */
int insar_broadcast_hw_gate(const InsarAct* src, const InsarAct* dst, const InsarAct* gate, float factor, int32_t accumulate, void* stream);
/* Dropout(p): make_mask != 0 draws mask[B][H][W][c_len] from (seed, element index) and stores it; == 0 applies `mask`. */
int insar_dropout(const InsarAct* x, const InsarAct* dst, uint8_t* mask, uint64_t seed, const int64_t* counter /*nullable, device:
                  mixed into the seed so that every replay of a captured step draws a new mask*/, float p, int32_t make_mask, void* stream);
/* F_T.resize(x, size, BILINEAR) (:160): bilinear, align_corners = False, on `planes` fp32 maps; and its adjoint. */
int insar_bilinear_fwd(const float* in, float* out, int32_t planes, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, void* stream);
int insar_bilinear_bwd(const float* dout, float* din, int32_t planes, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo, void* stream);

/* The same update with the step count and the bias corrections kept on the device: state = float[4] {t, 1-beta1^t,
 * sqrt(1-beta2^t), -}. The call first advances t by one (a one-thread launch), then updates the parameters; nothing
 * in the launch arguments changes from step to step, so a captured hipGraph of the training step can be replayed. */
int insar_adam_step_dev(const int64_t* table, const int32_t* chunks, int32_t nchunks, int32_t chunk_elems,
                        float lr, double beta1, double beta2, float eps, float* state, float grad_scale, void* stream);

/* ---- small helpers ---------------------------------------------------------------------------- */
int insar_scale_f32(float* p, int64_t n, float s, void* stream);
/* out[i] = x[i] * *scale, the factor in DEVICE memory: backward of the loss entry points (criterion(...).backward() at :345
 * hands d loss as a device scalar; no host read-back, no framework kernel). out and x 16-byte aligned. */
int insar_mul_dev_f32(float* out, const float* x, int64_t n, const float* scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* INSAR_HIP_H */
