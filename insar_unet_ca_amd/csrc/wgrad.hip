// Weight-gradient GEMM on the gfx950 matrix cores (split-K over pixels, no atomics).
//
//   part[split][tap][co][ci] = sum_{p in split} dY[pixB(p)+offB(tap), co] * X[pixA(p)+offA(tap), ci]
//
// replaces autograd's weight gradient of nn.Conv2d(3x3) (Unet-ChannalAttention.py:81,84) and of
// nn.ConvTranspose2d(k2,s2) (:112-121) inside loss.backward() (:345).
//
// Both operands are pixel-major ([pixel][channel], the contraction index is the slow one), so:
//  * K slabs of 64 pixels of X[., ci-tile] and dY[., co-tile] are gathered row by row into LDS by
//    LDS-DMA through int32 pixel-index tables (one table load per row per K step instead of an
//    integer division), double-buffered;
//  * bf16 fragments are read with the CDNA4 transposing LDS read ds_read_b64_tr_b16 (4 pixels x
//    16 channels -> per-lane 4 consecutive pixels of one channel), rows XOR-swizzled on the DMA
//    source so the transposed reads are bank-conflict-free; fp32 uses ds_read_b32 +
//    v_mfma_f32_16x16x4_f32;
//  * tiles: (Cin, Cout) = 64/128 x 64/128 with 4 waves and two work-groups per CU, or 256 x 256 with 8 waves
//    (wave tile 128 x 64) where both channel counts allow it: half the DMA pieces per MFMA (734 vs ~600 TF/s);
//  * each work-group owns one (split, tap, ci-tile, co-tile) and writes its fp32 partial tile with
//    16-byte stores; insar_wgrad_reduce folds the splits and re-lays the result out to the torch
//    parameter layout. Bitwise reproducible.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define WG_BKP 64  // pixels per K step

struct WgradArgs {
  const char* x; const char* dy; const int32_t* tabx; const int32_t* tabdy; float* part;
  long long ksteps;       // Mpad / 64
  int nsplit, ntaps, steps_per_split;
  int Cx, cx_off, Cin; int Cdy, cdy_off, Cout;
  int mtc, ntc;
  int offx[12], offdy[12];
  long long tabx_tap_stride;   // 0: one x table for every tap; else table of tap t starts at tabx + t*stride
};


// swizzle (in 16-byte chunks) applied to a tile row of RB bytes
template <int RB>
__device__ __forceinline__ int wg_swz(int row) {
  if constexpr (RB == 256) return (row & 7) << 1;
  else return ((row >> 1) & 3) << 1;
}

// NW waves = 2 (along TM) x NW/2 (along TN). Operand rows wider than 256 bytes are kept as SUB side-by-side
// sub-tiles of 256-byte rows ([sub][64 pixels][256 B]) so that one swizzle serves every width.
template <typename T, int TM, int TN, int NW>
struct WgradCfg {
  static constexpr int ES = sizeof(T);
  static constexpr int THREADS = NW * 64;
  static constexpr int SUBX = (TM * ES > 256) ? TM * ES / 256 : 1, SUBY = (TN * ES > 256) ? TN * ES / 256 : 1;
  static constexpr int RBX = TM * ES / SUBX, RBY = TN * ES / SUBY;   // sub-tile row bytes (128 or 256)
  static constexpr int X_STAGE = WG_BKP * TM * ES, Y_STAGE = WG_BKP * TN * ES;
  static constexpr int STAGE = X_STAGE + Y_STAGE;
  static constexpr int LDS_BYTES = 2 * STAGE + 128;          // + tap offsets
  static constexpr int CPRX = RBX / 16, CPRY = RBY / 16;
  static constexpr int NX = X_STAGE / 16 / THREADS;          // DMA chunks per thread per step
  static constexpr int NY = Y_STAGE / 16 / THREADS;
  static constexpr int MTW = TM / 32, NTW = TN / (NW / 2) / 16;   // 16x16 tiles per wave
};

template <typename T, int TM, int TN, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 4 ? 2 : 1)) void wgrad_kernel(WgradArgs a) {
  using Cfg = WgradCfg<T, TM, TN, NW>;
  constexpr int WG_THREADS = Cfg::THREADS;
  constexpr int ES = Cfg::ES;
  constexpr int MTW = Cfg::MTW, NTW = Cfg::NTW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* soff = (int*)(smem + 2 * Cfg::STAGE);   // [0..11] offx, [12..23] offdy

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tap = t % a.ntaps; t /= a.ntaps;
  const int ni = t % a.ntc; t /= a.ntc;
  const int mi = t % a.mtc; t /= a.mtc;
  const int split = t;

  if (tid == WG_THREADS - 1) {
#pragma unroll
    for (int i = 0; i < 12; ++i) { soff[i] = a.offx[i]; soff[12 + i] = a.offdy[i]; }
  }
  __syncthreads();
  const long long offx = soff[tap], offdy = soff[12 + tap];

  const long long ks0 = (long long)split * a.steps_per_split;
  long long ks1 = ks0 + a.steps_per_split;
  if (ks1 > a.ksteps) ks1 = a.ksteps;

  // staging geometry: chunk q = i*THREADS + tid -> sub-tile q / (64*CPR), row (q / CPR) % 64, position q % CPR
  const long long xcol = ((long long)a.cx_off + mi * TM) * ES;
  const long long ycol = ((long long)a.cdy_off + ni * TN) * ES;
  // per-thread constants of its DMA chunks (hoisted: they do not depend on the K step)
  int xrow_i[Cfg::NX], yrow_i[Cfg::NY];
  int xoff_i[Cfg::NX], yoff_i[Cfg::NY];                       // byte offset inside the pixel's channel run
#pragma unroll
  for (int i = 0; i < Cfg::NX; ++i) {
    const int q = i * WG_THREADS + tid;
    const int sub = q / (WG_BKP * Cfg::CPRX), row = (q / Cfg::CPRX) % WG_BKP, pos = q % Cfg::CPRX;
    xrow_i[i] = row;
    xoff_i[i] = sub * Cfg::RBX + (pos ^ wg_swz<Cfg::RBX>(row)) * 16;
  }
#pragma unroll
  for (int i = 0; i < Cfg::NY; ++i) {
    const int q = i * WG_THREADS + tid;
    const int sub = q / (WG_BKP * Cfg::CPRY), row = (q / Cfg::CPRY) % WG_BKP, pos = q % Cfg::CPRY;
    yrow_i[i] = row;
    yoff_i[i] = sub * Cfg::RBY + (pos ^ wg_swz<Cfg::RBY>(row)) * 16;
  }
  const char* xbase = a.x + offx * a.Cx * ES + xcol;
  const char* ybase = a.dy + offdy * a.Cdy * ES + ycol;
  const long long xpitch = (long long)a.Cx * ES, ypitch = (long long)a.Cdy * ES;

  int32_t px[Cfg::NX], py[Cfg::NY];
  const int32_t* tabx_tap = a.tabx + (long long)tap * a.tabx_tap_stride;
  auto load_tabs = [&](long long ks) {
#pragma unroll
    for (int i = 0; i < Cfg::NX; ++i) px[i] = tabx_tap[ks * WG_BKP + xrow_i[i]];
#pragma unroll
    for (int i = 0; i < Cfg::NY; ++i) py[i] = a.tabdy[ks * WG_BKP + yrow_i[i]];
  };
  const uint32_t lds0 = lds_offset_of(smem);
  auto stage = [&](int buf) {
    const uint32_t lx = lds0 + buf * Cfg::STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < Cfg::NX; ++i)
      lds_dma16_untracked(xbase + (long long)px[i] * xpitch + xoff_i[i], lx + i * (WG_THREADS * 16));
    const uint32_t ly = lx + Cfg::X_STAGE;
#pragma unroll
    for (int i = 0; i < Cfg::NY; ++i)
      lds_dma16_untracked(ybase + (long long)py[i] * ypitch + yoff_i[i], ly + i * (WG_THREADS * 16));
  };

  f32x4_t acc[MTW][NTW];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int wm = wave & 1, wn = wave >> 1;
  const int r16 = lane & 15, kq = lane >> 4;

  if (ks0 < ks1) {
    load_tabs(ks0);
    stage(0);
    if (ks0 + 1 < ks1) load_tabs(ks0 + 1);
    dma_drain_and_barrier();
    for (long long ks = ks0; ks < ks1; ++ks) {
      const int buf = (int)((ks - ks0) & 1);
      if (ks + 1 < ks1) {
        stage(buf ^ 1);                       // uses tables loaded one iteration ago
        if (ks + 2 < ks1) load_tabs(ks + 2);
      }
      const char* sX = smem + buf * Cfg::STAGE;
      const char* sY = sX + Cfg::X_STAGE;
      if constexpr (ES == 2) {
        // two k32 sub-steps; per sub-step each 16-lane group (kq) owns pixels 8kq..8kq+7, read as two
        // transposed 4x16 blocks. Ordering the two reads by kq parity keeps each 32-lane half on
        // 8 distinct (row & 7) values -> conflict-free with the source swizzle.
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8_t xf[MTW], yf[NTW];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int sel = h ^ (kq & 1);
            const int row = s * 32 + kq * 8 + sel * 4 + (r16 >> 2);
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
              const int colw = (wm * (TM / 2) + mt * 16 + (r16 & 3) * 4) * 2;      // byte column in the tile row
              int sub = 0, colb = colw;
              if constexpr (Cfg::SUBX > 1) { sub = colw / Cfg::RBX; colb = colw % Cfg::RBX; }
              const int pc = (colb >> 4) ^ wg_swz<Cfg::RBX>(row);
              const char* p = sX + sub * (WG_BKP * Cfg::RBX) + row * Cfg::RBX + pc * 16 + (colb & 15);
              s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
              xf[mt][4 * h + 0] = v[0]; xf[mt][4 * h + 1] = v[1]; xf[mt][4 * h + 2] = v[2]; xf[mt][4 * h + 3] = v[3];
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
              const int colw = (wn * (NTW * 16) + nt * 16 + (r16 & 3) * 4) * 2;
              int sub = 0, colb = colw;
              if constexpr (Cfg::SUBY > 1) { sub = colw / Cfg::RBY; colb = colw % Cfg::RBY; }
              const int pc = (colb >> 4) ^ wg_swz<Cfg::RBY>(row);
              const char* p = sY + sub * (WG_BKP * Cfg::RBY) + row * Cfg::RBY + pc * 16 + (colb & 15);
              s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
              yf[nt][4 * h + 0] = v[0]; yf[nt][4 * h + 1] = v[1]; yf[nt][4 * h + 2] = v[2]; yf[nt][4 * h + 3] = v[3];
            }
          }
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[mt], yf[nt], acc[mt][nt], 0, 0, 0);
        }
      } else {
#pragma unroll 4
        for (int jj = 0; jj < WG_BKP / 4; ++jj) {
          const int row = jj * 4 + kq;
          float xf[MTW], yf[NTW];
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt) {
            const int colw = (wm * (TM / 2) + mt * 16 + r16) * 4;
            const int sub = colw / Cfg::RBX, colb = colw % Cfg::RBX;
            const int pc = (colb >> 4) ^ wg_swz<Cfg::RBX>(row);
            xf[mt] = *(const float*)(sX + sub * (WG_BKP * Cfg::RBX) + row * Cfg::RBX + pc * 16 + (colb & 15));
          }
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            const int colw = (wn * (NTW * 16) + nt * 16 + r16) * 4;
            const int sub = colw / Cfg::RBY, colb = colw % Cfg::RBY;
            const int pc = (colb >> 4) ^ wg_swz<Cfg::RBY>(row);
            yf[nt] = *(const float*)(sY + sub * (WG_BKP * Cfg::RBY) + row * Cfg::RBY + pc * 16 + (colb & 15));
          }
#pragma unroll
          for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[mt], yf[nt], acc[mt][nt], 0, 0, 0);
        }
      }
      dma_drain_and_barrier();
    }
  }

  // C layout: row (ci) = kq*4 + reg, col (co) = r16  ->  16-byte stores into [co][ci]
  float* out = a.part + ((long long)split * a.ntaps + tap) * a.Cout * a.Cin;
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int ci = mi * TM + wm * (TM / 2) + mt * 16 + kq * 4;
      const int co = ni * TN + wn * (NTW * 16) + nt * 16 + r16;
      *(f32x4_t*)(out + (long long)co * a.Cin + ci) = acc[mt][nt];
    }
}

template <typename T, int TM, int TN, int NW>
static int launch_wgrad(WgradArgs& a, hipStream_t s) {
  using Cfg = WgradCfg<T, TM, TN, NW>;
  static std::atomic<uint64_t> attr_mask{0};     // per-device, see common.h
  {
    hipError_t e = insar_set_lds_once(attr_mask, (const void*)wgrad_kernel<T, TM, TN, NW>, Cfg::LDS_BYTES);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  a.mtc = a.Cin / TM; a.ntc = a.Cout / TN;
  const long long grid = (long long)a.nsplit * a.ntaps * a.mtc * a.ntc;
  if (grid > 0x7fffffffLL) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad: grid too large");
  hipLaunchKernelGGL((wgrad_kernel<T, TM, TN, NW>), dim3((unsigned)grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, s, a);
  INSAR_CHECK_LAUNCH("insar_wgrad");
  return INSAR_OK;
}

// Tile extent along a channel dimension of C channels (the Cin x Cout tile of a launch is tile(Cin) x tile(Cout),
// except that 256 is used only when BOTH dimensions allow it: 8 waves, 128 KB of LDS, half the DMA pieces per MFMA).
extern "C" int insar_wgrad_tile(int32_t C, int32_t dtype) {
  if (dtype != INSAR_BF16) return (C % 128) == 0 ? 128 : 64;      // fp32: 128 x 128 (8 waves) only when BOTH allow it
  const int cap = insar_knob(KNOB_WGRAD_TILE_MAX);
  if (cap > 0 && cap < 256) return (C % 128) == 0 ? 128 : 64;
  return (C % 256) == 0 ? 256 : ((C % 128) == 0 ? 128 : 64);
}

// The (Cin, Cout) tile insar_wgrad uses for a layer: bf16 256x256 (8 waves), 128x128, 128x64, 64x128, 64x64 (4 waves);
// fp32 128x128 (8 waves) or 64x64.
static void wgrad_tile_pair(int Cin, int Cout, int dtype, int& tm, int& tn) {
  tm = insar_wgrad_tile(Cin, dtype); tn = insar_wgrad_tile(Cout, dtype);
  if (dtype != INSAR_BF16) { if (!(tm == 128 && tn == 128)) tm = tn = 64; return; }
  if (tm == 256 && tn == 256) return;
  // (256 x 128 / 128 x 256 tiles with 8 waves were measured: no gain over 128 x 128 in the step, same-box A/B)
  tm = tm > 128 ? 128 : tm; tn = tn > 128 ? 128 : tn;
}
extern "C" int insar_wgrad_tile_pair(int32_t Cin, int32_t Cout, int32_t dtype) {      // (tile(Cin) << 16) | tile(Cout)
  int tm, tn;
  wgrad_tile_pair(Cin, Cout, dtype, tm, tn);
  return (tm << 16) | tn;
}

extern "C" int insar_wgrad(const InsarWgrad* d, void* stream) {
  if (!d || !d->x.ptr || !d->dy.ptr || !d->tabx || !d->tabdy || !d->part) INSAR_FAIL(INSAR_E_ARG, "insar_wgrad: null pointer");
  int rc;
  if ((rc = insar_check_act(&d->x, "insar_wgrad", "x"))) return rc;
  if ((rc = insar_check_act(&d->dy, "insar_wgrad", "dy"))) return rc;
  if (d->x.dtype != d->dy.dtype) INSAR_FAIL(INSAR_E_DTYPE, "insar_wgrad: x/dy dtype differ");
  if (d->Mpad < WG_BKP || d->Mpad % WG_BKP) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad: Mpad must be a positive multiple of 64");
  if (d->nsplit < 1 || d->ntaps < 1 || d->ntaps > 12) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad: nsplit/ntaps");
  const int Cin = d->x.c_len, Cout = d->dy.c_len;
  if (Cin % 64 || Cout % 64) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad: Cin=%d/Cout=%d must be multiples of 64", Cin, Cout);
  WgradArgs a;
  a.x = (const char*)d->x.ptr; a.dy = (const char*)d->dy.ptr; a.tabx = d->tabx; a.tabdy = d->tabdy; a.part = d->part;
  a.ksteps = d->Mpad / WG_BKP;
  a.nsplit = d->nsplit; a.ntaps = d->ntaps;
  a.steps_per_split = (int)((a.ksteps + d->nsplit - 1) / d->nsplit);
  a.Cx = d->x.C; a.cx_off = d->x.c_off; a.Cin = Cin;
  a.Cdy = d->dy.C; a.cdy_off = d->dy.c_off; a.Cout = Cout;
  for (int t = 0; t < 12; ++t) { a.offx[t] = t < d->ntaps ? d->offx[t] : 0; a.offdy[t] = t < d->ntaps ? d->offdy[t] : 0; }
  a.tabx_tap_stride = d->tabx_tap_stride;
  hipStream_t s = (hipStream_t)stream;
  int tm, tn;
  wgrad_tile_pair(Cin, Cout, d->x.dtype, tm, tn);
  if (d->x.dtype == INSAR_BF16) {
    if (tm == 256 && tn == 256) return launch_wgrad<bf16_t, 256, 256, 8>(a, s);
    if (tm == 128 && tn == 128) return launch_wgrad<bf16_t, 128, 128, 4>(a, s);
    if (tm == 128) return launch_wgrad<bf16_t, 128, 64, 4>(a, s);
    if (tn == 128) return launch_wgrad<bf16_t, 64, 128, 4>(a, s);
    return launch_wgrad<bf16_t, 64, 64, 4>(a, s);
  }
  // fp32 (v_mfma_f32_16x16x4_f32): 128 x 128 tiles with 8 waves (wave tile 64 x 32: 6 LDS reads per 8 MFMAs
  // instead of 4 per 4, half the DMA pieces per MFMA) where both channel counts are multiples of 128
  if (tm == 128 && tn == 128) return launch_wgrad<float, 128, 128, 8>(a, s);
  return launch_wgrad<float, 64, 64, 4>(a, s);
}

// ---------------------------------------------------------------------------------------------
// Fold the split-K partial slabs and re-lay [tap][co][ci] out to the torch parameter layout.
// One block owns (co, 128 consecutive ci): slab reads are coalesced along ci, the tile is
// transposed through LDS ([ci][tap], odd pitch), and the parameter-layout writes are
// contiguous (Conv2d: 128*ntaps consecutive floats).
// ---------------------------------------------------------------------------------------------
// part_out[g][e] = sum_{sp in group g} part_in[sp][e]  (group = `group` consecutive splits): first
// stage of the deterministic split-K fold when there are many splits (elementwise, float4, HBM-bound).
__global__ void wgrad_fold_kernel(const float4* __restrict__ in, float4* __restrict__ out, long long slab4, int nsplit,
                                  int group) {
  const int g = blockIdx.y;
  const int s0 = g * group;
  int s1 = s0 + group; if (s1 > nsplit) s1 = nsplit;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < slab4; e += (long long)gridDim.x * blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (int sp = s0; sp < s1; ++sp) {
      const float4 v = in[(long long)sp * slab4 + e];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    out[(long long)g * slab4 + e] = acc;
  }
}

extern "C" int insar_wgrad_fold(const float* part_in, float* part_out, int64_t slab_floats, int32_t nsplit, int32_t group,
                                void* stream) {
  if (!part_in || !part_out) INSAR_FAIL(INSAR_E_ARG, "insar_wgrad_fold: null pointer");
  if (slab_floats < 4 || (slab_floats & 3) || nsplit < 1 || group < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_fold: bad shape");
  const long long slab4 = slab_floats / 4;
  const int groups = (nsplit + group - 1) / group;
  dim3 grid(insar_grid_cap((slab4 + 255) / 256, 1024), groups);
  hipLaunchKernelGGL(wgrad_fold_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float4*)part_in, (float4*)part_out,
                     slab4, nsplit, group);
  INSAR_CHECK_LAUNCH("insar_wgrad_fold");
  return INSAR_OK;
}

#define WR_CIT 128
// VEC: 16-byte slab reads (Ci % 4 == 0, 16-byte aligned slabs): a thread owns four consecutive ci of one tap and keeps eight
// slabs in flight (fixed-order tree: deterministic). The split-K slabs are the second largest traffic of the weight-gradient
// stream (0.8 GB per step of config 2): with 4-byte reads and four in flight the launches ran at 2.6 TB/s alone.
template <bool VEC>
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ grad, int nsplit, int ntaps,
                                    int Co, int Ci, int layout, int accumulate) {
  __shared__ float t[WR_CIT * 12];
  const int co = blockIdx.x, ci0 = blockIdx.y * WR_CIT;
  const int nci = (Ci - ci0) < WR_CIT ? (Ci - ci0) : WR_CIT;
  const long long total = (long long)Co * Ci;
  const long long slab = total * ntaps;
  if constexpr (VEC) {
    const int nc4 = nci >> 2;
    for (int idx = threadIdx.x; idx < ntaps * nc4; idx += blockDim.x) {
      const int tap = idx / nc4, c = (idx - tap * nc4) << 2;
      const float* p = part + (long long)tap * total + (long long)co * Ci + ci0 + c;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      int sp = 0;
      for (; sp + 8 <= nsplit; sp += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(p + (long long)(sp + u) * slab);
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u].x += v[u + 4].x; v[u].y += v[u + 4].y; v[u].z += v[u + 4].z; v[u].w += v[u + 4].w; }
        s.x += (v[0].x + v[1].x) + (v[2].x + v[3].x); s.y += (v[0].y + v[1].y) + (v[2].y + v[3].y);
        s.z += (v[0].z + v[1].z) + (v[2].z + v[3].z); s.w += (v[0].w + v[1].w) + (v[2].w + v[3].w);
      }
      for (; sp < nsplit; ++sp) {
        const float4 v = *(const float4*)(p + (long long)sp * slab);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      t[(c + 0) * ntaps + tap] = s.x; t[(c + 1) * ntaps + tap] = s.y;
      t[(c + 2) * ntaps + tap] = s.z; t[(c + 3) * ntaps + tap] = s.w;
    }
  } else {
  for (int idx = threadIdx.x; idx < ntaps * nci; idx += blockDim.x) {
    const int tap = idx / nci, c = idx - tap * nci;
    const float* p = part + (long long)tap * total + (long long)co * Ci + ci0 + c;
    // four interleaved partial sums (fixed order: deterministic), so that four slab reads are in flight per thread
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int sp = 0;
    for (; sp + 4 <= nsplit; sp += 4) {
      s0 += p[(long long)sp * slab]; s1 += p[(long long)(sp + 1) * slab];
      s2 += p[(long long)(sp + 2) * slab]; s3 += p[(long long)(sp + 3) * slab];
    }
    for (; sp < nsplit; ++sp) s0 += p[(long long)sp * slab];
    t[c * ntaps + tap] = (s0 + s1) + (s2 + s3);
  }
  }
  __syncthreads();
  if (layout == 0) {
    float* g = grad + ((long long)co * Ci + ci0) * ntaps;
    for (int j = threadIdx.x; j < nci * ntaps; j += blockDim.x) g[j] = accumulate ? g[j] + t[j] : t[j];
  } else {
    for (int j = threadIdx.x; j < nci * ntaps; j += blockDim.x) {
      const int c = j / ntaps, tap = j - c * ntaps;
      float* g = grad + ((long long)(ci0 + c) * Co + co) * ntaps + tap;
      *g = accumulate ? *g + t[j] : t[j];
    }
  }
}

extern "C" int insar_wgrad_reduce(const float* part, float* grad, int32_t nsplit, int32_t ntaps, int32_t Co, int32_t Ci,
                                  int32_t layout, int32_t accumulate, void* stream) {
  if (!part || !grad) INSAR_FAIL(INSAR_E_ARG, "insar_wgrad_reduce: null pointer");
  if (layout != 0 && layout != 1) INSAR_FAIL(INSAR_E_ARG, "insar_wgrad_reduce: layout");
  if (ntaps < 1 || ntaps > 12 || Co < 1 || Ci < 1 || nsplit < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_reduce: bad shape");
  dim3 grid(Co, (Ci + WR_CIT - 1) / WR_CIT);
  if ((Ci & 3) == 0 && insar_aligned16(part))
    hipLaunchKernelGGL(wgrad_reduce_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, part, grad, nsplit, ntaps, Co, Ci,
                       layout, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, part, grad, nsplit, ntaps, Co, Ci,
                       layout, accumulate);
  INSAR_CHECK_LAUNCH("insar_wgrad_reduce");
  return INSAR_OK;
}
