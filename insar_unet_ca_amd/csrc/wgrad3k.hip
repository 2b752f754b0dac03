// Weight gradient of a 3x3 / stride-1 convolution (Unet-ChannalAttention.py:81,84 inside loss.backward(), :345) for the
// layers with 64 or 128 channels on BOTH sides (the 256^2 and 128^2 levels of the U-Net): the row-of-taps decomposition
//
//   part[slab][3*ty + tx][co][ci] = sum_{p in slab's pixels} dY[p, co] * X[p + (ty-1)*(W+2) + (tx-1), ci],   tx = 0, 1, 2
//
// with the structure of wgrad3x.hip — 64 (ci) x 64 (co) x three-tap wave tiles (192 accumulator registers, 16 MFMAs per
// 8 + 8 transposing fragment reads), phases of [fragment reads + LDS-DMA issue | 16 MFMAs] with the two waves of a SIMD
// in opposite roles, fragment addresses that are one lane register + an immediate, a ring of LDS slots filled two
// steps ahead behind a counted vmcnt — for tiles too small to give eight waves a 64 x 64 tile each: the eight waves of a
// work-group split the PIXELS instead. A (TM x TN) tile has WT = TM/64 * TN/64 wave tiles; the work-group stages a K step
// of KS * 32 consecutive pixels of one image row (KS = 8 / WT) and wave w takes the 32-pixel slice w / WT of it, i.e. a
// work-group is KS split-K slices that share one staging pipeline (and their slices' halo pixels), and writes KS slabs.
// The 128 x 128 row-of-taps kernel of wgrad3.hip gives its waves 64 x 32 tiles (8 MFMAs per 8 + 4 reads), the 64 x 64 one
// 32 x 32 (4 per 4 + 4): LDS-read bound, and in the step they cost as much side-queue time as the ten deep layers.
//
// LDS image: 64-byte plane rows = the two adjacent 16-channel blocks (2j, 2j + 1) of a pixel; block b sits in slot
// (b & 1) ^ ((row >> 2) & 1) of row `row` of plane b >> 1 (four rows per 256-byte bank line: the eight pixel rows a
// 32-lane half of a ds_read_b64_tr_b16 touches land in eight different 32-byte bank segments whatever the first row is).
// A wave whose operand side is 128 channels wide reads ONE block per plane (block parity = its position), one whose side
// is 64 wide reads both blocks of its two planes: a lane base register per (k half h, tap tx, block parity).
// A K step lies inside one image row (W % (KS*32) == 0, checked by insar_wgrad_conv3k_tile): X row r of a step is padded
// pixel p0 - 1 + r, dY row k is p0 + k, and tap tx of dY row k reads X row k + tx.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct Wgrad3kArgs {
  const char* x; const char* dy; float* part;
  long long ksteps;             // B*H*W / PX
  int nsplit, steps_per_split;
  int H, W, Wp, spr;            // spr: K steps per image row
  int Cx, cx_off, Cin; int Cdy, cdy_off, Cout;
  int mtc, ntc;
};

__device__ __forceinline__ void wk_dma(const char* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ int wk_g(int row) { return (row >> 2) & 1; }
template <int N> __device__ __forceinline__ void wk_wait_vm_lgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

template <int TM, int TN>
struct Wgrad3kCfg {
  static constexpr int THREADS = 512, NW = 8;
  static constexpr int WM = TM / 64, WN = TN / 64, WT = WM * WN;     // wave tiles along ci / co / per pixel slice
  static constexpr int KS = NW / WT;                                  // pixel slices (= slabs) per work-group
  static constexpr int PX = KS * 32;                                  // pixels per K step
  static constexpr int XR = PX + 16;                                  // staged X rows (PX + 2 needed; whole 16-row DMA pieces)
  static constexpr int NPLX = TM / 32, NPLY = TN / 32;                // 64-byte-row planes
  static constexpr int XPL = XR * 64, YPL = PX * 64;                  // bytes per plane
  static constexpr int X_STAGE = NPLX * XPL, Y_STAGE = NPLY * YPL;
  static constexpr int STAGE = X_STAGE + Y_STAGE;
  static constexpr int NSLOT = 3 * STAGE <= 160 * 1024 ? 3 : 2;
  static constexpr int LDS_BYTES = NSLOT * STAGE;
  static constexpr int XP = X_STAGE / 1024, YP = Y_STAGE / 1024, NP = XP + YP;   // 1-KB pieces per step
  static constexpr int NPW = (NP + NW - 1) / NW;                      // DMA instructions per wave per step (waves >= TAIL: one less)
  static constexpr int TAIL = NP % NW;                                // waves that issue NPW pieces (0: all)
  static constexpr int NLX = WM == 1 ? 2 : 1, NLY = WN == 1 ? 2 : 1;  // block parities a wave reads per plane row
  static_assert(WT == 1 || WT == 2 || WT == 4, "64 / 128 channels a side");
  static_assert(X_STAGE % 1024 == 0 && Y_STAGE % 1024 == 0, "whole DMA pieces");
  static_assert(LDS_BYTES <= 160 * 1024, "ring must fit the CU's LDS");
};

template <int TM, int TN>
__global__ __launch_bounds__(512) void wgrad3k_kernel(Wgrad3kArgs a) {
  using Cfg = Wgrad3kCfg<TM, TN>;
  constexpr int WM = Cfg::WM, WT = Cfg::WT, KS = Cfg::KS, PX = Cfg::PX, XR = Cfg::XR;
  constexpr int NPW = Cfg::NPW, XP = Cfg::XP, NSLOT = Cfg::NSLOT, NLX = Cfg::NLX, NLY = Cfg::NLY;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                                          // 0: waves 0-3, 1: waves 4-7 (SIMD partners)
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int ty = t % 3; t /= 3;
  const int ni = t % a.ntc; t /= a.ntc;
  const int mi = t % a.mtc; t /= a.mtc;
  const int split = t;

  const int ks0 = split * a.steps_per_split;
  int ks1 = ks0 + a.steps_per_split;
  if (ks1 > (int)a.ksteps) ks1 = (int)a.ksteps;
  const int nsteps = ks1 > ks0 ? ks1 - ks0 : 0;

  // per-lane byte offsets of its DMA chunks relative to the step's scalar base. Piece q = i*8 + wave (q < XP: X, else dY)
  // holds the 16-byte chunks q*64 + lane of the stage: plane, row, slot -> the block stored there and its half. X rows
  // beyond the PX + 2 a step needs are fetched from the last needed row (whole pieces, constant counts, nothing outside
  // the buffer); they are never read.
  const long long xpitch = (long long)a.Cx * 2, ypitch = (long long)a.Cdy * 2;
  uint32_t off_i[NPW];
  bool isx_i[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = i * 8 + wave;
    if (q < XP) {
      const int c = q * 64 + lane;
      const int plane = c / (XR * 4), row = (c >> 2) % XR, cc = c & 3;
      const int blk = plane * 2 + ((cc >> 1) ^ wk_g(row));
      const int srow = row < PX + 2 ? row : PX + 1;
      off_i[i] = (uint32_t)(srow * xpitch) + blk * 32 + (cc & 1) * 16;
      isx_i[i] = true;
    } else {
      const int c = (q - XP) * 64 + lane;
      const int plane = c / (PX * 4), row = (c >> 2) % PX, cc = c & 3;
      const int blk = plane * 2 + ((cc >> 1) ^ wk_g(row));
      off_i[i] = (uint32_t)(row * ypitch) + blk * 32 + (cc & 1) * 16;
      isx_i[i] = false;
    }
  }
  const char* xbase = a.x + ((long long)(ty - 1) * a.Wp - 1) * xpitch + ((long long)a.cx_off + mi * TM) * 2;
  const char* ybase = a.dy + ((long long)a.cdy_off + ni * TN) * 2;

  // padded index of the first pixel of a K step, advanced step by step (wave-uniform)
  int seg, hrow, img;
  {
    const int g = ks0 / a.spr;
    seg = ks0 - g * a.spr;
    img = g / a.H;
    hrow = g - img * a.H;
  }
  auto next_pixel = [&]() -> long long {
    const long long p = ((long long)img * (a.H + 2) + hrow + 1) * a.Wp + seg * PX + 1;
    if (++seg == a.spr) {
      seg = 0;
      if (++hrow >= a.H) { hrow = 0; ++img; }
    }
    return p;
  };
  const uint32_t ldsb = lds_offset_of(smem);
  const uint32_t lds0 = ldsb + wave * 1024;
  const bool full = Cfg::TAIL == 0 || wave < Cfg::TAIL;               // this wave issues NPW pieces a step, else NPW - 1
  auto piece = [&](int slot, const char* sx, const char* sy, int j) {
    if (j == NPW - 1 && !full) return;
    // X pieces fill [0, X_STAGE) of the slot, dY pieces the rest: piece q lands at q * 1 KB either way
    wk_dma(isx_i[j] ? sx : sy, off_i[j], lds0 + slot * Cfg::STAGE + j * 8192);
  };

  f32x4_t acc[3][4][4];
#pragma unroll
  for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t3][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int ksl = wave / WT, pos = wave % WT;                         // pixel slice, wave tile inside it
  const int wm = pos % WM, wn = pos / WM;
  const int r16 = lane & 15, kq = lane >> 4;

  if (nsteps > 0) {
    {
      const long long p0 = next_pixel();
      const char* sx = xbase + p0 * xpitch; const char* sy = ybase + p0 * ypitch;
#pragma unroll
      for (int j = 0; j < NPW; ++j) piece(0, sx, sy, j);
    }
    if (NSLOT == 3 && nsteps > 1) {
      const long long p1 = next_pixel();
      const char* sx = xbase + p1 * xpitch; const char* sy = ybase + p1 * ypitch;
#pragma unroll
      for (int j = 0; j < NPW; ++j) piece(1, sx, sy, j);
      if (full) wk_wait_vm_lgkm<NPW>(); else wk_wait_vm_lgkm<NPW - 1>();       // step 0 landed, step 1 in flight
    } else {
      wk_wait_vm_lgkm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();                       // group 1 runs one barrier behind

    // lane bases of the fragment reads (slot 0): row base + slot of the block + 8-byte column; block parity lo is the
    // wave's position where its side is 128 wide (one block per plane), both parities where it is 64 wide
    uint32_t by[2][NLY], bx[2][3][NLX];
    {
      const int p8 = (r16 & 3) * 8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int lr = ksl * 32 + kq * 8 + ((h ^ (kq & 1)) << 2) + (r16 >> 2);
#pragma unroll
        for (int l = 0; l < NLY; ++l) {
          const int lo = NLY == 1 ? wn : l;
          by[h][l] = ldsb + Cfg::X_STAGE + lr * 64 + ((lo ^ wk_g(lr)) << 5) + p8;
        }
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
          const int r = lr + t3;
#pragma unroll
          for (int l = 0; l < NLX; ++l) {
            const int lo = NLX == 1 ? wm : l;
            bx[h][t3][l] = ldsb + r * 64 + ((lo ^ wk_g(r)) << 5) + p8;
          }
        }
      }
    }
    typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_p;
    int slot = 0;
    for (int k = 0; k < nsteps; ++k) {
      const bool more = k + (NSLOT - 1) < nsteps;                      // a step to stage during this one
      int slotn = slot + (NSLOT - 1); if (slotn >= NSLOT) slotn -= NSLOT;
      const char* nsx = xbase; const char* nsy = ybase;
      if (more) {
        const long long pn = next_pixel();
        nsx = xbase + pn * xpitch; nsy = ybase + pn * ypitch;
      }
      bf16x8_t yf[4], xf[4];
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3) {
        // ---- load part ----
        if (t3 == 0) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
              // tile nt of the wave is block nt*WN + wn: plane and parity
              const int pl = NLY == 1 ? nt : nt >> 1, l = NLY == 1 ? 0 : nt & 1;
              s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(uintptr_t)(by[h][l] + pl * Cfg::YPL));
              yf[nt][4 * h + 0] = v[0]; yf[nt][4 * h + 1] = v[1]; yf[nt][4 * h + 2] = v[2]; yf[nt][4 * h + 3] = v[3];
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int pl = NLX == 1 ? mt : mt >> 1, l = NLX == 1 ? 0 : mt & 1;
            s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(uintptr_t)(bx[h][t3][l] + pl * Cfg::XPL));
            xf[mt][4 * h + 0] = v[0]; xf[mt][4 * h + 1] = v[1]; xf[mt][4 * h + 2] = v[2]; xf[mt][4 * h + 3] = v[3];
          }
        // DMA pieces of the step NSLOT - 1 ahead, spread over the three phases (fewest beside the 16-read phase)
        if (more) {
          constexpr int N0 = NPW / 4, N1 = (NPW - N0 + 1) / 2;        // pieces in phase 0 / 1; the rest in phase 2
          if (t3 == 0) {
#pragma unroll
            for (int j = 0; j < N0; ++j) piece(slotn, nsx, nsy, j);
          } else if (t3 == 1) {
#pragma unroll
            for (int j = N0; j < N0 + N1; ++j) piece(slotn, nsx, nsy, j);
          } else {
#pragma unroll
            for (int j = N0 + N1; j < NPW; ++j) piece(slotn, nsx, nsy, j);
          }
        }
        if (t3 == 2) {
          // the vector-memory wait of the step. Three slots: everything but the pieces issued during THIS step (they are
          // for the step after next) has landed; two slots: the next step's pieces were issued during this step.
          if (NSLOT == 2 || !more) wk_wait_vm_lgkm<0>();
          else if (full) wk_wait_vm_lgkm<NPW>();
          else wk_wait_vm_lgkm<NPW - 1>();
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        // ---- compute part ----
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            acc[t3][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[mt], yf[nt], acc[t3][mt][nt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
      // next ring slot: the lane bases move with it, in place
      const int delta = slot == NSLOT - 1 ? -(NSLOT - 1) * Cfg::STAGE : Cfg::STAGE;
      if (++slot == NSLOT) slot = 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int l = 0; l < NLY; ++l) by[h][l] += delta;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
          for (int l = 0; l < NLX; ++l) bx[h][t3][l] += delta;
      }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                       // group 0 meets group 1's last barrier
  }

  // slab of this wave's pixel slice: C layout of a 16x16 accumulator: row (ci) = kq*4 + reg, col (co) = r16 -> 16-byte
  // stores into [co][ci]; tile mt of the wave is channel block mt*WM + wm, tile nt block nt*WN + wn
#pragma unroll
  for (int t3 = 0; t3 < 3; ++t3) {
    float* out = a.part + (((long long)split * KS + ksl) * 9 + ty * 3 + t3) * a.Cout * a.Cin;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int ci = mi * TM + (mt * WM + wm) * 16 + kq * 4;
        const int co = ni * TN + (nt * Cfg::WN + wn) * 16 + r16;
        *(f32x4_t*)(out + (long long)co * a.Cin + ci) = acc[t3][mt][nt];
      }
  }
}

template <int TM, int TN>
static int launch_wgrad3k(Wgrad3kArgs& a, hipStream_t s) {
  using Cfg = Wgrad3kCfg<TM, TN>;
  static std::atomic<uint64_t> attr_mask{0};     // per-device, see common.h
  {
    hipError_t e = insar_set_lds_once(attr_mask, (const void*)wgrad3k_kernel<TM, TN>, Cfg::LDS_BYTES);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_wgrad_conv3k: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  a.mtc = a.Cin / TM; a.ntc = a.Cout / TN;
  const long long grid = (long long)a.nsplit * 3 * a.mtc * a.ntc;
  if (grid > 0x7fffffffLL) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3k: grid too large");
  hipLaunchKernelGGL((wgrad3k_kernel<TM, TN>), dim3((unsigned)grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, s, a);
  INSAR_CHECK_LAUNCH("insar_wgrad_conv3k");
  return INSAR_OK;
}

static int wgrad3k_pick(const InsarAct* x, int32_t Cout, int* tm, int* tn, int* ks) {
  if (!x || x->dtype != INSAR_BF16) return 0;
  const int cin = x->c_len;
  // 64 or 128 channels a side: wider sides belong to wgrad3x.hip (256 x 128 tiles)
  const int m = (cin % 128 == 0) ? 128 : (cin % 64 == 0 ? 64 : 0);
  const int n = (Cout % 128 == 0) ? 128 : (Cout % 64 == 0 ? 64 : 0);
  if (!m || !n) return 0;
  const int k = 8 / ((m / 64) * (n / 64));
  if (x->W % (k * 32)) return 0;                   // a K step of k*32 pixels lies inside one image row
  *tm = m; *tn = n; *ks = k;
  return 1;
}

// (tile(Cin) << 16) | tile(Cout) of this kernel for the layer, or 0 where it does not apply; insar_wgrad_conv3k_slices:
// the number of slabs a work-group writes (pixel slices KS = 8 / wave tiles): `part` holds nsplit * KS slabs.
extern "C" int insar_wgrad_conv3k_tile(const InsarAct* x, int32_t Cout) {
  int tm, tn, ks;
  return wgrad3k_pick(x, Cout, &tm, &tn, &ks) ? (tm << 16) | tn : 0;
}
extern "C" int insar_wgrad_conv3k_slices(const InsarAct* x, int32_t Cout) {
  int tm, tn, ks;
  return wgrad3k_pick(x, Cout, &tm, &tn, &ks) ? ks : 0;
}

// part[nsplit * KS][tap][co][ci] (tap = 3*ty + tx; fold with insar_wgrad_reduce over nsplit * KS slabs)
extern "C" int insar_wgrad_conv3k(const InsarAct* x, const InsarAct* dy, float* part, int32_t nsplit, void* stream) {
  if (!x || !dy || !part) INSAR_FAIL(INSAR_E_ARG, "insar_wgrad_conv3k: null pointer");
  int rc;
  if ((rc = insar_check_act(x, "insar_wgrad_conv3k", "x"))) return rc;
  if ((rc = insar_check_act(dy, "insar_wgrad_conv3k", "dy"))) return rc;
  if (x->B != dy->B || x->H != dy->H || x->W != dy->W) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3k: x/dy grids differ");
  if (x->dtype != dy->dtype) INSAR_FAIL(INSAR_E_DTYPE, "insar_wgrad_conv3k: x/dy dtype differ");
  int tm, tn, ks;
  if (!wgrad3k_pick(x, dy->c_len, &tm, &tn, &ks))
    INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3k: unsupported layer (bf16, channel counts multiples of 64, W a multiple of the K step); use insar_wgrad_conv3");
  if (nsplit < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3k: nsplit");
  const int px = ks * 32;
  if ((long long)(px + 16) * x->C * 2 >= 0x7fffffffLL || (long long)px * dy->C * 2 >= 0x7fffffffLL)
    INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3k: channel pitch too large");
  Wgrad3kArgs a;
  a.x = (const char*)x->ptr; a.dy = (const char*)dy->ptr; a.part = part;
  a.ksteps = (long long)x->B * x->H * x->W / px;
  a.nsplit = nsplit;
  a.steps_per_split = (int)((a.ksteps + nsplit - 1) / nsplit);
  a.H = x->H; a.W = x->W; a.Wp = x->W + 2; a.spr = x->W / px;
  a.Cx = x->C; a.cx_off = x->c_off; a.Cin = x->c_len;
  a.Cdy = dy->C; a.cdy_off = dy->c_off; a.Cout = dy->c_len;
  hipStream_t s = (hipStream_t)stream;
  if (tm == 64 && tn == 64) return launch_wgrad3k<64, 64>(a, s);
  if (tm == 128 && tn == 64) return launch_wgrad3k<128, 64>(a, s);
  if (tm == 64 && tn == 128) return launch_wgrad3k<64, 128>(a, s);
  return launch_wgrad3k<128, 128>(a, s);
}
