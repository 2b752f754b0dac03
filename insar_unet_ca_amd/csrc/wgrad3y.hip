// Weight gradient of a 3x3 / stride-1 convolution (Unet-ChannalAttention.py:81,84 inside loss.backward(), :345): wgrad3x.hip's
// row-of-taps decomposition
//
//   part[split][3*ty + tx][co][ci] = sum_p dY[p, co] * X[p + (ty-1)*(W+2) + (tx-1), ci],   tx = 0, 1, 2
//
// built for TWO CO-RESIDENT WORK-GROUPS PER CU, as conv3x3_flat2.hip is for the forward / input-gradient convolutions:
//   * a work-group is 4 waves (one per SIMD) on a 128 (ci) x 128 (co) tile, each wave the same 64 x 64 x three-tap tile as in
//     wgrad3x.hip (192 accumulator registers, same plane-layout LDS image: every fragment address a lane register plus an
//     immediate, same ds_read_b64_tr_b16 reads);
//   * LDS: a TWO-slot ring of 34 KB stages (X 72 pixel rows x 128 ci + dY 64 x 128 co) = 68 KB — two work-groups per CU, each
//     wave with the 256-register budget of two waves per SIMD;
//   * a K step (64 pixels) = [LDS-DMA of step k + 1 (9 one-KB pieces per wave, spread over the step's first phases) | six
//     phases of 8 (+ 8) fragment reads and 16 MFMAs, NO barrier between them] -> s_waitcnt vmcnt(0) (the pieces are a whole
//     step old) -> ONE s_barrier. The two work-groups of a CU are not coupled at all: the MFMA pipe alternates between the two
//     waves of a SIMD by itself, and one group's slab stores / first-step wait run under the other's K loop.
// wgrad3x.hip couples its eight waves through twelve barriers per K step (two wave groups in opposite roles) and reaches
// 59 - 63 % MFMA-busy; the decoupled form is what took the flat kernel's K loop from ~57 % to ~70 %. It stages 89 bytes per MFMA
// through the L2 -> LDS path (wgrad3x: 68), still under the ~88 B/MFMA that path sustains at the MFMA peak.
// Same products in the same order per accumulator as wgrad3.hip / wgrad3x.hip: the slabs are bit for bit theirs at the same
// split.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short wy_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short wy_s16x4_t;
typedef __attribute__((ext_vector_type(4))) float wy_f32x4_t;

#define WY_BKP 64       // pixels per K step
#define WY_XR 72        // X rows staged per step: padded pixel p0 - 1 + r
#define WY_RB 256       // bytes per LDS row (128 channels)

struct Wgrad3yArgs {
  const char* x; const char* dy; float* part;
  long long ksteps;
  int nsplit, steps_per_split;
  int H, W, Wp;
  int spr, rpk, lw;             // K steps per image row (W >= 64) / image rows per K step (W < 64) / log2(W) (6 if W >= 64)
  int Cx, cx_off, Cin; int Cdy, cdy_off, Cout;
  int mtc, ntc;
};

#ifdef INSAR_STAMPS
__device__ unsigned long long g_wgrad3y_stamps[1024 * 8];
#define WY_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[k] += now_ - stamp_prev; stamp_prev = now_; } while (0)
extern "C" int insar_debug_wgrad3y_stamps(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgrad3y_stamps), sizeof(g_wgrad3y_stamps)) != hipSuccess) return -1;
  if (reset) { static unsigned long long z[1024 * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad3y_stamps), z, sizeof(z)) != hipSuccess) return -2; }
  return 0;
}
#else
#define WY_STAMP(k)
#endif

// LDS-DMA of one 1-KB piece: per-lane source = scalar base + 32-bit lane offset, wave-uniform LDS destination in M0
__device__ __forceinline__ void wy_dma(const char* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// LDS image of an operand tile (X: 72 pixel rows x TM channels, dY: 64 x TN), cut for ADDRESSES THAT NEED NO ARITHMETIC:
// the wave at position w of NB (= 4 or 2) along the operand's channel dimension owns the 16-channel blocks t*NB + w,
// t = 0..3 (its four MFMA tiles), and block t*NB + w lives in PLANE t: plane t = [rows][NB blocks x 32 bytes], the NB
// adjacent blocks t*NB .. t*NB + NB - 1 of a pixel (one 128- or 64-byte run of the NHWC row: whole-line gathers). A tile
// index is then a compile-time plane offset, and inside a plane row the 32-byte slot of block w is w ^ f(row) with
// f(row) = (row / RPL) % NB, RPL = 8 / NB rows per 256-byte bank line: the eight pixel rows a 32-lane half of a
// ds_read_b64_tr_b16 touches (r0 .. r0+3 from one 16-lane group, the other four residues mod 8 from its partner) land in
// eight different 32-byte bank segments whatever r0 is — so the tap shift (row + tx) and the pixel half (row + 32) only
// move r0. What depends on the lane — row base, f(row), the 8-byte column inside the block — is folded into ONE base
// register per (k half h, tap tx) (and per pixel half where the halo rows make + 32 pixels a shift that is not a multiple
// of 8 rows: W < 64), kept across the loop and advanced in place by one ring slot per step; every fragment read is that
// register + an immediate. (The first build of this kernel recomputed XOR-swizzled addresses per phase, ~1.5 VALU
// operations per MFMA in the load part: 41 % of its run time, profiles/r04_wgrad3x_ablation.txt.)
template <int NB> __device__ __forceinline__ int wy_f(int row) { return NB == 4 ? (row >> 1) & 3 : (row >> 2) & 1; }

template <int TM, int TN>
struct Wgrad3yCfg {
  static constexpr int THREADS = 256, NW = 4;
  static constexpr int WM = TM / 64, WN = TN / 64;                    // waves along ci / co = blocks per plane row
  static constexpr int RBX = 32 * WM, RBY = 32 * WN;                  // bytes per plane row
  static constexpr int XPL = WY_XR * RBX, YPL = WY_BKP * RBY;         // bytes per plane
  static constexpr int X_STAGE = 4 * XPL, Y_STAGE = 4 * YPL;
  static constexpr int STAGE = X_STAGE + Y_STAGE;                     // 34 KB
  static constexpr int NSLOT = 2;
  static constexpr int LDS_BYTES = NSLOT * STAGE;
  static constexpr int XP = X_STAGE / 1024, YP = Y_STAGE / 1024;      // 1-KB pieces per step: 18 + 16
  static constexpr int NXI = (XP + NW - 1) / NW, NYI = YP / NW;       // DMA instructions per wave per step (the last X one: waves < XP % 4 only)
  static constexpr int XTAIL = XP % NW;                               // waves that issue the last X instruction (0: all)
  static_assert(WM * WN == NW, "four 64 x 64 wave tiles");
  static_assert(YP % NW == 0, "dY stage: whole block-wide DMA instructions");
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two work-groups per CU");
};

template <int N> __device__ __forceinline__ void wy_wait_vm_lgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// SINV: W >= 64 — a K step lies inside one image row, + 32 pixels is + 32 staged rows for X too (an immediate).
template <int TM, int TN, bool SINV>
__global__ __launch_bounds__(256, 2) void wgrad3y_kernel(Wgrad3yArgs a) {
  using Cfg = Wgrad3yCfg<TM, TN>;
  constexpr int NW = Cfg::NW;
  constexpr int NXI = Cfg::NXI, NYI = Cfg::NYI, NPW = NXI + NYI;      // pieces per wave per step (waves beyond XTAIL: one less)
  constexpr int WM = Cfg::WM, WN = Cfg::WN, RBX = Cfg::RBX, RBY = Cfg::RBY;
  constexpr int NSX = SINV ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef INSAR_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long stamp_t0 = stamp_prev, stamp_r0 = __builtin_amdgcn_s_memrealtime();   // [6] / [7]: shader clock vs 100 MHz
#endif
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

  // per-lane byte offsets of its DMA chunks relative to the step's scalar base. Piece q = i*4 + wave holds the 16-byte
  // chunks q*64 + lane of the stage: plane, row, slot in the row -> the block stored there (slot ^ f(row)) and its half.
  // X rows the step does not need (beyond 64 + 2*rpk) are fetched from the last needed row instead: every piece is issued
  // whole, with the same count on every step (counted vmcnt), nothing outside the buffer is touched; those LDS rows are
  // never read.
  const long long xpitch = (long long)a.Cx * 2, ypitch = (long long)a.Cdy * 2;
  const int xneed = WY_BKP + 2 * a.rpk;
  uint32_t xoff_i[NXI], yoff_i[NYI];
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    constexpr int CPR = RBX / 16;
    const int c = (i * NW + wave) * 64 + lane;
    const int plane = c / (WY_XR * CPR), row = (c / CPR) % WY_XR, cc = c % CPR;
    const int blk = plane * WM + ((cc >> 1) ^ wy_f<WM>(row));
    const int srow = row < xneed ? row : xneed - 1;
    xoff_i[i] = (uint32_t)(srow * xpitch) + blk * 32 + (cc & 1) * 16;
  }
#pragma unroll
  for (int i = 0; i < NYI; ++i) {
    constexpr int CPR = RBY / 16;
    const int c = (i * NW + wave) * 64 + lane;
    const int plane = c / (WY_BKP * CPR), row = (c / CPR) % WY_BKP, cc = c % CPR;
    const int blk = plane * WN + ((cc >> 1) ^ wy_f<WN>(row));
    yoff_i[i] = (uint32_t)((row + 2 * (row >> a.lw)) * ypitch) + blk * 32 + (cc & 1) * 16;
  }
  // X row r of a step <-> padded pixel p0 + (ty-1)*Wp - 1 + r; dY row k <-> padded pixel p0 + k + 2*(k / W)
  const char* xbase = a.x + ((long long)(ty - 1) * a.Wp - 1) * xpitch + ((long long)a.cx_off + mi * TM) * 2;
  const char* ybase = a.dy + ((long long)a.cdy_off + ni * TN) * 2;

  // padded index of the first pixel of a K step, advanced step by step (wave-uniform)
  int seg, hrow, img;
  {
    const int g = ks0 / a.spr, gpi = a.H / a.rpk;
    seg = ks0 - g * a.spr;
    img = g / gpi;
    hrow = (g - img * gpi) * a.rpk;
  }
  auto next_pixel = [&]() -> long long {
    const long long p = ((long long)img * (a.H + 2) + hrow + 1) * a.Wp + seg * WY_BKP + 1;
    if (++seg == a.spr) {
      seg = 0; hrow += a.rpk;
      if (hrow >= a.H) { hrow = 0; ++img; }
    }
    return p;
  };
  const uint32_t ldsb = lds_offset_of(smem);
  const uint32_t lds0 = ldsb + wave * 1024;
  const bool xtail = Cfg::XTAIL == 0 || wave < Cfg::XTAIL;            // this wave issues the last X instruction
  // DMA instruction j of a step (j < NXI: X piece j*8 + wave, else dY piece (j - NXI)*8 + wave) into ring slot `slot`
  auto piece = [&](int slot, const char* sx, const char* sy, int j) {
    const uint32_t l = lds0 + slot * Cfg::STAGE;
    if (j < NXI) {
      if (j < NXI - 1 || xtail) wy_dma(sx, xoff_i[j], l + j * (NW * 1024));
    } else {
      wy_dma(sy, yoff_i[j - NXI], l + Cfg::X_STAGE + (j - NXI) * (NW * 1024));
    }
  };

  wy_f32x4_t acc[3][4][4];
#pragma unroll
  for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[t3][i][j] = (wy_f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int wm = wave % WM, wn = wave / WM;
  const int r16 = lane & 15, kq = lane >> 4;

  if (nsteps > 0) {
    WY_STAMP(0);        // set-up
    {
      const long long p0 = next_pixel();
      const char* sx = xbase + p0 * xpitch; const char* sy = ybase + p0 * ypitch;
#pragma unroll
      for (int j = 0; j < NPW; ++j) piece(0, sx, sy, j);
    }
    wy_wait_vm_lgkm<0>();
    __builtin_amdgcn_s_barrier();
    WY_STAMP(1);        // first step landed

    // lane bases of the fragment reads (slot 0): see wgrad3x.hip
    //   dY row of (s, h)      = lrow[h] + 32*s                    lrow[h] = kq*8 + ((h ^ (kq & 1)) << 2) + (r16 >> 2)
    //   X  row of (s, h, tx)  = xrow[h] + s*SS + tx               xrow[h] = lrow[h] + 2*(lrow[h] >> lw), SS = 32 + 2*(32 >> lw)
    uint32_t by[2], bx[NSX][2][3];
    {
      const int SS = 32 + 2 * (32 >> a.lw);
      const int p8 = (r16 & 3) * 8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int lr = kq * 8 + ((h ^ (kq & 1)) << 2) + (r16 >> 2);
        by[h] = ldsb + Cfg::X_STAGE + lr * RBY + ((wn ^ wy_f<WN>(lr)) << 5) + p8;
        const int xr = lr + 2 * (lr >> a.lw);
#pragma unroll
        for (int sx = 0; sx < NSX; ++sx)
#pragma unroll
          for (int t3 = 0; t3 < 3; ++t3) {
            const int r = xr + sx * SS + t3;
            bx[sx][h][t3] = ldsb + r * RBX + ((wm ^ wy_f<WM>(r)) << 5) + p8;
          }
      }
    }
    typedef __attribute__((address_space(3))) wy_s16x4_t* lds_s16x4_p;
    int slot = 0;
    for (int k = 0; k < nsteps; ++k) {
      // the other slot was last read in step k - 1, whose reads every wave finished before the barrier that ended it
      const bool more = k + 1 < nsteps;
      const char* nsx = xbase; const char* nsy = ybase;
      if (more) {
        const long long p1 = next_pixel();
        nsx = xbase + p1 * xpitch; nsy = ybase + p1 * ypitch;
      }
      wy_bf16x8_t yf[4], xf[4];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) {
          const int ph = s * 3 + t3;
          if (t3 == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int nt = 0; nt < 4; ++nt) {
                wy_s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(uintptr_t)(by[h] + (s * 32 * RBY + nt * Cfg::YPL)));
                yf[nt][4 * h + 0] = v[0]; yf[nt][4 * h + 1] = v[1]; yf[nt][4 * h + 2] = v[2]; yf[nt][4 * h + 3] = v[3];
              }
          }
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
              wy_s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (lds_s16x4_p)(uintptr_t)(bx[SINV ? 0 : s][h][t3] + ((SINV ? s * 32 * RBX : 0) + mt * Cfg::XPL)));
              xf[mt][4 * h + 0] = v[0]; xf[mt][4 * h + 1] = v[1]; xf[mt][4 * h + 2] = v[2]; xf[mt][4 * h + 3] = v[3];
            }
          // the pieces of step k + 1, two per phase behind the phase's reads (the last phase issues none)
          if (more) {
#pragma unroll
            for (int j = 2 * ph; j < 2 * ph + 2; ++j)
              if (j < NPW && ph < 5) piece(slot ^ 1, nsx, nsy, j);
          }
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
              acc[t3][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[mt], yf[nt], acc[t3][mt][nt], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
        }
      }
      // the pieces of step k + 1 were issued at least a phase ago, most of them a whole step: wait them out, then meet
      wy_wait_vm_lgkm<0>();
      __builtin_amdgcn_s_barrier();
      const int delta = slot ? -Cfg::STAGE : Cfg::STAGE;
      slot ^= 1;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        by[h] += delta;
#pragma unroll
        for (int sx = 0; sx < NSX; ++sx)
#pragma unroll
          for (int t3 = 0; t3 < 3; ++t3) bx[sx][h][t3] += delta;
      }
    }
  }

  WY_STAMP(2);          // K loop
  // C layout of a 16x16 accumulator: row (ci) = kq*4 + reg, col (co) = r16  ->  16-byte stores into [co][ci];
  // tile mt of wave wm is channel block mt*WM + wm (see the LDS image above)
#pragma unroll
  for (int t3 = 0; t3 < 3; ++t3) {
    float* out = a.part + ((long long)split * 9 + ty * 3 + t3) * a.Cout * a.Cin;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int ci = mi * TM + (mt * WM + wm) * 16 + kq * 4;
        const int co = ni * TN + (nt * WN + wn) * 16 + r16;
        *(wy_f32x4_t*)(out + (long long)co * a.Cin + ci) = acc[t3][mt][nt];
      }
  }
#ifdef INSAR_STAMPS
  WY_STAMP(3);          // slab stores
  stamp_acc[6] = __builtin_amdgcn_s_memtime() - stamp_t0;
  stamp_acc[7] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&g_wgrad3y_stamps[(blockIdx.x & 1023) * 8 + k], stamp_acc[k]);
  }
#endif
}

template <int TM, int TN, bool SINV>
static int launch_wgrad3y_s(Wgrad3yArgs& a, hipStream_t s) {
  using Cfg = Wgrad3yCfg<TM, TN>;
  static std::atomic<uint64_t> attr_mask{0};     // per-device, see common.h
  {
    hipError_t e = insar_set_lds_once(attr_mask, (const void*)wgrad3y_kernel<TM, TN, SINV>, Cfg::LDS_BYTES);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_wgrad_conv3y: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  a.mtc = a.Cin / TM; a.ntc = a.Cout / TN;
  const long long grid = (long long)a.nsplit * 3 * a.mtc * a.ntc;
  if (grid > 0x7fffffffLL) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3y: grid too large");
  hipLaunchKernelGGL((wgrad3y_kernel<TM, TN, SINV>), dim3((unsigned)grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, s, a);
  INSAR_CHECK_LAUNCH("insar_wgrad_conv3y");
  return INSAR_OK;
}

// (128 << 16) | 128 where this kernel applies (bf16, image rows that are whole K steps or K steps that are whole image rows,
// both channel counts multiples of 128), else 0
extern "C" int insar_wgrad_conv3y_tile(const InsarAct* x, int32_t Cout) {
  if (!x || x->dtype != INSAR_BF16) return 0;
  const bool rows_ok = (x->W % WY_BKP) == 0 || ((x->W == 16 || x->W == 32) && (x->H % (WY_BKP / x->W)) == 0);
  if (!rows_ok) return 0;
  return (x->c_len % 128 == 0 && Cout % 128 == 0) ? ((128 << 16) | 128) : 0;
}

// part[split][tap][co][ci] as insar_wgrad_conv3 / insar_wgrad_conv3x write it (same fold: insar_wgrad_reduce)
extern "C" int insar_wgrad_conv3y(const InsarAct* x, const InsarAct* dy, float* part, int32_t nsplit, void* stream) {
  if (!x || !dy || !part) INSAR_FAIL(INSAR_E_ARG, "insar_wgrad_conv3y: null pointer");
  int rc;
  if ((rc = insar_check_act(x, "insar_wgrad_conv3y", "x"))) return rc;
  if ((rc = insar_check_act(dy, "insar_wgrad_conv3y", "dy"))) return rc;
  if (x->B != dy->B || x->H != dy->H || x->W != dy->W) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3y: x/dy grids differ");
  if (x->dtype != dy->dtype) INSAR_FAIL(INSAR_E_DTYPE, "insar_wgrad_conv3y: x/dy dtype differ");
  if (!insar_wgrad_conv3y_tile(x, dy->c_len))
    INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3y: unsupported layer (bf16; W %% 64 == 0, or W = 16 / 32 with whole K steps per image; channels in 128s); use insar_wgrad_conv3");
  if (nsplit < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3y: nsplit");
  if ((long long)WY_XR * x->C * 2 >= 0x7fffffffLL || (long long)WY_XR * dy->C * 2 >= 0x7fffffffLL)
    INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3y: channel pitch too large");
  Wgrad3yArgs a;
  a.x = (const char*)x->ptr; a.dy = (const char*)dy->ptr; a.part = part;
  a.ksteps = (long long)x->B * x->H * x->W / WY_BKP;
  a.nsplit = nsplit;
  a.steps_per_split = (int)((a.ksteps + nsplit - 1) / nsplit);
  a.H = x->H; a.W = x->W; a.Wp = x->W + 2;
  a.spr = x->W >= WY_BKP ? x->W / WY_BKP : 1;
  a.rpk = x->W >= WY_BKP ? 1 : WY_BKP / x->W;
  a.lw = x->W >= WY_BKP ? 6 : (x->W == 32 ? 5 : 4);
  a.Cx = x->C; a.cx_off = x->c_off; a.Cin = x->c_len;
  a.Cdy = dy->C; a.cdy_off = dy->c_off; a.Cout = dy->c_len;
  hipStream_t s = (hipStream_t)stream;
  return a.lw == 6 ? launch_wgrad3y_s<128, 128, true>(a, s) : launch_wgrad3y_s<128, 128, false>(a, s);
}
