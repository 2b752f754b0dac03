// 3x3 / stride-1 convolution (forward and input-gradient) as an implicit GEMM over the FLAT PADDED
// pixel space, for the large-M levels of the U-Net (Unet-ChannalAttention.py:81,84 and their dgrad).
//
// Because activations are NHWC with a one-pixel zero halo, the padded image is one flat array of pixels
// in which the 3x3 neighbours of pixel q are q + dy*(W+2) + dx - also across image-row boundaries. So:
//   * one M tile = 256 consecutive padded pixels; for a given dy the A operand of all three dx taps is
//     the SAME 256-row LDS tile read at row offsets -1/0/+1. A rows are fetched once per (K slab, dy)
//     instead of once per tap: 3x less A traffic through the L2->LDS path, which is what bounds the
//     per-tap kernel (igemm.hip) on these shapes (measured: ~12 TB/s of LDS-DMA traffic whatever the tile).
//   * tiles step by 254 pixels (rows 0 and 255 only serve as neighbours), halo pixels are computed but
//     never stored (and never counted in the BatchNorm partial sums);
//   * A ring of 2 slots (one per dy group), B ring of 3 slots (one weight slab per tap), all by untracked
//     LDS-DMA with counted vmcnt waits; 8 waves (4 x 2), wave tile 64 x BN/2, same MFMA / swizzle /
//     epilogue scheme as igemm.hip.
//   * GEO = 1, "row tiles" (round 3, the deep levels: W in {16, 32, 64, 128, 256}): a tile is 256 REAL output pixels = 256 / W
//     whole image rows, and the A slot holds those rows WITH their two halo pixels each ((256 / W) * (W + 2) <= 320 staged
//     rows): the three dx taps still share one staged A tile, no halo pixel is computed, and the tile count is the per-tap
//     kernel's (M / 256: exactly one round of work-groups where that kernel has one) — the flat geometry's 254-pixel step
//     gives 275 / 292 tiles where 256 fit, and loses there to tile quantisation what it wins on operand traffic.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define FL_BM 256
#define FL_STEP (FL_BM - 2)
#define FL_THREADS 512
#define FL_ROWB 128

#include "flat_args.h"

template <typename T> struct FMma;
template <> struct FMma<bf16_t> {
  __device__ __forceinline__ static void run(const uint4& wa, const uint4& xb, f32x4_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xb), acc, 0, 0, 0);
  }
};
template <> struct FMma<float> {
  __device__ __forceinline__ static void run(const uint4& wa, const uint4& xb, f32x4_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.x), __uint_as_float(xb.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.y), __uint_as_float(xb.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.z), __uint_as_float(xb.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.w), __uint_as_float(xb.w), acc, 0, 0, 0);
  }
};

#define FL_GEO1_ROWS 320                                   // staged A rows of a row tile: (256 / W) * (W + 2) <= 288, in 64-row DMA units
template <typename T, int BN, int GEO = 0>
struct FlatCfg {
  static constexpr int ES = sizeof(T);
  static constexpr int BKe = FL_ROWB / ES;
  static constexpr int A_ROWS = GEO ? FL_GEO1_ROWS : FL_BM;         // GEO: 0 flat pixel space, 1 row tiles, 2 row tiles of a dilated conv
  static constexpr int A_SLOT = A_ROWS * FL_ROWB;         // 32 KB (40 KB: row tiles)
  static constexpr int B_SLOT = BN * FL_ROWB;             // 16 / 8 KB
  static constexpr int A_DMA = A_ROWS * 8 / FL_THREADS;   // 4 (5)
  static constexpr int B_DMA = BN * 8 / FL_THREADS;       // 2 / 1
  static constexpr int RING = 2 * A_SLOT + 3 * B_SLOT;
  static constexpr int PITCH = BN * ES + 16;
  static constexpr int TILE = FL_BM * PITCH;
  static constexpr int MAIN = RING > TILE ? RING : TILE;
  static constexpr int ROWINFO = FL_BM * 8;               // rowOut[256] (int64)
  static constexpr int STATB = 8 * BN * 2 * 4;
  // bf16, 128 columns: the kernel sits at the 256-register limit and the 16 carried BatchNorm sums of a persistent
  // work-group were spilled to scratch (17 / 22 spilled registers, reloaded and stored again at every tile); they live in
  // LDS instead, one 64-byte slot per thread (2 spills left; same-box 7.272 -> 7.256 ms/step)
  static constexpr bool LDS_CARRY = sizeof(T) == 2 && BN == 128 && GEO == 0;     // (row tiles: the 40 KB A slots leave no room)
  static constexpr int CARRYB = LDS_CARRY ? FL_THREADS * 2 * (16 / ES) * 4 : 0;
  static constexpr int LDS_BYTES = MAIN + ROWINFO + STATB + CARRYB;
};

// Diagnostic build only (-DINSAR_STAMPS, tools/stamp_flat.py): s_memtime stamps of the phases of a tile, summed per
// work-group by wave 0; the product library contains none of this.
#ifdef INSAR_STAMPS
__device__ unsigned long long g_flat_stamps[1024 * 8];
#define FL_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[k] += now_ - stamp_prev; stamp_prev = now_; } while (0)
extern "C" int insar_debug_flat_stamps(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_flat_stamps), sizeof(g_flat_stamps)) != hipSuccess) return -1;
  if (reset) { static unsigned long long z[1024 * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_flat_stamps), z, sizeof(z)) != hipSuccess) return -2; }
  return 0;
}
#else
#define FL_STAMP(k)
#endif

template <int N>
__device__ __forceinline__ void fl_wait_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");   // lgkmcnt: see common.h, dma_drain_and_barrier
  __builtin_amdgcn_s_barrier();
}

// PP (bf16): every tap step as one ping-pong phase of the two wave groups (waves 0-3 / 4-7, the two waves of every SIMD), as
// in igemm.hip: [load part: the step's fragment reads + the LDS-DMA pieces the plain loop issues in this step] -> s_barrier
// -> [its 2*NT*MT MFMAs under s_setprio 1] -> s_barrier, group 1 one barrier behind group 0. A staged slab is waited for
// one phase after it was issued and read one phase after that wait. Same accumulation order as the plain loop.
// BS: BatchNorm-backward sums in the statistics slab (InsarBstat; an instantiation of its own: the 128-column kernel is
// at the register limit and the plain launches keep their code).
template <typename T, int BN, bool PP = false, bool BS = false, int GEO = 0>
__global__ __launch_bounds__(FL_THREADS, 2) void conv3x3_flat_kernel(FlatArgs a) {
  using Cfg = FlatCfg<T, BN, GEO>;
  constexpr int ES = Cfg::ES, BKe = Cfg::BKe, CH = Chunk<T>::N;
  constexpr int NT = BN / 32, MT = 4;
  constexpr int AD = Cfg::A_DMA, BD = Cfg::B_DMA;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  long long* rowOut = (long long*)(smem + Cfg::MAIN);
  float* sstat = (float*)(smem + Cfg::MAIN + Cfg::ROWINFO);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Work-groups are PERSISTENT when the grid is smaller than the tile count (a.total_tiles; the launch picks the grid):
  // virtual block vb = blockIdx + k * gridDim walks this work-group's tiles; with gridDim % 8 == 0 it stays on its XCD, so
  // the XCD-aware tile order below is that of the one-tile-per-work-group launch.
  // persistent work-groups with one N tile: the BatchNorm partial sums of all of a work-group's tiles are carried in
  // registers (a thread owns the same channel chunk in every tile) and folded ONCE, into row blockIdx.x of the slab:
  // gridDim.x rows instead of one per tile (no pre-fold launch before bn_finalize), no per-tile fold and barrier
  constexpr int CHc = Chunk<T>::N;
  float cs1[CHc], cs2[CHc];
#pragma unroll
  for (int j = 0; j < CHc; ++j) { cs1[j] = 0.f; cs2[j] = 0.f; }
  // LDS_CARRY only: 16-byte slot q of thread t at [q][t] (consecutive lanes, consecutive slots: no bank conflict)
  float4* carry = (float4*)(smem + Cfg::MAIN + Cfg::ROWINFO + Cfg::STATB) + threadIdx.x;
  constexpr int CST = FL_THREADS;      // stride between a thread's slots
  if constexpr (Cfg::LDS_CARRY) {
#pragma unroll
    for (int j = 0; j < 2 * CHc / 4; ++j) carry[j * CST] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#ifdef INSAR_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // wave-uniform (scalar registers)
  unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif
  for (int vb = blockIdx.x; vb < a.total_tiles; vb += gridDim.x) {
  FL_STAMP(7);          // loop turn-around (+ kernel start for the first tile)
  int t;
  {
    const int nwg = a.total_tiles, bid = vb;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int mtile = t / a.num_ntiles, ntile = t - mtile * a.num_ntiles;
  const int n0 = ntile * BN;
  const int Wp = a.W + 2;
  const long long q0 = (long long)mtile * FL_STEP - 1;      // tile row r <-> padded pixel q0 + r
  const int Pm1 = (int)(a.P - 1);

  // output offsets of the tile's rows: computed AFTER the first LDS-DMA pieces are issued (below), under their flight
  auto fill_row_out = [&]() {
    if constexpr (GEO >= 1) {
      if (tid < FL_BM) {                                      // every row of a row tile is a real output pixel
        const int hw = a.H * a.W;
        const long long m = (long long)mtile * FL_BM + tid;
        const int n = (int)(m / hw);
        const int rem = (int)(m - (long long)n * hw);
        const int h = rem / a.W, wcol = rem - h * a.W;
        rowOut[tid] = ((long long)(n * (a.H + 2) + h + 1) * Wp + wcol + 1) * a.Cy + a.cy_off;
      }
      return;
    }
    if (tid < FL_BM) {
      const int q = (int)q0 + tid;                            // P < 2^31 (checked by the host side)
      long long ro = -1;
      if (tid >= 1 && tid <= FL_STEP && q >= 0 && q <= Pm1) {
        const int img = (a.H + 2) * Wp;
        const int n = q / img;
        const int rem = q - n * img;
        const int hr = rem / Wp, wc = rem - hr * Wp;
        if (hr >= 1 && hr <= a.H && wc >= 1 && wc <= a.W) ro = (long long)q * a.Cy + a.cy_off;
      }
      rowOut[tid] = ro;
    }
  };

  // staging geometry (chunk c = i*512 + tid -> LDS row c>>3, lane-linear position c&7)
  const int srow = tid >> 3;                                 // + 64*i
  const int schunk = ((tid & 7) ^ (srow & 7)) * 16;
  int a_pix[AD];
  unsigned a_ok = 0;                 // dilated row tiles: bit 3 i + dyi = piece i of kernel row dyi reads a real (or halo) pixel, else the zero pixel
  if constexpr (GEO == 2) {
    // as GEO = 1 below with d columns either side of an image row: staged row L = ir * (W + 2 d) + c is column c - d of the
    // tile's image row ir. Taps that reach beyond the one-pixel halo of the activation buffers are read from pixel 0 of the
    // buffer — the top-left halo pixel, zero in every channel (as the per-tap kernel's OOB variant does).
    const int d = a.dil, Ws = a.W + 2 * d;
    const int S = (FL_BM / a.W) * Ws;
    const long long m0 = (long long)mtile * FL_BM;
    const int hw = a.H * a.W;
    const int n0i = (int)(m0 / hw);
    const int h0 = (int)(m0 - (long long)n0i * hw) / a.W;
#pragma unroll
    for (int i = 0; i < AD; ++i) {
      int L = srow + 64 * i;
      L = L < S ? L : S - 1;
      const int ir = L / Ws, c = L - ir * Ws;
      const int win = c - d;                                  // input column, -d .. W + d - 1
      a_pix[i] = (n0i * (a.H + 2) + h0 + ir + 1) * Wp + win + 1;
      const bool colok = win >= -1 && win <= a.W;
#pragma unroll
      for (int dyi = 0; dyi < 3; ++dyi) {
        const int hin = h0 + ir + (dyi - 1) * d;
        if (colok && hin >= -1 && hin <= a.H) a_ok |= 1u << (3 * i + dyi);
      }
    }
  } else if constexpr (GEO == 1) {
    // staged row L = ir * (W + 2) + c: padded column c of the tile's image row ir (dy = 0; stageA adds the dy shift)
    const int S = (FL_BM / a.W) * Wp;
    const long long m0 = (long long)mtile * FL_BM;
    const int hw = a.H * a.W;
    const int n0i = (int)(m0 / hw);
    const int h0 = (int)(m0 - (long long)n0i * hw) / a.W;
#pragma unroll
    for (int i = 0; i < AD; ++i) {
      int L = srow + 64 * i;
      L = L < S ? L : S - 1;                                  // rows beyond the staged span: the last one again
      const int ir = L / Wp, c = L - ir * Wp;
      a_pix[i] = (n0i * (a.H + 2) + h0 + ir + 1) * Wp + c;
    }
  } else {
#pragma unroll
    for (int i = 0; i < AD; ++i) a_pix[i] = (int)q0 + srow + 64 * i;
  }
  const char* xbase = a.x + (long long)a.cx_off * ES + schunk;
  const long long xpitch = (long long)a.Cx * ES;
  const char* wbase = a.w + ((long long)(n0 + srow) * a.K) * ES + schunk;
  const long long w_tap = (long long)a.N * a.K * ES;
  const long long w_row64 = (long long)64 * a.K * ES;
  const uint32_t ldsA = lds_offset_of(smem) + wave * 1024;
  const uint32_t ldsB = ldsA + 2 * Cfg::A_SLOT;

  auto stageA = [&](int slot, int kc, int dyi) {
    const int shift = (dyi - 1) * Wp * (GEO == 2 ? a.dil : 1);
    const long long koff = (long long)kc * BKe * ES;
#pragma unroll
    for (int i = 0; i < AD; ++i) {
      int pix = a_pix[i] + shift;
      if constexpr (GEO == 2) pix = ((a_ok >> (3 * i + dyi)) & 1u) ? pix : 0;
      pix = pix < 0 ? 0 : (pix > Pm1 ? Pm1 : pix);
      lds_dma16_untracked(xbase + (long long)pix * xpitch + koff, ldsA + slot * Cfg::A_SLOT + i * (FL_THREADS * 16));
    }
  };
  auto stageB = [&](int slot, int kc, int g9) {
    const int tap = a.flip ? 8 - g9 : g9;
    const char* wb = wbase + tap * w_tap + (long long)kc * BKe * ES;
#pragma unroll
    for (int i = 0; i < BD; ++i) lds_dma16_untracked(wb + i * w_row64, ldsB + slot * Cfg::B_SLOT + i * (FL_THREADS * 16));
  };

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int wm = wave & 3, wn = wave >> 2;
  const int r16 = lane & 15, kq = lane >> 4;
  const int b_frag = (wn * (BN / 2) + r16) * FL_ROWB;
  const int swb = r16 & 7;
  const int arow0 = wm * 64 + r16;
  // row tiles: pixel m = arow0 + 16 * mt of the tile sits in staged row m + 2 * (m / W) + 1 (two halo pixels per image row
  // before it, one at the head of its own row); tap dx reads one row further per step. W is a power of two >= 16: a 16-pixel
  // fragment never leaves its image row.
  const int kcs = a.kc_count;
  const int nsteps = kcs * 9, ngroups = kcs * 3;

  if constexpr (PP) {
    static_assert(sizeof(T) == 2, "ping-pong loop: bf16");
    const int grp = wave >> 2;
    FL_STAMP(0);        // tile geometry
    stageA(0, 0, 0);
    stageB(0, 0, 0);
    stageB(1, 0, 1);
    FL_STAMP(1);        // issue of the first slabs
    fill_row_out();
    FL_STAMP(2);        // output row offsets
    fl_wait_and_barrier<0>();           // everything landed; rowOut visible
    FL_STAMP(3);        // first slabs landed
    if (grp == 1) __builtin_amdgcn_s_barrier();                 // group 1 runs one barrier behind
    for (int kc = 0; kc < kcs; ++kc) {
#pragma unroll
      for (int dyi = 0; dyi < 3; ++dyi) {
        const int g = kc * 3 + dyi;
        const char* sA = smem + (g & 1) * Cfg::A_SLOT;
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
          const int s = g * 3 + dxi;
          const bool issueA = (dxi == 0) && (g + 1 < ngroups);
          const bool issueB = (s + 2 < nsteps);
          const char* sB = smem + 2 * Cfg::A_SLOT + dxi * Cfg::B_SLOT;
          // ---- load part ----
          uint4 xf[2][MT], wf[2][NT];
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              int row;
              if constexpr (GEO == 2) {
                const int m = arow0 + mt * 16;
                row = m + (m >> a.lw) * (2 * a.dil) + dxi * a.dil;
              } else if constexpr (GEO == 1) {
                const int m = arow0 + mt * 16;
                row = m + ((m >> a.lw) << 1) + dxi;
              } else {
                row = arow0 + mt * 16 + (dxi - 1);
                row = row < 0 ? 0 : (row > FL_BM - 1 ? FL_BM - 1 : row);
              }
              xf[sub][mt] = *(const uint4*)(sA + row * FL_ROWB + (((kq + 4 * sub) ^ (row & 7)) << 4));
            }
            const int pcb = ((kq + 4 * sub) ^ swb) << 4;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[sub][nt] = *(const uint4*)(sB + b_frag + nt * 16 * FL_ROWB + pcb);
          }
          if (issueA) stageA((g + 1) & 1, dyi == 2 ? kc + 1 : kc, dyi == 2 ? 0 : dyi + 1);
          if (issueB) {
            const int dx2 = (dxi + 2) % 3;
            const int gg = g + (dxi + 2) / 3;
            const int kc2 = gg / 3, dy2 = gg - kc2 * 3;
            stageB(dx2, kc2, dy2 * 3 + dx2);
          }
          // everything issued BEFORE this phase has landed (it is at least a phase old); this phase's reads are complete
          if (issueA && issueB) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AD + BD) : "memory");
          else if (issueA) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AD) : "memory");
          else if (issueB) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BD) : "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) FMma<T>::run(wf[sub][nt], xf[sub][mt], acc[nt][mt]);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
        }
      }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                 // group 0 meets group 1's last barrier
  } else {
  // prologue: A(group 0), B(step 0), B(step 1)
  stageA(0, 0, 0);
  stageB(0, 0, 0);
  stageB(1, 0, 1);
  fill_row_out();
  fl_wait_and_barrier<BD>();          // A(0) and B(0) landed, B(1) may still fly; rowOut visible

  for (int kc = 0; kc < kcs; ++kc) {
#pragma unroll
    for (int dyi = 0; dyi < 3; ++dyi) {
      const int g = kc * 3 + dyi;
      const char* sA = smem + (g & 1) * Cfg::A_SLOT;
#pragma unroll
      for (int dxi = 0; dxi < 3; ++dxi) {
        const int s = g * 3 + dxi;
        const bool issueA = (dxi == 0) && (g + 1 < ngroups);
        const bool issueB = (s + 2 < nsteps);
        if (issueA) {
          const int g1 = g + 1;
          stageA(g1 & 1, dyi == 2 ? kc + 1 : kc, dyi == 2 ? 0 : dyi + 1);
        }
        if (issueB) {
          // step s+2: tap index within its group = (dxi+2)%3, group = g + (dxi+2)/3
          const int dx2 = (dxi + 2) % 3;
          const int gg = g + (dxi + 2) / 3;
          const int kc2 = gg / 3, dy2 = gg - kc2 * 3;
          stageB(dx2, kc2, dy2 * 3 + dx2);
        }
        const char* sB = smem + 2 * Cfg::A_SLOT + dxi * Cfg::B_SLOT;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          uint4 xf[MT], wf[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            int row;
            if constexpr (GEO == 2) {
              const int m = arow0 + mt * 16;
              row = m + (m >> a.lw) * (2 * a.dil) + dxi * a.dil;
            } else if constexpr (GEO == 1) {
              const int m = arow0 + mt * 16;
              row = m + ((m >> a.lw) << 1) + dxi;
            } else {
              row = arow0 + mt * 16 + (dxi - 1);
              row = row < 0 ? 0 : (row > FL_BM - 1 ? FL_BM - 1 : row);
            }
            xf[mt] = *(const uint4*)(sA + row * FL_ROWB + (((kq + 4 * sub) ^ (row & 7)) << 4));
          }
          const int pcb = ((kq + 4 * sub) ^ swb) << 4;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) wf[nt] = *(const uint4*)(sB + b_frag + nt * 16 * FL_ROWB + pcb);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) FMma<T>::run(wf[nt], xf[mt], acc[nt][mt]);
        }
        // B(s+1) (and, at the end of a dy group, A(g+1)) must have landed before the next step
        if (dxi == 0) {
          if (issueA && issueB) fl_wait_and_barrier<AD + BD>();
          else if (issueB) fl_wait_and_barrier<BD>();
          else fl_wait_and_barrier<0>();
        } else {
          if (issueB) fl_wait_and_barrier<BD>();
          else fl_wait_and_barrier<0>();
        }
      }
    }
  }
  }
  __syncthreads();
  FL_STAMP(4);          // K loop

  // ---- epilogue (as igemm.hip): registers -> LDS tile -> 16-byte NHWC stores + BN partial sums ------
  char* tile = smem;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = wm * 64 + mt * 16 + r16;
      const int col = wn * (BN / 2) + nt * 16 + kq * 4;
      char* p = tile + row * Cfg::PITCH + col * ES;
      if constexpr (ES == 2) {
        uint2 v;
        v.x = pack2_bf16(acc[nt][mt][0], acc[nt][mt][1]);
        v.y = pack2_bf16(acc[nt][mt][2], acc[nt][mt][3]);
        *(uint2*)p = v;
      } else {
        *(f32x4_t*)p = acc[nt][mt];
      }
    }
  __syncthreads();

  constexpr int CPR = BN * ES / 16;
  constexpr int ITER = FL_BM * CPR / FL_THREADS;
  constexpr int RSTEP = FL_THREADS / CPR;
  const int cc = tid % CPR;
  const long long col_off = n0 + cc * CH;
  float s1[CH], s2[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  // BS (see igemm.hip): the consumer unit's y at the output's positions, half of this thread's chunks requested at a time
  float bsc[BS ? CH : 1], bsh[BS ? CH : 1];
  if constexpr (BS) {
#pragma unroll
    for (int j = 0; j < CH; ++j) { bsc[j] = a.bscale[col_off + j]; bsh[j] = a.bshift[col_off + j]; }
  }
  constexpr int HALF = ITER / 2;
  static_assert(ITER % 2 == 0, "epilogue chunk batches");
#pragma unroll
  for (int i0 = 0; i0 < ITER; i0 += HALF) {
    uint4 yv[BS ? HALF : 1];
    if constexpr (BS) {
#pragma unroll
      for (int i = 0; i < HALF; ++i) {
        const long long ro = rowOut[(i0 + i) * RSTEP + tid / CPR];
        yv[i] = ro >= 0 ? *(const uint4*)(a.by + (ro + col_off) * ES) : make_uint4(0u, 0u, 0u, 0u);
      }
    }
#pragma unroll
    for (int i = 0; i < HALF; ++i) {
      const int row = (i0 + i) * RSTEP + tid / CPR;
      const long long ro = rowOut[row];
      if (ro >= 0) {
        const uint4 u = *(const uint4*)(tile + row * Cfg::PITCH + cc * 16);
        float f[CH];
        Chunk<T>::unpack(u, f);
        if constexpr (BS) {
          float yy[CH];
          Chunk<T>::unpack(yv[i], yy);
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const float m = fmaf(yy[j], bsc[j], bsh[j]) > 0.f ? f[j] : 0.f;
            s1[j] += m; s2[j] = fmaf(m, yy[j], s2[j]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < CH; ++j) { s1[j] += f[j]; s2[j] = fmaf(f[j], f[j], s2[j]); }
        }
        *(uint4*)(a.y + (ro + col_off) * ES) = u;
      }
    }
  }
  if (GEO == 0 && a.stats && a.carry) {            // (row tiles never carry: one slab row per tile)
    if constexpr (Cfg::LDS_CARRY) {
      // same sums in the same order as the register carry: slot j of the thread += this tile's sum
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) {
        float4 c1v = carry[q * CST], c2v = carry[(CH / 4 + q) * CST];
        c1v.x += s1[4 * q]; c1v.y += s1[4 * q + 1]; c1v.z += s1[4 * q + 2]; c1v.w += s1[4 * q + 3];
        c2v.x += s2[4 * q]; c2v.y += s2[4 * q + 1]; c2v.z += s2[4 * q + 2]; c2v.w += s2[4 * q + 3];
        carry[q * CST] = c1v; carry[(CH / 4 + q) * CST] = c2v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < CH; ++j) { cs1[j] += s1[j]; cs2[j] += s2[j]; }
    }
  } else if (a.stats) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
#pragma unroll
      for (int o = CPR; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) { LDS_PIN(s1[j]); LDS_PIN(s2[j]); }
    if (lane < CPR) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        sstat[(wave * BN + lane * CH + j) * 2 + 0] = s1[j];
        sstat[(wave * BN + lane * CH + j) * 2 + 1] = s2[j];
      }
    }
    LDS_DRAIN();
#pragma unroll
    for (int j = 0; j < CH; ++j) { LDS_KEEP(s1[j]); LDS_KEEP(s2[j]); }
    __syncthreads();
    if (tid < BN) {
      float v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { v1 += sstat[(w * BN + tid) * 2 + 0]; v2 += sstat[(w * BN + tid) * 2 + 1]; }
      a.stats[((long long)mtile * 2 + 0) * a.N + n0 + tid] = v1;
      a.stats[((long long)mtile * 2 + 1) * a.N + n0 + tid] = v2;
    }
  }
  FL_STAMP(5);          // epilogue: transpose through LDS, stores, statistics
  __syncthreads();          // the next tile's LDS-DMA rewrites the ring the epilogue tile aliases
  FL_STAMP(6);          // the other waves' epilogues
  }
  if (GEO == 0 && a.stats && a.carry) {
    constexpr int CPR = BN * ES / 16;
    constexpr int CH = CHc;
    if constexpr (Cfg::LDS_CARRY) {
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) {
        const float4 c1v = carry[q * CST], c2v = carry[(CH / 4 + q) * CST];
        cs1[4 * q] = c1v.x; cs1[4 * q + 1] = c1v.y; cs1[4 * q + 2] = c1v.z; cs1[4 * q + 3] = c1v.w;
        cs2[4 * q] = c2v.x; cs2[4 * q + 1] = c2v.y; cs2[4 * q + 2] = c2v.z; cs2[4 * q + 3] = c2v.w;
      }
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
#pragma unroll
      for (int o = CPR; o < 64; o <<= 1) { cs1[j] += __shfl_xor(cs1[j], o, 64); cs2[j] += __shfl_xor(cs2[j], o, 64); }
    }
    if (lane < CPR) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        sstat[(wave * BN + lane * CH + j) * 2 + 0] = cs1[j];
        sstat[(wave * BN + lane * CH + j) * 2 + 1] = cs2[j];
      }
    }
    __syncthreads();
    if (tid < BN) {
      float v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { v1 += sstat[(w * BN + tid) * 2 + 0]; v2 += sstat[(w * BN + tid) * 2 + 1]; }
      a.stats[((long long)blockIdx.x * 2 + 0) * a.N + tid] = v1;       // one N tile: n0 = 0
      a.stats[((long long)blockIdx.x * 2 + 1) * a.N + tid] = v2;
    }
  }
#ifdef INSAR_STAMPS
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&g_flat_stamps[(blockIdx.x & 1023) * 8 + k], stamp_acc[k]);
  }
#endif
}

// ---- host side -----------------------------------------------------------------------------------
static inline long long flat_pixels(const InsarAct& x) { return (long long)x.B * (x.H + 2) * (x.W + 2); }
static inline int flat_mtiles(long long P) { return (int)((P + FL_STEP - 1) / FL_STEP); }

// The flat kernel pays for computing halo pixels and needs enough tiles to fill the chip.
extern "C" int insar_conv3x3_flat_ok(const InsarAct* x, int32_t N) {
  if (!x || x->W < 30 || x->H < 30) return 0;
  // measured (tools/gemm_bench.py, B=16): the shared-A tile wins where the GEMM is shallow (K, N <= 128:
  // the 256^2 and 128^2 levels); for deeper K the per-tap 256-row kernel is faster (halo rows cost more
  // than the A re-reads it saves once the weight slabs dominate the LDS traffic).
  if (x->c_len > 128 || N > 128) return 0;
  const long long P = flat_pixels(*x);
  if (P >= 0x7fffffffLL) return 0;
  const int bn = (N % 128) == 0 ? 128 : 64;
  return (long long)flat_mtiles(P) * (N / bn) >= 256 ? 1 : 0;
}
extern "C" int insar_conv3x3_flat_num_mtiles(const InsarAct* x) { return x ? flat_mtiles(flat_pixels(*x)) : 0; }
// Row tiles (flip bit 3): 256 real output pixels = 256 / W whole image rows per tile. bf16, W a power of two in 16 .. 256,
// H a multiple of 256 / W (a tile never straddles two images).
static inline bool flat_rows_geometry(const InsarAct& x, int dil = 1) {
  if (x.dtype != INSAR_BF16 || dil < 1 || dil > 15) return false;
  if (x.W < 16 || x.W > FL_BM || (FL_BM % x.W) != 0 || (x.W & (x.W - 1)) != 0) return false;
  if (x.H % (FL_BM / x.W)) return false;
  if ((FL_BM / x.W) * (x.W + 2 * dil) > FL_GEO1_ROWS) return false;      // the staged rows of a tile must fit the A slot
  return flat_pixels(x) < 0x7fffffffLL;
}
extern "C" int insar_conv3x3_flat_rows_ok(const InsarAct* x, int32_t N) {
  return (x && (N % 64) == 0 && (x->c_len % 64) == 0 && flat_rows_geometry(*x)) ? 1 : 0;
}
// ... with taps dilated by `dil` (flip bits 8-11; padding = dil: DeepLabV3's layer3 / layer4, DeepLabV3-ChannelAttention.py:87-137
// through torchvision's replace_stride_with_dilation): the staged rows carry dil columns either side, 256 / W * (W + 2 dil) <= 320
extern "C" int insar_conv3x3_flat_rows_dil_ok(const InsarAct* x, int32_t N, int32_t dil) {
  return (x && (N % 64) == 0 && (x->c_len % 64) == 0 && flat_rows_geometry(*x, dil)) ? 1 : 0;
}
extern "C" int insar_conv3x3_flat2_rows_ok(const InsarAct* x, int32_t N) {
  return (x && (N % 64) == 0 && (x->c_len % 32) == 0 && insar_flat2_rows_geometry(*x)) ? 1 : 0;
}
// Rows of the statistics slab a launch with these flags writes: one per M tile, or — persistent work-groups (flip bit 2)
// with one N tile — one per work-group.
extern "C" int insar_conv3x3_flat_stat_rows(const InsarAct* x, int32_t N, int32_t flip) {
  if (!x) return 0;
  const int mt = (flip & 8) ? (int)(((long long)x->B * x->H * x->W) / FL_BM) : flat_mtiles(flat_pixels(*x));
  if (!(flip & 4) || ((flip & 8) && !(flip & 32))) return mt;      // (the two-work-group kernel's row tiles have a persistent form)
  const int bn = ((N % 128) == 0 && !(flip & 16)) ? 128 : 64;
  const int cus = (flip & 32) ? insar_flat2_persistent_grid(bn) : (insar_num_cus() & ~7);      // bit 5: two (three) work-groups per CU
  const long long grid = (long long)mt * (N / bn);
  return (cus >= 8 && grid > cus && N / bn == 1) ? cus : mt;
}

template <typename T, int BN, bool PP = false, bool BS = false, int GEO = 0>
static int launch_flat(FlatArgs& a, hipStream_t s) {
  using Cfg = FlatCfg<T, BN, GEO>;
  static std::atomic<uint64_t> attr_mask{0};     // per-device, see common.h
  {
    hipError_t e = insar_set_lds_once(attr_mask, (const void*)conv3x3_flat_kernel<T, BN, PP, BS, GEO>, Cfg::LDS_BYTES);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_conv3x3_flat: hipFuncSetAttribute(%d bytes LDS): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
  }
  a.num_ntiles = a.N / BN;
  long long grid = (long long)a.num_mtiles * a.num_ntiles;
  a.total_tiles = (int)grid;
  a.carry = 0;
  if (a.persist) {                                   // one work-group per CU (the LDS allows no more), each walking its tiles
    const int cus = insar_num_cus() & ~7;
    if (cus >= 8 && grid > cus) { grid = cus; a.carry = (a.num_ntiles == 1 && GEO == 0) ? 1 : 0; }
  }
  hipLaunchKernelGGL((conv3x3_flat_kernel<T, BN, PP, BS, GEO>), dim3((unsigned)grid), dim3(FL_THREADS), Cfg::LDS_BYTES, s, a);
  INSAR_CHECK_LAUNCH("insar_conv3x3_flat");
  return INSAR_OK;
}

// y = conv3x3(x, w) over the same (B, H, W) grid. w: [9][N][K] in (dy, dx) raster order of the FORWARD
// taps; flip != 0 walks the slabs backwards (the dgrad operand produced by insar_weight_prep keeps the
// forward raster order of (r, s), whose spatial offsets are (1-r, 1-s)).
static int flat_impl(const InsarAct* x, const InsarAct* y, const void* w, int32_t flip, float* stats, const InsarBstat* bstat,
                     void* stream) {
  if (!x || !y || !w) INSAR_FAIL(INSAR_E_ARG, "insar_conv3x3_flat: null pointer");
  int rc;
  if ((rc = insar_check_act(x, "insar_conv3x3_flat", "x"))) return rc;
  if ((rc = insar_check_act(y, "insar_conv3x3_flat", "y"))) return rc;
  if (x->dtype != y->dtype) INSAR_FAIL(INSAR_E_DTYPE, "insar_conv3x3_flat: dtype differ");
  if (x->B != y->B || x->H != y->H || x->W != y->W) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_flat: x/y grids differ");
  const int es = x->dtype == INSAR_BF16 ? 2 : 4;
  const int bke = (flip & 32) ? 32 : FL_ROWB / es;      // bit 5 (conv3x3_flat2.hip): 32-channel slabs
  if (x->c_len % bke) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_flat: K=%d must be a multiple of %d", x->c_len, bke);
  if (y->c_len % 64) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_flat: N=%d must be a multiple of 64", y->c_len);
  if (!insar_aligned16(w)) INSAR_FAIL(INSAR_E_ALIGN, "insar_conv3x3_flat: weights not 16-byte aligned");
  const long long P = flat_pixels(*x);
  if (P >= 0x7fffffffLL) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_flat: too many pixels");
  FlatArgs a;
  a.x = (const char*)x->ptr; a.w = (const char*)w; a.y = (char*)y->ptr; a.stats = stats;
  a.by = nullptr; a.bscale = a.bshift = nullptr;
  if (bstat && bstat->y) {
    if (!stats || !bstat->scale || !bstat->shift) INSAR_FAIL(INSAR_E_ARG, "insar_conv3x3_flat_bstat: needs a stats slab, scale and shift");
    if (!insar_aligned16(bstat->y)) INSAR_FAIL(INSAR_E_ALIGN, "insar_conv3x3_flat_bstat: bstat.y not 16-byte aligned");
    a.by = (const char*)bstat->y; a.bscale = bstat->scale; a.bshift = bstat->shift;
  }
  a.P = P; a.B = x->B; a.H = x->H; a.W = x->W; a.lw = 0;
  a.Cx = x->C; a.cx_off = x->c_off; a.K = x->c_len;
  a.Cy = y->C; a.cy_off = y->c_off; a.N = y->c_len;
  a.kc_count = x->c_len / bke; a.flip = (flip & 1) ? 1 : 0; a.persist = (flip & 4) ? 1 : 0;
  const bool pp = (flip & 2) != 0;
  a.num_mtiles = flat_mtiles(P);
  hipStream_t s = (hipStream_t)stream;
  const bool wide = (a.N % 128) == 0 && !(flip & 16);      // bit 4: 64-column tiles whatever N (grids of 256 work-groups on the 16^2 level)
  a.dil = 1;
  if ((flip & 8) && (flip & 32)) {      // row tiles of the two-work-group kernel (conv3x3_flat2.hip)
    if (!insar_flat2_rows_geometry(*x)) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_flat: the two-work-group kernel's row tiles need bf16, W a power of two in 16..256 and H a multiple of 256 / W (got %d x %d)", x->H, x->W);
    if ((flip >> 8) & 15) INSAR_FAIL(INSAR_E_ARG, "insar_conv3x3_flat: the two-work-group kernel has no dilated form");
    a.num_mtiles = (int)(((long long)x->B * x->H * x->W) / FL_BM);
    while ((1 << a.lw) < x->W) ++a.lw;
    return insar_flat2_launch(a, wide ? 128 : 64, a.by != nullptr, true, s);
  }
  if (flip & 8) {       // row tiles (bf16, ping-pong loop)
    a.dil = ((flip >> 8) & 15) ? ((flip >> 8) & 15) : 1;
    if (!flat_rows_geometry(*x, a.dil)) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_flat: row tiles need bf16, W a power of two in 16..256, H a multiple of 256 / W and 256 / W * (W + 2 * dilation) <= 320 (got %d x %d, dilation %d)", x->H, x->W, a.dil);
    // the channel conditions insar_conv3x3_flat_rows_ok / _dil_ok promise, re-checked for a direct caller of the C ABI
    if (x->c_len % 64 || y->c_len % 64) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_flat: row tiles need K and N multiples of 64 (got %d, %d)", x->c_len, y->c_len);
    if (flip & 4) INSAR_FAIL(INSAR_E_ARG, "insar_conv3x3_flat: the 8-wave kernel's row tiles (flip bit 3 without bit 5) have no persistent form (flip bit 2)");
    a.num_mtiles = (int)(((long long)x->B * x->H * x->W) / FL_BM);
    a.lw = 0;
    while ((1 << a.lw) < x->W) ++a.lw;
    if (a.dil > 1) {
      if (a.by) return wide ? launch_flat<bf16_t, 128, true, true, 2>(a, s) : launch_flat<bf16_t, 64, true, true, 2>(a, s);
      return wide ? launch_flat<bf16_t, 128, true, false, 2>(a, s) : launch_flat<bf16_t, 64, true, false, 2>(a, s);
    }
    if (a.by) return wide ? launch_flat<bf16_t, 128, true, true, 1>(a, s) : launch_flat<bf16_t, 64, true, true, 1>(a, s);
    return wide ? launch_flat<bf16_t, 128, true, false, 1>(a, s) : launch_flat<bf16_t, 64, true, false, 1>(a, s);
  }
  if (flip & 32) {      // two co-resident 4-wave work-groups per CU (conv3x3_flat2.hip): bf16, flat geometry
    if (x->dtype != INSAR_BF16) INSAR_FAIL(INSAR_E_DTYPE, "insar_conv3x3_flat: the two-work-group kernel (flip bit 5) is bf16 only");
    return insar_flat2_launch(a, wide ? 128 : 64, a.by != nullptr, false, s);
  }
  if (a.by) {           // bf16: the ping-pong loop whatever the flag says (same results bit for bit)
    if (x->dtype == INSAR_BF16) return wide ? launch_flat<bf16_t, 128, true, true>(a, s) : launch_flat<bf16_t, 64, true, true>(a, s);
    return wide ? launch_flat<float, 128, false, true>(a, s) : launch_flat<float, 64, false, true>(a, s);
  }
  if (x->dtype == INSAR_BF16 && pp) return wide ? launch_flat<bf16_t, 128, true>(a, s) : launch_flat<bf16_t, 64, true>(a, s);
  if (x->dtype == INSAR_BF16) return wide ? launch_flat<bf16_t, 128>(a, s) : launch_flat<bf16_t, 64>(a, s);
  return wide ? launch_flat<float, 128>(a, s) : launch_flat<float, 64>(a, s);
}

extern "C" int insar_conv3x3_flat(const InsarAct* x, const InsarAct* y, const void* w, int32_t flip, float* stats,
                                  void* stream) {
  return flat_impl(x, y, w, flip, stats, nullptr, stream);
}
extern "C" int insar_conv3x3_flat_bstat(const InsarAct* x, const InsarAct* y, const void* w, int32_t flip, float* stats,
                                        const InsarBstat* bstat, void* stream) {
  return flat_impl(x, y, w, flip, stats, bstat, stream);
}
