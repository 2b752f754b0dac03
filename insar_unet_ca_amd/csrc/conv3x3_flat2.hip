// 3x3 / stride-1 convolution over the flat padded pixel space (conv3x3_flat.hip's GEO 0: forward and input gradient of the
// 256^2 and 128^2 levels, Unet-ChannalAttention.py:81,84 and their dgrad), built for TWO CO-RESIDENT WORK-GROUPS PER CU.
//
// Why a second build. The 8-wave kernel owns its CU (112 KB of LDS): whatever one work-group does outside its K loop —
// tile geometry, the output-offset table with its divisions, the wait for the first slabs, the transposing epilogue, the
// BatchNorm sums — runs with the matrix pipes idle, and at the shallow levels K is short (9-18 tap steps of 64 channels), so
// that serial part is 25-45 % of a launch (profiles/r04_stamps_flat.txt). Here a work-group is 4 waves (one per SIMD) on the
// same 256-pixel x BN tile with HALF-WIDTH K slabs (32 channels = 64-byte LDS rows): ring 2 x 16 KB (A, one slot per dy
// group) + 3 x 8 KB (B, one slot per tap) = 56 KB, epilogue tile 68 KB, 74 KB in all — two work-groups fit the 160 KB (three
// on 64-column tiles: 49 KB, 64 accumulator registers), each
// wave has the 256-register budget of two waves per SIMD, and one group's prologue / epilogue runs under the other's K loop
// (the hardware alternates the two waves of a SIMD by itself: the MFMA pipe is the one thing they cannot both have).
// Second effect: the wave tile is 128 pixels x 64 channels (BN = 128; 64 x 64 for BN = 64): 8 + 4 fragment reads per 32
// MFMAs instead of 8 + 8 — a quarter less LDS read traffic per FLOP, on the path that bounds these kernels.
//
// One tap step = [counted vmcnt wait -> s_barrier -> LDS-DMA of step s+2's weights (and, at dx = 0, of the next dy group's
// pixels) -> 12 ds_read_b128 -> 32 MFMAs]; a slab is waited for two steps after it was issued (never vmcnt(0) inside the
// loop), all LDS-DMA issued from inline asm (common.h). Fragment addresses are one lane base per dx tap plus an immediate:
// 64-byte rows put four rows in a 256-byte bank line, the 16-byte chunk of channel group q of row r sits at slot
// q ^ (2 * bit2(r)) — conflict-free for ds_read_b128's four 16-lane groups at ANY row offset (the dx shift), and the
// 16-row MFMA tiles of a wave differ by multiples of 16 rows, which leave bit 2 alone.
//
// Results: the K loop adds the same products in channel order 32 by 32 instead of 64 by 64 per tap — equal to the 8-wave
// kernel up to fp32 summation order (tests/test_parity_gpu.py: both against the float64 oracle; bit-reproducible run to run).
#include "common.h"
#include "flat_args.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short f2_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f2_f32x4_t;

#define F2_BM 256
#define F2_STEP (F2_BM - 2)
#define F2_THREADS 256
#define F2_ROWB 64
// work-groups per CU (= waves per SIMD): 64-column tiles need 49 KB of LDS and 64 accumulator registers — three fit
#ifndef F2_WGS64
#define F2_WGS64 3
#endif
#define F2_WGS(bn) ((bn) == 64 ? F2_WGS64 : 2)

template <int BN, int GEO = 0>
struct Flat2Cfg {
  static constexpr int NBS = 3;                            // weight slots (one per tap of a dy group)
  static constexpr int A_SLOT = F2_BM * F2_ROWB;           // 16 KB
  static constexpr int B_SLOT = BN * F2_ROWB;              // 8 / 4 KB
  static constexpr int AD = F2_BM * 4 / F2_THREADS;        // 4 LDS-DMA pieces per thread and A slab
  static constexpr int BD = BN * 4 / F2_THREADS;           // 2 / 1
  static constexpr int A_BASE = NBS * B_SLOT;              // the weight ring first: row -1 of A slot 0 is addressable
  static constexpr int RING = A_BASE + 2 * A_SLOT + F2_ROWB;
  static constexpr int PITCH = BN * 2 + 16;
  static constexpr int TILE = F2_BM * PITCH;
  static constexpr int MAIN = RING > TILE ? RING : TILE;
  static constexpr int ROWINFO = F2_BM * 8;
  static constexpr int STATB = 4 * BN * 2 * 4;
  static constexpr int LDS_BYTES = MAIN + ROWINFO + STATB;
  static_assert(F2_WGS(BN) * LDS_BYTES <= 160 * 1024, "co-resident work-groups per CU");
};

#ifdef INSAR_STAMPS
__device__ unsigned long long g_flat2_stamps[1024 * 8];
#define F2_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[k] += now_ - stamp_prev; stamp_prev = now_; } while (0)
extern "C" int insar_debug_flat2_stamps(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_flat2_stamps), sizeof(g_flat2_stamps)) != hipSuccess) return -1;
  if (reset) { static unsigned long long z[1024 * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_flat2_stamps), z, sizeof(z)) != hipSuccess) return -2; }
  return 0;
}
#else
#define F2_STAMP(k)
#endif

// LDS-DMA of one 1-KB piece: per-lane source = scalar base + 32-bit lane offset, wave-uniform LDS destination in M0
__device__ __forceinline__ void f2_dma(const char* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void f2_wait_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");   // lgkmcnt: common.h, dma_drain_and_barrier
  __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ int f2_swz(int row) { return ((row >> 2) & 1) << 1; }

// GEO 1, "row tiles" (W a power of two in 16 .. 256): a tile is 256 REAL output pixels = 256 / W whole image rows, and
// the A slot holds exactly those 256 pixels (of the image row dy - 1 .. dy + 1 above / below): the halo pixel either side of
// an image row is zero in memory, so it is not staged — the one fragment per dx = -1 / +1 step whose lane 0 / lane 15 would
// read it (the first / last 16 pixels of an image row) has that lane's registers cleared instead (4 v_and; at W >= 128 one
// fragment per wave and step, at W = 16 every fragment). No halo pixel is
// computed, the tile count is M / 256 — 1024 / 4096 tiles at B = 16 where the flat geometry's 254-pixel step gives
// 1065 / 4194: exactly 2 / 8 tiles for each of the 512 persistent work-groups instead of 3 / 9 for the unlucky ones.
template <int BN, bool BS, int GEO>
__global__ __launch_bounds__(F2_THREADS, F2_WGS(BN)) void conv3x3_flat2_kernel(FlatArgs a) {
  using Cfg = Flat2Cfg<BN, GEO>;
  constexpr int CH = 8, NT = 4;
  constexpr int WGM = BN == 128 ? 2 : 4;                  // waves along the pixel dimension
  constexpr int WROWS = F2_BM / WGM, MT = WROWS / 16;     // 128 x 64 (8 x 4 MFMA tiles) / 64 x 64 (4 x 4) per wave
  constexpr int AD = Cfg::AD, BD = Cfg::BD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  long long* rowOut = (long long*)(smem + Cfg::MAIN);
  float* sstat = (float*)(smem + Cfg::MAIN + Cfg::ROWINFO);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float cs1[CH], cs2[CH];                                 // BatchNorm sums carried over a persistent work-group's tiles
#pragma unroll
  for (int j = 0; j < CH; ++j) { cs1[j] = 0.f; cs2[j] = 0.f; }

  // ---- what does not depend on the tile ----
  const int Wp = a.W + 2;
  const int Pm1 = (int)(a.P - 1);
  const int xpitch = a.Cx * 2;
  const int srow = tid >> 2;                                            // staged row (+ 64 i) of this thread's DMA chunks
  const int schunk = ((tid & 3) ^ f2_swz(srow)) * 16;                   // the source chunk its lane-linear LDS position holds
  const int xlo = a.cx_off * 2 + schunk;                                // lane offset inside a pixel (channel slice + this lane's chunk)
  uint32_t w_off[BD];
#pragma unroll
  for (int i = 0; i < BD; ++i) w_off[i] = (uint32_t)((srow + 64 * i) * a.K * 2 + schunk);
  const long long w_tap = (long long)a.N * a.K * 2;
  const uint32_t lds0 = lds_offset_of(smem);
  const uint32_t ldsB = lds0 + wave * 1024;
  const uint32_t ldsA = lds0 + Cfg::A_BASE + wave * 1024;

  const int wm = wave % WGM, wn = wave / WGM;
  const int r16 = lane & 15, kq = lane >> 4;
  int aBase[3];                                                         // byte offset in smem of this lane's fragment row, per dx
#pragma unroll
  for (int dxi = 0; dxi < 3; ++dxi) {
    const int R = wm * WROWS + r16 + dxi - 1;                           // -1 .. 256 (flat: rows 0 / 255 of a tile are neighbours only)
    aBase[dxi] = Cfg::A_BASE + R * F2_ROWB + ((kq ^ f2_swz(R)) << 4);
  }
  const int bBase = (wn * 64 + r16) * F2_ROWB + ((kq ^ f2_swz(r16)) << 4);
  // row tiles: the lane whose dx = -1 / +1 neighbour is the (unstaged, zero) halo pixel of its image row — lane 0 / 15 of the
  // 16-pixel fragments that begin / end an image row (bit mt of edgeL / edgeR; W >= 16: a fragment never straddles two rows)
  const uint32_t keepL = r16 == 0 ? 0u : 0xffffffffu, keepR = r16 == 15 ? 0u : 0xffffffffu;
  uint32_t edgeL = 0, edgeR = 0;
  if constexpr (GEO == 1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (((wm * WROWS + mt * 16) & (a.W - 1)) == 0) edgeL |= 1u << mt;
      if (((wm * WROWS + mt * 16 + 16) & (a.W - 1)) == 0) edgeR |= 1u << mt;
    }
  }
  const int kcs = a.kc_count;                                           // 32-channel slabs

#ifdef INSAR_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#endif
  for (int vb = blockIdx.x; vb < a.total_tiles; vb += gridDim.x) {
    F2_STAMP(7);
    int t;
    {
      const int nwg = a.total_tiles, bid = vb;
      const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
      t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int mtile = t / a.num_ntiles, ntile = t - mtile * a.num_ntiles;
    const int n0 = ntile * BN;
    const int q0 = mtile * F2_STEP - 1;                                 // flat: tile row r <-> padded pixel q0 + r (P < 2^31: host)
    int prow0 = 0;                                                      // row tiles: padded row of the tile's first image row
    if constexpr (GEO == 1) {
      const int hw = a.H * a.W;
      const int m0 = mtile * F2_BM;
      const int n0i = m0 / hw;
      prow0 = n0i * (a.H + 2) + ((m0 - n0i * hw) >> a.lw) + 1;
    }

    // Lane offsets are 32-bit and RELATIVE to the tile's lowest staged pixel (scalar 64-bit base): whatever the size of the
    // activation buffer, a tile spans 256 pixels + an image row either side.
    const int base_pix = GEO == 1 ? (prow0 - 1) * Wp : (q0 - Wp > 0 ? q0 - Wp : 0);
    const char* xtile = a.x + (long long)base_pix * xpitch;
    int a_off[AD];                                                      // this thread's staged pixels (dy = 0)
#pragma unroll
    for (int i = 0; i < AD; ++i) {
      if constexpr (GEO == 1) {
        const int m = srow + 64 * i;                                    // pixel m of the tile: image row m / W, column m % W
        a_off[i] = ((1 + (m >> a.lw)) * Wp + (m & (a.W - 1)) + 1) * xpitch + xlo;
      } else {
        a_off[i] = (q0 + srow + 64 * i - base_pix) * xpitch + xlo;
      }
    }
    // flat geometry: the first / last tiles reach beyond the buffer — clamp to pixel 0 / P - 1 (relative, saturated far from the ends)
    int xmin = 0, xmax = 0;
    if constexpr (GEO == 0) {
      const long long lo = xlo - (long long)base_pix * xpitch, hi = xlo + (long long)(Pm1 - base_pix) * xpitch;
      xmin = lo < -(1 << 30) ? -(1 << 30) : (int)lo;
      xmax = hi > (1 << 30) ? (1 << 30) : (int)hi;
    }
    const char* wtile = a.w + (long long)n0 * a.K * 2;

    auto stageA = [&](int slot, int kc, int dyi) {
      const char* sb = xtile + kc * F2_ROWB;
      const int shift = (dyi - 1) * Wp * xpitch;
#pragma unroll
      for (int i = 0; i < AD; ++i) {
        int off = a_off[i] + shift;
        if constexpr (GEO == 0) off = off < xmin ? xmin : (off > xmax ? xmax : off);
        f2_dma(sb, (uint32_t)off, ldsA + slot * Cfg::A_SLOT + i * (F2_THREADS * 16));
      }
    };
    auto stageB = [&](int slot, int kc, int g9) {
      const int tap = a.flip ? 8 - g9 : g9;
      const char* sb = wtile + tap * w_tap + kc * F2_ROWB;
#pragma unroll
      for (int i = 0; i < BD; ++i) f2_dma(sb, w_off[i], ldsB + slot * Cfg::B_SLOT + i * (F2_THREADS * 16));
    };
    auto fill_row_out = [&]() {
      if constexpr (GEO == 1) {                                         // every row of a row tile is a real output pixel
        const int ir = tid >> a.lw, wcol = tid & (a.W - 1);
        rowOut[tid] = ((long long)(prow0 + ir) * Wp + wcol + 1) * a.Cy + a.cy_off;
        return;
      }
      const int q = q0 + tid;
      long long ro = -1;
      if (tid >= 1 && tid <= F2_STEP && q >= 0 && q <= Pm1) {
        const int img = (a.H + 2) * Wp;
        const int n = q / img;
        const int rem = q - n * img;
        const int hr = rem / Wp, wc = rem - hr * Wp;
        if (hr >= 1 && hr <= a.H && wc >= 1 && wc <= a.W) ro = (long long)q * a.Cy + a.cy_off;
      }
      rowOut[tid] = ro;
    };

    f2_f32x4_t acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = (f2_f32x4_t){0.f, 0.f, 0.f, 0.f};

    F2_STAMP(0);          // tile geometry
    stageA(0, 0, 0);
    stageB(0, 0, 0);
    stageB(1, 0, 1);
    F2_STAMP(1);          // issue of the first slabs
    fill_row_out();
    F2_STAMP(2);          // output row offsets

    f2_wait_and_barrier<BD>();          // A(0) and B(0) landed, B(1) may still fly; rowOut visible
    F2_STAMP(3);          // first slabs landed
    // one 32-channel slab = 9 tap steps. LAST: the tile's last slab, whose final steps have nothing left to prefetch (the
    // flags are compile-time there, so that neither copy of the loop body carries a branch)
    auto slab = [&](const int kc, auto last_c) {
      constexpr bool LAST = decltype(last_c)::value;
#pragma unroll
      for (int dyi = 0; dyi < 3; ++dyi) {
        const int aslot = ((kc + dyi) & 1) * Cfg::A_SLOT;               // group g = 3 kc + dy: g & 1 = (kc + dy) & 1
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
          const bool issueA = dxi == 0 && !(LAST && dyi == 2);
          const bool issueB = !(LAST && dyi * 3 + dxi + 2 >= 9);
          if (issueA) stageA(((kc + dyi + 1) & 1), dyi == 2 ? kc + 1 : kc, dyi == 2 ? 0 : dyi + 1);
          if (issueB) {
            const int dx2 = (dxi + 2) % 3;
            const int d2 = dyi + (dxi + 2) / 3;                         // dy of step s + 2, 3 = first group of the next slab
            stageB(dx2, d2 == 3 ? kc + 1 : kc, (d2 == 3 ? 0 : d2) * 3 + dx2);
          }
          uint4 wf[NT], xf[MT];
          const char* pb = smem + bBase + dxi * Cfg::B_SLOT;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) wf[nt] = *(const uint4*)(pb + nt * 16 * F2_ROWB);
          const char* pa = smem + aBase[dxi] + aslot;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) xf[mt] = *(const uint4*)(pa + mt * 16 * F2_ROWB);
          if constexpr (GEO == 1) {
            if (dxi != 1) {
              const uint32_t edge = dxi == 0 ? edgeL : edgeR, keep = dxi == 0 ? keepL : keepR;
#pragma unroll
              for (int mt = 0; mt < MT; ++mt)
                if (edge & (1u << mt)) { xf[mt].x &= keep; xf[mt].y &= keep; xf[mt].z &= keep; xf[mt].w &= keep; }
            }
          }
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(f2_bf16x8_t, wf[nt]),
                                                                    __builtin_bit_cast(f2_bf16x8_t, xf[mt]), acc[nt][mt], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
          // the next step's weights (and, after dx = 2, the next group's pixels) were issued a step or more ago: only what
          // this step issued may still fly
          if (issueA && issueB) f2_wait_and_barrier<AD + BD>();
          else if (issueB) f2_wait_and_barrier<BD>();
          else f2_wait_and_barrier<0>();
        }
      }
    };
    for (int kc = 0; kc < kcs - 1; ++kc) slab(kc, std::false_type{});
    slab(kcs - 1, std::true_type{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    F2_STAMP(4);          // K loop

    // ---- epilogue (as conv3x3_flat.hip): registers -> LDS tile -> 16-byte NHWC stores + BN partial sums ------
    char* tile = smem;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = wm * WROWS + mt * 16 + r16;
        const int col = wn * 64 + nt * 16 + kq * 4;
        uint2 v;
        v.x = pack2_bf16(acc[nt][mt][0], acc[nt][mt][1]);
        v.y = pack2_bf16(acc[nt][mt][2], acc[nt][mt][3]);
        *(uint2*)(tile + row * Cfg::PITCH + col * 2) = v;
      }
    __syncthreads();

    constexpr int CPR = BN * 2 / 16;
    constexpr int ITER = F2_BM * CPR / F2_THREADS;
    constexpr int RSTEP = F2_THREADS / CPR;
    const int cc = tid % CPR;
    const long long col_off = n0 + cc * CH;
    float s1[CH], s2[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    float bsc[BS ? CH : 1], bsh[BS ? CH : 1];
    if constexpr (BS) {
#pragma unroll
      for (int j = 0; j < CH; ++j) { bsc[j] = a.bscale[col_off + j]; bsh[j] = a.bshift[col_off + j]; }
    }
    constexpr int HALF = ITER / 2;
#pragma unroll
    for (int i0 = 0; i0 < ITER; i0 += HALF) {
      uint4 yv[BS ? HALF : 1];
      if constexpr (BS) {
#pragma unroll
        for (int i = 0; i < HALF; ++i) {
          const long long ro = rowOut[(i0 + i) * RSTEP + tid / CPR];
          yv[i] = ro >= 0 ? *(const uint4*)(a.by + (ro + col_off) * 2) : make_uint4(0u, 0u, 0u, 0u);
        }
      }
#pragma unroll
      for (int i = 0; i < HALF; ++i) {
        const int row = (i0 + i) * RSTEP + tid / CPR;
        const long long ro = rowOut[row];
        if (ro >= 0) {
          const uint4 u = *(const uint4*)(tile + row * Cfg::PITCH + cc * 16);
          float f[CH];
          Chunk<bf16_t>::unpack(u, f);
          if constexpr (BS) {
            float yy[CH];
            Chunk<bf16_t>::unpack(yv[i], yy);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
              const float m = fmaf(yy[j], bsc[j], bsh[j]) > 0.f ? f[j] : 0.f;
              s1[j] += m; s2[j] = fmaf(m, yy[j], s2[j]);
            }
          } else {
#pragma unroll
            for (int j = 0; j < CH; ++j) { s1[j] += f[j]; s2[j] = fmaf(f[j], f[j], s2[j]); }
          }
          *(uint4*)(a.y + (ro + col_off) * 2) = u;
        }
      }
    }
    if (a.stats && a.carry) {
#pragma unroll
      for (int j = 0; j < CH; ++j) { cs1[j] += s1[j]; cs2[j] += s2[j]; }
    } else if (a.stats) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
#pragma unroll
        for (int o = CPR; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o, 64); s2[j] += __shfl_xor(s2[j], o, 64); }
      }
      if (lane < CPR) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          sstat[(wave * BN + lane * CH + j) * 2 + 0] = s1[j];
          sstat[(wave * BN + lane * CH + j) * 2 + 1] = s2[j];
        }
      }
      __syncthreads();
      if (tid < BN) {
        float v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { v1 += sstat[(w * BN + tid) * 2 + 0]; v2 += sstat[(w * BN + tid) * 2 + 1]; }
        a.stats[((long long)mtile * 2 + 0) * a.N + n0 + tid] = v1;
        a.stats[((long long)mtile * 2 + 1) * a.N + n0 + tid] = v2;
      }
    }
    F2_STAMP(5);          // epilogue: transpose through LDS, stores, statistics
    __syncthreads();      // the next tile's LDS-DMA rewrites the ring the epilogue tile aliases
    F2_STAMP(6);          // the other waves' epilogues
  }
  if (a.stats && a.carry) {
    constexpr int CPR = BN * 2 / 16;
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
      for (int w = 0; w < 4; ++w) { v1 += sstat[(w * BN + tid) * 2 + 0]; v2 += sstat[(w * BN + tid) * 2 + 1]; }
      a.stats[((long long)blockIdx.x * 2 + 0) * a.N + tid] = v1;       // one N tile: n0 = 0
      a.stats[((long long)blockIdx.x * 2 + 1) * a.N + tid] = v2;
    }
  }
#ifdef INSAR_STAMPS
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&g_flat2_stamps[(blockIdx.x & 1023) * 8 + k], stamp_acc[k]);
  }
#endif
}

// ---- host side -----------------------------------------------------------------------------------
int insar_flat2_persistent_grid(int bn) { return F2_WGS(bn) * (insar_num_cus() & ~7); }

template <int BN, bool BS, int GEO>
static int launch_flat2(FlatArgs& a, hipStream_t s) {
  using Cfg = Flat2Cfg<BN, GEO>;
  static std::atomic<uint64_t> attr_mask{0};
  {
    hipError_t e = insar_set_lds_once(attr_mask, (const void*)conv3x3_flat2_kernel<BN, BS, GEO>, Cfg::LDS_BYTES);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_conv3x3_flat: hipFuncSetAttribute(%d bytes LDS): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
  }
  a.kc_count = a.K / 32;
  a.num_ntiles = a.N / BN;
  long long grid = (long long)a.num_mtiles * a.num_ntiles;
  a.total_tiles = (int)grid;
  a.carry = 0;
  if (a.persist) {                                   // two work-groups per CU, each walking its tiles
    const int slots = insar_flat2_persistent_grid(BN);
    if (slots >= 16 && grid > slots) { grid = slots; a.carry = a.num_ntiles == 1 ? 1 : 0; }
  }
  hipLaunchKernelGGL((conv3x3_flat2_kernel<BN, BS, GEO>), dim3((unsigned)grid), dim3(F2_THREADS), Cfg::LDS_BYTES, s, a);
  INSAR_CHECK_LAUNCH("insar_conv3x3_flat");
  return INSAR_OK;
}

bool insar_flat2_rows_geometry(const InsarAct& x) {
  return x.dtype == INSAR_BF16 && x.W >= 16 && x.W <= F2_BM && (x.W & (x.W - 1)) == 0 && x.H % (F2_BM / x.W) == 0 &&
         (long long)x.B * (x.H + 2) * (x.W + 2) < 0x7fffffffLL;
}

int insar_flat2_launch(FlatArgs& a, int bn, bool bstat, bool rows, hipStream_t s) {
  if (a.K % 32) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_flat: K=%d must be a multiple of 32", a.K);
  // 32-bit lane offsets: relative to the tile for the activations (any buffer size), to the N tile for the weights; one tile
  // spans 256 pixels + an image row either side
  if ((long long)(F2_BM + 2 * (a.W + 2) + 2) * a.Cx * 2 >= (1LL << 30) || (long long)bn * a.K * 2 >= (1LL << 30))
    INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_flat: rows too long for the two-work-group kernel's 32-bit tile offsets");
  if (rows) {
    if (bn == 128) return bstat ? launch_flat2<128, true, 1>(a, s) : launch_flat2<128, false, 1>(a, s);
    return bstat ? launch_flat2<64, true, 1>(a, s) : launch_flat2<64, false, 1>(a, s);
  }
  if (bn == 128) return bstat ? launch_flat2<128, true, 0>(a, s) : launch_flat2<128, false, 0>(a, s);
  return bstat ? launch_flat2<64, true, 0>(a, s) : launch_flat2<64, false, 0>(a, s);
}
