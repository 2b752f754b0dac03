// Implicit-GEMM convolution on the gfx950 matrix cores.
//
//   Y[m, n] = sum_{tap} sum_{k} X[in_pix(m, tap), k] * Wt[tap][n][k]
//
// replaces nn.Conv2d(3x3, pad 1) forward (Unet-ChannalAttention.py:81,84), its input-gradient
// (same kernel, flipped taps + transposed weights), nn.ConvTranspose2d(k2,s2) forward
// (:112-121; one tap, N = 4*Cout, scatter epilogue) and its input-gradient (4 taps, stride 2).
//
// Design (CDNA4):
//  * activations are NHWC with a zero halo, so every tap of every output pixel is an in-bounds,
//    channel-contiguous 128-byte row segment: A rows are gathered straight into LDS by LDS-DMA
//    (global_load_lds_dwordx4, per-lane source address = pixel row + tap offset).
//  * 128(M pixels) x BN(out channels) x 128-byte K slabs, double-buffered in LDS; 4 waves (2x2),
//    each owning 64 x BN/2 of the tile as 16x16 MFMA tiles. LDS rows are XOR-swizzled on the
//    DMA *source* address (chunk ^= row & 7) so the ds_read_b128 fragment reads are conflict-free.
//  * MFMA A operand = weights, B operand = activations, so each lane ends up holding 4 consecutive
//    output channels of one pixel; the tile is transposed through LDS and stored as full 16-byte
//    channel chunks (NHWC rows), and the per-channel sum / sum-of-squares partials that
//    BatchNorm needs (:82,85) are taken from the *stored* values on the way out (one slab row per
//    M tile, no atomics).
//  * bf16: v_mfma_f32_16x16x32_bf16 (fp32 accumulate); fp32: v_mfma_f32_16x16x4_f32 (exact fp32).
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define IG_ROWB 128  // bytes of K per LDS row

struct IgemmArgs {
  const char* x; const char* w; char* y; const float* bias; float* stats;
  long long M;
  int Ho, Wo, stride;
  int Hi, Wi, Cx, cx_off, K;
  int Hy, Wy, Cy, cy_off;
  int N, ntaps, mode, Cout;
  int kc_per_tap;
  int num_mtiles, num_ntiles;
  int tapoff[12];  // element offset of each tap relative to the tap-(0,0) pixel
  signed char tdy[12], tdx[12];   // the taps themselves (OOB variant: per-row bounds test)
  int out_stride, out_oy, out_ox;   // mode 0: row (ho, wo) is stored at output pixel (ho*out_stride + out_oy, wo*out_stride + out_ox)
  const char* add; // nullable: tensor with y's layout added to the result in the epilogue (residual / gradient sum)
  const char* gate; // nullable: tensor with y's layout; the result is stored as zero where it is <= 0 (ReLU mask of a residual block's input)
  const char* by; const float* bscale; const float* bshift;   // BatchNorm-backward sums in the stats slab (InsarBstat)
};

__device__ __forceinline__ void lds_dma16(const char* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ __forceinline__ static void run(const uint4& wa, const uint4& xb, f32x4_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xb), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  __device__ __forceinline__ static void run(const uint4& wa, const uint4& xb, f32x4_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.x), __uint_as_float(xb.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.y), __uint_as_float(xb.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.z), __uint_as_float(xb.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.w), __uint_as_float(xb.w), acc, 0, 0, 0);
  }
};

template <typename T, int BM, int BN, int NSTAGE>
struct IgemmCfg {
  static constexpr int ES = sizeof(T);
  static constexpr int BKe = IG_ROWB / ES;
  static constexpr int WAVES_M = BM / 64;                  // each wave owns 64 x BN/2 of the tile
  static constexpr int NWAVES = WAVES_M * 2;
  static constexpr int THREADS = NWAVES * 64;              // == 2 * BM
  static constexpr int A_STAGE = BM * IG_ROWB;
  static constexpr int B_STAGE = BN * IG_ROWB;
  static constexpr int STAGE = A_STAGE + B_STAGE;
  static constexpr int A_DMA = 4;                          // BM*8 chunks / THREADS
  static constexpr int B_DMA = BN * 8 / THREADS;
  static constexpr int PITCH = BN * ES + 16;               // epilogue tile row pitch (bytes)
  static constexpr int TILE = BM * PITCH;
  static constexpr int MAIN = (NSTAGE * STAGE > TILE) ? NSTAGE * STAGE : TILE;
  static constexpr int ROWINFO = BM * 8 * 2 + BM * 4;      // rowIn[BM], rowOut[BM] (int64), rowHW[BM] (int32)
  static constexpr int STATB = NWAVES * BN * 2 * 4;        // per-wave channel partials
  static constexpr int TAPB = 64 + 32 + 128;                // tap offsets (12 ints), tap dy/dx (12 + 12 bytes), OOB: live flags / live-tap list (2 x 16 ints)
  static constexpr int LDS_BYTES = MAIN + ROWINFO + STATB + TAPB;
};

// Diagnostic build only (-DINSAR_STAMPS, tools/stamp_igemm.py): s_memtime stamps of the phases of a work-group
#ifdef INSAR_STAMPS
__device__ unsigned long long g_igemm_stamps[1024 * 8];
#define IG_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[k] += now_ - stamp_prev; stamp_prev = now_; } while (0)
extern "C" int insar_debug_igemm_stamps(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_igemm_stamps), sizeof(g_igemm_stamps)) != hipSuccess) return -1;
  if (reset) { static unsigned long long z[1024 * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_igemm_stamps), z, sizeof(z)) != hipSuccess) return -2; }
  return 0;
}
#else
#define IG_STAMP(k)
#endif

// drain the LDS-DMA queue down to N outstanding per wave, then meet the other waves
template <int N>
__device__ __forceinline__ void dma_wait_and_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");   // lgkmcnt: see common.h, dma_drain_and_barrier
  __builtin_amdgcn_s_barrier();
}

// OOB: taps may leave the padded input (dilated convolutions, DeepLabV3's ASPP / layer3-4): such (row, tap) pairs
// are redirected to pixel 0 of the buffer — the top-left halo pixel, all zeros in every channel — by a per-row bounds
// test on the LDS-DMA source address; rows beyond M read zeros the same way. The U-Net path never needs it.
// PP (256-row tiles, bf16): the K loop as a ping-pong of the two wave groups of the work-group (waves 0-3 / 4-7 = the two
// waves of every SIMD). A K tile is two phases (one half of the wave's 64 pixels each, all of its channels, both K halves);
// a phase = [load part: the phase's LDS fragment reads + part of the LDS-DMA pieces of the NEXT K tile] -> s_barrier ->
// [compute part: the phase's MFMAs under s_setprio 1] -> s_barrier, and group 1 runs one barrier behind group 0, so that
// on every SIMD one wave's MFMA cluster runs beside its partner's LDS reads and DMA issue, instead of both waves issuing
// their DMA burst together and then contending for the matrix pipe (the role alternation of the guide's 8-phase schedule,
// cdna_hip_programming.md §5; four phases of 16 MFMAs measured the same as two of 32, so the variant with fewer barriers
// is kept). Same accumulation order as the plain loop: results are bitwise equal (test_pingpong_k_loop_...).
// BS: BatchNorm-backward sums of the consumer unit in the statistics slab (InsarBstat) — instantiations of their own, so that
// the plain launches keep the code (and the registers) they had without it.
template <typename T, int BM, int BN, int NSTAGE, bool OOB = false, int PP = 0, bool BS = false>
__global__ __launch_bounds__(2 * BM, (BN >= 256 ? 1 : 2)) void igemm_kernel(IgemmArgs a) {
  using Cfg = IgemmCfg<T, BM, BN, NSTAGE>;
  constexpr int ES = Cfg::ES, BKe = Cfg::BKe, CH = Chunk<T>::N;
  constexpr int NT = BN / 32, MT = 4;
  constexpr int THREADS = Cfg::THREADS, NWAVES = Cfg::NWAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  long long* rowIn = (long long*)(smem + Cfg::MAIN);
  long long* rowOut = rowIn + BM;
  int* rowHW = (int*)(rowOut + BM);
  float* sstat = (float*)(smem + Cfg::MAIN + Cfg::ROWINFO);
  int* stap = (int*)(smem + Cfg::MAIN + Cfg::ROWINFO + Cfg::STATB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef INSAR_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long stamp_t0 = stamp_prev, stamp_r0 = __builtin_amdgcn_s_memrealtime();   // [6] / [7]: shader clock vs 100 MHz
#endif
  // XCD-aware tile order: blocks that share an XCD (blockIdx % 8) get consecutive tiles, so the
  // A rows / halo rows shared by neighbouring tiles hit in that XCD's L2 (bijective remap).
  int t;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int mtile = t / a.num_ntiles, ntile = t - mtile * a.num_ntiles;
  const long long m0 = (long long)mtile * BM;
  const int n0 = ntile * BN;

  int* slive = stap + 24;            // OOB: [0,12) live flag per tap, [12] number of live taps, [16, 28) the live taps in order
  if (tid == THREADS - 1) {
#pragma unroll
    for (int i = 0; i < 12; ++i) stap[i] = a.tapoff[i];
    signed char* sd = (signed char*)(stap + 16);
#pragma unroll
    for (int i = 0; i < 12; ++i) { sd[i] = a.tdy[i]; sd[16 + i] = a.tdx[i]; }
    if constexpr (OOB) {
#pragma unroll
      for (int i = 0; i < 16; ++i) slive[i] = 0;
    }
  }
  if (tid < BM) {
    const long long m = m0 + tid;
    const bool valid = m < a.M;
    const long long mm = valid ? m : 0;
    const long long hw = (long long)a.Ho * a.Wo;
    const int n = (int)(mm / hw);
    const int rem = (int)(mm - (long long)n * hw);
    const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
    rowIn[tid] = (((long long)n * (a.Hi + 2) + ho * a.stride + 1) * (a.Wi + 2) + wo * a.stride + 1) * a.Cx + a.cx_off;
    const int so = a.mode == 1 ? 2 : a.out_stride;
    const long long ro = (((long long)n * (a.Hy + 2) + ho * so + a.out_oy + 1) * (a.Wy + 2) + wo * so + a.out_ox + 1) * a.Cy + a.cy_off;
    rowOut[tid] = valid ? ro : -1;
    // input coordinates of tap (0,0), biased by +1 (halo) and packed; rows beyond M get a row no tap can reach
    rowHW[tid] = valid ? (((ho * a.stride + 1) << 16) | (wo * a.stride + 1)) : (0x4000 << 16);
  }
  __syncthreads();
  if constexpr (OOB) {
    // Taps that no row of this tile can reach inside the image (dilation 12 / 24 on a 32 x 32 map: a tile of 8 image rows
    // sees the +-24 rows from one quarter of the map only) are dropped from the tile's K loop instead of being multiplied
    // against the zero pixel: every row marks the taps it reaches (the halo counts as outside: it is zero), one thread
    // compacts the list. Adding exact zeros is what is skipped: the sums are unchanged.
    if (tid < BM) {
      const int hw = rowHW[tid];
      const int h0 = hw >> 16, w0 = hw & 0xffff;
      if (h0 < 0x4000) {
        for (int t = 0; t < a.ntaps; ++t) {
          const int hh = h0 + a.tdy[t], ww = w0 + a.tdx[t];
          if (hh >= 1 && hh <= a.Hi && ww >= 1 && ww <= a.Wi) slive[t] = 1;
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      int n = 0;
      for (int t = 0; t < a.ntaps; ++t)
        if (slive[t]) slive[16 + n++] = t;
      slive[12] = n;
    }
    __syncthreads();
  }
  IG_STAMP(0);          // row tables

  // per-thread staging geometry: chunk q = i*THREADS + tid -> LDS row q>>3, lane-linear position q&7
  constexpr int RPI = THREADS / 8;                  // rows covered by one DMA instruction of the block
  const int srow = tid >> 3;                        // + RPI*i
  const int schunk = ((tid & 7) ^ (srow & 7)) * 16; // swizzled source chunk (bytes)
  // PP: the LDS image is ordered by (half of the wave's sub-tile, wave, row), so that a phase reads ONE 128-row half of
  // A or B: LDS row L = h*128 + wm*32 + i holds tile pixel wm*64 + h*32 + i; B row L = c*128 + wn*64 + j holds channel
  // wn*128 + c*64 + j. The four block-wide DMA instructions of an operand are then its half 0 (two) and half 1 (two).
  auto a_tile_row = [&](int L) { return PP ? ((L & 127) >> 5) * 64 + (L >> 7) * 32 + (L & 31) : L; };
  auto b_tile_row = [&](int L) { return L; };       // B is read whole in the first phase of a K tile: natural row order
  const char* a_ptr[Cfg::A_DMA];
  int a_hw[Cfg::A_DMA];
#pragma unroll
  for (int i = 0; i < Cfg::A_DMA; ++i) {
    a_ptr[i] = a.x + rowIn[a_tile_row(srow + RPI * i)] * ES + schunk;
    a_hw[i] = OOB ? rowHW[a_tile_row(srow + RPI * i)] : 0;
  }
  const char* a_zero = a.x + (long long)a.cx_off * ES + schunk;      // pixel 0 = zero halo
  const char* b_ptr = a.w + ((long long)(n0 + (PP ? 0 : srow)) * a.K) * ES + schunk;     // PP: per-instruction rows below
  long long b_row_off[Cfg::B_DMA];
#pragma unroll
  for (int i = 0; i < Cfg::B_DMA; ++i) b_row_off[i] = (long long)(PP ? b_tile_row(srow + RPI * i) : RPI * i) * a.K * ES;
  const long long b_tap_bytes = (long long)a.N * a.K * ES;
  const int nk = (OOB ? slive[12] : a.ntaps) * a.kc_per_tap;
  const uint32_t lds0 = lds_offset_of(smem);

  auto stage = [&](int buf, int ks) {
    const int ti = ks / a.kc_per_tap;
    const int kc = ks - ti * a.kc_per_tap;
    const int tap = OOB ? slive[16 + ti] : ti;
    const long long xoff = ((long long)stap[tap] + (long long)kc * BKe) * ES;
    const uint32_t la = lds0 + buf * Cfg::STAGE + wave * 1024;
    if constexpr (OOB) {
      const signed char* sd = (const signed char*)(stap + 16);
      const int tdy = sd[tap], tdx = sd[16 + tap];
      const long long koff = (long long)kc * BKe * ES;
#pragma unroll
      for (int i = 0; i < Cfg::A_DMA; ++i) {
        const int hh = (a_hw[i] >> 16) + tdy, ww = (a_hw[i] & 0xffff) + tdx;      // padded coordinates of the tap
        const bool in = (unsigned)hh <= (unsigned)(a.Hi + 1) && (unsigned)ww <= (unsigned)(a.Wi + 1);
        lds_dma16_untracked(in ? a_ptr[i] + xoff : a_zero + koff, la + i * (THREADS * 16));
      }
    } else {
#pragma unroll
      for (int i = 0; i < Cfg::A_DMA; ++i) lds_dma16_untracked(a_ptr[i] + xoff, la + i * (THREADS * 16));
    }
    const char* wb = b_ptr + tap * b_tap_bytes + (long long)kc * BKe * ES;
    const uint32_t lb = la + Cfg::A_STAGE;
#pragma unroll
    for (int i = 0; i < Cfg::B_DMA; ++i) lds_dma16_untracked(wb + b_row_off[i], lb + i * (THREADS * 16));
  };
  // one DMA piece of a slab (PP schedule): q < A_DMA -> A instruction q, else B instruction q - A_DMA; xoff / wb are the
  // slab's tap and K-chunk offsets, computed once per K tile
  auto stage_piece = [&](int buf, long long xoff, const char* wb, int q) {
    const uint32_t la = lds0 + buf * Cfg::STAGE + wave * 1024;
    if (q < Cfg::A_DMA) lds_dma16_untracked(a_ptr[q] + xoff, la + q * (THREADS * 16));
    else lds_dma16_untracked(wb + b_row_off[q - Cfg::A_DMA], la + Cfg::A_STAGE + (q - Cfg::A_DMA) * (THREADS * 16));
  };

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int wm = wave % Cfg::WAVES_M, wn = wave / Cfg::WAVES_M;
  const int r16 = lane & 15, kq = lane >> 4;
  const int a_frag = (wm * 64 + r16) * IG_ROWB;            // + mt*16*128
  const int b_frag = (wn * (BN / 2) + r16) * IG_ROWB;      // + nt*16*128
  const int sw = r16 & 7;

  if constexpr (PP) {
    static_assert(BM == 256 && NSTAGE == 2 && sizeof(T) == 2 && !OOB, "ping-pong loop: 256-row bf16 tiles, two LDS slabs");
    // ---- K loop, ping-pong schedule (see the kernel's header comment) ---------------------------------------
    // staging order of a K tile's 8 pieces (A0 A0' | B0 B0' B1 | B1' A1 A1'): what phase 1 reads first; phase 4 stages
    // nothing, so every piece is at least one full phase old when its wait comes
    const int grp = wave >> 2;                                   // 0: waves 0-3, 1: waves 4-7 (SIMD partners)
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    IG_STAMP(1);        // first slab landed
    if (grp == 1) __builtin_amdgcn_s_barrier();                   // group 1 runs one barrier behind
    {
    // two phases per K tile: pixel half 0, then pixel half 1 of the wave's 64 pixels, each against all of the wave's BN/2
    // channels (2 * NT MFMAs per phase and K half). Phase A reads x0 + every w fragment and stages A0 A0' + all of B of
    // the next tile; phase B reads x1 and stages A1 A1'. Waits: phase B retires the pieces of phase A (a phase old, its own
    // two stay in flight); phase A retires the two A1 pieces issued at the end of the tile before (read in phase B).
    uint4 xf[2][2], wf[NT][2];
    const int xrow0 = (wm * 32 + r16) * IG_ROWB;
    const int pc0 = ((kq) ^ sw) * 16, pc1 = ((kq + 4) ^ sw) * 16;
    for (int ks = 0; ks < nk; ++ks) {
      const int buf = ks & 1;
      const bool more = ks + 1 < nk;
      const char* sA = smem + buf * Cfg::STAGE;
      const char* sB = sA + Cfg::A_STAGE;
      long long nxoff = 0;
      const char* nwb = b_ptr;
      if (more) {
        const int tap = (ks + 1) / a.kc_per_tap, kc = (ks + 1) - tap * a.kc_per_tap;
        nxoff = ((long long)stap[tap] + (long long)kc * BKe) * ES;
        nwb = b_ptr + tap * b_tap_bytes + (long long)kc * BKe * ES;
      }
#pragma unroll
      for (int pa = 0; pa < 2; ++pa) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const char* p = sA + pa * (128 * IG_ROWB) + xrow0 + m * 16 * IG_ROWB;
          xf[m][0] = *(const uint4*)(p + pc0); xf[m][1] = *(const uint4*)(p + pc1);
        }
        if (pa == 0) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const char* p = sB + b_frag + n * 16 * IG_ROWB;
            wf[n][0] = *(const uint4*)(p + pc0); wf[n][1] = *(const uint4*)(p + pc1);
          }
        }
        if (more) {
          if (pa == 0) {
            stage_piece(buf ^ 1, nxoff, nwb, 0); stage_piece(buf ^ 1, nxoff, nwb, 1);
#pragma unroll
            for (int i = 0; i < Cfg::B_DMA; ++i) stage_piece(buf ^ 1, nxoff, nwb, Cfg::A_DMA + i);
          } else { stage_piece(buf ^ 1, nxoff, nwb, 2); stage_piece(buf ^ 1, nxoff, nwb, 3); }
        }
        if (!more) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (pa == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 + Cfg::B_DMA) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
#pragma unroll
          for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < 2; ++m) Mma<T>::run(wf[n][kh], xf[m][kh], acc[n][pa * 2 + m]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
    }
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                   // group 0 meets group 1's last barrier
  } else {
  // ---- K loop: NSTAGE-deep LDS ring, NSTAGE-1 slabs in flight ------------------------------------
  constexpr int PER = Cfg::A_DMA + Cfg::B_DMA;    // DMA instructions per wave per slab
#pragma unroll
  for (int p = 0; p < NSTAGE - 1; ++p)
    if (p < nk) stage(p, p);
  if (nk >= NSTAGE - 1) dma_wait_and_barrier<(NSTAGE - 2) * PER>();
  else dma_wait_and_barrier<0>();
  IG_STAMP(1);          // first slab landed
  int buf = 0, pbuf = NSTAGE - 1;                   // pbuf = ring slot the next prefetch goes to
  for (int ks = 0; ks < nk; ++ks) {
    const bool more = ks + NSTAGE - 1 < nk;
    if (more) stage(pbuf, ks + NSTAGE - 1);
    const char* sA = smem + buf * Cfg::STAGE;
    const char* sB = sA + Cfg::A_STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int pc = ((kq + 4 * s) ^ sw) * 16;
      uint4 xf[MT], wf[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[mt] = *(const uint4*)(sA + a_frag + mt * 16 * IG_ROWB + pc);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wf[nt] = *(const uint4*)(sB + b_frag + nt * 16 * IG_ROWB + pc);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) Mma<T>::run(wf[nt], xf[mt], acc[nt][mt]);
    }
    // slab ks+1 must have landed (and every wave must be done reading slab ks) before the next step;
    // the NSTAGE-2 younger slabs stay in flight across the barrier.
    if (more) dma_wait_and_barrier<(NSTAGE - 2) * PER>();
    else dma_wait_and_barrier<0>();
    buf = (buf + 1 == NSTAGE) ? 0 : buf + 1;
    pbuf = (pbuf + 1 == NSTAGE) ? 0 : pbuf + 1;
  }
  }
  __syncthreads();
  IG_STAMP(2);          // K loop

  // ---- epilogue: registers -> LDS tile [pixel][channel] -> 16-byte NHWC stores (+stats) ----------
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

  IG_STAMP(3);          // accumulators -> LDS tile
  constexpr int CPR = BN * ES / 16;               // 16-byte chunks per tile row
  constexpr int ITER = BM * CPR / THREADS;
  constexpr int RSTEP = THREADS / CPR;
  const int cc = tid % CPR;
  const int ncol = n0 + cc * CH;
  float bias[CH];
  int bias_base = ncol;
  long long col_off = ncol;                        // mode 0: channel offset inside the pixel
  if (a.mode == 1) {
    const int q4 = ncol / a.Cout;
    const int co = ncol - q4 * a.Cout;
    bias_base = co;
    col_off = ((long long)(q4 >> 1) * (a.Wy + 2) + (q4 & 1)) * a.Cy + co;
  }
#pragma unroll
  for (int j = 0; j < CH; ++j) bias[j] = a.bias ? a.bias[bias_base + j] : 0.f;
  float s1[CH], s2[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  // bstat: the consumer unit's y at the output's positions, all of this thread's chunks requested before the first is used
  uint4 yv[BS ? ITER : 1];
  float bsc[BS ? CH : 1], bsh[BS ? CH : 1];
  if constexpr (BS) {
#pragma unroll
    for (int j = 0; j < CH; ++j) { bsc[j] = a.bscale[ncol + j]; bsh[j] = a.bshift[ncol + j]; }
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const long long ro = rowOut[i * RSTEP + tid / CPR];
      yv[i] = ro >= 0 ? *(const uint4*)(a.by + (ro + col_off) * ES) : make_uint4(0u, 0u, 0u, 0u);
    }
  }
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int row = i * RSTEP + tid / CPR;
    const long long ro = rowOut[row];
    if (ro >= 0) {
      float f[CH];
      Chunk<T>::unpack(*(const uint4*)(tile + row * Cfg::PITCH + cc * 16), f);
      if constexpr (!BS) {
#pragma unroll
        for (int j = 0; j < CH; ++j) { s1[j] += f[j]; s2[j] = fmaf(f[j], f[j], s2[j]); }
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) f[j] += bias[j];
      if (a.add) {
        float g[CH];
        Chunk<T>::unpack(*(const uint4*)(a.add + (ro + col_off) * ES), g);
#pragma unroll
        for (int j = 0; j < CH; ++j) f[j] += g[j];
      }
      if (a.gate) {
        float q[CH];
        Chunk<T>::unpack(*(const uint4*)(a.gate + (ro + col_off) * ES), q);
#pragma unroll
        for (int j = 0; j < CH; ++j) f[j] = q[j] > 0.f ? f[j] : 0.f;
      }
      const uint4 pk = Chunk<T>::pack(f);
      *(uint4*)(a.y + (ro + col_off) * ES) = pk;
      if constexpr (BS) {
        // BatchNorm-backward sums of the consumer over the values AS STORED (after add / gate, rounded to T: what a reduce
        // pass over the output would read; without add / gate that is the tile value itself, bit for bit)
        float yy[CH], fr[CH];
        Chunk<T>::unpack(yv[i], yy);
        Chunk<T>::unpack(pk, fr);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const float m = fmaf(yy[j], bsc[j], bsh[j]) > 0.f ? fr[j] : 0.f;
          s1[j] += m; s2[j] = fmaf(m, yy[j], s2[j]);
        }
      }
    }
  }
  IG_STAMP(4);          // stores (+ sums)
  if (a.stats) {
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
    LDS_DRAIN();               // see common.h: keep the store's source registers intact until it has drained
#pragma unroll
    for (int j = 0; j < CH; ++j) { LDS_KEEP(s1[j]); LDS_KEEP(s2[j]); }
    __syncthreads();
    if (tid < BN) {
      float v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVES; ++w) { v1 += sstat[(w * BN + tid) * 2 + 0]; v2 += sstat[(w * BN + tid) * 2 + 1]; }
      a.stats[((long long)mtile * 2 + 0) * a.N + n0 + tid] = v1;
      a.stats[((long long)mtile * 2 + 1) * a.N + n0 + tid] = v2;
    }
  }
#ifdef INSAR_STAMPS
  IG_STAMP(5);          // statistics fold
  stamp_acc[6] = stamp_prev - stamp_t0; stamp_acc[7] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&g_igemm_stamps[(blockIdx.x & 1023) * 8 + k], stamp_acc[k]);
  }
#endif
}

// Tile selection: the 256-row / 8-wave / 3-slab-ring variant needs enough tiles to fill 256 CUs: 256 x 128
// tiles where that gives >= 256 of them, else 256 x 64 tiles if THOSE fill the chip (the 16x16 level with
// 1024 output channels: 1 % of the step faster than 128 x 64 tiles); the remaining small grids keep the
// 128-row / 2-slab variant at 2 blocks per CU.
static inline int igemm_bm_for(long long M, int N) {
  const int bn = (N % 128) == 0 ? 128 : 64;
  const long long tiles256 = ((M + 255) / 256) * (N / bn);
  if (tiles256 >= 256) return 256;
  // 256 x 64 tiles when they fill at least HALF the chip: still the 8-wave / 3-slab kernel (128 eight-wave
  // work-groups beat 256 four-wave ones: the 16x16-level dgrad with N = 512 121 -> 96 us)
  if (((M + 255) / 256) * (N / 64) >= 128) return 256;
  return 128;
}
// Output-channel tile: 128 wide where N allows, except on grids so small that 128x128 tiles would leave one
// 4-wave work-group per CU (the 16x16 level): 128x64 tiles double the work-groups (two per CU, two waves
// per SIMD to hide each other's LDS and barrier latency).
static inline int igemm_bn_for(long long M, int N) {
  if (N % 128) return 64;
  if (igemm_bm_for(M, N) == 256) {
    const int min_tiles = insar_knob(KNOB_IGEMM_WIDE_MIN) > 0 ? insar_knob(KNOB_IGEMM_WIDE_MIN) : 256;
    return ((M + 255) / 256) * (N / 128) >= min_tiles ? 128 : 64;
  }
  const long long tiles = ((M + 127) / 128) * (N / 128);
  return tiles <= 256 ? 64 : 128;
}
// 256 x 256 tiles (bf16 only: the fp32 epilogue tile would not fit the LDS): wave tile 128 x 64, two slabs. A third
// less operand traffic through the L2 -> LDS path per FLOP than 256 x 128, which is what bounds these kernels
// (same-box A/B: the launches concerned 87.5 -> 80.7 us on average, the step 9.12 -> 8.98 ms); only where they still
// give every CU a work-group.
static inline bool igemm_xwide(long long M, int N, int dtype) {
  const int min_tiles = insar_knob(KNOB_IGEMM_XWIDE_MIN) > 0 ? insar_knob(KNOB_IGEMM_XWIDE_MIN) : 256;
  return dtype == INSAR_BF16 && igemm_bm_for(M, N) == 256 && (N % 256) == 0 && ((M + 255) / 256) * (N / 256) >= min_tiles;
}
extern "C" int insar_igemm_tile_cols_dt(int64_t M, int32_t N, int32_t dtype) {
  return igemm_xwide(M, N, dtype) ? 256 : igemm_bn_for(M, N);
}
extern "C" int insar_igemm_tile_rows(int64_t M, int32_t N) { return igemm_bm_for(M, N); }
extern "C" int insar_igemm_tile_cols(int64_t M, int32_t N) { return igemm_bn_for(M, N); }
extern "C" int insar_igemm_num_mtiles(int64_t M, int32_t N) {
  const int bm = igemm_bm_for(M, N);
  return (int)((M + bm - 1) / bm);
}

template <typename T, int BM, int BN, int NSTAGE, bool OOB, int PP, bool BS>
static int launch_igemm_bs(IgemmArgs& a, hipStream_t s) {
  using Cfg = IgemmCfg<T, BM, BN, NSTAGE>;
  static std::atomic<uint64_t> attr_mask{0};     // per-device, see common.h
  {
    hipError_t e = insar_set_lds_once(attr_mask, (const void*)igemm_kernel<T, BM, BN, NSTAGE, OOB, PP, BS>, Cfg::LDS_BYTES);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_igemm: hipFuncSetAttribute(%d bytes LDS): %s", Cfg::LDS_BYTES, hipGetErrorString(e));
  }
  a.num_mtiles = (int)((a.M + BM - 1) / BM);
  a.num_ntiles = a.N / BN;
  const int grid = a.num_mtiles * a.num_ntiles;
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, NSTAGE, OOB, PP, BS>), dim3(grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, s, a);
  INSAR_CHECK_LAUNCH("insar_igemm");
  return INSAR_OK;
}
template <typename T, int BM, int BN, int NSTAGE, bool OOB = false, int PP = 0>
static int launch_igemm(IgemmArgs& a, hipStream_t s) {
  return a.by ? launch_igemm_bs<T, BM, BN, NSTAGE, OOB, PP, true>(a, s) : launch_igemm_bs<T, BM, BN, NSTAGE, OOB, PP, false>(a, s);
}

extern "C" int insar_igemm(const InsarIgemm* d, void* stream) {
  if (!d || !d->x.ptr || !d->y.ptr || !d->w) INSAR_FAIL(INSAR_E_ARG, "insar_igemm: null pointer");
  int rc;
  if ((rc = insar_check_act(&d->x, "insar_igemm", "x"))) return rc;
  if ((rc = insar_check_act(&d->y, "insar_igemm", "y"))) return rc;
  if (d->x.dtype != d->y.dtype) INSAR_FAIL(INSAR_E_DTYPE, "insar_igemm: x/y dtype differ");
  const int es = d->x.dtype == INSAR_BF16 ? 2 : 4;
  const int bke = IG_ROWB / es;
  const int K = d->x.c_len;
  if (K % bke) INSAR_FAIL(INSAR_E_SHAPE, "insar_igemm: K=%d must be a multiple of %d", K, bke);
  if (d->N % 64) INSAR_FAIL(INSAR_E_SHAPE, "insar_igemm: N=%d must be a multiple of 64", d->N);
  if (d->ntaps < 1 || d->ntaps > 12) INSAR_FAIL(INSAR_E_SHAPE, "insar_igemm: ntaps=%d", d->ntaps);
  if (d->stride != 1 && d->stride != 2) INSAR_FAIL(INSAR_E_SHAPE, "insar_igemm: stride=%d", d->stride);
  if (d->x.B != d->y.B) INSAR_FAIL(INSAR_E_SHAPE, "insar_igemm: batch differs");
  if (!insar_aligned16(d->w)) INSAR_FAIL(INSAR_E_ALIGN, "insar_igemm: weights not 16-byte aligned");
  int cout = d->N;
  const int os = d->out_stride > 0 ? d->out_stride : 1;
  if (d->mode == 0) {
    if (os == 1 && (d->out_oy || d->out_ox)) INSAR_FAIL(INSAR_E_ARG, "insar_igemm: output offset without an output stride");
    if (d->out_oy < 0 || d->out_ox < 0 || d->out_oy >= os || d->out_ox >= os) INSAR_FAIL(INSAR_E_ARG, "insar_igemm: output offset outside [0, out_stride)");
    const bool fits = os == 1 ? (d->y.H == d->Ho && d->y.W == d->Wo)
                              : ((d->Ho - 1) * os + d->out_oy < d->y.H && (d->Wo - 1) * os + d->out_ox < d->y.W);
    if (!fits || d->y.c_len != d->N) INSAR_FAIL(INSAR_E_SHAPE, "insar_igemm: output slice does not match Ho/Wo/N");
  } else if (d->mode == 1) {
    cout = d->N / 4;
    if (d->y.H != 2 * d->Ho || d->y.W != 2 * d->Wo || d->y.c_len != cout || (cout % 8)) INSAR_FAIL(INSAR_E_SHAPE, "insar_igemm: convT output slice mismatch");
  } else INSAR_FAIL(INSAR_E_ARG, "insar_igemm: mode=%d", d->mode);
  const bool oob = (d->flags & INSAR_IGEMM_OOB_ZERO) != 0;
  if (oob && d->x.H + 2 > 0x3fff) INSAR_FAIL(INSAR_E_SHAPE, "insar_igemm: input too tall for the out-of-bounds variant");
  if (d->add && d->mode != 0) INSAR_FAIL(INSAR_E_ARG, "insar_igemm: `add` needs mode 0");
  if (d->add && !insar_aligned16(d->add)) INSAR_FAIL(INSAR_E_ALIGN, "insar_igemm: add not 16-byte aligned");
  if (d->gate && (d->mode != 0 || (d->stats && !d->bstat.y))) INSAR_FAIL(INSAR_E_ARG, "insar_igemm: `gate` needs mode 0 and no forward statistics (BatchNorm-backward sums are taken over the gated values)");
  if (d->gate && !insar_aligned16(d->gate)) INSAR_FAIL(INSAR_E_ALIGN, "insar_igemm: gate not 16-byte aligned");
  // every tap of every row must stay inside the padded input (unless out-of-bounds taps read zeros)
  for (int t = 0; t < d->ntaps && !oob; ++t) {
    const int ymin = d->dy[t], ymax = (d->Ho - 1) * d->stride + d->dy[t];
    const int xmin = d->dx[t], xmax = (d->Wo - 1) * d->stride + d->dx[t];
    if (ymin < -1 || xmin < -1 || ymax > d->x.H || xmax > d->x.W)
      INSAR_FAIL(INSAR_E_SHAPE, "insar_igemm: tap %d (%d,%d) leaves the padded input", t, d->dy[t], d->dx[t]);
  }
  IgemmArgs a;
  a.x = (const char*)d->x.ptr; a.w = (const char*)d->w; a.y = (char*)d->y.ptr; a.bias = d->bias; a.stats = d->stats;
  a.M = (long long)d->x.B * d->Ho * d->Wo;
  a.Ho = d->Ho; a.Wo = d->Wo; a.stride = d->stride;
  a.Hi = d->x.H; a.Wi = d->x.W; a.Cx = d->x.C; a.cx_off = d->x.c_off; a.K = K;
  a.Hy = d->y.H; a.Wy = d->y.W; a.Cy = d->y.C; a.cy_off = d->y.c_off;
  a.N = d->N; a.ntaps = d->ntaps; a.mode = d->mode; a.Cout = cout;
  a.kc_per_tap = K / bke;
  for (int t = 0; t < 12; ++t) {
    a.tapoff[t] = t < d->ntaps ? (d->dy[t] * (d->x.W + 2) + d->dx[t]) * d->x.C : 0;
    a.tdy[t] = t < d->ntaps ? d->dy[t] : 0; a.tdx[t] = t < d->ntaps ? d->dx[t] : 0;
  }
  a.add = (const char*)d->add; a.gate = (const char*)d->gate;
  a.by = (const char*)d->bstat.y; a.bscale = d->bstat.scale; a.bshift = d->bstat.shift;
  if (a.by) {
    if (d->mode != 0 || os != 1 || !d->stats || !a.bscale || !a.bshift || d->bias)
      INSAR_FAIL(INSAR_E_ARG, "insar_igemm: bstat needs mode 0, a dense output, a stats slab, scale / shift, no bias");
    if (!insar_aligned16(a.by)) INSAR_FAIL(INSAR_E_ALIGN, "insar_igemm: bstat.y not 16-byte aligned");
  }
  a.out_stride = os; a.out_oy = d->mode == 0 ? d->out_oy : 0; a.out_ox = d->mode == 0 ? d->out_ox : 0;
  hipStream_t s = (hipStream_t)stream;
  const bool wide = igemm_bn_for(a.M, d->N) == 128;
  const bool big = igemm_bm_for(a.M, d->N) == 256;
  if (oob) {     // dilated taps (DeepLabV3's layer3 / layer4 / ASPP, 32x32 maps at B = 16): bf16 follows the tile choice of the
                 // in-bounds path (ASPP's input gradient has N = 2048: 256 x 256 tiles; layer4's N = 512: 256 x 128), fp32 keeps two shapes
    if (d->x.dtype == INSAR_BF16) {
      if (igemm_xwide(a.M, d->N, INSAR_BF16)) return launch_igemm<bf16_t, 256, 256, 2, true>(a, s);
      if (big) return wide ? launch_igemm<bf16_t, 256, 128, 3, true>(a, s) : launch_igemm<bf16_t, 256, 64, 3, true>(a, s);
      return launch_igemm<bf16_t, 128, 64, 2, true>(a, s);
    }
    return big ? launch_igemm<float, 256, 64, 3, true>(a, s) : launch_igemm<float, 128, 64, 2, true>(a, s);
  }
  if (d->x.dtype == INSAR_BF16) {
    if (igemm_xwide(a.M, d->N, INSAR_BF16))
      return (d->flags & INSAR_IGEMM_PINGPONG) ? launch_igemm<bf16_t, 256, 256, 2, false, 2>(a, s) : launch_igemm<bf16_t, 256, 256, 2>(a, s);
    // (the ping-pong loop on 256 x 128 tiles was measured too: 512 -> 512 at 32 x 32 75.1 -> 75.5 us, 1024 -> 512 forward
    //  140.8 -> 147.6 us: with half the MFMAs per LDS-DMA piece the load part of a phase outlasts its partner's compute part)
    if (big) return wide ? launch_igemm<bf16_t, 256, 128, 3>(a, s) : launch_igemm<bf16_t, 256, 64, 3>(a, s);
    return wide ? launch_igemm<bf16_t, 128, 128, 2>(a, s) : launch_igemm<bf16_t, 128, 64, 2>(a, s);
  }
  if (big) return wide ? launch_igemm<float, 256, 128, 3>(a, s) : launch_igemm<float, 256, 64, 3>(a, s);
  return wide ? launch_igemm<float, 128, 128, 2>(a, s) : launch_igemm<float, 128, 64, 2>(a, s);
}
