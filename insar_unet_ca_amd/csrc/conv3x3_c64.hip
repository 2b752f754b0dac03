// 3x3 / stride-1 convolution with 64 input channels (forward and input-gradient), bf16, for the
// full-resolution level of the U-Net (Unet-ChannalAttention.py:81,84 with 64 channels and their dgrad).
//
// At K = 64 a per-tap implicit GEMM spends its time on synchronisation and on moving operands: 9 K-slabs
// per tile, each with its own LDS-DMA pieces, wait and barrier, for only 16 MFMAs per wave. This kernel
// removes both costs:
//   * WEIGHTS LIVE IN REGISTERS. A wave owns 32 (or 64) output channels; its share of all nine taps'
//     weights (32 x 64 x 9 bf16 = 36 KB per wave = 144 VGPRs per lane) is loaded once per work-group and
//     is the MFMA A operand for the whole launch.
//   * ACTIVATIONS STREAM THROUGH A ROLLING LDS WINDOW over the flat padded pixel space (NHWC with a zero
//     halo: the 3x3 neighbours of flat pixel q are q + dy*(W+2) + dx). Work-groups are persistent and walk
//     consecutive 256-pixel tiles; the window holds the pixels [tile - (W+3), tile + 256 + (W+3)) as a
//     ring of 128-byte rows, so EVERY INPUT PIXEL IS FETCHED ONCE (LDS-DMA, 4 pieces per wave per tile
//     instead of 54), all nine taps read it at row offsets, and there is ONE barrier per tile.
//   * outputs go from the accumulators straight to HBM (each lane holds 4 consecutive channels of a pixel:
//     8-byte stores; the two N-halves of a pixel complete its 128-byte line), BatchNorm partial sums of the
//     stored values are carried in registers across the tiles and written once per work-group.
// Halo pixels are computed and dropped (1.5 % at W = 256).
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;      // a register quad an inline-asm operand can name

#define C6_THREADS 512
#define C6_TILE 256
#define C6_ROWB 128
#define C6_MAX_LDS (160 * 1024)

struct C64Args {
  const char* x; const char* w; char* y; float* stats;
  long long P;                 // B*(H+2)*(W+2) padded pixels
  int H, W, Wp, img;           // img = (H+2)*Wp
  float inv_img, inv_wp;
  int Cx, cx_off, Cy, cy_off, N;
  int flip;
  int A;                       // reach of a tap in flat pixels: Wp + 1
  int Af;                      // A rounded down to a multiple of 64 (front lead of the loaded window)
  int o;                       // tile origin shift: Af - A (<= 0)
  int R;                       // ring rows (multiple of 64)
  int ntiles;
  const char* by; const float* bscale; const float* bshift;   // BatchNorm-backward sums in the stats slab (InsarBstat)
};

__device__ __forceinline__ void c6_mma(const u32x4_t& wa, const uint4& xb, f32x4_t& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wa), __builtin_bit_cast(bf16x8_t, xb), acc, 0, 0, 0);
}

// NTW = 16-channel tiles per wave along N (2: N = 64 with waves 4(M) x 2(N), wave tile 64 x 32)
// BS: BatchNorm-backward sums of the consumer unit in the statistics slab (InsarBstat) instead of (sum, sum of squares)
template <int NTW, bool BS = false>
__global__ __launch_bounds__(C6_THREADS, 1) void conv3x3_c64_kernel(C64Args a) {
  constexpr int MT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sstat = (float*)(smem + (size_t)a.R * C6_ROWB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int r16 = lane & 15, kq = lane >> 4;
  const int G = gridDim.x, g = blockIdx.x;
  const int t0 = (int)((long long)g * a.ntiles / G), t1 = (int)((long long)(g + 1) * a.ntiles / G);
  const int R = a.R, Wp = a.Wp;
  const int Pm1 = (int)(a.P - 1);

  // ---- prologue: initial window by LDS-DMA, then this wave's weights into registers ----------------
  // unit = 64 consecutive flat pixels (one block-wide DMA instruction: thread -> row tid>>3, chunk tid&7)
  const int urow = tid >> 3;
  const int schunk = ((tid & 7) ^ (urow & 7)) * 16;          // source-side XOR swizzle (ring rows of a unit start at a multiple of 64)
  const char* xbase = a.x + (long long)a.cx_off * 2 + schunk;
  const long long xpitch = (long long)a.Cx * 2;
  const uint32_t lds0 = lds_offset_of(smem) + wave * 1024;
  int fpix;                                                    // next pixel to fetch (multiple of 64, may start negative)
  {
    const int first = t0 * C6_TILE + a.o - a.A;               // first pixel tile t0 needs
    fpix = (first >= 0 ? first / 64 : -((-first + 63) / 64)) * 64;
  }
  const int u0 = fpix;                                         // ring row 0 <-> flat pixel u0
  int frow = 0;                                                // ring row of fpix
  auto fetch_unit = [&]() {
    int pix = fpix + urow;
    pix = pix < 0 ? 0 : (pix > Pm1 ? Pm1 : pix);
    lds_dma16_untracked(xbase + (long long)pix * xpitch, lds0 + (uint32_t)frow * C6_ROWB);
    fpix += 64;
    frow += 64;
    if (frow >= R) frow -= R;
  };
  if (t0 < t1) {
    const int front = t0 * C6_TILE + C6_TILE + a.Af;          // exclusive frontier tile t0 needs
    while (fpix < front) fetch_unit();
  }

  u32x4_t wr[9][2][NTW];
#pragma unroll
  for (int g9 = 0; g9 < 9; ++g9) {
    const int tap = a.flip ? 8 - g9 : g9;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int n = wn * (16 * NTW) + nt * 16 + r16;
        wr[g9][sub][nt] = *(const u32x4_t*)(a.w + (((long long)tap * a.N + n) * 64 + (kq + 4 * sub) * 8) * 2);
      }
  }

  float s1[NTW][4], s2[NTW][4];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[nt][j] = 0.f; s2[nt][j] = 0.f; }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ring row of output pixel i = 0 of the current tile
  int rb = (t0 * C6_TILE + a.o - u0) % R;
  const int lrow = wm * 64 + r16;                              // + mt*16: this lane's pixel inside the tile
  // Addressing: a wave reads 64 CONSECUTIVE ring rows per tap (r16 + 16*mt on top of a wave-uniform start),
  // so the ring wrap is decided on the scalar unit; only a span that straddles the ring end (6 % of the
  // taps) takes the per-lane path. R is a multiple of 8, so the swizzle term survives the wrap.
  const int wm_u = __builtin_amdgcn_readfirstlane(wm);
  const uint32_t lane_off = (uint32_t)r16 * C6_ROWB;           // + mt*16*128 as an immediate offset

  f32x4_t acc[NTW][MT];
  // (padded row, padded column) of this lane's first pixel of a tile, by reciprocal multiplies (+ one correction step each:
  // exact for q / img < 2^22 and img < 2^24, which c64_geometry() guarantees), on q + img so that the few negative pixel
  // indices of the first tile decompose like the others; the other three pixels are 16, 32, 48 further on: one conditional
  // wrap each instead of the two divisions. okm[mt]: an interior pixel of the buffer; off[mt]: byte offset of this lane's
  // first channel of it in y (and in the consumer's y of the BS variant: same layout).
  auto pixel_offsets = [&](int tile, bool (&okm)[MT], long long (&off)[MT]) {
    const int qt = tile * C6_TILE + a.o;
    int hr, wc;
    {
      const int qb = qt + lrow + a.img;
      const int n = (int)((float)qb * a.inv_img);
      int rem = qb - n * a.img;
      if (rem < 0) { rem += a.img; } else if (rem >= a.img) { rem -= a.img; }
      hr = (int)((float)rem * a.inv_wp);
      wc = rem - hr * Wp;
      if (wc < 0) { wc += Wp; --hr; } else if (wc >= Wp) { wc -= Wp; ++hr; }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int q = qt + lrow + mt * 16;
      if (mt > 0) {
        wc += 16;
        if (wc >= Wp) { wc -= Wp; ++hr; if (hr >= a.H + 2) hr = 0; }
      }
      okm[mt] = q >= 0 && q <= Pm1 && hr >= 1 && hr <= a.H && wc >= 1 && wc <= a.W;
      off[mt] = ((long long)q * a.Cy + a.cy_off + wn * (16 * NTW) + kq * 4) * 2;
    }
  };
  // BS: the consumer's y at this lane's 4 pixels x NTW x 4 channels. The loads are issued right after the first two taps of
  // a tile and take over those taps' weight registers (the kernel sits at 256): they fly under the other seven taps
  // (~3 500 cycles) instead of being requested at the head of the epilogue, where their HBM latency was exposed once per
  // tile (16 tiles x ~2.5 us: the 46 us per launch this variant cost over the plain one); the two taps' weights are
  // read again (8 KB per wave, L2) at the end of the epilogue.
  uint2 yv[MT][NTW];
  auto load_y = [&](int tile) {
    // unconditional loads (pixels beyond the buffer clamped into it): what they return for halo pixels is never used
    const int qt = tile * C6_TILE + a.o + lrow;
    const char* yb = a.by + ((long long)a.cy_off + wn * (16 * NTW) + kq * 4) * 2;
    const long long ypitch = (long long)a.Cy * 2;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int q = qt + mt * 16;
      q = q < 0 ? 0 : (q > Pm1 ? Pm1 : q);
      const char* yp = yb + q * ypitch;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) yv[mt][nt] = *(const uint2*)(yp + nt * 32);
    }
  };
  constexpr int NREL = 2;                                      // leading taps whose weights are re-read per tile (BS)
  auto load_first_weights = [&]() {
#pragma unroll
    for (int g9 = 0; g9 < NREL; ++g9) {
      const int tap = a.flip ? 8 - g9 : g9;
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int n = wn * (16 * NTW) + nt * 16 + r16;
          // issued from inline asm: hipcc's own wait for a load it knows about would sit at the first use of these registers,
          // behind the next tile's LDS-DMA pieces in the queue, and wait for those too (the loop's DMA is invisible to it)
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(wr[g9][sub][nt])
                       : "v"(a.w + (((long long)tap * a.N + n) * 64 + (kq + 4 * sub) * 8) * 2) : "memory");
        }
    }
  };
  auto wait_first_weights = [&]() {        // ... and waited for by hand: every destination named, so nothing reads them earlier
    static_assert(NREL == 2 && NTW == 2, "operand list below");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(wr[0][0][0]), "+v"(wr[0][0][1]), "+v"(wr[0][1][0]), "+v"(wr[0][1][1]),
                   "+v"(wr[1][0][0]), "+v"(wr[1][0][1]), "+v"(wr[1][1][0]), "+v"(wr[1][1][1]) :: "memory");
  };
  // stores + BatchNorm partial sums (interior pixels only) of the tile whose sums are in `acc`
  auto epilogue = [&](int tile) {
    bool okm[MT];
    long long off[MT];
    pixel_offsets(tile, okm, off);
    if constexpr (BS) {
      // the consumer's scale / shift re-read per tile (L2 hits) rather than held in 16 registers beside the weights
      f32x4_t bsc[NTW], bsh[NTW];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        bsc[nt] = *(const f32x4_t*)(a.bscale + wn * (16 * NTW) + nt * 16 + kq * 4);
        bsh[nt] = *(const f32x4_t*)(a.bshift + wn * (16 * NTW) + nt * 16 + kq * 4);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        if (okm[mt]) {
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            uint2 v;
            v.x = pack2_bf16(acc[nt][mt][0], acc[nt][mt][1]);
            v.y = pack2_bf16(acc[nt][mt][2], acc[nt][mt][3]);
            *(uint2*)(a.y + off[mt] + nt * 32) = v;
            const float f[4] = {__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u)};
            const float yy[4] = {__uint_as_float(yv[mt][nt].x << 16), __uint_as_float(yv[mt][nt].x & 0xffff0000u),
                                 __uint_as_float(yv[mt][nt].y << 16), __uint_as_float(yv[mt][nt].y & 0xffff0000u)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float m = fmaf(yy[j], bsc[nt][j], bsh[nt][j]) > 0.f ? f[j] : 0.f;
              s1[nt][j] += m; s2[nt][j] = fmaf(m, yy[j], s2[nt][j]);
            }
          }
        }
      return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (okm[mt]) {
        char* yp = a.y + off[mt];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          uint2 v;
          v.x = pack2_bf16(acc[nt][mt][0], acc[nt][mt][1]);
          v.y = pack2_bf16(acc[nt][mt][2], acc[nt][mt][3]);
          *(uint2*)(yp + nt * 32) = v;
          const float f0 = __uint_as_float(v.x << 16), f1 = __uint_as_float(v.x & 0xffff0000u);
          const float f2 = __uint_as_float(v.y << 16), f3 = __uint_as_float(v.y & 0xffff0000u);
          s1[nt][0] += f0; s2[nt][0] = fmaf(f0, f0, s2[nt][0]);
          s1[nt][1] += f1; s2[nt][1] = fmaf(f1, f1, s2[nt][1]);
          s1[nt][2] += f2; s2[nt][2] = fmaf(f2, f2, s2[nt][2]);
          s1[nt][3] += f3; s2[nt][3] = fmaf(f3, f3, s2[nt][3]);
        }
      }
    }
  };

  for (int t = t0; t < t1; ++t) {
    // next tile's 256 new pixels (their ring rows were last read by tile t-1; every wave is past it)
    if (t + 1 < t1) {
#pragma unroll
      for (int u = 0; u < C6_TILE / 64; ++u) fetch_unit();
    }

#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int rb_u = __builtin_amdgcn_readfirstlane(rb);
    // 18 half-tap steps (tap g9, K half sub), software-pipelined by hand: the fragments of step s+1 are
    // requested before the MFMAs of step s are issued (two fragment sets ping-pong), so an LDS read has a
    // whole step (8 MFMAs) to return instead of two MFMAs.
    auto tap_addr = [&](int g9, uint32_t (&addr)[MT]) {
      const int doff = (g9 / 3 - 1) * Wp + (g9 % 3 - 1);
      int start = rb_u + wm_u * 64 + doff;                     // wave-uniform: ring row of this wave's first pixel
      start = start < 0 ? start + R : (start >= R ? start - R : start);
      if (start + 63 < R) {
        const uint32_t sw = (uint32_t)((kq ^ ((start + r16) & 7)) << 4);
        const uint32_t base = (uint32_t)start * C6_ROWB + lane_off + sw;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) addr[mt] = base + mt * 16 * C6_ROWB;
      } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          int row = start + r16 + mt * 16;
          row = row >= R ? row - R : row;
          addr[mt] = (uint32_t)row * C6_ROWB + (uint32_t)((kq ^ (row & 7)) << 4);
        }
      }
    };
    uint4 xf[2][MT];
    uint32_t addr[MT];
    tap_addr(0, addr);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xf[0][mt] = *(const uint4*)(smem + addr[mt]);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
      const int g9 = st >> 1, sub = st & 1;
      if (st + 1 < 18) {
        if (sub == 1) tap_addr(g9 + 1, addr);                  // next step starts a new tap
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[(st + 1) & 1][mt] = *(const uint4*)(smem + (addr[mt] ^ (uint32_t)((sub ^ 1) * 64)));
      }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) c6_mma(wr[g9][sub][nt], xf[st & 1][mt], acc[nt][mt]);
      if constexpr (BS) {
        if (st == 2 * NREL - 1) load_y(t);                     // the first taps are done: their weight registers carry the y loads now
      }
    }

    // the prefetched rows must have landed before any wave starts the next tile (they had this tile's
    // whole compute phase); the previous tile's stores are long done, so the wait is free.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // lgkmcnt: see common.h, dma_drain_and_barrier

    epilogue(t);
    if constexpr (BS) {
      load_first_weights();                                  // (unconditionally: a guarded re-read would keep the old registers live across the tile)
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (BS) {
      wait_first_weights();            // before the next tile's LDS-DMA pieces are issued
    }
    rb += C6_TILE;
    if (rb >= R) rb -= R;
  }

  // ---- BatchNorm partial sums of this work-group: one slab row [2][N] -------------------------------
  if (a.stats) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1) { s1[nt][j] += __shfl_xor(s1[nt][j], sh, 64); s2[nt][j] += __shfl_xor(s2[nt][j], sh, 64); }
      }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) { LDS_PIN(s1[nt][j]); LDS_PIN(s2[nt][j]); }
    __syncthreads();                                           // every wave is done with the ring
    if (r16 == 0) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = nt * 16 + kq * 4 + j;                  // channel inside this wave's N slice
          sstat[((wave * (16 * NTW)) + c) * 2 + 0] = s1[nt][j];
          sstat[((wave * (16 * NTW)) + c) * 2 + 1] = s2[nt][j];
        }
    }
    LDS_DRAIN();
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) { LDS_KEEP(s1[nt][j]); LDS_KEEP(s2[nt][j]); }
    __syncthreads();
    if (tid < a.N) {
      const int hn = tid / (16 * NTW), c = tid - hn * (16 * NTW);   // N half (wn) and channel inside it
      float v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int w = hn * 4 + m;
        v1 += sstat[(w * (16 * NTW) + c) * 2 + 0];
        v2 += sstat[(w * (16 * NTW) + c) * 2 + 1];
      }
      a.stats[((long long)g * 2 + 0) * a.N + tid] = v1;
      a.stats[((long long)g * 2 + 1) * a.N + tid] = v2;
    }
  }
}

// ---- host side -----------------------------------------------------------------------------------
static int c64_num_cus() { return insar_num_cus(); }

static bool c64_geometry(const InsarAct& x, C64Args& a) {
  const long long P = (long long)x.B * (x.H + 2) * (x.W + 2);
  if (P >= 0x7fffff00LL) return false;
  a.P = P; a.H = x.H; a.W = x.W; a.Wp = x.W + 2; a.img = (x.H + 2) * (x.W + 2);
  if (a.img >= (1 << 24) || P / a.img >= (1 << 22)) return false;      // reciprocal-multiply index split stays exact
  a.inv_img = 1.0f / (float)a.img; a.inv_wp = 1.0f / (float)a.Wp;
  a.A = a.Wp + 1;
  a.Af = (a.A / 64) * 64;
  a.o = a.Af - a.A;
  // Ring rows: while tile t is computed, its own window [t*TILE + o - A, t*TILE + TILE + Af) is being read AND the
  // four units of tile t+1, up to t*TILE + 2*TILE + Af, are landing by LDS-DMA: the live span is 2*TILE + Af + A - o
  // (= 2*TILE + 2*A) pixels. (Round 1 sized it without the -o term: the newest prefetched rows aliased the oldest
  // rows still being read whenever (W+3) % 64 >= 33; tests/test_host_logic.py walks this geometry for every W.)
  a.R = ((2 * C6_TILE + a.Af + a.A - a.o + 63) / 64) * 64;
  a.ntiles = (int)((P - a.o + C6_TILE - 1) / C6_TILE);
  return (long long)a.R * C6_ROWB + 8 * 64 * 2 * 4 <= C6_MAX_LDS;
}

// bf16, 64 -> 64 channels, image rows short enough for the window to fit the LDS.
extern "C" int insar_conv3x3_c64_ok(const InsarAct* x, int32_t N) {
  if (!x || x->dtype != INSAR_BF16 || x->c_len != 64 || N != 64) return 0;
  C64Args a;
  return c64_geometry(*x, a) ? 1 : 0;
}

// Window geometry of the launch for x (host-side unit tests): out = {A, Af, o, R, ntiles, TILE}. Returns 1 when the
// kernel accepts the shape, 0 otherwise.
extern "C" int insar_conv3x3_c64_geometry(const InsarAct* x, int32_t* out) {
  C64Args a;
  if (!x || !out || !c64_geometry(*x, a)) return 0;
  out[0] = a.A; out[1] = a.Af; out[2] = a.o; out[3] = a.R; out[4] = a.ntiles; out[5] = C6_TILE;
  return 1;
}

// number of work-groups = rows of the BatchNorm partial-sum slab this launch writes
extern "C" int insar_conv3x3_c64_rows(const InsarAct* x) {
  C64Args a;
  if (!x || !c64_geometry(*x, a)) return 0;
  const int cus = c64_num_cus(), kb = insar_knob(KNOB_C64_GRID_BWD);
  const int g = kb > cus ? kb : cus;                  // rows for either direction (rows a launch does not write stay zero)
  return a.ntiles < g ? a.ntiles : g;
}

// y = conv3x3(x, w), 64 -> 64 channels, same (B, H, W) grid. w: [9][64][64] bf16 in (dy, dx) raster order;
// flip != 0 walks the taps backwards (the dgrad operand of insar_weight_prep). stats: [rows][2][64] or null.
static int c64_impl(const InsarAct* x, const InsarAct* y, const void* w, int32_t flip, float* stats, const InsarBstat* bstat,
                    void* stream) {
  if (!x || !y || !w) INSAR_FAIL(INSAR_E_ARG, "insar_conv3x3_c64: null pointer");
  int rc;
  if ((rc = insar_check_act(x, "insar_conv3x3_c64", "x"))) return rc;
  if ((rc = insar_check_act(y, "insar_conv3x3_c64", "y"))) return rc;
  if (x->dtype != INSAR_BF16 || y->dtype != INSAR_BF16) INSAR_FAIL(INSAR_E_DTYPE, "insar_conv3x3_c64: bf16 only");
  if (x->B != y->B || x->H != y->H || x->W != y->W) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_c64: x/y grids differ");
  if (x->c_len != 64 || y->c_len != 64) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_c64: 64 -> 64 channels only (got %d -> %d)", x->c_len, y->c_len);
  if (!insar_aligned16(w)) INSAR_FAIL(INSAR_E_ALIGN, "insar_conv3x3_c64: weights not 16-byte aligned");
  C64Args a;
  if (!c64_geometry(*x, a)) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_c64: W=%d too wide for the LDS window (or too many pixels)", x->W);
  a.x = (const char*)x->ptr; a.w = (const char*)w; a.y = (char*)y->ptr; a.stats = stats;
  a.Cx = x->C; a.cx_off = x->c_off; a.Cy = y->C; a.cy_off = y->c_off; a.N = 64;
  a.flip = flip ? 1 : 0;
  a.by = nullptr; a.bscale = a.bshift = nullptr;
  if (bstat && bstat->y) {
    if (!stats || !bstat->scale || !bstat->shift) INSAR_FAIL(INSAR_E_ARG, "insar_conv3x3_c64_bstat: needs a stats slab, scale and shift");
    if (!insar_aligned16(bstat->y) || !insar_aligned16(bstat->scale) || !insar_aligned16(bstat->shift))
      INSAR_FAIL(INSAR_E_ALIGN, "insar_conv3x3_c64_bstat: bstat pointers not 16-byte aligned");
    a.by = (const char*)bstat->y; a.bscale = bstat->scale; a.bshift = bstat->shift;
  }
  const int lds = a.R * C6_ROWB + 8 * 64 * 2 * 4;
  static std::atomic<uint64_t> attr_mask{0}, attr_mask_bs{0};     // per-device, see common.h
  {
    hipError_t e = a.by ? insar_set_lds_once(attr_mask_bs, (const void*)conv3x3_c64_kernel<2, true>, C6_MAX_LDS)
                        : insar_set_lds_once(attr_mask, (const void*)conv3x3_c64_kernel<2>, C6_MAX_LDS);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_conv3x3_c64: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const int kb = insar_knob(KNOB_C64_GRID_BWD);
  const int cus = (a.flip && kb > 0) ? kb : c64_num_cus();
  const int grid = a.ntiles < cus ? a.ntiles : cus;
  if (a.by) hipLaunchKernelGGL((conv3x3_c64_kernel<2, true>), dim3(grid), dim3(C6_THREADS), lds, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((conv3x3_c64_kernel<2>), dim3(grid), dim3(C6_THREADS), lds, (hipStream_t)stream, a);
  INSAR_CHECK_LAUNCH("insar_conv3x3_c64");
  return INSAR_OK;
}

extern "C" int insar_conv3x3_c64(const InsarAct* x, const InsarAct* y, const void* w, int32_t flip, float* stats,
                                 void* stream) {
  return c64_impl(x, y, w, flip, stats, nullptr, stream);
}
extern "C" int insar_conv3x3_c64_bstat(const InsarAct* x, const InsarAct* y, const void* w, int32_t flip, float* stats,
                                       const InsarBstat* bstat, void* stream) {
  return c64_impl(x, y, w, flip, stats, bstat, stream);
}
