// Weight gradient of a 3x3 / stride-1 convolution (Unet-ChannalAttention.py:81,84 inside loss.backward(), :345)
// with the three taps of a kernel ROW computed by one work-group: the same split-K MFMA GEMM as wgrad.hip, but
//
//   part[split][3*ty + tx][co][ci] = sum_p dY[p, co] * X[p + (ty-1)*(W+2) + (tx-1), ci],   tx = 0, 1, 2
//
// shares its operands between the taps. The small-tile weight-gradient launches are bound by LDS-DMA issue and by
// the L2 -> LDS operand feed, not by the matrix pipe (with the staging removed the 64 x 64 kernel runs 2.1x faster:
// tools/gemm_bench.py, 202 -> 97 us): here a K step of 64 CONSECUTIVE pixels of one image row stages dY once
// (64 rows) and X once with a one-pixel lead and tail (66 rows of a 72-row slot; the zero halo of the padded NHWC
// layout is the conv padding), 17 one-KB pieces per step instead of 48, and the dY fragments stay in registers for
// the three taps. Needs W % 64 == 0 (a K step is a piece of one image row) or W = 16 / 32 (a K step is 4 / 2 whole
// image rows, whose X operand is still one contiguous run of <= 72 padded pixels): every level of the U-Net;
// anything else keeps the per-tap kernel. Same LDS image (pixel rows, XOR-swizzled on the DMA source side,
// ds_read_b64_tr_b16 transposing reads), same slab layout, same fold: bitwise reproducible.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define W3_BKP 64       // pixels per K step
#define W3_XR 72        // X rows staged per step: pixel p0 - 1 + r, r < 66 used

struct Wgrad3Args {
  const char* x; const char* dy; float* part;
  long long ksteps;
  int nsplit, steps_per_split;
  int H, W, Wp;
  int spr, rpk, lw;             // K steps per image row (W >= 64) / image rows per K step (W < 64) / log2(W) (6 if W >= 64)
  int Cx, cx_off, Cin; int Cdy, cdy_off, Cout;
  int mtc, ntc;
};

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// XOR of the 16-byte chunk index of a staged row (DMA source side and read side alike). MF = 16: the two lane groups
// of a 32-lane half read DIFFERENT pixel rows (0-3 and 12-15 of their 16) of the same 32-byte channel block. MF = 32
// (v_mfma_f32_32x32x16_bf16 fragments, 256-byte rows only): they read the SAME four pixel rows of two adjacent channel
// blocks, so four consecutive rows must land in four different 64-byte segments of the 256-byte bank row.
template <int RB, int MF = 16>
__device__ __forceinline__ int w3_swz(int row) {
  if constexpr (MF == 32) return ((row & 3) << 2) | (((row >> 2) & 1) << 1);
  else if constexpr (RB == 256) return (row & 7) << 1;
  else return ((row >> 1) & 3) << 1;
}

// Diagnostic build only (-DINSAR_STAMPS, tools/stamp_gemm.py)
#ifdef INSAR_STAMPS
__device__ unsigned long long g_wgrad3_stamps[1024 * 8];
#define W3_STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_acc[k] += now_ - stamp_prev; stamp_prev = now_; } while (0)
extern "C" int insar_debug_wgrad3_stamps(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgrad3_stamps), sizeof(g_wgrad3_stamps)) != hipSuccess) return -1;
  if (reset) { static unsigned long long z[1024 * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wgrad3_stamps), z, sizeof(z)) != hipSuccess) return -2; }
  return 0;
}
#else
#define W3_STAMP(k)
#endif

template <typename T, int TM, int TN, int NW>
struct Wgrad3Cfg {
  static constexpr int ES = sizeof(T);
  static constexpr int THREADS = NW * 64;
  static constexpr int SUBX = (TM * ES > 256) ? TM * ES / 256 : 1, SUBY = (TN * ES > 256) ? TN * ES / 256 : 1;
  static constexpr int RBX = TM * ES / SUBX, RBY = TN * ES / SUBY;
  static constexpr int X_STAGE = W3_XR * TM * ES, Y_STAGE = W3_BKP * TN * ES;
  static constexpr int STAGE = X_STAGE + Y_STAGE;
  static constexpr int LDS_BYTES = 2 * STAGE;
  static constexpr int CPRX = RBX / 16, CPRY = RBY / 16;
  static constexpr int XCHUNKS = X_STAGE / 16, YCHUNKS = Y_STAGE / 16;
  static constexpr int NX = (XCHUNKS + THREADS - 1) / THREADS;   // DMA chunks per thread per step (last one may be partial)
  static constexpr int NY = YCHUNKS / THREADS;
  static constexpr int MTW = TM / 32, NTW = TN / (NW / 2) / 16;
  static_assert(YCHUNKS % THREADS == 0, "dY stage must be whole block-wide DMA instructions");
};

template <typename T, int TM, int TN, int NW, int MF = 16>
__global__ __launch_bounds__(NW * 64, (NW == 4 ? 2 : 1)) void wgrad3_kernel(Wgrad3Args a) {
  using Cfg = Wgrad3Cfg<T, TM, TN, NW>;
  static_assert(MF == 16 || (MF == 32 && sizeof(T) == 2 && TM == 128 && TN == 128 && NW == 8), "32x32x16 fragments: 128 x 128 bf16 tiles");
  constexpr int THREADS = Cfg::THREADS, ES = Cfg::ES;
  constexpr int MTW = Cfg::MTW, NTW = Cfg::NTW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef INSAR_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
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

  const int ks0 = split * a.steps_per_split;            // B*H*W/64 K steps fit an int (buffers are < 4 GiB)
  int ks1 = ks0 + a.steps_per_split;
  if (ks1 > (int)a.ksteps) ks1 = (int)a.ksteps;

  // per-thread constants of its DMA chunks: chunk q = i*THREADS + tid -> sub-tile, row, swizzled source position
  int xrow_i[Cfg::NX], xoff_i[Cfg::NX];
  bool xok_i[Cfg::NX];
#pragma unroll
  for (int i = 0; i < Cfg::NX; ++i) {
    const int q = i * THREADS + tid;
    const int sub = q / (W3_XR * Cfg::CPRX), row = (q / Cfg::CPRX) % W3_XR, pos = q % Cfg::CPRX;
    xrow_i[i] = row;
    xoff_i[i] = sub * Cfg::RBX + (pos ^ w3_swz<Cfg::RBX, MF>(row)) * 16;
    xok_i[i] = q < Cfg::XCHUNKS && row < W3_BKP + 2 * a.rpk;
  }
  int yrow_i[Cfg::NY], yoff_i[Cfg::NY];
#pragma unroll
  for (int i = 0; i < Cfg::NY; ++i) {
    const int q = i * THREADS + tid;
    const int sub = q / (W3_BKP * Cfg::CPRY), row = (q / Cfg::CPRY) % W3_BKP, pos = q % Cfg::CPRY;
    yrow_i[i] = row + 2 * (row >> a.lw);              // padded pixel offset of dY row `row` (halo pixels skipped)
    yoff_i[i] = sub * Cfg::RBY + (pos ^ w3_swz<Cfg::RBY, MF>(row)) * 16;
  }
  const long long xpitch = (long long)a.Cx * ES, ypitch = (long long)a.Cdy * ES;
  // A K step is 64 interior pixels: a piece of one image row (W >= 64) or 64/W whole image rows (W = 16, 32). In the
  // padded layout (rows of W+2 pixels) its X operand is ONE contiguous run: X row r <-> padded pixel
  // p0 + (ty-1)*Wp - 1 + r, r < 64 + 2*rpk <= 72; dY row k <-> padded pixel p0 + k + 2*(k / W) (halo pixels skipped),
  // and the X row of (dY row k, tap tx) is k + 2*(k / W) + tx.
  const char* xbase = a.x + ((long long)(ty - 1) * a.Wp - 1) * xpitch + ((long long)a.cx_off + mi * TM) * ES;
  const char* ybase = a.dy + ((long long)a.cdy_off + ni * TN) * ES;

  // padded index of the first pixel of a K step, advanced step by step (all wave-uniform)
  int seg, hrow, img;
  {
    const int g = ks0 / a.spr, gpi = a.H / a.rpk;       // row group (rpk image rows) and groups per image
    seg = ks0 - g * a.spr;
    img = g / gpi;
    hrow = (g - img * gpi) * a.rpk;
  }
  auto next_pixel = [&]() -> long long {
    const long long p = ((long long)img * (a.H + 2) + hrow + 1) * a.Wp + seg * W3_BKP + 1;
    if (++seg == a.spr) {
      seg = 0; hrow += a.rpk;
      if (hrow >= a.H) { hrow = 0; ++img; }
    }
    return p;
  };
  const uint32_t lds0 = lds_offset_of(smem);
  auto stage = [&](int buf, long long p0) {
    const uint32_t lx = lds0 + buf * Cfg::STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < Cfg::NX; ++i)
      if (xok_i[i]) lds_dma16_untracked(xbase + (p0 + xrow_i[i]) * xpitch + xoff_i[i], lx + i * (THREADS * 16));
    const uint32_t ly = lx + Cfg::X_STAGE;
#pragma unroll
    for (int i = 0; i < Cfg::NY; ++i)
      lds_dma16_untracked(ybase + (p0 + yrow_i[i]) * ypitch + yoff_i[i], ly + i * (THREADS * 16));
  };

  // MF = 16: 16x16 accumulators [tap][MTW][NTW]; MF = 32: the wave's 64 (ci) x 32 (co) as two 32x32 accumulators per tap
  constexpr int A16 = MF == 16 ? 3 : 1, A32 = MF == 32 ? 3 : 1;
  f32x4_t acc[A16][MTW][NTW];
  f32x16_t acc32[A32][2];
#pragma unroll
  for (int t3 = 0; t3 < A16; ++t3)
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[t3][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t3 = 0; t3 < A32; ++t3)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc32[t3][i][e] = 0.f;

  const int wm = wave & 1, wn = wave >> 1;
  const int r16 = lane & 15, kq = lane >> 4;

  if (ks0 < ks1) {
    W3_STAMP(0);        // set-up
    stage(0, next_pixel());
    dma_drain_and_barrier();
    W3_STAMP(1);        // first step landed
    for (int ks = ks0; ks < ks1; ++ks) {
      const int buf = (ks - ks0) & 1;
      if (ks + 1 < ks1) stage(buf ^ 1, next_pixel());
      const char* sX = smem + buf * Cfg::STAGE;
      const char* sY = sX + Cfg::X_STAGE;
      if constexpr (ES == 2 && MF == 32) {
        // lane = 16*g + i: channel block cb = g & 1 (16 channels), k half hq = g >> 1 (8 pixels) of a 16-pixel MFMA step;
        // a transposing read hands lane i channel i of the four pixel rows its 16-lane group addresses (row q = i >> 2)
        const int cb = kq & 1, hq = kq >> 1;
  #pragma unroll
        for (int s = 0; s < 4; ++s) {
          bf16x8_t yf;
  #pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int row = s * 16 + hq * 8 + h * 4 + (r16 >> 2);
            const int colb = (wn * 32 + cb * 16 + (r16 & 3) * 4) * 2;
            const int pc = (colb >> 4) ^ w3_swz<Cfg::RBY, MF>(row);
            const char* p = sY + row * Cfg::RBY + pc * 16 + (colb & 15);
            s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
            yf[4 * h + 0] = v[0]; yf[4 * h + 1] = v[1]; yf[4 * h + 2] = v[2]; yf[4 * h + 3] = v[3];
          }
  #pragma unroll
          for (int t3 = 0; t3 < 3; ++t3) {
            bf16x8_t xf[2];
  #pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int k = s * 16 + hq * 8 + h * 4 + (r16 >> 2);
              const int row = k + 2 * (k >> a.lw) + t3;
  #pragma unroll
              for (int mt = 0; mt < 2; ++mt) {
                const int colb = (wm * 64 + mt * 32 + cb * 16 + (r16 & 3) * 4) * 2;
                const int pc = (colb >> 4) ^ w3_swz<Cfg::RBX, MF>(row);
                const char* p = sX + row * Cfg::RBX + pc * 16 + (colb & 15);
                s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
                xf[mt][4 * h + 0] = v[0]; xf[mt][4 * h + 1] = v[1]; xf[mt][4 * h + 2] = v[2]; xf[mt][4 * h + 3] = v[3];
              }
            }
  #pragma unroll
            for (int mt = 0; mt < 2; ++mt)
              acc32[t3][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[mt], yf, acc32[t3][mt], 0, 0, 0);
          }
        }
      } else if constexpr (ES == 2) {
  #pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8_t yf[NTW];
  #pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int sel = h ^ (kq & 1);
            const int row = s * 32 + kq * 8 + sel * 4 + (r16 >> 2);
  #pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
              const int colw = (wn * (NTW * 16) + nt * 16 + (r16 & 3) * 4) * 2;
              int sub = 0, colb = colw;
              if constexpr (Cfg::SUBY > 1) { sub = colw / Cfg::RBY; colb = colw % Cfg::RBY; }
              const int pc = (colb >> 4) ^ w3_swz<Cfg::RBY>(row);
              const char* p = sY + sub * (W3_BKP * Cfg::RBY) + row * Cfg::RBY + pc * 16 + (colb & 15);
              s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
              yf[nt][4 * h + 0] = v[0]; yf[nt][4 * h + 1] = v[1]; yf[nt][4 * h + 2] = v[2]; yf[nt][4 * h + 3] = v[3];
            }
          }
  #pragma unroll
          for (int t3 = 0; t3 < 3; ++t3) {
            bf16x8_t xf[MTW];
  #pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int sel = h ^ (kq & 1);
              const int k = s * 32 + kq * 8 + sel * 4 + (r16 >> 2);
              const int row = k + 2 * (k >> a.lw) + t3;                         // X rows lead the dY rows by one pixel
  #pragma unroll
              for (int mt = 0; mt < MTW; ++mt) {
                const int colw = (wm * (TM / 2) + mt * 16 + (r16 & 3) * 4) * 2;
                int sub = 0, colb = colw;
                if constexpr (Cfg::SUBX > 1) { sub = colw / Cfg::RBX; colb = colw % Cfg::RBX; }
                const int pc = (colb >> 4) ^ w3_swz<Cfg::RBX>(row);
                const char* p = sX + sub * (W3_XR * Cfg::RBX) + row * Cfg::RBX + pc * 16 + (colb & 15);
                s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
                xf[mt][4 * h + 0] = v[0]; xf[mt][4 * h + 1] = v[1]; xf[mt][4 * h + 2] = v[2]; xf[mt][4 * h + 3] = v[3];
              }
            }
  #pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
  #pragma unroll
              for (int nt = 0; nt < NTW; ++nt)
                acc[t3][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[mt], yf[nt], acc[t3][mt][nt], 0, 0, 0);
          }
        }
      } else {
        // fp32: v_mfma_f32_16x16x4_f32, one pixel row per lane group (kq) and step
#pragma unroll 4
        for (int jj = 0; jj < W3_BKP / 4; ++jj) {
          const int k = jj * 4 + kq;
          float yf[NTW];
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            const int colw = (wn * (NTW * 16) + nt * 16 + r16) * 4;
            const int sub = colw / Cfg::RBY, colb = colw % Cfg::RBY;
            const int pc = (colb >> 4) ^ w3_swz<Cfg::RBY>(k);
            yf[nt] = *(const float*)(sY + sub * (W3_BKP * Cfg::RBY) + k * Cfg::RBY + pc * 16 + (colb & 15));
          }
#pragma unroll
          for (int t3 = 0; t3 < 3; ++t3) {
            const int row = k + 2 * (k >> a.lw) + t3;
            float xf[MTW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
              const int colw = (wm * (TM / 2) + mt * 16 + r16) * 4;
              const int sub = colw / Cfg::RBX, colb = colw % Cfg::RBX;
              const int pc = (colb >> 4) ^ w3_swz<Cfg::RBX>(row);
              xf[mt] = *(const float*)(sX + sub * (W3_XR * Cfg::RBX) + row * Cfg::RBX + pc * 16 + (colb & 15));
            }
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
              for (int nt = 0; nt < NTW; ++nt)
                acc[t3][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[mt], yf[nt], acc[t3][mt][nt], 0, 0, 0);
          }
        }
      }
      dma_drain_and_barrier();
    }
  }

  W3_STAMP(2);          // K loop
  if constexpr (MF == 32) {
    // C layout of a 32x32 accumulator: col (co) = lane & 31, row (ci) = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
      float* out = a.part + ((long long)split * 9 + ty * 3 + t3) * a.Cout * a.Cin;
      const int co = ni * TN + wn * 32 + (lane & 31);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int ci = mi * TM + wm * 64 + mt * 32 + 8 * b + 4 * (lane >> 5);
          const f32x16_t c = acc32[t3][mt];
          *(f32x4_t*)(out + (long long)co * a.Cin + ci) = (f32x4_t){c[4 * b], c[4 * b + 1], c[4 * b + 2], c[4 * b + 3]};
        }
    }
    return;
  }
  // C layout: row (ci) = kq*4 + reg, col (co) = r16  ->  16-byte stores into [co][ci]
#pragma unroll
  for (int t3 = 0; t3 < 3; ++t3) {
    float* out = a.part + ((long long)split * 9 + ty * 3 + t3) * a.Cout * a.Cin;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const int ci = mi * TM + wm * (TM / 2) + mt * 16 + kq * 4;
        const int co = ni * TN + wn * (NTW * 16) + nt * 16 + r16;
        *(f32x4_t*)(out + (long long)co * a.Cin + ci) = acc[t3][mt][nt];
      }
  }
#ifdef INSAR_STAMPS
  W3_STAMP(3);          // slab stores
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&g_wgrad3_stamps[(blockIdx.x & 1023) * 8 + k], stamp_acc[k]);
  }
#endif
}

template <typename T, int TM, int TN, int NW, int MF = 16>
static int launch_wgrad3(Wgrad3Args& a, hipStream_t s) {
  using Cfg = Wgrad3Cfg<T, TM, TN, NW>;
  static std::atomic<uint64_t> attr_mask{0};     // per-device, see common.h
  {
    hipError_t e = insar_set_lds_once(attr_mask, (const void*)wgrad3_kernel<T, TM, TN, NW, MF>, Cfg::LDS_BYTES);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_wgrad_conv3: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  a.mtc = a.Cin / TM; a.ntc = a.Cout / TN;
  const long long grid = (long long)a.nsplit * 3 * a.mtc * a.ntc;
  if (grid > 0x7fffffffLL) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3: grid too large");
  hipLaunchKernelGGL((wgrad3_kernel<T, TM, TN, NW, MF>), dim3((unsigned)grid), dim3(Cfg::THREADS), Cfg::LDS_BYTES, s, a);
  INSAR_CHECK_LAUNCH("insar_wgrad_conv3");
  return INSAR_OK;
}

// (tile(Cin) << 16) | tile(Cout) of the row-of-taps kernel for this layer, or 0 when the per-tap kernel (insar_wgrad)
// has to be used: image rows that are whole K steps (or K steps that are whole image rows), and a tile whose three
// accumulator sets fit the registers.
extern "C" int insar_wgrad_conv3_tile(const InsarAct* x, int32_t Cout) {
  if (!x || x->c_len % 64 || Cout % 64) return 0;
  const bool rows_ok = (x->W % W3_BKP) == 0 || ((x->W == 16 || x->W == 32) && (x->H % (W3_BKP / x->W)) == 0);
  if (!rows_ok) return 0;
  const int pair = insar_wgrad_tile_pair(x->c_len, Cout, x->dtype);
  int tm = pair >> 16, tn = pair & 0xffff;
  // three accumulator sets of a 256 x 256 tile do not fit the registers: 128 x 128 (8 waves) there; sharing the
  // operands between the taps saves more staging than the wider tile did
  if (tm > 128) tm = 128;
  if (tn > 128) tn = 128;
  return (tm << 16) | tn;
}

// part[split][tap][co][ci] (tap = 3*ty + tx, the layout insar_wgrad writes) for a 3x3 / stride-1 / pad-1 convolution:
// x (B, H, W, Cin) and dy (B, H, W, Cout) on the same grid; nsplit splits of the B*H*W/64 K steps.
extern "C" int insar_wgrad_conv3(const InsarAct* x, const InsarAct* dy, float* part, int32_t nsplit, void* stream) {
  if (!x || !dy || !part) INSAR_FAIL(INSAR_E_ARG, "insar_wgrad_conv3: null pointer");
  int rc;
  if ((rc = insar_check_act(x, "insar_wgrad_conv3", "x"))) return rc;
  if ((rc = insar_check_act(dy, "insar_wgrad_conv3", "dy"))) return rc;
  if (x->B != dy->B || x->H != dy->H || x->W != dy->W) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3: x/dy grids differ");
  if (x->dtype != dy->dtype) INSAR_FAIL(INSAR_E_DTYPE, "insar_wgrad_conv3: x/dy dtype differ");
  const int pair = insar_wgrad_conv3_tile(x, dy->c_len);
  if (!pair) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3: unsupported layer (W %% 64 == 0, or W = 16 / 32 with whole K steps per image); use insar_wgrad");
  if (nsplit < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_wgrad_conv3: nsplit");
  Wgrad3Args a;
  a.x = (const char*)x->ptr; a.dy = (const char*)dy->ptr; a.part = part;
  a.ksteps = (long long)x->B * x->H * x->W / W3_BKP;
  a.nsplit = nsplit;
  a.steps_per_split = (int)((a.ksteps + nsplit - 1) / nsplit);
  a.H = x->H; a.W = x->W; a.Wp = x->W + 2;
  a.spr = x->W >= W3_BKP ? x->W / W3_BKP : 1;
  a.rpk = x->W >= W3_BKP ? 1 : W3_BKP / x->W;
  a.lw = x->W >= W3_BKP ? 6 : (x->W == 32 ? 5 : 4);
  a.Cx = x->C; a.cx_off = x->c_off; a.Cin = x->c_len;
  a.Cdy = dy->C; a.cdy_off = dy->c_off; a.Cout = dy->c_len;
  hipStream_t s = (hipStream_t)stream;
  const int tm = pair >> 16, tn = pair & 0xffff;
  if (x->dtype != INSAR_BF16) {          // fp32: 128 x 128 (8 waves) or 64 x 64, as insar_wgrad_tile_pair says
    if (tm == 128 && tn == 128) return launch_wgrad3<float, 128, 128, 8>(a, s);
    return launch_wgrad3<float, 64, 64, 4>(a, s);
  }
  if (tm == 128 && tn == 128)
    return insar_knob(KNOB_WGRAD3_M32) ? launch_wgrad3<bf16_t, 128, 128, 8, 32>(a, s) : launch_wgrad3<bf16_t, 128, 128, 8>(a, s);
  if (tm == 128) return launch_wgrad3<bf16_t, 128, 64, 4>(a, s);
  if (tn == 128) return launch_wgrad3<bf16_t, 64, 128, 4>(a, s);
  return launch_wgrad3<bf16_t, 64, 64, 4>(a, s);
}
