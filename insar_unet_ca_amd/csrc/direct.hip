// Direct (VALU, HBM-bound) kernels for the two layers whose channel count is too small for the
// matrix cores: the first 3x3 conv (Cin = in_channels <= 4, Unet-ChannalAttention.py:81 via :464)
// and outc, the 1x1 conv to num_classes (:125,162), forward and backward.
#include "common.h"

#define DR_THREADS 256
#define DR_MAXCI 4
#define DR_MAXK 8

template <typename T>
__device__ __forceinline__ float load_elem(const char* p) {
  if constexpr (sizeof(T) == 2) return bf16_to_f32(*(const uint16_t*)p);
  else return *(const float*)p;
}

// ---------------------------------------------------------------------------------------------
// first-layer conv forward: thread -> (pixel w, 16-byte chunk of output channels)
// stats: part[(n*H + h)][2][Co] partial sums of the stored output over the row.
// The thread's 9*CI*CH weights stay in registers for the whole grid-stride loop; the CI input channels
// of a tap are one 4- or 8-byte load when CI*sizeof(T) allows it.
// ---------------------------------------------------------------------------------------------
template <typename T, int CI>
__device__ __forceinline__ void load_pixel(const char* p, float (&v)[CI]) {
  if constexpr (sizeof(T) == 2 && CI == 2) {
    const uint32_t u = *(const uint32_t*)p;
    v[0] = bf16_to_f32((uint16_t)(u & 0xffffu)); v[1] = bf16_to_f32((uint16_t)(u >> 16));
  } else if constexpr (sizeof(T) == 2 && CI == 4) {
    const uint2 u = *(const uint2*)p;
    v[0] = bf16_to_f32((uint16_t)(u.x & 0xffffu)); v[1] = bf16_to_f32((uint16_t)(u.x >> 16));
    v[2] = bf16_to_f32((uint16_t)(u.y & 0xffffu)); v[3] = bf16_to_f32((uint16_t)(u.y >> 16));
  } else if constexpr (sizeof(T) == 4 && CI == 2) {
    const float2 u = *(const float2*)p; v[0] = u.x; v[1] = u.y;
  } else if constexpr (sizeof(T) == 4 && CI == 4) {
    const float4 u = *(const float4*)p; v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
  } else {
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) v[ci] = load_elem<T>(p + ci * sizeof(T));
  }
}

template <typename T, int CI>
__global__ void __launch_bounds__(DR_THREADS)
conv3x3_small_fwd_kernel(ActView x, const float* __restrict__ wt, ActView y, float* __restrict__ part) {
  constexpr int CH = Chunk<T>::N;
  __shared__ float red[DR_THREADS * (2 * CH + 1)];
  const int Co = y.c_len;
  const int cpp = Co / CH;
  const int rows = y.B * y.H;
  const int total = y.W * cpp;
  const int cc = threadIdx.x % cpp;                 // host guarantees blockDim % cpp == 0
  // weights and accumulators as float2 pairs: the multiply-adds become v_pk_fma_f32 (two lanes of work per
  // instruction; this kernel is VALU-bound: 9*CI*CH FMAs per 16 bytes stored)
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 wr[9][CI][CH / 2];                             // torch layout (Co, Ci, 3, 3)
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ci = 0; ci < CI; ++ci)
#pragma unroll
      for (int j = 0; j < CH / 2; ++j) {
        wr[tap][ci][j].x = wt[((int64_t)(cc * CH + 2 * j) * CI + ci) * 9 + tap];
        wr[tap][ci][j].y = wt[((int64_t)(cc * CH + 2 * j + 1) * CI + ci) * 9 + tap];
      }
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / y.H, h = r - n * y.H;
    float s1[CH], s2[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int w = e / cpp;
      f2 acc2[CH / 2];
#pragma unroll
      for (int j = 0; j < CH / 2; ++j) acc2[j] = (f2){0.f, 0.f};
      const char* p0 = x.base + x.elem_offset(n, h - 1, w - 1) * (int64_t)sizeof(T);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        float xv[CI];
        load_pixel<T, CI>(p0 + ((int64_t)(tap / 3) * (x.W + 2) + (tap % 3)) * x.C * (int64_t)sizeof(T), xv);
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) {
          const f2 xx = (f2){xv[ci], xv[ci]};
#pragma unroll
          for (int j = 0; j < CH / 2; ++j) acc2[j] = __builtin_elementwise_fma(xx, wr[tap][ci][j], acc2[j]);
        }
      }
      float acc[CH];
#pragma unroll
      for (int j = 0; j < CH / 2; ++j) { acc[2 * j] = acc2[j].x; acc[2 * j + 1] = acc2[j].y; }
      const uint4 packed = Chunk<T>::pack(acc);
      float f[CH];
      Chunk<T>::unpack(packed, f);                  // statistics of the stored (rounded) values
#pragma unroll
      for (int j = 0; j < CH; ++j) { s1[j] += f[j]; s2[j] = fmaf(f[j], f[j], s2[j]); }
      *(uint4*)(y.base + (y.elem_offset(n, h, w) + (int64_t)cc * CH) * (int64_t)sizeof(T)) = packed;
    }
    if (part) {
#pragma unroll
      for (int j = 0; j < CH; ++j) { red[threadIdx.x * (2 * CH + 1) + j] = s1[j]; red[threadIdx.x * (2 * CH + 1) + CH + j] = s2[j]; }
      __syncthreads();
      for (int o = threadIdx.x; o < 2 * Co; o += blockDim.x) {
        const int q = o / Co, c = o - q * Co;
        const int occ = c / CH, j = c - occ * CH;
        float s = 0.f;
        for (int t = occ; t < blockDim.x; t += cpp) s += red[t * (2 * CH + 1) + q * CH + j];
        part[((int64_t)r * 2 + q) * Co + c] = s;
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 first-layer forward on the matrix cores: K = 9*CI <= 32 fits ONE 16x16x32 MFMA step.
// A wave takes 16 consecutive pixels of an image row; lane (pixel r16, k group kq) gathers its 8 patch
// values k = 8*kq .. 8*kq+7 (k = tap*CI + ci; k >= 9*CI is zero) straight from global memory (L1 serves the
// overlap between neighbouring pixels), the weights of the four 16-channel output tiles stay in 32 VGPRs
// as MFMA A fragments (bf16 head + remainder), and every lane ends up with 4 consecutive channels of its pixel per tile: 8-byte
// stores. ~60 VGPRs instead of 208 for the VALU version, whose two waves per SIMD could not hide the
// load -> FMA -> store latency (98 us for a 134 MB write). Same per-image-row statistics slab.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) short d_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float d_f32x4_t;

template <int CI>
__global__ void __launch_bounds__(DR_THREADS) conv3x3_small_fwd_mfma_kernel(ActView x, const float* __restrict__ wt, ActView y,
                                                                           float* __restrict__ part) {
  constexpr int KV = 9 * CI;                        // valid k
  __shared__ float red[DR_THREADS / 64][2][64];     // per-wave channel sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int Co = y.c_len;                           // 64
  // A fragments: weights[cout = nt*16 + r16][k = 8*kq + i], torch layout (Co, CI, 3, 3) fp32 -> bf16
  // The fp32 master weight goes in as a bf16 head + bf16 remainder (two MFMAs per tile: ~16 mantissa bits,
  // as the VALU version's fp32 weights gave; the kernel is bound by its 134 MB of stores, not by the MFMAs).
  d_bf16x8_t wf[4], wl[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 8 * kq + i;
      float v = 0.f;
      if (k < KV) v = wt[((int64_t)(nt * 16 + r16) * CI + (k % CI)) * 9 + (k / CI)];
      const uint16_t hi = f32_to_bf16(v);
      wf[nt][i] = (short)hi;
      wl[nt][i] = (short)f32_to_bf16(v - bf16_to_f32(hi));
    }
  const int rows = y.B * y.H;
  const int groups = (y.W + 15) / 16;
  // BatchNorm partial sums of the stored values: carried in registers over all rows of this work-group,
  // written once: the slab has gridDim.x rows (insar_conv3x3_small_fwd_rows)
  float s1[4][4], s2[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[nt][j] = 0.f; s2[nt][j] = 0.f; }
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / y.H, h = r - n * y.H;
    for (int g = wave; g < groups; g += nw) {
      const int w = g * 16 + r16;
      const bool ok = w < y.W;
      const int wc = ok ? w : y.W - 1;              // clamp: lanes past the row end compute a duplicate, never store it
      d_bf16x8_t xf;
      const char* p0 = x.base + x.elem_offset(n, h - 1, wc - 1) * (int64_t)sizeof(bf16_t);
      if constexpr (CI == 2) {                      // both channels of a tap in one 4-byte load
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int tap = 4 * kq + i;
          uint32_t v = 0;
          if (tap < 9) v = *(const uint32_t*)(p0 + ((int64_t)(tap / 3) * (x.W + 2) + (tap % 3)) * x.C * 2);
          xf[2 * i] = (short)(v & 0xffffu); xf[2 * i + 1] = (short)(v >> 16);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = 8 * kq + i;
          short v = 0;
          if (k < KV) {
            const int tap = k / CI, ci = k - tap * CI;
            v = *(const short*)(p0 + (((int64_t)(tap / 3) * (x.W + 2) + (tap % 3)) * x.C + ci) * 2);
          }
          xf[i] = v;
        }
      }
      char* yp = y.base + (y.elem_offset(n, h, wc) + kq * 4) * (int64_t)2;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        d_f32x4_t acc = (d_f32x4_t){0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[nt], xf, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf, acc, 0, 0, 0);
        if (ok) {
          uint2 v;
          v.x = pack2_bf16(acc[0], acc[1]);
          v.y = pack2_bf16(acc[2], acc[3]);
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
  }
  if (part) {
    // lanes with equal kq hold the same channels: fold over r16, then over the waves
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int sh = 1; sh < 16; sh <<= 1) { s1[nt][j] += __shfl_xor(s1[nt][j], sh, 64); s2[nt][j] += __shfl_xor(s2[nt][j], sh, 64); }
        LDS_PIN(s1[nt][j]); LDS_PIN(s2[nt][j]);
      }
    if (r16 == 0) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          red[wave][0][nt * 16 + kq * 4 + j] = s1[nt][j];
          red[wave][1][nt * 16 + kq * 4 + j] = s2[nt][j];
        }
    }
    LDS_DRAIN();
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) { LDS_KEEP(s1[nt][j]); LDS_KEEP(s2[nt][j]); }
    __syncthreads();
    if (threadIdx.x < 2 * Co) {
      const int q = threadIdx.x / Co, c = threadIdx.x - q * Co;
      float v = 0.f;
      for (int wv = 0; wv < nw; ++wv) v += red[wv][q][c];
      part[((int64_t)blockIdx.x * 2 + q) * Co + c] = v;
    }
  }
}

static int check_small(const InsarAct* x, const InsarAct* y, const char* who) {
  if (!x || !y || !x->ptr || !y->ptr) INSAR_FAIL(INSAR_E_ARG, "%s: null pointer", who);
  if (x->dtype != y->dtype) INSAR_FAIL(INSAR_E_DTYPE, "%s: dtype differ", who);
  if (x->B != y->B || x->H != y->H || x->W != y->W) INSAR_FAIL(INSAR_E_SHAPE, "%s: grid differs", who);
  if (x->c_len < 1 || x->c_len > DR_MAXCI) INSAR_FAIL(INSAR_E_SHAPE, "%s: Cin=%d must be 1..%d", who, x->c_len, DR_MAXCI);
  const int ch = y->dtype == INSAR_BF16 ? 8 : 4;
  const int cpp = y->c_len / ch;
  if (y->c_len % ch || cpp < 1 || cpp > 64 || (cpp & (cpp - 1)) || (DR_THREADS % cpp))
    INSAR_FAIL(INSAR_E_SHAPE, "%s: Cout=%d unsupported", who, y->c_len);
  return insar_check_act(y, who, "y");
}

static inline bool small_fwd_uses_mfma(const InsarAct* x, const InsarAct* y) {
  return y->dtype == INSAR_BF16 && y->c_len == 64 && x->c_len <= 3;
}
// rows of the BatchNorm partial-sum slab insar_conv3x3_small_fwd writes: one per image row for the VALU
// version, one per work-group for the matrix-core version
extern "C" int insar_conv3x3_small_fwd_rows(const InsarAct* x, const InsarAct* y) {
  if (!x || !y) return 0;
  const int64_t r = (int64_t)y->B * y->H;
  if (small_fwd_uses_mfma(x, y)) return (int)(r > 1024 ? 1024 : r);
  return (int)r;
}

extern "C" int insar_conv3x3_small_fwd(const InsarAct* x, const float* w, const InsarAct* y, float* stats, void* stream) {
  int rc;
  if ((rc = check_small(x, y, "insar_conv3x3_small_fwd"))) return rc;
  if (!w) INSAR_FAIL(INSAR_E_ARG, "insar_conv3x3_small_fwd: null weights");
  if ((x->c_len == 2 || x->c_len == 4) && (x->C % x->c_len || x->c_off % x->c_len))
    INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_small_fwd: input slice must be aligned to its %d channels", x->c_len);
  int grid = insar_grid_cap((int64_t)y->B * y->H);
  hipStream_t s = (hipStream_t)stream;
  const ActView xv = make_view(*x), yv = make_view(*y);
#define SMALL_FWD(T, CI) hipLaunchKernelGGL((conv3x3_small_fwd_kernel<T, CI>), dim3(grid), dim3(DR_THREADS), 0, s, xv, w, yv, stats)
  if (small_fwd_uses_mfma(x, y)) {
    // matrix-core version (K = 9*Cin <= 32); its statistics slab has one row per work-group
    grid = insar_conv3x3_small_fwd_rows(x, y);
    switch (x->c_len) {
      case 1: hipLaunchKernelGGL((conv3x3_small_fwd_mfma_kernel<1>), dim3(grid), dim3(DR_THREADS), 0, s, xv, w, yv, stats); break;
      case 2: hipLaunchKernelGGL((conv3x3_small_fwd_mfma_kernel<2>), dim3(grid), dim3(DR_THREADS), 0, s, xv, w, yv, stats); break;
      default: hipLaunchKernelGGL((conv3x3_small_fwd_mfma_kernel<3>), dim3(grid), dim3(DR_THREADS), 0, s, xv, w, yv, stats); break;
    }
  } else if (y->dtype == INSAR_BF16) {
    switch (x->c_len) { case 1: SMALL_FWD(bf16_t, 1); break; case 2: SMALL_FWD(bf16_t, 2); break;
                        case 3: SMALL_FWD(bf16_t, 3); break; default: SMALL_FWD(bf16_t, 4); break; }
  } else {
    switch (x->c_len) { case 1: SMALL_FWD(float, 1); break; case 2: SMALL_FWD(float, 2); break;
                        case 3: SMALL_FWD(float, 3); break; default: SMALL_FWD(float, 4); break; }
  }
#undef SMALL_FWD
  INSAR_CHECK_LAUNCH("insar_conv3x3_small_fwd");
  return INSAR_OK;
}

// CI input values of one pixel as loaded (no conversion: a prefetched value must not be touched before its use, or the
// compiler waits for the load where it was issued)
template <typename T, int CI> struct RawPixel;
template <> struct RawPixel<bf16_t, 2> {
  uint32_t u;
  __device__ __forceinline__ void load(const char* p) { u = *(const uint32_t*)p; }
  __device__ __forceinline__ void get(float (&v)[2]) const { v[0] = __uint_as_float(u << 16); v[1] = __uint_as_float(u & 0xffff0000u); }
};
template <> struct RawPixel<bf16_t, 1> {
  uint16_t u;
  __device__ __forceinline__ void load(const char* p) { u = *(const uint16_t*)p; }
  __device__ __forceinline__ void get(float (&v)[1]) const { v[0] = bf16_to_f32(u); }
};
template <int CI> struct RawPixel<float, CI> {
  float f[CI];
  __device__ __forceinline__ void load(const char* p) {
#pragma unroll
    for (int c = 0; c < CI; ++c) f[c] = *(const float*)(p + 4 * c);
  }
  __device__ __forceinline__ void get(float (&v)[CI]) const {
#pragma unroll
    for (int c = 0; c < CI; ++c) v[c] = f[c];
  }
};

// ---------------------------------------------------------------------------------------------
// first-layer conv weight gradient: dW[co][ci][tap] = sum_pix x[pix+tap, ci] * dy[pix, co]
// Each block accumulates over its rows (grid-stride) in registers, reduces across the lanes that
// own the same channel chunk (wave shuffles, then LDS across waves) and writes ONE partial row:
// part[block][Co*Ci*9] in torch (Co,Ci,3,3) order. insar_colsum folds the blocks.
// ---------------------------------------------------------------------------------------------
template <typename T, int CIB>    // CIB input channels per pass over dy (2 for even Cin: dy is read once for Cin = 2)
__global__ void __launch_bounds__(DR_THREADS) conv3x3_small_wgrad_kernel(ActView x, ActView dy, float* __restrict__ part) {
  constexpr int CH = Chunk<T>::N;
  extern __shared__ float sm[];                     // [4 waves][cpp][9][CH]
  const int Ci = x.c_len, Co = dy.c_len;
  const int cpp = Co / CH;
  const int rows = dy.B * dy.H;
  const int total = dy.W * cpp;
  const int cc = threadIdx.x % cpp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int ci0 = 0; ci0 < Ci; ci0 += CIB) {
    typedef float wf2 __attribute__((ext_vector_type(2)));
    wf2 acc2[CIB][9][CH / 2];                        // channel pairs: the accumulation runs on v_pk_fma_f32
#pragma unroll
    for (int c = 0; c < CIB; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < CH / 2; ++j) acc2[c][t][j] = (wf2){0.f, 0.f};
    // One item = one 16-byte chunk of dy (a pixel's CH channels) with the 9 x CIB input values around that pixel. The kernel
    // is bound by its VALU instruction count (two waves a SIMD at ~200 VGPRs): the FMAs run packed (v_pk_fma_f32 on channel
    // pairs), the nine taps sit at uniform byte offsets from the centre pixel (one address per item, not nine), and the loads
    // of item i+1 are issued, raw, before the FMAs of item i (98 -> 89 us for the 134 MB of dy at B = 16; a three-deep queue
    // was slower: the compiler keeps it in AGPRs and pays the moves).
    const int64_t x_pix_bytes = (int64_t)x.C * (int64_t)sizeof(T), x_row_bytes = (int64_t)(x.W + 2) * x_pix_bytes;
    const int ipr = (total + blockDim.x - 1) / blockDim.x;                 // items per image row and thread
    const int nrow = (rows - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nit = nrow * ipr;
    uint4 gq = make_uint4(0u, 0u, 0u, 0u);
    RawPixel<T, CIB> xq[9];
    bool okq = false;
    auto fetch = [&](int k, int j) {                                       // k-th row of this block, j-th item of the row
      const int r = blockIdx.x + k * gridDim.x, e0 = threadIdx.x + j * blockDim.x;
      okq = e0 < total;
      const int e = okq ? e0 : threadIdx.x % cpp;          // beyond the row: load a valid item, drop it at use (no branch
      const int n = r / dy.H, h = r - n * dy.H, w = e / cpp; // around the loads: the counted waits need one load sequence)
      gq = *(const uint4*)(dy.base + (dy.elem_offset(n, h, w) + (int64_t)cc * CH) * (int64_t)sizeof(T));
      const char* xc = x.base + (x.elem_offset(n, h, w) + ci0) * (int64_t)sizeof(T);      // centre tap; the others at uniform offsets
#pragma unroll
      for (int t = 0; t < 9; ++t) xq[t].load(xc + (t / 3 - 1) * x_row_bytes + (t % 3 - 1) * x_pix_bytes);
    };
    int kq_ = 0, jq_ = 0;
    if (nit > 0) fetch(0, 0);
    for (int it = 0; it < nit; ++it) {
      const uint4 gv = gq;
      RawPixel<T, CIB> xr[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) xr[t] = xq[t];
      const bool ok = okq;
      if (++jq_ == ipr) { jq_ = 0; ++kq_; }
      fetch(it + 1 < nit ? kq_ : 0, it + 1 < nit ? jq_ : 0);   // past the end: re-load the first item (dropped)
      float g[CH];
      Chunk<T>::unpack(gv, g);
      wf2 g2[CH / 2];
#pragma unroll
      for (int j = 0; j < CH / 2; ++j) g2[j] = ok ? (wf2){g[2 * j], g[2 * j + 1]} : (wf2){0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        float xv[CIB];
        xr[t].get(xv);
#pragma unroll
        for (int c = 0; c < CIB; ++c) {
          const wf2 xx = (wf2){xv[c], xv[c]};
#pragma unroll
          for (int j = 0; j < CH / 2; ++j) acc2[c][t][j] = __builtin_elementwise_fma(xx, g2[j], acc2[c][t][j]);
        }
      }
    }
    float acc[CIB][9][CH];
#pragma unroll
    for (int c = 0; c < CIB; ++c)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < CH / 2; ++j) { acc[c][t][2 * j] = acc2[c][t][j].x; acc[c][t][2 * j + 1] = acc2[c][t][j].y; }
#pragma unroll
    for (int c = 0; c < CIB; ++c) {
      // lanes with equal (lane % cpp) own the same channels
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          float v = acc[c][t][j];
          for (int o = cpp; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
          acc[c][t][j] = v;
        }
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < CH; ++j) LDS_PIN(acc[c][t][j]);
      if (lane < cpp) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int j = 0; j < CH; ++j) sm[((wave * cpp + lane) * 9 + t) * CH + j] = acc[c][t][j];
      }
      LDS_DRAIN();
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < CH; ++j) LDS_KEEP(acc[c][t][j]);
      __syncthreads();
      for (int o = threadIdx.x; o < Co * 9; o += blockDim.x) {
        const int co = o / 9, t = o - co * 9;
        const int occ = co / CH, j = co - occ * CH;
        float s = 0.f;
        for (int w4 = 0; w4 < (int)(blockDim.x >> 6); ++w4) s += sm[((w4 * cpp + occ) * 9 + t) * CH + j];
        part[(int64_t)blockIdx.x * (Co * Ci * 9) + ((int64_t)co * Ci + ci0 + c) * 9 + t] = s;
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The same weight gradient on the matrix cores (bf16, Cin = 2, Cout = 64, W % 64 == 0: the U-Net's first layer,
// Unet-ChannalAttention.py:81 under loss.backward()). As a GEMM over pixels,
//     D[m][co] = sum_pix P[pix][m] * dY[pix][co],     m = 2 * tap + ci  (18 of 32 rows used),
// with P the im2col patch matrix, which is never in memory: every WAVE owns K steps of 64 consecutive pixels of one
// image row; per step it stages its dY tile (64 x 128 B) by LDS-DMA into one of two wave-private slots, writes the
// 64 x 32 patch tile from nine 4-byte loads a lane (lane = pixel) and feeds v_mfma_f32_16x16x32_bf16 through
// transposing LDS reads (the reduction dimension, pixels, is the row index of both tiles), as csrc/wgrad3.hip does.
// No work-group barrier in the loop (tiles are wave-private; a wave's own vmcnt / lgkmcnt waits order them); the VALU
// version spends 144 FMAs a chunk and is bound by them (90 us for 134 MB of dY), this one by HBM.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short d_s16x4_t;
#define SWM_WAVES 4
#define SWM_Y_TILE (64 * 128)        // dY tile: 64 pixel rows x 64 channels (bf16)
#define SWM_P_TILE (64 * 64)         // patch tile: 64 pixel rows x 32 columns (bf16)
#define SWM_WAVE_LDS (2 * SWM_Y_TILE + SWM_P_TILE)

struct SmallWgradMfmaArgs {
  const char* x; const char* dy; float* part;
  int H, W, Wp, spr;                 // spr = W / 64 K steps per image row
  int Cx, cx_off, Cdy, cdy_off;
  int ksteps;                        // B * H * spr
  // FUSED: dy is never in memory — the BatchNorm / ReLU backward apply pass of the unit (pointwise.hip,
  // bnrelu_bwd_apply_kernel, no SE gate) is evaluated on the way into the LDS tile from the unit's incoming gradient g
  // (a.dy, Cdy, cdy_off describe g) and its conv output y
  const char* y; int Cy, cy_off, relu;
  const float* scale; const float* shift; const float* mean; const float* invstd; const float* k1; const float* k2;
};

template <bool FUSED>
__global__ __launch_bounds__(SWM_WAVES * 64, 2) void conv3x3_small_wgrad_mfma_kernel(SmallWgradMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* sW = smem + wave * SWM_WAVE_LDS;                       // this wave's slots: Y0 | Y1 | P
  char* sP = sW + 2 * SWM_Y_TILE;
  const uint32_t ldsW = lds_offset_of(sW);
  const int r16 = lane & 15, kq = lane >> 4;

  // this wave's K steps: a contiguous range
  const int gw = blockIdx.x * SWM_WAVES + wave, GW = gridDim.x * SWM_WAVES;
  const int ks0 = (int)((long long)gw * a.ksteps / GW), ks1 = (int)((long long)(gw + 1) * a.ksteps / GW);

  // patch tile: zero once (columns 18..31 stay zero; 0..17 are rewritten every step)
  {
    uint4 z = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int c = 0; c < 4; ++c) *(uint4*)(sP + lane * 64 + c * 16) = z;
  }
  // padded index of the first pixel of K step ks
  auto first_pixel = [&](int ks) -> long long {
    const int g = ks / a.spr, seg = ks - g * a.spr;           // g = image row over the batch
    const int img = g / a.H, h = g - img * a.H;
    return ((long long)img * (a.H + 2) + h + 1) * a.Wp + seg * 64 + 1;
  };
  // dY tile by LDS-DMA: instruction i covers rows 8i .. 8i+7 (lane -> row 8i + lane/8, 16-byte chunk lane%8), XOR-swizzled
  // on the source side exactly as wgrad3.hip's 128-byte rows
  const int yrow = lane >> 3, ypos = lane & 7;
  const long long ypitch = (long long)a.Cdy * 2, xpitch = (long long)a.Cx * 2;
  const char* ybase = a.dy + (long long)a.cdy_off * 2;
  const char* xbase = a.x + (long long)a.cx_off * 2;
  auto stage_y = [&](int buf, long long p0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = i * 8 + yrow;
      lds_dma16_untracked(ybase + (p0 + row) * ypitch + ((ypos ^ (((row >> 1) & 3) << 1)) << 4), ldsW + buf * SWM_Y_TILE + i * 1024);
    }
  };
  // FUSED: this lane's eight (g, y) chunks of a step — rows 8i + yrow, the source chunk its LDS position holds (the XOR term
  // depends on yrow only, so the lane's eight channels and their constants are fixed) — and the apply pass's constants
  const int ychunk = ypos ^ (((yrow >> 1) & 3) << 1);
  uint4 vg[FUSED ? 8 : 1], vy[FUSED ? 8 : 1];
  float sc[FUSED ? 8 : 1], sh[FUSED ? 8 : 1], a0[FUSED ? 8 : 1], a1[FUSED ? 8 : 1];
  if constexpr (FUSED) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = ychunk * 8 + j;
      sc[j] = a.scale[c]; sh[j] = a.shift[c];
      a1[j] = -sc[j] * a.invstd[c] * a.k2[c];
      a0[j] = -sc[j] * a.k1[c] - a1[j] * a.mean[c];
    }
  }
  const long long gpitch = ypitch, y2pitch = (long long)a.Cy * 2;
  const char* y2base = FUSED ? a.y + (long long)a.cy_off * 2 : nullptr;
  auto load_gy = [&](long long p0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long pix = p0 + i * 8 + yrow;
      vg[i] = *(const uint4*)(ybase + pix * gpitch + (ychunk << 4));
      vy[i] = *(const uint4*)(y2base + pix * y2pitch + (ychunk << 4));
    }
  };
  auto write_dy = [&](int buf) {                      // dy = (mask ? g * scale : 0) + (a0 + a1 * y), rounded to bf16 as the pass stores it
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float f[8], gg[8], o[8];
      Chunk<bf16_t>::unpack(vy[i], f);
      Chunk<bf16_t>::unpack(vg[i], gg);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool on = !a.relu || fmaf(f[j], sc[j], sh[j]) > 0.f;
        const float ge = on ? fmaf(gg[j], sc[j], 0.f) : 0.f;
        o[j] = ge + fmaf(f[j], a1[j], a0[j]);
      }
      *(uint4*)(sW + buf * SWM_Y_TILE + i * 1024 + lane * 16) = Chunk<bf16_t>::pack(o);
    }
  };
  uint32_t xq[9];
  auto load_x = [&](long long p0) {
    const char* xc = xbase + (p0 + lane) * xpitch;            // this lane's pixel; taps at uniform offsets
#pragma unroll
    for (int t = 0; t < 9; ++t) xq[t] = *(const uint32_t*)(xc + ((t / 3 - 1) * a.Wp + (t % 3 - 1)) * xpitch);
  };

  d_f32x4_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (d_f32x4_t){0.f, 0.f, 0.f, 0.f};

  if (ks0 < ks1) {
    const long long p0 = first_pixel(ks0);
    if constexpr (FUSED) load_gy(p0); else stage_y(0, p0);
    load_x(p0);
  }
  const int pswz = (lane >> 2) & 3;                             // patch-tile swizzle of this lane's row (row = lane)
  for (int ks = ks0; ks < ks1; ++ks) {
    const int buf = (ks - ks0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this step's dY tile and x values have landed
    // patch row of this lane's pixel: dword t = (x[pix + tap t][0], x[..][1]) -> columns 2t, 2t+1; 16-byte chunks XOR-swizzled.
    // The x values are consumed BEFORE the next step's loads are issued: hipcc counts only its own loads, so a use after the
    // (inline-asm) LDS-DMA issue would make it wait for the next tile as well.
#pragma unroll
    for (int t = 0; t < 9; ++t) *(uint32_t*)(sP + lane * 64 + (((t >> 2) ^ pswz) << 4) + (t & 3) * 4) = xq[t];
    if constexpr (FUSED) write_dy(buf);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < ks1) {                                         // next step's loads fly during this step's LDS reads and MFMAs
      const long long p1 = first_pixel(ks + 1);
      if constexpr (FUSED) load_gy(p1); else stage_y(buf ^ 1, p1);
      load_x(p1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the tile is this wave's own: no barrier
    const char* sY = sW + buf * SWM_Y_TILE;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      d_bf16x8_t yf[4], pf[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int sel = h ^ (kq & 1);
        const int row = s2 * 32 + kq * 8 + sel * 4 + (r16 >> 2);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int colb = (nt * 16 + (r16 & 3) * 4) * 2;
          const char* q = sY + row * 128 + (((colb >> 4) ^ (((row >> 1) & 3) << 1)) << 4) + (colb & 15);
          d_s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) d_s16x4_t*)q);
          yf[nt][4 * h + 0] = v[0]; yf[nt][4 * h + 1] = v[1]; yf[nt][4 * h + 2] = v[2]; yf[nt][4 * h + 3] = v[3];
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int colb = (mt * 16 + (r16 & 3) * 4) * 2;
          const char* q = sP + row * 64 + (((colb >> 4) ^ ((row >> 2) & 3)) << 4) + (colb & 15);
          d_s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) d_s16x4_t*)q);
          pf[mt][4 * h + 0] = v[0]; pf[mt][4 * h + 1] = v[1]; pf[mt][4 * h + 2] = v[2]; pf[mt][4 * h + 3] = v[3];
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[mt], yf[nt], acc[mt][nt], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // reads done before the next step rewrites the patch tile
  }

  // fold the four waves' partial D (C layout: row m = kq*4 + reg, column co = r16) and write this block's row of `part`
  // in torch order: part[block][(co * 2 + ci) * 9 + tap], m = 2 * tap + ci
  __syncthreads();
  float* red = (float*)smem;                                    // [wave][m 32][co 64]
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[(wave * 32 + mt * 16 + kq * 4 + j) * 64 + nt * 16 + r16] = acc[mt][nt][j];
  __syncthreads();
  for (int o = threadIdx.x; o < 18 * 64; o += blockDim.x) {
    const int m = o / 64, co = o - m * 64;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < SWM_WAVES; ++w) v += red[(w * 32 + m) * 64 + co];
    a.part[(long long)blockIdx.x * (64 * 2 * 9) + (co * 2 + (m & 1)) * 9 + (m >> 1)] = v;
  }
}

static inline bool small_wgrad_uses_mfma(const InsarAct* x, const InsarAct* dy) {
  return dy->dtype == INSAR_BF16 && dy->c_len == 64 && x->c_len == 2 && (x->W % 64) == 0 && (x->C % 2) == 0 &&
         (x->c_off % 2) == 0 && (dy->C % 8) == 0 && (dy->c_off % 8) == 0;
}

extern "C" int insar_conv3x3_small_wgrad_blocks(int32_t B, int32_t H) {
  int64_t r = (int64_t)B * H;
  return (int)(r > 512 ? 512 : r);
}

extern "C" int insar_conv3x3_small_wgrad(const InsarAct* x, const InsarAct* dy, float* part, void* stream) {
  int rc;
  if ((rc = check_small(x, dy, "insar_conv3x3_small_wgrad"))) return rc;
  if (!part) INSAR_FAIL(INSAR_E_ARG, "insar_conv3x3_small_wgrad: null part");
  const int ch = dy->dtype == INSAR_BF16 ? 8 : 4;
  const int cpp = dy->c_len / ch;
  size_t lds = (size_t)4 * cpp * 9 * ch * sizeof(float);
  int grid = insar_conv3x3_small_wgrad_blocks(dy->B, dy->H);
  hipStream_t s = (hipStream_t)stream;
  if (small_wgrad_uses_mfma(x, dy)) {
    static std::atomic<uint64_t> attr_mask{0};     // per-device, see common.h
    const int lds_bytes = SWM_WAVES * SWM_WAVE_LDS;
    hipError_t e = insar_set_lds_once(attr_mask, (const void*)conv3x3_small_wgrad_mfma_kernel<false>, lds_bytes);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_conv3x3_small_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
    SmallWgradMfmaArgs a = {};
    a.x = (const char*)x->ptr; a.dy = (const char*)dy->ptr; a.part = part;
    a.H = x->H; a.W = x->W; a.Wp = x->W + 2; a.spr = x->W / 64;
    a.Cx = x->C; a.cx_off = x->c_off; a.Cdy = dy->C; a.cdy_off = dy->c_off;
    const long long ksteps = (long long)x->B * x->H * a.spr;
    if (ksteps > 0x7fffffffLL) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_small_wgrad: too many pixels");
    a.ksteps = (int)ksteps;
    hipLaunchKernelGGL(conv3x3_small_wgrad_mfma_kernel<false>, dim3(grid), dim3(SWM_WAVES * 64), lds_bytes, s, a);
    INSAR_CHECK_LAUNCH("insar_conv3x3_small_wgrad");
    return INSAR_OK;
  }
  const bool pair = (x->c_len % 2) == 0 && (x->C % 2) == 0 && (x->c_off % 2) == 0;      // aligned 2-channel loads
  if (dy->dtype == INSAR_BF16) {
    if (pair) hipLaunchKernelGGL((conv3x3_small_wgrad_kernel<bf16_t, 2>), dim3(grid), dim3(DR_THREADS), lds, s, make_view(*x), make_view(*dy), part);
    else hipLaunchKernelGGL((conv3x3_small_wgrad_kernel<bf16_t, 1>), dim3(grid), dim3(DR_THREADS), lds, s, make_view(*x), make_view(*dy), part);
  } else {
    if (pair) hipLaunchKernelGGL((conv3x3_small_wgrad_kernel<float, 2>), dim3(grid), dim3(DR_THREADS), lds, s, make_view(*x), make_view(*dy), part);
    else hipLaunchKernelGGL((conv3x3_small_wgrad_kernel<float, 1>), dim3(grid), dim3(DR_THREADS), lds, s, make_view(*x), make_view(*dy), part);
  }
  INSAR_CHECK_LAUNCH("insar_conv3x3_small_wgrad");
  return INSAR_OK;
}

// The same weight gradient with the unit's BatchNorm / ReLU backward apply pass evaluated on the way in: dy (which only this
// launch would read: the network's first layer has no input gradient) is never written. g: the unit's incoming gradient,
// y: its conv output (bf16, 64 channels each). Results are bit for bit those of insar_bnrelu_bwd_apply + insar_conv3x3_small_wgrad.
extern "C" int insar_conv3x3_small_wgrad_fused_ok(const InsarAct* x, const InsarAct* y) {
  if (!x || !y) return 0;
  return small_wgrad_uses_mfma(x, y) ? 1 : 0;
}

extern "C" int insar_conv3x3_small_wgrad_fused(const InsarAct* x, const InsarAct* g, const InsarAct* y, const float* scale,
                                               const float* shift, const float* mean, const float* invstd, const float* k1,
                                               const float* k2, int32_t relu, float* part, void* stream) {
  int rc;
  if ((rc = check_small(x, y, "insar_conv3x3_small_wgrad_fused"))) return rc;
  if (!g || !g->ptr || !part || !scale || !shift || !mean || !invstd || !k1 || !k2)
    INSAR_FAIL(INSAR_E_ARG, "insar_conv3x3_small_wgrad_fused: null pointer");
  if ((rc = insar_check_act(g, "insar_conv3x3_small_wgrad_fused", "g"))) return rc;
  if (!small_wgrad_uses_mfma(x, y) || !small_wgrad_uses_mfma(x, g) || g->B != y->B || g->H != y->H || g->W != y->W)
    INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_small_wgrad_fused: bf16, Cin = 2, Cout = 64, W %% 64 == 0, g and y on one grid");
  static std::atomic<uint64_t> attr_mask{0};     // per-device, see common.h
  const int lds_bytes = SWM_WAVES * SWM_WAVE_LDS;
  hipError_t e = insar_set_lds_once(attr_mask, (const void*)conv3x3_small_wgrad_mfma_kernel<true>, lds_bytes);
  if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_conv3x3_small_wgrad_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
  SmallWgradMfmaArgs a = {};
  a.x = (const char*)x->ptr; a.dy = (const char*)g->ptr; a.part = part;
  a.H = x->H; a.W = x->W; a.Wp = x->W + 2; a.spr = x->W / 64;
  a.Cx = x->C; a.cx_off = x->c_off; a.Cdy = g->C; a.cdy_off = g->c_off;
  a.y = (const char*)y->ptr; a.Cy = y->C; a.cy_off = y->c_off; a.relu = relu ? 1 : 0;
  a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.k1 = k1; a.k2 = k2;
  const long long ksteps = (long long)x->B * x->H * a.spr;
  if (ksteps > 0x7fffffffLL) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv3x3_small_wgrad_fused: too many pixels");
  a.ksteps = (int)ksteps;
  const int grid = insar_conv3x3_small_wgrad_blocks(y->B, y->H);
  hipLaunchKernelGGL(conv3x3_small_wgrad_mfma_kernel<true>, dim3(grid), dim3(SWM_WAVES * 64), lds_bytes, (hipStream_t)stream, a);
  INSAR_CHECK_LAUNCH("insar_conv3x3_small_wgrad_fused");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// outc forward: logits[n][k][h][w] = bias[k] + sum_c W[k][c] x[n,h,w,c]   (NCHW fp32 out)
// cpp consecutive lanes share a pixel (one 16-byte chunk each) and combine with xor-shuffles.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void conv1x1_out_fwd_kernel(ActView x, const float* __restrict__ wt, const float* __restrict__ bias,
                                       float* __restrict__ logits, int K) {
  constexpr int CH = Chunk<T>::N;
  extern __shared__ float sm[];                     // [K][C]
  const int C = x.c_len;
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) sm[i] = wt[i];
  __syncthreads();
  const int cpp = C / CH;
  const int rows = x.B * x.H;
  const int total = x.W * cpp;
  const int64_t HW = (int64_t)x.H * x.W;
  const int cc = threadIdx.x % cpp;
  const int tot_pad = (total + blockDim.x - 1) / blockDim.x * blockDim.x;   // keep whole waves in the shuffles
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / x.H, h = r - n * x.H;
    for (int e = threadIdx.x; e < tot_pad; e += blockDim.x) {
      const int w = e / cpp;
      const bool ok = e < total;
      float f[CH];
      if (ok) Chunk<T>::unpack(*(const uint4*)(x.base + (x.elem_offset(n, h, w) + (int64_t)cc * CH) * (int64_t)sizeof(T)), f);
      else {
#pragma unroll
        for (int j = 0; j < CH; ++j) f[j] = 0.f;
      }
      for (int k = 0; k < K; ++k) {
        float a = 0.f;
        const float* wr = sm + k * C + cc * CH;
#pragma unroll
        for (int j = 0; j < CH; ++j) a = fmaf(f[j], wr[j], a);
        for (int o = 1; o < cpp; o <<= 1) a += __shfl_xor(a, o, 64);
        if (ok && cc == 0) logits[((int64_t)n * K + k) * HW + (int64_t)h * x.W + w] = a + (bias ? bias[k] : 0.f);
      }
    }
  }
}

static int check_out(const InsarAct* x, int K, const char* who) {
  int rc;
  if ((rc = insar_check_act(x, who, "x"))) return rc;
  const int ch = x->dtype == INSAR_BF16 ? 8 : 4;
  const int cpp = x->c_len / ch;
  if (cpp < 1 || cpp > 64 || (cpp & (cpp - 1))) INSAR_FAIL(INSAR_E_SHAPE, "%s: C=%d unsupported", who, x->c_len);
  if (K < 1 || K > DR_MAXK) INSAR_FAIL(INSAR_E_SHAPE, "%s: num_classes=%d must be 1..%d", who, K, DR_MAXK);
  return INSAR_OK;
}

extern "C" int insar_conv1x1_out_fwd(const InsarAct* x, const float* w, const float* bias, float* logits, int32_t K, void* stream) {
  int rc;
  if ((rc = check_out(x, K, "insar_conv1x1_out_fwd"))) return rc;
  if (!w || !logits) INSAR_FAIL(INSAR_E_ARG, "insar_conv1x1_out_fwd: null pointer");
  size_t lds = (size_t)K * x->c_len * sizeof(float);
  int grid = insar_grid_cap((int64_t)x->B * x->H);
  hipStream_t s = (hipStream_t)stream;
  if (x->dtype == INSAR_BF16) hipLaunchKernelGGL(conv1x1_out_fwd_kernel<bf16_t>, dim3(grid), dim3(DR_THREADS), lds, s, make_view(*x), w, bias, logits, K);
  else hipLaunchKernelGGL(conv1x1_out_fwd_kernel<float>, dim3(grid), dim3(DR_THREADS), lds, s, make_view(*x), w, bias, logits, K);
  INSAR_CHECK_LAUNCH("insar_conv1x1_out_fwd");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// outc backward: dx[n,h,w,c] = sum_k dl[n,k,h,w] W[k][c];  dW[k][c] = sum dl*x;  db[k] = sum dl
// part[block][K*C + K] partial sums (one row per block), folded by insar_colsum.
// ---------------------------------------------------------------------------------------------
// FROMY: x holds the raw conv output y of the last unit and outc's input is recomputed,
//   z = round_T(relu(y*scale + shift) * gate[n]), instead of read (insar_bn_relu_apply_outc does not store it).
template <typename T, bool FROMY = false>
__global__ void conv1x1_out_bwd_kernel(ActView x, const float* __restrict__ wt, const float* __restrict__ dl,
                                       int K, ActView dx, float* __restrict__ part,
                                       const float* __restrict__ zscale = nullptr, const float* __restrict__ zshift = nullptr,
                                       const float* __restrict__ zgate = nullptr) {
  constexpr int CH = Chunk<T>::N;
  extern __shared__ float sm[];                     // [K][C] weights, then [4][K*C + K] partials
  const int C = x.c_len;
  float* sw = sm;
  float* sp = sm + K * C;
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) sw[i] = wt[i];
  __syncthreads();
  const int cpp = C / CH;
  const int rows = x.B * x.H;
  const int total = x.W * cpp;
  const int64_t HW = (int64_t)x.H * x.W;
  const int cc = threadIdx.x % cpp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float aw[DR_MAXK][CH], ab[DR_MAXK];
#pragma unroll
  for (int k = 0; k < DR_MAXK; ++k) {
    ab[k] = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) aw[k][j] = 0.f;
  }
  float zsc[CH], zsh[CH], zgt[CH];                   // FROMY: this thread's channel chunk never changes
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    zsc[j] = FROMY ? zscale[cc * CH + j] : 0.f; zsh[j] = FROMY ? zshift[cc * CH + j] : 0.f; zgt[j] = 1.f;
  }
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / x.H, h = r - n * x.H;
    if constexpr (FROMY) {
      if (zgate) {
#pragma unroll
        for (int j = 0; j < CH; ++j) zgt[j] = zgate[(int64_t)n * C + cc * CH + j];
      }
    }
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int w = e / cpp;
      float f[CH], o[CH];
      Chunk<T>::unpack(*(const uint4*)(x.base + (x.elem_offset(n, h, w) + (int64_t)cc * CH) * (int64_t)sizeof(T)), f);
      if constexpr (FROMY) {
        float zz[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) zz[j] = fmaxf(fmaf(f[j], zsc[j], zsh[j]), 0.f) * zgt[j];
        Chunk<T>::unpack(Chunk<T>::pack(zz), f);
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) o[j] = 0.f;
#pragma unroll
      for (int k = 0; k < DR_MAXK; ++k) {
        if (k < K) {
          const float g = dl[((int64_t)n * K + k) * HW + (int64_t)h * x.W + w];
          const float* wr = sw + k * C + cc * CH;
#pragma unroll
          for (int j = 0; j < CH; ++j) { o[j] = fmaf(g, wr[j], o[j]); aw[k][j] = fmaf(g, f[j], aw[k][j]); }
          if (cc == 0) ab[k] += g;
        }
      }
      if (dx.base) *(uint4*)(dx.base + (dx.elem_offset(n, h, w) + (int64_t)cc * CH) * (int64_t)sizeof(T)) = Chunk<T>::pack(o);
    }
  }
  const int pw = K * C + K;
#pragma unroll
  for (int k = 0; k < DR_MAXK; ++k) {
    if (k < K) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        float v = aw[k][j];
        for (int o = cpp; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
        aw[k][j] = v;
        LDS_PIN(aw[k][j]);
      }
      float b = ab[k];
      for (int o = 1; o < 64; o <<= 1) b += __shfl_xor(b, o, 64);
      ab[k] = b;
      LDS_PIN(ab[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < DR_MAXK; ++k) {
    if (k < K) {
      if (lane < cpp) {
#pragma unroll
        for (int j = 0; j < CH; ++j) sp[wave * pw + k * C + lane * CH + j] = aw[k][j];
      }
      if (lane == 0) sp[wave * pw + K * C + k] = ab[k];
    }
  }
  LDS_DRAIN();
#pragma unroll
  for (int k = 0; k < DR_MAXK; ++k) {
    LDS_KEEP(ab[k]);
#pragma unroll
    for (int j = 0; j < CH; ++j) LDS_KEEP(aw[k][j]);
  }
  __syncthreads();
  for (int o = threadIdx.x; o < pw; o += blockDim.x) {
    float s = 0.f;
    for (int w4 = 0; w4 < (int)(blockDim.x >> 6); ++w4) s += sp[w4 * pw + o];
    part[(int64_t)blockIdx.x * pw + o] = s;
  }
}

extern "C" int insar_conv1x1_out_bwd_blocks(int32_t B, int32_t H) {
  int64_t r = (int64_t)B * H;
  return (int)(r > 1024 ? 1024 : r);
}

extern "C" int insar_conv1x1_out_bwd(const InsarAct* x, const float* w, const float* dlogits, int32_t K,
                                     const InsarAct* dx, float* part, void* stream) {
  int rc;
  if ((rc = check_out(x, K, "insar_conv1x1_out_bwd"))) return rc;
  if ((rc = insar_check_act(dx, "insar_conv1x1_out_bwd", "dx"))) return rc;
  if (!w || !dlogits || !part) INSAR_FAIL(INSAR_E_ARG, "insar_conv1x1_out_bwd: null pointer");
  if (x->B != dx->B || x->H != dx->H || x->W != dx->W || x->c_len != dx->c_len || x->dtype != dx->dtype)
    INSAR_FAIL(INSAR_E_SHAPE, "insar_conv1x1_out_bwd: x/dx mismatch");
  size_t lds = (size_t)(K * x->c_len + 4 * (K * x->c_len + K)) * sizeof(float);
  int grid = insar_conv1x1_out_bwd_blocks(x->B, x->H);
  hipStream_t s = (hipStream_t)stream;
  if (x->dtype == INSAR_BF16) hipLaunchKernelGGL(conv1x1_out_bwd_kernel<bf16_t>, dim3(grid), dim3(DR_THREADS), lds, s, make_view(*x), w, dlogits, K, make_view(*dx), part);
  else hipLaunchKernelGGL(conv1x1_out_bwd_kernel<float>, dim3(grid), dim3(DR_THREADS), lds, s, make_view(*x), w, dlogits, K, make_view(*dx), part);
  INSAR_CHECK_LAUNCH("insar_conv1x1_out_bwd");
  return INSAR_OK;
}

// Parameter gradients only (part as above), for callers that recompute the input gradient where it is consumed
// (insar_bnrelu_bwd_reduce_outc / insar_bnrelu_bwd_apply_outc): no 64-channel gradient tensor is written.
extern "C" int insar_conv1x1_out_wgrad(const InsarAct* x, const float* w, const float* dlogits, int32_t K, float* part,
                                       void* stream) {
  int rc;
  if ((rc = check_out(x, K, "insar_conv1x1_out_wgrad"))) return rc;
  if (!w || !dlogits || !part) INSAR_FAIL(INSAR_E_ARG, "insar_conv1x1_out_wgrad: null pointer");
  size_t lds = (size_t)(K * x->c_len + 4 * (K * x->c_len + K)) * sizeof(float);
  int grid = insar_conv1x1_out_bwd_blocks(x->B, x->H);
  hipStream_t s = (hipStream_t)stream;
  ActView none = make_view(*x); none.base = nullptr;
  if (x->dtype == INSAR_BF16) hipLaunchKernelGGL(conv1x1_out_bwd_kernel<bf16_t>, dim3(grid), dim3(DR_THREADS), lds, s, make_view(*x), w, dlogits, K, none, part);
  else hipLaunchKernelGGL(conv1x1_out_bwd_kernel<float>, dim3(grid), dim3(DR_THREADS), lds, s, make_view(*x), w, dlogits, K, none, part);
  INSAR_CHECK_LAUNCH("insar_conv1x1_out_wgrad");
  return INSAR_OK;
}

// The same with outc's input recomputed from the last unit's raw conv output: z = round(relu(y*scale+shift) * gate[n])
// (gate nullable), the companion of insar_bn_relu_apply_outc.
extern "C" int insar_conv1x1_out_wgrad_y(const InsarAct* y, const float* scale, const float* shift, const float* gate,
                                         const float* w, const float* dlogits, int32_t K, float* part, void* stream) {
  int rc;
  if ((rc = check_out(y, K, "insar_conv1x1_out_wgrad_y"))) return rc;
  if (!w || !dlogits || !part || !scale || !shift) INSAR_FAIL(INSAR_E_ARG, "insar_conv1x1_out_wgrad_y: null pointer");
  size_t lds = (size_t)(K * y->c_len + 4 * (K * y->c_len + K)) * sizeof(float);
  int grid = insar_conv1x1_out_bwd_blocks(y->B, y->H);
  hipStream_t s = (hipStream_t)stream;
  ActView none = make_view(*y); none.base = nullptr;
  if (y->dtype == INSAR_BF16) hipLaunchKernelGGL((conv1x1_out_bwd_kernel<bf16_t, true>), dim3(grid), dim3(DR_THREADS), lds, s, make_view(*y), w, dlogits, K, none, part, scale, shift, gate);
  else hipLaunchKernelGGL((conv1x1_out_bwd_kernel<float, true>), dim3(grid), dim3(DR_THREADS), lds, s, make_view(*y), w, dlogits, K, none, part, scale, shift, gate);
  INSAR_CHECK_LAUNCH("insar_conv1x1_out_wgrad_y");
  return INSAR_OK;
}
