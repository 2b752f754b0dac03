// Kernels that config 5 (DeepLabV3-CA, /root/reference/DeepLabV3-ChannelAttention.py:83-162) needs on top of the
// U-Net-CA set. The network's arithmetic lives in torchvision (models/resnet.py, models/segmentation/deeplabv3.py,
// absent from the build container: SURVEY 8c), re-stated here from its published architecture:
//   * stem: Conv2d(1, 64, 7, stride 2, padding 3, bias=False) (:105-118 swaps the 3-channel stem for this one),
//     MaxPool2d(3, stride 2, padding 1);
//   * Bottleneck residual: out = relu(bn3(conv3(.)) + identity);
//   * ASPP image-pooling branch: AdaptiveAvgPool2d(1) -> 1x1 conv -> BN -> ReLU -> bilinear up (= broadcast of a 1x1 map);
//   * Dropout(0.5) after the ASPP projection;
//   * F_T.resize(logits, input_shape, BILINEAR) (:160) = bilinear, align_corners=False.
// Every 1x1 / 3x3 / strided / dilated convolution of the backbone and head runs on the implicit-GEMM kernels
// (igemm.hip with INSAR_IGEMM_OOB_ZERO, wgrad.hip with per-tap pixel tables built here).
// All kernels are HBM- or latency-bound helpers: 16-byte accesses, one block per image row, deterministic sums.
#include "common.h"

#define DL_THREADS 256

template <typename T>
__device__ __forceinline__ const uint4* dl_chunk(const ActView& v, int n, int h, int w, int cc) {
  return (const uint4*)(v.base + (v.elem_offset(n, h, w) + (int64_t)cc * Chunk<T>::N) * (int64_t)sizeof(T));
}
template <typename T>
__device__ __forceinline__ uint4* dl_chunk_w(const ActView& v, int n, int h, int w, int cc) {
  return (uint4*)(v.base + (v.elem_offset(n, h, w) + (int64_t)cc * Chunk<T>::N) * (int64_t)sizeof(T));
}
template <typename T> __device__ __forceinline__ float dl_round(float f);
template <> __device__ __forceinline__ float dl_round<float>(float f) { return f; }
template <> __device__ __forceinline__ float dl_round<bf16_t>(float f) { return bf16_to_f32(f32_to_bf16(f)); }

static int dl_same_grid(const InsarAct* a, const InsarAct* b, const char* who) {
  if (a->B != b->B || a->H != b->H || a->W != b->W || a->c_len != b->c_len || a->dtype != b->dtype)
    INSAR_FAIL(INSAR_E_SHAPE, "%s: operands differ in shape or dtype", who);
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// per-tap pixel tables for insar_wgrad: tab[t][p] = padded pixel index of (n, ho*s + dy[t], wo*s + dx[t]) in a buffer
// of interior (Hb, Wb), or 0 (the zero halo corner) when that position lies outside the padded buffer or p >= M.
// ---------------------------------------------------------------------------------------------------------
struct TapList { int n; signed char dy[12], dx[12]; };

__global__ void pixel_table_taps_kernel(int32_t* tab, int64_t Mpad, int B, int H, int W, int s, int Hb, int Wb, TapList taps) {
  const int64_t M = (int64_t)B * H * W;
  const int64_t total = Mpad * taps.n;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(q / Mpad);
    const int64_t p = q - (int64_t)t * Mpad;
    int32_t v = 0;
    if (p < M) {
      const int n = (int)(p / ((int64_t)H * W));
      const int64_t rem = p - (int64_t)n * H * W;
      const int h = (int)(rem / W), w = (int)(rem - (int64_t)h * W);
      const int hh = h * s + taps.dy[t] + 1, ww = w * s + taps.dx[t] + 1;       // padded coordinates
      if (hh >= 0 && hh <= Hb + 1 && ww >= 0 && ww <= Wb + 1) v = (int32_t)(((int64_t)n * (Hb + 2) + hh) * (Wb + 2) + ww);
    }
    tab[q] = v;
  }
}

extern "C" int insar_pixel_table_taps(int32_t* tab, int64_t Mpad, int32_t B, int32_t H, int32_t W, int32_t s, int32_t Hb,
                                      int32_t Wb, int32_t ntaps, const int8_t* dy, const int8_t* dx, void* stream) {
  if (!tab || !dy || !dx) INSAR_FAIL(INSAR_E_ARG, "insar_pixel_table_taps: null pointer");
  if (ntaps < 1 || ntaps > 12) INSAR_FAIL(INSAR_E_SHAPE, "insar_pixel_table_taps: ntaps=%d", ntaps);
  if ((int64_t)B * (Hb + 2) * (Wb + 2) > 0x7fffffffLL) INSAR_FAIL(INSAR_E_SHAPE, "insar_pixel_table_taps: pixel index overflows int32");
  if (Mpad < (int64_t)B * H * W) INSAR_FAIL(INSAR_E_SHAPE, "insar_pixel_table_taps: Mpad smaller than the grid");
  TapList t; t.n = ntaps;
  for (int i = 0; i < 12; ++i) { t.dy[i] = i < ntaps ? dy[i] : 0; t.dx[i] = i < ntaps ? dx[i] : 0; }
  int grid = insar_grid_cap((Mpad * ntaps + 255) / 256);
  hipLaunchKernelGGL(pixel_table_taps_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, tab, Mpad, B, H, W, s, Hb, Wb, t);
  INSAR_CHECK_LAUNCH("insar_pixel_table_taps");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// stem: y = conv7x7(x, stride 2, padding 3), 1 -> 64 channels, no bias. x is the module's input itself
// (NCHW fp32, one channel); operands are rounded to the compute type, sums are fp32. One block per output row;
// thread = (pixel lane, 8-channel group). stats[row][2][64]: BatchNorm partial sums of the stored values.
// ---------------------------------------------------------------------------------------------------------
#define ST_K 7
#define ST_CO 64
template <typename T>
__global__ void __launch_bounds__(DL_THREADS) stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, ActView y,
                                                              float* __restrict__ stats, int H, int W) {
  extern __shared__ float sm[];
  float* sW = sm;                         // [49][64]
  const int inw = 2 * y.W + 5;
  float* sIn = sm + 49 * ST_CO;           // [7][inw]
  float* sRed = sIn + 7 * inw;            // [32][16 floats x 8 groups] reduction scratch
  for (int i = threadIdx.x; i < 49 * ST_CO; i += blockDim.x) {
    const int tap = i / ST_CO, co = i - tap * ST_CO;
    sW[i] = dl_round<T>(w[co * 49 + tap]);
  }
  const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;      // 8 channel groups x 32 pixel lanes
  for (int r = blockIdx.x; r < y.B * y.H; r += gridDim.x) {
    const int n = r / y.H, ho = r - n * y.H;
    __syncthreads();
    for (int i = threadIdx.x; i < 7 * inw; i += blockDim.x) {
      const int ky = i / inw, c = i - ky * inw;
      const int hi = 2 * ho + ky - 3, wi = c - 3;
      float v = 0.f;
      if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = dl_round<T>(x[((int64_t)n * H + hi) * W + wi]);
      sIn[i] = v;
    }
    __syncthreads();
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    for (int wo = pl; wo < y.W; wo += 32) {
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
      for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const float xv = sIn[ky * inw + 2 * wo + kx];
          const float* wr = sW + (ky * 7 + kx) * ST_CO + cg * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv, wr[j], acc[j]);
        }
      if constexpr (sizeof(T) == 2) {
        *dl_chunk_w<T>(y, n, ho, wo, cg) = Chunk<bf16_t>::pack(acc);
      } else {
        *dl_chunk_w<T>(y, n, ho, wo, 2 * cg) = Chunk<float>::pack(acc);
        *dl_chunk_w<T>(y, n, ho, wo, 2 * cg + 1) = Chunk<float>::pack(acc + 4);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = dl_round<T>(acc[j]); s1[j] += f; s2[j] = fmaf(f, f, s2[j]); }
    }
    if (stats) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { sRed[(pl * 8 + cg) * 16 + j] = s1[j]; sRed[(pl * 8 + cg) * 16 + 8 + j] = s2[j]; }
      __syncthreads();
      if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, co = threadIdx.x & 63;     // 0: sum, 1: sum of squares
        float v = 0.f;
        for (int p = 0; p < 32; ++p) v += sRed[(p * 8 + (co >> 3)) * 16 + which * 8 + (co & 7)];
        stats[((int64_t)r * 2 + which) * ST_CO + co] = v;
      }
    }
  }
}

extern "C" int insar_conv7x7s2_fwd_rows(int32_t B, int32_t H) { return B * ((H + 1) / 2); }

extern "C" int insar_conv7x7s2_fwd(const float* x, int32_t H, int32_t W, const float* w, const InsarAct* y, float* stats, void* stream) {
  int rc;
  if (!x || !w) INSAR_FAIL(INSAR_E_ARG, "insar_conv7x7s2_fwd: null pointer");
  if ((rc = insar_check_act(y, "insar_conv7x7s2_fwd", "y"))) return rc;
  if (H % 2 || W % 2 || y->H != H / 2 || y->W != W / 2 || y->c_len != ST_CO)
    INSAR_FAIL(INSAR_E_SHAPE, "insar_conv7x7s2_fwd: needs even H, W and a (B, H/2, W/2, 64) output slice");
  const int inw = 2 * y->W + 5;
  const size_t lds = (size_t)(49 * ST_CO + 7 * inw + 256 * 16) * sizeof(float);
  if (lds > 160 * 1024) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv7x7s2_fwd: W=%d too wide", W);
  int grid = insar_grid_cap((int64_t)y->B * y->H);
  hipStream_t s = (hipStream_t)stream;
  if (y->dtype == INSAR_BF16) {
    static std::atomic<uint64_t> m{0};
    hipError_t e = insar_set_lds_once(m, (const void*)stem_fwd_kernel<bf16_t>, 160 * 1024);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_conv7x7s2_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(stem_fwd_kernel<bf16_t>, dim3(grid), dim3(DL_THREADS), lds, s, x, w, make_view(*y), stats, H, W);
  } else {
    static std::atomic<uint64_t> m{0};
    hipError_t e = insar_set_lds_once(m, (const void*)stem_fwd_kernel<float>, 160 * 1024);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_conv7x7s2_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(stem_fwd_kernel<float>, dim3(grid), dim3(DL_THREADS), lds, s, x, w, make_view(*y), stats, H, W);
  }
  INSAR_CHECK_LAUNCH("insar_conv7x7s2_fwd");
  return INSAR_OK;
}

// weight gradient of the stem: part[block][64*49] (torch (64,1,7,7) order), block = ST_RPB consecutive output rows;
// thread = (output channel, every 4th tap). Folded by insar_colsum.
#define ST_RPB 8
template <typename T>
__global__ void __launch_bounds__(DL_THREADS) stem_wgrad_kernel(const float* __restrict__ x, ActView dy, float* __restrict__ part,
                                                                int H, int W) {
  extern __shared__ float sm[];
  const int inw = 2 * dy.W + 5;
  float* sIn = sm;                        // [7][inw]
  float* sDy = sm + 7 * inw;              // [W][64]
  const int co = threadIdx.x >> 2, tq = threadIdx.x & 3;
  float acc[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) acc[i] = 0.f;
  const int rows = dy.B * dy.H;
  const int r0 = blockIdx.x * ST_RPB;
  for (int r = r0; r < r0 + ST_RPB && r < rows; ++r) {
    const int n = r / dy.H, ho = r - n * dy.H;
    __syncthreads();
    for (int i = threadIdx.x; i < 7 * inw; i += blockDim.x) {
      const int ky = i / inw, c = i - ky * inw;
      const int hi = 2 * ho + ky - 3, wi = c - 3;
      float v = 0.f;
      if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = dl_round<T>(x[((int64_t)n * H + hi) * W + wi]);
      sIn[i] = v;
    }
    constexpr int CH = Chunk<T>::N;
    const int cpp = ST_CO / CH;
    for (int e = threadIdx.x; e < dy.W * cpp; e += blockDim.x) {
      const int wo = e / cpp, cc = e - wo * cpp;
      float f[CH];
      Chunk<T>::unpack(*dl_chunk<T>(dy, n, ho, wo, cc), f);
#pragma unroll
      for (int j = 0; j < CH; ++j) sDy[wo * ST_CO + cc * CH + j] = f[j];
    }
    __syncthreads();
    for (int wo = 0; wo < dy.W; ++wo) {
      const float g = sDy[wo * ST_CO + co];
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        const int tap = tq + 4 * i;
        if (tap < 49) acc[i] = fmaf(g, sIn[(tap / 7) * inw + 2 * wo + (tap % 7)], acc[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 13; ++i) {
    const int tap = tq + 4 * i;
    if (tap < 49) part[(int64_t)blockIdx.x * (ST_CO * 49) + co * 49 + tap] = acc[i];
  }
}

extern "C" int insar_conv7x7s2_wgrad_blocks(int32_t B, int32_t Ho) { return (B * Ho + ST_RPB - 1) / ST_RPB; }

extern "C" int insar_conv7x7s2_wgrad(const float* x, int32_t H, int32_t W, const InsarAct* dy, float* part, void* stream) {
  int rc;
  if (!x || !part) INSAR_FAIL(INSAR_E_ARG, "insar_conv7x7s2_wgrad: null pointer");
  if ((rc = insar_check_act(dy, "insar_conv7x7s2_wgrad", "dy"))) return rc;
  if (H % 2 || W % 2 || dy->H != H / 2 || dy->W != W / 2 || dy->c_len != ST_CO)
    INSAR_FAIL(INSAR_E_SHAPE, "insar_conv7x7s2_wgrad: needs even H, W and a (B, H/2, W/2, 64) gradient slice");
  const int inw = 2 * dy->W + 5;
  const size_t lds = (size_t)(7 * inw + dy->W * ST_CO) * sizeof(float);
  if (lds > 160 * 1024) INSAR_FAIL(INSAR_E_SHAPE, "insar_conv7x7s2_wgrad: W=%d too wide", W);
  const int grid = insar_conv7x7s2_wgrad_blocks(dy->B, dy->H);
  hipStream_t s = (hipStream_t)stream;
  if (dy->dtype == INSAR_BF16) {
    static std::atomic<uint64_t> m{0};
    hipError_t e = insar_set_lds_once(m, (const void*)stem_wgrad_kernel<bf16_t>, 160 * 1024);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_conv7x7s2_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(stem_wgrad_kernel<bf16_t>, dim3(grid), dim3(DL_THREADS), lds, s, x, make_view(*dy), part, H, W);
  } else {
    static std::atomic<uint64_t> m{0};
    hipError_t e = insar_set_lds_once(m, (const void*)stem_wgrad_kernel<float>, 160 * 1024);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_conv7x7s2_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(stem_wgrad_kernel<float>, dim3(grid), dim3(DL_THREADS), lds, s, x, make_view(*dy), part, H, W);
  }
  INSAR_CHECK_LAUNCH("insar_conv7x7s2_wgrad");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, padding 1): forward keeps the window position (ky*3+kx, first maximum in scan order,
// NaN wins: torch's rule) of every output; backward gathers, for every input pixel, the <= 4 windows that contain it.
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool3s2_fwd_kernel(ActView x, ActView y, uint8_t* __restrict__ arg) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = y.c_len / CH;
  const int total = y.W * cpp;
  for (int r = blockIdx.x; r < y.B * y.H; r += gridDim.x) {
    const int n = r / y.H, ho = r - n * y.H;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int wo = e / cpp, cc = e - wo * cpp;
      float m[CH];
      int am[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) { m[j] = -INFINITY; am[j] = -1; }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int h = 2 * ho + ky - 1, w = 2 * wo + kx - 1;
          const bool valid = (unsigned)h < (unsigned)x.H && (unsigned)w < (unsigned)x.W;
          float f[CH];
          Chunk<T>::unpack(*dl_chunk<T>(x, n, valid ? h : 0, valid ? w : 0, cc), f);      // clamped address, predicated use
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const bool take = valid && (am[j] < 0 || f[j] > m[j] || (f[j] != f[j] && m[j] == m[j]));
            m[j] = take ? f[j] : m[j];
            am[j] = take ? ky * 3 + kx : am[j];
          }
        }
      *dl_chunk_w<T>(y, n, ho, wo, cc) = Chunk<T>::pack(m);
      uint8_t* ap = arg + (((int64_t)n * y.H + ho) * y.W + wo) * y.c_len + cc * CH;
#pragma unroll
      for (int j = 0; j < CH; ++j) ap[j] = (uint8_t)am[j];
    }
  }
}

template <typename T>
__global__ void maxpool3s2_bwd_kernel(ActView dy, const uint8_t* __restrict__ arg, ActView dx) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = dx.c_len / CH;
  const int total = dx.W * cpp;
  for (int r = blockIdx.x; r < dx.B * dx.H; r += gridDim.x) {
    const int n = r / dx.H, h = r - n * dx.H;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int w = e / cpp, cc = e - w * cpp;
      float o[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) o[j] = 0.f;
      // windows ho with 2ho-1 <= h <= 2ho+1
      const int ho0 = (h & 1) ? (h - 1) / 2 : h / 2, ho1 = (h & 1) ? (h + 1) / 2 : h / 2;
      const int wo0 = (w & 1) ? (w - 1) / 2 : w / 2, wo1 = (w & 1) ? (w + 1) / 2 : w / 2;
      for (int ho = ho0; ho <= ho1; ++ho) {
        if (ho >= dy.H) continue;
        for (int wo = wo0; wo <= wo1; ++wo) {
          if (wo >= dy.W) continue;
          const int pos = (h - 2 * ho + 1) * 3 + (w - 2 * wo + 1);
          float g[CH];
          Chunk<T>::unpack(*dl_chunk<T>(dy, n, ho, wo, cc), g);
          const uint8_t* ap = arg + (((int64_t)n * dy.H + ho) * dy.W + wo) * dy.c_len + cc * CH;
#pragma unroll
          for (int j = 0; j < CH; ++j) if (ap[j] == pos) o[j] += g[j];
        }
      }
      *dl_chunk_w<T>(dx, n, h, w, cc) = Chunk<T>::pack(o);
    }
  }
}

static int check_pool3(const InsarAct* x, const InsarAct* y, const char* who) {
  if (x->B != y->B || x->c_len != y->c_len || x->dtype != y->dtype || y->H != (x->H + 1) / 2 || y->W != (x->W + 1) / 2)
    INSAR_FAIL(INSAR_E_SHAPE, "%s: output must be (B, ceil(H/2), ceil(W/2), C) of the input", who);
  return INSAR_OK;
}

extern "C" int insar_maxpool3s2_fwd(const InsarAct* x, const InsarAct* y, uint8_t* arg, void* stream) {
  int rc;
  if ((rc = insar_check_act(x, "insar_maxpool3s2_fwd", "x"))) return rc;
  if ((rc = insar_check_act(y, "insar_maxpool3s2_fwd", "y"))) return rc;
  if (!arg) INSAR_FAIL(INSAR_E_ARG, "insar_maxpool3s2_fwd: null arg map");
  if ((rc = check_pool3(x, y, "insar_maxpool3s2_fwd"))) return rc;
  int grid = insar_grid_cap((int64_t)y->B * y->H);
  hipStream_t s = (hipStream_t)stream;
  if (x->dtype == INSAR_BF16) hipLaunchKernelGGL(maxpool3s2_fwd_kernel<bf16_t>, dim3(grid), dim3(DL_THREADS), 0, s, make_view(*x), make_view(*y), arg);
  else hipLaunchKernelGGL(maxpool3s2_fwd_kernel<float>, dim3(grid), dim3(DL_THREADS), 0, s, make_view(*x), make_view(*y), arg);
  INSAR_CHECK_LAUNCH("insar_maxpool3s2_fwd");
  return INSAR_OK;
}

extern "C" int insar_maxpool3s2_bwd(const InsarAct* dy, const uint8_t* arg, const InsarAct* dx, void* stream) {
  int rc;
  if ((rc = insar_check_act(dy, "insar_maxpool3s2_bwd", "dy"))) return rc;
  if ((rc = insar_check_act(dx, "insar_maxpool3s2_bwd", "dx"))) return rc;
  if (!arg) INSAR_FAIL(INSAR_E_ARG, "insar_maxpool3s2_bwd: null arg map");
  if ((rc = check_pool3(dx, dy, "insar_maxpool3s2_bwd"))) return rc;
  int grid = insar_grid_cap((int64_t)dx->B * dx->H);
  hipStream_t s = (hipStream_t)stream;
  if (dx->dtype == INSAR_BF16) hipLaunchKernelGGL(maxpool3s2_bwd_kernel<bf16_t>, dim3(grid), dim3(DL_THREADS), 0, s, make_view(*dy), arg, make_view(*dx));
  else hipLaunchKernelGGL(maxpool3s2_bwd_kernel<float>, dim3(grid), dim3(DL_THREADS), 0, s, make_view(*dy), arg, make_view(*dx));
  INSAR_CHECK_LAUNCH("insar_maxpool3s2_bwd");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// residual: out = relu(y*scale + shift + res) (Bottleneck.forward: out += identity; out = relu(out)), and the
// gradient gate of that ReLU: g = dout where out > 0, else 0 (g may alias dout).
// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void bn_add_relu_kernel(ActView y, const float* __restrict__ scale, const float* __restrict__ shift, ActView res,
                                   ActView dst, int relu) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = y.c_len / CH;
  const int total = y.W * cpp;
  for (int r = blockIdx.x; r < y.B * y.H; r += gridDim.x) {
    const int n = r / y.H, h = r - n * y.H;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int w = e / cpp, cc = e - w * cpp;
      float f[CH], q[CH];
      Chunk<T>::unpack(*dl_chunk<T>(y, n, h, w, cc), f);
      Chunk<T>::unpack(*dl_chunk<T>(res, n, h, w, cc), q);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        float v = fmaf(f[j], scale[cc * CH + j], shift[cc * CH + j]) + q[j];
        f[j] = relu ? fmaxf(v, 0.f) : v;
      }
      *dl_chunk_w<T>(dst, n, h, w, cc) = Chunk<T>::pack(f);
    }
  }
}

extern "C" int insar_bn_add_relu(const InsarAct* y, const float* scale, const float* shift, const InsarAct* res,
                                 const InsarAct* dst, int32_t relu, void* stream) {
  int rc;
  if ((rc = insar_check_act(y, "insar_bn_add_relu", "y"))) return rc;
  if ((rc = insar_check_act(res, "insar_bn_add_relu", "res"))) return rc;
  if ((rc = insar_check_act(dst, "insar_bn_add_relu", "dst"))) return rc;
  if (!scale || !shift) INSAR_FAIL(INSAR_E_ARG, "insar_bn_add_relu: null pointer");
  if ((rc = dl_same_grid(y, res, "insar_bn_add_relu"))) return rc;
  if ((rc = dl_same_grid(y, dst, "insar_bn_add_relu"))) return rc;
  int grid = insar_grid_cap((int64_t)y->B * y->H);
  hipStream_t s = (hipStream_t)stream;
  if (y->dtype == INSAR_BF16) hipLaunchKernelGGL(bn_add_relu_kernel<bf16_t>, dim3(grid), dim3(DL_THREADS), 0, s, make_view(*y), scale, shift, make_view(*res), make_view(*dst), relu);
  else hipLaunchKernelGGL(bn_add_relu_kernel<float>, dim3(grid), dim3(DL_THREADS), 0, s, make_view(*y), scale, shift, make_view(*res), make_view(*dst), relu);
  INSAR_CHECK_LAUNCH("insar_bn_add_relu");
  return INSAR_OK;
}

template <typename T>
__global__ void relu_gate_bwd_kernel(ActView dout, ActView out, ActView g) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = out.c_len / CH;
  const int total = out.W * cpp;
  for (int r = blockIdx.x; r < out.B * out.H; r += gridDim.x) {
    const int n = r / out.H, h = r - n * out.H;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int w = e / cpp, cc = e - w * cpp;
      float d[CH], o[CH];
      Chunk<T>::unpack(*dl_chunk<T>(dout, n, h, w, cc), d);
      Chunk<T>::unpack(*dl_chunk<T>(out, n, h, w, cc), o);
#pragma unroll
      for (int j = 0; j < CH; ++j) d[j] = o[j] > 0.f ? d[j] : 0.f;
      *dl_chunk_w<T>(g, n, h, w, cc) = Chunk<T>::pack(d);
    }
  }
}

extern "C" int insar_relu_gate_bwd(const InsarAct* dout, const InsarAct* out, const InsarAct* g, void* stream) {
  int rc;
  if ((rc = insar_check_act(dout, "insar_relu_gate_bwd", "dout"))) return rc;
  if ((rc = insar_check_act(out, "insar_relu_gate_bwd", "out"))) return rc;
  if ((rc = insar_check_act(g, "insar_relu_gate_bwd", "g"))) return rc;
  if ((rc = dl_same_grid(dout, out, "insar_relu_gate_bwd"))) return rc;
  if ((rc = dl_same_grid(dout, g, "insar_relu_gate_bwd"))) return rc;
  int grid = insar_grid_cap((int64_t)out->B * out->H);
  hipStream_t s = (hipStream_t)stream;
  if (out->dtype == INSAR_BF16) hipLaunchKernelGGL(relu_gate_bwd_kernel<bf16_t>, dim3(grid), dim3(DL_THREADS), 0, s, make_view(*dout), make_view(*out), make_view(*g));
  else hipLaunchKernelGGL(relu_gate_bwd_kernel<float>, dim3(grid), dim3(DL_THREADS), 0, s, make_view(*dout), make_view(*out), make_view(*g));
  INSAR_CHECK_LAUNCH("insar_relu_gate_bwd");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// global pooling over the image (AdaptiveAvgPool2d(1) of the ASPP pooling branch, and the adjoint of its broadcast):
// out[n, 0, 0, c] = factor * sum_hw x[n, h, w, c]; out is a (B, 1, 1, C) slice. One block per (image, 64 channels).
// ---------------------------------------------------------------------------------------------------------
template <typename T, typename TO>
__global__ void __launch_bounds__(DL_THREADS) sum_hw_kernel(ActView x, ActView out, float factor) {
  constexpr int CH = Chunk<T>::N;
  constexpr int CPB = 64 / CH;              // chunks per block (64 channels)
  constexpr int PL = DL_THREADS / CPB;      // pixel lanes
  __shared__ float red[DL_THREADS * CH];
  const int nblk_c = x.c_len / 64;
  const int n = blockIdx.x / nblk_c, cb = blockIdx.x - n * nblk_c;
  const int cc = cb * CPB + (threadIdx.x % CPB), pl = threadIdx.x / CPB;
  float acc[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) acc[j] = 0.f;
  const int HW = x.H * x.W;
  for (int p = pl; p < HW; p += PL) {
    const int h = p / x.W, w = p - h * x.W;
    float f[CH];
    Chunk<T>::unpack(*dl_chunk<T>(x, n, h, w, cc), f);
#pragma unroll
    for (int j = 0; j < CH; ++j) acc[j] += f[j];
  }
#pragma unroll
  for (int j = 0; j < CH; ++j) red[threadIdx.x * CH + j] = acc[j];
  __syncthreads();
  // one thread per output channel of the block's 64: fixed summation order over the pixel lanes
  if (threadIdx.x < 64) {
    const int c = threadIdx.x, chunk = c / CH, j = c % CH;
    float tot = 0.f;
    for (int p = 0; p < PL; ++p) tot += red[(p * CPB + chunk) * CH + j];
    tot *= factor;
    char* o = out.base + (out.elem_offset(n, 0, 0) + (int64_t)cb * 64 + c) * (int64_t)sizeof(TO);
    if constexpr (sizeof(TO) == 2) *(uint16_t*)o = f32_to_bf16(tot);
    else *(float*)o = tot;
  }
}

/* out may be fp32 while x is bf16: the pooled vector of the ASPP pooling branch is kept in fp32 (its sample-to-sample
 * spread is far below bf16's resolution of its mean, and the BatchNorm that follows divides by that spread). */
extern "C" int insar_sum_hw(const InsarAct* x, const InsarAct* out, float factor, void* stream) {
  int rc;
  if ((rc = insar_check_act(x, "insar_sum_hw", "x"))) return rc;
  if ((rc = insar_check_act(out, "insar_sum_hw", "out"))) return rc;
  if (out->B != x->B || out->H != 1 || out->W != 1 || out->c_len != x->c_len || x->c_len % 64)
    INSAR_FAIL(INSAR_E_SHAPE, "insar_sum_hw: out must be the (B, 1, 1, C) slice of x's channels (C %% 64 == 0)");
  if (out->dtype != x->dtype && out->dtype != INSAR_F32) INSAR_FAIL(INSAR_E_DTYPE, "insar_sum_hw: out must have x's dtype or be fp32");
  const int grid = x->B * (x->c_len / 64);
  hipStream_t s = (hipStream_t)stream;
  if (x->dtype == INSAR_BF16 && out->dtype == INSAR_BF16) hipLaunchKernelGGL((sum_hw_kernel<bf16_t, bf16_t>), dim3(grid), dim3(DL_THREADS), 0, s, make_view(*x), make_view(*out), factor);
  else if (x->dtype == INSAR_BF16) hipLaunchKernelGGL((sum_hw_kernel<bf16_t, float>), dim3(grid), dim3(DL_THREADS), 0, s, make_view(*x), make_view(*out), factor);
  else hipLaunchKernelGGL((sum_hw_kernel<float, float>), dim3(grid), dim3(DL_THREADS), 0, s, make_view(*x), make_view(*out), factor);
  INSAR_CHECK_LAUNCH("insar_sum_hw");
  return INSAR_OK;
}

// dst[n,h,w,c] = (accumulate ? dst : 0) + factor * src[n,0,0,c]: bilinear up-sampling of a 1x1 map (forward of the pooling
// branch, accumulate = 0) and the gradient of the global average (accumulate = 1, factor = 1/HW).
// gate (has_gate): the result is stored as zero where gate <= 0 (the ReLU mask of the residual block whose incoming gradient dst is)
template <typename T, typename TS>
__global__ void broadcast_hw_kernel(ActView src, ActView dst, float factor, int accumulate, ActView gate, int has_gate) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = dst.c_len / CH;
  const int total = dst.W * cpp;
  for (int r = blockIdx.x; r < dst.B * dst.H; r += gridDim.x) {
    const int n = r / dst.H, h = r - n * dst.H;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int w = e / cpp, cc = e - w * cpp;
      float f[CH], d[CH];
      const char* sp = src.base + (src.elem_offset(n, 0, 0) + (int64_t)cc * CH) * (int64_t)sizeof(TS);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        if constexpr (sizeof(TS) == 2) f[j] = bf16_to_f32(((const uint16_t*)sp)[j]);
        else f[j] = ((const float*)sp)[j];
      }
      if (accumulate) {
        Chunk<T>::unpack(*dl_chunk<T>(dst, n, h, w, cc), d);
#pragma unroll
        for (int j = 0; j < CH; ++j) f[j] = fmaf(f[j], factor, d[j]);
      } else {
#pragma unroll
        for (int j = 0; j < CH; ++j) f[j] *= factor;
      }
      if (has_gate) {
        float q[CH];
        Chunk<T>::unpack(*dl_chunk<T>(gate, n, h, w, cc), q);
#pragma unroll
        for (int j = 0; j < CH; ++j) f[j] = q[j] > 0.f ? f[j] : 0.f;
      }
      *dl_chunk_w<T>(dst, n, h, w, cc) = Chunk<T>::pack(f);
    }
  }
}

/* src may be fp32 while dst is bf16 (see insar_sum_hw). */
static int broadcast_hw_impl(const InsarAct* src, const InsarAct* dst, const InsarAct* gate, float factor, int32_t accumulate, void* stream);
extern "C" int insar_broadcast_hw(const InsarAct* src, const InsarAct* dst, float factor, int32_t accumulate, void* stream) {
  return broadcast_hw_impl(src, dst, nullptr, factor, accumulate, stream);
}
/* the same, storing zero where `gate` (dst's grid, channels and dtype) is <= 0: the last writer of a residual block's incoming
 * gradient applies the block's ReLU mask (as InsarIgemm.gate does for the GEMM writers) */
extern "C" int insar_broadcast_hw_gate(const InsarAct* src, const InsarAct* dst, const InsarAct* gate, float factor, int32_t accumulate,
                                       void* stream) {
  if (!gate) INSAR_FAIL(INSAR_E_ARG, "insar_broadcast_hw_gate: null gate");
  return broadcast_hw_impl(src, dst, gate, factor, accumulate, stream);
}
static int broadcast_hw_impl(const InsarAct* src, const InsarAct* dst, const InsarAct* gate, float factor, int32_t accumulate, void* stream) {
  int rc;
  if ((rc = insar_check_act(src, "insar_broadcast_hw", "src"))) return rc;
  if ((rc = insar_check_act(dst, "insar_broadcast_hw", "dst"))) return rc;
  if (src->B != dst->B || src->H != 1 || src->W != 1 || src->c_len != dst->c_len)
    INSAR_FAIL(INSAR_E_SHAPE, "insar_broadcast_hw: src must be the (B, 1, 1, C) slice of dst's channels");
  if (src->dtype != dst->dtype && src->dtype != INSAR_F32) INSAR_FAIL(INSAR_E_DTYPE, "insar_broadcast_hw: src must have dst's dtype or be fp32");
  if (gate) {
    if ((rc = insar_check_act(gate, "insar_broadcast_hw_gate", "gate"))) return rc;
    if (gate->dtype != dst->dtype || gate->B != dst->B || gate->H != dst->H || gate->W != dst->W || gate->c_len != dst->c_len)
      INSAR_FAIL(INSAR_E_SHAPE, "insar_broadcast_hw_gate: gate must have dst's grid, channels and dtype");
  }
  int grid = insar_grid_cap((int64_t)dst->B * dst->H);
  hipStream_t s = (hipStream_t)stream;
  const ActView gv = gate ? make_view(*gate) : make_view(*dst);
  const int hg = gate ? 1 : 0;
  if (dst->dtype == INSAR_BF16 && src->dtype == INSAR_BF16) hipLaunchKernelGGL((broadcast_hw_kernel<bf16_t, bf16_t>), dim3(grid), dim3(DL_THREADS), 0, s, make_view(*src), make_view(*dst), factor, accumulate, gv, hg);
  else if (dst->dtype == INSAR_BF16) hipLaunchKernelGGL((broadcast_hw_kernel<bf16_t, float>), dim3(grid), dim3(DL_THREADS), 0, s, make_view(*src), make_view(*dst), factor, accumulate, gv, hg);
  else hipLaunchKernelGGL((broadcast_hw_kernel<float, float>), dim3(grid), dim3(DL_THREADS), 0, s, make_view(*src), make_view(*dst), factor, accumulate, gv, hg);
  INSAR_CHECK_LAUNCH("insar_broadcast_hw");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Dropout(p) (ASPP projection, training mode): keep[i] = hash(seed, i) >= p, out = keep ? x / (1-p) : 0. The mask is
// stored (one byte per element, the layout of the slice) for the backward pass and for parity checks: torch's own
// Philox stream cannot be reproduced, so the oracle is given this mask.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t dl_hash(uint64_t seed, uint64_t i) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 32);
}

template <typename T>
__global__ void dropout_kernel(ActView x, ActView dst, uint8_t* __restrict__ mask, uint64_t seed, const int64_t* __restrict__ counter,
                               float p, float inv_keep, int make_mask) {
  if (counter) seed = ((uint64_t)dl_hash(seed, (uint64_t)*counter) << 32) | dl_hash(seed ^ 0x5851F42D4C957F2Dull, (uint64_t)*counter);
  constexpr int CH = Chunk<T>::N;
  const int cpp = x.c_len / CH;
  const int total = x.W * cpp;
  const uint32_t thr = (uint32_t)((double)p * 4294967296.0 > 4294967295.0 ? 4294967295.0 : (double)p * 4294967296.0);
  for (int r = blockIdx.x; r < x.B * x.H; r += gridDim.x) {
    const int n = r / x.H, h = r - n * x.H;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int w = e / cpp, cc = e - w * cpp;
      const int64_t base = (((int64_t)n * x.H + h) * x.W + w) * x.c_len + cc * CH;
      float f[CH];
      Chunk<T>::unpack(*dl_chunk<T>(x, n, h, w, cc), f);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        uint8_t keep;
        if (make_mask) { keep = dl_hash(seed, (uint64_t)(base + j)) >= thr ? 1 : 0; mask[base + j] = keep; }
        else keep = mask[base + j];
        f[j] = keep ? f[j] * inv_keep : 0.f;
      }
      *dl_chunk_w<T>(dst, n, h, w, cc) = Chunk<T>::pack(f);
    }
  }
}

/* make_mask != 0: draw the mask from (seed, element index) and store it; == 0: apply the stored mask (backward, or a
 * forward under a mask supplied by the caller). */
extern "C" int insar_dropout(const InsarAct* x, const InsarAct* dst, uint8_t* mask, uint64_t seed, const int64_t* counter, float p,
                             int32_t make_mask, void* stream) {
  int rc;
  if ((rc = insar_check_act(x, "insar_dropout", "x"))) return rc;
  if ((rc = insar_check_act(dst, "insar_dropout", "dst"))) return rc;
  if (!mask) INSAR_FAIL(INSAR_E_ARG, "insar_dropout: null mask");
  if (!(p >= 0.f && p < 1.f)) INSAR_FAIL(INSAR_E_ARG, "insar_dropout: p=%f outside [0, 1)", p);
  if ((rc = dl_same_grid(x, dst, "insar_dropout"))) return rc;
  int grid = insar_grid_cap((int64_t)x->B * x->H);
  hipStream_t s = (hipStream_t)stream;
  const float inv_keep = 1.f / (1.f - p);
  if (x->dtype == INSAR_BF16) hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(grid), dim3(DL_THREADS), 0, s, make_view(*x), make_view(*dst), mask, seed, counter, p, inv_keep, make_mask);
  else hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid), dim3(DL_THREADS), 0, s, make_view(*x), make_view(*dst), mask, seed, counter, p, inv_keep, make_mask);
  INSAR_CHECK_LAUNCH("insar_dropout");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------------------
// bilinear resize of NCHW fp32 maps, align_corners = False (F_T.resize(..., BILINEAR), :160), and its adjoint.
// src index of output o: max((o + 0.5) * in/out - 0.5, 0); i0 = floor, i1 = min(i0 + 1, in - 1), weight of i1 = frac.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dl_bilin(int o, float scale, int in, int& i0, int& i1, float& l1) {
  float src = ((float)o + 0.5f) * scale - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + 1 < in ? i0 + 1 : in - 1;
  l1 = src - (float)i0;
}

__global__ void bilinear_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int Hi, int Wi, int Ho, int Wo,
                                    float sh, float sw) {
  const int64_t total = (int64_t)planes * Ho * Wo;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int wo = (int)(q % Wo);
    const int ho = (int)((q / Wo) % Ho);
    const int64_t pl = q / ((int64_t)Wo * Ho);
    int h0, h1, w0, w1; float lh, lw;
    dl_bilin(ho, sh, Hi, h0, h1, lh);
    dl_bilin(wo, sw, Wi, w0, w1, lw);
    const float* p = in + pl * Hi * Wi;
    const float top = (1.f - lw) * p[h0 * Wi + w0] + lw * p[h0 * Wi + w1];
    const float bot = (1.f - lw) * p[h1 * Wi + w0] + lw * p[h1 * Wi + w1];
    out[q] = (1.f - lh) * top + lh * bot;
  }
}

// gather form of the adjoint: every input pixel sums the output pixels whose stencil touches it, in raster order
__global__ void bilinear_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, int planes, int Hi, int Wi, int Ho, int Wo,
                                    float sh, float sw) {
  const int64_t total = (int64_t)planes * Hi * Wi;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int wi = (int)(q % Wi);
    const int hi = (int)((q / Wi) % Hi);
    const int64_t pl = q / ((int64_t)Wi * Hi);
    int ho_lo = (int)floorf(((float)hi - 0.5f) / sh - 0.5f) - 1, ho_hi = (int)ceilf(((float)hi + 1.5f) / sh - 0.5f) + 1;
    int wo_lo = (int)floorf(((float)wi - 0.5f) / sw - 0.5f) - 1, wo_hi = (int)ceilf(((float)wi + 1.5f) / sw - 0.5f) + 1;
    if (hi == 0) ho_lo = 0;
    if (hi == Hi - 1) ho_hi = Ho - 1;
    if (wi == 0) wo_lo = 0;
    if (wi == Wi - 1) wo_hi = Wo - 1;
    ho_lo = ho_lo < 0 ? 0 : ho_lo; ho_hi = ho_hi > Ho - 1 ? Ho - 1 : ho_hi;
    wo_lo = wo_lo < 0 ? 0 : wo_lo; wo_hi = wo_hi > Wo - 1 ? Wo - 1 : wo_hi;
    const float* g = dout + pl * Ho * Wo;
    float acc = 0.f;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
      int h0, h1; float lh;
      dl_bilin(ho, sh, Hi, h0, h1, lh);
      float wh = 0.f;
      if (h0 == hi) wh += 1.f - lh;
      if (h1 == hi) wh += lh;
      if (wh == 0.f) continue;
      float row = 0.f;
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        int w0, w1; float lw;
        dl_bilin(wo, sw, Wi, w0, w1, lw);
        float ww = 0.f;
        if (w0 == wi) ww += 1.f - lw;
        if (w1 == wi) ww += lw;
        if (ww != 0.f) row = fmaf(ww, g[ho * Wo + wo], row);
      }
      acc = fmaf(wh, row, acc);
    }
    din[q] = acc;
  }
}

extern "C" int insar_bilinear_fwd(const float* in, float* out, int32_t planes, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                  void* stream) {
  if (!in || !out) INSAR_FAIL(INSAR_E_ARG, "insar_bilinear_fwd: null pointer");
  if (planes < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_bilinear_fwd: empty map");
  int grid = insar_grid_cap(((int64_t)planes * Ho * Wo + 255) / 256);
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, out, planes, Hi, Wi, Ho, Wo,
                     (float)Hi / (float)Ho, (float)Wi / (float)Wo);
  INSAR_CHECK_LAUNCH("insar_bilinear_fwd");
  return INSAR_OK;
}

extern "C" int insar_bilinear_bwd(const float* dout, float* din, int32_t planes, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                  void* stream) {
  if (!dout || !din) INSAR_FAIL(INSAR_E_ARG, "insar_bilinear_bwd: null pointer");
  if (planes < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_bilinear_bwd: empty map");
  int grid = insar_grid_cap(((int64_t)planes * Hi * Wi + 255) / 256);
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dout, din, planes, Hi, Wi, Ho, Wo,
                     (float)Hi / (float)Ho, (float)Wi / (float)Wo);
  INSAR_CHECK_LAUNCH("insar_bilinear_bwd");
  return INSAR_OK;
}
