// Loss entry points, metrics counts and the optimizer step of the training loop
// (Unet-ChannalAttention.py:344-346, 215-240, 465-466). All HBM-bound.
#include "common.h"

#define LO_THREADS 256
#define LO_MAXBLOCKS 1024
#define LO_MAXK 16

extern "C" int insar_ce_blocks(int64_t npix) {
  int64_t b = (npix + LO_THREADS - 1) / LO_THREADS;
  if (b < 1) b = 1;
  return (int)(b > LO_MAXBLOCKS ? LO_MAXBLOCKS : b);
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

// ws layout: [0] n_valid, [1] 1/n_valid, [2 .. 2+nb) count partials, [2+nb .. 2+2nb) loss partials
__global__ void ce_count_kernel(const int64_t* __restrict__ target, int64_t npix, int64_t ignore_index, float* ws) {
  __shared__ float red[8];
  float c = 0.f;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x)
    c += (target[p] != ignore_index) ? 1.f : 0.f;
  c = block_sum(c, red);
  if (threadIdx.x == 0) ws[2 + blockIdx.x] = c;
}

__global__ void ce_count_final_kernel(float* ws, int nb) {
  __shared__ float red[8];
  float c = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) c += ws[2 + i];
  c = block_sum(c, red);
  if (threadIdx.x == 0) { ws[0] = c; ws[1] = 1.f / c; }
}

// CrossEntropyLoss(ignore_index) forward + gradient in one pass over the logits (NCHW fp32).
__global__ void ce_main_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int K, int64_t HW,
                               int64_t npix, int64_t ignore_index, float* __restrict__ dlogits, float* ws, int nb) {
  __shared__ float red[8];
  const float inv = ws[1];
  float lsum = 0.f;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = p / HW, hw = p - n * HW;
    const float* lp = logits + n * K * HW + hw;
    float* gp = dlogits + n * K * HW + hw;
    const int64_t t = target[p];
    const bool valid = t != ignore_index;
    float mx = lp[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, lp[k * HW]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += __expf(lp[k * HW] - mx);
    const float lse = mx + __logf(se);
    const float rse = 1.f / se;
    for (int k = 0; k < K; ++k) {
      const float v = lp[k * HW];
      const float sm = __expf(v - mx) * rse;
      gp[k * HW] = valid ? (sm - (k == t ? 1.f : 0.f)) * inv : 0.f;
      if (valid && k == t) lsum += lse - v;
    }
  }
  lsum = block_sum(lsum, red);
  if (threadIdx.x == 0) ws[2 + nb + blockIdx.x] = lsum;
}

__global__ void ce_loss_final_kernel(float* ws, int nb, float* loss_out) {
  __shared__ float red[8];
  float c = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) c += ws[2 + nb + i];
  c = block_sum(c, red);
  if (threadIdx.x == 0) loss_out[0] = c * ws[1];
}

extern "C" int insar_cross_entropy(const float* logits, const int64_t* target, int32_t B, int32_t K, int64_t HW,
                                   int64_t ignore_index, float* dlogits, float* loss_out, float* ws, void* stream) {
  if (!logits || !target || !dlogits || !loss_out || !ws) INSAR_FAIL(INSAR_E_ARG, "insar_cross_entropy: null pointer");
  if (K < 1 || K > 4096 || B < 1 || HW < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_cross_entropy: bad shape");
  const int64_t npix = (int64_t)B * HW;
  const int nb = insar_ce_blocks(npix);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_count_kernel, dim3(nb), dim3(LO_THREADS), 0, s, target, npix, ignore_index, ws);
  hipLaunchKernelGGL(ce_count_final_kernel, dim3(1), dim3(LO_THREADS), 0, s, ws, nb);
  hipLaunchKernelGGL(ce_main_kernel, dim3(nb), dim3(LO_THREADS), 0, s, logits, target, K, HW, npix, ignore_index, dlogits, ws, nb);
  hipLaunchKernelGGL(ce_loss_final_kernel, dim3(1), dim3(LO_THREADS), 0, s, ws, nb, loss_out);
  INSAR_CHECK_LAUNCH("insar_cross_entropy");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// Soft-Dice on softmax probabilities (build-side addition; the reference has no Dice loss).
//   L = 1 - mean_c (2 I_c + s) / (D_c + s),  I_c = sum p_c*1[t=c]*v,  D_c = sum p_c*v + sum 1[t=c]*v
// pass 1: per-block partial (I_c, P_c, T_c); pass 2: fold; pass 3: gradient.
// ws layout: [0 .. 3K) totals I,P,T ; [3K .. 3K + nb*3K) block partials. K <= LO_MAXK.
// ---------------------------------------------------------------------------------------------
__global__ void dice_partial_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int K, int64_t HW,
                                    int64_t npix, int64_t ignore_index, float* ws) {
  __shared__ float red[8];
  float aI[LO_MAXK], aP[LO_MAXK], aT[LO_MAXK];
#pragma unroll
  for (int k = 0; k < LO_MAXK; ++k) { aI[k] = 0.f; aP[k] = 0.f; aT[k] = 0.f; }
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = target[p];
    if (t == ignore_index) continue;
    const int64_t n = p / HW, hw = p - n * HW;
    const float* lp = logits + n * K * HW + hw;
    float mx = lp[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, lp[k * HW]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += __expf(lp[k * HW] - mx);
    const float rse = 1.f / se;
#pragma unroll
    for (int k = 0; k < LO_MAXK; ++k) {
      if (k < K) {
        const float pr = __expf(lp[k * HW] - mx) * rse;
        aP[k] += pr;
        if (k == t) { aI[k] += pr; aT[k] += 1.f; }
      }
    }
  }
  float* out = ws + 3 * K + (int64_t)blockIdx.x * 3 * K;
#pragma unroll
  for (int k = 0; k < LO_MAXK; ++k) {
    if (k < K) {
      const float i = block_sum(aI[k], red), pp = block_sum(aP[k], red), tt = block_sum(aT[k], red);
      if (threadIdx.x == 0) { out[k] = i; out[K + k] = pp; out[2 * K + k] = tt; }
    }
  }
}

__global__ void dice_final_kernel(float* ws, int K, int nb, float smooth, float* loss_out) {
  __shared__ float red[8];
  __shared__ float tot[3 * LO_MAXK];
  for (int q = 0; q < 3 * K; ++q) {
    float c = 0.f;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) c += ws[3 * K + (int64_t)i * 3 * K + q];
    c = block_sum(c, red);
    if (threadIdx.x == 0) { ws[q] = c; tot[q] = c; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += (2.f * tot[k] + smooth) / (tot[K + k] + tot[2 * K + k] + smooth);
    loss_out[0] = 1.f - acc / (float)K;
  }
}

__global__ void dice_grad_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int K, int64_t HW,
                                 int64_t npix, int64_t ignore_index, float smooth, const float* __restrict__ ws,
                                 float* __restrict__ dlogits) {
  float num[LO_MAXK], den[LO_MAXK];
#pragma unroll
  for (int k = 0; k < LO_MAXK; ++k) {
    num[k] = k < K ? 2.f * ws[k] + smooth : 0.f;
    den[k] = k < K ? ws[K + k] + ws[2 * K + k] + smooth : 1.f;
  }
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = p / HW, hw = p - n * HW;
    const float* lp = logits + n * K * HW + hw;
    float* gp = dlogits + n * K * HW + hw;
    const int64_t t = target[p];
    if (t == ignore_index) { for (int k = 0; k < K; ++k) gp[k * HW] = 0.f; continue; }
    float mx = lp[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, lp[k * HW]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += __expf(lp[k * HW] - mx);
    const float rse = 1.f / se;
    float pr[LO_MAXK], gk[LO_MAXK];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < LO_MAXK; ++k) {
      if (k < K) {
        pr[k] = __expf(lp[k * HW] - mx) * rse;
        // dL/dp_k = -(1/K) * (2*1[t=k]*den - num) / den^2
        gk[k] = -((k == t ? 2.f * den[k] : 0.f) - num[k]) / (den[k] * den[k]) / (float)K;
        dot = fmaf(pr[k], gk[k], dot);
      }
    }
#pragma unroll
    for (int k = 0; k < LO_MAXK; ++k)
      if (k < K) gp[k * HW] = pr[k] * (gk[k] - dot);
  }
}

extern "C" int insar_dice(const float* logits, const int64_t* target, int32_t B, int32_t K, int64_t HW, int64_t ignore_index,
                          float smooth, float* dlogits, float* loss_out, float* ws, void* stream) {
  if (!logits || !target || !dlogits || !loss_out || !ws) INSAR_FAIL(INSAR_E_ARG, "insar_dice: null pointer");
  if (K < 1 || K > LO_MAXK) INSAR_FAIL(INSAR_E_SHAPE, "insar_dice: num_classes=%d must be 1..%d", K, LO_MAXK);
  const int64_t npix = (int64_t)B * HW;
  const int nb = insar_ce_blocks(npix);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dice_partial_kernel, dim3(nb), dim3(LO_THREADS), 0, s, logits, target, K, HW, npix, ignore_index, ws);
  hipLaunchKernelGGL(dice_final_kernel, dim3(1), dim3(LO_THREADS), 0, s, ws, K, nb, smooth, loss_out);
  hipLaunchKernelGGL(dice_grad_kernel, dim3(nb), dim3(LO_THREADS), 0, s, logits, target, K, HW, npix, ignore_index, smooth, ws, dlogits);
  INSAR_CHECK_LAUNCH("insar_dice");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// ce_weight * CrossEntropy + dice_weight * Dice in three launches (one statistics pass over the logits,
// one fold, one gradient pass) instead of the seven of the two separate entry points.
// ws layout (floats): [0] n_valid [1] 1/n_valid [2] sum CE  [3 .. 3+3K) totals I,P,T ;
//                     then per block: [count, ce_sum, I[K], P[K], T[K]]  (2 + 3K floats each)
// ---------------------------------------------------------------------------------------------
__global__ void dicece_partial_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int K, int64_t HW,
                                      int64_t npix, int64_t ignore_index, float* ws) {
  __shared__ float red[8];
  float aI[LO_MAXK], aP[LO_MAXK], aT[LO_MAXK];
  float cnt = 0.f, lsum = 0.f;
#pragma unroll
  for (int k = 0; k < LO_MAXK; ++k) { aI[k] = 0.f; aP[k] = 0.f; aT[k] = 0.f; }
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = target[p];
    if (t == ignore_index) continue;
    const int64_t n = p / HW, hw = p - n * HW;
    const float* lp = logits + n * K * HW + hw;
    float mx = lp[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, lp[k * HW]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += __expf(lp[k * HW] - mx);
    const float rse = 1.f / se;
    const float lse = mx + __logf(se);
    cnt += 1.f;
#pragma unroll
    for (int k = 0; k < LO_MAXK; ++k) {
      if (k < K) {
        const float v = lp[k * HW];
        const float pr = __expf(v - mx) * rse;
        aP[k] += pr;
        if (k == t) { aI[k] += pr; aT[k] += 1.f; lsum += lse - v; }
      }
    }
  }
  float* out = ws + 3 + 3 * K + (int64_t)blockIdx.x * (2 + 3 * K);
  const float c = block_sum(cnt, red), l = block_sum(lsum, red);
  if (threadIdx.x == 0) { out[0] = c; out[1] = l; }
#pragma unroll
  for (int k = 0; k < LO_MAXK; ++k) {
    if (k < K) {
      const float i = block_sum(aI[k], red), pp = block_sum(aP[k], red), tt = block_sum(aT[k], red);
      if (threadIdx.x == 0) { out[2 + k] = i; out[2 + K + k] = pp; out[2 + 2 * K + k] = tt; }
    }
  }
}

__global__ void dicece_final_kernel(float* ws, int K, int nb, float smooth, float ce_w, float dice_w, float* loss_out) {
  __shared__ float red[8];
  __shared__ float tot[2 + 3 * LO_MAXK];
  const int stride = 2 + 3 * K;
  // every (block, quantity) partial this thread folds is requested before the first block_sum (the launch is pure latency
  // between the statistics and the gradient pass: one round trip instead of one per quantity)
  constexpr int QM = 2 + 3 * 4;
  if (stride <= QM && nb <= 4 * (int)blockDim.x) {
    float v[4][QM];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = threadIdx.x + r * blockDim.x;
#pragma unroll
      for (int q = 0; q < QM; ++q) v[r][q] = (i < nb && q < stride) ? ws[3 + 3 * K + (int64_t)i * stride + q] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < QM; ++q) {
      if (q < stride) {                       // uniform
        const float c = block_sum(((v[0][q] + v[1][q]) + v[2][q]) + v[3][q], red);
        if (threadIdx.x == 0) tot[q] = c;
      }
    }
  } else {
    for (int q = 0; q < stride; ++q) {
      float c = 0.f;
      for (int i = threadIdx.x; i < nb; i += blockDim.x) c += ws[3 + 3 * K + (int64_t)i * stride + q];
      c = block_sum(c, red);
      if (threadIdx.x == 0) tot[q] = c;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float nvalid = tot[0];
    ws[0] = nvalid; ws[1] = 1.f / nvalid; ws[2] = tot[1];
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      ws[3 + k] = tot[2 + k]; ws[3 + K + k] = tot[2 + K + k]; ws[3 + 2 * K + k] = tot[2 + 2 * K + k];
      acc += (2.f * tot[2 + k] + smooth) / (tot[2 + K + k] + tot[2 + 2 * K + k] + smooth);
    }
    const float ce = tot[1] / nvalid, dice = 1.f - acc / (float)K;
    loss_out[0] = ce_w * ce + dice_w * dice;
    loss_out[1] = ce;
    loss_out[2] = dice;
  }
}

// K <= 4 and HW a multiple of 4: four consecutive pixels of an image per thread and trip, their targets (two 16-byte
// loads) and logits (one 16-byte load per class) requested together — the per-pixel loops above wait for the target, then
// for each class plane in turn (18.6 + 13.4 us for 16 MB of logits and targets at config 2; these take one round trip).
// Same per-pixel arithmetic as dicece_partial_kernel / dicece_grad_kernel; only the grouping of the block sums differs.
template <int KT>
__global__ void dicece_partial_v4(const float* __restrict__ logits, const int64_t* __restrict__ target, int64_t HW,
                                  int64_t npix, int64_t ignore_index, float* ws) {
  __shared__ float red[8];
  float aI[KT], aP[KT], aT[KT];
  float cnt = 0.f, lsum = 0.f;
#pragma unroll
  for (int k = 0; k < KT; ++k) { aI[k] = 0.f; aP[k] = 0.f; aT[k] = 0.f; }
  const int64_t ngroups = npix >> 2;
  for (int64_t gi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; gi < ngroups; gi += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = gi << 2;
    const int64_t n = p / HW, hw = p - n * HW;
    const longlong2 t01 = *(const longlong2*)(target + p), t23 = *(const longlong2*)(target + p + 2);
    float4 lv[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) lv[k] = *(const float4*)(logits + (n * KT + k) * HW + hw);
    const int64_t tt[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t t = tt[e];
      if (t == ignore_index) continue;
      float v[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) v[k] = ((const float*)&lv[k])[e];
      float mx = v[0];
#pragma unroll
      for (int k = 1; k < KT; ++k) mx = fmaxf(mx, v[k]);
      float se = 0.f;
#pragma unroll
      for (int k = 0; k < KT; ++k) se += __expf(v[k] - mx);
      const float rse = 1.f / se;
      const float lse = mx + __logf(se);
      cnt += 1.f;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const float pr = __expf(v[k] - mx) * rse;
        aP[k] += pr;
        if (k == t) { aI[k] += pr; aT[k] += 1.f; lsum += lse - v[k]; }
      }
    }
  }
  float* out = ws + 3 + 3 * KT + (int64_t)blockIdx.x * (2 + 3 * KT);
  const float c = block_sum(cnt, red), l = block_sum(lsum, red);
  if (threadIdx.x == 0) { out[0] = c; out[1] = l; }
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const float i = block_sum(aI[k], red), pp = block_sum(aP[k], red), tt = block_sum(aT[k], red);
    if (threadIdx.x == 0) { out[2 + k] = i; out[2 + KT + k] = pp; out[2 + 2 * KT + k] = tt; }
  }
}

template <int KT>
__global__ void dicece_grad_v4(const float* __restrict__ logits, const int64_t* __restrict__ target, int64_t HW,
                               int64_t npix, int64_t ignore_index, float smooth, float ce_w, float dice_w,
                               const float* __restrict__ ws, float* __restrict__ dlogits) {
  float num[KT], den[KT];
  const float inv = ws[1] * ce_w;
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    num[k] = 2.f * ws[3 + k] + smooth;
    den[k] = ws[3 + KT + k] + ws[3 + 2 * KT + k] + smooth;
  }
  const int64_t ngroups = npix >> 2;
  for (int64_t gi = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; gi < ngroups; gi += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = gi << 2;
    const int64_t n = p / HW, hw = p - n * HW;
    const longlong2 t01 = *(const longlong2*)(target + p), t23 = *(const longlong2*)(target + p + 2);
    float4 lv[KT], gv[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) lv[k] = *(const float4*)(logits + (n * KT + k) * HW + hw);
    const int64_t tt[4] = {t01.x, t01.y, t23.x, t23.y};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t t = tt[e];
      float v[KT], o[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) { v[k] = ((const float*)&lv[k])[e]; o[k] = 0.f; }
      if (t != ignore_index) {
        float mx = v[0];
#pragma unroll
        for (int k = 1; k < KT; ++k) mx = fmaxf(mx, v[k]);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) se += __expf(v[k] - mx);
        const float rse = 1.f / se;
        float pr[KT], gk[KT];
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          pr[k] = __expf(v[k] - mx) * rse;
          gk[k] = -((k == t ? 2.f * den[k] : 0.f) - num[k]) / (den[k] * den[k]) / (float)KT;
          dot = fmaf(pr[k], gk[k], dot);
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) o[k] = (pr[k] - (k == t ? 1.f : 0.f)) * inv + dice_w * pr[k] * (gk[k] - dot);
      }
#pragma unroll
      for (int k = 0; k < KT; ++k) ((float*)&gv[k])[e] = o[k];
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) *(float4*)(dlogits + (n * KT + k) * HW + hw) = gv[k];
  }
}

__global__ void dicece_grad_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int K, int64_t HW,
                                   int64_t npix, int64_t ignore_index, float smooth, float ce_w, float dice_w,
                                   const float* __restrict__ ws, float* __restrict__ dlogits) {
  float num[LO_MAXK], den[LO_MAXK];
  const float inv = ws[1] * ce_w;
#pragma unroll
  for (int k = 0; k < LO_MAXK; ++k) {
    num[k] = k < K ? 2.f * ws[3 + k] + smooth : 0.f;
    den[k] = k < K ? ws[3 + K + k] + ws[3 + 2 * K + k] + smooth : 1.f;
  }
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = p / HW, hw = p - n * HW;
    const float* lp = logits + n * K * HW + hw;
    float* gp = dlogits + n * K * HW + hw;
    const int64_t t = target[p];
    if (t == ignore_index) { for (int k = 0; k < K; ++k) gp[k * HW] = 0.f; continue; }
    float mx = lp[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, lp[k * HW]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += __expf(lp[k * HW] - mx);
    const float rse = 1.f / se;
    float pr[LO_MAXK], gk[LO_MAXK];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < LO_MAXK; ++k) {
      if (k < K) {
        pr[k] = __expf(lp[k * HW] - mx) * rse;
        gk[k] = -((k == t ? 2.f * den[k] : 0.f) - num[k]) / (den[k] * den[k]) / (float)K;
        dot = fmaf(pr[k], gk[k], dot);
      }
    }
#pragma unroll
    for (int k = 0; k < LO_MAXK; ++k)
      if (k < K) gp[k * HW] = (pr[k] - (k == t ? 1.f : 0.f)) * inv + dice_w * pr[k] * (gk[k] - dot);
  }
}

extern "C" int insar_dice_ce(const float* logits, const int64_t* target, int32_t B, int32_t K, int64_t HW, int64_t ignore_index,
                             float smooth, float ce_weight, float dice_weight, float* dlogits, float* loss_out, float* ws,
                             void* stream) {
  if (!logits || !target || !dlogits || !loss_out || !ws) INSAR_FAIL(INSAR_E_ARG, "insar_dice_ce: null pointer");
  if (K < 1 || K > LO_MAXK) INSAR_FAIL(INSAR_E_SHAPE, "insar_dice_ce: num_classes=%d must be 1..%d", K, LO_MAXK);
  if (B < 1 || HW < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_dice_ce: bad shape");
  const int64_t npix = (int64_t)B * HW;
  const int nb = insar_ce_blocks(npix);
  hipStream_t s = (hipStream_t)stream;
  const bool v4 = K >= 2 && K <= 4 && (HW & 3) == 0 && insar_aligned16(logits) && insar_aligned16(target) && insar_aligned16(dlogits);
#define DICECE_V4(KT)                                                                                                              \
  do {                                                                                                                             \
    hipLaunchKernelGGL(dicece_partial_v4<KT>, dim3(nb), dim3(LO_THREADS), 0, s, logits, target, HW, npix, ignore_index, ws);        \
    hipLaunchKernelGGL(dicece_final_kernel, dim3(1), dim3(LO_THREADS), 0, s, ws, K, nb, smooth, ce_weight, dice_weight, loss_out); \
    hipLaunchKernelGGL(dicece_grad_v4<KT>, dim3(nb), dim3(LO_THREADS), 0, s, logits, target, HW, npix, ignore_index, smooth,        \
                       ce_weight, dice_weight, ws, dlogits);                                                                        \
  } while (0)
  if (v4 && K == 2) DICECE_V4(2);
  else if (v4 && K == 3) DICECE_V4(3);
  else if (v4 && K == 4) DICECE_V4(4);
  else {
    hipLaunchKernelGGL(dicece_partial_kernel, dim3(nb), dim3(LO_THREADS), 0, s, logits, target, K, HW, npix, ignore_index, ws);
    hipLaunchKernelGGL(dicece_final_kernel, dim3(1), dim3(LO_THREADS), 0, s, ws, K, nb, smooth, ce_weight, dice_weight, loss_out);
    hipLaunchKernelGGL(dicece_grad_kernel, dim3(nb), dim3(LO_THREADS), 0, s, logits, target, K, HW, npix, ignore_index, smooth,
                       ce_weight, dice_weight, ws, dlogits);
  }
#undef DICECE_V4
  INSAR_CHECK_LAUNCH("insar_dice_ce");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// compute_metrics counts (Unet-ChannalAttention.py:220-240): argmax over classes with ties going to
// the LOWER class index (torch.max, :220), pixels with target == 255 ignored (:223);
// counts[0][c] = TP, counts[1][c] = FP, counts[2][c] = FN  (int64, exact).
// ---------------------------------------------------------------------------------------------
__global__ void confusion_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, int K, int64_t HW,
                                 int64_t npix, int64_t ignore_index, unsigned long long* counts) {
  extern __shared__ unsigned int sc[];               // [3][K]
  for (int i = threadIdx.x; i < 3 * K; i += blockDim.x) sc[i] = 0;
  __syncthreads();
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = target[p];
    if (t == ignore_index) continue;
    const int64_t n = p / HW, hw = p - n * HW;
    const float* lp = logits + n * K * HW + hw;
    int best = 0; float mx = lp[0];
    for (int k = 1; k < K; ++k) { const float v = lp[k * HW]; if (v > mx) { mx = v; best = k; } }
    if (best == (int)t) atomicAdd(&sc[best], 1u);
    else {
      atomicAdd(&sc[K + best], 1u);
      if (t >= 0 && t < K) atomicAdd(&sc[2 * K + (int)t], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * K; i += blockDim.x)
    if (sc[i]) atomicAdd(&counts[i], (unsigned long long)sc[i]);
}

extern "C" int insar_confusion(const float* logits, const int64_t* target, int32_t B, int32_t K, int64_t HW,
                               int64_t ignore_index, int64_t* counts, void* stream) {
  if (!logits || !target || !counts) INSAR_FAIL(INSAR_E_ARG, "insar_confusion: null pointer");
  if (K < 1 || K > 1024) INSAR_FAIL(INSAR_E_SHAPE, "insar_confusion: bad num_classes");
  const int64_t npix = (int64_t)B * HW;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)3 * K * sizeof(int64_t), s);
  if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_confusion: memset: %s", hipGetErrorString(e));
  // per-block LDS counters are 32-bit: bound pixels per block below 2^32
  const int nb = insar_ce_blocks(npix);
  hipLaunchKernelGGL(confusion_kernel, dim3(nb), dim3(LO_THREADS), (size_t)3 * K * sizeof(unsigned int), s, logits, target, K, HW,
                     npix, ignore_index, (unsigned long long*)counts);
  INSAR_CHECK_LAUNCH("insar_confusion");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults, Unet-ChannalAttention.py:466): multi-tensor, one launch.
//   m = m + (g - m)(1-b1);  v = b2 v + (1-b2) g^2;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// 28 bytes of HBM traffic per parameter; 16-byte vector accesses on 16-byte aligned tensors.
// ---------------------------------------------------------------------------------------------
// DEV: the bias corrections come from device memory (state[1] = 1 - beta1^t, state[2] = sqrt(1 - beta2^t), written by
// adam_advance_kernel), so that a captured hipGraph of the training step stays valid from one replay to the next.
template <bool DEV>
__global__ void adam_kernel(const int64_t* __restrict__ table, const int32_t* __restrict__ chunks, int chunk_elems,
                            float lr_over_bc1, float b1, float b2, float eps, float inv_bc2_sqrt, float gscale,
                            const float* __restrict__ state) {
  if constexpr (DEV) {
    lr_over_bc1 = lr_over_bc1 / state[1];          // the host passes lr here
    inv_bc2_sqrt = 1.f / state[2];
  }
  const int ti = chunks[2 * blockIdx.x], ci = chunks[2 * blockIdx.x + 1];
  float* p = (float*)table[5 * ti + 0];
  const float* g = (const float*)table[5 * ti + 1];
  float* m = (float*)table[5 * ti + 2];
  float* v = (float*)table[5 * ti + 3];
  const int64_t numel = table[5 * ti + 4];
  const int64_t beg = (int64_t)ci * chunk_elems;
  int64_t end = beg + chunk_elems; if (end > numel) end = numel;
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  const float omb1 = 1.f - b1, omb2 = 1.f - b2;
  if (vec) {
    const int64_t end4 = beg + ((end - beg) & ~(int64_t)3);
    for (int64_t i = beg + threadIdx.x * 4; i < end4; i += (int64_t)blockDim.x * 4) {
      float4 pp = *(float4*)(p + i), gg = *(const float4*)(g + i), mm = *(float4*)(m + i), vv = *(float4*)(v + i);
      float* pa = (float*)&pp; float* ga = (float*)&gg; float* ma = (float*)&mm; float* va = (float*)&vv;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gj = ga[j] * gscale;
        ma[j] = ma[j] + (gj - ma[j]) * omb1;
        va[j] = va[j] * b2 + omb2 * gj * gj;
        pa[j] = pa[j] - lr_over_bc1 * (ma[j] / (sqrtf(va[j]) * inv_bc2_sqrt + eps));
      }
      *(float4*)(p + i) = pp; *(float4*)(m + i) = mm; *(float4*)(v + i) = vv;
    }
    for (int64_t i = end4 + threadIdx.x; i < end; i += blockDim.x) {
      const float gj = g[i] * gscale;
      const float mj = m[i] + (gj - m[i]) * omb1;
      const float vj = v[i] * b2 + omb2 * gj * gj;
      m[i] = mj; v[i] = vj;
      p[i] = p[i] - lr_over_bc1 * (mj / (sqrtf(vj) * inv_bc2_sqrt + eps));
    }
  } else {
    for (int64_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
      const float gj = g[i] * gscale;
      const float mj = m[i] + (gj - m[i]) * omb1;
      const float vj = v[i] * b2 + omb2 * gj * gj;
      m[i] = mj; v[i] = vj;
      p[i] = p[i] - lr_over_bc1 * (mj / (sqrtf(vj) * inv_bc2_sqrt + eps));
    }
  }
}

extern "C" int insar_adam_step(const int64_t* table, const int32_t* chunks, int32_t nchunks, int32_t chunk_elems, float lr,
                               float beta1, float beta2, float eps, float bias_correction1, float bias_correction2_sqrt,
                               float grad_scale, void* stream) {
  if (!table || !chunks) INSAR_FAIL(INSAR_E_ARG, "insar_adam_step: null pointer");
  if (nchunks < 1 || chunk_elems < 4 || (chunk_elems & 3)) INSAR_FAIL(INSAR_E_SHAPE, "insar_adam_step: bad chunking");
  hipLaunchKernelGGL(adam_kernel<false>, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, chunks, chunk_elems,
                     lr / bias_correction1, beta1, beta2, eps, 1.f / bias_correction2_sqrt, grad_scale, (const float*)nullptr);
  INSAR_CHECK_LAUNCH("insar_adam_step");
  return INSAR_OK;
}

// state: float[4] = {t, 1 - beta1^t, sqrt(1 - beta2^t), unused}; t counts the steps taken (exact up to 2^24)
__global__ void adam_advance_kernel(float* state, double b1, double b2) {
  const double t = (double)state[0] + 1.0;
  state[0] = (float)t;
  state[1] = (float)(1.0 - pow(b1, t));
  state[2] = (float)sqrt(1.0 - pow(b2, t));
}

extern "C" int insar_adam_step_dev(const int64_t* table, const int32_t* chunks, int32_t nchunks, int32_t chunk_elems, float lr,
                                   double beta1, double beta2, float eps, float* state, float grad_scale, void* stream) {
  if (!table || !chunks || !state) INSAR_FAIL(INSAR_E_ARG, "insar_adam_step_dev: null pointer");
  if (nchunks < 1 || chunk_elems < 4 || (chunk_elems & 3)) INSAR_FAIL(INSAR_E_SHAPE, "insar_adam_step_dev: bad chunking");
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, beta1, beta2);
  hipLaunchKernelGGL(adam_kernel<true>, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, chunks, chunk_elems,
                     lr, (float)beta1, (float)beta2, eps, 1.f, grad_scale, (const float*)state);
  INSAR_CHECK_LAUNCH("insar_adam_step_dev");
  return INSAR_OK;
}
