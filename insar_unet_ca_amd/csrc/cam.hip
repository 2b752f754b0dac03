// ChannelAttentionModule (DeepLabV3-ChannelAttention.py:49-79, config 5): per image and channel the spatial
// mean and maximum go through one shared bias-free MLP (1x1 conv C -> C/r, ReLU, 1x1 conv C/r -> C), the two
// results are added, squashed by a sigmoid and multiply the feature map:
//     out = x * sigmoid( W2 relu(W1 avg(x)) + W2 relu(W1 max(x)) )
// Forward: insar_cam_pool (row partials: sum, max, arg-max) -> insar_cam_excite (fold, MLP twice, gate) ->
//          insar_bn_relu_apply(x, 1, 0, gate) (the multiply).
// Backward: insar_bnrelu_bwd_reduce(dout, x) (sum dout*x) -> insar_cam_bwd_coef (MLP backward, weight grads,
//          per-(n,c) mean-branch term coefB = davg/HW and max-branch term dmax) ->
//          insar_bnrelu_bwd_apply (dx = dout*gate + coefB) -> insar_cam_scatter_max (dx[argmax] += dmax).
// The maximum follows torch's AdaptiveMaxPool2d: first maximum in scan order (h, then w) wins.
#include "common.h"

#define CAM_THREADS 256
#define CAM_COEF_THREADS 1024

template <typename T>
__device__ __forceinline__ const uint4* cam_chunk(const ActView& v, int n, int h, int w, int cc) {
  return (const uint4*)(v.base + (v.elem_offset(n, h, w) + (int64_t)cc * Chunk<T>::N) * (int64_t)sizeof(T));
}

// part r = (image n, rows [h0, h0+rpp)): psum/pmax/parg[r][C]; parg = flat index h*W + w of the first maximum
template <typename T>
__global__ void cam_pool_kernel(ActView x, float* __restrict__ psum, float* __restrict__ pmax, int* __restrict__ parg, int rpp) {
  constexpr int CH = Chunk<T>::N;
  __shared__ float rs[CAM_THREADS][CH + 1];
  __shared__ float rm[CAM_THREADS][CH + 1];
  __shared__ int ra[CAM_THREADS][CH + 1];
  const int cpp = x.c_len / CH;
  const int ppi = (x.H + rpp - 1) / rpp;
  const int nparts = x.B * ppi;
  const int cc = threadIdx.x % cpp, wstep = blockDim.x / cpp;      // host guarantees blockDim % cpp == 0
  for (int r = blockIdx.x; r < nparts; r += gridDim.x) {
    const int n = r / ppi, h0 = (r - n * ppi) * rpp;
    const int h1 = min(x.H, h0 + rpp);
    float s[CH], m[CH];
    int a[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) { s[j] = 0.f; m[j] = -INFINITY; a[j] = 0x7fffffff; }
    for (int h = h0; h < h1; ++h)
      for (int w = threadIdx.x / cpp; w < x.W; w += wstep) {
        float f[CH];
        Chunk<T>::unpack(*cam_chunk<T>(x, n, h, w, cc), f);
        const int idx = h * x.W + w;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          s[j] += f[j];
          if (a[j] == 0x7fffffff || f[j] > m[j] || (f[j] != f[j] && m[j] == m[j])) { m[j] = f[j]; a[j] = idx; }   // strictly greater (or first NaN): earlier index stays
        }
      }
#pragma unroll
    for (int j = 0; j < CH; ++j) { rs[threadIdx.x][j] = s[j]; rm[threadIdx.x][j] = m[j]; ra[threadIdx.x][j] = a[j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < x.c_len; c += blockDim.x) {
      const int occ = c / CH, j = c - occ * CH;
      float ss = 0.f, mm = -INFINITY;
      int aa = 0x7fffffff;
      for (int t = occ; t < blockDim.x; t += cpp) {
        ss += rs[t][j];
        const float v = rm[t][j];
        const int ai = ra[t][j];
        if (ai != 0x7fffffff && (aa == 0x7fffffff || v > mm || (v == mm && ai < aa) || (v != v && mm == mm))) { mm = v; aa = ai; }
      }
      psum[(int64_t)r * x.c_len + c] = ss;
      pmax[(int64_t)r * x.c_len + c] = mm;
      parg[(int64_t)r * x.c_len + c] = aa;
    }
    __syncthreads();
  }
}

extern "C" int insar_cam_pool(const InsarAct* x, float* psum, float* pmax, int32_t* parg, int32_t rows_per_part, void* stream) {
  int rc;
  if ((rc = insar_check_act(x, "insar_cam_pool", "x"))) return rc;
  if (!psum || !pmax || !parg) INSAR_FAIL(INSAR_E_ARG, "insar_cam_pool: null pointer");
  if (rows_per_part < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_cam_pool: rows_per_part");
  const int ch = x->dtype == INSAR_BF16 ? 8 : 4;
  const int cpp = x->c_len / ch;
  if (x->c_len % ch || cpp < 1 || CAM_THREADS % cpp) INSAR_FAIL(INSAR_E_SHAPE, "insar_cam_pool: C=%d unsupported", x->c_len);
  if ((int64_t)x->H * x->W >= 0x7fffffffLL) INSAR_FAIL(INSAR_E_SHAPE, "insar_cam_pool: image too large");
  const int rpp = rows_per_part;
  const int grid = insar_grid_cap((int64_t)x->B * ((x->H + rpp - 1) / rpp));
  hipStream_t s = (hipStream_t)stream;
  if (x->dtype == INSAR_BF16) hipLaunchKernelGGL(cam_pool_kernel<bf16_t>, dim3(grid), dim3(CAM_THREADS), 0, s, make_view(*x), psum, pmax, parg, rpp);
  else hipLaunchKernelGGL(cam_pool_kernel<float>, dim3(grid), dim3(CAM_THREADS), 0, s, make_view(*x), psum, pmax, parg, rpp);
  INSAR_CHECK_LAUNCH("insar_cam_pool");
  return INSAR_OK;
}

// ---- excitation: one block per image ---------------------------------------------------------------
__global__ void __launch_bounds__(CAM_COEF_THREADS) cam_excite_kernel(InsarCam d) {
  extern __shared__ float sm[];
  float* avg = sm;               // [C]
  float* mx = sm + d.C;          // [C]
  float* hs = mx + d.C;          // [Cr]  ha + hm
  const int n = blockIdx.x;
  const float inv_hw = 1.f / ((float)d.H * (float)d.W);
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    float ss = 0.f, mm = -INFINITY;
    int aa = 0x7fffffff;
    for (int r = 0; r < d.rows; ++r) {               // parts in scan order: strictly greater replaces
      const int64_t o = ((int64_t)n * d.rows + r) * d.C + c;
      ss += d.psum[o];
      const float v = d.pmax[o];
      if (aa == 0x7fffffff || v > mm || (v != v && mm == mm)) { mm = v; aa = d.parg[o]; }
    }
    const float a = ss * inv_hw;
    avg[c] = a; mx[c] = mm;
    d.avg[(int64_t)n * d.C + c] = a;
    d.mx[(int64_t)n * d.C + c] = mm;
    d.arg[(int64_t)n * d.C + c] = aa;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int j = wave; j < d.Cr; j += nw) {
    float aa = 0.f, am = 0.f;
#pragma unroll 4
    for (int c = lane; c < d.C; c += 64) {
      const float w = d.w1[(int64_t)j * d.C + c];
      aa = fmaf(w, avg[c], aa); am = fmaf(w, mx[c], am);
    }
    aa = wave_sum(aa); am = wave_sum(am);
    if (lane == 0) {
      const float ha = fmaxf(aa, 0.f), hm = fmaxf(am, 0.f);
      d.ha[(int64_t)n * d.Cr + j] = ha;
      d.hm[(int64_t)n * d.Cr + j] = hm;
      hs[j] = ha + hm;
    }
  }
  __syncthreads();
  for (int c0 = wave; c0 < d.C; c0 += 8 * nw) {
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + u * nw;
      acc[u] = 0.f;
      if (c < d.C)
        for (int j = lane; j < d.Cr; j += 64) acc[u] = fmaf(d.w2[(int64_t)c * d.Cr + j], hs[j], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + u * nw;
      const float t = wave_sum(acc[u]);
      if (lane == 0 && c < d.C) d.gate[(int64_t)n * d.C + c] = 1.f / (1.f + __expf(-t));
    }
  }
}

static int cam_check(const InsarCam* d, const char* who) {
  if (!d || !d->w1 || !d->w2 || !d->avg || !d->mx || !d->arg || !d->ha || !d->hm || !d->gate)
    INSAR_FAIL(INSAR_E_ARG, "%s: null pointer", who);
  if (d->B < 1 || d->C < 1 || d->Cr < 1 || d->C > 8192 || d->rows < 1 || d->H < 1 || d->W < 1) INSAR_FAIL(INSAR_E_SHAPE, "%s: bad shape", who);
  return INSAR_OK;
}

extern "C" int insar_cam_excite(const InsarCam* d, void* stream) {
  int rc;
  if ((rc = cam_check(d, "insar_cam_excite"))) return rc;
  if (!d->psum || !d->pmax || !d->parg) INSAR_FAIL(INSAR_E_ARG, "insar_cam_excite: null partials");
  const size_t lds = (size_t)(2 * d->C + d->Cr) * sizeof(float);
  hipLaunchKernelGGL(cam_excite_kernel, dim3(d->B), dim3(CAM_COEF_THREADS), lds, (hipStream_t)stream, *d);
  INSAR_CHECK_LAUNCH("insar_cam_excite");
  return INSAR_OK;
}

// ---- backward coefficients --------------------------------------------------------------------------
// red[n][rows][2][C] from insar_bnrelu_bwd_reduce(dout, x, relu=0): [.][1][c] = sum_hw dout*x.
// stage 1 (block per image): du = ds*s*(1-s); t = W2^T du; dta = t*[ha>0], dtm = t*[hm>0];
//   coefB = (W1^T dta)/HW, dmax = W1^T dtm;  ws: du[B][C] | dta[B][Cr] | dtm[B][Cr]
// stage 2 (thread per weight): dW2[c][j] = sum_n du[n][c]*(ha+hm)[n][j];  dW1[j][c] = sum_n dta[n][j]*avg[n][c] + dtm[n][j]*mx[n][c]
struct CamBwdArgs { InsarCam d; const float* red; int rows; };

__global__ void __launch_bounds__(CAM_COEF_THREADS) cam_bwd_stage1(CamBwdArgs a) {
  extern __shared__ float sm[];
  const InsarCam& d = a.d;
  float* du_s = sm;                 // [C]
  float* ta_s = sm + d.C;           // [Cr]
  float* tm_s = ta_s + d.Cr;        // [Cr]
  float* scratch = tm_s + d.Cr;     // [blockDim]
  const int n = blockIdx.x;
  float* du_g = d.ws + (int64_t)n * d.C;
  float* ta_g = d.ws + (int64_t)d.B * d.C + (int64_t)n * d.Cr;
  float* tm_g = ta_g + (int64_t)d.B * d.Cr;
  const float inv_hw = 1.f / ((float)d.H * (float)d.W);
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    float ds = 0.f;
    for (int r = 0; r < a.rows; ++r) ds += a.red[(((int64_t)n * a.rows + r) * 2 + 1) * d.C + c];
    const float s = d.gate[(int64_t)n * d.C + c];
    const float du = ds * s * (1.f - s);
    du_s[c] = du; du_g[c] = du;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int j0 = 0; j0 < d.Cr; j0 += 64) {
    const int j = j0 + lane;
    float acc = 0.f;
    if (j < d.Cr) {
#pragma unroll 8
      for (int c = wave; c < d.C; c += nw) acc = fmaf(du_s[c], d.w2[(int64_t)c * d.Cr + j], acc);
    }
    scratch[wave * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0 && j < d.Cr) {
      float t = 0.f;
      for (int w = 0; w < nw; ++w) t += scratch[w * 64 + lane];
      const float ta = d.ha[(int64_t)n * d.Cr + j] > 0.f ? t : 0.f;
      const float tm = d.hm[(int64_t)n * d.Cr + j] > 0.f ? t : 0.f;
      ta_s[j] = ta; tm_s[j] = tm; ta_g[j] = ta; tm_g[j] = tm;
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    float da = 0.f, dm = 0.f;
#pragma unroll 8
    for (int j = 0; j < d.Cr; ++j) {
      const float w = d.w1[(int64_t)j * d.C + c];
      da = fmaf(ta_s[j], w, da); dm = fmaf(tm_s[j], w, dm);
    }
    d.coefB[(int64_t)n * d.C + c] = da * inv_hw;
    d.dmax[(int64_t)n * d.C + c] = dm;
  }
}

__global__ void cam_bwd_stage2(CamBwdArgs a) {
  const InsarCam& d = a.d;
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const float* du_g = d.ws;
  const float* ta_g = d.ws + (int64_t)d.B * d.C;
  const float* tm_g = ta_g + (int64_t)d.B * d.Cr;
  if (tid >= (int64_t)d.C * d.Cr) return;
  {
    const int c = (int)(tid / d.Cr), j = (int)(tid - (int64_t)c * d.Cr);     // dW2[c][j]
    float acc = 0.f;
    for (int n = 0; n < d.B; ++n)
      acc = fmaf(du_g[(int64_t)n * d.C + c], d.ha[(int64_t)n * d.Cr + j] + d.hm[(int64_t)n * d.Cr + j], acc);
    if (d.accumulate) d.dw2[tid] += acc; else d.dw2[tid] = acc;
  }
  {
    const int j = (int)(tid / d.C), c = (int)(tid - (int64_t)j * d.C);       // dW1[j][c]
    float acc = 0.f;
    for (int n = 0; n < d.B; ++n) {
      acc = fmaf(ta_g[(int64_t)n * d.Cr + j], d.avg[(int64_t)n * d.C + c], acc);
      acc = fmaf(tm_g[(int64_t)n * d.Cr + j], d.mx[(int64_t)n * d.C + c], acc);
    }
    if (d.accumulate) d.dw1[tid] += acc; else d.dw1[tid] = acc;
  }
}

extern "C" int insar_cam_bwd_coef(const InsarCam* d, const float* red, int32_t rows, void* stream) {
  int rc;
  if ((rc = cam_check(d, "insar_cam_bwd_coef"))) return rc;
  if (!red || !d->coefB || !d->dmax || !d->ws || !d->dw1 || !d->dw2) INSAR_FAIL(INSAR_E_ARG, "insar_cam_bwd_coef: null pointer");
  if (rows < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_cam_bwd_coef: rows");
  CamBwdArgs a; a.d = *d; a.red = red; a.rows = rows;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)(d->C + 2 * d->Cr + CAM_COEF_THREADS) * sizeof(float);
  hipLaunchKernelGGL(cam_bwd_stage1, dim3(d->B), dim3(CAM_COEF_THREADS), lds, s, a);
  const int64_t work = (int64_t)d->C * d->Cr;
  hipLaunchKernelGGL(cam_bwd_stage2, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, a);
  INSAR_CHECK_LAUNCH("insar_cam_bwd_coef");
  return INSAR_OK;
}

// dx[n, arg[n][c], c] += dmax[n][c]
template <typename T>
__global__ void cam_scatter_kernel(ActView dx, const float* __restrict__ dmax, const int* __restrict__ arg) {
  const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (tid >= (int64_t)dx.B * dx.c_len) return;
  const int n = (int)(tid / dx.c_len), c = (int)(tid - (int64_t)n * dx.c_len);
  const int idx = arg[tid];
  const int h = idx / dx.W, w = idx - h * dx.W;
  char* p = dx.base + (dx.elem_offset(n, h, w) + c) * (int64_t)sizeof(T);
  if constexpr (sizeof(T) == 2) {
    uint16_t* q = (uint16_t*)p;
    *q = f32_to_bf16(bf16_to_f32(*q) + dmax[tid]);
  } else {
    *(float*)p += dmax[tid];
  }
}

extern "C" int insar_cam_scatter_max(const InsarAct* dx, const float* dmax, const int32_t* arg, void* stream) {
  int rc;
  if ((rc = insar_check_act(dx, "insar_cam_scatter_max", "dx"))) return rc;
  if (!dmax || !arg) INSAR_FAIL(INSAR_E_ARG, "insar_cam_scatter_max: null pointer");
  const int64_t work = (int64_t)dx->B * dx->c_len;
  hipStream_t s = (hipStream_t)stream;
  if (dx->dtype == INSAR_BF16) hipLaunchKernelGGL(cam_scatter_kernel<bf16_t>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, make_view(*dx), dmax, arg);
  else hipLaunchKernelGGL(cam_scatter_kernel<float>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, make_view(*dx), dmax, arg);
  INSAR_CHECK_LAUNCH("insar_cam_scatter_max");
  return INSAR_OK;
}
