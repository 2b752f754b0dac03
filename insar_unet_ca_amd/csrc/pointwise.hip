// Bandwidth-bound kernels of the U-Net-CA path: layout conversion, BatchNorm(+ReLU) apply /
// finalize / backward, SE squeeze + excitation, MaxPool2d(2), partial-sum reductions.
// All are HBM-bound: 16-byte vector accesses, one block per image row, per-channel constants
// hoisted out of the pixel loop, partial sums written as slabs (no float atomics, so results
// are bitwise reproducible).
#include "common.h"

#define PW_THREADS 256

// One block walks image rows r = (n, h); a thread walks 16-byte chunks e of the row:
// w = e / cpp, cc = e % cpp (cpp = chunks per pixel of the slice).
struct RowIter {
  int n, h;
};

template <typename T>
__device__ __forceinline__ const uint4* chunk_ptr(const ActView& v, int n, int h, int w, int cc) {
  return (const uint4*)(v.base + (v.elem_offset(n, h, w) + (int64_t)cc * Chunk<T>::N) * (int64_t)sizeof(T));
}
template <typename T>
__device__ __forceinline__ uint4* chunk_ptr_w(const ActView& v, int n, int h, int w, int cc) {
  return (uint4*)(v.base + (v.elem_offset(n, h, w) + (int64_t)cc * Chunk<T>::N) * (int64_t)sizeof(T));
}

// ---------------------------------------------------------------------------------------------
// pack / unpack (nn.Module boundary)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_scalar_kernel(const float* __restrict__ src, ActView dst, int Csrc) {
  // thread per pixel, loops channels; used when c_len is not a multiple of the chunk width
  const int64_t HW = (int64_t)dst.H * dst.W;
  const int64_t total = (int64_t)dst.B * HW;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    int n = (int)(p / HW);
    int64_t rem = p - (int64_t)n * HW;
    int h = (int)(rem / dst.W), w = (int)(rem - (int64_t)h * dst.W);
    T* out = (T*)(dst.base + dst.elem_offset(n, h, w) * (int64_t)sizeof(T));
    if constexpr (sizeof(T) == 2) {
      if (Csrc == 2 && dst.C == 2) {      // the two-channel InSAR tile: both planes' values in one 4-byte store
        const float v0 = src[((int64_t)n * 2 + 0) * HW + rem], v1 = src[((int64_t)n * 2 + 1) * HW + rem];
        *(uint32_t*)out = pack2_bf16(v0, v1);
        continue;
      }
    }
    for (int c = 0; c < Csrc; ++c) {
      float v = src[((int64_t)n * Csrc + c) * HW + rem];
      if constexpr (sizeof(T) == 2) ((uint16_t*)out)[c] = f32_to_bf16(v);
      else ((float*)out)[c] = v;
    }
  }
}

template <typename T>
__global__ void pack_chunk_kernel(const float* __restrict__ src, ActView dst) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = dst.c_len / CH;
  const int64_t HW = (int64_t)dst.H * dst.W;
  const int rows = dst.B * dst.H;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / dst.H, h = r - n * dst.H;
    const int total = dst.W * cpp;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int cc = e / dst.W, w = e - cc * dst.W;   // w fastest: coalesced plane reads
      float f[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j)
        f[j] = src[((int64_t)n * dst.c_len + cc * CH + j) * HW + (int64_t)h * dst.W + w];
      *chunk_ptr_w<T>(dst, n, h, w, cc) = Chunk<T>::pack(f);
    }
  }
}

template <typename T>
__global__ void unpack_kernel(ActView src, float* __restrict__ dst) {
  const int64_t HW = (int64_t)src.H * src.W;
  const int rows = src.B * src.H;
  constexpr int CH = Chunk<T>::N;
  if (src.c_len % CH == 0) {
    const int cpp = src.c_len / CH;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
      const int n = r / src.H, h = r - n * src.H;
      const int total = src.W * cpp;
      for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int cc = e / src.W, w = e - cc * src.W;
        float f[CH];
        Chunk<T>::unpack(*chunk_ptr<T>(src, n, h, w, cc), f);
#pragma unroll
        for (int j = 0; j < CH; ++j)
          dst[((int64_t)n * src.c_len + cc * CH + j) * HW + (int64_t)h * src.W + w] = f[j];
      }
    }
  } else {
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
      const int n = r / src.H, h = r - n * src.H;
      for (int e = threadIdx.x; e < src.W * src.c_len; e += blockDim.x) {
        const int c = e / src.W, w = e - c * src.W;
        const T* p = (const T*)(src.base + (src.elem_offset(n, h, w) + c) * (int64_t)sizeof(T));
        float v;
        if constexpr (sizeof(T) == 2) v = bf16_to_f32(*(const uint16_t*)p); else v = *(const float*)p;
        dst[((int64_t)n * src.c_len + c) * HW + (int64_t)h * src.W + w] = v;
      }
    }
  }
}

extern "C" int insar_pack_nchw(const float* src, const InsarAct* dst, void* stream) {
  if (!src || !dst || !dst->ptr) INSAR_FAIL(INSAR_E_ARG, "insar_pack_nchw: null pointer");
  ActView v = make_view(*dst);
  const int ch = dst->dtype == INSAR_BF16 ? 8 : 4;
  hipStream_t s = (hipStream_t)stream;
  if (dst->c_len % ch == 0 && dst->c_off % ch == 0 && dst->C % ch == 0) {
    int grid = insar_grid_cap((int64_t)dst->B * dst->H);
    if (dst->dtype == INSAR_BF16) hipLaunchKernelGGL(pack_chunk_kernel<bf16_t>, dim3(grid), dim3(PW_THREADS), 0, s, src, v);
    else hipLaunchKernelGGL(pack_chunk_kernel<float>, dim3(grid), dim3(PW_THREADS), 0, s, src, v);
  } else {
    int64_t total = (int64_t)dst->B * dst->H * dst->W;
    int grid = insar_grid_cap((total + PW_THREADS - 1) / PW_THREADS);
    if (dst->dtype == INSAR_BF16) hipLaunchKernelGGL(pack_scalar_kernel<bf16_t>, dim3(grid), dim3(PW_THREADS), 0, s, src, v, dst->c_len);
    else hipLaunchKernelGGL(pack_scalar_kernel<float>, dim3(grid), dim3(PW_THREADS), 0, s, src, v, dst->c_len);
  }
  INSAR_CHECK_LAUNCH("insar_pack_nchw");
  return INSAR_OK;
}

extern "C" int insar_unpack_nchw(const InsarAct* src, float* dst, void* stream) {
  if (!src || !dst || !src->ptr) INSAR_FAIL(INSAR_E_ARG, "insar_unpack_nchw: null pointer");
  ActView v = make_view(*src);
  int grid = insar_grid_cap((int64_t)src->B * src->H);
  hipStream_t s = (hipStream_t)stream;
  if (src->dtype == INSAR_BF16) hipLaunchKernelGGL(unpack_kernel<bf16_t>, dim3(grid), dim3(PW_THREADS), 0, s, v, dst);
  else hipLaunchKernelGGL(unpack_kernel<float>, dim3(grid), dim3(PW_THREADS), 0, s, v, dst);
  INSAR_CHECK_LAUNCH("insar_unpack_nchw");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// weight re-layout: out[(t*N + n)*K + k] = cast(in[t*st + n*sn + k*sk]); LDS-tiled transpose so
// both the strided reads and the K-contiguous writes stay coalesced.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void weight_prep_kernel(const float* __restrict__ in, T* __restrict__ out, int N, int K,
                                   int64_t st, int64_t sn, int64_t sk) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int n0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  const bool k_fast_in = sk <= sn;                           // which index is contiguous-ish in `in`
  for (int j = ty; j < 32; j += 8) {
    int n = k_fast_in ? n0 + j : n0 + tx;
    int k = k_fast_in ? k0 + tx : k0 + j;
    float v = 0.f;
    if (n < N && k < K) v = in[(int64_t)t * st + (int64_t)n * sn + (int64_t)k * sk];
    if (k_fast_in) tile[j][tx] = v; else tile[tx][j] = v;   // tile[n_local][k_local]
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int n = n0 + j, k = k0 + tx;
    if (n < N && k < K) {
      float v = tile[j][tx];
      int64_t o = ((int64_t)t * N + n) * K + k;
      if constexpr (sizeof(T) == 2) ((uint16_t*)out)[o] = f32_to_bf16(v); else ((float*)out)[o] = v;
    }
  }
}

extern "C" int insar_weight_prep(const float* in, void* out, int32_t dtype, int32_t T, int32_t N, int32_t K,
                                 int64_t st, int64_t sn, int64_t sk, void* stream) {
  if (!in || !out) INSAR_FAIL(INSAR_E_ARG, "insar_weight_prep: null pointer");
  if (T < 1 || N < 1 || K < 1 || T > 65535) INSAR_FAIL(INSAR_E_SHAPE, "insar_weight_prep: bad T/N/K %d/%d/%d", T, N, K);
  dim3 grid((K + 31) / 32, (N + 31) / 32, T);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == INSAR_BF16) hipLaunchKernelGGL(weight_prep_kernel<bf16_t>, grid, dim3(256), 0, s, in, (bf16_t*)out, N, K, st, sn, sk);
  else if (dtype == INSAR_F32) hipLaunchKernelGGL(weight_prep_kernel<float>, grid, dim3(256), 0, s, in, (float*)out, N, K, st, sn, sk);
  else INSAR_FAIL(INSAR_E_DTYPE, "insar_weight_prep: dtype %d", dtype);
  INSAR_CHECK_LAUNCH("insar_weight_prep");
  return INSAR_OK;
}

// Batched form: one launch re-lays out every weight of the network. jobs: int64[njobs][10] =
// {in*, out*, T, N, K, st, sn, sk, first_tile, dtype}; tiles of 32x32 (n x k) per tap, numbered job after job.
template <typename T>
__device__ __forceinline__ void weight_prep_tile(const float* __restrict__ in, T* __restrict__ out, int N, int K, int64_t st,
                                                 int64_t sn, int64_t sk, int t, int n0, int k0, float (*tile)[33]) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const bool k_fast_in = sk <= sn;
  for (int j = ty; j < 32; j += 8) {
    int n = k_fast_in ? n0 + j : n0 + tx;
    int k = k_fast_in ? k0 + tx : k0 + j;
    float v = 0.f;
    if (n < N && k < K) v = in[(int64_t)t * st + (int64_t)n * sn + (int64_t)k * sk];
    if (k_fast_in) tile[j][tx] = v; else tile[tx][j] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int n = n0 + j, k = k0 + tx;
    if (n < N && k < K) {
      float v = tile[j][tx];
      int64_t o = ((int64_t)t * N + n) * K + k;
      if constexpr (sizeof(T) == 2) ((uint16_t*)out)[o] = f32_to_bf16(v); else ((float*)out)[o] = v;
    }
  }
}

__global__ void weight_prep_batch_kernel(const int64_t* __restrict__ jobs, int njobs) {
  __shared__ float tile[32][33];
  __shared__ int sjob;
  if (threadIdx.x == 0) {
    int lo = 0, hi = njobs - 1;                       // last job whose first_tile <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid * 10 + 8] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    sjob = lo;
  }
  __syncthreads();
  const int64_t* j = jobs + sjob * 10;
  const float* in = (const float*)j[0];
  void* out = (void*)j[1];
  const int N = (int)j[3], K = (int)j[4];
  const int local = (int)(blockIdx.x - j[8]);
  const int kt = (K + 31) / 32, nt = (N + 31) / 32;
  const int t = local / (kt * nt), rem = local - t * kt * nt;
  const int n0 = (rem / kt) * 32, k0 = (rem % kt) * 32;
  if (j[9] == INSAR_BF16) weight_prep_tile<bf16_t>(in, (bf16_t*)out, N, K, j[5], j[6], j[7], t, n0, k0, tile);
  else weight_prep_tile<float>(in, (float*)out, N, K, j[5], j[6], j[7], t, n0, k0, tile);
}

extern "C" int insar_weight_prep_batch(const int64_t* jobs, int32_t njobs, int64_t total_tiles, void* stream) {
  if (!jobs || njobs < 1 || total_tiles < 1 || total_tiles > 0x7fffffffLL) INSAR_FAIL(INSAR_E_ARG, "insar_weight_prep_batch: bad arguments");
  hipLaunchKernelGGL(weight_prep_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, jobs, njobs);
  INSAR_CHECK_LAUNCH("insar_weight_prep_batch");
  return INSAR_OK;
}

// Paired form: both GEMM layouts of a weight from ONE read of the master. in[a][b][T] fp32 (T taps
// contiguous: Conv2d (Co,Ci,3,3) -> a=co, b=ci, T=9; ConvTranspose2d (Ci,Co,2,2) -> a=ci, b=co, T=4) goes to
// out_ab[t][a][b] and out_ba[t][b][a] in the compute dtype. A block takes a 32(a) x 32(b) tile with all its
// taps: 32 contiguous runs of 32*T floats in, 2*T*32 rows of 32 elements out.
// jobs: int64[njobs][8] = {in*, out_ab*, out_ba*, A, B, T, first_tile, dtype}; T <= 9.
template <typename TO>
__device__ __forceinline__ void weight_prep_pair_tile(const float* __restrict__ in, TO* __restrict__ oab, TO* __restrict__ oba,
                                                      int A, int B, int T, int a0, int b0, float (*tile)[32 * 9 + 1]) {
  const int nb = min(32, B - b0), na = min(32, A - a0);
  const int rowlen = nb * T;
  const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;       // 8 groups of 32 lanes
  for (int a = grp; a < na; a += 8) {
    const float* src = in + ((int64_t)(a0 + a) * B + b0) * T;
    for (int i = lane; i < rowlen; i += 32) tile[a][i] = src[i];
  }
  __syncthreads();
  for (int r = grp; r < T * 32; r += 8) {
    const int t = r >> 5, q = r & 31;
    if (q < na && lane < nb) {                                      // row (t, a = q), lanes over b
      const float v = tile[q][lane * T + t];
      const int64_t o = ((int64_t)t * A + a0 + q) * B + b0 + lane;
      if constexpr (sizeof(TO) == 2) ((uint16_t*)oab)[o] = f32_to_bf16(v); else ((float*)oab)[o] = v;
    }
    if (q < nb && lane < na) {                                      // row (t, b = q), lanes over a
      const float v = tile[lane][q * T + t];
      const int64_t o = ((int64_t)t * B + b0 + q) * A + a0 + lane;
      if constexpr (sizeof(TO) == 2) ((uint16_t*)oba)[o] = f32_to_bf16(v); else ((float*)oba)[o] = v;
    }
  }
}

__global__ void weight_prep_pair_kernel(const int64_t* __restrict__ jobs, int njobs) {
  __shared__ float tile[32][32 * 9 + 1];
  __shared__ int sjob;
  if (threadIdx.x == 0) {
    int lo = 0, hi = njobs - 1;                       // last job whose first_tile <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid * 8 + 6] <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    sjob = lo;
  }
  __syncthreads();
  const int64_t* j = jobs + sjob * 8;
  const int A = (int)j[3], B = (int)j[4], T = (int)j[5];
  const int local = (int)(blockIdx.x - j[6]);
  const int tb = (B + 31) / 32;
  const int a0 = (local / tb) * 32, b0 = (local % tb) * 32;
  if (j[7] == INSAR_BF16) weight_prep_pair_tile<bf16_t>((const float*)j[0], (bf16_t*)j[1], (bf16_t*)j[2], A, B, T, a0, b0, tile);
  else weight_prep_pair_tile<float>((const float*)j[0], (float*)j[1], (float*)j[2], A, B, T, a0, b0, tile);
}

extern "C" int insar_weight_prep_pair_batch(const int64_t* jobs, int32_t njobs, int64_t total_tiles, void* stream) {
  if (!jobs || njobs < 1 || total_tiles < 1 || total_tiles > 0x7fffffffLL) INSAR_FAIL(INSAR_E_ARG, "insar_weight_prep_pair_batch: bad arguments");
  hipLaunchKernelGGL(weight_prep_pair_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, jobs, njobs);
  INSAR_CHECK_LAUNCH("insar_weight_prep_pair_batch");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// segmented column sums of partial slabs: out[s][c] (+)= sum_{r in split} part[s][r][c]
// grid = (col blocks of 64, row splits, segments); block = 64 cols x 4 row lanes.
// ---------------------------------------------------------------------------------------------
__global__ void colsum_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t rows,
                              int cols, int rows_per_split, int accumulate, int64_t out_split_stride, int64_t ld) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int lane_r = threadIdx.x >> 6;
  const int split = blockIdx.y, seg = blockIdx.z;
  const int64_t r0 = (int64_t)split * rows_per_split;
  int64_t r1 = r0 + rows_per_split; if (r1 > rows) r1 = rows;
  float acc = 0.f;
  if (c < cols) {
    const float* p = part + ((int64_t)seg * rows) * ld + c;
    for (int64_t r = r0 + lane_r; r < r1; r += 4) acc += p[r * ld];
  }
  red[lane_r][threadIdx.x & 63] = acc;
  __syncthreads();
  if (lane_r == 0 && c < cols) {
    float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    float* o = out + (int64_t)split * out_split_stride + (int64_t)seg * cols + c;
    if (accumulate) v += *o;
    *o = v;
  }
}

static int colsum_impl(const char* who, const float* part, float* out, int32_t segments, int64_t rows, int32_t cols, int64_t ld,
                       int32_t accumulate, float* tmp, int64_t tmp_floats, void* stream) {
  if (!part || !out) INSAR_FAIL(INSAR_E_ARG, "%s: null pointer", who);
  if (segments < 1 || rows < 1 || cols < 1 || ld < cols) INSAR_FAIL(INSAR_E_SHAPE, "%s: bad shape", who);
  hipStream_t s = (hipStream_t)stream;
  const int cb = (cols + 63) / 64;
  if (rows <= 256 || !tmp) {
    hipLaunchKernelGGL(colsum_kernel, dim3(cb, 1, segments), dim3(256), 0, s, part, out, rows, cols,
                       (int)(rows > 0x7fffffff ? 0x7fffffff : rows), accumulate, (int64_t)0, ld);
    INSAR_CHECK_LAUNCH(who);
    return INSAR_OK;
  }
  int rps = 128;
  int64_t nsplit = (rows + rps - 1) / rps;
  while (nsplit > 512) { rps *= 2; nsplit = (rows + rps - 1) / rps; }
  if (nsplit * segments * (int64_t)cols > tmp_floats)
    INSAR_FAIL(INSAR_E_WS, "%s: tmp too small (%lld < %lld floats)", who, (long long)tmp_floats,
               (long long)(nsplit * segments * (int64_t)cols));
  hipLaunchKernelGGL(colsum_kernel, dim3(cb, (int)nsplit, segments), dim3(256), 0, s, part, tmp, rows, cols, rps,
                     0, (int64_t)segments * cols, ld);
  // stage 2: treat tmp as [1 segment][nsplit rows][segments*cols]
  const int cols2 = segments * cols;
  hipLaunchKernelGGL(colsum_kernel, dim3((cols2 + 63) / 64, 1, 1), dim3(256), 0, s, tmp, out, nsplit, cols2,
                     (int)nsplit, accumulate, (int64_t)0, (int64_t)cols2);
  INSAR_CHECK_LAUNCH(who);
  return INSAR_OK;
}

// Two-stage when there are many rows: stage 1 writes [nsplit][segments][cols] into `tmp`
// (caller-provided, may be null when rows <= 256), stage 2 folds the splits.
extern "C" int insar_colsum(const float* part, float* out, int32_t segments, int64_t rows, int32_t cols,
                            int32_t accumulate, float* tmp, int64_t tmp_floats, void* stream) {
  return colsum_impl("insar_colsum", part, out, segments, rows, cols, cols, accumulate, tmp, tmp_floats, stream);
}
// The first `cols` columns of rows that are `ld` floats apart (a gradient summed straight into its place in the flat
// gradient buffer: the transposed convs' bias, outc's weight and bias; no staging tensor, no copy).
extern "C" int insar_colsum_ld(const float* part, float* out, int32_t segments, int64_t rows, int32_t cols, int64_t ld,
                               int32_t accumulate, float* tmp, int64_t tmp_floats, void* stream) {
  return colsum_impl("insar_colsum_ld", part, out, segments, rows, cols, ld, accumulate, tmp, tmp_floats, stream);
}

// First stage only: out[split][c] = sum of rows [split*rps, (split+1)*rps) of part[rows][cols]; the consumer
// (insar_bn_finalize) folds the ceil(rows/rps) remaining rows itself.
extern "C" int insar_colsum_partial(const float* part, float* out, int64_t rows, int32_t cols, int32_t rps, void* stream) {
  if (!part || !out) INSAR_FAIL(INSAR_E_ARG, "insar_colsum_partial: null pointer");
  if (rows < 1 || cols < 1 || rps < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_colsum_partial: bad shape");
  const int64_t nsplit = (rows + rps - 1) / rps;
  if (nsplit > 65535) INSAR_FAIL(INSAR_E_SHAPE, "insar_colsum_partial: too many splits");
  hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, (unsigned)nsplit, 1), dim3(256), 0, (hipStream_t)stream, part, out,
                     rows, cols, rps, 0, (int64_t)cols, (int64_t)cols);
  INSAR_CHECK_LAUNCH("insar_colsum_partial");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm finalize (Unet-ChannalAttention.py:82,85; nn.BatchNorm2d training/eval semantics)
// sums[0][c] = sum y_raw, sums[1][c] = sum y_raw^2 over `count` pixels (y_raw = conv w/o bias).
// ---------------------------------------------------------------------------------------------
#define BNF_LANES 16
__global__ void __launch_bounds__(64 * BNF_LANES) bn_finalize_kernel(InsarBnFinalize d) {
  // 64 channels per block, BNF_LANES row lanes per channel (threadIdx = lane*64 + channel)
  __shared__ double fold[2][BNF_LANES][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  if (c == 0 && rl == 0 && d.training && d.num_batches_tracked) *d.num_batches_tracked += 1;
  // the channel's parameters: requested before the fold, used after it
  const bool fin = rl == 0 && c < d.C;
  const float cb = (fin && d.conv_bias) ? d.conv_bias[c] : 0.f;
  const float gamma_c = fin ? d.gamma[c] : 0.f, beta_c = fin ? d.beta[c] : 0.f;
  const float rm_c = (fin && d.running_mean) ? d.running_mean[c] : 0.f, rv_c = (fin && d.running_var) ? d.running_var[c] : 0.f;
  if (d.training) {
    double s1 = 0.0, s2 = 0.0;
    if (c < d.C) {
      // fold the remaining partial rows [rows][2][C]: eight rows of a lane requested before the first is added (this
      // launch sits between a conv and its BN/ReLU pass on the forward chain: it is pure latency — rows / 128 dependent
      // round trips with 16 lanes x 8 in flight, where 8 lanes x 4 took rows / 32)
      int64_t r = rl;
      for (; r + 7 * BNF_LANES < d.rows; r += 8 * BNF_LANES) {
        float v1[8], v2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v1[u] = d.part[((r + BNF_LANES * u) * 2 + 0) * d.C + c];
          v2[u] = d.part[((r + BNF_LANES * u) * 2 + 1) * d.C + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += (double)v1[u]; s2 += (double)v2[u]; }
      }
      for (; r < d.rows; r += BNF_LANES) {
        s1 += (double)d.part[(r * 2 + 0) * d.C + c];
        s2 += (double)d.part[(r * 2 + 1) * d.C + c];
      }
    }
    fold[0][rl][cl] = s1; fold[1][rl][cl] = s2;
    __syncthreads();
  }
  if (!fin) return;
  float mean_raw, invstd;
  if (d.training) {
    const double n = (double)d.count;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < BNF_LANES; ++k) { s1 += fold[0][k][cl]; s2 += fold[1][k][cl]; }
    const double m = s1 / n;
    double var = s2 / n - m * m;
    if (var < 0) var = 0;
    mean_raw = (float)m;
    invstd = (float)(1.0 / sqrt(var + (double)d.eps));
    if (d.running_mean) {
      const double unbiased = n > 1 ? var * n / (n - 1) : var;
      d.running_mean[c] = (1.f - d.momentum) * rm_c + d.momentum * (float)(m + cb);
      d.running_var[c] = (1.f - d.momentum) * rv_c + d.momentum * (float)unbiased;
    }
  } else {
    // eval: y = (y_raw + cb - running_mean) / sqrt(running_var + eps): "mean of y_raw" = rm - cb
    mean_raw = rm_c - cb;
    invstd = 1.f / sqrtf(rv_c + d.eps);
  }
  const float sc = gamma_c * invstd;
  d.scale[c] = sc;
  d.shift[c] = beta_c - mean_raw * sc;
  d.mean[c] = mean_raw;
  d.invstd[c] = invstd;
}

extern "C" int insar_bn_finalize(const InsarBnFinalize* d, void* stream) {
  if (!d || !d->gamma || !d->beta || !d->scale || !d->shift || !d->mean || !d->invstd)
    INSAR_FAIL(INSAR_E_ARG, "insar_bn_finalize: null pointer");
  if (d->training && (!d->part || d->count < 1 || d->rows < 1 || d->rows > 4096))
    INSAR_FAIL(INSAR_E_ARG, "insar_bn_finalize: training needs 1..4096 rows of partial sums");
  if (!d->training && (!d->running_mean || !d->running_var)) INSAR_FAIL(INSAR_E_ARG, "insar_bn_finalize: eval needs running stats");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((d->C + 63) / 64), dim3(64 * BNF_LANES), 0, (hipStream_t)stream, *d);
  INSAR_CHECK_LAUNCH("insar_bn_finalize");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// z = relu(y*scale + shift) * gate[n][c]   (:82-83 / :85-86 and the SE scale :72)
// ---------------------------------------------------------------------------------------------
#define PW_UNROLL 4   // 16-byte chunks in flight per thread and operand (HBM latency hiding)

template <typename T>
__global__ void bn_relu_apply_kernel(ActView y, const float* __restrict__ scale, const float* __restrict__ shift,
                                     const float* __restrict__ gate, ActView dst, int relu) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = y.c_len / CH;
  const int rows = y.B * y.H;
  const int total = y.W * cpp;
  const bool inv = (blockDim.x % cpp) == 0;
  float sc[CH], sh[CH], gt[CH];
  if (inv) {
    // the thread's channel chunk never changes: constants hoisted, PW_UNROLL loads in flight
    const int cc = threadIdx.x % cpp, wstep = blockDim.x / cpp;
#pragma unroll
    for (int j = 0; j < CH; ++j) { sc[j] = scale[cc * CH + j]; sh[j] = shift[cc * CH + j]; gt[j] = 1.f; }
    int n_loaded = -1;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
      const int n = r / y.H, h = r - n * y.H;
      if (gate && n != n_loaded) {
#pragma unroll
        for (int j = 0; j < CH; ++j) gt[j] = gate[(int64_t)n * y.c_len + cc * CH + j];
        n_loaded = n;
      }
      for (int w0 = threadIdx.x / cpp; w0 < y.W; w0 += PW_UNROLL * wstep) {
        uint4 v[PW_UNROLL];
#pragma unroll
        for (int u = 0; u < PW_UNROLL; ++u)
          if (w0 + u * wstep < y.W) v[u] = *chunk_ptr<T>(y, n, h, w0 + u * wstep, cc);
#pragma unroll
        for (int u = 0; u < PW_UNROLL; ++u)
          if (w0 + u * wstep < y.W) {
            float f[CH];
            Chunk<T>::unpack(v[u], f);
#pragma unroll
            for (int j = 0; j < CH; ++j) { const float z = fmaf(f[j], sc[j], sh[j]); f[j] = (relu ? fmaxf(z, 0.f) : z) * gt[j]; }
            *chunk_ptr_w<T>(dst, n, h, w0 + u * wstep, cc) = Chunk<T>::pack(f);
          }
      }
    }
    return;
  }
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / y.H, h = r - n * y.H;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int w = e / cpp, cc = e - w * cpp;
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        sc[j] = scale[cc * CH + j]; sh[j] = shift[cc * CH + j];
        gt[j] = gate ? gate[(int64_t)n * y.c_len + cc * CH + j] : 1.f;
      }
      float f[CH];
      Chunk<T>::unpack(*chunk_ptr<T>(y, n, h, w, cc), f);
#pragma unroll
      for (int j = 0; j < CH; ++j) { const float z = fmaf(f[j], sc[j], sh[j]); f[j] = (relu ? fmaxf(z, 0.f) : z) * gt[j]; }
      *chunk_ptr_w<T>(dst, n, h, w, cc) = Chunk<T>::pack(f);
    }
  }
}

static int check_same_grid(const InsarAct* a, const InsarAct* b, const char* who) {
  if (a->B != b->B || a->H != b->H || a->W != b->W || a->c_len != b->c_len || a->dtype != b->dtype)
    INSAR_FAIL(INSAR_E_SHAPE, "%s: mismatched activation slices", who);
  return INSAR_OK;
}

extern "C" int insar_bn_relu_apply(const InsarAct* y, const float* scale, const float* shift, const float* gate,
                                   const InsarAct* dst, int32_t relu, void* stream) {
  int rc;
  if ((rc = insar_check_act(y, "insar_bn_relu_apply", "y"))) return rc;
  if ((rc = insar_check_act(dst, "insar_bn_relu_apply", "dst"))) return rc;
  if ((rc = check_same_grid(y, dst, "insar_bn_relu_apply"))) return rc;
  if (!scale || !shift) INSAR_FAIL(INSAR_E_ARG, "insar_bn_relu_apply: null scale/shift");
  int grid = insar_grid_cap((int64_t)y->B * y->H);
  hipStream_t s = (hipStream_t)stream;
  if (y->dtype == INSAR_BF16) hipLaunchKernelGGL(bn_relu_apply_kernel<bf16_t>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*y), scale, shift, gate, make_view(*dst), relu);
  else hipLaunchKernelGGL(bn_relu_apply_kernel<float>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*y), scale, shift, gate, make_view(*dst), relu);
  INSAR_CHECK_LAUNCH("insar_bn_relu_apply");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// z = relu(y*scale + shift) * gate AND its 2x2 max-pool in one pass (encoder blocks: :96-97 followed by
// MaxPool2d(2) :106-109). A work-group takes two image rows; a thread a 2x2 window of one channel chunk:
// four loads, four stores of z, one store of the window maximum. Rounding to the storage type is monotone,
// so max-then-round equals the max-pool of the stored z bit for bit.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void bn_relu_apply_pool_kernel(ActView y, const float* __restrict__ scale, const float* __restrict__ shift,
                                          const float* __restrict__ gate, ActView dst, ActView pooled, int relu,
                                          uint8_t* __restrict__ arg) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = y.c_len / CH;
  const int Hp = y.H / 2, Wp2 = y.W / 2;
  const int rows = y.B * Hp;
  const int cc = threadIdx.x % cpp, wstep = blockDim.x / cpp;          // host guarantees blockDim % cpp == 0
  float sc[CH], sh[CH], gt[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) { sc[j] = scale[cc * CH + j]; sh[j] = shift[cc * CH + j]; gt[j] = 1.f; }
  int n_loaded = -1;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / Hp, h2 = r - n * Hp;
    if (gate && n != n_loaded) {
#pragma unroll
      for (int j = 0; j < CH; ++j) gt[j] = gate[(int64_t)n * y.c_len + cc * CH + j];
      n_loaded = n;
    }
    for (int w2 = threadIdx.x / cpp; w2 < Wp2; w2 += wstep) {
      uint4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *chunk_ptr<T>(y, n, 2 * h2 + (q >> 1), 2 * w2 + (q & 1), cc);
      float m[CH], mr[CH];
      uint32_t best[CH];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float f[CH], fr[CH];
        Chunk<T>::unpack(v[q], f);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const float z = fmaf(f[j], sc[j], sh[j]);
          f[j] = (relu ? fmaxf(z, 0.f) : z) * gt[j];
          m[j] = q == 0 ? f[j] : fmaxf(m[j], f[j]);
        }
        const uint4 packed = Chunk<T>::pack(f);
        *chunk_ptr_w<T>(dst, n, 2 * h2 + (q >> 1), 2 * w2 + (q & 1), cc) = packed;
        if (arg) {           // arg-max of the STORED values, insar_maxpool2_bwd's rule: first maximum in scan order, NaN wins
          Chunk<T>::unpack(packed, fr);
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            if (q == 0) { mr[j] = fr[j]; best[j] = 0; }
            else if (fr[j] > mr[j] || fr[j] != fr[j]) { mr[j] = fr[j]; best[j] = q; }
          }
        }
      }
      *chunk_ptr_w<T>(pooled, n, h2, w2, cc) = Chunk<T>::pack(m);
      if (arg) {
        uint8_t* ap = arg + (((int64_t)n * Hp + h2) * Wp2 + w2) * y.c_len + cc * CH;      // [B][H/2][W/2][C] bytes
        if constexpr (CH == 8) {
          uint2 u;
          u.x = best[0] | (best[1] << 8) | (best[2] << 16) | (best[3] << 24);
          u.y = best[4] | (best[5] << 8) | (best[6] << 16) | (best[7] << 24);
          *(uint2*)ap = u;
        } else {
          *(uint32_t*)ap = best[0] | (best[1] << 8) | (best[2] << 16) | (best[3] << 24);
        }
      }
    }
  }
}

static int launch_apply_pool(const InsarAct* y, const float* scale, const float* shift, const float* gate,
                             const InsarAct* dst, const InsarAct* pooled, int32_t relu, uint8_t* arg, void* stream) {
  int rc;
  if ((rc = insar_check_act(y, "insar_bn_relu_apply_pool", "y"))) return rc;
  if ((rc = insar_check_act(dst, "insar_bn_relu_apply_pool", "dst"))) return rc;
  if ((rc = insar_check_act(pooled, "insar_bn_relu_apply_pool", "pooled"))) return rc;
  if ((rc = check_same_grid(y, dst, "insar_bn_relu_apply_pool"))) return rc;
  if (!scale || !shift) INSAR_FAIL(INSAR_E_ARG, "insar_bn_relu_apply_pool: null scale/shift");
  if ((y->H & 1) || (y->W & 1) || pooled->B != y->B || pooled->H != y->H / 2 || pooled->W != y->W / 2 ||
      pooled->c_len != y->c_len || pooled->dtype != y->dtype)
    INSAR_FAIL(INSAR_E_SHAPE, "insar_bn_relu_apply_pool: pooled slice must be (B, H/2, W/2, C) of an even grid");
  const int ch = y->dtype == INSAR_BF16 ? 8 : 4;
  if (PW_THREADS % (y->c_len / ch)) INSAR_FAIL(INSAR_E_SHAPE, "insar_bn_relu_apply_pool: C=%d unsupported", y->c_len);
  int grid = insar_grid_cap((int64_t)y->B * (y->H / 2));
  hipStream_t s = (hipStream_t)stream;
  if (y->dtype == INSAR_BF16) hipLaunchKernelGGL(bn_relu_apply_pool_kernel<bf16_t>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*y), scale, shift, gate, make_view(*dst), make_view(*pooled), relu, arg);
  else hipLaunchKernelGGL(bn_relu_apply_pool_kernel<float>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*y), scale, shift, gate, make_view(*dst), make_view(*pooled), relu, arg);
  INSAR_CHECK_LAUNCH("insar_bn_relu_apply_pool");
  return INSAR_OK;
}

extern "C" int insar_bn_relu_apply_pool(const InsarAct* y, const float* scale, const float* shift, const float* gate,
                                        const InsarAct* dst, const InsarAct* pooled, int32_t relu, void* stream) {
  return launch_apply_pool(y, scale, shift, gate, dst, pooled, relu, nullptr, stream);
}

// The same pass, also recording WHICH element of each 2x2 window is the maximum (insar_maxpool2_bwd's rule on the
// stored values): arg[B][H/2][W/2][C] bytes in 0..3 = 2*(row parity) + (column parity). With it the pooled gradient is
// routed inside the BatchNorm-backward passes of this unit (insar_bnrelu_bwd_reduce_pool / _apply_pool) and
// insar_maxpool2_bwd (one read of the full-resolution activation, a read-modify-write of its gradient) is not needed.
extern "C" int insar_bn_relu_apply_pool_arg(const InsarAct* y, const float* scale, const float* shift, const float* gate,
                                            const InsarAct* dst, const InsarAct* pooled, uint8_t* arg, int32_t relu,
                                            void* stream) {
  if (!arg) INSAR_FAIL(INSAR_E_ARG, "insar_bn_relu_apply_pool_arg: null arg map");
  return launch_apply_pool(y, scale, shift, gate, dst, pooled, relu, arg, stream);
}

// ---------------------------------------------------------------------------------------------
// z = relu(y*scale + shift) * gate of the LAST unit, consumed on the spot by the 1x1 output conv (outc,
// Unet-ChannalAttention.py:125,162): logits[n,k,h,w] = bias[k] + sum_c round_T(z[c]) * W[k][c], fp32 NCHW. z itself is
// not written: backward needs only y (insar_conv1x1_out_wgrad_y recomputes z). The 8 (bf16) / 16 (fp32) lanes that
// share a pixel reduce their partial dot products exactly as conv1x1_out_fwd_kernel does (fma chain over the chunk,
// xor-shuffle tree over the chunks), so the logits are bitwise those of apply + insar_conv1x1_out_fwd.
// ---------------------------------------------------------------------------------------------
template <typename T, int KM>
__global__ void __launch_bounds__(512) bn_relu_apply_outc_kernel(ActView y, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ gate,
                                                                 const float* __restrict__ wout, const float* __restrict__ bias,
                                                                 float* __restrict__ logits, int K, int relu) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = y.c_len / CH;                       // host guarantees: power of two, <= 64, divides blockDim
  const int rows = y.B * y.H;
  const int cc = threadIdx.x % cpp, wstep = blockDim.x / cpp;
  const int64_t HW = (int64_t)y.H * y.W;
  float sc[CH], sh[CH], gt[CH], wk[KM][CH], bk[KM];
#pragma unroll
  for (int j = 0; j < CH; ++j) { sc[j] = scale[cc * CH + j]; sh[j] = shift[cc * CH + j]; gt[j] = 1.f; }
#pragma unroll
  for (int k = 0; k < KM; ++k) {
    bk[k] = (bias && k < K) ? bias[k] : 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) wk[k][j] = k < K ? wout[k * y.c_len + cc * CH + j] : 0.f;
  }
  const int wpad = (y.W + wstep - 1) / wstep * wstep;      // whole waves take part in the shuffles
  int n_loaded = -1;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / y.H, h = r - n * y.H;
    if (gate && n != n_loaded) {
#pragma unroll
      for (int j = 0; j < CH; ++j) gt[j] = gate[(int64_t)n * y.c_len + cc * CH + j];
      n_loaded = n;
    }
    for (int w0 = threadIdx.x / cpp; w0 < wpad; w0 += PW_UNROLL * wstep) {
      uint4 v[PW_UNROLL];
#pragma unroll
      for (int u = 0; u < PW_UNROLL; ++u)
        if (w0 + u * wstep < y.W) v[u] = *chunk_ptr<T>(y, n, h, w0 + u * wstep, cc);
#pragma unroll
      for (int u = 0; u < PW_UNROLL; ++u) {
        const int w = w0 + u * wstep;
        if (w >= wpad) continue;
        const bool ok = w < y.W;
        float f[CH], z[CH];
        if (ok) Chunk<T>::unpack(v[u], f);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const float t = ok ? fmaf(f[j], sc[j], sh[j]) : 0.f;
          f[j] = (relu ? fmaxf(t, 0.f) : t) * gt[j];
        }
        Chunk<T>::unpack(Chunk<T>::pack(f), z);            // the rounding a stored z would have had
#pragma unroll
        for (int k = 0; k < KM; ++k)
          if (k < K) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < CH; ++j) a = fmaf(z[j], wk[k][j], a);
            for (int o = 1; o < cpp; o <<= 1) a += __shfl_xor(a, o, 64);
            if (ok && cc == 0) logits[((int64_t)n * K + k) * HW + (int64_t)h * y.W + w] = a + bk[k];
          }
      }
    }
  }
}

extern "C" int insar_bn_relu_apply_outc(const InsarAct* y, const float* scale, const float* shift, const float* gate,
                                        const float* wout, const float* bias, float* logits, int32_t K, int32_t relu,
                                        void* stream) {
  int rc;
  if ((rc = insar_check_act(y, "insar_bn_relu_apply_outc", "y"))) return rc;
  if (!scale || !shift || !wout || !logits) INSAR_FAIL(INSAR_E_ARG, "insar_bn_relu_apply_outc: null pointer");
  if (K < 1 || K > 4) INSAR_FAIL(INSAR_E_SHAPE, "insar_bn_relu_apply_outc: num_classes=%d must be 1..4", K);
  const int ch = y->dtype == INSAR_BF16 ? 8 : 4;
  const int cpp = y->c_len / ch;
  if (y->c_len % ch || cpp < 1 || cpp > 64 || (cpp & (cpp - 1)) || PW_THREADS % cpp)
    INSAR_FAIL(INSAR_E_SHAPE, "insar_bn_relu_apply_outc: C=%d unsupported", y->c_len);
  int grid = insar_grid_cap((int64_t)y->B * y->H);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_AO(TT, KK) hipLaunchKernelGGL((bn_relu_apply_outc_kernel<TT, KK>), dim3(grid), dim3(PW_THREADS), 0, s, make_view(*y), scale, shift, gate, wout, bias, logits, K, relu)
  if (y->dtype == INSAR_BF16) { if (K <= 2) LAUNCH_AO(bf16_t, 2); else LAUNCH_AO(bf16_t, 4); }
  else { if (K <= 2) LAUNCH_AO(float, 2); else LAUNCH_AO(float, 4); }
#undef LAUNCH_AO
  INSAR_CHECK_LAUNCH("insar_bn_relu_apply_outc");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// "Virtual" gradient of the 1x1 output conv (outc, Unet-ChannalAttention.py:125,162): the gradient wrt outc's input,
//   g[n,h,w,c] = round_T( sum_k dlogits[n,k,h,w] * W[k][c] )        (k ascending, fma chain, as conv1x1_out_bwd stores it)
// costs 2K multiply-adds per element, so the BatchNorm-backward reduce and apply passes of the unit that feeds outc
// recompute it from the K-channel fp32 dlogits instead of reading a materialised 64-channel tensor twice (and
// conv1x1_out_bwd does not write it): 3 activation-sized HBM passes less per step, bitwise the same numbers.
// ---------------------------------------------------------------------------------------------
// "Virtual" sum of the two gradients that meet at an encoder block's output: the skip gradient g (stored) and the
// max-pool gradient, routed by the forward pass's arg-max map:  gg = round_T(g + (arg == position ? dp : 0))
// (the expression insar_maxpool2_bwd evaluates when it accumulates into g).
struct PoolGrad {
  ActView dp;           // gradient wrt the pooled activation (B, H/2, W/2, C)
  const uint8_t* arg;   // [B][H/2][W/2][C] bytes 0..3, or null: no pooled gradient
};
template <typename T>
__device__ __forceinline__ void pool_grad_chunk(const uint4& vdp, const uint8_t* argp, int pos, float (&gg)[Chunk<T>::N]) {
  constexpr int CH = Chunk<T>::N;
  float dpf[CH];
  Chunk<T>::unpack(vdp, dpf);
  uint32_t a[CH];
  if constexpr (CH == 8) {
    const uint2 u = *(const uint2*)argp;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = (u.x >> (8 * j)) & 0xffu; a[4 + j] = (u.y >> (8 * j)) & 0xffu; }
  } else {
    const uint32_t u = *(const uint32_t*)argp;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = (u >> (8 * j)) & 0xffu;
  }
  float o[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) o[j] = gg[j] + ((int)a[j] == pos ? dpf[j] : 0.f);
  Chunk<T>::unpack(Chunk<T>::pack(o), gg);
}

#define OG_MAXK 4
struct OutcGrad {
  const float* dl;      // [B][K][H][W] fp32, or null: read the gradient tensor
  const float* w;       // [K][C] fp32
  int K;
  // reduce pass only, nullable: also produce the 1x1 output conv's own parameter gradients from the same read of y —
  // wpart[part row][K*C + K] = (sum dl[k] * z[c], sum dl[k]) over the row part, z = round_T(relu(y*scale+shift) * gate[n])
  // (the arithmetic of conv1x1_out_bwd_kernel<T, true>, csrc/direct.hip); gate [B][C] or null
  const float* gate;
  float* wpart;
};
template <typename T, int KM>
__device__ __forceinline__ void outc_grad_chunk(const float (&dlv)[KM], const float (&wk)[KM][Chunk<T>::N], int K,
                                                float (&gg)[Chunk<T>::N]) {
  constexpr int CH = Chunk<T>::N;
  float o[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) o[j] = 0.f;
#pragma unroll
  for (int k = 0; k < KM; ++k)
    if (k < K) {
#pragma unroll
      for (int j = 0; j < CH; ++j) o[j] = fmaf(dlv[k], wk[k][j], o[j]);
    }
  Chunk<T>::unpack(Chunk<T>::pack(o), gg);          // the rounding a stored gradient tensor would have had
}

// ---------------------------------------------------------------------------------------------
// Row reductions. Both produce part[(n*H + h)][2][C] per-row partial sums over w:
//   se_squeeze      : q0 = mask,          q1 = mask * y          (mask = y*scale+shift > 0)
//   bnrelu_bwd_reduce: q0 = dout * mask,  q1 = dout * mask * y
// Cross-thread reduction through LDS (threads that own the same channel chunk).
// ---------------------------------------------------------------------------------------------
// VK = 0: dout read from memory; VK = 2 / 4: dout recomputed from dlogits with K <= VK classes (own instantiations with
// their own register budget: the plain passes keep theirs)
template <typename T, bool WITH_G, int VK = 0, bool VP = false>
__global__ void __launch_bounds__((VK || VP) ? 512 : 1024) row_reduce_kernel(ActView g, ActView y, const float* __restrict__ scale,
                                  const float* __restrict__ shift, float* __restrict__ part, int relu, int rpp,
                                  OutcGrad og, PoolGrad pg) {
  constexpr int CH = Chunk<T>::N;
  __shared__ float red[PW_THREADS][2 * CH + 1];
  const int cpp = y.c_len / CH;
  const int ppi = (y.H + rpp - 1) / rpp;          // partials per image
  const int nparts = y.B * ppi;
  const bool inv = (blockDim.x % cpp) == 0;
  for (int r = blockIdx.x; r < nparts; r += gridDim.x) {
    const int n = r / ppi, h0 = (r - n * ppi) * rpp;
    const int h1 = min(y.H, h0 + rpp);
    if (inv) {
      float a0[CH], a1[CH], sc[CH], sh[CH];
      const int cc = threadIdx.x % cpp;
#pragma unroll
      for (int j = 0; j < CH; ++j) { a0[j] = 0.f; a1[j] = 0.f; sc[j] = scale[cc * CH + j]; sh[j] = shift[cc * CH + j]; }
      constexpr bool virt = WITH_G && VK > 0;
      constexpr int KM = virt ? VK : 1;
      float wk[KM][CH];
      if constexpr (virt) {
#pragma unroll
        for (int k = 0; k < KM; ++k)
#pragma unroll
          for (int j = 0; j < CH; ++j) wk[k][j] = k < og.K ? og.w[k * y.c_len + cc * CH + j] : 0.f;
      }
      const int64_t HW = (int64_t)y.H * y.W;
      const int wstep = blockDim.x / cpp;
      // (virt, og.wpart) the output conv's weight / bias gradient of this row part
      const bool wgrad = virt && og.wpart != nullptr;
      float aw[KM][CH], ab[KM], zg[CH];
      if constexpr (virt) {
#pragma unroll
        for (int k = 0; k < KM; ++k) {
          ab[k] = 0.f;
#pragma unroll
          for (int j = 0; j < CH; ++j) aw[k][j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) zg[j] = (wgrad && og.gate) ? og.gate[(int64_t)n * y.c_len + cc * CH + j] : 1.f;
      }
      for (int h = h0; h < h1; ++h) {
        for (int w0 = threadIdx.x / cpp; w0 < y.W; w0 += PW_UNROLL * wstep) {
          uint4 vy[PW_UNROLL], vg[PW_UNROLL], vp[VP ? PW_UNROLL : 1];
          float dlv[PW_UNROLL][KM];
#pragma unroll
          for (int u = 0; u < PW_UNROLL; ++u)
            if (w0 + u * wstep < y.W) {
              vy[u] = *chunk_ptr<T>(y, n, h, w0 + u * wstep, cc);
              if constexpr (virt) {
#pragma unroll
                for (int k = 0; k < KM; ++k)
                  if (k < og.K) dlv[u][k] = og.dl[((int64_t)n * og.K + k) * HW + (int64_t)h * y.W + w0 + u * wstep];
              } else if constexpr (WITH_G) {
                vg[u] = *chunk_ptr<T>(g, n, h, w0 + u * wstep, cc);
                if constexpr (VP) vp[u] = *chunk_ptr<T>(pg.dp, n, h >> 1, (w0 + u * wstep) >> 1, cc);
              }
            }
#pragma unroll
          for (int u = 0; u < PW_UNROLL; ++u)
            if (w0 + u * wstep < y.W) {
              float f[CH], gg[CH];
              Chunk<T>::unpack(vy[u], f);
              if constexpr (virt) outc_grad_chunk<T, KM>(dlv[u], wk, og.K, gg);
              else if constexpr (WITH_G) Chunk<T>::unpack(vg[u], gg);
              if constexpr (VP) {
                const int w = w0 + u * wstep;
                pool_grad_chunk<T>(vp[u], pg.arg + ((((int64_t)n * (y.H >> 1) + (h >> 1)) * (y.W >> 1) + (w >> 1)) * y.c_len + cc * CH),
                                   ((h & 1) << 1) | (w & 1), gg);
              }
#pragma unroll
              for (int j = 0; j < CH; ++j) {
                const bool on = !relu || fmaf(f[j], sc[j], sh[j]) > 0.f;
                const float m = WITH_G ? (on ? gg[j] : 0.f) : (on ? 1.f : 0.f);
                a0[j] += m; a1[j] = fmaf(m, f[j], a1[j]);
              }
              if constexpr (virt) {
                if (wgrad) {
                  float zz[CH], zr[CH];
#pragma unroll
                  for (int j = 0; j < CH; ++j) {
                    const float t = fmaf(f[j], sc[j], sh[j]);
                    zz[j] = (relu ? fmaxf(t, 0.f) : t) * zg[j];
                  }
                  Chunk<T>::unpack(Chunk<T>::pack(zz), zr);        // the rounding the stored activation would have had
#pragma unroll
                  for (int k = 0; k < KM; ++k)
                    if (k < og.K) {
#pragma unroll
                      for (int j = 0; j < CH; ++j) aw[k][j] = fmaf(dlv[u][k], zr[j], aw[k][j]);
                      if (cc == 0) ab[k] += dlv[u][k];
                    }
                }
              }
            }
        }
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) { red[threadIdx.x][j] = a0[j]; red[threadIdx.x][CH + j] = a1[j]; }
      __syncthreads();
      // thread t < cpp*CH*2 handles output element (q, c)
      for (int o = threadIdx.x; o < 2 * y.c_len; o += blockDim.x) {
        const int q = o / y.c_len, c = o - q * y.c_len;
        const int occ = c / CH, j = c - occ * CH;
        float s = 0.f;
        for (int t = occ; t < blockDim.x; t += cpp) s += red[t][q * CH + j];
        part[((int64_t)r * 2 + q) * y.c_len + c] = s;
      }
      __syncthreads();
      if constexpr (virt) {
        if (wgrad) {
          const int pw = og.K * y.c_len + og.K;
#pragma unroll
          for (int k = 0; k < KM; ++k)
            if (k < og.K) {
#pragma unroll
              for (int j = 0; j < CH; ++j) red[threadIdx.x][j] = aw[k][j];
              red[threadIdx.x][CH] = ab[k];
              __syncthreads();
              for (int o = threadIdx.x; o <= y.c_len; o += blockDim.x) {
                float s = 0.f;
                if (o < y.c_len) {
                  const int occ = o / CH, j = o - occ * CH;
                  for (int t = occ; t < blockDim.x; t += cpp) s += red[t][j];
                  og.wpart[(int64_t)r * pw + k * y.c_len + o] = s;
                } else {
                  for (int t = 0; t < blockDim.x; t += cpp) s += red[t][CH];
                  og.wpart[(int64_t)r * pw + og.K * y.c_len + k] = s;
                }
              }
              __syncthreads();
            }
        }
      }
    } else {
      // generic (slow) path: thread per channel
      for (int c = threadIdx.x; c < y.c_len; c += blockDim.x) {
        float s0 = 0.f, s1 = 0.f;
        const float sc = scale[c], sh = shift[c];
        for (int h = h0; h < h1; ++h)
          for (int w = 0; w < y.W; ++w) {
            const T* py = (const T*)(y.base + (y.elem_offset(n, h, w) + c) * (int64_t)sizeof(T));
            float f, gg = 1.f;
            if constexpr (sizeof(T) == 2) f = bf16_to_f32(*(const uint16_t*)py); else f = *(const float*)py;
            if constexpr (WITH_G) {
              const T* pg = (const T*)(g.base + (g.elem_offset(n, h, w) + c) * (int64_t)sizeof(T));
              if constexpr (sizeof(T) == 2) gg = bf16_to_f32(*(const uint16_t*)pg); else gg = *(const float*)pg;
            }
            const float m = (!relu || fmaf(f, sc, sh) > 0.f) ? gg : 0.f;
            s0 += m; s1 = fmaf(m, f, s1);
          }
        part[((int64_t)r * 2 + 0) * y.c_len + c] = s0;
        part[((int64_t)r * 2 + 1) * y.c_len + c] = s1;
      }
    }
  }
}

extern "C" int insar_se_squeeze(const InsarAct* y, const float* scale, const float* shift, float* part, int32_t relu,
                                int32_t rows_per_part, void* stream) {
  int rc;
  if ((rc = insar_check_act(y, "insar_se_squeeze", "y"))) return rc;
  if (!scale || !shift || !part) INSAR_FAIL(INSAR_E_ARG, "insar_se_squeeze: null pointer");
  if (rows_per_part < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_se_squeeze: rows_per_part");
  const int rpp = rows_per_part;
  int grid = insar_grid_cap((int64_t)y->B * ((y->H + rpp - 1) / rpp));
  hipStream_t s = (hipStream_t)stream;
  ActView v = make_view(*y);
  if (y->dtype == INSAR_BF16) hipLaunchKernelGGL((row_reduce_kernel<bf16_t, false>), dim3(grid), dim3(PW_THREADS), 0, s, v, v, scale, shift, part, relu, rpp, OutcGrad{nullptr, nullptr, 0, nullptr, nullptr}, PoolGrad{ActView{}, nullptr});
  else hipLaunchKernelGGL((row_reduce_kernel<float, false>), dim3(grid), dim3(PW_THREADS), 0, s, v, v, scale, shift, part, relu, rpp, OutcGrad{nullptr, nullptr, 0, nullptr, nullptr}, PoolGrad{ActView{}, nullptr});
  INSAR_CHECK_LAUNCH("insar_se_squeeze");
  return INSAR_OK;
}

static int check_outc_grad(const InsarAct* y, const float* dlogits, const float* wout, int32_t K, const char* who) {
  if (!dlogits || !wout) INSAR_FAIL(INSAR_E_ARG, "%s: null dlogits / weight", who);
  if (K < 1 || K > OG_MAXK) INSAR_FAIL(INSAR_E_SHAPE, "%s: num_classes=%d must be 1..%d", who, K, OG_MAXK);
  const int ch = y->dtype == INSAR_BF16 ? 8 : 4;
  if (y->c_len % ch || PW_THREADS % (y->c_len / ch)) INSAR_FAIL(INSAR_E_SHAPE, "%s: C=%d unsupported", who, y->c_len);
  return INSAR_OK;
}

static int check_pool_grad(const InsarAct* y, const InsarAct* dp, const uint8_t* arg, const char* who) {
  int rc;
  if (!dp || !arg) INSAR_FAIL(INSAR_E_ARG, "%s: null pooled gradient / arg-max map", who);
  if ((rc = insar_check_act(dp, who, "dpooled"))) return rc;
  if ((y->H & 1) || (y->W & 1) || dp->B != y->B || dp->H != y->H / 2 || dp->W != y->W / 2 || dp->c_len != y->c_len || dp->dtype != y->dtype)
    INSAR_FAIL(INSAR_E_SHAPE, "%s: pooled gradient must be (B, H/2, W/2, C) of an even grid", who);
  const int ch = y->dtype == INSAR_BF16 ? 8 : 4;
  if (y->c_len % ch || PW_THREADS % (y->c_len / ch)) INSAR_FAIL(INSAR_E_SHAPE, "%s: C=%d unsupported", who, y->c_len);
  return INSAR_OK;
}

static int launch_bwd_reduce(const char* who, const InsarAct* dout, const InsarAct* y, const float* scale, const float* shift,
                             float* part, int32_t relu, int32_t rows_per_part, OutcGrad og, PoolGrad pg, void* stream) {
  int rc;
  if ((rc = insar_check_act(y, who, "y"))) return rc;
  if (!og.dl) {
    if ((rc = insar_check_act(dout, who, "dout"))) return rc;
    if ((rc = check_same_grid(y, dout, who))) return rc;
  }
  if (!scale || !shift || !part) INSAR_FAIL(INSAR_E_ARG, "%s: null pointer", who);
  if (rows_per_part < 1) INSAR_FAIL(INSAR_E_SHAPE, "%s: rows_per_part", who);
  const int rpp = rows_per_part;
  int grid = insar_grid_cap((int64_t)y->B * ((y->H + rpp - 1) / rpp));
  hipStream_t s = (hipStream_t)stream;
  const ActView vy = make_view(*y), vg = og.dl ? vy : make_view(*dout);
  if (og.dl && og.K <= 2) {
    if (y->dtype == INSAR_BF16) hipLaunchKernelGGL((row_reduce_kernel<bf16_t, true, 2>), dim3(grid), dim3(PW_THREADS), 0, s, vg, vy, scale, shift, part, relu, rpp, og, pg);
    else hipLaunchKernelGGL((row_reduce_kernel<float, true, 2>), dim3(grid), dim3(PW_THREADS), 0, s, vg, vy, scale, shift, part, relu, rpp, og, pg);
  } else if (og.dl) {
    if (y->dtype == INSAR_BF16) hipLaunchKernelGGL((row_reduce_kernel<bf16_t, true, OG_MAXK>), dim3(grid), dim3(PW_THREADS), 0, s, vg, vy, scale, shift, part, relu, rpp, og, pg);
    else hipLaunchKernelGGL((row_reduce_kernel<float, true, OG_MAXK>), dim3(grid), dim3(PW_THREADS), 0, s, vg, vy, scale, shift, part, relu, rpp, og, pg);
  } else if (pg.arg) {
    if (y->dtype == INSAR_BF16) hipLaunchKernelGGL((row_reduce_kernel<bf16_t, true, 0, true>), dim3(grid), dim3(PW_THREADS), 0, s, vg, vy, scale, shift, part, relu, rpp, og, pg);
    else hipLaunchKernelGGL((row_reduce_kernel<float, true, 0, true>), dim3(grid), dim3(PW_THREADS), 0, s, vg, vy, scale, shift, part, relu, rpp, og, pg);
  } else {
    if (y->dtype == INSAR_BF16) hipLaunchKernelGGL((row_reduce_kernel<bf16_t, true>), dim3(grid), dim3(PW_THREADS), 0, s, vg, vy, scale, shift, part, relu, rpp, og, pg);
    else hipLaunchKernelGGL((row_reduce_kernel<float, true>), dim3(grid), dim3(PW_THREADS), 0, s, vg, vy, scale, shift, part, relu, rpp, og, pg);
  }
  INSAR_CHECK_LAUNCH(who);
  return INSAR_OK;
}

extern "C" int insar_bnrelu_bwd_reduce(const InsarAct* dout, const InsarAct* y, const float* scale, const float* shift,
                                       float* part, int32_t relu, int32_t rows_per_part, void* stream) {
  return launch_bwd_reduce("insar_bnrelu_bwd_reduce", dout, y, scale, shift, part, relu, rows_per_part,
                           OutcGrad{nullptr, nullptr, 0, nullptr, nullptr}, PoolGrad{ActView{}, nullptr}, stream);
}

// The same sums with dout = round(dskip + (arg == position ? dpooled : 0)): the gradient that reaches an encoder block's
// output through its skip connection plus the one routed back through MaxPool2d(2) (Unet-ChannalAttention.py:106-109)
// by the arg-max map of insar_bn_relu_apply_pool_arg, summed on the fly instead of by insar_maxpool2_bwd.
extern "C" int insar_bnrelu_bwd_reduce_pool(const InsarAct* dskip, const InsarAct* dpooled, const uint8_t* arg, const InsarAct* y,
                                            const float* scale, const float* shift, float* part, int32_t relu,
                                            int32_t rows_per_part, void* stream) {
  int rc;
  if ((rc = insar_check_act(y, "insar_bnrelu_bwd_reduce_pool", "y"))) return rc;
  if ((rc = check_pool_grad(y, dpooled, arg, "insar_bnrelu_bwd_reduce_pool"))) return rc;
  return launch_bwd_reduce("insar_bnrelu_bwd_reduce_pool", dskip, y, scale, shift, part, relu, rows_per_part,
                           OutcGrad{nullptr, nullptr, 0, nullptr, nullptr}, PoolGrad{make_view(*dpooled), arg}, stream);
}

// The same sums with dout = gradient of the 1x1 output conv's input, recomputed from dlogits [B][K][H][W] (fp32) and
// the conv's weight [K][C] instead of read from memory (see OutcGrad above).
// wpart (nullable, with gate [B][C] nullable): also write the output conv's parameter-gradient partials of every row part,
// wpart[B * ceil(H / rows_per_part)][K*C + K] in the layout of insar_conv1x1_out_wgrad_y's `part` (insar_colsum folds
// them): the weight gradient then needs no pass of its own over y.
extern "C" int insar_bnrelu_bwd_reduce_outc(const float* dlogits, const float* wout, int32_t K, const InsarAct* y,
                                            const float* scale, const float* shift, float* part, int32_t relu,
                                            int32_t rows_per_part, const float* gate, float* wpart, void* stream) {
  int rc;
  if ((rc = insar_check_act(y, "insar_bnrelu_bwd_reduce_outc", "y"))) return rc;
  if ((rc = check_outc_grad(y, dlogits, wout, K, "insar_bnrelu_bwd_reduce_outc"))) return rc;
  return launch_bwd_reduce("insar_bnrelu_bwd_reduce_outc", nullptr, y, scale, shift, part, relu, rows_per_part,
                           OutcGrad{dlogits, wout, K, gate, wpart}, PoolGrad{ActView{}, nullptr}, stream);
}

// Sum `rows` rows of a [rows][cols] fp32 slab into out[cols] (LDS), all threads of the block cooperating:
// columns across lanes (coalesced), row lanes when cols < blockDim. `scratch` needs blockDim floats.
#define COEF_THREADS 1024   // per-image coefficient kernels (se_excite, bnse_bwd_stage1): 16 waves

__device__ __forceinline__ float strided_sum16(const float* __restrict__ p, int first, int rows, int step, int64_t ld) {
  // sixteen independent partial sums so that the loads of one thread overlap: these folds sit at the head of latency-bound
  // launches (se_excite on the forward chain), where every dependent round trip is ~0.5 us (fixed order => deterministic)
  float a[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) a[u] = 0.f;
  int r = first;
  for (; r + 15 * step < rows; r += 16 * step) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(int64_t)(r + u * step) * ld];
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] += v[u];
  }
  if (r + 7 * step < rows) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(r + u * step) * ld];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += v[u];
    r += 8 * step;
  }
  if (r + 3 * step < rows) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = p[(int64_t)(r + u * step) * ld];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] += v[u];
    r += 4 * step;
  }
  for (; r < rows; r += step) a[0] += p[(int64_t)r * ld];
  return (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
         (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
}

__device__ __forceinline__ void block_colsum(const float* __restrict__ slab, int rows, int cols, float* out, float* scratch) {
  const int nt = blockDim.x;
  if (cols >= nt) {
    for (int c = threadIdx.x; c < cols; c += nt) out[c] = strided_sum16(slab + c, 0, rows, 1, cols);
    __syncthreads();
  } else {
    const int lanes = nt / cols;                       // row lanes; threads beyond lanes*cols idle
    const int c = threadIdx.x % cols, rl = threadIdx.x / cols;
    scratch[threadIdx.x] = rl < lanes ? strided_sum16(slab + c, rl, rows, lanes, cols) : 0.f;
    __syncthreads();
    if (threadIdx.x < cols) {
      float t = 0.f;
      for (int l = 0; l < lanes; ++l) t += scratch[l * cols + threadIdx.x];
      out[threadIdx.x] = t;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// SE excitation (Unet-ChannalAttention.py:54-59,65-68): one block per image.
// pooled[n][0][c] = sum mask, pooled[n][1][c] = sum mask*y  =>  mean_hw(z) = (scale*q1 + shift*q0)/HW
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(COEF_THREADS) se_excite_kernel(InsarSeFwd d) {
  extern __shared__ float sm[];
  float* sq = sm;            // [C]
  float* hid = sm + d.C;     // [Cr]
  float* pool = hid + d.Cr;  // [2C]
  float* scratch = pool + 2 * d.C;   // [blockDim]
  // grid (B, S): the S work-groups of an image all fold its squeeze rows and run the first Linear (cheap,
  // L2-served), and share the rows of the second Linear: this kernel is latency-bound, and with one
  // work-group per image only B of the 256 CUs had anything to do (56 us at C = 1024).
  const int n = blockIdx.x;
  const bool writer = blockIdx.y == 0;
  const float inv_hw = 1.f / ((float)d.H * (float)d.W);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int per = (d.C + gridDim.y - 1) / gridDim.y;
  const int cbeg = blockIdx.y * per, cend = min(d.C, cbeg + per);
  // Both Linears' weights of this wave are requested BEFORE the squeeze rows are folded (they depend on nothing): the
  // launch is a chain of dependent round trips on the forward path — fold, first Linear, second Linear — and this takes
  // the two weight fetches out of it. Shapes of the U-Net (C <= 1024, C / r <= 64, 16 waves); others take the plain loops.
  constexpr int PJ = 4, PC = 16, PU = 8;
  const bool pre = d.C <= 64 * PC && d.Cr <= PJ * 16 && nw == 16 && per <= PU * 16 && d.Cr <= 64;
  float w1v[PJ][PC], w2v[PU];
  if (pre) {
#pragma unroll
    for (int q = 0; q < PJ; ++q) {
      const int j = wave + q * 16;
#pragma unroll
      for (int e = 0; e < PC; ++e) {
        const int c = lane + 64 * e;
        w1v[q][e] = (j < d.Cr && c < d.C) ? d.w1[(int64_t)j * d.C + c] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int c = cbeg + wave + u * 16;
      w2v[u] = (c < cend && lane < d.Cr) ? d.w2[(int64_t)c * d.Cr + lane] : 0.f;
    }
  }
  // fold this image's squeeze rows: part[(n*rows + r)][2][C]
  block_colsum(d.part + (int64_t)n * d.rows * 2 * d.C, d.rows, 2 * d.C, pool, scratch);
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    const float q0 = pool[c];
    const float q1 = pool[d.C + c];
    const float m = (d.scale[c] * q1 + d.shift[c] * q0) * inv_hw;
    sq[c] = m;
    if (writer) {
      d.pooled[((int64_t)n * 2 + 0) * d.C + c] = q0;
      d.pooled[((int64_t)n * 2 + 1) * d.C + c] = q1;
      d.sq[(int64_t)n * d.C + c] = m;
    }
  }
  __syncthreads();
  if (pre) {
    // same products in the same order as the plain loops below (c = lane, lane + 64, ...; j = lane)
#pragma unroll
    for (int q = 0; q < PJ; ++q) {
      const int j = wave + q * 16;
      if (j < d.Cr) {                                   // wave-uniform
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < PC; ++e) {
          const int c = lane + 64 * e;
          if (c < d.C) acc = fmaf(w1v[q][e], sq[c], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) {
          const float hv = fmaxf(acc, 0.f);
          hid[j] = hv;
          if (writer) d.hid[(int64_t)n * d.Cr + j] = hv;
        }
      }
    }
    __syncthreads();
    const float hl = lane < d.Cr ? hid[lane] : 0.f;
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int c = cbeg + wave + u * 16;
      const float t = wave_sum(lane < d.Cr ? fmaf(w2v[u], hl, 0.f) : 0.f);
      if (lane == 0 && c < cend) d.gate[(int64_t)n * d.C + c] = 1.f / (1.f + __expf(-t));
    }
    return;
  }
  for (int j = wave; j < d.Cr; j += nw) {
    float acc = 0.f;
#pragma unroll 8
    for (int c = lane; c < d.C; c += 64) acc = fmaf(d.w1[(int64_t)j * d.C + c], sq[c], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
      const float hv = fmaxf(acc, 0.f);
      hid[j] = hv;
      if (writer) d.hid[(int64_t)n * d.Cr + j] = hv;
    }
  }
  __syncthreads();
  // gate[c] = sigmoid(sum_j w2[c][j] * hid[j]): one wave per row of w2, lanes over j (coalesced)
  // (8 rows per trip so that their loads are in flight together); this work-group's slice of the rows
  for (int c0 = cbeg + wave; c0 < cend; c0 += 8 * nw) {
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + u * nw;
      acc[u] = 0.f;
      if (c < cend)
        for (int j = lane; j < d.Cr; j += 64) acc[u] = fmaf(d.w2[(int64_t)c * d.Cr + j], hid[j], acc[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + u * nw;
      const float t = wave_sum(acc[u]);
      if (lane == 0 && c < cend) d.gate[(int64_t)n * d.C + c] = 1.f / (1.f + __expf(-t));
    }
  }
}

extern "C" int insar_se_excite(const InsarSeFwd* d, void* stream) {
  if (!d || !d->part || !d->pooled || !d->scale || !d->shift || !d->w1 || !d->w2 || !d->sq || !d->hid || !d->gate)
    INSAR_FAIL(INSAR_E_ARG, "insar_se_excite: null pointer");
  if (d->B < 1 || d->C < 1 || d->Cr < 1 || d->C > 4096 || d->rows < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_se_excite: bad shape");
  size_t lds = (size_t)(3 * d->C + d->Cr + COEF_THREADS) * sizeof(float);
  int split = d->C / 128;                          // work-groups per image: 128 rows of the second Linear each
  split = split < 1 ? 1 : (split > 8 ? 8 : split);
  hipLaunchKernelGGL(se_excite_kernel, dim3(d->B, split), dim3(COEF_THREADS), lds, (hipStream_t)stream, *d);
  INSAR_CHECK_LAUNCH("insar_se_excite");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// backward coefficients of [BN(train|eval) -> ReLU -> SE gate]
//   red[n][0][c] = sum_hw dout*mask (P2)      red[n][1][c] = sum_hw dout*mask*y (Q)
// stage 1 (block per image): SE MLP backward for that image, coefB[n][c] = dsq/HW, and the
//   image's contribution to dbeta/dgamma  -> ws.
// stage 2 (thread per channel / per weight): dgamma, dbeta, k1, k2, dW1, dW2, conv-bias grad.
// ws layout (floats): du[B][C] | dt[B][Cr] | tb[B][C] | tg[B][C]
// ---------------------------------------------------------------------------------------------
// Hand-off of stage-1 results to the work-group that runs stage 2 in the SAME launch (bnse_bwd_fused): written through to
// memory and read past the non-coherent caches (agent-scope relaxed atomics = `sc1` stores / loads): no release / acquire
// fence (a release fence writes back every dirty line of the XCD's L2 — the activations the step has just produced).
template <bool COH> __device__ __forceinline__ void hand_st(float* p, float v) {
  if constexpr (COH) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
template <bool COH> __device__ __forceinline__ float hand_ld(const float* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return *p;
}

struct BnSeBwdArgs {
  InsarBnSeBwd d;
  const float* red; int rows; const float* scale; const float* shift;
  float* ws; float* dconv_bias; int training;
};

template <bool COH>
__device__ __forceinline__ void bnse_stage1_body(const BnSeBwdArgs& a, float* sm) {
  const InsarBnSeBwd& d = a.d;
  float* du_s = sm;              // [C]
  float* dt_s = sm + d.C;        // [Cr]
  float* red_s = dt_s + d.Cr + 1;    // [2C]: this image's folded (P2, Q)
  float* scratch = red_s + 2 * d.C;  // [blockDim]
  const int n = blockIdx.x;
  block_colsum(a.red + (int64_t)n * a.rows * 2 * d.C, a.rows, 2 * d.C, red_s, scratch);
  float* du_g = a.ws + (int64_t)n * d.C;
  float* dt_g = a.ws + (int64_t)d.B * d.C + (int64_t)n * d.Cr;
  float* tb_g = a.ws + (int64_t)d.B * (d.C + d.Cr) + (int64_t)n * d.C;
  float* tg_g = tb_g + (int64_t)d.B * d.C;
  const float inv_hw = 1.f / ((float)d.H * (float)d.W);
  if (d.use_se) {
    for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
      const float p2 = red_s[c];
      const float q = red_s[d.C + c];
      const float ds = a.scale[c] * q + a.shift[c] * p2;         // sum dout * z
      const float s = d.gate[(int64_t)n * d.C + c];
      const float du = ds * s * (1.f - s);
      du_s[c] = du; hand_st<COH>(du_g + c, du);
    }
    __syncthreads();
    // dt[j] = relu'(hid[j]) * sum_c du[c] * w2[c][j]: waves stride over the rows c of w2, lanes over j
    // (coalesced), per-wave partials folded through LDS in wave order.
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
        const float dtv = d.hid[(int64_t)n * d.Cr + j] > 0.f ? t : 0.f;
        dt_s[j] = dtv; hand_st<COH>(dt_g + j, dtv);
      }
      __syncthreads();
    }
  }
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    const float p2 = red_s[c];
    const float q = red_s[d.C + c];
    const float mean = d.mean[c], istd = d.invstd[c];
    const float p4 = istd * (q - mean * p2);                     // sum dout*mask*xhat
    float s = 1.f, cb = 0.f, cnt = 0.f, p5 = 0.f;
    if (d.use_se) {
      float dsq = 0.f;
#pragma unroll 8
      for (int j = 0; j < d.Cr; ++j) dsq = fmaf(dt_s[j], d.w1[(int64_t)j * d.C + c], dsq);
      cb = dsq * inv_hw;
      s = d.gate[(int64_t)n * d.C + c];
      cnt = d.pooled[((int64_t)n * 2 + 0) * d.C + c];
      const float sy = d.pooled[((int64_t)n * 2 + 1) * d.C + c];
      p5 = istd * (sy - mean * cnt);
      d.coefB[(int64_t)n * d.C + c] = cb;
    }
    hand_st<COH>(tb_g + c, s * p2 + cb * cnt);
    hand_st<COH>(tg_g + c, s * p4 + cb * p5);
  }
}

__global__ void __launch_bounds__(COEF_THREADS) bnse_bwd_stage1(BnSeBwdArgs a) {
  extern __shared__ float sm[];
  bnse_stage1_body<false>(a, sm);
}

// element `tid` of stage 2 (a channel and / or an SE weight)
// (The three sums over the images are dependent load + fma chains, 48 L2 round trips = 12 us per launch. Requesting the
// sixteen images' operands together brings the launch to 6 us and the STEP from 7.34 to 7.39 ms, same-box, five
// interleaved rounds: on the input-gradient chain a latency-bound launch is a window in which the side stream's weight
// gradient has the chip to itself, and closing it moves that work beside the next MFMA-bound launch of the chain. Kept slow.)
template <bool COH>
__device__ __forceinline__ void bnse_stage2_elem(const BnSeBwdArgs& a, int64_t tid) {
  const InsarBnSeBwd& d = a.d;
  const float* du_g = a.ws;
  const float* dt_g = a.ws + (int64_t)d.B * d.C;
  const float* tb_g = a.ws + (int64_t)d.B * (d.C + d.Cr);
  const float* tg_g = tb_g + (int64_t)d.B * d.C;
  if (tid < d.C) {
    const int c = (int)tid;
    float db = 0.f, dg = 0.f;
    for (int n = 0; n < d.B; ++n) { db += hand_ld<COH>(tb_g + (int64_t)n * d.C + c); dg += hand_ld<COH>(tg_g + (int64_t)n * d.C + c); }
    const float invN = 1.f / ((float)d.B * (float)d.H * (float)d.W);
    d.k1[c] = a.training ? db * invN : 0.f;
    d.k2[c] = a.training ? dg * invN : 0.f;
    if (d.accumulate) { d.dgamma[c] += dg; d.dbeta[c] += db; } else { d.dgamma[c] = dg; d.dbeta[c] = db; }
    if (a.dconv_bias) {
      // training: sum dy == 0 exactly (batch-mean subtraction); eval: sum dy = scale * dbeta
      const float v = a.training ? 0.f : a.scale[c] * db;
      if (d.accumulate) a.dconv_bias[c] += v; else a.dconv_bias[c] = v;
    }
  }
  if (!d.use_se) return;
  const int64_t nw = (int64_t)d.C * d.Cr;
  if (tid < nw) {                       // dW2[c][j] = sum_n du[n][c] * hid[n][j]
    const int c = (int)(tid / d.Cr), j = (int)(tid - (int64_t)c * d.Cr);
    float acc = 0.f;
    for (int n = 0; n < d.B; ++n) acc = fmaf(hand_ld<COH>(du_g + (int64_t)n * d.C + c), d.hid[(int64_t)n * d.Cr + j], acc);
    if (d.accumulate) d.dw2[tid] += acc; else d.dw2[tid] = acc;
    // dW1[j][c] = sum_n dt[n][j] * sq[n][c]   (index tid2 = j*C + c)
    const int j1 = (int)(tid / d.C), c1 = (int)(tid - (int64_t)j1 * d.C);
    float acc1 = 0.f;
    for (int n = 0; n < d.B; ++n) acc1 = fmaf(hand_ld<COH>(dt_g + (int64_t)n * d.Cr + j1), d.sq[(int64_t)n * d.C + c1], acc1);
    if (d.accumulate) d.dw1[tid] += acc1; else d.dw1[tid] = acc1;
  }
}

__global__ void bnse_bwd_stage2(BnSeBwdArgs a) {
  bnse_stage2_elem<false>(a, blockIdx.x * (int64_t)blockDim.x + threadIdx.x);
}

// (Since round 3 the DEFAULT for units without an SE gate is insar_bn_bwd_coef below: one channel-parallel launch, no
// hand-off between work-groups at all. This kernel stays as the INSAR_COEF_SIMPLE=0 variant. Its hand-off relies on
// gfx950 behaviour rather than on the HSA memory model: the stage-1 results are relaxed agent-scope atomic stores, which
// the compiler emits as `global_store ... sc1` — written through to memory, past the XCD's L2 —, every wave drains them
// with s_waitcnt vmcnt(0) before the work-group's barrier, and only then does lane 0 draw the ticket (a relaxed agent-scope
// RMW, executed at the memory side); the last arriver reads with relaxed agent-scope atomic loads (`sc1`: served past its
// own L1 / L2). MI355X_MICROARCH.md, "Valid forms", measures exactly this store / drain / counter / sc1-load sequence;
// what it does not give is a formal release / acquire edge, which is why it is no longer the default.)
// Both stages in ONE launch, for units WITHOUT an SE gate (stage 2 is then C elements of 2 * B loads): every work-group runs
// stage 1 for its image with the hand-off results written through to memory, drains its stores and draws a ticket; the
// work-group that draws the last one runs stage 2, reading the hand-off data past the caches. One launch less on the
// input-gradient chain per such unit. The ticket resets itself. Same arithmetic in the same order: bitwise equal.
__global__ void __launch_bounds__(COEF_THREADS) bnse_bwd_fused(BnSeBwdArgs a, unsigned int* ticket) {
  extern __shared__ float sm[];
  __shared__ int s_last;
  bnse_stage1_body<true>(a, sm);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's write-through stores have reached memory
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int prev = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = prev == gridDim.x - 1;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch starts from zero
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  const InsarBnSeBwd& d = a.d;
  int64_t work = d.C;
  if (d.use_se && (int64_t)d.C * d.Cr > work) work = (int64_t)d.C * d.Cr;
  for (int64_t e = threadIdx.x; e < work; e += blockDim.x) bnse_stage2_elem<true>(a, e);
}

static int launch_bnse_coef(const char* who, const InsarBnSeBwd* d, const float* red, int32_t rows, const float* scale,
                            const float* shift, float* ws, float* dconv_bias, int32_t training, int stages, void* stream) {
  if (!d || !red || !scale || !shift || !ws || !d->mean || !d->invstd || !d->dgamma || !d->dbeta || !d->k1 || !d->k2)
    INSAR_FAIL(INSAR_E_ARG, "%s: null pointer", who);
  if (d->use_se && (!d->pooled || !d->sq || !d->hid || !d->gate || !d->w1 || !d->w2 || !d->dw1 || !d->dw2 || !d->coefB))
    INSAR_FAIL(INSAR_E_ARG, "%s: SE pointers missing", who);
  if (d->C > 8192 || d->B < 1) INSAR_FAIL(INSAR_E_SHAPE, "%s: bad shape", who);
  if (rows < 1) INSAR_FAIL(INSAR_E_SHAPE, "%s: rows", who);
  BnSeBwdArgs a; a.d = *d; a.red = red; a.rows = rows; a.scale = scale; a.shift = shift; a.ws = ws; a.dconv_bias = dconv_bias; a.training = training;
  hipStream_t s = (hipStream_t)stream;
  if (stages & 1) {
    size_t lds = (size_t)(3 * d->C + d->Cr + 1 + COEF_THREADS) * sizeof(float);
    if (lds > 64 * 1024) {
      static std::atomic<uint64_t> attr_mask{0};
      hipError_t e = insar_set_lds_once(attr_mask, (const void*)bnse_bwd_stage1, 160 * 1024);
      if (e != hipSuccess) INSAR_FAIL(-(int)e, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(bnse_bwd_stage1, dim3(d->B), dim3(COEF_THREADS), lds, s, a);
  }
  if (stages & 2) {
    int64_t work = d->C;
    if (d->use_se && (int64_t)d->C * d->Cr > work) work = (int64_t)d->C * d->Cr;
    hipLaunchKernelGGL(bnse_bwd_stage2, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, a);
  }
  INSAR_CHECK_LAUNCH(who);
  return INSAR_OK;
}

extern "C" int insar_bnse_bwd_coef(const InsarBnSeBwd* d, const float* red, int32_t rows, const float* scale,
                                   const float* shift, float* ws, float* dconv_bias, int32_t training, void* stream) {
  return launch_bnse_coef("insar_bnse_bwd_coef", d, red, rows, scale, shift, ws, dconv_bias, training, 3, stream);
}

// The two stages in one launch (see bnse_bwd_fused). ticket: one zero-initialised 32-bit word per call site, reset by the
// kernel itself; calls that share a ticket must be ordered on one stream.
extern "C" int insar_bnse_bwd_coef_fused(const InsarBnSeBwd* d, const float* red, int32_t rows, const float* scale,
                                         const float* shift, float* ws, float* dconv_bias, int32_t training,
                                         uint32_t* ticket, void* stream) {
  if (!ticket) INSAR_FAIL(INSAR_E_ARG, "insar_bnse_bwd_coef_fused: null ticket");
  if (!d || !red || !scale || !shift || !ws || !d->mean || !d->invstd || !d->dgamma || !d->dbeta || !d->k1 || !d->k2)
    INSAR_FAIL(INSAR_E_ARG, "insar_bnse_bwd_coef_fused: null pointer");
  if (d->use_se && (!d->pooled || !d->sq || !d->hid || !d->gate || !d->w1 || !d->w2 || !d->dw1 || !d->dw2 || !d->coefB))
    INSAR_FAIL(INSAR_E_ARG, "insar_bnse_bwd_coef_fused: SE pointers missing");
  if (d->C > 8192 || d->B < 1 || rows < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_bnse_bwd_coef_fused: bad shape");
  BnSeBwdArgs a; a.d = *d; a.red = red; a.rows = rows; a.scale = scale; a.shift = shift; a.ws = ws; a.dconv_bias = dconv_bias; a.training = training;
  size_t lds = (size_t)(3 * d->C + d->Cr + 1 + COEF_THREADS) * sizeof(float);
  if (lds > 64 * 1024) {           // C above ~5000: more dynamic LDS than a kernel gets without the attribute (per device)
    static std::atomic<uint64_t> attr_mask{0};
    hipError_t e = insar_set_lds_once(attr_mask, (const void*)bnse_bwd_fused, 160 * 1024);
    if (e != hipSuccess) INSAR_FAIL(-(int)e, "insar_bnse_bwd_coef_fused: hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(bnse_bwd_fused, dim3(d->B), dim3(COEF_THREADS), lds, (hipStream_t)stream, a, (unsigned int*)ticket);
  INSAR_CHECK_LAUNCH("insar_bnse_bwd_coef_fused");
  return INSAR_OK;
}

// Units WITHOUT an SE gate: the coefficients are linear in the slab rows, so the per-image stage is not needed — ONE launch
// whose work-groups own 64 channels each and fold ALL rows (B * rows of them, any partition of the pixels):
//   P = sum_rows red[.][0][c], Q = sum_rows red[.][1][c];  dbeta = P;  dgamma = invstd * (Q - mean * P);
//   k1 = dbeta / N, k2 = dgamma / N (training; 0 in eval mode);  conv-bias gradient 0 (training) / scale * dbeta (eval).
// 1024 threads = 64 channels x 16 row groups (coalesced 256-byte row segments), eight rows of a group in flight, folded
// through LDS in a fixed order: deterministic, no hand-off between work-groups (the two-stage kernels above run 16
// work-groups and then one). The launch is pure latency on the input-gradient chain (C / 64 work-groups reading a slab of up
// to 1024 rows): with 4 row groups and 4 rows in flight it took rows / 16 dependent L2 round trips, now rows / 128.
#define BNC_GROUPS 16
__global__ void __launch_bounds__(64 * BNC_GROUPS) bn_bwd_coef_kernel(const float* __restrict__ red, int64_t rows, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ scale, float inv_count, int training,
                                                          int accumulate, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ k1, float* __restrict__ k2, float* __restrict__ dconv_bias) {
  __shared__ float sp[BNC_GROUPS][64], sq[BNC_GROUPS][64];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float p = 0.f, q = 0.f;
  if (c < C) {
    int64_t r = g;
    const int64_t st = (int64_t)BNC_GROUPS * 2 * C;            // floats between two rows of this group
    for (; r + 7 * BNC_GROUPS < rows; r += 8 * BNC_GROUPS) {
      const float* b = red + r * 2 * C + c;
      float pv[8], qv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { pv[u] = b[u * st]; qv[u] = b[u * st + C]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { p += pv[u]; q += qv[u]; }
    }
    for (; r < rows; r += BNC_GROUPS) { p += red[r * 2 * C + c]; q += red[(r * 2 + 1) * C + c]; }
  }
  sp[g][cl] = p; sq[g][cl] = q;
  __syncthreads();
  if (g == 0 && c < C) {
    float P = 0.f, Q = 0.f;
#pragma unroll
    for (int k = 0; k < BNC_GROUPS; ++k) { P += sp[k][cl]; Q += sq[k][cl]; }
    const float dg = invstd[c] * (Q - mean[c] * P);
    k1[c] = training ? P * inv_count : 0.f;
    k2[c] = training ? dg * inv_count : 0.f;
    if (accumulate) { dgamma[c] += dg; dbeta[c] += P; } else { dgamma[c] = dg; dbeta[c] = P; }
    if (dconv_bias) {
      const float v = training ? 0.f : scale[c] * P;
      if (accumulate) dconv_bias[c] += v; else dconv_bias[c] = v;
    }
  }
}

extern "C" int insar_bn_bwd_coef(const InsarBnSeBwd* d, const float* red, int64_t rows_total, const float* scale,
                                 float* dconv_bias, int32_t training, void* stream) {
  if (!d || !red || !scale || !d->mean || !d->invstd || !d->dgamma || !d->dbeta || !d->k1 || !d->k2)
    INSAR_FAIL(INSAR_E_ARG, "insar_bn_bwd_coef: null pointer");
  if (d->use_se) INSAR_FAIL(INSAR_E_ARG, "insar_bn_bwd_coef: units with an SE gate need insar_bnse_bwd_coef (per-image stage)");
  if (d->C < 1 || d->B < 1 || rows_total < 1) INSAR_FAIL(INSAR_E_SHAPE, "insar_bn_bwd_coef: bad shape");
  const float inv_count = 1.f / ((float)d->B * (float)d->H * (float)d->W);
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((d->C + 63) / 64), dim3(64 * BNC_GROUPS), 0, (hipStream_t)stream, red, rows_total, d->C, d->mean,
                     d->invstd, scale, inv_count, training, d->accumulate, d->dgamma, d->dbeta, d->k1, d->k2, dconv_bias);
  INSAR_CHECK_LAUNCH("insar_bn_bwd_coef");
  return INSAR_OK;
}

// One stage of the above: 1 = per-image stage (SE backward, coefB, the per-image partial sums in ws), 2 = batch fold
// (k1, k2, dgamma, dbeta, dW1, dW2, conv-bias gradient). The dgrad chain needs only stage 1 when the apply pass
// folds k1 / k2 itself (insar_bnrelu_bwd_apply_part); stage 2 can then run off the critical path.
// ws floats: du[B][C] | dt[B][Cr] | tb[B][C] | tg[B][C].
extern "C" int insar_bnse_bwd_coef_stage(const InsarBnSeBwd* d, const float* red, int32_t rows, const float* scale,
                                         const float* shift, float* ws, float* dconv_bias, int32_t training,
                                         int32_t stage, void* stream) {
  if (stage != 1 && stage != 2) INSAR_FAIL(INSAR_E_ARG, "insar_bnse_bwd_coef_stage: stage must be 1 or 2");
  return launch_bnse_coef("insar_bnse_bwd_coef_stage", d, red, rows, scale, shift, ws, dconv_bias, training, stage, stream);
}

// dy = scale * ( (dout*gate + coefB) * mask - k1 - xhat*k2 ),  xhat = (y - mean)*invstd, evaluated with the per-channel
// constants folded:  dy = (mask ? dout*p + q : 0) + (a0 + a1*y),  p = scale*gate[n], q = scale*coefB[n],
// a1 = -scale*invstd*k2, a0 = -scale*k1 - a1*mean  (six constants per channel instead of eight, five VALU operations per
// element instead of nine). The plain instantiation is bounded by the 256 threads it is launched with, not by 1024: at
// 128 registers it spilled eight of them and reloaded them (with a full vmcnt wait) at every image row; 135 registers,
// three waves per SIMD, no spill: 88 -> 76 us at the 256^2 level, the step 7.38 -> 7.31 ms (profiles/r03_pass_bench.txt).
#define BWD_APPLY_MAXC 1024
template <typename T, int VK = 0, bool VP = false>       // VK, VP as in row_reduce_kernel
__global__ void __launch_bounds__((VK || VP) ? 512 : 256) bnrelu_bwd_apply_kernel(ActView g, ActView y, const float* __restrict__ scale,
                                        const float* __restrict__ shift, const float* __restrict__ mean,
                                        const float* __restrict__ invstd, const float* __restrict__ gate,
                                        const float* __restrict__ coefB, const float* k1,
                                        const float* k2, ActView dy, int relu,
                                        const float* __restrict__ tb, const float* __restrict__ tg, OutcGrad og, PoolGrad pg) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = y.c_len / CH;
  const int rows = y.B * y.H;
  const int total = y.W * cpp;
  const bool inv = (blockDim.x % cpp) == 0;
  // k1 / k2 from the per-image partials of bnse_bwd_stage1 (tb, tg: [nimg][C]), folded here in image order
  // exactly as bnse_bwd_stage2 folds them: the batch-fold launch leaves the dgrad chain (it still produces the
  // parameter gradients, on the side stream).
  __shared__ float sk[2 * BWD_APPLY_MAXC];
  if (tb) {
    const float invN = 1.f / ((float)y.B * (float)y.H * (float)y.W);      // as in bnse_bwd_stage2
    for (int c = threadIdx.x; c < y.c_len; c += blockDim.x) {
      float db = 0.f, dg = 0.f;
      for (int n = 0; n < y.B; ++n) { db += tb[(int64_t)n * y.c_len + c]; dg += tg[(int64_t)n * y.c_len + c]; }
      sk[c] = db * invN; sk[BWD_APPLY_MAXC + c] = dg * invN;
    }
    __syncthreads();
    k1 = sk; k2 = sk + BWD_APPLY_MAXC;
  }
  float sc[CH], sh[CH], fp[CH], fq[CH], a0[CH], a1[CH];
  auto load_channel_consts = [&](int c0) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int c = c0 + j;
      sc[j] = scale[c]; sh[j] = shift[c];
      a1[j] = -sc[j] * invstd[c] * k2[c];
      a0[j] = -sc[j] * k1[c] - a1[j] * mean[c];
      fp[j] = sc[j]; fq[j] = 0.f;
    }
  };
  // the image's gate / coefB chunk: 16-byte loads issued together behind ONE test of each pointer (written per element,
  // `if (gate) fp[j] = ...` became sixteen dependent load + full-wait pairs at the head of every image row)
  auto load_image_consts = [&](int n, int c0) {
    float4 gv[CH / 4], bv[CH / 4];
    const int64_t off = (int64_t)n * y.c_len + c0;
    if (gate) {
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) gv[q] = *(const float4*)(gate + off + 4 * q);
    }
    if (coefB) {
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) bv[q] = *(const float4*)(coefB + off + 4 * q);
    }
    if (gate) {
#pragma unroll
      for (int j = 0; j < CH; ++j) fp[j] = sc[j] * ((const float*)gv)[j];
    }
    if (coefB) {
#pragma unroll
      for (int j = 0; j < CH; ++j) fq[j] = sc[j] * ((const float*)bv)[j];
    }
  };
  if (inv) {
    const int cc = threadIdx.x % cpp, wstep = blockDim.x / cpp;
    load_channel_consts(cc * CH);
    constexpr bool virt = VK > 0;                     // dout recomputed from dlogits (see OutcGrad): own instantiation
    constexpr int KM = virt ? VK : 1;
    float wk[KM][CH];
    if constexpr (virt) {
#pragma unroll
      for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int j = 0; j < CH; ++j) wk[k][j] = k < og.K ? og.w[k * y.c_len + cc * CH + j] : 0.f;
    }
    const int64_t HW = (int64_t)y.H * y.W;
    int n_loaded = -1;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
      const int n = r / y.H, h = r - n * y.H;
      if (n != n_loaded) {
        load_image_consts(n, cc * CH);
        n_loaded = n;
      }
      for (int w0 = threadIdx.x / cpp; w0 < y.W; w0 += PW_UNROLL * wstep) {
        uint4 vy[PW_UNROLL], vg[PW_UNROLL], vp[VP ? PW_UNROLL : 1];
        float dlv[PW_UNROLL][KM];
#pragma unroll
        for (int u = 0; u < PW_UNROLL; ++u)
          if (w0 + u * wstep < y.W) {
            vy[u] = *chunk_ptr<T>(y, n, h, w0 + u * wstep, cc);
            if constexpr (virt) {
#pragma unroll
              for (int k = 0; k < KM; ++k)
                if (k < og.K) dlv[u][k] = og.dl[((int64_t)n * og.K + k) * HW + (int64_t)h * y.W + w0 + u * wstep];
            } else {
              vg[u] = *chunk_ptr<T>(g, n, h, w0 + u * wstep, cc);
              if constexpr (VP) vp[u] = *chunk_ptr<T>(pg.dp, n, h >> 1, (w0 + u * wstep) >> 1, cc);
            }
          }
#pragma unroll
        for (int u = 0; u < PW_UNROLL; ++u)
          if (w0 + u * wstep < y.W) {
            float f[CH], gg[CH], o[CH];
            Chunk<T>::unpack(vy[u], f);
            if constexpr (virt) outc_grad_chunk<T, KM>(dlv[u], wk, og.K, gg);
            else Chunk<T>::unpack(vg[u], gg);
            if constexpr (VP) {
              const int w = w0 + u * wstep;
              pool_grad_chunk<T>(vp[u], pg.arg + ((((int64_t)n * (y.H >> 1) + (h >> 1)) * (y.W >> 1) + (w >> 1)) * y.c_len + cc * CH),
                                 ((h & 1) << 1) | (w & 1), gg);
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
              const bool on = !relu || fmaf(f[j], sc[j], sh[j]) > 0.f;
              const float ge = on ? fmaf(gg[j], fp[j], fq[j]) : 0.f;
              o[j] = ge + fmaf(f[j], a1[j], a0[j]);
            }
            *chunk_ptr_w<T>(dy, n, h, w0 + u * wstep, cc) = Chunk<T>::pack(o);
          }
      }
    }
    return;
  }
  // generic (slow) path: any channel count
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / y.H, h = r - n * y.H;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int w = e / cpp, cc = e - w * cpp;
      load_channel_consts(cc * CH);
      load_image_consts(n, cc * CH);
      float f[CH], gg[CH], o[CH];
      Chunk<T>::unpack(*chunk_ptr<T>(y, n, h, w, cc), f);
      Chunk<T>::unpack(*chunk_ptr<T>(g, n, h, w, cc), gg);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const bool on = !relu || fmaf(f[j], sc[j], sh[j]) > 0.f;
        const float ge = on ? fmaf(gg[j], fp[j], fq[j]) : 0.f;
        o[j] = ge + fmaf(f[j], a1[j], a0[j]);
      }
      *chunk_ptr_w<T>(dy, n, h, w, cc) = Chunk<T>::pack(o);
    }
  }
}

static int launch_bwd_apply(const char* who, const InsarAct* dout, const InsarAct* y, const float* scale, const float* shift,
                            const float* mean, const float* invstd, const float* gate, const float* coefB,
                            const float* k1, const float* k2, const InsarAct* dy, int32_t relu,
                            const float* tb, const float* tg, OutcGrad og, PoolGrad pg, void* stream) {
  int rc;
  if ((rc = insar_check_act(y, who, "y"))) return rc;
  if (!og.dl) {
    if ((rc = insar_check_act(dout, who, "dout"))) return rc;
    if ((rc = check_same_grid(y, dout, who))) return rc;
  }
  if ((rc = insar_check_act(dy, who, "dy"))) return rc;
  if ((rc = check_same_grid(y, dy, who))) return rc;
  if (!scale || !shift || !mean || !invstd) INSAR_FAIL(INSAR_E_ARG, "%s: null pointer", who);
  if (!tb && (!k1 || !k2)) INSAR_FAIL(INSAR_E_ARG, "%s: null k1/k2", who);
  if ((gate && !insar_aligned16(gate)) || (coefB && !insar_aligned16(coefB))) INSAR_FAIL(INSAR_E_ALIGN, "%s: gate / coefB not 16-byte aligned", who);
  if (tb && (!tg || y->c_len > BWD_APPLY_MAXC))
    INSAR_FAIL(INSAR_E_SHAPE, "%s: partial sums need tg and C <= %d", who, BWD_APPLY_MAXC);
  int grid = insar_grid_cap((int64_t)y->B * y->H);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH_APPLY_VK(TT, VKK) hipLaunchKernelGGL((bnrelu_bwd_apply_kernel<TT, VKK>), dim3(grid), dim3(PW_THREADS), 0, s, make_view(*y), make_view(*y), scale, shift, mean, invstd, gate, coefB, k1, k2, make_view(*dy), relu, tb, tg, og, pg)
  if (og.dl) {
    if (y->dtype == INSAR_BF16) { if (og.K <= 2) LAUNCH_APPLY_VK(bf16_t, 2); else LAUNCH_APPLY_VK(bf16_t, OG_MAXK); }
    else { if (og.K <= 2) LAUNCH_APPLY_VK(float, 2); else LAUNCH_APPLY_VK(float, OG_MAXK); }
#undef LAUNCH_APPLY_VK
  } else if (pg.arg) {
    if (y->dtype == INSAR_BF16)
      hipLaunchKernelGGL((bnrelu_bwd_apply_kernel<bf16_t, 0, true>), dim3(grid), dim3(PW_THREADS), 0, s, make_view(*dout), make_view(*y), scale, shift, mean, invstd, gate, coefB, k1, k2, make_view(*dy), relu, tb, tg, og, pg);
    else
      hipLaunchKernelGGL((bnrelu_bwd_apply_kernel<float, 0, true>), dim3(grid), dim3(PW_THREADS), 0, s, make_view(*dout), make_view(*y), scale, shift, mean, invstd, gate, coefB, k1, k2, make_view(*dy), relu, tb, tg, og, pg);
  } else if (y->dtype == INSAR_BF16)
    hipLaunchKernelGGL(bnrelu_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*dout), make_view(*y), scale, shift, mean, invstd, gate, coefB, k1, k2, make_view(*dy), relu, tb, tg, og, pg);
  else
    hipLaunchKernelGGL(bnrelu_bwd_apply_kernel<float>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*dout), make_view(*y), scale, shift, mean, invstd, gate, coefB, k1, k2, make_view(*dy), relu, tb, tg, og, pg);
  INSAR_CHECK_LAUNCH(who);
  return INSAR_OK;
}

extern "C" int insar_bnrelu_bwd_apply(const InsarAct* dout, const InsarAct* y, const float* scale, const float* shift,
                                      const float* mean, const float* invstd, const float* gate, const float* coefB,
                                      const float* k1, const float* k2, const InsarAct* dy, int32_t relu, void* stream) {
  return launch_bwd_apply("insar_bnrelu_bwd_apply", dout, y, scale, shift, mean, invstd, gate, coefB, k1, k2, dy, relu,
                          nullptr, nullptr, OutcGrad{nullptr, nullptr, 0, nullptr, nullptr}, PoolGrad{ActView{}, nullptr}, stream);
}

// The same pass with dout = round(dskip + (arg == position ? dpooled : 0)) (see insar_bnrelu_bwd_reduce_pool).
extern "C" int insar_bnrelu_bwd_apply_pool(const InsarAct* dskip, const InsarAct* dpooled, const uint8_t* arg, const InsarAct* y,
                                           const float* scale, const float* shift, const float* mean, const float* invstd,
                                           const float* gate, const float* coefB, const float* k1, const float* k2,
                                           const InsarAct* dy, int32_t relu, void* stream) {
  int rc;
  if ((rc = insar_check_act(y, "insar_bnrelu_bwd_apply_pool", "y"))) return rc;
  if ((rc = check_pool_grad(y, dpooled, arg, "insar_bnrelu_bwd_apply_pool"))) return rc;
  return launch_bwd_apply("insar_bnrelu_bwd_apply_pool", dskip, y, scale, shift, mean, invstd, gate, coefB, k1, k2, dy, relu,
                          nullptr, nullptr, OutcGrad{nullptr, nullptr, 0, nullptr, nullptr}, PoolGrad{make_view(*dpooled), arg}, stream);
}

// The same pass with dout = gradient of the 1x1 output conv's input recomputed from dlogits (see OutcGrad above).
extern "C" int insar_bnrelu_bwd_apply_outc(const float* dlogits, const float* wout, int32_t K, const InsarAct* y,
                                           const float* scale, const float* shift, const float* mean, const float* invstd,
                                           const float* gate, const float* coefB, const float* k1, const float* k2,
                                           const InsarAct* dy, int32_t relu, void* stream) {
  int rc;
  if ((rc = insar_check_act(y, "insar_bnrelu_bwd_apply_outc", "y"))) return rc;
  if ((rc = check_outc_grad(y, dlogits, wout, K, "insar_bnrelu_bwd_apply_outc"))) return rc;
  return launch_bwd_apply("insar_bnrelu_bwd_apply_outc", nullptr, y, scale, shift, mean, invstd, gate, coefB, k1, k2, dy, relu,
                          nullptr, nullptr, OutcGrad{dlogits, wout, K, nullptr, nullptr}, PoolGrad{ActView{}, nullptr}, stream);
}

// The same pass with k1 / k2 taken from stage 1's per-image partial sums (ws of insar_bnse_bwd_coef_stage, stage 1):
// k1[c] = sum_n tb[n][c] / (B*H*W), k2[c] = sum_n tg[n][c] / (B*H*W) (training-mode BatchNorm), tb/tg: [B][C].
extern "C" int insar_bnrelu_bwd_apply_part(const InsarAct* dout, const InsarAct* y, const float* scale, const float* shift,
                                           const float* mean, const float* invstd, const float* gate, const float* coefB,
                                           const float* tb, const float* tg, const InsarAct* dy, int32_t relu, void* stream) {
  if (!tb) INSAR_FAIL(INSAR_E_ARG, "insar_bnrelu_bwd_apply_part: null partial sums");
  return launch_bwd_apply("insar_bnrelu_bwd_apply_part", dout, y, scale, shift, mean, invstd, gate, coefB, nullptr, nullptr, dy,
                          relu, tb, tg, OutcGrad{nullptr, nullptr, 0, nullptr, nullptr}, PoolGrad{ActView{}, nullptr}, stream);
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(2) (Unet-ChannalAttention.py:106-109). Backward routes to the first maximum in
// scan order (0,0),(0,1),(1,0),(1,1) with torch's "val > max || isnan(val)" update rule.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool2_fwd_kernel(ActView x, ActView y) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = y.c_len / CH;
  const int rows = y.B * y.H;
  const int total = y.W * cpp;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / y.H, h = r - n * y.H;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int w = e / cpp, cc = e - w * cpp;
      float a[CH], b[CH], c[CH], d[CH], o[CH];
      Chunk<T>::unpack(*chunk_ptr<T>(x, n, 2 * h, 2 * w, cc), a);
      Chunk<T>::unpack(*chunk_ptr<T>(x, n, 2 * h, 2 * w + 1, cc), b);
      Chunk<T>::unpack(*chunk_ptr<T>(x, n, 2 * h + 1, 2 * w, cc), c);
      Chunk<T>::unpack(*chunk_ptr<T>(x, n, 2 * h + 1, 2 * w + 1, cc), d);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        float m = a[j];
        if (b[j] > m || b[j] != b[j]) m = b[j];
        if (c[j] > m || c[j] != c[j]) m = c[j];
        if (d[j] > m || d[j] != d[j]) m = d[j];
        o[j] = m;
      }
      *chunk_ptr_w<T>(y, n, h, w, cc) = Chunk<T>::pack(o);
    }
  }
}

template <typename T>
__global__ void maxpool2_bwd_kernel(ActView x, ActView dy, ActView dx, int accumulate) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = dy.c_len / CH;
  const int rows = dy.B * dy.H;
  const int total = dy.W * cpp;
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int n = r / dy.H, h = r - n * dy.H;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int w = e / cpp, cc = e - w * cpp;
      float v[4][CH], g[CH], o[4][CH];
      Chunk<T>::unpack(*chunk_ptr<T>(x, n, 2 * h, 2 * w, cc), v[0]);
      Chunk<T>::unpack(*chunk_ptr<T>(x, n, 2 * h, 2 * w + 1, cc), v[1]);
      Chunk<T>::unpack(*chunk_ptr<T>(x, n, 2 * h + 1, 2 * w, cc), v[2]);
      Chunk<T>::unpack(*chunk_ptr<T>(x, n, 2 * h + 1, 2 * w + 1, cc), v[3]);
      Chunk<T>::unpack(*chunk_ptr<T>(dy, n, h, w, cc), g);
      if (accumulate) {
        Chunk<T>::unpack(*chunk_ptr<T>(dx, n, 2 * h, 2 * w, cc), o[0]);
        Chunk<T>::unpack(*chunk_ptr<T>(dx, n, 2 * h, 2 * w + 1, cc), o[1]);
        Chunk<T>::unpack(*chunk_ptr<T>(dx, n, 2 * h + 1, 2 * w, cc), o[2]);
        Chunk<T>::unpack(*chunk_ptr<T>(dx, n, 2 * h + 1, 2 * w + 1, cc), o[3]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int j = 0; j < CH; ++j) o[q][j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        int best = 0; float m = v[0][j];
#pragma unroll
        for (int q = 1; q < 4; ++q)
          if (v[q][j] > m || v[q][j] != v[q][j]) { m = v[q][j]; best = q; }
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q][j] += (q == best) ? g[j] : 0.f;
      }
      *chunk_ptr_w<T>(dx, n, 2 * h, 2 * w, cc) = Chunk<T>::pack(o[0]);
      *chunk_ptr_w<T>(dx, n, 2 * h, 2 * w + 1, cc) = Chunk<T>::pack(o[1]);
      *chunk_ptr_w<T>(dx, n, 2 * h + 1, 2 * w, cc) = Chunk<T>::pack(o[2]);
      *chunk_ptr_w<T>(dx, n, 2 * h + 1, 2 * w + 1, cc) = Chunk<T>::pack(o[3]);
    }
  }
}

// nn.MaxPool2d(2) floors: an odd last row / column of the input belongs to no window
static int check_pool(const InsarAct* x, const InsarAct* y, const char* who) {
  if (x->B != y->B || y->H != x->H / 2 || y->W != x->W / 2 || y->H < 1 || y->W < 1 || x->c_len != y->c_len || x->dtype != y->dtype)
    INSAR_FAIL(INSAR_E_SHAPE, "%s: pooled grid must be (H/2, W/2) of the input grid, rounded down", who);
  return INSAR_OK;
}

extern "C" int insar_maxpool2_fwd(const InsarAct* x, const InsarAct* y, void* stream) {
  int rc;
  if ((rc = insar_check_act(x, "insar_maxpool2_fwd", "x"))) return rc;
  if ((rc = insar_check_act(y, "insar_maxpool2_fwd", "y"))) return rc;
  if ((rc = check_pool(x, y, "insar_maxpool2_fwd"))) return rc;
  int grid = insar_grid_cap((int64_t)y->B * y->H);
  hipStream_t s = (hipStream_t)stream;
  if (x->dtype == INSAR_BF16) hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16_t>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*x), make_view(*y));
  else hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*x), make_view(*y));
  INSAR_CHECK_LAUNCH("insar_maxpool2_fwd");
  return INSAR_OK;
}

extern "C" int insar_maxpool2_bwd(const InsarAct* x, const InsarAct* dy, const InsarAct* dx, int32_t accumulate, void* stream) {
  int rc;
  if ((rc = insar_check_act(x, "insar_maxpool2_bwd", "x"))) return rc;
  if ((rc = insar_check_act(dy, "insar_maxpool2_bwd", "dy"))) return rc;
  if ((rc = insar_check_act(dx, "insar_maxpool2_bwd", "dx"))) return rc;
  if ((rc = check_pool(x, dy, "insar_maxpool2_bwd"))) return rc;
  if ((rc = check_same_grid(x, dx, "insar_maxpool2_bwd"))) return rc;
  if (!accumulate && ((x->H | x->W) & 1))
    INSAR_FAIL(INSAR_E_SHAPE, "insar_maxpool2_bwd: an odd grid's last row / column belongs to no window: accumulate into a defined dx");
  int grid = insar_grid_cap((int64_t)dy->B * dy->H);
  hipStream_t s = (hipStream_t)stream;
  if (x->dtype == INSAR_BF16) hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16_t>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*x), make_view(*dy), make_view(*dx), accumulate);
  else hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*x), make_view(*dy), make_view(*dx), accumulate);
  INSAR_CHECK_LAUNCH("insar_maxpool2_bwd");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// bilinear resize of an NHWC slice, align_corners = False: the reference's fallback for tile sizes that are not
// multiples of 16 (Unet-ChannalAttention.py:138-139,144-145,150-151,156-157: F_T.resize(x, size=skip, BILINEAR) of the
// transposed-conv output, one pixel larger than 2 x the pooled size), and its adjoint (gather form, deterministic).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void rs_coef(int o, float scale, int in, int& i0, int& i1, float& l1) {
  float src = ((float)o + 0.5f) * scale - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + 1 < in ? i0 + 1 : in - 1;
  l1 = src - (float)i0;
}

template <typename T>
__global__ void resize_fwd_kernel(ActView src, ActView dst, float sh, float sw) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = dst.c_len / CH;
  const int total = dst.W * cpp;
  for (int r = blockIdx.x; r < dst.B * dst.H; r += gridDim.x) {
    const int n = r / dst.H, h = r - n * dst.H;
    int h0, h1; float lh;
    rs_coef(h, sh, src.H, h0, h1, lh);
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int w = e / cpp, cc = e - w * cpp;
      int w0, w1; float lw;
      rs_coef(w, sw, src.W, w0, w1, lw);
      float a[CH], b[CH], c[CH], d[CH], o[CH];
      Chunk<T>::unpack(*chunk_ptr<T>(src, n, h0, w0, cc), a);
      Chunk<T>::unpack(*chunk_ptr<T>(src, n, h0, w1, cc), b);
      Chunk<T>::unpack(*chunk_ptr<T>(src, n, h1, w0, cc), c);
      Chunk<T>::unpack(*chunk_ptr<T>(src, n, h1, w1, cc), d);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const float top = (1.f - lw) * a[j] + lw * b[j], bot = (1.f - lw) * c[j] + lw * d[j];
        o[j] = (1.f - lh) * top + lh * bot;
      }
      *chunk_ptr_w<T>(dst, n, h, w, cc) = Chunk<T>::pack(o);
    }
  }
}

template <typename T>
__global__ void resize_bwd_kernel(ActView ddst, ActView dsrc, float sh, float sw) {
  constexpr int CH = Chunk<T>::N;
  const int cpp = dsrc.c_len / CH;
  const int total = dsrc.W * cpp;
  for (int r = blockIdx.x; r < dsrc.B * dsrc.H; r += gridDim.x) {
    const int n = r / dsrc.H, hi = r - n * dsrc.H;
    int ho_lo = (int)floorf(((float)hi - 0.5f) / sh - 0.5f) - 1, ho_hi = (int)ceilf(((float)hi + 1.5f) / sh - 0.5f) + 1;
    if (hi == 0) ho_lo = 0;
    if (hi == dsrc.H - 1) ho_hi = ddst.H - 1;
    ho_lo = ho_lo < 0 ? 0 : ho_lo; ho_hi = ho_hi > ddst.H - 1 ? ddst.H - 1 : ho_hi;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int wi = e / cpp, cc = e - wi * cpp;
      int wo_lo = (int)floorf(((float)wi - 0.5f) / sw - 0.5f) - 1, wo_hi = (int)ceilf(((float)wi + 1.5f) / sw - 0.5f) + 1;
      if (wi == 0) wo_lo = 0;
      if (wi == dsrc.W - 1) wo_hi = ddst.W - 1;
      wo_lo = wo_lo < 0 ? 0 : wo_lo; wo_hi = wo_hi > ddst.W - 1 ? ddst.W - 1 : wo_hi;
      float acc[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) acc[j] = 0.f;
      for (int ho = ho_lo; ho <= ho_hi; ++ho) {
        int h0, h1; float lh;
        rs_coef(ho, sh, dsrc.H, h0, h1, lh);
        float wh = 0.f;
        if (h0 == hi) wh += 1.f - lh;
        if (h1 == hi) wh += lh;
        if (wh == 0.f) continue;
        for (int wo = wo_lo; wo <= wo_hi; ++wo) {
          int w0, w1; float lw;
          rs_coef(wo, sw, dsrc.W, w0, w1, lw);
          float ww = 0.f;
          if (w0 == wi) ww += 1.f - lw;
          if (w1 == wi) ww += lw;
          if (ww == 0.f) continue;
          float g[CH];
          Chunk<T>::unpack(*chunk_ptr<T>(ddst, n, ho, wo, cc), g);
          const float k = wh * ww;
#pragma unroll
          for (int j = 0; j < CH; ++j) acc[j] = fmaf(k, g[j], acc[j]);
        }
      }
      *chunk_ptr_w<T>(dsrc, n, hi, wi, cc) = Chunk<T>::pack(acc);
    }
  }
}

static int check_resize(const InsarAct* a, const InsarAct* b, const char* who) {
  if (a->B != b->B || a->c_len != b->c_len || a->dtype != b->dtype) INSAR_FAIL(INSAR_E_SHAPE, "%s: batch / channels / dtype differ", who);
  const int ch = a->dtype == INSAR_BF16 ? 8 : 4;
  if (a->c_len % ch) INSAR_FAIL(INSAR_E_SHAPE, "%s: C=%d unsupported", who, a->c_len);
  return INSAR_OK;
}

extern "C" int insar_resize_bilinear_fwd(const InsarAct* src, const InsarAct* dst, void* stream) {
  int rc;
  if ((rc = insar_check_act(src, "insar_resize_bilinear_fwd", "src"))) return rc;
  if ((rc = insar_check_act(dst, "insar_resize_bilinear_fwd", "dst"))) return rc;
  if ((rc = check_resize(src, dst, "insar_resize_bilinear_fwd"))) return rc;
  int grid = insar_grid_cap((int64_t)dst->B * dst->H);
  hipStream_t s = (hipStream_t)stream;
  const float sh = (float)src->H / (float)dst->H, sw = (float)src->W / (float)dst->W;
  if (src->dtype == INSAR_BF16) hipLaunchKernelGGL(resize_fwd_kernel<bf16_t>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*src), make_view(*dst), sh, sw);
  else hipLaunchKernelGGL(resize_fwd_kernel<float>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*src), make_view(*dst), sh, sw);
  INSAR_CHECK_LAUNCH("insar_resize_bilinear_fwd");
  return INSAR_OK;
}

/* dsrc = adjoint of insar_resize_bilinear_fwd applied to ddst (the gradient of the resized tensor) */
extern "C" int insar_resize_bilinear_bwd(const InsarAct* ddst, const InsarAct* dsrc, void* stream) {
  int rc;
  if ((rc = insar_check_act(ddst, "insar_resize_bilinear_bwd", "ddst"))) return rc;
  if ((rc = insar_check_act(dsrc, "insar_resize_bilinear_bwd", "dsrc"))) return rc;
  if ((rc = check_resize(ddst, dsrc, "insar_resize_bilinear_bwd"))) return rc;
  int grid = insar_grid_cap((int64_t)dsrc->B * dsrc->H);
  hipStream_t s = (hipStream_t)stream;
  const float sh = (float)dsrc->H / (float)ddst->H, sw = (float)dsrc->W / (float)ddst->W;
  if (dsrc->dtype == INSAR_BF16) hipLaunchKernelGGL(resize_bwd_kernel<bf16_t>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*ddst), make_view(*dsrc), sh, sw);
  else hipLaunchKernelGGL(resize_bwd_kernel<float>, dim3(grid), dim3(PW_THREADS), 0, s, make_view(*ddst), make_view(*dsrc), sh, sw);
  INSAR_CHECK_LAUNCH("insar_resize_bilinear_bwd");
  return INSAR_OK;
}

// ---------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------
__global__ void pixel_table_kernel(int32_t* tab, int64_t Mpad, int B, int H, int W, int s, int Hb, int Wb, int tail) {
  const int64_t M = (int64_t)B * H * W;
  for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < Mpad; p += (int64_t)gridDim.x * blockDim.x) {
    int32_t v = tail;
    if (p < M) {
      const int n = (int)(p / ((int64_t)H * W));
      const int64_t rem = p - (int64_t)n * H * W;
      const int h = (int)(rem / W), w = (int)(rem - (int64_t)h * W);
      v = (int32_t)(((int64_t)n * (Hb + 2) + h * s + 1) * (Wb + 2) + w * s + 1);
    }
    tab[p] = v;
  }
}

extern "C" int insar_pixel_table(int32_t* tab, int64_t Mpad, int32_t B, int32_t H, int32_t W, int32_t s, int32_t Hb,
                                 int32_t Wb, int32_t tail, void* stream) {
  if (!tab) INSAR_FAIL(INSAR_E_ARG, "insar_pixel_table: null pointer");
  if ((int64_t)B * (Hb + 2) * (Wb + 2) > 0x7fffffffLL) INSAR_FAIL(INSAR_E_SHAPE, "insar_pixel_table: pixel index overflows int32");
  int grid = insar_grid_cap((Mpad + 255) / 256);
  hipLaunchKernelGGL(pixel_table_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, tab, Mpad, B, H, W, s, Hb, Wb, tail);
  INSAR_CHECK_LAUNCH("insar_pixel_table");
  return INSAR_OK;
}

__global__ void scale_kernel(float* p, int64_t n, float s) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] *= s;
}
extern "C" int insar_scale_f32(float* p, int64_t n, float s, void* stream) {
  if (!p) INSAR_FAIL(INSAR_E_ARG, "insar_scale_f32: null pointer");
  hipLaunchKernelGGL(scale_kernel, dim3(insar_grid_cap((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, n, s);
  INSAR_CHECK_LAUNCH("insar_scale_f32");
  return INSAR_OK;
}

// out[i] = x[i] * *scale with the factor read from device memory: the loss functions' backward (d loss / d logits times the
// incoming gradient of the scalar loss, which autograd hands over as a device tensor) without a host read-back.
__global__ void mul_dev_kernel(float* __restrict__ out, const float* __restrict__ x, int64_t n, const float* __restrict__ scale) {
  const float s = *scale;
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = ((const float4*)x)[i];
    v.x *= s; v.y *= s; v.z *= s; v.w *= s;
    ((float4*)out)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) out[(n4 << 2) + threadIdx.x] = x[(n4 << 2) + threadIdx.x] * s;
}
extern "C" int insar_mul_dev_f32(float* out, const float* x, int64_t n, const float* scale, void* stream) {
  if (!out || !x || !scale) INSAR_FAIL(INSAR_E_ARG, "insar_mul_dev_f32: null pointer");
  if (!insar_aligned16(out) || !insar_aligned16(x)) INSAR_FAIL(INSAR_E_ALIGN, "insar_mul_dev_f32: buffers not 16-byte aligned");
  if (n < 1) return INSAR_OK;
  hipLaunchKernelGGL(mul_dev_kernel, dim3(insar_grid_cap((n / 4 + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, out, x, n, scale);
  INSAR_CHECK_LAUNCH("insar_mul_dev_f32");
  return INSAR_OK;
}
