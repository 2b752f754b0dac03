// Shared device/host helpers for libinsar_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "insar_hip.h"

#define INSAR_WAVE 64

// ---- error plumbing -----------------------------------------------------------------------
void insar_set_error(const char* fmt, ...);
#define INSAR_FAIL(code, ...)        \
  do {                               \
    insar_set_error(__VA_ARGS__);    \
    return (code);                   \
  } while (0)
#define INSAR_CHECK_LAUNCH(name)                                         \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      insar_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return -(int)e__;                                                  \
    }                                                                    \
  } while (0)

// ---- kernel-variant knobs (insar_tune_set / insar_tune_get, api.hip) ---------------------------
enum InsarKnob { KNOB_WGRAD3_M32 = 0, KNOB_IGEMM_XWIDE_MIN, KNOB_IGEMM_WIDE_MIN, KNOB_WGRAD_TILE_MAX, KNOB_WGRAD3X_VAR, KNOB_C64_GRID_BWD, KNOB_COUNT };
int insar_knob(int id);

static inline bool insar_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- element types ------------------------------------------------------------------------
struct bf16_t { uint16_t v; };

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> { static constexpr int kPerChunk = 4; static constexpr int kDtype = INSAR_F32; };
template <> struct ElemTraits<bf16_t> { static constexpr int kPerChunk = 8; static constexpr int kDtype = INSAR_BF16; };

__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(uint16_t, h);
}
// two values -> one packed dword (lo in bits 0-15) as ONE v_cvt_pk_bf16_f32: written with scalar conversions and shifts
// hipcc pairs the conversions across dwords and repairs the halves with and / shl / or (6-8 VALU per 4 values instead of 2)
typedef float insar_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 insar_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  const insar_f32x2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, insar_bf16x2_t));
}

// 16-byte chunk <-> float lanes
template <typename T> struct Chunk;
template <> struct Chunk<float> {
  static constexpr int N = 4;
  __device__ __forceinline__ static void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
  }
  __device__ __forceinline__ static uint4 pack(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
  }
};
template <> struct Chunk<bf16_t> {
  static constexpr int N = 8;
  __device__ __forceinline__ static void unpack(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
  }
  __device__ __forceinline__ static uint4 pack(const float* f) {
    uint4 u;
    u.x = pack2_bf16(f[0], f[1]); u.y = pack2_bf16(f[2], f[3]);
    u.z = pack2_bf16(f[4], f[5]); u.w = pack2_bf16(f[6], f[7]);
    return u;
  }
};

// ---- padded NHWC view ---------------------------------------------------------------------
struct ActView {
  char* base;      // byte pointer to element 0 of the padded buffer
  int B, H, W, C, c_off, c_len;
  int esize;       // bytes per element
  __host__ __device__ __forceinline__ int64_t pixel_index(int n, int h, int w) const {
    // interior pixel (h, w) of image n -> index of the padded pixel
    return ((int64_t)n * (H + 2) + (h + 1)) * (W + 2) + (w + 1);
  }
  __host__ __device__ __forceinline__ int64_t elem_offset(int n, int h, int w) const {
    return pixel_index(n, h, w) * C + c_off;
  }
};

static inline ActView make_view(const InsarAct& a) {
  ActView v;
  v.base = (char*)a.ptr; v.B = a.B; v.H = a.H; v.W = a.W; v.C = a.C; v.c_off = a.c_off; v.c_len = a.c_len;
  v.esize = (a.dtype == INSAR_BF16) ? 2 : 4;
  return v;
}

// Validate an activation slice: aligned base, 16-byte-aligned channel slice.
int insar_check_act(const InsarAct* a, const char* who, const char* what);

// ---- LDS store guard (retired) ----------------------------------------------------------------
// Round 1 saw nondeterministic BatchNorm partial sums in ~1 % of the work-groups and attributed them to a wide LDS store
// (ds_write_b128) whose data registers are overwritten by VALU instructions a few slots later; the "guard" materialised
// the values (LDS_PIN), stored, drained (LDS_DRAIN) and kept the source registers alive across the drain (LDS_KEEP). The
// real cause turned out to be the WAR race at the K-step barrier (dma_drain_and_barrier below): with that fixed,
// a build WITHOUT the guard gives one result over 80 repeats of a 24-step training run (tools/lds_guard_experiment.sh,
// round 2), the same as the guarded build. The macros are therefore no-ops; -DINSAR_LDS_GUARD brings the guard back for
// an A/B should the symptom ever return.
#ifdef INSAR_LDS_GUARD
#define LDS_PIN(x) asm volatile("" : "+v"(x))
#define LDS_KEEP(x) asm volatile("" ::"v"(x))
#define LDS_DRAIN() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define LDS_PIN(x)
#define LDS_KEEP(x)
#define LDS_DRAIN()
#endif

// ---- LDS-DMA issued behind the compiler's back ------------------------------------------------
// hipcc (ROCm 7.2) makes every ds_read that follows a __builtin_amdgcn_global_load_lds wait vmcnt(0)
// (it cannot tell the DMA's LDS destination from the buffer being read), which serialises "prefetch the
// next K slab" with "compute the current one". Issued from inline asm the DMA is invisible to the
// waitcnt pass; the kernels then order it themselves: s_waitcnt vmcnt(0) + s_barrier at the end of the
// K step that prefetched, before any wave reads that buffer. M0 carries the wave-uniform LDS byte address.
__device__ __forceinline__ uint32_t lds_offset_of(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void lds_dma16_untracked(const char* gsrc, uint32_t lds_wave_base) {
  uint32_t keep;
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// The barrier that ends a K step does two jobs: it publishes the slab the waves have just waited for (RAW, the
// vmcnt part) and it FREES the slab they have just read for the next LDS-DMA (WAR). hipcc sinks the tail MFMAs of
// a step and the lgkmcnt wait of their ds_reads below a raw s_barrier, so a wave could pass the barrier with its
// last reads still queued in the LDS pipe while a faster wave's DMA was already on its way to the same slot: seen
// as one stale MFMA fragment in ~1e-3 of the 64x64 weight-gradient launches when the DMA path was otherwise idle
// (tools/debug_race.py). Hence lgkmcnt(0) BEFORE every such barrier.
__device__ __forceinline__ void dma_drain_and_barrier() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// ---- wave helpers -------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- per-device one-time launch setup --------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of the loaded code object: the flag that
// remembers "already set" is kept per device (a std::atomic bitmask, one bit per ordinal; devices >= 64 simply set
// the attribute on every launch) so that a process driving several GPUs, or several host threads, never launches a
// > 64 KB LDS kernel on a device that has not had the attribute set. One static mask per call site (per kernel).
#include <atomic>
static inline int insar_current_device() {
  int dev = 0;
  return hipGetDevice(&dev) == hipSuccess ? dev : 0;
}
static inline hipError_t insar_set_lds_once(std::atomic<uint64_t>& mask, const void* func, int bytes) {
  const int dev = insar_current_device();
  const uint64_t bit = dev < 64 ? (1ull << dev) : 0ull;
  if (bit && (mask.load(std::memory_order_acquire) & bit)) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && bit) mask.fetch_or(bit, std::memory_order_release);
  return e;
}
// CU count of the current device (cached per ordinal)
static inline int insar_num_cus() {
  static std::atomic<int> cache[64];
  const int dev = insar_current_device();
  if (dev < 64) { const int c = cache[dev].load(std::memory_order_relaxed); if (c > 0) return c; }
  hipDeviceProp_t p;
  int cus = 256;
  if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
  if (dev < 64) cache[dev].store(cus, std::memory_order_relaxed);
  return cus;
}

static inline int insar_grid_cap(int64_t want, int cap = 2048 * 4) {
  if (want < 1) want = 1;
  return (int)(want > cap ? cap : want);
}
