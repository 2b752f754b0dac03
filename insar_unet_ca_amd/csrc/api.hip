// Error plumbing and argument validation shared by every entry point of libinsar_hip.so.
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void insar_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* insar_last_error(void) { return g_err; }
extern "C" int insar_version(void) { return INSAR_ABI_VERSION; }

// ---- kernel-variant knobs ------------------------------------------------------------------------
#include <atomic>
static const char* const g_knob_names[KNOB_COUNT] = {
    "wgrad3_m32",     // row-of-taps weight gradient, 128 x 128 bf16 tiles: 1 = v_mfma_f32_32x32x16_bf16 fragments
    "igemm_xwide_min", // per-tap implicit GEMM: 256 x 256 tiles where they give at least this many work-groups (0 = default 256)
    "igemm_wide_min",  // ... 256 x 128 tiles (instead of 256 x 64) where they give at least this many work-groups (0 = default 256)
    "wgrad_tile_max",  // per-tap weight gradient (insar_wgrad), bf16: largest tile edge (0 = default 256; 128 = no 256 x 256 tiles)
    "wgrad3x_var",     // experiment builds only (-DINSAR_EXP_WX): timing ablations of wgrad3x.hip's K loop (wrong results); ignored by the product library
    "c64_grid_bwd",    // 64 -> 64 kernel, input-gradient launches (flip = 1): persistent work-groups (0 = one per CU)
};
static std::atomic<int> g_knobs[KNOB_COUNT] = {};     // defaults: 0
int insar_knob(int id) { return g_knobs[id].load(std::memory_order_relaxed); }
static int knob_index(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < KNOB_COUNT; ++i)
    if (!strcmp(name, g_knob_names[i])) return i;
  return -1;
}
extern "C" int insar_tune_set(const char* name, int32_t value) {
  const int i = knob_index(name);
  if (i < 0) INSAR_FAIL(INSAR_E_ARG, "insar_tune_set: unknown knob '%s'", name ? name : "(null)");
  g_knobs[i].store(value, std::memory_order_relaxed);
  return INSAR_OK;
}
extern "C" int insar_tune_get(const char* name) {
  const int i = knob_index(name);
  if (i < 0) INSAR_FAIL(INSAR_E_ARG, "insar_tune_get: unknown knob '%s'", name ? name : "(null)");
  return g_knobs[i].load(std::memory_order_relaxed);
}

int insar_check_act(const InsarAct* a, const char* who, const char* what) {
  if (!a || !a->ptr) INSAR_FAIL(INSAR_E_ARG, "%s: %s is null", who, what);
  if (a->dtype != INSAR_F32 && a->dtype != INSAR_BF16) INSAR_FAIL(INSAR_E_DTYPE, "%s: %s has dtype %d", who, what, a->dtype);
  const int ch = a->dtype == INSAR_BF16 ? 8 : 4;
  if (a->B < 1 || a->H < 1 || a->W < 1 || a->C < 1 || a->c_len < 1 || a->c_off < 0 || a->c_off + a->c_len > a->C)
    INSAR_FAIL(INSAR_E_SHAPE, "%s: %s has a bad extent B=%d H=%d W=%d C=%d slice=[%d,+%d)", who, what, a->B, a->H, a->W,
               a->C, a->c_off, a->c_len);
  if ((a->C % ch) || (a->c_off % ch) || (a->c_len % ch))
    INSAR_FAIL(INSAR_E_SHAPE, "%s: %s channel slice must be a multiple of %d elements (16 bytes)", who, what, ch);
  if (!insar_aligned16(a->ptr)) INSAR_FAIL(INSAR_E_ALIGN, "%s: %s is not 16-byte aligned", who, what);
  const long long bytes = (long long)a->B * (a->H + 2) * (a->W + 2) * a->C * (a->dtype == INSAR_BF16 ? 2 : 4);
  if (bytes <= 0) INSAR_FAIL(INSAR_E_SHAPE, "%s: %s size overflow", who, what);
  return INSAR_OK;
}
