"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/insar_hip.h declares;
the ctypes mirrors of the descriptor structs have the C compiler's size and field offsets."""
import ctypes
import os
import re
import subprocess

import pytest

from insar_unet_ca_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "insar_hip.h")


def _declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(insar_[a-z0-9_]+)\s*\(", txt)))


def test_library_present_and_loads():
    assert os.path.isfile(_lib.LIB_PATH), "build with: python -c 'import __graft_entry__ as g; g.build()'"
    lib = _lib.load()
    assert lib.insar_version() == _lib.ABI_VERSION
    assert lib.insar_last_error() is not None


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/insar_hip.h but not exported"
    # and the Python binding knows every one of them
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared


def test_struct_layouts_match_the_c_compiler(tmp_path):
    structs = {"InsarAct": _lib.InsarAct, "InsarIgemm": _lib.InsarIgemm, "InsarWgrad": _lib.InsarWgrad,
               "InsarBnFinalize": _lib.InsarBnFinalize, "InsarSeFwd": _lib.InsarSeFwd, "InsarBnSeBwd": _lib.InsarBnSeBwd,
               "InsarCam": _lib.InsarCam, "InsarBstat": _lib.InsarBstat}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, st in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = dict(line.split() for line in out.strip().splitlines())
    for cname, st in structs.items():
        assert int(got[cname]) == ctypes.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(st, fname).offset, f"{cname}.{fname}"


def test_argument_validation_without_a_gpu():
    """Entry points validate shapes/pointers before touching the device: exercise the error path."""
    a = _lib.InsarAct(0, 1, 16, 16, 64, 0, 64, _lib.F32, 0)
    with pytest.raises(_lib.InsarError, match="null"):
        _lib.call("insar_maxpool2_fwd", ctypes.byref(a), ctypes.byref(a), None)
    assert _lib.call("insar_igemm_num_mtiles", 129, 128) == 2
    assert _lib.call("insar_igemm_tile_rows", 1 << 20, 128) == 256
    assert _lib.call("insar_ce_blocks", 1) == 1


def test_tile_selection_queries_without_a_gpu():
    """The launch-geometry queries are host arithmetic: the rules DESIGN.md states, checked on the layer shapes of the
    benchmark configuration (B = 16, 256 x 256 tiles)."""
    F32, BF16 = _lib.F32, _lib.BF16
    call = _lib.call
    # per-tap implicit GEMM: 256 x 256 tiles only in bf16, only with N % 256 == 0 and >= 256 such tiles
    assert call("insar_igemm_tile_cols_dt", 16 * 64 * 64, 256, BF16) == 256          # 64^2 level, 256 M tiles x 1
    assert call("insar_igemm_tile_cols_dt", 16 * 64 * 64, 256, F32) == 128
    assert call("insar_igemm_tile_cols_dt", 16 * 32 * 32, 512, BF16) == 128          # 32^2 level: only 128 such tiles
    assert call("insar_igemm_tile_rows", 16 * 16 * 16, 512) == 256                   # 256 x 64 tiles from half-chip grids on
    assert call("insar_igemm_tile_cols_dt", 16 * 16 * 16, 512, BF16) == 64
    # weight gradient: (tile(Cin) << 16) | tile(Cout)
    pair = lambda ci, co, dt: divmod(call("insar_wgrad_tile_pair", ci, co, dt), 1 << 16)
    assert pair(1024, 512, BF16) == (256, 256) and pair(256, 128, BF16) == (128, 128) and pair(128, 64, BF16) == (128, 64)
    assert pair(64, 64, BF16) == (64, 64) and pair(256, 256, F32) == (128, 128) and pair(128, 64, F32) == (64, 64)
    # row-of-taps kernel: every level of the U-Net (W % 64 == 0, or W = 16 / 32 with whole K steps per image), tiles <= 128
    act = lambda h, w, c, dt: ctypes.byref(_lib.InsarAct(0, 16, h, w, c, 0, c, dt, 0))
    rows = lambda h, w, ci, co, dt: divmod(call("insar_wgrad_conv3_tile", act(h, w, ci, dt), co), 1 << 16)
    assert rows(256, 256, 64, 64, BF16) == (64, 64) and rows(64, 64, 512, 256, BF16) == (128, 128)
    assert rows(32, 32, 512, 512, BF16) == (128, 128) and rows(16, 16, 1024, 1024, F32) == (128, 128)
    assert rows(48, 80, 64, 64, BF16) == (0, 0) and rows(3, 16, 64, 64, BF16) == (0, 0)     # per-tap kernel there
    # ... and its 256 x 128 / 128 x 256 six-phase variant: bf16, 256 channels on one side and 128 on the other
    rowsx = lambda h, w, ci, co, dt: divmod(call("insar_wgrad_conv3x_tile", act(h, w, ci, dt), co), 1 << 16)
    assert rowsx(64, 64, 512, 256, BF16) == (256, 128) and rowsx(16, 16, 1024, 1024, BF16) == (256, 128)
    assert rowsx(64, 64, 128, 256, BF16) == (128, 256) and rowsx(128, 128, 256, 128, BF16) == (256, 128)
    rowsk = lambda h, w, ci, co, dt: (divmod(call("insar_wgrad_conv3k_tile", act(h, w, ci, dt), co), 1 << 16), call("insar_wgrad_conv3k_slices", act(h, w, ci, dt), co))
    assert rowsk(256, 256, 64, 64, BF16) == ((64, 64), 8) and rowsk(256, 256, 128, 64, BF16) == ((128, 64), 4)
    assert rowsk(128, 128, 64, 128, BF16) == ((64, 128), 4) and rowsk(128, 128, 128, 128, BF16) == ((128, 128), 2)
    assert rowsk(128, 128, 64, 64, BF16) == ((0, 0), 0) and rowsk(256, 256, 64, 64, F32) == ((0, 0), 0)     # a 256-pixel K step needs W % 256 == 0
    assert rowsx(128, 128, 128, 128, BF16) == (0, 0) and rowsx(64, 64, 512, 256, F32) == (0, 0) and rowsx(48, 80, 512, 256, BF16) == (0, 0)
    # statistics slab of the flat 3x3 kernel: one row per M tile; persistent work-groups (flip bit 2) with one N tile carry
    # the sums over their tiles: one row per work-group (= CUs, 256 when no device answers), unless the grid is smaller
    big, small = act(256, 256, 128, BF16), act(32, 32, 128, BF16)
    mt = call("insar_conv3x3_flat_num_mtiles", big)
    assert mt == (16 * 258 * 258 + 253) // 254
    assert call("insar_conv3x3_flat_stat_rows", big, 64, 0) == mt and call("insar_conv3x3_flat_stat_rows", big, 64, 4) == 256
    assert call("insar_conv3x3_flat_stat_rows", small, 64, 4) == call("insar_conv3x3_flat_num_mtiles", small) < 256


def test_row_tile_geometry_queries():
    """Applicability of the flat kernel's row tiles (flip bit 3) and the rows of their statistics slab: host-only queries.
    A tile is 256 real pixels = 256 / W whole image rows of one image, staged with d halo columns either side."""
    lib = _lib.load()
    call = lambda name, *a: getattr(lib, name)(*a)
    BF16, F32 = _lib.BF16, _lib.F32
    act = lambda b, h, w, c, dt: ctypes.byref(_lib.InsarAct(0, b, h, w, c, 0, c, dt, 0))
    ok = lambda b, h, w, c, n, dt=BF16: call("insar_conv3x3_flat_rows_ok", act(b, h, w, c, dt), n)
    # every level of the U-Net at 256 x 256 tiles, and the reference's other tile sizes that keep W a power of two
    assert all(ok(16, s, s, 64, 64) == 1 for s in (16, 32, 64, 128, 256))
    assert ok(1, 16, 16, 1024, 1024) == 1 and ok(3, 8, 32, 64, 128) == 1 and ok(2, 2, 128, 64, 64) == 1
    assert ok(16, 256, 256, 64, 64, F32) == 0                     # bf16 only
    assert ok(16, 8, 8, 64, 64) == 0 and ok(16, 512, 512, 64, 64) == 0 and ok(16, 48, 48, 64, 64) == 0    # W < 16, > 256, not 2^k
    assert ok(2, 3, 128, 64, 64) == 0 and ok(2, 12, 32, 64, 64) == 0   # H not a multiple of 256 / W: a tile would straddle images
    assert ok(16, 32, 32, 96, 64) == 0 and ok(16, 32, 32, 64, 96) == 0  # channels in 64s
    dil = lambda h, w, d: call("insar_conv3x3_flat_rows_dil_ok", act(16, h, w, 256, BF16), 256, d)
    assert dil(32, 32, 1) == 1 and dil(32, 32, 2) == 1 and dil(32, 32, 4) == 1 and dil(32, 32, 5) == 0   # 8 * (32 + 2 d) <= 320
    assert dil(16, 16, 2) == 1 and dil(16, 16, 3) == 0 and dil(64, 64, 8) == 1 and dil(64, 64, 9) == 0
    assert dil(32, 32, 0) == 0 and dil(32, 32, 16) == 0
    # statistics slab: one row per tile, whatever the other bits say
    x = act(16, 32, 32, 512, BF16)
    for flags in (8, 8 | 16, 8 | 4, 8 | 2 | (2 << 8)):
        assert call("insar_conv3x3_flat_stat_rows", x, 512, flags) == 16 * 32 * 32 // 256
    # the two-work-group kernel (flip bit 5): row tiles with the same grid conditions, K in 32s; persistent work-groups
    # (bit 2; two per CU) carry the sums: one slab row per work-group where one N tile has more tiles than that
    ok2 = lambda b, h, w, c, n, dt=BF16: call("insar_conv3x3_flat2_rows_ok", act(b, h, w, c, dt), n)
    assert ok2(16, 256, 256, 64, 64) == 1 and ok2(16, 128, 128, 32, 128) == 1 and ok2(2, 2, 128, 64, 64) == 1
    assert ok2(16, 64, 64, 64, 64) == 1 and ok2(16, 16, 16, 1024, 1024) == 1
    assert ok2(16, 8, 8, 64, 64) == 0 and ok2(2, 3, 128, 64, 64) == 0 and ok2(16, 128, 128, 48, 64) == 0 and ok2(16, 128, 128, 64, 64, F32) == 0
    big = act(16, 128, 128, 128, BF16)
    tiles = 16 * 128 * 128 // 256
    assert call("insar_conv3x3_flat_stat_rows", big, 128, 32 | 8) == tiles
    per_wg = call("insar_conv3x3_flat_stat_rows", big, 128, 32 | 8 | 4)
    assert per_wg == call("insar_conv3x3_flat_stat_rows", big, 128, 32 | 4) and 16 <= per_wg <= tiles and per_wg % 16 == 0
    assert call("insar_conv3x3_flat_stat_rows", big, 256, 32 | 8 | 4) == tiles          # two N tiles: per tile
