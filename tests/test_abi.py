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
    # statistics slab of the flat 3x3 kernel: one row per M tile; persistent work-groups (flip bit 2) with one N tile carry
    # the sums over their tiles: one row per work-group (= CUs, 256 when no device answers), unless the grid is smaller
    big, small = act(256, 256, 128, BF16), act(32, 32, 128, BF16)
    mt = call("insar_conv3x3_flat_num_mtiles", big)
    assert mt == (16 * 258 * 258 + 253) // 254
    assert call("insar_conv3x3_flat_stat_rows", big, 64, 0) == mt and call("insar_conv3x3_flat_stat_rows", big, 64, 4) == 256
    assert call("insar_conv3x3_flat_stat_rows", small, 64, 4) == call("insar_conv3x3_flat_num_mtiles", small) < 256
