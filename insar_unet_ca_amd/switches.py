"""The INSAR_* environment switches, seen from outside: which ones exist, their defaults, and which are set to something else.

Every switch is read where it is used (`os.environ.get("INSAR_X", "default")` in engine.py, deeplab.py, optim.py, tape.py,
_lib.py, bench.py); they select between launch paths the GPU tests show to be equal, for diagnostics and same-box A/B runs.
A stray variable on a box would change WHICH kernels a measurement runs without leaving a trace, so the bench line carries
`switches` (this module's `non_default()`), and the default `bench.py` run refuses to measure with any of them set
(`--allow-switches` overrides). The table is read off the package's own source, so a new switch cannot be forgotten here.
(The reference has no such knobs: Unet-ChannalAttention.py runs one code path.)"""
from __future__ import annotations

import os
import re
from typing import Dict, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
_READ = re.compile(r"""os\.environ(?:\.get\(\s*|\[)["'](INSAR_[A-Z0-9_]+)["'](?:\s*,\s*["']([^"']*)["'])?""")
# switches that do not change which kernels run or how they are launched (host-side behaviour only)
HOST_ONLY = {"INSAR_TAPE"}


def declared(extra_files=()) -> Dict[str, Optional[str]]:
    """name -> default (None: unset by default, any value counts as a change) of every INSAR_* variable the package reads."""
    out: Dict[str, Optional[str]] = {}
    files = [os.path.join(_HERE, f) for f in sorted(os.listdir(_HERE)) if f.endswith(".py") and f != "switches.py"] + list(extra_files)
    for path in files:
        try:
            src = open(path).read()
        except OSError:
            continue
        for name, default in _READ.findall(src):
            if out.get(name) is None:
                out[name] = default or None
    return out


def non_default(extra_files=()) -> Dict[str, dict]:
    """The INSAR_* variables of this process's environment whose value differs from the default (or that the package does
    not know at all): name -> {"value", "default", "kernel_selecting"}."""
    known = declared(extra_files)
    res = {}
    for name, value in sorted(os.environ.items()):
        if not name.startswith("INSAR_"):
            continue
        default = known.get(name)
        if name in known and default is not None and value == default:
            continue
        res[name] = {"value": value, "default": default, "kernel_selecting": name not in HOST_ONLY,
                     **({} if name in known else {"unknown": True})}
    return res
