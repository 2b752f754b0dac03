"""Instruction mix of the MFMA loops of a HIP source file (CPU only: hipcc cross-compiles gfx950).
usage: python tools/isa_mix.py insar_unet_ca_amd/csrc/wgrad3.hip [name-filter]
For every kernel whose mangled name contains the filter, every basic block with MFMAs: MFMA / LDS / VALU / SALU / VMEM
counts and the issue-slot estimate of MI355X_MICROARCH.md (MFMA holds the SIMD's vector issue for 8 cycles, a plain VALU
or LDS instruction ~4 from one wave), next to the matrix-pipe cycles of the block; plus VGPR count and spills."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                    "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", s):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    for f in re.split(r"\n(?=_Z\S*:)", s):
        name = f.split(":")[0]
        if not name.startswith("_Z") or flt not in name:
            continue
        blocks, cur = [], None
        for ln in f.split("\n"):
            m = re.match(r"^(\.LBB\d+_\d+):", ln)
            if m:
                cur = [m.group(1), []]
                blocks.append(cur)
            elif cur is not None and ln.startswith("\t") and not ln.startswith("\t."):
                cur[1].append(ln.strip())
        print(name, "vgpr/spill", meta.get(name))
        for lab, ins in blocks:
            mf16 = sum(1 for i in ins if i.startswith("v_mfma") and "16x16" in i)
            mf32 = sum(1 for i in ins if i.startswith("v_mfma") and "32x32" in i)
            if mf16 + mf32 == 0:
                continue
            ds = sum(1 for i in ins if i.startswith("ds_"))
            va = sum(1 for i in ins if i.startswith("v_") and not i.startswith("v_mfma"))
            sa = sum(1 for i in ins if i.startswith("s_"))
            vm = sum(1 for i in ins if i.startswith(("global_", "buffer_", "flat_")))
            pipe = 16 * mf16 + 32 * mf32
            if any("_f32 " in i or i.endswith("_f32") for i in ins if i.startswith("v_mfma") and "x4_f32" in i):
                pipe = 32 * mf16
            issue = 8 * (mf16 + mf32) + 4 * (va + ds) + 2 * sa + 8 * vm
            print(f"  {lab}: {len(ins)} instr: mfma16 {mf16} mfma32 {mf32} lds {ds} valu {va} salu {sa} vmem {vm}"
                  f" | matrix-pipe {pipe} cyc, issue estimate {issue} cyc per wave")


if __name__ == "__main__":
    main()
