"""VGPR / SGPR / spill / LDS figures of every kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).
python tools/kernel_resources.py /tmp/isa/backend.s [name-substring]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    def g(k):
        m = re.search(r"\." + k + r":\s+(\S+)", blk)
        return m.group(1) if m else "?"
    name = g("name")
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"fftk::|\(fftk::\w+<\w+>\)", "", name)
    if pat in name:
        rows.append((name[:110], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
print("%-110s %5s %6s %5s %6s %8s %8s" % ("kernel", "vgpr", "vspill", "sgpr", "sspill", "lds", "scratch"))
for r in rows:
    print("%-110s %5s %6s %5s %6s %8s %8s" % r)
