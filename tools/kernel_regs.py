"""Print VGPR / AGPR / SGPR counts, spills and scratch of every kernel in a hipcc --save-temps .s file.
Usage: kernel_regs.py file.s [name-filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    blk = "  - .agpr_count:" + blk
    f = {k: v for k, v in re.findall(r"\.(\w+):\s+(\S+)", blk.split("  - .agpr_count:", 2)[1] if blk.count("  - .agpr_count:") > 1 else blk)}
    name = f.get("name", "?")
    if flt and flt not in name:
        continue
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        pass
    print(f"{name:70s} vgpr {f.get('vgpr_count'):>4s} agpr {f.get('agpr_count'):>4s} sgpr {f.get('sgpr_count'):>4s} "
          f"spill {f.get('vgpr_spill_count'):>4s} scratch {f.get('private_segment_fixed_size'):>5s} lds {f.get('group_segment_fixed_size')}")
