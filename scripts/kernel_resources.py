"""Register / static-LDS / workgroup-size table of the kernels in the emitted gfx950 assembly (build.py --emit-asm)."""
import glob
import os
import re
import subprocess
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "is-dqn_amd", "build")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for f in sorted(glob.glob(os.path.join(root, "*.s"))):
    s = open(f).read()
    for blk in re.split(r"\n  - \.agpr_count:", s)[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
        name = g("name").group(1)
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        d = re.sub(r"\(.*", "", d).replace("isdqn::", "").replace("void ", "")
        if pat and not re.search(pat, d):
            continue
        ag = re.match(r"\s*(\d+)", blk).group(1)
        print(f"{d[:105]:105s} vgpr {g('vgpr_count').group(1):>4} agpr {ag:>4} sgpr {g('sgpr_count').group(1):>4} "
              f"lds {g('group_segment_fixed_size').group(1):>6} wg {g('max_flat_workgroup_size').group(1):>5} spill {g('vgpr_spill_count').group(1)}")
