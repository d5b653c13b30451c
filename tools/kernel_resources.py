#!/usr/bin/env python3
"""Print per-kernel register / LDS / spill figures from the gfx950 assembly metadata
(`make -C <pkg>/csrc asm` writes fql_int4.gfx950.s)."""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "fused-4-bit-dequantize-linear-cuda-kernel_amd/csrc/fql_int4.gfx950.s"
txt = open(path).read()
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - ", meta)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    print(f"{name[:70]:70s} vgpr={g('vgpr_count'):>4s} agpr={g('agpr_count'):>3s} sgpr={g('sgpr_count'):>3s} "
          f"lds={g('group_segment_fixed_size'):>6s} spill={g('vgpr_spill_count')} scratch={g('private_segment_fixed_size')}")
