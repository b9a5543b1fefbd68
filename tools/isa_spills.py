#!/usr/bin/env python3
"""Spill counts per kernel from the metadata of a hipcc -S listing (tools/isa_stats.sh). usage: tools/isa_spills.py /tmp/isa_x.s [substring]"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
md = txt[txt.index('amdhsa.kernels:'):]
for blk in md.split('  - .agpr_count:')[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
    if pat in name:
        print(f"{name[:70]:70s} sgpr_spill {g('sgpr_spill_count'):>4s} vgpr_spill {g('vgpr_spill_count'):>3s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>4s}")
