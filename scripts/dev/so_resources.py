#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel in a built library, read from the code objects inside it (no GPU):
   scripts/dev/so_resources.py qbold_vi_amd/libqbold_hip.so [name-substring]"""
import re, struct, subprocess, sys, tempfile
data = open(sys.argv[1], "rb").read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
pos = 1
while True:
    pos = data.find(b"\x7fELF", pos)
    if pos < 0:
        break
    e_machine = struct.unpack_from("<H", data, pos + 18)[0]
    if e_machine == 224:   # EM_AMDGPU
        shoff, = struct.unpack_from("<Q", data, pos + 40)
        shentsize, shnum = struct.unpack_from("<HH", data, pos + 58)
        size = shoff + shentsize * shnum
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(data[pos:pos + size]); f.flush()
            notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
            if pat in name:
                print(f"vgpr {get('vgpr_count'):>4} agpr {blk.split()[0]:>4} sgpr {get('sgpr_count'):>4} scratch {get('private_segment_fixed_size'):>5} "
                      f"lds {get('group_segment_fixed_size'):>6} wg {get('max_flat_workgroup_size'):>5}  {name[:120]}")
    pos += 4
