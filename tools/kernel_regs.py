#!/usr/bin/env python3
"""Register / scratch usage of the kernels in a built library: carves the gfx950 code objects out of the clang offload
bundles inside the .so and prints .vgpr_count / .sgpr_count / spills per kernel (llvm-readelf --notes).
    python tools/kernel_regs.py [library.so] [name-filter]"""
import os, struct, subprocess, sys, tempfile

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "point_cloud_registration_amd", "libpcr_hip.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
data = open(lib, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
pos = 0
with tempfile.TemporaryDirectory() as d:
    k = 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
            trip = data[off:off + tl].decode(); off += tl
            if "gfx950" in trip and sz > 0:
                path = os.path.join(d, f"co{k}.elf"); k += 1
                open(path, "wb").write(data[i + o:i + o + sz])
                out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
                cur = {}
                for line in out.splitlines():
                    line = line.strip()
                    for key in (".name:", ".vgpr_count:", ".sgpr_count:", ".vgpr_spill_count:", ".private_segment_fixed_size:"):
                        if line.startswith(key) or line.startswith("- " + key):
                            cur[key] = line.split(key)[1].strip()
                    if len(cur) == 5:
                        if flt in cur[".name:"]:
                            print(f"{cur['.name:']:90s} vgpr {cur['.vgpr_count:']:>4s} sgpr {cur['.sgpr_count:']:>4s} "
                                  f"spill {cur['.vgpr_spill_count:']:>3s} scratch {cur['.private_segment_fixed_size:']:>5s}")
                        cur = {}
        pos = i + 24
