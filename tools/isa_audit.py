#!/usr/bin/env python3
"""Audit of every gfx950 kernel in the built libpdwt_hip.so (no GPU needed): the code objects are cut out of the library's offload
bundles and disassembled; per kernel: instructions, v_readlane / v_writelane (SGPR spills parked in VGPR lanes and read back per use),
scratch instructions, and the spill counts of the kernel metadata.  Found the 6.7k-15.6k v_readlane of the first long-bank SWT inverse.
usage: python tools/isa_audit.py [--all]     (default: kernels with read-lanes > 2 % of their instructions, or any scratch / VGPR spill)"""
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def audit(lib=None):
    """[(mangled name, Counter of opcodes, metadata dict)] for every gfx950 kernel of the library."""
    lib = lib or os.path.join(ROOT, "pdwt_amd", "lib", "libpdwt_hip.so")
    data = open(lib, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    rows = []
    with tempfile.TemporaryDirectory() as td:
        i, n = 0, 0
        while True:
            i = data.find(magic, i)
            if i < 0:
                break
            q = i + len(magic)
            ne = struct.unpack_from("<Q", data, q)[0]
            q += 8
            for _ in range(ne):
                off, size, tl = struct.unpack_from("<QQQ", data, q)
                q += 24
                triple = data[q:q + tl].decode()
                q += tl
                if "gfx950" not in triple or size == 0:
                    continue
                path = os.path.join(td, "co_%03d.elf" % n)
                n += 1
                open(path, "wb").write(data[i + off:i + off + size])
                notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
                meta = {}
                for blk in notes.split("- .agpr_count:")[1:]:
                    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                    meta[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size")}
                dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
                cur, cnt = None, None
                for line in dis.split("\n"):
                    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                    if m:
                        if cur:
                            rows.append((cur, cnt, meta.get(cur, {})))
                        cur, cnt = m.group(1), collections.Counter()
                        continue
                    t = line.strip().split()
                    if cur and t and not t[0].startswith(("/", ".")):
                        cnt[t[0]] += 1
                if cur:
                    rows.append((cur, cnt, meta.get(cur, {})))
            i += 1
    return [r for r in rows if r[2]]


def demangle(name):
    return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void pdwt::", "")


if __name__ == "__main__":
    show_all = "--all" in sys.argv
    rows = audit()
    print("| kernel | instructions | v_readlane | v_writelane | scratch | SGPR spills | VGPR spills | VGPRs |")
    print("|---|---|---|---|---|---|---|---|")
    flagged = 0
    for name, c, m in sorted(rows, key=lambda r: -r[1]["v_readlane_b32"]):
        tot = sum(c.values())
        rl, wl = c["v_readlane_b32"], c["v_writelane_b32"]
        sc = sum(v for k, v in c.items() if k.startswith("scratch_"))
        bad = rl > 0.02 * tot or sc > 0 or m.get("vgpr_spill_count", 0) > 0
        flagged += bad
        if show_all or bad:
            print("| `%s` | %d | %d | %d | %d | %d | %d | %d |" % (demangle(name), tot, rl, wl, sc, m.get("sgpr_spill_count", 0), m.get("vgpr_spill_count", 0), m.get("vgpr_count", 0)))
    print("\n%d kernels, %d flagged" % (len(rows), flagged))
