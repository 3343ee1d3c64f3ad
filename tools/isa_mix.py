#!/usr/bin/env python3
"""Instruction mix of one gfx950 kernel, from hipcc's own assembly (no GPU needed).

usage: tools/isa_mix.py <source.hip> <substring of the mangled kernel name> [--md] [--loop]
Compiles the translation unit with `-S --cuda-device-only`, cuts the named kernel out and counts opcodes by class
(packed / scalar FMA, moves, DPP moves, VMEM, LDS, SALU, waits).  `--md` prints a markdown table (profiles/).
The counts are STATIC (instructions in the kernel text); the steady-state loop dominates these kernels' text because
it is fully unrolled over a ring period.
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op, line):
    if op.startswith("v_pk_fma") or op.startswith("v_pk_mul") or op.startswith("v_pk_add"):
        return "valu_packed_math"
    if op.startswith("v_fma") or op.startswith("v_fmac") or op.startswith("v_mul_f") or op.startswith("v_add_f") or op.startswith("v_mac"):
        return "valu_dpp_math" if "dpp" in op or "row_" in line or "wave_" in line else "valu_scalar_math"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "valu_mov_dpp" if ("dpp" in op or "wave_sh" in line or "row_sh" in line) else "valu_mov"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
        return "valu_lane"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
        return "vmem_load"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"):
        return "vmem_store"
    if op.startswith("global_atomic") or op.startswith("flat_atomic"):
        return "vmem_atomic"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    md = "--md" in sys.argv
    src, pat = args[0], args[1]
    out = "/tmp/isa_mix_%d.s" % os.getpid()
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                           "-I" + os.path.join(ROOT, "include"), src, "-o", out], stderr=subprocess.DEVNULL)
    txt = open(out).read()
    os.unlink(out)
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", txt, re.M)]
    found = [(p, n) for p, n in starts if pat in n]
    if not found:
        sys.exit("no kernel matches %r; kernels: %s" % (pat, ", ".join(n for _, n in starts)))
    for pos, name in found:
        body = txt[pos:]
        # the kernel text ends at its .Lfunc_end label (a kernel may hold several s_endpgm: the wave programs of the cascade kernels each
        # end the program themselves)
        m = re.search(r"^\.Lfunc_end\d+:", body, re.M)
        body = body[:m.start()] if m else body[:body.index("s_endpgm")]
        meta = txt[pos:]
        res = {}
        for key in ("NumSgprs", "NumVgprs", "ScratchSize", "Occupancy", "sgpr_spill_count", "vgpr_spill_count"):
            m = re.search(r"[;.] *%s[: ]+(\d+)" % key, meta)
            if m:
                res[key] = int(m.group(1))
        cls, ops = collections.Counter(), collections.Counter()
        for l in body.split("\n"):
            l = l.split(";")[0].strip()
            if not l or l.startswith(".") or l.endswith(":"):
                continue
            op = l.split()[0]
            ops[op] += 1
            cls[classify(op, l)] += 1
        total = sum(cls.values())
        valu = sum(v for k, v in cls.items() if k.startswith("valu"))
        if md:
            print("### `%s`\n" % name)
            print("resources: " + ", ".join("%s %d" % kv for kv in res.items()) + "\n")
            print("| class | instructions | share of all | share of VALU |\n|---|---|---|---|")
            for k, v in cls.most_common():
                print("| %s | %d | %.1f %% | %s |" % (k, v, 100.0 * v / total, ("%.1f %%" % (100.0 * v / valu)) if k.startswith("valu") else ""))
            print("| **total** | %d | | VALU %d |\n" % (total, valu))
        else:
            print(name, res)
            for k, v in cls.most_common():
                print("  %-18s %6d  %5.1f %%" % (k, v, 100.0 * v / total))
            print("  total %d  VALU %d" % (total, valu))
            print("  top ops:", ", ".join("%s %d" % kv for kv in ops.most_common(14)))


if __name__ == "__main__":
    main()
