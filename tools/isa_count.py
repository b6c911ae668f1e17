#!/usr/bin/env python3
"""Instruction counts of one kernel in `hipcc -S` output (tools/isa.sh): by kind, plus registers / scratch / LDS.

    python tools/isa_count.py /tmp/isa/pg_anchor.s 'k_probe<6, false, 1, 8, false, false>'

The name is matched against the demangled symbol (c++filt) as a prefix."""
import collections
import re
import subprocess
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    text = open(path).read()
    syms = sorted(set(re.findall(r"^(_Z\w+):", text, re.M)))
    dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.split("\n")
    hits = [(s, d) for s, d in zip(syms, dem) if d.startswith("void pg::" + want) or d.startswith(want) or want in d]
    for s, d in hits:
        body = text.split(f"\n{s}:", 1)[1].split(".Lfunc_end", 1)[0]
        kinds = collections.Counter()
        for ln in body.split("\n"):
            ln = ln.strip()
            if not ln or ln.startswith((";", ".", "//")) or ln.endswith(":"):
                continue
            op = ln.split()[0]
            if op.startswith("v_"):
                kinds["valu"] += 1
            elif op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
                kinds[op.split("_")[1]] += 1
            elif op.startswith(("s_cbranch", "s_branch")):
                kinds["branch"] += 1
            elif op.startswith("s_"):
                kinds["salu"] += 1
            elif op.startswith("ds_"):
                kinds["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                kinds["vmem" if not op.startswith("scratch_") else "scratch"] += 1
            else:
                kinds["other"] += 1
        meta = {}
        blk = text.split(f".amdhsa_kernel {s}", 1)
        if len(blk) > 1:
            for key in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size"):
                m = re.search(rf"\.amdhsa_{key} (\d+)", blk[1])
                if m:
                    meta[key.replace("next_free_", "").replace("_fixed_size", "")] = int(m.group(1))
        print(d)
        print("   ", dict(kinds), meta)


if __name__ == "__main__":
    main()
