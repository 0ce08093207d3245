#!/usr/bin/env python3
"""Static instruction budget of a kernel, per phase.  Compiles one .hip source to gfx950 assembly, takes the kernel whose mangled name contains
KERNEL, and counts the instructions between the `; MARK name` comments the source plants (SNP_MARK) -- in LAYOUT order: blocks the compiler moved
(rare paths) are attributed to the marker they follow in the listing, so treat the split as a guide and the total as exact.

    python scripts/isa_budget.py snappier_amd/csrc/decode_chains.hip k_decode_chainsILb1 [-DFOO=1 ...]     (--json for one JSON object)
"""
import json, os, re, subprocess, sys, tempfile

def classify(op):
    if op.startswith(("ds_",)): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait_nop"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_barrier")): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"

def main():
    args = [a for a in sys.argv[1:] if a != "--json"]
    as_json = "--json" in sys.argv
    src, kernel, extra = args[0], args[1], args[2:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fconstexpr-steps=100000000", "-Wno-sometimes-uninitialized",
               "-Wno-unused-function", "--cuda-device-only", "-S", src, "-o", out] + extra
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\S*%s\S*:" % re.escape(kernel), l))
    end = next(i for i in range(start, len(text)) if text[i].strip().startswith("s_endpgm") and any(".section" in t or ".Lfunc_end" in t for t in text[i:i + 400]))
    end = next(i for i in range(start, len(text)) if ".Lfunc_end" in text[i])
    sections, cur = {}, "prologue"
    order = ["prologue"]
    for l in text[start + 1:end]:
        m = re.search(r"; MARK (\w+)", l)
        if m:
            cur = m.group(1)
            if cur not in order: order.append(cur)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"): continue
        op = t.split()[0]
        sections.setdefault(cur, {}).setdefault(classify(op), 0)
        sections[cur][classify(op)] += 1
    meta = {}
    for l in text[end:end + 200]:
        m = re.match(r";\s*(NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize):\s*(\d+)", l)
        if m: meta[m.group(1)] = int(m.group(2))
    cols = ["valu", "salu", "branch", "wait_nop", "lds", "vmem", "smem", "other"]
    tot = {c: sum(sections.get(s, {}).get(c, 0) for s in order) for c in cols}
    if as_json:
        print(json.dumps({"source": src, "kernel": kernel, "flags": extra, "meta": meta, "sections": {s: sections.get(s, {}) for s in order}, "total": tot}))
        return
    print(f"{'section':18s} " + " ".join(f"{c:>8s}" for c in cols) + "      all")
    for s in order:
        d = sections.get(s, {})
        print(f"{s:18s} " + " ".join(f"{d.get(c, 0):8d}" for c in cols) + f" {sum(d.values()):8d}")
    print(f"{'TOTAL':18s} " + " ".join(f"{tot[c]:8d}" for c in cols) + f" {sum(tot.values()):8d}")
    print(meta)

if __name__ == "__main__":
    main()
