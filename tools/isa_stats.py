#!/usr/bin/env python
"""Instruction histogram of the loops of one kernel in a gfx950 assembly listing.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -fno-slp-vectorize torchmd_amd/csrc/pair_fast_f32.hip -o /tmp/nb.s
    python tools/isa_stats.py /tmp/nb.s 'list_pair_fast_f32_kernelILi8ELb1ELb1ELb0ELb0E' [--dump N]

Prints registers/occupancy from the kernel's metadata comment block and, for every backward branch (loop),
the mnemonic histogram of its body with a cycle estimate from tools/ubench/valu_rates.hip (plain VALU 2.3,
packed / shift / mul24 4.3, v_cmp 5.3, transcendental 8.2)."""
import collections
import re
import sys


def cost(m, line=""):
    """Issue cost in cycles per wave-instruction per SIMD (tools/ubench/valu_detail.hip, valu_rates.hip; gfx950, 6-8 waves
    per SIMD, 2.4 GHz): plain VALU 2.2; ANY SGPR source operand, v_cmp, v_cndmask with an SGPR/VCC mask, SDWA, v_mov_b64,
    shifts / mul24 / packed fp32 and the VOP3-only integer ops 4.1; transcendentals ~10 (8.1 back to back, 19 alone)."""
    if not m.startswith("v_"):
        return 0.0
    if m.startswith(("v_rsq", "v_rcp", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return 10.0
    half = (m.startswith(("v_pk_", "v_lshl", "v_lshr", "v_ashr", "v_cmp", "v_cndmask", "v_mov_b64", "v_perm", "v_bfe", "v_alignbit",
                          "v_and_or", "v_mbcnt", "v_readlane", "v_writelane", "v_readfirstlane", "v_mad_u", "v_mad_i", "v_mul_lo",
                          "v_mul_hi", "v_lshl_add", "v_add_lshl", "v_mul_f64", "v_fma_f64", "v_add_f64"))
            or "u24" in m or "i24" in m or "sdwa" in m)
    ops = line.split(None, 1)[1] if " " in line.strip() else ""
    srcs = ops.split(",")[1:]
    sgpr_src = any(re.match(r"\s*-?\|?(s\d+|s\[\d+:\d+\]|vcc|exec)", o) for o in srcs)
    return 4.1 if (half or sgpr_src) else 2.2


def main():
    path, pat = sys.argv[1], sys.argv[2]
    dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else -1
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and pat in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    tail = next((i for i in range(end, min(end + 400, len(lines))) if ".end_amdhsa_kernel" in lines[i] or "; NumVgprs" in lines[i]), end)
    for l in lines[end:tail + 40]:
        if re.search(r"; (NumVgprs|NumSgprs|Occupancy|ScratchSize|LDSByteSize|codeLenInByte)", l):
            print(l.strip())
    body = lines[start + 1:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\w+)", l) or re.search(r"s_branch\s+(\.LBB\w+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    print(f"{len(body)} lines, loops: {loops}")
    for k, (a, b) in enumerate(loops):
        hist = collections.Counter()
        cyc = 0.0
        for l in body[a:b + 1]:
            t = l.strip().split()
            if not t or t[0].startswith((".", ";")) or t[0].endswith(":"):
                continue
            hist[t[0]] += 1
            cyc += cost(t[0], l.strip())
        nv = sum(v for m, v in hist.items() if m.startswith("v_"))
        print(f"--- loop {k}: lines {a}..{b}: {sum(hist.values())} instr, {nv} VALU, est. {cyc:.0f} VALU cycles")
        print("   " + ", ".join(f"{m} {v}" for m, v in sorted(hist.items(), key=lambda kv: -kv[1])))
        if k == dump:
            print("\n".join(body[a:b + 1]))


if __name__ == "__main__":
    main()
