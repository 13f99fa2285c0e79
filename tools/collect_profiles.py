#!/usr/bin/env python
"""Copy the rocprofv3 summaries of a profile pass from gpurun_out/ into profiles/ under a round's names and derive the
`*_pmc_traffic.json` that bench.py quotes (HBM bytes per launch = FETCH_SIZE x 2 + WRITE_SIZE in KB, the guide's gfx950
correction; VALU wave-instructions per launch).

    python tools/collect_profiles.py prof_r06 r06 'list_pair_fast_f32_kernel<8, true, true, false, false, 2>' [--commit HASH]
    python tools/collect_profiles.py pmc_c5_r06 r06_c5 'list_pair_fast_f32_kernel<4, true, false, false, false, 2>'

(gpurun_out/<dir>/{kernel_stats.csv, pmc_fetch.txt, pmc_write.txt, pmc_sq.txt, pmc_tcp.txt} as tools/profile_round.sh and
tools/pmc_c5.sh leave them.)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters_of(text, kernel):
    """{counter: value per dispatch} of the block of `kernel` in a summarize_pmc.py listing."""
    out, on = {}, False
    for ln in text.splitlines():
        if not ln.startswith(" "):
            on = kernel in ln
            m = re.search(r"avg_us=([\d.]+)", ln)
            if on and m and "_avg_us" not in out:  # (the first pass listed: FETCH_SIZE)
                out["_avg_us"] = float(m.group(1))
        elif on:
            k, v = ln.split()[:2]
            out[k] = float(v)
    return out


def main():
    src, tag, kernel = sys.argv[1:4]
    commit = sys.argv[sys.argv.index("--commit") + 1] if "--commit" in sys.argv else subprocess.check_output(
        ["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
    d = os.path.join(ROOT, "gpurun_out", src)
    with open(os.path.join(d, "kernel_stats.csv")) as fh:
        stats = fh.read()
    with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w") as fh:
        fh.write(stats)
    pmc = ""
    for name in ("pmc_fetch.txt", "pmc_write.txt", "pmc_sq.txt", "pmc_tcp.txt"):
        p = os.path.join(d, name)
        if os.path.exists(p):
            pmc += open(p).read()
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc.txt"), "w") as fh:
        fh.write(f"# rocprofv3 --pmc passes ({src}; one counter group per pass, --kernel-trace only beside them), code of commit {commit}\n" + pmc)
    c = counters_of(pmc, kernel)
    traffic = {
        "_comment": f"HBM-side traffic of one launch of the kernel below from rocprofv3 --pmc passes (profiles/{tag}_pmc.txt). FETCH_SIZE doubled "
                    "per MI355X_MICROARCH.md (gfx950, wide coalesced streams); WRITE_SIZE uncalibrated.",
        "kernel": kernel,
        "fetch_size_kb_raw": c.get("FETCH_SIZE"),
        "write_size_kb_raw": c.get("WRITE_SIZE"),
        "hbm_bytes_per_launch": int(round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0)),
        "valu_wave_instr_per_launch": c.get("SQ_INSTS_VALU"),
        "avg_us_in_the_fetch_size_pass": c.get("_avg_us"),
        "commit": commit,
    }
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
    m = re.search(r'"[^"\n]*' + re.escape(kernel.split("<")[0]) + r'[^"\n]*",(\d+),([\d.]+),([\d.]+)', stats)
    print(tag, "traffic", traffic["hbm_bytes_per_launch"], "B, VALU", traffic["valu_wave_instr_per_launch"],
          "| kernel trace:", next((ln[-60:] for ln in stats.splitlines() if kernel[:40] in ln.replace('"', "")), m and m.group(0)))


if __name__ == "__main__":
    main()
