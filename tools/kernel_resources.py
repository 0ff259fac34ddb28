#!/usr/bin/env python3
"""Developer tool: registers, scratch and occupancy of every gfx950 kernel, as the compiler reports them.
   python tools/kernel_resources.py            # table on stdout
   python tools/kernel_resources.py --json F   # also written to F (profiles/rNN_kernel_resources.json)
Runs `make -C meng_zhang_amd/csrc asm` (hipcc -S -Rpass-analysis=kernel-resource-usage; no GPU needed)."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
        "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
        "LDS Size [bytes/block]": "lds_static"}


def collect():
    out = subprocess.run(["make", "-B", "-C", os.path.join(ROOT, "meng_zhang_amd", "csrc"), "asm"], capture_output=True, text=True)
    if out.returncode:
        sys.stderr.write(out.stderr)
        raise SystemExit(out.returncode)
    kernels, cur = [], None
    for line in out.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass-analysis", line)
        if not m:
            continue
        body = m.group(1)
        if body.startswith("Function Name:"):
            mangled = body.split(":", 1)[1].strip()
            name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
            cur = {"kernel": re.sub(r"^void ", "", name).split("(")[0], "mangled": mangled}
            kernels.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            if k.strip() in KEYS:
                cur[KEYS[k.strip()]] = int(v)
    return kernels


def main():
    ks = collect()
    print("%-78s %5s %5s %7s %5s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "spill", "waves"))
    for k in ks:
        print("%-78s %5d %5d %7d %5d %6d" % (k["kernel"][:78], k["vgpr"], k["sgpr"], k["scratch"], k["vgpr_spill"], k["occupancy"]))
    if "--json" in sys.argv:
        path = sys.argv[sys.argv.index("--json") + 1]
        with open(path, "w") as fh:
            json.dump({"source": "hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage (make asm)", "kernels": ks}, fh, indent=1)


if __name__ == "__main__":
    main()
