#!/usr/bin/env python3
"""Per-kernel VGPR / AGPR / scratch / occupancy / LDS of the library build (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/resource_usage.py [substring ...]   (no GPU needed)"""
import os, re, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "gencomm_amd", "csrc", "gencomm_abi.hip")


def main():
    pats = sys.argv[1:]
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function",
                            "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-Rpass-analysis=kernel-resource-usage",
                            SRC, "-o", os.path.join(d, "lib.so")], capture_output=True, text=True)
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    names = [b.split("\n")[0].strip() for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    print(f"{'VGPR':>5} {'AGPR':>5} {'scratch':>8} {'occ':>4} {'LDS':>7}  kernel")
    for b, n in zip(blocks, dem):
        if pats and not any(p in n for p in pats):
            continue
        g = lambda k: int(m.group(1)) if (m := re.search(k + r": (\d+)", b)) else -1
        vals = (g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"))
        print("%5d %5d %8d %4d %7d  %s" % (*vals, n[:110]))


if __name__ == "__main__":
    main()
