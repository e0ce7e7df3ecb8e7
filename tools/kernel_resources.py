"""Developer tool: registers, scratch and occupancy of the kernels of ndlqr_hip.hip (or another unit), from
hipcc -Rpass-analysis=kernel-resource-usage (device-only compile, no GPU needed).

    python tools/kernel_resources.py [name-filter ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    filters = [a for a in sys.argv[1:] if not a.startswith("-")] or [""]
    src = os.path.join(ROOT, "rslqr_amd", "csrc", "ndlqr_hip.hip")
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC",
               "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "rslqr_amd", "csrc"),
               "-c", src, "-o", os.path.join(tmp, "dev.o"), "-Rpass-analysis=kernel-resource-usage"]
        txt = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
    filt = subprocess.run(["c++filt"], input=txt, stdout=subprocess.PIPE, text=True).stdout
    for blk in re.split(r"remark: [^\n]*Function Name: ", filt)[1:]:
        name = blk.split("\n")[0].split("(")[0].replace("void ndlqr::", "").replace("ndlqr::", "")
        if not any(f in name for f in filters):
            continue
        g = lambda k: re.search(k + r": (\S+)", blk).group(1)  # noqa: E731
        print("%-58s VGPR %3s AGPR %3s scratch %4s occ %s LDS %s" % (
            name[:58], g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
            g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main()
