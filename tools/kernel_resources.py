"""Developer tool: registers, scratch and occupancy of the kernels of ndlqr_hip.hip (or another unit), from
hipcc -Rpass-analysis=kernel-resource-usage (device-only compile, no GPU needed).

    python tools/kernel_resources.py [name-filter ...]
    python tools/kernel_resources.py --instance 12 4 [name-filter ...]     (small_instance.hip for one block size)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    extra = []
    src = os.path.join(ROOT, "rslqr_amd", "csrc", "ndlqr_hip.hip")
    if args[:1] == ["--instance"]:
        extra = ["-DNDLQR_INST_NX=" + args[1], "-DNDLQR_INST_NU=" + args[2]]
        src = os.path.join(ROOT, "rslqr_amd", "csrc", "small_instance.hip")
        args = args[3:]
    filters = [a for a in args if not a.startswith("-")] or [""]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC",
               "--cuda-device-only"] + extra + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "rslqr_amd", "csrc"),
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
