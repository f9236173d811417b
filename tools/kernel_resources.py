"""Register / LDS use of the kernels in one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
    python tools/kernel_resources.py fewshot_detection_amd/csrc/conv.hip [substring ...]
"""
import re
import subprocess
import sys


def main():
    src, needles = sys.argv[1], sys.argv[2:]
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude", "-c", src, "-o",
                          "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    for block in out.split("Function Name: ")[1:]:
        name = block.split()[0]
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        demangled = demangled.replace("(anonymous namespace)::", "").split("(")[0]
        if needles and not all(n in demangled for n in needles):
            continue
        def g(key):
            m = re.search(re.escape(key) + r": (\d+)", block)
            return m.group(1) if m else "?"
        print("%-90s VGPR %3s AGPR %3s spill %s occ %s" % (demangled[:90], g("    VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main()
