#!/usr/bin/env python3
"""Compiles every HIP source of the product library to gfx950 assembly (device only) and lists the
kernels whose vector registers spilled to scratch (.vgpr_spill_count > 0) with
their VGPR counts -- a spill inside a hot loop is scratch traffic on the kernel's critical path
(round 3: the 8-wave CTC lattice kernels, found by reading the metadata, -4 % once removed).
usage: tools/check_spills.py [source.hip ...]     exit status 1 if any kernel spills"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stanford-ctc_amd", "csrc")


def main():
    srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in srcs:
            out = os.path.join(tmp, os.path.basename(src) + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + CSRC,
                            "-S", "--cuda-device-only", "-o", out, src], check=True,
                           stderr=subprocess.DEVNULL)
            text = open(out).read()
            if "amdhsa.kernels" not in text:
                print("%-20s no kernels" % os.path.basename(src))
                continue
            blocks = text[text.index("amdhsa.kernels"):].split("  - .agpr_count")[1:]
            n_bad = n_sg = 0
            for b in blocks:
                name = re.search(r"\.name:\s+(\S+)", b).group(1)
                vg = int(re.search(r"\.vgpr_count:\s+(\d+)", b).group(1))
                vs = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1))
                ss = int(re.search(r"\.sgpr_spill_count:\s+(\d+)", b).group(1))
                n_sg += 1 if ss else 0
                if vs:          # SGPR spills go to VGPR lanes (v_writelane), not to memory: counted, not flagged
                    n_bad += 1
                    print("  SPILL %s: %d VGPRs, %d VGPR / %d SGPR spills" % (name, vg, vs, ss))
            print("%-20s %3d kernels, %d with VGPR spills (scratch), %d with SGPR spills (to VGPR lanes)"
                  % (os.path.basename(src), len(blocks), n_bad, n_sg))
            bad += n_bad
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
