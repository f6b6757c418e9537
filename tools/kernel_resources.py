"""Per-kernel register / spill / LDS / occupancy table of the HIP library's sources (hipcc -Rpass-analysis=kernel-resource-usage; compiles to
/dev/null, nothing is installed).  `python tools/kernel_resources.py [repo_root] [extra hipcc flags...]` -- used to check that an edit left the
hot instantiations' allocation alone before spending GPU time on an A/B."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taco_amd import build as B  # noqa: E402


def resources(root=ROOT, extra=()):
    src = os.path.join(root, "taco_amd", "csrc", "taco_capi.hip")
    cmd = [B.HIPCC] + [f for f in B.FLAGS if f not in ("-shared",)] + ["-c", "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src] + list(extra)
    err = subprocess.run(cmd, cwd=os.path.dirname(src), capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"^void |taco::|\(taco::\w+\)$|\(.*\)$", "", name)}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return rows


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else ROOT
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    print(f"{'kernel':100s} vgpr agpr sgpr scratch occ sspill vspill   lds")
    for r in resources(root, extra):
        print(f"{r['name'][:100]:100s} {r.get('vgpr', 0):4d} {r.get('agpr', 0):4d} {r.get('sgpr', 0):4d} {r.get('scratch', 0):7d} {r.get('occ', 0):3d} {r.get('sspill', 0):6d} {r.get('vspill', 0):6d} {r.get('lds', 0):5d}")
