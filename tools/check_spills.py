"""List the gfx950 kernels of multimodal_amd/csrc/*.hip that use scratch memory (register spills) or lose occupancy:
    python tools/check_spills.py [file.hip ...]
Compiles each file with -Rpass-analysis=kernel-resource-usage (about a minute per large file) and prints VGPRs / scratch bytes per lane /
SGPRs for every kernel whose scratch size is non-zero.  A spilling hot kernel once cost 45 % of the step (round 2, LN-fold epilogues)."""
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "multimodal_amd" / "csrc"


def check(src: Path):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage", "-c", str(src),
           "-o", f"/tmp/_spill_{src.stem}.o"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    out = []
    for b in re.split(r"remark: [^\n]*Function Name: ", res.stderr)[1:]:
        name = b.split("\n")[0].split(" [-R")[0].strip()

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1

        out.append((name, g("VGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("SGPRs"), g(r"Occupancy \[waves/SIMD\]")))
    return src.name, res.returncode, out


if __name__ == "__main__":
    files = [Path(a) for a in sys.argv[1:]] or sorted(CSRC.glob("*.hip"))
    bad = 0
    with ThreadPoolExecutor(max_workers=8) as ex:
        for name, rc, rows in ex.map(check, files):
            spilled = [r for r in rows if r[2] > 0]
            print(f"{name}: rc={rc}, {len(rows)} kernels, {len(spilled)} with scratch")
            for r in spilled:
                print("   ", subprocess.run(["c++filt", r[0]], capture_output=True, text=True).stdout.strip()[:150], "VGPR", r[1], "scratch B/lane", r[2], "SGPR", r[3])
            bad += len(spilled)
    sys.exit(0)
