"""ISA check of the direct-W GEMM kernels (gemm.hip, BDIR): between a `buffer_load_dwordx4` into a W slot and the `s_waitcnt vmcnt` that covers it,
nothing but the load itself may touch the slot's registers -- the loads are inline asm, so a compiler-made copy (v_mov) of a slot in that window reads
registers whose data has not landed.  Prints the VMEM / wait skeleton of the K loop and every suspicious instruction.
    python tools/check_bdir_isa.py [gemm.s]        (without an argument: compiles multimodal_amd/csrc/gemm.hip to /tmp/mmamd_gemm.s first)"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def regs_of(tok: str):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(path: Path, verbose: bool) -> int:
    text = path.read_text()
    bad = 0
    for m in re.finditer(r"^(_ZN5mmamd2[23]gemm_bf16_nt_kernel_ppg?I\w*):", text, re.M):
        name = m.group(1)
        body = text[m.end():text.index(".Lfunc_end", m.end())]
        if "buffer_load_dwordx4" not in body:
            continue
        lines = [ln.strip() for ln in body.split("\n")]
        slots = set()
        for t in lines:
            if t.startswith("buffer_load_dwordx4"):
                slots |= regs_of(t.split()[1].rstrip(","))
        # (1) inside the K loop (blocks of loop depth 2) a slot register is only ever written by its load and read by v_mfma: no copies, no spills
        # (2) inside a block, nothing touches a slot between its load and the next vmcnt wait of the block
        depth2 = False
        pending = set()
        nload = nbad = 0
        for t in lines:
            if re.match(r"^\.LBB\d+_\d+:", t):
                depth2 = "Depth=2" in t
                pending = set()
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            op = t.split()[0]
            rest = t[len(op):].split(";")[0]
            touched = set()
            for tok in re.findall(r"v\[\d+:\d+\]|v\d+", rest):
                touched |= regs_of(tok)
            if op == "buffer_load_dwordx4":
                pending |= regs_of(rest.split(",")[0].strip())
                nload += 1
                continue
            if op == "s_waitcnt" and "vmcnt" in t:
                pending = set()
                continue
            if op.startswith("s_"):
                continue
            why = None
            if touched & pending:
                why = "touches a slot whose load is in flight"
            elif depth2 and (op.startswith("v_mov") or op.startswith("scratch_") or op.startswith("v_accvgpr")) and touched & slots:
                why = "copies / spills a W slot inside the K loop"
            if why:
                nbad += 1
                if verbose or nbad <= 6:
                    print(f"  SUSPECT ({why}): {t[:100]}")
        print(f"{name[:84]}: {nload} slot loads, {nbad} suspect instructions")
        bad += nbad
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1 and not sys.argv[1].startswith("-"):
        src = Path(sys.argv[1])
    else:
        src = Path("/tmp/mmamd_gemm.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT / 'include'}", "-S", "--cuda-device-only",
                        str(ROOT / "multimodal_amd" / "csrc" / "gemm.hip"), "-o", str(src)], check=True, capture_output=True)
    sys.exit(1 if check(src, "-v" in sys.argv) else 0)
