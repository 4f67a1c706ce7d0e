"""Static hazard check of the shipped gfx950 code objects (advisor finding r3, the ROLL stencil of pdegym_1d_body.h).

Some DPP operations are written out in ``asm volatile`` blocks with hand-counted wait states (pdegym_1d_body.h, instantiated by pdegym_1d_rollout.hip: the ROLL form of the
parabolic stencil; pdegym_ns_common.h: the Jacobi row blocks).  The compiler's hazard recognizer does not look inside an asm
block, so a compiler upgrade that schedules differently AROUND the blocks could silently violate
  * VALU writes a VGPR  -> DPP reads that VGPR:     2 wait states,
  * VALU writes EXEC    -> DPP operation:           5 wait states
(CDNA3/4 ISA guide, "manually inserted wait states").  This test disassembles every device code object of the library as built
and checks both rules for EVERY DPP instruction in program order -- a violation fails the CPU suite instead of corrupting a
rollout on the GPU.  (Dynamic predecessors across branches are not followed; the asm blocks carry their two wait states inside
the block, so for them only the EXEC rule depends on the surroundings.)
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _vregs(operand):
    out = set()
    for m in _REG.finditer(operand):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _disassemble(obj, tmp):
    work = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, work)                      # --offloading writes the extracted bundles next to its input
    subprocess.run([OBJDUMP, "--offloading", work], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    co = [f for f in os.listdir(tmp) if f.startswith(os.path.basename(obj)) and "hipv4-amdgcn" in f]
    assert co, f"no gfx950 bundle in {obj}"
    return subprocess.run([OBJDUMP, "-d", os.path.join(tmp, co[0])], stdout=subprocess.PIPE, check=True).stdout.decode()


def _check(text):
    """Returns (number of DPP instructions, list of violations)."""
    insts = []              # (mnemonic, operands string) of the current function, program order
    bad, n_dpp, func = [], 0, "?"
    for line in text.splitlines():
        if line.endswith(">:"):
            func, insts = line.split("<")[-1][:-2], []
            continue
        if not line.startswith("\t"):
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        mn, _, ops = body.partition(" ")
        if "_dpp" in mn:
            n_dpp += 1
            parts = [p.strip() for p in ops.split(",")]
            src0 = _vregs(parts[1].split(" ")[0]) if len(parts) > 1 else set()
            waited = 0
            for pm, pops in reversed(insts):
                if waited >= 5:
                    break
                first = pops.split(",")[0].strip()
                writes_exec = pm.startswith("v_cmpx") or (pm.startswith("v_") and first.startswith("exec"))
                if writes_exec:
                    bad.append(f"{func}: {pm} {pops}  ->  {body}  (EXEC written {waited} wait states ahead)")
                if waited < 2 and pm.startswith("v_") and not pm.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                    if _vregs(first) & src0:
                        bad.append(f"{func}: {pm} {pops}  ->  {body}  (DPP source written {waited} wait states ahead)")
                waited += (int(pops.strip() or 0) + 1) if pm == "s_nop" else 1
        insts.append((mn, ops))
        if len(insts) > 16:
            del insts[0]
    return n_dpp, bad


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
@pytest.mark.parametrize("unit", ["pdegym_1d", "pdegym_1d_rollout", "pdegym_ns2d", "pdegym_ns256", "pdegym_ns256_f64", "pdegym_traffic"])
def test_every_dpp_instruction_has_its_wait_states(unit, tmp_path):
    from pdecontrolgym_amd import build
    build.build()
    n_dpp, bad = _check(_disassemble(os.path.join(build.LIBDIR, unit + ".o"), str(tmp_path)))
    assert n_dpp > 10, f"{unit}: only {n_dpp} DPP instructions found -- did the disassembly format change?"
    assert not bad, "\n".join(bad[:20])


def test_the_checker_sees_a_planted_hazard():
    text = ("0000 <k>:\n\tv_cmpx_eq_u32_e32 1, v3  // 0\n\tv_mov_b32_e32 v9, v1  // 0\n\ts_nop 1  // 0\n"
            "\tv_add_f32_dpp v1, v2, v2 wave_shr:1 row_mask:0xf bank_mask:0xf  // 0\n"
            "\tv_mul_f32_e32 v5, v1, v1  // 0\n\tv_mov_b32_dpp v7, v5 row_shr:1 row_mask:0xf bank_mask:0xf  // 0\n"
            "\tv_mul_f32_e32 v8, v1, v1  // 0\n\ts_nop 1  // 0\n\tv_mov_b32_dpp v7, v8 row_shr:1 row_mask:0xf bank_mask:0xf  // 0\n")
    n, bad = _check(text)
    assert n == 3 and len(bad) == 2 and "EXEC written 3" in bad[0] and "DPP source written 0" in bad[1]
