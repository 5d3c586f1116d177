"""Static check of the BUILT gfx950 code for a hazard hipcc cannot see inside inline asm (round 4).

A VALU instruction that reads the result of a transcendental instruction (v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos)
needs one wait state between them on gfx940-class parts.  hipcc's hazard recogniser inserts it for the instructions IT emits, not for
instructions that arrive as inline-asm text: vit_attention.hip's soft-max (an asm v_add_f32 behind the compiler's v_exp_f32) produced
run-to-run differences the moment the scheduler put such a pair back to back.  So: disassemble every kernel of the shipped library and
require that no instruction reads a register that the instruction directly before it wrote with a transcendental opcode.  Runs on the
CPU (llvm-objdump); nothing is executed.

The same holds for MFMA results: a VALU / memory instruction that reads (or overwrites) a matrix instruction's destination needs
passes + 2 wait states behind a plain fp32 MFMA and passes + 4 behind an XDL one on gfx950 (LLVM's GCNHazardRecogniser:
checkMAIVALUHazards), and the recogniser does not look inside inline asm either -- the soft-max and epilogue asm statements of
vit_attention.hip / vit_gemm*.hip consume accumulators.  Second check below: no such instruction inside the window (straight-line
code only: the scan forgets what is pending at an unconditional branch)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "patchioner_amd", "libpatchioner_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f32|f16|legacy_f32)")


def _regs(tok):
    """vN or v[a:b] -> set of VGPR numbers (the token may carry modifiers: -v3, |v3|, v3.l ...)"""
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]|(?<![a-z_\d])v(\d+)", tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def _regs_va(tok):
    """like _regs for both register files: {('v', n), ('a', n)}"""
    out = set()
    for m in re.finditer(r"([va])\[(\d+):(\d+)\]|(?<![a-z_\d])([va])(\d+)", tok):
        if m.group(1) is not None:
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def _mfma_window(op):
    """wait states an MFMA's destination must not be touched for (gfx950; passes of 4 cycles)"""
    if "32x32x16" in op or "16x16x16" in op:
        return 8 + 4          # XDL, 8 passes
    if "16x16x32" in op:
        return 4 + 4          # XDL, 4 passes
    if "16x16x4_f32" in op or "16x16x4f32" in op:
        return 8 + 2          # fp32: not XDL
    if "32x32x2" in op:
        return 16 + 2
    if "4x4" in op:
        return 2 + 4
    return 16 + 4             # anything else: the longest


def scan_mfma(path, tmp_path):
    """-> (violations, MFMA instructions seen)"""
    lib = os.path.join(tmp_path, "lib.so")
    shutil.copy(path, lib)
    subprocess.run([OBJDUMP, "--offloading", lib], check=True, capture_output=True, cwd=tmp_path)
    objs = [os.path.join(tmp_path, f) for f in os.listdir(tmp_path) if "gfx950" in f]
    bad, seen = [], 0
    for o in objs:
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", o], check=True, capture_output=True, text=True).stdout
        func, pend = None, []                                      # pend: (destination registers, wait states still needed, text)
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                func, pend = m.group(1), []
                continue
            ins = line.strip().split("//")[0].strip()
            if not ins or ins.endswith(":"):
                continue
            parts = ins.split(None, 1)
            op, args = parts[0], (parts[1] if len(parts) > 1 else "")
            toks = [t.strip() for t in args.split(",")]
            if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                pend = []
                continue
            mfma = op.startswith(("v_mfma", "v_smfmac"))
            if not mfma and op.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
                touched = set()
                for t in toks:                                     # sources and destination alike: RAW and WAW
                    touched |= _regs_va(t)
                for dst, left, text in pend:
                    if touched & dst:
                        bad.append("%s: %s -> (%d wait states short) %s" % (func, text, left, ins))
            step = int(toks[0], 0) + 1 if op == "s_nop" else 1
            pend = [(d, left - step, t) for d, left, t in pend if left - step > 0]
            if mfma:
                seen += 1
                pend.append((_regs_va(toks[0]), _mfma_window(op), ins))
    return bad, seen


def scan(path, tmp_path):
    """-> (violations, kernels seen, transcendental instructions seen) of the gfx950 code objects bundled in ``path``"""
    lib = os.path.join(tmp_path, "lib.so")
    shutil.copy(path, lib)
    subprocess.run([OBJDUMP, "--offloading", lib], check=True, capture_output=True, cwd=tmp_path)
    objs = [os.path.join(tmp_path, f) for f in os.listdir(tmp_path) if "gfx950" in f]
    assert objs, "no gfx950 code object in the library"
    bad, kernels, trans = [], 0, 0
    for o in objs:
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", o], check=True, capture_output=True, text=True).stdout
        func, prev = None, None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                func, prev = m.group(1), None
                kernels += 1
                continue
            ins = line.strip().split("//")[0].strip()
            if not ins or ins.endswith(":"):
                continue
            parts = ins.split(None, 1)
            op, args = parts[0], (parts[1] if len(parts) > 1 else "")
            toks = [t.strip() for t in args.split(",")]
            if prev is not None:
                dst, pop = prev
                srcs = toks[1:] if op.startswith("v_") and not op.startswith("v_cmp") else toks   # VOPC writes vcc / an SGPR: every token is a source
                if op.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_")) and any(_regs(t) & dst for t in srcs):
                    bad.append("%s: %s -> %s" % (func, pop, ins))
            prev = None
            if TRANS.match(op) and toks:
                trans += 1
                prev = (_regs(toks[0]), ins)
    return bad, kernels, trans


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_no_transcendental_result_is_read_by_the_next_instruction(tmp_path):
    bad, kernels, trans = scan(LIB, tmp_path)
    assert kernels > 20 and trans > 100, (kernels, trans)          # the scan saw the library's kernels and their transcendentals
    assert not bad, "a transcendental's result is read one instruction later (missing wait state):\n" + "\n".join(bad[:20])


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_no_mfma_destination_is_touched_inside_its_hazard_window(tmp_path):
    bad, seen = scan_mfma(LIB, tmp_path)
    assert seen > 5000, seen
    assert not bad, "an MFMA's destination is read or overwritten before its wait states have passed:\n" + "\n".join(bad[:20])


def _sregs(tok):
    """sN or s[a:b] -> set of SGPR numbers"""
    out = set()
    for m in re.finditer(r"s\[(\d+):(\d+)\]|(?<![a-z_\d])s(\d+)(?![\d:])", tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scan_valu_sgpr_vmem(path, tmp_path):
    """Third hazard hipcc cannot see inside inline asm (round 5: a memory fault): a VMEM instruction that reads an SGPR (address base,
    resource, scalar offset) which a VALU instruction wrote (v_readlane / v_readfirstlane: SGPR spill reloads; VOP3 compares) needs
    FIVE wait states in between (GCNHazardRecogniser: checkVMEMHazards, VmemSgprWaitStates).  An asm `global_load ..., s[a:b]` whose
    pointer had just been reloaded by v_readlane read the stale high half.  -> (violations, VALU SGPR writes seen)"""
    lib = os.path.join(tmp_path, "lib3.so")
    shutil.copy(path, lib)
    subprocess.run([OBJDUMP, "--offloading", lib], check=True, capture_output=True, cwd=tmp_path)
    objs = [os.path.join(tmp_path, f) for f in os.listdir(tmp_path) if "gfx950" in f]
    bad, seen = [], 0
    for o in objs:
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", o], check=True, capture_output=True, text=True).stdout
        func, pend = None, []                                      # pend: (SGPRs written, wait states still needed, text)
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                func, pend = m.group(1), []
                continue
            ins = line.strip().split("//")[0].strip()
            if not ins or ins.endswith(":"):
                continue
            parts = ins.split(None, 1)
            op, args = parts[0], (parts[1] if len(parts) > 1 else "")
            toks = [t.strip() for t in args.split(",")]
            if op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                pend = []
                continue
            if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                read = set()
                for t in toks[1:] if not op.startswith(("buffer_store", "global_store", "scratch_store", "flat_store")) else toks:
                    read |= _sregs(t)
                for dst, left, text in pend:
                    if read & dst:
                        bad.append("%s: %s -> (%d wait states short) %s" % (func, text, left, ins))
            step = int(toks[0], 0) + 1 if op == "s_nop" else 1
            pend = [(d, left - step, t) for d, left, t in pend if left - step > 0]
            if op.startswith(("v_readlane_b32", "v_readfirstlane_b32")) or (op.startswith("v_cmp") and op.endswith("_e64")):
                d = _sregs(toks[0])
                if d:
                    seen += 1
                    pend.append((d, 5, ins))
    return bad, seen


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_no_vmem_instruction_reads_an_sgpr_a_valu_has_just_written(tmp_path):
    bad, seen = scan_valu_sgpr_vmem(LIB, tmp_path)
    assert seen > 500, seen
    assert not bad, "a VMEM instruction reads an SGPR written by a VALU fewer than five wait states earlier:\n" + "\n".join(bad[:20])
