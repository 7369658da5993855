"""Hazard audit of every built gfx950 code object, from the BINARY (CPU suite; no GPU needed).

CDNA3/4 ISA: a VMEM store of more than 64 bits reads its data VGPRs late -- an instruction that writes those
registers needs two wait states behind the store.  hipcc pads that for stores it emits itself, but an
`asm volatile("global_store_dwordx4 ...")` is opaque to its hazard recognizer
(/opt/skills/guides/cdna_hip_programming.md section 5.7 item 1).  The engine's write-through stores
(`csrc/nasr_wave.h`: `store_wt_f4`, `store_wt_u4`, "sc0 sc1") are such asm; they carry the residual stream of
the reference's src/nemo-stream.cpp:631-634 / :682-685, so a corrupted lane there is a silently wrong transcript.

Checked here, on a copy of each library outside the tree:
 1. every `*_store_dwordx3/x4 ... sc0 sc1` (only inline asm produces that modifier pair in this code base) is
    IMMEDIATELY followed by `s_nop N`, N >= 1 -- the pad inside the asm string;
 2. for EVERY dwordx3/x4 store, whoever emitted it: no instruction within the next two wait states writes a
    register of the store's data operand.
"""
from __future__ import annotations

import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
LLVM = Path("/opt/rocm/lib/llvm/bin")
LIBS = [
    ROOT / "nemotron-asr.cpp_amd" / "libnemotron_asr_amd.so",
    ROOT / "nemotron-asr.cpp_amd" / "libnemotron_asr_amd_stamps.so",     # diagnostic build (make stamps); audited when present
    ROOT / "tests" / "helpers" / "liblds_poison.so",
]

INSN = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//")
WIDE_STORE = re.compile(r"^(global|buffer|flat|scratch)_store_dwordx[34]$")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(operand: str) -> set[tuple[str, int]]:
    out: set[tuple[str, int]] = set()
    for m in REG.finditer(operand):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def split_operands(ops: str) -> list[str]:
    parts, depth, cur = [], 0, ""
    for ch in ops:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur.strip())
    return parts


def data_operand(mnemonic: str, ops: list[str]) -> str:
    # global/flat/scratch: addr, data, ...   buffer: data, addr|off, rsrc, ...
    return ops[0] if mnemonic.startswith("buffer") else ops[1]


def written_regs(mnemonic: str, ops: list[str]) -> set[tuple[str, int]]:
    """VGPR / AGPR registers a VALU instruction writes (its first operand).  The hazard is about VALU writes in the next two issue slots;
    a load that targets the same registers (ds_read, global_load) returns its data tens of cycles after the store has read its own
    (hipcc itself places `ds_read_b128 v[130:133]` right behind `global_store_dwordx4 ..., v[130:133]`)."""
    if not ops or not mnemonic.startswith("v_"):
        return set()
    if mnemonic.startswith(("v_cmp", "v_nop", "v_readlane", "v_readfirstlane")):
        return set()
    w = regs_of(ops[0])
    if mnemonic.startswith(("v_swap", "v_permlane")) and len(ops) > 1:
        w |= regs_of(ops[1])
    return w


def wait_states(mnemonic: str, ops: list[str]) -> int:
    if mnemonic == "s_nop" and ops:
        return int(ops[0], 0) + 1
    return 1


def disassemble(lib: Path, tmp: Path) -> list[tuple[str, list[tuple[str, list[str]]]]]:
    """-> [(code object name, [(mnemonic, operands)...])] of every gfx950 code object bundled into lib"""
    work = tmp / lib.stem
    work.mkdir()
    shutil.copy(lib, work / lib.name)                    # llvm-objdump --offloading writes beside its input: keep that out of the tree
    subprocess.check_call([str(LLVM / "llvm-objdump"), "--offloading", lib.name], cwd=work, stdout=subprocess.DEVNULL)
    out = []
    for co in sorted(work.glob("*.hipv4-amdgcn-amd-amdhsa--gfx950")):
        text = subprocess.check_output([str(LLVM / "llvm-objdump"), "-d", co.name], cwd=work, text=True)
        insns = []
        for line in text.splitlines():
            m = INSN.match(line)
            if m:
                insns.append((m.group(1), split_operands(m.group(2))))
            elif line.endswith(":") and not line.startswith(" "):
                insns.append(("<label>", []))            # a branch target: a window never extends across it backwards, and is checked from the store on
        out.append((co.name, insns))
    return out


def audit(insns: list[tuple[str, list[str]]]) -> tuple[int, int, list[str]]:
    n_wide = n_asm = 0
    bad: list[str] = []
    for i, (mn, ops) in enumerate(insns):
        if not WIDE_STORE.match(mn):
            continue
        n_wide += 1
        tail = " ".join(ops)
        data = regs_of(data_operand(mn, ops))
        from_asm = "sc0 sc1" in tail
        if from_asm:
            n_asm += 1
            nxt = insns[i + 1] if i + 1 < len(insns) else ("<end>", [])
            if not (nxt[0] == "s_nop" and int(nxt[1][0], 0) >= 1):
                bad.append(f"#{i} {mn} {tail}: asm write-through store not followed by s_nop >= 1 (next: {nxt[0]} {' '.join(nxt[1])})")
        served, j = 0, i + 1
        while served < 2 and j < len(insns):
            m2, o2 = insns[j]
            if m2 == "<label>":
                j += 1
                continue
            if m2 in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                break                                    # the listing's next line is not what runs next; a taken branch is worth more than two states
            hit = written_regs(m2, o2) & data
            if hit:
                bad.append(f"#{i} {mn} {tail}: data registers {sorted(hit)} written {served} wait states later by {m2} {', '.join(o2)}")
                break
            served += wait_states(m2, o2)
            j += 1
    return n_wide, n_asm, bad


def test_parser_sees_the_hazard():
    """the audit itself: a synthetic sequence with the hazard is flagged, the padded one is not"""
    st = ("global_store_dwordx4", ["v[0:1]", "v[4:7]", "off sc0 sc1"])
    clobber = ("v_lshl_or_b32", ["v5", "v9", "16", "v8"])
    _, n_asm, bad = audit([st, clobber])
    assert n_asm == 1 and len(bad) == 2                                            # no pad, and the data register is rewritten
    _, _, bad = audit([st, ("s_nop", ["1"]), clobber])
    assert bad == []
    _, _, bad = audit([("global_store_dwordx4", ["v[0:1]", "v[4:7]", "off"]), ("s_nop", ["0"]), clobber])
    assert len(bad) == 1                                                           # one state is not two
    _, _, bad = audit([("global_store_dwordx4", ["v[0:1]", "v[4:7]", "off"]), ("s_branch", ["65197"]), clobber])
    assert bad == []                                                               # linear listing behind an unconditional branch is not the successor
    _, _, bad = audit([("global_store_dwordx4", ["v[0:1]", "v[4:7]", "off"]), ("ds_read_b128", ["v[4:7]", "v9"])])
    assert bad == []                                                               # a load into the data registers lands long after the store has read them
    _, _, bad = audit([("buffer_store_dwordx3", ["v[4:6]", "v1", "s[0:3]", "0 offen"]), ("v_mov_b32_e32", ["v7", "v1"]), ("v_mov_b32_e32", ["v8", "v1"]), ("v_mov_b32_e32", ["v4", "v1"])])
    assert bad == []                                                               # third state: safe


@pytest.mark.skipif(not (LLVM / "llvm-objdump").exists(), reason="llvm-objdump not in this image")
@pytest.mark.parametrize("lib", LIBS, ids=lambda p: p.name)
def test_wide_stores_are_hazard_safe(lib: Path, tmp_path: Path):
    if not lib.exists():
        if lib.name == "libnemotron_asr_amd.so":
            pytest.fail(f"{lib} not built: run __graft_entry__.build()")
        pytest.skip(f"{lib.name} not built")
    objs = disassemble(lib, tmp_path)
    assert objs, f"no gfx950 code object in {lib.name}"
    total_wide = total_asm = 0
    problems: list[str] = []
    for name, insns in objs:
        n_wide, n_asm, bad = audit(insns)
        total_wide += n_wide
        total_asm += n_asm
        problems += [f"{name}: {b}" for b in bad]
    assert not problems, "\n".join(problems[:40])
    if lib.name.startswith("libnemotron_asr_amd"):
        assert total_asm >= 100, f"expected the engine's write-through stores in {lib.name}, found {total_asm}"    # the audit is looking at the right thing
        assert total_wide > total_asm
