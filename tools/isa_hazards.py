#!/usr/bin/env python3
"""MFMA dependency lint over the SHIPPED gfx950 code: disassembles every kernel of a built libidee_hip.so (the code objects of its
.hip_fatbin section, `llvm-objdump -d`) and checks every instruction that follows an MFMA within its dependency window against
the wait-state table below.

Why: DESIGN.md §3e — a three-link P.V MFMA chain (C != D in every link) once returned wrong tiles until 32 wait states were put
behind it.  The table is the union of (i) what LLVM's GCNHazardRecognizer pads for gfx950 (checkMAIHazards90A /
checkMAIVALUHazards, restated from the upstream source) and (ii) what the HARDWARE needs as measured by
tools/micro/mfma_hazard.hip on an MI355X (`MEASURED`) — the larger of the two per rule.  A rule the compiler does
not know (MEASURED > LLVM) is exactly the kind of gap that chain fell into; the lint proves that no shipped kernel sits in it.

"Wait states" are counted as LLVM counts them: one per instruction issued between the two, `s_nop k` = k + 1.  The walk is a
forward data-flow over basic blocks (in-flight MFMAs of the last 24 states, minimum age over all predecessors), so loops and
branches are covered.

    python tools/isa_hazards.py [path/to/libidee_hip.so] [--stats]      exit status 1 on a violation
"""
import os
import re
import struct
import subprocess
import sys
import tempfile
from collections import defaultdict

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
MAXAGE = 24

# The MFMAs this library issues.  Passes = issue-to-issue cycles / 4 on gfx950 (tools/micro/mfma_k16.hip, MI355X_MICROARCH.md):
# v_mfma_f32_16x16x32_{f16,bf16} 16 cycles = 4 passes, XDL; v_mfma_f32_16x16x4_f32 32 cycles = 8 passes, "SMFMA" (f32 inputs,
# not XDL).  Any other MFMA in a listing is an error of the lint's coverage, reported as such.
MFMA = {
    "v_mfma_f32_16x16x32_f16": ("xdl", 4),
    "v_mfma_f32_16x16x32_bf16": ("xdl", 4),
    "v_mfma_f32_16x16x4_f32": ("smfma", 8),
}
XDL_P, SM_P = 4, 8

# rule -> {kind of the FIRST (in-flight) MFMA: wait states LLVM's GCNHazardRecognizer inserts for gfx950} (the GFX940_*WaitStates
# helpers with IsGFX950 = true; XDL: NumPasses = 4, SMFMA: NumPasses = 8; restated from the upstream source).  The compiler
# applies them to instructions it models; an instruction inside an `asm` statement is NOT modelled (MI355X guide, §5.7 item 2),
# so a pair below these numbers marks hand-written asm (or a rule this restatement has wrong) — reported, not fatal.
LLVM = {
    "raw_read":       {"xdl": XDL_P + 3 + 1, "smfma": SM_P + 2},   # VALU / memory / LDS / export reads the MFMA's vDst
    "waw_valu":       {"xdl": XDL_P + 3 + 1, "smfma": SM_P + 2},   # VALU writes the MFMA's vDst
    "war_srcc":       {"xdl": 3, "smfma": 0},                      # VALU writes the MFMA's SrcC (C != D): XDL 4-pass 3; SMFMA exempt on gfx940+
    "war_srcab":      {"xdl": 0, "smfma": 0},                      # VALU writes the MFMA's SrcA / SrcB (no LLVM rule)
    "mfma_srcc_same": {"xdl": 0, "smfma": 0},                      # next MFMA's SrcC is EXACTLY the first's vDst (hardware interlock)
    "mfma_srcc_part": {"xdl": XDL_P + 1 + 1, "smfma": SM_P},       # next MFMA's SrcC overlaps the first's vDst partially
    "mfma_srcab":     {"xdl": XDL_P + 3 + 1, "smfma": SM_P + 2},   # next MFMA reads the first's vDst as SrcA / SrcB
    "mfma_waw":       {"xdl": 0, "smfma": 0},                      # next MFMA writes (without reading) the first's vDst (no LLVM rule: in-order pipe)
    "valu_to_mfma":   {"xdl": 0, "smfma": 0},                      # VALU write -> MFMA operand read (key: the READING MFMA's kind; no LLVM rule)
}
# What the HARDWARE needs: tools/micro/mfma_hazard.hip on an MI355X (ROCm 7.2; profiles/r05_mfma_hazard.txt), minimum SAFE wait
# states, one wave per SIMD.  Key: rule -> {(kind of the first MFMA, kind of the second instruction): N}; "v" = a VALU / memory
# instruction, "x" = xdl, "s" = smfma.  THIS is the table a violation is judged against.
#   * the f32 MFMA (smfma) is fully interlocked against VALU reads / writes and against MFMA operand reads — only a PARTIAL SrcC
#     overlap with another f32 MFMA's vDst is not (9; LLVM pads 8);
#   * the f16 / bf16 MFMA (xdl, 4 passes) needs 7 before any reader of its vDst (VALU, memory, SrcA / SrcB of either MFMA kind, a
#     partially overlapping SrcC of another xdl MFMA; LLVM pads 8 / 8 / 6 — the last is one short), 4 before a VALU write of its
#     vDst (LLVM 8), none before a write of its SrcA / SrcB / SrcC; an exactly matching SrcC (C != D or C == D) is interlocked;
#     its own operands need 1 wait state after a VALU write;
#   * vDst may sit in the instruction's own SrcA / SrcB registers (d_on_a / d_on_b).
HW = {
    "raw_read":       {("xdl", "v"): 7, ("smfma", "v"): 0},
    "waw_valu":       {("xdl", "v"): 4, ("smfma", "v"): 0},
    "war_srcc":       {("xdl", "v"): 0, ("smfma", "v"): 0},
    "war_srcab":      {("xdl", "v"): 0, ("smfma", "v"): 0},
    "mfma_srcc_same": {("xdl", "xdl"): 0, ("xdl", "smfma"): 0, ("smfma", "xdl"): 0, ("smfma", "smfma"): 0},
    "mfma_srcc_part": {("xdl", "xdl"): 7, ("xdl", "smfma"): 0, ("smfma", "xdl"): 0, ("smfma", "smfma"): 9},
    "mfma_srcab":     {("xdl", "xdl"): 7, ("xdl", "smfma"): 7, ("smfma", "xdl"): 0, ("smfma", "smfma"): 0},
    "mfma_waw":       {("xdl", "xdl"): 0, ("xdl", "smfma"): 0, ("smfma", "xdl"): 0, ("smfma", "smfma"): 0},
    "valu_to_mfma":   {("v", "xdl"): 1, ("v", "smfma"): 0},
}
SAME_MFMA_DST_ON_SRCAB_OK = True     # d_on_a / d_on_b of the probe: vDst in the registers of the instruction's own SrcA / SrcB


def need(rule, first, second):
    return HW[rule][(first, second)]


# ---------------------------------------------------------------------------------------------------------------------------
# extraction + parsing
# ---------------------------------------------------------------------------------------------------------------------------
def code_objects(so_path, outdir):
    """The gfx950 ELF images inside the .hip_fatbin section of a host shared object (one clang offload bundle per source file)."""
    fat = os.path.join(outdir, "fat.bin")
    subprocess.run([f"{LLVM_BIN}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", so_path], check=True)
    data = open(fat, "rb").read()
    magic, out, i = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:
        o = data.find(magic, i)
        if o < 0:
            break
        i = o + 1
        n, = struct.unpack_from("<Q", data, o + 24)
        p = o + 32
        for _ in range(n):
            eo, es, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and es:
                path = os.path.join(outdir, f"co_{len(out)}.elf")
                open(path, "wb").write(data[o + eo:o + eo + es])
                out.append(path)
    return out


def disassemble(elf):
    return subprocess.run([f"{LLVM_BIN}/llvm-objdump", "-d", "--mcpu=gfx950", elf], check=True, capture_output=True, text=True).stdout


_REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])(?![\w\[])")
_LINE = re.compile(r"^\s+(\S+)(?:\s+(.*?))?\s*// ([0-9A-F]{12}):")
_FUNC = re.compile(r"^[0-9a-f]{16} <(.+)>:$")


def regs(operand):
    """'v[4:7]' -> {('v',4),..}; 'a3' -> {('a',3)}; anything else -> empty."""
    s = set()
    for m in _REG.finditer(operand):
        if m.group(2) is not None:
            s.add((m.group(1), int(m.group(2))))
        else:
            for r in range(int(m.group(3)), int(m.group(4)) + 1):
                s.add((m.group(1), r))
    return s


def split_operands(text):
    out, depth, cur = [], 0, ""
    for ch in text or "":
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


class Ins:
    __slots__ = ("addr", "mn", "ops", "wr", "rd", "mfma", "A", "B", "C", "D", "ws", "branch", "target", "ends", "valu")

    def __init__(self, addr, mn, ops):
        self.addr, self.mn, self.ops = addr, mn, ops
        self.mfma = None
        self.A = self.B = self.C = self.D = frozenset()
        self.ws = 1
        self.branch, self.target, self.ends = False, None, False
        self.valu = False
        self.wr, self.rd = set(), set()
        base = mn[:-4] if mn.endswith(("_e32", "_e64")) else mn
        base = re.sub(r"_(sdwa|dpp)$", "", base)
        if base.startswith("v_mfma") or base.startswith("v_smfmac"):
            self.mfma = base
            self.D, self.A, self.B, self.C = (frozenset(regs(o)) for o in (ops + ["", "", "", ""])[:4])
            self.wr, self.rd = set(self.D), set(self.A | self.B | self.C)
            return
        if base == "s_nop":
            self.ws = int(ops[0], 0) + 1
            return
        if base.startswith("s_cbranch") or base == "s_branch":
            self.branch = True
            self.ends = base == "s_branch"
            return
        if base in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            self.ends = True
            return
        allr = [regs(o.split(" ")[0]) if not o.startswith(("neg", "op_sel", "offset", "clamp")) else set() for o in ops]
        if base.startswith("v_"):
            self.valu = True
            if base in ("v_swap_b32", "v_permlane16_swap_b32", "v_permlane32_swap_b32"):
                self.wr = allr[0] | allr[1]
                self.rd = allr[0] | allr[1]
                return
            if base.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane")):
                self.rd = set().union(*allr) if allr else set()
                return
            self.wr = set(allr[0]) if allr else set()
            self.rd = set().union(*allr[1:]) if len(allr) > 1 else set()
            if re.search(r"mac|dot\dc|fmaak|accvgpr_mov", base) or base.startswith(("v_cndmask", "v_writelane", "v_mov_b32_dpp", "v_movrel")) or "dpp" in mn or "sdwa" in mn:
                self.rd |= self.wr          # (dpp / sdwa with bound_ctrl / dst_unused keep or read the old destination)
            return
        if base.startswith(("global_load_lds", "buffer_load") ) and base.startswith("global_load_lds"):
            self.rd = set().union(*allr) if allr else set()
            return
        if re.match(r"(global|flat|scratch|buffer)_load|ds_read|ds_bpermute|ds_permute|ds_swizzle|ds_consume|ds_append|s_buffer_load|s_load", base):
            if " lds" in " ".join(ops):
                self.rd = set().union(*allr) if allr else set()
                return
            self.wr = set(allr[0]) if allr else set()
            self.rd = set().union(*allr[1:]) if len(allr) > 1 else set()
            return
        if re.match(r"(global|flat|buffer)_atomic|ds_.*rtn", base):
            self.rd = set().union(*allr) if allr else set()
            if "sc0" in " ".join(ops) or "glc" in " ".join(ops) or "rtn" in base:
                self.wr = set(allr[0]) if allr else set()
            return
        self.rd = set().union(*allr) if allr else set()      # stores, ds_write, exports, everything else: readers only


def parse(text):
    """-> {function name: [Ins]}"""
    funcs, cur = {}, None
    for line in text.splitlines():
        m = _FUNC.match(line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        m = _LINE.match(line)
        if not m:
            continue
        mn, optext, addr = m.group(1), m.group(2) or "", int(m.group(3), 16)
        ins = Ins(addr, mn, split_operands(optext))
        if ins.branch:
            try:
                simm = int(ins.ops[0], 0)
                if simm >= 32768:
                    simm -= 65536
                ins.target = addr + 4 + 4 * simm
            except (ValueError, IndexError):
                ins.target = None
        cur.append(ins)
    return funcs


# ---------------------------------------------------------------------------------------------------------------------------
# analysis
# ---------------------------------------------------------------------------------------------------------------------------
class Report:
    def __init__(self):
        self.violations = []
        self.min_slack = defaultdict(lambda: (10 ** 9, None))       # (rule, kinds) -> (observed wait states - needed, where)
        self.counts = defaultdict(int)
        self.unpadded = defaultdict(int)                            # rule -> pairs below LLVM's own padding (asm statements)
        self.unknown_mfma = set()

    def observe(self, rule, k1, k2, ws, fn, first, second):
        n = need(rule, k1, k2)
        key = f"{rule}[{k1}->{k2}]"
        self.counts[key] += 1
        slack = ws - n
        if slack < self.min_slack[key][0]:
            self.min_slack[key] = (slack, f"{short(fn)} @{second.addr:X}: {first.mn} @{first.addr:X} -> {second.mn} after {ws} wait states (hardware needs {n})")
        llvm = LLVM[rule][k2 if rule == "valu_to_mfma" else k1]
        if ws < llvm:
            self.unpadded[key] += 1
        if ws < n:
            self.violations.append(f"{key}: {short(fn)}: {first.mn} {', '.join(first.ops)} @{first.addr:X} -> {second.mn} {', '.join(second.ops)} @{second.addr:X} "
                                   f"after {ws} wait states, the hardware needs {n}")


def short(fn):
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", fn)
    if m:
        n = int(m.group(1))
        return fn[m.end():m.end() + n] + "<" + fn[m.end() + n:m.end() + n + 40] + ">"
    return fn[:80]


def check_function(fn, code, rep):
    if not any(i.mfma for i in code):
        return
    idx_of = {i.addr: k for k, i in enumerate(code)}
    # basic blocks
    leaders = {0}
    for k, i in enumerate(code):
        if i.branch or i.ends:
            if k + 1 < len(code):
                leaders.add(k + 1)
            if i.target in idx_of:
                leaders.add(idx_of[i.target])
    starts = sorted(leaders)
    block_of = {}
    for b, s in enumerate(starts):
        e = starts[b + 1] if b + 1 < len(starts) else len(code)
        for k in range(s, e):
            block_of[k] = b
    succs = defaultdict(set)
    for b, s in enumerate(starts):
        e = (starts[b + 1] if b + 1 < len(starts) else len(code)) - 1
        last = code[e]
        if last.branch and last.target in idx_of:
            succs[b].add(block_of[idx_of[last.target]])
        if not last.ends and e + 1 < len(code):
            succs[b].add(b + 1)
    # state: {index of an in-flight MFMA: (age, frozenset of its vDst registers NOT rewritten since)} — a later full or partial
    # rewrite of a vDst register (by an MFMA or a VALU) shadows it for READ-type rules: whoever reads that register afterwards
    # depends on the later writer, and the pipe completes MFMAs in order (waw_* rows of the probe)
    entry = {b: None for b in range(len(starts))}
    entry[0] = ({}, {})
    work = [0]
    seen = set()
    while work:
        b = work.pop()
        state = dict(entry[b][0])
        vw = dict(entry[b][1])                 # recent VALU writes: register -> age
        s = starts[b]
        e = starts[b + 1] if b + 1 < len(starts) else len(code)
        for k in range(s, e):
            ins = code[k]
            k2 = MFMA.get(ins.mfma, ("xdl",))[0] if ins.mfma else "v"
            for mk, (age, live) in state.items():
                first = code[mk]
                info = MFMA.get(first.mfma)
                if info is None:
                    continue
                k1 = info[0]
                new_pair = (mk, k) not in seen
                seen.add((mk, k))
                if ins.mfma:
                    if ins.C and ins.C & live:
                        if ins.C == first.D and live == first.D:
                            rep.observe("mfma_srcc_same", k1, k2, age, fn, first, ins)
                            if ins.D != ins.C and new_pair:
                                rep.counts["chain links with C != D"] += 1
                        else:
                            rep.observe("mfma_srcc_part", k1, k2, age, fn, first, ins)
                    if (ins.A | ins.B) & live:
                        rep.observe("mfma_srcab", k1, k2, age, fn, first, ins)
                    if ins.D & first.D and not (ins.C & first.D):
                        rep.observe("mfma_waw", k1, k2, age, fn, first, ins)
                else:
                    if ins.rd & live:
                        rep.observe("raw_read", k1, "v", age, fn, first, ins)
                    if ins.valu and ins.wr:
                        if ins.wr & first.D:
                            rep.observe("waw_valu", k1, "v", age, fn, first, ins)
                        if first.C != first.D and ins.wr & first.C:
                            rep.observe("war_srcc", k1, "v", age, fn, first, ins)
                        if ins.wr & (first.A | first.B):
                            rep.observe("war_srcab", k1, "v", age, fn, first, ins)
            if ins.mfma:
                if ins.mfma not in MFMA:
                    rep.unknown_mfma.add(ins.mfma)
                for r in ins.rd:
                    if r in vw:
                        n = need("valu_to_mfma", "v", k2)
                        key = f"valu_to_mfma[v->{k2}]"
                        rep.counts[key] += 1
                        if vw[r] - n < rep.min_slack[key][0]:
                            rep.min_slack[key] = (vw[r] - n, f"{short(fn)} @{ins.addr:X}: VALU write of {r[0]}{r[1]} -> {ins.mn} after {vw[r]} wait states (hardware needs {n})")
                        if vw[r] < n:
                            rep.violations.append(f"{key}: {short(fn)}: VALU write of {r[0]}{r[1]} -> {ins.mn} {', '.join(ins.ops)} @{ins.addr:X} after {vw[r]} wait states, the hardware needs {n}")
                if (k, k) not in seen:
                    seen.add((k, k))
                    if ins.D & (ins.A | ins.B):
                        rep.counts["vDst on its own SrcA / SrcB"] += 1
                        if not SAME_MFMA_DST_ON_SRCAB_OK:
                            rep.violations.append(f"dst_on_srcab: {short(fn)}: {ins.mn} {', '.join(ins.ops)} @{ins.addr:X}")
                    if ins.C and ins.C != ins.D:
                        rep.counts["MFMAs with C != D (register SrcC)"] += 1
            # this instruction's writes shadow older in-flight results; then age everything by its wait states
            nstate = {}
            for mk, (age, live) in state.items():
                if age + ins.ws <= MAXAGE:
                    nstate[mk] = (age + ins.ws, live - ins.wr if (ins.wr and (ins.valu or ins.mfma)) else live)
            state = nstate
            vw = {r: a + ins.ws for r, a in vw.items() if a + ins.ws <= 2}
            if ins.valu:
                for r in ins.wr:
                    vw[r] = 0
            if ins.mfma:
                state[k] = (0, frozenset(ins.D))
        for nb in succs[b]:
            old = entry[nb]
            if old is None:
                entry[nb] = (dict(state), dict(vw))
                work.append(nb)
            else:
                ms, mv, changed = dict(old[0]), dict(old[1]), False
                for mk, (a, live) in state.items():
                    if mk not in ms or a < ms[mk][0] or (live - ms[mk][1]):
                        ms[mk] = (min(a, ms[mk][0]) if mk in ms else a, (live | ms[mk][1]) if mk in ms else live)
                        changed = True
                for r, a in vw.items():
                    if r not in mv or a < mv[r]:
                        mv[r] = a
                        changed = True
                if changed:
                    entry[nb] = (ms, mv)
                    work.append(nb)


def check_library(so_path, rep=None):
    rep = rep or Report()
    nk = 0
    with tempfile.TemporaryDirectory() as td:
        for elf in code_objects(so_path, td):
            for fn, code in parse(disassemble(elf)).items():
                nk += 1
                check_function(fn, code, rep)
    return rep, nk


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    so = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "idee_amd", "libidee_hip.so")
    rep, nk = check_library(so)
    print(f"{so}: {nk} functions")
    for key in sorted(rep.min_slack):
        slack, where = rep.min_slack[key]
        print(f"  {key:30s} {rep.counts[key]:8d} pairs in the window, {rep.unpadded[key]:6d} below LLVM's padding, tightest slack {slack:+d}: {where}")
    for k in ("MFMAs with C != D (register SrcC)", "chain links with C != D", "vDst on its own SrcA / SrcB"):
        print(f"  {k}: {rep.counts[k]}")
    if rep.unknown_mfma:
        print("  MFMA instructions outside the table:", sorted(rep.unknown_mfma))
    for v in rep.violations[:50]:
        print("VIOLATION", v)
    print(f"{len(rep.violations)} violation(s)")
    return 1 if rep.violations or rep.unknown_mfma else 0


if __name__ == "__main__":
    sys.exit(main())
