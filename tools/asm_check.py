#!/usr/bin/env python3
"""Developer tool / CPU test helper: where do the ADDRESS registers of a kernel's global-memory instructions come from?

Round 5 found hipcc 7.2 building `switch (place)` in annp_fe_force_sh wrongly when `place` was a value it knew nothing about:
the fourth place's branch jumped past the address computation of its first four moment loads and issued them from a register
pair that held a double of some earlier computation (memory access fault; DESIGN.md 4.3b "places").  The workaround is one
`& 3` in the source, i.e. a dependency on what this compiler happens to do.  This file pins it from the assembly `make asm`
writes (no GPU needed): for every global load / store / atomic of a kernel, every definition that can reach one of its address
VGPRs along any path of the control-flow graph must be an instruction that can produce an address -- integer arithmetic, a
move, a select, a lane read -- and there must be one on every path.  A floating-point result, a value loaded from LDS or from
memory, or nothing at all in an address register is reported.

   python tools/asm_check.py [file.s] [kernel-name-substring ...]      (default: meng_zhang_amd/csrc/annp_hip.s, annp_fe_force_sh)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
GLOBAL = re.compile(r"^(global|flat)_(load|store|atomic)")
BRANCH = re.compile(r"^s_(cbranch_\w+|branch)\s+(\S+)")
# mnemonics whose VGPR result can be (part of) an address
INT_OK = re.compile(
    r"^v_(add|sub|subrev|addc|subb|subbrev|lshl|lshr|ashr|lshlrev|lshrrev|ashrrev|mad_u|mad_i|mul_lo|mul_hi|mul_u|mul_i|and|or|xor|not|"
    r"mov_b|cndmask|bfe|bfi|mbcnt|readlane|readfirstlane|min_[iu]|max_[iu]|med3_[iu]|lshl_or|and_or|or3|add3|lshl_add|add_lshl|xad|perm|"
    r"alignbit|alignbyte|accvgpr_read|swap|permlane|writelane|sad|bcnt|ffb|cvt_[iu]\d+_|cvt_pk_[iu]|dot)")
NO_DEST = re.compile(r"^(s_|v_cmp|v_cmpx|ds_write|ds_add_(?!rtn)|ds_(min|max|and|or|xor|inc|dec)_(?!rtn)|ds_nop|buffer_store|global_store|flat_store|scratch_store|"
                     r"global_load_lds|buffer_wbl2|buffer_inv|v_nop|v_readfirstlane|v_readlane|;)")


def vregs(operand):
    out = []
    for m in REG.finditer(operand):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_operands(rest):
    return [o.strip() for o in rest.split(",")] if rest else []


class Ins:
    __slots__ = ("idx", "line", "mn", "ops", "defs", "addr")

    def __init__(self, idx, line, text):
        self.idx, self.line = idx, line
        parts = text.split(None, 1)
        self.mn = parts[0]
        self.ops = split_operands(parts[1] if len(parts) > 1 else "")
        self.defs, self.addr = [], []
        mn = self.mn
        if GLOBAL.match(mn):
            kind = GLOBAL.match(mn).group(2)
            if mn.endswith("_lds") or "load_lds" in mn:                    # global_load_lds_dwordx4 vaddr, saddr|off
                self.addr = vregs(self.ops[0]) if self.ops else []
            elif kind == "load":                                          # vdst, vaddr, saddr|off
                self.defs = vregs(self.ops[0])
                self.addr = vregs(self.ops[1]) if len(self.ops) > 1 else []
            elif kind == "store":                                         # vaddr, vdata, saddr|off
                self.addr = vregs(self.ops[0]) if self.ops else []
            else:                                                         # atomic: [vdst,] vaddr, vdata, saddr|off  (vdst with sc0 / glc)
                returns = any(o.split()[-1] in ("sc0", "glc") or " sc0" in o or " glc" in o for o in self.ops)
                if returns:
                    self.defs = vregs(self.ops[0])
                    self.addr = vregs(self.ops[1]) if len(self.ops) > 1 else []
                else:
                    self.addr = vregs(self.ops[0]) if self.ops else []
        elif not NO_DEST.match(mn) and self.ops:
            self.defs = vregs(self.ops[0])
            if mn.startswith(("v_swap", "v_permlane16_swap", "v_permlane32_swap")) and len(self.ops) > 1:
                self.defs += vregs(self.ops[1])
            # carry-out forms: v_add_co_u32 v1, vcc, .. -- the first operand is still the VGPR

    def address_capable(self):
        return bool(INT_OK.match(self.mn))


def kernel_bodies(path, wanted):
    """{mangled name: [(line number, text)]} of the kernels whose demangled name contains one of `wanted`"""
    lines = open(path).read().splitlines()
    out, cur, name = {}, None, None
    for n, raw in enumerate(lines, 1):
        m = re.match(r"^(_Z\w+):\s*; @", raw)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is None:
            continue
        s = raw.strip()
        if s.startswith(".Lfunc_end"):
            out[name] = cur
            cur = None
            continue
        cur.append((n, raw))
    picked = {}
    for mangled, body in out.items():
        dem = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        if any(w in dem for w in wanted) and "s_endpgm" in "\n".join(t for _, t in body):
            picked[dem] = body
    return picked


def analyse(body):
    """-> list of findings (line, text, register, what) for one kernel body"""
    ins, label_at = [], {}
    for n, raw in body:
        s = raw.split(";")[0].strip() if not raw.strip().startswith(";") else ""
        if not s:
            continue
        if s.endswith(":"):
            label_at[s[:-1]] = len(ins)
            continue
        if s.startswith("."):
            continue
        ins.append(Ins(len(ins), n, s))
    # basic blocks: leaders at 0, label targets, instructions behind a branch
    leaders = {0} | set(label_at.values())
    for i, q in enumerate(ins):
        if BRANCH.match(q.mn + " " + ",".join(q.ops)) or q.mn in ("s_endpgm", "s_setpc_b64"):
            leaders.add(i + 1)
    leaders = sorted(x for x in leaders if x < len(ins))
    block_of, blocks = {}, []
    for b, start in enumerate(leaders):
        end = leaders[b + 1] if b + 1 < len(leaders) else len(ins)
        blocks.append((start, end))
        for i in range(start, end):
            block_of[i] = b
    succ = [[] for _ in blocks]
    for b, (start, end) in enumerate(blocks):
        last = ins[end - 1]
        m = BRANCH.match(last.mn + " " + ",".join(last.ops))
        if last.mn in ("s_endpgm", "s_setpc_b64"):
            continue
        if m:
            tgt = m.group(2)
            if tgt in label_at and label_at[tgt] < len(ins):
                succ[b].append(block_of[label_at[tgt]])
            if m.group(1) != "branch" and end < len(ins):
                succ[b].append(block_of[end])
        elif end < len(ins):
            succ[b].append(block_of[end])
    pred = [[] for _ in blocks]
    for b, ss in enumerate(succ):
        for t in ss:
            pred[t].append(b)
    # reaching definitions per VGPR: IN[b][reg] = set of defining instruction indices, -1 = "nothing written since the kernel began"
    ENTRY = -1
    nreg = 1 + max([r for q in ins for r in q.defs + q.addr] + [0])
    entry_defined = {0, 1, 2}                 # work-item ids
    IN = [None] * len(blocks)
    OUT = [None] * len(blocks)

    def transfer(b, state):
        st = dict(state)
        for i in range(*blocks[b]):
            for r in ins[i].defs:
                st[r] = frozenset([i])
        return st

    IN[0] = {r: frozenset([ENTRY]) for r in range(nreg) if r not in entry_defined}
    for r in entry_defined:
        IN[0][r] = frozenset([-2])            # defined by the launch
    work = [0]
    OUT[0] = None
    while work:
        b = work.pop()
        out = transfer(b, IN[b])
        if out == OUT[b]:
            continue
        OUT[b] = out
        for t in succ[b]:
            if IN[t] is None:
                IN[t] = dict(out)
                work.append(t)
            else:
                changed = False
                for r, ds in out.items():
                    u = IN[t].get(r, frozenset()) | ds
                    if u != IN[t].get(r):
                        IN[t][r] = u
                        changed = True
                if changed or OUT[t] is None:
                    work.append(t)
    findings = []
    for b, (start, end) in enumerate(blocks):
        if IN[b] is None:
            continue                           # unreachable
        st = dict(IN[b])
        for i in range(start, end):
            q = ins[i]
            for r in q.addr:
                for d in st.get(r, frozenset([ENTRY])):
                    if d == ENTRY:
                        findings.append((q.line, q.mn + " " + ", ".join(q.ops), "v%d" % r, "reachable with nothing written to it"))
                    elif d >= 0 and not ins[d].address_capable():
                        findings.append((q.line, q.mn + " " + ", ".join(q.ops), "v%d" % r,
                                         "can hold the result of line %d: %s %s" % (ins[d].line, ins[d].mn, ", ".join(ins[d].ops))))
            for r in q.defs:
                st[r] = frozenset([i])
    return findings, len(ins), sum(1 for q in ins if q.addr)


def check(path, wanted=("annp_fe_force_sh",)):
    """{kernel: findings}; raises if no kernel matches"""
    ks = kernel_bodies(path, wanted)
    if not ks:
        raise RuntimeError("no kernel of %s matches %r" % (path, wanted))
    res = {}
    for dem, body in ks.items():
        f, n, nmem = analyse(body)
        res[dem] = {"findings": f, "instructions": n, "memory_instructions": nmem}
    return res


def main():
    args = sys.argv[1:]
    path = args[0] if args and args[0].endswith(".s") else os.path.join(ROOT, "meng_zhang_amd", "csrc", "annp_hip.s")
    wanted = tuple(a for a in args if not a.endswith(".s")) or ("annp_fe_force_sh",)
    bad = 0
    for dem, r in check(path, wanted).items():
        print("%s: %d instructions, %d global-memory instructions, %d findings" % (dem.split("(")[0], r["instructions"], r["memory_instructions"], len(r["findings"])))
        for line, text, reg, what in r["findings"][:20]:
            print("   line %d: %s -- %s %s" % (line, text, reg, what))
        bad += len(r["findings"])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
