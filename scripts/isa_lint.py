#!/usr/bin/env python3
"""Static lint of the gfx950 assembly hipcc emits for the HIP kernels (`build.py --emit-asm` -> build/*.s).

Why: round 1 chased "run-to-run unstable tiles" through hand-padded inline-asm MFMAs.  Round 2 found (DESIGN.md
section 5) that (a) with the MFMA *builtin* hipcc's hazard recogniser pads every MFMA dependency itself, and (b) the
one reproducible instability sat in SLP-vectorised packed-fp32 epilogue code.  This lint keeps both facts true of
whatever is built from now on:

  R1  no v_mfma inside an inline-asm block (;;#ASMSTART .. ;;#ASMEND): the compiler cannot pad what it cannot see;
  R2  the MFMA wait-state table of hipcc ROCm 7.2 for gfx950 (derived from compiler output for probe kernels,
      scripts/probes/mfma_hazard_table.hip) holds on the final instruction stream, loop back-edges included:
        VALU / v_accvgpr_write result  ->  MFMA SrcA/B/C ....................... >= 2 wait states
        MFMA vDst -> any non-MFMA reader or writer (VALU, LDS/global store data)
                     and MFMA SrcA/B ............................................ >= passes + 4
        MFMA vDst -> v_accvgpr_read / _write / _mov of those registers ......... >= passes + 3
        MFMA vDst -> MFMA SrcC, same registers exactly ......................... 0 (hardware interlock)
        MFMA vDst -> MFMA SrcC, overlapping but not identical .................. >= passes + 2
      (passes: 16x16x32 bf16 = 4, 32x32x16 bf16 = 8; an instruction is one wait state, `s_nop N` is N + 1)
  R3  no element-shuffled packed fp32 arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with op_sel / op_sel_hi /
      neg modifiers on a vector-register source): the forms the SLP vectoriser emits; the library is built with
      -fno-slp-vectorize, so an occurrence is a build-flag regression (plain packed ops from explicit float4 code pass);
  R4  (report only) scratch: scratch_* instructions / .private_segment_fixed_size != 0 name kernels that spill.

usage: isa_lint.py file.s [file.s ...]   (exit status 1 when a rule is violated)
"""
import re
import sys

MFMA_PASSES = {"16x16x32": 4, "32x32x16": 8, "16x16x16": 4, "32x32x8": 8, "4x4x4": 1, "32x32x4": 8, "16x16x4": 4}
REG = re.compile(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b")
WINDOW = 24  # instructions looked back (max requirement is 12 wait states)


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


class Inst:
    __slots__ = ("line", "text", "op", "ops", "in_asm", "ws", "is_mfma", "passes", "dst", "srcs", "srcc")

    def __init__(self, line, text, in_asm):
        self.line, self.text, self.in_asm = line, text, in_asm
        parts = text.split(None, 1)
        self.op = parts[0]
        self.ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        self.ws = 1
        if self.op == "s_nop" and self.ops:
            self.ws = int(self.ops[0], 0) + 1
        self.is_mfma = self.op.startswith("v_mfma") or self.op.startswith("v_smfmac")
        self.passes = 0
        if self.is_mfma:
            shape = next((s for s in MFMA_PASSES if s in self.op), None)
            self.passes = MFMA_PASSES.get(shape, 8)
        # destination = first operand for everything that writes a vector register
        writes = self.op.startswith(("v_", "ds_read", "ds_bpermute", "ds_permute", "global_load", "buffer_load", "flat_load",
                                     "scratch_load")) and not self.op.startswith(("v_cmp", "v_nop"))
        if self.op.startswith("v_cmp") and self.ops and not self.ops[0].startswith(("s", "vcc")):
            writes = False
        self.dst = regs_of(self.ops[0]) if (writes and self.ops) else set()
        src_ops = self.ops[1:] if writes else self.ops
        self.srcs = set()
        for o in src_ops:
            self.srcs |= regs_of(o)
        self.srcc = regs_of(self.ops[3]) if (self.is_mfma and len(self.ops) > 3) else set()
        if self.op.startswith("v_readlane") or self.op.startswith("v_readfirstlane"):
            self.dst = set()


def shuffled_pk(text):
    """v_pk_*_f32 whose VGPR sources are element-shuffled or negated (op_sel with a 1, op_sel_hi with a 0, neg_lo / neg_hi on
    a vector-register operand): the forms hipcc's SLP vectoriser builds out of scalar code.  Plain packed ops (explicit
    float2 / float4 arithmetic in the source) and broadcasts of an SGPR or a constant are not flagged."""
    body = text.split(None, 1)[1]
    operands = [o.strip() for o in re.split(r",(?![^\[]*\])", re.split(r"\s+(?:op_sel|op_sel_hi|neg_lo|neg_hi):", body)[0])]
    srcs = operands[1:]
    is_vec = [bool(re.match(r"[av](\[|\d)", o)) for o in srcs]

    def bits(key):
        m = re.search(key + r":\[([01,]+)\]", body)
        return [int(b) for b in m.group(1).split(",")] if m else None

    sel, sel_hi, nlo, nhi = bits("op_sel"), bits("op_sel_hi"), bits("neg_lo"), bits("neg_hi")
    for i, vec in enumerate(is_vec):
        if not vec:
            continue
        if sel and i < len(sel) and sel[i] == 1:
            return True
        if sel_hi and i < len(sel_hi) and sel_hi[i] == 0:
            return True
        if (nlo and i < len(nlo) and nlo[i]) or (nhi and i < len(nhi) and nhi[i]):
            return True
    return False


def parse(path):
    kernels, cur, name, in_asm, priv = {}, None, None, False, {}
    labels = {}
    for ln, raw in enumerate(open(path), 1):
        s = raw.rstrip("\n")
        st = s.strip()
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^([A-Za-z_][\w$.]*):", s)
        if m and not s.startswith(".L") and not s.startswith("\t"):
            name = m.group(1)
            cur = kernels.setdefault(name, [])
            labels[name] = {}
            continue
        m = re.match(r"^(\.L[\w$.]*):", s)
        if m and cur is not None:
            labels[name][m.group(1)] = len(cur)
            continue
        m = re.match(r"\s*\.private_segment_fixed_size\s+(\d+)", s)
        if m and name:
            priv[name] = int(m.group(1))
        if cur is None or not s.startswith("\t"):
            continue
        code = st.split(";")[0].strip()
        if not code or code.startswith("."):
            continue
        if code.startswith("s_endpgm"):
            cur.append(Inst(ln, code, in_asm))
            cur = None
            continue
        cur.append(Inst(ln, code, in_asm))
    return kernels, labels, priv


def check_stream(name, seq, errs, path):
    """seq: list of Inst in execution order (straight line)."""
    for i, cur in enumerate(seq):
        dist = 0
        for j in range(i - 1, max(-1, i - 1 - WINDOW), -1):
            prev = seq[j]
            if prev.op in ("s_branch", "s_endpgm", "s_setpc_b64"):
                break  # `cur` is not reached by falling through from here
            need = 0
            why = ""
            if cur.is_mfma:
                if prev.is_mfma:
                    ab = cur.srcs - cur.srcc
                    if prev.dst & ab:
                        need, why = prev.passes + 4, "MFMA vDst -> MFMA SrcA/B"
                    elif prev.dst & cur.srcc and prev.dst != cur.srcc:
                        need, why = prev.passes + 2, "MFMA vDst -> overlapping MFMA SrcC"
                elif prev.op.startswith("v_") and prev.dst & cur.srcs:
                    need, why = 2, "VALU write -> MFMA operand"
            elif prev.is_mfma and (prev.dst & (cur.srcs | cur.dst)):
                # (hipcc asks one state less in front of the accumulator copy instructions than in front of other readers)
                need, why = prev.passes + (3 if cur.op.startswith("v_accvgpr") else 4), "MFMA vDst -> non-MFMA access"
            if need and dist < need:
                errs.append(f"{path}:{cur.line}: [{name}] R2 {why}: {dist} wait states between `{prev.text}` (line {prev.line}) "
                            f"and `{cur.text}`, need {need}")
            dist += prev.ws
            if dist >= 12:
                break


def lint(path):
    kernels, labels, priv = parse(path)
    errs = []
    n_mfma = 0
    for name, insts in kernels.items():
        if not insts:
            continue
        for ins in insts:
            if ins.is_mfma:
                n_mfma += 1
                if ins.in_asm:
                    errs.append(f"{path}:{ins.line}: [{name}] R1 v_mfma inside an inline-asm block: `{ins.text}`")
            if re.match(r"v_pk_(add|mul|fma)_f32", ins.op) and shuffled_pk(ins.text):
                errs.append(f"{path}:{ins.line}: [{name}] R3 element-shuffled packed fp32 `{ins.text}` (build with -fno-slp-vectorize)")
            if ins.op.startswith("scratch_"):
                errs.append(f"{path}:{ins.line}: [{name}] R4 scratch access `{ins.text}`")
        if priv.get(name, 0) != 0:
            errs.append(f"{path}: [{name}] R4 .private_segment_fixed_size {priv[name]}")
        check_stream(name, insts, errs, path)
        # loop back-edges: tail of the loop body followed by its head
        for idx, ins in enumerate(insts):
            if ins.op.startswith(("s_cbranch", "s_branch")) and ins.ops:
                tgt = labels[name].get(ins.ops[-1])
                if tgt is not None and tgt <= idx:
                    tail = insts[max(0, idx - WINDOW):idx + 1]
                    head = insts[tgt:tgt + WINDOW]
                    sub = []
                    check_stream(name, tail + head, sub, path)
                    # keep only findings whose consumer lies in the head and producer in the tail
                    head_lines = {h.line for h in head}
                    for e in sub:
                        m = re.search(r":(\d+): \[", e)
                        if m and int(m.group(1)) in head_lines and e not in errs and "(line " in e:
                            pl = int(re.search(r"\(line (\d+)\)", e).group(1))
                            if pl >= tail[0].line and pl > insts[tgt].line:
                                errs.append(e + "  [across the loop back-edge]")
    return errs, sum(1 for k in kernels.values() if k), n_mfma


def main(argv):
    total = []
    for p in argv:
        errs, nk, nm = lint(p)
        print(f"{p}: {nk} functions, {nm} MFMA instructions, {len(errs)} findings")
        total += errs
    by_rule = {}
    for e in total:
        m = re.search(r"\] (R\d)", e)
        by_rule.setdefault(m.group(1) if m else "?", []).append(e)
    for rule in sorted(by_rule):
        print(f"{rule}: {len(by_rule[rule])} findings")
        for e in by_rule[rule][:8]:
            print("   ", e)
    return 1 if any(r != "R4" for r in by_rule) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
