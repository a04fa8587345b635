#!/usr/bin/env python3
"""Audit of matrix-instruction result hazards in the shipped code objects.

    python tools/hazard_audit.py [path/to/libcsvae.so] [--json out.json] [--verbose] [--slack N]

A matrix instruction (v_mfma_*) writes its destination registers several passes after it issues.  The
hardware does not interlock every consumer: between an MFMA and the first instruction that reads or
overwrites its destination (other than an MFMA that takes exactly the destination as its C
operand: a dependent chain, which the matrix pipe interlocks) a number of wait states must pass (an issued instruction = 1 state,
`s_nop N` = N + 1 states).  The compiler pads its own code; it cannot see inside inline asm, and this
library consumes accumulators from inline-asm VALU and hand-scheduled streams.  This tool checks the
result instead of the intent:

  1. the gfx950 code objects are taken out of the .so (.hip_fatbin section, clang offload bundles);
  2. each is disassembled (llvm-objdump -d --mcpu=gfx950);
  3. for every v_mfma in every kernel the instruction stream is followed forward -- through both sides
     of conditional branches and across unconditional ones -- until an instruction touches one of the
     destination registers or REQUIRED states have passed;
  4. per kernel it reports the number of MFMAs, the smallest distance found, and every MFMA whose
     destination is touched early (with the touching instruction).

REQUIRED states by the instruction's passes (1 pass = 4 cycles of the matrix pipe; a wait state is one
issue slot of the wave, 4 cycles).  The documented figure, which the compiler pads its own code to, is
passes + 4 counted from the ISSUE of the MFMA (12 for the 8-pass forms 32x32x16 bf16 and 16x16x4 f32).
That is not enough behind a run of back-to-back MFMAs: issue runs ahead of execution by up to one queued
instruction, so the last MFMA of a run may start `passes` states after it issued.  A build of this
library in which the compiler's own 12 states separated the last MFMA of the output layer from the
first v_accvgpr_read of its last register showed exactly that (last bits of two output columns differed
in a few launches per hundred; the registers of a tile are written pass by pass, the last ones last).  This
audit therefore demands, for a consumer that touches register i of an n-register destination,
passes + 4 + ceil((i + 1) * passes / n) states: 13 for the first and 20 for the last register of an 8-pass
tile, 36 for the last register of the 16-pass 32x32x2 f32 form.
The walk continues --slack N states (default 8) past REQUIRED, so the report also shows the margin of the
closest consumer of every kernel.  Exit status 1 when any early touch is found.
"""
import collections
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PASSES = {"v_mfma_f32_32x32x16_bf16": 8, "v_mfma_f32_32x32x16_f16": 8, "v_mfma_f32_16x16x32_bf16": 4, "v_mfma_f32_16x16x32_f16": 4, "v_mfma_f32_32x32x2_f32": 16,
          "v_mfma_f32_16x16x4_f32": 8}
NO_DST = ("global_store", "buffer_store", "flat_store", "scratch_store", "ds_write", "s_", "global_atomic", "buffer_atomic",
          "v_cmp", "v_cmpx")


def code_objects(so_path, tmp):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", so_path, os.path.join(tmp, "copy.so")])
    data = open(fat, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), data):
        base = m.start()
        (n,) = struct.unpack_from("<Q", data, base + len(MAGIC))
        pos = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, pos)
            triple = data[pos + 24:pos + 24 + tlen].decode()
            pos += 24 + tlen
            if "gfx950" in triple and size:
                path = os.path.join(tmp, f"co{len(out)}.o")
                with open(path, "wb") as f:
                    f.write(data[base + off:base + off + size])
                out.append(path)
    return out


REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(text):
    s = set()
    for m in REG.finditer(text):
        if m.group(1):
            s.add((m.group(1), int(m.group(2))))
        else:
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                s.add((m.group(3), i))
    return s


def disassemble(path):
    txt = subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", "--no-show-raw-insn", path], text=True)
    funcs, cur, start = {}, None, 0
    for line in txt.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            cur, start = m.group(2), int(m.group(1), 16)
            funcs[cur] = {"start": start, "ins": []}
            continue
        m = re.match(r"^\s+(\S.*?)\s*// ([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            funcs[cur]["ins"].append((int(m.group(2), 16), m.group(1).strip()))
    return funcs


def audit_function(name, fn, slack=0):
    ins = fn["ins"]
    by_addr = {a: i for i, (a, _) in enumerate(ins)}
    parsed = []
    for addr, text in ins:
        mn = text.split()[0]
        ops = text[len(mn):]
        target = None
        if mn.startswith("s_cbranch") or mn == "s_branch":
            m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", ops)
            if m:
                target = by_addr.get(fn["start"] + int(m.group(1), 16))
            else:
                m = re.search(r"<([^>+]+)>", ops)
                target = by_addr.get(fn["start"]) if m else None
        parsed.append((mn, ops, target))
    n_mfma, worst, early = 0, None, []
    hist = collections.Counter()
    for i, (mn, ops, _) in enumerate(parsed):
        if not mn.startswith("v_mfma"):
            continue
        n_mfma += 1
        passes = PASSES.get(mn, 16)
        oper = [o.strip() for o in ops.split(",")]
        dst = regs_of(oper[0])
        order = {r: k for k, r in enumerate(sorted(dst))}
        nreg = len(dst)
        horizon = 2 * passes + 4 + slack   # look this far; touches past their requirement are reported as the margin
        # forward walk: (index, states so far)
        best, who = None, None
        stack, seen = [(i + 1, 0)], {}
        while stack:
            j, st = stack.pop()
            while j < len(parsed) and st < horizon:
                if seen.get(j, 1 << 30) <= st:
                    break
                seen[j] = st
                m2, o2, tgt = parsed[j]
                if m2 == "s_endpgm":
                    break
                touched = regs_of(o2) & dst
                if touched:
                    chain = False
                    if m2.startswith("v_mfma"):
                        op2 = [o.strip() for o in o2.split(",")]
                        # dependent chain: the C operand is exactly the destination (the matrix pipe interlocks it,
                        # whatever the consumer's own destination), A / B elsewhere
                        chain = len(op2) >= 4 and regs_of(op2[3]) == dst and not (regs_of(op2[1]) | regs_of(op2[2])) & dst
                    if not chain:
                        last = max(order[r] for r in touched)
                        need = passes + 4 + -(-(last + 1) * passes // nreg)
                        if best is None or st - need < best[0] - best[1]:
                            best, who = (st, need), f"{ins[j][0]:x}: {ins[j][1]}"
                    break
                if m2 == "s_nop":
                    st += int(o2.strip() or 0) + 1
                else:
                    st += 1
                if m2 == "s_branch" and tgt is not None:
                    j = tgt
                    continue
                if m2.startswith("s_cbranch") and tgt is not None:
                    stack.append((tgt, st))
                j += 1
        if best is not None:
            best, req = best
            hist[best - req] += 1
            if worst is None or best - req < worst[0] - worst[1]:
                worst = (best, req, f"{ins[i][0]:x}: {ins[i][1]}", who)
            if best < req:
                early.append({"mfma": f"{ins[i][0]:x}: {ins[i][1]}", "states": best, "required": req, "touched_by": who})
    return {"kernel": name, "mfma": n_mfma, "early": early,
            "min_states_to_first_touch": worst[0] if worst else None, "required_at_min": worst[1] if worst else None,
            "closest": {"mfma": worst[2], "touched_by": worst[3]} if worst else None,
            "margin_histogram": {str(k): v for k, v in sorted(hist.items())}}


def main():
    import argparse
    ap = argparse.ArgumentParser(description="matrix-instruction result hazards in the code objects of libcsvae.so")
    ap.add_argument("so", nargs="?", default=os.path.join(ROOT, "vae_conditional_sampling_amd", "libcsvae.so"))
    ap.add_argument("--json")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--slack", type=int, default=8)
    a = ap.parse_args()
    so, verbose, slack = a.so, a.verbose, a.slack
    report = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(so, tmp):
            for name, fn in disassemble(co).items():
                if not any(t.startswith("v_mfma") for _, t in fn["ins"]):
                    continue
                report.append(audit_function(name, fn, slack))
    bad = 0
    for r in report:
        try:
            short = subprocess.run(["c++filt", r["kernel"]], capture_output=True, text=True).stdout.strip() or r["kernel"]
        except OSError:
            short = r["kernel"]
        r["kernel_demangled"] = short
        print(f"{short[:100]:100s} mfma {r['mfma']:5d}  early touches {len(r['early']):3d}"
              + (f"  closest touch {r['min_states_to_first_touch']} states (need {r['required_at_min']})"
                 if r["min_states_to_first_touch"] is not None else "  no touch within the horizon"))
        bad += len(r["early"])
        if verbose:
            for e in r["early"][:8]:
                print(f"      {e['mfma']}\n        touched after {e['states']} states (need {e['required']}) by {e['touched_by']}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump(report, f, indent=1)
    print(f"{len(report)} kernels with matrix instructions, {sum(r['mfma'] for r in report)} MFMAs, {bad} early touches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
