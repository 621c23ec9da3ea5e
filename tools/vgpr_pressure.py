"""Where a gfx950 kernel's VGPR demand peaks: a liveness pass over the assembly hipcc leaves with -save-temps.

    hipcc ... -gline-tables-only -save-temps -c probe.hip         (tools/probe_*.hip instantiate one kernel each)
    python tools/vgpr_pressure.py probe-hip-amdgcn-amd-amdhsa-gfx950.s [kernel-substring] [top]

Backward dataflow over the kernel's basic blocks (labels and s_branch / s_cbranch targets); a write kills the register (writes
under a partial EXEC mask are treated the same, so the figure is a lower bound near divergent code). Prints the source lines
(.loc) at which most registers are live, and at the peak, how long each live register has been live (the long-lived ones are
the state carried across the hot loops — the ones worth re-deriving or parking in LDS). A development tool; nothing imports it."""
import collections
import os
import re
import sys

STORE = re.compile(r"^(global_store|flat_store|scratch_store|buffer_store|ds_write|ds_store|exp|global_atomic(?!.*\bsc0\b)|ds_add_u32|ds_min_u32|s_|buffer_wbl2|global_wb)")
RMW = re.compile(r"^(v_writelane|v_fmac|v_mac|v_pk_fmac|v_dot2c|v_movrel)")


def regs_of(op):
    out = []
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", op):
        out.extend(range(int(a), int(b) + 1))
    out.extend(int(x) for x in re.findall(r"\bv(\d+)\b", op))
    return out


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "k_path"
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    files, cur, kern = {}, None, None
    insts = []          # (mnemonic, defs, uses, loc, label-or-None, target-or-None, falls-through)
    labels = {}
    for line in open(path):
        line = line.split(";")[0].rstrip()
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]; continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2))); continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)
        if m:
            name = m.group(1)
            if name.startswith("_Z") or not name.startswith("."):
                kern = name if want in name else None
                if kern:
                    insts, labels = [], {}
            elif kern is not None:
                labels[name] = len(insts)
            continue
        if kern is None or not line.startswith("\t") or line.strip().startswith("."):
            continue
        txt = line.strip()
        if not txt:
            continue
        mn, _, rest = txt.partition(" ")
        if mn == "s_endpgm":
            insts.append((mn, [], [], cur, None, False)); continue
        ops = [o.strip() for o in rest.split(",")] if rest else []
        defs, uses = [], []
        if mn.startswith("s_cbranch") or mn == "s_branch":
            insts.append((mn, [], [], cur, ops[0], mn != "s_branch")); continue
        if STORE.match(mn) or not ops:
            for o in ops: uses.extend(regs_of(o))
        else:
            defs = regs_of(ops[0])
            for o in ops[1:]: uses.extend(regs_of(o))
            if RMW.match(mn): uses.extend(defs)
            if mn.startswith("v_swap"): uses.extend(defs); defs = defs + regs_of(ops[1])
        insts.append((mn, defs, uses, cur, None, True))
    n = len(insts)
    if not n:
        print("kernel not found"); return
    succ = []
    for i, (mn, d, u, loc, tgt, fall) in enumerate(insts):
        s = []
        if tgt is not None and tgt in labels: s.append(labels[tgt])
        if fall and i + 1 < n and mn != "s_endpgm": s.append(i + 1)
        succ.append(s)
    live_in = [frozenset()] * n
    changed = True
    rounds = 0
    while changed:
        changed = False; rounds += 1
        for i in range(n - 1, -1, -1):
            out = set()
            for s in succ[i]: out |= live_in[s]
            mn, d, u, loc, tgt, fall = insts[i]
            new = frozenset((out - set(d)) | set(u))
            if new != live_in[i]:
                live_in[i] = new; changed = True
    peak = max(range(n), key=lambda i: len(live_in[i]))
    at = os.environ.get("PEAK_AT")              # e.g. PEAK_AT=spectrum.h:192 : the busiest instruction of that source line instead
    if at:
        f, l = at.split(":")
        cand = [i for i in range(n) if insts[i][3] == (f, int(l))]
        if cand: peak = max(cand, key=lambda i: len(live_in[i]))
    print("%d instructions, %d dataflow rounds; peak %d live VGPRs at instruction %d (%s, %s)" %
          (n, rounds, len(live_in[peak]), peak, insts[peak][0], insts[peak][3]))
    by = collections.defaultdict(int)
    for i in range(n):
        loc = insts[i][3]
        by[loc] = max(by[loc], len(live_in[i]))
    for loc, v in sorted(by.items(), key=lambda kv: -kv[1])[:top]:
        print("  %-28s %4d" % ("%s:%d" % loc if loc else "?", v))
    # for the registers live at the peak: where (source line) each was last written before the peak in program order
    last = {}
    for i in range(peak):
        for r in insts[i][1]: last[r] = i
    ages = collections.Counter()
    for r in live_in[peak]:
        ages[insts[last[r]][3] if r in last else ("kernel entry", 0)] += 1
    print("registers live at the peak, by the source line that last wrote them:")
    for loc, c in ages.most_common(top): print("  %-28s %4d" % ("%s:%d" % loc, c))
    if os.environ.get("DETAIL"):                # every live register: the instruction that last wrote it, and how far back
        for r in sorted(live_in[peak], key=lambda r: last.get(r, -1)):
            i = last.get(r, -1)
            print("  v%-4d %7d  %-24s %s" % (r, peak - i if i >= 0 else -1, insts[i][0] if i >= 0 else "-", "%s:%d" % insts[i][3] if i >= 0 and insts[i][3] else ""))


if __name__ == "__main__":
    main()
