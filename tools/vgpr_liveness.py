#!/usr/bin/env python
"""VGPR liveness of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only): where the register pressure
peaks and which registers are live there.  A developer aid for the register-bound kernels (knn_f16.hpp's pruned walk).

usage: tools/vgpr_liveness.py file.s <substring of the kernel's symbol> [--top N] [--at LINE]

The listing's instructions are parsed into basic blocks (labels, s_branch / s_cbranch_*), every VGPR operand is
classified as definition (first operand of anything that is not a store / compare / lane read) or use, and the usual
backward data-flow is iterated to a fixed point.  Writes under a partial exec mask count as full definitions, so the
numbers are a slight under-estimate; they are for finding WHAT is live, not for predicting the allocator.
"""
import re, sys

STORE = re.compile(r"^(global_store|flat_store|scratch_store|buffer_store|ds_write|ds_store|global_atomic|flat_atomic|buffer_atomic|s_|v_cmp|v_cmpx|v_readlane|v_readfirstlane|ds_gws|ds_nop|buffer_wbl2|buffer_inv|global_load_lds)")
RMW = re.compile(r"^(v_fmac|v_mac|v_pk_fmac|v_writelane|v_dot\d+c|v_mfma.*|v_smfmac.*|v_swap)")


def regs_of(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def parse(lines):
    """-> list of blocks: dict(label, insts=[(line_no, mnemonic, defs, uses)], succ=[labels], fall=bool)"""
    blocks, cur = [], dict(label="entry", insts=[], succ=[], fall=True)
    for n, raw in enumerate(lines):
        l = raw.split(";")[0].strip()
        if not l:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = dict(label=m.group(1), insts=[], succ=[], fall=True)
            continue
        if l.startswith(".") or l.endswith(":"):
            continue
        parts = l.split(None, 1)
        mn = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        if mn == "s_branch":
            cur["succ"].append(ops[0]); cur["fall"] = False
            blocks.append(cur); cur = dict(label=None, insts=[], succ=[], fall=True)
            continue
        if mn.startswith("s_cbranch"):
            cur["succ"].append(ops[0])
            blocks.append(cur); cur = dict(label=None, insts=[], succ=[], fall=True)
            continue
        if mn in ("s_endpgm", "s_setpc_b64"):
            cur["fall"] = False
            blocks.append(cur); cur = dict(label=None, insts=[], succ=[], fall=True)
            continue
        defs, uses = [], []
        if ops:
            if STORE.match(mn):
                for o in ops: uses += regs_of(o)
            else:
                defs = regs_of(ops[0])
                for o in ops[1:]: uses += regs_of(o)
                if RMW.match(mn) and not (mn.startswith("v_mfma") and len(ops) > 3 and not regs_of(ops[3])):
                    uses += defs
                if mn.startswith("v_mfma") and len(ops) > 3:
                    uses += regs_of(ops[3])
        cur["insts"].append((n, mn, set(defs), set(uses)))
    blocks.append(cur)
    return blocks


def main():
    path, sym = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
    at = int(sys.argv[sys.argv.index("--at") + 1]) if "--at" in sys.argv else None
    text = open(path).read().split("\n")
    start = next(i for i, l in enumerate(text) if re.match(r"^_Z\S*:", l) and sym in l)
    end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
    lines = text[start:end + 1]
    blocks = parse(lines)
    index = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    succ = []
    for i, b in enumerate(blocks):
        s = [index[t] for t in b["succ"] if t in index]
        if b["fall"] and i + 1 < len(blocks): s.append(i + 1)
        succ.append(s)
    live_in = [set() for _ in blocks]
    changed = True
    while changed:
        changed = False
        for i in range(len(blocks) - 1, -1, -1):
            live = set()
            for s in succ[i]: live |= live_in[s]
            for (_, _, d, u) in reversed(blocks[i]["insts"]):
                live = (live - d) | u
            if live != live_in[i]:
                live_in[i] = live; changed = True
    # per-instruction pressure
    rows = []
    for i, b in enumerate(blocks):
        live = set()
        for s in succ[i]: live |= live_in[s]
        for (n, mn, d, u) in reversed(b["insts"]):
            rows.append((len(live | d), n, mn, frozenset(live | d)))
            live = (live - d) | u
    rows.sort(key=lambda r: r[1])
    print("kernel lines %d..%d, %d blocks; peak live VGPRs %d" % (start + 1, end + 1, len(blocks), max(r[0] for r in rows)))
    # pressure profile: max per 50 source lines
    prof = {}
    for c, n, mn, _ in rows: prof[n // 50] = max(prof.get(n // 50, 0), c)
    print("max live per 50 listing lines:", " ".join("%d:%d" % (k * 50, v) for k, v in sorted(prof.items())))
    last_def = {}
    defline = {}
    for b in blocks:
        for (n, mn, d, u) in b["insts"]:
            for r in d: defline.setdefault(r, []).append(n)
    peaks = sorted(rows, key=lambda r: -r[0])[:top] if at is None else [r for r in rows if r[1] == at - start - 1] or [min(rows, key=lambda r: abs(r[1] - (at - start - 1)))]
    seen = set()
    for c, n, mn, live in peaks:
        if n // 20 in seen: continue
        seen.add(n // 20)
        print("\nline %d (%s): %d live" % (start + 1 + n, mn, c))
        groups = {}
        for r in sorted(live):
            ds = [x for x in defline.get(r, []) if x <= n]
            key = ds[-1] if ds else -1
            groups.setdefault(key, []).append(r)
        for key in sorted(groups):
            src = lines[key].strip()[:70] if key >= 0 else "(kernel entry)"
            print("  def@%d %-72s v%s" % (start + 1 + key, src, ",".join(map(str, groups[key]))))


if __name__ == "__main__":
    main()
