#!/usr/bin/env python3
"""Control flow of one kernel of a code object: basic blocks, their instruction mix, and every branch with what it skips.

    scripts/isa_cfg.py <hsaco> <kernel-name substring> [--loop]     (the .hsaco files are left by scripts/isa_dump.sh in its output directory)

A lone wave (the strict side launch's critical ray) pays ~30 cycles for every branch that waits for a vector compare, so a branch around fewer than
~8 vector instructions costs more than it saves there; this lists the candidates."""
import re, subprocess, sys

hsaco, sub = sys.argv[1], sys.argv[2]
txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", hsaco], capture_output=True, text=True).stdout.split("\n")
start = end = None
for i, l in enumerate(txt):
    if re.match(r"^[0-9a-f]+ <.*>:$", l):
        if start is not None:
            end = i
            break
        if sub in l:
            start = i
end = end or len(txt)
ins = []
for l in txt[start + 1:end]:
    m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", l)
    if m:
        ins.append((int(m.group(2), 16), m.group(1)))
addr_index = {a: i for i, (a, _) in enumerate(ins)}


def target(i):
    a, t = ins[i]
    op, arg = t.split()[0], t.split()[1]
    off = int(arg)
    if off >= 32768:
        off -= 65536
    return addr_index.get(a + 4 + 4 * off)


leaders = {0}
for i, (a, t) in enumerate(ins):
    if t.startswith(("s_cbranch", "s_branch")):
        tg = target(i)
        if tg is not None:
            leaders.add(tg)
        leaders.add(i + 1)
    if t.startswith(("s_endpgm", "s_setpc")):
        leaders.add(i + 1)
leaders = sorted(x for x in leaders if x < len(ins))
blocks = [(leaders[k], leaders[k + 1] if k + 1 < len(leaders) else len(ins)) for k in range(len(leaders))]
block_of = {}
for k, (b, e) in enumerate(blocks):
    for i in range(b, e):
        block_of[i] = k


def mix(b, e):
    v = sum(1 for i in range(b, e) if ins[i][1].startswith("v_"))
    f = sum(1 for i in range(b, e) if ins[i][1].startswith("v_") and "f64" in ins[i][1].split()[0])
    s = sum(1 for i in range(b, e) if ins[i][1].startswith("s_"))
    m = sum(1 for i in range(b, e) if ins[i][1].startswith(("global_", "scratch_", "flat_", "buffer_", "ds_")))
    return v, f, s, m


print(f"{len(ins)} instructions, {len(blocks)} blocks")
for k, (b, e) in enumerate(blocks):
    v, f, s, m = mix(b, e)
    last = ins[e - 1][1]
    br = ""
    if last.startswith(("s_cbranch", "s_branch")):
        tg = target(e - 1)
        br = f"{last.split()[0]} -> B{block_of.get(tg, '?')}"
        if last.startswith("s_cbranch") and tg is not None and tg > e - 1:
            sv, sf, ss, sm = mix(e, tg)
            br += f"   skips {tg - e} instr ({sv} valu, {sm} mem)"
        # where does the condition come from?
        cond = "vcc" if "vcc" in last else "exec" if "exec" in last else "scc"
        for j in range(e - 2, max(b - 1, e - 12), -1):
            t = ins[j][1]
            if (cond == "vcc" and re.search(r"\bvcc\b", t.split(",")[0]) and t.startswith("v_cmp")) or (cond == "exec" and "exec" in t.split(",")[0]) or (cond == "scc" and t.startswith(("s_cmp", "s_and", "s_or", "s_bit", "s_andn2"))):
                br += f"   [{t.split()[0]} {e - 1 - j} before]"
                break
    print(f"B{k:<3} [{b:5}..{e:5})  {e - b:4} instr  valu {v:4} (f64 {f:4})  salu {s:3}  mem {m:2}   {br}")
