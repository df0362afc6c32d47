#!/usr/bin/env python3
"""Compares two ISA dumps made by scripts/isa_dump.sh, function by function.

    scripts/isa_diff.py /tmp/isa_a /tmp/isa_b [--mix SUBSTR]

Prints which functions exist only on one side and which differ (instruction counts on both sides); pc-relative call
offsets (the s_add_u32 / s_addc_u32 pair after s_getpc_b64) are masked, since they move when other functions come or go.
--mix SUBSTR: instruction mix of the functions of dump B whose (mangled) name contains SUBSTR.
"""
import collections
import re
import sys


def functions(path):
    out, name, body = {}, None, []
    for line in open(path):
        m = re.match(r"^<(.+)>:$", line.strip())
        if m:
            if name:
                out[name] = body
            name, body = m.group(1), []
        elif name and line.strip():
            body.append(line.strip())
    if name:
        out[name] = body
    return out


def masked(body):
    res, after_getpc = [], 0
    for ins in body:
        if ins.startswith("s_getpc_b64"):
            after_getpc = 2
        elif after_getpc and (ins.startswith("s_add_u32") or ins.startswith("s_addc_u32")):
            ins = re.sub(r", [^,]+$", ", <rel>", ins)
            after_getpc -= 1
        if ins.startswith(("s_cbranch", "s_branch")):
            ins = ins.split()[0] + " <target>"        # (relative targets move with any insertion in between)
        res.append(ins)
    return res


def mix(body):
    c = collections.Counter()
    for ins in body:
        op = ins.split()[0]
        if op.startswith("v_") and "f64" in op:
            c["valu_f64"] += 1
            if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
                c["valu_f64_quarter_rate"] += 1
        elif op.startswith("v_"):
            c["valu_other"] += 1
        elif op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
            c["branch"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith(("global_", "flat_", "scratch_", "buffer_")):
            c["vmem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        else:
            c["other"] += 1
    c["total"] = len(body)
    return dict(c)


def main():
    a_dir, b_dir = sys.argv[1], sys.argv[2]
    sub = sys.argv[sys.argv.index("--mix") + 1] if "--mix" in sys.argv else None
    rc = 0
    for f in ("kr_trace.s", "kr_post.s", "kr_capi.s"):
        fa, fb = functions(f"{a_dir}/{f}"), functions(f"{b_dir}/{f}")
        only_a, only_b = sorted(set(fa) - set(fb)), sorted(set(fb) - set(fa))
        diff = [n for n in fa if n in fb and masked(fa[n]) != masked(fb[n])]
        print(f"{f}: {len(fa)} / {len(fb)} functions, {len(only_a)} only in A, {len(only_b)} only in B, {len(diff)} differ")
        for n in only_a:
            print("   only A:", n[:150])
        for n in only_b:
            print("   only B:", n[:150])
        for n in diff:
            print(f"   differs: {n[:150]}  ({len(fa[n])} -> {len(fb[n])} instructions)")
            rc = 1
        if sub:
            for n, body in fb.items():
                if sub in n:
                    print("   mix", n[:120], mix(body))
    return rc


if __name__ == "__main__":
    sys.exit(main())
