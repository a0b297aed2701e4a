#!/usr/bin/env python3
"""Static instruction statistics of the dominant kernels' main loops, from the gfx950 disassembly of THIS build
(hipcc -save-temps; no GPU needed).  Writes profiles/isa_stats.json, keyed by kernel and tagged with the kernel source
fingerprint bench.py checks, so `roofline.valu` in the bench line can be recomputed from a tracked file.

usage: python tools/isa_stats.py [round-tag, e.g. r02]"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "tee_optical_flow_amd", "csrc")

CLASSES = [
    ("f64", re.compile(r"^v_(fma|fmac|mul|add|max|min|ldexp|trunc|floor|rndne|fract|cmp\w*)_f64|^v_cvt_f64|^v_cvt_\w+_f64|^v_(rsq|rcp|sqrt)_f64")),
    ("transcendental_f32", re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32")),
    ("packed_f32", re.compile(r"^v_pk_")),
    ("mov", re.compile(r"^v_mov_b(32|64)|^v_accvgpr")),
    ("select_cmp", re.compile(r"^v_cndmask|^v_cmp|^v_cmpx")),
    ("cvt_round", re.compile(r"^v_cvt_|^v_rndne_f32|^v_floor_f32|^v_trunc_f32|^v_div_fixup")),
    ("f32_alu", re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|mad|max|min|med3|max3|min3)_(f32|legacy_f32)")),
    ("int_alu", re.compile(r"^v_")),
]


def flags():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"FLAGS\s*:=\s*(.*?)\n\S", mk.replace("\\\n", " "), re.S)
    fl = m.group(1).replace("$(ARCH)", "gfx950").split()
    return fl


def kernel_text(asm, mangled_prefix):
    out, on = [], False
    for line in asm.splitlines():
        if not on and re.match(rf"^{mangled_prefix}\w*:", line):
            on = True
        if on:
            out.append(line)
            if ".end_amdhsa_kernel" in line:
                break
    return out


def loop_stats(lines):
    """The innermost loop with the most instructions: blocks tagged 'in Loop: Header=<H>' plus the header block."""
    label_at = [(i, m.group(1), l) for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m]
    headers = [(i, name) for i, name, l in label_at if "Inner Loop Header" in l]
    best = None
    for hi, hname in headers:
        tag = "Header=" + hname[2:]          # ".LBB6_88" -> "BB6_88"
        member = [i for i, name, l in label_at if name == hname or ("in Loop: " + tag) in l]
        if not member:
            continue
        lo = min(member)
        last = max(member)
        nxt = [i for i, _, _ in label_at if i > last]
        hi_end = nxt[0] if nxt else len(lines)
        # blocks without a label comment (fallthrough pieces) inside [lo, hi_end) belong to the loop as well
        body = [l.strip() for l in lines[lo:hi_end] if re.match(r"^\s+[a-z]", l)]
        if best is None or len(body) > len(best):
            best = body
    return best or []


def classify(body):
    res = {"valu": 0, "salu": 0, "lds": 0, "vmem_load": 0, "vmem_store": 0, "s_nop": 0, "s_waitcnt": 0, "s_barrier": 0, "branch": 0}
    by = {c: 0 for c, _ in CLASSES}
    hist = {}
    for ins in body:
        op = ins.split()[0]
        hist[op] = hist.get(op, 0) + 1
        if op.startswith("v_"):
            res["valu"] += 1
            for c, rx in CLASSES:
                if rx.match(op):
                    by[c] += 1
                    break
        elif op == "s_nop":
            res["s_nop"] += 1
        elif op == "s_waitcnt":
            res["s_waitcnt"] += 1
        elif op == "s_barrier":
            res["s_barrier"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            res["branch"] += 1
        elif op.startswith("s_"):
            res["salu"] += 1
        elif op.startswith("ds_"):
            res["lds"] += 1
        elif op.startswith(("global_load", "buffer_load", "flat_load")):
            res["vmem_load"] += 1
        elif op.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")):
            res["vmem_store"] += 1
    res["valu_by_class"] = by
    res["top_opcodes"] = dict(sorted(hist.items(), key=lambda kv: -kv[1])[:25])
    return res


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    from bench import kernel_source_fingerprint
    with tempfile.TemporaryDirectory() as td:
        cmd = ["/opt/rocm/bin/hipcc"] + flags() + ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage",
                                                   os.path.join(CSRC, "teeflow.hip"), "-o", os.path.join(td, "t.so")]
        r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if r.returncode:
            raise SystemExit(r.stderr[-2000:])
        asm = open([os.path.join(td, f) for f in os.listdir(td) if f.endswith("gfx950.s")][0]).read()
        remarks = r.stderr
    out = {}
    for kern, prefix, unit, per_step in (("k_iter2_rows", "_Z12k_iter2_rows", "px_iterations_per_wave_step", 64 * 4 * 2),):
        lines = kernel_text(asm, prefix)
        st = classify(loop_stats(lines))
        m = re.search(rf"Function Name: {prefix}\w*.*?VGPRs: (\d+).*?Occupancy \[waves/SIMD\]: (\d+)", remarks, re.S)
        sg = re.search(rf"Function Name: {prefix}\w*.*?TotalSGPRs: (\d+)", remarks, re.S)
        out[kern] = {"round": tag, "source_fingerprint": kernel_source_fingerprint(),
                     "what": "static count over the main loop body (one pipeline step of a 256-thread block = per wave: 64 lanes x 4 px x 2 iterations); "
                             "the REPLAY-only stores sit inside the same loop behind a block-uniform branch",
                     unit: per_step, "valu_insts_per_wave_step": st["valu"], "vgprs": int(m.group(1)) if m else None,
                     "waves_per_simd": int(m.group(2)) if m else None, "sgprs": int(sg.group(1)) if sg else None, **st}
    path = os.path.join(ROOT, "profiles", "isa_stats.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    k = out["k_iter2_rows"]
    print(json.dumps({kk: k[kk] for kk in ("source_fingerprint", "valu_insts_per_wave_step", "vgprs", "waves_per_simd", "salu", "lds",
                                           "vmem_load", "vmem_store", "s_nop", "s_barrier", "valu_by_class")}, indent=1))


if __name__ == "__main__":
    main()
