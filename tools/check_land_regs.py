#!/usr/bin/env python3
"""Build-time check of the landing zone's contract (tools/gen_land_regs.py): in the compiler's assembly output, no instruction of the
one-wave-per-SIMD tvl1_iter kernels outside the inline-assembly blocks may touch the accumulation registers of the landing zone.
usage: python3 tools/check_land_regs.py <file.s>   (exit 1 on a violation)"""
import re
import sys


def main():
    text = open(sys.argv[1]).read().splitlines()
    base_of = {"4": 220, "6": 202, "8": 184}          # one zone (k_iter2_wave<PX, true>): a[256 - 9 PX : 255]
    base3_of = {"4": 184, "6": 148, "8": 112}         # two zones (k_iter3_wave<PX>): a[256 - 18 PX : 255]
    kern = None
    in_asm = False
    bad = 0
    seen = {}
    areg = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")
    for ln, line in enumerate(text, 1):
        m = re.match(r"^(_Z12k_iter[23]_waveILi(\d)E[A-Za-z0-9_]*):", line)
        if m:
            name = m.group(1)
            pf = "k_iter3" in name or "ELb1E" in name
            kern = (name, (base3_of if "k_iter3" in name else base_of)[m.group(2)]) if pf else None
            if kern:
                seen[name] = 0
            continue
        if kern is None:
            continue
        if "s_endpgm" in line:
            kern = None
            continue
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        code = line.split(";")[0]
        for mm in areg.finditer(code):
            hi = int(mm.group(1)) if mm.group(1) is not None else int(mm.group(3))
            if hi >= kern[1]:
                if in_asm:
                    seen[kern[0]] += 1
                else:
                    print(f"{sys.argv[1]}:{ln}: {kern[0]}: compiler-generated instruction touches the landing zone: {line.strip()}")
                    bad += 1
    if not seen or any(v == 0 for v in seen.values()):
        print("check_land_regs: expected kernels / landing-zone instructions not found:", seen)
        return 1
    print(f"check_land_regs: {len(seen)} kernels, landing zone untouched by the compiler" if not bad else f"check_land_regs: {bad} violations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
