"""Audit of tools/attention_pw/attention_pw.hip's generated code (run on the .s of `hipcc --save-temps`): every ds_read_b64_tr_b16 issued from an
asm statement (the compiler does not count it) must be followed by an `s_waitcnt ... lgkmcnt(0)` before any instruction reads or
overwrites its destination registers.  Prints the violations (none = exit 0)."""
import re
import sys


def main(path):
    lines = open(path).read().splitlines()
    # the product instantiations only (debug / timing variants carry a non-zero DBG template argument: ...Li<N>E...)
    starts = [i for i, l in enumerate(lines) if re.match(r"_ZN.*attn_pw\w*kernel\w*EEv.*:", l)]
    ends = [i for i, l in enumerate(lines) if "s_endpgm" in l]
    bad = n = 0
    for s0 in starts:
        e0 = min(e for e in ends if e > s0)
        for i in range(s0, e0):
            m = re.search(r"ds_read_b64_tr_b16 v\[(\d+):(\d+)\]", lines[i])
            if not m:
                continue
            n += 1
            regs = set(range(int(m.group(1)), int(m.group(2)) + 1))
            j = i + 1
            while j < e0 and not ("s_waitcnt" in lines[j] and "lgkmcnt(0)" in lines[j]):
                code = lines[j].split(";")[0]
                if "ds_read_b64_tr_b16" not in code:
                    hit = any(re.search(r"\bv%d\b" % r, code) for r in regs) or \
                        any(int(a) <= r <= int(b) for r in regs for a, b in re.findall(r"v\[(\d+):(\d+)\]", code))
                    if hit:
                        bad += 1
                        if bad <= 10:
                            print("line %d: destination of the read at line %d touched before lgkmcnt(0): %s" % (j + 1, i + 1, code.strip()))
                j += 1
    print("%d asm transposed reads, %d violations" % (n, bad))
    # asm MFMAs (between ;;#ASMSTART / ;;#ASMEND): the compiler pads nothing behind them, so for 18 wait states (16-pass XDL write ->
    # VALU / memory access) no instruction but another MFMA may touch the destination tuple - neither a read nor a copy
    nm = bad2 = 0
    for s0 in starts:
        e0 = min(e for e in ends if e > s0)
        in_asm = False
        for i in range(s0, e0):
            if "#ASMSTART" in lines[i]:
                in_asm = True
            elif "#ASMEND" in lines[i]:
                in_asm = False
            m = re.search(r"v_mfma\w+ ([av])\[(\d+):(\d+)\]", lines[i]) if in_asm else None
            if not m:
                continue
            nm += 1
            kind, lo, hi = m.group(1), int(m.group(2)), int(m.group(3))
            states, j = 0, i + 1
            while j < e0 and states < 18:
                code = lines[j].split(";")[0].strip()
                j += 1
                if not code or code.startswith("."):
                    continue
                mm = re.match(r"s_nop (\d+)", code)
                if mm:
                    states += int(mm.group(1)) + 1
                    continue
                if code.startswith("v_mfma"):
                    states += 8          # an MFMA holds the issue port for at least 8 cycles (4 cycles per wait state: >= 2)
                    continue
                states += 1
                touched = any(lo <= int(r) <= hi for r in re.findall(r"\b%s(\d+)\b" % kind, code)) or \
                    any(not (int(b) < lo or int(a) > hi) for a, b in re.findall(r"\b%s\[(\d+):(\d+)\]" % kind, code))
                if touched:
                    bad2 += 1
                    if bad2 <= 10:
                        print("line %d: destination %s[%d:%d] of the asm MFMA at line %d touched %d wait states later: %s"
                              % (j, kind, lo, hi, i + 1, states, code))
    print("%d asm MFMAs, %d destination-hazard violations" % (nm, bad2))
    return 1 if (bad or bad2) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
