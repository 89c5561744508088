"""Audit of csrc/attention_pw.hip's generated code (run on the .s of `hipcc --save-temps`): every ds_read_b64_tr_b16 issued from an
asm statement (the compiler does not count it) must be followed by an `s_waitcnt ... lgkmcnt(0)` before any instruction reads or
overwrites its destination registers.  Prints the violations (none = exit 0)."""
import re
import sys


def main(path):
    lines = open(path).read().splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"_ZN.*attn_pw\w*kernel.*:", l)]
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
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
