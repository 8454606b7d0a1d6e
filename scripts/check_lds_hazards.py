#!/usr/bin/env python3
"""Static check of a kernel's ISA for the hazard hand-placed LDS reads invite: a register that an in-flight
ds_read is still writing must not be touched (read, copied or overwritten) before the s_waitcnt lgkmcnt that
covers it -- the compiler does not know the inline-asm read is asynchronous and may, e.g., copy its destination
into a register tuple early.  Usage: check_lds_hazards.py file.s [kernel-name-substring]"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def main():
    path, key = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    lines = open(path).read().splitlines()
    inside, pending, bad, nreads = key == "", [], [], 0   # pending: list of register sets, oldest first
    for no, ln in enumerate(lines, 1):
        s = ln.strip()
        m = re.match(r"^([A-Za-z_][\w$]*):", s)     # a function label (".LBB" block labels start with a dot)
        if m:
            inside = key in m.group(1)
            pending = []
            continue
        if not inside or not s or s.startswith((";", ".")):
            continue
        op, _, rest = s.partition(" ")
        toks = [t.strip() for t in re.split(r"[,\s]+", rest.split(";")[0]) if t.strip()]
        if op.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if n else []
            continue
        touched = set()
        for t in toks:
            touched |= regs(t)
        inflight = set().union(*pending) if pending else set()
        if touched & inflight:
            bad.append((no, s, sorted(touched & inflight)))
        if op.startswith("ds_read"):
            pending.append(regs(toks[0]))
            nreads += 1
    print(f"{path}: {nreads} LDS reads checked, {len(bad)} hazards")
    for no, s, r in bad[:20]:
        print(f"  line {no}: {s}   <- in flight: v{r}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
