#!/usr/bin/env python3
"""Static check of the one-launch wide encoder's weight ring (wide_fused_kernels.hip): replays the kernel's ISA --
prologue once, the pass loop three times -- with the hardware's rule that vector-memory instructions retire in
order and `s_waitcnt vmcnt(N)` returns once at most N are outstanding, and verifies at every ring handshake
(`s_waitcnt vmcnt(N)` ... `s_barrier`) that the stage the waves go on to read has landed whatever N the source
chose there (sync_extras: allowances for the signal loads and head stores known to be younger than that stage).
Handshake b (counting from the prologue's) releases stage b; a stage is four LDS-direct loads per wave.
Usage: check_vmcnt_ring.py file.s [kernel-name-substring]"""
import re
import sys


def main():
    path, key = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "wide_fused_kernel")
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_][\w$]*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    # The pass loop, as the compiler lays it out: a run of blocks marked "in Loop" / "Loop Header" (its header
    # may sit at the bottom, entered by a long jump).  Replayed in layout order -- the blocks besides the body
    # hold no vector-memory instruction but a full drain (s_waitcnt vmcnt(0)) in the header, so layout order
    # differs from execution order only in skipping that drain before the first pass: the stricter replay.
    ins, loop_blocks, in_loop, cur, block_start = [], [], False, None, -1
    for l in lines[start + 1:end]:
        code = l.split(";")[0].strip()
        if re.match(r"^(\.LBB[\w$]+):", code) or re.match(r"^; %bb\.\d+:", l):   # a block starts
            in_loop = "Loop" in l
            cur = [len(ins), len(ins)] if in_loop else None
            block_start = len(ins)
            if cur:
                loop_blocks.append(cur)
            continue
        if not code and "Loop" in l and cur is None and block_start == len(ins):
            # the loop annotation on its own comment line under the label (-save-temps -fverbose-asm puts the IR
            # block name on the label's line and the loop membership on the next)
            in_loop, cur = True, [len(ins), len(ins)]
            loop_blocks.append(cur)
            continue
        if not code or code.startswith("."):
            continue
        ins.append(code)
        if cur:
            cur[1] = len(ins)
    # merge adjacent loop blocks into spans, keep the span with the LDS-direct loads
    spans = []
    for a0, a1 in loop_blocks:
        if spans and spans[-1][1] == a0:
            spans[-1][1] = a1
        else:
            spans.append([a0, a1])
    spans = [sp for sp in spans if any(x.startswith("global_load_lds") for x in ins[sp[0]:sp[1]])]
    if len(spans) != 1:
        print(f"{path}: expected one loop holding LDS-direct loads, found {len(spans)}")
        return 1
    top, bot = spans[0]
    order = list(range(0, top)) + list(range(top, bot)) * 3 + list(range(bot, len(ins)))
    queue, ndma, nsync, bad, last_wait, max_n = [], 0, 0, [], None, 0   # queue: outstanding ops, oldest first
    for pos, i in enumerate(order):
        s = ins[i]
        op = s.split()[0]
        if op.startswith("global_load_lds"):
            queue.append(("dma", ndma // 4))
            ndma += 1
        elif re.match(r"(global|buffer|flat|scratch)_(load|store|atomic)", op):
            queue.append(("x", op))
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                queue = queue[len(queue) - n:] if n < len(queue) else queue
                last_wait = (pos, n)
        elif op == "s_barrier":
            if last_wait is None or pos - last_wait[0] > 4:
                bad.append(f"instruction {i}: s_barrier without a vmcnt wait in front of it")
                continue
            max_n = max(max_n, last_wait[1])
            pending = [q for q in queue if q[0] == "dma" and q[1] <= nsync]
            if pending:
                bad.append(f"handshake {nsync} (instruction {i}, vmcnt({last_wait[1]})): stage {nsync} still has "
                           f"{len(pending)} load(s) in flight")
            nsync += 1
    for b in bad[:20]:
        print(b)
    print(f"{path}: {nsync} handshakes replayed, {ndma // 4} stages, largest vmcnt {max_n}, {len(bad)} violations")
    return 1 if bad or nsync == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
