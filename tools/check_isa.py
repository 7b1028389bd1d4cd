#!/usr/bin/env python3
"""Build-time ISA checks (run by `make -C pipeline-pointcloud_amd/csrc check-isa`, and by tests/test_cabi_cpu.py).

1. Every device-wide barrier of os_sort_fused_kernel (csrc/binning.hip: os_grid_barrier) must wait for the wave's outstanding
   stores and atomics (`s_waitcnt vmcnt(0)`) BEFORE the block barrier in front of the count-in atomic.  A workgroup-scope
   release fence does not emit that wait on gfx950.
"""
import re
import sys


def functions(text):
    cur, body = None, []
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur and line.startswith(".Lfunc_end"):
            yield cur, body
            cur = None
            continue
        if cur is not None:
            s = line.strip()
            if s and not s.startswith((";", ".")):
                body.append(s)


def check_grid_barriers(path):
    text = open(path).read()
    sites = bad = 0
    for name, body in functions(text):
        if "os_sort_fused_kernel" not in name:
            continue
        for i, ins in enumerate(body):
            # the count-in: a non-returning global_atomic_add through an SGPR base, the first VMEM instruction after an s_barrier
            if not (ins.startswith("global_atomic_add ") and " sc0" not in ins and re.search(r"s\[\d+:\d+\]", ins)):
                continue
            j = i - 1
            while j >= 0 and not body[j].startswith(("s_barrier", "global_", "buffer_", "flat_")):
                j -= 1
            if j < 0 or not body[j].startswith("s_barrier"):
                continue
            sites += 1
            # walk back from the s_barrier to the previous memory instruction: a vmcnt(0) must lie in between
            k, ok = j - 1, False
            while k >= 0 and not body[k].startswith(("global_", "buffer_", "flat_", "s_barrier")):
                if re.match(r"s_waitcnt\b.*vmcnt\(0\)", body[k]):
                    ok = True
                    break
                k -= 1
            if not ok:
                bad += 1
                print(f"check_isa: {name}: grid barrier at instruction {i} has no s_waitcnt vmcnt(0) in front of its s_barrier",
                      file=sys.stderr)
    return sites, bad


def main():
    path = sys.argv[1]
    sites, bad = check_grid_barriers(path)
    if sites < 4:          # two instantiations (DROP / not) x (histogram barrier + one per pass boundary, possibly unrolled)
        print(f"check_isa: expected at least 4 grid-barrier sites in {path}, found {sites}", file=sys.stderr)
        return 1
    print(f"check_isa: {sites} grid-barrier sites, {bad} without the wait")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
