#!/usr/bin/env python
"""Static check of split-phase asm loads: between an inline-asm `global_load_*` and the inline-asm `s_waitcnt vmcnt(0)` that awaits it,
no instruction may touch the load's destination registers.  The compiler believes an asm load's "=v" output is valid the moment the
statement ends, so under register pressure it may copy it (live-range split) BEFORE the wait -- right results on a warm cache, garbage on
a cold one (seen once in round 4, DESIGN K2).  usage: check_inflight_regs.py <file.hip> <kernel name prefix> -> exit 1 on a violation."""
import os, re, subprocess, sys, tempfile


def regs_of(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return [int(m.group(1))] if m else []


def scan(asm_text, prefix):
    lines = asm_text.split("\n")
    out, kernels = [], 0
    i = 0
    while i < len(lines):
        if lines[i].startswith(prefix) and lines[i].rstrip().split(";")[0].rstrip().endswith(":"):
            kernels += 1
            inasm, pending, loads = False, {}, 0
            j = i + 1
            while j < len(lines) and "s_endpgm" not in lines[j]:
                t = lines[j].strip()
                if "#ASMSTART" in t:
                    inasm = True
                elif "#ASMEND" in t:
                    inasm = False
                elif inasm and t.startswith("global_load_dword") and "lds" not in t.split()[0]:
                    loads += 1
                    for r in regs_of(t.split()[1].rstrip(",")):
                        pending[r] = j
                elif inasm and t.startswith("s_waitcnt vmcnt(0)"):
                    pending = {}
                elif pending and t and not t.startswith(";"):
                    for o in re.findall(r"v\[\d+:\d+\]|v\d+", t):
                        for r in regs_of(o):
                            if r in pending:
                                out.append((lines[i].split(":")[0][:60], j - i, t, r))
                j += 1
            out.append(("loads", lines[i].split(":")[0][:60], loads))
            i = j
        i += 1
    return kernels, out


def main(src, prefix):
    here = os.path.dirname(os.path.abspath(src))
    with tempfile.TemporaryDirectory() as d:
        s = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-I" + here, src, "-o", s],
                       check=True, capture_output=True)
        kernels, res = scan(open(s).read(), prefix)
    bad = [r for r in res if r[0] != "loads"]
    n_loads = sum(r[2] for r in res if r[0] == "loads")
    print(f"{kernels} kernel(s) matching {prefix!r}, {n_loads} split-phase asm loads, {len(bad)} touch(es) of an in-flight destination register")
    for b in bad[:10]:
        print("  ", b)
    return kernels, n_loads, bad


if __name__ == "__main__":
    k, n, bad = main(sys.argv[1], sys.argv[2])
    sys.exit(1 if (bad or k == 0) else 0)
