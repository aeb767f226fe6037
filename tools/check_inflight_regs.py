#!/usr/bin/env python
"""Static check of split-phase inline-asm loads in the gfx950 ISA of a .hip file: between an inline-asm load whose destination is a
VGPR (`global_load_dword*`, `buffer_load_dword*`, `ds_read_*` -- not the LDS-DMA forms, which have no register destination) and the
`s_waitcnt` that retires it, no instruction may read or write the load's destination registers.

Why: the compiler believes an asm statement's "=v" output is valid the moment the statement ends.  Under register pressure it may
copy such a value (live-range split, spill to an AGPR) BEFORE the asm `s_waitcnt` that the kernel author placed -- right results on a
warm cache, garbage on a cold one (seen once in round 4, DESIGN K2).  Loads the compiler itself emits are tracked by the compiler;
only the asm ones are the author's responsibility, so only those are tracked here.

Counter model (gfx9): vmcnt counts every VMEM instruction (loads, LDS-DMA loads, stores, atomics) and returns in order, so
`vmcnt(N)` retires all but the youngest N; lgkmcnt counts LDS and scalar-memory instructions -- in order only while no SMEM is
outstanding, so `lgkmcnt(N > 0)` retires all but the youngest N only if no scalar load has been issued since the last `lgkmcnt(0)`,
otherwise nothing.  A compiler-emitted s_waitcnt retires like an asm one (a wait is a wait).

usage: check_inflight_regs.py <file.hip> [kernel name prefix]  -> exit 1 on a violation (or when no kernel matches)."""
import os, re, subprocess, sys, tempfile

_REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(text):
    """every VGPR / AGPR named in `text` as ('v'|'a', index)"""
    out = []
    for m in _REG.finditer(text):
        if m.group(1):
            out += [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
        else:
            out.append((m.group(4), int(m.group(5))))
    return out


def _is_vmem(op):
    return op.startswith(("global_", "buffer_", "flat_", "scratch_")) and not op.startswith(("buffer_wbl2", "buffer_inv", "buffer_gl"))


def _vm_dest_load(op):
    """a VMEM load with a VGPR destination (LDS-DMA loads -- `... lds` / global_load_lds_* -- land in LDS)"""
    return op.startswith(("global_load_", "buffer_load_", "flat_load_", "scratch_load_")) and "_lds_" not in op


def _is_lds(op):
    return op.startswith("ds_")


def _lds_dest_load(op):
    return op.startswith(("ds_read", "ds_load", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append", "ds_ordered_count")) or "_rtn" in op


def _is_smem(op):
    return op.startswith(("s_load_", "s_buffer_load_", "s_scratch_load", "s_memtime", "s_memrealtime", "s_dcache", "s_atc_probe"))


def scan_kernel(name, body):
    """body: list of (line number, text).  Returns (n_asm_loads, violations)."""
    inasm = False
    vm, lgkm = [], []              # in-flight ops in issue order: set of destination registers (empty for stores / untracked loads)
    smem_outstanding = False
    loads, bad = 0, []
    for ln, raw in body:
        t = raw.strip()
        if "#ASMSTART" in t:
            inasm = True
            continue
        if "#ASMEND" in t:
            inasm = False
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        parts = t.split(None, 1)
        op, rest = parts[0], (parts[1] if len(parts) > 1 else "")
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            if m:
                n = int(m.group(1))
                vm = vm[len(vm) - n:] if n else []
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                n = int(m.group(1))
                if n == 0:
                    lgkm, smem_outstanding = [], False
                elif not smem_outstanding:
                    lgkm = lgkm[len(lgkm) - n:]
            if re.fullmatch(r"\s*(0x[0-9a-fA-F]+|\d+)\s*", rest):       # raw immediate form: treat 0 as wait-all, else ignore
                if int(rest.strip(), 0) == 0:
                    vm, lgkm, smem_outstanding = [], [], False
            continue
        # a touch of an in-flight destination?
        pending = set().union(*vm) | set().union(*lgkm) if (vm or lgkm) else set()
        if pending:
            touched = [r for r in regs_of(rest) if r in pending]
            if touched:
                bad.append((name[:70], ln, t, touched[:4]))
        if _is_vmem(op):
            dest = set()
            if inasm and _vm_dest_load(op) and " lds" not in (" " + rest):
                dest = set(regs_of(rest.split(",")[0]))
                loads += 1
            vm.append(dest)
        elif _is_lds(op):
            dest = set()
            if inasm and _lds_dest_load(op):
                dest = set(regs_of(rest.split(",")[0]))
                loads += 1
            lgkm.append(dest)
        elif _is_smem(op):
            smem_outstanding = True
            lgkm.append(set())
    return loads, bad


def scan(asm_text, prefix=""):
    lines = asm_text.split("\n")
    kernels = [m.group(1) for m in (re.match(r"\s*\.type\s+(\S+),@function", l) for l in lines) if m]
    start = {}
    for i, l in enumerate(lines):
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", l)
        if m and m.group(1) in kernels and m.group(1) not in start:
            start[m.group(1)] = i
    out, n_k = [], 0
    for name in kernels:
        if name not in start or not name.startswith(prefix):
            continue
        n_k += 1
        i = start[name]
        body = []
        j = i + 1
        while j < len(lines) and not lines[j].lstrip().startswith((".Lfunc_end", ".end_amdhsa_kernel")):
            body.append((j - i, lines[j]))
            j += 1
        loads, bad = scan_kernel(name, body)
        out += bad
        out.append(("loads", name[:70], loads))
    return n_k, out


def allocated_vgprs(asm_text, name_prefix):
    """{kernel symbol: VGPRs the kernel descriptor allocates per lane (.amdhsa_next_free_vgpr)} for kernels whose symbol starts with
    name_prefix.  A kernel that lands asm loads in hard-coded registers ABOVE the range it is compiled for (amdgpu_num_vgpr) relies on
    the compiler counting the asm clobber list into this number: were it not counted, the wave would be given fewer registers than the
    asm writes -- the caller asserts the allocation covers the highest register named."""
    out, cur = {}, None
    for l in asm_text.split("\n"):
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", l)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"\s*\.amdhsa_next_free_vgpr\s+(\d+)", l)
        if m and cur is not None and cur.startswith(name_prefix):
            out[cur] = int(m.group(1))
        if ".end_amdhsa_kernel" in l:
            cur = None
    return out


def compile_to_asm(src):
    here = os.path.dirname(os.path.abspath(src))
    with tempfile.TemporaryDirectory() as d:
        s = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-I" + here, src, "-o", s],
                       check=True, capture_output=True)
        return open(s).read()


def main(src, prefix=""):
    kernels, res = scan(compile_to_asm(src), prefix)
    bad = [r for r in res if r[0] != "loads"]
    n_loads = sum(r[2] for r in res if r[0] == "loads")
    print(f"{os.path.basename(src)}: {kernels} kernel(s) matching {prefix!r}, {n_loads} split-phase asm loads, {len(bad)} touch(es) of an in-flight destination register")
    for b in bad[:10]:
        print("  ", b)
    return kernels, n_loads, bad


if __name__ == "__main__":
    k, n, bad = main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    sys.exit(1 if (bad or k == 0) else 0)
