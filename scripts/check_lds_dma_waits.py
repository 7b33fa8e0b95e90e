#!/usr/bin/env python3
"""Static check of the gfx950 ISA: no s_barrier may be reached while a global_load_lds (LDS-DMA) issued by the
same wave is still un-waited.

Why: __syncthreads() is a workgroup-scope fence + s_barrier, and on gfx9 in non-tgsplit mode that fence only
needs lgkmcnt(0).  The LDS write of a global_load_lds is tracked by vmcnt, so a loop that prefetches the next
tile with LDS-DMA and relies on __syncthreads() alone can read a tile that has not landed.  Kernels must place an
explicit s_waitcnt vmcnt(N) before the barrier; this script proves they did, on the control-flow graph of the
compiled code (forward dataflow, union at joins).  Any vmcnt wait counts as a drain: counted waits (N > 0) are the
deliberate deep-ring accounting in gemm.hip, which is covered by its own parity tests.

The deep-ring GEMM instantiations (STAGES >= 3) keep tiles in flight ACROSS barriers on purpose with counted
s_waitcnt vmcnt(N); a path-insensitive analysis reports them, so they are skipped unless --strict is given.

usage: check_lds_dma_waits.py [--strict] [file.hip ...]     (default: every csrc/*.hip that uses LDS-DMA)
exit status 1 when a barrier can be reached with LDS-DMA pending.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sliders_amd", "csrc")


def per_file_flags(src):
    """Target-specific CXXFLAGS of the library's Makefile (`$(OBJDIR)/<name>.o: CXXFLAGS += ...`): the checks must look at the
    code the build produces, not at a differently-flagged compile of the same source."""
    base = os.path.splitext(os.path.basename(src))[0]
    mk = os.path.join(os.path.dirname(os.path.abspath(src)), "Makefile")
    flags = []
    if os.path.exists(mk):
        for line in open(mk):
            m = re.match(r"\$\(OBJDIR\)/" + re.escape(base) + r"\.o:\s*CXXFLAGS\s*\+=\s*(.*)", line)
            if m:
                flags += m.group(1).split()
    return flags


def device_asm(src):
    src = os.path.abspath(src)
    out = tempfile.mkdtemp(prefix="ldsdma_")
    base = os.path.splitext(os.path.basename(src))[0]
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *per_file_flags(src), "-I", os.path.join(ROOT, "include"),
                    "-c", src, "-o", os.path.join(out, base + ".o"), "-save-temps=obj"],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=out)
    for f in os.listdir(out):
        if f.endswith(".s") and "amdgcn" in f:
            return open(os.path.join(out, f)).read()
    raise RuntimeError("no device assembly for " + src)


def kernel_resources(asm):
    """{mangled name: {"scratch": bytes, "lds": bytes, "vgpr": n, "agpr_offset": n}} from the .amdhsa_kernel blocks"""
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", asm, re.S):
        def field(key, body=m.group(2)):
            f = re.search(r"\.amdhsa_" + key + r"\s+(\d+)", body)
            return int(f.group(1)) if f else -1
        out[m.group(1)] = {"scratch": field("private_segment_fixed_size"), "lds": field("group_segment_fixed_size"),
                           "vgpr": field("next_free_vgpr"), "agpr_offset": field("accum_offset")}
    return out


def kernels(asm):
    for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)\.end_amdhsa_kernel", asm, re.S | re.M):
        yield m.group(1), m.group(2)


def is_counted_ring(mangled):
    """deep-ring GEMM instantiation (template argument STAGES >= 3): counted vmcnt(N) accounting, see the header"""
    m = re.search(r"gemm_kernelILi\d+ELi\d+ELi\d+ELi(\d+)E", mangled)
    return bool(m) and int(m.group(1)) >= 3


def check_kernel(body):
    """returns the list of (instruction index, text) of barriers reachable with LDS-DMA pending"""
    ins = []
    for l in body.split("\n"):
        t = l.strip()
        if not t or t.startswith(";") or (t.startswith(".") and not re.match(r"\.LBB\S*:", t)):
            continue
        ins.append(t.split(";")[0].strip())
    label_at = {}
    for i, t in enumerate(ins):
        m = re.match(r"(\.LBB\S*):", t)
        if m:
            label_at[m.group(1)] = i

    def succ(i):
        t = ins[i]
        if t.startswith("s_endpgm"):
            return []
        m = re.match(r"s_branch\s+(\S+)", t)
        if m:
            return [label_at[m.group(1)]]
        m = re.match(r"s_cbranch_\w+\s+(\S+)", t)
        if m:
            return [label_at[m.group(1)], i + 1]
        return [i + 1] if i + 1 < len(ins) else []

    # pending_in[i]: some path reaches instruction i with an un-waited global_load_lds
    pending_in = [False] * len(ins)
    reached = [False] * len(ins)
    work = [0]
    reached[0] = True
    while work:
        i = work.pop()
        t = ins[i]
        p = pending_in[i]
        if t.startswith("global_load_lds") or (t.startswith("buffer_load") and " lds" in t):
            p = True
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            p = False
        for j in succ(i):
            if not reached[j] or (p and not pending_in[j]):
                reached[j] = True
                pending_in[j] = pending_in[j] or p
                work.append(j)
    return [(i, ins[i]) for i in range(len(ins)) if ins[i].startswith("s_barrier") and pending_in[i]]


def serial_load_sites(body):
    """(sites, loads): global / buffer loads that are waited for with vmcnt(0) before another load is issued - the signature of
    hipcc branching around a load that sits behind a condition (`if (c < C) v = *p;`) and waiting at the join: N such loads
    are N serial memory round trips.  Kernels avoid it with unconditional loads from a selected / clamped address."""
    ins = []
    for l in body.split("\n"):
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        ins.append(t.split(";")[0].strip())
    is_load = lambda t: t.startswith(("global_load", "buffer_load")) and "_lds" not in t.split()[0]
    hits = 0
    for i, t in enumerate(ins):
        if not is_load(t):
            continue
        for k in range(1, 4):
            if i + k >= len(ins) or is_load(ins[i + k]):
                break
            if ins[i + k].startswith("s_waitcnt") and "vmcnt(0)" in ins[i + k]:
                hits += 1
                break
    return hits, sum(1 for t in ins if is_load(t))


STRICT = False


def main():
    global STRICT
    files = [a for a in sys.argv[1:] if a != "--strict"]
    STRICT = "--strict" in sys.argv[1:]
    if not files:
        files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")
                 and "glds16" in open(os.path.join(CSRC, f)).read()]
    bad = 0
    for src in files:
        asm = device_asm(src)
        n = ring = 0
        for name, body in kernels(asm):
            if "global_load_lds" not in body:
                continue
            if is_counted_ring(name) and not STRICT:
                ring += 1            # counted vmcnt(N) ring: path-insensitive analysis cannot follow the accounting
                continue
            n += 1
            hits = check_kernel(body)
            if hits:
                bad += 1
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                print("UNWAITED LDS-DMA at barrier: %s  %s" % (os.path.basename(src), dem[:120]))
                for i, t in hits:
                    print("    instruction %d: %s" % (i, t))
        print("%s: %d LDS-DMA kernels checked%s" % (os.path.basename(src), n,
              ", %d deep-ring GEMM instantiations skipped (explicit counted waits; --strict lists them)" % ring if ring else ""))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
