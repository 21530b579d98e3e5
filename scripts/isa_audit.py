"""Static audit of the gfx950 code in librobosat_hip.so: an LDS WRITE that another wave reads behind the next s_barrier must have been
waited for (s_waitcnt lgkmcnt) by the writing wave BEFORE it signals the barrier.

Why this exists (round 6, profiles/r06/dma_order.txt): the kernels whose LDS-DMA waits are inline asm (s_waitcnt vmcnt(N)) call
__syncthreads() for the barrier, and hipcc drops that call's own `s_waitcnt lgkmcnt(0)` in loops like theirs -- the round-5 "counted wait"
defect was a gather table written by wave 0 (ds_write_b32), a bare s_barrier, and the other waves' ds_read_b32 three instructions behind it.
With a neighbour's LDS traffic on the CU the reads overtook the write.  Nothing in a parity test or on an idle CU shows it; the ISA does.

Rule R1, per kernel, as a forward dataflow over the control-flow graph of the disassembly:
    state = no LDS write pending | oldest pending LDS write has `age` younger DS operations behind it
    ds_write* / ds atomics          -> pending (age 0) if nothing was pending, else age + 1
    any other ds_* operation        -> age + 1            (DS operations of a wave complete in order)
    s_waitcnt lgkmcnt(n), n <= age  -> nothing pending    (outstanding <= n <= age: the write has completed; SMEM only raises the count)
    s_barrier with a write pending  -> VIOLATION
    joins                           -> pending if any predecessor is pending, the smaller age

Usage:  python scripts/isa_audit.py [path/to/librobosat_hip.so]      (exit code 1 on an unexpected violation)
The one kernel that is SUPPOSED to violate R1 is the race screen's positive control (conv_wgrad_bf16<128, 64, .., RING = 4, DEAD = 0>).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
# the positive control of tests/test_gpu_race_screen.py: kept exactly as round 5 shipped it (mangled-name fragment)
EXPECTED = ("conv_wgrad_bf16ILi128ELi64ELi2ELi2ELi64ELb0ELi4ELi0EE",)

_WRITE = re.compile(r"^ds_(write|add|sub|rsub|inc|dec|min|max|and|or|xor|mskor|cmpst|wrxchg|wrap|pk_add|condxchg)")
_LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def code_objects(so_path, workdir):
    """Every gfx950 code object bundled into the shared library (one per translation unit)."""
    fat = os.path.join(workdir, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path], check=True)
    data = open(fat, "rb").read()
    starts = []
    i = data.find(MAGIC)
    while i >= 0:
        starts.append(i)
        i = data.find(MAGIC, i + 1)
    out = []
    for k, s in enumerate(starts):
        e = starts[k + 1] if k + 1 < len(starts) else len(data)
        b = os.path.join(workdir, "b%d.bin" % k)
        co = os.path.join(workdir, "b%d.co" % k)
        open(b, "wb").write(data[s:e])
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + b, "--targets=" + TARGET,
                            "--output=" + co], capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def disassemble(co):
    """{kernel: [(address, mnemonic, operands)] with ('label', name) markers}"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--symbolize-operands", co], capture_output=True, text=True, check=True).stdout
    funcs, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            name = m.group(1)
            if re.fullmatch(r"L\d+", name):
                if cur is not None:
                    cur.append(("label", name, ""))
            else:
                cur = funcs.setdefault(name, [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body, _, tail = line.strip().partition("//")
        parts = body.split(None, 1)
        if not parts:
            continue
        addr = tail.split(":")[0].strip()
        cur.append((addr, parts[0], parts[1].strip() if len(parts) > 1 else ""))
    return funcs


def _blocks(ins):
    """basic blocks: list of dicts(label, body, succ labels / fallthrough)"""
    blocks, cur = [], {"label": None, "body": [], "succ": [], "fall": True}
    for it in ins:
        if it[0] == "label":
            if cur["body"] or cur["label"] is not None:
                blocks.append(cur)
            cur = {"label": it[1], "body": [], "succ": [], "fall": True}
            continue
        cur["body"].append(it)
        op, args = it[1], it[2]
        if op == "s_branch":
            cur["succ"].append(args.split()[0])
            cur["fall"] = False
        elif op.startswith("s_cbranch"):
            cur["succ"].append(args.split()[-1])
        elif op in ("s_endpgm", "s_setpc_b64"):
            cur["fall"] = False
        else:
            continue
        blocks.append(cur)
        cur = {"label": None, "body": [], "succ": [], "fall": True}
    if cur["body"] or cur["label"] is not None:
        blocks.append(cur)
    return blocks


def _merge(a, b):
    if a is None:
        return b
    if b is None:
        return a
    return (min(a[0], b[0]), a[1])


def audit_kernel(ins):
    """R1 violations of one kernel: [(address of the s_barrier, address of the pending write)]"""
    blocks = _blocks(ins)
    index = {b["label"]: i for i, b in enumerate(blocks) if b["label"] is not None}
    n = len(blocks)
    state_in = [None] * n  # None | (age, write address)
    reached = [False] * n
    reached[0] = True
    work = [0]
    violations = {}
    while work:
        i = work.pop()
        st = state_in[i]
        for addr, op, args in blocks[i]["body"]:
            if op.startswith("ds_"):
                if _WRITE.match(op):
                    st = (0, addr) if st is None else (st[0] + 1, st[1])
                elif st is not None:
                    st = (st[0] + 1, st[1])
            elif op == "s_waitcnt":
                m = _LGKM.search(args)
                if m is None and re.fullmatch(r"(0x)?[0-9a-f]+", args.strip()):  # a raw immediate: lgkmcnt = bits 11:8
                    cnt = (int(args.strip(), 0) >> 8) & 0xF
                elif m is not None:
                    cnt = int(m.group(1))
                else:
                    cnt = None
                if cnt is not None and st is not None and cnt <= st[0]:
                    st = None
            elif op == "s_barrier" and st is not None:
                violations[addr] = st[1]
                st = None  # (report each barrier once; what follows it is judged on its own)
        succ = [index[s] for s in blocks[i]["succ"] if s in index]
        if blocks[i]["fall"] and i + 1 < n:
            succ.append(i + 1)
        for j in succ:
            new = _merge(state_in[j], st) if reached[j] else st
            if not reached[j] or new != state_in[j]:
                reached[j] = True
                state_in[j] = new
                work.append(j)
    return sorted(violations.items())


def audit_library(so_path):
    """{kernel: violations} for every kernel of the library with at least one R1 violation"""
    bad = {}
    with tempfile.TemporaryDirectory() as wd:
        for co in code_objects(so_path, wd):
            for name, ins in disassemble(co).items():
                v = audit_kernel(ins)
                if v:
                    bad[name] = v
    return bad


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robosat_amd", "librobosat_hip.so")
    bad = audit_library(so)
    unexpected = 0
    for name, v in sorted(bad.items()):
        exp = any(e in name for e in EXPECTED)
        unexpected += 0 if exp else 1
        print("%s %s" % ("control " if exp else "VIOLATION", name))
        for barrier, write in v:
            print("    s_barrier at %s with the LDS write at %s not waited for" % (barrier, write))
    print("isa_audit: %d kernel(s) publish an LDS write through a barrier without waiting for it (%d unexpected)" % (len(bad), unexpected))
    return 1 if unexpected else 0


if __name__ == "__main__":
    sys.exit(main())
