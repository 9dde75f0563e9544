"""Static check of the COUNTED `s_waitcnt vmcnt(N)` pipelines in libdmvs_hip.so.

Two kernels keep vector-memory loads in flight across a wait and rely on the number of VMEM instructions the compiler emits:

* K3r `coarse_kernel<KD, CIN, NCB, V4>` (csrc/conv3d_coarse.hip): a ring of three LDS stages; at the end of stage k a wave
  waits `vmcnt(NS)` -- the NS LDS-DMA loads it issued for stage k + 2 may stay in flight, everything older (stage k + 1) must
  have landed before the barrier.  Safe iff, at EVERY `s_barrier` on EVERY path, at most the NS most recent VMEM instructions
  can still be outstanding AND exactly NS LDS-DMA loads were issued since the previous barrier ((RING - 1) NS before the first).
* conv11's residual prefetch `deconv_mfma_kernel<..., PREF = true, ...>` (csrc/conv3d_mfma.hip): the wait of chunk 1 is
  `vmcnt(16)` -- the 16 residual loads issued after chunk 1's tile loads may stay in flight.  Safe iff exactly 16 VMEM
  instructions, none of them an LDS-DMA load or a store, sit between the last LDS-DMA load and that wait on every path.

The check disassembles the device code objects of the shared library (`llvm-objdump --offloading`, then `-d`), builds the
control-flow graph of each kernel and runs a forward data-flow analysis over the abstract state
    (lds  = LDS-DMA loads since the last barrier,
     g    = upper bound of VMEM instructions that can still be outstanding: +1 per VMEM instruction, min(g, N) at vmcnt(N),
     tail = VMEM instructions since the last LDS-DMA load, or None once a store / LDS-DMA load intervened)
as a SET of states per instruction (all paths), reset at barriers.  No GPU needed: `pytest -m "not gpu"` runs it
(tests/test_static_isa.py) on the library build() produced, so a toolchain bump that changes the instruction counts fails the
CPU suite instead of producing an intermittent stale-tile read.

    python scripts/isa_check.py [path/to/libdmvs_hip.so]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
CAP = 255


def disassemble(lib_path):
    """-> {mangled kernel name: [(addr, mnemonic, operands, branch target addr | None)]} over all gfx950 bundles of the library."""
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    if not os.path.exists(objdump):
        objdump = shutil.which("llvm-objdump")
    if not objdump:
        raise RuntimeError("llvm-objdump not found")
    kernels = {}
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, so)
        subprocess.run([objdump, "--offloading", so], cwd=tmp, check=True, capture_output=True)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            text = subprocess.run([objdump, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            cur, base = None, 0
            for line in text.splitlines():
                m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
                if m:
                    cur, base = m.group(2), int(m.group(1), 16)
                    kernels[cur] = []
                    continue
                m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):", line)
                if not (m and cur):
                    continue
                mn, ops_, addr = m.group(1), m.group(2), int(m.group(3), 16)
                tgt = None
                if mn.startswith("s_cbranch") or mn == "s_branch":
                    t = re.search(r"<([^>+]+)(?:\+0x([0-9a-f]+))?>\s*$", line)
                    if t:
                        tgt = base + (int(t.group(2), 16) if t.group(2) else 0)
                kernels[cur].append((addr, mn, ops_, tgt))
    return kernels


def _is_vmem(mn):
    return mn.startswith(("buffer_load", "buffer_store", "global_load", "global_store", "buffer_atomic", "global_atomic",
                          "flat_load", "flat_store", "scratch_"))


def _vmcnt(mn, ops_):
    if mn != "s_waitcnt":
        return None
    m = re.search(r"vmcnt\((\d+)\)", ops_)
    return int(m.group(1)) if m else None


def successors(insts, exec_active=False):
    """CFG edges.  ``exec_active``: a wave never has an all-zero EXEC mask inside the kernel (the exec-masked regions are guarded by
    lane-dependent conditions that hold for some lane), so `s_cbranch_execz` always falls through and `s_cbranch_execnz` is always
    taken -- the paths on which an exec-masked load region is skipped as a whole are not real."""
    index = {a: i for i, (a, _, _, _) in enumerate(insts)}
    n = len(insts)
    succ = [[] for _ in range(n)]
    for i, (a, mn, ops_, tgt) in enumerate(insts):
        if mn == "s_endpgm":
            continue
        if mn == "s_branch" or (exec_active and mn == "s_cbranch_execnz"):
            succ[i].append(index[tgt])
            continue
        if i + 1 < n:
            succ[i].append(i + 1)
        if mn.startswith("s_cbranch") and tgt is not None and not (exec_active and mn == "s_cbranch_execz"):
            succ[i].append(index[tgt])
    return succ


def analyse(insts, exec_active=False):
    """Forward data-flow over the kernel's CFG.  Returns (states_at_barrier, states_at_wait):
    states_at_barrier: [(addr, set of (lds, g, tail), guards)] for every s_barrier, state BEFORE the barrier; ``guards`` = a matrix
    instruction is reachable from it before the next barrier (it protects tile reads, not only the exchange area);
    states_at_wait: {N: [(addr, set of states BEFORE the wait)]} for every `s_waitcnt vmcnt(N)`."""
    n = len(insts)
    succ = successors(insts, exec_active)
    state_in = [set() for _ in range(n)]
    state_in[0].add((0, 0, None))
    work = [0]
    while work:
        i = work.pop()
        a, mn, ops_, _ = insts[i]
        out = set()
        for lds, g, tail in state_in[i]:
            if mn == "s_barrier":
                lds, tail = 0, tail
            elif _is_vmem(mn):
                g = min(g + 1, CAP)
                if " lds" in (" " + ops_) and mn.startswith("buffer_load"):
                    lds, tail = min(lds + 1, CAP), 0
                elif "store" in mn or "atomic" in mn:
                    tail = None
                elif tail is not None:
                    tail = min(tail + 1, CAP)
            else:
                v = _vmcnt(mn, ops_)
                if v is not None:
                    g = min(g, v)
            out.add((lds, g, tail))
        for j in succ[i]:
            new = out - state_in[j]
            if new:
                state_in[j] |= new
                work.append(j)
    def guards(i):   # an MFMA before the next barrier?
        seen, stack = set(), list(succ[i])
        while stack:
            j = stack.pop()
            if j in seen:
                continue
            seen.add(j)
            if insts[j][1].startswith("v_mfma"):
                return True
            if insts[j][1] != "s_barrier":
                stack.extend(succ[j])
        return False

    barriers = [(insts[i][0], state_in[i], guards(i)) for i in range(n) if insts[i][1] == "s_barrier"]
    waits = {}
    for i in range(n):
        v = _vmcnt(insts[i][1], insts[i][2])
        if v is not None:
            waits.setdefault(v, []).append((insts[i][0], state_in[i]))
    return barriers, waits


def coarse_ns(kd, cin, v4):
    """NS of CoarseGeom<KD, CIN, NCB, V4> (csrc/conv3d_coarse.hip): LDS-DMA loads per wave and ring stage."""
    cps = 16 if kd == 3 else 32
    ps0 = kd * 200
    ps = ps0 + (32 - ps0 % 64 + 64) % 64
    pf = 4 if v4 else 1
    ni = (cps * ps // pf + 63) // 64
    return (ni + 7) // 8


def zmarch_ns(cin=16):
    """NS of ZGeom<RING> (csrc/conv3d_zmarch.hip): one input plane of 16 channels x (10 x 20 floats padded to 224) per stage,
    16-byte pieces, 4 waves."""
    ni = (cin * 224 // 4 + 63) // 64
    return (ni + 3) // 4


def check_ring(insts, ns, ring=3, first_lds=None, roles=False, exec_active=False):
    """The K3r rule (also used by K3z / K3z0, csrc/conv3d_zmarch.hip).  ``ring`` LDS stages: the loads of stage k + ring - 1 are issued
    during stage k, so at a barrier (ring - 2) * ns VMEM instructions may still be outstanding -- ns for the ring of three (the
    counted wait), none for a ring of two (vmcnt(0)).  Returns a list of violation strings (empty = pass)."""
    barriers, waits = analyse(insts, exec_active=exec_active)
    bad = []
    if not barriers:
        return ["no s_barrier found"]
    first = min(a for a, _, _ in barriers)
    first_lds = (ring - 1) * ns if first_lds is None else first_lds
    allowed = (ring - 2) * ns
    counted_seen = False
    for addr, states, guards in barriers:
        if not states or not guards:
            continue   # unreachable, or the closing barrier of the kernel (exchange area only: no tile is read behind it)
        for lds, g, _ in states:
            # ``roles``: the workgroup has waves that never load tiles (K3z0's finisher waves: exchange reads and stores only,
            # on the other side of a wave-uniform branch) -- a path without a load since the last barrier is theirs and reads no tile
            if roles and lds == 0:
                continue
            # the first barrier of a peeled loop copy can also be reached from the loop: accept either count there
            if not (lds == ns or lds == first_lds):
                bad.append(f"barrier @{addr:#x}: {lds} LDS-DMA loads since the previous barrier, expected {ns}")
            if g > allowed:
                bad.append(f"barrier @{addr:#x}: up to {g} VMEM instructions may be outstanding, the ring allows {allowed}")
            counted_seen |= g == allowed
    if allowed and allowed not in waits:
        bad.append(f"no `s_waitcnt vmcnt({allowed})` in the kernel: the counted wait is gone")
    if not counted_seen:
        bad.append("no barrier is reached with the counted wait in effect")
    return bad


def check_prefetch(insts, n=16):
    """conv11's rule: every `s_waitcnt vmcnt(n)` is reached with exactly n plain loads since the last LDS-DMA load."""
    _, waits = analyse(insts, exec_active=True)   # the tile loads sit in `lane < row width` exec regions
    if n not in waits:
        return [f"no `s_waitcnt vmcnt({n})` in the kernel"]
    bad = []
    for addr, states in waits[n]:
        for _, g, tail in states:
            if tail != n:
                bad.append(f"vmcnt({n}) @{addr:#x}: {tail} plain VMEM loads since the last LDS-DMA load, expected {n}")
    return bad


def demangled(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main(lib_path):
    kernels = disassemble(lib_path)
    names = demangled(list(kernels))
    rc = 0
    for k, insts in kernels.items():
        d = names[k]
        m = re.search(r"coarse_kernel<(\d+), (\d+), (\d+), (true|false)>", d)
        if m:
            ns = coarse_ns(int(m.group(1)), int(m.group(2)), m.group(4) == "true")
            bad = check_ring(insts, ns)
        elif re.search(r"zmarch_kernel<(\d+)>", d):
            ring = int(re.search(r"zmarch_kernel<(\d+)>", d).group(1))
            ns = zmarch_ns()
            bad = check_ring(insts, ns, ring=ring)
            m = True
        elif "deconv_mfma_kernel<" in d and re.search(r"deconv_mfma_kernel<\d+, \d+, \d+, \d+, \d+, (true|false), true", d):
            bad = check_prefetch(insts, 16)
        else:
            continue
        short = re.sub(r"\(anonymous namespace\)::|void ", "", d).split("(")[0]
        print(("FAIL " if bad else "ok   ") + short + (f"  NS = {ns}" if m else "  vmcnt(16)"))
        for b in bad[:8]:
            print("     " + b)
        rc |= bool(bad)
    return rc


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "dmvsnet_amd", "csrc", "libdmvs_hip.so")))
