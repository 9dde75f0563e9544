"""CPU (-m "not gpu"): the counted `s_waitcnt vmcnt(N)` pipelines of the built library against its own disassembly
(scripts/isa_check.py).  K3r's ring (csrc/conv3d_coarse.hip) and conv11's residual prefetch
(csrc/conv3d_mfma.hip) assume an exact number of VMEM loads per stage; the dynamic gates (bit equality against the vmcnt(0)
forms, -m gpu) only catch a mis-count that happens to race on the test box -- this one reads the code object."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import isa_check  # noqa: E402

LIB = os.path.join(ROOT, "dmvsnet_amd", "csrc", "libdmvs_hip.so")


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    ks = isa_check.disassemble(LIB)
    return ks, isa_check.demangled(list(ks))


def test_k3r_ring_counts_match_the_counted_wait(kernels):
    ks, names = kernels
    seen = set()
    for k, insts in ks.items():
        m = re.search(r"coarse_kernel<(\d+), (\d+), (\d+), (true|false)>", names[k])
        if not m:
            continue
        kd, cin, v4 = int(m.group(1)), int(m.group(2)), m.group(4) == "true"
        ns = isa_check.coarse_ns(kd, cin, v4)
        assert isa_check.check_ring(insts, ns) == [], names[k]
        seen.add((kd, cin, v4))
    # every instantiation the launcher can pick (conv3d_coarse.hip: dmvs_conv3d_coarse)
    assert seen == {(kd, cin, v4) for kd in (1, 3) for cin in (32, 64) for v4 in (True, False)}


def test_zmarch_ring_counts_match_the_counted_wait(kernels):
    """K3z (csrc/conv3d_zmarch.hip) uses the same ring discipline; its NS comes from the same geometry formula."""
    ks, names = kernels
    found = 0
    for k, insts in ks.items():
        m = re.search(r"zmarch_kernel<(\d+)>", names[k])
        if not m:
            continue
        found += 1
        # ring of 2 (the product build): every plane load has landed at the barrier (vmcnt(0)); ring of 3: the counted form
        assert isa_check.check_ring(insts, isa_check.zmarch_ns(), ring=int(m.group(1))) == [], names[k]
    assert found, "no zmarch_kernel in the library"



def test_conv11_prefetch_has_sixteen_loads_behind_the_last_tile_load(kernels):
    ks, names = kernels
    pref = [k for k in ks if re.search(r"deconv_mfma_kernel<\d+, \d+, \d+, \d+, \d+, (true|false), true", names[k])]
    assert pref, "no PREF instantiation of deconv_mfma_kernel found"
    for k in pref:
        assert isa_check.check_prefetch(ks[k], 16) == [], names[k]


def _prog(lines):
    return [(4 * i, mn, ops_, None) for i, (mn, ops_) in enumerate(lines)]


def test_the_analysis_flags_miscounts():
    """The analysis has teeth: one load too few per stage, a wait that is too loose, or a load the compiler turned into a
    plain (non-LDS) load are all reported; the well-formed ring passes."""
    lds = ("buffer_load_dwordx4", "v1, s[0:3], 0 offen lds")
    mfma = ("v_mfma_f32_16x16x4_f32", "a[0:3], v0, v1, a[0:3]")
    def ring(ns_issue, wait, stages=3, ns=2):
        p = [lds] * (2 * ns)
        for _ in range(stages):
            p += [("s_waitcnt", f"vmcnt({wait})"), ("s_barrier", ""), mfma] + [lds] * ns_issue + [mfma]
        p += [("s_waitcnt", "lgkmcnt(0)"), ("s_barrier", ""), ("buffer_store_dwordx4", "v[0:3], v4, s[4:7], 0 offen"),
              ("s_waitcnt", "vmcnt(0)"), ("s_endpgm", "")]
        return _prog(p)
    assert isa_check.check_ring(ring(2, 2), 2) == []
    assert isa_check.check_ring(ring(2, 0)[2:], 2, ring=2) == []                               # a ring of two drains at every barrier
    assert any("may be outstanding" in b for b in isa_check.check_ring(ring(2, 1)[2:], 2, ring=2))
    assert any("LDS-DMA loads since" in b for b in isa_check.check_ring(ring(1, 2), 2))      # a load went missing
    assert any("may be outstanding" in b for b in isa_check.check_ring(ring(2, 3), 2))       # the wait is too loose
    assert any("counted wait is gone" in b for b in isa_check.check_ring(ring(2, 0), 2))     # the optimisation is gone (vmcnt(0))
    pre = [lds, lds] + [("global_load_dwordx2", "v[2:3], v[4:5], off")] * 16 + [("s_waitcnt", "vmcnt(16)"), ("s_endpgm", "")]
    assert isa_check.check_prefetch(_prog(pre), 16) == []
    assert isa_check.check_prefetch(_prog(pre[:10] + pre[11:]), 16) != []                     # 15 loads behind the tile loads
    assert isa_check.check_prefetch(_prog(pre[:5] + [lds] + pre[5:]), 16) != []               # a tile load among the residual loads
