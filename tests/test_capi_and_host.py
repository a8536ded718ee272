"""`-m "not gpu"`: the C-ABI library builds for gfx950, loads, exports every symbol include/floria_hip.h
declares, fails loudly without a device, and its host-side functions match the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(hip_lib):
    hdr = open(os.path.join(ROOT, "include", "floria_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(floria_hip_\w+)\s*\(", hdr)))
    assert declared == sorted(hip_lib.SYMBOLS), "lib.SYMBOLS out of sync with the header"
    L = hip_lib.load()
    for s in declared:
        assert hasattr(L, s), f"libfloria_hip.so does not export {s}"
    assert b"gfx950" in L.floria_hip_version()


def test_no_cpu_fallback_when_device_missing(hip_lib):
    """Without a usable HIP device create() must fail with FLORIA_E_DEVICE (the product never computes on CPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hip_lib.FloriaHipError) as ei:
        hip_lib.FloriaHip(0)
    assert ei.value.code == -2


def test_product_does_not_reference_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "floria_amd")):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libfloria_oracle" not in txt and "oracle/" not in txt.replace("the host oracle", ""), f


def test_block_ranges_match_oracle_and_vcf_fixture(hip_lib, oracle_mod):
    pos = np.load(os.path.join(ROOT, "tests", "golden", "test_vcf_positions.npy"))
    for L in (10000, 5000, 500, 123):
        s1, e1 = hip_lib.get_range_with_lengths(pos, L)
        s2, e2 = oracle_mod.block_ranges(pos, L)
        assert np.array_equal(s1, s2) and np.array_equal(e1, e2)
    assert len(hip_lib.get_range_with_lengths(pos, 10000)[0]) == 17
    rng = np.random.default_rng(5)
    for _ in range(20):
        g = np.cumsum(rng.geometric(1 / rng.integers(20, 400), size=int(rng.integers(2, 400))))
        L = int(rng.integers(50, 5000))
        d = float(rng.choice([0.0005, 0.005, 0.02]))
        a = hip_lib.get_range_with_lengths(g, L, None, d)
        b = oracle_mod.block_ranges(g, L, None, d)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_block_ranges_rejects_decreasing_positions(hip_lib):
    # utils_frags.rs:422-425: the reference logs "VCF malformed" and exits
    with pytest.raises(hip_lib.FloriaHipError) as ei:
        hip_lib.get_range_with_lengths(np.array([10, 20, 15, 40, 50]), 100)
    assert ei.value.code == -1 and "not increasing" in str(ei.value)


def test_lpt_assignment_is_balanced_and_deterministic():
    from floria_amd.shard import lpt_assign, my_items
    costs = np.random.default_rng(1).integers(1, 100, size=101)
    o = lpt_assign(costs, 4)
    assert np.array_equal(o, lpt_assign(costs, 4))
    loads = [costs[my_items(o, r)].sum() for r in range(4)]
    assert max(loads) - min(loads) <= costs.max()
    assert sorted(np.concatenate([my_items(o, r) for r in range(4)])) == list(range(101))


def test_ctypes_mirrors_match_the_header_layout(tmp_path):
    """sizeof / offsetof of every struct in include/floria_hip.h, measured by gcc, equal the ctypes mirrors in floria_amd/_capi.py
    (a drifted mirror would read garbage through the C ABI without any error)."""
    import ctypes as C
    import subprocess
    from floria_amd import _capi as capi
    pairs = {"floria_pileup": capi.CPileup, "floria_params": capi.CParams, "floria_block_result": capi.CBlockResult,
             "floria_groups": capi.CGroups, "floria_ranges": capi.CRanges, "floria_timing": capi.CTiming, "floria_hap_graph": capi.CHapGraph}
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "floria_hip.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for f, _ in cls._fields_:
            lines.append(f'  printf(" %zu", offsetof({cname}, {f}));')
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-std=c11", "-I", inc, str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    for line in out:
        name, size, *offs = line.split()
        cls = pairs[name]
        assert int(size) == C.sizeof(cls), f"{name}: header {size} B, ctypes {C.sizeof(cls)} B"
        assert [int(o) for o in offs] == [getattr(cls, f).offset for f, _ in cls._fields_], name


def test_rust_binding_sketch_covers_every_declared_symbol():
    # INTEGRATION.md §2 is what a Rust maintainer would paste into ffi.rs; no rustc exists here, so the least it must do is name every entry point the header
    # declares (a host following a sketch without floria_hip_hap_graph_free leaks every graph) and nothing the header does not have.
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "floria_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(floria_hip_[a-z0-9_]+)\s*\(", code))
    sketch = open(os.path.join(root, "INTEGRATION.md")).read()
    bound = set(re.findall(r"\bpub fn (floria_hip_[a-z0-9_]+)\s*\(", sketch))
    assert declared - bound == set(), f"not bound in INTEGRATION.md: {sorted(declared - bound)}"
    assert bound - declared == set(), f"bound but not declared: {sorted(bound - declared)}"


def test_header_documents_every_option_key():
    # floria_hip_set_option takes string keys: the header's comment is the only place a host learns them.  Every key the library accepts must be named there
    # (and the one key that changes results, "arith", must be described as such).
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "floria_amd", "csrc", "floria_hip.hip")).read()
    body = src[src.index("int floria_hip_set_option("):]
    body = body[:body.index("\n}\n")]
    keys = set(re.findall(r'k == "([a-z_0-9]+)"', body))
    assert {"arith", "speculate", "groups"} <= keys
    header = open(os.path.join(root, "include", "floria_hip.h")).read()
    doc = header[header.index('The one option that selects WHICH function is computed'):header.index("int  floria_hip_set_option(")]
    missing = sorted(k for k in keys if f'"{k}"' not in doc)
    assert not missing, f"option keys the header does not mention: {missing}"
