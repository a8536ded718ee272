"""The C++ host mirror of the reference's call sites (floria_amd/host/floria_host.{hpp,cpp}) driven like floria.rs drives
the reference: Frags -> sort -> generate_hap_graph -> process_reads_for_final_parts, compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from floria_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def build_driver():
    subprocess.check_call(["make", "-C", CPP, "-B", "host_driver"], stdout=subprocess.DEVNULL)
    return os.path.join(CPP, "host_driver")


def test_cpp_host_compiles_against_the_c_abi(hip_lib):
    # `-m "not gpu"`: the mirror is plain C++17 on top of include/floria_hip.h and links with g++ (no HIP toolchain needed)
    assert os.path.exists(build_driver())


def write_fixture(path, contig):
    p = contig.pileup
    with open(path, "w") as f:
        f.write(f"{p.n_reads} {len(contig.snp_pos)}\n")
        f.write(" ".join(str(int(x)) for x in contig.snp_pos) + "\n")
        for r in range(p.n_reads):
            s, a, q = p.read(r)
            f.write(str(len(s)) + " " + " ".join(f"{int(x)} {int(y)} {int(z)}" for x, y, z in zip(s, a, q)) + "\n")


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,idx,scale", [(1, 0, 1.0), (4, 2, 0.5)])
def test_cpp_host_matches_oracle(hip_lib, oracle_mod, tmp_path, cfg, idx, scale):
    C = synth.CONFIGS[cfg]
    c = synth.make_config_contig(cfg, idx, scale)
    fx = tmp_path / "pileup.txt"
    write_fixture(fx, c)
    eps = 0.03125
    out = subprocess.run([build_driver(), str(fx), str(eps), str(C["block_length"]), str(C["max_ploidy"]), str(C["beam"])],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    nodes, groups, err, hapq = [], [], None, None
    for line in out.stdout.splitlines():
        t = line.split()
        if t[0] == "NODE":
            i_reads, i_out, i_in = t.index("reads"), t.index("out"), t.index("in")
            nodes.append(dict(col=int(t[1]), row=int(t[2]), id=int(t[3]), lo=int(t[4]), hi=int(t[5]), cov=float(t[6]),
                              reads=[int(x) for x in t[i_reads + 1:i_out]],
                              out=[(int(x.split(":")[0]), float(x.split(":")[1])) for x in t[i_out + 1:i_in]],
                              inn=[(int(x.split(":")[0]), float(x.split(":")[1])) for x in t[i_in + 1:]]))
        elif t[0] == "GROUP":
            groups.append(((int(t[1]), int(t[2])), [int(x) for x in t[4:]]))
        elif t[0] == "ERRCHECK":
            err = t
        elif t[0] == "HAPQ":
            hapq = [int(x) for x in t[1:]]
    assert err and err[1] == "-1" and "not increasing" in " ".join(err)          # utils_frags.rs:422-425
    # ---- S1 + hap graph vs the oracle -------------------------------------------------------------------------------
    s, e = oracle_mod.block_ranges(c.snp_pos, C["block_length"])
    ro = oracle_mod.phase_blocks(c.pileup, s, e, oracle_mod.make_params(eps, C["max_ploidy"], C["beam"]), threads=4)
    cov, ew = oracle_mod.hap_graph(c.pileup, s, e, ro)
    exp_nodes, cols = [], []
    for b in range(len(s)):
        if ro.best_ploidy[b]:
            cols.append(b)
            for k, part in enumerate(ro.partitions(b)):
                exp_nodes.append((len(cols) - 1, k, int(s[b]), int(e[b]), [int(x) for x in part]))
    assert len(nodes) == len(exp_nodes)
    for i, (n, x) in enumerate(zip(nodes, exp_nodes)):
        assert (n["col"], n["row"], n["lo"], n["hi"], n["reads"]) == x and n["id"] == i
        assert n["cov"] == cov[i]
    # edges: row-major p1 x p2 matrices of consecutive columns, kept when >= 2 (graph_processing.rs:51)
    off = 0
    by_col = {}
    for n in nodes:
        by_col.setdefault(n["col"], []).append(n)
    for ci in range(len(cols) - 1):
        p1, p2 = int(ro.best_ploidy[cols[ci]]), int(ro.best_ploidy[cols[ci + 1]])
        m = ew[off:off + p1 * p2].reshape(p1, p2); off += p1 * p2
        for j in range(p1):
            assert by_col[ci][j]["out"] == [(l, float(m[j, l])) for l in range(p2) if m[j, l] >= 2]
        for l in range(p2):
            assert by_col[ci + 1][l]["inn"] == [(j, float(m[j, l])) for j in range(p1) if m[j, l] >= 2]
    # ---- S2 vs the oracle ------------------------------------------------------------------------------------------------
    g_in = [np.array(n["reads"], np.uint32) for n in nodes if n["reads"]]
    r_in = [(n["lo"], n["hi"]) for n in nodes if n["reads"]]
    go = oracle_mod.reassign(c.pileup, g_in, r_in, eps)
    assert go.n_groups == len(groups)
    for k, (rng, reads) in enumerate(groups):
        assert rng == (int(go.range[k][0]), int(go.range[k][1])) and reads == [int(x) for x in go.group(k)]
    # ---- get_hapq on the reassigned haplosets vs the oracle (part_block_manip.rs:517-616) ------------------------------------------
    ohq, _, _ = oracle_mod.hapq(c.pileup, [go.group(k) for k in range(go.n_groups)], [tuple(int(x) for x in go.range[k]) for k in range(go.n_groups)],
                                c.snp_pos, C["block_length"])
    assert hapq == [int(x) for x in ohq]
